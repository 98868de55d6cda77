set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_b5
mkdir -p $O
python scripts/r2_tn_check.py check > $O/tn_check.log 2>&1; tail -2 $O/tn_check.log
python scripts/r3_mid_sweep.py check > $O/mid_check.log 2>&1; tail -2 $O/mid_check.log
for r in 1 2; do
for lc in 32 1; do
  PG_TN_LINE_COLS=$lc python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --sustain 0 > $O/headline_lc${lc}_$r.json 2>/dev/null
  PG_TN_LINE_COLS=$lc python bench.py --workload config2 --steps 50 --warmup 5 --no-cpu-baseline --no-also > $O/config2_lc${lc}_$r.json 2>/dev/null
  PG_TN_LINE_COLS=$lc python bench.py --m 131072 --n 131072 --steps 20 --warmup 3 --no-cpu-baseline --no-also > $O/long_lc${lc}_$r.json 2>/dev/null
  PG_TN_LINE_COLS=$lc python bench.py --m 2048 --n 1048576 --steps 30 --warmup 5 --no-cpu-baseline --no-also > $O/short_lc${lc}_$r.json 2>/dev/null
  PG_TN_LINE_COLS=$lc python bench.py --m 10240 --n 262144 --steps 30 --warmup 5 --no-cpu-baseline --no-also > $O/m10240_lc${lc}_$r.json 2>/dev/null
  PG_TN_LINE_COLS=$lc python bench.py --m 4096 --n 524288 --steps 30 --warmup 5 --no-cpu-baseline --no-also > $O/m4096_lc${lc}_$r.json 2>/dev/null
  PG_TN_LINE_COLS=$lc python bench.py --m 24576 --n 87380 --steps 30 --warmup 5 --no-cpu-baseline --no-also > $O/m24576_lc${lc}_$r.json 2>/dev/null
done; done
python scripts/r3_mid_sweep.py ab 8192 262144 10240 209712 12288 174760 24576 87380 > $O/mid_ab.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep_kernels_steady_state or fixed_step_iterate_sequence or lasso_small_known or config2_iterates" > $O/pytest_sweeps.log 2>&1; tail -3 $O/pytest_sweeps.log
