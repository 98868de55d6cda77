set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_b12
mkdir -p $O
B="--steps 20 --warmup 3 --no-cpu-baseline --no-also --sweeps two"
for r in 1 2; do for lc in 32 1; do
  PG_T_LINE_COLS=$lc python bench.py $B > $O/headline_lc${lc}_$r.json 2>/dev/null
  PG_T_LINE_COLS=$lc python bench.py --m 2048 --n 1048576 $B > $O/m2048_lc${lc}_$r.json 2>/dev/null
  PG_T_LINE_COLS=$lc python bench.py --m 8192 --n 262144 $B > $O/m8192_lc${lc}_$r.json 2>/dev/null
  PG_T_LINE_COLS=$lc python bench.py --m 131072 --n 131072 $B > $O/m131072_lc${lc}_$r.json 2>/dev/null
  PG_T_LINE_COLS=$lc python bench.py --m 2048 --n 1048576 --force-comm --sharding rows $B > $O/rows2048_lc${lc}_$r.json 2>/dev/null
done; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemv or least_squares or sharded_payload or two_ranks_one_gpu or lasso_small_known or headline_size_properties" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
