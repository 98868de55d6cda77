set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_b6
mkdir -p $O
python scripts/r3_mid_sweep.py ab > $O/mid_ab.log 2>&1
python scripts/r2_tn_check.py check > $O/tn_check.log 2>&1; tail -n 2 $O/tn_check.log
for r in 1 2; do for lc in 32 1; do
  PG_TN_LINE_COLS=$lc python bench.py --m 2048 --n 1048576 --steps 30 --warmup 5 --no-cpu-baseline --no-also > $O/short_lc${lc}_$r.json 2>/dev/null
  PG_TN_LINE_COLS=$lc python bench.py --m 4096 --n 524288 --steps 30 --warmup 5 --no-cpu-baseline --no-also > $O/m4096_lc${lc}_$r.json 2>/dev/null
  PG_TN_LINE_COLS=$lc python bench.py --m 3072 --n 699048 --steps 30 --warmup 5 --no-cpu-baseline --no-also > $O/m3072_lc${lc}_$r.json 2>/dev/null
  PG_TN_LINE_COLS=$lc python bench.py --m 1024 --n 1048576 --steps 30 --warmup 5 --no-cpu-baseline --no-also > $O/m1024_lc${lc}_$r.json 2>/dev/null
done; done
