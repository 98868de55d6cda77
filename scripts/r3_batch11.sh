set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3_b11
mkdir -p $O
B="--steps 20 --warmup 3 --no-cpu-baseline --no-also"
for r in 1 2; do
  for tree in r3 r2; do
    if [ $tree = r2 ]; then cd $GRAFT_REPO_ROOT/_r2tree; else cd $GRAFT_REPO_ROOT; fi
    python bench.py --m 50000 --n 84000 $B > $O/odd50000_${tree}_$r.json 2>/dev/null
    python bench.py --m 100000 --n 84000 $B > $O/odd100000_${tree}_$r.json 2>/dev/null
    python bench.py --m 65536 --n 131072 --dtype f64 $B > $O/f64long_${tree}_$r.json 2>/dev/null
    python bench.py --m 131072 --n 131072 $B > $O/long_${tree}_$r.json 2>/dev/null
    python bench.py --m 65536 --n 262144 $B > $O/long65536_${tree}_$r.json 2>/dev/null
  done
  cd $GRAFT_REPO_ROOT
  for lc in 1 32 64; do for m in 3072 4096 5120 6144; do
    PG_TN_LINE_COLS=$lc python bench.py --m $m --n $((2147483648 / m / 4 * 4 / 4)) $B > $O/tnc${m}_lc${lc}_$r.json 2>/dev/null
  done; done
done
