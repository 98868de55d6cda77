set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_b7
mkdir -p $O
for r in 1 2 3; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --sustain 0 > $O/headline_tn_$r.json 2>/dev/null
  PG_TN_KERNEL=mid PG_TN_U=16 PG_TN_C=2 PG_TN_WAVES=4 PG_TN_DB=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --sustain 0 > $O/headline_tnm2_$r.json 2>/dev/null
  PG_TN_KERNEL=mid PG_TN_U=16 PG_TN_C=2 PG_TN_WAVES=4 PG_TN_DB=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --sustain 0 > $O/headline_tnm1_$r.json 2>/dev/null
done
