set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_b10
mkdir -p $O
B="--steps 20 --warmup 3 --no-cpu-baseline --no-also"
for r in 1 2; do
for v in "default:" "plain:PG_TN_TEAM_PLAIN=1" "lc1:PG_TN_LINE_COLS=1" "plain_lc1:PG_TN_TEAM_PLAIN=1 PG_TN_LINE_COLS=1"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs python bench.py --m 50000 --n 84000 $B > $O/odd50000_${name}_$r.json 2>/dev/null
  env $envs python bench.py --m 100000 --n 84000 $B > $O/odd100000_${name}_$r.json 2>/dev/null
  env $envs python bench.py --m 65536 --n 131072 --dtype f64 $B > $O/f64long_${name}_$r.json 2>/dev/null
  env $envs python bench.py --m 131072 --n 131072 $B > $O/long_${name}_$r.json 2>/dev/null
done; done
