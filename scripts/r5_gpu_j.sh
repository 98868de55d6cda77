#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5j; mkdir -p $O
run() { # label, env...
  lbl=$1; shift
  env "$@" python bench.py --gpus 2 --share-device --backend gloo --sharding rows --row-teams --m 32768 --n 131072 --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-row-teams 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$lbl', d['value'], d['roofline']['avg_launch_ms'], c.get('a_passes_per_step'), c.get('row_team_stats'))"
}
for r in 1 2 3; do
  run default PG_TUNE=0
  run r4geom PG_TUNE=1 PG_TNP_C=1 PG_TNP_LAG=2 PG_TNP_LAGR=0 PG_TNP_PF=2 PG_TNP_WGS=1 PG_TNP_W=4
  run lagr1 PG_TUNE=1 PG_TNP_C=1 PG_TNP_LAG=2 PG_TNP_LAGR=1 PG_TNP_PF=2 PG_TNP_WGS=1 PG_TNP_W=4
done
python tests/tools/row_team_sweep.py --m 32768 --n 131072 --repeat 2 --delays off --geoms 1:2:0:2:1:4,1:2:1:2:1:4,default 2>/dev/null | cut -c1-330
