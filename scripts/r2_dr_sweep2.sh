#!/bin/bash
export PG_TUNE=1  # the library reads its tuning variables only when this is set
# dr_step: non-temporal vs regular stores of y (and of r / z / res), a few geometries each
for nt in 1 0; do for g in 1024x1x2 512x2x2 1024x2x1; do
  PG_DR_NT=$nt PG_DR_STEP_GEOM=$g python tests/tools/bench_dr.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())['modes']
print('nt=$nt', '$g', 'x_y_only %.4f ms (%.0f GB/s)' % (d['x_y_only']['roofline']['avg_launch_ms'], d['x_y_only']['roofline']['achieved']), 'full_state %.4f ms (%.0f GB/s)' % (d['full_state']['roofline']['avg_launch_ms'], d['full_state']['roofline']['achieved']))
"
done; done
