#!/bin/bash
export PG_TUNE=1  # the library reads its tuning variables only when this is set
# Sweep of the Douglas-Rachford stepping kernel's launch geometry (PG_DR_STEP_GEOM = threads x blocks/CU x vectors/trip);
# one process per setting (the knob is read once).  Output: one line per setting with the kernel's average launch time.
for g in 1024x1x2 1024x1x1 1024x1x4 1024x2x1 1024x2x2 512x2x2 512x2x4 512x4x1 512x4x2 256x4x2 256x4x4 256x8x1 256x8x2 256x8x4; do
  PG_DR_STEP_GEOM=$g python tests/tools/bench_dr.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())['modes']
print('$g', 'x_y_only %.4f ms (%.0f GB/s)' % (d['x_y_only']['roofline']['avg_launch_ms'], d['x_y_only']['roofline']['achieved']), 'full_state %.4f ms (%.0f GB/s)' % (d['full_state']['roofline']['avg_launch_ms'], d['full_state']['roofline']['achieved']), 'block16 %.4f ms = %.0f it/s' % (d['device_loop_block16']['roofline']['avg_launch_ms'], d['device_loop_block16']['value']), 'block8 %.4f ms = %.0f it/s' % (d['device_loop_block8']['roofline']['avg_launch_ms'], d['device_loop_block8']['value']))
"
done
