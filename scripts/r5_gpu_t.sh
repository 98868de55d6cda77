#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5t; mkdir -p $O
run() { lbl=$1; shift; env PG_TUNE=1 "$@" python bench.py --m 131072 --n 131072 --steps 30 --warmup 5 --no-also --no-cpu-baseline 2>$O/$lbl.err | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t else {}
print('$lbl', d.get('value'), (d.get('roofline') or {}).get('avg_launch_ms'), (d.get('roofline') or {}).get('frac'), d.get('config',{}).get('sweep_fallbacks'), d.get('error'))"; }
for r in 1 2; do
run base_w4
run w2 PG_TNT_WAVES=2 PG_TNT_U=16
run w2_lagr2 PG_TNT_WAVES=2 PG_TNT_U=16 PG_TNT_LAGR=2
done
PG_TUNE=1 PG_TNT_WAVES=2 PG_TNT_U=16 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_sweep_kernels_steady_state_iterates_match_oracle and (65536 or 131072)" 2>&1 | tail -2
tail -3 $O/w2.err
