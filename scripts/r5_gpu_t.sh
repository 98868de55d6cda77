#!/bin/bash
# round 5: the three-point sweep -- parity and rate (iterating on the kernel)
mkdir -p gpurun_out/r5t
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "three_point_sweep or three" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 600 python scripts/r5_pair_sweep_rate.py --reps 12 2>&1 | grep '^{' | tee gpurun_out/r5t/rate.log
timeout 600 python scripts/bench_panoc.py --algo zerofpr --steps 23 --warmup 0 2>&1 | grep '^{' | tee gpurun_out/r5t/zerofpr.json | cut -c1-330
