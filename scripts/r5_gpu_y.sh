#!/bin/bash
# round 5, run Y: the three-point sweep -- parity, rate at config 4's size, ZeroFPR on it
mkdir -p gpurun_out/r5y
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "three_point_sweep or two_point_sweep or zerofpr" 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r5y/pytest.log
echo "pytest rc ${PIPESTATUS[0]}" >> gpurun_out/r5y/pytest.log
tail -4 gpurun_out/r5y/pytest.log
timeout 600 python scripts/r5_pair_sweep_rate.py --reps 12 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5y/rate.log
for t in 1 0; do
  timeout 600 python scripts/bench_panoc.py --algo zerofpr --steps 23 --warmup 0 --trio-trials $t 2>&1 | grep '^{' | tee gpurun_out/r5y/zerofpr_trio$t.json | cut -c1-900
done
timeout 600 python scripts/bench_panoc.py --algo zerofpr --steps 60 --warmup 0 2>&1 | grep '^{' | tee gpurun_out/r5y/zerofpr_60.json | cut -c1-900
