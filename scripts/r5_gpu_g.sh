#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5g; mkdir -p $O
python scripts/r5_pair_sweep_rate.py --variants 8:2 2>&1 | grep -v amdgpu.ids > $O/pair_sweep_rate.log
for pt in 1 0; do
  timeout 600 python scripts/bench_panoc.py --algo zerofpr --pair-trials $pt --steps 23 --warmup 0 > $O/zerofpr_$pt.json 2> /dev/null
  timeout 600 python scripts/bench_panoc.py --algo zerofpr --pair-trials $pt --steps 60 --warmup 0 > $O/zerofpr_${pt}_60.json 2> /dev/null
done
timeout 3000 python -m pytest tests -m gpu -q --durations=30 > $O/gpu_suite.log 2>&1; echo "rc $?" >> $O/gpu_suite.log
for f in $O/zerofpr_*.json; do echo $f; cut -c1-500 $f; done; cat $O/pair_sweep_rate.log; tail -50 $O/gpu_suite.log
