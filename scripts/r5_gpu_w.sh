#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5w; mkdir -p $O
bash scripts/collect_round5_profiles.sh pmc > $O/collect_pmc.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err
timeout 3000 python -m pytest tests -m gpu -q --durations=30 > gpurun_out/r5_gpu_suite_final.log 2>&1; echo "rc $?" >> gpurun_out/r5_gpu_suite_final.log
tail -6 gpurun_out/r5_gpu_suite_final.log; tail -2 $O/collect_pmc.log
