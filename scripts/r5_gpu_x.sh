#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err
timeout 3000 python -m pytest tests -m gpu -q --durations=30 > gpurun_out/r5_gpu_suite_final.log 2>&1; echo "rc $?" >> gpurun_out/r5_gpu_suite_final.log
tail -5 gpurun_out/r5_gpu_suite_final.log; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5/bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("traffic_stale"))
print(d["config"].get("also")); print(len([k for k,v in d["config"].items() if not isinstance(v,(dict,list))]), d["job"]["wall"])
PY
