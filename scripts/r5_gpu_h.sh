#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5h; mkdir -p $O
bash scripts/collect_round5_profiles.sh bench > $O/collect_bench.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "zerofpr_two_trial or two_point_sweep" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -5 $O/pytest.log; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5/bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("traffic_stale"))
print(d["config"].get("also"))
for a in d.get("also",[]): print(a.get("label"), a.get("value"), (a.get("roofline") or {}).get("frac"), a.get("error"))
print(d.get("cpu_baseline",{}).get("value"))
PY
