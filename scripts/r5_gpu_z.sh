#!/bin/bash
# round 5, run Z: ZeroFPR with three trial points per sweep -- the records of profiles/ that it changes
O=gpurun_out/r5
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "zerofpr or newton_family" 2>&1 | grep -v amdgpu.ids | tail -5 > $O/pytest_zerofpr.log
tail -2 $O/pytest_zerofpr.log
python scripts/bench_panoc.py --algo zerofpr --steps 23 --warmup 0 2>/dev/null | grep '^{' > $O/bench_zerofpr.json
python scripts/bench_panoc.py --algo zerofpr --trio-trials 0 --steps 23 --warmup 0 2>/dev/null | grep '^{' > $O/bench_zerofpr_two_points.json
python scripts/bench_panoc.py --algo zerofpr --steps 60 --warmup 0 2>/dev/null | grep '^{' > $O/bench_zerofpr_60_iterations.json
python scripts/r5_pair_sweep_rate.py --reps 12 2>&1 | grep '^{' > $O/pair_sweep_rate.log
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -d $O/prof_zerofpr -- python3 scripts/bench_panoc.py --algo zerofpr --steps 23 --warmup 0 > $O/prof_zerofpr.log 2>&1
python scripts/rocpd_summary.py $O/prof_zerofpr/*/*_results.db > $O/prof_zerofpr.md 2>&1
rm -rf $O/prof_zerofpr
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
for f in bench_zerofpr bench_zerofpr_two_points bench_zerofpr_60_iterations; do python - $O/$f.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 2), "it/s", round(d["A_passes_per_step"], 3), "reads", d["accepted_tau_histogram"], d["pair_sweeps"], d["trio_sweeps"])
P
done
cat $O/pair_sweep_rate.log
python - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/r5/bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("traffic_stale"))
print(d["config"].get("also"))
P
head -30 $O/prof_zerofpr.md | cut -c1-200
