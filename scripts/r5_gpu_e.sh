#!/bin/bash
# round 5, fifth GPU run: ZeroFPR with two trial points per sweep (parity + rate), the Newton family to its stopping rule, the rest of the multi-rank tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=10 -k "zerofpr or panocplus or image_slab or panoc_single_sweep or fused_single_sweep" > $O/pytest_newton.log 2>&1; echo "rc $?" >> $O/pytest_newton.log
for pt in 1 0; do for r in 1 2; do
  timeout 600 python scripts/bench_panoc.py --algo zerofpr --pair-trials $pt --steps 23 --warmup 0 > $O/zerofpr_pair${pt}_r$r.json 2> $O/zerofpr_pair${pt}_r$r.err
done; done
timeout 600 python scripts/bench_panoc.py --algo zerofpr --pair-trials 1 --steps 40 --warmup 23 > $O/zerofpr_pair1_late.json 2> /dev/null
timeout 600 python scripts/bench_panoc.py --algo zerofpr --pair-trials 0 --steps 40 --warmup 23 > $O/zerofpr_pair0_late.json 2> /dev/null
timeout 1800 python tests/tools/newton_stop_rule.py --n 32768 --tols 1e-3,3e-4,1e-4 > $O/newton_stop_rule.jsonl 2> $O/newton_stop_rule.err
timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q --durations=15 -k "self_launched or four_ranks or rank_failure or fuzz_row_teams or fuzz_ranks_as_processes or resume_into or saved_state or team_sweep or graph_replay" > $O/pytest_ranks.log 2>&1; echo "rc $?" >> $O/pytest_ranks.log
tail -15 $O/pytest_newton.log; cat $O/zerofpr_pair*.json | cut -c1-400; cat $O/newton_stop_rule.jsonl; tail -30 $O/pytest_ranks.log
