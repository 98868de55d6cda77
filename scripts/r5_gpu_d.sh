#!/bin/bash
# round 5, fourth GPU run: the final row-team geometry table -- parity at every block length, the suite's multi-rank tests, the latency curve before / after
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5d; mkdir -p $O
timeout 2400 python scripts/r5_peer_geometry_parity.py > $O/geometry_parity.log 2>&1
timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=15 -k "row_team or two_ranks_one_gpu or fuzz_row_teams or fuzz_ranks_as_processes or self_launched or four_ranks or rank_failure or resume_into or saved_state" > $O/pytest_ranks.log 2>&1; echo "rc $?" >> $O/pytest_ranks.log
D=off,0,2000,4000,6000,8000,12000,16000
timeout 600 python tests/tools/row_team_sweep.py --m 4096 --n 1048576 --two-sweeps --repeat 2 --delays $D --geoms 2:2:0:2:4:4,default > $O/sweep_2048.jsonl 2> $O/sweep_2048.err
timeout 600 python tests/tools/row_team_sweep.py --m 8192 --n 524288 --two-sweeps --repeat 2 --delays $D --geoms 2:2:0:2:2:4,default > $O/sweep_4096.jsonl 2> $O/sweep_4096.err
timeout 600 python tests/tools/row_team_sweep.py --m 16384 --n 262144 --two-sweeps --repeat 2 --delays $D --geoms 1:2:0:2:2:4,default > $O/sweep_8192.jsonl 2> $O/sweep_8192.err
timeout 600 python tests/tools/row_team_sweep.py --m 32768 --n 131072 --two-sweeps --repeat 2 --delays $D --geoms 1:2:0:2:1:4,1:2:1:2:1:4,default > $O/sweep_16384.jsonl 2> $O/sweep_16384.err
timeout 600 python tests/tools/row_team_sweep.py --m 16384 --n 262144 --ranks 4 --repeat 1 --delays off,0,4000,8000 --geoms 2:2:0:2:2:4,default > $O/sweep_4x4096.jsonl 2> $O/sweep_4x4096.err
tail -3 $O/geometry_parity.log; tail -25 $O/pytest_ranks.log; wc -l $O/*.jsonl
