#!/bin/bash
# round 5, first GPU run: the refactored team sweep (register lag + injector) -- existing row-team tests, new geometries' parity, first rate sweep
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5a; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "row_team_iterates or graph_replay_of_a_team or team_sweep_timeout" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 1500 python scripts/r5_peer_geometry_parity.py > $O/geometry_parity.log 2>&1
D=off,0,2000,4000,6000,8000,12000,16000
timeout 600 python tests/tools/row_team_sweep.py --m 4096 --n 1048576 --two-sweeps --delays $D --geoms 2:2:0:2:4,2:2:1:2:3,2:2:1:1:3,2:2:2:1:3,2:2:2:2:3,2:3:2:2:3,2:3:3:2:3,2:3:3:2:2,2:0:4:2:2,4:2:0:2:2,4:2:2:2:2,4:2:3:2:1 > $O/sweep_2048.jsonl 2> $O/sweep_2048.err
timeout 600 python tests/tools/row_team_sweep.py --m 8192 --n 524288 --two-sweeps --delays $D --geoms 2:2:0:2:2,2:2:1:2:2,2:2:2:2:2,2:2:2:2:1,1:2:0:2:4,1:2:1:2:3,1:3:2:2:3,1:3:3:2:2,1:2:2:1:2 > $O/sweep_4096.jsonl 2> $O/sweep_4096.err
timeout 600 python tests/tools/row_team_sweep.py --m 16384 --n 262144 --two-sweeps --delays $D --geoms 1:2:0:2:2,1:2:1:2:2,1:2:1:2:1,1:4:2:2:1,1:4:3:2:1 > $O/sweep_8192.jsonl 2> $O/sweep_8192.err
timeout 600 python tests/tools/row_team_sweep.py --m 32768 --n 131072 --two-sweeps --delays $D --geoms 1:2:0:2:1,1:2:1:2:1,1:2:2:2:1,1:2:2:1:1,1:2:3:1:1 > $O/sweep_16384.jsonl 2> $O/sweep_16384.err
tail -3 $O/pytest.log; tail -5 $O/geometry_parity.log; wc -l $O/*.jsonl; tail -2 $O/*.err
