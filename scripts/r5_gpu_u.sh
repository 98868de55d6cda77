#!/bin/bash
# Float64 row teams under injected latency (the Float32 table by row groups: 2 x 1024 / 2 x 8192 Float64 rows = 8 / 64 row groups per rank)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5u; mkdir -p $O
D=off,0,4000,8000,12000,16000
timeout 600 python tests/tools/row_team_sweep.py --dtype f64 --m 2048 --n 1048576 --two-sweeps --delays $D --geoms 2:2:0:2:4:4,default > $O/sweep_f64_1024.jsonl 2> $O/err1
timeout 600 python tests/tools/row_team_sweep.py --dtype f64 --m 16384 --n 131072 --two-sweeps --delays $D --geoms 1:2:0:2:1:4,default > $O/sweep_f64_8192.jsonl 2> $O/err2
python scripts/r5_sweep_table.py $O/sweep_f64_1024.jsonl $O/sweep_f64_8192.jsonl; tail -2 $O/err1
