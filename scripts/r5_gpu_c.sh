#!/bin/bash
# round 5, third GPU run: one- and two-wave workgroups for short row blocks (a wave holds the whole / half column), the LDS-free U = 16 form
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c; mkdir -p $O
timeout 1500 python scripts/r5_peer_geometry_parity.py > $O/geometry_parity.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "row_team_iterates or resume_into_a_batched or saved_state" > $O/pytest_row_team.log 2>&1; echo "rc $?" >> $O/pytest_row_team.log
D=off,0,2000,4000,6000,8000,12000,16000
timeout 600 python tests/tools/row_team_sweep.py --m 4096 --n 1048576 --two-sweeps --delays $D --geoms 2:2:0:2:4:4,4:2:2:2:2:4,2:3:3:2:3:4,2:2:0:2:4:1,2:2:1:2:4:1,2:2:2:2:4:1,2:2:3:2:4:1,2:2:2:1:4:1 > $O/sweep_2048.jsonl 2> $O/sweep_2048.err
timeout 600 python tests/tools/row_team_sweep.py --m 8192 --n 524288 --two-sweeps --delays $D --geoms 2:2:0:2:2:4,2:2:1:2:2:4,2:2:2:2:2:4,1:2:0:2:4:1,1:2:1:2:4:1,1:2:2:2:4:1,2:2:0:2:2:2,2:2:2:2:2:2 > $O/sweep_4096.jsonl 2> $O/sweep_4096.err
timeout 600 python tests/tools/row_team_sweep.py --m 16384 --n 262144 --two-sweeps --delays $D --geoms 1:2:0:2:2:4,1:2:1:2:2:4,1:2:0:2:2:2,1:2:1:2:2:2,1:2:2:2:2:2 > $O/sweep_8192.jsonl 2> $O/sweep_8192.err
timeout 600 python tests/tools/row_team_sweep.py --m 32768 --n 131072 --two-sweeps --delays $D --geoms 1:2:0:2:1:4,1:2:1:2:1:4,1:2:2:2:1:4,1:0:2:2:1:4 > $O/sweep_16384.jsonl 2> $O/sweep_16384.err
for round in 1 2; do for v in "2 0" "2 1" "0 2"; do set -- $v
  PG_TUNE=1 PG_TNT_LAG=$1 PG_TNT_LAGR=$2 timeout 300 python bench.py --m 131072 --n 131072 --steps 30 --warmup 5 --no-also --no-cpu-baseline > $O/bench_long_$1_$2_r$round.json 2> $O/bench_long_$1_$2_r$round.err
done; done
tail -3 $O/geometry_parity.log; tail -2 $O/pytest_row_team.log; wc -l $O/*.jsonl
