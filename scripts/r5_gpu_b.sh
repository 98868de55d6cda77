#!/bin/bash
# round 5, second GPU run: poll-ahead (OPT 1) and barrier-free dot exchange (OPT 2) of the team sweep -- parity, row-team latency sweep, single-device team sweep
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5b; mkdir -p $O
# parity of the single-device team sweep under the options (steady-state iterates against the oracle: 65536 / 131072 rows are gemv_tnt<16,...>)
for v in "0 0" "0 1" "0 2" "0 3" "1 0" "1 3" "2 3" "1 2"; do set -- $v
  PG_TUNE=1 PG_TNT_LAGR=$1 PG_TNT_OPT=$2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_sweep_kernels_steady_state_iterates_match_oracle and (65536 or 131072)" > $O/pytest_tnt_$1_$2.log 2>&1; echo "LAGR=$1 OPT=$2 rc $? $(tail -1 $O/pytest_tnt_$1_$2.log)" >> $O/pytest_tnt.log
done
timeout 1500 python scripts/r5_peer_geometry_parity.py > $O/geometry_parity.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "row_team_iterates" > $O/pytest_row_team.log 2>&1; echo "rc $?" >> $O/pytest_row_team.log
D=off,0,2000,4000,6000,8000,12000,16000
timeout 600 python tests/tools/row_team_sweep.py --m 4096 --n 1048576 --delays $D --geoms 2:2:0:2:4:0,2:2:0:2:4:1,2:2:0:2:4:2,2:2:0:2:4:3,2:2:1:2:3:3,2:3:3:2:3:0,2:3:3:2:3:3,4:2:0:2:2:0,4:2:0:2:2:3,4:2:2:2:2:0,4:2:2:2:2:3 > $O/sweep_2048.jsonl 2> $O/sweep_2048.err
timeout 600 python tests/tools/row_team_sweep.py --m 8192 --n 524288 --delays $D --geoms 2:2:0:2:2:0,2:2:0:2:2:3,2:2:1:2:2:0,2:2:1:2:2:3 > $O/sweep_4096.jsonl 2> $O/sweep_4096.err
timeout 600 python tests/tools/row_team_sweep.py --m 16384 --n 262144 --delays $D --geoms 1:2:0:2:2:0,1:2:0:2:2:3,1:2:1:2:2:0,1:2:1:2:2:3 > $O/sweep_8192.jsonl 2> $O/sweep_8192.err
timeout 600 python tests/tools/row_team_sweep.py --m 32768 --n 131072 --delays $D --geoms 1:2:0:2:1:0,1:2:0:2:1:3,1:2:1:2:1:0,1:2:1:2:1:3,1:2:2:2:1:0,1:2:2:2:1:3 > $O/sweep_16384.jsonl 2> $O/sweep_16384.err
# the single-device team sweep (131072 x 131072: config 5's block under column shards), two rounds interleaved
for round in 1 2; do for v in "0 0" "0 1" "0 2" "0 3" "1 0" "1 3" "2 3" "1 2"; do set -- $v
  PG_TUNE=1 PG_TNT_LAGR=$1 PG_TNT_OPT=$2 timeout 300 python bench.py --m 131072 --n 131072 --steps 30 --warmup 5 --no-also --no-cpu-baseline > $O/bench_long_$1_$2_r$round.json 2> $O/bench_long_$1_$2_r$round.err
done; done
cat $O/pytest_tnt.log; tail -3 $O/geometry_parity.log; tail -2 $O/pytest_row_team.log; wc -l $O/*.jsonl
