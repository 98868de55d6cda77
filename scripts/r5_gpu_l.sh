#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5l; mkdir -p $O
timeout 2400 python scripts/r5_peer_geometry_parity.py > $O/geometry_parity.log 2>&1
{ echo "## python tests/tools/fuzz_row_team.py 200 870000    (after the geometry table was extended to Float64: one / two / four waves per column in both element types)"
  timeout 1500 python tests/tools/fuzz_row_team.py 200 870000 2>&1 | grep -v amdgpu.ids | tail -6; } > $O/fuzz_row_team_f64.log 2>&1
timeout 3000 python -m pytest tests -m gpu -q --durations=30 > $O/gpu_suite.log 2>&1; echo "rc $?" >> $O/gpu_suite.log
python bench.py --gpus 2 --share-device --backend gloo --m 4096 --n 1048576 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_2rank_rows_2048.json 2> /dev/null
python bench.py --gpus 2 --share-device --backend gloo --m 32768 --n 131072 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_2rank_rows_16384.json 2> /dev/null
tail -4 $O/geometry_parity.log; grep -c "^ok" $O/geometry_parity.log; grep FAIL $O/geometry_parity.log | cut -c1-300; cat $O/fuzz_row_team_f64.log; tail -45 $O/gpu_suite.log; cut -c1-300 $O/bench_2rank_rows_16384.json
