#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5f; mkdir -p $O
for pt in 1 always every 0; do
  timeout 600 python scripts/bench_panoc.py --algo zerofpr --pair-trials $pt --steps 23 --warmup 0 > $O/zerofpr_$pt.json 2> $O/zerofpr_$pt.err
  timeout 600 python scripts/bench_panoc.py --algo zerofpr --pair-trials $pt --steps 60 --warmup 0 > $O/zerofpr_${pt}_60.json 2> /dev/null
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=10 -k "newton_family or four_ranks" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
for f in $O/zerofpr_*.json; do echo $f; cut -c1-900 $f; done; tail -15 $O/pytest.log
