#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "default_line" > $O/alone.log 2>&1; echo "alone rc $?"; tail -3 $O/alone.log
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lasso_small_known or default_line" > $O/after_small.log 2>&1; echo "after small rc $?"; tail -3 $O/after_small.log
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_ranks_one_gpu or default_line" > $O/after_two.log 2>&1; echo "after two_ranks rc $?"; tail -3 $O/after_two.log
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "row_team_iterates or default_line" > $O/after_rt.log 2>&1; echo "after row_team rc $?"; tail -3 $O/after_rt.log
grep -h "AssertionError: (\[" $O/*.log | cut -c1-300
