set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/collect_round_profiles.sh > gpurun_out/r3_collect.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -x -q -m gpu --durations=6 > gpurun_out/r3/pytest_gpu_final.log 2>&1; tail -12 gpurun_out/r3/pytest_gpu_final.log
