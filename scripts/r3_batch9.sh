set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3
mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 1700 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu_final.log 2>&1; tail -14 $O/pytest_gpu_final.log
