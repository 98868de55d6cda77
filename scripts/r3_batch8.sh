set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_b8
mkdir -p $O
timeout 1700 python -m pytest tests -x -q -m gpu --durations=40 > $O/pytest_gpu.log 2>&1; tail -60 $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
