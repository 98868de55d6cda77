# Round 3 batch 2: the load-pattern decomposition (item 4) and the column-shard step's launch sequence BEFORE the fold (item 3)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_b2
mkdir -p $O
hipcc -O3 --offload-arch=gfx950 scripts/tile_pattern.hip -o /tmp/tile_pattern && /tmp/tile_pattern > $O/tile_pattern.log 2>&1
for n in 131072; do
  rocprofv3 --kernel-trace -d $O/trace_cols_$n -- python3 bench.py --m 16384 --n $n --force-comm --sharding cols --collective torch --steps 40 --warmup 5 --no-cpu-baseline --no-also --kernel-events none > $O/trace_cols_$n.log 2>&1
  python scripts/step_trace.py "$O/trace_cols_$n/*/*_results.db" > $O/step_cols_$n.md 2>&1
done
rocprofv3 --kernel-trace -d $O/trace_long -- python3 bench.py --m 131072 --n 131072 --force-comm --sharding cols --collective torch --steps 20 --warmup 3 --no-cpu-baseline --no-also --kernel-events none > $O/trace_long.log 2>&1
python scripts/step_trace.py "$O/trace_long/*/*_results.db" > $O/step_long.md 2>&1
python bench.py --m 16384 --n 131072 --force-comm --sharding cols --collective torch --steps 200 --warmup 10 --no-cpu-baseline --no-also > $O/bench_cols_torch.json 2> $O/bench_cols_torch.err
python bench.py --m 16384 --n 131072 --force-comm --sharding cols --collective native --steps 200 --warmup 10 --no-cpu-baseline --no-also > $O/bench_cols_native.json 2> $O/bench_cols_native.err
python bench.py --m 16384 --n 131072 --steps 200 --warmup 10 --no-cpu-baseline --no-also > $O/bench_cols_nocomm.json 2> $O/bench_cols_nocomm.err
ls -la $O
