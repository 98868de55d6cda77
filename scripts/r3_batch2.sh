# Round 3 batch 2: new GPU tests, the load-pattern decomposition (item 4), the column-shard step's launch sequence after the
# fold (item 3), cooperative vs plain launch of the team sweep (item 2)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_b2
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rank_failure or team_sweep_timeout or two_ranks_one_gpu or self_launched or four_ranks or bench_line_contract" > $O/pytest_new.log 2>&1
tail -5 $O/pytest_new.log
hipcc -O3 --offload-arch=gfx950 scripts/tile_pattern.hip -o /tmp/tile_pattern && /tmp/tile_pattern > $O/tile_pattern.log 2>&1
rocprofv3 --kernel-trace -d $O/trace_cols -- python3 bench.py --m 16384 --n 131072 --force-comm --sharding cols --collective torch --steps 40 --warmup 5 --no-cpu-baseline --no-also --kernel-events none > $O/trace_cols.log 2>&1
python scripts/step_trace.py "$O/trace_cols/*/*_results.db" > $O/step_cols_torch.md 2>&1
rocprofv3 --kernel-trace -d $O/trace_cols_native -- python3 bench.py --m 16384 --n 131072 --force-comm --sharding cols --collective native --steps 40 --warmup 5 --no-cpu-baseline --no-also --kernel-events none > $O/trace_cols_native.log 2>&1
python scripts/step_trace.py "$O/trace_cols_native/*/*_results.db" > $O/step_cols_native.md 2>&1
rocprofv3 --kernel-trace -d $O/trace_long -- python3 bench.py --m 131072 --n 131072 --force-comm --sharding cols --collective native --steps 20 --warmup 3 --no-cpu-baseline --no-also --kernel-events none > $O/trace_long.log 2>&1
python scripts/step_trace.py "$O/trace_long/*/*_results.db" > $O/step_long_native.md 2>&1
for c in torch native; do python bench.py --m 16384 --n 131072 --force-comm --sharding cols --collective $c --steps 200 --warmup 10 --no-cpu-baseline --no-also > $O/bench_cols_$c.json 2> $O/bench_cols_$c.err; done
python bench.py --m 16384 --n 131072 --steps 200 --warmup 10 --no-cpu-baseline --no-also > $O/bench_cols_nocomm.json 2> $O/bench_cols_nocomm.err
# cooperative vs plain launch of the team sweep at 131072 x 131072, interleaved
for r in 1 2 3; do
  python bench.py --m 131072 --n 131072 --steps 20 --warmup 3 --no-cpu-baseline --no-also > $O/bench_long_coop_$r.json 2>/dev/null
  PG_TN_TEAM_PLAIN=1 python bench.py --m 131072 --n 131072 --steps 20 --warmup 3 --no-cpu-baseline --no-also > $O/bench_long_plain_$r.json 2>/dev/null
done
ls -la $O
