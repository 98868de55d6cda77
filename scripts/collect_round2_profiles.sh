# Collect the round's measurements on a GPU box (run through gpurun); outputs land in gpurun_out/r2/ and the summaries are
# copied into profiles/ afterwards (scripts/pmc_to_traffic.py refreshes profiles/pmc_traffic.json from the two PMC passes).
# PMC passes are separate runs with --kernel-trace only (no other trace domain), the program directly after `--`.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2
mkdir -p $O
# the driver's command: headline + also[] (adaptive, configs 2 / 3 / 4) + CPU leg
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
# per-kernel durations of the same command (HIP-event timing in the line must agree)
rocprofv3 --kernel-trace --stats -d $O/prof_default -- python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 > $O/prof_default.log 2>&1
# the headline alone (kernel averages of this run are directly comparable with roofline.avg_launch_ms in its line)
rocprofv3 --kernel-trace --stats -d $O/prof_headline -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain 0 --no-also > $O/prof_headline.log 2>&1
python bench.py --workload config2 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_config2.json 2>/dev/null
python bench.py --m 8192 --n $((1<<20)) --dtype f64 --steps 20 --warmup 3 --no-cpu-baseline --no-also > $O/bench_f64_8192.json 2>/dev/null
python bench.py --m 65536 --n 131072 --dtype f64 --steps 20 --warmup 3 --no-cpu-baseline --no-also > $O/bench_f64_long_65536.json 2>/dev/null
# HBM traffic of the headline sweep kernel
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --no-also > $O/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_write -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --no-also > $O/prof_write.log 2>&1
# long columns (teams of workgroups): BASELINE config 5's per-GPU block under column shards, and 65536 rows
python bench.py --m 131072 --n 131072 --steps 20 --warmup 3 --no-cpu-baseline --no-also > $O/bench_long_131072.json 2>/dev/null
python bench.py --m 65536 --n 262144 --steps 20 --warmup 3 --no-cpu-baseline --no-also > $O/bench_long_65536.json 2>/dev/null
# column lengths that fill no power of two: exact-U team members (50000, 100000 rows) and the single-member team (10000 rows)
python bench.py --m 50000 --n 84000 --steps 30 --warmup 5 --no-cpu-baseline --no-also > $O/bench_odd_50000.json 2>/dev/null
python bench.py --m 100000 --n 84000 --steps 30 --warmup 5 --no-cpu-baseline --no-also > $O/bench_odd_100000.json 2>/dev/null
python bench.py --m 10000 --n 420000 --steps 30 --warmup 5 --no-cpu-baseline --no-also > $O/bench_odd_10000.json 2>/dev/null
python bench.py --m 131072 --n 131072 --mode adaptive --steps 20 --warmup 3 --no-cpu-baseline --no-also > $O/bench_long_131072_adaptive.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/prof_long -- python3 bench.py --m 131072 --n 131072 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --no-also > $O/prof_long.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_long_fetch -- python3 bench.py --m 131072 --n 131072 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --no-also > $O/prof_long_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_long_write -- python3 bench.py --m 131072 --n 131072 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --no-also > $O/prof_long_write.log 2>&1
# HBM traffic of config 2's sweep (gemv_tn<4,8,8>) and of the short-column sweep (gemv_tnw, 2048 x 2^20)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_c2_fetch -- python3 bench.py --workload config2 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --no-also > $O/prof_c2_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_c2_write -- python3 bench.py --workload config2 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --no-also > $O/prof_c2_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_short_fetch -- python3 bench.py --m 2048 --n 1048576 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --no-also > $O/prof_short_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_short_write -- python3 bench.py --m 2048 --n 1048576 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --no-also > $O/prof_short_write.log 2>&1
# short columns (one wave per column group): the per-GPU shapes of north_star's row layout at N = 8 and below
for m in 4096 2048 1024 512; do python bench.py --m $m --n $((1<<20)) --steps 30 --warmup 5 --no-cpu-baseline --no-also > $O/bench_short_$m.json 2>/dev/null; done
python bench.py --m 512 --n $((1<<22)) --steps 30 --warmup 5 --no-cpu-baseline --no-also > $O/bench_short_512x4M.json 2>/dev/null
# per-GPU shapes of the N = 2 / 4 / 8 column-block runs with the collective attached (one rank)
for n in 524288 262144 131072; do python bench.py --m 16384 --n $n --force-comm --sharding cols --no-cpu-baseline --no-also > $O/bench_colshard_n$n.json 2>/dev/null; done
# Douglas-Rachford (config 3): stepping / in-library loop, and the VALU counters of the blocked kernel
python tests/tools/bench_dr.py > $O/bench_dr.json 2>/dev/null
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/prof_dr_valu -- python3 tests/tools/bench_dr.py --no-cpu-baseline --steps 64 > $O/prof_dr_valu.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_dr_fetch -- python3 tests/tools/bench_dr.py --no-cpu-baseline --steps 64 > $O/prof_dr_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_dr_write -- python3 tests/tools/bench_dr.py --no-cpu-baseline --steps 64 > $O/prof_dr_write.log 2>&1
python scripts/bench_panoc.py > $O/bench_panoc.json 2>/dev/null
# the device's streaming-read ceiling on the same box (grid-stride and the sweeps' wave-contiguous runs), 64 GiB
hipcc -O3 --offload-arch=gfx950 scripts/stream_ceiling.hip -o /tmp/stream_ceiling && /tmp/stream_ceiling 64 > $O/stream_ceiling.log 2>&1
for d in prof_default prof_headline prof_fetch prof_write prof_long prof_long_fetch prof_long_write prof_dr_valu prof_dr_fetch prof_dr_write prof_c2_fetch prof_c2_write prof_short_fetch prof_short_write; do python scripts/rocpd_summary.py $O/$d/*/*_results.db > $O/$d.md 2>&1; done
cp $O/prof_fetch/*/*_results.db $O/fetch.db; cp $O/prof_write/*/*_results.db $O/write.db
cp $O/prof_long_fetch/*/*_results.db $O/long_fetch.db; cp $O/prof_long_write/*/*_results.db $O/long_write.db
cp $O/prof_c2_fetch/*/*_results.db $O/c2_fetch.db; cp $O/prof_c2_write/*/*_results.db $O/c2_write.db
cp $O/prof_short_fetch/*/*_results.db $O/short_fetch.db; cp $O/prof_short_write/*/*_results.db $O/short_write.db
ls -la $O
