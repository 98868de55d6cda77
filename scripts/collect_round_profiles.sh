# Collect round 3's measurements on a GPU box (run through gpurun); outputs land in gpurun_out/r3/ and the summaries are
# copied into profiles/r3_* by scripts/publish_round_profiles.sh (which also refreshes profiles/pmc_traffic.json).
# PMC passes are separate runs with --kernel-trace only (no other trace domain), the program directly after `--`.
# (Round 2's version of this script: scripts/collect_round2_profiles.sh.)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3
mkdir -p $O
B="--no-cpu-baseline --no-also"
# the driver's command: headline + also[] + CPU leg
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats -d $O/prof_default -- python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 > $O/prof_default.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_headline -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --sustain 0 $B > $O/prof_headline.log 2>&1
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the four sweep families at the shapes the bench line names
pmc() { key=$1; shift
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_${key}_fetch -- python3 bench.py "$@" --steps 10 --warmup 2 --sustain 0 $B > $O/prof_${key}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_${key}_write -- python3 bench.py "$@" --steps 10 --warmup 2 --sustain 0 $B > $O/prof_${key}_write.log 2>&1
  cp $O/prof_${key}_fetch/*/*_results.db $O/${key}_fetch.db; cp $O/prof_${key}_write/*/*_results.db $O/${key}_write.db
  python scripts/rocpd_summary.py --match gemv_tn $O/${key}_fetch.db $O/${key}_write.db > $O/prof_${key}_pmc.md 2>&1; }
pmc headline
pmc config2 --workload config2
pmc long --m 131072 --n 131072
pmc short --m 2048 --n 1048576
rocprofv3 --kernel-trace --stats -d $O/prof_long -- python3 bench.py --m 131072 --n 131072 --steps 10 --warmup 2 --sustain 0 $B > $O/prof_long.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_config2 -- python3 bench.py --workload config2 --steps 20 --warmup 3 --sustain 0 $B > $O/prof_config2.log 2>&1
# bench lines by column length
python bench.py --workload config2 --steps 50 --warmup 5 $B > $O/bench_config2.json 2>/dev/null
for mn in "131072 131072 long_131072" "65536 262144 long_65536" "50000 84000 odd_50000" "100000 84000 odd_100000" "10000 420000 odd_10000" \
          "10240 209715 mid_10240" "12288 174762 mid_12288" "24576 87381 mid_24576" "32768 65536 mid_32768" "7168 299593 mid_7168" \
          "4096 1048576 short_4096" "2048 1048576 short_2048" "1024 1048576 short_1024" "512 1048576 short_512" "512 4194304 short_512x4M"; do
  set -- $mn; python bench.py --m $1 --n $2 --steps 30 --warmup 5 $B > $O/bench_$3.json 2>/dev/null
done
python bench.py --m 131072 --n 131072 --mode adaptive --steps 20 --warmup 3 $B > $O/bench_long_131072_adaptive.json 2>/dev/null
# the team sweep (members wait for each other through memory) kept running for 45 s: `sustained` and config.sweep_fallbacks (= 0)
python bench.py --m 131072 --n 131072 --steps 20 --warmup 3 --sustain 45 $B > $O/bench_long_131072_soak.json 2>/dev/null
python bench.py --m 8192 --n 1048576 --dtype f64 --steps 20 --warmup 3 $B > $O/bench_f64_8192.json 2>/dev/null
python bench.py --m 65536 --n 131072 --dtype f64 --steps 20 --warmup 3 $B > $O/bench_f64_long_65536.json 2>/dev/null
# per-GPU shapes of the N = 2 / 4 / 8 column-block runs with the collective attached (one rank, the library's own RCCL communicator)
for n in 524288 262144 131072; do python bench.py --m 16384 --n $n --force-comm --sharding cols --steps 100 --warmup 10 $B > $O/bench_colshard_n$n.json 2>/dev/null; done
rocprofv3 --kernel-trace -d $O/trace_cols -- python3 bench.py --m 16384 --n 131072 --force-comm --sharding cols --steps 40 --warmup 5 --kernel-events none $B > $O/trace_cols.log 2>&1
python scripts/step_trace.py "$O/trace_cols/*/*_results.db" > $O/step_cols.md 2>&1
# Douglas-Rachford (config 3)
python tests/tools/bench_dr.py > $O/bench_dr.json 2>/dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR -d $O/prof_dr_sq -- python3 tests/tools/bench_dr.py --no-cpu-baseline --steps 64 > $O/prof_dr_sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $O/prof_dr_tcc -- python3 tests/tools/bench_dr.py --no-cpu-baseline --steps 64 > $O/prof_dr_tcc.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_dr_stats -- python3 tests/tools/bench_dr.py --no-cpu-baseline --steps 64 > $O/prof_dr_stats.log 2>&1
python scripts/bench_panoc.py > $O/bench_panoc.json 2>/dev/null
# the device's ceilings on the same box: streaming read (random data), the sweep's load pattern with its ingredients added one at a time
hipcc -O3 --offload-arch=gfx950 scripts/stream_ceiling.hip -o /tmp/stream_ceiling && /tmp/stream_ceiling 64 > $O/stream_ceiling.log 2>&1
hipcc -O3 --offload-arch=gfx950 scripts/tile_pattern.hip -o /tmp/tile_pattern && /tmp/tile_pattern > $O/tile_pattern.log 2>&1 && /tmp/tile_pattern zeros > $O/tile_pattern_zeros.log 2>&1
for d in prof_default prof_headline prof_long prof_config2 prof_dr_stats; do python scripts/rocpd_summary.py $O/$d/*/*_results.db > $O/$d.md 2>&1; done
for d in prof_dr_sq prof_dr_tcc; do python scripts/rocpd_summary.py --sum-per-dispatch $O/$d/*/*_results.db > $O/$d.md 2>&1; done
ls -la $O | head -80
