# Collect round 5's measurements on a GPU box (run through gpurun); outputs land in gpurun_out/r5/ and the summaries are copied
# into profiles/r5_* by scripts/publish_round5_profiles.sh (which also refreshes profiles/pmc_traffic.json).
# PMC passes are separate runs with --kernel-trace only (no other trace domain), the program directly after `--`.
# Usage: collect_round5_profiles.sh [part]   part in {bench, pmc, counters, misc, all}   (the row-team latency sweep: scripts/r5_gpu_d.sh)
set -x
PART=${1:-all}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5
mkdir -p $O
B="--no-cpu-baseline --no-also"
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
if want bench; then
  # the driver's command: headline + also[] + CPU leg
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
  rocprofv3 --kernel-trace --stats -d $O/prof_default -- python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 > $O/prof_default.log 2>&1
  rocprofv3 --kernel-trace --stats -d $O/prof_headline -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --sustain 0 $B > $O/prof_headline.log 2>&1
  # config 4 (PANOC) on its own, with its kernel stats: one read of A per iteration
  python scripts/bench_panoc.py > $O/bench_panoc.json 2>/dev/null
  rocprofv3 --kernel-trace --stats -d $O/prof_panoc -- python3 scripts/bench_panoc.py > $O/prof_panoc.log 2>&1
  for d in prof_default prof_headline prof_panoc; do python scripts/rocpd_summary.py $O/$d/*/*_results.db > $O/$d.md 2>&1; done
fi
if want pmc; then
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
fi
if want counters; then
  # the counter row of the team sweep (131072 x 131072, gemv_tnt<16,1,4,2,2>) beside the headline's
  # gemv_tnm<16,2,4,2> -- same columns as profiles/r3_mid_columns_counters.md
  rocprofv3 -L > $O/counters_available.txt 2>&1
  pick() { out=""; for c in "$@"; do if grep -qw "$c" $O/counters_available.txt; then out="$out $c"; fi; done; echo $out; }
  P1=$(pick SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU)
  P2=$(pick SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA)
  P3=$(pick TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum)
  P4=$(pick SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_LEVEL_WAVES GRBM_GUI_ACTIVE)
  for shape in "team 131072 131072" "headline 16384 1048576"; do
    set -- $shape
    BB="python3 bench.py --m $2 --n $3 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --no-also"
    rocprofv3 --kernel-trace --stats -d $O/cstats_$1 -- $BB > $O/cstats_$1.log 2>&1
    [ -n "$P1" ] && rocprofv3 --kernel-trace --pmc $P1 -d $O/cp1_$1 -- $BB > $O/cp1_$1.log 2>&1
    [ -n "$P2" ] && rocprofv3 --kernel-trace --pmc $P2 -d $O/cp2_$1 -- $BB > $O/cp2_$1.log 2>&1
    [ -n "$P3" ] && rocprofv3 --kernel-trace --pmc $P3 -d $O/cp3_$1 -- $BB > $O/cp3_$1.log 2>&1
    [ -n "$P4" ] && rocprofv3 --kernel-trace --pmc $P4 -d $O/cp4_$1 -- $BB > $O/cp4_$1.log 2>&1
    for p in cstats cp1 cp2 cp3 cp4; do python scripts/rocpd_summary.py --sum-per-dispatch --match gemv_tn $O/${p}_$1/*/*_results.db > $O/${p}_$1.md 2>&1; done
  done
fi
if want misc; then
  # bench lines by column length (the odd team lengths with the even deal of row groups and, for the A/B, without it)
  for mn in "131072 131072 long_131072" "65536 262144 long_65536" "50000 84000 odd_50000" "100000 84000 odd_100000" "150000 56000 odd_150000" \
            "2048 1048576 short_2048" "4096 1048576 short_4096"; do
    set -- $mn; python bench.py --m $1 --n $2 --steps 30 --warmup 5 $B > $O/bench_$3.json 2>/dev/null
  done
  for mn in "50000 84000 odd_50000" "100000 84000 odd_100000" "150000 56000 odd_150000"; do
    set -- $mn; PG_TUNE=1 PG_TNT_EVEN=0 python bench.py --m $1 --n $2 --steps 30 --warmup 5 $B > $O/bench_$3_padded.json 2>/dev/null
  done
  python bench.py --workload config2 --steps 50 --warmup 5 $B > $O/bench_config2.json 2>/dev/null
  # the cooperative team sweep and the library's own RCCL all-reduce alternating on one stream (world size 1)
  python bench.py --m 131072 --n 131072 --force-comm --collective native --sharding cols --steps 30 --warmup 5 $B > $O/bench_long_cols_native.json 2> $O/bench_long_cols_native.err
  rocprofv3 --kernel-trace -d $O/trace_long_cols -- python3 bench.py --m 131072 --n 131072 --force-comm --collective native --sharding cols --steps 20 --warmup 3 --kernel-events none $B > $O/trace_long_cols.log 2>&1
  python scripts/step_trace.py "$O/trace_long_cols/*/*_results.db" > $O/step_long_cols.md 2>&1
  for n in 524288 262144 131072; do python bench.py --m 16384 --n $n --force-comm --sharding cols --steps 100 --warmup 10 $B > $O/bench_colshard_n$n.json 2>/dev/null; done
  # BASELINE config 4's family: PANOC (one read of A per iteration), ZeroFPR with two trial points per sweep / one (round 4), PANOCplus
  python scripts/bench_panoc.py --algo zerofpr --steps 23 --warmup 0 > $O/bench_zerofpr.json 2>/dev/null
  python scripts/bench_panoc.py --algo zerofpr --pair-trials 0 --steps 23 --warmup 0 > $O/bench_zerofpr_single_trials.json 2>/dev/null
  python scripts/bench_panoc.py --algo panocplus --steps 23 --warmup 0 > $O/bench_panocplus.json 2>/dev/null
  rocprofv3 --kernel-trace --stats -d $O/prof_zerofpr -- python3 scripts/bench_panoc.py --algo zerofpr --steps 23 --warmup 0 > $O/prof_zerofpr.log 2>&1
  python scripts/rocpd_summary.py $O/prof_zerofpr/*/*_results.db > $O/prof_zerofpr.md 2>&1
  # north_star's layout as the driver's N > 1 command runs it, two rank processes on this one device (gloo): rows on top, upgraded to the row team
  python bench.py --gpus 2 --share-device --backend gloo --m 4096 --n 1048576 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_2rank_rows_2048.json 2> $O/bench_2rank_rows_2048.err
  python bench.py --gpus 2 --share-device --backend gloo --m 32768 --n 131072 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_2rank_rows_16384.json 2> $O/bench_2rank_rows_16384.err
  hipcc -O3 --offload-arch=gfx950 scripts/stream_ceiling.hip -o /tmp/stream_ceiling && /tmp/stream_ceiling 64 > $O/stream_ceiling.log 2>&1
fi
ls -la $O | head -100
