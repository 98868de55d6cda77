#!/bin/bash
# Round 6's measurements on a GPU box (run through gpurun): outputs under gpurun_out/r6/, summaries are copied into profiles/r6_* by hand
# or by scripts/publish_round6_profiles.sh.  PMC passes are separate runs with --kernel-trace only, the program directly after `--`.
# Usage: collect_round6_profiles.sh part [part ...]   parts: parity ranks ab counters calib latency bench pmc newtests startup fuzzopt fuzzgamma fuzzteam finalfuzz suite
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O
want() { for p in $PARTS; do [ "$p" = "$1" ] && return 0; done; return 1; }
PARTS="$*"
RT="python3 tests/tools/row_team.py --bench --m 4096 --n 1048576 --steps 20 --max-wgs -2"
if want parity; then
  timeout 2400 python scripts/peer_geometry_parity.py > $O/geometry_parity.log 2>&1; tail -3 $O/geometry_parity.log
fi
if want ranks; then
  timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=15 -k "row_team or two_ranks or fuzz_row_teams or fuzz_ranks_as_processes or self_launched or four_ranks or rank_failure" > $O/pytest_ranks.log 2>&1; echo "rc $?" >> $O/pytest_ranks.log; tail -25 $O/pytest_ranks.log
fi
if want ab; then
  # 2 x 2048 rows: round 5's kernel (K1 = 0) | round 6's one-wave sweep with the poll at the start of its own step (AHEAD = 0) | the default
  # (poll one step ahead) | one post per two steps on top; injector off / 0 / 4 / 8 / 12 us.  geometry = C:LAG:LAGR:PF:WGS:W:K1:PAIR:AHEAD
  timeout 1200 python tests/tools/row_team_sweep.py --m 4096 --n 1048576 --two-sweeps --repeat 2 --delays off,0,4000,8000,12000 --geoms 2:2:2:2:4:1:0,2:2:2:2:4:1:1:0:0,default,2:2:2:2:4:1:1:1:1 > $O/ab_2048.jsonl 2> $O/ab_2048.err
  timeout 600 python tests/tools/row_team_sweep.py --m 2048 --n 1048576 --ranks 2 --repeat 1 --delays off,0,8000 --geoms default,4:2:2:2:4:1:1:1:1 > $O/ab_1024.jsonl 2> $O/ab_1024.err
  timeout 600 python tests/tools/row_team_sweep.py --m 8192 --n 524288 --repeat 1 --delays off,0,8000,12000 --geoms 2:2:2:2:2:2:1:0:0,default > $O/ab_4096.jsonl 2> $O/ab_4096.err
  timeout 600 python tests/tools/row_team_sweep.py --m 32768 --n 131072 --repeat 1 --delays off,0,8000,12000 --geoms 1:2:2:2:1:4:1:0:0,default > $O/ab_16384.jsonl 2> $O/ab_16384.err
  timeout 600 python tests/tools/row_team_sweep.py --m 2048 --n 1048576 --dtype f64 --repeat 1 --delays off,0,8000 --geoms 2:2:2:2:4:1:1:0:0,default > $O/ab_f64_1024.jsonl 2> $O/ab_f64_1024.err
  timeout 600 python tests/tools/row_team_sweep.py --m 16384 --n 262144 --ranks 8 --repeat 1 --delays off,0,8000 --geoms 2:2:2:2:4:1:1:0:0,default,2:2:2:2:4:1:1:1:1 > $O/ab_8x2048.jsonl 2> $O/ab_8x2048.err
fi
if want calib; then
  # the same shapes with NO exchange, kernels of two contexts / two processes side by side (what "two ranks share one device" can reach at all)
  timeout 600 python scripts/side_by_side.py --m 2048 --n 1048576 --ranks 1 --mode threads > $O/side_by_side.jsonl 2> $O/side_by_side.err
  timeout 600 python scripts/side_by_side.py --m 2048 --n 1048576 --ranks 2 --mode threads >> $O/side_by_side.jsonl 2>> $O/side_by_side.err
  timeout 600 python scripts/side_by_side.py --m 2048 --n 1048576 --ranks 2 --mode processes >> $O/side_by_side.jsonl 2>> $O/side_by_side.err
  cat $O/side_by_side.jsonl
fi
if want counters; then
  # counter rows of the row-team sweep on 2 x 2048 rows (round 5's kernel, round 6's) next to gemv_tnw<8,4,4> on 2048 x 2^20
  rocprofv3 -L > $O/counters_available.txt 2>&1
  pick() { out=""; for c in "$@"; do if grep -qw "$c" $O/counters_available.txt; then out="$out $c"; fi; done; echo $out; }
  P1=$(pick SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU)
  P2=$(pick SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA)
  P3=$(pick SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_LEVEL_WAVES GRBM_GUI_ACTIVE)
  run() { key=$1; shift
    rocprofv3 --kernel-trace --stats -d $O/c_${key}_stats -- "$@" > $O/c_${key}_stats.log 2>&1
    [ -n "$P1" ] && rocprofv3 --kernel-trace --pmc $P1 -d $O/c_${key}_p1 -- "$@" > $O/c_${key}_p1.log 2>&1
    [ -n "$P2" ] && rocprofv3 --kernel-trace --pmc $P2 -d $O/c_${key}_p2 -- "$@" > $O/c_${key}_p2.log 2>&1
    [ -n "$P3" ] && rocprofv3 --kernel-trace --pmc $P3 -d $O/c_${key}_p3 -- "$@" > $O/c_${key}_p3.log 2>&1
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/c_${key}_fetch -- "$@" > $O/c_${key}_fetch.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/c_${key}_write -- "$@" > $O/c_${key}_write.log 2>&1
    for p in stats p1 p2 p3 fetch write; do python scripts/rocpd_summary.py --sum-per-dispatch --match gemv_tn $O/c_${key}_$p/*/*_results.db > $O/c_${key}_$p.md 2>&1; rm -rf $O/c_${key}_$p; done; }
  # ONE rank as a team of one (pg_ctx_test_team_fault kind 4): the sweep exchanges its granules with itself and runs ALONE on the device --
  # rocprofv3 --pmc serialises kernels, two ranks' sweeps would only wait for each other
  RT1="python3 tests/tools/row_team.py --bench --solo --ranks 1 --m 2048 --n 1048576 --steps 20 --max-wgs -1"
  run k1 $RT1
  PG_TUNE=1 PG_TNP_AHEAD=0 run k1_own_step $RT1
  PG_TUNE=1 PG_TNP_K1=0 run r5 $RT1
  run tnw python3 bench.py --m 2048 --n 1048576 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --no-also
  $RT1 > $O/solo_k1.json 2>/dev/null; PG_TUNE=1 PG_TNP_AHEAD=0 $RT1 > $O/solo_k1_own_step.json 2>/dev/null; PG_TUNE=1 PG_TNP_K1=0 $RT1 > $O/solo_r5.json 2>/dev/null
  cut -c1-200 $O/solo_k1.json $O/solo_k1_own_step.json $O/solo_r5.json
  for k in k1 k1_own_step r5 tnw; do echo "== $k"; grep -h "gemv_tn" $O/c_${k}_stats.md | head -3 | cut -c1-200; done
fi
if want latency; then
  # the latency curve (geometry = C:LAG:LAGR:PF:WGS:W:K1:PAIR:AHEAD): round 5's kernel | round 6's one-wave sweep (default: poll one step ahead) | with
  # the poll in its own step | one post per two steps; the longer blocks' defaults (several waves per column: wave_allsum without LDS)
  D=off,0,2000,4000,6000,8000,12000,16000
  timeout 900 python tests/tools/row_team_sweep.py --m 4096 --n 1048576 --two-sweeps --repeat 2 --delays $D --geoms 2:2:2:2:4:1:0,default,2:2:2:2:4:1:1:0:0,2:2:2:2:4:1:1:1:1 > $O/sweep_2048.jsonl 2> $O/sweep_2048.err
  timeout 600 python tests/tools/row_team_sweep.py --m 8192 --n 524288 --two-sweeps --repeat 2 --delays $D --geoms default > $O/sweep_4096.jsonl 2> $O/sweep_4096.err
  timeout 600 python tests/tools/row_team_sweep.py --m 32768 --n 131072 --two-sweeps --repeat 2 --delays $D --geoms default > $O/sweep_16384.jsonl 2> $O/sweep_16384.err
  timeout 600 python tests/tools/row_team_sweep.py --m 16384 --n 262144 --ranks 4 --repeat 1 --delays off,0,4000,8000 --geoms default > $O/sweep_4x4096.jsonl 2> $O/sweep_4x4096.err
  timeout 600 python tests/tools/row_team_sweep.py --m 16384 --n 262144 --ranks 8 --repeat 1 --delays off,0,4000,8000 --geoms default,2:2:2:2:4:1:1:1:1 > $O/sweep_8x2048.jsonl 2> $O/sweep_8x2048.err
  timeout 600 python tests/tools/row_team_sweep.py --m 2048 --n 1048576 --dtype f64 --repeat 1 --delays off,0,4000,8000,12000 --geoms 2:2:2:2:4:1:0,default > $O/sweep_f64_1024.jsonl 2> $O/sweep_f64_1024.err
fi
if want bench; then
  B="--no-cpu-baseline --no-also"
  # the driver's command: headline + also[] + CPU leg; then the kernel stats of the same command and of the headline alone
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 1200 $O/bench_default.json
  rocprofv3 --kernel-trace --stats -d $O/prof_default -- python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 > $O/prof_default.log 2>&1
  rocprofv3 --kernel-trace --stats -d $O/prof_headline -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --sustain 0 $B > $O/prof_headline.log 2>&1
  for d in prof_default prof_headline; do python scripts/rocpd_summary.py $O/$d/*/*_results.db > $O/$d.md 2>&1; rm -rf $O/$d; done
  # north_star's layout as the driver's N > 1 command runs it, two rank processes on this one device (gloo): rows on top, upgraded to the row team
  python bench.py --gpus 2 --share-device --backend gloo --m 4096 --n 1048576 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_2rank_rows_2048.json 2> $O/bench_2rank_rows_2048.err
  python bench.py --gpus 2 --share-device --backend gloo --m 32768 --n 131072 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_2rank_rows_16384.json 2> $O/bench_2rank_rows_16384.err
  rocprofv3 --kernel-trace --stats -d $O/prof_row_team -- python3 tests/tools/row_team.py --bench --m 4096 --n 1048576 --steps 20 --max-wgs -2 > $O/prof_row_team.log 2>&1
  python scripts/rocpd_summary.py $O/prof_row_team/*/*_results.db > $O/prof_row_team.md 2>&1; rm -rf $O/prof_row_team
  for mn in "2048 1048576 short_2048" "4096 1048576 short_4096" "131072 131072 long_131072"; do
    set -- $mn; python bench.py --m $1 --n $2 --steps 30 --warmup 5 $B > $O/bench_$3.json 2>/dev/null
  done
  python bench.py --workload config2 --steps 50 --warmup 5 $B > $O/bench_config2.json 2>/dev/null
fi
if want pmc; then
  # HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the four sweep families at the shapes the bench line names: the sources of
  # these kernels changed in round 6 (wave_allsum without LDS), so profiles/pmc_traffic.json is re-taken on them
  B="--no-cpu-baseline --no-also"
  pmc() { key=$1; tkey=$2; shift 2
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_${key}_fetch -- python3 bench.py "$@" --steps 10 --warmup 2 --sustain 0 $B > $O/prof_${key}_fetch.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_${key}_write -- python3 bench.py "$@" --steps 10 --warmup 2 --sustain 0 $B > $O/prof_${key}_write.log 2>&1
    python scripts/rocpd_summary.py --match gemv_tn $O/prof_${key}_fetch/*/*_results.db $O/prof_${key}_write/*/*_results.db > $O/prof_${key}_pmc.md 2>&1
    python scripts/pmc_to_traffic.py $tkey $O/prof_${key}_fetch/*/*_results.db $O/prof_${key}_write/*/*_results.db profiles/r6_sweeps_pmc_fetch_write.md $O/pmc_traffic.json > /dev/null
    rm -rf $O/prof_${key}_fetch $O/prof_${key}_write; }
  pmc headline headline
  pmc config2 config2 --workload config2
  pmc long long_columns --m 131072 --n 131072
  pmc short short_columns --m 2048 --n 1048576
  python3 -c "
import json; d=json.load(open('$O/pmc_traffic.json'))
for k,v in d.items(): print(k, v['kernel_source_sha256'][:12], {n: round(r['hbm_bytes']/1e9,3) for n,r in v['kernels'].items()})"
fi
if want newtests; then
  timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gamma_too_small or fuzz_differential or lbfgs or blas1 or prox_operators" > $O/pytest_new.log 2>&1; echo "rc $?" >> $O/pytest_new.log; tail -15 $O/pytest_new.log
fi
if want startup; then
  # the step-size search with three candidates per read of A against one product per candidate
  timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=8 -k "mul_multi or three_candidates or zerofpr or panoc or newton_family" > $O/pytest_startup.log 2>&1; echo "rc $?" >> $O/pytest_startup.log; tail -12 $O/pytest_startup.log
  for algo in zerofpr panoc; do for gc in 1 3; do
    timeout 600 python scripts/bench_panoc.py --algo $algo --steps 23 --warmup 0 --gamma-candidates $gc > $O/startup_${algo}_gc$gc.json 2> $O/startup_${algo}_gc$gc.err
  done; done
fi
if want fuzzopt; then
  timeout 2400 python tests/tools/fuzz_parity.py ${FUZZ_CASES:-2000} 20000 options > $O/fuzz_options.log 2>&1; tail -12 $O/fuzz_options.log | cut -c1-600
fi
if want fuzzgamma; then
  timeout 1500 python tests/tools/fuzz_gamma_search.py ${GAMMA_CASES:-150} 31000 > $O/fuzz_gamma_search.log 2>&1; tail -6 $O/fuzz_gamma_search.log | cut -c1-400
fi
if want finalfuzz; then
  # every randomised tool once more on the round's last build (seeds no earlier campaign used); one log, the summary lines are published
  F=$O/fuzz_final.log; : > $F
  ff() { echo "## python $*" >> $F; timeout 1500 python "$@" > $O/fuzz_final_one.log 2>&1; grep "^FAIL" $O/fuzz_final_one.log | cut -c1-400 >> $F; tail -1 $O/fuzz_final_one.log | cut -c1-300 >> $F; }
  ff tests/tools/fuzz_parity.py 1500 1000000
  ff tests/tools/fuzz_parity.py 300 1010000 tall
  ff tests/tools/fuzz_newton.py 1500 1020000
  ff tests/tools/fuzz_newton.py 600 1030000 tall
  ff tests/tools/fuzz_newton.py 20 1040000 wide
  ff tests/tools/fuzz_resume.py 800 1050000
  ff tests/tools/fuzz_gamma_search.py 300 1060000
  ff tests/tools/fuzz_row_team.py 60 1070000
  ff tests/tools/fuzz_row_team.py 20 1080000 many
  ff tests/tools/fuzz_row_team.py 40 1090000 pe
  ff tests/tools/fuzz_row_team.py 40 1100000 cols
  ff tests/tools/fuzz_bench_ranks.py 30 1110000
  rm -f $O/fuzz_final_one.log; cat $F
fi
if want fuzzteam; then
  timeout 1500 python tests/tools/fuzz_row_team.py 150 9000 > $O/fuzz_row_team.log 2>&1; tail -5 $O/fuzz_row_team.log | cut -c1-400
fi
if want suite; then
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 > $O/gpu_suite.log 2>&1; echo "rc $?" >> $O/gpu_suite.log; tail -40 $O/gpu_suite.log
fi
# gpurun copies gpurun_out/ back only while it is below 64 MiB: the raw rocprofv3 databases stay on the box
find $O -name "*.db" -size +1M -delete 2>/dev/null
du -sh $O | tail -1
# a digest at the very end (gpurun shows the tail of stdout)
echo "==== digest"
for f in $O/ab_*.jsonl $O/sweep_*.jsonl; do [ -f "$f" ] && { echo "-- $f"; python3 - "$f" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    try: d = json.loads(ln)
    except ValueError: continue
    if "error" in d: print("  ERROR", str(d)[:300]); continue
    print("  %-16s delay %-6s %.3f TB/s  %.1f it/s  passes %.2f late %s slack %s" % (d.get("geometry"), d.get("delay_ns"), d.get("TBps_all_ranks", 0), d.get("it_per_s", 0), d.get("a_passes_per_step", 0), d.get("late_waves"), d.get("slack_us")))
PY
}; done
[ -f $O/side_by_side.jsonl ] && cat $O/side_by_side.jsonl
for f in $O/solo_*.json; do [ -f "$f" ] && { echo "-- $f"; cut -c1-230 $f; }; done
[ -f $O/geometry_parity.log ] && { grep -c "^ok\|^OK" $O/geometry_parity.log; grep "FAIL" $O/geometry_parity.log | cut -c1-300; tail -2 $O/geometry_parity.log | cut -c1-300; }
[ -f $O/pytest_ranks.log ] && tail -4 $O/pytest_ranks.log | cut -c1-300
[ -f $O/pytest_new.log ] && tail -4 $O/pytest_new.log | cut -c1-300
for f in $O/startup_*.json; do [ -f "$f" ] && { echo "-- $f"; python3 -c "
import json,sys
d=json.load(open('$f')); print({k: d[k] for k in ('value','A_passes_per_step','A_passes_with_first_iteration','gamma_candidates_ahead','ms_per_step')}, d['final'])" 2>&1 | cut -c1-400; }; done
[ -f $O/pytest_startup.log ] && tail -4 $O/pytest_startup.log | cut -c1-300
