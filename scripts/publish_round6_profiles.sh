#!/bin/bash
# Copy the summaries scripts/collect_round6_profiles.sh left under gpurun_out/r6/ into profiles/r6_* (the tracked, judged copies),
# refresh profiles/pmc_traffic.json from the PMC passes (taken on the box: gpurun_out/r6/pmc_traffic.json) and regenerate the tables.
# Run in the repo root after the gpurun calls returned.
set -e
O=gpurun_out/r6
P=profiles
line() { grep '^{' "$1" | tail -1; }
for f in default config2 long_131072 short_2048 short_4096 2rank_rows_2048 2rank_rows_16384; do
  [ -s $O/bench_$f.json ] && cp $O/bench_$f.json $P/r6_bench_$f.json
done
[ -s $O/prof_default.md ] && {
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0: the driver's command incl. also[]"
  line $O/prof_default.log; echo; cat $O/prof_default.md; } > $P/r6_default_kernel_stats.md
[ -s $O/prof_headline.md ] && {
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain 0 --no-also (headline only)"
  echo "# the run's own JSON line (HIP-event timing, to compare with the gemv_tnm row below):"
  line $O/prof_headline.log; echo; cat $O/prof_headline.md; } > $P/r6_headline_kernel_stats.md
[ -s $O/prof_row_team.md ] && {
  echo "# rocprofv3 --kernel-trace --stats -- python3 tests/tools/row_team.py --bench --m 4096 --n 1048576 --steps 20 --max-wgs -2: the row-team sweep, two ranks as contexts of ONE process"
  echo "# (46 launches of gemv_tnp1_kernel = 23 per rank: 3 + 20 iterations; each launch sweeps ONE rank's 8.59 GB block while its peer's runs beside it)"
  line $O/prof_row_team.log; echo; cat $O/prof_row_team.md; } > $P/r6_row_team_kernel_stats.md
[ -s $O/prof_headline_pmc.md ] && {
  echo "# separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (bench.py ... --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --no-also), round 6's sources:"
  echo "# headline 16384 x 2^20 (gemv_tnm<16,2,4>), config 2 8192 x 2^18 (gemv_tnm<4,4,8>), 131072 x 131072 (gemv_tnt), 2048 x 2^20 (gemv_tnw)"
  cat $O/prof_headline_pmc.md $O/prof_config2_pmc.md $O/prof_long_pmc.md $O/prof_short_pmc.md; } > $P/r6_sweeps_pmc_fetch_write.md
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json $P/pmc_traffic.json
[ -s $O/startup_zerofpr_gc3.json ] && {
  echo "# scripts/bench_panoc.py --algo {zerofpr,panoc} --steps 23 --warmup 0 --gamma-candidates {1,3}: config 4's instance (16384 x 10^6 f32, logistic + L1),"
  echo "# the first 24 iterations; gc1 = one product A z per candidate of the step-size search (round 5), gc3 = three candidates per read (pg_mat_mul_multi)."
  echo "# One box, back to back.  Same final gamma / tau / residual to the last bit: the decisions are the same."
  for a in zerofpr panoc; do for g in 1 3; do echo "## $a gamma_candidates=$g"; line $O/startup_${a}_gc$g.json; done; done; } > $P/r6_startup_step_size_search.md
[ -s $O/geometry_parity.log ] && cp $O/geometry_parity.log $P/r6_row_team_geometry_parity.log
[ -s $O/gpu_suite.log ] && cp $O/gpu_suite.log $P/r6_gpu_suite_final.log
[ -s gpurun_out/gpu_rates.json ] && cp gpurun_out/gpu_rates.json $P/r6_gpu_rates.json
[ -s gpurun_out/bench_8rank_dry.json ] && cp gpurun_out/bench_8rank_dry.json $P/r6_bench_8rank_dry.json
HIST=$(sed -n '/^## history/,$p' $P/r6_fuzz_campaigns.log 2>/dev/null)  # (the hand-written history below the summary lines is kept)
{
  echo "# Randomised campaigns of round 6 (summary lines; every campaign also lists its failing cases, none below unless said)"
  [ -s $O/fuzz_options.log ] && { echo "## tests/tools/fuzz_parity.py 2000 20000 options (iterator options drawn: mf, sequences, reduce_gamma, minimum_gamma)"; grep -c "^FAIL" $O/fuzz_options.log | sed 's/^/failing cases: /'; grep "^FAIL" $O/fuzz_options.log | cut -c1-400; tail -1 $O/fuzz_options.log; }
  [ -s $O/fuzz_row_team.log ] && { echo "## tests/tools/fuzz_row_team.py 150 9000 (row teams, ranks as threads of one process)"; tail -1 $O/fuzz_row_team.log; }
  [ -s $O/fuzz_gamma_search.log ] && { echo "## tests/tools/fuzz_gamma_search.py 150 31000 (step-size search of PANOC / ZeroFPR: three candidates per read of A against one product per candidate)"; grep "^FAIL" $O/fuzz_gamma_search.log | cut -c1-400; tail -1 $O/fuzz_gamma_search.log; }
  [ -s $O/fuzz_final.log ] && { echo "## every randomised tool once more on the round's LAST build (one gpurun call; seeds no earlier campaign used)"; sed 's/^## /### /' $O/fuzz_final.log; }
  [ -n "$HIST" ] && { echo; echo "$HIST"; }
} > $P/r6_fuzz_campaigns.log.new && mv $P/r6_fuzz_campaigns.log.new $P/r6_fuzz_campaigns.log
python scripts/r6_counter_table.py > /dev/null && echo "counters: profiles/r6_peer_sweep_counters.md is written by hand around scripts/r6_counter_table.py's table"
python scripts/r6_latency_table.py > /dev/null
python scripts/design_table.py --apply
python - <<'PYEOF'
import json
d = json.load(open("profiles/pmc_traffic.json"))
for k, v in d.items():
    if isinstance(v, dict) and "kernels" in v:
        print(k, v["kernel_source_sha256"][:12], {n: round(r["hbm_bytes"] / 1e9, 3) for n, r in v["kernels"].items()})
PYEOF
