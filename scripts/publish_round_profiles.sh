# Copy the summaries scripts/collect_round_profiles.sh left under gpurun_out/r3/ into profiles/r3_* (the tracked, judged copies)
# and refresh profiles/pmc_traffic.json from the PMC passes.  Run in the repo root after the gpurun call returned.
set -e
O=gpurun_out/r3
P=profiles
line() { grep '^{' "$1" | tail -1; }
for f in default config2 long_131072 long_131072_adaptive long_131072_soak long_65536 odd_50000 odd_100000 odd_10000 mid_10240 mid_12288 mid_24576 mid_32768 mid_7168 \
         short_4096 short_2048 short_1024 short_512 short_512x4M colshard_n524288 colshard_n262144 colshard_n131072 f64_8192 f64_long_65536 dr panoc; do
  cp $O/bench_$f.json $P/r3_bench_$f.json
done
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0: the driver's command incl. also[]"
  line $O/prof_default.log; echo
  cat $O/prof_default.md
} > $P/r3_default_kernel_stats.md
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain 0 --no-also (headline only)"
  echo "# the run's own JSON line (HIP-event timing, to compare with the gemv_tnm row below):"
  line $O/prof_headline.log; echo
  cat $O/prof_headline.md
} > $P/r3_headline_kernel_stats.md
{
  echo "# separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (bench.py ... --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --no-also):"
  echo "# headline 16384 x 2^20 (gemv_tnm<16,2,4>), config 2 8192 x 2^18 (gemv_tnm<4,4,8>), 131072 x 131072 (gemv_tnt), 2048 x 2^20 (gemv_tnw)"
  cat $O/prof_headline_pmc.md $O/prof_config2_pmc.md $O/prof_long_pmc.md $O/prof_short_pmc.md
} > $P/r3_sweeps_pmc_fetch_write.md
{
  echo "# kernel stats: 131072 x 131072 (teams of workgroups, cooperative launch) and config 2"
  line $O/prof_long.log; echo; cat $O/prof_long.md
  line $O/prof_config2.log; echo; cat $O/prof_config2.md
} > $P/r3_long_and_config2_kernel_stats.md
{
  echo "# Douglas-Rachford kernels (tests/tools/bench_dr.py --no-cpu-baseline --steps 64): kernel stats, then SQ and TCC counter passes (summed per dispatch)"
  echo "#"
  echo "# Reading (VERDICT r2 item 6a).  dr_step (DRStepF; the rows mix the x/y-only form, 3 vectors in + 2 out = 200 MB, and the full-state form,"
  echo "# 3 in + 5 out) is a copy-like stream: its waves are parked on s_waitcnt for ~0.72 of their time (SQ_WAIT_ANY / SQ_WAVE_CYCLES), issue-stalled"
  echo "# ~0.18, issuing 0.10 (VALU 0.08): nothing on the compute side is short.  On the memory side the L2's write requests to the fabric stall"
  echo "# (TCC_EA0_WRREQ_STALL: 0.5-4.8 M cycles per launch against ~80 k cycles of kernel time per channel group), i.e. the read+write mix is what the"
  echo "# fabric limits: the device's measured read+write (copy) ceiling is 5.5-6.0 TB/s (r3_stream_ceiling.log) and dr_step runs 5.9 TB/s by the"
  echo "# profiler's clock (34 us for 200 MB).  Its 200 MB working set sits inside the 256 MiB Infinity Cache and gains nothing from it: resident data"
  echo "# streams at the HBM rate (r3_mall_panel.md).  Bound: the copy ceiling, 0.73 of the 8 TB/s READ peak the roofline is priced against."
  echo "# dr_block<64> is VALU-issue-bound instead: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = 0.30 per wave at 3-4 waves per SIMD (r2_dr_counters.md has the floor)."
  echo
  cat $O/prof_dr_stats.md $O/prof_dr_sq.md $O/prof_dr_tcc.md
} > $P/r3_dr_counters.md
cp $O/step_cols.md $P/r3_colshard_step_trace_final.md
grep -v amdgpu.ids $O/stream_ceiling.log > $P/r3_stream_ceiling.log
cp $O/tile_pattern.log $P/r3_tile_pattern.log; cp $O/tile_pattern_zeros.log $P/r3_tile_pattern_zeros.log
python scripts/pmc_to_traffic.py headline $O/headline_fetch.db $O/headline_write.db profiles/r3_sweeps_pmc_fetch_write.md > /dev/null
python scripts/pmc_to_traffic.py config2 $O/config2_fetch.db $O/config2_write.db profiles/r3_sweeps_pmc_fetch_write.md > /dev/null
python scripts/pmc_to_traffic.py long_columns $O/long_fetch.db $O/long_write.db profiles/r3_sweeps_pmc_fetch_write.md > /dev/null
python scripts/pmc_to_traffic.py short_columns $O/short_fetch.db $O/short_write.db profiles/r3_sweeps_pmc_fetch_write.md > /dev/null
python - <<'PYEOF'
import json
d = json.load(open("profiles/pmc_traffic.json"))
for k, v in d.items():
    if isinstance(v, dict) and "kernels" in v:
        print(k, v["kernel_source_sha256"][:12], {n: round(r["hbm_bytes"] / 1e9, 3) for n, r in v["kernels"].items()})
PYEOF
