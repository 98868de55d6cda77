# Round 3, VERDICT item 4: counter table for the sweep kernels at three column lengths (same box, same n = 2^18):
#   16384 rows  gemv_tn<16,2,4>  (the headline geometry: 0.88-0.90 of 8 TB/s)
#    8192 rows  gemv_tn<4,8,8>   (config 2: 0.83)
#   10240 rows  single-member team gemv_tnt<U=10> (0.76)
# Separate PMC passes (SQ has 8 slots, TCC 4), --kernel-trace only, the program directly after `--`.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_mid
mkdir -p $O
rocprofv3 -L > $O/counters_available.txt 2>&1
pick() { out=""; for c in "$@"; do if grep -qw "$c" $O/counters_available.txt; then out="$out $c"; fi; done; echo $out; }
P1=$(pick SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU)
P2=$(pick SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA)
P3=$(pick TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum)
P4=$(pick SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE)
echo "P1=$P1"; echo "P2=$P2"; echo "P3=$P3"; echo "P4=$P4"
for m in 16384 8192 10240; do
  B="python3 bench.py --m $m --n 262144 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 --no-also"
  rocprofv3 --kernel-trace --stats -d $O/stats_$m -- $B > $O/stats_$m.log 2>&1
  [ -n "$P1" ] && rocprofv3 --kernel-trace --pmc $P1 -d $O/p1_$m -- $B > $O/p1_$m.log 2>&1
  [ -n "$P2" ] && rocprofv3 --kernel-trace --pmc $P2 -d $O/p2_$m -- $B > $O/p2_$m.log 2>&1
  [ -n "$P3" ] && rocprofv3 --kernel-trace --pmc $P3 -d $O/p3_$m -- $B > $O/p3_$m.log 2>&1
  [ -n "$P4" ] && rocprofv3 --kernel-trace --pmc $P4 -d $O/p4_$m -- $B > $O/p4_$m.log 2>&1
  for p in stats p1 p2 p3 p4; do python scripts/rocpd_summary.py --sum-per-dispatch --match gemv_tn $O/${p}_$m/*/*_results.db > $O/${p}_$m.md 2>&1; done
done
# the Infinity-Cache panel experiment (item 5)
hipcc -O3 --offload-arch=gfx950 scripts/mall_panel.hip -o /tmp/mall_panel && /tmp/mall_panel > $O/mall_panel.log 2>&1
ls -la $O
