#!/bin/bash
# round 5, run AB: the counter row of the three-point sweep beside the pair sweep and the single sweep (same columns as profiles/r4_team_counters.md)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
rocprofv3 -L > $O/counters_available.txt 2>&1
pick() { out=""; for c in "$@"; do if grep -qw "$c" $O/counters_available.txt; then out="$out $c"; fi; done; echo $out; }
P1=$(pick SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU)
P2=$(pick SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA)
P4=$(pick SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_LEVEL_WAVES GRBM_GUI_ACTIVE)
BB="python3 scripts/r5_pair_sweep_rate.py --reps 4"
[ -n "$P1" ] && rocprofv3 --kernel-trace --pmc $P1 -d $O/cp1 -- $BB > $O/cp1.log 2>&1
[ -n "$P2" ] && rocprofv3 --kernel-trace --pmc $P2 -d $O/cp2 -- $BB > $O/cp2.log 2>&1
[ -n "$P4" ] && rocprofv3 --kernel-trace --pmc $P4 -d $O/cp4 -- $BB > $O/cp4.log 2>&1
for p in cp1 cp2 cp4; do python scripts/rocpd_summary.py --sum-per-dispatch --match gemv_tnm $O/$p/*/*_results.db > $O/$p.md 2>&1; rm -rf $O/$p; done
cat $O/cp1.md $O/cp2.md $O/cp4.md | grep -v "^$" | cut -c1-250
