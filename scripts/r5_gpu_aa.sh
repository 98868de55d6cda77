#!/bin/bash
# round 5, run AA: kernel stats of the driver's command on the final code; HBM traffic (PMC, separate passes) of the two- and three-point sweeps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof_default -- python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --sustain 0 > $O/prof_default.log 2>&1
python scripts/rocpd_summary.py $O/prof_default/*/*_results.db > $O/prof_default.md 2>&1
rm -rf $O/prof_default
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_trio_fetch -- python3 scripts/r5_pair_sweep_rate.py --reps 4 > $O/prof_trio_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_trio_write -- python3 scripts/r5_pair_sweep_rate.py --reps 4 > $O/prof_trio_write.log 2>&1
cp $O/prof_trio_fetch/*/*_results.db $O/trio_fetch.db; cp $O/prof_trio_write/*/*_results.db $O/trio_write.db
python scripts/rocpd_summary.py --match gemv_tnm $O/trio_fetch.db $O/trio_write.db > $O/prof_trio_pmc.md 2>&1
rm -rf $O/prof_trio_fetch $O/prof_trio_write $O/trio_fetch.db $O/trio_write.db
head -12 $O/prof_default.md | cut -c1-220
cat $O/prof_trio_pmc.md | cut -c1-260
