#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5i; mkdir -p $O
bash scripts/collect_round5_profiles.sh pmc > $O/collect_pmc.log 2>&1
bash scripts/collect_round5_profiles.sh misc > $O/collect_misc.log 2>&1
ls gpurun_out/r5 | head -80; tail -3 $O/collect_pmc.log; for f in gpurun_out/r5/bench_zerofpr.json gpurun_out/r5/bench_2rank_rows_2048.json; do cut -c1-600 $f; echo; done
