#!/bin/bash
# round 5, run AC: after the pair sweep's two-tile instantiations (pg_gemv_tn3.hip changed): the PMC passes again, then the full GPU suite and the default line
bash scripts/collect_round5_profiles.sh pmc > gpurun_out/r5_collect_pmc.log 2>&1
ls -la gpurun_out/r5/*_fetch.db gpurun_out/r5/*_write.db | awk '{print $5, $9}'
bash scripts/r5_gpu_x.sh
