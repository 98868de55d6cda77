#!/bin/bash
# profiles/r6_solo_experiments.log: timing experiments on the one-wave row-team sweep, ONE rank as a team of one (the kernel alone on the device), pieces of the step switched off through PG_TNT_DBG (wrong results by design); needs scripts/kernel_lab/build_experiment_lib.sh -DPG_TNT_EXPERIMENT
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6x; mkdir -p $O
RT1="python3 tests/tools/row_team.py --bench --solo --ranks 1 --m 2048 --n 1048576 --steps 20 --max-wgs -1"
run() { tag=$1; shift
  env PG_LIB_PATH=$PWD/build/libproxgrad_hip_exp.so PG_TUNE=1 "$@" $RT1 > $O/$tag.json 2> $O/$tag.err
  python3 - $tag $O/$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = d["ranks_out"][0]["kernels"]
    print("%-28s %.1f it/s  sweep avg ms %s  A TB/s (kernel) %.3f" % (sys.argv[1], d["it_per_s"], k.get("gemv_tn"), 2048 * 1048576 * 4 / (k["gemv_tn"][1] * 1e-3) / 1e12))
except Exception as e:
    print(sys.argv[1], "failed", e, open(sys.argv[2]).read()[-300:])
PY
}
for d in ${DBGS:-0 257 16 25}; do run dbg_$d PG_TNT_DBG=$d; done
run pair PG_TNP_PAIR=1
run pair_dbg16 PG_TNP_PAIR=1 PG_TNT_DBG=16
run own_step PG_TNP_AHEAD=0
run r5 PG_TNP_K1=0
