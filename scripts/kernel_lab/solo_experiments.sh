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
E="env PG_LIB_PATH=$PWD/build/libproxgrad_hip_exp.so PG_TUNE=1"
if [ -n "$DENSE" ]; then
  # two sweeping waves per SIMD: C = 1 (8 KiB tiles), half the register file per wave, eight workgroups per compute unit
  for g in "1 2 1 8" "1 2 2 8" "1 3 1 6" "1 2 1 6"; do set -- $g
    $E PG_TNP_C=$1 PG_TNP_LAG=$2 PG_TNP_LAGR=$3 PG_TNP_WGS=$4 timeout 300 python3 tests/tools/row_team.py --m 4096 --n 3001 --ranks 2 --steps 12 2>&1 | tail -2 | cut -c1-300
    run dense_C$1_LAG$2_LAGR$3_WGS$4 PG_TNP_C=$1 PG_TNP_LAG=$2 PG_TNP_LAGR=$3 PG_TNP_WGS=$4
    run dense_C$1_LAG$2_LAGR$3_WGS$4_nostores PG_TNP_C=$1 PG_TNP_LAG=$2 PG_TNP_LAGR=$3 PG_TNP_WGS=$4 PG_TNT_DBG=25
  done
  # two ranks sharing the device (threads of one process), default against the densest
  RT2="python3 tests/tools/row_team.py --bench --ranks 2 --m 4096 --n 1048576 --steps 20 --max-wgs -2"
  for g in "2 2 2 4" "1 2 1 8" "2 2 2 4" "1 2 1 8"; do set -- $g
    $E PG_TNP_C=$1 PG_TNP_LAG=$2 PG_TNP_LAGR=$3 PG_TNP_WGS=$4 $RT2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('two ranks C=$1 LAG=$2 LAGR=$3 WGS=$4', {k: d.get(k) for k in ('it_per_s','TBps_all_ranks','late_waves','geometry','a_passes_per_step')})"
  done
fi
if [ -n "$LANDING" ]; then
  # what lands in a device's inbox in a team of N: N pieces of 16 bytes per step (dbg 8193 = never wait + N separate stores into N lines; 24577: as one store)
  run landing_base PG_TNT_DBG=1
  run landing_none PG_TNT_DBG=257
  for nf in 1 2 4 8 16; do run landing_$nf PG_TNT_DBG=8193 PG_TNT_FAKE=$nf; done
  for nf in 2 8 16; do run landing_${nf}_one_store PG_TNT_DBG=24577 PG_TNT_FAKE=$nf; done
  run landing_base PG_TNT_DBG=1
fi
if [ -n "$SPOST" ]; then
  # the post as scalar stores (bit 65536): parity first, then the rates alone and with two ranks on the device
  for sh in "4096 3001" "4096 70001"; do set -- $sh
    $E PG_TNT_DBG=65536 timeout 300 python3 tests/tools/row_team.py --m $1 --n $2 --ranks 2 --steps 12 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
if 'error' in d: print('parity $1 x $2 scalar post: ERROR', str(d)[:400])
else:
    st=d['steps'][0]; print('parity $1 x $2 scalar post: ranks bitwise', d['ranks_agree_bitwise'], 'max dz/scale %.3g' % max(x['dz']/x['z_scale'] for x in st), 'flags', sorted(set(x['flags'] for x in st)), 'a_passes', [x['a_passes'] for x in st][:6], d['geometry'][0])" 2>&1 | cut -c1-400
  done
  for d in 0 65536 0 65536 257; do run spost_$d PG_TNT_DBG=$d; done
  RT2="python3 tests/tools/row_team.py --bench --ranks 2 --m 4096 --n 1048576 --steps 20 --max-wgs -2"
  for d in 0 65536 0 65536; do
    $E PG_TNT_DBG=$d $RT2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('two ranks dbg $d', d.get('it_per_s'), [r.get('late_waves') for r in d.get('ranks_out', [])], [r.get('fallbacks') for r in d.get('ranks_out', [])])"
  done
fi
if [ -n "$BOXID" ]; then
  # which box is this?  clocks / power next to the sweep's rate with and without its post (boxes of the pool differ by up to 9 % on this kernel)
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showperflevel --showmemvendor 2>&1 | grep -v "^=\|^$" | cut -c1-160 | head -30
  for d in 0 257 0 257; do run box_$d PG_TNT_DBG=$d; done
  /opt/rocm/bin/rocm-smi --showclocks 2>&1 | grep -i "mclk\|sclk\|fclk" | cut -c1-160
fi
if [ -n "$LINES" ]; then
  # chunks of 16 / 32 / 64 columns: the same bytes stored with twice / once / half the store instructions
  for l in 32 64 16 32 64 16; do run line_cols_$l PG_TNT_LINE_COLS=$l; done
  run line_cols_64_pair PG_TNT_LINE_COLS=64 PG_TNP_PAIR=1
  run line_cols_64_nopost PG_TNT_LINE_COLS=64 PG_TNT_DBG=257
  $E PG_TNT_LINE_COLS=64 timeout 300 python3 tests/tools/row_team.py --m 4096 --n 3001 --ranks 2 --steps 12 2>&1 | tail -1 | cut -c1-200
fi
[ -n "$ONLY_DBGS" ] && exit 0
run pair PG_TNP_PAIR=1
run pair_dbg16 PG_TNP_PAIR=1 PG_TNT_DBG=16
run own_step PG_TNP_AHEAD=0
run r5 PG_TNP_K1=0
