#!/bin/bash
# registers / scratch / LDS of ONE gemv_tnp1_kernel instantiation and its listing (seconds, where the whole translation unit takes minutes):
#   scripts/kernel_lab/one1.sh float 8 2 2 2 2          -> /tmp/p1_float82222.s    (T U C LAG PF LAGR; PAIR=true / AHEAD=true in the environment)
#   python scripts/kernel_lab/loopstat.py /tmp/p1_float82222.s gemv_tnp1_kernel 2000   instruction histogram and s_waitcnt values of the steady loop
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
T=$1; U=$2; C=$3; LAG=$4; PF=$5; LAGR=$6; shift 6
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I "$ROOT/include" -I "$ROOT/proximalalgorithms.jl_amd/csrc" -fno-gpu-rdc --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage -DONE_T=$T -DONE_U=$U -DONE_C=$C -DONE_LAG=$LAG -DONE_PF=$PF -DONE_LAGR=$LAGR -DONE_PAIR=${PAIR:-false} -DONE_AHEAD=${AHEAD:-true} "$@" \
  -S "$ROOT/scripts/kernel_lab/one_tnp1.hip" -o /tmp/p1_$T$U$C$LAG$PF$LAGR.s 2>&1 | grep -E "VGPRs:|AGPRs|Scratch|Occupancy|LDS Size|error" | sed 's/.*remark: //; s/\[-Rpass-analysis=kernel-resource-usage\]//' | tr -s ' ' | tr '\n' ' '; echo
