#!/bin/bash
# An experimental build of the whole library with extra flags (e.g. -DPG_TNT_EXPERIMENT: the timing switches of pg_gemv_tnp1.h / pg_gemv_tnt.h,
# PG_TNT_DBG) into build/libproxgrad_hip_exp.so; use it with PG_LIB_PATH=$PWD/build/libproxgrad_hip_exp.so.  Never the product build.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$ROOT"
mkdir -p build/exp_obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I include -I proximalalgorithms.jl_amd/csrc -Wall -Wno-unused-function -fno-gpu-rdc $*"
for f in pg_core pg_gemv pg_gemv_tn2 pg_gemv_tn3 pg_gemv_tn4 pg_gemv_tn4d pg_gemv_tn5 pg_gemv_dys pg_vec pg_iter pg_persist pg_lbfgs pg_comm; do
  /opt/rocm/bin/hipcc $FLAGS -c proximalalgorithms.jl_amd/csrc/$f.hip -o build/exp_obj/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/libproxgrad_hip_exp.so build/exp_obj/*.o -ldl -Wl,-rpath,/opt/rocm/lib
rm -rf build/exp_obj
ls -la build/libproxgrad_hip_exp.so
