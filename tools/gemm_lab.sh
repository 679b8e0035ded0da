#!/bin/bash
# Build variants of the GEMM WITH the lab tilings / probe kernels (-DDV_LAB): argument n = 10000*DV_STAMP + 1000*DV_STAGGER + DV_DBG (see gemm.hip) into build_lab/ (CPU side), then on the
# GPU box: DRVAE_HIP_LIB=build_lab/libdv_dbgN.so python tools/gemm_bench.py ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build_lab
for n in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDV_LAB -DDV_DBG=$((n % 100)) -DDV_STAGGER=$(((n / 1000) % 10)) -DDV_STAMP=$((n / 10000)) -c drvae_amd/csrc/gemm.hip -o build_lab/gemm_dbg$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_lab/libdv_dbg$n.so build_lab/gemm_dbg$n.o drvae_amd/csrc/rows.o drvae_amd/csrc/optim.o ) &
done
wait
ls -la build_lab/*.so
