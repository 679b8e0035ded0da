// One-kernel tuning build of the hand-pipelined GEMM loop (drvae_amd/csrc/gemm_pipe.inc): compiles in seconds, so the
// emitted ISA of ONE instantiation can be read after every edit:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps -c tools/pipe_lab.hip -DPL_BM=128 -DPL_BN=256 ...
// Never part of the product library.
#include "../drvae_amd/csrc/gemm_common.inc"
namespace {
#include "../drvae_amd/csrc/gemm_pipe.inc"
#ifndef PL_BM
#define PL_BM 128
#define PL_BN 256
#define PL_BK 16
#define PL_S 3
#define PL_WG 2
#endif
#ifndef PL_AKC
#define PL_AKC true
#define PL_BKC true
#endif
template __global__ void gemm_pipe_kernel<PL_BM, PL_BN, PL_BK, 2, 2, PL_S, PL_AKC, PL_BKC, PL_WG>(const dv_gemm_desc, const LoadCfg);
}
extern "C" int pl_launch(const dv_gemm_desc* g, int map, int tiles, void* st) {
    LoadCfg lc{4, 4, 4, 4, map};
    hipLaunchKernelGGL((gemm_pipe_kernel<PL_BM, PL_BN, PL_BK, 2, 2, PL_S, PL_AKC, PL_BKC, PL_WG>), dim3(tiles), dim3(256), 0,
                       static_cast<hipStream_t>(st), *g, lc);
    return 0;
}
