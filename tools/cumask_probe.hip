// Which compute units does bit i of a hipExtStreamCreateWithCUMask mask name?  For every group of 8 mask bits: launch a
// kernel on a stream masked to those bits and histogram (XCC_ID, SE, CU) of the workgroups (tuning tool).
//   hipcc --offload-arch=gfx950 -O3 -o tools/cumask_probe tools/cumask_probe.hip && tools/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
__global__ void where(unsigned* o) {
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    if (threadIdx.x == 0) { o[blockIdx.x * 2] = hw; o[blockIdx.x * 2 + 1] = xcc; }
    for (int i = 0; i < 50; ++i) __builtin_amdgcn_s_sleep(100);
}
int main() {
    unsigned* w; hipMalloc(&w, 4096 * 8);
    std::vector<unsigned> h(4096 * 2);
    for (int g = 0; g < 32; ++g) {
        unsigned mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = g * 8; b < g * 8 + 8; ++b) mask[b / 32] |= 1u << (b % 32);
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("mask create failed\n"); return 1; }
        hipLaunchKernelGGL(where, dim3(64), dim3(256), 0, st, w);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), w, 64 * 8, hipMemcpyDeviceToHost);
        std::set<unsigned> cus;
        for (int b = 0; b < 64; ++b) cus.insert(((h[b * 2 + 1] & 15) << 16) | (((h[b * 2] >> 13) & 7) << 8) | ((h[b * 2] >> 8) & 15));
        printf("mask bits %3d..%3d ->", g * 8, g * 8 + 7);
        for (unsigned c : cus) printf(" x%u.se%u.cu%u", c >> 16, (c >> 8) & 255, c & 255);
        printf("\n");
        hipStreamDestroy(st);
    }
    return 0;
}
