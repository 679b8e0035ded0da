// Round 6: does the optimiser sweep's bandwidth depend on how its four arenas sit relative to each other in memory?  (all four
// streams advance in lock-step: with identically aligned starts they hit the same HBM channels at the same time.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/adam_align_probe tools/adam_align_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void one(float& p, float g, float& m, float& v) {
    g = g + 0.05f * p; m = m + 0.1f * (g - m); v = v * 0.999f + (0.001f * g) * g;
    p = p - 5e-4f * (m / (sqrtf(v) / 0.03f + 1e-8f));
}
__global__ __launch_bounds__(256) void adam(v4f* __restrict__ p, const v4f* __restrict__ g, v4f* __restrict__ m, v4f* __restrict__ v, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride * 2) {
        v4f pp[2], gg[2], mm[2], vv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long j = i + u * stride < n4 ? i + u * stride : i;
            pp[u] = __builtin_nontemporal_load(p + j); gg[u] = __builtin_nontemporal_load(g + j);
            mm[u] = __builtin_nontemporal_load(m + j); vv[u] = __builtin_nontemporal_load(v + j);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float a = pp[u][e], b = mm[u][e], c = vv[u][e]; one(a, gg[u][e], b, c); pp[u][e] = a; mm[u][e] = b; vv[u][e] = c; }
            const long j = i + u * stride;
            if (j < n4) { __builtin_nontemporal_store(pp[u], p + j); __builtin_nontemporal_store(mm[u], m + j); __builtin_nontemporal_store(vv[u], v + j); }
        }
    }
}
int main() {
    const long n = 124500000, n4 = n / 4;
    const long slot = ((n * 4 + (2 << 20) - 1) / (2 << 20)) * (2 << 20);      // 2-MiB aligned slots
    char* base; hipMalloc(&base, 4 * slot + (64 << 20)); hipMemset(base, 0, 4 * slot + (64 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (long stag : {0L, 256L, 4096L, 65536L, 1L << 20, 3L << 20, (1L << 20) + 4096 + 256}) {
        v4f* p = (v4f*)(base); v4f* g = (v4f*)(base + slot + stag); v4f* m = (v4f*)(base + 2 * slot + 2 * stag); v4f* v = (v4f*)(base + 3 * slot + 3 * stag);
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(adam, dim3(16384), dim3(256), 0, 0, p, g, m, v, n4);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (rep > 0 && ms < best) best = ms;
        }
        printf("arenas in 2-MiB aligned slots, each shifted by k x %8ld B against the first: %.3f ms  %.2f TB/s\n", stag, best, n4 * 16.0 * 7 / (best * 1e-3) / 1e12);
    }
    return 0;
}
