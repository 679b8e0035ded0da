// Tuning probe: the optimiser sweep (7 words per parameter) at the wide configuration's size, streaming variants.
//   hipcc --offload-arch=gfx950 -O3 -o tools/adam_probe tools/adam_probe.hip && tools/adam_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
#define float4 v4f
__device__ __forceinline__ void one(float& p, float g, float& m, float& v) {
    g = g + 0.05f * p; m = m + 0.1f * (g - m); v = v * 0.999f + (0.001f * g) * g;
    p = p - 5e-4f * (m / (sqrtf(v) / 0.03f + 1e-8f));
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void adam(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m, float4* __restrict__ v, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride * U) {
        float4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long j = i + u * stride < n4 ? i + u * stride : i;
            if (NT) {
                pp[u] = __builtin_nontemporal_load(p + j); gg[u] = __builtin_nontemporal_load(g + j);
                mm[u] = __builtin_nontemporal_load(m + j); vv[u] = __builtin_nontemporal_load(v + j);
            } else { pp[u] = p[j]; gg[u] = g[j]; mm[u] = m[j]; vv[u] = v[j]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float a = pp[u][e], b = mm[u][e], c = vv[u][e]; one(a, gg[u][e], b, c); pp[u][e] = a; mm[u][e] = b; vv[u][e] = c; }
            const long j = i + u * stride;
            if (j < n4) {
                if (NT) { __builtin_nontemporal_store(pp[u], p + j); __builtin_nontemporal_store(mm[u], m + j); __builtin_nontemporal_store(vv[u], v + j); }
                else { p[j] = pp[u]; m[j] = mm[u]; v[j] = vv[u]; }
            }
        }
    }
}
template <int U, bool NT>
void run(const char* name, float4* p, float4* g, float4* m, float4* v, long n4, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((adam<U, NT>), dim3(blocks), dim3(256), 0, 0, p, g, m, v, n4);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep > 0 && ms < best) best = ms;
    }
    printf("%-28s blocks %5d: %.3f ms  %.2f TB/s\n", name, blocks, best, n4 * 16.0 * 7 / (best * 1e-3) / 1e12);
}
int main() {
    const long n = 125000000, n4 = n / 4;
    float4 *p, *g, *m, *v;
    hipMalloc(&p, n * 4); hipMalloc(&g, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4);
    hipMemset(p, 0, n * 4); hipMemset(g, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
    for (int blocks : {2048, 4096, 8192, 16384}) {
        run<1, false>("unroll 1", p, g, m, v, n4, blocks);
        run<2, false>("unroll 2", p, g, m, v, n4, blocks);
        run<1, true>("unroll 1 nontemporal", p, g, m, v, n4, blocks);
        run<2, true>("unroll 2 nontemporal", p, g, m, v, n4, blocks);
    }
    return 0;
}
