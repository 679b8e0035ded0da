// Probe: effective shader clock, f32-MFMA issue rate, global/L2 load latency, barrier cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void mfma_chain(float* out, unsigned long long* t, int n) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    float s = 0; for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}

__global__ void chase(const int* p, int n, int* out, unsigned long long* t) {
    int idx = 0;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) idx = p[idx];
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[0] = idx; t[0] = c1 - c0; t[1] = r1 - r0;
}

__global__ void barriers(int n, unsigned long long* t, float* out) {
    __shared__ float s[256];
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    float v = threadIdx.x;
    for (int i = 0; i < n; ++i) { s[threadIdx.x] = v; __syncthreads(); v += s[(threadIdx.x + 1) & 255]; __syncthreads(); }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) t[0] = c1 - c0;
}

int main() {
    float* out; unsigned long long* t; hipMalloc(&out, 1 << 24); hipMalloc(&t, 64);
    unsigned long long h[2];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 256, 1024}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_chain, dim3(blocks), dim3(256), 0, 0, out, t, 20000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            printf("mfma_chain blocks=%4d: %.1f cycles/mfma, clock %.3f GHz (memtime/realtime@100MHz), wall %.3f ms -> %.1f TFLOP/s\n",
                   blocks, h[0] / 20000.0, h[0] / (h[1] * 10.0) , ms, blocks * 4 * 20000.0 * 4096 / (ms * 1e-3) / 1e12);
        }
    }
    for (size_t bytes : {size_t(1) << 16, size_t(1) << 21, size_t(1) << 25, size_t(1) << 29}) {
        size_t n = bytes / 4; std::vector<int> hp(n);
        size_t stride = 4099 * 16;   // jump by > a cache line, co-prime-ish walk
        for (size_t i = 0; i < n; ++i) hp[i] = (int)((i + stride) % n);
        int* p; hipMalloc(&p, bytes); hipMemcpy(p, hp.data(), bytes, hipMemcpyHostToDevice);
        int* o; hipMalloc(&o, 4);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, p, 20000, o, t); hipDeviceSynchronize();
        }
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("chase footprint %8zu KB: %.0f cycles/load = %.0f ns (clock %.3f GHz)\n", bytes >> 10, h[0] / 20000.0, h[1] * 10.0 / 20000.0, h[0] / (h[1] * 10.0));
        hipFree(p); hipFree(o);
    }
    hipLaunchKernelGGL(barriers, dim3(1), dim3(256), 0, 0, 10000, t, out); hipDeviceSynchronize();
    hipMemcpy(h, t, 8, hipMemcpyDeviceToHost);
    printf("lds write+barrier+lds read+barrier: %.0f cycles per iteration\n", h[0] / 10000.0);
    return 0;
}
