// Probe (tuning only): what the fp32 matrix pipe sustains on gfx950 under the conditions of a GEMM K loop --
// bare MFMAs, + LDS fragment reads, + LDS-DMA staging -- as wall TFLOP/s, cycles per MFMA and the in-kernel shader
// clock (s_memtime / s_memrealtime at 100 MHz), on random operands.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe tools/mfma_probe.hip && tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long cyc, real; };

// MODE 0: 32x32x2, NACC accumulators, operands in registers
// MODE 1: 16x16x4, NACC*4 accumulators
// MODE 2: 32x32x2 + per 32 MFMAs six ds_read_b128 of the NEXT operands (two register sets) -- the pipe loop's LDS diet
// MODE 3: as 2, + per 64 MFMAs six LDS-DMA pieces (two back to back, three times) from an L2-resident buffer and one raw barrier
// MODE 4: as 3 with the six pieces four MFMAs apart          MODE 5: as 4, DMA addressed as SGPR base + 32-bit lane offset
// MODE 6: as 4 with register staging instead (global_load_dwordx4 four MFMAs apart, six ds_write_b128 in the other half)
// STAG: workgroups with (blockIdx.x >> STAG_SHIFT) & 1 start half an iteration late (s_sleep)
template <int MODE, int WPS, int STAG_SHIFT>
__global__ __launch_bounds__(256, WPS) void probe(const float* __restrict__ src, float* out, Stamp* st, int iters) {
    __shared__ __attribute__((aligned(1024))) float lds[18432];      // 72 KB
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 18432; i += 256) lds[i] = src[(i * 7 + blockIdx.x) & 65535];
    __syncthreads();
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f32x4 acc4[32];
    for (int a = 0; a < 32; ++a)
        for (int r = 0; r < 4; ++r) acc4[a][r] = 0.f;
    float fa[2][8], fb[2][16];
    for (int i = 0; i < 8; ++i) fa[0][i] = fa[1][i] = src[(tid * 8 + i) & 65535];
    for (int i = 0; i < 16; ++i) fb[0][i] = fb[1][i] = src[(tid * 16 + i + 4096) & 65535];
    const float* gp = src + ((size_t)blockIdx.x * 6144 + tid * 4) % 4000000;
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x4 stg[6];
    for (int i = 0; i < 6; ++i) stg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned voff = (unsigned)(((size_t)blockIdx.x * 6144 + tid * 4) % 4000000) * 4;
    if (STAG_SHIFT >= 0 && ((blockIdx.x >> (STAG_SHIFT < 0 ? 0 : STAG_SHIFT)) & 1)) __builtin_amdgcn_s_sleep(33);   // ~2100 cycles
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (MODE >= 2) {
                // six b128 reads into the other register set
                const float* base = lds + (half * 6144 + (lane & 31) * 16 + (lane >> 5) * 8 + ((it & 3) * 512));
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(base + q * 2048);
                    fa[half ^ 1][q * 4 + 0] = v[0]; fa[half ^ 1][q * 4 + 1] = v[1]; fa[half ^ 1][q * 4 + 2] = v[2]; fa[half ^ 1][q * 4 + 3] = v[3];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(base + 4096 + q * 512);
                    fb[half ^ 1][q * 4 + 0] = v[0]; fb[half ^ 1][q * 4 + 1] = v[1]; fb[half ^ 1][q * 4 + 2] = v[2]; fb[half ^ 1][q * 4 + 3] = v[3];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 1) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int a = 0; a < 32; ++a)
                        acc4[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[half][(a >> 3) + s], fb[half][(a & 7) * 2 + (s & 1)], acc4[a], 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int a = 0; a < 8; ++a) {
                        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[half][(a >> 2) * 4 + s], fb[half][(a & 3) * 4 + s], acc[a], 0, 0, 0);
                        if (MODE >= 4 && half == 1 && (a == 3 || a == 7) && s < 3) {
                            const int pi = s * 2 + (a == 7);
                            const unsigned dst = lds0 + 49152 + (((it % 3) * 6 + pi) & 15) * 1024;
                            if (MODE == 4)
                                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp + pi * 1024 + (it & 63) * 16), "s"(dst) : "memory");
                            else if (MODE == 5)
                                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(voff + (pi * 1024 + (it & 63) * 16) * 4), "s"(dst), "s"(src) : "memory");
                            else
                                stg[pi] = *reinterpret_cast<const f32x4*>(gp + pi * 1024 + (it & 63) * 16);
                        }
                        if (MODE == 6 && half == 0 && (a == 3 || a == 7) && s < 3) {
                            const int pi = s * 2 + (a == 7);
                            *reinterpret_cast<f32x4*>(lds + 12288 + (((it % 3) * 6 + pi) & 15) * 256 + tid * 4 % 256) = stg[pi];
                        }
                        if (MODE == 3 && half == 1 && (a == 3) && s < 3) {
                            // two DMA pieces per k-step of the second half (six per iteration)
#pragma unroll
                            for (int p = 0; p < 2; ++p) {
                                const unsigned dst = lds0 + 49152 + (((it % 3) * 6 + s * 2 + p) & 15) * 1024 + wave * 0;
                                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp + (s * 2 + p) * 1024 + (it & 63) * 16), "s"(dst) : "memory");
                            }
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((MODE >= 3 && MODE <= 5) && half == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
            }
            if (MODE == 6 && half == 0) __syncthreads();
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int a = 0; a < 32; ++a)
        for (int r = 0; r < 4; ++r) s += acc4[a][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].real = r1 - r0; }
}

template <int MODE, int WPS, int STAG_SHIFT = -1>
void run(const char* name, const float* src, float* out, Stamp* st, int iters) {
    const int blocks = 256 * WPS;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<Stamp> h(blocks);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, WPS, STAG_SHIFT>), dim3(blocks), dim3(256), 0, 0, src, out, st, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), st, blocks * sizeof(Stamp), hipMemcpyDeviceToHost);
        std::vector<double> cyc, clk;
        for (auto& s : h) { cyc.push_back((double)s.cyc); clk.push_back(s.cyc / (s.real * 10.0)); }
        std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
        const double nm = (MODE == 1 ? 256.0 : 64.0) * iters;            // MFMAs per wave
        const double flop = (double)blocks * 4 * iters * 64.0 * 4096.0;
        if (rep == 2)
            printf("%-44s stag %2d %d wave/SIMD: %7.2f TF/s  %.3f ms  cycles/MFMA(32x32x2-equiv) %.1f  clock %.3f GHz (median; min %.3f)\n", name, STAG_SHIFT, WPS,
                   flop / (ms * 1e-3) / 1e12, ms, cyc[blocks / 2] / (64.0 * iters) / WPS, clk[blocks / 2], clk[0]);
        (void)nm;
    }
}

__global__ void where(unsigned* o) {
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    if ((threadIdx.x & 63) == 0) { o[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; o[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(100);
}

int main() {
    float *src, *out; Stamp* st;
    hipMalloc(&src, 4100000 * 4 + 65536 * 4); hipMalloc(&out, 1 << 22); hipMalloc(&st, 2048 * sizeof(Stamp));
    std::vector<float> h(4100000 + 65536);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    {
        unsigned* w; hipMalloc(&w, 512 * 4 * 2 * 4);
        hipLaunchKernelGGL(where, dim3(512), dim3(256), 0, 0, w);
        std::vector<unsigned> hw(512 * 8);
        hipMemcpy(hw.data(), w, hw.size() * 4, hipMemcpyDeviceToHost);
        printf("placement of a 512-block launch (256 threads): block: xcc/se/cu, simd of waves 0-3\n");
        for (int b : {0, 1, 2, 7, 8, 9, 16, 255, 256, 257, 264, 511}) {
            printf("  block %3d:", b);
            for (int wv = 0; wv < 4; ++wv) {
                unsigned h = hw[(b * 4 + wv) * 2], x = hw[(b * 4 + wv) * 2 + 1];
                printf("  xcc %u se %u cu %u simd %u wave %u |", x & 15, (h >> 13) & 7, (h >> 8) & 15, (h >> 4) & 3, h & 15);
            }
            printf("\n");
        }
        // which blocks share (xcc, se, cu) with block 0?
        auto key = [&](int b) { unsigned h = hw[b * 8], x = hw[b * 8 + 1]; return ((x & 15) << 16) | (h & 0xff00); };
        printf("  blocks on block 0's CU:");
        for (int b = 0; b < 512; ++b) if (key(b) == key(0)) printf(" %d", b);
        printf("\n");
    }
    const int iters = 4000;
    for (int round = 0; round < 2; ++round) {
        run<0, 1>("bare 32x32x2, 8 accumulators", src, out, st, iters);
        run<0, 2>("bare 32x32x2, 8 accumulators", src, out, st, iters);
        run<1, 1>("bare 16x16x4, 32 accumulators", src, out, st, iters);
        run<1, 2>("bare 16x16x4, 32 accumulators", src, out, st, iters);
        run<2, 1>("32x32x2 + 6 ds_read_b128 / 32 MFMA", src, out, st, iters);
        run<2, 2>("32x32x2 + 6 ds_read_b128 / 32 MFMA", src, out, st, iters);
        run<3, 1>("  + 6 LDS-DMA (2x3) + barrier / 64 MFMA", src, out, st, iters);
        run<3, 2>("  + 6 LDS-DMA (2x3) + barrier / 64 MFMA", src, out, st, iters);
        run<4, 1>("  + 6 LDS-DMA spaced + barrier", src, out, st, iters);
        run<4, 2>("  + 6 LDS-DMA spaced + barrier", src, out, st, iters);
        run<4, 2, 0>("  + 6 LDS-DMA spaced + barrier", src, out, st, iters);
        run<4, 2, 3>("  + 6 LDS-DMA spaced + barrier", src, out, st, iters);
        run<4, 2, 8>("  + 6 LDS-DMA spaced + barrier", src, out, st, iters);
        run<5, 1>("  + 6 LDS-DMA spaced saddr + barrier", src, out, st, iters);
        run<5, 2>("  + 6 LDS-DMA spaced saddr + barrier", src, out, st, iters);
        run<5, 2, 8>("  + 6 LDS-DMA spaced saddr + barrier", src, out, st, iters);
        run<6, 1>("  + 6 global_load + 6 ds_write + barrier", src, out, st, iters);
        run<6, 2>("  + 6 global_load + 6 ds_write + barrier", src, out, st, iters);
        run<6, 2, 8>("  + 6 global_load + 6 ds_write + barrier", src, out, st, iters);
    }
    return 0;
}
