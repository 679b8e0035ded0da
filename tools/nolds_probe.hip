// Feasibility probe (GPU box only): an LDS-free fp32-MFMA GEMM for the mid-size products of the
// train step.  C[M,N] = A[M,K] . B[N,K]^T, both operands K-contiguous: every wave loads the MFMA
// operand fragments of its own TM x TN tile straight from global/L2 as float4 per lane (the k order
// inside an 8-chunk is permuted identically for A and B, which a dot product does not care about),
// so the K loop has no LDS traffic and no barriers; KS waves of a workgroup split K (interleaved
// 8-chunks, so the 4 waves walk the same cache lines) and reduce once through LDS at the end.
//   hipcc --offload-arch=gfx950 -O3 tools/nolds_probe.hip -o /tmp/nolds && /tmp/nolds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float v16f __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int TM, int TN, int KS, int PF>
__global__ __launch_bounds__(64 * KS) void gemm_nolds(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                     int ldb, float* __restrict__ C, int ldc, int M, int N, int K,
                                                     int tiles_n) {
    constexpr int IM = TM / 32, JN = TN / 32;
    __shared__ float red[(KS > 1 ? (KS / 2) : 1) * TM * TN];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 31, h = l >> 5;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;
    const float* pa[IM];
    const float* pb[JN];
#pragma unroll
    for (int i = 0; i < IM; ++i) {
        int row = m0 + 32 * i + r;
        row = row < M ? row : M - 1;
        pa[i] = A + (int64_t)row * lda + 8 * w + 4 * h;
    }
#pragma unroll
    for (int j = 0; j < JN; ++j) {
        int col = n0 + 32 * j + r;
        col = col < N ? col : N - 1;
        pb[j] = B + (int64_t)col * ldb + 8 * w + 4 * h;
    }
    v16f acc[IM][JN];
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nchunk = K / 8;                       // probe: K % 8 == 0
    const int mine = (nchunk - w + KS - 1) / KS;    // chunks w, w+KS, ...
    float4 fa[PF][IM], fb[PF][JN];
    // prologue: PF chunks in flight
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        const bool ok = s < mine;
#pragma unroll
        for (int i = 0; i < IM; ++i) fa[s][i] = *reinterpret_cast<const float4*>(ok ? pa[i] + (int64_t)s * 8 * KS : pa[i]);
#pragma unroll
        for (int j = 0; j < JN; ++j) fb[s][j] = *reinterpret_cast<const float4*>(ok ? pb[j] + (int64_t)s * 8 * KS : pb[j]);
    }
    int c = 0;
    for (; c + PF <= mine; c += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            float4 a[IM], b[JN];
#pragma unroll
            for (int i = 0; i < IM; ++i) a[i] = fa[s][i];
#pragma unroll
            for (int j = 0; j < JN; ++j) b[j] = fb[s][j];
            const int nxt = c + s + PF;
            const int64_t off = (int64_t)(nxt < mine ? nxt : 0) * 8 * KS;      // stay in bounds, value unused
#pragma unroll
            for (int i = 0; i < IM; ++i) fa[s][i] = *reinterpret_cast<const float4*>(pa[i] + off);
#pragma unroll
            for (int j = 0; j < JN; ++j) fb[s][j] = *reinterpret_cast<const float4*>(pb[j] + off);
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < JN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }
    // remainder (< PF chunks), already loaded
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        if (c + s < mine) {
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < JN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][i].x, fb[s][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][i].y, fb[s][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][i].z, fb[s][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][i].w, fb[s][j].w, acc[i][j], 0, 0, 0);
                }
        }
    }
    // ---- K-split reduction through LDS (tree), lane-major layout: conflict-free
    if (KS > 1) {
        for (int half = KS / 2; half >= 1; half >>= 1) {
            if (w >= half && w < 2 * half) {
                float* dst = red + (w - half) * TM * TN;
#pragma unroll
                for (int i = 0; i < IM; ++i)
#pragma unroll
                    for (int j = 0; j < JN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) dst[((i * JN + j) * 16 + e) * 64 + l] = acc[i][j][e];
            }
            __syncthreads();
            if (w < half) {
                const float* src = red + w * TM * TN;
#pragma unroll
                for (int i = 0; i < IM; ++i)
#pragma unroll
                    for (int j = 0; j < JN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] += src[((i * JN + j) * 16 + e) * 64 + l];
            }
            __syncthreads();
        }
    }
    if (w == 0) {
#pragma unroll
        for (int i = 0; i < IM; ++i)
#pragma unroll
            for (int j = 0; j < JN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h, col = n0 + 32 * j + r;
                    if (row < M && col < N) C[(int64_t)row * ldc + col] = acc[i][j][e];
                }
    }
}

typedef float v4f __attribute__((ext_vector_type(4)));

// same idea on v_mfma_f32_16x16x4_f32: lane (r = l%16, q = l/16) loads A[row r][16c + 4q .. +3] -- 4 lanes per
// 64 B of a row (the 32x32x2 layout has 2 lanes per 32 B: twice the cache-line touches per byte)
template <int TM, int TN, int KS, int PF>
__global__ __launch_bounds__(64 * KS) void gemm_nolds16(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                       int ldb, float* __restrict__ C, int ldc, int M, int N, int K,
                                                       int tiles_n) {
    constexpr int IM = TM / 16, JN = TN / 16;
    __shared__ float red[(KS > 1 ? (KS / 2) : 1) * TM * TN];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 15, q = l >> 4;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;
    const float* pa[IM];
    const float* pb[JN];
#pragma unroll
    for (int i = 0; i < IM; ++i) {
        int row = m0 + 16 * i + r;
        row = row < M ? row : M - 1;
        pa[i] = A + (int64_t)row * lda + 16 * w + 4 * q;
    }
#pragma unroll
    for (int j = 0; j < JN; ++j) {
        int col = n0 + 16 * j + r;
        col = col < N ? col : N - 1;
        pb[j] = B + (int64_t)col * ldb + 16 * w + 4 * q;
    }
    v4f acc[IM][JN];
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    const int nchunk = K / 16;                      // probe: K % 16 == 0
    const int mine = (nchunk - w + KS - 1) / KS;
    float4 fa[PF][IM], fb[PF][JN];
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        const bool ok = s < mine;
#pragma unroll
        for (int i = 0; i < IM; ++i) fa[s][i] = *reinterpret_cast<const float4*>(ok ? pa[i] + (int64_t)s * 16 * KS : pa[i]);
#pragma unroll
        for (int j = 0; j < JN; ++j) fb[s][j] = *reinterpret_cast<const float4*>(ok ? pb[j] + (int64_t)s * 16 * KS : pb[j]);
    }
    int c = 0;
    for (; c + PF <= mine; c += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            float4 a[IM], b[JN];
#pragma unroll
            for (int i = 0; i < IM; ++i) a[i] = fa[s][i];
#pragma unroll
            for (int j = 0; j < JN; ++j) b[j] = fb[s][j];
            const int nxt = c + s + PF;
            const int64_t off = (int64_t)(nxt < mine ? nxt : 0) * 16 * KS;
#pragma unroll
            for (int i = 0; i < IM; ++i) fa[s][i] = *reinterpret_cast<const float4*>(pa[i] + off);
#pragma unroll
            for (int j = 0; j < JN; ++j) fb[s][j] = *reinterpret_cast<const float4*>(pb[j] + off);
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < JN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        if (c + s < mine) {
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < JN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][i].x, fb[s][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][i].y, fb[s][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][i].z, fb[s][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][i].w, fb[s][j].w, acc[i][j], 0, 0, 0);
                }
        }
    }
    if (KS > 1) {
        for (int half = KS / 2; half >= 1; half >>= 1) {
            if (w >= half && w < 2 * half) {
                float* dst = red + (w - half) * TM * TN;
#pragma unroll
                for (int i = 0; i < IM; ++i)
#pragma unroll
                    for (int j = 0; j < JN; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) dst[((i * JN + j) * 4 + e) * 64 + l] = acc[i][j][e];
            }
            __syncthreads();
            if (w < half) {
                const float* src = red + w * TM * TN;
#pragma unroll
                for (int i = 0; i < IM; ++i)
#pragma unroll
                    for (int j = 0; j < JN; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][e] += src[((i * JN + j) * 4 + e) * 64 + l];
            }
            __syncthreads();
        }
    }
    if (w == 0) {
#pragma unroll
        for (int i = 0; i < IM; ++i)
#pragma unroll
            for (int j = 0; j < JN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m0 + 16 * i + 4 * q + e, col = n0 + 16 * j + r;
                    if (row < M && col < N) C[(int64_t)row * ldc + col] = acc[i][j][e];
                }
    }
}

template <int TM, int TN, int KS, int PF, bool M16 = false>
static void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K, const std::vector<float>& ref,
                int reps) {
    const int tiles_m = (M + TM - 1) / TM, tiles_n = (N + TN - 1) / TN;
    dim3 grid(tiles_m * tiles_n), block(64 * KS);
    const int lda = K, ldb = K, ldc = N;
    CK(hipMemset(C, 0, sizeof(float) * M * N));
    if constexpr (M16) hipLaunchKernelGGL((gemm_nolds16<TM, TN, KS, PF>), grid, block, 0, 0, A, lda, B, ldb, C, ldc, M, N, K, tiles_n);
    else hipLaunchKernelGGL((gemm_nolds<TM, TN, KS, PF>), grid, block, 0, 0, A, lda, B, ldb, C, ldc, M, N, K, tiles_n);
    CK(hipDeviceSynchronize());
    std::vector<float> out((size_t)M * N);
    CK(hipMemcpy(out.data(), C, sizeof(float) * M * N, hipMemcpyDeviceToHost));
    double err = 0, mx = 0;
    for (size_t i = 0; i < out.size(); ++i) {
        err = fmax(err, fabs((double)out[i] - ref[i]));
        mx = fmax(mx, fabs((double)ref[i]));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i)
        if constexpr (M16) hipLaunchKernelGGL((gemm_nolds16<TM, TN, KS, PF>), grid, block, 0, 0, A, lda, B, ldb, C, ldc, M, N, K, tiles_n);
    else hipLaunchKernelGGL((gemm_nolds<TM, TN, KS, PF>), grid, block, 0, 0, A, lda, B, ldb, C, ldc, M, N, K, tiles_n);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
        if constexpr (M16) hipLaunchKernelGGL((gemm_nolds16<TM, TN, KS, PF>), grid, block, 0, 0, A, lda, B, ldb, C, ldc, M, N, K, tiles_n);
    else hipLaunchKernelGGL((gemm_nolds<TM, TN, KS, PF>), grid, block, 0, 0, A, lda, B, ldb, C, ldc, M, N, K, tiles_n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("  %-22s wgs=%5d  %7.2f us  %6.1f TF/s  max|err|=%.2e (max|ref| %.1f)\n", name, tiles_m * tiles_n, us,
           2.0 * M * N * K / us * 1e-6, err, mx);
}

int main(int argc, char** argv) {
    const int shapes[][3] = {{596, 1956, 608}, {596, 600, 1952}, {224, 800, 976}, {596, 600, 112}, {224, 200, 800},
                             {4096, 4096, 2048}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K), ref((size_t)M * N);
        srand(1);
        for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
        for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
        if ((double)M * N * K < 2e9) {
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < N; ++n) {
                    double a = 0;
                    for (int k = 0; k < K; ++k) a += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
                    ref[(size_t)m * N + n] = (float)a;
                }
        }
        float *A, *B, *C;
        CK(hipMalloc(&A, sizeof(float) * M * K));
        CK(hipMalloc(&B, sizeof(float) * N * K));
        CK(hipMalloc(&C, sizeof(float) * M * N));
        CK(hipMemcpy(A, hA.data(), sizeof(float) * M * K, hipMemcpyHostToDevice));
        CK(hipMemcpy(B, hB.data(), sizeof(float) * N * K, hipMemcpyHostToDevice));
        printf("M=%d N=%d K=%d\n", M, N, K);
        const int reps = 50;
        run<32, 32, 4, 2>("32x32 KS4 PF2", A, B, C, M, N, K, ref, reps);
        run<64, 64, 1, 4>("64x64 KS1 PF4", A, B, C, M, N, K, ref, reps);
        run<32, 32, 4, 2, true>("m16 32x32 KS4 PF2", A, B, C, M, N, K, ref, reps);
        run<32, 32, 4, 3, true>("m16 32x32 KS4 PF3", A, B, C, M, N, K, ref, reps);
        run<64, 32, 4, 2, true>("m16 64x32 KS4 PF2", A, B, C, M, N, K, ref, reps);
        run<32, 64, 4, 2, true>("m16 32x64 KS4 PF2", A, B, C, M, N, K, ref, reps);
        run<64, 32, 2, 2, true>("m16 64x32 KS2 PF2", A, B, C, M, N, K, ref, reps);
        run<64, 64, 4, 2, true>("m16 64x64 KS4 PF2", A, B, C, M, N, K, ref, reps);
        run<64, 64, 2, 2, true>("m16 64x64 KS2 PF2", A, B, C, M, N, K, ref, reps);
        run<64, 64, 1, 3, true>("m16 64x64 KS1 PF3", A, B, C, M, N, K, ref, reps);
        run<48, 32, 4, 2, true>("m16 48x32 KS4 PF2", A, B, C, M, N, K, ref, reps);
        run<80, 16, 4, 2, true>("m16 80x16 KS4 PF2", A, B, C, M, N, K, ref, reps);
        run<80, 32, 4, 2, true>("m16 80x32 KS4 PF2", A, B, C, M, N, K, ref, reps);
        CK(hipFree(A));
        CK(hipFree(B));
        CK(hipFree(C));
    }
    return 0;
}
