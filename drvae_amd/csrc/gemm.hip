// fp32 MFMA GEMM family for the Linear layers of the Dr.VAE hot path (gfx950 only).
//
//   C[M,N] = epilogue(alpha * sum_k Aop[m,k] * Bop[k,n]) + beta*C
//
// One LDS-tiled kernel template covers forward (x W^T), backward-data (dy W) and
// backward-weight (dy^T x) by choosing which operand index is contiguous in memory;
// tiles are staged global -> VGPR -> LDS in their natural memory layout (coalesced
// along the contiguous index), and MFMA fragments are read from LDS as ds_read_b128
// (k-contiguous operand) or ds_read_b32 (row-contiguous operand, conflict-free).
//
// Matrix core: v_mfma_f32_32x32x2_f32 -- exact fp32 fma chain (no xf32/TF32 on gfx950),
// A: lane l holds A[i=l&31][k=l>>5], B: B[k=l>>5][j=l&31],
// C/D: reg r of lane l is row (r&3)+8*(r>>2)+4*(l>>5), col l&31.
// The k positions fed to the two lane halves are a permutation of the K tile (half h
// takes k in [h*KH,(h+1)*KH)): both operands use the same permutation, so each
// fragment is KH *contiguous* k values = vector LDS reads.
//
// Three tilings (all 256 threads = 4 wave64):
//   T64 : 64x64x32 tile, 2x2 waves, one 32x32 accumulator each      (default)
//   T32K: 32x32x64 tile, the 4 waves split K and reduce through LDS  (small M*N: 4x the workgroups)
//   T128: 128x128x32 tile, 2x2 waves, 2x2 accumulators each          (wide config)
#include "dv_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct LoadCfg {
    int vecA, vecB;   // widest aligned vector width (4, 2 or 1 floats) per operand
};

__device__ __forceinline__ float4 ld_chunk(const float* p, int valid, int vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid >= 4) {
        if (vec == 4) {
            v = *reinterpret_cast<const float4*>(p);
        } else if (vec == 2) {
            float2 a = *reinterpret_cast<const float2*>(p);
            float2 b = *reinterpret_cast<const float2*>(p + 2);
            v = make_float4(a.x, a.y, b.x, b.y);
        } else {
            v = make_float4(p[0], p[1], p[2], p[3]);
        }
    } else if (valid > 0) {
        v.x = p[0];
        if (valid > 1) v.y = p[1];
        if (valid > 2) v.z = p[2];
    }
    return v;
}

__device__ __forceinline__ int clamp04(int v) { return v < 0 ? 0 : (v > 4 ? 4 : v); }

// epilogue + store of one element
__device__ __forceinline__ void epi_store(const dv_gemm_desc& g, int row, int col, float v) {
    if (row >= g.M || col >= g.N) return;
    v *= g.alpha;
    if (g.epilogue == DV_EPI_FWD) {
        if (g.scale) v *= g.scale[col];
        if (g.bias) v += g.bias[col];
        const bool first = col < g.split;
        v = dv_act(first ? g.act0 : g.act1, v) + (first ? g.shift0 : g.shift1);
        if (g.resid && col < g.resid_cols) v += g.resid[(int64_t)row * g.ldr + col];
    } else if (g.epilogue == DV_EPI_BWD) {
        const bool first = col < g.split;
        const float y = g.yref[(int64_t)row * g.ldy + col] - (first ? g.shift0 : g.shift1);
        v *= dv_dact_from_y(first ? g.act0 : g.act1, y);
    }
    float* c = g.C + (int64_t)row * g.ldc + col;
    if (g.beta != 0.f) v += g.beta * (*c);
    *c = v;
}

template <int BM, int BN, int BK, int WM, int WN, int KS, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_kernel(const dv_gemm_desc g, const LoadCfg lc) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int KW = BK / KS, KH = KW / 2;
    constexpr int LDA_S = (AKC ? BK : BM) + 4;
    constexpr int LDB_S = (BKC ? BK : BN) + 4;
    constexpr int A_ELEMS = (AKC ? BM : BK) * LDA_S;
    constexpr int B_ELEMS = (BKC ? BN : BK) * LDB_S;
    constexpr int RED_ELEMS = (KS > 1) ? KS * 32 * 33 : 0;
    constexpr int SMEM = (A_ELEMS + B_ELEMS) > RED_ELEMS ? (A_ELEMS + B_ELEMS) : RED_ELEMS;
    static_assert(WM * WN * KS == 4, "4 waves");
    static_assert(KS == 1 || (BM == 32 && BN == 32), "K-split tiling is 32x32");
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* sA = smem;
    float* sB = smem + A_ELEMS;

    // ---- workgroup -> tile, XCD-aware: blocks b and b+8 share an XCD (and its L2), so give
    // each XCD a contiguous run of tiles (neighbouring tiles share an A row-panel).
    const int tiles_n = (g.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int m0 = (bid / tiles_n) * BM;
    const int n0 = (bid % tiles_n) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int ks_id = wave / (WM * WN);
    const int wm = (wave % (WM * WN)) / WN, wn = wave % WN;

    // ---- staging geometry: a chunk = 4 consecutive floats along the contiguous global index
    constexpr int A_CPL = (AKC ? BK : BM) / 4;      // chunks per staged line
    constexpr int A_LPP = 256 / A_CPL;              // lines per pass
    constexpr int A_NP = (AKC ? BM : BK) / A_LPP;   // passes
    constexpr int B_CPL = (BKC ? BK : BN) / 4;
    constexpr int B_LPP = 256 / B_CPL;
    constexpr int B_NP = (BKC ? BN : BK) / B_LPP;
    static_assert(A_NP >= 1 && B_NP >= 1, "tile too small for 256 threads");
    const int a_c = tid % A_CPL, a_l = tid / A_CPL;
    const int b_c = tid % B_CPL, b_l = tid / B_CPL;
    float4 ra[A_NP], rb[B_NP];
    // fused bias gradient: the first column-tile of every row-panel sums its A tiles over k
    const bool do_colsum = !AKC && g.a_colsum != nullptr && n0 == 0;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);

    auto load_a = [&](int k0) {
#pragma unroll
        for (int p = 0; p < A_NP; ++p) {
            const int line = a_l + p * A_LPP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (AKC) {   // line = m, chunk along k
                const int m = m0 + line, k = k0 + a_c * 4;
                if (m < g.M && k < g.K) {
                    if (lc.vecA > 1) {
                        if (g.A2 && k >= g.K1)
                            v = ld_chunk(g.A2 + (int64_t)m * g.lda2 + (k - g.K1), clamp04(g.K - k), lc.vecA);
                        else
                            v = ld_chunk(g.A + (int64_t)m * g.lda + k, clamp04((g.A2 ? g.K1 : g.K) - k), lc.vecA);
                    } else {
                        float e[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int kk = k + j;
                            e[j] = kk >= g.K ? 0.f
                                   : (g.A2 && kk >= g.K1) ? g.A2[(int64_t)m * g.lda2 + (kk - g.K1)]
                                                          : g.A[(int64_t)m * g.lda + kk];
                        }
                        v = make_float4(e[0], e[1], e[2], e[3]);
                    }
                    if (g.a_kscale) {
                        v.x *= g.a_kscale[k];
                        if (k + 1 < g.K) v.y *= g.a_kscale[k + 1];
                        if (k + 2 < g.K) v.z *= g.a_kscale[k + 2];
                        if (k + 3 < g.K) v.w *= g.a_kscale[k + 3];
                    }
                }
            } else {     // line = k, chunk along m
                const int k = k0 + line, m = m0 + a_c * 4;
                if (k < g.K && m < g.M) v = ld_chunk(g.A + (int64_t)k * g.lda + m, clamp04(g.M - m), lc.vecA);
                if (do_colsum) {
                    csum.x += v.x;
                    csum.y += v.y;
                    csum.z += v.z;
                    csum.w += v.w;
                }
            }
            ra[p] = v;
        }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int p = 0; p < B_NP; ++p) {
            const int line = b_l + p * B_LPP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BKC) {   // line = n, chunk along k
                const int n = n0 + line, k = k0 + b_c * 4;
                if (n < g.N && k < g.K) v = ld_chunk(g.B + (int64_t)n * g.ldb + k, clamp04(g.K - k), lc.vecB);
            } else {     // line = k, chunk along n
                const int k = k0 + line, n = n0 + b_c * 4;
                if (k < g.K && n < g.N) v = ld_chunk(g.B + (int64_t)k * g.ldb + n, clamp04(g.N - n), lc.vecB);
            }
            rb[p] = v;
        }
    };
    auto store_ab = [&]() {
#pragma unroll
        for (int p = 0; p < A_NP; ++p)
            *reinterpret_cast<float4*>(&sA[(a_l + p * A_LPP) * LDA_S + a_c * 4]) = ra[p];
#pragma unroll
        for (int p = 0; p < B_NP; ++p)
            *reinterpret_cast<float4*>(&sB[(b_l + p * B_LPP) * LDB_S + b_c * 4]) = rb[p];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = (g.K + BK - 1) / BK;
    const int kb = ks_id * KW + lh * KH;   // this lane's first k inside the tile
    load_a(0);
    load_b(0);
    store_ab();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) {   // global loads of the next tile fly under this tile's MFMAs
            load_a((kt + 1) * BK);
            load_b((kt + 1) * BK);
        }
#pragma unroll
        for (int s = 0; s < KH; s += 4) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * TM * 32 + i * 32 + li;
                if (AKC) {
                    fa[i] = *reinterpret_cast<const float4*>(&sA[row * LDA_S + kb + s]);
                } else {
                    fa[i] = make_float4(sA[(kb + s) * LDA_S + row], sA[(kb + s + 1) * LDA_S + row],
                                        sA[(kb + s + 2) * LDA_S + row], sA[(kb + s + 3) * LDA_S + row]);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * TN * 32 + j * 32 + li;
                if (BKC) {
                    fb[j] = *reinterpret_cast<const float4*>(&sB[col * LDB_S + kb + s]);
                } else {
                    fb[j] = make_float4(sB[(kb + s) * LDB_S + col], sB[(kb + s + 1) * LDB_S + col],
                                        sB[(kb + s + 2) * LDB_S + col], sB[(kb + s + 3) * LDB_S + col]);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            store_ab();
            __syncthreads();
        }
    }

    if (do_colsum) {   // block-uniform; the main loop ended on a barrier, the tiles are dead
        reinterpret_cast<float4*>(smem)[tid] = csum;
        __syncthreads();
        if (tid < BM) {
            const int ch = tid >> 2, e = tid & 3;
            float s = 0.f;
            for (int l = 0; l < A_LPP; ++l) s += smem[(l * A_CPL + ch) * 4 + e];
            const int m = m0 + tid;
            if (m < g.M) g.a_colsum[m] = (g.colsum_beta != 0.f ? g.colsum_beta * g.a_colsum[m] : 0.f) + s;
        }
        __syncthreads();
    }

    if (KS == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int col = n0 + wn * TN * 32 + j * 32 + li;
                    epi_store(g, row, col, acc[i][j][r]);
                }
    } else {
        // the 4 waves hold partial sums over disjoint k: reduce through LDS (tiles are dead now)
        float* red = smem;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            red[(ks_id * 32 + row) * 33 + li] = acc[0][0][r];
        }
        __syncthreads();
        const int row = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < KS; ++w) v += red[(w * 32 + row) * 33 + c4 + e];
            epi_store(g, m0 + row, n0 + c4 + e, v);
        }
    }
}

inline int vec_width(const void* p, int64_t ld) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    if ((a & 15) == 0 && (ld & 3) == 0) return 4;
    if ((a & 7) == 0 && (ld & 1) == 0) return 2;
    return 1;
}

template <int BM, int BN, int BK, int WM, int WN, int KS>
int launch_cfg(const dv_gemm_desc& g, const LoadCfg& lc, hipStream_t st) {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    dim3 grid(tiles), block(256);
    if (g.a_kcontig && g.b_kcontig)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, WM, WN, KS, true, true>), grid, block, 0, st, g, lc);
    else if (g.a_kcontig && !g.b_kcontig)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, WM, WN, KS, true, false>), grid, block, 0, st, g, lc);
    else if (!g.a_kcontig && !g.b_kcontig)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, WM, WN, KS, false, false>), grid, block, 0, st, g, lc);
    else
        return DV_ERR_UNSUPPORTED;
    DV_RETURN_LAUNCH();
}

}  // namespace

static int g_force_tiling = 0;   // 0 = heuristic; 1 = T64, 2 = T32K, 3 = T128 (tests / tuning)

extern "C" int dv_gemm_force_tiling(int t) {
    g_force_tiling = t;
    return DV_OK;
}

extern "C" int dv_gemm(const dv_gemm_desc* d, dv_stream_t stream) {
    DV_REQUIRE(d != nullptr);
    const dv_gemm_desc& g = *d;
    DV_REQUIRE(g.M >= 0 && g.N >= 0 && g.K >= 0);
    if (g.M == 0 || g.N == 0) return DV_OK;
    DV_REQUIRE(g.A && g.B && g.C);
    DV_REQUIRE(g.epilogue == DV_EPI_PLAIN || g.epilogue == DV_EPI_FWD || g.epilogue == DV_EPI_BWD);
    DV_REQUIRE(g.epilogue != DV_EPI_BWD || g.yref != nullptr);
    DV_REQUIRE(g.A2 == nullptr || (g.a_kcontig && g.K1 >= 0 && g.K1 <= g.K));
    DV_REQUIRE(g.a_kscale == nullptr || g.a_kcontig);
    DV_REQUIRE(g.a_colsum == nullptr || !g.a_kcontig);
    LoadCfg lc;
    lc.vecA = vec_width(g.A, g.lda);
    if (g.A2) {
        const int v2 = vec_width(g.A2, g.lda2);
        lc.vecA = v2 < lc.vecA ? v2 : lc.vecA;
        if (g.K1 & 3) lc.vecA = 1;   // a 4-chunk could straddle the two sources
    }
    lc.vecB = vec_width(g.B, g.ldb);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t t64 = (int64_t)((g.M + 63) / 64) * ((g.N + 63) / 64);
    const int64_t t128 = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128);
    int tiling = g_force_tiling;
    if (tiling == 0) tiling = (t128 >= 512) ? 3 : (t64 >= 192 ? 1 : 2);
    if (tiling == 3) return launch_cfg<128, 128, 32, 2, 2, 1>(g, lc, st);
    if (tiling == 1) return launch_cfg<64, 64, 32, 2, 2, 1>(g, lc, st);
    return launch_cfg<32, 32, 64, 1, 1, 4>(g, lc, st);
}
