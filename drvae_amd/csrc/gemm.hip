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
#include "gemm_common.inc"

namespace {

#include "gemm_pipe.inc"


// One staged operand.  Element (line, pos) lives at base[line*ld + pos]:
//   k-contiguous operand  : line = output row (m or n), pos = k
//   row-contiguous operand: line = k,                   pos = output row
// A staging chunk is 4 consecutive `pos` values of one line.
struct Operand {
    const float* base;
    int64_t ld;
    int n_line, n_pos;   // extents along line / pos
};

// Fast path: the whole chunk is in bounds -> ONE unconditional (vector) load, no branches, so
// all of a thread's staging loads are in flight together (a per-load `if` makes hipcc wait
// vmcnt(0) after each one: dependent L2 round trips).
__device__ __forceinline__ float4 ld_fast(const Operand& o, int line, int pos, int vec) {
    const float* p = o.base + (int64_t)line * o.ld + pos;
    if (vec == 4) return *reinterpret_cast<const float4*>(p);
    if (vec == 2) {
        const float2 a = *reinterpret_cast<const float2*>(p), b = *reinterpret_cast<const float2*>(p + 2);
        return make_float4(a.x, a.y, b.x, b.y);
    }
    return make_float4(p[0], p[1], p[2], p[3]);
}

// Edge path: indices clamped into the matrix, out-of-range elements selected to zero --
// still branch-free (4 scalar loads + v_cndmask).
__device__ __forceinline__ float4 ld_edge(const Operand& o, int line, int pos) {
    const bool lok = line < o.n_line;
    const float* row = o.base + (int64_t)(lok ? line : o.n_line - 1) * o.ld;
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = pos + j;
        const float v = row[q < o.n_pos ? q : o.n_pos - 1];
        e[j] = (lok && q < o.n_pos) ? v : 0.f;
    }
    return make_float4(e[0], e[1], e[2], e[3]);
}

// first of the RPT consecutive rows of a paired-heads tile that the 16- / 32-lane group `grp` of the epilogue owns.  Two
// neighbouring groups share a 32-lane LDS access; with the reduction buffer's row stride of BN + 1 floats, rows r and
// r + RPT put them on overlapping banks (0.11-0.22 LDS bank conflicts per access in the round-4 counters) -- half a tile
// apart (BM / 2 rows = 16 banks) they are disjoint.  A bijection of the groups onto the tile's rows.
template <int BM, int RPT>
__device__ __forceinline__ int heads_row0(int grp) { return (grp & 1) * (BM / 2) + (grp >> 1) * RPT; }

// Epilogue of a paired-heads tile: thread -> (column c of the half tile, RPT consecutive rows); both heads of
// an element are reduced over the KS partial sums by the same thread, pass through the DV_EPI_FWD column
// epilogue, and feed the row work of the mode (see dv_heads_epi in drvae_hip.h).
template <int BM, int BN, int KS, int NT>
__device__ __forceinline__ void heads_epilogue(const dv_gemm_desc& g, const dv_heads_epi& he, const float* red, int m0,
                                               int tn, int tiles_n, int pre_s0, int pre_s1) {
    constexpr int HB = BN / 2, RPT = BM / (NT / HB);
    constexpr bool PRE = (KS == 8 && BN == 32);      // (segment bounds of the thread's one row loaded by the caller)
    static_assert(!PRE || RPT == 1, "prefetched bounds: one row per thread");
    static_assert((HB == 32 || HB == 16) && RPT >= 1, "half tile = 16 or 32 adjacent lanes");
    static_assert((NT / HB) % 2 == 0 && (BM / 2) % RPT == 0, "row groups come in pairs");
    const int tid = threadIdx.x, c = tid % HB, r0 = heads_row0<BM, RPT>(tid / HB);
    const int col0 = tn * HB + c, col1 = g.split + col0;
    const bool ok = col0 < g.split && col1 < g.N;
    const int cc0 = ok ? col0 : g.split - 1, cc1 = ok ? col1 : g.N - 1;
    float sc0 = 1.f, sc1 = 1.f, bi0 = 0.f, bi1 = 0.f;
    if (g.scale) {
        sc0 = g.scale[cc0];
        sc1 = g.scale[cc1];
    }
    if (g.bias) {
        bi0 = g.bias[cc0];
        bi1 = g.bias[cc1];
    }
    const bool use_res = g.resid != nullptr && col0 < g.resid_cols;
#pragma unroll
    for (int e = 0; e < RPT; ++e) {
        const int row = m0 + r0 + e;
        const bool rok = row < g.M;
        const int rc = rok ? row : g.M - 1;
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int w = 0; w < KS; ++w) {
            v0 += red[(w * BM + r0 + e) * (BN + 1) + c];
            v1 += red[(w * BM + r0 + e) * (BN + 1) + HB + c];
        }
        float a0 = dv_act(g.act0, v0 * g.alpha * sc0 + bi0) + g.shift0;
        if (use_res) a0 += g.resid[(int64_t)rc * g.ldr + cc0];
        if (he.mode == DV_HEADS_NLL && g.act1 == DV_ACT_SOFTPLUS) {
            // the decoder's sigma head on hardware transcendentals (one exp, two logs, two reciprocals per element):
            //   p = pre-activation, e = exp(-|p|): softplus(p) = max(p, 0) + log(1 + e), sigmoid(p) = (p >= 0 ? 1 : e) / (1 + e)
            const float pre = v1 * g.alpha * sc1 + bi1;
            const float e_ = __expf(-fabsf(pre));
            const float r1 = __frcp_rn(1.f + e_);
            const float sd = fmaxf(pre, 0.f) + __logf(1.f + e_) + g.shift1;
            const float sig = (pre >= 0.f ? 1.f : e_) * r1;
            const float is = __frcp_rn(sd);
            const float xv = he.x[(int64_t)(he.xidx ? he.xidx[rc] : rc) * he.ldx + cc0];
            const float cf = he.coef[rc];
            const float t = (xv - a0) * is;
            float acc = ok ? kLog2PiG + 2.f * __logf(sd) + t * t : 0.f;
            if (ok && rok) {
                g.C[(int64_t)row * g.ldc + col0] = cf * t * is;
                g.C[(int64_t)row * g.ldc + col1] = cf * (t * t - 1.f) * is * sig;
            }
#pragma unroll
            for (int off = HB / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (c == 0 && rok) he.part[(int64_t)row * tiles_n + tn] = -0.5f * acc;
            continue;
        }
        const float a1 = dv_act(g.act1, v1 * g.alpha * sc1 + bi1) + g.shift1;
        if (he.mode == DV_HEADS_SAMPLE) {
            if (ok && rok) {
                g.C[(int64_t)row * g.ldc + col0] = a0;
                g.C[(int64_t)row * g.ldc + col1] = a1;
                if (row < he.n_src) {
                    const float std_ = expf(0.5f * a1);
                    const int s0 = he.seg_ptr ? (PRE ? pre_s0 : he.seg_ptr[row]) : row;
                    const int s1 = he.seg_ptr ? (PRE ? pre_s1 : he.seg_ptr[row + 1]) : row + 1;
                    for (int t = s0; t < s1; ++t) {
                        const int64_t s = he.seg_rows ? he.seg_rows[t] : t;
                        const float z = he.eps[s * he.lde + col0] * std_ + a0;
                        he.out[s * he.ldo + col0] = z;
                        if (he.out2) he.out2[s * he.ldo2 + col0] = z - he.sub[s * he.lds + col0];
                        if (he.out3) {
                            const int t3 = he.out3_idx[s];
                            if (t3 >= 0) he.out3[(int64_t)t3 * he.ldo3 + col0] = z;
                        }
                        if (he.out4)      // CSR fan-out of sample row s: rows [out4_ptr[s], out4_ptr[s+1]) of out4
                            for (int u = he.out4_ptr[s]; u < he.out4_ptr[s + 1]; ++u) he.out4[(int64_t)u * he.ldo4 + col0] = z;
                    }
                }
            }
        } else {   // DV_HEADS_NLL
            const float xv = he.x[(int64_t)(he.xidx ? he.xidx[rc] : rc) * he.ldx + cc0];
            const float cf = he.coef[rc];
            const float d = xv - a0, v = a1 * a1;
            float acc = ok ? kLog2PiG + logf(v) + d * d / v : 0.f;
            float gm = d / v, gs = -1.f / a1 + d * d / (v * a1);
            if (g.act1 != DV_ACT_IDENTITY) gs *= dv_dact_from_y(g.act1, a1 - g.shift1);
            gm *= cf;
            gs *= cf;
            if (ok && rok) {
                g.C[(int64_t)row * g.ldc + col0] = gm;
                g.C[(int64_t)row * g.ldc + col1] = gs;
            }
            // sum over the 32 columns of the half tile = the 32 lanes of this half-wave, fixed order
#pragma unroll
            for (int off = HB / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (c == 0 && rok) he.part[(int64_t)row * tiles_n + tn] = -0.5f * acc;
        }
    }
}

template <int BM, int BN, int BK, int KS, bool AKC, bool BKC, int PS = 0>
constexpr int gemm_smem_floats() {
    if constexpr (PS > 0) return PipeGeom<BM, BN, BK, BM / 32, BN / 32, KS, PS, AKC, BKC>::SMEM_FL;
    constexpr int stage = (AKC ? BM : BK) * ((AKC ? BK : BM) + 4) + (BKC ? BN : BK) * ((BKC ? BK : BN) + 4);
    constexpr int red = (KS > 1) ? KS * BM * (BN + 1) : 0;
    return 2 * stage > red ? 2 * stage : red;
}

// One workgroup's share of one product: `bid` of `nwg` workgroups.  PAIR (dv_gemm_heads): the BN staged B lines
// are the head-0 rows [tn*BN/2, +BN/2) followed by the head-1 rows [split + tn*BN/2, +BN/2) of W, and the
// epilogue sees both heads of an element in one thread.
// MI16: the product runs on v_mfma_f32_16x16x4_f32 (16x16 accumulator tiles, four k-groups of lanes) instead of
// v_mfma_f32_32x32x2_f32 -- the same matrix-pipe cycles, LDS reads and registers per output element.
// PS > 0: the K loop is the hand-pipelined LDS-DMA ring of PS slots (gemm_pipe.inc; the launcher has checked pipe_ok)
template <int BM, int BN, int BK, int WM, int WN, int KS, bool AKC, bool BKC, bool PAIR = false, bool MI16 = false, int PS = 0>
__device__ __forceinline__ void gemm_body(const dv_gemm_desc& g, const LoadCfg& lc, float* smem, int bid, int nwg,
                                          const dv_heads_epi* he = nullptr) {
    static_assert(!PAIR || (AKC && BKC && KS > 1 && WM == 1 && WN == 1), "paired heads: forward layout, K-split tiling");
    constexpr int HB = BN / 2;
    static_assert(!MI16 || (KS == 1 && !PAIR), "16x16x4 tiles: unsplit K only");
    constexpr int TS = MI16 ? 16 : 32;            // edge of an MFMA output tile
    constexpr int AR = MI16 ? 4 : 16;             // accumulator registers per tile
    constexpr int TM = BM / WM / TS, TN = BN / WN / TS;
    constexpr int KW = BK / KS, KH = KW / (MI16 ? 4 : 2);     // consecutive k of a K tile held by one lane
    constexpr int LDA_S = (AKC ? BK : BM) + 4;
    constexpr int LDB_S = (BKC ? BK : BN) + 4;
    constexpr int A_ELEMS = (AKC ? BM : BK) * LDA_S;
    constexpr int B_ELEMS = (BKC ? BN : BK) * LDB_S;
    constexpr int STAGE = A_ELEMS + B_ELEMS;
    constexpr int RED_ELEMS = (KS > 1) ? KS * BM * (BN + 1) : 0;
    constexpr int NT = 64 * WM * WN * KS;   // threads per workgroup
    static_assert(PS > 0 || (2 * STAGE <= gemm_smem_floats<BM, BN, BK, KS, AKC, BKC>() && RED_ELEMS <= gemm_smem_floats<BM, BN, BK, KS, AKC, BKC>()), "smem");
    static_assert(PS == 0 || (!MI16 && TM == 1 && TN == 1), "pipe loop inside gemm_body: one 32x32 tile per wave");
    static_assert(NT == 256 || NT == 512 || NT == 1024, "4, 8 or 16 waves");

    const bool ones_col = BM < 128 && !AKC && !BKC && (g.flags & DV_FLAG_ONES_COL);
    const int tiles_m = (g.M + BM - 1) / BM;
    const int tiles_n = PAIR ? (g.split + HB - 1) / HB : (g.N + (ones_col ? DV_ONES_OFF(g) + 1 : 0) + BN - 1) / BN;
    int tm, tn;
    tile_of_block(bid, nwg, tiles_m, tiles_n, lc.map, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    // row of B (= output column) staged as line l of the B tile, clamped into its head
    auto bline = [&](int l) -> int {
        if (PAIR) {
            const int c = tn * HB + (l < HB ? l : l - HB);
            return l < HB ? (c < g.split ? c : g.split - 1) : (g.split + c < g.N ? g.split + c : g.N - 1);
        }
        return n0 + l < g.N ? n0 + l : g.N - 1;
    };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = MI16 ? (lane & 15) : (lane & 31), lh = MI16 ? (lane >> 4) : (lane >> 5);
    const int ks_id = wave / (WM * WN);
    const int wm = (wave % (WM * WN)) / WN, wn = wave % WN;

    // ---- staging geometry: a chunk = 4 consecutive floats along the contiguous global index
    constexpr int A_CPL = (AKC ? BK : BM) / 4;      // chunks per staged line
    constexpr int A_LPP = NT / A_CPL;               // lines per pass
    constexpr int A_NP = (AKC ? BM : BK) / A_LPP;   // passes
    constexpr int B_CPL = (BKC ? BK : BN) / 4;
    constexpr int B_LPP = NT / B_CPL;
    constexpr int B_NP = (BKC ? BN : BK) / B_LPP;
    static_assert(A_NP >= 1 && B_NP >= 1, "tile too small for the workgroup");
    const int a_c = tid % A_CPL, a_l = tid / A_CPL;
    const int b_c = tid % B_CPL, b_l = tid / B_CPL;
    // fused bias gradient (dy^T x layout, 32- and 64-wide tiles): the staged B column N -- the first one past the
    // matrix: padding of the last column tile, or the single useful column of one extra tile per row panel -- is
    // overwritten with ones on its way into LDS, so the accumulator column N comes out as sum_k A[k, m];
    // `oc` = which element of this thread's B chunk that column is (-1: none)
    const int oc = (ones_col && n0 <= g.N && g.N < n0 + BN) ? g.N - n0 - b_c * 4 : -1;

    const Operand oa{g.A, g.lda, AKC ? g.M : g.K, AKC ? g.K : g.M};
    const Operand ob{g.B, g.ldb, BKC ? g.N : g.K, BKC ? g.K : g.N};
    // workgroup-uniform: may this tile use the unconditional vector loads?  k-contiguous
    // operands only need a full K tile (rows past the edge are clamped: their products are
    // never stored); row-contiguous operands also need the row range interior.
    const bool a_plain = g.A2 == nullptr;
    const bool a_int = AKC ? true : (m0 + BM <= g.M);
    const bool b_int = BKC ? true : (n0 + BN <= g.N);
    // edge tiles of row-contiguous operands may still use the aligned vector loads when the
    // caller vouches that a row end can be over-read (flags); chunks entirely past the edge are
    // redirected to chunk 0 of the line (their values only feed outputs that are never stored)
    const bool wg_fast = a_plain && (a_int || (g.flags & 1)) && (b_int || (g.flags & 2)) && g.a_kscale == nullptr;

    // generic (edge / two-source / k-scaled) staging loads: clamped indices + selects, branch-free per element
    auto gen_load = [&](int k0, float4 (&ra)[A_NP], float4 (&rb)[B_NP]) {
#pragma unroll
        for (int p = 0; p < A_NP; ++p) {
            const int l = a_l + p * A_LPP;
            const int line = AKC ? (m0 + l) : (k0 + l);
            const int pos = AKC ? (k0 + a_c * 4) : (m0 + a_c * 4);
            float4 v;
            if (a_plain) {
                v = ld_edge(oa, line, pos);
            } else {   // two concatenated sources (AKC only): per-element source select
                const bool lok = line < g.M;
                const int64_t r = lok ? line : g.M - 1;
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = pos + j, qc = q < g.K ? q : g.K - 1;
                    const float* src = qc >= g.K1 ? g.A2 + r * g.lda2 + (qc - g.K1) : g.A + r * g.lda + qc;
                    const float x = *src;
                    e[j] = (lok && q < g.K) ? x : 0.f;
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            }
            if (AKC && g.a_kscale) {   // uniform
                const int kq = g.K - 1;
                v.x *= g.a_kscale[pos < kq ? pos : kq];
                v.y *= g.a_kscale[pos + 1 < kq ? pos + 1 : kq];
                v.z *= g.a_kscale[pos + 2 < kq ? pos + 2 : kq];
                v.w *= g.a_kscale[pos + 3 < kq ? pos + 3 : kq];
            }
            ra[p] = v;
        }
#pragma unroll
        for (int p = 0; p < B_NP; ++p) {
            const int l = b_l + p * B_LPP;
            rb[p] = ld_edge(ob, BKC ? (PAIR ? bline(l) : n0 + l) : (k0 + l), BKC ? (k0 + b_c * 4) : (n0 + b_c * 4));
        }
    };
    auto stage_store = [&](float* buf, const float4 (&ra)[A_NP], const float4 (&rb)[B_NP]) {
        float* sA = buf;
        float* sB = buf + A_ELEMS;
#pragma unroll
        for (int p = 0; p < A_NP; ++p) {
            *reinterpret_cast<float4*>(&sA[(a_l + p * A_LPP) * LDA_S + a_c * 4]) = ra[p];
        }
#pragma unroll
        for (int p = 0; p < B_NP; ++p) {
            float4 v = rb[p];
            if (BM < 128 && !AKC && !BKC) {
                v.x = oc == 0 ? 1.f : v.x;
                v.y = oc == 1 ? 1.f : v.y;
                v.z = oc == 2 ? 1.f : v.z;
                v.w = oc == 3 ? 1.f : v.w;
            }
            *reinterpret_cast<float4*>(&sB[(b_l + p * B_LPP) * LDB_S + b_c * 4]) = v;
        }
    };

    typename std::conditional<MI16, f32x4, f32x16>::type acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < AR; ++r) acc[i][j][r] = 0.f;

    // (paired heads, sample epilogue, one output row per thread: the row's segment bounds are loaded HERE, under the K
    // loop, instead of at the head of the epilogue's dependent chain bounds -> sample rows -> noise -> stores: cfg 2
    // 0.1945 -> 0.1925 ms in a same-box A/B.  Fetching the sample rows behind the first barrier as well gave it back)
    int pre_s0 = 0, pre_s1 = 0;
    if constexpr (PAIR && KS == 8 && BN == 32) {
        if (he->mode == DV_HEADS_SAMPLE && he->seg_ptr != nullptr) {
            const int prow = m0 + heads_row0<BM, 1>(tid / (BN / 2));      // (the row heads_epilogue gives this thread)
            if (prow < he->n_src && prow < g.M) {
                pre_s0 = he->seg_ptr[prow];
                pre_s1 = he->seg_ptr[prow + 1];
            }
        }
    }
    const int kb = ks_id * KW + lh * KH;   // this lane's first k inside the tile
    // all fragment reads of a tile are issued up front (in-order LDS returns: the first MFMA only
    // waits for the first read, the rest land under the MFMA chain)
    auto read_frags = [&](const float* sA, float (&fa)[TM][KH], float (&fb)[TN][KH]) {
        const float* sB = sA + A_ELEMS;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * TM * TS + i * TS + li;
#pragma unroll
            for (int s = 0; s < KH; s += 4) {
                if (AKC) {
                    const float4 v = *reinterpret_cast<const float4*>(&sA[row * LDA_S + kb + s]);
                    fa[i][s] = v.x;
                    fa[i][s + 1] = v.y;
                    fa[i][s + 2] = v.z;
                    fa[i][s + 3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fa[i][s + e] = sA[(kb + s + e) * LDA_S + row];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * TN * TS + j * TS + li;
#pragma unroll
            for (int s = 0; s < KH; s += 4) {
                if (BKC) {
                    const float4 v = *reinterpret_cast<const float4*>(&sB[col * LDB_S + kb + s]);
                    fb[j][s] = v.x;
                    fb[j][s + 1] = v.y;
                    fb[j][s + 2] = v.z;
                    fb[j][s + 3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fb[j][s + e] = sB[(kb + s + e) * LDB_S + col];
                }
            }
        }
    };
    auto mma = [&](const float (&fa)[TM][KH], const float (&fb)[TN][KH]) {
#pragma unroll
        for (int s = 0; s < KH; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (MI16)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
                    else if (DV_DBG & 2)
                        acc[i][j][s & 15] += fa[i][s] * fb[j][s];
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
                }
    };
    auto compute = [&](const float* sA) {
        float fa[TM][KH], fb[TN][KH];
        read_frags(sA, fa, fb);
        mma(fa, fb);
    };

    const int nkt = (g.K + BK - 1) / BK, nfull = g.K / BK;
    float* const buf0 = smem;
    float* const buf1 = smem + STAGE;
    if constexpr (PS > 0) {
        if (!(DV_DBG & 64))   // knock-out (tuning builds): no K loop at all
            pipe_loop<BM, BN, BK, WM, WN, KS, PS, AKC, BKC, (BM < 128 && !AKC && !BKC)>(
                g, smem, m0, n0, [&](int l) { return m0 + l < g.M ? m0 + l : g.M - 1; }, bline, acc);
        if (DV_DBG & 128) {   // knock-out (tuning builds): no reduction, no epilogue (the accumulators stay live)
            if (acc[0][0][0] == 12345.678f) g.C[0] = acc[0][0][1];
            return;
        }
        __syncthreads();      // (K-split reduction below reuses the ring)
    } else if (wg_fast && nfull >= 1) {
        // ---- interior workgroups: per-thread staging pointers advanced by one K tile per step,
        // unconditional vector loads, TWO register sets (two tiles in flight: each load has two
        // MFMA phases to land), LDS double buffer (one barrier per K tile).
        const float* pa[A_NP];
        const float* pb[B_NP];
#pragma unroll
        for (int p = 0; p < A_NP; ++p) {
            const int l = a_l + p * A_LPP;
            int line = AKC ? (m0 + l) : l;
            if (AKC) line = line < g.M ? line : g.M - 1;
            pa[p] = g.A + (int64_t)line * g.lda + (AKC ? a_c * 4 : (m0 + a_c * 4 < g.M ? m0 + a_c * 4 : 0));
        }
#pragma unroll
        for (int p = 0; p < B_NP; ++p) {
            const int l = b_l + p * B_LPP;
            const int line = BKC ? bline(l) : l;
            pb[p] = g.B + (int64_t)line * g.ldb + (BKC ? b_c * 4 : (n0 + b_c * 4 < g.N ? n0 + b_c * 4 : 0));
        }
        const int64_t step_a = AKC ? (int64_t)BK : (int64_t)BK * g.lda;
        const int64_t step_b = BKC ? (int64_t)BK : (int64_t)BK * g.ldb;
        const int va = lc.vecA, vb = lc.vecB;
        auto ldv = [](const float* p, int vec) -> float4 {
            if (vec == 4) return *reinterpret_cast<const float4*>(p);
            if (vec == 2) {
                const float2 a = *reinterpret_cast<const float2*>(p), b = *reinterpret_cast<const float2*>(p + 2);
                return make_float4(a.x, a.y, b.x, b.y);
            }
            return make_float4(p[0], p[1], p[2], p[3]);
        };
        // chunk of a k-contiguous operand in the K tail: `left` = K - (k of the chunk's first element)
        auto ldv_tail = [](const float* p, int vec, int left) -> float4 {
            const float* z = dv_zero_chunk;
            if (vec == 4) return *reinterpret_cast<const float4*>(left >= 4 ? p : z);
            if (vec == 2) {
                const float2 a = *reinterpret_cast<const float2*>(left >= 2 ? p : z);
                const float2 b = *reinterpret_cast<const float2*>(left >= 4 ? p + 2 : z);
                return make_float4(a.x, a.y, b.x, b.y);
            }
            return make_float4(*(left >= 1 ? p : z), *(left >= 2 ? p + 1 : z), *(left >= 3 ? p + 2 : z),
                               *(left >= 4 ? p + 3 : z));
        };
        auto fetch = [&](int t, float4 (&ra)[A_NP], float4 (&rb)[B_NP]) {
            if (t < nfull) {
#pragma unroll
                for (int p = 0; p < A_NP; ++p) {
                    ra[p] = ldv(pa[p], va);
                    pa[p] += step_a;
                }
#pragma unroll
                for (int p = 0; p < B_NP; ++p) {
                    rb[p] = ldv(pb[p], vb);
                    pb[p] += step_b;
                }
            } else {
                // the partial K tail tile: same loads, but every sub-load (of the widest width that divides
                // K, so that none straddles it) past K reads a block of zeros instead -- no branches, and
                // nothing here waits on the loads
                const int kt0 = nfull * BK;
#pragma unroll
                for (int p = 0; p < A_NP; ++p) {
                    if (AKC) {
                        ra[p] = ldv_tail(pa[p], lc.vecA_t, g.K - (kt0 + a_c * 4));
                    } else {
                        ra[p] = ldv((kt0 + a_l + p * A_LPP < g.K) ? pa[p] : dv_zero_chunk, va);
                    }
                }
#pragma unroll
                for (int p = 0; p < B_NP; ++p) {
                    if (BKC) {
                        rb[p] = ldv_tail(pb[p], lc.vecB_t, g.K - (kt0 + b_c * 4));
                    } else {
                        rb[p] = ldv((kt0 + b_l + p * B_LPP < g.K) ? pb[p] : dv_zero_chunk, vb);
                    }
                }
            }
        };
        float4 a0[A_NP], b0[B_NP], a1[A_NP], b1[B_NP];
#if DV_STAMP
        const bool stamp_on = dv_stamp_buf != nullptr && wave == 0;
        int si = 2;
#endif
        STAMP(0);
        fetch(0, a0, b0);
        if (nkt > 1) fetch(1, a1, b1);
        stage_store(buf0, a0, b0);
        if (nkt > 2) fetch(2, a0, b0);
        __syncthreads();
        STAMP(1);
        int kt = 0;
#if DV_DBG == 0
        // steady state (every tile touched is a full one): no conditionals, so each phase is ONE
        // scheduling region -- hipcc then spreads the LDS stores and global loads of the phase over its
        // MFMA chain instead of issuing them after it (they sat in a conditional block of their own);
        // measured: decoder-head products 26.4 -> 25.1 us, 128x128 tiling 118 -> 122 TF/s
        for (; kt + 4 < nfull; kt += 2) {
            compute(buf0);
#if DV_STAMP
            STAMP(si); ++si;
#endif
            stage_store(buf1, a1, b1);
#if DV_STAMP
            STAMP(si); ++si;
#endif
            fetch(0, a1, b1);
#if DV_STAMP
            STAMP(si); ++si;
#endif
            __syncthreads();
#if DV_STAMP
            STAMP(si); ++si;
#endif
            compute(buf1);
#if DV_STAMP
            STAMP(si); ++si;
#endif
            stage_store(buf0, a0, b0);
#if DV_STAMP
            STAMP(si); ++si;
#endif
            fetch(0, a0, b0);
#if DV_STAMP
            STAMP(si); ++si;
#endif
            __syncthreads();
#if DV_STAMP
            STAMP(si); ++si;
#endif
        }
#endif
        for (; kt < nkt; kt += 2) {
            compute(buf0);
            if (kt + 1 < nkt) {
                if (!(DV_DBG & 4)) stage_store(buf1, a1, b1);
                if (kt + 3 < nkt && !(DV_DBG & 1)) fetch(kt + 3, a1, b1);
            }
            if (!(DV_DBG & 8)) __syncthreads();
            if (kt + 1 >= nkt) break;
            compute(buf1);
            if (kt + 2 < nkt) {
                if (!(DV_DBG & 4)) stage_store(buf0, a0, b0);
                if (kt + 4 < nkt && !(DV_DBG & 1)) fetch(kt + 4, a0, b0);
            }
            if (!(DV_DBG & 8)) __syncthreads();
        }
    } else {
        // ---- edge tiles / concatenated sources / k-scaled operand: simple loop, generic loads
        float4 ra[A_NP], rb[B_NP];
        gen_load(0, ra, rb);
        stage_store(buf0, ra, rb);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + 1 < nkt) gen_load((kt + 1) * BK, ra, rb);
            compute((kt & 1) ? buf1 : buf0);
            if (kt + 1 < nkt) stage_store((kt & 1) ? buf0 : buf1, ra, rb);
            __syncthreads();
        }
    }

    if constexpr (MI16) {
        // a lane holds rows 4*(lane/16) .. +3 of column lane%16 of each 16x16 tile
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float a4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) a4[r] = acc[i][j][r];
                const int rbase = m0 + wm * TM * TS + i * TS + 4 * lh;
                epi_store_col<4, (BM < 128)>(g, a4, n0 + wn * TN * TS + j * TS + li, [rbase](int r) { return rbase + r; });
            }
    } else if constexpr (KS == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float a16[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) a16[r] = acc[i][j][r];
                const int rbase = m0 + wm * TM * 32 + i * 32 + 4 * lh;
                epi_store_col<16, (BM < 128)>(g, a16, n0 + wn * TN * 32 + j * 32 + li,
                                  [rbase](int r) { return rbase + (r & 3) + 8 * (r >> 2); });
            }
    } else {
        // the KS wave groups hold partial sums over disjoint k: reduce through LDS (tiles are dead now)
        float* red = smem;
        if constexpr (PAIR) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    red[(ks_id * BM + row) * (BN + 1) + j * 32 + li] = acc[0][j][r];
                }
            __syncthreads();
            heads_epilogue<BM, BN, KS, NT>(g, *he, red, m0, tn, tiles_n, pre_s0, pre_s1);
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int col = wn * TN * 32 + j * 32 + li;
                    red[(ks_id * BM + row) * (BN + 1) + col] = acc[i][j][r];
                }
        __syncthreads();
        // thread -> (RPT consecutive rows, one column): column-wise like the register epilogue
        constexpr int RPT = BM / (NT / BN);
        const int col = tid % BN, r0 = (tid / BN) * RPT;
        float a4[RPT];
#pragma unroll
        for (int e = 0; e < RPT; ++e) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < KS; ++w) v += red[(w * BM + r0 + e) * (BN + 1) + col];
            a4[e] = v;
        }
        const int rbase = m0 + r0;
        epi_store_col<RPT>(g, a4, n0 + col, [rbase](int r) { return rbase + r; });
    }
}


template <int BM, int BN, int BK, int WM, int WN, int KS, bool AKC, bool BKC, bool MI16 = false>
__global__ __launch_bounds__(64 * WM * WN * KS, (WM * WN * KS == 16) ? 4 : (BM >= 128 || BK >= 128) ? 2 : (BM * BN >= 6144) ? (WM * WN * KS) / 4 : (WM * WN * KS == 8 ? (BM * BN == 1024 ? DV_LB8 : 2) : (BK == 32 && BM == 32 ? ((AKC || BKC) ? DV_DENSE_WG : DV_DENSE_WG - 1) : 4))) void gemm_kernel(const dv_gemm_desc g, const LoadCfg lc) {
    __shared__ __attribute__((aligned(16))) float smem[gemm_smem_floats<BM, BN, BK, KS, AKC, BKC>()];
    publish_on_entry(g);
#if DV_STAGGER
    {   // tuning: de-phase the workgroups that share a CU (dispatch is round-robin over the CUs)
        const int ph = (blockIdx.x / DV_STAGGER) & 3;
        if (ph == 1) __builtin_amdgcn_s_sleep(6);
        if (ph == 2) __builtin_amdgcn_s_sleep(12);
        if (ph == 3) __builtin_amdgcn_s_sleep(18);
    }
#endif
    gemm_body<BM, BN, BK, WM, WN, KS, AKC, BKC, false, MI16>(g, lc, smem, blockIdx.x, gridDim.x);
}

#ifdef DV_LAB   // probe kernels of the tuning build (tools/gemm_lab.sh): never selected by the dispatcher
// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant of the 32x32 K-split tiling for the forward layout (both operands k-contiguous): the tiles go
// global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write), one 1-KiB piece = 4 tile
// rows x 256 B per wave-instruction.  The LDS image is the unpadded [32 rows][64 k] tile; the 16-B chunk c of row r
// sits at chunk position c ^ (r & 15) -- the permutation is applied to the per-lane SOURCE address (the DMA writes
// lane-linearly) and again when the fragments are read, which makes the ds_read_b128 fragment reads conflict-free.
// Double buffer, one barrier per K tile: [wait own pieces of tile t] [barrier] [issue tile t+1] [MFMA on tile t].
// Requires K % 4 == 0 and 16-B aligned rows (else the register-staged kernel runs).
template <int KS, int NBUF = 3>
__global__ __launch_bounds__(64 * KS, NBUF > 4 ? KS / 4 : (KS == 8 ? 2 : 4)) void gemm_dma_kernel(const dv_gemm_desc g, const LoadCfg lc) {
    constexpr int BM = 32, BN = 32, BK = 64, NT = 64 * KS, KW = BK / KS, KH = KW / 2;
    constexpr int TILE = 32 * BK, STAGE = 2 * TILE, RED = KS * BM * (BN + 1);
    constexpr int PIECES = 16 / KS;                     // 1-KiB pieces per wave per K tile (8 of A, then 8 of B)
    // NBUF - 1 K tiles in flight behind the one being multiplied (the loop is bound by the latency of its loads)
    static_assert((NBUF - 2) * PIECES <= 63, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(16))) float smem[NBUF * STAGE > RED ? NBUF * STAGE : RED];
    publish_on_entry(g);
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    int tm, tn;
    tile_of_block(blockIdx.x, gridDim.x, tiles_m, tiles_n, lc.map, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    // this wave's pieces: piece q in [0,16): operand = q / 8, rows 4*(q%8) .. +3
    const float* src[PIECES];
    int kofs[PIECES];          // k offset (floats) of this lane's chunk inside a K tile
    int dst[PIECES];           // float offset of the piece inside a stage
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int q = wave + i * KS, o = q >> 3, row = 4 * (q & 7) + (lane >> 4), p = lane & 15;
        const int c = p ^ (row & 15);
        int line = (o == 0 ? m0 : n0) + row;
        const int lim = o == 0 ? g.M : g.N;
        line = line < lim ? line : lim - 1;
        src[i] = (o == 0 ? g.A + (int64_t)line * g.lda : g.B + (int64_t)line * g.ldb) + c * 4;
        kofs[i] = c * 4;
        dst[i] = o * TILE + (q & 7) * 256;
    }
    auto issue = [&](int kt, float* stage) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const float* p = (kt * BK + kofs[i] < g.K) ? src[i] + (int64_t)kt * BK : dv_zero_chunk;
            __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) float*)(stage + dst[i]), 16, 0, 0);
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int kb = wave * KW + lh * KH;
    const int nkt = (g.K + BK - 1) / BK;
    // tiles past the end of K are issued too (they read the zero block into a buffer nobody multiplies): the number
    // of pieces in flight is then the same at every iteration and ONE counted wait serves the whole loop
#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t) issue(t, smem + t * STAGE);
    int bi = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        float* cur = smem + bi * STAGE;
        // this wave's pieces of tile kt have landed (the pieces of the NBUF - 2 tiles behind it may still be in
        // flight: counted wait); raw barrier -- __syncthreads() would drain the DMA queue (vmcnt(0)) -- then
        // everyone's pieces have landed and everyone is done reading tile kt-1, whose buffer the issue below overwrites
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * PIECES) : "memory");
        asm volatile("s_barrier" ::: "memory");
        int bn = bi + NBUF - 1;
        bn = bn >= NBUF ? bn - NBUF : bn;
        issue(kt + NBUF - 1, smem + bn * STAGE);
        bi = bi + 1 == NBUF ? 0 : bi + 1;
        float fa[KH], fb[KH];
#pragma unroll
        for (int s = 0; s < KH; s += 4) {
            const int pc = (((kb + s) >> 2) ^ (li & 15)) << 2;
            const float4 va = *reinterpret_cast<const float4*>(&cur[li * BK + pc]);
            const float4 vb = *reinterpret_cast<const float4*>(&cur[TILE + li * BK + pc]);
            fa[s] = va.x; fa[s + 1] = va.y; fa[s + 2] = va.z; fa[s + 3] = va.w;
            fb[s] = vb.x; fb[s + 1] = vb.y; fb[s + 2] = vb.z; fb[s + 3] = vb.w;
        }
#pragma unroll
        for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc, 0, 0, 0);
    }
    __syncthreads();
    float* red = smem;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * BM + (r & 3) + 8 * (r >> 2) + 4 * lh) * (BN + 1) + li] = acc[r];
    __syncthreads();
    constexpr int RPT = BM / (NT / BN);
    const int col = tid % BN, r0 = (tid / BN) * RPT;
    float a4[RPT];
#pragma unroll
    for (int e = 0; e < RPT; ++e) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < KS; ++w) v += red[(w * BM + r0 + e) * (BN + 1) + col];
        a4[e] = v;
    }
    const int rbase = m0 + r0;
    epi_store_col<RPT>(g, a4, n0 + col, [rbase](int r) { return rbase + r; });
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-private variant of the 32x32 K-split tiling (lab tiling 13; forward layout): each of the four waves owns a
// contiguous quarter of K and stages ITS OWN 16-deep slices of both tiles (global -> registers -> its private LDS
// region, used only to transpose the coalesced loads into MFMA fragments), so the K loop has no workgroup barrier
// at all -- the waves free-run, two slices of loads in flight each; one barrier for the final K-split reduction.
__global__ __launch_bounds__(256, 4) void gemm_wp_kernel(const dv_gemm_desc g, const LoadCfg lc) {
    constexpr int BM = 32, BN = 32, KS = 4, SK = 16, LD = SK + 4;       // slice depth, padded LDS row
    constexpr int OPER = 32 * LD, STAGE = 2 * OPER, RED = KS * BM * (BN + 1);
    __shared__ __attribute__((aligned(16))) float smem[KS * 2 * STAGE > RED ? KS * 2 * STAGE : RED];
    publish_on_entry(g);
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    int tm, tn;
    tile_of_block(blockIdx.x, gridDim.x, tiles_m, tiles_n, lc.map, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // this wave's K range: quarters rounded up to whole slices
    const int nsl = (g.K + SK - 1) / SK, per = (nsl + KS - 1) / KS;
    const int s0 = wave * per, s1 = (s0 + per < nsl) ? s0 + per : nsl;      // slices [s0, s1)
    float* my = smem + wave * 2 * STAGE;
    // staging: lane -> (row = lane/4 + 16 j, chunk = lane%4), j = 0,1, for A and for B.  Every load is unconditional:
    // the slice index is clamped into the wave's range (a re-read slice is never multiplied) and a chunk past K reads
    // the last valid chunk of its row and is multiplied by 0 (K % 4 == 0: a chunk is wholly inside or outside)
    const int r_ = lane >> 2, c_ = lane & 3;
    int ra0 = m0 + r_, ra1 = m0 + r_ + 16, rb0 = n0 + r_, rb1 = n0 + r_ + 16;
    ra0 = ra0 < g.M ? ra0 : g.M - 1;
    ra1 = ra1 < g.M ? ra1 : g.M - 1;
    rb0 = rb0 < g.N ? rb0 : g.N - 1;
    rb1 = rb1 < g.N ? rb1 : g.N - 1;
    const float* pa0 = g.A + (int64_t)ra0 * g.lda;
    const float* pa1 = g.A + (int64_t)ra1 * g.lda;
    const float* pb0 = g.B + (int64_t)rb0 * g.ldb;
    const float* pb1 = g.B + (int64_t)rb1 * g.ldb;
    const int slast = s1 - 1, klast = g.K - 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 a00, a01, b00, b01, a10, a11, b10, b11;      // two register sets (x: set, y: row half)
#define WP_FETCH(S, A0, A1, B0, B1)                                                        \
    {                                                                                      \
        const int ss = (S) < slast ? (S) : slast;                                          \
        const int k = ss * SK + c_ * 4, kc = k < klast ? k : klast;                        \
        const float mk = k < g.K ? 1.f : 0.f;                                              \
        A0 = *reinterpret_cast<const float4*>(pa0 + kc);                                   \
        A1 = *reinterpret_cast<const float4*>(pa1 + kc);                                   \
        B0 = *reinterpret_cast<const float4*>(pb0 + kc);                                   \
        B1 = *reinterpret_cast<const float4*>(pb1 + kc);                                   \
        A0.x *= mk; A0.y *= mk; A0.z *= mk; A0.w *= mk;                                    \
        A1.x *= mk; A1.y *= mk; A1.z *= mk; A1.w *= mk;                                    \
    }
#define WP_STASH(ST, A0, A1, B0, B1)                                                       \
    {                                                                                      \
        *reinterpret_cast<float4*>(&(ST)[r_ * LD + c_ * 4]) = A0;                          \
        *reinterpret_cast<float4*>(&(ST)[(r_ + 16) * LD + c_ * 4]) = A1;                   \
        *reinterpret_cast<float4*>(&(ST)[OPER + r_ * LD + c_ * 4]) = B0;                   \
        *reinterpret_cast<float4*>(&(ST)[OPER + (r_ + 16) * LD + c_ * 4]) = B1;            \
    }
#define WP_MMA(ST)                                                                         \
    {                                                                                      \
        const float4 x0 = *reinterpret_cast<const float4*>(&(ST)[li * LD + lh * 8]);       \
        const float4 x1 = *reinterpret_cast<const float4*>(&(ST)[li * LD + lh * 8 + 4]);   \
        const float4 y0 = *reinterpret_cast<const float4*>(&(ST)[OPER + li * LD + lh * 8]);     \
        const float4 y1 = *reinterpret_cast<const float4*>(&(ST)[OPER + li * LD + lh * 8 + 4]); \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.x, y0.x, acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, y0.y, acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.z, y0.z, acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.w, y0.w, acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.x, y1.x, acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.y, y1.y, acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.z, y1.z, acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.w, y1.w, acc, 0, 0, 0);              \
    }
    if (s0 < s1) {
        float* st0 = my;
        float* st1 = my + STAGE;
        WP_FETCH(s0, a00, a01, b00, b01);
        WP_FETCH(s0 + 1, a10, a11, b10, b11);
        WP_STASH(st0, a00, a01, b00, b01);
        WP_FETCH(s0 + 2, a00, a01, b00, b01);
        const int n = s1 - s0;
        int s = s0;
        for (int it = 0; it < n / 2; ++it, s += 2) {
            WP_STASH(st1, a10, a11, b10, b11);
            WP_FETCH(s + 3, a10, a11, b10, b11);
            WP_MMA(st0);
            WP_STASH(st0, a00, a01, b00, b01);
            WP_FETCH(s + 4, a00, a01, b00, b01);
            WP_MMA(st1);
        }
        if (n & 1) WP_MMA(st0);
    }
#undef WP_FETCH
#undef WP_STASH
#undef WP_MMA
    __syncthreads();
    float* red = smem;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * BM + (r & 3) + 8 * (r >> 2) + 4 * lh) * (BN + 1) + li] = acc[r];
    __syncthreads();
    constexpr int RPT = BM / (256 / BN);
    const int col = tid % BN, r0 = (tid / BN) * RPT;
    float a4[RPT];
#pragma unroll
    for (int e = 0; e < RPT; ++e) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < KS; ++w) v += red[(w * BM + r0 + e) * (BN + 1) + col];
        a4[e] = v;
    }
    const int rbase = m0 + r0;
    epi_store_col<RPT>(g, a4, n0 + col, [rbase](int r) { return rbase + r; });
}

#endif   // DV_LAB

template <int BM, int BN, int BK, int KS>
__global__ __launch_bounds__(64 * KS, KS == 8 ? 2 : (BK == 32 ? DV_DENSE_WG : 4)) void gemm_heads_kernel(const dv_gemm_desc g, const LoadCfg lc,
                                                               const dv_heads_epi he) {
    __shared__ __attribute__((aligned(16))) float smem[gemm_smem_floats<BM, BN, BK, KS, true, true>()];
    publish_on_entry(g);
    gemm_body<BM, BN, BK, 1, 1, KS, true, true, true>(g, lc, smem, blockIdx.x, gridDim.x, &he);
}

// Two independent products in ONE launch (workgroups [0,tiles1) run the first, the rest the
// second): the weight-gradient dW = dy^T x and the data-gradient dx = dy W of a layer both only
// need dy, so they share a launch slot instead of paying the per-launch latency chain twice.
template <int BM, int BN, int BK, int WM, int WN, int KS, bool A1, bool B1, bool A2, bool B2>
__global__ __launch_bounds__(256, BK == 32 ? DV_DENSE_WG : 4) void gemm_pair_kernel(const dv_gemm_desc g1, const LoadCfg lc1, const dv_gemm_desc g2,
                                                        const LoadCfg lc2, int tiles1) {
    constexpr int S1 = gemm_smem_floats<BM, BN, BK, KS, A1, B1>(), S2 = gemm_smem_floats<BM, BN, BK, KS, A2, B2>();
    __shared__ __attribute__((aligned(16))) float smem[S1 > S2 ? S1 : S2];
    publish_on_entry(g1);
    if ((int)blockIdx.x < tiles1)
        gemm_body<BM, BN, BK, WM, WN, KS, A1, B1>(g1, lc1, smem, blockIdx.x, tiles1);
    else
        gemm_body<BM, BN, BK, WM, WN, KS, A2, B2>(g2, lc2, smem, blockIdx.x - tiles1, gridDim.x - tiles1);
}

// ---- the same three launch shapes with the hand-pipelined K loop (gemm_pipe.inc): PS ring slots, WPS workgroups' worth
// of waves per SIMD the register allocation is held to
template <int BM, int BN, int BK, int KS, bool AKC, bool BKC, int PS, int WPS>
__global__ __launch_bounds__(64 * KS * (BM / 32) * (BN / 32), WPS) void gemm_kpipe_kernel(const dv_gemm_desc g, const LoadCfg lc) {
    __shared__ __attribute__((aligned(1024))) float smem[gemm_smem_floats<BM, BN, BK, KS, AKC, BKC, PS>()];
    publish_on_entry(g);
    gemm_body<BM, BN, BK, BM / 32, BN / 32, KS, AKC, BKC, false, false, PS>(g, lc, smem, blockIdx.x, gridDim.x);
}

template <int BM, int BN, int BK, int KS, int PS, int WPS>
__global__ __launch_bounds__(64 * KS, WPS) void gemm_heads_pipe_kernel(const dv_gemm_desc g, const LoadCfg lc, const dv_heads_epi he) {
    __shared__ __attribute__((aligned(1024))) float smem[gemm_smem_floats<BM, BN, BK, KS, true, true, PS>()];
    publish_on_entry(g);
    gemm_body<BM, BN, BK, 1, 1, KS, true, true, true, false, PS>(g, lc, smem, blockIdx.x, gridDim.x, &he);
}

template <int BM, int BN, int BK, int KS, bool A1, bool B1, bool A2, bool B2, int PS, int WPS>
__global__ __launch_bounds__(256, WPS) void gemm_pair_pipe_kernel(const dv_gemm_desc g1, const LoadCfg lc1, const dv_gemm_desc g2,
                                                                    const LoadCfg lc2, int tiles1) {
    constexpr int S1 = gemm_smem_floats<BM, BN, BK, KS, A1, B1, PS>(), S2 = gemm_smem_floats<BM, BN, BK, KS, A2, B2, PS>();
    __shared__ __attribute__((aligned(1024))) float smem[S1 > S2 ? S1 : S2];
    publish_on_entry(g1);
    if ((int)blockIdx.x < tiles1)
        gemm_body<BM, BN, BK, 1, 1, KS, A1, B1, false, false, PS>(g1, lc1, smem, blockIdx.x, tiles1);
    else
        gemm_body<BM, BN, BK, 1, 1, KS, A2, B2, false, false, PS>(g2, lc2, smem, blockIdx.x - tiles1, gridDim.x - tiles1);
}

inline int vec_width(const void* p, int64_t ld) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    if ((a & 15) == 0 && (ld & 3) == 0) return 4;
    if ((a & 7) == 0 && (ld & 1) == 0) return 2;
    return 1;
}

// tuning (dv_gemm_set_option(1, bytes)): extra dynamic LDS per workgroup of the 32x32-tile launches whose grids
// fill the chip -- caps how many of them a CU holds, which leaves wave slots / registers on EVERY CU for the other
// launch chain's small kernels (an alternative to the CU partition of the two chains)
// the caller's tuning block (dv_gemm_desc.tune; NULL = the defaults): read at launch time, per call -- the library keeps
// no tuning state of its own
const dv_gemm_tune k_default_tune = {0, {-1, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
inline const dv_gemm_tune& tune_of(const dv_gemm_desc& g) { return g.tune != nullptr ? *g.tune : k_default_tune; }
inline int lds_pad(const dv_gemm_desc& g, int bm, int tiles) { return (bm <= 32 && tiles >= 512) ? tune_of(g).opt[1] : 0; }

// output columns incl. the ones column of the fused bias gradient (see gemm_body)
inline int cols_eff(const dv_gemm_desc& g) { return g.N + ((g.flags & DV_FLAG_ONES_COL) ? DV_ONES_OFF(g) + 1 : 0); }

template <int BM, int BN, int BK, int WM, int WN, int KS, bool MI16 = false>
int launch_cfg(const dv_gemm_desc& g, const LoadCfg& lc, hipStream_t st) {
    const int tiles = ((g.M + BM - 1) / BM) * ((cols_eff(g) + BN - 1) / BN);
    dim3 grid(tiles), block(64 * WM * WN * KS);
    if (g.a_kcontig && g.b_kcontig)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, WM, WN, KS, true, true, MI16>), grid, block, lds_pad(g, BM, tiles), st, g, lc);
    else if (g.a_kcontig && !g.b_kcontig)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, WM, WN, KS, true, false, MI16>), grid, block, lds_pad(g, BM, tiles), st, g, lc);
    else if (!g.a_kcontig && !g.b_kcontig)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, WM, WN, KS, false, false, MI16>), grid, block, lds_pad(g, BM, tiles), st, g, lc);
    else
        return DV_ERR_UNSUPPORTED;
    DV_RETURN_LAUNCH();
}

// the bias-gradient column of the LDS-DMA kernels: the first multiple of 4 at or past N (see DV_ONES_OFF)
inline void pipe_ones_off(dv_gemm_desc& g) {
    if (g.flags & DV_FLAG_ONES_COL) g.flags = (g.flags & ~0x300) | (((4 - (g.N & 3)) & 3) << 8);
}

template <int BM, int BN, int BK, int KS, int PS, int WPS>
int launch_kpipe(const dv_gemm_desc& g_in, const LoadCfg& lc, hipStream_t st) {
    dv_gemm_desc g = g_in;
    pipe_ones_off(g);
    const int tiles = ((g.M + BM - 1) / BM) * ((cols_eff(g) + BN - 1) / BN);
    dim3 grid(tiles), block(64 * KS * (BM / 32) * (BN / 32));
    if (g.a_kcontig && g.b_kcontig)
        hipLaunchKernelGGL((gemm_kpipe_kernel<BM, BN, BK, KS, true, true, PS, WPS>), grid, block, 0, st, g, lc);
    else if (g.a_kcontig && !g.b_kcontig)
        hipLaunchKernelGGL((gemm_kpipe_kernel<BM, BN, BK, KS, true, false, PS, WPS>), grid, block, 0, st, g, lc);
    else if (!g.a_kcontig && !g.b_kcontig)
        hipLaunchKernelGGL((gemm_kpipe_kernel<BM, BN, BK, KS, false, false, PS, WPS>), grid, block, 0, st, g, lc);
    else
        return DV_ERR_UNSUPPORTED;
    DV_RETURN_LAUNCH();
}

}  // namespace

// which tilings this build of the library carries (tests / tuning: dv_gemm_tune.tiling); the lab tilings exist in the
// tuning build only (-DDV_LAB)
extern "C" int dv_gemm_has_tiling(int t) {
#ifndef DV_LAB
    return ((t >= 0 && t <= 3) || t == 17 || t == 40 || t == 46) ? 1 : 0;
#else
    return t >= 0 ? 1 : 0;
#endif
}

#if DV_STAMP
extern "C" int dv_gemm_stamp_buffer(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(dv_stamp_buf), &p, sizeof(p)) == hipSuccess ? DV_OK : DV_ERR_LAUNCH;
}
#endif

static int gemm_prepare(const dv_gemm_desc* d, LoadCfg& lc, int& tiling) {
    DV_REQUIRE(d != nullptr);
    const dv_gemm_desc& g = *d;
    DV_REQUIRE(g.M >= 0 && g.N >= 0 && g.K >= 1);
    if (g.M == 0 || g.N == 0) {
        tiling = -1;
        return DV_OK;
    }
    DV_REQUIRE(g.A && g.B && g.C);
    DV_REQUIRE(g.epilogue == DV_EPI_PLAIN || g.epilogue == DV_EPI_FWD || g.epilogue == DV_EPI_BWD || g.epilogue == DV_EPI_KLQ);
    DV_REQUIRE(g.epilogue != DV_EPI_BWD || g.yref != nullptr);
    DV_REQUIRE(g.epilogue != DV_EPI_KLQ || (g.yref && g.resid && g.bias && g.scale && g.split > 0 && g.split <= g.N &&
                                            g.ldc >= 2 * (int64_t)g.split && g.beta == 0.f && g.a_colsum == nullptr));
    DV_REQUIRE(g.A2 == nullptr || (g.a_kcontig && g.K1 >= 0 && g.K1 <= g.K));
    DV_REQUIRE(g.a_kscale == nullptr || g.a_kcontig);
    DV_REQUIRE(g.a_colsum == nullptr || !g.a_kcontig);
    DV_REQUIRE(!( !g.a_kcontig && g.b_kcontig));
    DV_REQUIRE(g.pub_flag == nullptr || g.pub_ctr != nullptr);
    lc.vecA = vec_width(g.A, g.lda);
    if (g.A2) {
        const int v2 = vec_width(g.A2, g.lda2);
        lc.vecA = v2 < lc.vecA ? v2 : lc.vecA;
        if (g.K1 & 3) lc.vecA = 1;   // a 4-chunk could straddle the two sources
    }
    lc.vecB = vec_width(g.B, g.ldb);
    lc.vecA_t = lc.vecA;
    lc.vecB_t = lc.vecB;
    while (g.K % lc.vecA_t) lc.vecA_t >>= 1;
    while (g.K % lc.vecB_t) lc.vecB_t >>= 1;
    const int64_t t64 = (int64_t)((g.M + 63) / 64) * ((g.N + 63) / 64);
    const int64_t t128 = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128);
    const dv_gemm_tune& T = tune_of(g);
    if (!dv_gemm_has_tiling(T.tiling)) return DV_ERR_UNSUPPORTED;
    tiling = T.tiling;
    // measured on MI355X (tools/gemm_bench.py): below ~4 workgroups per CU the 32x32 K-split
    // tiling wins (more resident workgroups hide the per-K-tile latency chain); the larger
    // tiles only pay once their grids alone fill the chip several times over
    const int64_t t64_min = T.opt[3] > 0 ? T.opt[3] : 1024;
    // ... or once K is long enough to amortise a tile's prologue over many K steps: 1536 x 2048 x 20000 (encoder L1 of
    // the wide configuration, 768 tiles of 64x64) runs 956 us on the 64x64 tiling, 1090 us on 32x32, 1336 us on 128x128
    // (whole-set evaluation, tools/eval_gemm_bench.py) the 128-class tiles only where N fills its 256-wide tiles to 7/8
    // (N = 600 / 800: 78 % -- 12288 x 800 x 978 runs 265 us there, 184-188 on the smaller tiles; 32768 x 600 x 100:
    // 122 -> 81 us); below them K <= 256 -- a product that is all prologue and epilogue -- takes the 32x32 tiling
    // whatever the grid (24576 x 200 x 102: 36 -> 27 us)
    const int n256 = (g.N + 255) / 256 * 256;
    const bool n_fills = (int64_t)g.N * 8 >= (int64_t)n256 * 7;
    if (tiling == 0)
        tiling = (t128 >= 1024 && n_fills) ? 3 : ((g.K > 256 && (t64 >= t64_min || (t64 >= 512 && g.K >= 8192))) ? 1 : 2);
    // the chip-filling products run on the hand-pipelined LDS-DMA tiling (gemm_pipe.inc) when their operands allow it
    // (16-B aligned rows, K % 4 == 0 ...): 8192 x 8192 x 2048 134 -> 144-149 TFLOP/s (dv_gemm_set_option(3, -1): off)
    if (tiling == 3 && T.tiling == 0 && T.opt[3] != -1 && pipe_ok(g, lc)) tiling = 40;
    // (64x64 tiles, three workgroups per CU: one resident round up to 768 tiles; measured 1536 x 2048 x 20000: 983 -> 962 us,
    // but 2048 x 2048 x 4096 = 1024 tiles: 271 -> 323 us)
    if (tiling == 1 && T.tiling == 0 && T.opt[3] != -1 && g.K >= 1024 && t64 <= 768 && pipe_ok(g, lc)) tiling = 46;
    // workgroup -> tile map: XCD chunk-major for the small grids (each XCD keeps a compact band of the
    // output, its panels stay in its L2); for the grids that fill the chip many times over, bands of 16 tile
    // rows swept column by column (measured, wide configuration: chunk-major 121.5, linear 127.0, bands of 16
    // 128.3 TF/s over the step's products; no difference at the cfg-2 sizes)
    lc.map = T.opt[0] >= 0 ? T.opt[0] : (tiling == 3 ? 16 : (tiling == 40 ? 32 : 1));
    return DV_OK;
}

// tiles of 32x32 from which a product runs on the high-occupancy tiling (dv_gemm_set_option(8, n); 0 = default)
static int dense_min_tiles(const dv_gemm_desc& g) { return tune_of(g).opt[8] > 0 ? tune_of(g).opt[8] : 512; }
static bool pipe_on(const dv_gemm_desc& g) { return tune_of(g).opt[3] != -1; }

// fused bias gradient of a dy^T x product: the ones column (DV_FLAG_ONES_COL) -- free where the last column tile has
// padding, one extra tile per row panel where N is a multiple of the tile width.  Not in the 128x128 tiling (its
// kernels sit at the 256-register limit; an extra tile would cost 1/(N/128) of the product): there the column sums
// are a launch of their own, noise next to a product of that size
static int colsum_setup(dv_gemm_desc& g, int tiling, hipStream_t st) {
    g.flags &= 3;
    if (g.a_colsum == nullptr || g.a_kcontig) return DV_OK;
    if (tiling == 3 || tiling == 16 || (tiling >= 40 && tiling < 50)) {
        const int rc = dv_colsum(g.A, g.lda, g.K, g.M, g.a_colsum, g.colsum_beta, st);
        g.a_colsum = nullptr;
        return rc;
    }
    g.flags |= DV_FLAG_ONES_COL;
    return DV_OK;
}

// The 128 x 256 pipe kernel runs two workgroups per CU.  OPT-IN (dv_gemm_tune.opt[5] = percent of the estimate below; 0 =
// off, the default): under a plain epilogue (stores only) the second resident workgroup of the first dispatch round starts
// half a tile late (gemm_pipe_kernel), so that one workgroup's epilogue and prologue run under the other's K loop.  Half a
// tile of two co-resident workgroups = one tile's matrix time alone: BM x BN x K x 2 flop at 0.6145 TFLOP/s per CU; in
// ticks of 10 ns.  Measured (tools/stagger_probe.py, launches back to back): 32768 x 1956 x 600 627 -> 581 us, 2048 x 20000 x
// 1536 (2.47 rounds of the chip's 512 slots) 1029 -> 872 us, nothing from K = 2048 on, +1.5 % on a product of exactly one
// round -- and NOTHING inside the whole-set evaluation's graph (the same 32768-row product: 707 -> 716 us in situ,
// evaluation 2.67-2.73 ms either way): the launches around it de-phase the workgroups already.  Hence not a default.
static int pipe_stagger_ticks(const dv_gemm_desc& g, int bm, int bn) {
    const int o5 = tune_of(g).opt[5];
    const int64_t tiles = (int64_t)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn);
    if (o5 <= 0 || tiles < 768 || g.K > 2048 || g.epilogue != DV_EPI_PLAIN) return 0;
    const double us = 2.0 * bm * bn * (double)g.K / 0.6145e6;
    return (int)(us * o5);
}

static int gemm_launch(const dv_gemm_desc& g_in, const LoadCfg& lc, int tiling, hipStream_t st) {
    if (tiling < 0) return DV_OK;
    dv_gemm_desc g = g_in;
    {
        const int rc = colsum_setup(g, tiling, st);
        if (rc != DV_OK) return rc;
    }
    if (tiling == 3) return launch_cfg<128, 128, 32, 2, 2, 1>(g, lc, st);
    if (tiling >= 40 && tiling < 50 && !pipe_ok(g, lc)) return launch_cfg<128, 128, 32, 2, 2, 1>(g, lc, st);
    if (tiling == 40) {
        LoadCfg l2 = lc;
        l2.stagger = pipe_stagger_ticks(g, 128, 256);
        return launch_pipe<128, 256, 16, 2, 2, 3, 2>(g, l2, st);
    }
    // few output tiles, long K (encoder layer 1 of the wide configuration: 1536 x 2048 x 20000 = 768 tiles of 64 x 64 =
    // three per CU, all resident): the 64x64 tiling's pipelined form
    if (tiling == 46 && pipe_ok(g, lc)) return launch_pipe<64, 64, 32, 2, 2, 3, 3>(g, lc, st);
    if (tiling == 46) return launch_cfg<64, 64, 32, 2, 2, 1>(g, lc, st);
#ifdef DV_LAB
    // 64-row / 64-column tiles of the K-split family: half / two thirds of the L2 -> LDS bytes per flop of the 32 x 32 tile
    if (tiling == 60 && pipe_ok(g, lc)) return launch_kpipe<64, 32, 32, 2, 3, 4>(g, lc, st);
    if (tiling == 61 && pipe_ok(g, lc)) return launch_kpipe<64, 32, 32, 2, 2, 6>(g, lc, st);
    if (tiling == 62 && pipe_ok(g, lc)) return launch_kpipe<64, 32, 64, 4, 2, 3>(g, lc, st);
    if (tiling == 63 && pipe_ok(g, lc)) return launch_kpipe<32, 64, 32, 2, 3, 4>(g, lc, st);
    if (tiling == 64 && pipe_ok(g, lc)) return launch_kpipe<64, 64, 32, 2, 3, 2>(g, lc, st);
    if (tiling == 41) return launch_pipe<128, 128, 16, 2, 2, 3, 3>(g, lc, st);
    if (tiling == 42) return launch_pipe<128, 256, 16, 2, 2, 4, 1>(g, lc, st);
    if (tiling == 43) return launch_pipe<256, 128, 16, 2, 2, 3, 2>(g, lc, st);
    if (tiling == 44) return launch_pipe<128, 128, 32, 2, 2, 2, 2>(g, lc, st);
#endif
#ifdef DV_LAB
    if (tiling == 16) return launch_cfg<128, 128, 32, 2, 2, 1, true>(g, lc, st);   // the 128x128 tiling on v_mfma_f32_16x16x4_f32
    if (tiling == 30 && g.a_kcontig && g.b_kcontig) return launch_cfg<96, 64, 64, 1, 2, 4>(g, lc, st);   // lab: one round of big tiles
    if (tiling == 32 && g.a_kcontig && g.b_kcontig) return launch_cfg<96, 64, 32, 1, 2, 2>(g, lc, st);
    if (tiling == 34 && g.a_kcontig) return launch_cfg<64, 32, 64, 2, 1, 4>(g, lc, st);
    if (tiling == 4) return launch_cfg<32, 32, 128, 1, 1, 4>(g, lc, st);
    if (tiling == 5) return launch_cfg<64, 64, 64, 2, 2, 2>(g, lc, st);   // 8 waves: 2 per SIMD
    if (tiling == 6) return launch_cfg<64, 32, 64, 2, 1, 2>(g, lc, st);   // 2 row blocks x 2-way K split
    if (tiling == 7) return launch_cfg<32, 64, 64, 1, 2, 2>(g, lc, st);   // 2 column blocks x 2-way K split
    if (tiling == 8) return launch_cfg<32, 32, 128, 1, 1, 8>(g, lc, st);   // 8-way K split: twice the waves, half the chain
    if (tiling == 9) return launch_cfg<32, 32, 64, 1, 1, 8>(g, lc, st);
    if (tiling == 10) return launch_cfg<32, 32, 128, 1, 1, 16>(g, lc, st);
    if (tiling == 13) {                   // wave-private staging, no barrier in the K loop (same restrictions)
        const bool ok = g.a_kcontig && g.b_kcontig && g.A2 == nullptr && g.a_kscale == nullptr && (g.K & 3) == 0 &&
                        lc.vecA == 4 && lc.vecB == 4;
        if (ok) {
            const int tiles = ((g.M + 31) / 32) * ((g.N + 31) / 32);
            hipLaunchKernelGGL(gemm_wp_kernel, dim3(tiles), dim3(256), 0, st, g, lc);
            DV_RETURN_LAUNCH();
        }
        return launch_cfg<32, 32, 64, 1, 1, 8>(g, lc, st);
    }
    if (tiling == 11 || tiling == 12 || (tiling >= 21 && tiling <= 24)) {   // LDS-DMA staging (forward layout, 16-B aligned rows, K % 4 == 0)
        const bool ok = g.a_kcontig && g.b_kcontig && g.A2 == nullptr && g.a_kscale == nullptr && (g.K & 3) == 0 &&
                        lc.vecA == 4 && lc.vecB == 4;
        if (ok) {
            const int tiles = ((g.M + 31) / 32) * ((g.N + 31) / 32);
            if (tiling == 11)
                hipLaunchKernelGGL((gemm_dma_kernel<8>), dim3(tiles), dim3(512), 0, st, g, lc);
            else if (tiling == 12)
                hipLaunchKernelGGL((gemm_dma_kernel<4>), dim3(tiles), dim3(256), 0, st, g, lc);
            else if (tiling == 21)
                hipLaunchKernelGGL((gemm_dma_kernel<8, 5>), dim3(tiles), dim3(512), 0, st, g, lc);
            else if (tiling == 22)
                hipLaunchKernelGGL((gemm_dma_kernel<8, 9>), dim3(tiles), dim3(512), 0, st, g, lc);
            else if (tiling == 23)
                hipLaunchKernelGGL((gemm_dma_kernel<4, 5>), dim3(tiles), dim3(256), 0, st, g, lc);
            else
                hipLaunchKernelGGL((gemm_dma_kernel<4, 9>), dim3(tiles), dim3(256), 0, st, g, lc);
            DV_RETURN_LAUNCH();
        }
        return launch_cfg<32, 32, 64, 1, 1, 8>(g, lc, st);
    }
#endif   // DV_LAB
    if (tiling == 17) return launch_cfg<32, 32, 32, 1, 1, 4>(g, lc, st);    // forced: the seven-per-CU tiling (below)
    if (tiling != 1 && tiling != 2) return DV_ERR_UNSUPPORTED;
    if (tiling == 1) return launch_cfg<64, 64, 32, 2, 2, 1>(g, lc, st);
    // 32x32 K-split tiling: the k-contiguous-A layouts (x W^T, dy W) run with EIGHT waves splitting each 64-deep K
    // tile (half the MFMA chain per wave, twice the waves to overlap its latency: 4-10 % faster on every cfg-2
    // product of these layouts, tools/gemm_bench.py --tilings 2,9); dy^T x keeps four (its big products lose with eight)
    // grids that fill the chip (>= dense_min_tiles() tiles of 32x32): half the K tile and <= 72 registers, so that SEVEN
    // workgroups fit a CU instead of four and the whole grid is resident in one round -- the 1178 workgroups of the
    // decoder-head products took two rounds on 256 CUs (three on the main chain's 192): x W^T 24.3 -> 21.2 us,
    // dy^T x 23.6 -> 21.0 us alone (tools/gemm_bench.py --tilings 2,9,17)
    const int tiles32 = ((g.M + 31) / 32) * ((cols_eff(g) + 31) / 32);
    // each of the three on the hand-pipelined LDS-DMA loop where the operands allow it (pipe_ok; dv_gemm_set_option(3, -1): never)
    const bool pipe = pipe_on(g) && pipe_ok(g, lc);
    if (tiles32 >= dense_min_tiles(g)) return pipe ? launch_kpipe<32, 32, 32, 4, 2, DV_DENSE_WG>(g, lc, st) : launch_cfg<32, 32, 32, 1, 1, 4>(g, lc, st);
    if (g.a_kcontig && tune_of(g).opt[7] == 0) return pipe ? launch_kpipe<32, 32, 64, 8, 2, 2>(g, lc, st) : launch_cfg<32, 32, 64, 1, 1, 8>(g, lc, st);
    return pipe ? launch_kpipe<32, 32, 64, 4, 2, 4>(g, lc, st) : launch_cfg<32, 32, 64, 1, 1, 4>(g, lc, st);
}

// A chip-filling product on the 128x256 tiling whose N is no multiple of 256: the ragged last column of tiles costs
// every CU a whole tile time whenever it pushes the tiles per CU over an integer -- 8192 x 40000 x 2048 (decoder heads of
// the wide configuration, 64 x 157 = 10048 tiles = 39.25 per CU): 9129 us, exactly the 9106 us of N = 40960 (40 per CU),
// against 8878 us for N = 39936 (39 per CU); time is linear in tiles per CU, rounded UP (measured, round 5).  A plain
// product then runs as two launches: columns [0, N - N % 256) on the big tiles, the rest (< 256 columns) on the
// small-tile family (8192 x 64 x 2048: ~30 us) -- fused epilogues included (bias / scale / the two heads' split / residual /
// activation backward move with their columns; whole-set evaluation: 32768 x 1956 x 600 with the heads' epilogue, 8 -> 7
// tiles per CU).  Only where it removes a tile per CU.
static bool ragged_n_split_pays(const dv_gemm_desc& g, int tiling) {
    if (tiling != 40 || tune_of(g).tiling != 0 || tune_of(g).opt[6] == -1) return false;
    const int n_tail = g.N % 256, n_main = g.N - n_tail;
    if (n_tail == 0 || n_main == 0) return false;
    const int64_t rows_t = (g.M + 127) / 128, kCUs = 256;
    const int64_t full = rows_t * ((g.N + 255) / 256), main = rows_t * (n_main / 256);
    return (full + kCUs - 1) / kCUs > (main + kCUs - 1) / kCUs;
}

extern "C" int dv_gemm(const dv_gemm_desc* d, dv_stream_t stream) {
    LoadCfg lc;
    int tiling = 0;
    const int rc = gemm_prepare(d, lc, tiling);
    if (rc != DV_OK) return rc;
    if (ragged_n_split_pays(*d, tiling)) {
        // columns [0, n_main) | [n_main, N): every per-column operand of the epilogue moves with its columns
        const int n_tail = d->N % 256, n_main = d->N - n_tail;
        dv_gemm_desc g1 = *d, g2 = *d;
        g1.N = n_main;
        g2.N = n_tail;
        g2.B = d->B + (d->b_kcontig ? (int64_t)n_main * d->ldb : (int64_t)n_main);
        g2.C = d->C + n_main;
        g1.split = d->split < n_main ? d->split : n_main;
        g2.split = d->split > n_main ? d->split - n_main : 0;
        if (d->scale) g2.scale = d->scale + n_main;
        if (d->bias) g2.bias = d->bias + n_main;
        if (d->yref) g2.yref = d->yref + n_main;
        if (d->resid) {
            g1.resid_cols = d->resid_cols < n_main ? d->resid_cols : n_main;
            g2.resid_cols = d->resid_cols > n_main ? d->resid_cols - n_main : 0;
            g2.resid = g2.resid_cols > 0 ? d->resid + n_main : nullptr;
        }
        g2.a_colsum = nullptr;                  // (column sums of A: once, with the first part)
        g2.pub_flag = nullptr;                  // (the launch publishes once, on entry of the first)
        g2.tune = nullptr;
        LoadCfg l1, l2;
        int t1 = 0, t2 = 0;
        int r = gemm_prepare(&g1, l1, t1);
        if (r != DV_OK) return r;
        r = gemm_prepare(&g2, l2, t2);
        if (r != DV_OK) return r;
        r = gemm_launch(g1, l1, t1, static_cast<hipStream_t>(stream));
        if (r != DV_OK) return r;
        return gemm_launch(g2, l2, t2, static_cast<hipStream_t>(stream));
    }
    return gemm_launch(*d, lc, tiling, static_cast<hipStream_t>(stream));
}

// half-tile width of the paired-heads launch: 16 (default: 32x(16+16) tiles = the workgroup count and MFMA chain of
// the 32x32 K-split tiling) or 32 (tune_of(g).opt[4] = 1 / 2: 32x(32+32) tiles with 4 / 8 waves; measured slower at cfg 2)
#ifdef DV_LAB
static int heads_hb() { return 16; }       // (lab: the 32-wide half tiles are chosen per call, dv_gemm_tune.opt[4])
#else
static int heads_hb() { return 16; }
#endif
extern "C" int dv_gemm_heads_tiles(int32_t split) { return split > 0 ? (split + heads_hb() - 1) / heads_hb() : 0; }

extern "C" int dv_gemm_heads(const dv_gemm_desc* d, const dv_heads_epi* e, dv_stream_t stream) {
    LoadCfg lc;
    int tiling = 0;
    DV_REQUIRE(d != nullptr && e != nullptr);
    const int rc = gemm_prepare(d, lc, tiling);
    if (rc != DV_OK) return rc;
    if (tiling < 0) return DV_OK;
    const dv_gemm_desc& g = *d;
    DV_REQUIRE(g.a_kcontig && g.b_kcontig && g.epilogue == DV_EPI_FWD && g.beta == 0.f && g.a_colsum == nullptr);
    DV_REQUIRE(g.split > 0 && g.N == 2 * g.split && g.resid_cols <= g.split);
    DV_REQUIRE(e->mode == DV_HEADS_SAMPLE || e->mode == DV_HEADS_NLL);
    if (e->mode == DV_HEADS_SAMPLE) {
        DV_REQUIRE(e->eps && e->out && e->n_src >= 0 && e->n_src <= g.M);
        DV_REQUIRE((e->seg_ptr == nullptr) == (e->seg_rows == nullptr));
        DV_REQUIRE(e->out2 == nullptr || e->sub != nullptr);
        DV_REQUIRE(e->out3 == nullptr || e->out3_idx != nullptr);
        DV_REQUIRE(e->out4 == nullptr || e->out4_ptr != nullptr);
    } else {
        DV_REQUIRE(e->x && e->coef && e->part);
    }
    lc.map = tune_of(g).opt[0] >= 0 ? tune_of(g).opt[0] : 1;
    const int tiles = ((g.M + 31) / 32) * dv_gemm_heads_tiles(g.split);
#ifdef DV_LAB
    if (tune_of(g).opt[4] == 1) {
        hipLaunchKernelGGL((gemm_heads_kernel<32, 64, 64, 4>), dim3(tiles), dim3(256), 0,
                           static_cast<hipStream_t>(stream), g, lc, *e);
        DV_RETURN_LAUNCH();
    }
    if (tune_of(g).opt[4] == 2) {   // 8 waves split K: per wave the MFMA chain of the 32x32 K-split tiling, A staged once
        hipLaunchKernelGGL((gemm_heads_kernel<32, 64, 64, 8>), dim3(tiles), dim3(512), 0,
                           static_cast<hipStream_t>(stream), g, lc, *e);
        DV_RETURN_LAUNCH();
    }
    if (tune_of(g).opt[4] == -1) {
        hipLaunchKernelGGL((gemm_heads_kernel<32, 32, 64, 4>), dim3(tiles), dim3(256), 0,
                           static_cast<hipStream_t>(stream), g, lc, *e);
        DV_RETURN_LAUNCH();
    }
#endif
    const bool pipe = pipe_on(g) && pipe_ok(g, lc);
    if (tune_of(g).opt[4] == 3 || (tune_of(g).opt[4] == 0 && tiles >= dense_min_tiles(g))) {
        // chip-filling grids: four waves, half the K tile, seven workgroups per CU (see gemm_launch): 29.1 -> 26.4 us
        // for the decoder heads + NLL alone
        if (pipe)
            hipLaunchKernelGGL((gemm_heads_pipe_kernel<32, 32, 32, 4, 2, DV_DENSE_WG>), dim3(tiles), dim3(256), 0,
                               static_cast<hipStream_t>(stream), g, lc, *e);
        else
            hipLaunchKernelGGL((gemm_heads_kernel<32, 32, 32, 4>), dim3(tiles), dim3(256), 0,
                               static_cast<hipStream_t>(stream), g, lc, *e);
    } else if (pipe) {
        hipLaunchKernelGGL((gemm_heads_pipe_kernel<32, 32, 64, 8, 2, 2>), dim3(tiles), dim3(512), 0,
                           static_cast<hipStream_t>(stream), g, lc, *e);
    } else {  // default: the 32x32 tiling's eight-wave K split, B lines = 16 + 16 rows of the two heads
        hipLaunchKernelGGL((gemm_heads_kernel<32, 32, 64, 8>), dim3(tiles), dim3(512), lds_pad(g, 32, tiles),
                           static_cast<hipStream_t>(stream), g, lc, *e);
    }
    DV_RETURN_LAUNCH();
}

extern "C" int dv_gemm_pair(const dv_gemm_desc* d1, const dv_gemm_desc* d2, dv_stream_t stream) {
    LoadCfg lc1, lc2;
    int t1 = 0, t2 = 0;
    int rc = gemm_prepare(d1, lc1, t1);
    if (rc != DV_OK) return rc;
    rc = gemm_prepare(d2, lc2, t2);
    if (rc != DV_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // a paired launch publishes once, on entry, for its first descriptor: at most one of the two may carry a publish
    // (it is moved to whichever runs first); two would silently drop one and leave its consumer chain spinning
    DV_REQUIRE(!(d1->pub_flag != nullptr && d2->pub_flag != nullptr));
    // fused form: both products on the 32x32 K-split tiling, (dy^T x) + (dy W) layouts
    dv_gemm_desc e1 = *d1, e2 = *d2;
    e2.flags &= 3;
    if (t1 == 2 && t2 == 2) {      // (32-wide tiles: the bias gradient rides on the ones column)
        rc = colsum_setup(e1, 2, st);
        if (rc != DV_OK) return rc;
    }
    const bool pipe = pipe_on(e1) && t1 == 2 && t2 == 2 && pipe_ok(e1, lc1) && pipe_ok(e2, lc2);
    if (pipe) pipe_ones_off(e1);
    const int tiles1 = ((e1.M + 31) / 32) * ((cols_eff(e1) + 31) / 32), tiles2 = ((e2.M + 31) / 32) * ((e2.N + 31) / 32);
    // pairing pays for the latency-bound small products; a product that already fills the chip
    // several times over (>= 4 workgroups per CU) gains nothing from a partner (measured: the
    // decoder-heads pair ran 66 us paired vs 29 + 29 us alone)
    const bool fuse = t1 == 2 && t2 == 2 && !d1->a_kcontig && !d1->b_kcontig && d2->a_kcontig && !d2->b_kcontig &&
                      tune_of(*d1).opt[2] == 0 && tiles1 < 1024 && tiles2 < 1024;
    // both products of a chip-filling layer in ONE launch of the high-occupancy tiling (dv_gemm_set_option(9, 1)
    // switches it off): the long-K data-gradient tiles first, the weight-gradient tiles fill the CUs around them --
    // with seven workgroups per CU the two grids really overlap (decoder heads at cfg 2: 21 + 25 us as two launches,
    // ~32 us as one; the old four-per-CU pairing of two chip-filling products was SLOWER than two launches)
    if (!fuse && tune_of(*d1).opt[9] != 1 && t1 == 2 && t2 == 2 && !d1->a_kcontig && !d1->b_kcontig && d2->a_kcontig &&
        !d2->b_kcontig && tiles1 >= dense_min_tiles(e1) && tiles1 + tiles2 < 4096) {
        dv_gemm_desc first = e2, second = e1;
        if (first.pub_flag == nullptr && second.pub_flag != nullptr) {      // the kernel publishes for its first product
            first.pub_flag = second.pub_flag;
            first.pub_ctr = second.pub_ctr;
            first.pub_add = second.pub_add;
            second.pub_flag = nullptr;
        }
        if (pipe)
            hipLaunchKernelGGL((gemm_pair_pipe_kernel<32, 32, 32, 4, true, false, false, false, 2, DV_DENSE_WG>), dim3(tiles1 + tiles2),
                               dim3(256), 0, st, first, lc2, second, lc1, tiles2);
        else
            hipLaunchKernelGGL((gemm_pair_kernel<32, 32, 32, 1, 1, 4, true, false, false, false>), dim3(tiles1 + tiles2),
                               dim3(256), 0, st, first, lc2, second, lc1, tiles2);
        DV_RETURN_LAUNCH();
    }
    if (!fuse) {
        // (gemm_launch runs its own colsum_setup / ones offset on the caller's descriptors)
        rc = gemm_launch(*d1, lc1, t1, st);
        if (rc != DV_OK) return rc;
        return gemm_launch(*d2, lc2, t2, st);
    }
    dv_gemm_desc first = e1;
    if (first.pub_flag == nullptr && d2->pub_flag != nullptr) {
        first.pub_flag = d2->pub_flag;
        first.pub_ctr = d2->pub_ctr;
        first.pub_add = d2->pub_add;
    }
    if (pipe)
        hipLaunchKernelGGL((gemm_pair_pipe_kernel<32, 32, 64, 4, false, false, true, false, 2, 4>), dim3(tiles1 + tiles2),
                           dim3(256), 0, st, first, lc1, e2, lc2, tiles1);
    else
        hipLaunchKernelGGL((gemm_pair_kernel<32, 32, 64, 1, 1, 4, false, false, true, false>), dim3(tiles1 + tiles2),
                           dim3(256), 0, st, first, lc1, e2, lc2, tiles1);
    DV_RETURN_LAUNCH();
}
