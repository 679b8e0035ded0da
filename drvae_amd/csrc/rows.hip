// Row-wise kernels of the Dr.VAE ELBO hot path (gfx950): reparameterisation, diagonal-
// Gaussian KL with free bits, Gaussian log-likelihood over genes, the categorical head,
// the y-marginalisation, row gathers / segment sums, WeightNorm scale + backward.
// All are HBM/L2-bound streaming kernels: coalesced row reads (16-B per lane where the
// layout allows), one 64-lane wavefront per row for the reductions (__shfl), no atomics
// (results are bitwise reproducible).
#include "dv_common.h"

namespace {

constexpr float kLog2Pi = 1.8378770664093453f;   // float(np.log(2*np.pi)), src/blocks.py:196,234

inline int grid_for(int64_t work, int per_block, int cap = 4096) {
    int64_t b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// ------------------------------------------------------------------ colsum / act_bwd
// 64 columns x 4 row-lanes per block; rows strided by 4*gridDim.y
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int64_t ldx, int M, int N,
                                                     float* __restrict__ out, float beta) {
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    float s = 0.f;
    if (col < N) {
        // eight independent loads per trip (the loop is a chain of memory round trips otherwise: 0.88 ms for the
        // 4096 x 40000 gradient of the wide configuration's decoder heads, i.e. 0.7 TB/s); fixed summation order
        const float* p = X + (int64_t)rl * ldx + col;
        const int64_t st = 4 * ldx;
        int m = rl;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
        for (; m + 28 < M; m += 32, p += 8 * st) {
            const float v0 = p[0], v1 = p[st], v2 = p[2 * st], v3 = p[3 * st];
            const float v4 = p[4 * st], v5 = p[5 * st], v6 = p[6 * st], v7 = p[7 * st];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
            a4 += v4; a5 += v5; a6 += v6; a7 += v7;
        }
        s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
        for (; m < M; m += 4, p += st) s += p[0];
    }
    part[rl][c] = s;
    __syncthreads();
    if (rl == 0 && col < N) {
        s = part[0][c] + part[1][c] + part[2][c] + part[3][c];
        out[col] = (beta != 0.f ? beta * out[col] : 0.f) + s;
    }
}

__global__ void act_bwd_kernel(float* __restrict__ dY, int64_t ldd, const float* __restrict__ Y, int64_t ldy, int M,
                               int N, int split, int act0, int act1, float shift0, float shift1) {
    const int64_t total = (int64_t)M * N;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(e / N), n = (int)(e % N);
        const bool first = n < split;
        const float y = Y[(int64_t)m * ldy + n] - (first ? shift0 : shift1);
        dY[(int64_t)m * ldd + n] *= dv_dact_from_y(first ? act0 : act1, y);
    }
}

// --------------------------------------------------------------------- WeightNorm
__global__ __launch_bounds__(256) void wn_scale_kernel(const float* __restrict__ W, int64_t ldw,
                                                       const float* __restrict__ g, int N, int K,
                                                       float* __restrict__ scale, float* __restrict__ norm) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* w = W + (int64_t)row * ldw;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += w[k] * w[k];
    s = dv_wave_sum_all(s);
    if (lane == 0) {
        const float nrm = sqrtf(s);
        if (norm) norm[row] = nrm;
        scale[row] = g[row] / nrm;
    }
}

__global__ __launch_bounds__(256) void wn_bwd_kernel(const float* __restrict__ dWraw, int64_t ldr,
                                                     const float* __restrict__ W, int64_t ldw,
                                                     const float* __restrict__ g, const float* __restrict__ norm,
                                                     int N, int K, float* __restrict__ dW, int64_t ldd,
                                                     float* __restrict__ dg, float beta) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* w = W + (int64_t)row * ldw;
    const float* r = dWraw + (int64_t)row * ldr;
    float dot = 0.f;
    for (int k = lane; k < K; k += 64) dot += w[k] * r[k];
    dot = dv_wave_sum_all(dot);
    const float nrm = norm[row], gg = g[row];
    const float sc = gg / nrm, cw = dot * gg / (nrm * nrm * nrm);
    float* o = dW + (int64_t)row * ldd;
    for (int k = lane; k < K; k += 64) {
        const float v = sc * r[k] - cw * w[k];
        o[k] = (beta != 0.f ? beta * o[k] : 0.f) + v;
    }
    if (lane == 0) dg[row] = (beta != 0.f ? beta * dg[row] : 0.f) + dot / nrm;
}

// ---------------------------------------------------------------------- reparam
__global__ void reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ sd, int64_t ldq,
                                   const int32_t* __restrict__ src_idx, int n, int reps, int Z,
                                   const float* __restrict__ eps, int64_t lde, int mode, float* __restrict__ out,
                                   int64_t ldo, const float* __restrict__ sub, int64_t lds,
                                   float* __restrict__ out2, int64_t ldo2, float* __restrict__ out3, int64_t ldo3,
                                   const int32_t* __restrict__ out3_idx) {
    const int64_t total = (int64_t)n * reps * Z;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / Z), d = (int)(e % Z);
        const int j = r % n;
        const int64_t qi = src_idx ? src_idx[j] : j;
        const float s = sd[qi * ldq + d];
        const float std_ = mode == DV_GAUSS_LOGVAR ? expf(0.5f * s) : s;
        const float z = eps[(int64_t)r * lde + d] * std_ + mu[qi * ldq + d];
        out[(int64_t)r * ldo + d] = z;
        if (out2) out2[(int64_t)r * ldo2 + d] = z - sub[(int64_t)r * lds + d];
        if (out3) {
            const int t = out3_idx[r];
            if (t >= 0) out3[(int64_t)t * ldo3 + d] = z;
        }
    }
}

__global__ void reparam_bwd_kernel(const float* __restrict__ dz, int64_t ldz, const float* __restrict__ eps,
                                   int64_t lde, const float* __restrict__ sd, int64_t ldq,
                                   const int32_t* __restrict__ src_idx, int n, int reps, int Z, int mode,
                                   float* __restrict__ dmu, float* __restrict__ dsd, int64_t lddq, float beta) {
    const int64_t total = (int64_t)n * Z;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e / Z), d = (int)(e % Z);
        const int64_t qi = src_idx ? src_idx[j] : j;
        float a = 0.f, b = 0.f;
        for (int l = 0; l < reps; ++l) {
            const int64_t r = (int64_t)l * n + j;
            const float g = dz[r * ldz + d];
            a += g;
            b += g * eps[r * lde + d];
        }
        if (mode == DV_GAUSS_LOGVAR) b *= 0.5f * expf(0.5f * sd[qi * ldq + d]);
        float* pm = dmu + qi * lddq + d;
        float* ps = dsd + qi * lddq + d;
        *pm = (beta != 0.f ? beta * *pm : 0.f) + a;
        *ps = (beta != 0.f ? beta * *ps : 0.f) + b;
    }
}

// CSR form: q row i collects the samples listed in seg_rows[seg_ptr[i]..seg_ptr[i+1]) (any number of
// draws per row, e.g. L z1-samples + L z2-samples of a paired row) plus, optionally, row-aligned
// (dmu | dsd) contributions listed in a second CSR (the KL(q(z1|x)||p(z1|z3,y)) gradients of its
// fprop rows).  One launch replaces reparam_bwd x2 + a segment sum; deterministic.
// ---- joins folded into consumers (dv_wait / dv_bump arguments): a launch may first park every workgroup
// on another chain's flag (thread 0 polls, bounded), and may end by advancing device counters
#define DV_MAX_PARKED_GRID 512
typedef dv_wait ParkArgs;
typedef dv_bump CounterBump;

__device__ __forceinline__ void bump_counters(const CounterBump& bump) {
    for (int t = 0; t < 2; ++t) {
        int32_t* c = bump.c[t];
        if (c == nullptr) continue;
        if (bump.n[t] == 1) {
            c[0] = (int32_t)(c[0] + bump.inc[t]);
        } else {
            uint64_t v = ((uint64_t)(uint32_t)c[1] << 32) | (uint32_t)c[0];
            v += (uint64_t)bump.inc[t];
            c[0] = (int32_t)(uint32_t)v;
            c[1] = (int32_t)(uint32_t)(v >> 32);
        }
    }
}

__global__ void reparam_bwd_seg_kernel(const float* __restrict__ dz, int64_t ldz, const float* __restrict__ eps,
                                       int64_t lde, const float* __restrict__ sd, int64_t ldq,
                                       const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_rows,
                                       int nq, int Z, int mode, const float* __restrict__ extra, int64_t ldx,
                                       const int32_t* __restrict__ ex_ptr, const int32_t* __restrict__ ex_rows,
                                       float* __restrict__ dmu, float* __restrict__ dsd, int64_t lddq, float beta,
                                       CounterBump bump, const float* __restrict__ dz2, int64_t ldz2, int n2,
                                       ParkArgs park, dv_prior_kl pk) {
    park_block(park);
    // (the counters are not read by this kernel: whoever starts first may advance them)
    if (blockIdx.x == 0 && threadIdx.x == 0) bump_counters(bump);
    const int64_t total = (int64_t)nq * Z;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / Z), d = (int)(e % Z);
        float a = 0.f, b = 0.f;
        for (int t = seg_ptr[i]; t < seg_ptr[i + 1]; ++t) {
            const int64_t r = seg_rows[t];
            float g = dz[r * ldz + d];
            if (r < n2) g += dz2[r * ldz2 + d];
            a += g;
            b += g * eps[r * lde + d];
        }
        if (mode == DV_GAUSS_LOGVAR) b *= 0.5f * expf(0.5f * sd[(int64_t)i * ldq + d]);
        if (extra) {
            for (int t = ex_ptr[i]; t < ex_ptr[i + 1]; ++t) {
                const int64_t r = ex_rows[t];
                a += extra[r * ldx + d];
                b += extra[r * ldx + Z + d];
            }
        }
        float* pm = dmu + (int64_t)i * lddq + d;
        float* ps = dsd + (int64_t)i * lddq + d;
        float vm = (beta != 0.f ? beta * *pm : 0.f) + a;
        float vs = (beta != 0.f ? beta * *ps : 0.f) + b;
        if (pk.coef != nullptr) {       // + d/d(mu, logvar) of coef * max(KL(q_i || N(0,I)), kl_min): what dv_kl_rows_bwd(beta = 1) added
            const float rv = pk.raw[i];
            const float c = pk.coef[i] * (rv > pk.kl_min ? 1.f : (rv == pk.kl_min ? 0.5f : 0.f));
            const float mq = pk.mu[(int64_t)i * pk.ld + d], vq = expf(sd[(int64_t)i * ldq + d]);
            vm += c * (mq * 1.f);
            vs += c * (-0.5f * (1.f - vq * 1.f));
        }
        *pm = vm;
        *ps = vs;
    }
}

// Backward of everything that hangs on the z2Fz1 samples (src/DrVAE.py:431-433,459-487) in one
// pass over (row i, dim d), looping the L samples:
//   dz2F = DZ2F[(l,i)] (+ gradient of the decoded copy for pairs)
//   DP2[(l,i)] = (dmu2 | dlv2) of p(z2|z1): reparam backward (+ KL(q(z2|x2)||p(z2|z1)) wrt p for pairs)
//   DZ1[(l,i)] += dmu2 (residual path mu2 = z1 + ...) (+ DZ1B[(l,i)], the side chain's share)
//   DQ2[jp]    = sum_l KL gradient wrt q(z2|x2)           (pairs only)
struct Z2FArgs {
    const float* dz2f; int64_t ld_dz2f;       // (L*B, Z)
    const float* dzdec_pert; int64_t ld_pert;  // (L*Np, Z) gradient of the decoded z2Fz1 copies, or NULL
    const int32_t* pair_slot;                  // (B) jp or -1
    const float* eps; int64_t lde;             // (L*B, Z)
    const float* p2; int64_t ldp2;             // (L*B, 2Z)  mu2 | lv2
    const float* q2; int64_t ldq2;             // (Np, 2Z)   mu | lv of q(z2|x2)
    const float* coef; const float* raw; float kl_min;   // (L*Np) KL coefficients / raw KL (free bits)
    const float* dz1b; int64_t ld_dz1b;        // (L*B, Z) or NULL
    float* dp2; int64_t ld_dp2;                // (L*B, 2Z) out
    float* dz1; int64_t ld_dz1;                // (L*B, Z) in/out (+=)
    float* dq2; int64_t ld_dq2;                // (Np, 2Z) out, or NULL
    int L, B, Np, Z;
    const float* prior_coef; const float* prior_raw;   // (Np) or NULL: the prior-KL gradient of q2 rides along (dv_prior_kl)
};

__global__ void z2f_post_bwd_kernel(Z2FArgs a, ParkArgs park) {
    park_block(park);
    const int64_t total = (int64_t)a.B * a.Z;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / a.Z), d = (int)(e % a.Z);
        const int jp = a.pair_slot ? a.pair_slot[i] : -1;
        float mq = 0.f, lq = 0.f, gq_mu = 0.f, gq_lv = 0.f;
        if (jp >= 0) {
            mq = a.q2[(int64_t)jp * a.ldq2 + d];
            lq = a.q2[(int64_t)jp * a.ldq2 + a.Z + d];
        }
        for (int l = 0; l < a.L; ++l) {
            const int64_t r = (int64_t)l * a.B + i;
            const float mp = a.p2[r * a.ldp2 + d], lp = a.p2[r * a.ldp2 + a.Z + d];
            float g = a.dz2f ? a.dz2f[r * a.ld_dz2f + d] : 0.f;
            if (jp >= 0 && a.dzdec_pert) g += a.dzdec_pert[((int64_t)l * a.Np + jp) * a.ld_pert + d];
            float dmu = g, dlv = g * a.eps[r * a.lde + d] * 0.5f * expf(0.5f * lp);
            if (jp >= 0) {
                const int64_t kr = (int64_t)l * a.Np + jp;
                const float rv = a.raw[kr];
                const float c = a.coef[kr] * (rv > a.kl_min ? 1.f : (rv == a.kl_min ? 0.5f : 0.f));
                const float dm = mq - mp, ivp = expf(-lp), vq = expf(lq);
                gq_mu += c * dm * ivp;
                gq_lv += c * (-0.5f * (1.f - vq * ivp));
                dmu += -c * dm * ivp;
                dlv += c * (-0.5f * (-1.f + (dm * dm + vq) * ivp));
            }
            a.dp2[r * a.ld_dp2 + d] = dmu;
            a.dp2[r * a.ld_dp2 + a.Z + d] = dlv;
            float* z1 = a.dz1 + r * a.ld_dz1 + d;
            *z1 += dmu + (a.dz1b ? a.dz1b[r * a.ld_dz1b + d] : 0.f);
        }
        if (jp >= 0 && a.dq2) {
            if (a.prior_coef != nullptr) {      // + the prior term of q(z2|x2) row jp (what dv_kl_rows_bwd(beta = 1) added)
                const float rv = a.prior_raw[jp];
                const float c = a.prior_coef[jp] * (rv > a.kl_min ? 1.f : (rv == a.kl_min ? 0.5f : 0.f));
                gq_mu += c * (mq * 1.f);
                gq_lv += c * (-0.5f * (1.f - expf(lq) * 1.f));
            }
            a.dq2[(int64_t)jp * a.ld_dq2 + d] = gq_mu;
            a.dq2[(int64_t)jp * a.ld_dq2 + a.Z + d] = gq_lv;
        }
    }
}

// ---------------------------------------------------------------------- KL rows
struct KlArgs {
    const float *mu_q, *sd_q;
    int64_t ldq;
    const int32_t* qidx;
    const float *mu_p, *sd_p;
    int64_t ldp;
    const int32_t* pidx;
    float prior_mu, prior_sd;
    int n, reps, Z, mode;
};

__device__ __forceinline__ float kl_term(int mode, float mq, float sq, float mp, float sp) {
    const float dm = mq - mp;
    if (mode == DV_GAUSS_LOGVAR) return 1.f - sp + sq - (dm * dm + expf(sq)) / expf(sp);
    const float vq = sq * sq, vp = sp * sp;
    return 1.f - logf(vp) + logf(vq) - (dm * dm + vq) / vp;
}

__global__ __launch_bounds__(256) void kl_rows_fwd_kernel(KlArgs a, int free_bits, float kl_min,
                                                          float* __restrict__ raw_out, float* __restrict__ out,
                                                          const float* __restrict__ add,
                                                          const float* __restrict__ eps, int64_t lde,
                                                          float* __restrict__ zout, int64_t ldz, ParkArgs park,
                                                          const float* __restrict__ mu2, const float* __restrict__ sd2,
                                                          int64_t ld2, int Z2, float* __restrict__ raw2_out) {
    park_block(park);
    const int lane = threadIdx.x & 63;
    const int rows = a.n * a.reps;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
        const int j = r % a.n;
        const int64_t qi = a.qidx ? a.qidx[j] : j;
        const int64_t pi = a.pidx ? a.pidx[r] : r;
        float s = 0.f, s2 = 0.f;
        if (mu2)   // second term of the same row: KL(N(mu2, sd2) || N(prior_mu, prior_sd)), row-aligned
            for (int d = lane; d < Z2; d += 64)
                s2 += kl_term(a.mode, mu2[(int64_t)r * ld2 + d], sd2[(int64_t)r * ld2 + d], a.prior_mu, a.prior_sd);
        for (int d = lane; d < a.Z; d += 64) {
            const float mq = a.mu_q[qi * a.ldq + d], sq = a.sd_q[qi * a.ldq + d];
            const float mp = a.mu_p ? a.mu_p[pi * a.ldp + d] : a.prior_mu;
            const float sp = a.mu_p ? a.sd_p[pi * a.ldp + d] : a.prior_sd;
            s += kl_term(a.mode, mq, sq, mp, sp);
            if (zout)   // fused reparameterised sample of q (same row pass)
                zout[(int64_t)r * ldz + d] =
                    mq + eps[(int64_t)r * lde + d] * (a.mode == DV_GAUSS_LOGVAR ? expf(0.5f * sq) : sq);
        }
        s = dv_wave_sum_all(s);
        if (mu2) s2 = dv_wave_sum_all(s2);
        if (lane == 0) {
            const float raw = -0.5f * s;
            if (raw_out) raw_out[r] = raw;
            float v = (free_bits ? fmaxf(raw, kl_min) : raw) + (add ? add[r] : 0.f);
            if (mu2) {
                const float raw2 = -0.5f * s2;
                if (raw2_out) raw2_out[r] = raw2;
                v += free_bits ? fmaxf(raw2, kl_min) : raw2;
            }
            out[r] = v;
        }
    }
}

// Two independent sets of KL rows in ONE launch (round 5; PVAE's main chain runs KL(q(z1|x) || N(0,I)) and the pairs'
// KL(q(z2|x2) || p(z2|z1)) back to back: two launches of ~40 workgroups each): workgroups [0, blocks_a) take set a, the rest
// set b; per row the arithmetic and summation order of kl_rows_fwd_kernel (no fused sample, no second term, no park).
struct KlFwd {
    KlArgs k;
    int free_bits;
    float kl_min;
    float* raw_out;
    float* out;
    const float* add;
};

__global__ __launch_bounds__(256) void kl_rows_fwd_pair_kernel(KlFwd sa, KlFwd sb, int blocks_a) {
    const bool first = (int)blockIdx.x < blocks_a;
    const KlFwd& s = first ? sa : sb;
    const KlArgs& a = s.k;
    const int blk = first ? blockIdx.x : blockIdx.x - blocks_a;
    const int lane = threadIdx.x & 63;
    const int rows = a.n * a.reps;
    const int r = blk * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int j = r % a.n;
    const int64_t qi = a.qidx ? a.qidx[j] : j;
    const int64_t pi = a.pidx ? a.pidx[r] : r;
    float sum = 0.f;
    for (int d = lane; d < a.Z; d += 64) {
        const float mq = a.mu_q[qi * a.ldq + d], sq = a.sd_q[qi * a.ldq + d];
        const float mp = a.mu_p ? a.mu_p[pi * a.ldp + d] : a.prior_mu;
        const float sp = a.mu_p ? a.sd_p[pi * a.ldp + d] : a.prior_sd;
        sum += kl_term(a.mode, mq, sq, mp, sp);
    }
    sum = dv_wave_sum_all(sum);
    if (lane == 0) {
        const float raw = -0.5f * sum;
        if (s.raw_out) s.raw_out[r] = raw;
        s.out[r] = (s.free_bits ? fmaxf(raw, s.kl_min) : raw) + (s.add ? s.add[r] : 0.f);
    }
}

__global__ void kl_rows_bwd_kernel(KlArgs a, const float* __restrict__ coef, const float* __restrict__ raw,
                                   int free_bits, float kl_min, float* __restrict__ dq_mu,
                                   float* __restrict__ dq_sd, int64_t lddq, float* __restrict__ dp_mu,
                                   float* __restrict__ dp_sd, int64_t lddp, float beta,
                                   const float* __restrict__ dz, int64_t ldz, const float* __restrict__ eps,
                                   int64_t lde) {
    const int64_t total = (int64_t)a.n * a.reps * a.Z;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / a.Z), d = (int)(e % a.Z);
        const int j = r % a.n;
        const int64_t qi = a.qidx ? a.qidx[j] : j;
        const int64_t pi = a.pidx ? a.pidx[r] : r;
        float c = coef[r];
        if (free_bits) {   // d max(raw, kl_min)/d raw: 1 above, 0 below, 1/2 on a tie (torch.max)
            const float rv = raw[r];
            c *= rv > kl_min ? 1.f : (rv == kl_min ? 0.5f : 0.f);
        }
        const float mq = a.mu_q[qi * a.ldq + d], sq = a.sd_q[qi * a.ldq + d];
        const float mp = a.mu_p ? a.mu_p[pi * a.ldp + d] : a.prior_mu;
        const float sp = a.mu_p ? a.sd_p[pi * a.ldp + d] : a.prior_sd;
        const float dm = mq - mp;
        float gmq, gsq, gmp, gsp;
        if (a.mode == DV_GAUSS_LOGVAR) {
            const float ivp = expf(-sp), vq = expf(sq);
            gmq = dm * ivp;
            gsq = -0.5f * (1.f - vq * ivp);
            gmp = -gmq;
            gsp = -0.5f * (-1.f + (dm * dm + vq) * ivp);
        } else {
            const float vp = sp * sp;
            gmq = dm / vp;
            gsq = -1.f / sq + sq / vp;
            gmp = -gmq;
            gsp = 1.f / sp - (dm * dm + sq * sq) / (vp * sp);
        }
        float tq_mu = c * gmq, tq_sd = c * gsq;
        if (dz) {   // the sample drawn from q in the forward pass: z = mu + eps*std
            const float g = dz[(int64_t)r * ldz + d];
            tq_mu += g;
            tq_sd += g * eps[(int64_t)r * lde + d] * (a.mode == DV_GAUSS_LOGVAR ? 0.5f * expf(0.5f * sq) : 1.f);
        }
        const int64_t oq = (int64_t)r * lddq + d;
        dq_mu[oq] = (beta != 0.f ? beta * dq_mu[oq] : 0.f) + tq_mu;
        dq_sd[oq] = (beta != 0.f ? beta * dq_sd[oq] : 0.f) + tq_sd;
        if (dp_mu) {
            const int64_t op = (int64_t)r * lddp + d;
            dp_mu[op] = (beta != 0.f ? beta * dp_mu[op] : 0.f) + c * gmp;
            dp_sd[op] = (beta != 0.f ? beta * dp_sd[op] : 0.f) + c * gsp;
        }
    }
}

// -------------------------------------------------------- Gaussian NLL over genes
__device__ __forceinline__ float nll_term(int mode, float x, float m, float s) {
    const float d = x - m;
    if (mode == DV_GAUSS_SIGMA) {
        const float v = s * s;
        return kLog2Pi + logf(v) + d * d / v;
    }
    return kLog2Pi + s + d * d / expf(s);
}

// one term of a row's Gaussian log-likelihood from the heads' RAW product (x W^T without bias / activation): mu = m + bm,
// sd = softplus(s + bs) + shift on the hardware transcendentals (the forward half of nll_raw_sp_elem below)
__device__ __forceinline__ float nll_raw_term(float shift, float xv, float mraw, float sraw, float bm, float bs) {
    const float a = sraw + bs;
    const float e = __expf(-fabsf(a));
    const float sdv = fmaxf(a, 0.f) + __logf(1.f + e) + shift;
    const float t = (xv - (mraw + bm)) * __frcp_rn(sdv);
    return kLog2Pi + 2.f * __logf(sdv) + t * t;
}

// RAW (evaluation passes behind a plain heads product, round 5): mu / sd are the raw products, finished on the way
template <int VEC, bool RAW>
__global__ __launch_bounds__(256) void nll_rows_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                           const int32_t* __restrict__ xidx,
                                                           const float* __restrict__ mu,
                                                           const float* __restrict__ sd, int64_t ldp, int M, int X,
                                                           int mode, float* __restrict__ out,
                                                           const float* __restrict__ bias_mu,
                                                           const float* __restrict__ bias_sd, float shift) {
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < M; r += gridDim.x * 4) {
        const float* xr = x + (int64_t)(xidx ? xidx[r] : r) * ldx;
        const float* mr = mu + (int64_t)r * ldp;
        const float* sr = sd + (int64_t)r * ldp;
        float s = 0.f;
        auto term = [&](float xv, float m, float sv, int g) -> float {
            if constexpr (RAW) return nll_raw_term(shift, xv, m, sv, bias_mu[g], bias_sd[g]);
            else return nll_term(mode, xv, m, sv);
        };
        if constexpr (VEC == 4) {
            const int X4 = X >> 2;
            for (int c = lane; c < X4; c += 64) {
                const float4 xv = reinterpret_cast<const float4*>(xr)[c];
                const float4 mv = reinterpret_cast<const float4*>(mr)[c];
                const float4 sv = reinterpret_cast<const float4*>(sr)[c];
                s += term(xv.x, mv.x, sv.x, 4 * c) + term(xv.y, mv.y, sv.y, 4 * c + 1) +
                     term(xv.z, mv.z, sv.z, 4 * c + 2) + term(xv.w, mv.w, sv.w, 4 * c + 3);
            }
            for (int g = (X4 << 2) + lane; g < X; g += 64) s += term(xr[g], mr[g], sr[g], g);
        } else if constexpr (VEC == 2) {
            // (978-gene rows: the sigma half starts 8 B off a 16-B boundary -- 8-B loads, two pairs per trip; dword loads
            // hold a row pass at 3.9 TB/s, round 5)
            const int X2 = X >> 1;
            auto pair = [&](int c) -> float {
                const float2 xv = reinterpret_cast<const float2*>(xr)[c];
                const float2 mv = reinterpret_cast<const float2*>(mr)[c];
                const float2 sv = reinterpret_cast<const float2*>(sr)[c];
                if constexpr (RAW) {
                    const float2 bm = reinterpret_cast<const float2*>(bias_mu)[c];
                    const float2 bs = reinterpret_cast<const float2*>(bias_sd)[c];
                    return nll_raw_term(shift, xv.x, mv.x, sv.x, bm.x, bs.x) + nll_raw_term(shift, xv.y, mv.y, sv.y, bm.y, bs.y);
                } else {
                    return nll_term(mode, xv.x, mv.x, sv.x) + nll_term(mode, xv.y, mv.y, sv.y);
                }
            };
            int c = lane;
            for (; c + 64 < X2; c += 128) s += pair(c) + pair(c + 64);
            for (; c < X2; c += 64) s += pair(c);
            if ((X & 1) && lane == 0) s += term(xr[X - 1], mr[X - 1], sr[X - 1], X - 1);
        } else {
            // (four independent elements per trip)
            int g = lane;
            for (; g + 192 < X; g += 256)
                s += (term(xr[g], mr[g], sr[g], g) + term(xr[g + 64], mr[g + 64], sr[g + 64], g + 64)) +
                     (term(xr[g + 128], mr[g + 128], sr[g + 128], g + 128) + term(xr[g + 192], mr[g + 192], sr[g + 192], g + 192));
            for (; g < X; g += 64) s += term(xr[g], mr[g], sr[g], g);
        }
        s = dv_wave_sum_all(s);
        if (lane == 0) out[r] = -0.5f * s;
    }
}

// forward AND backward of the reconstruction term in one row pass: the loss is linear in the
// per-row log-likelihoods with coefficients known before the forward pass, so d/d(mu, pre-act of
// sd) can be emitted while the row sum is being formed (3 reads + 2 writes per element instead
// of 6 + 2 over two launches)
__device__ __forceinline__ void nll_fb_elem(int mode, int sd_act, float sd_shift, float c, float xv, float m, float s,
                                            float& acc, float& gm, float& gs) {
    const float d = xv - m;
    if (mode == DV_GAUSS_SIGMA) {
        const float v = s * s;
        acc += kLog2Pi + logf(v) + d * d / v;
        gm = d / v;
        gs = -1.f / s + d * d / (v * s);
    } else {
        const float iv = expf(-s);
        acc += kLog2Pi + s + d * d * iv;
        gm = d * iv;
        gs = -0.5f * (1.f - d * d * iv);
    }
    if (sd_act != DV_ACT_IDENTITY) gs *= dv_dact_from_y(sd_act, s - sd_shift);
    gm *= c;
    gs *= c;
}

template <bool VEC2>
__global__ __launch_bounds__(256) void nll_rows_fwdbwd_kernel(const float* __restrict__ coef,
                                                              const float* __restrict__ x, int64_t ldx,
                                                              const int32_t* __restrict__ xidx,
                                                              const float* __restrict__ mu,
                                                              const float* __restrict__ sd, int64_t ldp, int M, int X,
                                                              int mode, int sd_act, float sd_shift,
                                                              float* __restrict__ out, float* __restrict__ dmu,
                                                              float* __restrict__ dsd, int64_t ldd,
                                                              const float* __restrict__ bias_mu,
                                                              const float* __restrict__ bias_sd) {
    // one WORKGROUP per row, its four waves splitting the genes: a few hundred rows of ~1000 genes is
    // too little for wave-per-row to fill the chip (596 waves = 149 workgroups), and each wave's
    // dependent load->store chain is then 4x shorter
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x; r < M; r += gridDim.x) {
        const float* xr = x + (int64_t)(xidx ? xidx[r] : r) * ldx;
        const float* mr = mu + (int64_t)r * ldp;
        const float* sr = sd + (int64_t)r * ldp;
        float* gmr = dmu + (int64_t)r * ldd;
        float* gsr = dsd + (int64_t)r * ldd;
        const float c = coef[r];
        float acc = 0.f;
        // bias_mu != NULL: mu / sd hold the heads' RAW products (x W^T, no bias, no activation) -- the chip-filling
        // heads product then runs with the plain epilogue (8192 x 40000 x 2048: 9.97 ms instead of 10.87 with bias +
        // softplus in its 128x128 tiles' epilogue) and this HBM-bound pass applies them on its way
        const bool raw = bias_mu != nullptr;
        auto fin_m = [&](float v, int g) { return raw ? v + bias_mu[g] : v; };
        auto fin_s = [&](float v, int g) { return raw ? dv_act(sd_act, v + bias_sd[g]) + sd_shift : v; };
        if (VEC2) {
            const int X2 = X >> 1;
            for (int q = threadIdx.x; q < X2; q += 256) {
                const float2 xv = reinterpret_cast<const float2*>(xr)[q];
                float2 mv = reinterpret_cast<const float2*>(mr)[q];
                float2 sv = reinterpret_cast<const float2*>(sr)[q];
                mv.x = fin_m(mv.x, 2 * q);
                mv.y = fin_m(mv.y, 2 * q + 1);
                sv.x = fin_s(sv.x, 2 * q);
                sv.y = fin_s(sv.y, 2 * q + 1);
                float2 gm, gs;
                nll_fb_elem(mode, sd_act, sd_shift, c, xv.x, mv.x, sv.x, acc, gm.x, gs.x);
                nll_fb_elem(mode, sd_act, sd_shift, c, xv.y, mv.y, sv.y, acc, gm.y, gs.y);
                reinterpret_cast<float2*>(gmr)[q] = gm;
                reinterpret_cast<float2*>(gsr)[q] = gs;
            }
            if ((X & 1) && threadIdx.x == 0) {
                float gm, gs;
                nll_fb_elem(mode, sd_act, sd_shift, c, xr[X - 1], fin_m(mr[X - 1], X - 1), fin_s(sr[X - 1], X - 1), acc, gm, gs);
                gmr[X - 1] = gm;
                gsr[X - 1] = gs;
            }
        } else {
            for (int g = threadIdx.x; g < X; g += 256) {
                float gm, gs;
                nll_fb_elem(mode, sd_act, sd_shift, c, xr[g], fin_m(mr[g], g), fin_s(sr[g], g), acc, gm, gs);
                gmr[g] = gm;
                gsr[g] = gs;
            }
        }
        acc = dv_wave_sum_all(acc);
        if (lane == 0) part[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) out[r] = -0.5f * ((part[0] + part[1]) + (part[2] + part[3]));
        __syncthreads();
    }
}

// The raw-heads case of the pass above for the decoder the models build (sigma head = softplus + shift), written for
// the size it exists for (wide configuration: 8192 rows x 20000 genes, 2.9 GB of traffic): four genes per lane and the
// hardware transcendentals -- one exp, two logs, two reciprocals per gene instead of the library softplus / log / exp /
// divisions, which made the pass compute-bound (1.0 ms; this form 0.6 ms = the HBM time).
//   a = sd_raw + b_sd;  e = exp(-|a|);  softplus(a) = max(a, 0) + log(1 + e);  sigmoid(a) = (a >= 0 ? 1 : e) / (1 + e)
__device__ __forceinline__ void nll_raw_sp_elem(float c, float shift, float xv, float mraw, float sraw, float bm, float bs,
                                                float& acc, float& gm, float& gs) {
    const float m = mraw + bm, a = sraw + bs;
    const float e = __expf(-fabsf(a));
    const float r1 = __frcp_rn(1.f + e);
    const float s = fmaxf(a, 0.f) + __logf(1.f + e) + shift;
    const float sig = (a >= 0.f ? 1.f : e) * r1;
    const float is = __frcp_rn(s), d = xv - m, t = d * is;
    acc += kLog2Pi + 2.f * __logf(s) + t * t;
    gm = c * t * is;
    gs = c * (t * t - 1.f) * is * sig;
}

__global__ __launch_bounds__(256) void nll_rows_raw_sp_kernel(const float* __restrict__ coef, const float* __restrict__ x,
                                                              int64_t ldx, const int32_t* __restrict__ xidx,
                                                              const float* __restrict__ mu, const float* __restrict__ sd,
                                                              int64_t ldp, int M, int X, float shift,
                                                              float* __restrict__ out, float* __restrict__ dmu,
                                                              float* __restrict__ dsd, int64_t ldd,
                                                              const float* __restrict__ bias_mu,
                                                              const float* __restrict__ bias_sd) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int X4 = X >> 2;
    for (int r = blockIdx.x; r < M; r += gridDim.x) {
        const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)(xidx ? xidx[r] : r) * ldx);
        const float4* mr = reinterpret_cast<const float4*>(mu + (int64_t)r * ldp);
        const float4* sr = reinterpret_cast<const float4*>(sd + (int64_t)r * ldp);
        float4* gmr = reinterpret_cast<float4*>(dmu + (int64_t)r * ldd);
        float4* gsr = reinterpret_cast<float4*>(dsd + (int64_t)r * ldd);
        const float c = coef[r];
        float acc = 0.f;
        // (the raw heads are read once and the gradients written once, 2.6 GB of the pass's 3.3: non-temporal, so that
        // they do not push the targets -- shared by a row's L samples -- and the biases out of the caches)
        typedef float nt_f4 __attribute__((ext_vector_type(4)));
        for (int q = threadIdx.x; q < X4; q += 256) {
            const float4 xv = xr[q];
            const nt_f4 mv = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(mr) + q);
            const nt_f4 sv = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(sr) + q);
            const float4 bm = reinterpret_cast<const float4*>(bias_mu)[q], bs = reinterpret_cast<const float4*>(bias_sd)[q];
            nt_f4 gm, gs;
            float a, b;
            nll_raw_sp_elem(c, shift, xv.x, mv[0], sv[0], bm.x, bs.x, acc, a, b); gm[0] = a; gs[0] = b;
            nll_raw_sp_elem(c, shift, xv.y, mv[1], sv[1], bm.y, bs.y, acc, a, b); gm[1] = a; gs[1] = b;
            nll_raw_sp_elem(c, shift, xv.z, mv[2], sv[2], bm.z, bs.z, acc, a, b); gm[2] = a; gs[2] = b;
            nll_raw_sp_elem(c, shift, xv.w, mv[3], sv[3], bm.w, bs.w, acc, a, b); gm[3] = a; gs[3] = b;
            __builtin_nontemporal_store(gm, reinterpret_cast<nt_f4*>(gmr) + q);
            __builtin_nontemporal_store(gs, reinterpret_cast<nt_f4*>(gsr) + q);
        }
        acc = dv_wave_sum_all(acc);
        if (lane == 0) part[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) out[r] = -0.5f * ((part[0] + part[1]) + (part[2] + part[3]));
        __syncthreads();
    }
}

// The same pass with the bias gradient of the heads folded in (round 5): db = column sums of (dmu | dsd) over the rows
// used to be a pass of its own over the 1.3 GB of gradients this kernel has just written (wide configuration: 0.23 ms
// per step).  Here a workgroup owns 256 four-gene groups x a block of kNllCsRows rows: every lane keeps the column sums
// of its four genes in registers while it walks the rows, and a row's log-likelihood comes out as one partial per
// column chunk (the loss assembly sums a row's partials: dv_loss_term.row_len).  Per-block column sums go to a small
// workspace (row_blocks x ldw), summed in a fixed order by dv_colsum over row_blocks rows: deterministic, no atomics.
constexpr int kNllCsRows = 64;
struct NllCsArgs {
    const float* coef; const float* x; int64_t ldx; const int32_t* xidx; const float* mu; const float* sd; int64_t ldp;
    int M, X; float shift; float* out_part; int chunks; float* dmu; float* dsd; int64_t ldd;
    const float* bias_mu; const float* bias_sd; float* ws; int64_t ldw; int64_t sd_off;
};

template <bool FWD_ONLY>
__global__ __launch_bounds__(256) void nll_rows_raw_cs_kernel(NllCsArgs a) {
    __shared__ float part[4][kNllCsRows];
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int X4 = a.X >> 2;
    const int chunk = blockIdx.x, rb = blockIdx.y;
    const int q = chunk * 256 + threadIdx.x;
    const bool on = q < X4;
    const int qc = on ? q : X4 - 1;
    const int r0 = rb * kNllCsRows, r1 = r0 + kNllCsRows < a.M ? r0 + kNllCsRows : a.M;
    const float4 bm = reinterpret_cast<const float4*>(a.bias_mu)[qc], bs = reinterpret_cast<const float4*>(a.bias_sd)[qc];
    float4 cm = make_float4(0.f, 0.f, 0.f, 0.f), cs = make_float4(0.f, 0.f, 0.f, 0.f);
    // two rows per trip: six 16-B loads in flight per lane (the loop is a chain of memory round trips otherwise)
    for (int r = r0; r < r1; r += 2) {
        const bool two = r + 1 < r1;
        const int ra = r, rbb = two ? r + 1 : r;
        const float c0 = a.coef[ra], c1 = a.coef[rbb];
        const float4 x0 = reinterpret_cast<const float4*>(a.x + (int64_t)(a.xidx ? a.xidx[ra] : ra) * a.ldx)[qc];
        const float4 x1 = reinterpret_cast<const float4*>(a.x + (int64_t)(a.xidx ? a.xidx[rbb] : rbb) * a.ldx)[qc];
        const nt_f4 m0 = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(a.mu + (int64_t)ra * a.ldp) + qc);
        const nt_f4 s0 = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(a.sd + (int64_t)ra * a.ldp) + qc);
        const nt_f4 m1 = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(a.mu + (int64_t)rbb * a.ldp) + qc);
        const nt_f4 s1 = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(a.sd + (int64_t)rbb * a.ldp) + qc);
        float acc0 = 0.f, acc1 = 0.f;
        nt_f4 gm0, gs0, gm1, gs1;
        float u, v;
        nll_raw_sp_elem(c0, a.shift, x0.x, m0[0], s0[0], bm.x, bs.x, acc0, u, v); gm0[0] = u; gs0[0] = v;
        nll_raw_sp_elem(c0, a.shift, x0.y, m0[1], s0[1], bm.y, bs.y, acc0, u, v); gm0[1] = u; gs0[1] = v;
        nll_raw_sp_elem(c0, a.shift, x0.z, m0[2], s0[2], bm.z, bs.z, acc0, u, v); gm0[2] = u; gs0[2] = v;
        nll_raw_sp_elem(c0, a.shift, x0.w, m0[3], s0[3], bm.w, bs.w, acc0, u, v); gm0[3] = u; gs0[3] = v;
        nll_raw_sp_elem(c1, a.shift, x1.x, m1[0], s1[0], bm.x, bs.x, acc1, u, v); gm1[0] = u; gs1[0] = v;
        nll_raw_sp_elem(c1, a.shift, x1.y, m1[1], s1[1], bm.y, bs.y, acc1, u, v); gm1[1] = u; gs1[1] = v;
        nll_raw_sp_elem(c1, a.shift, x1.z, m1[2], s1[2], bm.z, bs.z, acc1, u, v); gm1[2] = u; gs1[2] = v;
        nll_raw_sp_elem(c1, a.shift, x1.w, m1[3], s1[3], bm.w, bs.w, acc1, u, v); gm1[3] = u; gs1[3] = v;
        if constexpr (FWD_ONLY) {         // (evaluation: the row terms only -- no gradients, no column sums)
            if (!on) acc0 = acc1 = 0.f;
        } else if (on) {
            __builtin_nontemporal_store(gm0, reinterpret_cast<nt_f4*>(a.dmu + (int64_t)ra * a.ldd) + q);
            __builtin_nontemporal_store(gs0, reinterpret_cast<nt_f4*>(a.dsd + (int64_t)ra * a.ldd) + q);
            // (fixed order: row r, then row r + 1 -- the same sums whatever the grid)
            cm.x += gm0[0]; cm.y += gm0[1]; cm.z += gm0[2]; cm.w += gm0[3];
            cs.x += gs0[0]; cs.y += gs0[1]; cs.z += gs0[2]; cs.w += gs0[3];
            if (two) {
                __builtin_nontemporal_store(gm1, reinterpret_cast<nt_f4*>(a.dmu + (int64_t)rbb * a.ldd) + q);
                __builtin_nontemporal_store(gs1, reinterpret_cast<nt_f4*>(a.dsd + (int64_t)rbb * a.ldd) + q);
                cm.x += gm1[0]; cm.y += gm1[1]; cm.z += gm1[2]; cm.w += gm1[3];
                cs.x += gs1[0]; cs.y += gs1[1]; cs.z += gs1[2]; cs.w += gs1[3];
            }
        } else {
            acc0 = acc1 = 0.f;
        }
        acc0 = dv_wave_sum_all(acc0);
        acc1 = dv_wave_sum_all(acc1);
        if (lane == 0) {
            part[wave][ra - r0] = acc0;
            if (two) part[wave][rbb - r0] = acc1;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < r1 - r0) {
        const int t = threadIdx.x;
        a.out_part[(int64_t)(r0 + t) * a.chunks + chunk] = -0.5f * ((part[0][t] + part[1][t]) + (part[2][t] + part[3][t]));
    }
    if constexpr (!FWD_ONLY) {
        if (on) {
            float* w = a.ws + (int64_t)rb * a.ldw;
            reinterpret_cast<float4*>(w)[q] = cm;
            reinterpret_cast<float4*>(w + a.sd_off)[q] = cs;
        }
    }
}

// Bernoulli / Poisson reconstruction rows (type_rec = 'binary' / 'poisson': the decoders src/DrVAE.py:124-129 names;
// the reference ships neither class -- labelled extension).  v = the head's POST-activation output
// (probability = sigmoid(a), or rate = softplus(a) + shift); out[r] = sum_g log p(x|v); with coef != NULL also
// dpre[r,g] = coef[r] * d log p / d a (the gradient w.r.t. the head's pre-activation a).
//   Bernoulli: pc = clamp(v, 1e-10, 1 - 1e-10) (the clamp of src/blocks.py:463 applied to this decoder's probabilities);
//              log p = x log pc + (1-x) log(1-pc);  d/da = x - v inside the clamp, 0 outside
//   Poisson  : log p = x log v - v - lgamma(x+1);   d/da = (x/v - 1) * (1 - exp(-(v - shift)))
__device__ __forceinline__ float rec_term(int kind, float shift, float xv, float v, float& g) {
    if (kind == DV_REC_BERNOULLI) {
        const float lo = 1e-10f, hi = (float)(1.0 - 1e-10);
        const float pc = fminf(fmaxf(v, lo), hi);
        g = (v > lo && v < hi) ? xv - v : 0.f;
        return xv * logf(pc) + (1.f - xv) * logf(1.f - pc);
    }
    g = (xv / v - 1.f) * (1.f - expf(-(v - shift)));
    return xv * logf(v) - v - lgammaf(xv + 1.f);
}

__global__ __launch_bounds__(256) void rec_nll_rows_kernel(int kind, float shift, const float* __restrict__ coef,
                                                           const float* __restrict__ x, int64_t ldx,
                                                           const int32_t* __restrict__ xidx,
                                                           const float* __restrict__ v, int64_t ldv, int M, int X,
                                                           float* __restrict__ out, float* __restrict__ dpre,
                                                           int64_t ldd) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x; r < M; r += gridDim.x) {
        const float* xr = x + (int64_t)(xidx ? xidx[r] : r) * ldx;
        const float* vr = v + (int64_t)r * ldv;
        const float c = coef ? coef[r] : 0.f;
        float acc = 0.f;
        for (int g = threadIdx.x; g < X; g += 256) {
            float gr;
            acc += rec_term(kind, shift, xr[g], vr[g], gr);
            if (coef) dpre[(int64_t)r * ldd + g] = c * gr;
        }
        acc = dv_wave_sum_all(acc);
        if (lane == 0) part[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) out[r] = (part[0] + part[1]) + (part[2] + part[3]);
        __syncthreads();
    }
}

__global__ void nll_rows_bwd_kernel(const float* __restrict__ coef, const float* __restrict__ x, int64_t ldx,
                                    const int32_t* __restrict__ xidx, const float* __restrict__ mu,
                                    const float* __restrict__ sd, int64_t ldp, int M, int X, int mode, int sd_act,
                                    float sd_shift, float* __restrict__ dmu, float* __restrict__ dsd, int64_t ldd,
                                    float* __restrict__ dx, int64_t lddx, float beta) {
    const int64_t total = (int64_t)M * X;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / X), g = (int)(e % X);
        const float c = coef[r];
        const float xv = x[(int64_t)(xidx ? xidx[r] : r) * ldx + g];
        const float m = mu[(int64_t)r * ldp + g], s = sd[(int64_t)r * ldp + g];
        const float d = xv - m;
        float gm, gs;
        if (mode == DV_GAUSS_SIGMA) {
            const float v = s * s;
            gm = d / v;
            gs = -1.f / s + d * d / (v * s);
        } else {
            const float iv = expf(-s);
            gm = d * iv;
            gs = -0.5f * (1.f - d * d * iv);
        }
        if (sd_act != DV_ACT_IDENTITY) gs *= dv_dact_from_y(sd_act, s - sd_shift);
        const int64_t o = (int64_t)r * ldd + g;
        dmu[o] = (beta != 0.f ? beta * dmu[o] : 0.f) + c * gm;
        dsd[o] = (beta != 0.f ? beta * dsd[o] : 0.f) + c * gs;
        if (dx) {
            const int64_t ox = (int64_t)r * lddx + g;
            dx[ox] = (beta != 0.f ? beta * dx[ox] : 0.f) - c * gm;
        }
    }
}

// ------------------------------------------------------------------- categorical
constexpr float kPMin = 1e-10f;
constexpr float kPMax = 1.f - 1e-10f;   // == 1.0f in fp32, as in the reference's fp32 clamp

__global__ void softmax_clamp_fwd_kernel(const float* __restrict__ logits, int64_t ldl, int M, int Y, int sigmoid1,
                                         float* __restrict__ probs, int64_t ldp) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const float* a = logits + (int64_t)r * ldl;
    float* p = probs + (int64_t)r * ldp;
    if (sigmoid1) {
        const float s = 1.f / (1.f + expf(-a[0]));
        p[0] = fminf(fmaxf(1.f - s, kPMin), kPMax);
        p[1] = fminf(fmaxf(s, kPMin), kPMax);
        return;
    }
    float mx = a[0];
    for (int j = 1; j < Y; ++j) mx = fmaxf(mx, a[j]);
    float den = 0.f;
    for (int j = 0; j < Y; ++j) den += expf(a[j] - mx);
    for (int j = 0; j < Y; ++j) p[j] = fminf(fmaxf(expf(a[j] - mx) / den, kPMin), kPMax);
}

__global__ void softmax_clamp_bwd_kernel(const float* __restrict__ dprobs, int64_t lddp,
                                         const float* __restrict__ probs, int64_t ldp, int M, int Y, int sigmoid1,
                                         float* __restrict__ dlogits, int64_t ldl, float beta) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const float* g = dprobs + (int64_t)r * lddp;
    const float* p = probs + (int64_t)r * ldp;
    float* o = dlogits + (int64_t)r * ldl;
    // clamp passes the gradient only where the unclamped value lies inside [1e-10, 1]
    if (sigmoid1) {
        const float s = p[1], g1 = p[1] > kPMin ? g[1] : 0.f, g0 = p[0] > kPMin ? g[0] : 0.f;
        const float v = (g1 - g0) * s * (1.f - s);
        o[0] = (beta != 0.f ? beta * o[0] : 0.f) + v;
        return;
    }
    float dot = 0.f;
    for (int j = 0; j < Y; ++j) dot += (p[j] > kPMin ? g[j] : 0.f) * p[j];
    for (int j = 0; j < Y; ++j) {
        const float v = p[j] * ((p[j] > kPMin ? g[j] : 0.f) - dot);
        o[j] = (beta != 0.f ? beta * o[j] : 0.f) + v;
    }
}

__global__ void cat_terms_fwd_kernel(const float* __restrict__ probs, int64_t ldp, int M, int Y,
                                     const int32_t* __restrict__ labels, const float* __restrict__ prior,
                                     int64_t ldpr, float* __restrict__ logp, float* __restrict__ kl, int64_t ldk,
                                     float* __restrict__ ent, int32_t* __restrict__ best) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const float* p = probs + (int64_t)r * ldp;
    if (logp) logp[r] = logf(p[labels[r]]);
    float e = 0.f, bv = p[0];
    int bi = 0;
    for (int j = 0; j < Y; ++j) {
        const float lp = logf(p[j]);
        if (kl) kl[(int64_t)r * ldk + j] = -p[j] * (logf(prior[(int64_t)r * ldpr + j]) - lp);
        e += p[j] * lp;
        if (p[j] > bv) {
            bv = p[j];
            bi = j;
        }
    }
    if (ent) ent[r] = -e;
    if (best) best[r] = bi;
}

__global__ void cat_terms_bwd_kernel(const float* __restrict__ probs, int64_t ldp, int M, int Y,
                                     const int32_t* __restrict__ labels, const float* __restrict__ prior,
                                     int64_t ldpr, const float* __restrict__ c_logp, const float* __restrict__ g_kl,
                                     int64_t ldg, const float* __restrict__ c_ent, float* __restrict__ dprobs,
                                     int64_t lddp, float beta) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const float* p = probs + (int64_t)r * ldp;
    for (int j = 0; j < Y; ++j) {
        const float lp = logf(p[j]);
        float v = 0.f;
        if (c_logp && labels[r] == j) v += c_logp[r] / p[j];
        if (g_kl) v += g_kl[(int64_t)r * ldg + j] * (lp - logf(prior[(int64_t)r * ldpr + j]) + 1.f);
        if (c_ent) v -= c_ent[r] * (lp + 1.f);
        float* o = dprobs + (int64_t)r * lddp + j;
        *o = (beta != 0.f ? beta * *o : 0.f) + v;
    }
}

// ------------------------------------------------- small-N linear head (the classifier)
// q(y|.) = clamp(softmax([a1|a2] W^T + b)) with N = dim_y <= 8 outputs: an MFMA tile would be
// >90 % padding, so one wavefront per row does the N dot products, the softmax and the clamp.
constexpr int kMaxSmallN = 8;

// y-marginalisation of one row right behind its class probabilities (dv_smalln_linear_fwd with a dv_ymarg argument:
// the classifier head and the labeled / marginalised KLD assembly of src/DrVAE.py:503-534 in one launch); same
// arithmetic as ymarg_fwdbwd_kernel below
// (cfp_out != NULL: the coefficients written to y.cfp are also returned, slot by slot)
__device__ __forceinline__ void ymarg_row(const dv_ymarg& y, int r, int Y, const float* q, float* cfp_out = nullptr) {
    const int f0 = y.fp_ptr[r], nf = y.fp_ptr[r + 1] - f0;
    const float ck = y.c_kld[r];
    float* dq = y.dqy + (int64_t)r * y.lddq;
    const int lab0 = y.label[r];
    if (nf == 1 || lab0 <= -2) {
        const int lab = nf == 1 ? lab0 : -2 - lab0;
        y.yl[r] = logf(q[lab]);
        y.kld[r] = y.klfp[nf == 1 ? f0 : f0 + lab];
        for (int j = 0; j < Y; ++j) dq[j] = (j == lab) ? y.c_yl[r] / q[j] : 0.f;
        if (nf == 1) {
            y.cfp[f0] = ck;
            if (cfp_out) cfp_out[0] = ck;
        } else {
            for (int j = 0; j < Y; ++j) {
                y.cfp[f0 + j] = (j == lab) ? ck : 0.f;
                if (cfp_out) cfp_out[j] = (j == lab) ? ck : 0.f;
            }
        }
    } else {
        float a = 0.f, b = 0.f;
        for (int j = 0; j < Y; ++j) {
            const float lp = y.log_prior_v ? y.log_prior_v[j] : y.log_prior, lq = logf(q[j]), kf = y.klfp[f0 + j];
            a += q[j] * kf;
            b += -q[j] * (lp - lq);
            y.cfp[f0 + j] = ck * q[j];
            if (cfp_out) cfp_out[j] = ck * q[j];
            dq[j] = ck * (kf + lq - lp + 1.f);
        }
        y.yl[r] = 0.f;
        y.kld[r] = a + b;
    }
}

__global__ __launch_bounds__(256) void smalln_fwd_kernel(const float* __restrict__ a1, int64_t lda1, int K1,
                                                         const float* __restrict__ a2, int64_t lda2, int K2,
                                                         const float* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ bias, int M, int N,
                                                         float* __restrict__ logits, int64_t ldl,
                                                         float* __restrict__ probs, int64_t ldp, dv_ymarg ym,
                                                         ParkArgs park, dv_fprop_kl kf) {
    park_block(park);
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    // (fprop rows of this classifier row: their KL terms first -- the y-marginalisation below consumes them)
    float raw1[kMaxSmallN];
    int f0 = 0, nf = 0;
    if (kf.mu_q != nullptr) {
        f0 = ym.fp_ptr[r];
        nf = ym.fp_ptr[r + 1] - f0;
#pragma unroll
        for (int u = 0; u < kMaxSmallN; ++u) {
            if (u >= nf) break;
            const int t = f0 + u;
            const float* q = kf.mu_q + (int64_t)kf.qidx[t] * kf.ldq;
            const float* pp = kf.mu_p + (int64_t)t * kf.ldp;
            const float* q3 = kf.mu3 + (int64_t)t * kf.ld3;
            float s1 = 0.f, s3 = 0.f;
            for (int d = lane; d < kf.Z1; d += 64) s1 += kl_term(DV_GAUSS_LOGVAR, q[d], q[kf.Z1 + d], pp[d], pp[kf.Z1 + d]);
            for (int d = lane; d < kf.Z3; d += 64) s3 += kl_term(DV_GAUSS_LOGVAR, q3[d], q3[kf.Z3 + d], 0.f, 0.f);
            s1 = dv_wave_sum_all(s1);
            s3 = dv_wave_sum_all(s3);
            raw1[u] = -0.5f * s1;
            if (lane == 0) {
                kf.raw1[t] = raw1[u];
                kf.raw3[t] = -0.5f * s3;
                kf.klfp[t] = fmaxf(raw1[u], kf.kl_min) + fmaxf(-0.5f * s3, kf.kl_min);
            }
        }
    }
    float acc[kMaxSmallN];
#pragma unroll
    for (int j = 0; j < kMaxSmallN; ++j) acc[j] = 0.f;
    for (int k = lane; k < K1 + K2; k += 64) {
        const float x = k < K1 ? a1[(int64_t)r * lda1 + k] : a2[(int64_t)r * lda2 + (k - K1)];
#pragma unroll
        for (int j = 0; j < kMaxSmallN; ++j)
            if (j < N) acc[j] += x * W[(int64_t)j * ldw + k];
    }
#pragma unroll
    for (int j = 0; j < kMaxSmallN; ++j) acc[j] = dv_wave_sum_all(acc[j]);
    if (lane == 0) {
        float mx = -3.4e38f;
        for (int j = 0; j < N; ++j) {
            acc[j] += bias ? bias[j] : 0.f;
            if (logits) logits[(int64_t)r * ldl + j] = acc[j];
            mx = fmaxf(mx, acc[j]);
        }
        if (probs) {
            float den = 0.f;
            for (int j = 0; j < N; ++j) den += expf(acc[j] - mx);
            float q[kMaxSmallN];
            for (int j = 0; j < N; ++j) {
                q[j] = fminf(fmaxf(expf(acc[j] - mx) / den, kPMin), kPMax);
                probs[(int64_t)r * ldp + j] = q[j];
            }
            if (ym.fp_ptr != nullptr) ymarg_row(ym, r, N, q, kf.mu_q != nullptr ? acc : nullptr);   // (acc: reused for cfp)
        }
    }
    if (kf.mu_q != nullptr) {
        // backward of the z1 term with the coefficients of the y-marginalisation (lane 0 holds them): dv_kl_rows_bwd
#pragma unroll
        for (int u = 0; u < kMaxSmallN; ++u) {
            if (u >= nf) break;
            const int t = f0 + u;
            float c = __shfl(acc[u], 0, 64);
            c *= raw1[u] > kf.kl_min ? 1.f : (raw1[u] == kf.kl_min ? 0.5f : 0.f);
            const float* q = kf.mu_q + (int64_t)kf.qidx[t] * kf.ldq;
            const float* pp = kf.mu_p + (int64_t)t * kf.ldp;
            float* dq = kf.dq + (int64_t)t * kf.lddq;
            float* dp = kf.dp + (int64_t)t * kf.lddp;
            for (int d = lane; d < kf.Z1; d += 64) {
                const float mq = q[d], sq = q[kf.Z1 + d], mp = pp[d], sp = pp[kf.Z1 + d];
                const float dm = mq - mp, ivp = expf(-sp), vq = expf(sq);
                const float gmq = dm * ivp, gsq = -0.5f * (1.f - vq * ivp);
                dq[d] = c * gmq;
                dq[kf.Z1 + d] = c * gsq;
                dp[d] = c * -gmq;
                dp[kf.Z1 + d] = c * (-0.5f * (-1.f + (dm * dm + vq) * ivp));
            }
        }
    }
}

// The same launch for the train step's common case -- two classes, [z1 | z2F - z1] of <= 256 columns, latent widths <= 128 --
// with every operand in flight before the first result is formed (round 5).  The generic kernel above walks
// [fp_ptr -> qidx -> KL rows -> reduce] per fprop row, then the classifier's operands, then re-reads the KL rows for the
// backward: seven dependent round trips on the side chain's critical path (10.9 us isolated); here three --
// [classifier operands, fp_ptr] -> [qidx] -> [all KL rows] -- and the backward works from registers.  Same summation orders
// (the results agree to the rounding of contracted multiply-adds).  cfg 2: 10.9 -> 8.1 us isolated, and -- the side chain in
// front of the join is the step's critical path -- 0.1906 -> 0.1863 ms per step.
__global__ __launch_bounds__(256) void smalln_fwd2_kernel(const float* __restrict__ a1, int64_t lda1, int K1,
                                                          const float* __restrict__ a2, int64_t lda2, int K2,
                                                          const float* __restrict__ W, int64_t ldw,
                                                          const float* __restrict__ bias, int M, int N,
                                                          float* __restrict__ logits, int64_t ldl,
                                                          float* __restrict__ probs, int64_t ldp, dv_ymarg ym,
                                                          ParkArgs park, dv_fprop_kl kf) {
    park_block(park);
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const int Kt = K1 + K2, Z1 = kf.Z1, Z3 = kf.Z3;
    // ---- round trip 1: the classifier's operands and the row's fprop range
    float xs[4], w0[4], w1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = lane + 64 * c;
        xs[c] = w0[c] = w1[c] = 0.f;
        if (k < Kt) {
            xs[c] = k < K1 ? a1[(int64_t)r * lda1 + k] : a2[(int64_t)r * lda2 + (k - K1)];
            w0[c] = W[k];
            if (N > 1) w1[c] = W[ldw + k];
        }
    }
    const int f0 = ym.fp_ptr[r];
    int nf = ym.fp_ptr[r + 1] - f0;
    nf = nf < 2 ? nf : 2;                       // (one fprop row for a labeled row, one per class otherwise; N <= 2)
    // ---- round trip 2: which q row each fprop row reads
    int qi[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (u < nf) qi[u] = kf.qidx[f0 + u];
    // ---- round trip 3: the KL rows
    float qm[2][2], ql[2][2], pm[2][2], pl[2][2], tm[2][2], tl[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int d = lane + 64 * c;
            qm[u][c] = ql[u][c] = pm[u][c] = pl[u][c] = tm[u][c] = tl[u][c] = 0.f;
            if (u < nf) {
                const int t = f0 + u;
                if (d < Z1) {
                    const float* q = kf.mu_q + (int64_t)qi[u] * kf.ldq;
                    const float* pp = kf.mu_p + (int64_t)t * kf.ldp;
                    qm[u][c] = q[d]; ql[u][c] = q[Z1 + d]; pm[u][c] = pp[d]; pl[u][c] = pp[Z1 + d];
                }
                if (d < Z3) {
                    const float* q3 = kf.mu3 + (int64_t)t * kf.ld3;
                    tm[u][c] = q3[d]; tl[u][c] = q3[Z3 + d];
                }
            }
        }
    // ---- the fprop rows' KL terms
    float raw1[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (u >= nf) break;
        float s1 = 0.f, s3 = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (lane + 64 * c < Z1) s1 += kl_term(DV_GAUSS_LOGVAR, qm[u][c], ql[u][c], pm[u][c], pl[u][c]);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (lane + 64 * c < Z3) s3 += kl_term(DV_GAUSS_LOGVAR, tm[u][c], tl[u][c], 0.f, 0.f);
        }
        s1 = dv_wave_sum_all(s1);
        s3 = dv_wave_sum_all(s3);
        raw1[u] = -0.5f * s1;
        if (lane == 0) {
            const int t = f0 + u;
            kf.raw1[t] = raw1[u];
            kf.raw3[t] = -0.5f * s3;
            kf.klfp[t] = fmaxf(raw1[u], kf.kl_min) + fmaxf(-0.5f * s3, kf.kl_min);
        }
    }
    // ---- the classifier and the y-marginalisation
    float acc[kMaxSmallN];
#pragma unroll
    for (int j = 0; j < kMaxSmallN; ++j) acc[j] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (lane + 64 * c < Kt) {
            acc[0] += xs[c] * w0[c];
            if (N > 1) acc[1] += xs[c] * w1[c];
        }
    acc[0] = dv_wave_sum_all(acc[0]);
    acc[1] = dv_wave_sum_all(acc[1]);
    if (lane == 0) {
        float mx = -3.4e38f;
        for (int j = 0; j < N; ++j) {
            acc[j] += bias ? bias[j] : 0.f;
            if (logits) logits[(int64_t)r * ldl + j] = acc[j];
            mx = fmaxf(mx, acc[j]);
        }
        float den = 0.f;
        for (int j = 0; j < N; ++j) den += expf(acc[j] - mx);
        float q[kMaxSmallN];
        for (int j = 0; j < N; ++j) {
            q[j] = fminf(fmaxf(expf(acc[j] - mx) / den, kPMin), kPMax);
            probs[(int64_t)r * ldp + j] = q[j];
        }
        ymarg_row(ym, r, N, q, acc);         // (acc: reused for the fprop rows' coefficients)
    }
    // ---- backward of the z1 term with those coefficients (lane 0 holds them), from the registers
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (u >= nf) break;
        const int t = f0 + u;
        float c = __shfl(acc[u], 0, 64);
        c *= raw1[u] > kf.kl_min ? 1.f : (raw1[u] == kf.kl_min ? 0.5f : 0.f);
        float* dq = kf.dq + (int64_t)t * kf.lddq;
        float* dp = kf.dp + (int64_t)t * kf.lddp;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int d = lane + 64 * cc;
            if (d < Z1) {
                const float mq = qm[u][cc], sq = ql[u][cc], mp = pm[u][cc], sp = pl[u][cc];
                const float dm = mq - mp, ivp = expf(-sp), vq = expf(sq);
                const float gmq = dm * ivp, gsq = -0.5f * (1.f - vq * ivp);
                dq[d] = c * gmq;
                dq[Z1 + d] = c * gsq;
                dp[d] = c * -gmq;
                dp[Z1 + d] = c * (-0.5f * (-1.f + (dm * dm + vq) * ivp));
            }
        }
    }
}

// d logits of row r from (d probs, probs) through clamp + softmax, see softmax_clamp_bwd_kernel
__device__ __forceinline__ void smalln_dlogits(const float* g, const float* p, int N, float* dl) {
    float dot = 0.f;
    for (int j = 0; j < N; ++j) dot += (p[j] > kPMin ? g[j] : 0.f) * p[j];
    for (int j = 0; j < N; ++j) dl[j] = p[j] * ((p[j] > kPMin ? g[j] : 0.f) - dot);
}

struct SmallNDst {
    float* dst[3];
    int64_t ld[3];
    int col0[3], ncol[3];     // columns [col0, col0+ncol) of W feed this destination
    int col1[3];              // optional second column block (alpha2 != 0): `[z1, z2F - z1]` inputs
    float alpha[3], alpha2[3], beta[3];
    int n;
    // destination 0 may start from a segment sum instead of beta*dst: sum of rows [seg_ptr[r], seg_ptr[r+1]) of seg_src
    const float* seg_src;
    int64_t ld_seg;
    const int32_t* seg_ptr;
};

// dst_t[r, c] = beta*dst_t[r, c] + sum_j dlogit[r,j] * (alpha*W[j, col0_t + c] + alpha2*W[j, col1_t + c])
__global__ void smalln_bwd_data_kernel(const float* __restrict__ dprobs, int64_t lddp,
                                       const float* __restrict__ probs, int64_t ldp, int from_probs,
                                       const float* __restrict__ W, int64_t ldw, int M, int N, SmallNDst d) {
    int total_cols = 0;
    for (int t = 0; t < d.n; ++t) total_cols += d.ncol[t];
    const int64_t total = (int64_t)M * total_cols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / total_cols);
        int c = (int)(e % total_cols), t = 0;
        while (c >= d.ncol[t]) {
            c -= d.ncol[t];
            ++t;
        }
        float dl[kMaxSmallN];
        if (from_probs) {
            smalln_dlogits(dprobs + (int64_t)r * lddp, probs + (int64_t)r * ldp, N, dl);
        } else {
            for (int j = 0; j < N; ++j) dl[j] = dprobs[(int64_t)r * lddp + j];
        }
        float s = 0.f, s2 = 0.f;
        for (int j = 0; j < N; ++j) s += dl[j] * W[(int64_t)j * ldw + d.col0[t] + c];
        if (d.alpha2[t] != 0.f)
            for (int j = 0; j < N; ++j) s2 += dl[j] * W[(int64_t)j * ldw + d.col1[t] + c];
        float* o = d.dst[t] + (int64_t)r * d.ld[t] + c;
        float base;
        if (t == 0 && d.seg_src != nullptr) {      // (dv_rows_segment_sum's sum, in its order)
            base = 0.f;
            for (int u = d.seg_ptr[r]; u < d.seg_ptr[r + 1]; ++u) base += d.seg_src[(int64_t)u * d.ld_seg + c];
        } else {
            base = d.beta[t] != 0.f ? d.beta[t] * *o : 0.f;
        }
        *o = base + d.alpha[t] * s + d.alpha2[t] * s2;
    }
}

// dW[j, k] = beta*dW + sum_r dlogit[r,j] * [a1|a2][r,k];  db[j] = beta*db + sum_r dlogit[r,j]
// block = 16 columns x 64 row-groups (1024 threads): every thread walks only M/64 rows, so the
// chain of dependent global loads is short; fixed-order LDS tree over the row-groups (deterministic)
constexpr int kSnCols = 16, kSnRG = 64;
// gridDim.y > 1 (round 5): the rows are split over gridDim.y workgroups per column block -- 26 workgroups of a 256-CU chip
// walked all 4096 rows of the wide configuration's classifier (85-113 us for 6 MB of input) -- each writes its partial
// sums to ws[split][N][KT + 1]; smalln_wgrad_reduce_kernel adds the splits up in a fixed order (deterministic)
__global__ __launch_bounds__(1024) void smalln_bwd_weight_kernel(const float* __restrict__ dprobs, int64_t lddp,
                                                                 const float* __restrict__ probs, int64_t ldp,
                                                                 int from_probs, const float* __restrict__ a1,
                                                                 int64_t lda1, int K1, const float* __restrict__ a2,
                                                                 int64_t lda2, int K2, int M, int N,
                                                                 float* __restrict__ dW, int64_t ldd,
                                                                 float* __restrict__ db, float beta, dv_publish pub,
                                                                 float* __restrict__ ws, int rows_per_split) {
    // (one spare class slot: a row-group stride of 144 floats = 16 banks, so the two row groups of a 32-lane access sit on
    // disjoint banks -- with 128 they shared them: 0.49 LDS bank conflicts per access in the round-5 counters)
    __shared__ float part[kSnRG][kMaxSmallN + 1][kSnCols];
    publish_block0(pub);
    const int c = threadIdx.x % kSnCols, rg = threadIdx.x / kSnCols;
    const int k = blockIdx.x * kSnCols + c, KT = K1 + K2;   // k == KT is the bias column
    const int m0 = blockIdx.y * rows_per_split, m1 = min(M, m0 + rows_per_split);
    float acc[kMaxSmallN];
#pragma unroll
    for (int j = 0; j < kMaxSmallN; ++j) acc[j] = 0.f;
    if (k <= KT && m1 > m0) {
        // four rows per trip with their loads issued together (rows past the end read the last row and are masked out):
        // the kernel is a chain of dependent-load round trips, so fewer trips is what shortens it
        const float* xs = k < K1 ? a1 + k : a2 + (k - K1);
        const int64_t ldx = k < K1 ? lda1 : lda2;
        for (int r0 = m0 + rg; r0 < m1; r0 += 4 * kSnRG) {
            float x[4], dl[4][kMaxSmallN];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + u * kSnRG, rc = r < m1 ? r : m1 - 1;
                x[u] = r < m1 ? (k == KT ? 1.f : xs[(int64_t)rc * ldx]) : 0.f;
                if (from_probs) {
                    smalln_dlogits(dprobs + (int64_t)rc * lddp, probs + (int64_t)rc * ldp, N, dl[u]);
                } else {
                    for (int j = 0; j < N; ++j) dl[u][j] = dprobs[(int64_t)rc * lddp + j];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < kMaxSmallN; ++j)
                    if (j < N) acc[j] += dl[u][j] * x[u];
        }
    }
#pragma unroll
    for (int j = 0; j < kMaxSmallN; ++j) part[rg][j][c] = acc[j];
    __syncthreads();
    for (int h = kSnRG / 2; h >= 1; h >>= 1) {
        if (rg < h)
            for (int j = 0; j < N; ++j) part[rg][j][c] += part[rg + h][j][c];
        __syncthreads();
    }
    if (rg == 0 && k <= KT) {
        for (int j = 0; j < N; ++j) {
            if (ws) {
                ws[((int64_t)blockIdx.y * N + j) * (KT + 1) + k] = part[0][j][c];
                continue;
            }
            float* o = k == KT ? (db ? db + j : nullptr) : dW + (int64_t)j * ldd + k;
            if (o) *o = (beta != 0.f ? beta * *o : 0.f) + part[0][j][c];
        }
    }
}

__global__ __launch_bounds__(256) void smalln_wgrad_reduce_kernel(const float* __restrict__ ws, int splits, int N, int KT,
                                                                  float* __restrict__ dW, int64_t ldd, float* __restrict__ db,
                                                                  float beta) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= N * (KT + 1)) return;
    const int j = e / (KT + 1), k = e % (KT + 1);
    float s_ = 0.f;
    for (int t = 0; t < splits; ++t) s_ += ws[((int64_t)t * N + j) * (KT + 1) + k];
    float* o = k == KT ? (db ? db + j : nullptr) : dW + (int64_t)j * ldd + k;
    if (o) *o = (beta != 0.f ? beta * *o : 0.f) + s_;
}

// ------------------------------------------------------------- y-marginalisation
__global__ void ymarg_fwd_kernel(const float* __restrict__ qy, int64_t ldq, const int32_t* __restrict__ label,
                                 const int32_t* __restrict__ fp_ptr, const float* __restrict__ klfp,
                                 float log_prior, const float* __restrict__ log_prior_v, int R, int Y,
                                 float* __restrict__ yl, float* __restrict__ kld) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* q = qy + (int64_t)r * ldq;
    const int f0 = fp_ptr[r], nf = fp_ptr[r + 1] - f0;
    if (nf == 1) {
        yl[r] = logf(q[label[r]]);
        kld[r] = klfp[f0];
    } else if (label[r] <= -2) {
        // labeled row of a batch-independent ("universal") plan: all Y class slots are materialised, the
        // true class c = -2 - label[r] is the one that counts (see dv_batch_masks)
        const int lab = -2 - label[r];
        yl[r] = logf(q[lab]);
        kld[r] = klfp[f0 + lab];
    } else {
        float a = 0.f, b = 0.f;
        for (int j = 0; j < Y; ++j) {
            a += q[j] * klfp[f0 + j];
            b += -q[j] * ((log_prior_v ? log_prior_v[j] : log_prior) - logf(q[j]));
        }
        yl[r] = 0.f;
        kld[r] = a + b;
    }
}

__global__ void ymarg_bwd_kernel(const float* __restrict__ qy, int64_t ldq, const int32_t* __restrict__ label,
                                 const int32_t* __restrict__ fp_ptr, const float* __restrict__ klfp,
                                 float log_prior, const float* __restrict__ log_prior_v,
                                 const float* __restrict__ c_kld, const float* __restrict__ c_yl, int R, int Y,
                                 float* __restrict__ cfp, float* __restrict__ dqy, int64_t lddq) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* q = qy + (int64_t)r * ldq;
    float* dq = dqy + (int64_t)r * lddq;
    const int f0 = fp_ptr[r], nf = fp_ptr[r + 1] - f0;
    const float ck = c_kld[r];
    if (nf == 1) {
        const int lab = label[r];
        for (int j = 0; j < Y; ++j) dq[j] = (j == lab) ? c_yl[r] / q[j] : 0.f;
        cfp[f0] = ck;
    } else if (label[r] <= -2) {
        const int lab = -2 - label[r];
        for (int j = 0; j < Y; ++j) {
            dq[j] = (j == lab) ? c_yl[r] / q[j] : 0.f;
            cfp[f0 + j] = (j == lab) ? ck : 0.f;
        }
    } else {
        for (int j = 0; j < Y; ++j) {
            cfp[f0 + j] = ck * q[j];
            dq[j] = ck * (klfp[f0 + j] + logf(q[j]) - (log_prior_v ? log_prior_v[j] : log_prior) + 1.f);
        }
    }
}

// forward and backward of the y-marginalisation in one pass (train step: the coefficients are known)
__global__ void ymarg_fwdbwd_kernel(const float* __restrict__ qy, int64_t ldq, const int32_t* __restrict__ label,
                                    const int32_t* __restrict__ fp_ptr, const float* __restrict__ klfp,
                                    float log_prior, const float* __restrict__ log_prior_v,
                                    const float* __restrict__ c_kld, const float* __restrict__ c_yl, int R, int Y,
                                    float* __restrict__ yl, float* __restrict__ kld, float* __restrict__ cfp,
                                    float* __restrict__ dqy, int64_t lddq) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* q = qy + (int64_t)r * ldq;
    float* dq = dqy + (int64_t)r * lddq;
    const int f0 = fp_ptr[r], nf = fp_ptr[r + 1] - f0;
    const float ck = c_kld[r];
    if (nf == 1) {
        const int lab = label[r];
        yl[r] = logf(q[lab]);
        kld[r] = klfp[f0];
        for (int j = 0; j < Y; ++j) dq[j] = (j == lab) ? c_yl[r] / q[j] : 0.f;
        cfp[f0] = ck;
    } else if (label[r] <= -2) {
        const int lab = -2 - label[r];
        yl[r] = logf(q[lab]);
        kld[r] = klfp[f0 + lab];
        for (int j = 0; j < Y; ++j) {
            dq[j] = (j == lab) ? c_yl[r] / q[j] : 0.f;
            cfp[f0 + j] = (j == lab) ? ck : 0.f;
        }
    } else {
        float a = 0.f, b = 0.f;
        for (int j = 0; j < Y; ++j) {
            const float lp = log_prior_v ? log_prior_v[j] : log_prior, lq = logf(q[j]), kf = klfp[f0 + j];
            a += q[j] * kf;
            b += -q[j] * (lp - lq);
            cfp[f0 + j] = ck * q[j];
            dq[j] = ck * (kf + lq - lp + 1.f);
        }
        yl[r] = 0.f;
        kld[r] = a + b;
    }
}

// ------------------------------------------------ regression head (type_y = 'cont')
// q(y|.) = N(mu, var) with mu = sigmoid(.) and a FIXED variance (src/DrVAE.py:167-169).  Per classifier
// row r = (l, i):  labeled -> yl[r] = log N(y_i; mu, var), the fprop input takes the true y;
// unlabeled -> y is sampled (SGVB, src/DrVAE.py:527-529): yv = mu + sd*eps, yl[r] = 0.
// The y columns of the two fprop inputs [z1 | y] and [z3 | y] are written here.
__global__ void ycont_fwd_kernel(const float* __restrict__ mu, int64_t ldm, const float* __restrict__ ylab,
                                 const int32_t* __restrict__ has_y, const float* __restrict__ eps, int64_t lde,
                                 float logvar, int sqerr, int R, int B, int Y, float* __restrict__ yl,
                                 float* __restrict__ fpin_y, int64_t ld1, float* __restrict__ z3in_y, int64_t ld2) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int i = r % B;
    const bool lab = has_y[i] != 0;
    const float sd = expf(0.5f * logvar), iv = expf(-logvar);
    float acc = 0.f;
    for (int d = 0; d < Y; ++d) {
        const float m = mu[(int64_t)r * ldm + d];
        float yv;
        if (lab) {
            yv = ylab[(int64_t)i * Y + d];
            const float df = yv - m;
            acc += sqerr ? 2.f * df * df : kLog2Pi + logvar + df * df * iv;    // (x -0.5 below)
        } else {
            yv = m + sd * eps[(int64_t)r * lde + d];
        }
        fpin_y[(int64_t)r * ld1 + d] = yv;
        z3in_y[(int64_t)r * ld2 + d] = yv;
    }
    yl[r] = lab ? -0.5f * acc : 0.f;
}

// dlogit[r,d] = dmu * mu(1-mu):  labeled dmu = c_yl[r]*(y-mu)/var;  unlabeled dmu = d/d(y columns of
// the two fprop inputs) (the sample is mu + const*eps).  cfp[r] = c_kld[r] (one fprop row per row).
__global__ void ycont_bwd_kernel(const float* __restrict__ mu, int64_t ldm, const float* __restrict__ ylab,
                                 const int32_t* __restrict__ has_y, float logvar, int sqerr,
                                 const float* __restrict__ c_yl, const float* __restrict__ c_kld,
                                 const float* __restrict__ dfpin_y, int64_t ld1,
                                 const float* __restrict__ dz3in_y, int64_t ld2, int R, int B, int Y,
                                 float* __restrict__ dlogit, int64_t ldd, float* __restrict__ cfp, int write_cfp) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int i = r % B;
    const bool lab = has_y[i] != 0;
    const float iv = expf(-logvar);
    if (write_cfp) {
        cfp[r] = c_kld[r];
        return;
    }
    for (int d = 0; d < Y; ++d) {
        const float m = mu[(int64_t)r * ldm + d];
        const float dmu = lab ? c_yl[r] * (ylab[(int64_t)i * Y + d] - m) * (sqerr ? 2.f : iv)
                              : dfpin_y[(int64_t)r * ld1 + d] + dz3in_y[(int64_t)r * ld2 + d];
        dlogit[(int64_t)r * ldd + d] = dmu * m * (1.f - m);
    }
}

// --------------------------------------------------- random-Fourier-feature MMD (K10)
// rf(x)[r] = c * cos(theta[i,r]), theta = a * x W + 2 pi b (the MFMA GEMM writes theta with its scale/bias
// epilogue);  diff[r] = mean_i rf(x1_i)[r] - mean_j rf(x2_j)[r];  mmd2 = sum_r diff[r]^2  (src/blocks.py:40-55).
// One block per 64 features, its 4 waves splitting the rows of theta1 then theta2 (lanes = consecutive r).
__global__ __launch_bounds__(256) void mmd_cos_means_kernel(const float* __restrict__ th1, int64_t ld1, int n1,
                                                            const float* __restrict__ th2, int64_t ld2, int n2, int R,
                                                            float c, float* __restrict__ diff) {
    __shared__ float part[2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 64 + lane, rc = r < R ? r : R - 1;
    for (int side = 0; side < 2; ++side) {
        const float* th = side ? th2 : th1;
        const int64_t ld = side ? ld2 : ld1;
        const int n = side ? n2 : n1;
        float acc = 0.f;
        for (int i = wave; i < n; i += 4) acc += cosf(th[(int64_t)i * ld + rc]);
        part[side][wave][lane] = acc;
    }
    __syncthreads();
    if (wave == 0 && r < R) {
        const float s1 = (part[0][0][lane] + part[0][1][lane]) + (part[0][2][lane] + part[0][3][lane]);
        const float s2 = (part[1][0][lane] + part[1][1][lane]) + (part[1][2][lane] + part[1][3][lane]);
        diff[r] = c * (s1 / (float)n1 - s2 / (float)n2);
    }
}

// out[0] = sum_r diff[r]^2  (single block, fixed-order tree: deterministic)
__global__ __launch_bounds__(256) void mmd_sumsq_kernel(const float* __restrict__ diff, int R, float* __restrict__ out) {
    __shared__ float part[4];
    float s = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) s += diff[r] * diff[r];
    s = dv_wave_sum_all(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (part[0] + part[1]) + (part[2] + part[3]);
}

// G[i,r] = coef * gout[0] * (-diff[r]) * sin(theta[i,r])   -- d mmd2 / d theta up to the caller's constant
__global__ void mmd_dtheta_kernel(const float* __restrict__ th, int64_t ld, int n, int R,
                                  const float* __restrict__ diff, const float* __restrict__ gout, float coef,
                                  float* __restrict__ G, int64_t ldg) {
    const int64_t total = (int64_t)n * R;
    const float s = coef * gout[0];
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / R), r = (int)(e % R);
        G[(int64_t)i * ldg + r] = -s * diff[r] * sinf(th[(int64_t)i * ld + r]);
    }
}

// ---------------------------------------------------------------- row movement
__global__ void rows_gather_kernel(const float* __restrict__ src, int64_t lds, const int32_t* __restrict__ idx,
                                   int n, int W, const float* __restrict__ noise, int64_t ldn, float sigma,
                                   const int32_t* __restrict__ onehot_cls, int Y, float* __restrict__ out,
                                   int64_t ldo, ParkArgs park) {
    park_block(park);
    const int WT = W + (onehot_cls ? Y : 0);
    const int64_t total = (int64_t)n * WT;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / WT), d = (int)(e % WT);
        float v;
        if (d < W) {
            v = src[(int64_t)(idx ? idx[r] : r) * lds + d];
            if (noise) v = fmaf(sigma, noise[(int64_t)r * ldn + d], v);
        } else {
            v = (d - W) == onehot_cls[r] ? 1.f : 0.f;
        }
        out[(int64_t)r * ldo + d] = v;
    }
}

// Per-batch masks of a batch-INDEPENDENT ("universal") step plan: the plan materialises every row as a pair with
// every class slot, and which rows really are pairs / labeled is data -- coefficient and weight vectors computed
// here, on the device, from the flags of the batch (src/DrVAE.py:565-624: group split + per-example normalisers
// N_total / max(1, N_pairs) / max(1, N_labeled), with N_pairs and N_labeled counted per batch).  One workgroup.
// Row i of the batch is dataset row table[b, i] (graph-resident epoch feed) or i itself (table == NULL).
struct MaskArgs {
    const int32_t* table;
    int n_batches;
    const int32_t* ctr;
    const int32_t* base;
    const int32_t* hx;
    const int32_t* hy;
    const int32_t* y;
    int B, L, Np;     // Np: rows [0, Np) of the batch have pair slots (the batch-independent plan: Np == B)
    float n_tot, kl_rate, pert_rate, yl_rate;
    const float* beta;
    float* c_nll;     // (3 L B): z1 rows | z2 rows | z2Fz1 rows
    float* c_klz2;    // (L B)
    float* c_yl;      // (L B)
    float* w_recl;    // (2 L B)  weights of RECL over the z1 | z2 rows
    float* w_pert;    // (L B)
    float* w_yl;      // (L B)
    int32_t* label;   // (L B): -2 - class for labeled rows, 0 otherwise (see ymarg_*_kernel)
    float* c_klp;     // (2 B), optional: KL-to-prior rows of q(z1|x1) | q(z2|x2) (PVAE, src/PVAE.py:330-345)
    const int32_t* one_slot;   // (B), optional: rows of the plan with ONE class slot (labeled for sure): label = class
    const int32_t* gcounts;    // (n_batches, 2), optional: GLOBAL (N_pairs, N_labeled) of every batch (data parallelism)
};

__device__ __forceinline__ void batch_masks_body(const MaskArgs& a) {
    __shared__ int cnt[2];
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int32_t* tb = nullptr;
    int b = 0;
    if (a.table) {
        b = a.ctr[0] - a.base[0];
        b = b < 0 ? 0 : (b >= a.n_batches ? a.n_batches - 1 : b);
        tb = a.table + (int64_t)b * a.B;
    }
    if (a.gcounts) {
        // data parallelism: this rank runs a slice of the global batch; the normalisers are the GLOBAL counts of the batch
        // (table data: every rank drew the same global table), not this slice's own
        if (threadIdx.x < 2) cnt[threadIdx.x] = a.gcounts[2 * b + threadIdx.x];
    } else {
        int np = 0, nl = 0;
        for (int i = threadIdx.x; i < a.B; i += blockDim.x) {
            const int src = tb ? tb[i] : i;
            np += (a.hx && a.hx[src] != 0) ? 1 : 0;
            nl += (a.hy && a.hy[src] != 0) ? 1 : 0;
        }
        // integer counts: order-independent, so atomics keep the step reproducible
        if (np) atomicAdd(&cnt[0], np);
        if (nl) atomicAdd(&cnt[1], nl);
    }
    __syncthreads();
    const float Lf = (float)a.L, beta = a.beta ? a.beta[0] : 1.f;
    const float n_pairs = cnt[0] > 0 ? (float)cnt[0] : 1.f, n_lab = cnt[1] > 0 ? (float)cnt[1] : 1.f;
    const float c_tot = 1.f / (Lf * a.n_tot);
    const int LB = a.L * a.B, LP = a.L * a.Np;
    // pair slots: slot q = (l, j) belongs to batch row j < Np (rows from Np on have no x2 / z2 rows in this plan:
    // the feed puts the batch's pairs first)
    if (a.c_nll && a.hx)
        for (int q = threadIdx.x; q < LP; q += blockDim.x) {
            const int j = q % a.Np, src = tb ? tb[j] : j;
            const bool px = a.hx[src] != 0;
            a.c_nll[LB + q] = px ? -c_tot : 0.f;
            a.w_recl[LB + q] = px ? c_tot : 0.f;
            a.c_nll[LB + LP + q] = px ? -beta * a.pert_rate / (Lf * n_pairs) : 0.f;
            a.w_pert[q] = px ? 1.f / (Lf * n_pairs) : 0.f;
            a.c_klz2[q] = px ? beta * a.kl_rate * c_tot : 0.f;
        }
    for (int r = threadIdx.x; r < LB; r += blockDim.x) {
        const int i = r % a.B, src = tb ? tb[i] : i;
        const bool py = a.hy && a.hy[src] != 0;
        if (a.c_nll) {
            a.c_nll[r] = -c_tot;
            a.w_recl[r] = c_tot;
        }
        if (a.hy) {
            a.c_yl[r] = py ? -a.yl_rate / (Lf * n_lab) : 0.f;
            a.w_yl[r] = 1.f / (Lf * n_lab);
            a.label[r] = (a.one_slot && a.one_slot[i]) ? a.y[src] : (py ? -2 - a.y[src] : 0);
        }
    }
    if (a.c_klp)
        for (int i = threadIdx.x; i < a.B; i += blockDim.x) {
            const int src = tb ? tb[i] : i;
            a.c_klp[i] = 1.f / a.n_tot;
            if (i < a.Np) a.c_klp[a.B + i] = (a.hx && a.hx[src] != 0) ? 1.f / a.n_tot : 0.f;
        }
}

__global__ __launch_bounds__(1024) void batch_masks_kernel(MaskArgs a) { batch_masks_body(a); }

// Graph-resident input feed: batch `b = ctr - base` of an epoch's index table is gathered from
// the HBM-resident dataset straight into the step's input rows ([x1 rows ; x2 rows of the pairs],
// + training noise), and the label-dependent index buffers of the step are refreshed, in ONE
// launch whose arguments never change -- so the whole epoch is replays of one captured graph.
__global__ __launch_bounds__(256) void batch_feed_kernel(
    const float* __restrict__ x1, int64_t ld1, const float* __restrict__ x2, int64_t ld2,
    const int32_t* __restrict__ y, const int32_t* __restrict__ table, int n_batches,
    const int32_t* __restrict__ ctr, const int32_t* __restrict__ base, int B, const int32_t* __restrict__ pair_rows,
    int Np, int X, const float* __restrict__ noise, int64_t ldn, float sigma, float* __restrict__ xin, int64_t ldo,
    const int32_t* __restrict__ has_y, int L, int32_t* __restrict__ label_r, const int32_t* __restrict__ fp_i,
    const int32_t* __restrict__ fp_lab, const int32_t* __restrict__ fp_slot, int Mf, int32_t* __restrict__ fp_cls,
    float* __restrict__ onehot, int64_t ldh, int Y, int row_blocks, int vec4, const float* __restrict__ yf,
    float* __restrict__ ylab, int Yc, float* __restrict__ onehot2, int64_t ldh2, int mask_block, MaskArgs masks,
    ParkArgs park) {
    park_block(park);
    if ((int)blockIdx.x == mask_block) {        // (batch-independent plan: the batch's masks ride on this launch)
        batch_masks_body(masks);
        return;
    }
    int b = ctr[0] - base[0];
    b = b < 0 ? 0 : (b >= n_batches ? n_batches - 1 : b);
    const int32_t* tb = table + (int64_t)b * B;
    if ((int)blockIdx.x < row_blocks) {
        const int lane = threadIdx.x & 63;
        const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (r >= B + Np) return;
        const float* src = r < B ? x1 + (int64_t)tb[r] * ld1 : x2 + (int64_t)tb[pair_rows[r - B]] * ld2;
        const float* nz = noise ? noise + (int64_t)r * ldn : nullptr;
        float* dst = xin + (int64_t)r * ldo;
        if (vec4) {
            // (a wave per row: four 16-B groups per lane in flight at a time -- a 978-gene row is ONE round trip of loads,
            // not four load -> store iterations; this launch is the head of the step's critical path)
            const int X4 = X >> 2;
            for (int q0 = 0; q0 < X4; q0 += 256) {
                float4 v[4], e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + lane + 64 * u, qc = q < X4 ? q : X4 - 1;
                    v[u] = reinterpret_cast<const float4*>(src)[qc];
                    e[u] = nz ? reinterpret_cast<const float4*>(nz)[qc] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + lane + 64 * u;
                    if (q < X4)
                        reinterpret_cast<float4*>(dst)[q] = make_float4(fmaf(sigma, e[u].x, v[u].x), fmaf(sigma, e[u].y, v[u].y),
                                                                        fmaf(sigma, e[u].z, v[u].z), fmaf(sigma, e[u].w, v[u].w));
                }
            }
            for (int g = (X & ~3) + lane; g < X; g += 64) dst[g] = nz ? fmaf(sigma, nz[g], src[g]) : src[g];
        } else {
            for (int g = lane; g < X; g += 64) dst[g] = nz ? fmaf(sigma, nz[g], src[g]) : src[g];
        }
        return;
    }
    const int t = ((int)blockIdx.x - row_blocks) * 256 + threadIdx.x;
    if (label_r && t < L * B) {
        const int i = t % B;
        label_r[t] = has_y[i] ? y[tb[i]] : 0;
    }
    if (ylab && t < B * Yc) {      // regression targets (type_y='cont'): ylab[i,:] = yf[row of slot i,:]
        const int i = t / Yc, d = t % Yc;
        ylab[t] = yf[(int64_t)tb[i] * Yc + d];
    }
    if (fp_cls && t < Mf) {
        const int cls = fp_lab[t] ? y[tb[fp_i[t]]] : fp_slot[t];
        fp_cls[t] = cls;
        if (onehot)
            for (int c = 0; c < Y; ++c) onehot[(int64_t)t * ldh + c] = c == cls ? 1.f : 0.f;
        if (onehot2)     // (a second copy of the block: the class columns of both fprop inputs)
            for (int c = 0; c < Y; ++c) onehot2[(int64_t)t * ldh2 + c] = c == cls ? 1.f : 0.f;
    }
}

__global__ void rows_segment_sum_kernel(const float* __restrict__ src, int64_t lds,
                                        const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_rows,
                                        const float* __restrict__ w, int n, int W,
                                        const int32_t* __restrict__ dst_idx, float* __restrict__ dst, int64_t ldd,
                                        float beta, ParkArgs park) {
    park_block(park);
    const int64_t total = (int64_t)n * W;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / W), d = (int)(e % W);
        float s = 0.f;
        if (seg_ptr) {
            for (int t = seg_ptr[i]; t < seg_ptr[i + 1]; ++t)
                s += (w ? w[t] : 1.f) * src[(int64_t)(seg_rows ? seg_rows[t] : t) * lds + d];
        } else {
            s = (w ? w[i] : 1.f) * src[(int64_t)(seg_rows ? seg_rows[i] : i) * lds + d];
        }
        float* o = dst + (int64_t)(dst_idx ? dst_idx[i] : i) * ldd + d;
        *o = (beta != 0.f ? beta * *o : 0.f) + s;
    }
}

__global__ __launch_bounds__(256) void weighted_sum_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const int32_t* __restrict__ idx, int n, float scale,
                                                           float* __restrict__ out, float beta) {
    __shared__ float part[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += (w ? w[i] : 1.f) * x[idx ? idx[i] : i];
    s = dv_wave_sum_all(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (part[0] + part[1]) + (part[2] + part[3]);
        out[0] = (beta != 0.f ? beta * out[0] : 0.f) + scale * s;
    }
}

// ------------------------------------------------ reconstruction metrics (evaluation, N1)
// per row i: [ sum (x-r)^2, mean x, mean r, sum (x-mx)^2, sum (r-mr)^2, sum (x-mx)(r-mr) ]
// two passes over the (cache-hot) row so that the centred sums do not cancel in fp32; the
// per-row Pearson r and the RMSE of src/DGMMixin.py:128-156 follow from these six numbers.
__global__ __launch_bounds__(256) void recon_row_stats_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const float* __restrict__ r, int64_t ldr, int M, int X,
                                                              float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < M; i += gridDim.x * 4) {
        const float* xr = x + (int64_t)i * ldx;
        const float* rr = r + (int64_t)i * ldr;
        float sx = 0.f, sr = 0.f, sse = 0.f;
        for (int g = lane; g < X; g += 64) {
            const float a = xr[g], b = rr[g];
            sx += a;
            sr += b;
            sse += (a - b) * (a - b);
        }
        sx = dv_wave_sum_all(sx);
        sr = dv_wave_sum_all(sr);
        sse = dv_wave_sum_all(sse);
        const float mx = sx / X, mr = sr / X;
        float cxx = 0.f, crr = 0.f, cxr = 0.f;
        for (int g = lane; g < X; g += 64) {
            const float a = xr[g] - mx, b = rr[g] - mr;
            cxx += a * a;
            crr += b * b;
            cxr += a * b;
        }
        cxx = dv_wave_sum_all(cxx);
        crr = dv_wave_sum_all(crr);
        cxr = dv_wave_sum_all(cxr);
        if (lane == 0) {
            float* o = out + (int64_t)i * 6;
            o[0] = sse; o[1] = mx; o[2] = mr; o[3] = cxx; o[4] = crr; o[5] = cxr;
        }
    }
}

// per column g: sum_i x[i,g] and sum_i x[i,g]^2 (double accumulation) and sum_i (x-r)^2:
// the ingredients of the variance-weighted R^2 (sklearn r2_score, src/DGMMixin.py:137)
constexpr int kCmRG = 16;       // row groups (waves) of a column-moment workgroup
__global__ __launch_bounds__(64 * kCmRG) void col_moments_kernel(const float* __restrict__ x, int64_t ldx,
                                                                 const float* __restrict__ r, int64_t ldr, int M, int X,
                                                                 double* __restrict__ out, int rows_per_block,
                                                                 const int32_t* __restrict__ sel,
                                                                 const float* __restrict__ r_bias) {
    // workgroup = (64 columns, one row block): 16 row groups of 64 lanes walk the block's rows, coalesced along the
    // columns; partial sums of the block go to out[blockIdx.y] (the caller adds the blocks up in a fixed order).  Sixteen
    // waves per block, not four: the same waves in flight over a quarter of the blocks -- dv_recon_finalize, ONE
    // workgroup, walks every block's partials (8192 rows: 64 -> 16 blocks, 28 -> 10 us)
    __shared__ double part[kCmRG][3][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int g = blockIdx.x * 64 + c;
    const int i0 = blockIdx.y * rows_per_block, i1 = min(M, i0 + rows_per_block);
    double s1 = 0., s2 = 0., se = 0.;
    if (g < X) {
        const float rb = r_bias ? r_bias[g] : 0.f;      // (r = a raw heads product: its bias is added here)
        for (int ii = i0 + rg; ii < i1; ii += kCmRG) {
            const int i = sel ? sel[ii] : ii;       // (sel: the M rows that count, e.g. the rows with a second profile)
            const double a = x[(int64_t)i * ldx + g], b = r_bias ? r[(int64_t)i * ldr + g] + rb : r[(int64_t)i * ldr + g];
            s1 += a;
            s2 += a * a;
            se += (a - b) * (a - b);
        }
    }
    part[rg][0][c] = s1;
    part[rg][1][c] = s2;
    part[rg][2][c] = se;
    __syncthreads();
    if (rg < 3 && g < X) {          // (wave k adds statistic k up, row groups in order)
        double t = 0.;
#pragma unroll
        for (int q = 0; q < kCmRG; ++q) t += part[q][rg][c];
        out[((int64_t)blockIdx.y * 3 + rg) * X + g] = t;
    }
}

// dv_recon_row_stats + the log-likelihood rows in ONE pass over (x, mu, sd) for rows of up to 1024 columns (whole-set
// evaluation: 978 genes): a wave holds its row in registers (16 columns per lane), the centred second pass costs no
// memory traffic.  Same outputs (and, on the dword path, summation orders) as recon_row_stats_kernel / nll_rows_fwd_kernel.
// RAW: mu / sd are the heads' raw products, finished on the way (mu + bias_mu, softplus(sd + bias_sd) + shift) -- the
// decoder's heads of the inference pass run as a plain product.
constexpr int kRcG = 16;
struct RcArgs {
    const float* x; int64_t ldx; const float* mu; const float* sd; int64_t ldp;
    const float* bias_mu; const float* bias_sd; float shift;
    int M, X;
    float* rows; float* ll;
};

// V2: 8-B loads -- a lane holds the column PAIRS lane + 64 k (even X, 8-B aligned rows); element e of the lane's registers is
// column 2 (lane + 64 (e / 2)) + e % 2
template <bool RAW, bool V2>
__global__ __launch_bounds__(256) void recon_rows_kernel(RcArgs a) {
    const int lane = threadIdx.x & 63;
    const int X = a.X;
    auto col = [lane](int e) { return V2 ? 2 * (lane + 64 * (e >> 1)) + (e & 1) : lane + 64 * e; };
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < a.M; i += gridDim.x * 4) {
        const float* xr = a.x + (int64_t)i * a.ldx;
        const float* mr = a.mu + (int64_t)i * a.ldp;
        const float* sr = a.sd + (int64_t)i * a.ldp;
        float av[kRcG], bv[kRcG], tv[kRcG];
        auto elem = [&](int k, float xv, float m, float sv, float bm, float bs) {
            av[k] = xv;
            if constexpr (RAW) {
                bv[k] = m + bm;
                tv[k] = nll_raw_term(a.shift, xv, m, sv, bm, bs);
            } else {
                bv[k] = m;
                tv[k] = nll_term(DV_GAUSS_SIGMA, xv, m, sv);
            }
        };
#pragma unroll
        for (int k = 0; k < kRcG; ++k) av[k] = bv[k] = tv[k] = 0.f;
        if constexpr (V2) {
#pragma unroll
            for (int k = 0; k < kRcG / 2; ++k) {
                const int c = lane + 64 * k;
                if (2 * c < X) {        // (X even: the pair is whole)
                    const float2 xv = reinterpret_cast<const float2*>(xr)[c];
                    const float2 mv = reinterpret_cast<const float2*>(mr)[c];
                    const float2 sv = reinterpret_cast<const float2*>(sr)[c];
                    float2 bm = make_float2(0.f, 0.f), bs = make_float2(0.f, 0.f);
                    if constexpr (RAW) {
                        bm = reinterpret_cast<const float2*>(a.bias_mu)[c];
                        bs = reinterpret_cast<const float2*>(a.bias_sd)[c];
                    }
                    elem(2 * k, xv.x, mv.x, sv.x, bm.x, bs.x);
                    elem(2 * k + 1, xv.y, mv.y, sv.y, bm.y, bs.y);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < kRcG; ++k) {
                const int g = lane + 64 * k;
                if (g < X) elem(k, xr[g], mr[g], sr[g], RAW ? a.bias_mu[g] : 0.f, RAW ? a.bias_sd[g] : 0.f);
            }
        }
        float sx = 0.f, sb = 0.f, sse = 0.f, sl = 0.f;
#pragma unroll
        for (int k = 0; k < kRcG; ++k)
            if (col(k) < X) {
                sx += av[k];
                sb += bv[k];
                sse += (av[k] - bv[k]) * (av[k] - bv[k]);
            }
        // (the log-likelihood terms in the order of nll_rows_fwd_kernel's scalar path: four per trip while four fit)
#pragma unroll
        for (int k = 0; k < kRcG; k += 4) {
            if (col(k + 3) < X && col(k) < X) {
                sl += (tv[k] + tv[k + 1]) + (tv[k + 2] + tv[k + 3]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (col(k + q) < X) sl += tv[k + q];
            }
        }
        sx = dv_wave_sum_all(sx);
        sb = dv_wave_sum_all(sb);
        sse = dv_wave_sum_all(sse);
        sl = dv_wave_sum_all(sl);
        const float mx = sx / X, mb = sb / X;
        float cxx = 0.f, cbb = 0.f, cxb = 0.f;
#pragma unroll
        for (int k = 0; k < kRcG; ++k)
            if (col(k) < X) {
                const float p = av[k] - mx, q = bv[k] - mb;
                cxx += p * p;
                cbb += q * q;
                cxb += p * q;
            }
        cxx = dv_wave_sum_all(cxx);
        cbb = dv_wave_sum_all(cbb);
        cxb = dv_wave_sum_all(cxb);
        if (lane == 0) {
            float* o = a.rows + (int64_t)i * 6;
            o[0] = sse; o[1] = mx; o[2] = mb; o[3] = cxx; o[4] = cbb; o[5] = cxb;
            if (a.ll) a.ll[i] = -0.5f * sl;
        }
    }
}

// ------------------------------------------------------------------ kernel-mixture MMD (src/blocks.py:29-38,59-76; round 5)
// mmd_objective(kernel = 'rbf' | 'poly'): mean_ij k(a_i, b_j) with k = 1/nb sum_b f(., gamma_b) over three Gram products
// (x1 x1^T, x2 x2^T, x1 x2^T: dv_gemm).  The element-wise mixture, its mean and its derivative were library element-wise
// chains (pow / exp per bandwidth, five times over); here one row pass each way.
//   poly: f = (gamma G_ij + 1)^2                     (degree 2, bias 1: what mmd_objective calls, src/blocks.py:34-35,70-74)
//   rbf : f = exp(-gamma (|a_i|^2 + |b_j|^2 - 2 G_ij))   (the Gram form of src/blocks.py:29-32's intent)
// |a_i|^2 = the diagonal of a a^T: sa[i * sa_stride], no pass of its own.
struct MixArgs {
    const float* G; int64_t ldg; int M, N, kind, nb; float gam[8];
    const float* sa; int64_t sa_stride; const float* sb; int64_t sb_stride;
};

__device__ __forceinline__ float mix_val(const MixArgs& a, float g, float d2, float& dv) {
    float v = 0.f;
    dv = 0.f;
    for (int b = 0; b < a.nb; ++b) {
        const float gm = a.gam[b];
        if (a.kind == 0) {
            const float t = fmaf(gm, g, 1.f);
            v += t * t;
            dv += 2.f * gm * t;             // d/dG
        } else {
            const float e = __expf(-gm * d2);
            v += e;
            dv -= gm * e;                   // d/d(d2)
        }
    }
    const float inv = 1.f / (float)a.nb;
    dv *= inv;
    return v * inv;
}

// part[i] = sum_j k_ij (fwd) | W[i, j] = c * dk_ij, rs[i] = sum_j W[i, j] (bwd; c = coef * gout[0]); one workgroup per row
template <bool BWD>
__global__ __launch_bounds__(256) void mmd_mix_kernel(MixArgs a, float* __restrict__ part, const float* __restrict__ gout,
                                                      float coef, float* __restrict__ W, int64_t ldw, float* __restrict__ rs) {
    __shared__ float red[4];
    const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* g = a.G + (int64_t)i * a.ldg;
    const float sai = a.kind ? a.sa[(int64_t)i * a.sa_stride] : 0.f;
    const float c = BWD ? coef * gout[0] : 0.f;
    float acc = 0.f;
    for (int j = threadIdx.x; j < a.N; j += 256) {
        const float gij = g[j];
        const float d2 = a.kind ? fmaxf(sai + a.sb[(int64_t)j * a.sb_stride] - 2.f * gij, 0.f) : 0.f;
        float dv;
        const float v = mix_val(a, gij, d2, dv);
        if (BWD) {
            const float w = c * dv;
            W[(int64_t)i * ldw + j] = w;
            acc += w;
        } else {
            acc += v;
        }
    }
    acc = dv_wave_sum_all(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) (BWD ? rs : part)[i] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = m11 - 2 m12 + m22 (the MMD^2), out[1..3] = the three means; fixed summation order, one thread
__global__ void mmd_mix_combine_kernel(const float* p11, int n11, float c11, const float* p12, int n12, float c12,
                                       const float* p22, int n22, float c22, float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s11 = 0., s12 = 0., s22 = 0.;
    for (int i = 0; i < n11; ++i) s11 += p11[i];
    for (int i = 0; i < n12; ++i) s12 += p12[i];
    for (int i = 0; i < n22; ++i) s22 += p22[i];
    const double m11 = s11 / c11, m12 = s12 / c12, m22 = s22 / c22;
    out[0] = (float)(m11 - 2. * m12 + m22);
    out[1] = (float)m11;
    out[2] = (float)m12;
    out[3] = (float)m22;
}

// identity kernel (src/blocks.py:37-38): out[0] = || mean(x1, 0) - mean(x2, 0) ||^2, diff[d] = the difference of the means
__global__ __launch_bounds__(256) void mmd_identity_fwd_kernel(const float* __restrict__ x1, int64_t ld1, int n1,
                                                               const float* __restrict__ x2, int64_t ld2, int n2, int Z,
                                                               float* __restrict__ diff, float* __restrict__ out) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int d = threadIdx.x; d < Z; d += 256) {
        float s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < n1; ++i) s1 += x1[(int64_t)i * ld1 + d];
        for (int i = 0; i < n2; ++i) s2 += x2[(int64_t)i * ld2 + d];
        const float df = s1 / (float)n1 - s2 / (float)n2;
        diff[d] = df;
        acc += df * df;
    }
    acc = dv_wave_sum_all(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

// dx[i, d] = coef * gout[0] * diff[d]     (d ||diff||^2 / d x1[i, d] = 2 diff[d] / n1: coef = +-2 / n)
__global__ void mmd_identity_bwd_kernel(const float* __restrict__ diff, const float* __restrict__ gout, float coef, int n,
                                        int Z, float* __restrict__ dx, int64_t ldd) {
    const int64_t total = (int64_t)n * Z;
    const float c = coef * gout[0];
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
        dx[(e / Z) * ldd + (e % Z)] = c * diff[e % Z];
}

// ------------------------------------------------------------------ evaluation tail (round 5)
// What follows the big kernels of a whole-set evaluation (src/DGMMixin.py:128-190) used to be ~75 small library launches
// per evaluation (two sorts, scans, gathers, float64 element-wise chains); as three launches:
//   recon_finalize : the float64 combination of dv_recon_row_stats / dv_col_moments / the log-likelihood rows
//   pair_counts    : per labeled row i and class: how many rows (positives) score above / equal -- O(n^2) integer
//                    counting over the chip instead of a sort (n <= 32768: 8192 rows = 67 M pairs = a few us)
//   rank_finalize  : accuracy, ROC-AUC and average precision from those counts
// ROC-AUC = sum over negatives j of (#pos above j + 1/2 #pos tied with j) / (n_pos n_neg)  -- the trapezoids over the
// distinct thresholds, sklearn's roc_auc_score; integer arithmetic, exact.  AP = 1/n_pos sum over positives i of
// (#pos >= s_i) / (#all >= s_i) -- sklearn's average_precision_score (one threshold per distinct score).

// out[0..3] = rmse, variance-weighted R^2, mean per-row Pearson r, mean log-likelihood (float64); rows: (M_all, 6) of
// dv_recon_row_stats; sel (n rows, optional): the rows that count; cols: (row_blocks, 3, X) partials of dv_col_moments
// over the same rows; ll (per row, optional).  One workgroup, fixed summation order.
__global__ __launch_bounds__(1024) void recon_finalize_kernel(const float* __restrict__ rows, const int32_t* __restrict__ sel,
                                                              int n, int X, const double* __restrict__ cols, int row_blocks,
                                                              const float* __restrict__ ll, double* __restrict__ out) {
    __shared__ double red[5][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double sse = 0., pr = 0., sl = 0., num = 0., den = 0.;
    for (int e = threadIdx.x; e < n; e += 1024) {
        const int i = sel ? sel[e] : e;
        const float* o = rows + (int64_t)i * 6;
        sse += (double)o[0];
        pr += (double)o[5] / sqrt((double)o[3] * (double)o[4]);
        if (ll) sl += (double)ll[i];
    }
    for (int g = threadIdx.x; g < X; g += 1024) {
        double c0 = 0., c1 = 0., c2 = 0.;
        for (int b = 0; b < row_blocks; ++b) {
            const double* c = cols + (int64_t)b * 3 * X;
            c0 += c[g];
            c1 += c[X + g];
            c2 += c[2 * (int64_t)X + g];
        }
        num += c2;
        den += c1 - c0 * c0 / (double)n;
    }
    double v[5] = {sse, pr, sl, num, den};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double t = v[k];
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
        if (lane == 0) red[k][wave] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[5];
        for (int k = 0; k < 5; ++k) {
            t[k] = 0.;
            for (int w = 0; w < 16; ++w) t[k] += red[k][w];
        }
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        out[0] = n ? sqrt(t[0] / ((double)n * X)) : nan;
        out[1] = 1.0 - t[3] / t[4];
        out[2] = n ? t[1] / n : nan;
        out[3] = (ll && n) ? t[2] / n : nan;
    }
}

// counts[c][e][0..3] += { rows scoring above e, rows tied with e (e itself included), positives above, positives tied }
// over the j-range of this workgroup; e, j index the selected rows; class c scores proba[row, c0 + c]
constexpr int kPairJ = 512;
__global__ __launch_bounds__(256) void pair_counts_kernel(const float* __restrict__ proba, int64_t ldp,
                                                          const int32_t* __restrict__ y, const int32_t* __restrict__ sel,
                                                          int n, int c0, int binary, int32_t* __restrict__ counts) {
    __shared__ float sj[kPairJ];
    __shared__ int pj[kPairJ];
    const int cls = c0 + blockIdx.z;
    const int j0 = blockIdx.y * kPairJ, j1 = min(n, j0 + kPairJ);
    for (int t = threadIdx.x; t < j1 - j0; t += 256) {
        const int row = sel ? sel[j0 + t] : j0 + t;
        sj[t] = proba[(int64_t)row * ldp + cls];
        pj[t] = binary ? (y[row] > 0) : (y[row] == cls);
    }
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const float si = proba[(int64_t)(sel ? sel[e] : e) * ldp + cls];
    int ga = 0, ea = 0, gp = 0, ep = 0;
    for (int t = 0; t < j1 - j0; ++t) {
        const float v = sj[t];
        const int p = pj[t];
        const int g = v > si, q = v == si;
        ga += g; ea += q; gp += g & p; ep += q & p;
    }
    int32_t* o = counts + ((int64_t)blockIdx.z * n + e) * 4;
    // (integer atomics: order-independent, the result is reproducible)
    atomicAdd(o + 0, ga); atomicAdd(o + 1, ea); atomicAdd(o + 2, gp); atomicAdd(o + 3, ep);
}

// per class c (one workgroup each): out[c*2 + 0] = ROC-AUC, out[c*2 + 1] = average precision (nan / 0 for the degenerate
// cases like metrics.roc_auc / average_precision); class 0's workgroup also writes acc -> out[2*n_cls].  Zeroes the
// counts it has read: the buffer is ready for the next evaluation.
__global__ __launch_bounds__(1024) void rank_finalize_kernel(int32_t* __restrict__ counts, const int32_t* __restrict__ y,
                                                             const int32_t* __restrict__ pred, const int32_t* __restrict__ sel,
                                                             int n, int c0, int binary, int n_cls, double* __restrict__ out) {
    __shared__ double red[16];
    __shared__ long long redi[3][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x, cls = c0 + c;
    long long s2 = 0, npos = 0, hit = 0;
    double ap = 0.;
    for (int e = threadIdx.x; e < n; e += 1024) {
        const int row = sel ? sel[e] : e;
        int32_t* o = counts + ((int64_t)c * n + e) * 4;
        const int ga = o[0], ea = o[1], gp = o[2], ep = o[3];
        o[0] = o[1] = o[2] = o[3] = 0;
        const int p = binary ? (y[row] > 0) : (y[row] == cls);
        if (p) {
            npos += 1;
            ap += (double)(gp + ep) / (double)(ga + ea);
        } else {
            s2 += 2LL * gp + ep;
        }
        if (pred) hit += pred[row] == y[row];
    }
    long long vi[3] = {s2, npos, hit};
    for (int off = 32; off >= 1; off >>= 1) ap += __shfl_xor(ap, off);
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int off = 32; off >= 1; off >>= 1) vi[k] += __shfl_xor(vi[k], off);
    if (lane == 0) {
        red[wave] = ap;
        for (int k = 0; k < 3; ++k) redi[k][wave] = vi[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.;
        long long t[3] = {0, 0, 0};
        for (int w = 0; w < 16; ++w) {
            a += red[w];
            for (int k = 0; k < 3; ++k) t[k] += redi[k][w];
        }
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        const long long np = t[1], nn = (long long)n - np;
        out[2 * c + 0] = (n == 0 || np == 0 || nn == 0) ? nan : ((double)t[0] * 0.5) / ((double)np * (double)nn);
        out[2 * c + 1] = (n == 0 || np == 0) ? 0.0 : a / (double)np;
        if (c == 0) out[2 * n_cls] = n ? (double)((float)t[2] / (float)n) : nan;
    }
}

// ---------------------------------------------------------------------- BatchNorm1d / Dropout (blocks.MLP options)
// nn.BatchNorm1d(affine=True) over the rows of x (M, N) (src/blocks.py:137-149): one workgroup per 64 columns,
// 4 row groups; training: batch mean and biased variance (two passes: mean, then sum of squared deviations), the
// running statistics move by `momentum` (running_var takes the UNBIASED variance, like torch); eval: the running
// statistics.  mean / rstd of the batch are kept for the backward pass.
__global__ __launch_bounds__(256) void bn_fwd_kernel(const float* __restrict__ x, int64_t ldx, int M, int N,
                                                     const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     float* __restrict__ y, int64_t ldy, float* __restrict__ rmean,
                                                     float* __restrict__ rvar, float momentum, int training) {
    __shared__ float part[4][64];
    __shared__ float stat[2][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int g = blockIdx.x * 64 + c;
    const bool ok = g < N;
    if (training) {
        float s = 0.f;
        if (ok)
            for (int i = rg; i < M; i += 4) s += x[(int64_t)i * ldx + g];
        part[rg][c] = s;
        __syncthreads();
        if (rg == 0) stat[0][c] = ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) / (float)M;
        __syncthreads();
        const float mu = stat[0][c];
        float q = 0.f;
        if (ok)
            for (int i = rg; i < M; i += 4) {
                const float d = x[(int64_t)i * ldx + g] - mu;
                q += d * d;
            }
        __syncthreads();
        part[rg][c] = q;
        __syncthreads();
        if (rg == 0) {
            const float var = ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) / (float)M;
            stat[1][c] = 1.f / sqrtf(var + eps);
            if (ok) {
                mean_out[g] = mu;
                rstd_out[g] = stat[1][c];
                if (rmean) rmean[g] = (1.f - momentum) * rmean[g] + momentum * mu;
                if (rvar) rvar[g] = (1.f - momentum) * rvar[g] + momentum * (M > 1 ? var * (float)M / (float)(M - 1) : var);
            }
        }
        __syncthreads();
    } else {
        if (rg == 0) {
            stat[0][c] = ok ? rmean[g] : 0.f;
            stat[1][c] = ok ? 1.f / sqrtf(rvar[g] + eps) : 0.f;
            if (ok) {
                mean_out[g] = stat[0][c];
                rstd_out[g] = stat[1][c];
            }
        }
        __syncthreads();
    }
    if (ok) {
        const float mu = stat[0][c], rs = stat[1][c], ww = w ? w[g] : 1.f, bb = b ? b[g] : 0.f;
        for (int i = rg; i < M; i += 4) y[(int64_t)i * ldy + g] = (x[(int64_t)i * ldx + g] - mu) * rs * ww + bb;
    }
}

// backward: dw = sum dy*xhat, db = sum dy; training: dx = w*rstd*(dy - db/M - xhat*dw/M); eval: dx = dy*w*rstd
__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ dy, int64_t ldd, const float* __restrict__ x,
                                                     int64_t ldx, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ w, int M,
                                                     int N, float* __restrict__ dx, int64_t lddx, float* __restrict__ dw,
                                                     float* __restrict__ db, int training) {
    __shared__ float p1[4][64], p2[4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int g = blockIdx.x * 64 + c;
    const bool ok = g < N;
    const float mu = ok ? mean[g] : 0.f, rs = ok ? rstd[g] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    if (ok)
        for (int i = rg; i < M; i += 4) {
            const float d = dy[(int64_t)i * ldd + g];
            s1 += d;
            s2 += d * ((x[(int64_t)i * ldx + g] - mu) * rs);
        }
    p1[rg][c] = s1;
    p2[rg][c] = s2;
    __syncthreads();
    const float sd = (p1[0][c] + p1[1][c]) + (p1[2][c] + p1[3][c]);
    const float sx = (p2[0][c] + p2[1][c]) + (p2[2][c] + p2[3][c]);
    if (!ok) return;
    if (rg == 0) {
        if (dw) dw[g] = sx;
        if (db) db[g] = sd;
    }
    if (dx == nullptr) return;
    const float ww = w ? w[g] : 1.f, im = 1.f / (float)M;
    for (int i = rg; i < M; i += 4) {
        const float d = dy[(int64_t)i * ldd + g];
        const float xh = (x[(int64_t)i * ldx + g] - mu) * rs;
        dx[(int64_t)i * lddx + g] = training ? ww * rs * (d - sd * im - xh * sx * im) : d * ww * rs;
    }
}

// y = x * mask * scale (nn.Dropout forward with its keep mask, and its backward: the same map on dy)
__global__ void mask_scale_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ mask, int64_t ldm,
                                  float scale, int M, int N, float* __restrict__ y, int64_t ldy) {
    const int64_t total = (int64_t)M * N;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / N, j = e - i * N;
        y[i * ldy + j] = x[i * ldx + j] * mask[i * ldm + j] * scale;
    }
}

struct LossTerms {
    dv_loss_term t[DV_MAX_LOSS_TERMS];
    int n;
};

// all loss scalars of a step in ONE single-workgroup launch: loss[out] += scale * sum_i w[i]*x[i]
// per term, then ELBO = <w_elbo, loss[0:3]>, CMPL = <w_cmpl, loss[0:8]>  (src/DrVAE.py:611-624)
constexpr int kLossThreads = 1024;
__global__ __launch_bounds__(kLossThreads) void loss_assemble_kernel(LossTerms lt, const float* __restrict__ w_elbo,
                                                            const float* __restrict__ w_cmpl,
                                                            float* __restrict__ loss, int32_t* flag,
                                                            const int32_t* ctr, int add, int32_t* err, int max_spins,
                                                            CounterBump bump, const int32_t* __restrict__ halt,
                                                            int n_halt, float* __restrict__ accum) {
    __shared__ float part[DV_MAX_LOSS_TERMS][kLossThreads / 64];
    __shared__ float acc[8];
    if (flag != nullptr) {      // park until the other launch chain has published its results
        if (threadIdx.x == 0) {
            const int want = ctr[0] + add;
            const long long t0 = wall_clock64();
            int n = 0;
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want < 0) {
                __builtin_amdgcn_s_sleep(4);
                if (++n > max_spins) {
                    atomicExch(err, 1);
                    break;
                }
            }
            atomicAdd(err + 1, (int32_t)(wall_clock64() - t0));
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every wave acquires (see park_block)
    }
    // every term's per-thread partial sum first -- all loads of all terms are independent and in flight together
    // (this is ONE workgroup: its time is the number of dependent load round trips) -- then one reduction stage;
    // fixed order throughout
    // (the weights and the running sums thread 0 combines at the end travel together with the terms' loads)
    float we[3] = {0.f, 0.f, 0.f}, wc[8], ac[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wc[i] = ac[i] = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) we[i] = w_elbo[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) wc[i] = w_cmpl[i];
        if (accum != nullptr) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ac[i] = accum[i];
        }
    }
    float ps[DV_MAX_LOSS_TERMS];
#pragma unroll
    for (int k = 0; k < DV_MAX_LOSS_TERMS; ++k) {
        ps[k] = 0.f;
        if (k >= lt.n) continue;
        const dv_loss_term t = lt.t[k];
        if (t.w == nullptr && t.n >= 4 * kLossThreads && (reinterpret_cast<uintptr_t>(t.x) & 15) == 0) {
            // long unweighted terms (per-tile partial sums of dv_gemm_heads): 16-B loads, four per trip
            const float4* x4 = reinterpret_cast<const float4*>(t.x);
            const int n4 = t.n >> 2;
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
            int i = threadIdx.x;
            for (; i + 3 * kLossThreads < n4; i += 4 * kLossThreads) {
                const float4 v0 = x4[i], v1 = x4[i + kLossThreads], v2 = x4[i + 2 * kLossThreads],
                             v3 = x4[i + 3 * kLossThreads];
                a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
                a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
                a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
                a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
            }
            for (; i < n4; i += kLossThreads) {
                const float4 v0 = x4[i];
                a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
            }
            float q = ((a0.x + a0.y) + (a0.z + a0.w)) + ((a1.x + a1.y) + (a1.z + a1.w)) +
                      (((a2.x + a2.y) + (a2.z + a2.w)) + ((a3.x + a3.y) + (a3.z + a3.w)));
            for (int j = (n4 << 2) + threadIdx.x; j < t.n; j += kLossThreads) q += t.x[j];
            ps[k] = q;
        } else {
            float q = 0.f;
            const int rl = t.row_len > 1 ? t.row_len : 1;     // x is (rows, row_len): one weight per ROW
            if (rl == 1) {
                for (int i = threadIdx.x; i < t.n; i += kLossThreads) q += (t.w ? t.w[i] : 1.f) * t.x[i];
            } else if (t.w != nullptr) {
                // one thread per ROW: its weight once, its row_len partials as independent loads (the element loop
                // below pays an integer division and a dependent weight load per element: 38 trips per thread for the
                // 620 x 62 per-tile partials of a bucketed sampler plan)
                const int rows = t.n / rl;
                for (int r = threadIdx.x; r < rows; r += kLossThreads) {
                    const float* xr = t.x + (int64_t)r * rl;
                    // (sixteen loads per trip: this is one workgroup, its time is the number of dependent round trips --
                    // the 62 per-tile partials of a row in four trips instead of sixteen)
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                    int j = 0;
                    for (; j + 15 < rl; j += 16) {
                        float v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[u] = xr[j + u];
                        s0 += (v[0] + v[4]) + (v[8] + v[12]);
                        s1 += (v[1] + v[5]) + (v[9] + v[13]);
                        s2 += (v[2] + v[6]) + (v[10] + v[14]);
                        s3 += (v[3] + v[7]) + (v[11] + v[15]);
                    }
                    for (; j + 3 < rl; j += 4) {
                        s0 += xr[j];
                        s1 += xr[j + 1];
                        s2 += xr[j + 2];
                        s3 += xr[j + 3];
                    }
                    for (; j < rl; ++j) s0 += xr[j];
                    q += t.w[r] * ((s0 + s1) + (s2 + s3));
                }
            } else {
                for (int i = threadIdx.x; i < t.n; i += kLossThreads) q += t.x[i];
            }
            ps[k] = q;
        }
    }
#pragma unroll
    for (int k = 0; k < DV_MAX_LOSS_TERMS; ++k) {
        if (k >= lt.n) continue;
        const float q = dv_wave_sum_all(ps[k]);
        if ((threadIdx.x & 63) == 0) part[k][threadIdx.x >> 6] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int k = 0; k < lt.n; ++k) {
            float q = 0.f;
            for (int w = 0; w < kLossThreads / 64; ++w) q += part[k][w];
            acc[lt.t[k].out] += lt.t[k].scale * q;
        }
    }
    if (threadIdx.x == 0) {
        if (!(flag != nullptr && lt.n == 0)) {     // (parked variant without terms: only the wait and the counters)
            acc[5] = we[0] * acc[0] + we[1] * acc[1] + we[2] * acc[2];
            float c = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) c += wc[i] * acc[i];
            acc[6] = c;
            // a device-side wait of this step's chains has timed out at some point (sticky error words):
            // whatever was computed since is built on stale data -- poison the scalars the host reads
            bool bad = false;
            for (int i = 0; i < n_halt; ++i)
                bad = bad || __hip_atomic_load(halt + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            for (int i = 0; i < 8; ++i) loss[i] = bad ? __builtin_nanf("") : acc[i];
            // running sums of a training epoch (only the launch that assembled the scalars adds them: the parked
            // variant without terms of the dual-graph step must not add a second time)
            if (accum != nullptr) {
#pragma unroll
                for (int i = 0; i < 8; ++i) accum[i] = ac[i] + (bad ? __builtin_nanf("") : acc[i]);
            }
        }
        // end of the step's use of the device counters on this chain: advance them here (saves the
        // separate counter launch in front of the optimiser)
        bump_counters(bump);
    }
}

__global__ void axpby_kernel(const float* __restrict__ x, float a, float* __restrict__ y, float b, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = a * x[i] + (b != 0.f ? b * y[i] : 0.f);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

#define ST(s) static_cast<hipStream_t>(s)

static bool park_ok(const dv_wait* p) { return p == nullptr || p->flag == nullptr || (p->ctr && p->err && p->max_spins > 0); }
static bool bump_ok(const dv_bump* b) {
    return b == nullptr || ((!b->c[0] || b->n[0] == 1 || b->n[0] == 2) && (!b->c[1] || b->n[1] == 1 || b->n[1] == 2));
}

extern "C" int dv_colsum(const float* X, int64_t ldx, int32_t M, int32_t N, float* out, float beta,
                         dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && N >= 0);
    if (N == 0) return DV_OK;
    DV_REQUIRE(X && out);
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, ST(stream), X, ldx, M, N, out, beta);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_act_bwd(float* dY, int64_t ldd, const float* Y, int64_t ldy, int32_t M, int32_t N, int32_t split,
                          int32_t act0, int32_t act1, float shift0, float shift1, dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && N >= 0);
    if (M == 0 || N == 0) return DV_OK;
    DV_REQUIRE(dY && Y);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for((int64_t)M * N, 256)), dim3(256), 0, ST(stream), dY, ldd, Y,
                       ldy, M, N, split, act0, act1, shift0, shift1);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_wn_scale(const float* W, int64_t ldw, const float* g, int32_t N, int32_t K, float* scale,
                           float* norm, dv_stream_t stream) {
    DV_REQUIRE(N >= 0 && K >= 0);
    if (N == 0) return DV_OK;
    DV_REQUIRE(W && g && scale);
    hipLaunchKernelGGL(wn_scale_kernel, dim3((N + 3) / 4), dim3(256), 0, ST(stream), W, ldw, g, N, K, scale, norm);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_wn_bwd(const float* dWraw, int64_t ldr, const float* W, int64_t ldw, const float* g,
                         const float* norm, int32_t N, int32_t K, float* dW, int64_t ldd, float* dg, float beta,
                         dv_stream_t stream) {
    DV_REQUIRE(N >= 0 && K >= 0);
    if (N == 0) return DV_OK;
    DV_REQUIRE(dWraw && W && g && norm && dW && dg);
    hipLaunchKernelGGL(wn_bwd_kernel, dim3((N + 3) / 4), dim3(256), 0, ST(stream), dWraw, ldr, W, ldw, g, norm, N,
                       K, dW, ldd, dg, beta);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_reparam_fwd(const float* mu, const float* sd, int64_t ldq, const int32_t* src_idx, int32_t n,
                              int32_t reps, int32_t Z, const float* eps, int64_t lde, int32_t mode, float* out,
                              int64_t ldo, const float* sub, int64_t lds, float* out2, int64_t ldo2, float* out3,
                              int64_t ldo3, const int32_t* out3_idx, dv_stream_t stream) {
    DV_REQUIRE(n >= 0 && reps >= 0 && Z >= 0);
    if (n == 0 || reps == 0 || Z == 0) return DV_OK;
    DV_REQUIRE(mu && sd && eps && out);
    DV_REQUIRE(out2 == nullptr || sub != nullptr);
    DV_REQUIRE(out3 == nullptr || out3_idx != nullptr);
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(grid_for((int64_t)n * reps * Z, 256)), dim3(256), 0, ST(stream), mu,
                       sd, ldq, src_idx, n, reps, Z, eps, lde, mode, out, ldo, sub, lds, out2, ldo2, out3, ldo3,
                       out3_idx);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_reparam_bwd(const float* dz, int64_t ldz, const float* eps, int64_t lde, const float* sd,
                              int64_t ldq, const int32_t* src_idx, int32_t n, int32_t reps, int32_t Z, int32_t mode,
                              float* dmu, float* dsd, int64_t lddq, float beta, dv_stream_t stream) {
    DV_REQUIRE(n >= 0 && reps >= 0 && Z >= 0);
    if (n == 0 || Z == 0) return DV_OK;
    DV_REQUIRE(dz && eps && sd && dmu && dsd);
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(grid_for((int64_t)n * Z, 256)), dim3(256), 0, ST(stream), dz, ldz,
                       eps, lde, sd, ldq, src_idx, n, reps, Z, mode, dmu, dsd, lddq, beta);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_reparam_bwd_seg(const float* dz, int64_t ldz, const float* eps, int64_t lde, const float* sd,
                                  int64_t ldq, const int32_t* seg_ptr, const int32_t* seg_rows, int32_t nq,
                                  int32_t Z, int32_t mode, const float* extra, int64_t ldx, const int32_t* ex_ptr,
                                  const int32_t* ex_rows, float* dmu, float* dsd, int64_t lddq, float beta,
                                  const dv_bump* bump_in, const dv_seg_add* add, const dv_wait* park_in,
                                  const dv_prior_kl* prior, dv_stream_t stream) {
    DV_REQUIRE(bump_ok(bump_in) && park_ok(park_in));
    DV_REQUIRE(prior == nullptr || prior->coef == nullptr || (prior->raw && prior->mu && mode == DV_GAUSS_LOGVAR));
    const dv_prior_kl pk = prior ? *prior : dv_prior_kl{};
    const CounterBump bump = bump_in ? *bump_in : CounterBump{};
    const ParkArgs park = park_in ? *park_in : ParkArgs{};
    DV_REQUIRE(nq >= 0 && Z >= 0);
    DV_REQUIRE(!(bump.c[0] || bump.c[1] || park.flag) || (nq > 0 && Z > 0));      // a bump / a wait needs a launch to ride on
    DV_REQUIRE(add == nullptr || add->n == 0 || (add->dz && add->n > 0));
    if (park.flag != nullptr && grid_for((int64_t)nq * Z, 256) > DV_MAX_PARKED_GRID) return DV_ERR_UNSUPPORTED;
    if (nq == 0 || Z == 0) return DV_OK;
    DV_REQUIRE(dz && eps && sd && seg_ptr && seg_rows && dmu && dsd);
    DV_REQUIRE(extra == nullptr || (ex_ptr && ex_rows));
    hipLaunchKernelGGL(reparam_bwd_seg_kernel, dim3(grid_for((int64_t)nq * Z, 256)), dim3(256), 0, ST(stream), dz,
                       ldz, eps, lde, sd, ldq, seg_ptr, seg_rows, nq, Z, mode, extra, ldx, ex_ptr, ex_rows, dmu, dsd,
                       lddq, beta, bump, add ? add->dz : nullptr, add ? add->ld : 0, add ? add->n : 0, park, pk);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_z2f_post_bwd(const dv_z2f_desc* dsc, const dv_wait* park_in, dv_stream_t stream) {
    DV_REQUIRE(dsc != nullptr && park_ok(park_in));
    const dv_z2f_desc& d = *dsc;
    const ParkArgs park = park_in ? *park_in : ParkArgs{};
    DV_REQUIRE(d.L >= 0 && d.B >= 0 && d.Np >= 0 && d.Z >= 0);
    DV_REQUIRE(park.flag == nullptr || (d.L > 0 && d.B > 0 && d.Z > 0));
    // every workgroup of a parked launch polls: keep such grids far below what the chip holds resident
    // (256 CUs x 8 workgroups), or the chain that is to publish may find no slot to run in
    if (park.flag != nullptr && grid_for((int64_t)d.B * d.Z, 256) > DV_MAX_PARKED_GRID) return DV_ERR_UNSUPPORTED;
    if (d.L == 0 || d.B == 0 || d.Z == 0) return DV_OK;
    DV_REQUIRE(d.eps && d.p2 && d.dp2 && d.dz1);   // dz2f == NULL: nothing flows into the z2Fz1 samples from a classifier
    DV_REQUIRE(d.Np == 0 || (d.pair_slot && d.q2 && d.coef && d.raw));
    DV_REQUIRE((d.prior_coef == nullptr) == (d.prior_raw == nullptr));
    Z2FArgs a{d.dz2f, d.ld_dz2f, d.dzdec_pert, d.ld_pert, d.Np ? d.pair_slot : nullptr, d.eps, d.lde, d.p2, d.ldp2,
              d.q2, d.ldq2, d.coef, d.raw, d.kl_min, d.dz1b, d.ld_dz1b, d.dp2, d.ld_dp2, d.dz1, d.ld_dz1, d.dq2,
              d.ld_dq2, d.L, d.B, d.Np, d.Z, d.Np ? d.prior_coef : nullptr, d.Np ? d.prior_raw : nullptr};
    hipLaunchKernelGGL(z2f_post_bwd_kernel, dim3(grid_for((int64_t)d.B * d.Z, 256)), dim3(256), 0, ST(stream), a, park);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_kl_rows_fwd(const dv_kl_rows_desc* dsc, const dv_wait* park_in, dv_stream_t stream) {
    DV_REQUIRE(dsc != nullptr);
    const dv_kl_rows_desc& d = *dsc;
    DV_REQUIRE(d.n >= 0 && d.reps >= 0 && d.Z >= 0 && park_ok(park_in));
    DV_REQUIRE((d.mu2 == nullptr) == (d.sd2 == nullptr) && (d.mu2 == nullptr || d.Z2 >= 0));
    const ParkArgs park = park_in ? *park_in : ParkArgs{};
    DV_REQUIRE(park.flag == nullptr || (d.n > 0 && d.reps > 0));
    if (park.flag != nullptr && grid_for((int64_t)d.n * d.reps, 4) > DV_MAX_PARKED_GRID) return DV_ERR_UNSUPPORTED;
    if (d.n == 0 || d.reps == 0) return DV_OK;
    DV_REQUIRE(d.mu_q && d.sd_q && d.out);
    DV_REQUIRE((d.mu_p == nullptr) == (d.sd_p == nullptr));
    DV_REQUIRE(d.zout == nullptr || d.eps != nullptr);
    KlArgs a{d.mu_q, d.sd_q, d.ldq, d.qidx, d.mu_p, d.sd_p, d.ldp, d.pidx, d.prior_mu, d.prior_sd, d.n, d.reps, d.Z, d.mode};
    hipLaunchKernelGGL(kl_rows_fwd_kernel, dim3(grid_for((int64_t)d.n * d.reps, 4)), dim3(256), 0, ST(stream), a,
                       d.free_bits, d.kl_min, d.raw_out, d.out, d.add, d.eps, d.lde, d.zout, d.ldz, park, d.mu2, d.sd2,
                       d.ld2, d.Z2, d.raw2_out);
    DV_RETURN_LAUNCH();
}

static int kl_fwd_of(const dv_kl_rows_desc& d, KlFwd& f) {
    DV_REQUIRE(d.n >= 0 && d.reps >= 0 && d.Z >= 0 && d.mu_q && d.sd_q && d.out);
    DV_REQUIRE((d.mu_p == nullptr) == (d.sd_p == nullptr));
    DV_REQUIRE(d.zout == nullptr && d.eps == nullptr && d.mu2 == nullptr && d.sd2 == nullptr);   // (plain rows only)
    f = KlFwd{KlArgs{d.mu_q, d.sd_q, d.ldq, d.qidx, d.mu_p, d.sd_p, d.ldp, d.pidx, d.prior_mu, d.prior_sd, d.n, d.reps, d.Z,
                     d.mode},
              d.free_bits, d.kl_min, d.raw_out, d.out, d.add};
    return DV_OK;
}

extern "C" int dv_kl_rows_fwd_pair(const dv_kl_rows_desc* d1, const dv_kl_rows_desc* d2, dv_stream_t stream) {
    DV_REQUIRE(d1 != nullptr && d2 != nullptr);
    KlFwd a, b;
    int rc = kl_fwd_of(*d1, a);
    if (rc != DV_OK) return rc;
    rc = kl_fwd_of(*d2, b);
    if (rc != DV_OK) return rc;
    const int64_t ba = ((int64_t)d1->n * d1->reps + 3) / 4, bb = ((int64_t)d2->n * d2->reps + 3) / 4;
    if (ba + bb == 0) return DV_OK;
    if (ba + bb > 0x7fffffff) return DV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kl_rows_fwd_pair_kernel, dim3((unsigned)(ba + bb)), dim3(256), 0, ST(stream), a, b, (int)ba);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_kl_rows_bwd(const dv_kl_rows_desc* dsc, const dv_kl_rows_grad* grad, dv_stream_t stream) {
    DV_REQUIRE(dsc != nullptr && grad != nullptr);
    const dv_kl_rows_desc& d = *dsc;
    const dv_kl_rows_grad& g = *grad;
    DV_REQUIRE(d.n >= 0 && d.reps >= 0 && d.Z >= 0);
    if (d.n == 0 || d.reps == 0 || d.Z == 0) return DV_OK;
    DV_REQUIRE(g.coef && d.mu_q && d.sd_q && g.dq_mu && g.dq_sd);
    DV_REQUIRE(!d.free_bits || d.raw_out != nullptr);
    DV_REQUIRE((d.mu_p == nullptr) == (d.sd_p == nullptr));
    DV_REQUIRE((g.dp_mu == nullptr) == (g.dp_sd == nullptr));
    DV_REQUIRE(g.dp_mu == nullptr || d.mu_p != nullptr);
    DV_REQUIRE(g.dz == nullptr || d.eps != nullptr);
    KlArgs a{d.mu_q, d.sd_q, d.ldq, d.qidx, d.mu_p, d.sd_p, d.ldp, d.pidx, d.prior_mu, d.prior_sd, d.n, d.reps, d.Z, d.mode};
    hipLaunchKernelGGL(kl_rows_bwd_kernel, dim3(grid_for((int64_t)d.n * d.reps * d.Z, 256)), dim3(256), 0, ST(stream), a,
                       g.coef, (const float*)d.raw_out, d.free_bits, d.kl_min, g.dq_mu, g.dq_sd, g.lddq, g.dp_mu, g.dp_sd,
                       g.lddp, g.beta, g.dz, g.ldz, d.eps, d.lde);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_gauss_nll_rows_fwd(const float* x, int64_t ldx, const int32_t* xidx, const float* mu,
                                     const float* sd, int64_t ldp, int32_t M, int32_t X, int32_t mode, float* out,
                                     const float* bias_mu, const float* bias_sd, float sd_shift, dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && X >= 0);
    if (M == 0) return DV_OK;
    DV_REQUIRE(x && mu && sd && out && ((bias_mu == nullptr) == (bias_sd == nullptr)));
    DV_REQUIRE(bias_mu == nullptr || mode == DV_GAUSS_SIGMA);      // (raw heads: the sigma head = softplus + shift)
    const bool v4 = aligned16(x) && aligned16(mu) && aligned16(sd) && (ldx % 4 == 0) && (ldp % 4 == 0);
    auto a8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
    const bool v2 = a8(x) && a8(mu) && a8(sd) && (ldx % 2 == 0) && (ldp % 2 == 0) && (!bias_mu || (a8(bias_mu) && a8(bias_sd)));
    const dim3 grid(grid_for(M, 4, 8192)), block(256);
#define DV_NLL_FWD(VEC, RAW)                                                                                             \
    hipLaunchKernelGGL((nll_rows_fwd_kernel<VEC, RAW>), grid, block, 0, ST(stream), x, ldx, xidx, mu, sd, ldp, M, X, mode, \
                       out, bias_mu, bias_sd, sd_shift)
    if (bias_mu) {
        if (v4) DV_NLL_FWD(4, true);
        else if (v2) DV_NLL_FWD(2, true);
        else DV_NLL_FWD(1, true);
    } else {
        if (v4) DV_NLL_FWD(4, false);
        else if (v2) DV_NLL_FWD(2, false);
        else DV_NLL_FWD(1, false);
    }
#undef DV_NLL_FWD
    DV_RETURN_LAUNCH();
}

extern "C" int dv_gauss_nll_rows_fwdbwd(const float* coef, const float* x, int64_t ldx, const int32_t* xidx,
                                        const float* mu, const float* sd, int64_t ldp, int32_t M, int32_t X,
                                        int32_t mode, int32_t sd_act, float sd_shift, float* out, float* dmu,
                                        float* dsd, int64_t ldd, const float* bias_mu, const float* bias_sd,
                                        dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && X >= 0);
    if (M == 0) return DV_OK;
    DV_REQUIRE(coef && x && mu && sd && out && dmu && dsd && ((bias_mu == nullptr) == (bias_sd == nullptr)));
    auto a8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
    const bool v2 = a8(x) && a8(mu) && a8(sd) && a8(dmu) && a8(dsd) && (ldx % 2 == 0) && (ldp % 2 == 0) && (ldd % 2 == 0);
    const dim3 grid(M < 16384 ? M : 16384), block(256);
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (bias_mu && mode == DV_GAUSS_SIGMA && sd_act == DV_ACT_SOFTPLUS && X % 4 == 0 && ldx % 4 == 0 && ldp % 4 == 0 &&
        ldd % 4 == 0 && a16(x) && a16(mu) && a16(sd) && a16(dmu) && a16(dsd) && a16(bias_mu) && a16(bias_sd)) {
        hipLaunchKernelGGL(nll_rows_raw_sp_kernel, grid, block, 0, ST(stream), coef, x, ldx, xidx, mu, sd, ldp, M, X,
                           sd_shift, out, dmu, dsd, ldd, bias_mu, bias_sd);
        DV_RETURN_LAUNCH();
    }
    if (v2)
        hipLaunchKernelGGL(nll_rows_fwdbwd_kernel<true>, grid, block, 0, ST(stream), coef, x, ldx, xidx, mu, sd, ldp, M,
                           X, mode, sd_act, sd_shift, out, dmu, dsd, ldd, bias_mu, bias_sd);
    else
        hipLaunchKernelGGL(nll_rows_fwdbwd_kernel<false>, grid, block, 0, ST(stream), coef, x, ldx, xidx, mu, sd, ldp,
                           M, X, mode, sd_act, sd_shift, out, dmu, dsd, ldd, bias_mu, bias_sd);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_gauss_nll_rows_raw_cs(const dv_nll_raw_cs_desc* dsc, dv_stream_t stream) {
    DV_REQUIRE(dsc != nullptr);
    const dv_nll_raw_cs_desc& d = *dsc;
    DV_REQUIRE(d.M >= 0 && d.X >= 4 && d.X % 4 == 0);
    if (d.M == 0) return DV_OK;
    const bool fwd_only = d.dmu == nullptr;        // (evaluation: row terms only)
    DV_REQUIRE(d.x && d.mu && d.sd && d.out_part && d.bias_mu && d.bias_sd);
    DV_REQUIRE(fwd_only ? (d.dsd == nullptr && d.ws == nullptr) : (d.coef && d.dsd && d.ws));
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    DV_REQUIRE(a16(d.x) && a16(d.mu) && a16(d.sd) && a16(d.bias_mu) && a16(d.bias_sd));
    DV_REQUIRE(d.ldx % 4 == 0 && d.ldp % 4 == 0);
    if (!fwd_only) {
        DV_REQUIRE(a16(d.dmu) && a16(d.dsd) && a16(d.ws) && d.ldd % 4 == 0 && d.ldw % 4 == 0 && d.sd_off % 4 == 0);
        DV_REQUIRE(d.sd_off >= d.X && d.sd_off + d.X <= d.ldw);
    }
    const int chunks = dv_nll_raw_cs_chunks(d.X), row_blocks = dv_nll_raw_cs_row_blocks(d.M);
    DV_REQUIRE(d.chunks == chunks && d.row_blocks == row_blocks && row_blocks <= 65535);
    NllCsArgs a{d.coef, d.x, d.ldx, d.xidx, d.mu, d.sd, d.ldp, d.M, d.X, d.shift, d.out_part, chunks, d.dmu, d.dsd, d.ldd,
                d.bias_mu, d.bias_sd, d.ws, d.ldw, d.sd_off};
    if (fwd_only)
        hipLaunchKernelGGL(nll_rows_raw_cs_kernel<true>, dim3(chunks, row_blocks), dim3(256), 0, ST(stream), a);
    else
        hipLaunchKernelGGL(nll_rows_raw_cs_kernel<false>, dim3(chunks, row_blocks), dim3(256), 0, ST(stream), a);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_nll_raw_cs_chunks(int32_t X) { return X < 4 ? 0 : ((X >> 2) + 255) / 256; }
extern "C" int dv_nll_raw_cs_row_blocks(int32_t M) { return M <= 0 ? 0 : (M + kNllCsRows - 1) / kNllCsRows; }

extern "C" int dv_rec_nll_rows(int32_t kind, float shift, const float* coef, const float* x, int64_t ldx,
                               const int32_t* xidx, const float* v, int64_t ldv, int32_t M, int32_t X, float* out,
                               float* dpre, int64_t ldd, dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && X >= 1 && (kind == DV_REC_BERNOULLI || kind == DV_REC_POISSON));
    if (M == 0) return DV_OK;
    DV_REQUIRE(x && v && out && (coef == nullptr || dpre != nullptr));
    hipLaunchKernelGGL(rec_nll_rows_kernel, dim3(grid_for(M, 1, 4096)), dim3(256), 0, ST(stream), kind, shift, coef, x,
                       ldx, xidx, v, ldv, M, X, out, dpre, ldd);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_gauss_nll_rows_bwd(const float* coef, const float* x, int64_t ldx, const int32_t* xidx,
                                     const float* mu, const float* sd, int64_t ldp, int32_t M, int32_t X,
                                     int32_t mode, int32_t sd_act, float sd_shift, float* dmu, float* dsd,
                                     int64_t ldd, float* dx, int64_t lddx, float beta, dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && X >= 0);
    if (M == 0 || X == 0) return DV_OK;
    DV_REQUIRE(coef && x && mu && sd && dmu && dsd);
    hipLaunchKernelGGL(nll_rows_bwd_kernel, dim3(grid_for((int64_t)M * X, 256, 8192)), dim3(256), 0, ST(stream),
                       coef, x, ldx, xidx, mu, sd, ldp, M, X, mode, sd_act, sd_shift, dmu, dsd, ldd, dx, lddx, beta);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_softmax_clamp_fwd(const float* logits, int64_t ldl, int32_t M, int32_t Y, int32_t sigmoid1,
                                    float* probs, int64_t ldp, dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && Y >= 1);
    if (M == 0) return DV_OK;
    DV_REQUIRE(logits && probs);
    DV_REQUIRE(!sigmoid1 || Y == 2);
    hipLaunchKernelGGL(softmax_clamp_fwd_kernel, dim3((M + 255) / 256), dim3(256), 0, ST(stream), logits, ldl, M, Y,
                       sigmoid1, probs, ldp);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_softmax_clamp_bwd(const float* dprobs, int64_t lddp, const float* probs, int64_t ldp, int32_t M,
                                    int32_t Y, int32_t sigmoid1, float* dlogits, int64_t ldl, float beta,
                                    dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && Y >= 1);
    if (M == 0) return DV_OK;
    DV_REQUIRE(dprobs && probs && dlogits);
    DV_REQUIRE(!sigmoid1 || Y == 2);
    hipLaunchKernelGGL(softmax_clamp_bwd_kernel, dim3((M + 255) / 256), dim3(256), 0, ST(stream), dprobs, lddp,
                       probs, ldp, M, Y, sigmoid1, dlogits, ldl, beta);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_cat_terms_fwd(const float* probs, int64_t ldp, int32_t M, int32_t Y, const int32_t* labels,
                                const float* prior, int64_t ldpr, float* logp, float* kl, int64_t ldk, float* ent,
                                int32_t* best, dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && Y >= 1);
    if (M == 0) return DV_OK;
    DV_REQUIRE(probs);
    DV_REQUIRE(logp == nullptr || labels != nullptr);
    DV_REQUIRE(kl == nullptr || prior != nullptr);
    hipLaunchKernelGGL(cat_terms_fwd_kernel, dim3((M + 255) / 256), dim3(256), 0, ST(stream), probs, ldp, M, Y,
                       labels, prior, ldpr, logp, kl, ldk, ent, best);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_cat_terms_bwd(const float* probs, int64_t ldp, int32_t M, int32_t Y, const int32_t* labels,
                                const float* prior, int64_t ldpr, const float* c_logp, const float* g_kl,
                                int64_t ldg, const float* c_ent, float* dprobs, int64_t lddp, float beta,
                                dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && Y >= 1);
    if (M == 0) return DV_OK;
    DV_REQUIRE(probs && dprobs);
    DV_REQUIRE(c_logp == nullptr || labels != nullptr);
    DV_REQUIRE(g_kl == nullptr || prior != nullptr);
    hipLaunchKernelGGL(cat_terms_bwd_kernel, dim3((M + 255) / 256), dim3(256), 0, ST(stream), probs, ldp, M, Y,
                       labels, prior, ldpr, c_logp, g_kl, ldg, c_ent, dprobs, lddp, beta);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_smalln_linear_fwd(const float* a1, int64_t lda1, int32_t K1, const float* a2, int64_t lda2,
                                    int32_t K2, const float* W, int64_t ldw, const float* bias, int32_t M, int32_t N,
                                    float* logits, int64_t ldl, float* probs, int64_t ldp, const dv_ymarg* ymarg,
                                    const dv_wait* park_in, const dv_fprop_kl* kf_in, dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && N >= 1 && N <= kMaxSmallN && K1 >= 0 && K2 >= 0 && park_ok(park_in));
    const ParkArgs park = park_in ? *park_in : ParkArgs{};
    DV_REQUIRE(park.flag == nullptr || M > 0);
    if (park.flag != nullptr && (M + 3) / 4 > DV_MAX_PARKED_GRID) return DV_ERR_UNSUPPORTED;
    if (M == 0) return DV_OK;
    DV_REQUIRE(a1 && W && (a2 || K2 == 0) && (logits || probs));
    dv_ymarg ym{};
    if (ymarg != nullptr && ymarg->fp_ptr != nullptr) {
        ym = *ymarg;
        DV_REQUIRE(probs && ym.label && ym.klfp && ym.c_kld && ym.c_yl && ym.yl && ym.kld && ym.cfp && ym.dqy);
    }
    dv_fprop_kl kf{};
    if (kf_in != nullptr && kf_in->mu_q != nullptr) {
        kf = *kf_in;
        DV_REQUIRE(ym.fp_ptr != nullptr && kf.klfp == ym.klfp && kf.qidx && kf.mu_p && kf.mu3 && kf.raw1 && kf.raw3 &&
                   kf.dq && kf.dp && kf.Z1 >= 0 && kf.Z3 >= 0);
    }
    if (kf.mu_q != nullptr && probs != nullptr && N <= 2 && K1 + K2 <= 256 && kf.Z1 <= 128 && kf.Z3 <= 128) {
        // (the train step's launch at its common shape: every operand in flight first, see smalln_fwd2_kernel)
        hipLaunchKernelGGL(smalln_fwd2_kernel, dim3((M + 3) / 4), dim3(256), 0, ST(stream), a1, lda1, K1, a2, lda2, K2, W,
                           ldw, bias, M, N, logits, ldl, probs, ldp, ym, park, kf);
        DV_RETURN_LAUNCH();
    }
    hipLaunchKernelGGL(smalln_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, ST(stream), a1, lda1, K1, a2, lda2, K2, W,
                       ldw, bias, M, N, logits, ldl, probs, ldp, ym, park, kf);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_smalln_linear_bwd_data(const float* dprobs, int64_t lddp, const float* probs, int64_t ldp,
                                         const float* W, int64_t ldw, int32_t M, int32_t N, int32_t n_dst,
                                         float* const* dst, const int64_t* ld, const int32_t* col0,
                                         const int32_t* ncol, const float* alpha, const float* beta,
                                         const int32_t* col1, const float* alpha2, const float* seg_src,
                                         int64_t ld_seg, const int32_t* seg_ptr, dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && N >= 1 && N <= kMaxSmallN && n_dst >= 1 && n_dst <= 3);
    DV_REQUIRE((seg_src == nullptr) == (seg_ptr == nullptr));
    if (M == 0) return DV_OK;
    DV_REQUIRE(dprobs && W && dst && ld && col0 && ncol && alpha && beta);
    SmallNDst d;
    d.n = n_dst;
    d.seg_src = seg_src;
    d.ld_seg = ld_seg;
    d.seg_ptr = seg_ptr;
    int cols = 0;
    for (int t = 0; t < n_dst; ++t) {
        DV_REQUIRE(dst[t] != nullptr && ncol[t] >= 0);
        d.dst[t] = dst[t];
        d.ld[t] = ld[t];
        d.col0[t] = col0[t];
        d.ncol[t] = ncol[t];
        d.alpha[t] = alpha[t];
        d.beta[t] = beta[t];
        d.col1[t] = col1 ? col1[t] : 0;
        d.alpha2[t] = alpha2 ? alpha2[t] : 0.f;
        cols += ncol[t];
    }
    if (cols == 0) return DV_OK;
    hipLaunchKernelGGL(smalln_bwd_data_kernel, dim3(grid_for((int64_t)M * cols, 256)), dim3(256), 0, ST(stream),
                       dprobs, lddp, probs, ldp, probs != nullptr, W, ldw, M, N, d);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_smalln_linear_bwd_weight(const float* dprobs, int64_t lddp, const float* probs, int64_t ldp,
                                           const float* a1, int64_t lda1, int32_t K1, const float* a2,
                                           int64_t lda2, int32_t K2, int32_t M, int32_t N, float* dW, int64_t ldd,
                                           float* db, float beta, const dv_publish* pub_in, float* ws, int32_t ws_splits,
                                           dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && N >= 1 && N <= kMaxSmallN && K1 >= 0 && K2 >= 0);
    DV_REQUIRE(dprobs && a1 && dW && (a2 || K2 == 0) && ws_splits >= 0 && (ws || ws_splits <= 1));
    dv_publish pub = pub_in ? *pub_in : dv_publish{nullptr, nullptr, 0};
    DV_REQUIRE(pub.flag == nullptr || pub.ctr != nullptr);
    const int col_blocks = (K1 + K2 + 1 + kSnCols - 1) / kSnCols;
    // rows split over workgroups only where a column block's workgroup would walk many rows (>= 1024) on a mostly idle
    // chip; the caller's workspace holds ws_splits x N x (K1 + K2 + 1) floats
    int splits = (ws && ws_splits > 1) ? (M + 255) / 256 : 1;
    if (splits > ws_splits) splits = ws_splits > 0 ? ws_splits : 1;
    if (M < 1024 || splits < 2) splits = 1;
    const int rps = splits > 1 ? (M + splits - 1) / splits : (M > 0 ? M : 1);
    hipLaunchKernelGGL(smalln_bwd_weight_kernel, dim3(col_blocks, splits), dim3(1024), 0,
                       ST(stream), dprobs,
                       lddp, probs, ldp, probs != nullptr, a1, lda1, K1, a2, lda2, K2, M, N, dW, ldd, db, beta, pub,
                       splits > 1 ? ws : (float*)nullptr, rps);
    if (splits > 1)
        hipLaunchKernelGGL(smalln_wgrad_reduce_kernel, dim3((N * (K1 + K2 + 1) + 255) / 256), dim3(256), 0, ST(stream), ws,
                           splits, N, K1 + K2, dW, ldd, db, beta);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_ymarg_fwd(const float* qy, int64_t ldq, const int32_t* label, const int32_t* fp_ptr,
                            const float* klfp, float log_prior, const float* log_prior_v, int32_t R, int32_t Y,
                            float* yl, float* kld, dv_stream_t stream) {
    DV_REQUIRE(R >= 0 && Y >= 1);
    if (R == 0) return DV_OK;
    DV_REQUIRE(qy && label && fp_ptr && klfp && yl && kld);
    hipLaunchKernelGGL(ymarg_fwd_kernel, dim3((R + 255) / 256), dim3(256), 0, ST(stream), qy, ldq, label, fp_ptr,
                       klfp, log_prior, log_prior_v, R, Y, yl, kld);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_ymarg_fwdbwd(const float* qy, int64_t ldq, const int32_t* label, const int32_t* fp_ptr,
                               const float* klfp, float log_prior, const float* log_prior_v, const float* c_kld,
                               const float* c_yl, int32_t R, int32_t Y, float* yl, float* kld, float* cfp,
                               float* dqy, int64_t lddq, dv_stream_t stream) {
    DV_REQUIRE(R >= 0 && Y >= 1);
    if (R == 0) return DV_OK;
    DV_REQUIRE(qy && label && fp_ptr && klfp && c_kld && c_yl && yl && kld && cfp && dqy);
    hipLaunchKernelGGL(ymarg_fwdbwd_kernel, dim3((R + 255) / 256), dim3(256), 0, ST(stream), qy, ldq, label, fp_ptr,
                       klfp, log_prior, log_prior_v, c_kld, c_yl, R, Y, yl, kld, cfp, dqy, lddq);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_ymarg_bwd(const float* qy, int64_t ldq, const int32_t* label, const int32_t* fp_ptr,
                            const float* klfp, float log_prior, const float* log_prior_v, const float* c_kld,
                            const float* c_yl, int32_t R, int32_t Y, float* cfp, float* dqy, int64_t lddq,
                            dv_stream_t stream) {
    DV_REQUIRE(R >= 0 && Y >= 1);
    if (R == 0) return DV_OK;
    DV_REQUIRE(qy && label && fp_ptr && klfp && c_kld && c_yl && cfp && dqy);
    hipLaunchKernelGGL(ymarg_bwd_kernel, dim3((R + 255) / 256), dim3(256), 0, ST(stream), qy, ldq, label, fp_ptr,
                       klfp, log_prior, log_prior_v, c_kld, c_yl, R, Y, cfp, dqy, lddq);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_ycont_fwd(const float* mu, int64_t ldm, const float* ylab, const int32_t* has_y, const float* eps,
                            int64_t lde, float logvar, int32_t sqerr, int32_t R, int32_t B, int32_t Y, float* yl,
                            float* fpin_y, int64_t ld1, float* z3in_y, int64_t ld2, dv_stream_t stream) {
    DV_REQUIRE(R >= 0 && B >= 1 && Y >= 1);
    if (R == 0) return DV_OK;
    DV_REQUIRE(mu && ylab && has_y && eps && yl && fpin_y && z3in_y);
    hipLaunchKernelGGL(ycont_fwd_kernel, dim3((R + 255) / 256), dim3(256), 0, ST(stream), mu, ldm, ylab, has_y, eps, lde,
                       logvar, sqerr, R, B, Y, yl, fpin_y, ld1, z3in_y, ld2);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_ycont_bwd(const float* mu, int64_t ldm, const float* ylab, const int32_t* has_y, float logvar,
                            int32_t sqerr, const float* c_yl, const float* c_kld, const float* dfpin_y, int64_t ld1,
                            const float* dz3in_y, int64_t ld2, int32_t R, int32_t B, int32_t Y, float* dlogit,
                            int64_t ldd, float* cfp, dv_stream_t stream) {
    DV_REQUIRE(R >= 0 && B >= 1 && Y >= 1);
    if (R == 0) return DV_OK;
    DV_REQUIRE(has_y && ((cfp && c_kld) || (mu && ylab && c_yl && dfpin_y && dz3in_y && dlogit)));
    hipLaunchKernelGGL(ycont_bwd_kernel, dim3((R + 255) / 256), dim3(256), 0, ST(stream), mu, ldm, ylab, has_y, logvar,
                       sqerr, c_yl, c_kld, dfpin_y, ld1, dz3in_y, ld2, R, B, Y, dlogit, ldd, cfp, dlogit == nullptr ? 1 : 0);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_mmd_rff_fwd(const float* th1, int64_t ld1, int32_t n1, const float* th2, int64_t ld2, int32_t n2,
                              int32_t R, float c, float* diff, float* mmd2, dv_stream_t stream) {
    DV_REQUIRE(n1 >= 1 && n2 >= 1 && R >= 1);
    DV_REQUIRE(th1 && th2 && diff && mmd2);
    hipLaunchKernelGGL(mmd_cos_means_kernel, dim3((R + 63) / 64), dim3(256), 0, ST(stream), th1, ld1, n1, th2, ld2, n2, R,
                       c, diff);
    hipLaunchKernelGGL(mmd_sumsq_kernel, dim3(1), dim3(256), 0, ST(stream), diff, R, mmd2);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_mmd_rff_bwd(const float* th, int64_t ld, int32_t n, int32_t R, const float* diff, const float* gout,
                              float coef, float* G, int64_t ldg, dv_stream_t stream) {
    DV_REQUIRE(n >= 1 && R >= 1);
    DV_REQUIRE(th && diff && gout && G);
    hipLaunchKernelGGL(mmd_dtheta_kernel, dim3(grid_for((int64_t)n * R, 256)), dim3(256), 0, ST(stream), th, ld, n, R,
                       diff, gout, coef, G, ldg);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_rows_gather(const float* src, int64_t lds, const int32_t* idx, int32_t n, int32_t W,
                              const float* noise, int64_t ldn, float sigma, const int32_t* onehot_cls, int32_t Y,
                              float* out, int64_t ldo, const dv_wait* park_in, dv_stream_t stream) {
    DV_REQUIRE(n >= 0 && W >= 0 && Y >= 0 && park_ok(park_in));
    const ParkArgs park = park_in ? *park_in : ParkArgs{};
    const int64_t total = (int64_t)n * (W + (onehot_cls ? Y : 0));
    DV_REQUIRE(park.flag == nullptr || total > 0);
    if (n == 0) return DV_OK;
    DV_REQUIRE(out && (src || W == 0));
    if (total == 0) return DV_OK;
    // (a parked launch polls from every workgroup: its grid stays small -- 128 workgroups, the element loop strides --
    // so that the polling does not crowd the chain it waits for)
    hipLaunchKernelGGL(rows_gather_kernel, dim3(grid_for(total, 256, park.flag != nullptr ? 128 : 4096)),
                       dim3(256), 0, ST(stream), src, lds, idx, n, W, noise, ldn, sigma, onehot_cls, Y, out, ldo, park);
    DV_RETURN_LAUNCH();
}

static bool masks_ok(const dv_batch_masks_desc& m, int B) {
    return m.Np >= 0 && m.Np <= B && m.n_tot > 0.f && m.c_nll && m.w_recl && (m.hx == nullptr || (m.c_klz2 && m.w_pert)) &&
           (m.hy == nullptr || (m.y && m.c_yl && m.w_yl && m.label));
}

static MaskArgs mask_args(const dv_batch_masks_desc& m, const int32_t* table, int n_batches, const int32_t* ctr,
                          const int32_t* base, int B, int L) {
    return MaskArgs{table, n_batches, ctr, base, m.hx, m.hy, m.y, B, L, m.Np, m.n_tot, m.kl_rate, m.pert_rate, m.yl_rate,
                    m.beta, m.c_nll, m.c_klz2, m.c_yl, m.w_recl, m.w_pert, m.w_yl, m.label, m.c_klp, m.one_slot, m.gcounts};
}

extern "C" int dv_batch_feed(const dv_batch_feed_desc* dsc, const dv_batch_masks_desc* masks, const dv_wait* park_in,
                             dv_stream_t stream) {
    DV_REQUIRE(dsc != nullptr && park_ok(park_in));
    const dv_batch_feed_desc& d = *dsc;
    const ParkArgs park = park_in ? *park_in : ParkArgs{};
    const int B = d.B, Np = d.Np, L = d.L, Mf = d.Mf;
    DV_REQUIRE(B >= 0 && Np >= 0 && d.X >= 0 && d.n_batches >= 1 && L >= 1 && Mf >= 0 && d.Y >= 0 && d.Yc >= 0);
    DV_REQUIRE(!d.ylab || (d.yf && d.Yc >= 1));
    if (B == 0) return DV_OK;
    DV_REQUIRE(d.x1 && d.table && d.ctr && d.base && d.xin && (Np == 0 || (d.x2 && d.pair_rows)));
    DV_REQUIRE(!d.label_r || (d.y && d.has_y));
    DV_REQUIRE(!d.fp_cls || Mf == 0 || (d.y && d.fp_i && d.fp_lab && d.fp_slot));
    const int row_blocks = (B + Np + 3) / 4;
    int nlab = (d.label_r ? L * B : 0) > (d.fp_cls ? Mf : 0) ? (d.label_r ? L * B : 0) : (d.fp_cls ? Mf : 0);
    if (d.ylab && B * d.Yc > nlab) nlab = B * d.Yc;
    const int lab_blocks = (nlab + 255) / 256;
    const bool v4 = aligned16(d.x1) && (Np == 0 || aligned16(d.x2)) && aligned16(d.xin) && (!d.noise || aligned16(d.noise)) &&
                    d.ld1 % 4 == 0 && (Np == 0 || d.ld2 % 4 == 0) && d.ldo % 4 == 0 && (!d.noise || d.ldn % 4 == 0);
    MaskArgs ma{};
    if (masks) {
        DV_REQUIRE(masks_ok(*masks, B));
        ma = mask_args(*masks, d.table, d.n_batches, d.ctr, d.base, B, L);
    }
    const int feed_blocks = row_blocks + lab_blocks;
    if (park.flag != nullptr && feed_blocks + 1 > DV_MAX_PARKED_GRID) return DV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(batch_feed_kernel, dim3(feed_blocks + (masks ? 1 : 0)), dim3(256), 0, ST(stream), d.x1, d.ld1, d.x2,
                       d.ld2, d.y, d.table, d.n_batches, d.ctr, d.base, B, d.pair_rows, Np, d.X, d.noise, d.ldn, d.sigma,
                       d.xin, d.ldo, d.has_y, L, d.label_r, d.fp_i, d.fp_lab, d.fp_slot, Mf, d.fp_cls, d.onehot, d.ldh,
                       d.Y, row_blocks, v4 ? 1 : 0, d.yf, d.ylab, d.Yc, d.onehot2, d.ldh2, masks ? feed_blocks : -1, ma,
                       park);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_batch_masks(const dv_batch_masks_desc* m, const int32_t* table, int32_t n_batches, const int32_t* ctr,
                              const int32_t* base, int32_t B, int32_t L, dv_stream_t stream) {
    DV_REQUIRE(m != nullptr && B >= 1 && L >= 1 && masks_ok(*m, B));
    DV_REQUIRE(table == nullptr || (ctr && base && n_batches >= 1));
    MaskArgs a = mask_args(*m, table, n_batches, ctr, base, B, L);
    hipLaunchKernelGGL(batch_masks_kernel, dim3(1), dim3(1024), 0, ST(stream), a);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_rows_segment_sum(const float* src, int64_t lds, const int32_t* seg_ptr, const int32_t* seg_rows,
                                   const float* w, int32_t n, int32_t W, const int32_t* dst_idx, float* dst,
                                   int64_t ldd, float beta, const dv_wait* park_in, dv_stream_t stream) {
    DV_REQUIRE(park_ok(park_in));
    const ParkArgs park = park_in ? *park_in : ParkArgs{};
    DV_REQUIRE(n >= 0 && W >= 0);
    DV_REQUIRE(park.flag == nullptr || (n > 0 && W > 0));
    if (park.flag != nullptr && grid_for((int64_t)n * W, 256) > DV_MAX_PARKED_GRID) return DV_ERR_UNSUPPORTED;
    if (n == 0 || W == 0) return DV_OK;
    DV_REQUIRE(src && dst);
    hipLaunchKernelGGL(rows_segment_sum_kernel, dim3(grid_for((int64_t)n * W, 256)), dim3(256), 0, ST(stream), src,
                       lds, seg_ptr, seg_rows, w, n, W, dst_idx, dst, ldd, beta, park);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_weighted_sum(const float* x, const float* w, const int32_t* idx, int32_t n, float scale,
                               float* out, float beta, dv_stream_t stream) {
    DV_REQUIRE(n >= 0 && out);
    DV_REQUIRE(n == 0 || x);
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(256), 0, ST(stream), x, w, idx, n, scale, out, beta);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_recon_row_stats(const float* x, int64_t ldx, const float* r, int64_t ldr, int32_t M, int32_t X,
                                  float* out, dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && X >= 1);
    if (M == 0) return DV_OK;
    DV_REQUIRE(x && r && out);
    hipLaunchKernelGGL(recon_row_stats_kernel, dim3(grid_for(M, 4, 8192)), dim3(256), 0, ST(stream), x, ldx, r, ldr,
                       M, X, out);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_col_moments(const float* x, int64_t ldx, const float* r, int64_t ldr, int32_t M, int32_t X,
                              double* out, int32_t row_blocks, const int32_t* sel, const float* r_bias,
                              dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && X >= 1 && row_blocks >= 1 && row_blocks <= 65535);
    DV_REQUIRE(x && r && out);
    const int rpb = (M + row_blocks - 1) / row_blocks;
    hipLaunchKernelGGL(col_moments_kernel, dim3((X + 63) / 64, row_blocks), dim3(64 * kCmRG), 0, ST(stream), x, ldx, r, ldr, M, X, out,
                       rpb > 0 ? rpb : 1, sel, r_bias);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_recon_rows(const dv_recon_rows_desc* dsc, dv_stream_t stream) {
    DV_REQUIRE(dsc != nullptr);
    const dv_recon_rows_desc& d = *dsc;
    DV_REQUIRE(d.M >= 0 && d.X >= 1);
    if (d.X > DV_RECON_ROWS_MAX_X) return DV_ERR_UNSUPPORTED;
    if (d.M == 0) return DV_OK;
    DV_REQUIRE(d.x && d.mu && d.sd && d.rows);
    DV_REQUIRE((d.bias_mu == nullptr) == (d.bias_sd == nullptr));
    RcArgs a{d.x, d.ldx, d.mu, d.sd, d.ldp, d.bias_mu, d.bias_sd, d.sd_shift, d.M, d.X, d.rows, d.ll};
    const dim3 grid(grid_for(d.M, 4, 8192)), block(256);
    auto a8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
    const bool v2 = d.X % 2 == 0 && d.ldx % 2 == 0 && d.ldp % 2 == 0 && a8(d.x) && a8(d.mu) && a8(d.sd) &&
                    (!d.bias_mu || (a8(d.bias_mu) && a8(d.bias_sd)));
    if (d.bias_mu) {
        if (v2) hipLaunchKernelGGL((recon_rows_kernel<true, true>), grid, block, 0, ST(stream), a);
        else hipLaunchKernelGGL((recon_rows_kernel<true, false>), grid, block, 0, ST(stream), a);
    } else {
        if (v2) hipLaunchKernelGGL((recon_rows_kernel<false, true>), grid, block, 0, ST(stream), a);
        else hipLaunchKernelGGL((recon_rows_kernel<false, false>), grid, block, 0, ST(stream), a);
    }
    DV_RETURN_LAUNCH();
}

static int mix_args(MixArgs& a, const float* G, int64_t ldg, int32_t M, int32_t N, int32_t kind, const float* gammas,
                    int32_t nb, const float* sa, int64_t sa_stride, const float* sb, int64_t sb_stride) {
    DV_REQUIRE(G && M >= 1 && N >= 1 && (kind == 0 || kind == 1) && gammas && nb >= 1 && nb <= 8);
    DV_REQUIRE(kind == 0 || (sa && sb));
    a = MixArgs{G, ldg, M, N, kind, nb, {0}, sa, sa_stride, sb, sb_stride};
    for (int b = 0; b < nb; ++b) a.gam[b] = gammas[b];
    return DV_OK;
}

extern "C" int dv_mmd_mix_fwd(const float* G, int64_t ldg, int32_t M, int32_t N, int32_t kind, const float* gammas,
                              int32_t nb, const float* sa, int64_t sa_stride, const float* sb, int64_t sb_stride,
                              float* part, dv_stream_t stream) {
    MixArgs a;
    const int rc = mix_args(a, G, ldg, M, N, kind, gammas, nb, sa, sa_stride, sb, sb_stride);
    if (rc != DV_OK) return rc;
    DV_REQUIRE(part != nullptr);
    hipLaunchKernelGGL(mmd_mix_kernel<false>, dim3(M), dim3(256), 0, ST(stream), a, part, (const float*)nullptr, 0.f,
                       (float*)nullptr, (int64_t)0, (float*)nullptr);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_mmd_mix_bwd(const float* G, int64_t ldg, int32_t M, int32_t N, int32_t kind, const float* gammas,
                              int32_t nb, const float* sa, int64_t sa_stride, const float* sb, int64_t sb_stride,
                              const float* gout, float coef, float* W, int64_t ldw, float* rs, dv_stream_t stream) {
    MixArgs a;
    const int rc = mix_args(a, G, ldg, M, N, kind, gammas, nb, sa, sa_stride, sb, sb_stride);
    if (rc != DV_OK) return rc;
    DV_REQUIRE(gout && W && rs);
    hipLaunchKernelGGL(mmd_mix_kernel<true>, dim3(M), dim3(256), 0, ST(stream), a, (float*)nullptr, gout, coef, W, ldw, rs);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_mmd_mix_combine(const float* p11, int32_t n11, float c11, const float* p12, int32_t n12, float c12,
                                  const float* p22, int32_t n22, float c22, float* out, dv_stream_t stream) {
    DV_REQUIRE(p11 && p12 && p22 && out && n11 >= 1 && n12 >= 1 && n22 >= 1 && c11 > 0.f && c12 > 0.f && c22 > 0.f);
    hipLaunchKernelGGL(mmd_mix_combine_kernel, dim3(1), dim3(64), 0, ST(stream), p11, n11, c11, p12, n12, c12, p22, n22, c22, out);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_mmd_identity_fwd(const float* x1, int64_t ld1, int32_t n1, const float* x2, int64_t ld2, int32_t n2,
                                   int32_t Z, float* diff, float* out, dv_stream_t stream) {
    DV_REQUIRE(x1 && x2 && diff && out && n1 >= 1 && n2 >= 1 && Z >= 1);
    hipLaunchKernelGGL(mmd_identity_fwd_kernel, dim3(1), dim3(256), 0, ST(stream), x1, ld1, n1, x2, ld2, n2, Z, diff, out);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_mmd_identity_bwd(const float* diff, const float* gout, float coef, int32_t n, int32_t Z, float* dx,
                                   int64_t ldd, dv_stream_t stream) {
    DV_REQUIRE(diff && gout && dx && n >= 1 && Z >= 1);
    hipLaunchKernelGGL(mmd_identity_bwd_kernel, dim3(grid_for((int64_t)n * Z, 256)), dim3(256), 0, ST(stream), diff, gout,
                       coef, n, Z, dx, ldd);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_recon_finalize(const float* rows, const int32_t* sel, int32_t n, int32_t X, const double* cols,
                                 int32_t row_blocks, const float* ll, double* out, dv_stream_t stream) {
    DV_REQUIRE(n >= 0 && X >= 1 && row_blocks >= 1 && rows && cols && out);
    hipLaunchKernelGGL(recon_finalize_kernel, dim3(1), dim3(1024), 0, ST(stream), rows, sel, n, X, cols, row_blocks, ll, out);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_rank_metrics(const float* proba, int64_t ldp, const int32_t* y, const int32_t* pred, const int32_t* sel,
                               int32_t n, int32_t c0, int32_t n_cls, int32_t binary, int32_t* counts, double* out,
                               dv_stream_t stream) {
    DV_REQUIRE(n >= 0 && c0 >= 0 && n_cls >= 1 && n_cls <= 64 && proba && y && counts && out);
    if (n > DV_RANK_MAX_ROWS) return DV_ERR_UNSUPPORTED;
    if (n > 0)
        hipLaunchKernelGGL(pair_counts_kernel, dim3((n + 255) / 256, (n + kPairJ - 1) / kPairJ, n_cls), dim3(256), 0,
                           ST(stream), proba, ldp, y, sel, n, c0, binary, counts);
    hipLaunchKernelGGL(rank_finalize_kernel, dim3(n_cls), dim3(1024), 0, ST(stream), counts, y, pred, sel, n, c0, binary,
                       n_cls, out);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_bn_fwd(const float* x, int64_t ldx, int32_t M, int32_t N, const float* w, const float* b, float eps,
                         float* mean_out, float* rstd_out, float* y, int64_t ldy, float* running_mean,
                         float* running_var, float momentum, int32_t training, dv_stream_t stream) {
    DV_REQUIRE(M >= 1 && N >= 1 && x && y && mean_out && rstd_out && eps >= 0.f);
    DV_REQUIRE(training || (running_mean && running_var));
    hipLaunchKernelGGL(bn_fwd_kernel, dim3((N + 63) / 64), dim3(256), 0, ST(stream), x, ldx, M, N, w, b, eps, mean_out,
                       rstd_out, y, ldy, running_mean, running_var, momentum, training);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_bn_bwd(const float* dy, int64_t ldd, const float* x, int64_t ldx, const float* mean, const float* rstd,
                         const float* w, int32_t M, int32_t N, float* dx, int64_t lddx, float* dw, float* db,
                         int32_t training, dv_stream_t stream) {
    DV_REQUIRE(M >= 1 && N >= 1 && dy && x && mean && rstd);
    hipLaunchKernelGGL(bn_bwd_kernel, dim3((N + 63) / 64), dim3(256), 0, ST(stream), dy, ldd, x, ldx, mean, rstd, w, M, N,
                       dx, lddx, dw, db, training);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_mask_scale(const float* x, int64_t ldx, const float* mask, int64_t ldm, float scale, int32_t M,
                             int32_t N, float* y, int64_t ldy, dv_stream_t stream) {
    DV_REQUIRE(M >= 0 && N >= 0);
    if (M == 0 || N == 0) return DV_OK;
    DV_REQUIRE(x && mask && y);
    hipLaunchKernelGGL(mask_scale_kernel, dim3(grid_for((int64_t)M * N, 256)), dim3(256), 0, ST(stream), x, ldx, mask,
                       ldm, scale, M, N, y, ldy);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_loss_assemble(const dv_loss_term* terms, int32_t n_terms, const float* w_elbo,
                                const float* w_cmpl, float* loss, const int32_t* halt, int32_t n_halt,
                                float* accum, dv_stream_t stream) {
    DV_REQUIRE(n_terms >= 0 && n_terms <= DV_MAX_LOSS_TERMS && (terms || n_terms == 0));
    DV_REQUIRE(n_halt >= 0 && (halt || n_halt == 0));
    DV_REQUIRE(w_elbo && w_cmpl && loss);
    LossTerms lt;
    lt.n = n_terms;
    for (int i = 0; i < n_terms; ++i) {
        DV_REQUIRE(terms[i].out >= 0 && terms[i].out < 5 && terms[i].n >= 0 && (terms[i].x || terms[i].n == 0));
        lt.t[i] = terms[i];
    }
    hipLaunchKernelGGL(loss_assemble_kernel, dim3(1), dim3(kLossThreads), 0, ST(stream), lt, w_elbo, w_cmpl, loss,
                       (int32_t*)nullptr, (const int32_t*)nullptr, 0, (int32_t*)nullptr, 0, CounterBump{}, halt, n_halt,
                       accum);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_loss_assemble_after(const dv_wait* wait, const dv_loss_term* terms, int32_t n_terms,
                                      const float* w_elbo, const float* w_cmpl, float* loss, const dv_bump* bump_in,
                                      const int32_t* halt, int32_t n_halt, float* accum, dv_stream_t stream) {
    DV_REQUIRE(n_halt >= 0 && (halt || n_halt == 0));
    DV_REQUIRE(n_terms >= 0 && n_terms <= DV_MAX_LOSS_TERMS && (terms || n_terms == 0));
    const dv_wait w = wait ? *wait : dv_wait{};
    DV_REQUIRE(w_elbo && w_cmpl && loss && (!w.flag || (w.ctr && w.err && w.max_spins > 0)));     // flag == NULL: no wait
    const CounterBump bump = bump_in ? *bump_in : CounterBump{};
    for (int i = 0; i < 2; ++i) DV_REQUIRE(!bump.c[i] || bump.n[i] == 1 || bump.n[i] == 2);
    LossTerms lt;
    lt.n = n_terms;
    for (int i = 0; i < n_terms; ++i) {
        DV_REQUIRE(terms[i].out >= 0 && terms[i].out <= 4 && terms[i].n >= 0 && (terms[i].x || terms[i].n == 0));
        lt.t[i] = terms[i];
    }
    hipLaunchKernelGGL(loss_assemble_kernel, dim3(1), dim3(kLossThreads), 0, ST(stream), lt, w_elbo, w_cmpl, loss, w.flag,
                       w.ctr, w.add, w.err, w.max_spins, bump, halt, n_halt, accum);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_axpby(const float* x, float a, float* y, float b, int64_t n, dv_stream_t stream) {
    DV_REQUIRE(n >= 0);
    if (n == 0) return DV_OK;
    DV_REQUIRE(x && y);
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(stream), x, a, y, b, n);
    DV_RETURN_LAUNCH();
}
