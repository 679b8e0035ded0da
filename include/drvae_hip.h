/* drvae_hip.h -- C-ABI of libdrvae_hip.so: the Dr.VAE ELBO training hot path as
 * hand-written HIP kernels for gfx950 (MI355X / CDNA4).
 *
 * The reference (rampasek/DrVAE, pure Python on PyTorch CPU ATen, no native layer,
 * no FFI) reaches this arithmetic only through `import blocks as blk` /
 * `import layers as lyr` (src/DrVAE.py:21, src/blocks.py:12).  The Python modules
 * `drvae_amd.blocks` / `drvae_amd.layers` mirror that interface and call the entry
 * points below through ctypes (INTEGRATION.md shows the stub).  Each entry point
 * cites the reference expression it replaces.
 *
 * Conventions
 *  - plain C: raw DEVICE pointers, int sizes, leading dimensions in ELEMENTS,
 *    `dv_stream_t` = hipStream_t passed as void*.  No torch types.
 *  - all matrices fp32 row-major with unit inner stride; index arrays int32.
 *  - every call only ENQUEUES work on `stream`: no allocation, no host sync, no
 *    hidden state (nothing is armed, cached or remembered between calls: whatever a launch
 *    waits on or advances is an explicit argument of THAT call) -> re-entrant and
 *    hipGraph-capturable (SURVEY.md 8(b) threading).
 *  - returns DV_OK (0) or a negative DV_ERR_* code; never throws, never aborts.
 *  - `beta` arguments: out = beta*out + result (beta == 0 never reads out).
 */
#ifndef DRVAE_HIP_H
#define DRVAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dv_stream_t;

#define DV_ABI_VERSION 12

enum { DV_OK = 0, DV_ERR_ARG = -1, DV_ERR_LAUNCH = -2, DV_ERR_UNSUPPORTED = -3 };

/* activation ids -- the `nonlinearities` table of src/blocks.py:21-24 (+ cos for the
 * random-Fourier MMD features of src/blocks.py:53-54; forward only) */
enum {
    DV_ACT_IDENTITY = 0, DV_ACT_ELU = 1, DV_ACT_SOFTPLUS = 2, DV_ACT_SIGMOID = 3, DV_ACT_TANH = 4,
    DV_ACT_RELU = 5, DV_ACT_LEAKY_RELU = 6, DV_ACT_SELU = 7, DV_ACT_SOFTSIGN = 8, DV_ACT_COS = 9
};

/* Gaussian parametrisation: GaussianLogVarMixin (src/blocks.py:166-202) or
 * GaussianSigmaMixin (src/blocks.py:204-240) */
enum { DV_GAUSS_LOGVAR = 0, DV_GAUSS_SIGMA = 1 };

enum { DV_EPI_PLAIN = 0, DV_EPI_FWD = 1, DV_EPI_BWD = 2, DV_EPI_KLQ = 3 };

int dv_abi_version(void);
const char* dv_error_string(int code);
/* hex sha256 over the sources this binary was built from (every file of drvae_amd/csrc/ that is compiled or included,
 * in name order, + this header; `drvae_amd.build.source_hash()` recomputes it from a tree) -- "unhashed" for a build that
 * did not go through drvae_amd/build.py.  The reference has no native layer, hence no counterpart. */
const char* dv_source_hash(void);

/* A device-side wait carried by a launch (see dv_flag_wait): the launch first parks until
 * flag[0] >= ctr[0] + add.  Bounded: after max_spins polls it records err[0] = 1 (sticky: nothing on
 * the device clears it) and goes on; err[1] accumulates the time parked (wall_clock64 ticks).  Every
 * consumer of err words -- dv_loss_assemble*, dv_adam_l2*, dv_adamax_l2 through their `halt`
 * argument -- then poisons the loss scalars with NaN and freezes the parameters, so a timed-out wait
 * can never train on stale data silently.  flag == NULL: no wait. */
typedef struct dv_wait {
    int32_t* flag;
    const int32_t* ctr;
    int32_t add;
    int32_t max_spins;
    int32_t* err;
} dv_wait;

/* "this launch has started, so everything in front of it in its stream is complete": a launch that carries a
 * dv_publish stores flag[0] = ctr[0] + add (release, agent scope) on entry, like dv_flag_publish but without a
 * launch of its own (dv_gemm_desc has the same three fields).  flag == NULL / NULL pointer: nothing published */
typedef struct dv_publish {
    int32_t* flag;
    const int32_t* ctr;
    int32_t add;
} dv_publish;

/* The KL rows of the fprop rows riding on the classifier-head launch, in front of and behind its y-marginalisation
 * (dv_smalln_linear_fwd with a dv_ymarg AND a dv_fprop_kl argument; train step): for every fprop row t of classifier
 * row r (fp_ptr of the dv_ymarg): raw1[t] = KL(q(z1|x)[qidx[t]] || p(z1|z3,y)[t]), raw3[t] = KL(q(z3|z1,y)[t] || N(0,I)),
 * klfp[t] = max(raw1, kl_min) + max(raw3, kl_min) (dv_kl_rows_fwd with its second term; LOGVAR parametrisation), then the
 * classifier head and the y-marginalisation, then the backward of the z1 term with the coefficient the y-marginalisation
 * has just produced: row-aligned d/d(q) -> dq, d/d(p) -> dp (dv_kl_rows_bwd, free bits, beta = 0).  mu_q == NULL: none */
typedef struct dv_fprop_kl {
    const float* mu_q;         /* (mu | logvar) rows of q(z1|x), ld = ldq; logvar at column offset Z1 of the same row */
    int64_t ldq;
    const int32_t* qidx;
    const float* mu_p;         /* (mu | logvar) of p(z1|z3,y), row t */
    int64_t ldp;
    const float* mu3;          /* (mu | logvar) of q(z3|z1,y), row t; Z3 columns each */
    int64_t ld3;
    int32_t Z1;
    int32_t Z3;
    float kl_min;
    float* klfp;               /* out, per fprop row (the dv_ymarg's klfp must be this buffer) */
    float* raw1;
    float* raw3;
    float* dq;                 /* out (Mf, 2*Z1): d/d(mu_q | logvar_q), row-aligned */
    int64_t lddq;
    float* dp;                 /* out (Mf, 2*Z1): d/d(mu_p | logvar_p) */
    int64_t lddp;
} dv_fprop_kl;

/* up to two device counters (1 or 2 int32 words each: int32 / uint64 little-endian) advanced by a
 * launch that carries the bump (see dv_counters_add2); c == NULL: unused slot */
/* y-marginalisation riding on the classifier-head launch (dv_smalln_linear_fwd): the arguments of dv_ymarg_fwdbwd
 * (src/DrVAE.py:503-534); fp_ptr == NULL: none */
typedef struct dv_ymarg {
    const int32_t* label;
    const int32_t* fp_ptr;
    const float* klfp;
    float log_prior;
    const float* log_prior_v;
    const float* c_kld;
    const float* c_yl;
    float* yl;
    float* kld;
    float* cfp;
    float* dqy;
    int64_t lddq;
} dv_ymarg;

typedef struct dv_bump {
    int32_t* c[2];
    int32_t n[2];
    int64_t inc[2];
} dv_bump;

/* ------------------------------------------------------------------ GEMM family
 * C[M,N] = epilogue( alpha * sum_k Aop[m,k] * Bop[k,n] ) + beta*C      (fp32 MFMA,
 * v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma chain -- no reduced precision).
 *   a_kcontig: Aop[m,k] = A[m*lda + k]   else A[k*lda + m]
 *   b_kcontig: Bop[k,n] = B[n*ldb + k]   else B[k*ldb + n]
 * which covers the three products of a Linear layer y = x W^T (+b):
 *   forward      y  = x  W^T      a_kcontig=1 b_kcontig=1   F.linear, src/layers.py:38,
 *                                                            nn.Linear in src/blocks.py:144,278-279,396-397,446
 *   backward dx  dx = dy W        a_kcontig=1 b_kcontig=0   (autograd AddmmBackward in the reference)
 *   backward dW  dW = dy^T x      a_kcontig=0 b_kcontig=0
 * A2/K1: optional second A source for k >= K1 -- `torch.cat(inputs, 1)` of
 *   src/blocks.py:161 without materialising the concat (a_kcontig only).
 * a_kscale: optional per-k multiplier on Aop (WeightNorm backward: dy * g/||W||).
 *
 * epilogue DV_EPI_FWD (per element, col = n):
 *   v = scale[n]*v (WeightNorm g/||W||, src/layers.py:39-40) + bias[n];
 *   v = act(v) + shift, with (act0,shift0) for n < split and (act1,shift1) otherwise
 *       -- dual heads sharing one GEMM: `logvar = lv(h) - 2.` (src/blocks.py:296,360),
 *          `std = softplus(sg(h)) + 1e-3` (src/blocks.py:401,415);
 *   v += resid[m*ldr+n] for n < resid_cols  -- `mu = x + F.linear(x,W_mu) + b` (src/blocks.py:357).
 * epilogue DV_EPI_BWD: v *= act'(yref[m*ldy+n] - shift) with the same (split, act, shift)
 * epilogue DV_EPI_KLQ (round 5): the product is d/dz of a reparameterised sample z = mu + eps * exp(logvar / 2) of q rows that
 *   also carry a prior term coef[m] * max(KL(q_m || N(0,I)), kl_min): with Z = split, for columns n < Z the launch writes the
 *   gradient w.r.t. (mu | logvar) instead of v -- C[m, n] = c_m * mu + v, C[m, Z + n] = c_m * (e^logvar - 1) / 2 +
 *   v * eps[m, n] * e^(logvar / 2) / 2, c_m = coef[m] * gate(raw[m]) (1 above kl_min, 1/2 on a tie, 0 below) -- what
 *   dv_kl_rows_bwd(dz = v, eps) computes in a launch of its own; columns n >= Z are dropped.  Operands travel in fields this
 *   epilogue does not otherwise use: yref / ldy = the q rows (mu | logvar), resid / ldr = eps, bias = coef (per ROW),
 *   scale = raw (per ROW), shift0 = kl_min; ldc >= 2 Z, beta = 0.
 *   selection: the activation backward of the layer BELOW fused into the dx GEMM.
 */
/* Per-call steering of the GEMM dispatcher (ABI 9; replaces the process-global dv_gemm_force_tiling / dv_gemm_set_option
 * hooks of ABI <= 8).  tiling: see dv_gemm_has_tiling.  opt[0] = workgroup -> tile map (-1 by tiling, 0 linear, 1 XCD
 * chunk-major, >= 2 bands of that many tile rows); opt[1] extra dynamic LDS of the chip-filling 32x32 launches; opt[2] = 1:
 * no fused form of dv_gemm_pair; opt[3]: tiles of 64x64 from which the 64x64 tiling runs (-1: never the hand-pipelined
 * LDS-DMA kernels); opt[4] heads kernel variant; opt[5] > 0 (opt-in, tuning): the 128x256 tiling's second resident workgroup of a CU starts that many percent of an estimated half tile late under a plain epilogue; opt[6] = -1: a chip-filling plain product with a ragged last column of
 * 128x256 tiles is NOT split into [full tiles | narrow rest] (dv_gemm: two launches where that saves a tile per CU); opt[7] = 1: four-wave K split for the k-contiguous layouts; opt[8]: tiles of
 * 32x32 from which the seven-per-CU tiling runs; opt[9] = 1: chip-filling dW || dX as two launches */
typedef struct dv_gemm_tune {
    int32_t tiling;
    int32_t opt[10];
} dv_gemm_tune;

typedef struct dv_gemm_desc {
    int32_t M, N, K;
    int32_t a_kcontig, b_kcontig;
    const float* A;
    int64_t lda;
    const float* A2;
    int64_t lda2;
    int32_t K1;
    const float* a_kscale;
    const float* B;
    int64_t ldb;
    float* C;
    int64_t ldc;
    float alpha, beta;
    int32_t epilogue;
    const float* scale;
    const float* bias;
    int32_t split;
    int32_t act0, act1;
    float shift0, shift1;
    const float* resid;
    int64_t ldr;
    int32_t resid_cols;
    const float* yref;
    int64_t ldy;
    /* optional fused bias gradient (a_kcontig == 0 only, i.e. the dW product where
     * Aop[m,k] = dy[k, m]):  a_colsum[m] = colsum_beta*a_colsum[m] + sum_k Aop[m,k]
     * -- `db = dy.sum(0)` computed from the dy tiles the dW GEMM stages anyway. */
    float* a_colsum;
    float colsum_beta;
    /* bit0 / bit1: reading up to 3 floats past the END OF ANY ROW of A / B stays inside the
     * caller's allocation (rows padded to 4 floats, or followed by more of the same buffer).
     * Lets edge tiles of row-contiguous operands use aligned 16-B loads (the over-read values
     * only feed output elements that are never stored).  0 = never over-read (default). */
    int32_t flags;
    /* optional: on kernel ENTRY (i.e. once everything before this launch in its stream is complete)
     * publish pub_flag[0] = pub_ctr[0] + pub_add like dv_flag_publish -- saves the separate launch. */
    int32_t* pub_flag;
    const int32_t* pub_ctr;
    int32_t pub_add;
    /* optional (tests / tuning; NULL = the dispatcher's heuristics): read on the host at launch time, per call */
    const struct dv_gemm_tune* tune;
} dv_gemm_desc;

int dv_gemm(const dv_gemm_desc* desc, dv_stream_t stream);
/* Two independent products in one launch when both take the 32x32 K-split tiling with the
 * (dy^T x) and (dy W) layouts -- the weight- and data-gradient of one Linear layer, which both
 * only need dy; otherwise exactly dv_gemm(d1) followed by dv_gemm(d2).  At most ONE of the two descriptors may
 * carry a publish (pub_flag): the launch publishes once, on entry (DV_ERR_ARG when both do). */
int dv_gemm_pair(const dv_gemm_desc* d1, const dv_gemm_desc* d2, dv_stream_t stream);
/* Dual-head Linear with the ROW work that consumes both heads fused into the epilogue (SURVEY.md K2+K3 /
 * K2+K5): y = x W^T with W = [W_head0 ; W_head1] (desc->split = rows of head 0, desc->N = 2*split), forward
 * layout (a_kcontig = b_kcontig = 1), epilogue DV_EPI_FWD exactly as dv_gemm (scale / bias / act0,shift0 |
 * act1,shift1 / resid).  Each workgroup computes the tile of head-0 columns [c, c+32) TOGETHER with the
 * head-1 columns [split+c, split+c+32) of the same rows, so element (m, c) sees a0 = head0(m,c) and
 * a1 = head1(m,c) and neither has to round-trip through HBM for the per-element work that follows:
 *
 *  DV_HEADS_SAMPLE  (a0 = mu, a1 = logvar: `DiagGaussianModule.forward` + `sample`, src/blocks.py:291-301,
 *    170-174; `DiagGaussianModuleLinear`, src/blocks.py:349-361): C = (mu | logvar) is written as by dv_gemm,
 *    and for every destination row s of source row m (seg_rows[seg_ptr[m] .. seg_ptr[m+1]); seg_ptr == NULL:
 *    s = m; rows m >= n_src emit nothing):   z = eps[s,c]*exp(logvar/2) + mu;   out[s,c] = z;
 *    out2[s,c] = z - sub[s,c] (optional);  out3[out3_idx[s], c] = z where out3_idx[s] >= 0 (optional)
 *    -- the L Monte-Carlo samples of a row (and the z2 samples of a pair, src/DrVAE.py:421-427) leave the
 *    encoder-heads launch directly.
 *  DV_HEADS_NLL  (a0 = mu, a1 = std = act1(.) + shift1: `DiagGaussianSigmaModule.forward` + `logp_perx`,
 *    src/blocks.py:410-416, 233-234), train step only: with xv = x[xidx ? xidx[m] : m, c]
 *      part[m*n_tiles + t] = -1/2 sum_{c in tile t} [log 2pi + log std^2 + (xv-mu)^2/std^2]   (row NLL = sum_t)
 *      C[m, c]       = coef[m] * (xv-mu)/std^2                                   = d/d mu
 *      C[m, split+c] = coef[m] * (-1/std + (xv-mu)^2/std^3) * act1'(.)           = d/d(pre-activation of std)
 *    i.e. C receives the GRADIENTS (the loss is linear in the row terms with coefficients known up front);
 *    mu / std themselves are never stored.  n_tiles = ceil(split/32) (dv_gemm_heads_tiles). */
enum { DV_HEADS_SAMPLE = 1, DV_HEADS_NLL = 2 };
typedef struct dv_heads_epi {
    int32_t mode;
    /* SAMPLE */
    const int32_t* seg_ptr;
    const int32_t* seg_rows;
    int32_t n_src;
    const float* eps;
    int64_t lde;
    float* out;
    int64_t ldo;
    const float* sub;
    int64_t lds;
    float* out2;
    int64_t ldo2;
    float* out3;
    int64_t ldo3;
    const int32_t* out3_idx;
    float* out4;               /* optional: sample row s is also copied to rows [out4_ptr[s], out4_ptr[s+1]) of out4 */
    int64_t ldo4;              /* (the z1 columns of the fprop rows of that sample, src/DrVAE.py:337-341) */
    const int32_t* out4_ptr;
    /* NLL */
    const float* x;
    int64_t ldx;
    const int32_t* xidx;
    const float* coef;
    float* part;
} dv_heads_epi;
int dv_gemm_heads(const dv_gemm_desc* desc, const dv_heads_epi* epi, dv_stream_t stream);
int dv_gemm_heads_tiles(int32_t split);
/* Does this build of the library carry tiling code t (dv_gemm_tune.tiling)?  0 = heuristic, 1 = 64x64, 2 = 32x32 K-split,
 * 3 = 128x128, 17 = 32x32x32 seven-per-CU, 40 = 128x256x16 hand-pipelined LDS-DMA ring, 46 = 64x64x32 pipelined; any
 * other code names a lab tiling of the tuning build (-DDV_LAB).  A pure function: the library keeps NO tuning state --
 * tests and tools pass a dv_gemm_tune with the descriptor of the call they want to steer. */
int dv_gemm_has_tiling(int tiling);

/* out[n] = beta*out[n] + sum_m X[m*ldx+n]            (bias gradient) */
int dv_colsum(const float* X, int64_t ldx, int32_t M, int32_t N, float* out, float beta, dv_stream_t stream);

/* dY[m,n] *= act'(Y[m,n]-shift), (act,shift) chosen by n<split as in DV_EPI_FWD */
int dv_act_bwd(float* dY, int64_t ldd, const float* Y, int64_t ldy, int32_t M, int32_t N, int32_t split,
               int32_t act0, int32_t act1, float shift0, float shift1, dv_stream_t stream);

/* WeightNorm (src/layers.py:39): norm[n] = ||W[n,:]||_2, scale[n] = g[n]/norm[n] */
int dv_wn_scale(const float* W, int64_t ldw, const float* g, int32_t N, int32_t K, float* scale, float* norm,
                dv_stream_t stream);
/* WeightNorm backward from dWraw = dpre^T x (dpre NOT yet multiplied by scale):
 *   dot = <W[n,:], dWraw[n,:]>;  dg[n] = beta*dg[n] + dot/norm[n]
 *   dW[n,k] = beta*dW[n,k] + scale[n]*dWraw[n,k] - dot*g[n]/norm[n]^3 * W[n,k] */
int dv_wn_bwd(const float* dWraw, int64_t ldr, const float* W, int64_t ldw, const float* g, const float* norm,
              int32_t N, int32_t K, float* dW, int64_t ldd, float* dg, float beta, dv_stream_t stream);

/* ------------------------------------------------------- reparameterisation (K3)
 * out row r = l*n + j (l < reps, j < n) samples from q row qi = src_idx ? src_idx[j] : j:
 *   out[r,d] = mu[qi,d] + eps[r,d] * (mode==LOGVAR ? exp(0.5*sd[qi,d]) : sd[qi,d])
 * (src/blocks.py:170-174, 208-211; eps is explicit so parity tests can inject it).
 * Optional out2[r,d] = out[r,d] - sub[r,d]   (`z2Fz1_sample - z1_sample`, src/DrVAE.py:495).
 * Optional out3[out3_idx[r], d] = out[r,d] for out3_idx[r] >= 0 (a second, scattered copy:
 * the z2Fz1 samples of paired rows also feed the decoder, src/DrVAE.py:455-459).
 * backward: dmu[qi] = beta*dmu[qi] + sum_l dz[l*n+j]; dsd likewise with the chain factor. */
int dv_reparam_fwd(const float* mu, const float* sd, int64_t ldq, const int32_t* src_idx, int32_t n, int32_t reps,
                   int32_t Z, const float* eps, int64_t lde, int32_t mode, float* out, int64_t ldo,
                   const float* sub, int64_t lds, float* out2, int64_t ldo2, float* out3, int64_t ldo3,
                   const int32_t* out3_idx, dv_stream_t stream);
int dv_reparam_bwd(const float* dz, int64_t ldz, const float* eps, int64_t lde, const float* sd, int64_t ldq,
                   const int32_t* src_idx, int32_t n, int32_t reps, int32_t Z, int32_t mode, float* dmu, float* dsd,
                   int64_t lddq, float beta, dv_stream_t stream);
/* CSR form of the backward: q row i (nq rows) sums the sample rows seg_rows[seg_ptr[i]..seg_ptr[i+1])
 * of dz/eps (any number of draws per q row), and optionally the row-aligned (dmu|dsd) contributions
 * `extra[ex_rows[t], :2Z]`, t in [ex_ptr[i], ex_ptr[i+1]).  dmu/dsd[i] = beta*old + sums.
 * `add` (optional, ABI 10): a second gradient source for the first add->n sample rows, g[r] = dz[r] + add->dz[r] for
 * r < add->n (VFAE: the classifier / fprop chain's share of d/dz1, which used to be summed into dz by a launch of its
 * own); `park` (optional): the launch parks on another chain's flag first, like dv_z2f_post_bwd. */
/* KL(q || N(0, I)) of the launch's OWN q rows with per-row free bits (PVAE's prior term, src/PVAE.py): its gradient w.r.t.
 * (mu, logvar) -- coef[i] * gate(raw[i]) * (mu, (e^logvar - 1) / 2) -- is added to the row's result on the way out instead
 * of a dv_kl_rows_bwd launch behind it (round 5).  LOGVAR parameterisation; the log-variances are the launch's own operand. */
typedef struct dv_prior_kl {
    const float* coef;       /* (rows) dLoss / d KL */
    const float* raw;        /* (rows) raw KL of the forward (the free-bits gate) */
    float kl_min;
    const float* mu;         /* (rows, ld) the rows' means; NULL where the launch has them already (dv_z2f_post_bwd: q2) */
    int64_t ld;
} dv_prior_kl;
typedef struct dv_seg_add {
    const float* dz;
    int64_t ld;
    int32_t n;
} dv_seg_add;
int dv_reparam_bwd_seg(const float* dz, int64_t ldz, const float* eps, int64_t lde, const float* sd, int64_t ldq,
                       const int32_t* seg_ptr, const int32_t* seg_rows, int32_t nq, int32_t Z, int32_t mode,
                       const float* extra, int64_t ldx, const int32_t* ex_ptr, const int32_t* ex_rows, float* dmu,
                       float* dsd, int64_t lddq, float beta, const dv_bump* bump, const dv_seg_add* add,
                       const dv_wait* park, const dv_prior_kl* prior, dv_stream_t stream);
/* `prior` (optional, round 5): the prior-KL gradient of q row i is added to (dmu, dsd)[i], see dv_prior_kl */
/* Backward of everything hanging on the z2Fz1 samples (src/DrVAE.py:431-433, 459-487) in one pass
 * over (row i < B, dim d < Z), looping the L samples r = l*B + i; jp = pair_slot[i] (-1: singleton):
 *   g       = dz2f[r] (0 when dz2f == NULL: a model without a classifier) + (jp >= 0 ? dzdec_pert[l*Np + jp] : 0)
 *   dp2[r]  = (g | g*eps*0.5*exp(0.5*lv2)) + [jp >= 0] d KL(q2[jp] || p2[r]) / d p2 * coef[l*Np+jp]*mask
 *   dz1[r] += dp2[r].mu (residual mu2 = z1 + ..., src/blocks.py:357) + (dz1b ? dz1b[r] : 0)
 *   dq2[jp] = sum_l d KL / d q2 * coef*mask                      (mask = free-bits gate on raw)      */
/* ABI 11: the operands travel in a descriptor (28 positional arguments before); shapes in the comments */
typedef struct dv_z2f_desc {
    const float* dz2f;       /* (L*B, Z) or NULL */
    int64_t ld_dz2f;
    const float* dzdec_pert; /* (L*Np, Z) gradient of the decoded z2Fz1 copies, or NULL */
    int64_t ld_pert;
    const int32_t* pair_slot; /* (B) jp or -1 */
    const float* eps;        /* (L*B, Z) */
    int64_t lde;
    const float* p2;         /* (L*B, 2Z) mu2 | lv2 */
    int64_t ldp2;
    const float* q2;         /* (Np, 2Z) mu | lv of q(z2|x2) */
    int64_t ldq2;
    const float* coef;       /* (L*Np) KL coefficients */
    const float* raw;        /* (L*Np) raw KL (free bits) */
    float kl_min;
    const float* dz1b;       /* (L*B, Z) or NULL */
    int64_t ld_dz1b;
    float* dp2;              /* (L*B, 2Z) out */
    int64_t ld_dp2;
    float* dz1;              /* (L*B, Z) in/out (+=) */
    int64_t ld_dz1;
    float* dq2;              /* (Np, 2Z) out, or NULL */
    int64_t ld_dq2;
    int32_t L;
    int32_t B;
    int32_t Np;
    int32_t Z;
    const float* prior_coef; /* (Np) or NULL (with prior_raw): the prior-KL gradient of q2 row jp is added to dq2[jp], */
    const float* prior_raw;  /* see dv_prior_kl (kl_min as above) */
} dv_z2f_desc;
int dv_z2f_post_bwd(const dv_z2f_desc* d, const dv_wait* park, dv_stream_t stream);

/* ------------------------------------------------ diagonal-Gaussian KL per row (K4)
 * row r = l*n + j: q row = qidx ? qidx[j] : j ; p row = pidx ? pidx[r] : r, or the
 * scalar prior (prior_mu, prior_sd) when mu_p == NULL (prior_sd is a log-variance in
 * LOGVAR mode, a std in SIGMA mode).
 *   LOGVAR: raw = -1/2 sum_d [1 - lv_p + lv_q - ((mu_q-mu_p)^2 + e^lv_q)/e^lv_p]   src/blocks.py:180-182
 *   SIGMA : same with log(std^2), std^2                                              src/blocks.py:217-220
 *   out = free_bits ? max(raw, kl_min) : raw      (per-ROW free bits, src/DGMMixin.py:68-75)
 *   out[r] += add[r] (optional);  zout[r,d] = mu_q + eps[r,d]*std_q (optional): the reparameterised
 *   sample of q drawn in the same row pass (q(z3|z1,y): src/DrVAE.py:341-347).
 *   mu2 != NULL: a second term of the same row, KL(N(mu2[r], sd2[r]) || the scalar prior) over Z2 columns, with its
 *   own free bits: out[r] += max(raw2, kl_min), raw2_out[r] = raw2 (the z3 term of an fprop row, src/DrVAE.py:347,358,
 *   evaluated next to the z1 term instead of in a launch of its own).
 * backward: coef[r] = dLoss/d out[r]; gradients are written ROW-ALIGNED (row r of
 * dq_* / dp_*); callers reduce duplicates with dv_rows_segment_sum.  With dz != NULL the
 * backward of that fused sample (dq_mu += dz, dq_sd += dz*eps*dstd/dsd) rides along. */
/* ABI 11: the forward's operands travel in a descriptor (29 positional arguments before: one transposed int64 away
 * from silent corruption); field names as in the comment above */
typedef struct dv_kl_rows_desc {
    const float* mu_q;
    const float* sd_q;
    int64_t ldq;
    const int32_t* qidx;
    const float* mu_p;
    const float* sd_p;
    int64_t ldp;
    const int32_t* pidx;
    float prior_mu;
    float prior_sd;
    int32_t n;
    int32_t reps;
    int32_t Z;
    int32_t mode;
    int32_t free_bits;
    float kl_min;
    float* raw_out;
    float* out;
    const float* add;
    const float* eps;
    int64_t lde;
    float* zout;
    int64_t ldz;
    const float* mu2;
    const float* sd2;
    int64_t ld2;
    int32_t Z2;
    float* raw2_out;
} dv_kl_rows_desc;
int dv_kl_rows_fwd(const dv_kl_rows_desc* d, const dv_wait* park, dv_stream_t stream);
/* two independent sets of plain KL rows (no fused sample, no second term) in ONE launch -- PVAE's prior term and the pairs' term
 * sit back to back on its main chain (round 5) */
int dv_kl_rows_fwd_pair(const dv_kl_rows_desc* d1, const dv_kl_rows_desc* d2, dv_stream_t stream);
/* the backward reads the forward's descriptor -- operands, n / reps / Z / mode, free_bits, kl_min, eps / lde, and raw_out
 * as the raw KL the forward stored (required with free_bits); out / add / zout / mu2.. are not looked at -- plus its own
 * outputs (30 positional arguments before ABI 11) */
typedef struct dv_kl_rows_grad {
    const float* coef;       /* (n*reps) dLoss / d out */
    float* dq_mu;
    float* dq_sd;
    int64_t lddq;
    float* dp_mu;            /* NULL (with dp_sd): no gradient towards p */
    float* dp_sd;
    int64_t lddp;
    float beta;              /* outputs = beta * old + gradient */
    const float* dz;         /* gradient of the fused sample, or NULL */
    int64_t ldz;
} dv_kl_rows_grad;
int dv_kl_rows_bwd(const dv_kl_rows_desc* d, const dv_kl_rows_grad* g, dv_stream_t stream);

/* ------------------------------------- Gaussian log-likelihood over genes per row (K5)
 * out[r] = -1/2 sum_g [log 2pi + log var + (x-mu)^2/var], x row = xidx ? xidx[r] : r
 *   SIGMA : var = sd^2  (src/blocks.py:233-234, the reconstruction term of src/DrVAE.py:442,452,460)
 *   LOGVAR: var = e^sd  (src/blocks.py:195-196)
 * backward: coef[r] = dLoss/d out[r]; writes d/dmu and d/d(sd) -- the latter chained
 * through sd = sd_act(pre) + sd_shift when sd_act != IDENTITY, i.e. directly the
 * gradient w.r.t. the head's pre-activation (src/blocks.py:415).  dx optional. */
int dv_gauss_nll_rows_fwd(const float* x, int64_t ldx, const int32_t* xidx, const float* mu, const float* sd,
                          int64_t ldp, int32_t M, int32_t X, int32_t mode, float* out, const float* bias_mu,
                          const float* bias_sd, float sd_shift, dv_stream_t stream);
/* bias_mu / bias_sd (both or neither, SIGMA mode; ABI 11): mu / sd hold the heads' RAW products and the pass finishes them
 * on its way (mu + bias_mu, softplus(sd + bias_sd) + sd_shift) -- evaluation passes behind a plain heads product */
int dv_gauss_nll_rows_bwd(const float* coef, const float* x, int64_t ldx, const int32_t* xidx, const float* mu,
                          const float* sd, int64_t ldp, int32_t M, int32_t X, int32_t mode, int32_t sd_act,
                          float sd_shift, float* dmu, float* dsd, int64_t ldd, float* dx, int64_t lddx, float beta,
                          dv_stream_t stream);

/* EXTENSION: log-likelihood rows of the Bernoulli / Poisson data decoders that src/DrVAE.py:124-129
 * names (`type_rec='binary'` / `'poisson'`; the reference ships neither class).  v = the head's post-activation output
 * (probability = sigmoid(a), or rate = softplus(a) + shift); x row = xidx ? xidx[r] : r.
 *   out[r] = sum_g log p(x[.,g] | v[r,g]);  coef != NULL: dpre[r,g] = coef[r] * d log p / d a  (a = pre-activation)
 *   BERNOULLI: pc = clamp(v, 1e-10, 1 - 1e-10); log p = x log pc + (1-x) log(1-pc); d/da = x - v inside the clamp, else 0
 *   POISSON  : log p = x log v - v - lgamma(x+1);  d/da = (x/v - 1) (1 - exp(-(v - shift))) */
enum { DV_REC_BERNOULLI = 0, DV_REC_POISSON = 1 };
int dv_rec_nll_rows(int32_t kind, float shift, const float* coef, const float* x, int64_t ldx, const int32_t* xidx,
                    const float* v, int64_t ldv, int32_t M, int32_t X, float* out, float* dpre, int64_t ldd,
                    dv_stream_t stream);

/* The raw-heads forward + backward pass below (dv_gauss_nll_rows_fwdbwd with bias_mu / bias_sd, sigma head = softplus +
 * shift) with the heads' BIAS GRADIENT folded in (wide configuration: saves the separate 1.3-GB column-sum pass over the
 * gradients this kernel writes).  A workgroup owns 1024 genes x 64 rows:
 *   out_part[r, c] = partial log-likelihood of row r over gene chunk c  (M x chunks; out[r] = sum_c out_part[r, c] --
 *                    the loss assembly sums them: dv_loss_term.row_len = chunks)
 *   dmu / dsd      = as dv_gauss_nll_rows_fwdbwd
 *   ws[b, g]          = sum over the rows of row block b of dmu[r, g]        } (row_blocks x ldw; the caller sums the
 *   ws[b, sd_off + g] = the same of dsd[r, g]                                } blocks: dv_colsum(ws, ldw, row_blocks, ..))
 * chunks = dv_nll_raw_cs_chunks(X), row_blocks = dv_nll_raw_cs_row_blocks(M).  X % 4 == 0, 16-B aligned rows.
 * Deterministic (fixed summation order, no atomics).
 * dmu == dsd == ws == NULL (coef unused): FORWARD ONLY -- the row partials of an evaluation pass behind a plain heads
 * product (the whole-set evaluation's loss pass: no activation epilogue in the GEMM, no finished heads written and
 * read back). */
typedef struct dv_nll_raw_cs_desc {
    const float* coef;
    const float* x;
    int64_t ldx;
    const int32_t* xidx;
    const float* mu;
    const float* sd;
    int64_t ldp;
    int32_t M;
    int32_t X;
    float shift;
    float* out_part;
    int32_t chunks;
    float* dmu;
    float* dsd;
    int64_t ldd;
    const float* bias_mu;
    const float* bias_sd;
    float* ws;
    int64_t ldw;
    int64_t sd_off;
    int32_t row_blocks;
} dv_nll_raw_cs_desc;
int dv_gauss_nll_rows_raw_cs(const dv_nll_raw_cs_desc* d, dv_stream_t stream);
int dv_nll_raw_cs_chunks(int32_t X);
int dv_nll_raw_cs_row_blocks(int32_t M);

/* forward + backward in one row pass (the loss is linear in the row terms with coefficients
 * known up front): out[r] as _fwd, dmu/dsd[r,:] = coef[r] * d out[r]/d(mu, pre-activation of sd).
 * bias_mu / bias_sd (both or neither; X floats each): mu / sd then hold the heads' RAW products (x W^T without bias and
 * activation) and this pass finishes them on its way: mu = mu_raw + bias_mu, sd = act(sd_raw + bias_sd) + sd_shift --
 * for heads whose product fills the chip, where the plain GEMM epilogue is the faster one (wide configuration). */
int dv_gauss_nll_rows_fwdbwd(const float* coef, const float* x, int64_t ldx, const int32_t* xidx, const float* mu,
                             const float* sd, int64_t ldp, int32_t M, int32_t X, int32_t mode, int32_t sd_act,
                             float sd_shift, float* out, float* dmu, float* dsd, int64_t ldd, const float* bias_mu,
                             const float* bias_sd, dv_stream_t stream);

/* --------------------------------------------------------- categorical head (K6/K7)
 * probs = clamp(softmax(logits), 1e-10, 1-1e-10)   (src/blocks.py:446-463); with
 * sigmoid1 != 0 logits is (M,1) and probs = clamp(cat(1-s, s)) (src/blocks.py:448-451,461-462). */
int dv_softmax_clamp_fwd(const float* logits, int64_t ldl, int32_t M, int32_t Y, int32_t sigmoid1, float* probs,
                         int64_t ldp, dv_stream_t stream);
int dv_softmax_clamp_bwd(const float* dprobs, int64_t lddp, const float* probs, int64_t ldp, int32_t M, int32_t Y,
                         int32_t sigmoid1, float* dlogits, int64_t ldl, float beta, dv_stream_t stream);
/* per-row terms of a categorical q (any may be NULL):
 *   logp[r]   = log probs[r, labels[r]]                               src/blocks.py:473-474
 *   kl[r,j]   = -p (log prior[r,j] - log p)   (elementwise)           src/blocks.py:479-480
 *   ent[r]    = -sum_j p log p                                        src/blocks.py:476-477
 *   best[r]   = argmax_j p                                            src/blocks.py:485-486 */
int dv_cat_terms_fwd(const float* probs, int64_t ldp, int32_t M, int32_t Y, const int32_t* labels,
                     const float* prior, int64_t ldpr, float* logp, float* kl, int64_t ldk, float* ent,
                     int32_t* best, dv_stream_t stream);
/* dprobs[r,j] = beta*dprobs + c_logp[r]*[j==label]/p + g_kl[r,j]*(log p - log prior + 1) - c_ent[r]*(log p + 1) */
int dv_cat_terms_bwd(const float* probs, int64_t ldp, int32_t M, int32_t Y, const int32_t* labels,
                     const float* prior, int64_t ldpr, const float* c_logp, const float* g_kl, int64_t ldg,
                     const float* c_ent, float* dprobs, int64_t lddp, float beta, dv_stream_t stream);

/* Small-N linear head, N <= 8 outputs -- the classifier q(y|z1,z2) when `dim_h_clf=[]`
 * (src/DrVAE.py:165, src/blocks.py:446): logits = [a1|a2] W^T + b, probs = clamp(softmax).
 * One wavefront per row instead of a >90 %-padded MFMA tile.  Either output may be NULL.
 * Backward takes d(probs) (+ probs: chained through clamp+softmax) or, with probs == NULL,
 * d(logits) directly:
 *   bwd_data  : dst_t[r,c] = beta_t*dst_t[r,c] + sum_j dlogit[r,j] (alpha_t W[j,col0_t+c] + alpha2_t W[j,col1_t+c])
 *               for up to 3 destinations; col1/alpha2 may be NULL (the `[z1, z2F - z1]` input of
 *               src/DrVAE.py:495 sends W1 - W2 to z1 and W2 to z2F)
 *   bwd_weight: dW[j,k] = beta*dW + sum_r dlogit[r,j] [a1|a2][r,k];  db[j] likewise (db may be NULL)
 *               seg_src != NULL: destination 0 starts from the segment sum sum_{u in [seg_ptr[r], seg_ptr[r+1])}
 *               seg_src[u, c] instead of beta_0*dst_0[r,c] (the fprop rows' d/dz1 of src/DrVAE.py:337-358 folded into
 *               the classifier's data gradient: one launch instead of dv_rows_segment_sum + this one)
 * The pointer/size arrays of bwd_data are HOST arrays. */
int dv_smalln_linear_fwd(const float* a1, int64_t lda1, int32_t K1, const float* a2, int64_t lda2, int32_t K2,
                         const float* W, int64_t ldw, const float* bias, int32_t M, int32_t N, float* logits,
                         int64_t ldl, float* probs, int64_t ldp, const dv_ymarg* ymarg, const dv_wait* park,
                         const dv_fprop_kl* fprop_kl, dv_stream_t stream);
int dv_smalln_linear_bwd_data(const float* dprobs, int64_t lddp, const float* probs, int64_t ldp, const float* W,
                              int64_t ldw, int32_t M, int32_t N, int32_t n_dst, float* const* dst,
                              const int64_t* ld, const int32_t* col0, const int32_t* ncol, const float* alpha,
                              const float* beta, const int32_t* col1, const float* alpha2, const float* seg_src, int64_t ld_seg,
                              const int32_t* seg_ptr, dv_stream_t stream);
int dv_smalln_linear_bwd_weight(const float* dprobs, int64_t lddp, const float* probs, int64_t ldp,
                                const float* a1, int64_t lda1, int32_t K1, const float* a2, int64_t lda2,
                                int32_t K2, int32_t M, int32_t N, float* dW, int64_t ldd, float* db, float beta,
                                const dv_publish* pub, float* ws, int32_t ws_splits, dv_stream_t stream);
/* ws / ws_splits (optional, ABI 11): a workspace of ws_splits x N x (K1 + K2 + 1) floats lets the launch split the rows
 * over up to ws_splits workgroups per block of 16 weight columns (M >= 1024 rows: the wide configuration's 4096) and add
 * the partial sums up in a fixed order with a second tiny launch -- deterministic; without it one workgroup per column
 * block walks all M rows */

/* y-marginalisation of src/DrVAE.py:503-534 / src/VFAE.py:331-390 over the stacked
 * "fprop" rows.  For every (l, i) classifier row r (R rows):
 *   labeled   (fp_ptr[r+1]-fp_ptr[r] == 1): yl[r] = log qy[r, label[r]]; kld[r] = klfp[fp row]
 *   unlabeled (== Y fp rows, class j at fp_ptr[r]+j): yl[r] = 0;
 *       kld[r] = sum_j qy[r,j]*klfp[fp_ptr[r]+j] + sum_j qy[r,j](log qy[r,j] - log prior_j)
 * log prior_j = log_prior_v[j] when log_prior_v != NULL (a class prior given as data, `prior_y` of
 * src/DrVAE.py:83-85,389), else the scalar log_prior ('uniform').
 * backward: c_kld[r], c_yl[r] are dLoss/d kld[r], d yl[r]; writes cfp[fp row] = dLoss/d klfp
 * and dqy (R,Y). */
int dv_ymarg_fwd(const float* qy, int64_t ldq, const int32_t* label, const int32_t* fp_ptr, const float* klfp,
                 float log_prior, const float* log_prior_v, int32_t R, int32_t Y, float* yl, float* kld,
                 dv_stream_t stream);
int dv_ymarg_bwd(const float* qy, int64_t ldq, const int32_t* label, const int32_t* fp_ptr, const float* klfp,
                 float log_prior, const float* log_prior_v, const float* c_kld, const float* c_yl, int32_t R,
                 int32_t Y, float* cfp, float* dqy, int64_t lddq, dv_stream_t stream);

/* _fwd and _bwd in one launch (the train step knows the coefficients before the forward pass) */
int dv_ymarg_fwdbwd(const float* qy, int64_t ldq, const int32_t* label, const int32_t* fp_ptr, const float* klfp,
                    float log_prior, const float* log_prior_v, const float* c_kld, const float* c_yl, int32_t R,
                    int32_t Y, float* yl, float* kld, float* cfp, float* dqy, int64_t lddq, dv_stream_t stream);

/* Regression head (`type_y='cont'`, src/DrVAE.py:159-169,503-530): q(y|.) = N(sigmoid(.), fixed var).
 * Per classifier row r=(l,i), i = r % B:
 *   labeled   (has_y[i]): yl[r] = log N(ylab[i,:]; mu[r,:], var);  yv = ylab[i,:]
 *   unlabeled            : yl[r] = 0;                               yv = mu + sd*eps[r,:]   (SGVB sample)
 * yv is written into the y columns of both fprop inputs (fpin_y, z3in_y point at column Z of [z | y]).
 * backward: dlogit = dmu * mu(1-mu) with dmu = c_yl*(y-mu)/var (labeled) or the sum of the gradients of
 * the two y columns (unlabeled); called with dlogit == NULL it only writes cfp[r] = c_kld[r]. */
/* sqerr != 0: the labeled rows are scored by squared error instead, yl[r] = -sum_d (y-mu)^2 (VFAE.py:351). */
int dv_ycont_fwd(const float* mu, int64_t ldm, const float* ylab, const int32_t* has_y, const float* eps, int64_t lde,
                 float logvar, int32_t sqerr, int32_t R, int32_t B, int32_t Y, float* yl, float* fpin_y, int64_t ld1,
                 float* z3in_y, int64_t ld2, dv_stream_t stream);
int dv_ycont_bwd(const float* mu, int64_t ldm, const float* ylab, const int32_t* has_y, float logvar, int32_t sqerr,
                 const float* c_yl, const float* c_kld, const float* dfpin_y, int64_t ld1, const float* dz3in_y,
                 int64_t ld2, int32_t R, int32_t B, int32_t Y, float* dlogit, int64_t ldd, float* cfp,
                 dv_stream_t stream);

/* ------------------------------------------- random-Fourier-feature MMD (K10)
 * `mmd_fourier` of src/blocks.py:40-55.  The projections theta = a * x W + 2 pi b (a = sqrt(2/bandwidth)/sqrt(Z))
 * come from dv_gemm (scale/bias epilogue); these finish the statistic:
 *   _fwd: diff[r] = c*(mean_i cos(th1[i,r]) - mean_j cos(th2[j,r])),  mmd2[0] = sum_r diff[r]^2   (c = sqrt(2/R))
 *   _bwd: G[i,r] = coef * gout[0] * (-diff[r]) * sin(th[i,r]) = d mmd2 / d theta for one input when
 *         coef = +-2c/n (sign - for the second input); d/dx follows as a * G W^T (dv_gemm). */
int dv_mmd_rff_fwd(const float* th1, int64_t ld1, int32_t n1, const float* th2, int64_t ld2, int32_t n2, int32_t R,
                   float c, float* diff, float* mmd2, dv_stream_t stream);
int dv_mmd_rff_bwd(const float* th, int64_t ld, int32_t n, int32_t R, const float* diff, const float* gout, float coef,
                   float* G, int64_t ldg, dv_stream_t stream);

/* ------------------------------------------------------------- row movement
 * out[r,:W] = src[idx?idx[r]:r, :W] (+ sigma*noise[r,:W])  -- the group gathers of
 *   src/DrVAE.py:585-608 and the training-noise augmentation of src/DrVAE.py:404-407
 *   (N(0,1)*add_noise_var) in one pass.
 * onehot_cls != NULL additionally writes out[r, W + c] = (c == onehot_cls[r]) for c < Y:
 *   `torch.cat([z, one_hot(y)], 1)` of src/blocks.py:161 + src/blocks.py:78-92. */
int dv_rows_gather(const float* src, int64_t lds, const int32_t* idx, int32_t n, int32_t W, const float* noise,
                   int64_t ldn, float sigma, const int32_t* onehot_cls, int32_t Y, float* out, int64_t ldo,
                   const dv_wait* park, dv_stream_t stream);
/* Graph-resident minibatch feed (input pipeline of src/run_drvae.py:150-166 + the group gathers of
 * src/DrVAE.py:585-608, kept on the device).  `table` holds the dataset row of every slot of every
 * batch of an epoch (n_batches x B, drawn by the weighted sampler on the device); batch
 * b = clamp(ctr[0] - base[0], 0, n_batches-1) is written into the step's buffers:
 *   xin[r]     = x1[table[b,r]]                (r <  B)      } + sigma*noise[r]
 *   xin[B + j] = x2[table[b,pair_rows[j]]]     (j <  Np)     }
 *   label_r[l*B+i] = has_y[i] ? y[table[b,i]] : 0
 *   fp_cls[m]  = fp_lab[m] ? y[table[b,fp_i[m]]] : fp_slot[m];  onehot[m,c] = (c == fp_cls[m])
 *   ylab[i,:]  = yf[table[b,i],:]   (Yc floats per row; regression targets of type_y='cont', or NULL)
 * ctr/base are DEVICE scalars (ctr = the optimiser's step counter), so the launch arguments are
 * constant and the launch can live inside the captured train-step graph. */
/* `masks` (optional): the per-batch masks of a batch-independent plan (dv_batch_masks below, same table / ctr / base /
 * B / L) written by one more workgroup of this launch instead of a launch of their own. */
/* `park` (optional): the launch first parks like dv_flag_wait -- the first launch of a step waiting for the previous
 * step's other launch chain (DV_ERR_UNSUPPORTED for grids above DV_MAX_PARKED_GRID workgroups). */
typedef struct dv_batch_masks_desc {
    const int32_t* hx;
    const int32_t* hy;
    const int32_t* y;
    int32_t Np;
    float n_tot, kl_rate, pert_rate, yl_rate;
    const float* beta;
    float* c_nll;
    float* c_klz2;
    float* c_yl;
    float* w_recl;
    float* w_pert;
    float* w_yl;
    int32_t* label;
    float* c_klp;
    const int32_t* one_slot;
    const int32_t* gcounts;
} dv_batch_masks_desc;
/* ABI 11: the feed's operands as a descriptor (names as in the comment above; 37 positional arguments before) */
typedef struct dv_batch_feed_desc {
    const float* x1;
    int64_t ld1;
    const float* x2;
    int64_t ld2;
    const int32_t* y;
    const int32_t* table;
    int32_t n_batches;
    const int32_t* ctr;
    const int32_t* base;
    int32_t B;
    const int32_t* pair_rows;
    int32_t Np;
    int32_t X;
    const float* noise;
    int64_t ldn;
    float sigma;
    float* xin;
    int64_t ldo;
    const int32_t* has_y;
    int32_t L;
    int32_t* label_r;
    const int32_t* fp_i;
    const int32_t* fp_lab;
    const int32_t* fp_slot;
    int32_t Mf;
    int32_t* fp_cls;
    float* onehot;
    int64_t ldh;
    int32_t Y;
    const float* yf;
    float* ylab;
    int32_t Yc;
    float* onehot2;
    int64_t ldh2;
} dv_batch_feed_desc;
int dv_batch_feed(const dv_batch_feed_desc* d, const dv_batch_masks_desc* masks, const dv_wait* park,
                  dv_stream_t stream);
/* dst[di,:W] = beta*dst[di,:W] + sum_{t in [seg_ptr[i],seg_ptr[i+1])} w[t]*src[seg_rows[t],:W],
 * di = dst_idx?dst_idx[i]:i; seg_ptr==NULL: segment i is the single row (seg_rows?seg_rows[i]:i).
 * Deterministic (no atomics): the transpose of every gather above. */
/* Per-batch masks of a batch-independent ("universal") step plan -- every row materialised as a pair with all Y
 * class slots; which rows really are pairs / labeled is DATA: the group split and the per-example normalisers of
 * src/DrVAE.py:565-624 (N_total fixed = n_tot, N_pairs and N_labeled counted per batch, max(1, .)) become
 * coefficient / weight vectors written on the device from the batch's flags.  Row i of the batch is dataset row
 * table[b, i] with b = clamp(ctr[0] - base[0]) (graph-resident epoch feed, as dv_batch_feed) or row i itself
 * (table == NULL: hx / hy / y are batch-local).  With r = l*B + i, px / py = the row is a pair / labeled:
 *   c_nll[r] = -1/(L n_tot);  c_nll[LB + r] = px ? -1/(L n_tot) : 0;  c_nll[2LB + r] = px ? -beta pert_rate/(L N_pairs) : 0
 *   w_recl = |c_nll[0 : 2LB]|;  w_pert[r] = px ? 1/(L N_pairs) : 0;  c_klz2[r] = px ? beta kl_rate/(L n_tot) : 0
 *   c_yl[r] = py ? -yl_rate/(L N_labeled) : 0;  w_yl[r] = 1/(L N_labeled);  label[r] = py ? -2 - y : 0
 *   c_klp[i] = 1/n_tot, c_klp[B + i] = px ? 1/n_tot : 0   (optional; B + B rows: KL to the prior of q(z1|x1) | q(z2|x2))
 * (label <= -2 is what dv_ymarg_* read as "labeled row, all class slots materialised").  hx == NULL / hy == NULL:
 * model without pairs / labels.  beta: DEVICE scalar (perturbation annealing coefficient).  One workgroup.
 * Np <= B: only rows [0, Np) of the batch have pair slots (a plan with fewer x2 / z2 rows for feeds that put a
 * batch's pairs first: DeviceBatcher(mode='sampler', pair_bucket=...)); the pair-indexed vectors then have L*Np
 * entries, slot q = l*Np + j, and sit behind the L*B row entries: c_nll[LB + q], c_nll[LB + L*Np + q], w_recl[LB + q],
 * w_pert[q], c_klz2[q], c_klp[B + j].  Np == B is the layout written out above.
 * one_slot (B flags, optional): rows the plan gives ONE class slot because the feed guarantees they are labeled
 * (DeviceBatcher(label_bucket=...)): label[r] = y, the form dv_ymarg_* read for a single-slot row. */
/* gcounts (optional; data parallelism): the GLOBAL (N_pairs, N_labeled) of batch b as two int32 per batch
 * of the table (table == NULL: one pair) -- every rank draws the same global index table from the shared seed and runs its
 * slice of each batch, so the global counts are table data; with them the normalisers above use the global counts instead
 * of this shard's own (n_tot is then the global number of rows): summed shard gradients == the gradient of the
 * concatenated batch. */
int dv_batch_masks(const dv_batch_masks_desc* m, const int32_t* table, int32_t n_batches, const int32_t* ctr,
                   const int32_t* base, int32_t B, int32_t L, dv_stream_t stream);
int dv_rows_segment_sum(const float* src, int64_t lds, const int32_t* seg_ptr, const int32_t* seg_rows,
                        const float* w, int32_t n, int32_t W, const int32_t* dst_idx, float* dst, int64_t ldd,
                        float beta, const dv_wait* park, dv_stream_t stream);
/* out[0] = beta*out[0] + scale * sum_i w[i]*x[idx?idx[i]:i]   (loss scalars; one workgroup) */
int dv_weighted_sum(const float* x, const float* w, const int32_t* idx, int32_t n, float scale, float* out,
                    float beta, dv_stream_t stream);
/* Reconstruction metrics of `eval_x_reconstruction` (src/DGMMixin.py:128-156), the O(M*X) part:
 *   dv_recon_row_stats: out[i, 0..5] = { sum_g (x-r)^2, mean_g x, mean_g r, sum (x-mx)^2, sum (r-mr)^2,
 *                                        sum (x-mx)(r-mr) }   -> RMSE and the per-row Pearson r
 *   dv_col_moments   : out[b,0,g] = sum_i x, out[b,1,g] = sum_i x^2, out[b,2,g] = sum_i (x-r)^2  (fp64) over the rows i
 *                      of row block b = [b*ceil(M/row_blocks), ...): `out` is (row_blocks, 3, X); the caller adds the
 *                      blocks up (a fixed order: no atomics) -- ABI 9: one workgroup per (64 columns, row block)
 *                      instead of 16 workgroups walking all M rows (8192 x 978: 591 -> ~15 us)
 *                      -> variance-weighted R^2 (sklearn r2_score(multioutput='variance_weighted')) */
int dv_recon_row_stats(const float* x, int64_t ldx, const float* r, int64_t ldr, int32_t M, int32_t X, float* out,
                       dv_stream_t stream);
int dv_col_moments(const float* x, int64_t ldx, const float* r, int64_t ldr, int32_t M, int32_t X, double* out,
                   int32_t row_blocks, const int32_t* sel, const float* r_bias, dv_stream_t stream);
/* `sel` (optional, ABI 11): the M rows that count, as indices into x / r (e.g. the rows with a second profile) */
/* `r_bias` (X, optional, round 5): r holds a RAW heads product, r + r_bias is the reconstruction.
 * dv_recon_rows: dv_recon_row_stats AND the Gaussian log-likelihood rows (dv_gauss_nll_rows_fwd, SIGMA mode) in ONE pass
 * over (x, mu, sd) for rows of up to DV_RECON_ROWS_MAX_X columns -- the whole-set evaluation's 978 genes: a wave holds its
 * row in registers.  Same outputs: rows (M, 6), ll (M, optional) -- bitwise those of dv_recon_row_stats on the dword path
 * (odd X or unaligned rows); rows of even width are read with 8-B loads (a lane owns column pairs: another summation order).  bias_mu / bias_sd (both or neither): mu / sd are the heads' RAW products and are finished on the way
 * (mu + bias_mu, softplus(sd + bias_sd) + sd_shift).  X beyond the limit: DV_ERR_UNSUPPORTED (callers take the two
 * separate passes). */
#define DV_RECON_ROWS_MAX_X 1024
typedef struct dv_recon_rows_desc {
    const float* x;
    int64_t ldx;
    const float* mu;
    const float* sd;
    int64_t ldp;
    const float* bias_mu;
    const float* bias_sd;
    float sd_shift;
    int32_t M;
    int32_t X;
    float* rows;
    float* ll;
} dv_recon_rows_desc;
int dv_recon_rows(const dv_recon_rows_desc* d, dv_stream_t stream);

/* Kernel-mixture MMD of `mmd_objective(kernel='poly' | 'rbf')` and the `identity` kernel (src/blocks.py:29-38,59-76; round
 * 5).  The three Gram products x1 x1^T, x2 x2^T, x1 x2^T are dv_gemm calls; on a Gram matrix G (M x N):
 *   dv_mmd_mix_fwd : part[i] = sum_j k_ij, k = 1/nb sum_b f(., gammas[b]) (HOST array, nb <= 8)
 *                    kind 0 poly: f = (gamma G_ij + 1)^2;  kind 1 rbf: f = exp(-gamma d2_ij), d2_ij = sa[i sa_stride] +
 *                    sb[j sb_stride] - 2 G_ij (the squared row norms: the diagonals of the self products)
 *   dv_mmd_mix_bwd : W[i, j] = coef * gout[0] * dk_ij / d(G_ij | d2_ij),  rs[i] = sum_j W[i, j]
 *   dv_mmd_mix_combine: out[0] = m11 - 2 m12 + m22 (MMD^2; m = sum(p) / c), out[1..3] = the means
 *   dv_mmd_identity_fwd: diff[d] = mean_i x1[i, d] - mean_i x2[i, d], out[0] = sum_d diff^2;  _bwd: dx[i, d] = coef gout[0] diff[d] */
int dv_mmd_mix_fwd(const float* G, int64_t ldg, int32_t M, int32_t N, int32_t kind, const float* gammas, int32_t nb,
                   const float* sa, int64_t sa_stride, const float* sb, int64_t sb_stride, float* part, dv_stream_t stream);
int dv_mmd_mix_bwd(const float* G, int64_t ldg, int32_t M, int32_t N, int32_t kind, const float* gammas, int32_t nb,
                   const float* sa, int64_t sa_stride, const float* sb, int64_t sb_stride, const float* gout, float coef,
                   float* W, int64_t ldw, float* rs, dv_stream_t stream);
int dv_mmd_mix_combine(const float* p11, int32_t n11, float c11, const float* p12, int32_t n12, float c12, const float* p22,
                       int32_t n22, float c22, float* out, dv_stream_t stream);
int dv_mmd_identity_fwd(const float* x1, int64_t ld1, int32_t n1, const float* x2, int64_t ld2, int32_t n2, int32_t Z,
                        float* diff, float* out, dv_stream_t stream);
int dv_mmd_identity_bwd(const float* diff, const float* gout, float coef, int32_t n, int32_t Z, float* dx, int64_t ldd,
                        dv_stream_t stream);

/* The tail of a whole-set evaluation (round 5) in three launches instead of ~75 small library ones:
 * dv_recon_finalize: out[0..3] = rmse, variance-weighted R^2, mean per-row Pearson r, mean log-likelihood (float64) from
 *   dv_recon_row_stats' rows (M_all x 6; `sel`: the n rows that count, NULL = rows 0..n-1), dv_col_moments' partials over
 *   the same rows (row_blocks x 3 x X) and the per-row log-likelihoods `ll` (optional: out[3] = nan without) --
 *   `eval_x_reconstruction`, src/DGMMixin.py:128-156.  One workgroup, fixed summation order.
 * dv_rank_metrics: accuracy, ROC-AUC and average precision of `eval_y_prediction` (src/DGMMixin.py:158-190; sklearn's
 *   roc_auc_score / average_precision_score: one threshold per distinct score) WITHOUT a sort: for the n selected rows
 *   (`sel`, NULL = all) and each class c in [c0, c0 + n_cls) the launch counts, per row, how many rows / positives
 *   score above and equal (O(n^2) integer counting spread over the chip; n <= DV_RANK_MAX_ROWS, DV_ERR_UNSUPPORTED
 *   above), then   AUC = sum_{negatives j} (#pos above j + #pos tied with j / 2) / (n_pos n_neg)   (integer sums: exact)
 *                  AP  = 1/n_pos sum_{positives i} #pos(score >= s_i) / #rows(score >= s_i)
 *   positive = binary ? y > 0 : y == c; score = proba[row, c].  out[2c] = AUC (nan: one class only / empty), out[2c + 1]
 *   = AP (0: no positive), out[2 n_cls] = accuracy mean(pred == y) (pred optional).  `counts` (n_cls x n x 4 int32) must
 *   be zero on entry and is zero again on return.  Deterministic. */
#define DV_RANK_MAX_ROWS 32768
int dv_recon_finalize(const float* rows, const int32_t* sel, int32_t n, int32_t X, const double* cols, int32_t row_blocks,
                      const float* ll, double* out, dv_stream_t stream);
int dv_rank_metrics(const float* proba, int64_t ldp, const int32_t* y, const int32_t* pred, const int32_t* sel, int32_t n,
                    int32_t c0, int32_t n_cls, int32_t binary, int32_t* counts, double* out, dv_stream_t stream);

/* All loss scalars of one step in a single launch (src/DrVAE.py:611-624):
 *   loss[0..4] = 0; for each term: loss[out] += scale * sum_i w[i]*x[i]   (w == NULL: plain sum)
 *   loss[5] (ELBO) = <w_elbo[0..2], loss[0..2]>;  loss[6] (CMPL) = <w_cmpl[0..7], loss[0..7]>
 * `terms` is a HOST array (copied into the kernel arguments); x/w are device pointers. */
#define DV_MAX_LOSS_TERMS 8
typedef struct dv_loss_term {
    const float* x;
    const float* w;
    int32_t n;
    float scale;
    int32_t out;     /* 0 RECL, 1 KLD, 2 PERT, 3 YL, 4 MMD */
    int32_t row_len; /* > 1: x is (n / row_len, row_len) and w holds one weight per ROW (w[i / row_len]); else 0 / 1 */
} dv_loss_term;
/* accum (optional, 8 floats): accum[i] += loss[i] in the same launch -- the running sums of a training epoch
 * (src/DrVAE.py:787-795 averages the per-batch objective), read by the host once per epoch */
int dv_loss_assemble(const dv_loss_term* terms, int32_t n_terms, const float* w_elbo, const float* w_cmpl,
                     float* loss, const int32_t* halt, int32_t n_halt, float* accum, dv_stream_t stream);
/* same, but first parks on `wait` (a dv_wait, like dv_flag_wait) inside the launch (the terms of another chain are
 * read only after the wait; saves the separate wait launch), and last advances the counters of `bump` (a dv_bump, like
 * dv_counters_add2; a counter may alias wait->ctr: it is read before; saves the counter launch in front of the
 * optimiser).  With n_terms == 0 `loss` is left untouched: the launch only parks and advances the counters (another
 * chain assembles the scalars).  wait == NULL / wait->flag == NULL: no wait: the launch assembles the scalars and
 * advances the counters; bump == NULL: no counters.  (ABI 11: the wait and the counters were ten positional scalars) */
int dv_loss_assemble_after(const dv_wait* wait, const dv_loss_term* terms, int32_t n_terms, const float* w_elbo,
                           const float* w_cmpl, float* loss, const dv_bump* bump, const int32_t* halt, int32_t n_halt,
                           float* accum, dv_stream_t stream);
/* nn.BatchNorm1d(N, affine=True) over the rows of x (M, N) -- the `batch_norm=True` option of blocks.MLP
 * (src/blocks.py:137-149).  training != 0: batch statistics (biased variance for the normalisation; running_mean /
 * running_var, when given, move by `momentum`, running_var with the unbiased variance, as torch); training == 0: the
 * running statistics.  mean_out / rstd_out (N each) keep what the backward pass needs. */
int dv_bn_fwd(const float* x, int64_t ldx, int32_t M, int32_t N, const float* w, const float* b, float eps,
              float* mean_out, float* rstd_out, float* y, int64_t ldy, float* running_mean, float* running_var,
              float momentum, int32_t training, dv_stream_t stream);
/* dw = sum_i dy*xhat, db = sum_i dy, dx = w*rstd*(dy - db/M - xhat*dw/M) (training) | dy*w*rstd (eval); dx, dw, db
 * optional */
int dv_bn_bwd(const float* dy, int64_t ldd, const float* x, int64_t ldx, const float* mean, const float* rstd,
              const float* w, int32_t M, int32_t N, float* dx, int64_t lddx, float* dw, float* db, int32_t training,
              dv_stream_t stream);
/* y = x * mask * scale: nn.Dropout with its keep mask (forward, and backward on dy) -- hidden-layer dropout of
 * blocks.MLP (src/blocks.py:140-141) */
int dv_mask_scale(const float* x, int64_t ldx, const float* mask, int64_t ldm, float scale, int32_t M, int32_t N,
                  float* y, int64_t ldy, dv_stream_t stream);
/* y[i] = a*x[i] + b*y[i] over n contiguous floats */
int dv_axpby(const float* x, float a, float* y, float b, int64_t n, dv_stream_t stream);

/* ------------------------------------------------------------------ optimiser (K11)
 * torch.optim.Adam as configured by src/DGMMixin.py:36 (coupled L2: g += wd*p), in the
 * arithmetic order of torch 2.x `_single_tensor_adam`, on a flat fp32 arena.
 * step_dev[0] (int32, device) is the 1-based step used for the bias corrections; it is
 * read, not modified (bump it with dv_counter_add so graph replays advance). gscale
 * multiplies g first (1/world_size style scaling; 1.0 for summed gradients). */
typedef struct dv_adam_hyper {      /* ABI 11: six same-typed scalars travel by name, not by position */
    float lr;
    float beta1;
    float beta2;
    float eps;
    float weight_decay;
    float gscale;
} dv_adam_hyper;
int dv_adam_l2(float* p, const float* g, float* m, float* v, int64_t n, const dv_adam_hyper* h,
               const int32_t* step_dev, const int32_t* halt, int32_t n_halt, dv_stream_t stream);
/* same sweep, but the elements [lo, hi) are touched only after another launch chain has published `gate->flag`
 * (flag[0] >= ctr[0] + add, see dv_flag_publish): only the workgroups overlapping the range park (bounded
 * like dv_flag_wait: err[0] = 1 on time-out, err[1] += ticks parked), so gradients that are leaves of
 * the backward pass may still be in flight on the other chain when the optimiser launch starts */
int dv_adam_l2_gated(float* p, const float* g, float* m, float* v, int64_t n, const dv_adam_hyper* h,
                     const int32_t* step_dev, const dv_wait* gate, int64_t lo, int64_t hi, const int32_t* halt,
                     int32_t n_halt, dv_stream_t stream);
/* torch.optim.Adamax with coupled L2 (the `optim_alg='adamax'` branch of src/DGMMixin.py:37-38):
 * u is the exponentially weighted infinity norm; same conventions as dv_adam_l2. */
int dv_adamax_l2(float* p, const float* g, float* m, float* u, int64_t n, const dv_adam_hyper* h,
                 const int32_t* step_dev, const int32_t* halt, int32_t n_halt, dv_stream_t stream);
int dv_counter_add(int32_t* counter_lo_hi, int32_t n_words, int64_t inc, dv_stream_t stream);
/* Joins folded into their consumers: dv_z2f_post_bwd, dv_reparam_bwd_seg and dv_rows_segment_sum take an optional `park`
 * (every workgroup of the launch first parks like dv_flag_wait: the first consumer of another chain's
 * results waits for them itself, no separate wait launch; keep such grids well below the chip's resident
 * capacity), dv_reparam_bwd_seg an optional `bump` (the launch also advances up to two device counters
 * like dv_counters_add2).  `halt` arguments (here and above): n_halt error words at halt[0], halt[2],
 * halt[4] ... (the err[0] of the dv_wait sites of a step, laid out as (err, ticks) pairs); if any is
 * non-zero the loss scalars come out NaN and the optimiser sweep leaves p / m / v untouched. */
/* two device counters in one launch (the optimiser step count and the Philox counter of a train step) */
/* `pub` (optional): published on entry, BEFORE the counters move (its ctr may be one of them) */
int dv_counters_add2(int32_t* c1, int32_t n1, int64_t inc1, int32_t* c2, int32_t n2, int64_t inc2,
                     const dv_publish* pub, dv_stream_t stream);

/* Device-side fork/join between two launch chains that run concurrently (two root branches of one
 * hipGraph; no reference counterpart -- replaces graph edges, which cost ~27 us per fork+join on the
 * ROCm executor): dv_flag_publish stores flag[0] = ctr[0] + add (release) after the kernels before it
 * in its stream; dv_flag_wait parks ONE thread until flag[0] >= ctr[0] + add (acquire), so the kernels
 * after it in ITS stream see the producer's results.  It never hangs: after max_spins polls (~0.1 us
 * each) it sets err[0] = 1 and returns.  err[1] accumulates the time spent parked (wall_clock64 ticks). */
int dv_flag_publish(int32_t* flag, const int32_t* ctr, int32_t add, dv_stream_t stream);
/* (pub, optional: published on entry of the wait launch, before it starts to poll) */
int dv_flag_wait(int32_t* flag, const int32_t* ctr, int32_t add, int32_t* err, int32_t max_spins, const dv_publish* pub,
                 dv_stream_t stream);

/* out[i] ~ N(0,1), Philox4x32-10 keyed by `seed`, counter = ctr_dev[0..1] (uint64 as two
 * int32 words, device) + i/4, Box-Muller on the four outputs.  (the `normal_()` draws of
 * src/blocks.py:172,210 and src/DrVAE.py:405,415 moved on device). */
int dv_fill_normal(float* out, int64_t n, uint64_t seed, const int32_t* ctr_dev, dv_stream_t stream);
/* Row-keyed form for the train step's noise arena (SURVEY.md 8(e) "RNG under DP": results must not depend
 * on the number of ranks).  desc = n_rows x {offset into arena (floats), width, draw id, GLOBAL minibatch
 * row} (int32 x 4, 16-B aligned); element (row, col) = Philox(key = seed (^ ctr_dev[1] in the high word),
 * counter = (col/4, global row, draw id, ctr_dev[0]))[col % 4] through Box-Muller.  ctr_dev counts draw
 * EVENTS (one per train step: advance it by 1), so every rank holds the same value; a rank that owns rows
 * [rB, (r+1)B) of the global minibatch draws exactly the values a single process would draw for them. */
int dv_fill_normal_rows(float* arena, const int32_t* desc, int32_t n_rows, uint64_t seed, const int32_t* ctr_dev,
                        const dv_wait* park, dv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DRVAE_HIP_H */
