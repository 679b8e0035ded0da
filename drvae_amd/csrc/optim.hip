// Fused Adam (coupled L2) over a flat fp32 arena, device-side step/RNG counters and the
// on-device N(0,1) generator (Philox4x32-10 + Box-Muller) of the Dr.VAE train step.
// All HBM-streaming: 16-B per lane, grid-stride, 7 words of traffic per parameter.
#include "dv_common.h"
#include <cstring>

namespace {

struct AdamC {
    float lr_over_bc1, inv_unused, bc2_sqrt, eps, wd, one_minus_b1, b2, one_minus_b2, gscale;
};

// bias corrections from the device-side step counter, in double like torch's python floats
__device__ __forceinline__ void adam_consts(float lr, float b1, float b2, const int32_t* step_dev, float* step_size,
                                            float* bc2_sqrt) {
    const double t = (double)step_dev[0];
    const double bc1 = 1.0 - pow((double)b1, t);
    const double bc2 = 1.0 - pow((double)b2, t);
    *step_size = (float)((double)lr / bc1);
    *bc2_sqrt = (float)sqrt(bc2);
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float step_size, float bc2_sqrt,
                                         float eps, float wd, float w1, float b2, float w2, float gscale) {
    // torch/optim/adam.py::_single_tensor_adam (2.x): grad += wd*p; m.lerp_(g, 1-b1);
    // v = v*b2 + (1-b2)*g*g; denom = sqrt(v)/sqrt(bc2) + eps; p += -(lr/bc1) * (m/denom)
    g *= gscale;
    if (wd != 0.f) g = g + wd * p;
    m = m + w1 * (g - m);
    v = v * b2 + (w2 * g) * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p + (-step_size) * (m / denom);
}

// sticky error words of the step's device-side waits: (err, ticks) pairs, see dv_wait in drvae_hip.h
__device__ __forceinline__ int any_halt(const int32_t* halt, int n_halt) {
    int bad = 0;
    for (int i = 0; i < n_halt; ++i)
        bad |= __hip_atomic_load(halt + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    return bad;
}

// optional gate of the optimiser sweep: the elements [lo, hi) of the arena (gradients another launch chain
// still writes) are touched only after that chain has published `flag` (see dv_flag_publish)
struct AdamGate {
    int32_t* flag;
    const int32_t* ctr;
    int add;
    int32_t* err;
    int max_spins;
    int64_t lo, hi;
};

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float wd, float gscale,
                                                   const int32_t* __restrict__ step_dev, int vec4, AdamGate gate,
                                                   const int32_t* __restrict__ halt, int n_halt) {
    __shared__ float sc[2];
    __shared__ int halted;
    // only the workgroups whose elements overlap the gated range park (typically one): a parked
    // workgroup or two can never starve the chain that is to publish
    bool need = false;
    if (gate.flag != nullptr) {
        const int64_t per = vec4 ? 4 : 1, span = (int64_t)blockDim.x * per;
        const int64_t bstride = (int64_t)gridDim.x * span;
        need = gate.hi > (vec4 ? ((n >> 2) << 2) : n) && gate.lo < n;       // the scalar tail (all workgroups sweep it)
        for (int64_t e0 = blockIdx.x * span; e0 < n; e0 += bstride) need = need || (e0 < gate.hi && e0 + span > gate.lo);
    }
    if (threadIdx.x == 0) {
        // (the gate's counter and flag and the error words are loaded FIRST: their round trips run under the two
        // double-precision pow() of the bias corrections instead of behind them)
        int want = 0, seen = 0;
        if (need) {
            want = gate.ctr[0] + gate.add;
            seen = __hip_atomic_load(gate.flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        }
        int h = any_halt(halt, n_halt);
        adam_consts(lr, b1, b2, step_dev, &sc[0], &sc[1]);
        if (need && seen - want < 0) {
            const long long t0 = wall_clock64();
            int k = 0;
            while (__hip_atomic_load(gate.flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want < 0) {
                __builtin_amdgcn_s_sleep(4);
                if (++k > gate.max_spins) {
                    atomicExch(gate.err, 1);
                    break;
                }
            }
            atomicAdd(gate.err + 1, (int32_t)(wall_clock64() - t0));
            // one decision per workgroup, taken by the thread that waited (a per-thread re-read of the error words
            // could split a workgroup between updated and untouched elements)
            h = h || any_halt(halt, n_halt);
        }
        halted = h;
    }
    if (need) {
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every wave acquires (see park_block)
    }
    __syncthreads();
    // a chain wait has timed out (now or in an earlier step): the gradients may be built on stale data --
    // leave parameters and moments as they are (the loss scalars come out NaN, the host raises).  In the one step in
    // which THIS launch's gate times out only the parked workgroups (the gated slice) see it: the others have swept
    // their elements already -- that step is applied everywhere but on the gated slice, every later one nowhere
    if (halted) return;
    const float step_size = sc[0], bc2_sqrt = sc[1];
    // (float)(1 - beta) computed in double first, as python does before the op sees it
    const float w1 = (float)(1.0 - (double)b1), w2 = (float)(1.0 - (double)b2);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (vec4) {
        const int64_t n4 = n >> 2;
        float4* p4 = reinterpret_cast<float4*>(p);
        const float4* g4 = reinterpret_cast<const float4*>(g);
        float4* m4 = reinterpret_cast<float4*>(m);
        float4* v4 = reinterpret_cast<float4*>(v);
        for (int64_t i = t0; i < n4; i += stride) {
            float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
            adam_one(pp.x, gg.x, mm.x, vv.x, step_size, bc2_sqrt, eps, wd, w1, b2, w2, gscale);
            adam_one(pp.y, gg.y, mm.y, vv.y, step_size, bc2_sqrt, eps, wd, w1, b2, w2, gscale);
            adam_one(pp.z, gg.z, mm.z, vv.z, step_size, bc2_sqrt, eps, wd, w1, b2, w2, gscale);
            adam_one(pp.w, gg.w, mm.w, vv.w, step_size, bc2_sqrt, eps, wd, w1, b2, w2, gscale);
            p4[i] = pp;
            m4[i] = mm;
            v4[i] = vv;
        }
        for (int64_t i = (n4 << 2) + t0; i < n; i += stride)
            adam_one(p[i], g[i], m[i], v[i], step_size, bc2_sqrt, eps, wd, w1, b2, w2, gscale);
    } else {
        for (int64_t i = t0; i < n; i += stride)
            adam_one(p[i], g[i], m[i], v[i], step_size, bc2_sqrt, eps, wd, w1, b2, w2, gscale);
    }
}

// The same sweep for arenas far larger than the Infinity Cache (wide configuration: 125 M parameters = 3.5 GB per step):
// two 16-B groups per thread in flight, non-temporal loads and stores (nothing of the sweep is read again before the next
// step has streamed gigabytes through the caches), 16384 workgroups.  tools/adam_probe.hip alone: 0.629 -> 0.564 ms = 5.57 -> 6.20
// TB/s (the guide's float4-copy figure is 6.29); inside the step (behind the dW products) 0.68 -> 0.63 ms.  Ungated (the chip-filling steps have no side chain to gate on).
typedef float adam_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void adam_stream_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                          float b1, float b2, float eps, float wd, float gscale,
                                                          const int32_t* __restrict__ step_dev,
                                                          const int32_t* __restrict__ halt, int n_halt) {
    __shared__ float sc[2];
    __shared__ int halted;
    if (threadIdx.x == 0) {
        adam_consts(lr, b1, b2, step_dev, &sc[0], &sc[1]);
        halted = any_halt(halt, n_halt);
    }
    __syncthreads();
    if (halted) return;
    const float step_size = sc[0], bc2_sqrt = sc[1];
    const float w1 = (float)(1.0 - (double)b1), w2 = (float)(1.0 - (double)b2);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, n4 = n >> 2;
    const int64_t t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    adam_f4* p4 = reinterpret_cast<adam_f4*>(p);
    const adam_f4* g4 = reinterpret_cast<const adam_f4*>(g);
    adam_f4* m4 = reinterpret_cast<adam_f4*>(m);
    adam_f4* v4 = reinterpret_cast<adam_f4*>(v);
    for (int64_t i = t0; i < n4; i += 2 * stride) {
        adam_f4 pp[2], gg[2], mm[2], vv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t j = i + u * stride < n4 ? i + u * stride : i;
            pp[u] = __builtin_nontemporal_load(p4 + j);
            gg[u] = __builtin_nontemporal_load(g4 + j);
            mm[u] = __builtin_nontemporal_load(m4 + j);
            vv[u] = __builtin_nontemporal_load(v4 + j);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = pp[u][e], b = mm[u][e], c = vv[u][e];
                adam_one(a, gg[u][e], b, c, step_size, bc2_sqrt, eps, wd, w1, b2, w2, gscale);
                pp[u][e] = a; mm[u][e] = b; vv[u][e] = c;
            }
            const int64_t j = i + u * stride;
            if (j < n4) {
                __builtin_nontemporal_store(pp[u], p4 + j);
                __builtin_nontemporal_store(mm[u], m4 + j);
                __builtin_nontemporal_store(vv[u], v4 + j);
            }
        }
    }
    for (int64_t i = (n4 << 2) + t0; i < n; i += stride)
        adam_one(p[i], g[i], m[i], v[i], step_size, bc2_sqrt, eps, wd, w1, b2, w2, gscale);
}

__global__ void counter_add_kernel(int32_t* c, int n_words, int64_t inc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (n_words == 1) {
        c[0] = (int32_t)(c[0] + inc);
    } else {
        uint64_t v = ((uint64_t)(uint32_t)c[1] << 32) | (uint32_t)c[0];
        v += (uint64_t)inc;
        c[0] = (int32_t)(uint32_t)v;
        c[1] = (int32_t)(uint32_t)(v >> 32);
    }
}

__global__ void counters_add2_kernel(int32_t* c1, int n1, int64_t inc1, int32_t* c2, int n2, int64_t inc2, dv_publish pub) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    publish_block0(pub);      // (before the counters move: pub.ctr may be one of them)
    int32_t* cs[2] = {c1, c2};
    const int ns[2] = {n1, n2};
    const int64_t incs[2] = {inc1, inc2};
    for (int t = 0; t < 2; ++t) {
        int32_t* c = cs[t];
        if (ns[t] == 1) {
            c[0] = (int32_t)(c[0] + incs[t]);
        } else {
            uint64_t v = ((uint64_t)(uint32_t)c[1] << 32) | (uint32_t)c[0];
            v += (uint64_t)incs[t];
            c[0] = (int32_t)(uint32_t)v;
            c[1] = (int32_t)(uint32_t)(v >> 32);
        }
    }
}

// ------------------------------------------------------------------ Philox4x32-10
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                             uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)x + 0.5f) * 2.3283064365386963e-10f; }

__device__ __forceinline__ void philox_normal4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1, float (&z)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const float r0 = sqrtf(-2.f * logf(u01(c0))), r1 = sqrtf(-2.f * logf(u01(c2)));
    float s0, co0, s1, co1;
    sincosf(6.283185307179586f * u01(c1), &s0, &co0);
    sincosf(6.283185307179586f * u01(c3), &s1, &co1);
    z[0] = r0 * co0;
    z[1] = r0 * s0;
    z[2] = r1 * co1;
    z[3] = r1 * s1;
}

// Row-keyed draws (SURVEY.md 8(e) "RNG under DP"): element (row, col) of the noise arena is
// Philox(key = seed ^ step_hi, counter = (col/4, global row, draw id, step_lo))[col%4], so a value depends only
// on WHICH draw of WHICH global minibatch row of WHICH step it is -- not on how rows are grouped into
// buffers or sharded over ranks.  desc[r] = {offset into arena (floats), width, draw id, global row};
// one wave per row.
__global__ __launch_bounds__(256) void fill_normal_rows_kernel(float* __restrict__ arena,
                                                               const int4* __restrict__ desc, int n_rows,
                                                               uint64_t seed, const int32_t* ctr_dev, dv_wait park) {
    park_block(park);         // (the counter below is read behind the wait: its advance is part of what is waited for)
    uint32_t step_lo = 0, step_hi = 0;
    if (ctr_dev) {
        step_lo = (uint32_t)ctr_dev[0];
        step_hi = (uint32_t)ctr_dev[1];
    }
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ step_hi;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwave = gridDim.x * (blockDim.x >> 6);
    for (int r = wave; r < n_rows; r += nwave) {
        const int4 d = desc[r];
        float* out = arena + d.x;
        const int w = d.y, w4 = (w + 3) >> 2;
        const bool al = ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
        for (int c = lane; c < w4; c += 64) {
            float z[4];
            philox_normal4((uint32_t)c, (uint32_t)d.w, (uint32_t)d.z, step_lo, k0, k1, z);
            const int o = c << 2;
            if (al && o + 4 <= w) {
                *reinterpret_cast<float4*>(out + o) = make_float4(z[0], z[1], z[2], z[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (o + e < w) out[o + e] = z[e];
            }
        }
    }
}

__global__ __launch_bounds__(256) void fill_normal_kernel(float* __restrict__ out, int64_t n, uint64_t seed,
                                                          const int32_t* __restrict__ ctr_dev) {
    uint64_t base = 0;
    if (ctr_dev) base = ((uint64_t)(uint32_t)ctr_dev[1] << 32) | (uint32_t)ctr_dev[0];
    const int64_t n4 = (n + 3) >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t ctr = base + (uint64_t)i;
        uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c0, c1, c2, c3, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const float r0 = sqrtf(-2.f * logf(u01(c0))), r1 = sqrtf(-2.f * logf(u01(c2)));
        float s0, co0, s1, co1;
        sincosf(6.283185307179586f * u01(c1), &s0, &co0);
        sincosf(6.283185307179586f * u01(c3), &s1, &co1);
        const float z[4] = {r0 * co0, r0 * s0, r1 * co1, r1 * s1};
        const int64_t o = i << 2;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (o + e < n) out[o + e] = z[e];
    }
}

}  // namespace

#define ST(s) static_cast<hipStream_t>(s)

extern "C" int dv_abi_version(void) { return DV_ABI_VERSION; }

// torch.optim.Adamax (2.x `_single_tensor_adamax`), the other branch of src/DGMMixin.py:37-38:
// grad += wd*p; m.lerp_(g, 1-b1); u = max(u*b2, |g|+eps); p += -(lr/(1-b1^t)) * (m/u)
__global__ __launch_bounds__(256) void adamax_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ u, int64_t n, float lr,
                                                     float b1, float b2, float eps, float wd, float gscale,
                                                     const int32_t* __restrict__ step_dev,
                                                     const int32_t* __restrict__ halt, int n_halt) {
    __shared__ float sc[2];
    __shared__ int halted;
    if (threadIdx.x == 0) {
        adam_consts(lr, b1, b2, step_dev, &sc[0], &sc[1]);
        halted = any_halt(halt, n_halt);
    }
    __syncthreads();
    if (halted) return;
    const float clr = sc[0];
    const float w1 = (float)(1.0 - (double)b1);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
        float gg = g[i] * gscale;
        const float pp = p[i];
        if (wd != 0.f) gg = gg + wd * pp;
        const float mm = m[i] + w1 * (gg - m[i]);
        const float uu = fmaxf(u[i] * b2, fabsf(gg) + eps);
        m[i] = mm;
        u[i] = uu;
        p[i] = pp + (-clr) * (mm / uu);
    }
}

/* sha256 over the library's sources (csrc/ *.hip *.inc *.h + include/drvae_hip.h, see drvae_amd/build.py::source_hash),
 * baked in at build time: lets a test on the GPU box assert that the shipped binary was built from the tree next to it */
#ifndef DV_SOURCE_HASH
#define DV_SOURCE_HASH "unhashed"
#endif
static const char dv_src_tag[] = DV_SOURCE_HASH;      /* "dv-src-sha256:<hex>": the tag lets build.py find it in the file */
extern "C" const char* dv_source_hash(void) {
    const char* c = strchr(dv_src_tag, ':');
    return c ? c + 1 : dv_src_tag;
}

extern "C" const char* dv_error_string(int code) {
    switch (code) {
        case DV_OK: return "ok";
        case DV_ERR_ARG: return "invalid argument";
        case DV_ERR_LAUNCH: return "kernel launch failed";
        case DV_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}

static int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                       float eps, float weight_decay, float gscale, const int32_t* step_dev, const AdamGate& gate,
                       const int32_t* halt, int32_t n_halt, dv_stream_t stream) {
    DV_REQUIRE(n >= 0 && n_halt >= 0 && (halt || n_halt == 0));
    if (n == 0) return DV_OK;
    DV_REQUIRE(p && g && m && v && step_dev);
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const int vec4 = al(p) && al(g) && al(m) && al(v);
    int64_t blocks = ((vec4 ? (n >> 2) : n) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (vec4 && gate.flag == nullptr && n >= (int64_t(16) << 20)) {       // far beyond the Infinity Cache: the streaming sweep
        hipLaunchKernelGGL(adam_stream_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, ST(stream), p, g, m, v,
                           n, lr, beta1, beta2, eps, weight_decay, gscale, step_dev, halt, n_halt);
        DV_RETURN_LAUNCH();
    }
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), p, g, m, v, n, lr, beta1,
                       beta2, eps, weight_decay, gscale, step_dev, vec4, gate, halt, n_halt);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_adam_l2(float* p, const float* g, float* m, float* v, int64_t n, const dv_adam_hyper* h,
                          const int32_t* step_dev, const int32_t* halt, int32_t n_halt, dv_stream_t stream) {
    DV_REQUIRE(h != nullptr);
    return adam_launch(p, g, m, v, n, h->lr, h->beta1, h->beta2, h->eps, h->weight_decay, h->gscale, step_dev, AdamGate{},
                       halt, n_halt, stream);
}

extern "C" int dv_adam_l2_gated(float* p, const float* g, float* m, float* v, int64_t n, const dv_adam_hyper* h,
                                const int32_t* step_dev, const dv_wait* gate, int64_t lo, int64_t hi,
                                const int32_t* halt, int32_t n_halt, dv_stream_t stream) {
    DV_REQUIRE(h != nullptr && gate != nullptr);
    DV_REQUIRE(gate->flag && gate->ctr && gate->err && gate->max_spins > 0 && lo >= 0 && hi >= lo && hi <= n);
    return adam_launch(p, g, m, v, n, h->lr, h->beta1, h->beta2, h->eps, h->weight_decay, h->gscale, step_dev,
                       AdamGate{gate->flag, gate->ctr, gate->add, gate->err, gate->max_spins, lo, hi}, halt, n_halt, stream);
}

extern "C" int dv_adamax_l2(float* p, const float* g, float* m, float* u, int64_t n, const dv_adam_hyper* h,
                            const int32_t* step_dev, const int32_t* halt, int32_t n_halt, dv_stream_t stream) {
    DV_REQUIRE(h != nullptr && n >= 0 && n_halt >= 0 && (halt || n_halt == 0));
    if (n == 0) return DV_OK;
    DV_REQUIRE(p && g && m && u && step_dev);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adamax_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), p, g, m, u, n, h->lr, h->beta1,
                       h->beta2, h->eps, h->weight_decay, h->gscale, step_dev, halt, n_halt);
    DV_RETURN_LAUNCH();
}

// Device-side ordering between two concurrently running launch chains (two root branches of one
// hipGraph): the producer chain publishes flag = ctr + add after the kernel whose results are
// needed, the consumer chain parks a one-thread kernel on the flag.  Data visibility is provided by
// the ordinary kernel boundaries on either side; only the flag itself is accessed atomically.
__global__ void flag_publish_kernel(int32_t* flag, const int32_t* ctr, int add) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, ctr[0] + add, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void flag_wait_kernel(int32_t* flag, const int32_t* ctr, int add, int32_t* err, int max_spins, dv_publish pub) {
    if (threadIdx.x != 0) return;
    // (pub: published on entry, like dv_flag_publish -- everything before this launch in its stream is complete)
    if (pub.flag != nullptr) __hip_atomic_store(pub.flag, pub.ctr[0] + pub.add, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const int want = ctr[0] + add;
    const long long t0 = wall_clock64();
    int n = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want < 0) {
        __builtin_amdgcn_s_sleep(4);
        if (++n > max_spins) {       // never hang the device: report and let the chain run on
            atomicExch(err, 1);
            break;
        }
    }
    atomicAdd(err + 1, (int32_t)(wall_clock64() - t0));   // time parked, in constant-clock ticks (tuning statistic; no-return atomic)
}

extern "C" int dv_flag_publish(int32_t* flag, const int32_t* ctr, int32_t add, dv_stream_t stream) {
    DV_REQUIRE(flag && ctr);
    hipLaunchKernelGGL(flag_publish_kernel, dim3(1), dim3(64), 0, ST(stream), flag, ctr, add);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_flag_wait(int32_t* flag, const int32_t* ctr, int32_t add, int32_t* err, int32_t max_spins,
                            const dv_publish* pub_in, dv_stream_t stream) {
    DV_REQUIRE(flag && ctr && err && max_spins > 0);
    const dv_publish pub = pub_in ? *pub_in : dv_publish{nullptr, nullptr, 0};
    DV_REQUIRE(pub.flag == nullptr || pub.ctr != nullptr);
    hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(64), 0, ST(stream), flag, ctr, add, err, max_spins, pub);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_counter_add(int32_t* counter_lo_hi, int32_t n_words, int64_t inc, dv_stream_t stream) {
    DV_REQUIRE(counter_lo_hi && (n_words == 1 || n_words == 2));
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, ST(stream), counter_lo_hi, n_words, inc);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_counters_add2(int32_t* c1, int32_t n1, int64_t inc1, int32_t* c2, int32_t n2, int64_t inc2,
                                const dv_publish* pub_in, dv_stream_t stream) {
    DV_REQUIRE(c1 && c2 && (n1 == 1 || n1 == 2) && (n2 == 1 || n2 == 2));
    dv_publish pub = pub_in ? *pub_in : dv_publish{nullptr, nullptr, 0};
    DV_REQUIRE(pub.flag == nullptr || pub.ctr != nullptr);
    hipLaunchKernelGGL(counters_add2_kernel, dim3(1), dim3(64), 0, ST(stream), c1, n1, inc1, c2, n2, inc2, pub);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_fill_normal_rows(float* arena, const int32_t* desc, int32_t n_rows, uint64_t seed,
                                   const int32_t* ctr_dev, const dv_wait* park_in, dv_stream_t stream) {
    DV_REQUIRE(n_rows >= 0);
    dv_wait park = park_in ? *park_in : dv_wait{nullptr, nullptr, 0, 0, nullptr};
    DV_REQUIRE(park.flag == nullptr || (park.ctr && park.err && park.max_spins > 0));
    if (n_rows == 0) return park.flag ? DV_ERR_UNSUPPORTED : DV_OK;
    DV_REQUIRE(arena && desc && (reinterpret_cast<uintptr_t>(desc) & 15) == 0);
    int blocks = (n_rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (park.flag != nullptr && blocks > 512) blocks = 512;      // a parked grid stays well below the chip's resident capacity
    hipLaunchKernelGGL(fill_normal_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), arena,
                       reinterpret_cast<const int4*>(desc), n_rows, seed, ctr_dev, park);
    DV_RETURN_LAUNCH();
}

extern "C" int dv_fill_normal(float* out, int64_t n, uint64_t seed, const int32_t* ctr_dev, dv_stream_t stream) {
    DV_REQUIRE(n >= 0);
    if (n == 0) return DV_OK;
    DV_REQUIRE(out);
    int64_t blocks = (((n + 3) >> 2) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_normal_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), out, n, seed, ctr_dev);
    DV_RETURN_LAUNCH();
}
