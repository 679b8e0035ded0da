// Shared device helpers for the Dr.VAE ELBO hot path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/drvae_hip.h"

#define DV_WAVE 64

// Launch-check used by every C-ABI entry point: never throws, never syncs.
#define DV_RETURN_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        return e__ == hipSuccess ? DV_OK : DV_ERR_LAUNCH;    \
    } while (0)

#define DV_REQUIRE(cond)              \
    do {                              \
        if (!(cond)) return DV_ERR_ARG; \
    } while (0)

// ---------------------------------------------------------------- activations
// Forward value and derivative-from-OUTPUT (so backward never needs the pre-activation):
//   elu(a=1):  y = x>0 ? x : expm1(x)         dy/dx = y>0 ? 1 : y+1
//   softplus:  y = x>20 ? x : log1p(exp(x))   dy/dx = 1-exp(-y)        (torch beta=1, threshold=20)
//   sigmoid:   dy/dx = y(1-y);  tanh: 1-y^2;  relu: y>0;  leaky(0.1): y>0?1:0.1
//   selu:      y = l*(x>0?x:a*expm1(x))       dy/dx = y>0 ? l : y+l*a
//   softsign:  y = x/(1+|x|)                  dy/dx = (1-|y|)^2
// (reference table: src/blocks.py:21-24)
__device__ __forceinline__ float dv_act(int act, float x) {
    switch (act) {
        case DV_ACT_ELU: {
            // expm1 on ONE hardware transcendental (round 5; round 3's Kahan quotient took three -- exp, log, rcp -- and the
            // hidden layers' epilogues run this on every element: +37 us of VALU time on a 32768 x 600 product):
            //   x <= -0.5: exp(x) - 1 loses nothing that matters (|result| >= 0.39: an ulp of exp is <= 1.6 ulp of it);
            //   -0.5 < x <= 0: the Taylor polynomial to x^8 (next term x^9 / 9! < 1.1e-8 |x|: a fifth of an ulp).
            // Branch-free: both on min(x, 0), selected at the end.
            const float xm = fminf(x, 0.f);
            const float e = __expf(xm) - 1.f;
            float p = fmaf(xm, 2.48015873e-5f, 1.98412698e-4f);      // 1/8!, 1/7!
            p = fmaf(p, xm, 1.38888889e-3f);                          // 1/6!
            p = fmaf(p, xm, 8.33333333e-3f);                          // 1/5!
            p = fmaf(p, xm, 4.16666667e-2f);                          // 1/4!
            p = fmaf(p, xm, 1.66666667e-1f);                          // 1/3!
            p = fmaf(p, xm, 0.5f);
            p = fmaf(p, xm, 1.f);
            const float r = xm > -0.5f ? p * xm : e;
            return x > 0.f ? x : r;
        }
        case DV_ACT_SOFTPLUS: {
            // max(x, 0) + log1p(exp(-|x|)) on the hardware transcendentals (one v_exp, one v_log, one v_rcp instead of the
            // ~50 instructions of log1pf(expf(x)): the decoder's sigma head runs this on every (row, gene)); log1p by
            // Kahan's quotient, log(u) * e / (u - 1) with u = 1 + e, which keeps it accurate to a few ulp for tiny e
            const float e = __expf(-fabsf(x)), u = 1.f + e, d = u - 1.f;
            const float l = d == 0.f ? e : __logf(u) * __fdividef(e, d);
            return x > 20.f ? x : fmaxf(x, 0.f) + l;
        }
        case DV_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
        case DV_ACT_TANH: return tanhf(x);
        case DV_ACT_RELU: return x > 0.f ? x : 0.f;
        case DV_ACT_LEAKY_RELU: return x > 0.f ? x : 0.1f * x;
        case DV_ACT_SELU: {
            const float l = 1.0507009873554804934193349852946f, a = 1.6732632423543772848170429916717f;
            return l * (x > 0.f ? x : a * expm1f(x));
        }
        case DV_ACT_SOFTSIGN: return x / (1.f + fabsf(x));
        case DV_ACT_COS: return cosf(x);
        default: return x;
    }
}

__device__ __forceinline__ float dv_dact_from_y(int act, float y) {
    switch (act) {
        case DV_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
        case DV_ACT_SOFTPLUS: return 1.f - expf(-y);
        case DV_ACT_SIGMOID: return y * (1.f - y);
        case DV_ACT_TANH: return 1.f - y * y;
        case DV_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case DV_ACT_LEAKY_RELU: return y > 0.f ? 1.f : 0.1f;
        case DV_ACT_SELU: {
            const float l = 1.0507009873554804934193349852946f, a = 1.6732632423543772848170429916717f;
            return y > 0.f ? l : y + l * a;
        }
        case DV_ACT_SOFTSIGN: {
            float t = 1.f - fabsf(y);
            return t * t;
        }
        default: return 1.f;
    }
}

// ---------------------------------------------------------------- wave reductions
__device__ __forceinline__ float dv_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;   // valid in lane 0
}

__device__ __forceinline__ float dv_wave_sum_all(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;   // valid in every lane
}
// ---- chain ordering carried by launches (dv_wait / dv_publish arguments)
// every workgroup of the launch parks on another chain's flag first (thread 0 polls, bounded; see dv_flag_wait)
__device__ __forceinline__ void park_block(const dv_wait& pk) {
    if (pk.flag == nullptr) return;
    if (threadIdx.x == 0) {
        const int want = pk.ctr[0] + pk.add;
        const long long t0 = wall_clock64();
        int n = 0;
        while (__hip_atomic_load(pk.flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want < 0) {
            __builtin_amdgcn_s_sleep(4);
            if (++n > pk.max_spins) {
                atomicExch(pk.err, 1);
                break;
            }
        }
        // (ticks parked: a no-return atomic -- a read-modify-write here would put one more round trip in front of the barrier)
        if (blockIdx.x == 0) atomicAdd(pk.err + 1, (int32_t)(wall_clock64() - t0));
    }
    __syncthreads();
    // every wave acquires: thread 0's acquire load has seen the flag (and invalidated this CU's vector L1 behind it); an
    // acquire FENCE for the others -- the cache invalidate without another trip to memory (the acquire load it replaces was
    // a round trip of ~1 us at the head of every parked launch)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}


// "this launch has started": workgroup 0 publishes on entry (see dv_flag_publish)
__device__ __forceinline__ void publish_block0(const dv_publish& pub) {
    if (pub.flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(pub.flag, pub.ctr[0] + pub.add, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}


