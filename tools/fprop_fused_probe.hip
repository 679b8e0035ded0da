// Round-6 probe: the side chain's fprop block forward -- four dependent 450-row Linear layers
// (102 -> 200 -> 200 | 102 -> 200 -> 200, src/DrVAE.py:333-365) -- as
//   (a) four launches of one 32x32-tile kernel (what the step does today, minus its fused epilogues), against
//   (b) ONE launch that keeps the chip as wide: the same 15 x 7 = 105 workgroups, each running its tile of every layer,
//       layer n+1's tile parked on a per-row-block completion counter of layer n; operands written through (sc1 stores,
//       every storing wave drained, ONE agent-scope add per workgroup) and read back with sc1 loads (no acquire fence:
//       MI355X_MICROARCH.md "Valid forms", third table row), and
//   (c) the same single launch with the price list's other form: plain loads behind ONE agent-scope acquire fence.
// Same tile body in all three (fragments straight from L2, v_mfma_f32_32x32x2_f32, four waves split K, LDS reduction, ELU,
// 16-B stores), outputs compared bitwise.  Timed from a hipGraph (50 chains per replay) on the whole chip and on a
// 64-CU-masked stream (the side chain's reserve).
//   hipcc --offload-arch=gfx950 -O3 -o tools/fprop_fused_probe tools/fprop_fused_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

struct Layer {
    const float* A;   // [M, lda] input rows (k-contiguous, K a multiple of 8, zero padded)
    int lda;
    const float* B;   // [N, ldb] weight rows
    int ldb;
    int K8;           // K / 8
    float* C;         // [M, ldc]
    int ldc;
    int N;
    int act;          // 1 = ELU
};
struct Chain {
    Layer l[4];
    int M, tiles_n;
    int* cnt;         // [4][row blocks] completion counters (monotonic over launches)
    int* epoch;       // launches completed so far
    int* done;        // workgroups through with the last layer (monotonic)
    int* err;
};

__device__ __forceinline__ f32x4 ld_sc1(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sc1(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// one 32 x 32 output tile of one layer by 256 threads.  MODE 0: plain loads / stores; 1: A by sc1 loads, C by sc1 stores;
// 2: A by plain loads (behind the caller's acquire fence), C by sc1 stores
template <int MODE>
__device__ __forceinline__ void tile(const Layer& L, int M, int rb, int ct, float* red) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int ra = min(rb * 32 + i, M - 1), rbn = min(ct * 32 + i, L.N - 1);
    const float* pa = L.A + (int64_t)ra * L.lda + 4 * h;
    const float* pb = L.B + (int64_t)rbn * L.ldb + 4 * h;
    constexpr int NJ = 7;      // <= 25 chunks of 8 over 4 waves
    f32x4 a[NJ], b[NJ];
    const float* qa[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = w + 4 * j, cc = c < L.K8 ? c : 0;
        b[j] = *reinterpret_cast<const f32x4*>(pb + 8 * cc);
        qa[j] = pa + 8 * cc;
        if (MODE != 1) a[j] = *reinterpret_cast<const f32x4*>(qa[j]);
    }
    if (MODE == 1)      // the seven sc1 loads and their wait in ONE asm block (the compiler does not count vmcnt for asm loads)
        asm volatile(
            "global_load_dwordx4 %0, %7, off sc1\n\tglobal_load_dwordx4 %1, %8, off sc1\n\tglobal_load_dwordx4 %2, %9, off sc1\n\t"
            "global_load_dwordx4 %3, %10, off sc1\n\tglobal_load_dwordx4 %4, %11, off sc1\n\tglobal_load_dwordx4 %5, %12, off sc1\n\t"
            "global_load_dwordx4 %6, %13, off sc1\n\ts_waitcnt vmcnt(0)"
            : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6])
            : "v"(qa[0]), "v"(qa[1]), "v"(qa[2]), "v"(qa[3]), "v"(qa[4]), "v"(qa[5]), "v"(qa[6])
            : "memory");
    f32x16 acc = {0};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const bool ok = w + 4 * j < L.K8;
        const f32x4 av = ok ? a[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b[j].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b[j].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b[j].w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(w * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 33 + i] = acc[r];
    __syncthreads();
    const int row = tid >> 3, c0 = (tid & 7) * 4;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) s += red[(ww * 32 + row) * 33 + c0 + e];
        if (L.act == 1) s = s > 0.f ? s : __expf(s) - 1.f;
        v[e] = s;
    }
    const int gr = rb * 32 + row, gc = ct * 32 + c0;
    if (gr < M && gc < L.N) {      // (N a multiple of 4: whole 16-B groups)
        float* pc = L.C + (int64_t)gr * L.ldc + gc;
        if (MODE != 0)
            st_sc1(pc, v);
        else
            *reinterpret_cast<f32x4*>(pc) = v;
    }
    __syncthreads();      // red is reused by the next tile
}

__global__ __launch_bounds__(256) void layer_kernel(const Chain ch, int s) {
    __shared__ float red[4 * 32 * 33];
    const int rb = blockIdx.x / ch.tiles_n, ct = blockIdx.x % ch.tiles_n;
    tile<0>(ch.l[s], ch.M, rb, ct, red);
}

// ACQ: plain loads behind an agent-scope acquire fence instead of sc1 loads
template <bool ACQ>
__global__ __launch_bounds__(256) void chain_kernel(const Chain ch) {
    __shared__ float red[4 * 32 * 33];
    __shared__ int s_epoch;
    const int rb = blockIdx.x / ch.tiles_n, ct = blockIdx.x % ch.tiles_n, nrb = (ch.M + 31) / 32;
    if (threadIdx.x == 0) s_epoch = __hip_atomic_load(ch.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int target = (s_epoch + 1) * ch.tiles_n;
    for (int s = 0; s < 4; ++s) {
        if (s > 0) {
            if (threadIdx.x == 0) {
                int n = 0;
                while (__hip_atomic_load(ch.cnt + (s - 1) * nrb + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++n > 2000000) {
                        atomicExch(ch.err, 1);
                        break;
                    }
                }
                if (ACQ) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __syncthreads();
        }
        if (ACQ)
            tile<2>(ch.l[s], ch.M, rb, ct, red);        // plain loads behind the fence, write-through stores
        else
            tile<1>(ch.l[s], ch.M, rb, ct, red);        // sc1 loads, write-through stores
        // every storing wave drains its write-through stores, then ONE add per workgroup
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (s < 3)
                __hip_atomic_fetch_add(ch.cnt + s * nrb + rb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else {
                const int old = __hip_atomic_fetch_add(ch.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((old + 1) % (int)gridDim.x == 0)
                    __hip_atomic_fetch_add(ch.epoch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ void empty_kernel(int* p) {
    if (p && threadIdx.x == 9999) p[0] = 1;
}

template <typename F>
static float time_graph(hipStream_t st, F body, int n) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) body(st);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return best * 1e3f / n;
}

int main() {
    const int M = 450, H = 200, K1 = 104;
    const int nrb = (M + 31) / 32, tn = (H + 31) / 32, nwg = nrb * tn;
    std::vector<float> hx((size_t)M * K1), hw1((size_t)H * K1), hw2((size_t)H * H), hw3((size_t)H * K1), hw4((size_t)H * H);
    srand(3);
    auto rnd = [] { return (rand() / (float)RAND_MAX - 0.5f); };
    for (auto& v : hx) v = rnd();
    for (int r = 0; r < M; ++r) hx[(size_t)r * K1 + 102] = hx[(size_t)r * K1 + 103] = 0.f;
    for (auto& v : hw1) v = rnd() * 0.2f;
    for (auto& v : hw2) v = rnd() * 0.15f;
    for (auto& v : hw3) v = rnd() * 0.2f;
    for (auto& v : hw4) v = rnd() * 0.15f;
    float *x, *w1, *w2, *w3, *w4, *o[2][4];
    CK(hipMalloc(&x, hx.size() * 4));
    CK(hipMalloc(&w1, hw1.size() * 4));
    CK(hipMalloc(&w2, hw2.size() * 4));
    CK(hipMalloc(&w3, hw3.size() * 4));
    CK(hipMalloc(&w4, hw4.size() * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w1, hw1.data(), hw1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w3, hw3.data(), hw3.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w4, hw4.data(), hw4.size() * 4, hipMemcpyHostToDevice));
    for (int v = 0; v < 2; ++v)
        for (int s = 0; s < 4; ++s) {
            CK(hipMalloc(&o[v][s], (size_t)M * H * 4));
            CK(hipMemset(o[v][s], 0, (size_t)M * H * 4));
        }
    int* sync;
    CK(hipMalloc(&sync, (4 * nrb + 16) * 4));
    CK(hipMemset(sync, 0, (4 * nrb + 16) * 4));
    auto mk = [&](int v) {
        Chain c;
        c.M = M;
        c.tiles_n = tn;
        c.l[0] = Layer{x, K1, w1, K1, K1 / 8, o[v][0], H, H, 1};
        c.l[1] = Layer{o[v][0], H, w2, H, H / 8, o[v][1], H, H, 0};
        c.l[2] = Layer{o[v][1], H, w3, K1, K1 / 8, o[v][2], H, H, 1};     // (the "sample": the first 104 columns of the heads)
        c.l[3] = Layer{o[v][2], H, w4, H, H / 8, o[v][3], H, H, 0};
        c.cnt = sync;
        c.epoch = sync + 4 * nrb;
        c.done = sync + 4 * nrb + 4;
        c.err = sync + 4 * nrb + 8;
        return c;
    };
    Chain ca = mk(0), cb = mk(1);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    // a CU-masked stream like the side chain's reserve: the first 64 mask bits = 2 CUs of every shader engine of every XCD
    hipStream_t st64;
    uint32_t mask[8] = {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0};
    CK(hipExtStreamCreateWithCUMask(&st64, 8, mask));

    // correctness: (a) vs (b) vs (c) bitwise, (a) vs a host fp64 reference on a sample
    for (int s = 0; s < 4; ++s) hipLaunchKernelGGL(layer_kernel, dim3(nwg), dim3(256), 0, st, ca, s);
    hipLaunchKernelGGL(chain_kernel<false>, dim3(nwg), dim3(256), 0, st, cb);
    CK(hipStreamSynchronize(st));
    std::vector<float> ra((size_t)M * H), rb_((size_t)M * H);
    CK(hipMemcpy(ra.data(), o[0][3], ra.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(rb_.data(), o[1][3], rb_.size() * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < ra.size(); ++i) diff += ra[i] != rb_[i];
    CK(hipMemset(o[1][3], 0, (size_t)M * H * 4));
    hipLaunchKernelGGL(chain_kernel<true>, dim3(nwg), dim3(256), 0, st, cb);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(rb_.data(), o[1][3], rb_.size() * 4, hipMemcpyDeviceToHost));
    size_t diff2 = 0;
    for (size_t i = 0; i < ra.size(); ++i) diff2 += ra[i] != rb_[i];
    {   // host reference of a few rows
        double worst = 0;
        for (int r : {0, 17, 449}) {
            std::vector<double> h1(H), h2(H), h3(H), h4(H);
            for (int n = 0; n < H; ++n) {
                double s = 0;
                for (int k = 0; k < K1; ++k) s += (double)hx[(size_t)r * K1 + k] * hw1[(size_t)n * K1 + k];
                h1[n] = s > 0 ? s : std::expm1(s);
            }
            for (int n = 0; n < H; ++n) {
                double s = 0;
                for (int k = 0; k < H; ++k) s += h1[k] * hw2[(size_t)n * H + k];
                h2[n] = s;
            }
            for (int n = 0; n < H; ++n) {
                double s = 0;
                for (int k = 0; k < K1; ++k) s += h2[k] * hw3[(size_t)n * K1 + k];
                h3[n] = s > 0 ? s : std::expm1(s);
            }
            for (int n = 0; n < H; ++n) {
                double s = 0;
                for (int k = 0; k < H; ++k) s += h3[k] * hw4[(size_t)n * H + k];
                h4[n] = s;
                worst = fmax(worst, fabs(s - ra[(size_t)r * H + n]));
            }
        }
        int herr;
        CK(hipMemcpy(&herr, cb.err, 4, hipMemcpyDeviceToHost));
        printf("check: four launches vs one launch (sc1 loads): %zu differing elements; vs one launch (acquire fence): %zu; "
               "max |err| vs host fp64 on 3 rows: %.2e; wait time-outs: %d\n", diff, diff2, worst, herr);
    }
    const int n = 50;
    for (int pass = 0; pass < 2; ++pass) {
        hipStream_t s_ = pass ? st64 : st;
        const float t_empty = time_graph(s_, [&](hipStream_t q) { hipLaunchKernelGGL(empty_kernel, dim3(nwg), dim3(256), 0, q, (int*)nullptr); }, 4 * n) * 4;
        const float ta = time_graph(s_, [&](hipStream_t q) {
            for (int s = 0; s < 4; ++s) hipLaunchKernelGGL(layer_kernel, dim3(nwg), dim3(256), 0, q, ca, s);
        }, n);
        const float t1 = time_graph(s_, [&](hipStream_t q) { hipLaunchKernelGGL(layer_kernel, dim3(nwg), dim3(256), 0, q, ca, 1); }, 4 * n);
        const float t0 = time_graph(s_, [&](hipStream_t q) { hipLaunchKernelGGL(layer_kernel, dim3(nwg), dim3(256), 0, q, ca, 0); }, 4 * n);
        printf("%s: the K = 104 layer with ELU re-issued: %.2f us each\n", pass ? "64-CU mask " : "whole chip ", t0);
        const float tb = time_graph(s_, [&](hipStream_t q) { hipLaunchKernelGGL(chain_kernel<false>, dim3(nwg), dim3(256), 0, q, cb); }, n);
        const float tc = time_graph(s_, [&](hipStream_t q) { hipLaunchKernelGGL(chain_kernel<true>, dim3(nwg), dim3(256), 0, q, cb); }, n);
        printf("%s: four empty %d-workgroup launches %.2f us | (a) four layer launches %.2f us (one 200x200 layer re-issued: %.2f us each) | "
               "(b) ONE launch, counters + sc1 loads %.2f us | (c) ONE launch, counters + acquire fence %.2f us\n",
               pass ? "64-CU mask " : "whole chip ", nwg, t_empty, ta, t1, tb, tc);
    }
    int herr;
    CK(hipMemcpy(&herr, cb.err, 4, hipMemcpyDeviceToHost));
    printf("wait time-outs after timing: %d\n", herr);
    return 0;
}
