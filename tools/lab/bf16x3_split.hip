// LAB ONLY (tools/bf16x3_step.py; not part of libdrvae_hip.so, not part of the C-ABI): split an fp32 matrix into three bf16
// terms (hi + mid + lo, round-to-nearest-even each) and lay them out as the six K-segments of ONE bf16 GEMM that emulates the
// fp32 product: sum over (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid).  One pass: 4 B read + 12 B written per element.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/lab/libbf16x3.so tools/lab/bf16x3_split.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t rne_bf16(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// side 0 (left operand of the product): segments hi, hi, mid, hi, lo, mid;  side 1 (right): hi, mid, hi, lo, hi, mid

// axis 1: dst is [R, 6 C] (segments along the columns: the reduction index of src's use is its COLUMN index);
// axis 0: dst is [6 R, C] (segments along the rows: the reduction index is its ROW index)
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ src, int64_t ld, int R, int C,
                                                    uint16_t* __restrict__ dst, int axis, int side) {
    // 8 consecutive elements per thread: two 16-B loads, six 16-B non-temporal stores (one per K-segment)
    const int64_t C8 = C >> 3, total = (int64_t)R * C8;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / C8;
        const int c = (int)(e - r * C8) * 8;
        const f4 v0 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src + r * ld + c));
        const f4 v1 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src + r * ld + c + 4));
        const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        uint32_t t[3][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t h = rne_bf16(x[j]);
            const float r1 = x[j] - __uint_as_float(h << 16);
            const uint32_t m = rne_bf16(r1);
            const float r2 = r1 - __uint_as_float(m << 16);
            t[0][j] = h;
            t[1][j] = m;
            t[2][j] = rne_bf16(r2);
        }
        u4 pk[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            pk[k] = u4{t[k][0] | (t[k][1] << 16), t[k][2] | (t[k][3] << 16), t[k][4] | (t[k][5] << 16), t[k][6] | (t[k][7] << 16)};
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int k = side == 0 ? (s == 2 || s == 5 ? 1 : (s == 4 ? 2 : 0)) : (s == 1 || s == 5 ? 1 : (s == 3 ? 2 : 0));
            uint16_t* o = axis == 1 ? dst + r * (6 * (int64_t)C) + (int64_t)s * C + c : dst + ((int64_t)s * R + r) * C + c;
            __builtin_nontemporal_store(pk[k], reinterpret_cast<u4*>(o));
        }
    }
}

extern "C" int bf16x3_split(const float* src, int64_t ld, int R, int C, uint16_t* dst, int axis, int side, hipStream_t st) {
    if (!src || !dst || R <= 0 || C <= 0 || (C & 7) || (ld & 3)) return -1;
    const int64_t total = (int64_t)R * (C >> 3);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, ld, R, C, dst, axis, side);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
