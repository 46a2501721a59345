// Device helpers shared by the HBM-bound kernels (elementwise.hip, backward.hip): SiLU, block reductions and the fixed-order
// combination of GroupNorm (mean, M2) partials.
#pragma once
#include "common.h"

namespace fc {

__device__ __forceinline__ float silu_e(float z) { return z / (1.0f + __expf(-z)); }

__device__ __forceinline__ float block_sum(float v, float* red /*[4]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global store
// (s_waitcnt vmcnt(0)) -- a store round trip of idle time when the barrier only guards an LDS reduction that follows the stores.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float block_sum_lds(float v, float* red /*[4]*/) {   // 256 threads
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    lds_barrier();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    lds_barrier();
    return red[0] + red[1] + red[2] + red[3];
}

// (mean, M2) partials of a GroupNorm group in two halves: request() only issues the loads of the first kPartPre partial pairs (a
// following block of loads then shares their round trip), finish() combines T equal-count partials in slot order (slots beyond
// kPartPre, rare, are read there).
constexpr int kPartPre = 8;
struct PartPre { float m[kPartPre], q[kPartPre]; };
__device__ __forceinline__ void partials_request(const SrcXform& xf, int b, int g, PartPre& r) {
    const float* sp = xf.stats + (size_t)(b * xf.G + g) * xf.T * 2;
#pragma unroll
    for (int t = 0; t < kPartPre; ++t) {
        r.m[t] = t < xf.T ? sp[2 * t] : 0.f;
        r.q[t] = t < xf.T ? sp[2 * t + 1] : 0.f;
    }
}
__device__ __forceinline__ void partials_finish(const SrcXform& xf, int b, int g, const PartPre& r, float* mean_out, float* rstd_out) {
    const float* sp = xf.stats + (size_t)(b * xf.G + g) * xf.T * 2;
    float sm = 0.f;
#pragma unroll
    for (int t = 0; t < kPartPre; ++t) if (t < xf.T) sm += r.m[t];
    for (int t = kPartPre; t < xf.T; ++t) sm += sp[2 * t];
    const float mean = sm / (float)xf.T;
    float m2 = 0.f, dv = 0.f;
#pragma unroll
    for (int t = 0; t < kPartPre; ++t)
        if (t < xf.T) {
            const float d = r.m[t] - mean;
            m2 += r.q[t];
            dv += d * d;
        }
    for (int t = kPartPre; t < xf.T; ++t) {
        const float d = sp[2 * t] - mean;
        m2 += sp[2 * t + 1];
        dv += d * d;
    }
    const float var = (m2 + xf.n_t * dv) / (xf.n_t * (float)xf.T);
    *mean_out = mean;
    *rstd_out = 1.0f / sqrtf(var + xf.eps);
}

// combine T equal-count (mean, M2) partials of group g of sample b.  All (up to kPartPre) partial pairs are requested before the
// first one is used: written as `for t < T: sum += sp[2 t]` the loads become T dependent round trips (the trip count is a run-time
// value, so the compiler keeps the loop), twice over -- several microseconds at the head of every kernel that normalises its input.
__device__ __forceinline__ void combine_partials(const SrcXform& xf, int b, int g, float* mean_out, float* rstd_out) {
    PartPre r;
    partials_request(xf, b, g, r);
    partials_finish(xf, b, g, r, mean_out, rstd_out);
}

__device__ __forceinline__ float silu_grad_e(float z) {   // d/dz [z sigmoid(z)]
    const float s = 1.0f / (1.0f + __expf(-z));
    return s * (1.0f + z * (1.0f - s));
}

}  // namespace fc
