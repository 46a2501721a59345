// Greedy mini-batch OT pairing (ot.py:63-78 compute_ot_pairing_approximate):
//   d = cdist(source, target) (L2);  for i = 0..B-1:  perm[i] = argmin over still-unused j of d[i][j] (first minimum).
// Kernel 1 forms the BxB distance matrix with direct differences (no |x|^2+|y|^2-2xy cancellation).
// Kernel 2 is the inherently sequential sweep: ONE wave64, column j owned by lane j%64, the `used` set as one
// bit per owned column in a register, (value, index) lexicographic min by cross-lane shuffles -- no LDS,
// no barriers, next row prefetched while the current one is reduced.
#include "common.h"

namespace fc {

constexpr int OT_T = 16, OT_K = 64;

// grid (ceil(B/16), ceil(B/16)), 256 threads: thread (r, c) of a 16x16 tile
__global__ void __launch_bounds__(256) ot_dist_kernel(const float* src, const float* tgt, int B, long D, float* dist) {
    __shared__ float sa[OT_T][OT_K + 1], sb[OT_T][OT_K + 1];
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    const int i0 = blockIdx.y * OT_T, j0 = blockIdx.x * OT_T;
    float acc = 0.f;
    for (long k0 = 0; k0 < D; k0 += OT_K) {
        for (int e = threadIdx.x; e < OT_T * OT_K; e += 256) {
            const int rr = e / OT_K, kk = e % OT_K;
            const bool kin = k0 + kk < D;
            sa[rr][kk] = (kin && i0 + rr < B) ? src[(size_t)(i0 + rr) * D + k0 + kk] : 0.f;
            sb[rr][kk] = (kin && j0 + rr < B) ? tgt[(size_t)(j0 + rr) * D + k0 + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll 16
        for (int kk = 0; kk < OT_K; ++kk) { const float d = sa[r][kk] - sb[c][kk]; acc += d * d; }
        __syncthreads();
    }
    if (i0 + r < B && j0 + c < B) dist[(size_t)(i0 + r) * B + j0 + c] = sqrtf(acc);
}

// Small batches (the training step's 32 - 128 rows): a T x T tile per workgroup leaves 256 / T^2 threads per pair, which split the
// feature axis (lane-strided, coalesced) and meet in a fixed butterfly.  With 16 x 16 tiles a batch of 64 was 16 workgroups, each
// thread walking all D features alone: 234 us for 64 x 64 distances over 4096 features.
template <int T>
__global__ void __launch_bounds__(256) ot_dist_small_kernel(const float* src, const float* tgt, int B, long D, float* dist) {
    constexpr int KG = 256 / (T * T);
    const int pair = threadIdx.x / KG, kg = threadIdx.x % KG;
    const int i = blockIdx.y * T + pair / T, j = blockIdx.x * T + pair % T;
    const bool in = i < B && j < B;
    const float* a = src + (size_t)(in ? i : 0) * D;
    const float* b = tgt + (size_t)(in ? j : 0) * D;
    float acc = 0.f;
    for (long k = kg; k < D; k += KG) { const float d = a[k] - b[k]; acc += d * d; }
#pragma unroll
    for (int o = KG / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (in && kg == 0) dist[(size_t)i * B + j] = sqrtf(acc);
}

// 1 block of 64 threads; B <= 4096
__global__ void __launch_bounds__(64) ot_sweep_kernel(const float* dist, int B, long long* perm) {
    // the sweep is one dependent step per row; with the matrix in LDS (B <= 128) a step is an LDS read and six shuffles instead of a
    // round trip to memory (59 -> 15 us at B = 64)
    __shared__ float sd[128 * 128];
    const int lane = threadIdx.x, per = (B + 63) / 64;
    const bool staged = B <= 128;
    if (staged) {
        for (int e = lane; e < B * B; e += 64) sd[e] = dist[e];
        __syncthreads();
    }
    unsigned long long used = 0ull;
    for (int i = 0; i < B; ++i) {
        const float* row = staged ? sd + i * B : dist + (size_t)i * B;
        float best = INFINITY;
        int bj = 0x7fffffff;
        for (int q = 0; q < per; ++q) {
            const int j = q * 64 + lane;
            if (j < B && !((used >> q) & 1ull)) {
                const float v = row[j];
                if (v < best) { best = v; bj = j; }   // ascending j per lane: keeps the first minimum
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o);
            const int oj = __shfl_xor(bj, o);
            if (ov < best || (ov == best && oj < bj)) { best = ov; bj = oj; }
        }
        if (bj == 0x7fffffff) {   // no finite minimum in this row (NaN / inf distances): take the first unused column, keep perm a permutation
            for (int q = 0; q < per; ++q) {
                const int j = q * 64 + lane;
                if (j < B && !((used >> q) & 1ull)) { bj = j; break; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int oj = __shfl_xor(bj, o); bj = oj < bj ? oj : bj; }
        }
        if ((bj & 63) == lane) used |= 1ull << (bj >> 6);
        if (lane == 0) perm[i] = bj;
    }
}

int ot_launch(const float* src, const float* tgt, int B, int64_t D, float* dist, int64_t* perm, hipStream_t s) {
    if (B < 1 || B > 4096) return fail(FC_E_SHAPE, "ot: batch must be in [1, 4096]");
    if (B <= 64) hipLaunchKernelGGL(ot_dist_small_kernel<4>, dim3(cdiv(B, 4), cdiv(B, 4)), dim3(256), 0, s, src, tgt, B, (long)D, dist);
    else if (B <= 128) hipLaunchKernelGGL(ot_dist_small_kernel<8>, dim3(cdiv(B, 8), cdiv(B, 8)), dim3(256), 0, s, src, tgt, B, (long)D, dist);
    else hipLaunchKernelGGL(ot_dist_kernel, dim3(cdiv(B, OT_T), cdiv(B, OT_T)), dim3(256), 0, s, src, tgt, B, (long)D, dist);
    FC_HIP(hipGetLastError());
    hipLaunchKernelGGL(ot_sweep_kernel, dim3(1), dim3(64), 0, s, dist, B, reinterpret_cast<long long*>(perm));
    FC_HIP(hipGetLastError());
    return FC_OK;
}

int ot_sweep_only_launch(const float* dist, int B, int64_t* perm, hipStream_t s) {
    if (B < 1 || B > 4096) return fail(FC_E_SHAPE, "ot: batch must be in [1, 4096]");
    hipLaunchKernelGGL(ot_sweep_kernel, dim3(1), dim3(64), 0, s, dist, B, reinterpret_cast<long long*>(perm));
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc
