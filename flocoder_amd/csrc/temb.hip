// Conditioning path of the velocity U-Net: sinusoidal time embedding + time MLP + class MLP (unet.py:18-30,
// 199-212,310-316) in two launches (hidden layers, then the sum of the two output layers), and every ResnetBlock's SiLU->Linear scale/shift
// projection (unet.py:79-82,90-92) for the whole network in a third.  Linear weights are stored transposed ([in][out]) so that
// consecutive lanes read consecutive addresses.
#include "common.h"

namespace fc {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// ---- batched form -------------------------------------------------------------------------------------------------------------
// One sample per workgroup made every CU pull the whole weight matrix through its own L1 (3.25 MB at 64 B/clk = 21 us at dim 128,
// and 64x the L2 traffic).  Here a workgroup owns 64 output columns for SB samples: 256 threads = 64 columns x 4 K-quarters (one
// wave per quarter), the SB input vectors sit in LDS as [k][SB] so that one weight load feeds SB FMAs off two broadcast
// ds_read_b128, and the four partial sums meet in LDS.  Weight bytes through an L1 drop by SB, the dependent-load chain by 4.
constexpr int kCondSB = 8;

template <int SB>
__device__ __forceinline__ void dense_rows(const float* __restrict__ xs, const float* __restrict__ wt, int ld, int j, int k0, int k1,
                                           float (&acc)[SB]) {
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = wt[(size_t)(k + u) * ld + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4* xv = reinterpret_cast<const float4*>(xs + (size_t)(k + u) * SB);
#pragma unroll
            for (int q = 0; q < SB / 4; ++q) {
                const float4 x = xv[q];
                acc[q * 4 + 0] += x.x * w[u]; acc[q * 4 + 1] += x.y * w[u]; acc[q * 4 + 2] += x.z * w[u]; acc[q * 4 + 3] += x.w * w[u];
            }
        }
    }
    for (; k < k1; ++k) {
        const float w = wt[(size_t)k * ld + j];
#pragma unroll
        for (int q = 0; q < SB; ++q) acc[q] += xs[(size_t)k * SB + q] * w;
    }
}

// K rows split over the 4 waves in contiguous quarters (rounded up to whole rows)
__device__ __forceinline__ void k_range(int K, int wave, int& k0, int& k1) {
    const int per = (K + 3) >> 2;
    k0 = min(K, wave * per);
    k1 = min(K, k0 + per);
}

// sum the 4 waves' partials: red[wave][SB][64]; afterwards thread (wave, lane) owns samples {wave, wave + 4} of column lane
template <int SB>
__device__ __forceinline__ void meet(float* red, const float (&acc)[SB], int wave, int lane, float (&out)[SB / 4]) {
#pragma unroll
    for (int q = 0; q < SB; ++q) red[(wave * SB + q) * 64 + lane] = acc[q];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SB / 4; ++r) {
        const int q = wave + 4 * r;
        out[r] = (red[(0 * SB + q) * 64 + lane] + red[(1 * SB + q) * 64 + lane]) + (red[(2 * SB + q) * 64 + lane] + red[(3 * SB + q) * 64 + lane]);
    }
    __syncthreads();
}

__device__ __forceinline__ long class_of(const TembArgs& a, int b) {
    long cid = -1;
    if (a.class_ids && a.n_classes > 0 && b < a.B && !(a.null_from > 0 && b >= a.null_from))
        cid = a.class_ids[a.class_batch_mod > 0 ? b % a.class_batch_mod : b];
    return cid >= a.n_classes ? -1 : cid;
}

// grid (ceil(td/64), ceil(B/SB), 1 or 2): z = 0 -> h = gelu(W1 sinemb(t) + b1), z = 1 -> c1 = gelu(CW1 emb[class] + cb1)
__global__ void __launch_bounds__(256) cond_hidden_kernel(const TembArgs a, float* __restrict__ h, float* __restrict__ c1) {
    constexpr int SB = kCondSB;
    extern __shared__ float sm[];   // xs[K][SB] | red[4][SB][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b0 = blockIdx.y * SB;
    const bool cls = blockIdx.z == 1;
    const int K = cls ? a.td : a.dim, half = a.dim / 2;
    float* xs = sm;
    float* red = sm + (size_t)a.td * SB;
    for (int i = tid; i < K * SB; i += 256) {
        const int k = i / SB, q = i % SB, b = b0 + q;
        float v = 0.f;
        if (b < a.B) {
            if (cls) {
                const long cid = class_of(a, b);
                if (cid >= 0) v = a.emb[(size_t)cid * a.td + k];
            } else {
                const int kk = k < half ? k : k - half;
                const float arg = a.time[b] * a.freqs[kk];   // table built on the host exactly as unet.py:26-27 does
                v = k < half ? sinf(arg) : cosf(arg);
            }
        }
        xs[i] = v;
    }
    __syncthreads();
    const int j = blockIdx.x * 64 + lane, jj = min(j, a.td - 1);
    float acc[SB] = {};
    int k0, k1;
    k_range(K, wave, k0, k1);
    dense_rows<SB>(xs, cls ? a.cw1t : a.w1t, a.td, jj, k0, k1, acc);
    float o[SB / 4];
    meet<SB>(red, acc, wave, lane, o);
    if (j >= a.td) return;
    const float bias = cls ? a.cb1[j] : a.b1[j];
#pragma unroll
    for (int r = 0; r < SB / 4; ++r) {
        const int b = b0 + wave + 4 * r;
        if (b < a.B) (cls ? c1 : h)[(size_t)b * a.td + j] = gelu_erf(bias + o[r]);
    }
}

// grid (ceil(td/64), ceil(B/SB)): t_out = (b2 + W2 h) + (cb2 + CW2 c1 where the sample has a class)
__global__ void __launch_bounds__(256) cond_out_kernel(const TembArgs a, const float* __restrict__ h, const float* __restrict__ c1, int with_class) {
    constexpr int SB = kCondSB;
    extern __shared__ float sm[];   // xs[td][SB] | red[4][SB][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b0 = blockIdx.y * SB, td = a.td;
    float* xs = sm;
    float* red = sm + (size_t)td * SB;
    const int j = blockIdx.x * 64 + lane, jj = min(j, td - 1);
    int k0, k1;
    k_range(td, wave, k0, k1);
    float ot[SB / 4], oc[SB / 4];
    for (int pass = 0; pass < (with_class ? 2 : 1); ++pass) {
        const float* src = pass ? c1 : h;
        for (int i = tid; i < td * SB; i += 256) {
            const int k = i / SB, b = b0 + i % SB;
            xs[i] = b < a.B ? src[(size_t)b * td + k] : 0.f;
        }
        __syncthreads();
        float acc[SB] = {};
        dense_rows<SB>(xs, pass ? a.cw2t : a.w2t, td, jj, k0, k1, acc);
        meet<SB>(red, acc, wave, lane, pass ? oc : ot);
    }
    if (j >= td) return;
#pragma unroll
    for (int r = 0; r < SB / 4; ++r) {
        const int b = b0 + wave + 4 * r;
        if (b >= a.B) continue;
        float s = a.b2[j] + ot[r];
        if (with_class && class_of(a, b) >= 0) s += a.cb2[j] + oc[r];
        a.t_out[(size_t)b * td + j] = s;
    }
}

__global__ void ss_kernel(const float* __restrict__ t, const float* __restrict__ wt, const float* __restrict__ bias, float* __restrict__ ss,
                          int B, int td, int S);

static int cond_lds(int td, size_t* lds) {
    *lds = ((size_t)td * kCondSB + 4 * kCondSB * 64) * sizeof(float);
    if (*lds > 160 * 1024) return fail(FC_E_SHAPE, "temb: time_dim too large for the LDS-resident conditioning kernels");
    return FC_OK;
}

// once per process, before any launch or graph capture (large `dim`: more than the default 64 KiB of dynamic LDS)
int temb_init() {
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cond_hidden_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cond_out_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ss_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return FC_OK;
}

int temb_launch(const TembArgs& a, float* h, float* c1, hipStream_t s) {
    if (a.dim < 4 || (a.dim & 1)) return fail(FC_E_SHAPE, "temb: dim must be even and >= 4");
    const int with_class = a.class_ids && a.n_classes > 0;
    size_t lds;
    FC_TRY(cond_lds(a.td, &lds));
    const dim3 grid(cdiv(a.td, 64), cdiv(a.B, kCondSB), with_class ? 2 : 1);
    hipLaunchKernelGGL(cond_hidden_kernel, grid, dim3(256), lds, s, a, h, c1);
    hipLaunchKernelGGL(cond_out_kernel, dim3(grid.x, grid.y), dim3(256), lds, s, a, h, c1, with_class);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// grid (ceil(S/64), ceil(B/SB)): ss[b][j] = bias[j] + sum_i silu(t[b][i]) wt[i][j]
__global__ void __launch_bounds__(256) ss_kernel(const float* __restrict__ t, const float* __restrict__ wt, const float* __restrict__ bias,
                                                 float* __restrict__ ss, int B, int td, int S) {
    constexpr int SB = kCondSB;
    extern __shared__ float sm[];   // xs[td][SB] | red[4][SB][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b0 = blockIdx.y * SB;
    float* xs = sm;
    float* red = sm + (size_t)td * SB;
    for (int i = tid; i < td * SB; i += 256) {
        const int k = i / SB, b = b0 + i % SB;
        const float v = b < B ? t[(size_t)b * td + k] : 0.f;
        xs[i] = v / (1.0f + expf(-v));
    }
    __syncthreads();
    const int j = blockIdx.x * 64 + lane, jj = min(j, S - 1);
    float acc[SB] = {};
    int k0, k1;
    k_range(td, wave, k0, k1);
    dense_rows<SB>(xs, wt, S, jj, k0, k1, acc);
    float o[SB / 4];
    meet<SB>(red, acc, wave, lane, o);
    if (j >= S) return;
#pragma unroll
    for (int r = 0; r < SB / 4; ++r) {
        const int b = b0 + wave + 4 * r;
        if (b < B) ss[(size_t)b * S + j] = bias[j] + o[r];
    }
}

int ss_launch(const float* t, const float* wt, const float* bias, float* ss, int B, int td, int S, hipStream_t s) {
    size_t lds;
    FC_TRY(cond_lds(td, &lds));
    hipLaunchKernelGGL(ss_kernel, dim3(cdiv(S, 64), cdiv(B, kCondSB)), dim3(256), lds, s, t, wt, bias, ss, B, td, S);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc
