// Conditioning path of the velocity U-Net: sinusoidal time embedding + time MLP + class MLP (unet.py:18-30,
// 199-212,310-316) in two launches (hidden layers, then the sum of the two output layers), and every ResnetBlock's SiLU->Linear scale/shift
// projection (unet.py:79-82,90-92) for the whole network in a third.  Linear weights are stored transposed ([in][out]) so that
// consecutive lanes read consecutive addresses.
#include "common.h"

namespace fc {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// ---- batched form -------------------------------------------------------------------------------------------------------------
// One sample per workgroup made every CU pull the whole weight matrix through its own L1 (3.25 MB at 64 B/clk = 21 us at dim 128,
// and 64x the L2 traffic).  Here a workgroup owns 64 output columns for SB samples: 512 threads = 64 columns x 8 K-splits (one
// wave per split), the SB input vectors sit in LDS as [k][SB] so that one weight load feeds SB FMAs off two broadcast
// ds_read_b128, and the eight partial sums meet in LDS.  Inside a sampler step these kernels find their weights cold (a whole
// forward's traffic has passed through the L2 since the last use), so what matters is the number of DEPENDENT load rounds per
// wave: 32 rows are requested at a time, which makes it one or two rounds at dim 32 (was eight of 8 rows on four waves).
constexpr int kCondSB = 8, kCondKS = 8, kCondThreads = 64 * kCondKS, kCondPF = 32;
static_assert(kCondThreads % kCondSB == 0, "a thread stages one sample");

template <int SB>
__device__ __forceinline__ void dense_rows(const float* __restrict__ xs, const float* __restrict__ wt, int ld, int j, int k0, int k1,
                                           float (&acc)[SB]) {
    int k = k0;
    for (; k + kCondPF <= k1; k += kCondPF) {
        float w[kCondPF];
#pragma unroll
        for (int u = 0; u < kCondPF; ++u) w[u] = wt[(size_t)(k + u) * ld + j];
#pragma unroll
        for (int u = 0; u < kCondPF; ++u) {
            const float4* xv = reinterpret_cast<const float4*>(xs + (size_t)(k + u) * SB);
#pragma unroll
            for (int q = 0; q < SB / 4; ++q) {
                const float4 x = xv[q];
                acc[q * 4 + 0] += x.x * w[u]; acc[q * 4 + 1] += x.y * w[u]; acc[q * 4 + 2] += x.z * w[u]; acc[q * 4 + 3] += x.w * w[u];
            }
        }
    }
    for (; k + 4 <= k1; k += 4) {
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = wt[(size_t)(k + u) * ld + j];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < SB; ++q) acc[q] += xs[(size_t)(k + u) * SB + q] * w[u];
    }
    for (; k < k1; ++k) {
        const float w = wt[(size_t)k * ld + j];
#pragma unroll
        for (int q = 0; q < SB; ++q) acc[q] += xs[(size_t)k * SB + q] * w;
    }
}

// K rows split over the waves in contiguous parts (rounded up to whole rows)
__device__ __forceinline__ void k_range(int K, int wave, int& k0, int& k1) {
    const int per = (K + kCondKS - 1) / kCondKS;
    k0 = min(K, wave * per);
    k1 = min(K, k0 + per);
}

// sum the waves' partials in wave order: red[wave][SB][64]; afterwards thread (wave, lane) owns sample `wave` of column `lane`
template <int SB>
__device__ __forceinline__ float meet(float* red, const float (&acc)[SB], int wave, int lane) {
    static_assert(SB == kCondKS, "one sample per wave after the meeting");
#pragma unroll
    for (int q = 0; q < SB; ++q) red[(wave * SB + q) * 64 + lane] = acc[q];
    __syncthreads();
    float s[kCondKS];
#pragma unroll
    for (int w = 0; w < kCondKS; ++w) s[w] = red[(w * SB + wave) * 64 + lane];
    const float out = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    return out;
}

__device__ __forceinline__ long class_of(const TembArgs& a, int b) {
    long cid = -1;
    const int q = a.rows_per_eval > 0 ? b % a.rows_per_eval : b;      // row inside its evaluation
    if (a.class_ids && a.n_classes > 0 && b < a.B && !(a.null_from > 0 && q >= a.null_from))
        cid = a.class_ids[a.class_batch_mod > 0 ? q % a.class_batch_mod : q];
    return cid >= a.n_classes ? -1 : cid;
}

// Stage n values into LDS, eight requests in flight per thread: xs[i] = post(*src(i)) (src(i) == nullptr: 0).  Written as a plain
// `for (i = tid; i < n; i += T) xs[i] = f(global[..])` loop the compiler waits for every load before the next one (tools/isa_audit.py).
template <class Src, class Post>
__device__ __forceinline__ void stage8(float* xs, int n, int tid, Src src, Post post) {
    for (int i0 = tid; i0 < n; i0 += 8 * kCondThreads) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * kCondThreads;
            const float* p = i < n ? src(i) : nullptr;
            v[u] = p ? *p : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * kCondThreads;
            if (i < n) xs[i] = post(v[u]);
        }
    }
}

// grid (ceil(td/64), ceil(B/SB), 1 or 2): z = 0 -> h = gelu(W1 sinemb(t) + b1), z = 1 -> c1 = gelu(CW1 emb[class] + cb1)
__global__ void __launch_bounds__(kCondThreads) cond_hidden_kernel(const TembArgs a, float* __restrict__ h, float* __restrict__ c1) {
    constexpr int SB = kCondSB;
    extern __shared__ float sm[];   // xs[2 td][SB] | red[KS][SB][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b0 = blockIdx.y * SB;
    const bool cls = blockIdx.z == 1;
    const int K = cls ? a.td : a.dim, half = a.dim / 2;
    float* xs = sm;
    float* red = sm + (size_t)2 * a.td * SB;
    const int myb = b0 + tid % SB;              // this thread's sample, the same in every pass (kCondThreads % SB == 0)
    if (cls) {
        const long cid = class_of(a, myb);      // read once: inside the staging loop every staged load waited for an id load
        const float* erow = a.emb + (size_t)(cid >= 0 ? cid : 0) * a.td;
        stage8(xs, K * SB, tid, [&](int i) -> const float* { return cid >= 0 ? erow + i / SB : nullptr; }, [](float v) { return v; });
    } else {
        const float tb = myb < a.B ? a.time[a.rows_per_eval > 0 ? myb / a.rows_per_eval : myb] : 0.f;
        for (int i = tid; i < K * SB; i += kCondThreads) {
            const int k = i / SB, kk = k < half ? k : k - half;
            const float arg = tb * a.freqs[kk];   // table built on the host exactly as unet.py:26-27 does
            xs[i] = myb < a.B ? (k < half ? sinf(arg) : cosf(arg)) : 0.f;
        }
    }
    __syncthreads();
    const int j = blockIdx.x * 64 + lane, jj = min(j, a.td - 1);
    float acc[SB] = {};
    int k0, k1;
    k_range(K, wave, k0, k1);
    dense_rows<SB>(xs, cls ? a.cw1t : a.w1t, a.td, jj, k0, k1, acc);
    const float o = meet<SB>(red, acc, wave, lane);
    const int b = b0 + wave;
    if (j < a.td && b < a.B) (cls ? c1 : h)[(size_t)b * a.td + j] = gelu_erf((cls ? a.cb1[j] : a.b1[j]) + o);
}

// grid (ceil(td/64), ceil(B/SB)): t_out = b2 + W2 h (+ cb2 + CW2 c1 where the sample has a class), the two products as ONE sweep
// over the stacked rows [h ; c1] . [W2 ; CW2] -- a sample without a class contributes exact zeros
__global__ void __launch_bounds__(kCondThreads) cond_out_kernel(const TembArgs a, const float* __restrict__ h, const float* __restrict__ c1, int with_class) {
    constexpr int SB = kCondSB;
    extern __shared__ float sm[];   // xs[2 td][SB] | red[KS][SB][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b0 = blockIdx.y * SB, td = a.td;
    float* xs = sm;
    float* red = sm + (size_t)2 * td * SB;
    const int K = with_class ? 2 * td : td;
    const int myb = b0 + tid % SB;                                         // this thread's sample, the same in every pass
    const bool my_class = with_class && class_of(a, myb) >= 0;
    const float* hrow = h + (size_t)min(myb, a.B - 1) * td;
    const float* crow = c1 + (size_t)min(myb, a.B - 1) * td;
    stage8(xs, K * SB, tid,
           [&](int i) -> const float* {
               const int k = i / SB;
               if (myb >= a.B || !(k < td || my_class)) return nullptr;
               return k < td ? hrow + k : crow + (k - td);
           },
           [](float v) { return v; });
    __syncthreads();
    const int j = blockIdx.x * 64 + lane, jj = min(j, td - 1);
    int k0, k1;
    k_range(K, wave, k0, k1);
    float acc[SB] = {};
    // rows [k0, k1) of the stacked matrix: the part below td from W2, the part above from CW2
    const int ka = min(k1, td), kb = max(k0, td);
    if (k0 < ka) dense_rows<SB>(xs, a.w2t, td, jj, k0, ka, acc);
    if (kb < k1) dense_rows<SB>(xs + (size_t)td * SB, a.cw2t, td, jj, kb - td, k1 - td, acc);
    const float o = meet<SB>(red, acc, wave, lane);
    const int b = b0 + wave;
    if (j < td && b < a.B) {
        float s = a.b2[j] + o;
        if (with_class && class_of(a, b) >= 0) s += a.cb2[j];
        a.t_out[(size_t)b * td + j] = s;
    }
}

__global__ void ss_kernel(const float* __restrict__ t, const float* __restrict__ wt, const float* __restrict__ bias, float* __restrict__ ss,
                          int B, int td, int S);

static int cond_lds(int td, size_t* lds) {
    *lds = ((size_t)2 * td * kCondSB + kCondKS * kCondSB * 64) * sizeof(float);
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
    hipLaunchKernelGGL(cond_hidden_kernel, grid, dim3(kCondThreads), lds, s, a, h, c1);
    hipLaunchKernelGGL(cond_out_kernel, dim3(grid.x, grid.y), dim3(kCondThreads), lds, s, a, h, c1, with_class);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// grid (ceil(S/64), ceil(B/SB)): ss[b][j] = bias[j] + sum_i silu(t[b][i]) wt[i][j]
__global__ void __launch_bounds__(kCondThreads) ss_kernel(const float* __restrict__ t, const float* __restrict__ wt, const float* __restrict__ bias,
                                                          float* __restrict__ ss, int B, int td, int S) {
    constexpr int SB = kCondSB;
    extern __shared__ float sm[];   // xs[2 td][SB] | red[KS][SB][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b0 = blockIdx.y * SB;
    float* xs = sm;
    float* red = sm + (size_t)2 * td * SB;
    const int myb = b0 + tid % SB;              // this thread's sample, the same in every pass
    const float* trow = t + (size_t)min(myb, B - 1) * td;
    stage8(xs, td * SB, tid, [&](int i) -> const float* { return myb < B ? trow + i / SB : nullptr; },
           [](float v) { return v / (1.0f + expf(-v)); });
    __syncthreads();
    const int j = blockIdx.x * 64 + lane, jj = min(j, S - 1);
    float acc[SB] = {};
    int k0, k1;
    k_range(td, wave, k0, k1);
    dense_rows<SB>(xs, wt, S, jj, k0, k1, acc);
    const float o = meet<SB>(red, acc, wave, lane);
    const int b = b0 + wave;
    if (j < S && b < B) ss[(size_t)b * S + j] = bias[j] + o;
}

int ss_launch(const float* t, const float* wt, const float* bias, float* ss, int B, int td, int S, hipStream_t s) {
    size_t lds;
    FC_TRY(cond_lds(td, &lds));
    hipLaunchKernelGGL(ss_kernel, dim3(cdiv(S, 64), cdiv(B, kCondSB)), dim3(kCondThreads), lds, s, t, wt, bias, ss, B, td, S);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc
