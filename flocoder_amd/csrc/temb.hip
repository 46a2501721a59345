// Conditioning path of the velocity U-Net: sinusoidal time embedding + time MLP + class MLP (unet.py:18-30,
// 199-212,310-316) in one kernel, and every ResnetBlock's SiLU->Linear scale/shift projection (unet.py:79-82,
// 90-92) for the whole network in a second one.  Linear weights are stored transposed ([in][out]) so that
// consecutive lanes read consecutive addresses.
#include "common.h"

namespace fc {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// y[j] = bias[j] + sum_i x[i] * wt[i][j] for one output column j; 4 accumulators x 4-deep unroll keep 16 independent
// (coalesced across threads) weight loads in flight -- this kernel is pure latency, its weights arrive cold from HBM.
__device__ __forceinline__ float matvec_col(const float* __restrict__ x, const float* __restrict__ wt, int n_in, int ld, int j, float bias) {
    float s0 = bias, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = 0;
    for (; i + 16 <= n_in; i += 16) {
        float w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = wt[(size_t)(i + u) * ld + j];
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
            s0 += x[i + u] * w[u]; s1 += x[i + u + 1] * w[u + 1]; s2 += x[i + u + 2] * w[u + 2]; s3 += x[i + u + 3] * w[u + 3];
        }
    }
    for (; i < n_in; ++i) s0 += x[i] * wt[(size_t)i * ld + j];
    return (s0 + s1) + (s2 + s3);
}

// grid (B), 256 threads
__global__ void __launch_bounds__(256) temb_kernel(const TembArgs a) {
    extern __shared__ float sm[];   // e[dim] | h[td] | c0[td] | c1[td]
    float* e = sm;
    float* h = e + a.dim;
    float* c0 = h + a.td;
    float* c1 = c0 + a.td;
    const int b = blockIdx.x, tid = threadIdx.x, half = a.dim / 2;
    const float t = a.time[b];
    for (int i = tid; i < a.dim; i += 256) {
        const int k = i < half ? i : i - half;
        const float arg = t * a.freqs[k];   // table built on the host exactly as unet.py:26-27 does
        e[i] = i < half ? sinf(arg) : cosf(arg);
    }
    long cid = -1;
    if (a.class_ids && a.n_classes > 0 && !(a.null_from > 0 && b >= a.null_from))
        cid = a.class_ids[a.class_batch_mod > 0 ? b % a.class_batch_mod : b];
    if (cid >= a.n_classes) cid = -1;
    __syncthreads();
    for (int j = tid; j < a.td; j += 256) {
        h[j] = gelu_erf(matvec_col(e, a.w1t, a.dim, a.td, j, a.b1[j]));
        if (cid >= 0) c0[j] = a.emb[(size_t)cid * a.td + j];
    }
    __syncthreads();
    if (cid >= 0) {
        for (int j = tid; j < a.td; j += 256) c1[j] = gelu_erf(matvec_col(c0, a.cw1t, a.td, a.td, j, a.cb1[j]));
    }
    __syncthreads();
    for (int j = tid; j < a.td; j += 256) {
        float s = matvec_col(h, a.w2t, a.td, a.td, j, a.b2[j]);
        if (cid >= 0) s += matvec_col(c1, a.cw2t, a.td, a.td, j, a.cb2[j]);
        a.t_out[(size_t)b * a.td + j] = s;
    }
}

int temb_launch(const TembArgs& a, hipStream_t s) {
    if (a.dim < 4 || (a.dim & 1)) return fail(FC_E_SHAPE, "temb: dim must be even and >= 4");
    hipLaunchKernelGGL(temb_kernel, dim3(a.B), dim3(256), (size_t)(a.dim + 3 * a.td) * sizeof(float), s, a);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// grid (ceil(S/256), B)
__global__ void __launch_bounds__(256) ss_kernel(const float* t, const float* wt, const float* bias, float* ss, int td, int S) {
    extern __shared__ float st[];  // silu(t[b])
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < td; i += 256) {
        const float v = t[(size_t)b * td + i];
        st[i] = v / (1.0f + expf(-v));
    }
    __syncthreads();
    if (j >= S) return;
    ss[(size_t)b * S + j] = matvec_col(st, wt, td, S, j, bias[j]);
}

int ss_launch(const float* t, const float* wt, const float* bias, float* ss, int B, int td, int S, hipStream_t s) {
    hipLaunchKernelGGL(ss_kernel, dim3(cdiv(S, 256), B), dim3(256), (size_t)td * sizeof(float), s, t, wt, bias, ss, td, S);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc
