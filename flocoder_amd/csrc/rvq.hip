// Residual vector quantisation, inference form (vector_quantize_pytorch.ResidualVQ as VQVAE.quantize uses it, codecs.py:456-467,
// 504-521; eval mode: no EMA / k-means / rotation trick / losses).  THIRD-PARTY ALGORITHM, PARITY UNPINNED: the package is absent
// offline; this follows its published definition -- per level the nearest codeword in Euclidean distance (first index on ties) of
// the running residual, z_q = sum of the chosen codewords.
//
// One thread per latent vector, reading / writing the NCHW boundary tensors directly (the reference permutes to [N, D] and back,
// codecs.py:506-520); all levels' codebooks sit in LDS.   grid (ceil(N/256)), LDS L*K*D floats
#include "common.h"

namespace fc {

constexpr int RVQ_MAXD = 16;

__global__ void __launch_bounds__(256) rvq_kernel(const float* z, const float* cb, float* zq, int64_t* idx, int B, int D, int HW, int K, int L) {
    extern __shared__ __attribute__((aligned(16))) float cs[];   // [L][K][D]
    for (int i = threadIdx.x; i < L * K * D; i += 256) cs[i] = cb[i];
    __syncthreads();
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= (long)B * HW) return;
    const int b = (int)(n / HW), p = (int)(n % HW);
    const float* zp = z + (size_t)b * D * HW + p;
    float r[RVQ_MAXD], q[RVQ_MAXD];
#pragma unroll
    for (int d = 0; d < RVQ_MAXD; ++d) { r[d] = d < D ? zp[(size_t)d * HW] : 0.f; q[d] = 0.f; }
    for (int l = 0; l < L; ++l) {
        const float* cl = cs + (size_t)l * K * D;
        float best = INFINITY;
        int bi = 0;
        for (int k = 0; k < K; ++k) {
            float dist = 0.f;
#pragma unroll
            for (int d = 0; d < RVQ_MAXD; ++d)
                if (d < D) { const float t = r[d] - cl[k * D + d]; dist += t * t; }
            if (dist < best) { best = dist; bi = k; }
        }
#pragma unroll
        for (int d = 0; d < RVQ_MAXD; ++d)
            if (d < D) { const float e = cl[bi * D + d]; r[d] -= e; q[d] += e; }
        if (idx) idx[n * L + l] = bi;
    }
    float* qp = zq + (size_t)b * D * HW + p;
#pragma unroll
    for (int d = 0; d < RVQ_MAXD; ++d)
        if (d < D) qp[(size_t)d * HW] = q[d];
}

}  // namespace fc

extern "C" int fc_rvq_quantize(const float* z_dev, const float* codebooks_dev, float* zq_out_dev, int64_t* indices_out_dev, int batch, int dim,
                               int hw, int codebook_size, int levels, void* stream) {
    using namespace fc;
    if (!z_dev || !codebooks_dev || !zq_out_dev || batch < 1 || hw < 1 || levels < 1 || codebook_size < 1) return fail(FC_E_ARG, "fc_rvq_quantize: bad argument");
    if (dim < 1 || dim > RVQ_MAXD) return fail(FC_E_SHAPE, "fc_rvq_quantize: embedding dim must be 1..16");
    const size_t lds = (size_t)levels * codebook_size * dim * sizeof(float);
    if (lds > 64 * 1024) return fail(FC_E_SHAPE, "fc_rvq_quantize: codebooks exceed 64 KB");
    const long n = (long)batch * hw;
    hipLaunchKernelGGL(rvq_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, static_cast<hipStream_t>(stream), z_dev, codebooks_dev,
                       zq_out_dev, indices_out_dev, batch, dim, hw, codebook_size, levels);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
