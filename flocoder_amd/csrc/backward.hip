// Backward kernels of the flow training step (train_flow.py:350-397): everything except the convolution gradients
// (conv_wgrad.hip; data gradients reuse the forward implicit-GEMM kernel on flipped weights).  All HBM-bound or tiny:
//   * GroupNorm / FiLM / SiLU backward as a per-(sample, channel) reduction plus one elementwise pass, recomputing the
//     normalised activation from the raw convolution output and the forward's (mean, M2) partials -- nothing but the raw
//     tensors the forward keeps anyway is stored;
//   * LinearAttention / Attention backward (unet.py:99-150);
//   * the conditioning MLPs (dense layers with <= 256 rows);
//   * loss, gradient norm, Adam + EMA on flat parameter vectors (reference: torch.optim.Adam, clip_grad_norm_, EMA.update).
// Every reduction runs in a fixed order (no atomics): a step is bit-reproducible.
#include <cmath>
#include <cstdlib>
#include <string>

#include "common.h"
#include "stats_dev.h"

namespace fc {

constexpr int DH = 32;

static inline int grid_1d(size_t total, int cap = 4096) {
    const size_t g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}

// ===================================================================================================
// GroupNorm (+FiLM, +SiLU) backward.   Forward: u = (gamma*xhat + beta)*(sc+1) + sh,  y = act(u),  xhat = (h - mean)*rstd.
// Pass 1: per 64-pixel chunk, s1[c] = sum du, s2[c] = sum du*xhat with du = dy*act'(u) -> s12p[b][chunk][c][2].
// grid (nchunks, B); 256 threads = (256/CW) pixel rows x CW channels, CW = min(64, C rounded up to a power of two)
constexpr int kGnChunk = 64;
__global__ void __launch_bounds__(256) gn_bwd_reduce_kernel(const GnBwdArgs a, int CW, int nchunks) {
    __shared__ float r1[256], r2[256];
    const int b = blockIdx.y, chunk = blockIdx.x, lane = threadIdx.x % CW, row = threadIdx.x / CW, rows = 256 / CW;
    const int p0 = chunk * kGnChunk, p1 = min(a.HW, p0 + kGnChunk), cpg = a.C / a.xf.G;
    const bool act = a.xf.mode == 2;
    for (int c0 = 0; c0 < a.C; c0 += CW) {
        const int c = c0 + lane;
        float s1 = 0.f, s2 = 0.f;
        if (c < a.C) {
            float mean, rstd;
            combine_partials(a.xf, b, c / cpg, &mean, &rstd);
            float A = rstd * a.xf.gamma[c], Bv = a.xf.beta[c] - mean * A;
            if (a.xf.ss) {
                const float sc = a.xf.ss[(size_t)b * a.xf.ss_stride + c] + 1.0f;
                const float sh = a.xf.ss[(size_t)b * a.xf.ss_stride + a.C + c];
                A *= sc;
                Bv = Bv * sc + sh;
            }
            const float* hp = a.h + (size_t)b * a.HW * a.C + c;
            const float* dp = a.dy + (size_t)b * a.HW * a.C + c;
            for (int p = p0 + row; p < p1; p += rows) {
                const float hv = hp[(size_t)p * a.C];
                float du = dp[(size_t)p * a.C];
                if (act) du *= silu_grad_e(A * hv + Bv);
                s1 += du;
                s2 += du * ((hv - mean) * rstd);
            }
        }
        __syncthreads();
        r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
        __syncthreads();
        if (row == 0 && c < a.C) {
            float t1 = 0.f, t2 = 0.f;
            for (int r = 0; r < rows; ++r) { t1 += r1[r * CW + lane]; t2 += r2[r * CW + lane]; }
            float* d = a.s12p + (((size_t)b * nchunks + chunk) * a.C + c) * 2;
            d[0] = t1;
            d[1] = t2;
        }
    }
}

// Pass 2: dh = rstd * (gamma' du - P1/n - xhat P2/n),  gamma' = gamma (sc+1),  P1 = sum_{c in g} gamma' s1,  P2 = sum gamma' s2.
// grid (bps, B); dynamic LDS: [G][4] (mean, rstd, k1, k2) | A[C] | Bv[C] | ga[C] | s1[C] | s2[C]
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const GnBwdArgs a, int bps, int nchunks) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* gt = sm;
    float* A = gt + 4 * a.xf.G;
    float* Bv = A + a.C;
    float* ga = Bv + a.C;
    float* f1 = ga + a.C;
    float* f2 = f1 + a.C;
    const int b = blockIdx.y, tid = threadIdx.x, cpg = a.C / a.xf.G;
    for (int g = tid; g < a.xf.G; g += 256) combine_partials(a.xf, b, g, &gt[4 * g], &gt[4 * g + 1]);
    for (int c = tid; c < a.C; c += 256) {   // fold the chunk partials of this sample (fixed order); block 0 keeps the result for the parameter gradients
        float t1 = 0.f, t2 = 0.f;
        for (int k = 0; k < nchunks; ++k) {
            const float* d = a.s12p + (((size_t)b * nchunks + k) * a.C + c) * 2;
            t1 += d[0];
            t2 += d[1];
        }
        f1[c] = t1; f2[c] = t2;
        if (blockIdx.x == 0) { a.s12[((size_t)b * a.C + c) * 2] = t1; a.s12[((size_t)b * a.C + c) * 2 + 1] = t2; }
    }
    __syncthreads();
    for (int c = tid; c < a.C; c += 256) {
        const int g = c / cpg;
        const float mean = gt[4 * g], rstd = gt[4 * g + 1];
        float gam = a.xf.gamma[c];
        float s = rstd * gam, t = a.xf.beta[c] - mean * s;
        if (a.xf.ss) {
            const float sc = a.xf.ss[(size_t)b * a.xf.ss_stride + c] + 1.0f;
            const float sh = a.xf.ss[(size_t)b * a.xf.ss_stride + a.C + c];
            s *= sc; t = t * sc + sh; gam *= sc;
        }
        A[c] = s; Bv[c] = t; ga[c] = gam * rstd;
    }
    __syncthreads();
    for (int g = tid; g < a.xf.G; g += 256) {
        float p1 = 0.f, p2 = 0.f;
        const float rstd = gt[4 * g + 1];
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            const float gp = ga[c] / rstd;     // gamma'
            p1 += gp * f1[c];
            p2 += gp * f2[c];
        }
        const float inv_n = 1.0f / ((float)a.HW * (float)cpg);
        gt[4 * g + 2] = rstd * p1 * inv_n;
        gt[4 * g + 3] = rstd * p2 * inv_n;
    }
    __syncthreads();
    const int per = a.HW * a.C / bps;
    const size_t base = (size_t)b * a.HW * a.C + (size_t)blockIdx.x * per;
    const bool act = a.xf.mode == 2;
    for (int i = 4 * tid; i < per; i += 1024) {
        const int c0 = (int)((blockIdx.x * (size_t)per + i) % a.C);
        const float4 hv = *reinterpret_cast<const float4*>(a.h + base + i);
        const float4 dv = *reinterpret_cast<const float4*>(a.dy + base + i);
        const float hs[4] = {hv.x, hv.y, hv.z, hv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j, g = c / cpg;
            float du = ds[j];
            if (act) du *= silu_grad_e(A[c] * hs[j] + Bv[c]);
            const float xh = (hs[j] - gt[4 * g]) * gt[4 * g + 1];
            o[j] = ga[c] * du - gt[4 * g + 2] - xh * gt[4 * g + 3];
        }
        float4* dst = reinterpret_cast<float4*>(a.dh + base + i);
        if (a.accumulate) { const float4 p = *dst; o[0] += p.x; o[1] += p.y; o[2] += p.z; o[3] += p.w; }
        if (a.plus) { const float4 p = *reinterpret_cast<const float4*>(a.plus + base + i); o[0] += p.x; o[1] += p.y; o[2] += p.z; o[3] += p.w; }
        *dst = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// The same two passes as ONE launch when a sample's tensor is small (HW*C <= 8192 floats, C a power of two): one workgroup per
// sample reduces, meets in LDS and applies, re-reading h / dy from cache.  At the training shapes (B = 32, 16x16 latents) the two
// launches above were pure latency, 55 times per step.  256 threads; thread t owns float4 columns c0 = 4t mod C of rows t*4/C + k*1024/C.
template <int NTH>
__global__ void __launch_bounds__(NTH) gn_bwd_small_kernel(const GnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* gt = sm;                       // [G][4]
    float* A = gt + 4 * a.xf.G;
    float* Bv = A + a.C;
    float* ga = Bv + a.C;
    float* f1 = ga + a.C;
    float* f2 = f1 + a.C;
    float* r1 = f2 + a.C;                 // [NTH][4]
    float* r2 = r1 + 4 * NTH;
    const int b = blockIdx.x, tid = threadIdx.x, C = a.C, cpg = C / a.xf.G, total = a.HW * C;
    const size_t base = (size_t)b * total;
    const bool act = a.xf.mode == 2;
    // operands that depend on nothing computed here: requested together
    const int c0 = (4 * tid) & (C - 1);
    float4 hv[8], dv[8];                  // up to 8192 / 1024 float4 per thread of each tensor
    const int nk = (total + 4 * NTH - 1) / (4 * NTH);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = 4 * tid + 4 * NTH * k;
        const bool in = k < nk && i < total;
        hv[k] = in ? *reinterpret_cast<const float4*>(a.h + base + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        dv[k] = in ? *reinterpret_cast<const float4*>(a.dy + base + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // What the result is added to (the gradient accumulated so far, the residual branch's gradient) depends on nothing computed here either:
    // requested with the operands above instead of behind the last barrier, where it was one more cold round trip on every launch of the
    // data-gradient chain (round 4).  Only in the 256-thread form: 1024 threads leave 128 registers per lane, the operands alone take 64.
    constexpr bool PRE_ADD = NTH == 256;
    float4 av[PRE_ADD ? 8 : 1];
    if (PRE_ADD) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = 4 * tid + 4 * NTH * k;
            const bool in = k < nk && i < total;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in && a.accumulate) v = *reinterpret_cast<const float4*>(a.dh + base + i);
            if (in && a.plus) { const float4 p = *reinterpret_cast<const float4*>(a.plus + base + i); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
            av[PRE_ADD ? k : 0] = v;
        }
    }
    // (the norm's parameters and the FiLM row of this thread's first channel: requested here, in front of the barrier, not behind it)
    float pgam = 0.f, pbet = 0.f, psc = 0.f, psh = 0.f;
    if (tid < C) {
        pgam = a.xf.gamma[tid]; pbet = a.xf.beta[tid];
        if (a.xf.ss) { psc = a.xf.ss[(size_t)b * a.xf.ss_stride + tid]; psh = a.xf.ss[(size_t)b * a.xf.ss_stride + C + tid]; }
    }
    for (int g = tid; g < a.xf.G; g += NTH) combine_partials(a.xf, b, g, &gt[4 * g], &gt[4 * g + 1]);
    __syncthreads();
    for (int c = tid; c < C; c += NTH) {
        const int g = c / cpg;
        const bool pre = c == tid;
        const float mean = gt[4 * g], rstd = gt[4 * g + 1];
        float gam = pre ? pgam : a.xf.gamma[c];
        float s = rstd * gam, t = (pre ? pbet : a.xf.beta[c]) - mean * s;
        if (a.xf.ss) {
            const float sc = (pre ? psc : a.xf.ss[(size_t)b * a.xf.ss_stride + c]) + 1.0f;
            const float sh = pre ? psh : a.xf.ss[(size_t)b * a.xf.ss_stride + C + c];
            s *= sc; t = t * sc + sh; gam *= sc;
        }
        A[c] = s; Bv[c] = t; ga[c] = gam * rstd;
    }
    __syncthreads();
    // pass 1: du = dy * act'(u) kept in dv; per-thread sums of du and du * xhat for its four channels
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float mean4[4], rstd4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int g = (c0 + j) / cpg; mean4[j] = gt[4 * g]; rstd4[j] = gt[4 * g + 1]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k >= nk || 4 * tid + 4 * NTH * k >= total) break;
        float hs[4] = {hv[k].x, hv[k].y, hv[k].z, hv[k].w}, ds[4] = {dv[k].x, dv[k].y, dv[k].z, dv[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float du = ds[j];
            if (act) du *= silu_grad_e(A[c0 + j] * hs[j] + Bv[c0 + j]);
            ds[j] = du;
            s1[j] += du;
            s2[j] += du * ((hs[j] - mean4[j]) * rstd4[j]);
        }
        dv[k] = make_float4(ds[0], ds[1], ds[2], ds[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { r1[4 * tid + j] = s1[j]; r2[4 * tid + j] = s2[j]; }
    __syncthreads();
    const int q = C >> 2, rows = NTH / q;          // threads t, t + q, t + 2q ... share the same four channels (C <= 1024: q <= 256)
    for (int c = tid; c < C; c += NTH) {
        float t1 = 0.f, t2 = 0.f;
        for (int r = 0; r < rows; ++r) { t1 += r1[(r * q + (c >> 2)) * 4 + (c & 3)]; t2 += r2[(r * q + (c >> 2)) * 4 + (c & 3)]; }
        f1[c] = t1; f2[c] = t2;
        a.s12[((size_t)b * C + c) * 2] = t1;
        a.s12[((size_t)b * C + c) * 2 + 1] = t2;
    }
    __syncthreads();
    for (int g = tid; g < a.xf.G; g += NTH) {
        float p1 = 0.f, p2 = 0.f;
        const float rstd = gt[4 * g + 1];
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            const float gp = ga[c] / rstd;
            p1 += gp * f1[c];
            p2 += gp * f2[c];
        }
        const float inv_n = 1.0f / ((float)a.HW * (float)cpg);
        gt[4 * g + 2] = rstd * p1 * inv_n;
        gt[4 * g + 3] = rstd * p2 * inv_n;
    }
    __syncthreads();
    float k1[4], k2[4], gaj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int g = (c0 + j) / cpg; k1[j] = gt[4 * g + 2]; k2[j] = gt[4 * g + 3]; gaj[j] = ga[c0 + j]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = 4 * tid + 4 * NTH * k;
        if (k >= nk || i >= total) break;
        const float hs[4] = {hv[k].x, hv[k].y, hv[k].z, hv[k].w}, ds[4] = {dv[k].x, dv[k].y, dv[k].z, dv[k].w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = gaj[j] * ds[j] - k1[j] - ((hs[j] - mean4[j]) * rstd4[j]) * k2[j];
        float4* dst = reinterpret_cast<float4*>(a.dh + base + i);
        if (PRE_ADD) { const float4 p = av[PRE_ADD ? k : 0]; o[0] += p.x; o[1] += p.y; o[2] += p.z; o[3] += p.w; }
        else {
            if (a.accumulate) { const float4 p = *dst; o[0] += p.x; o[1] += p.y; o[2] += p.z; o[3] += p.w; }
            if (a.plus) { const float4 p = *reinterpret_cast<const float4*>(a.plus + base + i); o[0] += p.x; o[1] += p.y; o[2] += p.z; o[3] += p.w; }
        }
        *dst = make_float4(o[0], o[1], o[2], o[3]);
    }
}

int gn_bwd_chunks(int HW) { return (HW + kGnChunk - 1) / kGnChunk; }

int gn_bwd_launch(const GnBwdArgs& a, hipStream_t s) {
    if (a.C & 3) return fail(FC_E_SHAPE, "gn_bwd: C must be a multiple of 4");
    if (!a.xf.mode || !a.xf.stats || !a.s12 || !a.s12p || !a.dh) return fail(FC_E_ARG, "gn_bwd: needs statistics and workspaces");
    static const bool no_small = std::getenv("FLOCODER_AMD_GN_BWD_SPLIT") != nullptr;
    if (!no_small && a.HW * a.C <= 8192 && (a.C & (a.C - 1)) == 0 && a.C <= 1024 && a.C % a.xf.G == 0) {
        const size_t lds = (size_t)(4 * a.xf.G + 5 * a.C + 2048) * sizeof(float);
        hipLaunchKernelGGL(gn_bwd_small_kernel<256>, dim3(a.B), dim3(256), lds, s, a);
        FC_HIP(hipGetLastError());
        return FC_OK;
    }
    // up to 32768 elements per sample (32 channels at 32x32): the same one-pass form with 1024 threads instead of reduce + apply
    if (!no_small && a.HW * a.C <= 32768 && (a.C & (a.C - 1)) == 0 && a.C <= 1024 && a.C % a.xf.G == 0) {
        const size_t lds = (size_t)(4 * a.xf.G + 5 * a.C + 8192) * sizeof(float);
        hipLaunchKernelGGL(gn_bwd_small_kernel<1024>, dim3(a.B), dim3(1024), lds, s, a);
        FC_HIP(hipGetLastError());
        return FC_OK;
    }
    int CW = 4;
    while (CW < a.C && CW < 64) CW *= 2;
    const int nchunks = gn_bwd_chunks(a.HW);
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(nchunks, a.B), dim3(256), 0, s, a, CW, nchunks);
    FC_HIP(hipGetLastError());
    const int bps = finalize_blocks_per_sample(a.HW, a.C);
    const size_t lds = (size_t)(4 * a.xf.G + 5 * a.C) * sizeof(float);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(bps, a.B), dim3(256), lds, s, a, bps, nchunks);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// Parameter gradients of one norm layer from its s12 table: dgamma[c] = sum_b (sc+1) s2, dbeta[c] = sum_b (sc+1) s1, and the
// FiLM gradients dss[b][c] = gamma s2 + beta s1 (scale), dss[b][C + c] = s1 (shift).      grid (ceil(C/256))
__global__ void __launch_bounds__(256) norm_param_grads_kernel(const float* s12, const float* gamma, const float* beta, const float* ss,
                                                               int ss_stride, float* dgamma, float* dbeta, float* dss, int B, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float dg = 0.f, db = 0.f;
    const float gm = gamma[c], bt = beta[c];
    for (int b = 0; b < B; ++b) {
        const float s1 = s12[((size_t)b * C + c) * 2], s2 = s12[((size_t)b * C + c) * 2 + 1];
        float sc = 1.0f;
        if (ss) {
            sc = ss[(size_t)b * ss_stride + c] + 1.0f;
            dss[(size_t)b * ss_stride + c] = gm * s2 + bt * s1;
            dss[(size_t)b * ss_stride + C + c] = s1;
        }
        dg += sc * s2;
        db += sc * s1;
    }
    dgamma[c] = dg;
    dbeta[c] = db;
}
int norm_param_grads_launch(const float* s12, const float* gamma, const float* beta, const float* ss, int ss_stride, float* dgamma,
                            float* dbeta, float* dss, int B, int C, hipStream_t s) {
    hipLaunchKernelGGL(norm_param_grads_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, s12, gamma, beta, ss, ss_stride, dgamma, dbeta, dss, B, C);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// The same for every norm layer of the network in one launch (a step has ~55 of them, each a few microseconds of work).
// grid (ceil(maxC/64), njobs); 256 threads = 4 batch rows x 64 channels
__global__ void __launch_bounds__(256) norm_param_grads_table_kernel(const NormJob* jobs, float* grads, float* dss_base, int ss_stride, int B) {
    __shared__ float rg[4][64], rb[4][64];
    const NormJob j = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63, row = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= j.C) return;
    float dg = 0.f, db = 0.f;
    if (c < j.C) {
        const float gm = j.gamma[c], bt = j.beta[c];
        for (int b = row; b < B; b += 4) {
            const float s1 = j.s12[((size_t)b * j.C + c) * 2], s2 = j.s12[((size_t)b * j.C + c) * 2 + 1];
            float sc = 1.0f;
            if (j.ss) {
                sc = j.ss[(size_t)b * ss_stride + c] + 1.0f;
                float* d = dss_base + (size_t)b * ss_stride + j.ss_col;
                d[c] = gm * s2 + bt * s1;
                d[j.C + c] = s1;
            }
            dg += sc * s2;
            db += sc * s1;
        }
    }
    rg[row][lane] = dg; rb[row][lane] = db;
    __syncthreads();
    if (row == 0 && c < j.C) {
        grads[j.dgamma + c] = (rg[0][lane] + rg[1][lane]) + (rg[2][lane] + rg[3][lane]);
        grads[j.dbeta + c] = (rb[0][lane] + rb[1][lane]) + (rb[2][lane] + rb[3][lane]);
    }
}
int norm_param_grads_table_launch(const NormJob* jobs_dev, int njobs, int maxC, float* grads, float* dss, int ss_stride, int B, hipStream_t s) {
    if (!njobs) return FC_OK;
    hipLaunchKernelGGL(norm_param_grads_table_kernel, dim3(cdiv(maxC, 64), njobs), dim3(256), 0, s, jobs_dev, grads, dss, ss_stride, B);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// ===================================================================================================
// LinearAttention backward (unet.py:137-149).  With p = softmax_d(q_raw), q = p/sqrt(32), k = softmax_n(k_raw):
//   ctx[d][e] = sum_n k[d][n] v[e][n],  out[e][n] = sum_d ctx[d][e] q[d][n]
//   dctx[d][e] = sum_n q[d][n] dout[e][n];  dq[d][n] = sum_e ctx[d][e] dout[e][n];  dk[d][n] = sum_e dctx[d][e] v[e][n];
//   dv[e][n] = sum_d dctx[d][e] k[d][n];   softmax_n backward needs sum_n k dk = sum_e dctx[d][e] ctx[d][e] =: rr[d].
// Kernel 1, grid (B*heads): column statistics of k_raw, dctx and rr.
template <int NW, int KV>        // waves; values of a k column a thread keeps in registers (n <= 2 NW KV runs from registers)
__global__ void __launch_bounds__(NW * 64) linattn_bwd_ctx_kernel(const float* qkv, const float* dout, const float* ctx, float* dctx, float* kst,
                                                                  float* rr, int n, int heads) {
    constexpr int NG = 2 * NW;                     // row groups of the column statistics
    __shared__ float red[NG][DH];
    __shared__ float kmax[DH];
    __shared__ float wred[(NW - 1) * 16 * 64];     // waves 1.. park their accumulators here
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x;
    const int C3 = 3 * heads * DH, CO = heads * DH;
    const float* qb = qkv + (size_t)b * n * C3 + h * DH;
    const float* kb = qb + heads * DH;
    const float* db = dout + (size_t)b * n * CO + h * DH;
    {   // max and 1 / sum exp of every k column over the n pixels.  Up to 1024 pixels a thread's share (n / 16 values) is requested in one
        // go and both passes run from registers: one memory round trip instead of 2 n / 8 dependent ones.
        const int d = tid & 31, grp = tid >> 5;
        const bool cached = n <= NG * KV;
        float kv[KV];
        float m = -INFINITY;
        if (cached) {
#pragma unroll
            for (int k = 0; k < KV; ++k) { const int i = grp + NG * k; kv[k] = i < n ? kb[(size_t)i * C3 + d] : -INFINITY; }
#pragma unroll
            for (int k = 0; k < KV; ++k) m = fmaxf(m, kv[k]);
        } else {
            for (int i = grp; i < n; i += NG) m = fmaxf(m, kb[(size_t)i * C3 + d]);
        }
        red[grp][d] = m;
        __syncthreads();
        if (tid < DH) {
            float mm = red[0][tid];
            for (int g = 1; g < NG; ++g) mm = fmaxf(mm, red[g][tid]);
            kmax[tid] = mm;
        }
        __syncthreads();
        float z = 0.f;
        const float km = kmax[d];
        if (cached) {
#pragma unroll
            for (int k = 0; k < KV; ++k) z += grp + NG * k < n ? __expf(kv[k] - km) : 0.f;
        } else {
            for (int i = grp; i < n; i += NG) z += __expf(kb[(size_t)i * C3 + d] - km);
        }
        __syncthreads();
        red[grp][d] = z;
        __syncthreads();
        if (tid < DH) {
            float zz = 0.f;
            for (int g = 0; g < NG; ++g) zz += red[g][tid];
            kst[((size_t)blockIdx.x * DH + tid) * 2] = kmax[tid];
            kst[((size_t)blockIdx.x * DH + tid) * 2 + 1] = 1.0f / zz;
        }
    }
    // dctx = q^T dout as a 32 x 32 x n product on the matrix pipe: k-step j covers pixels 2j, 2j+1 (one per half-wave), a lane holds
    // q[pix][d = lane & 31] (softmax over the 32 lanes of its half by shuffles) and dout[pix][e = lane & 31].  The eight waves take
    // every eighth k-step, eight of them (sixteen loads) in flight at a time, and meet in LDS in a fixed order.
    const float scale = 0.17677669529663687f;
    const int lane = tid & 63, half = lane >> 5, l31 = lane & 31, wave = tid >> 6;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int npairs = (n + 1) >> 1;
    constexpr int UB = NW;                         // k-steps (pairs of loads) a wave keeps in flight
    for (int j0 = wave; j0 < npairs; j0 += NW * UB) {
        float qv[UB], gv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int j = j0 + NW * u;
            const bool in = 2 * j + half < n;
            const size_t pix = (size_t)(in ? 2 * j + half : 0);
            qv[u] = in ? qb[pix * C3 + l31] : -INFINITY;
            gv[u] = in ? db[pix * CO + l31] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (j0 + NW * u >= npairs) break;                 // wave-uniform
            const bool in = 2 * (j0 + NW * u) + half < n;
            float m = qv[u];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            float e = in ? __expf(qv[u] - m) : 0.f, sum = e;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float pq = in ? e * (scale / sum) : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pq, gv[u], acc, 0, 0, 0);
        }
    }
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) wred[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r];
#pragma unroll
            for (int w = 0; w < NW - 1; ++w) v += wred[(w * 16 + r) * 64 + lane];
            const int d = (r & 3) + 8 * (r >> 2) + 4 * half;
            const size_t o = ((size_t)blockIdx.x * DH + d) * DH + l31;
            dctx[o] = v;
            float part = v * ctx[o];
#pragma unroll
            for (int w = 16; w > 0; w >>= 1) part += __shfl_xor(part, w);
            if (l31 == 0) rr[(size_t)blockIdx.x * DH + d] = part;
        }
    }
}

// Kernel 2, grid (ceil(n/32), B*heads), 256 threads: a half-wave per pixel, one lane per channel, eight pixels per wave (two at a time).
// Every product is a 32 x 32 matrix-vector one: the matrices sit in LDS (ctx and dctx transposed with a padded row, so the lanes of a
// pixel read consecutive banks), the pixel's three vectors (dout, v, k) are exchanged through a wave-private LDS row and read back as
// broadcasts.  One THREAD per pixel and head (the first form of this kernel) did the 3072 multiply-adds and 512 LDS reads of a pixel
// serially: 33 us per launch whatever n was, a quarter of a millisecond per training step.
__global__ void __launch_bounds__(256) linattn_bwd_apply_kernel(const float* qkv, const float* dout, const float* ctx, const float* dctx,
                                                                const float* kst, const float* rr, float* dqkv, int n, int heads) {
    __shared__ float ct[DH][DH + 1];     // ct[e][d]  = ctx[d][e]
    __shared__ float dct[DH][DH + 1];    // dct[e][d] = dctx[d][e]
    __shared__ float dc[DH][DH + 1];     // dc[d][e]  = dctx[d][e]
    __shared__ float vec[4][2][3][DH];
    const int bh = blockIdx.y, b = bh / heads, h = bh % heads, tid = threadIdx.x;
    const int C3 = 3 * heads * DH, CO = heads * DH;
    for (int i = tid; i < DH * DH; i += 256) {
        const int d = i >> 5, e = i & 31;
        const float c = ctx[(size_t)bh * DH * DH + i], g = dctx[(size_t)bh * DH * DH + i];
        ct[e][d] = c; dct[e][d] = g; dc[d][e] = g;
    }
    const int lane = tid & 63, half = lane >> 5, c = lane & 31, w = tid >> 6;
    const float kmax = kst[((size_t)bh * DH + c) * 2], kinv = kst[((size_t)bh * DH + c) * 2 + 1], rrc = rr[(size_t)bh * DH + c];
    const float scale = 0.17677669529663687f;
    __syncthreads();
    for (int it = 0; it < 4; ++it) {
        const int pix = blockIdx.x * 32 + w * 8 + it * 2 + half;
        const bool in = pix < n;
        const size_t row = (size_t)b * n + (in ? pix : 0);
        const float* base = qkv + row * C3 + h * DH;
        const float qraw = base[c], kraw = base[heads * DH + c], v = base[2 * heads * DH + c], g = dout[row * CO + h * DH + c];
        float m = qraw;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float p = __expf(qraw - m), sum = p;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        p /= sum;
        const float k = __expf(kraw - kmax) * kinv;
        float* vr = &vec[w][half][0][0];
        vr[c] = g; vr[DH + c] = v; vr[2 * DH + c] = k;
        __syncthreads();
        float dq = 0.f, dk = 0.f, dvv = 0.f;
#pragma unroll
        for (int e = 0; e < DH; ++e) {
            dq += ct[e][c] * vr[e];              // sum_e ctx[c][e] dout[e]
            dk += dct[e][c] * vr[DH + e];        // sum_e dctx[c][e] v[e]
            dvv += dc[e][c] * vr[2 * DH + e];    // sum_d dctx[d][c] k[d]
        }
        const float dp = dq * scale;
        float dot = p * dp;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        if (in) {
            float* outp = dqkv + row * C3 + h * DH;
            outp[c] = p * (dp - dot);                      // dq_raw = p (s dq - sum_d p s dq)
            outp[heads * DH + c] = k * (dk - rrc);         // dk_raw = k (dk - rr)
            outp[2 * heads * DH + c] = dvv;
        }
        __syncthreads();
    }
}

int linattn_bwd_launch(const float* qkv, const float* dout, const float* ctx, float* dctx, float* kst, float* rr, float* dqkv, int B, int n,
                       int heads, hipStream_t s) {
    if (n <= 256) hipLaunchKernelGGL((linattn_bwd_ctx_kernel<4, 32>), dim3(B * heads), dim3(256), 0, s, qkv, dout, ctx, dctx, kst, rr, n, heads);
    else hipLaunchKernelGGL((linattn_bwd_ctx_kernel<8, 64>), dim3(B * heads), dim3(512), 0, s, qkv, dout, ctx, dctx, kst, rr, n, heads);
    FC_HIP(hipGetLastError());
    if (heads < 1 || heads > 4) return fail(FC_E_SHAPE, "linattn_bwd: at most 4 heads");
    hipLaunchKernelGGL(linattn_bwd_apply_kernel, dim3(cdiv(n, 32), B * heads), dim3(256), 0, s, qkv, dout, ctx, dctx, kst, rr, dqkv, n, heads);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// ===================================================================================================
// Attention backward (unet.py:110-121), n <= 64 tokens.   grid (B*heads), 64 threads (one per token).
__global__ void __launch_bounds__(64) attn_small_bwd_kernel(const float* qkv, const float* dout, float* dqkv, int n, int heads) {
    __shared__ float Q[64][DH + 1], K[64][DH + 1], V[64][DH + 1], DO[64][DH + 1];
    __shared__ float P[64][65], DS[64][65];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, i = threadIdx.x;
    const int C3 = 3 * heads * DH, CO = heads * DH;
    const float scale = 0.17677669529663687f;
    if (i < n) {
        const float* base = qkv + ((size_t)b * n + i) * C3 + h * DH;
        const float* dop = dout + ((size_t)b * n + i) * CO + h * DH;
        for (int d = 0; d < DH; ++d) { Q[i][d] = base[d]; K[i][d] = base[heads * DH + d]; V[i][d] = base[2 * heads * DH + d]; DO[i][d] = dop[d]; }
    }
    __syncthreads();
    if (i < n) {
        float m = -INFINITY;
        for (int j = 0; j < n; ++j) {
            float sdot = 0.f;
            for (int d = 0; d < DH; ++d) sdot += (Q[i][d] * scale) * K[j][d];
            P[i][j] = sdot;
            m = fmaxf(m, sdot);
        }
        float sum = 0.f;
        for (int j = 0; j < n; ++j) { const float e = __expf(P[i][j] - m); P[i][j] = e; sum += e; }
        const float inv = 1.0f / sum;
        float rd = 0.f;
        for (int j = 0; j < n; ++j) {
            const float pj = P[i][j] * inv;
            float dpj = 0.f;
            for (int d = 0; d < DH; ++d) dpj += DO[i][d] * V[j][d];
            P[i][j] = pj;
            DS[i][j] = dpj;
            rd += pj * dpj;
        }
        for (int j = 0; j < n; ++j) DS[i][j] = P[i][j] * (DS[i][j] - rd);
        float* outp = dqkv + ((size_t)b * n + i) * C3 + h * DH;
        for (int d = 0; d < DH; ++d) {
            float acc = 0.f;
            for (int j = 0; j < n; ++j) acc += DS[i][j] * K[j][d];
            outp[d] = acc * scale;
        }
    }
    __syncthreads();
    if (i < n) {
        float* outp = dqkv + ((size_t)b * n + i) * C3 + h * DH;
        for (int d = 0; d < DH; ++d) {
            float ak = 0.f, av = 0.f;
            for (int r = 0; r < n; ++r) { ak += DS[r][i] * Q[r][d]; av += P[r][i] * DO[r][d]; }
            outp[heads * DH + d] = ak * scale;
            outp[2 * heads * DH + d] = av;
        }
    }
}
int attn_small_bwd_launch(const float* qkv, const float* dout, float* dqkv, int B, int n, int heads, hipStream_t s) {
    if (n > 64) return fail(FC_E_SHAPE, "attn_small_bwd: more than 64 tokens");
    hipLaunchKernelGGL(attn_small_bwd_kernel, dim3(B * heads), dim3(64), 0, s, qkv, dout, dqkv, n, heads);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// ===================================================================================================
// Dense layers of the conditioning path (B <= a few hundred rows).  `xpre` is the layer input BEFORE its activation
// in_act (0 none, 1 GELU(erf), 2 SiLU); weights are the reference's [O][I].
__device__ __forceinline__ float act_apply(float x, int k) {
    if (k == 1) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    if (k == 2) return x / (1.0f + expf(-x));
    return x;
}
__device__ __forceinline__ float act_grad(float x, int k) {
    if (k == 1) return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
    if (k == 2) { const float s = 1.0f / (1.0f + expf(-x)); return s * (1.0f + x * (1.0f - s)); }
    return 1.0f;
}
__global__ void __launch_bounds__(256) dense_fwd_kernel(const float* xpre, int in_act, const float* w, const float* bias, float* y, int B, int I, int O) {
    const size_t total = (size_t)B * O;
    for (size_t t = blockIdx.x * 256ull + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int o = (int)(t % O), b = (int)(t / O);
        float acc = bias ? bias[o] : 0.f;
        for (int i = 0; i < I; ++i) acc += act_apply(xpre[(size_t)b * I + i], in_act) * w[(size_t)o * I + i];
        y[t] = acc;
    }
}
int dense_fwd_launch(const float* xpre, int in_act, const float* w, const float* bias, float* y, int B, int I, int O, hipStream_t s) {
    hipLaunchKernelGGL(dense_fwd_kernel, dim3(grid_1d((size_t)B * O)), dim3(256), 0, s, xpre, in_act, w, bias, y, B, I, O);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
// dW[o][i] = sum_b dy[b][o] act(xpre[b][i]);  db[o] = sum_b dy[b][o]      (dy row stride ldy)
__global__ void __launch_bounds__(256) dense_bwd_w_kernel(const float* dy, int ldy, const float* xpre, int in_act, float* dw, float* db, int B, int I, int O) {
    const size_t total = (size_t)O * I;
    for (size_t t = blockIdx.x * 256ull + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int i = (int)(t % I), o = (int)(t / I);
        float acc = 0.f, bs = 0.f;
        for (int b = 0; b < B; ++b) {
            const float d = dy[(size_t)b * ldy + o];
            acc += d * act_apply(xpre[(size_t)b * I + i], in_act);
            bs += d;
        }
        dw[t] = acc;
        if (i == 0 && db) db[o] = bs;
    }
}
int dense_bwd_w_launch(const float* dy, int ldy, const float* xpre, int in_act, float* dw, float* db, int B, int I, int O, hipStream_t s) {
    hipLaunchKernelGGL(dense_bwd_w_kernel, dim3(grid_1d((size_t)O * I)), dim3(256), 0, s, dy, ldy, xpre, in_act, dw, db, B, I, O);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
// Every ResnetBlock.mlp weight / bias gradient in one launch: job j owns rows [col, col + O) of the concatenated FiLM gradient
// dss [B][S]; x = act(xpre) is shared.   blocks[k] = (job, chunk of 256 (o, i) pairs)
__global__ void __launch_bounds__(256) dense_bwd_w_table_kernel(const DenseWJob* jobs, const int2* blocks, const float* dss, int S, const float* xpre,
                                                                int in_act, float* grads, int B, int I) {
    extern __shared__ float xs[];   // act(xpre) [B][I] when it fits, else read through
    const int2 bj = blocks[blockIdx.x];
    const DenseWJob j = jobs[bj.x];
    const size_t t = (size_t)bj.y * 256 + threadIdx.x;
    if (t >= (size_t)j.O * I) return;
    const int i = (int)(t % I), o = (int)(t / I);
    float acc = 0.f, bs = 0.f;
    for (int b = 0; b < B; ++b) {
        const float d = dss[(size_t)b * S + j.col + o];
        acc += d * act_apply(xpre[(size_t)b * I + i], in_act);
        bs += d;
    }
    grads[j.dw + t] = acc;
    if (i == 0) grads[j.db + o] = bs;
}
int dense_bwd_w_table_launch(const DenseWJob* jobs_dev, const int2* blocks_dev, int nblocks, const float* dss, int S, const float* xpre, int in_act,
                             float* grads, int B, int I, hipStream_t s) {
    if (!nblocks) return FC_OK;
    hipLaunchKernelGGL(dense_bwd_w_table_kernel, dim3(nblocks), dim3(256), 0, s, jobs_dev, blocks_dev, dss, S, xpre, in_act, grads, B, I);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// dxpre[b][i] = act'(xpre[b][i]) * sum_o dy[b][o] W[o][i]     (w_t: weights stored [I][ldw] instead of [O][I])
__global__ void __launch_bounds__(256) dense_bwd_x_kernel(const float* dy, int ldy, const float* w, int w_t, int ldw, const float* xpre, int in_act,
                                                          float* dx, int accumulate, int B, int I, int O) {
    const size_t total = (size_t)B * I;
    for (size_t t = blockIdx.x * 256ull + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int i = (int)(t % I), b = (int)(t / I);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const float* dyb = dy + (size_t)b * ldy;
        int o = 0;
        if (w_t) {
            const float* wr = w + (size_t)i * ldw;
            for (; o + 4 <= O; o += 4) { a0 += dyb[o] * wr[o]; a1 += dyb[o + 1] * wr[o + 1]; a2 += dyb[o + 2] * wr[o + 2]; a3 += dyb[o + 3] * wr[o + 3]; }
            for (; o < O; ++o) a0 += dyb[o] * wr[o];
        } else {
            for (; o + 4 <= O; o += 4) {
                a0 += dyb[o] * w[(size_t)o * I + i]; a1 += dyb[o + 1] * w[(size_t)(o + 1) * I + i];
                a2 += dyb[o + 2] * w[(size_t)(o + 2) * I + i]; a3 += dyb[o + 3] * w[(size_t)(o + 3) * I + i];
            }
            for (; o < O; ++o) a0 += dyb[o] * w[(size_t)o * I + i];
        }
        float v = (a0 + a1) + (a2 + a3);
        if (xpre) v *= act_grad(xpre[t], in_act);
        dx[t] = accumulate ? dx[t] + v : v;
    }
}
// The same product with one WAVE per output element: the lanes split the reduction over O (lane-strided, coalesced for the transposed
// weight layout) and meet in a fixed butterfly.  For the FiLM gradient of the whole network (O = every block's scale/shift column,
// thousands) one thread per output was a serial loop of O steps -- 80 us for 2048 outputs.
__global__ void __launch_bounds__(256) dense_bwd_x_wave_kernel(const float* dy, int ldy, const float* w, int w_t, int ldw, const float* xpre, int in_act,
                                                               float* dx, int accumulate, int B, int I, int O) {
    const int lane = threadIdx.x & 63;
    const size_t t = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= (size_t)B * I) return;
    const int i = (int)(t % I), b = (int)(t / I);
    const float* dyb = dy + (size_t)b * ldy;
    float acc = 0.f;
    if (w_t) {
        const float* wr = w + (size_t)i * ldw;
        for (int o = lane; o < O; o += 64) acc += dyb[o] * wr[o];
    } else {
        for (int o = lane; o < O; o += 64) acc += dyb[o] * w[(size_t)o * I + i];
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) {
        float v = acc;
        if (xpre) v *= act_grad(xpre[t], in_act);
        dx[t] = accumulate ? dx[t] + v : v;
    }
}
int dense_bwd_x_launch(const float* dy, int ldy, const float* w, int w_t, int ldw, const float* xpre, int in_act, float* dx, int accumulate,
                       int B, int I, int O, hipStream_t s) {
    if (O >= 256) {
        hipLaunchKernelGGL(dense_bwd_x_wave_kernel, dim3((unsigned)(((size_t)B * I + 3) / 4)), dim3(256), 0, s, dy, ldy, w, w_t, ldw, xpre, in_act, dx,
                           accumulate, B, I, O);
        FC_HIP(hipGetLastError());
        return FC_OK;
    }
    hipLaunchKernelGGL(dense_bwd_x_kernel, dim3(grid_1d((size_t)B * I)), dim3(256), 0, s, dy, ldy, w, w_t, ldw, xpre, in_act, dx, accumulate, B, I, O);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// sinusoidal embedding rows (unet.py:24-29) and class-embedding gather / scatter
__global__ void __launch_bounds__(256) sin_emb_kernel(const float* time, const float* freqs, float* e, int B, int dim) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B * dim) return;
    const int i = t % dim, b = t / dim, half = dim / 2;
    const float arg = time[b] * freqs[i < half ? i : i - half];
    e[t] = i < half ? sinf(arg) : cosf(arg);
}
int sin_emb_launch(const float* time, const float* freqs, float* e, int B, int dim, hipStream_t s) {
    hipLaunchKernelGGL(sin_emb_kernel, dim3(cdiv(B * dim, 256)), dim3(256), 0, s, time, freqs, e, B, dim);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
// An id outside [0, R) contributes nothing, exactly as the forward treats it (temb.hip class_of): no out-of-bounds read, forward and
// backward agree.  (The host mirror raises IndexError for such ids before they get here, as nn.Embedding does upstream.)
__global__ void __launch_bounds__(256) gather_rows_kernel(const float* table, const int64_t* ids, float* out, int B, int D, int R) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B * D) return;
    const int64_t id = ids[t / D];
    out[t] = (id >= 0 && id < R) ? table[(size_t)id * D + t % D] : 0.f;
}
int gather_rows_launch(const float* table, const int64_t* ids, float* out, int B, int D, int R, hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(B * D, 256)), dim3(256), 0, s, table, ids, out, B, D, R);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
// out[b][:] = d[b][:] where ids[b] is a valid class, 0 elsewhere: rows without a class embedding (id < 0, fc_unet_forward) take no part
// in the class MLP's backward either
__global__ void __launch_bounds__(256) mask_rows_kernel(const float* d, int ld, const int64_t* ids, float* out, int B, int D, int R) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B * D) return;
    const int64_t id = ids[t / D];
    out[t] = (id >= 0 && id < R) ? d[(size_t)(t / D) * ld + t % D] : 0.f;
}
int mask_rows_launch(const float* d, int ld, const int64_t* ids, float* out, int B, int D, int R, hipStream_t s) {
    hipLaunchKernelGGL(mask_rows_kernel, dim3(cdiv(B * D, 256)), dim3(256), 0, s, d, ld, ids, out, B, D, R);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
// dtable[r][k] = sum over rows b with ids[b] == r of d[b][k], in batch order
__global__ void __launch_bounds__(256) scatter_rows_kernel(const float* d, const int64_t* ids, float* dtable, int B, int D, int R) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= R * D) return;
    const int k = t % D, r = t / D;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) if (ids[b] == r) acc += d[(size_t)b * D + k];
    dtable[t] = acc;
}
int scatter_rows_launch(const float* d, const int64_t* ids, float* dtable, int B, int D, int R, hipStream_t s) {
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv(R * D, 256)), dim3(256), 0, s, d, ids, dtable, B, D, R);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// ===================================================================================================
// Elementwise helpers
__global__ void __launch_bounds__(256) axpy_kernel(float* dst, const float* src, size_t n4) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 a = reinterpret_cast<float4*>(dst)[i];
        const float4 b = reinterpret_cast<const float4*>(src)[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        reinterpret_cast<float4*>(dst)[i] = a;
    }
}
int add_into_launch(float* dst, const float* src, size_t n, hipStream_t s) {
    if (n & 3) return fail(FC_E_SHAPE, "add_into: length must be a multiple of 4");
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_1d(n / 4)), dim3(256), 0, s, dst, src, n / 4);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
// y = SiLU(z) (+ add);  dz = dy * SiLU'(z)      (the mask-conditioning convolutions keep their pre-activations when training)
__global__ void __launch_bounds__(256) silu_fwd_kernel(const float* z, const float* add, float* y, size_t n4) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(z)[i];
        v.x = silu_e(v.x); v.y = silu_e(v.y); v.z = silu_e(v.z); v.w = silu_e(v.w);
        if (add) { const float4 a = reinterpret_cast<const float4*>(add)[i]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
        reinterpret_cast<float4*>(y)[i] = v;
    }
}
int silu_fwd_launch(const float* z, const float* add, float* y, size_t n, hipStream_t s) {
    if (n & 3) return fail(FC_E_SHAPE, "silu_fwd: length must be a multiple of 4");
    hipLaunchKernelGGL(silu_fwd_kernel, dim3(grid_1d(n / 4)), dim3(256), 0, s, z, add, y, n / 4);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
__global__ void __launch_bounds__(256) silu_bwd_kernel(const float* dy, const float* z, float* dz, size_t n4) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 g = reinterpret_cast<const float4*>(dy)[i], v = reinterpret_cast<const float4*>(z)[i];
        reinterpret_cast<float4*>(dz)[i] = make_float4(g.x * silu_grad_e(v.x), g.y * silu_grad_e(v.y), g.z * silu_grad_e(v.z), g.w * silu_grad_e(v.w));
    }
}
int silu_bwd_launch(const float* dy, const float* z, float* dz, size_t n, hipStream_t s) {
    if (n & 3) return fail(FC_E_SHAPE, "silu_bwd: length must be a multiple of 4");
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(grid_1d(n / 4)), dim3(256), 0, s, dy, z, dz, n / 4);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
// Adjoint of F.interpolate(mode='bilinear', align_corners=False) on NHWC: gsrc[b][Y][X][c] += sum over destination pixels of
// weight(dst -> src) * gdst.  One thread per source element walking every destination pixel (tiny tensors; fixed order).
__global__ void __launch_bounds__(256) bilinear_bwd_kernel(const float* gdst, float* gsrc, int B, int C, int Hs, int Ws, int Hd, int Wd) {
    const size_t total = (size_t)B * Hs * Ws * C;
    const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int X = (int)(r % Ws); r /= Ws;
        const int Y = (int)(r % Hs), b = (int)(r / Hs);
        float acc = 0.f;
        for (int y = 0; y < Hd; ++y) {
            float fy = ((float)y + 0.5f) * sy - 0.5f;
            fy = fy < 0.f ? 0.f : fy;
            const int y0 = (int)fy, y1 = y0 + (y0 < Hs - 1 ? 1 : 0);
            const float ly = fy - (float)y0;
            const float wy = (y0 == Y ? 1.f - ly : 0.f) + (y1 == Y ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int x = 0; x < Wd; ++x) {
                float fx = ((float)x + 0.5f) * sx - 0.5f;
                fx = fx < 0.f ? 0.f : fx;
                const int x0 = (int)fx, x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
                const float lx = fx - (float)x0;
                const float wx = (x0 == X ? 1.f - lx : 0.f) + (x1 == X ? lx : 0.f);
                if (wx != 0.f) acc += wy * wx * gdst[(((size_t)b * Hd + y) * Wd + x) * C + c];
            }
        }
        gsrc[i] += acc;
    }
}
int bilinear_bwd_launch(const float* gdst, float* gsrc, int B, int C, int Hs, int Ws, int Hd, int Wd, hipStream_t s) {
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(grid_1d((size_t)B * Hs * Ws * C)), dim3(256), 0, s, gdst, gsrc, B, C, Hs, Ws, Hd, Wd);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// nearest x2 backward: dst[b][y][x][c] (+)= sum of the 2x2 block of src [B][2H][2W][C]
__global__ void __launch_bounds__(256) sumpool2_kernel(const float* src, float* dst, int B, int H, int W, int C, int accumulate) {
    const size_t total = (size_t)B * H * W * C;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H), b = (int)(r / H);
        const float* p = src + (((size_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + c;
        const float v = (p[0] + p[C]) + (p[(size_t)2 * W * C] + p[(size_t)2 * W * C + C]);
        dst[i] = accumulate ? dst[i] + v : v;
    }
}
int sumpool2_nhwc_launch(const float* src, float* dst, int B, int H, int W, int C, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(sumpool2_kernel, dim3(grid_1d((size_t)B * H * W * C)), dim3(256), 0, s, src, dst, B, H, W, C, accumulate);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
// space-to-depth backward: src [B][H][W][4C] (channel c*4 + i*2 + j) -> dst [B][2H][2W][C] (+)=
__global__ void __launch_bounds__(256) depth_to_space_kernel(const float* src, float* dst, int B, int H, int W, int C, int accumulate) {
    const size_t total = (size_t)B * 4 * H * W * C;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int X = (int)(r % (2 * W)); r /= 2 * W;
        const int Y = (int)(r % (2 * H)), b = (int)(r / (2 * H));
        const float v = src[(((size_t)b * H + (Y >> 1)) * W + (X >> 1)) * 4 * C + c * 4 + (Y & 1) * 2 + (X & 1)];
        dst[i] = accumulate ? dst[i] + v : v;
    }
}
int depth_to_space_launch(const float* src, float* dst, int B, int H, int W, int C, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(depth_to_space_kernel, dim3(grid_1d((size_t)B * 4 * H * W * C)), dim3(256), 0, s, src, dst, B, H, W, C, accumulate);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// ===================================================================================================
// Flow-matching step pieces (train_flow.py:350-358,392-397)
// x = (1 - t) s + t g ; v* = g - s      with t per sample
__global__ void __launch_bounds__(256) flow_interp_kernel(const float* src, const float* tgt, const float* t, float* x, float* v, size_t total, int per) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float tt = t[i / per], s = src[i], g = tgt[i];
        x[i] = (1.0f - tt) * s + tt * g;
        v[i] = g - s;
    }
}
int flow_interp_launch(const float* src, const float* tgt, const float* t, float* x, float* v, int B, int per, hipStream_t s) {
    const size_t total = (size_t)B * per;
    hipLaunchKernelGGL(flow_interp_kernel, dim3(grid_1d(total)), dim3(256), 0, s, src, tgt, t, x, v, total, per);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// The whole torch prologue of a training step in one launch (train_flow.py:346-357): t = warp_time(u (1 - eps) + eps) with the very
// operations torch runs (sampling.py:23-33: 4(1-s) t^3 + 6(s-1) t^2 + (3-2s) t, each product and sum rounded once), the U-Net's time
// input t * t_scale, x = (1 - t) s + t g and v* = g - s with the OT pairing's gather of the target rows folded in, and the range
// check of the class ids (nn.Embedding raises for an id outside the table: here a sticky device flag the host reads when it next
// synchronises).   grid (blocks over a row, B)
__global__ void __launch_bounds__(256) flow_prepare_kernel(const float* src, const float* tgt, const long long* perm, const float* u, float one_minus_eps,
                                                           float eps, float a3, float a2, float a1, float t_scale, const long long* ids, int n_classes,
                                                           float* t_out, float* time_out, float* x, float* v, int* flag, int per) {
    const int b = blockIdx.y;
    const float t0 = __fadd_rn(__fmul_rn(u[b], one_minus_eps), eps);
    const float t2 = __fmul_rn(t0, t0), t3 = __fmul_rn(t2, t0);
    const float tt = __fadd_rn(__fadd_rn(__fmul_rn(a3, t3), __fmul_rn(a2, t2)), __fmul_rn(a1, t0));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        t_out[b] = tt;
        time_out[b] = __fmul_rn(tt, t_scale);
        if (ids && flag) { const long long id = ids[b]; if (id < 0 || id >= n_classes) atomicOr(flag, 1); }
    }
    // a pairing entry outside [0, B) would be an out-of-bounds read (target[ot_indices] raises IndexError in torch): the row falls back to
    // its own target and the sticky flag's bit 1 tells the host
    long long pb = perm ? perm[b] : b;
    if (pb < 0 || pb >= (long long)gridDim.y) {
        if (flag && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(flag, 2);
        pb = b;
    }
    const float* s = src + (size_t)b * per;
    const float* g = tgt + (size_t)pb * per;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < per; i += gridDim.x * 256) {
        const float sv = s[i], gv = g[i];
        x[(size_t)b * per + i] = (1.0f - tt) * sv + tt * gv;
        v[(size_t)b * per + i] = gv - sv;
    }
}
int flow_prepare_launch(const float* src, const float* tgt, const int64_t* perm, const float* u, float t_eps, float warp_s, float t_scale,
                        const int64_t* ids, int n_classes, float* t_out, float* time_out, float* x, float* v, int* flag, int B, int per, hipStream_t s) {
    int gx = cdiv(per, 1024);
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    const double ws = warp_s;
    hipLaunchKernelGGL(flow_prepare_kernel, dim3(gx, B), dim3(256), 0, s, src, tgt, reinterpret_cast<const long long*>(perm), u, (float)(1.0 - (double)t_eps),
                       t_eps, (float)(4.0 * (1.0 - ws)), (float)(6.0 * (ws - 1.0)), (float)(3.0 - 2.0 * ws), t_scale,
                       reinterpret_cast<const long long*>(ids), n_classes, t_out, time_out, x, v, flag, per);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

constexpr int kRedBlocks = 256;
// loss = mean((v - v*)^2) ; dv = 2 (v - v*) / n.    ws: kRedBlocks floats
__global__ void __launch_bounds__(256) mse_partial_kernel(const float* v, const float* tgt, float* dv, float* ws, size_t n, float two_over_n) {
    __shared__ float red[4];
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = v[i] - tgt[i];
        acc += d * d;
        if (dv) dv[i] = d * two_over_n;
    }
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) mse_final_kernel(const float* ws, float* loss, float inv_n) {
    __shared__ float red[4];
    const float tot = block_sum(ws[threadIdx.x], red);
    if (threadIdx.x == 0) *loss = tot * inv_n;
}
int mse_loss_grad_launch(const float* v, const float* tgt, float* dv, float* loss, float* ws, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(mse_partial_kernel, dim3(kRedBlocks), dim3(256), 0, s, v, tgt, dv, ws, n, 2.0f / (float)n);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, s, ws, loss, 1.0f / (float)n);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// clip_grad_norm_: norm = sqrt(sum g^2) over the ranges given; coef = min(1, max_norm / (norm + 1e-6)).   out: {norm, coef}
__global__ void __launch_bounds__(256) sumsq_partial_kernel(const float* g, size_t n0, const float* g2, size_t n1, float* ws) {
    __shared__ float red[4];
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n0; i += (size_t)gridDim.x * 256) acc += g[i] * g[i];
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n1; i += (size_t)gridDim.x * 256) acc += g2[i] * g2[i];
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) clip_final_kernel(const float* ws, float* out, float max_norm) {
    __shared__ float red[4];
    const float tot = block_sum(ws[threadIdx.x], red);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(tot);
        out[0] = norm;
        out[1] = fminf(max_norm / (norm + 1e-6f), 1.0f);
    }
}
int grad_clip_coef_launch(const float* g, size_t n0, const float* g2, size_t n1, float max_norm, float* out2, float* ws, hipStream_t s) {
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(kRedBlocks), dim3(256), 0, s, g, n0, g2, n1, ws);
    hipLaunchKernelGGL(clip_final_kernel, dim3(1), dim3(256), 0, s, ws, out2, max_norm);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// Adam as torch.optim.Adam computes it (no weight decay, no amsgrad), on g*coef, followed by the EMA recurrence of train_flow.py:46-54.
__global__ void __launch_bounds__(256) adam_ema_kernel(float* p, const float* g, float* m, float* v, float* ema, size_t n, const float* coef_ptr,
                                                       float lr_over_bc1, float b1, float b2, float inv_sqrt_bc2, float eps, float decay,
                                                       float one_minus_decay, int do_adam, const int* skip) {
    if (skip && *skip) return;   // a step whose inputs were flagged invalid (class id / pairing out of range) updates nothing
    const float coef = coef_ptr ? *coef_ptr : 1.0f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float pv = p[i];
        if (do_adam) {
            const float gv = g[i] * coef;
            float mv = m[i], vv = v[i];
            mv = mv + (gv - mv) * (1.0f - b1);                 // lerp_
            vv = vv * b2 + (1.0f - b2) * gv * gv;              // mul_ + addcmul_
            const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
            pv = pv - lr_over_bc1 * (mv / denom);
            m[i] = mv; v[i] = vv; p[i] = pv;
        }
        if (ema) ema[i] = decay * ema[i] + one_minus_decay * pv;
    }
}
int adam_ema_launch(float* p, const float* g, float* m, float* v, float* ema, size_t n, const float* coef_dev, float lr, float b1, float b2,
                    float eps, int step, float ema_decay, int do_adam, hipStream_t s, const int* skip_flag_dev) {
    if (!n) return FC_OK;
    const double bc1 = 1.0 - std::pow((double)b1, step), bc2 = 1.0 - std::pow((double)b2, step);
    const float lr_over_bc1 = do_adam ? (float)((double)lr / bc1) : 0.f;
    const float inv_sqrt_bc2 = do_adam ? (float)(1.0 / std::sqrt(bc2)) : 0.f;
    hipLaunchKernelGGL(adam_ema_kernel, dim3(grid_1d(n, 2048)), dim3(256), 0, s, p, g, m, v, ema, n, coef_dev, lr_over_bc1, b1, b2, inv_sqrt_bc2,
                       eps, ema_decay, (float)(1.0 - (double)ema_decay), do_adam, skip_flag_dev);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc
