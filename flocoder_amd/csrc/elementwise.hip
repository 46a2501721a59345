// HBM-bound helper kernels: GroupNorm finalisation with residual, the boundary 1x1 convolutions that
// change layout (NCHW <-> NHWC), layout converters, bilinear resize, standalone GroupNorm statistics.
#include "common.h"
#include "stats_dev.h"

namespace fc {

// ---------------------------------------------------------------------------------------------------
// y = act(GroupNorm(h) [FiLM]) + res ; optional GroupNorm(1) partials of y.   grid (bps, B)
// Closes Block/ResnetBlock (unet.py:66-72,96) and Residual(PreNorm(LinearAttention)) (unet.py:39,133).
__device__ __forceinline__ void finalize_body(const FinalizeArgs& a, const int bps, const int bx, const int b) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* gm = sm;                 // [G][2]
    float* A = sm + 2 * a.xf.G;     // [C]
    float* Bv = A + a.C;            // [C]
    float* gr = Bv + a.C;           // [Gres][2], [C], [C]: the same for a normalised residual
    float* Ar = gr + 2 * (a.xf_res.mode ? a.xf_res.G : 0);
    float* Br = Ar + (a.xf_res.mode ? a.C : 0);
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int per = a.HW * a.C / bps;  // elements of this block (multiple of 4)
    const size_t base = (size_t)b * a.HW * a.C + (size_t)bx * per;
    const bool act = a.xf.mode == 2 && !a.act_after_add, act2 = a.act_after_add != 0, nres = a.xf_res.mode != 0;
    // Nothing below depends on an earlier load except through the two tables, so everything is requested up front: the first NPF
    // rounds of h / res and this thread's gamma / beta / FiLM entries travel together with the statistics -- one memory round trip
    // instead of three dependent ones (statistics -> affine operands -> data) for a kernel that lives 5 us.
    constexpr int NPF = 4;
    float4 hv[NPF], rv[NPF];
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
        const int i = 4 * tid + 1024 * k;
        hv[k] = rv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < per) {
            hv[k] = *reinterpret_cast<const float4*>(a.h + base + i);
            if (a.res) rv[k] = *reinterpret_cast<const float4*>(a.res + base + i);
        }
    }
    float pg = 1.f, pb = 0.f, psc = 0.f, psh = 0.f;
    if (tid < a.C) {
        pg = a.xf.gamma[tid];
        pb = a.xf.beta[tid];
        if (a.xf.ss) {
            psc = a.xf.ss[(size_t)b * a.xf.ss_stride + tid];
            psh = a.xf.ss[(size_t)b * a.xf.ss_stride + a.C + tid];
        }
    }
    for (int g = tid; g < a.xf.G; g += 256) combine_partials(a.xf, b, g, &gm[2 * g], &gm[2 * g + 1]);
    if (a.xf_res.mode)
        for (int g = tid; g < a.xf_res.G; g += 256) combine_partials(a.xf_res, b, g, &gr[2 * g], &gr[2 * g + 1]);
    __syncthreads();
    if (a.xf_res.mode) {
        const int cpr = a.C / a.xf_res.G;
        for (int c = tid; c < a.C; c += 256) {
            const float s = gr[2 * (c / cpr) + 1] * a.xf_res.gamma[c];
            Ar[c] = s;
            Br[c] = a.xf_res.beta[c] - gr[2 * (c / cpr)] * s;
        }
    }
    const int cpg = a.C / a.xf.G;
    for (int c = tid; c < a.C; c += 256) {
        const int g = c / cpg;
        const bool pre = c == tid;
        float s = gm[2 * g + 1] * (pre ? pg : a.xf.gamma[c]);
        float t = (pre ? pb : a.xf.beta[c]) - gm[2 * g] * s;
        if (a.xf.ss) {
            const float sc = (pre ? psc : a.xf.ss[(size_t)b * a.xf.ss_stride + c]) + 1.0f;
            const float sh = pre ? psh : a.xf.ss[(size_t)b * a.xf.ss_stride + a.C + c];
            s *= sc;
            t = t * sc + sh;
        }
        A[c] = s;
        Bv[c] = t;
    }
    __syncthreads();
    float s = 0.f, q = 0.f;
    auto finish = [&](int i, float4 v, float4 r) {
        const int c = (int)((bx * (size_t)per + i) % a.C);
        v.x = A[c] * v.x + Bv[c];
        v.y = A[c + 1] * v.y + Bv[c + 1];
        v.z = A[c + 2] * v.z + Bv[c + 2];
        v.w = A[c + 3] * v.w + Bv[c + 3];
        if (act) { v.x = silu_e(v.x); v.y = silu_e(v.y); v.z = silu_e(v.z); v.w = silu_e(v.w); }
        if (a.res) {
            if (nres) { r.x = Ar[c] * r.x + Br[c]; r.y = Ar[c + 1] * r.y + Br[c + 1]; r.z = Ar[c + 2] * r.z + Br[c + 2]; r.w = Ar[c + 3] * r.w + Br[c + 3]; }
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (act2) { v.x = silu_e(v.x); v.y = silu_e(v.y); v.z = silu_e(v.z); v.w = silu_e(v.w); }
        *reinterpret_cast<float4*>(a.y + base + i) = v;
        s += (v.x + v.y) + (v.z + v.w);
        q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    };
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
        const int i = 4 * tid + 1024 * k;
        if (i < per) finish(i, hv[k], rv[k]);
    }
    for (int i = 4 * tid + 1024 * NPF; i < per; i += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(a.h + base + i);
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.res) r = *reinterpret_cast<const float4*>(a.res + base + i);
        finish(i, v, r);
    }
    if (a.stats_out) {
        const float S = block_sum_lds(s, red);    // LDS-only barriers: the output stores above stay in flight
        const float Q = block_sum_lds(q, red);
        if (tid == 0) {
            const float mean = S / (float)per;
            float* d = a.stats_out + ((size_t)b * bps + bx) * 2;
            d[0] = mean;
            d[1] = Q - S * mean;
        }
    }
}
__global__ void __launch_bounds__(256) finalize_kernel(const FinalizeArgs a, int bps) { finalize_body(a, bps, blockIdx.x, blockIdx.y); }

// Many finalize passes in one launch (the training backward recomputes every normalised activation a weight gradient reads; none of
// them is on the data-gradient chain): block -> (job, bx + bps * sample)
__global__ void __launch_bounds__(256) finalize_table_kernel(const FinalizeArgs* __restrict__ jobs, const int* __restrict__ job_bps,
                                                             const int2* __restrict__ blocks) {
    const int2 bj = blocks[blockIdx.x];
    const FinalizeArgs a = jobs[bj.x];
    const int bps = job_bps[bj.x];
    finalize_body(a, bps, bj.y % bps, bj.y / bps);
}
size_t finalize_lds_bytes(const FinalizeArgs& a) {
    return (size_t)(2 * a.xf.G + 2 * a.C + (a.xf_res.mode ? 2 * a.xf_res.G + 2 * a.C : 0)) * sizeof(float);
}
int finalize_table_launch(const FinalizeArgs* jobs_dev, const int* bps_dev, const int2* blocks_dev, int nblocks, size_t lds_bytes, hipStream_t s) {
    if (!nblocks) return FC_OK;
    hipLaunchKernelGGL(finalize_table_kernel, dim3(nblocks), dim3(256), lds_bytes, s, jobs_dev, bps_dev, blocks_dev);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

int finalize_blocks_per_sample(int HW, int C) {
    int per = HW * C;  // elements per sample
    int bps = 1;
    while (bps < 64 && per / (bps * 2) >= 4096 && (per / (bps * 2)) % 4 == 0) bps *= 2;
    return bps;
}

int finalize_launch(const FinalizeArgs& a, hipStream_t s) {
    if (a.C & 3) return fail(FC_E_SHAPE, "finalize: C must be a multiple of 4");
    if (!a.xf.mode || !a.xf.stats) return fail(FC_E_ARG, "finalize: needs GroupNorm statistics");
    const int bps = finalize_blocks_per_sample(a.HW, a.C);
    const size_t lds = (size_t)(2 * a.xf.G + 2 * a.C + (a.xf_res.mode ? 2 * a.xf_res.G + 2 * a.C : 0)) * sizeof(float);
    hipLaunchKernelGGL(finalize_kernel, dim3(bps, a.B), dim3(256), lds, s, a, bps);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// ---------------------------------------------------------------------------------------------------
// Standalone GroupNorm statistics (T = 1) for tensors with fewer than 16 pixels per sample, where the
// conv epilogue cannot produce them.  grid (G, B)
__global__ void __launch_bounds__(256) gn_stats_kernel(const float* x, float* stats, int HW, int C, int G) {
    __shared__ float red[4];
    const int b = blockIdx.y, g = blockIdx.x, cpg = C / G, n = HW * cpg;
    const float* xb = x + (size_t)b * HW * C + g * cpg;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += xb[(size_t)(i / cpg) * C + i % cpg];
    const float mean = block_sum(s, red) / (float)n;
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const float d = xb[(size_t)(i / cpg) * C + i % cpg] - mean; q += d * d; }
    const float M2 = block_sum(q, red);
    if (threadIdx.x == 0) { stats[((size_t)b * G + g) * 2] = mean; stats[((size_t)b * G + g) * 2 + 1] = M2; }
}

// Fold the T equal-count partials of every (sample, group) into one: consumers combine partials in their prologue, one thread per
// group walking all T slots -- fine for the 8 tiles of a 32x32 latent, a 200 us serial walk per workgroup for the 512 tiles of a
// 256x256 image (measured: the SD-VAE's 256x256 layers ran at 31 TFLOP/s, the 64x64 ones at 91).   grid (G, B), one wave
__global__ void __launch_bounds__(64) gn_fold_kernel(const float* in, float* out, int G, int T, float n_t) {
    const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
    const float* sp = in + (size_t)(b * G + g) * T * 2;
    float sm = 0.f;
    for (int t = lane; t < T; t += 64) sm += sp[2 * t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
    const float mean = sm / (float)T;
    float m2 = 0.f, dv = 0.f;
    for (int t = lane; t < T; t += 64) { const float d = sp[2 * t] - mean; m2 += sp[2 * t + 1]; dv += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m2 += __shfl_xor(m2, o); dv += __shfl_xor(dv, o); }
    if (lane == 0) {
        out[(size_t)(b * G + g) * 2] = mean;
        out[(size_t)(b * G + g) * 2 + 1] = m2 + n_t * dv;
    }
}
int gn_fold_launch(const float* in, float* out, int B, int G, int T, float n_t, hipStream_t s) {
    hipLaunchKernelGGL(gn_fold_kernel, dim3(G, B), dim3(64), 0, s, in, out, G, T, n_t);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

int gn_stats_launch(const float* x, float* stats, int B, int HW, int C, int G, hipStream_t s) {
    hipLaunchKernelGGL(gn_stats_kernel, dim3(G, B), dim3(256), 0, s, x, stats, HW, C, G);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// One wave that does nothing for `cycles` shader clocks: offsets the second row-range chain against the first so that their
// load / MFMA / store phases interleave on a CU instead of marching in step (FLOCODER_AMD_CHAINS=2, unet.hip run_forward).
__global__ void __launch_bounds__(64) delay_kernel(long long cycles) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}
int delay_launch(long long cycles, hipStream_t s) {
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s, cycles);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// ---------------------------------------------------------------------------------------------------
// init_conv (unet.py:185-186,295): 1x1 conv reading the NCHW boundary tensor, writing NHWC.
// Blocks beyond `conv_blocks` fetch this evaluation's precomputed scale / shift rows into the plan's table (CondFetch, common.h): the
// first reader is two launches away, so the copy costs no launch of its own.
__global__ void __launch_bounds__(256) init_conv_kernel(const float* x, int bmod, const float* w, const float* bias, float* out,
                                                        int B, int Cin, int HW, int Cout, int conv_blocks, const CondFetch f) {
    if ((int)blockIdx.x >= conv_blocks) {
        const float4* src = reinterpret_cast<const float4*>(f.all) + (size_t)(*f.evalc) * f.n4;
        float4* dst = reinterpret_cast<float4*>(f.dst);
        const int nb = gridDim.x - conv_blocks;
        for (int i = (blockIdx.x - conv_blocks) * 256 + threadIdx.x; i < f.n4; i += nb * 256) dst[i] = src[i];
        return;
    }
    const int q = Cout / 4;
    const size_t total = (size_t)B * HW * q;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)conv_blocks * 256) {
        const int co = (int)(i % q) * 4;
        const size_t bp = i / q;
        const int pix = (int)(bp % HW), b = (int)(bp / HW);
        const float* xb = x + (size_t)(b % bmod) * Cin * HW + pix;
        float4 acc = bias ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ci = 0; ci < Cin; ++ci) {
            const float xv = xb[(size_t)ci * HW];
            const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)ci * Cout + co);
            acc.x += xv * wv.x; acc.y += xv * wv.y; acc.z += xv * wv.z; acc.w += xv * wv.w;
        }
        *reinterpret_cast<float4*>(out + bp * Cout + co) = acc;
    }
}

int init_conv_launch(const CondFetch& fetch, const float* x, int bmod, const float* w, const float* bias, float* out, int B, int Cin, int HW,
                     int Cout, hipStream_t s) {
    if (Cout & 3) return fail(FC_E_SHAPE, "init_conv: Cout must be a multiple of 4");
    const size_t total = (size_t)B * HW * (Cout / 4);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    const int extra = fetch.all ? (fetch.n4 + 255) / 256 < 512 ? (fetch.n4 + 255) / 256 : 512 : 0;
    hipLaunchKernelGGL(init_conv_kernel, dim3(grid + extra), dim3(256), 0, s, x, bmod, w, bias, out, B, Cin, HW, Cout, grid, fetch);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// final_conv (unet.py:286,372): 1x1 conv reading NHWC, writing the NCHW boundary tensor -- or, inside the legacy Euler integrator,
// applying y += v * dt (one rounding per operation, as sampling.py:47 does) and publishing the next interval's time.
__global__ void __launch_bounds__(256) final_conv_kernel(const float* x, const float* w, const float* bias, float* out, int B, int Cin,
                                                         int HW, int Cout, const EulerTail e) {
    extern __shared__ float wsm[];  // [Cin][Cout] + [Cout]
    for (int i = threadIdx.x; i < Cin * Cout; i += 256) wsm[i] = w[i];
    for (int i = threadIdx.x; i < Cout; i += 256) wsm[Cin * Cout + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const size_t total = (size_t)B * HW;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int pix = (int)(i % HW), b = (int)(i / HW);
        const float* xp = x + i * Cin;
        for (int co0 = 0; co0 < Cout; co0 += 4) {
            float acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = (co0 + j < Cout) ? wsm[Cin * Cout + co0 + j] : 0.f;
            for (int ci = 0; ci < Cin; ci += 4) {
                const float4 xv = *reinterpret_cast<const float4*>(xp + ci);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (co0 + j < Cout) acc[j] += xs[k] * wsm[(ci + k) * Cout + co0 + j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (co0 + j < Cout) {
                    const size_t o = ((size_t)b * Cout + co0 + j) * HW + pix;
                    if (e.y) e.y[o] = __fadd_rn(e.y[o], __fmul_rn(acc[j], e.dt));    // x + pred * dt
                    else out[o] = acc[j];
                }
        }
    }
    if (e.y && blockIdx.x == 0) {   // what ode_time_kernel does at the head of a step, for the step that follows
        const int i = *e.step;
        const float t = e.ts[i];
        __syncthreads();            // everyone has read the counter before it moves
        if (threadIdx.x == 0) {
            e.sc[0] = t;
            e.sc[1] = 0.f;
            *e.step = i + 1;
        }
        const float tv = __fmul_rn(t, e.t_scale);
        for (int r = threadIdx.x; r < e.rows; r += 256) e.tvec[r] = tv;
    }
    if (e.evalc && blockIdx.x == 0 && threadIdx.x == 0) *e.evalc += 1;   // this forward is done: the next init_conv fetches the next slice
}

// The usual shapes (Cout <= 4 latent channels, Cin = dim a compile-time constant): one pixel per thread with its whole input row and
// its state values requested up front.  In the generic kernel above the channel loop stays rolled with a wait per 16-byte load
// (run-time Cin), eight dependent cold round trips per pixel inside a sampler step.
template <int CIN>
__global__ void __launch_bounds__(256) final_conv_small_kernel(const float* x, const float* w, const float* bias, float* out, int B, int HW,
                                                               int Cout, const EulerTail e) {
    __shared__ float wsm[CIN * 4 + 4];
    const size_t total = (size_t)B * HW, i = blockIdx.x * 256ull + threadIdx.x;
    float4 xv[CIN / 4];
    float yv[4] = {0.f, 0.f, 0.f, 0.f};
    const int pix = (int)(i % HW), b = (int)(i / HW);
    if (i < total) {
#pragma unroll
        for (int k = 0; k < CIN / 4; ++k) xv[k] = *reinterpret_cast<const float4*>(x + i * CIN + 4 * k);
        if (e.y) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j < Cout) yv[j] = e.y[((size_t)b * Cout + j) * HW + pix];
        }
    }
    for (int k = threadIdx.x; k < CIN * 4; k += 256) wsm[k] = (k & 3) < Cout ? w[(k >> 2) * Cout + (k & 3)] : 0.f;   // [CIN][4]
    if (threadIdx.x < 4) wsm[CIN * 4 + threadIdx.x] = (bias && (int)threadIdx.x < Cout) ? bias[threadIdx.x] : 0.f;
    __syncthreads();
    if (i < total) {
        float acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = wsm[CIN * 4 + j];
#pragma unroll
        for (int k = 0; k < CIN / 4; ++k) {
            const float xs[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += xs[q] * wsm[(4 * k + q) * 4 + j];   // same order as the generic kernel: ci ascending
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < Cout) {
                const size_t o = ((size_t)b * Cout + j) * HW + pix;
                if (e.y) e.y[o] = __fadd_rn(yv[j], __fmul_rn(acc[j], e.dt));    // x + pred * dt
                else out[o] = acc[j];
            }
    }
    if (e.y && blockIdx.x == 0) {   // what ode_time_kernel does at the head of a step, for the step that follows
        const int s = *e.step;
        const float t = e.ts[s];
        __syncthreads();            // everyone has read the counter before it moves
        if (threadIdx.x == 0) {
            e.sc[0] = t;
            e.sc[1] = 0.f;
            *e.step = s + 1;
        }
        const float tv = __fmul_rn(t, e.t_scale);
        for (int r = threadIdx.x; r < e.rows; r += 256) e.tvec[r] = tv;
    }
    if (e.evalc && blockIdx.x == 0 && threadIdx.x == 0) *e.evalc += 1;   // this forward is done: the next init_conv fetches the next slice
}

int final_conv_launch(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int HW, int Cout, const EulerTail& tail,
                      hipStream_t s) {
    if (Cin & 3) return fail(FC_E_SHAPE, "final_conv: Cin must be a multiple of 4");
    const size_t total = (size_t)B * HW;
    if (Cout <= 4 && total < (1ull << 31) && (Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64)) {
        const dim3 g((unsigned)((total + 255) / 256));
        switch (Cin) {
            case 8: hipLaunchKernelGGL(final_conv_small_kernel<8>, g, dim3(256), 0, s, x, w, bias, out, B, HW, Cout, tail); break;
            case 16: hipLaunchKernelGGL(final_conv_small_kernel<16>, g, dim3(256), 0, s, x, w, bias, out, B, HW, Cout, tail); break;
            case 32: hipLaunchKernelGGL(final_conv_small_kernel<32>, g, dim3(256), 0, s, x, w, bias, out, B, HW, Cout, tail); break;
            default: hipLaunchKernelGGL(final_conv_small_kernel<64>, g, dim3(256), 0, s, x, w, bias, out, B, HW, Cout, tail); break;
        }
        FC_HIP(hipGetLastError());
        return FC_OK;
    }
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(final_conv_kernel, dim3(grid), dim3(256), (size_t)(Cin * Cout + Cout) * sizeof(float), s, x, w, bias, out, B,
                       Cin, HW, Cout, tail);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* src, float* dst, int B, int C, int HW, int Cpad, int bmod) {
    const size_t total = (size_t)B * HW * Cpad;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % Cpad);
        const size_t bp = i / Cpad;
        const int pix = (int)(bp % HW), b = (int)(bp / HW);
        dst[i] = c < C ? src[((size_t)(b % bmod) * C + c) * HW + pix] : 0.f;
    }
}
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const float* src, float* dst, int B, int C, int HW, int Cpad) {
    const size_t total = (size_t)B * C * HW;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int pix = (int)(i % HW);
        const size_t bc = i / HW;
        const int c = (int)(bc % C), b = (int)(bc / C);
        dst[i] = src[((size_t)b * HW + pix) * Cpad + c];
    }
}
static int grid_for(size_t total) { size_t g = (total + 255) / 256; return (int)(g < 8192 ? (g ? g : 1) : 8192); }

int nchw_to_nhwc_launch(const float* src, float* dst, int B, int C, int HW, int Cpad, int bmod, hipStream_t s) {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((size_t)B * HW * Cpad)), dim3(256), 0, s, src, dst, B, C, HW, Cpad,
                       bmod > 0 ? bmod : B);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
int nhwc_to_nchw_launch(const float* src, float* dst, int B, int C, int HW, int Cpad, hipStream_t s) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((size_t)B * C * HW)), dim3(256), 0, s, src, dst, B, C, HW, Cpad);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

__global__ void __launch_bounds__(256) pixel_shuffle2_kernel(const float* src, float* dst, int B, int H, int W, int C) {
    const size_t total = (size_t)B * 4 * H * W * C;          // one thread per destination element
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int X = (int)(r % (2 * W)); r /= 2 * W;
        const int Y = (int)(r % (2 * H)), b = (int)(r / (2 * H));
        dst[i] = src[(((size_t)b * H + (Y >> 1)) * W + (X >> 1)) * 4 * C + c * 4 + (Y & 1) * 2 + (X & 1)];
    }
}
int pixel_shuffle2_nhwc_launch(const float* src, float* dst, int B, int H, int W, int C, hipStream_t s) {
    hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3(grid_for((size_t)B * 4 * H * W * C)), dim3(256), 0, s, src, dst, B, H, W, C);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// F.interpolate(mode='bilinear', align_corners=False) on NHWC (unet.py:338,362)
__global__ void __launch_bounds__(256) bilinear_kernel(const float* src, float* dst, int B, int C, int Hs, int Ws, int Hd, int Wd) {
    const size_t total = (size_t)B * Hd * Wd * C;
    const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int x = (int)(r % Wd); r /= Wd;
        const int y = (int)(r % Hd), b = (int)(r / Hd);
        float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
        fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* sb = src + (size_t)b * Hs * Ws * C + c;
        const float v00 = sb[((size_t)y0 * Ws + x0) * C], v01 = sb[((size_t)y0 * Ws + x1) * C];
        const float v10 = sb[((size_t)y1 * Ws + x0) * C], v11 = sb[((size_t)y1 * Ws + x1) * C];
        dst[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
}
int bilinear_nhwc_launch(const float* src, float* dst, int B, int C, int Hs, int Ws, int Hd, int Wd, hipStream_t s) {
    hipLaunchKernelGGL(bilinear_kernel, dim3(grid_for((size_t)B * Hd * Wd * C)), dim3(256), 0, s, src, dst, B, C, Hs, Ws, Hd, Wd);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc
