// The WHOLE U-Net forward of one sample in ONE workgroup (round 4; BASELINE config 5's flow leg: dim 8, latents 4x8x8, mask-conditioned).
//
// Why: at that shape the ordinary plan is 115 launches of a few hundred multiply-adds each -- 842 us per evaluation for 9.4 MFLOP per
// sample (0.7 TFLOP/s), nothing but launch-to-launch latency, and neither several chains inside the captured step nor several trajectories
// in flight overlap at that size (measured, tools/bench_inpaint.py).  Every activation of a sample is a few KB there, so a workgroup keeps
// the sample's whole forward in its LDS and walks a PROGRAM of steps (unet.py:289-372 unrolled by the host: convolutions, GroupNorm +
// FiLM + SiLU, the linear / full attention modules, bilinear mask resizes), one workgroup barrier between steps instead of a launch
// boundary; weights stream from L2 (the same packed copies the ordinary kernels read), the FiLM rows from the conditioning table.  The
// arithmetic is plain fp32 FMA on the vector pipe: the layers are far too small for matrix tiles (8..64 channels, 64..1 pixels).
// Same results as the ordinary plan to summation order (tests/test_gpu_unet.py compares both with the oracle).
#include <cstdlib>
#include <string>

#include "common.h"
#include "unet_sample.h"

namespace fc {

namespace {
constexpr int NT = 256;
constexpr int WCH = 4096;                 // floats per weight chunk (two buffers)
constexpr int NO = 4;                     // elements per thread at most (the host checks C * H * W <= NO * NT for every tensor)

__device__ __forceinline__ float silu(float z) { return z / (1.0f + __expf(-z)); }

// sum of `v` over the workgroup; `red` = 8 floats of LDS.  Every thread gets the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// y = [act]( (x - mean_g) * rstd_g * gamma + beta  [* (scale + 1) + shift] ) [+ res]     GroupNorm over (channels of a group) x pixels
__device__ void op_norm(const SStep& s, float* L, const float* ss, float* red) {
    const int tid = threadIdx.x, C = s.C0, HW = s.Hi * s.Wi, G = s.G, cpg = C / G, n = HW * C;
    const float* x = L + s.in0;
    float* y = L + s.out;
    // this thread's affine parameters first: their round trip to L2 passes behind the two reductions below
    float pg[NO], pb[NO], psc[NO], psh[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int e = tid + j * NT, c = e % C;
        pg[j] = 1.f; pb[j] = 0.f; psc[j] = 0.f; psh[j] = 0.f;
        if (e < n) {
            pg[j] = s.gamma[c]; pb[j] = s.beta[c];
            if (s.ss_off >= 0) { psc[j] = ss[s.ss_off + c]; psh[j] = ss[s.ss_off + C + c]; }
        }
    }
    float mean[8], rstd[8];
    const float cnt = (float)(cpg * HW);
    for (int g = 0; g < G; ++g) {
        float a = 0.f;
        for (int e = tid; e < n; e += NT) { const int c = e % C; if (c / cpg == g) a += x[e]; }
        mean[g] = block_sum(a, red) / cnt;
    }
    for (int g = 0; g < G; ++g) {
        float a = 0.f;
        for (int e = tid; e < n; e += NT) { const int c = e % C; if (c / cpg == g) { const float d = x[e] - mean[g]; a += d * d; } }
        rstd[g] = 1.0f / sqrtf(block_sum(a, red) / cnt + s.eps);
    }
    const float* res = s.res >= 0 ? L + s.res : nullptr;
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int e = tid + j * NT;
        if (e >= n) continue;
        const int c = e % C, g = c / cpg;
        float m = mean[0], r = rstd[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) if (k < G && g == k) { m = mean[k]; r = rstd[k]; }
        float v = (x[e] - m) * r * pg[j] + pb[j];
        if (s.ss_off >= 0) v = v * (psc[j] + 1.0f) + psh[j];
        if (s.act) v = silu(v);
        if (res) v += res[e];
        y[e] = v;
    }
}

// Convolution over one or two (concatenated) NHWC sources in LDS.  The weights (packed [tap][Cin][Cout] in global memory, L2-resident) pass
// through LDS in chunks of whole rows: read straight from global inside the multiply-add loop every product waited for its own load (the
// first version: 1.18 ms per evaluation, slower than the 115 launches it replaced).  The next chunk travels global -> registers while the
// current one is multiplied out of LDS; only the taps some output pixel can reach are staged (at 1x1 resolution: the centre tap).
__device__ void op_conv(const SStep& s, float* L, float* wbuf, int2* chunks, int* nchunks_s) {
    const int tid = threadIdx.x, C0 = s.C0, C1 = s.C1, Cin = C0 + C1, Cout = s.Cout, KS = s.KS;
    const float* a0 = L + s.in0;
    const float* a1 = C1 ? L + s.in1 : nullptr;
    const int Hin = s.Hi << s.ups, Win = s.Wi << s.ups, total = s.Ho * s.Wo * Cout;
    if (tid == 0) {   // the chunk list: per reachable kernel row ky the contiguous weight rows of its reachable kx, cut into chunks of <= WCH floats
        const int RW = WCH / Cout;
        int kx_lo = KS, kx_hi = -1, n = 0;
        for (int kx = 0; kx < KS; ++kx) if ((s.Wo - 1) * s.stride - s.pad + kx >= 0 && -s.pad + kx < Win) { if (kx < kx_lo) kx_lo = kx; kx_hi = kx; }
        for (int ky = 0; ky < KS && kx_hi >= kx_lo; ++ky) {
            if (!((s.Ho - 1) * s.stride - s.pad + ky >= 0 && -s.pad + ky < Hin)) continue;
            const int r0 = (ky * KS + kx_lo) * Cin, r1 = (ky * KS + kx_hi + 1) * Cin;
            for (int r = r0; r < r1 && n < 64; r += RW) chunks[n++] = make_int2(r, r1 - r < RW ? r1 - r : RW);
        }
        *nchunks_s = n;
    }
    int co[NO], oy[NO], ox[NO];
    float acc[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int o = tid + j * NT;
        co[j] = 0; oy[j] = -1; ox[j] = 0; acc[j] = 0.f;
        if (o < total) {
            co[j] = o % Cout;
            const int pix = o / Cout;
            oy[j] = pix / s.Wo; ox[j] = pix - oy[j] * s.Wo;
            acc[j] = s.bias ? s.bias[co[j]] : 0.f;
        }
    }
    __syncthreads();
    const int nch = *nchunks_s;
    float4 pre[WCH / 4 / NT];
    auto fetch = [&](int k) {
        const int2 c = chunks[k];
        const float4* src = reinterpret_cast<const float4*>(s.w + (size_t)c.x * Cout);
        const int n4 = c.y * Cout / 4;
#pragma unroll
        for (int i = 0; i < WCH / 4 / NT; ++i) { const int e = tid + i * NT; pre[i] = e < n4 ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f); }
    };
    auto stash = [&](int k) {
        const int n4 = chunks[k].y * Cout / 4;
        float4* dst = reinterpret_cast<float4*>(wbuf + (k & 1) * WCH);
#pragma unroll
        for (int i = 0; i < WCH / 4 / NT; ++i) { const int e = tid + i * NT; if (e < n4) dst[e] = pre[i]; }
    };
    if (nch > 0) { fetch(0); stash(0); }
    __syncthreads();
    for (int k = 0; k < nch; ++k) {
        if (k + 1 < nch) fetch(k + 1);
        const int2 c = chunks[k];
        const float* wb = wbuf + (k & 1) * WCH;
        for (int row = c.x; row < c.x + c.y;) {           // segments of one tap: channels [ci0, ci0 + seg)
            const int tap = row / Cin, ci0 = row - tap * Cin, ky = tap / KS, kx = tap - ky * KS;
            const int seg = (Cin - ci0 < c.x + c.y - row) ? Cin - ci0 : c.x + c.y - row;
            const float* wr = wb + (size_t)(row - c.x) * Cout;
#pragma unroll
            for (int j = 0; j < NO; ++j) {
                if (oy[j] < 0) continue;
                const int iy = oy[j] * s.stride - s.pad + ky, ix = ox[j] * s.stride - s.pad + kx;
                if (iy < 0 || iy >= Hin || ix < 0 || ix >= Win) continue;
                const int sp = (iy >> s.ups) * s.Wi + (ix >> s.ups);
                const float* w = wr + co[j];
                float a = acc[j];
                int ci = ci0;
                const int e0 = ci0 + seg < C0 ? ci0 + seg : C0;     // part of the segment inside the first source
                const float* p0 = a0 + sp * C0;
                for (; ci < e0; ++ci) a += p0[ci] * w[(size_t)(ci - ci0) * Cout];
                if (ci < ci0 + seg) {
                    const float* p1 = a1 + sp * C1 - C0;
                    for (; ci < ci0 + seg; ++ci) a += p1[ci] * w[(size_t)(ci - ci0) * Cout];
                }
                acc[j] = a;
            }
            row += seg;
        }
        if (k + 1 < nch) stash(k + 1);
        __syncthreads();
    }
    const float* res = s.res >= 0 ? L + s.res : nullptr;
    float* y = L + s.out;
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int o = tid + j * NT;
        if (o < total) {
            float v = acc[j];
            if (s.act) v = silu(v);
            if (res) v += res[o];
            y[o] = v;
        }
    }
}

// F.interpolate(mode='bilinear', align_corners=False) of an NHWC tensor in LDS (elementwise.hip bilinear_kernel, same arithmetic)
__device__ void op_bilinear(const SStep& s, float* L) {
    const int C = s.C0, Hs = s.Hi, Ws = s.Wi, Hd = s.Ho, Wd = s.Wo, total = Hd * Wd * C;
    const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
    const float* src = L + s.in0;
    float* dst = L + s.out;
    for (int i = threadIdx.x; i < total; i += NT) {
        const int c = i % C, r = i / C, x = r % Wd, y = r / Wd;
        float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
        fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float v00 = src[(y0 * Ws + x0) * C + c], v01 = src[(y0 * Ws + x1) * C + c];
        const float v10 = src[(y1 * Ws + x0) * C + c], v11 = src[(y1 * Ws + x1) * C + c];
        dst[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
}

// GroupNorm(1) statistics of an [n][C] tensor in LDS (mean, rstd); every thread gets them
__device__ __forceinline__ void gn1_stats(const float* x, int n, float eps, float* red, float* mean, float* rstd) {
    float a = 0.f;
    for (int e = threadIdx.x; e < n; e += NT) a += x[e];
    const float m = block_sum(a, red) / (float)n;
    float q = 0.f;
    for (int e = threadIdx.x; e < n; e += NT) { const float d = x[e] - m; q += d * d; }
    *mean = m;
    *rstd = 1.0f / sqrtf(block_sum(q, red) / (float)n + eps);
}

// Residual(PreNorm(LinearAttention)) (unet.py:125-161) / Residual(PreNorm(Attention)) (unet.py:99-122) on an [n][C] tensor in LDS.
// scratch: xn [n][C] | y [n][C] | q, k, v [n][32] each | ctx [32][32] or sim [n][n] | o [n][32]
__device__ void op_attention(const SStep& s, float* L, float* red, float* wbuf, bool full) {
    const int tid = threadIdx.x, C = s.C0, n = s.Hi * s.Wi, nC = n * C;
    constexpr int DH = 32, HID = 128, C3 = 384;
    const float* x = L + s.in0;
    float* xn = L + s.scratch;
    float* y = xn + nC;
    float* q = y + nC;
    float* k = q + n * DH;
    float* v = k + n * DH;
    float* cx = v + n * DH;                              // ctx [32][32], or sim [n][n]
    float* o = cx + (full ? n * n : DH * DH);
    float mean, rstd;
    gn1_stats(x, nC, s.eps, red, &mean, &rstd);
    for (int e = tid; e < nC; e += NT) {
        const int c = e % C;
        xn[e] = (x[e] - mean) * rstd * s.gamma[c] + s.beta[c];
        y[e] = s.b2[c];                                   // to_out bias; the heads' shares are added below
    }
    __syncthreads();
    const float scale = 0.17677669529663687f;            // dim_head^-0.5
    // this head's weights through LDS (the host checks C <= 64): wq [C][96] = the head's q | k | v columns of to_qkv, wo [32][C] = its rows of to_out;
    // head h + 1's travel global -> registers while head h is computed
    float* wq = wbuf;
    float* wo = wbuf + 6144;
    constexpr int NWQ = 6144 / 4 / NT, NWO = 2048 / 4 / NT;      // float4's per thread
    float4 pq[NWQ], po[NWO];
    const int nq4 = C * 24, no4 = 8 * C;                  // float4's in wq (C rows x 96) and wo (32 rows x C)
    auto fetch = [&](int h) {
#pragma unroll
        for (int i = 0; i < NWQ; ++i) {
            const int e = tid + i * NT, c = e / 24, r = e - c * 24, which = r >> 3, d4 = r & 7;
            pq[i] = e < nq4 ? *reinterpret_cast<const float4*>(s.w + (size_t)c * C3 + which * HID + h * DH + 4 * d4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NWO; ++i) {
            const int e = tid + i * NT;
            po[i] = e < no4 ? *reinterpret_cast<const float4*>(s.w2 + (size_t)(h * DH) * C + 4 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < NWQ; ++i) { const int e = tid + i * NT; if (e < nq4) reinterpret_cast<float4*>(wq)[e] = pq[i]; }
#pragma unroll
        for (int i = 0; i < NWO; ++i) { const int e = tid + i * NT; if (e < no4) reinterpret_cast<float4*>(wo)[e] = po[i]; }
    };
    fetch(0);
    for (int h = 0; h < 4; ++h) {
        stash();                                          // (the previous head's last reader passed the barrier that closes its loop body)
        __syncthreads();
        if (h + 1 < 4) fetch(h + 1);
        // q | k | v of this head: [n][32] each
        for (int i = tid; i < 3 * n * DH; i += NT) {
            const int which = i / (n * DH), r = i - which * (n * DH), pix = r / DH, d = r - pix * DH;
            const float* w = wq + which * DH + d;
            const float* xp = xn + pix * C;
            float acc = 0.f;
#pragma unroll 4
            for (int c = 0; c < C; ++c) acc += xp[c] * w[c * 96];
            q[i] = acc;                                   // q, k, v are contiguous
        }
        __syncthreads();
        if (!full) {
            // k: softmax over the positions (per channel d); thread (d, part): 8 threads per column
            {
                const int d = tid >> 3, part = tid & 7;
                float m = -INFINITY;
                for (int p = part; p < n; p += 8) m = fmaxf(m, k[p * DH + d]);
                m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
                float sum = 0.f;
                for (int p = part; p < n; p += 8) { const float e = __expf(k[p * DH + d] - m); k[p * DH + d] = e; sum += e; }
                sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
                const float inv = 1.0f / sum;
                for (int p = part; p < n; p += 8) k[p * DH + d] *= inv;
            }
            // q: softmax over the head's 32 channels (per position), * scale; thread (pix, part): 4 threads per position
            for (int p0 = 0; p0 < n; p0 += 64) {
                const int pix = p0 + (tid >> 2), part = tid & 3;
                const bool on = pix < n;
                float m = -INFINITY;
                if (on) for (int d = part; d < DH; d += 4) m = fmaxf(m, q[pix * DH + d]);
                m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2));
                float sum = 0.f;
                if (on) for (int d = part; d < DH; d += 4) { const float e = __expf(q[pix * DH + d] - m); q[pix * DH + d] = e; sum += e; }
                sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
                const float f = scale / sum;
                if (on) for (int d = part; d < DH; d += 4) q[pix * DH + d] *= f;
            }
            __syncthreads();
            for (int i = tid; i < DH * DH; i += NT) {     // ctx[d][e] = sum_n k[n][d] v[n][e]
                const int d = i >> 5, e = i & 31;
                float acc = 0.f;
                for (int p = 0; p < n; ++p) acc += k[p * DH + d] * v[p * DH + e];
                cx[i] = acc;
            }
            __syncthreads();
            for (int i = tid; i < n * DH; i += NT) {      // o[n][e] = sum_d q[n][d] ctx[d][e]
                const int pix = i >> 5, e = i & 31;
                float acc = 0.f;
#pragma unroll 8
                for (int d = 0; d < DH; ++d) acc += q[pix * DH + d] * cx[d * DH + e];
                o[i] = acc;
            }
        } else {
            for (int i = tid; i < n * n; i += NT) {       // sim[i][j] = (q_i * scale) . k_j
                const int a = i / n, b = i - a * n;
                float acc = 0.f;
#pragma unroll 8
                for (int d = 0; d < DH; ++d) acc += (q[a * DH + d] * scale) * k[b * DH + d];
                cx[i] = acc;
            }
            __syncthreads();
            for (int a = tid; a < n; a += NT) {           // softmax over the keys, one thread per query row (n <= 64)
                float m = -INFINITY;
                for (int b = 0; b < n; ++b) m = fmaxf(m, cx[a * n + b]);
                float sum = 0.f;
                for (int b = 0; b < n; ++b) { const float e = __expf(cx[a * n + b] - m); cx[a * n + b] = e; sum += e; }
                const float inv = 1.0f / sum;
                for (int b = 0; b < n; ++b) cx[a * n + b] *= inv;
            }
            __syncthreads();
            for (int i = tid; i < n * DH; i += NT) {      // o[i][d] = sum_j attn[i][j] v[j][d]
                const int a = i >> 5, d = i & 31;
                float acc = 0.f;
                for (int b = 0; b < n; ++b) acc += cx[a * n + b] * v[b * DH + d];
                o[i] = acc;
            }
        }
        __syncthreads();
        for (int e = tid; e < nC; e += NT) {              // y[n][c] += sum_e o[n][e] Wout[32 h + e][c]
            const int pix = e / C, c = e - pix * C;
            const float* w = wo + c;
            float acc = 0.f;
#pragma unroll 8
            for (int d = 0; d < DH; ++d) acc += o[pix * DH + d] * w[d * C];
            y[e] += acc;
        }
        __syncthreads();
    }
    float* out = L + s.out;
    if (!full) {                                          // to_out.1: GroupNorm(1), then the residual
        gn1_stats(y, nC, s.eps, red, &mean, &rstd);
        for (int e = tid; e < nC; e += NT) { const int c = e % C; out[e] = ((y[e] - mean) * rstd * s.g2[c] + s.be2[c]) + x[e]; }
    } else {
        for (int e = tid; e < nC; e += NT) out[e] = y[e] + x[e];
    }
}

__global__ void __launch_bounds__(NT) unet_sample_kernel(const SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    __shared__ float red[8];
    __shared__ int last, nchunks;
    __shared__ int2 chunks[64];
    const int b = blockIdx.x, tid = threadIdx.x, HW = a.HW, ch = a.ch;
    float* wbuf = L + a.wbuf_off;                         // 2 x WCH floats: weights on their way to the multiply-add loops
    SStep* prog = reinterpret_cast<SStep*>(L + a.prog_off);     // the program itself: a step descriptor read from global memory was a cold round trip per step
    {
        const int nw = a.nsteps * (int)(sizeof(SStep) / 4);
        const unsigned* src = reinterpret_cast<const unsigned*>(a.prog);
        unsigned* dst = reinterpret_cast<unsigned*>(prog);
        for (int i = tid; i < nw; i += NT) dst[i] = src[i];
    }
    const int eval = a.evalc ? *a.evalc : 0;
    const float* ss = a.ss_all ? a.ss_all + ((size_t)eval * a.rows + b) * a.S : a.ss + (size_t)b * a.S;
    const bool has_mask = a.mask != nullptr;
    // the sample's inputs: NCHW in global memory -> NHWC in LDS
    {
        const float* xb = a.x + (size_t)(b % a.x_mod) * ch * HW;
        for (int i = tid; i < ch * HW; i += NT) { const int c = i / HW, p = i - c * HW; L[a.x_off + p * ch + c] = xb[i]; }
        if (has_mask) {
            const float* mb = a.mask + (size_t)(b % a.x_mod) * ch * HW;
            for (int i = tid; i < ch * HW; i += NT) { const int c = i / HW, p = i - c * HW; L[a.mask_off + p * ch + c] = mb[i]; }
        }
    }
    __syncthreads();
    for (int i = 0; i < a.nsteps; ++i) {
        const SStep& s = prog[i];
        if (a.stamps && b == 0 && tid == 0) {
            a.stamps[2 * i] = __builtin_amdgcn_s_memrealtime();
            a.stamps[2 * i + 1] = (unsigned long long)s.op | ((unsigned long long)s.KS << 4) | ((unsigned long long)s.Cout << 8) | ((unsigned long long)(s.C0 + s.C1) << 20) |
                                  ((unsigned long long)s.Hi << 32) | ((unsigned long long)s.guard << 40);
        }
        // guard: 0 always | 1 only with a mask | 2 only when mask_fusion_conv runs | 3 only without a mask | 4 mask but no fusion | 5 no fusion
        const int g = s.guard;
        const bool on = g == 0 || (g == 1 && has_mask) || (g == 2 && a.mask_fuse) || (g == 3 && !has_mask) || (g == 4 && has_mask && !a.mask_fuse) ||
                        (g == 5 && !a.mask_fuse);
        if (!on) continue;                                // (uniform over the workgroup)
        switch (s.op) {
            case S_CONV: op_conv(s, L, wbuf, chunks, &nchunks); break;
            case S_NORM: op_norm(s, L, ss, red); break;
            case S_BILINEAR: op_bilinear(s, L); break;
            case S_LINATTN: op_attention(s, L, red, wbuf, false); break;
            case S_ATTN: op_attention(s, L, red, wbuf, true); break;
            case S_COPY: for (int e = tid; e < s.Cout; e += NT) L[s.out + e] = L[s.in0 + e]; break;
        }
        __syncthreads();
    }
    if (a.stamps && b == 0 && tid == 0) { a.stamps[2 * a.nsteps] = __builtin_amdgcn_s_memrealtime(); a.stamps[2 * a.nsteps + 1] = 255; }
    // the velocity: NHWC in LDS -> NCHW in global memory, or the legacy Euler update y += v * dt (final_conv's tail, elementwise.hip)
    {
        const float* vsrc = L + a.v_off;
        const EulerTail& e = a.euler;
        for (int i = tid; i < ch * HW; i += NT) {
            const int c = i / HW, p = i - c * HW;
            const float val = vsrc[p * ch + c];
            const size_t o = (size_t)b * ch * HW + i;
            if (e.y) e.y[o] = __fadd_rn(e.y[o], __fmul_rn(val, e.dt));
            else a.out[o] = val;
        }
    }
    // the workgroup that finishes LAST moves the integrator's counters (every workgroup has read them by then)
    if (a.euler.evalc || a.euler.y) {
        __threadfence();
        __syncthreads();
        if (tid == 0) last = (atomicAdd(a.done, 1u) == gridDim.x - 1) ? 1 : 0;
        __syncthreads();
        if (last) {
            const EulerTail& e = a.euler;
            if (e.y) {
                const int st = *e.step;
                const float t = e.ts[st];
                const float tv = __fmul_rn(t, e.t_scale);
                for (int r = tid; r < e.rows; r += NT) e.tvec[r] = tv;
                if (tid == 0) { e.sc[0] = t; e.sc[1] = 0.f; *e.step = st + 1; }
            }
            if (tid == 0) {
                if (e.evalc) *e.evalc += 1;
                *a.done = 0u;
            }
        }
    }
}
}  // namespace

int unet_sample_init() {
    static bool done = false;
    if (done) return FC_OK;
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(unet_sample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));   // (the kernel's static LDS: ~0.6 KB)
    done = true;
    return FC_OK;
}

int unet_sample_launch(const SampleArgs& a, int B, size_t lds_bytes, hipStream_t s) {
    hipLaunchKernelGGL(unet_sample_kernel, dim3(B), dim3(NT), lds_bytes, s, a);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc
