// The WHOLE U-Net forward of one sample in ONE workgroup (round 4; BASELINE config 5's flow leg: dim 8, latents 4x8x8, mask-conditioned).
//
// Why: at that shape the ordinary plan is 115 launches of a few hundred multiply-adds each -- 842 us per evaluation for 9.4 MFLOP per
// sample (0.7 TFLOP/s), nothing but launch-to-launch latency, and neither several chains inside the captured step nor several trajectories
// in flight overlap at that size (measured, tools/bench_inpaint.py).  Every activation of a sample is a few KB there, so a workgroup keeps
// the sample's whole forward in its LDS and walks a PROGRAM of steps (unet.py:289-372 unrolled by the host: convolutions, GroupNorm +
// FiLM + SiLU, the linear / full attention modules, bilinear mask resizes), one workgroup barrier between steps instead of a launch
// boundary; weights stream from L2 (the same packed copies the ordinary kernels read) through LDS, the FiLM rows come from the conditioning
// table.  The arithmetic is plain fp32 FMA on the vector pipe: the layers are far too small for matrix tiles (8..64 channels, 64..1 pixels).
// Same results as the ordinary plan to summation order (tests/test_gpu_unet.py compares both with the goldens and the oracle).
//
// What decides its speed (profiles/r04_sample_kernel_stamps.txt; the history is in DESIGN.md section 7): four waves on a CU hide nothing, so a
// step costs its LDS round trips and barriers in full (~3 us before any arithmetic).  Hence (1) nothing waits for its own weights -- the
// first chunk of the NEXT weight-reading step is requested (global -> registers) before the current step computes, later chunks of a step
// travel while the previous one is multiplied; (2) every hot LDS access is a 16-byte one and operands that do not change inside a loop sit
// in registers (a convolution thread owns four output channels of a pixel, an attention lane four channels of every eighth position);
// (3) idle lanes are given a share of the reduction (the channel quads of a tap are split over neighbouring lanes and summed by shuffles);
// (4) GroupNorm lives in the convolution's epilogue and the heads of a small attention run one per wave, because each removed step or
// barrier is worth more than the arithmetic it carries; the reductions use shifts and lane masks, never an integer division in a loop
// (all extents are powers of two; the host checks).
#include <cstdlib>
#include <string>

#include "common.h"
#include "unet_sample.h"

namespace fc {

namespace {
constexpr int NT = SAMPLE_THREADS;
constexpr int WCH = 4096;                 // floats per weight chunk (two buffers)
constexpr int NO = 4;                     // elements per thread at most (the host checks C * H * W <= NO * NT for every tensor)

__device__ __forceinline__ float silu(float z) { return z / (1.0f + __expf(-z)); }

// The workgroup barrier of the step loop: it orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL access
// (s_waitcnt vmcnt(0)), i.e. for the weights requested a step ahead; every barrier up to the hand-over at the end of the kernel guards LDS
// data (activations, staged weights, reductions), and the registers a global load fills are waited for by the compiler where they are
// used.  (Measured: no change, 438.1 against 438.6 us per evaluation -- the 0.8 us a convolution spends "staging" is the descriptor reads
// and address arithmetic of stash + look-ahead + request, not a wait for memory.)
__device__ __forceinline__ void lds_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// sum of `v` over the workgroup; `red` = NT / 64 floats of LDS.  Every thread gets the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    lds_bar();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    lds_bar();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) s += red[w];
    return s;
}

// ---- weights on their way: global -> registers (requested early) -> LDS (when the step starts) ---------------------------------------
// The registers are NAMED members, not an array: as an array that lives across the step loop the compiler kept them in scratch memory and
// waited for every "prefetch" on the spot (ISA of the first version: flat_load, s_waitcnt vmcnt(0), scratch_store).  The weight pointers are
// read from the LDS copy of the program, so their address space is unknown to the compiler: re-tagged as global, the loads are
// global_load (counted on vmcnt alone, in order) instead of flat_load (which also holds up every LDS wait).
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f4v* gf4p;
__device__ __forceinline__ gf4p gptr(const float* p) { return (gf4p)(size_t)p; }
typedef const __attribute__((address_space(1))) float* gfp;
__device__ __forceinline__ gfp gsc(const float* p) { return (gfp)(size_t)p; }      // a scalar parameter vector in global memory
struct Pre { f4v r0, r1, r2, r3, r4, r5, r6, r7; };
constexpr int NCV = WCH / 4 / NT;         // float4's per thread of a convolution chunk
constexpr int NWQ = 6144 / 4 / NT, NWO = 2048 / 4 / NT;      // ... of an attention head's wq / wo
static_assert(NCV <= 4 && NWQ == 6 && NWO == 2, "the named registers below are laid out for 256 threads");
#define FC_LD4(dst, cond, ptr) do { f4v z_ = {0.f, 0.f, 0.f, 0.f}; dst = z_; if (cond) dst = *(ptr); } while (0)
__device__ __forceinline__ void issue_conv_chunk(const SStep& s, int k, Pre& r) {
    gf4p src = gptr(s.w + (size_t)s.crow[k] * s.Cout);
    const int n4 = s.cn[k] * s.Cout / 4, t = threadIdx.x;
    FC_LD4(r.r0, t < n4, src + t); FC_LD4(r.r1, t + NT < n4, src + t + NT); FC_LD4(r.r2, t + 2 * NT < n4, src + t + 2 * NT); FC_LD4(r.r3, t + 3 * NT < n4, src + t + 3 * NT);
}
__device__ __forceinline__ void stash_conv_chunk(const SStep& s, int k, const Pre& r, float* wbuf) {
    const int n4 = s.cn[k] * s.Cout / 4, t = threadIdx.x;
    f4v* dst = reinterpret_cast<f4v*>(wbuf + (k & 1) * WCH);
    if (t < n4) dst[t] = r.r0;
    if (t + NT < n4) dst[t + NT] = r.r1;
    if (t + 2 * NT < n4) dst[t + 2 * NT] = r.r2;
    if (t + 3 * NT < n4) dst[t + 3 * NT] = r.r3;
}
// one head of an attention module: wq [C][96] = the head's q | k | v columns of to_qkv ([C][384] in memory), wo [32][C] = its rows of to_out
__device__ __forceinline__ gf4p wq_src(const SStep& s, int h, int e) {
    const int c = e / 24, q = e - c * 24, which = q >> 3, d4 = q & 7;
    return gptr(s.w + (size_t)c * 384 + which * 128 + h * 32 + 4 * d4);
}
__device__ __forceinline__ void issue_head(const SStep& s, int h, Pre& r) {
    const int C = s.C0, nq4 = C * 24, no4 = 8 * C, t = threadIdx.x;
    FC_LD4(r.r0, t < nq4, wq_src(s, h, t)); FC_LD4(r.r1, t + NT < nq4, wq_src(s, h, t + NT)); FC_LD4(r.r2, t + 2 * NT < nq4, wq_src(s, h, t + 2 * NT));
    FC_LD4(r.r3, t + 3 * NT < nq4, wq_src(s, h, t + 3 * NT)); FC_LD4(r.r4, t + 4 * NT < nq4, wq_src(s, h, t + 4 * NT)); FC_LD4(r.r5, t + 5 * NT < nq4, wq_src(s, h, t + 5 * NT));
    gf4p wo = gptr(s.w2 + (size_t)(h * 32) * C);
    FC_LD4(r.r6, t < no4, wo + t); FC_LD4(r.r7, t + NT < no4, wo + t + NT);
}
__device__ __forceinline__ void stash_head(const SStep& s, const Pre& r, float* wbuf) {
    const int C = s.C0, nq4 = C * 24, no4 = 8 * C, t = threadIdx.x;
    f4v* wq = reinterpret_cast<f4v*>(wbuf);
    f4v* wo = reinterpret_cast<f4v*>(wbuf + 6144);
    if (t < nq4) wq[t] = r.r0;
    if (t + NT < nq4) wq[t + NT] = r.r1;
    if (t + 2 * NT < nq4) wq[t + 2 * NT] = r.r2;
    if (t + 3 * NT < nq4) wq[t + 3 * NT] = r.r3;
    if (t + 4 * NT < nq4) wq[t + 4 * NT] = r.r4;
    if (t + 5 * NT < nq4) wq[t + 5 * NT] = r.r5;
    if (t < no4) wo[t] = r.r6;
    if (t + NT < no4) wo[t + NT] = r.r7;
}
__device__ __forceinline__ bool reads_weights(const SStep& s) { return s.op == S_CONV || s.op == S_LINATTN || s.op == S_ATTN || s.op == S_ATTN1 || s.op == S_LINATTN_W; }
__device__ __forceinline__ void issue_wv(const SStep& s, Pre& r);
__device__ __forceinline__ void issue_allheads(const SStep& s, Pre& r);
__device__ __forceinline__ void stash_flat8(const Pre& r, int n4, float* wbuf);
__device__ __forceinline__ void issue_first(const SStep& s, Pre& r) { if (s.op == S_CONV) issue_conv_chunk(s, 0, r); else if (s.op == S_ATTN1) issue_wv(s, r); else if (s.op == S_LINATTN_W) issue_allheads(s, r); else issue_head(s, 0, r); }
__device__ __forceinline__ void stash_first(const SStep& s, const Pre& r, float* wbuf) { if (s.op == S_CONV) stash_conv_chunk(s, 0, r, wbuf); else if (s.op == S_ATTN1) stash_flat8(r, s.C0 * 32, wbuf); else if (s.op == S_LINATTN_W) stash_flat8(r, s.C0 * 128, wbuf); else stash_head(s, r, wbuf); }

// (RETIRED as a step of its own -- the host folds every GroupNorm into the convolution in front of it, conv_epilogue below -- and KEPT in the
// kernel on purpose: without this function and its case in the step switch the compiler's code for the rest is 3.6 % slower, 454 against
// 438 us per evaluation, two builds alternating on one box (round 4).  The same holds for mean and variance in ONE pairwise (Chan) reduction
// in conv_epilogue: one barrier instead of four, no faster.)
// y = [act]( (x - mean_g) * rstd_g * gamma + beta  [* (scale + 1) + shift] ) [+ res]     GroupNorm over (channels of a group) x pixels.
// A thread's elements e = tid + j NT are the SAME channel c = tid & (C - 1) of different pixels (C <= 64 divides NT), so they lie in one
// group: a group's sum is reduced over exactly those lane bits that do not select the group (the bits below log2(channels per group) and
// the bits from log2(C) up), then over the waves through LDS.
__device__ __forceinline__ void op_norm(const SStep& s, float* L, const float* ss, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, C = s.C0, n = s.Hi * s.Wi * C, lc = s.lc, lcpg = s.lcpg;
    const float* x = L + s.in0;
    float* y = L + s.out;
    const int c = tid & (C - 1), g = c >> lcpg;
    const float pg = gsc(s.gamma)[c], pb = gsc(s.beta)[c];
    float psc = 0.f, psh = 0.f;
    if (s.ss_off >= 0) { psc = gsc(ss)[s.ss_off + c]; psh = gsc(ss)[s.ss_off + C + c]; }
    float v[NO], a = 0.f;
#pragma unroll
    for (int j = 0; j < NO; ++j) { const int e = tid + j * NT; v[j] = e < n ? x[e] : 0.f; a += v[j]; }
    const float cnt = (float)((n >> lc) << lcpg);          // pixels x channels per group
    const int rmask = (((1 << lcpg) - 1) | ~((1 << lc) - 1)) & 63;     // the lane bits a group's sum runs over
    auto group_sum = [&](float t) {
#pragma unroll
        for (int b = 0; b < 6; ++b) if ((rmask >> b) & 1) t += __shfl_xor(t, 1 << b);
        lds_bar();
        if ((lane & rmask) == 0) red[(tid >> 6) * 8 + g] = t;       // one lane per (wave, group)
        lds_bar();
        float r = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) r += red[w * 8 + g];
        return r;
    };
    const float mean = group_sum(a) / cnt;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NO; ++j) { const int e = tid + j * NT; if (e < n) { const float d = v[j] - mean; q += d * d; } }
    const float rstd = 1.0f / sqrtf(group_sum(q) / cnt + s.eps);
    const float* res = s.res >= 0 ? L + s.res : nullptr;
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int e = tid + j * NT;
        if (e >= n) continue;
        float t = (v[j] - mean) * rstd * pg + pb;
        if (s.ss_off >= 0) t = t * (psc + 1.0f) + psh;
        if (s.act) t = silu(t);
        if (res) t += res[e];
        y[e] = t;
    }
}

// Convolution over one or two (concatenated) NHWC sources in LDS; weights packed [tap][Cin][Cout], staged through LDS in the host's chunk
// list (whole taps some output pixel can reach).  Chunk 0 is in wbuf[0] when this is called.
//
// The multiply-add phase: a thread owns FOUR output channels of one pixel (a quad: one 16-byte weight read per input channel, one 16-byte
// activation read per four input channels -> five LDS reads per sixteen multiply-adds, four independent accumulators), and (1 << lks)
// neighbouring lanes share a quad, each taking every (1 << lks)-th channel quad of a tap, their partial sums added by lane shuffles:
// the layers have 128..1024 outputs and 72..432 products per output, so without the split most of the workgroup idles behind a serial
// chain of LDS latencies (the first version: one output per thread pass, 8 reads per 4 multiply-adds, 7..16 us per convolution).
// The raw sums go to the result tensor; the epilogue (bias, GroupNorm + FiLM, SiLU, residual) re-reads them in the channel-major thread
// layout the group reduction wants.
__device__ __forceinline__ void conv_mac(const SStep& s, const SStep& sl, float* L, float* wbuf, const float* zl) {
    const int tid = threadIdx.x, C0 = s.C0, C1 = s.C1, Cin = C0 + C1, Cout = s.Cout, KS = s.KS;
    const int CQ0 = C0 >> 2, lks = s.lks, KSPL = 1 << lks, lnq = s.lco - 2;
    const int ks = tid & (KSPL - 1), ot = tid >> lks, q = ot & ((1 << lnq) - 1), pix = ot >> lnq;
    const float* a0 = L + s.in0;
    const float* a1 = C1 ? L + s.in1 : zl;
    const int Hin = s.Hi << s.ups, Win = s.Wi << s.ups, npix = s.Ho * s.Wo;
    const bool live = pix < npix;
    const int oy = live ? pix / s.Wo : 0, ox = live ? pix - oy * s.Wo : 0;
    const int by = oy * s.stride - s.pad, bx = ox * s.stride - s.pad;
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    Pre cur;
    const int nch = s.nchunk;
    for (int k = 0; k < nch; ++k) {
        if (k + 1 < nch) issue_conv_chunk(sl, k + 1, cur);
        const float* wb = wbuf + (k & 1) * WCH + 4 * q;
        const int r0 = sl.crow[k], r1 = r0 + sl.cn[k];
        const int tap0 = r0 / Cin;                                         // (uniform; once per chunk)
        int ci0 = r0 - tap0 * Cin, ky = tap0 / KS, kx = tap0 - ky * KS;
        for (int row = r0; row < r1;) {                                    // segments of one tap: channels [ci0, ci0 + seg), whole quads
            const int seg = (Cin - ci0 < r1 - row) ? Cin - ci0 : r1 - row;
            const int iy = by + ky, ix = bx + kx;
            const bool ok = live && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
            const int sp = ok ? (iy >> s.ups) * s.Wi + (ix >> s.ups) : 0;
            const float* p0 = ok ? a0 + sp * C0 : zl;                      // (out of the picture: the zero line, no branch in the loop)
            const float* p1 = (ok && C1) ? a1 + sp * C1 - C0 : zl;
            const float* w = wb + (size_t)(row - r0 - ci0) * Cout;         // row (tap * Cin + c) of the weights sits at w + c * Cout
            const int cqe = (ci0 + seg) >> 2;
#pragma unroll 2
            for (int cq = (ci0 >> 2) + ks; cq < cqe; cq += KSPL) {
                const float* xp = (cq < CQ0 ? p0 : p1) + 4 * cq;
                const f4v x = *reinterpret_cast<const f4v*>(xp);
                const float* wr = w + (size_t)(4 * cq) * Cout;
                const f4v w0 = *reinterpret_cast<const f4v*>(wr), w1 = *reinterpret_cast<const f4v*>(wr + Cout);
                const f4v w2 = *reinterpret_cast<const f4v*>(wr + 2 * Cout), w3 = *reinterpret_cast<const f4v*>(wr + 3 * Cout);
                acc += x.x * w0; acc += x.y * w1; acc += x.z * w2; acc += x.w * w3;
            }
            row += seg; ci0 += seg;
            if (ci0 == Cin) { ci0 = 0; if (++kx == KS) { kx = 0; ++ky; } }
        }
        if (k + 1 < nch) stash_conv_chunk(sl, k + 1, cur, wbuf);
        lds_bar();
    }
    for (int b = 0; b < lks; ++b) {
        acc.x += __shfl_xor(acc.x, 1 << b); acc.y += __shfl_xor(acc.y, 1 << b); acc.z += __shfl_xor(acc.z, 1 << b); acc.w += __shfl_xor(acc.w, 1 << b);
    }
    if (live && ks == 0) *reinterpret_cast<f4v*>(L + s.out + pix * Cout + 4 * q) = acc;
    lds_bar();
}

// The same multiply-add phase for the common case -- ONE chunk holding NTAPS whole taps (all nine of a 3x3 kernel, or the single tap a 1x1
// kernel has / a 1x1 image can reach) and exactly NQI channel quads per lane per tap -- with both loops unrolled: the general loop above
// walks the taps one after the other, each a pointer set-up, five LDS reads and a wait (~0.3 us per tap, 2..3 us per convolution, most of
// the step); unrolled, the address arithmetic of all taps is independent straight-line code and their reads are in flight together.
template <int NTAPS, int NQI>
__device__ __forceinline__ void conv_mac_fast(const SStep& s, float* L, float* wbuf, const float* zl) {
    const int tid = threadIdx.x, C0 = s.C0, C1 = s.C1, Cin = C0 + C1, Cout = s.Cout, KS = s.KS;
    const int CQ0 = C0 >> 2, lks = s.lks, KSPL = 1 << lks, lnq = s.lco - 2;
    const int ks = tid & (KSPL - 1), ot = tid >> lks, q = ot & ((1 << lnq) - 1), pix = ot >> lnq;
    const float* a0 = L + s.in0;
    const float* a1 = C1 ? L + s.in1 : zl;
    const int Hin = s.Hi << s.ups, Win = s.Wi << s.ups, npix = s.Ho * s.Wo;
    const bool live = pix < npix;
    const int oy = live ? pix / s.Wo : 0, ox = live ? pix - oy * s.Wo : 0;
    const int by = oy * s.stride - s.pad, bx = ox * s.stride - s.pad;
    const int tap0 = s.crow[0] / Cin, ky0 = tap0 / KS, kx0 = tap0 - ky0 * KS;      // (NTAPS == 9: zero)
    const float* wb = wbuf + 4 * q;
    f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const int ky = NTAPS == 1 ? ky0 : t / 3, kx = NTAPS == 1 ? kx0 : t % 3;
        const int iy = by + ky, ix = bx + kx;
        const bool ok = live && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
        const int sp = ok ? (iy >> s.ups) * s.Wi + (ix >> s.ups) : 0;
        const float* p0 = ok ? a0 + sp * C0 : zl;
        const float* p1 = (ok && C1) ? a1 + sp * C1 - C0 : zl;
        const float* w = wb + (size_t)(t * Cin) * Cout;
#pragma unroll
        for (int j = 0; j < NQI; ++j) {
            const int cq = ks + j * KSPL;
            const float* xp = (cq < CQ0 ? p0 : p1) + 4 * cq;
            const f4v x = *reinterpret_cast<const f4v*>(xp);
            const float* wr = w + (size_t)(4 * cq) * Cout;
            const f4v w0 = *reinterpret_cast<const f4v*>(wr), w1 = *reinterpret_cast<const f4v*>(wr + Cout);
            const f4v w2 = *reinterpret_cast<const f4v*>(wr + 2 * Cout), w3 = *reinterpret_cast<const f4v*>(wr + 3 * Cout);
            acc += x.x * w0; acc += x.y * w1; acc += x.z * w2; acc += x.w * w3;
        }
    }
    lds_bar();                                       // (the general loop's end-of-chunk barrier: every wave is past its weight reads)
    for (int b = 0; b < lks; ++b) {
        acc.x += __shfl_xor(acc.x, 1 << b); acc.y += __shfl_xor(acc.y, 1 << b); acc.z += __shfl_xor(acc.z, 1 << b); acc.w += __shfl_xor(acc.w, 1 << b);
    }
    if (live && ks == 0) *reinterpret_cast<f4v*>(L + s.out + pix * Cout + 4 * q) = acc;
    lds_bar();
}
template <int NTAPS>
__device__ __forceinline__ void conv_mac_fast_n(const SStep& s, float* L, float* wbuf, const float* zl) {
    switch (s.nqi) {
        case 1: conv_mac_fast<NTAPS, 1>(s, L, wbuf, zl); break;
        case 2: conv_mac_fast<NTAPS, 2>(s, L, wbuf, zl); break;
        case 3: conv_mac_fast<NTAPS, 3>(s, L, wbuf, zl); break;
        default: conv_mac_fast<NTAPS, 4>(s, L, wbuf, zl); break;
    }
}

// The epilogue's per-channel parameters (global memory): requested BEFORE the multiply-add phase, used after it.
struct ConvParams { float bias, pg, pb, psc, psh; };
__device__ __forceinline__ ConvParams conv_params(const SStep& s, const float* ss) {
    const int co = threadIdx.x & (s.Cout - 1);
    ConvParams p = {0.f, 1.f, 0.f, 0.f, 0.f};
    if (s.bias) p.bias = gsc(s.bias)[co];
    if (s.fnorm) {
        p.pg = gsc(s.gamma)[co]; p.pb = gsc(s.beta)[co];
        if (s.ss_off >= 0) { p.psc = gsc(ss)[s.ss_off + co]; p.psh = gsc(ss)[s.ss_off + s.Cout + co]; }
    }
    return p;
}

// NJ = elements of the result per thread (channel co = tid & (Cout - 1) of NJ pixels, i.e. ONE GroupNorm group): a compile-time count.
template <int NJ>
__device__ __forceinline__ void conv_epilogue(const SStep& s, float* L, const ConvParams& cp, float* red) {
    const int tid = threadIdx.x, Cout = s.Cout, lco = s.lco, npix = s.Ho * s.Wo;
    const int co = tid & (Cout - 1), pstep = NT >> lco;
    const float bias = cp.bias;
    const float* res = s.res >= 0 ? L + s.res : nullptr;
    float* y = L + s.out;
    float acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int pix = (tid >> lco) + j * pstep; acc[j] = pix < npix ? y[pix * Cout + co] : 0.f; }
    if (s.fnorm) {
        // GroupNorm (+ FiLM) of the result right here: the group's sums run over the lane bits that do not select it (op_norm's reduction,
        // on registers), then over the waves.  A step less per Block half.
        const int lane = tid & 63, lcpg = s.lcpg, g = co >> lcpg;
        const float pg = cp.pg, pb = cp.pb, psc = cp.psc, psh = cp.psh;
        float v[NJ], a = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const bool on = (tid >> lco) + j * pstep < npix; v[j] = on ? acc[j] + bias : 0.f; a += v[j]; }
        const float cnt = (float)(npix << lcpg);
        const int rmask = (((1 << lcpg) - 1) | ~((1 << lco) - 1)) & 63;
        auto group_sum = [&](float t) {
#pragma unroll
            for (int b2 = 0; b2 < 6; ++b2) if ((rmask >> b2) & 1) t += __shfl_xor(t, 1 << b2);
            lds_bar();
            if ((lane & rmask) == 0) red[(tid >> 6) * 8 + g] = t;
            lds_bar();
            float r = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < NT / 64; ++w2) r += red[w2 * 8 + g];
            return r;
        };
        const float mean = group_sum(a) / cnt;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) if ((tid >> lco) + j * pstep < npix) { const float d = v[j] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(group_sum(q) / cnt + s.eps);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int pix = (tid >> lco) + j * pstep;
            if (pix < npix) {
                const int o = pix * Cout + co;
                float t = (v[j] - mean) * rstd * pg + pb;
                if (s.ss_off >= 0) t = t * (psc + 1.0f) + psh;
                if (s.act) t = silu(t);
                if (res) t += res[o];
                y[o] = t;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int pix = (tid >> lco) + j * pstep;
        if (pix < npix) {
            const int o = pix * Cout + co;
            float v = acc[j] + bias;
            if (s.act) v = silu(v);
            if (res) v += res[o];
            y[o] = v;
        }
    }
}

__device__ __forceinline__ void op_conv_epilogue(const SStep& s, float* L, const ConvParams& cp, float* red) {
    const int pstep = NT >> s.lco, nj = (s.Ho * s.Wo + pstep - 1) / pstep;        // uniform over the workgroup
    if (nj <= 1) conv_epilogue<1>(s, L, cp, red);
    else if (nj == 2) conv_epilogue<2>(s, L, cp, red);
    else conv_epilogue<4>(s, L, cp, red);
}

// F.interpolate(mode='bilinear', align_corners=False) of an NHWC tensor in LDS (elementwise.hip bilinear_kernel, same arithmetic)
__device__ __forceinline__ void op_bilinear(const SStep& s, float* L) {
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
    float v[NO], a = 0.f;
#pragma unroll
    for (int j = 0; j < NO; ++j) { const int e = threadIdx.x + j * NT; v[j] = e < n ? x[e] : 0.f; a += v[j]; }
    const float m = block_sum(a, red) / (float)n;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NO; ++j) { const int e = threadIdx.x + j * NT; if (e < n) { const float d = v[j] - m; q += d * d; } }
    *mean = m;
    *rstd = 1.0f / sqrtf(block_sum(q, red) / (float)n + eps);
}

// Residual(PreNorm(LinearAttention)) (unet.py:125-161) / Residual(PreNorm(Attention)) (unet.py:99-122) on an [n][C] tensor in LDS.
// scratch: xn [n][C] | y [n][C] | q, k, v [n][32] each | ctx [32][32] or sim [n][n] | o [n][32].  Head 0's weights are in wbuf when called.
__device__ __forceinline__ void op_attention(const SStep& s, float* L, float* red, float* wbuf, bool full) {
    const int tid = threadIdx.x, C = s.C0, lc = s.lc, n = s.Hi * s.Wi, ln = s.ln, nC = n * C;
    constexpr int DH = 32;
    const float* x = L + s.in0;
    float* xn = L + s.scratch;
    float* y = xn + nC;
    float* q = y + nC;
    float* k = q + n * DH;
    float* v = k + n * DH;
    float* cx = v + n * DH;                              // ctx [32][32], or sim [n][n]
    float* o = cx + (full ? n * n : DH * DH);
    const float* wq = wbuf;
    const float* wo = wbuf + 6144;
    const int cc = tid & (C - 1);
    const float pg = gsc(s.gamma)[cc], pb = gsc(s.beta)[cc], pbo = gsc(s.b2)[cc];
    float mean, rstd;
    gn1_stats(x, nC, s.eps, red, &mean, &rstd);
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int e = tid + j * NT;
        if (e < nC) { xn[e] = (x[e] - mean) * rstd * pg + pb; y[e] = pbo; }       // y starts as the to_out bias; the heads' shares are added below
    }
    lds_bar();
    const float scale = 0.17677669529663687f;            // dim_head^-0.5
    const int d = tid & 31, p8 = tid >> 5, PP = NT >> 5;  // (channel of the head, first position) of this thread in the [n][32] loops
    Pre nxt;
    for (int h = 0; h < 4; ++h) {
        if (h > 0) { stash_head(s, nxt, wbuf); lds_bar(); }
        if (h + 1 < 4) issue_head(s, h + 1, nxt);
        // q | k | v of this head: [n][32] each; four positions per thread share a weight value
        for (int which = 0; which < 3; ++which) {
            float* dst = q + which * n * DH;
            const float* w = wq + which * DH + d;
            for (int p0 = p8; p0 < n; p0 += 4 * PP) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                const float* xp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xp[j] = xn + ((p0 + j * PP) < n ? (p0 + j * PP) : p0) * C;
#pragma unroll 4
                for (int c = 0; c < C; ++c) {
                    const float wv = w[c * 96];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += xp[j][c] * wv;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) if (p0 + j * PP < n) dst[(p0 + j * PP) * DH + d] = acc[j];
            }
        }
        lds_bar();
        if (!full) {
            if (tid < 256) {   // k: softmax over the positions (per channel); 8 threads per column
                const int dd = tid >> 3, part = tid & 7;
                float m = -INFINITY;
                for (int p = part; p < n; p += 8) m = fmaxf(m, k[p * DH + dd]);
                m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
                float sum = 0.f;
                for (int p = part; p < n; p += 8) { const float e = __expf(k[p * DH + dd] - m); k[p * DH + dd] = e; sum += e; }
                sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
                const float inv = 1.0f / sum;
                for (int p = part; p < n; p += 8) k[p * DH + dd] *= inv;
            }
            // q: softmax over the head's 32 channels (per position), * scale; 4 threads per position
            for (int p0 = 0; p0 < n; p0 += NT / 4) {
                const int pix = p0 + (tid >> 2), part = tid & 3;
                const bool on = pix < n;
                float t[8], m = -INFINITY;
#pragma unroll
                for (int i = 0; i < 8; ++i) { t[i] = on ? q[pix * DH + part + 4 * i] : 0.f; m = fmaxf(m, t[i]); }
                m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2));
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) { t[i] = __expf(t[i] - m); sum += t[i]; }
                sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
                const float f = scale / sum;
                if (on) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) q[pix * DH + part + 4 * i] = t[i] * f;
                }
            }
            lds_bar();
            {   // ctx[dd][e] = sum_n k[n][dd] v[n][e]: thread (e = d, rows dd = p8 + PP j)
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                for (int p = 0; p < n; ++p) {
                    const float vv = v[p * DH + d];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += k[p * DH + ((p8 + j * PP) & 31)] * vv;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) if (p8 + j * PP < DH) cx[(p8 + j * PP) * DH + d] = acc[j];
            }
            lds_bar();
            for (int p0 = p8; p0 < n; p0 += 4 * PP) {     // o[pix][e] = sum_dd q[pix][dd] ctx[dd][e]
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                const float* qp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) qp[j] = q + ((p0 + j * PP) < n ? (p0 + j * PP) : p0) * DH;
#pragma unroll 8
                for (int dd = 0; dd < DH; ++dd) {
                    const float cv = cx[dd * DH + d];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += qp[j][dd] * cv;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) if (p0 + j * PP < n) o[(p0 + j * PP) * DH + d] = acc[j];
            }
        } else {
            for (int i = tid; i < n * n; i += NT) {       // sim[a][b] = (q_a * scale) . k_b
                const int a = i >> ln, b = i & (n - 1);
                float acc = 0.f;
#pragma unroll 8
                for (int dd = 0; dd < DH; ++dd) acc += (q[a * DH + dd] * scale) * k[b * DH + dd];
                cx[i] = acc;
            }
            lds_bar();
            for (int a = tid; a < n; a += NT) {           // softmax over the keys, one thread per query row (n <= 64)
                float m = -INFINITY;
                for (int b = 0; b < n; ++b) m = fmaxf(m, cx[a * n + b]);
                float sum = 0.f;
                for (int b = 0; b < n; ++b) { const float e = __expf(cx[a * n + b] - m); cx[a * n + b] = e; sum += e; }
                const float inv = 1.0f / sum;
                for (int b = 0; b < n; ++b) cx[a * n + b] *= inv;
            }
            lds_bar();
            for (int i = tid; i < n * DH; i += NT) {      // o[a][dd] = sum_b attn[a][b] v[b][dd]
                const int a = i >> 5, dd = i & 31;
                float acc = 0.f;
                for (int b = 0; b < n; ++b) acc += cx[a * n + b] * v[b * DH + dd];
                o[i] = acc;
            }
        }
        lds_bar();
        {   // y[pix][c] += sum_e o[pix][e] Wout[32 h + e][c]: thread (c = cc, positions (tid >> lc) + j (NT >> lc))
            const int pc = tid >> lc, PC = NT >> lc;
            for (int p0 = pc; p0 < n; p0 += 4 * PC) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                const float* op[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) op[j] = o + ((p0 + j * PC) < n ? (p0 + j * PC) : p0) * DH;
#pragma unroll 8
                for (int e = 0; e < DH; ++e) {
                    const float wv = wo[e * C + cc];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += op[j][e] * wv;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) if (p0 + j * PC < n) y[(p0 + j * PC) * C + cc] += acc[j];
            }
        }
        lds_bar();
    }
    float* out = L + s.out;
    if (!full) {                                          // to_out.1: GroupNorm(1), then the residual
        const float g2 = gsc(s.g2)[cc], b2 = gsc(s.be2)[cc];
        gn1_stats(y, nC, s.eps, red, &mean, &rstd);
#pragma unroll
        for (int j = 0; j < NO; ++j) { const int e = tid + j * NT; if (e < nC) out[e] = ((y[e] - mean) * rstd * g2 + b2) + x[e]; }
    } else {
#pragma unroll
        for (int j = 0; j < NO; ++j) { const int e = tid + j * NT; if (e < nC) out[e] = y[e] + x[e]; }
    }
}

// Either attention module on ONE position (the 1x1 level): the softmaxes collapse -- LinearAttention: k's softmax over a single position is
// 1, so the context is v itself and out = (sum_d softmax_d(q) * scale) v = scale * v; Attention: one key, attn = 1, out = v -- and the module is
// two matrix-vector products: v = GN1(x) . Wv (all four heads: the last 128 columns of to_qkv), y = b + f v . Wout, then to_out.1's
// GroupNorm(1) (LinearAttention only) and the residual.  The general kernel spent ~30 us here on 4 heads x 7 barrier-separated phases.
// wbuf holds Wv ([C][128], requested a step ahead) on entry; Wout ([128][C]) is staged here.
__device__ __forceinline__ void issue_wv(const SStep& s, Pre& r) {        // row c, columns 256..383 of to_qkv: 32 float4's per row
    const int n4 = s.C0 * 32, t = threadIdx.x;
#define FC_WV(k) gptr(s.w + (size_t)((t + (k) * NT) >> 5) * 384 + 256 + 4 * ((t + (k) * NT) & 31))
    FC_LD4(r.r0, t < n4, FC_WV(0)); FC_LD4(r.r1, t + NT < n4, FC_WV(1)); FC_LD4(r.r2, t + 2 * NT < n4, FC_WV(2)); FC_LD4(r.r3, t + 3 * NT < n4, FC_WV(3));
    FC_LD4(r.r4, t + 4 * NT < n4, FC_WV(4)); FC_LD4(r.r5, t + 5 * NT < n4, FC_WV(5)); FC_LD4(r.r6, t + 6 * NT < n4, FC_WV(6)); FC_LD4(r.r7, t + 7 * NT < n4, FC_WV(7));
#undef FC_WV
}
__device__ __forceinline__ void issue_flat8(const float* src, int n4, Pre& r) {   // up to 8 float4's per thread of a contiguous block
    gf4p p = gptr(src);
    const int t = threadIdx.x;
    FC_LD4(r.r0, t < n4, p + t); FC_LD4(r.r1, t + NT < n4, p + t + NT); FC_LD4(r.r2, t + 2 * NT < n4, p + t + 2 * NT); FC_LD4(r.r3, t + 3 * NT < n4, p + t + 3 * NT);
    FC_LD4(r.r4, t + 4 * NT < n4, p + t + 4 * NT); FC_LD4(r.r5, t + 5 * NT < n4, p + t + 5 * NT); FC_LD4(r.r6, t + 6 * NT < n4, p + t + 6 * NT); FC_LD4(r.r7, t + 7 * NT < n4, p + t + 7 * NT);
}
__device__ __forceinline__ void stash_flat8(const Pre& r, int n4, float* wbuf) {
    f4v* d = reinterpret_cast<f4v*>(wbuf);
    const int t = threadIdx.x;
    if (t < n4) d[t] = r.r0;
    if (t + NT < n4) d[t + NT] = r.r1;
    if (t + 2 * NT < n4) d[t + 2 * NT] = r.r2;
    if (t + 3 * NT < n4) d[t + 3 * NT] = r.r3;
    if (t + 4 * NT < n4) d[t + 4 * NT] = r.r4;
    if (t + 5 * NT < n4) d[t + 5 * NT] = r.r5;
    if (t + 6 * NT < n4) d[t + 6 * NT] = r.r6;
    if (t + 7 * NT < n4) d[t + 7 * NT] = r.r7;
}
__device__ __forceinline__ void issue_allheads(const SStep& s, Pre& r) {   // to_qkv [C][384] then to_out.0 [128][C], as one run of C * 128 float4's (C <= 16)
    const int nq4 = s.C0 * 96, n4 = s.C0 * 128, t = threadIdx.x;
    gf4p q = gptr(s.w), o = gptr(s.w2);
#define FC_AH(k) ((t + (k) * NT) < nq4 ? q + (t + (k) * NT) : o + ((t + (k) * NT) - nq4))
    FC_LD4(r.r0, t < n4, FC_AH(0)); FC_LD4(r.r1, t + NT < n4, FC_AH(1)); FC_LD4(r.r2, t + 2 * NT < n4, FC_AH(2)); FC_LD4(r.r3, t + 3 * NT < n4, FC_AH(3));
    FC_LD4(r.r4, t + 4 * NT < n4, FC_AH(4)); FC_LD4(r.r5, t + 5 * NT < n4, FC_AH(5)); FC_LD4(r.r6, t + 6 * NT < n4, FC_AH(6)); FC_LD4(r.r7, t + 7 * NT < n4, FC_AH(7));
#undef FC_AH
}
__device__ __forceinline__ void op_attention1(const SStep& s, float* L, float* red, float* wbuf) {
    const int tid = threadIdx.x, C = s.C0;
    const float* x = L + s.in0;
    float* xn = L + s.scratch;          // [C]
    float* vv = xn + C;                 // [128]
    float* y = vv + 128;                // [C]
    Pre wo;
    issue_flat8(s.w2, 32 * C, wo);      // Wout [128][C]: in flight while v is computed
    const int cc = tid & (C - 1);
    const float pg = gsc(s.gamma)[cc], pb = gsc(s.beta)[cc], pbo = gsc(s.b2)[cc];
    float g2 = 1.f, b2 = 0.f;
    if (!s.full) { g2 = gsc(s.g2)[cc]; b2 = gsc(s.be2)[cc]; }
    float mean, rstd;
    gn1_stats(x, C, s.eps, red, &mean, &rstd);
    if (tid < C) xn[tid] = (x[tid] - mean) * rstd * pg + pb;
    lds_bar();
    {   // v[j] = sum_c xn[c] Wv[c][j]: thread (j = tid & 127, half of the channels), halves met by one LDS add
        const int j = tid & 127, part = tid >> 7, ch = C >> 1;
        const float* w = wbuf + j;
        float acc = 0.f;
#pragma unroll 4
        for (int c = part * ch; c < (part + 1) * ch; ++c) acc += xn[c] * w[c * 128];
        if (part == 1) vv[j] = acc;
        lds_bar();
        if (part == 0) vv[j] += acc;
    }
    lds_bar();
    stash_flat8(wo, 32 * C, wbuf);      // every reader of Wv has passed the barrier above
    lds_bar();
    {   // y[c] = b[c] + f sum_j v[j] Wout[j][c]: thread (c = tid & (C - 1), slice of the 128 rows)
        const int parts = NT >> s.lc, part = tid >> s.lc, per = 128 / parts;
        const float* w = wbuf + cc;
        float acc = 0.f;
#pragma unroll 4
        for (int j = part * per; j < (part + 1) * per; ++j) acc += vv[j] * w[j * C];
        float* tmp = L + s.scratch + 2 * C + 128;      // [parts][C]
        tmp[part * C + cc] = acc;
        lds_bar();
        if (tid < C) {
            float t = 0.f;
            for (int p2 = 0; p2 < parts; ++p2) t += tmp[p2 * C + tid];
            y[tid] = pbo + (s.full ? 1.0f : 0.17677669529663687f) * t;
        }
    }
    lds_bar();
    float* out = L + s.out;
    if (!s.full) {
        gn1_stats(y, C, s.eps, red, &mean, &rstd);
        if (tid < C) out[tid] = ((y[tid] - mean) * rstd * g2 + b2) + x[tid];
    } else if (tid < C) out[tid] = y[tid] + x[tid];
}

// LinearAttention (unet.py:151-176) of a small module, ONE WAVE PER HEAD, all four heads at once: every phase of a head is a short serial
// chain (n <= 64 positions, 32 channels), so four of them side by side is the parallelism there is.  All four heads' weights sit in the
// staging buffer (to_qkv [C][384] then to_out.0 [128][C]: C * 512 floats, C <= 16).  Per head: a[n][32] (k, then q, then the head's output
// IN PLACE) | b[max(n, 32)][32] (v, then the 32 x 32 context where v was).
// Every LDS access of the phases is a 16-byte one and the operands that do not change with the position live in registers: a lane owns
// four channels of every eighth position, so a projection reads its C weight quads ONCE and then C/4 quads of x per position; the context
// is a 4 x 4 block per lane (one quad of k and one of v per position); the output keeps its 32 context rows, the head's share of to_out
// its 32 weight rows in registers.  (The first version read scalars: ~5000 LDS instructions per wave at n = 64, 50 us; this one ~450.)
// GW: the weights are read from global memory (L2) straight into those registers instead of the staging buffer -- C = 32 at n <= 16, where
// the four heads' weights (64 KB) do not fit it and the general form below takes 40 us for 4 positions.
template <int C, bool GW>
__device__ __forceinline__ void op_linattn_w(const SStep& s, float* L, float* red, float* wbuf) {
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6, n = s.Hi * s.Wi, nC = n * C;
    constexpr int DH = 32, CQ = C / 4;
    const float* x = L + s.in0;
    float* xn = L + s.scratch;
    float* yh = xn + nC;
    const int nb = n > DH ? n : DH;
    float* ha = yh + 4 * nC + h * ((n + nb) * DH);
    float* hb = ha + n * DH;
    auto ldq = [&](int off) -> f4v { if constexpr (GW) return *gptr(s.w + off); else return *reinterpret_cast<const f4v*>(wbuf + off); };              // to_qkv [C][384]
    auto ldo = [&](int off) -> f4v { if constexpr (GW) return *gptr(s.w2 + off); else return *reinterpret_cast<const f4v*>(wbuf + C * 384 + off); };   // to_out.0 [128][C]
    const int cc = tid & (C - 1);
    const float pg = gsc(s.gamma)[cc], pb = gsc(s.beta)[cc], pbo = gsc(s.b2)[cc], g2 = gsc(s.g2)[cc], b2 = gsc(s.be2)[cc];
    float mean, rstd;
    gn1_stats(x, nC, s.eps, red, &mean, &rstd);
#pragma unroll
    for (int j = 0; j < NO; ++j) { const int e = tid + j * NT; if (e < nC) xn[e] = (x[e] - mean) * rstd * pg + pb; }
    lds_bar();
    const int l8 = lane & 7, g8 = lane >> 3;              // a lane: quad l8 (channels 4 l8 .. 4 l8 + 3) of positions g8, g8 + 8, ...
    // a projection of this head (which: 0 q, 1 k, 2 v); q gets its softmax over the 32 channels (the eight lanes of a position) and the scale
    auto project = [&](int which, float* dst, bool qsoft) {
        f4v w[C];
#pragma unroll
        for (int c = 0; c < C; ++c) w[c] = ldq(c * 384 + which * 128 + h * DH + 4 * l8);
        for (int p = g8; p < n; p += 8) {
            f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) {
                const f4v xv = *reinterpret_cast<const f4v*>(xn + p * C + 4 * cq);
                acc += xv.x * w[4 * cq]; acc += xv.y * w[4 * cq + 1]; acc += xv.z * w[4 * cq + 2]; acc += xv.w * w[4 * cq + 3];
            }
            if (qsoft) {
                float m = fmaxf(fmaxf(acc.x, acc.y), fmaxf(acc.z, acc.w));
                m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
                acc.x = __expf(acc.x - m); acc.y = __expf(acc.y - m); acc.z = __expf(acc.z - m); acc.w = __expf(acc.w - m);
                float sum = (acc.x + acc.y) + (acc.z + acc.w);
                sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
                acc *= 0.17677669529663687f / sum;
            }
            *reinterpret_cast<f4v*>(dst + p * DH + 4 * l8) = acc;
        }
    };
    project(1, ha, false);                                // k
    project(2, hb, false);                                // v
    __builtin_amdgcn_wave_barrier();
    {   // k: softmax over the positions, column d; the two halves of the wave split the positions, a lane's values stay in registers
        const int d = lane & 31, hf = lane >> 5;
        float kv[32], m = -INFINITY;
#pragma unroll
        for (int j = 0; j < 32; ++j) { const int p = hf + 2 * j; kv[j] = p < n ? ha[p * DH + d] : -INFINITY; m = fmaxf(m, kv[j]); }
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) { kv[j] = (hf + 2 * j < n) ? __expf(kv[j] - m) : 0.f; sum += kv[j]; }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int j = 0; j < 32; ++j) { const int p = hf + 2 * j; if (p < n) ha[p * DH + d] = kv[j] * inv; }
    }
    __builtin_amdgcn_wave_barrier();
    {   // ctx[dd][e] = sum_p k[p][dd] v[p][e]: a 4 x 4 block per lane (rows 4 g8 .., columns 4 l8 ..)
        f4v c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
#pragma unroll 4
        for (int p = 0; p < n; ++p) {
            const f4v kk = *reinterpret_cast<const f4v*>(ha + p * DH + 4 * g8);
            const f4v vv = *reinterpret_cast<const f4v*>(hb + p * DH + 4 * l8);
            c0 += kk.x * vv; c1 += kk.y * vv; c2 += kk.z * vv; c3 += kk.w * vv;
        }
        __builtin_amdgcn_wave_barrier();                   // every lane's reads of v are done (one wave per head, in lockstep): the context takes its place
        float* cb = hb + (4 * g8) * DH + 4 * l8;
        *reinterpret_cast<f4v*>(cb) = c0; *reinterpret_cast<f4v*>(cb + DH) = c1; *reinterpret_cast<f4v*>(cb + 2 * DH) = c2; *reinterpret_cast<f4v*>(cb + 3 * DH) = c3;
    }
    __builtin_amdgcn_wave_barrier();
    project(0, ha, true);                                 // q (softmax over the channels, * scale), where k was
    __builtin_amdgcn_wave_barrier();
    {   // o[p][e] = sum_dd q[p][dd] ctx[dd][e], written over q's row p: the eight lanes that read a row are the ones that write it, after their last read
        f4v cr[DH];
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) cr[dd] = *reinterpret_cast<const f4v*>(hb + dd * DH + 4 * l8);
        for (int p = g8; p < n; p += 8) {
            f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f4v qv = *reinterpret_cast<const f4v*>(ha + p * DH + 4 * j);
                acc += qv.x * cr[4 * j]; acc += qv.y * cr[4 * j + 1]; acc += qv.z * cr[4 * j + 2]; acc += qv.w * cr[4 * j + 3];
            }
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<f4v*>(ha + p * DH + 4 * l8) = acc;
        }
    }
    __builtin_amdgcn_wave_barrier();
    {   // this head's share of to_out.0: yh[h][p][c] = sum_e o[p][e] Wout[32 h + e][c]: lane -> (channel quad, positions pl, pl + 64 / CQ, ...)
        const int cq = lane & (CQ - 1), pl = lane / CQ, PL = 64 / CQ;
        f4v wr[DH];
#pragma unroll
        for (int e = 0; e < DH; ++e) wr[e] = ldo((h * DH + e) * C + 4 * cq);
        for (int p = pl; p < n; p += PL) {
            f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f4v ov = *reinterpret_cast<const f4v*>(ha + p * DH + 4 * j);
                acc += ov.x * wr[4 * j]; acc += ov.y * wr[4 * j + 1]; acc += ov.z * wr[4 * j + 2]; acc += ov.w * wr[4 * j + 3];
            }
            *reinterpret_cast<f4v*>(yh + (size_t)(h * n + p) * C + 4 * cq) = acc;
        }
    }
    lds_bar();
    // y = bias + the four shares (fixed order), then to_out.1's GroupNorm(1) and the residual
    float yv[NO], a1 = 0.f;
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int e = tid + j * NT;
        yv[j] = 0.f;
        if (e < nC) { yv[j] = pbo + ((yh[e] + yh[nC + e]) + (yh[2 * nC + e] + yh[3 * nC + e])); a1 += yv[j]; }
    }
    const float m2 = block_sum(a1, red) / (float)nC;
    float q2 = 0.f;
#pragma unroll
    for (int j = 0; j < NO; ++j) { const int e = tid + j * NT; if (e < nC) { const float dlt = yv[j] - m2; q2 += dlt * dlt; } }
    const float r2 = 1.0f / sqrtf(block_sum(q2, red) / (float)nC + s.eps);
    float* out = L + s.out;
#pragma unroll
    for (int j = 0; j < NO; ++j) { const int e = tid + j * NT; if (e < nC) out[e] = ((yv[j] - m2) * r2 * g2 + b2) + x[e]; }
}

__global__ void __launch_bounds__(NT) unet_sample_kernel(const SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    __shared__ float red[64];
    __shared__ int last;
    const int b = blockIdx.x, tid = threadIdx.x, HW = a.HW, ch = a.ch;
    float* wbuf = L + a.wbuf_off;                         // 2 x WCH floats: weights on their way to the multiply-add loops
    const float* zl = L + a.zero_off;                     // 128 zeros: what a tap outside the image reads
    SStep* prog = reinterpret_cast<SStep*>(L + a.prog_off);     // the program itself: a step descriptor read from global memory was a cold round trip per step
    {
        const int nw = a.nsteps * (int)(sizeof(SStep) / 4);
        const unsigned* src = reinterpret_cast<const unsigned*>(a.prog);
        unsigned* dst = reinterpret_cast<unsigned*>(prog);
        for (int i = tid; i < nw; i += NT) dst[i] = src[i];
        for (int i = tid; i < 128; i += NT) L[a.zero_off + i] = 0.f;
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
    lds_bar();
    // guard: 0 always | 1 only with a mask | 2 only when mask_fusion_conv runs | 3 only without a mask | 4 mask but no fusion | 5 no fusion
    auto runs = [&](const SStep& s) {
        const int g = s.guard;
        return g == 0 || (g == 1 && has_mask) || (g == 2 && a.mask_fuse) || (g == 3 && !has_mask) || (g == 4 && has_mask && !a.mask_fuse) || (g == 5 && !a.mask_fuse);
    };
    unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, tA = 0, tB = 0;       // diagnostics: convolution steps split into stage | multiply-add | epilogue
    const bool clk = a.stamps && b == 0;
    Pre pre;
    int have = -1;                                        // the step whose first weights sit in `pre`
    for (int i = 0; i < a.nsteps; ++i) {
        const SStep& sl = prog[i];
        // the step's scalars in REGISTERS (a copy the LDS stores of the step cannot alias: read through the LDS reference, every field was
        // re-read after every store and barrier, a chain of LDS latencies per step); the chunk list stays behind `sl` (indexed at run time)
        const SStep s = sl;
        if (a.stamps && b == 0 && tid == 0) {
            a.stamps[2 * i] = __builtin_amdgcn_s_memrealtime();
            a.stamps[2 * i + 1] = (unsigned long long)s.op | ((unsigned long long)s.KS << 4) | ((unsigned long long)s.Cout << 8) | ((unsigned long long)(s.C0 + s.C1) << 20) |
                                  ((unsigned long long)s.Hi << 32) | ((unsigned long long)s.guard << 40);
        }
        if (!runs(s)) continue;                           // (uniform over the workgroup)
        if (clk) tA = __builtin_amdgcn_s_memrealtime();
        const bool needs = reads_weights(s);
        if (needs) {
            if (have != i) issue_first(s, pre);           // nobody asked ahead (the first such step)
            stash_first(s, pre, wbuf);
            have = -1;
        }
        {   // `pre` is free: the first weights of the NEXT step that reads any start travelling now, a whole step ahead of their use
            int nx = i + 1;
            while (nx < a.nsteps && !(runs(prog[nx]) && reads_weights(prog[nx]))) ++nx;
            if (nx < a.nsteps) { issue_first(prog[nx], pre); have = nx; }
        }
        if (needs) lds_bar();
        if (clk && s.op == S_CONV) { tB = __builtin_amdgcn_s_memrealtime(); ph0 += tB - tA; if (tid == 0) a.stamps[2 * a.nsteps + 8 + 3 * i] = tB - tA; }
        switch (s.op) {
            case S_CONV: {
                const ConvParams cp = conv_params(s, ss);
                if (s.fast == 9) conv_mac_fast_n<9>(s, L, wbuf, zl);
                else if (s.fast == 1) conv_mac_fast_n<1>(s, L, wbuf, zl);
                else conv_mac(s, sl, L, wbuf, zl);
                if (clk) { tA = __builtin_amdgcn_s_memrealtime(); ph1 += tA - tB; if (tid == 0) a.stamps[2 * a.nsteps + 9 + 3 * i] = tA - tB; }
                op_conv_epilogue(s, L, cp, red);
                if (clk) { tB = __builtin_amdgcn_s_memrealtime(); ph2 += tB - tA; if (tid == 0) a.stamps[2 * a.nsteps + 10 + 3 * i] = tB - tA; }
                break;
            }
            case S_NORM: op_norm(s, L, ss, red); break;
            case S_BILINEAR: op_bilinear(s, L); break;
            case S_LINATTN: op_attention(s, L, red, wbuf, false); break;
            case S_ATTN: op_attention(s, L, red, wbuf, true); break;
            case S_ATTN1: op_attention1(s, L, red, wbuf); break;
            case S_LINATTN_W: if (s.C0 == 8) op_linattn_w<8, false>(s, L, red, wbuf); else op_linattn_w<16, false>(s, L, red, wbuf); break;
            case S_LINATTN_G: op_linattn_w<32, true>(s, L, red, wbuf); break;
            case S_COPY: for (int e = tid; e < s.Cout; e += NT) L[s.out + e] = L[s.in0 + e]; break;
        }
        lds_bar();
    }
    if (a.stamps && b == 0 && tid == 0) {
        a.stamps[2 * a.nsteps] = __builtin_amdgcn_s_memrealtime(); a.stamps[2 * a.nsteps + 1] = 255;
        a.stamps[2 * a.nsteps + 2] = ph0; a.stamps[2 * a.nsteps + 3] = ph1; a.stamps[2 * a.nsteps + 4] = ph2;
    }
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
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(unet_sample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));   // (the kernel's static LDS: ~0.3 KB)
    done = true;
    return FC_OK;
}

int unet_sample_launch(const SampleArgs& a, int B, size_t lds_bytes, hipStream_t s) {
    hipLaunchKernelGGL(unet_sample_kernel, dim3(B), dim3(NT), lds_bytes, s, a);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc
