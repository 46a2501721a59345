// Residual(PreNorm(LinearAttention)) (unet.py:125-161,250) at the low-resolution levels (n = H*W <= 64 positions) in two launches.
//
// At 8x8 / 4x4 the unfused chain (1x1 conv | context | apply | 1x1 conv | finalize) is five launches of a few microseconds of work
// each, every boundary a ~7 us memory round trip; the arithmetic of the module is 0.2 - 0.4 GFLOP per batch.  The heads of a sample
// never meet before to_out, so:
//
//   la_head   grid (heads, B), 4 waves: GroupNorm(x[b]) into LDS; waves 0..2 compute this head's q / k / v (one 32-column MFMA tile
//             each, all rows, B operand streamed from global straight into the MFMA register layout) while wave 3 brings the head's
//             32 rows of to_out.0 into LDS; softmax_d(q) * scale and softmax_n(k) happen in the accumulator registers (lane shuffles);
//             context = k^T v; o = q . context; the head's share  o . Wout[32h .. 32h+32, :]  of to_out.0 goes to part[b][h].
//   la_join   grid (B): y = bias + sum_h part[b][h] in registers, GroupNorm(1) (to_out.1) by a two-pass block reduction, + x.
//
// A one-launch form with a workgroup per sample was measured first: 24 - 37 us per module, bound by one CU's fp32 matrix pipe
// (6.3 MFLOP of to_qkv per sample at 256 FLOP/clk = 10 us).  Splitting by head puts a sample on four CUs.
#include <cstdlib>
#include <string>

#include "common.h"
#include "stats_dev.h"

namespace fc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define FC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {
constexpr int DH = 32, HEADS = 4, HID = HEADS * DH, C3 = 3 * HID, QS = 3 * DH + 1, PS = 33, PF = 16, JT = 512, XPT = 16;

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// rows [k0, k0 + 2*PF) of a 32-column weight tile into the MFMA B layout (lane: row parity = half, column = l31)
__device__ __forceinline__ void wfetch(const float* __restrict__ wp, int ldw, int k0, float (&w)[PF]) {
#pragma unroll
    for (int u = 0; u < PF; ++u) w[u] = wp[(size_t)(k0 + 2 * u) * ldw];
}

// acc[mt] += A[mt*32 .. +32][0..K) . W[0..K)[32 columns]; A in LDS (row stride lda), W in global; K % (2*PF) == 0; `cur` already
// holds rows [0, 2*PF).  The next PF loads are in flight while the current PF feed the matrix pipe.
// Round 3: two register sets that swap roles (the loop is unrolled by two) and an UNCONDITIONAL request (the last round re-requests
// itself).  The round-2 form -- `if (more) wfetch(nxt)` ... `cur = nxt` -- was the pattern found in the convolution kernel's register-fed
// tile (conv_pipe.hip DB4): the compiler hoists the copy to the last use of each `cur` register, a copy of a load's destination waits for
// the load, and with a conditional request s_waitcnt (an immediate) must count for the path that skipped it -- every round then waited
// for the round trip it had just started, 4-8 dependent cold round trips per launch of a kernel that lives ~15 us.
template <int MT>
__device__ __forceinline__ void gemm_stream(const float* __restrict__ A, int lda, const float* __restrict__ wp, int ldw, int K, int l31, int half,
                                            float (&cur)[PF], f32x16 (&acc)[MT]) {
    float nxt[PF];
    auto round = [&](int k0, float (&wc)[PF], float (&wn)[PF]) {
        wfetch(wp, ldw, k0 + 2 * PF < K ? k0 + 2 * PF : k0, wn);
        __builtin_amdgcn_sched_barrier(0);      // the requests stay in front of this round's MFMAs
#pragma unroll
        for (int u = 0; u < PF; ++u) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = FC_MFMA(A[(mt * 32 + l31) * lda + k0 + 2 * u + half], wc[u], acc[mt]);
        }
    };
    if ((K & (4 * PF - 1)) == 0) {        // an even number of rounds (every U-Net width from 64 channels up): no branch between the two halves,
        for (int k0 = 0; k0 < K; k0 += 4 * PF) {      // so no path join for the wait counts to be pessimistic about
            round(k0, cur, nxt);
            round(k0 + 2 * PF, nxt, cur);
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += 4 * PF) {
            round(k0, cur, nxt);
            if (k0 + 2 * PF < K) round(k0 + 2 * PF, nxt, cur);
        }
    }
}

// y = bias + sum_h part[b][h], GroupNorm(1) (to_out.1) by a two-pass block reduction, + x -> out[b]: the closing step of the module, run by the
// LAST workgroup of a sample inside la_head (one-launch form) -- 256 threads, at most JE float4's each.  `red`: 8 floats of LDS.
constexpr int JE = 8;
template <bool GN>
__device__ __forceinline__ void join_tail(const LaArgs& a, int b, float* red) {
    const int tid = threadIdx.x, C = a.C, total4 = a.n * C / 4;
    const size_t per = (size_t)a.n * C;
    const float* pb = a.part + (size_t)b * HEADS * per;
    const float* xb = a.x + (size_t)b * per;
    float* ob = a.out + (size_t)b * per;
    auto bsum = [&](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    auto share4 = [](const float* q) {      // four values another workgroup (another XCD) has just written: device-scope loads, past this XCD's L2
        const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(q) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
    };
    float4 v[JE];
    float S = 0.f;
#pragma unroll
    for (int e = 0; e < JE; ++e) {
        const int i = tid + e * 256;
        v[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < total4) {
            const float4 p0 = share4(pb + (size_t)i * 4), p1 = share4(pb + per + (size_t)i * 4);
            const float4 p2 = share4(pb + 2 * per + (size_t)i * 4), p3 = share4(pb + 3 * per + (size_t)i * 4);
            const float4 bi = *reinterpret_cast<const float4*>(a.bout + (i * 4) % C);
            v[e].x = bi.x + ((p0.x + p1.x) + (p2.x + p3.x)); v[e].y = bi.y + ((p0.y + p1.y) + (p2.y + p3.y));
            v[e].z = bi.z + ((p0.z + p1.z) + (p2.z + p3.z)); v[e].w = bi.w + ((p0.w + p1.w) + (p2.w + p3.w));
            S += (v[e].x + v[e].y) + (v[e].z + v[e].w);
        }
    }
    float mu = 0.f, rs = 1.f;
    if (GN) {
        const float cnt = (float)a.n * (float)C;
        mu = bsum(S) / cnt;
        float Q = 0.f;
#pragma unroll
        for (int e = 0; e < JE; ++e)
            if (tid + e * 256 < total4) {
                const float dx = v[e].x - mu, dy = v[e].y - mu, dz = v[e].z - mu, dw = v[e].w - mu;
                Q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        rs = 1.0f / sqrtf(bsum(Q) / cnt + a.eps2);
    }
#pragma unroll
    for (int e = 0; e < JE; ++e) {
        const int i = tid + e * 256;
        if (i < total4) {
            const float4 x = *reinterpret_cast<const float4*>(xb + (size_t)i * 4);
            float4 o;
            if (GN) {
                const float4 g = *reinterpret_cast<const float4*>(a.g2 + (i * 4) % C), be = *reinterpret_cast<const float4*>(a.b2 + (i * 4) % C);
                o.x = ((v[e].x - mu) * rs * g.x + be.x) + x.x; o.y = ((v[e].y - mu) * rs * g.y + be.y) + x.y;
                o.z = ((v[e].z - mu) * rs * g.z + be.z) + x.z; o.w = ((v[e].w - mu) * rs * g.w + be.w) + x.w;
            } else {
                o.x = v[e].x + x.x; o.y = v[e].y + x.y; o.z = v[e].z + x.z; o.w = v[e].w + x.w;
            }
            *reinterpret_cast<float4*>(ob + (size_t)i * 4) = o;
        }
    }
}

template <int MT, bool FULL>   // MT = ceil(n / 32) row tiles; FULL: softmax(q k^T) v (Attention, unet.py:99-122) instead of the linear form
__global__ void __launch_bounds__(256) la_head_kernel(const LaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NP = MT * 32;
    const int C = a.C, n = a.n, XS = C + 1, WS = C + 32, CT = C >> 5;
    float* Ab = sm;
    float* Bb = Ab + C;
    float* ctxl = Bb + C;            // [32][PS]
    float* qkv = ctxl + DH * PS;     // [NP][QS]: q | k | v of this head
    float* os = qkv + NP * QS;       // [NP][PS]
    float* Wo = os + NP * PS;        // [32][WS]: rows 32h .. 32h+32 of to_out.0
    float* xs = Wo + DH * WS;        // [NP][XS]
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const float* xb = a.x + (size_t)b * n * C;

    // Nothing requested here depends on anything computed here, and inside a sampler step all of it is cold: the first round of
    // weights, the x tile and the norm parameters are in flight before the statistics are read -- one memory round trip for the
    // whole prologue instead of four dependent ones.
    const float* wp = a.wqkv + (size_t)half * C3 + (wave < 3 ? wave : 0) * HID + h * DH + l31;
    float cur[PF];
    if (wave < 3) wfetch(wp, C3, 0, cur);
    const int q4 = C >> 2, nx = n * q4;            // float4's of the (unpadded) x tile: at most XPT per thread
    float4 xr[XPT];
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
        const int i = tid + 256 * k;
        xr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nx) xr[k] = *reinterpret_cast<const float4*>(xb + (size_t)i * 4);
    }
    float pg[2], pbt[2];                           // C <= 512: two channels per thread
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = tid + 256 * k;
        pg[k] = c < C ? a.xf.gamma[c] : 0.f;
        pbt[k] = c < C ? a.xf.beta[c] : 0.f;
    }
    float mean, rstd;
    combine_partials(a.xf, b, 0, &mean, &rstd);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = tid + 256 * k;
        if (c < C) {
            const float sc = rstd * pg[k];
            Ab[c] = sc;
            Bb[c] = pbt[k] - mean * sc;
        }
    }
    for (int i = tid + nx; i < NP * q4; i += 256) {     // padding rows
        const int row = i / q4, c = (i - row * q4) * 4;
        float* d = xs + row * XS + c;
        d[0] = 0.f; d[1] = 0.f; d[2] = 0.f; d[3] = 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
        const int i = tid + 256 * k;
        if (i < nx) {
            const int row = i / q4, c = (i - row * q4) * 4;
            float4 v = xr[k];
            v.x = Ab[c] * v.x + Bb[c]; v.y = Ab[c + 1] * v.y + Bb[c + 1]; v.z = Ab[c + 2] * v.z + Bb[c + 2]; v.w = Ab[c + 3] * v.w + Bb[c + 3];
            float* d = xs + row * XS + c;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    __syncthreads();

    if (wave < 3) {
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
        gemm_stream<MT>(xs, XS, wp, C3, C, l31, half, cur, acc);
        if (FULL) {
            if (wave == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][r] *= 0.17677669529663687f;
            }
        } else if (wave == 0) {   // q: softmax over the head's 32 channels (the 32 lanes of a half-wave), * dim_head^-0.5
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[mt][r];
                    float m = v;
                    m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
                    m = fmaxf(m, __shfl_xor(m, 8)); m = fmaxf(m, __shfl_xor(m, 16));
                    const float e = __expf(v - m);
                    float s = e;
                    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8); s += __shfl_xor(s, 16);
                    acc[mt][r] = e * (0.17677669529663687f / s);
                }
        } else if (wave == 1) {   // k: softmax over the positions = the 16*MT rows here and the 16*MT of lane ^ 32
            float m = -INFINITY;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (mt * 32 + acc_row(r, half) < n) m = fmaxf(m, acc[mt][r]);
            m = fmaxf(m, __shfl_xor(m, 32));
            float s = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = (mt * 32 + acc_row(r, half) < n) ? __expf(acc[mt][r] - m) : 0.f;
                    acc[mt][r] = e;
                    s += e;
                }
            s += __shfl_xor(s, 32);
            const float f = 1.0f / s;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][r] *= f;
        }
        float* dst = qkv + wave * DH + l31;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(mt * 32 + acc_row(r, half)) * QS] = acc[mt][r];
    } else {
        // the head's 32 rows of to_out.0, sixteen 16-byte loads in flight per lane (a load -> store loop here was 32 dependent cold
        // round trips, 26 k cycles, and the whole workgroup waited for it at the barrier below)
        const float* wo = a.wout + (size_t)h * DH * C;
        const int tot = DH * q4;
        for (int i0 = lane; i0 < tot; i0 += 64 * 16) {
            float4 wv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = i0 + 64 * k;
                wv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < tot) { const int r = i / q4, c = (i - r * q4) * 4; wv[k] = *reinterpret_cast<const float4*>(wo + (size_t)r * C + c); }
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = i0 + 64 * k;
                if (i < tot) { const int r = i / q4, c = (i - r * q4) * 4; *reinterpret_cast<float4*>(Wo + r * WS + c) = wv[k]; }
            }
        }
    }
    __syncthreads();

    if (FULL) {
        // sim[i][j] = q_i . k_j (q already scaled), softmax over the keys j, o = attn . v; wave -> 32 query rows.  The probabilities
        // take the place of the x tile (C >= NP, checked on the host).
        constexpr int PSS = NP + 1;
        float* ps = xs;
        if (wave < MT) {
            f32x16 st[MT];
#pragma unroll
            for (int nt = 0; nt < MT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[nt][r] = 0.f;
            const float* qp = qkv + (wave * 32 + l31) * QS;
#pragma unroll 4
            for (int s = 0; s < DH / 2; ++s) {
                const float qa = qp[2 * s + half];
#pragma unroll
                for (int nt = 0; nt < MT; ++nt) st[nt] = FC_MFMA(qa, qkv[(nt * 32 + l31) * QS + DH + 2 * s + half], st[nt]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float m = -INFINITY;
#pragma unroll
                for (int nt = 0; nt < MT; ++nt) if (nt * 32 + l31 < n) m = fmaxf(m, st[nt][r]);
                m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
                m = fmaxf(m, __shfl_xor(m, 8)); m = fmaxf(m, __shfl_xor(m, 16));
                float e[MT], sum = 0.f;
#pragma unroll
                for (int nt = 0; nt < MT; ++nt) { e[nt] = (nt * 32 + l31 < n) ? __expf(st[nt][r] - m) : 0.f; sum += e[nt]; }
                sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8); sum += __shfl_xor(sum, 16);
                const float inv = 1.0f / sum;
#pragma unroll
                for (int nt = 0; nt < MT; ++nt) ps[(wave * 32 + acc_row(r, half)) * PSS + nt * 32 + l31] = e[nt] * inv;
            }
        }
        __syncthreads();
        if (wave < MT) {
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
            const float* pp = ps + (wave * 32 + l31) * PSS;
            const float* vp = qkv + 2 * DH + l31;
#pragma unroll 4
            for (int s = 0; s < NP / 2; ++s) o = FC_MFMA(pp[2 * s + half], vp[(2 * s + half) * QS], o);
#pragma unroll
            for (int r = 0; r < 16; ++r) os[(wave * 32 + acc_row(r, half)) * PS + l31] = o[r];
        }
        __syncthreads();
    } else {
        // context[d][e] = sum_n k[n][d] v[n][e]
        if (wave == 0) {
            f32x16 c;
    #pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = 0.f;
            const float* kp = qkv + DH + l31;
            const float* vp = qkv + 2 * DH + l31;
    #pragma unroll 4
            for (int s = 0; s < NP / 2; ++s) c = FC_MFMA(kp[(2 * s + half) * QS], vp[(2 * s + half) * QS], c);
    #pragma unroll
            for (int r = 0; r < 16; ++r) ctxl[acc_row(r, half) * PS + l31] = c[r];
        }
        __syncthreads();
        // o[n][e] = sum_d q[n][d] ctx[d][e]
        if (wave < MT) {
            f32x16 o;
    #pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
            const float* qp = qkv + (wave * 32 + l31) * QS;
    #pragma unroll 4
            for (int s = 0; s < DH / 2; ++s) o = FC_MFMA(qp[2 * s + half], ctxl[(2 * s + half) * PS + l31], o);
    #pragma unroll
            for (int r = 0; r < 16; ++r) os[(wave * 32 + acc_row(r, half)) * PS + l31] = o[r];
        }
        __syncthreads();
    }
    // this head's share of to_out.0: part[b][h][n][C] = o . Wout[32h .. 32h+32, :]; wave -> channel tiles wave, wave + 4
    float* pb = a.part + ((size_t)b * HEADS + h) * n * C;
    for (int ct = wave; ct < CT; ct += 4) {
        f32x16 y[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[mt][r] = 0.f;
#pragma unroll 4
        for (int s = 0; s < DH / 2; ++s) {
            const float w = Wo[(2 * s + half) * WS + ct * 32 + l31];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) y[mt] = FC_MFMA(os[(mt * 32 + l31) * PS + 2 * s + half], w, y[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + acc_row(r, half);
                if (row >= n) continue;
                float* dst = pb + (size_t)row * C + ct * 32 + l31;
                if (a.tickets) __hip_atomic_store(dst, y[mt][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: through the per-XCD L2
                else *dst = y[mt][r];
            }
    }
    if (a.tickets) {
        // One-launch form: the workgroup of a sample that draws the last ticket of this launch closes the module for it.  No workgroup waits
        // for another, so nothing here depends on how many of them are resident (unlike the convolution tails' meetings).  The heads of a
        // sample sit on different XCDs, whose L2s are not coherent with each other: the shares travel as device-scope relaxed atomics (sc1
        // stores above, sc1 loads in join_tail), ordered by vmcnt(0) + the workgroup barrier in front of the ticket -- NOT by a release /
        // acquire fence pair, which writes back and invalidates the whole L2 in every workgroup (measured here: 58 instead of 22 us per
        // module; conv_dev.h has the same finding for the convolution tails).  The counters only ever grow: `heads` per launch.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) sm[0] = (__hip_atomic_fetch_add(a.tickets + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % HEADS == HEADS - 1) ? 1.f : 0.f;
        __syncthreads();
        if (sm[0] == 0.f) return;
        join_tail<!FULL>(a, b, sm + 8);
    }
}

// ---- eight-wave form (round 4) ------------------------------------------------------------------------------------------------
// The same module with 512 threads: the q / k / v projections are K-split over two waves each (waves 0-2: channels [0, C/2), waves 3-5:
// [C/2, C)), the halves meet in LDS; waves 6-7 bring the head's rows of to_out.0 into LDS.  The weights come from the k-step-quad copy
// (pack kind 6: a lane's B operands of four k-steps are ONE 16-byte load, four loads per round of 16 k-steps instead of sixteen), two
// register sets swapping roles.  Why: with four waves a (sample, head) workgroup -- the only one on its CU -- ran the projections on
// three waves of one SIMD each, 4 KB of weights in flight per wave, and the phase took 9.3 of the kernel's ~15 us for 3.4 us of matrix
// work (round-3 knock-outs): it waits for weights.  Twice the waves stream twice the bytes at a time and halve the dependent MFMA chain.
constexpr int PQ = 4;   // 16-byte weight quads per round = 16 k-steps
__device__ __forceinline__ void wfetch4(const float4* __restrict__ wq, int blk_stride, int blk0, float4 (&w)[PQ]) {
#pragma unroll
    for (int i = 0; i < PQ; ++i) w[i] = wq[(size_t)(blk0 + i) * blk_stride];
}
template <int MT>
__device__ __forceinline__ void gemm_stream4(const float* __restrict__ A, int lda, const float4* __restrict__ wq, int blk_stride, int blk0, int nblk, int l31, int half,
                                             float4 (&cur)[PQ], f32x16 (&acc)[MT]) {
    float4 nxt[PQ];
    auto round = [&](int b0, float4 (&wc)[PQ], float4 (&wn)[PQ]) {
        wfetch4(wq, blk_stride, b0 + PQ < blk0 + nblk ? b0 + PQ : b0, wn);      // unconditional: the last round re-requests itself
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PQ; ++i) {
            const float* ap = A + 8 * (b0 + i) + half;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float* ar = ap + (mt * 32 + l31) * lda;
                acc[mt] = FC_MFMA(ar[0], wc[i].x, acc[mt]);
                acc[mt] = FC_MFMA(ar[2], wc[i].y, acc[mt]);
                acc[mt] = FC_MFMA(ar[4], wc[i].z, acc[mt]);
                acc[mt] = FC_MFMA(ar[6], wc[i].w, acc[mt]);
            }
        }
    };
    const int rounds = nblk / PQ;
    for (int r = 0; r < rounds; r += 2) {
        round(blk0 + r * PQ, cur, nxt);
        if (r + 1 < rounds) round(blk0 + (r + 1) * PQ, nxt, cur);
    }
}

template <int MT, bool FULL>
__global__ void __launch_bounds__(512) la_head8_kernel(const LaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NP = MT * 32, NT = 512, XP8 = XPT / 2;
    const int C = a.C, n = a.n, XS = C + 1, WS = C + 32, CT = C >> 5;
    float* Ab = sm;
    float* Bb = Ab + C;
    float* ctxl = Bb + C;            // [32][PS]
    float* qkv = ctxl + DH * PS;     // [NP][QS]: q | k | v of this head
    float* os = qkv + NP * QS;       // [NP][PS]
    float* Wo = os + NP * PS;        // [32][WS]: rows 32h .. 32h+32 of to_out.0
    float* xs = Wo + DH * WS;        // [NP][XS]
    float* red = xs + NP * XS;       // [3][MT][16][64]: the upper K-halves' partial accumulators
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const float* xb = a.x + (size_t)b * n * C;
    const int proj = wave % 3, part = wave / 3;          // waves 0-5: projection, K half; waves 6-7 (part == 2): to_out rows
    const int nb = C >> 3, nbh = nb >> 1;                // 8-channel blocks: all, per K half
    // every cold operand first: the first round of weights, the x tile, the norm parameters -- then the statistics
    const float4* wq = reinterpret_cast<const float4*>(a.wqkv4) + (size_t)half * C3 + proj * HID + h * DH + l31;
    float4 cur[PQ];
    if (part < 2) wfetch4(wq, 2 * C3, part * nbh, cur);
    const int q4 = C >> 2, nx = n * q4;
    float4 xr[XP8];
#pragma unroll
    for (int k = 0; k < XP8; ++k) {
        const int i = tid + NT * k;
        xr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nx) xr[k] = *reinterpret_cast<const float4*>(xb + (size_t)i * 4);
    }
    const float pg = tid < C ? a.xf.gamma[tid] : 0.f, pbt = tid < C ? a.xf.beta[tid] : 0.f;     // C <= 512 = threads
    float mean, rstd;
    combine_partials(a.xf, b, 0, &mean, &rstd);
    if (tid < C) {
        const float sc = rstd * pg;
        Ab[tid] = sc;
        Bb[tid] = pbt - mean * sc;
    }
    for (int i = tid + nx; i < NP * q4; i += NT) {     // padding rows
        const int row = i / q4, c = (i - row * q4) * 4;
        float* d = xs + row * XS + c;
        d[0] = 0.f; d[1] = 0.f; d[2] = 0.f; d[3] = 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < XP8; ++k) {
        const int i = tid + NT * k;
        if (i < nx) {
            const int row = i / q4, c = (i - row * q4) * 4;
            float4 v = xr[k];
            v.x = Ab[c] * v.x + Bb[c]; v.y = Ab[c + 1] * v.y + Bb[c + 1]; v.z = Ab[c + 2] * v.z + Bb[c + 2]; v.w = Ab[c + 3] * v.w + Bb[c + 3];
            float* d = xs + row * XS + c;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    __syncthreads();

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    if (part < 2) {
        gemm_stream4<MT>(xs, XS, wq, 2 * C3, part * nbh, nbh, l31, half, cur, acc);
        if (part == 1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((proj * MT + mt) * 16 + r) * 64 + lane] = acc[mt][r];
        }
    } else {
        // the head's 32 rows of to_out.0 on two waves, sixteen 16-byte loads in flight per lane
        const float* wo = a.wout + (size_t)h * DH * C;
        const int tot = DH * q4, t2 = tid - 384;       // 0..127
        for (int i0 = t2; i0 < tot; i0 += 128 * 16) {
            float4 wv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = i0 + 128 * k;
                wv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < tot) { const int r = i / q4, c = (i - r * q4) * 4; wv[k] = *reinterpret_cast<const float4*>(wo + (size_t)r * C + c); }
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = i0 + 128 * k;
                if (i < tot) { const int r = i / q4, c = (i - r * q4) * 4; *reinterpret_cast<float4*>(Wo + r * WS + c) = wv[k]; }
            }
        }
    }
    __syncthreads();
    if (part == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] += red[((proj * MT + mt) * 16 + r) * 64 + lane];
        if (FULL) {
            if (wave == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][r] *= 0.17677669529663687f;
            }
        } else if (wave == 0) {   // q: softmax over the head's 32 channels (the 32 lanes of a half-wave), * dim_head^-0.5
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[mt][r];
                    float m = v;
                    m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
                    m = fmaxf(m, __shfl_xor(m, 8)); m = fmaxf(m, __shfl_xor(m, 16));
                    const float e = __expf(v - m);
                    float s_ = e;
                    s_ += __shfl_xor(s_, 1); s_ += __shfl_xor(s_, 2); s_ += __shfl_xor(s_, 4); s_ += __shfl_xor(s_, 8); s_ += __shfl_xor(s_, 16);
                    acc[mt][r] = e * (0.17677669529663687f / s_);
                }
        } else if (wave == 1) {   // k: softmax over the positions = the 16*MT rows here and the 16*MT of lane ^ 32
            float m = -INFINITY;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (mt * 32 + acc_row(r, half) < n) m = fmaxf(m, acc[mt][r]);
            m = fmaxf(m, __shfl_xor(m, 32));
            float s_ = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = (mt * 32 + acc_row(r, half) < n) ? __expf(acc[mt][r] - m) : 0.f;
                    acc[mt][r] = e;
                    s_ += e;
                }
            s_ += __shfl_xor(s_, 32);
            const float f = 1.0f / s_;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][r] *= f;
        }
        float* dst = qkv + wave * DH + l31;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(mt * 32 + acc_row(r, half)) * QS] = acc[mt][r];
    }
    __syncthreads();

    if (FULL) {
        constexpr int PSS = NP + 1;
        float* ps = xs;
        if (wave < MT) {
            f32x16 st[MT];
#pragma unroll
            for (int nt = 0; nt < MT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[nt][r] = 0.f;
            const float* qp = qkv + (wave * 32 + l31) * QS;
#pragma unroll 4
            for (int s_ = 0; s_ < DH / 2; ++s_) {
                const float qa = qp[2 * s_ + half];
#pragma unroll
                for (int nt = 0; nt < MT; ++nt) st[nt] = FC_MFMA(qa, qkv[(nt * 32 + l31) * QS + DH + 2 * s_ + half], st[nt]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float m = -INFINITY;
#pragma unroll
                for (int nt = 0; nt < MT; ++nt) if (nt * 32 + l31 < n) m = fmaxf(m, st[nt][r]);
                m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
                m = fmaxf(m, __shfl_xor(m, 8)); m = fmaxf(m, __shfl_xor(m, 16));
                float e[MT], sum = 0.f;
#pragma unroll
                for (int nt = 0; nt < MT; ++nt) { e[nt] = (nt * 32 + l31 < n) ? __expf(st[nt][r] - m) : 0.f; sum += e[nt]; }
                sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8); sum += __shfl_xor(sum, 16);
                const float inv = 1.0f / sum;
#pragma unroll
                for (int nt = 0; nt < MT; ++nt) ps[(wave * 32 + acc_row(r, half)) * PSS + nt * 32 + l31] = e[nt] * inv;
            }
        }
        __syncthreads();
        if (wave < MT) {
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
            const float* pp = ps + (wave * 32 + l31) * PSS;
            const float* vp = qkv + 2 * DH + l31;
#pragma unroll 4
            for (int s_ = 0; s_ < NP / 2; ++s_) o = FC_MFMA(pp[2 * s_ + half], vp[(2 * s_ + half) * QS], o);
#pragma unroll
            for (int r = 0; r < 16; ++r) os[(wave * 32 + acc_row(r, half)) * PS + l31] = o[r];
        }
        __syncthreads();
    } else {
        if (wave == 0) {
            f32x16 c;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = 0.f;
            const float* kp = qkv + DH + l31;
            const float* vp = qkv + 2 * DH + l31;
#pragma unroll 4
            for (int s_ = 0; s_ < NP / 2; ++s_) c = FC_MFMA(kp[(2 * s_ + half) * QS], vp[(2 * s_ + half) * QS], c);
#pragma unroll
            for (int r = 0; r < 16; ++r) ctxl[acc_row(r, half) * PS + l31] = c[r];
        }
        __syncthreads();
        if (wave < MT) {
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
            const float* qp = qkv + (wave * 32 + l31) * QS;
#pragma unroll 4
            for (int s_ = 0; s_ < DH / 2; ++s_) o = FC_MFMA(qp[2 * s_ + half], ctxl[(2 * s_ + half) * PS + l31], o);
#pragma unroll
            for (int r = 0; r < 16; ++r) os[(wave * 32 + acc_row(r, half)) * PS + l31] = o[r];
        }
        __syncthreads();
    }
    // this head's share of to_out.0: part[b][h][n][C] = o . Wout[32h .. 32h+32, :]; wave -> channel tiles wave, wave + 8
    float* pb = a.part + ((size_t)b * HEADS + h) * n * C;
    for (int ct = wave; ct < CT; ct += 8) {
        f32x16 y[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[mt][r] = 0.f;
#pragma unroll 4
        for (int s_ = 0; s_ < DH / 2; ++s_) {
            const float w = Wo[(2 * s_ + half) * WS + ct * 32 + l31];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) y[mt] = FC_MFMA(os[(mt * 32 + l31) * PS + 2 * s_ + half], w, y[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + acc_row(r, half);
                if (row >= n) continue;
                pb[(size_t)row * C + ct * 32 + l31] = y[mt][r];
            }
    }
}

__device__ __forceinline__ float block_sum_j(float v, float* red /*[JT/64]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < JT / 64; ++w) s += red[w];
    return s;
}

template <int EPT, bool GN>   // float4's per thread: EPT * JT * 4 >= n * C; GN: to_out.1's GroupNorm(1) before the residual
__global__ void __launch_bounds__(JT) la_join_kernel(const LaArgs a) {
    __shared__ float red[JT / 64];
    const int b = blockIdx.x, tid = threadIdx.x, C = a.C, total4 = a.n * C / 4;
    const size_t per = (size_t)a.n * C;
    const float* pb = a.part + (size_t)b * HEADS * per;
    const float* xb = a.x + (size_t)b * per;
    float* ob = a.out + (size_t)b * per;
    float4 v[EPT], xv[EPT], gv[EPT], bev[EPT];   // residual and norm parameters requested with the shares: nothing cold after the reductions
    float S = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * JT;
        v[e] = xv[e] = gv[e] = bev[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < total4) {
            xv[e] = *reinterpret_cast<const float4*>(xb + (size_t)i * 4);
            if (GN) {
                gv[e] = *reinterpret_cast<const float4*>(a.g2 + (i * 4) % C);
                bev[e] = *reinterpret_cast<const float4*>(a.b2 + (i * 4) % C);
            }
            const float4 p0 = *reinterpret_cast<const float4*>(pb + (size_t)i * 4), p1 = *reinterpret_cast<const float4*>(pb + per + (size_t)i * 4);
            const float4 p2 = *reinterpret_cast<const float4*>(pb + 2 * per + (size_t)i * 4), p3 = *reinterpret_cast<const float4*>(pb + 3 * per + (size_t)i * 4);
            const float4 bi = *reinterpret_cast<const float4*>(a.bout + (i * 4) % C);
            v[e].x = bi.x + ((p0.x + p1.x) + (p2.x + p3.x)); v[e].y = bi.y + ((p0.y + p1.y) + (p2.y + p3.y));
            v[e].z = bi.z + ((p0.z + p1.z) + (p2.z + p3.z)); v[e].w = bi.w + ((p0.w + p1.w) + (p2.w + p3.w));
            S += (v[e].x + v[e].y) + (v[e].z + v[e].w);
        }
    }
    float mu = 0.f, rs = 1.f;
    if (GN) {
        const float cnt = (float)a.n * (float)C;
        mu = block_sum_j(S, red) / cnt;
        float Q = 0.f;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (tid + e * JT < total4) {
                const float dx = v[e].x - mu, dy = v[e].y - mu, dz = v[e].z - mu, dw = v[e].w - mu;
                Q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        rs = 1.0f / sqrtf(block_sum_j(Q, red) / cnt + a.eps2);
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * JT;
        if (i < total4) {
            const float4 x = xv[e];
            float4 o;
            if (GN) {
                const float4 g = gv[e], be = bev[e];
                o.x = ((v[e].x - mu) * rs * g.x + be.x) + x.x; o.y = ((v[e].y - mu) * rs * g.y + be.y) + x.y;
                o.z = ((v[e].z - mu) * rs * g.z + be.z) + x.z; o.w = ((v[e].w - mu) * rs * g.w + be.w) + x.w;
            } else {
                o.x = v[e].x + x.x; o.y = v[e].y + x.y; o.z = v[e].z + x.z; o.w = v[e].w + x.w;
            }
            *reinterpret_cast<float4*>(ob + (size_t)i * 4) = o;
        }
    }
}

size_t head_lds(int n, int C) {
    const int NP = n <= 32 ? 32 : 64;
    return (2 * (size_t)C + DH * PS + (size_t)NP * QS + (size_t)NP * PS + (size_t)DH * (C + 32) + (size_t)NP * (C + 1)) * sizeof(float);
}
}  // namespace

// the eight-wave form needs the k-step-quad copy of to_qkv, whole rounds of 16 k-steps in each K half (C % 64 == 0), C <= 512 threads, no tickets
static bool head8_ok(const LaArgs& a) {
    static const bool off = [] { const char* e = std::getenv("FLOCODER_AMD_LA_HEAD"); return e && std::string(e) == "4"; }();
    return !off && a.wqkv4 && !a.tickets && (a.C % 64) == 0 && a.C <= 512 && (size_t)a.n * a.C <= (size_t)(XPT / 2) * 512 * 4 &&
           head_lds(a.n, a.C) + (size_t)3 * (a.n <= 32 ? 1 : 2) * 1024 * sizeof(float) <= 160 * 1024;
}
template <bool FULL>
static void launch_pair(const LaArgs& a, hipStream_t s) {
    const size_t lds = head_lds(a.n, a.C);
    if (head8_ok(a)) {
        const size_t lds8 = lds + (size_t)3 * (a.n <= 32 ? 1 : 2) * 1024 * sizeof(float);
        if (a.n <= 32) hipLaunchKernelGGL((la_head8_kernel<1, FULL>), dim3(HEADS, a.B), dim3(512), lds8, s, a);
        else hipLaunchKernelGGL((la_head8_kernel<2, FULL>), dim3(HEADS, a.B), dim3(512), lds8, s, a);
    }
    else if (a.n <= 32) hipLaunchKernelGGL((la_head_kernel<1, FULL>), dim3(HEADS, a.B), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((la_head_kernel<2, FULL>), dim3(HEADS, a.B), dim3(256), lds, s, a);
    if (a.tickets) return;          // the last workgroup of every sample has done la_join's work
    const int total4 = a.n * a.C / 4;
    if (total4 <= JT) hipLaunchKernelGGL((la_join_kernel<1, !FULL>), dim3(a.B), dim3(JT), 0, s, a);
    else if (total4 <= 2 * JT) hipLaunchKernelGGL((la_join_kernel<2, !FULL>), dim3(a.B), dim3(JT), 0, s, a);
    else if (total4 <= 4 * JT) hipLaunchKernelGGL((la_join_kernel<4, !FULL>), dim3(a.B), dim3(JT), 0, s, a);
    else hipLaunchKernelGGL((la_join_kernel<8, !FULL>), dim3(a.B), dim3(JT), 0, s, a);
}

int linattn_sample_init() {
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_head_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_head_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_head_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_head_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_head8_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_head8_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_head8_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_head8_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return FC_OK;
}

bool linattn_sample_supported(int n, int C, int heads) {
    return heads == HEADS && n >= 1 && n <= 64 && (C % 32) == 0 && C >= 32 && C <= 512 && n * C <= 8 * JT * 4 && n * C <= XPT * 256 * 4 &&
           head_lds(n, C) <= 160 * 1024;
}

// the one-launch form (LaArgs::tickets): the closing workgroup holds a sample's n*C values in JE float4's per thread
bool linattn_sample_one_launch(int n, int C) { return (size_t)n * C <= (size_t)JE * 256 * 4; }

bool attn_sample_supported(int n, int C, int heads) { return linattn_sample_supported(n, C, heads) && C >= (n <= 32 ? 32 : 64); }

int linattn_sample_launch(const LaArgs& a, hipStream_t s) {
    if (!linattn_sample_supported(a.n, a.C, a.heads)) return fail(FC_E_SHAPE, "linattn_sample: unsupported shape");
    if (a.xf.mode != 1 || a.xf.G != 1 || !a.xf.stats) return fail(FC_E_ARG, "linattn_sample: needs GroupNorm(1) statistics of x");
    if (!a.g2 || !a.b2 || !a.out || !a.part || !a.bout) return fail(FC_E_ARG, "linattn_sample: to_out parameters / scratch / output missing");
    if (a.tickets && !linattn_sample_one_launch(a.n, a.C)) return fail(FC_E_SHAPE, "linattn_sample: sample too large for the one-launch form");
    launch_pair<false>(a, s);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// Residual(PreNorm(Attention)) (unet.py:99-122,262) on the same two kernels: softmax(q k^T) v per head, to_out without a norm
int attn_sample_launch(const LaArgs& a, hipStream_t s) {
    if (!attn_sample_supported(a.n, a.C, a.heads)) return fail(FC_E_SHAPE, "attn_sample: unsupported shape");
    if (a.xf.mode != 1 || a.xf.G != 1 || !a.xf.stats) return fail(FC_E_ARG, "attn_sample: needs GroupNorm(1) statistics of x");
    if (!a.out || !a.part || !a.bout) return fail(FC_E_ARG, "attn_sample: to_out parameters / scratch / output missing");
    if (a.tickets && !linattn_sample_one_launch(a.n, a.C)) return fail(FC_E_SHAPE, "attn_sample: sample too large for the one-launch form");
    launch_pair<true>(a, s);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc
