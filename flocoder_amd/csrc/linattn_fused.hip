// Residual(PreNorm(LinearAttention)) (unet.py:125-161) without ever materialising q, k, v or the attention output:
//
//   la_ctx    grid (heads, B): k, v = GroupNorm(x) . Wk, Wv on the matrix pipe, 32 positions at a time per wave, fed straight into
//             an ONLINE softmax over the positions (running column maximum, rescaled accumulators) and the 32x32 context
//             ctx[d][e] = sum_n softmax_n(k)[d][n] v[e][n], itself an MFMA over the positions.  The four waves' partial
//             (max, sum, context) triples are merged at the end.  Traffic: x once per head, the context out.
//   la_apply  grid (ceil(n/128), B): q = GroupNorm(x) . Wq, softmax over the head dimension by lane shuffles, out = q . ctx,
//             y = out . Wout + b, all per 32-row tile per wave with the operands handed between the three GEMMs through a
//             4 KB LDS tile; epilogue writes y and the GroupNorm(1) partial of the tile for to_out.1.
//
// At the 32x32 level this replaces a 100 MB qkv round trip (to_qkv 32->384 over 65536 pixels) and two 33 MB passes by one
// read of x per kernel.  fp32 MFMA throughout (exact products), so results match the unfused kernels to summation order.
#include <cstdlib>
#include <string>

#include "common.h"
#include "stats_dev.h"

namespace fc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define FC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int LDH = 32, LHEADS = 4, LHID = LHEADS * LDH, LC3 = 3 * LHID, PS = 33;

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// stage rows [r0, r0+32) x channels [c0, c0+cl) of GroupNorm(x[b]) into a wave-private tile xw[32][XS]; rows >= n are zero
__device__ __forceinline__ void stage_x(const LaArgs& a, const float* Ab, const float* Bb, float* xw, int XS, int b, int r0, int c0, int cl, int lane) {
    const int q4 = cl >> 2;
    for (int idx = lane; idx < 32 * q4; idx += 64) {
        const int row = idx / q4, q = idx - row * q4, nn = r0 + row, c = c0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nn < a.n) {
            v = *reinterpret_cast<const float4*>(a.x + ((size_t)b * a.n + nn) * a.C + c);
            v.x = Ab[c] * v.x + Bb[c]; v.y = Ab[c + 1] * v.y + Bb[c + 1]; v.z = Ab[c + 2] * v.z + Bb[c + 2]; v.w = Ab[c + 3] * v.w + Bb[c + 3];
        }
        float* d = xw + row * XS + 4 * q;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

__global__ void __launch_bounds__(256) la_ctx_kernel(const LaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = a.C, CC = C < 64 ? C : 64, XS = CC + 1;
    float* Ab = sm;
    float* Bb = Ab + C;
    float* Wl = Bb + C;                 // [C][64]: k columns | v columns of this head
    float* xt = Wl + C * 64;            // [4][32][XS]
    float* Et = xt + 4 * 32 * XS;       // [4][32][PS]
    float* Vt = Et + 4 * 32 * PS;       // [4][32][PS]
    float* sct = Vt + 4 * 32 * PS;      // [4][32]
    float* mz = sct + 128;              // [4][2][32]
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    float mean, rstd;
    combine_partials(a.xf, b, 0, &mean, &rstd);
    for (int c = tid; c < C; c += 256) {
        const float s = rstd * a.xf.gamma[c];
        Ab[c] = s;
        Bb[c] = a.xf.beta[c] - mean * s;
    }
    for (int i = tid; i < C * 64; i += 256) {
        const int c = i >> 6, j = i & 63;
        Wl[i] = a.wqkv[(size_t)c * LC3 + (j < 32 ? LHID + h * LDH + j : 2 * LHID + h * LDH + (j - 32))];
    }
    float* xw = xt + wave * 32 * XS;
    float* Ew = Et + wave * 32 * PS;
    float* Vw = Vt + wave * 32 * PS;
    float* scw = sct + wave * 32;
    const int nblk = (a.n + 31) >> 5, iters = (nblk + 3) >> 2;
    f32x16 cacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) cacc[r] = 0.f;
    float m_run = -INFINITY, z_run = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int rb = it * 4 + wave, r0 = rb * 32;
        const bool active = rb < nblk;
        f32x16 ak, av;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ak[r] = 0.f; av[r] = 0.f; }
        for (int c0 = 0; c0 < C; c0 += CC) {
            const int cl = C - c0 < CC ? C - c0 : CC;
            __syncthreads();
            if (active) stage_x(a, Ab, Bb, xw, XS, b, r0, c0, cl, lane);
            __syncthreads();
            if (active) {
                for (int s = 0; s < (cl >> 1); ++s) {
                    const float xa = xw[l31 * XS + 2 * s + half];
                    const float* wr = Wl + (c0 + 2 * s + half) * 64 + l31;
                    ak = FC_MFMA(xa, wr[0], ak);
                    av = FC_MFMA(xa, wr[32], av);
                }
            }
        }
        if (active) {
            float bm = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) if (r0 + acc_row(r, half) < a.n) bm = fmaxf(bm, ak[r]);
            bm = fmaxf(bm, __shfl_xor(bm, 32));
            const float m_new = fmaxf(m_run, bm);
            const float sc = __expf(m_run - m_new);
            float zs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(r, half);
                const float e = (r0 + row < a.n) ? __expf(ak[r] - m_new) : 0.f;
                zs += e;
                Ew[row * PS + l31] = e;
                Vw[row * PS + l31] = av[r];
            }
            zs += __shfl_xor(zs, 32);
            z_run = z_run * sc + zs;
            m_run = m_new;
            if (half == 0) scw[l31] = sc;
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cacc[r] *= scw[acc_row(r, half)];
#pragma unroll 4
            for (int s = 0; s < 16; ++s) cacc = FC_MFMA(Ew[(2 * s + half) * PS + l31], Vw[(2 * s + half) * PS + l31], cacc);
        }
    }
    // merge the four waves' (max, sum, context)
    __syncthreads();
    if (half == 0) { mz[(wave * 2) * 32 + l31] = m_run; mz[(wave * 2 + 1) * 32 + l31] = z_run; }
    __syncthreads();
    if (tid < 128) {
        const int w = tid >> 5, d = tid & 31;
        float M = mz[d];
        for (int k = 1; k < 4; ++k) M = fmaxf(M, mz[(k * 2) * 32 + d]);
        float Z = 0.f;
        for (int k = 0; k < 4; ++k) Z += mz[(k * 2 + 1) * 32 + d] * __expf(mz[(k * 2) * 32 + d] - M);
        sct[w * 32 + d] = __expf(mz[(w * 2) * 32 + d] - M) / Z;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = acc_row(r, half);
        Et[(wave * 32 + d) * PS + l31] = cacc[r] * sct[wave * 32 + d];
    }
    __syncthreads();
    for (int i = tid; i < LDH * LDH; i += 256) {
        const int d = i >> 5, e = i & 31;
        a.ctx[((size_t)(b * LHEADS + h) * LDH + d) * LDH + e] = (Et[d * PS + e] + Et[(32 + d) * PS + e]) + (Et[(64 + d) * PS + e] + Et[(96 + d) * PS + e]);
    }
}

template <int CT>   // CT = ceil(C / 32) output-channel tiles of to_out
__global__ void __launch_bounds__(256) la_apply_kernel(const LaArgs a, int T, float n_t) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = a.C, CC = C < 64 ? C : 64, XS = CC + 1, WO = CT * 32;
    const int wc_floats = CC * LHID > 32 * WO ? CC * LHID : 32 * WO;
    float* Ab = sm;
    float* Bb = Ab + C;
    float* Wc = Bb + C;                   // [CC][128] (Wq chunk), later [32][WO] (one head's rows of Wout)
    float* xt = Wc + wc_floats;           // [4][32][XS]
    float* Pt = xt + 4 * 32 * XS;         // [4][32][PS]
    float* ctxl = Pt + 4 * 32 * PS;       // [4][32][PS]
    __shared__ float red[4];
    const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int r0 = tile * 128 + wave * 32;
    float mean, rstd;
    combine_partials(a.xf, b, 0, &mean, &rstd);
    for (int c = tid; c < C; c += 256) {
        const float s = rstd * a.xf.gamma[c];
        Ab[c] = s;
        Bb[c] = a.xf.beta[c] - mean * s;
    }
    for (int i = tid; i < LHEADS * LDH * LDH; i += 256) ctxl[(i >> 5) * PS + (i & 31)] = a.ctx[(size_t)b * LHEADS * LDH * LDH + i];
    float* xw = xt + wave * 32 * XS;
    float* Pw = Pt + wave * 32 * PS;
    f32x16 q[LHEADS];
#pragma unroll
    for (int hh = 0; hh < LHEADS; ++hh)
#pragma unroll
        for (int r = 0; r < 16; ++r) q[hh][r] = 0.f;
    for (int c0 = 0; c0 < C; c0 += CC) {
        const int cl = C - c0 < CC ? C - c0 : CC;
        __syncthreads();
        for (int i = tid; i < cl * (LHID / 4); i += 256) {
            const int cc = i / (LHID / 4), j = (i - cc * (LHID / 4)) * 4;
            *reinterpret_cast<float4*>(Wc + cc * LHID + j) = *reinterpret_cast<const float4*>(a.wqkv + (size_t)(c0 + cc) * LC3 + j);
        }
        stage_x(a, Ab, Bb, xw, XS, b, r0, c0, cl, lane);
        __syncthreads();
        for (int s = 0; s < (cl >> 1); ++s) {
            const float xa = xw[l31 * XS + 2 * s + half];
            const float* wr = Wc + (2 * s + half) * LHID + l31;
#pragma unroll
            for (int hh = 0; hh < LHEADS; ++hh) q[hh] = FC_MFMA(xa, wr[hh * 32], q[hh]);
        }
    }
    // softmax over the 32 head channels of every (row, head): a row's channels sit in the 32 lanes of one half-wave
    const float scale = 0.17677669529663687f;
#pragma unroll
    for (int hh = 0; hh < LHEADS; ++hh)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = q[hh][r];
            float m = v;
            m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
            m = fmaxf(m, __shfl_xor(m, 8)); m = fmaxf(m, __shfl_xor(m, 16));
            const float e = __expf(v - m);
            float s = e;
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8); s += __shfl_xor(s, 16);
            q[hh][r] = e * (scale / s);
        }
    f32x16 y[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[ct][r] = 0.f;
#pragma unroll
    for (int hh = 0; hh < LHEADS; ++hh) {
        __syncthreads();
        for (int i = tid; i < 32 * WO; i += 256) {
            const int k = i / WO, c = i - k * WO;
            Wc[i] = c < C ? a.wout[(size_t)(hh * LDH + k) * C + c] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Pw[acc_row(r, half) * PS + l31] = q[hh][r];
        __syncthreads();
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) o = FC_MFMA(Pw[l31 * PS + 2 * s + half], ctxl[(hh * LDH + 2 * s + half) * PS + l31], o);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) Pw[acc_row(r, half) * PS + l31] = o[r];
        __syncthreads();
#pragma unroll 2
        for (int s = 0; s < 16; ++s) {
            const float oa = Pw[l31 * PS + 2 * s + half];
            const float* wr = Wc + (2 * s + half) * WO + l31;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) y[ct] = FC_MFMA(oa, wr[ct * 32], y[ct]);
        }
    }
    // statistics before the stores: a barrier after them would wait for the store round trip
    float S = 0.f, Q = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = ct * 32 + l31;
        const float bias = (c < C && a.bout) ? a.bout[c] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = y[ct][r] + bias;
            y[ct][r] = v;
            if (r0 + acc_row(r, half) < a.n && c < C) { S += v; Q += v * v; }
        }
    }
    const float St = block_sum(S, red);
    const float Qt = block_sum(Q, red);
    if (tid == 0) {
        const float mt = St / n_t;
        float* d = a.stats_out + ((size_t)b * T + tile) * 2;
        d[0] = mt;
        d[1] = Qt - St * mt;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = ct * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = r0 + acc_row(r, half);
            if (nn < a.n && c < C) a.y[((size_t)b * a.n + nn) * C + c] = y[ct][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Fast path, C <= 64 and C % 8 == 0 (every n >= 256 level of the dim 16 / 32 models): the K loop is a single chunk, so all
// weights stay in LDS for the life of the workgroup, the next 32-row tile of x travels HBM -> registers while the current one
// is on the matrix pipe, and every LDS tile is private to its wave -- no workgroup barrier inside the loops (a wave's LDS
// operations execute in order).
template <int NQ>   // NQ = C / 8 float4 loads per lane per 32-row tile
__device__ __forceinline__ void fetch_x(const LaArgs& a, int b, int r0, int lane, float4 (&pre)[NQ]) {
    constexpr int q4 = NQ * 2;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int idx = lane + 64 * j, row = idx / q4, q = idx - row * q4, nn = r0 + row;
        pre[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nn < a.n) pre[j] = *reinterpret_cast<const float4*>(a.x + ((size_t)b * a.n + nn) * (NQ * 8) + 4 * q);
    }
}
template <int NQ>
__device__ __forceinline__ void store_x(const LaArgs& a, const float* Ab, const float* Bb, float* xw, int XS, int r0, int lane, const float4 (&pre)[NQ]) {
    constexpr int q4 = NQ * 2;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int idx = lane + 64 * j, row = idx / q4, q = idx - row * q4, c = 4 * q;
        float4 v = pre[j];
        if (r0 + row < a.n) { v.x = Ab[c] * v.x + Bb[c]; v.y = Ab[c + 1] * v.y + Bb[c + 1]; v.z = Ab[c + 2] * v.z + Bb[c + 2]; v.w = Ab[c + 3] * v.w + Bb[c + 3]; }
        float* d = xw + row * XS + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

// NW = waves per workgroup (4 or 8).  Eight (round 4) where a (sample, head) has at least eight 32-position blocks: the workgroup is the
// only one on its CU, so with four waves every SIMD held ONE wave and nothing covered its LDS round trips and softmax arithmetic between
// the matrix phases; with eight, two waves share a SIMD's matrix pipe.  The split over positions stays inside the workgroup (the merge of
// the partial (max, sum, context) triples is the same LDS pass over NW instead of four entries).
template <int NQ, int NW>
__global__ void __launch_bounds__(64 * NW) la_ctx_fast_kernel(const LaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int C = NQ * 8, XS = C + 1, NT = 64 * NW;
    float* Ab = sm;
    float* Bb = Ab + C;
    float* Wl = Bb + C;                 // [C][64]
    float* xt = Wl + C * 64;            // [NW][32][XS]
    float* Et = xt + NW * 32 * XS;      // [NW][32][PS]
    float* Vt = Et + NW * 32 * PS;
    float* sct = Vt + NW * 32 * PS;     // [NW][32]
    float* mz = sct + NW * 32;          // [NW][2][32]
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int nblk = (a.n + 31) >> 5;
    // every cold operand is requested before anything is waited for: first x tile, statistics, norm parameters, weights
    float4 pre[NQ];
    if (wave < nblk) fetch_x<NQ>(a, b, wave * 32, lane, pre);
    PartPre pp;
    partials_request(a.xf, b, 0, pp);
    const float pg = tid < C ? a.xf.gamma[tid] : 0.f, pbt = tid < C ? a.xf.beta[tid] : 0.f;   // C <= 64 < 256 threads
    // explicit registers: written as a load -> LDS store loop the compiler waits for every load before the next (ISA: vmcnt(0) per
    // iteration) -- dependent cold round trips at the head of the kernel
    constexpr int NWL = C * 64 / NT;
    float wl[NWL];
#pragma unroll
    for (int k = 0; k < NWL; ++k) {
        const int i = tid + NT * k, c = i >> 6, j = i & 63;
        wl[k] = a.wqkv[(size_t)c * LC3 + (j < 32 ? LHID + h * LDH + j : 2 * LHID + h * LDH + (j - 32))];
    }
#pragma unroll
    for (int k = 0; k < NWL; ++k) Wl[tid + NT * k] = wl[k];
    float mean, rstd;
    partials_finish(a.xf, b, 0, pp, &mean, &rstd);
    if (tid < C) {
        const float s = rstd * pg;
        Ab[tid] = s;
        Bb[tid] = pbt - mean * s;
    }
    __syncthreads();
    float* xw = xt + wave * 32 * XS;
    float* Ew = Et + wave * 32 * PS;
    float* Vw = Vt + wave * 32 * PS;
    float* scw = sct + wave * 32;
    f32x16 cacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) cacc[r] = 0.f;
    float m_run = -INFINITY, z_run = 0.f;
    for (int rb = wave; rb < nblk; rb += NW) {
        const int r0 = rb * 32;
        store_x<NQ>(a, Ab, Bb, xw, XS, r0, lane, pre);
        __builtin_amdgcn_wave_barrier();
        if (rb + NW < nblk) fetch_x<NQ>(a, b, r0 + 32 * NW, lane, pre);
        f32x16 ak, av;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ak[r] = 0.f; av[r] = 0.f; }
#pragma unroll 4
        for (int s = 0; s < (C >> 1); ++s) {
            const float xa = xw[l31 * XS + 2 * s + half];
            const float* wr = Wl + (2 * s + half) * 64 + l31;
            ak = FC_MFMA(xa, wr[0], ak);
            av = FC_MFMA(xa, wr[32], av);
        }
        float bm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) if (r0 + acc_row(r, half) < a.n) bm = fmaxf(bm, ak[r]);
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float m_new = fmaxf(m_run, bm);
        const float sc = __expf(m_run - m_new);
        float zs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, half);
            const float e = (r0 + row < a.n) ? __expf(ak[r] - m_new) : 0.f;
            zs += e;
            Ew[row * PS + l31] = e;
            Vw[row * PS + l31] = av[r];
        }
        zs += __shfl_xor(zs, 32);
        z_run = z_run * sc + zs;
        m_run = m_new;
        if (half == 0) scw[l31] = sc;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) cacc[r] *= scw[acc_row(r, half)];
#pragma unroll 4
        for (int s = 0; s < 16; ++s) cacc = FC_MFMA(Ew[(2 * s + half) * PS + l31], Vw[(2 * s + half) * PS + l31], cacc);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (half == 0) { mz[(wave * 2) * 32 + l31] = m_run; mz[(wave * 2 + 1) * 32 + l31] = z_run; }
    __syncthreads();
    if (tid < 32 * NW) {
        const int w = tid >> 5, d = tid & 31;
        float M = mz[d];
        for (int k = 1; k < NW; ++k) M = fmaxf(M, mz[(k * 2) * 32 + d]);
        float Z = 0.f;
        for (int k = 0; k < NW; ++k) Z += mz[(k * 2 + 1) * 32 + d] * __expf(mz[(k * 2) * 32 + d] - M);
        sct[w * 32 + d] = __expf(mz[(w * 2) * 32 + d] - M) / Z;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = acc_row(r, half);
        Et[(wave * 32 + d) * PS + l31] = cacc[r] * sct[wave * 32 + d];
    }
    __syncthreads();
    for (int i = tid; i < LDH * LDH; i += NT) {
        const int d = i >> 5, e = i & 31;
        float v = (Et[d * PS + e] + Et[(32 + d) * PS + e]) + (Et[(64 + d) * PS + e] + Et[(96 + d) * PS + e]);
        if (NW == 8) v += (Et[(128 + d) * PS + e] + Et[(160 + d) * PS + e]) + (Et[(192 + d) * PS + e] + Et[(224 + d) * PS + e]);
        a.ctx[((size_t)(b * LHEADS + h) * LDH + d) * LDH + e] = v;
    }
}

template <int NQ, int CT>
__global__ void __launch_bounds__(256) la_apply_fast_kernel(const LaArgs a, int T, float n_t) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int C = NQ * 8, XS = C + 1 > PS ? C + 1 : PS, WO = CT * 32;
    float* Ab = sm;
    float* Bb = Ab + C;
    float* Wq = Bb + C;                    // [C][128]
    float* Wo = Wq + C * LHID;             // [128][WO]
    float* ctxl = Wo + LHID * WO;          // [4][32][PS]
    float* xt = ctxl + LHEADS * LDH * PS;  // [4][32][XS]: x tile, then the P / out tile of the head in flight
    __shared__ float red[4];
    __shared__ unsigned epoch_s;
    __shared__ float gv[2 * kPartPre];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int ntiles = (a.n + 127) >> 7;
    // the fused close (a.gran; one tile per workgroup, T <= kPartPre): this launch's epoch from the sample's arrival counter
    unsigned arrival = 0;
    if (a.gran && tid == 0) arrival = __hip_atomic_fetch_add(a.sync + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // every cold operand is requested before anything is waited for: x tile, statistics, norm parameters, to_out bias, weights, context
    float4 pre[NQ];
    fetch_x<NQ>(a, b, blockIdx.x * 128 + wave * 32, lane, pre);
    PartPre pp;
    partials_request(a.xf, b, 0, pp);
    const float pg = tid < C ? a.xf.gamma[tid] : 0.f, pbt = tid < C ? a.xf.beta[tid] : 0.f;   // C <= 64 < 256 threads
    float pbias[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) pbias[ct] = (ct * 32 + l31 < C && a.bout) ? a.bout[ct * 32 + l31] : 0.f;
    // Wq, Wout and the context into explicit registers first, LDS after: as load -> store loops they were 4 + 8 + 16 dependent round
    // trips (ISA: s_waitcnt vmcnt(0) inside each loop), about half of this kernel's 30 us inside a sampler step
    constexpr int NWO = WO / 2;
    float4 wq[NQ], cx[4];
    float wo[NWO];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int i = tid + 256 * k, cc = i / (LHID / 4), j = (i - cc * (LHID / 4)) * 4;
        wq[k] = *reinterpret_cast<const float4*>(a.wqkv + (size_t)cc * LC3 + j);
    }
#pragma unroll
    for (int k = 0; k < NWO; ++k) {
        const int i = tid + 256 * k, kk = i / WO, c = i - kk * WO;
        wo[k] = c < C ? a.wout[(size_t)kk * C + c] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) cx[k] = *reinterpret_cast<const float4*>(a.ctx + (size_t)b * LHEADS * LDH * LDH + 4 * (tid + 256 * k));
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int i = tid + 256 * k, cc = i / (LHID / 4), j = (i - cc * (LHID / 4)) * 4;
        *reinterpret_cast<float4*>(Wq + cc * LHID + j) = wq[k];
    }
#pragma unroll
    for (int k = 0; k < NWO; ++k) Wo[tid + 256 * k] = wo[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = 4 * (tid + 256 * k);
        float* d = ctxl + (i >> 5) * PS + (i & 31);
        d[0] = cx[k].x; d[1] = cx[k].y; d[2] = cx[k].z; d[3] = cx[k].w;
    }
    {
        float mean, rstd;
        partials_finish(a.xf, b, 0, pp, &mean, &rstd);
        if (tid < C) {
            const float s = rstd * pg;
            Ab[tid] = s;
            Bb[tid] = pbt - mean * s;
        }
        if (a.gran && tid == 0) epoch_s = arrival / (unsigned)T + 1u;
    }
    __syncthreads();
    float* xw = xt + wave * 32 * XS;
    const float scale = 0.17677669529663687f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int r0 = tile * 128 + wave * 32;
        store_x<NQ>(a, Ab, Bb, xw, XS, r0, lane, pre);
        __builtin_amdgcn_wave_barrier();
        if (tile + (int)gridDim.x < ntiles) fetch_x<NQ>(a, b, r0 + 128 * gridDim.x, lane, pre);
        f32x16 q[LHEADS];
#pragma unroll
        for (int hh = 0; hh < LHEADS; ++hh)
#pragma unroll
            for (int r = 0; r < 16; ++r) q[hh][r] = 0.f;
#pragma unroll 2
        for (int s = 0; s < (C >> 1); ++s) {
            const float xa = xw[l31 * XS + 2 * s + half];
            const float* wr = Wq + (2 * s + half) * LHID + l31;
#pragma unroll
            for (int hh = 0; hh < LHEADS; ++hh) q[hh] = FC_MFMA(xa, wr[hh * 32], q[hh]);
        }
        f32x16 y[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[ct][r] = 0.f;
        float* Pw = xw;                     // the x tile is dead: its buffer carries P, then out, of each head (row stride XS >= 33)
#pragma unroll
        for (int hh = 0; hh < LHEADS; ++hh) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) Pw[acc_row(r, half) * XS + l31] = q[hh][r];
            __builtin_amdgcn_wave_barrier();
            {   // softmax over the head's 32 channels: lane (row l31, half) owns 16 of them
                float v[16];
                float* pr = Pw + l31 * XS + half * 16;
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < 16; ++j) { v[j] = pr[j]; m = fmaxf(m, v[j]); }
                m = fmaxf(m, __shfl_xor(m, 32));
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) { v[j] = __expf(v[j] - m); sum += v[j]; }
                sum += __shfl_xor(sum, 32);
                const float f = scale / sum;
#pragma unroll
                for (int j = 0; j < 16; ++j) pr[j] = v[j] * f;
            }
            __builtin_amdgcn_wave_barrier();
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll 4
            for (int s = 0; s < 16; ++s) o = FC_MFMA(Pw[l31 * XS + 2 * s + half], ctxl[(hh * LDH + 2 * s + half) * PS + l31], o);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) Pw[acc_row(r, half) * XS + l31] = o[r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll 2
            for (int s = 0; s < 16; ++s) {
                const float oa = Pw[l31 * XS + 2 * s + half];
                const float* wr = Wo + (hh * LDH + 2 * s + half) * WO + l31;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) y[ct] = FC_MFMA(oa, wr[ct * 32], y[ct]);
            }
        }
        float S = 0.f, Q = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 32 + l31;
            const float bias = pbias[ct];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = y[ct][r] + bias;
                y[ct][r] = v;
                if (r0 + acc_row(r, half) < a.n && c < C) { S += v; Q += v * v; }
            }
        }
        const float St = block_sum(S, red);
        const float Qt = block_sum(Q, red);
        if (a.gran) {
            // ---- the module closed here: out = GroupNorm(1)(y) * g2 + b2 + x, the statistics completed by the sample's other tiles ----
            // The partials travel as ONE 8-byte write-through store each, {epoch, bits}: the data is its own flag (the convolution tails'
            // hand-off, conv_dev.h); sc1 accesses that bypass the per-XCD L2, no fence.  finalize_kernel's arithmetic (stats_dev.h
            // partials_finish) in the same order, so the exclusive and the shared plan agree to the last bit.
            const unsigned epoch = epoch_s;
            if (tid == 0) {
                const float mt = St / n_t;
                unsigned long long* gp = a.gran + ((size_t)b * T + tile) * 2;
                const unsigned long long tag = (unsigned long long)epoch << 32;
                __hip_atomic_store(gp, tag | __float_as_uint(mt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gp + 1, tag | __float_as_uint(Qt - St * mt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // the residual and the norm parameters are requested BEFORE the wait: they depend on nothing computed here
            float rs[CT][16], g2[CT], b2[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = ct * 32 + l31;
                g2[ct] = c < C ? a.g2[c] : 0.f;
                b2[ct] = c < C ? a.b2[c] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nn = r0 + acc_row(r, half);
                    rs[ct][r] = (nn < a.n && c < C) ? a.x[((size_t)b * a.n + nn) * C + c] : 0.f;
                }
            }
            if (tid < 2 * T) {          // one granule per thread, polled until it carries this launch's epoch.  Bounded: a residency mistake
                                        // must not hang the device -- the statistic becomes NaN and the handle's error word is set
                const unsigned long long* gp = a.gran + (size_t)b * T * 2 + tid;
                unsigned long long v = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while ((unsigned)(v >> 32) != epoch) {
                    if (++spins > (1 << 20)) { if (a.err) *a.err = 1; v = 0x7fc00000ull; break; }
                    __builtin_amdgcn_s_sleep(2);
                    v = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                gv[tid] = __uint_as_float((unsigned)v);
            }
            __syncthreads();
            float sm2 = 0.f;
            for (int t = 0; t < T; ++t) sm2 += gv[2 * t];
            const float mean = sm2 / (float)T;
            float m2 = 0.f, dv = 0.f;
            for (int t = 0; t < T; ++t) {
                const float d = gv[2 * t] - mean;
                m2 += gv[2 * t + 1];
                dv += d * d;
            }
            const float rstd = 1.0f / sqrtf((m2 + n_t * dv) / (n_t * (float)T) + a.eps2);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = ct * 32 + l31;
                const float A = rstd * g2[ct], Bc = b2[ct] - mean * A;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nn = r0 + acc_row(r, half);
                    if (nn < a.n && c < C) a.out[((size_t)b * a.n + nn) * C + c] = (A * y[ct][r] + Bc) + rs[ct][r];
                }
            }
            continue;                   // (one tile per workgroup: the loop ends here)
        }
        if (tid == 0) {
            const float mt = St / n_t;
            float* d = a.stats_out + ((size_t)b * T + tile) * 2;
            d[0] = mt;
            d[1] = Qt - St * mt;
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = r0 + acc_row(r, half);
                if (nn < a.n && c < C) a.y[((size_t)b * a.n + nn) * C + c] = y[ct][r];
            }
        }
    }
}

static size_t la_ctx_fast_lds(int C, int NW) { return (size_t)(2 * C + C * 64 + NW * 32 * (C + 1) + 2 * NW * 32 * PS + 32 * NW + 64 * NW) * sizeof(float); }
static int la_ctx_waves(int n) {      // eight waves once every one of them has a 32-position block of its own (FLOCODER_AMD_LA_CTX_WAVES=4: never)
    static const int forced = [] { const char* e = std::getenv("FLOCODER_AMD_LA_CTX_WAVES"); return e ? std::atoi(e) : 0; }();
    if (forced == 4 || forced == 8) return (forced == 8 && n >= 256) ? 8 : 4;
    return n >= 256 ? 8 : 4;
}
static size_t la_apply_fast_lds(int C, int CT) {
    const int XS = C + 1 > PS ? C + 1 : PS;
    return (size_t)(2 * C + C * LHID + LHID * CT * 32 + LHEADS * LDH * PS + 4 * 32 * XS) * sizeof(float);
}
template <int NQ>
static int launch_fast(const LaArgs& a, hipStream_t s) {
    if (la_ctx_waves(a.n) == 8) hipLaunchKernelGGL((la_ctx_fast_kernel<NQ, 8>), dim3(LHEADS, a.B), dim3(512), la_ctx_fast_lds(a.C, 8), s, a);
    else hipLaunchKernelGGL((la_ctx_fast_kernel<NQ, 4>), dim3(LHEADS, a.B), dim3(256), la_ctx_fast_lds(a.C, 4), s, a);
    FC_HIP(hipGetLastError());
    const int T = linattn_fused_tiles(a.n);
    const float n_t = linattn_fused_nt(a.n, a.C);
    const int tiles = cdiv(a.n, 128);
    int gx = tiles;                                   // two tiles per workgroup once that still leaves >= 256 workgroups
    // Round 4: one 128-position tile per workgroup again.  Two tiles per workgroup (round 2: half the weight staging) leave 256 workgroups of
    // four waves -- one wave per SIMD, nothing to cover the LDS round trips between the five dependent phases of a head; with one tile each,
    // two workgroups share a CU (67 KB of LDS each).  Measured per module at n = 1024: 55.0 -> 51.0 and 50.9 -> 47.0 us; sampler 793 -> 805
    // samples/s on one box (791 / 788 with four-wave la_ctx and two tiles).  FLOCODER_AMD_LA_APPLY_GX=half: the old form.
    static const bool gx_half = [] { const char* e = std::getenv("FLOCODER_AMD_LA_APPLY_GX"); return e && std::string(e) == "half"; }();
    if (!a.gran && gx_half && tiles >= 2 && (tiles / 2) * a.B >= 256) gx = tiles / 2;      // (the fused close: one tile per workgroup, always)
    const dim3 grid(gx, a.B);
    if (a.C <= 32) hipLaunchKernelGGL((la_apply_fast_kernel<NQ, 1>), grid, dim3(256), la_apply_fast_lds(a.C, 1), s, a, T, n_t);
    else hipLaunchKernelGGL((la_apply_fast_kernel<NQ, 2>), grid, dim3(256), la_apply_fast_lds(a.C, 2), s, a, T, n_t);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
template <int NQ>
static int init_fast() {
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_ctx_fast_kernel<NQ, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_ctx_fast_kernel<NQ, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_apply_fast_kernel<NQ, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_apply_fast_kernel<NQ, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    return FC_OK;
}

static size_t la_ctx_lds(int C) {
    const int CC = C < 64 ? C : 64;
    return (size_t)(2 * C + C * 64 + 4 * 32 * (CC + 1) + 2 * 4 * 32 * PS + 128 + 256) * sizeof(float);
}
static size_t la_apply_lds(int C, int CT) {
    const int CC = C < 64 ? C : 64, WO = CT * 32;
    const int wc = CC * LHID > 32 * WO ? CC * LHID : 32 * WO;
    return (size_t)(2 * C + wc + 4 * 32 * (CC + 1) + 2 * 4 * 32 * PS) * sizeof(float);
}

int linattn_fused_tiles(int n) { return n >= 128 ? n / 128 : 1; }
float linattn_fused_nt(int n, int C) { return (float)((n >= 128 ? 128 : n) * C); }

int linattn_fused_init() {
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_ctx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_apply_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_apply_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_apply_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(la_apply_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    FC_TRY(init_fast<1>()); FC_TRY(init_fast<2>()); FC_TRY(init_fast<4>()); FC_TRY(init_fast<8>());
    return FC_OK;
}

bool linattn_fused_supported(int n, int C, int heads) {
    if (heads != LHEADS || (C & 3) || C > 256 || n < 1) return false;
    if (n >= 128 && (n & 127)) return false;
    return la_ctx_lds(C) <= 160 * 1024;
}

// May the apply launch close the module itself (LaArgs::gran)?  Its workgroups wait for the other tiles of their sample, so every
// workgroup of the grid must be resident at once: occupancy of the kernel at this C times the CU count >= tiles * B.  Fast path only.
template <int NQ>
static int apply_blocks_per_cu(int C) {
    int nb = 0;
    const hipError_t e = C <= 32 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, la_apply_fast_kernel<NQ, 1>, 256, la_apply_fast_lds(C, 1))
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, la_apply_fast_kernel<NQ, 2>, 256, la_apply_fast_lds(C, 2));
    return e == hipSuccess ? nb : 0;
}
bool linattn_fused_meeting_ok(int B, int n, int C) {
    static const bool off = [] { const char* e = std::getenv("FLOCODER_AMD_LA_CLOSE"); return e && std::string(e) == "0"; }();
    static const bool no_fast = std::getenv("FLOCODER_AMD_LINATTN_GENERAL") != nullptr;
    if (off || no_fast || n < 256 || (n & 127) || !(C == 8 || C == 16 || C == 32 || C == 64)) return false;
    const int T = linattn_fused_tiles(n);
    if (T != n / 128 || T > kPartPre) return false;
    if (linattn_fused_init() != FC_OK) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    const int per = C == 8 ? apply_blocks_per_cu<1>(C) : C == 16 ? apply_blocks_per_cu<2>(C) : C == 32 ? apply_blocks_per_cu<4>(C) : apply_blocks_per_cu<8>(C);
    return (long long)per * cus >= (long long)T * B;
}

int linattn_fused_launch(const LaArgs& a, hipStream_t s) {
    if (!linattn_fused_supported(a.n, a.C, a.heads)) return fail(FC_E_SHAPE, "linattn_fused: unsupported shape");
    if (a.xf.mode != 1 || a.xf.G != 1 || !a.xf.stats) return fail(FC_E_ARG, "linattn_fused: needs GroupNorm(1) statistics of x");
    if (a.gran && (!a.sync || !a.out || !a.g2 || !a.b2 || !(a.C == 8 || a.C == 16 || a.C == 32 || a.C == 64) || std::getenv("FLOCODER_AMD_LINATTN_GENERAL")))
        return fail(FC_E_ARG, "linattn_fused: the fused close needs the fast path, the arrival counters, to_out.1's parameters and the output");
    static const bool no_fast = std::getenv("FLOCODER_AMD_LINATTN_GENERAL") != nullptr;
    if (!no_fast && a.C <= 64 && (a.C & 7) == 0 && (a.C == 8 || a.C == 16 || a.C == 32 || a.C == 64)) {
        switch (a.C) {
            case 8: return launch_fast<1>(a, s);
            case 16: return launch_fast<2>(a, s);
            case 32: return launch_fast<4>(a, s);
            default: return launch_fast<8>(a, s);
        }
    }
    hipLaunchKernelGGL(la_ctx_kernel, dim3(LHEADS, a.B), dim3(256), la_ctx_lds(a.C), s, a);
    FC_HIP(hipGetLastError());
    const int T = linattn_fused_tiles(a.n);
    const float n_t = linattn_fused_nt(a.n, a.C);
    const int CT = a.C <= 32 ? 1 : a.C <= 64 ? 2 : a.C <= 128 ? 4 : 8;
    const dim3 grid(cdiv(a.n, 128), a.B);
    const size_t lds = la_apply_lds(a.C, CT);
    switch (CT) {
        case 1: hipLaunchKernelGGL(la_apply_kernel<1>, grid, dim3(256), lds, s, a, T, n_t); break;
        case 2: hipLaunchKernelGGL(la_apply_kernel<2>, grid, dim3(256), lds, s, a, T, n_t); break;
        case 4: hipLaunchKernelGGL(la_apply_kernel<4>, grid, dim3(256), lds, s, a, T, n_t); break;
        default: hipLaunchKernelGGL(la_apply_kernel<8>, grid, dim3(256), lds, s, a, T, n_t); break;
    }
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc
