// Producer/consumer implicit-GEMM convolution (the fast path; same contract and epilogue as conv_igemm.hip).
//
// Why: in the synchronous kernel one wave per SIMD does everything in series -- issue loads, wait, write LDS, barrier,
// MFMA with exposed LDS latency -- and co-resident workgroups march in lockstep, so the matrix pipe idles most of the
// time (in-kernel stamps: 128x32 tile 17.8 us for 3.8 us of MFMA; 32x32xK4 tile 30 % MFMA efficiency inside the chunk
// loop; profiles/r01_a_*).  Here a workgroup is 8 wave64 with fixed roles:
//
//   waves 0-3  CONSUMERS  own the accumulators; per Cin chunk they run the fully unrolled tap x k MFMA sequence on the
//                         stage that is ready and touch global memory only in the epilogue;
//   waves 4-7  LOADERS    fill the other stage: the weight slab [taps][CC][BN] (+ fused res_conv slab) by LDS-DMA
//                         (global_load_lds_dwordx4, 1 KiB per wave instruction, per-lane SOURCE address, out-of-range
//                         lanes read a 16-byte zero block), the input window through registers (all loads of a thread
//                         issued back to back, then GroupNorm/FiLM/SiLU applied on the way into LDS).
//
// One barrier per chunk hands stages over.  Each SIMD holds one consumer and one loader wave, so MFMA issue and the
// VALU/VMEM work of staging overlap inside a workgroup instead of relying on a lucky phase shift between workgroups.
// A loader thread's patch elements are the same pixels for every chunk (only the channel base moves): their source
// offsets live in registers and the per-(sample, channel) affine table is built once for all Cin.
#include "conv_dev.h"

namespace fc {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int WM, int WN, int WK, int MT, int NT, int CC, int NPL, int KS>
__global__ void __launch_bounds__(512) conv_pipe_kernel(const ConvDev p) {
    constexpr int BN = 32 * NT * WN, CS = CC + 1, KSTEPS = CC / 2, KPW = KSTEPS / WK, Q = CC / 4, PIXSTEP = 256 / Q, KK = KS * KS;
    static_assert(WM * WN * WK == 4, "4 consumer waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ConvArgs& a = p.a;
    float* gstat = smem + p.o_gstat;
    float2* aff = reinterpret_cast<float2*>(smem + p.o_aff);   // [TB][Cin]
    float* patch0 = smem + p.o_patch;
    float* wl0 = smem + p.o_wl;

    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
    const bool consumer = wave8 < 4;
    const int wave = wave8 & 3, ltid = tid & 255;
    const int half = lane >> 5, l31 = lane & 31;
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);

    const int bid = xcd_remap(blockIdx.x, p.nblocks);
    const int nt_i = bid % p.ntiles, mt_i = bid / p.ntiles;
    const int tx = mt_i % p.tiles_x, ty = (mt_i / p.tiles_x) % p.tiles_y, bg = mt_i / (p.tiles_x * p.tiles_y);
    const int TW = 1 << p.TWl, TH = 1 << p.THl;
    const int b0 = bg * p.TB, y0 = ty * TH, x0 = tx * TW, n0 = nt_i * BN;
    const int PW = p.PW, PHW = p.PH * p.PW;
    const int C0 = a.s0.C, C1 = a.s1.C, Cin = a.Cin, Cout = a.Cout;
    const bool has_res = a.res_out != nullptr;
    const int nchunks = p.nchunks;
    conv_stamp(p, 0);

    // ---- GroupNorm tables: moments per (sample, group), then the folded affine per (sample, channel) -- all 512 threads ----
    if (p.any_xf) {
        const int G0 = a.s0.xf.mode ? a.s0.xf.G : 0, G1 = a.s1.xf.mode ? a.s1.xf.G : 0;
        for (int i = tid; i < p.TB * (G0 + G1); i += 512) {
            const bool first = i < p.TB * G0;
            const SrcXform& xf = first ? a.s0.xf : a.s1.xf;
            const int j = first ? i : i - p.TB * G0;
            const int tb = j / xf.G, g = j - tb * xf.G, b = b0 + tb;
            float mean = 0.f, rstd = 0.f;
            if (b < a.B) {
                const float* sp = xf.stats + (size_t)(b * xf.G + g) * xf.T * 2;
                float sm = 0.f;
                for (int t = 0; t < xf.T; ++t) sm += sp[2 * t];
                mean = sm / (float)xf.T;
                float m2 = 0.f, dv = 0.f;
                for (int t = 0; t < xf.T; ++t) { const float d = sp[2 * t] - mean; m2 += sp[2 * t + 1]; dv += d * d; }
                rstd = 1.0f / sqrtf((m2 + xf.n_t * dv) / (xf.n_t * (float)xf.T) + xf.eps);
            }
            gstat[2 * i] = mean;
            gstat[2 * i + 1] = rstd;
        }
        __syncthreads();
        for (int i = tid; i < p.TB * Cin; i += 512) {
            const int tb = i / Cin, c = i - tb * Cin, b = b0 + tb;
            float A = 1.f, Bv = 0.f;
            const bool first = c < C0;
            const SrcXform& xf = first ? a.s0.xf : a.s1.xf;
            if (b < a.B && xf.mode) {
                const int cs = first ? c : c - C0, Cs = first ? C0 : C1;
                const float* gs = gstat + 2 * ((first ? 0 : p.TB * G0) + tb * xf.G + cs / (Cs / xf.G));
                A = gs[1] * xf.gamma[cs];
                Bv = xf.beta[cs] - gs[0] * A;
                if (xf.ss) {
                    const float sc = xf.ss[(size_t)b * xf.ss_stride + cs] + 1.0f;
                    A *= sc;
                    Bv = Bv * sc + xf.ss[(size_t)b * xf.ss_stride + Cs + cs];
                }
            }
            aff[i] = make_float2(A, Bv);
        }
        __syncthreads();
    }

    f32x16 acc[MT][NT], accr[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[mt][nt][r] = 0.f; accr[mt][nt][r] = 0.f; }

    if (!consumer) {
        // =========================================== LOADERS ===========================================
        const float* wbase = a.w + (size_t)b0 * a.w_batch_stride;
        const int q4 = (ltid % Q) * 4;
        int e_po[NPL], e_lds[NPL], e_tb[NPL];   // this thread's patch elements: pixel (ltid/Q + k*256/Q), channel quad ltid%Q
        {
            const int Hin = a.Hs << a.ups, Win = a.Ws << a.ups;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const int pix = ltid / Q + k * PIXSTEP;
                e_po[k] = -1; e_lds[k] = -1; e_tb[k] = 0;
                if (pix < p.P) {
                    const int tb = pix / PHW, r = pix - tb * PHW, py = r / PW, px = r - py * PW;
                    const int iy = y0 * a.stride - a.pad + py, ix = x0 * a.stride - a.pad + px, b = b0 + tb;
                    if (b < a.B && iy >= 0 && iy < Hin && ix >= 0 && ix < Win) e_po[k] = (b * a.Hs + (iy >> a.ups)) * a.Ws + (ix >> a.ups);
                    e_lds[k] = pix * CS + q4;
                    e_tb[k] = tb;
                }
            }
        }
        conv_stamp(p, 1);
        for (int i = 0; i < nchunks; ++i) {      // fill stage i while the consumers work on stage i-1
            const int c0 = i * CC;
            float* pb = patch0 + (i & 1) * p.patch_stride;
            float* wb = wl0 + (i & 1) * p.wl_stride;
            {   // weight slab: rows [0, KK*CC) conv taps, rows [KK*CC, KK*CC + CC) res_conv; BN floats per row; 1 KiB pieces
                const int rows_main = KK * CC, rows = rows_main + (has_res ? CC : 0), npieces = rows * BN / 256;
                for (int j = wave; j < npieces; j += 4) {
                    const int f = j * 256 + lane * 4, row = f / BN, n = n0 + (f % BN);
                    const float* src = p.zeros16;
                    if (n < Cout) {
                        if (row < rows_main) {
                            const int gc = c0 + (row % CC);
                            if (gc < Cin) src = wbase + ((size_t)(row / CC) * Cin + gc) * Cout + n;
                        } else {
                            const int gc = c0 + (row - rows_main);
                            if (gc < Cin) src = a.res_w + (size_t)gc * Cout + n;
                        }
                    }
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wb + j * 256), 16, 0, 0);
                }
            }
            const int c = c0 + q4;
            const bool live = c < Cin, first = c < C0, act = first ? p.act0 : p.act1;
            const float* base = first ? a.s0.p + c : a.s1.p + (c - C0);
            const int Cs = first ? C0 : C1;
            float4 v[NPL];
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live && e_po[k] >= 0) v[k] = *reinterpret_cast<const float4*>(base + (size_t)e_po[k] * Cs);
            }
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                if (e_lds[k] < 0) continue;
                float4 x = v[k];
                if (p.any_xf && live && e_po[k] >= 0) {
                    const float2* ab = aff + e_tb[k] * Cin + c;
                    x.x = ab[0].x * x.x + ab[0].y; x.y = ab[1].x * x.y + ab[1].y;
                    x.z = ab[2].x * x.z + ab[2].y; x.w = ab[3].x * x.w + ab[3].y;
                    if (act) { x.x = silu_f(x.x); x.y = silu_f(x.y); x.z = silu_f(x.z); x.w = silu_f(x.w); }
                }
                float* d = pb + e_lds[k];
                d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
            }
            if (i == 0) conv_stamp(p, 3);
            __syncthreads();    // stage i handed over (LDS-DMA drained: the barrier waits vmcnt(0)); stage i-1 is free again
        }
        __syncthreads();        // matches the consumers' barrier after the last chunk
    } else {
        // =========================================== CONSUMERS ===========================================
        int abase[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = (wm * MT + mt) * 32 + l31;
            const int tw = m & (TW - 1), th = (m >> p.TWl) & (TH - 1), tb = m >> (p.TWl + p.THl);
            abase[mt] = (tb * PHW + th * a.stride * PW + tw * a.stride) * CS + half;
        }
        const int bbase = half * BN + wn * NT * 32 + l31;
        const int kk0 = wk * KPW;
        conv_stamp(p, 1);
        __syncthreads();        // stage 0 ready
        conv_stamp(p, 4);
        for (int i = 0; i < nchunks; ++i) {
            const float* patch = patch0 + (i & 1) * p.patch_stride;
            const float* wl = wl0 + (i & 1) * p.wl_stride + bbase;
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) {
                const int tapoff = ((tap / KS) * PW + (tap % KS)) * CS;
#pragma unroll
                for (int kk = 0; kk < KPW; ++kk) {
                    const int k = 2 * (kk0 + kk);
                    float av[MT], bv[NT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) av[mt] = patch[abase[mt] + tapoff + k];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[nt] = wl[(tap * CC + k) * BN + nt * 32];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
                }
            }
            if (has_res) {   // fused res_conv: the centre tap against its own weight slab
                const int tapoff = (a.pad * PW + a.pad) * CS;
#pragma unroll
                for (int kk = 0; kk < KPW; ++kk) {
                    const int k = 2 * (kk0 + kk);
                    float av[MT], bv[NT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) av[mt] = patch[abase[mt] + tapoff + k];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[nt] = wl[(KK * CC + k) * BN + nt * 32];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            accr[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[nt], accr[mt][nt], 0, 0, 0);
                }
            }
            __syncthreads();    // stage i consumed; stage i+1 (if any) ready
        }
        conv_stamp(p, 5);
    }
    conv_epilogue<WM, WN, WK, MT, NT>(p, acc, accr, smem, tid, lane, wave, b0, y0, x0, n0, tx, ty, consumer, 512);
}

// ---------------------------------------------------------------------------------------------------
static float* g_zeros16 = nullptr;
const float* conv_zeros16() { return g_zeros16; }

#define FC_PIPE_TILES(X, KS)        \
    X(TILE_M128N32, 4, 1, 1, 1, 1, 16, KS)  \
    X(TILE_M128N64, 4, 1, 1, 1, 2, 16, KS)  \
    X(TILE_M64N32K2, 2, 1, 2, 1, 1, 32, KS) \
    X(TILE_M32N32K4, 1, 1, 4, 1, 1, 32, KS) \
    X(TILE_M64N64K2, 2, 1, 2, 1, 2, 16, KS)

template <int KS>
static int pipe_attr_ks() {
#define X(T, WM, WN, WK, MT, NT, CC, K)                                                                                   \
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pipe_kernel<WM, WN, WK, MT, NT, CC, 8, K>),            \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_PIPE_TILES(X, KS)
#undef X
    return FC_OK;
}

int conv_pipe_init() {
    static bool done = false;
    if (done) return FC_OK;
    FC_HIP(hipMalloc(reinterpret_cast<void**>(&g_zeros16), 256));
    FC_HIP(hipMemset(g_zeros16, 0, 256));
    FC_TRY(pipe_attr_ks<1>());
    FC_TRY(pipe_attr_ks<2>());
    FC_TRY(pipe_attr_ks<3>());
    FC_TRY(pipe_attr_ks<5>());
    done = true;
    return FC_OK;
}

bool conv_pipe_supports_ks(int ks) { return ks == 1 || ks == 2 || ks == 3 || ks == 5; }

template <int KS>
static int pipe_launch_ks(const ConvDev& d, int tile, int grid, size_t lds, hipStream_t s) {
    switch (tile) {
#define X(T, WM, WN, WK, MT, NT, CC, K) \
    case T: hipLaunchKernelGGL((conv_pipe_kernel<WM, WN, WK, MT, NT, CC, 8, K>), dim3(grid), dim3(512), lds, s, d); break;
        FC_PIPE_TILES(X, KS)
#undef X
        default: return fail(FC_E_ARG, "conv: bad tile id");
    }
    FC_HIP(hipGetLastError());
    return FC_OK;
}

int conv_pipe_launch(const ConvDev& d, int tile, int grid, size_t lds, hipStream_t s) {
    switch (d.a.KS) {
        case 1: return pipe_launch_ks<1>(d, tile, grid, lds, s);
        case 2: return pipe_launch_ks<2>(d, tile, grid, lds, s);
        case 3: return pipe_launch_ks<3>(d, tile, grid, lds, s);
        case 5: return pipe_launch_ks<5>(d, tile, grid, lds, s);
    }
    return fail(FC_E_SHAPE, "conv: kernel size not instantiated in the pipelined kernel");
}

}  // namespace fc
