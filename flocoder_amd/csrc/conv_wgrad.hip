// Weight gradient of a convolution on the exact-fp32 matrix pipe:
//     dW[co][ci][ky][kx] = sum over (b, y, x) of X[b][y*stride - pad + ky][x*stride - pad + kx][ci] * dY[b][y][x][co]
// One GEMM per tap with M = ci, N = co and the output pixels as the reduction axis.  A workgroup owns a 32(ci) x 32(co) block
// of every tap and walks a strided share of the pixel tiles (the forward kernel's tile geometry: TB samples x TH x TW pixels,
// patch with halo staged once in LDS, taps as constant LDS offsets); its four waves split the taps (or, for 1x1, the pixels).
// Partial sums of the `nsplit` workgroups that share a block go to a workspace and are added in a fixed order by a second
// kernel -- no atomics, so gradients are bit-reproducible.  The result is written in the reference's own [O][I][KH][KW] layout
// (for Downsample's space-to-depth conv that is exactly its [O][4*I][1][1] weight, unet.py:49-54), the bias gradient rides along.
#include "common.h"

namespace fc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CS = 33;

// (split, cicoc): which share of the pixel tiles and which 32 x 32 block of every tap this workgroup owns; dw / db: where the
// un-split result goes (the launch's own pointers, or offsets into the caller's flat gradient vector for a table-driven launch)
template <int KS>
__device__ __forceinline__ void wgrad_body(const WgradDev& p, const int split, const int cicoc, float* dw, float* db) {
    constexpr int KK = KS * KS, TPW = KS == 1 ? 1 : (KK + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ys = smem + p.o_ys;
    int* pixbase = reinterpret_cast<int*>(smem + p.o_pix);
    const WgradArgs& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int cic = cicoc % p.nci, coc = cicoc / p.nci;
    // which wave takes which taps rotates with the workgroup: 9 taps over 4 waves leaves one wave with three, and wave w of every
    // resident workgroup shares SIMD w -- unrotated, SIMD 0 carried 4/3 of the average matrix work of the whole chip
    const int tapw = (wave + split + cicoc) & 3;
    const int ci0 = cic * 32, co0 = coc * 32;
    const int TW = 1 << p.TWl, TH = 1 << p.THl, BM = p.BM;
    const int Hin = a.ups ? 2 * a.Hs : a.Hs, Win = a.ups ? 2 * a.Ws : a.Ws;
    if (tid < BM) {
        const int tw = tid & (TW - 1), th = (tid >> p.TWl) & (TH - 1), tb = tid >> (p.TWl + p.THl);
        pixbase[tid] = (tb * p.PH * p.PW + th * a.stride * p.PW + tw * a.stride) * CS;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    const int phw = p.PH * p.PW;
    for (int tile = split; tile < p.mtiles; tile += p.nsplit) {
        const int txi = tile % p.tiles_x, tyi = (tile / p.tiles_x) % p.tiles_y, tbi = tile / (p.tiles_x * p.tiles_y);
        const int b0 = tbi * p.TB, y0 = tyi * TH, x0 = txi * TW;
        __syncthreads();
        for (int idx = tid; idx < p.P * 8; idx += 256) {
            const int pix = idx >> 3, q = idx & 7, c = ci0 + q * 4;
            const int tb = pix / phw, rem = pix - tb * phw, py = rem / p.PW, px = rem - py * p.PW;
            const int iy = y0 * a.stride - a.pad + py, ix = x0 * a.stride - a.pad + px, b = b0 + tb;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < a.B && c < a.Cin && iy >= 0 && iy < Hin && ix >= 0 && ix < Win) {
                const int sy = a.ups ? iy >> 1 : iy, sx = a.ups ? ix >> 1 : ix;
                const size_t pixoff = ((size_t)b * a.Hs + sy) * a.Ws + sx;
                const float* src = c < a.C0 ? a.x0 + pixoff * a.C0 + c : a.x1 + pixoff * a.C1 + (c - a.C0);
                v = *reinterpret_cast<const float4*>(src);
            }
            float* d = xs + pix * CS + q * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        for (int idx = tid; idx < BM * 8; idx += 256) {
            const int m = idx >> 3, q = idx & 7, co = co0 + q * 4;
            const int tw = m & (TW - 1), th = (m >> p.TWl) & (TH - 1), tb = m >> (p.TWl + p.THl);
            const int b = b0 + tb;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < a.B && co < a.Cout) v = *reinterpret_cast<const float4*>(a.dy + (((size_t)b * a.H + y0 + th) * a.W + x0 + tw) * a.Cout + co);
            float* d = ys + m * CS + q * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        // bias gradient: every thread folds its share of the tile's rows (row group = tid / 32), the eight groups meet once after the
        // tile loop -- as one 32-lane serial sweep over the BM rows this was 128 dependent LDS reads per tile on the critical path
        if (db && cic == 0) {
            const int rg = tid >> 5;
            for (int m = rg; m < BM; m += 8) bsum += ys[m * CS + l31];
        }
        if (KS == 1) {
            const int per = BM / 8;                      // k-steps (pixel pairs) per wave
            for (int j = wave * per; j < (wave + 1) * per; ++j) {
                const int pk = 2 * j + half;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[pixbase[pk] + l31], ys[pk * CS + l31], acc[0], 0, 0, 0);
            }
        } else {
            // four k-steps per trip with every LDS operand requested before the first MFMA: rolled one step at a time the loop was a
            // chain of dependent LDS reads (pixbase -> patch address -> operand) in front of each group of MFMAs
            int tapoff[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int tap = tapw + 4 * t;
                tapoff[t] = tap < KK ? ((tap / KS) * p.PW + (tap % KS)) * CS + l31 : -1;
            }
            for (int j0 = 0; j0 < BM / 2; j0 += 4) {
                int pb[4];
                float bv[4], av[4][TPW];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pk = 2 * (j0 + u) + half;
                    pb[u] = pixbase[pk];
                    bv[u] = ys[pk * CS + l31];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < TPW; ++t) av[u][t] = tapoff[t] >= 0 ? xs[pb[u] + tapoff[t]] : 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < TPW; ++t)
                        if (tapoff[t] >= 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][t], bv[u], acc[t], 0, 0, 0);
            }
        }
    }
    if (db && cic == 0) {   // the eight row groups of the bias gradient: wave halves by shuffle, waves through LDS
        bsum += __shfl_xor(bsum, 32);
        __syncthreads();
        if (half == 0) xs[wave * 32 + l31] = bsum;
        __syncthreads();
        if (tid < 32) bsum = (xs[tid] + xs[32 + tid]) + (xs[64 + tid] + xs[96 + tid]);
    }
    if (KS == 1) {    // the four waves hold partial sums over different pixels of the same block
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xs[((wave - 1) * 16 + r) * 64 + lane] = acc[0][r];
        }
        __syncthreads();
        if (wave == 0) {
            for (int w = 0; w < 3; ++w)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += xs[(w * 16 + r) * 64 + lane];
        }
    }
    float* dst = p.nsplit > 1 ? a.ws + (size_t)split * p.part_stride : dw;
    const int co = co0 + l31;
    if (co < a.Cout) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tap = KS == 1 ? 0 : tapw + 4 * t;
            if (tap < KK && (KS != 1 || wave == 0)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    // partials are kept [tap][ci][co] (the 32 lanes of a store are 32 consecutive floats); the reduction writes the
                    // reference's [co][ci][tap] order.  In that order a store instruction touched 64 different cache lines.
                    if (ci < a.Cin) dst[p.nsplit > 1 ? ((size_t)tap * a.Cin + ci) * a.Cout + co : ((size_t)co * a.Cin + ci) * KK + tap] = acc[t][r];
                }
            }
        }
    }
    if (db && cic == 0 && tid < 32 && co0 + tid < a.Cout) {
        float* bd = p.nsplit > 1 ? a.ws + (size_t)split * p.part_stride + (size_t)a.Cout * a.Cin * KK : db;
        bd[co0 + tid] = bsum;
    }
}

// 1x1, stride 1: dW[co][ci] = sum over pixels of X[pix][ci] dY[pix][co] needs no window -- a k-step is two pixels, and a lane's MFMA
// operands are X[pix][ci0 + lane & 31] and dY[pix][co0 + lane & 31] for pix = 2j + (lane >> 5): two coalesced 128-byte rows per
// operand straight from global memory, eight k-steps in flight per wave.  The LDS-staged form above spent its time staging 128-pixel
// tiles for 16 MFMAs per wave (0.41 ms of the flowers-sized training step in one table launch).
__device__ __forceinline__ void wgrad1x1_body(const WgradDev& p, const int split, const int cicoc, float* dw, float* db) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WgradArgs& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int cic = cicoc % p.nci, coc = cicoc / p.nci;
    const int ci = cic * 32 + l31, co = coc * 32 + l31;
    const long NP = (long)a.B * a.H * a.W;
    const bool ciok = ci < a.Cin, cook = co < a.Cout, first = ci < a.C0;
    const float* xb = !ciok ? nullptr : (first ? a.x0 + ci : a.x1 + (ci - a.C0));
    const long xs = first ? a.C0 : a.C1;
    const float* yb = cook ? a.dy + co : nullptr;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    const int BM = p.BM;
    constexpr int UB = 8;
    for (int tile = split; tile < p.mtiles; tile += p.nsplit) {
        const long p0 = (long)tile * BM;
        for (int j0 = wave; j0 < BM / 2; j0 += 4 * UB) {
            float xv[UB], yv[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const long pix = p0 + 2 * (j0 + 4 * u) + half;
                const bool in = (j0 + 4 * u) < BM / 2 && pix < NP;
                xv[u] = (in && ciok) ? xb[pix * xs] : 0.f;
                yv[u] = (in && cook) ? yb[pix * a.Cout] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (j0 + 4 * u >= BM / 2) break;           // wave-uniform
                bsum += yv[u];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[u], yv[u], acc, 0, 0, 0);
            }
        }
    }
    float* red = smem;                                      // 3 x 16 x 64 floats, then 4 x 32 for the bias
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    bsum += __shfl_xor(bsum, 32);
    if (half == 0) red[3 * 16 * 64 + wave * 32 + l31] = bsum;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = ((acc[r] + red[r * 64 + lane]) + red[(16 + r) * 64 + lane]) + red[(32 + r) * 64 + lane];
        float* dst = p.nsplit > 1 ? a.ws + (size_t)split * p.part_stride : dw;
        if (cook) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = cic * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (c < a.Cin) dst[p.nsplit > 1 ? (size_t)c * a.Cout + co : (size_t)co * a.Cin + c] = acc[r];
            }
        }
        if (db && cic == 0 && half == 0 && cook) {
            const float* br = red + 3 * 16 * 64;
            float* bd = p.nsplit > 1 ? a.ws + (size_t)split * p.part_stride + (size_t)a.Cout * a.Cin : db;
            bd[co] = (br[l31] + br[32 + l31]) + (br[64 + l31] + br[96 + l31]);
        }
    }
}

template <int KS>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(const WgradDev p) {
    if (KS == 1 && p.direct1) { wgrad1x1_body(p, blockIdx.x, blockIdx.y, p.a.dw, p.a.db); return; }
    wgrad_body<KS>(p, blockIdx.x, blockIdx.y, p.a.dw, p.a.db);
}

// Every weight gradient of one kernel size in ONE launch: block -> (job, split + nsplit * cicoc) through a table.  At the training
// shapes a single layer's launch is a few dozen workgroups and ~10-20 us of latency (staging -> MFMA -> store); 70 of them in a row
// were a quarter of the step, side by side they fill the chip once.
template <int KS>
__global__ void __launch_bounds__(256) conv_wgrad_table_kernel(const WgradDev* __restrict__ jobs, const int2* __restrict__ blocks, float* grads) {
    const int2 bj = blocks[blockIdx.x];
    const WgradDev p = jobs[bj.x];
    if (KS == 1 && p.direct1) { wgrad1x1_body(p, bj.y % p.nsplit, bj.y / p.nsplit, grads + p.dw_off, p.db_off >= 0 ? grads + p.db_off : nullptr); return; }
    wgrad_body<KS>(p, bj.y % p.nsplit, bj.y / p.nsplit, grads + p.dw_off, p.db_off >= 0 ? grads + p.db_off : nullptr);
}

// element e of a partial vector ([tap][ci][co] | bias) -> its place in the [co][ci][tap] | bias result
__device__ __forceinline__ size_t wgrad_final_index(size_t e, size_t nw, int cout, int cin, int kk) {
    if (e >= nw) return e;
    const int co = (int)(e % cout), ci = (int)((e / cout) % cin), tap = (int)(e / ((size_t)cout * cin));
    return ((size_t)co * cin + ci) * kk + tap;
}

// 256 threads = 64 elements x 4 split lanes (fixed assignment and fixed combine order: bit-reproducible)
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* ws, int nsplit, size_t stride, float* dw, size_t nw, float* db, int nb, int cin,
                                                           int kk) {
    __shared__ float part[4][64];
    const size_t total = nw + (db ? nb : 0);
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    for (size_t e0 = (size_t)blockIdx.x * 64; e0 < total; e0 += (size_t)gridDim.x * 64) {
        const size_t e = e0 + el;
        float acc = 0.f;
        if (e < total)
            for (int s = sl; s < nsplit; s += 4) acc += ws[(size_t)s * stride + e];
        __syncthreads();
        part[sl][el] = acc;
        __syncthreads();
        if (sl == 0 && e < total) {
            const float v = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
            if (e < nw) dw[wgrad_final_index(e, nw, nb, cin, kk)] = v;
            else db[e - nw] = v;
        }
    }
}

// block (job, chunk): 64 consecutive elements of one job's partial vector, 4 split lanes each (same fixed order as wgrad_reduce_kernel)
__global__ void __launch_bounds__(256) wgrad_reduce_table_kernel(const WredJob* jobs, const int2* blocks, float* grads) {
    __shared__ float part[4][64];
    const int2 bj = blocks[blockIdx.x];
    const WredJob j = jobs[bj.x];
    const size_t total = j.nw + (j.db >= 0 ? (size_t)j.nb : 0), e = (size_t)bj.y * 64 + (threadIdx.x & 63);
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    float acc = 0.f;
    if (e < total)
        for (int s = sl; s < j.nsplit; s += 4) acc += j.ws[(size_t)s * j.stride + e];
    part[sl][el] = acc;
    __syncthreads();
    if (sl == 0 && e < total) {
        const float v = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
        if (e < j.nw) grads[j.dw + wgrad_final_index(e, j.nw, j.nb, j.cin, j.kk)] = v;
        else grads[j.db + (e - j.nw)] = v;
    }
}
int wgrad_reduce_table_launch(const WredJob* jobs_dev, const int2* blocks_dev, int nblocks, float* grads, hipStream_t s) {
    if (!nblocks) return FC_OK;
    hipLaunchKernelGGL(wgrad_reduce_table_kernel, dim3(nblocks), dim3(256), 0, s, jobs_dev, blocks_dev, grads);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

static int wgrad_geometry(const WgradArgs& a, WgradDev* d) {
    if (!is_pow2(a.H) || !is_pow2(a.W)) return fail(FC_E_SHAPE, "wgrad: H and W must be powers of two");
    if ((a.C0 & 3) || (a.C1 & 3) || (a.Cout & 3) || a.C0 + a.C1 != a.Cin) return fail(FC_E_SHAPE, "wgrad: channel counts must be multiples of 4");
    if (a.KS != 1 && a.KS != 2 && a.KS != 3 && a.KS != 5) return fail(FC_E_SHAPE, "wgrad: kernel size not instantiated");
    WgradDev& p = *d;
    p.a = a;
    for (p.BM = 128;; p.BM = 32) {
        const int TW = a.W < 16 ? a.W : 16;
        int TH = p.BM / TW;
        if (TH > a.H) TH = a.H;
        p.TB = p.BM / (TH * TW);
        p.TWl = ilog2(TW); p.THl = ilog2(TH);
        p.PH = TH * a.stride + a.KS - a.stride;
        p.PW = TW * a.stride + a.KS - a.stride;
        p.P = p.TB * p.PH * p.PW;
        p.tiles_x = a.W / TW; p.tiles_y = a.H / TH;
        if (p.P <= 640 || p.BM == 32) break;
    }
    if (p.P > 1024) return fail(FC_E_SHAPE, "wgrad: patch does not fit in LDS");
    p.mtiles = cdiv(a.B, p.TB) * p.tiles_x * p.tiles_y;
    p.nci = cdiv(a.Cin, 32); p.nco = cdiv(a.Cout, 32);
    int ns = cdiv(a.split_target > 0 ? a.split_target : 1024, p.nci * p.nco);
    if (ns > 256) ns = 256;
    if (ns > p.mtiles) ns = p.mtiles;
    p.part_stride = ((size_t)a.Cout * a.Cin * a.KS * a.KS + a.Cout + 3) & ~(size_t)3;
    if (ns > 1 && a.ws_floats) {
        const size_t fit = a.ws_floats / p.part_stride;
        if (fit < 2) ns = 1; else if ((size_t)ns > fit) ns = (int)fit;
    }
    p.nsplit = ns < 1 ? 1 : ns;
    const int xs_floats = p.P * CS > 3 * 1024 ? p.P * CS : 3 * 1024;
    static const bool no_direct = std::getenv("FLOCODER_AMD_WGRAD_1X1_STAGED") != nullptr;
    p.direct1 = (!no_direct && a.KS == 1 && a.stride == 1 && !a.ups && a.pad == 0 && a.Hs == a.H && a.Ws == a.W) ? 1 : 0;
    p.o_ys = (xs_floats + 3) & ~3;
    p.o_pix = (p.o_ys + p.BM * CS + 3) & ~3;
    return FC_OK;
}

size_t conv_wgrad_workspace(const WgradArgs& a) {   // floats wanted for the preferred split
    WgradArgs b = a;
    b.ws_floats = 0;
    WgradDev d;
    if (wgrad_geometry(b, &d) != FC_OK) return 0;
    return d.nsplit > 1 ? (size_t)d.nsplit * d.part_stride : 0;
}

int conv_wgrad_split(const WgradArgs& a, int* nsplit, size_t* part_stride) {
    WgradArgs b = a;
    b.ws_floats = 0;                         // the preferred split, not one squeezed into a shared workspace
    WgradDev d;
    FC_TRY(wgrad_geometry(b, &d));
    *nsplit = d.nsplit; *part_stride = d.part_stride;
    return FC_OK;
}

static int wgrad_launch_impl(const WgradArgs& a, bool reduce, hipStream_t s);
int conv_wgrad_launch(const WgradArgs& a, hipStream_t s) { return wgrad_launch_impl(a, true, s); }
int conv_wgrad_launch_noreduce(const WgradArgs& a, hipStream_t s) { return wgrad_launch_impl(a, false, s); }

static int wgrad_launch_impl(const WgradArgs& a, bool reduce, hipStream_t s) {
    WgradDev d;
    FC_TRY(wgrad_geometry(a, &d));
    if (d.nsplit > 1 && !a.ws) return fail(FC_E_ARG, "wgrad: workspace missing");
    const size_t lds = (size_t)(d.o_pix + 128) * sizeof(float);
    const dim3 grid(d.nsplit, d.nci * d.nco);
    switch (a.KS) {
        case 1: hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), lds, s, d); break;
        case 2: hipLaunchKernelGGL(conv_wgrad_kernel<2>, grid, dim3(256), lds, s, d); break;
        case 3: hipLaunchKernelGGL(conv_wgrad_kernel<3>, grid, dim3(256), lds, s, d); break;
        default: hipLaunchKernelGGL(conv_wgrad_kernel<5>, grid, dim3(256), lds, s, d); break;
    }
    FC_HIP(hipGetLastError());
    if (d.nsplit > 1 && reduce) {
        const size_t nw = (size_t)a.Cout * a.Cin * a.KS * a.KS;
        const size_t total = nw + (a.db ? a.Cout : 0);
        const int g = (int)((total + 63) / 64 < 4096 ? (total + 63) / 64 : 4096);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(g), dim3(256), 0, s, a.ws, d.nsplit, d.part_stride, a.dw, nw, a.db, a.Cout, a.Cin, a.KS * a.KS);
        FC_HIP(hipGetLastError());
    }
    return FC_OK;
}

int conv_wgrad_table_entry(const WgradArgs& a, int64_t dw_off, int64_t db_off, WgradDev* out, int* nblocks, size_t* lds_bytes) {
    WgradArgs b = a;
    b.dw = nullptr; b.db = nullptr;
    if (b.ws) b.ws_floats = 0;                // the entry's workspace was sized for its preferred split
    FC_TRY(wgrad_geometry(b, out));
    if (out->nsplit > 1 && !a.ws) return fail(FC_E_ARG, "wgrad table: a split entry needs a workspace of its own");
    out->dw_off = dw_off; out->db_off = db_off;
    *nblocks = out->nsplit * out->nci * out->nco;
    *lds_bytes = (size_t)(out->o_pix + 128) * sizeof(float);
    return FC_OK;
}

int conv_wgrad_table_launch(int KS, const WgradDev* jobs_dev, const int2* blocks_dev, int nblocks, size_t lds_bytes, float* grads, hipStream_t s) {
    if (!nblocks) return FC_OK;
    const dim3 grid(nblocks);
    switch (KS) {
        case 1: hipLaunchKernelGGL(conv_wgrad_table_kernel<1>, grid, dim3(256), lds_bytes, s, jobs_dev, blocks_dev, grads); break;
        case 2: hipLaunchKernelGGL(conv_wgrad_table_kernel<2>, grid, dim3(256), lds_bytes, s, jobs_dev, blocks_dev, grads); break;
        case 3: hipLaunchKernelGGL(conv_wgrad_table_kernel<3>, grid, dim3(256), lds_bytes, s, jobs_dev, blocks_dev, grads); break;
        case 5: hipLaunchKernelGGL(conv_wgrad_table_kernel<5>, grid, dim3(256), lds_bytes, s, jobs_dev, blocks_dev, grads); break;
        default: return fail(FC_E_SHAPE, "wgrad table: kernel size not instantiated");
    }
    FC_HIP(hipGetLastError());
    return FC_OK;
}

int conv_wgrad_init() {
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_table_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_table_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_table_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_table_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // > 64 KB of dynamic LDS needs the attribute on gfx950 too
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return FC_OK;
}

}  // namespace fc
