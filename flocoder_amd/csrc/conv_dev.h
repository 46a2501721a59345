// Shared between the two implicit-GEMM kernels (conv_igemm.hip: synchronous chunk loop, the fallback;
// conv_pipe.hip: software-pipelined chunk loop, the fast path).
#pragma once
#include "common.h"

namespace fc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) unsigned long long gu64;   // global-address-space words for in-launch hand-offs (never flat)
typedef __attribute__((address_space(1))) unsigned gu32;

struct ConvDev {
    ConvArgs a;
    int TWl, THl, TB, PH, PW, P;
    int tiles_x, tiles_y, ntiles, nblocks;
    int cpg, cpgt, NPG, rps;     // output-stats geometry
    int gsz, o_fin;              // fused tail: workgroups per sample group, LDS offset of the (mean, rstd) table
    int fin_local;               // fused tail: the tile holds whole GroupNorm groups -> statistics straight from LDS, nothing to wait for
    int act0, act1, any_xf;
    int o_pixoff, o_pixtb, o_gstat, o_aff, o_patch, o_wl, o_wres, o_red, o_part;
    int txl, tyl;                // log2 of tiles_x / tiles_y (both powers of two)
    unsigned magic_nt;           // ceil(2^32 / ntiles) (0: ntiles == 1, or the grid is too large for the 16-bit fast path: divide)
    int loader_prio;             // s_setprio level of the loader waves (the younger half of the workgroup loses issue arbitration to the MFMA waves otherwise)
    int o_out;                   // LDS image of the output tile for the wide (16-byte) stores, or -1: per-lane dword stores
    int o_epoch, o_gran;         // fused tail: LDS word holding this launch's epoch (outside the aliased region); gathered partials
    // pipelined kernel only
    const float* zeros16;        // 16 zero bytes in global memory: source of out-of-range LDS-DMA lanes
    int patch_stride, wl_stride; // floats between the two pipeline stages of each buffer
    int nchunks, nwb;            // Cin chunks; weight stages resident in LDS (2 or 3)
    unsigned magic_phw, magic_pw;  // ceil(2^32 / PH*PW), ceil(2^32 / PW): x / d == umulhi(x, magic) for the small x used here
    unsigned long long* stamps;  // diagnostic builds only: [block][8 waves][16] s_memtime samples, or null
    int bf3;                     // the launch runs on the split-bf16 instantiation (ConvArgs::prec == 1 and the tile / kernel size have one)
};

// Phase stamp (diagnostics; null pointer = one scalar branch).  Lane 0 of every wave records the shader clock.
__device__ __forceinline__ void conv_stamp(const ConvDev& p, int slot) {
    if (p.stamps) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 8) {
            p.stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + slot] = t;
            // slot 15: the device-wide 100 MHz clock at the wave's first stamp (the shader clock is not comparable between workgroups)
            if (slot == 0) p.stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + 15] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

struct TileInfo { int BM, BN, CC, WK, MTNT, WMWN; };

// x / d for 0 <= x < 2^16 and 1 < d < 2^16 with magic = floor(2^32 / d) + 1 (exact in that range); magic == 0 encodes d == 1
__device__ __forceinline__ int fastdiv(int x, unsigned magic) { return magic ? (int)__umulhi((unsigned)x, magic) : x; }

__device__ __forceinline__ float silu_f(float z) { return z / (1.0f + __expf(-z)); }
// workgroup barrier that orders LDS traffic only (no wait for outstanding global stores)
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// bijective XCD remap (cdna_hip_programming.md 5 "XCD swizzle must be bijective"): blocks that share an
// XCD (bid % 8) get a contiguous run of tile ids, so n-tiles of one m-tile and neighbouring m-tiles hit
// the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
// ConvArgs::par4: the quarter of the (remapped) grid a workgroup falls into is its parity class; returns the id inside that quarter
__device__ __forceinline__ int conv_parity_select(ConvDev& p, int bid) {
    if (!p.a.par4) return bid;
    const int nb1 = p.nblocks >> 2, par = bid >= 2 * nb1 ? (bid >= 3 * nb1 ? 3 : 2) : (bid >= nb1 ? 1 : 0);
    p.a.w += (size_t)par * 4 * p.a.Cin * p.a.Cout;
    if (p.a.w_b3) p.a.w_b3 += (size_t)par * 4 * (16 * p.nchunks) * p.a.Cout;      // 4 taps x Ipad x Cout floats per class (hi + lo halves)
    p.a.pad_y = 1 - (par >> 1); p.a.pad_x = 1 - (par & 1);
    p.a.out_oy = par >> 1; p.a.out_ox = par & 1;
    p.a.stats_toff = par;
    return bid - par * nb1;
}


// The kernel's parameter block, read through the VECTOR memory path: every lane loads one dword of the kernarg segment (three
// coalesced loads in flight at once), fields are then picked out with v_readlane.  Taken as `const ConvDev p` the compiler fetches the
// ~130 dwords with scalar loads, runs out of SGPRs, and the prologue becomes two dozen dependent s_load -> s_waitcnt round trips with
// v_writelane spills in between (1100 instructions before the first global load in the ISA of the M128N32 instantiation: 2-3 k cycles
// of every launch).  Pointers are re-tagged as global so the loads / stores through them stay global_* (not flat_*).
template <class T>
__device__ __forceinline__ T* as_global(T* q) {
    typedef __attribute__((address_space(1))) T* gp;
    return (T*)(gp)(size_t)q;
}
__device__ __forceinline__ void conv_params_from_lanes(ConvDev& p) {
    constexpr int ND = (int)(sizeof(ConvDev) / 4);
    static_assert(sizeof(ConvDev) % 4 == 0 && ND <= 256, "ConvDev must fit four lane-loads");
    const unsigned* ka = (const unsigned*)as_global((const unsigned*)__builtin_amdgcn_kernarg_segment_ptr());
    const int lane = threadIdx.x & 63;
    unsigned w[(ND + 63) / 64];
#pragma unroll
    for (int j = 0; j < (ND + 63) / 64; ++j) w[j] = (64 * j + lane < ND) ? ka[64 * j + lane] : 0u;
    unsigned* d = reinterpret_cast<unsigned*>(&p);
#pragma unroll
    for (int i = 0; i < ND; ++i) d[i] = __builtin_amdgcn_readlane(w[i / 64], i % 64);
    ConvArgs& a = p.a;
    a.s0.p = as_global(a.s0.p); a.s0.xf.stats = as_global(a.s0.xf.stats); a.s0.xf.gamma = as_global(a.s0.xf.gamma);
    a.s0.xf.beta = as_global(a.s0.xf.beta); a.s0.xf.ss = as_global(a.s0.xf.ss);
    a.s1.p = as_global(a.s1.p); a.s1.xf.stats = as_global(a.s1.xf.stats); a.s1.xf.gamma = as_global(a.s1.xf.gamma);
    a.s1.xf.beta = as_global(a.s1.xf.beta); a.s1.xf.ss = as_global(a.s1.xf.ss);
    a.w4 = as_global(a.w4); a.res_w4 = as_global(a.res_w4);
    a.w = as_global(a.w); a.bias = as_global(a.bias); a.out = as_global(a.out); a.add = as_global(a.add);
    a.stats_out = as_global(a.stats_out); a.res_w = as_global(a.res_w); a.res_b = as_global(a.res_b); a.res_out = as_global(a.res_out);
    a.fin.gamma = as_global(a.fin.gamma); a.fin.beta = as_global(a.fin.beta); a.fin.res = as_global(a.fin.res);
    a.fin.gn1_out = as_global(a.fin.gn1_out); a.fin.raw = as_global(a.fin.raw); a.fin.sync = as_global(a.fin.sync); a.fin.gran = as_global(a.fin.gran); a.fin.err = as_global(a.fin.err);
    p.zeros16 = as_global(p.zeros16); p.stamps = as_global(p.stamps);
}

// Shared epilogue: K-split reduction through LDS, + bias, GroupNorm partials, optional SiLU / residual, stores.
// FL: what this instantiation can do, as a bit mask -- 1 fused Block tail, 2 fused res_conv output, 4 statistics of the final value.
// A launch goes to the smallest flavour that covers it (conv_pipe.hip).  Not a matter of taste: a path that is compiled in but never
// taken still costs -- the taken path then jumps over it, every jump lands on a cold instruction-cache line (the cache is invalidated
// at every launch), and a launch is short.  Measured: with the tail code merely compiled OUT the un-fused sampler ran 2.9 % faster.
constexpr int FL_FIN = 1, FL_RES = 2, FL_POST = 4, FL_XF = 8, FL_CAT = 16, FL_STAMP = 32, FL_STATS = 64, FL_GN1 = 128, FL_POSTOP = 256,
              FL_NARROW = 512, FL_MULTI = 1024, FL_MEET = 2048, FL_ALL = 4095, FL_W4 = 4096;
// (4096, outside FL_ALL: the 32-row tile at 3x3 reads its weights from the k-step-quad copy -- ConvArgs::w4 -- 16 bytes per lane)
// (8: the input may carry a GroupNorm / FiLM / SiLU transform, 16: a second, concatenated source, 32: diagnostic phase stamps, 64: GroupNorm
// partials of the output, 128: GroupNorm(1) partials of the tail's result, 256: activation / addend on the output, 512: per-lane dword
// stores when the LDS image for the wide stores does not fit, 1024: several samples per tile -- always on for the 32-row tile,
// 2048: the fused tail may have to meet other workgroups (without it: only tails whose tile holds whole GroupNorm groups))
template <int WM, int WN, int WK, int MT, int NT, int FL = FL_ALL>
__device__ __forceinline__ void conv_epilogue(const ConvDev& p, f32x16 (&acc)[MT][NT], f32x16 (&accr)[MT][NT], float* smem, int tid, int lane,
                                              int wave, int b0, int y0, int x0, int n0, int tx, int ty, bool active = true,
                                              int nthr = 256, const float* pre = nullptr, bool staged_dead = false) {
    // `pre` (optional, 4*NT floats in the caller's registers): bias | res_conv bias | tail gamma | tail beta of this lane's columns,
    // requested before the main loop -- loaded here they are a cold miss on the epilogue's critical path (~2 k cycles per launch).
    // `active` = this wave holds accumulators (false for the loader waves of the producer/consumer kernel, which only
    // take part in the barriers and the statistics reduction); `nthr` = threads in the workgroup.
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    const ConvArgs& a = p.a;
    const int half = lane >> 5, l31 = lane & 31;
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    const int TW = 1 << p.TWl, TH = 1 << p.THl, Cout = a.Cout;
    // a lean flavour is only ever launched for an exact match of its mask (conv_pipe.hip), so there a set bit means "on", not "possible"
    constexpr bool LEAN = FL != FL_ALL;
    const bool has_res = LEAN ? bool(FL & FL_RES) : a.res_out != nullptr;
    const bool stats_post = (FL & FL_POST) && a.stats_post;
    // `staged_dead`: the caller's main loop ended with a workgroup barrier (the pipelined kernel hands every stage over through one), so
    // the staging buffers the epilogue's scratch aliases are dead already and the barriers that only said so are skipped
    if (WK > 1) {  // meet the K-split partials in LDS (patch/wl are dead now)
        if (!staged_dead) __syncthreads();
        float* red = smem + p.o_red;
        constexpr int TILE = 16 * 64;
        const int slot = ((wm * WN + wn) * (WK - 1) + (wk - 1)) * MT * NT * (has_res ? 2 : 1);
        if (active && wk > 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float* d = red + (size_t)(slot + mt * NT + nt) * TILE + lane;
#pragma unroll
                    for (int r = 0; r < 16; ++r) d[r * 64] = acc[mt][nt][r];
                    if (has_res) {
                        float* dr = red + (size_t)(slot + MT * NT + mt * NT + nt) * TILE + lane;
#pragma unroll
                        for (int r = 0; r < 16; ++r) dr[r * 64] = accr[mt][nt][r];
                    }
                }
        }
        __syncthreads();
        if (active && wk == 0) {
            for (int k2 = 1; k2 < WK; ++k2) {
                const int sl = ((wm * WN + wn) * (WK - 1) + (k2 - 1)) * MT * NT * (has_res ? 2 : 1);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float* d = red + (size_t)(sl + mt * NT + nt) * TILE + lane;
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][nt][r] += d[r * 64];
                        if (has_res) {
                            const float* dr = red + (size_t)(sl + MT * NT + mt * NT + nt) * TILE + lane;
#pragma unroll
                            for (int r = 0; r < 16; ++r) accr[mt][nt][r] += dr[r * 64];
                        }
                    }
            }
        }
    }

    // pixel index (b * H + y) * W + x of accumulator row r of M tile mt in this lane, or -1 for a sample beyond B
    auto pix_of = [&](int mt, int r) {
        const int m = (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int tw = m & (TW - 1), th = (m >> p.TWl) & (TH - 1), tb = m >> (p.TWl + p.THl);
        const int b = b0 + tb;
        return b < a.B ? (b * (a.H << a.out_sh) + ((y0 + th) << a.out_sh) + a.out_oy) * (a.W << a.out_sh) + ((x0 + tw) << a.out_sh) + a.out_ox : -1;
    };
    if (FL & FL_STAMP) conv_stamp(p, 6);
    const bool owner = active && (wk == 0);
    const bool fin = LEAN ? bool(FL & FL_FIN) : a.fin.gamma != nullptr;
    // this launch's epoch of the fused tail (drawn from the sample group's arrival counter at kernel start, parked in LDS)
    const bool meeting = LEAN ? bool(FL & FL_MEET) : (fin && !p.fin_local);
    const unsigned epoch = meeting ? __float_as_uint(smem[p.o_epoch]) : 0u;
    float* partS = smem + p.o_part;               // [BM/16][BN]
    float* partQ = partS + (BM / 16) * BN;        // [BM/16][BN]
    const bool do_stats = LEAN ? bool(FL & FL_STATS) : a.stats_out != nullptr;
    if (do_stats && !staged_dead) __syncthreads();   // patch/wl (aliased by part*) are dead for every wave

    // per-(16-row half-block, column) sums of the accumulators -> LDS
    auto block_sums = [&]() {
        if (owner) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int ncol = (wn * NT + nt) * 32 + l31;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        float s = 0.f, q = 0.f;
#pragma unroll
                        for (int r = 8 * hh; r < 8 * hh + 8; ++r) { const float v = acc[mt][nt][r]; s += v; q += v * v; }
                        s += __shfl_xor(s, 32);
                        q += __shfl_xor(q, 32);
                        if (half == 0) {
                            const int hb = (wm * MT + mt) * 2 + hh;
                            partS[hb * BN + ncol] = s;
                            partQ[hb * BN + ncol] = q;
                        }
                    }
                }
        }
    };
    // thread i <-> (sample tb, column col) sums its column over the sample's 16-row half-blocks, then the cpgt columns of a
    // group -- consecutive lanes of one wave, cpgt a power of two <= 64 -- meet by xor-shuffles; the group's first lane writes
    // (mean_t, M2_t) of this tile's share of group g of sample b.
    // (sum, sum of squares) of this tile's share of group (tb, gl) -> its (mean_t, M2_t) partial slot (or granules / the local table)
    auto publish = [&](float* dst, int G, int cpg, int cpgt, int NPG, bool coherent, bool ltab_on, int tb, int gl, float s, float q) {
        const int b = b0 + tb;
        const float n = (float)(p.rps * cpgt), mean = s / n;
        const int g = n0 / cpg + (cpg >= BN ? 0 : gl);
        const int nsub = (cpg >= BN) ? (n0 % cpg) / BN : 0;
        const int msub = (p.TB > 1) ? 0 : ty * p.tiles_x + tx;
        const int T = (p.TB > 1 ? 1 : p.tiles_x * p.tiles_y) * NPG;
        float* d = dst + (((size_t)(b * G + g) * a.stats_tmul + a.stats_toff) * T + msub * NPG + nsub) * 2;
        if (ltab_on) {    // the tile holds the whole group: (mean, rstd) for the tail below, no trip through memory
            float* ltab = smem + p.o_fin;
            const int ngt = cpg >= BN ? 1 : BN / cpg;
            ltab[2 * (tb * ngt + gl)] = mean;
            ltab[2 * (tb * ngt + gl) + 1] = 1.0f / sqrtf((q - s * mean) / n + a.fin.eps);
        }
        if (coherent) {   // read by the other workgroups of this sample group IN this launch: each value travels as ONE 8-byte
            // write-through store {epoch, bits} -- the data is its own flag (cdna_hip_programming.md G16, R2)
            gu64* gp = (gu64*)(a.fin.gran + (size_t)(d - dst));
            const unsigned long long tag = (unsigned long long)epoch << 32;
            __hip_atomic_store(gp, tag | __float_as_uint(mean), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gp + 1, tag | __float_as_uint(q - s * mean), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!LEAN && a.fin.raw) { d[0] = mean; d[1] = q - s * mean; }     // the backward reads them in the ordinary form
        } else {
            d[0] = mean;
            d[1] = q - s * mean;
        }
    };
    // thread i <-> (sample tb, column col) sums its column over the sample's 16-row half-blocks, then the cpgt columns of a
    // group -- consecutive lanes of one wave, cpgt a power of two <= 64 -- meet by xor-shuffles; the group's first lane writes
    // (mean_t, M2_t) of this tile's share of group g of sample b.
    auto emit = [&](float* dst, int G, int cpg, int cpgt, int NPG, bool coherent, bool ltab) {
        const int hb_per = p.rps >> 4;
        const int ncols = min(BN, Cout - n0);
        for (int i0 = 0; i0 < p.TB * BN; i0 += nthr) {
            const int i = i0 + tid;
            const bool live = i < p.TB * BN;
            const int tb = live ? i / BN : 0, col = live ? i - tb * BN : 0;
            float s = 0.f, q = 0.f;
            if (live) {
#pragma unroll 8
                for (int h = 0; h < hb_per; ++h) { s += partS[(tb * hb_per + h) * BN + col]; q += partQ[(tb * hb_per + h) * BN + col]; }
            }
            for (int o = cpgt >> 1; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
            if (live && (col & (cpgt - 1)) == 0 && col < ncols && b0 + tb < a.B) publish(dst, G, cpg, cpgt, NPG, coherent, ltab, tb, col / cpgt, s, q);
        }
    };
    // The same partials when the tile lies inside ONE sample (TB == 1, every layer with >= BM pixels per image): each accumulator wave
    // folds its own rows and its group's columns in registers (shuffles only), one LDS word pair per (wave row, group) crosses to the
    // publishing thread.  The general form above goes through [BM/16][BN] LDS tables and sums them again per thread: 5 k cycles of the
    // 10 k-cycle epilogue of a 32x32 layer, this is 2 k.
    auto stats_fast = [&](float* dst, int G, int cpg, int cpgt, int NPG, bool coherent, bool ltab) {
        const int lanes = cpgt < 32 ? cpgt : 32;          // columns of a group inside one 32-column accumulator block
        const int per = cpgt <= 32 ? 1 : cpgt / 32;       // accumulator blocks a group spans (cpgt == 64 with NT == 2)
        if (owner) {
            float sv[NT], qv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float s_ = 0.f, q_ = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float v = acc[mt][nt][r]; s_ += v; q_ += v * v; }
                s_ += __shfl_xor(s_, 32);
                q_ += __shfl_xor(q_, 32);
                for (int o = lanes >> 1; o > 0; o >>= 1) { s_ += __shfl_xor(s_, o); q_ += __shfl_xor(q_, o); }
                sv[nt] = s_; qv[nt] = q_;
            }
            if (half == 0 && (l31 & (lanes - 1)) == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (per > 1 && (nt % per)) continue;
                    float s_ = sv[nt], q_ = qv[nt];
                    if (per > 1) {
#pragma unroll
                        for (int k = 1; k < NT; ++k) if (k < per && nt + k < NT) { s_ += sv[nt + k]; q_ += qv[nt + k]; }
                    }
                    const int col = (wn * NT + nt) * 32 + l31;
                    partS[wm * BN + col] = s_;
                    partQ[wm * BN + col] = q_;
                }
            }
        }
        __syncthreads();
        const int ncols = min(BN, Cout - n0);
        for (int i = tid; i < BN / cpgt; i += nthr) {
            const int col = i * cpgt;
            if (col >= ncols || b0 >= a.B) continue;
            float s_ = 0.f, q_ = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WM; ++w2) { s_ += partS[w2 * BN + col]; q_ += partQ[w2 * BN + col]; }
            publish(dst, G, cpg, cpgt, NPG, coherent, ltab, 0, i, s_, q_);
        }
    };
    constexpr bool CAN_MULTI = (FL & FL_MULTI) || (WM * MT == 1);     // the 32-row tile is the small-image tile: samples share it
    const bool fast_stats = !CAN_MULTI || p.TB == 1;
    // The 32-row tile over two 16-pixel samples (every 4x4 layer of the U-Net): accumulator registers 0-7 of a lane are sample 0, 8-15
    // sample 1 (row = (r & 3) + 8 (r >> 2) + 4 half), so the partials are register sums + shuffles inside the one wave that owns the
    // tile -- no LDS tables, no barrier.  The general multi-sample form below costs 2.6 k cycles per pass here (stamps, round 3), and a
    // convolution that closes its Block runs it twice.
    constexpr bool PAIR_TILE = (WM * MT == 1 && WN * NT == 1);
    const bool pair_stats = PAIR_TILE && p.TB == 2 && p.rps == 16;
    auto stats_pair = [&](float* dst, int G, int cpg, int cpgt, int NPG, bool coherent, bool ltab) {
        if (owner) {
            float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) { const float v = acc[0][0][r], w = acc[0][0][r + 8]; s0 += v; q0 += v * v; s1 += w; q1 += w * w; }
            s0 += __shfl_xor(s0, 32); q0 += __shfl_xor(q0, 32); s1 += __shfl_xor(s1, 32); q1 += __shfl_xor(q1, 32);
            for (int o = cpgt >> 1; o > 0; o >>= 1) {
                s0 += __shfl_xor(s0, o); q0 += __shfl_xor(q0, o); s1 += __shfl_xor(s1, o); q1 += __shfl_xor(q1, o);
            }
            const int ncols = min(BN, Cout - n0);
            if (half == 0 && (l31 & (cpgt - 1)) == 0 && l31 < ncols) {
                if (b0 < a.B) publish(dst, G, cpg, cpgt, NPG, coherent, ltab, 0, l31 / cpgt, s0, q0);
                if (b0 + 1 < a.B) publish(dst, G, cpg, cpgt, NPG, coherent, ltab, 1, l31 / cpgt, s1, q1);
            }
        }
    };

    // Statistics first, stores last: a workgroup barrier waits for every outstanding global store (s_waitcnt vmcnt(0)), so a
    // barrier AFTER the output stores would park the whole workgroup for the store round trip.
    if (owner) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int ncol = (wn * NT + nt) * 32 + l31, n = n0 + ncol;
                const bool nok = n < Cout;
                const float bias = pre ? pre[nt] : ((a.bias && nok) ? a.bias[n] : 0.f);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] += bias;
                if (stats_post) {   // statistics of the FINAL value (activation and residual included): pre-norm resnets (SD-VAE)
                    // all sixteen residual values requested before the first is used (row by row it was sixteen dependent round trips)
                    float ad[16];
                    const bool addp = a.add && nok;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int px = pix_of(mt, r);
                        ad[r] = (addp && px >= 0) ? a.add[(size_t)px * Cout + n] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[mt][nt][r];
                        if (a.out_act) v = silu_f(v);
                        acc[mt][nt][r] = v + ad[r];
                    }
                }
            }
    }
    if (do_stats && !fast_stats && !pair_stats) block_sums();

    if (FL & FL_STAMP) conv_stamp(p, 7);
    if (do_stats) {
        if (pair_stats) stats_pair(a.stats_out, a.Gout, p.cpg, p.cpgt, p.NPG, meeting, fin && !meeting);
        else if (fast_stats) stats_fast(a.stats_out, a.Gout, p.cpg, p.cpgt, p.NPG, meeting, fin && !meeting);
        else {
            __syncthreads();
            emit(a.stats_out, a.Gout, p.cpg, p.cpgt, p.NPG, meeting, fin && !meeting);
        }
    }
    if (FL & FL_STAMP) conv_stamp(p, 14);

    if (!LEAN && fin && a.fin.raw && owner) {   // training: the pre-norm value stays (GroupNorm backward needs it); issued before the meeting's wait
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + (wn * NT + nt) * 32 + l31;
                if (n >= Cout) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int px = pix_of(mt, r);
                    if (px >= 0) a.fin.raw[(size_t)px * Cout + n] = acc[mt][nt][r];
                }
            }
    }
    if (fin) {
        // ---- meet the other workgroups of this sample group: their partials complete the GroupNorm statistics ----
        // The exchanged bytes (partials, counter) travel as device-scope relaxed atomics -- sc1 accesses that bypass the
        // non-coherent per-XCD L2 -- ordered by vmcnt(0) + the workgroup barrier; no cache write-back / invalidate, which
        // a release/acquire fence pair would cost every workgroup (measured: 120 us per launch instead of 25).
        float* tab = smem + p.o_fin;
        const int ngt = p.cpg >= BN ? 1 : BN / p.cpg;
        // The residual is requested BEFORE the wait for the other workgroups' partials: it depends on nothing computed here, it is a cold
        // miss (another kernel wrote it), and behind the meeting it was a second memory round trip on the tail's critical path (round 3).
        // (Round 4 tried it earlier still -- in front of the statistics, with LDS-only barriers for the accumulator waves so that the
        // statistics' barriers do not wait for it: every Block-closing launch 0.3-0.7 us SLOWER in the op table, 828-829 against 831-833
        // samples/s.  The wait that follows a request is in order, so whatever comes first behind it pays its round trip; here that is the
        // poll, which has to wait anyway.)
        float rs[MT][NT][16];
        if (owner) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = n0 + (wn * NT + nt) * 32 + l31;
                    const bool nok = n < Cout;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int px = pix_of(mt, r);
                        rs[mt][nt][r] = (a.fin.res && nok && px >= 0) ? a.fin.res[(size_t)px * Cout + n] : 0.f;
                    }
                }
        }
        if (meeting) {
            // One round trip when the others have already published: every thread polls ONE granule (sc1 loads that bypass this CU's
            // L1) until its tag is this launch's epoch, parks the value in LDS; then the Chan combination runs from LDS.  Bounded:
            // a residency mistake must not hang the device (err is checked by the host: fc_unet_fused_tail_errors).
            const int Tst = (p.TB > 1 ? 1 : p.tiles_x * p.tiles_y) * p.NPG;
            const int cnt = p.TB * ngt * Tst * 2;
            float* gv = smem + p.o_gran;
            for (int i = tid; i < cnt; i += nthr) {
                const int k = i & 1, t = (i >> 1) % Tst, j = (i >> 1) / Tst, tb = j / ngt, gl = j - tb * ngt;
                const int b = b0 + tb, g = n0 / p.cpg + gl;
                float val = 0.f;
                if (b < a.B && g < a.Gout) {
                    const gu64* gp = (const gu64*)(a.fin.gran + ((size_t)(b * a.Gout + g) * Tst + t) * 2 + k);
                    unsigned long long v = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    int spins = 0;
                    while ((unsigned)(v >> 32) != epoch) {
                        // A wait that gives up must not leave finite garbage behind: the statistic becomes NaN, so every value of this
                        // sample group is NaN from here to the sampler's output, and the handle's error word is set for the host
                        // (fc_unet_check / the next call on the handle return FC_E_STATE).
                        if (++spins > (1 << 20)) { if (a.fin.err) *a.fin.err = 1; v = 0x7fc00000ull; break; }
                        __builtin_amdgcn_s_sleep(2);
                        v = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    val = __uint_as_float((unsigned)v);
                }
                gv[i] = val;
            }
            lds_only_barrier();
            for (int i = tid; i < p.TB * ngt; i += nthr) {
                const int tb = i / ngt, gl = i - tb * ngt, b = b0 + tb, g = n0 / p.cpg + gl;
                float mean = 0.f, rstd = 0.f;
                if (b < a.B && g < a.Gout) {
                    const float* sp = gv + (size_t)i * Tst * 2;
                    const float nt_ = (float)(p.rps * p.cpgt);
                    float sm = 0.f;
                    for (int t = 0; t < Tst; ++t) sm += sp[2 * t];
                    mean = sm / (float)Tst;
                    float m2 = 0.f, dv = 0.f;
                    for (int t = 0; t < Tst; ++t) {
                        const float d = sp[2 * t] - mean;
                        m2 += sp[2 * t + 1];
                        dv += d * d;
                    }
                    rstd = 1.0f / sqrtf((m2 + nt_ * dv) / (nt_ * (float)Tst) + a.fin.eps);
                }
                tab[2 * i] = mean;
                tab[2 * i + 1] = rstd;
            }
        }
        lds_only_barrier();
        if ((FL & FL_STAMP) && active) conv_stamp(p, 12);  // (accumulator rows) the statistics of the whole group are in LDS: meeting over, or the local table
        if (owner) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int ncol = (wn * NT + nt) * 32 + l31, n = n0 + ncol;
                    const bool nok = n < Cout;
                    const int gl = p.cpg >= BN ? 0 : ncol / p.cpg;
                    const float gam = pre ? pre[2 * NT + nt] : (nok ? a.fin.gamma[n] : 0.f), bet = pre ? pre[3 * NT + nt] : (nok ? a.fin.beta[n] : 0.f);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        const int tb = m >> (p.TWl + p.THl);
                        const float mean = tab[2 * (tb * ngt + gl)], rstd = tab[2 * (tb * ngt + gl) + 1];
                        const float A = rstd * gam;
                        const float v = silu_f(A * acc[mt][nt][r] + (bet - mean * A)) + rs[mt][nt][r];
                        acc[mt][nt][r] = (nok && b0 + tb < a.B) ? v : 0.f;
                    }
                }
        }
        if ((FL & FL_STAMP) && active) conv_stamp(p, 2);   // (accumulator rows) normalised + activated + residual added
        if (LEAN ? bool(FL & FL_GN1) : a.fin.gn1_out != nullptr) {   // GroupNorm(1) partials of the final value for the PreNorm that follows (unet.py:156-160)
            lds_only_barrier();                    // every reader of part* / tab is done
            const int cpg1 = Cout, cpgt1 = Cout < BN ? Cout : BN, NPG1 = Cout >= BN ? Cout / BN : 1;
            if (pair_stats) stats_pair(a.fin.gn1_out, 1, cpg1, cpgt1, NPG1, false, false);
            else if (fast_stats) stats_fast(a.fin.gn1_out, 1, cpg1, cpgt1, NPG1, false, false);
            else {
                block_sums();
                lds_only_barrier();
                emit(a.fin.gn1_out, 1, cpg1, cpgt1, NPG1, false, false);
            }
        }
        if ((FL & FL_STAMP) && active) conv_stamp(p, 13);  // GroupNorm(1) partials out
    }

    // Output stores.  Written flat: the pixel index of every accumulator row first (one 32-bit value per row; -1 = sample beyond B),
    // then the optional activation / residual with all its loads in flight together, then the stores back to back.  The earlier
    // per-row form (decode, three uniform branches, load, wait, store -- sixteen times) took 4.8 k cycles of a 32x32 layer's 42 k.
    //
    // Wide form (p.o_out >= 0): an accumulator lane holds ONE channel of sixteen pixels, so its natural stores are sixteen dword
    // instructions of two 128-byte segments each -- store-ISSUE bound (3.9 k cycles of a 32x32 layer).  Through an LDS image
    // [row][BN + 4] every thread of the workgroup (the staging waves too) then writes whole 16-byte channel quads: a quarter of the
    // store instructions, each 1 KiB contiguous when Cout == BN.
    const bool post = !stats_post && !fin;                // activation / residual still to apply (else already in the accumulators)
    const bool act = (FL & FL_POSTOP) && post && a.out_act;
    const float* addp = ((FL & FL_POSTOP) && post) ? a.add : nullptr;
    if (owner) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int pix[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pix[r] = pix_of(mt, r);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int ncol = (wn * NT + nt) * 32 + l31, n = n0 + ncol;
                if (n >= Cout) continue;
                float* op = a.out + n;
                if (addp) {
                    float ad[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) ad[r] = pix[r] >= 0 ? addp[(size_t)pix[r] * Cout + n] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = (act ? silu_f(acc[mt][nt][r]) : acc[mt][nt][r]) + ad[r];
                } else if (act) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = silu_f(acc[mt][nt][r]);
                }
                if (!LEAN && p.o_out < 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (pix[r] >= 0) op[(size_t)pix[r] * Cout] = acc[mt][nt][r];
                    if (has_res) {
                        const float rbias = pre ? pre[NT + nt] : (a.res_b ? a.res_b[n] : 0.f);
                        float* rp = a.res_out + n;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (pix[r] >= 0) rp[(size_t)pix[r] * Cout] = accr[mt][nt][r] + rbias;
                    }
                }
            }
        }
    }
    if ((FL & FL_STAMP) && active) conv_stamp(p, 3);       // (accumulator rows; a staging wave's slot 3 is its first LDS store)
    if (LEAN || p.o_out >= 0) {      // lean flavours are only launched when the LDS image fits
        constexpr int OS = BN + 4, Q4 = BN / 4;
        float* ot = smem + p.o_out;
        for (int pass = 0; pass < (has_res ? 2 : 1); ++pass) {
            if (pass) lds_only_barrier();                   // the image of the first output has been read by everyone
            if (owner) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int ncol = (wn * NT + nt) * 32 + l31;
                        const float rbias = pass ? (pre ? pre[NT + nt] : ((a.res_b && n0 + ncol < Cout) ? a.res_b[n0 + ncol] : 0.f)) : 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            ot[m * OS + ncol] = pass ? accr[mt][nt][r] + rbias : acc[mt][nt][r];
                        }
                    }
            }
            if (active && !pass) conv_stamp(p, 9);
            lds_only_barrier();
            if (active && !pass) conv_stamp(p, 10);
            float* gout = pass ? a.res_out : a.out;
            for (int i = tid; i < BM * Q4; i += nthr) {
                const int m = i / Q4, c4 = (i - m * Q4) * 4, n = n0 + c4;
                const int tw = m & (TW - 1), th = (m >> p.TWl) & (TH - 1), tb = m >> (p.TWl + p.THl);
                const int b = b0 + tb;
                if (b < a.B && n < Cout) {
                    const float4 v = *reinterpret_cast<const float4*>(ot + m * OS + c4);
                    *reinterpret_cast<float4*>(gout + (size_t)((b * (a.H << a.out_sh) + ((y0 + th) << a.out_sh) + a.out_oy) * (a.W << a.out_sh) + ((x0 + tw) << a.out_sh) + a.out_ox) * Cout + n) = v;
                }
            }
        }
    }
    if (FL & FL_STAMP) conv_stamp(p, 8);
}

int conv_pipe_launch(const ConvDev& d, int tile, int grid, size_t lds, hipStream_t s);
int conv_pipe_blocks_per_cu(const ConvDev& d, int tile, size_t lds);   // occupancy of the instantiation that launch would use (0: unknown)
int conv_pipe_init();
bool conv_pipe_supports_ks(int ks);
const float* conv_zeros16();

}  // namespace fc
