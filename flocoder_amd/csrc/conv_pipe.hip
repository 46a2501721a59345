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
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include "conv_dev.h"
#include "stats_dev.h"

namespace fc {

typedef __attribute__((address_space(3))) void* lptr_t;

// One LDS-DMA wave instruction: 64 lanes x 16 B from per-lane global addresses to lds_dst + lane*16 (wave-uniform base in
// M0).  Inline asm on purpose (cdna_hip_programming.md 5.7): hipcc then keeps these out of its vmcnt bookkeeping, so the
// waits it inserts for the loaders' ordinary register loads do not drain a slab that was issued for a LATER stage; we
// count them by hand (explicit vmcnt(0) before a stage is handed over).  M0 is saved/restored inside the statement.
// (the destination travels as an LDS byte address, not as a generic pointer: casting a generic pointer back to LDS at every site made
// hipcc 7.2 emit an illegal aperture test -- "V_CMP_NE_U32 0, $src_shared_base" -- in some instantiations)
__device__ __forceinline__ void lds_dma16(const float* gsrc, unsigned lds_byte) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Register load the compiler does not count either (same reason): 16 B per lane; the caller waits with wait_vmcnt().
__device__ __forceinline__ void hidden_load16(f32x4& dst, const float* gsrc) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(gsrc) : "memory");
}
// s_waitcnt vmcnt(n) for a run-time n (the instruction takes an immediate).  vmcnt counts loads and LDS-DMA together in
// issue order, so "all but my n youngest" = everything issued before the slab that is allowed to stay in flight.
__device__ __forceinline__ void wait_vmcnt(int n) {
    // s_waitcnt ignores EXEC, so the selection must be a SCALAR branch: readfirstlane makes n provably wave-uniform
    switch (__builtin_amdgcn_readfirstlane(n)) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // more pieces than cases: drain (correct, just no lookahead)
    }
    __builtin_amdgcn_sched_barrier(0);   // nothing that reads the hidden loads' registers may move above the wait
}
__device__ __forceinline__ void loader_handover() {   // LDS stores of this wave visible, then the workgroup barrier; no vmcnt drain
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// NL = loader waves (4 or 8).  The small-M tiles stream a whole weight slab per 32 or 64 output rows: with four loader waves the
// slab's LDS-DMA issue (~190 cycles per 1 KB piece per wave) took 1.65x the consumers' MFMA time per chunk and the consumers sat
// at the chunk barrier half of the time (profiles/r01_c_stamps.txt); eight loader waves halve the issue time per wave.
// PREC = 1: split-bf16 arithmetic (round 3, codec decoders on request -- never the U-Net, never a default): every fp32 operand x is taken as
// hi + lo with hi = bf16(x), lo = bf16(x - hi), and a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulation; the
// dropped lo*lo term is 2^-16 relative).  Three MFMAs of 32 cycles cover sixteen channels where the exact-fp32 pipe needs eight of 64: 5.3x
// the matrix rate.  Nobody splits inside the accumulator waves' loop: the weights are split once, when the parameters are packed (pack kind 8,
// ConvArgs::w_b3) and reach LDS by the same LDS-DMA pieces as the fp32 slab; the window is split by the staging waves on its way into LDS
// (a pixel is read by up to nine taps).  The accumulator waves read 16-byte hi / lo operands and issue MFMAs.  The epilogue is the fp32 kernel's.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// four fp32 values -> their bf16 hi parts and lo parts, packed two per dword (channel order kept)
__device__ __forceinline__ void split_bf16x4(const f32x4& x, uint2& hi, uint2& lo) {
    union { bf16x2_t p; unsigned u; } H0, H1, L0, L1;
    f32x2_t a; a[0] = x.x; a[1] = x.y;
    f32x2_t b; b[0] = x.z; b[1] = x.w;
    H0.p = __builtin_convertvector(a, bf16x2_t);
    H1.p = __builtin_convertvector(b, bf16x2_t);
    f32x2_t ra; ra[0] = x.x - __uint_as_float(H0.u << 16); ra[1] = x.y - __uint_as_float(H0.u & 0xffff0000u);
    f32x2_t rb; rb[0] = x.z - __uint_as_float(H1.u << 16); rb[1] = x.w - __uint_as_float(H1.u & 0xffff0000u);
    L0.p = __builtin_convertvector(ra, bf16x2_t);
    L1.p = __builtin_convertvector(rb, bf16x2_t);
    hi = make_uint2(H0.u, H1.u); lo = make_uint2(L0.u, L1.u);
}

template <int WM, int WN, int WK, int MT, int NT, int CC, int NPL, int KS, int NL, int FL = FL_ALL, int PREC = 0>
__global__ void __launch_bounds__(256 + 64 * NL) conv_pipe_kernel(const ConvDev p_kernarg) {
    ConvDev p;
    conv_params_from_lanes(p);       // p_kernarg itself is never touched: see conv_dev.h
    constexpr int NTHR = 256 + 64 * NL, LT = 64 * NL;
    constexpr int BN = 32 * NT * WN, KSTEPS = CC / 2, KPW = KSTEPS / WK, Q = CC / 4, PIXSTEP = LT / Q, KK = KS * KS;
    static_assert(WM * WN * WK == 4, "4 consumer waves per workgroup");
    // DB ("direct B"): with one 32 x 32 output tile per workgroup every weight element feeds exactly one wave, so staging the slab in
    // LDS buys no reuse -- it only makes the loaders the pacemaker (stamps at 256->256 @4x4: loaders busy 37 k of the 43 k main-loop
    // cycles, consumers 43 % of theirs at the chunk barrier).  The consumers then load their K-quarter of the slab from global
    // straight into the MFMA B register layout (two 128-byte row segments per wave load), one chunk ahead, and the loaders keep
    // only the input window.
    constexpr bool DB = (WM * WN == 1 && MT == 1 && NT == 1 && KS == 3);
    // DB4 (round 3): the same with the weights in the k-step-quad layout (pack kind 6): a lane's B operands of the four k-steps its wave
    // owns in a chunk are ONE 16-byte load per tap (nine loads per chunk instead of 36, uniform base + 32-bit lane offset), and the two
    // register sets swap roles from chunk to chunk instead of being copied.  The copy was the round-2 kernel's hidden stall: the compiler
    // moved `wcur = wnxt` up to the last use of each wcur register, and every such move waits for a load issued moments earlier --
    // s_waitcnt vmcnt(35) in front of the chunk's FIRST MFMA, vmcnt(8) in front of its fourth: the "prefetch" was waited for at once
    // (ISA of <1,1,4,1,1,32,4,3,8,64>; 115 cycles per MFMA instead of 64).
    constexpr bool DB4 = DB && (FL & FL_W4) != 0;
#ifndef FC_DB4_CHAINS
#define FC_DB4_CHAINS 2
#endif
    constexpr int DB4_CHAINS = FC_DB4_CHAINS;
    // Pixel stride of the staged window in LDS.  CC + 1 keeps the dword operand reads of the MFMA lanes (one pixel per lane) off each other's
    // banks.  DB4 reads a lane's four k-steps of a tap as ONE 16-byte LDS read instead: inside every 8-channel block the window is stored
    // as [parity][4] (channel c at 8 (c / 8) + 4 (c & 1) + ((c >> 1) & 3)), so the channels 8 wk + 2 kk + half, kk = 0..3, are contiguous, and
    // the stride is a multiple of four floats (CC + 4) for the alignment the wide read needs.
    constexpr bool BF3 = PREC == 1;
    static_assert(!BF3 || (!DB && CC == 16 && FL == FL_ALL), "split-bf16: LDS-fed tiles, one 16-channel k-step per chunk, the all-in-one flavour");
    // BF3: the staging waves split the window on its way into LDS (a pixel is read by up to nine taps: splitting it at every read was half of
    // the accumulator waves' vector work): a pixel's 16 channels are stored as 16 bf16 hi parts | 16 bf16 lo parts (64 bytes) at the fp32
    // form's pitch of CC + 4 floats, so a lane's eight channels of either part are one 16-byte read.
    constexpr int CS = (DB4 || BF3) ? CC + 4 : CC + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ConvArgs& a = p.a;
    float* gstat = smem + p.o_gstat;
    float2* aff = reinterpret_cast<float2*>(smem + p.o_aff);   // [TB][Cin]
    float* patch0 = smem + p.o_patch;
    float* wl0 = smem + p.o_wl;
    const unsigned smem_lds = (unsigned)(size_t)(lptr_t)smem;      // LDS byte address of the dynamic segment

    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
    const bool consumer = wave8 < 4;
    const int wave = wave8 & 3;                                   // consumer wave 0..3 (meaningless in a loader wave)
    const int lw = (wave8 - 4) & (NL - 1), ltid = (tid - 256) & (LT - 1);   // loader wave / loader thread (meaningless in a consumer wave)
    const int half = lane >> 5, l31 = lane & 31;
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);

    const int bid = conv_parity_select(p, xcd_remap(blockIdx.x, p.nblocks));
    // tile decode without integer division: tiles_x / tiles_y are powers of two, ntiles goes through a host-made reciprocal
    const int mt_i = p.ntiles == 1 ? bid : (p.magic_nt ? (int)__umulhi((unsigned)bid, p.magic_nt) : bid / p.ntiles);
    const int nt_i = bid - mt_i * p.ntiles;
    const int tx = mt_i & (p.tiles_x - 1), ty = (mt_i >> p.txl) & (p.tiles_y - 1), bg = mt_i >> (p.txl + p.tyl);
    const int TW = 1 << p.TWl, TH = 1 << p.THl;
    const int b0 = bg * p.TB, y0 = ty * TH, x0 = tx * TW, n0 = nt_i * BN;
    const int PW = p.PW, PHW = p.PH * p.PW;
    const int C0 = a.s0.C, C1 = a.s1.C, Cin = a.Cin, Cout = a.Cout;
    constexpr bool LEAN = FL != FL_ALL;        // launched for an exact match of the mask only: a set bit means "on"
    const bool has_res = LEAN ? bool(FL & FL_RES) : a.res_out != nullptr;
    const bool xf_on = LEAN ? bool(FL & FL_XF) : p.any_xf != 0;
    const int nchunks = p.nchunks;
    if (FL & FL_STAMP) conv_stamp(p, 0);
    // Fused tail across workgroups: draw this launch's epoch from the sample group's arrival counter NOW -- the round trip hides behind
    // the whole main loop; every workgroup of the group gets the same quotient because launches of one op never overlap.
    const bool meet = LEAN ? bool(FL & FL_MEET) : (a.fin.gamma != nullptr && !p.fin_local);
    // Inline asm on purpose: through the builtin, hipcc's atomic optimizer waits for the returned value on the spot (a cold round trip
    // in front of everything else); here the wait sits where the value is used, after the GroupNorm tables.
    unsigned arrival = 0;
    if (meet && tid == 0) {
        const gu32* cnt = (const gu32*)(a.fin.sync + b0 / p.TB);
        asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(arrival) : "v"(cnt), "v"(1u) : "memory");
    }

    // ---- GroupNorm tables: moments per (sample, group), then the folded affine per (sample, channel).  The staging waves build them
    // AFTER they have put the first weight slabs in flight (two dependent global round trips that the slab transfer overlaps).
    // The tables are built by the staging waves alone (`worker`, thread `id` of `nthr`): the accumulator waves never read them, and a
    // second inlined copy of this code in their path was a few KB of cold instruction fetches per launch.  Both roles meet in the two barriers.
    auto gn_tables = [&](const bool worker, const int id, const int nthr) {
      if (xf_on) {
        const int G0 = a.s0.xf.mode ? a.s0.xf.G : 0, G1 = ((FL & FL_CAT) && a.s1.xf.mode) ? a.s1.xf.G : 0;
        // second-phase operands (gamma, beta, FiLM scale / shift of this thread's first (sample, channel) entry) are requested
        // together with the statistics: one memory round trip for the two tables instead of two dependent ones
        float pg = 1.f, pbt = 0.f, psc = 0.f, psh = 0.f;
        bool p_on = false, p_ss = false;
        int p_gs = 0;
       if (worker) {
        if (id < p.TB * Cin) {
            const int tb = id / Cin, c = id - tb * Cin, b = b0 + tb;
            const bool first = !(FL & FL_CAT) || c < C0;
            const SrcXform& xf = first ? a.s0.xf : a.s1.xf;
            if (b < a.B && xf.mode) {
                const int cs = first ? c : c - C0, Cs = first ? C0 : C1;
                p_on = true;
                p_gs = 2 * ((first ? 0 : p.TB * G0) + tb * xf.G + cs / (Cs / xf.G));
                pg = xf.gamma[cs];
                pbt = xf.beta[cs];
                if (xf.ss) {
                    p_ss = true;
                    psc = xf.ss[(size_t)b * xf.ss_stride + cs];
                    psh = xf.ss[(size_t)b * xf.ss_stride + Cs + cs];
                }
            }
        }
        for (int i = id; i < p.TB * (G0 + G1); i += nthr) {
            const bool first = !(FL & FL_CAT) || i < p.TB * G0;
            const SrcXform& xf = first ? a.s0.xf : a.s1.xf;
            const int j = first ? i : i - p.TB * G0;
            const int tb = j / xf.G, g = j - tb * xf.G, b = b0 + tb;
            float mean = 0.f, rstd = 0.f;
            if (b < a.B) combine_partials(xf, b, g, &mean, &rstd);   // all partial pairs requested at once (stats_dev.h)
            gstat[2 * i] = mean;
            gstat[2 * i + 1] = rstd;
        }
       }
        if ((FL & FL_STAMP) && worker) conv_stamp(p, 2);      // (staging rows) statistics combined: the first cold round trip is over
        // LDS-only barriers (round 4): the two tables are LDS data, and __syncthreads() also parks every wave until ALL its global
        // accesses have returned -- the accumulator waves' epoch atomic, epilogue operands and first weights, a second cold round trip
        // queued behind the statistics' one (tools/fin_stamps.py: statistics combined at 2.15 us, past this barrier at 3.12).  What a wave
        // needs from memory it waits for where it uses it; the staging waves' LDS-DMA has its own explicit wait before the hand-over.
        lds_only_barrier();
        if ((FL & FL_STAMP) && worker) conv_stamp(p, 4);
       if (worker) {
        if (id < p.TB * Cin) {
            float A = 1.f, Bv = 0.f;
            if (p_on) {
                A = gstat[p_gs + 1] * pg;
                Bv = pbt - gstat[p_gs] * A;
                if (p_ss) {
                    const float sc = psc + 1.0f;
                    A *= sc;
                    Bv = Bv * sc + psh;
                }
            }
            aff[id] = make_float2(A, Bv);
        }
        for (int i = id + nthr; i < p.TB * Cin; i += nthr) {
            const int tb = i / Cin, c = i - tb * Cin, b = b0 + tb;
            float A = 1.f, Bv = 0.f;
            const bool first = !(FL & FL_CAT) || c < C0;
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
       }
        lds_only_barrier();
        if ((FL & FL_STAMP) && worker) conv_stamp(p, 5);      // (staging rows) the folded affine tables are in LDS
      }
    };

    f32x16 acc[MT][NT], accr[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[mt][nt][r] = 0.f; accr[mt][nt][r] = 0.f; }

    float pre[4 * NT];   // epilogue operands of this lane's columns (conv_dev.h): requested now, used after the last chunk
#pragma unroll
    for (int i = 0; i < 4 * NT; ++i) pre[i] = 0.f;
    if (consumer) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + (wn * NT + nt) * 32 + l31;
            if (n < Cout) {
                if (a.bias) pre[nt] = a.bias[n];
                if (has_res && a.res_b) pre[NT + nt] = a.res_b[n];
                if (a.fin.gamma) { pre[2 * NT + nt] = a.fin.gamma[n]; pre[3 * NT + nt] = a.fin.beta[n]; }
            }
        }
    }

    if (!consumer) {
        // =========================================== LOADERS ===========================================
        // Waves 4+ are the younger half of the workgroup: at equal priority the SIMD's issue arbitration (priority, then age) hands them
        // the slots the MFMA waves leave over, and the consumers then sit at the chunk barrier waiting for a stage the loaders could
        // not issue fast enough (stamps: 5 k cycles to issue three loads per thread).  s_setprio is scalar: the branch is wave-uniform.
        {
            const int lp = __builtin_amdgcn_readfirstlane(p.loader_prio);
            if (lp == 1) __builtin_amdgcn_s_setprio(1);
            else if (lp == 2) __builtin_amdgcn_s_setprio(2);
            else if (lp == 3) __builtin_amdgcn_s_setprio(3);
        }
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
                    const int tb = fastdiv(pix, p.magic_phw), r = pix - tb * PHW, py = fastdiv(r, p.magic_pw), px = r - py * PW;
                    const int iy = y0 * a.stride - (a.pad_y >= 0 ? a.pad_y : a.pad) + py, ix = x0 * a.stride - (a.pad_x >= 0 ? a.pad_x : a.pad) + px, b = b0 + tb;
                    if (b < a.B && iy >= 0 && iy < Hin && ix >= 0 && ix < Win) e_po[k] = (b * a.Hs + (iy >> a.ups)) * a.Ws + (ix >> a.ups);
                    e_lds[k] = pix * CS + (DB4 ? 8 * ((ltid % Q) >> 1) + 2 * ((ltid % Q) & 1) : q4);
                    e_tb[k] = tb;
                }
            }
        }
        if (FL & FL_STAMP) conv_stamp(p, 1);
        const int nwb = p.nwb;                    // weight stages in LDS: 3 = slabs run two chunks ahead, 2 = one ahead
        // LDS-DMA plan of this lw: piece j = lw + 4m covers 256 floats of the slab = (256/BN) rows x BN columns.  A lane's
        // source pointer at chunk 0 is fixed for the whole kernel and advances by CC*Cout floats per chunk, so issuing a slab
        // costs one 64-bit add + one scalar add per piece (the address arithmetic used to outweigh the transfer: the loaders,
        // not the MFMA waves, set the pace of the small-M layers -- profiles/r01_c_stamps.txt).
        constexpr int MAXP = ((KK + 1) * CC * BN / 256 + NL - 1) / NL;
        const int rows_main = KK * CC, npieces = (rows_main + (has_res ? CC : 0)) * BN / 256;
        const int my_pieces = __builtin_amdgcn_readfirstlane(npieces > lw ? (npieces - lw + NL - 1) / NL : 0);
        const float* d_ptr[MAXP];   // nullptr: column beyond Cout -> zero block
        int d_row[MAXP];            // channel row inside the chunk (for the Cin tail test)
#pragma unroll
        for (int m = 0; m < MAXP; ++m) {
            const int j = lw + NL * m, f = j * 256 + lane * 4, row = f / BN, n = n0 + (f % BN);
            if constexpr (BF3) {
                // split-bf16 slab [part hi/lo][tap][channel block of 8 (two per chunk)][BN] x 16 bytes, read from the pre-split copy of the
                // weights (pack kind 8: per part [tap][Ipad/8][Cout] x 16 bytes, Ipad = 16 nchunks, zero beyond Cin): one lane, one column's
                // eight channels.  Same byte count per chunk as the fp32 slab, so the piece bookkeeping is shared
                const int e = j * 64 + lane, r = e / BN, nn = n0 + e % BN, part = r / (2 * KK), tap = (r >> 1) % KK, c8l = r & 1;
                d_row[m] = 0;
                d_ptr[m] = (nn >= Cout || part > 1) ? nullptr
                         : a.w_b3 + (size_t)part * ((size_t)KK * nchunks * 8 * Cout) + ((size_t)(tap * 2 * nchunks + c8l) * Cout + nn) * 4;
                continue;
            }
            const bool res = row >= rows_main;
            d_row[m] = res ? row - rows_main : row % CC;
            d_ptr[m] = n >= Cout ? nullptr : (res ? a.res_w + (size_t)d_row[m] * Cout + n : wbase + ((size_t)(row / CC) * Cin + d_row[m]) * Cout + n);
        }
        auto dma_weights = [&](int i) {
            const int c0 = i * CC;
            const size_t adv = BF3 ? (size_t)i * 8 * Cout : (size_t)c0 * Cout;
            const unsigned wb = smem_lds + 4u * (unsigned)(p.o_wl + (i % nwb) * p.wl_stride + lw * 256);
            const bool tail = !BF3 && c0 + CC > Cin;    // only the last chunk can run past Cin (the split-bf16 copy is padded with zeros)
#pragma unroll
            for (int m = 0; m < MAXP; ++m) {
                if (m >= my_pieces) break;
                const bool ok = d_ptr[m] != nullptr && (!tail || c0 + d_row[m] < Cin);
                lds_dma16(ok ? d_ptr[m] + adv : p.zeros16, wb + 4u * (unsigned)(m * (NL * 256)));
            }
        };
        f32x4 pv[NPL];
        const int nk = __builtin_amdgcn_readfirstlane((p.P * Q + LT - 1) / LT);   // element slots in use (scalar: cheap loop exits)
        auto issue_patch = [&](int i, f32x4 (&pv)[NPL]) {   // input window of chunk i -> registers, every load issued back to back
            const int c = i * CC + q4;
            const bool live = c < Cin, first = !(FL & FL_CAT) || c < C0;
            const float* base = first ? a.s0.p + c : a.s1.p + (c - C0);
            const int Cs = first ? C0 : C1;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {       // padding / out-of-range elements read the zero block: no branches, zeros arrive
                if (k >= nk) break;
                hidden_load16(pv[k], (live && e_po[k] >= 0) ? base + (size_t)e_po[k] * Cs : p.zeros16);
            }
        };
        auto store_patch = [&](int i, f32x4 (&pv)[NPL]) {   // ... and on into LDS with GroupNorm / FiLM / SiLU applied
            const int c = i * CC + q4;
            float* pb = patch0 + (i & 1) * p.patch_stride;
            const bool live = c < Cin, act = (c < C0) ? p.act0 : p.act1;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                if (k >= nk) break;
                if (e_lds[k] < 0) continue;
                f32x4 x = pv[k];
                if (xf_on && live && e_po[k] >= 0) {
                    const float2* ab = aff + e_tb[k] * Cin + c;
                    x.x = ab[0].x * x.x + ab[0].y; x.y = ab[1].x * x.y + ab[1].y;
                    x.z = ab[2].x * x.z + ab[2].y; x.w = ab[3].x * x.w + ab[3].y;
                    if (act) { x.x = silu_f(x.x); x.y = silu_f(x.y); x.z = silu_f(x.z); x.w = silu_f(x.w); }
                }
                float* d = pb + e_lds[k];
                if (BF3) {                                                        // hi parts at byte 2 c, lo parts at byte 2 CC + 2 c of the pixel
                    uint2 h2, l2;
                    split_bf16x4(x, h2, l2);
                    char* pc = reinterpret_cast<char*>(pb + (e_lds[k] - q4)) + 2 * q4;
                    *reinterpret_cast<uint2*>(pc) = h2;
                    *reinterpret_cast<uint2*>(pc + 2 * CC) = l2;
                }
                else if (DB4) { d[0] = x.x; d[1] = x.z; d[4] = x.y; d[5] = x.w; }      // [parity][4] inside the 8-channel block
                else { d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w; }
            }
        };
        unsigned long long dbg_mem = 0, dbg_store = 0, dbg_bar = 0, dbg_issue = 0, dbg_dma = 0;   // diagnostics: cycles per loader phase
        if (!DB) {
            dma_weights(0);
            if (nwb == 3 && nchunks > 1) dma_weights(1);
        }
        {   // window 0 does not depend on the tables: request it first, so that its round trip overlaps theirs.  Ordinary loads --
            // the compiler must see them pending across gn_tables() (hidden loads are only safe in straight-line code).
            const int c = q4;
            const bool live = c < Cin, first = !(FL & FL_CAT) || c < C0;
            const float* base = first ? a.s0.p + c : a.s1.p + (c - C0);
            const int Cs = first ? C0 : C1;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                if (k >= nk) break;
                pv[k] = *reinterpret_cast<const f32x4*>((live && e_po[k] >= 0) ? base + (size_t)e_po[k] * Cs : p.zeros16);
            }
        }
        gn_tables(true, ltid, LT);
        store_patch(0, pv);
        if (FL & FL_STAMP) conv_stamp(p, 3);
        // EXPLICIT, unconditional wait for this wave's slab pieces before stage 0 is handed over (round 4; the root cause of the results
        // that differed when two replicas ran side by side).  The LDS-DMA above is invisible to the compiler, so until now the only thing
        // that retired it was the compiler's own wait for the window loads in store_patch -- and that wait sits INSIDE the `e_lds >= 0`
        // branch: a loader wave none of whose lanes has a window element (the folded-upsampling launch at 4x4: 2 samples x 5 x 5 window
        // pixels x 8 channel quads = 400 elements for 512 loader threads, wave 7 idle) skipped it (s_cbranch_execz) and reached the barrier
        // with its pieces -- rows 56..63 of taps 1 and 3 -- possibly still in flight; the consumers then multiplied whatever the previous
        // workgroup had left in that LDS.  Alone on the GPU the pieces, requested a memory round trip before the window, had always landed;
        // beside a second stream's kernels they sometimes had not (tools/race_hunt.py: first differing tensor ups.0.3, one parity class, one
        // 32-column tile, exactly the pixels tap 1 or tap 3 reaches).  Busy waves drained to zero here already (the compiler's wait counts
        // only its own, younger loads), so this costs nothing.  Issued as the BUILTIN (a real S_WAITCNT instruction, vmcnt(0) with the other
        // counters left alone), not as inline asm: the compiler's wait-count pass reads it and then KNOWS that the window loads above have
        // completed on every path.  Without that knowledge it assumed they might still be pending wherever a lane had skipped its store, and
        // protected the registers they share with the hidden loads of the loop by an `s_waitcnt vmcnt(0)` in front of the loop's window
        // requests and another in front of every LDS store -- each of which also drained the slab that was meant to stay in flight.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        loader_handover();                        // stage 0 ready
        for (int g = 0; g < nchunks; ++g) {       // consumers are on chunk g
            const bool next = g + 1 < nchunks;
            int ahead = 0;
            unsigned long long ta = ((FL & FL_STAMP) && p.stamps) ? __builtin_amdgcn_s_memtime() : 0ull;
            if (!DB && next && nwb == 2) dma_weights(g + 1);
            if (next) issue_patch(g + 1, pv);
            unsigned long long tb_ = ((FL & FL_STAMP) && p.stamps) ? __builtin_amdgcn_s_memtime() : 0ull;
            if (!DB && nwb == 3 && g + 2 < nchunks) { dma_weights(g + 2); ahead = my_pieces; }
            unsigned long long t0 = ((FL & FL_STAMP) && p.stamps) ? __builtin_amdgcn_s_memtime() : 0ull;
            if ((FL & FL_STAMP) && p.stamps) { dbg_issue += tb_ - ta; dbg_dma += t0 - tb_; }
            wait_vmcnt(ahead);                    // window g+1 in registers, slab g+1 landed; slab g+2 stays in flight across the barrier
            unsigned long long t1 = ((FL & FL_STAMP) && p.stamps) ? __builtin_amdgcn_s_memtime() : 0ull;
            if (next) store_patch(g + 1, pv);
            unsigned long long t2 = ((FL & FL_STAMP) && p.stamps) ? __builtin_amdgcn_s_memtime() : 0ull;
            loader_handover();
            if ((FL & FL_STAMP) && p.stamps) { dbg_mem += t1 - t0; dbg_store += t2 - t1; dbg_bar += __builtin_amdgcn_s_memtime() - t2; }
        }
        if ((FL & FL_STAMP) && p.stamps && lane == 0) {
            unsigned long long* d = p.stamps + ((size_t)blockIdx.x * 8 + (wave8 < 8 ? wave8 : 7)) * 16;
            d[9] = dbg_mem; d[10] = dbg_store; d[11] = dbg_bar; d[12] = dbg_issue; d[13] = dbg_dma;
        }
        __builtin_amdgcn_s_setprio(0);   // staging is over: in the epilogue the accumulator waves are the ones with work to do
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
        const int abase4 = abase[0] - half + 8 * wk + 4 * half;     // DB4: this lane's [parity][4] quad of its wave's 8-channel block
        unsigned long long dbg_cbar = 0;
        constexpr int NB = (DB && !DB4) ? KK * KPW : 1, NR = (DB && !DB4) ? KPW : 1;
        float wcur[NB], wnxt[NB], rcur[NR], rnxt[NR];
        const int ncol = n0 + l31;
        const bool colok = ncol < Cout;                 // a column beyond Cout reads the zero block with zero strides
        const float* wq = colok ? a.w + (size_t)b0 * a.w_batch_stride + ncol : p.zeros16;
        const float* rq = (colok && has_res) ? a.res_w + ncol : p.zeros16;
        const size_t wcs = colok ? (size_t)Cout : 0, wts = colok ? (size_t)Cin * Cout : 0;
        auto fetch_w = [&](int i, float (&w)[NB], float (&r)[NR]) {   // this wave's rows of chunk i: channel c of every tap (+ res_conv row)
#pragma unroll
            for (int kk = 0; kk < KPW; ++kk) {
                int c = i * CC + 2 * (kk0 + kk) + half;
                c = c < Cin ? c : Cin - 1;               // past the Cin tail the window holds zeros: any finite weight will do
                const float* q = wq + (size_t)c * wcs;
#pragma unroll
                for (int tap = 0; tap < KK; ++tap) w[tap * KPW + kk] = q[(size_t)tap * wts];
                if (has_res) r[kk] = rq[(size_t)c * wcs];
            }
        };
        // DB4: registers A / B, each the nine taps' float4 (component j = k-step kk0 + j) of one chunk, plus the res_conv quad
        constexpr int NQ = DB4 ? KK : 1;
        f32x4 wA[NQ], wB[NQ], rA, rB;
        const unsigned lane_off4 = DB4 ? (unsigned)(half * Cout + ncol) * 4u : 0u;       // floats; DB4 launches have Cout % 32 == 0: every column exists
        const int wk_u = __builtin_amdgcn_readfirstlane(wk);
        auto fetch4 = [&](int i, f32x4 (&w)[NQ], f32x4& r) {
            // uniform base: chunk i, this wave's K-quarter = row block c8 = i * (CC / 8) + wk of [tap][Cin/8][half][Cout][4]
            const size_t blk = (size_t)((i * (CC / 8) + wk_u) * 2) * (size_t)Cout * 4;
            const float* ub = a.w4 + blk;
#pragma unroll
            for (int tap = 0; tap < NQ; ++tap) w[tap] = *reinterpret_cast<const f32x4*>(ub + (size_t)tap * ((size_t)Cin * Cout) + lane_off4);
            if (has_res) r = *reinterpret_cast<const f32x4*>(a.res_w4 + blk + lane_off4);
        };
        if (DB4) fetch4(0, wA, rA);
        else if (DB) fetch_w(0, wcur, rcur);
        if (FL & FL_STAMP) conv_stamp(p, 1);
        gn_tables(false, 0, 1);
        if (meet && tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(arrival) :: "memory");
            smem[p.o_epoch] = __uint_as_float(arrival / (unsigned)p.gsz + 1u);
        }
        __syncthreads();        // stage 0 ready
        if (FL & FL_STAMP) conv_stamp(p, 4);
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
        if constexpr (DB4) {
            // one chunk on register set `wc`, the next chunk's weights requested into `wn` first: they are not read before the NEXT chunk
            // (a whole chunk of MFMAs of lookahead), and nothing is copied.  The A operands of tap t + 1 are read from LDS before the
            // MFMAs of tap t are issued.
            auto chunk = [&](int i, f32x4 (&wc)[NQ], f32x4& rc, f32x4 (&wn)[NQ], f32x4& rn) {
                const float* patch = patch0 + (i & 1) * p.patch_stride;
                // UNCONDITIONAL (the last chunk re-requests itself): s_waitcnt takes an immediate, so with a conditional request the
                // compiler must count for the path that skipped it -- vmcnt(8) instead of vmcnt(17) in front of the first MFMA, i.e. a wait
                // for the request issued two lines above
                fetch4(i + 1 < nchunks ? i + 1 : i, wn, rn);
                __builtin_amdgcn_sched_barrier(0);       // the requests stay HERE: left alone, the scheduler sinks them behind the chunk's MFMAs
                static_assert(!DB4 || KPW == 4, "one 16-byte operand read = the four k-steps of a wave");
                (void)patch;
                // Pinned schedule: the operand quad of tap t + 1 is requested (asm: the compiler neither counts nor moves it), then the four
                // MFMAs of tap t run -- 256 cycles of cover for the LDS round trip --, then the wait.  Left to the scheduler the reads
                // ended up directly in front of their first use (s_waitcnt lgkmcnt right behind ds_read, every eighth MFMA).
                const unsigned pl = smem_lds + 4u * (unsigned)(p.o_patch + (i & 1) * p.patch_stride + abase4);
                const unsigned prow = 4u * (unsigned)(PW * CS);
                auto lds_quad = [&](f32x4& dst, int tapn) {
                    const unsigned ad = pl + (unsigned)(tapn / KS) * prow;
                    switch (tapn % KS) {         // column offset as an immediate
                        case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(ad) : "memory"); break;
                        case 1: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(4 * CS) : "memory"); break;
                        default: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(8 * CS) : "memory"); break;
                    }
                };
                f32x4 av, an;
                lds_quad(av, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av) :: "memory");
                an = av;
#pragma unroll
                for (int tap = 0; tap < KK; ++tap) {
                    // (after the last tap: the centre tap again, for the fused res_conv)
                    if (tap + 1 < KK || has_res) lds_quad(an, tap + 1 < KK ? tap + 1 : KK / 2);
                    __builtin_amdgcn_sched_barrier(0);
                    if (DB4_CHAINS == 2) {
                        // two independent accumulator chains: an MFMA that waits for its predecessor's result seems to hold the SIMD's issue
                        // port while it waits, and the staging waves that share the SIMD starve (stamps, round 3)
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], wc[tap][0], acc[0][0], 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], wc[tap][1], acc2, 0, 0, 0);
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], wc[tap][2], acc[0][0], 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], wc[tap][3], acc2, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int kk = 0; kk < KPW; ++kk) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], wc[tap][kk], acc[0][0], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap + 1 < KK || has_res) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(an) :: "memory");
                    av = an;
                }
                if (has_res) {
#pragma unroll
                    for (int kk = 0; kk < KPW; ++kk) accr[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], rc[kk], accr[0][0], 0, 0, 0);
                }
                unsigned long long t0 = ((FL & FL_STAMP) && p.stamps) ? __builtin_amdgcn_s_memtime() : 0ull;
                loader_handover();                       // not __syncthreads(): its vmcnt(0) would wait for the weights requested above
                if ((FL & FL_STAMP) && p.stamps) dbg_cbar += __builtin_amdgcn_s_memtime() - t0;
            };
            for (int i = 0; i < nchunks; i += 2) {
                chunk(i, wA, rA, wB, rB);
                if (i + 1 < nchunks) chunk(i + 1, wB, rB, wA, rA);
            }
            if (DB4_CHAINS == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][0][r] += acc2[r];
            }
        } else if constexpr (BF3) {
            // lane (row / column l31, half): channels 8 half .. 8 half + 7 of each 16-channel k-step, of its pixel (A) and of its column (B)
            int ab3[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ab3[mt] = abase[mt] - half + 4 * half;    // (16 bytes per half: eight bf16)
            const int bb3 = 4 * (half * BN + wn * NT * 32 + l31);    // floats: 16 bytes per (channel block, column)
            static_assert(CC == 16, "one k-step of sixteen channels per tap");
            typedef union { uint4 u; bf16x8_t v; } opnd_t;
            for (int i = 0; i < nchunks; ++i) {
                const float* patch = patch0 + (i & 1) * p.patch_stride;
                const float* wl = wl0 + (i % p.nwb) * p.wl_stride + bb3;
                // (The scheduler is left alone here.  A pinned schedule -- operands of tap t + 1 requested, then the 3 MT NT MFMAs of tap t
                // back to back -- measured 376 images/s against 421 on the SD-VAE decode: the staging waves' VALU work does not overlap the
                // MFMAs of the wave that shares their SIMD, DESIGN.md section 5, and a dense MFMA stream only moves the waiting around.)
#pragma unroll
                for (int tap = 0; tap < KK; ++tap) {
                    const int tapoff = ((tap / KS) * PW + (tap % KS)) * CS;
                    opnd_t A[MT][2], Bq[NT][2];       // [tile][hi / lo]
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {  // split by the staging waves: 16 bytes of hi parts, 16 bytes of lo parts per pixel and half
                        const char* pc = reinterpret_cast<const char*>(patch + ab3[mt] + tapoff);
                        A[mt][0].u = *reinterpret_cast<const uint4*>(pc);
                        A[mt][1].u = *reinterpret_cast<const uint4*>(pc + 2 * CC);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {  // split at pack time, staged by LDS-DMA
                        Bq[nt][0].u = *reinterpret_cast<const uint4*>(wl + 4 * ((tap * 2) * BN + nt * 32));
                        Bq[nt][1].u = *reinterpret_cast<const uint4*>(wl + 4 * (((KK + tap) * 2) * BN + nt * 32));
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][1].v, Bq[nt][0].v, acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][0].v, Bq[nt][1].v, acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][0].v, Bq[nt][0].v, acc[mt][nt], 0, 0, 0);
                        }
                }
                __syncthreads();    // stage i consumed; stage i + 1 (if any) ready
            }
        } else {
        for (int i = 0; i < nchunks; ++i) {
            const float* patch = patch0 + (i & 1) * p.patch_stride;
            const float* wl = wl0 + (i % p.nwb) * p.wl_stride + bbase;
            if (DB && i + 1 < nchunks) fetch_w(i + 1, wnxt, rnxt);
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
                    for (int nt = 0; nt < NT; ++nt) bv[nt] = DB ? wcur[tap * KPW + kk] : wl[(tap * CC + k) * BN + nt * 32];
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
                    for (int nt = 0; nt < NT; ++nt) bv[nt] = DB ? rcur[kk] : wl[(KK * CC + k) * BN + nt * 32];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            accr[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[nt], accr[mt][nt], 0, 0, 0);
                }
            }
            unsigned long long t0 = ((FL & FL_STAMP) && p.stamps) ? __builtin_amdgcn_s_memtime() : 0ull;
            if (DB) {
                loader_handover();                       // not __syncthreads(): its vmcnt(0) would wait for the slab prefetched above
#pragma unroll
                for (int j = 0; j < NB; ++j) wcur[j] = wnxt[j];
#pragma unroll
                for (int j = 0; j < NR; ++j) rcur[j] = rnxt[j];
            } else {
                __syncthreads();    // stage i consumed; stage i+1 (if any) ready
            }
            if ((FL & FL_STAMP) && p.stamps) dbg_cbar += __builtin_amdgcn_s_memtime() - t0;
        }
        }
        if ((FL & FL_STAMP) && p.stamps && lane == 0) p.stamps[((size_t)blockIdx.x * 8 + wave8) * 16 + 11] = dbg_cbar;
        if (FL & FL_STAMP) conv_stamp(p, 5);
    }
    conv_epilogue<WM, WN, WK, MT, NT, FL>(p, acc, accr, smem, tid, lane, wave, b0, y0, x0, n0, tx, ty, consumer, NTHR, pre, true);
}

// ---------------------------------------------------------------------------------------------------
static float* g_zeros16 = nullptr;
const float* conv_zeros16() { return g_zeros16; }

#define FC_PIPE_TILES(X, KS)               \
    X(TILE_M128N32, 4, 1, 1, 1, 1, 16, KS, 4)  \
    X(TILE_M128N64, 4, 1, 1, 1, 2, 16, KS, 4)  \
    X(TILE_M64N32K2, 2, 1, 2, 1, 1, 32, KS, (KS == 1 ? 4 : 8)) \
    X(TILE_M32N32K4, 1, 1, 4, 1, 1, 32, KS, (KS == 1 ? 4 : 8)) \
    X(TILE_M64N64K2, 2, 1, 2, 1, 2, 16, KS, 4) \
    X(TILE_M256N64, 4, 1, 1, 2, 2, 16, KS, 4)

template <int KS>
static int pipe_attr_ks() {
#define X(T, WM, WN, WK, MT, NT, CC, K, NL)                                                                               \
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pipe_kernel<WM, WN, WK, MT, NT, CC, 8, K, NL>),        \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_PIPE_TILES(X, KS)
#undef X
    return FC_OK;
}

// Lean flavours (conv_dev.h FL_*) of the tiles the U-Net runs on: 3x3 as plain / + fused res_conv / + fused tail, 1x1 and 2x2 plain.
// Everything else goes to the FL_ALL instantiation above.
#define FC_LEAN_TILES(X, KS)               \
    X(TILE_M128N32, 4, 1, 1, 1, 1, 16, KS, 4)  \
    X(TILE_M64N32K2, 2, 1, 2, 1, 1, 32, KS, (KS == 1 ? 4 : 8)) \
    X(TILE_M32N32K4, 1, 1, 4, 1, 1, 32, KS, (KS == 1 ? 4 : 8))

#define FC_LEAN_FLAVOURS_3(X) X(0) X(FL_POSTOP) X(FL_STATS) X(FL_STATS | FL_STAMP) X(FL_STATS | FL_CAT | FL_STAMP) X(FL_STATS | FL_RES) X(FL_STATS | FL_CAT) X(FL_STATS | FL_RES | FL_CAT) X(FL_STATS | FL_XF) \
    X(FL_STATS | FL_XF | FL_FIN) X(FL_STATS | FL_XF | FL_FIN | FL_GN1) X(FL_STATS | FL_XF | FL_FIN | FL_MEET) X(FL_STATS | FL_XF | FL_FIN | FL_GN1 | FL_MEET) \
    X(FL_STATS | FL_XF | FL_FIN | FL_STAMP) X(FL_STATS | FL_XF | FL_FIN | FL_GN1 | FL_STAMP) X(FL_STATS | FL_XF | FL_FIN | FL_MEET | FL_STAMP) X(FL_STATS | FL_XF | FL_FIN | FL_GN1 | FL_MEET | FL_STAMP)   /* diagnostics: tools/fin_stamps.py */
#define FC_LEAN_FLAVOURS_1(X) X(0) X(FL_POSTOP) X(FL_XF) X(FL_STATS) X(FL_STATS | FL_XF)

// the 64-column tiles the 1x1 projections of the larger models run on (to_qkv and its data gradient): lean flavours for 1x1 only
#define FC_LEAN_TILES_WIDE1(X)             \
    X(TILE_M128N64, 4, 1, 1, 1, 2, 16, 1, 4)   \
    X(TILE_M256N64, 4, 1, 1, 2, 2, 16, 1, 4)

// the lean kernels keep four window elements per staging thread (the all-in-one ones eight): every U-Net layer needs at most four, and
// the staging code is unrolled per element
constexpr int kLeanNPL = 4;
template <int KS, int FL>
static int lean_attr() {
#define X(T, WM, WN, WK, MT, NT, CC, K, NL)                                                                               \
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pipe_kernel<WM, WN, WK, MT, NT, CC, kLeanNPL, K, NL, FL>),  \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_LEAN_TILES(X, KS)
#undef X
    return FC_OK;
}

// the k-step-quad (FL_W4) flavours exist for the one tile that feeds its weights from registers: M32N32K4 at 3x3
template <int FL>
static int lean_attr_db4() {
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pipe_kernel<1, 1, 4, 1, 1, 32, kLeanNPL, 3, 8, FL | FL_W4>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return FC_OK;
}

template <int FL>
static int lean_attr_wide1() {
#define X(T, WM, WN, WK, MT, NT, CC, K, NL)                                                                               \
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pipe_kernel<WM, WN, WK, MT, NT, CC, kLeanNPL, K, NL, FL>),  \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_LEAN_TILES_WIDE1(X)
#undef X
    return FC_OK;
}

// Either launches `kernel` or, when `occ` is given, asks the runtime how many workgroups of it one CU holds at this LDS size (the
// fused tail's residency condition is derived from that, per instantiation -- conv_igemm.hip).
template <class K>
static int launch_or_query(K kernel, int nthr, const ConvDev& d, int grid, size_t lds, hipStream_t s, int* occ) {
    if (occ) {
        FC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, kernel, nthr, lds));
        return FC_OK;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(nthr), lds, s, d);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
template <int FL>
static int lean_launch_wide1(const ConvDev& d, int tile, int grid, size_t lds, hipStream_t s, int* occ) {
    switch (tile) {
#define X(T, WM, WN, WK, MT, NT, CC, K, NL)                                                                                       \
    case T:                                                                                                                       \
        if (d.P * (CC / 4) > 64 * NL * kLeanNPL) return -1;                                                                       \
        return launch_or_query((conv_pipe_kernel<WM, WN, WK, MT, NT, CC, kLeanNPL, K, NL, FL>), 256 + 64 * NL, d, grid, lds, s, occ);
        FC_LEAN_TILES_WIDE1(X)
#undef X
        default: return -1;
    }
}

template <int FL>
static int lean_launch_db4(const ConvDev& d, int tile, int grid, size_t lds, hipStream_t s, int* occ) {
    if (tile != TILE_M32N32K4 || d.P * (32 / 4) > 64 * 8 * kLeanNPL) return -1;
    return launch_or_query((conv_pipe_kernel<1, 1, 4, 1, 1, 32, kLeanNPL, 3, 8, FL | FL_W4>), 256 + 64 * 8, d, grid, lds, s, occ);
}

template <int KS, int FL>
static int lean_launch(const ConvDev& d, int tile, int grid, size_t lds, hipStream_t s, int* occ) {
    switch (tile) {
#define X(T, WM, WN, WK, MT, NT, CC, K, NL)                                                                                       \
    case T:                                                                                                                       \
        if (d.P * (CC / 4) > 64 * NL * kLeanNPL) return -1;    /* more window elements per staging thread than a lean kernel keeps */ \
        return launch_or_query((conv_pipe_kernel<WM, WN, WK, MT, NT, CC, kLeanNPL, K, NL, FL>), 256 + 64 * NL, d, grid, lds, s, occ);
        FC_LEAN_TILES(X, KS)
#undef X
        default: return -1;      // no lean flavour of this tile
    }
}

// (four staging waves as in the fp32 form: eight measured 364 against 380 images/s on the SD-VAE decode)
#define FC_BF3_TILES(X, KS)                \
    X(TILE_M128N32, 4, 1, 1, 1, 1, 16, KS, 4)  \
    X(TILE_M128N64, 4, 1, 1, 1, 2, 16, KS, 4)  \
    X(TILE_M256N64, 4, 1, 1, 2, 2, 16, KS, 4)
template <int KS>
static int bf3_attr_ks() {
#define X(T, WM, WN, WK, MT, NT, CC, K, NL)                                                                                  \
    FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pipe_kernel<WM, WN, WK, MT, NT, CC, 8, K, NL, FL_ALL, 1>), \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FC_BF3_TILES(X, KS)
#undef X
    return FC_OK;
}
template <int KS>
static int bf3_launch_ks(const ConvDev& d, int tile, int grid, size_t lds, hipStream_t s, int* occ) {
    switch (tile) {
#define X(T, WM, WN, WK, MT, NT, CC, K, NL) \
    case T: return launch_or_query((conv_pipe_kernel<WM, WN, WK, MT, NT, CC, 8, K, NL, FL_ALL, 1>), 256 + 64 * NL, d, grid, lds, s, occ);
        FC_BF3_TILES(X, KS)
#undef X
        default: return fail(FC_E_ARG, "conv: no split-bf16 form of this tile");
    }
}

int conv_pipe_init() {
    static bool done = false;
    if (done) return FC_OK;
    FC_TRY(bf3_attr_ks<1>());
    FC_TRY(bf3_attr_ks<2>());
    FC_TRY(bf3_attr_ks<3>());
    FC_HIP(hipMalloc(reinterpret_cast<void**>(&g_zeros16), 256));
    FC_HIP(hipMemset(g_zeros16, 0, 256));
    FC_TRY(pipe_attr_ks<1>());
    FC_TRY(pipe_attr_ks<2>());
    FC_TRY(pipe_attr_ks<3>());
    FC_TRY(pipe_attr_ks<5>());
#define X(F) FC_TRY((lean_attr<3, (F)>())); FC_TRY((lean_attr_db4<(F)>()));
    FC_LEAN_FLAVOURS_3(X)
#undef X
#define X(F) FC_TRY((lean_attr<1, (F)>())); FC_TRY((lean_attr_wide1<(F)>()));
    FC_LEAN_FLAVOURS_1(X)
#undef X
    FC_TRY((lean_attr<2, 0>()));
    done = true;
    return FC_OK;
}

bool conv_pipe_supports_ks(int ks) { return ks == 1 || ks == 2 || ks == 3 || ks == 5; }

template <int KS>
static int pipe_launch_ks(const ConvDev& d, int tile, int grid, size_t lds, hipStream_t s, int* occ) {
    switch (tile) {
#define X(T, WM, WN, WK, MT, NT, CC, K, NL) \
    case T: return launch_or_query((conv_pipe_kernel<WM, WN, WK, MT, NT, CC, 8, K, NL>), 256 + 64 * NL, d, grid, lds, s, occ);
        FC_PIPE_TILES(X, KS)
#undef X
        default: return fail(FC_E_ARG, "conv: bad tile id");
    }
}

static int conv_pipe_dispatch(const ConvDev& d, int tile, int grid, size_t lds, hipStream_t s, int* occ) {
    if (d.bf3) return d.a.KS == 1 ? bf3_launch_ks<1>(d, tile, grid, lds, s, occ) : d.a.KS == 2 ? bf3_launch_ks<2>(d, tile, grid, lds, s, occ) : bf3_launch_ks<3>(d, tile, grid, lds, s, occ);
    static const bool lean = [] { const char* e = std::getenv("FLOCODER_AMD_LEAN_KERNELS"); return !(e && std::string(e) == "0"); }();
    if (lean) {                  // the smallest flavour that covers this launch
        const bool small_tile = tile == TILE_M32N32K4;
        const int need = (d.a.fin.gamma ? FL_FIN : 0) | (d.a.res_out ? FL_RES : 0) | (d.a.stats_post ? FL_POST : 0) | (d.any_xf ? FL_XF : 0) |
                         (d.a.s1.C ? FL_CAT : 0) | (d.stamps ? FL_STAMP : 0) | (d.a.stats_out ? FL_STATS : 0) | (d.a.fin.gn1_out ? FL_GN1 : 0) |
                         ((d.a.out_act || d.a.add) ? FL_POSTOP : 0) | ((d.o_out < 0 || d.a.fin.raw) ? FL_NARROW : 0) |   // (a training tail that keeps its raw output also goes to the all-in-one kernel)
                         ((d.TB > 1 && !small_tile) ? FL_MULTI : 0) |
                         ((d.a.fin.gamma && !d.fin_local) ? FL_MEET : 0);
        int r = -1;
        if (d.a.KS == 3) {
            // register-fed weights in the k-step-quad layout: whole 32-channel chunks, whole 32-column tiles, the centre tap at (1, 1)
            static const bool no_w4 = std::getenv("FLOCODER_AMD_NO_W4") != nullptr;
            const bool w4 = !no_w4 && small_tile && d.a.w4 && d.a.Cin % 32 == 0 && d.a.Cout % 32 == 0 && d.a.pad == 1 && !d.a.w_batch_stride &&
                            (!d.a.res_out || d.a.res_w4);
            if (w4) {
#define X(F) if (r == -1 && need == (F)) r = lean_launch_db4<(F)>(d, tile, grid, lds, s, occ);
                FC_LEAN_FLAVOURS_3(X)
#undef X
            }
#define X(F) if (r == -1 && need == (F)) r = lean_launch<3, (F)>(d, tile, grid, lds, s, occ);
            FC_LEAN_FLAVOURS_3(X)
#undef X
        } else if (d.a.KS == 1) {
#define X(F) if (r == -1 && need == (F)) { r = lean_launch<1, (F)>(d, tile, grid, lds, s, occ); if (r == -1) r = lean_launch_wide1<(F)>(d, tile, grid, lds, s, occ); }
            FC_LEAN_FLAVOURS_1(X)
#undef X
        } else if (d.a.KS == 2 && need == 0) {
            r = lean_launch<2, 0>(d, tile, grid, lds, s, occ);
        }
        if (r != -1) return r;
    }
    switch (d.a.KS) {
        case 1: return pipe_launch_ks<1>(d, tile, grid, lds, s, occ);
        case 2: return pipe_launch_ks<2>(d, tile, grid, lds, s, occ);
        case 3: return pipe_launch_ks<3>(d, tile, grid, lds, s, occ);
        case 5: return pipe_launch_ks<5>(d, tile, grid, lds, s, occ);
    }
    return fail(FC_E_SHAPE, "conv: kernel size not instantiated in the pipelined kernel");
}

int conv_pipe_launch(const ConvDev& d, int tile, int grid, size_t lds, hipStream_t s) { return conv_pipe_dispatch(d, tile, grid, lds, s, nullptr); }

// Workgroups of the instantiation this launch would go to that ONE CU holds at once (registers, waves and LDS as the runtime counts
// them), cached per (tile, kernel size, flavour mask, LDS bytes).  0 = the query failed.
int conv_pipe_blocks_per_cu(const ConvDev& d, int tile, size_t lds) {
    static std::map<std::tuple<int, int, int, size_t>, int> cache;
    static std::mutex mu;
    const int mask = (d.a.fin.gamma ? FL_FIN : 0) | (d.a.res_out ? FL_RES : 0) | (d.a.stats_post ? FL_POST : 0) | (d.any_xf ? FL_XF : 0) |
                     (d.a.s1.C ? FL_CAT : 0) | (d.stamps ? FL_STAMP : 0) | (d.a.stats_out ? FL_STATS : 0) | (d.a.fin.gn1_out ? FL_GN1 : 0) |
                     ((d.a.out_act || d.a.add) ? FL_POSTOP : 0) | ((d.o_out < 0 || d.a.fin.raw) ? FL_NARROW : 0) | (d.TB > 1 ? FL_MULTI : 0) |
                     ((d.a.fin.gamma && !d.fin_local) ? FL_MEET : 0);
    const auto key = std::make_tuple(tile, d.a.KS, mask, lds);
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int occ = 0;
    if (conv_pipe_dispatch(d, tile, 1, lds, nullptr, &occ) != FC_OK) occ = 0;
    cache[key] = occ;
    return occ;
}

}  // namespace fc
