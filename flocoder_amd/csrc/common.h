// Shared declarations for the gfx950 kernels and the native plan builder.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/flocoder_amd.h"

namespace fc {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define FC_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return ::fc::fail(FC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
    } while (0)

#define FC_TRY(expr)                \
    do {                            \
        int _r = (expr);            \
        if (_r != FC_OK) return _r; \
    } while (0)

// Long-lived device buffers (devmem.hip): hipMalloc / hipFree, or poisoned + fenced allocations under FLOCODER_AMD_POISON=1
int dev_alloc(void** out, size_t bytes, const char* tag);
void dev_free(void* p);

static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics travel between kernels as per-tile partials: stats[b][g][t] = (mean_t, M2_t),
// every slot t covering exactly n_t elements.  Producers never use atomics, consumers combine the T
// slots in a fixed order (Chan's parallel update with equal counts), so results are run-to-run
// bit-identical.
// ---------------------------------------------------------------------------------------------
struct StatsRef {
    float* p = nullptr;  // [B][G][T][2]
    int G = 0, T = 0;
    float n_t = 0.f;
};

// How a consumer turns a stored raw tensor into its logical input while staging it:
//   mode 0: identity          mode 1: GroupNorm affine            mode 2: GroupNorm affine, FiLM, SiLU
struct SrcXform {
    int mode = 0;
    const float* stats = nullptr;
    int G = 1, T = 1;
    float n_t = 1.f;
    const float* gamma = nullptr;
    const float* beta = nullptr;
    const float* ss = nullptr;  // [B][ss_stride]: scale at ss[c], shift at ss[C + c]   (unet.py:92 chunk(2,dim=1))
    int ss_stride = 0;
    float eps = 1e-5f;
};

struct ConvSrc {
    const float* p = nullptr;  // NHWC [B][Hs][Ws][C]
    int C = 0;
    SrcXform xf;
};

// Fused tail of a Block (unet.py:66-72,96): instead of storing the raw convolution output, the workgroups of one sample (or
// sample group) exchange their GroupNorm partials through `stats_out` and a per-group arrival counter, then finish
// out = SiLU(GN(acc)) + res straight from the accumulators.  Needs every workgroup of the launch resident at once.
struct ConvFin {
    const float* gamma = nullptr;   // != nullptr switches the tail on
    const float* beta = nullptr;
    float eps = 1e-5f;
    const float* res = nullptr;     // NHWC [B][H][W][Cout] residual or null
    float* gn1_out = nullptr;       // optional GroupNorm(1) partials of the final value: [B][1][T1][2]
    unsigned* sync = nullptr;       // one arrival counter per sample group, never reset: arrival number / group size + 1 = this launch's epoch
    unsigned long long* gran = nullptr;   // [B][G][T][2] tagged granules {epoch << 32 | float bits}: the partials AS the hand-off (no flag, no fence)
    int* err = nullptr;             // set to 1 if a wait ever times out
    float* raw = nullptr;           // training plans: the convolution output BEFORE the tail (NHWC, the backward's h2) is stored too, and
                                    // the partial statistics also land in stats_out in the ordinary [B][G][T][2] form
};

struct ConvArgs {
    ConvSrc s0, s1;           // s1.C == 0: no channel concat
    const float* w = nullptr;  // packed [KS*KS][Cin][Cout]
    const float* bias = nullptr;
    float* out = nullptr;      // NHWC [B][H][W][Cout]
    int out_act = 0;           // 1: SiLU on (acc + bias) before `add`
    const float* add = nullptr;  // NHWC [B][H][W][Cout] added last (residuals)
    float* stats_out = nullptr;  // partial GroupNorm stats of `out`: after bias, before act/add -- or of the final value when stats_post
    int Gout = 0;
    int stats_post = 0;
    // fused 1x1 projection of the (untransformed) centre tap: ResnetBlock.res_conv, unet.py:86,96
    const float* res_w = nullptr;  // [Cin][Cout]
    const float* res_b = nullptr;
    float* res_out = nullptr;
    int B = 0, H = 0, W = 0;   // output extent
    int Hs = 0, Ws = 0;        // source extent (before the optional nearest x2)
    int Cin = 0, Cout = 0;
    int KS = 1, pad = 0, stride = 1, ups = 0;
    int w_batch_stride = 0;    // != 0: per-sample weights (w + b*stride), tiles then hold one sample
    // the same weights in the k-step-quad layout [tap][Cin/8][half][Cout][4] (pack kind 6), or null: the 32-row tile at 3x3 loads its MFMA B
    // operands straight from it, 16 bytes per lane (conv_pipe.hip)
    const float* w4 = nullptr;
    const float* res_w4 = nullptr;
    const float* w_b3 = nullptr;   // the weights as split bf16 (pack kind 8) or null: needed by prec == 1
    // One parity class of "nearest x2 upsampling + 3x3" as a 2x2 convolution on the low-resolution input (pack kinds 9 / 10): the window
    // starts pad_y / pad_x above / left of the output pixel (-1: `pad` for both) and output pixel (y, x) of this launch's H x W grid is
    // pixel (2y + out_oy, 2x + out_ox) of a 2H x 2W image when out_sh = 1; its GroupNorm partials go to slot stats_toff of stats_tmul
    // launches that share one statistics tensor.
    int pad_y = -1, pad_x = -1;
    int out_sh = 0, out_oy = 0, out_ox = 0;
    int stats_tmul = 1, stats_toff = 0;
    int par4 = 0;              // 1: ONE launch carries the four parity classes (quarter q of the grid = class q): w points at [4][4 taps][Cin][Cout]
                               // (w_b3 likewise) and pad_y / pad_x / out_oy / out_ox / stats_toff are derived per workgroup (out_sh = 1, stats_tmul = 4)
    int prec = 0;              // 1: split-bf16 arithmetic where the tile has that form (codec decoders on request; 0 = exact fp32, always for the U-Net)
    ConvFin fin;
};

// Tile configurations of the implicit-GEMM kernel (see conv_igemm.hip).
enum ConvTile { TILE_AUTO = -1, TILE_M128N32 = 0, TILE_M128N64 = 1, TILE_M64N32K2 = 2, TILE_M32N32K4 = 3, TILE_M64N64K2 = 4,
                TILE_M256N64 = 5, TILE_COUNT = 6 };

struct ConvGeom {  // filled by conv_plan(): what a consumer must know about `stats_out`
    int tile = 0, grid = 0, T = 0, pipe = 0;
    float n_t = 0.f;
    size_t lds = 0;
    int T1 = 0;        // fused tail: slots / count of the GroupNorm(1) partials of the final value
    float n_t1 = 0.f;
    int groups = 0;    // fused tail: arrival counters needed (sample groups)
    int fin_local = 0; // fused tail: every GroupNorm group of the output lies inside one workgroup's tile (no meeting, no residency condition)
};
bool conv_fin_possible(const ConvArgs& a, int tile);   // the fused tail's residency / shape conditions hold for this launch
int conv_init();
unsigned long long* conv_stamp_buffer();
void conv_set_stamp_buffer(unsigned long long* p);  // diagnostics: phase stamps of the pipelined kernel
int conv_plan(const ConvArgs& a, int tile, ConvGeom* g);
int conv_launch(const ConvArgs& a, int tile, hipStream_t s);

// ---- elementwise / small kernels (elementwise.hip, attention.hip, temb.hip) -------------------
struct FinalizeArgs {       // y = act(gn(h)) + res, optional GroupNorm(1) partials of y
    const float* h = nullptr;
    SrcXform xf;            // mode 1 or 2 (ss may be null)
    const float* res = nullptr;
    SrcXform xf_res;        // mode 1: the residual is GroupNorm'ed too (EncDecResidualBlock.downsample, codecs.py:164-167)
    int act_after_add = 0;  // 1: y = SiLU(gn(h) + res') instead of SiLU(gn(h)) + res   (codecs.py:204-206)
    float* y = nullptr;
    float* stats_out = nullptr;  // G = 1
    int B = 0, HW = 0, C = 0;
};
int finalize_blocks_per_sample(int HW, int C);
int finalize_launch(const FinalizeArgs& a, hipStream_t s);
size_t finalize_lds_bytes(const FinalizeArgs& a);
int finalize_table_launch(const FinalizeArgs* jobs_dev, const int* bps_dev, const int2* blocks_dev, int nblocks, size_t lds_bytes, hipStream_t s);
int gn_stats_launch(const float* x_nhwc, float* stats /*[B][G][1][2]*/, int B, int HW, int C, int G, hipStream_t s);
// [B][G][T][2] partials of n_t elements each -> [B][G][1][2] covering n_t*T elements
int gn_fold_launch(const float* in, float* out, int B, int G, int T, float n_t, hipStream_t s);

// Conditioning of every U-Net evaluation of an integration, computed up front (fc_unet_integrate): the time grid is known before
// the first step, so the sinusoid -> time MLP (+ class MLP) -> 19 FiLM projections chain (unet.py:310-316,79-82) runs ONCE over all
// (evaluation, row) pairs instead of as three launches at the head of each of the 64..396 forwards.  `all` holds
// [evaluation][row][S] scale / shift rows; init_conv, the first launch of a forward, copies evaluation *evalc's slice into the plan's
// table, and final_conv, the last one, advances *evalc.
struct CondFetch {
    const float* all = nullptr;    // null: the per-forward conditioning launches run as usual
    const int* evalc = nullptr;
    float* dst = nullptr;
    int n4 = 0;                    // float4 elements per evaluation (rows * S / 4)
};

int init_conv_launch(const CondFetch& fetch, const float* x_nchw, int x_batch_mod, const float* w /*[Cin][Cout]*/, const float* bias, float* out_nhwc,
                     int B, int Cin, int HW, int Cout, hipStream_t s);
// Legacy Euler step folded into final_conv (sampling.py:43-48): y += v * dt from the kernel that produces v, and block 0 publishes the
// next interval's time exactly as ode_time_launch would (reads ts[*step], writes tvec, advances the counter)
struct EulerTail {
    int* evalc = nullptr;          // evaluation counter of the integrator (advanced by every final_conv), or null
    float* y = nullptr;            // NCHW state, updated in place; null = plain final_conv
    float dt = 0.f;
    int* step = nullptr;
    const float* ts = nullptr;
    float t_scale = 1.f;
    float* sc = nullptr;
    float* tvec = nullptr;
    int rows = 0;
};
int final_conv_launch(const float* x_nhwc, const float* w /*[Cin][Cout]*/, const float* bias, float* out_nchw, int B, int Cin,
                      int HW, int Cout, const EulerTail& tail, hipStream_t s);
int nchw_to_nhwc_launch(const float* src, float* dst, int B, int C, int HW, int Cpad, int src_batch_mod, hipStream_t s);
int nhwc_to_nchw_launch(const float* src, float* dst, int B, int C, int HW, int Cpad, hipStream_t s);
// nn.PixelShuffle(2) on NHWC: src [B][H][W][4C] (channel c*4 + i*2 + j) -> dst [B][2H][2W][C]
int pixel_shuffle2_nhwc_launch(const float* src, float* dst, int B, int H, int W, int C, hipStream_t s);
int bilinear_nhwc_launch(const float* src, float* dst, int B, int C, int Hs, int Ws, int Hd, int Wd, hipStream_t s);

struct TembArgs {
    const float* time = nullptr;       // [B]
    const int64_t* class_ids = nullptr;  // [B] or null; id < 0 -> no class term
    int class_batch_mod = 0;           // ids index = b % mod (CFG second half passes null_from)
    int null_from = 0;                 // rows >= null_from get no class embedding (CFG), 0 = off
    int rows_per_eval = 0;             // > 0: row b is row b % rows_per_eval of evaluation b / rows_per_eval (all evaluations of an
                                       // integration in one launch): time[b / rows_per_eval], class / null_from taken within the evaluation
    const float* freqs = nullptr;      // [dim/2] exp(-k ln(1e4)/(dim/2-1))
    const float *w1t, *b1, *w2t, *b2;  // time_mlp.1 [dim][td], time_mlp.3 [td][td]  (transposed: [in][out])
    const float *emb, *cw1t, *cb1, *cw2t, *cb2;  // class_cond_mlp.{0,1,3}
    int n_classes = 0;
    float* t_out = nullptr;            // [B][td]
    int B = 0, dim = 0, td = 0;
};
int temb_init();
// h, c1: [B][td] scratch for the two hidden layers
int temb_launch(const TembArgs& a, float* h, float* c1, hipStream_t s);
// ss[b][j] = sum_i silu(t[b][i]) * wt[i][j] + bias[j]   for the concatenation of every ResnetBlock.mlp
int ss_launch(const float* t, const float* wt, const float* bias, float* ss, int B, int td, int S, hipStream_t s);

// Linear attention core (unet.py:142-149) on qkv NHWC [B][n][3*heads*32]
int linattn_ctx_launch(const float* qkv, float* ctx /*[B][heads][32][32]*/, int B, int n, int heads, hipStream_t s);
int linattn_apply_launch(const float* qkv, const float* ctx, float* out /*[B][n][heads*32]*/, int B, int n, int heads,
                         hipStream_t s);
// The whole Residual(PreNorm(LinearAttention)) body up to to_out.0 in two launches, nothing but x in and y out (linattn_fused.hip)
struct LaArgs {
    const float* x = nullptr;      // NHWC [B][n][C], raw (the PreNorm's GroupNorm(1) is applied on load)
    SrcXform xf;                   // mode 1, G = 1: statistics / gamma / beta of fn.norm
    const float* wqkv = nullptr;   // packed [C][3*128]
    const float* wout = nullptr;   // packed [128][C]
    const float* wqkv4 = nullptr;  // wqkv in the k-step-quad layout [C/8][half][384][4] (pack kind 6) or null: the eight-wave la_head reads it
    const float* bout = nullptr;   // [C]
    float* ctx = nullptr;          // [B][4][32][32] scratch
    float* y = nullptr;            // NHWC [B][n][C]: to_out.0 output (before to_out.1's GroupNorm)
    float* stats_out = nullptr;    // GroupNorm(1) partials of y: [B][1][T][2], T = linattn_fused_tiles(n), n_t = linattn_fused_nt(n, C)
    int B = 0, n = 0, C = 0, heads = 4;
    // whole-module form (linattn_sample.hip): to_out.1's GroupNorm(1) parameters and the module output out = gn(y) + x
    const float* g2 = nullptr;
    const float* b2 = nullptr;
    float eps2 = 1e-5f;
    float* out = nullptr;
    float* part = nullptr;         // [B][heads][n][C] scratch: every head's share of to_out.0
    // fused close of the n >= 256 module (linattn_fused.hip, la_apply): with `gran` the apply launch also normalises (to_out.1) and adds x --
    // the workgroups of a sample exchange their (mean, M2) partials of y as tagged granules and WAIT for each other, so the launch needs
    // its whole grid resident (exclusive plan only: linattn_fused_meeting_ok); y / stats_out are then not written
    unsigned long long* gran = nullptr;   // [B][T][2] {epoch << 32 | float bits}, zero when allocated
    unsigned* sync = nullptr;             // [B] arrival counters (epoch = arrival / T + 1), zero when allocated
    int* err = nullptr;                   // the handle's error word: set when a wait gives up (the sample's output is NaN then)
    unsigned* tickets = nullptr;   // [B] arrival counters, zero when allocated: with them the module is ONE launch -- the workgroup of a sample that
                                   // arrives last (ticket % heads == heads - 1) adds the shares, normalises and writes `out`; nobody waits
};
int linattn_fused_init();
bool linattn_fused_supported(int n, int C, int heads);
int linattn_fused_tiles(int n);
float linattn_fused_nt(int n, int C);
int linattn_fused_launch(const LaArgs& a, hipStream_t s);
bool linattn_fused_meeting_ok(int B, int n, int C);   // may LaArgs::gran be set: the apply launch's whole grid is resident at this shape
// The whole Residual(PreNorm(LinearAttention)) module in two launches, a workgroup per (sample, head) then per sample (n <= 64 positions)
int linattn_sample_init();
bool linattn_sample_supported(int n, int C, int heads);
bool linattn_sample_one_launch(int n, int C);      // may LaArgs::tickets be set for this shape
int linattn_sample_launch(const LaArgs& a, hipStream_t s);
// Residual(PreNorm(Attention)) on the same kernels (g2 / b2 unused: to_out has no norm)
bool attn_sample_supported(int n, int C, int heads);
int attn_sample_launch(const LaArgs& a, hipStream_t s);
// SpatialNonLocalAttention (codecs.py:337-383) for a handful of channels: x NHWC [B][n][C] -> x + out_proj(softmax(rope(q) rope(k)^T) v)
int rope_attn_launch(const float* x, const float* wq, const float* bq, const float* wk, const float* bk, const float* wv, const float* bv,
                     const float* wo, const float* bo, float* out, int B, int n, int C, int Cr, hipStream_t s);
// Softmax attention core (unet.py:114-121), n <= 64
int attn_small_launch(const float* qkv, float* out, int B, int n, int heads, hipStream_t s);

// ---- ODE state updates on the NCHW boundary tensors (ode.hip) -----------------------------------
// Device-resident integrator state: `step` (interval counter), `ts` (time grid), `sc` = {t, dt} of the
// interval in flight.  All arithmetic is fp32 in the reference's operation order (sampling.py:43-48,74),
// with FMA contraction disabled, so a step is reproducible against the CPU oracle to rounding.
// First kernel of a step: reads ts[*step], publishes sc/tvec, then advances the counter.
int delay_launch(long long cycles, hipStream_t s);
// 2-D neighbourhood attention on a fused NHWC qkv tensor (natten.hip)
int na2d_launch(const float* qkv, float* out, const float* gamma, int B, int H, int W, int C, int heads, int ksize, int mode, hipStream_t s);
int ode_all_times_launch(const float* ts, int n_steps, int rk4, float t_scale, float* tv_out, hipStream_t s);
int ode_time_launch(int* step, const float* ts, float t_scale, int rk4, float* sc, float* tvec, int rows, hipStream_t s);
// v = cfg_on ? v_nc + cfg*(v_c - v_nc) : v   with v2 = [v_c ; v_nc] (n elements each)
int ode_euler_update_launch(float* x, const float* v2, int n, int cfg_on, float cfg, float dt, hipStream_t s);
// k_out = v ; xs = y + (full ? dt*k : dt*k/2) ; tvec[:] = (t + (tsel==1 ? dt/2 : dt)) * t_scale
int ode_rk4_stage_launch(const float* sc, const float* y, float* xs, float* k_out, const float* v2, int n, int cfg_on, float cfg,
                         int full, int tsel, float t_scale, float* tvec, int rows, hipStream_t s);
// y += (dt/6) * (k1 + 2*k2 + 2*k3 + v)
int ode_rk4_final_launch(const float* sc, float* y, const float* k1, const float* k2, const float* k3, const float* v2, int n,
                         int cfg_on, float cfg, hipStream_t s);

// ---- weight packing (pack.hip) ----------------------------------------------------------------
struct PackJob { const float* src; float* dst; int kind, a, b, c, d, e; size_t total; };
constexpr int kPackPerBlock = 4096;          // elements a workgroup of the table kernel moves
struct PackTable {                            // device-resident job list + (job, block) map of one batched launch
    PackJob* jobs = nullptr;
    int2* blocks = nullptr;
    int nblocks = 0;
    void release();
};
int pack_conv_launch(const float* oihw, float* dst /*[KK][I][O]*/, int O, int I, int KH, int KW, hipStream_t s);
// k-step-quad layout [KK][I/8][half][O][4] (I % 8 == 0): ConvArgs::w4
int pack_conv_k8_launch(const float* oihw, float* dst, int O, int I, int KK, hipStream_t s);
// split-bf16 copy: hi parts then lo parts, each [KK][Ipad/8][O][8] 16-bit values (Ipad a multiple of 16; KK * Ipad * O floats in all): ConvArgs::w_b3
int pack_conv_b3_launch(const float* oihw, float* dst, int O, int I, int KK, int Ipad, hipStream_t s);
// same with zero padding of either channel count: dst [KK][Ipad][Opad]
int pack_conv_pad_launch(const float* oihw, float* dst, int O, int I, int KK, int Opad, int Ipad, hipStream_t s);
// operand of the data-gradient pass (forward kernel on dY): [taps flipped][O][nci] for input channels ci0..ci0+nci
int pack_conv_dgrad_launch(const float* oihw, float* dst, int O, int I, int KS, int ci0, int nci, hipStream_t s);
int pack_table_build(std::vector<PackJob> jobs, PackTable* out);   // uploads; the table keeps raw device pointers of src / dst
int pack_table_launch(const PackTable& t, hipStream_t s);
// [B][rows][cols] -> [B][cols][rows]
int transpose_batched_launch(const float* src, float* dst, int B, int rows, int cols, hipStream_t s);
// in place: x[r][:] = softmax(scale * x[r][:]) for `rows` rows of `cols` floats
int softmax_rows_launch(float* x, long rows, int cols, float scale, hipStream_t s);
int pack_s2d_conv_launch(const float* oi, float* dst /*[4][C][O]*/, int O, int C, hipStream_t s);  // Downsample: (c p1 p2) -> 2x2 s2
int pack_transpose_launch(const float* src /*[R][Cc]*/, float* dst /*[Cc][R]*/, int R, int Cc, int dst_ld, int dst_col0,
                          hipStream_t s);

// ---- training step (conv_wgrad.hip, backward.hip) ----------------------------------------------
struct WgradArgs {          // dW[co][ci][ky][kx] (+ db[co]) of a convolution from its input and its output gradient
    const float* x0 = nullptr; int C0 = 0;     // NHWC sources [B][Hs][Ws][C] (x1: second half of a channel concat)
    const float* x1 = nullptr; int C1 = 0;
    const float* dy = nullptr;                 // NHWC [B][H][W][Cout]
    float* dw = nullptr;                       // [Cout][Cin][KS][KS]
    float* db = nullptr;                       // [Cout] or null
    float* ws = nullptr; size_t ws_floats = 0; // split-reduction workspace
    int B = 0, H = 0, W = 0, Hs = 0, Ws = 0, Cin = 0, Cout = 0, KS = 1, pad = 0, stride = 1, ups = 0;
    int split_target = 1024;                   // workgroups wanted per launch: the pixel tiles are split until nci * nco * nsplit reaches it
};
struct WgradDev {            // a launch's geometry as the device sees it (conv_wgrad.hip)
    WgradArgs a;
    int BM, TWl, THl, TB, PH, PW, P;
    int tiles_x, tiles_y, mtiles, nsplit, nci, nco;
    int o_ys, o_pix;          // LDS offsets (floats)
    size_t part_stride;       // floats per split in the workspace
    int64_t dw_off = 0, db_off = -1;   // table-driven launches: where dw / db live in the caller's flat gradient vector
    int direct1 = 0;                   // 1x1, stride 1: operands straight from global memory in MFMA layout (no LDS staging)
};
int conv_wgrad_init();
size_t conv_wgrad_workspace(const WgradArgs& a);
int conv_wgrad_launch(const WgradArgs& a, hipStream_t s);
// The split partials of MANY weight-gradient launches summed by ONE table-driven launch at the end of the backward (each launch then
// needs a workspace of its own): 53 five-microsecond reduce launches per training step become one.
struct WredJob { const float* ws; int nsplit; int nb; size_t stride, nw; int64_t dw, db; int cin, kk; };   // dw / db: offsets into the flat gradient vector (db < 0: none)
int conv_wgrad_split(const WgradArgs& a, int* nsplit, size_t* part_stride);                   // geometry of the launch conv_wgrad_launch would make
int conv_wgrad_launch_noreduce(const WgradArgs& a, hipStream_t s);                           // partials into a.ws (nsplit > 1), nothing else
// Every weight gradient of one kernel size in one launch (full-batch training steps): the entries' inputs must all still be alive
int conv_wgrad_table_entry(const WgradArgs& a, int64_t dw_off, int64_t db_off, WgradDev* out, int* nblocks, size_t* lds_bytes);
int conv_wgrad_table_launch(int KS, const WgradDev* jobs_dev, const int2* blocks_dev, int nblocks, size_t lds_bytes, float* grads, hipStream_t s);
int wgrad_reduce_table_launch(const WredJob* jobs_dev, const int2* blocks_dev, int nblocks, float* grads, hipStream_t s);

struct GnBwdArgs {          // backward of y = act((gamma xhat + beta)(sc+1) + sh); xf describes the forward (mode 1: no act, 2: SiLU)
    const float* dy = nullptr;  // NHWC [B][HW][C]
    const float* h = nullptr;   // raw normalised tensor (the convolution output)
    SrcXform xf;
    float* s12 = nullptr;       // [B][C][2]: per-channel sums of du and du*xhat (kept for norm_param_grads_launch)
    float* s12p = nullptr;      // [B][gn_bwd_chunks(HW)][C][2]: the same per 64-pixel chunk (scratch between the two passes)
    float* dh = nullptr;        // NHWC [B][HW][C]
    int accumulate = 0;         // dh += instead of dh =
    const float* plus = nullptr;// optional addend of dh's shape (the residual branch's gradient: one pass instead of this + an axpy)
    int B = 0, HW = 0, C = 0;
};
int gn_bwd_chunks(int HW);
int gn_bwd_launch(const GnBwdArgs& a, hipStream_t s);
int norm_param_grads_launch(const float* s12, const float* gamma, const float* beta, const float* ss, int ss_stride, float* dgamma,
                            float* dbeta, float* dss, int B, int C, hipStream_t s);
struct NormJob { const float* s12; const float* gamma; const float* beta; const float* ss; int64_t dgamma, dbeta; int ss_col, C; };
int norm_param_grads_table_launch(const NormJob* jobs_dev, int njobs, int maxC, float* grads, float* dss, int ss_stride, int B, hipStream_t s);
int linattn_bwd_launch(const float* qkv, const float* dout, const float* ctx, float* dctx, float* kst, float* rr, float* dqkv, int B, int n,
                       int heads, hipStream_t s);
int attn_small_bwd_launch(const float* qkv, const float* dout, float* dqkv, int B, int n, int heads, hipStream_t s);
// dense layers of the conditioning path; in_act: 0 none, 1 GELU(erf), 2 SiLU applied to `xpre` on the way in
int dense_fwd_launch(const float* xpre, int in_act, const float* w, const float* bias, float* y, int B, int I, int O, hipStream_t s);
int dense_bwd_w_launch(const float* dy, int ldy, const float* xpre, int in_act, float* dw, float* db, int B, int I, int O, hipStream_t s);
struct DenseWJob { int col, O; int64_t dw, db; };
int dense_bwd_w_table_launch(const DenseWJob* jobs_dev, const int2* blocks_dev, int nblocks, const float* dss, int S, const float* xpre, int in_act,
                             float* grads, int B, int I, hipStream_t s);
int dense_bwd_x_launch(const float* dy, int ldy, const float* w, int w_t, int ldw, const float* xpre, int in_act, float* dx, int accumulate,
                       int B, int I, int O, hipStream_t s);
int sin_emb_launch(const float* time, const float* freqs, float* e, int B, int dim, hipStream_t s);
int gather_rows_launch(const float* table, const int64_t* ids, float* out, int B, int D, int R, hipStream_t s);
int mask_rows_launch(const float* d, int ld, const int64_t* ids, float* out, int B, int D, int R, hipStream_t s);
int scatter_rows_launch(const float* d, const int64_t* ids, float* dtable, int B, int D, int R, hipStream_t s);
int add_into_launch(float* dst, const float* src, size_t n, hipStream_t s);
int silu_fwd_launch(const float* z, const float* add, float* y, size_t n, hipStream_t s);
int silu_bwd_launch(const float* dy, const float* z, float* dz, size_t n, hipStream_t s);
int bilinear_bwd_launch(const float* gdst, float* gsrc /* += */, int B, int C, int Hs, int Ws, int Hd, int Wd, hipStream_t s);
int sumpool2_nhwc_launch(const float* src, float* dst, int B, int H, int W, int C, int accumulate, hipStream_t s);
int depth_to_space_launch(const float* src, float* dst, int B, int H, int W, int C, int accumulate, hipStream_t s);
int flow_interp_launch(const float* src, const float* tgt, const float* t, float* x, float* v, int B, int per, hipStream_t s);
int flow_prepare_launch(const float* src, const float* tgt, const int64_t* perm, const float* u, float t_eps, float warp_s, float t_scale,
                        const int64_t* ids, int n_classes, float* t_out, float* time_out, float* x, float* v, int* flag, int B, int per, hipStream_t s);
int mse_loss_grad_launch(const float* v, const float* tgt, float* dv, float* loss, float* ws /*256*/, size_t n, hipStream_t s);
int grad_clip_coef_launch(const float* g, size_t n0, const float* g2, size_t n1, float max_norm, float* out2 /*{norm, coef}*/, float* ws /*256*/,
                          hipStream_t s);
int adam_ema_launch(float* p, const float* g, float* m, float* v, float* ema, size_t n, const float* coef_dev, float lr, float b1, float b2,
                    float eps, int step, float ema_decay, int do_adam, hipStream_t s, const int* skip_flag_dev = nullptr);

// ---- OT (ot.hip) ------------------------------------------------------------------------------
int ot_launch(const float* src, const float* tgt, int B, int64_t D, float* dist, int64_t* perm, hipStream_t s);

}  // namespace fc
