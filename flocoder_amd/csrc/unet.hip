// Native runtime of the velocity U-Net: parameter table with the reference's state_dict names, weight
// packing, a static launch plan over a fixed activation arena, and the hipGraph-captured ODE step.
//
// Topology follows Unet._forward (unet.py:289-372); every kernel launch below cites the reference lines it
// covers.  The plan is built once per (max_batch, H, W) -- buffers never move, so one integration step can be
// captured in a hipGraph and replayed (the reference's per-call host syncs are gone).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "plan.h"
#include "unet_priv.h"
#include "unet_sample.h"

namespace fc {

static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
int fail(int code, const std::string& m) { g_err = m; return code; }
const char* last_error() { return g_err.c_str(); }

}  // namespace fc

using namespace fc;

namespace fc {

// ------------------------------------------------------------------------------------------- parameters
static void declare(fc_unet* u, const std::string& name, std::initializer_list<int64_t> shape) { u->declare(name, shape); }
static int64_t pk_alloc(fc_unet* u, const std::string& name, int64_t numel) { return u->pk_alloc(name, numel); }
static void decl_conv(fc_unet* u, const std::string& n, int O, int I, int K, bool bias = true) { u->decl_conv(n, O, I, K, bias); }
static void decl_linear_t(fc_unet* u, const std::string& n, int O, int I) { u->decl_linear_t(n, O, I); }
static void decl_norm(fc_unet* u, const std::string& n, int C) { u->decl_norm(n, C); }
static void decl_resblock(fc_unet* u, const std::string& p, int cin, int cout) {
    declare(u, p + ".mlp.1.weight", {2 * cout, u->td});
    declare(u, p + ".mlp.1.bias", {2 * cout});
    u->ss_off[p] = u->S;
    u->S += 2 * cout;
    decl_conv(u, p + ".block1.proj", cout, cin, 3);
    decl_norm(u, p + ".block1.norm", cout);
    decl_conv(u, p + ".block2.proj", cout, cout, 3);
    decl_norm(u, p + ".block2.norm", cout);
    if (cin != cout) decl_conv(u, p + ".res_conv", cout, cin, 1);
}
static void decl_linattn(fc_unet* u, const std::string& p, int C) {
    const int hid = u->heads * 32;
    decl_conv(u, p + ".fn.fn.to_qkv", 3 * hid, C, 1, false);   // PreNorm registers fn before norm (unet.py:156-157)
    decl_conv(u, p + ".fn.fn.to_out.0", C, hid, 1);
    decl_norm(u, p + ".fn.fn.to_out.1", C);
    decl_norm(u, p + ".fn.norm", C);
}

static int declare_all(fc_unet* u) {
    const fc_unet_config& c = u->cfg;
    const int dim = c.dim, ch = c.channels, L = c.n_levels;
    u->td = dim * 8;  // unet.py:197
    u->chans.assign(1, dim);
    for (int i = 0; i < L; ++i) u->chans.push_back(dim * c.dim_mults[i]);
    const std::vector<int>& cs = u->chans;
    decl_conv(u, "init_conv", dim, ch, 1);
    decl_linear_t(u, "time_mlp.1", u->td, dim);
    decl_linear_t(u, "time_mlp.3", u->td, u->td);
    if (c.n_classes > 0) {
        declare(u, "class_cond_mlp.0.weight", {c.n_classes, u->td});
        decl_linear_t(u, "class_cond_mlp.1", u->td, u->td);
        decl_linear_t(u, "class_cond_mlp.3", u->td, u->td);
    }
    if (c.mask_cond) {  // unet.py:214-235
        decl_conv(u, "mask_fusion_conv.0", 2 * dim, dim + ch, 5);
        decl_conv(u, "mask_fusion_conv.2", 2 * dim, 2 * dim, 3);
        decl_conv(u, "mask_fusion_conv.4", dim, 2 * dim, 3);
        for (int i = 0; i < 2 && i < L; ++i) decl_conv(u, "down_mask_fusions." + std::to_string(i) + ".0", cs[i], cs[i] + ch, 3);
        for (int i = 0; i < 2 && i < L; ++i) decl_conv(u, "up_mask_fusions." + std::to_string(i) + ".0", cs[L - i], cs[L - i] + ch, 3);
    }
    for (int i = 0; i < L; ++i) {  // unet.py:242-258
        const std::string p = "downs." + std::to_string(i);
        decl_resblock(u, p + ".0", cs[i], cs[i]);
        decl_resblock(u, p + ".1", cs[i], cs[i]);
        decl_linattn(u, p + ".2", cs[i]);
        if (i == L - 1) decl_conv(u, p + ".3", cs[i + 1], cs[i], 3);
        else {
            declare(u, p + ".3.1.weight", {cs[i + 1], 4 * cs[i], 1, 1});
            declare(u, p + ".3.1.bias", {cs[i + 1]});
            const int64_t dst = pk_alloc(u, p + ".3.1.weight", (int64_t)cs[i + 1] * 4 * cs[i]);
            u->packops.push_back({1, u->params[u->pidx[p + ".3.1.weight"]].offset, dst, cs[i + 1], cs[i], 0, 0});
        }
    }
    for (int i = 0; i < L; ++i) {  // unet.py:265-281, (dim_in, dim_out) = reversed(in_out)[i]
        const std::string p = "ups." + std::to_string(i);
        const int din = cs[L - 1 - i], dout = cs[L - i];
        decl_resblock(u, p + ".0", dout + din, dout);
        decl_resblock(u, p + ".1", dout + din, dout);
        decl_linattn(u, p + ".2", dout);
        if (i == L - 1) decl_conv(u, p + ".3", din, dout, 3);
        else u->decl_conv_up2(p + ".3.1", din, dout);      // Upsample: also the four parity kernels of the folded form (plan.h conv_up2)
    }
    const int mid = cs[L];
    decl_resblock(u, "mid_block1", mid, mid);
    decl_conv(u, "mid_attn.fn.fn.to_qkv", 3 * u->heads * 32, mid, 1, false);
    decl_conv(u, "mid_attn.fn.fn.to_out", mid, u->heads * 32, 1);
    decl_norm(u, "mid_attn.fn.norm", mid);
    decl_resblock(u, "mid_block2", mid, mid);
    decl_resblock(u, "final_res_block", 2 * dim, dim);
    decl_conv(u, "final_conv", ch, dim, 1);
    // concatenated scale/shift projection of every ResnetBlock: wt [td][S], bias [S]
    pk_alloc(u, "__ss_wt", (int64_t)u->td * u->S);
    pk_alloc(u, "__ss_bias", u->S);
    for (auto& kv : u->ss_off) {
        const Param& w = u->params[u->pidx[kv.first + ".mlp.1.weight"]];
        const Param& b = u->params[u->pidx[kv.first + ".mlp.1.bias"]];
        u->packops.push_back({2, w.offset, u->pk["__ss_wt"], (int)w.shape[0], u->td, u->S, kv.second});
        u->packops.push_back({3, b.offset, u->pk["__ss_bias"] + kv.second, (int)b.numel, 0, 0, 0});
    }
    return FC_OK;
}

// Fused Block tails across workgroups (ConvFin): on by default since the meeting became one tagged-granule round trip (+2.5 % on the
// sampler, profiles/r02_*); FLOCODER_AMD_FUSED_TAIL=local keeps only the meeting-free form, =0 turns both off.
static int g_fused_tail = -1;
static bool fused_tail_enabled() {
    if (g_fused_tail < 0) { const char* e = std::getenv("FLOCODER_AMD_FUSED_TAIL"); g_fused_tail = (e && (std::string(e) == "0" || std::string(e) == "local")) ? 0 : 1; }
    return g_fused_tail == 1;
}

// ------------------------------------------------------------------------------------------- plan builder
struct Builder : PlanBuilder {
    fc_unet* u = nullptr;
    Builder(fc_unet* u_, Plan* pl_, int B_) : u(u_) { pl = pl_; B = B_; fin_err_word = u_->dev_err; }
    // launches that wait for other workgroups: only on a device this handle has to itself, and never with two chains on two streams
    bool meeting_ok() const { return fused_tail_enabled() && !u->shared && u->nchains < 2; }

    // ResnetBlock (unet.py:76-96): conv1 [+res_conv] | conv2 with GN+FiLM+SiLU folded into its loader | finalize.
    Act resblock(const std::string& p, const Act& x, const Act* skip, int cout, bool want_gn1, Stat* gn1) {
        scope = p;
        const int G = u->cfg.groups, cin = x.C + (skip ? skip->C : 0);
        Act h1 = act(cout, x.H, x.W), h2 = act(cout, x.H, x.W), out = act(cout, x.H, x.W), rb;
        Stat st1, st2;
        ConvArgs a;
        a.s0.p = x.p; a.s0.C = x.C;
        if (skip) { a.s1.p = skip->p; a.s1.C = skip->C; }
        a.Hs = x.H; a.Ws = x.W; a.KS = 3; a.pad = 1;
        a.w = u->P(p + ".block1.proj.weight"); a.bias = u->R(p + ".block1.proj.bias");
        a.w4 = u->P8(p + ".block1.proj.weight");
        if (cin != cout) {
            rb = act(cout, x.H, x.W);
            a.res_w = u->P(p + ".res_conv.weight"); a.res_b = u->R(p + ".res_conv.bias"); a.res_out = rb.p;
            a.res_w4 = u->P8(p + ".res_conv.weight");
            if (!a.res_w4) a.w4 = nullptr;      // both operands or neither
        }
        conv(a, h1, G, &st1);
        if (pl->join_at == 0) pl->join_at = (int)pl->ops.size();   // first reader of the scale / shift table
        ConvArgs b;
        b.s0.p = h1.p; b.s0.C = cout;
        b.s0.xf = xf_of(st1, 2, u->R(p + ".block1.norm.weight"), u->R(p + ".block1.norm.bias"), pl->ss + u->ss_off.at(p), u->S);
        b.Hs = x.H; b.Ws = x.W; b.KS = 3; b.pad = 1;
        b.w = u->P(p + ".block2.proj.weight"); b.bias = u->R(p + ".block2.proj.bias");
        b.w4 = u->P8(p + ".block2.proj.weight");
        // inference plans close the Block inside conv2 (ConvFin) when the launch keeps its whole grid resident; training plans keep
        // the raw h2 and its statistics for the backward
        // Measured on one box, B=64 (profiles/README.md r01_g): 494 samples/s with the fused tails against 503 without -- the
        // in-kernel meeting (four dependent memory round trips) costs what a launch boundary plus the finalize pass cost.  Off
        // unless FLOCODER_AMD_FUSED_TAIL=1 / fc_debug_set_fused_tail(1).
        const float* resp = (cin != cout) ? rb.p : x.p;
        // Where a tile holds whole GroupNorm groups (dim 32: the 128-channel blocks at 4x4 and 8x8, 32 channels per group) there is nothing to meet
        // for and the tail is always fused (FLOCODER_AMD_FUSED_TAIL=0 turns that off too).
        static const bool no_local = [] { const char* e = std::getenv("FLOCODER_AMD_FUSED_TAIL"); return e && std::string(e) == "0"; }();
        // training plans can fuse the tail too, keeping what the backward reads (raw h2, its statistics in the ordinary form):
        // FLOCODER_AMD_TRAIN_FUSED_TAIL=1.  Off by default -- measured on one box (r02): stl_sd step 3.515 ms fused against 3.446 ms
        // with conv + finalize, flowers-sized step 7.91 against 7.92: at these small grids the meeting costs what the launch it saves does.
        static const bool train_fused = [] { const char* e = std::getenv("FLOCODER_AMD_TRAIN_FUSED_TAIL"); return e && std::string(e) == "1"; }();
        const bool fused = (!u->keep_all || train_fused) && (meeting_ok() || !no_local) &&
                           conv_fin(b, out, G, u->R(p + ".block2.norm.weight"), u->R(p + ".block2.norm.bias"), resp, want_gn1, gn1,
                                    !meeting_ok(), u->keep_all ? &h2 : nullptr, u->keep_all ? &st2 : nullptr);
        if (!fused) {
            conv(b, h2, G, &st2);
            FinalizeArgs f;
            f.h = h2.p; f.xf = xf_of(st2, 2, u->R(p + ".block2.norm.weight"), u->R(p + ".block2.norm.bias"));
            f.res = resp; f.y = out.p; f.HW = x.H * x.W; f.C = cout;
            if (want_gn1) {
                const int bps = finalize_blocks_per_sample(f.HW, f.C);
                *gn1 = stat(1, bps, (float)(f.HW * f.C / bps));
                f.stats_out = gn1->p;
            }
            if (!err) push([f](const FwdCtx& c, hipStream_t s) { FinalizeArgs g = f; g.B = c.B; return finalize_launch(g, s); }, "finalize");
        }
        pl->named[p] = out; pl->named[p + ".h1"] = h1; pl->named[p + ".h2"] = h2;
        ResRec rec;
        rec.p = p; rec.x = x; if (skip) rec.skip = *skip; rec.h1 = h1; rec.h2 = h2; rec.rb = rb; rec.out = out; rec.st1 = st1; rec.st2 = st2; rec.cout = cout;
        pl->tape.push_back({0, (int)pl->res.size()});
        pl->res.push_back(rec);
        return out;
    }

    // Residual(PreNorm(LinearAttention)) (unet.py:125-161,250): qkv conv with GroupNorm(1) in its loader | context |
    // apply | to_out conv | GroupNorm(1) + residual.
    Act linattn(const std::string& p, const Act& x, const Stat& gn1) {
        scope = p;
        const int hid = u->heads * 32, n = x.H * x.W, heads = u->heads;
        static const bool no_fuse = [] { const char* e = std::getenv("FLOCODER_AMD_LINATTN"); return e && std::string(e) == "unfused"; }();
        // measured (tools/op_table.py, B=64): fused 110 vs 201 us at n=1024, 43 vs 65 us at n=256; at n <= 64 the per-workgroup weight
        // loads dominate and the unfused chain wins (37 vs 67 us at n=16, C=256)
        if (!u->keep_all && !no_fuse && n >= 256 && linattn_fused_supported(n, x.C, heads)) return linattn_fused(p, x, gn1);
        // n <= 64: a workgroup per (sample, head), then one per sample (linattn_sample.hip): 2 launches instead of 5
        static const bool no_sample = std::getenv("FLOCODER_AMD_LINATTN_NO_SAMPLE") != nullptr;
        if (!u->keep_all && !no_fuse && !no_sample && linattn_sample_supported(n, x.C, heads)) return linattn_sample(p, x, gn1);
        Act qkv = act(3 * hid, x.H, x.W), lao = act(hid, x.H, x.W), yb = act(x.C, x.H, x.W), out = act(x.C, x.H, x.W);
        float* ctx = dmalloc((size_t)B * heads * 32 * 32);
        ConvArgs a;
        a.s0.p = x.p; a.s0.C = x.C; a.s0.xf = xf_of(gn1, 1, u->R(p + ".fn.norm.weight"), u->R(p + ".fn.norm.bias"));
        a.Hs = x.H; a.Ws = x.W; a.KS = 1; a.pad = 0;
        a.w = u->P(p + ".fn.fn.to_qkv.weight");
        conv(a, qkv, 0, nullptr);
        const float* qp = qkv.p; float* lp = lao.p;
        if (!err) {
            push([=](const FwdCtx& c, hipStream_t s) { return linattn_ctx_launch(qp, ctx, c.B, n, heads, s); }, "linattn_ctx", 2.0 * n * 32 * 32 * heads);
            push([=](const FwdCtx& c, hipStream_t s) { return linattn_apply_launch(qp, ctx, lp, c.B, n, heads, s); }, "linattn_apply", 2.0 * n * 32 * 32 * heads);
        }
        ConvArgs o;
        o.s0.p = lao.p; o.s0.C = hid; o.Hs = x.H; o.Ws = x.W; o.KS = 1; o.pad = 0;
        o.w = u->P(p + ".fn.fn.to_out.0.weight"); o.bias = u->R(p + ".fn.fn.to_out.0.bias");
        Stat sty;
        conv(o, yb, 1, &sty);
        FinalizeArgs f;
        f.h = yb.p; f.xf = xf_of(sty, 1, u->R(p + ".fn.fn.to_out.1.weight"), u->R(p + ".fn.fn.to_out.1.bias"));
        f.res = x.p; f.y = out.p; f.HW = n; f.C = x.C;
        if (!err) push([f](const FwdCtx& c, hipStream_t s) { FinalizeArgs g = f; g.B = c.B; return finalize_launch(g, s); }, "finalize");
        pl->named[p] = out; pl->named[p + ".qkv"] = qkv; pl->named[p + ".lao"] = lao; pl->named[p + ".y"] = yb;
        LinRec rec;
        rec.p = p; rec.x = x; rec.qkv = qkv; rec.lao = lao; rec.yb = yb; rec.out = out; rec.ctx = ctx; rec.gn1 = gn1; rec.sty = sty;
        pl->tape.push_back({1, (int)pl->lin.size()});
        pl->lin.push_back(rec);
        return out;
    }

    // the same module on the fused kernels (linattn_fused.hip): x -> context -> y in two launches, q/k/v never stored
    Act linattn_fused(const std::string& p, const Act& x, const Stat& gn1) {
        const int n = x.H * x.W, heads = u->heads, hid = heads * 32;
        Act yb = act(x.C, x.H, x.W), out = act(x.C, x.H, x.W);
        Stat sty = stat(1, linattn_fused_tiles(n), linattn_fused_nt(n, x.C));
        LaArgs a;
        a.x = x.p; a.xf = xf_of(gn1, 1, u->R(p + ".fn.norm.weight"), u->R(p + ".fn.norm.bias"));
        a.wqkv = u->P(p + ".fn.fn.to_qkv.weight"); a.wout = u->P(p + ".fn.fn.to_out.0.weight"); a.bout = u->R(p + ".fn.fn.to_out.0.bias");
        a.ctx = dmalloc((size_t)B * heads * 32 * 32); a.y = yb.p; a.stats_out = sty.p; a.n = n; a.C = x.C; a.heads = heads;
        const double fl = 2.0 * n * (double)x.C * 3 * hid + 2.0 * 2 * n * 32 * 32 * heads + 2.0 * n * (double)hid * x.C;
        // The module closed by its own apply launch (to_out.1's GroupNorm(1) + the residual on the tile still in registers: no y round trip,
        // no finalize launch) where the workgroups of a sample may wait for each other: the exclusive plan, the whole grid resident.
        if (!err && meeting_ok() && linattn_fused_meeting_ok(B, n, x.C)) {
            const int T = linattn_fused_tiles(n);
            a.g2 = u->R(p + ".fn.fn.to_out.1.weight"); a.b2 = u->R(p + ".fn.fn.to_out.1.bias"); a.out = out.p;
            a.gran = reinterpret_cast<unsigned long long*>(dmalloc((size_t)B * T * 2 * 2));
            a.sync = reinterpret_cast<unsigned*>(dmalloc((size_t)B));
            if (err) return out;
            if (hipMemset(a.gran, 0, (size_t)B * T * 2 * sizeof(unsigned long long)) != hipSuccess || hipMemset(a.sync, 0, (size_t)B * sizeof(unsigned)) != hipSuccess) {
                err = fail(FC_E_HIP, "hipMemset failed");
                return out;
            }
            a.err = fin_err_word;
            ++pl->n_meet; pl->fin_sync.push_back(a.sync); pl->fin_kind.push_back(1);
            push([a](const FwdCtx& c, hipStream_t s) { LaArgs b = a; b.B = c.B; return linattn_fused_launch(b, s); }, "linattn_fused+fin", fl);
            pl->named[p] = out;
            return out;
        }
        if (!err) push([a](const FwdCtx& c, hipStream_t s) { LaArgs b = a; b.B = c.B; return linattn_fused_launch(b, s); }, "linattn_fused", fl);
        FinalizeArgs f;
        f.h = yb.p; f.xf = xf_of(sty, 1, u->R(p + ".fn.fn.to_out.1.weight"), u->R(p + ".fn.fn.to_out.1.bias"));
        f.res = x.p; f.y = out.p; f.HW = n; f.C = x.C;
        if (!err) push([f](const FwdCtx& c, hipStream_t s) { FinalizeArgs g = f; g.B = c.B; return finalize_launch(g, s); }, "finalize");
        pl->named[p] = out; pl->named[p + ".y"] = yb;
        return out;
    }

    // arrival counters of the one-launch form of linattn_sample.hip (zeroed here, once: every launch adds `heads` per sample), or null.
    // Default: null -- the closing step stays a launch of its own (la_join).  FLOCODER_AMD_LA_JOIN=one selects the one-launch form: built and
    // tested in round 3, but it measures the same or slightly below the two-launch form (789 against 793 samples/s, three alternating pairs;
    // 1286 against 1279 us of kernel time per forward), so it is not the default.
    unsigned* la_tickets(int n, int C) {
        static const bool one = [] { const char* e = std::getenv("FLOCODER_AMD_LA_JOIN"); return e && std::string(e) == "one"; }();
        if (!one || !linattn_sample_one_launch(n, C) || err) return nullptr;
        unsigned* t = reinterpret_cast<unsigned*>(dmalloc((size_t)B));
        if (t && hipMemset(t, 0, (size_t)B * sizeof(unsigned)) != hipSuccess) { err = fail(FC_E_HIP, "hipMemset failed on the attention tickets"); return nullptr; }
        return t;
    }

    // the whole module in one launch, or two (linattn_sample.hip)
    Act linattn_sample(const std::string& p, const Act& x, const Stat& gn1) {
        const int n = x.H * x.W, heads = u->heads, hid = heads * 32;
        Act out = act(x.C, x.H, x.W);
        LaArgs a;
        a.x = x.p; a.xf = xf_of(gn1, 1, u->R(p + ".fn.norm.weight"), u->R(p + ".fn.norm.bias"));
        a.wqkv = u->P(p + ".fn.fn.to_qkv.weight"); a.wout = u->P(p + ".fn.fn.to_out.0.weight"); a.bout = u->R(p + ".fn.fn.to_out.0.bias");
        a.wqkv4 = u->P8(p + ".fn.fn.to_qkv.weight");
        a.g2 = u->R(p + ".fn.fn.to_out.1.weight"); a.b2 = u->R(p + ".fn.fn.to_out.1.bias"); a.out = out.p;
        a.n = n; a.C = x.C; a.heads = heads;
        a.part = dmalloc((size_t)B * heads * n * x.C);
        a.tickets = la_tickets(n, x.C);
        const double fl = 2.0 * n * (double)x.C * 3 * hid + 2.0 * 2 * n * 32 * 32 * heads + 2.0 * n * (double)hid * x.C;
        if (!err) push([a](const FwdCtx& c, hipStream_t s) { LaArgs b = a; b.B = c.B; return linattn_sample_launch(b, s); }, "linattn_sample", fl);
        pl->named[p] = out;
        return out;
    }

    // Residual(PreNorm(Attention)) (unet.py:99-122,262)
    Act midattn(const Act& x, const Stat& gn1) {
        scope = "mid_attn";
        const int hid = u->heads * 32, n = x.H * x.W, heads = u->heads;
        static const bool no_sample = std::getenv("FLOCODER_AMD_LINATTN_NO_SAMPLE") != nullptr;
        if (!u->keep_all && !no_sample && attn_sample_supported(n, x.C, heads)) {    // two launches instead of three (linattn_sample.hip)
            Act out = act(x.C, x.H, x.W);
            LaArgs a;
            a.x = x.p; a.xf = xf_of(gn1, 1, u->R("mid_attn.fn.norm.weight"), u->R("mid_attn.fn.norm.bias"));
            a.wqkv = u->P("mid_attn.fn.fn.to_qkv.weight"); a.wout = u->P("mid_attn.fn.fn.to_out.weight"); a.bout = u->R("mid_attn.fn.fn.to_out.bias");
            a.wqkv4 = u->P8("mid_attn.fn.fn.to_qkv.weight");
            a.out = out.p; a.n = n; a.C = x.C; a.heads = heads;
            a.part = dmalloc((size_t)B * heads * n * x.C);
            a.tickets = la_tickets(n, x.C);
            const double fl = 2.0 * n * (double)x.C * 3 * hid + 2.0 * 2.0 * n * n * 32 * heads + 2.0 * n * (double)hid * x.C;
            if (!err) push([a](const FwdCtx& c, hipStream_t s) { LaArgs b = a; b.B = c.B; return attn_sample_launch(b, s); }, "attn_sample", fl);
            pl->named["mid_attn"] = out;
            return out;
        }
        Act qkv = act(3 * hid, x.H, x.W), ao = act(hid, x.H, x.W), out = act(x.C, x.H, x.W);
        ConvArgs a;
        a.s0.p = x.p; a.s0.C = x.C; a.s0.xf = xf_of(gn1, 1, u->R("mid_attn.fn.norm.weight"), u->R("mid_attn.fn.norm.bias"));
        a.Hs = x.H; a.Ws = x.W; a.KS = 1;
        a.w = u->P("mid_attn.fn.fn.to_qkv.weight");
        conv(a, qkv, 0, nullptr);
        const float* qp = qkv.p; float* ap = ao.p;
        if (!err) {
            push([=](const FwdCtx& c, hipStream_t s) { return attn_small_launch(qp, ap, c.B, n, heads, s); }, "attn_small", 2.0 * 2.0 * n * n * 32 * heads);
        }
        ConvArgs o;
        o.s0.p = ao.p; o.s0.C = hid; o.Hs = x.H; o.Ws = x.W; o.KS = 1;
        o.w = u->P("mid_attn.fn.fn.to_out.weight"); o.bias = u->R("mid_attn.fn.fn.to_out.bias");
        o.add = x.p;
        conv(o, out, 0, nullptr);
        pl->named["mid_attn"] = out;
        MidRec rec;
        rec.x = x; rec.qkv = qkv; rec.ao = ao; rec.out = out; rec.gn1 = gn1;
        pl->tape.push_back({2, (int)pl->mid.size()});
        pl->mid.push_back(rec);
        return out;
    }

    // x + SiLU(conv3x3(cat[x, bilinear(mask)]))  (unet.py:336-340,360-364); a plain copy when no mask is given
    Act mask_inject(const std::string& name, const Act& x, const Act& mask_nhwc) {
        scope = name;
        Act mr = act(mask_nhwc.C, x.H, x.W), out = act(x.C, x.H, x.W);
        const float* mp = mask_nhwc.p; float* rp = mr.p;
        const int C = mask_nhwc.C, Hs = mask_nhwc.H, Ws = mask_nhwc.W, Hd = x.H, Wd = x.W;
        const bool keep = u->keep_all;             // training: the pre-activation z stays (SiLU' needs it), SiLU + residual as a pass of its own
        Act z = keep ? act(x.C, x.H, x.W) : Act();
        ConvArgs a;
        a.s0.p = x.p; a.s0.C = x.C; a.s1.p = mr.p; a.s1.C = C;
        a.Hs = x.H; a.Ws = x.W; a.KS = 3; a.pad = 1;
        if (!keep) { a.out_act = 1; a.add = x.p; }
        a.w = u->P(name + ".weight"); a.bias = u->R(name + ".bias");
        a.B = B; a.H = x.H; a.W = x.W; a.Cout = x.C; a.out = keep ? z.p : out.p; a.Cin = x.C + C;
        ConvGeom g;
        if ((err = conv_plan(a, TILE_AUTO, &g)) != FC_OK) return out;
        const int tile = g.tile;
        const size_t per = (size_t)x.H * x.W * x.C;
        const float* xp = x.p; float* op = out.p; const float* zp = z.p;
        push([=](const FwdCtx& c, hipStream_t s) -> int {
            if (!c.mask) { FC_HIP(hipMemcpyAsync(op, xp, per * sizeof(float) * c.B, hipMemcpyDeviceToDevice, s)); return FC_OK; }
            FC_TRY(bilinear_nhwc_launch(mp, rp, c.B, C, Hs, Ws, Hd, Wd, s));
            ConvArgs b = a; b.B = c.B;
            FC_TRY(conv_launch(b, tile, s));
            return keep ? silu_fwd_launch(zp, xp, op, per * c.B, s) : FC_OK;
        }, std::string("bilinear+") + kTileNames[tile], 2.0 * x.H * x.W * 9 * (double)a.Cin * a.Cout);
        InjRec rec;
        rec.name = name.substr(0, name.size() - 0); rec.x = x; rec.mr = mr; rec.z = z; rec.out = out;
        pl->tape.push_back({4, (int)pl->inj.size()});
        pl->inj.push_back(rec);
        return out;
    }
};

static void free_plan(fc_unet* u) {
    for (auto& kv : u->graphs) (void)hipGraphExecDestroy(kv.second);
    u->graphs.clear();
    for (Plan& pln : u->plan) pln.release();
    u->bwd.release();                       // the backward plan points into the forward arena
    u->dgrad_packs.clear();
    u->dgrad_table.release();
    for (void* p : u->int_allocs) dev_free(p);
    u->int_allocs.clear();
    u->maxB = 0;
    // a rebuilt plan starts clean (callers of free_plan have synchronised the device)
    if (u->dev_err) (void)hipMemset(u->dev_err, 0, sizeof(int));
    if (u->host_err) *u->host_err = 0;
    u->tail_failed = false;
}

// ---- one workgroup per sample (unet_sample.hip): the forward as a program over LDS-resident activations -------------------------------
// Used for inference plans of models whose whole per-sample state fits a CU's LDS (the dim-8 inpainting flow at 4x8x8: BASELINE config 5).
// Returns FC_OK and leaves ONE launch in the plan, or 1 when the model does not qualify (the caller then builds the ordinary plan).
static int build_sample_plan(fc_unet* u, Plan* pl, Builder& b, int maxB, int H, int W) {
    // ON for the models that qualify unless FLOCODER_AMD_SAMPLE_KERNEL=0 (round 4, config 5's dim-8 / 4x8x8 flow: 538 us per evaluation
    // against 842 us for the 115 launches of the ordinary plan; profiles/r04_sample_kernel_stamps.txt).  A sample
    // is ONE workgroup, so the kernel's time is that of one sample as long as every sample has a CU of its own: beyond that the ordinary
    // plan (whose launches grow wider, not longer) wins again -- batches larger than the CU count keep it.
    static const bool off = [] { const char* e = std::getenv("FLOCODER_AMD_SAMPLE_KERNEL"); return e && std::string(e) == "0"; }();
    static const int cus = [] { int dev = 0, n = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0; return n; }();
    if (maxB > cus) return 1;
    const fc_unet_config& c = u->cfg;
    const int L = c.n_levels, dim = c.dim, ch = c.channels, G = c.groups, heads = u->heads;
    if (off || u->keep_all || G > 8 || heads != 4 || u->nchains > 1) return 1;
    const std::vector<int>& cs = u->chans;
    struct T { int off = -1, C = 0, H = 0, W = 0; };
    std::vector<SStep> prog;
    int top = 0, peak = 0;
    double flops = 0.0;
    auto alloc = [&](int floats) { const int o = top; top += (floats + 3) & ~3; if (top > peak) peak = top; return o; };
    // block outputs whose last reader has been emitted serve later tensors of the same size (every tensor a step reads or writes is alive
    // for the whole step, so a buffer may be handed out again only AFTER the step that last read it has been pushed)
    std::multimap<int, int> pool;
    auto tensor = [&](int C, int h, int w) {
        T t; t.C = C; t.H = h; t.W = w;
        auto it = pool.find(C * h * w);
        if (it != pool.end()) { t.off = it->second; pool.erase(it); }
        else t.off = alloc(C * h * w);
        return t;
    };
    auto release = [&](const T& t) { if (t.off >= 0) pool.emplace(t.C * t.H * t.W, t.off); };
    struct NormSpec { std::string name; int groups = 0, ss_off = -1; };       // a GroupNorm folded into the convolution's epilogue (groups > 0)
    auto conv = [&](const T& x, const T* x1, const T& out, const std::string& name, int KS, int pad, int stride, int ups, int act, int res, int guard, bool bias = true,
                    const NormSpec* nm = nullptr) {
        SStep s;
        if (nm && nm->groups > 0) {
            s.fnorm = 1; s.G = nm->groups; s.ss_off = nm->ss_off; s.lcpg = ilog2(out.C / nm->groups);
            s.gamma = u->R(nm->name + ".weight"); s.beta = u->R(nm->name + ".bias");
        }
        s.op = S_CONV; s.guard = guard; s.in0 = x.off; s.C0 = x.C; s.in1 = x1 ? x1->off : -1; s.C1 = x1 ? x1->C : 0;
        s.out = out.off; s.Cout = out.C; s.Hi = x.H; s.Wi = x.W; s.Ho = out.H; s.Wo = out.W; s.KS = KS; s.pad = pad; s.stride = stride; s.ups = ups;
        s.act = act; s.res = res; s.w = u->P(name + ".weight"); s.bias = bias ? u->R(name + ".bias") : nullptr;
        s.lco = ilog2(out.C);
        {   // the weight rows some output pixel can reach, merged where contiguous, cut into chunks of <= 4096 floats
            const int Cin = s.C0 + s.C1, Hin = x.H << ups, Win = x.W << ups, RW0 = 4096 / out.C;
            int kx_lo = KS, kx_hi = -1;
            for (int kx = 0; kx < KS; ++kx) if ((out.W - 1) * stride - pad + kx >= 0 && -pad + kx < Win) { if (kx < kx_lo) kx_lo = kx; kx_hi = kx; }
            std::vector<std::pair<int, int>> ranges;
            for (int ky = 0; ky < KS && kx_hi >= kx_lo; ++ky) {
                if (!((out.H - 1) * stride - pad + ky >= 0 && -pad + ky < Hin)) continue;
                const int r0 = (ky * KS + kx_lo) * Cin, r1 = (ky * KS + kx_hi + 1) * Cin;
                if (!ranges.empty() && ranges.back().second == r0) ranges.back().second = r1;
                else ranges.push_back({r0, r1});
            }
            // (whole taps per chunk where a tap's rows fit one; whole channel quads otherwise)
            const int RW = RW0 >= Cin ? (RW0 / Cin) * Cin : (RW0 & ~3);
            int nck = 0;
            for (auto& rg : ranges)
                for (int r = rg.first; r < rg.second; r += RW) {
                    if (nck < 8) { s.crow[nck] = r; s.cn[nck] = rg.second - r < RW ? rg.second - r : RW; }
                    ++nck;
                }
            s.nchunk = nck;                              // > 8: the model does not qualify (checked below)
            // lanes per output quad: as many as fill the workgroup, at most the channel quads of a tap
            const int quads = out.H * out.W * out.C / 4, CQ = Cin / 4;
            while (s.lks < 6 && (quads << (s.lks + 1)) <= SAMPLE_THREADS && (2 << s.lks) <= CQ) ++s.lks;
            // the unrolled multiply-add: ONE chunk of one tap or of all nine taps of a 3x3 kernel, the lanes of a quad dividing a tap's channel
            // quads evenly, at most four each
            const int ntaps = nck == 1 ? s.cn[0] / Cin : 0;
            if (nck == 1 && s.cn[0] == ntaps * Cin && (ntaps == 1 || (ntaps == 9 && KS == 3 && s.crow[0] == 0))) {
                int lf = 0;
                while (lf < 6 && (quads << (lf + 1)) <= SAMPLE_THREADS && CQ % (2 << lf) == 0) ++lf;
                if (CQ >> lf <= 4) { s.lks = lf; s.nqi = CQ >> lf; s.fast = ntaps; }
            }
        }
        prog.push_back(s);
        return 2.0 * out.H * out.W * KS * KS * (double)(s.C0 + s.C1) * out.C;
    };
    auto copy = [&](const T& x, const T& out, int guard) {
        SStep s;
        s.op = S_COPY; s.guard = guard; s.in0 = x.off; s.out = out.off; s.Cout = x.C * x.H * x.W;
        prog.push_back(s);
    };
    auto resblock = [&](const std::string& p, const T& x, const T* skip, int cout) {
        T out = tensor(cout, x.H, x.W);
        const int mark = top, cin = x.C + (skip ? skip->C : 0);
        // Block = convolution with GroupNorm (+ FiLM) + SiLU in its epilogue (unet.py:57-73); block2 adds the residual there too
        T a1; a1.C = cout; a1.H = x.H; a1.W = x.W; a1.off = alloc(cout * x.H * x.W);      // (temporaries: stack-allocated above `mark`, never pooled)
        NormSpec n1{p + ".block1.norm", G, u->ss_off.at(p)}, n2{p + ".block2.norm", G, -1};
        flops += conv(x, skip, a1, p + ".block1.proj", 3, 1, 1, 0, 1, -1, 0, true, &n1);
        int res = x.off;
        if (cin != cout) {
            T rb; rb.C = cout; rb.H = x.H; rb.W = x.W; rb.off = alloc(cout * x.H * x.W);
            flops += conv(x, skip, rb, p + ".res_conv", 1, 0, 1, 0, 0, -1, 0);
            res = rb.off;
        }
        flops += conv(a1, nullptr, out, p + ".block2.proj", 3, 1, 1, 0, 1, res, 0, true, &n2);
        top = mark;
        return out;
    };
    auto attention = [&](const std::string& p, const T& x, bool full) {
        T out = tensor(x.C, x.H, x.W);
        const int mark = top, n = x.H * x.W, hid = heads * 32;
        SStep s;
        s.op = full ? S_ATTN : S_LINATTN; s.in0 = x.off; s.out = out.off; s.C0 = x.C; s.Hi = x.H; s.Wi = x.W;
        if (n == 1) { s.op = S_ATTN1; s.full = full ? 1 : 0; }       // one position: the closed form (unet_sample.hip op_attention1)
        else if (!full && n <= 64 && (x.C == 8 || x.C == 16)) s.op = S_LINATTN_W;
        else if (!full && n <= 16 && x.C == 32) s.op = S_LINATTN_G;      // the same with its weights read from L2 (they do not fit the staging buffer)   // a wave per head (all four heads' weights fit the staging buffer: C * 512 floats)
        s.scratch = alloc(n == 1 ? 2 * x.C + 128 + SAMPLE_THREADS
                          : (s.op == S_LINATTN_W || s.op == S_LINATTN_G) ? 5 * n * x.C + 4 * (n + std::max(n, 32)) * 32
                          : 2 * n * x.C + 3 * n * 32 + (full ? n * n : 32 * 32) + n * 32);
        s.gamma = u->R(p + ".fn.norm.weight"); s.beta = u->R(p + ".fn.norm.bias");
        s.lc = ilog2(x.C); s.ln = ilog2(n);
        s.w = u->P(p + ".fn.fn.to_qkv.weight");
        if (full) { s.w2 = u->P(p + ".fn.fn.to_out.weight"); s.b2 = u->R(p + ".fn.fn.to_out.bias"); }
        else {
            s.w2 = u->P(p + ".fn.fn.to_out.0.weight"); s.b2 = u->R(p + ".fn.fn.to_out.0.bias");
            s.g2 = u->R(p + ".fn.fn.to_out.1.weight"); s.be2 = u->R(p + ".fn.fn.to_out.1.bias");
        }
        prog.push_back(s);
        flops += 2.0 * n * (double)x.C * 3 * hid + (full ? 2.0 * 2.0 * n * n * 32 * heads : 2.0 * 2 * n * 32 * 32 * heads) + 2.0 * n * (double)hid * x.C;
        top = mark;
        return out;
    };
    T xin = tensor(ch, H, W), mask;
    if (c.mask_cond) mask = tensor(ch, H, W);
    T xi = tensor(dim, H, W), x0 = xi;
    flops += conv(xin, nullptr, xi, "init_conv", 1, 0, 1, 0, 0, -1, 0);
    if (c.mask_cond) {                                   // unet.py:298-305: replaces x when a mask is given that is not all ones
        x0 = tensor(dim, H, W);
        const int mark = top;
        T f1, f2;
        f1.C = f2.C = 2 * dim; f1.H = f2.H = H; f1.W = f2.W = W; f1.off = alloc(2 * dim * H * W); f2.off = alloc(2 * dim * H * W);
        conv(xi, &mask, f1, "mask_fusion_conv.0", 5, 2, 1, 0, 1, -1, 2);
        conv(f1, nullptr, f2, "mask_fusion_conv.2", 3, 1, 1, 0, 1, -1, 2);
        conv(f2, nullptr, x0, "mask_fusion_conv.4", 3, 1, 1, 0, 0, -1, 2);
        copy(xi, x0, 5);
        flops += 2.0 * H * W * (25.0 * (dim + ch) * 2 * dim + 9.0 * 2 * dim * 2 * dim + 9.0 * 2 * dim * dim);
        top = mark;
        release(xi);
    }
    auto inject = [&](const std::string& name, const T& x) {       // x + SiLU(conv3x3(cat[x, bilinear(mask)])), unet.py:336-340,360-364
        T out = tensor(x.C, x.H, x.W);
        const int mark = top;
        T mr; mr.C = ch; mr.H = x.H; mr.W = x.W; mr.off = alloc(ch * x.H * x.W);
        SStep s;
        s.op = S_BILINEAR; s.guard = 1; s.in0 = mask.off; s.out = mr.off; s.C0 = ch; s.Hi = H; s.Wi = W; s.Ho = x.H; s.Wo = x.W;
        prog.push_back(s);
        flops += conv(x, &mr, out, name, 3, 1, 1, 0, 1, x.off, 1);
        copy(x, out, 3);
        top = mark;
        return out;
    };
    std::vector<T> skips;
    T x = x0;
    // `step(y = f(x))`: the new tensor is taken BEFORE the old one is released (they are alive together in the step), the old one after
    auto next = [&](T& cur, const T& nw, bool keep_old) { if (!keep_old) release(cur); cur = nw; };
    for (int i = 0; i < L; ++i) {
        const std::string p = "downs." + std::to_string(i);
        next(x, resblock(p + ".0", x, nullptr, cs[i]), i == 0);                 // (level 0's input is x0: final_res_block reads it)
        skips.push_back(x);
        next(x, resblock(p + ".1", x, nullptr, cs[i]), true);                   // the input is a skip
        next(x, attention(p + ".2", x, false), false);
        skips.push_back(x);
        if (c.mask_cond && i < 2) next(x, inject("down_mask_fusions." + std::to_string(i) + ".0", x), true);     // the input is a skip
        const bool x_is_skip = !(c.mask_cond && i < 2);
        if (i == L - 1) { T o = tensor(cs[i + 1], x.H, x.W); flops += conv(x, nullptr, o, p + ".3", 3, 1, 1, 0, 0, -1, 0); next(x, o, x_is_skip); }
        else { T o = tensor(cs[i + 1], x.H / 2, x.W / 2); flops += conv(x, nullptr, o, p + ".3.1", 2, 0, 2, 0, 0, -1, 0); next(x, o, x_is_skip); }
    }
    next(x, resblock("mid_block1", x, nullptr, cs[L]), false);
    next(x, attention("mid_attn", x, true), false);
    next(x, resblock("mid_block2", x, nullptr, cs[L]), false);
    for (int i = 0; i < L; ++i) {
        const std::string p = "ups." + std::to_string(i);
        const int din = cs[L - 1 - i], dout = cs[L - i];
        T s1 = skips.back(); skips.pop_back();
        next(x, resblock(p + ".0", x, &s1, dout), false);
        release(s1);
        T s2 = skips.back(); skips.pop_back();
        next(x, resblock(p + ".1", x, &s2, dout), false);
        release(s2);
        next(x, attention(p + ".2", x, false), false);
        if (c.mask_cond && i < 2) next(x, inject("up_mask_fusions." + std::to_string(i) + ".0", x), false);
        if (i == L - 1) { T o = tensor(din, x.H, x.W); flops += conv(x, nullptr, o, p + ".3", 3, 1, 1, 0, 0, -1, 0); next(x, o, false); }
        else { T o = tensor(din, x.H * 2, x.W * 2); flops += conv(x, nullptr, o, p + ".3.1", 3, 1, 1, 1, 0, -1, 0); next(x, o, false); }
    }
    next(x, resblock("final_res_block", x, &x0, dim), false);
    T v = tensor(ch, H, W);
    flops += conv(x, nullptr, v, "final_conv", 1, 0, 1, 0, 0, -1, 0);
    // every tensor at most four elements per thread, attention at most 64 channels / 64 keys (unet_sample.hip's register and staging budgets)
    // (powers of two throughout: the kernel indexes by shifts and masks)
    const int cap = 4 * SAMPLE_THREADS;
    for (const SStep& s : prog) {
        if (s.op == S_CONV && (s.Ho * s.Wo * s.Cout > cap || !is_pow2(s.Cout) || s.Cout < 4 || s.Cout > SAMPLE_THREADS || s.nchunk > 8 || (s.C0 & 3) || (s.C1 & 3) || s.C0 + s.C1 > 128)) return 1;     // (channel quads; the zero line is 128 floats)
        if (s.op == S_NORM && (s.Hi * s.Wi * s.C0 > cap || !is_pow2(s.C0) || s.C0 > 64 || !is_pow2(s.C0 / s.G) || s.G > 8)) return 1;
        if (s.op == S_CONV && s.fnorm && (s.Cout > 64 || !is_pow2(s.Cout / s.G) || s.G > 8)) return 1;
        if (s.op == S_LINATTN_W && ((s.C0 != 8 && s.C0 != 16) || !is_pow2(s.Hi * s.Wi) || s.Hi * s.Wi > 64)) return 1;
        if (s.op == S_LINATTN_G && (s.C0 != 32 || !is_pow2(s.Hi * s.Wi) || s.Hi * s.Wi > 16)) return 1;
        if (s.op == S_ATTN1 && (s.C0 > 64 || s.C0 < 8 || !is_pow2(s.C0))) return 1;
        if ((s.op == S_ATTN || s.op == S_LINATTN) && (s.C0 > 64 || s.C0 < 4 || !is_pow2(s.C0) || !is_pow2(s.Hi * s.Wi) || s.Hi * s.Wi * s.C0 > cap)) return 1;
        if (s.op == S_ATTN && s.Hi * s.Wi > 64) return 1;
    }
    top = peak;                                           // behind every temporary (they were released, not forgotten)
    const int zero_off = alloc(128);
    const int wbuf_off = alloc(2 * 4096), prog_off = alloc((int)(prog.size() * sizeof(SStep) / 4) + 4);
    const size_t lds = (size_t)peak * sizeof(float);
    if (std::getenv("FLOCODER_AMD_SAMPLE_KERNEL_DEBUG")) fprintf(stderr, "[unet_sample] %zu steps, %zu bytes of LDS per sample\n", prog.size(), lds);
    if (lds > 158 * 1024) return 1;                       // the sample does not fit a CU: ordinary plan
    FC_TRY(unet_sample_init());
    SStep* dev = reinterpret_cast<SStep*>(b.dmalloc((prog.size() * sizeof(SStep) + 3) / 4 + 4));
    unsigned* done = reinterpret_cast<unsigned*>(b.dmalloc(4));
    if (b.err) return b.err;
    FC_HIP(hipMemcpy(dev, prog.data(), prog.size() * sizeof(SStep), hipMemcpyHostToDevice));
    FC_HIP(hipMemset(done, 0, 4 * sizeof(unsigned)));
    SampleArgs a;
    a.prog = dev; a.nsteps = (int)prog.size(); a.S = u->S; a.ss = pl->ss; a.ch = ch; a.HW = H * W;
    a.x_off = xin.off; a.mask_off = mask.off; a.v_off = v.off; a.done = done; a.wbuf_off = wbuf_off; a.prog_off = prog_off; a.zero_off = zero_off;
    const bool mask_cond = c.mask_cond != 0;
    b.scope = "unet (one workgroup per sample)";
    b.push([a, lds, mask_cond](const FwdCtx& cx, hipStream_t s) {
        SampleArgs q = a;
        q.x = cx.x; q.x_mod = cx.x_mod; q.mask = mask_cond ? cx.mask : nullptr; q.mask_fuse = cx.mask_fuse;
        q.ss_all = cx.fetch.all; q.evalc = cx.fetch.all ? cx.fetch.evalc : nullptr; q.rows = cx.B; q.out = cx.out; q.euler = cx.euler;
        q.stamps = 5 * q.nsteps + 16 <= 4096 ? conv_stamp_buffer() : nullptr;   // (2 words per step + the phase split of the convolutions; fc_debug_set_conv_stamps: >= 4096 words)
        return unet_sample_launch(q, cx.B, lds, s);
    }, "unet_sample", flops);
    return FC_OK;
}

static int build_plan(fc_unet* u, Plan* pl, int maxB, int H, int W) {
    const fc_unet_config& c = u->cfg;
    const int L = c.n_levels, dim = c.dim, ch = c.channels, HW = H * W;
    if (!is_pow2(H) || !is_pow2(W) || (H >> (L - 1)) < 1 || (W >> (L - 1)) < 1)
        return fail(FC_E_SHAPE, "unet: latent height/width must be powers of two >= 2^(levels-1)");
    if ((ch & 3) || (dim & 3)) return fail(FC_E_SHAPE, "unet: channels and dim must be multiples of 4");
    Builder b(u, pl, maxB);
    pl->flops = 0.0;
    pl->t_emb = b.dmalloc((size_t)maxB * u->td);
    pl->ss = b.dmalloc((size_t)maxB * u->S);
    const std::vector<int>& cs = u->chans;
    const int td = u->td, S = u->S, ncls = c.n_classes;

    // -- conditioning (unet.py:310-316 and every ResnetBlock.mlp) --
    {
        TembArgs t;
        t.freqs = u->freqs;
        t.w1t = u->P("time_mlp.1.weight"); t.b1 = u->R("time_mlp.1.bias");
        t.w2t = u->P("time_mlp.3.weight"); t.b2 = u->R("time_mlp.3.bias");
        t.emb = t.cw1t = t.cb1 = t.cw2t = t.cb2 = nullptr;
        if (ncls > 0) {
            t.emb = u->R("class_cond_mlp.0.weight");
            t.cw1t = u->P("class_cond_mlp.1.weight"); t.cb1 = u->R("class_cond_mlp.1.bias");
            t.cw2t = u->P("class_cond_mlp.3.weight"); t.cb2 = u->R("class_cond_mlp.3.bias");
        }
        t.n_classes = ncls; t.t_out = pl->t_emb; t.dim = dim; t.td = td;
        u->temb_proto = t;
        b.scope = "time_mlp";
        float *hid = b.dmalloc((size_t)maxB * td), *chid = b.dmalloc((size_t)maxB * td);
        b.push([t, hid, chid](const FwdCtx& cx, hipStream_t s) {
            if (cx.fetch.all) return (int)FC_OK;   // this evaluation's rows were computed before the first step (fc_unet_integrate)
            TembArgs a = t; a.B = cx.B; a.time = cx.time; a.class_ids = cx.ids; a.class_batch_mod = cx.ids_mod; a.null_from = cx.null_from;
            return temb_launch(a, hid, chid, s);
        }, "temb", 2.0 * ((double)dim * td + (double)td * td * (ncls > 0 ? 3 : 1)));
        const float *te = pl->t_emb, *wt = u->P("__ss_wt"), *sb = u->P("__ss_bias");
        float* ss = pl->ss;
        b.scope = "resblock.mlp";
        b.push([=](const FwdCtx& cx, hipStream_t s) { return cx.fetch.all ? (int)FC_OK : ss_launch(te, wt, sb, ss, cx.B, td, S, s); }, "ss", 2.0 * (double)td * S);
        pl->side_ops = (int)pl->ops.size();   // the conditioning chain reads only time / class ids: it runs beside init_conv and the first conv1
    }

    {   // small models: the whole forward of a sample in one workgroup (unet_sample.hip); 1 = does not qualify
        const int r = build_sample_plan(u, pl, b, maxB, H, W);
        if (r != 1) {
            if (r != FC_OK) return r;
            pl->join_at = (int)pl->ops.size();
            pl->maxB = maxB; pl->H = H; pl->W = W;
            return FC_OK;
        }
    }
    // -- init_conv (unet.py:295) and mask fusion (unet.py:298-305) --
    Act x0 = b.act(dim, H, W);
    pl->named["init"] = x0;
    pl->x0 = x0;
    Act mask_nhwc;
    {
        const float *w = u->P("init_conv.weight"), *bias = u->R("init_conv.bias");
        float* x0p = x0.p;
        b.scope = "init_conv";
        if (!c.mask_cond) {
            b.push([=](const FwdCtx& cx, hipStream_t s) { return init_conv_launch(cx.fetch, cx.x, cx.x_mod, w, bias, x0p, cx.B, ch, HW, dim, s); }, "init_conv", 2.0 * HW * ch * dim);
        } else {
            Act xi = b.act(dim, H, W), f1 = b.act(2 * dim, H, W), f2 = b.act(2 * dim, H, W);
            const bool keep = u->keep_all;         // training keeps the pre-activations z1, z2 of the two SiLU layers
            Act z1 = keep ? b.act(2 * dim, H, W) : Act(), z2 = keep ? b.act(2 * dim, H, W) : Act();
            mask_nhwc = b.act(ch, H, W);
            float *xip = xi.p, *mp = mask_nhwc.p;
            b.push([=](const FwdCtx& cx, hipStream_t s) -> int {
                if (cx.mask) FC_TRY(nchw_to_nhwc_launch(cx.mask, mp, cx.B, ch, HW, ch, cx.x_mod, s));
                return init_conv_launch(cx.fetch, cx.x, cx.x_mod, w, bias, cx.mask_fuse ? xip : x0p, cx.B, ch, HW, dim, s);
            }, "init_conv", 2.0 * HW * ch * dim);
            ConvArgs a[3];
            const char* names[3] = {"mask_fusion_conv.0", "mask_fusion_conv.2", "mask_fusion_conv.4"};
            const Act* srcs[3] = {&xi, &f1, &f2};
            const Act* dsts[3] = {keep ? &z1 : &f1, keep ? &z2 : &f2, &x0};
            int tiles[3];
            for (int i = 0; i < 3 && !b.err; ++i) {
                a[i].s0.p = srcs[i]->p; a[i].s0.C = srcs[i]->C;
                if (i == 0) { a[i].s1.p = mask_nhwc.p; a[i].s1.C = ch; }
                a[i].Hs = H; a[i].Ws = W; a[i].KS = i == 0 ? 5 : 3; a[i].pad = i == 0 ? 2 : 1; a[i].out_act = (i < 2 && !keep);
                a[i].w = u->P(std::string(names[i]) + ".weight"); a[i].bias = u->R(std::string(names[i]) + ".bias");
                a[i].B = maxB; a[i].H = H; a[i].W = W; a[i].Cout = dsts[i]->C; a[i].out = dsts[i]->p; a[i].Cin = a[i].s0.C + a[i].s1.C;
                ConvGeom g;
                b.err = conv_plan(a[i], TILE_AUTO, &g);
                tiles[i] = g.tile;
            }
            if (b.err) return b.err;
            const ConvArgs a0 = a[0], a1 = a[1], a2 = a[2];
            const int t0 = tiles[0], t1 = tiles[1], t2 = tiles[2];
            const float *z1p = z1.p, *z2p = z2.p;
            float *f1p = f1.p, *f2p = f2.p;
            const size_t nf = (size_t)HW * 2 * dim;
            b.scope = "mask_fusion_conv";
            b.push([=](const FwdCtx& cx, hipStream_t s) -> int {
                if (!cx.mask_fuse) return FC_OK;
                ConvArgs q = a0; q.B = cx.B; FC_TRY(conv_launch(q, t0, s));
                if (keep) FC_TRY(silu_fwd_launch(z1p, nullptr, f1p, nf * cx.B, s));
                q = a1; q.B = cx.B; FC_TRY(conv_launch(q, t1, s));
                if (keep) FC_TRY(silu_fwd_launch(z2p, nullptr, f2p, nf * cx.B, s));
                q = a2; q.B = cx.B; return conv_launch(q, t2, s);
            }, "mask_fusion(3 x conv_igemm)", 2.0 * HW * (25.0 * (dim + ch) * 2 * dim + 9.0 * 2 * dim * 2 * dim + 9.0 * 2 * dim * dim));
            pl->fuse.present = true;
            pl->fuse.xi = xi; pl->fuse.mask = mask_nhwc; pl->fuse.z1 = z1; pl->fuse.f1 = f1; pl->fuse.z2 = z2; pl->fuse.f2 = f2; pl->fuse.x0 = x0;
        }
    }

    // -- down path (unet.py:326-343) --
    std::vector<Act> skips;
    Act x = x0;
    for (int i = 0; i < L && !b.err; ++i) {
        const std::string p = "downs." + std::to_string(i);
        Stat gn1;
        x = b.resblock(p + ".0", x, nullptr, cs[i], false, nullptr);
        skips.push_back(x);
        x = b.resblock(p + ".1", x, nullptr, cs[i], true, &gn1);
        x = b.linattn(p + ".2", x, gn1);
        skips.push_back(x);
        if (c.mask_cond && i < 2) x = b.mask_inject("down_mask_fusions." + std::to_string(i) + ".0", x, mask_nhwc);
        b.scope = p + ".3";
        ConvArgs a;
        a.s0.p = x.p; a.s0.C = x.C; a.Hs = x.H; a.Ws = x.W;
        if (i == L - 1) {
            a.KS = 3; a.pad = 1; a.w = u->P(p + ".3.weight"); a.bias = u->R(p + ".3.bias"); a.w4 = u->P8(p + ".3.weight");
            Act o = b.act(cs[i + 1], x.H, x.W);
            b.conv(a, o, 0, nullptr);
            pl->tape.push_back({3, (int)pl->convs.size()});
            pl->convs.push_back({p + ".3", x, o, 3, 1, 1, 0});
            x = o;
            pl->named[p + ".3"] = o;
        } else {
            a.KS = 2; a.pad = 0; a.stride = 2; a.w = u->P(p + ".3.1.weight"); a.bias = u->R(p + ".3.1.bias");
            Act o = b.act(cs[i + 1], x.H / 2, x.W / 2);
            b.conv(a, o, 0, nullptr);
            pl->tape.push_back({3, (int)pl->convs.size()});
            pl->convs.push_back({p + ".3.1", x, o, 2, 0, 2, 0});
            x = o;
            pl->named[p + ".3"] = o;
        }
    }
    // -- bottleneck (unet.py:345-347) --
    {
        Stat gn1;
        x = b.resblock("mid_block1", x, nullptr, cs[L], true, &gn1);
        if (!b.err) x = b.midattn(x, gn1);
        if (!b.err) x = b.resblock("mid_block2", x, nullptr, cs[L], false, nullptr);
    }
    // -- up path (unet.py:350-367) --
    for (int i = 0; i < L && !b.err; ++i) {
        const std::string p = "ups." + std::to_string(i);
        const int din = cs[L - 1 - i], dout = cs[L - i];
        Stat gn1;
        Act s1 = skips.back(); skips.pop_back();
        x = b.resblock(p + ".0", x, &s1, dout, false, nullptr);
        Act s2 = skips.back(); skips.pop_back();
        x = b.resblock(p + ".1", x, &s2, dout, true, &gn1);
        x = b.linattn(p + ".2", x, gn1);
        if (c.mask_cond && i < 2) x = b.mask_inject("up_mask_fusions." + std::to_string(i) + ".0", x, mask_nhwc);
        b.scope = p + ".3";
        ConvArgs a;
        a.s0.p = x.p; a.s0.C = x.C; a.Hs = x.H; a.Ws = x.W; a.KS = 3; a.pad = 1;
        if (i == L - 1) {
            a.w = u->P(p + ".3.weight"); a.bias = u->R(p + ".3.bias"); a.w4 = u->P8(p + ".3.weight");
            Act o = b.act(din, x.H, x.W);
            b.conv(a, o, 0, nullptr);
            pl->tape.push_back({3, (int)pl->convs.size()});
            pl->convs.push_back({p + ".3", x, o, 3, 1, 1, 0});
            x = o;
        } else {  // nn.Upsample(nearest x2) folded into the conv's loader (unet.py:42-46)
            Act o = b.act(din, x.H * 2, x.W * 2);
            // (round 3) the upsampling folded into the WEIGHTS -- four 2x2 parity convolutions on x in one launch, 4/9 of the multiply-adds
            // (plan.h conv_up2; FLOCODER_AMD_UPS_FOLD=0: off).  Training plans too: it is the same function of (x, w) up to fp32 rounding, and
            // the backward differentiates it in its 3x3 form from the same saved x (flowers-sized step 6.49 -> 6.45 ms).
            if (!b.conv_up2(a.s0, x, u->PUP(p + ".3.1.weight"), u->R(p + ".3.1.bias"), o, 0, nullptr)) {
                a.ups = 1; a.w = u->P(p + ".3.1.weight"); a.bias = u->R(p + ".3.1.bias"); a.w4 = u->P8(p + ".3.1.weight");
                b.conv(a, o, 0, nullptr);
            }
            pl->tape.push_back({3, (int)pl->convs.size()});
            pl->convs.push_back({p + ".3.1", x, o, 3, 1, 1, 1});
            x = o;
        }
        pl->named[p + ".3"] = x;
    }
    if (b.err) return b.err;
    // -- head (unet.py:369-372) --
    x = b.resblock("final_res_block", x, &x0, dim, false, nullptr);
    if (b.err) return b.err;
    pl->head = x;
    {
        const float *xp = x.p, *w = u->P("final_conv.weight"), *bias = u->R("final_conv.bias");
        b.scope = "final_conv";
        b.push([=](const FwdCtx& cx, hipStream_t s) { return final_conv_launch(xp, w, bias, cx.out, cx.B, dim, HW, ch, cx.euler, s); }, "final_conv", 2.0 * HW * dim * ch);
    }
    if (b.err) return b.err;
    pl->maxB = maxB; pl->H = H; pl->W = W;
    return FC_OK;
}

// integrator state for `rows` U-Net rows (library-owned so captured graphs never see caller pointers)
static int alloc_integrator(fc_unet* u, int rows, int H, int W) {
    const size_t nstate = (size_t)rows * u->cfg.channels * H * W;
    auto get = [&](size_t floats, float** out) -> int {
        void* p = nullptr;
        FC_TRY(dev_alloc(&p, (floats ? floats : 1) * sizeof(float), "integrator"));
        u->int_allocs.push_back(p);
        *out = static_cast<float*>(p);
        return FC_OK;
    };
    float* tmp = nullptr;
    FC_TRY(get(nstate, &u->y)); FC_TRY(get(nstate, &u->xs));
    FC_TRY(get(nstate, &u->k1)); FC_TRY(get(nstate, &u->k2)); FC_TRY(get(nstate, &u->k3));
    FC_TRY(get(nstate, &u->v2)); FC_TRY(get(nstate, &u->mask_own));
    FC_TRY(get(rows, &u->tvec));
    FC_TRY(get(4, &u->sc));
    FC_TRY(get(4, &tmp)); u->step = reinterpret_cast<int*>(tmp);
    FC_TRY(get(2 * (size_t)rows, &tmp)); u->ids_own = reinterpret_cast<int64_t*>(tmp);
    return FC_OK;
}

// Rows [r0, r0 + n) of the caller's batch as a context of their own (row r reads sample r % x_mod, class id r % ids_mod,
// no class at all from row null_from on -- the CFG layout of fc_unet_integrate).
static FwdCtx slice_ctx(const FwdCtx& c, int r0, int n, size_t sample_floats) {
    FwdCtx k = c;
    k.B = n;
    k.time = c.time + r0;
    k.out = c.out + (size_t)r0 * sample_floats;
    const int xs = r0 % c.x_mod;                 // first sample this slice reads
    k.x = c.x + (size_t)xs * sample_floats;
    if (c.mask) k.mask = c.mask + (size_t)xs * sample_floats;
    k.x_mod = c.x_mod - xs;
    if (c.ids) {
        if (c.null_from > 0 && r0 >= c.null_from) { k.ids = nullptr; k.null_from = 0; }
        else {
            const int is = r0 % c.ids_mod;
            k.ids = c.ids + is;
            k.ids_mod = c.ids_mod - is;
            k.null_from = c.null_from > 0 ? c.null_from - r0 : 0;
        }
    }
    return k;
}

// ---- process-wide guard of the meeting launches ------------------------------------------------------------------------------
// A launch whose workgroups wait for each other is only safe while no OTHER such launch can hold part of the CUs: two of them, each
// half resident, would wait for workgroups that cannot start (bounded by the spin limit, then NaN + error -- never a hang, never silent).
// Inside one process that cannot happen: a plan with meeting launches is ordered behind the previous one of ANY handle when that ran
// on another stream (one hipStreamWaitEvent, nothing when the event has completed).  Kernels without meetings beside it only delay it.
// Other processes on the same GPU are beyond this guard: that is what fc_unet_set_shared is for.
struct MeetGuard { std::mutex mu; hipEvent_t ev = nullptr; hipStream_t s = nullptr; const fc_unet* owner = nullptr; };
static MeetGuard g_meet[16];
static MeetGuard& meet_guard(int device) { return g_meet[device & 15]; }

static int plan_meets(const fc_unet* u) { return u->plan[0].n_meet + u->plan[1].n_meet; }

static int meet_enter(fc_unet* u, hipStream_t s) {
    if (!plan_meets(u)) return FC_OK;
    MeetGuard& g = meet_guard(u->device);
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.ev && g.s != s && hipEventQuery(g.ev) != hipSuccess) FC_HIP(hipStreamWaitEvent(s, g.ev, 0));
    return FC_OK;
}
static int meet_leave(fc_unet* u, hipStream_t s) {
    if (!plan_meets(u)) return FC_OK;
    MeetGuard& g = meet_guard(u->device);
    std::lock_guard<std::mutex> lk(g.mu);
    FC_HIP(hipEventRecord(u->ev_meet, s));
    g.ev = u->ev_meet; g.s = s; g.owner = u;
    // the error word follows the work: the host sees it at its next synchronisation with `s` (fc_unet_check) or at the next call
    FC_HIP(hipMemcpyAsync(const_cast<int*>(u->host_err), u->dev_err, sizeof(int), hipMemcpyDeviceToHost, s));
    return FC_OK;
}
static void meet_forget(const fc_unet* u) {
    MeetGuard& g = meet_guard(u->device);
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.owner == u) { g.ev = nullptr; g.s = nullptr; g.owner = nullptr; }
}
// refuse to go on after a timed-out meeting: the arena holds NaN-poisoned activations and whatever was returned since is invalid
static int check_poison(fc_unet* u) {
    if (u->host_err && *u->host_err) u->tail_failed = true;
    if (u->tail_failed)
        return fail(FC_E_STATE, "unet: a fused Block tail timed out waiting for its sample group (the GPU was shared with other work while "
                                "the exclusive plan ran); the affected samples are NaN.  Rebuild the plan -- fc_unet_set_shared(handle, 1) "
                                "selects the plan without cross-workgroup waits");
    return FC_OK;
}

// one chain: the conditioning MLPs (time / class embedding -> every block's scale and shift) on the second stream, joined before
// the first conv2; inside a captured step this becomes a parallel branch of the graph
static int run_single(fc_unet* u, const Plan& pl, const FwdCtx& c, hipStream_t s) {
    // Opt-in (FLOCODER_AMD_SIDE=1): worth +0.9 % while the conditioning chain took 46 us; since it takes 8 + 12 + 11 us the two forms
    // measure the same (631.4 / 629.4 vs 630.6 / 631.5 samples/s), so the plain single-stream order is the default.
    static const bool no_side = std::getenv("FLOCODER_AMD_SIDE") == nullptr;
    const int ns = pl.side_ops, nj = pl.join_at;
    if (no_side || ns <= 0 || nj <= ns || !u->stream2) return run_plan(pl, c, s);
    FC_HIP(hipEventRecord(u->ev_fork, s));
    FC_HIP(hipStreamWaitEvent(u->stream2, u->ev_fork, 0));
    for (int i = 0; i < ns; ++i) FC_TRY(pl.ops[i](c, u->stream2));
    FC_HIP(hipEventRecord(u->ev_join, u->stream2));
    for (int i = ns; i < nj; ++i) FC_TRY(pl.ops[i](c, s));
    FC_HIP(hipStreamWaitEvent(s, u->ev_join, 0));
    for (size_t i = nj; i < pl.ops.size(); ++i) FC_TRY(pl.ops[i](c, s));
    return FC_OK;
}

static int run_forward(fc_unet* u, const FwdCtx& c, hipStream_t s) {
    if (u->nchains < 2 || c.B < 2) return run_single(u, u->plan[0], c, s);
    // two chains: with CFG the conditional and the unconditional rows, otherwise the two halves of the batch
    const int r0 = (c.null_from > 0 && c.null_from < c.B) ? c.null_from : (c.B + 1) / 2;
    if (r0 > u->plan[0].maxB || c.B - r0 > u->plan[1].maxB) return fail(FC_E_STATE, "unet: chain plans too small for this batch");
    const size_t sf = (size_t)u->cfg.channels * u->H * u->W;
    FC_HIP(hipEventRecord(u->ev_fork, s));
    FC_HIP(hipStreamWaitEvent(u->stream2, u->ev_fork, 0));
    static const long long delay = [] { const char* e = std::getenv("FLOCODER_AMD_CHAIN_DELAY_US"); return e ? (long long)(std::atof(e) * 2100.0) : 0ll; }();
    if (delay > 0) FC_TRY(delay_launch(delay, u->stream2));   // the second chain runs half a kernel behind the first
    FC_TRY(run_plan(u->plan[1], slice_ctx(c, r0, c.B - r0, sf), u->stream2));
    FC_HIP(hipEventRecord(u->ev_join, u->stream2));
    FC_TRY(run_plan(u->plan[0], slice_ctx(c, 0, r0, sf), s));
    FC_HIP(hipStreamWaitEvent(s, u->ev_join, 0));
    return FC_OK;
}

}  // namespace fc

// =============================================================================================== C ABI
extern "C" {

int fc_abi_version(void) { return FC_ABI_VERSION; }
const char* fc_last_error(void) { return fc::last_error(); }

int fc_check_device(int device) {
    hipDeviceProp_t prop;
    FC_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(FC_E_ARCH, std::string("flocoder_amd kernels are built for gfx950 only; device reports ") + prop.gcnArchName);
    return FC_OK;
}

int fc_unet_create(const fc_unet_config* cfg, int device, fc_unet** out) {
    if (!cfg || !out) return fail(FC_E_ARG, "fc_unet_create: null argument");
    if (cfg->n_levels < 1 || cfg->n_levels > 8 || cfg->dim < 4 || cfg->channels < 1 || cfg->groups < 1)
        return fail(FC_E_ARG, "fc_unet_create: bad config");
    for (int i = 0; i < cfg->n_levels; ++i)
        if (cfg->dim_mults[i] < 1 || !is_pow2(cfg->dim * cfg->dim_mults[i] / cfg->groups) || (cfg->dim * cfg->dim_mults[i]) % cfg->groups)
            return fail(FC_E_SHAPE, "fc_unet_create: channels per GroupNorm group must be a power of two");
    std::unique_ptr<fc_unet> u(new fc_unet);
    u->cfg = *cfg;
    u->device = device;
    u->want_k8 = std::getenv("FLOCODER_AMD_NO_K8") == nullptr;
    FC_TRY(declare_all(u.get()));
    if (device < 0) {  // description only: parameter table without touching a GPU
        *out = u.release();
        return FC_OK;
    }
    FC_TRY(fc_check_device(device));
    FC_HIP(hipSetDevice(device));
    FC_TRY(conv_init());
    FC_TRY(linattn_fused_init());
    FC_TRY(linattn_sample_init());
    FC_TRY(temb_init());
    FC_TRY(u->alloc_device());
    const int half = cfg->dim / 2;
    std::vector<float> fr(half);
    const double lf = std::log(10000.0) / (half - 1);  // unet.py:26
    for (int k = 0; k < half; ++k) fr[k] = (float)std::exp((double)((float)k * (float)-lf));
    FC_HIP(hipMalloc(reinterpret_cast<void**>(&u->freqs), half * sizeof(float)));
    FC_HIP(hipMemcpy(u->freqs, fr.data(), half * sizeof(float), hipMemcpyHostToDevice));
    FC_HIP(hipStreamCreateWithFlags(&u->stream, hipStreamNonBlocking));
    FC_HIP(hipEventCreateWithFlags(&u->ev_in, hipEventDisableTiming));
    FC_HIP(hipEventCreateWithFlags(&u->ev_out, hipEventDisableTiming));
    FC_HIP(hipStreamCreateWithFlags(&u->stream2, hipStreamNonBlocking));
    FC_HIP(hipEventCreateWithFlags(&u->ev_fork, hipEventDisableTiming));
    FC_HIP(hipEventCreateWithFlags(&u->ev_join, hipEventDisableTiming));
    FC_HIP(hipEventCreateWithFlags(&u->ev_meet, hipEventDisableTiming));
    FC_HIP(hipMalloc(reinterpret_cast<void**>(&u->dev_err), sizeof(int)));
    FC_HIP(hipMemset(u->dev_err, 0, sizeof(int)));
    { void* hp = nullptr; FC_HIP(hipHostMalloc(&hp, sizeof(int), hipHostMallocDefault)); u->host_err = static_cast<volatile int*>(hp); *u->host_err = 0; }
    *out = u.release();
    return FC_OK;
}

void fc_unet_destroy(fc_unet* u) {
    if (!u) return;
    if (u->device < 0) { delete u; return; }
    (void)hipSetDevice(u->device);
    (void)hipDeviceSynchronize();
    free_plan(u);
    if (u->ts_dev) dev_free(u->ts_dev);
    if (u->pre) dev_free(u->pre);
    u->free_device();
    if (u->freqs) (void)hipFree(u->freqs);
    if (u->stream) (void)hipStreamDestroy(u->stream);
    if (u->ev_in) (void)hipEventDestroy(u->ev_in);
    if (u->ev_out) (void)hipEventDestroy(u->ev_out);
    if (u->stream2) (void)hipStreamDestroy(u->stream2);
    if (u->ev_fork) (void)hipEventDestroy(u->ev_fork);
    if (u->ev_join) (void)hipEventDestroy(u->ev_join);
    meet_forget(u);
    if (u->ev_meet) (void)hipEventDestroy(u->ev_meet);
    if (u->dev_err) (void)hipFree(u->dev_err);
    if (u->host_err) (void)hipHostFree(const_cast<int*>(u->host_err));
    delete u;
}

int fc_unet_param_count(const fc_unet* u) { return u ? (int)u->params.size() : 0; }

int fc_unet_param_info(const fc_unet* u, int i, const char** name, int64_t shape[4], int64_t* offset) {
    if (!u) return fail(FC_E_ARG, "fc_unet_param_info: null handle");
    return u->info(i, name, shape, offset);
}

int64_t fc_unet_param_numel(const fc_unet* u) { return u ? u->raw_numel : 0; }

int fc_unet_set_time_freqs(fc_unet* u, const float* freqs_host, int n) {
    if (!u || !freqs_host || n != u->cfg.dim / 2) return fail(FC_E_ARG, "fc_unet_set_time_freqs: expected dim/2 entries");
    FC_HIP(hipMemcpy(u->freqs, freqs_host, n * sizeof(float), hipMemcpyHostToDevice));
    return FC_OK;
}

int fc_unet_load_params(fc_unet* u, const float* flat, int64_t numel, int on_device, void* stream) {
    if (!u || !flat) return fail(FC_E_ARG, "fc_unet_load_params: null argument");
    if (u->device < 0) return fail(FC_E_STATE, "unet: created with device < 0 (description only)");
    FC_HIP(hipSetDevice(u->device));
    ++u->param_version;
    return u->load(flat, numel, on_device, static_cast<hipStream_t>(stream));
}

int fc_unet_reserved(const fc_unet* u, int* max_batch, int* height, int* width) {
    if (!u) return fail(FC_E_ARG, "fc_unet_reserved: null handle");
    if (max_batch) *max_batch = u->maxB;
    if (height) *height = u->maxB ? u->H : 0;
    if (width) *width = u->maxB ? u->W : 0;
    return FC_OK;
}

int fc_unet_reserve(fc_unet* u, int max_batch, int height, int width) {
    if (!u || max_batch < 1) return fail(FC_E_ARG, "fc_unet_reserve: bad argument");
    if (u->device < 0) return fail(FC_E_STATE, "unet: created with device < 0 (description only)");
    if (u->maxB >= max_batch && u->H == height && u->W == width) return FC_OK;
    FC_HIP(hipSetDevice(u->device));
    FC_HIP(hipDeviceSynchronize());
    free_plan(u);
    u->arena_touched(0);
    static const int want_chains = [] { const char* e = std::getenv("FLOCODER_AMD_CHAINS"); return e ? std::atoi(e) : 1; }();   // measured: 2 half-batch chains 404 vs 1 chain 446 samples/s (profiles/r01_c_*)
    u->nchains = (want_chains >= 2 && max_batch >= 2) ? 2 : 1;
    const int rows0 = u->nchains == 2 ? (max_batch + 1) / 2 : max_batch;
    int r = build_plan(u, &u->plan[0], rows0, height, width);
    if (r == FC_OK && u->nchains == 2) r = build_plan(u, &u->plan[1], rows0, height, width);
    if (r == FC_OK) r = alloc_integrator(u, max_batch, height, width);
    if (r != FC_OK) { free_plan(u); return r; }
    u->maxB = max_batch; u->H = height; u->W = width;
    return FC_OK;
}

static int check_ready(const fc_unet* u, int rows, int H, int W) {
    if (!u) return fail(FC_E_ARG, "null fc_unet");
    if (u->device < 0) return fail(FC_E_STATE, "unet: created with device < 0 (description only)");
    if (!u->loaded) return fail(FC_E_STATE, "unet: weights not loaded (fc_unet_load_params)");
    if (u->maxB < rows || u->H != H || u->W != W)
        return fail(FC_E_STATE, "unet: no plan for this shape; call fc_unet_reserve(rows >= " + std::to_string(rows) + ")");
    return FC_OK;
}

int fc_unet_forward(fc_unet* u, const float* x, const float* time, const int64_t* ids, const float* mask, int mask_is_ones, float* out,
                    int B, int H, int W, void* stream) {
    FC_TRY(check_ready(u, B, H, W));
    if (!x || !time || !out || B < 1) return fail(FC_E_ARG, "fc_unet_forward: null argument");
    FwdCtx c;
    c.x = x; c.x_mod = B; c.time = time; c.ids = ids; c.ids_mod = B; c.null_from = 0;
    c.mask = u->cfg.mask_cond ? mask : nullptr;
    c.mask_fuse = (c.mask && !mask_is_ones) ? 1 : 0;
    c.out = out; c.B = B;
    FC_TRY(check_poison(u));
    u->arena_touched(u->keep_all ? B : 0);
    FC_TRY(meet_enter(u, static_cast<hipStream_t>(stream)));
    FC_TRY(run_forward(u, c, static_cast<hipStream_t>(stream)));
    return meet_leave(u, static_cast<hipStream_t>(stream));
}

int fc_unet_set_shared(fc_unet* u, int shared) {
    if (!u) return fail(FC_E_ARG, "fc_unet_set_shared: null handle");
    if (u->device < 0) return fail(FC_E_STATE, "unet: created with device < 0 (description only)");
    if (u->shared == (shared != 0)) return FC_OK;
    u->shared = shared != 0;
    if (u->maxB > 0) {   // the plan in place was built for the other mode: drop it, the next reserve rebuilds
        FC_HIP(hipSetDevice(u->device));
        FC_HIP(hipDeviceSynchronize());
        const bool train = u->keep_all;
        free_plan(u);
        u->arena_touched(0);
        u->keep_all = train;
    }
    return FC_OK;
}

int fc_unet_meeting_launches(const fc_unet* u) { return u ? plan_meets(u) : 0; }

int fc_unet_check(fc_unet* u, void* stream, int synchronize) {
    if (!u) return fail(FC_E_ARG, "fc_unet_check: null handle");
    if (u->device < 0) return FC_OK;
    if (synchronize) {
        FC_HIP(hipSetDevice(u->device));
        FC_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    }
    return check_poison(u);
}

uint64_t fc_unet_arena_serial(const fc_unet* u) { return u ? u->arena_serial : 0; }

// Time every launch of the current plan on its own: each op is enqueued `repeats` times back to back between two
// events on `stream` (ops are idempotent: they only read their inputs), so host launch gaps do not pollute kernels
// that run longer than a launch takes to issue.  ms_out[i] = average milliseconds of op i.
static int g_stamp_op = -1;                     // diagnostics: fc_unet_profile_ops runs this plan entry once more with the conv stamps on
static unsigned long long* g_stamp_op_buf = nullptr;
int fc_debug_set_stamp_op(int op_index, void* buf_dev) { g_stamp_op = buf_dev ? op_index : -1; g_stamp_op_buf = static_cast<unsigned long long*>(buf_dev); return FC_OK; }

int fc_unet_profile_ops(fc_unet* u, int batch, int repeats, float* ms_out, int n_out, void* stream) {
    if (!u || !ms_out || repeats < 1) return fail(FC_E_ARG, "fc_unet_profile_ops: bad argument");
    FC_TRY(check_ready(u, batch, u->H, u->W));
    const Plan& pl0 = u->plan[0];
    if (batch > pl0.maxB) batch = pl0.maxB;   // ops are timed on chain 0's plan, at the rows one chain carries
    u->arena_touched(0);
    const int n = (int)pl0.ops.size();
    if (n_out < n) return fail(FC_E_ARG, "fc_unet_profile_ops: output array too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    FwdCtx c;
    c.x = u->y; c.x_mod = batch; c.time = u->tvec; c.ids = nullptr; c.ids_mod = batch; c.out = u->v2; c.B = batch;
    FC_HIP(hipMemsetAsync(u->tvec, 0, batch * sizeof(float), s));
    FC_TRY(run_plan(pl0, c, s));  // warm: every buffer holds finite data
    std::vector<hipEvent_t> ev(2 * n);
    for (auto& e : ev) FC_HIP(hipEventCreate(&e));
    int rc = FC_OK;
    for (int i = 0; i < n && rc == FC_OK; ++i) {
        (void)hipEventRecord(ev[2 * i], s);
        for (int r = 0; r < repeats && rc == FC_OK; ++r) rc = pl0.ops[i](c, s);
        (void)hipEventRecord(ev[2 * i + 1], s);
        if (i == g_stamp_op && g_stamp_op_buf && rc == FC_OK) {      // the same launch once more, writing its in-kernel phase stamps
            conv_set_stamp_buffer(g_stamp_op_buf);
            rc = pl0.ops[i](c, s);
            (void)hipStreamSynchronize(s);
            conv_set_stamp_buffer(nullptr);
        }
    }
    (void)hipStreamSynchronize(s);
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
        ms_out[i] = ms / repeats;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

int fc_unet_op_info(const fc_unet* u, int i, const char** kernel, const char** module, double* flops_per_sample) {
    if (!u || i < 0 || i >= (int)u->plan[0].ops.size()) return fail(FC_E_ARG, "fc_unet_op_info: index out of range");
    if (kernel) *kernel = u->plan[0].op_kernel[i].c_str();
    if (module) *module = u->plan[0].op_what[i].c_str();
    if (flops_per_sample) *flops_per_sample = u->plan[0].op_flops[i];
    return FC_OK;
}

int fc_unet_op_bytes(const fc_unet* u, int i, double* bytes_per_sample, double* bytes_per_launch) {
    if (!u || i < 0 || i >= (int)u->plan[0].ops.size()) return fail(FC_E_ARG, "fc_unet_op_bytes: index out of range");
    if (bytes_per_sample) *bytes_per_sample = u->plan[0].op_bytes_ps[i];
    if (bytes_per_launch) *bytes_per_launch = u->plan[0].op_bytes_fixed[i];
    return FC_OK;
}

int fc_unet_chains(const fc_unet* u, int* rows_per_chain) {
    if (!u) return 0;
    if (rows_per_chain) *rows_per_chain = u->plan[0].maxB;
    return u->nchains;
}

int fc_unet_fused_tail_errors(const fc_unet* u, int* count) {
    if (!u || !count) return fail(FC_E_ARG, "fc_unet_fused_tail_errors: null argument");
    *count = 0;
    if (u->device < 0 || !u->dev_err) return FC_OK;
    int v = 0;
    FC_HIP(hipMemcpy(&v, u->dev_err, sizeof(int), hipMemcpyDeviceToHost));   // synchronises with the null stream; callers sync their own
    *count = (v != 0 || u->tail_failed) ? 1 : 0;
    return FC_OK;
}

int fc_unet_plan_launches(const fc_unet* u) { return u ? (int)u->plan[0].ops.size() : 0; }
double fc_unet_flops_per_sample(const fc_unet* u) { return u ? u->plan[0].flops : 0.0; }

// -------------------------------------------------------------------------------- integrator
static uint32_t fbits(float f) { uint32_t v; std::memcpy(&v, &f, 4); return v; }

// enqueue one integration step on `s` (captured into a graph by the caller)
// Legacy Euler without CFG on one chain: the step needs nothing outside the plan (fc_unet_integrate publishes the first time)
static bool euler_tail_ok(const fc_unet* u, int method, bool cfg_on) {
    static const bool off = std::getenv("FLOCODER_AMD_NO_EULER_TAIL") != nullptr;
    return !off && method == FC_METHOD_EULER && !cfg_on && u->nchains < 2;
}

static int enqueue_step(fc_unet* u, int method, int B, bool cfg_on, float cfg, float dt_euler, float t_scale, bool has_ids, int mask_mode,
                        bool pre_on, hipStream_t s) {
    const int rows = cfg_on ? 2 * B : B, n = B * u->cfg.channels * u->H * u->W;
    FwdCtx c;
    if (pre_on) {   // conditioning rows of every evaluation are in u->pre: init_conv fetches slice *evalc, final_conv advances the counter
        c.fetch.all = u->pre_ss; c.fetch.evalc = u->step + 1; c.fetch.dst = u->plan[0].ss; c.fetch.n4 = rows * u->S / 4;
        c.euler.evalc = u->step + 1;
    }
    c.x_mod = B; c.time = u->tvec; c.ids = has_ids ? u->ids_own : nullptr; c.ids_mod = B; c.null_from = cfg_on ? B : 0;
    c.mask = mask_mode ? u->mask_own : nullptr; c.mask_fuse = mask_mode == 1;
    c.out = u->v2; c.B = rows;
    if (euler_tail_ok(u, method, cfg_on)) {   // the update and the next interval's time ride in final_conv: no launches around the plan
        c.x = u->y;
        c.euler.y = u->y; c.euler.dt = dt_euler; c.euler.step = u->step; c.euler.ts = u->ts_dev; c.euler.t_scale = t_scale;
        c.euler.sc = u->sc; c.euler.tvec = u->tvec; c.euler.rows = rows;
        return run_forward(u, c, s);
    }
    FC_TRY(ode_time_launch(u->step, u->ts_dev, t_scale, method == FC_METHOD_RK4, u->sc, u->tvec, rows, s));
    if (method == FC_METHOD_EULER) {
        c.x = u->y;
        FC_TRY(run_forward(u, c, s));
        return ode_euler_update_launch(u->y, u->v2, n, cfg_on, cfg, dt_euler, s);
    }
    c.x = u->y;
    FC_TRY(run_forward(u, c, s));                                                                                        // k1 = f(y, t)
    FC_TRY(ode_rk4_stage_launch(u->sc, u->y, u->xs, u->k1, u->v2, n, cfg_on, cfg, 0, 1, t_scale, u->tvec, rows, s));      // y + dt*k1/2, t+dt/2
    c.x = u->xs;
    FC_TRY(run_forward(u, c, s));                                                                                        // k2
    FC_TRY(ode_rk4_stage_launch(u->sc, u->y, u->xs, u->k2, u->v2, n, cfg_on, cfg, 0, 1, t_scale, u->tvec, rows, s));      // y + dt*k2/2, t+dt/2
    FC_TRY(run_forward(u, c, s));                                                                                        // k3
    FC_TRY(ode_rk4_stage_launch(u->sc, u->y, u->xs, u->k3, u->v2, n, cfg_on, cfg, 1, 2, t_scale, u->tvec, rows, s));      // y + dt*k3, t+dt
    FC_TRY(run_forward(u, c, s));                                                                                        // k4
    return ode_rk4_final_launch(u->sc, u->y, u->k1, u->k2, u->k3, u->v2, n, cfg_on, cfg, s);
}

int fc_unet_integrate(fc_unet* u, int method, float* x_dev, int B, int H, int W, const float* ts_host, int n_points, float dt_euler,
                      float t_scale, const int64_t* ids, float cfg_strength, const float* mask, int mask_is_ones, void* stream) {
    if (!u || !x_dev || !ts_host || B < 1 || n_points < 1) return fail(FC_E_ARG, "fc_unet_integrate: bad argument");
    if (method != FC_METHOD_EULER && method != FC_METHOD_RK4) return fail(FC_E_ARG, "fc_unet_integrate: unknown method");
    const bool has_ids = ids != nullptr && u->cfg.n_classes > 0;
    const bool cfg_on = has_ids && cfg_strength != 0.0f;   // sampling.py:69
    const int rows = cfg_on ? 2 * B : B;
    FC_TRY(check_ready(u, rows, H, W));
    FC_TRY(check_poison(u));
    u->arena_touched(0);
    const int mask_mode = (mask && u->cfg.mask_cond) ? (mask_is_ones ? 2 : 1) : 0;
    const int n_steps = method == FC_METHOD_RK4 ? n_points - 1 : n_points;
    hipStream_t caller = static_cast<hipStream_t>(stream), s = u->stream;
    FC_HIP(hipSetDevice(u->device));
    if (n_points + 1 > u->ts_cap) {  // grows only when a longer grid than ever before arrives
        FC_HIP(hipStreamSynchronize(s));
        if (u->ts_dev) dev_free(u->ts_dev);
        u->ts_cap = n_points < 1024 ? 1024 : n_points + 1;   // + 1: the fused Euler tail reads one entry past the grid after the last step
        FC_TRY(dev_alloc(reinterpret_cast<void**>(&u->ts_dev), u->ts_cap * sizeof(float), "integrator.ts"));
        for (auto& kv : u->graphs) (void)hipGraphExecDestroy(kv.second);
        u->graphs.clear();  // captured graphs hold the old ts pointer
    }
    const size_t nbytes = (size_t)B * u->cfg.channels * H * W * sizeof(float);
    // the library stream picks up after everything already queued on the caller's stream
    FC_HIP(hipEventRecord(u->ev_in, caller));
    FC_HIP(hipStreamWaitEvent(s, u->ev_in, 0));
    // pageable source: the runtime stages it before returning, so ts_host may be freed by the caller right away
    FC_HIP(hipMemcpyAsync(u->ts_dev, ts_host, n_points * sizeof(float), hipMemcpyHostToDevice, s));
    FC_HIP(hipMemsetAsync(u->step, 0, 2 * sizeof(int), s));   // step counter | evaluation counter
    FC_HIP(hipMemcpyAsync(u->y, x_dev, nbytes, hipMemcpyDeviceToDevice, s));
    if (has_ids) FC_HIP(hipMemcpyAsync(u->ids_own, ids, (size_t)B * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    if (mask_mode) FC_HIP(hipMemcpyAsync(u->mask_own, mask, nbytes, hipMemcpyDeviceToDevice, s));

    // Conditioning of every evaluation, once: the grid is known, so time MLP / class MLP / FiLM projections of all (evaluation, row)
    // pairs are three launches here instead of three at the head of each forward (44 us of every 1.6 ms step inside the replayed graph:
    // cold weights, latency-bound).  Rows are bit-identical to the per-forward ones (same kernels, same time arithmetic).
    static const bool no_pre = std::getenv("FLOCODER_AMD_NO_PRECOND") != nullptr;
    const int n_evals = method == FC_METHOD_RK4 ? 4 * n_steps : n_steps;
    const size_t R = (size_t)n_evals * rows, tvn = ((size_t)n_evals + 3) & ~(size_t)3;
    const size_t need = tvn + R * u->td * 3 + R * u->S;
    const bool pre_on = !no_pre && u->nchains < 2 && n_steps >= 2 && need * sizeof(float) <= (2ull << 30) && R < (1u << 30) / (unsigned)u->S;
    if (pre_on) {
        if (need > u->pre_cap) {
            FC_HIP(hipStreamSynchronize(s));
            if (u->pre) dev_free(u->pre);
            u->pre = nullptr; u->pre_cap = 0;
            FC_TRY(dev_alloc(reinterpret_cast<void**>(&u->pre), need * sizeof(float), "integrator.cond_table"));
            u->pre_cap = need;
            for (auto& kv : u->graphs) (void)hipGraphExecDestroy(kv.second);
            u->graphs.clear();  // captured graphs hold the old table pointer
        }
        float *tv = u->pre, *te = tv + tvn, *hh = te + R * u->td, *c1 = hh + R * u->td;
        if (u->pre_ss != c1 + R * u->td) {   // the table moved inside the buffer (another number of evaluations): graphs bake its address
            for (auto& kv : u->graphs) (void)hipGraphExecDestroy(kv.second);
            u->graphs.clear();
        }
        u->pre_ss = c1 + R * u->td;
        FC_TRY(ode_all_times_launch(u->ts_dev, n_steps, method == FC_METHOD_RK4, t_scale, tv, s));
        TembArgs ta = u->temb_proto;
        ta.B = (int)R; ta.time = tv; ta.rows_per_eval = rows; ta.class_ids = has_ids ? u->ids_own : nullptr; ta.class_batch_mod = B;
        ta.null_from = cfg_on ? B : 0; ta.t_out = te;
        FC_TRY(temb_launch(ta, hh, c1, s));
        // every ResnetBlock.mlp (SiLU -> Linear td -> 2*Cout, unet.py:79-82) of every row as ONE GEMM [R x td] . [td x S] on the
        // implicit-GEMM kernel (a 1x1 convolution over R one-pixel "images"): the per-forward VALU kernel re-reads the 4 MB weight
        // matrix for every eight rows (1.7 ms at R = 4096), this takes a tenth of that
        FC_TRY(silu_fwd_launch(te, nullptr, hh, R * u->td, s));
        ConvArgs ca;
        ca.s0.p = hh; ca.s0.C = u->td; ca.Cin = u->td; ca.Cout = u->S; ca.B = (int)R; ca.H = ca.W = ca.Hs = ca.Ws = 1; ca.KS = 1;
        ca.w = u->P("__ss_wt"); ca.bias = u->P("__ss_bias"); ca.out = u->pre_ss;
        FC_TRY(conv_launch(ca, TILE_AUTO, s));
    }
    if (euler_tail_ok(u, method, cfg_on))   // time of the first interval; every step publishes its successor's
        FC_TRY(ode_time_launch(u->step, u->ts_dev, t_scale, 0, u->sc, u->tvec, rows, s));
    FC_TRY(meet_enter(u, s));
    static const bool no_graph = std::getenv("FLOCODER_AMD_NO_GRAPH") != nullptr;
    if (no_graph) {
        for (int i = 0; i < n_steps; ++i) FC_TRY(enqueue_step(u, method, B, cfg_on, cfg_strength, dt_euler, t_scale, has_ids, mask_mode, pre_on, s));
    } else {
        // One graph holds SEVERAL consecutive intervals (round 3): the step counter, the time grid and the conditioning slice index all
        // live on the device, so a captured interval is position-independent and k of them in a row are one hipGraphLaunch instead
        // of k (the per-interval form left ~4 % of the trajectory between replays: 64 launches of a 70-node graph).  Capped by node
        // count; FLOCODER_AMD_GRAPH_STEPS=1 restores one interval per graph.
        static const int steps_env = [] { const char* e = std::getenv("FLOCODER_AMD_GRAPH_STEPS"); return e ? std::atoi(e) : 0; }();
        const int nodes_per_step = (int)u->plan[0].ops.size() * (method == FC_METHOD_RK4 ? 4 : 1) * (u->nchains == 2 ? 2 : 1) + 16;
        int per = steps_env > 0 ? steps_env : (6144 / nodes_per_step > 0 ? 6144 / nodes_per_step : 1);
        if (per > 255) per = 255;
        for (int left = n_steps; left > 0;) {
            const int k = left < per ? left : per;
            const auto key = std::make_tuple(method, B, (int)cfg_on, mask_mode, fbits(cfg_strength), fbits(dt_euler), fbits(t_scale),
                                             (int)has_ids | ((int)pre_on << 1) | (k << 2));
            auto it = u->graphs.find(key);
            if (it == u->graphs.end()) {
                hipGraph_t graph = nullptr;
                FC_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                int r = FC_OK;
                for (int j = 0; j < k && r == FC_OK; ++j)
                    r = enqueue_step(u, method, B, cfg_on, cfg_strength, dt_euler, t_scale, has_ids, mask_mode, pre_on, s);
                const hipError_t e = hipStreamEndCapture(s, &graph);
                if (r != FC_OK) { if (graph) (void)hipGraphDestroy(graph); return r; }
                if (e != hipSuccess) return fail(FC_E_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
                hipGraphExec_t exec = nullptr;
                FC_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                FC_HIP(hipGraphDestroy(graph));
                it = u->graphs.emplace(key, exec).first;
            }
            // The FIRST replay of a call waits, on the host, for everything this call has put on the stream in front of it (round 4).  Under
            // AMD_DIRECT_DISPATCH=0 -- the mode the sampler ships with -- ROCm 7.2 submits a graph from the calling thread while the plain
            // launches and copies issued just before it are still queued in the runtime's own submission thread: the replay overtook them.
            // Measured (tools/inflight_distinct.py: five calls with different noise / class ids, one at a time): a call's trajectory ran on
            // the PREVIOUS call's conditioning table (rel-L2 1.6e-2 against the oracle, the same value every time, in two or three calls of
            // five); FLOCODER_AMD_NO_GRAPH=1, AMD_DIRECT_DISPATCH=1 and this wait each give 1e-7 in all of them, an event wait on the
            // same stream does not.  bench.py never saw it: every timed step integrates the same samples, so a stale table is the right
            // one.  Replays that follow a replay are ordered (RK4: five graphs per call); work issued behind a replay is ordered as well.
            // Cost: the host idles for the prologue (~0.1 ms per call of 80 ms).  FLOCODER_AMD_GRAPH_FENCE=0 removes the wait (measurements only).
            static const bool fence = [] { const char* e = std::getenv("FLOCODER_AMD_GRAPH_FENCE"); return !(e && std::atoi(e) == 0); }();
            if (fence && left == n_steps) FC_HIP(hipStreamSynchronize(s));
            FC_HIP(hipGraphLaunch(it->second, s));
            left -= k;
        }
    }
    FC_HIP(hipMemcpyAsync(x_dev, u->y, nbytes, hipMemcpyDeviceToDevice, s));
    FC_TRY(meet_leave(u, s));
    FC_HIP(hipEventRecord(u->ev_out, s));
    FC_HIP(hipStreamWaitEvent(caller, u->ev_out, 0));
    return FC_OK;
}

// ---- debug / test hooks --------------------------------------------------------------------------
int fc_unet_debug_tensor(const fc_unet* u, const char* name, const float** ptr, int* C, int* H, int* W) {
    if (!u || !name) return fail(FC_E_ARG, "fc_unet_debug_tensor: null argument");
    auto it = u->plan[0].named.find(name);   // chain 0 = the first rows of the batch
    if (it == u->plan[0].named.end()) {
        it = u->bwd.named.find(name);        // "grad:<tap>": gradient of that tap after fc_unet_backward
        if (it == u->bwd.named.end()) return fail(FC_E_ARG, std::string("fc_unet_debug_tensor: no tap named ") + name);
    }
    *ptr = it->second.p; *C = it->second.C; *H = it->second.H; *W = it->second.W;
    return FC_OK;
}

// Test hook: put the arrival counter of one meeting launch out of step, so that its workgroups draw different epochs and every wait of
// that launch times out (bounded spins) -- the failure a shared device can cause, on demand.
int fc_debug_unet_break_meeting(fc_unet* u) {
    if (!u || u->plan[0].fin_sync.empty()) return fail(FC_E_STATE, "fc_debug_unet_break_meeting: the plan has no meeting launch");
    FC_HIP(hipDeviceSynchronize());
    unsigned v = 0;
    FC_HIP(hipMemcpy(&v, u->plan[0].fin_sync[0], sizeof(unsigned), hipMemcpyDeviceToHost));
    ++v;
    FC_HIP(hipMemcpy(u->plan[0].fin_sync[0], &v, sizeof(unsigned), hipMemcpyHostToDevice));
    return FC_OK;
}

int fc_debug_unet_break_meeting_kind(fc_unet* u, int kind) {      // the same for the first meeting launch of a kind (0 Block tail, 1 linear attention close)
    if (!u) return fail(FC_E_ARG, "fc_debug_unet_break_meeting_kind: null handle");
    const Plan& pl = u->plan[0];
    for (size_t i = 0; i < pl.fin_sync.size(); ++i)
        if (pl.fin_kind[i] == kind) {
            FC_HIP(hipDeviceSynchronize());
            unsigned v = 0;
            FC_HIP(hipMemcpy(&v, pl.fin_sync[i], sizeof(unsigned), hipMemcpyDeviceToHost));
            ++v;
            FC_HIP(hipMemcpy(pl.fin_sync[i], &v, sizeof(unsigned), hipMemcpyHostToDevice));
            return FC_OK;
        }
    return fail(FC_E_STATE, "fc_debug_unet_break_meeting_kind: the plan has no meeting launch of that kind");
}

int fc_debug_set_fused_tail(int on) {   // plans built from now on use (1) / do not use (0) the fused Block tails; < 0: back to the default (environment)
    fc::g_fused_tail = on < 0 ? -1 : (on ? 1 : 0);
    return FC_OK;
}

int fc_debug_set_conv_stamps(void* buf_dev) {
    conv_set_stamp_buffer(static_cast<unsigned long long*>(buf_dev));
    return FC_OK;
}

int fc_debug_copy(void* dst_dev, const void* src_dev, int64_t bytes, void* stream) {
    FC_HIP(hipMemcpyAsync(dst_dev, src_dev, (size_t)bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return FC_OK;
}

static int g_debug_conv_prec = 0;
int fc_debug_set_conv_precision(int mode) { g_debug_conv_prec = mode == 1 ? 1 : 0; return FC_OK; }   // arithmetic of fc_debug_conv launches (tests)

int fc_debug_conv(const float* src0, int c0, const float* src1, int c1, const float* w_oihw, const float* bias, const float* add,
                  float* out, float* stats_out, int groups_out, int* stats_T, float* stats_nt, int batch, int hs, int ws, int cout,
                  int ksize, int pad, int stride, int upsample, int out_act, int tile_cfg, int repeats, float* ms_out, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    FC_TRY(conv_init());
    ConvArgs a;
    a.s0.p = src0; a.s0.C = c0; a.s1.p = src1; a.s1.C = src1 ? c1 : 0;
    a.Cin = a.s0.C + a.s1.C; a.Cout = cout; a.B = batch; a.Hs = hs; a.Ws = ws;
    a.KS = ksize; a.pad = pad; a.stride = stride; a.ups = upsample;
    a.H = upsample ? hs * 2 : (hs + 2 * pad - ksize) / stride + 1;
    a.W = upsample ? ws * 2 : (ws + 2 * pad - ksize) / stride + 1;
    a.bias = bias; a.add = add; a.out = out; a.out_act = out_act;
    a.stats_out = stats_out; a.Gout = groups_out;
    a.prec = g_debug_conv_prec;
    float* wp = nullptr;
    FC_HIP(hipMalloc(reinterpret_cast<void**>(&wp), (size_t)cout * a.Cin * ksize * ksize * sizeof(float)));
    int r = pack_conv_launch(w_oihw, wp, cout, a.Cin, ksize, ksize, s);
    a.w = wp;
    float* wp8 = nullptr;       // the k-step-quad copy where the shapes allow it, as a U-Net plan would carry it (conv_pipe.hip FL_W4)
    if (r == FC_OK && ksize == 3 && a.Cin % 32 == 0 && cout % 32 == 0 && std::getenv("FLOCODER_AMD_NO_K8") == nullptr) {
        FC_HIP(hipMalloc(reinterpret_cast<void**>(&wp8), (size_t)cout * a.Cin * 9 * sizeof(float)));
        r = pack_conv_k8_launch(w_oihw, wp8, cout, a.Cin, 9, s);
        a.w4 = wp8;
    }
    float* wp3 = nullptr;       // ... and the split-bf16 copy when that arithmetic is asked for, as a codec plan would (pack kind 8)
    if (r == FC_OK && a.prec == 1 && (ksize == 3 || ksize == 1)) {
        const int ipad = (a.Cin + 15) / 16 * 16;
        FC_HIP(hipMalloc(reinterpret_cast<void**>(&wp3), (size_t)cout * ipad * ksize * ksize * sizeof(float)));
        r = pack_conv_b3_launch(w_oihw, wp3, cout, a.Cin, ksize * ksize, ipad, s);
        a.w_b3 = wp3;
    }
    ConvGeom g;
    if (r == FC_OK) r = conv_plan(a, tile_cfg, &g);
    if (r == FC_OK) { if (stats_T) *stats_T = g.T; if (stats_nt) *stats_nt = g.n_t; r = conv_launch(a, g.tile, s); }
    if (r == FC_OK && repeats > 0 && ms_out) {   // back-to-back timing of the same launch (warm caches)
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < repeats && r == FC_OK; ++i) r = conv_launch(a, g.tile, s);
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *ms_out = ms / repeats;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(wp);
    if (wp8) (void)hipFree(wp8);
    if (wp3) (void)hipFree(wp3);
    return r;
}

int fc_debug_conv_wgrad(const float* src0, int c0, const float* src1, int c1, const float* dy, int cout, int batch, int hs, int ws, int ksize,
                        int pad, int stride, int upsample, float* dw_out, float* db_out, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    FC_TRY(conv_wgrad_init());
    WgradArgs a;
    a.x0 = src0; a.C0 = c0; a.x1 = src1; a.C1 = src1 ? c1 : 0; a.Cin = a.C0 + a.C1; a.Cout = cout; a.dy = dy;
    a.B = batch; a.Hs = hs; a.Ws = ws; a.KS = ksize; a.pad = pad; a.stride = stride; a.ups = upsample;
    a.H = upsample ? hs * 2 : (hs + 2 * pad - ksize) / stride + 1;
    a.W = upsample ? ws * 2 : (ws + 2 * pad - ksize) / stride + 1;
    a.dw = dw_out; a.db = db_out;
    const size_t need = conv_wgrad_workspace(a);
    float* wsp = nullptr;
    if (need) FC_HIP(hipMalloc(reinterpret_cast<void**>(&wsp), need * sizeof(float)));
    a.ws = wsp; a.ws_floats = need;
    const int r = conv_wgrad_launch(a, s);
    (void)hipStreamSynchronize(s);
    if (wsp) (void)hipFree(wsp);
    return r;
}

int fc_ot_pairing(const float* source_dev, const float* target_dev, int batch, int64_t dim, float* dist_ws_dev, int64_t* perm_out_dev,
                  void* stream) {
    if (!source_dev || !target_dev || !dist_ws_dev || !perm_out_dev) return fail(FC_E_ARG, "fc_ot_pairing: null argument");
    return ot_launch(source_dev, target_dev, batch, dim, dist_ws_dev, perm_out_dev, static_cast<hipStream_t>(stream));
}

}  // extern "C"
