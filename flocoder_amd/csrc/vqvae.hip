// Native runtime of flocoder's VQVAE codec, encode / decode path (codecs.py:150-574; NATTEN-less, eval): parameter table with
// the reference's state_dict names and one launch plan each for VQVAE.encode (= self.encoder) and VQVAE.decode (= self.decoder at
// noise_strength 0).  quantize() (ResidualVQ, third-party) is not built.
//
// EncDecResidualBlock is post-norm:  out = SiLU(GN(conv2(SiLU(GN(conv1 x)))) + identity),  identity = x or GN(conv1x1/s x).
// On the implicit-GEMM kernel that is: conv1 (+ partials) | conv2 with GN+SiLU in its loader (+ partials) | optional 1x1
// projection (+ partials) | one finalize pass that normalises both branches, adds and applies SiLU.  Upsampling is
// conv3x3 -> SiLU (epilogue) -> PixelShuffle(2) (one re-layout pass).
#include <memory>

#include "plan.h"

using namespace fc;

struct fc_vqvae_config_s { int in_channels, hidden_channels, num_downsamples, internal_dim, vq_embedding_dim, decoder_nonlocal, natten = 0; };

struct fc_vqvae : fc::ParamStore {
    int device = 0;
    fc_vqvae_config_s c{3, 256, 3, 256, 4, 1, 0};
    fc::Plan enc, dec;
    int prec = 0;     // fc_vqvae_set_precision
};

namespace fc {

static int gn_groups(int proposed, int channels) {   // codecs.py:34-44
    if (channels % proposed == 0) return proposed;
    for (int c = proposed; c < channels; ++c) if (channels % c == 0) return c;
    return 1;
}
static int pad4(int c) { return (c + 3) & ~3; }

// Conv2d with channel counts that are not multiples of 4 (pixel-space ends): zero-padded operands
static void decl_conv_any(fc_vqvae* v, const std::string& n, int O, int I, int K) {
    const int Op = pad4(O), Ip = pad4(I);
    if (Op == O && Ip == I) { v->decl_conv(n, O, I, K); return; }
    v->declare(n + ".weight", {O, I, K, K});
    v->declare(n + ".bias", {O});
    const int64_t dst = v->pk_alloc(n + ".weight", (int64_t)Op * Ip * K * K);
    v->packops.push_back({4, v->params[v->pidx[n + ".weight"]].offset, dst, O, I, K * K, Op, Ip});
    const int64_t bd = v->pk_alloc(n + ".bias", Op);
    v->packops.push_back({3, v->params[v->pidx[n + ".bias"]].offset, bd, O, 0, 0, 0});
}
static const float* bias_of(const fc_vqvae* v, const std::string& n) { return v->pk.count(n + ".bias") ? v->P(n + ".bias") : v->R(n + ".bias"); }

// attn: 0 none | 1 AttnBlock ('full', codecs.py:52-90) | 2 NATTENBlock ('natten', codecs.py:93-145: GroupNorm, bias-free qkv / proj Linears, scalar gate)
static void decl_block(fc_vqvae* v, const std::string& n, int ci, int co, int stride, int attn) {
    const bool full_attn = attn == 1;
    decl_conv_any(v, n + ".conv1", co, ci, 3);
    v->decl_norm(n + ".norm1", co);
    v->decl_conv(n + ".conv2", co, co, 3);
    v->decl_norm(n + ".norm2", co);
    if (stride != 1 || ci != co) { decl_conv_any(v, n + ".downsample.0", co, ci, 1); v->decl_norm(n + ".downsample.1", co); }
    if (full_attn) {
        v->decl_norm(n + ".attn.norm.norm", co);
        for (const char* p : {".attn.q", ".attn.k", ".attn.v", ".attn.proj_out"}) v->decl_conv(n + p, co, co, 1);
    }
    if (attn == 2) {   // state_dict order of NATTENBlock: its own parameter (gamma) first, then the sub-modules norm, qkv, proj
        v->declare(n + ".attn.gamma", {1});
        v->decl_norm(n + ".attn.norm", co);
        for (auto pr : {std::make_pair(".attn.qkv", 3 * co), std::make_pair(".attn.proj", co)}) {   // nn.Linear [out][in] -> operand [in][out]
            v->declare(n + pr.first + ".weight", {pr.second, co});
            const int64_t dst = v->pk_alloc(n + pr.first + ".weight", (int64_t)pr.second * co);
            v->packops.push_back({2, v->params[v->pidx[n + pr.first + ".weight"]].offset, dst, pr.second, co, pr.second, 0});
        }
    }
}

static int declare_all(fc_vqvae* v) {
    const auto& c = v->c;
    const int nd = c.num_downsamples, hid = c.hidden_channels, emb = c.vq_embedding_dim;
    int cur = c.in_channels;
    const int na = c.natten ? 2 : 0;       // with NATTEN: the last two encoder levels and the bottleneck block (codecs.py:414-429)
    for (int i = 0; i < nd; ++i) {
        const int co = hid << i, at = i >= nd - 2 ? na : 0;
        decl_block(v, "encoder." + std::to_string(2 * i), cur, co, 2, at);
        decl_block(v, "encoder." + std::to_string(2 * i + 1), co, co, 1, at);
        cur = co;
    }
    decl_block(v, "encoder." + std::to_string(2 * nd), cur, c.internal_dim, 1, na);
    v->decl_conv("encoder." + std::to_string(2 * nd + 1), c.internal_dim, c.internal_dim, 1);
    v->decl_conv("encoder." + std::to_string(2 * nd + 2), emb, c.internal_dim, 1);
    v->decl_norm("encoder." + std::to_string(2 * nd + 3), emb);
    v->decl_conv("encoder." + std::to_string(2 * nd + 5), emb, emb, 3);
    // decoder (codecs.py:245-316)
    const std::string L = "decoder.layers.";
    int i = 0;
    if (c.decoder_nonlocal) {
        const int cr = emb / 2 > 1 ? emb / 2 : 1;
        v->declare(L + "0.q_proj.weight", {cr, emb, 1, 1}); v->declare(L + "0.q_proj.bias", {cr});
        v->declare(L + "0.k_proj.weight", {cr, emb, 1, 1}); v->declare(L + "0.k_proj.bias", {cr});
        v->declare(L + "0.v_proj.weight", {emb, emb, 1, 1}); v->declare(L + "0.v_proj.bias", {emb});
        v->declare(L + "0.out_proj.weight", {emb, emb, 1, 1}); v->declare(L + "0.out_proj.bias", {emb});
        i = 1;
    }
    cur = hid << (nd - 1);
    v->decl_conv(L + std::to_string(i), c.internal_dim, emb, 1);
    v->decl_norm(L + std::to_string(i + 1), c.internal_dim);
    v->decl_conv(L + std::to_string(i + 3), cur, c.internal_dim, 1);
    decl_block(v, L + std::to_string(i + 5), cur, cur, 1, c.decoder_nonlocal ? 1 : na);   // attn = 'full' if decoder_nonlocal else 'natten' (codecs.py:266)
    i += 6;
    for (int lvl = nd - 1; lvl >= 0; --lvl) {
        int co = hid << (lvl - 1 > 0 ? lvl - 1 : 0);
        if (lvl == 0) co = hid;
        v->decl_conv(L + std::to_string(i), 4 * cur, cur, 3);
        decl_block(v, L + std::to_string(i + 4), cur, co, 1, lvl > nd - 2 ? na : 0);                // first upsampling level only (codecs.py:275-278)
        decl_block(v, L + std::to_string(i + 6), co, co, 1, 0);
        cur = co;
        i += 7;
    }
    v->decl_conv(L + std::to_string(i + 1), 64, cur, 3);
    decl_conv_any(v, L + std::to_string(i + 4), c.in_channels, 64, 3);
    return FC_OK;
}

struct QBuilder : PlanBuilder {
    fc_vqvae* v;
    QBuilder(fc_vqvae* v_, Plan* pl_, int B_) : v(v_) { pl = pl_; B = B_; store = v_; }

    SrcXform gn(const Stat& st, const std::string& norm, int mode, float eps = 1e-5f) {
        return xf_of(st, mode, v->R(norm + ".weight"), v->R(norm + ".bias"), nullptr, 0, eps);
    }

    // EncDecResidualBlock (codecs.py:150-214).  x is a raw NHWC tensor with x.C possibly padded to 4 (pixel input).
    Act block(const std::string& n, const Act& x, int co, int stride) {
        scope = n;
        const int G = gn_groups(8, co), Ho = x.H / stride, Wo = x.W / stride;
        Act h1 = act(co, Ho, Wo);
        Stat s1, s2, sd;
        ConvArgs a;
        a.s0.p = x.p; a.s0.C = x.C; a.Hs = x.H; a.Ws = x.W; a.KS = 3; a.pad = 1; a.stride = stride;
        a.w = v->P(n + ".conv1.weight"); a.bias = bias_of(v, n + ".conv1");
        conv(a, h1, G, &s1);
        Act mid_in = h1;
        SrcXform mid_xf = gn(s1, n + ".norm1", 2);
        bool mid_raw = false;
        if (v->has(n + ".attn.q.weight")) {                       // attention='full': AttnBlock on the activated tensor (codecs.py:190-192)
            Act a1 = act(co, Ho, Wo);
            FinalizeArgs f;
            f.h = h1.p; f.xf = mid_xf; f.y = a1.p; f.HW = Ho * Wo; f.C = co;
            push([f](const FwdCtx& c, hipStream_t s) { FinalizeArgs g = f; g.B = c.B; return finalize_launch(g, s); }, "finalize");
            const int G32 = gn_groups(32, co);
            Stat sa = stat(G32, 1, (float)(Ho * Wo * (co / G32)));
            float* sp = sa.p; const float* ap = a1.p; const int HW = Ho * Wo;
            push([=](const FwdCtx& c, hipStream_t s) { return gn_stats_launch(ap, sp, c.B, HW, co, G32, s); }, "gn_stats");
            AttnWeights w{v->P(n + ".attn.q.weight"), v->R(n + ".attn.q.bias"), v->P(n + ".attn.k.weight"), v->R(n + ".attn.k.bias"),
                          v->P(n + ".attn.v.weight"), v->R(n + ".attn.v.bias"), v->P(n + ".attn.proj_out.weight"), v->R(n + ".attn.proj_out.bias")};
            Act a2 = attention_block(a1, gn(sa, n + ".attn.norm.norm", 1, 1e-6f), w, 0, nullptr);
            release(a1);
            mid_in = a2; mid_raw = true;
        }
        if (v->has(n + ".attn.qkv.weight")) {                     // attention='natten': NATTENBlock on the activated tensor (codecs.py:93-145)
            Act a1 = act(co, Ho, Wo);
            FinalizeArgs f;
            f.h = h1.p; f.xf = mid_xf; f.y = a1.p; f.HW = Ho * Wo; f.C = co;
            push([f](const FwdCtx& c, hipStream_t s) { FinalizeArgs g = f; g.B = c.B; return finalize_launch(g, s); }, "finalize");
            const int G8 = gn_groups(8, co);
            Stat sa = stat(G8, 1, (float)(Ho * Wo * (co / G8)));
            float* sp = sa.p; const float* ap = a1.p; const int HW = Ho * Wo;
            push([=](const FwdCtx& c, hipStream_t s) { return gn_stats_launch(ap, sp, c.B, HW, co, G8, s); }, "gn_stats");
            Act qkv = act(3 * co, Ho, Wo), att = act(co, Ho, Wo), a2 = act(co, Ho, Wo);
            ConvArgs q;
            q.s0.p = a1.p; q.s0.C = co; q.s0.xf = gn(sa, n + ".attn.norm", 1);
            q.Hs = Ho; q.Ws = Wo; q.KS = 1;
            q.w = v->P(n + ".attn.qkv.weight");
            conv(q, qkv, 0, nullptr);
            const float *qp = qkv.p, *gam = v->R(n + ".attn.gamma");
            float* tp = att.p;
            const int mode = v->c.natten;
            push([=](const FwdCtx& c, hipStream_t s) { return na2d_launch(qp, tp, gam, c.B, Ho, Wo, co, 8, 7, mode, s); }, "na2d",
                 2.0 * 2 * 49 * (double)co * Ho * Wo);
            ConvArgs pj;                                            // identity + gamma * proj(att): gamma rides in na2d's output (proj is linear, no bias)
            pj.s0.p = att.p; pj.s0.C = co; pj.Hs = Ho; pj.Ws = Wo; pj.KS = 1;
            pj.w = v->P(n + ".attn.proj.weight"); pj.add = a1.p;
            conv(pj, a2, 0, nullptr);
            release(qkv); release(att); release(a1);
            mid_in = a2; mid_raw = true;
        }
        // (round 3) without an attention in between, the activated tensor is materialised as well instead of being normalised in conv2's staging
        // waves once per output-channel tile and halo copy (vae.hip prenorm_src has the measurements); FLOCODER_AMD_VAE_PRENORM=fused: as before
        static const bool fused_prenorm = [] { const char* e = std::getenv("FLOCODER_AMD_VAE_PRENORM"); return e && std::string(e) == "fused"; }();
        if (!mid_raw && !fused_prenorm && !err) {
            Act a1 = act(co, Ho, Wo);
            FinalizeArgs f;
            f.h = h1.p; f.xf = mid_xf; f.y = a1.p; f.HW = Ho * Wo; f.C = co;
            push([f](const FwdCtx& c, hipStream_t s) { FinalizeArgs g = f; g.B = c.B; return finalize_launch(g, s); }, "finalize");
            mid_in = a1; mid_raw = true;
        }
        Act h2 = act(co, Ho, Wo);
        ConvArgs b;
        b.s0.p = mid_in.p; b.s0.C = co;
        if (!mid_raw) b.s0.xf = mid_xf;
        b.Hs = Ho; b.Ws = Wo; b.KS = 3; b.pad = 1;
        b.w = v->P(n + ".conv2.weight"); b.bias = v->R(n + ".conv2.bias");
        conv(b, h2, G, &s2);
        if (mid_raw) release(mid_in);
        Act out = act(co, Ho, Wo);
        FinalizeArgs f;
        f.h = h2.p; f.xf = gn(s2, n + ".norm2", 1); f.y = out.p; f.HW = Ho * Wo; f.C = co; f.act_after_add = 1;
        Act d;
        if (v->has(n + ".downsample.0.weight")) {
            d = act(co, Ho, Wo);
            ConvArgs c;
            c.s0.p = x.p; c.s0.C = x.C; c.Hs = x.H; c.Ws = x.W; c.KS = 1; c.stride = stride;
            c.w = v->P(n + ".downsample.0.weight"); c.bias = bias_of(v, n + ".downsample.0");
            conv(c, d, G, &sd);
            f.res = d.p; f.xf_res = gn(sd, n + ".downsample.1", 1);
        } else {
            f.res = x.p;
        }
        if (!err) push([f](const FwdCtx& c, hipStream_t s) { FinalizeArgs g = f; g.B = c.B; return finalize_launch(g, s); }, "finalize");
        release(h1); release(h2);
        if (d.p) release(d);
        return out;
    }
};

static int build_encoder(fc_vqvae* v, int maxB, int H, int W) {
    v->enc.release();
    const auto& c = v->c;
    const int nd = c.num_downsamples, emb = c.vq_embedding_dim;
    if (!is_pow2(H) || !is_pow2(W) || (H >> nd) < 4 || (W >> nd) < 4) return fail(FC_E_SHAPE, "vqvae: image size must be a power of two, >= 4 latent pixels per side");
    QBuilder b(v, &v->enc, maxB);
    b.conv_prec = v->prec;
    const int ic = c.in_channels, icp = pad4(ic);
    Act x = b.act(icp, H, W);
    float* xp = x.p;
    b.scope = "input";
    b.push([=](const FwdCtx& cx, hipStream_t s) { return nchw_to_nhwc_launch(cx.x, xp, cx.B, ic, H * W, icp, cx.B, s); }, "nchw_to_nhwc");
    for (int i = 0; i < nd && !b.err; ++i) {
        Act y = b.block("encoder." + std::to_string(2 * i), x, c.hidden_channels << i, 2);
        b.release(x);
        Act z = b.block("encoder." + std::to_string(2 * i + 1), y, c.hidden_channels << i, 1);
        b.release(y);
        x = z;
    }
    if (b.err) return b.err;
    Act y = b.block("encoder." + std::to_string(2 * nd), x, c.internal_dim, 1);
    b.release(x);
    if (b.err) return b.err;
    Act t1 = b.act(c.internal_dim, y.H, y.W), t2 = b.act(emb, y.H, y.W), t3 = b.act(emb, y.H, y.W);
    Stat st;
    {
        b.scope = "encoder.compress";
        ConvArgs a;
        a.s0.p = y.p; a.s0.C = y.C; a.Hs = y.H; a.Ws = y.W; a.KS = 1;
        a.w = v->P("encoder." + std::to_string(2 * nd + 1) + ".weight"); a.bias = v->R("encoder." + std::to_string(2 * nd + 1) + ".bias");
        b.conv(a, t1, 0, nullptr);
        ConvArgs q;
        q.s0.p = t1.p; q.s0.C = t1.C; q.Hs = y.H; q.Ws = y.W; q.KS = 1;
        q.w = v->P("encoder." + std::to_string(2 * nd + 2) + ".weight"); q.bias = v->R("encoder." + std::to_string(2 * nd + 2) + ".bias");
        b.conv(q, t2, gn_groups(2, emb), &st);
        ConvArgs r;
        r.s0.p = t2.p; r.s0.C = emb; r.s0.xf = b.gn(st, "encoder." + std::to_string(2 * nd + 3), 2);
        r.Hs = y.H; r.Ws = y.W; r.KS = 3; r.pad = 1;
        r.w = v->P("encoder." + std::to_string(2 * nd + 5) + ".weight"); r.bias = v->R("encoder." + std::to_string(2 * nd + 5) + ".bias");
        b.conv(r, t3, 0, nullptr);
    }
    if (b.err) return b.err;
    const float* zp = t3.p;
    const int hw = y.H * y.W;
    b.scope = "z";
    b.push([=](const FwdCtx& cx, hipStream_t s) { return nhwc_to_nchw_launch(zp, cx.out, cx.B, emb, hw, emb, s); }, "nhwc_to_nchw");
    v->enc.maxB = maxB; v->enc.H = H; v->enc.W = W;
    return FC_OK;
}

static int build_decoder(fc_vqvae* v, int maxB, int h, int w) {
    v->dec.release();
    const auto& c = v->c;
    const int nd = c.num_downsamples, emb = c.vq_embedding_dim, hid = c.hidden_channels;
    if (!is_pow2(h) || !is_pow2(w) || h < 4 || w < 4) return fail(FC_E_SHAPE, "vqvae: latent size must be a power of two >= 4");
    if (emb & 3) return fail(FC_E_SHAPE, "vqvae: vq_embedding_dim must be a multiple of 4");
    QBuilder b(v, &v->dec, maxB);
    b.conv_prec = v->prec;
    const std::string L = "decoder.layers.";
    Act z = b.act(emb, h, w);
    float* zp = z.p;
    b.scope = "input";
    b.push([=](const FwdCtx& cx, hipStream_t s) { return nchw_to_nhwc_launch(cx.x, zp, cx.B, emb, h * w, emb, cx.B, s); }, "nchw_to_nhwc");
    int i = 0;
    if (c.decoder_nonlocal) {   // SpatialNonLocalAttention(vq_embedding_dim), codecs.py:252
        Act z2 = b.act(emb, h, w);
        const float *wq = v->R(L + "0.q_proj.weight"), *bq = v->R(L + "0.q_proj.bias"), *wk = v->R(L + "0.k_proj.weight"), *bk = v->R(L + "0.k_proj.bias");
        const float *wv = v->R(L + "0.v_proj.weight"), *bv = v->R(L + "0.v_proj.bias"), *wo = v->R(L + "0.out_proj.weight"), *bo = v->R(L + "0.out_proj.bias");
        float* op = z2.p; const float* ip = z.p;
        const int n = h * w, cr = emb / 2 > 1 ? emb / 2 : 1;
        b.scope = L + "0";
        b.push([=](const FwdCtx& cx, hipStream_t s) { return rope_attn_launch(ip, wq, bq, wk, bk, wv, bv, wo, bo, op, cx.B, n, emb, cr, s); }, "rope_attn");
        z = z2;
        i = 1;
    }
    int cur = hid << (nd - 1);
    Act t1 = b.act(c.internal_dim, h, w), x = b.act(cur, h, w);
    Stat st;
    {
        b.scope = L + std::to_string(i);
        ConvArgs a;
        a.s0.p = z.p; a.s0.C = emb; a.Hs = h; a.Ws = w; a.KS = 1;
        a.w = v->P(L + std::to_string(i) + ".weight"); a.bias = v->R(L + std::to_string(i) + ".bias");
        b.conv(a, t1, gn_groups(emb, c.internal_dim), &st);
        ConvArgs q;
        q.s0.p = t1.p; q.s0.C = t1.C; q.s0.xf = b.gn(st, L + std::to_string(i + 1), 2);
        q.Hs = h; q.Ws = w; q.KS = 1;
        q.w = v->P(L + std::to_string(i + 3) + ".weight"); q.bias = v->R(L + std::to_string(i + 3) + ".bias");
        b.conv(q, x, 0, nullptr);
    }
    if (b.err) return b.err;
    {
        Act y = b.block(L + std::to_string(i + 5), x, cur, 1);
        b.release(x);
        x = y;
    }
    i += 6;
    for (int lvl = nd - 1; lvl >= 0 && !b.err; --lvl) {
        int co = hid << (lvl - 1 > 0 ? lvl - 1 : 0);
        if (lvl == 0) co = hid;
        b.scope = L + std::to_string(i);
        Act u = b.act(4 * cur, x.H, x.W), ps = b.act(cur, 2 * x.H, 2 * x.W);
        ConvArgs a;
        a.s0.p = x.p; a.s0.C = cur; a.Hs = x.H; a.Ws = x.W; a.KS = 3; a.pad = 1; a.out_act = 1;     // conv, SiLU
        a.w = v->P(L + std::to_string(i) + ".weight"); a.bias = v->R(L + std::to_string(i) + ".bias");
        b.conv(a, u, 0, nullptr);
        const float* up = u.p; float* pp = ps.p;
        const int Hs = x.H, Ws = x.W, cc = cur;
        b.push([=](const FwdCtx& cx, hipStream_t s) { return pixel_shuffle2_nhwc_launch(up, pp, cx.B, Hs, Ws, cc, s); }, "pixel_shuffle");
        b.release(x); b.release(u);
        Act y1 = b.block(L + std::to_string(i + 4), ps, co, 1);
        b.release(ps);
        Act y2 = b.block(L + std::to_string(i + 6), y1, co, 1);
        b.release(y1);
        x = y2; cur = co;
        i += 7;
    }
    if (b.err) return b.err;
    const int ic = c.in_channels, icp = pad4(ic);
    Act f1 = b.act(64, x.H, x.W), f2 = b.act(icp, x.H, x.W);
    {
        b.scope = "decoder.final";
        ConvArgs a;
        a.s0.p = x.p; a.s0.C = cur; a.Hs = x.H; a.Ws = x.W; a.KS = 3; a.pad = 1; a.out_act = 1;
        a.w = v->P(L + std::to_string(i + 1) + ".weight"); a.bias = v->R(L + std::to_string(i + 1) + ".bias");
        b.conv(a, f1, 0, nullptr);
        ConvArgs q;
        q.s0.p = f1.p; q.s0.C = 64; q.Hs = x.H; q.Ws = x.W; q.KS = 3; q.pad = 1;
        q.w = v->P(L + std::to_string(i + 4) + ".weight"); q.bias = bias_of(v, L + std::to_string(i + 4));
        b.conv(q, f2, 0, nullptr);
    }
    if (b.err) return b.err;
    const float* yp = f2.p;
    const int HW = x.H * x.W;
    b.scope = "recon";
    b.push([=](const FwdCtx& cx, hipStream_t s) { return nhwc_to_nchw_launch(yp, cx.out, cx.B, ic, HW, icp, s); }, "nhwc_to_nchw");
    v->dec.maxB = maxB; v->dec.H = h; v->dec.W = w;
    return FC_OK;
}

}  // namespace fc

extern "C" {

int fc_vqvae_create_ex(int in_channels, int hidden_channels, int num_downsamples, int internal_dim, int vq_embedding_dim, int decoder_nonlocal,
                       int natten_layout, int device, fc_vqvae** out) {
    if (!out || in_channels < 1 || hidden_channels < 8 || num_downsamples < 1 || num_downsamples > 6 || internal_dim < 4 || vq_embedding_dim < 1)
        return fail(FC_E_ARG, "fc_vqvae_create: bad config");
    if (natten_layout < 0 || natten_layout > 2) return fail(FC_E_ARG, "fc_vqvae_create: natten_layout must be 0 (no NATTEN blocks), 1 or 2");
    if ((hidden_channels & 3) || (internal_dim & 3) || (vq_embedding_dim & 3)) return fail(FC_E_SHAPE, "vqvae: hidden, internal and embedding widths must be multiples of 4");
    if (natten_layout && ((hidden_channels % 32) || (internal_dim % 32)))
        return fail(FC_E_SHAPE, "vqvae: NATTEN blocks need widths that are multiples of 32 (8 heads, head_dim a multiple of 4)");
    std::unique_ptr<fc_vqvae> v(new fc_vqvae);
    v->device = device;
    v->c = {in_channels, hidden_channels, num_downsamples, internal_dim, vq_embedding_dim, decoder_nonlocal, natten_layout};
    v->want_b3 = device >= 0;      // the split-bf16 copies of the conv weights (set_precision): +1x the conv weights in HBM
    FC_TRY(declare_all(v.get()));
    if (device < 0) { *out = v.release(); return FC_OK; }
    FC_TRY(fc_check_device(device));
    FC_HIP(hipSetDevice(device));
    FC_TRY(conv_init());
    FC_TRY(v->alloc_device());
    FC_HIP(hipMemset(v->packed, 0, (size_t)(v->packed_numel ? v->packed_numel : 4) * sizeof(float)));
    *out = v.release();
    return FC_OK;
}
int fc_vqvae_create(int in_channels, int hidden_channels, int num_downsamples, int internal_dim, int vq_embedding_dim, int decoder_nonlocal,
                    int device, fc_vqvae** out) {
    return fc_vqvae_create_ex(in_channels, hidden_channels, num_downsamples, internal_dim, vq_embedding_dim, decoder_nonlocal, 0, device, out);
}
void fc_vqvae_destroy(fc_vqvae* v) {
    if (!v) return;
    if (v->device >= 0) { (void)hipSetDevice(v->device); (void)hipDeviceSynchronize(); v->enc.release(); v->dec.release(); v->free_device(); }
    delete v;
}
int fc_vqvae_param_count(const fc_vqvae* v) { return v ? (int)v->params.size() : 0; }
int64_t fc_vqvae_param_numel(const fc_vqvae* v) { return v ? v->raw_numel : 0; }
int fc_vqvae_param_info(const fc_vqvae* v, int i, const char** name, int64_t shape[4], int64_t* offset) {
    if (!v) return fail(FC_E_ARG, "fc_vqvae_param_info: null handle");
    return v->info(i, name, shape, offset);
}
int fc_vqvae_load_params(fc_vqvae* v, const float* flat, int64_t numel, int on_device, void* stream) {
    if (!v || !flat || v->device < 0) return fail(FC_E_ARG, "fc_vqvae_load_params: bad argument");
    FC_HIP(hipSetDevice(v->device));
    return v->load(flat, numel, on_device, static_cast<hipStream_t>(stream));
}
int fc_vqvae_set_precision(fc_vqvae* v, int mode) {
    if (!v || (mode != 0 && mode != 1)) return fail(FC_E_ARG, "fc_vqvae_set_precision: mode is 0 (fp32) or 1 (split-bf16)");
    if (v->prec == mode) return FC_OK;
    v->prec = mode;
    if (v->device >= 0) {
        FC_HIP(hipSetDevice(v->device));
        FC_HIP(hipDeviceSynchronize());
        v->enc.release(); v->dec.release();
    }
    return FC_OK;
}

int fc_vqvae_reserve_encode(fc_vqvae* v, int max_batch, int height, int width) {
    if (!v || max_batch < 1 || v->device < 0) return fail(FC_E_ARG, "fc_vqvae_reserve_encode: bad argument");
    if (v->enc.maxB >= max_batch && v->enc.H == height && v->enc.W == width) return FC_OK;
    FC_HIP(hipSetDevice(v->device));
    FC_HIP(hipDeviceSynchronize());
    const int r = build_encoder(v, max_batch, height, width);
    if (r != FC_OK) v->enc.release();
    return r;
}
int fc_vqvae_reserve_decode(fc_vqvae* v, int max_batch, int lat_height, int lat_width) {
    if (!v || max_batch < 1 || v->device < 0) return fail(FC_E_ARG, "fc_vqvae_reserve_decode: bad argument");
    if (v->dec.maxB >= max_batch && v->dec.H == lat_height && v->dec.W == lat_width) return FC_OK;
    FC_HIP(hipSetDevice(v->device));
    FC_HIP(hipDeviceSynchronize());
    const int r = build_decoder(v, max_batch, lat_height, lat_width);
    if (r != FC_OK) v->dec.release();
    return r;
}
static int run_vq(const fc_vqvae* v, bool decode, const float* in, float* out, int B, int H, int W, void* stream) {
    if (!v || !in || !out || B < 1) return fail(FC_E_ARG, "vqvae: null argument");
    if (!v->loaded) return fail(FC_E_STATE, "vqvae: weights not loaded (fc_vqvae_load_params)");
    const Plan& pl = decode ? v->dec : v->enc;
    if (pl.maxB < B || pl.H != H || pl.W != W) return fail(FC_E_STATE, "vqvae: no plan for this shape; call fc_vqvae_reserve_* first");
    FwdCtx c;
    c.x = in; c.x_mod = B; c.out = out; c.B = B;
    return run_plan(pl, c, static_cast<hipStream_t>(stream));
}
int fc_vqvae_encode(fc_vqvae* v, const float* x_dev, float* z_out_dev, int batch, int height, int width, void* stream) {
    return run_vq(v, false, x_dev, z_out_dev, batch, height, width, stream);
}
int fc_vqvae_decode(fc_vqvae* v, const float* z_dev, float* x_out_dev, int batch, int lat_height, int lat_width, void* stream) {
    return run_vq(v, true, z_dev, x_out_dev, batch, lat_height, lat_width, stream);
}
double fc_vqvae_flops_per_sample(const fc_vqvae* v, int decode) { return v ? (decode ? v->dec.flops : v->enc.flops) : 0.0; }
int fc_vqvae_plan_launches(const fc_vqvae* v, int decode) { return v ? (int)(decode ? v->dec.ops.size() : v->enc.ops.size()) : 0; }

int fc_vqvae_op_info(const fc_vqvae* v, int decode, int i, const char** kernel, const char** module, double* flops_per_sample,
                     double* bytes_per_sample, double* bytes_per_launch) {
    if (!v) return fail(FC_E_ARG, "fc_vqvae_op_info: null handle");
    const Plan& pl = decode ? v->dec : v->enc;
    if (i < 0 || i >= (int)pl.ops.size()) return fail(FC_E_ARG, "fc_vqvae_op_info: index out of range");
    if (kernel) *kernel = pl.op_kernel[i].c_str();
    if (module) *module = pl.op_what[i].c_str();
    if (flops_per_sample) *flops_per_sample = pl.op_flops[i];
    if (bytes_per_sample) *bytes_per_sample = pl.op_bytes_ps[i];
    if (bytes_per_launch) *bytes_per_launch = pl.op_bytes_fixed[i];
    return FC_OK;
}

int fc_vqvae_profile_ops(fc_vqvae* v, int decode, const float* in_dev, float* out_dev, int batch, int repeats, float* ms_out, int n_out, void* stream) {
    if (!v || !in_dev || !out_dev || !ms_out || repeats < 1) return fail(FC_E_ARG, "fc_vqvae_profile_ops: bad argument");
    const Plan& pl = decode ? v->dec : v->enc;
    if (pl.maxB < batch || pl.ops.empty()) return fail(FC_E_STATE, "vqvae: reserve the plan first");
    if (!v->loaded) return fail(FC_E_STATE, "vqvae: weights not loaded (fc_vqvae_load_params)");
    FwdCtx c;
    c.x = in_dev; c.x_mod = batch; c.out = out_dev; c.B = batch;
    return profile_plan(pl, c, repeats, ms_out, n_out, static_cast<hipStream_t>(stream));
}

}  // extern "C"
