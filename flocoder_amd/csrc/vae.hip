// Native runtime of the SD-VAE codec (flocoder's SD_VAE_Wrapper, codecs.py:631-663 -> diffusers AutoencoderKL): parameter
// table with the upstream state_dict names, weight packing, and one launch plan each for encode and decode.
//
// The architecture is third-party (diffusers is not part of the reference tree; parity unpinned -- DESIGN.md 2).  Every
// Conv2d runs on the implicit-GEMM kernel; the pre-norm resnets map onto it directly:
//     norm1 -> SiLU -> conv1      = conv with GroupNorm(32, eps 1e-6)+SiLU applied by the loader to the raw input
//     norm2 -> SiLU -> conv2 (+x) = the same, with the shortcut added in the epilogue and the statistics for the NEXT
//                                   block's norm1 taken from the final sum (ConvArgs::stats_post)
// so no normalised / activated tensor is ever materialised.  Downsample2D's pad(0,1,0,1)+conv3x3/s2 is a stride-2 launch
// with pad 0 (reads beyond the bottom/right edge are the loader's zero fill); Upsample2D's nearest x2 is folded into the
// following conv's loader.  The single-head mid-block attention (n = h*w tokens, d = 512) is three 1x1 convs, a batched
// transpose, two per-sample-weight GEMMs on the same kernel (scores = q k^T, out = softmax v) and a one-pass row softmax.
#include <cstring>
#include <memory>

#include "plan.h"

using namespace fc;

struct fc_vae : fc::ParamStore {
    int device = 0;
    int in_ch = 3, latent = 4, lpb = 2, groups = 32;
    std::vector<int> bo{128, 256, 512, 512};
    fc::Plan enc, dec;
    int enc_B = 0, enc_H = 0, enc_W = 0, dec_B = 0, dec_h = 0, dec_w = 0;
    int prec = 0;     // fc_vae_set_precision: arithmetic of the plans built from now on (0 exact fp32, 1 split-bf16)
};

namespace fc {

static constexpr float kEps = 1e-6f;

static void decl_resnet(fc_vae* v, const std::string& n, int ci, int co) {
    v->decl_norm(n + ".norm1", ci);
    v->decl_conv(n + ".conv1", co, ci, 3);
    v->decl_norm(n + ".norm2", co);
    v->decl_conv(n + ".conv2", co, co, 3);
    if (ci != co) v->decl_conv(n + ".conv_shortcut", co, ci, 1);
}
static void decl_mid(fc_vae* v, const std::string& n, int c) {
    decl_resnet(v, n + ".resnets.0", c, c);
    const std::string a = n + ".attentions.0";
    v->decl_norm(a + ".group_norm", c);
    v->decl_linear_t(a + ".to_q", c, c);
    v->decl_linear_t(a + ".to_k", c, c);
    v->decl_linear_t(a + ".to_v", c, c);
    v->decl_linear_t(a + ".to_out.0", c, c);
    decl_resnet(v, n + ".resnets.1", c, c);
}
// Conv2d whose channel counts are not multiples of 4 (RGB in / out): packed with zero padding to 4
static void decl_conv_padded(fc_vae* v, const std::string& n, int O, int I, int K, int Opad, int Ipad) {
    v->declare(n + ".weight", {O, I, K, K});
    v->declare(n + ".bias", {O});
    const int64_t dst = v->pk_alloc(n + ".weight", (int64_t)Opad * Ipad * K * K);
    v->packops.push_back({4, v->params[v->pidx[n + ".weight"]].offset, dst, O, I, K * K, Opad, Ipad});
    if (Opad != O) {   // bias padded with zeros: copy O floats into a zero-initialised packed slot
        const int64_t bd = v->pk_alloc(n + ".bias", Opad);
        v->packops.push_back({3, v->params[v->pidx[n + ".bias"]].offset, bd, O, 0, 0, 0});
    }
}

static int declare_all(fc_vae* v) {
    const std::vector<int>& bo = v->bo;
    const int L = (int)bo.size();
    decl_conv_padded(v, "encoder.conv_in", bo[0], v->in_ch, 3, bo[0], 4);
    int ci = bo[0];
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < v->lpb; ++j) decl_resnet(v, "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? ci : bo[i], bo[i]);
        if (i < L - 1) v->decl_conv("encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", bo[i], bo[i], 3);
        ci = bo[i];
    }
    decl_mid(v, "encoder.mid_block", bo[L - 1]);
    v->decl_norm("encoder.conv_norm_out", bo[L - 1]);
    v->decl_conv("encoder.conv_out", 2 * v->latent, bo[L - 1], 3);
    v->decl_conv("quant_conv", 2 * v->latent, 2 * v->latent, 1);
    v->decl_conv("post_quant_conv", v->latent, v->latent, 1);
    v->decl_conv("decoder.conv_in", bo[L - 1], v->latent, 3);
    decl_mid(v, "decoder.mid_block", bo[L - 1]);
    ci = bo[L - 1];
    for (int i = 0; i < L; ++i) {
        const int co = bo[L - 1 - i];
        for (int j = 0; j < v->lpb + 1; ++j) decl_resnet(v, "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? ci : co, co);
        if (i < L - 1) v->decl_conv_up2("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", co, co);
        ci = co;
    }
    v->decl_norm("decoder.conv_norm_out", bo[0]);
    decl_conv_padded(v, "decoder.conv_out", v->in_ch, bo[0], 3, 4, bo[0]);
    return FC_OK;
}

struct VBuilder : PlanBuilder {
    fc_vae* v;
    VBuilder(fc_vae* v_, Plan* pl_, int B_) : v(v_) { pl = pl_; B = B_; store = v_; }

    SrcXform gn(const Stat& st, const std::string& norm, int mode) { return xf_of(st, mode, v->R(norm + ".weight"), v->R(norm + ".bias"), nullptr, 0, kEps); }

    // Pre-norm input of a resnet convolution.  Round 3: one elementwise launch writes act(GN(x)) and the convolution reads that.  Rounds 1-2
    // normalised in the convolution's staging waves and never materialised the tensor (FLOCODER_AMD_VAE_PRENORM=fused: still available), which
    // saves one round trip through HBM but transforms every window element once per output-channel tile and halo copy -- ~10x at 512
    // channels -- and VALU work of the staging waves does not overlap the MFMAs of the waves they share a SIMD with:
    // measured per resnet, fp32: 1.40 -> 1.29 + 0.03 ms at 512 channels, 6.32 -> 5.79 + 0.43 ms at 128; split-bf16: 0.55 -> 0.45 + 0.03, 2.89 -> 2.38 + 0.43.
    bool materialize_prenorm() const {
        static const bool fused = [] { const char* e = std::getenv("FLOCODER_AMD_VAE_PRENORM"); return e && std::string(e) == "fused"; }();
        return !fused;
    }
    // fills `src` for a convolution that reads x through `xf`; *tmp is the tensor to release after the convolution (or empty)
    void prenorm_src(ConvSrc& src, const Act& x, const SrcXform& xf, Act* tmp) {
        src.p = x.p; src.C = x.C; src.xf = xf;
        *tmp = Act();
        if (!materialize_prenorm() || err) return;
        Act y = act(x.C, x.H, x.W);
        FinalizeArgs f;
        f.h = x.p; f.xf = xf; f.y = y.p; f.HW = x.H * x.W; f.C = x.C;
        push([f](const FwdCtx& c, hipStream_t s) { FinalizeArgs g = f; g.B = c.B; return finalize_launch(g, s); }, "finalize");
        src.p = y.p; src.xf = SrcXform();
        *tmp = y;
    }

    // x (+ its GroupNorm(32) partials) -> resnet output (+ partials when the consumer starts with a norm)
    Act resnet(const std::string& n, const Act& x, const Stat& sx, int co, bool want_stats, Stat* so) {
        scope = n;
        const int G = v->groups;
        Act h1 = act(co, x.H, x.W);
        Stat s1;
        ConvArgs a;
        Act t1;
        prenorm_src(a.s0, x, gn(sx, n + ".norm1", 2), &t1);
        a.Hs = x.H; a.Ws = x.W; a.KS = 3; a.pad = 1;
        a.w = v->P(n + ".conv1.weight"); a.bias = v->R(n + ".conv1.bias");
        conv(a, h1, G, &s1);
        if (t1.p) release(t1);
        Act sc = x;
        const bool proj = x.C != co;
        if (proj) {
            sc = act(co, x.H, x.W);
            ConvArgs c;
            c.s0.p = x.p; c.s0.C = x.C; c.Hs = x.H; c.Ws = x.W; c.KS = 1;
            c.w = v->P(n + ".conv_shortcut.weight"); c.bias = v->R(n + ".conv_shortcut.bias");
            conv(c, sc, 0, nullptr);
        }
        Act out = act(co, x.H, x.W);
        ConvArgs b;
        Act t2;
        prenorm_src(b.s0, h1, gn(s1, n + ".norm2", 2), &t2);
        b.Hs = x.H; b.Ws = x.W; b.KS = 3; b.pad = 1;
        b.w = v->P(n + ".conv2.weight"); b.bias = v->R(n + ".conv2.bias");
        b.add = sc.p; b.stats_post = 1;
        conv(b, out, want_stats ? G : 0, so);
        if (t2.p) release(t2);
        release(h1);
        if (proj) release(sc);
        return out;
    }

    // mid-block attention: out = to_out(softmax(q k^T / sqrt(C)) v) + x, q/k/v = Linear(GroupNorm(x))
    Act attention(const std::string& n, const Act& x, const Stat& sx, Stat* so) {
        scope = n;
        AttnWeights w{v->P(n + ".to_q.weight"), v->R(n + ".to_q.bias"), v->P(n + ".to_k.weight"), v->R(n + ".to_k.bias"),
                      v->P(n + ".to_v.weight"), v->R(n + ".to_v.bias"), v->P(n + ".to_out.0.weight"), v->R(n + ".to_out.0.bias")};
        return attention_block(x, gn(sx, n + ".group_norm", 1), w, v->groups, so);
    }

    Act mid(const std::string& n, Act x, Stat sx, Stat* so) {
        Stat s1, s2;
        Act a = resnet(n + ".resnets.0", x, sx, x.C, true, &s1);
        release(x);
        Act b = attention(n + ".attentions.0", a, s1, &s2);
        release(a);
        Act c = resnet(n + ".resnets.1", b, s2, x.C, true, so);
        release(b);
        return c;
    }
};

static int build_encoder(fc_vae* v, int maxB, int H, int W) {
    v->enc.release();
    const int L = (int)v->bo.size();
    if (!is_pow2(H) || !is_pow2(W) || (H >> (L - 1)) < 1 || (W >> (L - 1)) < 1) return fail(FC_E_SHAPE, "vae: image height/width must be powers of two >= 8");
    VBuilder b(v, &v->enc, maxB);
    b.conv_prec = v->prec;
    const int G = v->groups, ic = v->in_ch;
    Act xin = b.act(4, H, W);
    float* xp = xin.p;
    b.scope = "input";
    b.push([=](const FwdCtx& c, hipStream_t s) { return nchw_to_nhwc_launch(c.x, xp, c.B, ic, H * W, 4, c.B, s); }, "nchw_to_nhwc");
    Stat st;
    Act x = b.act(v->bo[0], H, W);
    {
        b.scope = "encoder.conv_in";
        ConvArgs a;
        a.s0.p = xin.p; a.s0.C = 4; a.Hs = H; a.Ws = W; a.KS = 3; a.pad = 1;
        a.w = v->P("encoder.conv_in.weight"); a.bias = v->R("encoder.conv_in.bias");
        b.conv(a, x, G, &st);
    }
    for (int i = 0; i < L && !b.err; ++i) {
        const std::string blk = "encoder.down_blocks." + std::to_string(i);
        for (int j = 0; j < v->lpb && !b.err; ++j) {
            const bool last = j == v->lpb - 1, down = i < L - 1;
            Stat so;
            Act y = b.resnet(blk + ".resnets." + std::to_string(j), x, st, v->bo[i], !(last && down), &so);
            b.release(x);
            x = y; st = so;
        }
        if (i < L - 1 && !b.err) {   // Downsample2D: pad (0,1,0,1) + conv3x3 stride 2
            b.scope = blk + ".downsamplers.0";
            Act y = b.act(v->bo[i], x.H / 2, x.W / 2);
            ConvArgs a;
            a.s0.p = x.p; a.s0.C = x.C; a.Hs = x.H; a.Ws = x.W; a.KS = 3; a.pad = 0; a.stride = 2;
            a.w = v->P(blk + ".downsamplers.0.conv.weight"); a.bias = v->R(blk + ".downsamplers.0.conv.bias");
            b.conv(a, y, G, &st);
            b.release(x);
            x = y;
        }
    }
    if (b.err) return b.err;
    Stat sm;
    x = b.mid("encoder.mid_block", x, st, &sm);
    if (b.err) return b.err;
    Act mo = b.act(2 * v->latent, x.H, x.W), qo = b.act(2 * v->latent, x.H, x.W);
    {
        b.scope = "encoder.conv_out";
        ConvArgs a;
        a.s0.p = x.p; a.s0.C = x.C; a.s0.xf = b.gn(sm, "encoder.conv_norm_out", 2);
        a.Hs = x.H; a.Ws = x.W; a.KS = 3; a.pad = 1;
        a.w = v->P("encoder.conv_out.weight"); a.bias = v->R("encoder.conv_out.bias");
        b.conv(a, mo, 0, nullptr);
        b.scope = "quant_conv";
        ConvArgs q;
        q.s0.p = mo.p; q.s0.C = mo.C; q.Hs = x.H; q.Ws = x.W; q.KS = 1;
        q.w = v->P("quant_conv.weight"); q.bias = v->R("quant_conv.bias");
        b.conv(q, qo, 0, nullptr);
    }
    if (b.err) return b.err;
    const float* qp = qo.p;
    const int lat = v->latent, hw = x.H * x.W;
    b.scope = "latent_dist.mean";
    b.push([=](const FwdCtx& c, hipStream_t s) { return nhwc_to_nchw_launch(qp, c.out, c.B, lat, hw, 2 * lat, s); }, "nhwc_to_nchw");
    v->enc.maxB = maxB; v->enc.H = H; v->enc.W = W;
    return FC_OK;
}

static int build_decoder(fc_vae* v, int maxB, int h, int w) {
    v->dec.release();
    const int L = (int)v->bo.size();
    if (!is_pow2(h) || !is_pow2(w)) return fail(FC_E_SHAPE, "vae: latent height/width must be powers of two");
    VBuilder b(v, &v->dec, maxB);
    b.conv_prec = v->prec;
    const int G = v->groups, lat = v->latent, top = v->bo[L - 1];
    Act zin = b.act(lat, h, w), z1 = b.act(lat, h, w);
    float* zp = zin.p;
    b.scope = "input";
    b.push([=](const FwdCtx& c, hipStream_t s) { return nchw_to_nhwc_launch(c.x, zp, c.B, lat, h * w, lat, c.B, s); }, "nchw_to_nhwc");
    Stat st;
    Act x = b.act(top, h, w);
    {
        b.scope = "post_quant_conv";
        ConvArgs q;
        q.s0.p = zin.p; q.s0.C = lat; q.Hs = h; q.Ws = w; q.KS = 1;
        q.w = v->P("post_quant_conv.weight"); q.bias = v->R("post_quant_conv.bias");
        b.conv(q, z1, 0, nullptr);
        b.scope = "decoder.conv_in";
        ConvArgs a;
        a.s0.p = z1.p; a.s0.C = lat; a.Hs = h; a.Ws = w; a.KS = 3; a.pad = 1;
        a.w = v->P("decoder.conv_in.weight"); a.bias = v->R("decoder.conv_in.bias");
        b.conv(a, x, G, &st);
    }
    if (b.err) return b.err;
    Stat sm;
    x = b.mid("decoder.mid_block", x, st, &sm);
    st = sm;
    for (int i = 0; i < L && !b.err; ++i) {
        const std::string blk = "decoder.up_blocks." + std::to_string(i);
        const int co = v->bo[L - 1 - i];
        for (int j = 0; j < v->lpb + 1 && !b.err; ++j) {
            const bool last = j == v->lpb, up = i < L - 1;
            Stat so;
            Act y = b.resnet(blk + ".resnets." + std::to_string(j), x, st, co, !(last && up), &so);
            b.release(x);
            x = y; st = so;
        }
        if (i < L - 1 && !b.err) {   // Upsample2D: nearest x2 + conv3x3
            b.scope = blk + ".upsamplers.0";
            Act y = b.act(co, x.H * 2, x.W * 2);
            // (round 3) the upsampling is folded into the weights: four 2x2 convolutions on x, 4/9 of the multiply-adds (plan.h conv_up2);
            // FLOCODER_AMD_UPS_FOLD=0: the 3x3 convolution over the upsampled window as in rounds 1-2
            ConvSrc src; src.p = x.p; src.C = x.C;
            if (!b.conv_up2(src, x, v->PUP(blk + ".upsamplers.0.conv.weight"), v->R(blk + ".upsamplers.0.conv.bias"), y, G, &st)) {
                ConvArgs a;
                a.s0.p = x.p; a.s0.C = x.C; a.Hs = x.H; a.Ws = x.W; a.KS = 3; a.pad = 1; a.ups = 1;
                a.w = v->P(blk + ".upsamplers.0.conv.weight"); a.bias = v->R(blk + ".upsamplers.0.conv.bias");
                b.conv(a, y, G, &st);
            }
            b.release(x);
            x = y;
        }
    }
    if (b.err) return b.err;
    Act yo = b.act(4, x.H, x.W);
    {
        b.scope = "decoder.conv_out";
        ConvArgs a;
        a.s0.p = x.p; a.s0.C = x.C; a.s0.xf = b.gn(st, "decoder.conv_norm_out", 2);
        a.Hs = x.H; a.Ws = x.W; a.KS = 3; a.pad = 1;
        a.w = v->P("decoder.conv_out.weight"); a.bias = v->P("decoder.conv_out.bias");
        b.conv(a, yo, 0, nullptr);
    }
    if (b.err) return b.err;
    const float* yp = yo.p;
    const int ic = v->in_ch, HW = x.H * x.W;
    b.scope = "sample";
    b.push([=](const FwdCtx& c, hipStream_t s) { return nhwc_to_nchw_launch(yp, c.out, c.B, ic, HW, 4, s); }, "nhwc_to_nchw");
    v->dec.maxB = maxB; v->dec.H = h; v->dec.W = w;
    return FC_OK;
}

}  // namespace fc

extern "C" {

int fc_vae_create(int device, fc_vae** out) {
    if (!out) return fail(FC_E_ARG, "fc_vae_create: null argument");
    std::unique_ptr<fc_vae> v(new fc_vae);
    v->device = device;
    v->want_b3 = device >= 0;      // the split-bf16 copies of the conv weights (set_precision): +1x the conv weights in HBM
    FC_TRY(declare_all(v.get()));
    if (device < 0) { *out = v.release(); return FC_OK; }
    FC_TRY(fc_check_device(device));
    FC_HIP(hipSetDevice(device));
    FC_TRY(conv_init());
    FC_TRY(v->alloc_device());
    FC_HIP(hipMemset(v->packed, 0, (size_t)v->packed_numel * sizeof(float)));   // padded bias slots stay zero beyond the copy
    *out = v.release();
    return FC_OK;
}

void fc_vae_destroy(fc_vae* v) {
    if (!v) return;
    if (v->device >= 0) {
        (void)hipSetDevice(v->device);
        (void)hipDeviceSynchronize();
        v->enc.release();
        v->dec.release();
        v->free_device();
    }
    delete v;
}

int fc_vae_param_count(const fc_vae* v) { return v ? (int)v->params.size() : 0; }
int64_t fc_vae_param_numel(const fc_vae* v) { return v ? v->raw_numel : 0; }
int fc_vae_param_info(const fc_vae* v, int i, const char** name, int64_t shape[4], int64_t* offset) {
    if (!v) return fail(FC_E_ARG, "fc_vae_param_info: null handle");
    return v->info(i, name, shape, offset);
}
int fc_vae_load_params(fc_vae* v, const float* flat, int64_t numel, int on_device, void* stream) {
    if (!v || !flat) return fail(FC_E_ARG, "fc_vae_load_params: null argument");
    if (v->device < 0) return fail(FC_E_STATE, "vae: created with device < 0 (description only)");
    FC_HIP(hipSetDevice(v->device));
    return v->load(flat, numel, on_device, static_cast<hipStream_t>(stream));
}

int fc_vae_reserve_encode(fc_vae* v, int max_batch, int height, int width) {
    if (!v || max_batch < 1 || v->device < 0) return fail(FC_E_ARG, "fc_vae_reserve_encode: bad argument");
    if (v->enc.maxB >= max_batch && v->enc.H == height && v->enc.W == width) return FC_OK;
    FC_HIP(hipSetDevice(v->device));
    FC_HIP(hipDeviceSynchronize());
    const int r = build_encoder(v, max_batch, height, width);
    if (r != FC_OK) v->enc.release();
    return r;
}
int fc_vae_set_precision(fc_vae* v, int mode) {
    if (!v || (mode != 0 && mode != 1)) return fail(FC_E_ARG, "fc_vae_set_precision: mode is 0 (fp32) or 1 (split-bf16)");
    if (v->prec == mode) return FC_OK;
    v->prec = mode;
    if (v->device >= 0) {          // plans in place were built for the other arithmetic: drop them, the next reserve rebuilds
        FC_HIP(hipSetDevice(v->device));
        FC_HIP(hipDeviceSynchronize());
        v->enc.release(); v->dec.release();
    }
    return FC_OK;
}

int fc_vae_reserve_decode(fc_vae* v, int max_batch, int lat_height, int lat_width) {
    if (!v || max_batch < 1 || v->device < 0) return fail(FC_E_ARG, "fc_vae_reserve_decode: bad argument");
    if (v->dec.maxB >= max_batch && v->dec.H == lat_height && v->dec.W == lat_width) return FC_OK;
    FC_HIP(hipSetDevice(v->device));
    FC_HIP(hipDeviceSynchronize());
    const int r = build_decoder(v, max_batch, lat_height, lat_width);
    if (r != FC_OK) v->dec.release();
    return r;
}

static int run_vae(const fc_vae* v, bool decode, const float* in, float* out, int B, int H, int W, void* stream) {
    if (!v || !in || !out || B < 1) return fail(FC_E_ARG, "vae: null argument");
    const Plan& pl = decode ? v->dec : v->enc;
    if (!v->loaded) return fail(FC_E_STATE, "vae: weights not loaded (fc_vae_load_params)");
    if (pl.maxB < B || pl.H != H || pl.W != W) return fail(FC_E_STATE, "vae: no plan for this shape; call fc_vae_reserve_* first");
    FwdCtx c;
    c.x = in; c.x_mod = B; c.out = out; c.B = B;
    return run_plan(pl, c, static_cast<hipStream_t>(stream));
}
int fc_vae_encode(fc_vae* v, const float* x_dev, float* mean_out_dev, int batch, int height, int width, void* stream) {
    return run_vae(v, false, x_dev, mean_out_dev, batch, height, width, stream);
}
int fc_vae_decode(fc_vae* v, const float* z_dev, float* x_out_dev, int batch, int lat_height, int lat_width, void* stream) {
    return run_vae(v, true, z_dev, x_out_dev, batch, lat_height, lat_width, stream);
}
double fc_vae_flops_per_sample(const fc_vae* v, int decode) { return v ? (decode ? v->dec.flops : v->enc.flops) : 0.0; }
int fc_vae_plan_launches(const fc_vae* v, int decode) { return v ? (int)(decode ? v->dec.ops.size() : v->enc.ops.size()) : 0; }

int fc_vae_op_info(const fc_vae* v, int decode, int i, const char** kernel, const char** module, double* flops_per_sample) {
    if (!v) return fail(FC_E_ARG, "fc_vae_op_info: null handle");
    const Plan& pl = decode ? v->dec : v->enc;
    if (i < 0 || i >= (int)pl.ops.size()) return fail(FC_E_ARG, "fc_vae_op_info: index out of range");
    if (kernel) *kernel = pl.op_kernel[i].c_str();
    if (module) *module = pl.op_what[i].c_str();
    if (flops_per_sample) *flops_per_sample = pl.op_flops[i];
    return FC_OK;
}

int fc_vae_op_bytes(const fc_vae* v, int decode, int i, double* bytes_per_sample, double* bytes_per_launch) {
    if (!v) return fail(FC_E_ARG, "fc_vae_op_bytes: null handle");
    const Plan& pl = decode ? v->dec : v->enc;
    if (i < 0 || i >= (int)pl.ops.size()) return fail(FC_E_ARG, "fc_vae_op_bytes: index out of range");
    if (bytes_per_sample) *bytes_per_sample = pl.op_bytes_ps[i];
    if (bytes_per_launch) *bytes_per_launch = pl.op_bytes_fixed[i];
    return FC_OK;
}

// Measurement hook: every launch of the encode / decode plan timed alone (`repeats` back-to-back launches between two events).
// in_dev / out_dev: valid input and output tensors for `batch` samples at the plan's shape.  Synchronises.
int fc_vae_profile_ops(fc_vae* v, int decode, const float* in_dev, float* out_dev, int batch, int repeats, float* ms_out, int n_out, void* stream) {
    if (!v || !in_dev || !out_dev || !ms_out || repeats < 1) return fail(FC_E_ARG, "fc_vae_profile_ops: bad argument");
    const Plan& pl = decode ? v->dec : v->enc;
    const int n = (int)pl.ops.size();
    if (pl.maxB < batch || n == 0) return fail(FC_E_STATE, "vae: reserve the plan first");
    if (n_out < n) return fail(FC_E_ARG, "fc_vae_profile_ops: output array too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    FwdCtx c;
    c.x = in_dev; c.x_mod = batch; c.out = out_dev; c.B = batch;
    FC_TRY(run_plan(pl, c, s));
    std::vector<hipEvent_t> ev(2 * n);
    for (auto& e : ev) FC_HIP(hipEventCreate(&e));
    int rc = FC_OK;
    for (int i = 0; i < n && rc == FC_OK; ++i) {
        (void)hipEventRecord(ev[2 * i], s);
        for (int r = 0; r < repeats && rc == FC_OK; ++r) rc = pl.ops[i](c, s);
        (void)hipEventRecord(ev[2 * i + 1], s);
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

}  // extern "C"
