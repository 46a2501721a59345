// Backward pass of the velocity U-Net for the flow training step (train_flow.py:358-371: loss.backward() through
// Unet._forward, unet.py:289-372), as a second static launch plan over the arena the forward plan already keeps.
//
// The forward stores raw convolution outputs and GroupNorm (mean, M2) partials only; the backward recomputes normalised /
// activated tensors where a weight gradient needs them (a finalize pass) and inside the GroupNorm-backward kernels.  Per module,
// in reverse order of the forward tape:
//   ResnetBlock      GN2+SiLU bwd | recompute a1 | wgrad conv2 | dgrad conv2 | GN1+FiLM+SiLU bwd | wgrad conv1 | dgrad conv1 (one
//                    launch per concat source) | res_conv wgrad + dgrad, or the identity residual
//   LinearAttention  GN bwd | wgrad/dgrad to_out | attention core bwd | recompute GN(x) | wgrad/dgrad to_qkv | GN bwd | residual
//   Attention        wgrad/dgrad to_out | core bwd | recompute GN(x) | wgrad/dgrad to_qkv | GN bwd | residual
//   Down/Upsample    wgrad (strided / nearest-x2 loader) | dgrad (+ depth-to-space / 2x2 sum)
// then the conditioning path: FiLM projections, time MLP, class-embedding MLP.  Data gradients reuse the forward implicit-GEMM
// kernel on flipped, transposed weights (re-packed whenever the parameters change); a tensor with several consumers gets its
// gradient contributions through the kernels' accumulate paths in a fixed order.  Parameter gradients land in the caller's flat
// vector in the parameter table's layout, each written exactly once.
#include <cstdlib>
#include <memory>
#include <set>

#include "plan.h"
#include "unet_priv.h"

using namespace fc;

namespace fc {

struct BwdBuilder : PlanBuilder {
    fc_unet* u;
    const Plan* fw;
    struct Slot { Act g; bool written = false; };
    std::map<const float*, Slot> gmap;
    float *ws = nullptr, *s12 = nullptr, *s12p = nullptr, *dss = nullptr;
    size_t ws_floats = 0;
    BwdBuilder(fc_unet* u_, const Plan* fw_, Plan* pl_, int B_) : u(u_), fw(fw_) { pl = pl_; B = B_; }

    int64_t off(const std::string& n) const { return u->params[u->pidx.at(n)].offset; }
    Slot& slot(const Act& a) {
        auto it = gmap.find(a.p);
        if (it == gmap.end()) {
            Slot s;
            s.g.C = a.C; s.g.H = a.H; s.g.W = a.W;
            s.g.p = dmalloc((size_t)B * a.H * a.W * a.C);
            it = gmap.emplace(a.p, s).first;
        }
        return it->second;
    }
    const Act& grad_of(const Act& a) {
        Slot& s = slot(a);
        if (!s.written) err = fail(FC_E_STATE, "backward: gradient consumed before it was produced (" + scope + ")");
        return s.g;
    }

    std::vector<WredJob> wred_jobs;         // deferred split reductions of the weight gradients: one table-driven launch at the end
    // Weight gradients themselves are deferred too (full-batch steps): every layer's launch is recorded as a table entry and all
    // entries of one kernel size run as ONE launch after the data-gradient chain.  Their operands (the layer input, possibly a
    // recomputed activation, and the output gradient) therefore stay alive to the end: release() leaves pinned buffers alone.
    bool batch_wgrad = std::getenv("FLOCODER_AMD_WGRAD_EACH") == nullptr && std::getenv("FLOCODER_AMD_WGRAD_REDUCE_EACH") == nullptr;
    std::set<const float*> pinned;
    struct WgradClass { std::vector<WgradDev> jobs; std::vector<int2> blocks; size_t lds = 0; };
    std::map<int, WgradClass> wclasses;     // by kernel size
    void release(const Act& a) { if (!pinned.count(a.p)) PlanBuilder::release(a); }
    void wgrad(const std::string& wname, const std::string& bname, const Act& x, const Act* skip, const Act& dy, int KS, int pad, int stride, int ups) {
        if (err) return;
        WgradArgs a;
        a.x0 = x.p; a.C0 = x.C;
        if (skip && skip->p) { a.x1 = skip->p; a.C1 = skip->C; }
        a.dy = dy.p; a.H = dy.H; a.W = dy.W; a.Hs = x.H; a.Ws = x.W; a.Cin = a.C0 + a.C1; a.Cout = dy.C;
        a.KS = KS; a.pad = pad; a.stride = stride; a.ups = ups; a.B = B;
        a.ws = ws; a.ws_floats = ws_floats;
        const int64_t wo = off(wname), bo = bname.empty() ? -1 : off(bname);
        // at the plan's full batch the launch keeps its partials in a workspace of its own and leaves the summation to the table launch
        int ns = 1; size_t stride_f = 0;
        static const bool no_defer = std::getenv("FLOCODER_AMD_WGRAD_REDUCE_EACH") != nullptr;
        float* own = nullptr;
        // a table launch runs ~70 layers side by side: a quarter of the stand-alone split fills the chip, with a quarter of the partials
        static const int table_target = [] { const char* e = std::getenv("FLOCODER_AMD_WGRAD_TABLE_SPLIT"); return e ? std::atoi(e) : 256; }();
        const bool to_table = batch_wgrad && !no_defer && guard == 0;
        if (to_table) a.split_target = table_target;
        if (!no_defer && guard == 0 && conv_wgrad_split(a, &ns, &stride_f) == FC_OK && ns > 1) {   // guarded (mask-branch) launches may not run: they reduce on the spot
            own = dmalloc((size_t)ns * stride_f);
            if (err) return;
            wred_jobs.push_back({own, ns, a.Cout, stride_f, (size_t)a.Cout * a.Cin * KS * KS, wo, bo, a.Cin, KS * KS});
        }
        const int maxB = B;
        const size_t own_floats = (size_t)ns * stride_f;
        bool deferred = false;
        if (to_table) {
            WgradArgs e = a;
            e.ws = own; e.ws_floats = 0;
            WgradDev d;
            int nb = 0; size_t lds = 0;
            if (conv_wgrad_table_entry(e, wo, bo, &d, &nb, &lds) == FC_OK && d.nsplit == ns) {
                WgradClass& wc = wclasses[KS];
                const int j = (int)wc.jobs.size();
                wc.jobs.push_back(d);
                for (int k = 0; k < nb; ++k) wc.blocks.push_back(make_int2(j, k));
                if (lds > wc.lds) wc.lds = lds;
                pinned.insert(x.p); pinned.insert(dy.p);
                if (skip && skip->p) pinned.insert(skip->p);
                deferred = true;
            }
        }
        // An activation recomputed only for this weight gradient (materialize(..., for_wgrad)) leaves the chain exactly when this entry
        // did: decided HERE, once, for both (round 2 decided it twice from conditions that merely happened to coincide -- an entry the
        // table refused would have read its input before the end-of-plan finalize launch had produced it).
        if (pending_mat.flag && pending_mat.y == x.p) {
            if (deferred) { *pending_mat.flag = true; fin_jobs.push_back(pending_mat.f); pinned.insert(pending_mat.y); }
            pending_mat = PendingMat();
        }
        push([a, wo, bo, own, own_floats, maxB, deferred](const FwdCtx& c, hipStream_t s) -> int {
            if (deferred && c.B == maxB) return FC_OK;      // runs in the table launch at the end of the plan
            WgradArgs b = a;
            if (!(own && c.B == maxB)) b.split_target = 1024;
            b.B = c.B; b.dw = c.grads + wo; b.db = bo >= 0 ? c.grads + bo : nullptr;
            if (own && c.B == maxB) { b.ws = own; b.ws_floats = own_floats; return conv_wgrad_launch_noreduce(b, s); }
            return conv_wgrad_launch(b, s);
        }, "conv_wgrad", 2.0 * dy.H * dy.W * KS * KS * (double)a.Cin * a.Cout);
    }

    // d(target) (+)= conv_transpose(dy, W[:, ci0:ci0+target.C]) for a stride-1 convolution of kernel KS / padding pad
    void dgrad(const std::string& wname, int O, int I, int KS, int pad, const Act& dy, int ci0, const Act& target, const float* plus = nullptr) {
        if (err) return;
        Slot& t = slot(target);
        dgrad_to(wname, O, I, KS, pad, dy, ci0, t.g, t.written ? t.g.p : plus);
        if (t.written && plus) err = fail(FC_E_STATE, "backward: dgrad with both an accumulate and an extra addend");
        t.written = true;
    }
    void dgrad_to(const std::string& wname, int O, int I, int KS, int pad, const Act& dy, int ci0, const Act& out, const float* add) {
        if (err) return;
        const int nci = out.C;
        float* wp = dmalloc((size_t)O * nci * KS * KS);
        u->dgrad_packs.push_back({off(wname), wp, O, I, KS, ci0, nci});
        ConvArgs a;
        a.s0.p = dy.p; a.s0.C = O; a.Hs = dy.H; a.Ws = dy.W; a.KS = KS; a.pad = KS - 1 - pad;
        a.w = wp; a.add = add;
        conv(a, out, 0, nullptr);
    }
    void accumulate(const Act& target, const Act& src) {   // d(target) (+)= src
        if (err) return;
        Slot& t = slot(target);
        float* dp = t.g.p; const float* sp = src.p;
        const size_t per = (size_t)target.H * target.W * target.C;
        if (t.written) push([=](const FwdCtx& c, hipStream_t s) { return add_into_launch(dp, sp, per * c.B, s); }, "add_into");
        else push([=](const FwdCtx& c, hipStream_t s) -> int { FC_HIP(hipMemcpyAsync(dp, sp, per * c.B * sizeof(float), hipMemcpyDeviceToDevice, s)); return FC_OK; }, "copy");
        t.written = true;
    }
    // GroupNorm(+FiLM)(+SiLU) backward of y = f(h): dh (+)= ..., parameter gradients of the norm, FiLM gradients into dss
    void gn_bwd(const Act& dy, const Act& h, const SrcXform& xf, float* dh, bool acc, const std::string& norm, int ss_col, const float* plus = nullptr) {
        if (err) return;
        GnBwdArgs g;
        g.plus = plus;
        float* s12l = dmalloc((size_t)B * h.C * 2);      // kept until the batched parameter-gradient launch at the end
        g.dy = dy.p; g.h = h.p; g.xf = xf; g.s12 = s12l; g.s12p = s12p; g.dh = dh; g.accumulate = acc ? 1 : 0; g.HW = h.H * h.W; g.C = h.C;
        push([g](const FwdCtx& c, hipStream_t s) { GnBwdArgs k = g; k.B = c.B; return gn_bwd_launch(k, s); }, "gn_bwd");
        norm_jobs.push_back({s12l, xf.gamma, xf.beta, xf.ss, off(norm + ".weight"), off(norm + ".bias"), ss_col, h.C});
    }
    std::vector<NormJob> norm_jobs;
    std::vector<std::string> film_blocks;   // ResnetBlocks whose backward has been emitted since the deferred tables last ran (their dss columns)
    // y = act(gn(h)).  `for_wgrad`: the only reader is a deferred weight-gradient entry, so at the full batch the pass joins the
    // table launch in front of the weight gradients instead of sitting on the data-gradient chain
    std::vector<FinalizeArgs> fin_jobs;
    struct PendingMat { FinalizeArgs f; std::shared_ptr<bool> flag; const float* y = nullptr; };
    PendingMat pending_mat;                 // a for_wgrad activation whose reader (the next wgrad() of y) has not been emitted yet
    void materialize(const Act& h, const SrcXform& xf, const Act& y, bool for_wgrad = false) {
        if (err) return;
        if (pending_mat.flag) { err = fail(FC_E_STATE, "backward: an activation materialised for a weight gradient was never read by one (" + scope + ")"); return; }
        FinalizeArgs f;
        f.h = h.p; f.xf = xf; f.y = y.p; f.HW = h.H * h.W; f.C = h.C;
        auto deferred = std::make_shared<bool>(false);      // set by the wgrad() that reads y, if and only if that entry joins the table launch
        if (for_wgrad) { pending_mat.f = f; pending_mat.f.B = B; pending_mat.flag = deferred; pending_mat.y = y.p; }
        const int maxB = B;
        push([f, deferred, maxB](const FwdCtx& c, hipStream_t s) -> int {
            if (*deferred && c.B == maxB) return FC_OK;
            FinalizeArgs g = f; g.B = c.B; return finalize_launch(g, s);
        }, "finalize");
    }

    void resblock(const ResRec& r) {
        scope = r.p;
        film_blocks.push_back(r.p);
        const std::string& p = r.p;
        const int cout = r.cout, cin = r.x.C + r.skip.C, H = r.x.H, W = r.x.W;
        const Act g_out = grad_of(r.out);
        if (err) return;
        const SrcXform xf2 = xf_of(r.st2, 2, u->R(p + ".block2.norm.weight"), u->R(p + ".block2.norm.bias"));
        const SrcXform xf1 = xf_of(r.st1, 2, u->R(p + ".block1.norm.weight"), u->R(p + ".block1.norm.bias"), fw->ss + u->ss_off.at(p), u->S);
        Act dh2 = act(cout, H, W), a1 = act(cout, H, W), da1 = act(cout, H, W);
        gn_bwd(g_out, r.h2, xf2, dh2.p, false, p + ".block2.norm", 0);
        materialize(r.h1, xf1, a1, true);
        wgrad(p + ".block2.proj.weight", p + ".block2.proj.bias", a1, nullptr, dh2, 3, 1, 1, 0);
        dgrad_to(p + ".block2.proj.weight", cout, cout, 3, 1, dh2, 0, da1, nullptr);
        release(a1);
        Act dh1 = pinned.count(dh2.p) ? act(cout, H, W) : dh2;   // dh2 is dead once conv2's gradients are out (unless a deferred launch still reads it)
        gn_bwd(da1, r.h1, xf1, dh1.p, false, p + ".block1.norm", u->ss_off.at(p));
        release(da1);
        const Act* sk = r.skip.p ? &r.skip : nullptr;
        wgrad(p + ".block1.proj.weight", p + ".block1.proj.bias", r.x, sk, dh1, 3, 1, 1, 0);
        const bool ident = cin == cout;
        // identity residual: d(x) = dgrad(conv1) + d(out), folded into the dgrad epilogue when d(x) has no earlier contribution
        const bool fold = ident && !slot(r.x).written;
        dgrad(p + ".block1.proj.weight", cout, cin, 3, 1, dh1, 0, r.x, fold ? g_out.p : nullptr);
        if (sk) dgrad(p + ".block1.proj.weight", cout, cin, 3, 1, dh1, r.x.C, r.skip);
        release(dh1);
        if (ident) {
            if (!fold) accumulate(r.x, g_out);
        } else {
            wgrad(p + ".res_conv.weight", p + ".res_conv.bias", r.x, sk, g_out, 1, 0, 1, 0);
            dgrad(p + ".res_conv.weight", cout, cin, 1, 0, g_out, 0, r.x);
            if (sk) dgrad(p + ".res_conv.weight", cout, cin, 1, 0, g_out, r.x.C, r.skip);
        }
    }

    // shared tail of both attention blocks: d(qkv) -> to_qkv gradients -> PreNorm GroupNorm(1) backward into d(x), plus the residual
    void qkv_tail(const std::string& wq, const std::string& norm, const Act& x, const Stat& gn1, const Act& dqkv, const Act& g_out) {
        const SrcXform xf = xf_of(gn1, 1, u->R(norm + ".weight"), u->R(norm + ".bias"));
        Act xn = act(x.C, x.H, x.W), dxn = act(x.C, x.H, x.W);
        materialize(x, xf, xn, true);
        wgrad(wq, "", xn, nullptr, dqkv, 1, 0, 1, 0);
        dgrad_to(wq, dqkv.C, x.C, 1, 0, dqkv, 0, dxn, nullptr);
        release(xn);
        Slot& t = slot(x);
        gn_bwd(dxn, x, xf, t.g.p, t.written, norm, 0, g_out.p);    // + the residual branch: d(x) (+)= GN'(dxn) + d(out)
        t.written = true;
        release(dxn);
    }

    void linattn(const LinRec& r) {
        scope = r.p;
        const std::string& p = r.p;
        const int hid = u->heads * 32, n = r.x.H * r.x.W, heads = u->heads, H = r.x.H, W = r.x.W;
        const Act g_out = grad_of(r.out);
        if (err) return;
        Act dyb = act(r.x.C, H, W), dlao = act(hid, H, W), dqkv = act(3 * hid, H, W);
        gn_bwd(g_out, r.yb, xf_of(r.sty, 1, u->R(p + ".fn.fn.to_out.1.weight"), u->R(p + ".fn.fn.to_out.1.bias")), dyb.p, false, p + ".fn.fn.to_out.1", 0);
        wgrad(p + ".fn.fn.to_out.0.weight", p + ".fn.fn.to_out.0.bias", r.lao, nullptr, dyb, 1, 0, 1, 0);
        dgrad_to(p + ".fn.fn.to_out.0.weight", r.x.C, hid, 1, 0, dyb, 0, dlao, nullptr);
        release(dyb);
        float* dctx = dmalloc((size_t)B * heads * 32 * 32);
        float* kst = dmalloc((size_t)B * heads * 32 * 2);
        float* rr = dmalloc((size_t)B * heads * 32);
        const float *qp = r.qkv.p, *dl = dlao.p, *cx = r.ctx;
        float* dq = dqkv.p;
        if (!err) push([=](const FwdCtx& c, hipStream_t s) { return linattn_bwd_launch(qp, dl, cx, dctx, kst, rr, dq, c.B, n, heads, s); }, "linattn_bwd",
                       2.0 * 4 * n * 32 * 32 * heads);
        release(dlao);
        qkv_tail(p + ".fn.fn.to_qkv.weight", p + ".fn.norm", r.x, r.gn1, dqkv, g_out);
        release(dqkv);
    }

    void midattn(const MidRec& r) {
        scope = "mid_attn";
        const int hid = u->heads * 32, n = r.x.H * r.x.W, heads = u->heads, H = r.x.H, W = r.x.W;
        const Act g_out = grad_of(r.out);
        if (err) return;
        Act dao = act(hid, H, W), dqkv = act(3 * hid, H, W);
        wgrad("mid_attn.fn.fn.to_out.weight", "mid_attn.fn.fn.to_out.bias", r.ao, nullptr, g_out, 1, 0, 1, 0);
        dgrad_to("mid_attn.fn.fn.to_out.weight", r.x.C, hid, 1, 0, g_out, 0, dao, nullptr);
        const float *qp = r.qkv.p, *dp = dao.p;
        float* dq = dqkv.p;
        if (!err) push([=](const FwdCtx& c, hipStream_t s) { return attn_small_bwd_launch(qp, dp, dq, c.B, n, heads, s); }, "attn_small_bwd");
        release(dao);
        qkv_tail("mid_attn.fn.fn.to_qkv.weight", "mid_attn.fn.norm", r.x, r.gn1, dqkv, g_out);
        release(dqkv);
    }

    void resample(const ConvRec& r) {
        scope = r.name;
        const Act g_out = grad_of(r.out);
        if (err) return;
        wgrad(r.name + ".weight", r.name + ".bias", r.x, nullptr, g_out, r.KS, r.pad, r.stride, r.ups);
        if (r.stride == 2) {            // Downsample: 1x1 over the space-to-depth channels, then depth-to-space
            Act t4 = act(4 * r.x.C, r.out.H, r.out.W);
            dgrad_to(r.name + ".weight", r.out.C, 4 * r.x.C, 1, 0, g_out, 0, t4, nullptr);
            Slot& t = slot(r.x);
            const float* sp = t4.p; float* dp = t.g.p;
            const int h = r.out.H, w = r.out.W, C = r.x.C, acc = t.written ? 1 : 0;
            if (!err) push([=](const FwdCtx& c, hipStream_t s) { return depth_to_space_launch(sp, dp, c.B, h, w, C, acc, s); }, "depth_to_space");
            t.written = true;
            release(t4);
        } else if (r.ups) {             // nearest x2: gradient at the fine resolution, summed over 2x2 blocks
            Act fine = act(r.x.C, r.out.H, r.out.W);
            dgrad_to(r.name + ".weight", r.out.C, r.x.C, r.KS, r.pad, g_out, 0, fine, nullptr);
            Slot& t = slot(r.x);
            const float* sp = fine.p; float* dp = t.g.p;
            const int h = r.x.H, w = r.x.W, C = r.x.C, acc = t.written ? 1 : 0;
            if (!err) push([=](const FwdCtx& c, hipStream_t s) { return sumpool2_nhwc_launch(sp, dp, c.B, h, w, C, acc, s); }, "sumpool2");
            t.written = true;
            release(fine);
        } else {
            dgrad(r.name + ".weight", r.out.C, r.x.C, r.KS, r.pad, g_out, 0, r.x);
        }
    }

    // ---- mask conditioning (unet.py:298-305,336-340,360-364).  The forward decides per call whether these branches run (a mask
    // given / not all ones); the backward's launches carry the same guards, and both variants leave d(x) written.
    Act gmask;                 // d(mask) NHWC, zeroed at the start of every backward and accumulated into

    void silu_bwd(const Act& dy, const Act& z, const Act& dz) {
        const float *dp = dy.p, *zp = z.p; float* op = dz.p;
        const size_t per = (size_t)z.H * z.W * z.C;
        push([=](const FwdCtx& c, hipStream_t s) { return silu_bwd_launch(dp, zp, op, per * c.B, s); }, "silu_bwd");
    }

    void inject(const InjRec& r) {
        scope = r.name;
        const Act g_out = grad_of(r.out);
        if (err) return;
        const bool x_written = slot(r.x).written;
        const int C = r.x.C, H = r.x.H, W = r.x.W;
        // -- a mask was given: out = x + SiLU(z), z = conv3x3(cat[x, bilinear(mask)]) --
        guard = 1;
        Act dz = act(C, H, W);
        silu_bwd(g_out, r.z, dz);
        wgrad(r.name + ".weight", r.name + ".bias", r.x, &r.mr, dz, 3, 1, 1, 0);
        dgrad(r.name + ".weight", C, C + r.mr.C, 3, 1, dz, 0, r.x);
        Act gmr = act(r.mr.C, H, W);
        dgrad_to(r.name + ".weight", C, C + r.mr.C, 3, 1, dz, C, gmr, nullptr);
        accumulate(r.x, g_out);
        {
            const float* gp = gmr.p; float* mp = gmask.p;
            const int mc = r.mr.C, Hs = gmask.H, Ws = gmask.W;
            push([=](const FwdCtx& c, hipStream_t s) { return bilinear_bwd_launch(gp, mp, c.B, mc, Hs, Ws, H, W, s); }, "bilinear_bwd");
        }
        release(dz); release(gmr);
        // -- no mask: out = x --
        guard = 3;
        slot(r.x).written = x_written;
        accumulate(r.x, g_out);
        guard = 0;
    }

    // returns the tensor holding d(init_conv output) per variant: with fusion d(xi), without d(x0)
    void fusion(const FuseRec& f) {
        scope = "mask_fusion_conv";
        const Act g_x0 = grad_of(f.x0);
        if (err) return;
        const int dim = f.xi.C, ch = f.mask.C, H = f.xi.H, W = f.xi.W;
        guard = 2;
        Act df2 = act(2 * dim, H, W), dz2 = act(2 * dim, H, W), df1 = act(2 * dim, H, W), dz1 = act(2 * dim, H, W), gm = act(ch, H, W);
        wgrad("mask_fusion_conv.4.weight", "mask_fusion_conv.4.bias", f.f2, nullptr, g_x0, 3, 1, 1, 0);
        dgrad_to("mask_fusion_conv.4.weight", dim, 2 * dim, 3, 1, g_x0, 0, df2, nullptr);
        silu_bwd(df2, f.z2, dz2);
        wgrad("mask_fusion_conv.2.weight", "mask_fusion_conv.2.bias", f.f1, nullptr, dz2, 3, 1, 1, 0);
        dgrad_to("mask_fusion_conv.2.weight", 2 * dim, 2 * dim, 3, 1, dz2, 0, df1, nullptr);
        silu_bwd(df1, f.z1, dz1);
        wgrad("mask_fusion_conv.0.weight", "mask_fusion_conv.0.bias", f.xi, &f.mask, dz1, 5, 2, 1, 0);
        Slot& sx = slot(f.xi);
        dgrad_to("mask_fusion_conv.0.weight", 2 * dim, dim + ch, 5, 2, dz1, 0, sx.g, nullptr);
        sx.written = true;
        dgrad_to("mask_fusion_conv.0.weight", 2 * dim, dim + ch, 5, 2, dz1, dim, gm, nullptr);
        {
            const float* gp = gm.p; float* mp = gmask.p;
            const size_t per = (size_t)H * W * ch;
            push([=](const FwdCtx& c, hipStream_t s) { return add_into_launch(mp, gp, per * c.B, s); }, "add_into");
        }
        release(df2); release(dz2); release(df1); release(dz1); release(gm);
        guard = 0;
    }
};

static size_t max_wgrad_ws(const fc_unet* u, const Plan& fw, int B) {
    size_t m = 0;
    auto want = [&](int C0, int C1, int Cout, int H, int W, int Hs, int Ws, int KS, int pad, int stride, int ups) {
        WgradArgs a;
        a.C0 = C0; a.C1 = C1; a.Cin = C0 + C1; a.Cout = Cout; a.H = H; a.W = W; a.Hs = Hs; a.Ws = Ws; a.KS = KS; a.pad = pad; a.stride = stride; a.ups = ups; a.B = B;
        const size_t w = conv_wgrad_workspace(a);
        if (w > m) m = w;
    };
    const int hid = u->heads * 32, ch = u->cfg.channels, dim = u->cfg.dim;
    for (const ResRec& r : fw.res) {
        want(r.cout, 0, r.cout, r.x.H, r.x.W, r.x.H, r.x.W, 3, 1, 1, 0);
        want(r.x.C, r.skip.C, r.cout, r.x.H, r.x.W, r.x.H, r.x.W, 3, 1, 1, 0);
        want(r.x.C, r.skip.C, r.cout, r.x.H, r.x.W, r.x.H, r.x.W, 1, 0, 1, 0);
    }
    for (const LinRec& r : fw.lin) { want(hid, 0, r.x.C, r.x.H, r.x.W, r.x.H, r.x.W, 1, 0, 1, 0); want(r.x.C, 0, 3 * hid, r.x.H, r.x.W, r.x.H, r.x.W, 1, 0, 1, 0); }
    for (const MidRec& r : fw.mid) { want(hid, 0, r.x.C, r.x.H, r.x.W, r.x.H, r.x.W, 1, 0, 1, 0); want(r.x.C, 0, 3 * hid, r.x.H, r.x.W, r.x.H, r.x.W, 1, 0, 1, 0); }
    for (const ConvRec& r : fw.convs) want(r.x.C, 0, r.out.C, r.out.H, r.out.W, r.x.H, r.x.W, r.KS, r.pad, r.stride, r.ups);
    want(dim, 0, ch, fw.H, fw.W, fw.H, fw.W, 1, 0, 1, 0);
    want(ch, 0, dim, fw.H, fw.W, fw.H, fw.W, 1, 0, 1, 0);
    return m;
}

int build_backward(fc_unet* u) {
    const fc_unet_config& c = u->cfg;
    const Plan& fw = u->plan[0];
    u->bwd.release();
    u->dgrad_packs.clear();
    u->dgrad_table.release();
    u->dgrad_version = ~0ull;
    if (u->nchains != 1) return fail(FC_E_STATE, "unet: training needs a single-chain plan (unset FLOCODER_AMD_CHAINS)");
    FC_TRY(conv_wgrad_init());
    const int B = fw.maxB, H = fw.H, W = fw.W, HW = H * W, dim = c.dim, ch = c.channels, td = u->td, S = u->S, ncls = c.n_classes;
    BwdBuilder b(u, &fw, &u->bwd, B);
    int maxC = 3 * u->heads * 32;
    for (int cc : u->chans) if (cc > maxC) maxC = cc;
    b.ws_floats = max_wgrad_ws(u, fw, B);
    b.ws = b.dmalloc(b.ws_floats ? b.ws_floats : 4);
    b.s12 = b.dmalloc((size_t)B * maxC * 2);
    {   // chunk partials of the GroupNorm backward: the largest (pixels/64) x channels product over the tape
        size_t m = (size_t)gn_bwd_chunks(H * W) * maxC;
        b.s12p = b.dmalloc((size_t)B * m * 2);
    }
    b.dss = b.dmalloc((size_t)B * S);
    float* redws = b.dmalloc(256);
    (void)redws;
    if (b.err) return b.err;
    if (ncls > 0) {
        u->class_lo = b.off("class_cond_mlp.0.weight");
        const Param& last = u->params[u->pidx.at("class_cond_mlp.3.bias")];
        u->class_hi = last.offset + ((last.numel + 3) & ~3ll);
    }

    if (c.mask_cond) {
        if (!fw.fuse.present) return fail(FC_E_STATE, "unet: mask-conditioned plan without a fusion record");
        b.gmask = b.act(ch, H, W);
        float* gp = b.gmask.p;
        const size_t per = (size_t)HW * ch;
        b.scope = "mask";
        b.push([=](const FwdCtx& cx, hipStream_t s) -> int { FC_HIP(hipMemsetAsync(gp, 0, per * cx.B * sizeof(float), s)); return FC_OK; }, "memset");
    }
    // -- head: d(out) NCHW -> NHWC, final_conv (unet.py:372) --
    b.scope = "final_conv";
    Act dv = b.act(ch, H, W);
    {
        float* dvp = dv.p;
        b.push([=](const FwdCtx& cx, hipStream_t s) { return nchw_to_nhwc_launch(cx.d_out, dvp, cx.B, ch, HW, ch, cx.B, s); }, "nchw_to_nhwc");
    }
    b.wgrad("final_conv.weight", "final_conv.bias", fw.head, nullptr, dv, 1, 0, 1, 0);
    b.dgrad("final_conv.weight", ch, dim, 1, 0, dv, 0, fw.head);
    // ---- the deferred launches: activations only weight gradients read, every weight gradient recorded so far (one launch per kernel size),
    // their split reduction, the norm / FiLM parameter gradients and the FiLM projections' weight gradients.  Emitted TWICE (round 3): once
    // behind mid_block1 -- final_*, ups.* and mid_* are complete in the flat gradient vector from then on, [grad_split, end), and a data-
    // parallel trainer can put that bucket on the wire while the rest of the chain runs (fc_unet_backward_parts) -- and once at the end.
    const float* te_film = fw.t_emb;
    auto emit_deferred = [&]() -> int {
    // -- the activations only the deferred weight gradients read, recomputed in one launch --
    if (!b.fin_jobs.empty()) {
        b.scope = "wgrad";
        std::vector<int> bps;
        std::vector<int2> blocks;
        size_t lds = 0;
        for (size_t j = 0; j < b.fin_jobs.size(); ++j) {
            const FinalizeArgs& f = b.fin_jobs[j];
            bps.push_back(finalize_blocks_per_sample(f.HW, f.C));
            for (int k = 0; k < bps.back() * B; ++k) blocks.push_back(make_int2((int)j, k));
            if (finalize_lds_bytes(f) > lds) lds = finalize_lds_bytes(f);
        }
        FinalizeArgs* jd = reinterpret_cast<FinalizeArgs*>(b.dmalloc((b.fin_jobs.size() * sizeof(FinalizeArgs) + 3) / 4 + 4));
        int* pd = reinterpret_cast<int*>(b.dmalloc(bps.size() + 4));
        int2* bd = reinterpret_cast<int2*>(b.dmalloc(blocks.size() * 2 + 4));
        if (b.err) return b.err;
        FC_HIP(hipMemcpy(jd, b.fin_jobs.data(), b.fin_jobs.size() * sizeof(FinalizeArgs), hipMemcpyHostToDevice));
        FC_HIP(hipMemcpy(pd, bps.data(), bps.size() * sizeof(int), hipMemcpyHostToDevice));
        FC_HIP(hipMemcpy(bd, blocks.data(), blocks.size() * sizeof(int2), hipMemcpyHostToDevice));
        const int nblk = (int)blocks.size();
        b.push([=](const FwdCtx& cx, hipStream_t s) { return cx.B == B ? finalize_table_launch(jd, pd, bd, nblk, lds, s) : (int)FC_OK; }, "finalize_table");
    }
    // -- the deferred weight gradients: one launch per kernel size --
    for (auto& kv : b.wclasses) {
        BwdBuilder::WgradClass& wc = kv.second;
        if (wc.jobs.empty()) continue;
        b.scope = "wgrad";
        WgradDev* jd = reinterpret_cast<WgradDev*>(b.dmalloc((wc.jobs.size() * sizeof(WgradDev) + 3) / 4 + 4));
        int2* bd = reinterpret_cast<int2*>(b.dmalloc(wc.blocks.size() * 2 + 4));
        if (b.err) return b.err;
        FC_HIP(hipMemcpy(jd, wc.jobs.data(), wc.jobs.size() * sizeof(WgradDev), hipMemcpyHostToDevice));
        FC_HIP(hipMemcpy(bd, wc.blocks.data(), wc.blocks.size() * sizeof(int2), hipMemcpyHostToDevice));
        const int nblk = (int)wc.blocks.size(), ks = kv.first;
        const size_t lds = wc.lds;
        double fl = 0;
        for (const WgradDev& d : wc.jobs) fl += 2.0 * d.a.H * d.a.W * ks * ks * (double)d.a.Cin * d.a.Cout;
        b.push([=](const FwdCtx& cx, hipStream_t s) { return cx.B == B ? conv_wgrad_table_launch(ks, jd, bd, nblk, lds, cx.grads, s) : (int)FC_OK; },
               "conv_wgrad_table", fl);
    }
    // -- the split partials of every weight gradient above, summed in one launch (full-batch steps; smaller batches reduced per launch) --
    if (!b.wred_jobs.empty()) {
        b.scope = "wgrad";
        std::vector<int2> blocks;
        for (size_t j = 0; j < b.wred_jobs.size(); ++j) {
            const WredJob& w = b.wred_jobs[j];
            const size_t total = w.nw + (w.db >= 0 ? (size_t)w.nb : 0);
            for (size_t k = 0; k < (total + 63) / 64; ++k) blocks.push_back(make_int2((int)j, (int)k));
        }
        WredJob* jd = reinterpret_cast<WredJob*>(b.dmalloc((b.wred_jobs.size() * sizeof(WredJob) + 3) / 4 + 4));
        int2* bd = reinterpret_cast<int2*>(b.dmalloc(blocks.size() * 2 + 4));
        if (b.err) return b.err;
        FC_HIP(hipMemcpy(jd, b.wred_jobs.data(), b.wred_jobs.size() * sizeof(WredJob), hipMemcpyHostToDevice));
        FC_HIP(hipMemcpy(bd, blocks.data(), blocks.size() * sizeof(int2), hipMemcpyHostToDevice));
        const int nblk = (int)blocks.size();
        b.push([=](const FwdCtx& cx, hipStream_t s) { return cx.B == B ? wgrad_reduce_table_launch(jd, bd, nblk, cx.grads, s) : (int)FC_OK; }, "wgrad_reduce");
    }
    // -- parameter gradients of every norm layer and the FiLM gradients, one launch --
    if (!b.norm_jobs.empty()) {
        b.scope = "norms";
        NormJob* jd = reinterpret_cast<NormJob*>(b.dmalloc((b.norm_jobs.size() * sizeof(NormJob) + 3) / 4 + 4));
        if (b.err) return b.err;
        FC_HIP(hipMemcpy(jd, b.norm_jobs.data(), b.norm_jobs.size() * sizeof(NormJob), hipMemcpyHostToDevice));
        const int nj = (int)b.norm_jobs.size();
        int mc = 0;
        for (const NormJob& j : b.norm_jobs) if (j.C > mc) mc = j.C;
        float* dssp = b.dss;
        b.push([=](const FwdCtx& cx, hipStream_t s) { return norm_param_grads_table_launch(jd, nj, mc, cx.grads, dssp, S, cx.B, s); }, "norm_param_grads");
    }
        if (!b.film_blocks.empty()) {
            b.scope = "resblock.mlp";
            const float* te = te_film;
            float* dss = b.dss;
        {   // every block's mlp.1 weight / bias gradient: one table-driven launch
            std::vector<DenseWJob> jobs;
            std::vector<int2> blocks;
            for (const std::string& p : b.film_blocks) {
                const int rows = (int)u->params[u->pidx.at(p + ".mlp.1.bias")].numel;
                jobs.push_back({u->ss_off.at(p), rows, b.off(p + ".mlp.1.weight"), b.off(p + ".mlp.1.bias")});
                const int nb = cdiv(rows * td, 256);
                for (int k = 0; k < nb; ++k) blocks.push_back(make_int2((int)jobs.size() - 1, k));
            }
            DenseWJob* jd = reinterpret_cast<DenseWJob*>(b.dmalloc((jobs.size() * sizeof(DenseWJob) + 3) / 4 + 4));
            int2* bd = reinterpret_cast<int2*>(b.dmalloc(blocks.size() * 2 + 4));
            if (b.err) return b.err;
            FC_HIP(hipMemcpy(jd, jobs.data(), jobs.size() * sizeof(DenseWJob), hipMemcpyHostToDevice));
            FC_HIP(hipMemcpy(bd, blocks.data(), blocks.size() * sizeof(int2), hipMemcpyHostToDevice));
            const int nblk = (int)blocks.size();
            b.push([=](const FwdCtx& cx, hipStream_t s) { return dense_bwd_w_table_launch(jd, bd, nblk, dss, S, te, 2, cx.grads, cx.B, td, s); }, "dense_bwd_w");
        }
        }
        b.fin_jobs.clear(); b.wclasses.clear(); b.wred_jobs.clear(); b.norm_jobs.clear(); b.film_blocks.clear();
        b.pinned.clear();          // what the tables read may be recycled by later plan entries
        return b.err;
    };
    // -- the tape in reverse --
    u->bwd_split_op = -1; u->grad_split = 0;
    // two buckets only for a trainer that asked for them (fc_unet_set_grad_buckets): a single process gains nothing from the second set of
    // table launches (stl_sd step 2.73 ms with them, 2.65 without)
    const bool no_buckets = !u->want_buckets || std::getenv("FLOCODER_AMD_NO_GRAD_BUCKETS") != nullptr;
    for (int i = (int)fw.tape.size() - 1; i >= 0 && !b.err; --i) {
        const TapeItem& t = fw.tape[i];
        if (t.kind == 0) {
            b.resblock(fw.res[t.idx]);
            if (!no_buckets && fw.res[t.idx].p == "mid_block1" && u->has("ups.0.0.mlp.1.weight")) {
                if (emit_deferred() != FC_OK) return b.err;
                u->bwd_split_op = (int)u->bwd.ops.size();
                u->grad_split = b.off("ups.0.0.mlp.1.weight");      // table order: ..., downs.*, ups.*, mid_*, final_*: the tail is one range
            }
            continue;
        }
        if (false) b.resblock(fw.res[t.idx]);
        else if (t.kind == 1) b.linattn(fw.lin[t.idx]);
        else if (t.kind == 2) b.midattn(fw.mid[t.idx]);
        else if (t.kind == 3) b.resample(fw.convs[t.idx]);
        else b.inject(fw.inj[t.idx]);
    }
    if (b.err) return b.err;
    // -- mask_fusion_conv (unet.py:298-305), then init_conv (unet.py:295); d(x) and d(mask) leave through the boundary when asked for --
    if (c.mask_cond) b.fusion(fw.fuse);
    if (b.err) return b.err;
    {
        b.scope = "init_conv";
        Act xin = b.act(ch, H, W);
        float* xp = xin.p;
        b.push([=](const FwdCtx& cx, hipStream_t s) { return nchw_to_nhwc_launch(cx.x, xp, cx.B, ch, HW, ch, cx.B, s); }, "nchw_to_nhwc");
        const Act g0 = b.grad_of(fw.x0);
        if (b.err) return b.err;
        const float* gx0 = g0.p;
        const float* gxi = c.mask_cond ? b.slot(fw.fuse.xi).g.p : g0.p;
        // the gradient of init_conv's output lives in d(xi) when the fusion ran, in d(x0) otherwise: two guarded variants
        WgradArgs a;
        a.x0 = xin.p; a.C0 = ch; a.H = H; a.W = W; a.Hs = H; a.Ws = W; a.Cin = ch; a.Cout = dim; a.KS = 1; a.B = B;
        a.ws = b.ws; a.ws_floats = b.ws_floats;
        const int64_t wo = b.off("init_conv.weight"), bo = b.off("init_conv.bias");
        b.push([=](const FwdCtx& cx, hipStream_t s) {
            WgradArgs q = a;
            q.B = cx.B; q.dy = cx.mask_fuse ? gxi : gx0; q.dw = cx.grads + wo; q.db = cx.grads + bo;
            return conv_wgrad_launch(q, s);
        }, "conv_wgrad", 2.0 * HW * (double)ch * dim);
        // d(x) = init_conv^T d(.) as an NHWC tensor, converted to the NCHW boundary layout on request (fc_unet_backward_ex)
        Act dxn = b.act(ch, H, W);
        float* wp = b.dmalloc((size_t)dim * ch);
        u->dgrad_packs.push_back({b.off("init_conv.weight"), wp, dim, ch, 1, 0, ch});
        ConvArgs q;
        q.s0.C = dim; q.Hs = H; q.Ws = W; q.KS = 1; q.w = wp;
        q.B = B; q.H = H; q.W = W; q.Cout = ch; q.out = dxn.p; q.Cin = dim;
        q.s0.p = gx0;
        ConvGeom g;
        if ((b.err = conv_plan(q, TILE_AUTO, &g)) != FC_OK) return b.err;
        const int tile = g.tile;
        const float* dxp = dxn.p;
        b.push([=](const FwdCtx& cx, hipStream_t s) -> int {
            if (!cx.dx_out) return FC_OK;
            ConvArgs k = q;
            k.B = cx.B; k.s0.p = cx.mask_fuse ? gxi : gx0;
            FC_TRY(conv_launch(k, tile, s));
            return nhwc_to_nchw_launch(dxp, cx.dx_out, cx.B, ch, HW, ch, s);
        }, "dgrad(init_conv)");
        if (c.mask_cond) {
            const float* gm = b.gmask.p;
            b.push([=](const FwdCtx& cx, hipStream_t s) -> int {
                if (!cx.dmask_out) return FC_OK;
                return nhwc_to_nchw_launch(gm, cx.dmask_out, cx.B, ch, HW, ch, s);
            }, "nhwc_to_nchw");
        }
    }
    if (b.err) return b.err;
    if (emit_deferred() != FC_OK) return b.err;
    // -- conditioning: every ResnetBlock.mlp (unet.py:79-82,90-92), then time_mlp / class_cond_mlp (unet.py:199-212,310-316) --
    {
        b.scope = "resblock.mlp";
        const float* te = fw.t_emb;
        float* dss = b.dss;
        float* dT = b.dmalloc((size_t)B * td);
        const float* wt = u->P("__ss_wt");
        b.push([=](const FwdCtx& cx, hipStream_t s) { return dense_bwd_x_launch(dss, S, wt, 1, S, te, 2, dT, 0, cx.B, td, S, s); }, "dense_bwd_x");

        b.scope = "time_mlp";
        float *se = b.dmalloc((size_t)B * dim), *z1 = b.dmalloc((size_t)B * td), *dz1 = b.dmalloc((size_t)B * td);
        const float *fr = u->freqs, *w1 = u->R("time_mlp.1.weight"), *b1 = u->R("time_mlp.1.bias"), *w3 = u->R("time_mlp.3.weight");
        const int64_t o1w = b.off("time_mlp.1.weight"), o1b = b.off("time_mlp.1.bias"), o3w = b.off("time_mlp.3.weight"), o3b = b.off("time_mlp.3.bias");
        b.push([=](const FwdCtx& cx, hipStream_t s) -> int {
            FC_TRY(sin_emb_launch(cx.time, fr, se, cx.B, dim, s));
            FC_TRY(dense_fwd_launch(se, 0, w1, b1, z1, cx.B, dim, td, s));
            FC_TRY(dense_bwd_w_launch(dT, td, z1, 1, cx.grads + o3w, cx.grads + o3b, cx.B, td, td, s));
            FC_TRY(dense_bwd_x_launch(dT, td, w3, 0, 0, z1, 1, dz1, 0, cx.B, td, td, s));
            return dense_bwd_w_launch(dz1, td, se, 0, cx.grads + o1w, cx.grads + o1b, cx.B, dim, td, s);
        }, "time_mlp_bwd");
        if (ncls > 0) {
            b.scope = "class_cond_mlp";
            float *e = b.dmalloc((size_t)B * td), *cz1 = b.dmalloc((size_t)B * td), *dcz1 = b.dmalloc((size_t)B * td), *de = b.dmalloc((size_t)B * td);
            float* dTm = b.dmalloc((size_t)B * td);   // d(t_emb) of the rows that HAVE a class (a row with id < 0 got no class term in the forward)
            const float *E = u->R("class_cond_mlp.0.weight"), *cw1 = u->R("class_cond_mlp.1.weight"), *cb1 = u->R("class_cond_mlp.1.bias"),
                        *cw3 = u->R("class_cond_mlp.3.weight");
            const int64_t oE = b.off("class_cond_mlp.0.weight"), c1w = b.off("class_cond_mlp.1.weight"), c1b = b.off("class_cond_mlp.1.bias"),
                          c3w = b.off("class_cond_mlp.3.weight"), c3b = b.off("class_cond_mlp.3.bias");
            b.push([=](const FwdCtx& cx, hipStream_t s) -> int {
                if (!cx.ids) return FC_OK;    // no conditioning this step: these parameters get no gradient (left zero, the optimiser skips them)
                FC_TRY(gather_rows_launch(E, cx.ids, e, cx.B, td, ncls, s));
                FC_TRY(dense_fwd_launch(e, 0, cw1, cb1, cz1, cx.B, td, td, s));
                FC_TRY(mask_rows_launch(dT, td, cx.ids, dTm, cx.B, td, ncls, s));
                FC_TRY(dense_bwd_w_launch(dTm, td, cz1, 1, cx.grads + c3w, cx.grads + c3b, cx.B, td, td, s));
                FC_TRY(dense_bwd_x_launch(dTm, td, cw3, 0, 0, cz1, 1, dcz1, 0, cx.B, td, td, s));
                FC_TRY(dense_bwd_w_launch(dcz1, td, e, 0, cx.grads + c1w, cx.grads + c1b, cx.B, td, td, s));
                FC_TRY(dense_bwd_x_launch(dcz1, td, cw1, 0, 0, nullptr, 0, de, 0, cx.B, td, td, s));
                return scatter_rows_launch(de, cx.ids, cx.grads + oE, cx.B, td, ncls, s);
            }, "class_mlp_bwd");
        }
    }
    if (b.err) return b.err;
    for (auto& kv : fw.named) {
        auto it = b.gmap.find(kv.second.p);
        if (it != b.gmap.end()) u->bwd.named["grad:" + kv.first] = it->second.g;
    }
    u->bwd.maxB = B; u->bwd.H = H; u->bwd.W = W;
    std::vector<PackJob> jobs;
    for (const auto& k : u->dgrad_packs) jobs.push_back({u->raw + k.src, k.dst, 5, k.O, k.I, k.KS, k.ci0, k.nci, 0});
    return pack_table_build(jobs, &u->dgrad_table);
}

}  // namespace fc

extern "C" {

int fc_unet_train_reserve(fc_unet* u, int max_batch, int height, int width) {
    if (!u) return fail(FC_E_ARG, "fc_unet_train_reserve: null handle");
    if (!u->keep_all && u->device >= 0) {   // from now on this object's plans keep q/k/v and the attention output for the backward
        u->keep_all = true;
        if (u->maxB > 0) {                  // an inference plan exists: rebuild it in keep-everything form
            const int mb = u->maxB > max_batch ? u->maxB : max_batch;
            u->maxB = 0;
            FC_TRY(fc_unet_reserve(u, mb, height, width));
        }
    }
    FC_TRY(fc_unet_reserve(u, max_batch, height, width));
    if (u->bwd.maxB == u->plan[0].maxB && u->bwd.H == height && u->bwd.W == width && u->bwd.maxB > 0) return FC_OK;
    FC_HIP(hipSetDevice(u->device));
    FC_HIP(hipDeviceSynchronize());
    const int r = build_backward(u);
    if (r != FC_OK) { u->bwd.release(); u->dgrad_packs.clear(); u->dgrad_table.release(); }
    return r;
}

int fc_unet_backward_ex(fc_unet* u, const float* x, const float* time, const int64_t* ids, const float* mask, int mask_is_ones,
                        const float* d_out, float* grads, int64_t numel, float* dx_out, float* dmask_out, int B, int H, int W, void* stream) {
    return fc_unet_backward_parts(u, x, time, ids, mask, mask_is_ones, d_out, grads, numel, dx_out, dmask_out, B, H, W, 0, 1, stream);
}

int fc_unet_set_grad_buckets(fc_unet* u, int on) {
    if (!u) return fail(FC_E_ARG, "fc_unet_set_grad_buckets: null handle");
    if (u->want_buckets == (on != 0)) return FC_OK;
    u->want_buckets = on != 0;
    if (u->device >= 0 && u->bwd.maxB > 0) {      // the backward plan in place has the other form: drop it, fc_unet_train_reserve rebuilds
        FC_HIP(hipSetDevice(u->device));
        FC_HIP(hipDeviceSynchronize());
        u->bwd.release(); u->dgrad_packs.clear(); u->dgrad_table.release();
        u->dgrad_version = ~0ull;
        u->arena_touched(0);
    }
    return FC_OK;
}

int fc_unet_grad_buckets(const fc_unet* u, int64_t* split_offset) {
    if (!u || !split_offset) return fail(FC_E_ARG, "fc_unet_grad_buckets: null argument");
    *split_offset = u->bwd_split_op >= 0 ? u->grad_split : 0;
    return u->bwd.maxB > 0 ? (u->bwd_split_op >= 0 ? 2 : 1) : 0;
}

int fc_unet_backward_parts(fc_unet* u, const float* x, const float* time, const int64_t* ids, const float* mask, int mask_is_ones,
                           const float* d_out, float* grads, int64_t numel, float* dx_out, float* dmask_out, int B, int H, int W,
                           int first_part, int last_part, void* stream) {
    if (!u || !x || !time || !d_out || !grads || B < 1) return fail(FC_E_ARG, "fc_unet_backward: null argument");
    if (u->bwd.maxB < B || u->bwd.H != H || u->bwd.W != W || u->plan[0].maxB < B) return fail(FC_E_STATE, "unet: no backward plan for this shape; call fc_unet_train_reserve");
    if (numel != u->raw_numel) return fail(FC_E_ARG, "fc_unet_backward: gradient vector must have " + std::to_string(u->raw_numel) + " floats (padded table layout)");
    if (!u->loaded) return fail(FC_E_STATE, "unet: weights not loaded");
    if (u->arena_train_rows != B)
        return fail(FC_E_STATE, "fc_unet_backward: the activation arena does not hold a training forward of this batch (another forward, an "
                                "integration, a profile run or a re-plan came in between); run fc_unet_forward after fc_unet_train_reserve again first");
    hipStream_t s = static_cast<hipStream_t>(stream);
    FC_HIP(hipSetDevice(u->device));
    if (u->dgrad_version != u->param_version) {     // data-gradient operands follow the parameters
        FC_TRY(pack_table_launch(u->dgrad_table, s));
        u->dgrad_version = u->param_version;
    }
    if (first_part < 0 || last_part > 1 || first_part > last_part) return fail(FC_E_ARG, "fc_unet_backward_parts: parts are 0 (through mid_block1) and 1 (the rest)");
    if (first_part == 0) FC_HIP(hipMemsetAsync(grads, 0, (size_t)numel * sizeof(float), s));
    FwdCtx c;
    c.x = x; c.x_mod = B; c.time = time; c.ids = u->cfg.n_classes > 0 ? ids : nullptr; c.ids_mod = B; c.B = B;
    c.mask = u->cfg.mask_cond ? mask : nullptr;
    c.mask_fuse = (c.mask && !mask_is_ones) ? 1 : 0;
    c.d_out = d_out; c.grads = grads; c.dx_out = dx_out; c.dmask_out = c.mask ? dmask_out : nullptr;
    const int nops = (int)u->bwd.ops.size(), split = (u->bwd_split_op >= 0 && u->bwd_split_op <= nops) ? u->bwd_split_op : nops;
    const int lo = first_part == 0 ? 0 : split, hi = last_part == 0 ? split : nops;
    for (int i = lo; i < hi; ++i) FC_TRY(u->bwd.ops[i](c, s));
    return FC_OK;
}

int fc_unet_backward(fc_unet* u, const float* x, const float* time, const int64_t* ids, const float* d_out, float* grads, int64_t numel,
                     int B, int H, int W, void* stream) {
    return fc_unet_backward_ex(u, x, time, ids, nullptr, 0, d_out, grads, numel, nullptr, nullptr, B, H, W, stream);
}

int fc_unet_class_param_range(const fc_unet* u, int64_t* lo, int64_t* hi) {
    if (!u || !lo || !hi) return fail(FC_E_ARG, "fc_unet_class_param_range: null argument");
    *lo = u->class_lo; *hi = u->class_hi;
    return FC_OK;
}

int fc_flow_interp(const float* source_dev, const float* target_dev, const float* t_dev, float* x_out_dev, float* v_out_dev, int batch,
                   int64_t per_sample, void* stream) {
    if (!source_dev || !target_dev || !t_dev || !x_out_dev || !v_out_dev || batch < 1) return fail(FC_E_ARG, "fc_flow_interp: null argument");
    return flow_interp_launch(source_dev, target_dev, t_dev, x_out_dev, v_out_dev, batch, (int)per_sample, static_cast<hipStream_t>(stream));
}

int fc_flow_prepare(const float* source_dev, const float* target_dev, const int64_t* pairing_dev, const float* u_dev, float t_eps, float warp_s,
                    float t_scale, const int64_t* class_ids_dev, int n_classes, float* t_out_dev, float* time_out_dev, float* x_out_dev,
                    float* v_out_dev, int* id_flag_dev, int batch, int64_t per_sample, void* stream) {
    if (!source_dev || !target_dev || !u_dev || !t_out_dev || !time_out_dev || !x_out_dev || !v_out_dev || batch < 1 || per_sample < 1)
        return fail(FC_E_ARG, "fc_flow_prepare: null argument");
    if (warp_s < 0.f || warp_s > 1.5f) return fail(FC_E_ARG, "fc_flow_prepare: warp parameter s out of bounds (sampling.py:27)");
    return flow_prepare_launch(source_dev, target_dev, pairing_dev, u_dev, t_eps, warp_s, t_scale, class_ids_dev, n_classes, t_out_dev, time_out_dev,
                               x_out_dev, v_out_dev, id_flag_dev, batch, (int)per_sample, static_cast<hipStream_t>(stream));
}

int fc_mse_loss_grad(const float* v_dev, const float* target_dev, float* dv_out_dev, float* loss_out_dev, float* ws256_dev, int64_t numel,
                     void* stream) {
    if (!v_dev || !target_dev || !loss_out_dev || !ws256_dev || numel < 1) return fail(FC_E_ARG, "fc_mse_loss_grad: null argument");
    return mse_loss_grad_launch(v_dev, target_dev, dv_out_dev, loss_out_dev, ws256_dev, (size_t)numel, static_cast<hipStream_t>(stream));
}

int fc_grad_clip_coef(const float* grads_dev, int64_t numel, const float* grads2_dev, int64_t numel2, float max_norm, float* norm_coef_out_dev,
                      float* ws256_dev, void* stream) {
    if (!grads_dev || !norm_coef_out_dev || !ws256_dev) return fail(FC_E_ARG, "fc_grad_clip_coef: null argument");
    return grad_clip_coef_launch(grads_dev, (size_t)numel, grads2_dev, grads2_dev ? (size_t)numel2 : 0, max_norm, norm_coef_out_dev, ws256_dev,
                                 static_cast<hipStream_t>(stream));
}

int fc_adam_ema_step(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, float* ema_dev, int64_t numel,
                     const float* clip_coef_dev, float lr, float beta1, float beta2, float eps, int step, float ema_decay, int apply_adam,
                     void* stream) {
    if (!params_dev || numel < 0 || (apply_adam && (!grads_dev || !exp_avg_dev || !exp_avg_sq_dev || step < 1)))
        return fail(FC_E_ARG, "fc_adam_ema_step: bad argument");
    return adam_ema_launch(params_dev, grads_dev, exp_avg_dev, exp_avg_sq_dev, ema_dev, (size_t)numel, clip_coef_dev, lr, beta1, beta2, eps, step,
                           ema_decay, apply_adam, static_cast<hipStream_t>(stream));
}

int fc_adam_ema_step_guarded(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, float* ema_dev, int64_t numel,
                             const float* clip_coef_dev, float lr, float beta1, float beta2, float eps, int step, float ema_decay, int apply_adam,
                             const int* skip_flag_dev, void* stream) {
    if (!params_dev || numel < 0 || (apply_adam && (!grads_dev || !exp_avg_dev || !exp_avg_sq_dev || step < 1)))
        return fail(FC_E_ARG, "fc_adam_ema_step_guarded: bad argument");
    return adam_ema_launch(params_dev, grads_dev, exp_avg_dev, exp_avg_sq_dev, ema_dev, (size_t)numel, clip_coef_dev, lr, beta1, beta2, eps, step,
                           ema_decay, apply_adam, static_cast<hipStream_t>(stream), skip_flag_dev);
}

}  // extern "C"
