// Generic pieces of the native runtimes (unet.hip, vae.hip): parameter store with the reference's state_dict names, weight
// packing, launch plan over a fixed activation arena, and the builder that emits implicit-GEMM launches.
#pragma once
#include <cmath>
#include <functional>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace fc {

struct Param {
    std::string name;
    int64_t shape[4] = {0, 0, 0, 0};
    int64_t numel = 0, offset = 0;
};

struct Act { float* p = nullptr; int C = 0, H = 0, W = 0; };
struct Stat { float* p = nullptr; int G = 0, T = 0; float n_t = 0.f; };

struct FwdCtx {             // per-call inputs of one forward
    const float* x = nullptr;          // NCHW [x_mod][C][H][W]
    int x_mod = 0;                     // row b reads sample b % x_mod (CFG: both halves share x)
    const float* time = nullptr;       // [B]
    const int64_t* ids = nullptr;      // [ids_mod] or null
    int ids_mod = 0, null_from = 0;
    const float* mask = nullptr;       // NCHW [x_mod][C][H][W] or null
    int mask_fuse = 0;                 // run mask_fusion_conv (mask present and not all ones)
    float* out = nullptr;              // NCHW [B][C][H][W]
    int B = 0;
    const float* d_out = nullptr;      // backward only: NCHW gradient of `out`
    float* grads = nullptr;            // backward only: flat parameter gradients, table layout
    float* dx_out = nullptr;           // backward only, optional: NCHW gradient of the input x
    float* dmask_out = nullptr;        // backward only, optional: NCHW gradient of the mask
    EulerTail euler;                   // integrator only: the Euler update rides in final_conv
    CondFetch fetch;                   // integrator only: conditioning rows of every evaluation were computed up front
};
using Op = std::function<int(const FwdCtx&, hipStream_t)>;

// What the backward pass needs to know about the forward's modules: the raw tensors and statistics each one left in the arena.
struct ResRec { std::string p; Act x, skip, h1, h2, rb, out; Stat st1, st2; int cout = 0; };
struct LinRec { std::string p; Act x, qkv, lao, yb, out; float* ctx = nullptr; Stat gn1, sty; };
struct MidRec { Act x, qkv, ao, out; Stat gn1; };
struct ConvRec { std::string name; Act x, out; int KS = 1, pad = 0, stride = 1, ups = 0; };
struct InjRec { std::string name; Act x, mr, z, out; };                      // x + SiLU(conv3x3(cat[x, bilinear(mask)])), unet.py:336-340
struct FuseRec { bool present = false; Act xi, mask, z1, f1, z2, f2, x0; };  // mask_fusion_conv, unet.py:298-305
struct TapeItem { int kind, idx; };   // 0 ResRec, 1 LinRec, 2 MidRec, 3 ConvRec, 4 InjRec -- in forward order

struct Plan {                 // one launch plan + activation arena for up to maxB rows
    int maxB = 0, H = 0, W = 0;
    std::vector<Op> ops;
    int side_ops = 0, join_at = 0;   // ops [0, side_ops) depend on nothing the ops [side_ops, join_at) produce or read: they may run beside them
    std::vector<std::string> op_kernel, op_what;  // parallel to ops: kernel family, reference module it serves
    std::vector<double> op_flops;                  // algorithmic FLOPs per sample of that launch
    std::vector<double> op_bytes_ps, op_bytes_fixed;   // algorithmic HBM bytes of that launch: per sample (activations in + out, each once) and per launch (weights); 0 = not stated
    std::vector<void*> allocs;
    double flops = 0.0;            // per sample, the reference's arithmetic (2 x MACs of every module as upstream computes it)
    double flops_executed = 0.0;   // per sample, what the launches execute (less where upsampling is folded into the weights)
    float *t_emb = nullptr, *ss = nullptr;
    std::map<std::string, Act> named;              // debug taps: block outputs by reference module name
    std::vector<ResRec> res; std::vector<LinRec> lin; std::vector<MidRec> mid; std::vector<ConvRec> convs; std::vector<InjRec> inj;
    FuseRec fuse;
    std::vector<TapeItem> tape;
    Act x0, head;                                   // init_conv output, final_res_block output
    int n_meet = 0;                                 // launches whose workgroups wait for each other (fused Block tails across workgroups)
    std::vector<unsigned*> fin_sync;                // their arrival counters (fc_debug_unet_break_meeting)
    std::vector<int> fin_kind;                      // ... and what waits on them: 0 a convolution's Block tail, 1 the linear attention's close
    void release() {
        for (void* p : allocs) dev_free(p);
        *this = Plan();
    }
};

struct PackOp { int kind; int64_t src, dst; int a, b, c, d, e = 0; };
// kind 0 conv OIHW(O=a,I=b,KH=c,KW=d), 1 s2d, 2 transpose(R=a,Cc=b,ld=c,col0=d), 3 copy(a floats), 4 conv with channel padding(O=a,I=b,KK=c,Opad=d,Ipad=e)

static const char* const kTileNames[] = {"conv_igemm<M128,N32>", "conv_igemm<M128,N64>", "conv_igemm<M64,N32,K2>", "conv_igemm<M32,N32,K4>",
                                         "conv_igemm<M64,N64,K2>", "conv_igemm<M256,N64>"};

// Parameters as the reference stores them (`raw`, flat, table order, every tensor 16-byte aligned) plus the operand
// layouts the kernels read (`packed`), and the list of re-layout launches that turns one into the other.
struct ParamStore {
    std::vector<Param> params;
    std::unordered_map<std::string, int> pidx;
    int64_t raw_numel = 0, packed_numel = 0;
    float *raw = nullptr, *packed = nullptr;
    std::unordered_map<std::string, int64_t> pk;  // name -> offset into packed
    std::vector<PackOp> packops;
    PackTable pack_table;                          // all of `packops` as one launch
    bool loaded = false;

    bool want_b3 = false;   // also keep 3x3 / 1x1 conv weights as split bf16 (pack kind 8): the codecs, for their opt-in split-bf16 arithmetic
    std::unordered_map<int64_t, int64_t> b3_of;   // packed offset of a weight -> packed offset of its split-bf16 copy
    const float* B3(const float* w) const {         // the split-bf16 copy of packed weight pointer `w`, or null
        auto it = b3_of.find(w - packed);
        return it == b3_of.end() ? nullptr : packed + it->second;
    }
    bool want_k8 = false;   // also keep 3x3 / 1x1 conv weights in the k-step-quad layout (pack kind 6) where the shapes allow: the U-Net's low-resolution layers
    // the k-step-quad copy of a convolution weight, or null when there is none
    const float* P8(const std::string& n) const { auto it = pk.find(n + "#k8"); return it == pk.end() ? nullptr : packed + it->second; }
    const float* R(const std::string& n) const { return raw + params[pidx.at(n)].offset; }
    const float* P(const std::string& n) const { return packed + pk.at(n); }
    bool has(const std::string& n) const { return pidx.count(n) != 0; }

    void declare(const std::string& name, std::initializer_list<int64_t> shape) {
        Param p;
        p.name = name;
        p.numel = 1;
        int i = 0;
        for (int64_t s : shape) { p.shape[i++] = s; p.numel *= s; }
        p.offset = raw_numel;
        raw_numel += (p.numel + 3) & ~3ll;  // keep every tensor 16-byte aligned inside `raw`
        pidx[name] = (int)params.size();
        params.push_back(p);
    }
    int64_t pk_alloc(const std::string& name, int64_t numel) {
        const int64_t off = packed_numel;
        pk[name] = off;
        packed_numel += (numel + 3) & ~3ll;
        return off;
    }
    void decl_conv(const std::string& n, int O, int I, int K, bool bias = true) {
        declare(n + ".weight", {O, I, K, K});
        if (bias) declare(n + ".bias", {O});
        const int64_t dst = pk_alloc(n + ".weight", (int64_t)O * I * K * K);
        packops.push_back({0, params[pidx[n + ".weight"]].offset, dst, O, I, K, K});
        if (want_b3 && (K == 3 || K == 1)) {
            const int Ipad = (I + 15) / 16 * 16;
            const int64_t d3 = pk_alloc(n + ".weight#b3", (int64_t)O * Ipad * K * K);
            packops.push_back({8, params[pidx[n + ".weight"]].offset, d3, O, I, K * K, Ipad});
            b3_of[dst] = d3;
        }
        if (want_k8 && (K == 3 || K == 1) && I % 32 == 0 && O % 32 == 0) {
            const int64_t d8 = pk_alloc(n + ".weight#k8", (int64_t)O * I * K * K);
            packops.push_back({6, params[pidx[n + ".weight"]].offset, d8, O, I, K * K, 0});
        }
    }
    // a 3x3 convolution behind nn.Upsample(nearest, x2): besides the plain packed copy, the four parity kernels of the folded form (pack kinds 9 / 10)
    void decl_conv_up2(const std::string& n, int O, int I) {
        decl_conv(n, O, I, 3);
        const int64_t d4 = pk_alloc(n + ".weight#up4", (int64_t)16 * O * I);
        packops.push_back({9, params[pidx[n + ".weight"]].offset, d4, O, I, 9, 0});
        if (want_b3) {
            const int Ipad = (I + 15) / 16 * 16;
            const int64_t d3 = pk_alloc(n + ".weight#up4b3", (int64_t)16 * O * Ipad);
            packops.push_back({10, params[pidx[n + ".weight"]].offset, d3, O, I, 9, Ipad});
            for (int par = 0; par < 4; ++par) b3_of[d4 + (int64_t)par * 4 * O * I] = d3 + (int64_t)par * 4 * O * Ipad;
        }
    }
    const float* PUP(const std::string& n) const {     // [parity][4 taps][I][O], or null when the layer was not declared with decl_conv_up2
        auto it = pk.find(n + "#up4");
        return it == pk.end() ? nullptr : packed + it->second;
    }
    void decl_linear_t(const std::string& n, int O, int I) {  // stored transposed [I][O]
        declare(n + ".weight", {O, I});
        declare(n + ".bias", {O});
        const int64_t dst = pk_alloc(n + ".weight", (int64_t)O * I);
        packops.push_back({2, params[pidx[n + ".weight"]].offset, dst, O, I, O, 0});
    }
    void decl_norm(const std::string& n, int C) {
        declare(n + ".weight", {C});
        declare(n + ".bias", {C});
    }
    int alloc_device() {
        FC_TRY(dev_alloc(reinterpret_cast<void**>(&raw), (size_t)(raw_numel ? raw_numel : 4) * sizeof(float), "params.raw"));
        FC_TRY(dev_alloc(reinterpret_cast<void**>(&packed), (size_t)(packed_numel ? packed_numel : 4) * sizeof(float), "params.packed"));
        FC_HIP(hipMemset(raw, 0, (size_t)(raw_numel ? raw_numel : 4) * sizeof(float)));
        std::vector<PackJob> jobs;
        for (const PackOp& o : packops) jobs.push_back({raw + o.src, packed + o.dst, o.kind, o.a, o.b, o.kind == 0 ? o.c * o.d : o.c, o.d, o.e, 0});
        return pack_table_build(jobs, &pack_table);
    }
    void free_device() {
        pack_table.release();
        if (raw) dev_free(raw);
        if (packed) dev_free(packed);
        raw = packed = nullptr;
    }
    int run_pack(hipStream_t s) const {
        if (pack_table.nblocks) return pack_table_launch(pack_table, s);
        for (const PackOp& o : packops) {
            const float* src = raw + o.src;
            float* dst = packed + o.dst;
            switch (o.kind) {
                case 0: FC_TRY(pack_conv_launch(src, dst, o.a, o.b, o.c, o.d, s)); break;
                case 1: FC_TRY(pack_s2d_conv_launch(src, dst, o.a, o.b, s)); break;
                case 2: FC_TRY(pack_transpose_launch(src, dst, o.a, o.b, o.c, o.d, s)); break;
                case 3: FC_HIP(hipMemcpyAsync(dst, src, (size_t)o.a * sizeof(float), hipMemcpyDeviceToDevice, s)); break;
                case 4: FC_TRY(pack_conv_pad_launch(src, dst, o.a, o.b, o.c, o.d, o.e, s)); break;
                case 6: return fail(FC_E_STATE, "pack: the k-step-quad layout exists in the table launch only");
            }
        }
        return FC_OK;
    }
    // flat fp32 vector in table order (padded layout) -> raw -> packed
    int load(const float* flat, int64_t numel, int on_device, hipStream_t s) {
        if (numel != raw_numel) return fail(FC_E_ARG, "load_params: expected " + std::to_string(raw_numel) + " floats (padded table layout)");
        FC_HIP(hipMemcpyAsync(raw, flat, (size_t)numel * sizeof(float), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
        FC_TRY(run_pack(s));
        if (!on_device) FC_HIP(hipStreamSynchronize(s));  // the host buffer may be freed by the caller on return
        loaded = true;
        return FC_OK;
    }
    int info(int i, const char** name, int64_t shape[4], int64_t* offset) const {
        if (i < 0 || i >= (int)params.size()) return fail(FC_E_ARG, "param_info: index out of range");
        const Param& p = params[i];
        if (name) *name = p.name.c_str();
        if (shape) for (int k = 0; k < 4; ++k) shape[k] = p.shape[k];
        if (offset) *offset = p.offset;
        return FC_OK;
    }
};

struct PlanBuilder {
    Plan* pl = nullptr;
    int B = 0;  // max batch
    int err = FC_OK;
    std::string scope;  // reference module the ops being emitted belong to
    int* fin_err_word = nullptr;   // device word a timed-out fused tail sets (owned by the handle the plan belongs to)
    int conv_prec = 0;             // ConvArgs::prec of every convolution this builder emits (codec plans: 1 = split-bf16 on request)
    const ParamStore* store = nullptr;   // where conv() finds the split-bf16 copy of a weight (codec builders set it)

    // guard: 0 always | 1 only when the call has a mask | 2 only when it runs mask_fusion_conv | 3 only when it has NO mask | 4 mask but no fusion
    int guard = 0;
    // flops: what the launch executes per sample (per-launch tables, rooflines); flops_ref (> 0): what the reference's arithmetic for the same
    // module is, when that differs (folded upsampling) -- Plan::flops, the per-sample figure of SURVEY 8(d), adds up the reference's
    void push(Op op, const std::string& kernel, double flops = 0.0, double bytes_ps = 0.0, double bytes_fixed = 0.0, double flops_ref = 0.0) {
        if (guard) {
            const int g = guard;
            Op inner = std::move(op);
            op = [inner, g](const FwdCtx& c, hipStream_t s) -> int {
                const bool on = g == 1 ? c.mask != nullptr : g == 2 ? c.mask_fuse != 0 : g == 3 ? c.mask == nullptr : (c.mask != nullptr && !c.mask_fuse);
                return on ? inner(c, s) : FC_OK;
            };
        }
        pl->ops.push_back(std::move(op));
        pl->op_kernel.push_back(kernel);
        pl->op_what.push_back(scope);
        pl->op_flops.push_back(flops);
        pl->op_bytes_ps.push_back(bytes_ps);
        pl->op_bytes_fixed.push_back(bytes_fixed);
        pl->flops += flops_ref > 0.0 ? flops_ref : flops;
        pl->flops_executed += flops;
    }
    float* dmalloc(size_t floats) {
        void* p = nullptr;
        if (dev_alloc(&p, (floats ? floats : 1) * sizeof(float), scope.c_str()) != FC_OK) { err = fail(FC_E_HIP, "hipMalloc failed while reserving the arena"); return nullptr; }
        pl->allocs.push_back(p);
        return static_cast<float*>(p);
    }
    std::multimap<size_t, float*> pool;   // buffers handed back by release(): later tensors of the same size reuse them
    Act act(int C, int H, int W) {
        Act a; a.C = C; a.H = H; a.W = W;
        const size_t n = (size_t)B * H * W * C;
        auto it = pool.find(n);
        if (it != pool.end()) { a.p = it->second; pool.erase(it); }
        else a.p = dmalloc(n);
        return a;
    }
    // The plan is a fixed sequence on one stream, so a buffer whose last reader has been emitted can serve a later tensor.
    void release(const Act& a) { if (a.p) pool.emplace((size_t)B * a.H * a.W * a.C, a.p); }
    Stat stat(int G, int T, float n_t) { Stat s; s.G = G; s.T = T; s.n_t = n_t; s.p = dmalloc((size_t)B * G * T * 2); return s; }

    static SrcXform xf_of(const Stat& st, int mode, const float* gamma, const float* beta, const float* ss = nullptr, int ss_stride = 0,
                          float eps = 1e-5f) {
        SrcXform x;
        x.mode = mode; x.stats = st.p; x.G = st.G; x.T = st.T; x.n_t = st.n_t;
        x.gamma = gamma; x.beta = beta; x.ss = ss; x.ss_stride = ss_stride; x.eps = eps;
        return x;
    }

    // Algorithmic HBM bytes of one convolution launch: every input / output / residual element once per sample, the
    // weights once per launch -- what a kernel with perfect on-chip reuse would move.
    static double conv_bytes_ps(const ConvArgs& a) {
        double e = (double)a.Hs * a.Ws * a.Cin + (double)a.H * a.W * a.Cout;
        if (a.res_out) e += (double)a.H * a.W * a.Cout;
        if (a.add) e += (double)a.H * a.W * a.Cout;
        if (a.fin.res) e += (double)a.H * a.W * a.Cout;
        if (a.fin.raw) e += (double)a.H * a.W * a.Cout;
        if (a.w_batch_stride) e += (double)a.KS * a.KS * a.Cin * a.Cout;
        return 4.0 * e;
    }
    static double conv_bytes_fixed(const ConvArgs& a) {
        double e = a.w_batch_stride ? 0.0 : (double)a.KS * a.KS * a.Cin * a.Cout;
        if (a.res_w) e += (double)a.Cin * a.Cout;
        return 4.0 * e;
    }

    // Emits one implicit-GEMM launch (plus a standalone statistics pass when the output has < 16 pixels per sample).
    // `want_G` > 0 asks for GroupNorm partials of the output; returns them in *st.
    void conv(ConvArgs a, const Act& out, int want_G, Stat* st) {
        if (err) return;
        a.B = B; a.H = out.H; a.W = out.W; a.Cout = out.C; a.out = out.p;
        a.Cin = a.s0.C + a.s1.C;
        a.prec = conv_prec;
        if (conv_prec && store && !a.w_batch_stride) a.w_b3 = store->B3(a.w);
        const bool fused = want_G > 0 && (out.H * out.W) % 16 == 0;
        ConvGeom g;
        if (fused) { a.Gout = want_G; a.stats_out = reinterpret_cast<float*>(1); }  // placeholder: geometry only
        if ((err = conv_plan(a, TILE_AUTO, &g)) != FC_OK) return;
        if (fused) { *st = stat(want_G, g.T, g.n_t); a.stats_out = st->p; }
        const int tile = g.tile;
        double fl = 2.0 * out.H * out.W * a.KS * a.KS * (double)a.Cin * a.Cout;
        if (a.res_out) fl += 2.0 * out.H * out.W * (double)a.Cin * a.Cout;
        push([a, tile](const FwdCtx& c, hipStream_t s) { ConvArgs b = a; b.B = c.B; return conv_launch(b, tile, s); }, kTileNames[tile], fl,
             conv_bytes_ps(a), conv_bytes_fixed(a));
        if (fused && st->T > 16) {   // many tiles per sample (large images): fold the partials once instead of in every consumer workgroup
            const Stat raw = *st;
            *st = stat(want_G, 1, raw.n_t * (float)raw.T);
            const float* ip = raw.p; float* op = st->p;
            const int G = want_G, T = raw.T; const float nt = raw.n_t;
            push([=](const FwdCtx& c, hipStream_t s) { return gn_fold_launch(ip, op, c.B, G, T, nt, s); }, "gn_fold");
        }
        if (want_G > 0 && !fused) {
            *st = stat(want_G, 1, (float)(out.H * out.W * (out.C / want_G)));
            float* sp = st->p; const float* xp = out.p; const int HW = out.H * out.W, C = out.C, G = want_G;
            push([=](const FwdCtx& c, hipStream_t s) { return gn_stats_launch(xp, sp, c.B, HW, C, G, s); }, "gn_stats");
        }
    }

    // nn.Upsample(nearest, x2) + 3x3 convolution as four 2x2 convolutions on the low-resolution input, one per output parity, with the taps
    // that fall on the same source pixel summed at pack time (ParamStore::decl_conv_up2): 16 instead of 36 multiply-adds per 2x2 output
    // block and channel pair -- exact in real arithmetic, a different summation order in fp32.  One launch carries the four classes
    // (ConvArgs::par4).  `w4` = PUP(name); x low resolution, out 2x.  Returns false, emitting nothing, when the shape is not covered
    // (the caller then emits the 3x3 convolution over the upsampled window).
    bool conv_up2(const ConvSrc& src, const Act& x, const float* w4, const float* bias, const Act& out, int want_G, Stat* st) {
        if (err || !w4) return false;
        static const bool no_fold = [] { const char* e = std::getenv("FLOCODER_AMD_UPS_FOLD"); return e && std::string(e) == "0"; }();
        if (no_fold || (want_G > 0 && (x.H * x.W) % 16 != 0)) return false;
        ConvArgs a;
        a.s0 = src; a.Hs = x.H; a.Ws = x.W; a.KS = 2; a.pad = 0; a.stride = 1;
        a.B = B; a.H = x.H; a.W = x.W; a.Cout = out.C; a.out = out.p; a.bias = bias;
        a.Cin = a.s0.C;
        a.prec = conv_prec;
        a.w = w4;
        if (conv_prec && store) a.w_b3 = store->B3(w4);
        a.par4 = 1; a.out_sh = 1; a.stats_tmul = 4; a.pad_y = 1; a.pad_x = 1;
        const bool fused = want_G > 0;
        ConvGeom g;
        if (fused) { a.Gout = want_G; a.stats_out = reinterpret_cast<float*>(1); }
        if (conv_plan(a, TILE_AUTO, &g) != FC_OK || !g.pipe) return false;
        if (fused) { *st = stat(want_G, 4 * g.T, g.n_t); a.stats_out = st->p; }
        const int tile = g.tile;
        const double fl = 2.0 * out.H * out.W * 4.0 * (double)a.Cin * a.Cout;      // four taps per OUTPUT pixel
        push([a, tile](const FwdCtx& c, hipStream_t s) { ConvArgs b = a; b.B = c.B; return conv_launch(b, tile, s); }, kTileNames[tile], fl,
             4.0 * ((double)x.H * x.W * a.Cin + (double)out.H * out.W * a.Cout), 4.0 * 16.0 * a.Cin * a.Cout,
             2.0 * out.H * out.W * 9.0 * (double)a.Cin * a.Cout);
        if (fused && st->T > 16) {
            const Stat raw = *st;
            *st = stat(want_G, 1, raw.n_t * (float)raw.T);
            const float* ip = raw.p; float* op = st->p;
            const int G = want_G, T = raw.T; const float nt = raw.n_t;
            push([=](const FwdCtx& c, hipStream_t s) { return gn_fold_launch(ip, op, c.B, G, T, nt, s); }, "gn_fold");
        }
        return true;
    }

    // A convolution whose epilogue finishes the Block: out = SiLU(GroupNorm(conv)) + res, the GroupNorm statistics exchanged between
    // the workgroups of a sample through a counter (ConvFin).  Returns false, emitting nothing, when the launch cannot keep its whole
    // grid resident or the shape is outside the fused tail's conditions -- the caller then emits conv + finalize.
    bool conv_fin(ConvArgs a, const Act& out, int G, const float* gamma, const float* beta, const float* res, bool want_gn1, Stat* gn1,
                  bool only_local = false, const Act* raw = nullptr, Stat* st_out = nullptr) {
        if (err) return false;
        a.B = B; a.H = out.H; a.W = out.W; a.Cout = out.C; a.out = out.p;
        a.Cin = a.s0.C + a.s1.C;
        if ((out.H * out.W) % 16) return false;
        a.Gout = G; a.stats_out = reinterpret_cast<float*>(16);
        a.fin.gamma = gamma; a.fin.beta = beta; a.fin.res = res;
        if (want_gn1) a.fin.gn1_out = reinterpret_cast<float*>(16);   // placeholders: the geometry (and the occupancy query behind the
        if (raw) a.fin.raw = reinterpret_cast<float*>(16);            // residency check) must see the flavour the launch will use
        ConvGeom g;
        if (conv_plan(a, TILE_AUTO, &g) != FC_OK || !g.pipe) return false;
        if (only_local && !g.fin_local) return false;    // the cross-workgroup meeting costs what the finalize launch costs; the local form is free
        Stat st = stat(G, g.T, g.n_t);
        a.stats_out = st.p;
        if (raw) { a.fin.raw = raw->p; if (st_out) *st_out = st; }
        if (want_gn1) { *gn1 = stat(1, g.T1, g.n_t1); a.fin.gn1_out = gn1->p; }
        unsigned* sync = reinterpret_cast<unsigned*>(dmalloc((size_t)g.groups + 1));
        if (err) return false;
        if (hipMemset(sync, 0, ((size_t)g.groups + 1) * sizeof(unsigned)) != hipSuccess) { err = fail(FC_E_HIP, "hipMemset failed"); return false; }
        const size_t ngran = (size_t)B * G * g.T * 2;                       // one 8-byte {epoch, value} granule per partial statistic
        a.fin.gran = reinterpret_cast<unsigned long long*>(dmalloc(2 * ngran));
        if (err) return false;
        if (hipMemset(a.fin.gran, 0, ngran * sizeof(unsigned long long)) != hipSuccess) { err = fail(FC_E_HIP, "hipMemset failed"); return false; }   // tag 0 = never a live epoch
        a.fin.sync = sync;
        a.fin.err = fin_err_word ? fin_err_word : reinterpret_cast<int*>(sync + g.groups);   // the handle's error word (one per object)
        if (!g.fin_local) { ++pl->n_meet; pl->fin_sync.push_back(sync); pl->fin_kind.push_back(0); }
        const int tile = g.tile;
        const double fl = 2.0 * out.H * out.W * a.KS * a.KS * (double)a.Cin * a.Cout;
        push([a, tile](const FwdCtx& c, hipStream_t s) { ConvArgs b = a; b.B = c.B; return conv_launch(b, tile, s); }, std::string(kTileNames[tile]) + "+fin", fl,
             conv_bytes_ps(a), conv_bytes_fixed(a));
        return true;
    }

    // Single-head softmax attention over the h*w tokens of x with d = C (diffusers' VAE mid-block attention, VQGAN's AttnBlock):
    //   out = proj(softmax(q k^T / sqrt(C)) v) + x,   q/k/v = 1x1 conv of GroupNorm(x)
    // as three 1x1 convs with the norm in their loader, a batched transpose, two per-sample-weight GEMMs on the conv kernel and a
    // one-pass row softmax.  w* are packed [Cin][Cout]; want_G > 0 also returns GroupNorm partials of the sum.
    struct AttnWeights { const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo; };
    Act attention_block(const Act& x, const SrcXform& norm_x, const AttnWeights& w, int want_G, Stat* so) {
        const int C = x.C, hw = x.H * x.W;
        Act q = act(C, x.H, x.W), k = act(C, x.H, x.W), vv = act(C, x.H, x.W);
        const float* ws[3] = {w.wq, w.wk, w.wv};
        const float* bs[3] = {w.bq, w.bk, w.bv};
        const Act* outs[3] = {&q, &k, &vv};
        for (int i = 0; i < 3; ++i) {
            ConvArgs a;
            a.s0.p = x.p; a.s0.C = C; a.s0.xf = norm_x;
            a.Hs = x.H; a.Ws = x.W; a.KS = 1;
            a.w = ws[i]; a.bias = bs[i];
            conv(a, *outs[i], 0, nullptr);
        }
        float* kt = dmalloc((size_t)B * C * hw);                    // k^T per sample: [C][n]
        const float* kp = k.p;
        push([=](const FwdCtx& c, hipStream_t s) { return transpose_batched_launch(kp, kt, c.B, hw, C, s); }, "transpose");
        Act sc = act(hw, x.H, x.W);                                 // scores [B][n][n]
        {
            ConvArgs a;
            a.s0.p = q.p; a.s0.C = C; a.Hs = x.H; a.Ws = x.W; a.KS = 1;
            a.w = kt; a.w_batch_stride = C * hw;
            conv(a, sc, 0, nullptr);
        }
        float* sp = sc.p;
        const float scale = 1.0f / sqrtf((float)C);
        push([=](const FwdCtx& c, hipStream_t s) { return softmax_rows_launch(sp, (long)c.B * hw, hw, scale, s); }, "softmax_rows");
        Act o = act(C, x.H, x.W);
        {
            ConvArgs a;
            a.s0.p = sc.p; a.s0.C = hw; a.Hs = x.H; a.Ws = x.W; a.KS = 1;
            a.w = vv.p; a.w_batch_stride = hw * C;
            conv(a, o, 0, nullptr);
        }
        Act out = act(C, x.H, x.W);
        {
            ConvArgs a;
            a.s0.p = o.p; a.s0.C = C; a.Hs = x.H; a.Ws = x.W; a.KS = 1;
            a.w = w.wo; a.bias = w.bo;
            a.add = x.p; a.stats_post = 1;
            conv(a, out, want_G, so);
        }
        release(q); release(k); release(vv); release(sc); release(o);
        return out;
    }
};

inline int run_plan(const Plan& pl, const FwdCtx& c, hipStream_t s) {
    for (const Op& op : pl.ops) FC_TRY(op(c, s));
    return FC_OK;
}

// Measurement: every launch of `pl` timed alone -- `repeats` back-to-back launches between two HIP events on `s` (ops only read their
// inputs, so repeating one is idempotent).  ms_out[i] = average milliseconds of op i.  Synchronises.
inline int profile_plan(const Plan& pl, const FwdCtx& c, int repeats, float* ms_out, int n_out, hipStream_t s) {
    const int n = (int)pl.ops.size();
    if (n_out < n) return fail(FC_E_ARG, "profile_ops: output array too small");
    FC_TRY(run_plan(pl, c, s));   // warm: every buffer holds finite data
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

}  // namespace fc
