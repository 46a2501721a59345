// Inpainting conditioning (flocoder/inpainting.py:161-253): MaskEncoder and mask_blending.
//
// MaskEncoder turns a 1x128x128 pixel mask into a 4x8x8 latent-shaped conditioning signal with five tiny, oddly shaped
// convolutions (1/17/33 channels, 4x4 stride 4).  That is ~3 MFLOP per sample -- launch-bound, not MFMA work -- so it
// runs as direct convolutions on the reference's NCHW layout: one thread per output element, weights read through the
// scalar/L1 path, activation fused, results written into a channel slice so torch.cat never materialises.
#include <memory>

#include "plan.h"

using namespace fc;

namespace fc {

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_SIGMOID = 2 };

// out[b][co_off + co][y][x] = act(bias[co] + sum_{ci,ky,kx} w[co][ci][ky][kx] * in[b][ci][y*s - p + ky][x*s - p + kx])
__global__ void __launch_bounds__(256) direct_conv_kernel(const float* in, const float* w, const float* bias, float* out, int B, int Cin, int Hin,
                                                          int Win, int Cout, int Ho, int Wo, int KS, int stride, int pad, int act, int CoutTot,
                                                          int co_off) {
    const size_t total = (size_t)B * Cout * Ho * Wo;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % Wo);
        size_t r = i / Wo;
        const int y = (int)(r % Ho); r /= Ho;
        const int co = (int)(r % Cout), b = (int)(r / Cout);
        float acc = bias ? bias[co] : 0.f;
        const float* wb = w + (size_t)co * Cin * KS * KS;
        for (int ci = 0; ci < Cin; ++ci) {
            const float* ip = in + ((size_t)b * Cin + ci) * Hin * Win;
            for (int ky = 0; ky < KS; ++ky) {
                const int iy = y * stride - pad + ky;
                if (iy < 0 || iy >= Hin) continue;
                for (int kx = 0; kx < KS; ++kx) {
                    const int ix = x * stride - pad + kx;
                    if (ix >= 0 && ix < Win) acc += wb[(ci * KS + ky) * KS + kx] * ip[(size_t)iy * Win + ix];
                }
            }
        }
        if (act == ACT_SILU) acc = acc / (1.0f + __expf(-acc));
        else if (act == ACT_SIGMOID) acc = 1.0f / (1.0f + __expf(-acc));
        out[(((size_t)b * CoutTot + co_off + co) * Ho + y) * Wo + x] = acc;
    }
}

// AvgPool2d(k, stride k) of channel `ci` of `in` into channel `co_off` of `out`
__global__ void __launch_bounds__(256) avgpool_kernel(const float* in, float* out, int B, int Cin, int ci, int Hin, int Win, int k, int CoutTot, int co_off) {
    const int Ho = Hin / k, Wo = Win / k;
    const size_t total = (size_t)B * Ho * Wo;
    const float inv = 1.0f / (float)(k * k);
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), b = (int)(i / ((size_t)Wo * Ho));
        const float* ip = in + ((size_t)b * Cin + ci) * Hin * Win;
        float s = 0.f;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) s += ip[(size_t)(y * k + dy) * Win + x * k + dx];
        out[(((size_t)b * CoutTot + co_off) * Ho + y) * Wo + x] = s * inv;
    }
}

// source + mask * (noise - source)   (inpainting.py:250-253)
__global__ void __launch_bounds__(256) mask_blend_kernel(const float* src, const float* mask, const float* noise, float* out, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = __fadd_rn(src[i], __fmul_rn(mask[i], __fsub_rn(noise[i], src[i])));
}

// ---- backward of the direct convolutions (MaskEncoder training, train_flow.py:312-318,361-369): ~10 MFLOP per sample, one thread
// per gradient element, every sum in a fixed order ----
// dz[b][c][p] = dy[b][dy_off + c][p] * act'(.)   with SiLU' from the recomputed pre-activation z, sigmoid' from the output y
__global__ void __launch_bounds__(256) dact_kernel(const float* dy, int dyC, int dy_off, const float* zy, int zC, int z_off, float* dz, int B, int C,
                                                   int HW, int act) {
    const size_t total = (size_t)B * C * HW;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int p = (int)(i % HW), c = (int)((i / HW) % C), b = (int)(i / ((size_t)HW * C));
        const float g = dy[((size_t)b * dyC + dy_off + c) * HW + p], v = zy[((size_t)b * zC + z_off + c) * HW + p];
        float d = 1.f;
        if (act == ACT_SILU) { const float sg = 1.0f / (1.0f + __expf(-v)); d = sg * (1.0f + v * (1.0f - sg)); }
        else if (act == ACT_SIGMOID) d = v * (1.0f - v);
        dz[i] = g * d;
    }
}
// dW[co][ci][ky][kx] (+)= sum_{b,y,x} dz[b][co][y][x] * in[b][ci][y*s-p+ky][x*s-p+kx];  db[co] (+)= sum dz
__global__ void __launch_bounds__(256) direct_wgrad_kernel(const float* in, const float* dz, float* dw, float* db, int B, int Cin, int Hin, int Win,
                                                           int Cout, int Ho, int Wo, int KS, int stride, int pad, int accumulate) {
    const int total = Cout * Cin * KS * KS;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < total + Cout; t += gridDim.x * 256) {
        float acc = 0.f;
        if (t < total) {
            const int kx = t % KS, ky = (t / KS) % KS, ci = (t / (KS * KS)) % Cin, co = t / (KS * KS * Cin);
            for (int b = 0; b < B; ++b) {
                const float* ip = in + ((size_t)b * Cin + ci) * Hin * Win;
                const float* gp = dz + ((size_t)b * Cout + co) * Ho * Wo;
                for (int y = 0; y < Ho; ++y) {
                    const int iy = y * stride - pad + ky;
                    if (iy < 0 || iy >= Hin) continue;
                    for (int x = 0; x < Wo; ++x) {
                        const int ix = x * stride - pad + kx;
                        if (ix >= 0 && ix < Win) acc += gp[y * Wo + x] * ip[(size_t)iy * Win + ix];
                    }
                }
            }
            dw[t] = accumulate ? dw[t] + acc : acc;
        } else {
            const int co = t - total;
            for (int b = 0; b < B; ++b) {
                const float* gp = dz + ((size_t)b * Cout + co) * Ho * Wo;
                for (int i = 0; i < Ho * Wo; ++i) acc += gp[i];
            }
            db[co] = accumulate ? db[co] + acc : acc;
        }
    }
}
// din[b][ci][iy][ix] = sum_{co,ky,kx} w[co][ci][ky][kx] * dz[b][co][(iy+p-ky)/s][(ix+p-kx)/s]   (where the division is exact)
__global__ void __launch_bounds__(256) direct_dgrad_kernel(const float* dz, const float* w, float* din, int B, int Cin, int Hin, int Win, int Cout,
                                                           int Ho, int Wo, int KS, int stride, int pad) {
    const size_t total = (size_t)B * Cin * Hin * Win;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ix = (int)(i % Win);
        size_t r = i / Win;
        const int iy = (int)(r % Hin); r /= Hin;
        const int ci = (int)(r % Cin), b = (int)(r / Cin);
        float acc = 0.f;
        for (int ky = 0; ky < KS; ++ky) {
            const int ty = iy + pad - ky;
            if (ty < 0 || ty % stride) continue;
            const int y = ty / stride;
            if (y >= Ho) continue;
            for (int kx = 0; kx < KS; ++kx) {
                const int tx = ix + pad - kx;
                if (tx < 0 || tx % stride) continue;
                const int x = tx / stride;
                if (x >= Wo) continue;
                for (int co = 0; co < Cout; ++co)
                    acc += w[((size_t)(co * Cin + ci) * KS + ky) * KS + kx] * dz[(((size_t)b * Cout + co) * Ho + y) * Wo + x];
            }
        }
        din[i] = acc;
    }
}

static int grid1(size_t total) { size_t g = (total + 255) / 256; return (int)(g < 4096 ? (g ? g : 1) : 4096); }

static int dconv(const float* in, const float* w, const float* bias, float* out, int B, int Cin, int Hin, int Win, int Cout, int KS, int stride, int pad,
                 int act, int CoutTot, int co_off, hipStream_t s) {
    const int Ho = (Hin + 2 * pad - KS) / stride + 1, Wo = (Win + 2 * pad - KS) / stride + 1;
    hipLaunchKernelGGL(direct_conv_kernel, dim3(grid1((size_t)B * Cout * Ho * Wo)), dim3(256), 0, s, in, w, bias, out, B, Cin, Hin, Win, Cout, Ho, Wo,
                       KS, stride, pad, act, CoutTot, co_off);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
static int apool(const float* in, float* out, int B, int Cin, int ci, int Hin, int Win, int k, int CoutTot, int co_off, hipStream_t s) {
    hipLaunchKernelGGL(avgpool_kernel, dim3(grid1((size_t)B * (Hin / k) * (Win / k))), dim3(256), 0, s, in, out, B, Cin, ci, Hin, Win, k, CoutTot, co_off);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc

// MaskEncoder(output_channels=4, shrink_fac=4, mode='pool', final_act=sigmoid)  (inpainting.py:161-245)
struct fc_mask_encoder : fc::ParamStore {
    int device = 0, out_ch = 4, shrink = 4;
    int B = 0, H = 0, W = 0;
    float *t1 = nullptr, *s1 = nullptr, *t2 = nullptr, *s2 = nullptr;   // conv1 outputs and [skip | learned] stages
    float *zb = nullptr, *ga = nullptr, *gb = nullptr;                   // backward scratch: recomputed pre-activation, two gradient buffers
};

extern "C" {

int fc_mask_encoder_create(int device, fc_mask_encoder** out) {
    if (!out) return fail(FC_E_ARG, "fc_mask_encoder_create: null argument");
    std::unique_ptr<fc_mask_encoder> m(new fc_mask_encoder);
    m->device = device;
    const int k = m->shrink;
    m->declare("layers.0.conv1.weight", {16, 1, k, k});  m->declare("layers.0.conv1.bias", {16});
    m->declare("layers.0.conv2.weight", {16, 16, 3, 3}); m->declare("layers.0.conv2.bias", {16});
    m->declare("layers.1.conv1.weight", {32, 17, k, k}); m->declare("layers.1.conv1.bias", {32});
    m->declare("layers.1.conv2.weight", {32, 32, 3, 3}); m->declare("layers.1.conv2.bias", {32});
    m->declare("layers.2.weight", {m->out_ch - 1, 33, 1, 1}); m->declare("layers.2.bias", {m->out_ch - 1});
    if (device < 0) { *out = m.release(); return FC_OK; }
    FC_TRY(fc_check_device(device));
    FC_HIP(hipSetDevice(device));
    FC_TRY(m->alloc_device());
    *out = m.release();
    return FC_OK;
}

static void me_free_buffers(fc_mask_encoder* m) {
    for (float** p : {&m->t1, &m->s1, &m->t2, &m->s2, &m->zb, &m->ga, &m->gb}) { if (*p) dev_free(*p); *p = nullptr; }
    m->B = 0;
}

void fc_mask_encoder_destroy(fc_mask_encoder* m) {
    if (!m) return;
    if (m->device >= 0) { (void)hipSetDevice(m->device); (void)hipDeviceSynchronize(); me_free_buffers(m); m->free_device(); }
    delete m;
}
int fc_mask_encoder_param_count(const fc_mask_encoder* m) { return m ? (int)m->params.size() : 0; }
int64_t fc_mask_encoder_param_numel(const fc_mask_encoder* m) { return m ? m->raw_numel : 0; }
int fc_mask_encoder_param_info(const fc_mask_encoder* m, int i, const char** name, int64_t shape[4], int64_t* offset) {
    if (!m) return fail(FC_E_ARG, "fc_mask_encoder_param_info: null handle");
    return m->info(i, name, shape, offset);
}
int fc_mask_encoder_load_params(fc_mask_encoder* m, const float* flat, int64_t numel, int on_device, void* stream) {
    if (!m || !flat || m->device < 0) return fail(FC_E_ARG, "fc_mask_encoder_load_params: bad argument");
    FC_HIP(hipSetDevice(m->device));
    return m->load(flat, numel, on_device, static_cast<hipStream_t>(stream));
}
int fc_mask_encoder_reserve(fc_mask_encoder* m, int max_batch, int height, int width) {
    if (!m || m->device < 0 || max_batch < 1) return fail(FC_E_ARG, "fc_mask_encoder_reserve: bad argument");
    const int k = m->shrink;
    if (height % (k * k) || width % (k * k)) return fail(FC_E_SHAPE, "mask encoder: mask height/width must be multiples of shrink_fac^2");
    if (m->B >= max_batch && m->H == height && m->W == width) return FC_OK;
    FC_HIP(hipSetDevice(m->device));
    FC_HIP(hipDeviceSynchronize());
    me_free_buffers(m);
    const size_t h1 = height / k, w1 = width / k, h2 = h1 / k, w2 = w1 / k, b = max_batch;
    FC_TRY(dev_alloc(reinterpret_cast<void**>(&m->t1), b * 16 * h1 * w1 * sizeof(float), "mask_encoder.t1"));
    FC_TRY(dev_alloc(reinterpret_cast<void**>(&m->s1), b * 17 * h1 * w1 * sizeof(float), "mask_encoder.s1"));
    FC_TRY(dev_alloc(reinterpret_cast<void**>(&m->t2), b * 32 * h2 * w2 * sizeof(float), "mask_encoder.t2"));
    FC_TRY(dev_alloc(reinterpret_cast<void**>(&m->s2), b * 33 * h2 * w2 * sizeof(float), "mask_encoder.s2"));
    m->B = max_batch; m->H = height; m->W = width;
    return FC_OK;
}

// mask_pixels_dev [B,1,H,W] fp32 (0/1) -> mask_latents_dev [B,4,H/16,W/16]: channel 0 = 16x average-pooled raw mask, channels 1-3
// learned + sigmoid  (inpainting.py:235-245)
int fc_mask_encoder_forward(fc_mask_encoder* m, const float* mask_pixels_dev, float* mask_latents_dev, int batch, int height, int width, void* stream) {
    if (!m || !mask_pixels_dev || !mask_latents_dev) return fail(FC_E_ARG, "fc_mask_encoder_forward: null argument");
    if (!m->loaded) return fail(FC_E_STATE, "mask encoder: weights not loaded");
    if (m->B < batch || m->H != height || m->W != width) return fail(FC_E_STATE, "mask encoder: call fc_mask_encoder_reserve first");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int k = m->shrink, B = batch, h1 = height / k, w1 = width / k, h2 = h1 / k, w2 = w1 / k, oc = m->out_ch;
    // DownsampleBlock 0: [avgpool4(mask) | silu(conv3x3(silu(conv4x4/s4(mask))))] -> 17 channels
    FC_TRY(dconv(mask_pixels_dev, m->R("layers.0.conv1.weight"), m->R("layers.0.conv1.bias"), m->t1, B, 1, height, width, 16, k, k, 0, ACT_SILU, 16, 0, s));
    FC_TRY(apool(mask_pixels_dev, m->s1, B, 1, 0, height, width, k, 17, 0, s));
    FC_TRY(dconv(m->t1, m->R("layers.0.conv2.weight"), m->R("layers.0.conv2.bias"), m->s1, B, 16, h1, w1, 16, 3, 1, 1, ACT_SILU, 17, 1, s));
    // DownsampleBlock 1 on the 17-channel stage (its channel 0 is the pooled mask) -> 33 channels
    FC_TRY(dconv(m->s1, m->R("layers.1.conv1.weight"), m->R("layers.1.conv1.bias"), m->t2, B, 17, h1, w1, 32, k, k, 0, ACT_SILU, 32, 0, s));
    FC_TRY(apool(m->s1, m->s2, B, 17, 0, h1, w1, k, 33, 0, s));
    FC_TRY(dconv(m->t2, m->R("layers.1.conv2.weight"), m->R("layers.1.conv2.bias"), m->s2, B, 32, h2, w2, 32, 3, 1, 1, ACT_SILU, 33, 1, s));
    // 1x1 conv 33 -> 3 + sigmoid into channels 1..3; channel 0 = AvgPool2d(16) of the raw mask
    FC_TRY(dconv(m->s2, m->R("layers.2.weight"), m->R("layers.2.bias"), mask_latents_dev, B, 33, h2, w2, oc - 1, 1, 1, 0, ACT_SIGMOID, oc, 1, s));
    return apool(mask_pixels_dev, mask_latents_dev, B, 1, 0, height, width, k * k, oc, 0, s);
}

// Parameter gradients of the LAST fc_mask_encoder_forward(mask_pixels) on this object (same mask again; its output in
// mask_latents_dev) for d(mask_latents) = d_latents_dev: grads_flat_dev in the parameter-table layout, overwritten or (accumulate != 0)
// added to -- the inpainting step sums three passes (the batch's masks, all ones, all zeros; train_flow.py:361-369).
int fc_mask_encoder_backward(fc_mask_encoder* m, const float* mask_pixels_dev, const float* mask_latents_dev, const float* d_latents_dev,
                             float* grads_flat_dev, int64_t numel, int accumulate, int batch, int height, int width, void* stream) {
    if (!m || !mask_pixels_dev || !mask_latents_dev || !d_latents_dev || !grads_flat_dev) return fail(FC_E_ARG, "fc_mask_encoder_backward: null argument");
    if (!m->loaded) return fail(FC_E_STATE, "mask encoder: weights not loaded");
    if (m->B < batch || m->H != height || m->W != width) return fail(FC_E_STATE, "mask encoder: call fc_mask_encoder_reserve first");
    if (numel != m->raw_numel) return fail(FC_E_ARG, "fc_mask_encoder_backward: gradient vector must have " + std::to_string(m->raw_numel) + " floats");
    hipStream_t s = static_cast<hipStream_t>(stream);
    FC_HIP(hipSetDevice(m->device));
    const int k = m->shrink, B = batch, h1 = height / k, w1 = width / k, h2 = h1 / k, w2 = w1 / k, oc = m->out_ch;
    if (!m->zb) {
        const size_t big = (size_t)m->B * 17 * h1 * w1;
        FC_TRY(dev_alloc(reinterpret_cast<void**>(&m->zb), big * sizeof(float), "mask_encoder.zb"));
        FC_TRY(dev_alloc(reinterpret_cast<void**>(&m->ga), big * sizeof(float), "mask_encoder.ga"));
        FC_TRY(dev_alloc(reinterpret_cast<void**>(&m->gb), big * sizeof(float), "mask_encoder.gb"));
    }
    if (!accumulate) FC_HIP(hipMemsetAsync(grads_flat_dev, 0, (size_t)numel * sizeof(float), s));
    auto G = [&](const char* n) { return grads_flat_dev + m->params[m->pidx.at(n)].offset; };
    auto dact = [&](const float* dy, int dyC, int off, const float* zy, int zC, int zoff, float* dz, int C, int HW, int act) -> int {
        hipLaunchKernelGGL(dact_kernel, dim3(grid1((size_t)B * C * HW)), dim3(256), 0, s, dy, dyC, off, zy, zC, zoff, dz, B, C, HW, act);
        FC_HIP(hipGetLastError());
        return FC_OK;
    };
    auto wgrad = [&](const float* in, const float* dz, const char* wn, const char* bn, int Cin, int Hin, int Win, int Cout, int KS, int stride, int pad) -> int {
        const int Ho = (Hin + 2 * pad - KS) / stride + 1, Wo = (Win + 2 * pad - KS) / stride + 1;
        hipLaunchKernelGGL(direct_wgrad_kernel, dim3(grid1((size_t)Cout * Cin * KS * KS + Cout)), dim3(256), 0, s, in, dz, G(wn), G(bn), B, Cin, Hin, Win,
                           Cout, Ho, Wo, KS, stride, pad, 1);
        FC_HIP(hipGetLastError());
        return FC_OK;
    };
    auto dgrad = [&](const float* dz, const char* wn, float* din, int Cin, int Hin, int Win, int Cout, int KS, int stride, int pad) -> int {
        const int Ho = (Hin + 2 * pad - KS) / stride + 1, Wo = (Win + 2 * pad - KS) / stride + 1;
        hipLaunchKernelGGL(direct_dgrad_kernel, dim3(grid1((size_t)B * Cin * Hin * Win)), dim3(256), 0, s, dz, m->R(wn), din, B, Cin, Hin, Win, Cout, Ho,
                           Wo, KS, stride, pad);
        FC_HIP(hipGetLastError());
        return FC_OK;
    };
    // layers.2 (1x1, 33 -> 3) + sigmoid: channels 1..3 of the output
    FC_TRY(dact(d_latents_dev, oc, 1, mask_latents_dev, oc, 1, m->ga, oc - 1, h2 * w2, ACT_SIGMOID));
    FC_TRY(wgrad(m->s2, m->ga, "layers.2.weight", "layers.2.bias", 33, h2, w2, oc - 1, 1, 1, 0));
    FC_TRY(dgrad(m->ga, "layers.2.weight", m->gb, 33, h2, w2, oc - 1, 1, 1, 0));                      // gb = d(s2) [B,33,h2,w2]
    // DownsampleBlock 1: s2[:,1:] = silu(conv2(t2)), t2 = silu(conv1(s1))
    FC_TRY(dconv(m->t2, m->R("layers.1.conv2.weight"), m->R("layers.1.conv2.bias"), m->zb, B, 32, h2, w2, 32, 3, 1, 1, ACT_NONE, 32, 0, s));
    FC_TRY(dact(m->gb, 33, 1, m->zb, 32, 0, m->ga, 32, h2 * w2, ACT_SILU));                           // ga = dz4
    FC_TRY(wgrad(m->t2, m->ga, "layers.1.conv2.weight", "layers.1.conv2.bias", 32, h2, w2, 32, 3, 1, 1));
    FC_TRY(dgrad(m->ga, "layers.1.conv2.weight", m->gb, 32, h2, w2, 32, 3, 1, 1));                    // gb = d(t2)
    FC_TRY(dconv(m->s1, m->R("layers.1.conv1.weight"), m->R("layers.1.conv1.bias"), m->zb, B, 17, h1, w1, 32, k, k, 0, ACT_NONE, 32, 0, s));
    FC_TRY(dact(m->gb, 32, 0, m->zb, 32, 0, m->ga, 32, h2 * w2, ACT_SILU));                           // ga = dz3
    FC_TRY(wgrad(m->s1, m->ga, "layers.1.conv1.weight", "layers.1.conv1.bias", 17, h1, w1, 32, k, k, 0));
    FC_TRY(dgrad(m->ga, "layers.1.conv1.weight", m->gb, 17, h1, w1, 32, k, k, 0));                    // gb = d(s1) [B,17,h1,w1]; channel 0 (pooled mask) has no parameters upstream
    // DownsampleBlock 0: s1[:,1:] = silu(conv2(t1)), t1 = silu(conv1(mask))
    FC_TRY(dconv(m->t1, m->R("layers.0.conv2.weight"), m->R("layers.0.conv2.bias"), m->zb, B, 16, h1, w1, 16, 3, 1, 1, ACT_NONE, 16, 0, s));
    FC_TRY(dact(m->gb, 17, 1, m->zb, 16, 0, m->ga, 16, h1 * w1, ACT_SILU));                           // ga = dz2
    FC_TRY(wgrad(m->t1, m->ga, "layers.0.conv2.weight", "layers.0.conv2.bias", 16, h1, w1, 16, 3, 1, 1));
    FC_TRY(dgrad(m->ga, "layers.0.conv2.weight", m->gb, 16, h1, w1, 16, 3, 1, 1));                    // gb = d(t1)
    FC_TRY(dconv(mask_pixels_dev, m->R("layers.0.conv1.weight"), m->R("layers.0.conv1.bias"), m->zb, B, 1, height, width, 16, k, k, 0, ACT_NONE, 16, 0, s));
    FC_TRY(dact(m->gb, 16, 0, m->zb, 16, 0, m->ga, 16, h1 * w1, ACT_SILU));                           // ga = dz1
    return wgrad(mask_pixels_dev, m->ga, "layers.0.conv1.weight", "layers.0.conv1.bias", 1, height, width, 16, k, k, 0);
}

int fc_mask_blend(const float* source_dev, const float* mask_dev, const float* noise_dev, float* out_dev, int64_t numel, void* stream) {
    if (!source_dev || !mask_dev || !noise_dev || !out_dev) return fail(FC_E_ARG, "fc_mask_blend: null argument");
    hipLaunchKernelGGL(mask_blend_kernel, dim3(grid1((size_t)numel)), dim3(256), 0, static_cast<hipStream_t>(stream), source_dev, mask_dev, noise_dev, out_dev,
                       (size_t)numel);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // extern "C"
