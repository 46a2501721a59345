// 2-D neighbourhood attention, the one native (CUDA) dependency of the reference: NATTENBlock (codecs.py:93-145) calls the
// third-party `natten` package (pyproject.toml:40, natten>=0.20.1; absent here: PARITY UNPINNED).  What is restated is the package's
// published definition of na2d (kernel_size k, dilation 1, no relative position bias, non-causal):
//     every query (x, y) attends to the k x k keys of the window  [sx, sx + k) x [sy, sy + k),   s = clamp(pos - k/2, 0, L - k)
// -- near a border the window SHIFTS inwards instead of being zero-padded, so every query sees exactly k*k keys -- with
//     out = softmax(q . K_window * scale) V_window,        scale = head_dim^-0.5 (the package's default; the reference's own
//     `self.scaling` attribute is never passed on, codecs.py:99,130-135).
//
// The kernel is layout-agnostic: q / k / v are read through (x, y, head, d) strides of ONE fused qkv tensor, because the reference
// hands na2d tensors shaped [B, heads, H, W, d] while natten >= 0.20 documents [B, X, Y, heads, d]: mode 1 is the
// intended reading (X, Y = image rows / columns, 8 heads), mode 2 what such a package version computes from the reference's call
// (X = the head index, Y = image rows, "heads" = image columns).  Both run here; which one a checkpoint was trained under is the
// loader's choice (VQVAE(natten_layout=...)).
//
// One wave per query: lane w < k*k owns one key of the window (dot product over d in registers), softmax by shuffles, then every lane
// owns output channels d = lane, lane + 64.  HBM / cache-bound gather, no MFMA: at 7x7 the arithmetic is 2 * 49 * d per query.
#include "common.h"

namespace fc {

struct NaArgs {
    const float* qkv;      // fused projection output; q at +0, k at +koff, v at +voff (floats)
    float* out;            // same indexing as q (x, y, head, d strides of `out`)
    const float* gamma;    // optional scalar: out *= gamma[0] (NATTENBlock's learned residual gate, folded before the linear proj)
    long koff, voff;
    long sb, sx, sy, sh;   // strides of qkv in floats: batch, x, y, head (d is contiguous)
    long ob, ox, oy, oh;   // strides of out
    int B, X, Y, NH, D, K;
    float scale;
};

__global__ void __launch_bounds__(256) na2d_kernel(const NaArgs a) {
    const int lane = threadIdx.x & 63;
    const long q_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long total = (long)a.B * a.X * a.Y * a.NH;
    if (q_id >= total) return;
    const int h = (int)(q_id % a.NH);
    const int y = (int)((q_id / a.NH) % a.Y);
    const int x = (int)((q_id / ((long)a.NH * a.Y)) % a.X);
    const int b = (int)(q_id / ((long)a.NH * a.Y * a.X));
    const int K = a.K, KK = K * K, r = K / 2;
    int sx = x - r; sx = sx < 0 ? 0 : (sx > a.X - K ? a.X - K : sx);
    int sy = y - r; sy = sy < 0 ? 0 : (sy > a.Y - K ? a.Y - K : sy);
    const float* qp = a.qkv + (long)b * a.sb + (long)x * a.sx + (long)y * a.sy + (long)h * a.sh;
    const bool live = lane < KK;
    const int wx = live ? lane / K : 0, wy = live ? lane % K : 0;
    const float* kp = a.qkv + a.koff + (long)b * a.sb + (long)(sx + wx) * a.sx + (long)(sy + wy) * a.sy + (long)h * a.sh;
    float dot = 0.f;
    for (int d = 0; d < a.D; d += 4) {
        const float4 qv = *reinterpret_cast<const float4*>(qp + d);
        const float4 kv = live ? *reinterpret_cast<const float4*>(kp + d) : make_float4(0.f, 0.f, 0.f, 0.f);
        dot += qv.x * kv.x + qv.y * kv.y + qv.z * kv.z + qv.w * kv.w;
    }
    float s = live ? dot * a.scale : -INFINITY;
    float mx = s;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float p = live ? expf(s - mx) : 0.f;
    float sum = p;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    p /= sum;
    const float g = a.gamma ? a.gamma[0] : 1.0f;
    const float* vb = a.qkv + a.voff + (long)b * a.sb + (long)h * a.sh;
    float* op = a.out + (long)b * a.ob + (long)x * a.ox + (long)y * a.oy + (long)h * a.oh;
    for (int d0 = 0; d0 < a.D; d0 += 64) {
        const int d = d0 + lane;
        float acc = 0.f;
        for (int w = 0; w < KK; ++w) {
            const float pw = __shfl(p, w);
            if (d < a.D) acc += pw * vb[(long)(sx + w / K) * a.sx + (long)(sy + w % K) * a.sy + d];
        }
        if (d < a.D) op[d] = acc * g;
    }
}

// qkv: NHWC [B][H][W][3C] with channel = which*C + head*hd + d (the reference's reshape, codecs.py:122-124); out NHWC [B][H][W][C].
// mode 1: windows over (image row, image column) per head;  mode 2: windows over (head index, image row) per image column.
int na2d_launch(const float* qkv, float* out, const float* gamma, int B, int H, int W, int C, int heads, int ksize, int mode, hipStream_t s) {
    if (C % heads || ((C / heads) & 3)) return fail(FC_E_SHAPE, "na2d: head_dim must be a multiple of 4");
    const int hd = C / heads;
    NaArgs a;
    a.qkv = qkv; a.out = out; a.gamma = gamma; a.koff = C; a.voff = 2L * C;
    a.B = B; a.D = hd; a.K = ksize; a.scale = 1.0f / sqrtf((float)hd);
    a.sb = (long)H * W * 3 * C; a.ob = (long)H * W * C;
    if (mode == 1) {
        a.X = H; a.Y = W; a.NH = heads;
        a.sx = (long)W * 3 * C; a.sy = 3L * C; a.sh = hd;
        a.ox = (long)W * C; a.oy = C; a.oh = hd;
    } else if (mode == 2) {
        a.X = heads; a.Y = H; a.NH = W;
        a.sx = hd; a.sy = (long)W * 3 * C; a.sh = 3L * C;
        a.ox = hd; a.oy = (long)W * C; a.oh = C;
    } else {
        return fail(FC_E_ARG, "na2d: layout mode must be 1 (spatial windows) or 2 (natten >= 0.20 reading of the reference's call)");
    }
    if (a.X < ksize || a.Y < ksize) return fail(FC_E_SHAPE, "na2d: the neighbourhood is larger than the attended axes (natten raises here too)");
    if (ksize * ksize > 64 || !(ksize & 1)) return fail(FC_E_SHAPE, "na2d: odd kernel sizes up to 7 are built");
    const long total = (long)B * a.X * a.Y * a.NH;
    hipLaunchKernelGGL(na2d_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, a);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc

using namespace fc;

// Stand-alone entry point (tests, and callers that bring their own projections): see include/flocoder_amd.h
extern "C" int fc_na2d(const float* qkv_nhwc_dev, float* out_nhwc_dev, const float* gamma_dev, int batch, int height, int width, int channels,
                       int heads, int kernel_size, int layout_mode, void* stream) {
    if (!qkv_nhwc_dev || !out_nhwc_dev || batch < 1) return fail(FC_E_ARG, "fc_na2d: null argument");
    return na2d_launch(qkv_nhwc_dev, out_nhwc_dev, gamma_dev, batch, height, width, channels, heads, kernel_size, layout_mode,
                       static_cast<hipStream_t>(stream));
}
