// Attention cores of the velocity U-Net on NHWC qkv tensors [B][n][3*heads*32] (channel = which*heads*32 + head*32 + d,
// the layout to_qkv's 1x1 conv produces; unet.py:110-113,137-140).  dim_head is fixed at 32 (unet.py:100,126).
#include "common.h"

namespace fc {

constexpr int DH = 32;

// ---------------------------------------------------------------------------------------------------
// LinearAttention part 1 (unet.py:143,146): ctx[d][e] = sum_n softmax_n(k)[d][n] * v[e][n].   grid (B*heads)
// Pass 1 finds max_n k[d][n]; pass 2 walks n in tiles of 64 rows staged in LDS (double-buffered, next tile in flight in
// registers), accumulating the 32x32 context (4 entries per thread) and the softmax denominators in registers.
__global__ void __launch_bounds__(256) linattn_ctx_kernel(const float* qkv, float* ctx, int n, int heads) {
    __shared__ float red[8][DH];
    __shared__ __attribute__((aligned(16))) float kmax[DH];
    __shared__ __attribute__((aligned(16))) float ek[2][64][DH];
    __shared__ __attribute__((aligned(16))) float vv[2][64][DH];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x;
    const int C3 = 3 * heads * DH;
    const float* kb = qkv + (size_t)b * n * C3 + heads * DH + h * DH;
    const float* vb = kb + heads * DH;
    {   // pass 1: max over n, 8 independent loads in flight per thread
        const int d = tid & 31, grp = tid >> 5;
        float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
        int i = grp;
        for (; i + 24 < n; i += 32) {
            const float a0 = kb[(size_t)i * C3 + d], a1 = kb[(size_t)(i + 8) * C3 + d];
            const float a2 = kb[(size_t)(i + 16) * C3 + d], a3 = kb[(size_t)(i + 24) * C3 + d];
            m0 = fmaxf(m0, a0); m1 = fmaxf(m1, a1); m2 = fmaxf(m2, a2); m3 = fmaxf(m3, a3);
        }
        for (; i < n; i += 8) m0 = fmaxf(m0, kb[(size_t)i * C3 + d]);
        red[grp][d] = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        __syncthreads();
        if (tid < DH) {
            float mm = red[0][tid];
#pragma unroll
            for (int g = 1; g < 8; ++g) mm = fmaxf(mm, red[g][tid]);
            kmax[tid] = mm;
        }
        __syncthreads();
    }
    // pass 2: tiles of 64 rows, double-buffered: tile t+1 travels HBM -> registers while tile t is consumed from LDS
    const int d = tid >> 3, e0 = (tid & 7) * 4;
    const int r0 = tid >> 3, c4 = (tid & 7) * 4;     // staging role: rows r0 and r0 + 32, float4 column c4
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 km = *reinterpret_cast<const float4*>(&kmax[c4]);
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f, ksum = 0.f;
    float4 kr[2], vr[2];
    auto fetch = [&](int i0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = i0 + r0 + 32 * u;
            kr[u] = zero4; vr[u] = zero4;
            if (r < n) {
                kr[u] = *reinterpret_cast<const float4*>(kb + (size_t)r * C3 + c4);
                vr[u] = *reinterpret_cast<const float4*>(vb + (size_t)r * C3 + c4);
            }
        }
    };
    auto stage = [&](int buf, int i0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = r0 + 32 * u;
            float4 e = zero4;
            if (i0 + r < n) e = make_float4(__expf(kr[u].x - km.x), __expf(kr[u].y - km.y), __expf(kr[u].z - km.z), __expf(kr[u].w - km.w));
            *reinterpret_cast<float4*>(&ek[buf][r][c4]) = e;
            *reinterpret_cast<float4*>(&vv[buf][r][c4]) = vr[u];
        }
    };
    fetch(0);
    stage(0, 0);
    __syncthreads();
    int buf = 0;
    for (int i0 = 0; i0 < n; i0 += 64, buf ^= 1) {
        const bool more = i0 + 64 < n;
        if (more) fetch(i0 + 64);
#pragma unroll 8
        for (int r = 0; r < 64; ++r) {
            const float kd = ek[buf][r][d];
            const float4 v4 = *reinterpret_cast<const float4*>(&vv[buf][r][e0]);
            acc0 += kd * v4.x; acc1 += kd * v4.y; acc2 += kd * v4.z; acc3 += kd * v4.w;
            ksum += kd;
        }
        if (more) stage(buf ^ 1, i0 + 64);
        __syncthreads();
    }
    const float inv = 1.0f / ksum;
    float* o = ctx + ((size_t)blockIdx.x * DH + d) * DH + e0;
    *reinterpret_cast<float4*>(o) = make_float4(acc0 * inv, acc1 * inv, acc2 * inv, acc3 * inv);
}

int linattn_ctx_launch(const float* qkv, float* ctx, int B, int n, int heads, hipStream_t s) {
    hipLaunchKernelGGL(linattn_ctx_kernel, dim3(B * heads), dim3(256), 0, s, qkv, ctx, n, heads);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// ---------------------------------------------------------------------------------------------------
// LinearAttention part 2 (unet.py:142,145,148): out[n][h*32+e] = sum_d ctx[h][d][e] * softmax_d(q[n][h,:])[d] * 32^-1/2.
// grid (ceil(n / PIX), B); one thread per (pixel, head); the sample's contexts sit in LDS.
__global__ void __launch_bounds__(256) linattn_apply_kernel(const float* qkv, const float* ctx, float* out, int n, int heads, int pix_per) {
    extern __shared__ __attribute__((aligned(16))) float cs[];  // [heads][32][32]
    const int b = blockIdx.y, tid = threadIdx.x;
    const int C3 = 3 * heads * DH, CO = heads * DH;
    for (int i = tid; i < heads * DH * DH; i += 256) cs[i] = ctx[(size_t)b * heads * DH * DH + i];
    __syncthreads();
    const float scale = 0.17677669529663687f;  // 32^-0.5
    for (int j = tid; j < pix_per * heads; j += 256) {
        const int h = j / pix_per, pix = blockIdx.x * pix_per + j % pix_per;   // one head per wave: context reads are pure broadcasts
        if (pix >= n) continue;
        const float* qp = qkv + ((size_t)b * n + pix) * C3 + h * DH;
        float q[DH];
#pragma unroll
        for (int i = 0; i < DH; i += 4) {
            const float4 t = *reinterpret_cast<const float4*>(qp + i);
            q[i] = t.x; q[i + 1] = t.y; q[i + 2] = t.z; q[i + 3] = t.w;
        }
        float m = q[0];
#pragma unroll
        for (int i = 1; i < DH; ++i) m = fmaxf(m, q[i]);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < DH; ++i) { q[i] = __expf(q[i] - m); sum += q[i]; }
        const float f = scale / sum;
        const float* ch = cs + h * DH * DH;
        float* op = out + ((size_t)b * n + pix) * CO + h * DH;
#pragma unroll
        for (int e = 0; e < DH; e += 4) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) {
                const float4 c4 = *reinterpret_cast<const float4*>(ch + dd * DH + e);
                acc.x += c4.x * q[dd]; acc.y += c4.y * q[dd]; acc.z += c4.z * q[dd]; acc.w += c4.w * q[dd];
            }
            *reinterpret_cast<float4*>(op + e) = make_float4(acc.x * f, acc.y * f, acc.z * f, acc.w * f);
        }
    }
}

int linattn_apply_launch(const float* qkv, const float* ctx, float* out, int B, int n, int heads, hipStream_t s) {
    const int pix_per = 256 / heads > 0 ? 256 / heads : 1;
    hipLaunchKernelGGL(linattn_apply_kernel, dim3(cdiv(n, pix_per), B), dim3(256), (size_t)heads * DH * DH * sizeof(float), s, qkv, ctx,
                       out, n, heads, pix_per);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// ---------------------------------------------------------------------------------------------------
// Softmax attention for the bottleneck (unet.py:114-121), n <= 64 tokens.   grid (B*heads)
// out[i][h*32+d] = sum_j softmax_j(q_i . k_j * 32^-1/2) v[j][d]
__global__ void __launch_bounds__(256) attn_small_kernel(const float* qkv, float* out, int n, int heads) {
    __shared__ float qs[64][DH + 1], ks[64][DH + 1], vs[64][DH + 1], sim[64][65];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x;
    const int C3 = 3 * heads * DH, CO = heads * DH;
    const float* base = qkv + (size_t)b * n * C3 + h * DH;
    const float scale = 0.17677669529663687f;
    for (int i = tid; i < n * DH; i += 256) {
        const int r = i / DH, c = i % DH;
        qs[r][c] = base[(size_t)r * C3 + c] * scale;
        ks[r][c] = base[(size_t)r * C3 + heads * DH + c];
        vs[r][c] = base[(size_t)r * C3 + 2 * heads * DH + c];
    }
    __syncthreads();
    for (int i = tid; i < n * n; i += 256) {
        const int r = i / n, c = i % n;
        float s = 0.f;
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) s += qs[r][dd] * ks[c][dd];
        sim[r][c] = s;
    }
    __syncthreads();
    for (int r = tid; r < n; r += 256) {
        float m = sim[r][0];
        for (int c = 1; c < n; ++c) m = fmaxf(m, sim[r][c]);
        float sum = 0.f;
        for (int c = 0; c < n; ++c) { const float e = __expf(sim[r][c] - m); sim[r][c] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int c = 0; c < n; ++c) sim[r][c] *= inv;
    }
    __syncthreads();
    for (int i = tid; i < n * DH; i += 256) {
        const int r = i / DH, dd = i % DH;
        float s = 0.f;
        for (int c = 0; c < n; ++c) s += sim[r][c] * vs[c][dd];
        out[((size_t)b * n + r) * CO + h * DH + dd] = s;
    }
}

// Row softmax for the SD-VAE mid-block attention scores (n x n per sample, n <= 4096): one wave per row, the row held in
// registers between the max, the exp-sum and the write-back, so scores make one round trip.   grid (ceil(rows/4)), 256 threads
__global__ void __launch_bounds__(256) softmax_rows_kernel(float* x, long rows, int cols, float scale) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* xr = x + row * cols;
    constexpr int MAXV = 16;                       // 16 float4 per lane = 4096 columns
    float4 v[MAXV];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
            v[i] = *reinterpret_cast<const float4*>(xr + c);
            v[i].x *= scale; v[i].y *= scale; v[i].z *= scale; v[i].w *= scale;
            m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
            v[i].x = __expf(v[i].x - m); v[i].y = __expf(v[i].y - m); v[i].z = __expf(v[i].z - m); v[i].w = __expf(v[i].w - m);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) *reinterpret_cast<float4*>(xr + c) = make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
    }
}

int softmax_rows_launch(float* x, long rows, int cols, float scale, hipStream_t s) {
    if (cols > 4096 || (cols & 3)) return fail(FC_E_SHAPE, "softmax_rows: columns must be a multiple of 4 and <= 4096");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, cols, scale);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

// SpatialNonLocalAttention (codecs.py:337-383): 1x1 q/k projections to Cr = max(1, C/2) channels (padded to even), rotary position
// encoding by token index, softmax(q k^T Cp^-1/2) v with v = 1x1 conv C->C, out_proj, residual.  C is the VQ embedding width (4):
// one thread per query token walks every key with an online softmax; k/v of the sample sit in LDS.   grid (B), 256 threads
constexpr int RA_MAXC = 8;
__global__ void __launch_bounds__(256) rope_attn_kernel(const float* x, const float* wq, const float* bq, const float* wk, const float* bk,
                                                        const float* wv, const float* bv, const float* wo, const float* bo, float* out, int n, int C,
                                                        int Cr) {
    extern __shared__ float sm[];       // k [n][Cp] | v [n][C]
    const int Cp = (Cr + 1) & ~1;
    float* ks = sm;
    float* vs = sm + (size_t)n * Cp;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* xb = x + (size_t)b * n * C;
    const float lscale = logf(10000.0f);
    auto project_rope = [&](const float* w, const float* bias, const float* xi, int pos, float* o) {
        float t[RA_MAXC];
        for (int r = 0; r < Cp; ++r) {
            float a = 0.f;
            if (r < Cr) { a = bias[r]; for (int c = 0; c < C; ++c) a += w[r * C + c] * xi[c]; }
            t[r] = a;
        }
        for (int p2 = 0; p2 < Cp / 2; ++p2) {                   // codecs.py:357-368
            const float ang = (float)pos * expf(-(float)p2 * lscale / (float)(Cp / 2));
            const float sn = sinf(ang), cs = cosf(ang);
            o[2 * p2] = t[2 * p2] * cs - t[2 * p2 + 1] * sn;
            o[2 * p2 + 1] = t[2 * p2 + 1] * cs + t[2 * p2] * sn;
        }
    };
    for (int j = tid; j < n; j += 256) {
        float xi[RA_MAXC];
        for (int c = 0; c < C; ++c) xi[c] = xb[(size_t)j * C + c];
        project_rope(wk, bk, xi, j, ks + (size_t)j * Cp);
        for (int c = 0; c < C; ++c) { float a = bv[c]; for (int d = 0; d < C; ++d) a += wv[c * C + d] * xi[d]; vs[(size_t)j * C + c] = a; }
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)Cp);
    for (int i = tid; i < n; i += 256) {
        float xi[RA_MAXC], q[RA_MAXC], acc[RA_MAXC];
        for (int c = 0; c < C; ++c) { xi[c] = xb[(size_t)i * C + c]; acc[c] = 0.f; }
        project_rope(wq, bq, xi, i, q);
        float m = -INFINITY, l = 0.f;
        for (int j = 0; j < n; ++j) {
            float sdot = 0.f;
            for (int r = 0; r < Cp; ++r) sdot += q[r] * ks[(size_t)j * Cp + r];
            sdot *= scale;
            const float mn = fmaxf(m, sdot), corr = __expf(m - mn), pj = __expf(sdot - mn);
            l = l * corr + pj;
            for (int c = 0; c < C; ++c) acc[c] = acc[c] * corr + pj * vs[(size_t)j * C + c];
            m = mn;
        }
        for (int c = 0; c < C; ++c) acc[c] /= l;
        for (int c = 0; c < C; ++c) {
            float a = bo[c];
            for (int d = 0; d < C; ++d) a += wo[c * C + d] * acc[d];
            out[((size_t)b * n + i) * C + c] = xi[c] + a;
        }
    }
}

int rope_attn_launch(const float* x, const float* wq, const float* bq, const float* wk, const float* bk, const float* wv, const float* bv,
                     const float* wo, const float* bo, float* out, int B, int n, int C, int Cr, hipStream_t s) {
    const int Cp = (Cr + 1) & ~1;
    if (C > RA_MAXC || Cp > RA_MAXC) return fail(FC_E_SHAPE, "rope_attn: more than 8 channels");
    const size_t lds = (size_t)n * (Cp + C) * sizeof(float);
    if (lds > 64 * 1024) return fail(FC_E_SHAPE, "rope_attn: too many tokens");
    hipLaunchKernelGGL(rope_attn_kernel, dim3(B), dim3(256), lds, s, x, wq, bq, wk, bk, wv, bv, wo, bo, out, n, C, Cr);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

int attn_small_launch(const float* qkv, float* out, int B, int n, int heads, hipStream_t s) {
    if (n > 64) return fail(FC_E_SHAPE, "attn_small: more than 64 tokens");
    hipLaunchKernelGGL(attn_small_kernel, dim3(B * heads), dim3(256), 0, s, qkv, out, n, heads);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc
