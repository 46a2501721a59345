// Weight re-layout from the reference's state_dict tensors to the kernels' operand layouts.  Runs once per
// fc_unet_load_params (and once per optimiser step when training), HBM-bound and tiny.
#include <vector>

#include "common.h"

namespace fc {

// OIHW -> [KH*KW][I][O]
__global__ void __launch_bounds__(256) pack_conv_kernel(const float* src, float* dst, int O, int I, int KK) {
    const size_t total = (size_t)O * I * KK;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int o = (int)(i % O);
        const size_t r = i / O;
        const int ci = (int)(r % I), tap = (int)(r / I);
        dst[i] = src[((size_t)o * I + ci) * KK + tap];
    }
}

// Downsample (unet.py:52-53): Rearrange 'b c (h p1) (w p2) -> b (c p1 p2) h w' followed by a 1x1 conv over 4C
// channels is a 2x2 stride-2 conv over C channels: dst[tap = p1*2+p2][c][o] = src[o][c*4 + p1*2 + p2].
__global__ void __launch_bounds__(256) pack_s2d_kernel(const float* src, float* dst, int O, int C) {
    const size_t total = (size_t)O * C * 4;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int o = (int)(i % O);
        const size_t r = i / O;
        const int c = (int)(r % C), tap = (int)(r / C);
        dst[i] = src[(size_t)o * 4 * C + c * 4 + tap];
    }
}

// src [R][Cc] -> dst[c][dst_col0 + r] with leading dimension dst_ld  (Linear [out][in] -> [in][out], optionally
// into a slice of a wider concatenated matrix)
__global__ void __launch_bounds__(256) pack_transpose_kernel(const float* src, float* dst, int R, int Cc, int dst_ld, int dst_col0) {
    const size_t total = (size_t)R * Cc;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i % R), c = (int)(i / R);
        dst[(size_t)c * dst_ld + dst_col0 + r] = src[(size_t)r * Cc + c];
    }
}

__global__ void __launch_bounds__(256) pack_conv_pad_kernel(const float* src, float* dst, int O, int I, int KK, int Opad, int Ipad) {
    const size_t total = (size_t)Opad * Ipad * KK;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int o = (int)(i % Opad);
        const size_t r = i / Opad;
        const int ci = (int)(r % Ipad), tap = (int)(r / Ipad);
        dst[i] = (o < O && ci < I) ? src[((size_t)o * I + ci) * KK + tap] : 0.f;
    }
}

// Data-gradient operand of a stride-1 convolution: the forward kernel run on dY with
//     dst[(KS-1-ky)*KS + (KS-1-kx)][o][i - ci0] = src[o][i][ky][kx]   for i in [ci0, ci0 + nci)
// ([taps][Cin' = O][Cout' = nci]) and padding KS-1-pad yields dX for input channels ci0..ci0+nci (a slice, so that the two
// sources of a channel concatenation get their gradients from two launches).
__global__ void __launch_bounds__(256) pack_conv_dgrad_kernel(const float* src, float* dst, int O, int I, int KS, int ci0, int nci) {
    const int KK = KS * KS;
    const size_t total = (size_t)O * nci * KK;
    for (size_t t = blockIdx.x * 256ull + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int i = (int)(t % nci);
        const size_t r = t / nci;
        const int o = (int)(r % O), tap = (int)(r / O);
        const int ky = KS - 1 - tap / KS, kx = KS - 1 - tap % KS;
        dst[t] = src[((size_t)o * I + ci0 + i) * KK + ky * KS + kx];
    }
}

// [B][R][Cc] -> [B][Cc][R] through a padded 32x32 LDS tile (coalesced on both sides).   grid (Cc/32, R/32, B), block (32, 8)
__global__ void __launch_bounds__(256) transpose_batched_kernel(const float* src, float* dst, int R, int Cc) {
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * R * Cc;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8)
        if (r0 + j < R && c0 + threadIdx.x < Cc) tile[j][threadIdx.x] = src[base + (size_t)(r0 + j) * Cc + c0 + threadIdx.x];
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8)
        if (c0 + j < Cc && r0 + threadIdx.x < R) dst[base + (size_t)(c0 + j) * R + r0 + threadIdx.x] = tile[threadIdx.x][j];
}

// ---- a whole table of re-layout jobs in ONE launch (a training step re-packs ~190 weight tensors: launch latency, not bytes) ----
// kinds: 0 conv OIHW -> [KK][I][O] (a=O b=I c=KK) | 1 s2d (a=O b=C) | 2 transpose (a=R b=Cc c=ld d=col0) | 3 copy (a=n)
//        4 conv with channel padding (a=O b=I c=KK d=Opad e=Ipad) | 5 data-gradient operand (a=O b=I c=KS d=ci0 e=nci)
//        6 conv in k-step-quad layout (a=O b=I c=KK) | 8 conv as split bf16 (a=O b=I c=KK d=Ipad)
__global__ void __launch_bounds__(256) pack_table_kernel(const PackJob* jobs, const int2* blocks) {
    const int2 bj = blocks[blockIdx.x];          // (job, block index inside the job)
    const PackJob j = jobs[bj.x];
    const float* src = j.src;
    float* dst = j.dst;
    const size_t lo = (size_t)bj.y * kPackPerBlock, hi = lo + kPackPerBlock < j.total ? lo + kPackPerBlock : j.total;
    for (size_t t = lo + threadIdx.x; t < hi; t += 256) {
        switch (j.kind) {
            case 0: { const int o = (int)(t % j.a); const size_t r = t / j.a; const int ci = (int)(r % j.b), tap = (int)(r / j.b);
                      dst[t] = src[((size_t)o * j.b + ci) * j.c + tap]; break; }
            case 1: { const int o = (int)(t % j.a); const size_t r = t / j.a; const int c = (int)(r % j.b), tap = (int)(r / j.b);
                      dst[t] = src[(size_t)o * 4 * j.b + c * 4 + tap]; break; }
            case 2: { const int r = (int)(t % j.a), c = (int)(t / j.a);
                      dst[(size_t)c * j.c + j.d + r] = src[(size_t)r * j.b + c]; break; }
            case 3: dst[t] = src[t]; break;
            case 4: { const int o = (int)(t % j.d); const size_t r = t / j.d; const int ci = (int)(r % j.e), tap = (int)(r / j.e);
                      dst[t] = (o < j.a && ci < j.b) ? src[((size_t)o * j.b + ci) * j.c + tap] : 0.f; break; }
            case 6: {   // conv OIHW -> [tap][I/8][half][O][4]: element (tap, ci = 8 c8 + 2 jj + half, o) -- the MFMA B operand of four consecutive
                        // k-steps of one lane as ONE 16-byte load (conv_pipe.hip, register-fed weights); a=O b=I c=KK
                const int jj = (int)(t & 3); size_t r = t >> 2; const int o = (int)(r % j.a); r /= j.a; const int half = (int)(r & 1); r >>= 1;
                const int c8 = (int)(r % (j.b / 8)), tap = (int)(r / (j.b / 8));
                dst[t] = src[((size_t)o * j.b + 8 * c8 + 2 * jj + half) * j.c + tap]; break; }
            case 8: {   // conv OIHW -> split bf16, two arrays (hi parts, then lo parts) of [tap][Ipad/8][O][8] 16-bit values, Ipad = d (channels
                        // beyond I are zeros): the B operand of v_mfma_f32_32x32x16_bf16 for one column and eight consecutive channels is 16
                        // contiguous bytes, and a 64-column row of one (tap, channel block) is one 1 KiB LDS-DMA piece; a=O b=I c=KK d=Ipad.
                        // t runs over the elements of ONE part; x = hi + lo with hi = bf16(x) (nearest even), lo = bf16(x - hi)
                const int jj = (int)(t & 7); size_t r = t >> 3; const int o = (int)(r % j.a); r /= j.a;
                const int c8 = (int)(r % (j.d / 8)), tap = (int)(r / (j.d / 8)), ci = 8 * c8 + jj;
                const float x = ci < j.b ? src[((size_t)o * j.b + ci) * j.c + tap] : 0.f;
                unsigned u = __float_as_uint(x);
                u += 0x7fffu + ((u >> 16) & 1u);
                const unsigned short hi = (unsigned short)(u >> 16);
                const float rem = x - __uint_as_float((unsigned)hi << 16);
                unsigned v = __float_as_uint(rem);
                v += 0x7fffu + ((v >> 16) & 1u);
                unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
                d16[t] = hi;
                d16[(size_t)j.a * j.d * j.c + t] = (unsigned short)(v >> 16);
                break; }
            case 9: case 10: {
                // nearest x2 upsampling folded into the weights of a 3x3 convolution (SD-VAE Upsample2D): output pixel (2y + a, 2x + b) reads
                // the low-resolution pixels of a 2x2 window whose position depends on its parity (a, b), each through the SUM of the 3x3
                // taps that land on it -- rows {ky} for ty: a = 0: {0}, {1, 2}; a = 1: {0, 1}, {2}; columns likewise.  Four 2x2 kernels,
                // 16 taps instead of 36 per 2x2 output block.  a=O b=I c=9 d=Ipad (kind 10).
                // kind 9: [parity][tap = 2 ty + tx][I][O] fp32; kind 10: per parity the split-bf16 form of kind 8 with KK = 4
                const bool b3 = j.kind == 10;
                const size_t per = b3 ? (size_t)4 * j.d * j.a : (size_t)4 * j.b * j.a;     // elements of one parity (one part)
                const int par = (int)(t / per); const size_t e = t - (size_t)par * per;
                int o, ci, tap;
                if (b3) { const int jj = (int)(e & 7); size_t r = e >> 3; o = (int)(r % j.a); r /= j.a; const int c8 = (int)(r % (j.d / 8)); tap = (int)(r / (j.d / 8)); ci = 8 * c8 + jj; }
                else { o = (int)(e % j.a); const size_t r = e / j.a; ci = (int)(r % j.b); tap = (int)(r / j.b); }
                const int pa = par >> 1, pb = par & 1, ty = tap >> 1, tx = tap & 1;
                const int ky0 = pa == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), ky1 = pa == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
                const int kx0 = pb == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), kx1 = pb == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
                float x = 0.f;
                if (ci < j.b) {
                    const float* w = src + ((size_t)o * j.b + ci) * 9;
                    for (int ky = ky0; ky <= ky1; ++ky)
                        for (int kx = kx0; kx <= kx1; ++kx) x += w[ky * 3 + kx];
                }
                if (!b3) { dst[t] = x; break; }
                unsigned u = __float_as_uint(x);
                u += 0x7fffu + ((u >> 16) & 1u);
                const unsigned short hi = (unsigned short)(u >> 16);
                unsigned v = __float_as_uint(x - __uint_as_float((unsigned)hi << 16));
                v += 0x7fffu + ((v >> 16) & 1u);
                unsigned short* d16 = reinterpret_cast<unsigned short*>(dst) + (size_t)par * 2 * per;
                d16[e] = hi;
                d16[per + e] = (unsigned short)(v >> 16);
                break; }
            default: { const int KK = j.c * j.c; const int i = (int)(t % j.e); const size_t r = t / j.e; const int o = (int)(r % j.a), tap = (int)(r / j.a);
                       const int ky = j.c - 1 - tap / j.c, kx = j.c - 1 - tap % j.c;
                       dst[t] = src[((size_t)o * j.b + j.d + i) * KK + ky * j.c + kx]; break; }
        }
    }
}

size_t pack_job_total(const PackJob& j) {
    switch (j.kind) {
        case 0: return (size_t)j.a * j.b * j.c;
        case 1: return (size_t)j.a * j.b * 4;
        case 2: return (size_t)j.a * j.b;
        case 3: return (size_t)j.a;
        case 4: return (size_t)j.d * j.e * j.c;
        case 6: return (size_t)j.a * j.b * j.c;
        case 8: return (size_t)j.a * j.d * j.c;
        case 9: return (size_t)16 * j.a * j.b;
        case 10: return (size_t)16 * j.a * j.d;
        default: return (size_t)j.a * j.e * j.c * j.c;
    }
}

int pack_table_build(std::vector<PackJob> jobs, PackTable* out) {
    out->release();
    if (jobs.empty()) return FC_OK;
    std::vector<int2> blocks;
    for (size_t i = 0; i < jobs.size(); ++i) {
        jobs[i].total = pack_job_total(jobs[i]);
        const int nb = (int)((jobs[i].total + kPackPerBlock - 1) / kPackPerBlock);
        for (int b = 0; b < nb; ++b) blocks.push_back(make_int2((int)i, b));
    }
    FC_TRY(dev_alloc(reinterpret_cast<void**>(&out->jobs), jobs.size() * sizeof(PackJob), "pack.jobs"));
    FC_TRY(dev_alloc(reinterpret_cast<void**>(&out->blocks), blocks.size() * sizeof(int2), "pack.blocks"));
    FC_HIP(hipMemcpy(out->jobs, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice));
    FC_HIP(hipMemcpy(out->blocks, blocks.data(), blocks.size() * sizeof(int2), hipMemcpyHostToDevice));
    out->nblocks = (int)blocks.size();
    return FC_OK;
}

void PackTable::release() {
    if (jobs) dev_free(jobs);
    if (blocks) dev_free(blocks);
    jobs = nullptr; blocks = nullptr; nblocks = 0;
}

int pack_table_launch(const PackTable& t, hipStream_t s) {
    if (!t.nblocks) return FC_OK;
    hipLaunchKernelGGL(pack_table_kernel, dim3(t.nblocks), dim3(256), 0, s, t.jobs, t.blocks);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

static int pgrid(size_t total) { size_t g = (total + 255) / 256; return (int)(g < 4096 ? (g ? g : 1) : 4096); }

int pack_conv_launch(const float* oihw, float* dst, int O, int I, int KH, int KW, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv_kernel, dim3(pgrid((size_t)O * I * KH * KW)), dim3(256), 0, s, oihw, dst, O, I, KH * KW);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
// OIHW -> [tap][I/8][half][O][4] (the table launch's kind 6, alone: test / debug entry points)
__global__ void __launch_bounds__(256) pack_conv_k8_kernel(const float* src, float* dst, int O, int I, int KK) {
    const size_t total = (size_t)O * I * KK;
    for (size_t t = blockIdx.x * 256ull + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int jj = (int)(t & 3); size_t r = t >> 2; const int o = (int)(r % O); r /= O; const int half = (int)(r & 1); r >>= 1;
        const int c8 = (int)(r % (I / 8)), tap = (int)(r / (I / 8));
        dst[t] = src[((size_t)o * I + 8 * c8 + 2 * jj + half) * KK + tap];
    }
}
int pack_conv_k8_launch(const float* oihw, float* dst, int O, int I, int KK, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv_k8_kernel, dim3(pgrid((size_t)O * I * KK)), dim3(256), 0, s, oihw, dst, O, I, KK);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
__global__ void __launch_bounds__(256) pack_conv_b3_kernel(const float* src, unsigned short* dst, int O, int I, int KK, int Ipad) {
    const size_t total = (size_t)O * Ipad * KK;
    for (size_t t = blockIdx.x * 256ull + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int jj = (int)(t & 7); size_t r = t >> 3; const int o = (int)(r % O); r /= O;
        const int c8 = (int)(r % (Ipad / 8)), tap = (int)(r / (Ipad / 8)), ci = 8 * c8 + jj;
        const float x = ci < I ? src[((size_t)o * I + ci) * KK + tap] : 0.f;
        unsigned u = __float_as_uint(x);
        u += 0x7fffu + ((u >> 16) & 1u);
        const unsigned short hi = (unsigned short)(u >> 16);
        unsigned v = __float_as_uint(x - __uint_as_float((unsigned)hi << 16));
        v += 0x7fffu + ((v >> 16) & 1u);
        dst[t] = hi;
        dst[total + t] = (unsigned short)(v >> 16);
    }
}
int pack_conv_b3_launch(const float* oihw, float* dst, int O, int I, int KK, int Ipad, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv_b3_kernel, dim3(pgrid((size_t)O * Ipad * KK)), dim3(256), 0, s, oihw, reinterpret_cast<unsigned short*>(dst), O, I, KK, Ipad);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
int pack_conv_pad_launch(const float* oihw, float* dst, int O, int I, int KK, int Opad, int Ipad, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv_pad_kernel, dim3(pgrid((size_t)Opad * Ipad * KK)), dim3(256), 0, s, oihw, dst, O, I, KK, Opad, Ipad);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
int pack_conv_dgrad_launch(const float* oihw, float* dst, int O, int I, int KS, int ci0, int nci, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv_dgrad_kernel, dim3(pgrid((size_t)O * nci * KS * KS)), dim3(256), 0, s, oihw, dst, O, I, KS, ci0, nci);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
int transpose_batched_launch(const float* src, float* dst, int B, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32), B), dim3(32, 8), 0, s, src, dst, rows, cols);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
int pack_s2d_conv_launch(const float* oi, float* dst, int O, int C, hipStream_t s) {
    hipLaunchKernelGGL(pack_s2d_kernel, dim3(pgrid((size_t)O * C * 4)), dim3(256), 0, s, oi, dst, O, C);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
int pack_transpose_launch(const float* src, float* dst, int R, int Cc, int dst_ld, int dst_col0, hipStream_t s) {
    hipLaunchKernelGGL(pack_transpose_kernel, dim3(pgrid((size_t)R * Cc)), dim3(256), 0, s, src, dst, R, Cc, dst_ld, dst_col0);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc
