// One-workgroup-per-sample U-Net forward for models whose activations fit a CU's LDS (unet_sample.hip): the program a workgroup walks.
#pragma once
#include "common.h"

namespace fc {

constexpr int SAMPLE_THREADS = 256;       // threads of a sample's workgroup

enum SampleOp { S_CONV = 0, S_NORM = 1, S_BILINEAR = 2, S_LINATTN = 3, S_ATTN = 4, S_COPY = 5, S_ATTN1 = 6, S_LINATTN_W = 7, S_LINATTN_G = 8 };   // S_ATTN1: either attention on ONE position; S_LINATTN_W: LinearAttention, a wave per head

struct SStep {
    int op = 0;
    int guard = 0;            // 0 always | 1 only with a mask | 2 only when mask_fusion_conv runs | 3 only without a mask | 4 mask, no fusion | 5 no fusion
    int in0 = 0, in1 = -1;    // LDS float offsets of the NHWC source(s); in1: second half of a channel concat
    int out = 0;              // LDS float offset of the NHWC result
    int res = -1;             // LDS float offset of a tensor of the result's shape added last, or -1
    int scratch = 0;          // attention: LDS float offset of its work area
    int C0 = 0, C1 = 0, Cout = 0, Hi = 0, Wi = 0, Ho = 0, Wo = 0;
    int KS = 1, pad = 0, stride = 1, ups = 0;
    int act = 0;              // SiLU on the result (before `res`)
    int G = 1, ss_off = -1;   // GroupNorm groups; column of the block's FiLM scale in the sample's conditioning row (shift at + C), or -1
    float eps = 1e-5f;
    // convolution: the weight rows it reads, in chunks of <= 4096 floats (only the taps some output pixel can reach; the host's list)
    int nchunk = 0;
    int crow[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // first row (tap * Cin + ci) of chunk k
    int cn[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // rows in chunk k
    int lco = 0;              // log2(Cout) (convolution)
    int lks = 0;              // convolution: log2 of the lanes that share one output quad (each takes every (1 << lks)-th channel quad of a tap)
    int fast = 0, nqi = 0;    // convolution: fast = taps of the ONE chunk (1 or 9) when the unrolled multiply-add applies, nqi = channel quads per lane per tap
    int fnorm = 0;            // convolution: 1 = GroupNorm (G, gamma, beta, ss_off, eps) + SiLU (`act`) + `res` applied to the result in the epilogue
    int full = 0;             // S_ATTN1: 1 = Attention (no to_out norm), 0 = LinearAttention
    int lc = 0, lcpg = 0;     // log2(C0), log2(channels per GroupNorm group) (norm, attention)
    int ln = 0;               // log2(pixels) (attention)
    const float* w = nullptr;      // conv: packed [tap][Cin][Cout]; attention: to_qkv [C][384]
    const float* bias = nullptr;
    const float* gamma = nullptr;  // norm weight / bias (attention: fn.norm)
    const float* beta = nullptr;
    const float* w2 = nullptr;     // attention: to_out [128][C]
    const float* b2 = nullptr;     // attention: to_out bias
    const float* g2 = nullptr;     // linear attention: to_out.1 norm weight / bias
    const float* be2 = nullptr;
};

struct SampleArgs {
    const SStep* prog = nullptr;
    int nsteps = 0;
    const float* x = nullptr;      // NCHW [x_mod][ch][HW]
    int x_mod = 1;
    const float* mask = nullptr;   // NCHW like x, or null
    int mask_fuse = 0;
    const float* ss_all = nullptr; // conditioning rows of every evaluation [evaluation][rows][S] (CondFetch) or null ...
    const int* evalc = nullptr;    // ... with the evaluation counter
    const float* ss = nullptr;     // ... else this forward's rows [rows][S]
    int rows = 0, S = 0;
    float* out = nullptr;          // NCHW [B][ch][HW]
    int ch = 0, HW = 0;
    int x_off = 0, mask_off = 0, v_off = 0;    // LDS float offsets: input, mask, velocity (all NHWC)
    int zero_off = 0;                          // LDS float offset of 128 zeros (what a tap outside the image reads)
    int wbuf_off = 0, prog_off = 0;            // LDS float offsets: the weight staging buffers (2 x 4096 floats), the program's copy
    EulerTail euler;               // integrator: Euler update instead of `out`, counters moved by the last workgroup
    unsigned* done = nullptr;      // arrival counter of that hand-over (zero between launches)
    unsigned long long* stamps = nullptr;   // diagnostics (fc_debug_set_conv_stamps): workgroup 0 records {100 MHz clock, step code} per step
};

int unet_sample_init();
int unet_sample_launch(const SampleArgs& a, int B, size_t lds_bytes, hipStream_t s);

}  // namespace fc
