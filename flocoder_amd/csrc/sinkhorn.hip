// Debiased Sinkhorn divergence between two point clouds -- the parity metric of flocoder/metrics.py:40-54
// (`SamplesLoss("sinkhorn", p=2, blur=0.05)` of the third-party geomloss package, absent here: PARITY UNPINNED).
//
// What is restated is geomloss's published tensorized algorithm (sinkhorn_divergence.py / sinkhorn_samples.py, v0.2.6):
//   cost        C(x, y) = |x - y|^2 / 2                                   (p = 2)
//   weights     uniform, a_i = 1/N, b_j = 1/M
//   schedule    eps_0 = diameter^2, then exp(arange(2 log diameter, 2 log blur, 2 log scaling)), then blur^2
//               diameter = | max over both clouds - min over both clouds |  (per coordinate, then the Euclidean norm)
//   potentials  initialised by one softmin at eps_0, then per eps the symmetrised updates
//                  ft_ba = softmin(eps, C_xy, log b + g_ab / eps)     gt_ab = softmin(eps, C_yx, log a + f_ba / eps)
//                  ft_aa = softmin(eps, C_xx, log a + f_aa / eps)     gt_bb = softmin(eps, C_yy, log b + g_bb / eps)
//                  f <- (f + ft) / 2 for all four; one last un-averaged update at the final eps ("extrapolation")
//               softmin(eps, C, h)_i = -eps * log sum_j exp(h_j - C_ij / eps)
//   value       S = <a, f_ba - f_aa> + <b, g_ab - g_bb>
//
// Device design: the four N x M cost matrices are formed ONCE (direct differences, fp64 accumulation -- geomloss uses the
// |x|^2 - 2xy + |y|^2 form in fp32, which cancels badly at D ~ 2e5 pixels); every eps step is one launch in which a wave owns a row of one of
// the four softmins (log-sum-exp in fp64: potentials reach diameter^2 ~ 1e5 while eps falls to 2.5e-3, far outside fp32's reach).
// The loop is ~20 tiny launches: latency, not bandwidth.  Deterministic: fixed reduction orders, no atomics.
#include <cmath>

#include "common.h"

namespace fc {

constexpr int SK_T = 16, SK_K = 32;

// C[i][j] = 0.5 * sum_k (x[i][k] - y[j][k])^2 ; grid (ceil(M/16), ceil(N/16)), 256 threads
__global__ void __launch_bounds__(256) sk_cost_kernel(const float* x, const float* y, int N, int M, long D, double* C) {
    __shared__ float sa[SK_T][SK_K + 1], sb[SK_T][SK_K + 1];
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    const int i0 = blockIdx.y * SK_T, j0 = blockIdx.x * SK_T;
    double acc = 0.0;
    for (long k0 = 0; k0 < D; k0 += SK_K) {
        for (int e = threadIdx.x; e < SK_T * SK_K; e += 256) {
            const int rr = e / SK_K, kk = e % SK_K;
            const bool kin = k0 + kk < D;
            sa[rr][kk] = (kin && i0 + rr < N) ? x[(size_t)(i0 + rr) * D + k0 + kk] : 0.f;
            sb[rr][kk] = (kin && j0 + rr < M) ? y[(size_t)(j0 + rr) * D + k0 + kk] : 0.f;
        }
        __syncthreads();
        float part = 0.f;   // 32 terms in fp32, then into the fp64 running sum: exact enough (relative 1e-7 per block), 4x cheaper than all-fp64
#pragma unroll
        for (int kk = 0; kk < SK_K; ++kk) { const float d = sa[r][kk] - sb[c][kk]; part += d * d; }
        acc += (double)part;
        __syncthreads();
    }
    if (i0 + r < N && j0 + c < M) C[(size_t)(i0 + r) * M + j0 + c] = 0.5 * acc;
}

// per-coordinate min / max over the rows of both clouds -> squared diameter partials; grid ceil(D/256)
__global__ void __launch_bounds__(256) sk_extent_kernel(const float* x, const float* y, int N, int M, long D, double* part) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    double sq = 0.0;
    if (k < D) {
        float lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < N; ++i) { const float v = x[(size_t)i * D + k]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
        for (int j = 0; j < M; ++j) { const float v = y[(size_t)j * D + k]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
        const double ext = (double)hi - (double)lo;
        sq = ext * ext;
    }
    __shared__ double red[256];
    red[threadIdx.x] = sq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

struct SkStep {
    const double *Cxy, *Cxx, *Cyy;     // [N][M], [N][N], [M][M]   (C_yx is C_xy read by columns)
    const double *f_ba, *g_ab, *f_aa, *g_bb;   // current potentials: on x, on y, on x, on y
    double *o_f_ba, *o_g_ab, *o_f_aa, *o_g_bb; // next
    int N, M;
    double eps, loga, logb;
    int average;                        // 1: new = (old + update) / 2 ; 0: new = update
    int init;                           // 1: potentials are zero (first softmin of the schedule)
};

// one wave per row: rows [0,N) f_ba, [N,N+M) g_ab, [N+M,2N+M) f_aa, [2N+M,2N+2M) g_bb
__global__ void __launch_bounds__(256) sk_step_kernel(const SkStep s) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int N = s.N, M = s.M;
    if (row >= 2 * (N + M)) return;
    int which, i, len;
    if (row < N) { which = 0; i = row; len = M; }
    else if (row < N + M) { which = 1; i = row - N; len = N; }
    else if (row < 2 * N + M) { which = 2; i = row - N - M; len = N; }
    else { which = 3; i = row - 2 * N - M; len = M; }
    const double* pot = which == 0 ? s.g_ab : which == 1 ? s.f_ba : which == 2 ? s.f_aa : s.g_bb;   // the potential on the OTHER cloud
    const double logw = (which == 0 || which == 3) ? s.logb : s.loga;
    const double inv = 1.0 / s.eps;
    auto cost = [&](int j) -> double {
        switch (which) {
            case 0: return s.Cxy[(size_t)i * M + j];
            case 1: return s.Cxy[(size_t)j * M + i];     // C_yx[i][j] = C_xy[j][i]
            case 2: return s.Cxx[(size_t)i * N + j];
            default: return s.Cyy[(size_t)i * M + j];
        }
    };
    // two passes: max, then sum of exp -- each lane strides the row
    double mx = -INFINITY;
    for (int j = lane; j < len; j += 64) {
        const double h = logw + (s.init ? 0.0 : pot[j] * inv) - cost(j) * inv;
        mx = fmax(mx, h);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    double sum = 0.0;
    for (int j = lane; j < len; j += 64) {
        const double h = logw + (s.init ? 0.0 : pot[j] * inv) - cost(j) * inv;
        sum += exp(h - mx);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) {
        const double upd = -s.eps * (mx + log(sum));
        const double* oldp = which == 0 ? s.f_ba : which == 1 ? s.g_ab : which == 2 ? s.f_aa : s.g_bb;
        double* outp = which == 0 ? s.o_f_ba : which == 1 ? s.o_g_ab : which == 2 ? s.o_f_aa : s.o_g_bb;
        outp[i] = s.average ? 0.5 * (oldp[i] + upd) : upd;
    }
}

// S = mean_i (f_ba - f_aa) + mean_j (g_ab - g_bb), one block, fixed order
__global__ void __launch_bounds__(256) sk_value_kernel(const double* f_ba, const double* f_aa, const double* g_ab, const double* g_bb, int N, int M,
                                                       double* out) {
    __shared__ double red[256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) a += f_ba[i] - f_aa[i];
    for (int j = threadIdx.x; j < M; j += 256) b += g_ab[j] - g_bb[j];
    red[threadIdx.x] = a / N + b / M;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) *out = red[0];
}

}  // namespace fc

using namespace fc;

extern "C" int fc_sinkhorn_divergence(const float* x_dev, const float* y_dev, int n, int m, int64_t dim, double blur, double scaling,
                                      double diameter, double* value_out_host, double* diameter_out_host, int* iterations_out_host, void* stream) {
    if (!x_dev || !y_dev || !value_out_host || n < 1 || m < 1 || dim < 1) return fail(FC_E_ARG, "fc_sinkhorn_divergence: bad argument");
    if (!(blur > 0.0) || !(scaling > 0.0 && scaling < 1.0)) return fail(FC_E_ARG, "fc_sinkhorn_divergence: need blur > 0 and 0 < scaling < 1");
    if (n > 8192 || m > 8192) return fail(FC_E_SHAPE, "fc_sinkhorn_divergence: at most 8192 points per cloud (cost matrices are kept whole)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nm = (size_t)n * m, nn = (size_t)n * n, mm = (size_t)m * m;
    const int nblk = (int)((dim + 255) / 256);
    double* ws = nullptr;
    const size_t total = nm + nn + mm + 4 * (size_t)(n + m) + (size_t)nblk + 4;
    FC_HIP(hipMalloc(reinterpret_cast<void**>(&ws), total * sizeof(double)));
    struct Free { double* p; ~Free() { (void)hipFree(p); } } guard{ws};
    double *Cxy = ws, *Cxx = Cxy + nm, *Cyy = Cxx + nn, *pot = Cyy + mm;   // pot: two sets of {f_ba[n], g_ab[m], f_aa[n], g_bb[m]}
    double *ext = pot + 4 * (size_t)(n + m), *val = ext + nblk;
    if (!(diameter > 0.0)) {
        hipLaunchKernelGGL(sk_extent_kernel, dim3(nblk), dim3(256), 0, s, x_dev, y_dev, n, m, (long)dim, ext);
        FC_HIP(hipGetLastError());
        std::vector<double> part(nblk);
        FC_HIP(hipMemcpyAsync(part.data(), ext, nblk * sizeof(double), hipMemcpyDeviceToHost, s));
        FC_HIP(hipStreamSynchronize(s));
        double sq = 0.0;
        for (double v : part) sq += v;
        diameter = std::sqrt(sq);
    }
    if (diameter_out_host) *diameter_out_host = diameter;
    if (!(diameter > 0.0)) {   // both clouds are one and the same point: the divergence is exactly zero
        *value_out_host = 0.0;
        if (iterations_out_host) *iterations_out_host = 0;
        return FC_OK;
    }
    hipLaunchKernelGGL(sk_cost_kernel, dim3(cdiv(m, SK_T), cdiv(n, SK_T)), dim3(256), 0, s, x_dev, y_dev, n, m, (long)dim, Cxy);
    hipLaunchKernelGGL(sk_cost_kernel, dim3(cdiv(n, SK_T), cdiv(n, SK_T)), dim3(256), 0, s, x_dev, x_dev, n, n, (long)dim, Cxx);
    hipLaunchKernelGGL(sk_cost_kernel, dim3(cdiv(m, SK_T), cdiv(m, SK_T)), dim3(256), 0, s, y_dev, y_dev, m, m, (long)dim, Cyy);
    FC_HIP(hipGetLastError());
    // eps schedule (geomloss epsilon_schedule, p = 2): [diam^2] + exp(arange(2 log diam, 2 log blur, 2 log scaling)) + [blur^2]
    std::vector<double> eps_list{diameter * diameter};
    const double stop = 2.0 * std::log(blur), step = 2.0 * std::log(scaling), start = 2.0 * std::log(diameter);
    if (start > stop) {
        const long cnt = (long)std::ceil((stop - start) / step);          // numpy.arange's length rule
        for (long k = 0; k < cnt; ++k) eps_list.push_back(std::exp(start + (double)k * step));
    }
    eps_list.push_back(blur * blur);
    const int nm2 = n + m;
    double* cur = pot;
    double* nxt = pot + 2 * (size_t)nm2;
    auto run = [&](double eps, int average, int init) -> int {
        SkStep a;
        a.Cxy = Cxy; a.Cxx = Cxx; a.Cyy = Cyy;
        a.f_ba = cur; a.g_ab = cur + n; a.f_aa = cur + nm2; a.g_bb = cur + nm2 + n;
        a.o_f_ba = nxt; a.o_g_ab = nxt + n; a.o_f_aa = nxt + nm2; a.o_g_bb = nxt + nm2 + n;
        a.N = n; a.M = m; a.eps = eps; a.loga = -std::log((double)n); a.logb = -std::log((double)m);
        a.average = average; a.init = init;
        hipLaunchKernelGGL(sk_step_kernel, dim3(cdiv(2 * nm2, 4)), dim3(256), 0, s, a);
        FC_HIP(hipGetLastError());
        std::swap(cur, nxt);
        return FC_OK;
    };
    FC_TRY(run(eps_list[0], 0, 1));                                        // initialisation at the largest eps
    for (double eps : eps_list) FC_TRY(run(eps, 1, 0));                    // eps-scaling descent, symmetrised updates
    FC_TRY(run(eps_list.back(), 0, 0));                                    // last extrapolation
    hipLaunchKernelGGL(sk_value_kernel, dim3(1), dim3(256), 0, s, cur, cur + nm2, cur + n, cur + nm2 + n, n, m, val);
    FC_HIP(hipGetLastError());
    FC_HIP(hipMemcpyAsync(value_out_host, val, sizeof(double), hipMemcpyDeviceToHost, s));
    FC_HIP(hipStreamSynchronize(s));
    if (iterations_out_host) *iterations_out_host = (int)eps_list.size();
    return FC_OK;
}
