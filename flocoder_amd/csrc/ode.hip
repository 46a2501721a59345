// ODE integrator state updates (sampling.py:36-48 rk4_step, :69-74 CFG blend; legacy Euler
// train_sd_flowers.py:58-64).  The state lives in the reference's NCHW boundary layout; every op is a
// single rounded fp32 operation in the reference's order (no FMA contraction), so the only differences
// against the CPU path come from the U-Net itself.
#include "common.h"

namespace fc {

__device__ __forceinline__ float mul_(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_(float a, float b) { return __fsub_rn(a, b); }

// v_no_class + cfg * (v - v_no_class), sampling.py:74
__device__ __forceinline__ float cfg_blend(float vc, float vn, float cfg) { return add_(vn, mul_(cfg, sub_(vc, vn))); }

__device__ __forceinline__ float4 load_v(const float* v2, int i, int n, int cfg_on, float cfg) {
    float4 v = *reinterpret_cast<const float4*>(v2 + i);
    if (cfg_on) {
        const float4 u = *reinterpret_cast<const float4*>(v2 + n + i);
        v.x = cfg_blend(v.x, u.x, cfg); v.y = cfg_blend(v.y, u.y, cfg);
        v.z = cfg_blend(v.z, u.z, cfg); v.w = cfg_blend(v.w, u.w, cfg);
    }
    return v;
}

// one block
__global__ void __launch_bounds__(256) ode_time_kernel(int* step, const float* ts, float t_scale, int rk4, float* sc, float* tvec,
                                                       int rows) {
    const int i = *step;
    const float t = ts[i];
    __syncthreads();   // everyone has read the counter before it moves
    if (threadIdx.x == 0) {
        sc[0] = t;
        sc[1] = rk4 ? sub_(ts[i + 1], t) : 0.f;   // dt = ts[i+1] - ts[i], sampling.py:117
        *step = i + 1;
    }
    const float tv = mul_(t, t_scale);            // t_vec * t_scale, sampling.py:60-63
    for (int r = threadIdx.x; r < rows; r += 256) tvec[r] = tv;
}

__global__ void __launch_bounds__(256) ode_euler_update_kernel(float* x, const float* v2, int n, int cfg_on, float cfg, float dt) {
    for (int i = 4 * (blockIdx.x * 256 + threadIdx.x); i < n; i += 4 * gridDim.x * 256) {
        const float4 v = load_v(v2, i, n, cfg_on, cfg);
        float4 xv = *reinterpret_cast<float4*>(x + i);
        xv.x = add_(xv.x, mul_(v.x, dt)); xv.y = add_(xv.y, mul_(v.y, dt));    // x + pred * dt
        xv.z = add_(xv.z, mul_(v.z, dt)); xv.w = add_(xv.w, mul_(v.w, dt));
        *reinterpret_cast<float4*>(x + i) = xv;
    }
}

__global__ void __launch_bounds__(256) ode_rk4_stage_kernel(const float* sc, const float* y, float* xs, float* k_out, const float* v2,
                                                            int n, int cfg_on, float cfg, int full, int tsel, float t_scale,
                                                            float* tvec, int rows) {
    const float t = sc[0], dt = sc[1];
    if (blockIdx.x == 0) {
        const float tn = tsel == 1 ? add_(t, dt * 0.5f) : add_(t, dt);     // t + dt/2 | t + dt
        const float tv = mul_(tn, t_scale);
        for (int r = threadIdx.x; r < rows; r += 256) tvec[r] = tv;
    }
    for (int i = 4 * (blockIdx.x * 256 + threadIdx.x); i < n; i += 4 * gridDim.x * 256) {
        const float4 k = load_v(v2, i, n, cfg_on, cfg);
        *reinterpret_cast<float4*>(k_out + i) = k;
        const float4 yv = *reinterpret_cast<const float4*>(y + i);
        float4 o;
        if (full) {   // y + dt*k3
            o.x = add_(yv.x, mul_(dt, k.x)); o.y = add_(yv.y, mul_(dt, k.y));
            o.z = add_(yv.z, mul_(dt, k.z)); o.w = add_(yv.w, mul_(dt, k.w));
        } else {      // y + dt*k/2
            o.x = add_(yv.x, mul_(dt, k.x) * 0.5f); o.y = add_(yv.y, mul_(dt, k.y) * 0.5f);
            o.z = add_(yv.z, mul_(dt, k.z) * 0.5f); o.w = add_(yv.w, mul_(dt, k.w) * 0.5f);
        }
        *reinterpret_cast<float4*>(xs + i) = o;
    }
}

__device__ __forceinline__ float rk4_comb(float y, float k1, float k2, float k3, float k4, float dt6) {
    // y + (dt/6)*(k1 + 2*k2 + 2*k3 + k4), left to right
    const float s = add_(add_(add_(k1, 2.0f * k2), 2.0f * k3), k4);
    return add_(y, mul_(dt6, s));
}

__global__ void __launch_bounds__(256) ode_rk4_final_kernel(const float* sc, float* y, const float* k1, const float* k2, const float* k3,
                                                            const float* v2, int n, int cfg_on, float cfg) {
    const float dt6 = __fdiv_rn(sc[1], 6.0f);
    for (int i = 4 * (blockIdx.x * 256 + threadIdx.x); i < n; i += 4 * gridDim.x * 256) {
        const float4 k4 = load_v(v2, i, n, cfg_on, cfg);
        const float4 a = *reinterpret_cast<const float4*>(k1 + i), b = *reinterpret_cast<const float4*>(k2 + i),
                     c = *reinterpret_cast<const float4*>(k3 + i);
        float4 yv = *reinterpret_cast<float4*>(y + i);
        yv.x = rk4_comb(yv.x, a.x, b.x, c.x, k4.x, dt6); yv.y = rk4_comb(yv.y, a.y, b.y, c.y, k4.y, dt6);
        yv.z = rk4_comb(yv.z, a.z, b.z, c.z, k4.z, dt6); yv.w = rk4_comb(yv.w, a.w, b.w, c.w, k4.w, dt6);
        *reinterpret_cast<float4*>(y + i) = yv;
    }
}

// Scaled time of EVERY evaluation of an integration, with the very operations the per-step kernels above use (so a conditioning table
// built from it is bit-identical to what the per-forward launches would compute): Euler: ts[i] * t_scale; RK4, interval i:
// t, t + dt/2, t + dt/2, t + dt with dt = ts[i+1] - ts[i].
__global__ void __launch_bounds__(256) ode_all_times_kernel(const float* ts, int n_steps, int rk4, float t_scale, float* tv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_steps) return;
    const float t = ts[i];
    if (!rk4) { tv[i] = mul_(t, t_scale); return; }
    const float dt = sub_(ts[i + 1], t);
    const float th = mul_(add_(t, dt * 0.5f), t_scale);
    tv[4 * i] = mul_(t, t_scale);
    tv[4 * i + 1] = th;
    tv[4 * i + 2] = th;
    tv[4 * i + 3] = mul_(add_(t, dt), t_scale);
}
int ode_all_times_launch(const float* ts, int n_steps, int rk4, float t_scale, float* tv, hipStream_t s) {
    hipLaunchKernelGGL(ode_all_times_kernel, dim3(cdiv(n_steps, 256)), dim3(256), 0, s, ts, n_steps, rk4, t_scale, tv);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

static int egrid(int n) { int g = (n / 4 + 255) / 256; return g < 1 ? 1 : (g > 2048 ? 2048 : g); }

int ode_time_launch(int* step, const float* ts, float t_scale, int rk4, float* sc, float* tvec, int rows, hipStream_t s) {
    hipLaunchKernelGGL(ode_time_kernel, dim3(1), dim3(256), 0, s, step, ts, t_scale, rk4, sc, tvec, rows);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
int ode_euler_update_launch(float* x, const float* v2, int n, int cfg_on, float cfg, float dt, hipStream_t s) {
    if (n & 3) return fail(FC_E_SHAPE, "ode: element count must be a multiple of 4");
    hipLaunchKernelGGL(ode_euler_update_kernel, dim3(egrid(n)), dim3(256), 0, s, x, v2, n, cfg_on, cfg, dt);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
int ode_rk4_stage_launch(const float* sc, const float* y, float* xs, float* k_out, const float* v2, int n, int cfg_on, float cfg, int full,
                         int tsel, float t_scale, float* tvec, int rows, hipStream_t s) {
    if (n & 3) return fail(FC_E_SHAPE, "ode: element count must be a multiple of 4");
    hipLaunchKernelGGL(ode_rk4_stage_kernel, dim3(egrid(n)), dim3(256), 0, s, sc, y, xs, k_out, v2, n, cfg_on, cfg, full, tsel, t_scale,
                       tvec, rows);
    FC_HIP(hipGetLastError());
    return FC_OK;
}
int ode_rk4_final_launch(const float* sc, float* y, const float* k1, const float* k2, const float* k3, const float* v2, int n, int cfg_on,
                         float cfg, hipStream_t s) {
    if (n & 3) return fail(FC_E_SHAPE, "ode: element count must be a multiple of 4");
    hipLaunchKernelGGL(ode_rk4_final_kernel, dim3(egrid(n)), dim3(256), 0, s, sc, y, k1, k2, k3, v2, n, cfg_on, cfg);
    FC_HIP(hipGetLastError());
    return FC_OK;
}

}  // namespace fc
