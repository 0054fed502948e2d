// Mish activation for gfx950: y = x * tanh(softplus(x)), softplus with torch's threshold of 20
// (reference models/PCWNet/submodule.py:11-18 `Mish`, :178-190 `FMish`; models/CFNet/submodule.py likewise).
// The PCWNet / CFNet family uses it wherever GwcNet uses ReLU.  First version: a separate
// streaming pass after the BatchNorm apply (float4 per lane, HBM-bound: read x, write y; backward reads gy and x,
// writes gx); fusing it into the BN apply / conv epilogue is the obvious next step once the family is measured.
#include "stx_common.h"

namespace {

constexpr int ACT_THREADS = 256;

// (the arithmetic lives in stx_common.h: stx_mish / stx_mish_grad -- n / (n + 2) with n = e^x (e^x + 2), one exp and one
// division instead of libm's log1pf + tanhf; GPU call O -> V: 0.19 -> 0.077 ms for a 212 MB volume)
__device__ __forceinline__ float mish_f(float x) { return stx_mish(x); }
__device__ __forceinline__ float mish_grad_f(float x) { return stx_mish_grad(x); }

__global__ __launch_bounds__(ACT_THREADS) void mish_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              size_t nquads) {
    for (size_t i = (size_t)blockIdx.x * ACT_THREADS + threadIdx.x; i < nquads; i += (size_t)gridDim.x * ACT_THREADS) {
        float4 v = stx_ld4(x + i * 4);
        v.x = mish_f(v.x); v.y = mish_f(v.y); v.z = mish_f(v.z); v.w = mish_f(v.w);
        stx_st4(y + i * 4, v);
    }
}

__global__ __launch_bounds__(ACT_THREADS) void mish_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                              float* __restrict__ gx, size_t nquads) {
    for (size_t i = (size_t)blockIdx.x * ACT_THREADS + threadIdx.x; i < nquads; i += (size_t)gridDim.x * ACT_THREADS) {
        const float4 g = stx_ld4(gy + i * 4), v = stx_ld4(x + i * 4);
        float4 o;
        o.x = g.x * mish_grad_f(v.x); o.y = g.y * mish_grad_f(v.y);
        o.z = g.z * mish_grad_f(v.z); o.w = g.w * mish_grad_f(v.w);
        stx_st4(gx + i * 4, o);
    }
}

int act_grid(size_t nquads) {
    const size_t g = (nquads + ACT_THREADS - 1) / ACT_THREADS;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

// x, y: n floats, n a multiple of 4 (dense channels-last activations always are); in place (y == x) allowed
extern "C" int stx_mish_fwd(const float* x, float* y, long long n, void* stream) {
    stx_begin();
    STX_REQUIRE(x && y && n > 0 && n % 4 == 0, "mish_fwd: need n %% 4 == 0 (n=%lld)", n);
    hipLaunchKernelGGL(mish_fwd_kernel, dim3(act_grid((size_t)n / 4)), dim3(ACT_THREADS), 0, (hipStream_t)stream, x, y,
                       (size_t)n / 4);
    return stx_check_launch("mish_fwd");
}

// gx = gy * mish'(x); x is the activation INPUT (pre-activation value)
extern "C" int stx_mish_bwd(const float* gy, const float* x, float* gx, long long n, void* stream) {
    stx_begin();
    STX_REQUIRE(gy && x && gx && n > 0 && n % 4 == 0, "mish_bwd: need n %% 4 == 0 (n=%lld)", n);
    hipLaunchKernelGGL(mish_bwd_kernel, dim3(act_grid((size_t)n / 4)), dim3(ACT_THREADS), 0, (hipStream_t)stream, gy, x,
                       gx, (size_t)n / 4);
    return stx_check_launch("mish_bwd");
}
