// Per-pixel, per-group L2 normalisation of an NCHW feature map: the pre-pass of FoundationStereo's group-wise correlation
// (reference models/FoundationStereo/submodule.py:388-397):
//
//     cost[b,g,h,w] = sum_{c in g} normalize(fea1)[b,c,h,w] * normalize(fea2)[b,c,h,w],
//     normalize(x)[b,c,h,w] = x[b,c,h,w] / max(||x[b, g(c)-th group, h, w]||_2, 1e-12)          (F.normalize(dim=2), fp32)
//
// The norm belongs to a pixel of ONE feature map, not to a (pixel, disparity) pair, so the normalised volume
// (FoundationStereo/submodule.py:399-413) is the plain group-wise correlation volume of the two normalised maps with the
// group SUM instead of the mean: stx_cost_volume_fwd(mean) of  out_scale = cpg  on the left map and 1 on the right one.
// The pass is HBM-bound and small (2 x 44 MB at 576x960 for 320 channels; the volume build that follows moves 0.5 GB).
//
// Layout: x, y [B][C][HW], C = G * cpg.  One lane owns one pixel of one (b, g): cpg strided reads (stride HW floats), each a
// fully coalesced 256-byte wave access; the second sweep re-reads the same lines (L2 hits) instead of holding cpg values in
// registers, so any cpg is served by one kernel.
#include "stx_common.h"

namespace {

constexpr int GN_THREADS = 256;
constexpr float GN_EPS = 1e-12f;        // F.normalize default eps

__global__ __launch_bounds__(GN_THREADS) void group_normalize_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                         int cpg, int HW, float out_scale) {
    const int p = blockIdx.x * GN_THREADS + threadIdx.x;
    if (p >= HW) return;
    const size_t base = (size_t)blockIdx.y * cpg * HW + p;      // blockIdx.y = b * G + g
    float ss = 0.f;
    for (int c = 0; c < cpg; ++c) {
        const float v = x[base + (size_t)c * HW];
        ss = fmaf(v, v, ss);
    }
    const float n = sqrtf(ss);
    const float r = out_scale / (n > GN_EPS ? n : GN_EPS);
    for (int c = 0; c < cpg; ++c) y[base + (size_t)c * HW] = x[base + (size_t)c * HW] * r;
}

// y = s x / n, n = max(||x||, eps):   gx = s / n * (gy - xh <xh, gy>),  xh = x / n      (||x|| > eps)
//                                      gx = s / eps * gy                                   (clamped norm: a constant divisor)
__global__ __launch_bounds__(GN_THREADS) void group_normalize_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                         float* __restrict__ gx, int cpg, int HW, float out_scale) {
    const int p = blockIdx.x * GN_THREADS + threadIdx.x;
    if (p >= HW) return;
    const size_t base = (size_t)blockIdx.y * cpg * HW + p;
    float ss = 0.f, dot = 0.f;
    for (int c = 0; c < cpg; ++c) {
        const float v = x[base + (size_t)c * HW];
        ss = fmaf(v, v, ss);
        dot = fmaf(v, gy[base + (size_t)c * HW], dot);
    }
    const float n = sqrtf(ss);
    const bool clamped = !(n > GN_EPS);
    const float r = out_scale / (clamped ? GN_EPS : n);
    const float k = clamped ? 0.f : dot / ss;                    // <xh, gy> / n = <x, gy> / n^2
    for (int c = 0; c < cpg; ++c)
        gx[base + (size_t)c * HW] = r * (gy[base + (size_t)c * HW] - k * x[base + (size_t)c * HW]);
}

int gn_check(const void* a, const void* b, int B, int C, int G, int HW, const char* what) {
    STX_REQUIRE(a && b && B > 0 && C > 0 && G > 0 && HW > 0, "%s: bad arguments B=%d C=%d G=%d HW=%d", what, B, C, G, HW);
    STX_REQUIRE(C % G == 0, "%s: C (%d) %% num_groups (%d) != 0", what, C, G);            // FoundationStereo/submodule.py:390
    STX_REQUIRE((long long)B * G < 65536, "%s: B * G = %lld exceeds the launch grid", what, (long long)B * G);
    STX_REQUIRE((long long)B * C * HW < (1ll << 40), "%s: tensor too large", what);
    return STX_OK;
}

}  // namespace

extern "C" int stx_group_normalize_fwd(const float* x, float* y, int B, int C, int G, int HW, float out_scale, void* stream) {
    stx_begin();
    if (int rc = gn_check(x, y, B, C, G, HW, "group_normalize_fwd")) return rc;
    hipLaunchKernelGGL(group_normalize_fwd_kernel, dim3(stx_cdiv(HW, GN_THREADS), B * G), dim3(GN_THREADS), 0,
                       (hipStream_t)stream, x, y, C / G, HW, out_scale);
    return stx_check_launch("group_normalize_fwd");
}

extern "C" int stx_group_normalize_bwd(const float* x, const float* gy, float* gx, int B, int C, int G, int HW, float out_scale,
                                       void* stream) {
    stx_begin();
    if (int rc = gn_check(x, gx, B, C, G, HW, "group_normalize_bwd")) return rc;
    STX_REQUIRE(gy != nullptr, "group_normalize_bwd: gy missing");
    hipLaunchKernelGGL(group_normalize_bwd_kernel, dim3(stx_cdiv(HW, GN_THREADS), B * G), dim3(GN_THREADS), 0,
                       (hipStream_t)stream, x, gy, gx, C / G, HW, out_scale);
    return stx_check_launch("group_normalize_bwd");
}
