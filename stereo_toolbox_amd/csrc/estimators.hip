// Modal disparity estimators for gfx950 (reference /root/reference/stereo_toolbox):
//   unimodal_disparity_estimator(x, maxdisp)        disparity_estimators/unimodal_disparity_estimator.py:4-25
//   dominant_modal_disparity_estimator(x, maxdisp)  disparity_estimators/dominant_modal_disparity_estimator.py:5-54
// Both take the probability volume x [B,D,H,W] and return a disparity map [B,1,H,W]: the expectation of d over one
// mode of the per-pixel distribution, re-normalised.
//
// The reference builds ~15 full-volume temporaries (repeat, diff, flip, int masks, products: ~6 GB of traffic
// at 576x960, D=192); here one thread owns one pixel and walks its D probabilities in place (lanes run along W,
// so every access of a wave is a contiguous row segment).  Roofline: HBM, algorithmic bytes = the volume once
// (the later passes of a pixel hit L2).
//
// Mode support around the arg-max of a sequence s[0..D) (extended by s[-1] = s[D] = 1):
//   hi = (first j > index with s[j] > s[j-1]) - 1     (-1 when there is none, i.e. s[D-1] >= 1)
//   lo = last j <= index with s[j] < s[j-1]           (D-1 when there is none)
// The reference tests the sign of the fp32 difference s[j] - s[j-1]; with gradual underflow that is the same
// predicate as the comparison used here.
#include "stx_common.h"

namespace {

constexpr int ES_THREADS = 256;

struct ModeRange { int index, lo, hi; };

// F: d -> s[d] for 0 <= d < D
template <class F>
__device__ __forceinline__ ModeRange es_mode_bounds(const F& f, int D, int index, float best) {
    ModeRange m;
    m.index = index;
    // right edge
    float prev = best;
    int j = index + 1;
    m.hi = -2;
    for (; j < D; ++j) {
        const float v = f(j);
        if (v > prev) { m.hi = j - 1; break; }
        prev = v;
    }
    if (m.hi == -2) m.hi = (1.0f > prev) ? D - 1 : -1;
    // left edge
    float cur = best;
    m.lo = D - 1;
    for (j = index; j >= 0; --j) {
        const float p = (j > 0) ? f(j - 1) : 1.0f;
        if (cur < p) { m.lo = j; break; }
        cur = p;
    }
    return m;
}

// expectation of d over [lo, hi] of x, re-normalised (0/0 -> NaN exactly like the reference's x / sum(x))
__device__ __forceinline__ float es_expect(const float* __restrict__ p, size_t HW, int lo, int hi, int xlo, int xhi) {
    // entries inside [xlo, xhi] are excluded (dominant-modal second mode); pass xlo > xhi for none
    float S = 0.f;
    for (int d = lo; d <= hi; ++d)
        if (d < xlo || d > xhi) S += p[(size_t)d * HW];
    float a = 0.f;
    bool any = false;
    for (int d = lo; d <= hi; ++d)
        if (d < xlo || d > xhi) { a = fmaf(p[(size_t)d * HW] / S, (float)d, a); any = true; }
    if (!any || S == 0.f) a = __builtin_nanf("");
    return a;
}

__global__ __launch_bounds__(ES_THREADS) void unimodal_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                             int D, int HW) {
    const int i = blockIdx.x * ES_THREADS + threadIdx.x, b = blockIdx.y;
    if (i >= HW) return;
    const float* p = x + (size_t)b * D * HW + i;
    float best = p[0];
    int bi = 0;
    for (int d = 1; d < D; ++d) {
        const float v = p[(size_t)d * HW];
        if (v > best) { best = v; bi = d; }
    }
    const size_t hw = (size_t)HW;
    auto f = [&](int d) { return p[(size_t)d * hw]; };
    const ModeRange m = es_mode_bounds(f, D, bi, best);
    out[(size_t)b * HW + i] = es_expect(p, hw, m.lo, m.hi, 1, 0);
}

// 5-tap box blur along D with zero padding (conv1d, padding='same', weight 1/5), entries of [zlo, zhi] forced to 0
struct BlurSeq {
    const float* p;
    size_t HW;
    int D, zlo, zhi;
    __device__ __forceinline__ float raw(int d) const { return (d >= 0 && d < D) ? p[(size_t)d * HW] : 0.f; }
    __device__ __forceinline__ float operator()(int d) const {
        if (d >= zlo && d <= zhi) return 0.f;
        float a = raw(d - 2) * 0.2f;
        a = fmaf(raw(d - 1), 0.2f, a);
        a = fmaf(raw(d), 0.2f, a);
        a = fmaf(raw(d + 1), 0.2f, a);
        a = fmaf(raw(d + 2), 0.2f, a);
        return a;
    }
};

// arg-max of a BlurSeq with a sliding 5-entry window (one new load per step)
__device__ __forceinline__ int es_argmax_blur(const BlurSeq& s, float* best_out) {
    float w0 = 0.f, w1 = 0.f, w2 = s.raw(0), w3 = s.raw(1), w4 = s.raw(2);
    float best = -1.f;
    int bi = 0;
    for (int d = 0; d < s.D; ++d) {
        float a = w0 * 0.2f;
        a = fmaf(w1, 0.2f, a);
        a = fmaf(w2, 0.2f, a);
        a = fmaf(w3, 0.2f, a);
        a = fmaf(w4, 0.2f, a);
        if (d >= s.zlo && d <= s.zhi) a = 0.f;
        if (d == 0 || a > best) { best = a; bi = d; }
        w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = s.raw(d + 3);
    }
    *best_out = best;
    return bi;
}

// support of the mode (dominant_modal_disparity_estimator.py:5-32): symmetrised around the arg-max when the
// arg-max sits 3 or more bins off the centre of [lo, hi]
__device__ __forceinline__ void es_modal_range(const BlurSeq& s, int* lo, int* hi) {
    float best;
    const int bi = es_argmax_blur(s, &best);
    const ModeRange m = es_mode_bounds(s, s.D, bi, best);
    int c = 2 * m.index - m.hi - m.lo;
    c = c < 0 ? -c : c;
    if (c < 3) { *lo = m.lo; *hi = m.hi; }
    else {
        const int r = (m.hi - m.index < m.index - m.lo) ? m.hi - m.index : m.index - m.lo;
        *lo = m.index - r; *hi = m.index + r;
    }
    if (*lo < 0) *lo = 0;                       // a range mask only ever selects 0 <= d < D
    if (*hi > s.D - 1) *hi = s.D - 1;
}

__global__ __launch_bounds__(ES_THREADS) void dominant_modal_kernel(const float* __restrict__ x,
                                                                   float* __restrict__ out, int D, int HW) {
    const int i = blockIdx.x * ES_THREADS + threadIdx.x, b = blockIdx.y;
    if (i >= HW) return;
    const float* p = x + (size_t)b * D * HW + i;
    const size_t hw = (size_t)HW;
    BlurSeq s{p, hw, D, 1, 0};
    int a1, b1, a2, b2;
    es_modal_range(s, &a1, &b1);                // main mode of the blurred volume
    s.zlo = a1; s.zhi = b1;
    es_modal_range(s, &a2, &b2);                // main mode of what is left
    // probability mass of the two candidate supports (second: its part outside the first)
    float sy = 0.f, sz = 0.f;
    for (int d = a1; d <= b1; ++d) sy += p[(size_t)d * hw];
    for (int d = a2; d <= b2; ++d)
        if (d < a1 || d > b1) sz += p[(size_t)d * hw];
    out[(size_t)b * HW + i] = (sy >= sz) ? es_expect(p, hw, a1, b1, 1, 0) : es_expect(p, hw, a2, b2, a1, b1);
}

}  // namespace

extern "C" int stx_unimodal_fwd(const float* x, float* out, int B, int D, int HW, void* stream) {
    stx_begin();
    STX_REQUIRE(x && out && B > 0 && D > 0 && HW > 0, "unimodal_fwd: bad shape");
    hipLaunchKernelGGL(unimodal_kernel, dim3(stx_cdiv(HW, ES_THREADS), B), dim3(ES_THREADS), 0, (hipStream_t)stream, x,
                       out, D, HW);
    return stx_check_launch("unimodal_fwd");
}

extern "C" int stx_dominant_modal_fwd(const float* x, float* out, int B, int D, int HW, void* stream) {
    stx_begin();
    STX_REQUIRE(x && out && B > 0 && D > 0 && HW > 0, "dominant_modal_fwd: bad shape");
    hipLaunchKernelGGL(dominant_modal_kernel, dim3(stx_cdiv(HW, ES_THREADS), B), dim3(ES_THREADS), 0,
                       (hipStream_t)stream, x, out, D, HW);
    return stx_check_launch("dominant_modal_fwd");
}
