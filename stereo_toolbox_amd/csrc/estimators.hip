// Modal disparity estimators for gfx950 (reference /root/reference/stereo_toolbox):
//   unimodal_disparity_estimator(x, maxdisp)        disparity_estimators/unimodal_disparity_estimator.py:4-25
//   dominant_modal_disparity_estimator(x, maxdisp)  disparity_estimators/dominant_modal_disparity_estimator.py:5-54
// Both take the probability volume x [B,D,H,W] and return a disparity map [B,1,H,W]: the expectation of d over one
// mode of the per-pixel distribution, re-normalised.
//
// The reference builds ~15 full-volume temporaries (repeat, diff, flip, int masks, products: ~6 GB of traffic
// at 576x960, D=192); here one thread owns one pixel (lanes run along W, so every access of a wave is a contiguous
// row segment).  The walks over a pixel's D probabilities are data dependent (arg-max, then outwards to the edges of
// the mode, then over the support) and touch every entry 3-10 times, so a wave first copies its 64 columns into LDS
// ([d][lane]: a lane only ever reads its own column -- no barrier, no bank conflict) with 64 loads in flight per lane,
// and walks them there: HBM sees the volume exactly once.  Roofline: HBM, algorithmic bytes = the volume once.
// Columns longer than ES_LDS_MAX_D entries (160 KiB / 256 B) are walked in place instead (the round-1 path).
//
// Mode support around the arg-max of a sequence s[0..D) (extended by s[-1] = s[D] = 1):
//   hi = (first j > index with s[j] > s[j-1]) - 1     (-1 when there is none, i.e. s[D-1] >= 1)
//   lo = last j <= index with s[j] < s[j-1]           (D-1 when there is none)
// The reference tests the sign of the fp32 difference s[j] - s[j-1]; with gradual underflow that is the same
// predicate as the comparison used here.
#include "stx_common.h"

namespace {

constexpr int ES_THREADS = 256;
constexpr int ES_WAVE = 64;            // LDS-staged kernels: one wave per workgroup, D * 256 B of LDS
constexpr int ES_LDS_MAX_D = 600;      // 150 KiB
constexpr int ES_STAGE = 64;           // loads in flight per lane while staging (3 waves per CU: 48 KiB in flight)

// copy column `p` (stride HW) into the lane's LDS column (stride ES_WAVE)
__device__ __forceinline__ void es_stage_column(const float* __restrict__ p, size_t HW, int D, float* col) {
    int d0 = 0;
    for (; d0 + ES_STAGE <= D; d0 += ES_STAGE) {
        float r[ES_STAGE];
#pragma unroll
        for (int k = 0; k < ES_STAGE; ++k) r[k] = p[(size_t)(d0 + k) * HW];
#pragma unroll
        for (int k = 0; k < ES_STAGE; ++k) col[(d0 + k) * ES_WAVE] = r[k];
    }
    for (; d0 < D; ++d0) col[d0 * ES_WAVE] = p[(size_t)d0 * HW];
}

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

// aux (optional, for the backward pass): [B][5][HW] = lo, hi, xlo, xhi (support = [lo, hi] minus [xlo, xhi]) and the
// mass S of the support.
__device__ __forceinline__ void es_write_aux(float* aux, size_t b, int HW, int i, int lo, int hi, int xlo, int xhi,
                                             const float* p, size_t hw) {
    if (!aux) return;
    float S = 0.f;
    for (int d = lo; d <= hi; ++d)
        if (d < xlo || d > xhi) S += p[(size_t)d * hw];
    float* a = aux + b * 5 * (size_t)HW + i;
    a[0] = (float)lo; a[(size_t)HW] = (float)hi; a[2 * (size_t)HW] = (float)xlo; a[3 * (size_t)HW] = (float)xhi;
    a[4 * (size_t)HW] = S;
}

// p: the pixel's column with stride hw (global memory in place, or its LDS copy)
__device__ __forceinline__ void unimodal_pixel(const float* p, size_t hw, float* __restrict__ out, float* __restrict__ aux,
                                               int D, int HW, int b, int i) {
    float best = p[0];
    int bi = 0;
    for (int d = 1; d < D; ++d) {
        const float v = p[(size_t)d * hw];
        if (v > best) { best = v; bi = d; }
    }
    auto f = [&](int d) { return p[(size_t)d * hw]; };
    const ModeRange m = es_mode_bounds(f, D, bi, best);
    out[(size_t)b * HW + i] = es_expect(p, hw, m.lo, m.hi, 1, 0);
    es_write_aux(aux, b, HW, i, m.lo, m.hi, 1, 0, p, hw);
}

__global__ __launch_bounds__(ES_THREADS) void unimodal_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                             float* __restrict__ aux, int D, int HW) {
    const int i = blockIdx.x * ES_THREADS + threadIdx.x, b = blockIdx.y;
    if (i >= HW) return;
    unimodal_pixel(x + (size_t)b * D * HW + i, (size_t)HW, out, aux, D, HW, b, i);
}

__global__ __launch_bounds__(ES_WAVE) void unimodal_lds_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                              float* __restrict__ aux, int D, int HW) {
    STX_DYN_SMEM(smem);
    float* col = reinterpret_cast<float*>(smem) + threadIdx.x;
    const int i = blockIdx.x * ES_WAVE + threadIdx.x, b = blockIdx.y;
    const int ic = i < HW ? i : HW - 1;                      // lanes past the row end stage a valid column and drop it
    es_stage_column(x + (size_t)b * D * HW + ic, (size_t)HW, D, col);
    if (i < HW) unimodal_pixel(col, (size_t)ES_WAVE, out, aux, D, HW, b, i);
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

__device__ __forceinline__ void dominant_modal_pixel(const float* p, size_t hw, float* __restrict__ out,
                                                     float* __restrict__ aux, int D, int HW, int b, int i) {
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
    if (sy >= sz) es_write_aux(aux, b, HW, i, a1, b1, 1, 0, p, hw);
    else es_write_aux(aux, b, HW, i, a2, b2, a1, b1, p, hw);
}

__global__ __launch_bounds__(ES_THREADS) void dominant_modal_kernel(const float* __restrict__ x,
                                                                   float* __restrict__ out, float* __restrict__ aux,
                                                                   int D, int HW) {
    const int i = blockIdx.x * ES_THREADS + threadIdx.x, b = blockIdx.y;
    if (i >= HW) return;
    dominant_modal_pixel(x + (size_t)b * D * HW + i, (size_t)HW, out, aux, D, HW, b, i);
}

__global__ __launch_bounds__(ES_WAVE) void dominant_modal_lds_kernel(const float* __restrict__ x,
                                                                    float* __restrict__ out, float* __restrict__ aux,
                                                                    int D, int HW) {
    STX_DYN_SMEM(smem);
    float* col = reinterpret_cast<float*>(smem) + threadIdx.x;
    const int i = blockIdx.x * ES_WAVE + threadIdx.x, b = blockIdx.y;
    const int ic = i < HW ? i : HW - 1;
    es_stage_column(x + (size_t)b * D * HW + ic, (size_t)HW, D, col);
    if (i < HW) dominant_modal_pixel(col, (size_t)ES_WAVE, out, aux, D, HW, b, i);
}

// split_mode(x, maxdisp) -> (mode, mask)  (loss_functions/split_mode.py:9-35): the support of modal_mask() taken on the
// RAW volume (no blur) -- arg-max, outwards to the edges of its mode, symmetrised around the arg-max when it sits 3 or
// more bins off the centre -- returned as the boolean mask [B][D][HW] and as mode = x * mask.  The column is staged in LDS
// once; the mask / mode rows are written coalesced (lanes along W).  `x * mask` keeps the reference's arithmetic: x * 1
// inside, x * 0 outside (a signed zero for negative x, NaN for a non-finite x), bit for bit.
__device__ __forceinline__ void split_mode_pixel(const float* p, size_t hw, float* __restrict__ mode,
                                                 unsigned char* __restrict__ mask, int D, int HW, int b, int i) {
    float best = p[0];
    int bi = 0;
    for (int d = 1; d < D; ++d) {
        const float v = p[(size_t)d * hw];
        if (v > best) { best = v; bi = d; }
    }
    auto f = [&](int d) { return p[(size_t)d * hw]; };
    const ModeRange m = es_mode_bounds(f, D, bi, best);
    int c = 2 * m.index - m.hi - m.lo;
    c = c < 0 ? -c : c;
    int lo = m.lo, hi = m.hi;
    if (c >= 3) {
        const int r = (m.hi - m.index < m.index - m.lo) ? m.hi - m.index : m.index - m.lo;
        lo = m.index - r; hi = m.index + r;
    }
    float* q = mode + (size_t)b * D * HW + i;
    unsigned char* mk = mask + (size_t)b * D * HW + i;
    for (int d = 0; d < D; ++d) {
        const bool in = d >= lo && d <= hi;
        q[(size_t)d * HW] = p[(size_t)d * hw] * (in ? 1.f : 0.f);
        mk[(size_t)d * HW] = in ? 1 : 0;
    }
}

__global__ __launch_bounds__(ES_THREADS) void split_mode_kernel(const float* __restrict__ x, float* __restrict__ mode,
                                                               unsigned char* __restrict__ mask, int D, int HW) {
    const int i = blockIdx.x * ES_THREADS + threadIdx.x, b = blockIdx.y;
    if (i >= HW) return;
    split_mode_pixel(x + (size_t)b * D * HW + i, (size_t)HW, mode, mask, D, HW, b, i);
}

__global__ __launch_bounds__(ES_WAVE) void split_mode_lds_kernel(const float* __restrict__ x, float* __restrict__ mode,
                                                                unsigned char* __restrict__ mask, int D, int HW) {
    STX_DYN_SMEM(smem);
    float* col = reinterpret_cast<float*>(smem) + threadIdx.x;
    const int i = blockIdx.x * ES_WAVE + threadIdx.x, b = blockIdx.y;
    const int ic = i < HW ? i : HW - 1;
    es_stage_column(x + (size_t)b * D * HW + ic, (size_t)HW, D, col);
    if (i < HW) split_mode_pixel(col, (size_t)ES_WAVE, mode, mask, D, HW, b, i);
}

// Backward of both estimators.  The support mask is a constant of the graph (`x * mask.data`,
// unimodal_disparity_estimator.py:20; boolean masks in dominant_modal_disparity_estimator.py:45-49), so with
// out = sum_d d x_d m_d / S, S = sum_d x_d m_d:   d out / d x_k = m_k (k - out) / S.
// One thread per pixel writes its D gradients (zero outside the support): lanes along W, coalesced rows.
__global__ __launch_bounds__(ES_THREADS) void modal_bwd_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                              const float* __restrict__ aux, float* __restrict__ gx,
                                                              int D, int HW) {
    const int i = blockIdx.x * ES_THREADS + threadIdx.x, b = blockIdx.y;
    if (i >= HW) return;
    const float* a = aux + (size_t)b * 5 * HW + i;
    const int lo = (int)a[0], hi = (int)a[(size_t)HW], xlo = (int)a[2 * (size_t)HW], xhi = (int)a[3 * (size_t)HW];
    const float S = a[4 * (size_t)HW];
    const float o = out[(size_t)b * HW + i], gg = g[(size_t)b * HW + i];
    float* q = gx + (size_t)b * D * HW + i;
    for (int d = 0; d < D; ++d) {
        const bool in = d >= lo && d <= hi && (d < xlo || d > xhi);
        q[(size_t)d * HW] = in ? gg * (((float)d - o) / S) : 0.f;
    }
}

}  // namespace

static int modal_launch(const float* x, float* out, float* aux, int B, int D, int HW, int kind, void* stream,
                        const char* what) {
    hipStream_t st = (hipStream_t)stream;
    if (D <= ES_LDS_MAX_D) {
        const dim3 grid(stx_cdiv(HW, ES_WAVE), B);
        const size_t lds = (size_t)D * ES_WAVE * sizeof(float);
        if (kind == 0) {
            if (lds > 64 * 1024)
                hipFuncSetAttribute((const void*)unimodal_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(unimodal_lds_kernel, grid, dim3(ES_WAVE), lds, st, x, out, aux, D, HW);
        } else {
            if (lds > 64 * 1024)
                hipFuncSetAttribute((const void*)dominant_modal_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds);
            hipLaunchKernelGGL(dominant_modal_lds_kernel, grid, dim3(ES_WAVE), lds, st, x, out, aux, D, HW);
        }
    } else {
        const dim3 grid(stx_cdiv(HW, ES_THREADS), B);
        if (kind == 0) hipLaunchKernelGGL(unimodal_kernel, grid, dim3(ES_THREADS), 0, st, x, out, aux, D, HW);
        else hipLaunchKernelGGL(dominant_modal_kernel, grid, dim3(ES_THREADS), 0, st, x, out, aux, D, HW);
    }
    return stx_check_launch(what);
}

extern "C" int stx_unimodal_fwd(const float* x, float* out, int B, int D, int HW, void* stream) {
    stx_begin();
    STX_REQUIRE(x && out && B > 0 && D > 0 && HW > 0, "unimodal_fwd: bad shape");
    return modal_launch(x, out, nullptr, B, D, HW, 0, stream, "unimodal_fwd");
}

extern "C" int stx_dominant_modal_fwd(const float* x, float* out, int B, int D, int HW, void* stream) {
    stx_begin();
    STX_REQUIRE(x && out && B > 0 && D > 0 && HW > 0, "dominant_modal_fwd: bad shape");
    return modal_launch(x, out, nullptr, B, D, HW, 1, stream, "dominant_modal_fwd");
}

// kind 0: unimodal, 1: dominant-modal; aux [B][5][HW] is what stx_modal_bwd needs (may be null: plain forward)
extern "C" int stx_modal_fwd(const float* x, float* out, float* aux, int B, int D, int HW, int kind, void* stream) {
    stx_begin();
    STX_REQUIRE(x && out && B > 0 && D > 0 && HW > 0 && (kind == 0 || kind == 1), "modal_fwd: bad arguments");
    return modal_launch(x, out, aux, B, D, HW, kind, stream, "modal_fwd");
}

extern "C" int stx_modal_bwd(const float* g, const float* out, const float* aux, float* gx, int B, int D, int HW,
                             void* stream) {
    stx_begin();
    STX_REQUIRE(g && out && aux && gx && B > 0 && D > 0 && HW > 0, "modal_bwd: bad arguments");
    hipLaunchKernelGGL(modal_bwd_kernel, dim3(stx_cdiv(HW, ES_THREADS), B), dim3(ES_THREADS), 0, (hipStream_t)stream, g,
                       out, aux, gx, D, HW);
    return stx_check_launch("modal_bwd");
}

// split_mode (loss_functions/split_mode.py:9-35): mode [B][D][HW] fp32 and mask [B][D][HW] bytes (0 / 1; torch.bool storage)
extern "C" int stx_split_mode(const float* x, float* mode, unsigned char* mask, int B, int D, int HW, void* stream) {
    stx_begin();
    STX_REQUIRE(x && mode && mask && B > 0 && D > 0 && HW > 0, "split_mode: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (D <= ES_LDS_MAX_D) {
        const size_t lds = (size_t)D * ES_WAVE * sizeof(float);
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute((const void*)split_mode_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return stx_set_error(STX_ERR_LAUNCH, "split_mode: %d bytes of dynamic LDS refused by this device", (int)lds);
        hipLaunchKernelGGL(split_mode_lds_kernel, dim3(stx_cdiv(HW, ES_WAVE), B), dim3(ES_WAVE), lds, st, x, mode, mask, D, HW);
    } else {
        hipLaunchKernelGGL(split_mode_kernel, dim3(stx_cdiv(HW, ES_THREADS), B), dim3(ES_THREADS), 0, st, x, mode, mask, D, HW);
    }
    return stx_check_launch("split_mode");
}
