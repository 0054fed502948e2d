// Fused disparity-regression head for gfx950 and the stand-alone disparity estimators.
//
// Replaces (reference, /root/reference/stereo_toolbox):
//   F.upsample(cost, [maxdisp,H,W], mode='trilinear') -> squeeze -> F.softmax(dim=1)
//       -> disparity_regression           models/GwcNet/gwcnet.py:197-224,
//                                         models/PSMNet/stackhourglass.py:139-153,
//                                         models/ACVNet/acv.py:206-251
//   disparity_regression(x, maxdisp)      models/GwcNet/submodule.py:23-27
//   disparityregression(maxdisp)(x)       models/PSMNet/submodule.py:46-54
//   softargmax_/argmax_disparity_estimator disparity_estimators/__init__.py:7-15
//   F.softmax(att_weights, dim=2)         models/ACVNet/acv.py:196
//
// The reference chain materialises three [B,192,H,W] tensors (425 MB each at 576x960); the fused
// kernel reads the quarter-resolution cost (6.6 MB) and writes the disparity map (2.2 MB):
// algorithmic bytes 8 847 360.  Roofline: HBM (in practice latency/exp bound).
//
// Interpolation follows ATen's align_corners=False rule on all three axes:
//   src = max(scale*(dst+0.5)-0.5, 0), i0 = floor(src), i1 = min(i0+1, n-1), t = src-i0.
// One thread owns one output pixel: it walks the 192 output disparities once, advancing a
// two-entry window of H/W-interpolated cost samples, with an online (running-max) softmax.
#include "stx_common.h"
#include <stdlib.h>

namespace {

constexpr int HD_THREADS = 256;

struct Lerp { int i0, i1; float t; };

// AC = false: align_corners=False (GwcNet / PSMNet / ACVNet heads): src = max(scale*(dst+0.5)-0.5, 0), scale = in/out
// AC = true : align_corners=True  (PCWNet / CFNet heads, models/PCWNet/pcwnet.py:446-470): src = scale*dst,
//             scale = (in-1)/(out-1)
template <bool AC>
__device__ __forceinline__ float hd_scale(int n_in, int n_out) {
    if (AC) return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
    return (float)n_in / (float)n_out;
}

template <bool AC>
__device__ __forceinline__ Lerp hd_src(int dst, float scale, int n) {
    float s;
    if (AC) {
        s = scale * (float)dst;
    } else {
        s = scale * ((float)dst + 0.5f) - 0.5f;
        s = s < 0.f ? 0.f : s;
    }
    int i0 = (int)s;
    if (i0 > n - 1) i0 = n - 1;
    Lerp r;
    r.i0 = i0;
    r.i1 = i0 + (i0 < n - 1 ? 1 : 0);
    r.t = s - (float)i0;
    return r;
}

__device__ __forceinline__ float hd_sample(const float* __restrict__ plane, int Wc, Lerp lh, Lerp lw) {
    const float a = plane[lh.i0 * Wc + lw.i0], b = plane[lh.i0 * Wc + lw.i1];
    const float c = plane[lh.i1 * Wc + lw.i0], d = plane[lh.i1 * Wc + lw.i1];
    return (1.f - lh.t) * ((1.f - lw.t) * a + lw.t * b) + lh.t * ((1.f - lw.t) * c + lw.t * d);
}

template <bool AC>
__global__ __launch_bounds__(HD_THREADS) void head_fwd_kernel(
    const float* __restrict__ cost, float* __restrict__ disp, float* __restrict__ stats,
    int Dc, int Hc, int Wc, int D, int H, int W) {
    const int w = blockIdx.x * HD_THREADS + threadIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    if (w >= W) return;
    const float rd = hd_scale<AC>(Dc, D), rh = hd_scale<AC>(Hc, H), rw = hd_scale<AC>(Wc, W);
    const Lerp lh = hd_src<AC>(h, rh, Hc), lw = hd_src<AC>(w, rw, Wc);
    const float* cb = cost + (size_t)b * Dc * Hc * Wc;
    const int plane = Hc * Wc;
    int cur = 0;
    float c0 = hd_sample(cb, Wc, lh, lw);
    float c1 = Dc > 1 ? hd_sample(cb + plane, Wc, lh, lw) : c0;
    float m = -3.0e38f, s = 0.f, t = 0.f;
    for (int d = 0; d < D; ++d) {
        const Lerp ld = hd_src<AC>(d, rd, Dc);
        while (cur < ld.i0) {
            ++cur;
            c0 = c1;
            const int nx = cur + 1 < Dc ? cur + 1 : Dc - 1;
            c1 = hd_sample(cb + (size_t)nx * plane, Wc, lh, lw);
        }
        const float hi = (ld.i1 == ld.i0) ? c0 : c1;
        const float x = (1.f - ld.t) * c0 + ld.t * hi;
        if (x > m) {
            const float f = stx_exp(m - x);
            s *= f;
            t *= f;
            m = x;
        }
        const float e = stx_exp(x - m);
        s += e;
        t = fmaf((float)d, e, t);
    }
    const size_t o = ((size_t)b * H + h) * W + w;
    disp[o] = t / s;
    if (stats) { stats[2 * o] = m; stats[2 * o + 1] = s; }
}

// Backward, deterministic and atomic-free, in two streaming passes:
//  pass 1 (one thread per output pixel): dlogit_d = g * p_d * (d - disp), pushed back through the D
//          lerp into the pixel's 48 H/W-interpolated cost samples -> gpix[b][dc][h][w];
//  pass 2 (one thread per cost cell): gathers gpix over the <= 8x8 pixels whose H/W lerp touches the
//          cell, with the same lerp weights.
template <bool AC>
__global__ __launch_bounds__(HD_THREADS) void head_bwd_pix_kernel(
    const float* __restrict__ gout, const float* __restrict__ cost, const float* __restrict__ disp,
    const float* __restrict__ stats, float* __restrict__ gpix, int Dc, int Hc, int Wc, int D, int H, int W) {
    const int w = blockIdx.x * HD_THREADS + threadIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    if (w >= W) return;
    const float rd = hd_scale<AC>(Dc, D), rh = hd_scale<AC>(Hc, H), rw = hd_scale<AC>(Wc, W);
    const Lerp lh = hd_src<AC>(h, rh, Hc), lw = hd_src<AC>(w, rw, Wc);
    const float* cb = cost + (size_t)b * Dc * Hc * Wc;
    const int plane = Hc * Wc;
    const size_t o = ((size_t)b * H + h) * W + w;
    const float g = gout[o], dv = disp[o], m = stats[2 * o], inv_s = 1.f / stats[2 * o + 1];
    float* gp = gpix + ((size_t)b * Dc * H + h) * W + w;      // + dc * H * W
    const size_t pstride = (size_t)H * W;
    int cur = 0;
    float c0 = hd_sample(cb, Wc, lh, lw);
    float c1 = Dc > 1 ? hd_sample(cb + plane, Wc, lh, lw) : c0;
    float a0 = 0.f, a1 = 0.f;   // gradient wrt c0 / c1
    for (int d = 0; d < D; ++d) {
        const Lerp ld = hd_src<AC>(d, rd, Dc);
        while (cur < ld.i0) {
            gp[cur * pstride] = a0;
            ++cur;
            c0 = c1; a0 = a1; a1 = 0.f;
            const int nx = cur + 1 < Dc ? cur + 1 : Dc - 1;
            c1 = hd_sample(cb + (size_t)nx * plane, Wc, lh, lw);
        }
        const bool same = ld.i1 == ld.i0;
        const float hi = same ? c0 : c1;
        const float x = (1.f - ld.t) * c0 + ld.t * hi;
        const float p = stx_exp(x - m) * inv_s;
        const float gl = g * p * ((float)d - dv);
        if (same) a0 += gl; else { a0 = fmaf(1.f - ld.t, gl, a0); a1 = fmaf(ld.t, gl, a1); }
    }
    gp[cur * pstride] = a0;
    if (cur + 1 < Dc) gp[(cur + 1) * pstride] = a1;
    for (int dc = cur + 2; dc < Dc; ++dc) gp[dc * pstride] = 0.f;
}

// ---- second generation of the two per-pixel kernels (same results up to the softmax shift; chosen when the LDS fits).
// The first versions spend ~30 VALU instructions per (pixel, disparity): the D-axis lerp coordinates -- identical for every
// lane -- are recomputed per lane and step, the softmax is the online (rescaling) form with a divergent branch, and the
// H/W-interpolated cost samples arrive through four dependent gathers every fourth step (head_fwd 0.092 ms at 576x960 where
// 192 x 553k exp + lerp need ~0.03 ms of VALU time).  Here a workgroup of 256 pixels of one image row
//   1. builds a per-launch table in LDS: {1-t, t, i0} of every output disparity (+ a sentinel),
//   2. gathers every lane's Dc H/W-interpolated samples once into LDS (cs[k][lane]: conflict-free), eight planes in
//      flight, keeping their maximum M (an upper bound of every interpolated logit: lerp weights are in [0, 1]),
//   3. walks the output disparities ONCE with everything it needs one step ahead in registers: the next table entry, and
//      c0 / c1 / the prefetched c(k+2) of the current coarse interval (i0 is non-decreasing and wave-uniform, so the
//      interval change is a scalar branch) -- no LDS latency on the dependent chain, one exp per step, softmax shift M.
// M can exceed the true maximum by the lerp gap of the steepest interval; should that ever underflow the whole sum of a
// lane (cost steps of several hundred between neighbouring planes), the workgroup redoes the walk with the exact maximum.
struct HdStep { float w0, w1; int k, pad; };   // 1 - t, t, i0

template <bool AC>
__device__ __forceinline__ void hd_build_table(HdStep* tw, int Dc, int D, float rd, int tid) {
    for (int d = tid; d <= D; d += HD_THREADS) {
        const Lerp ld = hd_src<AC>(d < D ? d : D - 1, rd, Dc);
        HdStep t;
        t.w0 = 1.f - ld.t; t.w1 = ld.t; t.k = ld.i0; t.pad = 0;
        tw[d] = t;
    }
}

constexpr int HD_STAGE_UNROLL = 8;

// cs[k][tid] = H/W-interpolated sample of coarse plane k for this lane's pixel; returns their maximum
__device__ __forceinline__ float hd_stage_samples(float* cs, const float* __restrict__ cb, int plane, int Dc, int Wc,
                                                  Lerp lh, Lerp lw, int tid) {
    const int o00 = lh.i0 * Wc + lw.i0, o01 = lh.i0 * Wc + lw.i1, o10 = lh.i1 * Wc + lw.i0, o11 = lh.i1 * Wc + lw.i1;
    const float wh0 = 1.f - lh.t, ww0 = 1.f - lw.t;
    float mx = -3.0e38f;
    int k = 0;
    for (; k + HD_STAGE_UNROLL <= Dc; k += HD_STAGE_UNROLL) {
        float a[HD_STAGE_UNROLL], b[HD_STAGE_UNROLL], c[HD_STAGE_UNROLL], d[HD_STAGE_UNROLL];
#pragma unroll
        for (int u = 0; u < HD_STAGE_UNROLL; ++u) {
            const float* pl = cb + (size_t)(k + u) * plane;
            a[u] = pl[o00]; b[u] = pl[o01]; c[u] = pl[o10]; d[u] = pl[o11];
        }
#pragma unroll
        for (int u = 0; u < HD_STAGE_UNROLL; ++u) {
            const float v = wh0 * (ww0 * a[u] + lw.t * b[u]) + lh.t * (ww0 * c[u] + lw.t * d[u]);
            cs[(k + u) * HD_THREADS + tid] = v;
            mx = fmaxf(mx, v);
        }
    }
    for (; k < Dc; ++k) {
        const float* pl = cb + (size_t)k * plane;
        const float v = wh0 * (ww0 * pl[o00] + lw.t * pl[o01]) + lh.t * (ww0 * pl[o10] + lw.t * pl[o11]);
        cs[k * HD_THREADS + tid] = v;
        mx = fmaxf(mx, v);
    }
    return mx;
}

static size_t hd_lds_bytes(int Dc, int D) {
    return (size_t)Dc * HD_THREADS * 4 + (size_t)(D + 1) * sizeof(HdStep);
}

// The pipelined walk over d = 0 .. D-1.  step(d as float, w0, w1, c0, c1) runs once per output disparity;
// leave(k, kn) runs when the coarse interval changes from k to kn > k (wave-uniform) and once at the end (kn = -1);
// tie(nk) pins the step's results and the prefetched interval index to one program point (STX_TIE3), so that the wait
// for the table entry sits behind the step's arithmetic instead of in front of it (hipcc hoists the branch otherwise).
template <typename Step, typename Leave, typename Tie>
__device__ __forceinline__ void hd_walk(const float* cs, const HdStep* tw, int Dc, int D, int tid, Step&& step, Leave&& leave,
                                        Tie&& tie) {
    HdStep cur = tw[0];
    int k = __builtin_amdgcn_readfirstlane(cur.k);
    auto at = [&](int kk) { return cs[(kk < Dc ? kk : Dc - 1) * HD_THREADS + tid]; };
    float c0 = at(k), c1 = at(k + 1), cn = at(k + 2);
    float fd = 0.f;
    for (int d = 0; d < D; ++d) {
        const HdStep nxt = tw[d + 1];                       // (in flight during this step's arithmetic)
        STX_SCHED_BARRIER();
        step(fd, cur.w0, cur.w1, c0, c1);
        fd += 1.f;
        int nk = nxt.k;
        tie(nk);
        const int kn = __builtin_amdgcn_readfirstlane(nk);
        if (kn != k) {
            leave(k, kn);
            if (kn == k + 1) { c0 = c1; c1 = cn; }
            else { c0 = at(kn); c1 = at(kn + 1); }
            k = kn;
            cn = at(k + 2);
        }
        cur = nxt;
    }
    leave(k, -1);
}

template <bool AC>
__global__ __launch_bounds__(HD_THREADS) void head_fwd_lds_kernel(
    const float* __restrict__ cost, float* __restrict__ disp, float* __restrict__ stats,
    int Dc, int Hc, int Wc, int D, int H, int W) {
    STX_DYN_SMEM(smem);
    __shared__ int redo;
    float* cs = reinterpret_cast<float*>(smem);                           // [Dc][256]
    HdStep* tw = reinterpret_cast<HdStep*>(cs + (size_t)Dc * HD_THREADS);   // [D + 1]
    const int tid = threadIdx.x;
    const int w_raw = blockIdx.x * HD_THREADS + tid;
    const int w = w_raw < W ? w_raw : W - 1;                             // (every lane reaches the barriers)
    const int h = blockIdx.y, b = blockIdx.z;
    const float rd = hd_scale<AC>(Dc, D), rh = hd_scale<AC>(Hc, H), rw = hd_scale<AC>(Wc, W);
    const Lerp lh = hd_src<AC>(h, rh, Hc), lw = hd_src<AC>(w, rw, Wc);
    const int plane = Hc * Wc;
    if (tid == 0) redo = 0;
    hd_build_table<AC>(tw, Dc, D, rd, tid);
    float m = hd_stage_samples(cs, cost + (size_t)b * Dc * plane, plane, Dc, Wc, lh, lw, tid);
    __syncthreads();
    float s = 0.f, acc = 0.f;
    hd_walk(cs, tw, Dc, D, tid,
            [&](float fd, float w0, float w1, float c0, float c1) {
                const float e = stx_exp(w0 * c0 + w1 * c1 - m);
                s += e;
                acc = fmaf(fd, e, acc);
            },
            [](int, int) {}, [&](int& nk) { STX_TIE3(s, acc, nk); });
    if (!(s > 1e-30f)) redo = 1;                                         // (also catches NaN logits: the exact walk reproduces them)
    __syncthreads();
    if (redo) {
        m = -3.0e38f;
        hd_walk(cs, tw, Dc, D, tid, [&](float, float w0, float w1, float c0, float c1) { m = fmaxf(m, w0 * c0 + w1 * c1); },
                [](int, int) {}, [](int&) {});
        s = 0.f; acc = 0.f;
        hd_walk(cs, tw, Dc, D, tid,
                [&](float fd, float w0, float w1, float c0, float c1) {
                    const float e = stx_exp(w0 * c0 + w1 * c1 - m);
                    s += e;
                    acc = fmaf(fd, e, acc);
                },
                [](int, int) {}, [](int&) {});
    }
    if (w_raw < W) {
        const size_t o = ((size_t)b * H + h) * W + w;
        disp[o] = acc / s;
        if (stats) { stats[2 * o] = m; stats[2 * o + 1] = s; }
    }
}

template <bool AC>
__device__ __forceinline__ float hd_weight(int dst, float scale, int n, int cell) {
    const Lerp l = hd_src<AC>(dst, scale, n);
    return (l.i0 == cell ? 1.f - l.t : 0.f) + (l.i1 == cell ? l.t : 0.f);
}

template <bool AC>
__global__ __launch_bounds__(HD_THREADS) void head_bwd_gather_kernel(
    const float* __restrict__ gpix, float* __restrict__ gcost, int Dc, int Hc, int Wc, int H, int W,
    int fh, int fw) {
    const int wc = blockIdx.x * HD_THREADS + threadIdx.x;
    const int hc = blockIdx.y % Hc, dc = blockIdx.y / Hc, b = blockIdx.z;
    if (wc >= Wc) return;
    const float rh = hd_scale<AC>(Hc, H), rw = hd_scale<AC>(Wc, W);
    // pixels that can touch cell (hc, wc): src in (cell-1, cell+1)  ->  dst in a window of ~2/scale
    int h_lo, w_lo;
    if (AC) {
        h_lo = rh > 0.f ? (int)(((float)hc - 1.f) / rh) - 1 : 0;
        w_lo = rw > 0.f ? (int)(((float)wc - 1.f) / rw) - 1 : 0;
    } else {
        h_lo = (int)(((float)hc - 1.f + 0.5f) / rh - 0.5f) - 1;
        w_lo = (int)(((float)wc - 1.f + 0.5f) / rw - 0.5f) - 1;
    }
    h_lo = h_lo < 0 ? 0 : h_lo;
    w_lo = w_lo < 0 ? 0 : w_lo;
    const float* gp = gpix + (((size_t)b * Dc + dc) * H) * W;
    float acc = 0.f;
    for (int i = 0; i < fh; ++i) {
        const int h = h_lo + i;
        if (h >= H) break;
        const float kh = hd_weight<AC>(h, rh, Hc, hc);
        if (kh == 0.f) continue;
        float row = 0.f;
        for (int j = 0; j < fw; ++j) {
            const int w = w_lo + j;
            if (w >= W) break;
            const float kw = hd_weight<AC>(w, rw, Wc, wc);
            if (kw != 0.f) row = fmaf(kw, gp[(size_t)h * W + w], row);
        }
        acc = fmaf(kh, row, acc);
    }
    gcost[(((size_t)b * Dc + dc) * Hc + hc) * Wc + wc] = acc;
}

constexpr int HD_GF = 12;      // footprint bound (rows) of the float4-window gather below

// Second generation of the gather.  The first version recomputes both lerp weights (two hd_src each) for every one of the
// fh x fw candidate pixels of a cell -- ~25 instructions per candidate, 0.132 ms at 576x960, pure VALU time.  (A branch-free
// variant with the weights hoisted and one dword load per candidate measured SLOWER, GPU call O/P: 0.21 ms -- its 144 loads
// per cell have a lane stride of 16 bytes, i.e. 16 texture-addresser cycles each.)  A cell's candidate pixels in a row are contiguous, and neighbouring cells' windows start `1/scale`
// pixels apart -- with the window start rounded down to a multiple of 4 pixels every lane reads whole float4s and, at the
// usual scale of 1/4, the wave's float4s of one load instruction are CONTIGUOUS (1 KiB per instruction).  16 pixels per
// row from the aligned start cover every footprint up to 13 pixels; hd_weight is zero for pixels that do not touch the
// cell, so no footprint bookkeeping is needed.  Requires W % 4 == 0 (16-byte aligned rows).
template <bool AC>
__global__ __launch_bounds__(HD_THREADS) void head_bwd_gather4_kernel(
    const float* __restrict__ gpix, float* __restrict__ gcost, int Dc, int Hc, int Wc, int H, int W, int fh) {
    const int wc = blockIdx.x * HD_THREADS + threadIdx.x;
    const int hc = blockIdx.y % Hc, dc = blockIdx.y / Hc, b = blockIdx.z;
    if (wc >= Wc) return;
    const float rh = hd_scale<AC>(Hc, H), rw = hd_scale<AC>(Wc, W);
    int h_lo, w_lo;
    if (AC) {
        h_lo = rh > 0.f ? (int)(((float)hc - 1.f) / rh) - 1 : 0;
        w_lo = rw > 0.f ? (int)(((float)wc - 1.f) / rw) - 1 : 0;
    } else {
        h_lo = (int)(((float)hc - 1.f + 0.5f) / rh - 0.5f) - 1;
        w_lo = (int)(((float)wc - 1.f + 0.5f) / rw - 0.5f) - 1;
    }
    h_lo = h_lo < 0 ? 0 : h_lo;
    w_lo = w_lo < 0 ? 0 : w_lo;
    const int a_lo = w_lo & ~3;                                   // aligned window start: pixels a_lo .. a_lo + 15
    float kw[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) kw[j] = (a_lo + j < W) ? hd_weight<AC>(a_lo + j, rw, Wc, wc) : 0.f;
    unsigned oq[4];                                               // the four float4s of a row (clamped into the row)
#pragma unroll
    for (int q = 0; q < 4; ++q) oq[q] = (unsigned)(a_lo + 4 * q < W ? a_lo + 4 * q : W - 4);
    const float* gp = gpix + (((size_t)b * Dc + dc) * H) * W;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < HD_GF; ++i) {
        const int hh = h_lo + i;
        const float kh = (i < fh && hh < H) ? hd_weight<AC>(hh, rh, Hc, hc) : 0.f;   // (uniform over the workgroup)
        const float* rowp = gp + (size_t)(hh < H ? hh : H - 1) * W;
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = stx_ld4(rowp + oq[q]);
        float row = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            row = fmaf(kw[4 * q + 0], v[q].x, row);
            row = fmaf(kw[4 * q + 1], v[q].y, row);
            row = fmaf(kw[4 * q + 2], v[q].z, row);
            row = fmaf(kw[4 * q + 3], v[q].w, row);
        }
        acc = fmaf(kh, row, acc);
        if ((i & 3) == 3) STX_SCHED_BARRIER();                   // 16 float4 loads in flight (all 48: 221 VGPRs)
    }
    gcost[(((size_t)b * Dc + dc) * Hc + hc) * Wc + wc] = acc;
}

// disp[b,h,w] = sum_d d * x[b,d,h,w]
__global__ __launch_bounds__(HD_THREADS) void softargmax_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                 int D, int HW) {
    const int i = blockIdx.x * HD_THREADS + threadIdx.x, b = blockIdx.y;
    if (i >= HW) return;
    const float* p = x + (size_t)b * D * HW + i;
    float a = 0.f;
    for (int d = 0; d < D; ++d) a = fmaf((float)d, p[(size_t)d * HW], a);
    out[(size_t)b * HW + i] = a;
}

// Same, four pixels per lane (HW % 4 == 0) and eight disparity planes in flight: 128 bytes of loads per lane instead of
// one dword per trip of a dependent fma chain (the scalar kernel reaches 0.31 of the HBM roofline at 576x960, D = 192).  The per-pixel fma order is unchanged.
constexpr int HD_SA_UNROLL = 8;
__global__ __launch_bounds__(HD_THREADS) void softargmax4_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                  int D, int HW) {
    const int i = (blockIdx.x * HD_THREADS + threadIdx.x) * 4, b = blockIdx.y;
    if (i >= HW) return;
    const float* p = x + (size_t)b * D * HW + i;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int d = 0;
    for (; d + HD_SA_UNROLL <= D; d += HD_SA_UNROLL) {
        float4 v[HD_SA_UNROLL];
#pragma unroll
        for (int k = 0; k < HD_SA_UNROLL; ++k) v[k] = stx_ld4(p + (size_t)(d + k) * HW);
#pragma unroll
        for (int k = 0; k < HD_SA_UNROLL; ++k) {
            const float f = (float)(d + k);
            a.x = fmaf(f, v[k].x, a.x); a.y = fmaf(f, v[k].y, a.y); a.z = fmaf(f, v[k].z, a.z); a.w = fmaf(f, v[k].w, a.w);
        }
    }
    for (; d < D; ++d) {
        const float4 v = stx_ld4(p + (size_t)d * HW);
        const float f = (float)d;
        a.x = fmaf(f, v.x, a.x); a.y = fmaf(f, v.y, a.y); a.z = fmaf(f, v.z, a.z); a.w = fmaf(f, v.w, a.w);
    }
    stx_st4(out + (size_t)b * HW + i, a);
}

__global__ __launch_bounds__(HD_THREADS) void argmax_kernel(const float* __restrict__ x, long long* __restrict__ out,
                                                             int D, int HW) {
    const int i = blockIdx.x * HD_THREADS + threadIdx.x, b = blockIdx.y;
    if (i >= HW) return;
    const float* p = x + (size_t)b * D * HW + i;
    float best = p[0];
    int bi = 0;
    for (int d = 1; d < D; ++d) {
        const float v = p[(size_t)d * HW];
        if (v > best) { best = v; bi = d; }   // first maximum wins, as torch.argmax
    }
    out[(size_t)b * HW + i] = bi;
}

// y[b,d,i] = softmax over d of x[b,d,i]
__global__ __launch_bounds__(HD_THREADS) void softmax_d_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                int D, int HW) {
    const int i = blockIdx.x * HD_THREADS + threadIdx.x, b = blockIdx.y;
    if (i >= HW) return;
    const float* p = x + (size_t)b * D * HW + i;
    float* q = y + (size_t)b * D * HW + i;
    float m = -3.0e38f;
    for (int d = 0; d < D; ++d) m = fmaxf(m, p[(size_t)d * HW]);
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += stx_exp(p[(size_t)d * HW] - m);
    const float inv = 1.f / s;
    for (int d = 0; d < D; ++d) q[(size_t)d * HW] = stx_exp(p[(size_t)d * HW] - m) * inv;
}

}  // namespace

// LDS-staged per-pixel kernels when their tables fit comfortably (two workgroups per CU).  STX_HEAD_V1 = bit mask of the
// kernels to run in their first generation: 1 forward, 2 backward per-pixel pass, 4 backward gather (7 = all)
// Default 2 (GPU call P, 576x960 D=192, kernel trace): forward 95.4 -> 76.7 us with the LDS kernel, but the backward
// per-pixel pass is SLOWER with it (93.1 -> 102.7 us: 50 KB of LDS per workgroup leave 3 waves per SIMD where the first
// version runs 8, and its walk is issue-bound either way); gather 132.3 -> 62.9 us (third generation).
static bool hd_use_lds(size_t lds) { return lds <= 80 * 1024; }    // (the first-generation kernels serve larger D)

template <bool AC>
static int head_fwd_launch(const float* cost, float* disp, float* stats, int B, int Dc, int Hc, int Wc, int D, int H,
                           int W, void* stream) {
    STX_REQUIRE(cost && disp && B > 0 && Dc > 0 && Hc > 0 && Wc > 0 && D > 0 && H > 0 && W > 0, "head_fwd: bad shape");
    dim3 grid(stx_cdiv(W, HD_THREADS), H, B);
    const size_t lds = hd_lds_bytes(Dc, D);
    if (hd_use_lds(lds)) {
        if (lds > 64 * 1024)
            hipFuncSetAttribute((const void*)head_fwd_lds_kernel<AC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(head_fwd_lds_kernel<AC>, grid, dim3(HD_THREADS), lds, (hipStream_t)stream, cost, disp, stats,
                           Dc, Hc, Wc, D, H, W);
    } else {
        hipLaunchKernelGGL(head_fwd_kernel<AC>, grid, dim3(HD_THREADS), 0, (hipStream_t)stream, cost, disp, stats, Dc, Hc,
                           Wc, D, H, W);
    }
    return stx_check_launch("head_fwd");
}

extern "C" int stx_head_fwd(const float* cost, float* disp, float* stats, int B, int Dc, int Hc, int Wc, int D,
                            int H, int W, void* stream) {
    stx_begin();
    return head_fwd_launch<false>(cost, disp, stats, B, Dc, Hc, Wc, D, H, W, stream);
}

// same with an explicit interpolation rule (align_corners != 0: PCWNet / CFNet heads)
extern "C" int stx_head_fwd2(const float* cost, float* disp, float* stats, int B, int Dc, int Hc, int Wc, int D,
                             int H, int W, int align_corners, void* stream) {
    stx_begin();
    return align_corners ? head_fwd_launch<true>(cost, disp, stats, B, Dc, Hc, Wc, D, H, W, stream)
                         : head_fwd_launch<false>(cost, disp, stats, B, Dc, Hc, Wc, D, H, W, stream);
}

extern "C" long long stx_head_bwd_workspace_floats(int B, int Dc, int H, int W) { return (long long)B * Dc * H * W; }

template <bool AC>
static int head_bwd_launch(const float* gout, const float* cost, const float* disp, const float* stats, float* gcost,
                           float* workspace, int B, int Dc, int Hc, int Wc, int D, int H, int W, void* stream) {
    STX_REQUIRE(gout && cost && disp && stats && gcost && workspace && B > 0, "head_bwd: null operand");
    hipStream_t st = (hipStream_t)stream;
    // (an LDS-staged version of this pass, like the forward kernel's, measured slower: 93.1 -> 102.7 us, 50 KB of LDS per
    //  workgroup leave 3 waves per SIMD where this one runs 8 -- round 2, call P)
    hipLaunchKernelGGL(head_bwd_pix_kernel<AC>, dim3(stx_cdiv(W, HD_THREADS), H, B), dim3(HD_THREADS), 0, st, gout, cost,
                       disp, stats, workspace, Dc, Hc, Wc, D, H, W);
    int rc = stx_check_launch("head_bwd(pixels)");
    if (rc) return rc;
    // footprint of a cost cell in output pixels: 2/scale (+ slack for the border clamps)
    int fh, fw;
    if (AC) {
        fh = Hc > 1 ? 2 * stx_cdiv(H - 1, Hc - 1) + 3 : H;
        fw = Wc > 1 ? 2 * stx_cdiv(W - 1, Wc - 1) + 3 : W;
    } else {
        fh = 2 * stx_cdiv(H, Hc) + 3;
        fw = 2 * stx_cdiv(W, Wc) + 3;
    }
    if (fh <= HD_GF && fw <= 13 && W % 4 == 0 && W >= 16)
        hipLaunchKernelGGL(head_bwd_gather4_kernel<AC>, dim3(stx_cdiv(Wc, HD_THREADS), Dc * Hc, B), dim3(HD_THREADS), 0, st,
                           workspace, gcost, Dc, Hc, Wc, H, W, fh);
    else
        hipLaunchKernelGGL(head_bwd_gather_kernel<AC>, dim3(stx_cdiv(Wc, HD_THREADS), Dc * Hc, B), dim3(HD_THREADS), 0, st,
                           workspace, gcost, Dc, Hc, Wc, H, W, fh, fw);
    return stx_check_launch("head_bwd(gather)");
}

extern "C" int stx_head_bwd(const float* gout, const float* cost, const float* disp, const float* stats,
                            float* gcost, float* workspace, int B, int Dc, int Hc, int Wc, int D, int H, int W,
                            void* stream) {
    stx_begin();
    return head_bwd_launch<false>(gout, cost, disp, stats, gcost, workspace, B, Dc, Hc, Wc, D, H, W, stream);
}

extern "C" int stx_head_bwd2(const float* gout, const float* cost, const float* disp, const float* stats,
                             float* gcost, float* workspace, int B, int Dc, int Hc, int Wc, int D, int H, int W,
                             int align_corners, void* stream) {
    stx_begin();
    return align_corners
               ? head_bwd_launch<true>(gout, cost, disp, stats, gcost, workspace, B, Dc, Hc, Wc, D, H, W, stream)
               : head_bwd_launch<false>(gout, cost, disp, stats, gcost, workspace, B, Dc, Hc, Wc, D, H, W, stream);
}

extern "C" int stx_softargmax_fwd(const float* x, float* out, int B, int D, int HW, void* stream) {
    stx_begin();
    STX_REQUIRE(x && out && B > 0 && D > 0 && HW > 0, "softargmax_fwd: bad shape");
    if (HW % 4 == 0 && ((size_t)x & 15) == 0 && ((size_t)out & 15) == 0)
        hipLaunchKernelGGL(softargmax4_kernel, dim3(stx_cdiv(HW / 4, HD_THREADS), B), dim3(HD_THREADS), 0,
                           (hipStream_t)stream, x, out, D, HW);
    else
        hipLaunchKernelGGL(softargmax_kernel, dim3(stx_cdiv(HW, HD_THREADS), B), dim3(HD_THREADS), 0,
                           (hipStream_t)stream, x, out, D, HW);
    return stx_check_launch("softargmax_fwd");
}

extern "C" int stx_argmax_fwd(const float* x, long long* out, int B, int D, int HW, void* stream) {
    stx_begin();
    STX_REQUIRE(x && out && B > 0 && D > 0 && HW > 0, "argmax_fwd: bad shape");
    hipLaunchKernelGGL(argmax_kernel, dim3(stx_cdiv(HW, HD_THREADS), B), dim3(HD_THREADS), 0, (hipStream_t)stream, x,
                       out, D, HW);
    return stx_check_launch("argmax_fwd");
}

extern "C" int stx_softmax_d_fwd(const float* x, float* y, int B, int D, int HW, void* stream) {
    stx_begin();
    STX_REQUIRE(x && y && B > 0 && D > 0 && HW > 0, "softmax_d_fwd: bad shape");
    hipLaunchKernelGGL(softmax_d_kernel, dim3(stx_cdiv(HW, HD_THREADS), B), dim3(HD_THREADS), 0, (hipStream_t)stream,
                       x, y, D, HW);
    return stx_check_launch("softmax_d_fwd");
}
