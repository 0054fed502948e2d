// 2-D helpers of the PCWNet / CFNet family on gfx950 (reference /root/reference/stereo_toolbox):
//   warp(x, disp)                                         models/PCWNet/submodule.py:137-176
//   build_corrleation_volume(ref, tgt, maxdisp, groups)   models/PCWNet/submodule.py:121-135 (dup CFNet/submodule.py:181-195)
//   disparity_variance(x, maxdisp, disparity)             models/CFNet/submodule.py:128-134
//   disparity_variance_confidence(x, samples, disparity)  models/CFNet/submodule.py:136-140
// Callers: PCWNet's refinement input (pcwnet.py:465-466, 498-499: full-resolution 32-channel maps, +-24 disparities, one
// group) and CFNet's search-range machinery (cfnet.py:540, 568).  All tensors fp32, NCHW / [B][D][HW] as in the reference.
//
// Every kernel here is HBM-bound streaming work (roofline: the operands once); none of it is GEMM-shaped:
//   * warp: one thread per pixel, the four bilinear corners / weights / validity computed once and reused for all C channels
//     (the reference runs grid_sample twice -- once on a tensor of ones for the mask -- and multiplies);
//   * correlation volume: the reference fills 2*maxdisp+1 slices with one (slice, mul, view, mean, strided copy) chain each,
//     re-reading both maps per slice; here a workgroup stages the target rows of a 128-column tile (+ maxdisp halo) in LDS
//     once, a lane owns two adjacent columns and keeps their maxdisp+1 sums in registers (the window of a channel row is
//     shared by both columns), and writes every slice of the tile itself -- zeros included, there is no memset;
//     the literal semantics of the negative slices (`ref[..., :-i]` with negative i = the FIRST |i| columns against the LAST
//     |i| target columns) touch maxdisp columns per row and run as a small second kernel;
//   * variance: one thread per pixel, one pass over the D probabilities.
#include "stx_common.h"

namespace {

constexpr int R2_THREADS = 256;

// ------------------------------------------------------------------------------------------------ warp
struct WarpTap {
    float w[4];        // bilinear weights nw, ne, sw, se (torch's grid_sample formulas)
    int off[4];        // element offsets inside one channel plane; -1 = outside the image
    float mask;        // 1 if the in-image weights sum to >= 0.999 (PCWNet/submodule.py:171-174), else 0; NaN for a NaN disparity
                       // (the reference's `mask < 0.999` and `mask > 0` are both false there: output * mask stays NaN)
    float dwx[4];      // d weight / d ix
};

__device__ __forceinline__ WarpTap warp_tap(float disp, int x, int y, int H, int W) {
    // PCWNet/submodule.py:158-166: grid normalised with (W-1) / (H-1), then grid_sample's default align_corners=False
    const float wm = (float)(W > 1 ? W - 1 : 1), hm = (float)(H > 1 ? H - 1 : 1);
    const float gx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, __fsub_rn((float)x, disp)), wm), 1.0f);
    const float gy = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, (float)y), hm), 1.0f);
    const float ix = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), (float)W), 1.0f), 0.5f);
    const float iy = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), (float)H), 1.0f), 0.5f);
    const float fx = floorf(ix), fy = floorf(iy);
    const float ax = __fsub_rn(__fadd_rn(fx, 1.0f), ix), bx = __fsub_rn(ix, fx);      // ix_se - ix, ix - ix_nw
    const float ay = __fsub_rn(__fadd_rn(fy, 1.0f), iy), by = __fsub_rn(iy, fy);
    WarpTap t;
    t.w[0] = __fmul_rn(ax, ay); t.w[1] = __fmul_rn(bx, ay); t.w[2] = __fmul_rn(ax, by); t.w[3] = __fmul_rn(bx, by);
    t.dwx[0] = -ay; t.dwx[1] = ay; t.dwx[2] = -by; t.dwx[3] = by;
    // (coordinates far outside the image: the comparison below fails for every corner; clamp before the int conversion)
    const float cfx = fminf(fmaxf(fx, -4.0f), (float)W + 4.0f), cfy = fminf(fmaxf(fy, -4.0f), (float)H + 4.0f);
    const int x0 = (int)cfx, y0 = (int)cfy;
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
        const bool in = xx >= 0 && xx < W && yy >= 0 && yy < H && ix == ix;      // NaN disparities touch no memory
        t.off[k] = in ? yy * W + xx : -1;
        if (in) m = __fadd_rn(m, t.w[k]);
    }
    t.mask = (ix == ix) ? ((m < 0.999f) ? 0.f : 1.f) : ix;     // NaN propagates like in the reference: a diverged
    return t;                                                  // refinement must surface as a NaN loss, not be zeroed silently
}

__global__ __launch_bounds__(R2_THREADS) void warp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ disp,
                                                             float* __restrict__ out, int C, int H, int W) {
    const int HW = H * W;
    const int i = blockIdx.x * R2_THREADS + threadIdx.x, b = blockIdx.y;
    if (i >= HW) return;
    const WarpTap t = warp_tap(disp[(size_t)b * HW + i], i % W, i / W, H, W);
    const float* xp = x + (size_t)b * C * HW;
    float* op = out + (size_t)b * C * HW + i;
    if (!(t.mask == 1.f)) {                                     // masked (0) or NaN disparity (NaN): no sampling
        for (int c = 0; c < C; ++c) op[(size_t)c * HW] = t.mask;
        return;
    }
    for (int c = 0; c < C; ++c) {
        const float* p = xp + (size_t)c * HW;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (t.off[k] >= 0) a = fmaf(p[t.off[k]], t.w[k], a);
        op[(size_t)c * HW] = a;
    }
}

// gx must be zero on entry (the host wrapper clears it): the corner updates are float atomics, like torch's own
// grid_sampler backward.  gdisp = - W/(W-1) * sum_c gout * d out / d ix (mask constant, PCWNet/submodule.py:171-174).
__global__ __launch_bounds__(R2_THREADS) void warp_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                             const float* __restrict__ disp, float* __restrict__ gx,
                                                             float* __restrict__ gdisp, int C, int H, int W) {
    const int HW = H * W;
    const int i = blockIdx.x * R2_THREADS + threadIdx.x, b = blockIdx.y;
    if (i >= HW) return;
    const WarpTap t = warp_tap(disp[(size_t)b * HW + i], i % W, i / W, H, W);
    float gix = (t.mask == t.mask) ? 0.f : t.mask;           // NaN disparity: NaN disparity gradient, nothing added to gx
    if (t.mask == 1.f) {
        const float* xp = x + (size_t)b * C * HW;
        float* gxp = gx ? gx + (size_t)b * C * HW : nullptr;
        const float* gp = gout + (size_t)b * C * HW + i;
        for (int c = 0; c < C; ++c) {
            const float g = gp[(size_t)c * HW];
            const float* p = xp + (size_t)c * HW;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t.off[k] >= 0) {
                    gix = fmaf(g * p[t.off[k]], t.dwx[k], gix);
                    if (gxp) atomicAdd(gxp + (size_t)c * HW + t.off[k], g * t.w[k]);
                }
        }
    }
    if (gdisp) {
        // d ix / d disp = -(2 / (W-1)) * (W / 2)
        const float wm = (float)(W > 1 ? W - 1 : 1);
        gdisp[(size_t)b * HW + i] = -gix * ((float)W / wm);
    }
}

// ------------------------------------------------------------------------------------------------ correlation volume
constexpr int CV2_TW = 128;            // columns per workgroup (one wave, two adjacent columns per lane)
constexpr int CV2_CH = 32;             // channels staged per pass

// slices maxdisp .. 2*maxdisp (i >= 0): vol[b][g][md+i][h][w] = (w >= i) ? mean_c ref[c][w] * tgt[c][w-i] : 0; the slices
// 0 .. maxdisp-1 (i < 0) are zero-filled here and their first |i| columns written by corr_neg_fwd_kernel afterwards.
template <int MD>
__global__ __launch_bounds__(64) void corr_fwd_kernel(const float* __restrict__ ref, const float* __restrict__ tgt,
                                                     float* __restrict__ vol, int C, int H, int W, int md, int G) {
    constexpr int MDE = (MD + 1) & ~1;                     // even halo: the lane's window starts 8-byte aligned
    constexpr int PITCH = CV2_TW + MDE;
    __shared__ float trow[CV2_CH][PITCH];
    const int tid = threadIdx.x, w0 = blockIdx.x * CV2_TW, h = blockIdx.y, b = blockIdx.z;
    const int cpg = C / G, HW = H * W, ND = 2 * md + 1;
    const int wa = w0 + 2 * tid;                             // the lane's two columns: wa, wa + 1
    const float inv = 1.0f / (float)cpg;
    for (int g = 0; g < G; ++g) {
        float acc0[MD + 1], acc1[MD + 1];
#pragma unroll
        for (int i = 0; i <= MD; ++i) acc0[i] = acc1[i] = 0.f;
        for (int c0 = 0; c0 < cpg; c0 += CV2_CH) {
            const int nc = (cpg - c0 < CV2_CH) ? cpg - c0 : CV2_CH;
            __syncthreads();
            for (int e = tid; e < nc * PITCH; e += 64) {
                const int cc = e / PITCH, k = e - cc * PITCH, col = w0 - MDE + k;
                trow[cc][k] = (col >= 0 && col < W) ? tgt[((size_t)(b * C + g * cpg + c0 + cc) * H + h) * W + col] : 0.f;
            }
            __syncthreads();
            for (int cc = 0; cc < nc; ++cc) {
                const float* rp = ref + ((size_t)(b * C + g * cpg + c0 + cc) * H + h) * W;
                const float l0 = wa < W ? rp[wa] : 0.f, l1 = wa + 1 < W ? rp[wa + 1] : 0.f;
                const float* tw = &trow[cc][2 * tid];       // tw[MDE + p - i] = tgt[wa + p - i]
#pragma unroll
                for (int i = 0; i <= MD; ++i) {
                    acc0[i] = fmaf(l0, tw[MDE - i], acc0[i]);
                    acc1[i] = fmaf(l1, tw[MDE + 1 - i], acc1[i]);
                }
            }
        }
        float* vp = vol + ((size_t)(b * G + g) * ND * H + h) * W;
#pragma unroll
        for (int i = 0; i <= MD; ++i) {
            if (i > md) break;
            float* q = vp + (size_t)(md + i) * HW;
            if (wa < W) q[wa] = acc0[i] * inv;              // (columns < i met zero-padded target columns: already 0)
            if (wa + 1 < W) q[wa + 1] = acc1[i] * inv;
        }
        for (int s = 0; s < md; ++s) {
            float* q = vp + (size_t)s * HW;
            if (wa < W) q[wa] = 0.f;
            if (wa + 1 < W) q[wa + 1] = 0.f;
        }
    }
}

// slices 0 .. maxdisp-1 (i = -n): vol[b][g][md-n][h][w] = mean_c ref[c][w] * tgt[c][W-n+w] for w < n (submodule.py:127-130)
__global__ __launch_bounds__(R2_THREADS) void corr_neg_fwd_kernel(const float* __restrict__ ref, const float* __restrict__ tgt,
                                                                 float* __restrict__ vol, int C, int H, int W, int md, int G) {
    const int h = blockIdx.x, b = blockIdx.y, cpg = C / G, HW = H * W, ND = 2 * md + 1;
    const int per_g = md * (md + 1) / 2;
    for (int e = threadIdx.x; e < G * per_g; e += R2_THREADS) {
        const int g = e / per_g;
        int r = e - g * per_g, n = 1;
        while (r >= n) { r -= n; ++n; }                      // item r of the n-th slice: w = r < n
        const int w = r;
        float a = 0.f;
        for (int cc = 0; cc < cpg; ++cc) {
            const size_t row = ((size_t)(b * C + g * cpg + cc) * H + h) * W;
            a = fmaf(ref[row + w], tgt[row + W - n + w], a);
        }
        vol[((size_t)(b * G + g) * ND + (md - n)) * HW + (size_t)h * W + w] = a / (float)cpg;
    }
}

// Backward of the i >= 0 slices:
//   gref[c][w] = 1/cpg * sum_{i <= min(md, w)}      gvol[md+i][w]   * tgt[c][w-i]
//   gtgt[c][x] = 1/cpg * sum_{i <= md, x+i < W}     gvol[md+i][x+i] * ref[c][x+i]
template <int MD>
__global__ __launch_bounds__(64) void corr_bwd_kernel(const float* __restrict__ gvol, const float* __restrict__ ref,
                                                     const float* __restrict__ tgt, float* __restrict__ gref,
                                                     float* __restrict__ gtgt, int C, int H, int W, int md, int G) {
    constexpr int MDE = (MD + 1) & ~1;
    constexpr int PITCH = CV2_TW + MDE;
    __shared__ float row[CV2_CH][PITCH];
    __shared__ float gtile[MD + 1][PITCH];
    const int tid = threadIdx.x, w0 = blockIdx.x * CV2_TW, h = blockIdx.y, b = blockIdx.z;
    const int cpg = C / G, HW = H * W, ND = 2 * md + 1;
    const int wa = w0 + 2 * tid;
    const float inv = 1.0f / (float)cpg;
    for (int g = 0; g < G; ++g) {
        const float* gp = gvol + ((size_t)(b * G + g) * ND * H + h) * W;
        // the gradient tile with a forward halo: gtile[i][k] = gvol[md+i][w0 + k], k < TW + MDE
        __syncthreads();
        for (int e = tid; e < (MD + 1) * PITCH; e += 64) {
            const int i = e / PITCH, k = e - i * PITCH, col = w0 + k;
            gtile[i][k] = (i <= md && col < W) ? gp[(size_t)(md + i) * HW + col] : 0.f;
        }
        __syncthreads();
        float gs0[MD + 1], gs1[MD + 1], gd0[MD + 1], gd1[MD + 1];
#pragma unroll
        for (int i = 0; i <= MD; ++i) {
            gd0[i] = gtile[i][2 * tid];          gd1[i] = gtile[i][2 * tid + 1];            // straight: gvol[md+i][w]
            gs0[i] = gtile[i][2 * tid + i];      gs1[i] = gtile[i][2 * tid + 1 + i];        // sheared:  gvol[md+i][x+i]
        }
        for (int c0 = 0; c0 < cpg; c0 += CV2_CH) {
            const int nc = (cpg - c0 < CV2_CH) ? cpg - c0 : CV2_CH;
            if (gref) {
                __syncthreads();
                for (int e = tid; e < nc * PITCH; e += 64) {           // target rows with a backward halo
                    const int cc = e / PITCH, k = e - cc * PITCH, col = w0 - MDE + k;
                    row[cc][k] = (col >= 0 && col < W) ? tgt[((size_t)(b * C + g * cpg + c0 + cc) * H + h) * W + col] : 0.f;
                }
                __syncthreads();
                for (int cc = 0; cc < nc; ++cc) {
                    const float* tw = &row[cc][2 * tid];
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int i = 0; i <= MD; ++i) {
                        a0 = fmaf(gd0[i], tw[MDE - i], a0);
                        a1 = fmaf(gd1[i], tw[MDE + 1 - i], a1);
                    }
                    float* q = gref + ((size_t)(b * C + g * cpg + c0 + cc) * H + h) * W;
                    if (wa < W) q[wa] = a0 * inv;
                    if (wa + 1 < W) q[wa + 1] = a1 * inv;
                }
            }
            if (gtgt) {
                __syncthreads();
                for (int e = tid; e < nc * PITCH; e += 64) {           // reference rows with a forward halo
                    const int cc = e / PITCH, k = e - cc * PITCH, col = w0 + k;
                    row[cc][k] = col < W ? ref[((size_t)(b * C + g * cpg + c0 + cc) * H + h) * W + col] : 0.f;
                }
                __syncthreads();
                for (int cc = 0; cc < nc; ++cc) {
                    const float* rw = &row[cc][2 * tid];
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int i = 0; i <= MD; ++i) {
                        a0 = fmaf(gs0[i], rw[i], a0);
                        a1 = fmaf(gs1[i], rw[1 + i], a1);
                    }
                    float* q = gtgt + ((size_t)(b * C + g * cpg + c0 + cc) * H + h) * W;
                    if (wa < W) q[wa] = a0 * inv;
                    if (wa + 1 < W) q[wa + 1] = a1 * inv;
                }
            }
        }
    }
}

// Backward of the i < 0 slices, accumulated onto what corr_bwd_kernel wrote (launched behind it on the same stream; every
// (channel, column) is owned by exactly one thread -> plain read-modify-write, deterministic):
//   gref[c][w]       += 1/cpg * sum_{n = w+1 .. md}   gvol[md-n][w]       * tgt[c][W-n+w]        (w < md)
//   gtgt[c][W-md+j]  += 1/cpg * sum_{n = md-j .. md}  gvol[md-n][j-md+n]  * ref[c][j-md+n]       (j < md)
__global__ __launch_bounds__(R2_THREADS) void corr_neg_bwd_kernel(const float* __restrict__ gvol, const float* __restrict__ ref,
                                                                 const float* __restrict__ tgt, float* __restrict__ gref,
                                                                 float* __restrict__ gtgt, int C, int H, int W, int md, int G) {
    const int h = blockIdx.x, b = blockIdx.y, cpg = C / G, HW = H * W, ND = 2 * md + 1;
    for (int e = threadIdx.x; e < 2 * C * md; e += R2_THREADS) {
        const int side = e / (C * md), r = e - side * C * md, c = r / md, j = r - c * md, g = c / cpg;
        const float* gp = gvol + ((size_t)(b * G + g) * ND * H + h) * W;
        const size_t rowo = ((size_t)(b * C + c) * H + h) * W;
        float a = 0.f;
        if (side == 0) {
            if (!gref) continue;
            const int w = j;
            for (int n = w + 1; n <= md; ++n) a = fmaf(gp[(size_t)(md - n) * HW + w], tgt[rowo + W - n + w], a);
            gref[rowo + w] += a / (float)cpg;
        } else {
            if (!gtgt) continue;
            const int x = W - md + j;
            for (int n = md - j; n <= md; ++n) {
                const int w = x - W + n;
                a = fmaf(gp[(size_t)(md - n) * HW + w], ref[rowo + w], a);
            }
            gtgt[rowo + x] += a / (float)cpg;
        }
    }
}

// ------------------------------------------------------------------------------------------------ variance
// out[b][i] = sum_d x[b][d][i] * (v_d - disp[b][i])^2, v_d = d (disparity_variance) or samples[b][d][i] (_confidence)
__global__ __launch_bounds__(R2_THREADS) void variance_fwd_kernel(const float* __restrict__ x, const float* __restrict__ disp,
                                                                 const float* __restrict__ samples, float* __restrict__ out,
                                                                 int D, int HW) {
    const int i = blockIdx.x * R2_THREADS + threadIdx.x, b = blockIdx.y;
    if (i >= HW) return;
    const float mu = disp[(size_t)b * HW + i];
    const float* xp = x + (size_t)b * D * HW + i;
    const float* sp = samples ? samples + (size_t)b * D * HW + i : nullptr;
    float a = 0.f;
    for (int d = 0; d < D; ++d) {
        // reference: (disp_values - disparity) ** 2 resp. (disparity - disparity_samples) ** 2, then sum(x * that)
        const float e = sp ? mu - sp[(size_t)d * HW] : (float)d - mu;
        a = __fadd_rn(a, __fmul_rn(xp[(size_t)d * HW], __fmul_rn(e, e)));
    }
    out[(size_t)b * HW + i] = a;
}

// gx[d] = g * e_d^2; gdisp = g * sum_d x_d * 2 e_d * (de/ddisp); gsamples[d] = g * x_d * 2 e_d * (de/dsample)
__global__ __launch_bounds__(R2_THREADS) void variance_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                 const float* __restrict__ disp, const float* __restrict__ samples,
                                                                 float* __restrict__ gx, float* __restrict__ gdisp,
                                                                 float* __restrict__ gsamples, int D, int HW) {
    const int i = blockIdx.x * R2_THREADS + threadIdx.x, b = blockIdx.y;
    if (i >= HW) return;
    const float mu = disp[(size_t)b * HW + i], gg = g[(size_t)b * HW + i];
    const size_t base = (size_t)b * D * HW + i;
    float s = 0.f;
    for (int d = 0; d < D; ++d) {
        const size_t o = base + (size_t)d * HW;
        const float xv = x[o];
        const float e = samples ? mu - samples[o] : (float)d - mu;
        if (gx) gx[o] = gg * e * e;
        s = fmaf(xv, e, s);
        if (gsamples) gsamples[o] = -2.0f * gg * xv * e;
    }
    if (gdisp) gdisp[(size_t)b * HW + i] = (samples ? 2.0f : -2.0f) * gg * s;
}

template <int MD>
void corr_launch(bool bwd, const float* a0, const float* a1, const float* a2, float* o0, float* o1, int B, int C, int H, int W,
                 int md, int G, hipStream_t st) {
    const dim3 grid(stx_cdiv(W, CV2_TW), H, B);
    if (!bwd) hipLaunchKernelGGL((corr_fwd_kernel<MD>), grid, dim3(64), 0, st, a0, a1, o0, C, H, W, md, G);
    else hipLaunchKernelGGL((corr_bwd_kernel<MD>), grid, dim3(64), 0, st, a0, a1, a2, o0, o1, C, H, W, md, G);
}

}  // namespace

extern "C" int stx_warp_fwd(const float* x, const float* disp, float* out, int B, int C, int H, int W, void* stream) {
    stx_begin();
    STX_REQUIRE(x && disp && out && B > 0 && C > 0 && H > 0 && W > 0, "warp_fwd: bad arguments");
    STX_REQUIRE((long long)C * H * W < (1ll << 31), "warp_fwd: one sample exceeds 2^31 elements");
    hipLaunchKernelGGL(warp_fwd_kernel, dim3(stx_cdiv(H * W, R2_THREADS), B), dim3(R2_THREADS), 0, (hipStream_t)stream, x, disp,
                       out, C, H, W);
    return stx_check_launch("warp_fwd");
}

extern "C" int stx_warp_bwd(const float* gout, const float* x, const float* disp, float* gx, float* gdisp, int B, int C, int H,
                            int W, void* stream) {
    stx_begin();
    STX_REQUIRE(gout && x && disp && (gx || gdisp) && B > 0 && C > 0 && H > 0 && W > 0, "warp_bwd: bad arguments");
    STX_REQUIRE((long long)C * H * W < (1ll << 31), "warp_bwd: one sample exceeds 2^31 elements");
    hipStream_t st = (hipStream_t)stream;
    if (gx && hipMemsetAsync(gx, 0, (size_t)B * C * H * W * sizeof(float), st) != hipSuccess)
        return stx_set_error(STX_ERR_LAUNCH, "warp_bwd: clearing the gradient buffer failed");
    hipLaunchKernelGGL(warp_bwd_kernel, dim3(stx_cdiv(H * W, R2_THREADS), B), dim3(R2_THREADS), 0, st, gout, x, disp, gx, gdisp,
                       C, H, W);
    return stx_check_launch("warp_bwd");
}

extern "C" int stx_corr_volume_fwd(const float* ref, const float* tgt, float* vol, int B, int C, int H, int W, int maxdisp,
                                   int groups, void* stream) {
    stx_begin();
    STX_REQUIRE(ref && tgt && vol && B > 0 && C > 0 && H > 0 && W > 0 && groups > 0, "corr_volume_fwd: bad arguments");
    STX_REQUIRE(C % groups == 0, "corr_volume_fwd: %d channels do not split into %d groups", C, groups);
    STX_REQUIRE(maxdisp >= 0 && maxdisp <= 48, "corr_volume_fwd: maxdisp %d outside [0, 48]", maxdisp);
    // (the reference's negative slices pair the first |i| with the last |i| columns: they need |i| <= W)
    STX_REQUIRE(maxdisp <= W, "corr_volume_fwd: maxdisp %d exceeds the width %d", maxdisp, W);
    hipStream_t st = (hipStream_t)stream;
    if (maxdisp <= 8) corr_launch<8>(false, ref, tgt, nullptr, vol, nullptr, B, C, H, W, maxdisp, groups, st);
    else if (maxdisp <= 24) corr_launch<24>(false, ref, tgt, nullptr, vol, nullptr, B, C, H, W, maxdisp, groups, st);
    else corr_launch<48>(false, ref, tgt, nullptr, vol, nullptr, B, C, H, W, maxdisp, groups, st);
    if (maxdisp > 0)
        hipLaunchKernelGGL(corr_neg_fwd_kernel, dim3(H, B), dim3(R2_THREADS), 0, st, ref, tgt, vol, C, H, W, maxdisp, groups);
    return stx_check_launch("corr_volume_fwd");
}

extern "C" int stx_corr_volume_bwd(const float* gvol, const float* ref, const float* tgt, float* gref, float* gtgt, int B, int C,
                                   int H, int W, int maxdisp, int groups, void* stream) {
    stx_begin();
    STX_REQUIRE(gvol && ref && tgt && (gref || gtgt) && B > 0 && C > 0 && H > 0 && W > 0 && groups > 0,
                "corr_volume_bwd: bad arguments");
    STX_REQUIRE(C % groups == 0, "corr_volume_bwd: %d channels do not split into %d groups", C, groups);
    STX_REQUIRE(maxdisp >= 0 && maxdisp <= 48 && maxdisp <= W, "corr_volume_bwd: maxdisp %d outside [0, min(48, W)]", maxdisp);
    hipStream_t st = (hipStream_t)stream;
    if (maxdisp <= 8) corr_launch<8>(true, gvol, ref, tgt, gref, gtgt, B, C, H, W, maxdisp, groups, st);
    else if (maxdisp <= 24) corr_launch<24>(true, gvol, ref, tgt, gref, gtgt, B, C, H, W, maxdisp, groups, st);
    else corr_launch<48>(true, gvol, ref, tgt, gref, gtgt, B, C, H, W, maxdisp, groups, st);
    if (maxdisp > 0)
        hipLaunchKernelGGL(corr_neg_bwd_kernel, dim3(H, B), dim3(R2_THREADS), 0, st, gvol, ref, tgt, gref, gtgt, C, H, W, maxdisp,
                           groups);
    return stx_check_launch("corr_volume_bwd");
}

extern "C" int stx_disparity_variance_fwd(const float* x, const float* disp, const float* samples, float* out, int B, int D,
                                          int HW, void* stream) {
    stx_begin();
    STX_REQUIRE(x && disp && out && B > 0 && D > 0 && HW > 0, "disparity_variance_fwd: bad arguments");
    hipLaunchKernelGGL(variance_fwd_kernel, dim3(stx_cdiv(HW, R2_THREADS), B), dim3(R2_THREADS), 0, (hipStream_t)stream, x, disp,
                       samples, out, D, HW);
    return stx_check_launch("disparity_variance_fwd");
}

extern "C" int stx_disparity_variance_bwd(const float* g, const float* x, const float* disp, const float* samples, float* gx,
                                          float* gdisp, float* gsamples, int B, int D, int HW, void* stream) {
    stx_begin();
    STX_REQUIRE(g && x && disp && (gx || gdisp || gsamples) && B > 0 && D > 0 && HW > 0, "disparity_variance_bwd: bad arguments");
    STX_REQUIRE(!gsamples || samples, "disparity_variance_bwd: a sample gradient without samples");
    hipLaunchKernelGGL(variance_bwd_kernel, dim3(stx_cdiv(HW, R2_THREADS), B), dim3(R2_THREADS), 0, (hipStream_t)stream, g, x,
                       disp, samples, gx, gdisp, gsamples, D, HW);
    return stx_check_launch("disparity_variance_bwd");
}
