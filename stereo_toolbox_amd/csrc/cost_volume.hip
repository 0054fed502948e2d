// 4-D cost-volume builders for gfx950: group-wise correlation + concat volume, fused,
// written straight into one channels-last (NDHWC) buffer.
//
// Replaces (reference, /root/reference/stereo_toolbox/models):
//   build_gwc_volume      GwcNet/submodule.py:53-63   (dup ACVNet/submodule.py:228-238)
//   groupwise_correlation GwcNet/submodule.py:44-50
//   build_concat_volume   GwcNet/submodule.py:30-41   (left half zeroed where w<d)
//                         PSMNet/stackhourglass.py:111-120 (inline twin)
//                         ACVNet/submodule.py:180-191 (left half NOT masked -> mask_left=0)
//   torch.cat((gwc, concat), 1)  GwcNet/gwcnet.py:180  (fused: both halves land in one buffer)
//   softmax(att, dim=2) * concat_volume  ACVNet/acv.py:196 (optional `scale` operand)
//
// vol[b,d,h,w, 0:G]        = w>=d ? mean_{c in group g} L[b,c,h,w] * R[b,c,h,w-d] : 0
// vol[b,d,h,w, G:G+Cc]     = (w>=d || !mask_left) ? Lc[b,c,h,w] : 0
// vol[b,d,h,w, G+Cc:G+2Cc] = w>=d ? Rc[b,c,h,w-d] : 0
//
// Roofline: HBM.  Algorithmic bytes = read L,R (+Lc,Rc) once + write vol once
// (516 464 640 B for GwcNet_GC at 576x960, D'=48).
//
// Mapping.  One workgroup owns (b, h, 16 output columns, 16 disparities).  The
// NCHW feature rows are transposed on the way into LDS ([column][channel], row
// stride C+4 dwords so both the 4-byte transposing writes and the 16-byte reads
// spread over the banks).  Each work item is one (column, 4-channel quad) of the
// output voxel: 16 quads of one voxel are 16 consecutive lanes, so a wave stores
// 1 KiB of contiguous NDHWC bytes per instruction.  The left operand of a
// gwc quad (4 groups x cpg channels) stays in registers across the disparity loop;
// the D-shifted right operand comes from the LDS tile.
#include "stx_common.h"

namespace {

constexpr int CV_WT = 16;       // output columns per workgroup
constexpr int CV_DC = 16;       // disparities per workgroup
constexpr int CV_THREADS = 256;

// Transposing stage of `ncols` columns x `C` channels from an NCHW row into lds[col][C+4].
// Lane l of a wave covers column (l>>2)&7 and channel (l&3)+4*(l>>5) of an 8x8 patch:
// 32 B contiguous per channel row from global, 32 distinct banks per half-wave into LDS.
__device__ __forceinline__ void cv_stage_rows(const float* __restrict__ src,  // &F[b][0][h][0]
                                              int C, int HW, int W, int x_first, int ncols,
                                              float* lds, int tid) {
    const int lane = tid & 63, wave = tid >> 6, nwaves = CV_THREADS >> 6;
    const int cl = (lane & 3) + 4 * (lane >> 5);
    const int xl = (lane >> 2) & 7;
    const int RS = C + 4;
    const int ncol8 = (ncols + 7) >> 3;
    const int nc8 = (C + 7) >> 3;
    for (int p = wave; p < ncol8 * nc8; p += nwaves) {
        const int c = (p / ncol8) * 8 + cl;
        const int col = (p % ncol8) * 8 + xl;
        const int x = x_first + col;
        if (c < C && col < ncols) {
            float v = 0.f;
            if (x >= 0 && x < W) v = src[(size_t)c * HW + x];
            lds[col * RS + c] = v;
        }
    }
}

template <int CPG>
__global__ __launch_bounds__(CV_THREADS) void cost_volume_fwd_kernel(
    const float* __restrict__ Lg, const float* __restrict__ Rg, int Cg, int G,
    const float* __restrict__ Lc, const float* __restrict__ Rc, int Cc,
    const float* __restrict__ scale, float* __restrict__ vol,
    int H, int W, int D, int mask_left) {
    STX_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int w0 = blockIdx.x * CV_WT;
    const int d0 = blockIdx.y * CV_DC;
    const int bh = blockIdx.z;
    const int b = bh / H, h = bh % H;
    const int HW = H * W;
    const int CT = G + 2 * Cc;
    const int Q = CT >> 2, GQ = G >> 2, CQ = Cc >> 2;
    const int RSg = Cg + 4, RSc = Cc + 4;
    const int NR = CV_WT + CV_DC - 1;                 // right-feature columns kept in LDS
    const int x_first = w0 - d0 - (CV_DC - 1);        // image column of LDS column 0

    float* Lg_s = reinterpret_cast<float*>(smem);
    float* Rg_s = Lg_s + (G ? CV_WT * RSg : 0);
    float* Lc_s = Rg_s + (G ? NR * RSg : 0);
    float* Rc_s = Lc_s + (Cc ? CV_WT * RSc : 0);

    if (G) {
        cv_stage_rows(Lg + ((size_t)b * Cg * H + h) * W, Cg, HW, W, w0, CV_WT, Lg_s, tid);
        cv_stage_rows(Rg + ((size_t)b * Cg * H + h) * W, Cg, HW, W, x_first, NR, Rg_s, tid);
    }
    if (Cc) {
        cv_stage_rows(Lc + ((size_t)b * Cc * H + h) * W, Cc, HW, W, w0, CV_WT, Lc_s, tid);
        cv_stage_rows(Rc + ((size_t)b * Cc * H + h) * W, Cc, HW, W, x_first, NR, Rc_s, tid);
    }
    __syncthreads();

    const int dend = (d0 + CV_DC < D) ? CV_DC : (D - d0);
    const float inv = 1.0f / (float)CPG;
    for (int item = tid; item < CV_WT * Q; item += CV_THREADS) {
        const int wl = item / Q, q = item - wl * Q;
        const int w = w0 + wl;
        if (w >= W) continue;
        float* out = vol + ((((size_t)b * D + d0) * H + h) * W + w) * CT + 4 * q;
        const size_t dstride = (size_t)H * W * CT;
        const float* sc = scale ? scale + (((size_t)b * D + d0) * H + h) * W + w : nullptr;
        if (q < GQ) {
            // 4 groups x CPG channels of the left feature stay in registers.
            float4 l[CPG];
#pragma unroll
            for (int j = 0; j < CPG; ++j) l[j] = stx_ld4(Lg_s + wl * RSg + q * 4 * CPG + 4 * j);
            for (int dd = 0; dd < dend; ++dd) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (w >= d0 + dd) {
                    const float* r = Rg_s + (wl + CV_DC - 1 - dd) * RSg + q * 4 * CPG;
                    float acc[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float s = 0.f;
#pragma unroll
                        for (int j = 0; j < CPG / 4; ++j) {
                            const float4 a = l[g * (CPG / 4) + j];
                            const float4 v = stx_ld4(r + (g * (CPG / 4) + j) * 4);
                            s = fmaf(a.x, v.x, s);
                            s = fmaf(a.y, v.y, s);
                            s = fmaf(a.z, v.z, s);
                            s = fmaf(a.w, v.w, s);
                        }
                        acc[g] = s * inv;
                    }
                    o = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    if (sc) { const float m = sc[(size_t)dd * HW]; o.x *= m; o.y *= m; o.z *= m; o.w *= m; }
                }
                stx_st4(out + dd * dstride, o);
            }
        } else if (q < GQ + CQ) {
            const float4 l = stx_ld4(Lc_s + wl * RSc + 4 * (q - GQ));
            for (int dd = 0; dd < dend; ++dd) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!mask_left || w >= d0 + dd) {
                    o = l;
                    if (sc) { const float m = sc[(size_t)dd * HW]; o.x *= m; o.y *= m; o.z *= m; o.w *= m; }
                }
                stx_st4(out + dd * dstride, o);
            }
        } else {
            for (int dd = 0; dd < dend; ++dd) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (w >= d0 + dd) {
                    o = stx_ld4(Rc_s + (wl + CV_DC - 1 - dd) * RSc + 4 * (q - GQ - CQ));
                    if (sc) { const float m = sc[(size_t)dd * HW]; o.x *= m; o.y *= m; o.z *= m; o.w *= m; }
                }
                stx_st4(out + dd * dstride, o);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward of the builders (scatter-free): one workgroup owns (b, h, 16 columns, channel chunk).
//   gLg[c][w] = 1/cpg * sum_{d<=w}      gvol[d][w][g(c)]     * Rg[c][w-d]
//   gRg[c][x] = 1/cpg * sum_{d, x+d<W}  gvol[d][x+d][g(c)]   * Lg[c][x+d]
//   gLc[c][w] =         sum_{d<=w or !mask_left} gvol[d][w][G+c]
//   gRc[c][x] =         sum_{d, x+d<W}  gvol[d][x+d][G+Cc+c]
// Work item = (column, channel); results are transposed through LDS so the NCHW rows are
// written 16 consecutive columns at a time.
constexpr int CVB_CH = 64;   // feature channels per pass

__global__ __launch_bounds__(CV_THREADS) void cost_volume_bwd_kernel(
    const float* __restrict__ gvol, const float* __restrict__ Lg, const float* __restrict__ Rg,
    int Cg, int G, int Cc, float* __restrict__ gLg, float* __restrict__ gRg,
    float* __restrict__ gLc, float* __restrict__ gRc, int H, int W, int D, int mask_left) {
    STX_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int w0 = blockIdx.x * CV_WT;
    const int bh = blockIdx.z;
    const int b = bh / H, h = bh % H;
    const int HW = H * W;
    const int CT = G + 2 * Cc;
    const int cpg = G ? Cg / G : 1;
    const float inv = 1.0f / (float)cpg;
    const int NC = CV_WT + D - 1;            // columns of the shifted operand
    const int RS = CVB_CH + 4;
    float* Ls = reinterpret_cast<float*>(smem);          // Lg[c][w0 .. w0+NC)      -> for gR
    float* Rs = Ls + NC * RS;                            // Rg[c][w0-D+1 .. w0+WT)  -> for gL
    float* Ts = Rs + NC * RS;                            // [2][CVB_CH][WT+1] transpose buffer
    const size_t dstride = (size_t)H * W * CT;
    const float* gv_row = gvol + (((size_t)b * D) * H + h) * W * CT;

    for (int c0 = 0; c0 < Cg; c0 += CVB_CH) {
        const int nch = (Cg - c0 < CVB_CH) ? (Cg - c0) : CVB_CH;
        __syncthreads();
        // stage the chunk's channels, transposed to [col][ch]
        {
            const int lane = tid & 63, wave = tid >> 6;
            const int cl = (lane & 3) + 4 * (lane >> 5), xl = (lane >> 2) & 7;
            const int ncol8 = (NC + 7) >> 3, nc8 = (nch + 7) >> 3;
            for (int p = wave; p < ncol8 * nc8; p += 4) {
                const int c = (p / ncol8) * 8 + cl, col = (p % ncol8) * 8 + xl;
                if (c < nch && col < NC) {
                    const size_t base = (((size_t)b * Cg + c0 + c) * H + h) * W;
                    const int xl_ = w0 + col, xr_ = w0 - (D - 1) + col;
                    Ls[col * RS + c] = (xl_ < W) ? Lg[base + xl_] : 0.f;
                    Rs[col * RS + c] = (xr_ >= 0 && xr_ < W) ? Rg[base + xr_] : 0.f;
                }
            }
        }
        __syncthreads();
        for (int item = tid; item < CV_WT * nch; item += CV_THREADS) {
            const int wl = item / nch, c = item - wl * nch;
            const int w = w0 + wl;
            const int g = (c0 + c) / cpg;
            float aL = 0.f, aR = 0.f;
            if (w < W) {
                for (int d = 0; d < D; ++d) {
                    // gL: voxel (d, w), right column w-d -> Rs col = wl + D-1-d
                    if (w >= d) aL = fmaf(gv_row[d * dstride + (size_t)w * CT + g], Rs[(wl + D - 1 - d) * RS + c], aL);
                    // gR: here w plays x; voxel (d, x+d), left column x+d -> Ls col = wl + d
                    if (w + d < W) aR = fmaf(gv_row[d * dstride + (size_t)(w + d) * CT + g], Ls[(wl + d) * RS + c], aR);
                }
            }
            Ts[c * (CV_WT + 1) + wl] = aL * inv;
            Ts[(CVB_CH + c) * (CV_WT + 1) + wl] = aR * inv;
        }
        __syncthreads();
        for (int item = tid; item < 2 * nch * CV_WT; item += CV_THREADS) {
            const int wl = item % CV_WT, r = item / CV_WT;
            const int which = r / nch, c = r - which * nch;
            const int w = w0 + wl;
            if (w < W) {
                float* dst = which ? gRg : gLg;
                dst[(((size_t)b * Cg + c0 + c) * H + h) * W + w] = Ts[(which * CVB_CH + c) * (CV_WT + 1) + wl];
            }
        }
    }
    // concat halves: plain disparity sums
    for (int item = tid; item < 2 * Cc * CV_WT; item += CV_THREADS) {
        const int wl = item % CV_WT, r = item / CV_WT;
        const int which = r / Cc, c = r - which * Cc;
        const int w = w0 + wl;
        if (w >= W) continue;
        float a = 0.f;
        if (!which) {
            for (int d = 0; d < D; ++d)
                if (!mask_left || w >= d) a += gv_row[d * dstride + (size_t)w * CT + G + c];
            gLc[(((size_t)b * Cc + c) * H + h) * W + w] = a;
        } else {
            for (int d = 0; d < D; ++d)
                if (w + d < W) a += gv_row[d * dstride + (size_t)(w + d) * CT + G + Cc + c];
            gRc[(((size_t)b * Cc + c) * H + h) * W + w] = a;
        }
    }
}

}  // namespace

extern "C" int stx_cost_volume_fwd(const float* Lg, const float* Rg, int Cg, int G, const float* Lc,
                                   const float* Rc, int Cc, const float* scale, float* vol, int B, int H,
                                   int W, int D, int mask_left, void* stream) {
    stx_begin();
    STX_REQUIRE(vol && B > 0 && H > 0 && W > 0 && D > 0, "cost_volume_fwd: bad shape B=%d H=%d W=%d D=%d", B, H, W, D);
    STX_REQUIRE(G >= 0 && Cc >= 0 && (G + Cc) > 0, "cost_volume_fwd: need G>0 or Cc>0");
    STX_REQUIRE(G % 4 == 0 && Cc % 4 == 0, "cost_volume_fwd: G (%d) and Cc (%d) must be multiples of 4", G, Cc);
    if (G) {
        STX_REQUIRE(Lg && Rg, "cost_volume_fwd: gwc features missing");
        STX_REQUIRE(Cg % G == 0, "cost_volume_fwd: C (%d) %% num_groups (%d) != 0", Cg, G);  // submodule.py:46
    }
    if (Cc) STX_REQUIRE(Lc && Rc, "cost_volume_fwd: concat features missing");
    const int cpg = G ? Cg / G : 4;
    STX_REQUIRE(cpg == 4 || cpg == 8 || cpg == 16, "cost_volume_fwd: channels per group %d not in {4,8,16}", cpg);
    const int NR = CV_WT + CV_DC - 1;
    size_t lds = 0;
    if (G) lds += (size_t)(CV_WT + NR) * (Cg + 4) * 4;
    if (Cc) lds += (size_t)(CV_WT + NR) * (Cc + 4) * 4;
    STX_REQUIRE(lds <= 160 * 1024, "cost_volume_fwd: feature tile (%zu B) exceeds LDS", lds);
    dim3 grid(stx_cdiv(W, CV_WT), stx_cdiv(D, CV_DC), B * H);
    hipStream_t st = (hipStream_t)stream;
#define CV_LAUNCH(CPG_)                                                                                       \
    {                                                                                                         \
        if (lds > 64 * 1024)                                                                                  \
            hipFuncSetAttribute((const void*)cost_volume_fwd_kernel<CPG_>,                                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
        hipLaunchKernelGGL(cost_volume_fwd_kernel<CPG_>, grid, dim3(CV_THREADS), lds, st, Lg, Rg, Cg, G, Lc, \
                           Rc, Cc, scale, vol, H, W, D, mask_left);                                           \
    }
    if (cpg == 4) CV_LAUNCH(4) else if (cpg == 8) CV_LAUNCH(8) else CV_LAUNCH(16)
#undef CV_LAUNCH
    return stx_check_launch("cost_volume_fwd");
}

extern "C" int stx_cost_volume_bwd(const float* gvol, const float* Lg, const float* Rg, int Cg, int G, int Cc,
                                   float* gLg, float* gRg, float* gLc, float* gRc, int B, int H, int W, int D,
                                   int mask_left, void* stream) {
    stx_begin();
    STX_REQUIRE(gvol && B > 0 && H > 0 && W > 0 && D > 0, "cost_volume_bwd: bad shape");
    STX_REQUIRE(G % 4 == 0 && Cc % 4 == 0 && (G + Cc) > 0, "cost_volume_bwd: bad channel counts");
    if (G) STX_REQUIRE(Lg && Rg && gLg && gRg && Cg % G == 0, "cost_volume_bwd: gwc operands missing");
    if (Cc) STX_REQUIRE(gLc && gRc, "cost_volume_bwd: concat outputs missing");
    const int NC = CV_WT + D - 1;
    const size_t lds = ((size_t)2 * NC * (CVB_CH + 4) + 2 * CVB_CH * (CV_WT + 1)) * 4;
    STX_REQUIRE(lds <= 160 * 1024, "cost_volume_bwd: D=%d too large for the LDS tile", D);
    dim3 grid(stx_cdiv(W, CV_WT), 1, B * H);
    if (lds > 64 * 1024)
        hipFuncSetAttribute((const void*)cost_volume_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(cost_volume_bwd_kernel, grid, dim3(CV_THREADS), lds, (hipStream_t)stream, gvol, Lg, Rg,
                       G ? Cg : 0, G, Cc, gLg, gRg, gLc, gRc, H, W, D, mask_left);
    return stx_check_launch("cost_volume_bwd");
}
