// 4-D cost-volume builders for gfx950: group-wise correlation + concat volume, fused,
// written straight into one channels-last (NDHWC) buffer -- FIRST-GENERATION kernels, kept as the any-shape fallback of
// the matrix-core builders (cost_volume_mfma.hip / cost_volume_bwd_mfma.hip: D' <= 96 / 48, voxels of <= 64 channels), which
// serve every model configuration; STX_CV_OLD / STX_CVB_OLD route every shape here (tests).  The row-persistent and the
// 8-channels-per-group specialisations of rounds 1-2 were removed in round 3.
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
#include "cost_volume.h"
#include "stx_common.h"
#include <stdlib.h>

namespace {

constexpr int CV_WT = 16;       // output columns per workgroup
constexpr int CV_THREADS = 256;    // backward kernel
constexpr int CVF_THREADS = 1024;  // forward kernel: 16 waves share one feature tile (latency hiding)
constexpr int CV_MAX_DC = 32;   // disparities per workgroup (runtime DC <= this)

// LDS image of a feature tile: [column][C+4 dwords].  A gwc work item (column, quad q) reads the NJ
// 16-byte chunks of its 4*CPG channels; with the plain layout quads 2 apart (NJ=8) land on the same
// 16-B slot of the 256-B bank row (3-way conflicts).  The chunks of quad q are therefore rotated by
// (q / (16/NJ)) inside the quad: slot(column, q, j) covers all 16 slots across the lanes of a
// ds_read_b128 group.  NJ = 0 (concat features) keeps the plain layout.
template <int NJ>
__device__ __forceinline__ int cv_phys_channel(int c) {
    if constexpr (NJ == 0) {
        return c;
    } else {
        constexpr int P = (NJ >= 16) ? 1 : 16 / NJ;
        const int q = c / (4 * NJ), j = (c / 4) % NJ, e = c & 3;
        return q * 4 * NJ + 4 * ((j + q / P) % NJ) + e;
    }
}

// Stage `ncols` columns x `C` channels of one NCHW feature row into lds[col][C+4] (transposed).
// Work is flattened over (channel, column) so all 64 lanes stay busy for any tile width, and every
// lane keeps BATCH independent global loads in flight before the first LDS write (the tile is read
// once per workgroup: latency, not bandwidth, is what this loop has to hide).  Lanes run along the
// image row, so a wave reads contiguous row segments.
template <int NJ, int NTHR>
__device__ __forceinline__ void cv_stage_rows(const float* __restrict__ src,  // &F[b][0][h][0]
                                              int C, int HW, int W, int x_first, int ncols, unsigned magic,
                                              float* lds, int tid) {
    // The kernel is VALU-bound if this loop does integer divisions per element (rocprofv3:
    // 5.8k VALU instructions per wave, 3/4 of them index math): e / ncols is one v_mul_hi with a
    // host-computed reciprocal (exact for e < 2^16), and (channel, column) are computed once per
    // element and reused for the LDS write.
    constexpr int BATCH = 8;
    const int RS = C + 4;
    const int total = C * ncols;
    for (int e0 = tid; e0 < total; e0 += NTHR * BATCH) {
        float v[BATCH];
        int dst[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const unsigned e = (unsigned)(e0 + k * NTHR);
            const int c = (int)__umulhi(e, magic);
            const int col = (int)e - c * ncols;
            const int x = x_first + col;
            const bool ok = (int)e < total;
            v[k] = (ok && x >= 0 && x < W) ? src[(size_t)c * HW + x] : 0.f;
            dst[k] = ok ? col * RS + cv_phys_channel<NJ>(c) : -1;
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
            if (dst[k] >= 0) lds[dst[k]] = v[k];
    }
}

template <int CPG>
__global__ __launch_bounds__(CVF_THREADS) void cost_volume_fwd_kernel(
    const float* __restrict__ Lg, const float* __restrict__ Rg, int Cg, int G,
    const float* __restrict__ Lc, const float* __restrict__ Rc, int Cc,
    const float* __restrict__ scale, float* __restrict__ vol,
    int H, int W, int D, int DC, int mask_left, unsigned magicL, unsigned magicR, int ablate) {
    STX_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    // XCD-aware order (workgroup b -> XCD b % 8, private L2s): all column tiles and disparity chunks
    // of one image row run on the same XCD, so its feature rows are fetched into one L2 only.
    int bid;
    {
        const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, k8 = blockIdx.x >> 3;
        bid = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k8;
    }
    const int nwt = (W + CV_WT - 1) / CV_WT, ndc = (D + DC - 1) / DC;
    const int w0 = (bid % nwt) * CV_WT;
    const int d0 = ((bid / nwt) % ndc) * DC;
    const int bh = bid / (nwt * ndc);
    const int b = bh / H, h = bh % H;
    const int HW = H * W;
    const int CT = G + 2 * Cc;
    const int Q = CT >> 2, GQ = G >> 2, CQ = Cc >> 2;
    const int RSg = Cg + 4, RSc = Cc + 4;
    const int NR = CV_WT + DC - 1;                 // right-feature columns kept in LDS
    const int x_first = w0 - d0 - (DC - 1);        // image column of LDS column 0

    float* Lg_s = reinterpret_cast<float*>(smem);
    float* Rg_s = Lg_s + (G ? CV_WT * RSg : 0);
    float* Lc_s = Rg_s + (G ? NR * RSg : 0);
    float* Rc_s = Lc_s + (Cc ? CV_WT * RSc : 0);

    if (G && ablate != 1) {
        cv_stage_rows<CPG, CVF_THREADS>(Lg + ((size_t)b * Cg * H + h) * W, Cg, HW, W, w0, CV_WT, magicL, Lg_s, tid);
        cv_stage_rows<CPG, CVF_THREADS>(Rg + ((size_t)b * Cg * H + h) * W, Cg, HW, W, x_first, NR, magicR, Rg_s, tid);
    }
    if (Cc) {
        cv_stage_rows<0, CVF_THREADS>(Lc + ((size_t)b * Cc * H + h) * W, Cc, HW, W, w0, CV_WT, magicL, Lc_s, tid);
        cv_stage_rows<0, CVF_THREADS>(Rc + ((size_t)b * Cc * H + h) * W, Cc, HW, W, x_first, NR, magicR, Rc_s, tid);
    }
    __syncthreads();

    const int dend_blk = (d0 + DC < D) ? DC : (D - d0);
    const float inv = 1.0f / (float)CPG;
    // The CV_WT*Q (column, quad) items of the tile are replicated over NG = threads/items groups that
    // split the chunk's disparities: 16 waves keep LDS reads, FMAs and stores of one tile in flight.
    const int nitems = CV_WT * Q;
    const int NG = CVF_THREADS / nitems > 0 ? CVF_THREADS / nitems : 1;
    const int dper = (dend_blk + NG - 1) / NG;
    for (int t = tid; t < nitems * NG; t += CVF_THREADS) {
        const int item = t % nitems, grp = t / nitems;
        const int wl = item / Q, q = item - wl * Q;
        const int w = w0 + wl;
        const int dbeg = grp * dper;                               // first disparity (within the chunk) of this group
        int dend = dbeg + dper < dend_blk ? dbeg + dper : dend_blk;
        if (w >= W || dbeg >= dend) continue;
        dend -= dbeg;                                             // disparities handled here: [0, dend) relative to dbeg
        float* out = vol + ((((size_t)b * D + d0 + dbeg) * H + h) * W + w) * CT + 4 * q;
        const size_t dstride = (size_t)H * W * CT;
        const float* sc = scale ? scale + (((size_t)b * D + d0 + dbeg) * H + h) * W + w : nullptr;
        const int dq0 = d0 + dbeg;                                // absolute disparity of relative index 0
        // disparities dd < nval are valid (w >= dq0+dd); the rest is zero-filled
        int nval = w - dq0 + 1;
        nval = nval < 0 ? 0 : (nval > dend ? dend : nval);
        if (ablate == 2) nval = 0;
        if (q < GQ) {
            // 4 groups x CPG channels of the left feature stay in registers.
            constexpr int P = (CPG >= 16) ? 1 : 16 / CPG;
            const int rot = q / P;                   // chunk rotation of this quad (cv_phys_channel)
            float4 l[CPG];
#pragma unroll
            for (int j = 0; j < CPG; ++j) l[j] = stx_ld4(Lg_s + wl * RSg + q * 4 * CPG + 4 * ((j + rot) % CPG));
            const float* r = Rg_s + (wl + DC - 1 - dbeg) * RSg + q * 4 * CPG;
            float* o_ = out;
            int dd = 0;
            for (; dd < nval; ++dd, r -= RSg, o_ += dstride) {
                float acc[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < CPG / 4; ++j) {
                        const float4 a = l[g * (CPG / 4) + j];
                        const float4 v = stx_ld4(r + ((g * (CPG / 4) + j + rot) % CPG) * 4);
                        s = fmaf(a.x, v.x, s);
                        s = fmaf(a.y, v.y, s);
                        s = fmaf(a.z, v.z, s);
                        s = fmaf(a.w, v.w, s);
                    }
                    acc[g] = s * inv;
                }
                float4 o = make_float4(acc[0], acc[1], acc[2], acc[3]);
                if (sc) { const float m = sc[(size_t)dd * HW]; o.x *= m; o.y *= m; o.z *= m; o.w *= m; }
                stx_st4(o_, o);
            }
            for (; dd < dend; ++dd, o_ += dstride) stx_st4(o_, make_float4(0.f, 0.f, 0.f, 0.f));
        } else if (q < GQ + CQ) {
            const float4 l = stx_ld4(Lc_s + wl * RSc + 4 * (q - GQ));
            for (int dd = 0; dd < dend; ++dd) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!mask_left || w >= dq0 + dd) {
                    o = l;
                    if (sc) { const float m = sc[(size_t)dd * HW]; o.x *= m; o.y *= m; o.z *= m; o.w *= m; }
                }
                stx_st4(out + dd * dstride, o);
            }
        } else {
            for (int dd = 0; dd < dend; ++dd) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (w >= dq0 + dd) {
                    o = stx_ld4(Rc_s + (wl + DC - 1 - dbeg - dd) * RSc + 4 * (q - GQ - CQ));
                    if (sc) { const float m = sc[(size_t)dd * HW]; o.x *= m; o.y *= m; o.z *= m; o.w *= m; }
                }
                stx_st4(out + dd * dstride, o);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward of the builders, scatter-free (no atomics).  blockIdx.y selects the side:
//   LEFT : gLg[c][t] = 1/cpg * sum_d gvol[d][t][g(c)]     * Rg[c][t-d]   (t >= d)
//          gLc[c][t] =         sum_d gvol[d][t][G+c]                     (t >= d or !mask_left)
//   RIGHT: gRg[c][t] = 1/cpg * sum_d gvol[d][t+d][g(c)]   * Lg[c][t+d]   (t+d < W)
//          gRc[c][t] =         sum_d gvol[d][t+d][G+Cc+c]                (t+d < W)
// A workgroup owns (b, h, 16 columns t) and walks channel chunks x disparity chunks.  Per
// (chunk, 16 disparities) it stages the needed slice of gvol ([dd][t][groups], 16-byte coalesced
// loads) and the D-shifted feature tile ([col][ch], transposed) into LDS -- out-of-range entries are
// staged as zeros so the inner loop is branch-free: 2 ds_read_b32 + 1 FMA per (item, disparity).
// A work item is (column, channel) with the channel fastest, so feature reads are conflict-free and
// the 8 lanes of a group broadcast-read the same gvol value.  Results leave through an LDS
// transpose as 64-byte NCHW row segments.
constexpr int CVB_CH = 160;    // gwc feature channels per pass (multiple of every supported cpg)
constexpr int CVB_DC = 16;     // disparities per staged slice
constexpr int CVB_ITEMS = CV_WT * CVB_CH / CV_THREADS;   // 10 accumulators per thread

__global__ __launch_bounds__(CV_THREADS) void cost_volume_bwd_kernel(
    const float* __restrict__ gvol, const float* __restrict__ Lg, const float* __restrict__ Rg,
    int Cg, int G, int Cc, float* __restrict__ gLg, float* __restrict__ gRg,
    float* __restrict__ gLc, float* __restrict__ gRc, int H, int W, int D, int mask_left, int pass) {
    STX_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * CV_WT;
    const bool right = blockIdx.y != 0;
    const int bh = blockIdx.z;
    const int b = bh / H, h = bh % H;
    const int HW = H * W;
    const int CT = G + 2 * Cc;
    const int cpg = G ? Cg / G : 1;
    const float inv = 1.0f / (float)cpg;
    constexpr int NCOL = CV_WT + CVB_DC - 1;       // 31 feature columns per slice
    constexpr int RS = CVB_CH + 4;
    float* gvs = reinterpret_cast<float*>(smem);           // [CVB_DC][CV_WT][GS] (GS = groups or channels)
    float* fs = gvs + CVB_DC * CV_WT * 40;                 // [NCOL][RS]
    float* ts = fs + NCOL * RS;                            // [CVB_CH][CV_WT + 1]
    const size_t dstride = (size_t)H * W * CT;
    const float* gv_row = gvol + (((size_t)b * D) * H + h) * W * CT;
    const float* feat = right ? Lg : Rg;
    float* gout = right ? gRg : gLg;

    // ---- gwc channels
    // `pass` <= CVB_CH channels per sweep over the disparities: a whole number of group quads (host: 160 for 4, 8 or
    // 20 channels per group, 144 for 12, 128 for 16, 112 for 28)
    for (int c0 = 0; c0 < Cg; c0 += pass) {
        const int nch = (Cg - c0 < pass) ? (Cg - c0) : pass;
        const int g0 = c0 / cpg, ng = nch / cpg;           // groups of this pass (ng <= 40)
        float acc[CVB_ITEMS];
#pragma unroll
        for (int k = 0; k < CVB_ITEMS; ++k) acc[k] = 0.f;
        for (int d0 = 0; d0 < D; d0 += CVB_DC) {
            __syncthreads();
            // gvol slice: gvs[dd][tl][j] = gvol[d0+dd][voxel column][g0+j], zero outside the image / D
            for (int idx = tid; idx < CVB_DC * CV_WT * (ng >> 2); idx += CV_THREADS) {
                const int f = idx % (ng >> 2), v = idx / (ng >> 2);
                const int tl = v % CV_WT, dd = v / CV_WT;
                const int d = d0 + dd, wv = right ? t0 + tl + d : t0 + tl;
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                if (d < D && wv < W) val = stx_ld4(gv_row + d * dstride + (size_t)wv * CT + g0 + 4 * f);
                stx_st4(gvs + (dd * CV_WT + tl) * ng + 4 * f, val);
            }
            // feature tile: column col <-> image column fc;  LEFT: fc = t0 + tl - d  -> col = tl - dd + 15
            //                                               RIGHT: fc = t0 + tl + d -> col = tl + dd
            {
                const int fc0 = right ? t0 + d0 : t0 - d0 - (CVB_DC - 1);
                const int lane = tid & 63, wave = tid >> 6;
                const int r = lane >> 5, xl = lane & 31;       // 2 channel rows x 32 columns per instruction
                const int fc = fc0 + xl;
                const bool ok = xl < NCOL && fc >= 0 && fc < W;
                const float* src = feat + (((size_t)b * Cg + c0) * H + h) * W;
                for (int cb = wave * 16; cb < nch; cb += 64) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int c = cb + 2 * k + r;
                        v[k] = (ok && c < nch) ? src[(size_t)c * HW + fc] : 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int c = cb + 2 * k + r;
                        if (xl < NCOL && c < nch) fs[xl * RS + c] = v[k];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < CVB_ITEMS; ++k) {
                const int item = tid + k * CV_THREADS;
                const int c = item % CVB_CH, tl = item / CVB_CH;
                if (c < nch) {
                    const int gl = c / cpg;
                    const float* gp = gvs + tl * ng + gl;
                    const float* fp = fs + c + (right ? tl : tl + CVB_DC - 1) * RS;
                    float a = acc[k];
#pragma unroll
                    for (int dd = 0; dd < CVB_DC; ++dd)
                        a = fmaf(gp[dd * CV_WT * ng], right ? fp[dd * RS] : fp[-dd * RS], a);
                    acc[k] = a;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CVB_ITEMS; ++k) {
            const int item = tid + k * CV_THREADS;
            const int c = item % CVB_CH, tl = item / CVB_CH;
            if (c < nch) ts[c * (CV_WT + 1) + tl] = acc[k] * inv;
        }
        __syncthreads();
        for (int idx = tid; idx < nch * CV_WT; idx += CV_THREADS) {
            const int tl = idx % CV_WT, c = idx / CV_WT;
            if (t0 + tl < W) gout[(((size_t)b * Cg + c0 + c) * H + h) * W + t0 + tl] = ts[c * (CV_WT + 1) + tl];
        }
    }

    // ---- concat channels: plain disparity sums of the matching gvol channels
    if (Cc) {
        float* cout = right ? gRc : gLc;
        const int coff = right ? G + Cc : G;
        for (int idx = tid; idx < Cc * CV_WT; idx += CV_THREADS) {
            const int c = idx % Cc, tl = idx / Cc;       // channel fastest: 4*Cc-byte contiguous gvol reads
            const int t = t0 + tl;
            if (t >= W) continue;
            float a0 = 0.f, a1 = 0.f;
            int d = 0;
            for (; d + 1 < D; d += 2) {
                const int w0_ = right ? t + d : t, w1_ = right ? t + d + 1 : t;
                const bool v0 = right ? (w0_ < W) : (!mask_left || t >= d);
                const bool v1 = right ? (w1_ < W) : (!mask_left || t >= d + 1);
                const float x0 = v0 ? gv_row[d * dstride + (size_t)w0_ * CT + coff + c] : 0.f;
                const float x1 = v1 ? gv_row[(d + 1) * dstride + (size_t)w1_ * CT + coff + c] : 0.f;
                a0 += x0;
                a1 += x1;
            }
            if (d < D) {
                const int w0_ = right ? t + d : t;
                const bool v0 = right ? (w0_ < W) : (!mask_left || t >= d);
                if (v0) a0 += gv_row[d * dstride + (size_t)w0_ * CT + coff + c];
            }
            cout[(((size_t)b * Cc + c) * H + h) * W + t] = a0 + a1;
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
    STX_REQUIRE(cpg == 4 || cpg == 8 || cpg == 12 || cpg == 16 || cpg == 20 || cpg == 28,
                "cost_volume_fwd: channels per group %d not in {4,8,12,16,20,28}", cpg);
    {   // second-generation builder (MFMA correlation, LDS-staged voxels): serves every configuration it accepts
        const int rc = stx_cv_fwd_mfma(Lg, Rg, Cg, G, Lc, Rc, Cc, scale, vol, B, H, W, D, mask_left, stream);
        if (rc >= 0) return rc;
    }
    STX_REQUIRE(cpg == 4 || cpg == 8 || cpg == 16, "cost_volume_fwd: %d channels per group are served by the MFMA builder only (D' <= 96, "
                "voxels of <= 64 channels; 20 / 28 per group: 4..16 groups)", cpg);
    // disparities per workgroup: split D evenly into chunks of <= 24 (two workgroups per CU for the
    // 320-channel gwc features: (16 + 16+24-1) columns x 1296 B = 71 KB of LDS each)
    const int nchunk = stx_cdiv(D, 24);
    int DC = stx_cdiv(D, nchunk);
    size_t lds = 0;
    for (;;) {
        const int NR = CV_WT + DC - 1;
        lds = 0;
        if (G) lds += (size_t)(CV_WT + NR) * (Cg + 4) * 4;
        if (Cc) lds += (size_t)(CV_WT + NR) * (Cc + 4) * 4;
        if (lds <= 160 * 1024 || DC == 1) break;
        DC = (DC + 1) / 2;
    }
    STX_REQUIRE(lds <= 160 * 1024 && DC <= CV_MAX_DC, "cost_volume_fwd: feature tile (%zu B) exceeds LDS", lds);
    dim3 grid(stx_cdiv(W, CV_WT) * stx_cdiv(D, DC) * B * H);
    hipStream_t st = (hipStream_t)stream;
    // reciprocals for e / ncols by multiply-high (exact while e < 2^16: the largest tile has Cg*NR elements)
    const int NRh = CV_WT + DC - 1;
    STX_REQUIRE((long long)(Cg > Cc ? Cg : Cc) * NRh < 65536, "cost_volume_fwd: feature tile too large");
    const unsigned magicL = (unsigned)(0x100000000ULL / CV_WT + 1), magicR = (unsigned)(0x100000000ULL / NRh + 1);
    const int ablate = 0;
#define CV_LAUNCH(CPG_)                                                                                       \
    {                                                                                                         \
        if (lds > 64 * 1024)                                                                                  \
            hipFuncSetAttribute((const void*)cost_volume_fwd_kernel<CPG_>,                                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
        hipLaunchKernelGGL(cost_volume_fwd_kernel<CPG_>, grid, dim3(CVF_THREADS), lds, st, Lg, Rg, Cg, G, Lc, \
                           Rc, Cc, scale, vol, H, W, D, DC, mask_left, magicL, magicR, ablate);               \
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
    if (G) {
        STX_REQUIRE(Lg && Rg && gLg && gRg && Cg % G == 0, "cost_volume_bwd: gwc operands missing");
        const int cpg = Cg / G;
        STX_REQUIRE(cpg == 4 || cpg == 8 || cpg == 12 || cpg == 16 || cpg == 20 || cpg == 28,
                    "cost_volume_bwd: channels per group %d not in {4,8,12,16,20,28}", cpg);
    }
    if (Cc) STX_REQUIRE(gLc && gRc, "cost_volume_bwd: concat outputs missing");
    if (G) {      // second generation: matrix-core kernel with loader waves (cost_volume_bwd_mfma.hip); -1 = not served
        const int rc = stx_cv_bwd_mfma(gvol, Lg, Rg, Cg, G, Cc, gLg, gRg, gLc, gRc, B, H, W, D, mask_left, stream);
        if (rc >= 0) return rc;
    }
    const size_t lds = ((size_t)CVB_DC * CV_WT * 40 + (size_t)(CV_WT + CVB_DC - 1) * (CVB_CH + 4) +
                        (size_t)CVB_CH * (CV_WT + 1)) * 4;
    dim3 grid(stx_cdiv(W, CV_WT), 2, B * H);
    const int cpg_ = G ? Cg / G : 8;
    const int pass = (CVB_CH / (4 * cpg_)) * (4 * cpg_);       // whole group quads per sweep
    hipLaunchKernelGGL(cost_volume_bwd_kernel, grid, dim3(CV_THREADS), lds, (hipStream_t)stream, gvol, Lg, Rg,
                       G ? Cg : 0, G, Cc, gLg, gRg, gLc, gRc, H, W, D, mask_left, pass);
    return stx_check_launch("cost_volume_bwd");
}
