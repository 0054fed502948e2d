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
// Row-persistent, software-pipelined builder (used when gwc features are present).
//
// The tile kernel above re-stages a (16 + 39)-column x 320-channel feature tile (71 KB) per
// workgroup and only then computes: rocprofv3 shows it latency-bound at 28 % of the HBM roofline.
// Here a workgroup (8 waves) owns (b, h, 16 disparities) and walks the image row left to right in
// 16-column tiles.  LDS holds the current 16 left columns and a 32-column *ring* of right columns
// (column x lives in slot x & 31): moving one tile to the right needs exactly 16 new left and 16
// new right columns, so every feature element is staged once per (row, disparity chunk).  Those 32
// columns for tile t+1 are loaded into registers (about 21 dwords per lane, all in flight) while
// tile t is being multiplied and stored; they are written into LDS between two barriers.
constexpr int CVR_THREADS = 512;
constexpr int CVR_DC = 16;        // disparities per workgroup
constexpr int CVR_RING = 32;      // right-feature ring (>= CV_WT + CVR_DC - 1 columns)
constexpr int CVR_MAXK = 10;      // max channel steps of 32 per operand (Cg <= 320)

template <int CPG>
__global__ __launch_bounds__(CVR_THREADS, 4) void cost_volume_fwd_row_kernel(
    const float* __restrict__ Lg, const float* __restrict__ Rg, int Cg, int G,
    const float* __restrict__ Lc, const float* __restrict__ Rc, int Cc,
    const float* __restrict__ scale, float* __restrict__ vol, int H, int W, int D, int mask_left) {
    STX_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    int bid;
    {
        const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, k8 = blockIdx.x >> 3;
        bid = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k8;
    }
    const int ndc = (D + CVR_DC - 1) / CVR_DC;
    const int d0 = (bid % ndc) * CVR_DC;
    const int bh = bid / ndc;
    const int b = bh / H, h = bh % H;
    const int HW = H * W;
    const int CT = G + 2 * Cc;
    const int Q = CT >> 2, GQ = G >> 2, CQ = Cc >> 2;
    const int RSg = Cg + 4, RSc = Cc + 4;
    float* Lg_s = reinterpret_cast<float*>(smem);          // [16][RSg]
    float* Rg_s = Lg_s + CV_WT * RSg;                      // [32][RSg] ring
    float* Lc_s = Rg_s + CVR_RING * RSg;                   // [16][RSc]
    float* Rc_s = Lc_s + CV_WT * RSc;                      // [32][RSc] ring
    const float* Lg_row = Lg + ((size_t)b * Cg * H + h) * W;
    const float* Rg_row = Rg + ((size_t)b * Cg * H + h) * W;
    const float* Lc_row = Cc ? Lc + ((size_t)b * Cc * H + h) * W : nullptr;
    const float* Rc_row = Cc ? Rc + ((size_t)b * Cc * H + h) * W : nullptr;
    const int dend_blk = (d0 + CVR_DC < D) ? CVR_DC : (D - d0);
    const float inv = 1.0f / (float)CPG;
    const int ntile = (W + CV_WT - 1) / CV_WT;
    // Staging map: lane -> column (tid & 15) and channel (tid >> 4) + 32*k: one base pointer per
    // operand and a constant channel stride, so a tile costs ~2*Cg/32 + 2 independent loads per lane
    // and almost no index arithmetic or live state (a generic flattened map blew up to 256 VGPRs).
    constexpr int CPS = CVR_THREADS / CV_WT;                  // channels covered per k step (32)
    const int scol = tid & (CV_WT - 1), sch = tid >> 4;
    const int nkg = (Cg + CPS - 1) / CPS;                     // <= CVR_MAXK (checked on the host)
    float stL[CVR_MAXK], stR[CVR_MAXK], stLc = 0.f, stRc = 0.f;
    auto prefetch = [&](int t) {                              // loads for tile t -> registers
        const int xl = t * CV_WT + scol, xr = xl - d0;
        const bool okl = xl < W, okr = xr >= 0 && xr < W;
        const float* pl = Lg_row + (size_t)sch * HW + xl;
        const float* pr = Rg_row + (size_t)sch * HW + xr;
#pragma unroll
        for (int k = 0; k < CVR_MAXK; ++k) {
            const bool ck = k < nkg && sch + CPS * k < Cg;
            stL[k] = (ck && okl) ? pl[(size_t)k * CPS * HW] : 0.f;
            stR[k] = (ck && okr) ? pr[(size_t)k * CPS * HW] : 0.f;
        }
        if (sch < Cc) {
            stLc = okl ? Lc_row[(size_t)sch * HW + xl] : 0.f;
            stRc = okr ? Rc_row[(size_t)sch * HW + xr] : 0.f;
        }
    };
    auto commit = [&](int t) {                                // registers -> LDS (left tile, right ring)
        const int rslot = (t * CV_WT + scol - d0 + 1024) & (CVR_RING - 1);
#pragma unroll
        for (int k = 0; k < CVR_MAXK; ++k) {
            const int c = sch + CPS * k;
            if (k < nkg && c < Cg) {
                const int pc = cv_phys_channel<CPG>(c);
                Lg_s[scol * RSg + pc] = stL[k];
                Rg_s[rslot * RSg + pc] = stR[k];
            }
        }
        if (sch < Cc) {
            Lc_s[scol * RSc + sch] = stLc;
            Rc_s[rslot * RSc + sch] = stRc;
        }
    };

    // work split of a tile: (column, quad) items replicated over NG disparity groups
    const int nitems = CV_WT * Q;
    const int NG = CVR_THREADS / nitems > 0 ? CVR_THREADS / nitems : 1;
    const int dper = (dend_blk + NG - 1) / NG;
    const size_t dstride = (size_t)H * W * CT;

    prefetch(0);
    for (int t = 0; t < ntile; ++t) {
        __syncthreads();                 // everyone is done reading tile t-1 from LDS
        commit(t);
        __syncthreads();
        if (t + 1 < ntile) prefetch(t + 1);
        const int w0 = t * CV_WT;
        for (int u = tid; u < nitems * NG; u += CVR_THREADS) {
            const int item = u % nitems, grp = u / nitems;
            const int wl = item / Q, q = item - wl * Q;
            const int w = w0 + wl;
            const int dbeg = grp * dper;
            int dend = dbeg + dper < dend_blk ? dbeg + dper : dend_blk;
            if (w >= W || dbeg >= dend) continue;
            dend -= dbeg;
            const int dq0 = d0 + dbeg;
            float* o_ = vol + ((((size_t)b * D + dq0) * H + h) * W + w) * CT + 4 * q;
            const float* sc = scale ? scale + (((size_t)b * D + dq0) * H + h) * W + w : nullptr;
            int nval = w - dq0 + 1;
            nval = nval < 0 ? 0 : (nval > dend ? dend : nval);
            if (q < GQ) {
                constexpr int P = (CPG >= 16) ? 1 : 16 / CPG;
                const int rot = q / P;
                float4 l[CPG];
#pragma unroll
                for (int j = 0; j < CPG; ++j) l[j] = stx_ld4(Lg_s + wl * RSg + q * 4 * CPG + 4 * ((j + rot) % CPG));
                int dd = 0;
                for (; dd < nval; ++dd, o_ += dstride) {
                    const float* r = Rg_s + ((w - dq0 - dd) & (CVR_RING - 1)) * RSg + q * 4 * CPG;
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
                for (int dd = 0; dd < dend; ++dd, o_ += dstride) {
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (!mask_left || dd < nval) {
                        o = l;
                        if (sc) { const float m = sc[(size_t)dd * HW]; o.x *= m; o.y *= m; o.z *= m; o.w *= m; }
                    }
                    stx_st4(o_, o);
                }
            } else {
                for (int dd = 0; dd < dend; ++dd, o_ += dstride) {
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (dd < nval) {
                        o = stx_ld4(Rc_s + ((w - dq0 - dd) & (CVR_RING - 1)) * RSc + 4 * (q - GQ - CQ));
                        if (sc) { const float m = sc[(size_t)dd * HW]; o.x *= m; o.y *= m; o.z *= m; o.w *= m; }
                    }
                    stx_st4(o_, o);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Lean row-persistent builder for 8 channels per group and voxels of <= 64 channels (GwcNet_GC: 40 groups +
// 2 x 12 concat channels = exactly 64; GwcNet_G / ACVNet gwc volume: 40; PSMNet: no groups, 2 x 32 concat channels).
//
// cost_volume_fwd_row_kernel above keeps every size a run-time value and maps a thread to (column, 4-channel quad):
// its compute loop issues ~150 instructions per stored float4 (run-time divisions and multiplies for the item
// map and the ring addresses, three divergent quad flavours per wave) and is VALU-issue bound at 13 us per
// 16-column tile.  Here one LANE is one channel of the voxel and a wave walks the disparities of one column:
//   lanes [0, G)          group g: L[w][8g..8g+8) stays in 8 registers, per disparity 2 ds_read_b128 of the right
//                         column + 8 FMAs;
//   lanes [G, G+Cc)       left concat channel (a register, masked);   lanes [G+Cc, G+2Cc): right concat (ds_read_b32);
// every disparity is one 256-byte dword store per wave and ~25 instructions.  The waves are specialised: waves 0-3
// stage (the 16 new left and 16 new right columns of tile t+1 go through registers into LDS between the two
// barriers of a tile), waves 4-15 multiply and store and never wait on a memory counter.
// Measured at 576x960 (ablations in one session): staging alone 0.045 ms, multiply + store alone 0.099 ms, both
// 0.172 ms (row kernel above: 0.198 ms); with 4 instead of 12 compute waves 0.21 ms; the stores cost nothing extra
// (multiply without stores 0.114 ms): the per-voxel chain ds_read -> 8 dependent FMAs is latency-bound.  LDS image of a column:
// [half][group][4] (+4 pad) so that the 16 lanes of a ds_read_b128 phase hit 16 distinct 16-byte bank groups.
constexpr int CVL_THREADS = 1024, CVL_LOADERS = 256;      // 4 staging waves + 12 compute waves
constexpr int CVL_DC = 16, CVL_RING = 32, CVL_FS = 324, CVL_CS = 33, CVL_MAXCC = 32;   // concat rows: odd stride
constexpr int CVL_MAXK = 20;                                   // channel steps of 16 per operand (Cg <= 320)

__global__ __launch_bounds__(CVL_THREADS, 8) void cost_volume_fwd_g8_kernel(
    const float* __restrict__ Lg, const float* __restrict__ Rg, int G, const float* __restrict__ Lc,
    const float* __restrict__ Rc, int Cc, float* __restrict__ vol, int H, int W, int D, int mask_left) {
    STX_DYN_SMEM(smem);
    const int fs = G ? CVL_FS : 0;                            // concat-only volumes (PSMNet) keep no gwc image
    float* Lg_s = reinterpret_cast<float*>(smem);             // [16][fs]
    float* Rg_s = Lg_s + CV_WT * fs;                          // [32][fs] ring
    float* Lc_s = Rg_s + CVL_RING * fs;                       // [16][CVL_CS]
    float* Rc_s = Lc_s + CV_WT * CVL_CS;                      // [32][CVL_CS] ring
    const int tid = threadIdx.x;
    int bid;
    {
        const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, k8 = blockIdx.x >> 3;
        bid = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k8;
    }
    const int ndc = (D + CVL_DC - 1) / CVL_DC;
    const int d0 = (bid % ndc) * CVL_DC;
    const int bh = bid / ndc;
    const int b = bh / H, h = bh - b * H;
    const int HW = H * W, Cg = 8 * G, CT = G + 2 * Cc;
    const int dend = (d0 + CVL_DC < D) ? CVL_DC : (D - d0);
    const int ntile = (W + CV_WT - 1) / CV_WT;
    const bool loader = __builtin_amdgcn_readfirstlane(tid >> 6) < CVL_LOADERS / 64;     // wave-uniform role

    if (loader) {
        const float* Lg_row = stx_uniform_ptr(Lg + ((size_t)b * Cg * H + h) * W);
        const float* Rg_row = stx_uniform_ptr(Rg + ((size_t)b * Cg * H + h) * W);
        const float* Lc_row = stx_uniform_ptr(Lc + ((size_t)b * Cc * H + h) * W);     // (only dereferenced when Cc > 0)
        const float* Rc_row = stx_uniform_ptr(Rc + ((size_t)b * Cc * H + h) * W);
        // staging map: lane -> column tid & 15, channel (tid >> 4) + 16 k
        const int scol = tid & (CV_WT - 1), sch = tid >> 4;
        float stL[CVL_MAXK], stR[CVL_MAXK], stLc[2] = {0.f, 0.f}, stRc[2] = {0.f, 0.f};
        auto prefetch = [&](int t) {
            const int xl = t * CV_WT + scol, xr = xl - d0;
            const bool okl = xl < W, okr = xr >= 0 && xr < W;
            unsigned ol = (unsigned)(sch * HW + xl), orr = (unsigned)(sch * HW + xr);
            STX_OPAQUE_VGPR(ol);         // (otherwise the 40 per-step offsets are hoisted out of the tile loop and spilled)
            STX_OPAQUE_VGPR(orr);
            const unsigned step = (unsigned)(16 * HW);
#pragma unroll
            for (int k = 0; k < CVL_MAXK; ++k) {
                const bool ck = sch + 16 * k < Cg;
                stL[k] = (ck && okl) ? Lg_row[ol] : 0.f;
                stR[k] = (ck && okr) ? Rg_row[orr] : 0.f;
                ol += step; orr += step;
            }
            unsigned cl_ = (unsigned)(sch * HW + xl), cr_ = (unsigned)(sch * HW + xr);
            STX_OPAQUE_VGPR(cl_);
            STX_OPAQUE_VGPR(cr_);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (sch + 16 * j < Cc) {
                    stLc[j] = okl ? Lc_row[cl_] : 0.f;
                    stRc[j] = okr ? Rc_row[cr_] : 0.f;
                }
                cl_ += step; cr_ += step;
            }
        };
        auto commit = [&](int t) {
            const int rslot = (t * CV_WT + scol - d0 + 1024) & (CVL_RING - 1);
            int r = sch;
            STX_OPAQUE_VGPR(r);
#pragma unroll
            for (int k = 0; k < CVL_MAXK; ++k) {
                const int c = r + 16 * k;
                if (c < Cg) {
                    const int pos = ((c >> 2) & 1) * 160 + (c >> 3) * 4 + (c & 3);
                    Lg_s[scol * CVL_FS + pos] = stL[k];
                    Rg_s[rslot * CVL_FS + pos] = stR[k];
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = sch + 16 * j;
                if (c < Cc) {
                    Lc_s[scol * CVL_CS + c] = stLc[j];
                    Rc_s[rslot * CVL_CS + c] = stRc[j];
                }
            }
        };
        prefetch(0);
        for (int t = 0; t < ntile; ++t) {
            __syncthreads();             // the compute waves are done reading tile t-1
            commit(t);
            __syncthreads();
            if (t + 1 < ntile) prefetch(t + 1);
        }
        return;
    }

    // ---- compute waves: lane = channel of the voxel; a work item is (column, group of 4 disparities), the 64 items of
    // a tile are dealt round-robin to the 12 compute waves (2 workgroups per CU = 24 compute waves: the per-voxel chain
    // ds_read -> 8 dependent FMAs -> store is latency-bound, measured 0.15 ms with 8 compute waves per CU)
    constexpr int NCW = (CVL_THREADS - CVL_LOADERS) / 64;
    const int lane = tid & 63, cw = (tid >> 6) - CVL_LOADERS / 64;
    const bool is_g = lane < G, is_l = !is_g && lane < G + Cc, is_r = lane >= G + Cc && lane < CT;
    const int gq = is_g ? lane * 4 : 0;
    const int cl = is_l ? lane - G : 0, cr = is_r ? lane - G - Cc : 0;
    const size_t dstride = (size_t)HW * CT;
    for (int t = 0; t < ntile; ++t) {
        __syncthreads();
        __syncthreads();                 // tile t is in LDS
#pragma unroll 1
        for (int item = cw; item < CV_WT * (CVL_DC / 4); item += NCW) {
            const int wl = item & (CV_WT - 1), dg = item >> 4;
            const int w = t * CV_WT + wl;
            if (w >= W || 4 * dg >= dend) continue;
            float4 l0 = make_float4(0.f, 0.f, 0.f, 0.f), l1 = l0;
            if (G) { l0 = stx_ld4(Lg_s + wl * CVL_FS + gq); l1 = stx_ld4(Lg_s + wl * CVL_FS + 160 + gq); }
            const float lcv = Lc_s[wl * CVL_CS + cl];
            float* o = vol + ((((size_t)b * D + d0 + 4 * dg) * H + h) * W + w) * CT + lane;
            const int x0 = w - d0 - 4 * dg;                      // right column of the item's first disparity
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const int slot = (x0 - dd + 1024) & (CVL_RING - 1);
                const float rc = Rc_s[slot * CVL_CS + cr];
                float s = 0.f;
                if (G) {                                         // wave-uniform
                    const float* rp = Rg_s + slot * CVL_FS + gq;
                    const float4 r0 = stx_ld4(rp), r1 = stx_ld4(rp + 160);
                    s = l0.x * r0.x;
                    s = fmaf(l0.y, r0.y, s); s = fmaf(l0.z, r0.z, s); s = fmaf(l0.w, r0.w, s);
                    s = fmaf(l1.x, r1.x, s); s = fmaf(l1.y, r1.y, s); s = fmaf(l1.z, r1.z, s); s = fmaf(l1.w, r1.w, s);
                }
                const bool valid = x0 - dd >= 0;
                float v = valid ? s * 0.125f : 0.f;              // (ring slots left of the image hold stale data)
                v = is_l ? ((valid || !mask_left) ? lcv : 0.f) : v;
                v = is_r ? (valid ? rc : 0.f) : v;
                if (lane < CT && 4 * dg + dd < dend) o[(size_t)dd * dstride] = v;
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
    // `pass` <= CVB_CH channels per sweep over the disparities: a whole number of group quads (host: 160 for 4 or 8
    // channels per group, 144 for 12, 128 for 16)
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


// ---------------------------------------------------------------------------------------------
// Backward for 8 channels per group (GwcNet / ACVNet: 320 channels in 40 groups), software-pipelined.
//
// The generic kernel above alternates "stage a slice" and "multiply it" (measured 1.27 ms at 576x960:
// 0.5 ms staging + 0.5 ms of 2-LDS-reads-per-FMA compute + 0.2 ms of a serial concat loop, nothing
// overlapped).  Here a workgroup (320 threads = 40 groups x 8 column pairs) owns (b, h, side, 16 output
// columns) and ALL channels, and walks the disparities in slices of 8:
//   * gvol slice: whole 64-channel voxels [8 dd][16 columns] (256-byte coalesced float4 loads; RIGHT side: the
//     sheared set w = t + d), staged once per slice and shared by the gwc and the concat sums;
//   * feature window: a 32-slot ring of image columns in LDS ([column][channel], transposed from NCHW on the
//     way in); moving to the next slice needs only 8 new columns;
//   * the loads of slice s+1 (8 float4 + 8 dwords per lane) are in flight while slice s is multiplied;
//   * a thread keeps 8 channels x 2 columns of accumulators: per slice 16 gvol scalars and 9 feature columns
//     (2 ds_read_b128 each) feed 128 FMAs (0.27 LDS reads per FMA instead of 2).
// LDS layouts are chosen so that every read is conflict-free: gvol voxel stride 68 dwords (lanes = 8 column
// pairs x 8 groups -> 64 distinct banks), feature column stride 324 dwords with the two 4-channel halves of
// all groups stored as two planes ([half][group][4]: a 16-lane ds_read_b128 phase covers 16 distinct 16-byte
// bank groups).
constexpr int CVG_THREADS = 320;
constexpr int CVG_DC = 8;            // disparities per slice
constexpr int CVG_GS = 68;           // dwords per staged gvol voxel
constexpr int CVG_RING = 32;         // feature-column ring (23-column window + 8 new columns)
constexpr int CVG_FS = 324;          // dwords per ring column

__global__ __launch_bounds__(CVG_THREADS, 3) void cost_volume_bwd_g8_kernel(
    const float* __restrict__ gvol, const float* __restrict__ Lg, const float* __restrict__ Rg, int G, int Cc,
    float* __restrict__ gLg, float* __restrict__ gRg, float* __restrict__ gLc, float* __restrict__ gRc, int H, int W,
    int D, int mask_left) {
    STX_DYN_SMEM(smem);
    float* gvs = reinterpret_cast<float*>(smem);                 // [8 dd][16 tl][CVG_GS]
    float* fs = gvs + CVG_DC * CV_WT * CVG_GS;                   // [CVG_RING][CVG_FS]
    const int tid = threadIdx.x;
    const int tp = tid & 7, g = tid >> 3;                        // column pair, group
    int bid;
    {   // consecutive work items on the same XCD (the 2 x ntile workgroups of a row share its gvol slab in L2)
        const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, k8 = blockIdx.x >> 3;
        bid = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k8;
    }
    const int ntile = (W + CV_WT - 1) / CV_WT;
    const bool right = bid & 1;
    const int tile = (bid >> 1) % ntile, bh = (bid >> 1) / ntile;
    const int b = bh / H, h = bh - b * H;
    const int t0 = tile * CV_WT;
    const int HW = H * W, Cg = 8 * G, CT = G + 2 * Cc;
    const size_t dstride = (size_t)HW * CT;
    const float* gv_row = gvol + (((size_t)b * D) * H + h) * W * CT;
    const float* feat = (right ? Lg : Rg) + ((size_t)b * Cg * H + h) * W;

    // ---- staging: global -> registers -> LDS
    // gvol slice: threads 0..255 = 16 columns x 16 float4 of a voxel; one disparity row per step, so the
    // 8 loads of a lane differ by a constant stride (one base pointer, no per-load index arithmetic)
    float4 sg[CVG_DC];
    float sf[8];
    const int gtl = (tid >> 4) & 15, gf4 = tid & 15;
    const bool gthread = tid < 256;
    const size_t gstep = right ? dstride + CT : dstride;          // RIGHT: w = t + d moves one voxel per disparity
    auto load_gv = [&](int d0) {
        const int w0 = right ? t0 + gtl + d0 : t0 + gtl;
        const float* p = gv_row + (size_t)d0 * dstride + (size_t)w0 * CT + 4 * gf4;
#pragma unroll
        for (int k = 0; k < CVG_DC; ++k) {
            const int w = right ? w0 + k : w0;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gthread && d0 + k < D && w < W && 4 * gf4 < CT) val = stx_ld4(p + (size_t)k * gstep);
            sg[k] = val;
        }
    };
    auto store_gv = [&]() {
        if (gthread) {
#pragma unroll
            for (int k = 0; k < CVG_DC; ++k) stx_st4(gvs + (k * CV_WT + gtl) * CVG_GS + 4 * gf4, sg[k]);
        }
    };
    // 8 image columns fc0 .. fc0+7 x all channels: lane -> column tid & 7, channel (tid >> 3) + 40 i
    // (STX_OPAQUE_VGPR: recompute the per-load offsets from the lane id each time instead of keeping ~40
    //  loop-invariant addresses alive across the disparity loop -- they cost the second resident workgroup)
    auto load_f = [&](int fc0) {
        const int fc = fc0 + (tid & 7);
        const bool ok = fc >= 0 && fc < W;
        int r = tid >> 3;
        STX_OPAQUE_VGPR(r);
        const float* p = feat + (size_t)r * HW + fc;
#pragma unroll
        for (int i = 0; i < 8; ++i) sf[i] = (ok && r + 40 * i < Cg) ? p[(size_t)(40 * i) * HW] : 0.f;
    };
    auto store_f = [&](int fc0) {
        const int slot = (fc0 + (tid & 7) + 4096) & (CVG_RING - 1);
        int r = tid >> 3;
        STX_OPAQUE_VGPR(r);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = r + 40 * i;
            fs[slot * CVG_FS + ((c >> 2) & 1) * 160 + (c >> 3) * 4 + (c & 3)] = sf[i];
        }
    };
    // feature window of slice d0: LEFT columns [t0 - d0 - 7, +23), RIGHT [t0 + d0, +23)
    const int lo0 = right ? t0 : t0 - (CVG_DC - 1);
#pragma unroll 1
    for (int k = 0; k < 3; ++k) {
        load_f(lo0 + 8 * k);
        store_f(lo0 + 8 * k);
    }
    load_gv(0);
    store_gv();
    __syncthreads();

    float acc[2][8];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[e][j] = 0.f;
    float cacc = 0.f;                                             // concat: thread -> (channel tid % Cc, column tid / Cc)
    const bool cthread = tid < Cc * CV_WT;
    const int cc = cthread ? tid % Cc : 0, ctl = cthread ? tid / Cc : 0;

    for (int d0 = 0; d0 < D; d0 += CVG_DC) {
        const bool more = d0 + CVG_DC < D;
        const int fnew = right ? t0 + d0 + 23 : t0 - d0 - 15;     // the 8 new columns of the next slice
        if (more) {
            load_gv(d0 + CVG_DC);
            load_f(fnew);
        }
        if (g < G) {
            float gvr[2][CVG_DC];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int dd = 0; dd < CVG_DC; ++dd) gvr[e][dd] = gvs[(dd * CV_WT + 2 * tp + e) * CVG_GS + g];
            const int fc0 = right ? t0 + 2 * tp + d0 : t0 + 2 * tp - d0 - (CVG_DC - 1);
#pragma unroll
            for (int k = 0; k <= CVG_DC; ++k) {
                const float* fp = fs + ((fc0 + k + 4096) & (CVG_RING - 1)) * CVG_FS + g * 4;
                const float4 f0 = stx_ld4(fp), f1 = stx_ld4(fp + 160);
                // column k meets (e = 0, dd = 7 - k) and (e = 1, dd = 8 - k) on the LEFT, (0, k) and (1, k - 1) on the RIGHT
                if (k < CVG_DC) {
                    const float m = right ? gvr[0][k] : gvr[0][CVG_DC - 1 - k];
                    acc[0][0] = fmaf(m, f0.x, acc[0][0]); acc[0][1] = fmaf(m, f0.y, acc[0][1]);
                    acc[0][2] = fmaf(m, f0.z, acc[0][2]); acc[0][3] = fmaf(m, f0.w, acc[0][3]);
                    acc[0][4] = fmaf(m, f1.x, acc[0][4]); acc[0][5] = fmaf(m, f1.y, acc[0][5]);
                    acc[0][6] = fmaf(m, f1.z, acc[0][6]); acc[0][7] = fmaf(m, f1.w, acc[0][7]);
                }
                if (k % 3 == 2) STX_SCHED_BARRIER();      // keep at most 3 columns (24 VGPRs) of reads in flight
                if (k >= 1) {
                    const float m = right ? gvr[1][k - 1] : gvr[1][CVG_DC - k];
                    acc[1][0] = fmaf(m, f0.x, acc[1][0]); acc[1][1] = fmaf(m, f0.y, acc[1][1]);
                    acc[1][2] = fmaf(m, f0.z, acc[1][2]); acc[1][3] = fmaf(m, f0.w, acc[1][3]);
                    acc[1][4] = fmaf(m, f1.x, acc[1][4]); acc[1][5] = fmaf(m, f1.y, acc[1][5]);
                    acc[1][6] = fmaf(m, f1.z, acc[1][6]); acc[1][7] = fmaf(m, f1.w, acc[1][7]);
                }
            }
        }
        if (cthread) {
            const int coff = right ? G + Cc : G;
#pragma unroll
            for (int dd = 0; dd < CVG_DC; ++dd) {
                const float x = gvs[(dd * CV_WT + ctl) * CVG_GS + coff + cc];
                if (right || !mask_left || t0 + ctl >= d0 + dd) cacc += x;
            }
        }
        __syncthreads();                 // slice d0 has been consumed
        if (more) {
            store_gv();
            store_f(fnew);
        }
        __syncthreads();
    }

    // ---- results: transpose through LDS so that a wave writes 64-byte row segments of the NCHW gradients
    float* ts = fs;                                               // [Cg][17]
    if (g < G) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 8; ++j) ts[(g * 8 + j) * (CV_WT + 1) + 2 * tp + e] = acc[e][j] * 0.125f;
    }
    __syncthreads();
    float* gout = (right ? gRg : gLg) + ((size_t)b * Cg * H + h) * W;
    for (int idx = tid; idx < Cg * CV_WT; idx += CVG_THREADS) {
        const int tl = idx & (CV_WT - 1), c = idx >> 4;
        if (t0 + tl < W) gout[(size_t)c * HW + t0 + tl] = ts[c * (CV_WT + 1) + tl];
    }
    if (cthread && t0 + ctl < W) {
        float* cout = (right ? gRc : gLc) + ((size_t)b * Cc * H + h) * W;
        cout[(size_t)cc * HW + t0 + ctl] = cacc;
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
    STX_REQUIRE(cpg == 4 || cpg == 8 || cpg == 12 || cpg == 16, "cost_volume_fwd: channels per group %d not in {4,8,12,16}", cpg);
    {   // second-generation builder (MFMA correlation, LDS-staged voxels): serves every configuration it accepts
        const int rc = stx_cv_fwd_mfma(Lg, Rg, Cg, G, Lc, Rc, Cc, scale, vol, B, H, W, D, mask_left, stream);
        if (rc >= 0) return rc;
    }
    STX_REQUIRE(cpg != 12, "cost_volume_fwd: 12 channels per group need the MFMA builder (STX_CV_OLD is set?)");
    hipStream_t st0 = (hipStream_t)stream;
    static const int no_row = getenv("STX_CV_NO_ROW") ? 1 : 0;
    const size_t lds_row = ((size_t)(CV_WT + CVR_RING) * (Cg + 4) + (size_t)(CV_WT + CVR_RING) * (Cc + 4)) * 4;
    static const int no_g8 = getenv("STX_CV_NO_G8") ? 1 : 0;
    if (!no_g8 && !scale && (G == 0 || cpg == 8) && G <= 40 && Cc <= CVL_MAXCC && G + 2 * Cc <= 64) {
        const size_t lds8 = ((size_t)(CV_WT + CVL_RING) * (G ? CVL_FS : 0) + (size_t)(CV_WT + CVL_RING) * CVL_CS) * 4;
        hipFuncSetAttribute((const void*)cost_volume_fwd_g8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
        hipLaunchKernelGGL(cost_volume_fwd_g8_kernel, dim3(B * H * stx_cdiv(D, CVL_DC)), dim3(CVL_THREADS), lds8, st0, Lg,
                           Rg, G, Lc, Rc, Cc, vol, H, W, D, mask_left);
        return stx_check_launch("cost_volume_fwd(g8)");
    }
    if (G && !no_row && Cg <= CVR_MAXK * (CVR_THREADS / CV_WT) && Cc <= CVR_THREADS / CV_WT && lds_row <= 160 * 1024) {
        dim3 grid(B * H * stx_cdiv(D, CVR_DC));
#define CVR_LAUNCH(CPG_)                                                                                          \
    {                                                                                                             \
        hipFuncSetAttribute((const void*)cost_volume_fwd_row_kernel<CPG_>,                                        \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_row);                            \
        hipLaunchKernelGGL(cost_volume_fwd_row_kernel<CPG_>, grid, dim3(CVR_THREADS), lds_row, st0, Lg, Rg, Cg,   \
                           G, Lc, Rc, Cc, scale, vol, H, W, D, mask_left);                                        \
    }
        if (cpg == 4) CVR_LAUNCH(4) else if (cpg == 8) CVR_LAUNCH(8) else CVR_LAUNCH(16)
#undef CVR_LAUNCH
        return stx_check_launch("cost_volume_fwd(row)");
    }
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
    static const int ablate = getenv("STX_CV_ABLATE") ? atoi(getenv("STX_CV_ABLATE")) : 0;   // profiling only
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
        STX_REQUIRE(cpg == 4 || cpg == 8 || cpg == 12 || cpg == 16, "cost_volume_bwd: channels per group %d not in {4,8,12,16}", cpg);
    }
    if (Cc) STX_REQUIRE(gLc && gRc, "cost_volume_bwd: concat outputs missing");
    if (G) {      // second generation: matrix-core kernel with loader waves (cost_volume_bwd_mfma.hip); -1 = not served
        const int rc = stx_cv_bwd_mfma(gvol, Lg, Rg, Cg, G, Cc, gLg, gRg, gLc, gRc, B, H, W, D, mask_left, stream);
        if (rc >= 0) return rc;
    }
    static const int no_g8 = getenv("STX_CVB_GENERIC") ? 1 : 0;
    if (G && Cg == 8 * G && G <= 40 && G + 2 * Cc <= 64 && Cc * CV_WT <= CVG_THREADS && !no_g8) {
        const size_t lds8 = ((size_t)CVG_DC * CV_WT * CVG_GS + (size_t)CVG_RING * CVG_FS) * 4;
        hipFuncSetAttribute((const void*)cost_volume_bwd_g8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
        hipLaunchKernelGGL(cost_volume_bwd_g8_kernel, dim3(2 * stx_cdiv(W, CV_WT) * B * H), dim3(CVG_THREADS), lds8,
                           (hipStream_t)stream, gvol, Lg, Rg, G, Cc, gLg, gRg, gLc, gRc, H, W, D, mask_left);
        return stx_check_launch("cost_volume_bwd(g8)");
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
