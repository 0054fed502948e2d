// Cost-volume backward, second generation: the group-wise correlation gradients on the matrix cores, the volume
// gradient streamed through LDS exactly once per side by loader waves that do nothing else.
//
// Backward of the same reference functions as cost_volume.hip (build_gwc_volume GwcNet/submodule.py:53-63,
// build_concat_volume GwcNet/submodule.py:30-41 / ACVNet/submodule.py:180-191, torch.cat gwcnet.py:180 -- autograd of
// those in the reference) for 4, 8, 12 or 16 channels per group and D' <= 48:
//   LEFT  side:  dL[c][w] = 1/cpg * sum_d gv[g(c)][d][w]     * R[c][w - d]      dLc[c][w] = sum_d gv[G + c][d][w] (masked w >= d)
//   RIGHT side:  dR[c][x] = 1/cpg * sum_d gv[g(c)][d][x + d] * L[c][x + d]      dRc[c][x] = sum_d gv[G + Cc + c][d][x + d]
// gv = gradient of the volume, NDHWC [B][D'][H][W][G + 2 Cc].
//
// Why.  The first-generation kernel (cost_volume_bwd_g8_kernel) is instruction- and latency-bound: 443 VALU instructions
// per wave and 8-disparity slice for 128 useful FMAs, 2 barriers per slice, ~4 waves per SIMD: 0.31 ms = 24 % of the HBM
// roofline at 576x960 although HBM sees the volume only once (PMC: FETCH 514 MB).
//
// Formulation.  A MACRO-UNIT is (b, h, side, 16 output columns), all D'; it is walked in CHUNKS of 8 disparities.  In
// "tile coordinates" j = 0..15 both sides look alike: the chunk image is img[dd][j][g] = gv[g][d0 + dd][col(j, dd)] with
// col = w0 + j (LEFT) or x0 + j + d0 + dd (RIGHT: the sheared set), and the partner feature column of (j, dd) is
// f = w0 + j - d0 - dd (LEFT, features R) or x0 + j + d0 + dd (RIGHT, features L).  For one feature column f the update
//   out[c][j] += feat[c][f] * img[dd(j, f)][j][g(c)]          (dd = j + 7 - fi  |  fi - j,  fi = f - first column of the chunk)
// is an outer product per group, i.e. one v_mfma_f32_4x4x1_16B_f32: 16 blocks = 4 groups x 4 column quads, block rows =
// 4 channels of the group, block columns = the 4 columns of the quad, K = 1 = the feature column.  A chunk needs 23 feature
// columns; per column and group quad one ds_read_b32 gathers the B operand (the image diagonal; entries outside the chunk
// read a zero word) and one ds_read_b32 per channel quad the A operand (broadcast over the column quads).
// LDS layouts make both gathers conflict-free for both sides: image voxel stride G, d-row pitch 16 G + pad with
// (pitch + G) mod 32 = 4  (LEFT walks +pitch+G per column, RIGHT -pitch+G: both odd multiples of 4 banks, the 4 groups of a
// block row fill the gaps); feature ring [64 columns][G][4][channel quad] (one 8-byte read = both A operands of a lane).
//
// Roles inside a workgroup (one per CU):
//   * LOADER waves: lane = (column j, float4 q of the voxel), one disparity row per load, NS chunks in flight in registers
//     (8 x 16 B per lane each); gwc quads -> ds_write_b128 into the image (double buffered), concat quads are summed in the
//     loader's registers (a lane keeps its (j, q) for the whole macro-unit: the RIGHT side's shear makes x = x0 + j constant
//     as well) and written at the end of the macro-unit.  All loads are buffer loads: rows past D', columns past W, the
//     masked part of the left concat half and the quads a side does not need get an out-of-range offset and read zeros
//     (no clamps, no selects, no divergent code: the first version of this loader spent 1 300 instructions per chunk,
//     mostly exec-mask and spill traffic, and that -- not memory -- set the kernel's time).  The feature ring slides with the tiles of an image row: 16 new
//     columns per macro-unit (every feature element is read once per row and side), a full refill at row starts.
//   * COMPUTE waves (4, i.e. 8 waves = 2 per SIMD with the loaders: 256 VGPRs each): up to three group quads each; per chunk 23 x (1 + cpg/4) LDS reads and 23 x cpg/4 MFMAs per group quad;
//     accumulators (4 VGPRs per group quad and channel quad) live for the whole macro-unit and are written as 64-byte
//     row segments of the NCHW gradients.
//   One barrier per chunk; LEFT macro-units walk d downwards and RIGHT ones upwards so that the ring slots the next tile
//   overwrites are dead by the time of the last chunk.
// Schedules.  RUN schedule (general): macro-units are dealt in equal contiguous runs to gridDim.x workgroups; the ring
// slides along the row, but the two sides of a row touch the same part of the volume many microseconds apart: L2 (4 MB
// per XCD) has lost it by then and the volume crosses the fabric twice (PMC: FETCH 1.03 GB, 0.30 ms whatever the prefetch
// depth).  TEAM schedule (chosen when 16 < 2 x tiles per row <= 32 and 40 < D' <= 48, i.e. the 576x960 benchmark shape):
// the 32 workgroups of an XCD form a team that owns an image row at a time, member = (tile, side); every member walks
// d upwards, so at any moment the whole team reads the same 8 disparity planes of the row (~0.5 MB) -- whoever comes
// first pays the miss, the others hit L2 and catch up.  A member keeps its tile while the rows change, so its feature
// ring cannot slide; the 16-column batches of a macro-unit's window are fetched progressively instead (the first two
// during the last chunks of the previous macro-unit: with 6 chunks those slots are dead by then), 4 x more feature
// traffic, all of it L2 hits within the team.
//
// Roofline: HBM; algorithmic bytes = volume once + features once + feature gradients once.
#include "cost_volume.h"
#include "stx_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int CVB2_T = 16;        // columns per macro-unit
constexpr int CVB2_DC = 8;        // disparities per chunk
constexpr int CVB2_NF = CVB2_T + CVB2_DC - 1;   // feature columns per chunk (23)
constexpr int CVB2_RING = 64;     // feature ring columns (window D' + 15 <= 63, + slack for the 16 incoming ones)
constexpr int CVB2_NCW = 4;       // compute waves
constexpr int CVB2_NLW = 4;       // loader waves
constexpr int CVB2_NLTHR = CVB2_NLW * 64;

struct CvbArgs {
    const float *gv, *Lg, *Rg;
    float *gLg, *gRg, *gLc, *gRc;
    int B, H, W, D, G, Cc, mask_left;
    int nt, nch, macros;           // tiles per row, chunks per macro-unit, B * H * 2 * nt
    int team, nteams;              // row-team schedule (see above) and the number of teams (= XCDs)
};

struct CvbCursor {                 // a chunk of the workgroup's run: macro-unit m (decoded) and step i of its d walk
    int m, i, b, h, t, side;
};

// image geometry for G groups: voxel stride G, d-row pitch 16 G + pad with (pitch + G) mod 32 == 4; ring column stride
__host__ __device__ constexpr int cvb_pitch(int G) { return CVB2_T * G + (((4 - 17 * G) % 32) + 32) % 32; }

__device__ __forceinline__ int cvb_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

__device__ __forceinline__ void cvb_decode(const CvbArgs& a, int m, CvbCursor& c) {
    c.m = m;
    c.t = m % a.nt;
    const int r = m / a.nt;
    c.side = r & 1;
    const int bh = r >> 1;
    c.h = bh % a.H;
    c.b = bh / a.H;
}
__device__ __forceinline__ void cvb_advance(const CvbArgs& a, CvbCursor& c) {   // (no divisions: runs every step)
    if (++c.i == a.nch) {
        c.i = 0;
        ++c.m;
        if (a.team) {                                  // same tile and side of the team's next image row
            c.h += a.nteams;
            while (c.h >= a.H) { c.h -= a.H; ++c.b; }
        } else if (++c.t == a.nt) {
            c.t = 0;
            if ((c.side ^= 1) == 0 && ++c.h == a.H) { c.h = 0; ++c.b; }
        }
    }
}
// first disparity of the chunk.  Run schedule: LEFT walks d downwards, RIGHT upwards (ring reuse along the row); team
// schedule: both upwards (the two sides and all tiles of a row read the same 8 disparity planes at the same time)
__device__ __forceinline__ int cvb_d0(const CvbArgs& a, const CvbCursor& c) {
    return CVB2_DC * ((c.side || a.team) ? c.i : a.nch - 1 - c.i);
}

__device__ __forceinline__ f32x4 cvb_zero4() {
    f32x4 z;
    z[0] = 0.f; z[1] = 0.f; z[2] = 0.f; z[3] = 0.f;
    return z;
}

// CPG channels per group; QPW group quads per compute wave (4 QPW >= G / 4); NS chunks of the volume gradient in flight in
// the loader waves' registers; GT / CCT: G and Cc as compile-time constants (0 / -1: taken from the arguments) -- with them
// every LDS offset of the inner loops is an instruction immediate.
// (Eight compute waves -- three waves per SIMD, the ten group quads of the GwcNet volume in 2 rounds per chunk instead of 3 --
//  measured the same 0.225 ms as four: GPU call G of round 3.  Neither the compute waves nor, with the team schedule, the
//  memory traffic alone set this kernel's time.)
template <int CPG, int QPW, int NS, int GT, int CCT>
__global__ __launch_bounds__((CVB2_NCW + CVB2_NLW) * 64) void cost_volume_bwd_mfma_kernel(CvbArgs a) {
    constexpr int NCW = CVB2_NCW;
    constexpr int NQ = CPG / 4;                                    // channel quads per group
    constexpr int NFR = 5 * CPG / 2;                               // feature loads per loader lane and 16-column batch: Cg <= 40 CPG
    STX_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, D = a.D, nch = a.nch;
    const int G = GT ? GT : a.G, Cc = CCT >= 0 ? CCT : a.Cc;
    const int HW = H * W, Cg = G * CPG, CT = G + 2 * Cc;
    const int GQ = G >> 2, CQ = Cc >> 2, Q = CT >> 2;
    const int S = G, PD = cvb_pitch(G), FS = Cg + 4;
    const int IMG = CVB2_DC * PD;                                  // dwords per image
    const int IMGZ = IMG + G;                                      // image + G zero words (B operand of entries outside the chunk)
    float* lds = reinterpret_cast<float*>(smem);                   // [2][IMGZ] (double buffer over chunks)
    float* ring = lds + 2 * IMGZ;                                  // [CVB2_RING][FS]

    const long long wg = cvb_xcd_remap(blockIdx.x, gridDim.x);
    int m0, m1;
    CvbCursor cc;                                                  // the chunk the compute waves work on
    if (a.team) {
        // logical workgroups 32 x .. 32 x + 31 share XCD x (gridDim.x == 256): team x, member = (tile, side), rows x, x + 8, ..
        const int team = (int)(wg >> 5), member = (int)(wg & 31), rows = a.B * H;
        if (member >= 2 * a.nt || team >= rows) return;
        m0 = 0;
        m1 = __builtin_amdgcn_readfirstlane((rows - team + a.nteams - 1) / a.nteams);
        cc.m = 0; cc.t = member >> 1; cc.side = member & 1;
        cc.h = team % H; cc.b = team / H;
    } else {
        m0 = __builtin_amdgcn_readfirstlane((int)((long long)a.macros * wg / gridDim.x));
        m1 = __builtin_amdgcn_readfirstlane((int)((long long)a.macros * (wg + 1) / gridDim.x));
        if (m0 >= m1) return;
        cvb_decode(a, m0, cc);
    }
    cc.i = 0;
    const int N = (m1 - m0) * nch;                                 // chunks of this run
    // run schedule: the next macro-unit continues the image row of `c` (ring slides by 16 columns) / starts a new row or
    // side (refill)
    auto next_slides = [&](const CvbCursor& c) { return !a.team && c.m + 1 < m1 && c.t + 1 < a.nt; };
    auto next_refills = [&](const CvbCursor& c) { return !a.team && c.m + 1 < m1 && c.t + 1 == a.nt; };
    // team schedule: first column of 16-column batch j of the window of macro-unit c (j = 0: the tile's own columns)
    auto batch0 = [&](const CvbCursor& c, int jb) { return c.t * CVB2_T + (c.side ? 16 * jb : -16 * jb); };

    // ---- feature ring helpers (both roles: a refill is shared): lane = (column ft & 15, channel (ft >> 4) + 16 i), 16
    // columns per batch; ring layout [slot][group][4][channel quad] (a lane's A operands of all quads: one 8- or 16-byte read).  The loads always execute (`live` = false: every lane
    // out of range), so the number of loads in flight is the same on every path and hipcc's waits stay counted.
    const int ft = tid & (CVB2_NLTHR - 1);                          // (256 lanes per role)
    const int fcol = ft & 15, fc0 = ft >> 4;
    auto ring_chan = [&](int ch) {                                  // offset of channel ch inside a ring column: [group][4][quad]
        const int g = ch / CPG, ci = ch - g * CPG;
        return (g * 4 + (ci & 3)) * NQ + (ci >> 2);
    };
    const int rchan0 = ring_chan(fc0);
    auto feat_issue = [&](float (&dst)[NFR], const CvbCursor& c, int col0, bool live) {
        const stx_bufrsrc rs = stx_make_rsrc((c.side ? a.Lg : a.Rg) + (size_t)c.b * Cg * HW, (unsigned)Cg * (unsigned)HW * 4u);
        const int col = col0 + fcol;
        const unsigned voff = (live && col >= 0 && col < W) ? (unsigned)(fc0 * HW + col) * 4u : STX_BUF_OOB;
        const unsigned srow = (unsigned)(c.h * W) * 4u, sstep = 16u * (unsigned)HW * 4u;
#pragma unroll
        for (int i = 0; i < NFR; ++i) dst[i] = stx_buf_ld1(rs, voff, srow + (unsigned)i * sstep);   // channels >= Cg: out of range
    };
    auto feat_commit = [&](const float (&src)[NFR], int col0) {     // (columns outside the image were read as zeros)
        float* rp = ring + ((col0 + fcol) & (CVB2_RING - 1)) * FS;
        if constexpr (16 % CPG == 0) {                              // channel + 16 = group + 16 / CPG = 16 dwords further
            rp += rchan0;
#pragma unroll
            for (int i = 0; i < NFR; ++i)
                if (fc0 + 16 * i < Cg) rp[i * 16] = src[i];
        } else {
#pragma unroll
            for (int i = 0; i < NFR; ++i)
                if (fc0 + 16 * i < Cg) rp[ring_chan(fc0 + 16 * i)] = src[i];
        }
    };
    // first column of the window of macro-unit c: LEFT [w0 - 8 nch + 1, w0 + 15], RIGHT [x0, x0 + 8 nch + 14]
    auto window0 = [&](const CvbCursor& c) { return c.side ? c.t * CVB2_T : c.t * CVB2_T - CVB2_DC * nch + 1; };
    // Full refill of the ring for macro-unit c (run start, new image row or side), while nobody reads the ring: the loader
    // waves fetch the last 16 of the 64 columns, the compute waves (their operand registers are dead here) the first 48 --
    // one round trip to memory instead of four.
    auto ring_refill_loader = [&](const CvbCursor& c) {
        float t0[NFR];
        feat_issue(t0, c, window0(c) + 48, true);
        feat_commit(t0, window0(c) + 48);
    };
    auto ring_refill_compute = [&](const CvbCursor& c) {
        float t0[NFR], t1[NFR], t2[NFR];
        const int f0 = window0(c);
        feat_issue(t0, c, f0, true);
        feat_issue(t1, c, f0 + 16, true);
        feat_issue(t2, c, f0 + 32, true);
        feat_commit(t0, f0);
        feat_commit(t1, f0 + 16);
        feat_commit(t2, f0 + 32);
    };

    if (wave >= NCW) {
        // ======================================================================= loader waves
        const int lt = tid - NCW * 64;
        const int q = lt & 15, j = (lt >> 4) & 15;
        const bool gq_lane = q < GQ;
        const bool lc_lane = q >= GQ && q < GQ + CQ, rc_lane = q >= GQ + CQ && q < Q;
        const int wimg = j * S + 4 * q;                              // my voxel quad inside an image row
        for (int i = lt; i < 2 * G; i += CVB2_NLTHR) lds[(i >= G ? IMGZ + IMG - G : IMG) + i] = 0.f;
        // ---- volume gradient: NS chunks in registers.  Lane (j, q) reads rows d0 .. d0 + 7 of one column (LEFT) or of the
        // sheared diagonal (RIGHT: + one voxel per row, the instruction's immediate offset when CT is a constant).
        float4 gvr[NS][CVB2_DC];
        float4 cacc = make_float4(0.f, 0.f, 0.f, 0.f);
        const unsigned gvbytes = (unsigned)D * (unsigned)HW * (unsigned)CT * 4u;     // one batch item (checked < 2 GiB)
        const unsigned dstep = (unsigned)HW * (unsigned)CT * 4u;
        auto issue = [&](float4 (&dst)[CVB2_DC], const CvbCursor& c, bool live) {
            const stx_bufrsrc rs = stx_make_rsrc(a.gv + (size_t)c.b * D * HW * CT, gvbytes);
            const int d0 = cvb_d0(a, c);
            const int col0 = c.t * CVB2_T + j + (c.side ? d0 : 0);
            // rows this lane wants: k < klim.  LEFT: the column must exist; the left concat half stops at d = w when masked.
            // RIGHT: row k sits at column col0 + k.
            int klim;
            if (c.side) {
                klim = (D - d0 < W - col0) ? D - d0 : W - col0;
                if (!(gq_lane || rc_lane)) klim = 0;
            } else {
                const int dlim = (lc_lane && a.mask_left && col0 + 1 < D) ? col0 + 1 : D;
                klim = (col0 < W && (gq_lane || lc_lane)) ? dlim - d0 : 0;
            }
            if (!live) klim = 0;
            const unsigned voff0 = (unsigned)(col0 * CT + 4 * q) * 4u, kstep = c.side ? (unsigned)CT * 4u : 0u;
            const unsigned srow = (unsigned)((d0 * H + c.h) * W) * (unsigned)CT * 4u;
#pragma unroll
            for (int k = 0; k < CVB2_DC; ++k)
                dst[k] = stx_buf_ld4(rs, k < klim ? voff0 + (unsigned)k * kstep : STX_BUF_OOB, srow + (unsigned)k * dstep);
        };
        auto commit = [&](const float4 (&src)[CVB2_DC], const CvbCursor& c, float* img) {
            if (gq_lane) {
#pragma unroll
                for (int k = 0; k < CVB2_DC; ++k) stx_st4(img + k * PD + wimg, src[k]);
            } else {
#pragma unroll
                for (int k = 0; k < CVB2_DC; ++k) { cacc.x += src[k].x; cacc.y += src[k].y; cacc.z += src[k].z; cacc.w += src[k].w; }
            }
            if (c.i == nch - 1) {                                    // last chunk of the macro-unit: concat sums are complete
                const int col = c.t * CVB2_T + j;
                const int qc = c.side ? q - GQ - CQ : q - GQ;
                if (!gq_lane && qc >= 0 && qc < CQ && col < W) {
                    float* o = (c.side ? a.gRc : a.gLc) + ((size_t)(c.b * Cc + 4 * qc) * H + c.h) * W + col;
                    o[0] = cacc.x; o[(size_t)HW] = cacc.y; o[2 * (size_t)HW] = cacc.z; o[3 * (size_t)HW] = cacc.w;
                }
                cacc = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        float fr[NFR];
        // the 16 columns the NEXT macro-unit of the row adds to the window of c (run schedule)
        auto incoming0 = [&](const CvbCursor& c) {
            return c.side ? c.t * CVB2_T + CVB2_DC * nch + CVB2_T - 1 : c.t * CVB2_T + CVB2_T;
        };

        // ---- prologue: chunks 0 .. NS-1 requested, chunk 0 in the image
        CvbCursor wc = cc, pc = cc;                                  // write cursor (chunk ci + 1), prefetch cursor (ci + NS)
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (k > 0 && k < N) cvb_advance(a, pc);
            issue(gvr[k], pc, k < N);
        }
        if (N > 1) cvb_advance(a, wc);
        ring_refill_loader(cc);
        commit(gvr[0], cc, lds);
        __syncthreads();
        auto step = [&](auto par, int ci) {
            constexpr int P = decltype(par)::value;                  // ci % NS: set P holds chunk ci (already in the image)
            // Loads retire in order and hipcc counts them only along straight-line code, so every step executes the same
            // loads in the same order: [NFR feature loads][8 loads of chunk ci + NS].  What differs is only their offsets
            // (a batch of ring columns is needed in some steps: in the others every lane is out of range) and which results
            // are committed: the ring batch requested in the previous step first (8 younger loads behind it), then chunk
            // ci + 1 (the loads of chunks ci + 2 .. ci + NS and of the feature batches between them behind it).
            CvbCursor nx = cc;                                       // my next macro-unit
            nx.i = nch - 1;
            cvb_advance(a, nx);
            const bool more = cc.m + 1 < m1;
            int ccol = 0, icol = 0;                                  // ring batch to commit now / to request now
            bool cdo = false, ido = false, inext = false;
            if (a.team) {
                if (cc.i == 1) { cdo = true; ccol = batch0(cc, 2); }
                else if (cc.i == 3) { cdo = true; ccol = batch0(cc, 3); }
                else if (cc.i == nch - 2 && more) { cdo = true; ccol = batch0(nx, 0); }
                else if (cc.i == nch - 1 && more) { cdo = true; ccol = batch0(nx, 1); }
                if (cc.i == 0) { ido = true; icol = batch0(cc, 2); }
                else if (cc.i == 2) { ido = true; icol = batch0(cc, 3); }
                else if (cc.i == nch - 3 && more) { ido = inext = true; icol = batch0(nx, 0); }
                else if (cc.i == nch - 2 && more) { ido = inext = true; icol = batch0(nx, 1); }
            } else {
                if (cc.i == nch - 1 && next_slides(cc)) { cdo = true; ccol = incoming0(cc); }
                if (cc.i == nch - 2 && next_slides(cc)) { ido = true; icol = incoming0(cc); }
            }
            if (cdo) feat_commit(fr, ccol);
            feat_issue(fr, inext ? nx : cc, icol, ido);
            if (ci + NS < N) cvb_advance(a, pc);
            issue(gvr[P], pc, ci + NS < N);
            if (ci + 1 < N) commit(gvr[(P + 1) % NS], wc, lds + ((ci + 1) & 1) * IMGZ);
            __syncthreads();
            if (cc.i == nch - 1 && next_refills(cc)) {
                CvbCursor rf = cc;
                cvb_advance(a, rf);
                ring_refill_loader(rf);
                __syncthreads();
            }
            if (ci + 1 < N) cvb_advance(a, wc);
            cvb_advance(a, cc);
        };
        for (int ci = 0; ci < N; ci += NS) {
            step(std::integral_constant<int, 0>{}, ci);
            if (ci + 1 < N) step(std::integral_constant<int, 1 % NS>{}, ci + 1);
            if constexpr (NS > 2) {
                if (ci + 2 < N) step(std::integral_constant<int, 2 % NS>{}, ci + 2);
            }
            if constexpr (NS > 3) {
                if (ci + 3 < N) step(std::integral_constant<int, 3 % NS>{}, ci + 3);
            }
        }
        return;
    }

    // =========================================================================== compute waves
    const int n = lane & 3, gl = (lane >> 2) & 3, jq = lane >> 4;
    const int j = 4 * jq + n;
    const int joff = j * S + gl + 4 * wave;                          // B operand: img[dd][j][4 gq + gl], gq = wave + u NCW
    const float* abase = ring + ((lane & 15) + 16 * wave) * NQ;      // A operand: ring[slot][4 gq + gl][n][cq]
    const float inv = 1.0f / (float)CPG;
    f32x4 acc[QPW][NQ];
#pragma unroll
    for (int u = 0; u < QPW; ++u)
#pragma unroll
        for (int cq = 0; cq < NQ; ++cq) acc[u][cq] = cvb_zero4();
    // image offset of the B operand per feature column of a chunk; depends on the side only: the entry (j, dd) pairs with
    // column fi = j + 7 - dd (LEFT) / j + dd (RIGHT); entries outside the chunk read the zero words behind the image
    int bo[CVB2_NF];
    int bo_side = -1;
    ring_refill_compute(cc);
    __syncthreads();                                                 // prologue: ring + image 0
    for (int ci = 0; ci < N; ++ci) {
        if (cc.side != bo_side) {
            bo_side = cc.side;
#pragma unroll
            for (int fi = 0; fi < CVB2_NF; ++fi) {
                const int dd = cc.side ? fi - j : j + (CVB2_DC - 1) - fi;
                bo[fi] = ((unsigned)dd < (unsigned)CVB2_DC) ? dd * PD + joff : IMG + gl + 4 * wave;
            }
        }
        const float* img = lds + (ci & 1) * IMGZ;
        const int d0 = cvb_d0(a, cc);
        const int tile0 = cc.t * CVB2_T;
        const int fbase = cc.side ? tile0 + d0 : tile0 - d0 - (CVB2_DC - 1);
        // operand addresses of the chunk's 23 feature columns, shared by all my group quads (constant offsets between them)
        const float *bp[CVB2_NF], *ap[CVB2_NF];
#pragma unroll
        for (int fi = 0; fi < CVB2_NF; ++fi) {
            bp[fi] = img + bo[fi];
            ap[fi] = abase + ((fbase + fi) & (CVB2_RING - 1)) * FS;
        }
        // Software pipeline over groups of 6 feature columns (12 LDS reads: the LDS counter tracks at most 15): the reads of
        // group gp are issued before the MFMAs of group gp - 1, so the matrix pipe works while the next operands arrive.
        constexpr int GS = 6, NGU = (CVB2_NF + GS - 1) / GS;         // 4 groups per group quad
        float bv[2][GS], av[2][GS][NQ];
#pragma unroll
        for (int gp = 0; gp <= NGU * QPW; ++gp) {
            if (gp < NGU * QPW) {
                const int u = gp / NGU, f0 = (gp % NGU) * GS;
                if (wave + u * NCW < GQ) {                           // wave-uniform
#pragma unroll
                    for (int t = 0; t < GS; ++t)
                        if (f0 + t < CVB2_NF) {
                            bv[gp & 1][t] = bp[f0 + t][4 * NCW * u];
                            const float* rp = ap[f0 + t] + 16 * NCW * NQ * u;
                            if constexpr (NQ == 2) {
                                const float2 v2 = *reinterpret_cast<const float2*>(rp);
                                av[gp & 1][t][0] = v2.x; av[gp & 1][t][1] = v2.y;
                            } else if constexpr (NQ == 4) {
                                const float4 v4 = stx_ld4(rp);
                                av[gp & 1][t][0] = v4.x; av[gp & 1][t][1] = v4.y; av[gp & 1][t][2] = v4.z; av[gp & 1][t][3] = v4.w;
                            } else {
#pragma unroll
                                for (int cq = 0; cq < NQ; ++cq) av[gp & 1][t][cq] = rp[cq];
                            }
                        }
                }
            }
            STX_SCHED_BARRIER();
            if (gp > 0) {
                const int u = (gp - 1) / NGU, f0 = ((gp - 1) % NGU) * GS;
                if (wave + u * NCW < GQ) {
#pragma unroll
                    for (int t = 0; t < GS; ++t)
                        if (f0 + t < CVB2_NF) {
#pragma unroll
                            for (int cq = 0; cq < NQ; ++cq)
                                acc[u][cq] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[(gp - 1) & 1][t][cq], bv[(gp - 1) & 1][t],
                                                                                 acc[u][cq], 0, 0, 0);
                        }
                }
            }
            STX_SCHED_BARRIER();
        }
        if (cc.i == nch - 1) {
            // macro-unit complete: lane holds out[channel 4 cq + r of group 4 gq + gl][column j] in acc[u][cq][r]
            float* gout = (cc.side ? a.gRg : a.gLg) + ((size_t)cc.b * Cg * H + cc.h) * W + tile0 + j;
            const bool okc = tile0 + j < W;
#pragma unroll
            for (int u = 0; u < QPW; ++u) {
                const int gq = wave + u * NCW;
                if (gq < GQ) {
#pragma unroll
                    for (int cq = 0; cq < NQ; ++cq) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = (4 * gq + gl) * CPG + 4 * cq + r;
                            if (okc) gout[(size_t)c * HW] = acc[u][cq][r] * inv;
                        }
                        acc[u][cq] = cvb_zero4();
                    }
                }
            }
        }
        __syncthreads();
        if (cc.i == nch - 1 && next_refills(cc)) {                   // ring refill for the next image row / side
            CvbCursor nx = cc;
            cvb_advance(a, nx);
            ring_refill_compute(nx);
            __syncthreads();
        }
        cvb_advance(a, cc);
    }
}

template <int CPG, int QPW, int NS, int GT, int CCT>
int cvb_launch(const CvbArgs& a, size_t lds, hipStream_t st) {
    auto kern = cost_volume_bwd_mfma_kernel<CPG, QPW, NS, GT, CCT>;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int grid = 256;
    if (stx_tune(STX_TUNE_CVB_GRID) > 0) grid = stx_tune(STX_TUNE_CVB_GRID);           // tests: force short / long runs
    if (grid > a.macros) grid = a.macros;
    if (a.team) grid = 256;                                 // 8 XCDs x 32 members
    hipLaunchKernelGGL(kern, dim3(grid), dim3((CVB2_NCW + CVB2_NLW) * 64), lds, st, a);
    return stx_check_launch("cost_volume_bwd(mfma)");
}

}  // namespace

// Returns -1 when the configuration is not served by this kernel (caller falls back to cost_volume.hip).
int stx_cv_bwd_mfma(const float* gvol, const float* Lg, const float* Rg, int Cg, int G, int Cc, float* gLg, float* gRg,
                    float* gLc, float* gRc, int B, int H, int W, int D, int mask_left, void* stream) {
    if (stx_tune(STX_TUNE_CVB_OLD) || !G) return -1;
    const int cpg = Cg / G;
    if (!(cpg == 4 || cpg == 8 || cpg == 12 || cpg == 16) || (G & 3) || (Cc & 3) || G > 40) return -1;
    const int CT = G + 2 * Cc, Q = CT / 4;
    if (Q > 16) return -1;                                 // loader lane = (column, one of 16 float4 of the voxel)
    const int nch = stx_cdiv(D, CVB2_DC);
    if (CVB2_DC * nch + CVB2_T - 1 > CVB2_RING - 1) return -1;   // feature window D' + 15 must leave one ring column free
    if (nch < 2) return -1;                                // the ring batch of the next tile is requested one step before it is committed
    CvbArgs a;
    a.gv = gvol; a.Lg = Lg; a.Rg = Rg; a.gLg = gLg; a.gRg = gRg; a.gLc = gLc; a.gRc = gRc;
    a.B = B; a.H = H; a.W = W; a.D = D; a.G = G; a.Cc = Cc; a.mask_left = mask_left;
    a.nt = stx_cdiv(W, CVB2_T); a.nch = nch;
    const long long macros = 2ll * B * H * a.nt;
    if (macros >= (1ll << 30) || (long long)B * Cg * H * W >= (1ll << 31)) return -1;
    // buffer resources of one batch item: < 2 GiB each (32-bit offsets, STX_BUF_OOB beyond every valid one)
    if ((long long)D * H * W * CT * 4 >= (1ll << 31) || (long long)Cg * H * W * 4 >= (1ll << 31)) return -1;
    a.macros = (int)macros;
    // team schedule (see above): one row team per XCD
    // (GPU call O, 576x960: run schedule 0.201 ms, team schedule 0.207-0.222 ms depending on the prefetch depth, first
    // generation 0.303 ms: the team schedule halves the HBM reads but its lock-step costs more than that saves -> opt-in)
    const int team_env = stx_tune(STX_TUNE_CVB_TEAM);
    a.nteams = 8;
    a.team = (team_env && 2 * a.nt > 16 && 2 * a.nt <= 32 && nch == 6 && B * H >= a.nteams && stx_tune(STX_TUNE_CVB_GRID) <= 0) ? 1 : 0;
    const size_t lds = ((size_t)2 * (CVB2_DC * cvb_pitch(G) + G) + (size_t)CVB2_RING * (Cg + 4)) * 4;
    if (lds > 160 * 1024) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int GQ = G / 4;
    // GwcNet / ACVNet volumes (320 channels in 40 groups, 12 or 0 concat channels): constants folded, STX_CVB_NSET chunks
    // in flight per loader lane (A/B switch)
    const int nset = stx_tune(STX_TUNE_CVB_NSET);
    if (cpg == 8 && G == 40 && (Cc == 12 || Cc == 0)) {
        if (Cc == 12) {
            if (nset == 2) return cvb_launch<8, 3, 2, 40, 12>(a, lds, st);
            if (nset == 4) return cvb_launch<8, 3, 4, 40, 12>(a, lds, st);
            return cvb_launch<8, 3, 3, 40, 12>(a, lds, st);
        }
        if (nset == 2) return cvb_launch<8, 3, 2, 40, 0>(a, lds, st);
        return cvb_launch<8, 3, 3, 40, 0>(a, lds, st);
    }
#define CVB_CASE(CPG_)                                                           \
    if (cpg == CPG_) {                                                           \
        if (GQ <= 4) return cvb_launch<CPG_, 1, 2, 0, -1>(a, lds, st);           \
        if (GQ <= 8) return cvb_launch<CPG_, 2, 2, 0, -1>(a, lds, st);           \
        return cvb_launch<CPG_, 3, 2, 0, -1>(a, lds, st);                        \
    }
    CVB_CASE(4) CVB_CASE(8) CVB_CASE(12) CVB_CASE(16)
#undef CVB_CASE
    return -1;
}
