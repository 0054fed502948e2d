// Cost-volume builder, second generation: group-wise correlation on the matrix cores, whole voxels staged in LDS,
// the volume written as one uninterrupted stream of 1-KiB wave stores by waves that never wait on memory.
//
// Replaces the same reference functions as cost_volume.hip (build_gwc_volume GwcNet/submodule.py:53-63,
// build_concat_volume GwcNet/submodule.py:30-41 / PSMNet/stackhourglass.py:111-120 / ACVNet/submodule.py:180-191,
// torch.cat gwcnet.py:180, softmax(att) * concat_volume acv.py:196) for every group configuration with 4, 8, 12 or
// 16 channels per group (GwcNet / ACVNet: 320 ch in 40 groups; IGEV-style volumes: 96 ch in 8 groups) and for
// concat-only volumes (PSMNet), D' <= 96.
//
// Why.  The builder is HBM-bound on paper (516 MB per GwcNet_GC pair, 1 GFLOP), but the first-generation kernels were
// VALU/LDS-latency-bound: one lane per voxel channel = ~25 instructions and 3 LDS reads per 256-byte voxel, every
// feature element re-staged once per 16-disparity chunk (0.156-0.175 ms = 37-41 % of the HBM roofline; a plain
// 425 MB store stream reaches 6.6 TB/s on this chip, tools/ubench/store_stream.hip).
// Here the correlation of one image row is a banded batch of tiny GEMMs:
//   for group g:  C_g[w][x] = sum_{c in g} L[c][w] * R[c][x],   vol[d = w - x][w][g] = C_g[w][x] / cpg,  0 <= w - x < D'
//
// Work decomposition.  A MACRO-UNIT is (b, h, 16 columns w0..): all D' disparities of one 16-column tile; it consists of
// nd = D'/16 UNITS (16 disparities each).  The x-range of unit k, [w0 - 16k - 15, w0 - 16k + 15], lies in the aligned
// 16-column tiles t-k and t-k-1 of R, so a macro-unit needs the left tile t and the right tiles t .. t-nd, and the next
// macro-unit of the row needs ONE new left and ONE new right tile: the right tiles rotate through a register ring.
// Every feature element is therefore loaded from memory exactly once per image row, straight into the MFMA operand
// layout (one dword per lane, 64-byte row segments of the NCHW features) -- there is no transposing LDS image of the
// features and no reliance on L2 reuse, and the loads of macro-unit m+1 are in flight during the nd units of m.
//
// Roles inside a workgroup (one per CU for the 64-channel volumes):
//   * COMPUTE waves (one or two channel quads = 4 or 8 groups each): loads -> per unit and group two 16x16 tiles with
//     v_mfma_f32_16x16x4_f32 (K = the group's channels, exact fp32, k-ordered fmaf chain) -> every lane keeps, of the
//     two results it holds for a (w, x) pair, the one whose d = w - x falls into the unit (tile 0 if w_l >= x_l else
//     tile 1) -> ds_write_b128 into an LDS image [16 d][16 w][G] (conflict-free: voxel stride G+4, d-row stride
//     16(G+4)+4 dwords).  Concat features go through small LDS tables [column][Cc], once per macro-unit.
//   * STORE waves: walk the image of the previous unit in memory order, one float4 per lane (gwc quads from the image,
//     left quads from the left table (masked), right quads from the right table at x = w - d; optional attention
//     scale): a wave stores 1 KiB contiguous, a d-row of a unit is one 4-KiB run.  They issue nothing but LDS reads
//     and global stores and never wait on a memory counter, so the store stream of a CU is limited by the hardware
//     queues only (loads and stores of a wave retire through one in-order counter: a wave that does both ends up
//     draining its stores whenever it waits for a load).
//   The LDS image is double buffered: one barrier per unit hands image i to the store waves while the compute waves
//   build image i+1.  Invalid entries (w < d, i.e. x < 0) are zero by construction: tiles left of the image are zeros.
// Macro-units are dealt in equal contiguous runs to exactly gridDim.x workgroups (no tail round).
//
// Roofline: HBM; algorithmic bytes = features once + volume once (516 464 640 B GwcNet_GC, 433 520 640 B
// PSMNet concat, 576x960, D' = 48).
#include "cost_volume.h"
#include "stx_common.h"
#include <stdlib.h>

namespace {

constexpr int CVM_T = 16;          // columns per unit = disparities per unit = MFMA tile edge
constexpr int CVM_MAXCC = 32;      // concat channels per side

struct CvmArgs {
    const float *Lg, *Rg, *Lc, *Rc, *scale;
    float* vol;
    int B, H, W, D, G, Cc, mask_left;
    int nd, nt, macros;             // d-chunks per macro-unit, w-tiles per row, B*H*nt
    int by_units;                   // work split at unit granularity (a workgroup's run may begin / end inside a macro-unit)
    int win;                        // macro-units per WINDOW: the launch walks the volume window by window, every window split over all
                                    // workgroups (round 6: the chip then writes ONE neighbourhood of every d-plane at a time); >= macros: one window
    unsigned magic_rowq, magic_q;   // ceil(2^32 / (16 Q)), ceil(2^32 / Q)
    unsigned magic_tc;              // ceil(2^32 / table columns)
    int nontemporal;
};

__device__ __forceinline__ int cvm_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

template <int V> struct cvm_int { static constexpr int value = V; };

__device__ __forceinline__ f32x4 cvm_zero4() {
    f32x4 z;
    z[0] = 0.f; z[1] = 0.f; z[2] = 0.f; z[3] = 0.f;
    return z;
}

// concat-table row stride (dwords): a multiple of 4 (float4 reads) whose quarter is odd, so that the transposing
// dword writes of consecutive columns spread over 8 bank groups
__host__ __device__ inline int cvm_cs(int Cc) { return ((Cc / 4) & 1) ? Cc + 8 : Cc + 4; }

// ND = compile-time bound of nd (register ring of ND + 1 right tiles).
// PF = feature prefetch scheme.  1: the tiles of macro-unit m + 1 are requested during m.  A 16-column tile is one 64-byte
//      piece of each channel row, i.e. HALF a 128-byte cache line; its other half belongs to the neighbouring tile, which the
//      same workgroup requests one macro-unit (~10 us, ~40 KB of other lines per CU, a whole L2 turn-over of streamed volume
//      per XCD) later -- by then the line is gone and is fetched again: FETCH_SIZE 1.76 x the feature bytes (round 3).
//   2: tiles are requested in PAIRS, both halves of every line back to back (even macro index = even global tile index when
//      W % 32 == 0 or rows are 64-byte-phase aligned; other shapes only lose the pairing), at the start of every odd
//      macro-unit for the following two.
template <int CPG, int QPW, int NCW, int NSW, int ND, bool SCALE, int PF>
__global__ __launch_bounds__((NCW + NSW) * 64) void cost_volume_fwd_mfma_kernel(CvmArgs a) {
    constexpr int NCTHR = NCW * 64, NSTHR = NSW * 64;
    constexpr int KK = CPG / 4;                                   // MFMA K steps per group
    constexpr int TCMAX = CVM_T * (ND + 2);                        // table columns: 16 left + 16 (nd + 1) right
    constexpr int NCL = (CVM_MAXCC * TCMAX + NCTHR - 1) / NCTHR;   // concat table elements per compute thread
    STX_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, D = a.D, G = a.G, Cc = a.Cc, nd = a.nd;
    const int HW = H * W, Cg = G * CPG, CT = G + 2 * Cc;
    const int GQ = G >> 2, CQ = Cc >> 2, Q = CT >> 2;
    const int VS = G + 4, DS = CVM_T * VS + 4, CS = cvm_cs(Cc);
    const int TC = CVM_T * (nd + 2);                               // table columns in use
    // LDS: two images [16 dd][DS] (double buffer over units) | two table sets [TC cols][CS] (double buffer over macros)
    const int IMG = G ? CVM_T * DS : 0, TAB = Cc ? TC * CS : 0;
    // (PF = 2: the compute waves' LDS-DMA slots come first -- the DMA destination base travels in M0 -- then images, tables)
    float* lds = reinterpret_cast<float*>(smem) + (PF == 2 ? NCW * QPW * 8 * KK * 64 : 0);
    float* tabs = lds + 2 * IMG;

    // The workgroup's run inside one WINDOW [wm0, wm1) of macro-units: units [u0, u1) of the window's (wm1 - wm0) * nd units in
    // (macro, k) order; whole macro-units unless by_units (2160 macro-units over 256 workgroups = 8 or 9 each: 6.6 % of the chip
    // idles through the last one; 6480 units = 25 or 26 each: 2.7 %).  A run that begins inside a macro-unit builds the ring like
    // any first macro-unit and skips the units in front of its range.  One window (a.win >= a.macros) is the scheme of rounds
    // 2-5: every workgroup walks its own contiguous piece of the whole volume.
    const long long wg = cvm_xcd_remap(blockIdx.x, gridDim.x);
    const int nwin = (a.macros + a.win - 1) / a.win;
    int m0 = 0, m1 = 0, kfirst = 0, klast = 0;                     // macro-units touched in this window, unit range in the first / last
    auto set_window = [&](int r) {
        const int wm0 = r * a.win, wm1 = (wm0 + a.win < a.macros) ? wm0 + a.win : a.macros;
        const long long wmac = wm1 - wm0, units = wmac * nd;
        const int u0 = __builtin_amdgcn_readfirstlane(wm0 * nd + (a.by_units ? (int)(units * wg / gridDim.x)
                                                                             : nd * (int)(wmac * wg / gridDim.x)));
        const int u1 = __builtin_amdgcn_readfirstlane(wm0 * nd + (a.by_units ? (int)(units * (wg + 1) / gridDim.x)
                                                                             : nd * (int)(wmac * (wg + 1) / gridDim.x)));
        m0 = u0 / nd;
        m1 = u1 > u0 ? (u1 + nd - 1) / nd : m0;
        kfirst = u0 - m0 * nd;
        klast = u1 - (m1 - 1) * nd;
    };
    auto decode = [&](int m, int& b, int& h, int& t) {
        t = m % a.nt;
        const int r = m / a.nt;
        h = r % H;
        b = r / H;
    };

    if (wave >= NCW) {
        // ================= store waves: flush the image of unit i while the compute waves build unit i+1.
        // Everything that does not depend on the unit is computed once per lane: slot s of a lane is float4 number
        // stid + s * NSTHR of the unit's 16 x 16 voxels in memory order -> (dd, wl, quad), its LDS source (gwc image /
        // left table / right table) and its offset in the volume.  A unit then costs a lane one LDS read, a few
        // selects and one 16-byte store per slot.
        constexpr int NSLOT = (CVM_T * CVM_T * 16 + NSTHR - 1) / NSTHR;      // Q <= 16 (checked by the host)
        const int stid = tid - NCTHR;
        const int rowq = CVM_T * Q, total = CVM_T * rowq;
        const size_t dstride = (size_t)HW * CT;
        int pk[NSLOT], soff[NSLOT];
        unsigned goff[NSLOT];
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const int idx = stid + sl * NSTHR;
            const int dd = (int)__umulhi((unsigned)idx, a.magic_rowq);
            const int rem = idx - dd * rowq;
            const int wl = (Q == 1) ? rem : (int)__umulhi((unsigned)rem, a.magic_q);   // (2^32 / 1 does not fit the magic)
            const int q = rem - wl * Q;
            const int ty = q < GQ ? 0 : (q < GQ + CQ ? 1 : 2);
            const bool live = idx < total;
            pk[sl] = live ? ((ty << 16) | (dd << 8) | wl) : -1;
            soff[sl] = ty == 0 ? dd * DS + wl * VS + 4 * q
                               : (ty == 1 ? wl * CS + 4 * (q - GQ) : (wl - dd) * CS + 4 * (q - GQ - CQ));
            goff[sl] = (unsigned)((size_t)dd * dstride + 4 * (size_t)rem);   // < 2^32 floats (checked by the host)
        }
        for (int r = 0; r < nwin; ++r) {
        set_window(r);
        if (r) STX_BARRIER_LDS();                                    // window hand-over: the images / tables start over at index 0
        int ui = 0;
        for (int m = m0; m < m1; ++m) {
            int b, h, t;
            decode(m, b, h, t);
            const int w0 = t * CVM_T;
            const float* tab = tabs + ((m - m0) & 1) * TAB;          // left table at column 0, right table from column 16
            const int kb = m == m0 ? kfirst : 0, ke = m == m1 - 1 ? klast : nd;
            for (int k = kb; k < ke; ++k, ++ui) {
                // image ui & 1 is complete.  Concat-only volumes (G = 0: PSMNet, ACVNet) have no image: the store waves need the
                // macro-unit's feature TABLES only, one hand-over per macro-unit instead of one per unit
                if (G || k == kb) STX_BARRIER_LDS();
                const int d0 = k * CVM_T;
                const float* stage = lds + (ui & 1) * IMG;
                const float* tabr = tab + (CVM_T + CVM_T * (nd - k)) * CS;   // right-table column of x = w0 - d0
                float* vrow = a.vol + (((size_t)b * D + d0) * H + h) * (size_t)W * CT + (size_t)w0 * CT;
                const float* srow = SCALE ? a.scale + (((size_t)b * D + d0) * H + h) * (size_t)W + w0 : nullptr;
                const int dlim = D - d0, wlim = W - w0, mthr = a.mask_left ? d0 - w0 : -1000;
                // all LDS reads of the unit in flight first, then the masks, then the stores (a wave never waits
                // between two of its reads)
                float4 v[NSLOT];
#pragma unroll
                for (int sl = 0; sl < NSLOT; ++sl) {
                    const int info = pk[sl], ty = info >> 16, dd = (info >> 8) & 0xff, wl = info & 0xff;
                    const bool ok = info >= 0 && dd < dlim && wl < wlim;
                    const float* bp = ty == 0 ? stage : (ty == 1 ? tab : tabr);
                    v[sl] = stx_ld4(ok ? bp + soff[sl] : lds);
                }
                STX_SCHED_BARRIER();          // (hipcc otherwise re-serialises read -> wait -> mask per slot)
                if (SCALE) {
                    float mm[NSLOT];
#pragma unroll
                    for (int sl = 0; sl < NSLOT; ++sl) {
                        const int info = pk[sl], dd = (info >> 8) & 0xff, wl = info & 0xff;
                        const bool ok = info >= 0 && dd < dlim && wl < wlim;
                        mm[sl] = srow[(size_t)(ok ? dd : 0) * HW + (ok ? wl : 0)];
                    }
#pragma unroll
                    for (int sl = 0; sl < NSLOT; ++sl) { v[sl].x *= mm[sl]; v[sl].y *= mm[sl]; v[sl].z *= mm[sl]; v[sl].w *= mm[sl]; }
                }
#pragma unroll
                for (int sl = 0; sl < NSLOT; ++sl) {
                    const int info = pk[sl], ty = info >> 16, dd = (info >> 8) & 0xff, wl = info & 0xff;
                    if (ty == 1 && wl - dd < mthr) v[sl] = make_float4(0.f, 0.f, 0.f, 0.f);   // left half masked where w < d
                }
                if (a.nontemporal) {
#pragma unroll
                    for (int sl = 0; sl < NSLOT; ++sl) {
                        const int info = pk[sl], dd = (info >> 8) & 0xff, wl = info & 0xff;
                        if (info >= 0 && dd < dlim && wl < wlim) stx_st4_nt(vrow + goff[sl], v[sl]);
                    }
                } else {
#pragma unroll
                    for (int sl = 0; sl < NSLOT; ++sl) {
                        const int info = pk[sl], dd = (info >> 8) & 0xff, wl = info & 0xff;
                        if (info >= 0 && dd < dlim && wl < wlim) stx_st4(vrow + goff[sl], v[sl]);
                    }
                }
            }
        }
        }
        return;
    }

    // ================= compute waves
    const int xl = lane & 15, kq = lane >> 4;
    const float inv = 1.0f / (float)CPG;
    // operands of my quads: A = left tile t, Rt[j] = right tile t - j (register ring), nA / nR = the prefetched tile of the
    // next macro-unit.  PF = 2: at every odd macro-unit the tiles of the NEXT TWO are requested together -- the even one
    // into nA / nR, its odd partner (the other half of the same cache lines) by LDS-DMA into this wave's private slot
    // (2 views x 4 groups x KK dwords x 64 lanes per quad; no second register set: with four waves per SIMD the kernel has 128 VGPRs), from where the
    // partner's macro-unit reads it back one macro-unit later.
    float A[QPW][4][KK], Rt[ND + 1][QPW][4][KK], nA[QPW][4][KK], nR[QPW][4][KK];
    float ct[NCL];
    unsigned ctok = 0;                                             // validity bits of ct[]
    bool n_ok = false, d_ok = false;                               // validity of the prefetched tiles' columns (registers / slot)
    float* dslot = reinterpret_cast<float*>(smem) + wave * (QPW * 8 * KK * 64);        // (PF = 2 only; the host sizes the LDS for it)
    // (element offsets in 32 bits from the wave-uniform tensor base: one address register per load instead of a pair; the
    //  host checks that a feature tensor has fewer than 2^30 elements)
    auto tile_off = [&](int b, int h, int cc, int q) {
        return (unsigned)(((b * Cg + 4 * q * CPG + kq) * H + h) * W + cc);
    };
    // one 16-column tile of the gwc features for my quads, branch-free: out-of-image columns read a clamped address and
    // are zeroed by a select where the values are consumed (a select here would make the wave wait for its own prefetch)
    auto load_tile = [&](const float* __restrict__ F, int b, int h, int c0, float (&dst)[QPW][4][KK], bool& ok) {
        const int col = c0 + xl;
        ok = col >= 0 && col < W;
        const int cc = ok ? col : 0;
#pragma unroll
        for (int j = 0; j < QPW; ++j) {
            const int q = wave + j * NCW;
            if (q < GQ) {                                          // wave-uniform
                const unsigned p = tile_off(b, h, cc, q);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) dst[j][g][kk] = (F + (size_t)(g * CPG + 4 * kk) * HW)[p];
            }
        }
    };
    // both views of macro-unit m -> nA / nR and, if dma, of macro-unit m + 1 -> the wave's LDS slot: every channel row's
    // two 64-byte pieces are requested back to back
    auto load_next = [&](int m, bool dma) {
        int b0, h0, t0, b1, h1, t1;
        decode(m, b0, h0, t0);
        decode(dma ? m + 1 : m, b1, h1, t1);
        const int col0 = t0 * CVM_T + xl, col1 = t1 * CVM_T + xl;
        const bool ok0 = col0 < W, ok1 = col1 < W;
#pragma unroll
        for (int j = 0; j < QPW; ++j) {
            const int q = wave + j * NCW;
            if (q < GQ) {
                const unsigned p0 = tile_off(b0, h0, ok0 ? col0 : 0, q), p1 = tile_off(b1, h1, ok1 ? col1 : 0, q);
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const float* __restrict__ F = v ? a.Rg : a.Lg;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int kk = 0; kk < KK; ++kk) {
                            const float* __restrict__ Fc = F + (size_t)(g * CPG + 4 * kk) * HW;    // (wave-uniform base)
                            (v ? nR : nA)[j][g][kk] = Fc[p0];
                            if (PF == 2) {
                                if (dma) STX_GLDS4(stx_uniform_ptr(Fc), p1, dslot + (((j * 2 + v) * 4 + g) * KK + kk) * 64);   // (wave-uniform branch)
                            }
                        }
                }
            }
        }
        n_ok = ok0;
        if (PF == 2 && dma) d_ok = ok1;
    };
    auto mask_tile = [&](float (&dst)[QPW][4][KK], const float (&src)[QPW][4][KK], bool ok) {
#pragma unroll
        for (int j = 0; j < QPW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) dst[j][g][kk] = ok ? src[j][g][kk] : 0.f;
    };
    // the slot's tile -> A / Rt[0] (the LDS-DMA of one macro-unit ago has long landed)
    auto take_slot = [&]() {
        STX_GLDS_WAIT();
#pragma unroll
        for (int j = 0; j < QPW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const float l = dslot[(((j * 2 + 0) * 4 + g) * KK + kk) * 64 + lane];
                    const float r = dslot[(((j * 2 + 1) * 4 + g) * KK + kk) * 64 + lane];
                    A[j][g][kk] = d_ok ? l : 0.f;
                    Rt[0][j][g][kk] = d_ok ? r : 0.f;
                }
    };
    // concat features of a macro-unit -> registers: columns [w0, w0+16) of Lc and [w0 - 16 nd, w0 + 16) of Rc
    auto load_tables = [&](int b, int h, int w0) {
#pragma unroll
        for (int i = 0; i < NCL; ++i) {
            const int idx = tid + i * NCTHR;
            const int c = (int)__umulhi((unsigned)idx, a.magic_tc), col = idx - c * TC;
            const bool left = col < CVM_T;
            const int x = left ? w0 + col : w0 - CVM_T * nd + (col - CVM_T);
            const bool ok = c < Cc && x >= 0 && x < W;
            const float* src = left ? a.Lc : a.Rc;
            ct[i] = src[((size_t)(b * Cc + (c < Cc ? c : 0)) * H + h) * W + (ok ? x : 0)];
            if (ok) ctok |= 1u << i; else ctok &= ~(1u << i);
        }
    };
    auto store_tables = [&](float* tab) {
#pragma unroll
        for (int i = 0; i < NCL; ++i) {
            const int idx = tid + i * NCTHR;
            const int c = (int)__umulhi((unsigned)idx, a.magic_tc), col = idx - c * TC;
            if (c < Cc) tab[col * CS + c] = ((ctok >> i) & 1u) ? ct[i] : 0.f;
        }
    };

    // a window's first macro-unit: ring and tiles requested (registers only: issued BEFORE the window barrier, so the loads are in
    // flight while the store waves flush the previous window's last image)
    auto prologue = [&]() {
        if (m0 < m1) {
            int b, h, t;
            decode(m0, b, h, t);
            const int w0 = t * CVM_T;
            if (G) {
                // the ring of the first macro-unit: tiles t-1 .. t-ND straight into Rt[1..ND] (zeros left of the image)
#pragma unroll
                for (int j = 1; j <= ND; ++j) {
                    bool ok;
                    load_tile(a.Rg, b, h, w0 - CVM_T * j, Rt[j], ok);
                    mask_tile(Rt[j], Rt[j], ok);
                }
                // the first macro-unit's own tile (registers) and, PF = 2 with an even index, its partner's (slot)
                load_next(m0, PF == 2 && (m0 & 1) == 0 && m0 + 1 < m1);
            }
            if (Cc) load_tables(b, h, w0);
        }
    };

    int ui = 0;
    // one macro-unit; from_slot (wave-uniform) = its own tiles arrive in the wave's LDS slot instead of nA / nR.  (ONE
    // instantiation of this body inside the loop: with one per source behind an if / else hipcc spilled 94 registers.)
    auto macro = [&](int m, bool from_slot) {
        int b, h, t;
        decode(m, b, h, t);
        if (G) {
            if (m > m0) {
                // rotate the ring; a new image row (t = 0) starts with every older tile left of the image
#pragma unroll
                for (int j = ND; j >= 1; --j) mask_tile(Rt[j], Rt[j - 1], t != 0);
            }
            if (PF == 2 && from_slot) {
                take_slot();
            } else {
                mask_tile(A, nA, n_ok);
                mask_tile(Rt[0], nR, n_ok);
            }
        }
        if (Cc) store_tables(tabs + ((m - m0) & 1) * TAB);   // free: the store waves finished macro m-2 before barrier ui-1
        if (m + 1 < m1) {                                    // tiles of the next macro-unit(s): in flight during the units of this one
            int nb, nh, ntl;
            decode(m + 1, nb, nh, ntl);
            if (G) {
                if (PF != 2) load_next(m + 1, false);
                else if (m & 1) load_next(m + 1, m + 2 < m1);      // registers and slot are free now: the next pair (even, odd)
            }
            if (Cc) load_tables(nb, nh, ntl * CVM_T);
        }
        const int kb = m == m0 ? kfirst : 0, ke = m == m1 - 1 ? klast : nd;
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            if (k >= kb && k < ke) {
                float* stage = lds + (ui & 1) * IMG;           // free: the store waves passed barrier ui-1 after flushing it
                if (G) {
#pragma unroll
                    for (int j = 0; j < QPW; ++j) {
                        const int q = wave + j * NCW;
                        if (q < GQ) {
                            // two groups at a time: 16 accumulator registers live instead of 32 (with four waves per SIMD the
                            // kernel has 128 VGPRs; the second prefetch slot of PF = 2 needs the room)
#pragma unroll
                            for (int gh = 0; gh < 2; ++gh) {
                                f32x4 acc0[2], acc1[2];
#pragma unroll
                                for (int g = 0; g < 2; ++g) { acc0[g] = cvm_zero4(); acc1[g] = cvm_zero4(); }
#pragma unroll
                                for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                                    for (int g = 0; g < 2; ++g) {
                                        acc0[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j][2 * gh + g][kk], Rt[k][j][2 * gh + g][kk], acc0[g], 0, 0, 0);
                                        acc1[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j][2 * gh + g][kk], Rt[k + 1][j][2 * gh + g][kk], acc1[g], 0, 0, 0);
                                    }
                                // lane holds C[w_l = 4 kq + r][x_l = xl] of both tiles: d - 16k = w_l - x_l (tile t-k) or + 16 (tile t-k-1)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int wl = 4 * kq + r;
                                    const int dd = (wl - xl) & (CVM_T - 1);
                                    const bool t0 = wl >= xl;
                                    float2 v;
                                    v.x = (t0 ? acc0[0][r] : acc1[0][r]) * inv;
                                    v.y = (t0 ? acc0[1][r] : acc1[1][r]) * inv;
                                    *reinterpret_cast<float2*>(stage + dd * DS + wl * VS + 4 * q + 2 * gh) = v;
                                }
                            }
                        }
                    }
                }
                if (G || k == kb) STX_BARRIER_LDS();                 // image ui & 1 (G = 0: the macro-unit's tables) handed to the store waves
                ++ui;
            }
        }
    };
    // the run's first macro-unit always arrives in registers; after it (PF = 2) even ones do, odd ones come from the slot
    for (int r = 0; r < nwin; ++r) {
        set_window(r);
        prologue();
        if (r) STX_BARRIER_LDS();                                    // (matches the store waves' window barrier)
        ui = 0;
        for (int m = m0; m < m1; ++m) macro(m, PF == 2 && m != m0 && (m & 1));
    }
}

constexpr int CVM_PF_DEFAULT = 1;      // one tile ahead.  STX_CV_PF = 2 selects the line-pair scheme: 11 % fewer bytes fetched (FETCH_SIZE
                                       // 60.3 vs 67.8 MB x 2) but 5-6 % MORE time, alone and inside the train step (GPU calls B / C of round 4:
                                       // 0.1071 vs 0.1007 ms) -- the builder is bound by its write stream, not by the feature reads

template <int CPG, int QPW, int NCW, int NSW, int ND, bool SCALE = false, int PF = 1>
int cvm_launch(const CvmArgs& a, int wgs_per_cu, size_t lds, hipStream_t st) {
    auto kern = cost_volume_fwd_mfma_kernel<CPG, QPW, NCW, NSW, ND, SCALE, PF>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return stx_set_error(STX_ERR_LAUNCH, "cost_volume_fwd: %d bytes of dynamic LDS refused by this device", (int)lds);
    int grid = 256 * wgs_per_cu;
    if (stx_tune(STX_TUNE_CV_GRID) > 0) grid = stx_tune(STX_TUNE_CV_GRID);            // tests: force multi-unit runs
    if (grid > a.macros) grid = a.macros;
    hipLaunchKernelGGL(kern, dim3(grid), dim3((NCW + NSW) * 64), lds, st, a);
    return stx_check_launch("cost_volume_fwd(mfma)");
}

}  // namespace

// Returns -1 when the configuration is not served by this kernel (caller falls back to cost_volume.hip).
int stx_cv_fwd_mfma(const float* Lg, const float* Rg, int Cg, int G, const float* Lc, const float* Rc, int Cc,
                    const float* scale, float* vol, int B, int H, int W, int D, int mask_left, void* stream) {
    if (stx_tune(STX_TUNE_CV_OLD)) return -1;
    const int cpg = G ? Cg / G : 8;
    const bool wide = cpg == 20 || cpg == 28;             // FoundationStereo's 160 / 224 channels in 8 groups (vitb / vitl)
    if (!(cpg == 4 || cpg == 8 || cpg == 12 || cpg == 16 || wide) || Cc > CVM_MAXCC || (G & 3) || (Cc & 3)) return -1;
    if (wide && (G < 4 || G > 16)) return -1;
    const int CT = G + 2 * Cc, Q = CT / 4, GQ = G / 4;
    if (scale && G) return -1;                            // the attention scale comes with concat-only volumes (acv.py:196)
    if (Q < 1 || Q > 16) return -1;                       // voxels of <= 64 channels (store-wave slot table)
    const int nd = stx_cdiv(D, CVM_T);
    if (nd > 6) return -1;
    const int VS = G + 4, DS = CVM_T * VS + 4, CS = cvm_cs(Cc), TC = CVM_T * (nd + 2);
    size_t lds = 2 * ((size_t)(G ? CVM_T * DS : 0) + (size_t)(Cc ? TC * CS : 0)) * 4;      // double-buffered image + tables
    if (lds > 160 * 1024) return -1;
    CvmArgs a;
    a.Lg = Lg; a.Rg = Rg; a.Lc = Lc; a.Rc = Rc; a.scale = scale; a.vol = vol;
    a.B = B; a.H = H; a.W = W; a.D = D; a.G = G; a.Cc = Cc; a.mask_left = mask_left;
    a.nd = nd; a.nt = stx_cdiv(W, CVM_T);
    const long long macros = (long long)B * H * a.nt;
    if (macros >= (1ll << 31) || (long long)B * (Cg > Cc ? Cg : Cc) * H * W >= (1ll << 30)) return -1;
    if (16ll * H * W * CT >= (1ll << 32)) return -1;      // 32-bit offsets inside a unit's 16 d-planes
    a.macros = (int)macros;
    a.magic_rowq = (unsigned)(0x100000000ULL / (unsigned)(CVM_T * Q) + 1);
    a.magic_q = (unsigned)(0x100000000ULL / (unsigned)Q + 1);
    a.magic_tc = (unsigned)(0x100000000ULL / (unsigned)TC + 1);
    // non-temporal stores keep the streamed volume from evicting the (small, re-read) feature rows from L2:
    // measured 0.173 -> 0.141 ms on the GwcNet_GC build
    a.nontemporal = 1;
    a.by_units = stx_tune(STX_TUNE_CV_UNITS) != 0;
    a.win = stx_tune(STX_TUNE_CV_WIN) > 0 && stx_tune(STX_TUNE_CV_WIN) < a.macros ? stx_tune(STX_TUNE_CV_WIN) : a.macros;
    int pf = stx_tune(STX_TUNE_CV_PF) == 1 || stx_tune(STX_TUNE_CV_PF) == 2 ? stx_tune(STX_TUNE_CV_PF) : CVM_PF_DEFAULT;
    if (wide || a.win < a.macros) pf = 1;                 // (the line-pair scheme's even / odd slot parity is per run: windows take the default)
    // PF = 2 keeps one tile per compute wave in an LDS slot (2 KiB per quad and 4 channels of a group): taken when it fits beside the images
    const int ncw = GQ <= 4 ? 4 : (GQ <= 10 ? 10 : 8), qpw = GQ <= 10 ? 1 : 2;
    const size_t slots = (size_t)ncw * qpw * 8 * (cpg / 4) * 64 * 4;
    if (pf == 2 && GQ > 0 && lds + slots <= 160 * 1024) lds += slots; else pf = 1;
    // workgroups per CU: what the LDS images admit, at most 2; the wave layouts below are sized for <= 16 waves per workgroup
    int wgs = (int)((160 * 1024) / (lds + 1024));
    wgs = wgs < 1 ? 1 : (wgs > 2 ? 2 : wgs);
    hipStream_t st = (hipStream_t)stream;
    // wave layouts <channels per group, quads per compute wave, compute waves, store waves, ring bound> (measured in round 2:
    // fewer store waves are much slower -- 4 instead of 6: 0.21 vs 0.104 ms; two quads per compute wave: no gain)
#define CVM_GWC(CPG_, ND_, QPW_, NCW_, NSW_)                                                                 \
    {                                                                                                        \
        if (pf == 2) return cvm_launch<CPG_, QPW_, NCW_, NSW_, ND_, false, 2>(a, wgs, lds, st);              \
        return cvm_launch<CPG_, QPW_, NCW_, NSW_, ND_, false, 1>(a, wgs, lds, st);                           \
    }
#define CVM_LAYOUT(CPG_, ND_)                                                                        \
    {                                                                                                \
        if (GQ == 0 && scale) return cvm_launch<CPG_, 1, 4, 8, ND_, true>(a, wgs, lds, st);          \
        if (GQ == 0) return cvm_launch<CPG_, 1, 4, 8, ND_>(a, wgs, lds, st);                         \
        if (GQ <= 4) CVM_GWC(CPG_, ND_, 1, 4, 4)                                                     \
        if (GQ <= 10) CVM_GWC(CPG_, ND_, 1, 10, 6)                                                   \
        if (GQ <= 16) CVM_GWC(CPG_, ND_, 2, 8, 8)                                                    \
        return -1;                                                                                   \
    }
#define CVM_CASE(CPG_)                                          \
    if (cpg == CPG_) {                                          \
        if (nd <= 3) CVM_LAYOUT(CPG_, 3) else CVM_LAYOUT(CPG_, 6) \
    }
    CVM_CASE(4) CVM_CASE(8) CVM_CASE(12) CVM_CASE(16)
    // 20 / 28 channels per group: 4 .. 16 groups (one compute wave per group quad), the one-tile-ahead prefetch only
#define CVM_WIDE(CPG_)                                                                             \
    if (cpg == CPG_) {                                                                             \
        if (nd <= 3) return cvm_launch<CPG_, 1, 4, 4, 3, false, 1>(a, wgs, lds, st);               \
        return cvm_launch<CPG_, 1, 4, 4, 6, false, 1>(a, wgs, lds, st);                            \
    }
    CVM_WIDE(20) CVM_WIDE(28)
#undef CVM_WIDE
#undef CVM_CASE
#undef CVM_LAYOUT
#undef CVM_GWC
    return -1;
}
