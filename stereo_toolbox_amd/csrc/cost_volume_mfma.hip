// Cost-volume builder, second generation: group-wise correlation on the matrix cores, whole voxels staged in LDS,
// the volume written as one uninterrupted stream of 1-KiB wave stores.
//
// Replaces the same reference functions as cost_volume.hip (build_gwc_volume GwcNet/submodule.py:53-63,
// build_concat_volume GwcNet/submodule.py:30-41 / PSMNet/stackhourglass.py:111-120 / ACVNet/submodule.py:180-191,
// torch.cat gwcnet.py:180, softmax(att) * concat_volume acv.py:196) for every group configuration with 4, 8, 12 or
// 16 channels per group (GwcNet / ACVNet: 320 ch in 40 groups; IGEV-style volumes: 96 ch in 8 groups) and for
// concat-only volumes (PSMNet).
//
// Why.  The builder is HBM-bound on paper (516 MB per GwcNet_GC pair, 1 GFLOP), but the first-generation kernels were
// VALU/LDS-latency-bound: one lane per voxel channel = ~25 instructions and 3 LDS reads per 256-byte voxel
// (0.156 ms = 41 % of the HBM roofline).  Here the correlation of one image row is a banded batch of tiny GEMMs:
//   for group g:  C_g[w][x] = sum_{c in g} L[c][w] * R[c][x],   vol[d = w - x][w][g] = C_g[w][x] / cpg,  0 <= w - x < D'
// A work UNIT is (b, h, 16 disparities d0.., 16 columns w0..).  Its x-range [w0 - d0 - 15, w0 - d0 + 15] lies in two
// aligned 16-column tiles of R, so a wave computes, per group, two 16x16 tiles with v_mfma_f32_16x16x4_f32 (K = the
// group's channels, exact fp32, k-ordered fmaf chain) and every lane keeps, of the two results it holds for a (w, x)
// pair, the one whose d = w - x falls into the unit (lane (x, w): tile 0 if w_l >= x_l else tile 1).  The MFMA A/B
// operands are ONE dword per lane, loaded straight from the NCHW features (64-byte row segments, L2-resident: every
// feature row is re-read by the 3 d-chunks x 2 tiles that need it) -- no transposing LDS image of the features at all.
// A wave owns QPW channel quads (4 groups each); the right tile of unit t is the left tile... of nothing: the second
// R tile of unit (.., t) is the first R tile of unit (.., t-1), so walking t keeps it in registers (8 new loads for
// L, 8 for R per quad and unit, issued one unit ahead).
//
// Data path of a unit:  MFMA -> select -> ds_write_b128 into an LDS image [16 d][16 w][G] of the gwc channels
// (conflict-free: voxel stride G+4, d-row stride 16(G+4)+4 dwords) | concat features -> small LDS tables
// [col][Cc] -> barrier -> FLUSH: all waves walk the unit's 16 x 16 voxels in memory order, one float4 per lane
// (gwc quads from the image, left quads from the left table (masked), right quads from the right table at x = w - d;
// optional attention scale), so a wave stores 1 KiB contiguous and a d-row of the unit is one 4-KiB run -> barrier.
// Invalid entries (w < d, i.e. x < 0) are zero by construction: features left of the image are loaded as zeros.
// Several workgroups per CU (48 KB of LDS each) run out of phase, so the store stream never pauses for the MFMA /
// staging part of a unit.  Units are dealt in equal contiguous runs to exactly gridDim.x workgroups (no tail round).
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
    int nd, nt, units;
    unsigned magic_rowq, magic_q;   // ceil(2^32 / (16 Q)), ceil(2^32 / Q)
    int nontemporal;
};

__device__ __forceinline__ int cvm_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

__device__ __forceinline__ f32x4 cvm_zero4() {
    f32x4 z;
    z[0] = 0.f; z[1] = 0.f; z[2] = 0.f; z[3] = 0.f;
    return z;
}

// concat-table row stride (dwords): a multiple of 4 (float4 reads) whose quarter is odd, so that the transposing
// dword writes of consecutive columns spread over 8 bank groups
__host__ __device__ inline int cvm_cs(int Cc) { return ((Cc / 4) & 1) ? Cc + 8 : Cc + 4; }

template <int CPG, int QPW, int NCW, int NSW>
__global__ __launch_bounds__((NCW + NSW) * 64) void cost_volume_fwd_mfma_kernel(CvmArgs a) {
    constexpr int NCTHR = NCW * 64, NSTHR = NSW * 64;
    constexpr int KK = CPG / 4;                                   // MFMA K steps per group
    constexpr int NCL = (CVM_MAXCC * 48 + NCTHR - 1) / NCTHR;     // concat table elements per compute thread
    STX_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, D = a.D, G = a.G, Cc = a.Cc;
    const int HW = H * W, Cg = G * CPG, CT = G + 2 * Cc;
    const int GQ = G >> 2, CQ = Cc >> 2, Q = CT >> 2;
    const int VS = G + 4, DS = CVM_T * VS + 4, CS = cvm_cs(Cc);
    // two LDS images (double buffer): [16 dd][DS] gwc voxels | [16 cols][CS] left table | [32 cols][CS] right table
    const int IMG = G ? CVM_T * DS : 0, BUF = IMG + (Cc ? 48 * CS : 0);
    float* lds = reinterpret_cast<float*>(smem);

    const long long wg = cvm_xcd_remap(blockIdx.x, gridDim.x);
    const int u0 = __builtin_amdgcn_readfirstlane((int)((long long)a.units * wg / gridDim.x));
    const int u1 = __builtin_amdgcn_readfirstlane((int)((long long)a.units * (wg + 1) / gridDim.x));
    auto decode = [&](int u, int& b, int& h, int& k, int& t) {
        t = u % a.nt;
        int r = u / a.nt;
        k = r % a.nd;
        r /= a.nd;
        h = r % H;
        b = r / H;
    };

    if (wave >= NCW) {
        // ================= store waves: flush the image of unit i while the compute waves build unit i+1.
        // They never wait on a memory counter: the store stream of a CU is bounded by the hardware queues only.
        const int stid = tid - NCTHR;
        const int rowq = CVM_T * Q, total = CVM_T * rowq;
        const size_t dstride = (size_t)HW * CT;
        for (int u = u0; u < u1; ++u) {
            __syncthreads();                                           // image (u - u0) & 1 is complete
            int b, h, k, t;
            decode(u, b, h, k, t);
            const int w0 = t * CVM_T, d0 = k * CVM_T;
            const float* stage = lds + ((u - u0) & 1) * BUF;
            const int lc_off = IMG, rc_off = IMG + (Cc ? CVM_T * CS : 0);
            float* vrow = a.vol + (((size_t)b * D + d0) * H + h) * (size_t)W * CT + (size_t)w0 * CT;
            const float* srow = a.scale ? a.scale + (((size_t)b * D + d0) * H + h) * (size_t)W + w0 : nullptr;
            for (int base = 0; base < total; base += 4 * NSTHR) {
                float4 v[4];
                int off[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int idx = base + stid + j * NSTHR;
                    const int dd = (int)__umulhi((unsigned)idx, a.magic_rowq);
                    const int rem = idx - dd * rowq;
                    const int wl = (Q == 1) ? rem : (int)__umulhi((unsigned)rem, a.magic_q);   // (2^32 / 1 does not fit the magic)
                    const int q = rem - wl * Q;
                    const bool ok = idx < total && d0 + dd < D && w0 + wl < W;
                    // LDS source of this quad: gwc image | left table | right table at x = w - d
                    int src = dd * DS + wl * VS + 4 * q;
                    const int sl = lc_off + wl * CS + 4 * (q - GQ);
                    const int sr = rc_off + (wl - dd + CVM_T) * CS + 4 * (q - GQ - CQ);
                    src = q < GQ ? src : (q < GQ + CQ ? sl : sr);
                    float4 tv = stx_ld4(stage + (ok ? src : 0));
                    if (a.mask_left && q >= GQ && q < GQ + CQ && w0 + wl < d0 + dd) tv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (srow) {
                        const float m = srow[(size_t)(ok ? dd : 0) * HW + (ok ? wl : 0)];
                        tv.x *= m; tv.y *= m; tv.z *= m; tv.w *= m;
                    }
                    v[j] = tv;
                    off[j] = ok ? dd * 65536 + rem : -1;             // (dd < 16, rem < 16 Q <= 1024)
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (off[j] >= 0) {
                        float* dst = vrow + (size_t)(off[j] >> 16) * dstride + 4 * (off[j] & 0xffff);
                        if (a.nontemporal) stx_st4_nt(dst, v[j]);
                        else stx_st4(dst, v[j]);
                    }
                }
            }
        }
        return;
    }

    // ================= compute waves: feature tiles -> MFMA -> LDS image; concat features -> LDS tables
    const int xl = lane & 15, kq = lane >> 4;
    const float inv = 1.0f / (float)CPG;
    // operands of my quads: A = left tile, B0 = right tile of the unit's first x-tile, B1 = of the second one
    float A[QPW][4][KK], B0[QPW][4][KK], B1[QPW][4][KK], nA[QPW][4][KK], nB0[QPW][4][KK];
    float ct[NCL];
    unsigned ctok = 0;                                             // validity bits of ct[]
    bool n_okw = false, n_okx = false, okx1 = false;               // validity of the prefetched / the rotated tiles' columns
    // loads of one 16-column tile of both gwc features for my quads: column base cw (left), cx (right).  Branch-free:
    // out-of-image columns read a clamped address and are zeroed by a select where the values are consumed (applied
    // here it would make the wave wait for its own prefetch right away)
    auto load_tiles = [&](int b, int h, int cw, int cx, float (&dA)[QPW][4][KK], float (&dB)[QPW][4][KK], bool wantA,
                          bool& okw, bool& okx) {
        const int colw = cw + xl, colx = cx + xl;
        if (wantA) okw = colw < W;
        okx = colx >= 0 && colx < W;
        const int cwc = colw < W ? colw : W - 1, cxc = okx ? colx : 0;
#pragma unroll
        for (int j = 0; j < QPW; ++j) {
            const int q = wave + j * NCW;
            if (q < GQ) {                                          // wave-uniform
                const size_t base = ((size_t)(b * Cg + 4 * q * CPG + kq) * H + h) * W;
                const float* pl = a.Lg + base + cwc;
                const float* pr = a.Rg + base + cxc;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) {
                        const size_t o = (size_t)(g * CPG + 4 * kk) * HW;
                        if (wantA) dA[j][g][kk] = pl[o];
                        dB[j][g][kk] = pr[o];
                    }
            }
        }
    };
    auto load_tables = [&](int b, int h, int w0, int xbase) {        // concat features of the unit -> registers
#pragma unroll
        for (int i = 0; i < NCL; ++i) {
            const int idx = tid + i * NCTHR;
            const int c = idx / 48, col = idx - c * 48;
            const bool left = col < CVM_T;
            const int x = left ? w0 + col : xbase + (col - CVM_T);
            const bool ok = c < Cc && x >= 0 && x < W;
            const float* src = left ? a.Lc : a.Rc;
            ct[i] = src[((size_t)(b * Cc + (c < Cc ? c : 0)) * H + h) * W + (ok ? x : 0)];
            if (ok) ctok |= 1u << i; else ctok &= ~(1u << i);
        }
    };
    auto store_tables = [&](float* img) {
        float* Lc_s = img + IMG;
        float* Rc_s = Lc_s + CVM_T * CS;
#pragma unroll
        for (int i = 0; i < NCL; ++i) {
            const int idx = tid + i * NCTHR;
            const int c = idx / 48, col = idx - c * 48;
            const float v = ((ctok >> i) & 1u) ? ct[i] : 0.f;
            if (c < Cc) {
                if (col < CVM_T) Lc_s[col * CS + c] = v;
                else Rc_s[(col - CVM_T) * CS + c] = v;
            }
        }
    };

    if (u0 < u1) {
        int b, h, k, t;
        decode(u0, b, h, k, t);
        const int w0 = t * CVM_T, x0 = w0 - k * CVM_T;
        if (G) {
            load_tiles(b, h, w0, x0 - CVM_T, nA, B0, false, n_okw, okx1);   // second x-tile of the first unit -> (future) B1
#pragma unroll
            for (int j = 0; j < QPW; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) B0[j][g][kk] = okx1 ? B0[j][g][kk] : 0.f;
            load_tiles(b, h, w0, x0, nA, nB0, true, n_okw, n_okx);
        }
        if (Cc) load_tables(b, h, w0, x0 - CVM_T);
    }

    for (int u = u0; u < u1; ++u) {
        int b, h, k, t;
        decode(u, b, h, k, t);
        float* stage = lds + ((u - u0) & 1) * BUF;     // free: the store waves passed barrier (u - u0 - 1) after flushing it
        if (G) {
            // rotate: the previous unit's first x-tile is this unit's second one (same row and chunk, t-1);
            // at t = 0 the second tile lies left of the image: zeros
            const bool fresh = (u == u0);
#pragma unroll
            for (int j = 0; j < QPW; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) {
                        B1[j][g][kk] = (t == 0 && !fresh) ? 0.f : B0[j][g][kk];
                        A[j][g][kk] = n_okw ? nA[j][g][kk] : 0.f;
                        B0[j][g][kk] = n_okx ? nB0[j][g][kk] : 0.f;
                    }
#pragma unroll
            for (int j = 0; j < QPW; ++j) {
                const int q = wave + j * NCW;
                if (q < GQ) {
                    f32x4 acc0[4], acc1[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) { acc0[g] = cvm_zero4(); acc1[g] = cvm_zero4(); }
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            acc0[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j][g][kk], B0[j][g][kk], acc0[g], 0, 0, 0);
                            acc1[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j][g][kk], B1[j][g][kk], acc1[g], 0, 0, 0);
                        }
                    // lane holds C[w_l = 4 kq + r][x_l = xl] of both tiles: d - d0 = w_l - x_l (tile 0) or + 16 (tile 1)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int wl = 4 * kq + r;
                        const int dd = (wl - xl) & (CVM_T - 1);
                        const bool t0 = wl >= xl;
                        float4 v;
                        v.x = (t0 ? acc0[0][r] : acc1[0][r]) * inv;
                        v.y = (t0 ? acc0[1][r] : acc1[1][r]) * inv;
                        v.z = (t0 ? acc0[2][r] : acc1[2][r]) * inv;
                        v.w = (t0 ? acc0[3][r] : acc1[3][r]) * inv;
                        stx_st4(stage + dd * DS + wl * VS + 4 * q, v);
                    }
                }
            }
        }
        if (Cc) store_tables(stage);
        // operands of the next unit: in flight while the store waves flush this one (these waves issue loads only, so
        // their memory counter never waits for a store)
        if (u + 1 < u1) {
            int nb, nh, nk, ntl;
            decode(u + 1, nb, nh, nk, ntl);
            const int nw0 = ntl * CVM_T, nx0 = nw0 - nk * CVM_T;
            if (G) load_tiles(nb, nh, nw0, nx0, nA, nB0, true, n_okw, n_okx);
            if (Cc) load_tables(nb, nh, nw0, nx0 - CVM_T);
        }
        __syncthreads();                                               // image (u - u0) & 1 handed to the store waves
    }
}

template <int CPG, int QPW, int NCW, int NSW>
int cvm_launch(const CvmArgs& a, int wgs_per_cu, size_t lds, hipStream_t st) {
    auto kern = cost_volume_fwd_mfma_kernel<CPG, QPW, NCW, NSW>;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int grid = 256 * wgs_per_cu;
    if (const char* e = getenv("STX_CV_GRID")) grid = atoi(e) > 0 ? atoi(e) : grid;   // tests: force multi-unit runs
    if (grid > a.units) grid = a.units;
    hipLaunchKernelGGL(kern, dim3(grid), dim3((NCW + NSW) * 64), lds, st, a);
    return stx_check_launch("cost_volume_fwd(mfma)");
}

}  // namespace

// Returns -1 when the configuration is not served by this kernel (caller falls back to cost_volume.hip).
int stx_cv_fwd_mfma(const float* Lg, const float* Rg, int Cg, int G, const float* Lc, const float* Rc, int Cc,
                    const float* scale, float* vol, int B, int H, int W, int D, int mask_left, void* stream) {
    static const int off = getenv("STX_CV_OLD") ? 1 : 0;
    if (off) return -1;
    const int cpg = G ? Cg / G : 8;
    if (!(cpg == 4 || cpg == 8 || cpg == 12 || cpg == 16) || Cc > CVM_MAXCC || (G & 3) || (Cc & 3)) return -1;
    const int CT = G + 2 * Cc, Q = CT / 4, GQ = G / 4;
    if (Q < 1 || Q > 64) return -1;
    const int VS = G + 4, DS = CVM_T * VS + 4, CS = cvm_cs(Cc);
    const size_t lds = 2 * ((size_t)(G ? CVM_T * DS : 0) + (size_t)(Cc ? 48 * CS : 0)) * 4;      // double-buffered image
    if (lds > 160 * 1024) return -1;
    CvmArgs a;
    a.Lg = Lg; a.Rg = Rg; a.Lc = Lc; a.Rc = Rc; a.scale = scale; a.vol = vol;
    a.B = B; a.H = H; a.W = W; a.D = D; a.G = G; a.Cc = Cc; a.mask_left = mask_left;
    a.nd = stx_cdiv(D, CVM_T); a.nt = stx_cdiv(W, CVM_T);
    const long long units = (long long)B * H * a.nd * a.nt;
    if (units >= (1ll << 31) || (long long)B * (Cg > Cc ? Cg : Cc) * H * W >= (1ll << 31)) return -1;
    a.units = (int)units;
    a.magic_rowq = (unsigned)(0x100000000ULL / (unsigned)(CVM_T * Q) + 1);
    a.magic_q = (unsigned)(0x100000000ULL / (unsigned)Q + 1);
    static const int nt_env = getenv("STX_CV_NT") ? atoi(getenv("STX_CV_NT")) : 0;
    a.nontemporal = nt_env;
    // workgroups per CU: what the LDS images admit, at most 2 (tuning switch STX_CV_WGS); the wave layouts below are
    // sized for <= 16 waves per workgroup
    static const int wgs_env = getenv("STX_CV_WGS") ? atoi(getenv("STX_CV_WGS")) : 0;
    int wgs = (int)((160 * 1024) / (lds + 1024));
    wgs = wgs < 1 ? 1 : (wgs > 2 ? 2 : wgs);
    if (wgs_env > 0) wgs = wgs_env;
    const int qpw_env = getenv("STX_CV_QPW") ? atoi(getenv("STX_CV_QPW")) : 0;
    static const int nsw_env = getenv("STX_CV_NSW") ? atoi(getenv("STX_CV_NSW")) : 0;
    hipStream_t st = (hipStream_t)stream;
    // wave layouts <channels per group, quads per compute wave, compute waves, store waves>; STX_CV_QPW = 2 selects the
    // fat-wave layouts, STX_CV_NSW = 4 / 8 fewer / more store waves (tuning switches)
#define CVM_CASE(CPG_)                                                                               \
    if (cpg == CPG_) {                                                                               \
        if (GQ == 0) return nsw_env == 4 ? cvm_launch<CPG_, 1, 4, 4>(a, wgs, lds, st)                \
                          : (nsw_env == 12 ? cvm_launch<CPG_, 1, 4, 12>(a, wgs, lds, st)             \
                                           : cvm_launch<CPG_, 1, 4, 8>(a, wgs, lds, st));            \
        if (GQ <= 4 && qpw_env != 2) return cvm_launch<CPG_, 1, 4, 4>(a, wgs, lds, st);              \
        if (GQ <= 10 && qpw_env != 2)                                                                \
            return nsw_env == 4 ? cvm_launch<CPG_, 1, 10, 4>(a, wgs, lds, st)                        \
                                : cvm_launch<CPG_, 1, 10, 6>(a, wgs, lds, st);                       \
        if (GQ <= 10) return nsw_env == 4 ? cvm_launch<CPG_, 2, 5, 4>(a, wgs, lds, st)               \
                                          : cvm_launch<CPG_, 2, 5, 8>(a, wgs, lds, st);              \
        if (GQ <= 16) return cvm_launch<CPG_, 2, 8, 8>(a, wgs, lds, st);                             \
        return -1;                                                                                   \
    }
    CVM_CASE(4) CVM_CASE(8) CVM_CASE(12) CVM_CASE(16)
#undef CVM_CASE
    return -1;
}
