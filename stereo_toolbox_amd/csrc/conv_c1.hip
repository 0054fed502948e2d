// Single-output-channel 3x3x3 convolution (the classifier tails `Conv3d(32, 1, 3, padding=1)` of
// reference models/GwcNet/gwcnet.py:139-153, PSMNet/stackhourglass.py:74-84, ACVNet/acv.py:122-144)
// and its weight gradient, for gfx950.
//
// With N = 1 the layer is not GEMM-shaped in the usual (voxels x Cout) sense: putting it on the 32-wide MFMA
// tile wastes 31/32 of the matrix work ("N=1: vector-dot kernel, HBM-bound: reads 212 MB").
// The 27 taps, however, make a fine N dimension:
//   forward (Cin = 32)      : P[v][tap] = x[v][:] . w[:][tap] on the matrix cores, then a 27-point gather (march along d);
//   wgrad   (Cin % 32 == 0) : dW[c][tap] = sum_v x[v][c] * gy[v - tap + 1], M = channels, N = taps, K = voxels;
//   dgrad                   : streaming VALU kernel (reads 1 channel, writes Cin);
//   other channel counts    : the first-generation VALU kernels over an LDS halo tile (kept as fallback).
// Roofline: HBM (algorithmic bytes = read x once + write 1 channel, or the reverse for dgrad).
#include "stx_common.h"
#include <stdlib.h>

namespace {

constexpr int C1_THREADS = 256;
constexpr int C1_TD = 2, C1_TH = 4, C1_TW = 32;        // 256 output voxels per tile, one per thread
constexpr int C1_ED = C1_TD + 2, C1_EH = C1_TH + 2, C1_EW = C1_TW + 2;
constexpr int C1_CK = 16, C1_VS = C1_CK + 4;           // 16-channel chunks: 816 voxels x 80 B = 65 KB

struct C1Args {
    const float* x;      // [B][D][H][W][Cin]
    const float* w;      // fwd: [Cin*27] torch layout [1][Cin][27]; wgrad: unused
    const float* gy;     // wgrad: [B][D][H][W]
    const float* res;    // fwd: optional residual [B][D][H][W]
    float* out;          // fwd: [B][D][H][W]; wgrad: partial slab [nblk][Cin*27]
    int B, D, H, W, Cin;
    int nDt, nHt, nWt, ntiles;
};

__device__ __forceinline__ void c1_stage(const C1Args& a, float* tile, int b, int d0, int h0, int w0, int c0, int tid) {
    for (int idx = tid; idx < C1_ED * C1_EH * C1_EW * (C1_CK / 4); idx += C1_THREADS) {
        const int v = idx / (C1_CK / 4), f = idx - v * (C1_CK / 4);
        const int wx = v % C1_EW, hy = (v / C1_EW) % C1_EH, dz = v / (C1_EW * C1_EH);
        const int gd = d0 - 1 + dz, gh = h0 - 1 + hy, gw = w0 - 1 + wx;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W)
            val = stx_ld4(a.x + ((((size_t)b * a.D + gd) * a.H + gh) * a.W + gw) * a.Cin + c0 + 4 * f);
        stx_st4(tile + v * C1_VS + 4 * f, val);
    }
}

__global__ __launch_bounds__(C1_THREADS) void conv_c1_fwd_kernel(C1Args a) {
    STX_DYN_SMEM(smem);
    float* tile = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x;
    int r = blockIdx.x;
    const int wt = r % a.nWt; r /= a.nWt;
    const int ht = r % a.nHt; r /= a.nHt;
    const int dt = r % a.nDt;
    const int b = r / a.nDt;
    const int d0 = dt * C1_TD, h0 = ht * C1_TH, w0 = wt * C1_TW;
    const int lw = tid % C1_TW, lh = (tid / C1_TW) % C1_TH, ld = tid / (C1_TW * C1_TH);
    const int base = ((ld * C1_EH + lh) * C1_EW + lw) * C1_VS;
    float acc = 0.f;
    for (int c0 = 0; c0 < a.Cin; c0 += C1_CK) {
        __syncthreads();
        c1_stage(a, tile, b, d0, h0, w0, c0, tid);
        __syncthreads();
        for (int tap = 0; tap < 27; ++tap) {
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            const float* p = tile + base + ((kd * C1_EH + kh) * C1_EW + kw) * C1_VS;
#pragma unroll
            for (int f = 0; f < C1_CK / 4; ++f) {
                const float4 v = stx_ld4(p + 4 * f);
                const float* wp = a.w + (size_t)(c0 + 4 * f) * 27 + tap;    // wave-uniform -> scalar loads
                acc = fmaf(v.x, wp[0], acc);
                acc = fmaf(v.y, wp[27], acc);
                acc = fmaf(v.z, wp[54], acc);
                acc = fmaf(v.w, wp[81], acc);
            }
        }
    }
    const int od = d0 + ld, oh = h0 + lh, ow = w0 + lw;
    if (od < a.D && oh < a.H && ow < a.W) {
        const size_t o = (((size_t)b * a.D + od) * a.H + oh) * a.W + ow;
        a.out[o] = a.res ? acc + a.res[o] : acc;
    }
}

// dW[c][tap] partials: thread (c_local = tid & 15, tap group tg = tid >> 4 in 0..15) owns taps tg, tg+16
// of channel c0 + c_local for the current 16-channel chunk, and reduces over the tile's 256 voxels.
__global__ __launch_bounds__(C1_THREADS) void conv_c1_wgrad_kernel(C1Args a) {
    STX_DYN_SMEM(smem);
    float* tile = reinterpret_cast<float*>(smem);
    float* gys = tile + C1_ED * C1_EH * C1_EW * C1_VS;    // [256]
    const int tid = threadIdx.x;
    const int cl = tid & 15, tg = tid >> 4;
    const int nchunk = a.Cin / C1_CK;
    float acc[4][2];                                       // [chunk<=4][tap slot]
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i][0] = 0.f; acc[i][1] = 0.f; }
    const int tapA = tg, tapB = tg + 16;                   // tapB valid when < 27
    const int offA = (((tapA / 9) * C1_EH + (tapA / 3) % 3) * C1_EW + tapA % 3) * C1_VS + cl;
    const int tB = tapB < 27 ? tapB : 0;
    const int offB = (((tB / 9) * C1_EH + (tB / 3) % 3) * C1_EW + tB % 3) * C1_VS + cl;

    for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        int r = t;
        const int wt = r % a.nWt; r /= a.nWt;
        const int ht = r % a.nHt; r /= a.nHt;
        const int dt = r % a.nDt;
        const int b = r / a.nDt;
        const int d0 = dt * C1_TD, h0 = ht * C1_TH, w0 = wt * C1_TW;
#pragma unroll
        for (int ck = 0; ck < 4; ++ck) {
            if (ck < nchunk) {
                __syncthreads();
                c1_stage(a, tile, b, d0, h0, w0, ck * C1_CK, tid);
                if (ck == 0) {
                    const int lw = tid % C1_TW, lh = (tid / C1_TW) % C1_TH, ld = tid / (C1_TW * C1_TH);
                    const int od = d0 + ld, oh = h0 + lh, ow = w0 + lw;
                    gys[tid] = (od < a.D && oh < a.H && ow < a.W)
                                   ? a.gy[(((size_t)b * a.D + od) * a.H + oh) * a.W + ow] : 0.f;
                }
                __syncthreads();
                float sA = 0.f, sB = 0.f;
                for (int v = 0; v < C1_TD * C1_TH * C1_TW; ++v) {
                    const int lw = v % C1_TW, lh = (v / C1_TW) % C1_TH, ld = v / (C1_TW * C1_TH);
                    const int vb = ((ld * C1_EH + lh) * C1_EW + lw) * C1_VS;
                    const float g = gys[v];
                    sA = fmaf(tile[vb + offA], g, sA);
                    sB = fmaf(tile[vb + offB], g, sB);
                }
                acc[ck][0] += sA;
                acc[ck][1] += sB;
            }
        }
    }
    float* dst = a.out + (size_t)blockIdx.x * a.Cin * 27;
#pragma unroll
    for (int ck = 0; ck < 4; ++ck) {
        if (ck < nchunk) {
            dst[(ck * C1_CK + cl) * 27 + tapA] = acc[ck][0];
            if (tapB < 27) dst[(ck * C1_CK + cl) * 27 + tapB] = acc[ck][1];
        }
    }
}

// sums[m] = sum_r partial[r][m] (fp64 accumulate), one workgroup per column.
__global__ __launch_bounds__(C1_THREADS) void c1_colsum_kernel(const float* __restrict__ partials, int nrows, int M,
                                                               float* __restrict__ sums) {
    __shared__ double red[C1_THREADS];
    const int m = blockIdx.x, tid = threadIdx.x;
    double s = 0.0;
    for (int r = tid; r < nrows; r += C1_THREADS) s += (double)partials[(size_t)r * M + m];
    red[tid] = s;
    __syncthreads();
    for (int k = C1_THREADS / 2; k > 0; k >>= 1) {
        if (tid < k) red[tid] += red[tid + k];
        __syncthreads();
    }
    if (tid == 0) sums[m] = (float)red[0];
}

constexpr size_t C1_TILE_BYTES = (size_t)C1_ED * C1_EH * C1_EW * C1_VS * 4;
constexpr int C1_WGRAD_BLOCKS = 512;


// Data gradient of Conv3d(Cin, 1, 3, padding=1): gx[v][c] = sum_tap gy[v - (tap - 1)] * w[0][c][tap].
// Pure streaming: reads the 1-channel gy (B*D*H*W*4 B, L2-resident), writes Cin channels.  A workgroup owns
// 1 x 4 x 32 voxels; the 3 x 6 x 34 halo of gy and the weights ([tap][Cin]) sit in LDS; a work item is
// (4 consecutive voxels along w, one 4-channel quad): per tap one ds_read_b128 of weights and 4 gy scalars
// feed 16 FMAs; the Cin/4 quads of a voxel are consecutive lanes, so stores are 16 B x (Cin/4) contiguous.
constexpr int C1D_TH = 4, C1D_TW = 32;
constexpr int C1D_EH = C1D_TH + 2, C1D_EW = C1D_TW + 2;

__global__ __launch_bounds__(C1_THREADS) void conv_c1_dgrad_kernel(const float* __restrict__ gy,
                                                                   const float* __restrict__ w, float* __restrict__ gx,
                                                                   int B, int D, int H, int W, int Cin, int nHt, int nWt) {
    __shared__ float gys[3 * C1D_EH * C1D_EW];
    __shared__ float ws[27 * 64];
    const int tid = threadIdx.x;
    int r = blockIdx.x;
    const int wt = r % nWt; r /= nWt;
    const int ht = r % nHt; r /= nHt;
    const int d = r % D;
    const int b = r / D;
    const int h0 = ht * C1D_TH, w0 = wt * C1D_TW;
    for (int idx = tid; idx < 3 * C1D_EH * C1D_EW; idx += C1_THREADS) {
        const int wx = idx % C1D_EW, hy = (idx / C1D_EW) % C1D_EH, dz = idx / (C1D_EW * C1D_EH);
        const int gd = d - 1 + dz, gh = h0 - 1 + hy, gw = w0 - 1 + wx;
        gys[idx] = (gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W)
                       ? gy[(((size_t)b * D + gd) * H + gh) * W + gw] : 0.f;
    }
    for (int idx = tid; idx < 27 * Cin; idx += C1_THREADS) {       // ws[tap][c] = w[c][tap]
        const int c = idx % Cin, tap = idx / Cin;
        ws[tap * Cin + c] = w[(size_t)c * 27 + tap];
    }
    __syncthreads();
    const int CQ = Cin >> 2;
    for (int item = tid; item < CQ * (C1D_TH * C1D_TW / 4); item += C1_THREADS) {
        const int q = item % CQ, grp = item / CQ;
        const int lw = (grp % (C1D_TW / 4)) * 4, lh = grp / (C1D_TW / 4);
        float4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                // gy row of the halo for this (kd, kh): output voxel (d, h, w) meets gy(d + 1 - kd, h + 1 - kh, w + 1 - kw)
                const float* gp = gys + (((2 - kd) * C1D_EH) + (lh + 2 - kh)) * C1D_EW + lw;
                float g6[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) g6[j] = gp[j];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float4 wv = stx_ld4(ws + ((kd * 3 + kh) * 3 + kw) * Cin + 4 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float g = g6[j + 2 - kw];
                        acc[j].x = fmaf(g, wv.x, acc[j].x);
                        acc[j].y = fmaf(g, wv.y, acc[j].y);
                        acc[j].z = fmaf(g, wv.z, acc[j].z);
                        acc[j].w = fmaf(g, wv.w, acc[j].w);
                    }
                }
            }
        const int oh = h0 + lh;
        if (oh < H) {
            float* o = gx + ((((size_t)b * D + d) * H + oh) * W + w0 + lw) * Cin + 4 * q;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (w0 + lw + j < W) stx_st4(o + (size_t)j * Cin, acc[j]);
        }
    }
}


// Data gradient on the matrix cores (Cin = 32): gx[v][c] = sum_tap gy[v - (tap - 1)] * w[c][tap] is a GEMM with M = voxels,
// N = 32 channels, K = 27 taps padded to 28 = 14 x v_mfma_f32_32x32x2f32 per 32 voxels.  A = the 1-channel gy at the tap's
// shifted voxel (lane (voxel i, half) reads tap 2 j + half from the small LDS halo of gy: 14 fixed offsets per lane), B = the
// weights (14 registers per lane, loaded once).  The accumulator holds one channel of 16 voxels per lane, so every store
// instruction writes two whole 128-byte voxels; rows go through a descriptor of the output plane (voxels outside the volume get
// the out-of-range offset).  The VALU kernel above needs 27 x 32 FMAs per voxel on the vector pipe (0.109 ms = 0.25 of the HBM
// rate for the 212 MB it writes); here the matrix pipe does them in 896 cycles per 4 KiB of output and wave.
// A workgroup = 4 waves = 4 rows x 32 columns of one plane; tiles are dealt round-robin.
__global__ __launch_bounds__(C1_THREADS) void conv_c1_dgrad_mfma_kernel(const float* __restrict__ gy,
                                                                        const float* __restrict__ w, float* __restrict__ gx,
                                                                        int B, int D, int H, int W, int nHt, int nWt, int ntiles) {
    __shared__ float gys[3 * C1D_EH * C1D_EW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, half = lane >> 5;
    float wreg[14];
    int aoff[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const int tap = 2 * j + half;
        const int tp = tap < 27 ? tap : 26;
        const int kd = tp / 9, kh = (tp / 3) % 3, kw = tp % 3;
        wreg[j] = tap < 27 ? w[(size_t)i * 27 + tap] : 0.f;
        // output voxel (d, h, w) meets gy(d + 1 - kd, h + 1 - kh, w + 1 - kw); the halo starts at (d - 1, h0 - 1, w0 - 1)
        aoff[j] = ((2 - kd) * C1D_EH + (wave + 2 - kh)) * C1D_EW + i + 2 - kw;
    }
    const unsigned plane_bytes = (unsigned)H * (unsigned)W * 128u;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int r_ = tile;
        const int wt = r_ % nWt; r_ /= nWt;
        const int ht = r_ % nHt; r_ /= nHt;
        const int d = r_ % D;
        const int b = r_ / D;
        const int h0 = ht * C1D_TH, w0 = wt * C1D_TW;
        __syncthreads();                                       // the previous tile's halo has been read
        for (int idx = tid; idx < 3 * C1D_EH * C1D_EW; idx += C1_THREADS) {
            const int wx = idx % C1D_EW, hy = (idx / C1D_EW) % C1D_EH, dz = idx / (C1D_EW * C1D_EH);
            const int gd = d - 1 + dz, gh = h0 - 1 + hy, gw = w0 - 1 + wx;
            gys[idx] = (gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W)
                           ? gy[(((size_t)b * D + gd) * H + gh) * W + gw] : 0.f;
        }
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float av[14];
#pragma unroll
        for (int j = 0; j < 14; ++j) av[j] = gys[aoff[j]];
#pragma unroll
        for (int j = 0; j < 14; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], wreg[j], acc, 0, 0, 0);
        // acc[r] = gx[voxel (r & 3) + 8 (r >> 2) + 4 half of the row][channel i]
        const int oh = h0 + wave;
        const stx_bufrsrc ors = stx_make_rsrc(gx + ((size_t)b * D + d) * H * W * 32, plane_bytes);
        const unsigned vbase = (unsigned)((((oh < H ? oh : 0) * W + w0 + 4 * half) * 32 + i) * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2);
            const bool ok = oh < H && w0 + 4 * half + m < W;
            stx_buf_st1(ors, ok ? vbase : STX_BUF_OOB, (unsigned)(m * 128), acc[r]);
        }
    }
}


// Weight gradient on the matrix cores (Cin a multiple of 32).  Re-indexed over INPUT voxels v:
//   dW[c][tap] = sum_v x[v][c] * gy[v - (tap - 1)]          (gy = 0 outside the volume)
// is a GEMM with M = 32 channels, N = 27 taps (padded to 32), K = voxels.  v_mfma_f32_32x32x2f32 takes two
// voxels per instruction: lane (i, half) supplies A = x[v0 + half][i] -- a wave's A loads are 256 contiguous
// bytes straight from global memory, x is never staged -- and B = gy[v0 + half - off(tap = i)] from a small LDS
// halo tile of gy (the 27 tap offsets of a lane are fixed; they land in distinct banks).  x is read exactly
// once (the VALU kernel above re-reads its 3.2x halo through LDS for 27 x Cin FMAs per voxel: 0.39 ms).
// A workgroup (4 waves) walks 2 x 8 x 32-voxel tiles; a wave owns 4 rows of 32 voxels per tile and keeps one
// 32 x 32 accumulator; the waves meet in LDS at the end and write one slab row per workgroup.
constexpr int C1M_TD = 2, C1M_TH = 8, C1M_TW = 32;
constexpr int C1M_ED = C1M_TD + 2, C1M_EH = C1M_TH + 2, C1M_EW = C1M_TW + 2;
constexpr int C1M_BLOCKS = 1024;

__global__ __launch_bounds__(C1_THREADS) void conv_c1_wgrad_mfma_kernel(const float* __restrict__ x,
                                                                        const float* __restrict__ gy,
                                                                        float* __restrict__ slab, int B, int D, int H,
                                                                        int W, int Cin, int nDt, int nHt, int nWt,
                                                                        int ntiles) {
    __shared__ float gys[C1M_ED * C1M_EH * C1M_EW];
    __shared__ float red[4][1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, half = lane >> 5;
    const int cb = blockIdx.y;                                   // 32-channel block
    const int kd = i / 9, kh = (i / 3) % 3, kw = i % 3;          // my tap (i < 27)
    const int toff = (i < 27) ? ((2 - kd) * C1M_EH + (2 - kh)) * C1M_EW + (2 - kw) + half : half;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int r = t;
        const int wt = r % nWt; r /= nWt;
        const int ht = r % nHt; r /= nHt;
        const int dt = r % nDt;
        const int b = r / nDt;
        const int d0 = dt * C1M_TD, h0 = ht * C1M_TH, w0 = wt * C1M_TW;
        __syncthreads();                                         // previous tile's gy halo is no longer read
        for (int idx = tid; idx < C1M_ED * C1M_EH * C1M_EW; idx += C1_THREADS) {
            const int wx = idx % C1M_EW, hy = (idx / C1M_EW) % C1M_EH, dz = idx / (C1M_EW * C1M_EH);
            const int gd = d0 - 1 + dz, gh = h0 - 1 + hy, gw = w0 - 1 + wx;
            gys[idx] = (gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W)
                           ? gy[(((size_t)b * D + gd) * H + gh) * W + gw] : 0.f;
        }
        __syncthreads();
        // my 4 rows of the tile: row = wave * 4 + k -> (ld, lh)
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const int row = wave * 4 + k;
            const int ld = row / C1M_TH, lh = row % C1M_TH;
            const int gd = d0 + ld, gh = h0 + lh;
            const bool rowok = gd < D && gh < H;
            const float* xr = x + ((((size_t)b * D + gd) * H + gh) * W + w0 + half) * Cin + cb * 32 + i;
            float av[16];
#pragma unroll
            for (int p = 0; p < 16; ++p)
                av[p] = (rowok && w0 + 2 * p + half < W) ? xr[(size_t)(2 * p) * Cin] : 0.f;
            const float* gp = gys + (ld * C1M_EH + lh) * C1M_EW + toff;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                float bv = gp[2 * p];
                bv = (i < 27) ? bv : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p], bv, acc, 0, 0, 0);
            }
        }
    }
    // D layout: acc[r] = element (row c = (r & 3) + 8 (r >> 2) + 4 half, col tap = i)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + i] = acc[r];
    __syncthreads();
    float* dst = slab + ((size_t)cb * gridDim.x + blockIdx.x) * 1024;
    for (int idx = tid; idx < 1024; idx += C1_THREADS)
        dst[idx] = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
}

// dw[(cb*32 + c)*27 + tap] = sum over the workgroup rows of slab[cb][row][c][tap] (fp64, fixed order)
__global__ __launch_bounds__(C1_THREADS) void c1m_reduce_kernel(const float* __restrict__ slab, int nrows,
                                                                float* __restrict__ dw) {
    __shared__ double red[C1_THREADS];
    const int m = blockIdx.x, tid = threadIdx.x;                 // m = (cb*32 + c) * 27 + tap
    const int tap = m % 27, cfull = m / 27, cb = cfull >> 5, c = cfull & 31;
    const float* p = slab + (size_t)cb * nrows * 1024 + c * 32 + tap;
    double s = 0.0;
    for (int r = tid; r < nrows; r += C1_THREADS) s += (double)p[(size_t)r * 1024];
    red[tid] = s;
    __syncthreads();
    for (int k = C1_THREADS / 2; k > 0; k >>= 1) {
        if (tid < k) red[tid] += red[tid + k];
        __syncthreads();
    }
    if (tid == 0) dw[m] = (float)red[0];
}


// Forward of Conv3d(32, 1, 3, padding=1) on the matrix cores.  Per INPUT voxel v the 27 products
//   P[v][tap] = sum_c x[v][c] * w[c][tap]
// are a GEMM (M = voxels, N = 27 taps padded to 32, K = 32 channels = 16 x v_mfma_f32_32x32x2f32 per 32
// voxels) whose A operand comes straight from global memory (lane (i, half) loads the 64 contiguous bytes
// x[v_i][16 half .. +16]); the output is the 27-point gather out[o] = sum_tap P[o + tap - 1][tap].
// A workgroup owns an 8 x 32 column and marches along d: per input plane p it multiplies the 10 x 34 halo plane
// (11 M-tiles over 4 waves) into an LDS image Ps[voxel][27] (double-buffered: one barrier per plane), then every
// thread gathers the three 9-tap sums of its column; plane p finishes output p-1, continues p, starts p+1
// (two rolling registers).  x is read 1.33 x (halo) instead of 3.2 x through LDS with 27 x 32 FMAs per voxel
// on the VALU kernel above (0.30 ms).  The (column, plane) work list is split evenly over the grid.
constexpr int C1F_TH = 8, C1F_TW = 32;
constexpr int C1F_EH = C1F_TH + 2, C1F_EW = C1F_TW + 2, C1F_NV = C1F_EH * C1F_EW;      // 340 halo voxels
constexpr int C1F_MT = (C1F_NV + 31) / 32;                                             // 11 M-tiles
constexpr int C1F_WGS = 512;

__global__ __launch_bounds__(C1_THREADS) void conv_c1_fwd_mfma_kernel(const float* __restrict__ x,
                                                                      const float* __restrict__ w,
                                                                      const float* __restrict__ res,
                                                                      float* __restrict__ out, int B, int D, int H, int W,
                                                                      int nHt, int nWt, int ncols) {
    STX_DYN_SMEM(smem);
    float* Ps = reinterpret_cast<float*>(smem);                  // [2][C1F_NV][27]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, half = lane >> 5;
    // B operand: step s multiplies channel 16 half + s; lane column = tap i
    float wreg[16];
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) wreg[s_] = (i < 27) ? w[(size_t)(16 * half + s_) * 27 + i] : 0.f;
    const int lh = tid / C1F_TW, lw = tid % C1F_TW;              // my output column of the tile

    // A operand of the NEXT plane in flight while this plane is multiplied and gathered (the first version loaded, waited
    // and multiplied M-tile after M-tile: three exposed memory round trips per plane, 0.102 ms = 0.27 of the HBM rate).  The
    // loads go through a descriptor of the input plane: per-lane byte offsets once per column (halo voxels outside the
    // volume get the out-of-range offset and read zeros), planes outside [0, D) an empty descriptor.
    constexpr int MTW = (C1F_MT + 3) / 4;                          // M-tiles per wave (3, 3, 3, 2)
    const long long units = (long long)ncols * D;
    int u = __builtin_amdgcn_readfirstlane((int)(units * blockIdx.x / gridDim.x));
    const int u_end = __builtin_amdgcn_readfirstlane((int)(units * (blockIdx.x + 1) / gridDim.x));
    const unsigned plane_bytes = (unsigned)H * (unsigned)W * 32u * 4u;
    while (u < u_end) {
        const int col = u / D;
        const int d_lo = u - col * D;
        const int left = u_end - u;
        const int d_hi = (D - d_lo < left) ? D : d_lo + left;
        u += d_hi - d_lo;
        const int wt = col % nWt, ht = (col / nWt) % nHt, b = col / (nWt * nHt);
        const int h0 = ht * C1F_TH, w0 = wt * C1F_TW;
        const int oh = h0 + lh, ow = w0 + lw;
        const bool ook = oh < H && ow < W;
        unsigned voff[MTW];
#pragma unroll
        for (int t = 0; t < MTW; ++t) {
            const int hv = (wave + 4 * t) * 32 + i;
            const int hy = hv / C1F_EW, wx = hv - hy * C1F_EW;
            const int gh = h0 - 1 + hy, gw = w0 - 1 + wx;
            const bool ok = wave + 4 * t < C1F_MT && hv < C1F_NV && gh >= 0 && gh < H && gw >= 0 && gw < W;
            voff[t] = ok ? (unsigned)(((gh * W + gw) * 32 + 16 * half) * 4) : STX_BUF_OOB;
        }
        float4 nxt[MTW][4];
        auto load_plane = [&](int p) {
            const bool in = p >= 0 && p < D && p <= d_hi;
            const stx_bufrsrc rs = stx_make_rsrc(x + ((size_t)b * D + (in ? p : 0)) * H * W * 32, in ? plane_bytes : 0u);
#pragma unroll
            for (int t = 0; t < MTW; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) nxt[t][q] = stx_buf_ld4(rs, voff[t], 16u * q);
        };
        float accA = 0.f, accB = 0.f;
        load_plane(d_lo - 1);
        __syncthreads();                                          // previous run is done with Ps
        for (int p = d_lo - 1; p <= d_hi; ++p) {
            float* Pb = Ps + ((p + 2) & 1) * (C1F_NV * 27);
            const bool pin = p >= 0 && p < D;
            float4 cur[MTW][4];
#pragma unroll
            for (int t = 0; t < MTW; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) cur[t][q] = nxt[t][q];
            load_plane(p + 1);
            if (pin) {
#pragma unroll
                for (int t = 0; t < MTW; ++t) {
                    const int m = wave + 4 * t;
                    if (m < C1F_MT) {
                        f32x16 acc;
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t][q].x, wreg[4 * q + 0], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t][q].y, wreg[4 * q + 1], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t][q].z, wreg[4 * q + 2], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t][q].w, wreg[4 * q + 3], acc, 0, 0, 0);
                        }
                        // D: acc[r] = P[voxel (r & 3) + 8 (r >> 2) + 4 half of the M-tile][tap i]
                        if (i < 27) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int v = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                                if (v < C1F_NV) Pb[v * 27 + i] = acc[r];
                            }
                        }
                    }
                }
            }
            __syncthreads();                                      // P[p] complete (and P[p-1] fully gathered)
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            if (pin) {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float* pp = Pb + ((lh + kh) * C1F_EW + lw + kw) * 27 + kh * 3 + kw;
                        s0 += pp[0];
                        s1 += pp[9];
                        s2 += pp[18];
                    }
            }
            // input plane p: tap kd = 2 finishes output p-1, kd = 1 continues output p, kd = 0 starts output p+1
            const float done = accA + s2;
            const int od = p - 1;
            if (od >= d_lo && od < d_hi && ook) {
                const size_t o = (((size_t)b * D + od) * H + oh) * W + ow;
                out[o] = res ? done + res[o] : done;
            }
            accA = accB + s1;
            accB = s0;
        }
    }
}

}  // namespace

extern "C" int stx_conv3d_c1_fwd(const float* x, const float* w, const float* residual, float* out, int B, int D,
                                 int H, int W, int Cin, void* stream) {
    stx_begin();
    STX_REQUIRE(x && w && out && B > 0 && D > 0 && H > 0 && W > 0, "conv3d_c1_fwd: bad shape");
    STX_REQUIRE(Cin % C1_CK == 0, "conv3d_c1_fwd: Cin=%d must be a multiple of %d", Cin, C1_CK);
    if (Cin == 32 && (long long)H * W * 128 < (1ll << 31)) {   // (other widths, planes beyond the descriptor range: VALU kernel below)
        const int nHt = stx_cdiv(H, C1F_TH), nWt = stx_cdiv(W, C1F_TW);
        const long long ncols = (long long)B * nHt * nWt, units = ncols * D;
        STX_REQUIRE(units < (1ll << 31), "conv3d_c1_fwd: volume too large");
        long long g = C1F_WGS;
        if (g > units / 4) g = units / 4 > 0 ? units / 4 : 1;
        const size_t lds = (size_t)2 * C1F_NV * 27 * 4;
        hipFuncSetAttribute((const void*)conv_c1_fwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(conv_c1_fwd_mfma_kernel, dim3((unsigned)g), dim3(C1_THREADS), lds, (hipStream_t)stream, x, w,
                           residual, out, B, D, H, W, nHt, nWt, (int)ncols);
        return stx_check_launch("conv3d_c1_fwd(mfma)");
    }
    C1Args a;
    a.x = x; a.w = w; a.gy = nullptr; a.res = residual; a.out = out;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin;
    a.nDt = stx_cdiv(D, C1_TD); a.nHt = stx_cdiv(H, C1_TH); a.nWt = stx_cdiv(W, C1_TW);
    a.ntiles = B * a.nDt * a.nHt * a.nWt;
    hipFuncSetAttribute((const void*)conv_c1_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C1_TILE_BYTES);
    hipLaunchKernelGGL(conv_c1_fwd_kernel, dim3(a.ntiles), dim3(C1_THREADS), C1_TILE_BYTES, (hipStream_t)stream, a);
    return stx_check_launch("conv3d_c1_fwd");
}

extern "C" long long stx_conv3d_c1_wgrad_workspace_floats(int Cin) {
    const long long valu = (long long)C1_WGRAD_BLOCKS * Cin * 27;
    const long long mfma = (long long)C1M_BLOCKS * 1024 * ((Cin + 31) / 32);
    return valu > mfma ? valu : mfma;
}

// dw: [1][Cin][27] (torch layout of the Conv3d(Cin, 1, 3) weight gradient)
extern "C" int stx_conv3d_c1_wgrad(const float* x, const float* gy, float* dw, float* workspace, int B, int D, int H,
                                   int W, int Cin, void* stream) {
    stx_begin();
    STX_REQUIRE(x && gy && dw && workspace && B > 0, "conv3d_c1_wgrad: null operand");
    STX_REQUIRE(Cin % C1_CK == 0 && Cin <= 4 * C1_CK, "conv3d_c1_wgrad: Cin=%d must be 16..64 in steps of 16", Cin);
    if (Cin % 32 == 0) {                               // (16- and 48-channel inputs: the VALU kernel below)
        const int nDt = stx_cdiv(D, C1M_TD), nHt = stx_cdiv(H, C1M_TH), nWt = stx_cdiv(W, C1M_TW);
        const long long nt = (long long)B * nDt * nHt * nWt;
        STX_REQUIRE(nt < (1ll << 31), "conv3d_c1_wgrad: volume too large");
        const int nblk = nt < C1M_BLOCKS ? (int)nt : C1M_BLOCKS;
        hipStream_t st = (hipStream_t)stream;
        hipLaunchKernelGGL(conv_c1_wgrad_mfma_kernel, dim3(nblk, Cin / 32), dim3(C1_THREADS), 0, st, x, gy, workspace, B,
                           D, H, W, Cin, nDt, nHt, nWt, (int)nt);
        int rc = stx_check_launch("conv3d_c1_wgrad(mfma)");
        if (rc) return rc;
        hipLaunchKernelGGL(c1m_reduce_kernel, dim3(Cin * 27), dim3(C1_THREADS), 0, st, workspace, nblk, dw);
        return stx_check_launch("conv3d_c1_wgrad(reduce)");
    }
    C1Args a;
    a.x = x; a.w = nullptr; a.gy = gy; a.res = nullptr; a.out = workspace;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin;
    a.nDt = stx_cdiv(D, C1_TD); a.nHt = stx_cdiv(H, C1_TH); a.nWt = stx_cdiv(W, C1_TW);
    a.ntiles = B * a.nDt * a.nHt * a.nWt;
    const int nblk = a.ntiles < C1_WGRAD_BLOCKS ? a.ntiles : C1_WGRAD_BLOCKS;
    const size_t lds = C1_TILE_BYTES + 256 * 4;
    hipStream_t st = (hipStream_t)stream;
    hipFuncSetAttribute((const void*)conv_c1_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(conv_c1_wgrad_kernel, dim3(nblk), dim3(C1_THREADS), lds, st, a);
    int rc = stx_check_launch("conv3d_c1_wgrad");
    if (rc) return rc;
    hipLaunchKernelGGL(c1_colsum_kernel, dim3(Cin * 27), dim3(C1_THREADS), 0, st, workspace, nblk, Cin * 27, dw);
    return stx_check_launch("conv3d_c1_colsum");
}

// gx: [B][D][H][W][Cin] = data gradient of Conv3d(Cin, 1, 3, padding=1) for the output gradient gy [B][D][H][W]
// (reference autograd of models/GwcNet/gwcnet.py:139-153 `classifN.2`).
extern "C" int stx_conv3d_c1_dgrad(const float* gy, const float* w, float* gx, int B, int D, int H, int W, int Cin,
                                   void* stream) {
    stx_begin();
    STX_REQUIRE(gy && w && gx && B > 0 && D > 0 && H > 0 && W > 0, "conv3d_c1_dgrad: bad shape");
    STX_REQUIRE(Cin % 4 == 0 && Cin <= 64, "conv3d_c1_dgrad: Cin=%d must be a multiple of 4, <= 64", Cin);
    const int nHt = stx_cdiv(H, C1D_TH), nWt = stx_cdiv(W, C1D_TW);
    const long long nt = (long long)B * D * nHt * nWt;
    if (Cin == 32 && (long long)H * W * 128 < (1ll << 31) && nt < (1ll << 31)) {   // (other widths: the VALU kernel below)
        const long long g = nt < 256 * 8 ? nt : 256 * 8;
        hipLaunchKernelGGL(conv_c1_dgrad_mfma_kernel, dim3((unsigned)g), dim3(C1_THREADS), 0, (hipStream_t)stream, gy, w, gx, B, D,
                           H, W, nHt, nWt, (int)nt);
        return stx_check_launch("conv3d_c1_dgrad(mfma)");
    }
    hipLaunchKernelGGL(conv_c1_dgrad_kernel, dim3((unsigned)((size_t)B * D * nHt * nWt)), dim3(C1_THREADS), 0,
                       (hipStream_t)stream, gy, w, gx, B, D, H, W, Cin, nHt, nWt);
    return stx_check_launch("conv3d_c1_dgrad");
}
