// ACVNet-specific pieces of the cost-volume path for gfx950 (reference models/ACVNet/acv.py):
//
//  * depth-wise "patch" convolutions  nn.Conv3d(C, C, (1,3,3), groups=C, dilation=d, padding=(0,d,d))
//      acv.py:109-112 (patch: 40 ch, d=1;  patch_l1/l2/l3: 8/16/16 ch, d=1/2/3), used at :183-187.
//    One kernel with a per-channel-quad dilation serves both stages (stage 2 = the three patch_l*
//    convs at once over channel slices 0:8 / 8:24 / 24:40, i.e. the torch.cat of :187 is free), its
//    input gradient (same kernel, taps flipped) and its weight gradient (deterministic partials).
//  * gradient of  softmax(att, dim=2) * concat_volume  (acv.py:196) with respect to the softmax
//    probabilities (the forward and the feature gradients live in cost_volume.hip via `scale`).
// All HBM-bound streaming kernels on channels-last volumes.
#include "stx_common.h"

namespace {

constexpr int ACV_THREADS = 256;

// workgroup id -> position in a walk that gives each of the 8 XCDs one contiguous range (ids are dealt round-robin to XCDs)
__device__ __forceinline__ long long acv_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

struct DwArgs {
    const float* x;      // [B][D][H][W][C]
    const float* w;      // [C][9]
    const int* dil;      // [C/4] dilation of each channel quad
    float* out;          // [B][D][H][W][C]
    int BD, H, W, C, flip;
};

// out[v][c] = sum_{i,j} w[c][3i+j] * x[(h + dil*(i-1), w + dil*(j-1))][c]   (zero padded; flip: taps mirrored)
// A thread keeps ONE channel quad for the whole launch (its 36 weights and its dilation live in registers) and walks voxels:
// a workgroup covers VPB = floor(256 / CQ) voxels x CQ quads per trip, so a voxel's quads are consecutive lanes (16-byte loads,
// CQ x 16 contiguous bytes per voxel) and the nine taps hit L1 / L2 lines the neighbouring voxels of the same trip brought in.
// (The first version dealt (voxel, quad) items round-robin: the quad changed every trip, so every item re-read its 36 weights
// from memory next to its nine float4 of data and decoded its index with 64-bit divisions: 1.23 ms for a 2 x 265 MB ACVNet
// attention volume = 0.11 of the HBM rate, 4.9 ms of the cfg4 train step.)
__global__ __launch_bounds__(ACV_THREADS) void dwconv_hw_kernel(DwArgs a) {
    const int CQ = a.C >> 2;
    const int vpb = ACV_THREADS / CQ;                                // voxels per workgroup and trip
    const int tid = threadIdx.x;
    const int cq = tid % CQ, vl = tid / CQ;
    if (vl >= vpb) return;                                           // (256 mod CQ idle lanes)
    const int dl = a.dil[cq];
    float wc[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) wc[k] = a.w[(size_t)cq * 36 + (k / 9) * 9 + (a.flip ? 8 - k % 9 : k % 9)];   // (flip: taps mirrored once, here)
    const long long nvox = (long long)a.BD * a.H * a.W;
    // XCD-aware contiguous runs (round 4): workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2; with a
    // grid-stride walk every XCD touched every image row and the nine-tap neighbourhoods were fetched by all of them
    // (FETCH_SIZE x 2 = 1.5 GB for a 265 MB volume).  Remapped, the workgroups of one XCD own one contiguous eighth of the
    // voxels, each workgroup a contiguous run inside it.
    const long long wg = acv_xcd_remap(blockIdx.x, gridDim.x);
    const long long trips = (nvox + vpb - 1) / vpb;
    const long long t0 = trips * wg / gridDim.x, t1 = trips * (wg + 1) / gridDim.x;
    for (long long v = t0 * vpb + vl; v < t1 * vpb && v < nvox; v += vpb) {
        const int w = (int)(v % a.W);
        const long long r = v / a.W;
        const int h = (int)(r % a.H);
        const float* xrow = a.x + (r - h) * a.W * a.C + 4 * cq;      // plane (b, d) of this voxel
        // all nine taps are requested before the first is used: out-of-image taps read a clamped address and are zeroed by a
        // select (round 4; the loads used to sit inside the bounds tests -- nine dependent branch / load / wait rounds per
        // output: 0.385 ms per 265 MB volume = 0.17 of the HBM rate, latency-bound)
        float4 xv[9];
        bool ok[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int hh = h + dl * (i - 1);
            const int hc = hh < 0 ? 0 : (hh >= a.H ? a.H - 1 : hh);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ww = w + dl * (j - 1);
                const int wc2 = ww < 0 ? 0 : (ww >= a.W ? a.W - 1 : ww);
                ok[3 * i + j] = hh >= 0 && hh < a.H && ww >= 0 && ww < a.W;
                xv[3 * i + j] = stx_ld4(xrow + ((size_t)hc * a.W + wc2) * a.C);
            }
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float4 t = ok[k] ? xv[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            acc.x = fmaf(t.x, wc[k], acc.x);
            acc.y = fmaf(t.y, wc[9 + k], acc.y);
            acc.z = fmaf(t.z, wc[18 + k], acc.z);
            acc.w = fmaf(t.w, wc[27 + k], acc.w);
        }
        stx_st4(a.out + (size_t)v * a.C + 4 * cq, acc);
    }
}

// The same convolution with a ROLLING WINDOW of input rows in LDS (round 5).  The kernel above asks the caches for nine float4 per
// output float4: at 576x960 it moves a 265 MB volume at 0.22 of the HBM rate, L2-bound.  Here a workgroup owns a strip of TW
// columns of one (b, d) plane (a run of rows of it) and walks down the rows: input row h + 4 is in flight into registers while
// output row h is computed from the seven resident rows h-3 .. h+3 (ring of eight: dilation <= 3), then written to the slot of
// row h-4 -- one barrier per row, every input element is fetched from memory once per strip ((TW + 6) / TW: 1.12 x for TW = 50).
// Out-of-image rows / columns are staged as zeros, so the nine taps are nine unconditional ds_read_b128.  A thread keeps one
// channel quad (its 36 weights, its dilation); TW = 2 x floor(256 / CQ) voxels = two rounds per row.
constexpr int DWR_RING = 8, DWR_HALO = 3;

__global__ __launch_bounds__(ACV_THREADS) void dwconv_hw_roll_kernel(DwArgs a, int TW, int nstrips, int nseg, int seg_rows) {
    STX_DYN_SMEM(smem);
    float* ring = reinterpret_cast<float*>(smem);                    // [DWR_RING][TW + 6][C]
    const int CQ = a.C >> 2, vpb = ACV_THREADS / CQ;
    const int tid = threadIdx.x, cq = tid % CQ, vl = tid / CQ;
    const int EW = TW + 2 * DWR_HALO, rowf = EW * a.C;               // floats per staged row
    const int dl = a.dil[cq < CQ ? cq : 0];
    float wc[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) wc[k] = a.w[(size_t)cq * 36 + (k / 9) * 9 + (a.flip ? 8 - k % 9 : k % 9)];
    // work item: (plane bd, strip, row segment)
    const int item = blockIdx.x;
    const int seg = item % nseg, strip = (item / nseg) % nstrips, bd = item / (nseg * nstrips);
    const int w0 = strip * TW, h_lo = seg * seg_rows, h_hi = (h_lo + seg_rows < a.H) ? h_lo + seg_rows : a.H;
    const float* xp = a.x + (size_t)bd * a.H * a.W * a.C;
    float* op = a.out + (size_t)bd * a.H * a.W * a.C;
    const int nf4 = EW * CQ;                                         // float4 per staged row
    constexpr int NST = 4;                                           // staging float4 per thread (host: nf4 <= 4 x 256)
    float4 stg[NST];
    auto load_row = [&](int hh) {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = tid + k * ACV_THREADS;
            const int v = e / CQ, q = e - v * CQ, ww = w0 - DWR_HALO + v;
            const bool ok = e < nf4 && hh >= 0 && hh < a.H && ww >= 0 && ww < a.W;
            stg[k] = ok ? stx_ld4(xp + ((size_t)hh * a.W + ww) * a.C + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_row = [&](int hh) {
        float* dst = ring + ((hh + 8 * DWR_RING) % DWR_RING) * rowf;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = tid + k * ACV_THREADS;
            if (e < nf4) stx_st4(dst + 4 * e, stg[k]);
        }
    };
    // rows h_lo - 3 .. h_lo + 3 resident before the first output row
    for (int hh = h_lo - DWR_HALO; hh <= h_lo + DWR_HALO; ++hh) {
        load_row(hh);
        store_row(hh);
    }
    __syncthreads();
    for (int h = h_lo; h < h_hi; ++h) {
        load_row(h + DWR_HALO + 1);                                  // in flight during this row's arithmetic
        if (vl < vpb) {
#pragma unroll
            for (int rnd = 0; rnd < 2; ++rnd) {
                const int v = rnd * vpb + vl;                        // column inside the strip
                if (v < TW && w0 + v < a.W) {
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float* rp = ring + ((h + dl * (i - 1) + 8 * DWR_RING) % DWR_RING) * rowf + 4 * cq;
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const float4 t = stx_ld4(rp + (v + DWR_HALO + dl * (j - 1)) * a.C);
                            const int k = 3 * i + j;
                            acc.x = fmaf(t.x, wc[k], acc.x);
                            acc.y = fmaf(t.y, wc[9 + k], acc.y);
                            acc.z = fmaf(t.z, wc[18 + k], acc.z);
                            acc.w = fmaf(t.w, wc[27 + k], acc.w);
                        }
                    }
                    stx_st4(op + ((size_t)h * a.W + w0 + v) * a.C + 4 * cq, acc);
                }
            }
        }
        __syncthreads();                                             // everyone is done with row h - 3 (= the slot of row h + 5)...
        store_row(h + DWR_HALO + 1);                                 // ... row h + 4 goes to the slot row h - 4 left behind
        __syncthreads();
    }
}

// partial[blk][c][k] = sum over the workgroup's voxels of gy[v][c] * x[v + off_k][c]
__global__ __launch_bounds__(ACV_THREADS) void dwconv_hw_wgrad_kernel(const float* __restrict__ x,
                                                                       const float* __restrict__ gy,
                                                                       const int* __restrict__ dil,
                                                                       float* __restrict__ partials, int BD, int H,
                                                                       int W, int C) {
    __shared__ float red[ACV_THREADS * 36];
    const int tid = threadIdx.x;
    const int CQ = C >> 2;
    const int cq = tid % CQ, vl = tid / CQ, VPB = ACV_THREADS / CQ;   // lanes beyond VPB*CQ idle (C=40: 250 of 256)
    const int dl = dil[cq];
    float s[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) s[k] = 0.f;
    const size_t nvox = (size_t)BD * H * W;
    const size_t wg = (size_t)acv_xcd_remap(blockIdx.x, gridDim.x);          // (contiguous runs per XCD: see dwconv_hw_kernel)
    const size_t trips = (nvox + VPB - 1) / VPB;
    const size_t t0 = trips * wg / gridDim.x, t1 = trips * (wg + 1) / gridDim.x;
    for (size_t v = t0 * VPB + vl; vl < VPB && v < t1 * VPB && v < nvox; v += VPB) {
        const int w = (int)(v % W), h = (int)((v / W) % H);
        const size_t bd = v / ((size_t)W * H);
        const float4 g = stx_ld4(gy + v * C + 4 * cq);
        float4 xv[9];                                  // (branch-free loads: see dwconv_hw_kernel)
        bool ok[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int hh = h + dl * (i - 1);
            const int hc = hh < 0 ? 0 : (hh >= H ? H - 1 : hh);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ww = w + dl * (j - 1);
                const int wc2 = ww < 0 ? 0 : (ww >= W ? W - 1 : ww);
                ok[3 * i + j] = hh >= 0 && hh < H && ww >= 0 && ww < W;
                xv[3 * i + j] = stx_ld4(x + ((bd * H + hc) * W + wc2) * C + 4 * cq);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float4 t = ok[k] ? xv[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            s[k] = fmaf(g.x, t.x, s[k]);
            s[9 + k] = fmaf(g.y, t.y, s[9 + k]);
            s[18 + k] = fmaf(g.z, t.z, s[18 + k]);
            s[27 + k] = fmaf(g.w, t.w, s[27 + k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 36; ++k) red[k * ACV_THREADS + tid] = s[k];
    __syncthreads();
    for (int idx = tid; idx < 36 * CQ; idx += ACV_THREADS) {
        const int k = idx / CQ, q = idx % CQ;          // k = comp*9 + tap
        float t = 0.f;
        for (int j = q; j < (ACV_THREADS / CQ) * CQ; j += CQ) t += red[k * ACV_THREADS + j];
        partials[(size_t)blockIdx.x * C * 9 + (size_t)(4 * q + k / 9) * 9 + (k % 9)] = t;
    }
}

// Weight gradient with the same rolling window (round 5): x rows staged once per strip, gy read as it is used; a workgroup owns one
// (plane, strip, row segment) item and writes ONE partial row [C][9] (items <= ACV_WGRAD_BLOCKS); the final cross-lane sum reuses
// the ring's LDS.
__global__ __launch_bounds__(ACV_THREADS) void dwconv_hw_wgrad_roll_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                            const int* __restrict__ dil, float* __restrict__ partials,
                                                                            int H, int W, int C, int TW, int nstrips, int nseg,
                                                                            int seg_rows) {
    STX_DYN_SMEM(smem);
    float* ring = reinterpret_cast<float*>(smem);                    // [DWR_RING][TW + 6][C]; afterwards the reduction buffer
    const int CQ = C >> 2, vpb = ACV_THREADS / CQ;
    const int tid = threadIdx.x, cq = tid % CQ, vl = tid / CQ;
    const int EW = TW + 2 * DWR_HALO, rowf = EW * C;
    const int dl = dil[cq];
    float s[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) s[k] = 0.f;
    const int item = blockIdx.x;
    const int seg = item % nseg, strip = (item / nseg) % nstrips, bd = item / (nseg * nstrips);
    const int w0 = strip * TW, h_lo = seg * seg_rows, h_hi = (h_lo + seg_rows < H) ? h_lo + seg_rows : H;
    const float* xp = x + (size_t)bd * H * W * C;
    const float* gp = gy + (size_t)bd * H * W * C;
    const int nf4 = EW * CQ;
    constexpr int NST = 4;
    float4 stg[NST];
    auto load_row = [&](int hh) {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = tid + k * ACV_THREADS;
            const int v = e / CQ, q = e - v * CQ, ww = w0 - DWR_HALO + v;
            const bool ok = e < nf4 && hh >= 0 && hh < H && ww >= 0 && ww < W;
            stg[k] = ok ? stx_ld4(xp + ((size_t)hh * W + ww) * C + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_row = [&](int hh) {
        float* dst = ring + ((hh + 8 * DWR_RING) % DWR_RING) * rowf;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = tid + k * ACV_THREADS;
            if (e < nf4) stx_st4(dst + 4 * e, stg[k]);
        }
    };
    for (int hh = h_lo - DWR_HALO; hh <= h_lo + DWR_HALO; ++hh) {
        load_row(hh);
        store_row(hh);
    }
    __syncthreads();
    for (int h = h_lo; h < h_hi; ++h) {
        load_row(h + DWR_HALO + 1);
        float4 g[2];
        bool on[2];
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            const int v = rnd * vpb + vl;
            on[rnd] = vl < vpb && v < TW && w0 + v < W;
            g[rnd] = on[rnd] ? stx_ld4(gp + ((size_t)h * W + w0 + v) * C + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            if (on[rnd]) {
                const int v = rnd * vpb + vl;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float* rp = ring + ((h + dl * (i - 1) + 8 * DWR_RING) % DWR_RING) * rowf + 4 * cq;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float4 t = stx_ld4(rp + (v + DWR_HALO + dl * (j - 1)) * C);
                        const int k = 3 * i + j;
                        s[k] = fmaf(g[rnd].x, t.x, s[k]);
                        s[9 + k] = fmaf(g[rnd].y, t.y, s[9 + k]);
                        s[18 + k] = fmaf(g[rnd].z, t.z, s[18 + k]);
                        s[27 + k] = fmaf(g[rnd].w, t.w, s[27 + k]);
                    }
                }
            }
        }
        __syncthreads();
        store_row(h + DWR_HALO + 1);
        __syncthreads();
    }
    // (the last two barriers have retired every read of the ring: it becomes the [36][256] reduction buffer; host: fits)
    float* red = ring;
#pragma unroll
    for (int k = 0; k < 36; ++k) red[k * ACV_THREADS + tid] = s[k];
    __syncthreads();
    for (int idx = tid; idx < 36 * CQ; idx += ACV_THREADS) {
        const int k = idx / CQ, q = idx % CQ;
        float t = 0.f;
        for (int j = q; j < vpb * CQ; j += CQ) t += red[k * ACV_THREADS + j];
        partials[(size_t)blockIdx.x * C * 9 + (size_t)(4 * q + k / 9) * 9 + (k % 9)] = t;
    }
}

__global__ __launch_bounds__(ACV_THREADS) void acv_colsum_kernel(const float* __restrict__ partials, int nrows, int M,
                                                                 float* __restrict__ sums) {
    __shared__ double red[ACV_THREADS];
    const int m = blockIdx.x, tid = threadIdx.x;
    double s = 0.0;
    for (int r = tid; r < nrows; r += ACV_THREADS) s += (double)partials[(size_t)r * M + m];
    red[tid] = s;
    __syncthreads();
    for (int k = ACV_THREADS / 2; k > 0; k >>= 1) {
        if (tid < k) red[tid] += red[tid + k];
        __syncthreads();
    }
    if (tid == 0) sums[m] = (float)red[0];
}

// gscale[b][d][h][w] = sum_c gvol[b][d][h][w][c]*Lc[b][c][h][w] + gvol[..][Cc+c]*(w>=d ? Rc[b][c][h][w-d] : 0)
// (ACV concat semantics: left half unmasked).
__global__ __launch_bounds__(ACV_THREADS) void cv_scale_bwd_kernel(const float* __restrict__ gvol,
                                                                    const float* __restrict__ Lc,
                                                                    const float* __restrict__ Rc,
                                                                    float* __restrict__ gscale, int D, int H, int W,
                                                                    int Cc, int mask_left) {
    const int w = blockIdx.x * ACV_THREADS + threadIdx.x;
    const int h = blockIdx.y % H, d = blockIdx.y / H, b = blockIdx.z;
    if (w >= W) return;
    const int HW = H * W;
    const float* gv = gvol + ((((size_t)b * D + d) * H + h) * W + w) * (2 * Cc);
    const float* lp = Lc + ((size_t)b * Cc * H + h) * W + w;
    const float* rp = Rc + ((size_t)b * Cc * H + h) * W + (w - d);
    const bool rv = w >= d, lv = rv || !mask_left;
    float acc = 0.f;
    for (int c = 0; c < Cc; c += 4) {
        const float4 gl = stx_ld4(gv + c), gr = stx_ld4(gv + Cc + c);
        if (lv) {
            acc = fmaf(gl.x, lp[(size_t)c * HW], acc);
            acc = fmaf(gl.y, lp[(size_t)(c + 1) * HW], acc);
            acc = fmaf(gl.z, lp[(size_t)(c + 2) * HW], acc);
            acc = fmaf(gl.w, lp[(size_t)(c + 3) * HW], acc);
        }
        if (rv) {
            acc = fmaf(gr.x, rp[(size_t)c * HW], acc);
            acc = fmaf(gr.y, rp[(size_t)(c + 1) * HW], acc);
            acc = fmaf(gr.z, rp[(size_t)(c + 2) * HW], acc);
            acc = fmaf(gr.w, rp[(size_t)(c + 3) * HW], acc);
        }
    }
    gscale[(((size_t)b * D + d) * H + h) * W + w] = acc;
}

// Backward of the attention concat volume  vol[d][w][c] = prob[d][w] * (c < Cc ? L[c][w] : R[c - Cc][w - d])  in ONE pass over
// the gradient volume (acv.py:196 + ACVNet/submodule.py:180-191):
//   gprob[d][w] = sum_c gvol[d][w][c] L[c][w] + gvol[d][w][Cc + c] R[c][w - d],
//   gL[c][w] = sum_d prob[d][w] gvol[d][w][c],     gR[c][x] = sum_d prob[d][x + d] gvol[d][x + d][Cc + c].
// A workgroup owns one image row (b, h).  Item (w, channel quad) walks d with its four left features in registers, item
// (x, quad) walks d along the sheared column x + d with its four right features: every element of the gradient volume is read
// exactly once, by one lane, as part of a 128-byte run of eight lanes; the two halves of gprob meet in an LDS image [D][W]
// (ds_add_f32) that is written out coalesced at the end.  The channel quads of a voxel are CQ consecutive lanes with the
// same trip count: their partial dot products are summed by a fixed-order butterfly (CQ a power of two) and ONE lane adds
// the half's total, so a cell receives exactly two adds (left + right) into a zero -- commutative, i.e. gprob is bitwise
// reproducible from run to run.  (Other CQ: one add per quad, order not fixed.)  The three-kernel form (stx_cost_volume_scale_bwd +
// stx_scale_channels + stx_cost_volume_bwd on the generic builder backward) read the 850 MB volume of the cfg4 step three
// times and wrote it once: 1.03 + 0.4 + 0.97 ms.
__global__ __launch_bounds__(ACV_THREADS) void ac_volume_bwd_kernel(const float* __restrict__ gvol, const float* __restrict__ Lc,
                                                                     const float* __restrict__ Rc, const float* __restrict__ prob,
                                                                     float* __restrict__ gL, float* __restrict__ gR,
                                                                     float* __restrict__ gprob, int Cc, int H, int W, int D,
                                                                     int mask_left) {
    STX_DYN_SMEM(smem);
    float* gs = reinterpret_cast<float*>(smem);                       // [D][W]
    const int tid = threadIdx.x;
    const int h = blockIdx.x % H, b = blockIdx.x / H;
    const int CQ = Cc >> 2, CT = 2 * Cc, HW = H * W;
    for (int k = tid; k < D * W; k += ACV_THREADS) gs[k] = 0.f;
    __syncthreads();
    const bool tree = (CQ & (CQ - 1)) == 0 && CQ <= 64;                          // (ACV_THREADS and 64 are multiples of CQ then)
    const float* gvrow = gvol + (((size_t)b * D) * H + h) * (size_t)W * CT;      // + d * H * W * CT + w * CT
    const float* prow = prob + (((size_t)b * D) * H + h) * (size_t)W;            // + d * H * W + w
    const size_t dstride = (size_t)H * W * CT, pstride = (size_t)H * W;
    // (wave-uniform trip counts: the butterfly below needs every lane of a quad group -- and, on the host emulator, of the
    //  wave -- at the shuffle; lanes beyond their range run predicated)
    for (int it0 = 0; it0 < W * CQ; it0 += ACV_THREADS) {
        const int it = it0 + tid;
        const bool valid = it < W * CQ;
        const int q = it % CQ, w = valid ? it / CQ : 0;
        const size_t fo = ((size_t)b * Cc + 4 * q) * HW + (size_t)h * W + w;
        // left half: voxel (d, w), channels 4q..4q+3
        {
            float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
            if (valid) { l0 = Lc[fo]; l1 = Lc[fo + HW]; l2 = Lc[fo + 2 * (size_t)HW]; l3 = Lc[fo + 3 * (size_t)HW]; }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int d1 = !valid ? 0 : (mask_left ? (w + 1 < D ? w + 1 : D) : D);       // (masked left half: only d <= w)
#pragma unroll 4
            for (int d = 0; d < D; ++d) {
                const bool on = d < d1;
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                float p = 0.f;
                if (on) {
                    g = stx_ld4(gvrow + d * dstride + (size_t)w * CT + 4 * q);
                    p = prow[d * pstride + w];
                }
                float dot = fmaf(g.x, l0, fmaf(g.y, l1, fmaf(g.z, l2, g.w * l3)));
                if (tree) {
                    for (int m = 1; m < CQ; m <<= 1) dot += __shfl_xor(dot, m);
                    if (on && q == 0) atomicAdd(&gs[d * W + w], dot);
                } else if (on) {
                    atomicAdd(&gs[d * W + w], dot);
                }
                acc.x = fmaf(g.x, p, acc.x); acc.y = fmaf(g.y, p, acc.y); acc.z = fmaf(g.z, p, acc.z); acc.w = fmaf(g.w, p, acc.w);
            }
            if (valid) { gL[fo] = acc.x; gL[fo + HW] = acc.y; gL[fo + 2 * (size_t)HW] = acc.z; gL[fo + 3 * (size_t)HW] = acc.w; }
        }
        // right half: voxel (d, x + d), channels Cc + 4q.. ; x = the item's column
        {
            const int x = w;
            float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
            if (valid) { r0 = Rc[fo]; r1 = Rc[fo + HW]; r2 = Rc[fo + 2 * (size_t)HW]; r3 = Rc[fo + 3 * (size_t)HW]; }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int d1 = !valid ? 0 : (W - x < D ? W - x : D);
#pragma unroll 4
            for (int d = 0; d < D; ++d) {
                const bool on = d < d1;
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                float p = 0.f;
                if (on) {
                    g = stx_ld4(gvrow + d * dstride + (size_t)(x + d) * CT + Cc + 4 * q);
                    p = prow[d * pstride + x + d];
                }
                float dot = fmaf(g.x, r0, fmaf(g.y, r1, fmaf(g.z, r2, g.w * r3)));
                if (tree) {
                    for (int m = 1; m < CQ; m <<= 1) dot += __shfl_xor(dot, m);
                    if (on && q == 0) atomicAdd(&gs[d * W + x + d], dot);
                } else if (on) {
                    atomicAdd(&gs[d * W + x + d], dot);
                }
                acc.x = fmaf(g.x, p, acc.x); acc.y = fmaf(g.y, p, acc.y); acc.z = fmaf(g.z, p, acc.z); acc.w = fmaf(g.w, p, acc.w);
            }
            if (valid) { gR[fo] = acc.x; gR[fo + HW] = acc.y; gR[fo + 2 * (size_t)HW] = acc.z; gR[fo + 3 * (size_t)HW] = acc.w; }
        }
    }
    __syncthreads();
    for (int k = tid; k < D * W; k += ACV_THREADS) {
        const int d = k / W, w = k - d * W;
        gprob[(((size_t)b * D + d) * H + h) * W + w] = gs[k];
    }
}

// out = a * s (s broadcast over the channel axis): gvol * prob, the feature-gradient input of the ac-volume
__global__ __launch_bounds__(ACV_THREADS) void scale_channels_kernel(const float* __restrict__ x,
                                                                      const float* __restrict__ s,
                                                                      float* __restrict__ out, size_t nq, int CQ) {
    for (size_t i = (size_t)blockIdx.x * ACV_THREADS + threadIdx.x; i < nq; i += (size_t)gridDim.x * ACV_THREADS) {
        const float m = s[i / CQ];
        float4 v = stx_ld4(x + i * 4);
        v.x *= m; v.y *= m; v.z *= m; v.w *= m;
        stx_st4(out + i * 4, v);
    }
}

constexpr int ACV_WGRAD_BLOCKS = 1024;

// Dynamic LDS above 64 KiB needs the kernel's MaxDynamicSharedMemorySize attribute raised.  Asked once per kernel and size
// (not on every launch); a refusal -- a device or partition mode with less LDS than gfx950's 160 KiB -- makes the caller take
// the cache-fed kernel instead of failing.
bool acv_lds_granted(const void* kern, size_t lds, int& granted) {
    if (lds <= 64 * 1024 || (int)lds <= granted) return true;
    if (granted < 0) return false;                                   // refused before
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        granted = -1;
        return false;
    }
    granted = (int)lds;
    return true;
}

int acv_grid(size_t n) {
    size_t g = (n + ACV_THREADS - 1) / ACV_THREADS;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int stx_dwconv_hw_fwd(const float* x, const float* w, const int* dil, float* out, int B, int D, int H,
                                 int W, int C, int flip, void* stream) {
    stx_begin();
    STX_REQUIRE(x && w && dil && out && B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "dwconv_hw_fwd: bad args");
    DwArgs a;
    a.x = x; a.w = w; a.dil = dil; a.out = out; a.BD = B * D; a.H = H; a.W = W; a.C = C; a.flip = flip;
    STX_REQUIRE(C / 4 <= ACV_THREADS, "dwconv_hw_fwd: C=%d (at most %d channels)", C, 4 * ACV_THREADS);
    const size_t nvox = (size_t)B * D * H * W, vpb = ACV_THREADS / (C / 4);
    // rolling-window kernel: strips of TW = 2 vpb columns, rows cut into segments so that the launch has >= ~3 workgroups per CU
    {
        const int TW = 2 * (int)vpb, EW = TW + 2 * DWR_HALO;
        const size_t lds = (size_t)DWR_RING * EW * C * sizeof(float);
        const int nf4 = EW * (C / 4);
        if (stx_tune(STX_TUNE_DWCONV_ROLL) && lds <= 160 * 1024 && nf4 <= 4 * ACV_THREADS && H >= 8) {
            const int nstrips = stx_cdiv(W, TW);
            const long long base = (long long)B * D * nstrips;
            int nseg = (int)((768 + base - 1) / base);
            if (nseg < 1) nseg = 1;
            if (nseg > H / 8) nseg = H / 8 > 0 ? H / 8 : 1;          // a segment re-stages 6 halo rows: keep it >= 8 rows
            const int seg_rows = stx_cdiv(H, nseg);
            nseg = stx_cdiv(H, seg_rows);
            static int granted = 0;
            if (base * nseg < (1ll << 31) && acv_lds_granted((const void*)dwconv_hw_roll_kernel, lds, granted)) {
                hipLaunchKernelGGL(dwconv_hw_roll_kernel, dim3((unsigned)(base * nseg)), dim3(ACV_THREADS), lds, (hipStream_t)stream, a,
                                   TW, nstrips, nseg, seg_rows);
                return stx_check_launch("dwconv_hw_fwd(roll)");
            }
        }
    }
    const size_t g = (nvox + vpb - 1) / vpb;
    hipLaunchKernelGGL(dwconv_hw_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(ACV_THREADS), 0, (hipStream_t)stream, a);
    return stx_check_launch("dwconv_hw_fwd");
}

extern "C" long long stx_dwconv_hw_wgrad_workspace_floats(int C) { return (long long)ACV_WGRAD_BLOCKS * C * 9; }

extern "C" int stx_dwconv_hw_wgrad(const float* x, const float* gy, const int* dil, float* dw, float* workspace,
                                   int B, int D, int H, int W, int C, void* stream) {
    stx_begin();
    STX_REQUIRE(x && gy && dil && dw && workspace && C % 4 == 0 && C / 4 <= ACV_THREADS,
                "dwconv_hw_wgrad: bad args (C=%d)", C);
    hipStream_t st = (hipStream_t)stream;
    int nrows = ACV_WGRAD_BLOCKS;
    bool rolled = false;
    {
        // rolling-window form: one partial row per (plane, strip, row segment) item, at most ACV_WGRAD_BLOCKS of them
        const int vpb = ACV_THREADS / (C / 4), TW = 2 * vpb, EW = TW + 2 * DWR_HALO, nstrips = stx_cdiv(W, TW);
        size_t lds = (size_t)DWR_RING * EW * C * sizeof(float);
        const size_t red = (size_t)36 * ACV_THREADS * sizeof(float);
        if (lds < red) lds = red;
        const long long base = (long long)B * D * nstrips;
        // (base > ACV_WGRAD_BLOCKS -- more (plane, strip) items than partial rows of the workspace, e.g. B >= 5 at 576x960 -- takes
        //  the cache-fed kernel below: about twice the time per launch, profiles/r05_dwconv_rolling_window_ab_callU.txt)
        static int granted = 0;
        if (stx_tune(STX_TUNE_DWCONV_ROLL) && lds <= 160 * 1024 && EW * (C / 4) <= 4 * ACV_THREADS && H >= 8 && base <= ACV_WGRAD_BLOCKS &&
            acv_lds_granted((const void*)dwconv_hw_wgrad_roll_kernel, lds, granted)) {
            int nseg = (int)(ACV_WGRAD_BLOCKS / base);
            if (nseg > H / 8) nseg = H / 8;
            if (nseg < 1) nseg = 1;
            const int seg_rows = stx_cdiv(H, nseg);
            nseg = stx_cdiv(H, seg_rows);
            nrows = (int)(base * nseg);
            hipLaunchKernelGGL(dwconv_hw_wgrad_roll_kernel, dim3(nrows), dim3(ACV_THREADS), lds, st, x, gy, dil, workspace, H, W, C, TW,
                               nstrips, nseg, seg_rows);
            rolled = true;
        }
    }
    if (!rolled)
        hipLaunchKernelGGL(dwconv_hw_wgrad_kernel, dim3(ACV_WGRAD_BLOCKS), dim3(ACV_THREADS), 0, st, x, gy, dil, workspace,
                           B * D, H, W, C);
    int rc = stx_check_launch("dwconv_hw_wgrad");
    if (rc) return rc;
    hipLaunchKernelGGL(acv_colsum_kernel, dim3(C * 9), dim3(ACV_THREADS), 0, st, workspace, nrows, C * 9, dw);
    return stx_check_launch("dwconv_hw_wgrad_colsum");
}

extern "C" int stx_cost_volume_scale_bwd(const float* gvol, const float* Lc, const float* Rc, float* gscale, int B,
                                         int Cc, int H, int W, int D, int mask_left, void* stream) {
    stx_begin();
    STX_REQUIRE(gvol && Lc && Rc && gscale && Cc % 4 == 0 && B > 0, "cost_volume_scale_bwd: bad args");
    dim3 grid(stx_cdiv(W, ACV_THREADS), D * H, B);
    hipLaunchKernelGGL(cv_scale_bwd_kernel, grid, dim3(ACV_THREADS), 0, (hipStream_t)stream, gvol, Lc, Rc, gscale, D, H,
                       W, Cc, mask_left);
    return stx_check_launch("cost_volume_scale_bwd");
}

// (the LDS image [D][W] must fit: D * W * 4 <= 150 KiB; larger rows take the three-kernel form)
extern "C" int stx_ac_volume_bwd(const float* gvol, const float* Lc, const float* Rc, const float* prob, float* gL, float* gR,
                                 float* gprob, int B, int Cc, int H, int W, int D, int mask_left, void* stream) {
    stx_begin();
    STX_REQUIRE(gvol && Lc && Rc && prob && gL && gR && gprob && Cc > 0 && Cc % 4 == 0 && B > 0 && H > 0 && W > 0 && D > 0,
                "ac_volume_bwd: bad args");
    const size_t lds = (size_t)D * W * 4;
    STX_REQUIRE(lds <= 150 * 1024, "ac_volume_bwd: D * W = %d x %d does not fit the LDS image (use the three-kernel form)", D, W);
    if (hipFuncSetAttribute((const void*)ac_volume_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return stx_set_error(STX_ERR_LAUNCH, "ac_volume_bwd: %d bytes of dynamic LDS refused by this device (the library is "
                                             "built for the 160 KiB LDS of gfx950)", (int)lds);
    hipLaunchKernelGGL(ac_volume_bwd_kernel, dim3((unsigned)(B * H)), dim3(ACV_THREADS), lds, (hipStream_t)stream, gvol, Lc, Rc, prob,
                       gL, gR, gprob, Cc, H, W, D, mask_left);
    return stx_check_launch("ac_volume_bwd");
}

extern "C" int stx_scale_channels(const float* x, const float* s, float* out, long long nvox, int C, void* stream) {
    stx_begin();
    STX_REQUIRE(x && s && out && nvox > 0 && C % 4 == 0, "scale_channels: bad args");
    const size_t nq = (size_t)nvox * (C / 4);
    hipLaunchKernelGGL(scale_channels_kernel, dim3(acv_grid(nq)), dim3(ACV_THREADS), 0, (hipStream_t)stream, x, s, out,
                       nq, C / 4);
    return stx_check_launch("scale_channels");
}
