// Train-mode BatchNorm3d around the MFMA convolutions, for gfx950 (channels-last activations).
//
// Replaces the nn.BatchNorm3d half of convbn_3d (reference models/GwcNet/submodule.py:17-20 and
// twins) plus the ReLU / residual adds that follow it (GwcNet/gwcnet.py:96-103,185;
// PSMNet/stackhourglass.py:31-48,123-132) when the module is in train() mode, and their backward.
// (In eval() mode BN is folded into the convolution epilogue and none of this runs.)
//
//   forward : conv epilogue emits per-workgroup (sum z, sum z^2) partials  -> stx_bn_finalize
//             (fp64 reduction; mean/var -> scale=gamma*invstd, shift=beta-mean*scale; running
//             stats updated with momentum and the unbiased variance, as torch does)
//             -> stx_bn_apply: y = act(z1*scale1+shift1 [+ z2*scale2+shift2 | + residual])
//   backward: stx_bn_bwd_reduce: sums of g, g*xhat1, g*xhat2 with g = gy*[y>0]
//             stx_bn_bwd_apply : dz_k = gamma_k*invstd_k*(g - mean(g) - xhat_k*mean(g*xhat_k))
// All kernels are HBM-bound streaming passes (16-B accesses, one float4 channel quad per lane).
// GROUPS: the streaming passes take `groups` consecutive slabs of `nvox` voxels each with their OWN statistics (scale / shift /
// mean / invstd / sums are [groups][C]; gamma is shared): blockIdx.y = group.  The 2-D feature CNN runs the left and the right
// view as one batch through its convolutions while every BatchNorm keeps per-view statistics, as the reference's two
// separate extractor calls do (models/GwcNet/gwcnet.py:172-173).
#include "stx_common.h"
#include <stdlib.h>

namespace {

constexpr int BN_THREADS = 256;
constexpr int BN_RED_BLOCKS = 1024;

// out[m] = sum_r partials[r][m]  (fp64 accumulate), one workgroup per column m.
__device__ __forceinline__ double bn_block_sum(double v, double* red, int tid) {
    red[tid] = v;
    __syncthreads();
    for (int s = BN_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(BN_THREADS) void bn_finalize_kernel(
    const float* __restrict__ partials, int nrows, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
    float momentum, float eps, float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
    float* __restrict__ invstd_out, int groups) {
    __shared__ double red[BN_THREADS];
    const int c = blockIdx.x, tid = threadIdx.x;
    // groups: slab after slab (own statistics each; the running statistics move once per slab, in order, like separate calls)
    for (int g = 0; g < groups; ++g, partials += (size_t)nrows * 2 * C, scale += C, shift += C, mean_out += C, invstd_out += C) {
    double s1 = 0.0, s2 = 0.0;
    for (int r = tid; r < nrows; r += BN_THREADS) {
        s1 += (double)partials[(size_t)r * 2 * C + c];
        s2 += (double)partials[(size_t)r * 2 * C + C + c];
    }
    s1 = bn_block_sum(s1, red, tid);
    s2 = bn_block_sum(s2, red, tid);
    if (tid == 0) {
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        const float sc = g * invstd;
        scale[c] = sc;
        shift[c] = bt - (float)mean * sc;
        mean_out[c] = (float)mean;
        invstd_out[c] = invstd;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        if (running_var) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
    }
}

// Same result for C % 4 == 0 and many partial rows (the L0 layers emit ~13 000 of them): one workgroup per channel QUAD,
// 512 threads, every thread two 16-byte loads per row (the first version read one dword per load with a 2C-float
// stride and a dependent add: 18-30 us per BN layer, ~0.6 ms of a GwcNet_GC train step).  fp64 throughout; the
// cross-lane sums go through the wave (bit moves of the two double halves), then one LDS round.  Fixed summation
// order -> run-to-run deterministic.
constexpr int BN_FIN_THREADS = 512;

__device__ __forceinline__ double bn_shfl_down_f64(double v, int delta) {
    union { double d; float f[2]; } a, b;
    a.d = v;
    b.f[0] = __shfl_down(a.f[0], delta);
    b.f[1] = __shfl_down(a.f[1], delta);
    return b.d;
}

__global__ __launch_bounds__(BN_FIN_THREADS) void bn_finalize4_kernel(
    const float* __restrict__ partials, int nrows, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
    float momentum, float eps, float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
    float* __restrict__ invstd_out, int groups) {
    __shared__ double red[BN_FIN_THREADS / 64][8];
    const int c0 = blockIdx.x * 4, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int g = 0; g < groups; ++g, partials += (size_t)nrows * 2 * C, scale += C, shift += C, mean_out += C, invstd_out += C) {
    if (g) __syncthreads();                                          // `red` of the previous slab has been read
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const float* p = partials + c0;
#pragma unroll 4
    for (int r = tid; r < nrows; r += BN_FIN_THREADS) {
        const float4 a = stx_ld4(p + (size_t)r * 2 * C), b = stx_ld4(p + (size_t)r * 2 * C + C);
        acc[0] += (double)a.x; acc[1] += (double)a.y; acc[2] += (double)a.z; acc[3] += (double)a.w;
        acc[4] += (double)b.x; acc[5] += (double)b.y; acc[6] += (double)b.z; acc[7] += (double)b.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc[k] += bn_shfl_down_f64(acc[k], d);
        if (lane == 0) red[wave][k] = acc[k];
    }
    __syncthreads();
    if (tid < 4) {
        double s1 = 0.0, s2 = 0.0;
        for (int w = 0; w < BN_FIN_THREADS / 64; ++w) { s1 += red[w][tid]; s2 += red[w][4 + tid]; }
        const int c = c0 + tid;
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        const float sc = g * invstd;
        scale[c] = sc;
        shift[c] = bt - (float)mean * sc;
        mean_out[c] = (float)mean;
        invstd_out[c] = invstd;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        if (running_var) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
    }
}

__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(
    const float* __restrict__ z1, const float* __restrict__ scale1, const float* __restrict__ shift1,
    const float* __restrict__ z2, const float* __restrict__ scale2, const float* __restrict__ shift2,
    float* __restrict__ out, size_t nquads, int CQ, int relu) {
    {   // group blockIdx.y: its slab of the activations, its rows of the per-group vectors
        const size_t go = (size_t)blockIdx.y * nquads * 4, gc = (size_t)blockIdx.y * CQ * 4;
        z1 += go; out += go; scale1 += gc; shift1 += gc;
        if (z2) z2 += go;
        if (scale2) { scale2 += gc; shift2 += gc; }
    }
    for (size_t i = (size_t)blockIdx.x * BN_THREADS + threadIdx.x; i < nquads; i += (size_t)gridDim.x * BN_THREADS) {
        const int cq = (int)(i % CQ) * 4;
        float4 v = stx_ld4(z1 + i * 4);
        const float4 a = stx_ld4(scale1 + cq), b = stx_ld4(shift1 + cq);
        v.x = fmaf(v.x, a.x, b.x); v.y = fmaf(v.y, a.y, b.y); v.z = fmaf(v.z, a.z, b.z); v.w = fmaf(v.w, a.w, b.w);
        if (z2) {
            float4 u = stx_ld4(z2 + i * 4);
            if (scale2) {
                const float4 a2 = stx_ld4(scale2 + cq), b2 = stx_ld4(shift2 + cq);
                u.x = fmaf(u.x, a2.x, b2.x); u.y = fmaf(u.y, a2.y, b2.y);
                u.z = fmaf(u.z, a2.z, b2.z); u.w = fmaf(u.w, a2.w, b2.w);
            }
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        if (relu) {                                             // activation code: 1 ReLU, 2 Mish (stx_common.h)
            v.x = stx_act(v.x, relu); v.y = stx_act(v.y, relu); v.z = stx_act(v.z, relu); v.w = stx_act(v.w, relu);
        }
        stx_st4(out + i * 4, v);
    }
}

// partial[blk][0..2][C] = sum g, sum g*xhat1, sum g*xhat2 over the workgroup's voxels.
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_reduce_kernel(
    const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ z1,
    const float* __restrict__ mean1, const float* __restrict__ invstd1, const float* __restrict__ z2,
    const float* __restrict__ mean2, const float* __restrict__ invstd2, const float* __restrict__ sc1,
    const float* __restrict__ sh1, const float* __restrict__ sc2, const float* __restrict__ sh2, float* __restrict__ partials,
    size_t nvox, int C, int relu) {
    __shared__ float red[BN_THREADS * 12];
    const int tid = threadIdx.x;
    const int CQ = C >> 2;
    const int cq = tid % CQ, vl = tid / CQ, VPB = BN_THREADS / CQ;
    {   // group blockIdx.y
        const size_t go = (size_t)blockIdx.y * nvox * C, gc = (size_t)blockIdx.y * C;
        gy += go; z1 += go; mean1 += gc; invstd1 += gc;
        if (y) y += go;
        if (z2) z2 += go;
        if (mean2) { mean2 += gc; invstd2 += gc; }
        if (sc1) { sc1 += gc; sh1 += gc; }
        if (sc2) { sc2 += gc; sh2 += gc; }
        partials += (size_t)blockIdx.y * gridDim.x * 3 * C;
    }
    const float4 m1 = stx_ld4(mean1 + 4 * cq), i1 = stx_ld4(invstd1 + 4 * cq);
    float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f), i2 = m2;
    const bool has2 = z2 && mean2;
    if (has2) { m2 = stx_ld4(mean2 + 4 * cq); i2 = stx_ld4(invstd2 + 4 * cq); }
    // ReLU mask without reading y: the forward pass computed y = relu(fmaf(z1, sc1, sh1) [+ fmaf(z2, sc2, sh2)]) -- the
    // same expression on the same operands gives the same sign bit for bit, and z1 / z2 are read here anyway
    const bool remask = relu && !y;
    float4 a1 = m2, b1 = m2, a2 = m2, b2 = m2;
    if (remask) {
        a1 = stx_ld4(sc1 + 4 * cq); b1 = stx_ld4(sh1 + 4 * cq);
        if (has2) { a2 = stx_ld4(sc2 + 4 * cq); b2 = stx_ld4(sh2 + 4 * cq); }
    }
    float s[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k] = 0.f;
    // (C / 4 need not divide the workgroup: the threads beyond VPB whole voxels idle and contribute zeros)
    for (size_t v = vl < VPB ? (size_t)blockIdx.x * VPB + vl : nvox; v < nvox; v += (size_t)gridDim.x * VPB) {
        const size_t o = v * C + 4 * cq;
        float4 g = stx_ld4(gy + o);
        const float4 a = stx_ld4(z1 + o);
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has2) c = stx_ld4(z2 + o);
        if (relu) {
            float4 yy;
            if (remask) {
                yy.x = fmaf(a.x, a1.x, b1.x); yy.y = fmaf(a.y, a1.y, b1.y); yy.z = fmaf(a.z, a1.z, b1.z); yy.w = fmaf(a.w, a1.w, b1.w);
                if (has2) {
                    yy.x += fmaf(c.x, a2.x, b2.x); yy.y += fmaf(c.y, a2.y, b2.y);
                    yy.z += fmaf(c.z, a2.z, b2.z); yy.w += fmaf(c.w, a2.w, b2.w);
                }
            } else {
                yy = stx_ld4(y + o);
            }
            // (Mish: yy is the pre-activation value, remask path only; ReLU / LeakyReLU: either value has the sign)
            g.x = stx_act_bwd(g.x, yy.x, relu); g.y = stx_act_bwd(g.y, yy.y, relu);
            g.z = stx_act_bwd(g.z, yy.z, relu); g.w = stx_act_bwd(g.w, yy.w, relu);
        }
        s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w;
        s[4] = fmaf(g.x, (a.x - m1.x) * i1.x, s[4]); s[5] = fmaf(g.y, (a.y - m1.y) * i1.y, s[5]);
        s[6] = fmaf(g.z, (a.z - m1.z) * i1.z, s[6]); s[7] = fmaf(g.w, (a.w - m1.w) * i1.w, s[7]);
        if (has2) {
            s[8] = fmaf(g.x, (c.x - m2.x) * i2.x, s[8]); s[9] = fmaf(g.y, (c.y - m2.y) * i2.y, s[9]);
            s[10] = fmaf(g.z, (c.z - m2.z) * i2.z, s[10]); s[11] = fmaf(g.w, (c.w - m2.w) * i2.w, s[11]);
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) red[k * BN_THREADS + tid] = s[k];
    __syncthreads();
    // threads with the same cq differ by multiples of CQ
    for (int idx = tid; idx < 12 * CQ; idx += BN_THREADS) {
        const int k = idx / CQ, q = idx % CQ;
        float t = 0.f;
        for (int j = q; j < BN_THREADS; j += CQ) t += red[k * BN_THREADS + j];
        // k = which*4 + component
        partials[((size_t)blockIdx.x * 3 + (k >> 2)) * C + 4 * q + (k & 3)] = t;
    }
}

// partials[blk][0][C] = sum z, partials[blk][1][C] = sum z^2 over the workgroup's voxels: batch statistics of a channels-last
// activation whose producer has no fused epilogue (the MIOpen convolutions of the 2-D feature CNN; the 3-D convolutions
// emit these sums themselves).  Same row format as the conv epilogues' slabs -> stx_bn_finalize.
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(const float* __restrict__ z, float* __restrict__ partials,
                                                              size_t nvox, int C) {
    __shared__ float red[BN_THREADS * 8];
    const int tid = threadIdx.x;
    const int CQ = C >> 2;
    const int cq = tid % CQ, vl = tid / CQ, VPB = BN_THREADS / CQ;
    z += (size_t)blockIdx.y * nvox * C;                             // group blockIdx.y
    partials += (size_t)blockIdx.y * gridDim.x * 2 * C;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    const size_t step = (size_t)gridDim.x * VPB;
    size_t v = vl < VPB ? (size_t)blockIdx.x * VPB + vl : nvox;  // (threads beyond VPB whole voxels idle: C / 4 need not divide the workgroup)
    for (; v + 3 * step < nvox; v += 4 * step) {                 // four independent 16-byte loads in flight per lane
        float4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = stx_ld4(z + (v + u * step) * C + 4 * cq);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s[0] += a[u].x; s[1] += a[u].y; s[2] += a[u].z; s[3] += a[u].w;
            s[4] = fmaf(a[u].x, a[u].x, s[4]); s[5] = fmaf(a[u].y, a[u].y, s[5]);
            s[6] = fmaf(a[u].z, a[u].z, s[6]); s[7] = fmaf(a[u].w, a[u].w, s[7]);
        }
    }
    for (; v < nvox; v += step) {
        const float4 a = stx_ld4(z + v * C + 4 * cq);
        s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w;
        s[4] = fmaf(a.x, a.x, s[4]); s[5] = fmaf(a.y, a.y, s[5]); s[6] = fmaf(a.z, a.z, s[6]); s[7] = fmaf(a.w, a.w, s[7]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[k * BN_THREADS + tid] = s[k];
    __syncthreads();
    for (int idx = tid; idx < 8 * CQ; idx += BN_THREADS) {
        const int k = idx / CQ, q = idx % CQ;
        float t = 0.f;
        for (int j = q; j < BN_THREADS; j += CQ) t += red[k * BN_THREADS + j];      // fixed order: deterministic
        partials[((size_t)blockIdx.x * 2 + (k >> 2)) * C + 4 * q + (k & 3)] = t;
    }
}

// sums[g][m] = sum over rows of partials[g][r][m], fp64 accumulate; one workgroup per column, the groups one after the other;
// with more than one group a further slab sums[groups][m] = sums[0][m] + sums[1][m] + .. (fp32, group order): the parameter
// gradients of a BatchNorm shared by the groups (the two views of the 2-D CNN) without a reduction launch of their own.
__global__ __launch_bounds__(BN_THREADS) void bn_colsum_kernel(const float* __restrict__ partials, int nrows, int M,
                                                               float* __restrict__ sums, int groups) {
    __shared__ double red[BN_THREADS];
    const int m = blockIdx.x, tid = threadIdx.x;
    float tot = 0.f;
    for (int g = 0; g < groups; ++g) {
        const float* p = partials + (size_t)g * nrows * M;
        double s = 0.0;
        for (int r = tid; r < nrows; r += BN_THREADS) s += (double)p[(size_t)r * M + m];
        s = bn_block_sum(s, red, tid);
        if (tid == 0) {
            sums[(size_t)g * M + m] = (float)s;
            tot += (float)s;
        }
    }
    if (groups > 1 && tid == 0) sums[(size_t)groups * M + m] = tot;
}

__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(
    const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ z1,
    const float* __restrict__ mean1, const float* __restrict__ invstd1, const float* __restrict__ gamma1,
    const float* __restrict__ z2, const float* __restrict__ mean2, const float* __restrict__ invstd2,
    const float* __restrict__ gamma2, const float* __restrict__ sc1, const float* __restrict__ sh1,
    const float* __restrict__ sc2, const float* __restrict__ sh2, const float* __restrict__ sums, float* __restrict__ dz1,
    float* __restrict__ dz2, float* __restrict__ gout, size_t nquads, int C, int relu, float inv_n) {
    const int CQ = C >> 2;
    {   // group blockIdx.y (gamma is shared between the groups)
        const size_t go = (size_t)blockIdx.y * nquads * 4, gc = (size_t)blockIdx.y * C;
        gy += go; z1 += go; dz1 += go; mean1 += gc; invstd1 += gc; sums += 3 * gc;
        if (y) y += go;
        if (z2) z2 += go;
        if (dz2) dz2 += go;
        if (gout) gout += go;
        if (mean2) { mean2 += gc; invstd2 += gc; }
        if (sc1) { sc1 += gc; sh1 += gc; }
        if (sc2) { sc2 += gc; sh2 += gc; }
    }
    const bool has2 = z2 && mean2 && dz2;
    const bool remask = relu && !y;                 // (see bn_bwd_reduce_kernel)
    for (size_t i = (size_t)blockIdx.x * BN_THREADS + threadIdx.x; i < nquads; i += (size_t)gridDim.x * BN_THREADS) {
        const int c = (int)(i % CQ) * 4;
        float4 g = stx_ld4(gy + i * 4);
        const float4 a = stx_ld4(z1 + i * 4);
        float4 a2v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has2) a2v = stx_ld4(z2 + i * 4);
        if (relu) {
            float4 yy;
            if (remask) {
                const float4 p1 = stx_ld4(sc1 + c), q1 = stx_ld4(sh1 + c);
                yy.x = fmaf(a.x, p1.x, q1.x); yy.y = fmaf(a.y, p1.y, q1.y); yy.z = fmaf(a.z, p1.z, q1.z); yy.w = fmaf(a.w, p1.w, q1.w);
                if (has2) {
                    const float4 p2 = stx_ld4(sc2 + c), q2 = stx_ld4(sh2 + c);
                    yy.x += fmaf(a2v.x, p2.x, q2.x); yy.y += fmaf(a2v.y, p2.y, q2.y);
                    yy.z += fmaf(a2v.z, p2.z, q2.z); yy.w += fmaf(a2v.w, p2.w, q2.w);
                }
            } else {
                yy = stx_ld4(y + i * 4);
            }
            // (Mish: yy is the pre-activation value, remask path only; ReLU / LeakyReLU: either value has the sign)
            g.x = stx_act_bwd(g.x, yy.x, relu); g.y = stx_act_bwd(g.y, yy.y, relu);
            g.z = stx_act_bwd(g.z, yy.z, relu); g.w = stx_act_bwd(g.w, yy.w, relu);
        }
        if (gout) stx_st4(gout + i * 4, g);
        const float4 sg = stx_ld4(sums + c);
        {
            const float4 m = stx_ld4(mean1 + c), is = stx_ld4(invstd1 + c);
            const float4 gm = gamma1 ? stx_ld4(gamma1 + c) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 sx = stx_ld4(sums + C + c);
            float4 d;
            d.x = gm.x * is.x * (g.x - sg.x * inv_n - (a.x - m.x) * is.x * sx.x * inv_n);
            d.y = gm.y * is.y * (g.y - sg.y * inv_n - (a.y - m.y) * is.y * sx.y * inv_n);
            d.z = gm.z * is.z * (g.z - sg.z * inv_n - (a.z - m.z) * is.z * sx.z * inv_n);
            d.w = gm.w * is.w * (g.w - sg.w * inv_n - (a.w - m.w) * is.w * sx.w * inv_n);
            stx_st4(dz1 + i * 4, d);
        }
        if (has2) {
            const float4 a = a2v, m = stx_ld4(mean2 + c), is = stx_ld4(invstd2 + c);
            const float4 gm = gamma2 ? stx_ld4(gamma2 + c) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 sx = stx_ld4(sums + 2 * C + c);
            float4 d;
            d.x = gm.x * is.x * (g.x - sg.x * inv_n - (a.x - m.x) * is.x * sx.x * inv_n);
            d.y = gm.y * is.y * (g.y - sg.y * inv_n - (a.y - m.y) * is.y * sx.y * inv_n);
            d.z = gm.z * is.z * (g.z - sg.z * inv_n - (a.z - m.z) * is.z * sx.z * inv_n);
            d.w = gm.w * is.w * (g.w - sg.w * inv_n - (a.w - m.w) * is.w * sx.w * inv_n);
            stx_st4(dz2 + i * 4, d);
        }
    }
}

int bn_grid(size_t nquads) {
    size_t g = (nquads + BN_THREADS - 1) / BN_THREADS;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int stx_bn_reduce_blocks(void) {
    stx_begin(); return BN_RED_BLOCKS; }

static int bn_finalize_launch(const float* partials, int nrows, int C, double count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                              float* mean, float* invstd, int groups, void* stream) {
    if (C % 4 == 0 && nrows >= 256)
        hipLaunchKernelGGL(bn_finalize4_kernel, dim3(C / 4), dim3(BN_FIN_THREADS), 0, (hipStream_t)stream, partials, nrows, C,
                           count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, invstd, groups);
    else
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(BN_THREADS), 0, (hipStream_t)stream, partials, nrows, C, count,
                           gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, invstd, groups);
    return stx_check_launch("bn_finalize");
}

extern "C" int stx_bn_finalize(const float* partials, int nrows, int C, double count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               float* scale, float* shift, float* mean, float* invstd, void* stream) {
    stx_begin();
    STX_REQUIRE(partials && nrows > 0 && C > 0 && count > 0 && scale && shift && mean && invstd, "bn_finalize: bad args");
    return bn_finalize_launch(partials, nrows, C, count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift,
                              mean, invstd, 1, stream);
}

extern "C" int stx_bn_finalize_groups(const float* partials, int nrows, int C, double count, const float* gamma,
                                      const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                      float* out, int groups, void* stream) {
    stx_begin();
    STX_REQUIRE(partials && nrows > 0 && C > 0 && count > 0 && out && groups >= 1, "bn_finalize_groups: bad args");
    const size_t gc = (size_t)groups * C;
    return bn_finalize_launch(partials, nrows, C, count, gamma, beta, running_mean, running_var, momentum, eps, out, out + gc,
                              out + 2 * gc, out + 3 * gc, groups, stream);
}

extern "C" int stx_bn_stats_rows(long long nvox, int C) {
    stx_begin();
    if (C < 4 || C % 4 != 0 || C / 4 > BN_THREADS || nvox < 1) return 0;
    const long long vpb = BN_THREADS / (C / 4);
    const long long g = (nvox + 4 * vpb - 1) / (4 * vpb);          // >= 4 voxels per lane before another workgroup is added
    // at most 256 rows: the tensors of the 2-D CNN are 4-18 MB, both passes are latency-bound and the finalize pass reads
    // every row (1024 rows: 7.6 us per layer in the rocprofv3 split of GPU call D)
    return (int)(g > 256 ? 256 : (g < 1 ? 1 : g));
}

extern "C" int stx_bn_stats(const float* z, float* partials, long long nvox, int C, int groups, void* stream) {
    stx_begin();
    STX_REQUIRE(z && partials && nvox > 0 && groups >= 1 && groups <= 65535, "bn_stats: bad args");
    STX_REQUIRE(C >= 4 && C % 4 == 0 && C / 4 <= BN_THREADS, "bn_stats: C=%d unsupported (a multiple of 4, at most 1024)", C);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(stx_bn_stats_rows(nvox, C), groups), dim3(BN_THREADS), 0, (hipStream_t)stream, z,
                       partials, (size_t)nvox, C);
    return stx_check_launch("bn_stats");
}

extern "C" int stx_bn_apply(const float* z1, const float* scale1, const float* shift1, const float* z2,
                            const float* scale2, const float* shift2, float* out, long long nvox, int C, int relu,
                            int groups, void* stream) {
    stx_begin();
    STX_REQUIRE(z1 && scale1 && shift1 && out && nvox > 0 && C > 0 && C % 4 == 0, "bn_apply: bad args (C=%d)", C);
    STX_REQUIRE(!scale2 || (z2 && shift2), "bn_apply: second affine needs z2 and shift2");
    STX_REQUIRE(groups >= 1 && groups <= 65535, "bn_apply: groups=%d", groups);
    const size_t nquads = (size_t)nvox * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_grid(nquads), groups), dim3(BN_THREADS), 0, (hipStream_t)stream, z1, scale1,
                       shift1, z2, scale2, shift2, out, nquads, C / 4, relu);
    return stx_check_launch("bn_apply");
}

extern "C" int stx_bn_bwd_reduce2(const float* gy, const float* y, const float* z1, const float* mean1,
                                  const float* invstd1, const float* z2, const float* mean2, const float* invstd2,
                                  const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                                  float* partials, float* sums, long long nvox, int C, int relu, int groups, void* stream) {
    stx_begin();
    STX_REQUIRE(gy && z1 && mean1 && invstd1 && partials && sums && nvox > 0, "bn_bwd_reduce: null operand");
    STX_REQUIRE(groups >= 1 && groups <= 65535, "bn_bwd_reduce: groups=%d", groups);
    STX_REQUIRE(C >= 4 && C % 4 == 0 && C / 4 <= BN_THREADS, "bn_bwd_reduce: C=%d unsupported (a multiple of 4, at most 1024)", C);
    STX_REQUIRE(!relu || y || (scale1 && shift1 && (!(z2 && mean2) || (scale2 && shift2))),
                "bn_bwd_reduce: the relu mask needs y or the forward pass's scale / shift vectors");
    STX_REQUIRE(relu != 2 || !y, "bn_bwd_reduce: Mish (activation code 2) differentiates the pre-activation value: pass y = NULL and the scale / shift vectors");
    hipStream_t st = (hipStream_t)stream;
    // small activations (the 2-D CNN's: <= 4.5 M elements): a quarter of the workgroups, so that the column-sum pass reads
    // 256 partial rows instead of 1024 (both passes are latency-bound there)
    const int nblk = ((long long)nvox * C <= (9ll << 19)) ? BN_RED_BLOCKS / 4 : BN_RED_BLOCKS;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk, groups), dim3(BN_THREADS), 0, st, gy, y, z1, mean1, invstd1,
                       z2, mean2, invstd2, scale1, shift1, scale2, shift2, partials, (size_t)nvox, C, relu);
    int rc = stx_check_launch("bn_bwd_reduce");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_colsum_kernel, dim3(3 * C), dim3(BN_THREADS), 0, st, partials, nblk, 3 * C, sums, groups);
    return stx_check_launch("bn_colsum");
}

extern "C" int stx_bn_bwd_apply2(const float* gy, const float* y, const float* z1, const float* mean1,
                                 const float* invstd1, const float* gamma1, const float* z2, const float* mean2,
                                 const float* invstd2, const float* gamma2, const float* scale1, const float* shift1,
                                 const float* scale2, const float* shift2, const float* sums, float* dz1, float* dz2,
                                 float* gout, long long nvox, int C, int relu, int groups, void* stream) {
    stx_begin();
    STX_REQUIRE(gy && z1 && mean1 && invstd1 && sums && dz1 && nvox > 0 && C % 4 == 0, "bn_bwd_apply: bad args");
    STX_REQUIRE(groups >= 1 && groups <= 65535, "bn_bwd_apply: groups=%d", groups);
    STX_REQUIRE(!relu || y || (scale1 && shift1 && (!(z2 && mean2 && dz2) || (scale2 && shift2))),
                "bn_bwd_apply: the relu mask needs y or the forward pass's scale / shift vectors");
    STX_REQUIRE(relu != 2 || !y, "bn_bwd_apply: Mish (activation code 2) differentiates the pre-activation value: pass y = NULL and the scale / shift vectors");
    const size_t nquads = (size_t)nvox * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bn_grid(nquads), groups), dim3(BN_THREADS), 0, (hipStream_t)stream, gy, y, z1,
                       mean1, invstd1, gamma1, z2, mean2, invstd2, gamma2, scale1, shift1, scale2, shift2, sums, dz1, dz2,
                       gout, nquads, C, relu, (float)(1.0 / (double)nvox));
    return stx_check_launch("bn_bwd_apply");
}

extern "C" int stx_bn_bwd_reduce(const float* gy, const float* y, const float* z1, const float* mean1,
                                 const float* invstd1, const float* z2, const float* mean2, const float* invstd2,
                                 float* partials, float* sums, long long nvox, int C, int relu, void* stream) {
    return stx_bn_bwd_reduce2(gy, y, z1, mean1, invstd1, z2, mean2, invstd2, nullptr, nullptr, nullptr, nullptr, partials,
                              sums, nvox, C, relu, 1, stream);
}

extern "C" int stx_bn_bwd_apply(const float* gy, const float* y, const float* z1, const float* mean1,
                                const float* invstd1, const float* gamma1, const float* z2, const float* mean2,
                                const float* invstd2, const float* gamma2, const float* sums, float* dz1, float* dz2,
                                float* gout, long long nvox, int C, int relu, void* stream) {
    return stx_bn_bwd_apply2(gy, y, z1, mean1, invstd1, gamma1, z2, mean2, invstd2, gamma2, nullptr, nullptr, nullptr,
                             nullptr, sums, dz1, dz2, gout, nvox, C, relu, 1, stream);
}
