// Elementwise passes for gfx950: Mish, and the two IGEV-family passes (depth_to_space, FeatureAtt gate) further down.
// Mish activation: y = x * tanh(softplus(x)), softplus with torch's threshold of 20
// (reference models/PCWNet/submodule.py:11-18 `Mish`, :178-190 `FMish`; models/CFNet/submodule.py likewise).
// The PCWNet / CFNet family uses it wherever GwcNet uses ReLU.  First version: a separate
// streaming pass after the BatchNorm apply (float4 per lane, HBM-bound: read x, write y; backward reads gy and x,
// writes gx); fusing it into the BN apply / conv epilogue is the obvious next step once the family is measured.
#include "stx_common.h"

namespace {

constexpr int ACT_THREADS = 256;

// (the arithmetic lives in stx_common.h: stx_mish / stx_mish_grad -- n / (n + 2) with n = e^x (e^x + 2), one exp and one
// division instead of libm's log1pf + tanhf; GPU call O -> V: 0.19 -> 0.077 ms for a 212 MB volume)
__device__ __forceinline__ float mish_f(float x) { return stx_mish(x); }
__device__ __forceinline__ float mish_grad_f(float x) { return stx_mish_grad(x); }

__global__ __launch_bounds__(ACT_THREADS) void mish_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              size_t nquads) {
    for (size_t i = (size_t)blockIdx.x * ACT_THREADS + threadIdx.x; i < nquads; i += (size_t)gridDim.x * ACT_THREADS) {
        float4 v = stx_ld4(x + i * 4);
        v.x = mish_f(v.x); v.y = mish_f(v.y); v.z = mish_f(v.z); v.w = mish_f(v.w);
        stx_st4(y + i * 4, v);
    }
}

__global__ __launch_bounds__(ACT_THREADS) void mish_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                              float* __restrict__ gx, size_t nquads) {
    for (size_t i = (size_t)blockIdx.x * ACT_THREADS + threadIdx.x; i < nquads; i += (size_t)gridDim.x * ACT_THREADS) {
        const float4 g = stx_ld4(gy + i * 4), v = stx_ld4(x + i * 4);
        float4 o;
        o.x = g.x * mish_grad_f(v.x); o.y = g.y * mish_grad_f(v.y);
        o.z = g.z * mish_grad_f(v.z); o.w = g.w * mish_grad_f(v.w);
        stx_st4(gx + i * 4, o);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// IGEV-family cost aggregation (models/IGEVStereo/igev_stereo.py:23-100, submodule.py:228-241).
//
// depth_to_space: ConvTranspose3d(k = 4, s = 2, p = 1) is eight output-parity classes of 2 x 2 x 2 taps each
//   out[2j + p] = x[j] w[1 + p] + x[j - 1 + 2p] w[3 - 3p]                       (per axis, p = 0 / 1)
// and runs on the 3 x 3 x 3 stride-1 MFMA kernels as ONE convolution with 8 x C class-major output channels (the class's
// 8 taps in a zero-filled 27-tap set, built on the host); this pass interleaves the classes:
//   out[b][2d + pd][2h + ph][2w + pw][c] = y[b][d][h][w][(4 pd + 2 ph + pw) C + c],
// optionally as act(y * scale[col] + shift[col]) (eval-mode BatchNorm folded, applied here so that the convolution's own
// epilogue stays the raw straight-line one).  inverse = 1: space_to_depth, the backward of the same map.
__global__ __launch_bounds__(ACT_THREADS) void depth_to_space_kernel(const float* __restrict__ y, float* __restrict__ out,
                                                                    int D, int H, int W, int CQ, size_t nquads, int inverse) {
    // one float4 of the FINE tensor per item: index (b, dd, hh, ww, cq)
    for (size_t i = (size_t)blockIdx.x * ACT_THREADS + threadIdx.x; i < nquads; i += (size_t)gridDim.x * ACT_THREADS) {
        size_t r = i;
        const int cq = (int)(r % CQ); r /= CQ;
        const int ww = (int)(r % (2 * W)); r /= (2 * W);
        const int hh = (int)(r % (2 * H)); r /= (2 * H);
        const int dd = (int)(r % (2 * D));
        const size_t b = r / (2 * D);
        const int cls = ((dd & 1) << 2) | ((hh & 1) << 1) | (ww & 1);
        const size_t coarse = ((((b * D + (dd >> 1)) * H + (hh >> 1)) * W + (ww >> 1)) * 8 + cls) * CQ + cq;
        if (inverse) stx_st4(out + coarse * 4, stx_ld4(y + i * 4));
        else stx_st4(out + i * 4, stx_ld4(y + coarse * 4));
    }
}

// FeatureAtt (submodule.py:228-241): cv[b][d][h][w][c] * sigmoid(att[b][h][w][c]) -- the gate is broadcast over the
// disparity axis.  One item = (b, hw, channel quad) walking d: the gate is computed once per item, the gate's gradient
// sum_d g * cv * s (1 - s) is a sequential (deterministic) sum in registers.
__device__ __forceinline__ float gate_sigmoid(float x) { return 1.f / (1.f + stx_exp(-x)); }

__global__ __launch_bounds__(ACT_THREADS) void gate_fwd_kernel(const float* __restrict__ cv, const float* __restrict__ att,
                                                              float* __restrict__ out, int D, size_t HW, int CQ, size_t nitems) {
    for (size_t i = (size_t)blockIdx.x * ACT_THREADS + threadIdx.x; i < nitems; i += (size_t)gridDim.x * ACT_THREADS) {
        const size_t hwq = i % (HW * CQ), b = i / (HW * CQ);
        const float4 a = stx_ld4(att + i * 4);
        float4 s;
        s.x = gate_sigmoid(a.x); s.y = gate_sigmoid(a.y); s.z = gate_sigmoid(a.z); s.w = gate_sigmoid(a.w);
        const size_t base = b * D * HW * CQ + hwq;
        for (int d = 0; d < D; ++d) {
            const size_t o = (base + (size_t)d * HW * CQ) * 4;
            float4 v = stx_ld4(cv + o);
            v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
            stx_st4(out + o, v);
        }
    }
}

__global__ __launch_bounds__(ACT_THREADS) void gate_bwd_kernel(const float* __restrict__ g, const float* __restrict__ cv,
                                                              const float* __restrict__ att, float* __restrict__ gcv,
                                                              float* __restrict__ gatt, int D, size_t HW, int CQ,
                                                              size_t nitems) {
    for (size_t i = (size_t)blockIdx.x * ACT_THREADS + threadIdx.x; i < nitems; i += (size_t)gridDim.x * ACT_THREADS) {
        const size_t hwq = i % (HW * CQ), b = i / (HW * CQ);
        const float4 a = stx_ld4(att + i * 4);
        float4 s;
        s.x = gate_sigmoid(a.x); s.y = gate_sigmoid(a.y); s.z = gate_sigmoid(a.z); s.w = gate_sigmoid(a.w);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const size_t base = b * D * HW * CQ + hwq;
        for (int d = 0; d < D; ++d) {
            const size_t o = (base + (size_t)d * HW * CQ) * 4;
            const float4 gg = stx_ld4(g + o), v = stx_ld4(cv + o);
            acc.x = fmaf(gg.x, v.x, acc.x); acc.y = fmaf(gg.y, v.y, acc.y);
            acc.z = fmaf(gg.z, v.z, acc.z); acc.w = fmaf(gg.w, v.w, acc.w);
            if (gcv) stx_st4(gcv + o, make_float4(gg.x * s.x, gg.y * s.y, gg.z * s.z, gg.w * s.w));
        }
        if (gatt) stx_st4(gatt + i * 4, make_float4(acc.x * s.x * (1.f - s.x), acc.y * s.y * (1.f - s.y),
                                                     acc.z * s.z * (1.f - s.z), acc.w * s.w * (1.f - s.w)));
    }
}

int act_grid(size_t nquads) {
    const size_t g = (nquads + ACT_THREADS - 1) / ACT_THREADS;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// Channel concatenation of up to four channels-last activations [nvox][C_k] -> [nvox][sum C_k] (the `torch.cat((l2, l3, l4),
// dim=1)` of the feature extractors, gwcnet.py:59 / acv.py:48, on channels_last tensors: torch's cat takes its generic strided
// copy there, 0.23 ms for the 2 x 320 x 144 x 240 map; this is one coalesced pass) and its inverse (the backward: three dense
// gradients out of the 320-channel one instead of three strided `contiguous()` copies).  One thread per float4 of the wide side.
struct CatArgs {
    const float* in[4];
    float* out[4];
    int cq[4];           // channels / 4 per part
};
template <bool SPLIT>
__global__ __launch_bounds__(ACT_THREADS) void cat_channels_kernel(CatArgs a, const float* __restrict__ wide_in,
                                                                   float* __restrict__ wide_out, int CQ, size_t nquads) {
    for (size_t i = (size_t)blockIdx.x * ACT_THREADS + threadIdx.x; i < nquads; i += (size_t)gridDim.x * ACT_THREADS) {
        const size_t v = i / CQ;
        int q = (int)(i - v * CQ), k = 0;
        while (k < 3 && q >= a.cq[k]) { q -= a.cq[k]; ++k; }
        const size_t part = (v * a.cq[k] + q) * 4;
        if (SPLIT) stx_st4(a.out[k] + part, stx_ld4(wide_in + i * 4));
        else stx_st4(wide_out + i * 4, stx_ld4(a.in[k] + part));
    }
}

// Batched 2-D transpose out[n][c][r] = in[n][r][c] (rows x cols -> cols x rows), 64 x 64 tiles through LDS: the channels-last
// <-> channel-major re-layout of the 320-channel feature maps in front of (and, in backward, behind) the cost-volume builders,
// whose kernels take NCHW rows.  torch's `contiguous()` does this with its generic strided copy (0.10 ms per 320 x 144 x 240
// map, 0.78 TB/s); both sides of this kernel are 256-byte runs.  rows % 4 == 0 and cols % 4 == 0.
constexpr int TR_T = 64;
__global__ __launch_bounds__(ACT_THREADS) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                                int cols, int ntr, int ntc, long long tiles) {
    __shared__ float tile[TR_T][TR_T + 1];
    const int tid = threadIdx.x, q = tid & 15, l = tid >> 4;
    const size_t per = (size_t)rows * cols;
    for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int n = (int)(t / ((long long)ntr * ntc));
        const int rem = (int)(t - (long long)n * ntr * ntc);
        const int r0 = (rem / ntc) * TR_T, c0 = (rem % ntc) * TR_T;
        const float* src = in + (size_t)n * per;
        float* dst = out + (size_t)n * per;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + l + 16 * k, c = c0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rows && c < cols) v = stx_ld4(src + (size_t)r * cols + c);
            tile[l + 16 * k][4 * q + 0] = v.x; tile[l + 16 * k][4 * q + 1] = v.y;
            tile[l + 16 * k][4 * q + 2] = v.z; tile[l + 16 * k][4 * q + 3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + l + 16 * k, r = r0 + 4 * q;
            if (c < cols && r < rows)
                stx_st4(dst + (size_t)c * rows + r, make_float4(tile[4 * q + 0][l + 16 * k], tile[4 * q + 1][l + 16 * k],
                                                               tile[4 * q + 2][l + 16 * k], tile[4 * q + 3][l + 16 * k]));
        }
        __syncthreads();
    }
}

}  // namespace

// x, y: n floats, n a multiple of 4 (dense channels-last activations always are); in place (y == x) allowed
extern "C" int stx_mish_fwd(const float* x, float* y, long long n, void* stream) {
    stx_begin();
    STX_REQUIRE(x && y && n > 0 && n % 4 == 0, "mish_fwd: need n %% 4 == 0 (n=%lld)", n);
    hipLaunchKernelGGL(mish_fwd_kernel, dim3(act_grid((size_t)n / 4)), dim3(ACT_THREADS), 0, (hipStream_t)stream, x, y,
                       (size_t)n / 4);
    return stx_check_launch("mish_fwd");
}

// gx = gy * mish'(x); x is the activation INPUT (pre-activation value)
extern "C" int stx_mish_bwd(const float* gy, const float* x, float* gx, long long n, void* stream) {
    stx_begin();
    STX_REQUIRE(gy && x && gx && n > 0 && n % 4 == 0, "mish_bwd: need n %% 4 == 0 (n=%lld)", n);
    hipLaunchKernelGGL(mish_bwd_kernel, dim3(act_grid((size_t)n / 4)), dim3(ACT_THREADS), 0, (hipStream_t)stream, gy, x,
                       gx, (size_t)n / 4);
    return stx_check_launch("mish_bwd");
}

// y [B][D][H][W][8 C] (class-major channels) -> out [B][2D][2H][2W][C]; inverse != 0: the other way (out is the coarse
// tensor).  C % 4 == 0.  See depth_to_space_kernel: the interleave of ConvTranspose3d(k4, s2, p1)'s parity classes.
extern "C" int stx_depth_to_space(const float* y, float* out, int B, int D, int H, int W, int C, int inverse, void* stream) {
    stx_begin();
    STX_REQUIRE(y && out && B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "depth_to_space: bad arguments (C=%d)", C);
    const size_t nquads = (size_t)B * D * H * W * 8 * (C / 4);
    hipLaunchKernelGGL(depth_to_space_kernel, dim3(act_grid(nquads)), dim3(ACT_THREADS), 0, (hipStream_t)stream, y, out, D, H,
                       W, C / 4, nquads, inverse);
    return stx_check_launch("depth_to_space");
}

// out[n][c][r] = in[n][r][c] for n < N: [N][rows][cols] -> [N][cols][rows]; rows % 4 == 0, cols % 4 == 0
extern "C" int stx_transpose(const float* in, float* out, int N, int rows, int cols, void* stream) {
    stx_begin();
    STX_REQUIRE(in && out && N > 0 && rows > 0 && cols > 0 && rows % 4 == 0 && cols % 4 == 0,
                "transpose: bad arguments (%d x %d x %d; rows and cols multiples of 4)", N, rows, cols);
    const int ntr = (rows + TR_T - 1) / TR_T, ntc = (cols + TR_T - 1) / TR_T;
    const long long tiles = (long long)N * ntr * ntc;
    const int grid = (int)(tiles > 8192 ? 8192 : tiles);
    hipLaunchKernelGGL(transpose_kernel, dim3(grid), dim3(ACT_THREADS), 0, (hipStream_t)stream, in, out, rows, cols, ntr, ntc, tiles);
    return stx_check_launch("transpose");
}

// out[v][0 .. C0 + C1 + C2 + C3) = in0[v][..] | in1[v][..] | in2[v][..] | in3[v][..] (unused parts: NULL with C = 0); every C_k % 4 == 0
extern "C" int stx_concat_channels(const float* in0, const float* in1, const float* in2, const float* in3, int C0, int C1, int C2,
                                   int C3, float* out, long long nvox, void* stream) {
    stx_begin();
    const int C = C0 + C1 + C2 + C3;
    STX_REQUIRE(out && nvox > 0 && C > 0 && C0 >= 0 && C1 >= 0 && C2 >= 0 && C3 >= 0 && !((C0 | C1 | C2 | C3) & 3),
                "concat_channels: bad arguments (C = %d + %d + %d + %d)", C0, C1, C2, C3);
    STX_REQUIRE((in0 || !C0) && (in1 || !C1) && (in2 || !C2) && (in3 || !C3), "concat_channels: null part");
    CatArgs a;
    a.in[0] = in0; a.in[1] = in1; a.in[2] = in2; a.in[3] = in3;
    a.out[0] = a.out[1] = a.out[2] = a.out[3] = nullptr;
    a.cq[0] = C0 / 4; a.cq[1] = C1 / 4; a.cq[2] = C2 / 4; a.cq[3] = C3 / 4;
    const size_t nquads = (size_t)nvox * (C / 4);
    hipLaunchKernelGGL(cat_channels_kernel<false>, dim3(act_grid(nquads)), dim3(ACT_THREADS), 0, (hipStream_t)stream, a,
                       (const float*)nullptr, out, C / 4, nquads);
    return stx_check_launch("concat_channels");
}

// the inverse: out_k[v][..] = in[v][its channel range]; a NULL out_k skips that part
extern "C" int stx_split_channels(const float* in, float* out0, float* out1, float* out2, float* out3, int C0, int C1, int C2,
                                  int C3, long long nvox, void* stream) {
    stx_begin();
    const int C = C0 + C1 + C2 + C3;
    STX_REQUIRE(in && nvox > 0 && C > 0 && C0 >= 0 && C1 >= 0 && C2 >= 0 && C3 >= 0 && !((C0 | C1 | C2 | C3) & 3),
                "split_channels: bad arguments (C = %d + %d + %d + %d)", C0, C1, C2, C3);
    STX_REQUIRE((out0 || !C0) && (out1 || !C1) && (out2 || !C2) && (out3 || !C3), "split_channels: null part");
    CatArgs a;
    a.in[0] = a.in[1] = a.in[2] = a.in[3] = nullptr;
    a.out[0] = out0; a.out[1] = out1; a.out[2] = out2; a.out[3] = out3;
    a.cq[0] = C0 / 4; a.cq[1] = C1 / 4; a.cq[2] = C2 / 4; a.cq[3] = C3 / 4;
    const size_t nquads = (size_t)nvox * (C / 4);
    hipLaunchKernelGGL(cat_channels_kernel<true>, dim3(act_grid(nquads)), dim3(ACT_THREADS), 0, (hipStream_t)stream, a, in,
                       (float*)nullptr, C / 4, nquads);
    return stx_check_launch("split_channels");
}

// FeatureAtt gate (IGEVStereo/submodule.py:228-241): out = cv * sigmoid(att), cv / out [B][D][HW][C], att [B][HW][C]
extern "C" int stx_gate_fwd(const float* cv, const float* att, float* out, int B, int D, long long HW, int C, void* stream) {
    stx_begin();
    STX_REQUIRE(cv && att && out && B > 0 && D > 0 && HW > 0 && C > 0 && C % 4 == 0, "gate_fwd: bad arguments (C=%d)", C);
    const size_t nitems = (size_t)B * HW * (C / 4);
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(act_grid(nitems)), dim3(ACT_THREADS), 0, (hipStream_t)stream, cv, att, out, D,
                       (size_t)HW, C / 4, nitems);
    return stx_check_launch("gate_fwd");
}

// gcv = g * sigmoid(att) (may be NULL), gatt = sum_d g * cv * s (1 - s) (may be NULL); cv = the gate's INPUT volume
extern "C" int stx_gate_bwd(const float* g, const float* cv, const float* att, float* gcv, float* gatt, int B, int D,
                            long long HW, int C, void* stream) {
    stx_begin();
    STX_REQUIRE(g && cv && att && (gcv || gatt) && B > 0 && D > 0 && HW > 0 && C > 0 && C % 4 == 0,
                "gate_bwd: bad arguments (C=%d)", C);
    const size_t nitems = (size_t)B * HW * (C / 4);
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(act_grid(nitems)), dim3(ACT_THREADS), 0, (hipStream_t)stream, g, cv, att, gcv,
                       gatt, D, (size_t)HW, C / 4, nitems);
    return stx_check_launch("gate_bwd");
}
