// Single-output-channel 3x3x3 convolution (the classifier tails `Conv3d(32, 1, 3, padding=1)` of
// reference models/GwcNet/gwcnet.py:139-153, PSMNet/stackhourglass.py:74-84, ACVNet/acv.py:122-144)
// and its weight gradient, for gfx950.
//
// With N = 1 the layer is not GEMM-shaped: putting it on the 32-wide MFMA tile wastes 31/32 of the
// matrix work ("N=1: vector-dot kernel, HBM-bound: reads 212 MB").  These are
// plain VALU kernels over the same LDS halo tile as the MFMA convolution:
//   forward : one thread per output voxel, 27 x Cin FMAs against scalar-loaded weights;
//   wgrad   : one thread per (channel, tap group) reducing over the tile's voxels, per-workgroup
//             partials -> deterministic column sum.
// Roofline: HBM (algorithmic bytes = read x once + write 1 channel).
#include "stx_common.h"

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

}  // namespace

extern "C" int stx_conv3d_c1_fwd(const float* x, const float* w, const float* residual, float* out, int B, int D,
                                 int H, int W, int Cin, void* stream) {
    stx_begin();
    STX_REQUIRE(x && w && out && B > 0 && D > 0 && H > 0 && W > 0, "conv3d_c1_fwd: bad shape");
    STX_REQUIRE(Cin % C1_CK == 0, "conv3d_c1_fwd: Cin=%d must be a multiple of %d", Cin, C1_CK);
    C1Args a;
    a.x = x; a.w = w; a.gy = nullptr; a.res = residual; a.out = out;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin;
    a.nDt = stx_cdiv(D, C1_TD); a.nHt = stx_cdiv(H, C1_TH); a.nWt = stx_cdiv(W, C1_TW);
    a.ntiles = B * a.nDt * a.nHt * a.nWt;
    hipFuncSetAttribute((const void*)conv_c1_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C1_TILE_BYTES);
    hipLaunchKernelGGL(conv_c1_fwd_kernel, dim3(a.ntiles), dim3(C1_THREADS), C1_TILE_BYTES, (hipStream_t)stream, a);
    return stx_check_launch("conv3d_c1_fwd");
}

extern "C" long long stx_conv3d_c1_wgrad_workspace_floats(int Cin) { return (long long)C1_WGRAD_BLOCKS * Cin * 27; }

// dw: [1][Cin][27] (torch layout of the Conv3d(Cin, 1, 3) weight gradient)
extern "C" int stx_conv3d_c1_wgrad(const float* x, const float* gy, float* dw, float* workspace, int B, int D, int H,
                                   int W, int Cin, void* stream) {
    stx_begin();
    STX_REQUIRE(x && gy && dw && workspace && B > 0, "conv3d_c1_wgrad: null operand");
    STX_REQUIRE(Cin % C1_CK == 0 && Cin <= 4 * C1_CK, "conv3d_c1_wgrad: Cin=%d must be 16..64 in steps of 16", Cin);
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
