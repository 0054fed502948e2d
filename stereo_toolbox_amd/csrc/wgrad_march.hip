// Weight gradient of the 3x3x3 stride-1 convolutions, "march" form (round 4).
//
//   G[tap][cf][cc] = sum_o x[o + tap - 1][cf] * gy[o][cc]          (autograd of reference GwcNet/gwcnet.py:68-153's Conv3d layers)
//
// MFMA view as in conv3d.hip's first weight-gradient kernel: D[32 cf][32 cc] += A[cf][k] * B[k][cc] with the GEMM-K axis
// running over output voxels, one v_mfma_f32_32x32x2_f32 per voxel PAIR and tap.  What that kernel left on the table
// (round-3 counters: matrix pipe 0.74 busy, 1.54 LDS instructions and 3.4 scalar instructions per MFMA, 2.7 x the
// algorithmic HBM traffic) and what changes here:
//
//   * Operands K-CONTIGUOUS.  Tiles are staged transposed, [channel][voxel]: a lane (channel i, K half) reads FOUR
//     consecutive voxels of its channel with one ds_read_b128, i.e. the operands of four MFMAs (the K order is free as long
//     as A and B agree: MFMA j of a group pairs voxels 8q + j and 8q + 4 + j).  The three kw taps of a (kd, kh) tap row
//     read the same row of x shifted by one voxel: two aligned b128 reads (8 voxels) hold all three windows, picked by
//     register index.  One gy read + two x reads feed 12 MFMAs: 0.25 LDS instructions per MFMA.  Channel pitches are
//     = 4 (mod 64) dwords: the b128 lane groups hit 16 distinct 4-bank slots (guide: LDS banking table).
//   * Every wave runs the SAME straight-line code, no exec masks: wave w owns tap row w = (kd, kh) of the first eight rows
//     (3 taps, all voxel groups); the ninth row (kd = kh = 2) is split over the waves by voxel group (wave w: group w) into
//     a second accumulator set that is summed through LDS once, at the end.  27 taps x 32 pairs = 864 MFMAs per step,
//     108 per wave, balanced.
//   * A D-MARCH with rolling windows.  A workgroup walks the planes of a (4 x 16)-voxel column: gy plane o is multiplied
//     with x planes o-1, o, o+1 (kd = 0, 1, 2), which stay in LDS for three steps: every x plane is staged ONCE per column
//     (the old kernel re-staged three planes per output plane), the halo costs 6/4 x 18/16 in H / W only.  Four x buffers
//     and two gy buffers: the planes of step o+1 are requested before the MFMA loop of step o and written to LDS behind
//     it (in the middle of the MFMA loop of step o, whose planes live in other buffers) -- one barrier per step.
//   * Work is cut at STEP granularity: the launch's (columns x planes) steps are dealt in equal contiguous runs to the
//     workgroups; a run that starts inside a column re-builds the window (3 planes) there.
//
// Slab layout and the final deterministic reduction are the first kernel's ([pair][chunk][27][32][32],
// conv3d_wgrad_reduce4_kernel).  Roofline: MFMA fp32; algorithmic FLOPs = 2 * voxels * 27 * CF * CC.
#include "stx_common.h"

namespace {

constexpr int WM_TH = 4, WM_TW = 16;              // coarse voxels per step and workgroup: 8 groups of 8 along W
constexpr int WM_NW = 8, WM_THR = WM_NW * 64;
constexpr int WM_EH = WM_TH + 2, WM_EW = WM_TW + 2;
constexpr int WM_EWP = 20;                        // x tile row pitch (dwords per channel): 18 -> 20 (16-byte aligned rows, the 8-voxel reads end inside)
constexpr int WM_CHP = 132;                       // x channel pitch: 6 * 20 = 120 -> 132 = 4 (mod 64)
constexpr int WM_XPLANE = 32 * WM_CHP;
constexpr int WM_GP = 68;                         // gy channel pitch: 64 -> 68 = 4 (mod 64)
constexpr int WM_GPLANE = 32 * WM_GP;
constexpr int WM_NXB = 4, WM_NGB = 2;             // x / gy plane buffers
constexpr int WM_NPX = (WM_EH * WM_EW * 8 + WM_THR - 1) / WM_THR;      // float4 per lane and x plane (2)
static_assert(WM_TH * WM_TW * 8 == WM_THR, "one gy float4 per lane");
constexpr size_t WM_LDS_BYTES = (size_t)(WM_NXB * WM_XPLANE + WM_NGB * WM_GPLANE) * 4;       // 85 KiB (>= 8 x 4 KiB for the final sum)

struct WmArgs {
    const float* x;     // [B][D][H][W][CF]
    const float* gy;    // [B][D][H][W][CC]
    float* slab;
    int B, D, H, W, CF, CC;
    int nHt, nWt, ncols;
    long long nsteps;   // ncols * D
};

__device__ __forceinline__ f32x16 wm_zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

__global__ __launch_bounds__(WM_THR) void conv3d_wgrad_march_kernel(WmArgs a) {
    STX_DYN_SMEM(smem);
    float* xl = reinterpret_cast<float*>(smem);              // [4][32 ch][CHP]
    float* gl = xl + WM_NXB * WM_XPLANE;                     // [2][32 ch][GP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, half = lane >> 5;
    const int ncf = a.CF / 32;
    const int cfb = blockIdx.y % ncf, ccb = blockIdx.y / ncf;
    const int kdw = wave / 3, khw = wave % 3;                // the wave's tap row (rows 0..7)

    f32x16 acc[3], acc8[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) { acc[t] = wm_zero16(); acc8[t] = wm_zero16(); }

    // staging maps: element e = tid + k * 512 of a plane tile -> (voxel, float4 of the 32-channel slice); byte offsets
    // relative to the tile's origin voxel, LDS positions of the element's four channels ([channel][voxel] layout)
    unsigned xoff[WM_NPX];
    int xhw[WM_NPX], xpos[WM_NPX];
#pragma unroll
    for (int k = 0; k < WM_NPX; ++k) {
        const int e = tid + k * WM_THR, v = e >> 3, f = e & 7;
        const int hy = v / WM_EW, wx = v % WM_EW;
        xhw[k] = e < WM_EH * WM_EW * 8 ? (hy << 16 | wx) : -1;
        xoff[k] = (unsigned)(((hy * a.W + wx) * a.CF + cfb * 32 + 4 * f) * 4);
        xpos[k] = (4 * f) * WM_CHP + hy * WM_EWP + wx;
    }
    const int gv = tid >> 3, gf = tid & 7;
    const int glh = gv / WM_TW, glw = gv % WM_TW;
    const unsigned goff = (unsigned)(((glh * a.W + glw) * a.CC + ccb * 32 + 4 * gf) * 4);
    const int gpos = (4 * gf) * WM_GP + glh * WM_TW + glw;
    const long long plane = (long long)a.H * a.W;

    // operand addresses: lane (channel i, K half) reads voxels [8 q + 4 half, + 8) of its channel's tile row
    const int grp8 = (wave >> 1) * WM_EWP + 8 * (wave & 1);  // the wave's voxel group of the ninth tap row
    const int grp8g = (wave >> 1) * WM_TW + 8 * (wave & 1);
    const int xlane = i * WM_CHP + 4 * half, glane = i * WM_GP + 4 * half;

    const long long s0 = a.nsteps * blockIdx.x / gridDim.x, s1 = a.nsteps * (blockIdx.x + 1) / gridDim.x;
    long long s = s0;
    while (s < s1) {
        const int col = (int)(s / a.D), ob = (int)(s - (long long)col * a.D);
        const int oe = (int)((long long)a.D - ob < s1 - s ? a.D : ob + (s1 - s));
        int r = col;
        const int wt = r % a.nWt; r /= a.nWt;
        const int ht = r % a.nHt;
        const int b = r / a.nHt;
        const int oh0 = ht * WM_TH, ow0 = wt * WM_TW;
        // per-column validity: voxels outside the volume read zeros through the descriptor's bounds check
        unsigned xvo[WM_NPX];
#pragma unroll
        for (int k = 0; k < WM_NPX; ++k) {
            const int gh = oh0 - 1 + (xhw[k] >> 16), gw = ow0 - 1 + (xhw[k] & 0xffff);
            xvo[k] = (xhw[k] >= 0 && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W) ? xoff[k] : STX_BUF_OOB;
        }
        const unsigned gvo = (oh0 + glh < a.H && ow0 + glw < a.W) ? goff : STX_BUF_OOB;
        const long long xorg = (long long)(oh0 - 1) * a.W + (ow0 - 1), gorg = (long long)oh0 * a.W + ow0;

        float4 sx[WM_NPX], sg;
        auto load_x = [&](int p) {
            const bool in = p >= 0 && p < a.D;
            const stx_bufrsrc rs = stx_make_rsrc(a.x + (((long long)b * a.D + (in ? p : 0)) * plane + xorg) * a.CF,
                                                 in ? (unsigned)((plane - xorg) * a.CF * 4) : 0u);
#pragma unroll
            for (int k = 0; k < WM_NPX; ++k) sx[k] = stx_buf_ld4(rs, xvo[k], 0u);
        };
        auto load_g = [&](int p) {
            const bool in = p >= 0 && p < a.D;
            const stx_bufrsrc rs = stx_make_rsrc(a.gy + (((long long)b * a.D + (in ? p : 0)) * plane + gorg) * a.CC,
                                                 in ? (unsigned)((plane - gorg) * a.CC * 4) : 0u);
            sg = stx_buf_ld4(rs, gvo, 0u);
        };
        auto store_x = [&](int p) {
            float* dst = xl + (p & 3) * WM_XPLANE;
#pragma unroll
            for (int k = 0; k < WM_NPX; ++k)
                if (xhw[k] >= 0) {
                    float* q = dst + xpos[k];
                    q[0] = sx[k].x; q[WM_CHP] = sx[k].y; q[2 * WM_CHP] = sx[k].z; q[3 * WM_CHP] = sx[k].w;
                }
        };
        auto store_g = [&](int p) {
            float* q = gl + (p & 1) * WM_GPLANE + gpos;
            q[0] = sg.x; q[WM_GP] = sg.y; q[2 * WM_GP] = sg.z; q[3 * WM_GP] = sg.w;
        };

        // window of the run's first step (the previous run ended behind a barrier: nobody reads LDS any more)
        for (int p = ob - 1; p <= ob + 1; ++p) { load_x(p); store_x(p); }
        load_g(ob);
        store_g(ob);
        __syncthreads();

        for (int o = ob; o < oe; ++o) {
            const bool more = o + 1 < oe;
            // (behind the run's last step an empty descriptor: the loads return zeros without touching memory and the
            //  code in front of the MFMA loop stays branch-free)
            {
                const bool in = more && o + 2 < a.D;
                const stx_bufrsrc rs = stx_make_rsrc(a.x + (((long long)b * a.D + (in ? o + 2 : 0)) * plane + xorg) * a.CF,
                                                     in ? (unsigned)((plane - xorg) * a.CF * 4) : 0u);
#pragma unroll
                for (int k = 0; k < WM_NPX; ++k) sx[k] = stx_buf_ld4(rs, xvo[k], 0u);
                const stx_bufrsrc rg = stx_make_rsrc(a.gy + (((long long)b * a.D + (more ? o + 1 : 0)) * plane + gorg) * a.CC,
                                                     more ? (unsigned)((plane - gorg) * a.CC * 4) : 0u);
                sg = stx_buf_ld4(rg, gvo, 0u);
            }
            const float* xa = xl + ((o + kdw - 1) & 3) * WM_XPLANE + khw * WM_EWP + xlane;
            const float* x8 = xl + ((o + 1) & 3) * WM_XPLANE + 2 * WM_EWP + grp8 + xlane;
            const float* gb = gl + (o & 1) * WM_GPLANE + glane;
            // operands of voxel group q + 1 are requested while group q is multiplied
            float4 A0[2], A1[2], Bv[2];
            A0[0] = stx_ld4(xa); A1[0] = stx_ld4(xa + 4); Bv[0] = stx_ld4(gb);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int cb = q & 1, nb = cb ^ 1;
                if (q < 7) {
                    const int h = (q + 1) >> 1, w8 = (q + 1) & 1;
                    A0[nb] = stx_ld4(xa + h * WM_EWP + 8 * w8);
                    A1[nb] = stx_ld4(xa + h * WM_EWP + 8 * w8 + 4);
                    Bv[nb] = stx_ld4(gb + h * WM_TW + 8 * w8);
                } else {                                         // then the wave's group of the ninth tap row
                    A0[nb] = stx_ld4(x8);
                    A1[nb] = stx_ld4(x8 + 4);
                    Bv[nb] = stx_ld4(gb + grp8g);
                }
                STX_SCHED_BARRIER();
                const float av[6] = {A0[cb].x, A0[cb].y, A0[cb].z, A0[cb].w, A1[cb].x, A1[cb].y};
                const float bv[4] = {Bv[cb].x, Bv[cb].y, Bv[cb].z, Bv[cb].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        acc[kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j + kw], bv[j], acc[kw], 0, 0, 0);
                STX_SCHED_BARRIER();
                if (q == 4) {
                    // the planes of step o + 1 (requested before the loop, landed long ago) go to LDS here, in the shadow of the
                    // MFMAs, instead of between the last MFMA and the barrier: their buffers are not read during this step
                    if (more) { store_x(o + 2); store_g(o + 1); }
                    STX_SCHED_BARRIER();
                }
            }
            {
                const float av[6] = {A0[0].x, A0[0].y, A0[0].z, A0[0].w, A1[0].x, A1[0].y};
                const float bv[4] = {Bv[0].x, Bv[0].y, Bv[0].z, Bv[0].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        acc8[kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j + kw], bv[j], acc8[kw], 0, 0, 0);
            }
            __syncthreads();
        }
        s += oe - ob;
    }

    // partial slab [pair][chunk][tap][cf 32][cc 32]: the wave's three taps straight from its registers
    float* dst = a.slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 27 * 1024;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cf = (r & 3) + 8 * (r >> 2) + 4 * half;
            dst[(size_t)(wave * 3 + kw) * 1024 + cf * 32 + i] = acc[kw][r];
        }
    // ninth tap row: the eight waves' partial sums meet in LDS, summed in wave order (deterministic)
    float* red = reinterpret_cast<float*>(smem);             // [8 waves][1024]
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        __syncthreads();                                      // (first trip: behind the last step's barrier anyway)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cf = (r & 3) + 8 * (r >> 2) + 4 * half;
            red[wave * 1024 + cf * 32 + i] = acc8[kw][r];
        }
        __syncthreads();
        for (int e = tid; e < 1024; e += WM_THR) {
            float t = red[e];
#pragma unroll
            for (int w = 1; w < WM_NW; ++w) t += red[w * 1024 + e];
            dst[(size_t)(24 + kw) * 1024 + e] = t;
        }
    }
}

}  // namespace

// Internal interface (conv3d.hip): launches the march kernel into `slab` ([pairs][chunks][27][1024]) with `nchunks`
// workgroups per channel-block pair; the caller runs the common reduction.  Returns -1 when the shape is not served.
int stx_wgrad_march_launch(const float* x, const float* gy, float* slab, int B, int D, int H, int W, int CF, int CC,
                           int nchunks, void* stream) {
    if (CF % 32 || CC % 32 || B < 1 || D < 1 || H < 1 || W < 1) return -1;
    if ((long long)H * W * CF * 4 >= (1ll << 31) || (long long)H * W * CC * 4 >= (1ll << 31)) return -1;   // descriptor range of a plane
    WmArgs a;
    a.x = x; a.gy = gy; a.slab = slab; a.B = B; a.D = D; a.H = H; a.W = W; a.CF = CF; a.CC = CC;
    a.nHt = stx_cdiv(H, WM_TH); a.nWt = stx_cdiv(W, WM_TW);
    const long long ncols = (long long)B * a.nHt * a.nWt;
    if (ncols >= (1ll << 31)) return -1;
    a.ncols = (int)ncols;
    a.nsteps = ncols * D;
    const int npairs = (CF / 32) * (CC / 32);
    if (hipFuncSetAttribute((const void*)conv3d_wgrad_march_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)WM_LDS_BYTES) != hipSuccess)
        return stx_set_error(STX_ERR_LAUNCH, "conv3d_wgrad(march): %d bytes of dynamic LDS refused by this device", (int)WM_LDS_BYTES);
    hipLaunchKernelGGL(conv3d_wgrad_march_kernel, dim3(nchunks, npairs), dim3(WM_THR), WM_LDS_BYTES, (hipStream_t)stream, a);
    return stx_check_launch("conv3d_wgrad(march)");
}

// workgroups per channel-block pair the march launch wants for a shape (the caller sizes the slab with it)
int stx_wgrad_march_chunks(int B, int D, int H, int W, int npairs) {
    const long long nsteps = (long long)B * stx_cdiv(H, WM_TH) * stx_cdiv(W, WM_TW) * D;
    long long c = 256 / npairs;
    if (c < 1) c = 1;
    if (c > nsteps) c = nsteps;
    return (int)c;
}
