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
    // BatchNorm-backward form (conv3d_wgrad_march_kernel<true>): `gy` is the gradient BEHIND the BatchNorm + activation, the
    // gradient of the convolution's raw output is formed on the way into LDS and written to `dz` for the data-gradient launch
    const float* z;     // the convolution's raw output [B][D][H][W][CC]
    float* dz;          // [B][D][H][W][CC]
    const float *bn_scale, *bn_shift, *bn_mean, *bn_invstd, *bn_gamma, *bn_sum_g, *bn_sum_gx;   // [CC] each (gamma may be NULL)
    float bn_inv_n;
    int bn_act;         // activation code of the block: 0 or 1 (ReLU)
};

__device__ __forceinline__ f32x16 wm_zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// BN = true: the coarse operand is  dz = gamma invstd (g' - sum_g / n - xhat sum_gx / n),  g' = gy masked by the activation
// (sign of fmaf(z, scale, shift)), xhat = (z - mean) invstd -- exactly what bn_bwd_apply_kernel (bn.hip) writes -- computed per
// element between the global load of (gy, z) and the LDS write; the workgroups of the first input-channel block also store it
// (every coarse voxel passes through exactly one step of every channel-block pair).  A train-mode conv + BN block then needs
// no bn_bwd_apply launch: its three volume passes (two reads, one write) ride inside this MFMA-bound kernel.
template <bool BN>
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
    // BN form: the per-channel vectors of this lane's four coarse channels
    float4 bsc, bsh, bmu, bis, bgm, bsg, bsx;
    if constexpr (BN) {
        const int c4 = ccb * 32 + 4 * gf;
        bsc = stx_ld4(a.bn_scale + c4); bsh = stx_ld4(a.bn_shift + c4); bmu = stx_ld4(a.bn_mean + c4);
        bis = stx_ld4(a.bn_invstd + c4); bsg = stx_ld4(a.bn_sum_g + c4); bsx = stx_ld4(a.bn_sum_gx + c4);
        bgm = a.bn_gamma ? stx_ld4(a.bn_gamma + c4) : make_float4(1.f, 1.f, 1.f, 1.f);
    }

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

        float4 sx[WM_NPX], sg, sz;
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
            if constexpr (BN) {
                const stx_bufrsrc rz = stx_make_rsrc(a.z + (((long long)b * a.D + (in ? p : 0)) * plane + gorg) * a.CC,
                                                     in ? (unsigned)((plane - gorg) * a.CC * 4) : 0u);
                sz = stx_buf_ld4(rz, gvo, 0u);
            }
        };
        // (gy, z) -> dz for the plane in the staging registers (same expression, operand order and activation rule as
        // bn_bwd_apply_kernel); plane p's slice goes to global memory from the first input-channel block
        auto bn_apply_staged = [&](int p) {
            float4 g = sg;
            if (a.bn_act) {
                g.x = stx_act_bwd(g.x, fmaf(sz.x, bsc.x, bsh.x), a.bn_act); g.y = stx_act_bwd(g.y, fmaf(sz.y, bsc.y, bsh.y), a.bn_act);
                g.z = stx_act_bwd(g.z, fmaf(sz.z, bsc.z, bsh.z), a.bn_act); g.w = stx_act_bwd(g.w, fmaf(sz.w, bsc.w, bsh.w), a.bn_act);
            }
            const float n1 = a.bn_inv_n;
            float4 d;
            d.x = bgm.x * bis.x * (g.x - bsg.x * n1 - (sz.x - bmu.x) * bis.x * bsx.x * n1);
            d.y = bgm.y * bis.y * (g.y - bsg.y * n1 - (sz.y - bmu.y) * bis.y * bsx.y * n1);
            d.z = bgm.z * bis.z * (g.z - bsg.z * n1 - (sz.z - bmu.z) * bis.z * bsx.z * n1);
            d.w = bgm.w * bis.w * (g.w - bsg.w * n1 - (sz.w - bmu.w) * bis.w * bsx.w * n1);
            if (gvo == STX_BUF_OOB) d = make_float4(0.f, 0.f, 0.f, 0.f);      // (voxels outside the volume: zero operands)
            else if (cfb == 0) stx_st4(a.dz + (((long long)b * a.D + p) * plane + gorg) * a.CC + (goff >> 2), d);
            sg = d;
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
            if constexpr (BN) bn_apply_staged(p);
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
                if constexpr (BN) {
                    const stx_bufrsrc rz = stx_make_rsrc(a.z + (((long long)b * a.D + (more ? o + 1 : 0)) * plane + gorg) * a.CC,
                                                         more ? (unsigned)((plane - gorg) * a.CC * 4) : 0u);
                    sz = stx_buf_ld4(rz, gvo, 0u);
                }
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


// ---------------------------------------------------------------------------------------------------------------------
// Stride-2 form (the hourglasses' stride-2 convolutions: fine = layer input, coarse = output gradient; and their
// ConvTranspose3d layers: fine = output gradient, coarse = layer input):
//   G[tap][cf][cc] = sum_o F[2 o + tap - 1][cf] * C[o][cc]
// Same inner loop as above -- K-contiguous operands, wave w = tap row w, the ninth row split by voxel group -- on a
// PARITY-SPLIT fine tile: per channel and fine row the even columns (tile columns 0, 2, .., 32 -> slots 0..16) and the odd
// ones (1, 3, .., 31 -> slots 20..35) are stored apart, so that the fine voxels of four consecutive coarse voxels are four
// consecutive slots: kw = 0 and kw = 2 read the even slots lw .. lw+4 (one shifted by one), kw = 1 the odd slots lw .. lw+3.
// A step = one coarse plane of a (4 x 16)-voxel column against the fine planes 2o-1, 2o, 2o+1; the window rolls by two
// planes per step, the shared one stays.  Three fine planes (9 x 33 voxels each, 41 KB) + two coarse planes fill the LDS,
// so the two new planes wait in registers during the MFMA loop and are written between two barriers; they are fetched
// TRANSPOSED (one dword per lane and slot: lanes = channels, 4 slots per item) and land with one ds_write_b128 per item.
constexpr int W2_EH = 2 * WM_TH + 1, W2_RP = 36;            // fine tile rows; row pitch: 17 even slots (-> 20) + 16 odd slots
constexpr int W2_CHP = W2_EH * W2_RP;                      // 324 = 4 (mod 64)
static_assert(W2_CHP % 64 == 4, "conflict-free b128 operand reads need a channel pitch = 4 (mod 64)");
constexpr int W2_PLANE = 32 * W2_CHP;
// staging items of one fine plane: 72 full slot quads (9 rows x (4 even + 4 odd)) + 9 singles (even slot 16, the halo column)
// = 5 rounds of 16 (8 waves x 2 halves; lanes of a half-wave = the 32 channels) + one single for group 0
constexpr int W2_NFULL = W2_EH * 8, W2_ITEMS = W2_NFULL + W2_EH;
constexpr int W2_LPP = 6, W2_NPART = 4;                    // 21 slot offsets in 4 parts of 6: fetched in the first MFMA groups of a step
static_assert(W2_ITEMS == 81, "the item schedule below is written for five rounds and one left-over single");
constexpr size_t W2_LDS_BYTES = (size_t)(3 * W2_PLANE + WM_NGB * WM_GPLANE) * 4;

__device__ __forceinline__ int w2_mod3(int p) { const int r = p % 3; return r < 0 ? r + 3 : r; }

__global__ __launch_bounds__(WM_THR) void conv3d_wgrad_march_s2_kernel(WmArgs a, int Df, int Hf, int Wf, int ablate) {
    STX_DYN_SMEM(smem);
    float* xl = reinterpret_cast<float*>(smem);              // [3][32 ch][W2_CHP]
    float* gl = xl + 3 * W2_PLANE;                           // [2][32 ch][GP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, half = lane >> 5;
    const int ncf = a.CF / 32;
    const int cfb = blockIdx.y % ncf, ccb = blockIdx.y / ncf;
    const int kdw = wave / 3, khw = wave % 3;

    f32x16 acc[3], acc8[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) { acc[t] = wm_zero16(); acc8[t] = wm_zero16(); }

    // staging items of this lane: item = g + 16 k, g = wave * 2 + half; rounds k = 0..4 carry four slots, round 5 (g = 0) one
    const int g16 = wave * 2 + half;
    auto item_geom = [&](int k, int& hy, int& col, int& nv) -> int {       // -> LDS position; col = first tile column
        const int it = g16 + 16 * k;
        int par, qd;
        if (it < W2_NFULL) { hy = it >> 3; par = (it >> 2) & 1; qd = it & 3; nv = 4; }
        else { hy = it - W2_NFULL; par = 0; qd = 4; nv = it < W2_ITEMS ? 1 : 0; if (nv == 0) hy = 0; }
        col = 8 * qd + par;
        return i * W2_CHP + hy * W2_RP + par * 20 + 4 * qd;
    };
    int lpos[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { int hy, col, nv; lpos[k] = item_geom(k, hy, col, nv); }
    const int gv = tid >> 3, gf = tid & 7;
    const int glh = gv / WM_TW, glw = gv % WM_TW;
    const unsigned goff = (unsigned)(((glh * a.W + glw) * a.CC + ccb * 32 + 4 * gf) * 4);
    const int gpos = (4 * gf) * WM_GP + glh * WM_TW + glw;
    const long long cplane = (long long)a.H * a.W, fplane = (long long)Hf * Wf;
    const unsigned fplane_bytes = (unsigned)(fplane * a.CF * 4);

    const int grp8 = (2 * (wave >> 1)) * W2_RP + 8 * (wave & 1);
    const int grp8g = (wave >> 1) * WM_TW + 8 * (wave & 1);
    const int xlane = i * W2_CHP + 4 * half, glane = i * WM_GP + 4 * half;

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
        const int fy0 = 2 * oh0 - 1, fx0 = 2 * ow0 - 1;      // fine coordinates of the tile's origin
        const unsigned gvo = (oh0 + glh < a.H && ow0 + glw < a.W) ? goff : STX_BUF_OOB;
        const long long gorg = (long long)oh0 * a.W + ow0;

        // per column: the items' byte offsets inside a fine plane and their slots' validity (bit 4 k + j)
        unsigned rel[6], okbits = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            int hy, col, nv;
            item_geom(k, hy, col, nv);
            const int fy = fy0 + hy, fxq = fx0 + col;
            const bool rowok = nv > 0 && fy >= 0 && fy < Hf;
            rel[k] = (unsigned)((((rowok ? fy : 0) * Wf + fxq) * a.CF + cfb * 32 + i) * 4);
#pragma unroll
            for (int j = 0; j < (k < 5 ? 4 : 1); ++j)
                if (rowok && j < nv && fxq + 2 * j >= 0 && fxq + 2 * j < Wf) okbits |= 1u << (4 * k + j);
        }
        const unsigned cs = (unsigned)(2 * a.CF * 4);       // bytes between two slots of a quad

        float sx[2][5][4], sx5[2];
        float4 sg;
        // fine planes pA, pB -> registers (transposed fetch: lane = channel, register j = slot j of the lane's quad); the
        // descriptors are wave-uniform (one per plane, empty for a plane outside the volume), row / column validity goes
        // into the per-lane offset
        // 21 slot offsets x 2 planes = 42 loads per lane and step, issued in W2_NPART parts so that the main loop can
        // spread them over its MFMA groups (the texture path takes ~16 cycles per wave-wide load: issued in one burst
        // ahead of the MFMAs, the eight waves' 336 loads would keep the matrix pipes idle for a third of the step)
        unsigned okv = 0;
        stx_bufrsrc rsA = stx_make_rsrc(a.x, 0u), rsB = rsA;
        auto load_begin = [&](int pA, int pB) {
            okv = okbits;
            asm volatile("" : "+v"(okv));                     // (keeps the 21 offsets out of loop-invariant registers)
            const bool inA = pA >= 0 && pA < Df, inB = pB >= 0 && pB < Df;
            rsA = stx_make_rsrc(a.x + ((long long)b * Df + (inA ? pA : 0)) * fplane * a.CF, inA ? fplane_bytes : 0u);
            rsB = stx_make_rsrc(a.x + ((long long)b * Df + (inB ? pB : 0)) * fplane * a.CF, inB ? fplane_bytes : 0u);
        };
        auto load_part = [&](const int P) {                  // (P is a constant after unrolling: sx stays in registers)
#pragma unroll
            for (int v = W2_LPP * P; v < W2_LPP * P + W2_LPP; ++v) {
                if (v < 20) {
                    const int k = v >> 2, j = v & 3;
                    const unsigned vo = (okv >> v) & 1u ? rel[k] + j * cs : STX_BUF_OOB;
                    sx[0][k][j] = stx_buf_ld1(rsA, vo, 0u);
                    sx[1][k][j] = stx_buf_ld1(rsB, vo, 0u);
                } else if (v == 20) {
                    const unsigned vo5 = (okv >> 20) & 1u ? rel[5] : STX_BUF_OOB;
                    sx5[0] = stx_buf_ld1(rsA, vo5, 0u);
                    sx5[1] = stx_buf_ld1(rsB, vo5, 0u);
                }
            }
        };
        auto load_planes = [&](int pA, int pB) {
            load_begin(pA, pB);
#pragma unroll
            for (int part = 0; part < W2_NPART; ++part) load_part(part);
        };
        auto store_planes = [&](int pA, int pB, bool two) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) break;
                float* dstp = xl + w2_mod3(u ? pB : pA) * W2_PLANE;
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    stx_st4(dstp + lpos[k], make_float4(sx[u][k][0], sx[u][k][1], sx[u][k][2], sx[u][k][3]));
                if (g16 == 0) dstp[lpos[5]] = sx5[u];
            }
        };
        auto load_g = [&](int p, bool on) {
            const bool in = on && p >= 0 && p < a.D;
            const stx_bufrsrc rs = stx_make_rsrc(a.gy + (((long long)b * a.D + (in ? p : 0)) * cplane + gorg) * a.CC,
                                                 in ? (unsigned)((cplane - gorg) * a.CC * 4) : 0u);
            sg = stx_buf_ld4(rs, gvo, 0u);
        };
        auto store_g = [&](int p) {
            float* q = gl + (p & 1) * WM_GPLANE + gpos;
            q[0] = sg.x; q[WM_GP] = sg.y; q[2 * WM_GP] = sg.z; q[3 * WM_GP] = sg.w;
        };

        // window of the run's first step: fine planes 2 ob - 1, 2 ob, 2 ob + 1 (the previous run ended behind a barrier)
        load_planes(2 * ob - 1, 2 * ob);
        store_planes(2 * ob - 1, 2 * ob, true);
        load_planes(2 * ob + 1, 0);
        store_planes(2 * ob + 1, 0, false);
        load_g(ob, true);
        store_g(ob);
        __syncthreads();

        for (int o = ob; o < oe; ++o) {
            const bool more = o + 1 < oe;
            // the two new fine planes and the coarse plane of step o + 1: fetched during this step's MFMA loop
            load_begin(more ? 2 * o + 2 : -1, more ? 2 * o + 3 : -1);
            const float* xa = xl + w2_mod3(2 * o - 1 + kdw) * W2_PLANE + khw * W2_RP + xlane;
            const float* x8 = xl + w2_mod3(2 * o + 1) * W2_PLANE + 2 * W2_RP + grp8 + xlane;
            const float* gb = gl + (o & 1) * WM_GPLANE + glane;
            float4 E0[2], O0[2], Bv[2];
            float E1[2];
            E0[0] = stx_ld4(xa); E1[0] = xa[4]; O0[0] = stx_ld4(xa + 20); Bv[0] = stx_ld4(gb);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int cb = q & 1, nb = cb ^ 1;
                if (q < 7) {
                    const int h = (q + 1) >> 1, w8 = (q + 1) & 1;
                    const float* p = xa + (2 * h) * W2_RP + 8 * w8;
                    E0[nb] = stx_ld4(p); E1[nb] = p[4]; O0[nb] = stx_ld4(p + 20);
                    Bv[nb] = stx_ld4(gb + h * WM_TW + 8 * w8);
                } else {
                    E0[nb] = stx_ld4(x8); E1[nb] = x8[4]; O0[nb] = stx_ld4(x8 + 20);
                    Bv[nb] = stx_ld4(gb + grp8g);
                }
                if (ablate != 1) { if (q < W2_NPART) load_part(q); else if (q == W2_NPART) load_g(o + 1, more); }
                STX_SCHED_BARRIER();
                const float ev[5] = {E0[cb].x, E0[cb].y, E0[cb].z, E0[cb].w, E1[cb]};
                const float od[4] = {O0[cb].x, O0[cb].y, O0[cb].z, O0[cb].w};
                const float bv[4] = {Bv[cb].x, Bv[cb].y, Bv[cb].z, Bv[cb].w};
                if (ablate != 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ev[j], bv[j], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(od[j], bv[j], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ev[j + 1], bv[j], acc[2], 0, 0, 0);
                }
                }
                STX_SCHED_BARRIER();
            }
            {
                const float ev[5] = {E0[0].x, E0[0].y, E0[0].z, E0[0].w, E1[0]};
                const float od[4] = {O0[0].x, O0[0].y, O0[0].z, O0[0].w};
                const float bv[4] = {Bv[0].x, Bv[0].y, Bv[0].z, Bv[0].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc8[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ev[j], bv[j], acc8[0], 0, 0, 0);
                    acc8[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(od[j], bv[j], acc8[1], 0, 0, 0);
                    acc8[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ev[j + 1], bv[j], acc8[2], 0, 0, 0);
                }
            }
            __syncthreads();                                  // every wave is done with the planes 2o-1 and 2o
            if (more && ablate != 3) { store_planes(2 * o + 2, 2 * o + 3, true); store_g(o + 1); }
            __syncthreads();
        }
        s += oe - ob;
    }

    float* dst = a.slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 27 * 1024;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cf = (r & 3) + 8 * (r >> 2) + 4 * half;
            dst[(size_t)(wave * 3 + kw) * 1024 + cf * 32 + i] = acc[kw][r];
        }
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        __syncthreads();
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
struct StxWgradBn {        // the BatchNorm-backward operands of stx_conv3d_wgrad_bn (conv3d.hip fills it)
    const float* z; float* dz;
    const float *scale, *shift, *mean, *invstd, *gamma, *sum_g, *sum_gx;
    float inv_n; int act;
};

int stx_wgrad_march_launch(const float* x, const float* gy, float* slab, int B, int D, int H, int W, int CF, int CC,
                           int nchunks, void* stream, const StxWgradBn* bn) {
    if (CF % 32 || CC % 32 || B < 1 || D < 1 || H < 1 || W < 1) return -1;
    if ((long long)H * W * CF * 4 >= (1ll << 31) || (long long)H * W * CC * 4 >= (1ll << 31)) return -1;   // descriptor range of a plane
    WmArgs a;
    a.x = x; a.gy = gy; a.slab = slab; a.B = B; a.D = D; a.H = H; a.W = W; a.CF = CF; a.CC = CC;
    a.z = nullptr; a.dz = nullptr;
    a.bn_scale = a.bn_shift = a.bn_mean = a.bn_invstd = a.bn_gamma = a.bn_sum_g = a.bn_sum_gx = nullptr;
    a.bn_inv_n = 0.f; a.bn_act = 0;
    if (bn) {
        a.z = bn->z; a.dz = bn->dz; a.bn_scale = bn->scale; a.bn_shift = bn->shift; a.bn_mean = bn->mean; a.bn_invstd = bn->invstd;
        a.bn_gamma = bn->gamma; a.bn_sum_g = bn->sum_g; a.bn_sum_gx = bn->sum_gx; a.bn_inv_n = bn->inv_n; a.bn_act = bn->act;
    }
    a.nHt = stx_cdiv(H, WM_TH); a.nWt = stx_cdiv(W, WM_TW);
    const long long ncols = (long long)B * a.nHt * a.nWt;
    if (ncols >= (1ll << 31)) return -1;
    a.ncols = (int)ncols;
    a.nsteps = ncols * D;
    const int npairs = (CF / 32) * (CC / 32);
    void (*kern)(WmArgs) = bn ? conv3d_wgrad_march_kernel<true> : conv3d_wgrad_march_kernel<false>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WM_LDS_BYTES) != hipSuccess)
        return stx_set_error(STX_ERR_LAUNCH, "conv3d_wgrad(march): %d bytes of dynamic LDS refused by this device", (int)WM_LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3(nchunks, npairs), dim3(WM_THR), WM_LDS_BYTES, (hipStream_t)stream, a);
    return stx_check_launch("conv3d_wgrad(march)");
}

// stride-2 form: fine [B][Df][Hf][Wf][CF], coarse [B][Dc][Hc][Wc][CC]; same slab / chunk convention
int stx_wgrad_march_s2_launch(const float* f, const float* c, float* slab, int B, int Df, int Hf, int Wf, int CF, int Dc, int Hc,
                              int Wc, int CC, int nchunks, void* stream) {
    if (CF % 32 || CC % 32 || B < 1 || Dc < 1 || Hc < 1 || Wc < 1) return -1;
    if ((long long)Df * Hf * Wf * CF * 4 >= (1ll << 32) || (long long)Hc * Wc * CC * 4 >= (1ll << 31)) return -1;   // (tile kernel)
    WmArgs a = {};
    a.x = f; a.gy = c; a.slab = slab; a.B = B; a.D = Dc; a.H = Hc; a.W = Wc; a.CF = CF; a.CC = CC;
    a.nHt = stx_cdiv(Hc, WM_TH); a.nWt = stx_cdiv(Wc, WM_TW);
    const long long ncols = (long long)B * a.nHt * a.nWt;
    if (ncols >= (1ll << 31)) return -1;
    a.ncols = (int)ncols;
    a.nsteps = ncols * Dc;
    const int npairs = (CF / 32) * (CC / 32);
    if (hipFuncSetAttribute((const void*)conv3d_wgrad_march_s2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)W2_LDS_BYTES) != hipSuccess)
        return stx_set_error(STX_ERR_LAUNCH, "conv3d_wgrad(march s2): %d bytes of dynamic LDS refused by this device", (int)W2_LDS_BYTES);
    hipLaunchKernelGGL(conv3d_wgrad_march_s2_kernel, dim3(nchunks, npairs), dim3(WM_THR), W2_LDS_BYTES, (hipStream_t)stream, a,
                       Df, Hf, Wf, stx_tune(STX_TUNE_WGRAD_ABLATE));     // (profiling switch: 1 no staging loads, 2 no MFMA groups, 3 no LDS writes)
    return stx_check_launch("conv3d_wgrad(march s2)");
}

// workgroups per channel-block pair the march launch wants for a shape (the caller sizes the slab with it)
int stx_wgrad_march_chunks(int B, int D, int H, int W, int npairs) {
    const long long nsteps = (long long)B * stx_cdiv(H, WM_TH) * stx_cdiv(W, WM_TW) * D;
    long long c = 256 / npairs;
    if (c < 1) c = 1;
    if (c > nsteps) c = nsteps;
    return (int)c;
}
