// Sampled (cascade) cost volume of the CFNet family for gfx950: the per-pixel disparity hypotheses of a cascade stage
// turned into the stage's NDHWC input volume in one pass, and its backward.
//
// Replaces (reference, /root/reference/stereo_toolbox/models/CFNet):
//   SpatialTransformer.forward                submodule.py:306-350  (right features gathered at column w - sample with a
//                                                                   clamped index, zeroed where the un-clamped column
//                                                                   leaves the image; left features broadcast over S)
//   groupwise_correlation_4D                  submodule.py:163-169
//   cfnet.cost_volume_generator               cfnet.py:470-497      (called twice per stage: "concat" and "gwc")
//   torch.cat((gwc, concat, samples), dim=1)  cfnet.py:560-566 / 591-597
// The reference chain materialises two [B,C,S,H,W] feature stacks per call (320 + 24 channels x 16 samples at 1/4
// resolution: ~0.8 GB written and re-read per stage); here the only HBM traffic is the volume itself
// ([B][S][H][W][CTp] floats, CTp = G + 2 Cc + 1 rounded up to 8: the GEMM-K step of the convolution that consumes it,
// pad channels written as zeros) -- the feature rows of a workgroup's 64 columns are re-read from L2 for every sample.
// Roofline: HBM, algorithmic bytes = volume once (+ features once).
//
// Forward: a workgroup owns (b, s, h, 64 columns).  Lane = column for the feature reads (NCHW rows: coalesced left
// reads, near-coalesced gathers: neighbouring pixels carry neighbouring hypotheses), the four waves split the groups /
// concat channels; results meet in an LDS tile [64][CTp + 1] and leave as one contiguous run of 16-byte stores.
// Backward: a workgroup owns (b, h, 64 columns) and walks the S samples: the left-feature gradients accumulate in
// registers (no atomics), the right-feature gradients are scattered with float atomics exactly like the reference's
// gather backward (index_put with accumulate) -- run-to-run summation order is not fixed, as in the reference.
#include "stx_common.h"
#include <stdlib.h>

namespace {

constexpr int SV_TW = 64;
constexpr int SV_THREADS = 256;
constexpr int SV_MAXGPW = 10;      // groups per wave (G <= 40)

struct SvArgs {
    const float *Lg, *Rg, *Lc, *Rc, *samples;
    float* vol;
    const float* gvol;
    float *gLg, *gRg, *gLc, *gRc;
    int B, H, W, S, G, cpg, Cc, CT, CTp;
    int nsplit;                    // backward: workgroups (grid.z) sharing the groups of one (b, h, tile)
};

// column of the right image a hypothesis points at: pos = w - sample (integer-valued floats), index clamped into the
// row, `valid` false where the un-clamped position leaves [0, W-1] (submodule.py:331-346)
__device__ __forceinline__ int sv_column(int w, float sample, int W, bool& valid) {
    const float pos = (float)w - sample;
    valid = !(pos < 0.f || pos > (float)(W - 1));
    const float cl = pos < 0.f ? 0.f : (pos > (float)(W - 1) ? (float)(W - 1) : pos);
    return (int)cl;
}

// CPG > 0: channels per group as a compile-time constant -- the left features of the lane's groups stay in registers over
// the SV_SPB hypotheses a workgroup handles (GPU call O: one hypothesis per workgroup re-read both feature rows from L2
// for every one of them, 0.18 ms for the 126 MB stage-3 volume = L2-bound); CPG = 0: any group width, one pass per sample.
constexpr int SV_SPB = 4;

template <int CPG>
__global__ __launch_bounds__(SV_THREADS) void sampled_volume_fwd_kernel(SvArgs a) {
    STX_DYN_SMEM(smem);
    float* tile = reinterpret_cast<float*>(smem);               // [64][TS]
    const int TS = a.CTp + 1;
    const int tid = threadIdx.x, wl = tid & 63, wave = tid >> 6;
    const int nsb = (a.S + SV_SPB - 1) / SV_SPB;
    const int w0 = blockIdx.x * SV_TW, h = blockIdx.y, b = blockIdx.z / nsb, s0 = (blockIdx.z % nsb) * SV_SPB;
    const int HW = a.H * a.W, cpg = CPG ? CPG : a.cpg, Cg = a.G * cpg;
    const int w = w0 + wl, wc = w < a.W ? w : a.W - 1;            // (ragged last tile: clamped reads, stores masked)
    // wave-uniform row bases + 32-bit lane offsets (one batch item's features are < 2^32 bytes: checked by the host)
    const float* Lrow = a.Lg + ((size_t)b * Cg * a.H + h) * a.W;
    const float* Rrow = a.Rg + ((size_t)b * Cg * a.H + h) * a.W;
    const float inv = 1.f / (float)cpg;
    float* mine = tile + wl * TS;
    constexpr int NL = CPG ? CPG : 1;
    float lv[SV_MAXGPW][NL], lc[4];
    if (CPG && a.G) {
#pragma unroll
        for (int k = 0; k < SV_MAXGPW; ++k)
#pragma unroll
            for (int c = 0; c < NL; ++c) {
                const int g = wave + 4 * k;
                lv[k][c] = g < a.G ? Lrow[(unsigned)(g * CPG + c) * (unsigned)HW + (unsigned)wc] : 0.f;
            }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = wave + 4 * k;
        lc[k] = c < a.Cc ? a.Lc[((size_t)(b * a.Cc + c) * a.H + h) * a.W + wc] : 0.f;
    }
    const int Q = a.CTp >> 2;
    const int ncol = (a.W - w0) < SV_TW ? (a.W - w0) : SV_TW;
    const int s1 = s0 + SV_SPB < a.S ? s0 + SV_SPB : a.S;
    for (int s = s0; s < s1; ++s) {
        const int bs = b * a.S + s;
        const float sample = a.samples[((size_t)bs * a.H + h) * a.W + wc];
        bool valid;
        const int xi = sv_column(wc, sample, a.W, valid);
        if (a.G) {
            if (CPG) {
#pragma unroll
                for (int k = 0; k < SV_MAXGPW; ++k) {
                    const int g = wave + 4 * k;
                    if (g < a.G) {
                        float r[NL];
#pragma unroll
                        for (int c = 0; c < NL; ++c) r[c] = Rrow[(unsigned)(g * CPG + c) * (unsigned)HW + (unsigned)xi];
                        float acc = 0.f;
#pragma unroll
                        for (int c = 0; c < NL; ++c) acc = fmaf(lv[k][c], r[c], acc);
                        mine[g] = valid ? acc * inv : 0.f;
                    }
                }
            } else {
                for (int g = wave; g < a.G; g += 4) {
                    float acc = 0.f;
#pragma unroll 4
                    for (int c = 0; c < cpg; ++c) {
                        const unsigned o = (unsigned)(g * cpg + c) * (unsigned)HW;
                        acc = fmaf(Lrow[o + (unsigned)wc], Rrow[o + (unsigned)xi], acc);
                    }
                    mine[g] = valid ? acc * inv : 0.f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = wave + 4 * k;
            if (c < a.Cc) {
                mine[a.G + c] = lc[k];
                const float r = a.Rc[((size_t)(b * a.Cc + c) * a.H + h) * a.W + xi];
                mine[a.G + a.Cc + c] = valid ? r : 0.f;
            }
        }
        if (wave == 0) {
            mine[a.G + 2 * a.Cc] = sample;
            for (int c = a.CT; c < a.CTp; ++c) mine[c] = 0.f;
        }
        __syncthreads();
        float* out = a.vol + (((size_t)bs * a.H + h) * a.W + w0) * a.CTp;
        for (int idx = tid; idx < ncol * Q; idx += SV_THREADS) {
            const int col = idx / Q, q = idx - col * Q;
            const float* src = tile + col * TS + 4 * q;
            stx_st4(out + (size_t)idx * 4, make_float4(src[0], src[1], src[2], src[3]));
        }
        __syncthreads();                                         // the tile is rewritten by the next hypothesis
    }
}

// PRIV: the right-feature gradients of the workgroup accumulate in an LDS window of SV_XW columns ([w0 - SV_XB, w0 + 63]:
// hypotheses are non-negative disparities, so the gather column lies at or left of the pixel) with LDS atomics and are added
// to memory once at the end -- one global atomic per (channel, window column) instead of one per (channel, pixel,
// hypothesis): 16 x fewer for CFNet's stage 3 (GPU call O: 88 M global float atomics = 1.6 ms).  Columns outside the window
// fall back to global atomics.  The groups are split over `a.nsplit` workgroups (grid.z) so that the window fits the LDS.
constexpr int SV_XB = 96, SV_XW = SV_XB + SV_TW;

template <int CPG, bool PRIV, int GPW>      // GPW: groups per wave (4 GPW >= groups of the workgroup)
__global__ __launch_bounds__(SV_THREADS, GPW == 5 ? 2 : 1) void sampled_volume_bwd_kernel(SvArgs a) {
    STX_DYN_SMEM(smem);
    float* tile = reinterpret_cast<float*>(smem);               // [64][TS]: the volume gradient of (s, h, 64 columns)
    const int TS = a.CTp + 1;
    float* accw = tile + SV_TW * TS;                             // PRIV: [GH * CPG + Cc][SV_XW]
    const int tid = threadIdx.x, wl = tid & 63, wave = tid >> 6;
    const int w0 = blockIdx.x * SV_TW, h = blockIdx.y, b = blockIdx.z / a.nsplit, part = blockIdx.z % a.nsplit;
    const int GH = (a.G + a.nsplit - 1) / a.nsplit;              // groups per workgroup
    const int g_lo = part * GH, g_hi = g_lo + GH < a.G ? g_lo + GH : a.G;
    const int ncc = part == 0 ? a.Cc : 0;                        // the concat channels ride with the first part
    const int HW = a.H * a.W, Cg = a.G * CPG;
    const int w = w0 + wl, wc = w < a.W ? w : a.W - 1;
    const bool live = w < a.W;
    const int Q = a.CTp >> 2;
    const int ncol = (a.W - w0) < SV_TW ? (a.W - w0) : SV_TW;
    const float inv = 1.f / (float)CPG;
    const int x_lo = w0 - SV_XB;
    const int nacc = (GH * CPG + a.Cc) * SV_XW;
    if (PRIV)
        for (int i = tid; i < nacc; i += SV_THREADS) accw[i] = 0.f;
    float gl[GPW][CPG];                                   // d/dLg of my groups' channels
#pragma unroll
    for (int k = 0; k < GPW; ++k)
#pragma unroll
        for (int c = 0; c < CPG; ++c) gl[k][c] = 0.f;
    float glc[4] = {0.f, 0.f, 0.f, 0.f};                        // d/dLc of concat channels wave, wave+4, .. (Cc <= 16)
    const size_t rowoff = ((size_t)b * Cg * a.H + h) * a.W;      // (wave-uniform; lane offsets below are 32-bit)
    for (int s = 0; s < a.S; ++s) {
        const int bs = b * a.S + s;
        __syncthreads();                                        // the previous sample's tile is no longer read
        const float* src = a.gvol + (((size_t)bs * a.H + h) * a.W + w0) * a.CTp;
        for (int idx = tid; idx < ncol * Q; idx += SV_THREADS) {
            const int col = idx / Q, q = idx - col * Q;
            const float4 v = stx_ld4(src + (size_t)idx * 4);
            float* dst = tile + col * TS + 4 * q;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
        __syncthreads();
        const float sample = a.samples[((size_t)bs * a.H + h) * a.W + wc];
        bool valid;
        const int xi = sv_column(wc, sample, a.W, valid);
        valid = valid && live;
        const int xl = xi - x_lo;
        const bool inwin = PRIV && xl >= 0 && xl < SV_XW;
        const float* mine = tile + wl * TS;
        if (a.G) {
            const float* Lrow = a.Lg + rowoff;
            const float* Rrow = a.Rg + rowoff;
            float* gRrow = a.gRg + rowoff;
#pragma unroll
            for (int k = 0; k < GPW; ++k) {
                const int g = g_lo + wave + 4 * k;
                if (g < g_hi && valid) {                           // (invalid hypotheses carry no gradient on either side)
                    const float gv = mine[g] * inv;
#pragma unroll
                    for (int c = 0; c < CPG; ++c) {
                        const unsigned o = (unsigned)(g * CPG + c) * (unsigned)HW;
                        gl[k][c] = fmaf(gv, Rrow[o + (unsigned)xi], gl[k][c]);
                        const float v = gv * Lrow[o + (unsigned)wc];
                        if (inwin) atomicAdd(accw + ((g - g_lo) * CPG + c) * SV_XW + xl, v);
                        else atomicAdd(gRrow + (o + (unsigned)xi), v);
                    }
                }
                STX_SCHED_BARRIER();       // one group's loads in flight at a time: all ten hoisted cost 440 VGPRs
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = wave + 4 * k;
            if (c < ncc && live) {
                glc[k] += mine[a.G + c];
                if (valid) {
                    const float v = mine[a.G + a.Cc + c];
                    if (inwin) atomicAdd(accw + (GH * CPG + c) * SV_XW + xl, v);
                    else atomicAdd(a.gRc + ((size_t)(b * a.Cc + c) * a.H + h) * a.W + xi, v);
                }
            }
        }
    }
    if (PRIV) {
        __syncthreads();
        // the window goes to memory: lane = window column (coalesced), one atomic per non-zero entry (neighbouring tiles'
        // windows overlap)
        const int nrow = (g_hi - g_lo) * CPG + ncc;
        for (int i = tid; i < nrow * SV_XW; i += SV_THREADS) {
            const int r = i / SV_XW, col = i - r * SV_XW;
            const int x = x_lo + col;
            const int rl = r < (g_hi - g_lo) * CPG ? r : GH * CPG + (r - (g_hi - g_lo) * CPG);     // LDS row of output row r
            const float v = accw[rl * SV_XW + col];
            if (x >= 0 && x < a.W && v != 0.f) {
                if (r < (g_hi - g_lo) * CPG)
                    atomicAdd(a.gRg + rowoff + (size_t)(g_lo * CPG + r) * HW + x, v);
                else
                    atomicAdd(a.gRc + ((size_t)(b * a.Cc + (r - (g_hi - g_lo) * CPG)) * a.H + h) * a.W + x, v);
            }
        }
    }
    if (!live) return;
    if (a.G) {
        float* gLrow = a.gLg + rowoff;
#pragma unroll
        for (int k = 0; k < GPW; ++k) {
            const int g = g_lo + wave + 4 * k;
            if (g < g_hi) {
#pragma unroll
                for (int c = 0; c < CPG; ++c) gLrow[(unsigned)(g * CPG + c) * (unsigned)HW + (unsigned)w] = gl[k][c];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = wave + 4 * k;
        if (c < ncc) a.gLc[((size_t)(b * a.Cc + c) * a.H + h) * a.W + w] = glc[k];
    }
}

int sv_check(const SvArgs& a, const char* who) {
    STX_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.S > 0, "%s: empty shape", who);
    STX_REQUIRE(a.G >= 0 && a.Cc >= 0 && (a.G > 0 || a.Cc > 0), "%s: no channels", who);
    STX_REQUIRE(a.CTp >= a.CT && a.CTp % 4 == 0, "%s: padded channel count %d (need >= %d, multiple of 4)", who, a.CTp, a.CT);
    STX_REQUIRE((long long)a.B * a.S <= 65535 && a.H <= 65535, "%s: grid too large", who);
    STX_REQUIRE((long long)a.G * a.cpg * a.H * a.W < (1ll << 30), "%s: feature maps of a batch item exceed 4 GiB", who);
    return STX_OK;
}

}  // namespace

// vol[b][s][h][w][:] = (group-wise correlation of Lg[.., w] and Rg[.., w - samples], G channels | Lc[.., w], Cc channels |
// Rc[.., w - samples], Cc channels | samples | zero pad), hypotheses pointing outside the image contribute zeros on the
// right-feature side.  Cg = G * (channels per group); CTp >= G + 2 Cc + 1, a multiple of 4.
extern "C" int stx_sampled_volume_fwd(const float* Lg, const float* Rg, int Cg, int G, const float* Lc, const float* Rc,
                                      int Cc, const float* samples, float* vol, int B, int H, int W, int S, int CTp,
                                      void* stream) {
    stx_begin();
    STX_REQUIRE(samples && vol, "sampled_volume_fwd: null operand");
    STX_REQUIRE(G == 0 || (Lg && Rg && Cg > 0 && Cg % G == 0), "sampled_volume_fwd: %d channels are not divisible into %d groups", Cg, G);
    STX_REQUIRE(Cc == 0 || (Lc && Rc), "sampled_volume_fwd: null concat features");
    SvArgs a{};
    a.Lg = Lg; a.Rg = Rg; a.Lc = Lc; a.Rc = Rc; a.samples = samples; a.vol = vol;
    a.B = B; a.H = H; a.W = W; a.S = S; a.G = G; a.cpg = G ? Cg / G : 0; a.Cc = Cc; a.CT = G + 2 * Cc + 1; a.CTp = CTp;
    if (int rc = sv_check(a, "sampled_volume_fwd")) return rc;
    const size_t lds = (size_t)SV_TW * (CTp + 1) * 4;
    STX_REQUIRE(lds <= 64 * 1024, "sampled_volume_fwd: %d channels exceed the LDS tile", CTp);
    STX_REQUIRE(Cc <= 16, "sampled_volume_fwd: %d concat channels exceed the kernel's register tile (16)", Cc);
    const dim3 grid(stx_cdiv(W, SV_TW), H, B * stx_cdiv(S, SV_SPB));
    hipStream_t st = (hipStream_t)stream;
    if (a.cpg == 4 && G <= 4 * SV_MAXGPW)
        hipLaunchKernelGGL(sampled_volume_fwd_kernel<4>, grid, dim3(SV_THREADS), lds, st, a);
    else if (a.cpg == 8 && G <= 4 * SV_MAXGPW)
        hipLaunchKernelGGL(sampled_volume_fwd_kernel<8>, grid, dim3(SV_THREADS), lds, st, a);
    else
        hipLaunchKernelGGL(sampled_volume_fwd_kernel<0>, grid, dim3(SV_THREADS), lds, st, a);
    return stx_check_launch("sampled_volume_fwd");
}

// Gradients of the above w.r.t. the four feature maps (the samples are integer hypotheses: no gradient).  gRg / gRc are
// accumulated with atomics and are zeroed here first.
extern "C" int stx_sampled_volume_bwd(const float* gvol, const float* Lg, const float* Rg, int Cg, int G, int Cc,
                                      const float* samples, float* gLg, float* gRg, float* gLc, float* gRc, int B, int H,
                                      int W, int S, int CTp, void* stream) {
    stx_begin();
    STX_REQUIRE(gvol && samples, "sampled_volume_bwd: null operand");
    STX_REQUIRE(G == 0 || (Lg && Rg && gLg && gRg && Cg > 0 && Cg % G == 0), "sampled_volume_bwd: bad group features");
    STX_REQUIRE(Cc == 0 || (gLc && gRc), "sampled_volume_bwd: null concat gradients");
    SvArgs a{};
    a.Lg = Lg; a.Rg = Rg; a.samples = samples; a.gvol = gvol; a.gLg = gLg; a.gRg = gRg; a.gLc = gLc; a.gRc = gRc;
    a.B = B; a.H = H; a.W = W; a.S = S; a.G = G; a.cpg = G ? Cg / G : 0; a.Cc = Cc; a.CT = G + 2 * Cc + 1; a.CTp = CTp;
    if (int rc = sv_check(a, "sampled_volume_bwd")) return rc;
    STX_REQUIRE(G <= 4 * SV_MAXGPW && Cc <= 16, "sampled_volume_bwd: G = %d / Cc = %d exceed the kernel's register tiles", G, Cc);
    STX_REQUIRE(G == 0 || a.cpg == 4 || a.cpg == 8, "sampled_volume_bwd: %d channels per group unsupported (4 or 8)", a.cpg);
    const size_t lds_tile = (size_t)SV_TW * (CTp + 1) * 4;
    STX_REQUIRE(lds_tile <= 64 * 1024, "sampled_volume_bwd: %d channels exceed the LDS tile", CTp);
    hipStream_t st = (hipStream_t)stream;
    const size_t HW = (size_t)H * W;
    if (G && hipMemsetAsync(gRg, 0, (size_t)B * Cg * HW * 4, st) != hipSuccess) return stx_set_error(STX_ERR_LAUNCH, "sampled_volume_bwd: memset");
    if (Cc && hipMemsetAsync(gRc, 0, (size_t)B * Cc * HW * 4, st) != hipSuccess) return stx_set_error(STX_ERR_LAUNCH, "sampled_volume_bwd: memset");
    // LDS-privatised right gradients: split the groups over 1, 2 or 4 workgroups until the window fits (<= 80 KB keeps
    // two workgroups per CU); STX_SV_BWD_V1 = global atomics only (first version, A/B)
    const int cpg = a.cpg ? a.cpg : 4;
    int nsplit = 0;
    if (!stx_tune(STX_TUNE_SV_BWD_V1))
        for (int n = 1; n <= 4 && !nsplit; n *= 2) {
            const size_t win = ((size_t)stx_cdiv(G, n) * cpg + Cc) * SV_XW * 4;
            if (lds_tile + win <= (n < 4 ? 80 : 160) * 1024 && (long long)B * n <= 65535) nsplit = n;
        }
    const bool priv = nsplit > 0;
    a.nsplit = priv ? nsplit : 1;
    const size_t lds = lds_tile + (priv ? ((size_t)stx_cdiv(G, a.nsplit) * cpg + Cc) * SV_XW * 4 : 0);
    dim3 grid(stx_cdiv(W, SV_TW), H, B * a.nsplit);
    const bool small = stx_cdiv(G, a.nsplit) <= 20;               // <= 5 groups per wave: half the accumulator registers
#define SV_BWD(CPG_, PRIV_, GPW_)                                                                                \
    {                                                                                                            \
        if (lds > 64 * 1024)                                                                                     \
            hipFuncSetAttribute((const void*)sampled_volume_bwd_kernel<CPG_, PRIV_, GPW_>,                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        hipLaunchKernelGGL((sampled_volume_bwd_kernel<CPG_, PRIV_, GPW_>), grid, dim3(SV_THREADS), lds, st, a);  \
    }
    if (a.cpg == 8) {
        if (priv && small) SV_BWD(8, true, 5) else if (priv) SV_BWD(8, true, 10) else SV_BWD(8, false, 10)
    } else {
        if (priv && small) SV_BWD(4, true, 5) else if (priv) SV_BWD(4, true, 10) else SV_BWD(4, false, 10)
    }
#undef SV_BWD
    return stx_check_launch("sampled_volume_bwd");
}
