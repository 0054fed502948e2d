// 3x3 stride-1 Conv2d (padding 1, no bias) of the 2-D feature CNN on the fp32 matrix cores, channels-last: forward and
// data gradient of the BasicBlock convolutions with 32 / 64 input channels (reference models/GwcNet/gwcnet.py:12-42,
// models/PSMNet/submodule.py:57-97, models/ACVNet/acv.py:15-40: `convbn(in, out, 3, 1, pad, 1)` at 1/2 and 1/4 resolution).
//
// Why a kernel of its own (round 6): these layers are small -- 5.1 GFLOP each for both views, 32 us at the fp32-MFMA peak
// -- and the stock library runs them at 0.36-0.53 of it (profiles/r06_conv2d_miopen_probe_callK.jsonl: 64 -> 64 at 144 x 240,
// B = 2: forward 68 us, data gradient 65 us; 32 -> 32 at 288 x 480: 73 / 90 us); together they are 5.2 ms of the 57.9 ms
// GwcNet_GC train step.  A tile-per-workgroup kernel cannot win here (one tile = a few hundred MFMAs against a prologue of the
// same order); what the 3-D march kernel taught carries over:
//   * a workgroup is persistent (one per CU) and keeps the WEIGHTS of its 16-output-channel slice in LDS (9 x Cin x 16 fp32
//     = 36.9 KB at Cin = 64): both MFMA operands come from LDS, the prologue is paid once per launch;
//   * it walks a contiguous run of (slice, 8 x 16-voxel tile) units; the halo tile of unit u + 1 is in flight (global ->
//     registers) during the first half of unit u's MFMA loop and written to the other LDS buffer in its middle, the finished
//     accumulators of unit u - 1 leave during the first steps of unit u: one barrier per unit, no exposed memory latency;
//   * 16-channel output slices (v_mfma_f32_16x16x4_f32, the same 64 FLOP / clk / SIMD as the 32 x 32 x 2 form) instead of 32:
//     540 tiles x 4 slices = 2160 units over 256 workgroups = 8.44 -> 9 rounds (0.94); with 32-channel slices 4.22 -> 5 (0.84).
// GEMM view: M = voxels (a wave owns two image rows of 16 columns = two 16-row MFMA tiles), N = 16 output channels,
// K = 9 taps x Cin.  LDS: halo tile [10][18][Cin + 8] (the padding spreads the lane = (voxel, 4-channel group) operand reads over
// the banks), double buffered; weights [tap][Cin / 16][lane][4] in MFMA operand order, gathered by the kernel itself from the
// parameter's channels_last storage (forward and flipped / transposed for the data gradient: no packing launches).
// Exact fp32: every output is one k-ordered fmaf chain of 9 x Cin terms, like the 3-D kernels' accumulators.
// Optional epilogue: per-channel sum / sum of squares of the raw output for train-mode BatchNorm (one row per workgroup;
// with `groups` views in the batch the grid is split so that no workgroup's run crosses a view: per-view statistics).
#include "stx_common.h"
#include <stdlib.h>

namespace {

constexpr int C2_THREADS = 256;
constexpr int C2_TH = 8, C2_TW = 16, C2_EH = C2_TH + 2, C2_EW = C2_TW + 2, C2_NV = C2_EH * C2_EW;

struct Conv2dArgs {
    const float* x;      // [B][H][W][Cin]
    const float* w;      // the layer's parameter in channels_last storage: [Co_w][3][3][Ci_w]
    float* out;          // [B][H][W][Cout]
    float* stats;        // [gridDim.x][2][Cout] partial sums of the raw output, or null
    int B, H, W, Cin, Cout;
    int nHt, nWt, groups;
    int dgrad;           // 0: Cin = Ci_w, Cout = Co_w (forward); 1: Cin = Co_w, Cout = Ci_w, flipped taps (data gradient)
};

__device__ __forceinline__ f32x4 c2_zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// ABL (profiling only, STX_C2_ABLATE): bit 0 no global loads of the next tile, bit 1 no LDS writes of it, bit 2 no output
// stores, bit 3 no per-unit barrier (results are wrong with any of them)
// PAD: floats between two voxels of a halo buffer.  8, not the 3-D kernels' 4: by the counters (profiles/r06_conv2d_lds_pad_callV.txt)
// the lane = (voxel, 4-channel group) ds_read_b128 of this kernel spends 38 % of its LDS cycles in bank conflicts at 4 and 12, 8 % at 8
// (SQ_LDS_BANK_CONFLICT 2.91 M -> 0.42 M per launch); the launch time does not move (56.8 -> 57.2 us: the loop is not LDS-bound).
template <int KQ, int ABL = 0, int PAD = 8>       // Cin = 16 * KQ
__global__ __launch_bounds__(C2_THREADS) void conv2d_march_kernel(Conv2dArgs a) {
    constexpr int CIN = 16 * KQ, VS = CIN + PAD, F4 = CIN / 4;
    constexpr int WFLOATS = 9 * KQ * 256;                            // floats of one slice's packed weights
    constexpr int NST = (C2_NV * F4 + C2_THREADS - 1) / C2_THREADS;  // staging float4 per thread
    // floats per halo buffer: room for every staging slot of every thread (NST * 256 / F4 = 192 voxels, 180 of them real), so
    // that the LDS writes inside the MFMA loop need no predicate (a predicated store is a branch: it would cut the loop's
    // scheduling region)
    constexpr int SLOT = (NST * C2_THREADS / F4) * VS;
    constexpr int NSTEP = 9 * KQ;                                    // (tap, 16-channel chunk) steps per unit
    STX_DYN_SMEM(smem);
    float* wl = reinterpret_cast<float*>(smem);                      // [9][KQ][64][4]
    float* bufs = wl + WFLOATS;                                      // [2][SLOT]
    float* ssum = bufs + 2 * SLOT;                                   // [2][Cout] statistics of this workgroup

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kg = lane >> 4;

    // work list: `groups` equal parts of the batch, each walked by its own share of the grid; inside a part the units are
    // (slice, tile) with the tile running fastest
    const int wgs_g = gridDim.x / a.groups;
    const int grp = blockIdx.x / wgs_g, wl_id = blockIdx.x - grp * wgs_g;
    const int tiles_img = a.nHt * a.nWt;
    const int tiles_g = (a.B / a.groups) * tiles_img;
    const int nsl = a.Cout / 16;
    const long long units = (long long)tiles_g * nsl;
    const bool idle = grp >= a.groups;                               // (grid not a multiple of groups: the surplus idles)
    int u = idle ? 0 : __builtin_amdgcn_readfirstlane((int)(units * wl_id / wgs_g));
    const int u_end = idle ? 0 : __builtin_amdgcn_readfirstlane((int)(units * (wl_id + 1) / wgs_g));

    if (a.stats) {
        for (int e = tid; e < 2 * a.Cout; e += C2_THREADS) ssum[e] = 0.f;
    }

    // staging map of this thread: element e = tid + 256 k -> (halo voxel, float4 of its channels)
    unsigned rel[NST];
    int crd[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int e = tid + k * C2_THREADS;
        const int v = e / F4, f = e - v * F4;
        const int hy = v / C2_EW, wx = v - hy * C2_EW;
        crd[k] = e < C2_NV * F4 ? (hy << 8 | wx) : -1;
        rel[k] = (unsigned)(((hy * a.W + wx) * CIN + 4 * f) * 4);
    }
    float4 stg[NST];
    auto decode = [&](int uu, int& slice, int& b, int& oh0, int& ow0) {
        slice = uu / tiles_g;
        const int t = uu - slice * tiles_g;
        const int bl = t / tiles_img, r = t - bl * tiles_img;
        const int ht = r / a.nWt, wt = r - ht * a.nWt;
        b = grp * (a.B / a.groups) + bl;
        oh0 = ht * C2_TH; ow0 = wt * C2_TW;
    };
    auto load_tile = [&](int b, int oh0, int ow0) {
        // descriptor base = the halo tile's origin voxel (may lie in front of the image: only in-range offsets are used)
        const long long org = (long long)(oh0 - 1) * a.W + (ow0 - 1), img = (long long)a.H * a.W;
        const stx_bufrsrc rs = stx_make_rsrc(a.x + ((long long)b * img + org) * CIN, (unsigned)((img - org) * CIN * 4));
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int gh = oh0 - 1 + (crd[k] >> 8), gw = ow0 - 1 + (crd[k] & 255);
            const bool ok = crd[k] >= 0 && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
            stg[k] = stx_buf_ld4(rs, ok ? rel[k] : STX_BUF_OOB, 0u);
        }
    };
    auto store_tile = [&](float* buf) {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = tid + k * C2_THREADS;
            const int v = e / F4, f = e - v * F4;
            if (e < C2_NV * F4) stx_st4(buf + v * VS + 4 * f, stg[k]);
        }
    };
    // The slice's weights, straight from the parameter (no packing launch): LDS [tap][s][lane = n + 16 kg][j] holds
    // Wk[tap][k = 16 s + 4 kg + j][n = 16 slice + n].  Forward: Wk[tap][k][n] = w[n][tap][k] -- a lane's four values are 16
    // contiguous bytes of the parameter.  Data gradient: Wk[tap][k][n] = w[k][8 - tap][n] (flipped taps, channels transposed) --
    // contiguous along n: read as float4 over four lanes' n, written as four dwords.
    auto load_weights = [&](int slice) {
        if (!a.dgrad) {
            for (int e = tid; e < WFLOATS / 4; e += C2_THREADS) {
                const int l = e & 63, ts = e >> 6, tap = ts / KQ, s = ts - tap * KQ;
                stx_st4(wl + 4 * e, stx_ld4(a.w + ((size_t)(16 * slice + (l & 15)) * 9 + tap) * CIN + 16 * s + 4 * (l >> 4)));
            }
        } else {
            for (int e = tid; e < CIN * 9 * 4; e += C2_THREADS) {
                const int q = e & 3, kt = e >> 2, tap = kt % 9, k = kt / 9;
                const float4 v = stx_ld4(a.w + ((size_t)k * 9 + (8 - tap)) * a.Cout + 16 * slice + 4 * q);
                float* dst = wl + ((tap * KQ + (k >> 4)) * 64 + 4 * q + 16 * ((k & 15) >> 2)) * 4 + (k & 3);
                dst[0] = v.x; dst[4] = v.y; dst[8] = v.z; dst[12] = v.w;
            }
        }
    };

    // The weights are the MFMA's A operand and the voxels its B operand (D = W x X^T): a lane ends up with FOUR CONSECUTIVE OUTPUT
    // CHANNELS (4 kg .. 4 kg + 3 of the slice) of ONE voxel (column i of the wave's row) per accumulator -- one 16-byte store per
    // image row instead of four dword stores (voxels x channels in the other operand order).
    // Finished accumulators of the previous unit and their byte offsets in its output image (bit 31 set: nothing to store --
    // the voxel contributes zeros and the store is dropped by the descriptor's range check)
    f32x4 pacc0 = c2_zero4(), pacc1 = c2_zero4();
    unsigned pvoff[2] = {STX_BUF_OOB, STX_BUF_OOB}, cvoff[2] = {STX_BUF_OOB, STX_BUF_OOB};
    stx_bufrsrc prs = stx_make_rsrc(a.out, 0u);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};   // statistics of channels 16 * slice + 4 kg + (0..3), this lane's voxels
    auto emit_prev = [&](int r) {                                    // r = 0 / 1: image row of the wave's pair
        f32x4 v = r ? pacc1 : pacc0;
        const bool ok = (int)pvoff[r] >= 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            v[c] = ok ? v[c] : 0.f;
            s1[c] += v[c];
            s2[c] = fmaf(v[c], v[c], s2[c]);
        }
        stx_buf_st4(prs, pvoff[r], 0u, make_float4(v[0], v[1], v[2], v[3]));
    };
    auto flush_stats = [&](int slice) {                              // this lane's sums -> the workgroup's table (one slice done)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) { s1[c] += __shfl_xor(s1[c], d); s2[c] += __shfl_xor(s2[c], d); }
        }
        // four waves add one after the other (fixed order: deterministic)
        for (int w = 0; w < 4; ++w) {
            if (wave == w && i == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ssum[16 * slice + 4 * kg + c] += s1[c];
                    ssum[a.Cout + 16 * slice + 4 * kg + c] += s2[c];
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
    };

    const int abase0 = ((2 * wave) * C2_EW + i) * VS + 4 * kg, abase1 = abase0 + C2_EW * VS;
    const float* wlane = wl + lane * 4;

    int slice = 0, b = 0, oh0 = 0, ow0 = 0;
    if (u < u_end) {
        decode(u, slice, b, oh0, ow0);
        load_weights(slice);
        load_tile(b, oh0, ow0);
        store_tile(bufs);
    }
    __syncthreads();
    int cur = 0;
    while (u < u_end) {
        // Everything of a unit that is not an MFMA is dealt over the steps of its MFMA loop (a wave alone on its SIMD pays for
        // every instruction it issues outside the shadow of the matrix pipe): step k < NST requests float4 k of the NEXT halo
        // tile (global -> register) and step NSTEP - NST + k writes it to the other LDS buffer (12 or 24 steps = 1.3 / 2.5 us
        // later); steps 0 and 4 store the PREVIOUS unit's two image rows; steps 8 and 9 form this unit's two store offsets.
        const bool more = u + 1 < u_end;
        int nslice = slice, nb = b, noh0 = oh0, now0 = ow0;
        if (more) decode(u + 1, nslice, nb, noh0, now0);
        const long long norg = (long long)(noh0 - 1) * a.W + (now0 - 1), img = (long long)a.H * a.W;
        // (behind the run's last unit: an empty descriptor -- the loads return zeros without touching memory)
        const stx_bufrsrc nrs = stx_make_rsrc(a.x + ((long long)nb * img + norg) * CIN, more ? (unsigned)((img - norg) * CIN * 4) : 0u);
        const stx_bufrsrc crs = stx_make_rsrc(a.out + (size_t)b * img * a.Cout, (unsigned)(img * a.Cout * 4));
        const float* pbuf = bufs + cur * SLOT;
        float* nbuf = bufs + (cur ^ 1) * SLOT;
        f32x4 acc0, acc1;
        float4 av0[3], av1[3], bv[3];                                // operands TWO steps ahead (one wave per SIMD: nobody else covers an LDS round trip)
        auto load_step = [&](int st, int slot) {
            const int tap = st / KQ, s = st - tap * KQ;
            const int kh = tap / 3, kw = tap - 3 * kh;
            const float* sl = pbuf + (kh * C2_EW + kw) * VS + 16 * s;
            av0[slot] = stx_ld4(sl + abase0);
            av1[slot] = stx_ld4(sl + abase1);
            bv[slot] = stx_ld4(wlane + st * 256);
        };
        load_step(0, 0);
        load_step(1, 1);
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            if (st + 2 < NSTEP) load_step(st + 2, (st + 2) % 3);
            const int cb = st % 3;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[cb].x, av0[cb].x, st == 0 ? c2_zero4() : acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[cb].x, av1[cb].x, st == 0 ? c2_zero4() : acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[cb].y, av0[cb].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[cb].y, av1[cb].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[cb].z, av0[cb].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[cb].z, av1[cb].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[cb].w, av0[cb].w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[cb].w, av1[cb].w, acc1, 0, 0, 0);
            if (st < NST) {                                          // float4 `st` of the next halo tile (branch-free)
                const unsigned gh = (unsigned)(noh0 - 1 + (crd[st] >> 8)), gw = (unsigned)(now0 - 1 + (crd[st] & 255));
                const bool ok = (crd[st] >= 0) & (gh < (unsigned)a.H) & (gw < (unsigned)a.W);
                if (!(ABL & 1)) stg[st] = stx_buf_ld4(nrs, ok ? rel[st] : STX_BUF_OOB, 0u);
            }
            if ((st == 0 || st == 4) && !(ABL & 4)) emit_prev(st >> 2);            // the previous unit's two image rows
            if (st == 8 || st == 9) {                                // this unit's two store offsets
                const int r = st - 8;
                const int oh = oh0 + 2 * wave + r, ow = ow0 + i;
                // (branch-free: a select between the offset and the marker is compiled into a skipped block)
                cvoff[r] = (unsigned)((((oh * a.W + ow) * a.Cout) + 16 * slice + 4 * kg) * 4) | (((oh < a.H) & (ow < a.W)) ? 0u : STX_BUF_OOB);
            }
            if (st >= NSTEP - NST) {                                 // (the other buffer is not read during this unit)
                const int k = st - (NSTEP - NST);
                const int e = tid + k * C2_THREADS;
                const int v = e / F4, f = e - v * F4;
                if (!(ABL & 2)) stx_st4(nbuf + v * VS + 4 * f, stg[k]);  // (slots behind voxel 179 are never read)
            }
            // one MFMA occupies the pipe for 32 cycles: the next step's three operand reads go out behind this step's first three
            // MFMAs (>= 5 MFMAs = 160 cycles before they are needed), everything else behind the later ones
            STX_SCHED_GROUP(0x008, 1); STX_SCHED_GROUP(0x100, 1);
            STX_SCHED_GROUP(0x008, 1); STX_SCHED_GROUP(0x100, 1);
            STX_SCHED_GROUP(0x008, 1); STX_SCHED_GROUP(0x100, 1);
            STX_SCHED_GROUP(0x008, 1); STX_SCHED_GROUP(0x002, 4);
            STX_SCHED_GROUP(0x008, 1); STX_SCHED_GROUP(0x010, 1); STX_SCHED_GROUP(0x002, 4);
            STX_SCHED_GROUP(0x008, 1); STX_SCHED_GROUP(0x010, 1); STX_SCHED_GROUP(0x002, 4);
            STX_SCHED_GROUP(0x008, 1); STX_SCHED_GROUP(0x200, 1); STX_SCHED_GROUP(0x002, 4);
            STX_SCHED_GROUP(0x008, 1);
            STX_SCHED_BARRIER();
        }
        // this unit becomes "previous"
        prs = crs;
        pvoff[0] = cvoff[0]; pvoff[1] = cvoff[1];
        pacc0 = acc0; pacc1 = acc1;
        if (!(ABL & 8)) __syncthreads();                             // next tile resident, this one's buffer free
        if (more && nslice != slice) {                               // (rare: the run crosses into the next slice)
            emit_prev(0); emit_prev(1);
            pvoff[0] = pvoff[1] = STX_BUF_OOB;
            if (a.stats) flush_stats(slice);
            load_weights(nslice);
            __syncthreads();
        }
        slice = nslice; b = nb; oh0 = noh0; ow0 = now0;
        cur ^= 1;
        ++u;
    }
    emit_prev(0); emit_prev(1);
    if (a.stats) {
        __syncthreads();
        if (u_end > 0 && !idle) flush_stats(slice);
        __syncthreads();
        for (int e = tid; e < 2 * a.Cout; e += C2_THREADS) a.stats[(size_t)blockIdx.x * 2 * a.Cout + e] = ssum[e];
    }
}

int conv2d_grid(int groups) {
    int g = 256;
    g -= g % groups;
    return g < groups ? groups : g;
}

}  // namespace

extern "C" int stx_conv2d_supported(int Cin, int Cout) {
    stx_begin();
    return (Cin == 32 || Cin == 64) && Cout >= 16 && Cout % 16 == 0 && Cout <= 256;
}

extern "C" long long stx_conv2d_stat_rows(int groups) {
    stx_begin();
    return groups >= 1 && groups <= 256 ? conv2d_grid(groups) : 0;
}

extern "C" int stx_conv2d_fwd(const float* x, const float* w, float* out, float* stats, int B, int H, int W, int Cin, int Cout,
                              int dgrad, int groups, void* stream) {
    stx_begin();
    STX_REQUIRE(x && w && out && B > 0 && H > 0 && W > 0 && (dgrad == 0 || dgrad == 1), "conv2d_fwd: bad args");
    STX_REQUIRE(stx_conv2d_supported(Cin, Cout), "conv2d_fwd: %d -> %d channels unsupported (Cin 32 or 64, Cout a multiple of 16)", Cin, Cout);
    STX_REQUIRE(groups >= 1 && groups <= 256 && B % groups == 0, "conv2d_fwd: batch %d does not split into %d groups", B, groups);
    STX_REQUIRE((long long)(H + 2) * W * (Cin > Cout ? Cin : Cout) * 4 < (1ll << 31), "conv2d_fwd: image too large for one descriptor");
    STX_REQUIRE(H < 32768 && W < 32768, "conv2d_fwd: image too large");
    Conv2dArgs a;
    a.x = x; a.w = w; a.out = out; a.stats = stats; a.dgrad = dgrad;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.nHt = stx_cdiv(H, C2_TH); a.nWt = stx_cdiv(W, C2_TW); a.groups = groups;
    const int grid = conv2d_grid(groups);
    static const int pad = getenv("STX_C2_PAD") ? atoi(getenv("STX_C2_PAD")) : 8;            // (profiling only: 4, 8 or 12)
    const size_t lds = ((size_t)9 * Cin * 16 + (size_t)2 * 192 * (Cin + pad) + (size_t)2 * Cout) * 4;   // (192: SLOT of the kernel)
    void (*k)(Conv2dArgs) = Cin == 64 ? conv2d_march_kernel<4> : conv2d_march_kernel<2>;
    if (pad == 4) k = Cin == 64 ? conv2d_march_kernel<4, 0, 4> : conv2d_march_kernel<2, 0, 4>;
    if (pad == 12) k = Cin == 64 ? conv2d_march_kernel<4, 0, 12> : conv2d_march_kernel<2, 0, 12>;
    static const int abl = getenv("STX_C2_ABLATE") ? atoi(getenv("STX_C2_ABLATE")) : 0;      // (profiling only)
    if (abl && Cin == 64 && pad == 8) {
        k = abl == 1 ? conv2d_march_kernel<4, 1> : abl == 2 ? conv2d_march_kernel<4, 2> : abl == 3 ? conv2d_march_kernel<4, 3>
          : abl == 4 ? conv2d_march_kernel<4, 4> : abl == 7 ? conv2d_march_kernel<4, 7> : abl == 8 ? conv2d_march_kernel<4, 8>
          : conv2d_march_kernel<4, 15>;
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    }
    // (set on every launch, like the 3-D kernels: the attribute belongs to the current device's copy of the function)
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)) != hipSuccess)
        return stx_set_error(STX_ERR_LAUNCH, "conv2d_fwd: %zu B of LDS refused", lds);
    hipLaunchKernelGGL(k, dim3(grid), dim3(C2_THREADS), lds, (hipStream_t)stream, a);
    return stx_check_launch("conv2d_fwd");
}
