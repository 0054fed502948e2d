// Conv3d / ConvTranspose3d aggregation for gfx950 as implicit GEMM on the fp32-input matrix
// cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD, 157.3 TFLOP/s chip peak).
//
// Replaces the nn.Conv3d / nn.ConvTranspose3d (+BatchNorm3d, ReLU, residual add) stacks of
// (reference, /root/reference/stereo_toolbox/models):
//   convbn_3d                         GwcNet/submodule.py:17-20, PSMNet/submodule.py:16-19,
//                                     ACVNet/submodule.py:94-97
//   dres0/dres1/classifN              GwcNet/gwcnet.py:124-153, PSMNet/stackhourglass.py:59-84,
//                                     ACVNet/acv.py:114-144
//   hourglass (conv1-6, redir1/2)     GwcNet/gwcnet.py:68-105, PSMNet/stackhourglass.py:10-50,
//                                     ACVNet/acv.py:56-93
// and their autograd backward (dgrad = the same two kernels on re-packed weights, wgrad below).
//
// Layout.  Activations are channels-last: [B][D][H][W][C] fp32, so C is the contiguous GEMM-K
// axis and one voxel of 32 channels is one 128-B line.  GEMM view of a layer:
//   M = output voxels (one MFMA row block = 32 consecutive output columns w),
//   N = Cout (32 per MFMA column block), K = taps x Cin.
// A workgroup (4 waves) stages the input halo tile of its TD x TH x 32 output voxels into LDS
// once per 32-channel K-chunk with coalesced 16-B loads ([voxel][CK+4] dwords: the +4 pad makes
// the operand reads -- ds_read_b128, lane = voxel -- bank-conflict free, guide 6/G4); every tap
// then reads its A operand from that tile.  The B operand (weights) is pre-packed on the device
// into exactly the per-lane MFMA order (stx_conv3d_pack_weight), so a wave fetches 1 KiB of
// contiguous, L2-resident weights per four MFMAs.  fp32 MFMA issues once per 64 cycles per SIMD,
// so one 16-B operand read feeds 256 cycles of matrix work: the kernels are MFMA-bound, not
// LDS- or HBM-bound (arithmetic intensity of a 32->32 layer = 216 FLOP/B).
//
// Epilogue (fused): optional per-channel affine (folded eval-mode BN), residual add, ReLU, and
// per-workgroup sum / sum-of-squares partials of the raw conv output for train-mode BatchNorm
// statistics (finalised by stx_bn_finalize; deterministic, no atomics).
#include "stx_common.h"
#include <stdlib.h>

namespace {

constexpr int CONV_THREADS = 256;

struct ConvArgs {
    const float* x;         // [B][Di][Hi][Wi][Cin]
    const float* wp;        // packed weights
    float* out;             // [B][Do][Ho][Wo][Cout]
    const float* scale;     // [Cout] or null
    const float* bias;      // [Cout] or null
    const float* residual;  // like out, or null
    float* stats;           // [nblocks][2][Cout] partial sums of raw output, or null
    int Di, Hi, Wi, Cin;
    int Do, Ho, Wo, Cout;
    int relu;               // activation code: 0 none, 1 ReLU, 2 Mish, 3 LeakyReLU (stx_act)
    int nDt, nHt, nWt;
};

// XCD-aware workgroup remap (guide T1): the dispatcher places workgroup b on XCD b % 8, each XCD has
// its own 4 MiB L2.  Give every XCD a contiguous run of tiles so that neighbouring tiles (which share
// halo voxels and, along W/H, whole input rows) hit the same L2.  Bijective for any grid size; only
// speed depends on the placement assumption.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// Staging map of a halo tile of ED x EH x EW voxels x NF4 float4 for the 256 threads of a workgroup: element
// e = tid + 256 k -> (voxel, float4 of the K chunk).  The loads go through a buffer descriptor whose base is the tile's
// ORIGIN voxel (wave-uniform; for tiles on the low edges it lies in front of the volume) and whose range ends with the
// batch item's volume: the per-lane byte offsets relative to the origin are computed once per kernel, a tile costs three
// range tests and a select per element (voxels outside the volume get the out-of-range offset and read zeros through the
// bounds check), a K chunk costs nothing (its channel offset rides in the instruction's scalar offset).  The first version
// of these kernels rebuilt a 64-bit address behind four bounds tests and a branch for every element of every chunk.
template <int ED, int EH, int EW, int NF4>
struct HaloMap {
    static constexpr int NE = ED * EH * EW * NF4;
    static constexpr int NST = (NE + CONV_THREADS - 1) / CONV_THREADS;
    unsigned rel[NST];     // byte offset relative to the tile origin
    int crd[NST];          // dz << 20 | hy << 10 | wx; -1 behind the last element
    __device__ __forceinline__ void init(int tid, int Hi, int Wi, int Cin) {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = tid + k * CONV_THREADS;
            const int v = e / NF4, f = e - v * NF4;
            const int wx = v % EW, hy = (v / EW) % EH, dz = v / (EW * EH);
            crd[k] = e < NE ? (dz << 20 | hy << 10 | wx) : -1;
            rel[k] = (unsigned)((((dz * Hi + hy) * Wi + wx) * Cin + 4 * f) * 4);
        }
    }
    __device__ __forceinline__ void offsets(int d0, int h0, int w0, int Di, int Hi, int Wi, unsigned (&vo)[NST]) const {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int gd = d0 + (crd[k] >> 20), gh = h0 + ((crd[k] >> 10) & 1023), gw = w0 + (crd[k] & 1023);
            const bool ok = crd[k] >= 0 && gd >= 0 && gd < Di && gh >= 0 && gh < Hi && gw >= 0 && gw < Wi;
            vo[k] = ok ? rel[k] : STX_BUF_OOB;
        }
    }
    static __device__ __forceinline__ stx_bufrsrc rsrc(const float* x, int b, int d0, int h0, int w0, int Di, int Hi, int Wi,
                                                       int Cin) {
        const long long org = ((long long)d0 * Hi + h0) * Wi + w0, vol = (long long)Di * Hi * Wi;
        return stx_make_rsrc(x + ((long long)b * vol + org) * Cin, (unsigned)((vol - org) * Cin * 4));
    }
};
// (host side: the descriptor range of a batch item plus the low-edge overhang of one plane must stay below 2 GiB)
static bool halo_range_ok(int Di, int Hi, int Wi, int Cin) { return (long long)(Di + 2) * Hi * Wi * Cin * 4 < (1ll << 31); }

// Fused epilogue for one 32x32 accumulator block. `vox0` = linear output voxel index of row 0 of the block (rows are voxels
// vox0 + row * rstride of ONE output row of the volume), nvalid_rows = rows that exist.
// Straight-line form (every activation code but Mish): the block's voxels go through a buffer descriptor that starts at row 0
// (wave-uniform base), one per-lane byte offset, the row's offset in the instruction's scalar offset; rows that do not exist
// get the out-of-range offset (store dropped, residual read as zero) and contribute zeros to the BN sums.  The residual rows
// are requested up front.  (The first version: a 64-bit address, a bounds branch, a residual branch and the activation switch
// per row -- for the transposed convolution, whose waves own 64-128 rows per 432-1728 MFMAs, a tenth to a quarter of the
// kernel.)
__device__ __forceinline__ void conv_epilogue_block(const ConvArgs& a, const f32x16& acc, size_t vox0, int rstride,
                                                    int nvalid_rows, int n, float sc, float bs, int lane,
                                                    float& s1, float& s2) {
    const int half = lane >> 5;
    if (a.relu < 2) {                             // (Mish / LeakyReLU: the general form below)
        const unsigned rowb = (unsigned)rstride * (unsigned)a.Cout * 4u;          // bytes between two rows of the block
        const unsigned span = 32u * rowb;                                          // (row 31 ends inside: n < Cout)
        const stx_bufrsrc ors = stx_make_rsrc(a.out + vox0 * a.Cout, span);
        const unsigned vb = (unsigned)(4 * half) * rowb + (unsigned)n * 4u;
        const bool nok = n < a.Cout;
        if (a.relu == 0 && !a.scale && !a.bias && !a.residual && nok && nvalid_rows >= 32) {
            // raw output of a whole block (every training-mode launch away from the volume's edges): two vector instructions
            // per row for the BN sums, the store takes the accumulator register itself -- in these kernels' instruction mix a
            // VALU instruction is what an epilogue costs (the transposed convolution owns 16 rows per 27 MFMAs)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row0 = (r & 3) + 8 * (r >> 2);
                s1 += acc[r];
                s2 = fmaf(acc[r], acc[r], s2);
                stx_buf_st1(ors, vb, (unsigned)row0 * rowb, acc[r]);
            }
            return;
        }
        const bool relu_on = a.relu == 1, has_res = a.residual != nullptr;
        // (an empty descriptor without a residual: its loads return zeros without touching memory)
        const stx_bufrsrc rrs = stx_make_rsrc(has_res ? a.residual + vox0 * a.Cout : a.out, has_res ? span : 0u);
#pragma unroll
        for (int g = 0; g < 4; ++g) {                                              // four rows at a time: residuals requested, then used
            float res[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row0 = q + 8 * g;
                res[q] = stx_buf_ld1(rrs, (nok && row0 + 4 * half < nvalid_rows) ? vb : STX_BUF_OOB, (unsigned)row0 * rowb);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * g + q, row0 = q + 8 * g;
                const bool ok = nok && row0 + 4 * half < nvalid_rows;
                float v = ok ? acc[r] : 0.f;
                s1 += v;
                s2 = fmaf(v, v, s2);
                v = fmaf(v, sc, bs) + res[q];
                const float vr = fmaxf(v, 0.f);
                v = relu_on ? vr : v;
                stx_buf_st1(ors, ok ? vb : STX_BUF_OOB, (unsigned)row0 * rowb, v);
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < nvalid_rows && n < a.Cout) {
            const size_t idx = (vox0 + (size_t)row * rstride) * a.Cout + n;
            float v = acc[r];
            s1 += v;
            s2 = fmaf(v, v, s2);
            v = fmaf(v, sc, bs);
            if (a.residual) v += a.residual[idx];
            v = stx_act(v, a.relu);
            a.out[idx] = v;
        }
    }
}

// Reduce per-lane stats (channel = lane&31 within column block nt) over the workgroup and write
// the partial slab row.  `red` = LDS scratch of 4*NT*32*2 floats.  NT = column blocks of the workgroup; WN = waves side by side
// along N (wave w owns blocks (w % WN) * NT/WN ... of rows group w / WN; WN = 1: every wave owns all blocks of its rows).
template <int NT, int WN = 1>
__device__ __forceinline__ void conv_write_stats(const ConvArgs& a, float (&s1)[NT / WN], float (&s2)[NT / WN], float* red,
                                                 int tid, size_t slab) {
    constexpr int NTW = NT / WN;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        s1[nt] += __shfl_xor(s1[nt], 32);
        s2[nt] += __shfl_xor(s2[nt], 32);
    }
    __syncthreads();   // LDS tile no longer read by anyone
    if (lane < 32) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            red[((wave * NTW + nt) * 32 + lane) * 2 + 0] = s1[nt];
            red[((wave * NTW + nt) * 32 + lane) * 2 + 1] = s2[nt];
        }
    }
    __syncthreads();
    if (tid < NT * 32 && tid < a.Cout) {
        const int nb = tid >> 5, wcol = nb / NTW, ntl = nb - wcol * NTW;
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int wr = 0; wr < 4 / WN; ++wr) {
            const int w = wr * WN + wcol;
            t1 += red[((w * NTW + ntl) * 32 + (tid & 31)) * 2 + 0];
            t2 += red[((w * NTW + ntl) * 32 + (tid & 31)) * 2 + 1];
        }
        a.stats[slab * 2 * a.Cout + tid] = t1;
        a.stats[slab * 2 * a.Cout + a.Cout + tid] = t2;
    }
}

// The 27 taps of one K chunk of the implicit-GEMM kernels (3x3x3).  A operands come from the LDS halo tile one tap ahead; the
// packed weights (B) come from L2 and are kept TWO taps ahead in a ring of three register sets that runs on across the chunks
// (27 = 0 mod 3: tap t always lives in set t % 3; the last two taps of a chunk prefetch the first two of the next from
// `wq_next`).  One tap is 4 MT NT NQC MFMAs = 0.2-0.4 us of matrix work per wave, an L2 round trip 0.5-0.8 us: with the weights
// only one tap ahead (rounds 1-2) every tap ended in an s_waitcnt that the two or three other waves of the SIMD could not
// always cover -- these kernels sat at 0.47-0.71 of the MFMA peak while the weights-in-LDS march kernel reached 0.84.
// NTT = column blocks of the packed weight layout (strides); NT = the blocks this wave multiplies (`wq` points at its first).
template <int S, int MT, int NT, int NQC, int EH, int EWS, int EWH, int VS, int NTT = NT>
__device__ __forceinline__ void conv_chunk_taps27(const float* tile, const int (&abase)[MT], const float* wq, const float* wq_next,
                                                  int NQ, f32x16 (&acc)[MT][NT], float4 (&bq)[3][NQC][NT]) {
    constexpr int T = 27;
    float4 av[2][NQC][MT];
    auto load_a = [&](int tap, int buf) {
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        const int toff = (S == 2) ? ((kd * EH + kh) * EWS + (kw & 1) * EWH + (kw >> 1)) * VS : ((kd * EH + kh) * EWS + kw) * VS;
#pragma unroll
        for (int q = 0; q < NQC; ++q)
#pragma unroll
            for (int m = 0; m < MT; ++m) av[buf][q][m] = stx_ld4(tile + abase[m] + toff + q * 8);
    };
    auto load_b = [&](const float* w, int tap, int slot) {
        const float* wtap = w + (size_t)tap * NQ * NTT * 256;
#pragma unroll
        for (int q = 0; q < NQC; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[slot][q][nt] = stx_ld4(wtap + (size_t)(q * NTT + nt) * 256);
    };
    load_a(0, 0);
#pragma unroll
    for (int tap = 0; tap < T; ++tap) {
        if (tap + 2 < T) load_b(wq, tap + 2, (tap + 2) % 3);
        else if (wq_next) load_b(wq_next, tap + 2 - T, (tap + 2) % 3);           // (wave-uniform)
        if (tap + 1 < T) load_a(tap + 1, (tap + 1) & 1);
        STX_SCHED_BARRIER();
        const int cb = tap & 1, sb = tap % 3;
#pragma unroll
        for (int q = 0; q < NQC; ++q)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][q][m].x, bq[sb][q][nt].x, acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][q][m].y, bq[sb][q][nt].y, acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][q][m].z, bq[sb][q][nt].z, acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][q][m].w, bq[sb][q][nt].w, acc[m][nt], 0, 0, 0);
                }
        STX_SCHED_BARRIER();
    }
}
// the ring's first two sets (taps 0 and 1 of a kernel's first chunk)
template <int NT, int NQC, int NTT = NT>
__device__ __forceinline__ void conv_ring_prologue(const float* wq, int NQ, float4 (&bq)[3][NQC][NT]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < NQC; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[t][q][nt] = stx_ld4(wq + (size_t)t * NQ * NTT * 256 + (size_t)(q * NTT + nt) * 256);
}

// ------------------------------------------------------------------------------------------
// Direct convolution, kernel KS^3 (pad KS/2), stride S.
// VPAD: LDS padding per staged voxel in floats (4 = conflict-free operand reads; 0 = the opt-in dense layout of the
// stride-2 kernel, whose 5 x 5 x 66-voxel halo tile then takes 53 KB instead of 79 KB -> three workgroups per CU)
// WN: waves side by side along N (round 5).  WN = 1: a wave owns MT rows of 32 voxels and ALL NT column blocks -- per tap NT
// packed-weight loads (L2) and one tile read per NQC x 4 x MT x NT MFMAs.  WN = 2 (64 output channels): a wave owns TWICE the rows and
// HALF the blocks: half the weight loads per MFMA, twice the (cheap, LDS) tile reads.  The tap loop in isolation
// (tools/ubench/igemm_loop, GPU call F of round 5): 0.71-0.745 of the fp32-MFMA peak with one row x two blocks at any occupancy,
// 0.80-0.85 with two rows x one block -- what the weights-in-LDS variant reaches (0.84), without its 55 KB of LDS.
template <int KS, int S, int TD, int TH, int NT, int CK, int VPAD = 4, int WN = 1>
// (occupancy hint: without it hipcc spends 132-180 VGPRs on the 1-2 column-block variants and two workgroups share a CU;
//  with it the 8- and 16-channel chunk variants take 101 without spilling and four do -- GPU call T: 64->64 L1
//  0.486 -> 0.430 ms with 8-channel chunks)
__global__ __launch_bounds__(CONV_THREADS, NT <= 2 && CK <= 16 ? (S == 2 ? 3 : 4) : 2) void conv3d_igemm_kernel(ConvArgs a) {
    constexpr int PAD = KS / 2;
    constexpr int ED = (TD - 1) * S + KS, EH = (TH - 1) * S + KS, EW = 31 * S + KS;
    constexpr int EWH = (EW + 1) / 2;
    constexpr int EWS = (S == 2) ? 2 * EWH : EW;     // LDS slots per row (S=2: even/odd de-interleaved)
    constexpr int MT = TD * TH / 4 * WN;             // rows of 32 voxels per wave
    constexpr int NTW = NT / WN;                     // column blocks per wave
    constexpr int VS = CK + VPAD;
    constexpr int NF4 = CK / 4;
    static_assert(TD * TH % 4 == 0, "tile must split over 4 waves");
    static_assert(NT % WN == 0 && 4 % WN == 0 && (WN == 1 || KS == 3), "wave grid");
    STX_DYN_SMEM(smem);
    float* tile = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar address math, no waterfall loops around the descriptors)
    const int i = lane & 31, half = lane >> 5;
    const int bid = xcd_remap(blockIdx.x, gridDim.x), b = blockIdx.y;
    const int wt = bid % a.nWt, ht = (bid / a.nWt) % a.nHt, dt = bid / (a.nWt * a.nHt);
    const int od0 = dt * TD, oh0 = ht * TH, ow0 = wt * 32;
    const int id0 = od0 * S - PAD, ih0 = oh0 * S - PAD, iw0 = ow0 * S - PAD;
    const int NQ = a.Cin / 8;
    constexpr int T = KS * KS * KS, NQC = CK / 8;

    const int wrow = wave / WN, wcol = wave % WN;    // (WN = 1: wrow = wave, wcol = 0)
    f32x16 acc[MT][NTW];
    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mtile = wrow * MT + m;
        const int td = mtile / TH, th = mtile % TH;
        abase[m] = ((td * S * EH + th * S) * EWS + i) * VS + 4 * half;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[m][nt] = zero16();
    }

    using HM = HaloMap<ED, EH, EW, NF4>;
    constexpr int NST = HM::NST;
    unsigned vo[NST];
    {
        HM hm;
        hm.init(tid, a.Hi, a.Wi, a.Cin);
        hm.offsets(id0, ih0, iw0, a.Di, a.Hi, a.Wi, vo);
    }
    const stx_bufrsrc xrs = HM::rsrc(a.x, b, id0, ih0, iw0, a.Di, a.Hi, a.Wi, a.Cin);
    // the halo tile of the NEXT K chunk is in flight (global -> registers) during this chunk's tap loop and goes to LDS
    // between the two barriers that separate the chunks: only the first chunk's memory latency is exposed
    float4 stg[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) stg[k] = stx_buf_ld4(xrs, vo[k], 0u);
    float4 bring[3][NQC][NTW];                       // packed weights, two taps ahead (3x3x3: conv_chunk_taps27)
    const float* wbase = a.wp + (size_t)lane * 4 + (size_t)wcol * NTW * 256;          // this wave's first column block
    if constexpr (KS == 3) conv_ring_prologue<NTW, NQC, NT>(wbase, NQ, bring);

    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        __syncthreads();                             // every wave is done reading the previous chunk's tile
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = tid + k * CONV_THREADS;
            const int v = e / NF4, f = e - v * NF4;
            const int wx = v % EW, hy = (v / EW) % EH, dz = v / (EW * EH);
            const int slot = (S == 2) ? (wx & 1) * EWH + (wx >> 1) : wx;
            if (e < HM::NE) stx_st4(tile + ((dz * EH + hy) * EWS + slot) * VS + 4 * f, stg[k]);
        }
        __syncthreads();
        {
            // (behind the last chunk: an empty descriptor range -- no memory traffic, no branch)
            const bool more = c0 + CK < a.Cin;
#pragma unroll
            for (int k = 0; k < NST; ++k) stg[k] = stx_buf_ld4(xrs, more ? vo[k] : STX_BUF_OOB, (unsigned)(c0 + CK) * 4u);
        }
        const float* wq = wbase + (size_t)(c0 / 8) * NT * 256;
        if constexpr (KS == 3) {
            conv_chunk_taps27<S, MT, NTW, NQC, EH, EWS, EWH, VS, NT>(tile, abase, wq,
                                                                     c0 + CK < a.Cin ? wq + (size_t)NQC * NT * 256 : nullptr, NQ, acc, bring);
        } else {
            // 1x1x1: both operands one step ahead in registers (A from the LDS tile, B = packed weights from L2); the
            // scheduling fences keep "issue the next loads, then this step's MFMAs"
            float4 bcur[NQC][NT], bnxt[NQC][NT], acur[NQC][MT], anxt[NQC][MT];
            auto load_tap = [&](int tap, float4 (&av)[NQC][MT], float4 (&bv)[NQC][NT]) {
                const float* wtap = wq + (size_t)tap * NQ * NT * 256;
#pragma unroll
                for (int q = 0; q < NQC; ++q) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[q][nt] = stx_ld4(wtap + (size_t)(q * NT + nt) * 256);
#pragma unroll
                    for (int m = 0; m < MT; ++m) av[q][m] = stx_ld4(tile + abase[m] + q * 8);
                }
            };
            load_tap(0, acur, bcur);
            for (int tap = 0; tap < T; ++tap) {
                if (tap + 1 < T) load_tap(tap + 1, anxt, bnxt);
                STX_SCHED_BARRIER();
#pragma unroll
                for (int q = 0; q < NQC; ++q)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q][m].x, bcur[q][nt].x, acc[m][nt], 0, 0, 0);
                            acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q][m].y, bcur[q][nt].y, acc[m][nt], 0, 0, 0);
                            acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q][m].z, bcur[q][nt].z, acc[m][nt], 0, 0, 0);
                            acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q][m].w, bcur[q][nt].w, acc[m][nt], 0, 0, 0);
                        }
                STX_SCHED_BARRIER();
#pragma unroll
                for (int q = 0; q < NQC; ++q) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bcur[q][nt] = bnxt[q][nt];
#pragma unroll
                    for (int m = 0; m < MT; ++m) acur[q][m] = anxt[q][m];
                }
            }
        }
    }

    float s1[NTW], s2[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) { s1[nt] = 0.f; s2[nt] = 0.f; }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mtile = wrow * MT + m;
        const int od = od0 + mtile / TH, oh = oh0 + mtile % TH;
        int nrows = (od < a.Do && oh < a.Ho) ? (a.Wo - ow0) : 0;
        nrows = nrows > 32 ? 32 : nrows;
        const size_t vox0 = (((size_t)b * a.Do + od) * a.Ho + oh) * a.Wo + ow0;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int n = (wcol * NTW + nt) * 32 + i;
            const float sc = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
            const float bs = (a.bias && n < a.Cout) ? a.bias[n] : 0.f;
            conv_epilogue_block(a, acc[m][nt], vox0, 1, nrows, n, sc, bs, lane, s1[nt], s2[nt]);
        }
    }
    if (a.stats) conv_write_stats<NT, WN>(a, s1, s2, tile, tid, (size_t)blockIdx.y * gridDim.x + blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// Persistent, software-pipelined variant of conv3d_igemm_kernel (same tile geometry, operand feed and epilogue).
//
// conv3d_igemm_kernel runs "stage the halo tile of a K chunk -> barrier -> 27 taps of MFMAs" once per workgroup and
// leaves the overlap of one workgroup's staging with the other's MFMAs to chance (two workgroups per CU): measured
// 0.41-0.62 of the fp32-MFMA peak, worst where a chunk carries few MFMAs per staged byte (stride 2: 8-channel chunks).
// Here a workgroup is persistent: it owns a contiguous run of output tiles and walks the flattened (tile, K chunk)
// sequence; the halo tile of item i+1 is loaded global -> registers right before the MFMA loop of item i and only
// written to LDS after it, between the two barriers that separate the items.  The global-memory latency of the
// staging is therefore always covered by a full tap loop; what stays exposed is the register -> LDS copy.
// WN: waves side by side along N, as in conv3d_igemm_kernel (128 output channels: WN = 2 -> two rows x two blocks per wave,
// WN = 4 -> four rows x one block).
template <int KS, int S, int TD, int TH, int NT, int CK, int WN = 1>
// (occupancy hint for the stride-1 8-channel-chunk variants only: 328 -> 221 VGPRs without spilling; the stride-2 ones
//  spill under the same cap and measured 0.152 -> 0.205 ms on 64->128, GPU call U)
__global__ __launch_bounds__(CONV_THREADS, (CK == 8 && S == 1) ? 2 : 1) void conv3d_pgemm_kernel(ConvArgs a, int ntiles_total) {
    constexpr int PAD = KS / 2;
    constexpr int ED = (TD - 1) * S + KS, EH = (TH - 1) * S + KS, EW = 31 * S + KS;
    constexpr int EWH = (EW + 1) / 2;
    constexpr int EWS = (S == 2) ? 2 * EWH : EW;
    constexpr int MT = TD * TH / 4 * WN;
    constexpr int NTW = NT / WN;
    constexpr int VS = CK + 4;
    constexpr int NF4 = CK / 4;
    constexpr int NE = ED * EH * EW * NF4;                       // float4 elements of a staged tile
    constexpr int NST = (NE + CONV_THREADS - 1) / CONV_THREADS;  // per thread
    constexpr int T = KS * KS * KS, NQC = CK / 8;
    static_assert(TD * TH % 4 == 0, "tile must split over 4 waves");
    static_assert(NT % WN == 0 && 4 % WN == 0, "wave grid");
    STX_DYN_SMEM(smem);
    float* tile = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar address math, no waterfall loops around the descriptors)
    const int i = lane & 31, half = lane >> 5;
    const int NQ = a.Cin / 8, nchunk = a.Cin / CK;
    const int tiles_per_b = a.nDt * a.nHt * a.nWt;
    // contiguous run of tiles (XCD-aware: neighbouring runs share halos in one L2)
    const long long wg = xcd_remap(blockIdx.x, gridDim.x);
    const int t_lo = __builtin_amdgcn_readfirstlane((int)((long long)ntiles_total * wg / gridDim.x));
    const int t_hi = __builtin_amdgcn_readfirstlane((int)((long long)ntiles_total * (wg + 1) / gridDim.x));

    const int wrow = wave / WN, wcol = wave % WN;
    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mtile = wrow * MT + m;
        const int td = mtile / TH, th = mtile % TH;
        abase[m] = ((td * S * EH + th * S) * EWS + i) * VS + 4 * half;
    }
    // staging map of this thread: element e = tid + k*256 -> (voxel v, float4 f); voxel -> (dz, hy, wx)
    float4 stg[NST];
    auto tile_origin = [&](int t, int& b, int& od0, int& oh0, int& ow0) {
        b = t / tiles_per_b;
        const int r = t - b * tiles_per_b;
        const int wt = r % a.nWt, ht = (r / a.nWt) % a.nHt, dt = r / (a.nWt * a.nHt);
        od0 = dt * TD; oh0 = ht * TH; ow0 = wt * 32;
    };
    using HM = HaloMap<ED, EH, EW, NF4>;
    HM hm;
    hm.init(tid, a.Hi, a.Wi, a.Cin);
    auto load_item = [&](int t, int c0) {
        int b, od0, oh0, ow0;
        tile_origin(t, b, od0, oh0, ow0);
        const int id0 = od0 * S - PAD, ih0 = oh0 * S - PAD, iw0 = ow0 * S - PAD;
        unsigned vo[NST];
        hm.offsets(id0, ih0, iw0, a.Di, a.Hi, a.Wi, vo);
        const stx_bufrsrc xrs = HM::rsrc(a.x, b, id0, ih0, iw0, a.Di, a.Hi, a.Wi, a.Cin);
#pragma unroll
        for (int k = 0; k < NST; ++k) stg[k] = stx_buf_ld4(xrs, vo[k], (unsigned)c0 * 4u);
    };
    auto store_item = [&]() {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = tid + k * CONV_THREADS;
            const int v = e / NF4, f = e - v * NF4;
            const int wx = v % EW, hy = (v / EW) % EH, dz = v / (EW * EH);
            const int slot = (S == 2) ? (wx & 1) * EWH + (wx >> 1) : wx;
            if (e < NE) stx_st4(tile + ((dz * EH + hy) * EWS + slot) * VS + 4 * f, stg[k]);
        }
    };

    float s1[NTW], s2[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) { s1[nt] = 0.f; s2[nt] = 0.f; }
    f32x16 acc[MT][NTW];
    float4 bring[3][NQC][NTW];                       // packed weights, two taps ahead (conv_chunk_taps27)
    static_assert(KS == 3, "the pipelined kernel serves 3x3x3 only");
    const float* wbase = a.wp + (size_t)lane * 4 + (size_t)wcol * NTW * 256;          // this wave's first column block
    if (t_lo < t_hi) {
        load_item(t_lo, 0);
        conv_ring_prologue<NTW, NQC, NT>(wbase, NQ, bring);
    }
    for (int t = t_lo; t < t_hi; ++t) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[m][nt] = zero16();
        for (int ch = 0; ch < nchunk; ++ch) {
            __syncthreads();                   // every wave is done reading the previous item's tile
            store_item();
            __syncthreads();
            {                                  // next item's halo tile: in flight during this item's tap loop
                const int nch = ch + 1 < nchunk ? ch + 1 : 0;
                const int ntl = ch + 1 < nchunk ? t : t + 1;
                if (ntl < t_hi) load_item(ntl, nch * CK);
            }
            const float* wq = wbase + (size_t)(ch * (CK / 8)) * NT * 256;
            // (the weight ring runs on across chunks AND tiles: behind a tile's last chunk come the first taps of chunk 0)
            const float* wq_next = ch + 1 < nchunk ? wq + (size_t)NQC * NT * 256 : (t + 1 < t_hi ? wbase : nullptr);
            conv_chunk_taps27<S, MT, NTW, NQC, EH, EWS, EWH, VS, NT>(tile, abase, wq, wq_next, NQ, acc, bring);
        }
        int b, od0, oh0, ow0;
        tile_origin(t, b, od0, oh0, ow0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int mtile = wrow * MT + m;
            const int od = od0 + mtile / TH, oh = oh0 + mtile % TH;
            int nrows = (od < a.Do && oh < a.Ho) ? (a.Wo - ow0) : 0;
            nrows = nrows > 32 ? 32 : nrows;
            const size_t vox0 = (((size_t)b * a.Do + od) * a.Ho + oh) * a.Wo + ow0;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int n = (wcol * NTW + nt) * 32 + i;
                const float sc = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
                const float bs = (a.bias && n < a.Cout) ? a.bias[n] : 0.f;
                conv_epilogue_block(a, acc[m][nt], vox0, 1, nrows, n, sc, bs, lane, s1[nt], s2[nt]);
            }
        }
    }
    if (a.stats) conv_write_stats<NT, WN>(a, s1, s2, tile, tid, (size_t)blockIdx.x);
}

// (1x1x1 convolutions -- the hourglass `redir` layers, HBM-bound -- go through the implicit-GEMM kernel with 32-channel chunks.
//  A persistent streaming form for 32 -> 32 channels without an LDS tile -- a wave owns 32 consecutive voxels per step, A operands
//  straight from memory one step ahead of the 16 MFMAs -- measured 0.117 ms against 0.108-0.112 ms, GPU call Y of round 3: removed.)
struct MarchArgs {
    ConvArgs c;
    int ncols;           // B * nHt * nWt workgroup columns; the (column, d) plane list is split evenly over the grid
    int ablate;          // profiling only (STX_MARCH_ABLATE): 1 = no plane staging, 2 = no epilogue stores
    // channel slicing of the second-generation kernel (32 x 32 slices of wider layers):
    int xs, xo;          // floats per input voxel, first input channel of the slice
    int os, oo, ncout;   // floats per output voxel, first output channel, valid output channels of the slice (<= 32)
    int wq_total, wq_off, wnt_total, wnt_off;   // slice of the packed weights [tap][K/8][N/32][lane][4]
    const float* acc_in; // partial sums of an earlier K slice (same layout as out), or null
};

// ------------------------------------------------------------------------------------------
// 3x3x3 stride-1 convolution in 32 x 32 channel slices ("march" kernel): the WEIGHTS live in LDS and the planes are
// input-stationary.
//
// (The first generation of this kernel -- rounds 1-2, removed in round 3 -- streamed the packed weights from L2 for every
// tap of every plane and needed two barriers + an accumulator exchange per plane: 0.60-0.63 of the fp32-MFMA peak.)
// A workgroup is 4 waves, ONE per SIMD, and
//   * the 27 x 32 x 32 packed weights (110 592 B, MFMA B-operand order) are loaded into LDS once per workgroup; both
//     operands of every MFMA then come from LDS (8 ds_read_b128 per 16 MFMAs per wave = 12 % of the LDS bandwidth);
//   * LDS holds two input planes (double buffer, 10 x 18 voxels x 36 dwords = 25 920 B each: a column of 8 x 16 output
//     voxels, one 2 x 16 row block per wave).  Input plane p is multiplied by ALL 27 taps while it is resident: its
//     kd = 2 / 1 / 0 tap planes accumulate into the outputs p-1 / p / p+1, whose three accumulator sets rotate through
//     registers (48 VGPRs).  Plane p+1 is in flight into registers meanwhile and is written to the other buffer at the
//     end of the step: ONE barrier per plane, no accumulator exchange;
//   * output p-1 is complete after the first 9 taps of step p: its epilogue (affine / residual / ReLU / BN partial
//     sums / stores) is emitted in slices between the remaining 18 taps' MFMA groups, i.e. in the shadow of the pipe.
// 110 592 + 2 x 25 920 = 162 432 B of the CU's 163 840 B of LDS: one workgroup per CU, 432 MFMAs per wave and plane.
constexpr int MW2_MW = 16, MW2_R = 2, MW2_TH = 8, MW2_EH = 10, MW2_EW = 18, MW2_VS = 36;
constexpr int MW2_SLOT = MW2_EH * MW2_EW * MW2_VS;                  // floats per plane buffer
constexpr int MW2_NF4 = (MW2_EH * MW2_EW * 8 + 255) / 256;          // staging float4 per thread
constexpr int MW2_WFLOATS = 27 * 4 * 256;                           // packed weights [tap][q][lane][4]

// BS ("blocked sums"): the 27 x 32 products of an output element are not accumulated as ONE sequential fp32 chain of 864
// fused multiply-adds: the contribution of each of the three input planes (288 products) gets its own accumulator,
// started from zero, and the three partial sums are added in the epilogue.  The rounding error of a sequential fp32 sum
// of n terms grows like n, that of c-term chunks like sqrt(n^2 / c + n c): 1.7x smaller here.  (Attribution on the
// emulator, GwcNet_GC(192): the sequential chains of these layers alone caused half of the product's distance from an
// fp64 evaluation.)  Cost: three more accumulator sets in registers (six rotate in two 3-cycles instead of three in
// one), two VALU adds per output element in the epilogue; the MFMA stream is unchanged.
// (Finer chunks -- one accumulator per kh row, added to the running sum with VALU adds between the taps -- were tried
//  first, GPU calls B/C/E of round 3: on the chip element 15 of the freshly written accumulator tuple came back without
//  the last one to three MFMAs of its chain whenever hipcc placed its v_accvgpr_read first behind the minimal wait, in
//  two of the three rotations; the host emulator and a stand-alone test of the same instruction pattern
//  (tools/ubench/mfma_tail_read.hip) are correct.  Not pursued: every accumulator read of this kernel is at least 16 MFMAs
//  behind the chain that wrote it.)
// The eight ds_read_b128 of the NEXT tap are dealt one per two MFMAs of the current tap (sched_group_barrier) instead of being
// issued as one 8 KB burst in front of them (a wave alone on its SIMD pays for its own read burst: MI355X guide, "two waves
// per SIMD", item 7; GPU call F of round 3: 0.776 -> 0.757 ms).
// EPI: a wave that is alone on its SIMD also pays for every instruction it issues that is not an MFMA: GPU call H2 of
// round 3 measured 0.766 ms with and 0.681 ms without the epilogue, and the listing showed why -- per output row three
// 64-bit multiply-adds for the address, an exec-mask save / restore, four to six branches (bounds, partial sums, residual,
// activation code) that cut the MFMA stream into basic blocks the scheduler cannot fill, and an s_waitcnt vmcnt(0) behind each
// conditional load.  EPI = 0 (no partial sums of an earlier K slice, no residual, activation none / ReLU: every
// training-mode launch and most inference ones) is straight-line code: the row's voxel goes through a buffer descriptor of the
// output PLANE (wave-uniform base in SGPRs, one per-lane byte offset per column of the work list, a wave-uniform row offset in
// soffset), rows outside the volume get the out-of-range offset (the store is dropped by the bounds check) and contribute
// zeros to the BN sums.  EPI = 2 is the same with the partial sums of an earlier K slice added (second half of a 64 -> 32
// layer: its 16 values per lane are loaded through the same descriptor at the START of the step, nine taps before the first
// is needed -- the general epilogue waited for each of them with s_waitcnt vmcnt(0): 1.10 vs 0.70 ms per launch in the
// step trace of GPU call N); EPI = 3 the same with a residual added behind the affine (inference: dres1's second conv).
// EPI = 4: EPI = 0 for raw outputs (no affine, no activation: every training-mode launch) -- three vector instructions per
// row less; EPI = 5: the same for whole tiles and 32 output channels (no per-row validity: three more).  EPI = 1 is the general
// epilogue (partial sums AND residual, Mish).  The plane staging loads go through a descriptor of the input
// plane the same way for both (offsets precomputed per column; halo voxels outside the volume and planes outside [0, Di)
// read zeros through the bounds check: no address clamps, no branches).
template <int BS, int EPI = 1>
__global__ __launch_bounds__(256) void conv3d_marchw_kernel(MarchArgs ma) {
    const ConvArgs& a = ma.c;
    STX_DYN_SMEM(smem);
    float* wl = reinterpret_cast<float*>(smem);                      // [27][4][64][4]
    float* planes = wl + MW2_WFLOATS;                                // [2][MW2_SLOT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // (this kernel forces uniformity where it matters: explicit readfirstlanes below)
    const int i = lane & 31, half = lane >> 5;
    const int th = wave;                                             // row block of this wave
    auto n_lane = [&]() { return i; };                               // output channel of this lane

    // weights -> LDS (once): slice [tap][wq_off .. +4][wnt_off] of the packed tensor
    for (int e = tid; e < MW2_WFLOATS / 4; e += 256) {
        const int tq = e >> 6, l4 = e & 63;                          // (tap * 4 + q), lane
        const int tap = tq >> 2, q = tq & 3;
        const size_t src = (((size_t)tap * ma.wq_total + ma.wq_off + q) * ma.wnt_total + ma.wnt_off) * 256 + 4 * l4;
        stx_st4(wl + 4 * e, stx_ld4(a.wp + src));
    }

    int b = 0, oh0 = 0, ow0 = 0;
    const long long units = (long long)ma.ncols * a.Do;
    const long long wg = xcd_remap(blockIdx.x, gridDim.x);
    int u = __builtin_amdgcn_readfirstlane((int)(units * wg / gridDim.x));
    const int u_end = __builtin_amdgcn_readfirstlane((int)(units * (wg + 1) / gridDim.x));

    float4 stg[MW2_NF4];
    // per column of the work list: byte offsets of this thread's staging float4s inside an input plane (out-of-range marker
    // for halo voxels outside the volume), of this lane's first output voxel inside an output plane, validity bits of its rows
    unsigned svoff[MW2_NF4], ovoff = 0, ovalid = 0;
    const unsigned xplane_bytes = (unsigned)a.Hi * (unsigned)a.Wi * (unsigned)ma.xs * 4u;
    const unsigned oplane_bytes = (unsigned)a.Ho * (unsigned)a.Wo * (unsigned)ma.os * 4u;
    auto column_offsets = [&]() {
#pragma unroll
        for (int k = 0; k < MW2_NF4; ++k) {
            const int idx = tid + k * 256;
            const int v = idx >> 3, f = idx & 7;
            const int wx = v % MW2_EW, hy = v / MW2_EW;
            const int gh = oh0 - 1 + hy, gw = ow0 - 1 + wx;
            const bool ok = v < MW2_EH * MW2_EW && gh >= 0 && gh < a.Hi && gw >= 0 && gw < a.Wi;
            svoff[k] = ok ? (unsigned)(((gh * a.Wi + gw) * ma.xs + ma.xo + 4 * f) * 4) : STX_BUF_OOB;
        }
        const int ohb = oh0 + th * MW2_R, owb = ow0 + 4 * half;
        ovoff = (unsigned)((((ohb * a.Wo + owb) * ma.os) + ma.oo + n_lane()) * 4);
        ovalid = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * ((r >> 2) & 1), dh = r >> 3;
            if (ohb + dh < a.Ho && owb + c < a.Wo && n_lane() < ma.ncout) ovalid |= 1u << r;
        }
    };
    auto load_plane = [&](int pd, bool on) {
        const bool in = on && pd >= 0 && pd < a.Di;                  // (wave-uniform) planes outside the volume read zeros
        const stx_bufrsrc rs = stx_make_rsrc(a.x + ((size_t)b * a.Di + (in ? pd : 0)) * a.Hi * a.Wi * ma.xs, in ? xplane_bytes : 0u);
#pragma unroll
        for (int k = 0; k < MW2_NF4; ++k) stg[k] = stx_buf_ld4(rs, svoff[k], 0u);
    };
    auto store_plane = [&](float* buf) {
#pragma unroll
        for (int k = 0; k < MW2_NF4; ++k) {
            const int idx = tid + k * 256;
            const int v = idx >> 3, f = idx & 7;
            if (v < MW2_EH * MW2_EW) stx_st4(buf + v * MW2_VS + 4 * f, stg[k]);
        }
    };

    float s1 = 0.f, s2 = 0.f;
    const int abase = ((th * MW2_R + i / MW2_MW) * MW2_EW + i % MW2_MW) * MW2_VS + 4 * half;
    const float* wlane = wl + lane * 4;
    const int n = i;                                                 // output channel of this lane
    const float sc = (a.scale && n < ma.ncout) ? a.scale[ma.oo + n] : 1.f;
    const float bs = (a.bias && n < ma.ncout) ? a.bias[ma.oo + n] : 0.f;

    // one output row (voxel) of a finished plane: partial sums of an earlier K slice, BN statistics, affine, residual, ReLU
    auto emit_row = [&](const f32x16& done0, const f32x16& done1, const f32x16& done, int r, int dprev) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int oh = oh0 + th * MW2_R + row / MW2_MW, ow = ow0 + row % MW2_MW;
        if (oh < a.Ho && ow < a.Wo && n < ma.ncout && ma.ablate != 2) {
            const size_t idx = ((((size_t)b * a.Do + dprev) * a.Ho + oh) * a.Wo + ow) * ma.os + ma.oo + n;
            float v = BS ? (done0[r] + done1[r]) + done[r] : done[r];        // BS: partial sums of the planes p-2, p-1, p
            if (ma.acc_in) v += ma.acc_in[idx];
            s1 += v;
            s2 = fmaf(v, v, s2);
            v = fmaf(v, sc, bs);
            if (a.residual) v += a.residual[idx];
            v = stx_act(v, a.relu);
            a.out[idx] = v;
        }
    };
    // EPI = 0: the same row as straight-line code (see the kernel comment); `vmask` = validity bits of this lane's rows in the
    // finished plane (zero when there is none), `ors` = descriptor of that output plane
    const bool relu_on = a.relu == 1;
    float accp[16];                                                  // EPI = 2: partial sums of the earlier K slice (rows of plane p-1)
    auto load_partials = [&](unsigned vmask, const stx_bufrsrc& rs) {      // (EPI = 3: the residual rows instead)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * ((r >> 2) & 1), dh = r >> 3;
            accp[r] = stx_buf_ld1(rs, ((vmask >> r) & 1u) ? ovoff : STX_BUF_OOB, (unsigned)((dh * a.Wo + c) * ma.os * 4));
        }
    };
    auto emit_row_plain = [&](const f32x16& done0, const f32x16& done1, const f32x16& done, int r, unsigned vmask,
                              const stx_bufrsrc& ors) {
        const int c = (r & 3) + 8 * ((r >> 2) & 1), dh = r >> 3;
        // EPI = 5 (raw output, whole tiles, 32 output channels): every row exists; without a finished plane (vmask = 0, wave-
        // uniform) the descriptor is empty, so the store needs no per-lane select and the sums one select on a scalar condition
        const bool ok = EPI == 5 ? vmask != 0u : (bool)((vmask >> r) & 1u);
        float v = BS ? (done0[r] + done1[r]) + done[r] : done[r];
        if (EPI == 2) v += accp[r];
        v = ok ? v : 0.f;
        s1 += v;
        s2 = fmaf(v, v, s2);
        if (EPI != 4 && EPI != 5) {                                  // (EPI = 4 / 5: raw output -- no affine, no activation)
            v = fmaf(v, sc, bs);
            if (EPI == 3) v += accp[r];
            const float vr = fmaxf(v, 0.f);
            v = relu_on ? vr : v;
        }
        stx_buf_st1(ors, (EPI == 5 || ok) ? ovoff : STX_BUF_OOB, (unsigned)((dh * a.Wo + c) * ma.os * 4), v);
    };

    // 9 taps of one kd plane into `acc`; operands one tap ahead (registers), optional epilogue slices of `done`
    // (the finished output plane dprev) interleaved with the MFMA groups
    // fresh: the accumulator starts here -- the first MFMA takes a zero C operand (an inline constant) instead of a register
    // set that 16 vector moves zeroed beforehand
    auto tap_plane = [&](const float* pbuf, int kd, f32x16& acc, bool fresh, bool with_epi, const f32x16& done0, const f32x16& done1,
                         const f32x16& done, int dprev, int epi0, unsigned vmask, const stx_bufrsrc& ors) {
        float4 av[2][4], bv[2][4];
        auto load_tap = [&](int t9, int buf) {
            const int kh = t9 / 3, kw = t9 % 3;
            const float* sl = pbuf + abase + (kh * MW2_EW + kw) * MW2_VS;
            const float* wt = wlane + (kd * 9 + t9) * 1024;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                av[buf][q] = stx_ld4(sl + 8 * q);
                bv[buf][q] = stx_ld4(wt + q * 256);
            }
        };
        load_tap(0, 0);
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
            if (t9 + 1 < 9) load_tap(t9 + 1, (t9 + 1) & 1);
            const int cb = t9 & 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][q].x, bv[cb][q].x, (fresh && t9 == 0 && q == 0) ? zero16() : acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][q].y, bv[cb][q].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][q].z, bv[cb][q].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][q].w, bv[cb][q].w, acc, 0, 0, 0);
            }
            // rows epi0 + t9 of the finished plane: 16 rows over the 18 taps of the kd = 1 and kd = 0 planes
            const int r = epi0 + t9;
            if (EPI != 1) {
                if (epi0 >= 0 && r < 16) emit_row_plain(done0, done1, done, r, vmask, ors);
            }
            if (t9 + 1 < 9) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    STX_SCHED_GROUP(0x008, 2);
                    STX_SCHED_GROUP(0x100, 1);
                    if (EPI != 1) STX_SCHED_GROUP(0x002, 2);
                }
            }
            if (EPI == 1 && with_epi && r < 16) emit_row(done0, done1, done, r, dprev);
            STX_SCHED_BARRIER();
        }
    };
    // one input plane p (resident in `pbuf`): kd = 2 -> output p-1 (finished here), kd = 1 -> output p, kd = 0 -> output p+1
    // (starts here).  BS = 0: one accumulator per output: aOld / aMid / aNew = pC / pE / pF (pA, pB, pD unused).
    // BS = 1: one accumulator per (output, input plane): output p-1 = pA (plane p-2) + pB (plane p-1) + pC (this plane),
    // output p = pD (plane p-1) + pE (this plane), output p+1 = pF (this plane); pC, pE, pF start from zero here.
    auto step = [&](int p, int d_lo, int d_hi, const float* pbuf, float* nbuf, bool stage, f32x16& pA, f32x16& pB,
                    f32x16& pC, f32x16& pD, f32x16& pE, f32x16& pF) {
        const bool live = p >= 0 && p < a.Di;                        // planes outside the volume are zero padding
        const bool vOld = p - 1 >= d_lo && p - 1 < d_hi, vMid = p >= d_lo && p < d_hi, vNew = p + 1 >= d_lo && p + 1 < d_hi;
        const bool st_on = vOld && ma.ablate != 2;
        const unsigned vmask = st_on ? ovalid : 0u;
        const stx_bufrsrc ors = stx_make_rsrc(a.out + ((size_t)b * a.Do + (st_on ? p - 1 : 0)) * a.Ho * a.Wo * ma.os,
                                              st_on ? oplane_bytes : 0u);
        if (EPI == 2) load_partials(vmask, ors);                     // (acc_in == out: the rows this step finishes, read before written)
        if (EPI == 3)
            load_partials(vmask, stx_make_rsrc(a.residual + ((size_t)b * a.Do + (st_on ? p - 1 : 0)) * a.Ho * a.Wo * ma.os,
                                               st_on ? oplane_bytes : 0u));
        if (live && vOld) tap_plane(pbuf, 2, pC, BS != 0, false, pA, pB, pC, 0, -1, 0u, ors);
        else if (BS) pC = zero16();
        // the next plane (in flight since the start of the step) goes into the other buffer now: that buffer has been
        // free since the barrier that ended the previous step, and the remaining 18 taps cover the LDS writes
        if (stage) store_plane(nbuf);
        // output p-1 is complete: its 16 rows leave in the shadow of the next 18 taps (or on their own at the edges)
        if (live && vMid) tap_plane(pbuf, 1, pE, BS != 0, vOld, pA, pB, pC, p - 1, 0, vmask, ors);
        else if (BS) pE = zero16();
        if (!(live && vMid) && vOld) {
            // no MFMAs to hide behind: plain epilogue of rows 0..8
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                if (EPI != 1) emit_row_plain(pA, pB, pC, r, vmask, ors);
                else emit_row(pA, pB, pC, r, p - 1);
            }
        }
        if (live && vNew) tap_plane(pbuf, 0, pF, true, vOld, pA, pB, pC, p - 1, 9, vmask, ors);
        else pF = zero16();
        if (!(live && vNew) && vOld) {
#pragma unroll
            for (int r = 9; r < 16; ++r) {
                if (EPI != 1) emit_row_plain(pA, pB, pC, r, vmask, ors);
                else emit_row(pA, pB, pC, r, p - 1);
            }
        }
    };

    __syncthreads();                                                 // weights are in LDS
    f32x16 r0 = zero16(), r1 = zero16(), r2 = zero16(), r3 = zero16(), r4 = zero16(), r5 = zero16();
    while (u < u_end) {
        int d_lo, d_hi;
        {
            const int col = u / a.Do;
            d_lo = u - col * a.Do;
            const int left = u_end - u;
            d_hi = (a.Do - d_lo < left) ? a.Do : d_lo + left;
            u += d_hi - d_lo;
            const int wt = col % a.nWt, ht = (col / a.nWt) % a.nHt;
            b = col / (a.nWt * a.nHt);
            oh0 = ht * MW2_TH; ow0 = wt * MW2_MW;
        }
        // input planes d_lo-1 .. d_hi feed the outputs d_lo .. d_hi-1
        column_offsets();
        __syncthreads();                                             // the previous run is done with both buffers
        load_plane(d_lo - 1, true);
        store_plane(planes);
        __syncthreads();
        int par = 0;
        auto advance = [&](int p, f32x16& pA, f32x16& pB, f32x16& pC, f32x16& pD, f32x16& pE, f32x16& pF) {
            const bool stage = p + 1 <= d_hi && ma.ablate != 1;
            load_plane(p + 1, stage);                                // in flight during this plane's first 9 taps
            step(p, d_lo, d_hi, planes + par * MW2_SLOT, planes + (par ^ 1) * MW2_SLOT, stage, pA, pB, pC, pD, pE, pF);
            __syncthreads();                                         // plane p+1 is resident, plane p's buffer is free
            par ^= 1;
        };
        if (BS) {
            // after a step: output p becomes "old" (A <- D, B <- E), output p+1 "mid" (D <- F); the three freed sets are
            // handed out so that the renaming is two 3-cycles (A D F) (B E C): three unrolled steps, as without BS
            for (int p = d_lo - 1; p <= d_hi; p += 3) {
                advance(p, r0, r1, r2, r3, r4, r5);
                if (p + 1 <= d_hi) advance(p + 1, r3, r4, r1, r5, r2, r0);
                if (p + 2 <= d_hi) advance(p + 2, r5, r2, r4, r0, r1, r3);
            }
        } else {
            for (int p = d_lo - 1; p <= d_hi; p += 3) {
                advance(p, r3, r4, r0, r5, r1, r2);                  // (aOld, aMid, aNew) = (r0, r1, r2) rotate
                if (p + 1 <= d_hi) advance(p + 1, r3, r4, r1, r5, r2, r0);
                if (p + 2 <= d_hi) advance(p + 2, r3, r4, r2, r5, r0, r1);
            }
        }
    }
    if (a.stats) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        __syncthreads();
        if (lane < 32) {
            planes[(wave * 32 + lane) * 2 + 0] = s1;
            planes[(wave * 32 + lane) * 2 + 1] = s2;
        }
        __syncthreads();
        if (tid < 32 && tid < ma.ncout) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                t1 += planes[(w * 32 + tid) * 2 + 0];
                t2 += planes[(w * 32 + tid) * 2 + 1];
            }
            const size_t slab = (size_t)blockIdx.x;
            a.stats[slab * 2 * a.Cout + ma.oo + tid] = t1;
            a.stats[slab * 2 * a.Cout + a.Cout + ma.oo + tid] = t2;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Transposed convolution k3, stride 2, padding 1, output_padding 1 (out = 2 x in), computed as its
// 8 output-parity classes with 1/2/2/2/4/4/4/8 taps: true MACs only, no zero insertion.
//   out[2m+p] (per axis): p=0 -> tap k=1 at input m;  p=1 -> tap k=0 at input m+1 and k=2 at m.
// A workgroup owns TWO rows of 32 input columns; each row is shared by two waves that split the 8
// classes 13/14 taps ({111,100,010,000} / {011,101,110,001}); accumulators of a wave's four
// classes persist over the K-chunks so the input tile is staged once per chunk.
// The (class, tap) work list of one wave set, enumerated at compile time: 13 entries for the classes {111,100,010,000},
// 14 for {011,101,110,001}.  slot = accumulator set of the class, tap = weight tap, (dd, dh, dw) = input offset.
struct DcEntries { int n; int slot[14], tap[14], dd[14], dh[14], dw[14]; };
constexpr DcEntries dc_entries(int cset) {
    DcEntries e{};
    const int tab[2][4] = {{7, 4, 2, 0}, {3, 5, 6, 1}};
    int n = 0;
    for (int c = 0; c < 4; ++c) {
        const int cls = tab[cset][c];
        const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
        for (int sd = 0; sd <= pd; ++sd)
            for (int sh = 0; sh <= ph; ++sh)
                for (int sw = 0; sw <= pw; ++sw) {
                    // parity 0: (k=1, delta=0); parity 1: s=0 -> (k=0, delta=1), s=1 -> (k=2, delta=0)
                    const int kd = pd ? (sd ? 2 : 0) : 1, kh = ph ? (sh ? 2 : 0) : 1, kw = pw ? (sw ? 2 : 0) : 1;
                    e.slot[n] = c;
                    e.tap[n] = (kd * 3 + kh) * 3 + kw;
                    e.dd[n] = pd ? (sd ? 0 : 1) : 0;
                    e.dh[n] = ph ? (sh ? 0 : 1) : 0;
                    e.dw[n] = pw ? (sw ? 0 : 1) : 0;
                    ++n;
                }
    }
    e.n = n;
    return e;
}

// One K chunk of the transposed convolution for a wave: the work list above as straight-line code, the weight operands
// of entry t+1 in flight (second register buffer) during the 4 * CK/8 * NT MFMAs of entry t.  (The first version walked
// the classes and taps in a loop nest with run-time bounds; hipcc kept it rolled and waited for each tap's loads right
// before its first MFMA, an L2 round trip per 16 MFMAs: 0.45 / 0.33 of the fp32-MFMA peak for the two GwcNet shapes, GPU
// call O of round 2.  Prefetching the LDS operands too, and dealing the loads between the MFMAs, measured 0-4 % slower:
// calls Q-T of round 2, call G of round 3; the weights SIX entries ahead in a ring of seven register sets that runs on
// across the K chunks: 0.138 / 0.245 -> 0.141 / 0.252 ms, call R of round 3 -- no gain, removed.  16-channel K chunks with the
// chunk prefetch in place: 0.140 / 0.253 vs 0.140 / 0.250 ms; the raw whole-block epilogue (two vector instructions per row
// instead of nine): no change.  Neither the weight latency, nor the barriers per chunk, nor the epilogue bound this kernel.)
template <int CSET, int NT, int CK>
__device__ __forceinline__ void deconv_chunk_taps(const float* atile, const float* wq, int NQ, f32x16 (&acc)[4][NT]) {
    constexpr DcEntries E = dc_entries(CSET);
    constexpr int EH = 3, EW = 33, VS = CK + 4, QS = CK / 8;
    float4 bv[2][QS][NT], av[2][QS];
    auto load_b = [&](int t, int buf) {
        const float* wtap = wq + (size_t)E.tap[t] * NQ * NT * 256;
#pragma unroll
        for (int q = 0; q < QS; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[buf][q][nt] = stx_ld4(wtap + (size_t)(q * NT + nt) * 256);
    };
    auto load_a = [&](int t, int buf) {
        const int toff = ((E.dd[t] * EH + E.dh[t]) * EW + E.dw[t]) * VS;
#pragma unroll
        for (int q = 0; q < QS; ++q) av[buf][q] = stx_ld4(atile + toff + q * 8);
    };
    load_b(0, 0);
#pragma unroll
    for (int t = 0; t < E.n; ++t) {
        if (t + 1 < E.n) load_b(t + 1, (t + 1) & 1);
        STX_SCHED_BARRIER();
        load_a(t, t & 1);
#pragma unroll
        for (int q = 0; q < QS; ++q) {
            const float4 a4 = av[t & 1][q];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 b = bv[t & 1][q][nt];
                acc[E.slot[t]][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b.x, acc[E.slot[t]][nt], 0, 0, 0);
                acc[E.slot[t]][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b.y, acc[E.slot[t]][nt], 0, 0, 0);
                acc[E.slot[t]][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b.z, acc[E.slot[t]][nt], 0, 0, 0);
                acc[E.slot[t]][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b.w, acc[E.slot[t]][nt], 0, 0, 0);
            }
        }
        STX_SCHED_BARRIER();
    }
}

template <int NT, int CK>
__global__ __launch_bounds__(CONV_THREADS, NT == 1 ? 3 : 2) void deconv3d_igemm_kernel(ConvArgs a) {
    constexpr int TH = 2;
    constexpr int ED = 2, EH = TH + 1, EW = 33;
    constexpr int VS = CK + 4, NF4 = CK / 4;
    STX_DYN_SMEM(smem);
    float* tile = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform wave index (scalar address math, no waterfall loops around the 4 x NT output descriptors): -5.5 % on 64 -> 32
    // (GPU call I of round 5); with 64 output channels the hoisted scalar state costs 100 spilled VGPRs under the 256-register
    // cap (0.146 -> 0.204 ms), so that instantiation keeps the per-lane form
    const int wave = NT == 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    const int i = lane & 31, half = lane >> 5;
    const int bid = xcd_remap(blockIdx.x, gridDim.x), b = blockIdx.y;
    const int wt = bid % a.nWt, ht = (bid / a.nWt) % a.nHt, dt = bid / (a.nWt * a.nHt);
    const int md0 = dt, mh0 = ht * TH, mw0 = wt * 32;
    const int NQ = a.Cin / 8;
    const int th = wave >> 1, cset = wave & 1;
    // class ids (pd<<2 | ph<<1 | pw) per wave set
    const int cls_tab[2][4] = {{7, 4, 2, 0}, {3, 5, 6, 1}};

    f32x16 acc[4][NT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[c][nt] = zero16();
    const int abase = ((th * EW) + i) * VS + 4 * half;

    using HM = HaloMap<ED, EH, EW, NF4>;
    constexpr int NST = HM::NST;
    unsigned vo[NST];
    {
        HM hm;
        hm.init(tid, a.Hi, a.Wi, a.Cin);
        hm.offsets(md0, mh0, mw0, a.Di, a.Hi, a.Wi, vo);
    }
    const stx_bufrsrc xrs = HM::rsrc(a.x, b, md0, mh0, mw0, a.Di, a.Hi, a.Wi, a.Cin);

    float4 val[NST];                                 // the next K chunk's tile: in flight during this chunk's taps
#pragma unroll
    for (int k = 0; k < NST; ++k) val[k] = stx_buf_ld4(xrs, vo[k], 0u);
    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = tid + k * CONV_THREADS;
            const int v = e / NF4, f = e - v * NF4;
            const int wx = v % EW, hy = (v / EW) % EH, dz = v / (EW * EH);
            if (e < HM::NE) stx_st4(tile + ((dz * EH + hy) * EW + wx) * VS + 4 * f, val[k]);
        }
        __syncthreads();
        {
            const bool more = c0 + CK < a.Cin;
#pragma unroll
            for (int k = 0; k < NST; ++k) val[k] = stx_buf_ld4(xrs, more ? vo[k] : STX_BUF_OOB, (unsigned)(c0 + CK) * 4u);
        }
        const float* wq = a.wp + ((size_t)(c0 / 8) * NT * 64 + lane) * 4;
        if (cset == 0) deconv_chunk_taps<0, NT, CK>(tile + abase, wq, NQ, acc);
        else deconv_chunk_taps<1, NT, CK>(tile + abase, wq, NQ, acc);
    }

    float s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { s1[nt] = 0.f; s2[nt] = 0.f; }
    // (epilogue: wave-uniform copies of the wave's row / class set in every instantiation -- the output descriptors and row
    //  offsets are scalar; the 64-channel kernel keeps the per-lane forms in its main loop, see above)
    const int thu = __builtin_amdgcn_readfirstlane(th), csetu = __builtin_amdgcn_readfirstlane(cset);
    const int mh = mh0 + thu;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int cls = cls_tab[csetu][c];
        const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
        const int od = 2 * md0 + pd, oh = 2 * mh + ph, ow_first = 2 * mw0 + pw;
        int nrows = 0;
        if (md0 < a.Di && mh < a.Hi && od < a.Do && oh < a.Ho) {
            nrows = (a.Wo - ow_first + 1) / 2;           // rows with 2*row+ow_first < Wo
            const int nin = a.Wi - mw0;                  // rows backed by an input column
            nrows = nrows < nin ? nrows : nin;
            nrows = nrows > 32 ? 32 : (nrows < 0 ? 0 : nrows);
        }
        const size_t vox0 = (((size_t)b * a.Do + od) * a.Ho + oh) * a.Wo + ow_first;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 32 + i;
            const float sc = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
            const float bs = (a.bias && n < a.Cout) ? a.bias[n] : 0.f;
            conv_epilogue_block(a, acc[c][nt], vox0, 2, nrows, n, sc, bs, lane, s1[nt], s2[nt]);
        }
    }
    if (a.stats) conv_write_stats<NT>(a, s1, s2, tile, tid, (size_t)blockIdx.y * gridDim.x + blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// Weight packing into the per-lane MFMA B-operand order:
//   wp[tap][q][nt][lane][j] = Wk[tap][k = 8q + 4*(lane>>5) + j][n = 32nt + (lane&31)]
// from a torch weight w[A][B][T]:
//   mode 0 (conv fwd, deconv dgrad):      K = B, N = A, Wk[tap][k][n] = w[n][k][tap]
//   mode 1 (stride-1 conv dgrad):         K = A, N = B, Wk[tap][k][n] = w[k][n][T-1-tap]
//   mode 2 (deconv fwd, stride-2 dgrad):  K = A, N = B, Wk[tap][k][n] = w[k][n][tap]
__global__ __launch_bounds__(CONV_THREADS) void conv3d_pack_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                                   int A, int Bd, int T, int mode, int K, int N,
                                                                   int NT, size_t total) {
    const size_t idx = (size_t)blockIdx.x * CONV_THREADS + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 3, lane = (idx >> 2) & 63;
    size_t r = idx >> 8;
    const int nt = r % NT; r /= NT;
    const int NQ = K / 8;
    const int q = r % NQ;
    const int tap = r / NQ;
    const int k = 8 * q + 4 * (lane >> 5) + j, n = 32 * nt + (lane & 31);
    float v = 0.f;
    if (n < N && k < K) {
        if (mode == 0) v = w[((size_t)n * Bd + k) * T + tap];
        else if (mode == 1) v = w[((size_t)k * Bd + n) * T + (T - 1 - tap)];
        else v = w[((size_t)k * Bd + n) * T + tap];
    }
    wp[idx] = v;
}

// ------------------------------------------------------------------------------------------
// Weight gradient: G[tap][cf][cc] = sum_o F[S*o + tap - pad][cf] * C[o][cc]
//   conv:   F = layer input x (fine), C = grad of output (coarse);  dW[cc][cf][tap] = G
//   deconv: F = grad of output (fine), C = layer input x (coarse);  dWt[cc][cf][tap] = G
// MFMA view: D[32 cf][32 cc] += A[cf][k = voxel pair] * B[k][cc]; K runs over output voxels.
// A workgroup owns one (cf-block, cc-block) pair, walks a strided set of spatial tiles
// (1 x TH x TW coarse voxels each) and keeps all its taps' 32x32 partial sums in registers
// (taps are dealt round-robin to the 4 waves; for 1x1x1 the waves split the voxels instead).
// Partials go to a slab [pair][chunk][(wave)][tap][32][32]; wgrad_reduce_kernel sums the slab in a
// fixed order (deterministic) and writes the torch-layout weight gradient.
struct WgradArgs {
    const float* f;     // [B][Df][Hf][Wf][CF]
    const float* c;     // [B][Dc][Hc][Wc][CC]
    float* slab;
    int B, Df, Hf, Wf, CF;
    int Dc, Hc, Wc, CC;
    int nHt, nWt, ntiles;
    int ablate;          // profiling only (STX_WGRAD_ABLATE): 1 = no tile staging, 2 = no MFMA loop
};

template <int KS, int S, int TH, int TW, int NW, bool PIPE>
__global__ __launch_bounds__(NW * 64) void conv3d_wgrad_kernel(WgradArgs a) {
    constexpr int NTHR = NW * 64;
    constexpr int PAD = KS / 2;
    constexpr int T = KS * KS * KS;
    constexpr int ED = KS, EH = (TH - 1) * S + KS, EW = (TW - 1) * S + KS;
    constexpr int EWH = (EW + 1) / 2;
    constexpr int EWS = (S == 2) ? 2 * EWH : EW;
    constexpr int NTAP = (T + NW - 1) / NW;         // taps per wave (KS=3: 7 with 4 waves, 4 with 8)
    constexpr int NV = TH * TW;                     // coarse voxels per tile
    STX_DYN_SMEM(smem);
    float* ftile = reinterpret_cast<float*>(smem);          // [ED*EH*EWS][32]
    float* ctile = ftile + ED * EH * EWS * 32;              // [NV][32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, half = lane >> 5;
    const int ncf = a.CF / 32;
    const int cfb = blockIdx.y % ncf, ccb = blockIdx.y / ncf;

    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) acc[t] = zero16();

    // Tile staging is split into "global -> registers" and "registers -> LDS".  With PIPE the loads
    // of the next tile are issued right before the current tile's MFMA loop and only written to LDS
    // after it (one workgroup per CU, 2 waves per SIMD: measured, more fp32-MFMA waves per SIMD lower
    // the matrix-pipe throughput); without PIPE both halves run back to back (small 1x1 case).
    // The loads go through buffer descriptors (one per input plane of the tile, wave-uniform: base = the tile's origin in
    // that plane, possibly in front of the plane for tiles on the low edges) with per-lane byte offsets that are computed
    // ONCE per kernel; voxels outside the volume get the out-of-range offset and read zeros through the bounds check, planes
    // outside [0, Df) get an empty descriptor.  Per tile that leaves two compares and a select per float4 -- the first
    // version recomputed a 64-bit address with four bounds tests and a branch per float4, in lock-step on all eight waves
    // (GPU call H2 of round 3: 0.817 ms with, 0.687 ms without the staging at 32 -> 32 L0).
    constexpr int NPL = (EH * EW * 8 + NTHR - 1) / NTHR;         // float4 per lane and fine-tile plane
    constexpr int NFL = ED * NPL;                                //                 fine tile
    constexpr int NCL = (NV * 8 + NTHR - 1) / NTHR;              //                 coarse tile
    float4 sf[NFL], sc[NCL];
    unsigned fbase[NPL], cbase[NCL];                             // byte offsets relative to the tile origin
    int fhw[NPL], chw[NCL];                                      // (row << 16 | column) inside the tile, -1 = no element
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int idx = tid + k * NTHR;
        const int v = idx >> 3, f = idx & 7;
        const int wx = v % EW, hy = v / EW;
        fhw[k] = idx < EH * EW * 8 ? (hy << 16 | wx) : -1;
        fbase[k] = (unsigned)(((hy * a.Wf + wx) * a.CF + cfb * 32 + 4 * f) * 4);
    }
#pragma unroll
    for (int k = 0; k < NCL; ++k) {
        const int idx = tid + k * NTHR;
        const int v = idx >> 3, f = idx & 7;
        chw[k] = idx < NV * 8 ? ((v / TW) << 16 | (v % TW)) : -1;
        cbase[k] = (unsigned)((((v / TW) * a.Wc + v % TW) * a.CC + ccb * 32 + 4 * f) * 4);
    }
    const long long fplane = (long long)a.Hf * a.Wf, cplane = (long long)a.Hc * a.Wc;
    auto load_tile = [&](int tile, bool on) {
        int r = tile;
        const int wt = r % a.nWt; r /= a.nWt;
        const int ht = r % a.nHt; r /= a.nHt;
        const int od = r % a.Dc;
        const int b = r / a.Dc;
        const int oh0 = ht * TH, ow0 = wt * TW;
        const int id0 = od * S - PAD, ih0 = oh0 * S - PAD, iw0 = ow0 * S - PAD;
        unsigned fvo[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int gh = ih0 + (fhw[k] >> 16), gw = iw0 + (fhw[k] & 0xffff);
            fvo[k] = (fhw[k] >= 0 && gh >= 0 && gh < a.Hf && gw >= 0 && gw < a.Wf) ? fbase[k] : STX_BUF_OOB;
        }
        const long long forg = (long long)ih0 * a.Wf + iw0;     // tile origin inside a plane (voxels; negative on the low edges)
#pragma unroll
        for (int dz = 0; dz < ED; ++dz) {
            const int gd = id0 + dz;
            const bool in = on && gd >= 0 && gd < a.Df;
            const stx_bufrsrc rs = stx_make_rsrc(a.f + (((long long)b * a.Df + (in ? gd : 0)) * fplane + forg) * a.CF,
                                                 in ? (unsigned)((fplane - forg) * a.CF * 4) : 0u);
#pragma unroll
            for (int k = 0; k < NPL; ++k) sf[dz * NPL + k] = stx_buf_ld4(rs, fvo[k], 0u);
        }
        const long long corg = (long long)oh0 * a.Wc + ow0;
        const stx_bufrsrc rc = stx_make_rsrc(a.c + (((long long)b * a.Dc + od) * cplane + corg) * a.CC,
                                             on ? (unsigned)((cplane - corg) * a.CC * 4) : 0u);
#pragma unroll
        for (int k = 0; k < NCL; ++k) {
            const bool ok = chw[k] >= 0 && oh0 + (chw[k] >> 16) < a.Hc && ow0 + (chw[k] & 0xffff) < a.Wc;
            sc[k] = stx_buf_ld4(rc, ok ? cbase[k] : STX_BUF_OOB, 0u);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int dz = 0; dz < ED; ++dz)
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const int wx = fhw[k] & 0xffff, hy = fhw[k] >> 16;
                const int slot = (S == 2) ? (wx & 1) * EWH + (wx >> 1) : wx;
                const int f = (tid + k * NTHR) & 7;
                if (fhw[k] >= 0) stx_st4(ftile + ((dz * EH + hy) * EWS + slot) * 32 + 4 * f, sf[dz * NPL + k]);
            }
#pragma unroll
        for (int k = 0; k < NCL; ++k) {
            const int idx = tid + k * NTHR;
            if (idx < NV * 8) stx_st4(ctile + (idx >> 3) * 32 + 4 * (idx & 7), sc[k]);
        }
    };

    if (PIPE && (int)blockIdx.x < a.ntiles) load_tile(blockIdx.x, a.ablate != 1);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        if (!PIPE) load_tile(tile, a.ablate != 1);
        __syncthreads();                 // every wave is done with the previous tile in LDS
        if (a.ablate != 1) store_tile();
        __syncthreads();
        if (PIPE) {                      // (an empty descriptor behind the last tile: no branch in front of the MFMA loop)
            const int nxt = tile + (int)gridDim.x;
            load_tile(nxt < a.ntiles ? nxt : tile, nxt < a.ntiles && a.ablate != 1);
        }
        if (a.ablate == 2) continue;
        if (KS == 1) {
            // waves split the voxel pairs
            for (int p = wave; p < NV / 2; p += NW) {
                const int v = 2 * p + half;
                const int lw = v % TW, lh = v / TW;
                const float bv = ctile[v * 32 + i];
                const float av = ftile[(lh * EWS + lw) * 32 + i];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[0], 0, 0, 0);
            }
        } else {
            // address of operand A = (voxel part) + (tap part): tap offsets live in NTAP registers, the
            // operands of voxel pair p+1 are fetched while pair p is multiplied (pinned by the fences).
            int toff[NTAP];
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int tap = (t * NW + wave < T) ? t * NW + wave : 0;
                const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                toff[t] = (S == 2) ? ((kd * EH + kh) * EWS + (kw & 1) * EWH + (kw >> 1)) * 32
                                   : ((kd * EH + kh) * EWS + kw) * 32;
            }
            // (Instantiating the loop per tap count -- three waves hold four taps, five hold three -- behind a wave-uniform
            //  branch removes the exec-mask save / restore around the conditional MFMAs but measured SLOWER, GPU call L of
            //  round 3: 32 -> 32 L0 0.795 -> 0.865 ms, 64 -> 64 L1 0.432 -> 0.476 ms; two unrolled loop bodies per workgroup.)
            float av[2][NTAP], bv[2];
            auto load_pair = [&](int p, int buf) {
                const int v = 2 * p + half;
                const int lw = v % TW, lh = v / TW;
                const int vb = ((lh * S) * EWS + lw) * 32 + i;
                bv[buf] = ctile[v * 32 + i];
#pragma unroll
                for (int t = 0; t < NTAP; ++t) av[buf][t] = ftile[vb + toff[t]];
            };
            auto mma_pair = [&](int buf) {
#pragma unroll
                for (int t = 0; t < NTAP; ++t)
                    if (t * NW + wave < T) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][t], bv[buf], acc[t], 0, 0, 0);
            };
            load_pair(0, 0);
            for (int p = 0; p < NV / 2; p += 2) {
                load_pair(p + 1, 1);
                STX_SCHED_BARRIER();
                mma_pair(0);
                STX_SCHED_BARRIER();
                load_pair((p + 2 < NV / 2) ? p + 2 : 0, 0);     // unconditional (wraps): keeps the lgkmcnt
                STX_SCHED_BARRIER();                            // pipeline one stage deep on every trip
                mma_pair(1);
                STX_SCHED_BARRIER();
            }
        }
    }
    // partial slab: KS=3: [blockIdx.y][blockIdx.x][tap][cf 32][cc 32]; KS=1: [..][wave][32][32]
    constexpr int ROWS = (KS == 1) ? NW : T;
    float* dst = a.slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * ROWS * 1024;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const int row = (KS == 1) ? wave : t * NW + wave;
        if (KS == 1 ? (t == 0) : (row < T)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cf = (r & 3) + 8 * (r >> 2) + 4 * half;
                dst[(size_t)row * 1024 + cf * 32 + i] = acc[t][r];
            }
        }
    }
}

// dW[cc][cf][tap] = sum over chunks (and waves for KS=1) of the slab, in a fixed order (deterministic): 16 lanes x float4
// cover the workgroup's 64 outputs, the four 16-lane groups of the four waves take every 16th slab row (16 rows of 256
// contiguous bytes in flight per wave instruction), then one LDS round.
__global__ __launch_bounds__(CONV_THREADS) void conv3d_wgrad_reduce4_kernel(const float* __restrict__ slab,
                                                                            float* __restrict__ dw, int CF, int CC,
                                                                            int T, int nchunks, int rows_per_chunk) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = wave * 4 + (lane >> 4), ql = lane & 15;
    const int idx0 = blockIdx.x * 64 + ql * 4;                     // over [pair][tap][cf32][cc32], 4 consecutive cc
    const int cc_l = idx0 & 31, cf_l = (idx0 >> 5) & 31;
    const int tap = (idx0 >> 10) % T, pair = (idx0 >> 10) / T;
    const float* p;
    size_t stride;
    int n;
    if (rows_per_chunk == T) {
        p = slab + ((size_t)pair * nchunks * T + tap) * 1024 + cf_l * 32 + cc_l;
        stride = (size_t)T * 1024; n = nchunks;
    } else {                                                        // KS=1: one slab row per wave of the producer
        p = slab + ((size_t)pair * nchunks * rows_per_chunk) * 1024 + cf_l * 32 + cc_l;
        stride = 1024; n = nchunks * rows_per_chunk;
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int c = part; c < n; c += 16) {
        const float4 v = stx_ld4(p + (size_t)c * stride);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    stx_st4(&red[part][ql * 4], s);
    __syncthreads();
    if (wave == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][lane];
        const int idx = blockIdx.x * 64 + lane;
        const int cc = idx & 31, cf = (idx >> 5) & 31;
        const int ncf = CF / 32, cfb = pair % ncf, ccb = pair / ncf;   // (tap and pair are the same for all 64 outputs)
        dw[((size_t)(ccb * 32 + cc) * CF + cfb * 32 + cf) * T + tap] = t;
    }
}

template <typename K>
int launch_with_lds(K kernel, dim3 grid, size_t lds, hipStream_t st, ConvArgs a) {
    if (lds > 160 * 1024) return stx_set_error(STX_ERR_ARG, "conv3d: LDS tile of %zu B exceeds 160 KiB", lds);
    if (lds > 64 * 1024)
        hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    static const bool report = getenv("STX_REPORT_OCCUPANCY") != nullptr;    // (diagnostic: resident workgroups per CU)
    if (report) {
        int nb = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kernel, CONV_THREADS, lds);
        fprintf(stderr, "[stx] conv3d launch: grid %u x %u, %zu B LDS, %d workgroups per CU\n", grid.x, grid.y, lds, nb);
    }
    hipLaunchKernelGGL(kernel, grid, dim3(CONV_THREADS), lds, st, a);
    return 0;
}

constexpr int CONV_TD = 2, CONV_TH = 2;

template <int KS, int S>
size_t conv_lds_bytes(int CK, int NT) {
    const int ED = (CONV_TD - 1) * S + KS, EH = (CONV_TH - 1) * S + KS, EW = 31 * S + KS;
    const int EWS = (S == 2) ? 2 * ((EW + 1) / 2) : EW;
    size_t t = (size_t)ED * EH * EWS * (CK + 4) * 4;
    size_t red = (size_t)4 * NT * 32 * 2 * 4;
    return t > red ? t : red;
}

// Persistent pipelined launch: 256 * (workgroups per CU the LDS tile admits, at most 2) workgroups share the tiles evenly.
template <typename K>
int launch_persistent(K kernel, size_t lds, hipStream_t st, ConvArgs a, int ntiles, int grid) {
    if (lds > 160 * 1024) return stx_set_error(STX_ERR_ARG, "conv3d: LDS tile of %zu B exceeds 160 KiB", lds);
    if (lds > 64 * 1024)
        hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(CONV_THREADS), lds, st, a, ntiles);
    return 0;
}

// Implicit-GEMM kernel for NT column blocks of 32 output channels and K chunks of CK input channels.
template <int KS, int S>
int conv_dispatch(const ConvArgs& a, int NT, int CK, dim3 grid, hipStream_t st) {
    const size_t lds = conv_lds_bytes<KS, S>(CK, NT);
#define CONV_CASE(NT_, CK_)                                                                                        \
    if (NT == NT_ && CK == CK_)                                                                                    \
        return launch_with_lds(conv3d_igemm_kernel<KS, S, CONV_TD, CONV_TH, NT_, CK_>, grid, lds, st, a);
    if constexpr (KS == 1) { CONV_CASE(1, 32) CONV_CASE(2, 32) CONV_CASE(4, 32) }      // (conv_pick_ck: 3x3x3 takes 8-channel chunks only)
    if constexpr (KS == 3) {
        // 64 output channels: two rows x one column block per wave (STX_CONV_WN, see the kernel's comment)
        if (NT == 2 && CK == 8 && stx_tune(STX_TUNE_CONV_WN) >= 2)
            return launch_with_lds(conv3d_igemm_kernel<KS, S, CONV_TD, CONV_TH, 2, 8, 4, 2>, grid, lds, st, a);
    }
    CONV_CASE(1, 8) CONV_CASE(2, 8) CONV_CASE(4, 8)
#undef CONV_CASE
    return stx_set_error(STX_ERR_ARG, "conv3d: unsupported NT=%d CK=%d", NT, CK);
}

}  // namespace

// Cin chunk of the implicit-GEMM kernels (Cin is a multiple of 8).  3x3x3: 8-channel chunks (GPU call T of round 2: the LDS
// tile shrinks to 26 KB and the register budget to ~100, so up to four workgroups share a CU instead of two: 64->64 L1
// 0.486 -> 0.430 ms, 128->128 L2 0.242 -> 0.232 ms); 1x1x1 (HBM-bound, no halo): 32-channel chunks where Cin allows.
static int conv_pick_ck(int Cin, int ks) { return (ks == 1 && Cin % 32 == 0) ? 32 : 8; }
// 32-wide MFMA column blocks used for N output channels (1, 2 or 4).
static int conv_nt(int N) { return N <= 32 ? 1 : (N <= 64 ? 2 : 4); }

extern "C" long long stx_conv3d_packed_floats(int K, int N, int T) {
    return (long long)T * (K / 8) * conv_nt(N) * 256;
}

extern "C" int stx_conv3d_pack_weight(const float* w, float* wp, int A, int Bd, int T, int mode, void* stream) {
    stx_begin();
    STX_REQUIRE(w && wp && A > 0 && Bd > 0 && (T == 1 || T == 27), "conv3d_pack_weight: bad args");
    STX_REQUIRE(mode >= 0 && mode <= 2, "conv3d_pack_weight: mode %d", mode);
    const int K = (mode == 0) ? Bd : A, N = (mode == 0) ? A : Bd;
    STX_REQUIRE(K % 8 == 0 && N <= 128, "conv3d_pack_weight: GEMM-K channels (%d) must be a multiple of 8, N (%d) <= 128", K, N);
    const int NT = conv_nt(N);
    const size_t total = (size_t)T * (K / 8) * NT * 256;
    hipLaunchKernelGGL(conv3d_pack_kernel, dim3((unsigned)((total + CONV_THREADS - 1) / CONV_THREADS)),
                       dim3(CONV_THREADS), 0, (hipStream_t)stream, w, wp, A, Bd, T, mode, K, N, NT, total);
    return stx_check_launch("conv3d_pack_weight");
}

// Workgroups of the march kernel: one per CU (its LDS admits only one), fewer for tiny volumes so that a workgroup still
// gets a few planes per 2-plane prologue.
static int march_wgs(long long units) {
    long long g = 256;
    if (g > units / 3) g = units / 3;
    return g < 1 ? 1 : (int)g;
}

// Rows of the `stats` partial slab stx_conv3d_fwd writes per batch item (>= its workgroup count / B for every kernel it
// may choose; unused rows are zero-filled).
extern "C" int stx_conv3d_fwd_blocks(int Do, int Ho, int Wo) {
    stx_begin();
    const int a = stx_cdiv(Do, CONV_TD) * stx_cdiv(Ho, CONV_TH) * stx_cdiv(Wo, 32);
    const int m = Do * stx_cdiv(Ho, 4) * stx_cdiv(Wo, 32);
    const int m16 = Do * stx_cdiv(Ho, 8) * stx_cdiv(Wo, 16);           // upper bound for the march kernel's workgroups
    return a > m ? (a > m16 ? a : m16) : (m > m16 ? m : m16);
}
extern "C" int stx_deconv3d_fwd_blocks(int Di, int Hi, int Wi) {
    stx_begin(); return Di * stx_cdiv(Hi, 2) * stx_cdiv(Wi, 32); }

// Which kernel stx_conv3d_fwd runs for a shape, and how many rows of the BN-statistics slab it writes (one per workgroup
// that owns outputs): the single source for the launch below and for stx_conv3d_fwd_stat_rows.
struct ConvPlan { int kind; int march_wgs; int pgrid; long long rows; };     // kind: 0 march, 1 pipelined (128 channels), 2 implicit GEMM
static int persistent_grid(size_t lds, long long ntiles) {
    int per_cu = (int)((160 * 1024) / (lds + 512));
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
    const long long g = 256ll * per_cu;
    return (int)(g > ntiles ? ntiles : g);
}
static ConvPlan conv_plan(int B, int Di, int Hi, int Wi, int Cin, int Cout, int ks, int stride) {
    const int pad = ks / 2;
    const int Do = (Di + 2 * pad - ks) / stride + 1, Ho = (Hi + 2 * pad - ks) / stride + 1, Wo = (Wi + 2 * pad - ks) / stride + 1;
    ConvPlan p{2, 0, 0, 0};
    // (STX_CONV_L1_MARCH: 64 -> 64 as 2 x 2 slices too -- the hourglass's second level, otherwise on the implicit-GEMM kernel)
    if (ks == 3 && stride == 1 && ((Cin == 32 && Cout <= 64) || (Cin == 64 && Cout <= 32) ||
                                   (Cin == 64 && Cout <= 64 && stx_tune(STX_TUNE_CONV_L1_MARCH)))) {
        const long long ncols = (long long)B * stx_cdiv(Ho, MW2_TH) * stx_cdiv(Wo, MW2_MW);
        // (planes are addressed through buffer descriptors with 32-bit byte offsets)
        if (ncols * Do < (1ll << 31) && (long long)Hi * Wi * Cin * 4 < (1ll << 31) && (long long)Ho * Wo * Cout * 4 < (1ll << 31)) {
            p.kind = 0; p.march_wgs = march_wgs(ncols * Do); p.rows = p.march_wgs;
            return p;
        }
    }
    const long long tiles = (long long)stx_cdiv(Do, CONV_TD) * stx_cdiv(Ho, CONV_TH) * stx_cdiv(Wo, 32);
    if (conv_nt(Cout) == 4 && ks == 3 && tiles * B < (1ll << 31)) {
        const size_t lds = stride == 1 ? conv_lds_bytes<3, 1>(8, 4) : conv_lds_bytes<3, 2>(8, 4);
        p.kind = 1; p.pgrid = persistent_grid(lds, tiles * B); p.rows = p.pgrid;
        return p;
    }
    p.rows = tiles * B;
    return p;
}

// Rows of the `stats` slab stx_conv3d_fwd writes for this call ([rows][2][Cout], every row written, nothing beyond):
// allocate exactly this many and hand them all to stx_bn_finalize.
extern "C" long long stx_conv3d_fwd_stat_rows(int B, int Di, int Hi, int Wi, int Cin, int Cout, int ks, int stride) {
    stx_begin();
    if (B < 1 || Di < 1 || Hi < 1 || Wi < 1 || Cin < 1 || Cout < 1 || !((ks == 3 && (stride == 1 || stride == 2)) || (ks == 1 && stride == 1)))
        return 0;
    return conv_plan(B, Di, Hi, Wi, Cin, Cout, ks, stride).rows;
}

extern "C" int stx_conv3d_fwd(const float* x, const float* wp, float* out, const float* scale, const float* bias,
                              const float* residual, float* stats, int B, int Di, int Hi, int Wi, int Cin, int Cout,
                              int ks, int stride, int relu, void* stream) {
    stx_begin();
    STX_REQUIRE(x && wp && out && B > 0 && Di > 0 && Hi > 0 && Wi > 0, "conv3d_fwd: bad shape");
    STX_REQUIRE(Cin % 8 == 0 && Cout >= 1 && Cout <= 128, "conv3d_fwd: Cin=%d (need %%8) Cout=%d (need <=128)", Cin, Cout);
    STX_REQUIRE((ks == 3 && (stride == 1 || stride == 2)) || (ks == 1 && stride == 1),
                "conv3d_fwd: kernel %d stride %d unsupported", ks, stride);
    ConvArgs a;
    a.x = x; a.wp = wp; a.out = out; a.scale = scale; a.bias = bias; a.residual = residual; a.stats = stats;
    a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.relu = relu;
    const int pad = ks / 2;
    a.Do = (Di + 2 * pad - ks) / stride + 1;
    a.Ho = (Hi + 2 * pad - ks) / stride + 1;
    a.Wo = (Wi + 2 * pad - ks) / stride + 1;
    a.nDt = stx_cdiv(a.Do, CONV_TD); a.nHt = stx_cdiv(a.Ho, CONV_TH); a.nWt = stx_cdiv(a.Wo, 32);
    const int NT = conv_nt(Cout);
    hipStream_t st = (hipStream_t)stream;
    const ConvPlan plan = conv_plan(B, Di, Hi, Wi, Cin, Cout, ks, stride);       // (the slab has plan.rows rows, all written)
    // March kernel (weights resident in LDS, input-stationary planes) for the 3x3x3 stride-1 layers in 32 x 32 channel
    // slices: 32 -> <=32 directly; 64 -> <=32 as two K slices (the second adds the first's partial sums and applies the
    // epilogue); 32 -> <=64 as two N slices (dgrad of the 64 -> 32 layer).
    if (plan.kind == 0) {
        MarchArgs m2;
        m2.c = a;
        m2.c.nHt = stx_cdiv(a.Ho, MW2_TH);
        m2.c.nWt = stx_cdiv(a.Wo, MW2_MW);
        m2.ncols = B * m2.c.nHt * m2.c.nWt;
        m2.ablate = stx_tune(STX_TUNE_MARCH_ABLATE);
        {
            const int nb2 = plan.march_wgs;
            const size_t lds2 = ((size_t)MW2_WFLOATS + 2 * (size_t)MW2_SLOT) * 4;
            // STX_MARCH_BS: 1 = one accumulator per (output, input plane), summed in the epilogue; 0 = one sequential chain per
            // output (GPU call F of round 3, 32 -> 32 L0: 0.757 -> 0.767 ms).  STX_MARCH_EPI: 1 = straight-line epilogue for
            // the launches that admit it, 0 = general epilogue always.
            const int bs = stx_tune(STX_TUNE_MARCH_BS), epi_fast = stx_tune(STX_TUNE_MARCH_EPI);
            void (*mk_gen)(MarchArgs) = bs ? conv3d_marchw_kernel<1, 1> : conv3d_marchw_kernel<0, 1>;
            void (*mk_plain)(MarchArgs) = bs ? conv3d_marchw_kernel<1, 0> : conv3d_marchw_kernel<0, 0>;
            void (*mk_acc)(MarchArgs) = bs ? conv3d_marchw_kernel<1, 2> : conv3d_marchw_kernel<0, 2>;
            void (*mk_res)(MarchArgs) = bs ? conv3d_marchw_kernel<1, 3> : conv3d_marchw_kernel<0, 3>;
            void (*mk_raw)(MarchArgs) = bs ? conv3d_marchw_kernel<1, 4> : conv3d_marchw_kernel<0, 4>;
            void (*mk_rawf)(MarchArgs) = bs ? conv3d_marchw_kernel<1, 5> : conv3d_marchw_kernel<0, 5>;
            hipFuncSetAttribute((const void*)mk_rawf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            const bool whole = a.Ho % MW2_TH == 0 && a.Wo % MW2_MW == 0;
            hipFuncSetAttribute((const void*)mk_raw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            hipFuncSetAttribute((const void*)mk_res, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            hipFuncSetAttribute((const void*)mk_plain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            hipFuncSetAttribute((const void*)mk_acc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            void (*mk)(MarchArgs) = mk_gen;
            hipFuncSetAttribute((const void*)mk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            m2.xs = Cin; m2.os = Cout; m2.wq_total = Cin / 8; m2.wnt_total = conv_nt(Cout);
            const int nk = Cin / 32, nn = stx_cdiv(Cout, 32);
            for (int ns = 0; ns < nn; ++ns)
                for (int kslice = 0; kslice < nk; ++kslice) {
                    MarchArgs m = m2;
                    m.xo = 32 * kslice; m.wq_off = 4 * kslice;
                    m.oo = 32 * ns; m.wnt_off = ns; m.ncout = Cout - 32 * ns < 32 ? Cout - 32 * ns : 32;
                    m.acc_in = kslice > 0 ? out : nullptr;
                    if (kslice + 1 < nk) {      // partial sums only: raw accumulators to `out`, epilogue in the last slice
                        m.c.scale = nullptr; m.c.bias = nullptr; m.c.residual = nullptr; m.c.stats = nullptr; m.c.relu = 0;
                    }
                    // (the straight-line epilogues read the partial sums through the OUTPUT descriptor: acc_in is `out`)
                    const bool plain = epi_fast && !(m.c.residual && m.acc_in) && m.c.relu < 2;
                    const bool raw = !m.acc_in && !m.c.residual && !m.c.scale && !m.c.bias && m.c.relu == 0;
                    hipLaunchKernelGGL(plain ? (raw ? (whole && m.ncout == 32 ? mk_rawf : mk_raw) : m.acc_in ? mk_acc
                                                        : (m.c.residual ? mk_res : mk_plain)) : mk,
                                       dim3(nb2), dim3(256), lds2, st, m);
                }
            return stx_check_launch("conv3d_fwd(march)");
        }
    }
    STX_REQUIRE(halo_range_ok(Di, Hi, Wi, Cin), "conv3d_fwd: a batch item of %d x %d x %d x %d floats exceeds the 2 GiB "
                "buffer-descriptor range", Di, Hi, Wi, Cin);
    // stride 2 stages a (2TD+1)(2TH+1)x65-voxel input tile: 8-channel K chunks (79 KB padded, 53 KB dense).
    const int CK = (stride == 2) ? 8 : conv_pick_ck(Cin, ks);
    dim3 grid(a.nDt * a.nHt * a.nWt, B);
    int rc;
    // 128 output channels: persistent software-pipelined kernel (its staging registers cost the second resident workgroup,
    // so it wins only there: measured at the 576x960 shapes 128->128 L2 0.283 -> 0.245 ms, stride-2 64->128 0.176 -> 0.154 ms,
    // but 64->64 L1 0.470 -> 0.537 ms, stride-2 32->64 0.327 -> 0.353 ms: round 2, profiles/r02_conv_ab.txt)
    if (plan.kind == 1) {
        const long long nt_all = (long long)grid.x * B;
        {
            // this kernel writes row blockIdx.x of the stats slab
            const size_t lds = stride == 1 ? conv_lds_bytes<3, 1>(8, 4) : conv_lds_bytes<3, 2>(8, 4);
            // (STX_CONV_WN: 2 = the default, only the 64-channel kernels change their wave grid; 3 / 4 = this kernel as two rows x two
            //  blocks / four rows x one block per wave: 0.242 / 0.239 vs 0.235 ms on 128->128 L2 -- no gain, kept as switches)
            const int wn = stx_tune(STX_TUNE_CONV_WN) == 4 ? 4 : (stx_tune(STX_TUNE_CONV_WN) == 3 ? 2 : 1);
            if (wn >= 4)
                rc = stride == 1 ? launch_persistent(conv3d_pgemm_kernel<3, 1, CONV_TD, CONV_TH, 4, 8, 4>, lds, st, a, (int)nt_all, plan.pgrid)
                                 : launch_persistent(conv3d_pgemm_kernel<3, 2, CONV_TD, CONV_TH, 4, 8, 4>, lds, st, a, (int)nt_all, plan.pgrid);
            else if (wn >= 2)
                rc = stride == 1 ? launch_persistent(conv3d_pgemm_kernel<3, 1, CONV_TD, CONV_TH, 4, 8, 2>, lds, st, a, (int)nt_all, plan.pgrid)
                                 : launch_persistent(conv3d_pgemm_kernel<3, 2, CONV_TD, CONV_TH, 4, 8, 2>, lds, st, a, (int)nt_all, plan.pgrid);
            else
            rc = stride == 1 ? launch_persistent(conv3d_pgemm_kernel<3, 1, CONV_TD, CONV_TH, 4, 8>, lds, st, a, (int)nt_all, plan.pgrid)
                             : launch_persistent(conv3d_pgemm_kernel<3, 2, CONV_TD, CONV_TH, 4, 8>, lds, st, a, (int)nt_all, plan.pgrid);
            if (rc) return rc;
            return stx_check_launch("conv3d_fwd(pipelined)");
        }
    }
    if (ks == 3 && stride == 1) rc = conv_dispatch<3, 1>(a, NT, CK, grid, st);
    else if (ks == 3 && NT == 2 && stx_tune(STX_TUNE_CONV_S2_DENSE)) {
        // stride 2, 32 -> 64 (the first convolution of every hourglass): dense (un-padded) LDS tile, 53 KB: three workgroups
        // per CU instead of two (GPU call A of round 3: 0.332 -> 0.312 ms)
        if (stx_tune(STX_TUNE_CONV_WN) >= 2)
            rc = launch_with_lds(conv3d_igemm_kernel<3, 2, CONV_TD, CONV_TH, 2, 8, 0, 2>, grid, (size_t)5 * 5 * 66 * 8 * 4, st, a);
        else
        rc = launch_with_lds(conv3d_igemm_kernel<3, 2, CONV_TD, CONV_TH, 2, 8, 0>, grid, (size_t)5 * 5 * 66 * 8 * 4, st, a);
    }
    else if (ks == 3) rc = conv_dispatch<3, 2>(a, NT, CK, grid, st);
    else rc = conv_dispatch<1, 1>(a, NT, CK, grid, st);
    if (rc) return rc;
    return stx_check_launch("conv3d_fwd");
}

extern "C" int stx_deconv3d_fwd(const float* x, const float* wp, float* out, const float* scale, const float* bias,
                                const float* residual, float* stats, int B, int Di, int Hi, int Wi, int Cin, int Cout,
                                int Do, int Ho, int Wo, int relu, void* stream) {
    stx_begin();
    STX_REQUIRE(x && wp && out && B > 0 && Di > 0 && Hi > 0 && Wi > 0, "deconv3d_fwd: bad shape");
    STX_REQUIRE(Cin % 32 == 0 && Cout >= 1 && Cout <= 64, "deconv3d_fwd: Cin=%d (need %%32) Cout=%d (need <=64)", Cin, Cout);
    STX_REQUIRE(Do <= 2 * Di && Do >= 2 * Di - 1 && Ho <= 2 * Hi && Ho >= 2 * Hi - 1 && Wo <= 2 * Wi && Wo >= 2 * Wi - 1,
                "deconv3d_fwd: output dims must be 2*in or 2*in-1");
    ConvArgs a;
    a.x = x; a.wp = wp; a.out = out; a.scale = scale; a.bias = bias; a.residual = residual; a.stats = stats;
    a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.relu = relu;
    a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    a.nDt = Di; a.nHt = stx_cdiv(Hi, 2); a.nWt = stx_cdiv(Wi, 32);
    STX_REQUIRE(halo_range_ok(Di, Hi, Wi, Cin), "deconv3d_fwd: a batch item of %d x %d x %d x %d floats exceeds the 2 GiB "
                "buffer-descriptor range", Di, Hi, Wi, Cin);
    dim3 grid(a.nDt * a.nHt * a.nWt, B);
    hipStream_t st = (hipStream_t)stream;
    // 8-channel K chunks (9.5 KB of LDS: four workgroups per CU; calls S/T of round 2: 32-channel chunks 0.153 / 0.266 ms,
    // 16-channel 0.157 / 0.280, 8-channel 0.140 / 0.251 for the two GwcNet shapes)
    const size_t lds = (size_t)2 * 3 * 33 * (8 + 4) * 4;
    int rc = conv_nt(Cout) == 1 ? launch_with_lds(deconv3d_igemm_kernel<1, 8>, grid, lds, st, a)
                                : launch_with_lds(deconv3d_igemm_kernel<2, 8>, grid, lds, st, a);
    if (rc) return rc;
    return stx_check_launch("deconv3d_fwd");
}

// (GPU call P of round 3, 32 -> 32 L0 / 64 -> 32 L0, 4 x 16-voxel tiles with eight waves = 0.800 / 1.575 ms: 8 x 16-voxel tiles
//  0.839 / 1.652; four waves, one per SIMD with seven taps each, 0.915 / 1.805 (4 x 16) and 0.883 / 1.743 (8 x 16).  Fewer
//  barriers per MFMA do not help and one wave per SIMD hurts: the variants were removed again.  GPU call S: TWO tiles in LDS --
//  tile i+1 written to the other buffer behind the first operand reads of tile i, one barrier per tile instead of "barrier,
//  store, barrier" -- 0.801 -> 0.845 / 1.568 -> 1.654 ms (250 instead of 187 VGPRs): removed as well.  GPU call W: the 128-VGPR
//  form (pair loop rolled, no spills) with TWO workgroups per CU, un-pipelined or pipelined staging: 0.800 -> 0.828 / 0.834 ms,
//  1.571 -> 1.637 ms -- four fp32-MFMA waves per SIMD run the matrix pipe slower than two, as round 1 measured: removed.
//  Counters (profiles/r03_pmc_instruction_mix.txt): 3.4 scalar instructions and 0.7 branches per MFMA (the exec-mask save /
//  restore around the fourth tap of three of the eight waves), SIMDs loaded 7 / 7 / 7 / 6 taps.  A BALANCED split -- every wave
//  three whole taps plus three of the 24 (tap 24-26, pair mod 8) combinations, walked in a per-wave rotated pair order so that
//  all waves run the same mask-free code, 27 MFMAs per eight pairs each -- removed both, but pays for the rotation with vector
//  address arithmetic: 3.4 VALU per MFMA 0.802 -> 0.887 ms, 1.6 VALU per MFMA 0.870 ms (all six shapes 8-15 % slower, GPU calls
//  BAL / BAL2): in this loop a VALU instruction costs far more than a scalar one.  Removed; the record stays.)
// Weight gradient: spatial tile (coarse voxels), whether tile staging is software-pipelined, workgroups along split-K.
// 3x3x3 stride 1 takes 4 x 16 instead of 2 x 32 voxels when the narrower tile wastes fewer columns (W' = 240 = 15 x 16 =
// 7.5 x 32: 6 % fewer MFMAs); the pipelined staging pays for stride 2 and for the one- or two-pair L0 layers (measured: the
// L1 / L2 stride-1 layers, few tiles per chunk, prefer the plain loop).
static bool wgrad_pipe(int ks, int stride, int npairs) { return ks == 3 && (stride == 2 || npairs <= 2); }
static void wgrad_tile(int ks, int stride, int Wc, bool pipe, int* TH, int* TW) {
    *TH = pipe ? 4 : 2; *TW = (stride == 2) ? 16 : 32;
    if (ks == 3 && stride == 1 && stx_cdiv(Wc, 16) * 16 < stx_cdiv(Wc, 32) * 32) { *TH = 4; *TW = 16; }
}
static int wgrad_chunks(int ntiles, int npairs, bool pipe) {
    int c = (pipe ? 256 : 512) / npairs;                            // workgroups in total
    if (c < 1) c = 1;
    if (c > ntiles) c = ntiles;
    const int forced = stx_tune(STX_TUNE_WGRAD_GRID);
    if (forced > 0 && forced < c) c = forced;
    return c;
}

// march form of the 3x3x3 stride-1 weight gradient (wgrad_march.hip)
struct StxWgradBn {        // (wgrad_march.hip)
    const float* z; float* dz;
    const float *scale, *shift, *mean, *invstd, *gamma, *sum_g, *sum_gx;
    float inv_n; int act;
};
int stx_wgrad_march_launch(const float* x, const float* gy, float* slab, int B, int D, int H, int W, int CF, int CC, int nchunks,
                           void* stream, const StxWgradBn* bn);
int stx_wgrad_march_chunks(int B, int D, int H, int W, int npairs);
int stx_wgrad_march_s2_launch(const float* f, const float* c, float* slab, int B, int Df, int Hf, int Wf, int CF, int Dc, int Hc,
                              int Wc, int CC, int nchunks, void* stream);

extern "C" long long stx_conv3d_wgrad_workspace_floats(int B, int Dc, int Hc, int Wc, int CF, int CC, int ks,
                                                       int stride) {
    const int npairs = (CF / 32) * (CC / 32);
    if (npairs < 1 || B < 1 || Dc < 1 || Hc < 1 || Wc < 1) return 0;   // stx_conv3d_wgrad rejects these shapes
    const int rows = (ks == 1) ? 8 : 27;
    const bool pipe = wgrad_pipe(ks, stride, npairs);
    int TH, TW;
    wgrad_tile(ks, stride, Wc, pipe, &TH, &TW);
    int c = wgrad_chunks(B * Dc * stx_cdiv(Hc, TH) * stx_cdiv(Wc, TW), npairs, pipe);
    if (ks == 3) {                                                      // (either kernel may serve the call: STX_WGRAD_MARCH)
        const int cm = stx_wgrad_march_chunks(B, Dc, Hc, Wc, npairs);
        if (cm > c) c = cm;
    }
    return (long long)npairs * c * rows * 1024;
}

extern "C" int stx_conv3d_wgrad(const float* f, const float* c, float* dw, float* workspace, int B, int Df, int Hf,
                                int Wf, int CF, int Dc, int Hc, int Wc, int CC, int ks, int stride, void* stream) {
    stx_begin();
    STX_REQUIRE(f && c && dw && workspace && B > 0, "conv3d_wgrad: null operand");
    STX_REQUIRE(CF >= 32 && CC >= 32 && CF % 32 == 0 && CC % 32 == 0,
                "conv3d_wgrad: channel counts (%d, %d) must be positive multiples of 32", CF, CC);
    STX_REQUIRE(Dc > 0 && Hc > 0 && Wc > 0, "conv3d_wgrad: empty volume");
    STX_REQUIRE((ks == 3 && (stride == 1 || stride == 2)) || (ks == 1 && stride == 1), "conv3d_wgrad: ks/stride");
    STX_REQUIRE((long long)Hf * Wf * CF * 4 < (1ll << 31) && (long long)Hc * Wc * CC * 4 < (1ll << 31),
                "conv3d_wgrad: a plane of %d x %d voxels exceeds the 2 GiB buffer-descriptor range", Hf, Wf);
    WgradArgs a;
    a.f = f; a.c = c; a.slab = workspace;
    a.B = B; a.Df = Df; a.Hf = Hf; a.Wf = Wf; a.CF = CF; a.Dc = Dc; a.Hc = Hc; a.Wc = Wc; a.CC = CC;
    const int npairs = (CF / 32) * (CC / 32);
    const bool pipe = wgrad_pipe(ks, stride, npairs);
    a.ablate = stx_tune(STX_TUNE_WGRAD_ABLATE);
    int TH, TW;
    wgrad_tile(ks, stride, Wc, pipe, &TH, &TW);
    a.nHt = stx_cdiv(Hc, TH); a.nWt = stx_cdiv(Wc, TW);
    a.ntiles = B * Dc * a.nHt * a.nWt;
    int nchunks = wgrad_chunks(a.ntiles, npairs, pipe);
    hipStream_t st = (hipStream_t)stream;
    const int T = ks == 1 ? 1 : 27;
    if (ks == 3 && (stx_tune(STX_TUNE_WGRAD_MARCH) & (stride == 1 ? 1 : 2))) {
        // march form (wgrad_march.hip): K-contiguous operands, rolling plane windows, runs cut at step granularity
        // (STX_WGRAD_MARCH: bit 0 = the stride-1 layers, bit 1 = stride 2 / transposed)
        int mc = stx_wgrad_march_chunks(B, Dc, Hc, Wc, npairs);
        const int forced = stx_tune(STX_TUNE_WGRAD_GRID);
        if (forced > 0 && forced < mc) mc = forced;
        const int rc = stride == 1 ? stx_wgrad_march_launch(f, c, workspace, B, Dc, Hc, Wc, CF, CC, mc, stream, nullptr)
                                   : stx_wgrad_march_s2_launch(f, c, workspace, B, Df, Hf, Wf, CF, Dc, Hc, Wc, CC, mc, stream);
        if (rc > 0) return rc;
        if (rc == 0) {
            const int total = npairs * T * 1024;
            hipLaunchKernelGGL(conv3d_wgrad_reduce4_kernel, dim3(total / 64), dim3(CONV_THREADS), 0, st, workspace, dw, CF, CC, T,
                               mc, 27);
            return stx_check_launch("conv3d_wgrad_reduce");
        }
    }
    dim3 grid(nchunks, npairs);
    // 3x3x3: eight waves, 3-4 taps (48-64 accumulator registers) each: measured +22 % over four waves with 7
#define WG_LAUNCH(KS_, S_, TH_, TW_, NW_, PIPE_, LDS_)                                                           \
    {                                                                                                             \
        const size_t lds = (LDS_);                                                                                \
        void (*wk)(WgradArgs) = conv3d_wgrad_kernel<KS_, S_, TH_, TW_, NW_, PIPE_>;                               \
        hipFuncSetAttribute((const void*)wk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        hipLaunchKernelGGL(wk, grid, dim3(NW_ * 64), lds, st, a);                                                 \
    }
    if (ks == 3 && stride == 1) {
        if (TW == 16 && pipe) WG_LAUNCH(3, 1, 4, 16, 8, true, ((size_t)3 * 6 * 18 + 64) * 32 * 4)
        else if (TW == 16) WG_LAUNCH(3, 1, 4, 16, 8, false, ((size_t)3 * 6 * 18 + 64) * 32 * 4)
        else if (pipe) WG_LAUNCH(3, 1, 4, 32, 8, true, ((size_t)3 * 6 * 34 + 128) * 32 * 4)
        else WG_LAUNCH(3, 1, 2, 32, 8, false, ((size_t)3 * 4 * 34 + 64) * 32 * 4)
    } else if (ks == 3) {
        WG_LAUNCH(3, 2, 4, 16, 8, true, ((size_t)3 * 9 * 34 + 64) * 32 * 4)      // (wgrad_pipe: always pipelined)
    } else {
        WG_LAUNCH(1, 1, 2, 32, 4, false, ((size_t)2 * 32 + 64) * 32 * 4)
    }
#undef WG_LAUNCH
    int rc = stx_check_launch("conv3d_wgrad");
    if (rc) return rc;
    const int total = npairs * T * 1024;
    hipLaunchKernelGGL(conv3d_wgrad_reduce4_kernel, dim3(total / 64), dim3(CONV_THREADS), 0, st, workspace, dw, CF, CC, T,
                       nchunks, ks == 1 ? 4 : 27);                   // (KS = 1: one slab row per wave of the 4-wave producer)
    return stx_check_launch("conv3d_wgrad_reduce");
}

// Whether stx_conv3d_wgrad_bn serves a 3x3x3 stride-1 layer of this shape (the march kernel's conditions + its tuning switch).
extern "C" int stx_conv3d_wgrad_bn_supported(int B, int D, int H, int W, int CF, int CC) {
    stx_begin();
    return (stx_tune(STX_TUNE_WGRAD_MARCH) & 1) && B > 0 && D > 0 && H > 0 && W > 0 && CF >= 32 && CC >= 32 && CF % 32 == 0 &&
           CC % 32 == 0 && (long long)H * W * CF * 4 < (1ll << 31) && (long long)H * W * CC * 4 < (1ll << 31) &&
           (long long)B * stx_cdiv(H, 4) * stx_cdiv(W, 16) < (1ll << 31);
}

// Weight gradient of a 3x3x3 stride-1 convolution whose raw output z went through train-mode BatchNorm (+ ReLU when act = 1)
// -- reference `convbn_3d` + `nn.ReLU` (models/GwcNet/submodule.py:17-20) under autograd -- taking the gradient gy BEHIND that
// BatchNorm / activation: the kernel forms  dz = gamma invstd (gy' - sum_g / n - xhat sum_gx / n)  on the way to the matrix cores
// (gy' = gy masked by the activation, recomputed from fmaf(z, scale, shift); xhat = (z - mean) invstd; sums = the [2][CC]
// rows "sum gy'" and "sum gy' xhat" of stx_bn_bwd_reduce2) and writes it to `dz` for the data-gradient launch: the
// stx_bn_bwd_apply2 pass of such a block is not needed.  dw / workspace as in stx_conv3d_wgrad (ks = 3, stride = 1).
extern "C" int stx_conv3d_wgrad_bn(const float* x, const float* gy, const float* z, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, const float* gamma, const float* sums, float inv_n,
                                   int act, float* dz, float* dw, float* workspace, int B, int D, int H, int W, int CF, int CC,
                                   void* stream) {
    stx_begin();
    STX_REQUIRE(x && gy && z && scale && shift && mean && invstd && sums && dz && dw && workspace, "conv3d_wgrad_bn: null operand");
    STX_REQUIRE(act == 0 || act == 1, "conv3d_wgrad_bn: activation code %d (0 or 1)", act);
    STX_REQUIRE(stx_conv3d_wgrad_bn_supported(B, D, H, W, CF, CC), "conv3d_wgrad_bn: shape (%d x %d x %d x %d, %d -> %d channels) is "
                "not served by the march kernel (ask stx_conv3d_wgrad_bn_supported)", B, D, H, W, CF, CC);
    const int npairs = (CF / 32) * (CC / 32);
    int mc = stx_wgrad_march_chunks(B, D, H, W, npairs);
    const int forced = stx_tune(STX_TUNE_WGRAD_GRID);
    if (forced > 0 && forced < mc) mc = forced;
    StxWgradBn bn;
    bn.z = z; bn.dz = dz; bn.scale = scale; bn.shift = shift; bn.mean = mean; bn.invstd = invstd; bn.gamma = gamma;
    bn.sum_g = sums; bn.sum_gx = sums + CC; bn.inv_n = inv_n; bn.act = act;
    const int rc = stx_wgrad_march_launch(x, gy, workspace, B, D, H, W, CF, CC, mc, stream, &bn);
    if (rc > 0) return rc;
    STX_REQUIRE(rc == 0, "conv3d_wgrad_bn: the march kernel refused the shape");
    const int total = npairs * 27 * 1024;
    hipLaunchKernelGGL(conv3d_wgrad_reduce4_kernel, dim3(total / 64), dim3(CONV_THREADS), 0, (hipStream_t)stream, workspace, dw, CF,
                       CC, 27, mc, 27);
    return stx_check_launch("conv3d_wgrad_reduce");
}
