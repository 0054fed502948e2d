// Shared device/host helpers for the gfx950 (CDNA4) kernels of the cost-volume hot path.
//
// Every kernel in this directory is written for MI355X only: 64-lane wavefronts,
// 160 KiB LDS per CU, fp32-input MFMA (v_mfma_f32_32x32x2_f32).  The STX_HIPEMU
// branch below is NOT a second backend: it is the hook through which tests/hipemu
// compiles these same sources for the host so that index math can be checked in a
// GPU-less container (see tests/hipemu/hipemu.h).  The product loads only the
// gfx950 shared object.
#pragma once
#ifdef STX_HIPEMU
#include "hipemu.h"
#define STX_DYN_SMEM(name) char* name = hipemu::dyn_smem()
#define stx_exp(x) expf(x)
#define stx_fdiv(a, b) ((a) / (b))
#define STX_SCHED_BARRIER() ((void)0)
#define STX_SCHED_GROUP(mask, n) ((void)0)
#define STX_OPAQUE_VGPR(x) ((void)0)
#define STX_TIE3(a, b, c) ((void)0)
#define STX_BARRIER_LDS() __syncthreads()
#define STX_GLDS4(base, off, ldsptr) ((void)(((float*)(ldsptr))[threadIdx.x & 63u] = (base)[(off)]))
#define STX_GLDS_WAIT() ((void)0)
#else
#include <hip/hip_runtime.h>
#define STX_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
// exp / division of the streaming kernels (softmax of the regression head, Mish, the FeatureAtt gate): the hardware
// approximations (v_exp_f32 behind one multiply, v_rcp_f32) unless the library is built with -DSTX_PRECISE_MATH (correctly
// rounded-to-1-ulp libm forms; an A/B build for error attribution, see tests/test_models.py "hand_written_path_isolated")
#ifdef STX_PRECISE_MATH
#define stx_exp(x) expf(x)
#define stx_fdiv(a, b) ((a) / (b))
#else
#define stx_exp(x) __expf(x)
#define stx_fdiv(a, b) __fdividef((a), (b))
#endif
// Instruction-scheduling fence (guide 5.4 rule 18 / T19): nothing moves across it.
#define STX_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// Instruction interleave inside one scheduling region (guide T19): the next `n` instructions of class `mask` (0x008 MFMA,
// 0x100 LDS read, 0x200 LDS write, 0x020 VMEM read, 0x002 VALU) form the next group of the region's pipeline.
#define STX_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
// Makes a VGPR value opaque to the optimiser at this point (blocks hoisting of address math that
// would otherwise be precomputed into dozens of live registers; guide 5.7 item 3).
#define STX_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))
// Orders three VGPR values at this point of the program: everything that produces a, b, c is issued before anything that
// consumes them afterwards (keeps a prefetched value's s_waitcnt BEHIND the arithmetic it is meant to overlap with).
#define STX_TIE3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
// Workgroup barrier that publishes this wave's LDS reads / writes only (lgkmcnt): __syncthreads() also drains vmcnt whenever
// an LDS-DMA is in flight (guide 5.4: "the workgroup release inside __syncthreads carries a vmcnt(0)"), which would make a
// wave wait for its own prefetch at the next barrier.  For kernels whose barriers hand over LDS data written with ds_write.
#define STX_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// LDS-DMA, one dword per lane: LDS[ldsptr + 4 * lane] = base[off] (ldsptr and base wave-uniform, off = the lane's element
// index).  Asynchronous, counted by vmcnt.  Written as an asm statement because hipcc treats the builtin's destination as
// aliasing every other LDS access of the kernel: it then drains vmcnt in front of each ds_write and between the DMAs
// themselves, i.e. the wave waits for its own prefetch.  The CALLER orders the accesses: STX_GLDS_WAIT() before reading the
// destination back, and no ds access to it while a DMA is in flight.  M0 (destination base) is saved and restored inside
// the statement (guide 5.7: the compiler owns M0 and does not preserve it around asm).
__device__ __forceinline__ void stx_glds4(const float* base, unsigned off, const float* ldsptr) {
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(const __attribute__((address_space(3))) void*)ldsptr);
    const unsigned boff = off * 4u;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(boff), "s"(base), "s"(dst)
                 : "memory");
}
#define STX_GLDS4(base, off, ldsptr) stx_glds4((base), (off), (ldsptr))
#define STX_GLDS_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
// Buffer loads through a resource descriptor (wave-uniform base + 32-bit lane offset + wave-uniform offset): lanes whose
// offset lies outside [0, bytes) read zeros without touching memory -- the bounds check replaces address clamps and
// selects (guide T8).  STX_BUF_OOB is the offset to pass for such lanes (bytes <= 2 GiB).
#define STX_BUF_OOB 0x80000000u
#ifdef STX_HIPEMU
struct stx_bufrsrc { const char* base; unsigned bytes; };
static inline stx_bufrsrc stx_make_rsrc(const void* p, unsigned bytes) { return stx_bufrsrc{(const char*)p, bytes}; }
static inline float4 stx_buf_ld4(stx_bufrsrc r, unsigned voff, unsigned soff) {
    const unsigned long long off = (unsigned long long)voff + soff;
    if (voff >= STX_BUF_OOB || off + 16 > r.bytes) return make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4*>(r.base + off);
}
static inline float stx_buf_ld1(stx_bufrsrc r, unsigned voff, unsigned soff) {
    const unsigned long long off = (unsigned long long)voff + soff;
    if (voff >= STX_BUF_OOB || off + 4 > r.bytes) return 0.f;
    return *reinterpret_cast<const float*>(r.base + off);
}
// Store through a descriptor: lanes with voff = STX_BUF_OOB are dropped.  (The hardware checks voff only, not voff + soff:
// an in-range voff with an out-of-range sum would corrupt memory on the chip -- the emulator aborts on it.)
static inline void stx_buf_st1(stx_bufrsrc r, unsigned voff, unsigned soff, float v) {
    if (voff >= STX_BUF_OOB || voff >= r.bytes) return;            // (what the hardware's range check drops)
    const unsigned long long off = (unsigned long long)voff + soff;
    if (off + 4 > r.bytes) abort();
    *reinterpret_cast<float*>(const_cast<char*>(r.base) + off) = v;
}
static inline void stx_buf_st4(stx_bufrsrc r, unsigned voff, unsigned soff, float4 v) {
    if (voff >= STX_BUF_OOB || voff >= r.bytes) return;            // (what the hardware's range check drops)
    const unsigned long long off = (unsigned long long)voff + soff;
    if (off + 16 > r.bytes) abort();
    *reinterpret_cast<float4*>(const_cast<char*>(r.base) + off) = v;
}
#else
typedef __amdgpu_buffer_rsrc_t stx_bufrsrc;
typedef unsigned int stx_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ stx_bufrsrc stx_make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 stx_buf_ld4(stx_bufrsrc r, unsigned voff, unsigned soff) {
    const stx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
__device__ __forceinline__ float stx_buf_ld1(stx_bufrsrc r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void stx_buf_st1(stx_bufrsrc r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void stx_buf_st4(stx_bufrsrc r, unsigned voff, unsigned soff, float4 v) {
    const stx_u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)voff, (int)soff, 0);
}
#endif
// A pointer the program knows to be wave-uniform, forced into SGPRs (two v_readfirstlane): loads through it become
// `global_load v, v_offset, s[base:base+1]` and it cannot be spilled to scratch as a VGPR pair.
template <typename T>
__device__ __forceinline__ T* stx_uniform_ptr(T* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define STX_OK 0
#define STX_ERR_ARG 1
#define STX_ERR_LAUNCH 2

// Records the message returned by stx_last_error() (thread local) and returns `code`.
int stx_set_error(int code, const char* fmt, ...);
int stx_check_launch(const char* what);
// Drops any stale (sticky) HIP error left by other users of the runtime in this thread, so that
// stx_check_launch reports only errors of the launch it follows.
static inline void stx_begin() { (void)hipGetLastError(); }

#define STX_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return stx_set_error(STX_ERR_ARG, __VA_ARGS__); \
    } while (0)

static inline int stx_cdiv(int a, int b) { return (a + b - 1) / b; }

// Tuning / A-B switches of the library.  Each has a default (the measured-best path), is read from its environment
// variable ONCE, when the library is loaded, and can be changed at run time through the C-ABI (stx_set_tuning: tests and
// tools/kernel_bench.py flip them inside one process).  None changes results beyond fp32 rounding.
enum StxTune {
    STX_TUNE_MARCH_BS,       // STX_MARCH_BS       1  march kernel: one accumulator per (output, input plane), summed in the epilogue
    STX_TUNE_MARCH_EPI,      // STX_MARCH_EPI      1  march kernel: straight-line epilogue for launches without partial sums / residual / Mish
    STX_TUNE_MARCH_ABLATE,   // STX_MARCH_ABLATE   0  profiling: 1 = no plane staging, 2 = no epilogue stores
    STX_TUNE_WGRAD_ABLATE,   // STX_WGRAD_ABLATE   0  profiling: 1 = no tile staging, 2 = no MFMA loop
    STX_TUNE_WGRAD_MARCH,    // STX_WGRAD_MARCH    3  weight gradient 3x3x3 on the march kernels (wgrad_march.hip): bit 0 = stride 1, bit 1 = stride 2 / transposed; 0 = the tile kernel of rounds 1-3
    STX_TUNE_WGRAD_GRID,     // STX_WGRAD_GRID     0  weight gradient: split-K workgroups per channel-block pair (tests: many tiles per workgroup)
    STX_TUNE_CONV_L1_MARCH,  // STX_CONV_L1_MARCH  0  3x3x3 stride-1 64 -> 64: march kernel in 2 x 2 channel slices instead of the implicit-GEMM kernel
    STX_TUNE_CONV_S2_DENSE,  // STX_CONV_S2_DENSE  1  stride-2 32->64 conv: un-padded LDS tile (three workgroups per CU)
    STX_TUNE_CONV_WN,        // STX_CONV_WN        2  implicit-GEMM 3x3x3 kernels, wave grid: 1 = one row x all column blocks per wave (rounds 1-4); 2 = 64 output channels as two rows x one block; 3 / 4 = also the 128-channel pipelined kernel as 2 x 2 / 4 x 1
    STX_TUNE_CV_OLD,         // STX_CV_OLD         0  cost volume forward: first-generation builders (fallback path) for every shape
    STX_TUNE_CV_GRID,        // STX_CV_GRID        0  cost volume forward: workgroups (tests: multi-unit runs)
    STX_TUNE_CV_PF,          // STX_CV_PF          0  cost volume forward: feature prefetch, 0 = default (1: one tile ahead), 2 = cache-line pairs through an LDS-DMA slot
    STX_TUNE_CV_UNITS,       // STX_CV_UNITS       1  cost volume forward: workgroup runs cut at units (0: at whole macro-units)
    STX_TUNE_CV_WIN,         // STX_CV_WIN         0  cost volume forward: macro-units per window (every window is split over ALL workgroups, the windows walked in order: the chip writes one neighbourhood of each d-plane at a time); 0 = one window
    STX_TUNE_CVB_OLD,        // STX_CVB_OLD        0  cost volume backward: first-generation kernels for every shape
    STX_TUNE_CVB_TEAM,       // STX_CVB_TEAM       0  cost volume backward: row-team schedule (one HBM pass, lock-step)
    STX_TUNE_CVB_GRID,       // STX_CVB_GRID       0  cost volume backward: workgroups (tests)
    STX_TUNE_CVB_NSET,       // STX_CVB_NSET       3  cost volume backward: chunks in flight per loader lane (2..4)
    STX_TUNE_SV_BWD_V1,      // STX_SV_BWD_V1      0  CFNet cascade-volume backward: global atomics only
    STX_TUNE_DWCONV_ROLL,    // STX_DWCONV_ROLL    1  ACVNet patch convolutions: rolling window of input rows in LDS (0 = the cache-fed kernel of rounds 3-4)
    STX_TUNE_COUNT
};
int stx_tune(StxTune id);

// Activation code of the conv / BN-apply epilogues (the C-ABI's `relu` argument): 0 none, 1 ReLU, 2 Mish,
// 3 LeakyReLU(0.01) (IGEV-family `BasicConv`, models/IGEVStereo/submodule.py:32-37: `nn.LeakyReLU()`, torch's default slope).
// Mish (reference models/PCWNet/submodule.py:11-18): x * tanh(softplus(x)) with tanh(log(1 + e^x)) = n / (n + 2),
// n = e^x (e^x + 2) -- one exp and one division, no cancellation (n >= 0); above torch's softplus threshold of 20 the
// factor is 1.  `w` returns e^min(x, 20) for the derivative.
__device__ __forceinline__ float stx_mish_tanh_sp(float x, float& w) {
    w = stx_exp(x < 20.f ? x : 20.f);
    const float n = w * (w + 2.f);
    return x > 20.f ? 1.f : stx_fdiv(n, n + 2.f);
}
__device__ __forceinline__ float stx_mish(float x) {
    float w;
    return x * stx_mish_tanh_sp(x, w);
}
// d/dx [x * tanh(softplus(x))] = t + x * (1 - t^2) * sigmoid(x)   (sigmoid -> 1 above the threshold)
__device__ __forceinline__ float stx_mish_grad(float x) {
    float w;
    const float t = stx_mish_tanh_sp(x, w);
    const float ds = x > 20.f ? 1.f : stx_fdiv(w, 1.f + w);
    return t + x * (1.f - t * t) * ds;
}
#define STX_LEAKY_SLOPE 0.01f
__device__ __forceinline__ float stx_act(float v, int code) {
    return code == 1 ? (v > 0.f ? v : 0.f) : (code == 2 ? stx_mish(v) : (code == 3 ? (v > 0.f ? v : STX_LEAKY_SLOPE * v) : v));
}
// gradient through the activation: g * act'(.) with `t` the value the sign is taken from -- for ReLU / LeakyReLU the
// activated OR the pre-activation value (same sign; torch differentiates both as `x > 0 ? g : slope * g`), for Mish the
// pre-activation value
__device__ __forceinline__ float stx_act_bwd(float g, float t, int code) {
    return code == 2 ? g * stx_mish_grad(t) : (code == 3 ? (t > 0.f ? g : STX_LEAKY_SLOPE * g) : (t > 0.f ? g : 0.f));
}

__device__ __forceinline__ float4 stx_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void stx_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// 16-byte store with the non-temporal hint (streamed output that the writing CU never re-reads)
__device__ __forceinline__ void stx_st4_nt(float* p, float4 v) {
#ifdef STX_HIPEMU
    *reinterpret_cast<float4*>(p) = v;
#else
    f32x4 t;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
#endif
}
