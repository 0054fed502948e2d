// hipemu.h -- TEST INFRASTRUCTURE ONLY (never part of the product path).
//
// A tiny host-side SIMT emulator that lets the *same* .hip kernel sources that
// ship in stereo_toolbox_amd/csrc be compiled for x86 (amdclang++ -x c++
// -DSTX_HIPEMU) and executed thread-by-thread on the CPU, so that index math,
// LDS tiling, barriers and the MFMA fragment layouts can be validated against
// the oracle in this GPU-less container before GPU minutes are spent.
//
//  * every GPU thread is a ucontext fiber; one workgroup runs at a time;
//  * __syncthreads() / wave collectives (shuffles, MFMA) are rendezvous points;
//  * MFMA lane layouts follow /opt/skills/guides/cdna_hip_programming.md §3
//    (32x32x2f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
//     D col=l&31,row=(r&3)+8*(r>>2)+4*(l>>5);
//     16x16x4f32: A[l&15][k=l>>4], B[k=l>>4][l&15], D col=l&15,row=4*(l>>4)+r)
//    and the k-ordered fmaf chain numerics stated there.
//
// The product never loads the emulator build: stereo_toolbox_amd/_capi.py only
// opens the gfx950 shared object. Only tests/ open libstx_emu.so.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 8

namespace hipemu {

// Minimal x86-64 SysV context switch (tests/hipemu/hipemu_impl.cpp): saves the callee-saved
// registers on the current stack, stores its stack pointer to *save_sp and resumes load_sp.
extern "C" void hipemu_switch(void** save_sp, void* load_sp);

struct Fiber {
    void* sp = nullptr;
    bool done = false;
    unsigned tid = 0;
};

struct State {
    void* sched_sp = nullptr;
    std::vector<Fiber> fibers;
    std::vector<char> stacks;
    unsigned nthreads = 0;
    unsigned cur = 0;
    // block barrier
    unsigned bar_count = 0, bar_gen = 0, live = 0;
    // per-wave rendezvous
    unsigned wave_count[64] = {0}, wave_gen[64] = {0};
    // wave exchange slots: up to 4 dwords per lane per collective
    float slot[64][64][4];
    std::vector<char> dyn;
    void (*entry)() = nullptr;
};
inline State& S() { static State s; return s; }

extern "C" inline void hipemu_trampoline() {
    State& s = S();
    s.entry();
    s.fibers[s.cur].done = true;
    s.live--;
    hipemu_switch(&s.fibers[s.cur].sp, s.sched_sp);
    abort();   // a finished fiber is never resumed
}

inline void yield_() {
    State& s = S();
    hipemu_switch(&s.fibers[s.cur].sp, s.sched_sp);
}

inline char* dyn_smem() { return S().dyn.data(); }

}  // namespace hipemu

// The scheduler refreshes these on every fiber switch.
inline uint3_emu threadIdx, blockIdx;
inline dim3 blockDim, gridDim;

static inline void __syncthreads() {
    hipemu::State& s = hipemu::S();
    unsigned gen = s.bar_gen;
    if (++s.bar_count >= s.live) { s.bar_count = 0; s.bar_gen++; return; }
    while (s.bar_gen == gen) hipemu::yield_();
}

namespace hipemu {
inline unsigned lane_() { return threadIdx.x & 63u; }
inline unsigned wave_() { return threadIdx.x >> 6; }
inline void wave_sync_() {
    State& s = S();
    unsigned w = wave_();
    unsigned nl = s.nthreads - w * 64 < 64 ? s.nthreads - w * 64 : 64;
    unsigned gen = s.wave_gen[w];
    if (++s.wave_count[w] >= nl) { s.wave_count[w] = 0; s.wave_gen[w]++; return; }
    while (s.wave_gen[w] == gen) yield_();
}
}  // namespace hipemu

static inline float __shfl_xor(float v, int mask, int width = 64) {
    using namespace hipemu;
    (void)width;
    State& s = S();
    unsigned w = wave_(), l = lane_();
    s.slot[w][l][0] = v;
    wave_sync_();
    float r = s.slot[w][(l ^ (unsigned)mask) & 63][0];
    wave_sync_();
    return r;
}
static inline float __shfl_down(float v, unsigned delta, int width = 64) {
    using namespace hipemu;
    (void)width;
    State& s = S();
    unsigned w = wave_(), l = lane_();
    s.slot[w][l][0] = v;
    wave_sync_();
    unsigned src = l + delta;
    float r = src < 64 ? s.slot[w][src][0] : v;
    wave_sync_();
    return r;
}
static inline float __shfl(float v, int src, int width = 64) {
    using namespace hipemu;
    (void)width;
    State& s = S();
    unsigned w = wave_(), l = lane_();
    s.slot[w][l][0] = v;
    wave_sync_();
    float r = s.slot[w][src & 63][0];
    wave_sync_();
    return r;
}

typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));

// D = A(32x2) * B(2x32) + C, k-ordered fmaf chain (guide §3 "Numerics").
static inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
    using namespace hipemu;
    State& s = S();
    unsigned w = wave_(), l = lane_();
    s.slot[w][l][0] = a;
    s.slot[w][l][1] = b;
    wave_sync_();
    unsigned col = l & 31;
    for (int r = 0; r < 16; ++r) {
        unsigned row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (unsigned k = 0; k < 2; ++k) acc = fmaf(s.slot[w][row + 32 * k][0], s.slot[w][col + 32 * k][1], acc);
        c[r] = acc;
    }
    wave_sync_();
    return c;
}
static inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    using namespace hipemu;
    State& s = S();
    unsigned w = wave_(), l = lane_();
    s.slot[w][l][0] = a;
    s.slot[w][l][1] = b;
    wave_sync_();
    unsigned col = l & 15;
    for (int r = 0; r < 4; ++r) {
        unsigned row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (unsigned k = 0; k < 4; ++k) acc = fmaf(s.slot[w][row + 16 * k][0], s.slot[w][col + 16 * k][1], acc);
        c[r] = acc;
    }
    wave_sync_();
    return c;
}

// 16 blocks of D(4x4) += A(4x1) * B(1x4): lane = 4 * block + (row of A | column of B), D[i][j] in register i of lane
// 4 * block + j.
static inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_4x4x1f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    using namespace hipemu;
    State& s = S();
    unsigned w = wave_(), l = lane_();
    s.slot[w][l][0] = a;
    wave_sync_();
    const unsigned blk = l >> 2;
    for (int i = 0; i < 4; ++i) c[i] = fmaf(s.slot[w][4 * blk + i][0], b, c[i]);
    wave_sync_();
    return c;
}

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // callers pass wave-uniform values

namespace hipemu {

template <typename F>
struct Thunk {
    static F* f;
    static void run() { (*f)(); }
};
template <typename F> F* Thunk<F>::f = nullptr;

template <typename F>
inline void run_grid(dim3 grid, dim3 block, size_t shmem, F body) {
    State& s = S();
    const unsigned nt = block.x * block.y * block.z;
    const size_t STK = 256 * 1024;
    if (nt > 1024) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
    if (shmem > 160 * 1024) { fprintf(stderr, "hipemu: %zu B of dynamic LDS exceeds the 160 KiB of a gfx950 CU\n", shmem); abort(); }
    s.nthreads = nt;
    if (s.fibers.size() < nt) s.fibers.resize(nt);
    if (s.stacks.size() < nt * STK) s.stacks.resize(nt * STK);
    s.dyn.assign(shmem + 64, 0);
    Thunk<F>::f = &body;
    s.entry = &Thunk<F>::run;
    blockDim = block;
    gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        s.bar_count = 0; s.live = nt;
        memset(s.wave_count, 0, sizeof(s.wave_count));
        for (unsigned t = 0; t < nt; ++t) {
            Fiber& f = s.fibers[t];
            f.done = false; f.tid = t;
            uintptr_t top = (uintptr_t)(s.stacks.data() + (size_t)(t + 1) * STK);
            top &= ~(uintptr_t)15;
            void** sp = (void**)top;
            *--sp = nullptr;                              // fake return address of the trampoline
            *--sp = (void*)hipemu_trampoline;             // popped by hipemu_switch's `ret`
            for (int r = 0; r < 6; ++r) *--sp = nullptr;  // rbp rbx r12 r13 r14 r15
            f.sp = (void*)sp;
        }
        unsigned remaining = nt;
        while (remaining) {
            remaining = 0;
            for (unsigned t = 0; t < nt; ++t) {
                Fiber& f = s.fibers[t];
                if (f.done) continue;
                s.cur = t;
                threadIdx.x = t % block.x;
                threadIdx.y = (t / block.x) % block.y;
                threadIdx.z = t / (block.x * block.y);
                blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
                hipemu_switch(&s.sched_sp, f.sp);
                if (!f.done) remaining++;
            }
        }
    }
}

}  // namespace hipemu

template <typename K, typename... Args>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
    hipemu::run_grid(grid, block, shmem, [=]() { kernel(args...); });
}
