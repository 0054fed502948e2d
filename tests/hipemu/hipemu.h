// hipemu.h -- TEST INFRASTRUCTURE ONLY (never part of the product path).
//
// A tiny host-side SIMT emulator that lets the *same* .hip kernel sources that
// ship in stereo_toolbox_amd/csrc be compiled for x86 (amdclang++ -x c++
// -DSTX_HIPEMU) and executed thread-by-thread on the CPU, so that index math,
// LDS tiling, barriers and the MFMA fragment layouts can be validated against
// the oracle in this GPU-less container before GPU minutes are spent.
//
//  * every GPU thread is a fiber; a workgroup runs on one host thread, up to STX_EMU_THREADS (default: the host's
//    cores, at most 8) workgroups of a launch run concurrently on their own host threads (workgroups of these kernels
//    never wait for each other; global atomics are real atomics);
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
#include <atomic>
#include <thread>
#include <vector>
#include <sys/mman.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local          // (one workgroup per host thread at a time)

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
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 1; return 0; }
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
    char* stacks = nullptr;            // mmap'ed lazily-committed fiber stacks (grow-only)
    size_t stacks_bytes = 0;
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
    ~State() { if (stacks) munmap(stacks, stacks_bytes); }
};
inline State& S() { static thread_local State s; return s; }

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
inline thread_local uint3_emu threadIdx, blockIdx;
inline thread_local dim3 blockDim, gridDim;

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

static inline float atomicAdd(float* p, float v) {
    unsigned* q = reinterpret_cast<unsigned*>(p);
    unsigned o = __atomic_load_n(q, __ATOMIC_RELAXED), n;
    float of;
    do {
        memcpy(&of, &o, 4);
        const float nf = of + v;
        memcpy(&n, &nf, 4);
    } while (!__atomic_compare_exchange_n(q, &o, n, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return of;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
// round-to-nearest single operations that must not be contracted into FMAs (the host build compiles with -ffp-contract=off
// semantics for these through volatile temporaries)
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // callers pass wave-uniform values

namespace hipemu {

template <typename F>
struct Thunk {
    static F* f;
    static void run() { (*f)(); }
};
template <typename F> F* Thunk<F>::f = nullptr;

inline unsigned host_threads() {
    static const unsigned n = [] {
        const char* e = getenv("STX_EMU_THREADS");
        unsigned v = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
        if (!e && v > 8) v = 8;
        return v < 1 ? 1u : v;
    }();
    return n;
}

// one workgroup on the calling host thread
inline void run_block(dim3 grid, dim3 block, size_t shmem, void (*entry)(), unsigned bx, unsigned by, unsigned bz) {
    State& s = S();
    const unsigned nt = block.x * block.y * block.z;
    const size_t STK = 256 * 1024;
    s.nthreads = nt;
    if (s.fibers.size() < nt) s.fibers.resize(nt);
    if (s.stacks_bytes < nt * STK) {
        if (s.stacks) munmap(s.stacks, s.stacks_bytes);
        s.stacks_bytes = nt * STK;
        s.stacks = (char*)mmap(nullptr, s.stacks_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (s.stacks == (char*)MAP_FAILED) { fprintf(stderr, "hipemu: mmap of fiber stacks failed\n"); abort(); }
    }
    s.dyn.assign(shmem + 64, 0);
    s.entry = entry;
    blockDim = block;
    gridDim = grid;
    s.bar_count = 0; s.live = nt;
    memset(s.wave_count, 0, sizeof(s.wave_count));
    for (unsigned t = 0; t < nt; ++t) {
        Fiber& f = s.fibers[t];
        f.done = false; f.tid = t;
        uintptr_t top = (uintptr_t)(s.stacks + (size_t)(t + 1) * STK);
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

template <typename F>
inline void run_grid(dim3 grid, dim3 block, size_t shmem, F body) {
    const unsigned nt = block.x * block.y * block.z;
    if (nt > 1024) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
    if (shmem > 160 * 1024) { fprintf(stderr, "hipemu: %zu B of dynamic LDS exceeds the 160 KiB of a gfx950 CU\n", shmem); abort(); }
    Thunk<F>::f = &body;                                  // (launches are serialised by the caller: one at a time)
    void (*entry)() = &Thunk<F>::run;
    const unsigned long long nblk = (unsigned long long)grid.x * grid.y * grid.z;
    auto block_at = [&](unsigned long long b) {
        const unsigned bx = (unsigned)(b % grid.x), by = (unsigned)((b / grid.x) % grid.y), bz = (unsigned)(b / ((unsigned long long)grid.x * grid.y));
        run_block(grid, block, shmem, entry, bx, by, bz);
    };
    unsigned nw = host_threads();
    if (nw > nblk) nw = (unsigned)nblk;
    if (nw <= 1) {
        for (unsigned long long b = 0; b < nblk; ++b) block_at(b);
        return;
    }
    std::atomic<unsigned long long> next{0};
    auto worker = [&]() {
        for (;;) {
            const unsigned long long b = next.fetch_add(1, std::memory_order_relaxed);
            if (b >= nblk) break;
            block_at(b);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(nw - 1);
    for (unsigned i = 1; i < nw; ++i) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
}

}  // namespace hipemu

template <typename K, typename... Args>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
    hipemu::run_grid(grid, block, shmem, [=]() { kernel(args...); });
}
