// Micro-benchmark of the wgrad inner loop shape: per stage 1 B read + NT A reads (ds_read_b32, one float per
// lane) feeding NT MFMAs on NT accumulator chains, prefetched DEPTH stages ahead; optional per-MFMA uniform
// branch (tap ownership test).  8 waves per workgroup, 1 workgroup per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// MODE bit0: LDS reads on; bit1: per-MFMA uniform branch; bit2: vector (exec-mask) instead of scalar branch;
// bit3: two stages per trip (DEPTH 2)
template <int NT, int MODE>
__global__ __launch_bounds__(512) void k(float* out, const float* __restrict__ w, int iters, int own) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 20480; i += blockDim.x) lds[i] = w[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = (MODE & 4) ? (threadIdx.x >> 6) : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[NT];
    for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float av[2][NT], bv[2];
    int toff[NT];
    for (int t = 0; t < NT; ++t) { toff[t] = (t * 8 + wave) * 131 % 97 * 32; asm volatile("" : "+v"(toff[t])); }
    auto load = [&](int p, int buf) {
        const int v = 2 * p + (lane >> 5);
        const int vb = ((v >> 5) * 34 + (v & 31)) * 32 + (lane & 31);
        if (MODE & 1) {
            bv[buf] = lds[16384 + (v & 63) * 32 + (lane & 31)];
            for (int t = 0; t < NT; ++t) av[buf][t] = lds[vb + toff[t]];
        }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (MODE & 2) { if (t * 8 + wave < own) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][t], bv[buf], acc[t], 0, 0, 0); }
            else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][t], bv[buf], acc[t], 0, 0, 0);
        }
    };
    for (int t = 0; t < NT; ++t) { av[0][t] = av[1][t] = w[lane + t]; }
    bv[0] = bv[1] = w[lane + 77];
    for (int it = 0; it < iters; ++it) {
        load(0, 0);
        for (int p = 0; p < 32; p += 2) {
            load(p + 1, 1);
            FENCE();
            mma(0);
            FENCE();
            load((p + 2) & 31, 0);
            FENCE();
            mma(1);
            FENCE();
        }
    }
    float s = 0.f;
    for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NT, int MODE>
void run(const float* w, float* out, int own) {
    const int iters = 60;
    dim3 grid(256), blk(512);
    const size_t ldsb = 20480 * 4;
    hipFuncSetAttribute((const void*)k<NT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NT, MODE>), grid, blk, ldsb, 0, out, w, 2, own);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NT, MODE>), grid, blk, ldsb, 0, out, w, iters, own);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double mf = 0;
    for (int wv = 0; wv < 8; ++wv) for (int t = 0; t < NT; ++t) if (!(MODE & 2) || t * 8 + wv < own) mf += 1;
    const double flops = 256.0 * mf * iters * 32 * 4096.0;
    printf("NT=%d mode=%d (lds=%d branch=%d vec=%d) own=%d : %.1f TFLOP/s (%.3f ms)\n", NT, MODE, MODE & 1, (MODE >> 1) & 1,
           (MODE >> 2) & 1, own, flops / ms / 1e9, ms);
}

int main() {
    float* h = (float*)malloc(65536 * 4);
    srand(1);
    for (int i = 0; i < 65536; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *w, *out;
    hipMalloc(&w, 65536 * 4); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(w, h, 65536 * 4, hipMemcpyHostToDevice);
    run<4, 0>(w, out, 32); run<4, 1>(w, out, 32); run<4, 3>(w, out, 32); run<4, 7>(w, out, 32);
    run<4, 3>(w, out, 27); run<4, 7>(w, out, 27);
    run<3, 0>(w, out, 32); run<3, 1>(w, out, 32);
    run<2, 0>(w, out, 32); run<2, 1>(w, out, 32);
    run<7, 1>(w, out, 32); run<8, 1>(w, out, 32);
    return 0;
}
