// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate on MI355X (a) with constant operands and
// (b) with random per-lane operands (data-dependent power / DVFS, guide MI355X_MICROARCH "DVFS give-back").
// hipcc --offload-arch=gfx950 -O3 mfma_f32_peak.hip -o mfma_f32_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ rnd, int iters) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a[8], b[8];
    for (int u = 0; u < 8; ++u) {
        a[u] = rnd[(threadIdx.x * 8 + u) & 4095];
        b[u] = rnd[(threadIdx.x * 8 + u + 2048) & 4095];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH>
void run(int blocks_per_cu, const float* rnd, const char* tag, int iters = 20000) {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    dim3 grid(256 * blocks_per_cu), blk(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CH>, grid, blk, 0, 0, out, rnd, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, grid, blk, 0, 0, out, rnd, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid.x * 4;
    const double flops = waves * iters * 8.0 * CH * 2.0 * 32 * 32 * 2;
    printf("%s chains=%d waves/SIMD=%d : %.1f TFLOP/s (%.2f ms)\n", tag, CH, blocks_per_cu, flops / ms / 1e9, ms);
    hipFree(out);
}

int main() {
    float h[4096], *dc, *dr;
    hipMalloc(&dc, sizeof(h)); hipMalloc(&dr, sizeof(h));
    for (int i = 0; i < 4096; ++i) h[i] = 1.0f;
    hipMemcpy(dc, h, sizeof(h), hipMemcpyHostToDevice);
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(dr, h, sizeof(h), hipMemcpyHostToDevice);
    run<1>(4, dr, "random short", 3000);
    run<2>(2, dr, "random short", 3000);
    run<1>(4, dr, "random long ", 40000);
    run<2>(2, dr, "random long ", 40000);
    run<1>(1, dr, "random long ", 160000);
    run<1>(2, dr, "random mid  ", 20000);
    run<1>(4, dc, "const  long ", 40000);
    return 0;
}
