// Micro-benchmark: v_mfma_f32_32x32x2_f32 stream fed like the conv kernels feed it -- operands
// re-loaded every 8 MFMAs from LDS (ds_read_b128) and/or global memory (1 KiB/wave, L2-resident),
// one tap ahead.  Finds which feed path (if any) drags the matrix pipe below its 155 TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: registers only, 1: +LDS A reads, 2: +global B loads, 3: both
__global__ __launch_bounds__(512) void k(float* out, const float* __restrict__ w, int iters) {
    __shared__ float lds[8192 + 64];
    for (int i = threadIdx.x; i < 8192 + 64; i += blockDim.x) lds[i] = w[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    float4 a[2][2], b[2][2];
    for (int u = 0; u < 2; ++u)
        for (int q = 0; q < 2; ++q) {
            a[u][q] = *(const float4*)(lds + lane * 36 + q * 8);
            b[u][q] = *(const float4*)(w + lane * 4 + q * 256);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int cur = t & 1, nxt = cur ^ 1;
            if (MODE & 1) {
                a[nxt][0] = *(const float4*)(lds + ((lane * 36 + t * 144 + (it & 7) * 4) & 8191));
                a[nxt][1] = *(const float4*)(lds + ((lane * 36 + t * 144 + 8 + (it & 7) * 4) & 8191));
            }
            if (MODE & 2) {
                b[nxt][0] = *(const float4*)(w + (size_t)((it * 8 + t) & 63) * 1024 + lane * 4);
                b[nxt][1] = *(const float4*)(w + (size_t)((it * 8 + t) & 63) * 1024 + 256 + lane * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0].x, b[cur][0].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][1].x, b[cur][1].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0].y, b[cur][0].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][1].y, b[cur][1].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0].z, b[cur][0].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][1].z, b[cur][1].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0].w, b[cur][0].w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][1].w, b[cur][1].w, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(int threads, int blocks_per_cu, const float* w, float* out) {
    const int iters = 4000;
    dim3 grid(256 * blocks_per_cu), blk(threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, grid, blk, 0, 0, out, w, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, grid, blk, 0, 0, out, w, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid.x * threads / 64;
    const double flops = waves * iters * 64.0 * 2.0 * 32 * 32 * 2;
    printf("mode=%d (lds=%d glb=%d) threads=%d blocks/CU=%d : %.1f TFLOP/s (%.2f ms)\n", MODE, MODE & 1, (MODE >> 1) & 1,
           threads, blocks_per_cu, flops / ms / 1e9, ms);
}

int main() {
    float* h = (float*)malloc(65536 * 4 * 4);
    srand(1);
    for (int i = 0; i < 65536 * 4; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *w, *out;
    hipMalloc(&w, 65536 * 4 * 4); hipMalloc(&out, 256 * 4 * 512 * 4);
    hipMemcpy(w, h, 65536 * 4 * 4, hipMemcpyHostToDevice);
    run<0>(512, 1, w, out); run<1>(512, 1, w, out); run<2>(512, 1, w, out); run<3>(512, 1, w, out);
    run<3>(256, 1, w, out); run<3>(256, 2, w, out);
    return 0;
}
