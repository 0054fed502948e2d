// Micro-benchmark (round 6): sustained rate of v_mfma_f32_16x16x4_f32 against v_mfma_f32_32x32x2_f32 on MI355X, one wave per
// SIMD, 2 / 4 independent accumulator chains, short (conv2d-sized, ~50 us) and long launches.
//   hipcc --offload-arch=gfx950 -O3 mfma16_peak.hip -o mfma16_peak && timeout 60 ./mfma16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH, bool BIG>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ rnd, int iters) {
    f32x4 acc4[CH];
    f32x16 acc16[CH];
    for (int c = 0; c < CH; ++c) {
        for (int i = 0; i < 4; ++i) acc4[c][i] = 0.f;
        for (int i = 0; i < 16; ++i) acc16[c][i] = 0.f;
    }
    float a[8], b[8];
    for (int u = 0; u < 8; ++u) {
        a[u] = rnd[(threadIdx.x * 8 + u) & 4095];
        b[u] = rnd[(threadIdx.x * 8 + u + 2048) & 4095];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (BIG) acc16[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc16[c], 0, 0, 0);
                else acc4[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc4[c], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) {
        for (int i = 0; i < 4; ++i) s += acc4[c][i];
        for (int i = 0; i < 16; ++i) s += acc16[c][i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH, bool BIG>
void run(const float* rnd, int iters) {
    float* out;
    hipMalloc(&out, 256 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<CH, BIG>), dim3(256), dim3(256), 0, 0, out, rnd, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<CH, BIG>), dim3(256), dim3(256), 0, 0, out, rnd, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flops = 256.0 * 4 * iters * 8.0 * CH * (BIG ? 4096.0 : 2048.0);
    printf("%s chains=%d iters=%6d : %7.1f us per launch, %6.1f TFLOP/s = %.3f of 157.3\n", BIG ? "32x32x2 " : "16x16x4 ", CH, iters,
           ms * 1e3, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
    hipFree(out);
}

int main() {
    float h[4096], *dr;
    hipMalloc(&dr, sizeof(h));
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(dr, h, sizeof(h), hipMemcpyHostToDevice);
    for (int iters : {150, 1500, 15000}) {      // 16x16x4 x 2 chains x 150 iterations = 2400 MFMAs = the conv2d kernel's per-wave count
        run<2, false>(dr, iters);
        run<4, false>(dr, iters);
        run<1, true>(dr, iters);
        run<2, true>(dr, iters / 2);
    }
    return 0;
}
