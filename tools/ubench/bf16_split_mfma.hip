// Sized experiment, NOT product code: what a 3-way bf16 split of both operands would buy the 32 -> 32 L0
// convolutions, whose fp32-input MFMA loop sits at 0.855 of a 157 TFLOP/s pipe while the bf16 pipe is 16 x wider.
//   a = a1 + a2 + a3 (bf16 pieces, round-to-nearest-even of the running remainder; 3 x 8 significant bits cover fp32's 24),
//   a * b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)        6 of the 9 cross terms; the dropped ones are <= 2^-24 |a b|
// Each term is one v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  Two questions:
//   (1) NUMERICS -- a 256 x 864 by 864 x 32 product (864 = 27 taps x 32 channels: one output of those layers) with conv-like
//       operands, against fp64: the exact-fp32 MFMA chain (v_mfma_f32_32x32x2_f32, what the product runs) vs the 6-term split
//       (small terms first / large terms first) vs a 3-term split (bf16x2-class accuracy) for scale.
//   (2) RATE -- the inner loop of a weights-and-plane-in-LDS kernel in that form: per K = 16 step 3 x ds_read_b128 of weight pieces
//       shared by R = 2 row blocks, 3 per row block of plane pieces, 6 MFMAs per row block; one wave per SIMD (the march kernel's
//       occupancy) and two.  Reported as fp32-EQUIVALENT TFLOP/s (2 M N K of the product computed) next to the fp32 pipe's 157.3.
//   hipcc --offload-arch=gfx950 -O3 bf16_split_mfma.hip -o bf16_split_mfma && ./bf16_split_mfma
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __bf16 to_bf16(float x) { return (__bf16)x; }          // RNE conversion (v_cvt_pk_bf16_f32 on gfx950)

// C[256][32] = A[256][K] * B[K][32]; one wave per 32 rows.  mode 0: fp32 MFMA chain; 1: 6-term split, small terms first;
// 2: 6-term split, large terms first; 3: 3-term split (a1 b1 + a1 b2 + a2 b1)
__global__ __launch_bounds__(64) void numerics(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K,
                                               int mode) {
    const int lane = threadIdx.x, r0 = blockIdx.x * 32;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (mode == 0) {
        // 32x32x2: lane holds A[row = lane % 32][k = lane / 32], B[k = lane / 32][col = lane % 32]
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)(r0 + (lane & 31)) * K + k + (lane >> 5)],
                                                        B[(size_t)(k + (lane >> 5)) * 32 + (lane & 31)], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            // 32x32x16: lane holds A[row = lane % 32][k0 + 8 (lane / 32) + j], B[k0 + 8 (lane / 32) + j][col = lane % 32], j < 8
            bf16x8 a[3], b[3];
            for (int j = 0; j < 8; ++j) {
                float x = A[(size_t)(r0 + (lane & 31)) * K + k + 8 * (lane >> 5) + j];
                float y = B[(size_t)(k + 8 * (lane >> 5) + j) * 32 + (lane & 31)];
                for (int p = 0; p < 3; ++p) {
                    const __bf16 xp = to_bf16(x), yp = to_bf16(y);
                    a[p][j] = xp; b[p][j] = yp;
                    x -= (float)xp; y -= (float)yp;                  // exact: the remainder fits fp32
                }
            }
            auto mm = [&](int i, int j) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc, 0, 0, 0); };
            if (mode == 1) { mm(2, 0); mm(1, 1); mm(0, 2); mm(1, 0); mm(0, 1); mm(0, 0); }
            else if (mode == 2) { mm(0, 0); mm(0, 1); mm(1, 0); mm(0, 2); mm(1, 1); mm(2, 0); }
            else { mm(1, 0); mm(0, 1); mm(0, 0); }
        }
    }
    // C/D layout of the 32x32 accumulators: col = lane % 32, row = 8 (i / 4) + 4 (lane / 32) + i % 4
    for (int i = 0; i < 16; ++i) C[(size_t)(r0 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3)) * 32 + (lane & 31)] = acc[i];
}

// Rate: everything from LDS, nothing from memory inside the loop.  R row blocks share the weight pieces of a K step.
template <int R, int SPLIT>
__global__ __launch_bounds__(256) void rate(float* out, int steps) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += 256) lds[i] = make_uint4(0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);   // bf16 ~1.0
    __syncthreads();
    f32x16 acc[R];
    for (int m = 0; m < R; ++m)
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
    const uint4* wq = lds + lane;                       // weight pieces: [step % 16][piece][lane]
    const uint4* aq = lds + 4096 + wave * 64 + lane;    // plane pieces:  [step % 4][m][piece][wave][lane]
    for (int s = 0; s < steps; ++s) {
        uint4 bw[3], av[R][3];
#pragma unroll
        for (int p = 0; p < (SPLIT ? 3 : 1); ++p) bw[p] = wq[((s & 15) * 3 + p) * 64];
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int p = 0; p < (SPLIT ? 3 : 1); ++p) av[m][p] = aq[(((s & 3) * R + m) * 3 + p) * 256];
#pragma unroll
        for (int m = 0; m < R; ++m) {
            auto mm = [&](int i, int j) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8*)&av[m][i], *(bf16x8*)&bw[j], acc[m], 0, 0, 0);
            };
            if (SPLIT) { mm(2, 0); mm(1, 1); mm(0, 2); mm(1, 0); mm(0, 1); mm(0, 0); }
            else mm(0, 0);
        }
    }
    float t = 0.f;
    for (int m = 0; m < R; ++m)
        for (int i = 0; i < 16; ++i) t += acc[m][i];
    out[(size_t)blockIdx.x * 256 + tid] = t;
}

template <int R, int SPLIT>
void run_rate(int per_cu, float* out) {
    const int steps = 20000;
    const size_t lds = (size_t)(160 * 1024 / per_cu) & ~(size_t)1023;
    hipFuncSetAttribute((const void*)rate<R, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<R, SPLIT>), dim3(256 * per_cu), dim3(256), lds, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate<R, SPLIT>), dim3(256 * per_cu), dim3(256), lds, 0, out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // fp32-equivalent work: per wave and step R blocks of 32 x 32 x 16 MACs
    const double flop = (double)256 * per_cu * 4 * steps * R * 32 * 32 * 16 * 2;
    const double mfma = (double)256 * per_cu * 4 * steps * R * (SPLIT ? 6 : 1);
    printf("rate  R=%d %s  %d workgroup(s)/CU : %.3f ms  %7.1f TFLOP/s fp32-equivalent (%.2f x the 157.3 fp32 pipe), bf16 pipe at %.2f of 2517 dense\n",
           R, SPLIT ? "6-term split" : "plain bf16  ", per_cu, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3, mfma * 32768 / ms / 1e9 / 2517.0);
}

int main() {
    // ---- (1) numerics
    const int M = 256, K = 864, N = 32;
    std::vector<float> A((size_t)M * K), B((size_t)K * N);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX; };
    // activations after BatchNorm + ReLU: half zeros, the rest ~|N(0,1)|-like; weights He-uniform with fan-in 864
    for (auto& v : A) { const float u = rnd(); v = u < 0.5f ? 0.f : 2.5f * (u - 0.5f) * (0.5f + rnd()); }
    const float bound = sqrtf(6.0f / K) * 0.85f;
    for (auto& v : B) v = bound * (2.f * rnd() - 1.f);
    std::vector<double> ref((size_t)M * N, 0.0);
    double scale = 0.0;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += (double)A[(size_t)m * K + k] * (double)B[(size_t)k * N + n];
            ref[(size_t)m * N + n] = s;
            scale = fmax(scale, fabs(s));
        }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    const char* names[4] = {"fp32 MFMA 32x32x2 chain (the product)", "bf16 x 3 split, 6 terms, small first",
                            "bf16 x 3 split, 6 terms, large first", "bf16 x 2 split, 3 terms"};
    std::vector<float> C((size_t)M * N);
    printf("numerics: %d x %d by %d x %d, max |C| = %.3f\n", M, K, K, N, scale);
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(numerics, dim3(M / 32), dim3(64), 0, 0, dA, dB, dC, K, mode);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0.0, esum = 0.0;
        for (size_t i = 0; i < C.size(); ++i) { const double e = fabs((double)C[i] - ref[i]); emax = fmax(emax, e); esum += e; }
        printf("  %-42s max err %.3e (%.2e of max |C|), mean err %.3e\n", names[mode], emax, emax / scale, esum / C.size());
    }
    // ---- (2) rate
    float* out;
    hipMalloc(&out, (size_t)512 * 256 * 4);
    run_rate<2, 0>(1, out);
    run_rate<2, 1>(1, out);
    run_rate<2, 1>(2, out);
    run_rate<1, 1>(1, out);
    run_rate<1, 1>(2, out);
    run_rate<4, 1>(1, out);
    return 0;
}
