// Micro-test: is a VALU read of an fp32-MFMA accumulator right behind a chain of DEPENDENT v_mfma_f32_32x32x2_f32 correct
// with the wait states hipcc inserts?  (GPU call B/C of round 3: every variant of the march kernel that adds a freshly
// accumulated chunk to another register set with VALU adds gave wrong sums on the chip, the host emulator agrees with
// the reference.)  a = b = 1 on every lane -> each MFMA adds 2 to every accumulator element: expected value 2 * K.
//   mode 0: accumulator wherever hipcc puts it; 1: forced into AGPRs before the chain; 2: chunk pattern of the march kernel
//           (chain from zero, copy out, new chain from zero in the same registers, add the copy to a third set)
//        3: as 1 with `s_nop 7` between the chain and the reads
// hipcc --offload-arch=gfx950 -O3 mfma_tail_read.hip -o mfma_tail_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int MODE>
__global__ __launch_bounds__(256) void k(float* out, float av, float bv, int nrep) {
    float a = av + 0.f * threadIdx.x, b = bv;
    f32x16 total;
    for (int i = 0; i < 16; ++i) total[i] = 0.f;
    for (int rep = 0; rep < nrep; ++rep) {
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+a"(acc[i]));
        }
#pragma unroll
        for (int kk = 0; kk < K; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        if (MODE == 3) {           // 8 more wait states between the chain and the first read
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7");
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 2) {
            f32x16 pend = acc;
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < K; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            total += pend;
        }
        total += acc;
        asm volatile("" : "+v"(a));
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += total[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int K, int MODE>
void run(float* dout) {
    const int nrep = 7;
    hipLaunchKernelGGL((k<K, MODE>), dim3(512), dim3(256), 0, 0, dout, 1.f, 1.f, nrep);
    static float h[512 * 256];
    hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    const float want = 16.f * 2.f * K * nrep * (MODE == 2 ? 2 : 1);
    int bad = 0;
    float worst = want;
    for (int i = 0; i < 512 * 256; ++i)
        if (h[i] != want) { ++bad; if (fabsf(h[i] - want) > fabsf(worst - want)) worst = h[i]; }
    printf("K=%2d mode=%d: expected %.0f, %d of %d lanes differ%s", K, MODE, want, bad, 512 * 256, bad ? "" : "\n");
    if (bad) printf(" (worst value %.0f)\n", worst);
}

int main() {
    float* d;
    hipMalloc(&d, 512 * 256 * 4);
    run<1, 0>(d); run<2, 0>(d); run<4, 0>(d); run<16, 0>(d); run<48, 0>(d);
    run<1, 1>(d); run<2, 1>(d); run<4, 1>(d); run<16, 1>(d); run<48, 1>(d);
    run<1, 2>(d); run<4, 2>(d); run<16, 2>(d); run<48, 2>(d);
    run<1, 3>(d); run<4, 3>(d); run<16, 3>(d); run<48, 3>(d);
    return 0;
}
