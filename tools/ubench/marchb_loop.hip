// Sized experiment, stage 0 of a split-bf16 "march" kernel (NOT product code): the plane step of conv3d_marchw_kernel with both
// operands as three bf16 pieces and 6 x v_mfma_f32_32x32x16_bf16 per K = 16 -- in the geometry that FITS the LDS:
//   * 4 waves, one per SIMD, one workgroup per CU, a wave owns 32 voxels x 32 output channels (R = 1: two row blocks per wave
//     would need 135 KB of three-piece plane buffers);
//   * the 27 x 32 x 32 weights as three bf16 pieces are 166 KB: they cannot be resident.  A ring of three kh-rows (3 taps x 3
//     pieces x 2 K steps x 1 KB = 18 KB each) is refilled from L2 through registers while the previous rows are multiplied
//     (loads at the start of a row, ds_write at its end, one barrier per row = 9 per plane);
//   * the next plane (fp32 in memory) is fetched during the step, split into three pieces in registers and written to the other
//     plane buffer in mid-step (10 x 18 voxels x 208 B = 37 KB per buffer).
// LDS: 54 + 2 x 37 = 128 KB.  Reports fp32-equivalent TFLOP/s next to the fp32 march kernel's measured 135 (0.680 ms per launch).
//   hipcc --offload-arch=gfx950 -O3 marchb_loop.hip -o marchb_loop && ./marchb_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int VS = 208;                 // bytes per staged voxel: 3 pieces x 64 B + 16 B pad
constexpr int PLANE = 10 * 18 * VS;     // 37 440 B
constexpr int ROW = 3 * 3 * 2 * 1024;   // one kh-row of weights: taps x pieces x K steps x 1 KiB
constexpr int RING = 3 * ROW;

template <int STREAM_W, int STAGE_P>
__global__ __launch_bounds__(256) void marchb(float* out, const uint4* __restrict__ wsplit, const float* __restrict__ x, int planes) {
    extern __shared__ char lds[];
    char* ring = lds;
    char* pl = lds + RING;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (RING + 2 * PLANE) / 16; i += 256) ((uint4*)lds)[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    __syncthreads();
    f32x16 acc[3];
    for (int m = 0; m < 3; ++m)
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
    const int abase = ((wave * 2 + (lane & 31) / 16) * 18 + (lane & 15)) * VS + (lane >> 5) * 16;
    const size_t xbase = (size_t)blockIdx.x * 180 * 32;
    for (int p = 0; p < planes; ++p) {
        const char* pbuf = pl + (p & 1) * PLANE;
        char* nbuf = pl + ((p & 1) ^ 1) * PLANE;
        float4 stg[6];
        if (STAGE_P) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int e = tid + k * 256;       // float4 index inside the 180-voxel x 32-channel plane (1440 float4)
                stg[k] = e < 1440 ? *(const float4*)(x + ((xbase + (size_t)p * 8192 * 32) % (1u << 24)) + (size_t)e * 4) : make_float4(0, 0, 0, 0);
            }
        }
#pragma unroll 1
        for (int row = 0; row < 9; ++row) {
            const int slot = (p * 9 + row) % 3;
            uint4 wst[5];
            if (STREAM_W) {
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int e = tid + k * 256;   // uint4 index inside the row (1152 per row)
                    wst[k] = e < 1152 ? wsplit[(size_t)((row + 2) % 9) * 1152 + e] : make_uint4(0, 0, 0, 0);
                }
            }
            const char* wrow = ring + slot * ROW + lane * 16;
            // the 6 K steps of the row (3 taps x 2), operands ONE K step ahead in registers
            bf16x8 a[2][3], b[2][3];
            auto load_k = [&](int ks, int buf) {
                const int kw = ks >> 1, sidx = ks & 1;
                const char* ap = pbuf + abase + ((row % 3) * 18 + kw) * VS;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    a[buf][q] = *(const bf16x8*)(ap + q * 64 + sidx * 32);
                    b[buf][q] = *(const bf16x8*)(wrow + ((kw * 3 + q) * 2 + sidx) * 1024);
                }
            };
            load_k(0, 0);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                if (ks + 1 < 6) load_k(ks + 1, (ks + 1) & 1);
                const int cb = ks & 1;
                f32x16& c = acc[row / 3];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cb][2], b[cb][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cb][1], b[cb][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cb][0], b[cb][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cb][1], b[cb][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cb][0], b[cb][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cb][0], b[cb][0], c, 0, 0, 0);
                if (ks + 1 < 6) {
#pragma unroll
                    for (int g = 0; g < 6; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (STAGE_P && row == 4) {
                // split the next plane into three bf16 pieces and write it to the other buffer: [voxel][piece][channel]
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int e = tid + k * 256;
                    if (e < 1440) {
                        const int v = e >> 3, f = e & 7;
                        float r[4] = {stg[k].x, stg[k].y, stg[k].z, stg[k].w};
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            bf16x4 pc;
#pragma unroll
                            for (int j = 0; j < 4; ++j) { pc[j] = (__bf16)r[j]; r[j] -= (float)pc[j]; }
                            *(bf16x4*)(nbuf + v * VS + q * 64 + f * 8) = pc;
                        }
                    }
                }
            }
            if (STREAM_W) {
                char* dst = ring + ((slot + 2) % 3) * ROW;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int e = tid + k * 256;
                    if (e < 1152) *(uint4*)(dst + e * 16) = wst[k];
                }
            }
            __syncthreads();
        }
    }
    float t = 0.f;
    for (int m = 0; m < 3; ++m)
        for (int i = 0; i < 16; ++i) t += acc[m][i];
    out[(size_t)blockIdx.x * 256 + tid] = t;
}

template <int STREAM_W, int STAGE_P>
void run(float* out, const uint4* w, const float* x) {
    const int planes = 96;
    const size_t lds = RING + 2 * PLANE;
    hipFuncSetAttribute((const void*)marchb<STREAM_W, STAGE_P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((marchb<STREAM_W, STAGE_P>), dim3(256), dim3(256), lds, 0, out, w, x, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((marchb<STREAM_W, STAGE_P>), dim3(256), dim3(256), lds, 0, out, w, x, planes);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)256 * 4 * planes * 27 * 32 * 32 * 32 * 2;       // per wave and plane: 27 taps x 32 voxels x 32 x 32 MACs
    printf("marchb  weights %s, planes %s : %.3f ms  %6.1f TFLOP/s fp32-equivalent = %.2f x the fp32 march kernel's 135.0 (%.2f of the 157.3 pipe)\n",
           STREAM_W ? "streamed (ring of 3 rows)" : "resident (no refill)    ", STAGE_P ? "staged + split" : "resident      ", ms, flop / ms / 1e9,
           flop / ms / 1e9 / 135.0, flop / ms / 1e9 / 157.3);
}

int main() {
    uint4* w; float *x, *out;
    hipMalloc(&w, (size_t)9 * 1152 * 16); hipMalloc(&x, (size_t)(1u << 24) * 4 + (1 << 20)); hipMalloc(&out, (size_t)256 * 256 * 4);
    hipMemset(w, 0x3f, (size_t)9 * 1152 * 16); hipMemset(x, 0, (size_t)(1u << 24) * 4 + (1 << 20));
    run<0, 0>(out, w, x);
    run<1, 0>(out, w, x);
    run<0, 1>(out, w, x);
    run<1, 1>(out, w, x);
    return 0;
}
