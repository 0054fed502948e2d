// Micro-benchmark: the tap loop of conv3d_igemm_kernel (csrc/conv3d.hip) in isolation -- per tap and wave 2 global_load_dwordx4
// of packed weights (ring, two taps ahead), 1 ds_read_b128 of the halo tile (one tap ahead), 8 v_mfma_f32_32x32x2_f32 on two
// accumulators -- with the kernel's structure around it switched on piece by piece:
//   BAR   two workgroup barriers every 27 taps (the K-chunk boundary)
//   STG   13 buffer-like global loads per chunk issued ahead of the taps and stored to LDS between the barriers
//   WSRC  0 = weights from global memory (221 KB, L2-resident), 1 = weights from LDS (one 55 KB chunk staged per chunk)
// and the number of resident workgroups per CU forced through the dynamic LDS size.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TAPS = 27, CHUNKS = 4;

template <int BAR, int STG, int WSRC>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ w, const float* __restrict__ x, int tiles) {
    extern __shared__ float lds[];
    float* tile = lds;                 // 13 200 floats (the dense stride-2 halo tile)
    float* wl = lds + 13200;           // 13 824 floats when WSRC == 1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 13200; i += 256) tile[i] = x[i];
    if (WSRC) for (int i = tid; i < 13824; i += 256) wl[i] = w[i];
    __syncthreads();
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const float* abase = tile + (wave * 66 + (lane & 31)) * 8 + 4 * (lane >> 5);
    float sink = 0.f;
    for (int t = 0; t < tiles; ++t) {
        for (int c = 0; c < CHUNKS; ++c) {
            float4 stg[13];
            if (STG) {
#pragma unroll
                for (int s = 0; s < 13; ++s) stg[s] = *(const float4*)(x + ((size_t)((t * CHUNKS + c) * 13 + s) * 1024 + tid * 4) % (1 << 22));
            }
            const float* wq = WSRC ? wl + lane * 4 : w + (size_t)c * 512 + lane * 4;
            const int wstride = WSRC ? 512 : 2048;    // floats between two taps
            float4 b[3][2], a[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { b[u][0] = *(const float4*)(wq + u * wstride); b[u][1] = *(const float4*)(wq + u * wstride + 256); }
            a[0] = *(const float4*)abase;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                if (tap + 2 < TAPS) {
                    b[(tap + 2) % 3][0] = *(const float4*)(wq + (tap + 2) * wstride);
                    b[(tap + 2) % 3][1] = *(const float4*)(wq + (tap + 2) * wstride + 256);
                }
                if (tap + 1 < TAPS) a[(tap + 1) & 1] = *(const float4*)(abase + ((tap + 1) % 9) * 264 + ((tap + 1) / 9) * 2640);
                __builtin_amdgcn_sched_barrier(0);
                const float4 av = a[tap & 1], b0 = b[tap % 3][0], b1 = b[tap % 3][1];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1.w, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (BAR) __syncthreads();
            if (STG) {
#pragma unroll
                for (int s = 0; s < 13; ++s) *(float4*)(tile + ((tid + s * 256) * 4) % 13200 / 4 * 4) = stg[s];
            }
            if (BAR) __syncthreads();
        }
        sink += acc0[0] + acc1[0];
    }
    float s = sink;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int BAR, int STG, int WSRC>
void run(int per_cu, const float* w, const float* x, float* out) {
    const int tiles = 12;
    const size_t lds = (size_t)(160 * 1024 / per_cu) & ~(size_t)1023;      // exactly per_cu workgroups fit
    if (lds < (13200 + (WSRC ? 13824 : 0)) * 4) { printf("bar=%d stg=%d wsrc=%d wg/CU=%d : tile does not fit\n", BAR, STG, WSRC, per_cu); return; }
    hipFuncSetAttribute((const void*)k<BAR, STG, WSRC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(256 * per_cu), blk(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BAR, STG, WSRC>), grid, blk, lds, 0, out, w, x, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BAR, STG, WSRC>), grid, blk, lds, 0, out, w, x, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid.x * 4 * tiles * CHUNKS * TAPS * 8;
    printf("bar=%d stg=%d wsrc=%d wg/CU=%d : %6.1f TFLOP/s  %.3f of 157.3 (%.3f ms)\n", BAR, STG, WSRC, per_cu,
           mfmas * 4096 / ms / 1e9, mfmas * 4096 / ms / 1e9 / 157.3, ms);
}

// The proposed replacement: one 8-wave workgroup per CU, (2 x 4 x 32)-voxel tile (dense stride-2 halo tile: 23 760 floats), the
// chunk's weights (27 taps x 8 input x 64 output channels: 13 824 floats) in LDS; per chunk 12 + 7 float4 per thread are fetched
// into registers during the taps (halo tile from a 64 MB region: HBM; weights from the 221 KB packed array: L2) and written to
// LDS between two barriers.  PF = 1: the LDS operands one tap ahead (as in conv_chunk_taps27).
template <int STG>
__global__ __launch_bounds__(512) void k8(float* out, const float* __restrict__ w, const float* __restrict__ x, int tiles) {
    extern __shared__ float lds[];
    float* tile = lds;
    float* wl = lds + 23760;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 23760; i += 512) tile[i] = x[i];
    for (int i = tid; i < 13824; i += 512) wl[i] = w[i];
    __syncthreads();
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const float* abase = tile + ((wave >> 2) * 2 * 594 + (wave & 3) * 2 * 66 + (lane & 31)) * 8 + 4 * (lane >> 5);
    const float* wq = wl + lane * 4;
    float sink = 0.f;
    const size_t xbase = (size_t)blockIdx.x * 65536;
    for (int t = 0; t < tiles; ++t) {
        for (int c = 0; c < CHUNKS; ++c) {
            float4 sa[12], sb[7];
            if (STG) {
#pragma unroll
                for (int s = 0; s < 12; ++s) sa[s] = *(const float4*)(x + (xbase + (size_t)((t * CHUNKS + c) * 12 + s) * 2048 + tid * 4) % (1 << 24));
#pragma unroll
                for (int s = 0; s < 7; ++s) sb[s] = *(const float4*)(w + ((size_t)c * 13824 + (s * 512 + tid) * 4) % (1 << 16));
            }
            float4 b[2][2], a[2];
            b[0][0] = *(const float4*)(wq); b[0][1] = *(const float4*)(wq + 256);
            a[0] = *(const float4*)abase;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                if (tap + 1 < TAPS) {
                    b[(tap + 1) & 1][0] = *(const float4*)(wq + (tap + 1) * 512);
                    b[(tap + 1) & 1][1] = *(const float4*)(wq + (tap + 1) * 512 + 256);
                    a[(tap + 1) & 1] = *(const float4*)(abase + ((tap + 1) % 3) * 264 + (((tap + 1) / 3) % 3) * 528 + ((tap + 1) / 9) * 4752);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float4 av = a[tap & 1], b0 = b[tap & 1][0], b1 = b[tap & 1][1];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1.w, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            if (STG) {
#pragma unroll
                for (int s = 0; s < 12; ++s) { const int e = s * 512 + tid; if (e < 5940) *(float4*)(tile + e * 4) = sa[s]; }
#pragma unroll
                for (int s = 0; s < 7; ++s) { const int e = s * 512 + tid; if (e < 3456) *(float4*)(wl + e * 4) = sb[s]; }
            }
            __syncthreads();
        }
        sink += acc0[0] + acc1[0];
    }
    float s = sink;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

template <int STG>
void run8(const float* w, const float* x, float* out) {
    const int tiles = 12;
    const size_t lds = (23760 + 13824) * 4;
    hipFuncSetAttribute((const void*)k8<STG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(256), blk(512);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k8<STG>), grid, blk, lds, 0, out, w, x, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k8<STG>), grid, blk, lds, 0, out, w, x, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid.x * 8 * tiles * CHUNKS * TAPS * 8;
    printf("8 waves, weights + tile in LDS, stg=%d : %6.1f TFLOP/s  %.3f of 157.3 (%.3f ms)\n", STG, mfmas * 4096 / ms / 1e9,
           mfmas * 4096 / ms / 1e9 / 157.3, ms);
}

// Wave mapping variant (round 5): a wave owns TWO 32-voxel rows and ONE 32-column block of the outputs (MT = 2, NT = 1) instead
// of one row and two blocks: per tap 1 global_load_dwordx4 of packed weights (ring, two taps ahead) + 2 ds_read_b128 of the
// tile for the same 8 MFMAs -- half the weight loads per MFMA (the waves of a pair still fetch the same half).
template <int BAR>
__global__ __launch_bounds__(256) void kmap(float* out, const float* __restrict__ w, const float* __restrict__ x, int tiles) {
    extern __shared__ float lds[];
    float* tile = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 13200; i += 256) tile[i] = x[i];
    __syncthreads();
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const float* abase0 = tile + (((wave >> 1) * 2 + 0) * 66 + (lane & 31)) * 8 + 4 * (lane >> 5);
    const float* abase1 = tile + (((wave >> 1) * 2 + 1) * 66 + (lane & 31)) * 8 + 4 * (lane >> 5);
    float sink = 0.f;
    for (int t = 0; t < tiles; ++t) {
        for (int c = 0; c < CHUNKS; ++c) {
            const float* wq = w + (size_t)c * 512 + (wave & 1) * 256 + lane * 4;
            const int wstride = 2048;
            float4 b[3], a[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) b[u] = *(const float4*)(wq + u * wstride);
            a[0][0] = *(const float4*)abase0; a[0][1] = *(const float4*)abase1;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                if (tap + 2 < TAPS) b[(tap + 2) % 3] = *(const float4*)(wq + (tap + 2) * wstride);
                if (tap + 1 < TAPS) {
                    const int o = ((tap + 1) % 9) * 264 + ((tap + 1) / 9) * 2640;
                    a[(tap + 1) & 1][0] = *(const float4*)(abase0 + o);
                    a[(tap + 1) & 1][1] = *(const float4*)(abase1 + o);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float4 a0 = a[tap & 1][0], a1 = a[tap & 1][1], bv = b[tap % 3];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bv.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bv.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bv.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bv.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bv.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bv.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bv.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bv.w, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (BAR) { __syncthreads(); __syncthreads(); }
        }
        sink += acc0[0] + acc1[0];
    }
    float s = sink;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int BAR>
void runmap(int per_cu, const float* w, const float* x, float* out) {
    const int tiles = 12;
    const size_t lds = (size_t)(160 * 1024 / per_cu) & ~(size_t)1023;
    hipFuncSetAttribute((const void*)kmap<BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(256 * per_cu), blk(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kmap<BAR>), grid, blk, lds, 0, out, w, x, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kmap<BAR>), grid, blk, lds, 0, out, w, x, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid.x * 4 * tiles * CHUNKS * TAPS * 8;
    printf("MT=2 NT=1 mapping, bar=%d wg/CU=%d : %6.1f TFLOP/s  %.3f of 157.3 (%.3f ms)\n", BAR, per_cu, mfmas * 4096 / ms / 1e9,
           mfmas * 4096 / ms / 1e9 / 157.3, ms);
}

int main() {
    const size_t nw = 1 << 18, nx = 1 << 24;
    float* h = (float*)malloc(nx * 4);
    srand(1);
    for (size_t i = 0; i < nx; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *w, *x, *out;
    hipMalloc(&w, nw * 4); hipMalloc(&x, nx * 4); hipMalloc(&out, 256 * 4 * 512 * 4);
    hipMemcpy(w, h, nw * 4, hipMemcpyHostToDevice);
    hipMemcpy(x, h, nx * 4, hipMemcpyHostToDevice);
    for (int pc = 1; pc <= 4; ++pc) run<0, 0, 0>(pc, w, x, out);
    for (int pc = 1; pc <= 4; ++pc) run<1, 0, 0>(pc, w, x, out);
    for (int pc = 1; pc <= 3; ++pc) run<1, 1, 0>(pc, w, x, out);
    for (int pc = 1; pc <= 2; ++pc) run<0, 0, 1>(pc, w, x, out);
    for (int pc = 1; pc <= 2; ++pc) run<1, 1, 1>(pc, w, x, out);
    run8<0>(w, x, out); run8<1>(w, x, out);
    for (int pc = 1; pc <= 4; ++pc) runmap<0>(pc, w, x, out);
    for (int pc = 2; pc <= 4; pc += 2) runmap<1>(pc, w, x, out);
    return 0;
}
