// Micro-benchmark (round 6): which WRITE PATTERN of the cost-volume builder's output stream the chip takes fastest.
//   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern && ./store_pattern
// The volume is [D'=48][H=144][W=240][64 ch] fp32 (424.7 MB): a voxel is 256 B, an image row of one d-plane 60 KiB, a d-plane
// 8.8 MB.  The builder's unit (round 2-5) is 16 d x 16 w: sixteen 4-KiB runs one d-plane apart, units walked k-fastest
// (all D' of a 16-column tile, then the next tile of the row); tools/ubench/store_stream.hip measured 5.05-5.27 TB/s for that
// against 6.46 TB/s for a flat fill.  Variants here: unit = R d-planes x T columns (run = T x 256 B) with R x T = 256 voxels
// (the LDS image the kernel can afford), both walk orders, 6 / 10 store waves, plain / non-temporal stores; plus whole-row units.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// unit = R planes x T columns; kfast: (h, tile, k) order (k fastest: the builder's), else (h, k, tile)
template <bool NT>
__global__ __launch_bounds__(1024) void fill_rt(float* __restrict__ p, int units, int tiles, int nk, int R, int T, int kfast,
                                                size_t plane4, size_t row4, float v) {
    f32x4 t = {v, v, v, v};
    const int u0 = (int)((long long)units * blockIdx.x / gridDim.x), u1 = (int)((long long)units * (blockIdx.x + 1) / gridDim.x);
    const int run4 = T * 16;                       // float4 per run (T voxels x 64 ch / 4)
    for (int u = u0; u < u1; ++u) {
        int tile, k, h;
        if (kfast) { k = u % nk; tile = (u / nk) % tiles; h = u / (nk * tiles); }
        else { tile = u % tiles; k = (u / tiles) % nk; h = u / (nk * tiles); }
        f32x4* base = reinterpret_cast<f32x4*>(p) + (size_t)k * R * plane4 + (size_t)h * row4 + (size_t)tile * run4;
        for (int idx = threadIdx.x; idx < R * run4; idx += blockDim.x) {
            f32x4* d = base + (size_t)(idx / run4) * plane4 + (idx % run4);
            if (NT) __builtin_nontemporal_store(t, d);
            else *d = t;
        }
        __syncthreads();
    }
}

template <typename F>
double timeit(F launch, int iters = 20) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

// Same units, dealt to the workgroups in CHUNKS of `chunk` consecutive units, chunk j * grid + b to workgroup b (chunk = units /
// grid is the contiguous-run assignment above; chunk = 1 makes the 256 workgroups write 256 adjacent units at any moment -- the
// chip-wide write front of the flat fill).
template <bool NT>
__global__ __launch_bounds__(1024) void fill_rt_il(float* __restrict__ p, int units, int tiles, int nk, int R, int T, int kfast,
                                                   int chunk, size_t plane4, size_t row4, float v) {
    f32x4 t = {v, v, v, v};
    const int run4 = T * 16;
    const int nchunks = (units + chunk - 1) / chunk;
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int u1 = (c + 1) * chunk < units ? (c + 1) * chunk : units;
        for (int u = c * chunk; u < u1; ++u) {
            int tile, k, h;
            if (kfast) { k = u % nk; tile = (u / nk) % tiles; h = u / (nk * tiles); }
            else { tile = u % tiles; k = (u / tiles) % nk; h = u / (nk * tiles); }
            f32x4* base = reinterpret_cast<f32x4*>(p) + (size_t)k * R * plane4 + (size_t)h * row4 + (size_t)tile * run4;
            for (int idx = threadIdx.x; idx < R * run4; idx += blockDim.x) {
                f32x4* d = base + (size_t)(idx / run4) * plane4 + (idx % run4);
                if (NT) __builtin_nontemporal_store(t, d);
                else *d = t;
            }
            __syncthreads();
        }
    }
}

int main() {
    const int D = 48, H = 144, W = 240, CT = 64;
    const size_t n = (size_t)D * H * W * CT, bytes = n * 4;
    float* a;
    hipMalloc(&a, bytes);
    hipMemset(a, 0, bytes);
    const size_t plane4 = (size_t)H * W * CT / 4, row4 = (size_t)W * CT / 4;
    printf("buffer %.1f MB; unit = R d-planes x T columns (run = T x 256 B)\n", bytes / 1e6);
    const int RT[][2] = {{16, 16}, {8, 32}, {4, 80}, {1, 240}, {16, 240}, {48, 16}};
    for (auto& rt : RT) {
        const int R = rt[0], T = rt[1];
        if (D % R || W % T) continue;
        const int tiles = W / T, nk = D / R, units = H * nk * tiles;
        for (int kfast = 0; kfast <= 1; ++kfast)
            for (int waves = 6; waves <= 10; waves += 4)
                for (int wg = 1; wg <= 2; ++wg) {
                    double ms = timeit([&] { hipLaunchKernelGGL(fill_rt<true>, dim3(256 * wg), dim3(waves * 64), 0, 0, a, units, tiles, nk, R, T, kfast, plane4, row4, 1.f); });
                    double ms2 = timeit([&] { hipLaunchKernelGGL(fill_rt<false>, dim3(256 * wg), dim3(waves * 64), 0, 0, a, units, tiles, nk, R, T, kfast, plane4, row4, 1.f); });
                    printf("R=%2d T=%3d (%5.1f KiB runs, unit %4d KiB) order=%s waves=%2d WG/CU=%d: nt %.4f ms %.2f TB/s | plain %.4f ms %.2f TB/s\n", R, T,
                           T * 0.25, R * T / 4, kfast ? "k-fast" : "tile-fast", waves, wg, ms, bytes / ms / 1e9, ms2, bytes / ms2 / 1e9);
                }
    }
    printf("\n-- interleaved assignment (chunk = consecutive units per deal; k-fast: a chunk of 3 = one macro-unit of the builder)\n");
    {
        const int R = 16, T = 16, tiles = W / T, nk = D / R, units = H * nk * tiles;
        for (int kfast = 0; kfast <= 1; ++kfast)
            for (int chunk : {1, 3, 6, 12, 24, 48, 96})
                for (int waves = 6; waves <= 10; waves += 4)
                    for (int wg = 1; wg <= 2; ++wg) {
                        double ms = timeit([&] { hipLaunchKernelGGL(fill_rt_il<true>, dim3(256 * wg), dim3(waves * 64), 0, 0, a, units, tiles, nk, R, T, kfast, chunk, plane4, row4, 1.f); });
                        double ms2 = timeit([&] { hipLaunchKernelGGL(fill_rt_il<false>, dim3(256 * wg), dim3(waves * 64), 0, 0, a, units, tiles, nk, R, T, kfast, chunk, plane4, row4, 1.f); });
                        printf("interleaved R=16 T=16 order=%s chunk=%2d waves=%2d WG/CU=%d: nt %.4f ms %.2f TB/s | plain %.4f ms %.2f TB/s\n", kfast ? "k-fast" : "tile-fast",
                               chunk, waves, wg, ms, bytes / ms / 1e9, ms2, bytes / ms2 / 1e9);
                    }
    }
    return 0;
}
