// Micro-benchmark: what a write-dominated HBM stream reaches on MI355X -- the practical ceiling of the cost-volume
// builders (algorithmic traffic of the GwcNet_GC build: 92 MB read + 425 MB written).
//   hipcc --offload-arch=gfx950 -O3 store_stream.hip -o store_stream && ./store_stream
// Variants: float4 stores of a 425 MB buffer (plain / non-temporal), 1 KiB contiguous per wave instruction, chunked
// per workgroup like the builder's units (4-KiB d-rows) or flat; a float4 copy (read 212 + write 212 MB) as the
// guide's reference point (6.29 TB/s); hipMemsetAsync.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void fill_flat(float* __restrict__ p, size_t n4, float v) {
    f32x4 t = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4* d = reinterpret_cast<f32x4*>(p) + i;
        if (NT) __builtin_nontemporal_store(t, d);
        else *d = t;
    }
}

// unit = 16 rows of 4 KiB, the rows `rstride` bytes apart (like (d, h, w-tile) rows of the NDHWC volume)
template <bool NT>
__global__ __launch_bounds__(640) void fill_units(float* __restrict__ p, int units, int tiles_per_row, int nk,
                                                   size_t plane4, size_t row4, float v) {
    f32x4 t = {v, v, v, v};
    const int u0 = (int)((long long)units * blockIdx.x / gridDim.x), u1 = (int)((long long)units * (blockIdx.x + 1) / gridDim.x);
    for (int u = u0; u < u1; ++u) {
        const int tile = u % tiles_per_row, rest = u / tiles_per_row, k = rest % nk, h = rest / nk;
        f32x4* base = reinterpret_cast<f32x4*>(p) + (size_t)k * 16 * plane4 + (size_t)h * row4 + (size_t)tile * 256;
        for (int idx = threadIdx.x; idx < 4096; idx += 640) {
            f32x4* d = base + (size_t)(idx >> 8) * plane4 + (idx & 255);
            if (NT) __builtin_nontemporal_store(t, d);
            else *d = t;
        }
        __syncthreads();
    }
}

// the builder's MIX: per unit (64 KiB written as 16 rows of 4 KiB) a workgroup also READS its share of the two 320-channel feature
// maps -- per macro-unit (3 units) one 16-column tile of every channel row of both views = 640 pieces of 64 bytes, HW * 4 bytes
// apart -- by `nload` loader waves while the remaining waves store.  92 MB read per 425 MB written, like the GwcNet_GC build.
template <bool NT>
__global__ __launch_bounds__(1024) void mix_units(float* __restrict__ p, const float* __restrict__ feat, float* __restrict__ sink,
                                                   int units, int tiles_per_row, int nk, size_t plane4, size_t row4, int HW, int W,
                                                   int nload, float v) {
    f32x4 t = {v, v, v, v};
    const int u0 = (int)((long long)units * blockIdx.x / gridDim.x), u1 = (int)((long long)units * (blockIdx.x + 1) / gridDim.x);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nstore = blockDim.x - nload * 64;
    float acc = 0.f;
    for (int u = u0; u < u1; ++u) {
        const int tile = u % tiles_per_row, rest = u / tiles_per_row, k = rest % nk, h = rest / nk;
        if (wave < nload) {
            if (k == 0) {       // one macro-unit's features: 640 channel rows x 16 columns, 4 rows per wave instruction
                for (int r = wave * 4 + (lane >> 4); r < 640; r += nload * 4)
                    acc += feat[(size_t)r * HW + (size_t)h * W + tile * 16 + (lane & 15)];
            }
        } else {
            f32x4* base = reinterpret_cast<f32x4*>(p) + (size_t)k * 16 * plane4 + (size_t)h * row4 + (size_t)tile * 256;
            for (int idx = threadIdx.x - nload * 64; idx < 4096; idx += nstore) {
                f32x4* d = base + (size_t)(idx >> 8) * plane4 + (idx & 255);
                if (NT) __builtin_nontemporal_store(t, d);
                else *d = t;
            }
        }
        __syncthreads();
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void copy4(const float* __restrict__ s, float* __restrict__ d, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        reinterpret_cast<f32x4*>(d)[i] = reinterpret_cast<const f32x4*>(s)[i];
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

int main() {
    const int D = 48, H = 144, W = 240, CT = 64;
    const size_t n = (size_t)D * H * W * CT, bytes = n * 4, n4 = n / 4;
    float *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    printf("buffer %.1f MB\n", bytes / 1e6);
    for (int wg = 1; wg <= 8; wg *= 2) {
        double ms = timeit([&] { hipLaunchKernelGGL(fill_flat<false>, dim3(256 * wg), dim3(256), 0, 0, a, n4, 1.f); });
        printf("fill flat plain  %d WG/CU: %.4f ms  %.2f TB/s\n", wg, ms, bytes / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(fill_flat<true>, dim3(256 * wg), dim3(256), 0, 0, a, n4, 1.f); });
        printf("fill flat nt     %d WG/CU: %.4f ms  %.2f TB/s\n", wg, ms, bytes / ms / 1e9);
    }
    // builder-shaped: units (h, k, tile): rows of a unit are one d-plane apart
    const int tiles = W / 16, units = H * (D / 16) * tiles;
    const size_t plane4 = (size_t)H * W * CT / 4, row4 = (size_t)W * CT / 4;     // one d step / one image row, in float4
    for (int wg = 1; wg <= 3; ++wg) {
        double ms = timeit([&] { hipLaunchKernelGGL(fill_units<false>, dim3(256 * wg), dim3(640), 0, 0, a, units, tiles, D / 16, plane4, row4, 1.f); });
        printf("fill units plain %d WG/CU: %.4f ms  %.2f TB/s\n", wg, ms, (double)units * 65536 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(fill_units<true>, dim3(256 * wg), dim3(640), 0, 0, a, units, tiles, D / 16, plane4, row4, 1.f); });
        printf("fill units nt    %d WG/CU: %.4f ms  %.2f TB/s\n", wg, ms, (double)units * 65536 / ms / 1e9);
    }
    {
        // the mixed stream: units in (h, k, tile) order as above; features [640][H][W]
        float* feat; float* sink;
        hipMalloc(&feat, (size_t)640 * H * W * 4); hipMalloc(&sink, 64);
        hipMemset(feat, 0, (size_t)640 * H * W * 4);
        const int units_hkt = H * (D / 16) * tiles;
        for (int nload = 2; nload <= 10; nload += 4)
            for (int nstorew = 6; nstorew <= 6; ++nstorew) {
                const int thr = (nload + nstorew) * 64;
                double ms = timeit([&] { hipLaunchKernelGGL(mix_units<true>, dim3(256), dim3(thr), 0, 0, a, feat, sink, units_hkt, tiles, D / 16, plane4, row4, H * W, W, nload, 1.f); });
                const double rd = (double)640 * H * W * 4;
                printf("mix units nt: %2d loader + %d store waves: %.4f ms  written %.2f TB/s, read+written %.2f TB/s (%.0f + %.0f MB)\n", nload, nstorew, ms,
                       (double)units_hkt * 65536 / ms / 1e9, ((double)units_hkt * 65536 + rd) / ms / 1e9, rd / 1e6, (double)units_hkt * 65536 / 1e6);
            }
    }
    {
        double ms = timeit([&] { hipLaunchKernelGGL(copy4, dim3(256 * 8), dim3(256), 0, 0, a, b, n4 / 2); });
        printf("copy float4 (read %.0f MB + write %.0f MB): %.4f ms  %.2f TB/s\n", bytes / 2e6, bytes / 2e6, ms, bytes / ms / 1e9);
        ms = timeit([&] { hipMemsetAsync(a, 0, bytes, 0); });
        printf("hipMemsetAsync: %.4f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
    }
    return 0;
}
