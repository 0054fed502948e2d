// Micro-benchmark (round 6): what a GRID-WIDE barrier costs on MI355X, and what a BatchNorm backward pass gains when its
// reduce / column-sum / apply launches become ONE persistent kernel that keeps its operands in registers across two barriers.
//   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier && timeout 120 ./grid_barrier
// The 2-D feature CNN's tensors (17.7 MB at 64 channels, both views) and the 3-D L1 / L2 volumes are read twice by the two
// streaming passes of a train-mode BatchNorm backward (13 + 5 + 10..19 us in three launches); 256 workgroups x 512 threads hold
// such a tensor pair in 36-72 VGPRs per lane.  Every spin loop gives up after ~1 s (no hang, wrong results flagged).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned SPIN_LIMIT = 1u << 24;

// sense-reversing barrier over all workgroups of the grid: bar[0] = arrivals, bar[1] = generation, bar[2] = timeout flag
__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned nblk) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1) {
            __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(&bar[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_LIMIT) { bar[2] = 1u; break; }
            }
        }
        __threadfence();
    }
    __syncthreads();
}

__global__ __launch_bounds__(512) void barrier_loop(unsigned* bar, int n) {
    for (int i = 0; i < n; ++i) grid_barrier(bar, gridDim.x);
}

// ---- the two-pass form (what bn.hip does today, simplified to one source, ReLU mask from y = z * sc + sh) ----
__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ gy, const float* __restrict__ z, const float* __restrict__ par,
                                                     float* __restrict__ partials, size_t nvox, int C) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x, CQ = C >> 2, cq = tid % CQ, vl = tid / CQ, VPB = 256 / CQ;
    const f32x4 m = *(const f32x4*)(par + 4 * cq), is = *(const f32x4*)(par + C + 4 * cq), sc = *(const f32x4*)(par + 2 * C + 4 * cq),
                sh = *(const f32x4*)(par + 3 * C + 4 * cq);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t v = (size_t)blockIdx.x * VPB + vl; v < nvox; v += (size_t)gridDim.x * VPB) {
        f32x4 g = *(const f32x4*)(gy + v * C + 4 * cq);
        const f32x4 a = *(const f32x4*)(z + v * C + 4 * cq);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            g[k] = (a[k] * sc[k] + sh[k]) > 0.f ? g[k] : 0.f;
            s[k] += g[k];
            s[4 + k] += g[k] * (a[k] - m[k]) * is[k];
        }
    }
    for (int k = 0; k < 8; ++k) red[k * 256 + tid] = s[k];
    __syncthreads();
    for (int idx = tid; idx < 8 * CQ; idx += 256) {
        const int k = idx / CQ, q = idx % CQ;
        float t = 0.f;
        for (int j = q; j < 256; j += CQ) t += red[k * 256 + j];
        partials[((size_t)blockIdx.x * 2 + (k >> 2)) * C + 4 * q + (k & 3)] = t;
    }
}
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ partials, int nrows, int M, float* __restrict__ sums) {
    __shared__ double red[256];
    const int m = blockIdx.x, tid = threadIdx.x;
    double s = 0.0;
    for (int r = tid; r < nrows; r += 256) s += (double)partials[(size_t)r * M + m];
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) sums[m] = (float)red[0];
}
__global__ __launch_bounds__(256) void apply_kernel(const float* __restrict__ gy, const float* __restrict__ z, const float* __restrict__ par,
                                                    const float* __restrict__ sums, float* __restrict__ dz, size_t nquads, int CQ, float inv_n) {
    const int C = CQ * 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nquads; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % CQ) * 4;
        f32x4 g = *(const f32x4*)(gy + i * 4);
        const f32x4 a = *(const f32x4*)(z + i * 4);
        const f32x4 m = *(const f32x4*)(par + c), is = *(const f32x4*)(par + C + c), sc = *(const f32x4*)(par + 2 * C + c),
                    sh = *(const f32x4*)(par + 3 * C + c), sg = *(const f32x4*)(sums + c), sx = *(const f32x4*)(sums + C + c);
        f32x4 d;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            g[k] = (a[k] * sc[k] + sh[k]) > 0.f ? g[k] : 0.f;
            d[k] = sc[k] * (g[k] - sg[k] * inv_n - (a[k] - m[k]) * is[k] * sx[k] * inv_n);
        }
        *(f32x4*)(dz + i * 4) = d;
    }
}

// ---- the persistent form: T float4 of gy and z per thread stay in registers across the two barriers ----
template <int T>
__global__ __launch_bounds__(512) void fused_kernel(const float* __restrict__ gy, const float* __restrict__ z, const float* __restrict__ par,
                                                    float* __restrict__ partials, float* __restrict__ sums, float* __restrict__ dz,
                                                    size_t nvox, int C, float inv_n, unsigned* bar) {
    __shared__ float red[512 * 8];
    const int tid = threadIdx.x, CQ = C >> 2, cq = tid % CQ, vl = tid / CQ, VPB = 512 / CQ;
    const f32x4 m = *(const f32x4*)(par + 4 * cq), is = *(const f32x4*)(par + C + 4 * cq), sc = *(const f32x4*)(par + 2 * C + 4 * cq),
                sh = *(const f32x4*)(par + 3 * C + 4 * cq);
    f32x4 g[T], a[T];
    const size_t v0 = (size_t)blockIdx.x * VPB + vl, vstep = (size_t)gridDim.x * VPB;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const size_t v = v0 + t * vstep;
        if (v < nvox) { g[t] = *(const f32x4*)(gy + v * C + 4 * cq); a[t] = *(const f32x4*)(z + v * C + 4 * cq); }
        else { g[t] = f32x4{0, 0, 0, 0}; a[t] = f32x4{0, 0, 0, 0}; }
    }
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            g[t][k] = (a[t][k] * sc[k] + sh[k]) > 0.f ? g[t][k] : 0.f;
            s[k] += g[t][k];
            s[4 + k] += g[t][k] * (a[t][k] - m[k]) * is[k];
        }
    for (int k = 0; k < 8; ++k) red[k * 512 + tid] = s[k];
    __syncthreads();
    for (int idx = tid; idx < 8 * CQ; idx += 512) {
        const int k = idx / CQ, q = idx % CQ;
        float t = 0.f;
        for (int j = q; j < 512; j += CQ) t += red[k * 512 + j];
        partials[((size_t)blockIdx.x * 2 + (k >> 2)) * C + 4 * q + (k & 3)] = t;
    }
    grid_barrier(bar, gridDim.x);
    // column sums: wave w of workgroup b owns column (b * 8 + w) -- fp64, fixed order
    {
        const int wave = tid >> 6, lane = tid & 63, col = blockIdx.x * 8 + wave, M = 2 * C;
        if (col < M) {
            double acc = 0.0;
            for (int r = lane; r < (int)gridDim.x; r += 64) acc += (double)__builtin_nontemporal_load(partials + (size_t)r * M + col);
            for (int d = 32; d > 0; d >>= 1) {
                union { double dd; int i[2]; } u, w;
                u.dd = acc;
                w.i[0] = __shfl_down(u.i[0], d);
                w.i[1] = __shfl_down(u.i[1], d);
                acc += w.dd;
            }
            if (lane == 0) sums[col] = (float)acc;
        }
    }
    grid_barrier(bar, gridDim.x);
    const f32x4 sg = __builtin_nontemporal_load((const f32x4*)(sums + 4 * cq)), sx = __builtin_nontemporal_load((const f32x4*)(sums + C + 4 * cq));
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const size_t v = v0 + t * vstep;
        if (v < nvox) {
            f32x4 d;
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = sc[k] * (g[t][k] - sg[k] * inv_n - (a[t][k] - m[k]) * is[k] * sx[k] * inv_n);
            *(f32x4*)(dz + v * C + 4 * cq) = d;
        }
    }
}

template <typename F>
double timeit(F launch, int iters = 50) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters * 1e3;
}

int main() {
    unsigned* bar;
    hipMalloc(&bar, 64);
    hipMemset(bar, 0, 64);
    for (int nb : {64, 128, 256, 512}) {
        for (int n : {1, 101}) {
            double us = timeit([&] { hipLaunchKernelGGL(barrier_loop, dim3(nb), dim3(512), 0, 0, bar, n); });
            printf("barrier_loop  grid %4d x 512  barriers %3d  %8.2f us per launch\n", nb, n, us);
        }
    }
    unsigned h[3];
    hipMemcpy(h, bar, 12, hipMemcpyDeviceToHost);
    printf("barrier state after the loops: count %u generation %u timeout %u\n", h[0], h[1], h[2]);

    // tensors: (label, voxels, channels): 2-D layer2 both views, layer1 both views, 3-D L1 volume, 3-D L2 volume
    struct Cfg { const char* name; size_t nvox; int C; } cfgs[] = {{"2-D 64ch 2x144x240", 2 * 144 * 240, 64}, {"2-D 32ch 2x288x480", 2 * 288 * 480, 32},
                                                                   {"2-D 128ch 2x144x240", 2 * 144 * 240, 128}, {"3-D L1 64ch 24x72x120", 24 * 72 * 120, 64},
                                                                   {"3-D L2 128ch 12x36x60", 12 * 36 * 60, 128}};
    for (const Cfg& c : cfgs) {
        const size_t n = c.nvox * c.C;
        float *gy, *z, *dz, *dz2, *par, *partials, *sums, *sums2;
        hipMalloc(&gy, n * 4); hipMalloc(&z, n * 4); hipMalloc(&dz, n * 4); hipMalloc(&dz2, n * 4);
        hipMalloc(&par, 4 * c.C * 4); hipMalloc(&partials, 1024 * 2 * c.C * 4); hipMalloc(&sums, 2 * c.C * 4); hipMalloc(&sums2, 2 * c.C * 4);
        float* hb = (float*)malloc(n * 4);
        for (size_t i = 0; i < n; ++i) hb[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
        hipMemcpy(gy, hb, n * 4, hipMemcpyHostToDevice);
        for (size_t i = 0; i < n; ++i) hb[i] = (float)((i * 40503u >> 4) & 0xffff) / 65536.f - 0.4f;
        hipMemcpy(z, hb, n * 4, hipMemcpyHostToDevice);
        float hp[4 * 128];
        for (int k = 0; k < c.C; ++k) { hp[k] = 0.1f; hp[c.C + k] = 1.3f; hp[2 * c.C + k] = 0.9f; hp[3 * c.C + k] = 0.05f; }
        hipMemcpy(par, hp, 4 * c.C * 4, hipMemcpyHostToDevice);
        const float inv_n = 1.f / (float)c.nvox;
        const int nred = (n <= (9u << 19)) ? 256 : 1024;
        const size_t nquads = n / 4;
        const int gapply = (int)((nquads + 255) / 256 > 4096 ? 4096 : (nquads + 255) / 256);
        auto two_pass = [&] {
            hipLaunchKernelGGL(reduce_kernel, dim3(nred), dim3(256), 0, 0, gy, z, par, partials, c.nvox, c.C);
            hipLaunchKernelGGL(colsum_kernel, dim3(2 * c.C), dim3(256), 0, 0, partials, nred, 2 * c.C, sums);
            hipLaunchKernelGGL(apply_kernel, dim3(gapply), dim3(256), 0, 0, gy, z, par, sums, dz, nquads, c.C / 4, inv_n);
        };
        const double t2 = timeit(two_pass);
        for (int nb : {128, 256}) {
            const int VPB = 512 / (c.C / 4);
            const size_t per = (c.nvox + (size_t)nb * VPB - 1) / ((size_t)nb * VPB);
            double tf = -1;
            auto run = [&](auto Tc) {
                constexpr int T = decltype(Tc)::value;
                tf = timeit([&] { hipLaunchKernelGGL(fused_kernel<T>, dim3(nb), dim3(512), 0, 0, gy, z, par, partials, sums2, dz2, c.nvox, c.C, inv_n, bar); });
            };
            if (per <= 4) run(std::integral_constant<int, 4>());
            else if (per <= 8) run(std::integral_constant<int, 8>());
            else if (per <= 12) run(std::integral_constant<int, 12>());
            else if (per <= 16) run(std::integral_constant<int, 16>());
            else if (per <= 24) run(std::integral_constant<int, 24>());
            // compare results
            float* r1 = (float*)malloc(n * 4);
            float* r2 = (float*)malloc(n * 4);
            hipMemcpy(r1, dz, n * 4, hipMemcpyDeviceToHost);
            hipMemcpy(r2, dz2, n * 4, hipMemcpyDeviceToHost);
            double md = 0, mx = 0;
            for (size_t i = 0; i < n; ++i) { double d = fabs((double)r1[i] - r2[i]); md = d > md ? d : md; mx = fabs(r1[i]) > mx ? fabs(r1[i]) : mx; }
            free(r1); free(r2);
            printf("%-24s  %6.1f MB/tensor  two-pass (3 launches) %7.2f us   fused grid %3d (T<=%2d) %7.2f us   max|diff| %.3g of %.3g\n", c.name,
                   n * 4 / 1e6, t2, nb, (int)per, tf, md, mx);
        }
        hipFree(gy); hipFree(z); hipFree(dz); hipFree(dz2); hipFree(par); hipFree(partials); hipFree(sums); hipFree(sums2);
        free(hb);
    }
    hipMemcpy(h, bar, 12, hipMemcpyDeviceToHost);
    printf("barrier state at the end: count %u generation %u timeout %u\n", h[0], h[1], h[2]);
    return 0;
}
