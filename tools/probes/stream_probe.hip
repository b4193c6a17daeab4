// How fast can one wave-per-slice kernel stream a SELL-like value array?  (diagnosis of the 4.3-4.5 TB/s of k_sell_spmv
// at 10 M DOF against 6.2 TB/s of the CG update kernel.)  Build: hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// variant 0: 8 B per lane, W loads of 512 B per slice, all issued before the sum (what dia_round does)
template <int W, bool NT>
__global__ void __launch_bounds__(256) k_stream8(const double* __restrict__ v, int64_t n_slices, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = 0.0;
    for (int64_t c = blockIdx.x; c * 4 + wave < n_slices; c += gridDim.x) {
        const double* p = v + (c * 4 + wave) * (int64_t)(W * 64) + lane;
        double t[W];
#pragma unroll
        for (int k = 0; k < W; ++k) t[k] = NT ? __builtin_nontemporal_load(&p[k * 64]) : p[k * 64];
#pragma unroll
        for (int k = 0; k < W; ++k) acc += t[k];
    }
    if (acc == 1.2345e300) out[0] = acc;
}
// variant 1: 16 B per lane
template <int W, bool NT>
__global__ void __launch_bounds__(256) k_stream16(const double* __restrict__ v, int64_t n_slices, double* __restrict__ out) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = 0.0;
    for (int64_t c = blockIdx.x; c * 4 + wave < n_slices; c += gridDim.x) {
        const v2d* p = reinterpret_cast<const v2d*>(v + (c * 4 + wave) * (int64_t)(W * 64)) + lane;
        v2d t[W / 2];
#pragma unroll
        for (int k = 0; k < W / 2; ++k) t[k] = NT ? __builtin_nontemporal_load(&p[k * 64]) : p[k * 64];
#pragma unroll
        for (int k = 0; k < W / 2; ++k) acc += t[k].x + t[k].y;
    }
    if (acc == 1.2345e300) out[0] = acc;
}
// variant 2: as 0, plus a second stream of the same size read at a shifted offset (the x vector of a DIA slice: L2 hits)
template <int W>
__global__ void __launch_bounds__(256) k_stream8x(const double* __restrict__ v, const double* __restrict__ x, int64_t n_slices, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = 0.0;
    for (int64_t c = blockIdx.x; c * 4 + wave < n_slices; c += gridDim.x) {
        const int64_t s = c * 4 + wave;
        const double* p = v + s * (int64_t)(W * 64) + lane;
        const double* q = x + s * 64 + lane;
        double t[W], u[W];
#pragma unroll
        for (int k = 0; k < W; ++k) t[k] = __builtin_nontemporal_load(&p[k * 64]);
#pragma unroll
        for (int k = 0; k < W; ++k) u[k] = q[(k / 3 % 3 - 1) * 216 + (k / 9 % 3 - 1) * 46656 + (k % 3 - 1) + 50000];
#pragma unroll
        for (int k = 0; k < W; ++k) acc += t[k] * u[k];
    }
    if (acc == 1.2345e300) out[0] = acc;
}

// variant 3: the x stream read once per run of consecutive offsets (-1, 0, +1), neighbours through wave shuffles
template <int W>
__global__ void __launch_bounds__(256) k_stream8x_shfl(const double* __restrict__ v, const double* __restrict__ x, int64_t n_slices, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = 0.0;
    for (int64_t c = blockIdx.x; c * 4 + wave < n_slices; c += gridDim.x) {
        const int64_t s = c * 4 + wave;
        const double* p = v + s * (int64_t)(W * 64) + lane;
        const double* q = x + s * 64 + lane;
        double t[W], u[W];
#pragma unroll
        for (int k = 0; k < W; ++k) t[k] = __builtin_nontemporal_load(&p[k * 64]);
#pragma unroll
        for (int g = 0; g * 3 < W; ++g) {          // run g: offsets base-1, base, base+1
            const int64_t base = (g % 3 - 1) * 216 + (g / 3 % 3 - 1) * 46656 + 50000;
            const double mid = q[base];
            const double lo_edge = q[base - 1 - lane];          // x[first - 1]   (same address for all lanes)
            const double hi_edge = q[base + 64 - lane];         // x[last + 1]
            double left = __shfl_up(mid, 1, 64), right = __shfl_down(mid, 1, 64);
            left = lane == 0 ? lo_edge : left;
            right = lane == 63 ? hi_edge : right;
            if (g * 3 < W) u[g * 3] = left;
            if (g * 3 + 1 < W) u[g * 3 + 1] = mid;
            if (g * 3 + 2 < W) u[g * 3 + 2] = right;
        }
#pragma unroll
        for (int k = 0; k < W; ++k) acc += t[k] * u[k];
    }
    if (acc == 1.2345e300) out[0] = acc;
}

int main() {
    const int W = 16;
    const int64_t n_slices = 157464;                 // 10 M rows
    const int64_t n = n_slices * W * 64;
    double *v, *x, *out;
    CHECK(hipMalloc(&v, n * 8));
    CHECK(hipMalloc(&x, (n_slices * 64 + 200000) * 8));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(v, 0, n * 8));
    CHECK(hipMemset(x, 0, (n_slices * 64 + 200000) * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch, double bytes) {
        launch();
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0, 0);
            for (int i = 0; i < 10; ++i) launch();
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-44s %8.1f us  %7.2f TB/s\n", name, best * 100, bytes / (best / 10 * 1e-3) / 1e12);
        return 0;
    };
    const double bytes = (double)n * 8;
    for (int grid : {512, 1024}) {
        printf("grid %d\n", grid);
        run("8 B/lane, 16 loads in flight", [&] { hipLaunchKernelGGL((k_stream8<W, false>), dim3(grid), dim3(256), 0, 0, v, n_slices, out); }, bytes);
        run("8 B/lane, non-temporal", [&] { hipLaunchKernelGGL((k_stream8<W, true>), dim3(grid), dim3(256), 0, 0, v, n_slices, out); }, bytes);
        run("16 B/lane, 8 loads in flight", [&] { hipLaunchKernelGGL((k_stream16<W, false>), dim3(grid), dim3(256), 0, 0, v, n_slices, out); }, bytes);
        run("16 B/lane, non-temporal", [&] { hipLaunchKernelGGL((k_stream16<W, true>), dim3(grid), dim3(256), 0, 0, v, n_slices, out); }, bytes);
        run("8 B/lane NT + shifted x reads (L2)", [&] { hipLaunchKernelGGL((k_stream8x<W>), dim3(grid), dim3(256), 0, 0, v, x, n_slices, out); }, bytes);
        run("8 B/lane NT + x once per run, shuffles", [&] { hipLaunchKernelGGL((k_stream8x_shfl<W>), dim3(grid), dim3(256), 0, 0, v, x, n_slices, out); }, bytes);
    }
    return 0;
}
