// Which kernel SHAPE moves the operands of a 15-offset DIA product (10 M rows, 216^3 Kuhn mesh) fastest?
// PMC on the production kernel (round 2): TA busy 70 %, 31 % issue stalls, time proportional to L1 accesses and
// insensitive to the HBM byte count -> the per-CU address/L1 path is the limit, not HBM.  Candidates:
//   A  one row per lane, 8 B loads: 15 value + 15 x loads per 64 rows                      (production shape)
//   B  two rows per lane, 16 B loads: 15 + 15 loads per 128 rows
//   C  A, but the x windows of a 256-row chunk staged once in LDS (7 runs of consecutive offsets)
//   D  B + C: 16 B value loads, x through LDS (chunk = 512 rows)
// Build: hipcc --offload-arch=gfx950 -O3 spmv_shape_probe.hip -o spmv_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int W = 15;
constexpr int NN = 216, PP = 216 * 216;
__constant__ int c_off[W] = {-PP - NN - 1, -PP - NN, -PP - 1, -PP, -NN - 1, -NN, -1, 0, 1, NN, NN + 1, PP, PP + 1, PP + NN, PP + NN + 1};
// runs of consecutive offsets: first offset, length, position of the first offset in c_off
__constant__ int c_run_base[7] = {-PP - NN - 1, -PP - 1, -NN - 1, -1, NN, PP, PP + NN};
__constant__ int c_run_len[7] = {2, 2, 2, 3, 2, 2, 2};
__constant__ int c_run_pos[7] = {0, 2, 4, 6, 9, 11, 13};

// A: production shape
template <bool NT>
__global__ void __launch_bounds__(256) k_A(const double* __restrict__ v, const double* __restrict__ x, double* __restrict__ y, int64_t n_slices) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t c = blockIdx.x; c * 4 + wave < n_slices; c += gridDim.x) {
        const int64_t s = c * 4 + wave;
        const double* p = v + s * (int64_t)(W * 64) + lane;
        const int64_t r = s * 64 + lane;
        double t[W], u[W];
#pragma unroll
        for (int k = 0; k < W; ++k) t[k] = NT ? __builtin_nontemporal_load(&p[k * 64]) : p[k * 64];
#pragma unroll
        for (int k = 0; k < W; ++k) u[k] = x[r + c_off[k]];
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < W; ++k) acc += t[k] * u[k];
        y[r] = acc;
    }
}
// B: two rows per lane; lanes 0-31 own slice 2q, lanes 32-63 slice 2q+1 (value planes stay 512-B contiguous per slice)
template <bool NT>
__global__ void __launch_bounds__(256) k_B(const double* __restrict__ v, const double* __restrict__ x, double* __restrict__ y, int64_t n_slices) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l2 = (lane & 31) * 2;
    for (int64_t c = blockIdx.x; (c * 4 + wave) * 2 + 1 < n_slices; c += gridDim.x) {
        const int64_t s = (c * 4 + wave) * 2 + half;
        const double* p = v + s * (int64_t)(W * 64) + l2;
        const int64_t r = s * 64 + l2;
        v2d t[W], u[W];
#pragma unroll
        for (int k = 0; k < W; ++k) t[k] = NT ? __builtin_nontemporal_load((const v2d*)&p[k * 64]) : *(const v2d*)&p[k * 64];
#pragma unroll
        for (int k = 0; k < W; ++k) { u[k].x = x[r + c_off[k]]; u[k].y = x[r + c_off[k] + 1]; }   // the compiler merges into one 16-B load when it can
        v2d acc = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < W; ++k) acc += t[k] * u[k];
        *(v2d*)&y[r] = acc;
    }
}
// C: x windows of the chunk (256 rows) in LDS
template <bool NT>
__global__ void __launch_bounds__(256) k_C(const double* __restrict__ v, const double* __restrict__ x, double* __restrict__ y, int64_t n_slices) {
    __shared__ double win[7][264];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t c = blockIdx.x; c * 4 + 3 < n_slices; c += gridDim.x) {
        const int64_t r0 = c * 256;
        const int64_t s = c * 4 + wave;
        const double* p = v + s * (int64_t)(W * 64) + lane;
        double t[W];
#pragma unroll
        for (int k = 0; k < W; ++k) t[k] = NT ? __builtin_nontemporal_load(&p[k * 64]) : p[k * 64];
        __syncthreads();     // previous chunk's readers are done
        // 7 windows x 258 doubles: thread i loads element i of every window, the first threads also the 2-3 tail elements
#pragma unroll
        for (int g = 0; g < 7; ++g) {
            win[g][threadIdx.x] = x[r0 + c_run_base[g] + threadIdx.x];
            if (threadIdx.x < c_run_len[g] - 1) win[g][256 + threadIdx.x] = x[r0 + c_run_base[g] + 256 + threadIdx.x];
        }
        __syncthreads();
        double acc = 0.0;
#pragma unroll
        for (int g = 0; g < 7; ++g)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < c_run_len[g]) acc += t[c_run_pos[g] + j] * win[g][threadIdx.x + j];
        y[r0 + threadIdx.x] = acc;
    }
}
// D: chunk = 512 rows, two rows per lane, 16-B value loads, x windows in LDS
template <bool NT>
__global__ void __launch_bounds__(256) k_D(const double* __restrict__ v, const double* __restrict__ x, double* __restrict__ y, int64_t n_slices) {
    __shared__ double win[7][520];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l2 = (lane & 31) * 2;
    for (int64_t c = blockIdx.x; c * 8 + 7 < n_slices; c += gridDim.x) {
        const int64_t r0 = c * 512;
        const int64_t s = c * 8 + wave * 2 + half;
        const double* p = v + s * (int64_t)(W * 64) + l2;
        const int row = (wave * 2 + half) * 64 + l2;      // within the chunk
        v2d t[W];
#pragma unroll
        for (int k = 0; k < W; ++k) t[k] = NT ? __builtin_nontemporal_load((const v2d*)&p[k * 64]) : *(const v2d*)&p[k * 64];
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 7; ++g) {
            // 514 doubles per window: 16 B per thread (x windows start at arbitrary 8-B alignment)
            const double* q = x + r0 + c_run_base[g] + 2 * threadIdx.x;
            win[g][2 * threadIdx.x] = q[0];
            win[g][2 * threadIdx.x + 1] = q[1];
            if (threadIdx.x < c_run_len[g] - 1) win[g][512 + threadIdx.x] = x[r0 + c_run_base[g] + 512 + threadIdx.x];
        }
        __syncthreads();
        v2d acc = {0.0, 0.0};
#pragma unroll
        for (int g = 0; g < 7; ++g)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < c_run_len[g]) {
                    v2d u;
                    u.x = win[g][row + j];
                    u.y = win[g][row + j + 1];
                    acc += t[c_run_pos[g] + j] * u;
                }
        *(v2d*)&y[r0 + row] = acc;
    }
}

int main() {
    const int64_t n_slices = 157464;                 // 10 M rows
    const int64_t n = n_slices * W * 64, pad = 200000;
    double *v, *xb, *y;
    CHECK(hipMalloc(&v, n * 8));
    CHECK(hipMalloc(&xb, (n_slices * 64 + 2 * pad) * 8));
    CHECK(hipMalloc(&y, (n_slices * 64 + 1024) * 8));
    CHECK(hipMemset(v, 0, n * 8));
    CHECK(hipMemset(xb, 0, (n_slices * 64 + 2 * pad) * 8));
    double* x = xb + pad;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
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
        const double bytes = (double)n * 8 + (double)n_slices * 64 * 16;     // values + x once + y
        printf("  %-52s %8.1f us  %6.2f TB/s\n", name, best * 100, bytes / (best / 10 * 1e-3) / 1e12);
        return 0;
    };
    for (int grid : {512, 1024, 2048}) {
        printf("grid %d\n", grid);
        run("A  1 row/lane, 8 B loads", [&] { hipLaunchKernelGGL((k_A<false>), dim3(grid), dim3(256), 0, 0, v, x, y, n_slices); });
        run("A  non-temporal values", [&] { hipLaunchKernelGGL((k_A<true>), dim3(grid), dim3(256), 0, 0, v, x, y, n_slices); });
        run("B  2 rows/lane, 16 B loads", [&] { hipLaunchKernelGGL((k_B<false>), dim3(grid), dim3(256), 0, 0, v, x, y, n_slices); });
        run("B  non-temporal values", [&] { hipLaunchKernelGGL((k_B<true>), dim3(grid), dim3(256), 0, 0, v, x, y, n_slices); });
        run("C  x windows in LDS, 8 B value loads", [&] { hipLaunchKernelGGL((k_C<false>), dim3(grid), dim3(256), 0, 0, v, x, y, n_slices); });
        run("C  non-temporal values", [&] { hipLaunchKernelGGL((k_C<true>), dim3(grid), dim3(256), 0, 0, v, x, y, n_slices); });
        run("D  x windows in LDS, 16 B value loads", [&] { hipLaunchKernelGGL((k_D<false>), dim3(grid), dim3(256), 0, 0, v, x, y, n_slices); });
        run("D  non-temporal values", [&] { hipLaunchKernelGGL((k_D<true>), dim3(grid), dim3(256), 0, 0, v, x, y, n_slices); });
    }
    return 0;
}
