// Does the fp64 matrix core buy anything for an fp64 contraction on gfx950?  (the element tensor of
// k_assemble_ns_wave, T = sum_q w nu g_a g_b^T, is a 30 x 30 x 14 product per cell.)  Two kernels of pure arithmetic, no memory:
//   k_fma : 16 independent chains of v_fma_f64 per lane
//   k_mfma: 8 independent chains of v_mfma_f64_16x16x4_f64 per wave (1024 fma per instruction)
// each at 1, 2, 4 waves per SIMD over the whole chip; prints TFLOP/s (2 flop per fma).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_f64_probe.hip -o mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_fma(int iters, double a, double b, double* __restrict__ out) {
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = threadIdx.x * 1e-3 + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = __builtin_fma(acc[k], a, b);
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += acc[k];
    if (s == 1.2345e300) out[0] = s;
}

__global__ void __launch_bounds__(256) k_mfma(int iters, double a, double b, double* __restrict__ out) {
    v4d acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = v4d{0.0, 0.0, 0.0, 0.0};
    const double av = a + threadIdx.x * 1e-6, bv = b - threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[k], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    if (s == 1.2345e300) out[0] = s;
}

int main() {
    double* out;
    CHECK(hipMalloc(&out, 8));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    printf("%d CUs, clock %.0f MHz\n", cus, prop.clockRate / 1e3);
    for (int wps = 1; wps <= 4; wps *= 2) {                  // waves per SIMD (4 SIMDs per CU, workgroups of 4 waves)
        const int grid = cus * wps;
        for (int which = 0; which < 2; ++which) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, iters, 1.0000001, 1e-9, out);
                else hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, iters, 1.0000001, 1e-9, out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0.f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double fma = which == 0 ? (double)grid * 256 * 16 * iters : (double)grid * 4 * 8 * 1024.0 * iters;
            printf("%-28s %d wave(s) per SIMD: %8.3f ms  %7.2f TFLOP/s\n", which == 0 ? "v_fma_f64" : "v_mfma_f64_16x16x4_f64", wps, best,
                   2.0 * fma / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
