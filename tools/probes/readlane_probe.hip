// What does handing a coefficient out of a lane-distributed register cost?  N x (2 v_readlane_b32 + RP v_fma_f64 with the SGPR pair as
// an operand) against N x RP v_fma_f64 with coefficients that are kernel arguments.  One wave per SIMD and four; cycles per coefficient.
//   hipcc --offload-arch=gfx950 -O3 -o readlane_probe tools/probes/readlane_probe.hip && ./readlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE, int RP>
__global__ void __launch_bounds__(256) k(double* out, const double* in, int iters, double a0, double a1, double a2, double a3) {
    const int lane = threadIdx.x & 63;
    double c = in[lane], acc[RP], xv[RP];
    for (int i = 0; i < RP; ++i) { acc[i] = 0.0; xv[i] = in[64 + lane + i]; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 64; ++p) {
            double cf;
            if (MODE == 0) {
                const int lo = __builtin_amdgcn_readlane((int)(__double_as_longlong(c) & 0xffffffffll), p);
                const int hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(c) >> 32), p);
                cf = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
            } else cf = (p & 3) == 0 ? a0 : ((p & 3) == 1 ? a1 : ((p & 3) == 2 ? a2 : a3));
#pragma unroll
            for (int i = 0; i < RP; ++i) acc[i] = fma(cf, xv[i], acc[i]);
        }
        c += 1e-9;
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
    for (int i = 0; i < RP; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0);
}
template <int MODE, int RP>
static void run(const char* name, int threads) {
    double *out, *in;
    hipMalloc(&out, ((1 << 20) + 8) * 8); hipMalloc(&in, 4096);
    hipMemset(in, 0, 4096);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, RP>), dim3(256), dim3(threads), 0, 0, out, in, iters, 1.0, 2.0, 3.0, 4.0);
    hipDeviceSynchronize();
    double cyc;
    hipMemcpy(&cyc, out + (1 << 20), 8, hipMemcpyDeviceToHost);
    printf("%-34s %d threads: %.2f clocks (s_memtime units) per coefficient (%d fma each)\n", name, threads, cyc / (iters * 64.0), RP);
    hipFree(out); hipFree(in);
}
int main() {
    run<1, 2>("SGPR coefficients (kernel args)", 256); run<0, 2>("v_readlane x 2 per coefficient", 256);
    run<1, 2>("SGPR coefficients (kernel args)", 512); run<0, 2>("v_readlane x 2 per coefficient", 512);
    run<1, 1>("SGPR coefficients (kernel args)", 512); run<0, 1>("v_readlane x 2 per coefficient", 512);
    run<1, 4>("SGPR coefficients (kernel args)", 512); run<0, 4>("v_readlane x 2 per coefficient", 512);
    return 0;
}
