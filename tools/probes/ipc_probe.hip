// Can two processes on this box share device memory through hipIpc (what a peer-to-peer halo exchange needs)?
// usage: ipc_probe owner /tmp/h   |   ipc_probe peer /tmp/h
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_write(double* p, unsigned long long* flag) { p[threadIdx.x] = 100.0 + threadIdx.x; __threadfence_system(); if (threadIdx.x == 0) __hip_atomic_store(flag, 7ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void k_wait(const double* p, unsigned long long* flag, double* out) {
    if (threadIdx.x == 0) { while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 7ull) __builtin_amdgcn_s_sleep(8); }
    __syncthreads();
    out[threadIdx.x] = p[threadIdx.x];
}
int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    if (!strcmp(argv[1], "owner")) {
        double* buf; unsigned long long* flag; double* out;
        CK(hipMalloc(&buf, 64 * sizeof(double))); CK(hipMalloc(&flag, 256)); CK(hipMalloc(&out, 64 * sizeof(double)));
        CK(hipMemset(buf, 0, 64 * sizeof(double))); CK(hipMemset(flag, 0, 256));
        hipIpcMemHandle_t h[2];
        CK(hipIpcGetMemHandle(&h[0], buf)); CK(hipIpcGetMemHandle(&h[1], flag));
        FILE* f = fopen(argv[2], "wb"); fwrite(h, sizeof(h), 1, f); fclose(f);
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, 0, buf, flag, out);     // spins until the peer has written
        CK(hipDeviceSynchronize());
        double host[64]; CK(hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost));
        printf("owner received %g %g ... %g\n", host[0], host[1], host[63]);
    } else {
        hipIpcMemHandle_t h[2];
        for (int t = 0; t < 200; ++t) { FILE* f = fopen(argv[2], "rb"); if (f && fread(h, sizeof(h), 1, f) == 1) { fclose(f); break; } if (f) fclose(f); usleep(50000); }
        double* buf; unsigned long long* flag;
        CK(hipIpcOpenMemHandle((void**)&buf, h[0], hipIpcMemLazyEnablePeerAccess));
        CK(hipIpcOpenMemHandle((void**)&flag, h[1], hipIpcMemLazyEnablePeerAccess));
        usleep(300000);
        hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, 0, buf, flag);
        CK(hipDeviceSynchronize());
        printf("peer wrote\n");
        CK(hipIpcCloseMemHandle(buf)); CK(hipIpcCloseMemHandle(flag));
    }
    return 0;
}
