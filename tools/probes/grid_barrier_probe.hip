// What does a device-wide barrier inside a persistent kernel cost on gfx950 (8 XCDs, 256 CUs)?
// Decides whether the CG iteration can become ONE launch (SpMV phase | barrier | update phase | barrier).
// Build: hipcc --offload-arch=gfx950 -O3 grid_barrier_probe.hip -o grid_barrier_probe
// Variants: 0 = one counter (all workgroups fetch_add the same word, then poll it)
//           1 = two-level: 8 sub-counters (workgroup & 7, = its XCD) + a top word polled by everybody
//           2 = two-level with 32 sub-counters
// Also verifies that plain stores before the barrier are visible to plain loads of OTHER workgroups after it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct bar_t {
    unsigned int top;          // completed barriers
    unsigned int pad0[31];
    unsigned int sub[32 * 32]; // sub-counters, 128 B apart
};

template <int NSUB>
__device__ __forceinline__ void grid_barrier(bar_t* b, unsigned int& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        ++epoch;
        if (NSUB == 0) {
            __hip_atomic_fetch_add(&b->top, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int target = epoch * gridDim.x;
            while (__hip_atomic_load(&b->top, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        } else {
            const int g = blockIdx.x % NSUB;
            const unsigned int members = (gridDim.x - g + NSUB - 1) / NSUB;       // workgroups with blockIdx % NSUB == g
            const unsigned int old = __hip_atomic_fetch_add(&b->sub[g * 32], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == epoch * members) {                                     // last of the group
                const unsigned int t = __hip_atomic_fetch_add(&b->top, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                (void)t;
            }
            const unsigned int target = epoch * NSUB;
            while (__hip_atomic_load(&b->top, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

template <int NSUB>
__global__ void __launch_bounds__(256) k_barriers(bar_t* b, int reps, double* data, int* errors) {
    unsigned int epoch = 0;
    for (int it = 0; it < reps; ++it) {
        // every workgroup publishes a value, everybody reads the value of a workgroup on another XCD after the barrier
        if (threadIdx.x == 255) data[blockIdx.x] = (double)(it * 100000 + blockIdx.x);   // written by wave 3, read by wave 0 elsewhere
        grid_barrier<NSUB>(b, epoch);
        const int other = (blockIdx.x + 3) % gridDim.x;
        if (threadIdx.x == 0 && data[other] != (double)(it * 100000 + other)) atomicAdd(errors, 1);
        grid_barrier<NSUB>(b, epoch);
    }
}

template <int NSUB>
static int run(int grid, int reps) {
    bar_t* b; double* data; int* err;
    CHECK(hipMalloc(&b, sizeof(bar_t)));
    CHECK(hipMalloc(&data, grid * sizeof(double)));
    CHECK(hipMalloc(&err, sizeof(int)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    int herr = 0;
    for (int trial = 0; trial < 3; ++trial) {
        CHECK(hipMemset(b, 0, sizeof(bar_t)));
        CHECK(hipMemset(err, 0, sizeof(int)));
        void* args[] = {&b, &reps, &data, &err};
        CHECK(hipEventRecord(e0));
        CHECK(hipLaunchCooperativeKernel((const void*)k_barriers<NSUB>, dim3(grid), dim3(256), args, 0, nullptr));
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        int h; CHECK(hipMemcpy(&h, err, sizeof(int), hipMemcpyDeviceToHost));
        herr += h;
    }
    printf("variant NSUB=%2d grid %5d: %.3f us per barrier (%d barriers), visibility errors %d\n", NSUB, grid, best * 1e3 / (2.0 * reps), 2 * reps, herr);
    hipFree(b); hipFree(data); hipFree(err);
    return 0;
}

int main() {
    for (int grid : {256, 512, 1024, 2048}) {
        if (run<0>(grid, 500)) return 1;
        if (run<8>(grid, 500)) return 1;
        if (run<32>(grid, 500)) return 1;
    }
    return 0;
}
