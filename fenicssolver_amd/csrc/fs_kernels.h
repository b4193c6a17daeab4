// Small device helpers and reduction kernels shared by the translation units of
// libfsamd.so (each TU gets its own static copy; no relocatable device code needed).
#pragma once
#include "fs_common.h"

// ---- XCD-aware persistent chunk mapping -------------------------------------------------
// Workgroup b is (observed) placed on XCD b % 8.  Give every XCD one contiguous 1/8 of the
// chunk range so the x-vector lines gathered by neighbouring rows stay in that XCD's L2.
// Correctness never depends on the placement.
struct chunk_iter {
    int64_t cur, end, step;
};
__device__ __forceinline__ chunk_iter xcd_chunks(int64_t n_chunks) {
    const int64_t per_xcd = (n_chunks + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int64_t j = blockIdx.x >> 3;
    chunk_iter it;
    it.step = gridDim.x >> 3;
    it.cur = xcd * per_xcd + j;
    const int64_t e = (xcd + 1) * per_xcd;
    it.end = e < n_chunks ? e : n_chunks;
    return it;
}

// ---- peer-to-peer exchange between the ranks of a node (fs_comm.hip sets it up, fs_krylov.hip fuses it into the CG kernels) ----
// Memory-ordering discipline of these kernels.  A system-scope release fence on gfx950 is `buffer_wbl2` - it writes back EVERY
// dirty line of the XCD's L2, and in the middle of a CG iteration that is the 4 MB per XCD the update and the product just wrote
// (measured: + 10 us on a 20 000-row kernel).  The exchanged data therefore never becomes a dirty cached line in the first place:
//   writer: every store into a peer's buffer is a system-scope (write-through, sc0 sc1) store; EVERY wave then waits for the
//           acknowledgement of its own stores with an explicit `s_waitcnt vmcnt(0)` (fs_p2p_stores_done - inline assembly: a
//           workgroup-scope release fence compiles to `s_waitcnt lgkmcnt(0)` only on gfx950 and leaves the data stores in
//           flight; the compiler cannot drop or move the asm), workgroup barrier, then the sequence number, again a
//           system-scope store.  Where several workgroups share one sequence number each of them does the above before its
//           (relaxed, agent-scope) count: the workgroup that sees the last count publishes for all, and every other
//           workgroup's stores were acknowledged before it counted;
//   reader: spins on the sequence number with system-scope loads, then reads the data with system-scope loads (the buffers are
//           fine-grained allocations: nothing of them is held in a cache).
// a word of pinned host memory the host polls (progress / status of a batch of launches): relaxed, system scope, no fence
__device__ __forceinline__ void fs_host_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void fs_p2p_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double fs_p2p_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void fs_p2p_stores_done() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's remote stores are acknowledged
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // (and the compiler keeps later accesses behind this point)
}
__device__ __forceinline__ void fs_p2p_publish(unsigned long long* flag, unsigned long long seq) {
    __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Wait until *flag >= seq.  A wait that times out raises *err and every later wait returns at once: a solve over a broken
// mapping finishes quickly with an error instead of hanging the device.
__device__ __forceinline__ bool fs_p2p_wait(const unsigned long long* flag, unsigned long long seq, long long timeout, int* err) {
    bool ok = true;
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        const long long t0 = (long long)wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(1);
            if ((long long)wall_clock64() - t0 > timeout) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok;
}

// halo part of the exchange kernel of the peer-to-peer CG iteration (k_cg_p2p_exchange, fs_krylov.hip)
struct fs_p2p_sendrows {
    const fs_p2p_peer* peers;       // per neighbour: where its part of the send list goes
    const int32_t* send_idx;        // concatenated send lists (owned dofs)
    int64_t total_send;
    uint32_t* counter;              // workgroups of this launch whose stores are out (the last one publishes the sequence numbers)
    unsigned long long* d_seq;      // sequence number of the last executed exchange of this plan
    int nn;
    // the receive of the same exchange: this rank's flags [2][nn] and buffer [2][recv_stride], the scatter list (nullptr: ghosts in
    // arrival order)
    const unsigned long long* own_flags;
    const double* own_recv;
    const int32_t* recv_idx;
    int64_t recv_stride, total_recv, n_owned;
    long long timeout;
    int* err;
};
// all-reduce part of the same kernel
struct fs_p2p_rowsred {
    const double* partials;         // [3][npart] of the product
    double* sums_out;               // [3]: the reduced sums for the update kernel that follows
    double* const* peer_buf;
    unsigned long long* const* peer_flags;
    const double* own_buf;
    const unsigned long long* own_flags;
    unsigned long long* d_seq;      // sequence number of the last executed all-reduce
    uint32_t* counter;              // workgroups through their all-reduce part (the last one advances d_seq)
    long long timeout;
    int* err;
    int npart, nr, me;
};

// wave64 shuffle reduction -> one LDS slot per wave -> thread 0 holds the block sum.
// Fixed order, so a given launch geometry always produces the same bits.
__device__ __forceinline__ double fs_block_sum(double v, double* lds4) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds4[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) t = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
    __syncthreads();
    return t;
}

static __global__ void __launch_bounds__(FS_BLOCK) k_dot_partial(const double* __restrict__ x,
                                                                 const double* __restrict__ y, int64_t n,
                                                                 double* __restrict__ partial) {
    __shared__ double lds4[4];
    double acc = 0.0;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) acc += x[i] * y[i];
    const double t = fs_block_sum(acc, lds4);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// fixed-order sum of nsums (<= 4) interleaved partial arrays (partial[j*count + i]) by ONE workgroup of
// 1024 threads -> out[j].  Latency-bound (it sits between the SpMV and the all-reduce on N>1 GPUs), so
// all loads of a thread are issued before the first reduction step.
#define FS_SUM_BLOCK 1024
static __global__ void __launch_bounds__(FS_SUM_BLOCK) k_sum_partials(const double* __restrict__ partial, int count,
                                                                      int nsums, double* __restrict__ out) {
    __shared__ double lds[16][4];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < count; i += FS_SUM_BLOCK) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nsums) acc[j] += partial[(int64_t)j * count + i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[j] += __shfl_down(acc[j], off, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[wave][j] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < 4 && (int)threadIdx.x < nsums) {
        double t = 0.0;
        for (int w = 0; w < FS_SUM_BLOCK / 64; ++w) t += lds[w][threadIdx.x];   // fixed order
        out[threadIdx.x] = t;
    }
}
