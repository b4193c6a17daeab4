// Multi-GPU communication of libfsamd.so: one process per GPU, RCCL over xGMI.
//
// Replaces the MPI traffic PETSc/DOLFIN generate under `mpirun`
// (FenicsSolver/SolverBase.py:102-118, 634):
//   VecDot/VecNorm  -> MPI_Allreduce   =>  ncclAllReduce of 3 doubles per CG iteration
//   VecScatter ghost update            =>  grouped ncclSend/ncclRecv with the z-slab
//                                          neighbours (<= 2 for slab partitions, each
//                                          on its own xGMI link) before every SpMV
// Both are latency-bound (24 B and tens of KB), so they are issued in-stream on the
// compute stream: no host synchronisation anywhere in the CG loop.
//
// librccl is dlopen()ed on first use, so single-GPU processes never depend on it.
#include "fs_common.h"
#include "fs_kernels.h"
#include <vector>
#include <algorithm>
#include <dlfcn.h>
#include <stdlib.h>
#include <rccl/rccl.h>
#include <chrono>

struct rccl_api {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static rccl_api g_nccl;

static int rccl_load() {
    if (g_nccl.handle) return FS_OK;
    // Prefer the RCCL of the ROCm install this library's HIP runtime comes from (a host process may carry another
    // copy under the same SONAME).  FS_RCCL_PATH names a specific library (a site build of RCCL; the test suite
    // points it at tests/shim/libfakerccl.so to run several ranks on one GPU).
    const char* names[] = {getenv("FS_RCCL_PATH"), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* nm : names) {
        if (!nm || !*nm) continue;
        h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) {
        fs_set_error("cannot load librccl: %s", dlerror());
        return FS_ERR_COMM;
    }
#define LOAD(field, sym)                                              \
    g_nccl.field = (decltype(g_nccl.field))dlsym(h, sym);             \
    if (!g_nccl.field) {                                              \
        fs_set_error("librccl lacks symbol %s", sym);                 \
        return FS_ERR_COMM;                                           \
    }
    LOAD(GetUniqueId, "ncclGetUniqueId")
    LOAD(CommInitRank, "ncclCommInitRank")
    LOAD(CommDestroy, "ncclCommDestroy")
    LOAD(AllReduce, "ncclAllReduce")
    LOAD(AllGather, "ncclAllGather")
    LOAD(Send, "ncclSend")
    LOAD(Recv, "ncclRecv")
    LOAD(GroupStart, "ncclGroupStart")
    LOAD(GroupEnd, "ncclGroupEnd")
    LOAD(GetErrorString, "ncclGetErrorString")
#undef LOAD
    g_nccl.handle = h;
    return FS_OK;
}

#define FS_NCCL(call)                                                                          \
    do {                                                                                       \
        ncclResult_t r__ = (call);                                                             \
        if (r__ != ncclSuccess) {                                                              \
            fs_set_error("%s failed: %s (%s:%d)", #call, g_nccl.GetErrorString(r__), __FILE__, __LINE__); \
            return FS_ERR_COMM;                                                                \
        }                                                                                      \
    } while (0)

static_assert(sizeof(ncclUniqueId) == FS_UNIQUE_ID_BYTES, "ncclUniqueId size");

extern "C" int fs_comm_get_unique_id(char id[FS_UNIQUE_ID_BYTES]) {
    FS_CHECK(fs_require_init());
    FS_CHECK(rccl_load());
    ncclUniqueId uid;
    FS_NCCL(g_nccl.GetUniqueId(&uid));
    memcpy(id, &uid, FS_UNIQUE_ID_BYTES);
    return FS_OK;
}

extern "C" int fs_comm_init(int n_ranks, int rank, const char id[FS_UNIQUE_ID_BYTES]) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks && id, "fs_comm_init: bad rank %d of %d", rank, n_ranks);
    fs_runtime& rt = fs_rt();
    FS_REQUIRE(rt.comm == nullptr, "fs_comm_init: communicator already initialised");
    FS_CHECK(rccl_load());
    ncclUniqueId uid;
    memcpy(&uid, id, FS_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    FS_NCCL(g_nccl.CommInitRank(&comm, n_ranks, uid, rank));
    rt.comm = (void*)comm;
    rt.n_ranks = n_ranks;
    rt.rank = rank;
    return FS_OK;
}

extern "C" int fs_comm_info(int* n_ranks, int* rank) {
    if (n_ranks) *n_ranks = fs_rt().n_ranks;
    if (rank) *rank = fs_rt().rank;
    return FS_OK;
}

extern "C" int fs_comm_finalize(void) {
    fs_runtime& rt = fs_rt();
    fs_p2p_reduce_teardown();
    if (rt.comm) {
        (void)hipStreamSynchronize(rt.stream);
        FS_NCCL(g_nccl.CommDestroy((ncclComm_t)rt.comm));
        rt.comm = nullptr;
    }
    rt.n_ranks = 1;
    rt.rank = 0;
    return FS_OK;
}

// host time spent inside the RCCL enqueue calls (FS_COMM_TIMING=1 prints it per solve: the distributed CG loop is bound by it
// when the kernels of an iteration are shorter than the enqueue of its collectives)
static double g_host_us[2] = {0.0, 0.0};
static long g_host_calls[2] = {0, 0};
struct host_timer {
    int k; std::chrono::steady_clock::time_point t0;
    explicit host_timer(int kind) : k(kind), t0(std::chrono::steady_clock::now()) {}
    ~host_timer() { g_host_us[k] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); ++g_host_calls[k]; }
};
void fs_comm_host_time(double* allreduce_us, long* allreduce_calls, double* halo_us, long* halo_calls, bool reset) {
    if (allreduce_us) *allreduce_us = g_host_us[0];
    if (allreduce_calls) *allreduce_calls = g_host_calls[0];
    if (halo_us) *halo_us = g_host_us[1];
    if (halo_calls) *halo_calls = g_host_calls[1];
    if (reset) { g_host_us[0] = g_host_us[1] = 0.0; g_host_calls[0] = g_host_calls[1] = 0; }
}

int fs_comm_allreduce_dev(double* d_inout, int n, hipStream_t s) {
    fs_runtime& rt = fs_rt();
    if (!rt.comm) return FS_OK;  // one rank
    if (fs_p2p_reduce_enabled() && n <= 8) return fs_p2p_allreduce_dev(nullptr, 0, d_inout, n, s);
    host_timer timer(0);
    FS_NCCL(g_nccl.AllReduce(d_inout, d_inout, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)rt.comm, s));
    return FS_OK;
}

int fs_comm_sum_allreduce_dev(const double* partials, int npart, int nv, double* out, hipStream_t s) {
    if (fs_rt().comm && fs_p2p_reduce_enabled() && nv <= 4) return fs_p2p_allreduce_dev(partials, npart, out, nv, s);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(FS_SUM_BLOCK), 0, s, partials, npart, nv, out);
    FS_KERNEL_CHECK();
    return fs_comm_allreduce_dev(out, nv, s);
}

extern "C" int fs_comm_allreduce_sum(double* host_inout, int n) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(host_inout && n > 0, "fs_comm_allreduce_sum: bad arguments");
    if (!fs_rt().comm) return FS_OK;
    hipStream_t s = fs_rt().stream;
    dbuf<double> d;
    FS_CHECK(d.alloc(n));
    FS_CHECK(d.upload(host_inout, n, s));
    FS_CHECK(fs_comm_allreduce_dev(d.p, n, s));
    FS_CHECK(d.download(host_inout, n, s));
    return FS_OK;
}

// Every rank contributes n_max doubles (its n_send values, padded); recv = [n_ranks][n_max] on every rank.  Used by the
// solver API to gather the owned parts of a solution (fenicssolver_amd/parallel.py) - one ncclAllGather.
extern "C" int fs_comm_allgather(const double* host_send, int64_t n_send, int64_t n_max, double* host_recv) {
    FS_CHECK(fs_require_init());
    fs_runtime& rt = fs_rt();
    FS_REQUIRE(host_recv && n_send >= 0 && n_send <= n_max && (n_send == 0 || host_send), "fs_comm_allgather: bad arguments");
    if (!rt.comm) {
        memcpy(host_recv, host_send, (size_t)n_send * sizeof(double));
        return FS_OK;
    }
    const int nr = rt.n_ranks;
    hipStream_t s = rt.stream;
    dbuf<double> ds, dr;
    FS_CHECK(ds.alloc(std::max<int64_t>(n_max, 1)));
    FS_CHECK(dr.alloc(std::max<int64_t>(n_max, 1) * nr));
    FS_CHECK(ds.zero(s));
    FS_CHECK(ds.upload(host_send, n_send, s));
    FS_NCCL(g_nccl.AllGather(ds.p, dr.p, (size_t)n_max, ncclDouble, (ncclComm_t)rt.comm, s));
    FS_CHECK(dr.download(host_recv, n_max * nr, s));
    return FS_OK;
}

// ---- halo ----------------------------------------------------------------------------------
__global__ void k_pack(const double* __restrict__ v, const int32_t* __restrict__ idx, int64_t n,
                       double* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = v[idx[i]];
}

__global__ void k_unpack(double* __restrict__ v, const int32_t* __restrict__ idx, int64_t n, const double* __restrict__ in) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) v[idx[i]] = in[i];
}

static int set_halo_impl(fs_space_t space, int n_neighbors, const int32_t* neighbor_ranks, const int64_t* send_counts,
                         const int32_t* send_idx, const int64_t* recv_counts, const int32_t* recv_idx);

// ---- peer-to-peer ghost refresh and all-reduce ----------------------------------------------------
// Nothing but kernels of the compute stream: no library call, no second stream, no event.  (Measured on MI355X: the
// grouped ncclSend / ncclRecv of two 80 KB planes on the communication stream costs 31 us per refresh, of which most is
// the two cross-stream event hops - the same two kernels below behind the same events took 36 us.)
//
// Send: `groups` workgroups per neighbour gather the values that neighbour needs and store them straight into ITS receive
// buffer (mapped through hipIpc, fine-grained memory: the stores travel over xGMI), the last one to finish publishes the
// sequence number of the exchange with a system-scope release.  Fire and forget: the rows that need no ghost value are
// multiplied behind it while the data is in flight.  Two slots alternate: a rank cannot start exchange q+2 before its
// neighbour consumed exchange q, because its own receive of q+1 completes only after the neighbour sent q+1, which the
// neighbour's stream orders after its receive of q.
static int* g_p2p_err = nullptr;                 // device flag: a wait timed out (peer died / mapping not coherent)
static long long g_p2p_timeout_ticks = 0;        // wall_clock64 ticks (100 MHz)

// Sequence numbers live ON THE DEVICE (one counter per halo plan, one for the all-reduce) and advance only when an exchange is
// really executed: kernels gated off by a solver's status word consume none, so consecutive executed exchanges always alternate
// between the two slots - which is what the no-overwrite argument above rests on - and a captured hipGraph can replay them.
// Every workgroup reads the counter when it starts; the LAST workgroup through (an atomic count) writes it back incremented.
__global__ void __launch_bounds__(FS_BLOCK) k_p2p_send(const fs_p2p_peer* __restrict__ peers, int groups, uint32_t* done, uint32_t* all_done,
                                                       unsigned long long* d_seq, const double* __restrict__ vec,
                                                       const int32_t* __restrict__ send_idx) {
    const unsigned long long seq = *d_seq + 1ull;
    const int slot = (int)(seq & 1ull);
    const int nb = blockIdx.x / groups, g = blockIdx.x - nb * groups;
    const fs_p2p_peer p = peers[nb];
    double* dst = p.recv + (int64_t)slot * p.peer_total + p.recv_offset;
    const int32_t* idx = send_idx + p.send_offset;
    for (int64_t k = (int64_t)g * FS_BLOCK + threadIdx.x; k < p.send_count; k += (int64_t)groups * FS_BLOCK) fs_p2p_store(dst + k, vec[idx[k]]);
    fs_p2p_stores_done();
    __syncthreads();
    if (threadIdx.x == 0) {
        bool last = true;
        if (groups > 1) {
            last = atomicAdd(done + nb, 1u) == (uint32_t)(groups - 1);
            if (last) __hip_atomic_store(done + nb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (last) fs_p2p_publish(p.flags + (int64_t)slot * p.peer_nn + p.peer_slot, seq);
        if (atomicAdd(all_done, 1u) == gridDim.x - 1) {
            __hip_atomic_store(all_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(d_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Receive of the exchange the last send kernel issued (*d_seq): wait until every neighbour delivered it, then move the values
// to the ghost entries.  `never`: test hook (FS_P2P_TEST=lost).
__global__ void __launch_bounds__(FS_BLOCK) k_p2p_recv(int nn, const unsigned long long* flags_base, const unsigned long long* d_seq,
                                                       unsigned long long never, const double* recv_base, int64_t slot_stride,
                                                       int64_t total, const int32_t* __restrict__ recv_idx,
                                                       double* __restrict__ vec, int64_t n_owned, long long timeout, int* err) {
    const unsigned long long seq = *d_seq;
    const int slot = (int)(seq & 1ull);
    if ((int)threadIdx.x < nn) fs_p2p_wait(flags_base + (int64_t)slot * nn + threadIdx.x, seq + never, timeout, err);
    __syncthreads();
    const double* recv = recv_base + (int64_t)slot * slot_stride;
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < total; k += stride) {
        const double v = fs_p2p_load(recv + k);
        vec[recv_idx ? (int64_t)recv_idx[k] : n_owned + k] = v;
    }
}

// All-reduce (sum) of up to 8 doubles: every rank stores its values into slot [me] of every rank's buffer and releases a
// sequence number there, waits for the sequence numbers of all ranks in its own buffer and adds the values up IN RANK ORDER -
// every rank obtains bit-identical sums (the convergence decision taken from them is the same everywhere).  With
// npart > 0 the values are first summed from the per-workgroup partials [nv][npart] (fixed order), which saves the separate
// launch.  The two-slot argument of the halo holds here too.
struct fs_p2p_reduce {
    bool enabled = false;
    double* buf = nullptr;                     // fine-grained [2][n_ranks][8]
    unsigned long long* flags = nullptr;       // fine-grained [2][n_ranks]
    std::vector<void*> opened;
    dbuf<double*> peer_buf;
    dbuf<unsigned long long*> peer_flags;
    dbuf<unsigned long long> d_seq;            // device-side sequence number of the last executed all-reduce
    dbuf<uint32_t> counter;                    // workgroups of the exchange kernel through their all-reduce part
    void release() {
        if (buf || flags || !opened.empty()) (void)hipDeviceSynchronize();
        for (void* q : opened) (void)hipIpcCloseMemHandle(q);
        opened.clear();
        if (buf) (void)hipFree(buf);
        if (flags) (void)hipFree(flags);
        buf = nullptr; flags = nullptr;
        peer_buf.release(); peer_flags.release();
        d_seq.release(); counter.release();
        enabled = false;
    }
};
static fs_p2p_reduce g_p2p_red;
static int g_p2p_spaces = 0;           // spaces whose exchange was turned on (and not off again) through the API

__global__ void __launch_bounds__(FS_SUM_BLOCK) k_p2p_allreduce(int nr, int me, int nv, const double* __restrict__ partials, int npart,
                                                                double* const* __restrict__ peer_buf,
                                                                unsigned long long* const* __restrict__ peer_flags,
                                                                const double* own_buf, const unsigned long long* own_flags,
                                                                unsigned long long* d_seq, double* inout,
                                                                long long timeout, int* err) {
    const unsigned long long seq = *d_seq + 1ull;
    const int slot = (int)(seq & 1ull);
    __shared__ double lds[FS_SUM_BLOCK / 64][4];
    __shared__ double mine[8];
    if (npart > 0) {                                     // the reduction of k_sum_partials (fs_kernels.h), same order, same bits
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int i = threadIdx.x; i < npart; i += FS_SUM_BLOCK) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nv) acc[j] += partials[(int64_t)j * npart + i];
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
        if ((int)threadIdx.x < nv) {
            double t = 0.0;
            for (int w = 0; w < FS_SUM_BLOCK / 64; ++w) t += lds[w][threadIdx.x];
            mine[threadIdx.x] = t;
        }
    } else if ((int)threadIdx.x < nv) mine[threadIdx.x] = inout[threadIdx.x];
    __syncthreads();
    const int t = threadIdx.x;
    if (t < nr) {
        double* dst = peer_buf[t] + ((int64_t)slot * nr + me) * 8;
        for (int v = 0; v < nv; ++v) fs_p2p_store(dst + v, mine[v]);
        fs_p2p_stores_done();
        fs_p2p_publish(peer_flags[t] + (int64_t)slot * nr + me, seq);
        fs_p2p_wait(own_flags + (int64_t)slot * nr + t, seq, timeout, err);
    }
    __syncthreads();
    if (t < nv) {
        double a = 0.0;
        for (int r = 0; r < nr; ++r) a += fs_p2p_load(own_buf + ((int64_t)slot * nr + r) * 8 + t);
        inout[t] = a;
    }
    if (t == 0) *d_seq = seq;              // (every thread read it before the first barrier)
}

// Also the destructor's path (fs_space_destroy with the exchange still on).  That is safe only when every rank is past its last
// exchange with this one - true after any solve returned, whose closing residual check is a collective behind the last exchange -
// and it is documented as a requirement in include/fenicssolver_amd.h: turn the exchange off (a collective) or destroy the
// space on all ranks at the same point of the program.  The count of live peer-to-peer spaces follows, so that the all-reduce
// does not stay on the peer-to-peer path for nobody (its buffers are freed by the next collective enable / disable or finalize).
void fs_p2p_halo::release() {
    if (enabled && g_p2p_spaces > 0) --g_p2p_spaces;
    if (recv || flags || !opened.empty()) (void)hipDeviceSynchronize();
    for (void* q : opened) (void)hipIpcCloseMemHandle(q);
    opened.clear();
    if (recv) (void)hipFree(recv);
    if (flags) (void)hipFree(flags);
    recv = nullptr; flags = nullptr;
    peers.release();
    done.release();
    counter.release();
    d_seq.release();
    pending = nullptr;
    enabled = false;
}

static int p2p_barrier() {               // host-level: an all-gather of one double
    fs_runtime& rt = fs_rt();
    if (!rt.comm) return FS_OK;
    double one = 1.0;
    std::vector<double> all((size_t)rt.n_ranks);
    return fs_comm_allgather(&one, 1, 1, all.data());
}

static constexpr int P2P_HANDLE_DOUBLES = (int)(sizeof(hipIpcMemHandle_t) / sizeof(double));

// every rank's verdict on a collective set-up step: all succeed or all back out (a rank waiting for stores that a failed
// neighbour will never issue would hang the device)
static int p2p_agree(bool ok, const char* what, const char* why = "") {
    fs_runtime& rt = fs_rt();
    double mine = ok ? 0.0 : 1.0;
    std::vector<double> all((size_t)rt.n_ranks, 0.0);
    FS_CHECK(fs_comm_allgather(&mine, 1, 1, all.data()));
    for (int r = 0; r < rt.n_ranks; ++r)
        if (all[(size_t)r] != 0.0) {
            if (ok || !*why) fs_set_error("%s: rank %d could not map its neighbours' buffers (hipIpc: one node, peer access)", what, r);
            else fs_set_error("%s: %s", what, why);
            return FS_ERR_COMM;
        }
    return FS_OK;
}

// FS_P2P_TEST (tests only): "openfail" - rank 0 pretends hipIpcOpenMemHandle failed; "lost" - receives wait for a sequence
// number that never comes; "late:K" - the same from the K-th receive on (a transport that breaks after it was chosen).  All
// failure paths must end in an error every rank agrees on, never in a hang.
static int g_p2p_test_late = -1;
static int p2p_test_mode() {
    const char* e = getenv("FS_P2P_TEST");
    if (!e) return 0;
    if (!strncmp(e, "late:", 5)) { g_p2p_test_late = atoi(e + 5); return 3; }
    return !strcmp(e, "openfail") ? 1 : (!strcmp(e, "lost") ? 2 : 0);
}

static int p2p_common_setup() {
    if (!g_p2p_err) {
        if (hipMalloc((void**)&g_p2p_err, sizeof(int)) != hipSuccess || hipMemset(g_p2p_err, 0, sizeof(int)) != hipSuccess) {
            (void)hipGetLastError();
            g_p2p_err = nullptr;
            return FS_ERR_HIP;         // (carried to the next agreement by the callers, not returned before a collective)
        }
    }
    const char* e = getenv("FS_P2P_TIMEOUT_MS");
    const double ms = e ? atof(e) : 10000.0;
    g_p2p_timeout_ticks = (long long)(ms * 1e5);          // wall_clock64: 100 MHz
    return FS_OK;
}

// The all-reduce side, set up once per communicator (collective).
static int p2p_reduce_setup() {
    fs_runtime& rt = fs_rt();
    fs_p2p_reduce& R = g_p2p_red;
    if (R.enabled) return FS_OK;
    const int nr = rt.n_ranks;
    FS_REQUIRE(nr <= 64, "peer-to-peer all-reduce: %d ranks, at most 64", nr);
    // (a step that fails on ONE rank - an allocation, hipIpcGetMemHandle without dmabuf IPC - must not make that rank leave before
    // the collective the others are heading for: every rank reaches the all-gather and the agreement, whatever happened locally)
    bool ok = hipExtMallocWithFlags((void**)&R.buf, (size_t)(2 * nr * 8) * sizeof(double), hipDeviceMallocFinegrained) == hipSuccess &&
              hipExtMallocWithFlags((void**)&R.flags, (size_t)(2 * nr) * sizeof(unsigned long long), hipDeviceMallocFinegrained) == hipSuccess &&
              hipMemset(R.buf, 0, (size_t)(2 * nr * 8) * sizeof(double)) == hipSuccess &&
              hipMemset(R.flags, 0, (size_t)(2 * nr) * sizeof(unsigned long long)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    hipIpcMemHandle_t hv[2];
    memset(hv, 0, sizeof(hv));
    ok = ok && hipIpcGetMemHandle(&hv[0], R.buf) == hipSuccess && hipIpcGetMemHandle(&hv[1], R.flags) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    const int rec_n = 2 * P2P_HANDLE_DOUBLES;
    std::vector<double> rec((size_t)rec_n), all((size_t)rec_n * nr);
    memcpy(rec.data(), hv, sizeof(hv));
    FS_CHECK(fs_comm_allgather(rec.data(), rec_n, rec_n, all.data()));
    std::vector<double*> pb((size_t)nr);
    std::vector<unsigned long long*> pf((size_t)nr);
    if (p2p_agree(ok, "peer-to-peer all-reduce", "the buffers could not be allocated or exported (hipIpcGetMemHandle)") != FS_OK) {
        R.release();
        return FS_ERR_COMM;
    }
    for (int r = 0; r < nr && ok; ++r) {
        if (r == rt.rank) { pb[(size_t)r] = R.buf; pf[(size_t)r] = R.flags; continue; }
        hipIpcMemHandle_t qh[2];
        memcpy(qh, all.data() + (size_t)r * rec_n, sizeof(qh));
        void *a = nullptr, *b = nullptr;
        ok = hipIpcOpenMemHandle(&a, qh[0], hipIpcMemLazyEnablePeerAccess) == hipSuccess;
        if (ok) { R.opened.push_back(a); ok = hipIpcOpenMemHandle(&b, qh[1], hipIpcMemLazyEnablePeerAccess) == hipSuccess; }
        if (ok) R.opened.push_back(b);
        pb[(size_t)r] = (double*)a; pf[(size_t)r] = (unsigned long long*)b;
    }
    if (!ok) (void)hipGetLastError();
    if (p2p_test_mode() == 1 && rt.rank == 0) ok = false;
    // (the device-side tables too BEFORE the last agreement: a local out-of-memory here must back every rank out, not leave the
    // others waiting for this one in the self-test all-reduce)
    ok = ok && R.peer_buf.alloc(nr) == FS_OK && R.peer_flags.alloc(nr) == FS_OK &&
         R.peer_buf.upload(pb.data(), nr, rt.stream) == FS_OK && R.peer_flags.upload(pf.data(), nr, rt.stream) == FS_OK &&
         R.d_seq.alloc(1) == FS_OK && R.d_seq.zero(rt.stream) == FS_OK && R.counter.alloc(1) == FS_OK && R.counter.zero(rt.stream) == FS_OK &&
         hipStreamSynchronize(rt.stream) == hipSuccess;
    const int rc = p2p_agree(ok, "peer-to-peer all-reduce");
    if (rc != FS_OK) { R.release(); return rc; }
    R.enabled = true;
    return FS_OK;
}

void fs_p2p_reduce_teardown() {          // fs_comm_finalize
    g_p2p_spaces = 0;
    g_p2p_red.release();
}

// 1 if the all-reduces of this process go through the peer-to-peer kernel (the solver then keeps them in the compute stream)
int fs_p2p_reduce_enabled() { return g_p2p_red.enabled ? 1 : 0; }

// sums of the per-workgroup partials [nv][npart] (npart > 0) or of inout itself (npart = 0) over all ranks -> inout
int fs_p2p_allreduce_dev(const double* partials, int npart, double* inout, int nv, hipStream_t s) {
    fs_runtime& rt = fs_rt();
    fs_p2p_reduce& R = g_p2p_red;
    hipLaunchKernelGGL(k_p2p_allreduce, dim3(1), dim3(FS_SUM_BLOCK), 0, s, rt.n_ranks, rt.rank, nv, partials, npart,
                       (double* const*)R.peer_buf.p, (unsigned long long* const*)R.peer_flags.p, R.buf, R.flags, R.d_seq.p,
                       inout, g_p2p_timeout_ticks, g_p2p_err);
    FS_KERNEL_CHECK();
    return FS_OK;
}

// a wait of a peer-to-peer kernel timed out since the last call?  (synchronises the stream; the solvers ask once per solve)
int fs_p2p_check(hipStream_t s) {
    if (!g_p2p_err) return FS_OK;
    int e = 0;
    FS_HIP(hipMemcpyAsync(&e, g_p2p_err, sizeof(int), hipMemcpyDeviceToHost, s));
    FS_HIP(hipStreamSynchronize(s));
    if (e) {
        FS_HIP(hipMemsetAsync(g_p2p_err, 0, sizeof(int), s));
        fs_set_error("peer-to-peer exchange: a neighbour's data did not arrive within FS_P2P_TIMEOUT_MS (peer process gone, or stores over hipIpc mappings not visible on this system) - results of this solve are invalid");
        return FS_ERR_P2P_TIMEOUT;
    }
    return FS_OK;
}

// Collective over the communicator (every rank calls it for its space, in the same order): allocate the receive side,
// publish its hipIpc handles and the neighbour table with one all-gather, map every neighbour's buffers; the all-reduce
// side is set up with the first space.  One node only (hipIpc); the RCCL exchange stays the default.  enable = 0 tears the
// space's side down (also collective: nobody frees memory a neighbour may still write).
static constexpr int P2P_MAX_NB = 16;
static constexpr int P2P_REC = 4 + 3 * P2P_MAX_NB + 2 * P2P_HANDLE_DOUBLES;
extern "C" int fs_space_enable_p2p_halo(fs_space_t space, int enable) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(space, "fs_space_enable_p2p_halo: null space");
    static_assert(sizeof(hipIpcMemHandle_t) % sizeof(double) == 0, "handle size");
    fs_runtime& rt = fs_rt();
    fs_halo_plan& h = space->halo;
    hipStream_t s = rt.stream;
    FS_HIP(hipStreamSynchronize(s));
    if (h.comm_stream) FS_HIP(hipStreamSynchronize(h.comm_stream));
    // (unconditional: every rank is in this call, whatever the state of ITS side - a barrier that depended on local state would
    // leave the ranks that still hold buffers waiting for those that do not)
    FS_CHECK(p2p_barrier());
    if (h.p2p.enabled || h.p2p.recv) {
        h.p2p.release();
        h.early = -1;
    }
    if (!enable) {
        // the last space turned off by hand takes the all-reduce back to RCCL (every rank is here: nobody writes into the buffers
        // being freed); spaces destroyed with the exchange on leave it as it is
        if (g_p2p_spaces == 0 && g_p2p_red.enabled) g_p2p_red.release();
        return FS_OK;
    }
    if (!rt.comm) {
        fs_set_error("fs_space_enable_p2p_halo: no communicator is up (call fs_comm_init)");
        return FS_ERR_COMM;
    }
    // (local failures - an allocation, too many neighbours - are carried to the agreement below, never returned before the
    // collectives the other ranks are heading for: see p2p_reduce_setup, whose own verdict is agreed and alike on every rank)
    char why[256] = "";
    const bool common_ok = p2p_common_setup() == FS_OK;
    FS_CHECK(p2p_reduce_setup());
    const int nn_plan = h.active ? (int)h.neighbors.size() : 0;
    const int nn = std::min(nn_plan, P2P_MAX_NB);
    fs_p2p_halo& pp = h.p2p;
    const int64_t total = std::max<int64_t>(h.total_recv, 1);
    bool ok = common_ok && nn_plan <= P2P_MAX_NB;
    if (!common_ok) snprintf(why, sizeof(why), "the error flag of the exchange kernels could not be allocated");
    else if (!ok) snprintf(why, sizeof(why), "%d neighbours, at most %d", nn_plan, P2P_MAX_NB);
    ok = ok && hipExtMallocWithFlags((void**)&pp.recv, (size_t)(2 * total) * sizeof(double), hipDeviceMallocFinegrained) == hipSuccess &&
              hipExtMallocWithFlags((void**)&pp.flags, (size_t)(2 * std::max(nn, 1)) * sizeof(unsigned long long), hipDeviceMallocFinegrained) == hipSuccess &&
              hipMemset(pp.flags, 0, (size_t)(2 * std::max(nn, 1)) * sizeof(unsigned long long)) == hipSuccess &&
              hipDeviceSynchronize() == hipSuccess;
    hipIpcMemHandle_t hv[2];
    memset(hv, 0, sizeof(hv));
    ok = ok && hipIpcGetMemHandle(&hv[0], pp.recv) == hipSuccess && hipIpcGetMemHandle(&hv[1], pp.flags) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); if (!why[0]) snprintf(why, sizeof(why), "the receive buffers could not be allocated or exported (hipIpcGetMemHandle)"); }
    std::vector<double> rec((size_t)P2P_REC, 0.0), all((size_t)P2P_REC * rt.n_ranks, 0.0);
    rec[0] = nn; rec[1] = (double)total;
    for (int i = 0; i < nn; ++i) {
        rec[(size_t)(4 + 3 * i)] = h.neighbors[(size_t)i];
        rec[(size_t)(5 + 3 * i)] = (double)h.recv_offsets[(size_t)i];
        rec[(size_t)(6 + 3 * i)] = (double)h.recv_counts[(size_t)i];
    }
    memcpy(rec.data() + 4 + 3 * P2P_MAX_NB, hv, sizeof(hv));
    FS_CHECK(fs_comm_allgather(rec.data(), P2P_REC, P2P_REC, all.data()));
    std::vector<fs_p2p_peer> peers((size_t)std::max(nn, 1));
    std::vector<std::pair<double*, unsigned long long*>> mapped((size_t)rt.n_ranks, {nullptr, nullptr});
    int64_t max_send = 0;
    for (int i = 0; i < nn && ok; ++i) {
        const int q = h.neighbors[(size_t)i];
        int occ = 0;
        for (int j = 0; j < i; ++j) occ += h.neighbors[(size_t)j] == q;
        const double* qr = q >= 0 && q < rt.n_ranks ? all.data() + (size_t)q * P2P_REC : nullptr;
        const int qnn = qr ? (int)qr[0] : 0;
        int slot = -1;
        for (int j = 0, seen = 0; j < qnn; ++j)
            if ((int)qr[4 + 3 * j] == rt.rank && seen++ == occ) { slot = j; break; }
        if (slot < 0) { ok = false; snprintf(why, sizeof(why), "rank %d does not list rank %d as a neighbour", q, rt.rank); break; }
        if ((int64_t)qr[6 + 3 * slot] != h.send_counts[(size_t)i]) {
            ok = false;
            snprintf(why, sizeof(why), "rank %d expects %lld values from rank %d, which sends %lld", q, (long long)qr[6 + 3 * slot], rt.rank,
                     (long long)h.send_counts[(size_t)i]);
            break;
        }
        if (q == rt.rank) mapped[(size_t)q] = {pp.recv, pp.flags};
        else if (!mapped[(size_t)q].first) {
            hipIpcMemHandle_t qh[2];
            memcpy(qh, qr + 4 + 3 * P2P_MAX_NB, sizeof(qh));
            void *a = nullptr, *b = nullptr;
            ok = hipIpcOpenMemHandle(&a, qh[0], hipIpcMemLazyEnablePeerAccess) == hipSuccess;
            if (ok) { pp.opened.push_back(a); ok = hipIpcOpenMemHandle(&b, qh[1], hipIpcMemLazyEnablePeerAccess) == hipSuccess; }
            if (ok) pp.opened.push_back(b);
            if (!ok) { (void)hipGetLastError(); snprintf(why, sizeof(why), "hipIpcOpenMemHandle of rank %d's buffers failed", q); break; }
            mapped[(size_t)q] = {(double*)a, (unsigned long long*)b};
        }
        fs_p2p_peer& e = peers[(size_t)i];
        e.recv = mapped[(size_t)q].first;
        e.flags = mapped[(size_t)q].second;
        e.recv_offset = (int64_t)qr[5 + 3 * slot];
        e.peer_total = (int64_t)qr[1];
        e.send_offset = h.send_offsets[(size_t)i];
        e.send_count = h.send_counts[(size_t)i];
        e.send_first = h.send_contiguous[(size_t)i] ? h.send_first[(size_t)i] : -1;
        e.peer_slot = slot;
        e.peer_nn = std::max(qnn, 1);
        max_send = std::max(max_send, e.send_count);
    }
    if (ok) {
        ok = pp.peers.alloc((int64_t)peers.size()) == FS_OK && pp.peers.upload(peers.data(), (int64_t)peers.size(), s) == FS_OK &&
             pp.done.alloc((int64_t)peers.size()) == FS_OK && pp.done.zero(s) == FS_OK && pp.counter.alloc(2) == FS_OK && pp.counter.zero(s) == FS_OK &&
             pp.d_seq.alloc(1) == FS_OK && pp.d_seq.zero(s) == FS_OK && hipStreamSynchronize(s) == hipSuccess;
        if (!ok) snprintf(why, sizeof(why), "the device tables of the exchange could not be allocated");
    }
    const int rc = p2p_agree(ok, "fs_space_enable_p2p_halo", why);
    if (rc != FS_OK) {
        pp.release();
        if (g_p2p_spaces == 0) g_p2p_red.release();
        return rc;
    }
    pp.send_groups = (int)std::min<int64_t>(16, std::max<int64_t>(1, (max_send + 8 * FS_BLOCK - 1) / (8 * FS_BLOCK)));
    pp.pending = nullptr;
    pp.enabled = true;
    static uint64_t generation = 0;
    pp.generation = ++generation;
    ++g_p2p_spaces;
    h.early = -1;              // the solver asks the ranks again which iteration they can all run
    // Self-test on the hardware at hand, before any solver trusts the transport: an all-reduce of (rank + 1) and a ghost refresh of
    // a vector holding (rank + 1) everywhere - every ghost must then read (its owner's rank + 1) - with a short time-out.  The
    // verdict is agreed over the library's all-gather: a mapping that opens but does not deliver backs out on every rank.
    {
        const long long keep_timeout = g_p2p_timeout_ticks;
        g_p2p_timeout_ticks = 50000000ll;       // 0.5 s
        bool good = true;
        const int64_t nl = space->n_dofs_local, no = space->n_dofs_owned;
        dbuf<double> v;
        std::vector<double> hv2((size_t)std::max<int64_t>(nl, 1) + 2, (double)(rt.rank + 1));
        int rc2 = v.alloc(nl + 2);
        if (rc2 == FS_OK) rc2 = v.upload(hv2.data(), nl + 2, s);
        if (rc2 == FS_OK) rc2 = fs_p2p_allreduce_dev(nullptr, 0, v.p + nl, 1, s);       // (the spare entry behind the local dofs)
        if (rc2 == FS_OK && nn > 0) rc2 = fs_halo_exchange_dev(space, v.p, s);
        if (rc2 == FS_OK) rc2 = v.download(hv2.data(), nl + 2, s);
        if (rc2 == FS_OK) rc2 = fs_p2p_check(s);
        good = rc2 == FS_OK && hv2[(size_t)nl] == 0.5 * rt.n_ranks * (rt.n_ranks + 1.0);
        if (good && nn > 0) {
            std::vector<int32_t> ridx;
            if (h.recv_idx.p) { ridx.resize((size_t)h.total_recv); good = h.recv_idx.download(ridx.data(), h.total_recv, s) == FS_OK; }
            for (int i = 0; i < nn && good; ++i)
                for (int64_t k2 = 0; k2 < h.recv_counts[(size_t)i] && good; ++k2) {
                    const int64_t pos = h.recv_offsets[(size_t)i] + k2;
                    const int64_t dof = ridx.empty() ? no + pos : (int64_t)ridx[(size_t)pos];
                    good = hv2[(size_t)dof] == (double)(h.neighbors[(size_t)i] + 1);
                }
        }
        g_p2p_timeout_ticks = keep_timeout;
        const int rc3 = p2p_agree(good, "fs_space_enable_p2p_halo", "the self-test of the mapped buffers failed on this rank (sums or ghost values wrong, or a wait timed out)");
        if (rc3 != FS_OK) {
            pp.release();
            if (g_p2p_spaces == 0) g_p2p_red.release();
            return rc3;
        }
    }
    return FS_OK;
}

static int p2p_exchange_begin(fs_space_s* space, double* d_vec, hipStream_t s) {
    fs_halo_plan& h = space->halo;
    fs_p2p_halo& pp = h.p2p;
    FS_REQUIRE(!pp.pending, "peer-to-peer halo: an exchange was begun and never received");
    const int nn = (int)h.neighbors.size();
    hipLaunchKernelGGL(k_p2p_send, dim3(nn * pp.send_groups), dim3(FS_BLOCK), 0, s, pp.peers.p, pp.send_groups, pp.done.p, pp.counter.p,
                       pp.d_seq.p, d_vec, h.send_idx.p);
    FS_KERNEL_CHECK();
    pp.pending = d_vec;
    return FS_OK;
}

static int p2p_exchange_end(fs_space_s* space, hipStream_t s) {
    fs_halo_plan& h = space->halo;
    fs_p2p_halo& pp = h.p2p;
    FS_REQUIRE(pp.pending, "peer-to-peer halo: receive without a send");
    const int nn = (int)h.neighbors.size();
    static const int test_mode = p2p_test_mode();
    static int n_receives = 0;
    ++n_receives;
    const unsigned long long never = test_mode == 2 || (test_mode == 3 && n_receives > g_p2p_test_late) ? 1000000000ull : 0ull;
    const int grid = (int)std::min<int64_t>(16, std::max<int64_t>(1, (h.total_recv + FS_BLOCK * 8 - 1) / (FS_BLOCK * 8)));
    hipLaunchKernelGGL(k_p2p_recv, dim3(grid), dim3(FS_BLOCK), 0, s, nn, pp.flags, pp.d_seq.p, never, pp.recv,
                       std::max<int64_t>(h.total_recv, 1), h.total_recv, h.recv_idx.p, pp.pending, space->n_dofs_owned,
                       g_p2p_timeout_ticks, g_p2p_err);
    FS_KERNEL_CHECK();
    pp.pending = nullptr;
    return FS_OK;
}

bool fs_p2p_fusable(const fs_space_s* space) {
    const fs_halo_plan& h = space->halo;
    return h.active && h.p2p.enabled && g_p2p_red.enabled && fs_rt().comm != nullptr;
}

int fs_p2p_exchange_args(fs_space_s* space, const double* partials, int npart, double* sums_out, fs_p2p_rowsred* red, fs_p2p_sendrows* out) {
    fs_runtime& rt = fs_rt();
    fs_halo_plan& h = space->halo;
    fs_p2p_halo& pp = h.p2p;
    fs_p2p_reduce& R = g_p2p_red;
    FS_REQUIRE(!pp.pending, "peer-to-peer halo: an exchange was begun and never received");
    out->peers = pp.peers.p;
    out->send_idx = h.send_idx.p;
    out->total_send = h.total_send;
    out->counter = pp.counter.p + 1;
    out->d_seq = pp.d_seq.p;
    out->nn = (int)h.neighbors.size();
    out->own_flags = pp.flags;
    out->own_recv = pp.recv;
    out->recv_stride = std::max<int64_t>(h.total_recv, 1);
    out->recv_idx = h.recv_idx.p;
    out->total_recv = h.total_recv;
    out->n_owned = space->n_dofs_owned;
    out->timeout = g_p2p_timeout_ticks;
    out->err = g_p2p_err;
    red->partials = partials; red->npart = npart; red->sums_out = sums_out;
    red->peer_buf = (double* const*)R.peer_buf.p;
    red->peer_flags = (unsigned long long* const*)R.peer_flags.p;
    red->own_buf = R.buf;
    red->own_flags = R.flags;
    red->d_seq = R.d_seq.p;
    red->counter = R.counter.p;
    red->timeout = g_p2p_timeout_ticks;
    red->err = g_p2p_err;
    red->nr = rt.n_ranks; red->me = rt.rank;
    return FS_OK;
}

// Interior / boundary split of the owned rows, at the granularity the SpMV works at (slices of 64 rows): a slice is a
// BOUNDARY slice if any of its structural entries names a ghost column.  Interior slices are multiplied while the
// halo is in flight, boundary slices after it (fs_krylov.hip, spmv_overlapped).  Both lists keep the processing
// order of the space (slice_order), so the XCD-contiguous sweep is unchanged.
__global__ void __launch_bounds__(FS_BLOCK) k_slice_has_ghost(int64_t n_slices, int64_t n_owned_nodes,
                                                              const int64_t* __restrict__ slice_ptr,
                                                              const int32_t* __restrict__ sell_col, int32_t* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t base = slice_ptr[s] + lane;
        const int width = (int)((slice_ptr[s + 1] - slice_ptr[s]) >> 6);
        int g = 0;
        for (int k = 0; k < width; ++k) g |= sell_col[base + (int64_t)k * FS_SLICE] >= n_owned_nodes;
        const unsigned long long any = __ballot(g);
        if (lane == 0) flag[s] = any != 0ull;
    }
}

static int build_slice_split(fs_space_s* sp) {
    fs_halo_plan& h = sp->halo;
    hipStream_t s = fs_rt().stream;
    h.interior.release();
    h.boundary.release();
    h.items_interior.release();
    h.items_boundary.release();
    h.n_items_interior = h.n_items_boundary = -1;
    h.n_interior = h.n_boundary = 0;
    const int64_t ns = sp->n_slices;
    if (ns == 0) return FS_OK;
    dbuf<int32_t> flag;
    FS_CHECK(flag.alloc(ns));
    hipLaunchKernelGGL(k_slice_has_ghost, dim3(fs_grid_for(ns * 64)), dim3(FS_BLOCK), 0, s, ns, sp->n_nodes_owned,
                       sp->slice_ptr.p, sp->sell_col.p, flag.p);
    FS_KERNEL_CHECK();
    std::vector<int32_t> hf((size_t)ns), order;
    FS_CHECK(flag.download(hf.data(), ns, s));
    if (sp->slice_order.p) {
        order.resize((size_t)ns);
        FS_CHECK(sp->slice_order.download(order.data(), ns, s));
    }
    std::vector<int32_t> in, bd;
    for (int64_t q = 0; q < ns; ++q) {
        const int32_t sl = order.empty() ? (int32_t)q : order[(size_t)q];
        (hf[(size_t)sl] ? bd : in).push_back(sl);
    }
    h.n_interior = (int64_t)in.size();
    h.n_boundary = (int64_t)bd.size();
    FS_CHECK(h.interior.alloc(std::max<int64_t>(h.n_interior, 1)));
    FS_CHECK(h.boundary.alloc(std::max<int64_t>(h.n_boundary, 1)));
    FS_CHECK(h.interior.upload(in.data(), h.n_interior, s));
    FS_CHECK(h.boundary.upload(bd.data(), h.n_boundary, s));
    return FS_OK;
}

extern "C" int fs_space_set_halo_indexed(fs_space_t space, int n_neighbors, const int32_t* neighbor_ranks,
                                         const int64_t* send_counts, const int32_t* send_idx, const int64_t* recv_counts,
                                         const int32_t* recv_idx) {
    FS_REQUIRE(n_neighbors == 0 || recv_idx, "fs_space_set_halo_indexed: null scatter list");
    return set_halo_impl(space, n_neighbors, neighbor_ranks, send_counts, send_idx, recv_counts, recv_idx);
}

extern "C" int fs_space_set_halo(fs_space_t space, int n_neighbors, const int32_t* neighbor_ranks,
                                 const int64_t* send_counts, const int32_t* send_idx, const int64_t* recv_counts) {
    return set_halo_impl(space, n_neighbors, neighbor_ranks, send_counts, send_idx, recv_counts, nullptr);
}

static int set_halo_impl(fs_space_t space, int n_neighbors, const int32_t* neighbor_ranks,
                         const int64_t* send_counts, const int32_t* send_idx, const int64_t* recv_counts, const int32_t* recv_idx) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(space && n_neighbors >= 0, "fs_space_set_halo: bad arguments");
    fs_halo_plan& h = space->halo;
    FS_REQUIRE(!h.p2p.enabled, "fs_space_set_halo: the peer-to-peer exchange of this space is on; turn it off first (fs_space_enable_p2p_halo(space, 0), collective)");
    h.active = false;
    h.neighbors.clear(); h.send_counts.clear(); h.send_offsets.clear(); h.recv_counts.clear();
    h.recv_offsets.clear(); h.send_contiguous.clear(); h.send_first.clear();
    h.send_idx.release(); h.send_buf.release(); h.recv_idx.release(); h.recv_buf.release();
    h.total_send = h.total_recv = 0;
    if (n_neighbors == 0) return FS_OK;
    FS_REQUIRE(neighbor_ranks && send_counts && send_idx && recv_counts, "fs_space_set_halo: null pointer");
    int64_t so = 0, ro = 0;
    for (int i = 0; i < n_neighbors; ++i) {
        FS_REQUIRE(send_counts[i] >= 0 && recv_counts[i] >= 0, "fs_space_set_halo: negative count");
        h.neighbors.push_back(neighbor_ranks[i]);
        h.send_counts.push_back(send_counts[i]);
        h.recv_counts.push_back(recv_counts[i]);
        h.send_offsets.push_back(so);
        h.recv_offsets.push_back(ro);
        bool contiguous = true;
        for (int64_t k = 0; k < send_counts[i]; ++k) {
            const int32_t d = send_idx[so + k];
            FS_REQUIRE(d >= 0 && d < space->n_dofs_owned, "fs_space_set_halo: send index %d is not an owned dof", d);
            if (k > 0 && d != send_idx[so + k - 1] + 1) contiguous = false;
        }
        h.send_contiguous.push_back(contiguous ? 1 : 0);
        h.send_first.push_back(send_counts[i] ? send_idx[so] : 0);
        so += send_counts[i];
        ro += recv_counts[i];
    }
    FS_REQUIRE(ro == space->n_dofs_local - space->n_dofs_owned,
               "fs_space_set_halo: %lld ghosts announced, space has %lld", (long long)ro,
               (long long)(space->n_dofs_local - space->n_dofs_owned));
    h.total_send = so;
    h.total_recv = ro;
    FS_CHECK(h.send_idx.alloc(so));
    FS_CHECK(h.send_buf.alloc(so));
    FS_CHECK(h.send_idx.upload(send_idx, so, fs_rt().stream));
    if (recv_idx) {
        for (int64_t k = 0; k < ro; ++k)
            FS_REQUIRE(recv_idx[k] >= space->n_dofs_owned && recv_idx[k] < space->n_dofs_local,
                       "fs_space_set_halo_indexed: scatter index %d is not a ghost dof", recv_idx[k]);
        FS_CHECK(h.recv_idx.alloc(ro));
        FS_CHECK(h.recv_buf.alloc(ro));
        FS_CHECK(h.recv_idx.upload(recv_idx, ro, fs_rt().stream));
    }
    FS_CHECK(build_slice_split(space));
    h.active = true;
    h.early = -1;          // a new plan: the ranks agree again on the early start of the exchange (fs_krylov.hip)
    h.begun = false;
    return FS_OK;
}

// The exchange is split in two so that the rows that need no ghost value can be multiplied while it is in flight
// (SURVEY section 8e):  begin = pack on the compute stream, then grouped ncclSend/ncclRecv (+ scatter of an indexed
// halo) on the library's COMMUNICATION stream behind an event;  end = the compute stream waits for that stream.
// Only device-side dependencies: the host never blocks.
int fs_halo_comm_stream(fs_space_s* space, hipStream_t* out) {
    fs_halo_plan& h = space->halo;
    if (!h.comm_stream) {
        // (default priority on purpose: with the highest stream priority the grouped send / recv of a 1-rank RCCL communicator took
        // 162 instead of 31 us and an iteration 303 instead of 66 us on MI355X / ROCm 7.2 - tools/probes/rccl_self_halo_probe.py)
        FS_HIP(hipStreamCreateWithFlags(&h.comm_stream, hipStreamNonBlocking));
        FS_HIP(hipEventCreateWithFlags(&h.ev_ready, hipEventDisableTiming));
        FS_HIP(hipEventCreateWithFlags(&h.ev_done, hipEventDisableTiming));
    }
    *out = h.comm_stream;
    return FS_OK;
}

int fs_halo_begin_dev(fs_space_s* space, double* d_vec, hipStream_t s) {
    fs_halo_plan& h = space->halo;
    if (!h.active) return FS_OK;
    fs_runtime& rt = fs_rt();
    if (!rt.comm) {
        fs_set_error("halo exchange requested but no communicator is up (call fs_comm_init)");
        return FS_ERR_COMM;
    }
    if (h.p2p.enabled) return p2p_exchange_begin(space, d_vec, s);
    hipStream_t cs = nullptr;
    FS_CHECK(fs_halo_comm_stream(space, &cs));
    const int nn = (int)h.neighbors.size();
    for (int i = 0; i < nn; ++i) {
        if (!h.send_contiguous[i] && h.send_counts[i] > 0) {
            hipLaunchKernelGGL(k_pack, dim3(fs_grid_for(h.send_counts[i])), dim3(FS_BLOCK), 0, s, d_vec,
                               h.send_idx.p + h.send_offsets[i], h.send_counts[i], h.send_buf.p + h.send_offsets[i]);
        }
    }
    FS_KERNEL_CHECK();
    FS_HIP(hipEventRecord(h.ev_ready, s));
    FS_HIP(hipStreamWaitEvent(cs, h.ev_ready, 0));
    double* ghosts = d_vec + space->n_dofs_owned;
    host_timer timer(1);
    FS_NCCL(g_nccl.GroupStart());
    for (int i = 0; i < nn; ++i) {
        if (h.send_counts[i] > 0) {
            const double* src = h.send_contiguous[i] ? d_vec + h.send_first[i] : h.send_buf.p + h.send_offsets[i];
            FS_NCCL(g_nccl.Send(src, (size_t)h.send_counts[i], ncclDouble, h.neighbors[i], (ncclComm_t)rt.comm, cs));
        }
        if (h.recv_counts[i] > 0)
            FS_NCCL(g_nccl.Recv((h.recv_idx.p ? h.recv_buf.p : ghosts) + h.recv_offsets[i], (size_t)h.recv_counts[i], ncclDouble,
                                h.neighbors[i], (ncclComm_t)rt.comm, cs));
    }
    FS_NCCL(g_nccl.GroupEnd());
    if (h.recv_idx.p && h.total_recv > 0) {
        hipLaunchKernelGGL(k_unpack, dim3(fs_grid_for(h.total_recv)), dim3(FS_BLOCK), 0, cs, d_vec, h.recv_idx.p, h.total_recv, h.recv_buf.p);
        FS_KERNEL_CHECK();
    }
    FS_HIP(hipEventRecord(h.ev_done, cs));
    return FS_OK;
}

int fs_halo_end_dev(fs_space_s* space, hipStream_t s) {
    fs_halo_plan& h = space->halo;
    if (!h.active) return FS_OK;
    if (h.p2p.enabled) return p2p_exchange_end(space, s);
    FS_HIP(hipStreamWaitEvent(s, h.ev_done, 0));
    return FS_OK;
}

int fs_halo_exchange_dev(fs_space_s* space, double* d_vec, hipStream_t s) {
    FS_CHECK(fs_halo_begin_dev(space, d_vec, s));
    return fs_halo_end_dev(space, s);
}

extern "C" int fs_halo_exchange(fs_space_t space, fs_vector_t v) {
    FS_REQUIRE(space && v && v->d.n >= space->n_dofs_local, "fs_halo_exchange: vector shorter than the local dofs");
    hipStream_t s = fs_rt().stream;
    FS_CHECK(fs_halo_exchange_dev(space, v->d.p, s));
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

// Latency of the two collectives of a CG iteration, as the solver issues them: `reps` back-to-back 3-double all-reduces
// in the compute stream, and `reps` ghost refreshes of a scratch vector of this space (pack, grouped send / recv on the
// communication stream, wait), each timed with HIP events on the compute stream.  Collective: every rank calls it.
extern "C" int fs_comm_benchmark(fs_space_t space, int reps, double* allreduce_ms, double* halo_ms) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(reps > 0, "fs_comm_benchmark: reps must be positive");
    if (allreduce_ms) *allreduce_ms = 0.0;
    if (halo_ms) *halo_ms = 0.0;
    fs_runtime& rt = fs_rt();
    if (!rt.comm) return FS_OK;
    hipStream_t s = rt.stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    FS_HIP(hipEventCreate(&e0));
    FS_HIP(hipEventCreate(&e1));
    int rc = FS_OK;
    float ms = 0.f;
    dbuf<double> three, v;
    if ((rc = three.alloc(4)) == FS_OK && (rc = three.zero(s)) == FS_OK) {
        for (int pass = 0; pass < 2 && rc == FS_OK; ++pass) {        // pass 0 warms the collective up
            (void)hipEventRecord(e0, s);
            for (int i = 0; i < reps && rc == FS_OK; ++i) rc = fs_comm_allreduce_dev(three.p, 3, s);
            (void)hipEventRecord(e1, s);
            (void)hipEventSynchronize(e1);
        }
        if (rc == FS_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && allreduce_ms) *allreduce_ms = (double)ms / reps;
    }
    if (rc == FS_OK && space && space->halo.active && halo_ms) {
        if ((rc = v.alloc(space->n_dofs_local + 2)) == FS_OK && (rc = v.zero(s)) == FS_OK) {
            for (int pass = 0; pass < 2 && rc == FS_OK; ++pass) {
                (void)hipEventRecord(e0, s);
                for (int i = 0; i < reps && rc == FS_OK; ++i) rc = fs_halo_exchange_dev(space, v.p, s);
                (void)hipEventRecord(e1, s);
                (void)hipEventSynchronize(e1);
            }
            if (rc == FS_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) *halo_ms = (double)ms / reps;
        }
    }
    (void)hipStreamSynchronize(s);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

void fs_comm_preload() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(k_pack));
    (void)hipGetLastError();
}
