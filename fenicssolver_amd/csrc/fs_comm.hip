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
    host_timer timer(0);
    FS_NCCL(g_nccl.AllReduce(d_inout, d_inout, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)rt.comm, s));
    return FS_OK;
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
