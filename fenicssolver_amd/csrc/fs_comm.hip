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
#include <dlfcn.h>
#include <stdlib.h>
#include <rccl/rccl.h>

struct rccl_api {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static rccl_api g_nccl;

static int rccl_load() {
    if (g_nccl.handle) return FS_OK;
    // Prefer the RCCL of the ROCm install this library's HIP runtime comes from; a process that has
    // imported torch also carries torch's bundled copy under the same SONAME.  FS_RCCL_PATH overrides.
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

int fs_comm_allreduce_dev(double* d_inout, int n, hipStream_t s) {
    fs_runtime& rt = fs_rt();
    if (!rt.comm) return FS_OK;  // one rank
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

// ---- halo ----------------------------------------------------------------------------------
__global__ void k_pack(const double* __restrict__ v, const int32_t* __restrict__ idx, int64_t n,
                       double* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = v[idx[i]];
}

extern "C" int fs_space_set_halo(fs_space_t space, int n_neighbors, const int32_t* neighbor_ranks,
                                 const int64_t* send_counts, const int32_t* send_idx, const int64_t* recv_counts) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(space && n_neighbors >= 0, "fs_space_set_halo: bad arguments");
    fs_halo_plan& h = space->halo;
    h.active = false;
    h.neighbors.clear(); h.send_counts.clear(); h.send_offsets.clear(); h.recv_counts.clear();
    h.recv_offsets.clear(); h.send_contiguous.clear(); h.send_first.clear();
    h.send_idx.release(); h.send_buf.release();
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
    h.active = true;
    return FS_OK;
}

int fs_halo_exchange_dev(fs_space_s* space, double* d_vec, hipStream_t s) {
    fs_halo_plan& h = space->halo;
    if (!h.active) return FS_OK;
    fs_runtime& rt = fs_rt();
    if (!rt.comm) {
        fs_set_error("halo exchange requested but no communicator is up (call fs_comm_init)");
        return FS_ERR_COMM;
    }
    const int nn = (int)h.neighbors.size();
    for (int i = 0; i < nn; ++i) {
        if (!h.send_contiguous[i] && h.send_counts[i] > 0) {
            hipLaunchKernelGGL(k_pack, dim3(fs_grid_for(h.send_counts[i])), dim3(FS_BLOCK), 0, s, d_vec,
                               h.send_idx.p + h.send_offsets[i], h.send_counts[i], h.send_buf.p + h.send_offsets[i]);
        }
    }
    FS_KERNEL_CHECK();
    double* ghosts = d_vec + space->n_dofs_owned;
    FS_NCCL(g_nccl.GroupStart());
    for (int i = 0; i < nn; ++i) {
        if (h.send_counts[i] > 0) {
            const double* src = h.send_contiguous[i] ? d_vec + h.send_first[i] : h.send_buf.p + h.send_offsets[i];
            FS_NCCL(g_nccl.Send(src, (size_t)h.send_counts[i], ncclDouble, h.neighbors[i], (ncclComm_t)rt.comm, s));
        }
        if (h.recv_counts[i] > 0)
            FS_NCCL(g_nccl.Recv(ghosts + h.recv_offsets[i], (size_t)h.recv_counts[i], ncclDouble, h.neighbors[i],
                                (ncclComm_t)rt.comm, s));
    }
    FS_NCCL(g_nccl.GroupEnd());
    return FS_OK;
}

extern "C" int fs_halo_exchange(fs_space_t space, fs_vector_t v) {
    FS_REQUIRE(space && v && v->d.n >= space->n_dofs_local, "fs_halo_exchange: vector shorter than the local dofs");
    hipStream_t s = fs_rt().stream;
    FS_CHECK(fs_halo_exchange_dev(space, v->d.p, s));
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}
