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
#include <atomic>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <sys/stat.h>
#include <time.h>
#include <rccl/rccl.h>

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

// ---- test transport: host-staged exchange through POSIX shared memory ---------------------------------------
// FS_COMM_TRANSPORT=shm replaces the RCCL calls by device<->host copies and a shared-memory mailbox, so that
// SEVERAL RANKS CAN SHARE ONE GPU (RCCL refuses duplicate devices).  It exists to run the complete distributed
// algorithm - partition, halo packing, ghost layout, fused dots + all-reduce, restarts - on the 1-GPU test
// boxes; it is slow by construction and never selected unless the variable is set.
struct shm_header {
    std::atomic<int> arrive;
    std::atomic<int> generation;
};
struct shm_comm {
    int fd = -1;
    char* base = nullptr;
    size_t bytes = 0;
    int n_ranks = 1, rank = 0;
    char name[64] = {0};
    static constexpr int64_t PAIR_CAP = 1 << 18;   // doubles per (src,dst) halo buffer (pages are touched only when used)
    static constexpr int RED_CAP = 8192;           // doubles per rank in the reduction mailbox (longer vectors go in rounds)
    shm_header* hdr() { return (shm_header*)base; }
    double* red(int r) { return (double*)(base + 4096) + (size_t)r * RED_CAP; }
    double* pair(int src, int dst) {
        return (double*)(base + 4096 + (size_t)n_ranks * RED_CAP * 8) + ((size_t)src * n_ranks + dst) * PAIR_CAP;
    }
    static size_t size_for(int n) { return 4096 + (size_t)n * RED_CAP * 8 + (size_t)n * n * PAIR_CAP * 8; }
    void barrier() {
        shm_header* h = hdr();
        const int gen = h->generation.load(std::memory_order_acquire);
        if (h->arrive.fetch_add(1, std::memory_order_acq_rel) == n_ranks - 1) {
            h->arrive.store(0, std::memory_order_relaxed);
            h->generation.store(gen + 1, std::memory_order_release);
        } else {
            while (h->generation.load(std::memory_order_acquire) == gen) usleep(20);
        }
    }
};
static bool g_use_shm() {
    const char* t = getenv("FS_COMM_TRANSPORT");
    return t && !strcmp(t, "shm");
}
static shm_comm* g_shm = nullptr;

static int shm_open_segment(shm_comm* c, bool create) {
    c->bytes = shm_comm::size_for(c->n_ranks);
    c->fd = shm_open(c->name, create ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
    if (c->fd < 0) { fs_set_error("shm transport: shm_open(%s) failed", c->name); return FS_ERR_COMM; }
    if (create && ftruncate(c->fd, (off_t)c->bytes) != 0) { fs_set_error("shm transport: ftruncate failed"); return FS_ERR_COMM; }
    void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
    if (p == MAP_FAILED) { fs_set_error("shm transport: mmap of %zu bytes failed", c->bytes); return FS_ERR_COMM; }
    c->base = (char*)p;
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
    if (g_use_shm()) {     // the id is the name of the segment; it is created by fs_comm_init of rank 0
        memset(id, 0, FS_UNIQUE_ID_BYTES);
        snprintf(id, 64, "/fsamd_%d_%ld", (int)getpid(), (long)time(nullptr));
        return FS_OK;
    }
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
    if (g_use_shm()) {
        FS_REQUIRE(n_ranks <= 4, "shm test transport: at most 4 ranks");
        shm_comm* c = new shm_comm();
        c->n_ranks = n_ranks; c->rank = rank;
        strncpy(c->name, id, 63);
        // rank 0 creates and zeroes the segment, the others wait until it exists with its full size
        int rc = FS_ERR_COMM;
        if (rank == 0) {
            rc = shm_open_segment(c, true);
            if (rc == FS_OK) { c->hdr()->arrive.store(0); c->hdr()->generation.store(1000); }
        } else {
            for (int tries = 0; tries < 20000 && rc != FS_OK; ++tries) {
                c->fd = shm_open(c->name, O_RDWR, 0600);
                if (c->fd >= 0) {
                    struct stat st;
                    if (fstat(c->fd, &st) == 0 && (size_t)st.st_size == shm_comm::size_for(n_ranks)) {
                        close(c->fd);
                        rc = shm_open_segment(c, false);
                        if (rc == FS_OK)
                            while (c->hdr()->generation.load() < 1000) usleep(100);
                        break;
                    }
                    close(c->fd);
                }
                usleep(500);
            }
        }
        if (rc != FS_OK) { delete c; if (rc == FS_ERR_COMM && !*fs_last_error()) fs_set_error("shm transport: rendezvous failed"); return rc; }
        g_shm = c;
        rt.comm = (void*)c;
        rt.n_ranks = n_ranks;
        rt.rank = rank;
        c->barrier();
        return FS_OK;
    }
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
    if (rt.comm && g_shm) {
        (void)hipStreamSynchronize(rt.stream);
        g_shm->barrier();
        munmap(g_shm->base, g_shm->bytes);
        close(g_shm->fd);
        if (g_shm->rank == 0) shm_unlink(g_shm->name);
        delete g_shm;
        g_shm = nullptr;
        rt.comm = nullptr;
    } else if (rt.comm) {
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
    if (g_shm) {
        std::vector<double> h((size_t)n);
        FS_HIP(hipMemcpyAsync(h.data(), d_inout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
        FS_HIP(hipStreamSynchronize(s));
        for (int64_t off = 0; off < n; off += shm_comm::RED_CAP) {
            const int m = (int)std::min<int64_t>(shm_comm::RED_CAP, n - off);
            memcpy(g_shm->red(g_shm->rank), h.data() + off, (size_t)m * sizeof(double));
            g_shm->barrier();
            for (int i = 0; i < m; ++i) {
                double acc = 0.0;
                for (int r = 0; r < g_shm->n_ranks; ++r) acc += g_shm->red(r)[i];   // rank order: same bits everywhere
                h[off + i] = acc;
            }
            g_shm->barrier();
        }
        FS_HIP(hipMemcpyAsync(d_inout, h.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
        FS_HIP(hipStreamSynchronize(s));
        return FS_OK;
    }
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
    if (g_shm) {
        // test transport: rounds of RED_CAP doubles per rank through the reduction mailbox
        for (int64_t off = 0; off < n_max; off += shm_comm::RED_CAP) {
            const int64_t m = std::min<int64_t>(shm_comm::RED_CAP, n_max - off);
            const int64_t mine = std::max<int64_t>(0, std::min<int64_t>(m, n_send - off));
            if (mine > 0) memcpy(g_shm->red(g_shm->rank), host_send + off, (size_t)mine * sizeof(double));
            g_shm->barrier();
            for (int r = 0; r < nr; ++r) memcpy(host_recv + (int64_t)r * n_max + off, g_shm->red(r), (size_t)m * sizeof(double));
            g_shm->barrier();
        }
        return FS_OK;
    }
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
    if (g_shm) {
        FS_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < nn; ++i) {
            if (h.send_counts[i] == 0) continue;
            FS_REQUIRE(h.send_counts[i] <= shm_comm::PAIR_CAP, "shm test transport: halo of %lld values", (long long)h.send_counts[i]);
            const double* src = h.send_contiguous[i] ? d_vec + h.send_first[i] : h.send_buf.p + h.send_offsets[i];
            FS_HIP(hipMemcpy(g_shm->pair(g_shm->rank, h.neighbors[i]), src, (size_t)h.send_counts[i] * sizeof(double), hipMemcpyDeviceToHost));
        }
        g_shm->barrier();
        double* rbase = h.recv_idx.p ? h.recv_buf.p : ghosts;
        for (int i = 0; i < nn; ++i)
            if (h.recv_counts[i] > 0)
                FS_HIP(hipMemcpy(rbase + h.recv_offsets[i], g_shm->pair(h.neighbors[i], g_shm->rank), (size_t)h.recv_counts[i] * sizeof(double), hipMemcpyHostToDevice));
        g_shm->barrier();
        if (h.recv_idx.p && h.total_recv > 0) {
            hipLaunchKernelGGL(k_unpack, dim3(fs_grid_for(h.total_recv)), dim3(FS_BLOCK), 0, s, d_vec, h.recv_idx.p, h.total_recv, h.recv_buf.p);
            FS_KERNEL_CHECK();
        }
        return FS_OK;
    }
    FS_NCCL(g_nccl.GroupStart());
    for (int i = 0; i < nn; ++i) {
        if (h.send_counts[i] > 0) {
            const double* src = h.send_contiguous[i] ? d_vec + h.send_first[i] : h.send_buf.p + h.send_offsets[i];
            FS_NCCL(g_nccl.Send(src, (size_t)h.send_counts[i], ncclDouble, h.neighbors[i], (ncclComm_t)rt.comm, s));
        }
        if (h.recv_counts[i] > 0)
            FS_NCCL(g_nccl.Recv((h.recv_idx.p ? h.recv_buf.p : ghosts) + h.recv_offsets[i], (size_t)h.recv_counts[i], ncclDouble,
                                h.neighbors[i], (ncclComm_t)rt.comm, s));
    }
    FS_NCCL(g_nccl.GroupEnd());
    if (h.recv_idx.p && h.total_recv > 0) {
        hipLaunchKernelGGL(k_unpack, dim3(fs_grid_for(h.total_recv)), dim3(FS_BLOCK), 0, s, d_vec, h.recv_idx.p, h.total_recv, h.recv_buf.p);
        FS_KERNEL_CHECK();
    }
    return FS_OK;
}

extern "C" int fs_halo_exchange(fs_space_t space, fs_vector_t v) {
    FS_REQUIRE(space && v && v->d.n >= space->n_dofs_local, "fs_halo_exchange: vector shorter than the local dofs");
    hipStream_t s = fs_rt().stream;
    FS_CHECK(fs_halo_exchange_dev(space, v->d.p, s));
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}
