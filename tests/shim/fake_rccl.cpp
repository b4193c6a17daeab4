// TEST INFRASTRUCTURE - not part of the product.
//
// A stand-in for librccl.so that lets SEVERAL RANKS SHARE ONE GPU: RCCL refuses two ranks on the same device and
// the test boxes have exactly one.  libfsamd.so dlopen()s whatever FS_RCCL_PATH names and resolves the ten nccl*
// entry points it uses (fs_comm.hip, rccl_load); pointed at this library, the multi-rank tests drive the very same
// FS_NCCL(...) call sites production uses - ncclGroupStart/Send/Recv/GroupEnd halos, the 3-double ncclAllReduce of
// the CG loop, ncclAllGather - with the data moved by device<->host copies and a POSIX shared-memory segment.
// Slow by construction.  Semantics kept: point-to-point ordering per (src, dst) pair, grouped sends/recvs progress
// together (no deadlock on symmetric exchanges), collectives are called by every rank, stream order is respected
// (the stream is drained before the host touches the buffers and the call returns with the result in place).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <vector>
#include <string.h>
#include <stdio.h>
#include <stdint.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

namespace {
constexpr int MAX_RANKS = 8;
constexpr size_t PAIR_BYTES = (size_t)8 << 20;   // per (src, dst) mailbox; pages are touched only when used
constexpr size_t RED_BYTES = (size_t)1 << 16;    // per rank, collectives go in rounds of this size

struct pair_hdr {
    std::atomic<uint64_t> written, consumed;
    uint64_t bytes;
    char pad[40];
};
struct seg_hdr {
    std::atomic<int> arrive, generation, ready;
    char pad[52];
    pair_hdr pairs[MAX_RANKS * MAX_RANKS];
};
struct shim_comm {
    int fd = -1, n = 1, rank = 0;
    char* base = nullptr;
    size_t bytes = 0;
    char name[64] = {0};
    seg_hdr* hdr() { return (seg_hdr*)base; }
    static size_t hdr_bytes() { return (sizeof(seg_hdr) + 4095) & ~(size_t)4095; }
    char* red(int r) { return base + hdr_bytes() + (size_t)r * RED_BYTES; }
    char* pair(int src, int dst) { return base + hdr_bytes() + (size_t)n * RED_BYTES + ((size_t)src * n + dst) * PAIR_BYTES; }
    pair_hdr& ph(int src, int dst) { return hdr()->pairs[src * MAX_RANKS + dst]; }
    static size_t size_for(int n) { return hdr_bytes() + (size_t)n * RED_BYTES + (size_t)n * n * PAIR_BYTES; }
    void barrier() {
        seg_hdr* h = hdr();
        const int gen = h->generation.load(std::memory_order_acquire);
        if (h->arrive.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
            h->arrive.store(0, std::memory_order_relaxed);
            h->generation.store(gen + 1, std::memory_order_release);
        } else {
            while (h->generation.load(std::memory_order_acquire) == gen) usleep(10);
        }
    }
};

struct p2p_op {
    bool send;
    void* buf;
    size_t bytes;
    int peer;
    shim_comm* c;
    hipStream_t s;
    bool done;
};
thread_local int g_group_depth = 0;
thread_local std::vector<p2p_op> g_ops;

size_t dtype_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

bool try_op(p2p_op& o) {
    shim_comm* c = o.c;
    if (o.send) {
        pair_hdr& h = c->ph(c->rank, o.peer);
        if (h.consumed.load(std::memory_order_acquire) != h.written.load(std::memory_order_relaxed)) return false;  // slot busy
        if (hipMemcpy(c->pair(c->rank, o.peer), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
        h.bytes = o.bytes;
        h.written.fetch_add(1, std::memory_order_release);
    } else {
        pair_hdr& h = c->ph(o.peer, c->rank);
        if (h.written.load(std::memory_order_acquire) == h.consumed.load(std::memory_order_relaxed)) return false;  // nothing yet
        if (h.bytes != o.bytes) fprintf(stderr, "[fake_rccl] rank %d: recv of %zu bytes from %d meets a send of %llu\n", c->rank, o.bytes, o.peer, (unsigned long long)h.bytes);
        if (hipMemcpy(o.buf, c->pair(o.peer, c->rank), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return false;
        h.consumed.fetch_add(1, std::memory_order_release);
    }
    o.done = true;
    return true;
}

ncclResult_t progress_all() {
    for (auto& o : g_ops)
        if (hipStreamSynchronize(o.s) != hipSuccess) return ncclUnhandledCudaError;
    size_t left = g_ops.size();
    const time_t t0 = time(nullptr);
    while (left) {
        bool any = false;
        for (auto& o : g_ops)
            if (!o.done && try_op(o)) { --left; any = true; }
        if (!any) {
            usleep(10);
            if (time(nullptr) - t0 > 120) { g_ops.clear(); return ncclSystemError; }   // a peer died
        }
    }
    g_ops.clear();
    return ncclSuccess;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, 64, "/fsamd_shim_%d_%ld", (int)getpid(), (long)time(nullptr));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    shim_comm* c = new shim_comm();
    c->n = nranks; c->rank = rank;
    strncpy(c->name, id.internal, 63);
    c->bytes = shim_comm::size_for(nranks);
    bool ok = false;
    if (rank == 0) {
        c->fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        ok = c->fd >= 0 && ftruncate(c->fd, (off_t)c->bytes) == 0;
    } else {
        for (int tries = 0; tries < 60000 && !ok; ++tries) {
            c->fd = shm_open(c->name, O_RDWR, 0600);
            if (c->fd >= 0) {
                struct stat st;
                if (fstat(c->fd, &st) == 0 && (size_t)st.st_size == c->bytes) { ok = true; break; }
                close(c->fd);
            }
            usleep(500);
        }
    }
    if (ok) {
        void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
        ok = p != MAP_FAILED;
        c->base = (char*)p;
    }
    if (!ok) { delete c; return ncclSystemError; }
    if (rank == 0) c->hdr()->ready.store(1, std::memory_order_release);   // fresh segments are zero-filled
    else while (c->hdr()->ready.load(std::memory_order_acquire) == 0) usleep(100);
    c->barrier();
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    shim_comm* c = (shim_comm*)comm;
    if (!c) return ncclSuccess;
    c->barrier();
    munmap(c->base, c->bytes);
    close(c->fd);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream) {
    shim_comm* c = (shim_comm*)comm;
    if (datatype != ncclFloat64 || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
    std::vector<double> h(count);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (count && hipMemcpy(h.data(), sendbuff, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    const size_t cap = RED_BYTES / 8;
    for (size_t off = 0; off < count; off += cap) {
        const size_t m = count - off < cap ? count - off : cap;
        memcpy(c->red(c->rank), h.data() + off, m * 8);
        c->barrier();
        for (size_t i = 0; i < m; ++i) {
            double acc = ((double*)c->red(0))[i];
            for (int r = 1; r < c->n; ++r) {      // rank order: every rank obtains the same bits
                const double v = ((double*)c->red(r))[i];
                acc = op == ncclSum ? acc + v : (v > acc ? v : acc);
            }
            h[off + i] = acc;
        }
        c->barrier();
    }
    if (count && hipMemcpy(recvbuff, h.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    shim_comm* c = (shim_comm*)comm;
    const size_t es = dtype_bytes(datatype);
    if (!es) return ncclInvalidArgument;
    const size_t bytes = sendcount * es;
    std::vector<char> h(bytes), all(bytes * c->n);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (bytes && hipMemcpy(h.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    for (size_t off = 0; off < bytes; off += RED_BYTES) {
        const size_t m = bytes - off < RED_BYTES ? bytes - off : RED_BYTES;
        memcpy(c->red(c->rank), h.data() + off, m);
        c->barrier();
        for (int r = 0; r < c->n; ++r) memcpy(all.data() + (size_t)r * bytes + off, c->red(r), m);
        c->barrier();
    }
    if (bytes && hipMemcpy(recvbuff, all.data(), bytes * c->n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    ++g_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth == 0) return progress_all();
    return ncclSuccess;
}

static ncclResult_t p2p(bool send, void* buf, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    shim_comm* c = (shim_comm*)comm;
    const size_t bytes = count * dtype_bytes(datatype);
    if (!dtype_bytes(datatype) || peer < 0 || peer >= c->n || bytes > PAIR_BYTES) return ncclInvalidArgument;
    g_ops.push_back(p2p_op{send, buf, bytes, peer, c, stream, false});
    if (g_group_depth == 0) return progress_all();
    return ncclSuccess;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return p2p(true, const_cast<void*>(sendbuff), count, datatype, peer, comm, stream);
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return p2p(false, recvbuff, count, datatype, peer, comm, stream);
}

const char* ncclGetErrorString(ncclResult_t result) {
    switch (result) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake_rccl: HIP call failed";
        case ncclSystemError: return "fake_rccl: shared-memory segment / peer failure";
        case ncclInvalidArgument: return "fake_rccl: invalid argument (dtype, op, peer or message larger than the mailbox)";
        case ncclInvalidUsage: return "fake_rccl: invalid usage";
        default: return "fake_rccl: error";
    }
}

}  // extern "C"
