// Runtime, vectors and meshes of libfsamd.so (gfx950).
#include "fs_common.h"
#include <thread>
#include <vector>
#include <chrono>
#include <atomic>
#include <mutex>
#include "fs_kernels.h"
#include <map>
#include <unordered_map>
#include <vector>
#include <stdlib.h>

// ---- errors / runtime --------------------------------------------------------------
static thread_local char g_err[1024] = "";

void fs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

fs_runtime& fs_rt() {
    static fs_runtime rt;
    return rt;
}

// ---- block cache (see fs_common.h) ----------------------------------------------------
namespace {
struct block_pool {
    std::multimap<size_t, void*> idle;               // size -> block
    std::unordered_map<void*, size_t> size_of;       // every block handed out or idle
    size_t live = 0, cached = 0;
    std::recursive_mutex mu;                         // ctypes releases the GIL: a handle destroyed by one Python thread
                                                     // (_Handle.__del__) may race an allocation made by another
    size_t limit() {
        static size_t v = [] {
            const char* e = getenv("FS_POOL_MAX_MB");
            return (size_t)(e ? strtoull(e, nullptr, 10) : 16384ull) << 20;
        }();
        return v;
    }
};
block_pool& pool() {
    static block_pool* p = new block_pool;   // never destroyed: buffers of static objects are released at exit, after
    return *p;                               // function-local statics of this file would be gone
}
}  // namespace

// ---- pinned staging buffer of the small copies (fs_common.h) ----
namespace {
std::mutex g_staging_mu;
void* g_staging = nullptr;          // FS_STAGING_BYTES of pinned host memory, or nullptr (then the runtime's own path)
bool g_staging_tried = false;
void* staging_buffer() {            // (g_staging_mu held)
    if (!g_staging_tried) {
        g_staging_tried = true;
        static const bool off = getenv("FS_STAGING") && getenv("FS_STAGING")[0] == '0';
        if (!off && hipHostMalloc(&g_staging, FS_STAGING_BYTES, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            g_staging = nullptr;
        }
    }
    return g_staging;
}
}  // namespace
// the staging buffer itself, for kernels that write a small result straight into host memory (it is device-accessible): locked
// until fs_staging_unlock; nullptr (and not locked) when there is none
void* fs_staging_lock() {
    g_staging_mu.lock();
    void* st = staging_buffer();
    if (!st) g_staging_mu.unlock();
    return st;
}
void fs_staging_unlock() { g_staging_mu.unlock(); }
void fs_staging_prepare() {
    std::lock_guard<std::mutex> lock(g_staging_mu);
    (void)staging_buffer();
}
int fs_staged_copy(void* dst, const void* src, size_t bytes, bool to_device, hipStream_t s) {
    if (bytes == 0) return FS_OK;
    static const bool trace = getenv("FS_COPY_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    struct report {
        const std::chrono::steady_clock::time_point t0; size_t bytes; bool to_device, on;
        ~report() { if (on) fprintf(stderr, "[fs copy] %s %zu bytes %.3f ms\n", to_device ? "H2D" : "D2H", bytes, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
    } rep{t0, bytes, to_device, trace};
    if (bytes <= FS_STAGING_BYTES) {
        std::lock_guard<std::mutex> lock(g_staging_mu);
        if (void* st = staging_buffer()) {
            if (to_device) {
                memcpy(st, src, bytes);
                FS_HIP(hipMemcpyAsync(dst, st, bytes, hipMemcpyHostToDevice, s));
                FS_HIP(hipStreamSynchronize(s));
            } else {
                FS_HIP(hipMemcpyAsync(st, src, bytes, hipMemcpyDeviceToHost, s));
                FS_HIP(hipStreamSynchronize(s));
                memcpy(dst, st, bytes);
            }
            return FS_OK;
        }
    }
    FS_HIP(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, s));
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

void fs_pool_trim(size_t keep_bytes) {
    block_pool& P = pool();
    std::lock_guard<std::recursive_mutex> lock(P.mu);
    while (P.cached > keep_bytes && !P.idle.empty()) {
        auto it = std::prev(P.idle.end());           // largest first
        (void)hipFree(it->second);
        P.size_of.erase(it->second);
        P.cached -= it->first;
        P.idle.erase(it);
    }
}

void* fs_pool_alloc(size_t bytes) {
    block_pool& P = pool();
    if (bytes == 0) return nullptr;
    bytes = (bytes + 255) & ~(size_t)255;       // (kernels that read whole 16-byte groups - k_box_spmv's loaders - may touch the bytes behind an odd count)
    std::lock_guard<std::recursive_mutex> lock(P.mu);
    // the smallest idle block that holds the request and wastes at most a quarter of itself (small blocks: half)
    auto it = P.idle.lower_bound(bytes);
    if (it != P.idle.end() && it->first - bytes <= (it->first < (1u << 20) ? it->first / 2 : it->first / 4)) {
        void* p = it->second;
        P.cached -= it->first;
        P.live += it->first;
        P.idle.erase(it);
        return p;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {                           // give the cache back and try once more
        (void)hipGetLastError();
        fs_pool_trim(0);
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        fs_set_error("hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    P.size_of[p] = bytes;
    P.live += bytes;
    return p;
}

void fs_pool_free(void* p) {
    block_pool& P = pool();
    std::lock_guard<std::recursive_mutex> lock(P.mu);
    auto it = P.size_of.find(p);
    if (it == P.size_of.end()) {                     // not ours (cannot happen): hand it to the driver
        (void)hipFree(p);
        return;
    }
    const size_t bytes = it->second;
    P.live -= bytes;
    if (bytes > P.limit()) {
        (void)hipFree(p);
        P.size_of.erase(it);
        return;
    }
    P.idle.emplace(bytes, p);
    P.cached += bytes;
    if (P.cached > P.limit()) {                      // over the limit: drop the largest idle blocks
        fs_pool_trim(P.limit());
    }
}

void fs_pool_stats(size_t* live_bytes, size_t* cached_bytes) {
    std::lock_guard<std::recursive_mutex> lock(pool().mu);
    if (live_bytes) *live_bytes = pool().live;
    if (cached_bytes) *cached_bytes = pool().cached;
}

extern "C" int fs_memory_trim(void) {
    FS_CHECK(fs_require_init());
    FS_HIP(hipStreamSynchronize(fs_rt().stream));
    fs_pool_trim(0);
    return FS_OK;
}

extern "C" int fs_memory_info(int64_t* live_bytes, int64_t* cached_bytes) {
    size_t l = 0, c = 0;
    fs_pool_stats(&l, &c);
    if (live_bytes) *live_bytes = (int64_t)l;
    if (cached_bytes) *cached_bytes = (int64_t)c;
    return FS_OK;
}

int fs_require_init() {
    if (!fs_rt().initialised) {
        // implicit init on device 0 keeps single-GPU callers short, but still fails
        // loudly when there is no GPU: there is no CPU path in this library.
        return fs_init(0);
    }
    return FS_OK;
}

std::recursive_mutex& fs_solve_mutex() {
    static std::recursive_mutex* m = new std::recursive_mutex;      // never destroyed (static-destruction order)
    return *m;
}

uint64_t fs_next_serial() {
    static std::atomic<uint64_t> next{1};
    return next.fetch_add(1);
}

extern "C" const char* fs_last_error(void) { return g_err; }
extern "C" const char* fs_version(void) { return "fenicssolver_amd 0.1 (gfx950, fp64)"; }

extern "C" int fs_device_count(int* count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        n = 0;
        (void)hipGetLastError();
    }
    *count = n;
    return FS_OK;
}

__global__ void k_profile_marker(int phase, int* sink);
__global__ void k_warm_noop(int* sink) { if (sink) *sink = 0; }

void fs_staging_prepare();

extern "C" int fs_init(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        (void)hipGetLastError();
        fs_set_error("no HIP device visible (hipGetDeviceCount: %s); libfsamd has no CPU fallback",
                     e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return FS_ERR_NO_DEVICE;
    }
    FS_REQUIRE(device_id >= 0 && device_id < n, "fs_init: device %d out of range (0..%d)", device_id, n - 1);
    fs_runtime& rt = fs_rt();
    if (rt.initialised && rt.device == device_id) return FS_OK;
    if (rt.initialised) fs_pool_trim(0);          // idle blocks of the device this process used before
    FS_HIP(hipSetDevice(device_id));
    // FS_WAIT=spin|yield|block selects how the host waits for the device (default: the runtime's choice; spinning
    // made no measurable difference to the Krylov loops on the MI355X test boxes)
    if (const char* w = getenv("FS_WAIT")) {
        unsigned flags = hipDeviceScheduleAuto;
        if (!strcmp(w, "spin")) flags = hipDeviceScheduleSpin;
        else if (!strcmp(w, "yield")) flags = hipDeviceScheduleYield;
        else if (!strcmp(w, "block")) flags = hipDeviceScheduleBlockingSync;
        (void)hipSetDeviceFlags(flags);     // fails harmlessly when the context already exists
        (void)hipGetLastError();
    }
    hipDeviceProp_t prop;
    FS_HIP(hipGetDeviceProperties(&prop, device_id));
    if (rt.stream) (void)hipStreamDestroy(rt.stream);
    FS_HIP(hipStreamCreateWithFlags(&rt.stream, hipStreamNonBlocking));
    rt.device = device_id;
    rt.compute_units = prop.multiProcessorCount;
    rt.initialised = true;
    // The code objects of the library are loaded here, not at the first launch out of each of them: the object of the set-up
    // kernels holds 3 300 rocPRIM instantiations (every algorithm for 13 architectures) and takes 35 ms to load - it used to be
    // seven eighths of the first sparsity pattern of a process.  FS_PRELOAD=0: load on first use, as the runtime does by itself.
    static const bool preload = !(getenv("FS_PRELOAD") && getenv("FS_PRELOAD")[0] == '0');
    if (preload) {
        // (what every solve needs.  The AMG object - 4.7 MB, its own rocPRIM sorts and scans, 40 ms to load - and the saddle-point
        // and communication objects are loaded by their first launch: a heat-conduction case never pays for them)
        static const bool init_timing = getenv("FS_INIT_TIMING") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!init_timing) return;
            const auto t1 = std::chrono::steady_clock::now();
            fprintf(stderr, "[fs_init timing] %-24s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        };
        // (FS_INIT_TIMING=1, round 5, MI355X box: 40 ms the set-up object with its rocPRIM instantiations, 3 + 6 ms the other two - of
        // 170 - 310 ms of fs_init, the rest being the HIP runtime's own start: hipGetDeviceCount 117 ms, the stream 20 ms)
        // The runtime's own first-use costs, taken HERE, next to the code objects, on a helper thread (they sit in other parts of the
        // runtime than the module loader): the first hipGraph a process instantiates (9 - 13 ms of graph machinery: the second step of
        // a time loop used to pay them, the first one ran graph-free to dodge them) and the first pageable host-to-device /
        // device-to-host copies (staging buffers, 8 ms each: the first apply_dirichlet of a process).  FS_WARM=0: not done.
        static const bool warm = !(getenv("FS_WARM") && getenv("FS_WARM")[0] == '0');
        std::thread warm_thread;
        if (warm) {
            hipStream_t main_stream = rt.stream;
            warm_thread = std::thread([device_id, main_stream] {
                if (hipSetDevice(device_id) != hipSuccess) return;
                fs_staging_prepare();
                hipStream_t ws = nullptr;
                if (hipStreamCreateWithFlags(&ws, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return; }
                void* d = nullptr;
                std::vector<char> h(1 << 16, 1);
                if (hipMalloc(&d, FS_STAGING_BYTES) == hipSuccess) {
                    (void)hipMemcpyAsync(d, h.data(), h.size(), hipMemcpyHostToDevice, ws);
                    (void)hipMemcpyAsync(h.data(), d, h.size(), hipMemcpyDeviceToHost, ws);
                    (void)hipStreamSynchronize(ws);
                    // (copies above 64 KB take another path of the runtime with a first-use cost of its own: 7 ms for the first
                    // 120 KB device-to-host copy of a process, pinned destination or not)
                    void* st = nullptr;
                    {
                        std::lock_guard<std::mutex> lock(g_staging_mu);
                        st = staging_buffer();
                    }
                    if (getenv("FS_INIT_TIMING")) fprintf(stderr, "[fs_init timing] helper: staging buffer %p\n", st);
                    if (st) {
                        // (on the library's OWN stream - nobody else uses it while fs_init runs -: the cost is per stream)
                        (void)hipMemcpyAsync(d, st, 120000, hipMemcpyHostToDevice, main_stream);
                        (void)hipStreamSynchronize(main_stream);
                        const auto tw = std::chrono::steady_clock::now();
                        const hipError_t e1 = hipMemcpyAsync(st, d, 120000, hipMemcpyDeviceToHost, main_stream);
                        const hipError_t e2 = hipStreamSynchronize(main_stream);
                        if (getenv("FS_INIT_TIMING")) fprintf(stderr, "[fs_init timing] helper: first 120000-byte D2H on the library's stream %.3f ms (%d %d)\n",
                                                              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count(), (int)e1, (int)e2);
                        (void)hipMemcpyAsync(d, st, FS_STAGING_BYTES, hipMemcpyHostToDevice, main_stream);
                        (void)hipMemcpyAsync(st, d, FS_STAGING_BYTES, hipMemcpyDeviceToHost, main_stream);
                        (void)hipStreamSynchronize(main_stream);
                    }
                }
                hipGraph_t g = nullptr;
                hipGraphExec_t ge = nullptr;
                if (hipStreamBeginCapture(ws, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    k_warm_noop<<<1, 64, 0, ws>>>(nullptr);       // (NOT k_profile_marker: tools/summarize_profiles.py counts those)
                    k_warm_noop<<<1, 64, 0, ws>>>(nullptr);
                    if (hipStreamEndCapture(ws, &g) == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
                        (void)hipGraphLaunch(ge, ws);
                        (void)hipStreamSynchronize(ws);
                    }
                }
                if (ge) (void)hipGraphExecDestroy(ge);
                if (g) (void)hipGraphDestroy(g);
                if (d) (void)hipFree(d);
                (void)hipStreamDestroy(ws);
                (void)hipGetLastError();
            });
        }
        // (measured and not kept, round 6: the three objects on three threads - the runtime's loader takes them one after the other
        // whoever asks: 49 + 11.5 ms, and the helper thread's first graph then queues behind them, 12 ms more)
        fs_symbolic_preload();
        lap("set-up object (rocPRIM)");
        fs_assemble_preload();
        lap("assembly object");
        fs_krylov_preload();
        lap("Krylov object");
        if (warm_thread.joinable()) warm_thread.join();
        lap("first graph + first copies (helper thread)");
    }
    return FS_OK;
}

extern "C" int fs_device_info(char* name, int name_len, int* compute_units, int64_t* hbm_bytes) {
    FS_CHECK(fs_require_init());
    hipDeviceProp_t prop;
    FS_HIP(hipGetDeviceProperties(&prop, fs_rt().device));
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return FS_OK;
}

extern "C" int fs_device_synchronize(void) {
    FS_CHECK(fs_require_init());
    FS_HIP(hipStreamSynchronize(fs_rt().stream));
    FS_HIP(hipDeviceSynchronize());
    return FS_OK;
}

// A launch that does nothing but carry a name into a kernel trace: the measurement scripts split the phases of one traced command
// (problem sizes, kernel variants) at these launches instead of at the name of some set-up kernel (tools/summarize_profiles.py).
__global__ void k_profile_marker(int phase, int* sink) {
    if (sink && phase < 0) *sink = phase;
}

extern "C" int fs_profile_marker(int phase) {
    FS_CHECK(fs_require_init());
    k_profile_marker<<<1, 64, 0, fs_rt().stream>>>(phase, nullptr);
    FS_HIP(hipGetLastError());
    return FS_OK;
}

// ---- vectors ------------------------------------------------------------------------
__global__ void k_fill(double* __restrict__ v, int64_t n, double a) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) v[i] = a;
}

__global__ void k_axpy(double* __restrict__ y, const double* __restrict__ x, int64_t n, double a) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] += a * x[i];
}

__global__ void k_assign_entries(double* __restrict__ v, int64_t n, const int32_t* __restrict__ dst, const int32_t* __restrict__ src, int bs) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n * bs; t += stride) {
        const int64_t i = t / bs;
        const int c = (int)(t - i * bs);
        v[(int64_t)dst[i] * bs + c] = v[(int64_t)src[i] * bs + c];
    }
}

extern "C" int fs_vector_assign_entries(fs_vector_t v, int64_t n, const int32_t* dst_nodes, const int32_t* src_nodes, int block) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(v && block >= 1 && (n == 0 || (dst_nodes && src_nodes)), "fs_vector_assign_entries: bad arguments");
    if (n == 0) return FS_OK;
    for (int64_t i = 0; i < n; ++i)
        FS_REQUIRE(dst_nodes[i] >= 0 && src_nodes[i] >= 0 && ((int64_t)dst_nodes[i] + 1) * block <= v->d.n && ((int64_t)src_nodes[i] + 1) * block <= v->d.n,
                   "fs_vector_assign_entries: pair %lld out of range", (long long)i);
    hipStream_t s = fs_rt().stream;
    dbuf<int32_t> d_d, d_s;
    FS_CHECK(d_d.alloc(n)); FS_CHECK(d_s.alloc(n));
    FS_CHECK(d_d.upload(dst_nodes, n, s)); FS_CHECK(d_s.upload(src_nodes, n, s));
    hipLaunchKernelGGL(k_assign_entries, dim3(fs_grid_for(n * block)), dim3(FS_BLOCK), 0, s, v->d.p, n, d_d.p, d_s.p, block);
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(s));      // the index buffers go back to the block cache
    return FS_OK;
}

extern "C" int fs_vector_create(int64_t n, fs_vector_t* out) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(n >= 0 && out, "fs_vector_create: bad arguments");
    fs_vector_s* v = new fs_vector_s();
    int rc = v->d.alloc(n);
    if (rc == FS_OK) rc = v->d.zero(fs_rt().stream);
    if (rc == FS_OK && hipStreamSynchronize(fs_rt().stream) != hipSuccess) rc = FS_ERR_HIP;
    if (rc != FS_OK) {
        delete v;
        return rc;
    }
    *out = v;
    return FS_OK;
}

extern "C" int fs_vector_size(fs_vector_t v, int64_t* n) {
    FS_REQUIRE(v && n, "fs_vector_size: null");
    *n = v->d.n;
    return FS_OK;
}

extern "C" int fs_vector_set(fs_vector_t v, const double* host, int64_t n) {
    FS_REQUIRE(v && host && n == v->d.n, "fs_vector_set: size mismatch (%lld vs %lld)", (long long)n,
               v ? (long long)v->d.n : -1LL);
    return v->d.upload(host, n, fs_rt().stream);
}

extern "C" int fs_vector_get(fs_vector_t v, double* host, int64_t n) {
    FS_REQUIRE(v && host && n <= v->d.n, "fs_vector_get: size mismatch");
    return v->d.download(host, n, fs_rt().stream);
}

extern "C" int fs_vector_fill(fs_vector_t v, double value) {
    FS_REQUIRE(v, "fs_vector_fill: null");
    if (v->d.n == 0) return FS_OK;
    hipLaunchKernelGGL(k_fill, dim3(fs_grid_for(v->d.n)), dim3(FS_BLOCK), 0, fs_rt().stream, v->d.p, v->d.n, value);
    FS_KERNEL_CHECK();
    return FS_OK;
}

extern "C" int fs_vector_copy(fs_vector_t dst, fs_vector_t src, int64_t n) {
    FS_REQUIRE(dst && src && n >= 0 && n <= dst->d.n && n <= src->d.n, "fs_vector_copy: bad arguments");
    if (n == 0) return FS_OK;
    FS_HIP(hipMemcpyAsync(dst->d.p, src->d.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, fs_rt().stream));
    FS_HIP(hipStreamSynchronize(fs_rt().stream));
    return FS_OK;
}

extern "C" int fs_vector_axpy(fs_vector_t y, double a, fs_vector_t x) {
    FS_REQUIRE(x && y && x->d.n == y->d.n, "fs_vector_axpy: size mismatch");
    if (y->d.n == 0) return FS_OK;
    hipLaunchKernelGGL(k_axpy, dim3(fs_grid_for(y->d.n)), dim3(FS_BLOCK), 0, fs_rt().stream, y->d.p, x->d.p, y->d.n, a);
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(fs_rt().stream));
    return FS_OK;
}

__global__ void k_add_entries(double* __restrict__ v, const int32_t* __restrict__ idx, const double* __restrict__ vals, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) atomicAdd(&v[idx[i]], vals[i]);
}

extern "C" int fs_vector_add_entries(fs_vector_t v, int64_t n, const int32_t* idx, const double* vals) {
    FS_REQUIRE(v && n >= 0 && (n == 0 || (idx && vals)), "fs_vector_add_entries: bad arguments");
    if (n == 0) return FS_OK;
    for (int64_t i = 0; i < n; ++i)
        FS_REQUIRE(idx[i] >= 0 && idx[i] < v->d.n, "fs_vector_add_entries: index %d outside the vector of %lld entries", idx[i], (long long)v->d.n);
    hipStream_t s = fs_rt().stream;
    dbuf<int32_t> di;
    dbuf<double> dv;
    FS_CHECK(di.alloc(n));
    FS_CHECK(dv.alloc(n));
    FS_CHECK(di.upload(idx, n, s));
    FS_CHECK(dv.upload(vals, n, s));
    hipLaunchKernelGGL(k_add_entries, dim3(fs_grid_for(n)), dim3(FS_BLOCK), 0, s, v->d.p, di.p, dv.p, n);
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

extern "C" int fs_vector_dot(fs_vector_t x, fs_vector_t y, double* result) {
    FS_REQUIRE(x && y && result, "fs_vector_dot: null");
    int64_t n = x->d.n < y->d.n ? x->d.n : y->d.n;
    hipStream_t s = fs_rt().stream;
    int grid = fs_grid_for(n, FS_BLOCK, FS_MAX_PARTIAL_BLOCKS);
    dbuf<double> part;
    FS_CHECK(part.alloc(grid + 1));
    hipLaunchKernelGGL(k_dot_partial, dim3(grid), dim3(FS_BLOCK), 0, s, x->d.p, y->d.p, n, part.p);
    FS_KERNEL_CHECK();
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(FS_SUM_BLOCK), 0, s, part.p, grid, 1, part.p + grid);
    FS_KERNEL_CHECK();
    FS_HIP(hipMemcpyAsync(result, part.p + grid, sizeof(double), hipMemcpyDeviceToHost, s));
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

extern "C" int fs_vector_destroy(fs_vector_t v) {
    delete v;
    return FS_OK;
}

// ---- meshes ---------------------------------------------------------------------------
__global__ void k_pad_xyz(const double* __restrict__ xyz3, double* __restrict__ xyz4, int64_t nv) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < nv; i += stride) {
        xyz4[4 * i + 0] = xyz3[3 * i + 0];
        xyz4[4 * i + 1] = xyz3[3 * i + 1];
        xyz4[4 * i + 2] = xyz3[3 * i + 2];
        xyz4[4 * i + 3] = 0.0;
    }
}

__global__ void k_iota64(int64_t* __restrict__ v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) v[i] = i;
}

extern "C" int fs_mesh_create(int gdim, int64_t nv, const double* xyz, int64_t nc, const int32_t* cells,
                              int verts_per_cell, int64_t n_owned, fs_mesh_t* out) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(out && xyz && cells, "fs_mesh_create: null pointer");
    const bool tri = gdim == 2 && verts_per_cell == 3;
    if (!tri && (gdim != 3 || verts_per_cell != 4)) {
        fs_set_error("fs_mesh_create: tetrahedral meshes in 3D and triangular meshes in 2D are supported (gdim=%d, verts_per_cell=%d)",
                     gdim, verts_per_cell);
        return FS_ERR_UNSUPPORTED;
    }
    FS_REQUIRE(nv > 0 && nc > 0 && n_owned >= 0 && n_owned <= nv, "fs_mesh_create: bad sizes");
    FS_REQUIRE(nv < (int64_t)INT32_MAX, "fs_mesh_create: vertex count exceeds int32");
    for (int64_t i = 0; i < nc * verts_per_cell; ++i) {
        if (cells[i] < 0 || cells[i] >= nv) {
            fs_set_error("fs_mesh_create: cell %lld references vertex %d outside [0,%lld)", (long long)(i / verts_per_cell),
                         cells[i], (long long)nv);
            return FS_ERR_INVALID;
        }
    }
    FS_REQUIRE(!tri || n_owned == nv, "fs_mesh_create: triangular meshes are single-GPU for now");
    hipStream_t s = fs_rt().stream;
    fs_mesh_s* m = new fs_mesh_s();
    m->nv = nv;
    m->nc = nc;
    m->n_owned = n_owned;
    m->tdim = tri ? 2 : 3;
    dbuf<double> tmp;
    int rc = FS_OK;
    // triangles are stored in the same padded layout: (x, y, 0, 0) and (v0, v1, v2, -1)
    std::vector<double> x3;
    std::vector<int32_t> c4;
    if (tri) {
        x3.resize((size_t)nv * 3);
        for (int64_t i = 0; i < nv; ++i) { x3[3 * i] = xyz[2 * i]; x3[3 * i + 1] = xyz[2 * i + 1]; x3[3 * i + 2] = 0.0; }
        c4.resize((size_t)nc * 4);
        for (int64_t c = 0; c < nc; ++c) { c4[4 * c] = cells[3 * c]; c4[4 * c + 1] = cells[3 * c + 1]; c4[4 * c + 2] = cells[3 * c + 2]; c4[4 * c + 3] = -1; }
        xyz = x3.data();
        cells = c4.data();
    }
    if ((rc = m->xyz.alloc(nv * 4)) != FS_OK || (rc = m->cells.alloc(nc * 4)) != FS_OK ||
        (rc = m->gid.alloc(nv)) != FS_OK || (rc = tmp.alloc(nv * 3)) != FS_OK ||
        (rc = tmp.upload(xyz, nv * 3, s)) != FS_OK || (rc = m->cells.upload(cells, nc * 4, s)) != FS_OK) {
        delete m;
        return rc;
    }
    hipLaunchKernelGGL(k_pad_xyz, dim3(fs_grid_for(nv)), dim3(FS_BLOCK), 0, s, tmp.p, m->xyz.p, nv);
    hipLaunchKernelGGL(k_iota64, dim3(fs_grid_for(nv)), dim3(FS_BLOCK), 0, s, m->gid.p, nv);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        fs_set_error("fs_mesh_create: kernel launch failed");
        delete m;
        return FS_ERR_HIP;
    }
    *out = m;
    return FS_OK;
}

// Slab of dolfin.BoxMesh generated on the device.  One thread per local vertex /
// per local hexahedron (6 tets).
struct box_desc {
    int64_t nx, ny, nz;      // global cells per axis
    double p0[3], p1[3];
    int64_t zb, ze;          // owned vertex planes [zb, ze)
    int64_t kz0, kz1;        // local cell layers [kz0, kz1)
    int64_t plane;           // (nx+1)*(ny+1)
    int64_t n_owned;
    int has_lower, has_upper;
};

__device__ __forceinline__ int32_t box_local_id(const box_desc& d, int64_t ix, int64_t iy, int64_t iz) {
    const int64_t inplane = iy * (d.nx + 1) + ix;
    if (iz >= d.zb && iz < d.ze) return (int32_t)((iz - d.zb) * d.plane + inplane);
    if (iz == d.zb - 1) return (int32_t)(d.n_owned + inplane);
    return (int32_t)(d.n_owned + (d.has_lower ? d.plane : 0) + inplane);  // iz == ze
}

__global__ void k_box_vertices(box_desc d, int64_t nv, double* __restrict__ xyz4, int64_t* __restrict__ gid) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < nv; i += stride) {
        int64_t iz, inplane;
        if (i < d.n_owned) {
            iz = d.zb + i / d.plane;
            inplane = i % d.plane;
        } else {
            int64_t j = i - d.n_owned;
            if (d.has_lower && j < d.plane) {
                iz = d.zb - 1;
                inplane = j;
            } else {
                iz = d.ze;
                inplane = j - (d.has_lower ? d.plane : 0);
            }
        }
        const int64_t iy = inplane / (d.nx + 1);
        const int64_t ix = inplane % (d.nx + 1);
        // same expression order as DOLFIN's BoxMesh: a + (i*(b-a))/n
        xyz4[4 * i + 0] = d.p0[0] + ((double)ix * (d.p1[0] - d.p0[0])) / (double)d.nx;
        xyz4[4 * i + 1] = d.p0[1] + ((double)iy * (d.p1[1] - d.p0[1])) / (double)d.ny;
        xyz4[4 * i + 2] = d.p0[2] + ((double)iz * (d.p1[2] - d.p0[2])) / (double)d.nz;
        xyz4[4 * i + 3] = 0.0;
        gid[i] = iz * d.plane + inplane;
    }
}

__global__ void k_box_cells(box_desc d, int64_t nhex, int32_t* __restrict__ cells) {
    int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; h < nhex; h += stride) {
        const int64_t ix = h % d.nx;
        const int64_t iy = (h / d.nx) % d.ny;
        const int64_t kz = d.kz0 + h / (d.nx * d.ny);
        int32_t v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            v[c] = box_local_id(d, ix + (c & 1), iy + ((c >> 1) & 1), kz + ((c >> 2) & 1));
        // six tets around the v0-v7 diagonal, vertices ascending by GLOBAL index
        // (global order of the hex corners is v0<v1<...<v7)
        const int t[6][4] = {{0, 1, 3, 7}, {0, 1, 5, 7}, {0, 4, 5, 7}, {0, 2, 3, 7}, {0, 4, 6, 7}, {0, 2, 6, 7}};
        int4* out = reinterpret_cast<int4*>(cells) + h * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) out[k] = make_int4(v[t[k][0]], v[t[k][1]], v[t[k][2]], v[t[k][3]]);
    }
}

extern "C" int fs_mesh_create_box(int64_t nx, int64_t ny, int64_t nz, const double p0[3], const double p1[3],
                                  int64_t zplane_begin, int64_t zplane_end, fs_mesh_t* out) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(out && p0 && p1, "fs_mesh_create_box: null pointer");
    FS_REQUIRE(nx > 0 && ny > 0 && nz > 0, "fs_mesh_create_box: cell counts must be positive");
    FS_REQUIRE(zplane_begin >= 0 && zplane_end <= nz + 1 && zplane_begin < zplane_end,
               "fs_mesh_create_box: owned plane range [%lld,%lld) outside [0,%lld]", (long long)zplane_begin,
               (long long)zplane_end, (long long)(nz + 1));
    box_desc d;
    d.nx = nx; d.ny = ny; d.nz = nz;
    for (int i = 0; i < 3; ++i) { d.p0[i] = p0[i]; d.p1[i] = p1[i]; }
    d.zb = zplane_begin; d.ze = zplane_end;
    d.plane = (nx + 1) * (ny + 1);
    d.has_lower = zplane_begin > 0;
    d.has_upper = zplane_end < nz + 1;
    d.n_owned = (zplane_end - zplane_begin) * d.plane;
    d.kz0 = zplane_begin > 0 ? zplane_begin - 1 : 0;
    d.kz1 = zplane_end < nz ? zplane_end : nz;
    const int64_t nv = d.n_owned + (d.has_lower + d.has_upper) * d.plane;
    const int64_t nhex = (d.kz1 - d.kz0) * nx * ny;
    const int64_t nc = nhex * 6;
    FS_REQUIRE(nv < (int64_t)INT32_MAX && nc < (int64_t)INT32_MAX, "fs_mesh_create_box: slab exceeds int32 indexing");
    hipStream_t s = fs_rt().stream;
    fs_mesh_s* m = new fs_mesh_s();
    m->nv = nv; m->nc = nc; m->n_owned = d.n_owned;
    m->box_h[0] = (p1[0] - p0[0]) / (double)nx;
    m->box_h[1] = (p1[1] - p0[1]) / (double)ny;
    m->box_h[2] = (p1[2] - p0[2]) / (double)nz;
    if (!d.has_lower && !d.has_upper) { m->box_n[0] = nx; m->box_n[1] = ny; m->box_n[2] = nz; }
    int rc = FS_OK;
    if ((rc = m->xyz.alloc(nv * 4)) != FS_OK || (rc = m->cells.alloc(nc * 4)) != FS_OK ||
        (rc = m->gid.alloc(nv)) != FS_OK) {
        delete m;
        return rc;
    }
    hipLaunchKernelGGL(k_box_vertices, dim3(fs_grid_for(nv)), dim3(FS_BLOCK), 0, s, d, nv, m->xyz.p, m->gid.p);
    hipLaunchKernelGGL(k_box_cells, dim3(fs_grid_for(nhex)), dim3(FS_BLOCK), 0, s, d, nhex, m->cells.p);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        fs_set_error("fs_mesh_create_box: kernel launch failed");
        delete m;
        return FS_ERR_HIP;
    }
    *out = m;
    return FS_OK;
}

extern "C" int fs_mesh_set_global_ids(fs_mesh_t mesh, const int64_t* global_ids) {
    FS_REQUIRE(mesh && global_ids, "fs_mesh_set_global_ids: null pointer");
    FS_CHECK(mesh->gid.upload(global_ids, mesh->nv, fs_rt().stream));
    return FS_OK;
}

extern "C" int fs_mesh_info(fs_mesh_t mesh, int64_t* nv, int64_t* nc, int64_t* n_owned) {
    FS_REQUIRE(mesh, "fs_mesh_info: null mesh");
    if (nv) *nv = mesh->nv;
    if (nc) *nc = mesh->nc;
    if (n_owned) *n_owned = mesh->n_owned;
    return FS_OK;
}

extern "C" int fs_mesh_get(fs_mesh_t mesh, double* xyz, int32_t* cells, int64_t* global_ids) {
    FS_REQUIRE(mesh, "fs_mesh_get: null mesh");
    hipStream_t s = fs_rt().stream;
    if (xyz) {
        std::vector<double> tmp((size_t)mesh->nv * 4);
        FS_CHECK(mesh->xyz.download(tmp.data(), mesh->nv * 4, s));
        for (int64_t i = 0; i < mesh->nv; ++i) {
            xyz[3 * i + 0] = tmp[4 * i + 0];
            xyz[3 * i + 1] = tmp[4 * i + 1];
            xyz[3 * i + 2] = tmp[4 * i + 2];
        }
    }
    if (cells) FS_CHECK(mesh->cells.download(cells, mesh->nc * 4, s));
    if (global_ids) FS_CHECK(mesh->gid.download(global_ids, mesh->nv, s));
    return FS_OK;
}

extern "C" int fs_mesh_destroy(fs_mesh_t mesh) {
    delete mesh;
    return FS_OK;
}
