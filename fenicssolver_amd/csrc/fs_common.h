// Internal definitions shared by the libfsamd.so translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdarg.h>
#include <vector>
#include <string>

#include "../../include/fenicssolver_amd.h"

#define FS_WAVE 64            // CDNA wavefront
#define FS_SLICE 64           // SELL slice height = one wavefront
#define FS_BLOCK 256          // workgroup size of every kernel (4 waves)
#define FS_MAX_PARTIAL_BLOCKS 4096

// ---- error plumbing -----------------------------------------------------------
void fs_set_error(const char* fmt, ...);

#define FS_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            fs_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                         __LINE__);                                                       \
            return FS_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

#define FS_CHECK(call)              \
    do {                            \
        int rc__ = (call);          \
        if (rc__ != FS_OK) return rc__; \
    } while (0)

#define FS_REQUIRE(cond, ...)         \
    do {                              \
        if (!(cond)) {                \
            fs_set_error(__VA_ARGS__); \
            return FS_ERR_INVALID;    \
        }                             \
    } while (0)

#define FS_KERNEL_CHECK() FS_HIP(hipGetLastError())

// ---- runtime state -------------------------------------------------------------
struct fs_runtime {
    bool initialised = false;
    int device = -1;
    int compute_units = 0;
    hipStream_t stream = nullptr;  // every kernel of the library runs on this stream
    // communicator (RCCL), see fs_comm.hip
    int n_ranks = 1;
    int rank = 0;
    void* comm = nullptr;
};
fs_runtime& fs_rt();
// Serial numbers of spaces and matrices: heap and pool addresses are recycled after a destroy, so anything cached per
// operator (the captured CG batch of fs_krylov.hip) is keyed on these, never on addresses alone.
uint64_t fs_next_serial();
// The solver workspaces (fs_krylov.hip, fs_saddle.hip) and the one compute stream belong to the process: solves are serialised
// by this lock, so that concurrent calls from several host threads are slow rather than wrong.
#include <mutex>
std::recursive_mutex& fs_solve_mutex();
struct fs_matrix_s;
void fs_ns_reset_dummy_rows(fs_matrix_s* J, hipStream_t s);   // fs_saddle.hip: unit diagonal on the dummy pressure slots
int fs_require_init();

// ---- device memory -------------------------------------------------------------
// Blocks released by the library are kept (up to FS_POOL_MAX_MB, default 16384) and handed out again for requests of
// about their size: hipFree synchronises the device and hipMalloc of a large block takes 0.1-1 ms, which a time loop
// re-assembling its operators and an AMG set-up with its dozens of temporaries pay every time.  A released block is
// re-used at once, which rests on this invariant: every kernel and copy that touches a pooled block is ordered on the
// one stream of fs_runtime - EXCEPT the halo send / recv / unpack and the overlapped all-reduce on the communication
// stream, and for those no buffer may be released between fs_halo_begin_dev and the fs_halo_end_dev that makes the
// compute stream wait for them (the only blocks they touch are the vector being exchanged, the plan's own send / recv
// buffers and the solver workspace, all of which outlive the solve).  The pool's book-keeping is guarded by a mutex.
void* fs_pool_alloc(size_t bytes);   // nullptr (error message set) when the device is out of memory
void fs_pool_free(void* p);
void fs_pool_trim(size_t keep_bytes);   // give cached blocks back to the driver until at most keep_bytes stay
void fs_pool_stats(size_t* live_bytes, size_t* cached_bytes);
// Small host <-> device copies go through ONE pinned staging buffer of the library (FS_STAGING_BYTES, allocated by fs_init's helper
// thread): a hipMemcpy between device and PAGEABLE host memory above a few KB makes the runtime pin the host pages first - 9 ms
// for the 56 KB list of a structure build (round 6, tools/probes/first_step_probe.py), every time the pages are new.
// Returns FS_OK / FS_ERR_HIP; synchronises the stream.  Larger copies: the runtime's own path (pinning pays there).
constexpr size_t FS_STAGING_BYTES = 1 << 20;
int fs_staged_copy(void* dst, const void* src, size_t bytes, bool to_device, hipStream_t s);
void* fs_staging_lock();             // the buffer itself (device-accessible host memory) for a kernel's small result; nullptr: none, not locked
void fs_staging_unlock();

// ---- device buffer -------------------------------------------------------------
template <typename T>
struct dbuf {
    T* p = nullptr;
    int64_t n = 0;
    int alloc(int64_t count) {
        release();
        n = count;
        if (count == 0) return FS_OK;
        p = static_cast<T*>(fs_pool_alloc((size_t)count * sizeof(T)));
        if (!p) { n = 0; return FS_ERR_HIP; }
        return FS_OK;
    }
    int zero(hipStream_t s) {
        if (n) FS_HIP(hipMemsetAsync(p, 0, (size_t)n * sizeof(T), s));
        return FS_OK;
    }
    int upload(const T* host, int64_t count, hipStream_t s) {
        if (count) return fs_staged_copy(p, host, (size_t)count * sizeof(T), true, s);
        return FS_OK;
    }
    int download(T* host, int64_t count, hipStream_t s) const {
        if (count) return fs_staged_copy(host, p, (size_t)count * sizeof(T), false, s);
        return FS_OK;
    }
    void release() {
        if (p) fs_pool_free(p);
        p = nullptr;
        n = 0;
    }
    void swap(dbuf& o) {
        T* tp = p; p = o.p; o.p = tp;
        const int64_t tn = n; n = o.n; o.n = tn;
    }
    ~dbuf() { release(); }
    dbuf() = default;
    dbuf(const dbuf&) = delete;
    dbuf& operator=(const dbuf&) = delete;
};

// ---- handles ---------------------------------------------------------------------
struct fs_mesh_s {
    int64_t nv = 0;       // local vertices (owned + ghost)
    int64_t nc = 0;       // local cells
    int64_t n_owned = 0;  // owned vertices (first n_owned local ids)
    int tdim = 3;         // 3: tetrahedra; 2: triangles (xyz stores (x,y,0,0), cells (v0,v1,v2,-1))
    dbuf<double> xyz;     // [nv][4] padded (x,y,z,0): two 16-B loads per vertex
    dbuf<int32_t> cells;  // [nc][4] local vertex ids
    dbuf<int64_t> gid;    // [nv] global vertex ids
    // uniform box meshes (fs_mesh_create_box): the grid spacing.  The P1 scalar assembly snaps every edge-vector component to a
    // multiple of it, which makes the rows of the operator translation-invariant BIT FOR BIT (vertex coordinates i * h are not:
    // their differences carry rounding noise of 1e-16 that differs from cell to cell) - what lets the solver find the few
    // dozen distinct rows of such an operator (fs_krylov.hip, row dictionary).  0 = not a uniform box: nothing is snapped.
    double box_h[3] = {0.0, 0.0, 0.0};
    // ... and, when this process holds the WHOLE box (one GPU), its cell counts: vertex v sits at (v % (nx + 1), ...), x fastest -
    // what the solver's lattice order of a CG2 space on it is computed from (fs_lattice.hip).  0 = a slab, or not a box.
    int64_t box_n[3] = {0, 0, 0};
    // reference rows of the P1 stiffness / mass matrix of a box mesh's six cell types (fs_assemble.hip, k_assemble_p1_box_gather):
    // [6 types][4 row vertices][5] = (|det J|, g_row . g_b for the row's rotated local vertices b = 0 .. 3), computed on the first
    // hexahedron with the SNAPPED geometry every cell of that type has bit for bit.  Empty until the first assembly that uses it.
    dbuf<double> box_ref;
    // the same for CG2 (k_assemble_p2_box_gather): [6 types][10 local dofs][11] = (volume, the row's ten quadrature sums of grad phi_a . grad phi_b)
    dbuf<double> box_ref2;
};

// Peer-to-peer ghost refresh (opt-in, one node): every rank owns a fine-grained receive buffer + arrival flags that its
// neighbours map through hipIpc and WRITE INTO from a kernel; the receiver's kernel waits for the flags (fs_comm.hip).
struct fs_p2p_peer {
    double* recv;                   // the neighbour's receive buffer (mapped), [2][peer_total]
    unsigned long long* flags;      // the neighbour's arrival flags (mapped), [2][peer_nn]
    int64_t recv_offset, peer_total;
    int64_t send_offset, send_count;
    int64_t send_first;             // first row of a contiguous send list (-1: not contiguous)
    int32_t peer_slot, peer_nn;
};
struct fs_p2p_halo {
    bool enabled = false;
    double* recv = nullptr;
    unsigned long long* flags = nullptr;
    std::vector<void*> opened;
    dbuf<fs_p2p_peer> peers;
    dbuf<uint32_t> done;            // per neighbour: workgroups of the running send that have stored their share
    int send_groups = 1;            // workgroups per neighbour
    double* pending = nullptr;      // vector of the exchange begun (plain send kernel) and not yet received
    dbuf<uint32_t> counter;         // [2]: workgroups through - of the plain send kernel, of the exchange kernel's send part
    dbuf<unsigned long long> d_seq; // device-side sequence number of the last executed exchange (fs_comm.hip)
    uint64_t generation = 0;        // which enable of the process this is: everything a captured batch bakes in (flags, counters,
                                    // index lists, mapped peers) is new after a disable / enable of the same space
    void release();
    ~fs_p2p_halo() { release(); }
};

// Non-Newtonian viscosity law of CoupledNavierStokesSolver.viscosity (:194-213) as the kernels take it: kind 0 Newtonian;
// 1: nu (p / pref)^ex (the branch without a temperature); 2: nu (1 + cp p/pref)(1 - ct T/tref) with the CG1 temperature T
// (one value per local NODE of the Taylor-Hood space, read at the vertex nodes) - the solving_temperature branch (:199-203)
struct fs_visc_dev {
    int kind = 0;
    double pref = 0.0, ex = 0.0, cp = 0.0, ct = 0.0, tref = 1.0;
    const double* T = nullptr;
};
#ifdef __HIPCC__
__device__ __forceinline__ double fs_viscosity(const fs_visc_dev& V, double nu, double p, double T) {
    if (V.kind == 1) return nu * pow(p / V.pref, V.ex);
    if (V.kind == 2) return (nu * (1.0 + (p / V.pref) * V.cp)) * (1.0 - (T / V.tref) * V.ct);
    return nu;
}
#endif

struct fs_halo_plan {
    bool active = false;
    fs_p2p_halo p2p;
    std::vector<int> neighbors;
    std::vector<int64_t> send_counts, send_offsets, recv_counts, recv_offsets;
    std::vector<char> send_contiguous;     // send list of neighbour i is a contiguous range
    std::vector<int64_t> send_first;       // first index of that range
    dbuf<int32_t> send_idx;                // concatenated
    dbuf<double> send_buf;                 // packed values (non-contiguous lists)
    dbuf<int32_t> recv_idx;                // optional scatter list (local dof of every received value)
    dbuf<double> recv_buf;
    int64_t total_send = 0, total_recv = 0;
    // overlap of the exchange with the interior rows: the grouped send/recv runs on comm_stream between two events
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    bool begun = false;                    // an exchange was started ahead of the product that will wait for it (fs_krylov.hip)
    int early = -1;                        // 1: EVERY rank sends a prefix / suffix of its rows (agreed by an all-reduce), 0: no, -1: not asked yet
    int64_t early_a = 0, early_b = 0;
    int fuse = 0;                          // 1: every rank can run the fused peer-to-peer iteration (agreed with `early`)
    // slices (in processing order) without / with ghost columns
    dbuf<int32_t> interior, boundary;
    dbuf<int32_t> items_interior, items_boundary;   // work items of the two lists for the row-dictionary product (fs_space_s::dict_items)
    int64_t n_items_interior = -1, n_items_boundary = -1;      // -1: not built yet
    int64_t n_interior = 0, n_boundary = 0;
    ~fs_halo_plan() {
        if (ev_ready) (void)hipEventDestroy(ev_ready);
        if (ev_done) (void)hipEventDestroy(ev_done);
        if (comm_stream) (void)hipStreamDestroy(comm_stream);
    }
};

struct fs_space_s {
    const uint64_t serial = fs_next_serial();
    fs_mesh_s* mesh = nullptr;
    fs_visc_dev visc;             // Taylor-Hood spaces: the law attached by fs_space_set_viscosity_law (kind 0: the form's own fields)
    int degree = 1;
    int ncomp = 1;
    // cell -> node table: P1 aliases mesh->cells (4 vertices); P2 holds [nc][10] = 4 vertices + 6 edge
    // nodes (nv + lexicographic edge index; UFC local edge order e0=(v2,v3) ... e5=(v0,v1))
    int ndof_cell = 4;
    const int32_t* cell_dofs = nullptr;
    dbuf<int32_t> cell_dofs_store;
    int64_t n_edges = 0;
    dbuf<int32_t> edges;          // [n_edges][2] (P2 only), ascending vertex pairs, in edge-node order
    dbuf<uint64_t> edge_keys;     // [n_edges] sorted search keys of the edge nodes
    int edge_grouped = 0;         // 0: key = (v0<<32|v1); 1: key = ((v1-v0)<<32|v0)  (<= 16 distinct v1-v0)
    // P2 node numbering: [owned vertices | owned edges | ghost vertices | ghost edges].  An edge is owned by the
    // rank that owns its endpoint of smaller GLOBAL id (that rank has every cell of the edge).  edge_node[i] =
    // node id of the edge at position i of the sorted key table; one GPU: nv + i.
    int64_t n_edges_owned = 0;
    dbuf<int32_t> edge_node;      // [n_edges]
    int64_t n_nodes_local = 0, n_nodes_owned = 0;  // node level
    int64_t n_dofs_local = 0, n_dofs_owned = 0;    // = nodes * ncomp
    // node-level sparsity: CSR + hybrid SELL-64 / per-slice DIA
    int64_t nnz_nodes = 0;        // node-pair entries
    int64_t n_slices = 0;
    int64_t n_dia_slices = 0;     // slices stored in diagonal (offset) form
    int64_t dia_entries = 0;      // stored entries that belong to DIA slices
    int64_t sell_entries = 0;     // stored node-pair entries (sum width*64), padding included
    int max_row = 0;              // largest slice width (storage positions per row)
    dbuf<int32_t> rowptr;         // [n_nodes_owned+1]
    dbuf<int32_t> colidx;         // [nnz_nodes]
    dbuf<int64_t> slice_ptr;      // [n_slices+1] offsets into the value/column arrays (entries)
    // column of every stored entry; NEGATIVE = not a structural entry (padding), value ~c is a safe
    // column to read.  Always present (assembly tables, Dirichlet, export); the SpMV reads it only
    // for SELL slices.
    dbuf<int32_t> sell_col;       // [sell_entries]
    // DIA slices: all 64 rows share one sorted list of (col - row) offsets, so the SpMV needs no
    // per-entry column index: dia_ptr[s] >= 0 indexes dia_off, -1 = SELL slice
    // processing order of the slices in the SpMV (empty = identity).  CG2 spaces number their edge nodes class by class,
    // so a contiguous eighth of the rows (what one XCD sweeps) is one edge class over the WHOLE mesh and every XCD
    // pulls all of x through its L2; ordered by the position of the slice's first node instead, an XCD sweeps one
    // slab of the mesh for all classes and x is fetched once
    dbuf<int32_t> slice_order;    // [n_slices]
    dbuf<int32_t> dia_ptr;        // [n_slices]
    dbuf<int32_t> dia_off;        // per DIA slice: [split][offset list A: width][offset list B: width, only if split < 64] -
                                  // rows [split, 64) of a SPLIT slice use list B (fs_symbolic.hip, k_slice_analyze)
    // two-rows-per-lane product (k_dia_pair_spmv): pairs of slices, consecutive in processing order, both complete DIA
    // slices with identical offset lists; pair_singles = every other slice, in processing order.  Built on first use.
    // row-dictionary product (fs_krylov.hip, k_dict_spmv): work items of <= 128 consecutive rows with one offset list, in processing
    // order, one 16-byte scalar load each: (first row, rows | edge << 16, first plan round, rounds); dict_plans: the run plans, 16
    // ints per round (dict_plan_round); dict_slots: doubles per class row in plan layout.  Built on first use (dict_structure_build).
    dbuf<int32_t> dict_items;     // [n_dict_items][4]
    dbuf<int32_t> dict_plans;
    int dict_slots = 0;
    int dict_run_len = 3;         // coefficient positions per run (2: CG2 spaces, where runs of three offsets are rare)
    // hints for dict_structure_build (set by fs_lattice.hip for the lattice-ordered shadow of a CG2 box space): rows repeat their
    // offset set with this period inside mesh lines of dict_line rows (0: unknown - segments are grown from nested sets), and a
    // line is one segment whose plan is the union of its rows' sets
    int dict_period = 1;
    int64_t dict_line = 0;
    int dict_runs = 8;            // runs per plan round (12: the lattice-ordered shadow - k_dict_spmv<.., 12>)
    int32_t box_a = 0;            // > 0: the whole space is ONE segment with the offset list of a Kuhn-split box, box_a rows per mesh line and
    int64_t box_b = 0;            // box_b rows per mesh plane (dict_structure_build): the marching-window product applies (fs_box.h)
    int lat_ny = 0, lat_nz = 0;   // lattice-ordered shadow: lines per plane, planes (row = x + dict_line * (y + lat_ny * z)); 0: not one
    // the solver's lattice-ordered shadow of a scalar CG2 space on a uniform box (fs_lattice.hip): 0 not looked at yet, 1 built,
    // -1 does not apply
    struct fs_lattice_shadow* lattice = nullptr;
    int lattice_state = 0;
    ~fs_space_s();
    int64_t n_dict_items = -1;    // -1: not built yet, 0: the pattern does not lend itself to the form
    dbuf<int32_t> pair_list;      // [2 * n_pairs]
    dbuf<int32_t> pair_singles;   // [n_pair_singles]
    int64_t n_pairs = -1, n_pair_singles = 0;      // -1: not built yet
    dbuf<int32_t> slots;          // [16][nc] SELL entry index of (a,b) of each cell, -1 = not owned (vector spaces)
    // two-pass assembly of block spaces with many dofs per cell (Taylor-Hood): element matrices are written to
    // elem_buf [cell*nd*nd + ab][bs*bs] and summed per stored block through the inverse of the slot table
    // (gmap_src[gmap_ptr[e] .. gmap_ptr[e+1]) = sources of entry e, ascending) - built on first use
    dbuf<int32_t> gmap_ptr, gmap_src;
    dbuf<double> elem_buf;
    // row-gather assembly tables (scalar spaces): the (cell, local vertex) incidences of every owned
    // row, SELL-64 laid out like the matrix; inc_pos packs the 4 in-row positions of the cell's vertices
    int64_t inc_entries = 0;
    int inc_max = 0;              // most incidences of any row
    dbuf<int64_t> inc_slice_ptr;  // [n_slices+1]
    dbuf<int32_t> inc_cell;       // [inc_entries] cell*4 + local vertex, -1 = padding
    dbuf<uint32_t> inc_pos;       // [pos_words][inc_entries]: ndof_cell x uint8 in-row positions, 4 per word
    int pos_words = 1;
    fs_halo_plan halo;
    // Dirichlet scratch kept across calls (re-assembly every time step must not hipMalloc)
    dbuf<uint8_t> bc_flag;        // [n_dofs_local]
    dbuf<double> bc_g;            // [n_dofs_local]
    dbuf<int32_t> bc_idx;         // [n_dofs_local] index of the last list entry naming the dof
};

struct fs_matrix_s {
    const uint64_t serial = fs_next_serial();
    fs_space_s* space = nullptr;
    int bs = 1;                   // block size (ncomp)
    bool taylor_hood = false;     // bs = 4 values written by fs_assemble_navier_stokes: pressure only on vertex nodes
    dbuf<double> val;             // [sell_entries * bs*bs]; block entry e, (i,j) at ((i*bs+j)*sell_entries + e)
};

struct fs_vector_s {
    dbuf<double> d;
};

__host__ __device__ static inline int32_t fs_col_decode(int32_t c) { return c < 0 ? ~c : c; }

// grid for a grid-stride elementwise / per-row kernel
static inline int fs_grid_for(int64_t work_items, int per_block = FS_BLOCK, int cap = 2048) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ---- cross-TU internals ------------------------------------------------------------------
// fs_lattice.hip
struct fs_lattice_shadow;
void fs_lattice_release(fs_lattice_shadow* L);
int fs_lattice_get(fs_space_s* sp, fs_lattice_shadow** out);
int fs_lattice_enter(fs_lattice_shadow* L, fs_matrix_s* A, const fs_vector_s* b, const fs_vector_s* x, bool use_guess, fs_matrix_s** A_sh,
                     fs_vector_s** b_sh, fs_vector_s** x_sh);
int fs_lattice_leave(fs_lattice_shadow* L, const fs_space_s* sp, fs_vector_s* x);
// fs_symbolic.hip: SELL-64 / DIA storage of a space from its CSR pattern (rowptr, colidx, n_nodes_owned, n_nodes_local set)
int fs_space_build_storage(fs_space_s* sp, hipStream_t s);
int fs_scan_exclusive_i32(const int32_t* in, int32_t* out, int64_t count, hipStream_t s);
// fs_assemble.hip: box meshes snap their edge vectors to the grid spacing (fs_set_option "box_snap", FS_BOX_SNAP=0: off)
void fs_set_box_snap(bool on);
void fs_set_box_assembly(bool on);   // fs_assemble.hip: the geometry-free P1 assembly of box meshes (option "box_assembly")
// RCCL (fs_comm.hip): in-stream collectives on device buffers; no-ops on one rank.
int fs_comm_allreduce_dev(double* d_inout, int n, hipStream_t s);
// sums over all ranks of the per-workgroup partials [nv][npart] (the k_sum_partials order) -> out[nv]: one kernel when the
// peer-to-peer all-reduce is on, k_sum_partials + ncclAllReduce otherwise
int fs_comm_sum_allreduce_dev(const double* partials, int npart, int nv, double* out, hipStream_t s);
int fs_p2p_allreduce_dev(const double* partials, int npart, double* inout, int nv, hipStream_t s);
int fs_p2p_reduce_enabled();
// fused peer-to-peer CG iteration (fs_krylov.hip): the structures the kernels take, with the sequence numbers advanced
struct fs_p2p_sendrows; struct fs_p2p_rowsred;
bool fs_p2p_fusable(const fs_space_s* space);                                   // peer-to-peer halo and all-reduce on
// arguments of k_cg_p2p_exchange for this space (nothing is advanced on the host: the sequence numbers live on the device)
int fs_p2p_exchange_args(fs_space_s* space, const double* partials, int npart, double* sums_out, fs_p2p_rowsred* red, fs_p2p_sendrows* snd);
int fs_p2p_check(hipStream_t s);
void fs_p2p_reduce_teardown();
void fs_comm_host_time(double* allreduce_us, long* allreduce_calls, double* halo_us, long* halo_calls, bool reset);
// inverse of the slot table: sources (cell*nd*nd + ab) of every stored block, ascending (gmap_ptr / gmap_src)
int fs_space_build_gather_map(fs_space_s* space, hipStream_t s);
int fs_halo_exchange_dev(fs_space_s* space, double* d_vec, hipStream_t s);
// the same in two halves: begin = pack + send/recv on the communication stream, end = compute stream waits for it
int fs_halo_begin_dev(fs_space_s* space, double* d_vec, hipStream_t s);
int fs_halo_end_dev(fs_space_s* space, hipStream_t s);
// the communication stream of the space's halo plan (created on first use): collectives that are to overlap with the
// compute stream are issued there, in the same order on every rank
int fs_halo_comm_stream(fs_space_s* space, hipStream_t* out);
// fs_krylov.hip: the row dictionary of A's current values for the products that follow (fs_spmv_dev), and its end
// each asks the runtime for the attributes of one of the file's kernels: that loads the file's code object (fs_init)
void fs_symbolic_preload();
void fs_assemble_preload();
void fs_krylov_preload();
void fs_amg_preload();
void fs_saddle_preload();
void fs_comm_preload();
int fs_dict_begin(fs_matrix_s* A, hipStream_t s);
void fs_dict_end();
int fs_dict_classes();
void fs_amg_set_coarse_fp32(int on);   // fs_amg.hip: hierarchies built from now on keep their coarse / transfer operators in fp32 (1) or fp64 (0)
// fs_krylov.hip: bare y = A x on the library stream, no halo exchange, no synchronisation.
int fs_spmv_dev(fs_matrix_s* A, const double* x, double* y, hipStream_t s);
// fs_amg.hip: z = M r (one V-cycle) on device pointers, no synchronisation.
struct fs_amg_s;
int fs_amg_apply_dev(fs_amg_s* amg, const double* r, double* z, hipStream_t s);
int64_t fs_amg_rows(const fs_amg_s* amg);      // scalar rows of the finest level

// fs_saddle.hip: the law a Taylor-Hood space carries, else kind 1 from (p_ref, exponent) > 0, else Newtonian
fs_visc_dev fs_space_viscosity(const fs_space_s* sp, double legacy_pref, double legacy_exp);
