// SpMV and Krylov solver of libfsamd.so (gfx950, fp64).
//
// Stands in for PETSc MatMult / VecDot / VecAXPY / KSPCG + PCJACOBI behind
// PETScKrylovSolver("cg", pc).solve(x, b)  (FenicsSolver/SolverBase.py:663-670) and the
// Krylov option of LinearVariationalSolver (SolverBase.py:608-612).
//
// Matrix layout: SELL-64.  One wavefront owns one slice of 64 consecutive rows; entry k
// of the 64 rows is contiguous in memory, so every wave-level load of values (8 B/lane)
// and column indices (4 B/lane) is one fully coalesced 512-B / 256-B transaction and the
// row sum needs no cross-lane reduction.  The kernel is HBM-bound (0.17 flop/B):
// algorithmic bytes per SpMV = nnz*(8+4) + n*(4+8+8)  (SURVEY.md section 8d).
//
// CG is the single-reduction (Chronopoulos-Gear) recurrence: per iteration ONE fused
// SpMV+3-dots kernel, one 1-workgroup partial-sum kernel (+ one 3-double all-reduce on
// >1 GPU) and ONE fused vector-update kernel.  alpha/beta/convergence live on the
// device; the host only polls a status word every `batch` iterations, two batches in
// flight, so the stream never drains.
#include "fs_common.h"
#include "fs_kernels.h"
#include <functional>
#include <chrono>
#include <string>
#include <unordered_map>
#include <math.h>
#include <stdlib.h>

#ifndef FS_BLOCK_ROUND
#define FS_BLOCK_ROUND 3      // 3x3 blocks per round of the vector-space product (measured: see DESIGN.md section 3)
#endif

#include "fs_krylov_stream.inc"      // the streaming products: hybrid SELL-64 / DIA kernels (k_sell_spmv, k_dia_pair_spmv)

#include "fs_krylov_dict.inc"      // the row-dictionary form of scalar DIA operators: plans, class tables (k_dict_insert / compact / finish), the work-item product k_dict_spmv

#define BOX_DC_AUX 2      // (fs_box.h: dot weights and class numbers are read once - streamed past the caches)
#include "fs_box.h"                // the marching-window product of P1 box operators (k_box_spmv) and its launch planning
#include "fs_latmarch.h"            // the marching-window product of CG2 box operators in lattice order (lm_march) and its tables
#include "fs_krylov_lattice.inc"      // the tile product of a lattice-ordered CG2 box operator (k_lattice_spmv), its tables, k_lat_march

#include "fs_krylov_dict3.inc"      // the row-dictionary product of 3 x 3 block rows (k_dict_spmv3)

struct row_dict {
    dbuf<uint16_t> cls, cls_slot;
    dbuf<double> values, slot_vals;
    dbuf<unsigned long long> keys;
    dbuf<int32_t> slot2cls, nnz;
    dbuf<int> info;
    int ncls = 0, S = 0, C = 0, bs = 1;     // classes, doubles per class row, most classes of any item, block size of the matrix
    const double* built_for = nullptr;      // the value array the classes describe (nullptr: plain form in use)
    uint64_t space_serial = 0;              // ... of this space
    uint64_t matrix_serial = 0;             // ... of this matrix (addresses are handed out again after a free: the pointer alone is no identity)
    uint64_t gave_up_on = 0;                // serial of a matrix whose rows were not repetitive: not tried again
    int64_t n_built = 0, n_failed = 0;      // statistics (fs_krylov_stats)
    // The tables of the last successful build (cls_slot, slot2cls, values, nnz) outlive the call they were built in: the next call
    // on the same space first has EVERY row of its matrix compared with the row's old class, bit for bit (k_dict_finish alone,
    // one pass over the values instead of three) - the matrix of a steady problem solved again, of a transient one with a constant
    // step.  A matrix that differs anywhere fails the comparison and is described from scratch; after a failure the next
    // attempts are skipped (1, 2, 4 .. 64 calls: Newton and Picard loops change their matrix every time).
    uint64_t tables_space = 0;              // serial of the space the tables describe (0: none)
    int tables_bs = 0, tables_S = 0, tables_ncls = 0;
    int reuse_backoff = 0, reuse_skip = 0;
    int64_t n_reused = 0;
    bool kept = false;                      // the tables in use were kept from the previous call
};
static row_dict g_dict;
// The dictionary describes the values of ONE call (a solve, fs_spmv_dictionary): whoever builds it drops it on the way out, so that no
// later product - another matrix whose values land on a freed address, the same matrix re-assembled - can meet a stale class table.
struct row_dict_scope {
    row_dict_scope() { g_dict.built_for = nullptr; }
    ~row_dict_scope() { g_dict.built_for = nullptr; }
};

#include "fs_krylov_iter.inc"      // the Krylov iterations' device side: CG / BiCGStab / pipelined CG update kernels, the one-launch iteration k_dict_cg_iter, the peer-to-peer exchange kernel

#include "fs_krylov_boxiter.inc"   // the one-launch CG iteration of P1 box operators in marching-window form (k_box_cg_iter)

// ---- host side --------------------------------------------------------------------------------
// tunables (fs_set_option): persistent grid size and row-loop unroll of the SpMV
static int g_spmv_blocks = 1024;
static int g_spmv_unroll = 4;
static bool g_spmv_blocks_pinned = false, g_spmv_unroll_pinned = false;
static int g_spmv_unroll4 = 2;   // 4x4-block matrices (Taylor-Hood)
// Scalar CG2 operators on uniform boxes solved in the lattice order of the half grid (fs_lattice.hip): automatic from FS_LATTICE_MIN_ROWS rows on (option "lattice_order" = -1; 0 = never, 1 = wherever the order exists) - measured on
// BASELINE configs[3] (9.94 M rows, tools/probes/p2_lattice_probe.py, round 5) the product takes 222 us with rounds of 8 runs, 272
// with rounds of 12 (139 VGPRs, three waves per SIMD), 240 with the items regrouped by class, against 191 us in the space's own
// numbering.  The rounds per work item do drop (3.4 -> 2.5 -> 1.5) but the time does not follow them: per row pair both orders
// spend the same 54 coefficient positions (fma + LDS read each) on 29 stored entries - in lattice order because a line's plan is the
// UNION of its two alternating row patterns (an x-edge row of 27 entries rides the 26 runs of its vertex neighbours).  The kernel
// is bound by those instructions, not by dependent rounds.  Option "lattice_order" / FS_LATTICE=1 turn it on.
static int g_lat_check = getenv("FS_LATTICE_CHECK") && getenv("FS_LATTICE_CHECK")[0] == '1' ? 1 : 0;      // option "lattice_check"
static int g_lat_march = getenv("FS_LATTICE_MARCH") ? atoi(getenv("FS_LATTICE_MARCH")) : 1;       // option "lattice_march": k_lat_march (fs_latmarch.h) where its tables can be built; 0: the tile product
static int g_lattice = getenv("FS_LATTICE") ? (getenv("FS_LATTICE")[0] == '1' ? 1 : (getenv("FS_LATTICE")[0] == '0' ? 0 : -1)) : -1;
constexpr int64_t FS_LATTICE_MIN_ROWS = 270000;     // automatic (-1): from here on (us per iteration, lattice order against the space's: 250 k rows 29.0 / 26.2, 275 k: 26.4 / 34.9,
                                                    // 300 k: 29.5 / 35.5, 1.03 M: 52.6 / 98.8 with the first tile kernel; it was 400 000 until the tile product got to 120 us at 10 M rows)
static inline bool bs_is_scalar_cg2(const fs_matrix_s* A) { return A->bs == 1 && A->space->degree == 2 && A->space->ncomp == 1; }
// k_cg_update_scaled with non-temporal accesses (vectors larger than the caches): which of its stores are ORDINARY ones all the same -
// bit 0: r, 1: p, 2: s, 3: x.  Measured (round 5, same box, A/B twice; tools/probes/exp_update_r_plain.sh), us per iteration at configs[3] /
// the 10 M-DOF P1 cube / its streaming form: none 237 - 240 / 177 - 181 / 394 - 397; r: 228 - 236 / 169 - 174 / 365 - 389; r + p: 241 / 183 / 400;
// r + s: 242 / 184 / 396; r + x: 245 / 180 / 389; all four: 258 / 194 / 427.  The residual is what the next product reads: default 1.
static const int g_upd_r_plain = getenv("FS_UPDATE_R_PLAIN") ? atoi(getenv("FS_UPDATE_R_PLAIN")) : 1;
static int g_cg_batch = 32;
// one-launch iteration on one GPU: launches go out g_cg_sub at a time (one hipGraph) whenever the device - its progress is read from
// pinned memory the kernel writes (krylov_ws::h_mirror) - has fewer than g_cg_ahead of them left to do; g_cg_mirror = 0: the batches
// of g_cg_batch with the status word copied back behind each (round 4)
static int g_cg_sub = 16, g_cg_ahead = 6, g_cg_mirror = 1;      // (tools/probes/cg_tail_probe.py: 6.80 -> 6.44 ms per solve at 1 M rows)
static int g_cg_fuse_sums = 1;
static int g_cg_graph = -1;      // -1: automatic (graphs, unless a profiler's tool library is in the process)
static int g_update_blocks = 1024;  // (round 3, with the 16 us row-dictionary product at 1 M rows: 256 / 512 / 768 / 1024 / 2048 workgroups: 10.59 / 10.42 / 10.20 / 10.12 / 11.22 ms per step; 10 M rows: flat)
static int g_cg_fused = -1;      // one launch per CG iteration on row-dictionary operators: -1 automatic, 0 never, 1 wherever it applies
static int g_row_dictionary = 1; // row-dictionary product where the operator allows it (0: always the streaming kernels)
// the marching-window product of P1 box operators (fs_box.h): option "box_spmv" (0: k_dict_spmv everywhere), from "box_min_rows" rows on
static int g_last_product_kind = 0;      // fs_last_product_kind()
// (OPT-IN: measured on MI355X, tools/probes/fused_iter_probe.py - 1 M rows 25.1 us per iteration two steps ahead / 29.5 one step ahead
// against 21.5 for k_dict_cg_iter; 3 M rows 50.2 against 50.8; 4.9 M rows 85 against 92 - 97 for the two launches; 10 M rows 208 against 170:
// a marching step costs 1.5 - 2 us whatever it moves - 31 global_load_lds issues per round on two loader waves, three dependent LDS round
// trips, a barrier -, which long marches over HBM-resident data amortise and the ten steps of a cache-resident launch do not)
static int g_box_iter = getenv("FS_BOX_ITER") ? atoi(getenv("FS_BOX_ITER")) : 0;                    // option "box_iter"
static int g_box_iter_ahead = getenv("FS_BOX_ITER_AHEAD") ? atoi(getenv("FS_BOX_ITER_AHEAD")) : 2;
static int64_t g_box_iter_min_rows = getenv("FS_BOX_ITER_MIN_ROWS") ? atoll(getenv("FS_BOX_ITER_MIN_ROWS")) : 400000;
static int g_box = getenv("FS_BOX_SPMV") ? atoi(getenv("FS_BOX_SPMV")) : 1;                    // option "box_spmv"
static int64_t g_box_min_rows = getenv("FS_BOX_MIN_ROWS") ? atoll(getenv("FS_BOX_MIN_ROWS")) : 1500000;

extern "C" int fs_set_option(const char* name, double value) {
    FS_REQUIRE(name, "fs_set_option: null name");
    if (!strcmp(name, "spmv_blocks")) {
        FS_REQUIRE(value >= 8 && value <= FS_MAX_PARTIAL_BLOCKS, "spmv_blocks must be in [8,%d]", FS_MAX_PARTIAL_BLOCKS);
        g_spmv_blocks = (int)value;
        g_spmv_blocks_pinned = true;
    } else if (!strcmp(name, "spmv_unroll")) {
        FS_REQUIRE(value == 2 || value == 4 || value == 8 || value == 16, "spmv_unroll must be 2, 4, 8 or 16");
        g_spmv_unroll = (int)value;
        g_spmv_unroll_pinned = true;
    } else if (!strcmp(name, "spmv_unroll4")) {
        FS_REQUIRE(value == 1 || value == 2 || value == 4, "spmv_unroll4 must be 1, 2 or 4");
        g_spmv_unroll4 = (int)value;
    } else if (!strcmp(name, "cg_fuse_sums")) {
        g_cg_fuse_sums = value != 0.0;
    } else if (!strcmp(name, "cg_graph")) {
        g_cg_graph = value < 0.0 ? -1 : (value != 0.0);
    } else if (!strcmp(name, "cg_fused")) {
        g_cg_fused = value < 0.0 ? -1 : (value != 0.0);
    } else if (!strcmp(name, "update_blocks")) {
        FS_REQUIRE(value >= 1 && value <= 65535, "update_blocks must be in [1,65535]");
        g_update_blocks = (int)value;
    } else if (!strcmp(name, "box_iter")) {
        g_box_iter = value != 0.0;
    } else if (!strcmp(name, "box_iter_min_rows")) {
        FS_REQUIRE(value >= 0, "box_iter_min_rows must be >= 0");
        g_box_iter_min_rows = (int64_t)value;
    } else if (!strcmp(name, "box_spmv")) {
        g_box = value != 0.0;
    } else if (!strcmp(name, "box_min_rows")) {
        FS_REQUIRE(value >= 0, "box_min_rows must be >= 0");
        g_box_min_rows = (int64_t)value;
    } else if (!strcmp(name, "row_dictionary")) {
        g_row_dictionary = value != 0.0;
    } else if (!strcmp(name, "box_snap")) {
        fs_set_box_snap(value != 0.0);
    } else if (!strcmp(name, "box_assembly")) {
        fs_set_box_assembly(value != 0.0);
    } else if (!strcmp(name, "cg_sub")) {
        FS_REQUIRE(value >= 2 && value <= 256 && ((int)value & 1) == 0, "cg_sub must be an even number in [2,256]");
        g_cg_sub = (int)value;
    } else if (!strcmp(name, "cg_ahead")) {
        FS_REQUIRE(value >= 1 && value <= 4096, "cg_ahead must be in [1,4096]");
        g_cg_ahead = (int)value;
    } else if (!strcmp(name, "lattice_check")) {
        g_lat_check = value != 0.0 ? 1 : 0;
    } else if (!strcmp(name, "lattice_march")) {
        if (g_lat_march != (value != 0.0 ? 1 : 0)) g_lat.tables_ok = false;       // (the tables of either form are made with the lists: lat_prepare)
        g_lat_march = value != 0.0 ? 1 : 0;
    } else if (!strcmp(name, "lattice_order")) {
        g_lattice = value < 0.0 ? -1 : (value != 0.0 ? 1 : 0);
    } else if (!strcmp(name, "cg_mirror")) {
        g_cg_mirror = value != 0.0 ? 1 : 0;
    } else if (!strcmp(name, "amg_coarse_fp32")) {
        fs_amg_set_coarse_fp32(value != 0.0);
    } else if (!strcmp(name, "cg_batch")) {
        FS_REQUIRE(value >= 1 && value <= 4096, "cg_batch must be in [1,4096]");
        g_cg_batch = (int)value;
    } else {
        fs_set_error("fs_set_option: unknown option '%s'", name);
        return FS_ERR_INVALID;
    }
    return FS_OK;
}

// Launch shape, unless fs_set_option pinned it (tools/tune_spmv.py + bench.py on MI355X, round 1): 1024 workgroups
// with 4-entry rounds for the 1 M-DOF class that lives in the Infinity Cache (more workgroups make the SpMV 1 us
// faster but the update kernel, which re-reduces the per-workgroup dot partials, 3 us slower); 512 workgroups
// with 16-entry rounds for HBM-resident sizes (-3 % solve time at 10 M DOF).
static int spmv_blocks_for(int64_t n_slices) {
    if (g_spmv_blocks_pinned) return g_spmv_blocks;
    return n_slices <= 32768 ? 1024 : 512;
}
static int spmv_unroll_for(int64_t n_slices) {
    if (g_spmv_unroll_pinned) return g_spmv_unroll;
    return n_slices <= 32768 ? 4 : 16;
}
// the matrix does not stay in the 256 MB Infinity Cache between two products
static bool spmv_nontemporal(const fs_space_s* sp, int bs) {
    static const char* e = getenv("FS_SPMV_NT");       // 0 / 1 pins the choice (tools/tune_spmv.py)
    if (e) return e[0] == '1';
    return sp->sell_entries * (int64_t)bs * bs * 8 > (int64_t)192 << 20;
}
static int spmv_grid(int64_t n_slices, int64_t n_slices_matrix) {
    const int64_t n_chunks = (n_slices + 3) / 4;
    const int64_t blocks = spmv_blocks_for(n_slices_matrix);
    int64_t g = n_chunks < blocks ? n_chunks : blocks;
    g = (g + 7) & ~(int64_t)7;  // multiple of 8 for the XCD mapping
    return (int)g;
}

// Pair / single split of the slices of a scalar space for k_dia_pair_spmv (host pass over four small arrays, once).
static int build_pair_lists(fs_space_s* sp, hipStream_t s) {
    const int64_t ns = sp->n_slices;
    std::vector<int32_t> dp((size_t)ns), off((size_t)std::max<int64_t>(sp->dia_off.n, 1)), order;
    std::vector<int64_t> ptr((size_t)ns + 1);
    FS_CHECK(sp->dia_ptr.download(dp.data(), ns, s));
    FS_CHECK(sp->dia_off.download(off.data(), sp->dia_off.n, s));
    FS_CHECK(sp->slice_ptr.download(ptr.data(), ns + 1, s));
    if (sp->slice_order.p) {
        order.resize((size_t)ns);
        FS_CHECK(sp->slice_order.download(order.data(), ns, s));
    }
    auto slice_at = [&](int64_t q) { return order.empty() ? (int32_t)q : order[(size_t)q]; };
    auto width = [&](int32_t sl) { return (int)((ptr[(size_t)sl + 1] - ptr[(size_t)sl]) >> 6); };
    auto pairable = [&](int32_t a, int32_t b) {
        if (dp[(size_t)a] < 0 || dp[(size_t)b] < 0) return false;
        if (off[(size_t)dp[(size_t)a]] < FS_SLICE || off[(size_t)dp[(size_t)b]] < FS_SLICE) return false;      // split slices: two lists each
        if ((int64_t)(a + 1) * FS_SLICE > sp->n_nodes_owned || (int64_t)(b + 1) * FS_SLICE > sp->n_nodes_owned) return false;
        const int w = width(a);
        if (w != width(b) || w == 0) return false;
        for (int k = 0; k < w; ++k)
            if (off[(size_t)dp[(size_t)a] + 1 + k] != off[(size_t)dp[(size_t)b] + 1 + k]) return false;
        return true;
    };
    // units are formed in the NATURAL numbering (rows of consecutive slices continue each other: same edge class on CG2
    // spaces, same mesh line on CG1) and then sorted into the processing order of the space by their first slice
    std::vector<int32_t> rank((size_t)ns);
    for (int64_t q = 0; q < ns; ++q) rank[(size_t)slice_at(q)] = (int32_t)q;
    struct unit { int32_t key, a, b; };
    std::vector<unit> units;
    for (int32_t sl = 0; sl < ns;) {
        if (sl + 1 < ns && pairable(sl, sl + 1)) { units.push_back({rank[(size_t)sl], sl, sl + 1}); sl += 2; }
        else { units.push_back({rank[(size_t)sl], sl, -1}); sl += 1; }
    }
    std::sort(units.begin(), units.end(), [](const unit& u, const unit& v) { return u.key < v.key; });
    std::vector<int32_t> pairs, singles;
    for (const unit& u : units) {
        if (u.b >= 0) { pairs.push_back(u.a); pairs.push_back(u.b); }
        else singles.push_back(u.a);
    }
    sp->n_pairs = (int64_t)pairs.size() / 2;
    sp->n_pair_singles = (int64_t)singles.size();
    if (getenv("FS_SPACE_DEBUG")) fprintf(stderr, "[fs_space] two-rows-per-lane product: %lld pairs, %lld single slices\n", (long long)sp->n_pairs, (long long)sp->n_pair_singles);
    FS_CHECK(sp->pair_list.alloc(std::max<int64_t>((int64_t)pairs.size(), 1)));
    FS_CHECK(sp->pair_singles.alloc(std::max<int64_t>(sp->n_pair_singles, 1)));
    FS_CHECK(sp->pair_list.upload(pairs.data(), (int64_t)pairs.size(), s));
    FS_CHECK(sp->pair_singles.upload(singles.data(), sp->n_pair_singles, s));
    return FS_OK;
}
// The pair kernel pays a second (small) launch for the unpaired slices: used where launches are long (matrix larger than
// the caches).  FS_SPMV_PAIRS = 0 / 1 pins the choice.
static bool spmv_use_pairs(const fs_space_s* sp, int bs) {
    if (bs != 1) return false;
    static const char* e = getenv("FS_SPMV_PAIRS");
    if (e) return e[0] == '1';
    return sp->n_slices > 32768;
}
static int spmv_pair_grid(const fs_space_s* sp) {
    const int64_t n_chunks = (sp->n_pairs + 3) / 4;
    static const int env_blocks = getenv("FS_PAIR_BLOCKS") ? atoi(getenv("FS_PAIR_BLOCKS")) : 0;
    int64_t g = std::min<int64_t>(n_chunks, env_blocks > 0 ? env_blocks : spmv_blocks_for(sp->n_slices));
    g = (g + 7) & ~(int64_t)7;
    return (int)std::max<int64_t>(g, 8);
}


static int spmv_partials_unsplit(const fs_space_s* sp, int bs);
static int dict_map_xcd();
// ---- structure of the row-dictionary product: segments, work items, run plans (once per space) ---------------------------------
// chg[r] = 1: the (col - row) offset set of row r differs from row r - 1's (compared through a 64-bit hash of both: a collision
// merges two rows into one segment whose plan then misses an entry - caught by the verification of every row in dict_build)
// period, line (fs_space_s::dict_period / dict_line): the rows of a mesh line of `line` rows repeat their sets with that period - a
// row is compared with the row `period` before it, and the first `period` rows of every line count as changes.
// (round 6: the rows that change are APPENDED to a list - count in list_n[0], the first `cap` of them stored, in any order - instead of
// a flag per row: the flags were n bytes to the host and a host loop over them, 9 of the 17 ms of the first solve of a process at 1 M rows)
__global__ void k_row_change(int64_t n_rows, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, int32_t* __restrict__ list,
                             unsigned long long* __restrict__ list_n, int64_t cap, int period = 1, int64_t line = 0) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_rows; r += stride) {
        unsigned long long h[2] = {0ull, 1ull};
        for (int w = 0; w < 2; ++w) {
            const int64_t q = r - w * period;
            if (q < 0) break;
            const int32_t s0 = rowptr[q], s1 = rowptr[q + 1];
            unsigned long long hh = 1469598103934665603ull ^ (unsigned long long)(s1 - s0);
            for (int32_t e = s0; e < s1; ++e) {
                hh = (hh ^ (unsigned long long)(unsigned)(colidx[e] - (int32_t)q)) * 1099511628211ull;
                hh ^= hh >> 29;
            }
            h[w] = hh;
        }
        const bool chg = r < period || h[0] != h[1] || (line > 0 && r % line < period);
        // one atomic per wave: the lanes with a changing row take consecutive places behind the wave's first
        const unsigned long long m = __ballot(chg);
        if (m) {
            const int lane = threadIdx.x & 63;
            unsigned long long base = 0;
            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(list_n, (unsigned long long)__popcll(m));
            base = __shfl(base, __ffsll((long long)m) - 1, 64);
            if (chg) {
                const unsigned long long at = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
                if ((int64_t)at < cap) list[at] = (int32_t)r;
            }
        }
    }
}
// offsets of the listed rows, concatenated (ptr = exclusive scan of their lengths)
__global__ void k_gather_offsets(int64_t n_list, const int32_t* __restrict__ list, const int64_t* __restrict__ ptr, const int32_t* __restrict__ rowptr,
                                 const int32_t* __restrict__ colidx, int32_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_list; i += stride) {
        const int32_t r = list[i];
        const int32_t s0 = rowptr[r], s1 = rowptr[r + 1];
        int32_t* __restrict__ o = out + ptr[i];
        for (int32_t e = s0; e < s1; ++e) o[e - s0] = colidx[e] - r;
    }
}
__global__ void k_gather_lengths(int64_t n_list, const int32_t* __restrict__ list, const int32_t* __restrict__ rowptr, int32_t* __restrict__ len) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_list; i += stride) len[i] = rowptr[list[i] + 1] - rowptr[list[i]];
}

// Segments, work items and run plans of the row-dictionary product (comment at k_dict_spmv) for the whole space and, on a
// decomposed space, for its interior / boundary slices.  One pass over the pattern on the device (which rows change the offset
// set), the few rows that do are looked at on the host.  sp->n_dict_items = 0: the pattern does not lend itself to the form.
static int dict_structure_build(fs_space_s* sp, hipStream_t s) {
    fs_halo_plan& h = sp->halo;
    const bool split = h.active && h.n_interior > 0;
    const bool need_space = sp->n_dict_items < 0, need_lists = split && (h.n_items_interior < 0 || h.n_items_boundary < 0);
    if (!need_space && !need_lists) return FS_OK;
    const int64_t n = sp->n_nodes_owned, ns = sp->n_slices, n_cols = sp->n_nodes_local;
    auto give_up = [&](const char* why) {
        if (getenv("FS_KRYLOV_DEBUG")) fprintf(stderr, "[fs_krylov] row-dictionary structure of space %llu: %s - not used\n", (unsigned long long)sp->serial, why);
        sp->n_dict_items = 0;
        h.n_items_interior = h.n_items_boundary = 0;
        return FS_OK;
    };
    if (n < 2 || !sp->rowptr.p || !sp->colidx.p) return give_up("no pattern");
    static const bool lap_on = getenv("FS_SOLVE_TIMING") != nullptr;
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!lap_on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[fs_krylov timing]     structure: %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - lap_t).count());
        lap_t = now;
    };
    // 1. rows whose offset set differs from the previous row's
    std::vector<int32_t> crow;
    int64_t nc = 0;
    {
        // The list goes straight into the library's pinned staging buffer when it fits (the kernel stores into host memory; the
        // count stays on the device - one returning atomic per wave over the bus would cost more than the whole pass): the first
        // device-to-host copy out of the freshly allocated list buffer took 7 - 9 ms of the first solve of a process (round 6,
        // tools/probes/first_step_probe.py with FS_COPY_TRACE=1; the same copy a second time: 0.04 ms).
        dbuf<unsigned long long> d_cnt;
        FS_CHECK(d_cnt.alloc(1));
        for (int pass = 0; pass < 2; ++pass) {
            const int64_t cap_host = (int64_t)(FS_STAGING_BYTES / sizeof(int32_t));
            void* st = pass == 0 ? fs_staging_lock() : nullptr;
            struct unlock { void* st; ~unlock() { if (st) fs_staging_unlock(); } } guard{st};
            dbuf<int32_t> d_list;
            const int64_t cap = st ? cap_host : n / 2 + 1;
            if (!st) FS_CHECK(d_list.alloc(cap));
            FS_CHECK(d_cnt.zero(s));
            // (the grid covers the rows once - whole waves, the ballot above needs every lane of a wave in the loop together)
            hipLaunchKernelGGL(k_row_change, dim3(fs_grid_for(n, FS_BLOCK, 65535)), dim3(FS_BLOCK), 0, s, n, sp->rowptr.p, sp->colidx.p,
                               st ? reinterpret_cast<int32_t*>(st) : d_list.p, d_cnt.p, cap, sp->dict_period, sp->dict_line);
            FS_KERNEL_CHECK();
            unsigned long long h_cnt = 0;
            FS_HIP(hipMemcpyAsync(&h_cnt, d_cnt.p, sizeof(h_cnt), hipMemcpyDeviceToHost, s));
            FS_HIP(hipStreamSynchronize(s));
            nc = (int64_t)h_cnt;
            if (nc * 2 > n) break;                   // an unstructured mesh: given up below
            if (nc > cap) continue;                  // more rows than the staging buffer holds: once more, into a device list
            crow.resize((size_t)nc);
            if (st) memcpy(crow.data(), st, (size_t)nc * sizeof(int32_t));
            else FS_CHECK(d_list.download(crow.data(), nc, s));
            std::sort(crow.begin(), crow.end());
            break;
        }
    }
    lap("changing rows");
    if (nc * 2 > n) return give_up("rows change their offset set too often");     // (an unstructured mesh)
    // 2. ... and their offset lists
    std::vector<int32_t> clen((size_t)nc), coff;
    std::vector<int64_t> cptr((size_t)nc + 1, 0);
    {
        dbuf<int32_t> d_list, d_len, d_off;
        dbuf<int64_t> d_ptr;
        FS_CHECK(d_list.alloc(nc));
        FS_CHECK(d_len.alloc(nc));
        FS_CHECK(d_list.upload(crow.data(), nc, s));
        hipLaunchKernelGGL(k_gather_lengths, dim3(fs_grid_for(nc)), dim3(FS_BLOCK), 0, s, nc, d_list.p, sp->rowptr.p, d_len.p);
        FS_CHECK(d_len.download(clen.data(), nc, s));
        for (int64_t i = 0; i < nc; ++i) cptr[(size_t)i + 1] = cptr[(size_t)i] + clen[(size_t)i];
        coff.resize((size_t)std::max<int64_t>(cptr[(size_t)nc], 1));
        FS_CHECK(d_ptr.alloc(nc + 1));
        FS_CHECK(d_ptr.upload(cptr.data(), nc + 1, s));
        FS_CHECK(d_off.alloc(std::max<int64_t>(cptr[(size_t)nc], 1)));
        hipLaunchKernelGGL(k_gather_offsets, dim3(fs_grid_for(nc)), dim3(FS_BLOCK), 0, s, nc, d_list.p, d_ptr.p, sp->rowptr.p, sp->colidx.p, d_off.p);
        FS_KERNEL_CHECK();
        FS_CHECK(d_off.download(coff.data(), cptr[(size_t)nc], s));
    }
    lap("their offset lists");
    // 3. segments: a row joins while its set is nested with the segment's list (which grows to the larger one)
    struct segment { int32_t first, end, list; };      // list = index of the change row whose offsets are the segment's list
    std::vector<segment> segs;
    if (sp->dict_line > 0) {
        // mesh lines of a known length whose rows alternate between sets (the lattice-ordered shadow of a CG2 box space): a line is
        // one segment, its list the UNION of the sets of its change rows - appended to the lists as one more
        const int64_t line = sp->dict_line;
        int64_t i = 0;
        std::vector<int32_t> uni, tmp;
        for (int64_t a = 0; a < n; a += line) {
            const int64_t e = std::min(a + line, n);
            uni.clear();
            for (; i < nc && crow[(size_t)i] < e; ++i) {
                tmp.clear();
                std::set_union(uni.begin(), uni.end(), coff.data() + cptr[(size_t)i], coff.data() + cptr[(size_t)i + 1], std::back_inserter(tmp));
                uni.swap(tmp);
            }
            const int32_t id = (int32_t)clen.size();
            coff.resize((size_t)cptr.back());          // (drop the padding element of an empty list array)
            coff.insert(coff.end(), uni.begin(), uni.end());
            clen.push_back((int32_t)uni.size());
            cptr.push_back(cptr.back() + (int64_t)uni.size());
            segs.push_back({(int32_t)a, (int32_t)e, id});
        }
    } else {
        auto set_of = [&](int64_t i) { return std::make_pair(coff.data() + cptr[(size_t)i], coff.data() + cptr[(size_t)i + 1]); };
        int64_t cur = 0;
        segs.push_back({0, 0, 0});
        for (int64_t i = 1; i < nc; ++i) {
            const auto M = set_of(cur), o = set_of(i);
            if (std::includes(M.first, M.second, o.first, o.second)) continue;
            if (std::includes(o.first, o.second, M.first, M.second)) { cur = i; segs.back().list = (int32_t)i; continue; }
            segs.back().end = crow[(size_t)i];
            segs.push_back({crow[(size_t)i], 0, (int32_t)i});
            cur = i;
        }
        segs.back().end = (int32_t)n;
        // A few segments whose lists are all contained in the longest one (a P1 box: the first mesh line, the rest of the first
        // plane, the first line of the second plane, everything else - the greedy pass above only ever grows a list by nesting)
        // are ONE segment with that list: a row has zero coefficients where it has no entry, as the boundary rows inside the bulk
        // segment already do, and the items whose loads could leave the vector are flagged `edge` below.  One plan for the whole
        // space = one coefficient layout for every class row (k_box_spmv relies on it).
        if (segs.size() > 1 && segs.size() <= 8) {
            size_t big = 0;
            for (size_t g = 1; g < segs.size(); ++g)
                if (clen[(size_t)segs[g].list] > clen[(size_t)segs[big].list]) big = g;
            const auto M = set_of(segs[big].list);
            bool nested = true;
            for (size_t g = 0; g < segs.size() && nested; ++g) {
                const auto o = set_of(segs[g].list);
                nested = std::includes(M.first, M.second, o.first, o.second);
            }
            if (nested) {
                const int32_t list = segs[big].list;
                segs.assign(1, segment{0, (int32_t)n, list});
            }
        }
    }
    const int NR = sp->dict_runs;       // runs per round: 8, or 12 for the lattice-ordered shadow of a CG2 box space
    // 4. run plans (identical lists share one) and items.  Runs of up to three consecutive offsets - or of up to two where longer
    // ones are rare (CG2: 0.5 % of the runs): a class row then has two coefficient positions per run instead of three
    int RL = 3;
    {
        int64_t n_runs3 = 0, n_long = 0;
        for (size_t g = 0; g < segs.size(); g += std::max<size_t>(segs.size() / 4096, 1)) {       // (a sample of the segments)
            const int32_t* o = coff.data() + cptr[(size_t)segs[g].list];
            const int w = clen[(size_t)segs[g].list];
            for (int k = 0; k < w;) {
                int len = 1;
                while (k + len < w && len < 3 && o[k + len] == o[k + len - 1] + 1) ++len;
                ++n_runs3;
                n_long += len == 3;
                k += len;
            }
        }
        if (n_long * 20 < n_runs3) RL = 2;
        static const char* rl_env = getenv("FS_DICT_RUN_LENGTH");
        if (rl_env && (rl_env[0] == '2' || rl_env[0] == '3')) RL = rl_env[0] - '0';
        if (NR == 12) RL = 3;       // (the only instantiation of the work-item product for rounds of twelve runs is k_dict_spmv<.., 3, 12>: ADVICE r5)
    }
    std::vector<dict_plan_round> rounds;
    struct plan_info { int32_t first, rounds, min_start, max_start; };
    std::vector<plan_info> seg_plan(segs.size());
    int max_rounds = 0;
    {
        std::unordered_map<std::string, int32_t> seen;      // offset list -> index into `infos`
        std::vector<plan_info> infos;
        for (size_t g = 0; g < segs.size(); ++g) {
            const int32_t* o = coff.data() + cptr[(size_t)segs[g].list];
            const int w = clen[(size_t)segs[g].list];
            std::string key(reinterpret_cast<const char*>(o), (size_t)w * sizeof(int32_t));
            auto f = seen.find(key);
            if (f != seen.end()) { seg_plan[g] = infos[(size_t)f->second]; continue; }
            plan_info pi = {(int32_t)rounds.size(), 0, 0, 0};
            dict_plan_round cur;
            memset(&cur, 0, sizeof(cur));
            int slot = 1;                               // slot 0 of round 0 is the z run
            for (int k = 0; k < w;) {
                int len = 1;
                while (k + len < w && len < RL && o[k + len] == o[k + len - 1] + 1) ++len;
                if (slot == NR) { rounds.push_back(cur); memset(&cur, 0, sizeof(cur)); slot = 0; }
                reinterpret_cast<int32_t*>(&cur)[slot] = o[k];                          // (dict_run_start / dict_run_len: NR starts, then NR lengths)
                (reinterpret_cast<uint8_t*>(&cur) + 4 * NR)[slot] = (uint8_t)len;
                pi.min_start = std::min(pi.min_start, o[k]);
                pi.max_start = std::max(pi.max_start, o[k]);
                ++slot;
                k += len;
            }
            rounds.push_back(cur);
            pi.rounds = (int32_t)rounds.size() - pi.first;
            max_rounds = std::max(max_rounds, (int)pi.rounds);
            seen.emplace(std::move(key), (int32_t)infos.size());
            infos.push_back(pi);
            seg_plan[g] = pi;
        }
        if (getenv("FS_KRYLOV_DEBUG") || getenv("FS_SPACE_DEBUG"))
            fprintf(stderr, "[fs_krylov] row-dictionary structure: %lld rows, %lld change their offset set, %zu segments, %zu distinct plans of runs <= %d, %zu rounds (longest plan %d)\n",
                    (long long)n, (long long)nc, segs.size(), infos.size(), RL, rounds.size(), max_rounds);
    }
    if (max_rounds > FS_DICT_MAX_ROUNDS) return give_up("a row has more runs of offsets than a plan holds");
    // processing order: by the position of the item's first row in the slice order of the space (an XCD then sweeps one slab of
    // the mesh for all node classes of a CG2 space, as the streaming kernels do)
    std::vector<int32_t> rank;
    if (sp->slice_order.p) {
        std::vector<int32_t> order((size_t)ns);
        FS_CHECK(sp->slice_order.download(order.data(), ns, s));
        rank.resize((size_t)ns);
        for (int64_t q = 0; q < ns; ++q) rank[(size_t)order[(size_t)q]] = (int32_t)q;
    }
    std::vector<uint8_t> bslice;        // slices with ghost columns (decomposed space)
    if (split) {
        bslice.assign((size_t)ns, 0);
        std::vector<int32_t> bl((size_t)h.n_boundary);
        if (h.n_boundary) FS_CHECK(h.boundary.download(bl.data(), h.n_boundary, s));
        for (int32_t sl : bl) bslice[(size_t)sl] = 1;
    }
    struct item { int64_t key; int32_t v[4]; uint8_t boundary; };
    std::vector<item> all;
    for (size_t g = 0; g < segs.size(); ++g) {
        const plan_info& pi = seg_plan[g];
        // (a segment longer than one item is cut at multiples of 126 rows - even rows: the 16-byte accesses of w and d are aligned
        // there; a shorter one - a mesh line of a CG2 space - is one item wherever it starts)
        // (a mesh line of known length - the lattice-ordered shadow -: equal pieces of an even number of rows, so that every lane's
        // first row is an even point of the line)
        const int32_t seg_len = segs[g].end - segs[g].first;
        const int32_t pieces = (seg_len + FS_DICT_ITEM_ROWS - 1) / FS_DICT_ITEM_ROWS;
        const int32_t piece = sp->dict_line > 0 ? (((seg_len + pieces - 1) / pieces + 1) & ~1) : 0;
        for (int32_t a = segs[g].first, e; a < segs[g].end; a = e) {
            e = segs[g].end - a <= FS_DICT_ITEM_ROWS ? segs[g].end : (a / FS_DICT_ITEM_ROWS + 1) * FS_DICT_ITEM_ROWS;
            if (piece > 0) e = std::min(a + piece, segs[g].end);
            // (all 64 lanes load, also those past the item's last row: in range means in range for 128 rows)
            const bool edge = (int64_t)a + pi.min_start < 0 || (int64_t)a + 127 + pi.max_start > n_cols - 1;      // (64 lanes x 2 values)
            item it;
            it.key = rank.empty() ? (int64_t)a : (int64_t)rank[(size_t)(a >> 6)] * FS_SLICE + (a & 63);
            it.v[0] = a; it.v[1] = (e - a) | ((int32_t)edge << 16); it.v[2] = pi.first; it.v[3] = pi.rounds;
            it.boundary = 0;
            if (split)
                for (int32_t sl = a >> 6; sl <= (e - 1) >> 6; ++sl) it.boundary |= bslice[(size_t)sl];
            all.push_back(it);
        }
    }
    if ((int64_t)all.size() * 8 > n) return give_up("segments of fewer than 8 rows");
    if (sp->dict_line > 0) {
        // Lattice order: consecutive mesh lines belong to different node classes, and a wave that takes them one after the other
        // copies the class rows of every item into its LDS region again (six rows of 864 bytes, one dependent round trip each:
        // measured 272 us per product at 10 M rows against 193 us in the space's own numbering).  Inside windows of 1024 items
        // (about two and a half mesh planes at n = 107) the items are taken plan by plan and piece by piece: the four consecutive
        // items of a wave are then the same piece of neighbouring lines of ONE class - same class rows, already in place.
        const int64_t line = sp->dict_line;
        constexpr size_t WINDOW = 1024;
        for (size_t w0 = 0; w0 < all.size(); w0 += WINDOW)
            std::sort(all.begin() + w0, all.begin() + std::min(w0 + WINDOW, all.size()), [line](const item& u, const item& v) {
                if (u.v[2] != v.v[2]) return u.v[2] < v.v[2];
                const int64_t pu = u.v[0] % line, pv = v.v[0] % line;
                return pu != pv ? pu < pv : u.v[0] < v.v[0];
            });
    }
    if (!rank.empty()) {
        std::sort(all.begin(), all.end(), [](const item& u, const item& v) { return u.key < v.key; });
        // ... and inside windows of 2048 items (about what one XCD has in flight) by row number again: the lines of ONE node class
        // of a CG2 space follow each other there, so that the items a wave takes in a row hold the same coefficient classes
        constexpr size_t WINDOW = 2048;
        for (size_t w0 = 0; w0 < all.size(); w0 += WINDOW)
            std::sort(all.begin() + w0, all.begin() + std::min(w0 + WINDOW, all.size()), [](const item& u, const item& v) { return u.v[0] < v.v[0]; });
    }
    auto upload_items = [&](dbuf<int32_t>& dst, int64_t& count, int which) {     // which: -1 all, 0 interior, 1 boundary
        std::vector<int32_t> flat;
        flat.reserve(all.size() * 4);
        for (const item& it : all)
            if (which < 0 || it.boundary == which) flat.insert(flat.end(), it.v, it.v + 4);
        count = (int64_t)flat.size() / 4;
        int rc = dst.alloc(std::max<int64_t>((int64_t)flat.size(), 4));
        if (rc == FS_OK && !flat.empty()) rc = dst.upload(flat.data(), (int64_t)flat.size(), s);
        return rc;
    };
    lap("segments, plans, items");
    const int64_t plan_ints = (int64_t)rounds.size() * 16;
    if (need_space) {
        // (the plans depend on the pattern only: a later build for new halo lists finds the same array in place)
        FS_CHECK(sp->dict_plans.alloc(std::max<int64_t>(plan_ints, 16)));
        FS_CHECK(sp->dict_plans.upload(reinterpret_cast<const int32_t*>(rounds.data()), plan_ints, s));
        sp->dict_slots = NR * RL * std::max(max_rounds, 1);
        sp->dict_run_len = RL;
        // ONE segment with the offset list of a Kuhn-split box (P1 on fs_mesh_create_box / BoxMesh, one GPU): the marching-window
        // product applies (fs_box.h, k_box_spmv)
        sp->box_a = 0; sp->box_b = 0;
        if (segs.size() == 1 && rounds.size() == 1 && NR == 8 && !h.active && n == n_cols) {
            box_geom bg;
            if (box_recognize(rounds[0].start, rounds[0].len, 8, n, RL, &bg)) { sp->box_a = bg.a; sp->box_b = bg.b; }
        }
        FS_CHECK(upload_items(sp->dict_items, sp->n_dict_items, -1));
    }
    if (need_lists) {
        FS_CHECK(upload_items(h.items_interior, h.n_items_interior, 0));
        FS_CHECK(upload_items(h.items_boundary, h.n_items_boundary, 1));
    }
    FS_HIP(hipStreamSynchronize(s));
    lap("uploads");
    if (getenv("FS_KRYLOV_DEBUG") || getenv("FS_SPACE_DEBUG"))
        fprintf(stderr, "[fs_krylov] row-dictionary work items: %lld for the space (%.1f rows each), interior %lld, boundary %lld; %d coefficient positions per class row\n",
                (long long)sp->n_dict_items, sp->n_dict_items ? (double)n / sp->n_dict_items : 0.0, (long long)h.n_items_interior,
                (long long)h.n_items_boundary, sp->dict_slots);
    return FS_OK;
}

// Try to describe `val` (the scalar DIA matrix the solver is about to multiply with) by row classes; leaves g_dict.built_for =
// val on success, nullptr otherwise.  One host synchronisation (16 bytes).  FS_SPMV_DICT=0 switches it off.
// raw, sc (scalar operators, one GPU; may be null): val = D^-1/2 raw D^-1/2 has NOT been written yet.  The comparison with the kept
// class table then walks raw and scales on the fly; only if that fails - or no table is kept - `materialize` writes val (k_scale_copy)
// before anything reads it.  On the kept path val stays unwritten: it is the KEY of the permission (built_for), and every product of
// the solve goes through the dictionary kernels (launch_spmv: whole-space launches of a space without a halo plan).
struct box_plan_s;
static const box_plan_s* box_plan_for(const fs_space_s* sp, int ncls);
static int dict_build_impl(fs_matrix_s* A, const double* val, hipStream_t s, const double* raw, const double* sc, const std::function<void()>& materialize) {
    row_dict& D = g_dict;
    D.built_for = nullptr;
    fs_space_s* sp = A->space;
    static const bool off = getenv("FS_SPMV_DICT") && getenv("FS_SPMV_DICT")[0] == '0';
    if (off || !g_row_dictionary || (A->bs != 1 && A->bs != 3) || sp->n_slices == 0 || sp->n_dia_slices != sp->n_slices || D.gave_up_on == A->serial) return FS_OK;
    // (block rows: finding and verifying the classes costs about 3 ms a solve and the item kernel has a latency floor - measured
    // on the AMG-PCG solve of the cantilever: 91 k DOF 6.5 -> 9.5 ms, 683 k DOF 14.0 -> 12.4 ms, 5.1 M DOF 100 -> 61 ms)
    static const int64_t min_nodes3 = getenv("FS_DICT3_MIN_NODES") ? atoll(getenv("FS_DICT3_MIN_NODES")) : 150000;      // (tests lower it)
    if (A->bs == 3 && sp->n_nodes_owned < min_nodes3) return FS_OK;
    static const bool lap_on = getenv("FS_SOLVE_TIMING") != nullptr;
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!lap_on) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[fs_krylov timing]   dictionary: %-24s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - lap_t).count());
        lap_t = now;
    };
    FS_CHECK(dict_structure_build(sp, s));
    lap("structure");
    if (sp->n_dict_items <= 0) return FS_OK;
    const int nq = A->bs * A->bs;               // values per stored entry (3 x 3 blocks of a vector space: the class rows are [position][9])
    const int S = sp->dict_slots * nq;
    const int64_t padded = sp->n_nodes_owned + 2 * FS_DICT_ITEM_ROWS;
    if (D.cls.n < padded) { D.tables_space = 0; FS_CHECK(D.cls.alloc(padded)); FS_CHECK(D.cls_slot.alloc(padded)); }
    if (!D.keys.p) {
        FS_CHECK(D.keys.alloc(FS_DICT_CAP));
        FS_CHECK(D.slot2cls.alloc(FS_DICT_CAP));
        FS_CHECK(D.nnz.alloc(FS_DICT_MAX));
        FS_CHECK(D.info.alloc(4));
    }
    if (D.slot_vals.n < (int64_t)FS_DICT_CAP * S) { D.tables_space = 0; FS_CHECK(D.slot_vals.alloc((int64_t)FS_DICT_CAP * S)); FS_CHECK(D.values.alloc((int64_t)FS_DICT_MAX * S)); }
    const int4* items = reinterpret_cast<const int4*>(sp->dict_items.p);
    const dict_plan_round* plans = reinterpret_cast<const dict_plan_round*>(sp->dict_plans.p);
    const int grid = fs_grid_for(sp->n_dict_items * 64, FS_BLOCK, 4096);
    const auto usable = [&](const int* h, int ncls) {
        // (worth it only where rows really repeat: at most one class per 16 rows)
        return h[1] == 0 && h[2] == 0 && h[3] > 0 && (int64_t)(FS_BLOCK / 64) * h[3] * S * (int64_t)sizeof(double) <= FS_DICT_LDS_BYTES &&
               ncls > 0 && ncls <= FS_DICT_MAX && (int64_t)ncls * 16 <= sp->n_nodes_owned;
    };
    const auto adopt = [&](const int* h, int ncls) {
        D.ncls = ncls;
        D.S = S;
        D.C = h[3];
        D.bs = A->bs;
        D.built_for = val;
        D.space_serial = sp->serial;
        D.matrix_serial = A->serial;
        if (A->bs == 1 && sp->dict_runs == 8 && sp->dict_run_len == 3) (void)box_plan_for(sp, ncls);     // (the launch plan of k_box_spmv: made here, outside any capture)
    };
    static const bool no_reuse = getenv("FS_DICT_REUSE") && getenv("FS_DICT_REUSE")[0] == '0';
    if (!no_reuse && D.tables_space == sp->serial && D.tables_bs == A->bs && D.tables_S == S && D.tables_ncls > 0) {
        if (D.reuse_skip > 0) --D.reuse_skip;
        else {
            FS_CHECK(D.info.zero(s));
            hipLaunchKernelGGL(k_dict_finish, dim3(grid), dim3(FS_BLOCK), 0, s, sp->n_dict_items, items, plans, sp->slice_ptr.p, sp->dia_ptr.p, sp->dia_off.p,
                               raw ? raw : val, nq, sp->sell_entries, S, sp->dict_run_len, D.slot2cls.p, D.values.p, D.nnz.p, D.cls_slot.p, D.cls.p, D.info.p,
                               raw ? sc : nullptr, sp->dict_runs);
            FS_KERNEL_CHECK();
            int h[4] = {0, 0, 0, 0};
            FS_CHECK(D.info.download(h, 4, s));
            const bool same = usable(h, D.tables_ncls);
            if (getenv("FS_KRYLOV_DEBUG"))
                fprintf(stderr, "[fs_krylov] row dictionary: the %d classes of the last call against the %lld rows of this one: %d mismatches -> %s\n",
                        D.tables_ncls, (long long)sp->n_nodes_owned, h[2], same ? "kept" : "built again");
            if (same) {
                adopt(h, D.tables_ncls);
                D.kept = true;
                D.reuse_backoff = 0;
                ++D.n_reused;
                return FS_OK;
            }
            D.reuse_backoff = D.reuse_backoff ? (D.reuse_backoff < 64 ? 2 * D.reuse_backoff : 64) : 1;
            D.reuse_skip = D.reuse_backoff;
        }
    }
    D.tables_space = 0;
    lap("tables allocated");
    materialize();                  // the classes are found from the scaled values themselves
    lap("scaled copy");
    FS_CHECK(D.keys.zero(s));
    FS_CHECK(D.info.zero(s));
    FS_HIP(hipMemsetAsync(D.slot_vals.p, 0, (size_t)FS_DICT_CAP * S * sizeof(double), s));
    hipLaunchKernelGGL(k_dict_insert, dim3(grid), dim3(FS_BLOCK), 0, s, sp->n_dict_items, items, plans, sp->slice_ptr.p, sp->dia_ptr.p, sp->dia_off.p,
                       val, nq, sp->sell_entries, S, sp->dict_run_len, D.keys.p, D.keys.p, D.slot_vals.p, D.cls_slot.p, D.info.p, sp->dict_runs);
    hipLaunchKernelGGL(k_dict_compact, dim3(1), dim3(1024), 0, s, D.keys.p, D.slot_vals.p, S, D.slot2cls.p, D.values.p, D.nnz.p);
    hipLaunchKernelGGL(k_dict_finish, dim3(grid), dim3(FS_BLOCK), 0, s, sp->n_dict_items, items, plans, sp->slice_ptr.p, sp->dia_ptr.p, sp->dia_off.p,
                       val, nq, sp->sell_entries, S, sp->dict_run_len, D.slot2cls.p, D.values.p, D.nnz.p, D.cls_slot.p, D.cls.p, D.info.p, nullptr, sp->dict_runs);
    FS_KERNEL_CHECK();
    int h[4] = {0, 0, 0, 0};
    FS_CHECK(D.info.download(h, 4, s));
    lap("classes found + verified");
    const bool ok = usable(h, h[0]);
    if (getenv("FS_KRYLOV_DEBUG"))
        fprintf(stderr, "[fs_krylov] row dictionary: %d distinct rows of %d positions among %lld, %d mismatches, at most %d classes per item -> %s\n", h[0], S,
                (long long)sp->n_nodes_owned, h[2], h[3], ok ? "compressed product" : "plain product");
    if (!ok) {
        D.gave_up_on = A->serial;
        ++D.n_failed;
        return FS_OK;
    }
    adopt(h, h[0]);
    D.kept = false;
    D.tables_space = sp->serial;
    D.tables_bs = A->bs;
    D.tables_S = S;
    D.tables_ncls = h[0];
    ++D.n_built;
    return FS_OK;
}

template <int DOTS>
static void launch_spmv(fs_matrix_s* A, const double* x, double* y, const double* rvec, double* partials,
                        int* status, hipStream_t s, const double* val_override = nullptr, const int32_t* list = nullptr, int64_t n_list = 0,
                        int part_base = 0, int part_stride = 0, int bump = 1);
__global__ void k_lat_fill(int64_t n, double* __restrict__ v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long h = ((unsigned long long)i + 1ull) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 31;
        v[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}
__global__ void k_lat_count_diff(int64_t n, const double* __restrict__ a, const double* __restrict__ b, int* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int c = 0;
    for (; i < n; i += stride) c += a[i] != b[i];
    if (c) atomicAdd(out, c);
}

// ---- the marching-window product of a lattice-ordered CG2 box operator (fs_latmarch.h, k_lat_march) -----------------------------------
// One workgroup per CU: eleven line waves - a wave a line of the patch - and a loader (three waves a SIMD); the rings (window, step lists, dot weights:
// three slots each, the loaders two steps ahead) take up to 150 KB of LDS.  A wave takes one or two 64-pair pieces of a line (RP), lines
// of more than 128 pairs several waves.
// Its tables (coefficient rows by stencil position, the row numbers of the lines, the step lists) are built - and every class, every
// line checked against what the kernel assumes - by lat_prepare behind the lists of the tile product; an operator that does not fit
// (an entry outside its parity's stencil, a line whose interior rows are of several classes) keeps the tile product.
struct lat_march_s {
    dbuf<int32_t> used;         // [ncls 8]
    dbuf<double> coefS;         // [ncls 8 + 1][LM_CS]; the last row: zeros
    dbuf<uint32_t> lc;          // [NY][LZ]: row numbers of the lines
    dbuf<int32_t> tab, rep, num;            // the hash table over the tuples, a tuple's first-come twin, its list number
    dbuf<int32_t> sl_line;      // [NS2][NYP]: list numbers
    dbuf<double> SL;            // [lists][LM_SL]
    int n_lists = 0;
    lm_geom g = {};
    size_t lds = 0;
    bool ok = false;            // the tables describe the dictionary g_lat's lists were made for
};
static lat_march_s g_lm;
template <int DOTS>
static void lm_set_lds_attribute() {
    (void)hipFuncSetAttribute((const void*)k_lat_march<DOTS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 << 10);
    (void)hipFuncSetAttribute((const void*)k_lat_march<DOTS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 << 10);
}
static void lm_prepare_kernels() {
    static bool done = false;
    if (done) return;
    done = true;
    lm_set_lds_attribute<0>(); lm_set_lds_attribute<1>(); lm_set_lds_attribute<2>(); lm_set_lds_attribute<3>(); lm_set_lds_attribute<4>();
    (void)hipGetLastError();
}
// the launch plan for a lattice; false: no shape fits
static bool lm_plan(int64_t SX, int64_t NY, int64_t NZ, lm_geom* out, size_t* lds_out) {
    const int cus = fs_rt().compute_units > 0 ? fs_rt().compute_units : 256;
    const int nh = (int)((SX / 2 + 63) / 64);
    if (nh < 1 || nh > 2 * LM_LW || SX * NY * (NZ + 8) >= ((int64_t)1 << 31)) return false;
    static const int py_env = getenv("FS_LATTICE_MARCH_PY") ? atoi(getenv("FS_LATTICE_MARCH_PY")) : 0;
    int py = LM_LW / ((nh + 1) / 2);         // (a wave takes two pieces of a line)
    if (py < 1) return false;
    if (py_env >= 1 && py_env <= py) py = py_env;
    for (; py >= 1; --py) {
        lm_geom g;
        lm_cut(&g, SX, NY, NZ, py, cus);
        const size_t lds = std::max(lm_lds_bytes(g, LM_D, true), lat_lds_bytes());
        if (lds > (size_t)(152 << 10) || (LM_D - 1) * (g.G + g.Gd) > 62) continue;
        *out = g; *lds_out = lds;
        return true;
    }
    return false;
}
template <int DOTS>
static void launch_lat_march(const double* x, double* y, const double* rvec, double* partials, int* status, int part_base, int part_stride, int bump, hipStream_t s) {
    const lm_geom& g = g_lm.g;
    const lat_geom& G = g_lat.geom;
    const size_t lds = std::max(lm_lds_bytes(g, LM_D, DOTS == 1 || DOTS == 2 || DOTS == 3), lat_lds_bytes());
    static const int lm_dbg = getenv("FS_LM_DBG") ? atoi(getenv("FS_LM_DBG")) : 0;       // (experiments: 1 no ends of the lines, 2 no line waves, 4 no loads)
    const int grid = G.n_extra + g.grid;
#define FS_LM_ARGS dim3(grid), dim3(LM_WAVES * 64), lds, s, G, g, g_dict.cls.p, g_lat.cnt.p, g_lat.coef.p, g_lat.off.p, g_lat.relc.p, g_lm.SL.p, g_lm.sl_line.p, x, y, rvec, partials, \
                   status, part_base, part_stride ? part_stride : grid, bump, lm_dbg
    if (g.NH == 1) hipLaunchKernelGGL((k_lat_march<DOTS, 1>), FS_LM_ARGS);
    else hipLaunchKernelGGL((k_lat_march<DOTS, 2>), FS_LM_ARGS);
#undef FS_LM_ARGS
}
static int lm_grid() { return g_lat.geom.n_extra + g_lm.g.grid; }

// the lists of the tile product (k_lattice_spmv) for the dictionary just built on a lattice-ordered operator; g_lat.ok says whether
// the product may be used (every class fits a list, every row's plan agrees with its class's list, no tile has too many classes)
static lat_geom lat_geometry(int64_t SX, int64_t NY, int64_t NZ) {
    lat_geom G;
    G.nxc = (int)((SX + LT_TX - 1) / LT_TX);
    const int nyt = (int)((NY + LT_TY - 1) / LT_TY), nzt = (int)((NZ + LT_TZ - 1) / LT_TZ);
    // the strips whose four lines are all interior lines in Y (LT_LOY .. NY - 1 - LT_HIY) as one long line, if that saves tiles
    static const bool no_wrap = getenv("FS_LATTICE_WRAP") && getenv("FS_LATTICE_WRAP")[0] == '0';
    G.w_ys = (LT_LOY + LT_TY - 1) / LT_TY;
    G.w_ye = (int)((NY - LT_HIY) / LT_TY);
    G.w_tiles = 0;
    if (!no_wrap && SX >= LT_TX + 8 && G.w_ye - G.w_ys >= 2 && (int64_t)(G.w_ye - G.w_ys) * SX < (int64_t)1 << 30) {
        G.w_tiles = (int)(((int64_t)(G.w_ye - G.w_ys) * SX + LT_TX - 1) / LT_TX);
        if (G.w_tiles >= (G.w_ye - G.w_ys) * G.nxc) G.w_tiles = 0;
    }
    if (!G.w_tiles) G.w_ys = G.w_ye = nyt;
    G.tiles_z = G.w_ys * G.nxc + G.w_tiles + (nyt - G.w_ye) * G.nxc;
    G.n_tiles = (int64_t)G.tiles_z * nzt;
    // Launch geometry.  The column tiles and the corner rows get workgroups of their own AT THE FRONT of the grid - they start first,
    // and the dispatcher hands their slots to tile workgroups as they finish -, the tiles one workgroup per slot of the chip behind
    // them (two workgroups of eight waves and 74 KB of LDS per CU: 512), each with its strided share of its XCD's tiles.  (Given to
    // workgroups that also had their share of tiles - the first form - the column tiles sat on the critical path of the launch: with
    // the three dots 31 us of 134 for 4 % of the rows.  Measured at configs[3], product alone / iteration of the solve, same box:
    // 256 tile workgroups 127 / 278 us, 384: 123 / 273, 512: 100 / 237 - 248, 640: 116 / 266, 768: 102 / 240, 1024: 99 / 246,
    // 2048: 98 / 248, 3072: 96 / 297 - the update kernel sums one partial per workgroup.)
    static const int tile_wgs_env = getenv("FS_LATTICE_TILE_WGS") ? atoi(getenv("FS_LATTICE_TILE_WGS")) : 512;
    const int64_t n_ct = (int64_t)2 * ((NY + CT_Y - 1) / CT_Y) * ((NZ + CT_Z - 1) / CT_Z);
    const int64_t corner_wgs = ((int64_t)(LT_LO + LT_HI) * ((NZ + 7) >> 3) + (LT_BLOCK / 64) - 1) / (LT_BLOCK / 64);
    G.n_ct_wgs = (int)std::min<int64_t>(n_ct, 1024);
    G.n_extra = (int)((G.n_ct_wgs + std::min<int64_t>(corner_wgs, 256) + 7) & ~(int64_t)7);
    int64_t tg = std::min<int64_t>(std::max<int64_t>(G.n_tiles, 8), std::max(tile_wgs_env, 8));
    tg = std::min<int64_t>((tg + 7) & ~(int64_t)7, (FS_MAX_PARTIAL_BLOCKS - G.n_extra) & ~7);
    G.grid = G.n_extra + (int)tg;
    return G;
}

static int lat_prepare(fs_matrix_s* A, const double* val, hipStream_t s) {
    fs_space_s* sp = A->space;
    g_lat.ok = false;
    g_lat.judged = true;
    g_lat.built_for = nullptr;
    static const bool off = getenv("FS_LATTICE_TILES") && getenv("FS_LATTICE_TILES")[0] == '0';
    if (off || sp->lat_ny <= 0 || A->bs != 1 || g_dict.built_for != val || g_dict.bs != 1 || g_dict.space_serial != sp->serial || sp->n_dict_items <= 0) return FS_OK;
    const int64_t SX = sp->dict_line, NY = sp->lat_ny, NZ = sp->lat_nz, n = SX * NY * NZ;
    const int ncls = g_dict.ncls;
    if (n != sp->n_nodes_owned || sp->n_nodes_local != sp->n_nodes_owned || NY < 2 * (LT_LOY + LT_HIY) || SX < 2 * (LT_LO + LT_HI) || (SX & 1) || ncls <= 0) return FS_OK;
    if (g_lat.rep.n < ncls) {
        FS_CHECK(g_lat.rep.alloc(ncls));
        FS_CHECK(g_lat.cnt.alloc(ncls));
        FS_CHECK(g_lat.off.alloc((int64_t)ncls * LT_ML + 64));
        FS_CHECK(g_lat.rel.alloc((int64_t)ncls * LT_ML + 64));
        FS_CHECK(g_lat.relc.alloc((int64_t)ncls * LT_ML + 64));
        FS_CHECK(g_lat.coef.alloc((int64_t)ncls * LT_ML + 64));
    }
    if (!g_lat.info.p) FS_CHECK(g_lat.info.alloc(4));
    // A KEPT dictionary (dict_build_impl: every row of this matrix compared with its old class, bit for bit - same class numbers, same
    // class rows) is the one these lists were made from and every row was checked against: nothing to do (configs[3]: 2.0 + 0.6 + 0.6 ms
    // of kernels, six fills and a host round trip per solve).  n_built counts the table builds: a build in between makes the lists stale.
    const bool lists_kept = g_dict.kept && g_lat.tables_ok && g_lat.space_serial == sp->serial && g_lat.ncls == ncls && g_lat.dict_built == g_dict.n_built;
    int h[4] = {0, 0, 0, 0};
    static const bool debug = getenv("FS_LATTICE_DEBUG") != nullptr;
    if (!lists_kept) {
    g_lat.tables_ok = false;
    FS_HIP(hipMemsetAsync(g_lat.rep.p, 0x7f, (size_t)ncls * 4, s));
    FS_CHECK(g_lat.cnt.zero(s));
    FS_CHECK(g_lat.coef.zero(s));        // (the padded positions of a list are read, and multiplied with nothing)
    FS_CHECK(g_lat.rel.zero(s));
    FS_CHECK(g_lat.relc.zero(s));
    FS_CHECK(g_lat.off.zero(s));
    FS_CHECK(g_lat.info.zero(s));
    hipLaunchKernelGGL(k_lat_rep, dim3(fs_grid_for(n, FS_BLOCK, 4096)), dim3(FS_BLOCK), 0, s, n, g_dict.cls.p, g_lat.rep.p);
    const int gi = fs_grid_for(sp->n_dict_items * 64, FS_BLOCK, 4096);
#define FS_LAT_TAB_ARGS dim3(gi), dim3(FS_BLOCK), 0, s, sp->n_dict_items, reinterpret_cast<const int4*>(sp->dict_items.p), \
                        reinterpret_cast<const dict_plan_round*>(sp->dict_plans.p), g_dict.cls.p, g_lat.rep.p, g_dict.values.p, g_dict.S, sp->dict_run_len, \
                        sp->dict_runs, SX, NY, g_lat.cnt.p, g_lat.coef.p, g_lat.rel.p, g_lat.off.p, g_lat.info.p, g_lat.relc.p
    hipLaunchKernelGGL(k_lat_table<true>, FS_LAT_TAB_ARGS);
    hipLaunchKernelGGL(k_lat_table<false>, FS_LAT_TAB_ARGS);
#undef FS_LAT_TAB_ARGS
    g_lat.geom = lat_geometry(SX, NY, NZ);
    if (g_lat.geom.n_tiles >= ((int64_t)1 << 26) || ncls > 65535) return FS_OK;        // (cannot be: 32-bit rows, 16-bit classes)
    if (g_lat.tile_cls.n < g_lat.geom.n_tiles * (LT_BLOCK / 64) * 4) FS_CHECK(g_lat.tile_cls.alloc(g_lat.geom.n_tiles * (LT_BLOCK / 64) * 4));
    hipLaunchKernelGGL(k_lat_tile_table, dim3((unsigned)std::min<int64_t>(g_lat.geom.n_tiles, 4096)), dim3(LT_BLOCK), 0, s, g_lat.geom, SX, NY, NZ,
                       g_dict.cls.p, g_lat.cnt.p, g_lat.tile_cls.p);
    FS_KERNEL_CHECK();
    // the tables of the marching-window product (fs_latmarch.h): coefficient rows by stencil position for every (class, parity class)
    // that occurs, the row numbers of the lines, the step lists; info[1]: entries outside their parity's stencil, info[2]: interior
    // rows of a line that are not of the line's class, info[3]: step lists
    g_lm.ok = false;
    bool lm_tried = false;
    int64_t n_tuples = 0;
    if (g_lat_march && ncls * 8 + 1 <= 65535 && lm_plan(SX, NY, NZ, &g_lm.g, &g_lm.lds)) {
        lm_tried = true;
        const lm_geom& mg = g_lm.g;
        n_tuples = (int64_t)NY * mg.NS2;
        int64_t tab_n = 1024;
        while (tab_n < 2 * n_tuples) tab_n *= 2;
        if (g_lm.used.n < (int64_t)ncls * 8) { FS_CHECK(g_lm.used.alloc((int64_t)ncls * 8)); FS_CHECK(g_lm.coefS.alloc(((int64_t)ncls * 8 + 1) * LM_CS)); }
        if (g_lm.lc.n < (int64_t)NY * mg.LZ) FS_CHECK(g_lm.lc.alloc((int64_t)NY * mg.LZ));
        if (g_lm.rep.n < n_tuples) { FS_CHECK(g_lm.rep.alloc(n_tuples)); FS_CHECK(g_lm.num.alloc(n_tuples)); }
        if (g_lm.sl_line.n < (int64_t)mg.NYP * mg.NS2) FS_CHECK(g_lm.sl_line.alloc((int64_t)mg.NYP * mg.NS2));
        if (g_lm.tab.n != tab_n) FS_CHECK(g_lm.tab.alloc(tab_n));
        FS_CHECK(g_lm.used.zero(s));
        FS_CHECK(g_lm.coefS.zero(s));
        FS_CHECK(g_lm.sl_line.zero(s));
        FS_HIP(hipMemsetAsync(g_lm.tab.p, 0xff, (size_t)tab_n * 4, s));
        hipLaunchKernelGGL(k_lm_used, dim3(fs_grid_for(n, FS_BLOCK, 4096)), dim3(FS_BLOCK), 0, s, n, SX, NY, g_dict.cls.p, g_lm.used.p);
        hipLaunchKernelGGL(k_lm_coef, dim3((ncls * 8 + 63) / 64), dim3(64), 0, s, ncls, SX, NY, g_lm.used.p, g_lat.cnt.p, g_lat.off.p, g_lat.coef.p, LT_ML, g_lm.coefS.p, g_lat.info.p);
        hipLaunchKernelGGL(k_lm_lines, dim3(fs_grid_for((int64_t)NY * mg.LZ * 64, FS_BLOCK, 4096)), dim3(FS_BLOCK), 0, s, mg, ncls * 8, g_dict.cls.p, g_lm.lc.p, g_lat.info.p);
        hipLaunchKernelGGL(k_lm_dedupe, dim3(fs_grid_for(n_tuples, FS_BLOCK, 4096)), dim3(FS_BLOCK), 0, s, mg, n_tuples, g_lm.lc.p, g_lm.tab.p, (uint32_t)(tab_n - 1), g_lm.rep.p);
        hipLaunchKernelGGL(k_lm_number, dim3(fs_grid_for(n_tuples, FS_BLOCK, 4096)), dim3(FS_BLOCK), 0, s, n_tuples, g_lm.rep.p, g_lm.num.p, g_lat.info.p);
        FS_KERNEL_CHECK();
        lm_prepare_kernels();
    }
    FS_CHECK(g_lat.info.download(h, 4, s));
    if (debug) fprintf(stderr, "[lattice tiles] %d classes, tiles of %d x %d x %d rows: %d entries / rows that do not fit\n", ncls, LT_TX, LT_TY, LT_TZ, h[0]);
    g_lat.tables_ok = h[0] == 0;
    g_lat.dict_built = g_dict.n_built;
    g_lm.ok = lm_tried && h[0] == 0 && h[1] == 0 && h[2] == 0 && h[3] > 0;
    if (g_lm.ok) {
        const lm_geom& mg = g_lm.g;
        g_lm.n_lists = h[3];
        if (g_lm.SL.n < (int64_t)h[3] * LM_SL + 64) FS_CHECK(g_lm.SL.alloc((int64_t)h[3] * LM_SL + 64));
        hipLaunchKernelGGL(k_lm_fill, dim3(fs_grid_for(n_tuples, FS_BLOCK, 4096)), dim3(FS_BLOCK), 0, s, mg, n_tuples, g_lm.lc.p, g_lm.rep.p, g_lm.num.p, g_lm.coefS.p,
                           g_lm.sl_line.p, g_lm.SL.p, h[3]);
        FS_KERNEL_CHECK();
    }
    if (debug || getenv("FS_KRYLOV_DEBUG"))
        fprintf(stderr, "[lattice march] %s: %d entries outside their stencil, %d interior rows not of their line's class, %d step lists; patches of %d lines, %d x %d units, %d workgroups, %zu B of LDS\n",
                g_lm.ok ? "on" : (lm_tried ? "refused" : "off"), h[1], h[2], h[3], g_lm.g.PY, g_lm.g.NP, g_lm.g.ZC, g_lm.g.grid, g_lm.lds);
    }
    g_lat.ok = g_lat.tables_ok;
    g_lat.built_for = val;
    g_lat.space_serial = sp->serial;
    g_lat.ncls = ncls;
    if (g_lat.ok && g_lat_check) {
        // option "lattice_check": the tile product against k_dict_spmv on a vector of pseudo-random numbers, every row, bit for bit
        dbuf<double> xv, y1, y2;
        FS_CHECK(xv.alloc(n + 2)); FS_CHECK(y1.alloc(n + 2)); FS_CHECK(y2.alloc(n + 2));
        hipLaunchKernelGGL(k_lat_fill, dim3(fs_grid_for(n + 2, FS_BLOCK, 4096)), dim3(FS_BLOCK), 0, s, n + 2, xv.p);
        launch_spmv<0>(A, xv.p, y1.p, nullptr, nullptr, nullptr, s, val, nullptr, 0, 0, 0, 0);
        g_lat.ok = false;
        launch_spmv<0>(A, xv.p, y2.p, nullptr, nullptr, nullptr, s, val, nullptr, 0, 0, 0, 0);
        g_lat.ok = true;
        FS_CHECK(g_lat.info.zero(s));
        hipLaunchKernelGGL(k_lat_count_diff, dim3(fs_grid_for(n, FS_BLOCK, 4096)), dim3(FS_BLOCK), 0, s, n, y1.p, y2.p, g_lat.info.p);
        FS_KERNEL_CHECK();
        FS_CHECK(g_lat.info.download(h, 1, s));
        if (debug) fprintf(stderr, "[lattice tiles] tile product against the work-item product: %d of %lld rows differ\n", h[0], (long long)n);
        if (debug) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            const int variants[] = {0, 8, 1, -1};
            for (int v : variants) {
                g_lt_dbg = v < 0 ? 0 : v;
                if (v < 0) g_lat.ok = false;        // the work-item product
                launch_spmv<0>(A, xv.p, y1.p, nullptr, nullptr, nullptr, s, val, nullptr, 0, 0, 0, 0);
                (void)hipEventRecord(e0, s);
                for (int it = 0; it < 10; ++it) launch_spmv<0>(A, xv.p, y1.p, nullptr, nullptr, nullptr, s, val, nullptr, 0, 0, 0, 0);
                (void)hipEventRecord(e1, s);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                fprintf(stderr, "[lattice tiles]   %-44s %.1f us per product\n", v < 0 ? "work-item product on this operator:" : (v == 0 ? "tile product:" : (v == 8 ? "tile product without the corner rows:" : "tile product without the ends of the lines:")), ms * 100.0);
            }
            g_lt_dbg = 0;
            g_lat.ok = true;
            // the same with the three fused dots, and with 1 GB streamed between the launches (x, y out of the caches as behind the
            // update kernel of an iteration): per-launch events
            if (getenv("FS_LATTICE_DEBUG")[0] == '2') {
                dbuf<double> big, part, rv2;
                dbuf<int> st;
                const int64_t nbig = (int64_t)128 << 20;
                if (big.alloc(nbig) == FS_OK && part.alloc(3 * 4096) == FS_OK && st.alloc(8) == FS_OK && rv2.alloc(n + 2) == FS_OK) {
                    (void)st.zero(s);
                    hipLaunchKernelGGL(k_lat_fill, dim3(fs_grid_for(n + 2, FS_BLOCK, 4096)), dim3(FS_BLOCK), 0, s, n + 2, rv2.p);
                    for (int tile = 5; tile >= 0; --tile)
                        for (int dots = 0; dots <= 3; dots += 3)
                            for (int cold = 0; cold <= 1; ++cold) {
                                if (tile >= 2 && (!dots || cold)) continue;
                                g_lt_dbg = tile == 5 ? 3 : (tile == 4 ? 1 : (tile == 3 ? 2 : (tile == 2 ? 4 : 0)));
                                if (tile >= 2) fprintf(stderr, "[lattice tiles]   (ablation %d: %s)\n", g_lt_dbg, g_lt_dbg == 3 ? "no ends of the lines, no loads of r" : (g_lt_dbg == 1 ? "no ends of the lines" : (g_lt_dbg == 2 ? "no loads of r" : "no dot accumulation")));
                                g_lat.ok = tile != 0;
                                float total = 0.f;
                                for (int it = 0; it < 6; ++it) {
                                    if (cold) hipLaunchKernelGGL(k_lat_fill, dim3(4096), dim3(FS_BLOCK), 0, s, nbig, big.p);
                                    (void)hipEventRecord(e0, s);
                                    if (dots) launch_spmv<3>(A, xv.p, y1.p, rv2.p, part.p, st.p, s, val, nullptr, 0, 0, 0, 0);
                                    else launch_spmv<0>(A, xv.p, y1.p, nullptr, nullptr, nullptr, s, val, nullptr, 0, 0, 0, 0);
                                    (void)hipEventRecord(e1, s);
                                    (void)hipEventSynchronize(e1);
                                    float ms = 0.f;
                                    (void)hipEventElapsedTime(&ms, e0, e1);
                                    if (it) total += ms;
                                }
                                fprintf(stderr, "[lattice tiles]   %s, %s, %s: %.1f us\n", tile ? "tile product" : "work-item product on this operator", dots ? "three dots" : "no dots",
                                        cold ? "caches streamed over" : "warm", total * 200.0);
                            }
                    g_lat.ok = true;
                }
            }
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        if (h[0] != 0) {
            g_lat.ok = false;
            g_lat.tables_ok = false;
            fs_set_error("lattice_check: the tile product differs from the work-item product on %d of %lld rows", h[0], (long long)n);
            return FS_ERR_NUMERIC;
        }
    }
    return FS_OK;
}

static int dict_build(fs_matrix_s* A, const double* val, hipStream_t s, const double* raw = nullptr, const double* sc = nullptr,
                      const std::function<void()>& scale_copy = nullptr) {
    bool copied = raw == nullptr;
    const auto materialize = [&]() {
        if (!copied) { scale_copy(); copied = true; }
    };
    const int rc = dict_build_impl(A, val, s, raw, sc, materialize);
    // every outcome but `the kept table describes this matrix` reads val: the streaming kernels, or a table just built from it
    if (!(g_dict.built_for == val && g_dict.kept)) materialize();
    if (rc == FS_OK && A->space->lat_ny > 0) FS_CHECK(lat_prepare(A, val, s));
    return rc;
}

// ---- the marching-window product of a P1 box operator (fs_box.h) ------------------------------------------------------------------
// Two launch shapes: mesh lines up to 320 rows - 6 compute waves x 2 rows per lane (patches of <= 768 rows, two workgroups per CU);
// longer lines - 8 x 3 (<= 1536 rows, one workgroup per CU: the window's halo of 2 (a + 1) positions is paid per patch).  Measured
// (tools/probes/box_spmv_probe.hip, profiles/r06_box_probe.txt): 10 M rows 51 us, 86 M rows 382 - 400 us inside an iteration-like
// sequence, where a plain streaming kernel over the same 26 B/row takes 57 and 421.
struct box_plan_s {
    unsigned long long space_serial = 0;
    int ncls = 0, shape = -1;       // shape 0: 6 x 2, 1: 8 x 3, -1: does not fit
    box_geom g;
    size_t lds = 0;
};
static box_plan_s g_box_plan;
// (the kernels take up to 160 KB of dynamic LDS: the attribute is set for every instantiation when the first plan is made - by
// dict_build, outside any stream capture - not at the launch, which may be a node of a CG batch being captured)
template <int DOTS>
static void box_set_lds_attribute() {
    (void)hipFuncSetAttribute((const void*)k_box_spmv<DOTS, 6, 2, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
    (void)hipFuncSetAttribute((const void*)k_box_spmv<DOTS, 8, 3, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
}
static void box_prepare_kernels() {
    static bool done = false;
    if (done) return;
    done = true;
    box_set_lds_attribute<0>(); box_set_lds_attribute<1>(); box_set_lds_attribute<2>(); box_set_lds_attribute<3>(); box_set_lds_attribute<4>();
    (void)hipGetLastError();
}
static const box_plan_s* box_plan_for(const fs_space_s* sp, int ncls) {
    if (!g_box || sp->box_a <= 0 || sp->n_nodes_owned < g_box_min_rows || sp->halo.active) return nullptr;
    box_plan_s& B = g_box_plan;
    if (B.space_serial != sp->serial || B.ncls != ncls) {
        B.space_serial = sp->serial; B.ncls = ncls; B.shape = -1;
        int32_t starts[8] = {0, (int32_t)-(sp->box_a + sp->box_b + 1), (int32_t)-(sp->box_b + 1), -(sp->box_a + 1), -1, sp->box_a, (int32_t)sp->box_b, (int32_t)(sp->box_a + sp->box_b)};
        const uint8_t lens[8] = {0, 2, 2, 2, 3, 2, 2, 2};
        box_geom g;
        if (ncls * BOX_TERMS * 8 <= (24 << 10) && box_recognize(starts, lens, 8, sp->n_nodes_owned, 3, &g)) {
            const int cus = fs_rt().compute_units > 0 ? fs_rt().compute_units : 256;
            static const int shape_env = getenv("FS_BOX_SHAPE") ? atoi(getenv("FS_BOX_SHAPE")) : -1;
            for (int shape = (shape_env >= 0 ? shape_env : (sp->box_a <= 320 ? 0 : 1)); shape <= 1; ++shape) {
                const int cw = shape == 0 ? 6 : 8, rp = shape == 0 ? 2 : 3;
                box_geom t = g;
                t.S = sp->dict_slots;
                box_cut(&t, cw * 64 * rp, cus, 1);
                const size_t lds = box_lds_bytes(t, ncls, 2, true);
                const int per_cu = (int)std::min<size_t>((size_t)(160 << 10) / (lds + 256), shape == 0 ? 2 : 1);
                if (per_cu < 1 || (t.G + (t.dslot >> 7) + (t.cslot >> 9)) > 62) continue;
                box_cut(&t, cw * 64 * rp, cus * per_cu, 1);
                B.g = t; B.lds = box_lds_bytes(t, ncls, 2, true); B.shape = shape;
                box_prepare_kernels();
                break;
            }
        }
        if (getenv("FS_KRYLOV_DEBUG"))
            fprintf(stderr, "[fs_krylov] marching-window product: shape %d, a %d b %lld, patches %d x %d rows, %d chunks of %d planes, %d workgroups, %zu B of LDS\n",
                    B.shape, sp->box_a, (long long)sp->box_b, B.g.P, B.g.L, B.g.ZC, B.g.nz, B.g.grid, B.lds);
    }
    return B.shape >= 0 ? &B : nullptr;
}
// launch plan of the one-launch iteration in marching-window form (k_box_cg_iter): patches of <= 512 rows (4 compute waves x 2 rows per
// lane + 2 loaders); the loaders one step ahead with two workgroups per CU, or two steps ahead with one (option "box_iter_ahead").
// The same geometry serves product 0 of the solve (k_box_spmv<3, 6, 2, 2, 2>: its patches may be anything up to 768 rows).
struct box_iter_plan_s {
    unsigned long long space_serial = 0;
    int ncls = 0, ahead = 0, ok = 0;
    box_geom g;
    size_t lds = 0, lds_product = 0;
};
static box_iter_plan_s g_box_iter_plan;
static const box_iter_plan_s* box_iter_plan_for(const fs_space_s* sp, int ncls) {
    if (!g_box_iter || !g_box || sp->box_a <= 0 || sp->n_nodes_owned < g_box_iter_min_rows || sp->halo.active) return nullptr;
    box_iter_plan_s& B = g_box_iter_plan;
    const int D = g_box_iter_ahead == 2 ? 2 : 1;
    if (B.space_serial != sp->serial || B.ncls != ncls || B.ahead != D) {
        B.space_serial = sp->serial; B.ncls = ncls; B.ahead = D; B.ok = 0;
        int32_t starts[8] = {0, (int32_t)-(sp->box_a + sp->box_b + 1), (int32_t)-(sp->box_b + 1), -(sp->box_a + 1), -1, sp->box_a, (int32_t)sp->box_b, (int32_t)(sp->box_a + sp->box_b)};
        const uint8_t lens[8] = {0, 2, 2, 2, 3, 2, 2, 2};
        box_geom g;
        if (ncls * BOX_TERMS * 8 <= (24 << 10) && box_recognize(starts, lens, 8, sp->n_nodes_owned, 3, &g)) {
            const int cus = fs_rt().compute_units > 0 ? fs_rt().compute_units : 256;
            g.S = sp->dict_slots;
            box_geom t = g;
            box_cut(&t, 512, cus, 1);
            size_t lds = box_iter_lds_bytes(t, ncls, D);
            const int per_cu = (int)std::min<size_t>((size_t)(160 << 10) / (lds + 256), 2);
            if (per_cu >= 1 && 3 * t.G * (D - 1) <= 62 && (3 * (t.dslot >> 7) + (t.cslot >> 9)) * (D - 1) <= 62) {
                t = g;
                box_cut(&t, 512, cus * per_cu, 1);
                B.g = t;
                B.lds = box_iter_lds_bytes(t, ncls, D);
                B.lds_product = box_lds_bytes(t, ncls, 2, true);
                B.ok = t.grid <= 1024 && B.lds_product <= (size_t)(160 << 10);
                if (B.ok) {
                    box_prepare_kernels();
                    static bool attr = false;
                    if (!attr) {
                        attr = true;
                        (void)hipFuncSetAttribute((const void*)k_box_cg_iter<4, 2, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
                        (void)hipFuncSetAttribute((const void*)k_box_cg_iter<4, 2, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
                        (void)hipGetLastError();
                    }
                }
            }
        }
        if (getenv("FS_KRYLOV_DEBUG"))
            fprintf(stderr, "[fs_krylov] marching-window iteration: ok %d, %d step(s) ahead, patches %d x %d rows, %d chunks of %d planes, %d workgroups, %zu B of LDS\n",
                    B.ok, D, B.g.P, B.g.L, B.g.ZC, B.g.nz, B.g.grid, B.lds);
    }
    return B.ok ? &B : nullptr;
}

template <int DOTS>
static void launch_box(const box_plan_s* B, const uint16_t* cls, const double* dict, int ncls, const double* x, double* y, const double* rvec,
                       double* partials, int* status, int part_base, int part_stride, int bump, hipStream_t s) {
    const box_geom& g = B->g;
    if (B->shape == 0) {
        auto kern = k_box_spmv<DOTS, 6, 2, 2, 2>;
        hipLaunchKernelGGL(kern, dim3(g.grid), dim3(8 * 64), B->lds, s, g, cls, dict, ncls, x, y, rvec, partials, status, part_base, part_stride ? part_stride : g.grid, bump);
    } else {
        auto kern = k_box_spmv<DOTS, 8, 3, 2, 2>;
        hipLaunchKernelGGL(kern, dim3(g.grid), dim3(10 * 64), B->lds, s, g, cls, dict, ncls, x, y, rvec, partials, status, part_base, part_stride ? part_stride : g.grid, bump);
    }
}

// `list` / `n_list`: multiply only these slices (the interior or the boundary slices of a decomposed space, in
// processing order); nullptr = all slices in the space's own order.
template <int DOTS>
static void launch_spmv(fs_matrix_s* A, const double* x, double* y, const double* rvec, double* partials,
                        int* status, hipStream_t s, const double* val_override,
                        const int32_t* list, int64_t n_list, int part_base, int part_stride, int bump) {
    const double* mat_val = val_override ? val_override : A->val.p;
    fs_space_s* sp = A->space;
    const int64_t ns = list ? n_list : sp->n_slices;
    if (ns == 0) return;
    const int32_t* order = list ? list : sp->slice_order.p;
    if (A->bs == 3 && DOTS == 0 && !list && g_dict.bs == 3 && g_dict.built_for && g_dict.built_for == mat_val && g_dict.matrix_serial == A->serial &&
        g_dict.space_serial == sp->serial && sp->n_dict_items > 0) {
        // 3 x 3 block rows from class numbers + class rows (the fine-level product of the elasticity AMG on a uniform box)
        const int64_t n_chunks = (sp->n_dict_items + 4 * FS_DICT_ITEMS_PER_WAVE - 1) / (4 * FS_DICT_ITEMS_PER_WAVE);
        const int gd = (int)std::max<int64_t>((std::min<int64_t>(n_chunks, 1024) + 7) & ~(int64_t)7, 8);
        const size_t lds = (size_t)(FS_BLOCK / 64) * g_dict.C * g_dict.S * sizeof(double);
#define FS_DICT3_ARGS dim3(gd), dim3(FS_BLOCK), lds, s, sp->n_nodes_local, sp->n_dict_items, reinterpret_cast<const int4*>(sp->dict_items.p), \
                      reinterpret_cast<const dict_plan_round*>(sp->dict_plans.p), g_dict.cls.p, g_dict.values.p, g_dict.S, g_dict.C, x, y, dict_map_xcd()
        if (sp->dict_run_len == 2) hipLaunchKernelGGL((k_dict_spmv3<2>), FS_DICT3_ARGS);
        else hipLaunchKernelGGL((k_dict_spmv3<3>), FS_DICT3_ARGS);
#undef FS_DICT3_ARGS
        g_last_product_kind = 4;
        return;
    }
    if (A->bs == 1 && g_dict.bs == 1 && g_dict.built_for && g_dict.built_for == mat_val && g_dict.matrix_serial == A->serial && g_dict.space_serial == sp->serial) {
        // the values are a few dozen distinct rows (dict_build): class numbers + dictionary in LDS instead of the value stream.
        // Whole space: its own launch geometry (spmv_partials_unsplit); a list of a decomposed space (interior / boundary slices):
        // the geometry of the streaming kernel, so that the two launches keep filling one partial array.
        const int32_t* items = sp->dict_items.p;
        int64_t n_items = sp->n_dict_items;
        int gd = spmv_partials_unsplit(sp, 1);
        if (list) {
            fs_halo_plan& h = sp->halo;
            const bool in = list == h.interior.p;
            items = in ? h.items_interior.p : (list == h.boundary.p ? h.items_boundary.p : nullptr);
            n_items = in ? h.n_items_interior : h.n_items_boundary;
            gd = spmv_grid(ns, sp->n_slices);
        }
        if (!list && g_lat.ok && g_lat.built_for == mat_val && g_lat.space_serial == sp->serial && g_lat.ncls == g_dict.ncls) {
            if (g_lm.ok && g_lat_march) {
                // ... in marching-window form (k_lat_march)
                launch_lat_march<DOTS>(x, y, rvec, partials, status, part_base, part_stride, bump, s);
                g_last_product_kind = 5;
                return;
            }
            // a lattice-ordered operator: tiles of 128 x 4 x 4 rows, x through LDS (k_lattice_spmv)
            const int64_t SX = sp->dict_line, NY = sp->lat_ny, NZ = sp->lat_nz;
            const size_t lds = lat_lds_bytes();
            auto kern = k_lattice_spmv<DOTS>;
            static bool attr_set = false;       // (one per instantiation)
            if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
            gd = g_lat.geom.grid;              // (spmv_partials_unsplit says the same: the dot partials of this launch)
            hipLaunchKernelGGL(kern, dim3(gd), dim3(LT_BLOCK), lds, s, g_lat.geom, g_lat.tile_cls.p, SX, NY, NZ, g_dict.cls.p,
                               g_lat.cnt.p, g_lat.coef.p, g_lat.rel.p, g_lat.off.p, g_lat.relc.p, x, y, rvec, partials, status, part_base, part_stride ? part_stride : gd, bump,
                               g_lt_dbg | g_lt_dbg_env);
            g_last_product_kind = 2;
            return;
        }
        if (!list && sp->dict_runs == 8 && sp->dict_run_len == 3) {
            if (const box_plan_s* B = box_plan_for(sp, g_dict.ncls)) {
                launch_box<DOTS>(B, g_dict.cls.p, g_dict.values.p, g_dict.ncls, x, y, rvec, partials, status, part_base, part_stride, bump, s);
                g_last_product_kind = 3;
                return;
            }
        }
        if (items && n_items >= 0) {
#define FS_DICT_ARGS(CC) sp->n_nodes_local, n_items, reinterpret_cast<const int4*>(items), reinterpret_cast<const dict_plan_round*>(sp->dict_plans.p), \
                         g_dict.cls.p, g_dict.values.p, g_dict.S, CC, x, y, rvec, partials, status, part_base, part_stride ? part_stride : gd, bump, dict_map_xcd()
            const size_t whole = (size_t)g_dict.ncls * g_dict.S * sizeof(double);
            const size_t per_wave = (size_t)(FS_BLOCK / 64) * g_dict.C * g_dict.S * sizeof(double);
            const bool rl2 = sp->dict_run_len == 2;
            if (whole <= (size_t)FS_DICT_WHOLE_LDS_BYTES && sp->dict_runs == 8) {
                if (rl2) hipLaunchKernelGGL((k_dict_spmv<DOTS, true, 2>), dim3(gd), dim3(FS_BLOCK), whole, s, FS_DICT_ARGS(g_dict.ncls));
                else hipLaunchKernelGGL((k_dict_spmv<DOTS, true, 3>), dim3(gd), dim3(FS_BLOCK), whole, s, FS_DICT_ARGS(g_dict.ncls));
            } else {
                if (sp->dict_runs == 12) hipLaunchKernelGGL((k_dict_spmv<DOTS, false, 3, 12>), dim3(gd), dim3(FS_BLOCK), per_wave, s, FS_DICT_ARGS(g_dict.C));
                else if (rl2) hipLaunchKernelGGL((k_dict_spmv<DOTS, false, 2>), dim3(gd), dim3(FS_BLOCK), per_wave, s, FS_DICT_ARGS(g_dict.C));
                else hipLaunchKernelGGL((k_dict_spmv<DOTS, false, 3>), dim3(gd), dim3(FS_BLOCK), per_wave, s, FS_DICT_ARGS(g_dict.C));
            }
#undef FS_DICT_ARGS
            g_last_product_kind = 1;
            return;
        }
    }
    g_last_product_kind = 0;
    const int grid = spmv_grid(ns, sp->n_slices);
    if (part_stride == 0) part_stride = grid;
#define FS_SPMV_ARGS dim3(grid), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->n_nodes_local, ns, sp->slice_ptr.p, sp->sell_col.p, sp->dia_ptr.p, sp->dia_off.p, mat_val, sp->sell_entries, x, y, rvec, partials, status, order, part_base, part_stride, bump
    if (A->bs == 1 && !list && sp->n_pairs > 0 && spmv_use_pairs(sp, 1)) {
        // two launches: the paired slices (two rows per lane), then the rest through the one-row-per-lane kernel
        const bool nt = spmv_nontemporal(sp, 1);
        const int gp = spmv_pair_grid(sp);
        const int gs = sp->n_pair_singles ? spmv_grid(sp->n_pair_singles, sp->n_slices) : 0;
        const int stride = gp + gs;
#define FS_PAIR_ARGS dim3(gp), dim3(FS_BLOCK), 0, s, sp->n_nodes_local, sp->n_pairs, sp->pair_list.p, sp->slice_ptr.p, sp->dia_ptr.p, sp->dia_off.p, mat_val, x, y, rvec, partials, status, 0, stride, bump
        if (nt) hipLaunchKernelGGL((k_dia_pair_spmv<DOTS, true>), FS_PAIR_ARGS);
        else hipLaunchKernelGGL((k_dia_pair_spmv<DOTS, false>), FS_PAIR_ARGS);
#undef FS_PAIR_ARGS
        if (gs) launch_spmv<DOTS>(A, x, y, rvec, partials, status, s, val_override, sp->pair_singles.p, sp->n_pair_singles, gp, stride, 0);
        return;
    }
    if (A->bs == 1) {
        const bool nt = spmv_nontemporal(sp, 1);
        switch (spmv_unroll_for(sp->n_slices)) {
            case 2: hipLaunchKernelGGL((k_sell_spmv<1, DOTS, 2>), FS_SPMV_ARGS); break;
            case 8:
                if (nt) hipLaunchKernelGGL((k_sell_spmv<1, DOTS, 8, true>), FS_SPMV_ARGS);
                else hipLaunchKernelGGL((k_sell_spmv<1, DOTS, 8>), FS_SPMV_ARGS);
                break;
            case 16:
                if (nt) hipLaunchKernelGGL((k_sell_spmv<1, DOTS, 16, true>), FS_SPMV_ARGS);
                else hipLaunchKernelGGL((k_sell_spmv<1, DOTS, 16>), FS_SPMV_ARGS);
                break;
            default:
                if (nt) hipLaunchKernelGGL((k_sell_spmv<1, DOTS, 4, true>), FS_SPMV_ARGS);
                else hipLaunchKernelGGL((k_sell_spmv<1, DOTS, 4>), FS_SPMV_ARGS);
                break;
        }
    } else if (A->bs == 2) {
        hipLaunchKernelGGL((k_sell_spmv<2, DOTS, 4>), FS_SPMV_ARGS);
    } else if (A->bs == 3) {
        if (spmv_nontemporal(sp, 3)) hipLaunchKernelGGL((k_sell_spmv<3, DOTS, 4, true>), FS_SPMV_ARGS);
        else hipLaunchKernelGGL((k_sell_spmv<3, DOTS, 4>), FS_SPMV_ARGS);
    } else {
        switch (g_spmv_unroll4) {
            case 1: hipLaunchKernelGGL((k_sell_spmv<4, DOTS, 1>), FS_SPMV_ARGS); break;
            case 4: hipLaunchKernelGGL((k_sell_spmv<4, DOTS, 4>), FS_SPMV_ARGS); break;
            default: hipLaunchKernelGGL((k_sell_spmv<4, DOTS, 2>), FS_SPMV_ARGS); break;
        }
    }
#undef FS_SPMV_ARGS
}

// Number of per-workgroup dot partials one (possibly split) product writes.
static bool spmv_is_split(const fs_space_s* sp) { return sp->halo.active && sp->halo.n_interior > 0; }
static int dict_map_xcd() {
    static const int m = getenv("FS_DICT_MAP") ? atoi(getenv("FS_DICT_MAP")) : 1;
    return m;
}
static int dict_grid(const fs_space_s* sp) {
    // (measured at 1 M rows, round 3: 256 / 512 / 768 / 1024 / 2048 workgroups: 31 / 21 / 19 / 19 / 18.5 us, the update kernel that sums
    // the partials + 0 / 0.5 / 1 / 1 / 3 us)
    static const int env_blocks = getenv("FS_DICT_BLOCKS") ? atoi(getenv("FS_DICT_BLOCKS")) : 0;
    const bool whole = (size_t)g_dict.ncls * g_dict.S * sizeof(double) <= (size_t)FS_DICT_WHOLE_LDS_BYTES;
    const int per_chunk = 4 * (whole ? 1 : FS_DICT_ITEMS_PER_WAVE);
    const int64_t n_chunks = (std::max<int64_t>(sp->n_dict_items, 1) + per_chunk - 1) / per_chunk;
    int64_t g = std::min<int64_t>(n_chunks, env_blocks > 0 ? env_blocks : 1024);
    g = (g + 7) & ~(int64_t)7;
    return (int)std::max<int64_t>(g, 8);
}
static int spmv_partials_unsplit(const fs_space_s* sp, int bs) {
    if (bs == 1 && g_dict.built_for && g_dict.space_serial == sp->serial) {
        // (the tile product of a lattice-ordered operator has its own geometry: launch_spmv's condition)
        if (g_lat.ok && g_lat.built_for == g_dict.built_for && g_lat.space_serial == sp->serial && g_lat.ncls == g_dict.ncls && g_lat.geom.grid > 0)
            return g_lm.ok && g_lat_march ? lm_grid() : g_lat.geom.grid;
        if (g_dict.bs == 1 && sp->dict_runs == 8 && sp->dict_run_len == 3)
            if (const box_plan_s* B = box_plan_for(sp, g_dict.ncls)) return B->g.grid;      // (launch_spmv's condition)
        return dict_grid(sp);
    }
    if (bs == 1 && sp->n_pairs > 0 && spmv_use_pairs(sp, 1))
        return spmv_pair_grid(sp) + (sp->n_pair_singles ? spmv_grid(sp->n_pair_singles, sp->n_slices) : 0);
    return spmv_grid(sp->n_slices, sp->n_slices);
}
static int spmv_partials(const fs_space_s* sp, int bs = 0) {
    if (!spmv_is_split(sp)) return spmv_partials_unsplit(sp, bs);
    return spmv_grid(sp->halo.n_interior, sp->n_slices) + (sp->halo.n_boundary ? spmv_grid(sp->halo.n_boundary, sp->n_slices) : 0);
}

// y = A x on a decomposed space with the ghost refresh of x hidden behind the interior rows:
//   pack + grouped send/recv on the communication stream | interior slices on the compute stream
//   compute stream waits for the halo                     | boundary slices
// One GPU (no halo plan): the plain product.
template <int DOTS>
static int spmv_overlapped(fs_matrix_s* A, double* x, double* y, const double* rvec, double* partials, int* status,
                           hipStream_t s, const double* val_override = nullptr) {
    fs_space_s* sp = A->space;
    if (!spmv_is_split(sp)) {
        if (sp->halo.begun) {           // started by the caller ahead of this product (pipelined CG)
            FS_CHECK(fs_halo_end_dev(sp, s));
            sp->halo.begun = false;
        } else
            FS_CHECK(fs_halo_exchange_dev(sp, x, s));
        launch_spmv<DOTS>(A, x, y, rvec, partials, status, s, val_override);
        return FS_OK;
    }
    const fs_halo_plan& h = sp->halo;
    const int gi = spmv_grid(h.n_interior, sp->n_slices), total = spmv_partials(sp);
    if (!sp->halo.begun) FS_CHECK(fs_halo_begin_dev(sp, x, s));     // (begun: started right after the rows it sends were updated)
    launch_spmv<DOTS>(A, x, y, rvec, partials, status, s, val_override, h.interior.p, h.n_interior, 0, total);
    FS_CHECK(fs_halo_end_dev(sp, s));
    sp->halo.begun = false;
    if (h.n_boundary) launch_spmv<DOTS>(A, x, y, rvec, partials, status, s, val_override, h.boundary.p, h.n_boundary, gi, total, 0);
    return FS_OK;
}

#include "fs_krylov_block4.inc"      // the product of 4 x 4-block Taylor-Hood operators (k_sell_spmv4_rows, k_sell_spmv4_ksplit)

int fs_spmv_dev(fs_matrix_s* A, const double* x, double* y, hipStream_t s) {
    if (A->bs == 4 && !getenv("FS_SPMV4_GENERIC")) {
        fs_space_s* sp = A->space;
        // one slice per workgroup while the grid allows it (dynamic balance)
        const int grid = (int)std::min<int64_t>(sp->n_slices, 65535);
        static const bool no_th = getenv("FS_SPMV4_NO_TH") != nullptr;
        const bool th = A->taylor_hood && sp->degree == 2 && !no_th;
        const int64_t nvo = sp->mesh->n_owned, gv0 = sp->n_nodes_owned, gv1 = sp->n_nodes_owned + (sp->mesh->nv - sp->mesh->n_owned);
#define FS_SPMV4_ARGS dim3(grid), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->n_nodes_local, sp->n_slices, sp->slice_ptr.p, sp->sell_col.p, \
                      sp->dia_ptr.p, sp->dia_off.p, A->val.p, sp->sell_entries, x, y, nvo, gv0, gv1
        static const bool ksplit = !(getenv("FS_SPMV4_KSPLIT") && getenv("FS_SPMV4_KSPLIT")[0] == '0');
        if (ksplit) {
            if (spmv_nontemporal(sp, 4)) {
                if (th) hipLaunchKernelGGL((k_sell_spmv4_ksplit<true, true>), FS_SPMV4_ARGS);
                else hipLaunchKernelGGL((k_sell_spmv4_ksplit<true, false>), FS_SPMV4_ARGS);
            } else {
                if (th) hipLaunchKernelGGL((k_sell_spmv4_ksplit<false, true>), FS_SPMV4_ARGS);
                else hipLaunchKernelGGL((k_sell_spmv4_ksplit<false, false>), FS_SPMV4_ARGS);
            }
        } else if (spmv_nontemporal(sp, 4)) {
            if (th) hipLaunchKernelGGL((k_sell_spmv4_rows<true, true>), FS_SPMV4_ARGS);
            else hipLaunchKernelGGL((k_sell_spmv4_rows<true, false>), FS_SPMV4_ARGS);
        } else {
            if (th) hipLaunchKernelGGL((k_sell_spmv4_rows<false, true>), FS_SPMV4_ARGS);
            else hipLaunchKernelGGL((k_sell_spmv4_rows<false, false>), FS_SPMV4_ARGS);
        }
#undef FS_SPMV4_ARGS
        return FS_OK;
    }
    launch_spmv<0>(A, x, y, nullptr, nullptr, nullptr, s);
    return FS_OK;
}

// fs_amg.hip: the products of one fs_amg_solve through the row dictionary where the fine operator's rows repeat (scalar and
// 3 x 3 block operators of uniform boxes); fs_dict_end drops the table (row_dict_scope by hand: the call sites are in another file)
int fs_dict_begin(fs_matrix_s* A, hipStream_t s) {
    g_dict.built_for = nullptr;
    return dict_build(A, A->val.p, s);
}
void fs_dict_end() { g_dict.built_for = nullptr; }
int fs_dict_classes() { return g_dict.built_for ? g_dict.ncls : 0; }

extern "C" int fs_last_product_kind(void) { return g_last_product_kind; }

extern "C" int fs_spmv(fs_matrix_t A, fs_vector_t x, fs_vector_t y) {
    FS_REQUIRE(A && x && y, "fs_spmv: null pointer");
    fs_space_s* sp = A->space;
    FS_REQUIRE(x->d.n >= sp->n_dofs_local, "fs_spmv: x has %lld entries, needs %lld (owned + ghost)", (long long)x->d.n, (long long)sp->n_dofs_local);
    FS_REQUIRE(y->d.n >= sp->n_dofs_owned, "fs_spmv: y too short");
    hipStream_t s = fs_rt().stream;
    if (A->bs == 1 && sp->n_pairs < 0 && spmv_use_pairs(sp, 1)) FS_CHECK(build_pair_lists(sp, s));
    FS_CHECK(fs_halo_exchange_dev(sp, x->d.p, s));
    launch_spmv<0>(A, x->d.p, y->d.p, nullptr, nullptr, nullptr, s);
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

// y = A x through the row-dictionary product where the rows of A repeat (classes found from A's values in this call and dropped
// at its end), else the streaming product; *row_classes = number of distinct rows used (0: streaming).  The two products agree bit
// for bit (same offsets, same order of summation) - this entry point exists so that tests and users can check exactly that.
extern "C" int fs_spmv_dictionary(fs_matrix_t A, fs_vector_t x, fs_vector_t y, int* row_classes) {
    std::lock_guard<std::recursive_mutex> solve_lock(fs_solve_mutex());
    FS_CHECK(fs_require_init());
    FS_REQUIRE(A && x && y, "fs_spmv_dictionary: null pointer");
    fs_space_s* sp = A->space;
    FS_REQUIRE(x->d.n >= sp->n_dofs_local, "fs_spmv_dictionary: x has %lld entries, needs %lld (owned + ghost)", (long long)x->d.n, (long long)sp->n_dofs_local);
    FS_REQUIRE(y->d.n >= sp->n_dofs_owned, "fs_spmv_dictionary: y too short");
    hipStream_t s = fs_rt().stream;
    row_dict_scope dict_scope;
    if (A->bs == 1 || A->bs == 3) FS_CHECK(dict_build(A, A->val.p, s));
    if (A->bs == 1 && !g_dict.built_for && sp->n_pairs < 0 && spmv_use_pairs(sp, 1)) FS_CHECK(build_pair_lists(sp, s));
    FS_CHECK(fs_halo_exchange_dev(sp, x->d.p, s));
    if (g_dict.built_for) launch_spmv<0>(A, x->d.p, y->d.p, nullptr, nullptr, nullptr, s);
    else FS_CHECK(fs_spmv_dev(A, x->d.p, y->d.p, s));
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(s));
    if (row_classes) *row_classes = g_dict.built_for ? g_dict.ncls : 0;
    return FS_OK;
}

extern "C" int fs_spmv_benchmark(fs_matrix_t A, fs_vector_t x, fs_vector_t y, int reps, double* ms_per_launch) {
    FS_REQUIRE(A && x && y && ms_per_launch && reps != 0, "fs_spmv_benchmark: bad arguments");
    // reps < 0: time the CG flavour (SpMV fused with the three dot products, r := y)
    const bool fused = reps < 0;
    if (fused) reps = -reps;
    fs_space_s* sp = A->space;
    FS_REQUIRE(x->d.n >= sp->n_dofs_local && y->d.n >= sp->n_dofs_owned, "fs_spmv_benchmark: vector too short");
    hipStream_t s = fs_rt().stream;
    if (A->bs == 1 && sp->n_pairs < 0 && spmv_use_pairs(sp, 1)) FS_CHECK(build_pair_lists(sp, s));
    hipEvent_t e0, e1;
    FS_HIP(hipEventCreate(&e0));
    FS_HIP(hipEventCreate(&e1));
    dbuf<double> partials, w;
    dbuf<int> status;
    FS_CHECK(partials.alloc(4 * (FS_MAX_PARTIAL_BLOCKS + 8)));
    FS_CHECK(status.alloc(4));
    FS_CHECK(status.zero(s));
    if (fused) FS_CHECK(w.alloc(sp->n_dofs_owned + 2));
    auto go = [&]() {
        if (fused) launch_spmv<1>(A, x->d.p, w.p, y->d.p, partials.p, status.p, s);
        else (void)fs_spmv_dev(A, x->d.p, y->d.p, s);      // the kernel the solvers use (block matrices: row-split)
    };
    go();  // warm-up
    FS_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) go();
    FS_HIP(hipEventRecord(e1, s));
    FS_HIP(hipEventSynchronize(e1));
    FS_KERNEL_CHECK();
    float ms = 0.f;
    FS_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = (double)ms / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return FS_OK;
}

// persistent Krylov workspace: allocation stays out of the timed solve
struct krylov_ws {
    int64_t n = 0, nl = 0;
    int hist_cap = 0;
    dbuf<double> dinv, r, z, w, p, s, partials, sums, ctrl, scal, hist;
    dbuf<double> rhat, t, y, partials2, bsums;   // BiCGStab only (allocated on first use)
    dbuf<double> aval, dvec, bhat, sc_local;     // diagonally scaled CG only
    dbuf<double> pw, pz;                         // pipelined CG only: w = A r (with ghost room: it is the exchanged vector), z = A s
    hipEvent_t ev_upd = nullptr, ev_red = nullptr;   // pipelined CG: update done -> all-reduce on the communication stream -> sums ready
    dbuf<int> d_err;                             // zero-diagonal counter of k_extract_dinv
    int64_t bicg_n = -1;
    dbuf<int> status;
    int* h_status = nullptr;  // pinned: 2 x 4 status ints (double-buffered polls) + [8] zero-diagonal count
    int* h_mirror = nullptr;  // pinned, written by the iteration kernel's leader lane: [0] status once stopped, [1] iteration in progress
    int* d_mirror = nullptr;  // ... its device address
    bool mirror_ok = true;    // false once a wait on it timed out (stores not visible on this system): the copied status word again
    double* h_vals = nullptr; // pinned: the sums [0 .. 8) and the control block [8 .. 12) at the end of a pass
    hipEvent_t poll[2] = {nullptr, nullptr};
    static const int NSAMPLE = 64;
    hipEvent_t ev[NSAMPLE][4];
    int sample_iter[NSAMPLE];   // iteration (within its pass) a sample was taken at
    bool sample_live[NSAMPLE];  // false: the sampled launches came after convergence (no-ops)
    bool events = false;
    std::vector<double> last_hist;
    // one batch of CG iterations captured as a hipGraph (same arguments every iteration: the update kernel reads its
    // iteration index from the device).  Re-instantiated when anything it bakes in changes.
    dbuf<double> sg;            // s on the ghost rows, in arrival order (peer-to-peer iteration: a rank advances its ghost r, s itself)
    hipGraphExec_t cg_graph = nullptr;
    const void* cg_key[32] = {};
    int64_t cg_key_i[8] = {};
    // one-launch iteration (k_dict_cg_iter): the second set of r, w, s (double-buffered by iteration parity), the iteration
    // counters of the device, its own captured batch
    dbuf<double> z2, w2, s2;
    dbuf<int> it_ctr;
    hipGraphExec_t cgf_graph = nullptr;
    const void* cgf_key[24] = {};
    int64_t cgf_key_i[8] = {};
};
static krylov_ws g_ws;

static int ws_prepare(krylov_ws& ws, int64_t n, int64_t nl, int max_iter) {
    if (ws.n != n || ws.nl != nl) {
        // (measured and not kept: the vectors of the update set apart by 256 B ... 1 MB so that equal
        // indices fall on different memory channels - no effect; the 100 - 117 us spread of the update kernel at 10 M rows is
        // the box's, not the addresses')
        FS_CHECK(ws.dinv.alloc(n + 2));
        FS_CHECK(ws.r.alloc(n + 2));
        FS_CHECK(ws.z.alloc(nl + 2));
        FS_CHECK(ws.w.alloc(nl + 2));        // (ghost room: the one-launch iteration on a decomposed space reads w and s on ghost columns)
        FS_CHECK(ws.p.alloc(n + 2));
        FS_CHECK(ws.s.alloc(nl + 2));
        ws.n = n;
        ws.nl = nl;
    }
    if (!ws.partials.p) {
        FS_CHECK(ws.partials.alloc(4 * (FS_MAX_PARTIAL_BLOCKS + 8)));
        FS_CHECK(ws.sums.alloc(8));
        FS_CHECK(ws.ctrl.alloc(4));
        FS_CHECK(ws.scal.alloc(4));
        FS_CHECK(ws.status.alloc(4));
        FS_CHECK(ws.d_err.alloc(1));
        FS_HIP(hipHostMalloc((void**)&ws.h_status, 16 * sizeof(int), hipHostMallocDefault));     // ([12 .. 16): the status word at the end of a pass)
        FS_HIP(hipHostMalloc((void**)&ws.h_vals, 16 * sizeof(double), hipHostMallocDefault));
        if (hipHostMalloc((void**)&ws.h_mirror, 16 * sizeof(int), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&ws.d_mirror, ws.h_mirror, 0) != hipSuccess) {
            (void)hipGetLastError();
            ws.h_mirror = ws.d_mirror = nullptr;
        }
        FS_HIP(hipEventCreateWithFlags(&ws.poll[0], hipEventDisableTiming));
        FS_HIP(hipEventCreateWithFlags(&ws.poll[1], hipEventDisableTiming));
        FS_HIP(hipEventCreateWithFlags(&ws.ev_upd, hipEventDisableTiming));
        FS_HIP(hipEventCreateWithFlags(&ws.ev_red, hipEventDisableTiming));
        for (int i = 0; i < krylov_ws::NSAMPLE; ++i)
            for (int j = 0; j < 4; ++j) FS_HIP(hipEventCreate(&ws.ev[i][j]));
        ws.events = true;
    }
    if (ws.hist_cap < max_iter + 2) {
        FS_CHECK(ws.hist.alloc(max_iter + 2));
        ws.hist_cap = max_iter + 2;
    }
    return FS_OK;
}

extern "C" int fs_krylov_solve(fs_matrix_t A, fs_vector_t b, fs_vector_t x, const fs_krylov_opts* opts,
                               fs_krylov_stats* stats) {
    std::lock_guard<std::recursive_mutex> solve_lock(fs_solve_mutex());
    FS_CHECK(fs_require_init());
    FS_REQUIRE(A && b && x && opts, "fs_krylov_solve: null pointer");
    if (opts->method != FS_KSP_CG && opts->method != FS_KSP_BICGSTAB) {
        fs_set_error("fs_krylov_solve: unknown Krylov method %d", opts->method);
        return FS_ERR_UNSUPPORTED;
    }
    const bool bicg = opts->method == FS_KSP_BICGSTAB;
    row_dict_scope dict_scope;           // the row dictionary describes the values of ONE solve (built below where it applies)
    FS_REQUIRE(A->bs != 4, "fs_krylov_solve: Taylor-Hood block systems are solved by fs_saddle_solve");
    FS_REQUIRE(opts->precond == FS_PC_NONE || opts->precond == FS_PC_JACOBI, "fs_krylov_solve: unknown preconditioner %d", opts->precond);
    FS_REQUIRE(opts->max_iter > 0 && opts->rtol >= 0.0 && opts->atol >= 0.0, "fs_krylov_solve: bad tolerances");
    fs_space_s* sp = A->space;
    const int64_t n = sp->n_dofs_owned, nl = sp->n_dofs_local;
    FS_REQUIRE(b->d.n >= n && x->d.n >= n, "fs_krylov_solve: b/x shorter than the owned dofs (%lld)", (long long)n);
    hipStream_t s = fs_rt().stream;
    // A scalar CG2 operator on a uniform box (one GPU) is solved in LATTICE order (fs_lattice.hip): values, b and x permuted into
    // the solver's shadow of the space, the solve run there (this function again, on the shadow's handles), x permuted back.
    // Automatic (option "lattice_order" = -1, the default): from FS_LATTICE_MIN_ROWS rows on, and only as long as the shadow's solves
    // run on the tile product (a shadow whose rows do not fit the tile form - lat_prepare - is given up after its first solve).
    if (bs_is_scalar_cg2(A) && (g_lattice > 0 || (g_lattice < 0 && sp->n_nodes_owned >= FS_LATTICE_MIN_ROWS)) && sp->lattice_state >= 0) {
        fs_lattice_shadow* L = nullptr;
        FS_CHECK(fs_lattice_get(sp, &L));
        if (L) {
            const auto t0 = std::chrono::steady_clock::now();
            fs_matrix_s* A2 = nullptr;
            fs_vector_s *b2 = nullptr, *x2 = nullptr;
            FS_CHECK(fs_lattice_enter(L, A, b, x, opts->nonzero_guess != 0, &A2, &b2, &x2));
            // (the shadow matrix has ONE serial and holds another operator's values on every call: the dictionary's marker `these rows
            // do not repeat`, keyed on the serial, must not outlive the matrix it was set for - ADVICE r5)
            if (g_dict.gave_up_on == A2->serial) g_dict.gave_up_on = 0;
            g_lat.ok = false;           // (decided by THIS solve: a flag left by an earlier solve on another space says nothing)
            g_lat.judged = false;
            FS_CHECK(fs_krylov_solve(A2, b2, x2, opts, stats));
            FS_CHECK(fs_lattice_leave(L, sp, x));
            FS_HIP(hipStreamSynchronize(s));
            if (g_lattice < 0 && g_lat.judged && !(g_lat.ok && g_lat.space_serial == A2->space->serial)) {
                // the shadow's rows do not fit the tile form: the space's own numbering from now on, and the shadow (a second copy of
                // the operator's structure and values: several GB at configs[3]) is given back
                sp->lattice_state = -1;
                fs_lattice_release(sp->lattice);
                sp->lattice = nullptr;
            }
            if (stats) {
                stats->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                stats->lattice_order = 1;
            }
            return FS_OK;
        }
    }
    // FS_SOLVE_TIMING=1: wall-clock laps of the phases of a solve on stderr (each lap synchronises the stream: a diagnostic, it
    // changes what it measures; tools/probes/first_step_probe.py uses it to split the first solve of a process)
    static const bool lap_on = getenv("FS_SOLVE_TIMING") != nullptr;
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!lap_on) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[fs_krylov timing] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - lap_t).count());
        lap_t = now;
    };
    krylov_ws& ws = g_ws;
    FS_CHECK(ws_prepare(ws, n, nl, opts->max_iter));
    lap("workspace");
    if (sp->halo.begun) {              // left over from a solve that ended in an error
        FS_CHECK(fs_halo_end_dev(sp, fs_rt().stream));
        sp->halo.begun = false;
    }
    if (bicg && (ws.bicg_n != n || ws.y.n != nl + 2)) {
        FS_CHECK(ws.rhat.alloc(n + 2));
        FS_CHECK(ws.t.alloc(n + 2));
        FS_CHECK(ws.y.alloc(nl + 2));
        FS_CHECK(ws.partials2.alloc(4 * (FS_MAX_PARTIAL_BLOCKS + 8)));
        FS_CHECK(ws.bsums.alloc(16));
        ws.bicg_n = n;
    }
    const int bs = A->bs;
    const bool ds = !bicg && opts->precond == FS_PC_JACOBI && opts->diagonal_scale != 0;
    if (ds) {
        if (ws.aval.n != A->val.n) FS_CHECK(ws.aval.alloc(A->val.n));
        if (ws.dvec.n != n + 2) {
            FS_CHECK(ws.dvec.alloc(n + 2));
            FS_CHECK(ws.bhat.alloc(n + 2));
        }
        if (ws.sc_local.n != nl + 2) FS_CHECK(ws.sc_local.alloc(nl + 2));
    }
    // Pipelined recurrence (k_pcg_update): only when asked for explicitly (> 0).
    static const char* pipe_env = getenv("FS_CG_PIPELINED");
    const int pipe_opt = pipe_env ? atoi(pipe_env) : opts->pipelined;
    if (pipe_opt > 0 && !ds) {
        fs_set_error("fs_krylov_solve: the pipelined recurrence needs CG + Jacobi with diagonal_scale = 1");
        return FS_ERR_UNSUPPORTED;
    }
    // OPT-IN since round 4 (opts->pipelined = 1, FS_CG_PIPELINED=1): in every measurement available the recurrence lost - 69.3
    // against 57.1 us per iteration over RCCL, 58.8 against 38.0 us over the peer-to-peer exchange at 1 M rows per rank
    // (profiles/r03_p2p_self_halo_timings.txt): its 40 B/DOF of extra vector traffic and two more launches cost more than the
    // 3-double all-reduce it hides.  It would pay from an all-reduce latency of about 20 us per iteration upwards (DESIGN.md
    // section 5) - a multi-node communicator, which this library does not target.
    const bool pipelined = ds && pipe_opt > 0;
    if (pipelined) {
        if (ws.pw.n != nl + 2) FS_CHECK(ws.pw.alloc(nl + 2));
        if (ws.pz.n != n + 2) FS_CHECK(ws.pz.alloc(n + 2));
    }
    const bool fuse_sums = g_cg_fuse_sums && fs_rt().comm == nullptr;
    const int vgrid = fs_grid_for(n / 2 + 1, FS_BLOCK, g_update_blocks);
    const int pgrid = fs_grid_for(n, FS_BLOCK, FS_MAX_PARTIAL_BLOCKS);
    if (A->bs == 1 && sp->n_pairs < 0 && spmv_use_pairs(sp, 1)) FS_CHECK(build_pair_lists(sp, s));
    int sgrid = spmv_partials(sp, A->bs);     // dot partials per product (pairs + singles, or interior + boundary on a decomposed space)
    const auto t_begin = std::chrono::steady_clock::now();

    // Jacobi diagonal (the zero-diagonal counter is read back with the first status poll: no extra sync here)
    {
        FS_CHECK(ws.d_err.zero(s));
        const int g = fs_grid_for(sp->n_nodes_owned);
        const int jmode = ds ? 2 : (opts->precond == FS_PC_JACOBI ? 1 : 0);
        double* dv = ds ? ws.dvec.p : nullptr;
        if (bs == 2)
            hipLaunchKernelGGL(k_extract_dinv<2>, dim3(g), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->slice_ptr.p, sp->sell_col.p, A->val.p, sp->sell_entries, jmode, ws.dinv.p, ws.d_err.p, dv);
        else if (bs == 1)
            hipLaunchKernelGGL(k_extract_dinv<1>, dim3(g), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->slice_ptr.p, sp->sell_col.p, A->val.p, sp->sell_entries, jmode, ws.dinv.p, ws.d_err.p, dv);
        else
            hipLaunchKernelGGL(k_extract_dinv<3>, dim3(g), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->slice_ptr.p, sp->sell_col.p, A->val.p, sp->sell_entries, jmode, ws.dinv.p, ws.d_err.p, dv);
        FS_KERNEL_CHECK();
        FS_HIP(hipMemcpyAsync(ws.h_status + 8, ws.d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
    }
    const bool pnorm = opts->norm_type == FS_NORM_PRECONDITIONED;
    if (pnorm && !ds) {
        fs_set_error("fs_krylov_solve: the preconditioned residual norm needs CG + Jacobi with diagonal_scale = 1");
        return FS_ERR_UNSUPPORTED;
    }
    // ||b||^2 -> threshold.  Preconditioned norm (PETSc's KSPCG default): ||D^-1 b||^2 = sum (dinv_s^2 b)^2,
    // where ws.dinv holds 1/sqrt(d) in scaled mode, and the residual weights become 1/d instead of d.
    if (pnorm) {
        hipLaunchKernelGGL(k_pointwise_mul, dim3(fs_grid_for(n)), dim3(FS_BLOCK), 0, s, ws.dinv.p, ws.dinv.p, n, ws.dvec.p);  // 1/d
        hipLaunchKernelGGL(k_pointwise_mul, dim3(fs_grid_for(n)), dim3(FS_BLOCK), 0, s, ws.dvec.p, b->d.p, n, ws.bhat.p);    // b/d
        hipLaunchKernelGGL(k_dot_partial, dim3(pgrid), dim3(FS_BLOCK), 0, s, ws.bhat.p, ws.bhat.p, n, ws.partials.p);
    } else
    hipLaunchKernelGGL(k_dot_partial, dim3(pgrid), dim3(FS_BLOCK), 0, s, b->d.p, b->d.p, n, ws.partials.p);
    FS_CHECK(fs_comm_sum_allreduce_dev(ws.partials.p, pgrid, 1, ws.sums.p + 4, s));
    hipLaunchKernelGGL(k_set_threshold, dim3(1), dim3(64), 0, s, ws.sums.p + 4, opts->rtol, opts->atol, ws.ctrl.p);
    const double* aval = nullptr;
    lap("diagonal, |b|, threshold");
    if (ds) {
        // the scaled system needs ghost scale factors too: refresh them through the halo
        // (no halo plan: no ghost entries - the factors are read where they are)
        double* sc_local = sp->halo.active ? ws.sc_local.p : ws.dinv.p;
        if (sp->halo.active) {
            FS_HIP(hipMemcpyAsync(sc_local, ws.dinv.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
            FS_CHECK(fs_halo_exchange_dev(sp, sc_local, s));
        }
        const int g2 = fs_grid_for(sp->n_slices * 64, FS_BLOCK, 8192);
        const auto scale_copy1 = [&]() {
            hipLaunchKernelGGL(k_scale_copy<1>, dim3(g2), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->sell_col.p, A->val.p, sp->sell_entries, sc_local, ws.aval.p);
        };
        // scalar operator on one GPU: the scaled copy is written only if somebody is going to read it - a matrix that the kept class
        // table describes (verified on the fly-scaled values, dict_build) is multiplied from the table alone
        static const bool lazy_env = !(getenv("FS_LAZY_SCALE_COPY") && getenv("FS_LAZY_SCALE_COPY")[0] == '0');
        const bool lazy_copy = bs == 1 && !sp->halo.active && lazy_env;
        if (bs == 2)
            hipLaunchKernelGGL(k_scale_copy<2>, dim3(g2), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->sell_col.p, A->val.p, sp->sell_entries, sc_local, ws.aval.p);
        else if (bs == 1) {
            if (!lazy_copy) scale_copy1();
        } else
            hipLaunchKernelGGL(k_scale_copy<3>, dim3(g2), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->sell_col.p, A->val.p, sp->sell_entries, sc_local, ws.aval.p);
        hipLaunchKernelGGL(k_pointwise_mul, dim3(fs_grid_for(n)), dim3(FS_BLOCK), 0, s, ws.dinv.p, b->d.p, n, ws.bhat.p);
        FS_KERNEL_CHECK();
        aval = ws.aval.p;
        if (bs == 1) {
            // a handful of distinct rows (uniform box, constant coefficient)?
            if (lazy_copy) FS_CHECK(dict_build(A, aval, s, A->val.p, sc_local, scale_copy1));
            else FS_CHECK(dict_build(A, aval, s));
            sgrid = spmv_partials(sp, bs);          // (the row-dictionary product has its own launch geometry)
        }
    }
    lap("scaling + row classes");
    // One launch per iteration (k_dict_cg_iter) where the product is the row-dictionary kernel with the whole dictionary in LDS,
    // on one GPU.  Automatic mode: FS_CG_FUSED_MAX_ROWS rows at most (the neighbour values of three vectors instead of one have
    // to stay in the L2 of an XCD; measured crossover in DESIGN.md section 3).
    static const char* fused_env = getenv("FS_CG_FUSED");
    static const int64_t fused_max_rows = getenv("FS_CG_FUSED_MAX_ROWS") ? atoll(getenv("FS_CG_FUSED_MAX_ROWS")) : (int64_t)3000000;
    const int fused_opt = fused_env ? atoi(fused_env) : g_cg_fused;
    const bool fused_common = ds && !pipelined && bs == 1 && fused_opt != 0 &&
                              g_dict.bs == 1 && g_dict.built_for && g_dict.built_for == aval && sp->n_dict_items > 0 &&
                              (size_t)g_dict.ncls * g_dict.S * sizeof(double) <= (size_t)FS_DICT_WHOLE_LDS_BYTES &&
                              nl < ((int64_t)1 << 29) && sp->dict_run_len == 3 && sp->dict_runs == 8 && (fused_opt > 0 || n <= fused_max_rows);
    // (round 6) P1 boxes from 400 k rows on: the iteration in marching-window form (k_box_cg_iter), with its own launch geometry
    const box_iter_plan_s* BI = (fused_common && !sp->halo.active && fuse_sums) ? box_iter_plan_for(sp, g_dict.ncls) : nullptr;
    const int igrid = fused_common ? (BI ? BI->g.grid : spmv_partials_unsplit(sp, bs)) : 0;        // workgroups (= dot partials) of the iteration kernel
    const bool fused_sized = fused_common && 3 * (int64_t)igrid * 2 <= (int64_t)ws.partials.n && igrid <= 4 * FS_BLOCK;
    const bool fused = fused_sized && fuse_sums && !sp->halo.active;
    // a decomposed space: the same kernel after the peer-to-peer exchange kernel (two launches per iteration instead of three) - where
    // that exchange is what the iteration uses (decided per pass below: p2p_fuse); a rank-local choice, the exchange protocol is the same
    static const bool fused_p2p_on = !(getenv("FS_CG_FUSED_P2P") && getenv("FS_CG_FUSED_P2P")[0] == '0');
    const bool fusedp_ok = fused_sized && !fuse_sums && sp->halo.active && fused_p2p_on;
    if (fused || fusedp_ok) {
        if (ws.z2.n != nl + 2) FS_CHECK(ws.z2.alloc(nl + 2));
        if (ws.w2.n != nl + 2) {
            FS_CHECK(ws.w2.alloc(nl + 2));
            FS_CHECK(ws.s2.alloc(nl + 2));
        }
        if (!ws.it_ctr.p) FS_CHECK(ws.it_ctr.alloc(2));
    }
    // A pass = fresh recurrences from the current x.  The single-reduction recurrences drift on
    // ill-conditioned operators (the recurrence residual can reach the threshold while b - A x has not):
    // the true residual is recomputed after every pass and, if it misses the tolerance, the solve
    // restarts from the current x (at most 8 passes, iteration budget shared).
    static const int env_batch = getenv("FS_CG_BATCH") ? atoi(getenv("FS_CG_BATCH")) : 0;
    const int batch = opts->batch > 0 ? opts->batch : (env_batch > 0 ? env_batch : g_cg_batch);
    // kernel durations (stats->spmv_ms / update_ms: the roofline of bench.py) are sampled with HIP events every
    // sample_every-th iteration, 4 events each.  The markers are not free: on the 1 M-DOF solve (293 iterations of 45 us)
    // sampling every 4th iteration costs 0.87 ms per solve (6 %), every 16th 0.4 ms - the default
    static const int sample_every = getenv("FS_CG_SAMPLE_EVERY") ? std::max(1, atoi(getenv("FS_CG_SAMPLE_EVERY"))) : 16;
    static const char* upd_nt_env = getenv("FS_UPDATE_NT");
    const bool upd_nt = upd_nt_env ? upd_nt_env[0] == '1' : (int64_t)sp->n_dofs_owned * 72 > ((int64_t)192 << 20);   // five vectors exceed the caches
    int total_iters = 0, n_samples = 0, n_pass = 0, n_launches = 0;
    bool fusedp_used = false;
    int h_status[4] = {0, 0, 0, 0};
    bool use_guess = opts->nonzero_guess != 0;
    double true_rr = 0.0, thresh = 0.0, bb_host = 0.0, prev_true_rr = 1e300;
    for (;;) {
        // initial state
        FS_CHECK(ws.status.zero(s));
        FS_CHECK(ws.scal.zero(s));
        // p, s, z and - from a zero guess - x in one launch
        hipLaunchKernelGGL(k_zero4, dim3(fs_grid_for(std::max<int64_t>(nl / 2, 1), FS_BLOCK, 1024)), dim3(FS_BLOCK), 0, s, ws.p.p, ws.p.n, ws.s.p, ws.s.n,
                           ws.z.p, ws.z.n, x->d.p, use_guess ? (int64_t)0 : x->d.n);
        if (sp->halo.active) {          // s on the ghost rows (k_cg_p2p_exchange)
            if (ws.sg.n < nl - n + 2) FS_CHECK(ws.sg.alloc(nl - n + 2));
            FS_CHECK(ws.sg.zero(s));
        }
        if (ds) {
            // scaled unknown xhat = D^1/2 x lives in the caller's x until the final un-scaling; rhat in ws.z
            if (use_guess) {
                if (n_pass == 0) hipLaunchKernelGGL(k_pointwise_div, dim3(fs_grid_for(n)), dim3(FS_BLOCK), 0, s, x->d.p, ws.dinv.p, n, x->d.p);
                FS_HIP(hipMemcpyAsync(ws.z.p, x->d.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
                FS_CHECK(fs_halo_exchange_dev(sp, ws.z.p, s));
                launch_spmv<0>(A, ws.z.p, ws.w.p, nullptr, nullptr, nullptr, s, aval);
                hipLaunchKernelGGL(k_residual, dim3(pgrid), dim3(FS_BLOCK), 0, s, ws.bhat.p, ws.w.p, n, ws.z.p, ws.partials.p);
            } else {
                FS_HIP(hipMemcpyAsync(ws.z.p, ws.bhat.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
            }
        } else {
            if (use_guess) {
                FS_HIP(hipMemcpyAsync(ws.z.p, x->d.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
                FS_CHECK(fs_halo_exchange_dev(sp, ws.z.p, s));
                launch_spmv<0>(A, ws.z.p, ws.w.p, nullptr, nullptr, nullptr, s);
                hipLaunchKernelGGL(k_residual, dim3(pgrid), dim3(FS_BLOCK), 0, s, b->d.p, ws.w.p, n, ws.r.p, ws.partials.p);
            } else {
                FS_HIP(hipMemcpyAsync(ws.r.p, b->d.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
            }
            hipLaunchKernelGGL(k_pointwise_mul, dim3(fs_grid_for(n)), dim3(FS_BLOCK), 0, s, ws.dinv.p, ws.r.p, n, ws.z.p);
        }
        FS_KERNEL_CHECK();
        // pipelined CG: the all-reduce of the sums runs on the communication stream of the halo plan, behind ev_upd, and the
        // compute stream waits for ev_red only when the next update needs the sums - the product sits in between.  Without
        // a halo plan (a replicated operator) the collective stays in-stream.
        hipStream_t red_stream = nullptr;
        // (the peer-to-peer all-reduce is a kernel of a few microseconds: it stays in the compute stream, a second stream's two event
        // hops cost more than it hides)
        if (pipelined && !fuse_sums && sp->halo.active && !fs_p2p_reduce_enabled()) FS_CHECK(fs_halo_comm_stream(sp, &red_stream));
        auto pcg_reduce = [&](int parity) -> int {
            hipStream_t q = red_stream ? red_stream : s;
            if (red_stream) {
                FS_HIP(hipEventRecord(ws.ev_upd, s));
                FS_HIP(hipStreamWaitEvent(red_stream, ws.ev_upd, 0));
            }
            FS_CHECK(fs_comm_sum_allreduce_dev(ws.partials.p + (int64_t)parity * 3 * vgrid, vgrid, 3, ws.sums.p, q));
            if (red_stream) FS_HIP(hipEventRecord(ws.ev_red, red_stream));
            return FS_OK;
        };
        if (pipelined) {
            // r0 sits in ws.z; w0 = A r0, z = 0 (p, s are zero already), sums of (r0, w0)
            FS_CHECK(ws.pz.zero(s));
            FS_CHECK(fs_halo_exchange_dev(sp, ws.z.p, s));
            launch_spmv<0>(A, ws.z.p, ws.pw.p, nullptr, nullptr, nullptr, s, aval);
            hipLaunchKernelGGL(k_pcg_dots, dim3(vgrid), dim3(FS_BLOCK), 0, s, n, ws.z.p, ws.pw.p, ws.dvec.p, ws.partials.p, vgrid);
            FS_KERNEL_CHECK();
            if (!fuse_sums) FS_CHECK(pcg_reduce(0));
        }
        if (bicg) {
            // rhat = r0; first (rhat.r, r.r) partials; v = 0 (ws.w), p = 0, y = 0
            FS_HIP(hipMemcpyAsync(ws.rhat.p, ws.r.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
            FS_CHECK(ws.w.zero(s));
            FS_CHECK(ws.y.zero(s));
            hipLaunchKernelGGL(k_dot2_partial, dim3(vgrid), dim3(FS_BLOCK), 0, s, ws.rhat.p, ws.r.p, ws.r.p, ws.r.p, n, ws.partials2.p);
            FS_KERNEL_CHECK();
        }

        // iteration pipeline
        const int max_iter = opts->max_iter - total_iters > 0 ? opts->max_iter - total_iters : 1;
        hipLaunchKernelGGL(k_set_iteration_limit, dim3(1), dim3(64), 0, s, ws.ctrl.p, max_iter);
        double* const hist_p = ws.hist.p + total_iters;
        int k = 0, slot = 0, pending = -1;
        bool finished = false;
        const int first_sample = n_samples;
        // Batches after the first (which carries the event samples) go out as ONE hipGraphLaunch of 2 x batch kernel nodes
        // on cache-resident problems: the gaps between consecutive launches (2 x 2.1 us of a 44 us iteration at 1 M DOF)
        // shrink to the graph's own node-to-node latency.
        static const char* graph_env = getenv("FS_CG_GRAPH");
        // rocprofv3 (ROCm 7.2) segfaults inside hipGraphLaunch once about 10 500 kernel nodes have been replayed under
        // --kernel-trace (168 launches of this 64-node graph; nothing to do with the graph's contents or age - renewing
        // the executable does not help, plain launches of the same kernels trace fine).  With the profiler's tool library
        // in the process the automatic mode therefore falls back to plain launches: the kernels and their durations are
        // the same, only the 2 x 2 us of launch gap per iteration come back.  FS_CG_GRAPH=1 still forces graphs.
        static const bool profiler_attached = getenv("ROCP_TOOL_LIBRARIES") != nullptr;
        int graph_mode = graph_env ? atoi(graph_env) : g_cg_graph;
        if (graph_mode < 0 && profiler_attached) {
            static bool told = false;
            if (!told) fprintf(stderr, "[libfsamd] rocprofiler tool library detected: CG batches go out as plain launches, not hipGraphs\n");
            told = true;
            graph_mode = 0;
        }
        // rows sent to the neighbours as a prefix [0, early_a) and / or a suffix [early_b, n) of the owned rows (z-slabs): see
        // the update below.  FS_HALO_EARLY=0 keeps the exchange inside the product.
        int64_t early_a = 0, early_b = n;
        int p2p_fuse = 0, p2p_rows_cap = 128;
        bool p2p_ghosts_in = false;      // fused peer-to-peer iteration: the ghosts of the next product were received by the rows kernel
        if (ds && !bicg && !fuse_sums && sp->halo.active) {     // (a condition every rank evaluates alike; !fuse_sums: a communicator is up)
            static const bool no_early = getenv("FS_HALO_EARLY") && getenv("FS_HALO_EARLY")[0] == '0';
            fs_halo_plan& hp = sp->halo;
            if (hp.early < 0) {
                bool ok = !no_early && spmv_is_split(sp);
                int64_t a = 0, b2 = n;
                for (size_t i = 0; ok && i < hp.neighbors.size(); ++i) {
                    if (hp.send_counts[i] <= 0) continue;
                    const int64_t first = hp.send_first[i], last = first + hp.send_counts[i];
                    if (!hp.send_contiguous[i]) ok = false;
                    else if (first == 0) a = std::max(a, last);
                    else if (last == n) b2 = std::min(b2, first);
                    else ok = false;
                }
                a = (a + 1) & ~(int64_t)1;          // the bulk update works on 16-byte pairs
                ok = ok && a < b2 && (a > 0 || b2 < n);
                // every rank has to take the same path: an exchange begun by one side only would never be matched
                // (+ 1024 per rank with the peer-to-peer exchange on: the kernel of the iteration below is gated by the status word,
                // the separate send / receive kernels are not - a mix would leave one side waiting)
                double flag = (ok ? 1.0 : 0.0) + (fs_p2p_fusable(sp) ? 1024.0 : 0.0);
                FS_HIP(hipMemcpyAsync(ws.sums.p + 6, &flag, sizeof(double), hipMemcpyHostToDevice, s));
                FS_CHECK(fs_comm_allreduce_dev(ws.sums.p + 6, 1, s));
                FS_HIP(hipMemcpyAsync(&flag, ws.sums.p + 6, sizeof(double), hipMemcpyDeviceToHost, s));
                FS_HIP(hipStreamSynchronize(s));
                const int nr_all = fs_rt().n_ranks, agreed = (int)(flag + 0.5);
                hp.early = agreed % 1024 == nr_all ? 1 : 0;
                hp.fuse = agreed / 1024 == nr_all ? 1 : 0;
                hp.early_a = a;
                hp.early_b = b2;
                if (getenv("FS_KRYLOV_DEBUG"))
                    fprintf(stderr, "[fs_krylov] rank %d: rows [0,%lld) and [%lld,%lld) are sent; early start of the exchange: %s\n", fs_rt().rank,
                            (long long)a, (long long)b2, (long long)n, hp.early == 1 ? "yes" : "no");
            }
            if (hp.early == 1) { early_a = hp.early_a; early_b = hp.early_b; }
            // Peer-to-peer exchange on every rank: the iteration is THREE kernels of the compute stream - the plain product, the
            // exchange kernel (k_cg_p2p_exchange: all-reduce, send, receive), the plain update - instead of seven launches on two
            // streams (below).  FS_P2P_FUSE=0 keeps the separate send / receive / all-reduce kernels of the same transport around a
            // split product.  Measured alternatives that lost on MI355X: ghost columns read straight from the (uncached,
            // fine-grained) receive buffer by the boundary rows of one merged product (67 instead of 39 us per product at 1 M rows),
            // and the sums posted by the last workgroup of the product (its agent-scope fence in every workgroup writes back the
            // whole L2 of the XCD: + 26 us).
            static const int fuse_env = getenv("FS_P2P_FUSE") ? atoi(getenv("FS_P2P_FUSE")) : 1;
            static const int rows_env = getenv("FS_P2P_ROWS_BLOCKS") ? atoi(getenv("FS_P2P_ROWS_BLOCKS")) : 128;
            p2p_rows_cap = rows_env > 0 ? rows_env : 128;
            if (hp.fuse == 1 && !pipelined && fuse_env) p2p_fuse = 1;
        }
        // (the peer-to-peer iteration is three kernels and no library call, with its sequence numbers on the device: it is captured the
        // same way; the RCCL iteration is not - ncclSend / ncclRecv / ncclAllReduce are host calls)
        // (automatic mode: every size - the launch gaps are 10 % of an iteration at 1 M rows and still 1.5 % at 10 M)
        const bool graph_sized = graph_mode != 0;
        // The FIRST graph a process instantiates costs 9 ms (the runtime's graph machinery; later ones 0.1 ms), twelve times what the
        // graphs save a 300-iteration solve at 1 M rows: the first solve of a process launches its iterations one by one - a
        // one-shot run never pays, a time loop pays in its second step (tools/probes/first_step_probe.py; FS_CG_GRAPH=1 forces graphs)
        // (round 6: fs_init instantiates a two-node graph on a helper thread while the code objects load - the machinery is paid for
        // there, and the first solve of a process goes out in graphs like every other; FS_WARM=0 restores the old rule)
        static bool first_solve_done = !(getenv("FS_WARM") && getenv("FS_WARM")[0] == '0');
        const bool graphs_allowed = first_solve_done || graph_mode > 0;
        struct mark_done { bool& f; ~mark_done() { f = true; } } mark_first_solve{first_solve_done};
        const bool fusedp = fusedp_ok && p2p_fuse && !bicg;
        if (fusedp) fusedp_used = true;
        const bool use_graph = ds && !bicg && !pipelined && graph_sized && !fused && !fusedp &&
                               ((fuse_sums && !sp->halo.active && bs == 1) || p2p_fuse);
        // One-launch iteration on one GPU: the host follows the device through two words of pinned memory the kernel's leader lane
        // writes (iteration in progress, status once stopped) and keeps between cg_ahead and cg_ahead + cg_sub launches enqueued.
        // With batches of 32 and the status word copied back behind each, a solve of 293 iterations enqueued 352 launches - the
        // 59 that returned on the status word cost 5.3 us each, 0.31 of a 6.8 ms solve - and ten in-stream copies.
        static const int mirror_env = getenv("FS_CG_MIRROR") ? atoi(getenv("FS_CG_MIRROR")) : -1;
        // (the same for the two-launch iteration on one GPU - streaming products, row-dictionary products above 3 M rows -, whose
        // update kernel writes the words: its batches of 32 left up to 59 x 2 launches behind the last iteration)
        const bool two_launch_1gpu = use_graph && !p2p_fuse && fuse_sums && !sp->halo.active;
        bool mirrored = ((fused && !fusedp) || two_launch_1gpu) && ws.d_mirror && ws.mirror_ok && (mirror_env >= 0 ? mirror_env != 0 : g_cg_mirror != 0);
        int* const mirror_dev = mirrored ? ws.d_mirror : nullptr;
        volatile int* const hm = ws.h_mirror;
        if (mirrored) { hm[0] = 0; hm[1] = 0; }
        const int bsz = mirrored ? g_cg_sub : batch;
        // the launches that carry the event samples go out one by one before the graphs take over: the first 32 iterations (samples
        // at iterations 1 and 17), or - with the progress words - the first 16 (samples at 1 and 9)
        const int first_plain = mirrored ? std::min(batch, 16) : batch;
        const int sample_step = mirrored ? std::min(sample_every, 8) : sample_every;
        int seen = 0;
        auto t_seen = std::chrono::steady_clock::now();
        while (!finished) {
            if (mirrored && k > 0) {
                if (hm[0] != 0) break;                  // the recurrence has stopped: what is enqueued returns on the status word
                const int done = hm[1];
                if (k - done > g_cg_ahead) {            // enough enqueued: wait for the device to get on
                    if (done != seen) { seen = done; t_seen = std::chrono::steady_clock::now(); }
                    else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_seen).count() > 2.0) {
                        // no progress seen for two seconds: the kernel's stores do not reach this host memory while it runs.  The
                        // copied status word from here on, for the rest of the process.
                        FS_HIP(hipStreamSynchronize(s));
                        ws.mirror_ok = false;
                        mirrored = false;
                        fprintf(stderr, "[libfsamd] CG progress words in pinned memory not updated by the device: polling the status word by copies\n");
                    }
                    continue;
                }
            }
            const int kend = (k + bsz < max_iter + 1) ? k + bsz : max_iter + 1;
            const int k_before = k;
            if (fused || fusedp) {
                // launch k = update k + product k + 1; buffers [k & 1] are read, [(k + 1) & 1] written.  Decomposed space (fusedp):
                // the exchange kernel goes first - it reduces the sums of the previous launch's partials over the ranks, stores the
                // neighbours' w into the ghost rows of the current w and advances the ghost rows of r and s into the next buffers
                double* const Z[2] = {ws.z.p, ws.z2.p};
                double* const W[2] = {ws.w.p, ws.w2.p};
                double* const SV[2] = {ws.s.p, ws.s2.p};
                double* const PT[2] = {ws.partials.p, ws.partials.p + 3 * (int64_t)igrid};
                const size_t lds = (size_t)g_dict.ncls * g_dict.S * sizeof(double);
                fs_p2p_rowsred red2[2] = {};
                fs_p2p_sendrows snd2[2] = {};
                if (fusedp)
                    for (int c = 0; c < 2; ++c) FS_CHECK(fs_p2p_exchange_args(sp, PT[c], igrid, ws.sums.p, &red2[c], &snd2[c]));
                auto launch_exchange = [&](int par) {
                    const int64_t work = std::max(snd2[par].total_send, snd2[par].total_recv);
                    const fs_pp_ghosts pp = {W[par], SV[par], SV[par ^ 1], Z[par], Z[par ^ 1], ws.it_ctr.p, par};
                    hipLaunchKernelGGL((k_cg_p2p_exchange<true>), dim3(fs_grid_for(std::max<int64_t>(work, 1), FS_BLOCK, p2p_rows_cap)), dim3(FS_BLOCK), 0, s,
                                       0, 0, ws.ctrl.p, ws.scal.p, ws.status.p, (double*)nullptr, W[par], (double*)nullptr, red2[par], snd2[par], pp);
                };
                auto launch_iter = [&](int par) {
#define FS_ITER_ARGS dim3(igrid), dim3(FS_BLOCK), lds, s, sp->n_nodes_local, sp->n_dict_items, reinterpret_cast<const int4*>(sp->dict_items.p), \
                     reinterpret_cast<const dict_plan_round*>(sp->dict_plans.p), g_dict.cls.p, g_dict.values.p, g_dict.S, g_dict.ncls, \
                     Z[par], W[par], SV[par], Z[par ^ 1], W[par ^ 1], SV[par ^ 1], ws.p.p, x->d.p, ws.dvec.p, PT[par], PT[par ^ 1], igrid, \
                     ws.ctrl.p, ws.scal.p, ws.status.p, ws.it_ctr.p, par, hist_p, dict_map_xcd(), ws.sums.p, mirror_dev
                    if (BI && !fusedp) {
#define FS_BOX_ITER_ARGS dim3(igrid), dim3(6 * 64), BI->lds, s, BI->g, g_dict.cls.p, g_dict.values.p, g_dict.ncls, Z[par], W[par], SV[par], Z[par ^ 1], W[par ^ 1], \
                         SV[par ^ 1], ws.p.p, x->d.p, ws.dvec.p, PT[par], PT[par ^ 1], igrid, ws.ctrl.p, ws.scal.p, ws.status.p, ws.it_ctr.p, par, hist_p, mirror_dev
                        if (BI->ahead == 2) hipLaunchKernelGGL((k_box_cg_iter<4, 2, 2, 2>), FS_BOX_ITER_ARGS);
                        else hipLaunchKernelGGL((k_box_cg_iter<4, 2, 1, 2>), FS_BOX_ITER_ARGS);
#undef FS_BOX_ITER_ARGS
                    } else if (fusedp) {
                        launch_exchange(par);
                        hipLaunchKernelGGL((k_dict_cg_iter<3, true>), FS_ITER_ARGS);
                    } else hipLaunchKernelGGL((k_dict_cg_iter<3, false>), FS_ITER_ARGS);
#undef FS_ITER_ARGS
                };
                if (k == 0) {
                    // product 0 (w_0 = A r_0 and its sums) by the plain product kernel; s_{-1} = p_{-1} = 0 were set above
                    FS_CHECK(ws.it_ctr.zero(s));
                    if (fusedp && !p2p_ghosts_in) {          // the ghost rows of r_0: the plain send and receive kernels
                        FS_CHECK(fs_halo_exchange_dev(sp, ws.z.p, s));
                        p2p_ghosts_in = true;
                    }
                    if (BI && !fusedp) {
                        // (the product kernel on the ITERATION's patches and chunks: one dot partial per workgroup of either kernel)
                        box_plan_s P0;
                        P0.g = BI->g; P0.lds = BI->lds_product; P0.shape = 0;
                        launch_box<3>(&P0, g_dict.cls.p, g_dict.values.p, g_dict.ncls, ws.z.p, ws.w.p, ws.dvec.p, ws.partials.p, ws.status.p, 0, igrid, 1, s);
                        g_last_product_kind = 3;
                    } else
                        launch_spmv<3>(A, ws.z.p, ws.w.p, ws.dvec.p, ws.partials.p, ws.status.p, s, aval, nullptr, 0, 0, 0, 0);
                }
                if (graph_sized && graphs_allowed && k >= first_plain && kend - k == bsz && kend <= max_iter && (bsz & 1) == 0 && (k & 1) == 0) {
                    const void* key[24] = {A, aval, x->d.p, hist_p, ws.z.p, ws.w.p, ws.partials.p, ws.status.p, ws.dvec.p, ws.p.p, ws.s.p,
                                           ws.z2.p, ws.w2.p, ws.s2.p, ws.it_ctr.p, g_dict.cls.p, g_dict.values.p, sp->dict_items.p, sp->dict_plans.p,
                                           ws.ctrl.p, ws.scal.p, snd2[0].own_recv, red2[0].own_buf,
                                           fusedp ? reinterpret_cast<const void*>((uintptr_t)sp->halo.p2p.generation + 1) : nullptr};
                    const int64_t key_i[8] = {n, ((int64_t)g_dict.ncls * 256 + g_dict.S) * 4 + (fusedp ? 1 : 0) + (mirror_dev ? 2 : 0), bsz + 4096 * (BI ? BI->ahead : 0), igrid,
                                              (int64_t)dict_map_xcd() + 16 * (int64_t)p2p_rows_cap,
                                              (int64_t)sp->n_dict_items, (int64_t)A->serial, (int64_t)sp->serial};
                    if (!ws.cgf_graph || memcmp(key, ws.cgf_key, sizeof(key)) || memcmp(key_i, ws.cgf_key_i, sizeof(key_i))) {
                        if (ws.cgf_graph) { (void)hipGraphExecDestroy(ws.cgf_graph); ws.cgf_graph = nullptr; }
                        hipGraph_t graph = nullptr;
                        FS_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                        for (int i = 0; i < bsz; ++i) launch_iter(i & 1);
                        FS_HIP(hipStreamEndCapture(s, &graph));
                        FS_HIP(hipGraphInstantiate(&ws.cgf_graph, graph, nullptr, nullptr, 0));
                        (void)hipGraphDestroy(graph);
                        memcpy(ws.cgf_key, key, sizeof(key));
                        memcpy(ws.cgf_key_i, key_i, sizeof(key_i));
                    }
                    FS_HIP(hipGraphLaunch(ws.cgf_graph, s));
                    k = kend;
                }
                for (; k < kend; ++k) {
                    const bool sample = (k % sample_step == 1 % sample_step) && n_samples < krylov_ws::NSAMPLE;
                    if (sample) {
                        ws.sample_iter[n_samples] = k;
                        FS_HIP(hipEventRecord(ws.ev[n_samples][0], s));
                    }
                    launch_iter(k & 1);
                    if (sample) {
                        FS_HIP(hipEventRecord(ws.ev[n_samples][1], s));
                        FS_HIP(hipEventRecord(ws.ev[n_samples][2], s));
                        FS_HIP(hipEventRecord(ws.ev[n_samples][3], s));
                        ++n_samples;
                    }
                }
            }
            if (use_graph && graphs_allowed && k >= first_plain && kend - k == bsz && kend <= max_iter) {
                // everything the captured launches bake in: the vectors of the workspace, the operator's value and
                // structure arrays - and the serial numbers of matrix and space, because a destroyed operator's heap
                // and pool addresses are handed out again (another mesh with the same row count would otherwise replay
                // this graph over column arrays that no longer exist)
                fs_p2p_rowsred red = {};
                fs_p2p_sendrows snd = {};
                const int fgrid = p2p_fuse ? spmv_partials_unsplit(sp, bs) : sgrid;
                if (p2p_fuse) FS_CHECK(fs_p2p_exchange_args(sp, ws.partials.p, fgrid, ws.sums.p, &red, &snd));
                const bool dict_on = g_dict.built_for && g_dict.built_for == aval;
                // (the tile product of a lattice-ordered operator bakes in the list tables, the tile table and the geometry of lat_prepare:
                // those buffers grow - are allocated anew - when another space brings more classes or tiles)
                const bool lat_on = dict_on && g_lat.ok && g_lat.built_for == aval && g_lat.space_serial == sp->serial;
                const bool lm_on = lat_on && g_lm.ok && g_lat_march;          // (k_lat_march bakes in its own tables and geometry)
                const void* key[32] = {A, aval, x->d.p, hist_p, ws.z.p, ws.w.p, ws.partials.p, ws.status.p,
                                       ws.dvec.p, ws.p.p, ws.s.p, sp->sell_col.p, snd.own_recv, snd.peers, red.own_buf, red.peer_buf,
                                       dict_on ? (const void*)g_dict.cls.p : nullptr, dict_on ? (const void*)g_dict.values.p : nullptr,
                                       dict_on ? (const void*)sp->dict_items.p : nullptr, p2p_fuse ? (const void*)ws.sg.p : nullptr,
                                       dict_on ? (const void*)sp->dict_plans.p : nullptr, dict_on ? (const void*)sp->halo.items_interior.p : nullptr,
                                       dict_on ? (const void*)sp->halo.items_boundary.p : nullptr,
                                       p2p_fuse ? reinterpret_cast<const void*>((uintptr_t)sp->halo.p2p.generation) : nullptr,
                                       lat_on ? (lm_on ? (const void*)g_lm.sl_line.p : (const void*)g_lat.tile_cls.p) : nullptr, lat_on ? (const void*)g_lat.cnt.p : nullptr,
                                       lat_on ? (const void*)g_lat.coef.p : nullptr, lat_on ? (lm_on ? (const void*)g_lm.SL.p : (const void*)g_lat.rel.p) : nullptr,
                                       lat_on ? (const void*)g_lat.off.p : nullptr, lat_on ? (const void*)g_lat.relc.p : nullptr,
                                       lat_on ? reinterpret_cast<const void*>((uintptr_t)g_lat.geom.n_tiles) : nullptr,
                                       lat_on ? reinterpret_cast<const void*>(((uintptr_t)g_lat.geom.grid << 32) | (uintptr_t)(uint32_t)g_lat.geom.w_tiles) : nullptr};
                // (+ whether the product is the row-dictionary kernel, with the class count and width its launch bakes in)
                const int64_t dict_sig = dict_on ? ((int64_t)g_dict.ncls * 256 + g_dict.S) * 256 + g_dict.C : 0;
                const int64_t key_i[8] = {n, (p2p_fuse ? (int64_t)p2p_rows_cap + 1 : 0) + 1024 * dict_sig, bsz, fgrid, vgrid,
                                          (int64_t)upd_nt * 2 + (int64_t)spmv_nontemporal(sp, bs) + (mirror_dev ? 4 : 0) + (lm_on ? 8 + 16 * (int64_t)g_lm.g.PY + 1024 * (int64_t)g_lm.g.ZC : 0),
                                          (int64_t)A->serial, (int64_t)sp->serial};
                if (!ws.cg_graph || memcmp(key, ws.cg_key, sizeof(key)) || memcmp(key_i, ws.cg_key_i, sizeof(key_i))) {
                    if (ws.cg_graph) { (void)hipGraphExecDestroy(ws.cg_graph); ws.cg_graph = nullptr; }
                    hipGraph_t graph = nullptr;
                    FS_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                    int rc_cap = FS_OK;
                    for (int i = 0; i < bsz && rc_cap == FS_OK; ++i) {
                        // iteration index (status[2]) and iteration limit (ctrl[2]) from the device: iter = -1
                        if (p2p_fuse) {
                            launch_spmv<3>(A, ws.z.p, ws.w.p, ws.dvec.p, ws.partials.p, ws.status.p, s, aval);
                            const int64_t work = std::max(snd.total_send, snd.total_recv);
                            hipLaunchKernelGGL((k_cg_p2p_exchange<false>), dim3(fs_grid_for(std::max<int64_t>(work, 1), FS_BLOCK, p2p_rows_cap)), dim3(FS_BLOCK), 0, s,
                                               -1, 0, ws.ctrl.p, ws.scal.p, ws.status.p, ws.z.p, ws.w.p, ws.sg.p, red, snd, fs_pp_ghosts{});
                            if (upd_nt) hipLaunchKernelGGL((k_cg_update_scaled<false, true>), dim3(vgrid), dim3(FS_BLOCK), 0, s, n, -1, 0, ws.partials.p, fgrid, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.z.p, ws.w.p, ws.p.p, ws.s.p, x->d.p);
                            else hipLaunchKernelGGL((k_cg_update_scaled<false, false>), dim3(vgrid), dim3(FS_BLOCK), 0, s, n, -1, 0, ws.partials.p, fgrid, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.z.p, ws.w.p, ws.p.p, ws.s.p, x->d.p);
                            continue;
                        }
                        rc_cap = spmv_overlapped<3>(A, ws.z.p, ws.w.p, ws.dvec.p, ws.partials.p, ws.status.p, s, aval);
                        if (upd_nt) hipLaunchKernelGGL((k_cg_update_scaled<true, true>), dim3(vgrid), dim3(FS_BLOCK), 0, s, n, -1, 0, ws.partials.p, sgrid, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.z.p, ws.w.p, ws.p.p, ws.s.p, x->d.p, mirror_dev, g_upd_r_plain);
                        else hipLaunchKernelGGL((k_cg_update_scaled<true, false>), dim3(vgrid), dim3(FS_BLOCK), 0, s, n, -1, 0, ws.partials.p, sgrid, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.z.p, ws.w.p, ws.p.p, ws.s.p, x->d.p, mirror_dev, g_upd_r_plain);
                    }
                    const hipError_t e_end = hipStreamEndCapture(s, &graph);
                    FS_CHECK(rc_cap);
                    FS_HIP(e_end);
                    FS_HIP(hipGraphInstantiate(&ws.cg_graph, graph, nullptr, nullptr, 0));
                    (void)hipGraphDestroy(graph);
                    memcpy(ws.cg_key, key, sizeof(key));
                    memcpy(ws.cg_key_i, key_i, sizeof(key_i));
                }
                FS_HIP(hipGraphLaunch(ws.cg_graph, s));
                k = kend;
            }
            for (; k < kend; ++k) {
                const bool sample = (k % sample_step == 1 % sample_step) && n_samples < krylov_ws::NSAMPLE;
                if (sample) ws.sample_iter[n_samples] = k;
                if (bicg) {
                    const int co = k == max_iter ? 1 : 0;
                    // K1: p, y
                    if (fuse_sums) {
                        hipLaunchKernelGGL(k_bicg_p<true>, dim3(vgrid), dim3(FS_BLOCK), 0, s, n, k, co, ws.partials2.p, vgrid, ws.bsums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.dinv.p, ws.r.p, ws.p.p, ws.w.p, ws.y.p);
                    } else {
                        FS_CHECK(fs_comm_sum_allreduce_dev(ws.partials2.p, vgrid, 2, ws.bsums.p, s));
                        hipLaunchKernelGGL(k_bicg_p<false>, dim3(vgrid), dim3(FS_BLOCK), 0, s, n, k, co, ws.partials2.p, vgrid, ws.bsums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.dinv.p, ws.r.p, ws.p.p, ws.w.p, ws.y.p);
                    }
                    // K2: v = A y, rhat.v
                    if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][0], s));
                    FS_CHECK(spmv_overlapped<2>(A, ws.y.p, ws.w.p, ws.rhat.p, ws.partials.p, ws.status.p, s));
                    if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][1], s));
                    // K3: s, z
                    if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][2], s));
                    if (fuse_sums) {
                        hipLaunchKernelGGL(k_bicg_s<true>, dim3(vgrid), dim3(FS_BLOCK), 0, s, n, k, ws.partials.p, sgrid, ws.bsums.p + 4, ws.scal.p, ws.status.p, ws.dinv.p, ws.r.p, ws.w.p, ws.s.p, ws.z.p);
                    } else {
                        FS_CHECK(fs_comm_sum_allreduce_dev(ws.partials.p, sgrid, 1, ws.bsums.p + 4, s));
                        hipLaunchKernelGGL(k_bicg_s<false>, dim3(vgrid), dim3(FS_BLOCK), 0, s, n, k, ws.partials.p, sgrid, ws.bsums.p + 4, ws.scal.p, ws.status.p, ws.dinv.p, ws.r.p, ws.w.p, ws.s.p, ws.z.p);
                    }
                    if (sample) {
                        FS_HIP(hipEventRecord(ws.ev[n_samples][3], s));
                        ++n_samples;
                    }
                    // K4: t = A z, (t.s, t.t)
                    FS_CHECK(spmv_overlapped<2>(A, ws.z.p, ws.t.p, ws.s.p, ws.partials.p, ws.status.p, s));
                    // K5: x, r, next (rhat.r, r.r)
                    if (fuse_sums) {
                        hipLaunchKernelGGL(k_bicg_x<true>, dim3(vgrid), dim3(FS_BLOCK), 0, s, n, k, ws.partials.p, sgrid, ws.bsums.p + 8, ws.scal.p, ws.status.p, x->d.p, ws.y.p, ws.z.p, ws.r.p, ws.s.p, ws.t.p, ws.rhat.p, ws.partials2.p);
                    } else {
                        FS_CHECK(fs_comm_sum_allreduce_dev(ws.partials.p, sgrid, 2, ws.bsums.p + 8, s));
                        hipLaunchKernelGGL(k_bicg_x<false>, dim3(vgrid), dim3(FS_BLOCK), 0, s, n, k, ws.partials.p, sgrid, ws.bsums.p + 8, ws.scal.p, ws.status.p, x->d.p, ws.y.p, ws.z.p, ws.r.p, ws.s.p, ws.t.p, ws.rhat.p, ws.partials2.p);
                    }
                    continue;
                }
                if (pipelined) {
                    const int co = k == max_iter ? 1 : 0;
                    // n = A w (its halo was begun behind the previous update), under which the sums of (r, w) are reduced
                    if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][0], s));
                    FS_CHECK(spmv_overlapped<4>(A, ws.pw.p, ws.w.p, nullptr, nullptr, ws.status.p, s, aval));
                    if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][1], s));
                    if (red_stream) FS_HIP(hipStreamWaitEvent(s, ws.ev_red, 0));
                    if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][2], s));
                    int64_t m0 = 0, m1 = n;
                    if (early_a > 0 || early_b < n) {
                        hipLaunchKernelGGL(k_pcg_update_rows, dim3(fs_grid_for(early_a + (n - early_b), FS_BLOCK, 256)), dim3(FS_BLOCK), 0, s,
                                           early_a, early_b, n, k, co, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, ws.z.p, ws.pw.p, ws.w.p, ws.p.p, ws.s.p, ws.pz.p, x->d.p);
                        FS_CHECK(fs_halo_begin_dev(sp, ws.pw.p, s));
                        sp->halo.begun = true;
                        m0 = early_a; m1 = early_b;
                    }
#define FS_PCG_ARGS dim3(vgrid), dim3(FS_BLOCK), 0, s, n, m0, m1, k, co, ws.partials.p, vgrid, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.dvec.p, ws.z.p, ws.pw.p, ws.w.p, ws.p.p, ws.s.p, ws.pz.p, x->d.p
                    if (fuse_sums) {
                        if (upd_nt) hipLaunchKernelGGL((k_pcg_update<true, true>), FS_PCG_ARGS);
                        else hipLaunchKernelGGL((k_pcg_update<true, false>), FS_PCG_ARGS);
                    } else {
                        if (upd_nt) hipLaunchKernelGGL((k_pcg_update<false, true>), FS_PCG_ARGS);
                        else hipLaunchKernelGGL((k_pcg_update<false, false>), FS_PCG_ARGS);
                    }
#undef FS_PCG_ARGS
                    if (sample) {
                        FS_HIP(hipEventRecord(ws.ev[n_samples][3], s));
                        ++n_samples;
                    }
                    if (!fuse_sums) {
                        // the halo of the new w goes first on the communication stream (the boundary rows of the next
                        // product wait for it), the reduction of the new sums behind it
                        // (whether a rank's product is split is a LOCAL property - a thin part may have no interior slice -
                        // while the order of halo and reduction on the communicator must be the same everywhere: the
                        // exchange is begun here on every rank with a plan, split or not)
                        if (!sp->halo.begun && sp->halo.active) {
                            FS_CHECK(fs_halo_begin_dev(sp, ws.pw.p, s));
                            sp->halo.begun = true;
                        }
                        FS_CHECK(pcg_reduce((k + 1) & 1));
                    }
                    continue;
                }
                if (p2p_fuse) {
                    const int co = k == max_iter ? 1 : 0;
                    fs_p2p_rowsred red = {};
                    fs_p2p_sendrows snd = {};
                    if (!p2p_ghosts_in) {          // first iteration of a pass: the plain send and receive kernels
                        FS_CHECK(fs_halo_exchange_dev(sp, ws.z.p, s));
                        p2p_ghosts_in = true;
                    }
                    if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][0], s));
                    launch_spmv<3>(A, ws.z.p, ws.w.p, ws.dvec.p, ws.partials.p, ws.status.p, s, aval);
                    if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][1], s));
                    if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][2], s));
                    const int fgrid = spmv_partials_unsplit(sp, bs);
                    FS_CHECK(fs_p2p_exchange_args(sp, ws.partials.p, fgrid, ws.sums.p, &red, &snd));
                    const int64_t work = std::max(snd.total_send, snd.total_recv);
                    hipLaunchKernelGGL((k_cg_p2p_exchange<false>), dim3(fs_grid_for(std::max<int64_t>(work, 1), FS_BLOCK, p2p_rows_cap)), dim3(FS_BLOCK), 0, s,
                                       k, co, ws.ctrl.p, ws.scal.p, ws.status.p, ws.z.p, ws.w.p, ws.sg.p, red, snd, fs_pp_ghosts{});
                    if (upd_nt) hipLaunchKernelGGL((k_cg_update_scaled<false, true>), dim3(vgrid), dim3(FS_BLOCK), 0, s, n, k, co, ws.partials.p, fgrid, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.z.p, ws.w.p, ws.p.p, ws.s.p, x->d.p);
                    else hipLaunchKernelGGL((k_cg_update_scaled<false, false>), dim3(vgrid), dim3(FS_BLOCK), 0, s, n, k, co, ws.partials.p, fgrid, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.z.p, ws.w.p, ws.p.p, ws.s.p, x->d.p);
                    if (sample) {
                        FS_HIP(hipEventRecord(ws.ev[n_samples][3], s));
                        ++n_samples;
                    }
                    continue;
                }
                if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][0], s));
                if (ds) FS_CHECK(spmv_overlapped<3>(A, ws.z.p, ws.w.p, ws.dvec.p, ws.partials.p, ws.status.p, s, aval));
                else FS_CHECK(spmv_overlapped<1>(A, ws.z.p, ws.w.p, ws.r.p, ws.partials.p, ws.status.p, s));
                if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][1], s));
                if (ds) {
                    const int co = k == max_iter ? 1 : 0;
                    if (fuse_sums) {
                        if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][2], s));
                        if (upd_nt) hipLaunchKernelGGL((k_cg_update_scaled<true, true>), dim3(vgrid), dim3(FS_BLOCK), 0, s, n, k, co, ws.partials.p, sgrid, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.z.p, ws.w.p, ws.p.p, ws.s.p, x->d.p, mirror_dev, g_upd_r_plain);
                        else hipLaunchKernelGGL((k_cg_update_scaled<true, false>), dim3(vgrid), dim3(FS_BLOCK), 0, s, n, k, co, ws.partials.p, sgrid, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.z.p, ws.w.p, ws.p.p, ws.s.p, x->d.p, mirror_dev, g_upd_r_plain);
                    } else {
                        FS_CHECK(fs_comm_sum_allreduce_dev(ws.partials.p, sgrid, 3, ws.sums.p, s));
                        if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][2], s));
                        // Slabs send a prefix and / or a suffix of their rows: those are updated first and their exchange is
                        // started, so that it runs under the rest of the update AND the interior product of the next iteration
                        // (the exchange alone used to start only with that product).
                        int64_t m0 = 0, m1 = n;
                        if (early_a > 0 || early_b < n) {
                            hipLaunchKernelGGL(k_cg_update_scaled_rows, dim3(fs_grid_for(early_a + (n - early_b), FS_BLOCK, 256)), dim3(FS_BLOCK), 0, s,
                                               early_a, early_b, n, k, co, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, ws.z.p, ws.w.p, ws.p.p, ws.s.p, x->d.p);
                            FS_CHECK(fs_halo_begin_dev(sp, ws.z.p, s));
                            sp->halo.begun = true;
                            m0 = early_a; m1 = early_b;
                        }
                        const int64_t nm = m1 - m0;
                        if (upd_nt) hipLaunchKernelGGL((k_cg_update_scaled<false, true>), dim3(vgrid), dim3(FS_BLOCK), 0, s, nm, k, co, ws.partials.p, sgrid, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.z.p + m0, ws.w.p + m0, ws.p.p + m0, ws.s.p + m0, x->d.p + m0);
                        else hipLaunchKernelGGL((k_cg_update_scaled<false, false>), dim3(vgrid), dim3(FS_BLOCK), 0, s, nm, k, co, ws.partials.p, sgrid, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.z.p + m0, ws.w.p + m0, ws.p.p + m0, ws.s.p + m0, x->d.p + m0);
                    }
                } else if (fuse_sums) {
                    if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][2], s));
                    hipLaunchKernelGGL(k_cg_update<true>, dim3(vgrid), dim3(FS_BLOCK), 0, s, n, k, k == max_iter ? 1 : 0, ws.partials.p, sgrid, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.dinv.p, ws.z.p, ws.w.p, ws.p.p, ws.s.p, x->d.p, ws.r.p);
                } else {
                    FS_CHECK(fs_comm_sum_allreduce_dev(ws.partials.p, sgrid, 3, ws.sums.p, s));
                    if (sample) FS_HIP(hipEventRecord(ws.ev[n_samples][2], s));
                    hipLaunchKernelGGL(k_cg_update<false>, dim3(vgrid), dim3(FS_BLOCK), 0, s, n, k, k == max_iter ? 1 : 0, ws.partials.p, sgrid, ws.sums.p, ws.ctrl.p, ws.scal.p, ws.status.p, hist_p, ws.dinv.p, ws.z.p, ws.w.p, ws.p.p, ws.s.p, x->d.p, ws.r.p);
                }
                if (sample) {
                    FS_HIP(hipEventRecord(ws.ev[n_samples][3], s));
                    ++n_samples;
                }
            }
            FS_KERNEL_CHECK();
            n_launches += k - k_before;
            if (!mirrored) {
                FS_HIP(hipMemcpyAsync(ws.h_status + 4 * slot, ws.status.p, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
                FS_HIP(hipEventRecord(ws.poll[slot], s));
                if (pending >= 0) {
                    FS_HIP(hipEventSynchronize(ws.poll[pending]));
                    if (ws.h_status[4 * pending] != 0) finished = true;
                }
                pending = slot;
                slot ^= 1;
            }
            if (k > max_iter) finished = true;
        }
        lap("iterations of the pass");
        if (sp->halo.begun) {          // the exchange started for a product that is not coming any more
            FS_CHECK(fs_halo_end_dev(sp, s));
            sp->halo.begun = false;
        }
        // pipelined: the reduction enqueued behind the last update still reads the partial sums on the communication stream;
        // nothing of the workspace is touched again before it is through
        if (red_stream) FS_HIP(hipStreamWaitEvent(s, ws.ev_red, 0));
        // The end of a pass used to be six host synchronisations (stream, status word, sums, control block, the un-scaling of x,
        // the history): 0.68 ms of fixed cost per solve at 1 M rows, a tenth of the whole solve.  Nothing of the true-residual
        // computation depends on what the host learns from the status word, so it is enqueued first and status, sums and control
        // block come back in ONE synchronisation.
        // true residual b - A x (scaled mode: sum d (bhat - Ahat xhat)^2, the same number)
        FS_HIP(hipMemcpyAsync(ws.z.p, x->d.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
        FS_CHECK(fs_halo_exchange_dev(sp, ws.z.p, s));
        launch_spmv<0>(A, ws.z.p, ws.w.p, nullptr, nullptr, nullptr, s, aval);
        if (ds) hipLaunchKernelGGL(k_residual_scaled, dim3(pgrid), dim3(FS_BLOCK), 0, s, ws.bhat.p, ws.w.p, ws.dvec.p, n, (double*)nullptr, ws.partials.p);
        else hipLaunchKernelGGL(k_residual, dim3(pgrid), dim3(FS_BLOCK), 0, s, b->d.p, ws.w.p, n, (double*)nullptr, ws.partials.p);
        FS_CHECK(fs_comm_sum_allreduce_dev(ws.partials.p, pgrid, 1, ws.sums.p + 5, s));
        FS_HIP(hipMemcpyAsync(ws.h_status + 12, ws.status.p, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
        FS_HIP(hipMemcpyAsync(ws.h_vals, ws.sums.p, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
        FS_HIP(hipMemcpyAsync(ws.h_vals + 8, ws.ctrl.p, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
        FS_HIP(hipStreamSynchronize(s));
        if (fs_p2p_reduce_enabled()) FS_CHECK(fs_p2p_check(s));
        for (int q = 0; q < 4; ++q) h_status[q] = ws.h_status[12 + q];
        const int iters = h_status[1];
        total_iters += iters;
        // launches enqueued after the recurrence stopped return on the status word: their samples time no-ops
        for (int i = first_sample; i < n_samples; ++i) ws.sample_live[i] = ws.sample_iter[i] < iters;
        if (ws.h_status[8] != 0) {
            fs_set_error("fs_krylov_solve: %d zero%s diagonal entries (Jacobi preconditioner undefined)", ws.h_status[8], ds ? " or negative" : "");
            return FS_ERR_NUMERIC;
        }
        const double* h_pass = ws.h_vals;
        const double* h_ctrl2 = ws.h_vals + 8;
        true_rr = h_pass[5];
        thresh = h_ctrl2[0];
        bb_host = h_ctrl2[1];
        ++n_pass;
        if (getenv("FS_KRYLOV_DEBUG"))
            fprintf(stderr, "[fs_krylov] pass %d: status %d, %d iterations (total %d), true ||r||^2 %.3e, threshold %.3e\n",
                    n_pass, h_status[0], h_status[1], total_iters, true_rr, thresh);
        const bool recurrence_converged = h_status[0] == 1;
        if (!recurrence_converged || true_rr <= thresh * 1.0201 || n_pass >= 8 || total_iters >= opts->max_iter) break;
        if (true_rr > 0.7 * prev_true_rr) break;   // no longer improving: attainable accuracy of fp64 reached
        prev_true_rr = true_rr;
        use_guess = true;   // restart: r := b - A x exactly, then continue
    }
    if (ds) {   // x = D^-1/2 xhat (no synchronisation of its own: the download of the history below waits for it)
        hipLaunchKernelGGL(k_pointwise_mul, dim3(fs_grid_for(n)), dim3(FS_BLOCK), 0, s, ws.dinv.p, x->d.p, n, x->d.p);
        FS_KERNEL_CHECK();
    }
    const int iters = total_iters;
    if (fs_rt().comm && getenv("FS_COMM_TIMING")) {
        double au, hu; long ac, hc;
        fs_comm_host_time(&au, &ac, &hu, &hc, true);
        fprintf(stderr, "[fs_krylov] host time inside RCCL enqueue calls: all-reduce %.1f us x %ld, grouped send/recv %.1f us x %ld; %d iterations in %.1f ms\n",
                ac ? au / ac : 0.0, ac, hc ? hu / hc : 0.0, hc, iters,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    }
#ifdef FS_ITER_TIMING
    if (fused && getenv("FS_ITER_TIMING")) {
        std::vector<long long> h(8 * 4096);
        FS_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_iter_dbg), h.size() * sizeof(long long)));
        long long first = h[0], last = 0;
        for (int b = 0; b < igrid; ++b) { first = std::min(first, h[8 * b]); last = std::max(last, h[8 * b + 7]); }
        double mean[8] = {0}, mx[8] = {0};
        for (int b = 0; b < igrid; ++b)
            for (int k2 = 0; k2 < 8; ++k2) { const double v = (h[8 * b + k2] - first) * 0.01; mean[k2] += v / igrid; mx[k2] = std::max(mx[k2], v); }
        fprintf(stderr, "[iter timing, last launch, us from the first workgroup's start] span %.2f;", (last - first) * 0.01);
        for (int k2 = 0; k2 < 8; ++k2) fprintf(stderr, " s%d mean %.2f max %.2f;", k2, mean[k2], mx[k2]);
        fprintf(stderr, "\n");
    }
#endif
    ws.last_hist.resize((size_t)iters + 1);
    FS_CHECK(ws.hist.download(ws.last_hist.data(), iters + 1, s));
    const auto t_end = std::chrono::steady_clock::now();

    if (stats) {
        memset(stats, 0, sizeof(*stats));
        const double bb = bb_host;
        stats->iterations = iters;
        // PETSc semantics: converged = the recurrence residual met the tolerance; the recomputed true
        // residual is reported next to it (restarts above keep the two together whenever fp64 allows)
        stats->converged = h_status[0] == 2 ? -1 : (h_status[0] == 1 ? 1 : 0);
        stats->bnorm = sqrt(bb);
        stats->rel_residual = bb > 0.0 ? sqrt(ws.last_hist[iters] / bb) : 0.0;
        stats->true_rel_residual = bb > 0.0 ? sqrt(true_rr / bb) : sqrt(true_rr);
        stats->solve_ms = std::chrono::duration<double, std::milli>(t_end - t_begin).count();
        double t_spmv = 0.0, t_upd = 0.0;
        int cnt = 0;
        for (int i = 0; i < n_samples; ++i) {
            if (!ws.sample_live[i]) continue;
            float a = 0.f, c = 0.f;
            if (hipEventElapsedTime(&a, ws.ev[i][0], ws.ev[i][1]) != hipSuccess) break;
            if (hipEventElapsedTime(&c, ws.ev[i][2], ws.ev[i][3]) != hipSuccess) break;
            t_spmv += a;
            t_upd += c;
            ++cnt;
        }
        if (cnt) {
            stats->spmv_ms = t_spmv / cnt;
            stats->update_ms = t_upd / cnt;
        }
        stats->spmv_bytes = sp->nnz_nodes * bs * bs * 12 + n * 20;
        stats->row_classes = g_dict.built_for ? g_dict.ncls : 0;
        stats->fused_iteration = fusedp_used ? 2 : (fused ? 1 : 0);
        stats->classes_kept = g_dict.built_for && g_dict.kept ? 1 : 0;
        stats->launches = n_launches;
        stats->product_kind = fused ? (BI ? 3 : 1) : g_last_product_kind;
        if (fused) stats->update_ms = 0.0;       // (spmv_ms is the whole iteration: one launch)
    }
    if (h_status[0] == 2) {
        fs_set_error("fs_krylov_solve: %s breakdown at iteration %d (operator not SPD / rho = 0 / NaN)", bicg ? "BiCGStab" : "CG", iters);
        return FS_ERR_NUMERIC;
    }
    return FS_OK;
}

extern "C" int fs_krylov_history(double* out, int capacity, int* count) {
    const int n = (int)g_ws.last_hist.size();
    if (count) *count = n;
    if (out)
        for (int i = 0; i < n && i < capacity; ++i) out[i] = g_ws.last_hist[i];
    return FS_OK;
}

void fs_krylov_preload() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(k_set_threshold));
    (void)hipGetLastError();
}
