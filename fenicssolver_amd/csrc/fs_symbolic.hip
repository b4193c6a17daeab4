// Function space + sparsity (symbolic phase) of libfsamd.so.
//
// Replaces what DOLFIN does implicitly inside the first assemble()/
// LinearVariationalSolver (SolverBase.py:595, 608-612, 644): dofmap + sparsity
// pattern.  Everything runs on the device:
//   cell->(row,col) keys  -> radix sort -> unique  = sorted-column CSR
//   CSR -> SELL-64 (slice = one wavefront, column-major inside a slice)
//   per cell the 16 SELL entry indices of its (a,b) pairs ("slot table"), so the
//   numeric assembly is a pure scatter with no searching.
#include "fs_common.h"
#include <hipcub/hipcub.hpp>
#include <stdlib.h>
#include <algorithm>

// ---- keys ------------------------------------------------------------------------------
// nd*(nd-1) directed pairs per cell (a != b) + one diagonal key per owned row.
// bits needed to write m itself (not m - 1): fields of sort keys whose all-ones value has to stay above every real entry
static int fs_bits_for(uint64_t m) {
    int b = 1;
    while (b < 32 && (m >> b) != 0) ++b;
    return b;
}

__global__ void k_pair_keys(const int32_t* __restrict__ cell_dofs, int nd, int64_t nc, int64_t n_rows,
                            uint64_t* __restrict__ keys) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < nc; c += stride) {
        int32_t v[10];
        for (int a = 0; a < nd; ++a) v[a] = cell_dofs[c * nd + a];
        int k = 0;
        for (int a = 0; a < nd; ++a) {
            for (int b = 0; b < nd; ++b) {
                if (a == b) continue;
                uint64_t key = ~0ULL;  // rows owned elsewhere sort to the end and are dropped
                if (v[a] < n_rows) key = ((uint64_t)(uint32_t)v[a] << 32) | (uint32_t)v[b];
                keys[(int64_t)k * nc + c] = key;
                ++k;
            }
        }
    }
}

// ---- P2: edge nodes ---------------------------------------------------------------------------
__device__ __constant__ int FS_EDGE_V[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};  // UFC

__device__ __forceinline__ uint64_t fs_edge_key(int32_t a, int32_t b, int grouped) {
    const uint32_t lo = (uint32_t)(a < b ? a : b), hi = (uint32_t)(a < b ? b : a);
    return grouped ? (((uint64_t)(hi - lo) << 32) | lo) : (((uint64_t)lo << 32) | hi);
}

// UFC edges of a triangle: edge i lies opposite vertex i
__device__ __constant__ int FS_TRI_EDGE_V[3][2] = {{1, 2}, {0, 2}, {0, 1}};

__global__ void k_edge_keys(const int32_t* __restrict__ cells, int64_t nc, int grouped, uint64_t* __restrict__ keys, int tdim = 3) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < nc; c += stride) {
        const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
        const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
        if (tdim == 2) {
            for (int e = 0; e < 3; ++e)
                keys[(int64_t)e * nc + c] = fs_edge_key(v[FS_TRI_EDGE_V[e][0]], v[FS_TRI_EDGE_V[e][1]], grouped);
        } else {
            for (int e = 0; e < 6; ++e)
                keys[(int64_t)e * nc + c] = fs_edge_key(v[FS_EDGE_V[e][0]], v[FS_EDGE_V[e][1]], grouped);
        }
    }
}

// v1 - v0 of every lexicographic unique edge, to count the distinct differences
__global__ void k_edge_deltas(const uint64_t* __restrict__ ukeys, int64_t ne, uint32_t* __restrict__ delta) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < ne; i += stride) delta[i] = (uint32_t)(ukeys[i] & 0xffffffffULL) - (uint32_t)(ukeys[i] >> 32);
}

__global__ void k_edge_table(const uint64_t* __restrict__ ukeys, int64_t ne, int grouped, int32_t* __restrict__ edges) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < ne; i += stride) {
        const uint32_t hi32 = (uint32_t)(ukeys[i] >> 32), lo32 = (uint32_t)(ukeys[i] & 0xffffffffULL);
        edges[2 * i] = (int32_t)(grouped ? lo32 : hi32);
        edges[2 * i + 1] = (int32_t)(grouped ? lo32 + hi32 : lo32);
    }
}

// ghost flag of every unique edge: owned iff its endpoint of smaller global id is an owned vertex
__global__ void k_edge_ghost_flag(const uint64_t* __restrict__ ukeys, int64_t ne, int grouped, const int64_t* __restrict__ gid,
                                  int64_t n_owned, int32_t* __restrict__ flag, int32_t* __restrict__ index) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < ne; i += stride) {
        const uint32_t hi32 = (uint32_t)(ukeys[i] >> 32), lo32 = (uint32_t)(ukeys[i] & 0xffffffffULL);
        const int32_t v0 = (int32_t)(grouped ? lo32 : hi32), v1 = (int32_t)(grouped ? lo32 + hi32 : lo32);
        const int32_t vmin = gid[v0] < gid[v1] ? v0 : v1;
        flag[i] = vmin < n_owned ? 0 : 1;
        index[i] = (int32_t)i;
    }
}
// order[j] = sorted-key position of the j-th edge in node order (owned first) -> edge_node and the edge table
__global__ void k_edge_nodes(const int32_t* __restrict__ order, int64_t ne, int64_t neo, int64_t nvo, int64_t nv,
                             const uint64_t* __restrict__ ukeys, int grouped, int32_t* __restrict__ edge_node,
                             int32_t* __restrict__ edges) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; j < ne; j += stride) {
        const int32_t i = order[j];
        edge_node[i] = (int32_t)(j < neo ? nvo + j : nv + j);     // ghost edges sit after the ghost vertices
        const uint32_t hi32 = (uint32_t)(ukeys[i] >> 32), lo32 = (uint32_t)(ukeys[i] & 0xffffffffULL);
        edges[2 * j] = (int32_t)(grouped ? lo32 : hi32);
        edges[2 * j + 1] = (int32_t)(grouped ? lo32 + hi32 : lo32);
    }
}

// cell_dofs[c] = {nodes of the 4 vertices, node of each of the 6 edges (edge_node of its sorted-key position)}
__global__ void k_p2_cell_dofs(const int32_t* __restrict__ cells, int64_t nc, int64_t nvo, int64_t neo,
                               const uint64_t* __restrict__ ukeys, int64_t ne, int grouped,
                               const int32_t* __restrict__ edge_node, int32_t* __restrict__ cell_dofs, int tdim = 3) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < nc; c += stride) {
        const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
        const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
        if (tdim == 2) {          // 3 vertices, then the 3 UFC edges
            for (int a = 0; a < 3; ++a) cell_dofs[c * 6 + a] = v[a] < nvo ? v[a] : (int32_t)(v[a] + neo);
            for (int e = 0; e < 3; ++e) {
                const uint64_t key = fs_edge_key(v[FS_TRI_EDGE_V[e][0]], v[FS_TRI_EDGE_V[e][1]], grouped);
                int64_t lo = 0, hi = ne;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (ukeys[mid] < key) lo = mid + 1; else hi = mid;
                }
                cell_dofs[c * 6 + 3 + e] = edge_node[lo];
            }
            continue;
        }
        for (int a = 0; a < 4; ++a) cell_dofs[c * 10 + a] = v[a] < nvo ? v[a] : (int32_t)(v[a] + neo);
        for (int e = 0; e < 6; ++e) {
            const uint64_t key = fs_edge_key(v[FS_EDGE_V[e][0]], v[FS_EDGE_V[e][1]], grouped);
            int64_t lo = 0, hi = ne;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (ukeys[mid] < key) lo = mid + 1; else hi = mid;
            }
            cell_dofs[c * 10 + 4 + e] = edge_node[lo];
        }
    }
}

__global__ void k_diag_keys(int64_t n_rows, uint64_t* __restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_rows; i += stride) keys[i] = ((uint64_t)i << 32) | (uint64_t)i;
}

// ---- the sparsity pattern row by row (CG1, CG2) --------------------------------------------------------------------------
// The columns of row r are the nodes of the cells around r.  With the (node, cell) incidences sorted by node - which the
// gather assembly needs anyway - a lane collects them for its row in a small ascending set in LDS (set[slot][thread]: no bank
// conflicts; binary search, a new entry shifts the tail) and writes it slot-major; an exclusive sum of the counts gives the row pointers and a second kernel packs the
// columns.  This replaces sorting 12 keys per cell (71 M at 1 M rows: 6 radix passes of 1.1 GB) and the unique pass over them.
// A row with more neighbours than the set holds sends the whole space back to the sorted-keys path.
// CAP / BLOCK: 32 entries x 256 rows per workgroup for CG1, 160 x 64 for CG2 (a vertex row of a tetrahedral CG2 space couples to
// about 65 nodes, an edge row to about 27): 32 / 40 KB of LDS.
template <int FS_ROWCOL_CAP, int FS_BLOCK_T>
__global__ void __launch_bounds__(FS_BLOCK_T) k_row_columns(const uint64_t* __restrict__ keys, const int32_t* __restrict__ inc_ptr,
                                                            const int32_t* __restrict__ cell_dofs, int nd, int64_t n_rows,
                                                            int32_t* __restrict__ cnt, int32_t* __restrict__ cols, int* __restrict__ overflow) {
    constexpr int RB = FS_BLOCK_T;       // rows per workgroup = the set's stride
    __shared__ int32_t set[FS_ROWCOL_CAP * RB];
    const int t = threadIdx.x;
    int64_t r = (int64_t)blockIdx.x * blockDim.x + t;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_rows; r += stride) {
        int n = 1;
        set[t] = (int32_t)r;                  // the diagonal, cells or not
        bool over = false;
        const int32_t first = inc_ptr[r], last = inc_ptr[r + 1];
        // the set is kept ascending (as the sorted keys delivered the columns): binary search, and a new entry shifts the tail up
        const auto insert = [&](int32_t v) {
            int lo = 0, hi = n;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (set[mid * RB + t] < v) lo = mid + 1; else hi = mid;
            }
            if (lo < n && set[lo * RB + t] == v) return;
            if (n >= FS_ROWCOL_CAP) { over = true; return; }
            for (int k = n; k > lo; --k) set[k * RB + t] = set[(k - 1) * RB + t];
            set[lo * RB + t] = v;
            ++n;
        };
        if (nd == 4) {
            // four cells at a time: their keys, then the four vertices of each as one 16-byte load, all in flight together
            for (int32_t j0 = first; j0 < last; j0 += 4) {
                int32_t q[4];
                int4 v4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = j0 + u < last ? (int32_t)(keys[j0 + u] & 0xffffffffULL) : -1;
#pragma unroll
                for (int u = 0; u < 4; ++u) v4[u] = reinterpret_cast<const int4*>(cell_dofs)[q[u] >= 0 ? (q[u] >> 2) : 0];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (q[u] < 0) continue;
                    insert(v4[u].x); insert(v4[u].y); insert(v4[u].z); insert(v4[u].w);
                }
            }
        } else {
            // (the dofs of a cell asked for together, the next cell's key already under way)
            int32_t q = first < last ? (int32_t)(keys[first] & 0xffffffffULL) : 0;
            for (int32_t j = first; j < last; ++j) {
                const int64_t c = q / nd;
                if (j + 1 < last) q = (int32_t)(keys[j + 1] & 0xffffffffULL);
                int32_t v[10];
#pragma unroll
                for (int b = 0; b < 10; ++b) v[b] = cell_dofs[c * nd + (b < nd ? b : 0)];
#pragma unroll
                for (int b = 0; b < 10; ++b)
                    if (b < nd) insert(v[b]);
            }
        }
        if (over) {
            atomicAdd(overflow, 1);
            cnt[r] = 0;
            continue;
        }
        cnt[r] = n;
        for (int k = 0; k < n; ++k) cols[(int64_t)k * n_rows + r] = set[k * RB + t];
    }
}
__global__ void k_row_columns_pack(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ cols, int64_t n_rows,
                                   int32_t* __restrict__ colidx) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_rows; r += stride) {
        const int32_t p0 = rowptr[r], n = rowptr[r + 1] - p0;
        for (int k = 0; k < n; ++k) colidx[p0 + k] = cols[(int64_t)k * n_rows + r];
    }
}

// unique keys of a sorted list: flag = first of its run; after the exclusive sum of the flags, pos[i] = where key i goes
__global__ void k_unique_flags(const uint64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}
__global__ void k_unique_scatter(const uint64_t* __restrict__ keys, const int32_t* __restrict__ pos, int64_t n,
                                 uint64_t* __restrict__ out, int64_t* __restrict__ count) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint64_t k = keys[i];
        const bool first = i == 0 || k != keys[i - 1];
        if (first) out[pos[i]] = k;
        if (i == n - 1) count[0] = (int64_t)pos[i] + (first ? 1 : 0);
    }
}

__global__ void k_split_keys(const uint64_t* __restrict__ keys, int64_t nnz, int32_t* __restrict__ colidx) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < nnz; i += stride) colidx[i] = (int32_t)(keys[i] & 0xffffffffULL);
}

// rowptr[r] = first index whose key >= (r << 32)
__global__ void k_rowptr(const uint64_t* __restrict__ keys, int64_t nnz, int64_t n_rows,
                         int32_t* __restrict__ rowptr) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r <= n_rows; r += stride) {
        const uint64_t target = (uint64_t)r << 32;
        int64_t lo = 0, hi = nnz;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < target) lo = mid + 1; else hi = mid;
        }
        rowptr[r] = (int32_t)lo;
    }
}

#define FS_DIA_CAP 72  // most distinct offsets a DIA slice may have (P2 vertex rows of a Kuhn mesh: 65)

// One wavefront per slice.  Computes the longest row and the sorted set of distinct (col - row)
// offsets used by the 64 rows (each row's columns are sorted, so a per-lane cursor walks them).
// A slice is stored in DIA form when that set is small and costs fewer bytes than SELL:
// nd*8 B/row (values only) against len*12 B/row (values + 4-B columns).
//
// SPLIT slices.  On a CG2 space of a structured mesh the rows of an edge class are numbered line by line, and where a slice
// runs over the end of a mesh line every offset shifts by the rows the line end skips: the union over the 64 rows is then
// twice the size and the slice used to fall back to SELL (36 % of the slices of the unit cube at any n > 64, measured).
// Such a slice is two DIA pieces: rows [0, split) share one offset list, rows [split, 64) another.  Both lists are kept
// (dia_off layout per DIA slice: [split][list A: width][list B: width, only if split < 64]); the product picks the list by
// lane, still without streaming a column index.  The split is searched among the rows whose first column offset or length
// differs from the row before (at most FS_SPLIT_TRIES candidates), keeping the narrowest admissible width.
#define FS_SPLIT_TRIES 12
// distinct offsets of the rows of the lanes with `active`, ascending; stops counting beyond cap + 1.  out (lane 0 writes) may be null.
__device__ __forceinline__ int fs_union_offsets(bool active, const int32_t* __restrict__ colidx, int32_t start, int32_t len,
                                                int64_t r, int lane, int cap, int32_t* __restrict__ out) {
    int cur = 0, nd = 0;
    const int64_t BIG = (int64_t)1 << 40;
    while (true) {
        const int64_t mine = (active && cur < len) ? (int64_t)colidx[start + cur] - r : BIG;
        int64_t m = mine;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int64_t o = __shfl_xor(m, off, 64);
            m = o < m ? o : m;
        }
        if (m == BIG) break;
        if (out && nd < cap && lane == 0) out[nd] = (int32_t)m;
        ++nd;
        if (mine == m) ++cur;
        if (nd > cap) break;
    }
    return nd;
}

__global__ void __launch_bounds__(FS_BLOCK) k_slice_analyze(const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ colidx, int64_t n_rows,
                                                            int64_t n_slices, int allow_dia,
                                                            int64_t* __restrict__ slice_entries,
                                                            int32_t* __restrict__ dia_cnt,
                                                            int32_t* __restrict__ tmp_off, int32_t* __restrict__ split_at,
                                                            int* __restrict__ max_w,
                                                            int* __restrict__ n_dia, unsigned long long* __restrict__ dia_entries,
                                                            int64_t line = 0) {
    // line > 0 (the lattice-ordered shadow of a CG2 box space, fs_lattice.hip): the rows come in mesh lines of that many, and the one
    // place a slice can be split is where a line ends inside it (rows ALTERNATE between two patterns there: every row is a
    // `candidate` of the search below, which tries the first dozen)
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    int my_max_w = 0, my_n_dia = 0;             // (lane 0: the wave's slices together, one round of atomics after the loop)
    unsigned long long my_dia_entries = 0ull;
    for (; s < n_slices; s += stride) {
        const int64_t r = s * FS_SLICE + lane;
        int32_t start = 0, len = 0;
        if (r < n_rows) {
            start = rowptr[r];
            len = rowptr[r + 1] - start;
        }
        int maxlen = len;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
        int32_t* const offA = tmp_off + s * (2 * FS_DIA_CAP);
        int32_t* const offB = offA + FS_DIA_CAP;
        auto admissible = [&](int nd) { return nd > 0 && nd <= FS_DIA_CAP && nd <= 255 && (int64_t)nd * 8 * 10 <= (int64_t)maxlen * 12 * 9; };
        int nd = 0, split = FS_SLICE;
        bool dia = false;
        if (allow_dia) {
            nd = fs_union_offsets(true, colidx, start, len, r, lane, FS_DIA_CAP, offA);
            dia = admissible(nd);
            if (!dia && allow_dia > 1) {
                // candidate split rows: the pattern of a row differs from the row before it
                const int64_t first = len > 0 ? (int64_t)colidx[start] - r : 0;
                const int64_t pf = __shfl_up(first, 1, 64);
                const int pl = __shfl_up(len, 1, 64);
                unsigned long long cand = __ballot(lane > 0 && r < n_rows && (first != pf || len != pl));
                if (line > 0) {
                    const int64_t to_end = line - (s * FS_SLICE) % line;          // rows of the slice before the next line starts
                    cand = to_end < FS_SLICE ? (1ull << to_end) : 0ull;
                }
                int best = 0, best_w = FS_DIA_CAP + 1;
                for (int t = 0; t < FS_SPLIT_TRIES && cand; ++t) {
                    const int sp = __ffsll((long long)cand) - 1;
                    cand &= cand - 1;
                    const int na = fs_union_offsets(lane < sp, colidx, start, len, r, lane, FS_DIA_CAP, nullptr);
                    if (na > FS_DIA_CAP) continue;
                    const int nb = fs_union_offsets(lane >= sp, colidx, start, len, r, lane, FS_DIA_CAP, nullptr);
                    const int w = na > nb ? na : nb;
                    if (admissible(w) && w < best_w) { best_w = w; best = sp; }
                }
                if (best > 0) {
                    const int na = fs_union_offsets(lane < best, colidx, start, len, r, lane, FS_DIA_CAP, offA);
                    const int nb = fs_union_offsets(lane >= best, colidx, start, len, r, lane, FS_DIA_CAP, offB);
                    // the shorter list is padded with its own last offset: those entries hold the value 0 and are marked
                    // non-structural by k_fill_sell (its cursor has passed the column by then)
                    if (lane == 0) {
                        for (int k = na; k < best_w; ++k) offA[k] = na > 0 ? offA[na - 1] : 0;
                        for (int k = nb; k < best_w; ++k) offB[k] = nb > 0 ? offB[nb - 1] : 0;      // (a piece of rows beyond n_rows is empty)
                    }
                    nd = best_w;
                    split = best;
                    dia = true;
                }
            }
        }
        const int width = dia ? nd : maxlen;
        if (lane == 0) {
            slice_entries[s] = (int64_t)width * FS_SLICE;
            dia_cnt[s] = dia ? 1 + nd * (split < FS_SLICE ? 2 : 1) : 0;      // ints this slice takes in dia_off
            split_at[s] = split;
            my_max_w = width > my_max_w ? width : my_max_w;
            if (dia) {
                ++my_n_dia;
                my_dia_entries += (unsigned long long)width * FS_SLICE;
            }
        }
    }
    if (lane == 0) {
        if (my_max_w > __hip_atomic_load(max_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_w, my_max_w);      // (a look first: thousands of equal maxima on one address)
        if (my_n_dia) {
            atomicAdd(n_dia, my_n_dia);
            atomicAdd(dia_entries, my_dia_entries);
        }
    }
}

__global__ void k_dia_ptr(const int32_t* __restrict__ dia_cnt, const int32_t* __restrict__ dia_scan, int64_t n_slices,
                          const int32_t* __restrict__ tmp_off, const int32_t* __restrict__ split_at, int32_t* __restrict__ dia_ptr,
                          int32_t* __restrict__ dia_off) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; s < n_slices; s += stride) {
        const int cnt = dia_cnt[s];
        dia_ptr[s] = cnt > 0 ? dia_scan[s] : -1;
        if (cnt == 0) continue;
        const int split = split_at[s];
        const int nd = split < FS_SLICE ? (cnt - 1) / 2 : cnt - 1;
        int32_t* out = dia_off + dia_scan[s];
        out[0] = split;
        for (int k = 0; k < nd; ++k) out[1 + k] = tmp_off[s * (2 * FS_DIA_CAP) + k];
        if (split < FS_SLICE)
            for (int k = 0; k < nd; ++k) out[1 + nd + k] = tmp_off[s * (2 * FS_DIA_CAP) + FS_DIA_CAP + k];
    }
}

// Slices with the same offset list share ONE copy of it: on a structured mesh nearly every slice has the list of its neighbours,
// and a product that walks 15 000 private copies misses the scalar cache on every slice (1 MB of offsets at 1 M rows) where it
// could hit the same 64 bytes.  Two passes over a hash table of list hashes, result independent of the insertion order: pass 1
// leaves in every occupied slot the SMALLEST slice index carrying that hash, pass 2 lets a slice adopt that slice's list if the
// contents really are equal (a collision keeps its own copy).  The unused copies stay where they are.
constexpr int FS_DEDUP_CAP = 1 << 16;
__device__ __forceinline__ unsigned long long dia_list_hash(const int32_t* __restrict__ l, int cnt) {
    unsigned long long h = 1469598103934665603ull ^ (unsigned long long)cnt;
    for (int k = 0; k < cnt; ++k) { h = (h ^ (unsigned long long)(uint32_t)l[k]) * 1099511628211ull; h ^= h >> 31; }
    return h ? h : 1ull;
}
__global__ void k_dia_dedup_insert(int64_t n_slices, const int32_t* __restrict__ dia_cnt, const int32_t* __restrict__ dia_ptr,
                                   const int32_t* __restrict__ dia_off, unsigned long long* keys, int32_t* rep) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; s < n_slices; s += stride) {
        const int cnt = dia_cnt[s];
        if (cnt == 0) continue;
        const unsigned long long h = dia_list_hash(dia_off + dia_ptr[s], cnt);
        int slot = (int)(h & (FS_DEDUP_CAP - 1));
        for (int probe = 0; probe < 256; ++probe) {      // (a full table: the slice keeps its own copy)
            unsigned long long old = keys[slot];
            if (old == 0ull) old = atomicCAS(&keys[slot], 0ull, h);
            if (old == 0ull || old == h) {
                // (the minimum only falls: a value read without the atomic that is already below s settles it - nearly every
                // slice of a box carries the same list)
                if (__hip_atomic_load(&rep[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (int32_t)s) atomicMin(&rep[slot], (int32_t)s);
                break;
            }
            slot = (slot + 1) & (FS_DEDUP_CAP - 1);
        }
    }
}
__global__ void k_dia_dedup_adopt(int64_t n_slices, const int32_t* __restrict__ dia_cnt, const int32_t* __restrict__ dia_ptr,
                                  const int32_t* __restrict__ dia_off, const unsigned long long* __restrict__ keys,
                                  const int32_t* __restrict__ rep, int32_t* __restrict__ dia_ptr_out) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; s < n_slices; s += stride) {
        const int cnt = dia_cnt[s];
        int32_t dp = dia_ptr[s];
        if (cnt > 0) {
            const int32_t* mine = dia_off + dp;
            const unsigned long long h = dia_list_hash(mine, cnt);
            int slot = (int)(h & (FS_DEDUP_CAP - 1));
            for (int probe = 0; probe < 256; ++probe) {
                const unsigned long long k = keys[slot];
                if (k == 0ull) break;
                if (k == h) {
                    const int32_t r = rep[slot];
                    if (r >= 0 && r < s && dia_cnt[r] == cnt) {
                        const int32_t* other = dia_off + dia_ptr[r];
                        bool same = true;
                        for (int q = 0; q < cnt; ++q) same = same && other[q] == mine[q];
                        if (same) dp = dia_ptr[r];
                    }
                    break;
                }
                slot = (slot + 1) & (FS_DEDUP_CAP - 1);
            }
        }
        dia_ptr_out[s] = dp;
    }
}

// column of every stored entry.  SELL slice: entry k of the row, padding = ~row.  DIA slice:
// row + offset k when the row really has that column, otherwise ~clamp(row + offset).
__global__ void __launch_bounds__(FS_BLOCK) k_fill_sell(const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ colidx, int64_t n_rows,
                                                        int64_t n_cols, int64_t n_slices,
                                                        const int64_t* __restrict__ slice_ptr,
                                                        const int32_t* __restrict__ dia_ptr,
                                                        const int32_t* __restrict__ dia_off,
                                                        int32_t* __restrict__ sell_col) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t base = slice_ptr[s];
        const int width = (int)((slice_ptr[s + 1] - base) >> 6);
        const int64_t r = s * FS_SLICE + lane;
        int32_t start = 0, len = 0;
        if (r < n_rows) {
            start = rowptr[r];
            len = rowptr[r + 1] - start;
        }
        const int32_t dp = dia_ptr[s];
        if (dp < 0) {
            const int32_t self = (int32_t)(r < n_rows ? r : n_rows - 1);
            for (int k = 0; k < width; ++k)
                sell_col[base + (int64_t)k * FS_SLICE + lane] = k < len ? colidx[start + k] : ~self;
        } else {
            int cur = 0;
            const int split = dia_off[dp];
            const int32_t* __restrict__ op = dia_off + dp + 1 + (lane >= split ? width : 0);      // this lane's offset list
            for (int k = 0; k < width; ++k) {
                int64_t c = r + (int64_t)op[k];
                bool structural = false;
                if (cur < len && (int64_t)colidx[start + cur] == c) {
                    structural = true;
                    ++cur;
                }
                if (c < 0) c = 0;
                if (c > n_cols - 1) c = n_cols - 1;
                sell_col[base + (int64_t)k * FS_SLICE + lane] = structural ? (int32_t)c : ~(int32_t)c;
            }
        }
    }
}

// storage position of column `target` in the row of `lane` of the slice starting at `base`, -1 if absent
__device__ __forceinline__ int fs_find_pos(const int32_t* __restrict__ sell_col, int64_t base_lane, int width,
                                           int32_t target) {
    for (int k = 0; k < width; ++k)
        if (sell_col[base_lane + (int64_t)k * FS_SLICE] == target) return k;
    return -1;
}

// slots[(a*4+b)*nc + c] = stored entry of (cells[c][a], cells[c][b]) or -1 when the row is not owned
__global__ void k_slots(const int32_t* __restrict__ cells, int64_t nc, int64_t n_rows,
                        const int32_t* __restrict__ sell_col, const int64_t* __restrict__ slice_ptr,
                        int32_t* __restrict__ slots, int* __restrict__ err) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < nc; c += stride) {
        const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
        const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int32_t row = v[a];
            const bool owned = row < n_rows;
            int64_t base = 0;
            int width = 0;
            if (owned) {
                const int64_t sp0 = slice_ptr[row >> 6];
                width = (int)((slice_ptr[(row >> 6) + 1] - sp0) >> 6);
                base = sp0 + (row & 63);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                int32_t slot = -1;
                if (owned) {
                    const int k = fs_find_pos(sell_col, base, width, v[b]);
                    if (k >= 0) slot = (int32_t)(base + (int64_t)k * FS_SLICE);
                    else atomicAdd(err, 1);
                }
                slots[(int64_t)(a * 4 + b) * nc + c] = slot;
            }
        }
    }
}

// ---- row-gather incidence tables ----------------------------------------------------------------
// key = (vertex << 32) | (cell*4 + local vertex) for every owned (cell, vertex) incidence
__global__ void k_inc_keys(const int32_t* __restrict__ cell_dofs, int64_t n_inc, int64_t n_rows,
                           uint64_t* __restrict__ keys) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; q < n_inc; q += stride) {
        const int32_t v = cell_dofs[q];
        keys[q] = v < n_rows ? (((uint64_t)(uint32_t)v << 32) | (uint64_t)q) : ~0ULL;
    }
}

// one wavefront per slice: width = most incidences of a row in the slice
__global__ void __launch_bounds__(FS_BLOCK) k_inc_width(const int32_t* __restrict__ inc_ptr, int64_t n_rows,
                                                        int64_t n_slices, int64_t* __restrict__ slice_entries,
                                                        int* __restrict__ max_cnt) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    int my_max = 0;         // (the wave's slices together: one atomic per wave after the loop)
    for (; s < n_slices; s += stride) {
        const int64_t r = s * FS_SLICE + lane;
        int cnt = 0;
        if (r < n_rows) cnt = inc_ptr[r + 1] - inc_ptr[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt = max(cnt, __shfl_xor(cnt, off, 64));
        if (lane == 0) slice_entries[s] = (int64_t)cnt * FS_SLICE;
        my_max = cnt > my_max ? cnt : my_max;
    }
    if (lane == 0 && my_max > 0) atomicMax(max_cnt, my_max);
}

__global__ void __launch_bounds__(FS_BLOCK) k_inc_fill(const uint64_t* __restrict__ keys,
                                                       const int32_t* __restrict__ inc_ptr,
                                                       const int32_t* __restrict__ cell_dofs, int nd,
                                                       const int32_t* __restrict__ sell_col,
                                                       const int64_t* __restrict__ slice_ptr, int64_t n_rows,
                                                       int64_t n_slices, const int64_t* __restrict__ inc_slice_ptr,
                                                       int64_t inc_entries, int32_t* __restrict__ inc_cell,
                                                       uint32_t* __restrict__ inc_pos, int* __restrict__ err) {
    const int lane = threadIdx.x & 63;
    const int words = (nd + 3) >> 2;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t base = inc_slice_ptr[s];
        const int width = (int)((inc_slice_ptr[s + 1] - base) >> 6);
        const int64_t mbase = slice_ptr[s] + lane;
        const int mwidth = (int)((slice_ptr[s + 1] - slice_ptr[s]) >> 6);
        const int64_t r = s * FS_SLICE + lane;
        int32_t first = 0, cnt = 0;
        if (r < n_rows) {
            first = inc_ptr[r];
            cnt = inc_ptr[r + 1] - first;
        }
        // Rows of at most 16 / 32 stored entries (CG1: 15 on a box, about 30 at most on a file mesh): the row's columns are read
        // ONCE into registers and every cell dof is looked up there - the search through memory (fs_find_pos: up to `mwidth`
        // dependent loads for each of the nd dofs of each of the row's cells) was 20 ms of the set-up at 10 M rows.
        const auto fill = [&](auto cw_tag) {
            constexpr int CW = decltype(cw_tag)::value;
            int32_t cols[CW > 0 ? CW : 1];
            if (CW > 0) {
#pragma unroll
                for (int k = 0; k < CW; ++k) cols[k] = k < mwidth ? sell_col[mbase + (int64_t)k * FS_SLICE] : -1;
            }
            if (CW > 0 && nd == 4) {
                // tetrahedra / CG1: four incidences at a time - their keys, then the four vertices of each cell as ONE 16-byte
                // load, all in flight together (incidence by incidence every row waited for two dependent loads per step)
                for (int j0 = 0; j0 < width; j0 += 4) {
                    int32_t q[4];
                    int4 v4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) q[u] = j0 + u < cnt ? (int32_t)(keys[first + j0 + u] & 0xffffffffULL) : -1;
#pragma unroll
                    for (int u = 0; u < 4; ++u) v4[u] = reinterpret_cast<const int4*>(cell_dofs)[q[u] >= 0 ? (q[u] >> 2) : 0];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (j0 + u >= width) break;
                        uint32_t packed0 = 0u;
                        if (q[u] >= 0) {
                            const int32_t tg[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                int k = -1;
#pragma unroll
                                for (int t = CW - 1; t >= 0; --t) k = cols[t] == tg[b] ? t : k;
                                if (k < 0) atomicAdd(err, 1);
                                packed0 |= (uint32_t)(k & 255) << (8 * b);
                            }
                        }
                        const int64_t e = base + (int64_t)(j0 + u) * FS_SLICE + lane;
                        inc_cell[e] = q[u];
                        inc_pos[e] = packed0;
                    }
                }
                return;
            }
            for (int j = 0; j < width; ++j) {
                int32_t q = -1;
                uint32_t packed[3] = {0u, 0u, 0u};
                if (j < cnt) {
                    q = (int32_t)(keys[first + j] & 0xffffffffULL);
                    const int64_t c = q / nd;
                    for (int b = 0; b < nd; ++b) {
                        const int32_t target = cell_dofs[c * nd + b];
                        int k = -1;
                        if (CW > 0) {
#pragma unroll
                            for (int t = CW - 1; t >= 0; --t) k = cols[t] == target ? t : k;      // (the first match, as the search finds it)
                        } else {
                            k = fs_find_pos(sell_col, mbase, mwidth, target);
                        }
                        if (k < 0 || k > 255) atomicAdd(err, 1);
                        packed[b >> 2] |= (uint32_t)(k & 255) << (8 * (b & 3));
                    }
                }
                const int64_t e = base + (int64_t)j * FS_SLICE + lane;
                inc_cell[e] = q;
                for (int w = 0; w < words; ++w) inc_pos[(int64_t)w * inc_entries + e] = packed[w];
            }
        };
        if (mwidth <= 16) fill(std::integral_constant<int, 16>{});
        else if (mwidth <= 32) fill(std::integral_constant<int, 32>{});
        else fill(std::integral_constant<int, 0>{});
    }
}

// generic element: slots[(a*nd+b)*nc + c] for the nd dofs of cell c (thread per (cell, a))
__global__ void k_slots_generic(const int32_t* __restrict__ cell_dofs, int nd, int64_t nc, int64_t n_rows,
                                const int32_t* __restrict__ sell_col, const int64_t* __restrict__ slice_ptr,
                                int32_t* __restrict__ slots, int* __restrict__ err) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nc * nd; t += stride) {
        const int64_t c = t / nd;
        const int a = (int)(t - c * nd);
        const int32_t row = cell_dofs[c * nd + a];
        const bool owned = row < n_rows;
        int64_t base = 0;
        int width = 0;
        if (owned) {
            const int64_t sp0 = slice_ptr[row >> 6];
            width = (int)((slice_ptr[(row >> 6) + 1] - sp0) >> 6);
            base = sp0 + (row & 63);
        }
        for (int b = 0; b < nd; ++b) {
            int32_t slot = -1;
            if (owned) {
                const int k = fs_find_pos(sell_col, base, width, cell_dofs[c * nd + b]);
                if (k >= 0) slot = (int32_t)(base + (int64_t)k * FS_SLICE);
                else atomicAdd(err, 1);
            }
            slots[(int64_t)(a * nd + b) * nc + c] = slot;
        }
    }
}

__global__ void k_compact_tri_cells(const int32_t* __restrict__ cells4, int64_t nc, int32_t* __restrict__ cells3) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < nc; c += stride) {
        cells3[3 * c] = cells4[4 * c];
        cells3[3 * c + 1] = cells4[4 * c + 1];
        cells3[3 * c + 2] = cells4[4 * c + 2];
    }
}

// ---- inverse of the slot table (two-pass / gather assembly of block spaces) -----------------------------------------
__global__ void k_gmap_keys(const int32_t* __restrict__ slots, int64_t nc, int nd2, int32_t sentinel, int32_t* __restrict__ key,
                            int32_t* __restrict__ src) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nc * nd2; t += stride) {          // slots[ab*nc + c] -> source index c*nd2 + ab
        const int64_t ab = t / nc, c = t - ab * nc;
        const int32_t sl = slots[t];
        key[t] = sl >= 0 ? sl : sentinel;
        src[t] = (int32_t)(c * nd2 + ab);
    }
}
__global__ void k_gmap_lower_bound(const int32_t* __restrict__ keys, int64_t n, int64_t n_entries, int32_t* __restrict__ ptr) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e <= n_entries; e += stride) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < (int32_t)e) lo = mid + 1; else hi = mid;
        }
        ptr[e] = (int32_t)lo;
    }
}
int fs_space_build_gather_map(fs_space_s* sp, hipStream_t s) {
    FS_REQUIRE(sp->slots.p, "gather map: the space has no slot table");
    const int nd2 = sp->ndof_cell * sp->ndof_cell;
    const int64_t n = sp->mesh->nc * nd2;
    FS_REQUIRE(n < (int64_t)INT32_MAX && sp->sell_entries < (int64_t)INT32_MAX - 1, "gather map: mesh too large for 32-bit element indices");
    dbuf<int32_t> k_in, k_out, v_in;
    FS_CHECK(k_in.alloc(n));
    FS_CHECK(k_out.alloc(n));
    FS_CHECK(v_in.alloc(n));
    FS_CHECK(sp->gmap_src.alloc(n));
    FS_CHECK(sp->gmap_ptr.alloc(sp->sell_entries + 1));
    hipLaunchKernelGGL(k_gmap_keys, dim3(fs_grid_for(n)), dim3(FS_BLOCK), 0, s, sp->slots.p, sp->mesh->nc, nd2, (int32_t)INT32_MAX, k_in.p, v_in.p);
    FS_KERNEL_CHECK();
    size_t tb = 0;
    FS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in.p, k_out.p, v_in.p, sp->gmap_src.p, (int)n, 0, 32, s));
    dbuf<char> tmp;
    FS_CHECK(tmp.alloc((int64_t)tb + 16));
    FS_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, k_in.p, k_out.p, v_in.p, sp->gmap_src.p, (int)n, 0, 32, s));
    hipLaunchKernelGGL(k_gmap_lower_bound, dim3(fs_grid_for(sp->sell_entries + 1)), dim3(FS_BLOCK), 0, s, k_out.p, n, sp->sell_entries, sp->gmap_ptr.p);
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

// bounding box of the vertices: one workgroup, grid-stride (set-up time)
// bounding box of the vertices: partial boxes per workgroup (part[b][6]), the last stage by one workgroup over the partials
// (one workgroup over all the vertices took 6 ms at 10 M of them)
__global__ void __launch_bounds__(1024) k_bbox(const double* __restrict__ xyz4, int64_t nv, int stride4, double* __restrict__ out) {
    __shared__ double lo[3][16], hi[3][16];
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    // stride4 = 4: vertices (x, y, z, pad), min and max of each coordinate; stride4 = 6: partial boxes (3 minima, 3 maxima)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x)
        for (int d = 0; d < 3; ++d) {
            const double a = xyz4[stride4 * i + d], b2 = xyz4[stride4 * i + (stride4 == 6 ? 3 + d : d)];
            mn[d] = a < mn[d] ? a : mn[d];
            mx[d] = b2 > mx[d] ? b2 : mx[d];
        }
    for (int d = 0; d < 3; ++d) {
        for (int off = 32; off > 0; off >>= 1) {
            const double a = __shfl_down(mn[d], off, 64), b2 = __shfl_down(mx[d], off, 64);
            mn[d] = a < mn[d] ? a : mn[d];
            mx[d] = b2 > mx[d] ? b2 : mx[d];
        }
        if ((threadIdx.x & 63) == 0) { lo[d][threadIdx.x >> 6] = mn[d]; hi[d][threadIdx.x >> 6] = mx[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        double a = lo[d][0], b2 = hi[d][0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { a = lo[d][w] < a ? lo[d][w] : a; b2 = hi[d][w] > b2 ? hi[d][w] : b2; }
        out[6 * (int64_t)blockIdx.x + d] = a;
        out[6 * (int64_t)blockIdx.x + 3 + d] = b2;
    }
}
// How fast does each coordinate vary along the row numbering?  cnt[a] = number of consecutive owned rows whose
// a-coordinates differ: a lexicographic box numbering gives n, n/N_fast, n/(N_fast N_mid) for its three axes.
__global__ void __launch_bounds__(FS_BLOCK) k_axis_counts(const double* __restrict__ xyz4, int64_t n_rows, unsigned long long* __restrict__ cnt) {
    unsigned long long c[3] = {0, 0, 0};
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i + 1 < n_rows; i += stride)
        for (int d = 0; d < 3; ++d) c[d] += xyz4[4 * i + d] != xyz4[4 * (i + 1) + d];
    for (int d = 0; d < 3; ++d) {
        for (int off = 32; off > 0; off >>= 1) c[d] += __shfl_down(c[d], off, 64);
        if ((threadIdx.x & 63) == 0 && c[d]) atomicAdd(&cnt[d], c[d]);
    }
}
// Pencil order of the slices of a lexicographically numbered box (fast / mid / slow axis): the mid axis is cut into
// n_tiles bands; a band is swept plane by plane along the slow axis, inside a plane in row order.  Rows that are
// neighbours across planes (offsets +-N_fast*N_mid and friends) are then one band-plane apart in the sweep instead of a
// whole plane: what the mirrored reads of the symmetric product (fs_krylov.hip) and the x windows need to meet in L2.
__global__ void k_slice_keys_pencil(int64_t n_slices, const double* __restrict__ xyz4, const double* __restrict__ box, int mid,
                                    int slow, int n_tiles, int n_slow, uint32_t* __restrict__ key, int32_t* __restrict__ idx) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; s < n_slices; s += stride) {
        const int64_t v = s * FS_SLICE;
        const double em = box[3 + mid] - box[mid], es = box[3 + slow] - box[slow];
        double tm = em > 0.0 ? (xyz4[4 * v + mid] - box[mid]) / em * n_tiles : 0.0;
        tm = tm < 0.0 ? 0.0 : (tm > n_tiles - 1.0 ? n_tiles - 1.0 : tm);
        double ts = es > 0.0 ? (xyz4[4 * v + slow] - box[slow]) / es * (n_slow - 1) + 0.5 : 0.0;
        ts = ts < 0.0 ? 0.0 : (ts > n_slow - 1.0 ? n_slow - 1.0 : ts);
        key[s] = ((uint32_t)tm << 20) | (uint32_t)ts;
        idx[s] = (int32_t)s;
    }
}
__device__ __forceinline__ uint32_t fs_spread3(uint32_t v) {      // 10 bits -> every third bit
    v &= 0x3ff;
    v = (v | (v << 16)) & 0x030000ff;
    v = (v | (v << 8)) & 0x0300f00f;
    v = (v | (v << 4)) & 0x030c30c3;
    v = (v | (v << 2)) & 0x09249249;
    return v;
}
// spatial key of a slice = Morton code (bits per axis given) of the vertex its first row sits at (an edge node: its
// endpoint of smaller id); equal keys keep their row order (stable sort)
__global__ void k_slice_keys(int64_t n_slices, int64_t nvo, int64_t n_rows, const int32_t* __restrict__ edges,
                             const double* __restrict__ xyz4, const double* __restrict__ box, int bits, uint32_t* __restrict__ key,
                             int32_t* __restrict__ idx) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double cells = (double)(1 << bits);
    for (; s < n_slices; s += stride) {
        const int64_t r = s * FS_SLICE;
        int64_t v;
        if (r < nvo || !edges) v = r;
        else {
            const int64_t j = r - nvo;
            const int32_t a = edges[2 * j], b2 = edges[2 * j + 1];
            v = a < b2 ? a : b2;
        }
        if (bits < 0) {      // sweep order of the vertices
            key[s] = (uint32_t)v;
            idx[s] = (int32_t)s;
            continue;
        }
        uint32_t q[3];
        for (int d = 0; d < 3; ++d) {
            const double ext = box[3 + d] - box[d];
            double t = ext > 0.0 ? (xyz4[4 * v + d] - box[d]) / ext * cells : 0.0;
            t = t < 0.0 ? 0.0 : (t > cells - 1.0 ? cells - 1.0 : t);
            q[d] = (uint32_t)t;
        }
        key[s] = fs_spread3(q[0]) | (fs_spread3(q[1]) << 1) | (fs_spread3(q[2]) << 2);
        idx[s] = (int32_t)s;
    }
}

// ---- API -----------------------------------------------------------------------------------
// extra node couplings (both directions) in the pattern: interior-facet integrals couple the two vertices opposite a facet
__global__ void k_extra_pair_keys(const int32_t* __restrict__ pairs, int64_t n, int64_t n_rows, uint64_t* __restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int32_t a = pairs[2 * i], b = pairs[2 * i + 1];
        keys[2 * i] = a < n_rows ? (((uint64_t)(uint32_t)a << 32) | (uint32_t)b) : ~0ULL;
        keys[2 * i + 1] = b < n_rows ? (((uint64_t)(uint32_t)b << 32) | (uint32_t)a) : ~0ULL;
    }
}

// out[i] = sum of in[0 .. i), count entries (the device scan of this file's code object, for translation units without one)
int fs_scan_exclusive_i32(const int32_t* in, int32_t* out, int64_t count, hipStream_t s) {
    size_t tb = 0;
    FS_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, (int)count, s));
    dbuf<char> tmp;
    FS_CHECK(tmp.alloc((int64_t)tb + 16));
    FS_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, in, out, (int)count, s));
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

// Hybrid SELL-64 / DIA storage of a space from its CSR pattern (sp->rowptr / colidx over sp->n_nodes_owned rows, columns below
// sp->n_nodes_local): slice_ptr, dia_ptr / dia_off, sell_col, the counters.  Step 4 of fs_space_create - and what the solver's
// lattice-ordered shadow of a CG2 box operator is built with (fs_lattice.hip).
int fs_space_build_storage(fs_space_s* sp, hipStream_t s) {
    const int64_t n_rows = sp->n_nodes_owned;
    const int64_t n_slices = (n_rows + FS_SLICE - 1) / FS_SLICE;
    sp->n_slices = n_slices;
    {
        dbuf<int64_t> entries;
        dbuf<int32_t> dia_cnt, dia_scan, tmp_off, split_at;
        dbuf<int> d_max, d_ndia;
        dbuf<unsigned long long> d_dia_entries;
        FS_CHECK(entries.alloc(n_slices + 1));
        FS_CHECK(entries.zero(s));
        FS_CHECK(dia_cnt.alloc(n_slices + 1));
        FS_CHECK(dia_cnt.zero(s));
        FS_CHECK(dia_scan.alloc(n_slices + 1));
        FS_CHECK(tmp_off.alloc(n_slices * 2 * FS_DIA_CAP));
        FS_CHECK(split_at.alloc(n_slices));
        FS_CHECK(d_dia_entries.alloc(1));
        FS_CHECK(d_dia_entries.zero(s));
        FS_CHECK(d_max.alloc(1));
        FS_CHECK(d_max.zero(s));
        FS_CHECK(d_ndia.alloc(1));
        FS_CHECK(d_ndia.zero(s));
        FS_CHECK(sp->slice_ptr.alloc(n_slices + 1));
        FS_CHECK(sp->dia_ptr.alloc(n_slices));
        const char* env = getenv("FS_DISABLE_DIA");
        const char* env_split = getenv("FS_DISABLE_DIA_SPLIT");
        // 0: SELL only, 1: whole-slice DIA only, 2: DIA with split slices (default)
        const int allow_dia = (env && *env && *env != '0') ? 0 : ((env_split && *env_split && *env_split != '0') ? 1 : 2);
        hipLaunchKernelGGL(k_slice_analyze, dim3(fs_grid_for(n_slices * 64, FS_BLOCK, 1024)), dim3(FS_BLOCK), 0, s, sp->rowptr.p, sp->colidx.p, n_rows, n_slices, allow_dia, entries.p, dia_cnt.p, tmp_off.p, split_at.p, d_max.p, d_ndia.p, d_dia_entries.p, sp->dict_line);
        FS_HIP(hipGetLastError());
        size_t tmp_bytes = 0, t2 = 0;
        FS_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, entries.p, sp->slice_ptr.p, (int)(n_slices + 1), s));
        FS_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, t2, dia_cnt.p, dia_scan.p, (int)(n_slices + 1), s));
        if (t2 > tmp_bytes) tmp_bytes = t2;
        dbuf<char> tmp;
        FS_CHECK(tmp.alloc((int64_t)tmp_bytes + 16));
        size_t tb = tmp_bytes;
        FS_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, entries.p, sp->slice_ptr.p, (int)(n_slices + 1), s));
        tb = tmp_bytes;
        FS_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, dia_cnt.p, dia_scan.p, (int)(n_slices + 1), s));
        int64_t total = 0;
        int32_t total_off = 0;
        int h_ndia = 0;
        FS_HIP(hipMemcpyAsync(&total, sp->slice_ptr.p + n_slices, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        FS_HIP(hipMemcpyAsync(&total_off, dia_scan.p + n_slices, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        FS_CHECK(d_ndia.download(&h_ndia, 1, s));
        FS_CHECK(d_max.download(&sp->max_row, 1, s));
        sp->sell_entries = total;
        sp->n_dia_slices = h_ndia;
        unsigned long long h_dia_entries = 0;
        FS_CHECK(d_dia_entries.download(&h_dia_entries, 1, s));
        sp->dia_entries = (int64_t)h_dia_entries;
        FS_CHECK(sp->dia_off.alloc((total_off > 0 ? total_off : 1) + 64));       // (+ 64: the row-dictionary product reads whole rounds of 16 offsets)
        FS_CHECK(sp->dia_off.zero(s));
        hipLaunchKernelGGL(k_dia_ptr, dim3(fs_grid_for(n_slices)), dim3(FS_BLOCK), 0, s, dia_cnt.p, dia_scan.p, n_slices, tmp_off.p, split_at.p, sp->dia_ptr.p, sp->dia_off.p);
        FS_HIP(hipGetLastError());
        static const bool no_dedup = getenv("FS_DIA_DEDUP") && getenv("FS_DIA_DEDUP")[0] == '0';
        if (h_ndia > 0 && !no_dedup) {       // identical offset lists -> one copy (see k_dia_dedup_insert)
            dbuf<unsigned long long> keys;
            dbuf<int32_t> rep, shared;
            FS_CHECK(keys.alloc(FS_DEDUP_CAP));
            FS_CHECK(rep.alloc(FS_DEDUP_CAP));
            FS_CHECK(shared.alloc(n_slices));
            FS_CHECK(keys.zero(s));
            FS_HIP(hipMemsetAsync(rep.p, 0x7f, (size_t)FS_DEDUP_CAP * sizeof(int32_t), s));
            hipLaunchKernelGGL(k_dia_dedup_insert, dim3(fs_grid_for(n_slices)), dim3(FS_BLOCK), 0, s, n_slices, dia_cnt.p, sp->dia_ptr.p, sp->dia_off.p, keys.p, rep.p);
            hipLaunchKernelGGL(k_dia_dedup_adopt, dim3(fs_grid_for(n_slices)), dim3(FS_BLOCK), 0, s, n_slices, dia_cnt.p, sp->dia_ptr.p, sp->dia_off.p, keys.p, rep.p, shared.p);
            FS_HIP(hipGetLastError());
            FS_HIP(hipMemcpyAsync(sp->dia_ptr.p, shared.p, (size_t)n_slices * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        }
        FS_HIP(hipStreamSynchronize(s));
    }
    if (sp->sell_entries >= (int64_t)INT32_MAX) {
        fs_set_error("fs_space_create: storage of %lld entries exceeds int32 slot indexing", (long long)sp->sell_entries);
        return FS_ERR_UNSUPPORTED;
    }
    FS_CHECK(sp->sell_col.alloc(sp->sell_entries));
    hipLaunchKernelGGL(k_fill_sell, dim3(fs_grid_for(n_slices * 64)), dim3(FS_BLOCK), 0, s, sp->rowptr.p, sp->colidx.p, n_rows, sp->n_nodes_local, n_slices, sp->slice_ptr.p, sp->dia_ptr.p, sp->dia_off.p, sp->sell_col.p);
    FS_HIP(hipGetLastError());
    return FS_OK;
}

static int space_create_impl(fs_mesh_t mesh, int family, int degree, int ncomp, int64_t n_extra, const int32_t* extra_pairs,
                             fs_space_t* out);

extern "C" int fs_space_create(fs_mesh_t mesh, int family, int degree, int ncomp, fs_space_t* out) {
    return space_create_impl(mesh, family, degree, ncomp, 0, nullptr, out);
}

extern "C" int fs_space_create_coupled(fs_mesh_t mesh, int family, int degree, int ncomp, int64_t n_pairs,
                                       const int32_t* node_pairs, fs_space_t* out) {
    FS_REQUIRE(n_pairs >= 0 && (n_pairs == 0 || node_pairs), "fs_space_create_coupled: bad pair list");
    FS_REQUIRE(degree == 1 || degree == 2 || n_pairs == 0, "fs_space_create_coupled: extra couplings are built for CG1 / CG2 spaces");
    return space_create_impl(mesh, family, degree, ncomp, n_pairs, node_pairs, out);
}

static int space_create_impl(fs_mesh_t mesh, int family, int degree, int ncomp, int64_t n_extra, const int32_t* extra_pairs,
                             fs_space_t* out) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(mesh && out, "fs_space_create: null pointer");
    // ncomp = 4 on CG2 nodes is the Taylor-Hood block layout (u_x, u_y, u_z, p) of fs_assemble_navier_stokes; on triangles the
    // u_z slot is a dummy unknown (unit row) so that the 2-D system runs through the same block-4 operator and solver
    if (family != FS_FAMILY_CG || (degree != 1 && degree != 2) || (ncomp != 1 && ncomp != 3 && ncomp != 4 && ncomp != 2) ||
        (degree == 1 && ncomp == 4) || (ncomp == 2 && mesh->tdim != 2) || (ncomp == 3 && mesh->tdim == 2)) {
        fs_set_error("fs_space_create: supported spaces are CG1 / CG2 with 1 or 3 components on tetrahedra, CG1 with 1 or 2 components on triangles and the 4-component CG2 node blocks of Taylor-Hood (family=%d degree=%d ncomp=%d on a %dD mesh)",
                     family, degree, ncomp, mesh->tdim);
        return FS_ERR_UNSUPPORTED;
    }
    hipStream_t s = fs_rt().stream;
    const int64_t nc = mesh->nc;
    FS_REQUIRE(mesh->n_owned > 0, "fs_space_create: process owns no vertices");
    fs_space_s* sp = new fs_space_s();
    sp->mesh = mesh;
    sp->degree = degree;
    sp->ncomp = ncomp;
    sp->ndof_cell = 4;
    sp->cell_dofs = mesh->cells.p;
    sp->n_nodes_local = mesh->nv;
    sp->n_nodes_owned = mesh->n_owned;

#define FS_SP(call)                \
    do {                           \
        int rc__ = (call);         \
        if (rc__ != FS_OK) {       \
            delete sp;             \
            return rc__;           \
        }                          \
    } while (0)
#define FS_SP_HIP(call)                                                                       \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            fs_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            delete sp;                                                                        \
            return FS_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

    if (mesh->tdim == 2) {
        // triangles: CG1 scalar and 2-vector spaces; the compact [nc][3] dof table feeds the generic pattern / incidence /
        // slot-table code
        FS_SP(sp->cell_dofs_store.alloc(3 * nc));
        hipLaunchKernelGGL(k_compact_tri_cells, dim3(fs_grid_for(nc)), dim3(FS_BLOCK), 0, s, mesh->cells.p, nc, sp->cell_dofs_store.p);
        FS_SP_HIP(hipGetLastError());
        sp->ndof_cell = 3;
        sp->cell_dofs = sp->cell_dofs_store.p;
    }
    if (degree == 2) {
        // edge nodes: unique (min,max) vertex pairs in lexicographic order = oracle/DOLFIN-style edge numbering
        const int tdim = mesh->tdim;
        const int n_cell_edges = tdim == 2 ? 3 : 6, n_cell_nodes = tdim == 2 ? 6 : 10;
        const int64_t n_ek = (int64_t)n_cell_edges * nc;
        FS_REQUIRE(n_ek < (int64_t)INT32_MAX, "fs_space_create: edge keys exceed int32");
        dbuf<uint64_t> ka, kb;
        dbuf<int> d_count;
        FS_SP(ka.alloc(n_ek));
        FS_SP(kb.alloc(n_ek));
        FS_SP(d_count.alloc(1));
        size_t tb1 = 0, tb2 = 0, tb3 = 0, tb4 = 0;
        FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb1, ka.p, kb.p, (int)n_ek, 0, 64, s));
        FS_SP_HIP(hipcub::DeviceSelect::Unique(nullptr, tb2, kb.p, ka.p, d_count.p, (int)n_ek, s));
        dbuf<uint32_t> da, db;
        FS_SP(da.alloc(n_ek));
        FS_SP(db.alloc(n_ek));
        FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb3, da.p, db.p, (int)n_ek, 0, 32, s));
        FS_SP_HIP(hipcub::DeviceSelect::Unique(nullptr, tb4, db.p, da.p, d_count.p, (int)n_ek, s));
        size_t tbm = tb1 > tb2 ? tb1 : tb2;
        if (tb3 > tbm) tbm = tb3;
        if (tb4 > tbm) tbm = tb4;
        dbuf<char> tmp;
        FS_SP(tmp.alloc((int64_t)tbm + 16));
        int h_ne = 0;
        // pass 0: lexicographic keys -> unique edges -> number of distinct v1 - v0; pass 1 (structured meshes
        // only, <= 16 distinct differences): regroup the edges by that difference
        for (int pass = 0; pass < 2; ++pass) {
            hipLaunchKernelGGL(k_edge_keys, dim3(fs_grid_for(nc)), dim3(FS_BLOCK), 0, s, mesh->cells.p, nc, sp->edge_grouped, ka.p, tdim);
            FS_SP_HIP(hipGetLastError());
            size_t tb = tbm;
            FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, ka.p, kb.p, (int)n_ek, 0, 64, s));
            tb = tbm;
            FS_SP_HIP(hipcub::DeviceSelect::Unique(tmp.p, tb, kb.p, ka.p, d_count.p, (int)n_ek, s));
            FS_SP(d_count.download(&h_ne, 1, s));
            if (pass == 1) break;
            hipLaunchKernelGGL(k_edge_deltas, dim3(fs_grid_for(h_ne)), dim3(FS_BLOCK), 0, s, ka.p, (int64_t)h_ne, da.p);
            FS_SP_HIP(hipGetLastError());
            tb = tbm;
            FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, da.p, db.p, h_ne, 0, 32, s));
            tb = tbm;
            FS_SP_HIP(hipcub::DeviceSelect::Unique(tmp.p, tb, db.p, da.p, d_count.p, h_ne, s));
            int h_nd = 0;
            FS_SP(d_count.download(&h_nd, 1, s));
            if (h_nd > 16) break;
            sp->edge_grouped = 1;
        }
        sp->n_edges = h_ne;
        FS_SP(sp->edges.alloc(2 * (int64_t)h_ne));
        FS_SP(sp->edge_keys.alloc(h_ne));
        FS_SP(sp->edge_node.alloc(h_ne));
        FS_SP(sp->cell_dofs_store.alloc((int64_t)n_cell_nodes * nc));
        FS_SP_HIP(hipMemcpyAsync(sp->edge_keys.p, ka.p, (size_t)h_ne * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
        // owned edges first (stable: key order inside each class)
        int h_neo = h_ne;
        {
            dbuf<int32_t> flag, flag2, idx, order;
            FS_SP(flag.alloc(h_ne)); FS_SP(flag2.alloc(h_ne)); FS_SP(idx.alloc(h_ne)); FS_SP(order.alloc(h_ne));
            hipLaunchKernelGGL(k_edge_ghost_flag, dim3(fs_grid_for(h_ne)), dim3(FS_BLOCK), 0, s, ka.p, (int64_t)h_ne, sp->edge_grouped, mesh->gid.p, mesh->n_owned, flag.p, idx.p);
            FS_SP_HIP(hipGetLastError());
            size_t tbs = 0;
            FS_SP_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tbs, flag.p, flag2.p, idx.p, order.p, h_ne, 0, 1, s));
            dbuf<char> tmps;
            FS_SP(tmps.alloc((int64_t)tbs + 16));
            FS_SP_HIP(hipcub::DeviceRadixSort::SortPairs(tmps.p, tbs, flag.p, flag2.p, idx.p, order.p, h_ne, 0, 1, s));
            if (mesh->n_owned != mesh->nv) {
                size_t tbr = 0;
                dbuf<int32_t> dsum;
                FS_SP(dsum.alloc(1));
                FS_SP_HIP(hipcub::DeviceReduce::Sum(nullptr, tbr, flag.p, dsum.p, h_ne, s));
                dbuf<char> tmpr;
                FS_SP(tmpr.alloc((int64_t)tbr + 16));
                FS_SP_HIP(hipcub::DeviceReduce::Sum(tmpr.p, tbr, flag.p, dsum.p, h_ne, s));
                int32_t h_ghost = 0;
                FS_SP(dsum.download(&h_ghost, 1, s));
                h_neo = h_ne - h_ghost;
            }
            hipLaunchKernelGGL(k_edge_nodes, dim3(fs_grid_for(h_ne)), dim3(FS_BLOCK), 0, s, order.p, (int64_t)h_ne, (int64_t)h_neo, mesh->n_owned, mesh->nv, ka.p, sp->edge_grouped, sp->edge_node.p, sp->edges.p);
            FS_SP_HIP(hipGetLastError());
            FS_SP_HIP(hipStreamSynchronize(s));
        }
        sp->n_edges_owned = h_neo;
        hipLaunchKernelGGL(k_p2_cell_dofs, dim3(fs_grid_for(nc)), dim3(FS_BLOCK), 0, s, mesh->cells.p, nc, mesh->n_owned, (int64_t)h_neo, ka.p, (int64_t)h_ne, sp->edge_grouped, sp->edge_node.p, sp->cell_dofs_store.p, tdim);
        FS_SP_HIP(hipGetLastError());
        FS_SP_HIP(hipStreamSynchronize(s));
        sp->ndof_cell = n_cell_nodes;
        sp->cell_dofs = sp->cell_dofs_store.p;
        sp->n_nodes_local = mesh->nv + h_ne;
        sp->n_nodes_owned = mesh->n_owned + h_neo;
        FS_REQUIRE(sp->n_nodes_local < (int64_t)INT32_MAX, "fs_space_create: CG2 dof count exceeds int32");
    }
    sp->n_dofs_local = sp->n_nodes_local * ncomp;
    sp->n_dofs_owned = sp->n_nodes_owned * ncomp;
    sp->pos_words = (sp->ndof_cell + 3) / 4;
    const int nd = sp->ndof_cell;
    const int64_t n_rows = sp->n_nodes_owned;

    // 1. keys
    const int64_t n_pairs = (int64_t)nd * (nd - 1);
    const int64_t n_keys = n_pairs * nc + n_rows + 2 * n_extra;
    for (int64_t i = 0; i < 2 * n_extra; ++i)
        if (extra_pairs[i] < 0 || extra_pairs[i] >= sp->n_nodes_local) {
            fs_set_error("fs_space_create_coupled: pair %lld names node %d", (long long)(i / 2), extra_pairs[i]);
            delete sp;
            return FS_ERR_INVALID;
        }
    if (ncomp == 1 && (int64_t)nd * nc >= (int64_t)INT32_MAX) {      // the assembly tables name (cell, local dof) in 32 bits
        fs_set_error("fs_space_create: %lld cell-dof incidences exceed the 32-bit assembly tables (largest P1 cube: n = 447)",
                     (long long)((int64_t)nd * nc));
        delete sp;
        return FS_ERR_UNSUPPORTED;
    }
    // (the sort and the unique pass take 64-bit item counts: a P1 space of 80 M dofs has 6.4e9 keys, 2 x 51 GB for a moment)
    int64_t nnz = 0;
    // CG1 / CG2 without extra couplings: the pattern row by row from the sorted (vertex, cell) incidences (k_row_columns); the sorted
    // incidences are kept for the assembly tables of step 5a.  FS_PATTERN_BY_ROWS=0: the sorted-keys path below.
    dbuf<uint64_t> inc_sorted;
    dbuf<int32_t> inc_ptr_pre;
    bool have_inc = false, by_rows = false;
    static const bool by_rows_off = getenv("FS_PATTERN_BY_ROWS") && getenv("FS_PATTERN_BY_ROWS")[0] == '0';
    if (!by_rows_off && n_extra == 0 && nd <= 10 && (int64_t)nd * nc < (int64_t)INT32_MAX && n_rows > 0) {
        const int rc_cap = nd <= 4 ? 32 : 160;
        const int64_t n_inc = (int64_t)nd * nc;
        dbuf<uint64_t> ka;
        FS_SP(ka.alloc(n_inc));
        FS_SP(inc_sorted.alloc(n_inc));
        FS_SP(inc_ptr_pre.alloc(n_rows + 1));
        hipLaunchKernelGGL(k_inc_keys, dim3(fs_grid_for(n_inc)), dim3(FS_BLOCK), 0, s, sp->cell_dofs, n_inc, n_rows, ka.p);
        FS_SP_HIP(hipGetLastError());
        // (the low half of a key is the incidence's own index, ascending as generated: a STABLE sort over the row bits alone leaves
        // the incidences of a row in that order - 3 radix passes instead of 8; the bits of n_rows itself keep the sentinel last)
        const int br_inc = fs_bits_for((uint64_t)n_rows);
        size_t tmp_bytes = 0, tmp2 = 0;
        dbuf<int32_t> cnt, cols;
        dbuf<int> d_over;
        FS_SP(cnt.alloc(n_rows + 1));
        // (the unsorted keys are dead once the sort has run: the per-row sets are written over them where they fit - one
        // allocation of 11 GB less at 86 M rows, where the set-up is mostly hipMalloc / hipFree of such blocks)
        const bool cols_in_ka = (int64_t)rc_cap * n_rows * (int64_t)sizeof(int32_t) <= n_inc * (int64_t)sizeof(uint64_t);
        if (!cols_in_ka) FS_SP(cols.alloc((int64_t)rc_cap * n_rows));
        int32_t* const cols_p = cols_in_ka ? reinterpret_cast<int32_t*>(ka.p) : cols.p;
        FS_SP(d_over.alloc(1));
        FS_SP(d_over.zero(s));
        FS_SP(sp->rowptr.alloc(n_rows + 1));
        FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, ka.p, inc_sorted.p, (int)n_inc, 32, 32 + br_inc, s));
        FS_SP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, cnt.p, sp->rowptr.p, (int)(n_rows + 1), s));
        if (tmp2 > tmp_bytes) tmp_bytes = tmp2;
        dbuf<char> tmp;
        FS_SP(tmp.alloc((int64_t)tmp_bytes + 16));
        size_t tb = tmp_bytes;
        FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, ka.p, inc_sorted.p, (int)n_inc, 32, 32 + br_inc, s));
        // rows of other ranks (key ~0) sort to the end: count of valid keys = first index of the sentinel row
        hipLaunchKernelGGL(k_rowptr, dim3(fs_grid_for(n_rows + 1)), dim3(FS_BLOCK), 0, s, inc_sorted.p, n_inc, n_rows, inc_ptr_pre.p);
        have_inc = true;
        FS_SP_HIP(hipMemsetAsync(cnt.p + n_rows, 0, sizeof(int32_t), s));
        if (nd <= 4)
            hipLaunchKernelGGL((k_row_columns<32, FS_BLOCK>), dim3(fs_grid_for(n_rows, FS_BLOCK, 8192)), dim3(FS_BLOCK), 0, s, inc_sorted.p, inc_ptr_pre.p,
                               sp->cell_dofs, nd, n_rows, cnt.p, cols_p, d_over.p);
        else
            hipLaunchKernelGGL((k_row_columns<160, 64>), dim3(fs_grid_for(n_rows, 64, 32768)), dim3(64), 0, s, inc_sorted.p, inc_ptr_pre.p,
                               sp->cell_dofs, nd, n_rows, cnt.p, cols_p, d_over.p);
        FS_SP_HIP(hipGetLastError());
        tb = tmp_bytes;
        FS_SP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, cnt.p, sp->rowptr.p, (int)(n_rows + 1), s));
        int h_over = 0;
        int32_t h_nnz = 0;
        FS_SP_HIP(hipMemcpyAsync(&h_nnz, sp->rowptr.p + n_rows, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        FS_SP(d_over.download(&h_over, 1, s));
        if (h_over == 0) {
            nnz = h_nnz;
            sp->nnz_nodes = nnz;
            FS_SP(sp->colidx.alloc(nnz));
            hipLaunchKernelGGL(k_row_columns_pack, dim3(fs_grid_for(n_rows, FS_BLOCK, 8192)), dim3(FS_BLOCK), 0, s, sp->rowptr.p, cols_p, n_rows, sp->colidx.p);
            FS_SP_HIP(hipGetLastError());
            FS_SP_HIP(hipStreamSynchronize(s));
            by_rows = true;
        }
        if (getenv("FS_SPACE_DEBUG"))
            fprintf(stderr, "[fs_symbolic] sparsity pattern row by row: %s (%lld rows, %lld node pairs)\n",
                    by_rows ? "yes" : "no, a row has more neighbours than the set holds: sorted keys", (long long)n_rows, (long long)nnz);
    }
    if (!by_rows) {
        dbuf<uint64_t> keys_a, keys_b;
        dbuf<int64_t> d_count;
        FS_SP(keys_a.alloc(n_keys));
        FS_SP(keys_b.alloc(n_keys));
        FS_SP(d_count.alloc(1));
        hipLaunchKernelGGL(k_pair_keys, dim3(fs_grid_for(nc)), dim3(FS_BLOCK), 0, s, sp->cell_dofs, nd, nc, n_rows, keys_a.p);
        hipLaunchKernelGGL(k_diag_keys, dim3(fs_grid_for(n_rows)), dim3(FS_BLOCK), 0, s, n_rows, keys_a.p + n_pairs * nc);
        FS_SP_HIP(hipGetLastError());
        if (n_extra > 0) {
            dbuf<int32_t> dp;
            FS_SP(dp.alloc(2 * n_extra));
            FS_SP(dp.upload(extra_pairs, 2 * n_extra, s));
            hipLaunchKernelGGL(k_extra_pair_keys, dim3(fs_grid_for(n_extra)), dim3(FS_BLOCK), 0, s, dp.p, n_extra, n_rows, keys_a.p + n_pairs * nc + n_rows);
            FS_SP_HIP(hipGetLastError());
            FS_SP_HIP(hipStreamSynchronize(s));
        }
        // 2. sort + unique.  A key is (row << 32) | column with zero bits between the column's and the row's: two stable sorts over
        // the bits that can differ - the columns' [0, bc), then the rows' [32, 32 + br) - are 6 radix passes at 1 M and 10 M rows
        // where the sort over all 64 bits was 8 (each pass streams the 12 keys per cell twice).  br counts the bits of n_rows itself:
        // the sentinel's row bits, all ones, then stay above every owned row and it still sorts to the end.
        const int bc = fs_bits_for((uint64_t)sp->n_nodes_local), br = fs_bits_for((uint64_t)n_rows);
        size_t tmp_bytes = 0, tmp1 = 0;
        FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, keys_a.p, keys_b.p, n_keys, 0, bc, s));
        FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp1, keys_b.p, keys_a.p, n_keys, 32, 32 + br, s));
        if (tmp1 > tmp_bytes) tmp_bytes = tmp1;
        size_t tmp2 = 0;
        FS_SP_HIP(hipcub::DeviceSelect::Unique(nullptr, tmp2, keys_a.p, keys_b.p, d_count.p, n_keys, s));
        if (tmp2 > tmp_bytes) tmp_bytes = tmp2;
        dbuf<char> tmp;
        FS_SP(tmp.alloc((int64_t)tmp_bytes + 16));
        size_t tb = tmp_bytes;
        FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, keys_a.p, keys_b.p, n_keys, 0, bc, s));
        tb = tmp_bytes;
        FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, keys_b.p, keys_a.p, n_keys, 32, 32 + br, s));
        if (n_keys < (int64_t)INT32_MAX) {
            // first-of-its-run flags, their exclusive sum, a scatter: 0.5 ms for the 71 M keys of a 1 M-row CG1 space where
            // DeviceSelect::Unique took 1.7 ms (its look-back partition runs at 75 GB/s on 8-byte items)
            dbuf<int32_t> upos;
            FS_SP(upos.alloc(n_keys));
            size_t tb3 = 0;
            FS_SP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb3, upos.p, upos.p, (int)n_keys, s));
            dbuf<char> tmp3;
            FS_SP(tmp3.alloc((int64_t)tb3 + 16));
            hipLaunchKernelGGL(k_unique_flags, dim3(fs_grid_for(n_keys)), dim3(FS_BLOCK), 0, s, keys_a.p, n_keys, upos.p);
            FS_SP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp3.p, tb3, upos.p, upos.p, (int)n_keys, s));
            hipLaunchKernelGGL(k_unique_scatter, dim3(fs_grid_for(n_keys)), dim3(FS_BLOCK), 0, s, keys_a.p, upos.p, n_keys, keys_b.p, d_count.p);
            FS_SP_HIP(hipGetLastError());
        } else {
            tb = tmp_bytes;
            FS_SP_HIP(hipcub::DeviceSelect::Unique(tmp.p, tb, keys_a.p, keys_b.p, d_count.p, n_keys, s));
        }
        keys_a.swap(keys_b);            // (the unique keys are what follows reads from keys_a)
        int64_t h_count = 0;
        FS_SP(d_count.download(&h_count, 1, s));
        nnz = h_count;
        if (nnz >= (int64_t)INT32_MAX) {
            fs_set_error("fs_space_create: %lld coupled node pairs exceed the 32-bit row pointers", (long long)nnz);
            delete sp;
            return FS_ERR_UNSUPPORTED;
        }
        // the sentinel (rows owned by other ranks) is the last unique key when present
        if (nnz > 0) {
            uint64_t last = 0;
            FS_SP_HIP(hipMemcpyAsync(&last, keys_a.p + (nnz - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, s));
            FS_SP_HIP(hipStreamSynchronize(s));
            if (last == ~0ULL) nnz -= 1;
        }
        sp->nnz_nodes = nnz;
        // 3. CSR
        FS_SP(sp->rowptr.alloc(n_rows + 1));
        FS_SP(sp->colidx.alloc(nnz));
        hipLaunchKernelGGL(k_split_keys, dim3(fs_grid_for(nnz)), dim3(FS_BLOCK), 0, s, keys_a.p, nnz, sp->colidx.p);
        hipLaunchKernelGGL(k_rowptr, dim3(fs_grid_for(n_rows + 1)), dim3(FS_BLOCK), 0, s, keys_a.p, nnz, n_rows, sp->rowptr.p);
        FS_SP_HIP(hipGetLastError());
        FS_SP_HIP(hipStreamSynchronize(s));
    }
    // 4. hybrid SELL-64 / DIA storage
    FS_SP(fs_space_build_storage(sp, s));
    const int64_t n_slices = sp->n_slices;
    // processing order of the slices (SpMV, gather assembly).  CG2: by the vertex the slice's first node sits at, i.e.
    // the sweep order of the vertices with the edge classes interleaved (bits = -1).  Measured inside the CG solve on
    // MI355X (round 1): P2 10 M DOF 1071 us unordered, 703 us by vertex, 735-755 us in Morton order (4-7 bits per
    // axis); P1 (already in sweep order) 328 us unordered, 342 us in Morton order.
    // CG1 spaces too large for the caches (> 2 M rows) on a lexicographically numbered box: PENCIL order (-2; bands of
    // ~4096 rows along the middle axis, each swept plane by plane).  Round 2, 10 M DOF: gather assembly 3.97 -> 2.91 ms (the
    // cells and coordinates of the plane below / above are still in L2 when they are needed again), CG product unchanged
    // within run-to-run noise (0.298-0.323 ms either way).  Anything else keeps its natural order.
    // FS_SLICE_ORDER = 0 | -1 | -2 | <Morton bits per axis> overrides.
    int order_bits = (degree == 2 && mesh->tdim == 3) ? -1 : (n_slices > 32768 && mesh->tdim == 3 ? -2 : 0);
    if (const char* e = getenv("FS_SLICE_ORDER")) order_bits = atoi(e);
    if (getenv("FS_NO_SLICE_ORDER")) order_bits = 0;
    if (order_bits > 10) order_bits = 10;
    if (order_bits != 0) {
        dbuf<uint32_t> k_in, k_out;
        dbuf<int32_t> v_in;
        dbuf<double> box, box_parts;
        FS_SP(box.alloc(6));
        const int bbox_grid = (int)std::min<int64_t>(512, (mesh->nv + 1023) / 1024);
        if (bbox_grid > 1) {
            FS_SP(box_parts.alloc(6 * (int64_t)bbox_grid));
            hipLaunchKernelGGL(k_bbox, dim3(bbox_grid), dim3(1024), 0, s, mesh->xyz.p, mesh->nv, 4, box_parts.p);
            hipLaunchKernelGGL(k_bbox, dim3(1), dim3(1024), 0, s, box_parts.p, (int64_t)bbox_grid, 6, box.p);
        } else {
            hipLaunchKernelGGL(k_bbox, dim3(1), dim3(1024), 0, s, mesh->xyz.p, mesh->nv, 4, box.p);
        }
        FS_SP(k_in.alloc(n_slices));
        FS_SP(k_out.alloc(n_slices));
        FS_SP(v_in.alloc(n_slices));
        FS_SP(sp->slice_order.alloc(n_slices));
        bool pencil = false, skip = false;
        if (order_bits == -2) {
            // pencil order: needs the numbering to be lexicographic on a box (axis speeds well separated)
            dbuf<unsigned long long> cnt;
            FS_SP(cnt.alloc(3));
            FS_SP(cnt.zero(s));
            hipLaunchKernelGGL(k_axis_counts, dim3(fs_grid_for(n_rows)), dim3(FS_BLOCK), 0, s, mesh->xyz.p, n_rows, cnt.p);
            unsigned long long hc[3] = {0, 0, 0};
            FS_SP(cnt.download(hc, 3, s));
            int ax[3] = {0, 1, 2};
            std::sort(ax, ax + 3, [&](int a, int b2) { return hc[a] > hc[b2]; });
            const int mid = ax[1], slow = ax[2];
            const double n_slow = (double)hc[slow] + 1.0, n_mid = ((double)hc[mid] + 1.0) / n_slow;
            const double n_fast = (double)n_rows / (n_mid * n_slow);
            static const double tile_rows = getenv("FS_TILE_ROWS") ? atof(getenv("FS_TILE_ROWS")) : 4096.0;
            if (mesh->tdim == 3 && hc[ax[0]] > 4 * hc[mid] && hc[mid] > 4 * hc[slow] && n_slow >= 4 && n_slow < (1 << 20) && n_mid >= 2) {
                double lines = tile_rows / n_fast;
                lines = lines < 1.0 ? 1.0 : lines;
                int n_tiles = (int)(n_mid / lines + 0.5);
                n_tiles = n_tiles < 1 ? 1 : (n_tiles > 4000 ? 4000 : n_tiles);
                hipLaunchKernelGGL(k_slice_keys_pencil, dim3(fs_grid_for(n_slices)), dim3(FS_BLOCK), 0, s, n_slices, mesh->xyz.p, box.p, mid, slow,
                                   n_tiles, (int)n_slow, k_in.p, v_in.p);
                pencil = true;
                if (getenv("FS_SPACE_DEBUG"))
                    fprintf(stderr, "[fs_space] pencil order: axes fast %d mid %d slow %d, %g x %g x %g nodes, %d bands\n", ax[0], mid, slow, n_fast, n_mid, n_slow, n_tiles);
            }
        }
        if (!pencil && order_bits == -2 && degree == 1) skip = true;     // no box numbering found: CG1 rows keep their order
        if (skip) sp->slice_order.release();
        else if (!pencil)
            hipLaunchKernelGGL(k_slice_keys, dim3(fs_grid_for(n_slices)), dim3(FS_BLOCK), 0, s, n_slices, mesh->n_owned, n_rows,
                               degree == 2 ? sp->edges.p : (const int32_t*)nullptr, mesh->xyz.p, box.p, order_bits == -2 ? -1 : order_bits, k_in.p, v_in.p);
        FS_SP_HIP(hipGetLastError());
        size_t tbs = 0;
        if (!skip) {
        FS_SP_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tbs, k_in.p, k_out.p, v_in.p, sp->slice_order.p, (int)n_slices, 0, 32, s));
        dbuf<char> tmps;
        FS_SP(tmps.alloc((int64_t)tbs + 16));
        FS_SP_HIP(hipcub::DeviceRadixSort::SortPairs(tmps.p, tbs, k_in.p, k_out.p, v_in.p, sp->slice_order.p, (int)n_slices, 0, 32, s));
        FS_SP_HIP(hipStreamSynchronize(s));
        }
    }
    if (ncomp == 1 && sp->max_row <= 255) {
        // 5a. scalar spaces: row-gather incidence tables (deterministic, atomic-free assembly)
        const int64_t n_inc = (int64_t)nd * nc;
        FS_REQUIRE(n_inc < (int64_t)INT32_MAX, "fs_space_create: cell-vertex incidences exceed int32");
        dbuf<uint64_t> ka, kb;
        dbuf<int32_t> inc_ptr;
        dbuf<int64_t> entries;
        dbuf<int> d_max, d_err;
        if (have_inc) {             // (sorted for the pattern already)
            kb.swap(inc_sorted);
            inc_ptr.swap(inc_ptr_pre);
        } else {
            FS_SP(ka.alloc(n_inc));
            FS_SP(kb.alloc(n_inc));
            FS_SP(inc_ptr.alloc(n_rows + 1));
        }
        FS_SP(entries.alloc(n_slices + 1));
        FS_SP(entries.zero(s));
        FS_SP(d_max.alloc(1));
        FS_SP(d_max.zero(s));
        FS_SP(d_err.alloc(1));
        FS_SP(d_err.zero(s));
        if (!have_inc) {
            hipLaunchKernelGGL(k_inc_keys, dim3(fs_grid_for(n_inc)), dim3(FS_BLOCK), 0, s, sp->cell_dofs, n_inc, n_rows, ka.p);
            FS_SP_HIP(hipGetLastError());
        }
        size_t tmp_bytes = 0;
        // (the low half of a key is the incidence's own index, ascending as generated: a STABLE sort over the row bits alone leaves
        // the incidences of a row in that order - 3 radix passes instead of 8; br as above keeps the sentinel last)
        const int br_inc = fs_bits_for((uint64_t)n_rows);
        FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, ka.p, kb.p, (int)n_inc, 32, 32 + br_inc, s));
        size_t tmp2 = 0;
        FS_SP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, entries.p, entries.p, (int)(n_slices + 1), s));
        if (tmp2 > tmp_bytes) tmp_bytes = tmp2;
        dbuf<char> tmp;
        FS_SP(tmp.alloc((int64_t)tmp_bytes + 16));
        size_t tb = tmp_bytes;
        if (!have_inc) {
            FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, ka.p, kb.p, (int)n_inc, 32, 32 + br_inc, s));
            // rows of other ranks (key ~0) sort to the end: count of valid keys = first index of the sentinel row
            hipLaunchKernelGGL(k_rowptr, dim3(fs_grid_for(n_rows + 1)), dim3(FS_BLOCK), 0, s, kb.p, n_inc, n_rows, inc_ptr.p);
        }
        FS_SP(sp->inc_slice_ptr.alloc(n_slices + 1));
        hipLaunchKernelGGL(k_inc_width, dim3(fs_grid_for(n_slices * 64, FS_BLOCK, 512)), dim3(FS_BLOCK), 0, s, inc_ptr.p, n_rows, n_slices, entries.p, d_max.p);
        FS_SP_HIP(hipGetLastError());
        tb = tmp_bytes;
        FS_SP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, entries.p, sp->inc_slice_ptr.p, (int)(n_slices + 1), s));
        int64_t total = 0;
        FS_SP_HIP(hipMemcpyAsync(&total, sp->inc_slice_ptr.p + n_slices, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        FS_SP(d_max.download(&sp->inc_max, 1, s));
        sp->inc_entries = total;
        FS_SP(sp->inc_cell.alloc(total));
        FS_SP(sp->inc_pos.alloc(total * sp->pos_words));
        hipLaunchKernelGGL(k_inc_fill, dim3(fs_grid_for(n_slices * 64)), dim3(FS_BLOCK), 0, s, kb.p, inc_ptr.p, sp->cell_dofs, nd, sp->sell_col.p, sp->slice_ptr.p, n_rows, n_slices, sp->inc_slice_ptr.p, total, sp->inc_cell.p, sp->inc_pos.p, d_err.p);
        FS_SP_HIP(hipGetLastError());
        int h_err = 0;
        FS_SP(d_err.download(&h_err, 1, s));
        if (h_err != 0) {
            fs_set_error("fs_space_create: internal error, %d incidences missing from the sparsity pattern", h_err);
            delete sp;
            return FS_ERR_INVALID;
        }
    } else {
        // 5b. vector spaces: slot table for the scatter assembly
        dbuf<int> d_err;
        FS_SP(d_err.alloc(1));
        FS_SP(d_err.zero(s));
        FS_SP(sp->slots.alloc((int64_t)nd * nd * nc));
        if (nd == 4)
            hipLaunchKernelGGL(k_slots, dim3(fs_grid_for(nc)), dim3(FS_BLOCK), 0, s, mesh->cells.p, nc, n_rows, sp->sell_col.p, sp->slice_ptr.p, sp->slots.p, d_err.p);
        else
            hipLaunchKernelGGL(k_slots_generic, dim3(fs_grid_for(nc * nd)), dim3(FS_BLOCK), 0, s, sp->cell_dofs, nd, nc, n_rows, sp->sell_col.p, sp->slice_ptr.p, sp->slots.p, d_err.p);
        FS_SP_HIP(hipGetLastError());
        int h_err = 0;
        FS_SP(d_err.download(&h_err, 1, s));
        if (h_err != 0) {
            fs_set_error("fs_space_create: internal error, %d cell pairs missing from the sparsity pattern", h_err);
            delete sp;
            return FS_ERR_INVALID;
        }
    }
#undef FS_SP
#undef FS_SP_HIP
    *out = sp;
    return FS_OK;
}

extern "C" int fs_space_info(fs_space_t space, int64_t* n_dofs_local, int64_t* n_dofs_owned, int64_t* nnz,
                             int64_t* sell_entries) {
    FS_REQUIRE(space, "fs_space_info: null space");
    if (n_dofs_local) *n_dofs_local = space->n_dofs_local;
    if (n_dofs_owned) *n_dofs_owned = space->n_dofs_owned;
    if (nnz) *nnz = space->nnz_nodes * space->ncomp * space->ncomp;
    if (sell_entries) *sell_entries = space->sell_entries * space->ncomp * space->ncomp;
    return FS_OK;
}

extern "C" int fs_space_destroy(fs_space_t space) {
    delete space;
    return FS_OK;
}

extern "C" int fs_space_format_info(fs_space_t space, int64_t* n_slices, int64_t* n_dia_slices, int64_t* spmv_bytes) {
    FS_REQUIRE(space, "fs_space_format_info: null space");
    if (n_slices) *n_slices = space->n_slices;
    if (n_dia_slices) *n_dia_slices = space->n_dia_slices;
    if (spmv_bytes) {
        // bytes the SpMV actually streams for the matrix: 8 B per stored value (x block) + 4 B per stored
        // column of SELL slices (DIA slices read one offset per entry row, negligible)
        const int64_t bs2 = (int64_t)space->ncomp * space->ncomp;
        *spmv_bytes = space->sell_entries * bs2 * 8 + (space->sell_entries - space->dia_entries) * 4;
    }
    return FS_OK;
}

extern "C" int fs_space_get_edges(fs_space_t space, int64_t* n_edges, int32_t* edges) {
    FS_REQUIRE(space, "fs_space_get_edges: null space");
    if (n_edges) *n_edges = space->n_edges;
    if (edges && space->n_edges) FS_CHECK(space->edges.download(edges, 2 * space->n_edges, fs_rt().stream));
    return FS_OK;
}

// ---- locality order of an uploaded mesh -------------------------------------------------------------------------------
// DOLFIN renumbers the dofs of every FunctionSpace for locality when it builds the dofmap (reorder_dofs_serial behind
// FunctionSpace(...), SolverBase.py:260-275): a mesh file's vertex order says nothing about which vertices are neighbours.
// The equivalent here is an order of the VERTICES (the dof nodes of CG1; CG2 edge nodes follow their end points) along the
// Morton curve of their coordinates - 21 bits per axis, one 64-bit radix sort on the device - and of the CELLS by the
// smallest new index of their vertices (a stable 32-bit sort, so cells keep their file order among equals).  Rows 64 by 64
// (a SELL slice, one wavefront) then gather x from a few nearby cache lines, and the contiguous eighth of the rows one XCD
// sweeps is one compact region of the mesh whose x entries stay in that XCD's L2.  The caller applies the two
// permutations when it uploads the mesh (fenicssolver_amd/fem.py does, for file meshes) and maps results back: the
// numbering the user sees never changes.
__device__ __forceinline__ uint64_t fs_spread21(uint64_t v) {      // ...abc -> ...a00b00c
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// Hilbert index of a point with 21-bit integer coordinates (Skilling, "Programming the Hilbert curve", 2004: the transpose
// form, then the bits interleaved).  Unlike the Morton curve the Hilbert curve never jumps: consecutive indices are always
// neighbouring cells of the lattice, so the rows of a SELL slice and of the contiguous chunk an XCD sweeps form compact blobs.
__device__ __forceinline__ uint64_t fs_hilbert_key(uint32_t X0, uint32_t X1, uint32_t X2, int n) {
    uint32_t X[3] = {X0, X1, X2};
    const uint32_t M = 1u << 20;
    for (uint32_t Q = M; Q > 1; Q >>= 1) {         // inverse undo
        const uint32_t P = Q - 1;
        for (int i = 0; i < n; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    for (int i = 1; i < n; ++i) X[i] ^= X[i - 1];   // Gray encode
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1)
        if (X[n - 1] & Q) t ^= Q - 1;
    for (int i = 0; i < n; ++i) X[i] ^= t;
    // interleave: bit b of X[0] is the most significant of the three at level b
    if (n == 3) return fs_spread21(X[0]) << 2 | fs_spread21(X[1]) << 1 | fs_spread21(X[2]);
    return fs_spread21(X[0]) << 1 | fs_spread21(X[1]);      // (2-D: 3-way spread of two words still orders correctly)
}

__global__ void k_morton_keys(int64_t nv, int gdim, const double* __restrict__ xyz, double x0, double y0, double z0,
                              double sx, double sy, double sz, int curve, uint64_t* __restrict__ key, int32_t* __restrict__ idx) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < nv; i += stride) {
        const double* p = xyz + i * gdim;
        const uint64_t a = (uint64_t)fmin(fmax((p[0] - x0) * sx, 0.0), 2097151.0);
        const uint64_t b = (uint64_t)fmin(fmax((p[1] - y0) * sy, 0.0), 2097151.0);
        const uint64_t c = gdim == 3 ? (uint64_t)fmin(fmax((p[2] - z0) * sz, 0.0), 2097151.0) : 0ull;
        key[i] = curve == 1 ? fs_hilbert_key((uint32_t)a, (uint32_t)b, (uint32_t)c, gdim)
                            : (fs_spread21(a) | fs_spread21(b) << 1 | fs_spread21(c) << 2);
        idx[i] = (int32_t)i;
    }
}

__global__ void k_invert_order(int64_t n, const int32_t* __restrict__ order, int32_t* __restrict__ rank) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) rank[order[i]] = (int32_t)i;
}

__global__ void k_cell_first_vertex(int64_t nc, int vpc, const int32_t* __restrict__ cells, const int32_t* __restrict__ rank,
                                    uint32_t* __restrict__ key, int32_t* __restrict__ idx) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < nc; c += stride) {
        int32_t m = INT32_MAX;
        for (int j = 0; j < vpc; ++j) {
            const int32_t r = rank[cells[c * vpc + j]];
            m = r < m ? r : m;
        }
        key[c] = (uint32_t)m;
        idx[c] = (int32_t)c;
    }
}

__global__ void k_renumbered_xyz(int64_t nv, const int32_t* __restrict__ order, const double* __restrict__ xyz3, double* __restrict__ xyz4,
                                 int64_t* __restrict__ gid) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < nv; k += stride) {
        const int64_t o = order[k];
        xyz4[4 * k] = xyz3[3 * o]; xyz4[4 * k + 1] = xyz3[3 * o + 1]; xyz4[4 * k + 2] = xyz3[3 * o + 2]; xyz4[4 * k + 3] = 0.0;
        gid[k] = o;
    }
}
__global__ void k_renumbered_cells(int64_t nc, const int32_t* __restrict__ cell_order, const int32_t* __restrict__ cells_in,
                                   const int32_t* __restrict__ rank, int32_t* __restrict__ cells_out) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < nc; c += stride) {
        const int4 v = reinterpret_cast<const int4*>(cells_in)[cell_order[c]];
        reinterpret_cast<int4*>(cells_out)[c] = make_int4(rank[v.x], rank[v.y], rank[v.z], rank[v.w]);     // the cell's own vertex order is kept
    }
}

// vertex_order[k] = old id of new vertex k, cell_order[c] = old id of new cell c.  mesh_out (optional): the mesh in that order,
// built on the device from the arrays already uploaded for the ordering - coordinates gathered, cells gathered and renamed, global
// ids = the old vertex ids (fs_mesh_create_renumbered; round 4: the host used to re-index 58 M cells with numpy, 6.7 s at 10 M DOF)
static int locality_order_impl(int gdim, int64_t nv, const double* xyz, int64_t nc, const int32_t* cells, int verts_per_cell,
                               int32_t* vertex_order, int32_t* cell_order, fs_mesh_t* mesh_out) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(xyz && cells && vertex_order && cell_order, "fs_mesh_locality_order: null pointer");
    FS_REQUIRE((gdim == 3 && verts_per_cell == 4) || (gdim == 2 && verts_per_cell == 3), "fs_mesh_locality_order: tetrahedra in 3-D or triangles in 2-D");
    FS_REQUIRE(!mesh_out || gdim == 3, "fs_mesh_create_renumbered: tetrahedral meshes");
    FS_REQUIRE(nv > 0 && nc > 0 && nv < (int64_t)INT32_MAX && nc < (int64_t)INT32_MAX, "fs_mesh_locality_order: bad sizes");
    hipStream_t s = fs_rt().stream;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = 0; i < nv; ++i)
        for (int d = 0; d < gdim; ++d) {
            const double v = xyz[i * gdim + d];
            lo[d] = v < lo[d] ? v : lo[d];
            hi[d] = v > hi[d] ? v : hi[d];
        }
    // one cell size for all axes (the curve then visits cubes, not bricks, of an elongated domain)
    double ext = 0.0;
    for (int d = 0; d < gdim; ++d) ext = hi[d] - lo[d] > ext ? hi[d] - lo[d] : ext;
    const double sc = ext > 0.0 ? 2097151.0 / ext : 0.0;
    for (int64_t i = 0; i < nc * verts_per_cell; ++i)
        FS_REQUIRE(cells[i] >= 0 && cells[i] < nv, "fs_mesh_locality_order: cell %lld references vertex %d outside [0,%lld)",
                   (long long)(i / verts_per_cell), cells[i], (long long)nv);
    dbuf<double> dx;
    dbuf<uint64_t> k_in, k_out;
    dbuf<int32_t> v_in, v_out, rank, dc, c_in, c_out;
    dbuf<uint32_t> ck_in, ck_out;
    dbuf<char> tmp;
    FS_CHECK(dx.alloc(nv * gdim));
    FS_CHECK(dx.upload(xyz, nv * gdim, s));
    FS_CHECK(k_in.alloc(nv)); FS_CHECK(k_out.alloc(nv)); FS_CHECK(v_in.alloc(nv)); FS_CHECK(v_out.alloc(nv)); FS_CHECK(rank.alloc(nv));
    static const char* curve_env = getenv("FS_LOCALITY_CURVE");          // "morton" | "hilbert" (default)
    const int curve = (curve_env && curve_env[0] == 'm') ? 0 : 1;
    hipLaunchKernelGGL(k_morton_keys, dim3(fs_grid_for(nv)), dim3(FS_BLOCK), 0, s, nv, gdim, dx.p, lo[0], lo[1], gdim == 3 ? lo[2] : 0.0,
                       sc, sc, sc, curve, k_in.p, v_in.p);
    FS_KERNEL_CHECK();
    size_t tb = 0;
    FS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in.p, k_out.p, v_in.p, v_out.p, (int)nv, 0, 63, s));
    FS_CHECK(tmp.alloc((int64_t)tb + 16));
    FS_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, k_in.p, k_out.p, v_in.p, v_out.p, (int)nv, 0, 63, s));
    hipLaunchKernelGGL(k_invert_order, dim3(fs_grid_for(nv)), dim3(FS_BLOCK), 0, s, nv, v_out.p, rank.p);
    FS_KERNEL_CHECK();
    FS_CHECK(v_out.download(vertex_order, nv, s));
    FS_CHECK(dc.alloc(nc * verts_per_cell));
    FS_CHECK(dc.upload(cells, nc * verts_per_cell, s));
    FS_CHECK(ck_in.alloc(nc)); FS_CHECK(ck_out.alloc(nc)); FS_CHECK(c_in.alloc(nc)); FS_CHECK(c_out.alloc(nc));
    hipLaunchKernelGGL(k_cell_first_vertex, dim3(fs_grid_for(nc)), dim3(FS_BLOCK), 0, s, nc, verts_per_cell, dc.p, rank.p, ck_in.p, c_in.p);
    FS_KERNEL_CHECK();
    size_t tb2 = 0;
    FS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb2, ck_in.p, ck_out.p, c_in.p, c_out.p, (int)nc, 0, 32, s));
    if ((int64_t)tb2 + 16 > tmp.n) FS_CHECK(tmp.alloc((int64_t)tb2 + 16));
    FS_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb2, ck_in.p, ck_out.p, c_in.p, c_out.p, (int)nc, 0, 32, s));
    FS_CHECK(c_out.download(cell_order, nc, s));
    if (mesh_out) {
        fs_mesh_s* m = new fs_mesh_s();
        m->nv = nv; m->nc = nc; m->n_owned = nv; m->tdim = 3;
        int rc = FS_OK;
        if ((rc = m->xyz.alloc(nv * 4)) != FS_OK || (rc = m->cells.alloc(nc * 4)) != FS_OK || (rc = m->gid.alloc(nv)) != FS_OK) {
            delete m;
            return rc;
        }
        hipLaunchKernelGGL(k_renumbered_xyz, dim3(fs_grid_for(nv)), dim3(FS_BLOCK), 0, s, nv, v_out.p, dx.p, m->xyz.p, m->gid.p);
        hipLaunchKernelGGL(k_renumbered_cells, dim3(fs_grid_for(nc)), dim3(FS_BLOCK), 0, s, nc, c_out.p, dc.p, rank.p, m->cells.p);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            fs_set_error("fs_mesh_create_renumbered: kernel launch failed");
            delete m;
            return FS_ERR_HIP;
        }
        *mesh_out = m;
    }
    return FS_OK;
}

extern "C" int fs_mesh_locality_order(int gdim, int64_t nv, const double* xyz, int64_t nc, const int32_t* cells, int verts_per_cell,
                                      int32_t* vertex_order, int32_t* cell_order) {
    return locality_order_impl(gdim, nv, xyz, nc, cells, verts_per_cell, vertex_order, cell_order, nullptr);
}

extern "C" int fs_mesh_create_renumbered(int64_t nv, const double* xyz, int64_t nc, const int32_t* cells, int32_t* vertex_order,
                                         int32_t* cell_order, fs_mesh_t* out) {
    FS_REQUIRE(out, "fs_mesh_create_renumbered: null pointer");
    return locality_order_impl(3, nv, xyz, nc, cells, 4, vertex_order, cell_order, out);
}

void fs_symbolic_preload() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(k_diag_keys));
    (void)hipGetLastError();
}
