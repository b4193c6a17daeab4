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
#define BOX_DC_AUX 2      // (fs_box.h: dot weights and class numbers are read once - streamed past the caches)
#include "fs_box.h"
#include <functional>
#include <chrono>
#include <string>
#include <unordered_map>
#include <math.h>
#include <stdlib.h>

#ifndef FS_BLOCK_ROUND
#define FS_BLOCK_ROUND 3      // 3x3 blocks per round of the vector-space product (measured: see DESIGN.md section 3)
#endif

// ---- SELL-64 SpMV, optionally fused with the three CG dot products -------------------------
// One round of N entries of a scalar row: all 2N loads are issued before the first FMA, so the latency of a
// round is one memory round trip whatever N is.  The remainder of a row (width % UNROLL entries) goes through
// the same code with N = remainder (compile-time if-chain) instead of a serial tail loop - on the 15-wide
// rows of a P1 Kuhn mesh a serial tail is 3 of the 6 round trips at UNROLL = 4.
// NT: matrix values and column indices are read once per product; when the matrix is larger than the caches
// (Infinity Cache 256 MB) a non-temporal load keeps them from evicting the x window out of L2 (measured on MI355X,
// 10 M DOF: P1 310 -> 290 us, P2 1095 -> 1016 us; at 1 M DOF, where the matrix stays cache-resident between
// iterations, the hint costs 25 %, so it is chosen by size).
template <bool NT, typename T>
__device__ __forceinline__ T fs_ldv(const T* p) {
    return NT ? __builtin_nontemporal_load(p) : *p;
}
// op / op2: the offset lists of the two pieces of a DIA slice (fs_symbolic.hip, "SPLIT slices"; op2 == op when the slice is
// not split), hi: this lane belongs to the second piece - two scalar loads and one select per entry, no column stream
template <int N, bool NT>
__device__ __forceinline__ void dia_round(const double* __restrict__ vp, const int32_t* __restrict__ op, const int32_t* __restrict__ op2,
                                          bool hi, int k, int32_t r, int32_t cmax, const double* __restrict__ x, double& acc) {
    double v[N], xv[N];
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = fs_ldv<NT>(&vp[(int64_t)(k + u) * FS_SLICE]);
#pragma unroll
    for (int u = 0; u < N; ++u) {
        int32_t c = r + (hi ? op2[k + u] : op[k + u]);
        c = c < 0 ? 0 : (c > cmax ? cmax : c);
        xv[u] = x[c];
    }
#pragma unroll
    for (int u = 0; u < N; ++u) acc += v[u] * xv[u];
}
template <int N, bool NT>
__device__ __forceinline__ void sell_round(const double* __restrict__ vp, const int32_t* __restrict__ cp, int k,
                                           const double* __restrict__ x, double& acc) {
    int32_t c[N];
    double v[N], xv[N];
#pragma unroll
    for (int u = 0; u < N; ++u) c[u] = fs_col_decode(fs_ldv<NT>(&cp[(int64_t)(k + u) * FS_SLICE]));
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = fs_ldv<NT>(&vp[(int64_t)(k + u) * FS_SLICE]);
#pragma unroll
    for (int u = 0; u < N; ++u) xv[u] = x[c[u]];
#pragma unroll
    for (int u = 0; u < N; ++u) acc += v[u] * xv[u];
}
template <int N, bool NT>
struct row_tail {
    static __device__ __forceinline__ void dia(int rem, const double* __restrict__ vp, const int32_t* __restrict__ op,
                                               const int32_t* __restrict__ op2, bool hi, int k,
                                               int32_t r, int32_t cmax, const double* __restrict__ x, double& acc) {
        if (rem == N) dia_round<N, NT>(vp, op, op2, hi, k, r, cmax, x, acc);
        else row_tail<N - 1, NT>::dia(rem, vp, op, op2, hi, k, r, cmax, x, acc);
    }
    static __device__ __forceinline__ void sell(int rem, const double* __restrict__ vp, const int32_t* __restrict__ cp, int k,
                                                const double* __restrict__ x, double& acc) {
        if (rem == N) sell_round<N, NT>(vp, cp, k, x, acc);
        else row_tail<N - 1, NT>::sell(rem, vp, cp, k, x, acc);
    }
};
template <bool NT>
struct row_tail<0, NT> {
    static __device__ __forceinline__ void dia(int, const double*, const int32_t*, const int32_t*, bool, int, int32_t, int32_t, const double*, double&) {}
    static __device__ __forceinline__ void sell(int, const double*, const int32_t*, int, const double*, double&) {}
};

// One round of R block entries (BS x BS values each, vector spaces): all R * (BS*BS + BS) loads are issued before the first FMA,
// as the scalar rounds above do - the serial loop it replaces had the 12 loads of ONE 3x3 block in flight per lane and was
// latency-bound (fine-level product of the elasticity AMG, configs[2]).
template <int BS, int R, bool NT>
__device__ __forceinline__ void block_round(const double* __restrict__ vp, int64_t plane, const int64_t (&c)[R], int k,
                                            const double* __restrict__ x, double (&acc)[BS]) {
    double v[R][BS * BS], xv[R][BS];
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
        for (int q = 0; q < BS * BS; ++q) v[u][q] = fs_ldv<NT>(&vp[(int64_t)q * plane + (int64_t)(k + u) * FS_SLICE]);
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
        for (int j = 0; j < BS; ++j) xv[u][j] = x[c[u] * BS + j];
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
        for (int j = 0; j < BS; ++j)
#pragma unroll
            for (int i = 0; i < BS; ++i) acc[i] += v[u][i * BS + j] * xv[u][j];
}

template <int BS, int DOTS, int UNROLL, bool NT = false>
__global__ void __launch_bounds__(FS_BLOCK) k_sell_spmv(int64_t n_rows, int64_t n_cols, int64_t n_slices,
                                                        const int64_t* __restrict__ slice_ptr,
                                                        const int32_t* __restrict__ sell_col,
                                                        const int32_t* __restrict__ dia_ptr,
                                                        const int32_t* __restrict__ dia_off,
                                                        const double* __restrict__ val, int64_t plane,
                                                        const double* __restrict__ x, double* __restrict__ y,
                                                        const double* __restrict__ rvec,
                                                        double* __restrict__ partials,
                                                        int* __restrict__ status,
                                                        const int32_t* __restrict__ order,
                                                        int part_base, int part_stride, int bump) {
    // part_base / part_stride: where this launch's per-workgroup dot partials go (partials[j*stride + base + wg]);
    // a product split into an interior and a boundary launch fills one array of stride = both grids
    // DOTS == 4: no dot products, but the launch is gated by the status word like the fused ones (the product of the
    // pipelined CG, whose dots are computed by its update kernel)
    if (DOTS) {
        if (status[0] != 0) return;  // converged earlier: the remaining launches of the batch are no-ops
        // status[2] = number of in-loop products launched so far: the update kernel of a captured batch (hipGraph: same
        // arguments every iteration) reads its iteration index from it.  Nobody else touches the word while we run.
        if (bump && blockIdx.x == 0 && threadIdx.x == 0) status[2] += 1;
    }
    __shared__ double lds4[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double d_rz = 0.0, d_wz = 0.0, d_rr = 0.0;
    const int64_t n_chunks = (n_slices + 3) >> 2;
    const int32_t cmax = (int32_t)(n_cols - 1);
    for (chunk_iter it = xcd_chunks(n_chunks); it.cur < it.end; it.cur += it.step) {
        const int64_t q = __builtin_amdgcn_readfirstlane((int)(it.cur * 4 + wave));
        if (q >= n_slices) continue;
        const int64_t s = order ? __builtin_amdgcn_readfirstlane(order[q]) : q;      // fs_space_s::slice_order
        const int64_t base = slice_ptr[s];
        const int width = (int)((slice_ptr[s + 1] - base) >> 6);
        const int32_t dp = dia_ptr[s];              // wave-uniform: >= 0 selects the DIA form
        const int64_t r = s * FS_SLICE + lane;
        const bool live = r < n_rows;
        const int32_t* __restrict__ cp = sell_col + base + lane;
        const double* __restrict__ vp = val + base + lane;
        double zi[BS], ri[BS], acc[BS];
        // operands of the fused dots are requested up front so their latency hides under the row loop
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            zi[i] = 0.0;
            ri[i] = 0.0;
            acc[i] = 0.0;
            if (DOTS && DOTS != 4 && live) {
                if (DOTS == 1 || DOTS == 3) zi[i] = x[r * BS + i];
                ri[i] = rvec[r * BS + i];
            }
        }
        if (dp >= 0) {
            // DIA slice: column = row + offset[k] (one scalar per entry row), so the x gather of the wave
            // is one contiguous 512-B read and no column index is streamed.  Entries a row does not have
            // hold the value 0 and read a clamped, valid address.
            const int split = dia_off[dp];                                  // rows [split, 64) use the second offset list
            const int32_t* __restrict__ op = dia_off + dp + 1;
            const int32_t* __restrict__ op2 = op + (split < FS_SLICE ? width : 0);
            const bool hi = lane >= split;
            int k = 0;
            if (BS == 1) {
                for (; k + UNROLL <= width; k += UNROLL) dia_round<UNROLL, NT>(vp, op, op2, hi, k, (int32_t)r, cmax, x, acc[0]);
                row_tail<UNROLL - 1, NT>::dia(width - k, vp, op, op2, hi, k, (int32_t)r, cmax, x, acc[0]);
                k = width;
            }
            if (BS > 1) {
                constexpr int R = BS == 3 ? FS_BLOCK_ROUND : 2;
                for (; k + R <= width; k += R) {
                    int64_t c[R];
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        c[u] = r + (hi ? op2[k + u] : op[k + u]);
                        c[u] = c[u] < 0 ? 0 : (c[u] > cmax ? cmax : c[u]);
                    }
                    block_round<BS, R, NT>(vp, plane, c, k, x, acc);
                }
            }
            for (; k < width; ++k) {
                int64_t c = r + (hi ? op2[k] : op[k]);
                c = c < 0 ? 0 : (c > cmax ? cmax : c);
#pragma unroll
                for (int j = 0; j < BS; ++j) {
                    const double xj = x[c * BS + j];
#pragma unroll
                    for (int i = 0; i < BS; ++i) acc[i] += fs_ldv<NT>(&vp[(int64_t)(i * BS + j) * plane + (int64_t)k * FS_SLICE]) * xj;
                }
            }
        } else {
            int k = 0;
            if (BS == 1) {
                for (; k + UNROLL <= width; k += UNROLL) sell_round<UNROLL, NT>(vp, cp, k, x, acc[0]);
                row_tail<UNROLL - 1, NT>::sell(width - k, vp, cp, k, x, acc[0]);
                k = width;
            }
            if (BS > 1) {
                constexpr int R = BS == 3 ? FS_BLOCK_ROUND : 2;
                for (; k + R <= width; k += R) {
                    int64_t c[R];
#pragma unroll
                    for (int u = 0; u < R; ++u) c[u] = fs_col_decode(fs_ldv<NT>(&cp[(int64_t)(k + u) * FS_SLICE]));
                    block_round<BS, R, NT>(vp, plane, c, k, x, acc);
                }
            }
            for (; k < width; ++k) {
                const int64_t c = fs_col_decode(cp[(int64_t)k * FS_SLICE]);
#pragma unroll
                for (int j = 0; j < BS; ++j) {
                    const double xj = x[c * BS + j];
#pragma unroll
                    for (int i = 0; i < BS; ++i) acc[i] += fs_ldv<NT>(&vp[(int64_t)(i * BS + j) * plane + (int64_t)k * FS_SLICE]) * xj;
                }
            }
        }
        if (live) {
#pragma unroll
            for (int i = 0; i < BS; ++i) {
                y[r * BS + i] = acc[i];
                if (DOTS == 1) {          // CG: r.z, w.z, r.r
                    d_rz += ri[i] * zi[i];
                    d_wz += acc[i] * zi[i];
                    d_rr += ri[i] * ri[i];
                } else if (DOTS == 2) {   // generic: w.u, w.w, u.u with u = rvec
                    d_rz += acc[i] * ri[i];
                    d_wz += acc[i] * acc[i];
                    d_rr += ri[i] * ri[i];
                } else if (DOTS == 3) {   // diagonally scaled CG (z == r): r.r, w.r, sum d r^2 with d = rvec
                    d_rz += zi[i] * zi[i];
                    d_wz += acc[i] * zi[i];
                    d_rr += ri[i] * zi[i] * zi[i];
                }
            }
        }
    }
    if (DOTS && DOTS != 4) {
        const double t0 = fs_block_sum(d_rz, lds4);
        const double t1 = fs_block_sum(d_wz, lds4);
        const double t2 = fs_block_sum(d_rr, lds4);
        if (threadIdx.x == 0) {
            partials[part_base + blockIdx.x] = t0;
            partials[part_stride + part_base + blockIdx.x] = t1;
            partials[2 * part_stride + part_base + blockIdx.x] = t2;
        }
    }
}

// ---- DIA slices, two rows per lane ------------------------------------------------------------------------------------
// PMC on the one-row-per-lane kernel at 10 M DOF (round 2: TA busy 70 %, 31 % of the wave cycles issue stalls, time
// proportional to the L1 accesses and insensitive to the HBM byte count) says the per-CU address / L1 path is the limit,
// not HBM.  This kernel halves the vector-memory instructions per row: a wave takes TWO slices with the same offset list
// (lanes 0-31 the first, 32-63 the second), every lane two consecutive rows, so the value planes are read as 16-byte
// lane loads, and inside a run of consecutive offsets (..., o, o+1, ...) the x value a lane needs for its second row at
// offset o is the one it needs for its first row at offset o+1: one new 8-byte load per further offset of a run.
// P1 Kuhn mesh (15 offsets in 7 runs): 15 + 22 + 2 loads per two rows instead of 64.
// Pairs are formed on the host (fs_space_s::pair_list: consecutive slices in processing order, both DIA, both complete,
// identical offset lists); everything else goes through k_sell_spmv as before.  Same per-row summation order.
template <int DOTS, bool NT>
__global__ void __launch_bounds__(FS_BLOCK) k_dia_pair_spmv(int64_t n_cols, int64_t n_pairs, const int32_t* __restrict__ pairs,
                                                            const int64_t* __restrict__ slice_ptr,
                                                            const int32_t* __restrict__ dia_ptr,
                                                            const int32_t* __restrict__ dia_off,
                                                            const double* __restrict__ val,
                                                            const double* __restrict__ x, double* __restrict__ y,
                                                            const double* __restrict__ rvec,
                                                            double* __restrict__ partials,
                                                            int* __restrict__ status, int part_base, int part_stride, int bump) {
    if (DOTS) {
        if (status[0] != 0) return;
        if (bump && blockIdx.x == 0 && threadIdx.x == 0) status[2] += 1;      // see k_sell_spmv
    }
    typedef double v2d __attribute__((ext_vector_type(2)));
    __shared__ double lds4[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l2 = (lane & 31) * 2;
    double d_rz = 0.0, d_wz = 0.0, d_rr = 0.0;
    const int64_t n_chunks = (n_pairs + 3) >> 2;
    const int32_t cmax = (int32_t)(n_cols - 1);
    constexpr int U = 16;
    for (chunk_iter it = xcd_chunks(n_chunks); it.cur < it.end; it.cur += it.step) {
        const int64_t q = __builtin_amdgcn_readfirstlane((int)(it.cur * 4 + wave));
        if (q >= n_pairs) continue;
        const int32_t sa = __builtin_amdgcn_readfirstlane(pairs[2 * q]), sb = __builtin_amdgcn_readfirstlane(pairs[2 * q + 1]);
        const int64_t s = half ? sb : sa;
        const int64_t base = slice_ptr[s];
        const int width = (int)((slice_ptr[sa + 1] - slice_ptr[sa]) >> 6);       // same for both slices of a pair
        const int32_t* __restrict__ op = dia_off + dia_ptr[sa] + 1;               // the shared offset list (pairs are never split slices)
        const int32_t r = (int32_t)(s * FS_SLICE + l2);
        const double* __restrict__ vp = val + base + l2;
        v2d zi = {0.0, 0.0}, ri = {0.0, 0.0};
        if (DOTS && DOTS != 4) {
            if (DOTS == 1 || DOTS == 3) zi = *reinterpret_cast<const v2d*>(&x[r]);
            ri = *reinterpret_cast<const v2d*>(&rvec[r]);
        }
        double a0 = 0.0, a1 = 0.0, prev_hi = 0.0;
        int32_t prev_o = INT32_MIN;
        for (int k0 = 0; k0 < width; k0 += U) {
            v2d t[U];
            double lo[U], hi[U];
            bool cont[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k0 + u < width) t[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(&vp[(int64_t)(k0 + u) * FS_SLICE]))
                                              : *reinterpret_cast<const v2d*>(&vp[(int64_t)(k0 + u) * FS_SLICE]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (k0 + u < width) {
                    const int32_t o = op[k0 + u];
                    cont[u] = o == (u == 0 ? prev_o : op[k0 + u - 1]) + 1;     // wave-uniform
                    int32_t c1 = r + o + 1;
                    c1 = c1 < 0 ? 0 : (c1 > cmax ? cmax : c1);
                    hi[u] = x[c1];
                    if (!cont[u]) {
                        int32_t c0 = r + o;
                        c0 = c0 < 0 ? 0 : (c0 > cmax ? cmax : c0);
                        lo[u] = x[c0];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (k0 + u < width) {
                    const double l = cont[u] ? (u == 0 ? prev_hi : hi[u - 1]) : lo[u];
                    a0 += t[u].x * l;
                    a1 += t[u].y * hi[u];
                }
            }
            const int last = (width - k0 < U ? width - k0 : U) - 1;
            prev_o = op[k0 + last];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (u == last) prev_hi = hi[u];
        }
        v2d out;
        out.x = a0; out.y = a1;
        *reinterpret_cast<v2d*>(&y[r]) = out;
        if (DOTS == 1) {
            d_rz += ri.x * zi.x + ri.y * zi.y;
            d_wz += a0 * zi.x + a1 * zi.y;
            d_rr += ri.x * ri.x + ri.y * ri.y;
        } else if (DOTS == 2) {
            d_rz += a0 * ri.x + a1 * ri.y;
            d_wz += a0 * a0 + a1 * a1;
            d_rr += ri.x * ri.x + ri.y * ri.y;
        } else if (DOTS == 3) {
            d_rz += zi.x * zi.x + zi.y * zi.y;
            d_wz += a0 * zi.x + a1 * zi.y;
            d_rr += ri.x * zi.x * zi.x + ri.y * zi.y * zi.y;
        }
    }
    if (DOTS && DOTS != 4) {
        const double t0 = fs_block_sum(d_rz, lds4);
        const double t1 = fs_block_sum(d_wz, lds4);
        const double t2 = fs_block_sum(d_rr, lds4);
        if (threadIdx.x == 0) {
            partials[part_base + blockIdx.x] = t0;
            partials[part_stride + part_base + blockIdx.x] = t1;
            partials[2 * part_stride + part_base + blockIdx.x] = t2;
        }
    }
}


// ---- row-dictionary form of a scalar DIA matrix ---------------------------------------------------------------------------------
// On a uniform box mesh with constant coefficients (BASELINE configs[0] / [1] / [3]: BoxMesh, k = 20) the assembled operator has a
// few dozen (P1) to a few hundred (CG2) DISTINCT rows - interior, the 26 kinds of boundary position, the rows next to Dirichlet
// faces - each repeated bit for bit (the box assembly snaps its edge vectors to the grid spacing, fs_assemble.hip, so that rows are
// translation-invariant).  The product then does not have to stream 8 B per entry: every row carries a 2-byte class number and
// the distinct coefficient rows are fetched per work item.  Built per solve from the values the solver is about to multiply with,
// every row verified bit for bit against its class, so it is lossless and needs no knowledge of where the matrix came from.
//
// Round 4: the product is organised by ROWS, not by the 64-row slices of the value storage it no longer reads.
//   * SEGMENTS (once per space, dict_structure_build): maximal runs of consecutive rows whose (col - row) offset sets are nested in
//     one list - on a box mesh a mesh line (CG2: 107 / 108 rows, every line its own list because the edge classes are numbered with
//     different line lengths) or, where all lines share one list (P1), the whole mesh.  Cut into WORK ITEMS of <= 128 rows.
//   * the segment's offset list is cut into RUNS of up to three consecutive offsets (o, o + 1, o + 2).  A lane holds TWO consecutive
//     rows; for a run starting at o its rows need x[r + o .. r + o + 3]: ONE 16-byte load x[r + o], x[r + o + 1] per lane, the
//     other two values are the NEXT lane's load (DPP wave shift, no LDS); lane 63 has no rows of its own - an item is 126 rows -
//     and loads what lane 62 needs (tests/test_gpu_kernels.py multiplies chains of dependent vectors bit for bit against the
//     streaming product).  3.5 vector-memory instructions per row on the Kuhn stencil
//     instead of the 19 of round 3, whose per-CU address path - not HBM - was the limit (0.27 of the peak on 26 B/row).
//   * a RUN PLAN per segment: rounds of 8 runs (start offset, length); slot 0 of round 0 is the run (0; no coefficients) whose
//     load IS z = x[r], x[r + 1] for the fused dots.  A class row holds its coefficients IN PLAN LAYOUT, [round][run][3], zero where
//     a run is shorter or the row has no such entry: the kernel reads them at fixed positions.  Ascending offsets = the storage
//     order of the streaming kernels, one fma each: same summation order, same bits (the extra terms add +0 * x).
//   * per item, the distinct classes of its rows (a mesh line: interior + the two ends) are copied into the wave's own LDS
//     region (sized for the item with the most classes, counted when the classes are found; at most 64 KB per workgroup) - the
//     dictionary itself may have any size (CG2: 350 KB).
//   * items whose loads could leave [0, n_cols) (first / last mesh plane) gather their four values per run one by one, clamped.
// Measured (tools/probes/dict_pair_probe.hip and profiles/r04_*): P1, 10 M rows: 123 -> 64 us.
constexpr int FS_DICT_CAP = 8192;       // hash slots
constexpr int FS_DICT_MAX = 4096;       // distinct rows accepted
constexpr int FS_DICT_ITEM_ROWS = 126;  // rows per work item: two per lane for lanes 0 .. 62; lane 63 only loads (its pair is what lane 62
                                        // needs from `the next lane`: no separate tail loads)
constexpr int FS_DICT_WHOLE_LDS_BYTES = 32 << 10;   // a dictionary up to this size is held whole by every workgroup
constexpr int FS_DICT_LDS_BYTES = 64 << 10;   // per workgroup: 4 waves x (most distinct classes of any item) x (doubles per class row)
constexpr int FS_DICT_ITEMS_PER_WAVE = 4;   // consecutive items a wave takes when it fetches class rows per item
constexpr int FS_DICT3_RUNS = 4;          // runs whose loads the block-row kernel has in flight at a time
constexpr int FS_DICT_MAX_ROUNDS = 8;   // rounds of 8 runs per plan (192 coefficient positions)

struct dict_plan_round {
    int32_t start[8];       // first offset of each run (0 for an empty slot: a harmless load of x[r], x[r + 1])
    uint8_t len[8];         // 0 .. 3
    uint8_t pad[24];
};
static_assert(sizeof(dict_plan_round) == 64, "one 64-byte scalar load per round");
// Rounds of TWELVE runs (fs_space_s::dict_runs = 12: the lattice-ordered shadow of a CG2 box space, fs_lattice.hip - three quarters
// of its mesh lines have 11 runs + the z run, one round instead of two): the same 64 bytes, 12 starts + 12 lengths.  Code that
// takes the number of runs per round at run time reads a round through these two:
struct dict_plan_round12 {
    int32_t start[12];
    uint8_t len[12];
    uint8_t pad[4];
};
static_assert(sizeof(dict_plan_round12) == 64, "one 64-byte scalar load per round");
__host__ __device__ __forceinline__ int32_t dict_run_start(const dict_plan_round* __restrict__ pl, int NR, int g) {
    return reinterpret_cast<const int32_t*>(pl + g / NR)[g % NR];
}
__host__ __device__ __forceinline__ int dict_run_len(const dict_plan_round* __restrict__ pl, int NR, int g) {
    return (int)(reinterpret_cast<const uint8_t*>(pl + g / NR) + 4 * NR)[g % NR];
}

// Coefficient position (plan layout) of the stored entry with offset o, walking the runs of a plan in ascending order from run g on
// (the entries of a row come in ascending offsets, as the runs do): -1 = the plan has no such offset.
__device__ __forceinline__ int dict_slot_of(const dict_plan_round* __restrict__ pl, int n_runs, int RL, int& g, int32_t o, int NR = 8) {
    while (g < n_runs) {
        const int32_t st = dict_run_start(pl, NR, g);
        const int ln = dict_run_len(pl, NR, g);
        if (ln > 0 && o < st + ln) return o >= st ? RL * g + (o - st) : -1;
        ++g;
    }
    return -1;
}

// One row's stored entries (DIA slice storage: value plane k, offset list of the row's piece of its slice), nonzero values only:
// f(position, value), position = slot * nq + q for component q of an entry's nq = bs * bs values (block entry e, component q at
// q * plane + e).  Returns false when an entry has no position in the plan.
// sc (scalar operators only; may be null): the row is walked as D^-1/2 A D^-1/2 - every stored value v at column r + o becomes
// (v sc[r]) sc[r + o], the expression and the bits of k_scale_copy - so that a matrix can be compared with the kept class table of
// its scaled form WITHOUT writing the scaled copy first (the copy is 300 MB of traffic at 1 M rows, and on the row-dictionary path
// nobody reads it).
template <typename F>
__device__ __forceinline__ bool dict_walk_row(int32_t r, const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ dia_ptr,
                                              const int32_t* __restrict__ dia_off, const double* __restrict__ val, int nq, int64_t plane,
                                              const dict_plan_round* __restrict__ pl, int n_runs, int RL, F f,
                                              const double* __restrict__ sc = nullptr, int NR = 8) {
    const int32_t sl = r >> 6, ln = r & 63;
    const int64_t base = slice_ptr[sl];
    const int width = (int)((slice_ptr[sl + 1] - base) >> 6);
    const int32_t dp = dia_ptr[sl];
    const int32_t* __restrict__ op = dia_off + dp + 1 + (ln >= dia_off[dp] ? width : 0);
    const double* __restrict__ vp = val + base + ln;
    bool ok = true;
    if (n_runs == 8 && NR == 8 && nq == 1) {
        // one-round plans of scalar operators (P1): the eight run starts and lengths are wave-uniform - in scalar registers, an
        // entry's position is found by comparing its offset with all of them - and the row is taken eight entries at a time,
        // offsets and values of a batch (and whatever f loads) in flight together; the walk below goes entry by entry, two loads
        // and a branch per step of the plan, each waited for
        int32_t st[8], le[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            st[j] = __builtin_amdgcn_readfirstlane(pl->start[j]);
            le[j] = __builtin_amdgcn_readfirstlane((int)pl->len[j]);
        }
        for (int k0 = 0; k0 < width; k0 += 8) {
            int32_t o[8];
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u < width ? k0 + u : width - 1;        // (clamped: the loads of a short batch stay inside the row)
                o[u] = op[k];
                v[u] = vp[(int64_t)k * FS_SLICE];
            }
            if (sc) {
                const double sr = sc[r];
                double scol[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) scol[u] = sc[v[u] != 0.0 ? r + o[u] : r];       // (a padded position has no column)
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = v[u] * sr * scol[u];
            }
            int slot[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                slot[u] = -1;
#pragma unroll
                for (int j = 1; j < 8; ++j)              // (run 0 is the z run: no coefficients)
                    if (le[j] > 0 && o[u] >= st[j] && o[u] < st[j] + le[j]) slot[u] = RL * j + (o[u] - st[j]);
                if (k0 + u >= width || v[u] == 0.0) slot[u] = -3;         // nothing stored there
                if (slot[u] == -1) ok = false;                           // a value outside the plan
            }
            if (!ok) return false;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (slot[u] >= 0) f(slot[u], v[u]);
        }
        return true;
    }
    int g = 1;                  // run 0 is the z run
    if (nq == 1) {
        // scalar operators with plans of several rounds (CG2; the lattice order's line plans): offsets, values and scale factors of
        // eight entries in flight together, then the walk along the plan for the eight - the entry-by-entry loop below waits for two
        // or three dependent loads per entry (configs[3]: 3.9 ms per solve for the comparison with the kept table)
        for (int k0 = 0; k0 < width; k0 += 8) {
            int32_t o[8];
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u < width ? k0 + u : width - 1;
                o[u] = op[k];
                v[u] = vp[(int64_t)k * FS_SLICE];
            }
            bool stored[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) stored[u] = k0 + u < width && v[u] != 0.0;
            if (sc) {
                const double sr = sc[r];
                double scol[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) scol[u] = sc[stored[u] ? r + o[u] : r];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = v[u] * sr * scol[u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (stored[u]) {
                    const int slot = dict_slot_of(pl, n_runs, RL, g, o[u], NR);
                    if (slot < 0) return false;
                    f(slot, v[u]);
                }
        }
        return true;
    }
    for (int k = 0; k < width && ok; ++k) {
        int slot = -2;          // not looked up yet
        for (int q = 0; q < nq; ++q) {
            double v = vp[(int64_t)k * FS_SLICE + (int64_t)q * plane];
            if (v == 0.0) continue;
            if (sc) v = v * sc[r] * sc[r + op[k]];
            if (slot == -2) slot = dict_slot_of(pl, n_runs, RL, g, op[k], NR);
            if (slot < 0) { ok = false; break; }
            f(slot * nq + q, v);
        }
    }
    return ok;
}

__device__ __forceinline__ unsigned long long dict_mix(unsigned long long h, int slot, double v) {
    h = (h ^ (unsigned long long)__double_as_longlong(v)) * 1099511628211ull;
    h = (h ^ (unsigned long long)(unsigned)slot) * 1099511628211ull;
    return h ^ (h >> 29);
}

// info[0] = classes found, info[1] = 1: gave up (too many classes), info[2] = rows that differ from their class or do not fit their
// plan (verification), info[3] = most distinct classes in any item
// Nearly all rows of such an operator carry the SAME hash, so the table is hit where it hurts: the lanes of a wave are first grouped
// by hash (a wave of interior rows is one group) and only the group's first lane goes to memory, and it looks at the slot with an
// ordinary cached load before any atomic (a slot goes from 0 to its final key once: a key seen there is final, a stale 0 merely
// sends the lane to the compare-and-swap, which returns the truth).
__global__ void __launch_bounds__(FS_BLOCK) k_dict_insert(int64_t n_items, const int4* __restrict__ items, const dict_plan_round* __restrict__ plans,
                                                          const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ dia_ptr,
                                                          const int32_t* __restrict__ dia_off, const double* __restrict__ val, int nq, int64_t plane, int S, int RL,
                                                          unsigned long long* keys, const unsigned long long* keys_cached, double* slot_vals,
                                                          uint16_t* __restrict__ cls_slot, int* info, int NR = 8) {
    const int lane = threadIdx.x & 63;
    int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; q < n_items; q += stride) {
        if (__hip_atomic_load(&info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        const int4 it = items[q];
        const int32_t first = it.x, nr = it.y & 0xffff;
        const dict_plan_round* __restrict__ pl = plans + it.z;
        const int n_runs = NR * it.w;
        for (int half = 0; half < 2; ++half) {
            const int i = half * 64 + lane;
            const bool live = i < nr;
            const int32_t r = first + i;
            unsigned long long h = 1469598103934665603ull;
            bool fits = true;
            if (live) fits = dict_walk_row(r, slice_ptr, dia_ptr, dia_off, val, nq, plane, pl, n_runs, RL, [&](int slot, double v) { h = dict_mix(h, slot, v); }, nullptr, NR);
            if (!h) h = 1ull;
            if (live && !fits) atomicAdd(&info[2], 1);
            int my_slot = 0;
            unsigned long long todo = __ballot(live);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const unsigned long long hl = ((unsigned long long)(unsigned)__shfl((int)(h >> 32), leader, 64) << 32) |
                                              (unsigned long long)(unsigned)__shfl((int)(h & 0xffffffffull), leader, 64);
                const bool mine = live && h == hl;
                int slot = (int)(hl & (FS_DICT_CAP - 1));
                if (lane == leader) {
                    for (int probe = 0; probe < FS_DICT_CAP; ++probe) {
                        unsigned long long old = keys_cached[slot];
                        if (old != hl) {
                            old = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (old == 0ull) old = atomicCAS(&keys[slot], 0ull, hl);
                            if (old == 0ull) {                  // this row is the representative of a new class
                                if (atomicAdd(&info[0], 1) >= FS_DICT_MAX) __hip_atomic_store(&info[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                double* __restrict__ dst = slot_vals + (int64_t)slot * S;       // (zero-filled by the host before the launch)
                                (void)dict_walk_row(r, slice_ptr, dia_ptr, dia_off, val, nq, plane, pl, n_runs, RL, [&](int sl2, double v) { if (sl2 < S) dst[sl2] = v; }, nullptr, NR);
                                break;
                            }
                        }
                        if (old == hl) break;
                        slot = (slot + 1) & (FS_DICT_CAP - 1);
                    }
                }
                slot = __shfl(slot, leader, 64);
                if (mine) my_slot = slot;
                todo &= ~__ballot(mine);
            }
            if (live) cls_slot[r] = (uint16_t)my_slot;
        }
    }
}

// number the occupied slots; values[id][S] = the class rows in plan layout, nnz[id] = their nonzero positions
__global__ void __launch_bounds__(1024) k_dict_compact(const unsigned long long* __restrict__ keys, const double* __restrict__ slot_vals,
                                                       int S, int32_t* __restrict__ slot2cls, double* __restrict__ values, int32_t* __restrict__ nnz) {
    constexpr int PER = FS_DICT_CAP / 1024;       // consecutive slots per thread
    __shared__ int cnt[1024];
    const int t = threadIdx.x;
    int mine = 0;
    for (int q = 0; q < PER; ++q) mine += keys[t * PER + q] != 0ull;
    cnt[t] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {       // inclusive scan
        const int v = t >= off ? cnt[t - off] : 0;
        __syncthreads();
        cnt[t] += v;
        __syncthreads();
    }
    int id = cnt[t] - mine;
    for (int q = 0; q < PER; ++q) {
        const int slot = t * PER + q;
        const bool used = keys[slot] != 0ull;
        slot2cls[slot] = used ? id : -1;
        if (used && id < FS_DICT_MAX) {
            int nz = 0;
            for (int k = 0; k < S; ++k) {
                const double v = slot_vals[(int64_t)slot * S + k];
                values[(int64_t)id * S + k] = v;
                nz += v != 0.0;
            }
            nnz[id] = nz;
        }
        id += used;
    }
}

// class numbers; EVERY row against its class, bit for bit (a hash collision ends here); the distinct classes of every item counted
__global__ void __launch_bounds__(FS_BLOCK) k_dict_finish(int64_t n_items, const int4* __restrict__ items, const dict_plan_round* __restrict__ plans,
                                                          const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ dia_ptr,
                                                          const int32_t* __restrict__ dia_off, const double* __restrict__ val, int nq, int64_t plane, int S, int RL,
                                                          const int32_t* __restrict__ slot2cls, const double* __restrict__ values,
                                                          const int32_t* __restrict__ nnz, const uint16_t* __restrict__ cls_slot,
                                                          uint16_t* __restrict__ cls, int* info, const double* __restrict__ sc = nullptr, int NR = 8) {
    const int lane = threadIdx.x & 63;
    int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    int bad = 0, crowded = 0;       // (crowded: most distinct classes of an item so far)
    for (; q < n_items; q += stride) {
        const int4 it = items[q];
        const int32_t first = it.x, nr = it.y & 0xffff;
        const dict_plan_round* __restrict__ pl = plans + it.z;
        const int n_runs = NR * it.w;
        int id2[2] = {-1, -1};
        for (int half = 0; half < 2; ++half) {
            const int i = half * 64 + lane;
            if (i >= nr) continue;
            const int32_t r = first + i;
            const int id = slot2cls[cls_slot[r]];
            cls[r] = (uint16_t)(id < 0 ? 0 : id);
            if (id < 0 || id >= FS_DICT_MAX) { ++bad; continue; }
            id2[half] = id;
            const double* __restrict__ dv = values + (int64_t)id * S;
            int nz = 0, diff = 0;
            const bool fits = dict_walk_row(r, slice_ptr, dia_ptr, dia_off, val, nq, plane, pl, n_runs, RL, [&](int slot, double v) {
                ++nz;
                diff += slot >= S || __double_as_longlong(v) != __double_as_longlong(dv[slot < S ? slot : 0]);
            }, sc, NR);
            bad += !fits || diff != 0 || nz != nnz[id];
        }
        // distinct classes among the item's rows (what its wave will have to hold in LDS)
        int distinct = 0;
        unsigned long long m0 = __ballot(id2[0] >= 0), m1 = __ballot(id2[1] >= 0);
        while (m0 | m1) {
            const bool from0 = m0 != 0ull;
            const int src = __ffsll((long long)(from0 ? m0 : m1)) - 1;
            const int cv = __shfl(from0 ? id2[0] : id2[1], src, 64);
            m0 &= ~__ballot(id2[0] == cv);
            m1 &= ~__ballot(id2[1] == cv);
            ++distinct;
        }
        crowded = distinct > crowded ? distinct : crowded;
    }
    if (bad) atomicAdd(&info[2], bad);
    // (a look before the atomic: nearly every wave holds the same maximum, and 8 000 atomics on one address were most of this
    // kernel's time at 1 M rows)
    if (crowded && lane == 0 && crowded > __hip_atomic_load(&info[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&info[3], crowded);
}

// the next lane's value (DPP wave shift, no LDS); lane 63, which has no next lane, owns no rows
__device__ __forceinline__ double fs_from_next_lane(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// n doubles (n even, both 16-byte aligned) global -> LDS by the whole workgroup: batches of four 16-byte loads per thread, all four
// in flight before the first LDS store (the plain loop `for i: lds[i] = g[i]` compiles to one load - wait - store per trip:
// eight dependent round trips for the 15 KB dictionary of a P1 box, about 4 us at the head of every launch)
__device__ __forceinline__ void fs_fill_lds(double* __restrict__ lds, const double* __restrict__ g, int n) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const int n2 = n >> 1;
    const v2d* __restrict__ g2 = reinterpret_cast<const v2d*>(g);
    v2d* __restrict__ l2 = reinterpret_cast<v2d*>(lds);
    for (int base = 0; base < n2; base += 4 * FS_BLOCK) {
        v2d t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * FS_BLOCK + (int)threadIdx.x;
            t[u] = g2[i < n2 ? i : n2 - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * FS_BLOCK + (int)threadIdx.x;
            if (i < n2) l2[i] = t[u];
        }
    }
}

// the same by ONE wave (a class row of n2 16-byte pairs into the wave's LDS region): four loads in flight per lane and batch
__device__ __forceinline__ void fs_wave_copy_pairs(double* __restrict__ lds, const double* __restrict__ g, int n2, int lane) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d* __restrict__ g2 = reinterpret_cast<const v2d*>(g);
    v2d* __restrict__ l2 = reinterpret_cast<v2d*>(lds);
    for (int base = 0; base < n2; base += 256) {
        v2d t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * 64 + lane;
            t[u] = g2[i < n2 ? i : n2 - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * 64 + lane;
            if (i < n2) l2[i] = t[u];
        }
    }
}

// item: x = first row, y = rows (1 .. 126) | edge << 16, z = first plan round, w = rounds.  S = doubles per class row (8 RL per round
// of the longest plan of the space), C = class rows a wave's LDS region holds.  Dynamic LDS: 4 waves x C x S doubles.
// LDSD: the whole dictionary fits the workgroup's LDS (P1: 78 class rows of 24 doubles) - loaded once per workgroup, a row's
// coefficients sit at class * S; otherwise (CG2: 361 rows of 120) each item's classes are copied into its wave's region.
// RL: longest run of the space's plans (coefficient positions per run): 3 on P1 Kuhn meshes (runs of 2, 2, 2, 3, 2, 2, 2), 2 on CG2
// spaces, where 67 % of the runs are one offset long, 32 % two and 0.5 % three (those are cut in two): a third fewer fmas and LDS
// reads on padded positions, one DPP shift per run instead of two.
// NR: runs per plan round (8; 12 on the lattice-ordered shadow of a CG2 box space: twelve 16-byte loads in flight per lane and round)
template <int DOTS, bool LDSD, int RL, int NR = 8>
__global__ void __launch_bounds__(FS_BLOCK) k_dict_spmv(int64_t n_cols, int64_t n_items, const int4* __restrict__ items,
                                                        const dict_plan_round* __restrict__ plans, const uint16_t* __restrict__ cls,
                                                        const double* __restrict__ dict, int S, int C,
                                                        const double* __restrict__ x, double* __restrict__ y,
                                                        const double* __restrict__ rvec, double* __restrict__ partials,
                                                        int* __restrict__ status, int part_base, int part_stride, int bump, int map_xcd) {
    // (the status word is asked for now and looked at after the dictionary's loads have gone out: one round trip instead of two)
    const int st0 = DOTS ? status[0] : 0;
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef double v2du __attribute__((ext_vector_type(2), aligned(8)));
    extern __shared__ __attribute__((aligned(16))) double sdict[];
    __shared__ double lds4[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double* __restrict__ wl = LDSD ? sdict : sdict + (int64_t)wave * C * S;
    if (LDSD) fs_fill_lds(sdict, dict, C * S);      // (C = number of classes here)
    if (DOTS) {
        if (st0 != 0) return;
        if (bump && blockIdx.x == 0 && threadIdx.x == 0) status[2] += 1;      // see k_sell_spmv
    }
    if (LDSD) __syncthreads();
    double d_rz = 0.0, d_wz = 0.0, d_rr = 0.0;
    // A wave takes K CONSECUTIVE items of a chunk of 4 K: with per-item class rows (!LDSD) neighbouring mesh lines hold the same
    // classes, and a class row already in the wave's region (tagv: lane k knows which class slot k holds) is not fetched again.
    constexpr int K = LDSD ? 1 : FS_DICT_ITEMS_PER_WAVE;
    const int64_t n_chunks = (n_items + 4 * K - 1) / (4 * K);
    const int32_t cmax = (int32_t)(n_cols - 1);
    chunk_iter it = xcd_chunks(n_chunks);
    if (!map_xcd) { it.cur = blockIdx.x; it.step = gridDim.x; it.end = n_chunks; }
    int tagv = -1, rr = 0;
    const unsigned long long cmask = C >= 64 ? ~0ull : ((1ull << C) - 1ull);
    for (; it.cur < it.end; it.cur += it.step)
    for (int kk = 0; kk < K; ++kk) {
        const int64_t q = (it.cur * 4 + wave) * K + kk;
        if (q >= n_items) break;
        const int4 ds = items[__builtin_amdgcn_readfirstlane((int)q)];
        const int32_t first = __builtin_amdgcn_readfirstlane(ds.x);
        const int nr = __builtin_amdgcn_readfirstlane(ds.y) & 0xffff;
        const int edge = __builtin_amdgcn_readfirstlane(ds.y) >> 16;
        const dict_plan_round* __restrict__ pl = plans + __builtin_amdgcn_readfirstlane(ds.z);
        const int rounds = __builtin_amdgcn_readfirstlane(ds.w);
        const int32_t r = first + 2 * lane;
        const bool ok0 = 2 * lane < nr, ok1 = 2 * lane + 1 < nr;
        v2d ri = {0.0, 0.0};
        if (DOTS && DOTS != 4) {
            if (ok1) ri = *reinterpret_cast<const v2du*>(&rvec[r]);
            else if (ok0) ri.x = rvec[r];
        }
        int c0 = -1, c1 = -1;
        if ((first & 1) == 0) {        // (wave-uniform) the two class numbers of a lane in one 4-byte load (the array is padded)
            const uint32_t two = *reinterpret_cast<const uint32_t*>(&cls[r]);
            if (ok0) c0 = (int)(two & 0xffffu);
            if (ok1) c1 = (int)(two >> 16);
        } else {
            if (ok0) c0 = (int)cls[r];
            if (ok1) c1 = (int)cls[r + 1];
        }
        // the x values of the first round are asked for before anything waits for the class numbers.  Items whose accesses could leave
        // [0, n_cols) (first / last rows of the vector) load the two values of a pair one by one, each clamped into x: a column
        // outside the vector has no entry, hence a zero coefficient, and every value inside it is the right one.
        const double* __restrict__ xr = x + r;
        v2d A[NR];
        struct starts8 { int32_t v[NR]; };
        auto read_starts = [&](const dict_plan_round* __restrict__ p) {       // (wave-uniform: one 32-byte scalar load)
            starts8 t;
#pragma unroll
            for (int j = 0; j < NR; ++j) t.v[j] = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int32_t*>(p)[j]);
            return t;
        };
        auto load_round = [&](v2d (&buf)[NR], const starts8& st) {
            if (!edge) {
#pragma unroll
                for (int j = 0; j < NR; ++j) buf[j] = *reinterpret_cast<const v2du*>(xr + st.v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int32_t c = r + st.v[j];
                    buf[j].x = x[c < 0 ? 0 : (c > cmax ? cmax : c)];
                    buf[j].y = x[c + 1 < 0 ? 0 : (c + 1 > cmax ? cmax : c + 1)];
                }
            }
        };
        load_round(A, read_starts(pl));
        // (per-item class rows = CG2: the run starts of a plan are the line's own - 20 MB of plans streamed once per product - and
        // the NEXT round's are asked for a round ahead, so that a round's loads wait for one memory round trip, not two: 196 ->
        // 187 us.  It costs 20 VGPRs (the compiler forms the next round's addresses early): not done where the dictionary sits whole
        // in LDS - P1, one round -, which it took from 6 to 4 waves per SIMD, 68 -> 78 us)
        starts8 st_next = read_starts(pl + ((!LDSD && rounds > 1) ? 1 : 0));
        int b0 = 0, b1 = 0;
        if (LDSD) {
            b0 = ok0 ? c0 * S : 0;
            b1 = ok1 ? c1 * S : 0;
        } else {
            // ---- the classes of this item's rows -> the wave's LDS region (C slots), unless they are there already ----
            unsigned long long inuse = 0ull;
            bool copied = false;
            unsigned long long m0 = __ballot(ok0), m1 = __ballot(ok1);
            while (m0 | m1) {
                const bool from0 = m0 != 0ull;
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)(from0 ? m0 : m1)) - 1);
                const int cv = __builtin_amdgcn_readlane(from0 ? c0 : c1, src);
                const unsigned long long hit = __ballot(tagv == cv) & cmask;
                int slot;
                if (hit) slot = __ffsll((long long)hit) - 1;
                else {
                    // victim: the first slot from the round-robin pointer on that this item does not use (there is one: C is the
                    // largest number of classes any item has)
                    const unsigned long long freeb = ~inuse & cmask, ahead = freeb & ~((1ull << rr) - 1ull);
                    slot = __ffsll((long long)(ahead ? ahead : freeb)) - 1;
                    rr = slot + 1 == C ? 0 : slot + 1;
                    fs_wave_copy_pairs(wl + slot * S, dict + (int64_t)cv * S, S >> 1, lane);
                    if (lane == slot) tagv = cv;
                    copied = true;
                }
                inuse |= 1ull << slot;
                const bool h0 = c0 == cv, h1 = c1 == cv;
                if (h0) b0 = slot * S;
                if (h1) b1 = slot * S;
                m0 &= ~__ballot(h0);
                m1 &= ~__ballot(h1);
            }
            if (copied) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        const double* __restrict__ v0 = wl + b0;
        const double* __restrict__ v1 = wl + b1;
        double a0 = 0.0, a1 = 0.0;
        v2d zi = {0.0, 0.0};
        // (measured and not kept, CG2 n = 107, product 204 us: the next round's loads in flight while this one is multiplied - a
        // second register set, 138 VGPRs, 3 waves per SIMD: 270 us; only the terms a run's length calls for, by wave-uniform branches
        // - most CG2 runs are one or two offsets long - : 339 us, the branches keep the coefficient reads from being batched)
        auto compute_round = [&](const v2d (&buf)[NR], int rd) {
            const double* __restrict__ w0 = v0 + NR * RL * rd;
            const double* __restrict__ w1 = v1 + NR * RL * rd;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const double e0 = buf[j].x, e1 = buf[j].y;
                const double e2 = fs_from_next_lane(buf[j].x);
                a0 = fma(w0[RL * j], e0, a0);     a1 = fma(w1[RL * j], e1, a1);
                a0 = fma(w0[RL * j + 1], e1, a0); a1 = fma(w1[RL * j + 1], e2, a1);
                if (RL == 3) {
                    const double e3 = fs_from_next_lane(buf[j].y);
                    a0 = fma(w0[RL * j + 2], e2, a0); a1 = fma(w1[RL * j + 2], e3, a1);
                }
                // (the coefficient positions are compile-time constants: left alone the compiler reads all 48 of a round
                // ahead of the first fma - 150 VGPRs, 3 waves per SIMD; a compiler barrier per run keeps it at the run's six)
                asm volatile("" ::: "memory");
            }
        };
        for (int rd = 0; rd < rounds; ++rd) {
            if (rd > 0) {
                if (LDSD) load_round(A, read_starts(pl + rd));
                else {
                    load_round(A, st_next);
                    st_next = read_starts(pl + (rd + 1 < rounds ? rd + 1 : rd));
                }
            }
            compute_round(A, rd);
            if (rd == 0) zi = A[0];
        }
        if (ok1) {
            v2d out;
            out.x = a0; out.y = a1;
            *reinterpret_cast<v2du*>(&y[r]) = out;
        } else if (ok0) y[r] = a0;
        if (!ok0) { a0 = 0.0; zi.x = 0.0; }
        if (!ok1) { a1 = 0.0; zi.y = 0.0; }
        if (DOTS == 1) {
            d_rz += ri.x * zi.x + ri.y * zi.y; d_wz += a0 * zi.x + a1 * zi.y; d_rr += ri.x * ri.x + ri.y * ri.y;
        } else if (DOTS == 2) {
            d_rz += a0 * ri.x + a1 * ri.y; d_wz += a0 * a0 + a1 * a1; d_rr += ri.x * ri.x + ri.y * ri.y;
        } else if (DOTS == 3) {
            d_rz += zi.x * zi.x + zi.y * zi.y; d_wz += a0 * zi.x + a1 * zi.y; d_rr += ri.x * zi.x * zi.x + ri.y * zi.y * zi.y;
        }
        if (!LDSD) __builtin_amdgcn_wave_barrier();            // (the next item's class rows may overwrite this one's)
    }
    if (DOTS && DOTS != 4) {
        const double t0 = fs_block_sum(d_rz, lds4);
        const double t1 = fs_block_sum(d_wz, lds4);
        const double t2 = fs_block_sum(d_rr, lds4);
        if (threadIdx.x == 0) {
            partials[part_base + blockIdx.x] = t0;
            partials[part_stride + part_base + blockIdx.x] = t1;
            partials[2 * part_stride + part_base + blockIdx.x] = t2;
        }
    }
}

// ---- the row-dictionary product on a LATTICE-ORDERED operator (fs_lattice.hip), x staged through LDS tiles ----------------------------
// k_dict_spmv fetches x with one 16-byte global load per run and lane: 20 - 39 vector-memory instructions per work item on a CG2 operator,
// whatever the numbering - the per-CU address path is what bounds it (0.17 of the HBM peak on BASELINE configs[3]).  In lattice order
// (row = X + SX (Y + NY Z) on the half grid) every neighbour of a row lies within +-2 in each direction, so a workgroup of eight waves
// takes a TILE of 128 x 4 x 4 rows and loads the (128 + 4) x 8 x 8 window of x around it into LDS with coalesced 16-byte loads (4.1 x
// 8 B per row, from L2), the even and the odd X of a line side by side in two halves of the window.  A wave then takes the 64 rows of
// ONE PARITY of a line (X = x0 + p, x0 + p + 2, ..): along a mesh line these are rows of one class (vertex rows, x-edge rows, ..), so
// the class's row - a list of (coefficient, window offset), the nonzero positions of the plan layout in plan order - is WAVE-UNIFORM.
// The wave loads it ahead of time (lane k: entry k), spreads it into its own LDS scratch and reads it back entry by entry as a
// broadcast; x comes from the window at consecutive addresses (no bank conflicts); one fma per STORED entry - no padded positions.  Two
// lines two planes apart (same parities: the same class) are multiplied together, one read of the list for both.  The order of a
// row's terms is that of k_dict_spmv and of the streaming kernels: the same bits (option "lattice_check" compares every row).
// Nothing is assumed about where classes change: a wave whose lanes are not of one class walks its DISTINCT classes one after the other
// (lat_wave_rows: the lanes of the other classes masked, the lists through scalar loads - slow, correct).  What the geometry buys is
// that this does not happen - EXCEPT at the ends of the lines: the first LT_LO and the last LT_HI rows of a line have classes of their
// own (boundary rows, and - the operator is scaled with its diagonal - the rows coupled to them), nine of 216 columns at configs[3].
// They are left out of the line waves: COLUMN tiles take them (the same machinery with X and Y exchanged, lanes along Y), in workgroups
// of their own at the front of the grid; the few rows at the ends of the end columns per lane from global memory.
// The lists come from the class rows and the plans (k_lat_table: one representative row per class; then EVERY row is checked: its
// plan puts its class's coefficients at the offsets of that list, its X has the parity the list's window offsets were worked out for -
// or the form is refused).
// MEASURED (round 5, MI355X, configs[3], 9.98 M rows, profiles/r05_p2_lattice_tiles.txt): inside the CG iteration, with the three dots,
// 155 us per product against 187 - 195 us for k_dict_spmv in the SPACE'S numbering and 222 - 272 us for it in lattice order; alone 134 us
// without / 149 us with the dots (the work-item product on the same operator: 168 / 230 us).  Of the 134 us the rows at the ends of the
// lines (4 % of the rows) take 30.  Steps on the way (all bit-identical): per-lane lists from LDS (three LDS reads per entry, loop lengths
// set by the vertex rows) 342 us; lists through scalar loads 689 us; lists handed out with v_readlane 240 us (16 cycles per readlane);
// LDS broadcast 184 us; paired lines 167 us; eight waves per tile 158 us; batches of four entries (no spills in the loop) 137 us; and the
// one that decided it: the compiler had hoisted every thread's nine window positions out of the tile loop into scratch and waited for
// each reload with vmcnt(0) - for the window load before it -, nine round trips per tile instead of one: with the dots 197 -> 154 us;
// whole window lines per wave (no division chains) 149 us.  Automatic from 270 000 rows on (option "lattice_order").
// Later in round 5 (DESIGN.md section 3 has the table): column tiles 150 us inside the iteration; interior strips as one long line
// (lat_tile_of) 139; a wave's classes from the per-tile table (k_lat_tile_table) 133; column / corner workgroups at the front of the
// grid, 512 tile workgroups, no private segment 120 us (105 - 115 in the trace; alone 99 us).
constexpr int LT_TX = 128, LT_TY = 4, LT_TZ = 4;
constexpr int LT_HX = LT_TX / 2 + 2;                // x positions of one parity in a window line
constexpr int LT_WY = LT_TY + 4, LT_WZ = LT_TZ + 4;
constexpr int LT_WINH = LT_HX * LT_WY * LT_WZ;      // doubles of one parity half of the window
constexpr int LT_WIN = 2 * LT_WINH;                 // 66 KB
constexpr int CT_XW = 8, CT_Y = 128, CT_Z = 4;     // column tiles (the ends of the lines): X positions of the window, rows along Y, planes
static_assert(2 * LT_HX * CT_XW * (CT_Z + 4) == LT_WIN && CT_Y / 2 + 2 == LT_HX, "a column tile's window is the tile window's LDS");
constexpr int LT_LOY = 4, LT_HIY = 4;               // rows at the two ends of a COLUMN whose classes are their own (as LT_LO / LT_HI; no dummy row in Y)
constexpr int LT_B2 = 4;                            // entries per batch of the paired-lines loop
constexpr int LT_BLOCK = 512;                       // threads of a tile's workgroup (eight waves share one window)
constexpr int LT_LO = 4, LT_HI = 5;                 // rows at the two ends of a line whose classes are their own: the boundary rows and - the operator is
                                                    // scaled with its diagonal - the rows coupled to them (and the dummy row that makes a line even)
constexpr int LT_ML = 72;                           // entries per class row (a CG2 vertex row of a Kuhn mesh has up to 65), padded to 8

// the tiles of a plane of tiles (k_lattice_spmv's tile loop and k_lat_tile_table walk them the same way)
struct lat_geom {
    int64_t n_tiles;
    int nxc, tiles_z, w_ys, w_ye, w_tiles;
    int n_ct_wgs, n_extra, grid;        // workgroups 0 .. n_ct_wgs - 1: column tiles; .. n_extra - 1: corner rows; .. grid - 1: tiles
};
struct lat_tile {
    int64_t x0, y0, z0, ylim;
    bool wrap;
};
// A plane of tiles: the strips (four lines in Y) before w_ys and from w_ye on in tiles of their own, nxc to a strip; the strips
// w_ys .. w_ye - 1 - every line of theirs an interior line in Y - as ONE long line per line number: a tile takes 128 consecutive
// positions of it, wherever they start, and where it runs over the end of a line it goes on at the start of the line FOUR lines
// up (the same line number in the next strip: same parities, normally the same class; if not, the wave takes its classes one by
// one as anywhere).  The window and the rows wrap the same way, so a row's neighbours stay where its list expects them: the
// first and last rows of a line are not tile rows.  (SX = 216 at configs[3]: two tiles a strip with 88 of 128 positions used in
// the second become 87 tiles for 51 strips - 5 022 tiles instead of 5 832.)
__device__ __forceinline__ lat_tile lat_tile_of(int64_t tile, const lat_geom& G, int64_t SX, int64_t NY) {
    const int tile_z = (int)(tile / G.tiles_z), tq = (int)(tile - (int64_t)tile_z * G.tiles_z);
    const int w_first = G.w_ys * G.nxc;
    int tx0, ty0;
    lat_tile T;
    T.wrap = false;
    if (tq < w_first) { tx0 = (tq % G.nxc) * LT_TX; ty0 = (tq / G.nxc) * LT_TY; }
    else if (tq < w_first + G.w_tiles) {
        const int g0 = (tq - w_first) * LT_TX, yo = g0 / (int)SX;
        tx0 = g0 - yo * (int)SX;
        ty0 = (G.w_ys + yo) * LT_TY;
        T.wrap = true;
    } else {
        const int q2 = tq - w_first - G.w_tiles;
        tx0 = (q2 % G.nxc) * LT_TX;
        ty0 = (G.w_ye + q2 / G.nxc) * LT_TY;
    }
    T.x0 = tx0; T.y0 = ty0; T.z0 = (int64_t)tile_z * LT_TZ;
    T.ylim = T.wrap ? (int64_t)G.w_ye * LT_TY : NY;           // (rows behind the long line's end belong to the tiles of strip w_ye)
    return T;
}
// row u of a lane: wave w takes the (line, parity) pairs w, w + 8, ..; -1: not a tile row
__device__ __forceinline__ int32_t lat_tile_row(const lat_tile& T, int u, int wave, int lane, int64_t SX, int64_t NY, int64_t NZ) {
    const int wl = wave + (LT_BLOCK / 64) * u, line = wl >> 1, p = wl & 1;
    int64_t X = T.x0 + 2 * lane + p, Y = T.y0 + (line % LT_TY);
    const int64_t Z = T.z0 + (line / LT_TY);
    if (T.wrap && X >= SX) { X -= SX; Y += LT_TY; }
    const bool in = X >= LT_LO && X <= SX - 1 - LT_HI && Y < T.ylim && Z < NZ;
    return in ? (int32_t)(X + SX * (Y + NY * Z)) : -1;
}

static const int g_lt_dbg_env = getenv("FS_LATTICE_BITS") ? atoi(getenv("FS_LATTICE_BITS")) : 0;      // (experiments: bits or-ed into the kernel's dbg argument)
static int g_lt_dbg = 0;     // (FS_LATTICE_DEBUG times k_lattice_spmv once more without the rows at the ends of the lines: 1)
struct lat_tables {
    dbuf<int32_t> rep, cnt, off, rel;   // [ncls] representative row, [ncls] entries, [ncls][LT_ML] column offsets (verification) / window offsets
    dbuf<int32_t> relc;                 // [ncls][LT_ML] the offsets in the window of a COLUMN tile (the ends of the lines: lanes along Y)
    dbuf<double> coef;                  // [ncls][LT_ML]
    dbuf<int> info;                     // [0] entries that do not fit / rows whose plan or parity disagrees
    dbuf<uint32_t> tile_cls;            // [tile][wave][u]: the class of the wave's u-th line (bits 0 - 15), its entries (16 - 23), all live
                                        // lanes of that class (24), any live lane (25): k_lat_tile_table
    lat_geom geom = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double* built_for = nullptr;
    uint64_t space_serial = 0;
    int ncls = 0;
    bool ok = false;
    bool judged = false;                // lat_prepare looked at a matrix since the flag was last cleared (fs_krylov_solve: a solve that never
                                        // builds a dictionary - BiCGStab, no diagonal scaling - says nothing about the tile form)
    bool tables_ok = false;             // the lists were built and every row fits them: for dictionary tables number dict_built of that space
    int64_t dict_built = -1;
};
static lat_tables g_lat;

__global__ void k_lat_rep(int64_t n, const uint16_t* __restrict__ cls, int32_t* __restrict__ rep) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n; r += stride) {
        const int c = cls[r];
        if (__hip_atomic_load(&rep[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (int32_t)r) atomicMin(&rep[c], (int32_t)r);
    }
}

// BUILD: the representative row of every class writes the class's list from its item's plan; !BUILD: one lane per distinct class of
// every item checks that the item's plan gives the same offsets, every row that its X has the parity of its class's representative
// (or its list holds nothing but the diagonal)
template <bool BUILD>
__global__ void __launch_bounds__(FS_BLOCK) k_lat_table(int64_t n_items, const int4* __restrict__ items, const dict_plan_round* __restrict__ plans,
                                                        const uint16_t* __restrict__ cls, const int32_t* __restrict__ rep,
                                                        const double* __restrict__ values, int S, int RL, int NR, int64_t SX, int64_t NY,
                                                        int32_t* __restrict__ cnt, double* __restrict__ coef, int32_t* __restrict__ rel,
                                                        int32_t* __restrict__ off, int* __restrict__ info, int32_t* __restrict__ relc) {
    const int lane = threadIdx.x & 63;
    int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t plane = SX * NY;
    int bad = 0;
    for (; q < n_items; q += stride) {
        const int4 it = items[q];
        const int32_t first = it.x, nr = it.y & 0xffff;
        const dict_plan_round* __restrict__ pl = plans + it.z;
        const int n_runs = NR * it.w;
        for (int half = 0; half < 2; ++half) {
            const int i = half * 64 + lane;
            const int32_t r = first + i;
            const int c = i < nr ? (int)cls[r] : -1;
            const int p = (int)((r % SX) & 1), py = (int)(((r / SX) % NY) & 1);
            bool work;
            if (BUILD) work = c >= 0 && rep[c] == r;
            else {      // the first lane of every class among these 64 rows
                if (c >= 0 && (((rep[c] % SX) & 1) != p || (((rep[c] / SX) % NY) & 1) != py) && !(cnt[c] == 1 && off[(int64_t)c * LT_ML] == 0)) ++bad;
                work = false;
                unsigned long long todo = __ballot(c >= 0);
                while (todo) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const int cv = __shfl(c, leader, 64);
                    if (lane == leader) work = true;
                    todo &= ~__ballot(c == cv);
                }
            }
            if (!work) continue;
            const double* __restrict__ dv = values + (int64_t)c * S;
            int k = 0;
            const int have = BUILD ? 0 : cnt[c];
            for (int g = 1; g < n_runs; ++g) {          // (run 0 is the z run: no coefficients)
                const int32_t st = dict_run_start(pl, NR, g);
                const int ln = dict_run_len(pl, NR, g);
                for (int t = 0; t < ln; ++t) {
                    const double v = RL * g + t < S ? dv[RL * g + t] : 0.0;
                    if (v == 0.0) continue;
                    const int32_t o = st + t;
                    if (BUILD) {
                        // o = dx + SX (dy + NY dz) with |dx|, |dy|, |dz| <= 2
                        const int64_t dz = (o + (o >= 0 ? plane / 2 : -(plane / 2))) / plane;
                        const int64_t rem = o - dz * plane;
                        const int64_t dy = (rem + (rem >= 0 ? SX / 2 : -(SX / 2))) / SX;
                        const int64_t dx = rem - dy * SX;
                        const bool fits = k < LT_ML && dx >= -2 && dx <= 2 && dy >= -2 && dy <= 2 && dz >= -2 && dz <= 2;
                        if (fits) {
                            // window offset from the row's own position: the other parity's half for odd dx, floor((p + dx) / 2) along x
                            const int pd = p + (int)dx, p2 = pd & 1, di = (pd - p2) / 2;
                            coef[(int64_t)c * LT_ML + k] = v;
                            off[(int64_t)c * LT_ML + k] = o;
                            rel[(int64_t)c * LT_ML + k] = (p2 - p) * LT_WINH + di + LT_HX * ((int)dy + LT_WY * (int)dz);
                            // column tiles: the two halves hold the even / odd Y, a line of the window runs along Y
                            const int qd = py + (int)dy, q2 = qd & 1, dj = (qd - q2) / 2;
                            relc[(int64_t)c * LT_ML + k] = (q2 - py) * LT_WINH + dj + LT_HX * ((int)dx + CT_XW * (int)dz);
                        } else ++bad;
                    } else if (k >= have || off[(int64_t)c * LT_ML + k] != o) ++bad;
                    ++k;
                }
            }
            if (BUILD) cnt[c] = k <= LT_ML ? k : LT_ML;
            else if (k != have) ++bad;
        }
    }
    if (bad) atomicAdd(&info[0], bad);
}

// what k_lattice_spmv needs to know about the four lines of a wave of a tile before it can ask for their lists: once per set of lists
__global__ void __launch_bounds__(LT_BLOCK) k_lat_tile_table(lat_geom G, int64_t SX, int64_t NY, int64_t NZ, const uint16_t* __restrict__ cls,
                                                              const int32_t* __restrict__ tcnt, uint32_t* __restrict__ tab) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t tile = blockIdx.x; tile < G.n_tiles; tile += gridDim.x) {
        const lat_tile T = lat_tile_of(tile, G, SX, NY);
        for (int u = 0; u < 2 * LT_TY * LT_TZ / (LT_BLOCK / 64); ++u) {
            const int32_t r = lat_tile_row(T, u, wave, lane, SX, NY, NZ);
            const int c = r >= 0 ? (int)cls[r] : -1;
            const unsigned long long live = __ballot(c >= 0);
            const int cm = live ? __shfl(c, __ffsll((long long)live) - 1, 64) : 0;
            const bool uni = __ballot(c == cm) == live;
            const int cn = live ? tcnt[cm] : 0;
            if (lane == 0)
                tab[(tile * (LT_BLOCK / 64) + wave) * 4 + u] = (uint32_t)cm | ((uint32_t)cn << 16) | ((uint32_t)uni << 24) | ((uint32_t)(live != 0) << 25);
        }
    }
}

// the rows of a wave: the distinct classes of its lanes one after the other, the class's list through scalar loads
__device__ __forceinline__ double lat_wave_rows(int c, int own, const double* __restrict__ win, const int32_t* __restrict__ tcnt,
                                                const double* __restrict__ tcoef, const int32_t* __restrict__ trel) {
    double a = 0.0;
    unsigned long long todo = __ballot(c >= 0);
    while (todo) {
        const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
        const int cm = __builtin_amdgcn_readlane(c, leader);
        const bool mine = c == cm;
        const int cn = __builtin_amdgcn_readfirstlane(tcnt[cm]);
        const double* __restrict__ cp = tcoef + (int64_t)cm * LT_ML;
        const int32_t* __restrict__ rp = trel + (int64_t)cm * LT_ML;
        for (int k0 = 0; k0 < cn; k0 += LT_B2) {        // (a list is padded to a multiple of 8 positions)
            double cf[LT_B2], xv[LT_B2];
            int32_t rl[LT_B2];
#pragma unroll
            for (int e = 0; e < LT_B2; ++e) {
                cf[e] = cp[k0 + e];
                rl[e] = k0 + e < cn ? rp[k0 + e] : 0;
            }
#pragma unroll
            for (int e = 0; e < LT_B2; ++e) xv[e] = win[mine ? own + rl[e] : own];     // (another class's offsets may leave the window)
#pragma unroll
            for (int e = 0; e < LT_B2; ++e)
                if (k0 + e < cn && mine) a = fma(cf[e], xv[e], a);
        }
        todo &= ~__ballot(mine);
    }
    return a;
}

// the same for a wave whose lanes are ALL of one class (the rule): the list was loaded into registers ahead of time, entry k in lane k
// (k + 64: second set); the wave spreads it into its own LDS scratch and every lane reads entry after entry from there - the same
// address in all lanes, a broadcast (2 cycles on gfx950) - then x at its own position + the entry's offset, one fma.  (v_readlane
// handed the entries out without LDS and took 16 cycles each: 220 instead of 110 us per product.)  The positions behind the end of a
// list hold (0.0, 0): they add +0 * x[row].
__device__ __forceinline__ double lat_wave_rows_uniform(int cn, int own, const double* __restrict__ win, double* __restrict__ sc, int* __restrict__ sr,
                                                        double vc0, int vr0, double vc1, int vr1) {
    const int lane = threadIdx.x & 63;
    sc[lane] = vc0;
    sr[lane] = vr0;
    if (lane < LT_ML - 64) { sc[64 + lane] = vc1; sr[64 + lane] = vr1; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double a = 0.0;
    for (int k0 = 0; k0 < cn; k0 += 8) {
        double xv[8], cf[8];
        int rl[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { cf[e] = sc[k0 + e]; rl[e] = sr[k0 + e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = win[own + rl[e]];
#pragma unroll
        for (int e = 0; e < 8; ++e) a = fma(cf[e], xv[e], a);
    }
    __builtin_amdgcn_wave_barrier();        // (the next line's list overwrites the scratch)
    return a;
}

// ... and for TWO lines of the same class (lines two planes apart: same parities): one read of the list serves both rows of a lane, whose
// two fma chains are independent
__device__ __forceinline__ void lat_wave_rows_uniform2(int cn, int own_a, int own_b, const double* __restrict__ win, double* __restrict__ sc, int* __restrict__ sr,
                                                       double vc0, int vr0, double vc1, int vr1, double& ra, double& rb) {
    const int lane = threadIdx.x & 63;
    sc[lane] = vc0;
    sr[lane] = vr0;
    if (lane < LT_ML - 64) { sc[64 + lane] = vc1; sr[64 + lane] = vr1; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double a = 0.0, b = 0.0;
    for (int k0 = 0; k0 < cn; k0 += LT_B2) {       // (batches of four: eight cost 28 more registers and spills at four waves per SIMD)
        double xa[LT_B2], xb[LT_B2], cf[LT_B2];
        int rl[LT_B2];
#pragma unroll
        for (int e = 0; e < LT_B2; ++e) { cf[e] = sc[k0 + e]; rl[e] = sr[k0 + e]; }
#pragma unroll
        for (int e = 0; e < LT_B2; ++e) { xa[e] = win[own_a + rl[e]]; xb[e] = win[own_b + rl[e]]; }
#pragma unroll
        for (int e = 0; e < LT_B2; ++e) { a = fma(cf[e], xa[e], a); b = fma(cf[e], xb[e], b); }
    }
    __builtin_amdgcn_wave_barrier();        // (the next line's list overwrites the scratch)
    ra = a;
    rb = b;
}

template <int DOTS>
__global__ void __launch_bounds__(LT_BLOCK, 4) k_lattice_spmv(lat_geom G, const uint32_t* __restrict__ tile_cls, int64_t SX, int64_t NY, int64_t NZ,
                                                           const uint16_t* __restrict__ cls, const int32_t* __restrict__ tcnt,
                                                           const double* __restrict__ tcoef, const int32_t* __restrict__ trel, const int32_t* __restrict__ toff, const int32_t* __restrict__ trelc,
                                                           const double* __restrict__ x, double* __restrict__ y,
                                                           const double* __restrict__ rvec, double* __restrict__ partials,
                                                           int* __restrict__ status, int part_base, int part_stride, int bump, int dbg) {
    const int st0 = DOTS ? status[0] : 0;
    typedef double v2d __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) double win[];      // [2][LT_WZ][LT_WY][LT_HX]: even X, odd X
    __shared__ double ldsw[LT_BLOCK / 64];
    __shared__ double list_c[LT_BLOCK / 64][LT_ML];                   // a wave's scratch: the class list of the line it multiplies
    __shared__ int list_r[LT_BLOCK / 64][LT_ML];
    if (DOTS) {
        if (st0 != 0) return;
        if (bump && blockIdx.x == 0 && threadIdx.x == 0) status[2] += 1;      // see k_sell_spmv
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = SX * NY * NZ;
    double d_rz = 0.0, d_wz = 0.0, d_rr = 0.0;
    constexpr int NL = LT_TY * LT_TZ;               // lines of a tile
    constexpr int NWV = LT_BLOCK / 64;              // waves
    constexpr int U = 2 * NL / NWV;                 // (line, parity) pairs per wave
    auto finish = [&](int64_t r, double a, double zi, double ri) {
        y[r] = a;
        if (DOTS && DOTS != 4 && !(dbg & 4)) {
            if (DOTS == 1) { d_rz += ri * zi; d_wz += a * zi; d_rr += ri * ri; }
            else if (DOTS == 2) { d_rz += a * ri; d_wz += a * a; d_rr += ri * ri; }
            else if (DOTS == 3) { d_rz += zi * zi; d_wz += a * zi; d_rr += ri * zi * zi; }
        }
    };
    // ---- FIRST the corners: rows of the end columns within LT_LOY / LT_HIY of the ends of their column; every lane its own list, everything
    // from global memory - a chain of dependent loads, 19 us for 0.2 % of the rows when it ran behind the tiles.  A wave: one column, eight
    // planes x the eight rows.  Corners and column tiles go to the workgroups counted from the LAST: the tiles are dealt out from the
    // first (xcd_chunks), so these have one tile less to do than the others.
    {
        const int nzb = (int)((NZ + 7) >> 3);
        const int64_t n_tasks = (int64_t)(LT_LO + LT_HI) * nzb;
        const bool corner_wg = (int)blockIdx.x >= G.n_ct_wgs && (int)blockIdx.x < G.n_extra;
        for (int64_t t = corner_wg ? (int64_t)((int)blockIdx.x - G.n_ct_wgs) * NWV + wave : n_tasks; t < n_tasks && !(dbg & 9); t += (int64_t)(G.n_extra - G.n_ct_wgs) * NWV) {
            const int col = (int)(t % (LT_LO + LT_HI)), d = lane & 7;
            const int64_t Z = (t / (LT_LO + LT_HI)) * 8 + (lane >> 3);
            const int64_t Y = d < LT_LOY ? d : NY - (LT_LOY + LT_HIY) + d;
            const int64_t X = col < LT_LO ? col : SX - (LT_LO + LT_HI) + col;
            const bool in = Z < NZ;
            const int64_t rr = X + SX * (Y + NY * (in ? Z : 0));
            const int cc = (int)cls[rr];
            int most = in ? tcnt[cc] : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int m2 = __shfl_xor(most, o, 64); most = m2 > most ? m2 : most; }
            const double* __restrict__ cp = tcoef + (int64_t)cc * LT_ML;
            const int32_t* __restrict__ op = toff + (int64_t)cc * LT_ML;
            double a = 0.0;
            for (int k0 = 0; k0 < most; k0 += 8) {       // (behind a list's end: (0.0, offset 0))
                double cf[8], xv[8];
                int32_t of[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { cf[e] = cp[k0 + e]; of[e] = op[k0 + e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[e] = x[rr + of[e]];
#pragma unroll
                for (int e = 0; e < 8; ++e) a = fma(cf[e], xv[e], a);
            }
            if (in) finish(rr, a, x[rr], (DOTS && DOTS != 4) ? rvec[rr] : 0.0);
        }
    }
    // ---- the first LT_LO and the last LT_HI rows of every line (classes of their own): COLUMN tiles.  Along Y such a column is what a line
    // is along X - rows of one class per parity -, so the same machinery runs with the roles of X and Y exchanged: a workgroup takes
    // the end columns of one side x 128 Y x 4 Z, loads the window (8 X positions x 132 Y x 8 Z: the same 66 KB, even and odd Y in two
    // halves, a line of the window along Y), a wave the 64 rows of one Y parity of one (column, plane), two planes two apart together.
    // The rows at the ends of the COLUMNS (first / last four Y: classes of their own again, 0.2 % of the rows) follow per lane below.
    const int nycc = (int)((NY + CT_Y - 1) / CT_Y), nzc = (int)((NZ + CT_Z - 1) / CT_Z);
    const int64_t n_ct = (int64_t)2 * nycc * nzc;
    for (int64_t ct = (int)blockIdx.x < G.n_ct_wgs ? (int64_t)blockIdx.x : n_ct; ct < n_ct && !(dbg & 1); ct += G.n_ct_wgs) {
        const int side = (int)(ct & 1);
        const int64_t ycn = (ct >> 1) % nycc, zcn = (ct >> 1) / nycc;
        const int64_t y0 = ycn * CT_Y, z0 = zcn * CT_Z;
        const int64_t xw0 = side ? SX - CT_XW : 0;                  // X of window position 0
        const int ncol = side ? LT_HI : LT_LO;
        const int64_t col0 = side ? SX - LT_HI : 0;                 // first end column of this side
        // window: wave w takes plane w; its items (Y line, pair of X positions): 132 x 4, four lanes a line's 64 bytes
        // (three loads in flight per lane at a time, a real loop: all nine at once - 36 registers - had this part of the kernel spill, and a
        // kernel with a private segment pays for it at EVERY launch, in front of the kernel where no event and no trace sees it: about 30 us
        // per launch of the 2 296 workgroups here)
        constexpr int NIT = (2 * LT_HX * (CT_XW / 2) + 63) / 64, NIH = 3;
        static_assert(NIT % NIH == 0, "the window of a column tile is loaded in whole batches");
        {
            const int64_t Z = z0 - 2 + wave;
#pragma unroll 1
            for (int u0 = 0; u0 < NIT; u0 += NIH) {
                v2d wv[NIH];
#pragma unroll
                for (int u = 0; u < NIH; ++u) {
                    const int item = (u0 + u) * 64 + lane, yl = item >> 2, xp = item & 3;
                    const int64_t Y = y0 - 2 + yl;
                    wv[u] = v2d{0.0, 0.0};
                    if (yl < 2 * LT_HX && Y >= 0 && Y < NY && Z >= 0 && Z < NZ)
                        wv[u] = *reinterpret_cast<const v2d*>(x + xw0 + 2 * xp + SX * (Y + NY * Z));
                }
#pragma unroll
                for (int u = 0; u < NIH; ++u) {
                    const int item = (u0 + u) * 64 + lane, yl = item >> 2, xp = item & 3;
                    if (yl < 2 * LT_HX) {
                        const int i = (yl & 1) * LT_WINH + (yl >> 1) + LT_HX * (2 * xp + CT_XW * wave);
                        win[i] = wv[u].x;
                        win[i + LT_HX] = wv[u].y;
                    }
                }
            }
        }
        // pairs of column lines: q = (column, Y parity, plane pair): planes zl and zl + 2
        const int npair = ncol * 4;
        constexpr int UC = (LT_HI * 4 + NWV - 1) / NWV;
        int32_t r[2 * UC];
        double ri[2 * UC];
        int cm[2 * UC], cn[2 * UC];
        bool uni[2 * UC];
        auto own_c = [&](int q, int half) {
            const int col = q >> 2, py = (q >> 1) & 1, zl = (q & 1) + 2 * half;
            return py * LT_WINH + (lane + 1) + LT_HX * ((int)(col0 - xw0) + col + CT_XW * (zl + 2));
        };
#pragma unroll
        for (int j = 0; j < UC; ++j)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int q = wave + NWV * j, col = q >> 2, py = (q >> 1) & 1, zl = (q & 1) + 2 * half;
                const int64_t X = col0 + col, Y = y0 + 2 * lane + py, Z = z0 + zl;
                const bool in = q < npair && Y >= LT_LOY && Y <= NY - 1 - LT_HIY && Z < NZ;
                const int u = 2 * j + half;
                r[u] = in ? (int32_t)(X + SX * (Y + NY * Z)) : -1;
                ri[u] = (DOTS && DOTS != 4 && in) ? rvec[r[u]] : 0.0;
                const int c = in ? (int)cls[r[u]] : -1;        // (a lane's own class number is not kept: registers)
                const unsigned long long live = __ballot(c >= 0);
                cm[u] = live ? __builtin_amdgcn_readlane(c, __builtin_amdgcn_readfirstlane(__ffsll((long long)live) - 1)) : 0;
                uni[u] = __ballot(c == cm[u]) == live;
                cn[u] = live ? __builtin_amdgcn_readfirstlane(tcnt[cm[u]]) : 0;
            }
        auto cls_of = [&](int32_t row) { return row >= 0 ? (int)cls[row] : -1; };
        struct lat_list { double c0, c1; int r0, r1; };
        auto load_list = [&](int cls_m) {
            lat_list L;
            const int64_t at = (int64_t)cls_m * LT_ML + lane;
            L.c0 = tcoef[at];
            L.r0 = trelc[at];
            L.c1 = tcoef[at + 64];
            L.r1 = trelc[at + 64];
            return L;
        };
        lat_list cur = load_list(cm[0]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < UC; ++j) {
            const int ua = 2 * j, ub = 2 * j + 1, q = wave + NWV * j;
            lat_list nxt = cur;
            if (j + 1 < UC) nxt = load_list(cm[2 * (j + 1)]);
            const bool both = uni[ua] && uni[ub] && cm[ua] == cm[ub] && cn[ua] != 0 && cn[ub] != 0;
            if (both) {
                double ra, rb;
                lat_wave_rows_uniform2(cn[ua], own_c(q, 0), own_c(q, 1), win, list_c[wave], list_r[wave], cur.c0, cur.r0, cur.c1, cur.r1, ra, rb);
                if (r[ua] >= 0) finish(r[ua], ra, win[own_c(q, 0)], ri[ua]);
                if (r[ub] >= 0) finish(r[ub], rb, win[own_c(q, 1)], ri[ub]);
            } else {
                if (cn[ua] != 0) {
                    double a;
                    if (uni[ua]) a = lat_wave_rows_uniform(cn[ua], own_c(q, 0), win, list_c[wave], list_r[wave], cur.c0, cur.r0, cur.c1, cur.r1);
                    else a = lat_wave_rows(cls_of(r[ua]), own_c(q, 0), win, tcnt, tcoef, trelc);
                    if (r[ua] >= 0) finish(r[ua], a, win[own_c(q, 0)], ri[ua]);
                }
                if (cn[ub] != 0) {
                    double a;
                    if (uni[ub]) {
                        const lat_list lb = load_list(cm[ub]);
                        a = lat_wave_rows_uniform(cn[ub], own_c(q, 1), win, list_c[wave], list_r[wave], lb.c0, lb.r0, lb.c1, lb.r1);
                    } else a = lat_wave_rows(cls_of(r[ub]), own_c(q, 1), win, tcnt, tcoef, trelc);
                    if (r[ub] >= 0) finish(r[ub], a, win[own_c(q, 1)], ri[ub]);
                }
            }
            cur = nxt;
        }
        __syncthreads();
    }
    // the tiles: the workgroups behind the first n_extra (a multiple of 8: workgroup b runs on XCD b mod 8), each XCD a contiguous eighth
    chunk_iter it;
    {
        const int b = (int)blockIdx.x - G.n_extra, tg = G.grid - G.n_extra;
        const int64_t per_xcd = (G.n_tiles + 7) >> 3, e = ((b & 7) + 1) * per_xcd;
        it.step = tg >> 3;
        it.cur = b >= 0 ? (b & 7) * per_xcd + (b >> 3) : G.n_tiles;
        it.end = e < G.n_tiles ? e : G.n_tiles;
    }
    for (; it.cur < it.end; it.cur += it.step) {
        const lat_tile T = lat_tile_of(it.cur, G, SX, NY);
        const int64_t x0 = T.x0, y0 = T.y0, z0 = T.z0;
        const bool wrap = T.wrap;
        // the classes of this wave's four lines: one scalar load (k_lat_tile_table wrote them when the lists were made), asked for with
        // the window - where the class numbers used to be read per lane, compared across the wave and their list lengths fetched, three
        // dependent round trips before the first list could be asked for
        const uint4 tc = *reinterpret_cast<const uint4*>(tile_cls + (it.cur * (LT_BLOCK / 64) + __builtin_amdgcn_readfirstlane(wave)) * 4);
        // ---- the window of x: lines (y0 - 2 .. y0 + LT_TY + 1) x (z0 - 2 .. ) from X = x0 - 2 on, one 16-byte load per (even, odd)
        // pair, all of a thread's loads in flight together; outside the lattice: zero
        // A wave takes LPW whole window lines: lane l the pair l + 1 of each (64 of the 66 pairs of a line, 1 KB per wave and load, the
        // line's start and validity wave-uniform), and in one more load the 2 x LPW pairs at the ends of its lines.  (Dealing the
        // window's pairs to the threads by index cost a division chain per load - which the compiler hoisted out of the tile loop
        // into scratch, whose reloads it then waited for with vmcnt(0): every window load behind the one before it, + 60 us.)
        constexpr int LPW = LT_WY * LT_WZ / NWV;        // window lines per wave
        static_assert(LPW * NWV == LT_WY * LT_WZ && 2 * LPW <= 64 && LT_HX == 66, "window lines are dealt to the waves whole");
        constexpr int NW = LPW + 1;
        v2d wv[NW];
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            // u < LPW: line wave * LPW + u, pair lane + 1; u == LPW: lanes 0 .. 2 LPW - 1: line wave * LPW + (lane >> 1), pair 0 / 65
            const int wline = wave * LPW + (u < LPW ? u : (lane >> 1) % LPW);
            const int xx = u < LPW ? lane + 1 : ((lane & 1) ? LT_HX - 1 : 0);
            const int yy = wline % LT_WY, zz = wline / LT_WY;
            int64_t Y = y0 - 2 + yy, Xw = x0 - 2 + 2 * xx;
            const int64_t Z = z0 - 2 + zz;
            if (wrap && Xw >= SX) { Xw -= SX; Y += LT_TY; }
            wv[u] = v2d{0.0, 0.0};
            if ((u < LPW || lane < 2 * LPW) && Y >= 0 && Y < NY && Z >= 0 && Z < NZ) {
                int64_t g = Xw + SX * (Y + NY * Z);                    // (even: SX and x0 are)
                g = g < 0 ? 0 : (g > n - 2 ? n - 2 : g);               // columns before / behind the vector carry no entry
                wv[u] = *reinterpret_cast<const v2d*>(x + g);
            }
        }
        // ---- the rows of this thread: U line waves (line, parity); their class numbers, then the list of the wave's first class (lane k:
        // entries k and k + 64), asked for before the window is waited for
        // (row numbers as 32-bit, a row's own window position recomputed where it is used: registers are what this kernel is short of)
        int32_t r[U];
        double ri[U];
        auto own_of = [&](int u) {
            const int wl = wave + NWV * u, line = wl >> 1, p = wl & 1;
            return p * LT_WINH + (lane + 1) + LT_HX * ((line % LT_TY) + 2 + LT_WY * ((line / LT_TY) + 2));
        };
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u] = lat_tile_row(T, u, wave, lane, SX, NY, NZ);
            const bool in = r[u] >= 0;
            // (the residual entries of the fused dots: asked for HERE.  Asked for next to the multiplication, their wait - the youngest
            // loads of the wave: vmcnt(0) - was also a wait for the previous lines' stores of y: + 60 us per product)
            ri[u] = (DOTS && DOTS != 4 && in && !(dbg & 2)) ? rvec[r[u]] : 0.0;
        }
        int cm[U], cn[U];
        bool uni[U];
        static_assert(U == 4, "a wave's four lines in one 16-byte word of the tile table");
        const uint32_t tcw[U] = {tc.x, tc.y, tc.z, tc.w};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cm[u] = (int)(tcw[u] & 0xffffu);
            cn[u] = (tcw[u] >> 25) & 1u ? (int)((tcw[u] >> 16) & 0xffu) : 0;
            uni[u] = (tcw[u] >> 24) & 1u;
        }
        struct lat_list { double c0, c1; int r0, r1; };
        auto load_list = [&](int cls_m) {
            lat_list L;
            const int64_t at = (int64_t)cls_m * LT_ML + lane;
            L.c0 = tcoef[at];
            L.r0 = trel[at];
            L.c1 = tcoef[at + 64];          // (the tables are padded: no branch, so that the wait for a list is counted exactly)
            L.r1 = trel[at + 64];
            return L;
        };
        auto cls_of = [&](int32_t row) { return row >= 0 ? (int)cls[row] : -1; };        // (a wave of several classes: per lane, when it comes to it)
        // lines u and u + U / 2 of a wave lie two planes apart (same parities in X, Y and Z: normally the same class): taken together
        static_assert(LT_TZ == 4 && LT_TY == 4, "the pairing of lines below assumes 4 x 4 lines per tile");
        lat_list cur = load_list(cm[0]);
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int wline = wave * LPW + (u < LPW ? u : (lane >> 1) % LPW);
            const int xx = u < LPW ? lane + 1 : ((lane & 1) ? LT_HX - 1 : 0);
            const int i = xx + LT_HX * wline;
            if (u < LPW || lane < 2 * LPW) { win[i] = wv[u].x; win[LT_WINH + i] = wv[u].y; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < U / 2; ++j) {
            // lines j and j + U / 2 of this wave: the same (line % LT_TY), Z two apart
            const int ua = j, ub = j + U / 2;
            lat_list nxt = cur;
            if (j + 1 < U / 2) nxt = load_list(cm[j + 1]);      // (in flight while these lines are multiplied)
            const bool both = uni[ua] && uni[ub] && cm[ua] == cm[ub] && cn[ua] != 0 && cn[ub] != 0;
            const double ria = ri[ua], rib = ri[ub];
            if (both) {
                double ra, rb;
                lat_wave_rows_uniform2(cn[ua], own_of(ua), own_of(ub), win, list_c[wave], list_r[wave], cur.c0, cur.r0, cur.c1, cur.r1, ra, rb);
                if (r[ua] >= 0) finish(r[ua], ra, win[own_of(ua)], ria);
                if (r[ub] >= 0) finish(r[ub], rb, win[own_of(ub)], rib);
            } else {
                if (cn[ua] != 0) {
                    double a;
                    if (uni[ua]) a = lat_wave_rows_uniform(cn[ua], own_of(ua), win, list_c[wave], list_r[wave], cur.c0, cur.r0, cur.c1, cur.r1);
                    else a = lat_wave_rows(cls_of(r[ua]), own_of(ua), win, tcnt, tcoef, trel);
                    if (r[ua] >= 0) finish(r[ua], a, win[own_of(ua)], ria);
                }
                if (cn[ub] != 0) {          // (its list was not asked for ahead of time: tiles where the class changes between the planes)
                    double a;
                    if (uni[ub]) {
                        const lat_list lb = load_list(cm[ub]);
                        a = lat_wave_rows_uniform(cn[ub], own_of(ub), win, list_c[wave], list_r[wave], lb.c0, lb.r0, lb.c1, lb.r1);
                    } else a = lat_wave_rows(cls_of(r[ub]), own_of(ub), win, tcnt, tcoef, trel);
                    if (r[ub] >= 0) finish(r[ub], a, win[own_of(ub)], rib);
                }
            }
            cur = nxt;
        }
        __syncthreads();            // (the next tile overwrites the window)
    }
    if (DOTS && DOTS != 4) {
        // (fixed order: shuffle reduction per wave, the waves' sums added in order by thread 0)
        double t[3] = {d_rz, d_wz, d_rr};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double v = t[q];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) ldsw[wave] = v;
            __syncthreads();
            if (threadIdx.x == 0) {
                double acc = 0.0;
                for (int w = 0; w < LT_BLOCK / 64; ++w) acc += ldsw[w];
                partials[q * part_stride + part_base + blockIdx.x] = acc;
            }
            __syncthreads();
        }
    }
}
static size_t lat_lds_bytes() { return (size_t)LT_WIN * 8; }

// ---- the same product for 3 x 3 block rows (vector P1 spaces: the elasticity operator of a uniform box; the fine level of its AMG) --
// A lane holds two consecutive NODES (six rows); a run's load is the six values x[3 (r + o)] .. x[3 (r + o) + 5] (three 16-byte
// loads), the next two nodes come from the next lane; class rows are [position][9].  Per stored block the terms are added in the
// streaming kernel's order (k_sell_spmv<3, ..>: column component outer, row component inner): same bits.  Rounds are taken in two
// halves of four runs (a round's 48 doubles of x per lane would not leave room for anything else).  No fused dots: this is the
// product of fs_spmv_dev - the AMG V-cycle's fine level and its CG.
template <int RL>
__global__ void __launch_bounds__(FS_BLOCK) k_dict_spmv3(int64_t n_cols, int64_t n_items, const int4* __restrict__ items,
                                                         const dict_plan_round* __restrict__ plans, const uint16_t* __restrict__ cls,
                                                         const double* __restrict__ dict, int S, int C,
                                                         const double* __restrict__ x, double* __restrict__ y, int map_xcd) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef double v2du __attribute__((ext_vector_type(2), aligned(8)));
    extern __shared__ __attribute__((aligned(16))) double sdict[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double* __restrict__ wl = sdict + (int64_t)wave * C * S;
    constexpr int K = FS_DICT_ITEMS_PER_WAVE;
    const int64_t n_chunks = (n_items + 4 * K - 1) / (4 * K);
    const int32_t cmax = (int32_t)(n_cols - 1);
    chunk_iter it = xcd_chunks(n_chunks);
    if (!map_xcd) { it.cur = blockIdx.x; it.step = gridDim.x; it.end = n_chunks; }
    int tagv = -1, rr = 0;
    const unsigned long long cmask = C >= 64 ? ~0ull : ((1ull << C) - 1ull);
    for (; it.cur < it.end; it.cur += it.step)
    for (int kk = 0; kk < K; ++kk) {
        const int64_t q = (it.cur * 4 + wave) * K + kk;
        if (q >= n_items) break;
        const int4 ds = items[__builtin_amdgcn_readfirstlane((int)q)];
        const int32_t first = __builtin_amdgcn_readfirstlane(ds.x);
        const int nr = __builtin_amdgcn_readfirstlane(ds.y) & 0xffff;
        const int edge = __builtin_amdgcn_readfirstlane(ds.y) >> 16;
        const dict_plan_round* __restrict__ pl = plans + __builtin_amdgcn_readfirstlane(ds.z);
        const int rounds = __builtin_amdgcn_readfirstlane(ds.w);
        const int32_t r = first + 2 * lane;
        const bool ok0 = 2 * lane < nr, ok1 = 2 * lane + 1 < nr;
        const int c0 = ok0 ? (int)cls[r] : -1, c1 = ok1 ? (int)cls[r + 1] : -1;
        // ---- the classes of this item's nodes -> the wave's LDS region, unless they are there already (k_dict_spmv) ----
        int b0 = 0, b1 = 0;
        {
            unsigned long long inuse = 0ull;
            bool copied = false;
            unsigned long long m0 = __ballot(ok0), m1 = __ballot(ok1);
            while (m0 | m1) {
                const bool from0 = m0 != 0ull;
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)(from0 ? m0 : m1)) - 1);
                const int cv = __builtin_amdgcn_readlane(from0 ? c0 : c1, src);
                const unsigned long long hit = __ballot(tagv == cv) & cmask;
                int slot;
                if (hit) slot = __ffsll((long long)hit) - 1;
                else {
                    const unsigned long long freeb = ~inuse & cmask, ahead = freeb & ~((1ull << rr) - 1ull);
                    slot = __ffsll((long long)(ahead ? ahead : freeb)) - 1;
                    rr = slot + 1 == C ? 0 : slot + 1;
                    fs_wave_copy_pairs(wl + slot * S, dict + (int64_t)cv * S, S >> 1, lane);
                    if (lane == slot) tagv = cv;
                    copied = true;
                }
                inuse |= 1ull << slot;
                const bool h0 = c0 == cv, h1 = c1 == cv;
                if (h0) b0 = slot * S;
                if (h1) b1 = slot * S;
                m0 &= ~__ballot(h0);
                m1 &= ~__ballot(h1);
            }
            if (copied) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        const double* __restrict__ v0 = wl + b0;
        const double* __restrict__ v1 = wl + b1;
        double a0[3] = {0.0, 0.0, 0.0}, a1[3] = {0.0, 0.0, 0.0};
        for (int rd = 0; rd < rounds; ++rd) {
            const dict_plan_round* __restrict__ p = pl + rd;
#pragma unroll 1
            for (int h4 = 0; h4 < 8; h4 += FS_DICT3_RUNS) {
                v2d A[FS_DICT3_RUNS][3];
                if (!edge) {
#pragma unroll
                    for (int j = 0; j < FS_DICT3_RUNS; ++j) {
                        const double* __restrict__ xp = x + 3 * ((int64_t)r + p->start[h4 + j]);
                        A[j][0] = *reinterpret_cast<const v2du*>(xp);
                        A[j][1] = *reinterpret_cast<const v2du*>(xp + 2);
                        A[j][2] = *reinterpret_cast<const v2du*>(xp + 4);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < FS_DICT3_RUNS; ++j) {
                        int32_t ca = r + p->start[h4 + j], cb = ca + 1;
                        ca = ca < 0 ? 0 : (ca > cmax ? cmax : ca);
                        cb = cb < 0 ? 0 : (cb > cmax ? cmax : cb);
                        const double* __restrict__ pa = x + 3 * (int64_t)ca;
                        const double* __restrict__ pb = x + 3 * (int64_t)cb;
                        A[j][0].x = pa[0]; A[j][0].y = pa[1]; A[j][1].x = pa[2];
                        A[j][1].y = pb[0]; A[j][2].x = pb[1]; A[j][2].y = pb[2];
                    }
                }
#pragma unroll
                for (int j = 0; j < FS_DICT3_RUNS; ++j) {
                    const double e0[3] = {A[j][0].x, A[j][0].y, A[j][1].x}, e1[3] = {A[j][1].y, A[j][2].x, A[j][2].y};
                    const double* __restrict__ w0 = v0 + (8 * RL * rd + RL * (h4 + j)) * 9;
                    const double* __restrict__ w1 = v1 + (8 * RL * rd + RL * (h4 + j)) * 9;
                    // (a compiler barrier after every 3 x 3 block: left alone the compiler reads a run's 54 coefficients - and the next
                    // runs' - ahead of the first fma: 256 VGPRs and spills, one wave per SIMD, 219 us)
                    auto block_terms = [&](const double* __restrict__ cb, const double (&e)[3], double (&acc)[3]) {
#pragma unroll
                        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
                            for (int i = 0; i < 3; ++i) acc[i] = fma(cb[i * 3 + jj], e[jj], acc[i]);
                        asm volatile("" ::: "memory");
                    };
                    block_terms(w0, e0, a0);      block_terms(w1, e1, a1);
                    const double e2[3] = {fs_from_next_lane(e0[0]), fs_from_next_lane(e0[1]), fs_from_next_lane(e0[2])};
                    block_terms(w0 + 9, e1, a0);  block_terms(w1 + 9, e2, a1);
                    if (RL == 3) {
                        const double e3[3] = {fs_from_next_lane(e1[0]), fs_from_next_lane(e1[1]), fs_from_next_lane(e1[2])};
                        block_terms(w0 + 18, e2, a0); block_terms(w1 + 18, e3, a1);
                    }
                }
            }
        }
        double* __restrict__ yp = y + 3 * (int64_t)r;
        if (ok1) {
            v2d o0, o1, o2;
            o0.x = a0[0]; o0.y = a0[1]; o1.x = a0[2]; o1.y = a1[0]; o2.x = a1[1]; o2.y = a1[2];
            *reinterpret_cast<v2du*>(yp) = o0; *reinterpret_cast<v2du*>(yp + 2) = o1; *reinterpret_cast<v2du*>(yp + 4) = o2;
        } else if (ok0) { yp[0] = a0[0]; yp[1] = a0[1]; yp[2] = a0[2]; }
        __builtin_amdgcn_wave_barrier();
    }
}

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

// ---- CG scalar state on the device ---------------------------------------------------------
// sums[0..2] = gamma=r.z, delta=w.z, rho=r.r of the current iteration (globally reduced)
// ctrl[0] = threshold on rho (max(rtol^2 b.b, atol^2)), ctrl[1] = b.b, ctrl[2] = max_iter (read by the update kernels of a
// captured batch, so that one instantiated graph serves solves with different iteration limits)
// scal[2][2] = (gamma, alpha) of the previous iteration, double-buffered by iteration parity
// status[0] = 0 running / 1 converged / 2 breakdown / 3 max_iter, status[1] = iterations
__global__ void k_set_threshold(const double* __restrict__ bb, double rtol, double atol, double* __restrict__ ctrl) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double t = rtol * rtol * bb[0];
        ctrl[0] = fmax(t, atol * atol);
        ctrl[1] = bb[0];
    }
}
__global__ void k_set_iteration_limit(double* __restrict__ ctrl, int max_iter) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ctrl[2] = (double)max_iter;
}

// FUSED: every workgroup first sums the SpMV's per-WG partials itself (same fixed order as
// k_sum_partials, so all workgroups obtain bit-identical gamma/delta/rho) - saves one launch per
// iteration on a single GPU; with a communicator the sums come from the all-reduced `sums`.
// alpha, beta of the single-reduction recurrences from the reduced sums and the previous iteration's (gamma, alpha).  ONE definition
// for every kernel that needs them - the update kernels, the peer-to-peer exchange kernel that advances the ghost rows of r and
// s on its own, the pipelined recurrence: the ghost copies stay bit-identical to the owner's rows only if all of them apply the
// same bits.  false: not SPD / NaN.
__device__ __forceinline__ bool cg_scalars_from(int iter, double gamma, double delta, double rho, double gamma_old, double alpha_old,
                                                 double& alpha, double& beta) {
    beta = 0.0;
    if (iter == 0) {
        alpha = gamma / delta;
    } else {
        beta = gamma / gamma_old;
        alpha = gamma / (delta - beta * gamma / alpha_old);
    }
    return (alpha > 0.0) && (alpha < 1e300) && (rho == rho);
}
__device__ __forceinline__ bool cg_scalars(int iter, double gamma, double delta, double rho, const double* __restrict__ scal,
                                            double& alpha, double& beta) {
    double gamma_old = 0.0, alpha_old = 0.0;
    if (iter != 0) {
        gamma_old = scal[2 * ((iter - 1) & 1) + 0];
        alpha_old = scal[2 * ((iter - 1) & 1) + 1];
    }
    return cg_scalars_from(iter, gamma, delta, rho, gamma_old, alpha_old, alpha, beta);
}

template <bool FUSED>
__global__ void __launch_bounds__(FS_BLOCK) k_cg_update(int64_t n, int iter, int check_only,
                                                        const double* __restrict__ partials, int npart,
                                                        const double* __restrict__ sums,
                                                        const double* __restrict__ ctrl, double* __restrict__ scal,
                                                        int* __restrict__ status, double* __restrict__ hist,
                                                        const double* __restrict__ dinv, double* z,
                                                        const double* __restrict__ w, double* __restrict__ p,
                                                        double* __restrict__ sv, double* __restrict__ x,
                                                        double* __restrict__ r) {
    if (status[0] != 0) return;
    double gamma, delta, rho;
    if (FUSED) {
        __shared__ double lds4[4];
        __shared__ double sh[3];
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int i = threadIdx.x; i < npart; i += FS_BLOCK) {
            a0 += partials[i];
            a1 += partials[npart + i];
            a2 += partials[2 * npart + i];
        }
        const double t0 = fs_block_sum(a0, lds4);
        const double t1 = fs_block_sum(a1, lds4);
        const double t2 = fs_block_sum(a2, lds4);
        if (threadIdx.x == 0) { sh[0] = t0; sh[1] = t1; sh[2] = t2; }
        __syncthreads();
        gamma = sh[0]; delta = sh[1]; rho = sh[2];
    } else {
        gamma = sums[0]; delta = sums[1]; rho = sums[2];
    }
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    if (leader) hist[iter] = rho;
    if (rho <= ctrl[0]) {  // every workgroup takes the same branch: inputs are identical
        if (leader) { status[1] = iter; status[0] = 1; }
        return;
    }
    if (check_only) {
        if (leader) { status[1] = iter; status[0] = 3; }
        return;
    }
    double beta, alpha;
    if (!cg_scalars(iter, gamma, delta, rho, scal, alpha, beta)) {  // not SPD / NaN
        if (leader) { status[1] = iter; status[0] = 2; }
        return;
    }
    if (leader) {
        scal[2 * (iter & 1) + 0] = gamma;
        scal[2 * (iter & 1) + 1] = alpha;
    }
    // 16-B vectorised body; n even part as double2, tail scalar
    const int64_t n2 = n >> 1;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double2* z2 = reinterpret_cast<double2*>(z);  // read and written in place: no restrict
    const double2* __restrict__ w2 = reinterpret_cast<const double2*>(w);
    const double2* __restrict__ d2 = reinterpret_cast<const double2*>(dinv);
    double2* __restrict__ p2 = reinterpret_cast<double2*>(p);
    double2* __restrict__ s2 = reinterpret_cast<double2*>(sv);
    double2* __restrict__ x2 = reinterpret_cast<double2*>(x);
    double2* __restrict__ r2 = reinterpret_cast<double2*>(r);
    for (; i < n2; i += stride) {
        const double2 zz = z2[i], ww = w2[i], dd = d2[i];
        double2 pp = p2[i], ss = s2[i], xx = x2[i], rr = r2[i];
        pp.x = zz.x + beta * pp.x;  pp.y = zz.y + beta * pp.y;
        ss.x = ww.x + beta * ss.x;  ss.y = ww.y + beta * ss.y;
        xx.x += alpha * pp.x;       xx.y += alpha * pp.y;
        rr.x -= alpha * ss.x;       rr.y -= alpha * ss.y;
        p2[i] = pp; s2[i] = ss; x2[i] = xx; r2[i] = rr;
        double2 zn; zn.x = dd.x * rr.x; zn.y = dd.y * rr.y;
        z2[i] = zn;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t t = n - 1;
        const double pp = z[t] + beta * p[t];
        const double ss = w[t] + beta * sv[t];
        p[t] = pp; sv[t] = ss;
        x[t] += alpha * pp;
        const double rr = r[t] - alpha * ss;
        r[t] = rr;
        z[t] = dinv[t] * rr;
    }
}

// ---- BiCGStab (non-symmetric operators: advection, ScalarTransportSolver.py:305-311) --------------
// Right-preconditioned (Jacobi) BiCGStab, PETSc KSPBCGS.  Per iteration: two fused SpMV+dots launches and
// three fused vector kernels; rho/alpha/omega, the threshold and the status word live on the device,
// exactly as for CG.  bscal = {rho[parity 0], rho[parity 1], alpha, omega}.
template <int NS>
__device__ __forceinline__ void wg_sum_partials(const double* __restrict__ partials, int npart, double (&out)[NS]) {
    __shared__ double lds4[4];
    __shared__ double sh[NS];
    double a[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) a[j] = 0.0;
    // four trips at a time, their 4 NS loads in flight together (the plain loop waits for every trip's loads before the next
    // trip's go out: four dependent round trips for 1024 partials, and the partials were written by the previous launch on other
    // XCDs - about 0.7 us each); a thread's partials are still added in ascending index: same bits
    for (int i0 = threadIdx.x; i0 < npart; i0 += 4 * FS_BLOCK) {
        double v[4][NS];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = i0 + t * FS_BLOCK;
#pragma unroll
            for (int j = 0; j < NS; ++j) v[t][j] = partials[(int64_t)j * npart + (i < npart ? i : npart - 1)];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (i0 + t * FS_BLOCK < npart) {
#pragma unroll
                for (int j = 0; j < NS; ++j) a[j] += v[t][j];
            }
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const double t = fs_block_sum(a[j], lds4);
        if (threadIdx.x == 0) sh[j] = t;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NS; ++j) out[j] = sh[j];
}

static __global__ void __launch_bounds__(FS_BLOCK) k_dot2_partial(const double* __restrict__ a, const double* __restrict__ b,
                                                                  const double* __restrict__ c, const double* __restrict__ d,
                                                                  int64_t n, double* __restrict__ partial) {
    __shared__ double lds4[4];
    double s0 = 0.0, s1 = 0.0;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        s0 += a[i] * b[i];
        s1 += c[i] * d[i];
    }
    const double t0 = fs_block_sum(s0, lds4);
    const double t1 = fs_block_sum(s1, lds4);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = t0;
        partial[gridDim.x + blockIdx.x] = t1;
    }
}

// K1: (rho_new = rhat.r, rr = r.r) -> convergence test, beta, p = r + beta (p - omega v), y = D^-1 p
template <bool FUSED>
__global__ void __launch_bounds__(FS_BLOCK) k_bicg_p(int64_t n, int iter, int check_only,
                                                     const double* __restrict__ partials, int npart,
                                                     const double* __restrict__ sums, const double* __restrict__ ctrl,
                                                     double* __restrict__ bscal, int* __restrict__ status,
                                                     double* __restrict__ hist, const double* __restrict__ dinv,
                                                     const double* __restrict__ r, double* __restrict__ p,
                                                     const double* __restrict__ v, double* __restrict__ y) {
    if (status[0] != 0) return;
    double sm[2];
    if (FUSED) wg_sum_partials<2>(partials, npart, sm);
    else { sm[0] = sums[0]; sm[1] = sums[1]; }
    const double rho_new = sm[0], rr = sm[1];
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    if (leader) hist[iter] = rr;
    if (rr <= ctrl[0]) {
        if (leader) { status[1] = iter; status[0] = 1; }
        return;
    }
    if (check_only) {
        if (leader) { status[1] = iter; status[0] = 3; }
        return;
    }
    double beta = 0.0, omega = 0.0;
    if (iter > 0) {
        const double rho_old = bscal[(iter - 1) & 1];
        const double alpha = bscal[2];
        omega = bscal[3];
        beta = (rho_new / rho_old) * (alpha / omega);
    }
    if (!(rho_new == rho_new) || rho_new == 0.0 || !(beta == beta) || !(fabs(beta) < 1e300)) {
        if (leader) { status[1] = iter; status[0] = 2; }
        return;
    }
    if (leader) bscal[iter & 1] = rho_new;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const double pn = iter == 0 ? r[i] : r[i] + beta * (p[i] - omega * v[i]);
        p[i] = pn;
        y[i] = dinv[i] * pn;
    }
}

// K3: alpha = rho / (rhat.v); s = r - alpha v; z = D^-1 s
template <bool FUSED>
__global__ void __launch_bounds__(FS_BLOCK) k_bicg_s(int64_t n, int iter, const double* __restrict__ partials,
                                                     int npart, const double* __restrict__ sums,
                                                     double* __restrict__ bscal, int* __restrict__ status,
                                                     const double* __restrict__ dinv, const double* __restrict__ r,
                                                     const double* __restrict__ v, double* __restrict__ sv,
                                                     double* __restrict__ z) {
    if (status[0] != 0) return;
    double sm[1];
    if (FUSED) wg_sum_partials<1>(partials, npart, sm);
    else sm[0] = sums[0];
    const double rv = sm[0];
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    const double alpha = bscal[iter & 1] / rv;
    if (rv == 0.0 || !(alpha == alpha) || !(fabs(alpha) < 1e300)) {
        if (leader) { status[1] = iter; status[0] = 2; }
        return;
    }
    if (leader) bscal[2] = alpha;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const double si = r[i] - alpha * v[i];
        sv[i] = si;
        z[i] = dinv[i] * si;
    }
}

// K5: omega = (t.s)/(t.t); x += alpha y + omega z; r = s - omega t; partial dots (rhat.r, r.r)
template <bool FUSED>
__global__ void __launch_bounds__(FS_BLOCK) k_bicg_x(int64_t n, int iter, const double* __restrict__ partials,
                                                     int npart, const double* __restrict__ sums,
                                                     double* __restrict__ bscal, int* __restrict__ status,
                                                     double* __restrict__ x, const double* __restrict__ y,
                                                     const double* __restrict__ z, double* __restrict__ r,
                                                     const double* __restrict__ sv, const double* __restrict__ t,
                                                     const double* __restrict__ rhat, double* __restrict__ pout) {
    if (status[0] != 0) return;
    double sm[2];
    if (FUSED) wg_sum_partials<2>(partials, npart, sm);
    else { sm[0] = sums[0]; sm[1] = sums[1]; }
    const double ts = sm[0], tt = sm[1];
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    const double omega = tt > 0.0 ? ts / tt : 0.0;   // t = 0 only when s = 0: x + alpha y is already exact
    const double alpha = bscal[2];
    if (!(omega == omega)) {
        if (leader) { status[1] = iter; status[0] = 2; }
        return;
    }
    if (leader) bscal[3] = omega;
    __shared__ double lds4b[4];
    double d0 = 0.0, d1 = 0.0;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        x[i] += alpha * y[i] + omega * z[i];
        const double ri = sv[i] - omega * t[i];
        r[i] = ri;
        d0 += rhat[i] * ri;
        d1 += ri * ri;
    }
    const double t0 = fs_block_sum(d0, lds4b);
    const double t1 = fs_block_sum(d1, lds4b);
    if (threadIdx.x == 0) {
        pout[blockIdx.x] = t0;
        pout[gridDim.x + blockIdx.x] = t1;
    }
}

// ---- CG on the symmetrically scaled system  D^-1/2 A D^-1/2 (PETSc KSPSetDiagonalScale) ------------
// Jacobi-PCG on A is unpreconditioned CG on the scaled operator, for which z == r: the update kernel
// drops the z and D^-1 streams (72 instead of 96 B/DOF per iteration).  rho stays the UNSCALED ||r||^2
// (sum d r^2, computed in the SpMV), so the stopping test is unchanged.
template <bool FUSED, bool NT = false>
__global__ void __launch_bounds__(FS_BLOCK) k_cg_update_scaled(int64_t n, int iter, int check_only,
                                                               const double* __restrict__ partials, int npart,
                                                               const double* __restrict__ sums,
                                                               const double* __restrict__ ctrl,
                                                               double* __restrict__ scal, int* __restrict__ status,
                                                               double* __restrict__ hist, double* __restrict__ r,
                                                               const double* __restrict__ w, double* __restrict__ p,
                                                               double* __restrict__ sv, double* __restrict__ x, int* __restrict__ mirror = nullptr, int r_plain = 0) {
    // mirror (one GPU; may be null): the pinned progress words of k_dict_cg_iter - [0] status once stopped, [1] iteration in progress
    // (the status word is looked at AFTER the first trip's loads have gone out - everything a launch reads first was written by the
    // previous launch on other XCDs, and each dependent load is a round trip of about a microsecond)
    const int st0 = status[0];
    const int st2 = status[2];
    const double it_max = ctrl[2], thresh = ctrl[0];
    const double sc_g0 = scal[0], sc_a0 = scal[1], sc_g1 = scal[2], sc_a1 = scal[3];      // (gamma, alpha) of the two iteration parities
    // NT (vectors larger than the caches, 10 M DOF): non-temporal loads and stores - the probe in tools/probes streams
    // 7.2 instead of 6.4 TB/s that way
    typedef double v2d __attribute__((ext_vector_type(2)));
    struct io {
        static __device__ __forceinline__ double2 ld(const double2* q) {
            if (NT) { const v2d t = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(q)); return make_double2(t.x, t.y); }
            return *q;
        }
        static __device__ __forceinline__ void st(double2* q, const double2& v) {
            if (NT) { v2d t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, reinterpret_cast<v2d*>(q)); }
            else *q = v;
        }
    };
    const int64_t n2 = n >> 1;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double2* __restrict__ w2 = reinterpret_cast<const double2*>(w);
    double2* __restrict__ p2 = reinterpret_cast<double2*>(p);
    double2* __restrict__ s2 = reinterpret_cast<double2*>(sv);
    double2* __restrict__ x2 = reinterpret_cast<double2*>(x);
    double2* __restrict__ r2 = reinterpret_cast<double2*>(r);
    // the operands of the first trip are requested BEFORE the sums are reduced (at 1 M rows a thread makes two trips and the
    // kernel is latency-bound: the reduction of 3 x 1024 partials and the ten loads of the trip used to be two round trips in a row)
    const bool first_trip = i + stride < n2;
    double2 wa0 = {0.0, 0.0}, wb0 = wa0, pa0 = wa0, sa0 = wa0, xa0 = wa0, ra0 = wa0, pb0 = wa0, sb0 = wa0, xb0 = wa0, rb0 = wa0;
    if (first_trip) {
        const int64_t j = i + stride;
        wa0 = io::ld(&w2[i]); wb0 = io::ld(&w2[j]);
        pa0 = io::ld(&p2[i]); sa0 = io::ld(&s2[i]); xa0 = io::ld(&x2[i]); ra0 = io::ld(&r2[i]);
        pb0 = io::ld(&p2[j]); sb0 = io::ld(&s2[j]); xb0 = io::ld(&x2[j]); rb0 = io::ld(&r2[j]);
    }
    if (st0 != 0) return;
    if (iter < 0) {                 // captured batch: the index of the product that preceded this launch
        iter = st2 - 1;
        check_only = iter >= (int)it_max ? 1 : 0;
    }
    double gamma, delta, rho;
    if (FUSED) {
        double sm[3];
        wg_sum_partials<3>(partials, npart, sm);
        gamma = sm[0]; delta = sm[1]; rho = sm[2];
    } else {
        gamma = sums[0]; delta = sums[1]; rho = sums[2];
    }
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    if (leader) hist[iter] = rho;
    auto stop = [&](int code) {
        if (leader) {
            status[1] = iter; status[0] = code;
            if (mirror) { fs_host_store(mirror + 1, iter); fs_host_store(mirror, code); }
        }
    };
    if (rho <= thresh) { stop(1); return; }
    if (check_only) { stop(3); return; }
    double beta, alpha;
    if (!cg_scalars_from(iter, gamma, delta, rho, ((iter - 1) & 1) ? sc_g1 : sc_g0, ((iter - 1) & 1) ? sc_a1 : sc_a0, alpha, beta)) { stop(2); return; }
    if (leader) {
        scal[2 * (iter & 1) + 0] = gamma;
        scal[2 * (iter & 1) + 1] = alpha;
        if (mirror) fs_host_store(mirror + 1, iter + 1);
    }
    // two strided elements per trip: ten 16-B loads in flight per lane (the kernel is latency-bound at 1 M DOF)
    bool prefetched = first_trip;
    for (; i + stride < n2; i += 2 * stride) {
        const int64_t j = i + stride;
        double2 wa, wb, pa, sa, xa, ra, pb, sb, xb, rb;
        if (prefetched) {
            wa = wa0; wb = wb0; pa = pa0; sa = sa0; xa = xa0; ra = ra0; pb = pb0; sb = sb0; xb = xb0; rb = rb0;
            prefetched = false;
        } else {
            wa = io::ld(&w2[i]); wb = io::ld(&w2[j]);
            pa = io::ld(&p2[i]); sa = io::ld(&s2[i]); xa = io::ld(&x2[i]); ra = io::ld(&r2[i]);
            pb = io::ld(&p2[j]); sb = io::ld(&s2[j]); xb = io::ld(&x2[j]); rb = io::ld(&r2[j]);
        }
        pa.x = ra.x + beta * pa.x;  pa.y = ra.y + beta * pa.y;
        pb.x = rb.x + beta * pb.x;  pb.y = rb.y + beta * pb.y;
        sa.x = wa.x + beta * sa.x;  sa.y = wa.y + beta * sa.y;
        sb.x = wb.x + beta * sb.x;  sb.y = wb.y + beta * sb.y;
        xa.x += alpha * pa.x;       xa.y += alpha * pa.y;
        xb.x += alpha * pb.x;       xb.y += alpha * pb.y;
        ra.x -= alpha * sa.x;       ra.y -= alpha * sa.y;
        rb.x -= alpha * sb.x;       rb.y -= alpha * sb.y;
        // (r_plain, bits: ordinary stores for r / p / s / x where the kernel otherwise streams its stores past the caches; g_upd_r_plain)
        if (r_plain & 2) { p2[i] = pa; p2[j] = pb; } else { io::st(&p2[i], pa); io::st(&p2[j], pb); }
        if (r_plain & 4) { s2[i] = sa; s2[j] = sb; } else { io::st(&s2[i], sa); io::st(&s2[j], sb); }
        if (r_plain & 8) { x2[i] = xa; x2[j] = xb; } else { io::st(&x2[i], xa); io::st(&x2[j], xb); }
        if (r_plain & 1) { r2[i] = ra; r2[j] = rb; } else { io::st(&r2[i], ra); io::st(&r2[j], rb); }
    }
    for (; i < n2; i += stride) {
        const double2 ww = w2[i];
        double2 pp = p2[i], ss = s2[i], xx = x2[i], rr = r2[i];
        pp.x = rr.x + beta * pp.x;  pp.y = rr.y + beta * pp.y;
        ss.x = ww.x + beta * ss.x;  ss.y = ww.y + beta * ss.y;
        xx.x += alpha * pp.x;       xx.y += alpha * pp.y;
        rr.x -= alpha * ss.x;       rr.y -= alpha * ss.y;
        p2[i] = pp; s2[i] = ss; x2[i] = xx; r2[i] = rr;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t t = n - 1;
        const double pp = r[t] + beta * p[t];
        const double ss = w[t] + beta * sv[t];
        p[t] = pp; sv[t] = ss;
        x[t] += alpha * pp;
        r[t] -= alpha * ss;
    }
}

// ---- ONE launch per CG iteration (row-dictionary operators, one GPU) ------------------------------------------------------------
// Launch k does the update of iteration k AND the product of iteration k + 1:
//      s_k = w_k + beta_k s_{k-1};   r_{k+1} = r_k - alpha_k s_k;   p_k = r_k + beta_k p_{k-1};   x_{k+1} = x_k + alpha_k p_k
//      w_{k+1} = A r_{k+1}   with the three sums (r.r, w.r, sum d r^2) of the new pair
// The product needs r_{k+1} on the NEIGHBOUR rows, which other workgroups own: instead of waiting for them (a device-wide barrier
// costs 8 us on the 8 XCDs of gfx950, DESIGN history) every lane recomputes r_{k+1}[j] = r_k[j] - alpha (w_k[j] + beta s_{k-1}[j])
// for the columns its runs touch - the same two fmas the owner applies, hence the same bits - from the OLD r, w, s, which are
// read-only during the launch: r, w, s are double-buffered by iteration parity, p and x are row-local and stay in place.  alpha_k,
// beta_k come from the dot partials launch k - 1 left (summed by every workgroup, as k_cg_update_scaled does).
// Per row: reads r, w, s (neighbour values from L1 / L2), p, x, d, the 2-byte class; writes r, w, s, p, x = 90 B instead of the
// 98 B and two launches of k_dict_spmv + k_cg_update_scaled; 3 x 3.5 vector loads per row for the runs instead of 3.5.
// The iteration number travels on the device (it_ctr[par] read by everybody, it_ctr[par ^ 1] = iter + 1 written by one lane) so
// that a captured batch replays with constant arguments; par = parity of the launch = which buffers are `old`.
// Whole dictionary in LDS only (P1 boxes: the operators whose iteration is launch-bound).
// Measured on MI355X (tools/probes/fused_iter_probe.py, hipGraph batches): 1 M rows: 27.4 us per iteration against 29.9 us for
// the two launches, iterates BIT-IDENTICAL (the same fmas on the same operands, the same partial-sum geometry); 10 M rows: 285 us
// against 187 us - with ONE neighbour vector instead of three (timing ablation, wrong numerics) still 220 us: the row-local
// streams (5 loads, 5 stores per row) reach 4.1 TB/s inside an item-by-item kernel against the 6.5 TB/s of the grid-stride update
// kernel, and three vectors x three mesh planes no longer fit the 4 MB L2 of an XCD.  Hence automatic only where the vectors
// stay in the Infinity Cache (FS_CG_FUSED_MAX_ROWS, default 3 M rows).
#ifdef FS_ITER_TIMING
__device__ long long g_iter_dbg[8 * 4096];
#define FS_STAMP(k) do { if (threadIdx.x == 0 && it_ctr[par] == 100) g_iter_dbg[8 * blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define FS_STAMP(k) do { } while (0)
#endif
// COMM: a decomposed space - the three sums come reduced over the ranks from `sums` (k_cg_p2p_exchange<true> before this launch put
// them there, stored the neighbours' w into the ghost rows of w_in and advanced the ghost rows of r_out / s_out); the neighbour
// columns of an item may then be ghost columns: n_cols counts them.
template <int RL, bool COMM>
__global__ void __launch_bounds__(FS_BLOCK) k_dict_cg_iter(int64_t n_cols, int64_t n_items, const int4* __restrict__ items,
                                                           const dict_plan_round* __restrict__ plans, const uint16_t* __restrict__ cls,
                                                           const double* __restrict__ dict, int S, int C,
                                                           const double* __restrict__ r_in, const double* __restrict__ w_in,
                                                           const double* __restrict__ s_in, double* __restrict__ r_out,
                                                           double* __restrict__ w_out, double* __restrict__ s_out,
                                                           double* __restrict__ pv, double* __restrict__ xv, const double* __restrict__ dvec,
                                                           const double* __restrict__ part_in, double* __restrict__ part_out, int npart,
                                                           const double* __restrict__ ctrl, double* __restrict__ scal, int* __restrict__ status,
                                                           int* __restrict__ it_ctr, int par, double* __restrict__ hist, int map_xcd,
                                                           const double* __restrict__ sums, int* __restrict__ mirror) {
    // mirror (may be null): two words of PINNED HOST memory - [0] the status word once the recurrence has stopped, [1] the number of
    // the iteration this launch is working on - written by one lane with relaxed system-scope stores (no fence: nothing is ordered
    // against them).  The host enqueues the next launches from what it reads there (fs_krylov_solve) instead of copying the status
    // word back behind every batch, and stops within a few launches of the end instead of a batch and a half after it.
    FS_STAMP(0);
    // The launch is latency-bound at the sizes it is used for (a wave has two work items at 1 M rows), and what it reads first was
    // written by the previous launch on other XCDs - every dependent load is a round trip to the Infinity Cache (1 - 2 us).  So
    // everything the prologue needs goes out BEFORE anything is waited for: status word, iteration number, threshold, both parities
    // of the previous (gamma, alpha), the twelve dot partials of this thread, the first item's header -> run starts -> first
    // twelve run loads, the dictionary.  (Timeline of the first version, tools/probes: 8.8 of 22.7 us were the prologue.)
    const int st0 = status[0];
    const int iter = it_ctr[par];
    const double thresh = ctrl[0], it_max = ctrl[2];
    const double sc_g0 = scal[0], sc_a0 = scal[1], sc_g1 = scal[2], sc_a1 = scal[3];
    double pl[3][4];
    double sum_g = 0.0, sum_d = 0.0, sum_r = 0.0;
    if (COMM) { sum_g = sums[0]; sum_d = sums[1]; sum_r = sums[2]; }
    else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = threadIdx.x + t * FS_BLOCK;
#pragma unroll
            for (int j = 0; j < 3; ++j) pl[j][t] = part_in[(int64_t)j * npart + (i < npart ? i : npart - 1)];      // (no branch: twelve loads in flight)
        }
    }
    FS_STAMP(1);
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef double v2du __attribute__((ext_vector_type(2), aligned(8)));
    extern __shared__ __attribute__((aligned(16))) double sdict[];
    __shared__ double lds34[3][4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t n_chunks = (n_items + 3) / 4;
    const int32_t cmax = (int32_t)(n_cols - 1);
    chunk_iter it = xcd_chunks(n_chunks);
    if (!map_xcd) { it.cur = blockIdx.x; it.step = gridDim.x; it.end = n_chunks; }
    struct item_hdr { int32_t first; int nr, edge, rounds; const dict_plan_round* pl; };
    auto decode = [&](int64_t q) {
        const int4 ds = items[__builtin_amdgcn_readfirstlane((int)q)];
        item_hdr H;
        H.first = __builtin_amdgcn_readfirstlane(ds.x);
        H.nr = __builtin_amdgcn_readfirstlane(ds.y) & 0xffff;
        H.edge = __builtin_amdgcn_readfirstlane(ds.y) >> 16;
        H.pl = plans + __builtin_amdgcn_readfirstlane(ds.z);
        H.rounds = __builtin_amdgcn_readfirstlane(ds.w);
        return H;
    };
    // loads of runs [4 h, 4 h + 4) of round rd: scalar base (vector + run start) + one 32-bit byte offset per lane - the `saddr` form
    // of the load, no 64-bit address arithmetic or address registers per load (rows < 2^29)
    auto load_half = [&](const item_hdr& H, int rd, int h, uint32_t boff, v2d (&A)[4], v2d (&Wb)[4], v2d (&Sb)[4]) {
        // (measured and not kept: run starts that need no load - kernel arguments for the plan most items have, or a per-item copy beside
        // the header - and the next item's header asked for an item ahead: with nothing left between header and run loads the
        // compiler schedules 164 - 177 VGPRs (and spills SGPRs into them), two or three waves per SIMD instead of four, and the
        // launch gets slower: 22.9 against 19.0 us at 1 M rows)
        int32_t sts[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sts[j] = __builtin_amdgcn_readfirstlane(H.pl[rd].start[4 * h + j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t st = sts[j];
            A[j] = *reinterpret_cast<const v2du*>(reinterpret_cast<const char*>(r_in + st) + boff);
            Wb[j] = *reinterpret_cast<const v2du*>(reinterpret_cast<const char*>(w_in + st) + boff);
            Sb[j] = *reinterpret_cast<const v2du*>(reinterpret_cast<const char*>(s_in + st) + boff);
        }
    };
    auto load_own = [&](const item_hdr& H, v2d& pp, v2d& xx, v2d& dd) {
        const int32_t r = H.first + 2 * lane;
        pp = xx = dd = v2d{0.0, 0.0};
        if (2 * lane + 1 < H.nr) {
            pp = *reinterpret_cast<const v2du*>(&pv[r]);
            xx = *reinterpret_cast<const v2du*>(&xv[r]);
            dd = *reinterpret_cast<const v2du*>(&dvec[r]);
        } else if (2 * lane < H.nr) { pp.x = pv[r]; xx.x = xv[r]; dd.x = dvec[r]; }
    };
    // Nothing the first item LOADS depends on alpha, beta: its first twelve run loads and its row-local operands are asked for
    // before the partial sums are reduced (a wave has about two items at 1 M rows; 35.3 -> 27.4 us per iteration with 1024 workgroups)
    item_hdr H0 = {};
    v2d PA[4], PW[4], PS[4], Ppp = {0.0, 0.0}, Pxx = {0.0, 0.0}, Pdd = {0.0, 0.0};
    const bool have0 = it.cur < it.end && it.cur * 4 + wave < n_items;
    if (have0) {
        H0 = decode(it.cur * 4 + wave);
        if (!H0.edge) load_half(H0, 0, 0, (uint32_t)(H0.first + 2 * lane) * 8u, PA, PW, PS);
        load_own(H0, Ppp, Pxx, Pdd);
    }
    FS_STAMP(2);
    fs_fill_lds(sdict, dict, C * S);             // (visible after the barrier of the sum below)
    if (st0 != 0) return;
    // the three sums of the previous launch's partials, in the order of wg_sum_partials / fs_block_sum (same bits as the update
    // kernel of the two-launch iteration computes), with ONE barrier: a thread's partials in ascending index, the wave's by the
    // shuffle tree, the four waves as (0 + 1) + (2 + 3) by every thread
    double sm[3];
    if (COMM) {
        sm[0] = sum_g; sm[1] = sum_d; sm[2] = sum_r;
        __syncthreads();                        // (the dictionary is in LDS)
    } else {                                    // (npart <= 4 x 256: the host launches this kernel with at most 1024 workgroups)
        double a[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if ((int)threadIdx.x + t * FS_BLOCK < npart) {
#pragma unroll
                for (int j = 0; j < 3; ++j) a[j] += pl[j][t];
            }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) a[j] += __shfl_down(a[j], off, 64);
            if (lane == 0) lds34[j][wave] = a[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 3; ++j) sm[j] = (lds34[j][0] + lds34[j][1]) + (lds34[j][2] + lds34[j][3]);
        __syncthreads();                        // (lds34 is written again at the end)
    }
    FS_STAMP(3);
    const double gamma = sm[0], delta = sm[1], rho = sm[2];
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    if (leader) hist[iter] = rho;
    auto stop = [&](int code) {
        if (leader) {
            status[1] = iter; status[0] = code;
            if (mirror) { fs_host_store(mirror + 1, iter); fs_host_store(mirror, code); }
        }
    };
    if (rho <= thresh) { stop(1); return; }     // every workgroup takes the same branch: the inputs are identical
    if (iter >= (int)it_max) { stop(3); return; }
    double beta, alpha;
    if (!cg_scalars_from(iter, gamma, delta, rho, ((iter - 1) & 1) ? sc_g1 : sc_g0, ((iter - 1) & 1) ? sc_a1 : sc_a0, alpha, beta)) { stop(2); return; }
    if (leader) {
        scal[2 * (iter & 1) + 0] = gamma;
        scal[2 * (iter & 1) + 1] = alpha;
        it_ctr[par ^ 1] = iter + 1;
        if (mirror) fs_host_store(mirror + 1, iter + 1);
    }
    const double nalpha = -alpha;
    FS_STAMP(4);
    double d_rz = 0.0, d_wz = 0.0, d_rr = 0.0;
    // PRE: this item's first half round and row-local operands were loaded before the prologue
    auto do_item = [&](const item_hdr& H, auto pre_tag) {
        constexpr bool PRE = decltype(pre_tag)::value;
        const int32_t r = H.first + 2 * lane;
        const bool ok0 = 2 * lane < H.nr, ok1 = 2 * lane + 1 < H.nr;
        int c0 = -1, c1 = -1;
        if ((H.first & 1) == 0) {
            const uint32_t two = *reinterpret_cast<const uint32_t*>(&cls[r]);
            if (ok0) c0 = (int)(two & 0xffffu);
            if (ok1) c1 = (int)(two >> 16);
        } else {
            if (ok0) c0 = (int)cls[r];
            if (ok1) c1 = (int)cls[r + 1];
        }
        v2d sn0 = {0.0, 0.0}, ro0 = {0.0, 0.0};
        v2d pp, xx, dd;
        if (PRE) { pp = Ppp; xx = Pxx; dd = Pdd; }
        else load_own(H, pp, xx, dd);
        const double* __restrict__ v0 = sdict + (ok0 ? c0 * S : 0);
        const double* __restrict__ v1 = sdict + (ok1 ? c1 * S : 0);
        double a0 = 0.0, a1 = 0.0;
        v2d zi = {0.0, 0.0};
        // one run: its terms in ascending offsets, one fma each (the order and the bits of k_dict_spmv)
        auto run_terms = [&](v2d rn, const double* __restrict__ w0, const double* __restrict__ w1) {
            const double e0 = rn.x, e1 = rn.y;
            const double e2 = fs_from_next_lane(rn.x);
            a0 = fma(w0[0], e0, a0); a1 = fma(w1[0], e1, a1);
            a0 = fma(w0[1], e1, a0); a1 = fma(w1[1], e2, a1);
            if (RL == 3) {
                const double e3 = fs_from_next_lane(rn.y);
                a0 = fma(w0[2], e2, a0); a1 = fma(w1[2], e3, a1);
            }
        };
        if (!H.edge) {
            const uint32_t boff = (uint32_t)r * 8u;
            for (int rd = 0; rd < H.rounds; ++rd) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v2d A[4], Wb[4], Sb[4];
                    if (PRE && h == 0) {
                        if (rd == 0) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) { A[j] = PA[j]; Wb[j] = PW[j]; Sb[j] = PS[j]; }
                        } else load_half(H, rd, h, boff, A, Wb, Sb);
                    } else load_half(H, rd, h, boff, A, Wb, Sb);
                    // the new r on these columns: r - alpha (w + beta s), two fmas per value - the owner's operations, the owner's bits
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v2d sn;
                        sn.x = fma(beta, Sb[j].x, Wb[j].x); sn.y = fma(beta, Sb[j].y, Wb[j].y);
                        if (h == 0 && j == 0 && rd == 0) { sn0 = sn; ro0 = A[0]; }
                        A[j].x = fma(nalpha, sn.x, A[j].x); A[j].y = fma(nalpha, sn.y, A[j].y);
                    }
                    if (h == 0 && rd == 0) zi = A[0];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        run_terms(A[j], v0 + RL * (8 * rd + 4 * h + j), v1 + RL * (8 * rd + 4 * h + j));
                        asm volatile("" ::: "memory");
                    }
                }
            }
        } else {
            // items whose loads could leave [0, n_cols) (first / last mesh plane; 1.4 % of the rows at 10 M): run by run, each value
            // clamped into the vector (a column outside it has a zero coefficient)
            const int n_runs = 8 * H.rounds;
#pragma unroll 1
            for (int g = 0; g < n_runs; ++g) {
                const int32_t c = r + __builtin_amdgcn_readfirstlane(H.pl[g >> 3].start[g & 7]);
                const int32_t ca = c < 0 ? 0 : (c > cmax ? cmax : c), cb = c + 1 < 0 ? 0 : (c + 1 > cmax ? cmax : c + 1);
                v2d rn, sn;
                sn.x = fma(beta, s_in[ca], w_in[ca]); sn.y = fma(beta, s_in[cb], w_in[cb]);
                const double ra = r_in[ca], rb = r_in[cb];
                if (g == 0) { sn0 = sn; ro0.x = ra; ro0.y = rb; }
                rn.x = fma(nalpha, sn.x, ra); rn.y = fma(nalpha, sn.y, rb);
                if (g == 0) zi = rn;
                run_terms(rn, v0 + RL * g, v1 + RL * g);
            }
        }
        // own rows: p, x, and the new s, r, w
        v2d pn, xn;
        pn.x = fma(beta, pp.x, ro0.x); pn.y = fma(beta, pp.y, ro0.y);
        xn.x = fma(alpha, pn.x, xx.x); xn.y = fma(alpha, pn.y, xx.y);
        if (ok1) {
            v2d out;
            out.x = a0; out.y = a1;
            *reinterpret_cast<v2du*>(&w_out[r]) = out;
            *reinterpret_cast<v2du*>(&r_out[r]) = zi;
            *reinterpret_cast<v2du*>(&s_out[r]) = sn0;
            *reinterpret_cast<v2du*>(&pv[r]) = pn;
            *reinterpret_cast<v2du*>(&xv[r]) = xn;
        } else if (ok0) {
            w_out[r] = a0; r_out[r] = zi.x; s_out[r] = sn0.x; pv[r] = pn.x; xv[r] = xn.x;
        }
        if (!ok0) { a0 = 0.0; zi.x = 0.0; }
        if (!ok1) { a1 = 0.0; zi.y = 0.0; }
        d_rz += zi.x * zi.x + zi.y * zi.y; d_wz += a0 * zi.x + a1 * zi.y; d_rr += dd.x * zi.x * zi.x + dd.y * zi.y * zi.y;
    };
    if (have0) {
        do_item(H0, std::true_type{});
        it.cur += it.step;
    }
    FS_STAMP(5);
    for (; it.cur < it.end; it.cur += it.step) {
        const int64_t q = it.cur * 4 + wave;
        if (q >= n_items) break;
        do_item(decode(q), std::false_type{});
    }
    FS_STAMP(6);
    // this workgroup's three dot partials: fs_block_sum's order (shuffle tree, then (0 + 1) + (2 + 3)), one barrier for the three
    double dsum[3] = {d_rz, d_wz, d_rr};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dsum[j] += __shfl_down(dsum[j], off, 64);
        if (lane == 0) lds34[j][wave] = dsum[j];
    }
    __syncthreads();
    FS_STAMP(7);
    if (threadIdx.x < 3) part_out[(int64_t)threadIdx.x * npart + blockIdx.x] =
        (lds34[threadIdx.x][0] + lds34[threadIdx.x][1]) + (lds34[threadIdx.x][2] + lds34[threadIdx.x][3]);
}

// The same update on the rows [0, a) and [b, n) only - the rows a slab sends to its neighbours - so that the halo
// exchange of the new r can start before the bulk of the update and the interior product are even launched (several
// GPUs: sums from the all-reduce).  No side effects: status, history and the scalars are written by the launch on the
// remaining rows, which comes later in the stream and takes the same decisions from the same inputs.
__global__ void __launch_bounds__(FS_BLOCK) k_cg_update_scaled_rows(int64_t a, int64_t b, int64_t n, int iter, int check_only,
                                                                    const double* __restrict__ sums, const double* __restrict__ ctrl,
                                                                    const double* __restrict__ scal, const int* __restrict__ status,
                                                                    double* __restrict__ r, const double* __restrict__ w,
                                                                    double* __restrict__ p, double* __restrict__ sv, double* __restrict__ x) {
    if (status[0] != 0) return;
    const double gamma = sums[0], delta = sums[1], rho = sums[2];
    if (rho <= ctrl[0] || check_only) return;
    double beta, alpha;
    if (!cg_scalars(iter, gamma, delta, rho, scal, alpha, beta)) return;
    const int64_t total = a + (n - b);
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < total; t += stride) {
        const int64_t i = t < a ? t : b + (t - a);
        const double pp = r[i] + beta * p[i];
        const double ss = w[i] + beta * sv[i];
        p[i] = pp; sv[i] = ss;
        x[i] += alpha * pp;
        r[i] -= alpha * ss;
    }
}

// ---- the exchange kernel of the peer-to-peer iteration (fs_comm.hip: hipIpc-mapped buffers; protocol in fs_kernels.h) -----------
// Everything of a CG iteration that crosses GPUs, in ONE launch between the product and the update.  What travels is w = A r on
// the rows some neighbour needs - known as soon as the product is through, with no dependence on the sums - and NOT the new
// residual: a rank advances the ghost copies of r and s itself (s_g <- w_g + beta s_g, r_g <- r_g - alpha s_g are row-local and it
// knows alpha, beta), with the operations the owner's update kernel applies to the same rows.
//   1. every workgroup stores its share of w[send rows] straight into the neighbours' receive buffers, the last one through
//      publishes the sequence number of the exchange at every neighbour;
//   2. (meanwhile on the wire) every workgroup sums the product's dot partials itself (same bits everywhere); workgroup 0
//      stores the three sums into slot [me] of the other ranks' all-reduce buffers; every workgroup waits for the other ranks'
//      sums and adds all up in rank order; workgroup 0 leaves the result in `sums` for the update kernel that follows;
//   3. every workgroup waits for the neighbours' sequence numbers and advances its share of the ghost rows with the received w.
// (No workgroup waits before its own stores are counted, so two ranks never wait for each other.)  Any halo plan (send lists
// need not be contiguous, ghosts may be scattered), any block size: rows are dofs here.  Gated by the status word; steps 1 and
// 2 always run together, step 3 only if the iteration goes on - decided from the reduced sums, hence alike on every rank.
// PP: the exchange of the ONE-LAUNCH iteration on a decomposed space (k_dict_cg_iter<3, true> follows): r, w, s are double-buffered by
// iteration parity and that kernel recomputes the new residual on its neighbour columns - ghost columns included - from the OLD r, w,
// s.  So the received w goes into the ghost rows of the current w (pp.w_cur), and the ghost rows of s and r are advanced from the
// current buffers into the next ones (pp.s_cur -> pp.s_nxt, pp.r_cur -> pp.r_nxt): the two fmas of the owner.  The iteration number
// comes from it_ctr[par] (written by the previous iteration kernel).
struct fs_pp_ghosts {
    double* w_cur;
    const double* s_cur;
    double* s_nxt;
    const double* r_cur;
    double* r_nxt;
    const int* it_ctr;
    int par;
};
template <bool PP>
__global__ void __launch_bounds__(FS_BLOCK) k_cg_p2p_exchange(int iter, int check_only, const double* __restrict__ ctrl,
                                                              const double* __restrict__ scal, const int* __restrict__ status,
                                                              double* r, const double* __restrict__ w, double* __restrict__ s_ghost,
                                                              const fs_p2p_rowsred red, const fs_p2p_sendrows snd, const fs_pp_ghosts pp) {
    // (every scalar the launch needs is asked for before the first of them is looked at: each was written by the previous launch,
    // a dependent load is a round trip of about a microsecond, and this kernel sits on the critical path of every iteration)
    const int st0 = status[0], st2 = status[2];
    const double it_max = ctrl[2], thresh = ctrl[0];
    const double sc_g0 = scal[0], sc_a0 = scal[1], sc_g1 = scal[2], sc_a1 = scal[3];
    // sequence numbers of THIS exchange: one past the last executed ones (fs_comm.hip); read by every workgroup at its start,
    // advanced by the last workgroup through each part
    const unsigned long long rseq = *red.d_seq + 1ull, hseq = *snd.d_seq + 1ull;
    const int it_dev = PP ? pp.it_ctr[pp.par] : 0;
    if (st0 != 0) return;
    if (PP) {
        iter = it_dev;
        check_only = iter >= (int)it_max ? 1 : 0;
    } else if (iter < 0) {          // captured batch: the index of the product that preceded this launch
        iter = st2 - 1;
        check_only = iter >= (int)it_max ? 1 : 0;
    }
    const int rslot = (int)(rseq & 1ull), hslot = (int)(hseq & 1ull);
    const int64_t e_first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int t = threadIdx.x;
    // 1. w on the interface rows -> the neighbours
    for (int64_t e = e_first; e < snd.total_send; e += stride) {
        const double wv = w[snd.send_idx[e]];
        int j = 0;
        while (j + 1 < snd.nn && e >= snd.peers[j + 1].send_offset) ++j;
        const fs_p2p_peer q = snd.peers[j];
        fs_p2p_store(q.recv + (int64_t)hslot * q.peer_total + q.recv_offset + (e - q.send_offset), wv);
    }
    fs_p2p_stores_done();
    __syncthreads();
    if (t == 0 && atomicAdd(snd.counter, 1u) == gridDim.x - 1) {
        __hip_atomic_store(snd.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(snd.d_seq, hseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int j = 0; j < snd.nn; ++j) {
            const fs_p2p_peer q = snd.peers[j];
            fs_p2p_publish(q.flags + (int64_t)hslot * q.peer_nn + q.peer_slot, hseq);
        }
    }
    // 2. the sums of all ranks
    double sm[3];
    wg_sum_partials<3>(red.partials, red.npart, sm);
    if (blockIdx.x == 0 && t < red.nr && t != red.me) {
        double* dst = red.peer_buf[t] + ((int64_t)rslot * red.nr + red.me) * 8;
        fs_p2p_store(dst, sm[0]); fs_p2p_store(dst + 1, sm[1]); fs_p2p_store(dst + 2, sm[2]);
        fs_p2p_stores_done();
        fs_p2p_publish(red.peer_flags[t] + (int64_t)rslot * red.nr + red.me, rseq);
    }
    if (t < red.nr && t != red.me) fs_p2p_wait(red.own_flags + (int64_t)rslot * red.nr + t, rseq, red.timeout, red.err);
    __syncthreads();
    __shared__ double tot[3];
    if (t < 3) {
        double a = 0.0;
        for (int q = 0; q < red.nr; ++q)
            a += q == red.me ? sm[t] : fs_p2p_load(red.own_buf + ((int64_t)rslot * red.nr + q) * 8 + t);
        tot[t] = a;
        if (blockIdx.x == 0) red.sums_out[t] = a;
    }
    if (t == 0 && atomicAdd(red.counter, 1u) == gridDim.x - 1) {          // every workgroup has read d_seq and the other ranks' sums
        __hip_atomic_store(red.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(red.d_seq, rseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const double gamma = tot[0], delta = tot[1], rho = tot[2];
    if (rho <= thresh || check_only) return;
    double beta, alpha;
    if (!cg_scalars_from(iter, gamma, delta, rho, ((iter - 1) & 1) ? sc_g1 : sc_g0, ((iter - 1) & 1) ? sc_a1 : sc_a0, alpha, beta)) return;
    // 3. ghost rows: s_g <- w_g + beta s_g, r_g <- r_g - alpha s_g (the two lines of the update kernel)
    if (t < snd.nn) fs_p2p_wait(snd.own_flags + (int64_t)hslot * snd.nn + t, hseq, snd.timeout, snd.err);
    __syncthreads();
    const double* own_recv = snd.own_recv + (int64_t)hslot * snd.recv_stride;
    for (int64_t k = e_first; k < snd.total_recv; k += stride) {
        const int64_t gi = snd.recv_idx ? (int64_t)snd.recv_idx[k] : snd.n_owned + k;
        if (PP) {
            const double wv = fs_p2p_load(own_recv + k);
            pp.w_cur[gi] = wv;
            const double ss = fma(beta, pp.s_cur[gi], wv);
            pp.s_nxt[gi] = ss;
            pp.r_nxt[gi] = fma(-alpha, ss, pp.r_cur[gi]);
        } else {
            const double ss = fs_p2p_load(own_recv + k) + beta * s_ghost[k];
            s_ghost[k] = ss;
            r[gi] -= alpha * ss;
        }
    }
}

// ---- pipelined CG (Ghysels & Vanroose, Parallel Computing 40 (2014)) on the scaled system --------------------------
// The single-reduction recurrence above still has the global sums between the product and the update: on several GPUs the
// 3-double all-reduce (latency, not bandwidth) is paid in full every iteration.  The pipelined recurrence carries two more
// vectors (w = A r, z = A s) so that the sums of iteration i - (r.r, w.r, sum d r^2) of r_i, w_i - are known BEFORE the
// product n_i = A w_i starts and are reduced while it runs:
//      beta = gamma_i / gamma_{i-1},  alpha = gamma_i / (delta_i - beta gamma_i / alpha_{i-1})
//      z = n + beta z;  s = w + beta s;  p = r + beta p;   x += alpha p;  r -= alpha s;  w -= alpha z
// In exact arithmetic the iterates are those of CG.  The update kernel computes the sums of the NEW r, w as it writes
// them (per-workgroup partials, double-buffered by iteration parity because the next launch reads one set while it writes
// the other).  112 instead of 72 B/DOF of vector traffic per iteration: it only pays where a collective is hidden.
// Rows [m0, m1) are updated by this launch; rows [0, m0) and [m1, n) - what a slab sends to its neighbours - were already
// updated by k_pcg_update_rows (so that their exchange could start) and only enter the sums here.
template <bool FUSED, bool NT>
__global__ void __launch_bounds__(FS_BLOCK) k_pcg_update(int64_t n, int64_t m0, int64_t m1, int iter, int check_only,
                                                         double* __restrict__ partials, int npart,
                                                         const double* __restrict__ sums, const double* __restrict__ ctrl,
                                                         double* __restrict__ scal, int* __restrict__ status,
                                                         double* __restrict__ hist, const double* __restrict__ dvec,
                                                         double* __restrict__ r, double* __restrict__ w,
                                                         const double* __restrict__ nv, double* __restrict__ p,
                                                         double* __restrict__ sv, double* __restrict__ z,
                                                         double* __restrict__ x) {
    if (status[0] != 0) return;
    double gamma, delta, rho;
    if (FUSED) {
        double sm[3];
        wg_sum_partials<3>(partials + (int64_t)(iter & 1) * 3 * npart, npart, sm);
        gamma = sm[0]; delta = sm[1]; rho = sm[2];
    } else {
        gamma = sums[0]; delta = sums[1]; rho = sums[2];
    }
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    if (leader) hist[iter] = rho;
    if (rho <= ctrl[0]) {
        if (leader) { status[1] = iter; status[0] = 1; }
        return;
    }
    if (check_only) {
        if (leader) { status[1] = iter; status[0] = 3; }
        return;
    }
    double alpha, beta;
    if (!cg_scalars(iter, gamma, delta, rho, scal, alpha, beta)) {
        if (leader) { status[1] = iter; status[0] = 2; }
        return;
    }
    if (leader) {
        scal[2 * (iter & 1) + 0] = gamma;
        scal[2 * (iter & 1) + 1] = alpha;
    }
    typedef double v2d __attribute__((ext_vector_type(2)));
    struct io {
        static __device__ __forceinline__ v2d ld(const double* q) {
            return NT ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(q)) : *reinterpret_cast<const v2d*>(q);
        }
        static __device__ __forceinline__ void st(double* q, const v2d& v) {
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v2d*>(q));
            else *reinterpret_cast<v2d*>(q) = v;
        }
    };
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t npair = (m1 - m0) >> 1;              // m0 is even
    for (int64_t q = tid; q < npair; q += stride) {
        const int64_t i = m0 + 2 * q;
        const v2d nn = io::ld(nv + i), dd = io::ld(dvec + i);
        v2d zz = io::ld(z + i), ss = io::ld(sv + i), pp = io::ld(p + i), ww = io::ld(w + i), rr = io::ld(r + i), xx = io::ld(x + i);
        zz = nn + beta * zz;
        ss = ww + beta * ss;
        pp = rr + beta * pp;
        xx += alpha * pp;
        rr -= alpha * ss;
        ww -= alpha * zz;
        io::st(z + i, zz); io::st(sv + i, ss); io::st(p + i, pp); io::st(x + i, xx); io::st(r + i, rr); io::st(w + i, ww);
        a0 += rr.x * rr.x + rr.y * rr.y;
        a1 += ww.x * rr.x + ww.y * rr.y;
        a2 += dd.x * rr.x * rr.x + dd.y * rr.y * rr.y;
    }
    if (((m1 - m0) & 1) && tid == 0) {
        const int64_t i = m1 - 1;
        const double zz = nv[i] + beta * z[i], ss = w[i] + beta * sv[i], pp = r[i] + beta * p[i];
        z[i] = zz; sv[i] = ss; p[i] = pp;
        x[i] += alpha * pp;
        const double rr = r[i] - alpha * ss, ww = w[i] - alpha * zz;
        r[i] = rr; w[i] = ww;
        a0 += rr * rr; a1 += ww * rr; a2 += dvec[i] * rr * rr;
    }
    // rows already updated by k_pcg_update_rows
    const int64_t extra = m0 + (n - m1);
    for (int64_t t = tid; t < extra; t += stride) {
        const int64_t i = t < m0 ? t : m1 + (t - m0);
        const double rr = r[i], ww = w[i];
        a0 += rr * rr; a1 += ww * rr; a2 += dvec[i] * rr * rr;
    }
    __shared__ double lds4[4];
    const double t0 = fs_block_sum(a0, lds4);
    const double t1 = fs_block_sum(a1, lds4);
    const double t2 = fs_block_sum(a2, lds4);
    if (threadIdx.x == 0) {
        double* out = partials + (int64_t)((iter + 1) & 1) * 3 * npart;
        out[blockIdx.x] = t0;
        out[npart + blockIdx.x] = t1;
        out[2 * npart + blockIdx.x] = t2;
    }
}

// The pipelined update on the rows [0, a) and [b, n) only (sums from the all-reduce): no side effects, the launch on
// the remaining rows takes the same decisions from the same inputs and writes status / history / scalars.
__global__ void __launch_bounds__(FS_BLOCK) k_pcg_update_rows(int64_t a, int64_t b, int64_t n, int iter, int check_only,
                                                              const double* __restrict__ sums, const double* __restrict__ ctrl,
                                                              const double* __restrict__ scal, const int* __restrict__ status,
                                                              double* __restrict__ r, double* __restrict__ w,
                                                              const double* __restrict__ nv, double* __restrict__ p,
                                                              double* __restrict__ sv, double* __restrict__ z, double* __restrict__ x) {
    if (status[0] != 0) return;
    const double gamma = sums[0], delta = sums[1], rho = sums[2];
    if (rho <= ctrl[0] || check_only) return;
    double alpha, beta;
    if (!cg_scalars(iter, gamma, delta, rho, scal, alpha, beta)) return;
    const int64_t total = a + (n - b);
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < total; t += stride) {
        const int64_t i = t < a ? t : b + (t - a);
        const double zz = nv[i] + beta * z[i], ss = w[i] + beta * sv[i], pp = r[i] + beta * p[i];
        z[i] = zz; sv[i] = ss; p[i] = pp;
        x[i] += alpha * pp;
        r[i] -= alpha * ss;
        w[i] -= alpha * zz;
    }
}

// sums of the first iterate of a pass: partials (parity 0) of (r.r, w.r, sum d r^2)
__global__ void __launch_bounds__(FS_BLOCK) k_pcg_dots(int64_t n, const double* __restrict__ r, const double* __restrict__ w,
                                                       const double* __restrict__ dvec, double* __restrict__ partials, int npart) {
    __shared__ double lds4[4];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const double rr = r[i], ww = w[i];
        a0 += rr * rr; a1 += ww * rr; a2 += dvec[i] * rr * rr;
    }
    const double t0 = fs_block_sum(a0, lds4);
    const double t1 = fs_block_sum(a1, lds4);
    const double t2 = fs_block_sum(a2, lds4);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = t0;
        partials[npart + blockIdx.x] = t1;
        partials[2 * npart + blockIdx.x] = t2;
    }
}

// aval = D^-1/2 A D^-1/2 (copy; the caller's matrix is left untouched), sc = 1/sqrt(diag)
template <int BS>
__global__ void __launch_bounds__(FS_BLOCK) k_scale_copy(int64_t n_rows, int64_t n_slices,
                                                         const int64_t* __restrict__ slice_ptr,
                                                         const int32_t* __restrict__ sell_col,
                                                         const double* __restrict__ val, int64_t plane,
                                                         const double* __restrict__ sc, double* __restrict__ aval) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t r = s * FS_SLICE + lane;
        const int64_t base = slice_ptr[s] + lane;
        const int width = (int)((slice_ptr[s + 1] - slice_ptr[s]) >> 6);
        for (int k = 0; k < width; ++k) {
            const int64_t e = base + (int64_t)k * FS_SLICE;
            const int32_t c = sell_col[e];
#pragma unroll
            for (int i = 0; i < BS; ++i)
#pragma unroll
                for (int j = 0; j < BS; ++j) {
                    const int64_t idx = (int64_t)(i * BS + j) * plane + e;
                    aval[idx] = (c >= 0 && r < n_rows) ? val[idx] * sc[r * BS + i] * sc[(int64_t)c * BS + j] : 0.0;
                }
        }
    }
}

__global__ void k_pointwise_div(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                double* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = a[i] / b[i];
}

// partial of sum d (b - w)^2 ; optionally r = b - w
__global__ void __launch_bounds__(FS_BLOCK) k_residual_scaled(const double* __restrict__ b, const double* __restrict__ w,
                                                              const double* __restrict__ d, int64_t n,
                                                              double* __restrict__ r, double* __restrict__ partial) {
    __shared__ double lds4[4];
    double acc = 0.0;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const double x = b[i] - w[i];
        if (r) r[i] = x;
        acc += d[i] * x * x;
    }
    const double t = fs_block_sum(acc, lds4);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

template <int BS>
__global__ void k_extract_dinv(int64_t n_rows, const int64_t* __restrict__ slice_ptr,
                               const int32_t* __restrict__ sell_col, const double* __restrict__ val, int64_t plane,
                               int jacobi, double* __restrict__ dinv, int* __restrict__ err,
                               double* __restrict__ dvec = nullptr) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_rows; r += stride) {
        const int64_t sp0 = slice_ptr[r >> 6];
        const int width = (int)((slice_ptr[(r >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (r & 63);
        int kd = -1;
        for (int k = 0; k < width; ++k)
            if (sell_col[base + (int64_t)k * FS_SLICE] == r) { kd = k; break; }
        for (int i = 0; i < BS; ++i) {
            double d = 1.0;
            if (jacobi) {
                d = kd >= 0 ? val[(int64_t)(i * BS + i) * plane + base + (int64_t)kd * FS_SLICE] : 0.0;
                if (!(d != 0.0) || (jacobi == 2 && !(d > 0.0))) { atomicAdd(err, 1); d = 1.0; }
                if (dvec) dvec[r * BS + i] = d;
                d = jacobi == 2 ? 1.0 / sqrt(d) : 1.0 / d;
            }
            dinv[r * BS + i] = d;
        }
    }
}

__global__ void k_pointwise_mul(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                double* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = a[i] * b[i];
}

// r = b - w ; partial of r.r
__global__ void __launch_bounds__(FS_BLOCK) k_residual(const double* __restrict__ b, const double* __restrict__ w,
                                                       int64_t n, double* __restrict__ r,
                                                       double* __restrict__ partial) {
    __shared__ double lds4[4];
    double acc = 0.0;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const double d = b[i] - w[i];
        if (r) r[i] = d;
        acc += d * d;
    }
    const double t = fs_block_sum(acc, lds4);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// up to four vectors zeroed by ONE launch (the start of a CG pass: p, s, z and x - four memsets of 8 MB each at 1 M rows were four
// launches with their gaps)
__global__ void __launch_bounds__(FS_BLOCK) k_zero4(double* __restrict__ a, int64_t na, double* __restrict__ b, int64_t nb,
                                                    double* __restrict__ c, int64_t nc, double* __restrict__ d, int64_t nd) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double* const ptr[4] = {a, b, c, d};
    const int64_t len[4] = {na, nb, nc, nd};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double* __restrict__ v = ptr[q];
        const int64_t n2 = len[q] >> 1;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) reinterpret_cast<v2d*>(v)[i] = v2d{0.0, 0.0};
        if ((len[q] & 1) && blockIdx.x == 0 && threadIdx.x == 0) v[len[q] - 1] = 0.0;
    }
}

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
    FS_CHECK(g_lat.info.download(h, 1, s));
    if (debug) fprintf(stderr, "[lattice tiles] %d classes, tiles of %d x %d x %d rows: %d entries / rows that do not fit\n", ncls, LT_TX, LT_TY, LT_TZ, h[0]);
    g_lat.tables_ok = h[0] == 0;
    g_lat.dict_built = g_dict.n_built;
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
        if (g_lat.ok && g_lat.built_for == g_dict.built_for && g_lat.space_serial == sp->serial && g_lat.ncls == g_dict.ncls && g_lat.geom.grid > 0) return g_lat.geom.grid;
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

// y = A x for 4x4-block matrices (Taylor-Hood): one workgroup per slice, wave i computes block-row i.  A slice of a
// CG2 pattern holds 30-65 entries of 16 planes each; giving every block-row its own wave quarters the serial
// chain of a wave and quadruples the loads in flight (the generic kernel walks all 16 planes in one wave).
// TH: the matrix is a Taylor-Hood operator (fs_assemble_navier_stokes).  Only vertex nodes carry a pressure, so
//   - plane (i, 3) - the pressure-gradient column - is structurally zero wherever the COLUMN node is an edge node: its
//     load is predicated on the column being a vertex ([0, nvo) or the ghost vertices [gv0, gv1));
//   - block-row 3 of an EDGE node is the dummy identity row: slices that lie entirely behind the vertex rows copy x.
// 16 planes of 8 B per stored block shrink to about 10 on average: the FGMRES iteration of configs[4] went from 1.02 to 0.90 ms.
template <bool NT, bool TH>
__global__ void __launch_bounds__(FS_BLOCK) k_sell_spmv4_rows(int64_t n_rows, int64_t n_cols, int64_t n_slices,
                                                              const int64_t* __restrict__ slice_ptr,
                                                              const int32_t* __restrict__ sell_col,
                                                              const int32_t* __restrict__ dia_ptr,
                                                              const int32_t* __restrict__ dia_off,
                                                              const double* __restrict__ val, int64_t plane,
                                                              const double* __restrict__ x, double* __restrict__ y,
                                                              int64_t nvo, int64_t gv0, int64_t gv1) {
    const int lane = threadIdx.x & 63;
    const int i = threadIdx.x >> 6;          // block-row of this wave
    const int64_t cmax = n_cols - 1;
    // Slices are dealt round-robin (consecutive slices to different XCDs).  Measured on the configs[4] matrix: 443-490 us,
    // with or without the spatial order of the slices; 567 us for XCD-contiguous eighths of the ordered slices and
    // 878 us for contiguous eighths of the rows (vertex rows are 65 blocks wide, edge rows 20-30: unbalanced)
    for (int64_t s = blockIdx.x; s < n_slices; s += gridDim.x) {
        const int64_t r = s * FS_SLICE + lane;
        if (TH && i == 3 && s * FS_SLICE >= nvo) {       // dummy pressure rows of edge nodes
            if (r < n_rows) y[r * 4 + 3] = x[r * 4 + 3];
            continue;
        }
        const int64_t base = slice_ptr[s];
        const int width = (int)((slice_ptr[s + 1] - base) >> 6);
        const int32_t dp = dia_ptr[s];
        const double* __restrict__ vp = val + (int64_t)(i * 4) * plane + base + lane;
        const int32_t* __restrict__ cp = sell_col + base + lane;
        // DIA slice: [split][list A][list B if split < 64] (fs_symbolic.hip, "SPLIT slices"); this lane's list
        const int split = dp >= 0 ? dia_off[dp] : FS_SLICE;
        const int32_t* __restrict__ opa = dia_off + (dp >= 0 ? dp + 1 : 0);
        const int32_t* __restrict__ opb = opa + (split < FS_SLICE ? width : 0);
        const bool hi = lane >= split;
        double acc = 0.0;
        int k = 0;
        constexpr int U = 4;       // entries per round (2: 391 us, 4: 374 us, 8: 476 us on the configs[4] matrix - register pressure)
        for (; k + U <= width; k += U) {
            int64_t c[U];
            bool pv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (dp >= 0) {
                    c[u] = r + (hi ? opb[k + u] : opa[k + u]);
                    c[u] = c[u] < 0 ? 0 : (c[u] > cmax ? cmax : c[u]);
                } else {
                    c[u] = fs_col_decode(cp[(int64_t)(k + u) * FS_SLICE]);
                }
                pv[u] = !TH || c[u] < nvo || (c[u] >= gv0 && c[u] < gv1);
            }
            double v[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int j = 0; j < 3; ++j) v[u][j] = fs_ldv<NT>(&vp[(int64_t)j * plane + (int64_t)(k + u) * FS_SLICE]);
                v[u][3] = pv[u] ? fs_ldv<NT>(&vp[(int64_t)3 * plane + (int64_t)(k + u) * FS_SLICE]) : 0.0;
            }
            double2 xa[U], xb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xa[u] = reinterpret_cast<const double2*>(x)[2 * c[u]];
                xb[u] = reinterpret_cast<const double2*>(x)[2 * c[u] + 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u][0] * xa[u].x + v[u][1] * xa[u].y + v[u][2] * xb[u].x + v[u][3] * xb[u].y;
        }
        for (; k < width; ++k) {
            int64_t c = dp >= 0 ? r + (hi ? opb[k] : opa[k]) : (int64_t)fs_col_decode(cp[(int64_t)k * FS_SLICE]);
            c = c < 0 ? 0 : (c > cmax ? cmax : c);
            const bool pc = !TH || c < nvo || (c >= gv0 && c < gv1);
            const double2 xa = reinterpret_cast<const double2*>(x)[2 * c], xb = reinterpret_cast<const double2*>(x)[2 * c + 1];
            acc += vp[(int64_t)k * FS_SLICE] * xa.x + vp[plane + (int64_t)k * FS_SLICE] * xa.y +
                   vp[2 * plane + (int64_t)k * FS_SLICE] * xb.x + (pc ? vp[3 * plane + (int64_t)k * FS_SLICE] : 0.0) * xb.y;
        }
        if (r < n_rows) y[r * 4 + i] = acc;
    }
}

// The same product with the ENTRIES of a slice dealt to the four waves (wave w takes entries w, w + 4, ...) instead of the block-rows:
// a wave multiplies whole 4 x 4 blocks, so the four values of x behind a column are loaded once per entry instead of once per wave
// and block-row - 16 + 2 load instructions per entry where the block-row kernel issues 4 x (4 + 2) - and the four partial sums of
// every row meet in LDS (8 KB, two barriers per slice), added in wave order.  Round 5: the block-row kernel streamed the 1.45 GB of
// the configs[4] operator at 3.8 TB/s, its texture-address units busy with the x gathers of all four waves.
#ifndef FS_SPMV4_U
#define FS_SPMV4_U 1      // (measured on the configs[4] operator: 1: 317 us, 2: 359 us; the block-row kernel: 388 us)
#endif
template <bool NT, bool TH>
__global__ void __launch_bounds__(FS_BLOCK) k_sell_spmv4_ksplit(int64_t n_rows, int64_t n_cols, int64_t n_slices,
                                                                const int64_t* __restrict__ slice_ptr,
                                                                const int32_t* __restrict__ sell_col,
                                                                const int32_t* __restrict__ dia_ptr,
                                                                const int32_t* __restrict__ dia_off,
                                                                const double* __restrict__ val, int64_t plane,
                                                                const double* __restrict__ x, double* __restrict__ y,
                                                                int64_t nvo, int64_t gv0, int64_t gv1) {
    __shared__ double part[4][4][FS_SLICE];         // [wave][block-row][lane]
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t cmax = n_cols - 1;
    for (int64_t s = blockIdx.x; s < n_slices; s += gridDim.x) {
        const int64_t r = s * FS_SLICE + lane;
        const bool edge_rows = TH && s * FS_SLICE >= nvo;         // block-row 3 of these nodes is the dummy identity row
        const int64_t base = slice_ptr[s];
        const int width = (int)((slice_ptr[s + 1] - base) >> 6);
        const int32_t dp = dia_ptr[s];
        const double* __restrict__ vp = val + base + lane;
        const int32_t* __restrict__ cp = sell_col + base + lane;
        const int split = dp >= 0 ? dia_off[dp] : FS_SLICE;
        const int32_t* __restrict__ opa = dia_off + (dp >= 0 ? dp + 1 : 0);
        const int32_t* __restrict__ opb = opa + (split < FS_SLICE ? width : 0);
        const bool hi = lane >= split;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        // U entries of this wave in flight together (U x 18 loads per lane); an entry past the end repeats the last one with weight 0
        constexpr int U = FS_SPMV4_U;
        for (int k0 = w; k0 < width; k0 += 4 * U) {
            double2 xa[U], xb[U];
            double v[U][4][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = k0 + 4 * u < width;
                const int k = live ? k0 + 4 * u : k0;
                int64_t c = dp >= 0 ? r + (hi ? opb[k] : opa[k]) : (int64_t)fs_col_decode(cp[(int64_t)k * FS_SLICE]);
                c = c < 0 ? 0 : (c > cmax ? cmax : c);
                const bool pc = !TH || c < nvo || (c >= gv0 && c < gv1);          // the column node carries a pressure
                xa[u] = reinterpret_cast<const double2*>(x)[2 * c];
                xb[u] = reinterpret_cast<const double2*>(x)[2 * c + 1];
                if (!live) { xa[u] = make_double2(0.0, 0.0); xb[u] = xa[u]; }
                const double* __restrict__ ve = vp + (int64_t)k * FS_SLICE;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool row_on = live && !(edge_rows && i == 3);
#pragma unroll
                    for (int j = 0; j < 3; ++j) v[u][i][j] = row_on ? fs_ldv<NT>(&ve[(int64_t)(i * 4 + j) * plane]) : 0.0;
                    v[u][i][3] = (row_on && pc) ? fs_ldv<NT>(&ve[(int64_t)(i * 4 + 3) * plane]) : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] += v[u][i][0] * xa[u].x + v[u][i][1] * xa[u].y + v[u][i][2] * xb[u].x + v[u][i][3] * xb[u].y;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) part[w][i][lane] = acc[i];
        __syncthreads();
        if (r < n_rows) {
            // wave w adds up block-row w
            const double sum = ((part[0][w][lane] + part[1][w][lane]) + part[2][w][lane]) + part[3][w][lane];
            y[r * 4 + w] = (edge_rows && w == 3) ? x[r * 4 + 3] : sum;
        }
        __syncthreads();
    }
}

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
    const int igrid = fused_common ? spmv_partials_unsplit(sp, bs) : 0;        // workgroups (= dot partials) of the iteration kernel
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
                    if (fusedp) {
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
                    launch_spmv<3>(A, ws.z.p, ws.w.p, ws.dvec.p, ws.partials.p, ws.status.p, s, aval, nullptr, 0, 0, 0, 0);
                }
                if (graph_sized && graphs_allowed && k >= first_plain && kend - k == bsz && kend <= max_iter && (bsz & 1) == 0 && (k & 1) == 0) {
                    const void* key[24] = {A, aval, x->d.p, hist_p, ws.z.p, ws.w.p, ws.partials.p, ws.status.p, ws.dvec.p, ws.p.p, ws.s.p,
                                           ws.z2.p, ws.w2.p, ws.s2.p, ws.it_ctr.p, g_dict.cls.p, g_dict.values.p, sp->dict_items.p, sp->dict_plans.p,
                                           ws.ctrl.p, ws.scal.p, snd2[0].own_recv, red2[0].own_buf,
                                           fusedp ? reinterpret_cast<const void*>((uintptr_t)sp->halo.p2p.generation + 1) : nullptr};
                    const int64_t key_i[8] = {n, ((int64_t)g_dict.ncls * 256 + g_dict.S) * 4 + (fusedp ? 1 : 0) + (mirror_dev ? 2 : 0), bsz, igrid,
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
                const void* key[32] = {A, aval, x->d.p, hist_p, ws.z.p, ws.w.p, ws.partials.p, ws.status.p,
                                       ws.dvec.p, ws.p.p, ws.s.p, sp->sell_col.p, snd.own_recv, snd.peers, red.own_buf, red.peer_buf,
                                       dict_on ? (const void*)g_dict.cls.p : nullptr, dict_on ? (const void*)g_dict.values.p : nullptr,
                                       dict_on ? (const void*)sp->dict_items.p : nullptr, p2p_fuse ? (const void*)ws.sg.p : nullptr,
                                       dict_on ? (const void*)sp->dict_plans.p : nullptr, dict_on ? (const void*)sp->halo.items_interior.p : nullptr,
                                       dict_on ? (const void*)sp->halo.items_boundary.p : nullptr,
                                       p2p_fuse ? reinterpret_cast<const void*>((uintptr_t)sp->halo.p2p.generation) : nullptr,
                                       lat_on ? (const void*)g_lat.tile_cls.p : nullptr, lat_on ? (const void*)g_lat.cnt.p : nullptr,
                                       lat_on ? (const void*)g_lat.coef.p : nullptr, lat_on ? (const void*)g_lat.rel.p : nullptr,
                                       lat_on ? (const void*)g_lat.off.p : nullptr, lat_on ? (const void*)g_lat.relc.p : nullptr,
                                       lat_on ? reinterpret_cast<const void*>((uintptr_t)g_lat.geom.n_tiles) : nullptr,
                                       lat_on ? reinterpret_cast<const void*>(((uintptr_t)g_lat.geom.grid << 32) | (uintptr_t)(uint32_t)g_lat.geom.w_tiles) : nullptr};
                // (+ whether the product is the row-dictionary kernel, with the class count and width its launch bakes in)
                const int64_t dict_sig = dict_on ? ((int64_t)g_dict.ncls * 256 + g_dict.S) * 256 + g_dict.C : 0;
                const int64_t key_i[8] = {n, (p2p_fuse ? (int64_t)p2p_rows_cap + 1 : 0) + 1024 * dict_sig, bsz, fgrid, vgrid,
                                          (int64_t)upd_nt * 2 + (int64_t)spmv_nontemporal(sp, bs) + (mirror_dev ? 4 : 0),
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
        stats->product_kind = fused ? 1 : g_last_product_kind;
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
