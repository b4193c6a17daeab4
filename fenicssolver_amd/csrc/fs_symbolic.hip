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

// ---- keys ------------------------------------------------------------------------------
// 12 directed pairs per tet (a != b) + one diagonal key per owned row.
__global__ void k_pair_keys(const int32_t* __restrict__ cells, int64_t nc, int64_t n_rows,
                            uint64_t* __restrict__ keys) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < nc; c += stride) {
        const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
        const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
        int k = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (a == b) continue;
                uint64_t key = ~0ULL;  // rows owned elsewhere sort to the end and are dropped
                if (v[a] < n_rows) key = ((uint64_t)(uint32_t)v[a] << 32) | (uint32_t)v[b];
                keys[(int64_t)k * nc + c] = key;
                ++k;
            }
        }
    }
}

__global__ void k_diag_keys(int64_t n_rows, uint64_t* __restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_rows; i += stride) keys[i] = ((uint64_t)i << 32) | (uint64_t)i;
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

// one wavefront per slice: width = max row length in the slice
__global__ void __launch_bounds__(FS_BLOCK) k_slice_width(const int32_t* __restrict__ rowptr, int64_t n_rows,
                                                          int64_t n_slices, int64_t* __restrict__ slice_entries,
                                                          int* __restrict__ max_row) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t r = s * FS_SLICE + lane;
        int len = 0;
        if (r < n_rows) len = rowptr[r + 1] - rowptr[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) len = max(len, __shfl_xor(len, off, 64));
        if (lane == 0) {
            slice_entries[s] = (int64_t)len * FS_SLICE;
            atomicMax(max_row, len);
        }
    }
}

// sell_col[slice_ptr[s] + k*64 + lane] = column k of row, padding = the row itself
__global__ void __launch_bounds__(FS_BLOCK) k_fill_sell(const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ colidx, int64_t n_rows,
                                                        int64_t n_slices, const int64_t* __restrict__ slice_ptr,
                                                        int32_t* __restrict__ sell_col) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t base = slice_ptr[s];
        const int width = (int)((slice_ptr[s + 1] - base) >> 6);
        const int64_t r = s * FS_SLICE + lane;
        int64_t start = 0;
        int len = 0;
        int32_t self = (int32_t)(r < n_rows ? r : n_rows - 1);
        if (r < n_rows) {
            start = rowptr[r];
            len = rowptr[r + 1] - (int32_t)start;
        }
        for (int k = 0; k < width; ++k) sell_col[base + (int64_t)k * FS_SLICE + lane] = k < len ? colidx[start + k] : self;
    }
}

// slots[(a*4+b)*nc + c] = SELL entry of (cells[c][a], cells[c][b]) or -1 when the row is not owned
__global__ void k_slots(const int32_t* __restrict__ cells, int64_t nc, int64_t n_rows,
                        const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                        const int64_t* __restrict__ slice_ptr, int32_t* __restrict__ slots,
                        int* __restrict__ err) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < nc; c += stride) {
        const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
        const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int32_t row = v[a];
            int32_t start = 0, end = 0;
            int64_t base = 0;
            const bool owned = row < n_rows;
            if (owned) {
                start = rowptr[row];
                end = rowptr[row + 1];
                base = slice_ptr[row >> 6] + (row & 63);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                int32_t slot = -1;
                if (owned) {
                    int32_t lo = start, hi = end;
                    const int32_t target = v[b];
                    while (lo < hi) {
                        int32_t mid = (lo + hi) >> 1;
                        if (colidx[mid] < target) lo = mid + 1; else hi = mid;
                    }
                    if (lo < end && colidx[lo] == target) {
                        slot = (int32_t)(base + (int64_t)(lo - start) * FS_SLICE);
                    } else {
                        atomicAdd(err, 1);
                    }
                }
                slots[(int64_t)(a * 4 + b) * nc + c] = slot;
            }
        }
    }
}

// ---- row-gather incidence tables ----------------------------------------------------------------
// key = (vertex << 32) | (cell*4 + local vertex) for every owned (cell, vertex) incidence
__global__ void k_inc_keys(const int32_t* __restrict__ cells, int64_t n_inc, int64_t n_rows,
                           uint64_t* __restrict__ keys) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; q < n_inc; q += stride) {
        const int32_t v = cells[q];
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
    for (; s < n_slices; s += stride) {
        const int64_t r = s * FS_SLICE + lane;
        int cnt = 0;
        if (r < n_rows) cnt = inc_ptr[r + 1] - inc_ptr[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt = max(cnt, __shfl_xor(cnt, off, 64));
        if (lane == 0) {
            slice_entries[s] = (int64_t)cnt * FS_SLICE;
            atomicMax(max_cnt, cnt);
        }
    }
}

__global__ void __launch_bounds__(FS_BLOCK) k_inc_fill(const uint64_t* __restrict__ keys,
                                                       const int32_t* __restrict__ inc_ptr,
                                                       const int32_t* __restrict__ cells,
                                                       const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ colidx, int64_t n_rows,
                                                       int64_t n_slices, const int64_t* __restrict__ inc_slice_ptr,
                                                       int32_t* __restrict__ inc_cell,
                                                       uint32_t* __restrict__ inc_pos, int* __restrict__ err) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t base = inc_slice_ptr[s];
        const int width = (int)((inc_slice_ptr[s + 1] - base) >> 6);
        const int64_t r = s * FS_SLICE + lane;
        int32_t first = 0, cnt = 0, rs = 0, re = 0;
        if (r < n_rows) {
            first = inc_ptr[r];
            cnt = inc_ptr[r + 1] - first;
            rs = rowptr[r];
            re = rowptr[r + 1];
        }
        for (int j = 0; j < width; ++j) {
            int32_t q = -1;
            uint32_t packed = 0;
            if (j < cnt) {
                q = (int32_t)(keys[first + j] & 0xffffffffULL);
                const int4 v4 = reinterpret_cast<const int4*>(cells)[q >> 2];
                const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    int32_t lo = rs, hi = re;
                    while (lo < hi) {
                        const int32_t mid = (lo + hi) >> 1;
                        if (colidx[mid] < v[b]) lo = mid + 1; else hi = mid;
                    }
                    if (!(lo < re && colidx[lo] == v[b]) || lo - rs > 255) atomicAdd(err, 1);
                    packed |= (uint32_t)((lo - rs) & 255) << (8 * b);
                }
            }
            inc_cell[base + (int64_t)j * FS_SLICE + lane] = q;
            inc_pos[base + (int64_t)j * FS_SLICE + lane] = packed;
        }
    }
}

// ---- API -----------------------------------------------------------------------------------
extern "C" int fs_space_create(fs_mesh_t mesh, int family, int degree, int ncomp, fs_space_t* out) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(mesh && out, "fs_space_create: null pointer");
    if (family != FS_FAMILY_CG || degree != 1 || (ncomp != 1 && ncomp != 3)) {
        fs_set_error("fs_space_create: only CG degree 1 with 1 or 3 components is supported (family=%d degree=%d ncomp=%d)",
                     family, degree, ncomp);
        return FS_ERR_UNSUPPORTED;
    }
    hipStream_t s = fs_rt().stream;
    const int64_t nc = mesh->nc, n_rows = mesh->n_owned;
    FS_REQUIRE(n_rows > 0, "fs_space_create: process owns no vertices");
    fs_space_s* sp = new fs_space_s();
    sp->mesh = mesh;
    sp->degree = degree;
    sp->ncomp = ncomp;
    sp->n_nodes_local = mesh->nv;
    sp->n_nodes_owned = n_rows;
    sp->n_dofs_local = mesh->nv * ncomp;
    sp->n_dofs_owned = n_rows * ncomp;

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

    // 1. keys
    const int64_t n_keys = 12 * nc + n_rows;
    FS_REQUIRE(n_keys < (int64_t)INT32_MAX, "fs_space_create: %lld pattern keys exceed int32 (mesh too large for one GPU pass)", (long long)n_keys);
    int64_t nnz = 0;
    {
        dbuf<uint64_t> keys_a, keys_b;
        dbuf<int> d_count;
        FS_SP(keys_a.alloc(n_keys));
        FS_SP(keys_b.alloc(n_keys));
        FS_SP(d_count.alloc(1));
        hipLaunchKernelGGL(k_pair_keys, dim3(fs_grid_for(nc)), dim3(FS_BLOCK), 0, s, mesh->cells.p, nc, n_rows, keys_a.p);
        hipLaunchKernelGGL(k_diag_keys, dim3(fs_grid_for(n_rows)), dim3(FS_BLOCK), 0, s, n_rows, keys_a.p + 12 * nc);
        FS_SP_HIP(hipGetLastError());
        // 2. sort + unique
        int end_bit = 64;
        size_t tmp_bytes = 0;
        FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, keys_a.p, keys_b.p, (int)n_keys, 0, end_bit, s));
        size_t tmp2 = 0;
        FS_SP_HIP(hipcub::DeviceSelect::Unique(nullptr, tmp2, keys_b.p, keys_a.p, d_count.p, (int)n_keys, s));
        if (tmp2 > tmp_bytes) tmp_bytes = tmp2;
        dbuf<char> tmp;
        FS_SP(tmp.alloc((int64_t)tmp_bytes + 16));
        size_t tb = tmp_bytes;
        FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, keys_a.p, keys_b.p, (int)n_keys, 0, end_bit, s));
        tb = tmp_bytes;
        FS_SP_HIP(hipcub::DeviceSelect::Unique(tmp.p, tb, keys_b.p, keys_a.p, d_count.p, (int)n_keys, s));
        int h_count = 0;
        FS_SP(d_count.download(&h_count, 1, s));
        nnz = h_count;
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
    // 4. SELL-64
    const int64_t n_slices = (n_rows + FS_SLICE - 1) / FS_SLICE;
    sp->n_slices = n_slices;
    {
        dbuf<int64_t> entries;
        dbuf<int> d_max;
        FS_SP(entries.alloc(n_slices + 1));
        FS_SP(entries.zero(s));
        FS_SP(d_max.alloc(1));
        FS_SP(d_max.zero(s));
        FS_SP(sp->slice_ptr.alloc(n_slices + 1));
        hipLaunchKernelGGL(k_slice_width, dim3(fs_grid_for(n_slices * 64)), dim3(FS_BLOCK), 0, s, sp->rowptr.p, n_rows, n_slices, entries.p, d_max.p);
        FS_SP_HIP(hipGetLastError());
        size_t tmp_bytes = 0;
        FS_SP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, entries.p, sp->slice_ptr.p, (int)(n_slices + 1), s));
        dbuf<char> tmp;
        FS_SP(tmp.alloc((int64_t)tmp_bytes + 16));
        FS_SP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, entries.p, sp->slice_ptr.p, (int)(n_slices + 1), s));
        int64_t total = 0;
        FS_SP_HIP(hipMemcpyAsync(&total, sp->slice_ptr.p + n_slices, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        FS_SP(d_max.download(&sp->max_row, 1, s));
        sp->sell_entries = total;
    }
    if (sp->sell_entries >= (int64_t)INT32_MAX) {
        fs_set_error("fs_space_create: SELL storage of %lld entries exceeds int32 slot indexing", (long long)sp->sell_entries);
        delete sp;
        return FS_ERR_UNSUPPORTED;
    }
    FS_SP(sp->sell_col.alloc(sp->sell_entries));
    hipLaunchKernelGGL(k_fill_sell, dim3(fs_grid_for(n_slices * 64)), dim3(FS_BLOCK), 0, s, sp->rowptr.p, sp->colidx.p, n_rows, n_slices, sp->slice_ptr.p, sp->sell_col.p);
    FS_SP_HIP(hipGetLastError());
    if (ncomp == 1 && sp->max_row <= 255) {
        // 5a. scalar spaces: row-gather incidence tables (deterministic, atomic-free assembly)
        const int64_t n_inc = 4 * nc;
        FS_REQUIRE(n_inc < (int64_t)INT32_MAX, "fs_space_create: cell-vertex incidences exceed int32");
        dbuf<uint64_t> ka, kb;
        dbuf<int32_t> inc_ptr;
        dbuf<int64_t> entries;
        dbuf<int> d_max, d_err;
        FS_SP(ka.alloc(n_inc));
        FS_SP(kb.alloc(n_inc));
        FS_SP(inc_ptr.alloc(n_rows + 1));
        FS_SP(entries.alloc(n_slices + 1));
        FS_SP(entries.zero(s));
        FS_SP(d_max.alloc(1));
        FS_SP(d_max.zero(s));
        FS_SP(d_err.alloc(1));
        FS_SP(d_err.zero(s));
        hipLaunchKernelGGL(k_inc_keys, dim3(fs_grid_for(n_inc)), dim3(FS_BLOCK), 0, s, mesh->cells.p, n_inc, n_rows, ka.p);
        FS_SP_HIP(hipGetLastError());
        size_t tmp_bytes = 0;
        FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, ka.p, kb.p, (int)n_inc, 0, 64, s));
        size_t tmp2 = 0;
        FS_SP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, entries.p, entries.p, (int)(n_slices + 1), s));
        if (tmp2 > tmp_bytes) tmp_bytes = tmp2;
        dbuf<char> tmp;
        FS_SP(tmp.alloc((int64_t)tmp_bytes + 16));
        size_t tb = tmp_bytes;
        FS_SP_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, ka.p, kb.p, (int)n_inc, 0, 64, s));
        // rows of other ranks (key ~0) sort to the end: count of valid keys = first index of the sentinel row
        hipLaunchKernelGGL(k_rowptr, dim3(fs_grid_for(n_rows + 1)), dim3(FS_BLOCK), 0, s, kb.p, n_inc, n_rows, inc_ptr.p);
        FS_SP(sp->inc_slice_ptr.alloc(n_slices + 1));
        hipLaunchKernelGGL(k_inc_width, dim3(fs_grid_for(n_slices * 64)), dim3(FS_BLOCK), 0, s, inc_ptr.p, n_rows, n_slices, entries.p, d_max.p);
        FS_SP_HIP(hipGetLastError());
        tb = tmp_bytes;
        FS_SP_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, entries.p, sp->inc_slice_ptr.p, (int)(n_slices + 1), s));
        int64_t total = 0;
        FS_SP_HIP(hipMemcpyAsync(&total, sp->inc_slice_ptr.p + n_slices, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        FS_SP(d_max.download(&sp->inc_max, 1, s));
        sp->inc_entries = total;
        FS_SP(sp->inc_cell.alloc(total));
        FS_SP(sp->inc_pos.alloc(total));
        hipLaunchKernelGGL(k_inc_fill, dim3(fs_grid_for(n_slices * 64)), dim3(FS_BLOCK), 0, s, kb.p, inc_ptr.p, mesh->cells.p, sp->rowptr.p, sp->colidx.p, n_rows, n_slices, sp->inc_slice_ptr.p, sp->inc_cell.p, sp->inc_pos.p, d_err.p);
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
        FS_SP(sp->slots.alloc(16 * nc));
        hipLaunchKernelGGL(k_slots, dim3(fs_grid_for(nc)), dim3(FS_BLOCK), 0, s, mesh->cells.p, nc, n_rows, sp->rowptr.p, sp->colidx.p, sp->slice_ptr.p, sp->slots.p, d_err.p);
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
