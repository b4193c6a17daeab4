// fs_lattice.hip - the solver's LATTICE-ORDERED SHADOW of a scalar CG2 operator on a uniform box (one GPU).
//
// A CG2 space numbers its nodes [vertices | edge nodes, class by class] (fs_space_create; the API and DOLFIN's own numbering know
// nothing better), so a row's neighbours sit in eight far-apart blocks and its (col - row) offsets come as 20 - 39 runs of mostly one
// or two consecutive columns: the row-dictionary product (fs_krylov.hip, k_dict_spmv) needs 3 - 5 DEPENDENT rounds of eight loads per
// work item and reached 0.17 of the HBM peak on BASELINE configs[3] ("a numbering problem, not a kernel-tuning
// one").  The dofs of a CG2 space on a Kuhn-split box are exactly the points of the half grid (2 nx + 1) x (2 ny + 1) x (2 nz + 1):
// vertex (i, j, k) at (2 i, 2 j, 2 k), the mid-point of an edge at the sum of its end points' grid coordinates.  Numbered along that
// lattice, x fastest - one dummy dof per line so that a line has an even number of rows - a row's neighbours lie in at most 19 lines
// at offsets -2 .. 2 around it: 26 runs of up to three columns on the lines through vertices (vertex and x-edge rows alternate,
// the x-edge row's set nested in the vertex row's), 11 on the other three quarters of the lines (two alternating sets of 19 / 27).
//
// The API numbering is not touched.  The Krylov solve (PETSc KSPSolve behind SolverBase.py:608-612) permutes into the shadow and
// back: the matrix values through a precomputed entry map (one scattered copy per solve, 1 % of a 10 M-row solve), b and x through
// the node permutation.  Everything structural is built once per space, on the device.
//
// MEASURED (round 5, MI355X, configs[3]): the WORK-ITEM product (k_dict_spmv) on this order is slower than on the space's numbering - 222 -
// 272 us against 191 us -, the TILE product written for it (fs_krylov.hip, k_lattice_spmv: x through LDS windows, a wave per line parity,
// class lists broadcast) is faster: 155 us, the iteration 282 against 297 us; at 1.03 M rows 52.6 against 98.8 us per iteration (there the
// space's numbering gets no dictionary at all).  Automatic from 270 000 rows on (option "lattice_order" = -1; 400 000 before the tile product's second half of round 5), given up for a space whose
// shadow does not fit the tile form.
#include "fs_common.h"

struct fs_lattice_shadow {
    fs_space_s* sp = nullptr;       // structure of P A P^T (pattern, SELL / DIA storage, dictionary hints); no mesh, no assembly tables
    fs_matrix_s* A = nullptr;       // its values
    dbuf<int32_t> perm;             // [n nodes of the space] node -> shadow row
    dbuf<int32_t> emap;             // [sell_entries of the space] stored entry -> shadow entry (-1: padding)
    dbuf<int32_t> imap;             // [sell_entries of the shadow] shadow entry -> stored entry of the space (-1: none): what the copy reads
    fs_vector_s b, x;
    int64_t n = 0;                  // shadow rows (dummies included)
    ~fs_lattice_shadow() {
        delete A;
        delete sp;
    }
};

void fs_lattice_release(fs_lattice_shadow* L) { delete L; }
fs_space_s::~fs_space_s() { fs_lattice_release(lattice); }

// node -> lattice row.  Vertices x fastest as fs_mesh_create_box numbers them; an edge node by the sum of its end points.
__global__ void k_lattice_perm(int64_t nv, int64_t n_nodes, const int32_t* __restrict__ edges, int64_t vx, int64_t vy, int64_t SX, int64_t NY,
                               int32_t* __restrict__ perm) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_nodes; i += stride) {
        int64_t X, Y, Z;
        if (i < nv) {
            X = 2 * (i % vx); Y = 2 * ((i / vx) % vy); Z = 2 * (i / (vx * vy));
        } else {
            const int64_t a = edges[2 * (i - nv)], b = edges[2 * (i - nv) + 1];
            X = a % vx + b % vx; Y = (a / vx) % vy + (b / vx) % vy; Z = a / (vx * vy) + b / (vx * vy);
        }
        perm[i] = (int32_t)(X + SX * (Y + NY * Z));
    }
}

__global__ void k_lattice_row_lengths(int64_t n_rows, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm, int32_t* __restrict__ len) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_rows; r += stride) len[perm[r]] = rowptr[r + 1] - rowptr[r];
}

// rows nobody wrote a length for are the dummies (or, were the geometry not what it is taken for, collisions: counted)
__global__ void k_lattice_dummy_lengths(int64_t n_sh, int32_t* __restrict__ len, int* __restrict__ n_dummy) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int c = 0;
    for (; r < n_sh; r += stride)
        if (len[r] == 0) { len[r] = 1; ++c; }
    if (c) atomicAdd(n_dummy, c);
}

// the permuted columns of every row, ascending; dummy rows: their diagonal.
// A WAVE per row (round 6; the first form - a thread per row, insertion sort in place in global memory, up to 65 x 65 / 4 dependent
// read-modify-writes - took 56 ms in one launch at configs[3], 0.6 of a steady step): a lane takes up to two of the row's entries,
// the wave's entries sit in LDS, and an entry's place is the number of entries smaller than it (the columns of a row are distinct:
// perm is a bijection).  Rows of more than 128 entries (none on a tetrahedral CG2 space) are sorted by one lane as before.
__global__ void __launch_bounds__(FS_BLOCK) k_lattice_fill_rows(int64_t n_rows, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                               const int32_t* __restrict__ perm, const int32_t* __restrict__ rowptr_sh,
                                                               int32_t* __restrict__ col_sh, uint8_t* __restrict__ written) {
    __shared__ int32_t buf[FS_BLOCK / 64][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; r < n_rows; r += stride) {
        const int32_t p = perm[r];
        int32_t* __restrict__ o = col_sh + rowptr_sh[p];
        const int32_t s0 = rowptr[r], w = rowptr[r + 1] - s0;
        if (w > 128) {
            if (lane == 0) {
                for (int k = 0; k < w; ++k) {
                    const int32_t c = perm[colidx[s0 + k]];
                    int j = k;
                    while (j > 0 && o[j - 1] > c) { o[j] = o[j - 1]; --j; }
                    o[j] = c;
                }
                written[p] = 1;
            }
            continue;
        }
        const int32_t c0 = lane < w ? perm[colidx[s0 + lane]] : 0x7fffffff;
        const int32_t c1 = lane + 64 < w ? perm[colidx[s0 + 64 + lane]] : 0x7fffffff;
        buf[wave][lane] = c0;
        buf[wave][64 + lane] = c1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int k0 = 0, k1 = 0;
        for (int k = 0; k < w; ++k) {
            const int32_t v = buf[wave][k];         // (the same address in every lane: a broadcast)
            k0 += v < c0;
            k1 += v < c1;
        }
        if (lane < w) o[k0] = c0;
        if (lane + 64 < w) o[k1] = c1;
        if (lane == 0) written[p] = 1;
        __builtin_amdgcn_wave_barrier();            // (the next row's entries overwrite these)
    }
}
__global__ void k_lattice_fill_dummies(int64_t n_sh, const int32_t* __restrict__ rowptr_sh, const uint8_t* __restrict__ written, int32_t* __restrict__ col_sh) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_sh; r += stride)
        if (!written[r]) col_sh[rowptr_sh[r]] = (int32_t)r;
}

// stored entry of the space -> stored entry of the shadow (both SELL-64 layouts: entry k of row r at slice_ptr[r >> 6] + (r & 63) + 64 k)
__global__ void k_lattice_entry_map(int64_t n_slices, int64_t n_rows, const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ sell_col,
                                    const int32_t* __restrict__ perm, const int64_t* __restrict__ slice_ptr_sh, const int32_t* __restrict__ sell_col_sh,
                                    int32_t* __restrict__ emap, int* __restrict__ missing) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    int miss = 0;
    for (; s < n_slices; s += stride) {
        const int64_t r = s * FS_SLICE + lane;
        const int64_t base = slice_ptr[s] + lane;
        const int width = (int)((slice_ptr[s + 1] - slice_ptr[s]) >> 6);
        int64_t base_sh = 0;
        int width_sh = 0;
        if (r < n_rows) {
            const int32_t p = perm[r];
            base_sh = slice_ptr_sh[p >> 6] + (p & 63);
            width_sh = (int)((slice_ptr_sh[(p >> 6) + 1] - slice_ptr_sh[p >> 6]) >> 6);
        }
        int k2 = 0;         // (both rows are ascending in the SAME order only inside a node class: search from the start)
        for (int k = 0; k < width; ++k) {
            const int64_t e = base + (int64_t)k * FS_SLICE;
            const int32_t c = sell_col[e];
            int32_t out = -1;
            if (r < n_rows && c >= 0) {
                const int32_t q = perm[c];
                for (k2 = 0; k2 < width_sh; ++k2)
                    if (sell_col_sh[base_sh + (int64_t)k2 * FS_SLICE] == q) { out = (int32_t)(base_sh + (int64_t)k2 * FS_SLICE); break; }
                if (out < 0) ++miss;
            }
            emap[e] = out;
        }
    }
    if (miss) atomicAdd(missing, miss);
}

// unit diagonal of the dummy rows (set once: the per-solve copy below never touches them)
__global__ void k_lattice_dummy_diag(int64_t n_sh, const uint8_t* __restrict__ written, const int64_t* __restrict__ slice_ptr_sh,
                                     const int32_t* __restrict__ sell_col_sh, double* __restrict__ val) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_sh; r += stride) {
        if (written[r]) continue;
        const int64_t base = slice_ptr_sh[r >> 6] + (r & 63);
        const int width = (int)((slice_ptr_sh[(r >> 6) + 1] - slice_ptr_sh[r >> 6]) >> 6);
        for (int k = 0; k < width; ++k)
            if (sell_col_sh[base + (int64_t)k * FS_SLICE] == (int32_t)r) val[base + (int64_t)k * FS_SLICE] = 1.0;
    }
}

// The values of the space's matrix into the shadow's storage, once per solve.  GATHER form (imap: shadow entry -> entry of the space's
// storage, -1: padding or a dummy row's unit diagonal, left as they are): the writes are whole lines, and the reads of a slice are two
// interleaved runs of consecutive rows (the vertex rows and the edge rows of a mesh line).  The scatter form it replaces (emap: entry
// of the space -> shadow entry) wrote every second 8-byte word of a line: 3.1 ms per solve at configs[3].
__global__ void k_lattice_copy_values(int64_t n_entries_sh, const int32_t* __restrict__ imap, const double* __restrict__ val, double* __restrict__ val_sh) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n_entries_sh; t += stride) {
        const int32_t e = imap[t];
        if (e >= 0) val_sh[t] = val[e];
    }
}
__global__ void k_lattice_invert_map(int64_t n_entries, const int32_t* __restrict__ emap, int32_t* __restrict__ imap) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < n_entries; e += stride) {
        const int32_t t = emap[e];
        if (t >= 0) imap[t] = (int32_t)e;
    }
}
__global__ void k_lattice_scatter(int64_t n, const int32_t* __restrict__ perm, const double* __restrict__ v, double* __restrict__ v_sh) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) v_sh[perm[i]] = v[i];
}
__global__ void k_lattice_gather(int64_t n, const int32_t* __restrict__ perm, const double* __restrict__ v_sh, double* __restrict__ v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) v[i] = v_sh[perm[i]];
}

static bool lattice_applies(const fs_space_s* sp) {
    const fs_mesh_s* m = sp->mesh;
    return m && sp->degree == 2 && sp->ncomp == 1 && m->tdim == 3 && !sp->halo.active && m->box_n[0] > 0 && m->n_owned == m->nv &&
           sp->n_nodes_owned == sp->n_nodes_local && sp->rowptr.p && sp->colidx.p && sp->edges.p &&
           // (shorter lines: a slice of 64 rows would span more than two of them and could not be a DIA slice)
           2 * m->box_n[0] + 2 >= 64;
}

// The shadow of a space: built on first use, nullptr where it does not apply (or its storage did not come out as DIA slices).
int fs_lattice_get(fs_space_s* sp, fs_lattice_shadow** out) {
    *out = nullptr;
    if (sp->lattice_state < 0) return FS_OK;
    if (sp->lattice_state > 0) { *out = sp->lattice; return FS_OK; }
    sp->lattice_state = -1;
    if (!lattice_applies(sp)) return FS_OK;
    hipStream_t s = fs_rt().stream;
    const fs_mesh_s* m = sp->mesh;
    const int64_t nx = m->box_n[0], ny = m->box_n[1], nz = m->box_n[2];
    const int64_t SX = 2 * nx + 2, NY = 2 * ny + 1, NZ = 2 * nz + 1;
    const int64_t n_sh = SX * NY * NZ, n = sp->n_nodes_owned;
    if (n_sh >= (int64_t)INT32_MAX || n != (2 * nx + 1) * NY * NZ) return FS_OK;
    fs_lattice_shadow* L = new fs_lattice_shadow();
    auto fail = [&](int rc) { delete L; return rc; };
    int rc = FS_OK;
    if ((rc = L->perm.alloc(n)) != FS_OK) return fail(rc);
    hipLaunchKernelGGL(k_lattice_perm, dim3(fs_grid_for(n)), dim3(FS_BLOCK), 0, s, m->nv, n, sp->edges.p, nx + 1, ny + 1, SX, NY, L->perm.p);
    fs_space_s* sh = L->sp = new fs_space_s();
    sh->degree = 2; sh->ncomp = 1;
    sh->n_nodes_owned = sh->n_nodes_local = sh->n_dofs_owned = sh->n_dofs_local = n_sh;
    sh->lattice_state = -1;             // (a shadow has no shadow)
    sh->dict_period = 2;
    sh->dict_line = SX;
    sh->dict_runs = 12;
    sh->lat_ny = (int)NY;
    sh->lat_nz = (int)NZ;
    dbuf<int32_t> len;
    dbuf<int> d_cnt;
    dbuf<uint8_t> written;
    if ((rc = len.alloc(n_sh + 1)) != FS_OK || (rc = len.zero(s)) != FS_OK || (rc = d_cnt.alloc(2)) != FS_OK || (rc = d_cnt.zero(s)) != FS_OK ||
        (rc = written.alloc(n_sh)) != FS_OK || (rc = written.zero(s)) != FS_OK || (rc = sh->rowptr.alloc(n_sh + 1)) != FS_OK) return fail(rc);
    hipLaunchKernelGGL(k_lattice_row_lengths, dim3(fs_grid_for(n)), dim3(FS_BLOCK), 0, s, n, sp->rowptr.p, L->perm.p, len.p);
    hipLaunchKernelGGL(k_lattice_dummy_lengths, dim3(fs_grid_for(n_sh)), dim3(FS_BLOCK), 0, s, n_sh, len.p, d_cnt.p);
    if ((rc = fs_scan_exclusive_i32(len.p, sh->rowptr.p, n_sh + 1, s)) != FS_OK) return fail(rc);
    int h_cnt[2] = {0, 0};
    int32_t h_nnz = 0;
    if ((rc = d_cnt.download(h_cnt, 2, s)) != FS_OK) return fail(rc);
    if (hipMemcpy(&h_nnz, sh->rowptr.p + n_sh, sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return fail(FS_ERR_HIP);
    // every node on its own lattice point, the dummies the only points left over: otherwise the mesh is not the box it is taken for
    if (h_cnt[0] != n_sh - n || (int64_t)h_nnz != sp->nnz_nodes + (n_sh - n)) {
        if (getenv("FS_KRYLOV_DEBUG")) fprintf(stderr, "[fs_lattice] space %llu: nodes do not fill the half grid (%d dummies for %lld) - not used\n",
                                               (unsigned long long)sp->serial, h_cnt[0], (long long)(n_sh - n));
        return fail(FS_OK);
    }
    sh->nnz_nodes = h_nnz;
    if ((rc = sh->colidx.alloc(h_nnz)) != FS_OK) return fail(rc);
    hipLaunchKernelGGL(k_lattice_fill_rows, dim3(fs_grid_for(n * 64, FS_BLOCK, 16384)), dim3(FS_BLOCK), 0, s, n, sp->rowptr.p, sp->colidx.p, L->perm.p, sh->rowptr.p, sh->colidx.p, written.p);
    hipLaunchKernelGGL(k_lattice_fill_dummies, dim3(fs_grid_for(n_sh)), dim3(FS_BLOCK), 0, s, n_sh, sh->rowptr.p, written.p, sh->colidx.p);
    if (hipGetLastError() != hipSuccess) return fail(FS_ERR_HIP);
    if ((rc = fs_space_build_storage(sh, s)) != FS_OK) return fail(rc);
    if (sh->n_dia_slices != sh->n_slices) {         // (the row-dictionary product wants DIA slices throughout)
        if (getenv("FS_KRYLOV_DEBUG")) fprintf(stderr, "[fs_lattice] space %llu: %lld of %lld shadow slices are DIA slices - not used\n",
                                               (unsigned long long)sp->serial, (long long)sh->n_dia_slices, (long long)sh->n_slices);
        return fail(FS_OK);
    }
    L->A = new fs_matrix_s();
    L->A->space = sh;
    L->A->bs = 1;
    if ((rc = L->A->val.alloc(sh->sell_entries)) != FS_OK || (rc = L->A->val.zero(s)) != FS_OK || (rc = L->emap.alloc(sp->sell_entries)) != FS_OK ||
        (rc = L->b.d.alloc(n_sh)) != FS_OK || (rc = L->x.d.alloc(n_sh)) != FS_OK || (rc = L->b.d.zero(s)) != FS_OK || (rc = L->x.d.zero(s)) != FS_OK) return fail(rc);
    hipLaunchKernelGGL(k_lattice_entry_map, dim3(fs_grid_for(sp->n_slices * 64, FS_BLOCK, 16384)), dim3(FS_BLOCK), 0, s, sp->n_slices, n, sp->slice_ptr.p, sp->sell_col.p,
                       L->perm.p, sh->slice_ptr.p, sh->sell_col.p, L->emap.p, d_cnt.p + 1);
    if ((rc = L->imap.alloc(sh->sell_entries)) != FS_OK) return fail(rc);
    if (hipMemsetAsync(L->imap.p, 0xff, (size_t)sh->sell_entries * sizeof(int32_t), s) != hipSuccess) return fail(FS_ERR_HIP);
    hipLaunchKernelGGL(k_lattice_invert_map, dim3(fs_grid_for(sp->sell_entries, FS_BLOCK, 16384)), dim3(FS_BLOCK), 0, s, sp->sell_entries, L->emap.p, L->imap.p);
    hipLaunchKernelGGL(k_lattice_dummy_diag, dim3(fs_grid_for(n_sh)), dim3(FS_BLOCK), 0, s, n_sh, written.p, sh->slice_ptr.p, sh->sell_col.p, L->A->val.p);
    if (hipGetLastError() != hipSuccess) return fail(FS_ERR_HIP);
    if ((rc = d_cnt.download(h_cnt, 2, s)) != FS_OK) return fail(rc);
    if (h_cnt[1] != 0) {
        fs_set_error("fs_lattice: internal error, %d stored entries have no place in the lattice-ordered pattern", h_cnt[1]);
        return fail(FS_ERR_INVALID);
    }
    L->emap.release();          // (read by k_lattice_invert_map only, which the download above has waited for: sell_entries int32s)
    L->n = n_sh;
    if (getenv("FS_KRYLOV_DEBUG") || getenv("FS_SPACE_DEBUG"))
        fprintf(stderr, "[fs_lattice] space %llu: %lld CG2 nodes on the half grid %lld x %lld x %lld (+ %lld dummy rows), %lld stored entries (space: %lld)\n",
                (unsigned long long)sp->serial, (long long)n, (long long)(2 * nx + 1), (long long)NY, (long long)NZ, (long long)(n_sh - n),
                (long long)sh->sell_entries, (long long)sp->sell_entries);
    sp->lattice = L;
    sp->lattice_state = 1;
    *out = L;
    return FS_OK;
}

// A's values, b and the current x into the shadow; the handles the inner solve runs on
int fs_lattice_enter(fs_lattice_shadow* L, fs_matrix_s* A, const fs_vector_s* b, const fs_vector_s* x, bool use_guess, fs_matrix_s** A_sh,
                     fs_vector_s** b_sh, fs_vector_s** x_sh) {
    hipStream_t s = fs_rt().stream;
    const fs_space_s* sp = A->space;
    const int64_t n = sp->n_nodes_owned;
    const int64_t ne_sh = L->A->space->sell_entries;
    hipLaunchKernelGGL(k_lattice_copy_values, dim3(fs_grid_for(ne_sh, FS_BLOCK, 16384)), dim3(FS_BLOCK), 0, s, ne_sh, L->imap.p, A->val.p, L->A->val.p);
    hipLaunchKernelGGL(k_lattice_scatter, dim3(fs_grid_for(n)), dim3(FS_BLOCK), 0, s, n, L->perm.p, b->d.p, L->b.d.p);
    if (use_guess) hipLaunchKernelGGL(k_lattice_scatter, dim3(fs_grid_for(n)), dim3(FS_BLOCK), 0, s, n, L->perm.p, x->d.p, L->x.d.p);
    else FS_CHECK(L->x.d.zero(s));
    FS_KERNEL_CHECK();
    *A_sh = L->A; *b_sh = &L->b; *x_sh = &L->x;
    return FS_OK;
}

int fs_lattice_leave(fs_lattice_shadow* L, const fs_space_s* sp, fs_vector_s* x) {
    hipStream_t s = fs_rt().stream;
    hipLaunchKernelGGL(k_lattice_gather, dim3(fs_grid_for(sp->n_nodes_owned)), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, L->perm.p, L->x.d.p, x->d.p);
    FS_KERNEL_CHECK();
    return FS_OK;
}
