// Numeric assembly of libfsamd.so: per-cell quadrature + local tensors, scatter into
// the SELL-64 matrix through the precomputed slot table, load vectors, boundary-facet
// terms and Dirichlet application.
//
// Stands in for FFC's generated tabulate_tensor + DOLFIN's Assembler/SystemAssembler +
// DirichletBC.apply reached from SolverBase.solve_linear_problem / solve_amg
// (FenicsSolver/SolverBase.py:592-613, 643-672).  All kernels are HBM/L2-bound scatter
// kernels: one thread per cell, 16-B coalesced loads of the cell's vertex ids and of the
// slot column, hardware fp64 atomics (global_atomic_add_f64) for the scatter.
#include "fs_common.h"
#include "fs_kernels.h"

__device__ __forceinline__ int fs_find_pos_local(const int32_t* __restrict__ sell_col, int64_t base_lane, int width, int32_t target) {
    for (int k = 0; k < width; ++k)
        if (sell_col[base_lane + (int64_t)k * FS_SLICE] == target) return k;
    return -1;
}

// ---- P1 geometry ---------------------------------------------------------------------------
struct tet_geom {
    double g[4][3];  // gradients of the barycentric basis
    double adet;     // |det J|
};

// 16 + 8 bytes: the pad of the 32-byte record is not fetched (the gather kernels are bound by the bytes their lanes pull
// through the per-CU address path, not by HBM)
__device__ __forceinline__ void load_vertex(const double* __restrict__ xyz4, int32_t v, double (&x)[3]) {
    const double2 a = reinterpret_cast<const double2*>(xyz4)[2 * (int64_t)v];
    x[0] = a.x; x[1] = a.y; x[2] = xyz4[4 * (int64_t)v + 2];
}

__device__ __forceinline__ tet_geom tet_geometry_x(const double (&x0)[3], const double (&x1)[3], const double (&x2)[3],
                                                   const double (&x3)[3]);
__device__ __forceinline__ tet_geom tet_geometry(const double* __restrict__ xyz4, const int32_t (&v)[4]) {
    double x0[3], x1[3], x2[3], x3[3];
    load_vertex(xyz4, v[0], x0);
    load_vertex(xyz4, v[1], x1);
    load_vertex(xyz4, v[2], x2);
    load_vertex(xyz4, v[3], x3);
    return tet_geometry_x(x0, x1, x2, x3);
}
// snap: grid spacing of a uniform box mesh and its reciprocal (fs_mesh_s::box_h; 0 = general mesh).  An edge-vector component of a
// box cell is -h, 0 or +h up to the rounding of the two coordinates it is the difference of; h * rint(e / h) removes exactly that
// noise, so every cell of the same Kuhn type yields the same bits wherever it sits.
struct box_snap { double h[3], inv[3]; };
__device__ __forceinline__ tet_geom tet_geometry_e(const double (&e1)[3], const double (&e2)[3], const double (&e3)[3]);
__device__ __forceinline__ tet_geom tet_geometry_x(const double (&x0)[3], const double (&x1)[3], const double (&x2)[3],
                                                   const double (&x3)[3]) {
    const double e1[3] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]};
    const double e2[3] = {x2[0] - x0[0], x2[1] - x0[1], x2[2] - x0[2]};
    const double e3[3] = {x3[0] - x0[0], x3[1] - x0[1], x3[2] - x0[2]};
    return tet_geometry_e(e1, e2, e3);
}
__device__ __forceinline__ tet_geom tet_geometry_snapped(const double (&x0)[3], const double (&x1)[3], const double (&x2)[3],
                                                         const double (&x3)[3], const box_snap& bx) {
    double e1[3], e2[3], e3[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        e1[d] = bx.h[d] * rint((x1[d] - x0[d]) * bx.inv[d]);
        e2[d] = bx.h[d] * rint((x2[d] - x0[d]) * bx.inv[d]);
        e3[d] = bx.h[d] * rint((x3[d] - x0[d]) * bx.inv[d]);
    }
    return tet_geometry_e(e1, e2, e3);
}
__device__ __forceinline__ tet_geom tet_geometry_box(const double* __restrict__ xyz4, const int32_t (&v)[4], const box_snap& bx);
static int g_box_snap = -1;          // -1: not set (environment FS_BOX_SNAP, default on)
void fs_set_box_snap(bool on) { g_box_snap = on; }
static int g_box_assembly = -1;      // -1: not set (environment FS_BOX_ASSEMBLY, default on): option "box_assembly"
void fs_set_box_assembly(bool on) { g_box_assembly = on; }
static box_snap make_box_snap(const fs_mesh_s* m) {
    static const bool env_off = getenv("FS_BOX_SNAP") && getenv("FS_BOX_SNAP")[0] == '0';
    const bool off = g_box_snap < 0 ? env_off : g_box_snap == 0;
    box_snap b;
    for (int d = 0; d < 3; ++d) {
        const bool on = !off && m->tdim == 3 && m->box_h[0] > 0.0 && m->box_h[1] > 0.0 && m->box_h[2] > 0.0;
        b.h[d] = on ? m->box_h[d] : 0.0;
        b.inv[d] = on ? 1.0 / m->box_h[d] : 0.0;
    }
    return b;
}
__device__ __forceinline__ tet_geom tet_geometry_e(const double (&e1)[3], const double (&e2)[3], const double (&e3)[3]) {
    // cofactors: grad lambda_1 = (e2 x e3)/det, grad lambda_2 = (e3 x e1)/det, grad lambda_3 = (e1 x e2)/det
    const double c1[3] = {e2[1] * e3[2] - e2[2] * e3[1], e2[2] * e3[0] - e2[0] * e3[2], e2[0] * e3[1] - e2[1] * e3[0]};
    const double c2[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
    const double c3[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double det = e1[0] * c1[0] + e1[1] * c1[1] + e1[2] * c1[2];
    const double inv = 1.0 / det;
    tet_geom t;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        t.g[1][d] = c1[d] * inv;
        t.g[2][d] = c2[d] * inv;
        t.g[3][d] = c3[d] * inv;
        t.g[0][d] = -(t.g[1][d] + t.g[2][d] + t.g[3][d]);
    }
    t.adet = fabs(det);
    return t;
}

// general mesh: the plain geometry; uniform box (bx.h > 0): edge vectors snapped to the grid spacing
__device__ __forceinline__ tet_geom tet_geometry_box(const double* __restrict__ xyz4, const int32_t (&v)[4], const box_snap& bx) {
    double x0[3], x1[3], x2[3], x3[3];
    load_vertex(xyz4, v[0], x0);
    load_vertex(xyz4, v[1], x1);
    load_vertex(xyz4, v[2], x2);
    load_vertex(xyz4, v[3], x3);
    return bx.h[0] > 0.0 ? tet_geometry_snapped(x0, x1, x2, x3, bx) : tet_geometry_x(x0, x1, x2, x3);
}

// SUPG parameter of a cell (ScalarTransportSolver.py:262-266): tau = 0.5 h / (4/(Pe h) + 2 |v|), h = 2 R with R the
// circumradius: R = sqrt((aA+bB+cC)(aA+bB-cC)(aA-bB+cC)(-aA+bB+cC)) / (24 V), (a,A) (b,B) (c,C) opposite edge pairs
__device__ __forceinline__ double supg_tau(const double* __restrict__ xyz4, const int32_t (&v)[4], double adet, double vnorm,
                                           double pe) {
    double x0[3], x1[3], x2[3], x3[3];
    load_vertex(xyz4, v[0], x0);
    load_vertex(xyz4, v[1], x1);
    load_vertex(xyz4, v[2], x2);
    load_vertex(xyz4, v[3], x3);
    auto dist = [](const double (&p)[3], const double (&q)[3]) {
        return sqrt((p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]) + (p[2] - q[2]) * (p[2] - q[2]));
    };
    const double aA = dist(x0, x1) * dist(x2, x3), bB = dist(x0, x2) * dist(x1, x3), cC = dist(x0, x3) * dist(x1, x2);
    const double prod = (aA + bB + cC) * (aA + bB - cC) * (aA - bB + cC) * (-aA + bB + cC);
    const double R = sqrt(prod > 0.0 ? prod : 0.0) / (4.0 * adet);     // 24 V = 4 |det J|
    const double h = 2.0 * R;
    return 0.5 * h / (4.0 / (pe * h) + 2.0 * vnorm);
}

__device__ __forceinline__ double tri_area(const double* __restrict__ xyz4, int32_t a, int32_t b, int32_t c) {
    double x0[3], x1[3], x2[3];
    load_vertex(xyz4, a, x0);
    load_vertex(xyz4, b, x1);
    load_vertex(xyz4, c, x2);
    const double u[3] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]};
    const double w[3] = {x2[0] - x0[0], x2[1] - x0[1], x2[2] - x0[2]};
    const double n[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
    return 0.5 * sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
}

struct coef_dev {
    int mode;
    double value;
    const double* data;
    double tensor[9];
};

// ---- scalar P1:  Ke = k vol grad_a.K.grad_b + m |J|/120 (1+delta_ab) ------------------------
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p1_scalar(const int32_t* __restrict__ cells,
                                                                 const double* __restrict__ xyz4,
                                                                 const int32_t* __restrict__ slots, int64_t nc,
                                                                 coef_dev kc, coef_dev mc,
                                                                 double* __restrict__ val) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < nc; c += stride) {
        const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
        const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
        const tet_geom t = tet_geometry(xyz4, v);
        const double vol = t.adet * (1.0 / 6.0);
        double ke[4][4];
        if (kc.mode == FS_COEF_TENSOR) {
            double kg[4][3];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    kg[a][i] = kc.tensor[3 * i + 0] * t.g[a][0] + kc.tensor[3 * i + 1] * t.g[a][1] + kc.tensor[3 * i + 2] * t.g[a][2];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    ke[a][b] = vol * (t.g[a][0] * kg[b][0] + t.g[a][1] * kg[b][1] + t.g[a][2] * kg[b][2]);
        } else {
            double kk = 0.0;
            if (kc.mode == FS_COEF_CONST) kk = kc.value;
            else if (kc.mode == FS_COEF_CELL) kk = kc.data[c];
            const double w = kk * vol;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = a; b < 4; ++b) {
                    const double x = w * (t.g[a][0] * t.g[b][0] + t.g[a][1] * t.g[b][1] + t.g[a][2] * t.g[b][2]);
                    ke[a][b] = x;
                    ke[b][a] = x;
                }
        }
        if (mc.mode != FS_COEF_NONE) {
            const double mm = (mc.mode == FS_COEF_CONST ? mc.value : mc.data[c]) * t.adet * (1.0 / 120.0);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) ke[a][b] += (a == b ? 2.0 : 1.0) * mm;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int32_t slot = slots[(int64_t)(a * 4 + b) * nc + c];
                if (slot >= 0) atomicAdd(&val[slot], ke[a][b]);
            }
    }
}

// ---- scalar P1, row-gather form: owner computes, no atomics, deterministic -----------------------
// One wavefront per SELL slice, lane = row.  Every row walks its (cell, local vertex) incidences
// (SELL-laid-out tables built by fs_space_create), recomputes the cell geometry with its own vertex
// rotated to local index 0 (so no register array is indexed dynamically), and accumulates the row of
// the local matrix into a thread-private LDS column at the precomputed in-row positions.  The row is
// then written with plain, fully coalesced stores.  Contributions are summed in ascending cell order,
// so the assembled matrix is bit-reproducible.
// Loads are issued ahead of their use: the incidence records of the next eight rounds while the current eight are worked
// on, the cell records of a group before its first coordinate load; the row's own vertex is read once, coordinates as
// 16 + 8 bytes.  10 M DOF: 2.94 -> 2.06 ms, 1 M DOF: 0.326 -> 0.241 ms (occupancy 3 waves per SIMD at 167 VGPRs; what was
// waited for were the three dependent loads record -> cell -> coordinates of every incidence).  Staging the coordinates
// of the row's columns in LDS instead (a fifth of the bytes through the address path) was SLOWER, 3.4 ms: 30 KB of
// LDS per wave leave one wave per SIMD (tools/experiments/r02_assemble_xlds.patch).
// MODE 0: A = form, 1: A += form, 2: matrix-free product val[row] = (form x)[row] - the same walk with the row of the
// local matrix multiplied into x instead of stored (fs_operator_apply; the operator is never formed, no LDS).
template <int MODE>
__global__ void __launch_bounds__(FS_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3))) k_assemble_p1_scalar_gather(
    int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ slice_ptr,
    const int64_t* __restrict__ inc_slice_ptr, const int32_t* __restrict__ inc_cell,
    const uint32_t* __restrict__ inc_pos, const int32_t* __restrict__ cells, const double* __restrict__ xyz4,
    coef_dev kc, coef_dev mc, coef_dev ac, double ascale, double supg_pe, double* __restrict__ val,
    const int32_t* __restrict__ order, const box_snap bx, const double* __restrict__ xvec = nullptr) {
    constexpr bool ADD = MODE == 1, APPLY = MODE == 2;
    const bool snapped = bx.h[0] > 0.0;
    extern __shared__ __attribute__((aligned(16))) double lds_acc[];  // [width][blockDim.x]
    const int tid = threadIdx.x, bd = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, wpb = bd >> 6;
    // XCD-contiguous chunk map (see xcd_chunks): the rows an XCD assembles reference a narrow band of
    // vertices, so their coordinates stay in that XCD's L2 instead of being re-fetched over the fabric
    const int64_t n_chunks = (n_slices + wpb - 1) / wpb;
    for (chunk_iter it = xcd_chunks(n_chunks); it.cur < it.end; it.cur += it.step) {
        const int64_t q0 = it.cur * wpb + wave;
        if (q0 >= n_slices) continue;
        const int64_t s = order ? order[q0] : q0;      // spatial order of the slices (fs_space_s::slice_order)
        const int64_t base = slice_ptr[s];
        const int width = (int)((slice_ptr[s + 1] - base) >> 6);
        const int64_t ibase = inc_slice_ptr[s];
        const int iwidth = (int)((inc_slice_ptr[s + 1] - ibase) >> 6);
        if (!APPLY)
            for (int k = 0; k < width; ++k) lds_acc[k * bd + tid] = 0.0;
        double xown[3] = {0.0, 0.0, 0.0};
        double yacc = 0.0, x_own = 0.0;
        if (s * FS_SLICE + lane < n_rows) {
            load_vertex(xyz4, (int32_t)(s * FS_SLICE + lane), xown);
            if (APPLY) x_own = xvec[s * FS_SLICE + lane];
        }
        // the incidence records of the next rounds are fetched ahead
        constexpr int PF = 8;
        int32_t qn[PF];
        uint32_t pn[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            qn[u] = u < iwidth ? inc_cell[ibase + (int64_t)u * FS_SLICE + lane] : -1;
            pn[u] = (u < iwidth) ? inc_pos[ibase + (int64_t)u * FS_SLICE + lane] : 0u;
        }
        for (int j0 = 0; j0 < iwidth; j0 += PF) {
          int32_t qc[PF];
          uint32_t pc[PF];
#pragma unroll
          for (int u = 0; u < PF; ++u) { qc[u] = qn[u]; pc[u] = pn[u]; }
#pragma unroll
          for (int u = 0; u < PF; ++u) {
              const int jn = j0 + PF + u;
              qn[u] = jn < iwidth ? inc_cell[ibase + (int64_t)jn * FS_SLICE + lane] : -1;
              pn[u] = jn < iwidth ? inc_pos[ibase + (int64_t)jn * FS_SLICE + lane] : 0u;
          }
          int4 vc[PF];
          // the cell records of the whole group go out before the first coordinate load
#pragma unroll
          for (int u = 0; u < PF; ++u) vc[u] = qc[u] >= 0 ? reinterpret_cast<const int4*>(cells)[qc[u] >> 2] : make_int4(0, 0, 0, 0);
#pragma unroll
          for (int u = 0; u < PF; ++u) {
            if (j0 + u >= iwidth) break;          // (also keeps the compiler from hoisting all eight rounds' coordinate loads)
            const int32_t q = qc[u];
            if (q < 0) continue;
            const uint32_t packed = pc[u];
            const int c = q >> 2, a = q & 3;
            // rotate so that this row's vertex is local vertex 0 (gradients are orientation-free)
            const uint32_t prot = a == 0 ? packed : ((packed >> (8 * a)) | (packed << (32 - 8 * a)));
            const int4 v4 = vc[u];
            int32_t vv[4];
            vv[0] = a == 0 ? v4.x : a == 1 ? v4.y : a == 2 ? v4.z : v4.w;      // = this row's vertex
            vv[1] = a == 0 ? v4.y : a == 1 ? v4.z : a == 2 ? v4.w : v4.x;
            vv[2] = a == 0 ? v4.z : a == 1 ? v4.w : a == 2 ? v4.x : v4.y;
            vv[3] = a == 0 ? v4.w : a == 1 ? v4.x : a == 2 ? v4.y : v4.z;
            double x1[3], x2[3], x3[3];
            load_vertex(xyz4, vv[1], x1);
            load_vertex(xyz4, vv[2], x2);
            load_vertex(xyz4, vv[3], x3);
            double xb[3] = {0.0, 0.0, 0.0};
            if (APPLY) { xb[0] = xvec[vv[1]]; xb[1] = xvec[vv[2]]; xb[2] = xvec[vv[3]]; }
            // (the row's own coordinates were loaded once; uniform box: edge vectors snapped to the grid spacing)
            const tet_geom t = snapped ? tet_geometry_snapped(xown, x1, x2, x3, bx) : tet_geometry_x(xown, x1, x2, x3);
            const double vol = t.adet * (1.0 / 6.0);
            double row[4];
            if (kc.mode == FS_COEF_TENSOR) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    double kg[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        kg[i] = kc.tensor[3 * i + 0] * t.g[b][0] + kc.tensor[3 * i + 1] * t.g[b][1] + kc.tensor[3 * i + 2] * t.g[b][2];
                    row[b] = vol * (t.g[0][0] * kg[0] + t.g[0][1] * kg[1] + t.g[0][2] * kg[2]);
                }
            } else if (kc.mode == FS_COEF_CELL_TENSOR) {      // one tensor per cell (an Expression of degree 0)
                const double* __restrict__ T = kc.data + 9 * (int64_t)c;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    double kg[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) kg[i] = T[3 * i + 0] * t.g[b][0] + T[3 * i + 1] * t.g[b][1] + T[3 * i + 2] * t.g[b][2];
                    row[b] = vol * (t.g[0][0] * kg[0] + t.g[0][1] * kg[1] + t.g[0][2] * kg[2]);
                }
            } else {
                double kk = 0.0;
                if (kc.mode == FS_COEF_CONST) kk = kc.value;
                else if (kc.mode == FS_COEF_CELL) kk = kc.data[c];
                const double w = kk * vol;
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    row[b] = w * (t.g[0][0] * t.g[b][0] + t.g[0][1] * t.g[b][1] + t.g[0][2] * t.g[b][2]);
            }
            if (mc.mode != FS_COEF_NONE) {
                const double mm = (mc.mode == FS_COEF_CONST ? mc.value : mc.data[c]) * t.adet * (1.0 / 120.0);
                row[0] += 2.0 * mm;
                row[1] += mm;
                row[2] += mm;
                row[3] += mm;
            }
            if (ac.mode != FS_COEF_NONE) {
                // Galerkin advection: C_ab = scale * (vol/4) * (v . grad phi_b), the same for every row a
                double vx, vy, vz;
                if (ac.mode == FS_COEF_CONST) { vx = ac.tensor[0]; vy = ac.tensor[1]; vz = ac.tensor[2]; }
                else {      // per cell, or per (cell, test function a) - the exact weights of a finite-element velocity
                    const int64_t o = ac.mode == FS_COEF_CELL_ROW ? 3 * (4 * (int64_t)c + a) : 3 * (int64_t)c;
                    vx = ac.data[o]; vy = ac.data[o + 1]; vz = ac.data[o + 2];
                }
                const double w4 = ascale * vol * 0.25;
#pragma unroll
                for (int b = 0; b < 4; ++b) row[b] += w4 * (vx * t.g[b][0] + vy * t.g[b][1] + vz * t.g[b][2]);
                if (supg_pe > 0.0) {
                    // test function q + tau (v . grad q): this row's vertex is local vertex 0 after the rotation
                    const double tau = supg_tau(xyz4, vv, t.adet, sqrt(vx * vx + vy * vy + vz * vz), supg_pe);
                    const double wa = tau * (vx * t.g[0][0] + vy * t.g[0][1] + vz * t.g[0][2]);
                    const double mval = mc.mode == FS_COEF_NONE ? 0.0 : (mc.mode == FS_COEF_CONST ? mc.value : mc.data[c]);
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        row[b] += wa * vol * (ascale * (vx * t.g[b][0] + vy * t.g[b][1] + vz * t.g[b][2]) + 0.25 * mval);
                }
            }
            if (APPLY) {
                yacc += row[0] * x_own + row[1] * xb[0] + row[2] * xb[1] + row[3] * xb[2];
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int k = (prot >> (8 * b)) & 255;
                    lds_acc[k * bd + tid] += row[b];
                }
            }
          }
        }
        if (APPLY) {
            if (s * FS_SLICE + lane < n_rows) val[s * FS_SLICE + lane] = yacc;
            continue;
        }
        for (int k = 0; k < width; ++k) {
            const int64_t e = base + (int64_t)k * FS_SLICE + lane;
            const double x = lds_acc[k * bd + tid];
            val[e] = ADD ? val[e] + x : x;
        }
    }
}

// ---- scalar P1 on a BOX mesh: the row-gather assembly without the geometry (round 6) ------------------------------------------------
// k_assemble_p1_scalar_gather recomputes a tetrahedron's four gradients for every (row, cell) incidence - 24 x about 170 vector
// instructions, three coordinate gathers and a cell record per incidence: 2.26 ms at 10 M rows, 0.13 of the HBM peak on its 240 B/row,
// counter traffic 2.8 x the required bytes.  On a mesh made by fs_mesh_create_box the SNAPPED edge vectors of a cell
// are a function of its TYPE alone (cell c of a hexahedron is type c % 6: k_box_cells) and so are |det J| and the products g_a . g_b:
// 6 types x 4 row vertices x (|det J|, four products) - 120 doubles, computed once per mesh by k_box_ref_rows with the very calls and
// expressions of the general kernel (same rotation of the row's vertex to local index 0, same tet_geometry_snapped, same sum), hence
// the same bits (tests/test_gpu_kernels.py::test_box_assembly_fast_path_bits compares every stored value).  Per incidence: the record
// (cell, positions: 8 B), the type's five numbers from LDS, the coefficient, four products.  Constant or per-cell scalar stiffness
// and mass coefficients; everything else (tensor conductivities, advection, SUPG, the matrix-free apply) stays with the general kernel.
__global__ void k_box_ref_rows(const int32_t* __restrict__ cells, const double* __restrict__ xyz4, const box_snap bx, double* __restrict__ ref) {
    const int i = threadIdx.x;
    if (i >= 24) return;
    const int ty = i >> 2, a = i & 3;
    const int4 v4 = reinterpret_cast<const int4*>(cells)[ty];
    int32_t vv[4];
    vv[0] = a == 0 ? v4.x : a == 1 ? v4.y : a == 2 ? v4.z : v4.w;
    vv[1] = a == 0 ? v4.y : a == 1 ? v4.z : a == 2 ? v4.w : v4.x;
    vv[2] = a == 0 ? v4.z : a == 1 ? v4.w : a == 2 ? v4.x : v4.y;
    vv[3] = a == 0 ? v4.w : a == 1 ? v4.x : a == 2 ? v4.y : v4.z;
    double x0[3], x1[3], x2[3], x3[3];
    load_vertex(xyz4, vv[0], x0);
    load_vertex(xyz4, vv[1], x1);
    load_vertex(xyz4, vv[2], x2);
    load_vertex(xyz4, vv[3], x3);
    const tet_geom t = tet_geometry_snapped(x0, x1, x2, x3, bx);
    ref[5 * i] = t.adet;
#pragma unroll
    for (int b = 0; b < 4; ++b) ref[5 * i + 1 + b] = t.g[0][0] * t.g[b][0] + t.g[0][1] * t.g[b][1] + t.g[0][2] * t.g[b][2];
}

template <bool ADD>
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p1_box_gather(
    int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ slice_ptr, const int64_t* __restrict__ inc_slice_ptr,
    const int32_t* __restrict__ inc_cell, const uint32_t* __restrict__ inc_pos, const double* __restrict__ ref,
    coef_dev kc, coef_dev mc, double* __restrict__ val, const int32_t* __restrict__ order, int acc_doubles) {
    extern __shared__ __attribute__((aligned(16))) double lds_acc[];  // [max_row][blockDim.x], then the 120 reference numbers
    const int tid = threadIdx.x, bd = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, wpb = bd >> 6;
    double* __restrict__ rf = lds_acc + acc_doubles;
    if (tid < 120) rf[tid] = ref[tid];
    __syncthreads();
    const int64_t n_chunks = (n_slices + wpb - 1) / wpb;
    for (chunk_iter it = xcd_chunks(n_chunks); it.cur < it.end; it.cur += it.step) {
        const int64_t q0 = it.cur * wpb + wave;
        if (q0 >= n_slices) continue;
        const int64_t s = order ? order[q0] : q0;
        const int64_t base = slice_ptr[s];
        const int width = (int)((slice_ptr[s + 1] - base) >> 6);
        const int64_t ibase = inc_slice_ptr[s];
        const int iwidth = (int)((inc_slice_ptr[s + 1] - ibase) >> 6);
        for (int k = 0; k < width; ++k) lds_acc[k * bd + tid] = 0.0;
        constexpr int PF = 8;
        for (int j0 = 0; j0 < iwidth; j0 += PF) {
            int32_t qc[PF];
            uint32_t pc[PF];
            double kk[PF], mv[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int j = j0 + u;
                qc[u] = j < iwidth ? inc_cell[ibase + (int64_t)j * FS_SLICE + lane] : -1;
                pc[u] = j < iwidth ? inc_pos[ibase + (int64_t)j * FS_SLICE + lane] : 0u;
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int c = qc[u] >= 0 ? qc[u] >> 2 : 0;
                kk[u] = kc.mode == FS_COEF_CONST ? kc.value : (kc.mode == FS_COEF_CELL ? kc.data[c] : 0.0);
                mv[u] = mc.mode == FS_COEF_CONST ? mc.value : (mc.mode == FS_COEF_CELL ? mc.data[c] : 0.0);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int32_t q = qc[u];
                if (q < 0) continue;
                const uint32_t packed = pc[u];
                const int c = q >> 2, a = q & 3;
                const uint32_t prot = a == 0 ? packed : ((packed >> (8 * a)) | (packed << (32 - 8 * a)));
                const double* __restrict__ R = rf + 5 * (4 * (c % 6) + a);
                const double adet = R[0];
                const double vol = adet * (1.0 / 6.0);
                const double w = kk[u] * vol;
                double row[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) row[b] = w * R[1 + b];
                if (mc.mode != FS_COEF_NONE) {
                    const double mm = mv[u] * adet * (1.0 / 120.0);
                    row[0] += 2.0 * mm;
                    row[1] += mm;
                    row[2] += mm;
                    row[3] += mm;
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int k = (prot >> (8 * b)) & 255;
                    lds_acc[k * bd + tid] += row[b];
                }
            }
        }
        for (int k = 0; k < width; ++k) {
            const int64_t e = base + (int64_t)k * FS_SLICE + lane;
            const double x = lds_acc[k * bd + tid];
            val[e] = ADD ? val[e] + x : x;
        }
    }
}

// ---- scalar P2, row-gather form --------------------------------------------------------------------
// Local dofs: 4 vertices, then the 6 UFC edges.  grad phi_vertex_i = (4 lambda_i - 1) grad lambda_i,
// grad phi_edge_ij = 4 (lambda_i grad lambda_j + lambda_j grad lambda_i); the stiffness integrand is
// quadratic, integrated exactly by the 4-point rule FFC picks for it.
__device__ __constant__ double FS_P2_QP[4][4] = {
    {0.5854101966249685, 0.1381966011250105, 0.1381966011250105, 0.1381966011250105},
    {0.1381966011250105, 0.5854101966249685, 0.1381966011250105, 0.1381966011250105},
    {0.1381966011250105, 0.1381966011250105, 0.5854101966249685, 0.1381966011250105},
    {0.1381966011250105, 0.1381966011250105, 0.1381966011250105, 0.5854101966249685}};
// exact P2 mass matrix of a unit-volume tetrahedron times 420
__device__ __constant__ double FS_P2_MASS420[10][10] = {
    {6, 1, 1, 1, -6, -6, -6, -4, -4, -4},   {1, 6, 1, 1, -6, -4, -4, -6, -6, -4},
    {1, 1, 6, 1, -4, -6, -4, -6, -4, -6},   {1, 1, 1, 6, -4, -4, -6, -4, -6, -6},
    {-6, -6, -4, -4, 32, 16, 16, 16, 16, 8}, {-6, -4, -6, -4, 16, 32, 16, 16, 8, 16},
    {-6, -4, -4, -6, 16, 16, 32, 8, 16, 16}, {-4, -6, -6, -4, 16, 16, 8, 32, 16, 16},
    {-4, -6, -4, -6, 16, 8, 16, 16, 32, 16}, {-4, -4, -6, -6, 8, 16, 16, 16, 16, 32}};

__device__ __forceinline__ void p2_basis_grads(const tet_geom& t, const double (&lam)[4], double (&gp)[10][3]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int d = 0; d < 3; ++d) gp[i][d] = (4.0 * lam[i] - 1.0) * t.g[i][d];
    // UFC edges: e0=(2,3) e1=(1,3) e2=(1,2) e3=(0,3) e4=(0,2) e5=(0,1)
    const int ei[6] = {2, 1, 1, 0, 0, 0}, ej[6] = {3, 3, 2, 3, 2, 1};
#pragma unroll
    for (int e = 0; e < 6; ++e)
#pragma unroll
        for (int d = 0; d < 3; ++d) gp[4 + e][d] = 4.0 * (lam[ei[e]] * t.g[ej[e]][d] + lam[ej[e]] * t.g[ei[e]][d]);
}

// degree-3 rule for the cubic integrand of the CG2 advection term phi_a (v . grad phi_b): Keast's 5 points (one negative weight)
__device__ __constant__ double FS_TET5_QP[5][4] = {{0.25, 0.25, 0.25, 0.25},
                                                   {0.5, 1.0 / 6.0, 1.0 / 6.0, 1.0 / 6.0}, {1.0 / 6.0, 0.5, 1.0 / 6.0, 1.0 / 6.0},
                                                   {1.0 / 6.0, 1.0 / 6.0, 0.5, 1.0 / 6.0}, {1.0 / 6.0, 1.0 / 6.0, 1.0 / 6.0, 0.5}};
__device__ __constant__ double FS_TET5_QW[5] = {-0.8, 0.45, 0.45, 0.45, 0.45};

// the 14-point degree-5 rule (the one the Taylor-Hood element uses): for a coefficient given at its points (FS_COEF_CELL_QP)
__device__ __constant__ double FS_TET14_QP[14][4] = {
    {0.0673422422100982, 0.3108859192633006, 0.3108859192633006, 0.3108859192633006},
    {0.3108859192633006, 0.0673422422100982, 0.3108859192633006, 0.3108859192633006},
    {0.3108859192633006, 0.3108859192633006, 0.0673422422100982, 0.3108859192633006},
    {0.3108859192633006, 0.3108859192633006, 0.3108859192633006, 0.0673422422100982},
    {0.7217942490673264, 0.0927352503108912, 0.0927352503108912, 0.0927352503108912},
    {0.0927352503108912, 0.7217942490673264, 0.0927352503108912, 0.0927352503108912},
    {0.0927352503108912, 0.0927352503108912, 0.7217942490673264, 0.0927352503108912},
    {0.0927352503108912, 0.0927352503108912, 0.0927352503108912, 0.7217942490673264},
    {0.0455037041256496, 0.0455037041256496, 0.4544962958743504, 0.4544962958743504},
    {0.0455037041256496, 0.4544962958743504, 0.0455037041256496, 0.4544962958743504},
    {0.0455037041256496, 0.4544962958743504, 0.4544962958743504, 0.0455037041256496},
    {0.4544962958743504, 0.0455037041256496, 0.0455037041256496, 0.4544962958743504},
    {0.4544962958743504, 0.0455037041256496, 0.4544962958743504, 0.0455037041256496},
    {0.4544962958743504, 0.4544962958743504, 0.0455037041256496, 0.0455037041256496}};
__device__ __constant__ double FS_TET14_QW[14] = {
    0.1126879257180159, 0.1126879257180159, 0.1126879257180159, 0.1126879257180159,
    0.0734930431163620, 0.0734930431163620, 0.0734930431163620, 0.0734930431163620,
    0.0425460207770815, 0.0425460207770815, 0.0425460207770815, 0.0425460207770815, 0.0425460207770815, 0.0425460207770815};

// ADV: + scale * int phi_a (v . grad phi_b) dx with a constant or per-cell velocity (inner(velocity, grad(T))*Tq*capacity*dx,
// ScalarTransportSolver.py:305-311, with fe_degree 2); a separate instantiation, the symmetric kernel keeps its registers
// APPLY (round 6, fs_operator_apply on CG2 spaces): the matrix-free product val[row] = (form x)[row] - the same walk with the row of
// the local matrix multiplied into the x values of the cell's ten nodes (cell_dofs) instead of stored; no LDS accumulator.
template <bool ADD, bool ADV = false, bool APPLY = false>
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p2_scalar_gather(
    int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ slice_ptr,
    const int64_t* __restrict__ inc_slice_ptr, int64_t inc_entries, const int32_t* __restrict__ inc_cell,
    const uint32_t* __restrict__ inc_pos, const int32_t* __restrict__ cells, const double* __restrict__ xyz4,
    coef_dev kc, coef_dev mc, double* __restrict__ val, const int32_t* __restrict__ order, const box_snap bx,
    coef_dev ac = coef_dev(), double ascale = 0.0, double supg_pe = 0.0,
    const int32_t* __restrict__ cell_dofs = nullptr, const double* __restrict__ xvec = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double lds_acc[];  // [width][blockDim.x]
    const int tid = threadIdx.x, bd = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, wpb = bd >> 6;
    const int64_t n_chunks = (n_slices + wpb - 1) / wpb;
    for (chunk_iter it = xcd_chunks(n_chunks); it.cur < it.end; it.cur += it.step) {
        const int64_t q = it.cur * wpb + wave;
        if (q >= n_slices) continue;
        const int64_t s = order ? order[q] : q;      // spatial order of the slices (fs_space_s::slice_order)
        const int64_t base = slice_ptr[s];
        const int width = (int)((slice_ptr[s + 1] - base) >> 6);
        const int64_t ibase = inc_slice_ptr[s];
        const int iwidth = (int)((inc_slice_ptr[s + 1] - ibase) >> 6);
        if (!APPLY)
            for (int k = 0; k < width; ++k) lds_acc[k * bd + tid] = 0.0;
        double yacc = 0.0;
        // loads issued ahead of their use, as in the P1 kernel: the records of the next group while this one is worked on,
        // the cell records of a group before its first coordinate load
        constexpr int PF = 4;
        int32_t qn[PF];
        uint32_t pn[PF][3];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int64_t e = ibase + (int64_t)u * FS_SLICE + lane;
            qn[u] = u < iwidth ? inc_cell[e] : -1;
#pragma unroll
            for (int w = 0; w < 3; ++w) pn[u][w] = u < iwidth ? inc_pos[w * inc_entries + e] : 0u;
        }
        for (int j0 = 0; j0 < iwidth; j0 += PF) {
          int32_t qc[PF];
          uint32_t pc[PF][3];
          int4 vc[PF];
#pragma unroll
          for (int u = 0; u < PF; ++u) {
              qc[u] = qn[u];
#pragma unroll
              for (int w = 0; w < 3; ++w) pc[u][w] = pn[u][w];
          }
#pragma unroll
          for (int u = 0; u < PF; ++u) {
              const int jn = j0 + PF + u;
              const int64_t e = ibase + (int64_t)jn * FS_SLICE + lane;
              qn[u] = jn < iwidth ? inc_cell[e] : -1;
#pragma unroll
              for (int w = 0; w < 3; ++w) pn[u][w] = jn < iwidth ? inc_pos[w * inc_entries + e] : 0u;
          }
#pragma unroll
          for (int u = 0; u < PF; ++u) vc[u] = qc[u] >= 0 ? reinterpret_cast<const int4*>(cells)[qc[u] / 10] : make_int4(0, 0, 0, 0);
#pragma unroll
          for (int u = 0; u < PF; ++u) {
            if (j0 + u >= iwidth) break;
            const int32_t q = qc[u];
            if (q < 0) continue;
            const int c = q / 10, a = q - 10 * c;
            const uint32_t pw[3] = {pc[u][0], pc[u][1], pc[u][2]};
            const int4 c4 = vc[u];      // vertex ids (node ids differ once ghosts exist)
            const int32_t vv[4] = {c4.x, c4.y, c4.z, c4.w};
            const tet_geom t = tet_geometry_box(xyz4, vv, bx);
            const double vol = t.adet * (1.0 / 6.0);
            double row[10];
#pragma unroll
            for (int b = 0; b < 10; ++b) row[b] = 0.0;
            // SUPG (supg_pe > 0, ScalarTransportSolver.py:259-270): this row's test function is q_a + tau (v . grad q_a) in every
            // term; its gradient is grad q_a + tau H_a v with the constant Hessian H_a of the quadratic q_a (vertex i:
            // 4 g_i g_i^T, edge ij: 4 (g_i g_j^T + g_j g_i^T)), v and tau constant on the cell
            double vx = 0.0, vy = 0.0, vz = 0.0, tau = 0.0, hv[3] = {0.0, 0.0, 0.0};
            if (ADV && ac.mode != FS_COEF_NONE) {
                if (ac.mode == FS_COEF_CONST) { vx = ac.tensor[0]; vy = ac.tensor[1]; vz = ac.tensor[2]; }
                else { vx = ac.data[3 * (int64_t)c]; vy = ac.data[3 * (int64_t)c + 1]; vz = ac.data[3 * (int64_t)c + 2]; }
                if (supg_pe > 0.0) {
                    tau = supg_tau(xyz4, vv, t.adet, sqrt(vx * vx + vy * vy + vz * vz), supg_pe);
                    double gv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) gv[i] = t.g[i][0] * vx + t.g[i][1] * vy + t.g[i][2] * vz;
                    const int hi[6] = {2, 1, 1, 0, 0, 0}, hj[6] = {3, 3, 2, 3, 2, 1};
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (b == a)
#pragma unroll
                            for (int d = 0; d < 3; ++d) hv[d] = 4.0 * tau * gv[b] * t.g[b][d];
#pragma unroll
                    for (int e = 0; e < 6; ++e)
                        if (4 + e == a)
#pragma unroll
                            for (int d = 0; d < 3; ++d) hv[d] = 4.0 * tau * (gv[hj[e]] * t.g[hi[e]][d] + gv[hi[e]] * t.g[hj[e]][d]);
                }
            }
            if (ADV && kc.mode == FS_COEF_CELL_QP) {      // (the non-symmetric instantiation also carries the rarely used modes)
                for (int qp = 0; qp < 14; ++qp) {
                    const double lam[4] = {FS_TET14_QP[qp][0], FS_TET14_QP[qp][1], FS_TET14_QP[qp][2], FS_TET14_QP[qp][3]};
                    double gp[10][3];
                    p2_basis_grads(t, lam, gp);
                    double ga[3] = {0.0, 0.0, 0.0};
#pragma unroll
                    for (int b = 0; b < 10; ++b)
                        if (b == a) { ga[0] = gp[b][0] + hv[0]; ga[1] = gp[b][1] + hv[1]; ga[2] = gp[b][2] + hv[2]; }
                    const double w = FS_TET14_QW[qp] * vol * kc.data[14 * (int64_t)c + qp];
#pragma unroll
                    for (int b = 0; b < 10; ++b) row[b] += w * (ga[0] * gp[b][0] + ga[1] * gp[b][1] + ga[2] * gp[b][2]);
                }
            } else if (kc.mode != FS_COEF_NONE) {
                double kk = kc.mode == FS_COEF_CONST ? kc.value : kc.data[c];
#pragma unroll
                for (int qp = 0; qp < 4; ++qp) {
                    const double lam[4] = {FS_P2_QP[qp][0], FS_P2_QP[qp][1], FS_P2_QP[qp][2], FS_P2_QP[qp][3]};
                    double gp[10][3];
                    p2_basis_grads(t, lam, gp);
                    double ga[3] = {0.0, 0.0, 0.0};   // gradient of this row's basis function, selected without
#pragma unroll                                         // indexing the register array dynamically
                    for (int b = 0; b < 10; ++b)
                        if (b == a) { ga[0] = gp[b][0]; ga[1] = gp[b][1]; ga[2] = gp[b][2]; }
                    if (ADV) { ga[0] += hv[0]; ga[1] += hv[1]; ga[2] += hv[2]; }
#pragma unroll
                    for (int b = 0; b < 10; ++b) row[b] += 0.25 * (ga[0] * gp[b][0] + ga[1] * gp[b][1] + ga[2] * gp[b][2]);
                }
                const double w = kk * vol;
#pragma unroll
                for (int b = 0; b < 10; ++b) row[b] *= w;
            }
            if (mc.mode != FS_COEF_NONE) {
                const double mm = (mc.mode == FS_COEF_CONST ? mc.value : mc.data[c]) * vol * (1.0 / 420.0);
#pragma unroll
                for (int b = 0; b < 10; ++b) row[b] += mm * FS_P2_MASS420[a][b];
            }
            if (ADV && ac.mode != FS_COEF_NONE) {
                const double msupg = (tau != 0.0 && mc.mode != FS_COEF_NONE) ? tau * (mc.mode == FS_COEF_CONST ? mc.value : mc.data[c]) : 0.0;
                for (int qp = 0; qp < 5; ++qp) {
                    const double lam[4] = {FS_TET5_QP[qp][0], FS_TET5_QP[qp][1], FS_TET5_QP[qp][2], FS_TET5_QP[qp][3]};
                    double gp[10][3];
                    p2_basis_grads(t, lam, gp);
                    // values of the basis functions: vertex lambda (2 lambda - 1), edge 4 lambda_i lambda_j (UFC edges)
                    const int ei[6] = {2, 1, 1, 0, 0, 0}, ej[6] = {3, 3, 2, 3, 2, 1};
                    double pb[10];
#pragma unroll
                    for (int b = 0; b < 4; ++b) pb[b] = lam[b] * (2.0 * lam[b] - 1.0);
#pragma unroll
                    for (int e = 0; e < 6; ++e) pb[4 + e] = 4.0 * lam[ei[e]] * lam[ej[e]];
                    double pa = 0.0, va = 0.0;      // this row's q_a and v . grad q_a at the point
#pragma unroll
                    for (int b = 0; b < 10; ++b)
                        if (b == a) { pa = pb[b]; va = vx * gp[b][0] + vy * gp[b][1] + vz * gp[b][2]; }
                    const double wq = FS_TET5_QW[qp] * vol;
                    const double w = ascale * wq * (pa + tau * va);
                    const double wm = msupg * wq * va;          // mass term against tau (v . grad q_a): cubic, this rule is exact
#pragma unroll
                    for (int b = 0; b < 10; ++b) row[b] += w * (vx * gp[b][0] + vy * gp[b][1] + vz * gp[b][2]) + wm * pb[b];
                }
            }
            if (APPLY) {
#pragma unroll
                for (int b = 0; b < 10; ++b) yacc += row[b] * xvec[cell_dofs[(int64_t)c * 10 + b]];
            } else {
#pragma unroll
                for (int b = 0; b < 10; ++b) {
                    const int k = (pw[b >> 2] >> (8 * (b & 3))) & 255;
                    lds_acc[k * bd + tid] += row[b];
                }
            }
          }
        }
        if (APPLY) {
            if (s * FS_SLICE + lane < n_rows) val[s * FS_SLICE + lane] = yacc;
            continue;
        }
        for (int k = 0; k < width; ++k) {
            const int64_t e = base + (int64_t)k * FS_SLICE + lane;
            const double x = lds_acc[k * bd + tid];
            val[e] = ADD ? val[e] + x : x;
        }
    }
}

// ---- scalar CG2 on a BOX mesh without the geometry (round 6; see k_assemble_p1_box_gather) ------------------------------------------
// Reference per (cell type, local dof a): the cell volume and the ten sums over the four quadrature points of
// 0.25 grad phi_a . grad phi_b - the loop of the general kernel's symmetric instantiation, verbatim - 660 doubles per mesh.
__global__ void k_box_ref_rows_p2(const int32_t* __restrict__ cells, const double* __restrict__ xyz4, const box_snap bx, double* __restrict__ ref) {
    const int i = threadIdx.x;
    if (i >= 60) return;
    const int ty = i / 10, a = i - 10 * ty;
    const int4 c4 = reinterpret_cast<const int4*>(cells)[ty];
    const int32_t vv[4] = {c4.x, c4.y, c4.z, c4.w};
    const tet_geom t = tet_geometry_box(xyz4, vv, bx);
    const double vol = t.adet * (1.0 / 6.0);
    double row[10];
#pragma unroll
    for (int b = 0; b < 10; ++b) row[b] = 0.0;
#pragma unroll
    for (int qp = 0; qp < 4; ++qp) {
        const double lam[4] = {FS_P2_QP[qp][0], FS_P2_QP[qp][1], FS_P2_QP[qp][2], FS_P2_QP[qp][3]};
        double gp[10][3];
        p2_basis_grads(t, lam, gp);
        double ga[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int b = 0; b < 10; ++b)
            if (b == a) { ga[0] = gp[b][0]; ga[1] = gp[b][1]; ga[2] = gp[b][2]; }
#pragma unroll
        for (int b = 0; b < 10; ++b) row[b] += 0.25 * (ga[0] * gp[b][0] + ga[1] * gp[b][1] + ga[2] * gp[b][2]);
    }
    ref[11 * i] = vol;
#pragma unroll
    for (int b = 0; b < 10; ++b) ref[11 * i + 1 + b] = row[b];
}

template <bool ADD>
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p2_box_gather(
    int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ slice_ptr, const int64_t* __restrict__ inc_slice_ptr, int64_t inc_entries,
    const int32_t* __restrict__ inc_cell, const uint32_t* __restrict__ inc_pos, const double* __restrict__ ref,
    coef_dev kc, coef_dev mc, double* __restrict__ val, const int32_t* __restrict__ order, int acc_doubles) {
    extern __shared__ __attribute__((aligned(16))) double lds_acc[];  // [max_row][blockDim.x], then the 660 reference numbers
    const int tid = threadIdx.x, bd = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, wpb = bd >> 6;
    double* __restrict__ rf = lds_acc + acc_doubles;
    for (int i = tid; i < 660; i += bd) rf[i] = ref[i];
    __syncthreads();
    const int64_t n_chunks = (n_slices + wpb - 1) / wpb;
    for (chunk_iter it = xcd_chunks(n_chunks); it.cur < it.end; it.cur += it.step) {
        const int64_t q0 = it.cur * wpb + wave;
        if (q0 >= n_slices) continue;
        const int64_t s = order ? order[q0] : q0;
        const int64_t base = slice_ptr[s];
        const int width = (int)((slice_ptr[s + 1] - base) >> 6);
        const int64_t ibase = inc_slice_ptr[s];
        const int iwidth = (int)((inc_slice_ptr[s + 1] - ibase) >> 6);
        for (int k = 0; k < width; ++k) lds_acc[k * bd + tid] = 0.0;
        constexpr int PF = 4;
        for (int j0 = 0; j0 < iwidth; j0 += PF) {
            int32_t qc[PF];
            uint32_t pc[PF][3];
            double kk[PF], mv[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int j = j0 + u;
                const int64_t e = ibase + (int64_t)j * FS_SLICE + lane;
                qc[u] = j < iwidth ? inc_cell[e] : -1;
#pragma unroll
                for (int w = 0; w < 3; ++w) pc[u][w] = j < iwidth ? inc_pos[w * inc_entries + e] : 0u;
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int c = qc[u] >= 0 ? qc[u] / 10 : 0;
                kk[u] = kc.mode == FS_COEF_CONST ? kc.value : (kc.mode == FS_COEF_CELL ? kc.data[c] : 0.0);
                mv[u] = mc.mode == FS_COEF_CONST ? mc.value : (mc.mode == FS_COEF_CELL ? mc.data[c] : 0.0);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int32_t q = qc[u];
                if (q < 0) continue;
                const int c = q / 10, a = q - 10 * c;
                const double* __restrict__ R = rf + 11 * (10 * (c % 6) + a);
                const double vol = R[0];
                double row[10];
#pragma unroll
                for (int b = 0; b < 10; ++b) row[b] = 0.0;
                if (kc.mode != FS_COEF_NONE) {
                    const double w = kk[u] * vol;
#pragma unroll
                    for (int b = 0; b < 10; ++b) row[b] = R[1 + b] * w;
                }
                if (mc.mode != FS_COEF_NONE) {
                    const double mm = mv[u] * vol * (1.0 / 420.0);
#pragma unroll
                    for (int b = 0; b < 10; ++b) row[b] += mm * FS_P2_MASS420[a][b];
                }
#pragma unroll
                for (int b = 0; b < 10; ++b) {
                    const int k = (pc[u][b >> 2] >> (8 * (b & 3))) & 255;
                    lds_acc[k * bd + tid] += row[b];
                }
            }
        }
        for (int k = 0; k < width; ++k) {
            const int64_t e = base + (int64_t)k * FS_SLICE + lane;
            const double x = lds_acc[k * bd + tid];
            val[e] = ADD ? val[e] + x : x;
        }
    }
}

// P2 load vector: int f phi_a dx; constant / per-cell f: V * (-1/20 vertex, 1/5 edge); nodal (P2) f: M_e f_e
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p2_source(const int32_t* __restrict__ cell_dofs,
                                                                 const int32_t* __restrict__ cells,
                                                                 const double* __restrict__ xyz4, int64_t nc,
                                                                 int64_t n_rows, coef_dev f, double* __restrict__ b) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < nc; c += stride) {
        int32_t d[10];
        for (int a = 0; a < 10; ++a) d[a] = cell_dofs[c * 10 + a];
        const int4 c4 = reinterpret_cast<const int4*>(cells)[c];
        const int32_t vv[4] = {c4.x, c4.y, c4.z, c4.w};
        const tet_geom t = tet_geometry(xyz4, vv);
        const double vol = t.adet * (1.0 / 6.0);
        if (f.mode == FS_COEF_NODAL) {
            double fe[10];
            for (int a = 0; a < 10; ++a) fe[a] = f.data[d[a]];
            for (int a = 0; a < 10; ++a) {
                if (d[a] >= n_rows) continue;          // rows of other ranks
                double acc = 0.0;
                for (int k = 0; k < 10; ++k) acc += FS_P2_MASS420[a][k] * fe[k];
                atomicAdd(&b[d[a]], acc * vol * (1.0 / 420.0));
            }
        } else {
            const double ff = (f.mode == FS_COEF_CONST ? f.value : f.data[c]) * vol;
            for (int a = 0; a < 4; ++a) if (d[a] < n_rows) atomicAdd(&b[d[a]], -0.05 * ff);
            for (int a = 4; a < 10; ++a) if (d[a] < n_rows) atomicAdd(&b[d[a]], 0.2 * ff);
        }
    }
}

// the same by row gather (lane = row, its cell incidences in ascending order: no atomics, reproducible)
// SUPG load of a NODAL (Function / Expression) source on CG2:  int S_h (v . grad q_a) dx / |K|  for the P2 interpolant S_h = sum_k S_k psi_k
// (ScalarTransportSolver.py:213-226 with Tq of :259-276).  grad q_a is linear in the barycentric coordinates - vertex a: (4 lambda_a - 1) g_a,
// edge ij: 4 (lambda_i g_j + lambda_j g_i) - so all that is needed are the first moments  L_m = int psi_k lambda_m dx / |K|  summed with S_k,
// exact monomial integrals: tetrahedron - vertex k: 0 (m = k), -1/60; edge ij: 1/15 (m in ij), 1/30;  triangle - vertex k: 1/30 (m = k),
// -1/60; edge ij: 2/15 (m in ij), 1/15.  vg[m] = v . g_m.  Local dof order: vertices, then the edges ei / ej of the caller.
template <int NV>
__device__ __forceinline__ double p2_supg_nodal_load(const double* __restrict__ S, const double* __restrict__ vg, int a,
                                                     const int* __restrict__ ei, const int* __restrict__ ej) {
    constexpr int NE = NV == 4 ? 6 : 3;
    constexpr double V_SAME = NV == 4 ? 0.0 : 1.0 / 30.0, V_OTHER = -1.0 / 60.0;
    constexpr double E_IN = NV == 4 ? 1.0 / 15.0 : 2.0 / 15.0, E_OUT = NV == 4 ? 1.0 / 30.0 : 1.0 / 15.0;
    double L[NV], total = 0.0;
#pragma unroll
    for (int m = 0; m < NV; ++m) {
        double l = 0.0;
#pragma unroll
        for (int k = 0; k < NV; ++k) l += (k == m ? V_SAME : V_OTHER) * S[k];
#pragma unroll
        for (int e = 0; e < NE; ++e) l += ((ei[e] == m || ej[e] == m) ? E_IN : E_OUT) * S[NV + e];
        L[m] = l;
        total += l;
    }
    if (a < NV) {
        double la = 0.0, va = 0.0;
#pragma unroll
        for (int m = 0; m < NV; ++m) if (m == a) { la = L[m]; va = vg[m]; }
        return va * (4.0 * la - total);
    }
    double li = 0.0, lj = 0.0, vi = 0.0, vj = 0.0;
#pragma unroll
    for (int m = 0; m < NV; ++m) {
        if (m == ei[a - NV]) { li = L[m]; vi = vg[m]; }
        if (m == ej[a - NV]) { lj = L[m]; vj = vg[m]; }
    }
    return 4.0 * (vj * li + vi * lj);
}

__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p2_source_gather(int64_t n_rows, int64_t n_slices,
                                                                        const int64_t* __restrict__ inc_slice_ptr,
                                                                        const int32_t* __restrict__ inc_cell,
                                                                        const int32_t* __restrict__ cell_dofs,
                                                                        const int32_t* __restrict__ cells,
                                                                        const double* __restrict__ xyz4, coef_dev f,
                                                                        double* __restrict__ b, coef_dev sv = coef_dev(),
                                                                        double supg_pe = 0.0) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t row = s * FS_SLICE + lane;
        const int64_t ibase = inc_slice_ptr[s];
        const int iwidth = (int)((inc_slice_ptr[s + 1] - ibase) >> 6);
        double acc = 0.0;
        for (int j = 0; j < iwidth; ++j) {
            const int32_t q = inc_cell[ibase + (int64_t)j * FS_SLICE + lane];
            if (q < 0) continue;
            const int c = q / 10, a = q - 10 * c;
            const int4 c4 = reinterpret_cast<const int4*>(cells)[c];
            const int32_t vv[4] = {c4.x, c4.y, c4.z, c4.w};
            const tet_geom t = tet_geometry(xyz4, vv);
            const double vol = t.adet * (1.0 / 6.0);
            if (f.mode == FS_COEF_NODAL) {
                double m = 0.0, S[10];
                for (int k = 0; k < 10; ++k) { S[k] = f.data[cell_dofs[(int64_t)c * 10 + k]]; m += FS_P2_MASS420[a][k] * S[k]; }
                acc += m * vol * (1.0 / 420.0);
                if (supg_pe > 0.0 && sv.mode != FS_COEF_NONE) {      // + int S_h tau (v . grad q_a) dx, exactly
                    double vx, vy, vz;
                    if (sv.mode == FS_COEF_CONST) { vx = sv.tensor[0]; vy = sv.tensor[1]; vz = sv.tensor[2]; }
                    else { vx = sv.data[3 * (int64_t)c]; vy = sv.data[3 * (int64_t)c + 1]; vz = sv.data[3 * (int64_t)c + 2]; }
                    const double tau = supg_tau(xyz4, vv, t.adet, sqrt(vx * vx + vy * vy + vz * vz), supg_pe);
                    const int ei[6] = {2, 1, 1, 0, 0, 0}, ej[6] = {3, 3, 2, 3, 2, 1};
                    double vg[4];
                    for (int k = 0; k < 4; ++k) vg[k] = vx * t.g[k][0] + vy * t.g[k][1] + vz * t.g[k][2];
                    acc += vol * tau * p2_supg_nodal_load<4>(S, vg, a, ei, ej);
                }
            } else {
                const double ff = (f.mode == FS_COEF_CONST ? f.value : f.data[c]) * vol;
                acc += (a < 4 ? -0.05 : 0.2) * ff;
                if (supg_pe > 0.0 && sv.mode != FS_COEF_NONE && a >= 4) {
                    // + int S tau (v . grad q_a) dx: the mean gradient of a vertex function vanishes, of the edge function ij it is g_i + g_j
                    double vx, vy, vz;
                    if (sv.mode == FS_COEF_CONST) { vx = sv.tensor[0]; vy = sv.tensor[1]; vz = sv.tensor[2]; }
                    else { vx = sv.data[3 * (int64_t)c]; vy = sv.data[3 * (int64_t)c + 1]; vz = sv.data[3 * (int64_t)c + 2]; }
                    const double tau = supg_tau(xyz4, vv, t.adet, sqrt(vx * vx + vy * vy + vz * vz), supg_pe);
                    const int ei[6] = {2, 1, 1, 0, 0, 0}, ej[6] = {3, 3, 2, 3, 2, 1};
                    const int i = ei[a - 4], jj = ej[a - 4];
                    acc += ff * tau * ((t.g[i][0] + t.g[jj][0]) * vx + (t.g[i][1] + t.g[jj][1]) * vy + (t.g[i][2] + t.g[jj][2]) * vz);
                }
            }
        }
        if (row < n_rows) b[row] += acc;
    }
}

// P2 boundary load: int g phi_a ds over a facet = g * area / 3 on each of its 3 edge nodes, 0 on the vertices
__global__ void k_facet_vector_p2(const double* __restrict__ xyz4, const int32_t* __restrict__ tri, int64_t nf,
                                  const double* __restrict__ g, const uint64_t* __restrict__ edge_keys, int64_t ne,
                                  int grouped, const int32_t* __restrict__ edge_node, int64_t n_rows, int ncomp,
                                  double* __restrict__ b, int* __restrict__ err) {
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; f < nf; f += stride) {
        const int32_t v[3] = {tri[3 * f], tri[3 * f + 1], tri[3 * f + 2]};
        const double w = tri_area(xyz4, v[0], v[1], v[2]) * (1.0 / 3.0);      // g: [nf][ncomp]
        const int pi[3] = {0, 0, 1}, pj[3] = {1, 2, 2};
        for (int e = 0; e < 3; ++e) {
            const int32_t a = v[pi[e]], bb = v[pj[e]];
            const uint32_t lo_v = (uint32_t)(a < bb ? a : bb), hi_v = (uint32_t)(a < bb ? bb : a);
            const uint64_t key = grouped ? (((uint64_t)(hi_v - lo_v) << 32) | lo_v) : (((uint64_t)lo_v << 32) | hi_v);
            int64_t lo = 0, hi = ne;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (edge_keys[mid] < key) lo = mid + 1; else hi = mid;
            }
            if (lo < ne && edge_keys[lo] == key) {
                if (edge_node[lo] < n_rows)                                      // edge rows of other ranks: theirs to add
                    for (int i = 0; i < ncomp; ++i) atomicAdd(&b[(int64_t)edge_node[lo] * ncomp + i], w * g[f * ncomp + i]);
            } else {
                atomicAdd(err, 1);
            }
        }
    }
}

// CG2 Robin / HTC matrix  int_F h phi_a phi_b ds: the exact P2 mass matrix of the facet triangle (nodes: its three
// vertices, then the edge nodes (0,1), (0,2), (1,2)), A/180 * [[6 -1 -1 0 0 -4], ...] - a vertex couples with -4 to the
// opposite edge node, with 0 to the adjacent ones; edge nodes 32 / 16.  Thread per (facet, row node).
__device__ const double FS_P2_TRI_MASS180[6][6] = {{6, -1, -1, 0, 0, -4}, {-1, 6, -1, 0, -4, 0}, {-1, -1, 6, -4, 0, 0},
                                                  {0, 0, -4, 32, 16, 16}, {0, -4, 0, 16, 32, 16}, {-4, 0, 0, 16, 16, 32}};
__global__ void k_facet_matrix_p2(const double* __restrict__ xyz4, const int32_t* __restrict__ tri, int64_t nf,
                                  const double* __restrict__ h, const uint64_t* __restrict__ edge_keys, int64_t ne,
                                  int grouped, const int32_t* __restrict__ edge_node, int64_t n_rows, int64_t nvo, int64_t neo,
                                  const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ sell_col,
                                  double* __restrict__ val, int* __restrict__ err) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nf * 6; t += stride) {
        const int64_t f = t / 6;
        const int a = (int)(t - f * 6);
        const int32_t v[3] = {tri[3 * f], tri[3 * f + 1], tri[3 * f + 2]};
        // node of a vertex: itself when owned, shifted behind the owned edge nodes when ghost
        int32_t node[6] = {(int32_t)(v[0] < nvo ? v[0] : v[0] + neo), (int32_t)(v[1] < nvo ? v[1] : v[1] + neo),
                           (int32_t)(v[2] < nvo ? v[2] : v[2] + neo), -1, -1, -1};
        const int pi[3] = {0, 0, 1}, pj[3] = {1, 2, 2};
        bool ok = true;
        for (int e = 0; e < 3; ++e) {
            const int32_t p = v[pi[e]], q = v[pj[e]];
            const uint32_t lo_v = (uint32_t)(p < q ? p : q), hi_v = (uint32_t)(p < q ? q : p);
            const uint64_t key = grouped ? (((uint64_t)(hi_v - lo_v) << 32) | lo_v) : (((uint64_t)lo_v << 32) | hi_v);
            int64_t lo = 0, hi = ne;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (edge_keys[mid] < key) lo = mid + 1; else hi = mid;
            }
            if (lo < ne && edge_keys[lo] == key) node[3 + e] = edge_node[lo];
            else ok = false;
        }
        if (!ok) { if (a == 0) atomicAdd(err, 1); continue; }
        const int32_t row = node[a];
        if (row >= n_rows) continue;
        const double w = h[f] * tri_area(xyz4, v[0], v[1], v[2]) * (1.0 / 180.0);
        const int64_t sp0 = slice_ptr[row >> 6];
        const int width = (int)((slice_ptr[(row >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (row & 63);
        for (int b = 0; b < 6; ++b) {
            const double m = FS_P2_TRI_MASS180[a][b];
            if (m == 0.0) continue;
            int k = -1;
            for (int kk = 0; kk < width; ++kk)
                if (sell_col[base + (int64_t)kk * FS_SLICE] == node[b]) { k = kk; break; }
            if (k >= 0) atomicAdd(&val[base + (int64_t)k * FS_SLICE], w * m);
            else atomicAdd(err, 1);
        }
    }
}

// ---- vector P1 elasticity: 3x3 block per node pair, plane (i*3+j) of the SELL value array ------
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p1_elasticity(const int32_t* __restrict__ cells,
                                                                     const double* __restrict__ xyz4,
                                                                     const int32_t* __restrict__ slots, int64_t nc,
                                                                     double mu, double lambda, coef_dev mc,
                                                                     int64_t plane, double* __restrict__ val) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < nc; c += stride) {
        const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
        const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
        const tet_geom t = tet_geometry(xyz4, v);
        const double vol = t.adet * (1.0 / 6.0);
        double mm = 0.0;
        if (mc.mode != FS_COEF_NONE) mm = (mc.mode == FS_COEF_CONST ? mc.value : mc.data[c]) * t.adet * (1.0 / 120.0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int32_t slot = slots[(int64_t)(a * 4 + b) * nc + c];
                if (slot < 0) continue;
                const double gg = t.g[a][0] * t.g[b][0] + t.g[a][1] * t.g[b][1] + t.g[a][2] * t.g[b][2];
                const double ms = (a == b ? 2.0 : 1.0) * mm;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        double x = vol * (lambda * t.g[a][i] * t.g[b][j] + mu * t.g[a][j] * t.g[b][i]);
                        if (i == j) x += vol * mu * gg + ms;
                        atomicAdd(&val[(int64_t)(i * 3 + j) * plane + slot], x);
                    }
            }
    }
}

// The same operator without atomics: one thread per STORED 3x3 block sums the contributions of the cells that hold both
// nodes (inverse slot table, ascending cell order: bit-reproducible), recomputing the four barycentric gradients of
// each cell - 60 flops against an 8-byte atomic per value.  configs[2] (9.86 M tets, 25 M blocks): 38.8 ms with
// 1.4 G device-scope fp64 atomics.
template <bool ADD>
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p1_elasticity_gather(int64_t n_entries, const int32_t* __restrict__ ptr,
                                                                            const int32_t* __restrict__ src,
                                                                            const int32_t* __restrict__ cells,
                                                                            const double* __restrict__ xyz4, double mu, double lambda,
                                                                            coef_dev mc, int64_t plane, double* __restrict__ val, const box_snap bx) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < n_entries; e += stride) {
        double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        const int32_t q1 = ptr[e + 1];
        // sources in groups of four: their indices, then their cell records, go out before the first coordinate load
        constexpr int PF = 4;
        for (int32_t q0 = ptr[e]; q0 < q1; q0 += PF) {
          int32_t sc[PF];
          int4 vc[PF];
#pragma unroll
          for (int u = 0; u < PF; ++u) sc[u] = q0 + u < q1 ? src[q0 + u] : -1;
#pragma unroll
          for (int u = 0; u < PF; ++u) vc[u] = sc[u] >= 0 ? reinterpret_cast<const int4*>(cells)[sc[u] >> 4] : make_int4(0, 0, 0, 0);
#pragma unroll
          for (int u = 0; u < PF; ++u) {
            if (q0 + u >= q1) break;
            const int32_t sidx = sc[u];
            const int64_t c = sidx >> 4;
            const int a = (sidx >> 2) & 3, b = sidx & 3;
            const int4 v4 = vc[u];
            const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
            // (box meshes: edge vectors snapped to the grid spacing, as the scalar kernels do - equal stencils become equal block rows
            // bit for bit, which the row-dictionary product of the AMG fine level lives on)
            const tet_geom t = tet_geometry_box(xyz4, v, bx);
            const double vol = t.adet * (1.0 / 6.0);
            double ga[3], gb[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ga[k] = a == 0 ? t.g[0][k] : a == 1 ? t.g[1][k] : a == 2 ? t.g[2][k] : t.g[3][k];
                gb[k] = b == 0 ? t.g[0][k] : b == 1 ? t.g[1][k] : b == 2 ? t.g[2][k] : t.g[3][k];
            }
            double ms = 0.0;
            if (mc.mode != FS_COEF_NONE)
                ms = (a == b ? 2.0 : 1.0) * (mc.mode == FS_COEF_CONST ? mc.value : mc.data[c]) * t.adet * (1.0 / 120.0);
            const double gg = ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double x = vol * (lambda * ga[i] * gb[j] + mu * ga[j] * gb[i]);
                    if (i == j) x += vol * mu * gg + ms;
                    acc[i][j] += x;
                }
          }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int64_t idx = (int64_t)(i * 3 + j) * plane + e;
                val[idx] = ADD ? val[idx] + acc[i][j] : acc[i][j];
            }
    }
}

// ---- vector P2 elasticity (the reference's own example: VectorFunctionSpace(mesh, 'CG', 2),
// examples/test_linear_elasticity.py:105-106; form LinearElasticitySolver.py:62-69, 215) --------------------------------
// Same scheme as the P1 operator: one thread per STORED 3x3 block sums, in ascending order, the (cell, a, b) sources the
// inverse slot table lists for it (source index = cell*100 + a*10 + b), recomputing the cell's barycentric gradients.
// K_ab[i][j] = int lambda d_i phi_a d_j phi_b + mu d_j phi_a d_i phi_b + mu delta_ij grad phi_a . grad phi_b dx, quadratic
// integrand, 4-point rule (exact; the rule FFC picks).  Local nodes: 4 vertices, then the 6 UFC edges.
__device__ __constant__ int FS_P2_EI[6] = {2, 1, 1, 0, 0, 0};
__device__ __constant__ int FS_P2_EJ[6] = {3, 3, 2, 3, 2, 1};
__device__ __forceinline__ double tet_g(const tet_geom& t, int a, int d) {      // t.g[a][d] without dynamic register indexing
    return a == 0 ? t.g[0][d] : a == 1 ? t.g[1][d] : a == 2 ? t.g[2][d] : t.g[3][d];
}
__device__ __forceinline__ void p2_grad_one(const tet_geom& t, int qp, int a, double (&ga)[3]) {
    if (a < 4) {
        const double w = 4.0 * FS_P2_QP[qp][a] - 1.0;
#pragma unroll
        for (int d = 0; d < 3; ++d) ga[d] = w * tet_g(t, a, d);
    } else {
        const int i = FS_P2_EI[a - 4], j = FS_P2_EJ[a - 4];
        const double li = 4.0 * FS_P2_QP[qp][i], lj = 4.0 * FS_P2_QP[qp][j];
#pragma unroll
        for (int d = 0; d < 3; ++d) ga[d] = li * tet_g(t, j, d) + lj * tet_g(t, i, d);
    }
}
template <bool ADD>
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p2_elasticity_gather(int64_t n_entries, const int32_t* __restrict__ ptr,
                                                                            const int32_t* __restrict__ src,
                                                                            const int32_t* __restrict__ cells,
                                                                            const double* __restrict__ xyz4, double mu, double lambda,
                                                                            coef_dev mc, int64_t plane, double* __restrict__ val) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < n_entries; e += stride) {
        double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        const int32_t q1 = ptr[e + 1];
        for (int32_t q = ptr[e]; q < q1; ++q) {
            const int32_t sidx = src[q];
            const int64_t c = sidx / 100;
            const int ab = sidx - (int32_t)c * 100, a = ab / 10, b = ab - 10 * a;
            const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
            const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
            const tet_geom t = tet_geometry(xyz4, v);
            const double w = t.adet * (1.0 / 24.0);            // volume * quadrature weight 1/4
#pragma unroll
            for (int qp = 0; qp < 4; ++qp) {
                double ga[3], gb[3];
                p2_grad_one(t, qp, a, ga);
                p2_grad_one(t, qp, b, gb);
                const double gg = mu * (ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2]);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        double x = lambda * ga[i] * gb[j] + mu * ga[j] * gb[i];
                        if (i == j) x += gg;
                        acc[i][j] += w * x;
                    }
            }
            if (mc.mode != FS_COEF_NONE) {
                const double ms = (mc.mode == FS_COEF_CONST ? mc.value : mc.data[c]) * t.adet * (1.0 / 2520.0) * FS_P2_MASS420[a][b];
                acc[0][0] += ms; acc[1][1] += ms; acc[2][2] += ms;
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int64_t idx = (int64_t)(i * 3 + j) * plane + e;
                val[idx] = ADD ? val[idx] + acc[i][j] : acc[i][j];
            }
    }
}

// Load vector of the vector P2 space: body force  int f . phi_a dx  (constant f: -V/20 on vertex nodes, V/5 on edge nodes)
// and the thermal-stress load  int c div v dx = int c d_i phi_a dx  with c constant, per cell, or P1 through its VERTEX
// values (nodal array over the space's nodes; exact for a P1 temperature):
//   vertex a:      V g_a (c_a / 5 - S / 20)                          S = sum of the four vertex values
//   edge (i, j):   V / 5 ((S + c_i) g_j + (S + c_j) g_i)             (constant c: 0 and c V (g_i + g_j))
// One thread per owned node; its (cell, a) incidences are the sources of its diagonal block, ascending (reproducible).
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p2_vector_source_gather(int64_t n_rows, const int64_t* __restrict__ slice_ptr,
                                                                               const int32_t* __restrict__ sell_col,
                                                                               const int32_t* __restrict__ gptr,
                                                                               const int32_t* __restrict__ gsrc,
                                                                               const int32_t* __restrict__ cells,
                                                                               const double* __restrict__ xyz4, double fx, double fy,
                                                                               double fz, coef_dev dv, int64_t nvo, int64_t neo,
                                                                               double* __restrict__ b) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_rows; r += stride) {
        const int64_t sp0 = slice_ptr[r >> 6];
        const int width = (int)((slice_ptr[(r >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (r & 63);
        int64_t e = -1;
        for (int k = 0; k < width; ++k)
            if (sell_col[base + (int64_t)k * FS_SLICE] == (int32_t)r) { e = base + (int64_t)k * FS_SLICE; break; }
        double acc[3] = {0.0, 0.0, 0.0};
        if (e >= 0) {
            for (int32_t q = gptr[e]; q < gptr[e + 1]; ++q) {
                const int32_t sidx = gsrc[q];
                const int64_t c = sidx / 100;
                const int a = (sidx - (int32_t)c * 100) / 10;
                const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
                const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
                const tet_geom t = tet_geometry(xyz4, v);
                const double vol = t.adet * (1.0 / 6.0);
                const double wf = vol * (a < 4 ? -0.05 : 0.2);
                acc[0] += wf * fx; acc[1] += wf * fy; acc[2] += wf * fz;
                if (dv.mode == FS_COEF_NONE) continue;
                double cv[4];
                if (dv.mode == FS_COEF_NODAL) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) cv[k] = dv.data[v[k] < nvo ? v[k] : v[k] + neo];
                } else {
                    const double cc = dv.mode == FS_COEF_CONST ? dv.value : dv.data[c];
                    cv[0] = cv[1] = cv[2] = cv[3] = cc;
                }
                const double S = (cv[0] + cv[1]) + (cv[2] + cv[3]);
                if (a < 4) {
                    const double ca = a == 0 ? cv[0] : a == 1 ? cv[1] : a == 2 ? cv[2] : cv[3];
                    const double w = vol * (0.2 * ca - 0.05 * S);
#pragma unroll
                    for (int d = 0; d < 3; ++d) acc[d] += w * tet_g(t, a, d);
                } else {
                    const int i = FS_P2_EI[a - 4], j = FS_P2_EJ[a - 4];
                    const double ci = i == 0 ? cv[0] : i == 1 ? cv[1] : i == 2 ? cv[2] : cv[3];
                    const double cj = j == 0 ? cv[0] : j == 1 ? cv[1] : j == 2 ? cv[2] : cv[3];
                    const double wi = 0.2 * vol * (S + ci), wj = 0.2 * vol * (S + cj);
#pragma unroll
                    for (int d = 0; d < 3; ++d) acc[d] += wi * tet_g(t, j, d) + wj * tet_g(t, i, d);
                }
            }
        }
        b[3 * r + 0] += acc[0];
        b[3 * r + 1] += acc[1];
        b[3 * r + 2] += acc[2];
    }
}

// ---- 2-D: CG1 on triangles ------------------------------------------------------------------------
// (the reference's runnable examples are 2-D: examples/test_heat_transfer.py:34, test_electrostatics.py:35)
struct tri_geom {
    double g[3][2];   // gradients of the barycentric basis
    double area;
};
__device__ __forceinline__ tri_geom tri_geometry2(const double* __restrict__ xyz4, int32_t a, int32_t b, int32_t c) {
    const double2 p0 = reinterpret_cast<const double2*>(xyz4)[2 * (int64_t)a];
    const double2 p1 = reinterpret_cast<const double2*>(xyz4)[2 * (int64_t)b];
    const double2 p2 = reinterpret_cast<const double2*>(xyz4)[2 * (int64_t)c];
    const double e1x = p1.x - p0.x, e1y = p1.y - p0.y, e2x = p2.x - p0.x, e2y = p2.y - p0.y;
    const double det = e1x * e2y - e1y * e2x;
    const double inv = 1.0 / det;
    tri_geom t;
    t.g[1][0] = e2y * inv;  t.g[1][1] = -e2x * inv;
    t.g[2][0] = -e1y * inv; t.g[2][1] = e1x * inv;
    t.g[0][0] = -(t.g[1][0] + t.g[2][0]);
    t.g[0][1] = -(t.g[1][1] + t.g[2][1]);
    t.area = 0.5 * fabs(det);
    return t;
}

// SUPG parameter on a triangle: tau = 0.5 h / (4/(Pe h) + 2 |v|), h = 2 Circumradius = a b c / (2 A)
__device__ __forceinline__ double supg_tau_tri(const double* __restrict__ xyz4, int32_t v0, int32_t v1, int32_t v2, double area,
                                               double vnorm, double pe) {
    const double x0 = xyz4[4 * (int64_t)v0], y0 = xyz4[4 * (int64_t)v0 + 1], x1 = xyz4[4 * (int64_t)v1], y1 = xyz4[4 * (int64_t)v1 + 1];
    const double x2 = xyz4[4 * (int64_t)v2], y2 = xyz4[4 * (int64_t)v2 + 1];
    const double a = sqrt((x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0)), b = sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1));
    const double c = sqrt((x0 - x2) * (x0 - x2) + (y0 - y2) * (y0 - y2));
    const double h = a * b * c / (2.0 * area);
    return 0.5 * h / (4.0 / (pe * h) + 2.0 * vnorm);
}

// ---- plane-strain elasticity on triangles (LinearElasticitySolver.py:62-69 with dimension 2; the reference sends 2D
// problems to solve_linear_problem, :247-253) ------------------------------------------------------------------------------
// One thread per stored 2x2 block sums, in ascending order, the (cell, a, b) sources of the inverse slot table
// (source index = cell*9 + a*3 + b): K_ab[i][j] = A (lambda d_i phi_a d_j phi_b + mu d_j phi_a d_i phi_b + mu delta_ij
// grad phi_a . grad phi_b) [+ mass (1 + delta_ab) A / 12 delta_ij].
template <bool ADD>
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_tri_elasticity_gather(int64_t n_entries, const int32_t* __restrict__ ptr,
                                                                             const int32_t* __restrict__ src,
                                                                             const int32_t* __restrict__ cells,
                                                                             const double* __restrict__ xyz4, double mu, double lambda,
                                                                             coef_dev mc, int64_t plane, double* __restrict__ val) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < n_entries; e += stride) {
        double acc[2][2] = {{0, 0}, {0, 0}};
        const int32_t q1 = ptr[e + 1];
        for (int32_t q = ptr[e]; q < q1; ++q) {
            const int32_t sidx = src[q];
            const int64_t c = sidx / 9;
            const int ab = sidx - (int32_t)(c * 9);
            const int a = ab / 3, b = ab - 3 * a;
            const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
            const tri_geom t = tri_geometry2(xyz4, v4.x, v4.y, v4.z);
            const double ga[2] = {a == 0 ? t.g[0][0] : (a == 1 ? t.g[1][0] : t.g[2][0]), a == 0 ? t.g[0][1] : (a == 1 ? t.g[1][1] : t.g[2][1])};
            const double gb[2] = {b == 0 ? t.g[0][0] : (b == 1 ? t.g[1][0] : t.g[2][0]), b == 0 ? t.g[0][1] : (b == 1 ? t.g[1][1] : t.g[2][1])};
            double ms = 0.0;
            if (mc.mode != FS_COEF_NONE)
                ms = (a == b ? 2.0 : 1.0) * (mc.mode == FS_COEF_CONST ? mc.value : mc.data[c]) * t.area * (1.0 / 12.0);
            const double gg = ga[0] * gb[0] + ga[1] * gb[1];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    double x = t.area * (lambda * ga[i] * gb[j] + mu * ga[j] * gb[i]);
                    if (i == j) x += t.area * mu * gg + ms;
                    acc[i][j] += x;
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t idx = (int64_t)(i * 2 + j) * plane + e;
                val[idx] = ADD ? val[idx] + acc[i][j] : acc[i][j];
            }
    }
}

// b_(a,i) += int f_i phi_a dx + int c d_i phi_a dx over the triangles of vertex a, listed by the diagonal block of the
// inverse slot table in ascending cell order
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_tri_vector_source_gather(int64_t n_rows, const int64_t* __restrict__ slice_ptr,
                                                                                const int32_t* __restrict__ sell_col,
                                                                                const int32_t* __restrict__ gptr,
                                                                                const int32_t* __restrict__ gsrc,
                                                                                const int32_t* __restrict__ cells,
                                                                                const double* __restrict__ xyz4, double fx, double fy,
                                                                                coef_dev dv, double* __restrict__ b) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_rows; r += stride) {
        const int64_t sp0 = slice_ptr[r >> 6];
        const int width = (int)((slice_ptr[(r >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (r & 63);
        int64_t e = -1;
        for (int k = 0; k < width; ++k)
            if (sell_col[base + (int64_t)k * FS_SLICE] == (int32_t)r) { e = base + (int64_t)k * FS_SLICE; break; }
        double acc[2] = {0.0, 0.0};
        if (e >= 0) {
            for (int32_t q = gptr[e]; q < gptr[e + 1]; ++q) {
                const int32_t sidx = gsrc[q];
                const int64_t c = sidx / 9;
                const int a = (sidx - (int32_t)(c * 9)) / 3;
                const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
                const tri_geom t = tri_geometry2(xyz4, v4.x, v4.y, v4.z);
                const double w = t.area * (1.0 / 3.0);
                double cd = 0.0;
                if (dv.mode == FS_COEF_CONST) cd = dv.value;
                else if (dv.mode == FS_COEF_CELL) cd = dv.data[c];
                else if (dv.mode == FS_COEF_NODAL) cd = ((dv.data[v4.x] + dv.data[v4.y]) + dv.data[v4.z]) * (1.0 / 3.0);
                cd *= t.area;
                acc[0] += w * fx + cd * (a == 0 ? t.g[0][0] : (a == 1 ? t.g[1][0] : t.g[2][0]));
                acc[1] += w * fy + cd * (a == 0 ? t.g[0][1] : (a == 1 ? t.g[1][1] : t.g[2][1]));
            }
        }
        b[2 * r + 0] += acc[0];
        b[2 * r + 1] += acc[1];
    }
}

// traction on boundary edges of a 2-vector space: b_(v,i) += g_i |edge| / 2 for both end points
__global__ void k_edge_vector2(const double* __restrict__ xyz4, const int32_t* __restrict__ ed, int64_t nf,
                               const double* __restrict__ g, int64_t n_rows, double* __restrict__ b) {
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; f < nf; f += stride) {
        const int32_t a = ed[2 * f], c = ed[2 * f + 1];
        const double dx = xyz4[4 * (int64_t)c] - xyz4[4 * (int64_t)a], dy = xyz4[4 * (int64_t)c + 1] - xyz4[4 * (int64_t)a + 1];
        const double w = 0.5 * sqrt(dx * dx + dy * dy);
        for (int i = 0; i < 2; ++i) {
            if (a < n_rows) atomicAdd(&b[2 * (int64_t)a + i], w * g[2 * f + i]);
            if (c < n_rows) atomicAdd(&b[2 * (int64_t)c + i], w * g[2 * f + i]);
        }
    }
}

// row-gather assembly, the triangle counterpart of k_assemble_p1_scalar_gather (lane = row, LDS row accumulator)
template <bool ADD>
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_tri_scalar_gather(
    int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ slice_ptr,
    const int64_t* __restrict__ inc_slice_ptr, const int32_t* __restrict__ inc_cell,
    const uint32_t* __restrict__ inc_pos, const int32_t* __restrict__ cells, const double* __restrict__ xyz4,
    coef_dev kc, coef_dev mc, coef_dev ac, double ascale, double* __restrict__ val, double supg_pe = 0.0) {
    extern __shared__ __attribute__((aligned(16))) double lds_acc[];
    const int tid = threadIdx.x, bd = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, wpb = bd >> 6;
    const int64_t n_chunks = (n_slices + wpb - 1) / wpb;
    for (chunk_iter it = xcd_chunks(n_chunks); it.cur < it.end; it.cur += it.step) {
        const int64_t s = it.cur * wpb + wave;
        if (s >= n_slices) continue;
        const int64_t base = slice_ptr[s];
        const int width = (int)((slice_ptr[s + 1] - base) >> 6);
        const int64_t ibase = inc_slice_ptr[s];
        const int iwidth = (int)((inc_slice_ptr[s + 1] - ibase) >> 6);
        for (int k = 0; k < width; ++k) lds_acc[k * bd + tid] = 0.0;
        for (int j = 0; j < iwidth; ++j) {
            const int32_t q = inc_cell[ibase + (int64_t)j * FS_SLICE + lane];
            if (q < 0) continue;
            const uint32_t packed = inc_pos[ibase + (int64_t)j * FS_SLICE + lane];
            const int c = q / 3, a = q - 3 * c;
            const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
            const tri_geom t = tri_geometry2(xyz4, v4.x, v4.y, v4.z);
            double ga[2];
            ga[0] = a == 0 ? t.g[0][0] : (a == 1 ? t.g[1][0] : t.g[2][0]);
            ga[1] = a == 0 ? t.g[0][1] : (a == 1 ? t.g[1][1] : t.g[2][1]);
            double row[3];
            if (kc.mode == FS_COEF_TENSOR) {       // 2x2 tensor in the leading block of the 3x3 storage
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const double kx = kc.tensor[0] * t.g[b][0] + kc.tensor[1] * t.g[b][1];
                    const double ky = kc.tensor[3] * t.g[b][0] + kc.tensor[4] * t.g[b][1];
                    row[b] = t.area * (ga[0] * kx + ga[1] * ky);
                }
            } else if (kc.mode == FS_COEF_CELL_TENSOR) {
                const double* __restrict__ T = kc.data + 9 * (int64_t)c;
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const double kx = T[0] * t.g[b][0] + T[1] * t.g[b][1];
                    const double ky = T[3] * t.g[b][0] + T[4] * t.g[b][1];
                    row[b] = t.area * (ga[0] * kx + ga[1] * ky);
                }
            } else {
                double kk = 0.0;
                if (kc.mode == FS_COEF_CONST) kk = kc.value;
                else if (kc.mode == FS_COEF_CELL) kk = kc.data[c];
#pragma unroll
                for (int b = 0; b < 3; ++b) row[b] = kk * t.area * (ga[0] * t.g[b][0] + ga[1] * t.g[b][1]);
            }
            if (mc.mode != FS_COEF_NONE) {
                const double mm = (mc.mode == FS_COEF_CONST ? mc.value : mc.data[c]) * t.area * (1.0 / 12.0);
#pragma unroll
                for (int b = 0; b < 3; ++b) row[b] += (b == a ? 2.0 : 1.0) * mm;
            }
            if (ac.mode != FS_COEF_NONE) {         // Galerkin advection: scale * (area/3) * (v . grad phi_b)
                double vx, vy;
                if (ac.mode == FS_COEF_CONST) { vx = ac.tensor[0]; vy = ac.tensor[1]; }
                else {
                    const int64_t o = ac.mode == FS_COEF_CELL_ROW ? 3 * (3 * (int64_t)c + a) : 3 * (int64_t)c;
                    vx = ac.data[o]; vy = ac.data[o + 1];
                }
                const double w3 = ascale * t.area * (1.0 / 3.0);
#pragma unroll
                for (int b = 0; b < 3; ++b) row[b] += w3 * (vx * t.g[b][0] + vy * t.g[b][1]);
                if (supg_pe > 0.0) {      // test function q + tau (v . grad q) on the advection and mass terms (:259-270)
                    const double tau = supg_tau_tri(xyz4, v4.x, v4.y, v4.z, t.area, sqrt(vx * vx + vy * vy), supg_pe);
                    const double wa = tau * (vx * ga[0] + vy * ga[1]);
                    const double mval = mc.mode == FS_COEF_NONE ? 0.0 : (mc.mode == FS_COEF_CONST ? mc.value : mc.data[c]);
#pragma unroll
                    for (int b = 0; b < 3; ++b) row[b] += wa * t.area * (ascale * (vx * t.g[b][0] + vy * t.g[b][1]) + mval * (1.0 / 3.0));
                }
            }
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int k = (packed >> (8 * b)) & 255;
                lds_acc[k * bd + tid] += row[b];
            }
        }
        for (int k = 0; k < width; ++k) {
            const int64_t e = base + (int64_t)k * FS_SLICE + lane;
            const double x = lds_acc[k * bd + tid];
            val[e] = ADD ? val[e] + x : x;
        }
    }
}

__global__ void __launch_bounds__(FS_BLOCK) k_assemble_tri_source(const int32_t* __restrict__ cells, const double* __restrict__ xyz4,
                                                                  int64_t nc, int64_t n_rows, coef_dev f, double* __restrict__ b,
                                                                  coef_dev sv, double supg_pe) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < nc; c += stride) {
        const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
        const int32_t v[3] = {v4.x, v4.y, v4.z};
        const tri_geom t = tri_geometry2(xyz4, v[0], v[1], v[2]);
        double be[3];
        if (f.mode == FS_COEF_NODAL) {
            const double fe[3] = {f.data[v[0]], f.data[v[1]], f.data[v[2]]};
            const double sum = fe[0] + fe[1] + fe[2];
            for (int a = 0; a < 3; ++a) be[a] = t.area * (1.0 / 12.0) * (sum + fe[a]);
            if (supg_pe > 0.0 && sv.mode != FS_COEF_NONE) {     // + int S_h tau (v . grad phi_a) dx = tau (v . g_a) |K| mean(S): exact
                double vx, vy;
                if (sv.mode == FS_COEF_CONST) { vx = sv.tensor[0]; vy = sv.tensor[1]; }
                else { vx = sv.data[3 * (int64_t)c]; vy = sv.data[3 * (int64_t)c + 1]; }
                const double tau = supg_tau_tri(xyz4, v[0], v[1], v[2], t.area, sqrt(vx * vx + vy * vy), supg_pe);
                for (int a = 0; a < 3; ++a) be[a] += sum * (1.0 / 3.0) * t.area * tau * (vx * t.g[a][0] + vy * t.g[a][1]);
            }
        } else {
            const double ff = f.mode == FS_COEF_CONST ? f.value : f.data[c];
            for (int a = 0; a < 3; ++a) be[a] = ff * t.area * (1.0 / 3.0);
            if (supg_pe > 0.0 && sv.mode != FS_COEF_NONE) {     // + int S tau (v . grad phi_a) dx
                double vx, vy;
                if (sv.mode == FS_COEF_CONST) { vx = sv.tensor[0]; vy = sv.tensor[1]; }
                else { vx = sv.data[3 * (int64_t)c]; vy = sv.data[3 * (int64_t)c + 1]; }
                const double tau = supg_tau_tri(xyz4, v[0], v[1], v[2], t.area, sqrt(vx * vx + vy * vy), supg_pe);
                for (int a = 0; a < 3; ++a) be[a] += ff * t.area * tau * (vx * t.g[a][0] + vy * t.g[a][1]);
            }
        }
        for (int a = 0; a < 3; ++a)
            if (v[a] < n_rows) atomicAdd(&b[v[a]], be[a]);
    }
}

// boundary edges: b_a += g len/2 ; A += h len/6 [[2,1],[1,2]]
__global__ void k_edge_vector(const double* __restrict__ xyz4, const int32_t* __restrict__ ed, int64_t nf,
                              const double* __restrict__ g, int64_t n_rows, double* __restrict__ b) {
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; f < nf; f += stride) {
        const int32_t a = ed[2 * f], c = ed[2 * f + 1];
        const double dx = xyz4[4 * (int64_t)c] - xyz4[4 * (int64_t)a], dy = xyz4[4 * (int64_t)c + 1] - xyz4[4 * (int64_t)a + 1];
        const double w = 0.5 * sqrt(dx * dx + dy * dy) * g[f];
        if (a < n_rows) atomicAdd(&b[a], w);
        if (c < n_rows) atomicAdd(&b[c], w);
    }
}
__global__ void k_edge_matrix(const double* __restrict__ xyz4, const int32_t* __restrict__ ed, int64_t nf,
                              const double* __restrict__ h, int64_t n_rows, const int64_t* __restrict__ slice_ptr,
                              const int32_t* __restrict__ sell_col, double* __restrict__ val, int* __restrict__ err) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nf * 4; t += stride) {
        const int64_t f = t >> 2;
        const int i = (int)(t & 3) >> 1, j = (int)(t & 1);
        const int32_t a = ed[2 * f], c = ed[2 * f + 1];
        const int32_t row = i ? c : a, col = j ? c : a;
        if (row >= n_rows) continue;
        const double dx = xyz4[4 * (int64_t)c] - xyz4[4 * (int64_t)a], dy = xyz4[4 * (int64_t)c + 1] - xyz4[4 * (int64_t)a + 1];
        const double w = h[f] * sqrt(dx * dx + dy * dy) * (1.0 / 6.0) * (i == j ? 2.0 : 1.0);
        const int64_t sp0 = slice_ptr[row >> 6];
        const int width = (int)((slice_ptr[(row >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (row & 63);
        const int k = fs_find_pos_local(sell_col, base, width, col);
        if (k >= 0) atomicAdd(&val[base + (int64_t)k * FS_SLICE], w);
        else atomicAdd(err, 1);
    }
}

// ---- scalar P2 on triangles (2-D meshes with fe_degree 2) --------------------------------------------------------------
// Local nodes: 3 vertices, then the 3 UFC edges (edge i opposite vertex i).  grad phi_vertex_i = (4 lambda_i - 1) grad
// lambda_i, grad phi_edge_(i,j) = 4 (lambda_i grad lambda_j + lambda_j grad lambda_i): the stiffness integrand is quadratic,
// the 3-point edge-midpoint rule exact; exact mass matrix A/180 * FS_P2_TRI_UFC_MASS180.
__device__ __constant__ double FS_P2_TRI_UFC_MASS180[6][6] = {{6, -1, -1, -4, 0, 0}, {-1, 6, -1, 0, -4, 0}, {-1, -1, 6, 0, 0, -4},
                                                               {-4, 0, 0, 32, 16, 16}, {0, -4, 0, 16, 32, 16}, {0, 0, -4, 16, 16, 32}};
__device__ __forceinline__ void p2tri_basis_grads(const tri_geom& t, const double (&lam)[3], double (&gp)[6][2]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int d = 0; d < 2; ++d) gp[i][d] = (4.0 * lam[i] - 1.0) * t.g[i][d];
    const int ei[3] = {1, 0, 0}, ej[3] = {2, 2, 1};
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
        for (int d = 0; d < 2; ++d) gp[3 + e][d] = 4.0 * (lam[ei[e]] * t.g[ej[e]][d] + lam[ej[e]] * t.g[ei[e]][d]);
}
// degree-3 rule on the triangle (Strang-Fix 4 points, one negative weight) for the cubic advection integrand
__device__ __constant__ double FS_TRI4_QP[4][3] = {{1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0}, {0.6, 0.2, 0.2}, {0.2, 0.6, 0.2}, {0.2, 0.2, 0.6}};
__device__ __constant__ double FS_TRI4_QW[4] = {-0.5625, 25.0 / 48.0, 25.0 / 48.0, 25.0 / 48.0};
template <bool ADD>
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p2tri_scalar_gather(
    int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ slice_ptr,
    const int64_t* __restrict__ inc_slice_ptr, int64_t inc_entries, const int32_t* __restrict__ inc_cell,
    const uint32_t* __restrict__ inc_pos, const int32_t* __restrict__ cells, const double* __restrict__ xyz4,
    coef_dev kc, coef_dev mc, double* __restrict__ val, coef_dev ac = coef_dev(), double ascale = 0.0, double supg_pe = 0.0) {
    extern __shared__ __attribute__((aligned(16))) double lds_acc[];  // [width][blockDim.x]
    const int tid = threadIdx.x, bd = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, wpb = bd >> 6;
    const int64_t n_chunks = (n_slices + wpb - 1) / wpb;
    for (chunk_iter it = xcd_chunks(n_chunks); it.cur < it.end; it.cur += it.step) {
        const int64_t s = it.cur * wpb + wave;
        if (s >= n_slices) continue;
        const int64_t base = slice_ptr[s];
        const int width = (int)((slice_ptr[s + 1] - base) >> 6);
        const int64_t ibase = inc_slice_ptr[s];
        const int iwidth = (int)((inc_slice_ptr[s + 1] - ibase) >> 6);
        for (int k = 0; k < width; ++k) lds_acc[k * bd + tid] = 0.0;
        for (int j = 0; j < iwidth; ++j) {
            const int64_t e = ibase + (int64_t)j * FS_SLICE + lane;
            const int32_t q = inc_cell[e];
            if (q < 0) continue;
            const int c = q / 6, a = q - 6 * c;
            const uint32_t pw[2] = {inc_pos[e], inc_pos[inc_entries + e]};
            const int4 c4 = reinterpret_cast<const int4*>(cells)[c];
            const tri_geom t = tri_geometry2(xyz4, c4.x, c4.y, c4.z);
            double row[6] = {0, 0, 0, 0, 0, 0};
            // SUPG: test function q_a + tau (v . grad q_a), its gradient grad q_a + tau H_a v (as in the kernel for tetrahedra)
            double vx = 0.0, vy = 0.0, tau = 0.0, hv[2] = {0.0, 0.0};
            if (ac.mode != FS_COEF_NONE) {
                if (ac.mode == FS_COEF_CONST) { vx = ac.tensor[0]; vy = ac.tensor[1]; }
                else { vx = ac.data[3 * (int64_t)c]; vy = ac.data[3 * (int64_t)c + 1]; }
                if (supg_pe > 0.0) {
                    tau = supg_tau_tri(xyz4, c4.x, c4.y, c4.z, t.area, sqrt(vx * vx + vy * vy), supg_pe);
                    const double gv[3] = {t.g[0][0] * vx + t.g[0][1] * vy, t.g[1][0] * vx + t.g[1][1] * vy, t.g[2][0] * vx + t.g[2][1] * vy};
                    const int hi[3] = {1, 0, 0}, hj[3] = {2, 2, 1};
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        if (b == a) { hv[0] = 4.0 * tau * gv[b] * t.g[b][0]; hv[1] = 4.0 * tau * gv[b] * t.g[b][1]; }
#pragma unroll
                    for (int e2 = 0; e2 < 3; ++e2)
                        if (3 + e2 == a) {
                            hv[0] = 4.0 * tau * (gv[hj[e2]] * t.g[hi[e2]][0] + gv[hi[e2]] * t.g[hj[e2]][0]);
                            hv[1] = 4.0 * tau * (gv[hj[e2]] * t.g[hi[e2]][1] + gv[hi[e2]] * t.g[hj[e2]][1]);
                        }
                }
            }
            if (kc.mode == FS_COEF_CELL_QP) {      // k at the 6 points of the degree-4 rule (data[c][14], the first 6 used)
                const double TQ[6][3] = {{0.108103018168070, 0.445948490915965, 0.445948490915965}, {0.445948490915965, 0.108103018168070, 0.445948490915965},
                                         {0.445948490915965, 0.445948490915965, 0.108103018168070}, {0.816847572980459, 0.091576213509771, 0.091576213509771},
                                         {0.091576213509771, 0.816847572980459, 0.091576213509771}, {0.091576213509771, 0.091576213509771, 0.816847572980459}};
                const double TW[6] = {0.223381589678011, 0.223381589678011, 0.223381589678011, 0.109951743655322, 0.109951743655322, 0.109951743655322};
                for (int qp = 0; qp < 6; ++qp) {
                    const double lam[3] = {TQ[qp][0], TQ[qp][1], TQ[qp][2]};
                    double gp[6][2];
                    p2tri_basis_grads(t, lam, gp);
                    double ga[2] = {0.0, 0.0};
#pragma unroll
                    for (int b = 0; b < 6; ++b)
                        if (b == a) { ga[0] = gp[b][0] + hv[0]; ga[1] = gp[b][1] + hv[1]; }
                    const double w = TW[qp] * t.area * kc.data[14 * (int64_t)c + qp];
#pragma unroll
                    for (int b = 0; b < 6; ++b) row[b] += w * (ga[0] * gp[b][0] + ga[1] * gp[b][1]);
                }
            } else if (kc.mode != FS_COEF_NONE) {
                const double kk = kc.mode == FS_COEF_CONST ? kc.value : kc.data[c];
#pragma unroll
                for (int qp = 0; qp < 3; ++qp) {
                    const double lam[3] = {qp == 0 ? 0.0 : 0.5, qp == 1 ? 0.0 : 0.5, qp == 2 ? 0.0 : 0.5};
                    double gp[6][2];
                    p2tri_basis_grads(t, lam, gp);
                    double ga[2] = {0.0, 0.0};
#pragma unroll
                    for (int b = 0; b < 6; ++b)
                        if (b == a) { ga[0] = gp[b][0] + hv[0]; ga[1] = gp[b][1] + hv[1]; }
#pragma unroll
                    for (int b = 0; b < 6; ++b) row[b] += (1.0 / 3.0) * (ga[0] * gp[b][0] + ga[1] * gp[b][1]);
                }
                const double w = kk * t.area;
#pragma unroll
                for (int b = 0; b < 6; ++b) row[b] *= w;
            }
            if (mc.mode != FS_COEF_NONE) {
                const double mm = (mc.mode == FS_COEF_CONST ? mc.value : mc.data[c]) * t.area * (1.0 / 180.0);
#pragma unroll
                for (int b = 0; b < 6; ++b) row[b] += mm * FS_P2_TRI_UFC_MASS180[a][b];
            }
            if (ac.mode != FS_COEF_NONE) {      // + scale int q_a (v . grad phi_b) dx, constant or per-cell velocity
                const double msupg = (tau != 0.0 && mc.mode != FS_COEF_NONE) ? tau * (mc.mode == FS_COEF_CONST ? mc.value : mc.data[c]) : 0.0;
                for (int qp = 0; qp < 4; ++qp) {
                    const double lam[3] = {FS_TRI4_QP[qp][0], FS_TRI4_QP[qp][1], FS_TRI4_QP[qp][2]};
                    double gp[6][2];
                    p2tri_basis_grads(t, lam, gp);
                    const int ei[3] = {1, 0, 0}, ej[3] = {2, 2, 1};
                    double pb[6];
#pragma unroll
                    for (int b = 0; b < 3; ++b) pb[b] = lam[b] * (2.0 * lam[b] - 1.0);
#pragma unroll
                    for (int e2 = 0; e2 < 3; ++e2) pb[3 + e2] = 4.0 * lam[ei[e2]] * lam[ej[e2]];
                    double pa = 0.0, va = 0.0;
#pragma unroll
                    for (int b = 0; b < 6; ++b)
                        if (b == a) { pa = pb[b]; va = vx * gp[b][0] + vy * gp[b][1]; }
                    const double wq = FS_TRI4_QW[qp] * t.area;
                    const double w = ascale * wq * (pa + tau * va);
                    const double wm = msupg * wq * va;      // SUPG part of the mass term (cubic: this rule is exact)
#pragma unroll
                    for (int b = 0; b < 6; ++b) row[b] += w * (vx * gp[b][0] + vy * gp[b][1]) + wm * pb[b];
                }
            }
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const int k = (pw[b >> 2] >> (8 * (b & 3))) & 255;
                lds_acc[k * bd + tid] += row[b];
            }
        }
        for (int k = 0; k < width; ++k) {
            const int64_t e = base + (int64_t)k * FS_SLICE + lane;
            const double x = lds_acc[k * bd + tid];
            val[e] = ADD ? val[e] + x : x;
        }
    }
}

// load vector: constant / per-cell f: A/3 on the edge nodes, 0 on the vertices; nodal (P2) f: M_e f_e
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p2tri_source_gather(int64_t n_rows, int64_t n_slices,
                                                                           const int64_t* __restrict__ inc_slice_ptr,
                                                                           const int32_t* __restrict__ inc_cell,
                                                                           const int32_t* __restrict__ cell_dofs,
                                                                           const int32_t* __restrict__ cells,
                                                                           const double* __restrict__ xyz4, coef_dev f,
                                                                           double* __restrict__ b, coef_dev sv = coef_dev(),
                                                                           double supg_pe = 0.0) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t row = s * FS_SLICE + lane;
        const int64_t ibase = inc_slice_ptr[s];
        const int iwidth = (int)((inc_slice_ptr[s + 1] - ibase) >> 6);
        double acc = 0.0;
        for (int j = 0; j < iwidth; ++j) {
            const int32_t q = inc_cell[ibase + (int64_t)j * FS_SLICE + lane];
            if (q < 0) continue;
            const int c = q / 6, a = q - 6 * c;
            const int4 c4 = reinterpret_cast<const int4*>(cells)[c];
            const tri_geom t = tri_geometry2(xyz4, c4.x, c4.y, c4.z);
            if (f.mode == FS_COEF_NODAL) {
                double m = 0.0, S[6];
                for (int k = 0; k < 6; ++k) { S[k] = f.data[cell_dofs[(int64_t)c * 6 + k]]; m += FS_P2_TRI_UFC_MASS180[a][k] * S[k]; }
                acc += m * t.area * (1.0 / 180.0);
                if (supg_pe > 0.0 && sv.mode != FS_COEF_NONE) {      // + int S_h tau (v . grad q_a) dx, exactly
                    double vx, vy;
                    if (sv.mode == FS_COEF_CONST) { vx = sv.tensor[0]; vy = sv.tensor[1]; }
                    else { vx = sv.data[3 * (int64_t)c]; vy = sv.data[3 * (int64_t)c + 1]; }
                    const double tau = supg_tau_tri(xyz4, c4.x, c4.y, c4.z, t.area, sqrt(vx * vx + vy * vy), supg_pe);
                    const int ei[3] = {1, 0, 0}, ej[3] = {2, 2, 1};
                    double vg[3];
                    for (int k = 0; k < 3; ++k) vg[k] = vx * t.g[k][0] + vy * t.g[k][1];
                    acc += t.area * tau * p2_supg_nodal_load<3>(S, vg, a, ei, ej);
                }
            } else {
                const double ff = (f.mode == FS_COEF_CONST ? f.value : f.data[c]) * t.area;
                acc += (a < 3 ? 0.0 : 1.0 / 3.0) * ff;
                if (supg_pe > 0.0 && sv.mode != FS_COEF_NONE) {
                    // + int S tau (v . grad q_a) dx: the mean gradient over a triangle (lambda = 1/3) is g_a / 3 for a vertex
                    // function, 4/3 (g_i + g_j) for the edge function ij
                    double vx, vy;
                    if (sv.mode == FS_COEF_CONST) { vx = sv.tensor[0]; vy = sv.tensor[1]; }
                    else { vx = sv.data[3 * (int64_t)c]; vy = sv.data[3 * (int64_t)c + 1]; }
                    const double tau = supg_tau_tri(xyz4, c4.x, c4.y, c4.z, t.area, sqrt(vx * vx + vy * vy), supg_pe);
                    const int ei[3] = {1, 0, 0}, ej[3] = {2, 2, 1};
                    double gx, gy;
                    if (a < 3) { gx = t.g[a][0] * (1.0 / 3.0); gy = t.g[a][1] * (1.0 / 3.0); }
                    else {
                        const int i = ei[a - 3], jj = ej[a - 3];
                        gx = (4.0 / 3.0) * (t.g[i][0] + t.g[jj][0]);
                        gy = (4.0 / 3.0) * (t.g[i][1] + t.g[jj][1]);
                    }
                    acc += ff * tau * (gx * vx + gy * vy);
                }
            }
        }
        if (row < n_rows) b[row] += acc;
    }
}

// node of the P2 edge (a, c) through the sorted edge keys
__device__ __forceinline__ int32_t p2_edge_node_of(int32_t a, int32_t c, const uint64_t* __restrict__ edge_keys, int64_t ne, int grouped,
                                                   const int32_t* __restrict__ edge_node) {
    const uint32_t lo_v = (uint32_t)(a < c ? a : c), hi_v = (uint32_t)(a < c ? c : a);
    const uint64_t key = grouped ? (((uint64_t)(hi_v - lo_v) << 32) | lo_v) : (((uint64_t)lo_v << 32) | hi_v);
    int64_t lo = 0, hi = ne;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (edge_keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < ne && edge_keys[lo] == key) ? edge_node[lo] : -1;
}
// P2 boundary load on an edge: g |e| (1/6, 1/6, 4/6) on its two vertices and its mid node (Simpson, exact for P2)
__global__ void k_edge_vector_p2(const double* __restrict__ xyz4, const int32_t* __restrict__ ed, int64_t nf,
                                 const double* __restrict__ g, const uint64_t* __restrict__ edge_keys, int64_t ne, int grouped,
                                 const int32_t* __restrict__ edge_node, int64_t n_rows, double* __restrict__ b, int* __restrict__ err) {
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; f < nf; f += stride) {
        const int32_t a = ed[2 * f], c = ed[2 * f + 1];
        const double dx = xyz4[4 * (int64_t)c] - xyz4[4 * (int64_t)a], dy = xyz4[4 * (int64_t)c + 1] - xyz4[4 * (int64_t)a + 1];
        const double w = sqrt(dx * dx + dy * dy) * g[f] * (1.0 / 6.0);
        const int32_t m = p2_edge_node_of(a, c, edge_keys, ne, grouped, edge_node);
        if (m < 0) { atomicAdd(err, 1); continue; }
        if (a < n_rows) atomicAdd(&b[a], w);
        if (c < n_rows) atomicAdd(&b[c], w);
        if (m < n_rows) atomicAdd(&b[m], 4.0 * w);
    }
}
// P2 Robin matrix on an edge: h |e| / 30 * [[4 -1 2], [-1 4 2], [2 2 16]] on (a, c, mid).  Thread per (edge, row node).
__global__ void k_edge_matrix_p2(const double* __restrict__ xyz4, const int32_t* __restrict__ ed, int64_t nf,
                                 const double* __restrict__ h, const uint64_t* __restrict__ edge_keys, int64_t ne, int grouped,
                                 const int32_t* __restrict__ edge_node, int64_t n_rows, const int64_t* __restrict__ slice_ptr,
                                 const int32_t* __restrict__ sell_col, double* __restrict__ val, int* __restrict__ err) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nf * 3; t += stride) {
        const int64_t f = t / 3;
        const int i = (int)(t - 3 * f);
        const int32_t a = ed[2 * f], c = ed[2 * f + 1];
        const int32_t m = p2_edge_node_of(a, c, edge_keys, ne, grouped, edge_node);
        if (m < 0) { atomicAdd(err, 1); continue; }
        const int32_t node[3] = {a, c, m};
        const int32_t row = node[i];
        if (row >= n_rows) continue;
        const double dx = xyz4[4 * (int64_t)c] - xyz4[4 * (int64_t)a], dy = xyz4[4 * (int64_t)c + 1] - xyz4[4 * (int64_t)a + 1];
        const double w = h[f] * sqrt(dx * dx + dy * dy) * (1.0 / 30.0);
        const double M3[3][3] = {{4, -1, 2}, {-1, 4, 2}, {2, 2, 16}};
        const int64_t sp0 = slice_ptr[row >> 6];
        const int width = (int)((slice_ptr[(row >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (row & 63);
        for (int j = 0; j < 3; ++j) {
            const int k = fs_find_pos_local(sell_col, base, width, node[j]);
            if (k >= 0) atomicAdd(&val[base + (int64_t)k * FS_SLICE], w * M3[i][j]);
            else atomicAdd(err, 1);
        }
    }
}

// ---- 2-vector P2 on triangles (plane-strain elasticity with fe_degree 2) ---------------------------------------------------
// gradient of local basis function a (0..2 vertices, 3..5 UFC edges) at edge-midpoint quadrature point qp
__device__ __forceinline__ void p2tri_grad_one(const tri_geom& t, int qp, int a, double (&ga)[2]) {
    const double lam[3] = {qp == 0 ? 0.0 : 0.5, qp == 1 ? 0.0 : 0.5, qp == 2 ? 0.0 : 0.5};
    auto g = [&](int n, int d) { return n == 0 ? t.g[0][d] : (n == 1 ? t.g[1][d] : t.g[2][d]); };
    auto l = [&](int n) { return n == 0 ? lam[0] : (n == 1 ? lam[1] : lam[2]); };
    if (a < 3) {
        const double w = 4.0 * l(a) - 1.0;
        ga[0] = w * g(a, 0); ga[1] = w * g(a, 1);
    } else {
        const int i = a == 3 ? 1 : 0, j = a == 5 ? 1 : 2;          // e0=(1,2) e1=(0,2) e2=(0,1)
        ga[0] = 4.0 * (l(i) * g(j, 0) + l(j) * g(i, 0));
        ga[1] = 4.0 * (l(i) * g(j, 1) + l(j) * g(i, 1));
    }
}
// one thread per stored 2x2 block: the (cell, a, b) sources of the inverse slot table (source = cell*36 + a*6 + b), the
// quadratic integrand by the 3-point edge-midpoint rule (exact)
template <bool ADD>
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p2tri_elasticity_gather(int64_t n_entries, const int32_t* __restrict__ ptr,
                                                                               const int32_t* __restrict__ src,
                                                                               const int32_t* __restrict__ cells,
                                                                               const double* __restrict__ xyz4, double mu, double lambda,
                                                                               coef_dev mc, int64_t plane, double* __restrict__ val) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < n_entries; e += stride) {
        double acc[2][2] = {{0, 0}, {0, 0}};
        const int32_t q1 = ptr[e + 1];
        for (int32_t q = ptr[e]; q < q1; ++q) {
            const int32_t sidx = src[q];
            const int64_t c = sidx / 36;
            const int ab = sidx - (int32_t)c * 36, a = ab / 6, b = ab - 6 * a;
            const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
            const tri_geom t = tri_geometry2(xyz4, v4.x, v4.y, v4.z);
            const double w = t.area * (1.0 / 3.0);
#pragma unroll
            for (int qp = 0; qp < 3; ++qp) {
                double ga[2], gb[2];
                p2tri_grad_one(t, qp, a, ga);
                p2tri_grad_one(t, qp, b, gb);
                const double gg = mu * (ga[0] * gb[0] + ga[1] * gb[1]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        double x = lambda * ga[i] * gb[j] + mu * ga[j] * gb[i];
                        if (i == j) x += gg;
                        acc[i][j] += w * x;
                    }
            }
            if (mc.mode != FS_COEF_NONE) {
                const double ms = (mc.mode == FS_COEF_CONST ? mc.value : mc.data[c]) * t.area * (1.0 / 180.0) * FS_P2_TRI_UFC_MASS180[a][b];
                acc[0][0] += ms; acc[1][1] += ms;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t idx = (int64_t)(i * 2 + j) * plane + e;
                val[idx] = ADD ? val[idx] + acc[i][j] : acc[i][j];
            }
    }
}
// load vector: body force int f . phi_a dx (A/3 on edge nodes) + int c d_i phi_a dx with c constant, per cell or P1 through
// its vertex values (nodal array over the space's nodes); c linear x grad phi linear: the 3-point rule is exact
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p2tri_vector_source_gather(int64_t n_rows, const int64_t* __restrict__ slice_ptr,
                                                                                  const int32_t* __restrict__ sell_col,
                                                                                  const int32_t* __restrict__ gptr,
                                                                                  const int32_t* __restrict__ gsrc,
                                                                                  const int32_t* __restrict__ cells,
                                                                                  const double* __restrict__ xyz4, double fx, double fy,
                                                                                  coef_dev dv, int64_t nvo, int64_t neo,
                                                                                  double* __restrict__ b) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_rows; r += stride) {
        const int64_t sp0 = slice_ptr[r >> 6];
        const int width = (int)((slice_ptr[(r >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (r & 63);
        int64_t e = -1;
        for (int k = 0; k < width; ++k)
            if (sell_col[base + (int64_t)k * FS_SLICE] == (int32_t)r) { e = base + (int64_t)k * FS_SLICE; break; }
        double acc[2] = {0.0, 0.0};
        if (e >= 0) {
            for (int32_t q = gptr[e]; q < gptr[e + 1]; ++q) {
                const int32_t sidx = gsrc[q];
                const int64_t c = sidx / 36;
                const int a = (sidx - (int32_t)c * 36) / 6;
                const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
                const tri_geom t = tri_geometry2(xyz4, v4.x, v4.y, v4.z);
                const double wf = a < 3 ? 0.0 : t.area * (1.0 / 3.0);
                acc[0] += wf * fx;
                acc[1] += wf * fy;
                if (dv.mode != FS_COEF_NONE) {
                    double cv[3] = {0.0, 0.0, 0.0};          // c at the three vertices of the cell
                    if (dv.mode == FS_COEF_CONST) cv[0] = cv[1] = cv[2] = dv.value;
                    else if (dv.mode == FS_COEF_CELL) cv[0] = cv[1] = cv[2] = dv.data[c];
                    else {
                        const int32_t vx[3] = {v4.x, v4.y, v4.z};
#pragma unroll
                        for (int k = 0; k < 3; ++k) cv[k] = dv.data[vx[k] < nvo ? vx[k] : vx[k] + neo];      // node of the vertex
                    }
#pragma unroll
                    for (int qp = 0; qp < 3; ++qp) {
                        const double cq = 0.5 * ((qp == 0 ? 0.0 : cv[0]) + (qp == 1 ? 0.0 : cv[1]) + (qp == 2 ? 0.0 : cv[2]));
                        double ga[2];
                        p2tri_grad_one(t, qp, a, ga);
                        acc[0] += t.area * (1.0 / 3.0) * cq * ga[0];
                        acc[1] += t.area * (1.0 / 3.0) * cq * ga[1];
                    }
                }
            }
        }
        b[2 * r + 0] += acc[0];
        b[2 * r + 1] += acc[1];
    }
}
// traction on boundary edges of a 2-vector P2 space: g_i |e| (1/6, 1/6, 4/6) on (a, c, mid)
__global__ void k_edge_vector2_p2(const double* __restrict__ xyz4, const int32_t* __restrict__ ed, int64_t nf,
                                  const double* __restrict__ g, const uint64_t* __restrict__ edge_keys, int64_t ne, int grouped,
                                  const int32_t* __restrict__ edge_node, int64_t n_rows, double* __restrict__ b, int* __restrict__ err) {
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; f < nf; f += stride) {
        const int32_t a = ed[2 * f], c = ed[2 * f + 1];
        const double dx = xyz4[4 * (int64_t)c] - xyz4[4 * (int64_t)a], dy = xyz4[4 * (int64_t)c + 1] - xyz4[4 * (int64_t)a + 1];
        const double w = sqrt(dx * dx + dy * dy) * (1.0 / 6.0);
        const int32_t m = p2_edge_node_of(a, c, edge_keys, ne, grouped, edge_node);
        if (m < 0) { atomicAdd(err, 1); continue; }
        for (int i = 0; i < 2; ++i) {
            const double gi = g[2 * f + i];
            if (a < n_rows) atomicAdd(&b[2 * (int64_t)a + i], w * gi);
            if (c < n_rows) atomicAdd(&b[2 * (int64_t)c + i], w * gi);
            if (m < n_rows) atomicAdd(&b[2 * (int64_t)m + i], 4.0 * w * gi);
        }
    }
}

// ---- load vectors -------------------------------------------------------------------------------
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p1_source(const int32_t* __restrict__ cells,
                                                                 const double* __restrict__ xyz4, int64_t nc,
                                                                 int64_t n_rows, coef_dev f, int ncomp, double fx,
                                                                 double fy, double fz, coef_dev dv, coef_dev sv,
                                                                 double supg_pe, double* __restrict__ b) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < nc; c += stride) {
        const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
        const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
        const tet_geom t = tet_geometry(xyz4, v);
        if (ncomp == 1) {
            double be[4];
            if (f.mode == FS_COEF_NODAL) {
                // b_e = M_e f_e with the exact P1 mass matrix
                const double fe[4] = {f.data[v[0]], f.data[v[1]], f.data[v[2]], f.data[v[3]]};
                const double sum = (fe[0] + fe[1]) + (fe[2] + fe[3]);
                const double m = t.adet * (1.0 / 120.0);
#pragma unroll
                for (int a = 0; a < 4; ++a) be[a] = m * (sum + fe[a]);
                if (supg_pe > 0.0 && sv.mode != FS_COEF_NONE) {     // + int S_h tau (v . grad phi_a) dx = tau (v . g_a) |K| mean(S): exact
                    double vx, vy, vz;
                    if (sv.mode == FS_COEF_CONST) { vx = sv.tensor[0]; vy = sv.tensor[1]; vz = sv.tensor[2]; }
                    else { vx = sv.data[3 * (int64_t)c]; vy = sv.data[3 * (int64_t)c + 1]; vz = sv.data[3 * (int64_t)c + 2]; }
                    const double tau = supg_tau(xyz4, v, t.adet, sqrt(vx * vx + vy * vy + vz * vz), supg_pe);
                    const double sw = 0.25 * sum * t.adet * (1.0 / 6.0) * tau;
#pragma unroll
                    for (int a = 0; a < 4; ++a) be[a] += sw * (vx * t.g[a][0] + vy * t.g[a][1] + vz * t.g[a][2]);
                }
            } else {
                const double ff = f.mode == FS_COEF_CONST ? f.value : f.data[c];
                const double w = ff * t.adet * (1.0 / 24.0);
#pragma unroll
                for (int a = 0; a < 4; ++a) be[a] = w;
                if (supg_pe > 0.0 && sv.mode != FS_COEF_NONE) {     // + int S tau (v . grad phi_a) dx
                    double vx, vy, vz;
                    if (sv.mode == FS_COEF_CONST) { vx = sv.tensor[0]; vy = sv.tensor[1]; vz = sv.tensor[2]; }
                    else { vx = sv.data[3 * (int64_t)c]; vy = sv.data[3 * (int64_t)c + 1]; vz = sv.data[3 * (int64_t)c + 2]; }
                    const double tau = supg_tau(xyz4, v, t.adet, sqrt(vx * vx + vy * vy + vz * vz), supg_pe);
                    const double sw = ff * t.adet * (1.0 / 6.0) * tau;
#pragma unroll
                    for (int a = 0; a < 4; ++a) be[a] += sw * (vx * t.g[a][0] + vy * t.g[a][1] + vz * t.g[a][2]);
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (v[a] < n_rows) atomicAdd(&b[v[a]], be[a]);
        } else {
            const double w = t.adet * (1.0 / 24.0);
            double cd = 0.0;  // int c div v dx = c_cell * vol * grad_a[i]
            if (dv.mode == FS_COEF_CONST) cd = dv.value;
            else if (dv.mode == FS_COEF_CELL) cd = dv.data[c];
            else if (dv.mode == FS_COEF_NODAL) cd = 0.25 * ((dv.data[v[0]] + dv.data[v[1]]) + (dv.data[v[2]] + dv.data[v[3]]));
            cd *= t.adet * (1.0 / 6.0);
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (v[a] < n_rows) {
                    atomicAdd(&b[3 * (int64_t)v[a] + 0], w * fx + cd * t.g[a][0]);
                    atomicAdd(&b[3 * (int64_t)v[a] + 1], w * fy + cd * t.g[a][1]);
                    atomicAdd(&b[3 * (int64_t)v[a] + 2], w * fz + cd * t.g[a][2]);
                }
        }
    }
}

__global__ void k_facet_vector(const double* __restrict__ xyz4, const int32_t* __restrict__ tri, int64_t nf,
                               const double* __restrict__ g, int ncomp, int64_t n_rows, double* __restrict__ b) {
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; f < nf; f += stride) {
        const int32_t v[3] = {tri[3 * f], tri[3 * f + 1], tri[3 * f + 2]};
        const double w = tri_area(xyz4, v[0], v[1], v[2]) * (1.0 / 3.0);
        for (int a = 0; a < 3; ++a) {
            if (v[a] >= n_rows) continue;
            for (int i = 0; i < ncomp; ++i) atomicAdd(&b[(int64_t)v[a] * ncomp + i], w * g[f * ncomp + i]);
        }
    }
}

__global__ void k_facet_matrix(const double* __restrict__ xyz4, const int32_t* __restrict__ tri, int64_t nf,
                               const double* __restrict__ h, int64_t n_rows, const int32_t* __restrict__ sell_col,
                               const int64_t* __restrict__ slice_ptr, double* __restrict__ val,
                               int* __restrict__ err) {
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; f < nf; f += stride) {
        const int32_t v[3] = {tri[3 * f], tri[3 * f + 1], tri[3 * f + 2]};
        const double w = h[f] * tri_area(xyz4, v[0], v[1], v[2]) * (1.0 / 12.0);
        for (int a = 0; a < 3; ++a) {
            const int32_t row = v[a];
            if (row >= n_rows) continue;
            const int64_t sp0 = slice_ptr[row >> 6];
            const int width = (int)((slice_ptr[(row >> 6) + 1] - sp0) >> 6);
            const int64_t base = sp0 + (row & 63);
            for (int b = 0; b < 3; ++b) {
                int k = -1;
                for (int kk = 0; kk < width; ++kk)
                    if (sell_col[base + (int64_t)kk * FS_SLICE] == v[b]) { k = kk; break; }
                if (k >= 0) atomicAdd(&val[base + (int64_t)k * FS_SLICE], (a == b ? 2.0 : 1.0) * w);
                else atomicAdd(err, 1);
            }
        }
    }
}

// ---- interior penalty (ScalarTransportSolver.py:312-315) ----------------------------------------------------------
//   + alpha avg(h)^2 jump(grad T, n) jump(grad q, n) capacity dS        over the interior facets, h = 2 circumradius.
// CG1: the gradients are constant per cell, so a facet F = K+ n K- contributes  w J_i J_j  to the five nodes of the two
// cells, J_i = grad phi_i^+ . n^+ + grad phi_i^- . n^-,  w = coef avg(h)^2 |F|.  The rows couple the two vertices
// opposite the facet, which share no cell: the space must have been created with those pairs
// (fs_space_create_coupled).  One thread per (facet, row node); fp64 atomics.
__device__ __forceinline__ double tet_circum_h(const double* __restrict__ xyz4, const int32_t (&v)[4], double adet) {
    double x0[3], x1[3], x2[3], x3[3];
    load_vertex(xyz4, v[0], x0);
    load_vertex(xyz4, v[1], x1);
    load_vertex(xyz4, v[2], x2);
    load_vertex(xyz4, v[3], x3);
    auto dist = [](const double (&p)[3], const double (&q)[3]) {
        return sqrt((p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]) + (p[2] - q[2]) * (p[2] - q[2]));
    };
    const double aA = dist(x0, x1) * dist(x2, x3), bB = dist(x0, x2) * dist(x1, x3), cC = dist(x0, x3) * dist(x1, x2);
    const double prod = (aA + bB + cC) * (aA + bB - cC) * (aA - bB + cC) * (-aA + bB + cC);
    return 2.0 * sqrt(prod > 0.0 ? prod : 0.0) / (4.0 * adet);
}
__global__ void k_interior_penalty(int64_t nf, const int32_t* __restrict__ facet_cells, const int32_t* __restrict__ cells,
                                   const double* __restrict__ xyz4, double coef, int64_t n_rows,
                                   const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ sell_col,
                                   double* __restrict__ val, int* __restrict__ err) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nf * 5; t += stride) {
        const int64_t f = t / 5;
        const int a = (int)(t - f * 5);
        const int64_t c0 = facet_cells[2 * f], c1 = facet_cells[2 * f + 1];
        const int4 p = reinterpret_cast<const int4*>(cells)[c0], q = reinterpret_cast<const int4*>(cells)[c1];
        const int32_t v0[4] = {p.x, p.y, p.z, p.w}, v1[4] = {q.x, q.y, q.z, q.w};
        int o0 = -1, o1 = -1;          // local index of the vertex opposite the facet in either cell
        for (int i = 0; i < 4; ++i) {
            bool in1 = false, in0 = false;
            for (int j = 0; j < 4; ++j) { in1 |= v0[i] == v1[j]; in0 |= v1[i] == v0[j]; }
            if (!in1) o0 = o0 < 0 ? i : 4;
            if (!in0) o1 = o1 < 0 ? i : 4;
        }
        if (o0 < 0 || o0 > 3 || o1 < 0 || o1 > 3) { if (a == 0) atomicAdd(err, 1); continue; }   // not a shared facet
        const tet_geom g0 = tet_geometry(xyz4, v0), g1 = tet_geometry(xyz4, v1);
        // outward normal of K+ on the facet: -grad lambda_opposite / |grad lambda_opposite|;  |F| = 3 V |grad lambda_opp|
        const double gn = sqrt(g0.g[o0][0] * g0.g[o0][0] + g0.g[o0][1] * g0.g[o0][1] + g0.g[o0][2] * g0.g[o0][2]);
        const double n[3] = {-g0.g[o0][0] / gn, -g0.g[o0][1] / gn, -g0.g[o0][2] / gn};
        const double area = 0.5 * g0.adet * gn;
        const double hbar = 0.5 * (tet_circum_h(xyz4, v0, g0.adet) + tet_circum_h(xyz4, v1, g1.adet));
        const double w = coef * hbar * hbar * area;
        // the five nodes: the vertices of K+ and the vertex of K- opposite the facet
        int32_t node[5];
        double J[5];
        for (int i = 0; i < 4; ++i) {
            node[i] = v0[i];
            double j = g0.g[i][0] * n[0] + g0.g[i][1] * n[1] + g0.g[i][2] * n[2];
            for (int k = 0; k < 4; ++k)
                if (v1[k] == v0[i]) j -= g1.g[k][0] * n[0] + g1.g[k][1] * n[1] + g1.g[k][2] * n[2];
            J[i] = j;
        }
        node[4] = v1[o1];
        J[4] = -(g1.g[o1][0] * n[0] + g1.g[o1][1] * n[1] + g1.g[o1][2] * n[2]);
        const int32_t row = node[a];
        if (row >= n_rows) continue;
        const int64_t sp0 = slice_ptr[row >> 6];
        const int width = (int)((slice_ptr[(row >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (row & 63);
        for (int b = 0; b < 5; ++b) {
            int k = -1;
            for (int kk = 0; kk < width; ++kk)
                if (sell_col[base + (int64_t)kk * FS_SLICE] == node[b]) { k = kk; break; }
            if (k >= 0) atomicAdd(&val[base + (int64_t)k * FS_SLICE], w * J[a] * J[b]);
            else atomicAdd(err, 1);
        }
    }
}

// The same on triangles: an interior edge E = K+ n K- and the four nodes of the two cells; h = 2 circumradius = abc / (2 A).
__device__ __forceinline__ double tri_circum_h(const double* __restrict__ xyz4, const int32_t (&v)[3], double area) {
    double x[3][2];
    for (int i = 0; i < 3; ++i) { x[i][0] = xyz4[4 * (int64_t)v[i]]; x[i][1] = xyz4[4 * (int64_t)v[i] + 1]; }
    auto dist = [&](int p, int q) { return sqrt((x[p][0] - x[q][0]) * (x[p][0] - x[q][0]) + (x[p][1] - x[q][1]) * (x[p][1] - x[q][1])); };
    return dist(0, 1) * dist(1, 2) * dist(2, 0) / (2.0 * area);
}
__global__ void k_interior_penalty_tri(int64_t nf, const int32_t* __restrict__ facet_cells, const int32_t* __restrict__ cells,
                                       const double* __restrict__ xyz4, double coef, int64_t n_rows,
                                       const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ sell_col,
                                       double* __restrict__ val, int* __restrict__ err) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nf * 4; t += stride) {
        const int64_t f = t >> 2;
        const int a = (int)(t & 3);
        const int64_t c0 = facet_cells[2 * f], c1 = facet_cells[2 * f + 1];
        const int4 p = reinterpret_cast<const int4*>(cells)[c0], q = reinterpret_cast<const int4*>(cells)[c1];
        const int32_t v0[3] = {p.x, p.y, p.z}, v1[3] = {q.x, q.y, q.z};
        int o0 = -1, o1 = -1;          // local index of the vertex opposite the shared edge in either cell
        for (int i = 0; i < 3; ++i) {
            bool in1 = false, in0 = false;
            for (int j = 0; j < 3; ++j) { in1 |= v0[i] == v1[j]; in0 |= v1[i] == v0[j]; }
            if (!in1) o0 = o0 < 0 ? i : 3;
            if (!in0) o1 = o1 < 0 ? i : 3;
        }
        if (o0 < 0 || o0 > 2 || o1 < 0 || o1 > 2) { if (a == 0) atomicAdd(err, 1); continue; }
        const tri_geom g0 = tri_geometry2(xyz4, v0[0], v0[1], v0[2]), g1 = tri_geometry2(xyz4, v1[0], v1[1], v1[2]);
        // outward normal of K+ on the edge: -grad lambda_opposite / |grad lambda_opposite|;  |E| = 2 A |grad lambda_opp|
        const double gn = sqrt(g0.g[o0][0] * g0.g[o0][0] + g0.g[o0][1] * g0.g[o0][1]);
        const double n[2] = {-g0.g[o0][0] / gn, -g0.g[o0][1] / gn};
        const double len = 2.0 * g0.area * gn;
        const double hbar = 0.5 * (tri_circum_h(xyz4, v0, g0.area) + tri_circum_h(xyz4, v1, g1.area));
        const double w = coef * hbar * hbar * len;
        int32_t node[4];
        double J[4];
        for (int i = 0; i < 3; ++i) {
            node[i] = v0[i];
            double j = g0.g[i][0] * n[0] + g0.g[i][1] * n[1];
            for (int k = 0; k < 3; ++k)
                if (v1[k] == v0[i]) j -= g1.g[k][0] * n[0] + g1.g[k][1] * n[1];
            J[i] = j;
        }
        node[3] = v1[o1];
        J[3] = -(g1.g[o1][0] * n[0] + g1.g[o1][1] * n[1]);
        const int32_t row = node[a];
        if (row >= n_rows) continue;
        const int64_t sp0 = slice_ptr[row >> 6];
        const int width = (int)((slice_ptr[(row >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (row & 63);
        for (int b = 0; b < 4; ++b) {
            int k = -1;
            for (int kk = 0; kk < width; ++kk)
                if (sell_col[base + (int64_t)kk * FS_SLICE] == node[b]) { k = kk; break; }
            if (k >= 0) atomicAdd(&val[base + (int64_t)k * FS_SLICE], w * J[a] * J[b]);
            else atomicAdd(err, 1);
        }
    }
}

// ---- interior penalty on CG2 spaces (the reference's term is degree-agnostic, ScalarTransportSolver.py:312-315) -------------
// jump(grad phi_a, n) of a P2 basis function is LINEAR along the facet, the integrand quadratic: the 3 edge mid-points of the
// facet (weights 1/3) on tetrahedra, 2-point Gauss-Legendre on the shared edge of two triangles - both exact.  Nodes of a facet
// patch: the NK nodes of K+ and the NK - NF nodes K- does not share (its opposite vertex and the edges from it): 14 / 9.
// Thread (facet, a): all jumps at the quadrature points (shared-node contributions of K- found by node id), then row a.
template <int TD>
__device__ __forceinline__ void ip_p2_grad_n(int n, const double* lam, const double (*gl)[3], const double* nrm, double* out) {
    // grad phi_n . nrm at barycentric lam (TD+1 coordinates), UFC edge order of the cell type
    constexpr int NV = TD + 1;
    const int EI3[6] = {2, 1, 1, 0, 0, 0}, EJ3[6] = {3, 3, 2, 3, 2, 1}, EI2[3] = {1, 0, 0}, EJ2[3] = {2, 2, 1};
    double g[3] = {0.0, 0.0, 0.0};
    if (n < NV) {
        const double d = 4.0 * lam[n] - 1.0;
        for (int k = 0; k < TD; ++k) g[k] = d * gl[n][k];
    } else {
        const int i = TD == 3 ? EI3[n - NV] : EI2[n - NV], j = TD == 3 ? EJ3[n - NV] : EJ2[n - NV];
        for (int k = 0; k < TD; ++k) g[k] = 4.0 * (lam[i] * gl[j][k] + lam[j] * gl[i][k]);
    }
    double r = 0.0;
    for (int k = 0; k < TD; ++k) r += g[k] * nrm[k];
    *out = r;
}

template <int TD>
__global__ void k_interior_penalty_p2(int64_t nf, const int32_t* __restrict__ facet_cells, const int32_t* __restrict__ cells,
                                      const int32_t* __restrict__ cell_dofs, const double* __restrict__ xyz4, double coef,
                                      int64_t n_rows, const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ sell_col,
                                      double* __restrict__ val, int* __restrict__ err) {
    constexpr int NV = TD + 1, NK = TD == 3 ? 10 : 6, NF = TD == 3 ? 6 : 3, NP = 2 * NK - NF, NQ = TD == 3 ? 3 : 2;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nf * NP; t += stride) {
        const int64_t f = t / NP;
        const int a = (int)(t - f * NP);
        const int64_t c0 = facet_cells[2 * f], c1 = facet_cells[2 * f + 1];
        int32_t v0[4], v1[4];
        for (int i = 0; i < NV; ++i) { v0[i] = cells[c0 * 4 + i]; v1[i] = cells[c1 * 4 + i]; }
        int o0 = -1, o1 = -1;
        for (int i = 0; i < NV; ++i) {
            bool in1 = false, in0 = false;
            for (int j = 0; j < NV; ++j) { in1 |= v0[i] == v1[j]; in0 |= v1[i] == v0[j]; }
            if (!in1) o0 = o0 < 0 ? i : NV;
            if (!in0) o1 = o1 < 0 ? i : NV;
        }
        if (o0 < 0 || o0 >= NV || o1 < 0 || o1 >= NV) { if (a == 0) atomicAdd(err, 1); continue; }
        double g0[4][3], g1[4][3], h0, h1, meas, nrm[3] = {0.0, 0.0, 0.0};
        if (TD == 3) {
            const int32_t w0[4] = {v0[0], v0[1], v0[2], v0[3]}, w1[4] = {v1[0], v1[1], v1[2], v1[3]};
            const tet_geom G0 = tet_geometry(xyz4, w0), G1 = tet_geometry(xyz4, w1);
            for (int i = 0; i < 4; ++i) for (int k = 0; k < 3; ++k) { g0[i][k] = G0.g[i][k]; g1[i][k] = G1.g[i][k]; }
            h0 = tet_circum_h(xyz4, w0, G0.adet);
            h1 = tet_circum_h(xyz4, w1, G1.adet);
            const double gn = sqrt(g0[o0][0] * g0[o0][0] + g0[o0][1] * g0[o0][1] + g0[o0][2] * g0[o0][2]);
            for (int k = 0; k < 3; ++k) nrm[k] = -g0[o0][k] / gn;
            meas = 0.5 * G0.adet * gn;
        } else {
            const int32_t w0[3] = {v0[0], v0[1], v0[2]}, w1[3] = {v1[0], v1[1], v1[2]};
            const tri_geom G0 = tri_geometry2(xyz4, w0[0], w0[1], w0[2]), G1 = tri_geometry2(xyz4, w1[0], w1[1], w1[2]);
            for (int i = 0; i < 3; ++i) for (int k = 0; k < 2; ++k) { g0[i][k] = G0.g[i][k]; g1[i][k] = G1.g[i][k]; }
            h0 = tri_circum_h(xyz4, w0, G0.area);
            h1 = tri_circum_h(xyz4, w1, G1.area);
            const double gn = sqrt(g0[o0][0] * g0[o0][0] + g0[o0][1] * g0[o0][1]);
            nrm[0] = -g0[o0][0] / gn; nrm[1] = -g0[o0][1] / gn;
            meas = 2.0 * G0.area * gn;
        }
        const double hbar = 0.5 * (h0 + h1);
        const double wf = coef * hbar * hbar * meas;
        // patch nodes: the NK nodes of K+, then the nodes of K- that K+ does not have
        int32_t node[NP];
        int loc1[NP];                  // local index of the patch node in K-, -1 if absent
        for (int i = 0; i < NK; ++i) { node[i] = cell_dofs[c0 * NK + i]; loc1[i] = -1; }
        int np = NK;
        for (int j = 0; j < NK; ++j) {
            const int32_t nd = cell_dofs[c1 * NK + j];
            int hit = -1;
            for (int i = 0; i < NK; ++i) if (node[i] == nd) hit = i;
            if (hit >= 0) loc1[hit] = j;
            else if (np < NP) { node[np] = nd; loc1[np] = j; ++np; }
        }
        if (np != NP) { if (a == 0) atomicAdd(err, 1); continue; }
        // quadrature points on the facet in the barycentric coordinates of either cell
        double J[NQ][NP];
        for (int q = 0; q < NQ; ++q) {
            double bary[3];               // weights of the facet's vertices, in the order they appear in K+
            if (TD == 3) { bary[0] = bary[1] = bary[2] = 0.5; bary[q] = 0.0; }
            else { const double s_ = q == 0 ? 0.21132486540518713 : 0.7886751345948129; bary[0] = 1.0 - s_; bary[1] = s_; bary[2] = 0.0; }
            double l0[4] = {0.0, 0.0, 0.0, 0.0}, l1[4] = {0.0, 0.0, 0.0, 0.0};
            int kf = 0;
            for (int i = 0; i < NV; ++i) {
                if (i == o0) continue;
                l0[i] = bary[kf];
                for (int j = 0; j < NV; ++j) if (v1[j] == v0[i]) l1[j] = bary[kf];
                ++kf;
            }
            for (int m = 0; m < NP; ++m) {
                double jv = 0.0, tmp;
                if (m < NK) { ip_p2_grad_n<TD>(m, l0, g0, nrm, &tmp); jv += tmp; }
                if (loc1[m] >= 0) { ip_p2_grad_n<TD>(loc1[m], l1, g1, nrm, &tmp); jv -= tmp; }
                J[q][m] = jv;
            }
        }
        const int32_t row = node[a];
        if (row >= n_rows) continue;
        const int64_t sp0 = slice_ptr[row >> 6];
        const int width = (int)((slice_ptr[(row >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (row & 63);
        for (int b = 0; b < NP; ++b) {
            double e = 0.0;
            for (int q = 0; q < NQ; ++q) e += J[q][a] * J[q][b];
            e *= wf / NQ;                  // equal weights: 1/3 (edge mid-points of the triangle), 1/2 (2-point Gauss)
            int k = -1;
            for (int kk = 0; kk < width; ++kk)
                if (sell_col[base + (int64_t)kk * FS_SLICE] == node[b]) { k = kk; break; }
            if (k >= 0) atomicAdd(&val[base + (int64_t)k * FS_SLICE], e);
            else atomicAdd(err, 1);
        }
    }
}

// ---- Dirichlet ------------------------------------------------------------------------------------
// "later entries win" without a host pass: first the largest list index naming each dof ...
__global__ void k_bc_last_index(const int32_t* __restrict__ dofs, int64_t n, int32_t* __restrict__ idx) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) atomicMax(&idx[dofs[i]], (int32_t)i);
}

// ... then only that entry writes flag and value (deterministic for any duplicate pattern)
__global__ void k_bc_scatter(const int32_t* __restrict__ dofs, const double* __restrict__ vals, int64_t n,
                             const int32_t* __restrict__ idx, uint8_t* __restrict__ flag, double* __restrict__ g) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int32_t d = dofs[i];
        if (idx[d] == (int32_t)i) {
            flag[d] = 1;
            g[d] = vals[i];
        }
    }
}

__global__ void k_bc_vector(const uint8_t* __restrict__ flag, const double* __restrict__ g, int64_t n_rows_dofs,
                            double* __restrict__ b) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_rows_dofs; i += stride)
        if (flag[i]) b[i] = g[i];
}

// one wavefront per slice, lane = node row; rows of constrained dofs become identity,
// (symmetric) constrained columns are folded into b and zeroed.  Non-structural entries (negative
// column) are left untouched: they are zero and stay zero.
template <int BS>
__global__ void __launch_bounds__(FS_BLOCK) k_dirichlet_sell(int64_t n_rows, int64_t n_slices,
                                                             const int64_t* __restrict__ slice_ptr,
                                                             const int32_t* __restrict__ sell_col,
                                                             double* __restrict__ val, int64_t plane,
                                                             const uint8_t* __restrict__ flag,
                                                             const double* __restrict__ g, double* __restrict__ b,
                                                             int symmetric) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t r = s * FS_SLICE + lane;
        if (r >= n_rows) continue;
        const int64_t base = slice_ptr[s] + lane;
        const int width = (int)((slice_ptr[s + 1] - slice_ptr[s]) >> 6);
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            const int64_t dofr = r * BS + i;
            const bool fr = flag[dofr] != 0;
            double bacc = 0.0;
            for (int k = 0; k < width; ++k) {
                const int64_t e = base + (int64_t)k * FS_SLICE;
                const int32_t c = sell_col[e];
                if (c < 0) continue;
#pragma unroll
                for (int j = 0; j < BS; ++j) {
                    const int64_t idx = (int64_t)(i * BS + j) * plane + e;
                    if (fr) {
                        val[idx] = (c == r && i == j) ? 1.0 : 0.0;
                    } else if (symmetric && flag[(int64_t)c * BS + j]) {
                        bacc += val[idx] * g[(int64_t)c * BS + j];
                        val[idx] = 0.0;
                    }
                }
            }
            if (b) {
                if (fr) b[dofr] = g[dofr];
                else if (symmetric) b[dofr] -= bacc;
            }
        }
    }
}

// ---- tied nodes (periodic constraints: FunctionSpace(..., constrained_domain=pb), SolverBase.py:260-275) ---------------
// DOLFIN removes the slave dofs; here they stay in the vectors and the assembled system is folded instead: with P the
// matrix that copies every master value to its slaves, A <- P^T A P + (unit diagonal on the slave rows), b <- P^T b (0 on
// the slaves); the solution of the folded system carries the masters' values, which fs_vector_assign_entries then copies
// to the slaves.  The sparsity pattern has to hold (master, fold(j)) for every neighbour j of a slave:
// fs_space_create_coupled with those pairs.
// Pass 1, lane = node row: the columns of slave nodes move to their masters' columns (one lane owns the row: no atomics).
template <int BS>
__global__ void __launch_bounds__(FS_BLOCK) k_tie_columns(int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ slice_ptr,
                                                          const int32_t* __restrict__ sell_col, double* __restrict__ val,
                                                          int64_t plane, const int32_t* __restrict__ master_of,
                                                          int* __restrict__ err) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t r = s * FS_SLICE + lane;
        if (r >= n_rows) continue;
        const int64_t base = slice_ptr[s] + lane;
        const int width = (int)((slice_ptr[s + 1] - slice_ptr[s]) >> 6);
        for (int k = 0; k < width; ++k) {
            const int64_t e = base + (int64_t)k * FS_SLICE;
            const int32_t c = sell_col[e];
            if (c < 0) continue;
            const int32_t m = master_of[c];
            if (m < 0) continue;
            int64_t em = -1;
            for (int kk = 0; kk < width; ++kk)
                if (sell_col[base + (int64_t)kk * FS_SLICE] == m) { em = base + (int64_t)kk * FS_SLICE; break; }
            bool any = false;
#pragma unroll
            for (int q = 0; q < BS * BS; ++q) any = any || val[(int64_t)q * plane + e] != 0.0;
            if (em < 0) {
                if (any) atomicAdd(err, 1);
                continue;
            }
#pragma unroll
            for (int q = 0; q < BS * BS; ++q) {
                val[(int64_t)q * plane + em] += val[(int64_t)q * plane + e];
                val[(int64_t)q * plane + e] = 0.0;
            }
        }
    }
}
// Pass 2, thread = tied pair: the slave's row is added to its master's row (several slaves may share a master: atomics;
// a handful of rows), the slave keeps a unit diagonal and a zero right-hand side.
template <int BS>
__global__ void k_tie_rows(int64_t n_pairs, const int32_t* __restrict__ slaves, const int32_t* __restrict__ masters,
                           int64_t n_rows, const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ sell_col,
                           double* __restrict__ val, int64_t plane, double* __restrict__ b, int* __restrict__ err) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n_pairs; t += stride) {
        const int64_t sl = slaves[t], ma = masters[t];
        if (sl >= n_rows) continue;           // a ghost slave (decomposed space): its row lives with its master, on another rank
        if (ma >= n_rows) { atomicAdd(err, 1); continue; }
        const int64_t sp_s = slice_ptr[sl >> 6], sp_m = slice_ptr[ma >> 6];
        const int w_s = (int)((slice_ptr[(sl >> 6) + 1] - sp_s) >> 6), w_m = (int)((slice_ptr[(ma >> 6) + 1] - sp_m) >> 6);
        const int64_t base_s = sp_s + (sl & 63), base_m = sp_m + (ma & 63);
        for (int k = 0; k < w_s; ++k) {
            const int64_t e = base_s + (int64_t)k * FS_SLICE;
            const int32_t c = sell_col[e];
            if (c < 0) continue;
            bool any = false;
#pragma unroll
            for (int q = 0; q < BS * BS; ++q) any = any || val[(int64_t)q * plane + e] != 0.0;
            if (any) {
                int64_t em = -1;
                for (int kk = 0; kk < w_m; ++kk)
                    if (sell_col[base_m + (int64_t)kk * FS_SLICE] == c) { em = base_m + (int64_t)kk * FS_SLICE; break; }
                if (em < 0) { atomicAdd(err, 1); continue; }
#pragma unroll
                for (int q = 0; q < BS * BS; ++q) atomicAdd(&val[(int64_t)q * plane + em], val[(int64_t)q * plane + e]);
            }
#pragma unroll
            for (int i = 0; i < BS; ++i)
#pragma unroll
                for (int j = 0; j < BS; ++j) val[(int64_t)(i * BS + j) * plane + e] = (c == (int32_t)sl && i == j) ? 1.0 : 0.0;
        }
        if (b) {
#pragma unroll
            for (int i = 0; i < BS; ++i) {
                atomicAdd(&b[ma * BS + i], b[sl * BS + i]);
                b[sl * BS + i] = 0.0;
            }
        }
    }
}
__global__ void k_fill_master_of(int64_t n_pairs, const int32_t* __restrict__ slaves, const int32_t* __restrict__ masters,
                                 int32_t* __restrict__ master_of) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n_pairs; t += stride) master_of[slaves[t]] = masters[t];
}

extern "C" int fs_matrix_tie_nodes(fs_matrix_t A, fs_vector_t b, int64_t n_pairs, const int32_t* slaves, const int32_t* masters) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(A && (n_pairs == 0 || (slaves && masters)), "fs_matrix_tie_nodes: null pointer");
    if (n_pairs == 0) return FS_OK;
    fs_space_s* sp = A->space;
    FS_REQUIRE(A->bs >= 1 && A->bs <= 4, "fs_matrix_tie_nodes: block size %d", A->bs);
    FS_REQUIRE(!b || b->d.n >= sp->n_dofs_owned, "fs_matrix_tie_nodes: right-hand side too short");
    // Decomposed spaces: local node numbers, ghosts included.  The COLUMNS of every local slave fold onto its (local) master in
    // the rows this rank owns; the ROW of a slave folds where the slave is owned - its master must be owned by the same rank
    // (the partition gives a slave its master's rank, partition.build_local_part(tied=...)).
    for (int64_t i = 0; i < n_pairs; ++i)
        FS_REQUIRE(slaves[i] >= 0 && slaves[i] < sp->n_nodes_local && masters[i] >= 0 && masters[i] < sp->n_nodes_local && slaves[i] != masters[i] &&
                   (slaves[i] >= sp->n_nodes_owned || masters[i] < sp->n_nodes_owned),
                   "fs_matrix_tie_nodes: pair %lld (%d -> %d) out of range (a slave this rank owns needs its master on this rank too)", (long long)i, slaves[i], masters[i]);
    hipStream_t s = fs_rt().stream;
    dbuf<int32_t> d_s, d_m, master_of;
    dbuf<int> d_err;
    FS_CHECK(d_s.alloc(n_pairs)); FS_CHECK(d_m.alloc(n_pairs)); FS_CHECK(master_of.alloc(sp->n_nodes_local)); FS_CHECK(d_err.alloc(1));
    FS_CHECK(d_s.upload(slaves, n_pairs, s)); FS_CHECK(d_m.upload(masters, n_pairs, s));
    FS_CHECK(d_err.zero(s));
    FS_HIP(hipMemsetAsync(master_of.p, 0xff, (size_t)sp->n_nodes_local * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_fill_master_of, dim3(fs_grid_for(n_pairs)), dim3(FS_BLOCK), 0, s, n_pairs, d_s.p, d_m.p, master_of.p);
    const int grid = fs_grid_for(sp->n_slices * 64, FS_BLOCK, 8192), gp = fs_grid_for(n_pairs);
    double* bp = b ? b->d.p : nullptr;
#define FS_TIE(BS_)                                                                                                                       \
    {                                                                                                                                     \
        hipLaunchKernelGGL(k_tie_columns<BS_>, dim3(grid), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p,        \
                           sp->sell_col.p, A->val.p, sp->sell_entries, master_of.p, d_err.p);                                             \
        hipLaunchKernelGGL(k_tie_rows<BS_>, dim3(gp), dim3(FS_BLOCK), 0, s, n_pairs, d_s.p, d_m.p, sp->n_nodes_owned, sp->slice_ptr.p,    \
                           sp->sell_col.p, A->val.p, sp->sell_entries, bp, d_err.p);                                                      \
    }
    if (A->bs == 1) FS_TIE(1) else if (A->bs == 2) FS_TIE(2) else if (A->bs == 3) FS_TIE(3) else FS_TIE(4)
#undef FS_TIE
    if (A->bs == 4 && A->taylor_hood) fs_ns_reset_dummy_rows(A, s);     // the folded dummy diagonals of edge masters: 2 -> 1
    FS_KERNEL_CHECK();
    int h_err = 0;
    FS_CHECK(d_err.download(&h_err, 1, s));
    FS_REQUIRE(h_err == 0, "fs_matrix_tie_nodes: %d folded entries have no place in the sparsity pattern (create the space with fs_space_create_coupled and the (master, neighbour-of-slave) pairs)", h_err);
    return FS_OK;
}

// ---- CSR export ---------------------------------------------------------------------------------------
template <int BS>
__global__ void k_export_csr(int64_t n_rows, const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ rowptr,
                             const int32_t* __restrict__ sell_col, const double* __restrict__ val, int64_t plane,
                             int32_t* __restrict__ out_rowptr, int32_t* __restrict__ out_col,
                             double* __restrict__ out_val) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_rows; r += stride) {
        const int64_t sp0 = slice_ptr[r >> 6];
        const int width = (int)((slice_ptr[(r >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (r & 63);
        const int32_t start = rowptr[r];
        const int len = rowptr[r + 1] - start;
        for (int i = 0; i < BS; ++i) {
            const int64_t o = (int64_t)BS * BS * start + (int64_t)i * BS * len;
            if (out_rowptr) out_rowptr[r * BS + i] = (int32_t)o;
            int kk = 0;  // structural entries appear in ascending column order in both storage forms
            for (int k = 0; k < width; ++k) {
                const int64_t e = base + (int64_t)k * FS_SLICE;
                const int32_t c = sell_col[e];
                if (c < 0) continue;
                for (int j = 0; j < BS; ++j) {
                    if (out_col) out_col[o + (int64_t)kk * BS + j] = c * BS + j;
                    if (out_val) out_val[o + (int64_t)kk * BS + j] = val[(int64_t)(i * BS + j) * plane + e];
                }
                ++kk;
            }
        }
        if (r == n_rows - 1 && out_rowptr) out_rowptr[n_rows * BS] = (int32_t)((int64_t)BS * BS * rowptr[n_rows]);
    }
}

__global__ void k_mat_axpy(double* __restrict__ y, const double* __restrict__ x, int64_t n, double a) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] += a * x[i];
}

// ---- host side -----------------------------------------------------------------------------------------
static int make_coef(const fs_coef& in, int64_t expect_len, dbuf<double>& store, coef_dev* out, const char* what) {
    out->mode = in.mode;
    out->value = in.value;
    out->data = nullptr;
    for (int i = 0; i < 9; ++i) out->tensor[i] = in.tensor[i];
    if (in.mode == FS_COEF_CELL_TENSOR) expect_len *= 9;
    if (in.mode == FS_COEF_CELL_QP) expect_len *= 14;
    if (in.mode == FS_COEF_CELL || in.mode == FS_COEF_NODAL || in.mode == FS_COEF_CELL_ROW || in.mode == FS_COEF_CELL_TENSOR ||
        in.mode == FS_COEF_CELL_QP) {
        FS_REQUIRE(in.data, "%s: coefficient data pointer is null", what);
        FS_CHECK(store.alloc(expect_len));
        FS_CHECK(store.upload(in.data, expect_len, fs_rt().stream));
        out->data = store.p;
    } else if (in.mode != FS_COEF_NONE && in.mode != FS_COEF_CONST && in.mode != FS_COEF_TENSOR) {
        fs_set_error("%s: unknown coefficient mode %d", what, in.mode);
        return FS_ERR_INVALID;
    }
    return FS_OK;
}

extern "C" int fs_matrix_create(fs_space_t space, fs_matrix_t* out) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(space && out, "fs_matrix_create: null pointer");
    fs_matrix_s* A = new fs_matrix_s();
    A->space = space;
    A->bs = space->ncomp;
    int rc = A->val.alloc(space->sell_entries * A->bs * A->bs);
    if (rc == FS_OK) rc = A->val.zero(fs_rt().stream);
    if (rc == FS_OK && hipStreamSynchronize(fs_rt().stream) != hipSuccess) rc = FS_ERR_HIP;
    if (rc != FS_OK) {
        delete A;
        return rc;
    }
    *out = A;
    return FS_OK;
}

extern "C" int fs_matrix_info(fs_matrix_t A, int64_t* n_rows, int64_t* n_cols, int64_t* nnz) {
    FS_REQUIRE(A, "fs_matrix_info: null matrix");
    if (n_rows) *n_rows = A->space->n_dofs_owned;
    if (n_cols) *n_cols = A->space->n_dofs_local;
    if (nnz) *nnz = A->space->nnz_nodes * A->bs * A->bs;
    return FS_OK;
}

extern "C" int fs_matrix_zero(fs_matrix_t A) {
    FS_REQUIRE(A, "fs_matrix_zero: null matrix");
    FS_CHECK(A->val.zero(fs_rt().stream));
    FS_HIP(hipStreamSynchronize(fs_rt().stream));
    return FS_OK;
}

extern "C" int fs_matrix_axpy(fs_matrix_t Y, double a, fs_matrix_t X) {
    FS_REQUIRE(X && Y && X->space == Y->space, "fs_matrix_axpy: matrices must share a function space");
    hipLaunchKernelGGL(k_mat_axpy, dim3(fs_grid_for(Y->val.n)), dim3(FS_BLOCK), 0, fs_rt().stream, Y->val.p, X->val.p, Y->val.n, a);
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(fs_rt().stream));
    return FS_OK;
}

extern "C" int fs_matrix_copy(fs_matrix_t dst, fs_matrix_t src) {
    FS_REQUIRE(dst && src && dst->space == src->space && dst->val.n == src->val.n, "fs_matrix_copy: matrices must share a function space");
    FS_HIP(hipMemcpyAsync(dst->val.p, src->val.p, (size_t)src->val.n * sizeof(double), hipMemcpyDeviceToDevice, fs_rt().stream));
    return FS_OK;
}

extern "C" int fs_matrix_destroy(fs_matrix_t A) {
    delete A;
    return FS_OK;
}

extern "C" int fs_matrix_get_csr(fs_matrix_t A, int32_t* rowptr, int32_t* colidx, double* vals) {
    FS_REQUIRE(A, "fs_matrix_get_csr: null matrix");
    fs_space_s* sp = A->space;
    hipStream_t s = fs_rt().stream;
    const int bs = A->bs;
    const int64_t n_rows = sp->n_nodes_owned, nnz = sp->nnz_nodes * bs * bs;
    FS_REQUIRE(nnz < (int64_t)INT32_MAX, "fs_matrix_get_csr: nnz exceeds int32");
    dbuf<int32_t> d_rp, d_ci;
    dbuf<double> d_v;
    if (rowptr) FS_CHECK(d_rp.alloc(n_rows * bs + 1));
    if (colidx) FS_CHECK(d_ci.alloc(nnz));
    if (vals) FS_CHECK(d_v.alloc(nnz));
    if (bs == 1)
        hipLaunchKernelGGL(k_export_csr<1>, dim3(fs_grid_for(n_rows)), dim3(FS_BLOCK), 0, s, n_rows, sp->slice_ptr.p, sp->rowptr.p, sp->sell_col.p, A->val.p, sp->sell_entries, d_rp.p, d_ci.p, d_v.p);
    else if (bs == 2)
        hipLaunchKernelGGL(k_export_csr<2>, dim3(fs_grid_for(n_rows)), dim3(FS_BLOCK), 0, s, n_rows, sp->slice_ptr.p, sp->rowptr.p, sp->sell_col.p, A->val.p, sp->sell_entries, d_rp.p, d_ci.p, d_v.p);
    else if (bs == 3)
        hipLaunchKernelGGL(k_export_csr<3>, dim3(fs_grid_for(n_rows)), dim3(FS_BLOCK), 0, s, n_rows, sp->slice_ptr.p, sp->rowptr.p, sp->sell_col.p, A->val.p, sp->sell_entries, d_rp.p, d_ci.p, d_v.p);
    else
        hipLaunchKernelGGL(k_export_csr<4>, dim3(fs_grid_for(n_rows)), dim3(FS_BLOCK), 0, s, n_rows, sp->slice_ptr.p, sp->rowptr.p, sp->sell_col.p, A->val.p, sp->sell_entries, d_rp.p, d_ci.p, d_v.p);
    FS_KERNEL_CHECK();
    if (rowptr) FS_CHECK(d_rp.download(rowptr, n_rows * bs + 1, s));
    if (colidx) FS_CHECK(d_ci.download(colidx, nnz, s));
    if (vals) FS_CHECK(d_v.download(vals, nnz, s));
    return FS_OK;
}

extern "C" int fs_assemble_matrix(fs_matrix_t A, const fs_bilinear_form* form, int add) {
    FS_REQUIRE(A && form, "fs_assemble_matrix: null pointer");
    fs_space_s* sp = A->space;
    fs_mesh_s* m = sp->mesh;
    hipStream_t s = fs_rt().stream;
    dbuf<double> kstore, mstore;
    coef_dev kc, mc;
    FS_CHECK(make_coef(form->mass, m->nc, mstore, &mc, "fs_assemble_matrix(mass)"));
    FS_REQUIRE(mc.mode == FS_COEF_NONE || mc.mode == FS_COEF_CONST || mc.mode == FS_COEF_CELL,
               "fs_assemble_matrix: mass coefficient must be constant or per cell");
    const int grid = fs_grid_for(m->nc, FS_BLOCK, 8192);
    if (m->tdim == 2 && A->bs == 2) {
        FS_REQUIRE(sp->slots.p, "fs_assemble_matrix: 2-vector space without slot table");
        FS_REQUIRE(form->stiffness.mode == FS_COEF_NONE && form->advection.mode == FS_COEF_NONE && !(form->supg_pe > 0.0),
                   "fs_assemble_matrix: a 2-vector space takes the Lame parameters (plane-strain elasticity) and a mass coefficient");
        if (!sp->gmap_ptr.p) FS_CHECK(fs_space_build_gather_map(sp, s));
        const int gg = fs_grid_for(sp->sell_entries, FS_BLOCK, 1 << 16);
        if (sp->degree == 2) {
            if (add)
                hipLaunchKernelGGL(k_assemble_p2tri_elasticity_gather<true>, dim3(gg), dim3(FS_BLOCK), 0, s, sp->sell_entries, sp->gmap_ptr.p, sp->gmap_src.p, m->cells.p, m->xyz.p, form->lame_mu, form->lame_lambda, mc, sp->sell_entries, A->val.p);
            else
                hipLaunchKernelGGL(k_assemble_p2tri_elasticity_gather<false>, dim3(gg), dim3(FS_BLOCK), 0, s, sp->sell_entries, sp->gmap_ptr.p, sp->gmap_src.p, m->cells.p, m->xyz.p, form->lame_mu, form->lame_lambda, mc, sp->sell_entries, A->val.p);
        } else if (add)
            hipLaunchKernelGGL(k_assemble_tri_elasticity_gather<true>, dim3(gg), dim3(FS_BLOCK), 0, s, sp->sell_entries, sp->gmap_ptr.p, sp->gmap_src.p, m->cells.p, m->xyz.p, form->lame_mu, form->lame_lambda, mc, sp->sell_entries, A->val.p);
        else
            hipLaunchKernelGGL(k_assemble_tri_elasticity_gather<false>, dim3(gg), dim3(FS_BLOCK), 0, s, sp->sell_entries, sp->gmap_ptr.p, sp->gmap_src.p, m->cells.p, m->xyz.p, form->lame_mu, form->lame_lambda, mc, sp->sell_entries, A->val.p);
    } else if (m->tdim == 2 && sp->degree == 2) {
        FS_REQUIRE(A->bs == 1 && sp->inc_cell.p, "fs_assemble_matrix: CG2 space on triangles without assembly tables");
        dbuf<double> astore4;
        coef_dev ac4;
        FS_CHECK(make_coef(form->advection, 3 * m->nc, astore4, &ac4, "fs_assemble_matrix(advection)"));
        FS_REQUIRE(ac4.mode == FS_COEF_NONE || ac4.mode == FS_COEF_CONST || ac4.mode == FS_COEF_CELL,
                   "fs_assemble_matrix: CG2 advection (and its SUPG test function) takes a constant or per-cell velocity");
        FS_CHECK(make_coef(form->stiffness, m->nc, kstore, &kc, "fs_assemble_matrix(stiffness)"));
        FS_REQUIRE(kc.mode == FS_COEF_NONE || kc.mode == FS_COEF_CONST || kc.mode == FS_COEF_CELL || kc.mode == FS_COEF_CELL_QP,
                   "fs_assemble_matrix: CG2 stiffness coefficient must be constant, per cell or per quadrature point");
        const int bd = (int64_t)sp->max_row * FS_BLOCK * 8 <= 64 * 1024 ? FS_BLOCK : 64;
        const size_t lds = (size_t)sp->max_row * bd * sizeof(double);
        FS_REQUIRE(lds <= 64 * 1024, "fs_assemble_matrix: rows of %d entries exceed the LDS accumulator", sp->max_row);
        const int wpb = bd / 64;
        const int g = (fs_grid_for((sp->n_slices + wpb - 1) / wpb, 1, 8192) + 7) & ~7;
        if (add)
            hipLaunchKernelGGL(k_assemble_p2tri_scalar_gather<true>, dim3(g), dim3(bd), lds, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->inc_slice_ptr.p, sp->inc_entries, sp->inc_cell.p, sp->inc_pos.p, m->cells.p, m->xyz.p, kc, mc, A->val.p, ac4, form->advection_scale, form->supg_pe);
        else
            hipLaunchKernelGGL(k_assemble_p2tri_scalar_gather<false>, dim3(g), dim3(bd), lds, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->inc_slice_ptr.p, sp->inc_entries, sp->inc_cell.p, sp->inc_pos.p, m->cells.p, m->xyz.p, kc, mc, A->val.p, ac4, form->advection_scale, form->supg_pe);
    } else if (m->tdim == 2) {
        FS_REQUIRE(A->bs == 1 && sp->inc_cell.p, "fs_assemble_matrix: triangular meshes carry scalar CG1 spaces");
        FS_CHECK(make_coef(form->stiffness, m->nc, kstore, &kc, "fs_assemble_matrix(stiffness)"));
        FS_REQUIRE(kc.mode != FS_COEF_NODAL, "fs_assemble_matrix: nodal stiffness coefficient is not supported");
        dbuf<double> astore2;
        coef_dev ac2;
        FS_CHECK(make_coef(form->advection, (form->advection.mode == FS_COEF_CELL_ROW ? 9 : 3) * m->nc, astore2, &ac2, "fs_assemble_matrix(advection)"));
        FS_REQUIRE(ac2.mode == FS_COEF_NONE || ac2.mode == FS_COEF_CONST || ac2.mode == FS_COEF_CELL || ac2.mode == FS_COEF_CELL_ROW,
                   "fs_assemble_matrix: advection velocity must be constant, per cell or per (cell, test function)");
        FS_REQUIRE(!(ac2.mode == FS_COEF_CELL_ROW && form->supg_pe > 0.0), "fs_assemble_matrix: SUPG takes a constant or per-cell velocity");
        const int bd = (int64_t)sp->max_row * FS_BLOCK * 8 <= 64 * 1024 ? FS_BLOCK : 64;
        const size_t lds = (size_t)sp->max_row * bd * sizeof(double);
        FS_REQUIRE(lds <= 64 * 1024, "fs_assemble_matrix: rows of %d entries exceed the LDS accumulator", sp->max_row);
        const int wpb = bd / 64;
        const int g = (fs_grid_for((sp->n_slices + wpb - 1) / wpb, 1, 8192) + 7) & ~7;
        if (add)
            hipLaunchKernelGGL(k_assemble_tri_scalar_gather<true>, dim3(g), dim3(bd), lds, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->inc_slice_ptr.p, sp->inc_cell.p, sp->inc_pos.p, m->cells.p, m->xyz.p, kc, mc, ac2, form->advection_scale, A->val.p, form->supg_pe);
        else
            hipLaunchKernelGGL(k_assemble_tri_scalar_gather<false>, dim3(g), dim3(bd), lds, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->inc_slice_ptr.p, sp->inc_cell.p, sp->inc_pos.p, m->cells.p, m->xyz.p, kc, mc, ac2, form->advection_scale, A->val.p, form->supg_pe);
    } else if (A->bs == 1 && sp->degree == 2) {
        FS_REQUIRE(sp->inc_cell.p, "fs_assemble_matrix: CG2 space has no assembly tables");
        FS_CHECK(make_coef(form->stiffness, m->nc, kstore, &kc, "fs_assemble_matrix(stiffness)"));
        FS_REQUIRE(kc.mode == FS_COEF_NONE || kc.mode == FS_COEF_CONST || kc.mode == FS_COEF_CELL || kc.mode == FS_COEF_CELL_QP,
                   "fs_assemble_matrix: CG2 stiffness coefficient must be constant, per cell or per quadrature point");
        dbuf<double> astore3;
        coef_dev ac3;
        FS_CHECK(make_coef(form->advection, 3 * m->nc, astore3, &ac3, "fs_assemble_matrix(advection)"));
        FS_REQUIRE(ac3.mode == FS_COEF_NONE || ac3.mode == FS_COEF_CONST || ac3.mode == FS_COEF_CELL,
                   "fs_assemble_matrix: CG2 advection (and its SUPG test function) takes a constant or per-cell velocity");
        const int bd = (int64_t)sp->max_row * FS_BLOCK * 8 <= 64 * 1024 ? FS_BLOCK : 64;
        const size_t lds = (size_t)sp->max_row * bd * sizeof(double);
        FS_REQUIRE(lds <= 64 * 1024, "fs_assemble_matrix: rows of %d entries exceed the LDS accumulator", sp->max_row);
        const int wpb = bd / 64;
        const int g = (fs_grid_for((sp->n_slices + wpb - 1) / wpb, 1, 8192) + 7) & ~7;
        static const bool box_env_off2 = getenv("FS_BOX_ASSEMBLY") && getenv("FS_BOX_ASSEMBLY")[0] == '0';
        const bool box_fast_off2 = g_box_assembly < 0 ? box_env_off2 : g_box_assembly == 0;
        const box_snap bxs2 = make_box_snap(m);
        if (!box_fast_off2 && bxs2.h[0] > 0.0 && m->nc >= 6 && ac3.mode == FS_COEF_NONE &&
            (kc.mode == FS_COEF_NONE || kc.mode == FS_COEF_CONST || kc.mode == FS_COEF_CELL) &&
            (mc.mode == FS_COEF_NONE || mc.mode == FS_COEF_CONST || mc.mode == FS_COEF_CELL) && lds + 660 * sizeof(double) <= 64 * 1024) {
            // a mesh made by fs_mesh_create_box: the geometry-free form (k_assemble_p2_box_gather)
            if (!m->box_ref2.p) {
                FS_CHECK(m->box_ref2.alloc(660));
                hipLaunchKernelGGL(k_box_ref_rows_p2, dim3(1), dim3(64), 0, s, m->cells.p, m->xyz.p, bxs2, m->box_ref2.p);
            }
            const int accd = sp->max_row * bd;
            if (add)
                hipLaunchKernelGGL(k_assemble_p2_box_gather<true>, dim3(g), dim3(bd), lds + 660 * sizeof(double), s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p,
                                   sp->inc_slice_ptr.p, sp->inc_entries, sp->inc_cell.p, sp->inc_pos.p, m->box_ref2.p, kc, mc, A->val.p, sp->slice_order.p, accd);
            else
                hipLaunchKernelGGL(k_assemble_p2_box_gather<false>, dim3(g), dim3(bd), lds + 660 * sizeof(double), s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p,
                                   sp->inc_slice_ptr.p, sp->inc_entries, sp->inc_cell.p, sp->inc_pos.p, m->box_ref2.p, kc, mc, A->val.p, sp->slice_order.p, accd);
        } else if (ac3.mode != FS_COEF_NONE || kc.mode == FS_COEF_CELL_QP) {
            if (add)
                hipLaunchKernelGGL((k_assemble_p2_scalar_gather<true, true>), dim3(g), dim3(bd), lds, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->inc_slice_ptr.p, sp->inc_entries, sp->inc_cell.p, sp->inc_pos.p, m->cells.p, m->xyz.p, kc, mc, A->val.p, sp->slice_order.p, make_box_snap(m), ac3, form->advection_scale, form->supg_pe);
            else
                hipLaunchKernelGGL((k_assemble_p2_scalar_gather<false, true>), dim3(g), dim3(bd), lds, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->inc_slice_ptr.p, sp->inc_entries, sp->inc_cell.p, sp->inc_pos.p, m->cells.p, m->xyz.p, kc, mc, A->val.p, sp->slice_order.p, make_box_snap(m), ac3, form->advection_scale, form->supg_pe);
        } else if (add)
            hipLaunchKernelGGL(k_assemble_p2_scalar_gather<true>, dim3(g), dim3(bd), lds, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->inc_slice_ptr.p, sp->inc_entries, sp->inc_cell.p, sp->inc_pos.p, m->cells.p, m->xyz.p, kc, mc, A->val.p, sp->slice_order.p, make_box_snap(m));
        else
            hipLaunchKernelGGL(k_assemble_p2_scalar_gather<false>, dim3(g), dim3(bd), lds, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->inc_slice_ptr.p, sp->inc_entries, sp->inc_cell.p, sp->inc_pos.p, m->cells.p, m->xyz.p, kc, mc, A->val.p, sp->slice_order.p, make_box_snap(m));
    } else if (A->bs == 1 && sp->inc_cell.p) {
        // row-gather path: every SELL entry (padding included) is written exactly once, no memset
        FS_CHECK(make_coef(form->stiffness, m->nc, kstore, &kc, "fs_assemble_matrix(stiffness)"));
        FS_REQUIRE(kc.mode != FS_COEF_NODAL, "fs_assemble_matrix: nodal stiffness coefficient is not supported");
        dbuf<double> astore;
        coef_dev ac;
        FS_CHECK(make_coef(form->advection, (form->advection.mode == FS_COEF_CELL_ROW ? 12 : 3) * m->nc, astore, &ac, "fs_assemble_matrix(advection)"));
        FS_REQUIRE(ac.mode == FS_COEF_NONE || ac.mode == FS_COEF_CONST || ac.mode == FS_COEF_CELL || ac.mode == FS_COEF_CELL_ROW,
                   "fs_assemble_matrix: advection velocity must be constant, per cell or per (cell, test function)");
        FS_REQUIRE(!(ac.mode == FS_COEF_CELL_ROW && form->supg_pe > 0.0), "fs_assemble_matrix: SUPG takes a constant or per-cell velocity");
        // LDS: one accumulator column per thread; fall to one wave per workgroup for very long rows
        const int bd = (int64_t)sp->max_row * FS_BLOCK * 8 <= 64 * 1024 ? FS_BLOCK : 64;
        const size_t lds = (size_t)sp->max_row * bd * sizeof(double);
        FS_REQUIRE(lds <= 64 * 1024, "fs_assemble_matrix: rows of %d entries exceed the LDS accumulator", sp->max_row);
        const int wpb = bd / 64;
        const int g = (fs_grid_for((sp->n_slices + wpb - 1) / wpb, 1, 8192) + 7) & ~7;  // multiple of 8: XCD map
        // a mesh made by fs_mesh_create_box, snapped geometry, scalar coefficients, no advection: the geometry-free form
        static const bool box_env_off = getenv("FS_BOX_ASSEMBLY") && getenv("FS_BOX_ASSEMBLY")[0] == '0';
        const bool box_fast_off = g_box_assembly < 0 ? box_env_off : g_box_assembly == 0;
        const box_snap bxs = make_box_snap(m);
        if (!box_fast_off && bxs.h[0] > 0.0 && m->nc >= 6 && ac.mode == FS_COEF_NONE && bd == FS_BLOCK &&
            (kc.mode == FS_COEF_CONST || kc.mode == FS_COEF_CELL) &&
            (mc.mode == FS_COEF_NONE || mc.mode == FS_COEF_CONST || mc.mode == FS_COEF_CELL) && lds + 120 * sizeof(double) <= 64 * 1024) {
            if (!m->box_ref.p) {
                FS_CHECK(m->box_ref.alloc(120));
                hipLaunchKernelGGL(k_box_ref_rows, dim3(1), dim3(64), 0, s, m->cells.p, m->xyz.p, bxs, m->box_ref.p);
            }
            const int accd = sp->max_row * bd;
            if (add)
                hipLaunchKernelGGL(k_assemble_p1_box_gather<true>, dim3(g), dim3(bd), lds + 120 * sizeof(double), s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p,
                                   sp->inc_slice_ptr.p, sp->inc_cell.p, sp->inc_pos.p, m->box_ref.p, kc, mc, A->val.p, sp->slice_order.p, accd);
            else
                hipLaunchKernelGGL(k_assemble_p1_box_gather<false>, dim3(g), dim3(bd), lds + 120 * sizeof(double), s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p,
                                   sp->inc_slice_ptr.p, sp->inc_cell.p, sp->inc_pos.p, m->box_ref.p, kc, mc, A->val.p, sp->slice_order.p, accd);
        } else if (add)
            hipLaunchKernelGGL(k_assemble_p1_scalar_gather<1>, dim3(g), dim3(bd), lds, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->inc_slice_ptr.p, sp->inc_cell.p, sp->inc_pos.p, m->cells.p, m->xyz.p, kc, mc, ac, form->advection_scale, form->supg_pe, A->val.p, sp->slice_order.p, make_box_snap(m));
        else
            hipLaunchKernelGGL(k_assemble_p1_scalar_gather<0>, dim3(g), dim3(bd), lds, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->inc_slice_ptr.p, sp->inc_cell.p, sp->inc_pos.p, m->cells.p, m->xyz.p, kc, mc, ac, form->advection_scale, form->supg_pe, A->val.p, sp->slice_order.p, make_box_snap(m));
    } else if (A->bs == 1) {
        FS_REQUIRE(sp->slots.p, "fs_assemble_matrix: space has no assembly tables");
        FS_REQUIRE(form->advection.mode == FS_COEF_NONE, "fs_assemble_matrix: advection needs the row-gather tables");
        if (!add) FS_CHECK(A->val.zero(s));
        FS_CHECK(make_coef(form->stiffness, m->nc, kstore, &kc, "fs_assemble_matrix(stiffness)"));
        FS_REQUIRE(kc.mode != FS_COEF_NODAL && kc.mode != FS_COEF_CELL_TENSOR, "fs_assemble_matrix: nodal / per-cell tensor stiffness coefficients need the row-gather tables");
        hipLaunchKernelGGL(k_assemble_p1_scalar, dim3(grid), dim3(FS_BLOCK), 0, s, m->cells.p, m->xyz.p, sp->slots.p, m->nc, kc, mc, A->val.p);
    } else if (getenv("FS_ELASTICITY_ATOMIC") && sp->degree == 1 && A->bs == 3) {
        if (!add) FS_CHECK(A->val.zero(s));
        hipLaunchKernelGGL(k_assemble_p1_elasticity, dim3(grid), dim3(FS_BLOCK), 0, s, m->cells.p, m->xyz.p, sp->slots.p, m->nc, form->lame_mu, form->lame_lambda, mc, sp->sell_entries, A->val.p);
    } else if (A->bs == 3 && sp->degree == 2) {
        if (!sp->gmap_ptr.p) FS_CHECK(fs_space_build_gather_map(sp, s));
        const int gg = fs_grid_for(sp->sell_entries, FS_BLOCK, 1 << 16);
        if (add)
            hipLaunchKernelGGL(k_assemble_p2_elasticity_gather<true>, dim3(gg), dim3(FS_BLOCK), 0, s, sp->sell_entries, sp->gmap_ptr.p, sp->gmap_src.p, m->cells.p, m->xyz.p, form->lame_mu, form->lame_lambda, mc, sp->sell_entries, A->val.p);
        else
            hipLaunchKernelGGL(k_assemble_p2_elasticity_gather<false>, dim3(gg), dim3(FS_BLOCK), 0, s, sp->sell_entries, sp->gmap_ptr.p, sp->gmap_src.p, m->cells.p, m->xyz.p, form->lame_mu, form->lame_lambda, mc, sp->sell_entries, A->val.p);
    } else {
        FS_REQUIRE(A->bs == 3 && sp->degree == 1, "fs_assemble_matrix: no operator for block size %d on CG%d nodes (Taylor-Hood systems: fs_assemble_navier_stokes)", A->bs, sp->degree);
        if (!sp->gmap_ptr.p) FS_CHECK(fs_space_build_gather_map(sp, s));
        const int gg = fs_grid_for(sp->sell_entries, FS_BLOCK, 1 << 16);
        if (add)
            hipLaunchKernelGGL(k_assemble_p1_elasticity_gather<true>, dim3(gg), dim3(FS_BLOCK), 0, s, sp->sell_entries, sp->gmap_ptr.p, sp->gmap_src.p, m->cells.p, m->xyz.p, form->lame_mu, form->lame_lambda, mc, sp->sell_entries, A->val.p, make_box_snap(m));
        else
            hipLaunchKernelGGL(k_assemble_p1_elasticity_gather<false>, dim3(gg), dim3(FS_BLOCK), 0, s, sp->sell_entries, sp->gmap_ptr.p, sp->gmap_src.p, m->cells.p, m->xyz.p, form->lame_mu, form->lame_lambda, mc, sp->sell_entries, A->val.p, make_box_snap(m));
    }
    FS_KERNEL_CHECK();
    // (no wait here: host arrays were consumed by make_coef's uploads, the coefficient stores go back to the pool in stream order,
    // and whatever the caller enqueues next - the load vector, the Dirichlet rows - is prepared while the assembly runs)
    return FS_OK;
}

// scalar CG1 load vector without atomics: lane = row, its cell incidences summed in ascending cell order (the tables of the
// row-gather matrix assembly).  1 M DOF: 0.85 ms (constant f) / 1.25 ms (nodal f) with 23 M device-scope fp64 atomics.
template <bool ADD>
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p1_source_gather(int64_t n_rows, int64_t n_slices,
                                                                        const int64_t* __restrict__ inc_slice_ptr,
                                                                        const int32_t* __restrict__ inc_cell,
                                                                        const int32_t* __restrict__ cells,
                                                                        const double* __restrict__ xyz4, coef_dev f,
                                                                        double* __restrict__ b) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t row = s * FS_SLICE + lane;
        const int64_t ibase = inc_slice_ptr[s];
        const int iwidth = (int)((inc_slice_ptr[s + 1] - ibase) >> 6);
        double acc = 0.0;
        constexpr int PF = 4;          // incidence records, then cell records, ahead of the coordinate loads
        for (int j0 = 0; j0 < iwidth; j0 += PF) {
          int32_t qc[PF];
          int4 vc[PF];
#pragma unroll
          for (int u = 0; u < PF; ++u) qc[u] = j0 + u < iwidth ? inc_cell[ibase + (int64_t)(j0 + u) * FS_SLICE + lane] : -1;
#pragma unroll
          for (int u = 0; u < PF; ++u) vc[u] = qc[u] >= 0 ? reinterpret_cast<const int4*>(cells)[qc[u] >> 2] : make_int4(0, 0, 0, 0);
#pragma unroll
          for (int u = 0; u < PF; ++u) {
            if (j0 + u >= iwidth) break;
            const int32_t q = qc[u];
            if (q < 0) continue;
            const int c = q >> 2, a = q & 3;
            const int4 v4 = vc[u];
            const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
            const tet_geom t = tet_geometry(xyz4, v);
            if (f.mode == FS_COEF_NODAL) {          // b_e = M_e f_e with the exact P1 mass matrix
                const double fe[4] = {f.data[v[0]], f.data[v[1]], f.data[v[2]], f.data[v[3]]};
                const double fa = a == 0 ? fe[0] : a == 1 ? fe[1] : a == 2 ? fe[2] : fe[3];
                acc += t.adet * (1.0 / 120.0) * (((fe[0] + fe[1]) + (fe[2] + fe[3])) + fa);
            } else {
                acc += (f.mode == FS_COEF_CONST ? f.value : f.data[c]) * t.adet * (1.0 / 24.0);
            }
          }
        }
        if (row < n_rows) b[row] = ADD ? b[row] + acc : acc;
    }
}

// ---- von Mises stress, right-hand side of its L2 projection onto P1 (LinearElasticitySolver.py:71-76: project(von_Mises, V)
// = solve  int w q dx = int sqrt(3/2 s:s) q dx) ---------------------------------------------------------------------------
// b_a = int vm(x) lambda_a dx over the cells of vertex a (row gather on the scalar CG1 space of the mesh: lane = vertex, its
// incidences in ascending cell order, no atomics).  s = dev(sigma(u)), sigma = 2 mu sym(grad u) + lambda div u I.
// P1 displacement: grad u is constant per cell -> vm V / 4 exactly.  P2 displacement: grad u is linear, vm is not a
// polynomial; it is integrated with the 4-point degree-2 rule against lambda_a (the same rule everywhere else on CG2).
__device__ __forceinline__ double von_mises_of(const double (&G)[3][3], double mu, double lambda) {
    const double tr = G[0][0] + G[1][1] + G[2][2];
    double sg[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) sg[i][j] = mu * (G[i][j] + G[j][i]) + (i == j ? lambda * tr : 0.0);
    const double p = (sg[0][0] + sg[1][1] + sg[2][2]) * (1.0 / 3.0);
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double d = sg[i][j] - (i == j ? p : 0.0);
            ss += d * d;
        }
    return sqrt(1.5 * ss);
}
template <int DEG>
__global__ void __launch_bounds__(FS_BLOCK) k_von_mises_load(int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ inc_slice_ptr,
                                                             const int32_t* __restrict__ inc_cell, const int32_t* __restrict__ cells,
                                                             const double* __restrict__ xyz4, const int32_t* __restrict__ u_dofs,
                                                             const double* __restrict__ u, double mu, double lambda,
                                                             double* __restrict__ b) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t row = s * FS_SLICE + lane;
        const int64_t ibase = inc_slice_ptr[s];
        const int iwidth = (int)((inc_slice_ptr[s + 1] - ibase) >> 6);
        double acc = 0.0;
        for (int j = 0; j < iwidth; ++j) {
            const int32_t q = inc_cell[ibase + (int64_t)j * FS_SLICE + lane];
            if (q < 0) continue;
            const int c = q >> 2, a = q & 3;
            const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
            const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
            const tet_geom t = tet_geometry(xyz4, v);
            const double vol = t.adet * (1.0 / 6.0);
            if (DEG == 1) {
                double G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int64_t d = (int64_t)u_dofs[(int64_t)c * 4 + n] * 3;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int k = 0; k < 3; ++k) G[i][k] += u[d + i] * t.g[n][k];
                }
                acc += 0.25 * vol * von_mises_of(G, mu, lambda);
            } else {
                double un[10][3];
#pragma unroll
                for (int n = 0; n < 10; ++n) {
                    const int64_t d = (int64_t)u_dofs[(int64_t)c * 10 + n] * 3;
                    un[n][0] = u[d]; un[n][1] = u[d + 1]; un[n][2] = u[d + 2];
                }
#pragma unroll
                for (int qp = 0; qp < 4; ++qp) {
                    const double lam[4] = {FS_P2_QP[qp][0], FS_P2_QP[qp][1], FS_P2_QP[qp][2], FS_P2_QP[qp][3]};
                    double gp[10][3];
                    p2_basis_grads(t, lam, gp);
                    double G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
                    for (int n = 0; n < 10; ++n)
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int k = 0; k < 3; ++k) G[i][k] += un[n][i] * gp[n][k];
                    const double la = a == 0 ? lam[0] : a == 1 ? lam[1] : a == 2 ? lam[2] : lam[3];
                    acc += 0.25 * vol * la * von_mises_of(G, mu, lambda);
                }
            }
        }
        if (row < n_rows) b[row] = acc;
    }
}

// 2-D (plane strain, P1 on triangles): the reference's expression with dimension 2 - sigma the 2x2 tensor, the deviator
// s = sigma - tr(sigma)/3 Identity(2) (LinearElasticitySolver.py:71-73 keeps the 1/3) - constant per cell: vm A / 3.
__device__ __forceinline__ double von_mises_2d(const double (&G)[2][2], double mu, double lambda) {
    const double tr = G[0][0] + G[1][1];
    const double s00 = 2.0 * mu * G[0][0] + lambda * tr, s11 = 2.0 * mu * G[1][1] + lambda * tr;
    const double s01 = mu * (G[0][1] + G[1][0]);
    const double pm = (s00 + s11) * (1.0 / 3.0);
    return sqrt(1.5 * ((s00 - pm) * (s00 - pm) + (s11 - pm) * (s11 - pm) + 2.0 * s01 * s01));
}
// P2 displacement on triangles: grad u is linear, vm is not a polynomial; 3-point edge-midpoint rule against lambda_a
__global__ void __launch_bounds__(FS_BLOCK) k_von_mises_load_tri_p2(int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ inc_slice_ptr,
                                                                    const int32_t* __restrict__ inc_cell, const int32_t* __restrict__ cells,
                                                                    const double* __restrict__ xyz4, const int32_t* __restrict__ u_dofs,
                                                                    const double* __restrict__ u, double mu, double lambda,
                                                                    double* __restrict__ b) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t row = s * FS_SLICE + lane;
        const int64_t ibase = inc_slice_ptr[s];
        const int iwidth = (int)((inc_slice_ptr[s + 1] - ibase) >> 6);
        double acc = 0.0;
        for (int j = 0; j < iwidth; ++j) {
            const int32_t q = inc_cell[ibase + (int64_t)j * FS_SLICE + lane];
            if (q < 0) continue;
            const int c = q / 3, a = q - 3 * c;
            const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
            const tri_geom t = tri_geometry2(xyz4, v4.x, v4.y, v4.z);
            double un[6][2];
#pragma unroll
            for (int n = 0; n < 6; ++n) {
                const int64_t d = (int64_t)u_dofs[(int64_t)c * 6 + n] * 2;
                un[n][0] = u[d]; un[n][1] = u[d + 1];
            }
#pragma unroll
            for (int qp = 0; qp < 3; ++qp) {
                double G[2][2] = {{0, 0}, {0, 0}};
#pragma unroll
                for (int n = 0; n < 6; ++n) {
                    double gn[2];
                    p2tri_grad_one(t, qp, n, gn);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int k = 0; k < 2; ++k) G[i][k] += un[n][i] * gn[k];
                }
                const double la = a == qp ? 0.0 : 0.5;
                acc += t.area * (1.0 / 3.0) * la * von_mises_2d(G, mu, lambda);
            }
        }
        if (row < n_rows) b[row] = acc;
    }
}
__global__ void __launch_bounds__(FS_BLOCK) k_von_mises_load_tri(int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ inc_slice_ptr,
                                                                 const int32_t* __restrict__ inc_cell, const int32_t* __restrict__ cells,
                                                                 const double* __restrict__ xyz4, const double* __restrict__ u, double mu,
                                                                 double lambda, double* __restrict__ b) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t row = s * FS_SLICE + lane;
        const int64_t ibase = inc_slice_ptr[s];
        const int iwidth = (int)((inc_slice_ptr[s + 1] - ibase) >> 6);
        double acc = 0.0;
        for (int j = 0; j < iwidth; ++j) {
            const int32_t q = inc_cell[ibase + (int64_t)j * FS_SLICE + lane];
            if (q < 0) continue;
            const int c = q / 3;
            const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
            const int32_t v[3] = {v4.x, v4.y, v4.z};
            const tri_geom t = tri_geometry2(xyz4, v4.x, v4.y, v4.z);
            double G[2][2] = {{0, 0}, {0, 0}};
#pragma unroll
            for (int n = 0; n < 3; ++n)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int k = 0; k < 2; ++k) G[i][k] += u[2 * (int64_t)v[n] + i] * t.g[n][k];
            const double tr = G[0][0] + G[1][1];
            const double s00 = 2.0 * mu * G[0][0] + lambda * tr, s11 = 2.0 * mu * G[1][1] + lambda * tr;
            const double s01 = mu * (G[0][1] + G[1][0]);
            const double pm = (s00 + s11) * (1.0 / 3.0);
            const double ss = (s00 - pm) * (s00 - pm) + (s11 - pm) * (s11 - pm) + 2.0 * s01 * s01;
            acc += t.area * (1.0 / 3.0) * sqrt(1.5 * ss);
        }
        if (row < n_rows) b[row] = acc;
    }
}

// Right-hand sides of the L2 projection of the fluid stress  sigma = nu (grad u + grad u^T) - p I  onto CG1, component by
// component (CoupledNavierStokesSolver.py:149-155: project(..., TensorFunctionSpace(mesh, 'CG', 1))).  Taylor-Hood iterate:
// block (u_x, u_y, u_z, p) per CG2 node, p on the vertex nodes.  grad u is linear and p is linear: the integrand against
// lambda_a is quadratic, the 4-point rule exact.  b[row*9 + 3 i + j] = int sigma_ij lambda_row dx.
__global__ void __launch_bounds__(FS_BLOCK) k_viscous_stress_load(int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ inc_slice_ptr,
                                                                  const int32_t* __restrict__ inc_cell, const int32_t* __restrict__ cells,
                                                                  const double* __restrict__ xyz4, const int32_t* __restrict__ w_dofs,
                                                                  const double* __restrict__ w, double nu0, fs_visc_dev VL, double* __restrict__ b) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t row = s * FS_SLICE + lane;
        const int64_t ibase = inc_slice_ptr[s];
        const int iwidth = (int)((inc_slice_ptr[s + 1] - ibase) >> 6);
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < iwidth; ++j) {
            const int32_t q = inc_cell[ibase + (int64_t)j * FS_SLICE + lane];
            if (q < 0) continue;
            const int c = q >> 2, a = q & 3;
            const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
            const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
            const tet_geom t = tet_geometry(xyz4, v);
            double un[10][3], pv[4];
#pragma unroll
            for (int n = 0; n < 10; ++n) {
                const int64_t d = (int64_t)w_dofs[(int64_t)c * 10 + n] * 4;
                un[n][0] = w[d]; un[n][1] = w[d + 1]; un[n][2] = w[d + 2];
                if (n < 4) pv[n] = w[d + 3];
            }
            double tv[4] = {0.0, 0.0, 0.0, 0.0};
            if (VL.kind == 2)
#pragma unroll
                for (int n = 0; n < 4; ++n) tv[n] = VL.T[w_dofs[(int64_t)c * 10 + n]];
            // 4-point rule (exact for the quadratic integrand of a constant nu); nu(p, T) makes it quartic: the 14-point degree-5 rule
            const int nq = VL.kind == 2 ? 14 : 4;
            for (int qp = 0; qp < nq; ++qp) {
                double lam[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) lam[i] = VL.kind == 2 ? FS_TET14_QP[qp][i] : FS_P2_QP[qp][i];
                const double wq = t.adet * (1.0 / 6.0) * (VL.kind == 2 ? FS_TET14_QW[qp] : 0.25);
                double gp[10][3];
                p2_basis_grads(t, lam, gp);
                double G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
                for (int n = 0; n < 10; ++n)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int k = 0; k < 3; ++k) G[i][k] += un[n][i] * gp[n][k];
                const double pq = (lam[0] * pv[0] + lam[1] * pv[1]) + (lam[2] * pv[2] + lam[3] * pv[3]);
                const double nu = fs_viscosity(VL, nu0, pq, (lam[0] * tv[0] + lam[1] * tv[1]) + (lam[2] * tv[2] + lam[3] * tv[3]));     // CoupledNavierStokesSolver.viscosity
                const double la = wq * (a == 0 ? lam[0] : a == 1 ? lam[1] : a == 2 ? lam[2] : lam[3]);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc[3 * i + k] += la * (nu * (G[i][k] + G[k][i]) - (i == k ? pq : 0.0));
            }
        }
        if (row < n_rows)
#pragma unroll
            for (int k = 0; k < 9; ++k) b[row * 9 + k] = acc[k];
    }
}


// The same on TRIANGLES (2-D Taylor-Hood: block (u_x, u_y, -, p) per CG2 node, cell_dofs [nc][6]): b[row*4 + 2 i + k] = int sigma_ik lambda_row dx,
// edge-midpoint rule (the integrand is quadratic for a constant viscosity).
__global__ void __launch_bounds__(FS_BLOCK) k_viscous_stress_load_tri(int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ inc_slice_ptr,
                                                                      const int32_t* __restrict__ inc_cell, const int32_t* __restrict__ cells,
                                                                      const double* __restrict__ xyz4, const int32_t* __restrict__ w_dofs,
                                                                      const double* __restrict__ w, double nu0, fs_visc_dev VL,
                                                                      double* __restrict__ b) {
    const int lane = threadIdx.x & 63;
    int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; s < n_slices; s += stride) {
        const int64_t row = s * FS_SLICE + lane;
        const int64_t ibase = inc_slice_ptr[s];
        const int iwidth = (int)((inc_slice_ptr[s + 1] - ibase) >> 6);
        double acc[4] = {0, 0, 0, 0};
        for (int j = 0; j < iwidth; ++j) {
            const int32_t q = inc_cell[ibase + (int64_t)j * FS_SLICE + lane];
            if (q < 0) continue;
            const int c = q / 3, a = q - 3 * c;
            const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
            const tri_geom t = tri_geometry2(xyz4, v4.x, v4.y, v4.z);
            double un[6][2], pv[3];
#pragma unroll
            for (int n = 0; n < 6; ++n) {
                const int64_t d = (int64_t)w_dofs[(int64_t)c * 6 + n] * 4;
                un[n][0] = w[d]; un[n][1] = w[d + 1];
                if (n < 3) pv[n] = w[d + 3];
            }
            double tv[3] = {0.0, 0.0, 0.0};
            if (VL.kind == 2)
#pragma unroll
                for (int n = 0; n < 3; ++n) tv[n] = VL.T[w_dofs[(int64_t)c * 6 + n]];
            if (VL.kind == 2) {      // nu(p, T): quartic integrand, the 6-point degree-4 rule
                const double TQ[6][3] = {{0.108103018168070, 0.445948490915965, 0.445948490915965}, {0.445948490915965, 0.108103018168070, 0.445948490915965},
                                         {0.445948490915965, 0.445948490915965, 0.108103018168070}, {0.816847572980459, 0.091576213509771, 0.091576213509771},
                                         {0.091576213509771, 0.816847572980459, 0.091576213509771}, {0.091576213509771, 0.091576213509771, 0.816847572980459}};
                const double TW[6] = {0.223381589678011, 0.223381589678011, 0.223381589678011, 0.109951743655322, 0.109951743655322, 0.109951743655322};
                for (int qp = 0; qp < 6; ++qp) {
                    const double lam[3] = {TQ[qp][0], TQ[qp][1], TQ[qp][2]};
                    double gp[6][2];
                    p2tri_basis_grads(t, lam, gp);
                    double G[2][2] = {{0, 0}, {0, 0}};
#pragma unroll
                    for (int n = 0; n < 6; ++n)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int k = 0; k < 2; ++k) G[i][k] += un[n][i] * gp[n][k];
                    const double pq = lam[0] * pv[0] + lam[1] * pv[1] + lam[2] * pv[2];
                    const double nu = fs_viscosity(VL, nu0, pq, lam[0] * tv[0] + lam[1] * tv[1] + lam[2] * tv[2]);
                    const double la = t.area * TW[qp] * (a == 0 ? lam[0] : a == 1 ? lam[1] : lam[2]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int k = 0; k < 2; ++k) acc[2 * i + k] += la * (nu * (G[i][k] + G[k][i]) - (i == k ? pq : 0.0));
                }
                continue;
            }
#pragma unroll
            for (int qp = 0; qp < 3; ++qp) {         // mid-point of the edge opposite to vertex qp: lambda_qp = 0, the others 1/2
                double G[2][2] = {{0, 0}, {0, 0}};
#pragma unroll
                for (int n = 0; n < 6; ++n) {
                    double gn[2];
                    p2tri_grad_one(t, qp, n, gn);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int k = 0; k < 2; ++k) G[i][k] += un[n][i] * gn[k];
                }
                const double pq = 0.5 * ((qp == 0 ? 0.0 : pv[0]) + (qp == 1 ? 0.0 : pv[1]) + (qp == 2 ? 0.0 : pv[2]));
                const double tq = 0.5 * ((qp == 0 ? 0.0 : tv[0]) + (qp == 1 ? 0.0 : tv[1]) + (qp == 2 ? 0.0 : tv[2]));
                const double nu = fs_viscosity(VL, nu0, pq, tq);
                const double la = t.area * (1.0 / 3.0) * (a == qp ? 0.0 : 0.5);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int k = 0; k < 2; ++k) acc[2 * i + k] += la * (nu * (G[i][k] + G[k][i]) - (i == k ? pq : 0.0));
            }
        }
        if (row < n_rows)
#pragma unroll
            for (int k = 0; k < 4; ++k) b[row * 4 + k] = acc[k];
    }
}

// Matrix-free product y = K(form) x on a scalar CG1 space over tetrahedra: the walk of the row-gather assembly with every
// local row multiplied into x on the fly (north_star's "matrix-free CG").  No Dirichlet rows: callers mask.
// Measured against the assembled hybrid SELL/DIA product in DESIGN.md section 3 (it re-reads connectivity and
// coordinates and recomputes 24 cell geometries per row for every product, the assembled form streams 15 doubles per row).
extern "C" int fs_operator_apply(fs_space_t V, const fs_bilinear_form* form, fs_vector_t x, fs_vector_t y, int reps,
                                 double* ms_per_launch) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(V && form && x && y, "fs_operator_apply: null pointer");
    fs_space_s* sp = V;
    fs_mesh_s* m = sp->mesh;
    FS_REQUIRE(m->tdim == 3 && (sp->degree == 1 || sp->degree == 2) && sp->ncomp == 1 && sp->inc_cell.p,
               "fs_operator_apply: built for scalar CG1 / CG2 spaces on tetrahedra");
    FS_REQUIRE(x->d.n >= sp->n_dofs_local && y->d.n >= sp->n_dofs_owned && x != y, "fs_operator_apply: vector too short (or x is y)");
    hipStream_t s = fs_rt().stream;
    dbuf<double> kstore, mstore, astore;
    coef_dev kc, mc, ac;
    FS_CHECK(make_coef(form->mass, m->nc, mstore, &mc, "fs_operator_apply(mass)"));
    FS_CHECK(make_coef(form->stiffness, m->nc, kstore, &kc, "fs_operator_apply(stiffness)"));
    FS_CHECK(make_coef(form->advection, (form->advection.mode == FS_COEF_CELL_ROW ? 12 : 3) * m->nc, astore, &ac, "fs_operator_apply(advection)"));
    FS_REQUIRE((mc.mode == FS_COEF_NONE || mc.mode == FS_COEF_CONST || mc.mode == FS_COEF_CELL) && kc.mode != FS_COEF_NODAL &&
               (ac.mode == FS_COEF_NONE || ac.mode == FS_COEF_CONST || ac.mode == FS_COEF_CELL || ac.mode == FS_COEF_CELL_ROW) &&
               !(ac.mode == FS_COEF_CELL_ROW && form->supg_pe > 0.0),
               "fs_operator_apply: coefficients must be constant or per cell");
    const int wpb = FS_BLOCK / 64;
    const int g = (fs_grid_for((sp->n_slices + wpb - 1) / wpb, 1, 8192) + 7) & ~7;
    if (sp->degree == 2) {
        FS_REQUIRE(sp->cell_dofs && ac.mode == FS_COEF_NONE && (kc.mode == FS_COEF_NONE || kc.mode == FS_COEF_CONST || kc.mode == FS_COEF_CELL),
                   "fs_operator_apply: CG2 spaces take constant or per-cell scalar coefficients and no advection");
    }
    auto go = [&]() {
        if (sp->degree == 2)
            hipLaunchKernelGGL((k_assemble_p2_scalar_gather<false, false, true>), dim3(g), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p,
                               sp->inc_slice_ptr.p, sp->inc_entries, sp->inc_cell.p, sp->inc_pos.p, m->cells.p, m->xyz.p, kc, mc, y->d.p, sp->slice_order.p,
                               make_box_snap(m), coef_dev(), 0.0, 0.0, sp->cell_dofs, x->d.p);
        else
            hipLaunchKernelGGL(k_assemble_p1_scalar_gather<2>, dim3(g), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p,
                               sp->inc_slice_ptr.p, sp->inc_cell.p, sp->inc_pos.p, m->cells.p, m->xyz.p, kc, mc, ac, form->advection_scale,
                               form->supg_pe, y->d.p, sp->slice_order.p, make_box_snap(m), x->d.p);
    };
    go();
    if (reps > 1 && ms_per_launch) {      // HIP events around reps further launches on the library's stream
        hipEvent_t e0, e1;
        FS_HIP(hipEventCreate(&e0));
        FS_HIP(hipEventCreate(&e1));
        FS_HIP(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) go();
        FS_HIP(hipEventRecord(e1, s));
        FS_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        FS_HIP(hipEventElapsedTime(&ms, e0, e1));
        *ms_per_launch = (double)ms / reps;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

extern "C" int fs_assemble_viscous_stress_nn(fs_space_t th_space, fs_vector_t w, double nu, fs_space_t p1_space, fs_vector_t b,
                                             double nn_pref, double nn_exp);
extern "C" int fs_assemble_viscous_stress(fs_space_t th_space, fs_vector_t w, double nu, fs_space_t p1_space, fs_vector_t b) {
    return fs_assemble_viscous_stress_nn(th_space, w, nu, p1_space, b, 0.0, 0.0);
}
extern "C" int fs_assemble_viscous_stress_nn(fs_space_t th_space, fs_vector_t w, double nu, fs_space_t p1_space, fs_vector_t b,
                                             double nn_pref, double nn_exp) {
    FS_REQUIRE(th_space && w && p1_space && b && nn_pref >= 0.0, "fs_assemble_viscous_stress: null pointer / negative reference pressure");
    FS_REQUIRE(th_space->mesh == p1_space->mesh, "fs_assemble_viscous_stress: the two spaces live on different meshes");
    FS_REQUIRE(th_space->ncomp == 4 && th_space->degree == 2, "fs_assemble_viscous_stress: needs the Taylor-Hood node-block space");
    FS_REQUIRE(p1_space->ncomp == 1 && p1_space->degree == 1 && p1_space->inc_cell.p, "fs_assemble_viscous_stress: the target is the scalar CG1 space of the mesh");
    fs_mesh_s* m = p1_space->mesh;
    const int nt = m->tdim * m->tdim;           // tensor components per vertex: 9 (tetrahedra) or 4 (triangles)
    FS_REQUIRE(w->d.n >= th_space->n_dofs_local && b->d.n >= nt * p1_space->n_dofs_owned, "fs_assemble_viscous_stress: vector too short");
    hipStream_t s = fs_rt().stream;
    const fs_visc_dev VL = fs_space_viscosity(th_space, nn_pref, nn_exp);
    if (m->tdim == 2)
        hipLaunchKernelGGL(k_viscous_stress_load_tri, dim3(fs_grid_for(p1_space->n_slices * 64, FS_BLOCK, 8192)), dim3(FS_BLOCK), 0, s, p1_space->n_nodes_owned,
                           p1_space->n_slices, p1_space->inc_slice_ptr.p, p1_space->inc_cell.p, m->cells.p, m->xyz.p, th_space->cell_dofs, w->d.p, nu, VL, b->d.p);
    else
    hipLaunchKernelGGL(k_viscous_stress_load, dim3(fs_grid_for(p1_space->n_slices * 64, FS_BLOCK, 8192)), dim3(FS_BLOCK), 0, s, p1_space->n_nodes_owned,
                       p1_space->n_slices, p1_space->inc_slice_ptr.p, p1_space->inc_cell.p, m->cells.p, m->xyz.p, th_space->cell_dofs, w->d.p, nu, VL, b->d.p);
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

extern "C" int fs_assemble_von_mises(fs_space_t disp_space, fs_vector_t u, double mu, double lambda, fs_space_t p1_space,
                                     fs_vector_t b) {
    FS_REQUIRE(disp_space && u && p1_space && b, "fs_assemble_von_mises: null pointer");
    FS_REQUIRE(disp_space->mesh == p1_space->mesh, "fs_assemble_von_mises: the two spaces live on different meshes");
    FS_REQUIRE((disp_space->ncomp == 3 && disp_space->mesh->tdim == 3) || (disp_space->ncomp == 2 && disp_space->mesh->tdim == 2),
               "fs_assemble_von_mises: needs a 3-vector displacement space on tetrahedra or a 2-vector space on triangles");
    FS_REQUIRE(p1_space->ncomp == 1 && p1_space->degree == 1 && p1_space->inc_cell.p, "fs_assemble_von_mises: the target is the scalar CG1 space of the mesh");
    FS_REQUIRE(u->d.n >= disp_space->n_dofs_local && b->d.n >= p1_space->n_dofs_owned, "fs_assemble_von_mises: vector too short");
    hipStream_t s = fs_rt().stream;
    fs_mesh_s* m = p1_space->mesh;
    const int g = fs_grid_for(p1_space->n_slices * 64, FS_BLOCK, 8192);
    if (m->tdim == 2 && disp_space->degree == 2)
        hipLaunchKernelGGL(k_von_mises_load_tri_p2, dim3(g), dim3(FS_BLOCK), 0, s, p1_space->n_nodes_owned, p1_space->n_slices, p1_space->inc_slice_ptr.p,
                           p1_space->inc_cell.p, m->cells.p, m->xyz.p, disp_space->cell_dofs, u->d.p, mu, lambda, b->d.p);
    else if (m->tdim == 2)
        hipLaunchKernelGGL(k_von_mises_load_tri, dim3(g), dim3(FS_BLOCK), 0, s, p1_space->n_nodes_owned, p1_space->n_slices, p1_space->inc_slice_ptr.p,
                           p1_space->inc_cell.p, m->cells.p, m->xyz.p, u->d.p, mu, lambda, b->d.p);
    else if (disp_space->degree == 1)
        hipLaunchKernelGGL(k_von_mises_load<1>, dim3(g), dim3(FS_BLOCK), 0, s, p1_space->n_nodes_owned, p1_space->n_slices, p1_space->inc_slice_ptr.p,
                           p1_space->inc_cell.p, m->cells.p, m->xyz.p, disp_space->cell_dofs, u->d.p, mu, lambda, b->d.p);
    else
        hipLaunchKernelGGL(k_von_mises_load<2>, dim3(g), dim3(FS_BLOCK), 0, s, p1_space->n_nodes_owned, p1_space->n_slices, p1_space->inc_slice_ptr.p,
                           p1_space->inc_cell.p, m->cells.p, m->xyz.p, disp_space->cell_dofs, u->d.p, mu, lambda, b->d.p);
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

// vector CG1 load vector without atomics: thread per owned node; the cells holding the node are the sources of its
// diagonal block in the inverse slot table (ascending cell order: the right-hand side - and with it the iteration
// counts of the elasticity solves - is reproducible)
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_p1_vector_source_gather(int64_t n_rows, const int64_t* __restrict__ slice_ptr,
                                                                               const int32_t* __restrict__ sell_col,
                                                                               const int32_t* __restrict__ gptr,
                                                                               const int32_t* __restrict__ gsrc,
                                                                               const int32_t* __restrict__ cells,
                                                                               const double* __restrict__ xyz4, double fx, double fy,
                                                                               double fz, coef_dev dv, double* __restrict__ b) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_rows; r += stride) {
        const int64_t sp0 = slice_ptr[r >> 6];
        const int width = (int)((slice_ptr[(r >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (r & 63);
        int64_t e = -1;
        for (int k = 0; k < width; ++k)
            if (sell_col[base + (int64_t)k * FS_SLICE] == (int32_t)r) { e = base + (int64_t)k * FS_SLICE; break; }
        double acc[3] = {0.0, 0.0, 0.0};
        if (e >= 0) {
            constexpr int PF = 4;      // sources in groups: indices, then cell records, ahead of the coordinate loads
            const int32_t q1 = gptr[e + 1];
            for (int32_t q0 = gptr[e]; q0 < q1; q0 += PF) {
              int32_t sc[PF];
              int4 vc[PF];
#pragma unroll
              for (int u = 0; u < PF; ++u) sc[u] = q0 + u < q1 ? gsrc[q0 + u] : -1;
#pragma unroll
              for (int u = 0; u < PF; ++u) vc[u] = sc[u] >= 0 ? reinterpret_cast<const int4*>(cells)[sc[u] >> 4] : make_int4(0, 0, 0, 0);
#pragma unroll
              for (int u = 0; u < PF; ++u) {
                if (q0 + u >= q1) break;
                const int32_t sidx = sc[u];
                const int64_t c = sidx >> 4;
                const int a = (sidx >> 2) & 3;
                const int4 v4 = vc[u];
                const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
                const tet_geom t = tet_geometry(xyz4, v);
                const double w = t.adet * (1.0 / 24.0);
                double cd = 0.0;  // int c div v dx = c_cell * vol * grad_a[i]
                if (dv.mode == FS_COEF_CONST) cd = dv.value;
                else if (dv.mode == FS_COEF_CELL) cd = dv.data[c];
                else if (dv.mode == FS_COEF_NODAL) cd = 0.25 * ((dv.data[v[0]] + dv.data[v[1]]) + (dv.data[v[2]] + dv.data[v[3]]));
                cd *= t.adet * (1.0 / 6.0);
                const double ga[3] = {a == 0 ? t.g[0][0] : a == 1 ? t.g[1][0] : a == 2 ? t.g[2][0] : t.g[3][0],
                                      a == 0 ? t.g[0][1] : a == 1 ? t.g[1][1] : a == 2 ? t.g[2][1] : t.g[3][1],
                                      a == 0 ? t.g[0][2] : a == 1 ? t.g[1][2] : a == 2 ? t.g[2][2] : t.g[3][2]};
                acc[0] += w * fx + cd * ga[0];
                acc[1] += w * fy + cd * ga[1];
                acc[2] += w * fz + cd * ga[2];
              }
            }
        }
        b[3 * r + 0] += acc[0];
        b[3 * r + 1] += acc[1];
        b[3 * r + 2] += acc[2];
    }
}

extern "C" int fs_assemble_vector(fs_space_t space, const fs_linear_form* form, fs_vector_t b, int add) {
    FS_REQUIRE(space && form && b, "fs_assemble_vector: null pointer");
    FS_REQUIRE(b->d.n >= space->n_dofs_owned, "fs_assemble_vector: vector shorter than the owned dofs");
    fs_mesh_s* m = space->mesh;
    hipStream_t s = fs_rt().stream;
    if (!add) FS_CHECK(b->d.zero(s));
    dbuf<double> store;
    coef_dev f;
    const int64_t len = form->source.mode == FS_COEF_NODAL ? space->n_nodes_local : m->nc;
    FS_CHECK(make_coef(form->source, len, store, &f, "fs_assemble_vector(source)"));
    dbuf<double> dstore;
    coef_dev dv;
    const int64_t dlen = form->div_coef.mode == FS_COEF_NODAL ? space->n_nodes_local : m->nc;
    FS_CHECK(make_coef(form->div_coef, dlen, dstore, &dv, "fs_assemble_vector(div_coef)"));
    FS_REQUIRE(dv.mode == FS_COEF_NONE || space->ncomp == 3 || space->ncomp == 2, "fs_assemble_vector: div_coef needs a vector space");
    if (space->ncomp == 1 && f.mode == FS_COEF_NONE) {
        FS_HIP(hipStreamSynchronize(s));
        return FS_OK;
    }
    if (m->tdim == 2 && space->ncomp == 2) {
        FS_REQUIRE(f.mode == FS_COEF_NONE, "fs_assemble_vector: vector spaces take their body force in vector_value");
        FS_REQUIRE(dv.mode != FS_COEF_TENSOR && !(form->supg_pe > 0.0), "fs_assemble_vector: unsupported option on a 2-vector space");
        FS_REQUIRE(space->slots.p, "fs_assemble_vector: 2-vector space without slot table");
        if (!space->gmap_ptr.p) FS_CHECK(fs_space_build_gather_map(space, s));
        if (space->degree == 2)
            hipLaunchKernelGGL(k_assemble_p2tri_vector_source_gather, dim3(fs_grid_for(space->n_nodes_owned, FS_BLOCK, 8192)), dim3(FS_BLOCK), 0, s,
                               space->n_nodes_owned, space->slice_ptr.p, space->sell_col.p, space->gmap_ptr.p, space->gmap_src.p, m->cells.p,
                               m->xyz.p, form->vector_value[0], form->vector_value[1], dv, m->n_owned, space->n_edges_owned, b->d.p);
        else
        hipLaunchKernelGGL(k_assemble_tri_vector_source_gather, dim3(fs_grid_for(space->n_nodes_owned, FS_BLOCK, 8192)), dim3(FS_BLOCK), 0, s,
                           space->n_nodes_owned, space->slice_ptr.p, space->sell_col.p, space->gmap_ptr.p, space->gmap_src.p, m->cells.p,
                           m->xyz.p, form->vector_value[0], form->vector_value[1], dv, b->d.p);
        FS_KERNEL_CHECK();
        FS_HIP(hipStreamSynchronize(s));
        return FS_OK;
    }
    if (m->tdim == 2 && space->degree == 2) {
        FS_REQUIRE(f.mode != FS_COEF_TENSOR && space->inc_cell.p, "fs_assemble_vector: unsupported option on a CG2 space on triangles");
        dbuf<double> sstore4;
        coef_dev sv4;
        FS_CHECK(make_coef(form->supg_velocity, 3 * m->nc, sstore4, &sv4, "fs_assemble_vector(supg_velocity)"));
        FS_REQUIRE(!(form->supg_pe > 0.0) || sv4.mode == FS_COEF_NONE || sv4.mode == FS_COEF_CONST || sv4.mode == FS_COEF_CELL,
                   "fs_assemble_vector: the SUPG source term is built for constant / per-cell velocities");
        hipLaunchKernelGGL(k_assemble_p2tri_source_gather, dim3(fs_grid_for(space->n_slices * 64, FS_BLOCK, 8192)), dim3(FS_BLOCK), 0, s,
                           space->n_nodes_owned, space->n_slices, space->inc_slice_ptr.p, space->inc_cell.p, space->cell_dofs, m->cells.p,
                           m->xyz.p, f, b->d.p, sv4, form->supg_pe);
        FS_KERNEL_CHECK();
        FS_HIP(hipStreamSynchronize(s));
        return FS_OK;
    }
    if (m->tdim == 2) {
        FS_REQUIRE(f.mode != FS_COEF_TENSOR, "fs_assemble_vector: unsupported option on a triangular mesh");
        dbuf<double> sstore2;
        coef_dev sv2;
        FS_CHECK(make_coef(form->supg_velocity, 3 * m->nc, sstore2, &sv2, "fs_assemble_vector(supg_velocity)"));
        FS_REQUIRE(!(form->supg_pe > 0.0) || sv2.mode == FS_COEF_NONE || sv2.mode == FS_COEF_CONST || sv2.mode == FS_COEF_CELL,
                   "fs_assemble_vector: the SUPG source term is built for constant / per-cell velocities");
        hipLaunchKernelGGL(k_assemble_tri_source, dim3(fs_grid_for(m->nc, FS_BLOCK, 8192)), dim3(FS_BLOCK), 0, s, m->cells.p, m->xyz.p, m->nc, space->n_nodes_owned, f, b->d.p,
                           sv2, form->supg_pe);
        FS_KERNEL_CHECK();
        FS_HIP(hipStreamSynchronize(s));
        return FS_OK;
    }
    if (space->degree == 2 && space->ncomp == 3) {
        FS_REQUIRE(f.mode == FS_COEF_NONE, "fs_assemble_vector: vector spaces take their body force in vector_value");
        FS_REQUIRE(dv.mode != FS_COEF_TENSOR, "fs_assemble_vector: tensor coefficient is meaningless here");
        FS_REQUIRE(space->slots.p, "fs_assemble_vector: vector CG2 space without slot table");
        if (!space->gmap_ptr.p) FS_CHECK(fs_space_build_gather_map(space, s));
        hipLaunchKernelGGL(k_assemble_p2_vector_source_gather, dim3(fs_grid_for(space->n_nodes_owned, FS_BLOCK, 8192)), dim3(FS_BLOCK), 0, s,
                           space->n_nodes_owned, space->slice_ptr.p, space->sell_col.p, space->gmap_ptr.p, space->gmap_src.p, m->cells.p,
                           m->xyz.p, form->vector_value[0], form->vector_value[1], form->vector_value[2], dv, m->n_owned,
                           space->n_edges_owned, b->d.p);
        FS_KERNEL_CHECK();
        FS_HIP(hipStreamSynchronize(s));
        return FS_OK;
    }
    if (space->degree == 2) {
        FS_REQUIRE(space->ncomp == 1, "fs_assemble_vector: %d-component CG2 node blocks have no load-vector kernel", space->ncomp);
        FS_REQUIRE(f.mode != FS_COEF_TENSOR, "fs_assemble_vector: tensor coefficient is meaningless here");
        dbuf<double> sstore3;
        coef_dev sv3;
        FS_CHECK(make_coef(form->supg_velocity, 3 * m->nc, sstore3, &sv3, "fs_assemble_vector(supg_velocity)"));
        const bool supg3 = form->supg_pe > 0.0 && sv3.mode != FS_COEF_NONE;
        FS_REQUIRE(!supg3 || ((sv3.mode == FS_COEF_CONST || sv3.mode == FS_COEF_CELL) && space->inc_cell.p),
                   "fs_assemble_vector: the SUPG source term is built for constant / per-cell velocities");
        if (space->ncomp == 1 && space->inc_cell.p && (supg3 || !getenv("FS_SOURCE_ATOMIC")))
            hipLaunchKernelGGL(k_assemble_p2_source_gather, dim3(fs_grid_for(space->n_slices * 64, FS_BLOCK, 8192)), dim3(FS_BLOCK), 0, s,
                               space->n_nodes_owned, space->n_slices, space->inc_slice_ptr.p, space->inc_cell.p, space->cell_dofs, m->cells.p,
                               m->xyz.p, f, b->d.p, sv3, form->supg_pe);
        else
            hipLaunchKernelGGL(k_assemble_p2_source, dim3(fs_grid_for(m->nc, FS_BLOCK, 8192)), dim3(FS_BLOCK), 0, s, space->cell_dofs, m->cells.p, m->xyz.p, m->nc, space->n_nodes_owned, f, b->d.p);
        FS_KERNEL_CHECK();
        FS_HIP(hipStreamSynchronize(s));
        return FS_OK;
    }
    FS_REQUIRE(f.mode != FS_COEF_TENSOR && dv.mode != FS_COEF_TENSOR, "fs_assemble_vector: tensor coefficient is meaningless here");
    dbuf<double> sstore;
    coef_dev sv;
    FS_CHECK(make_coef(form->supg_velocity, 3 * m->nc, sstore, &sv, "fs_assemble_vector(supg_velocity)"));
    FS_REQUIRE(!(form->supg_pe > 0.0) || sv.mode == FS_COEF_NONE || space->ncomp == 1,
               "fs_assemble_vector: the SUPG source term is built for scalar spaces");
    if (space->ncomp == 1 && space->inc_cell.p && !(form->supg_pe > 0.0 && sv.mode != FS_COEF_NONE) && !getenv("FS_SOURCE_ATOMIC")) {
        // b was zeroed above unless add: the gather kernel adds to what is there either way
        hipLaunchKernelGGL(k_assemble_p1_source_gather<true>, dim3(fs_grid_for(space->n_slices * 64, FS_BLOCK, 8192)), dim3(FS_BLOCK), 0, s,
                           space->n_nodes_owned, space->n_slices, space->inc_slice_ptr.p, space->inc_cell.p, m->cells.p, m->xyz.p, f, b->d.p);
        FS_KERNEL_CHECK();
        FS_HIP(hipStreamSynchronize(s));
        return FS_OK;
    }
    if (space->ncomp == 3 && space->slots.p && !getenv("FS_SOURCE_ATOMIC")) {
        if (!space->gmap_ptr.p) FS_CHECK(fs_space_build_gather_map(space, s));
        hipLaunchKernelGGL(k_assemble_p1_vector_source_gather, dim3(fs_grid_for(space->n_nodes_owned, FS_BLOCK, 8192)), dim3(FS_BLOCK), 0, s,
                           space->n_nodes_owned, space->slice_ptr.p, space->sell_col.p, space->gmap_ptr.p, space->gmap_src.p, m->cells.p,
                           m->xyz.p, form->vector_value[0], form->vector_value[1], form->vector_value[2], dv, b->d.p);
        FS_KERNEL_CHECK();
        FS_HIP(hipStreamSynchronize(s));
        return FS_OK;
    }
    hipLaunchKernelGGL(k_assemble_p1_source, dim3(fs_grid_for(m->nc, FS_BLOCK, 8192)), dim3(FS_BLOCK), 0, s, m->cells.p, m->xyz.p, m->nc, m->n_owned, f, space->ncomp, form->vector_value[0], form->vector_value[1], form->vector_value[2], dv, sv, form->supg_pe, b->d.p);
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

// SUPG part of the ds(i) terms: thread per (facet, cell vertex a)
__global__ void k_facet_supg(int64_t nf, const int32_t* __restrict__ facet_cell, const int32_t* __restrict__ facet_opp,
                             const double* __restrict__ g, const double* __restrict__ h, coef_dev vel, double pe,
                             const int32_t* __restrict__ cells, const double* __restrict__ xyz4, int64_t n_rows,
                             const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ sell_col,
                             double* __restrict__ val, double* __restrict__ b, int* __restrict__ err) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nf * 4; t += stride) {
        const int64_t f = t >> 2;
        const int a = (int)(t & 3);
        const int64_t c = facet_cell[f];
        const int o = facet_opp[f];
        const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
        const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
        const int32_t row = v[a];
        if (row >= n_rows) continue;
        const tet_geom tg = tet_geometry(xyz4, v);
        double vx, vy, vz;
        if (vel.mode == FS_COEF_CONST) { vx = vel.tensor[0]; vy = vel.tensor[1]; vz = vel.tensor[2]; }
        else { vx = vel.data[3 * c]; vy = vel.data[3 * c + 1]; vz = vel.data[3 * c + 2]; }
        const double tau = supg_tau(xyz4, v, tg.adet, sqrt(vx * vx + vy * vy + vz * vz), pe);
        const double wa = tau * (vx * tg.g[a][0] + vy * tg.g[a][1] + vz * tg.g[a][2]);
        const double gn = sqrt(tg.g[o][0] * tg.g[o][0] + tg.g[o][1] * tg.g[o][1] + tg.g[o][2] * tg.g[o][2]);
        const double area = 0.5 * tg.adet * gn;          // 3 V |grad lambda_o|
        if (b && g) atomicAdd(&b[row], g[f] * area * wa);
        if (val && h) {
            const int64_t sp0 = slice_ptr[row >> 6];
            const int width = (int)((slice_ptr[(row >> 6) + 1] - sp0) >> 6);
            const int64_t base = sp0 + (row & 63);
            for (int bb = 0; bb < 4; ++bb) {
                if (bb == o) continue;
                const int k = fs_find_pos_local(sell_col, base, width, v[bb]);
                if (k >= 0) atomicAdd(&val[base + (int64_t)k * FS_SLICE], h[f] * area * (1.0 / 3.0) * wa);
                else atomicAdd(err, 1);
            }
        }
    }
}

// the same on the boundary EDGES of a triangular mesh: |E| = 2 A |grad lambda_o|, the Robin matrix couples w_a with the two edge vertices
__global__ void k_facet_supg_tri(int64_t nf, const int32_t* __restrict__ facet_cell, const int32_t* __restrict__ facet_opp,
                                 const double* __restrict__ g, const double* __restrict__ h, coef_dev vel, double pe,
                                 const int32_t* __restrict__ cells, const double* __restrict__ xyz4, int64_t n_rows,
                                 const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ sell_col,
                                 double* __restrict__ val, double* __restrict__ b, int* __restrict__ err) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nf * 3; t += stride) {
        const int64_t f = t / 3;
        const int a = (int)(t - 3 * f);
        const int64_t c = facet_cell[f];
        const int o = facet_opp[f];
        const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
        const int32_t v[3] = {v4.x, v4.y, v4.z};
        const int32_t row = v[a];
        if (row >= n_rows) continue;
        const tri_geom tg = tri_geometry2(xyz4, v[0], v[1], v[2]);
        double vx, vy;
        if (vel.mode == FS_COEF_CONST) { vx = vel.tensor[0]; vy = vel.tensor[1]; }
        else { vx = vel.data[3 * c]; vy = vel.data[3 * c + 1]; }
        const double tau = supg_tau_tri(xyz4, v[0], v[1], v[2], tg.area, sqrt(vx * vx + vy * vy), pe);
        const double wa = tau * (vx * tg.g[a][0] + vy * tg.g[a][1]);
        const double len = 2.0 * tg.area * sqrt(tg.g[o][0] * tg.g[o][0] + tg.g[o][1] * tg.g[o][1]);
        if (b && g) atomicAdd(&b[row], g[f] * len * wa);
        if (val && h) {
            const int64_t sp0 = slice_ptr[row >> 6];
            const int width = (int)((slice_ptr[(row >> 6) + 1] - sp0) >> 6);
            const int64_t base = sp0 + (row & 63);
            for (int bb = 0; bb < 3; ++bb) {
                if (bb == o) continue;
                const int k = fs_find_pos_local(sell_col, base, width, v[bb]);
                if (k >= 0) atomicAdd(&val[base + (int64_t)k * FS_SLICE], h[f] * len * 0.5 * wa);
                else atomicAdd(err, 1);
            }
        }
    }
}

// CG2 spaces: thread per (facet, cell dof a).  Load: g |F| tau (v . grad q_a) at the facet centroid (the gradient is linear);
// Robin matrix: h tau int_F phi_b (v . grad q_a) ds over the six facet dofs b, cubic on the facet: 6-point degree-4 rule
__global__ void k_facet_supg_p2(int64_t nf, const int32_t* __restrict__ facet_cell, const int32_t* __restrict__ facet_opp,
                                const double* __restrict__ g, const double* __restrict__ h, coef_dev vel, double pe,
                                const int32_t* __restrict__ cells, const int32_t* __restrict__ cell_dofs, const double* __restrict__ xyz4,
                                int64_t n_rows, const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ sell_col,
                                double* __restrict__ val, double* __restrict__ b, int* __restrict__ err) {
    const double TQ[6][3] = {{0.108103018168070, 0.445948490915965, 0.445948490915965}, {0.445948490915965, 0.108103018168070, 0.445948490915965},
                             {0.445948490915965, 0.445948490915965, 0.108103018168070}, {0.816847572980459, 0.091576213509771, 0.091576213509771},
                             {0.091576213509771, 0.816847572980459, 0.091576213509771}, {0.091576213509771, 0.091576213509771, 0.816847572980459}};
    const double TW[6] = {0.223381589678011, 0.223381589678011, 0.223381589678011, 0.109951743655322, 0.109951743655322, 0.109951743655322};
    const int ei[6] = {2, 1, 1, 0, 0, 0}, ej[6] = {3, 3, 2, 3, 2, 1};
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nf * 10; t += stride) {
        const int64_t f = t / 10;
        const int a = (int)(t - 10 * f);
        const int64_t c = facet_cell[f];
        const int o = facet_opp[f];
        const int32_t row = cell_dofs[c * 10 + a];
        if (row >= n_rows) continue;
        const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
        const int32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
        const tet_geom tg = tet_geometry(xyz4, v);
        double vx, vy, vz;
        if (vel.mode == FS_COEF_CONST) { vx = vel.tensor[0]; vy = vel.tensor[1]; vz = vel.tensor[2]; }
        else { vx = vel.data[3 * c]; vy = vel.data[3 * c + 1]; vz = vel.data[3 * c + 2]; }
        const double tau = supg_tau(xyz4, v, tg.adet, sqrt(vx * vx + vy * vy + vz * vz), pe);
        const double gn = sqrt(tg.g[o][0] * tg.g[o][0] + tg.g[o][1] * tg.g[o][1] + tg.g[o][2] * tg.g[o][2]);
        const double area = 0.5 * tg.adet * gn;          // 3 V |grad lambda_o|
        if (b && g) {
            double lam[4];
            for (int i = 0; i < 4; ++i) lam[i] = i == o ? 0.0 : 1.0 / 3.0;
            double gp[10][3];
            p2_basis_grads(tg, lam, gp);
            atomicAdd(&b[row], g[f] * area * tau * (vx * gp[a][0] + vy * gp[a][1] + vz * gp[a][2]));
        }
        if (val && h) {
            double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int qp = 0; qp < 6; ++qp) {
                double lam[4];
                for (int i = 0, k = 0; i < 4; ++i) lam[i] = i == o ? 0.0 : TQ[qp][k++];
                double gp[10][3];
                p2_basis_grads(tg, lam, gp);
                const double w = TW[qp] * area * tau * (vx * gp[a][0] + vy * gp[a][1] + vz * gp[a][2]);
                for (int bb = 0; bb < 4; ++bb) acc[bb] += w * lam[bb] * (2.0 * lam[bb] - 1.0);
                for (int e = 0; e < 6; ++e) acc[4 + e] += w * 4.0 * lam[ei[e]] * lam[ej[e]];
            }
            const int64_t sp0 = slice_ptr[row >> 6];
            const int width = (int)((slice_ptr[(row >> 6) + 1] - sp0) >> 6);
            const int64_t base = sp0 + (row & 63);
            for (int bb = 0; bb < 10; ++bb) {
                if (bb < 4 ? bb == o : (ei[bb - 4] == o || ej[bb - 4] == o)) continue;      // functions that vanish on the facet
                const int k = fs_find_pos_local(sell_col, base, width, cell_dofs[c * 10 + bb]);
                if (k >= 0) atomicAdd(&val[base + (int64_t)k * FS_SLICE], h[f] * acc[bb]);
                else atomicAdd(err, 1);
            }
        }
    }
}

// CG2 on triangles: thread per (boundary edge, cell dof a); 3-point Gauss rule along the edge
__global__ void k_facet_supg_p2tri(int64_t nf, const int32_t* __restrict__ facet_cell, const int32_t* __restrict__ facet_opp,
                                   const double* __restrict__ g, const double* __restrict__ h, coef_dev vel, double pe,
                                   const int32_t* __restrict__ cells, const int32_t* __restrict__ cell_dofs, const double* __restrict__ xyz4,
                                   int64_t n_rows, const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ sell_col,
                                   double* __restrict__ val, double* __restrict__ b, int* __restrict__ err) {
    const double GQ[3] = {0.5 - 0.5 * 0.7745966692414834, 0.5, 0.5 + 0.5 * 0.7745966692414834};
    const double GW[3] = {5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0};
    const int ei[3] = {1, 0, 0}, ej[3] = {2, 2, 1};
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nf * 6; t += stride) {
        const int64_t f = t / 6;
        const int a = (int)(t - 6 * f);
        const int64_t c = facet_cell[f];
        const int o = facet_opp[f];
        const int32_t row = cell_dofs[c * 6 + a];
        if (row >= n_rows) continue;
        const int4 v4 = reinterpret_cast<const int4*>(cells)[c];
        const tri_geom tg = tri_geometry2(xyz4, v4.x, v4.y, v4.z);
        double vx, vy;
        if (vel.mode == FS_COEF_CONST) { vx = vel.tensor[0]; vy = vel.tensor[1]; }
        else { vx = vel.data[3 * c]; vy = vel.data[3 * c + 1]; }
        const double tau = supg_tau_tri(xyz4, v4.x, v4.y, v4.z, tg.area, sqrt(vx * vx + vy * vy), pe);
        const double len = 2.0 * tg.area * sqrt(tg.g[o][0] * tg.g[o][0] + tg.g[o][1] * tg.g[o][1]);
        if (b && g) {
            double lam[3];
            for (int i = 0; i < 3; ++i) lam[i] = i == o ? 0.0 : 0.5;
            double gp[6][2];
            p2tri_basis_grads(tg, lam, gp);
            atomicAdd(&b[row], g[f] * len * tau * (vx * gp[a][0] + vy * gp[a][1]));
        }
        if (val && h) {
            double acc[6] = {0, 0, 0, 0, 0, 0};
            for (int qp = 0; qp < 3; ++qp) {
                double lam[3];
                for (int i = 0, k = 0; i < 3; ++i) lam[i] = i == o ? 0.0 : (k++ == 0 ? GQ[qp] : 1.0 - GQ[qp]);
                double gp[6][2];
                p2tri_basis_grads(tg, lam, gp);
                const double w = GW[qp] * len * tau * (vx * gp[a][0] + vy * gp[a][1]);
                for (int bb = 0; bb < 3; ++bb) acc[bb] += w * lam[bb] * (2.0 * lam[bb] - 1.0);
                for (int e = 0; e < 3; ++e) acc[3 + e] += w * 4.0 * lam[ei[e]] * lam[ej[e]];
            }
            const int64_t sp0 = slice_ptr[row >> 6];
            const int width = (int)((slice_ptr[(row >> 6) + 1] - sp0) >> 6);
            const int64_t base = sp0 + (row & 63);
            for (int bb = 0; bb < 6; ++bb) {
                if (bb < 3 ? bb == o : (ei[bb - 3] == o || ej[bb - 3] == o)) continue;
                const int k = fs_find_pos_local(sell_col, base, width, cell_dofs[c * 6 + bb]);
                if (k >= 0) atomicAdd(&val[base + (int64_t)k * FS_SLICE], h[f] * acc[bb]);
                else atomicAdd(err, 1);
            }
        }
    }
}

extern "C" int fs_assemble_facet_supg(fs_space_t space, fs_matrix_t A, fs_vector_t b, int64_t n_facets, const int32_t* facet_cell,
                                      const int32_t* facet_opposite, const double* g, const double* h, const fs_coef* velocity,
                                      double supg_pe) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(space && velocity && supg_pe > 0.0 && n_facets >= 0, "fs_assemble_facet_supg: bad arguments");
    FS_REQUIRE(space->ncomp == 1 && (space->degree == 1 || space->cell_dofs), "fs_assemble_facet_supg: scalar CG1 / CG2 spaces only");
    FS_REQUIRE((!A || A->space == space) && (!b || b->d.n >= space->n_dofs_owned), "fs_assemble_facet_supg: operand mismatch");
    if (n_facets == 0 || ((!A || !h) && (!b || !g))) return FS_OK;
    fs_mesh_s* m = space->mesh;
    for (int64_t i = 0; i < n_facets; ++i)
        FS_REQUIRE(facet_cell[i] >= 0 && facet_cell[i] < m->nc && facet_opposite[i] >= 0 && facet_opposite[i] <= m->tdim,
                   "fs_assemble_facet_supg: facet %lld names cell %d / local vertex %d", (long long)i, facet_cell[i], facet_opposite[i]);
    hipStream_t s = fs_rt().stream;
    dbuf<int32_t> dc, dop;
    dbuf<double> dg, dh, vstore;
    dbuf<int> d_err;
    coef_dev vel;
    FS_CHECK(make_coef(*velocity, 3 * m->nc, vstore, &vel, "fs_assemble_facet_supg(velocity)"));
    FS_REQUIRE(vel.mode == FS_COEF_CONST || vel.mode == FS_COEF_CELL, "fs_assemble_facet_supg: velocity must be constant or per cell");
    FS_CHECK(dc.alloc(n_facets)); FS_CHECK(dop.alloc(n_facets)); FS_CHECK(d_err.alloc(1)); FS_CHECK(d_err.zero(s));
    FS_CHECK(dc.upload(facet_cell, n_facets, s));
    FS_CHECK(dop.upload(facet_opposite, n_facets, s));
    if (g) { FS_CHECK(dg.alloc(n_facets)); FS_CHECK(dg.upload(g, n_facets, s)); }
    if (h) { FS_CHECK(dh.alloc(n_facets)); FS_CHECK(dh.upload(h, n_facets, s)); }
    if (space->degree == 2 && m->tdim == 2)
        hipLaunchKernelGGL(k_facet_supg_p2tri, dim3(fs_grid_for(n_facets * 6)), dim3(FS_BLOCK), 0, s, n_facets, dc.p, dop.p, g ? dg.p : (const double*)nullptr,
                           h ? dh.p : (const double*)nullptr, vel, supg_pe, m->cells.p, space->cell_dofs, m->xyz.p, space->n_nodes_owned, space->slice_ptr.p,
                           space->sell_col.p, A ? A->val.p : (double*)nullptr, b ? b->d.p : (double*)nullptr, d_err.p);
    else if (space->degree == 2)
        hipLaunchKernelGGL(k_facet_supg_p2, dim3(fs_grid_for(n_facets * 10)), dim3(FS_BLOCK), 0, s, n_facets, dc.p, dop.p, g ? dg.p : (const double*)nullptr,
                           h ? dh.p : (const double*)nullptr, vel, supg_pe, m->cells.p, space->cell_dofs, m->xyz.p, space->n_nodes_owned, space->slice_ptr.p,
                           space->sell_col.p, A ? A->val.p : (double*)nullptr, b ? b->d.p : (double*)nullptr, d_err.p);
    else if (m->tdim == 2)
        hipLaunchKernelGGL(k_facet_supg_tri, dim3(fs_grid_for(n_facets * 3)), dim3(FS_BLOCK), 0, s, n_facets, dc.p, dop.p, g ? dg.p : (const double*)nullptr,
                           h ? dh.p : (const double*)nullptr, vel, supg_pe, m->cells.p, m->xyz.p, space->n_nodes_owned, space->slice_ptr.p,
                           space->sell_col.p, A ? A->val.p : (double*)nullptr, b ? b->d.p : (double*)nullptr, d_err.p);
    else
        hipLaunchKernelGGL(k_facet_supg, dim3(fs_grid_for(n_facets * 4)), dim3(FS_BLOCK), 0, s, n_facets, dc.p, dop.p, g ? dg.p : (const double*)nullptr,
                           h ? dh.p : (const double*)nullptr, vel, supg_pe, m->cells.p, m->xyz.p, space->n_nodes_owned, space->slice_ptr.p,
                           space->sell_col.p, A ? A->val.p : (double*)nullptr, b ? b->d.p : (double*)nullptr, d_err.p);
    FS_KERNEL_CHECK();
    int h_err = 0;
    FS_CHECK(d_err.download(&h_err, 1, s));
    FS_REQUIRE(h_err == 0, "fs_assemble_facet_supg: %d facet pairs missing from the sparsity pattern", h_err);
    return FS_OK;
}

extern "C" int fs_assemble_facet_vector(fs_space_t space, int64_t n_facets, const int32_t* tri, const double* g,
                                        fs_vector_t b) {
    FS_REQUIRE(space && b && (n_facets == 0 || (tri && g)), "fs_assemble_facet_vector: null pointer");
    if (n_facets == 0) return FS_OK;
    if (space->mesh->tdim == 2) {      // facets are edges: tri holds [n_facets][2] vertex pairs
        for (int64_t i = 0; i < 2 * n_facets; ++i)
            FS_REQUIRE(tri[i] >= 0 && tri[i] < space->n_nodes_local, "fs_assemble_facet_vector: edge vertex %d out of range", tri[i]);
        hipStream_t s2 = fs_rt().stream;
        dbuf<int32_t> d_ed;
        dbuf<double> d_g2;
        FS_CHECK(d_ed.alloc(2 * n_facets));
        FS_CHECK(d_g2.alloc(n_facets * space->ncomp));
        FS_CHECK(d_ed.upload(tri, 2 * n_facets, s2));
        FS_CHECK(d_g2.upload(g, n_facets * space->ncomp, s2));
        if (space->degree == 2) {
            dbuf<int> d_e2;
            FS_CHECK(d_e2.alloc(1));
            FS_CHECK(d_e2.zero(s2));
            if (space->ncomp == 2)
                hipLaunchKernelGGL(k_edge_vector2_p2, dim3(fs_grid_for(n_facets)), dim3(FS_BLOCK), 0, s2, space->mesh->xyz.p, d_ed.p, n_facets, d_g2.p, space->edge_keys.p, space->n_edges, space->edge_grouped, space->edge_node.p, space->n_nodes_owned, b->d.p, d_e2.p);
            else
            hipLaunchKernelGGL(k_edge_vector_p2, dim3(fs_grid_for(n_facets)), dim3(FS_BLOCK), 0, s2, space->mesh->xyz.p, d_ed.p, n_facets, d_g2.p, space->edge_keys.p, space->n_edges, space->edge_grouped, space->edge_node.p, space->n_nodes_owned, b->d.p, d_e2.p);
            FS_KERNEL_CHECK();
            int h_e2 = 0;
            FS_CHECK(d_e2.download(&h_e2, 1, s2));
            FS_REQUIRE(h_e2 == 0, "fs_assemble_facet_vector: %d boundary edges are not mesh edges", h_e2);
            return FS_OK;
        }
        if (space->ncomp == 2)
            hipLaunchKernelGGL(k_edge_vector2, dim3(fs_grid_for(n_facets)), dim3(FS_BLOCK), 0, s2, space->mesh->xyz.p, d_ed.p, n_facets, d_g2.p, space->n_nodes_owned, b->d.p);
        else
            hipLaunchKernelGGL(k_edge_vector, dim3(fs_grid_for(n_facets)), dim3(FS_BLOCK), 0, s2, space->mesh->xyz.p, d_ed.p, n_facets, d_g2.p, space->n_nodes_owned, b->d.p);
        FS_KERNEL_CHECK();
        FS_HIP(hipStreamSynchronize(s2));
        return FS_OK;
    }
    for (int64_t i = 0; i < 3 * n_facets; ++i)
        FS_REQUIRE(tri[i] >= 0 && tri[i] < space->n_nodes_local, "fs_assemble_facet_vector: facet vertex %d out of range", tri[i]);
    hipStream_t s = fs_rt().stream;
    dbuf<int32_t> d_tri;
    dbuf<double> d_g;
    FS_CHECK(d_tri.alloc(3 * n_facets));
    FS_CHECK(d_g.alloc(n_facets * space->ncomp));
    FS_CHECK(d_tri.upload(tri, 3 * n_facets, s));
    FS_CHECK(d_g.upload(g, n_facets * space->ncomp, s));
    if (space->degree == 2) {
        dbuf<int> d_err;
        FS_CHECK(d_err.alloc(1));
        FS_CHECK(d_err.zero(s));
        hipLaunchKernelGGL(k_facet_vector_p2, dim3(fs_grid_for(n_facets)), dim3(FS_BLOCK), 0, s, space->mesh->xyz.p, d_tri.p, n_facets, d_g.p, space->edge_keys.p, space->n_edges, space->edge_grouped, space->edge_node.p, space->n_nodes_owned, space->ncomp, b->d.p, d_err.p);
        FS_KERNEL_CHECK();
        int h_err = 0;
        FS_CHECK(d_err.download(&h_err, 1, s));
        FS_REQUIRE(h_err == 0, "fs_assemble_facet_vector: %d facet edges are not mesh edges", h_err);
        return FS_OK;
    }
    hipLaunchKernelGGL(k_facet_vector, dim3(fs_grid_for(n_facets)), dim3(FS_BLOCK), 0, s, space->mesh->xyz.p, d_tri.p, n_facets, d_g.p, space->ncomp, space->n_nodes_owned, b->d.p);
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

extern "C" int fs_assemble_facet_matrix(fs_matrix_t A, int64_t n_facets, const int32_t* tri, const double* h) {
    FS_REQUIRE(A && (n_facets == 0 || (tri && h)), "fs_assemble_facet_matrix: null pointer");
    if (A->bs != 1) {
        fs_set_error("fs_assemble_facet_matrix: only scalar spaces (Robin/HTC term) are supported");
        return FS_ERR_UNSUPPORTED;
    }
    if (n_facets == 0) return FS_OK;
    fs_space_s* sp = A->space;
    if (sp->degree == 2 && sp->mesh->tdim == 2) {
        for (int64_t i = 0; i < 2 * n_facets; ++i)
            FS_REQUIRE(tri[i] >= 0 && tri[i] < sp->mesh->nv, "fs_assemble_facet_matrix: edge vertex %d out of range", tri[i]);
        hipStream_t s4 = fs_rt().stream;
        dbuf<int32_t> d_e;
        dbuf<double> d_h4;
        dbuf<int> d_err4;
        FS_CHECK(d_e.alloc(2 * n_facets));
        FS_CHECK(d_h4.alloc(n_facets));
        FS_CHECK(d_err4.alloc(1));
        FS_CHECK(d_err4.zero(s4));
        FS_CHECK(d_e.upload(tri, 2 * n_facets, s4));
        FS_CHECK(d_h4.upload(h, n_facets, s4));
        hipLaunchKernelGGL(k_edge_matrix_p2, dim3(fs_grid_for(3 * n_facets)), dim3(FS_BLOCK), 0, s4, sp->mesh->xyz.p, d_e.p, n_facets, d_h4.p,
                           sp->edge_keys.p, sp->n_edges, sp->edge_grouped, sp->edge_node.p, sp->n_nodes_owned, sp->slice_ptr.p, sp->sell_col.p,
                           A->val.p, d_err4.p);
        FS_KERNEL_CHECK();
        int h_err4 = 0;
        FS_CHECK(d_err4.download(&h_err4, 1, s4));
        FS_REQUIRE(h_err4 == 0, "fs_assemble_facet_matrix: %d boundary edges / entries are not in the space", h_err4);
        return FS_OK;
    }
    if (sp->degree == 2) {
        FS_REQUIRE(sp->mesh->tdim == 3, "fs_assemble_facet_matrix: CG2 facet matrices are built for tetrahedral meshes");
        for (int64_t i = 0; i < 3 * n_facets; ++i)
            FS_REQUIRE(tri[i] >= 0 && tri[i] < sp->mesh->nv, "fs_assemble_facet_matrix: facet vertex %d out of range", tri[i]);
        hipStream_t s3 = fs_rt().stream;
        dbuf<int32_t> d_t;
        dbuf<double> d_h3;
        dbuf<int> d_err3;
        FS_CHECK(d_t.alloc(3 * n_facets));
        FS_CHECK(d_h3.alloc(n_facets));
        FS_CHECK(d_err3.alloc(1));
        FS_CHECK(d_err3.zero(s3));
        FS_CHECK(d_t.upload(tri, 3 * n_facets, s3));
        FS_CHECK(d_h3.upload(h, n_facets, s3));
        hipLaunchKernelGGL(k_facet_matrix_p2, dim3(fs_grid_for(6 * n_facets)), dim3(FS_BLOCK), 0, s3, sp->mesh->xyz.p, d_t.p, n_facets, d_h3.p,
                           sp->edge_keys.p, sp->n_edges, sp->edge_grouped, sp->edge_node.p, sp->n_nodes_owned, sp->mesh->n_owned, sp->n_edges_owned,
                           sp->slice_ptr.p, sp->sell_col.p, A->val.p, d_err3.p);
        FS_KERNEL_CHECK();
        int h_err3 = 0;
        FS_CHECK(d_err3.download(&h_err3, 1, s3));
        FS_REQUIRE(h_err3 == 0, "fs_assemble_facet_matrix: %d facet edges / entries are not in the space", h_err3);
        return FS_OK;
    }
    if (sp->mesh->tdim == 2) {
        for (int64_t i = 0; i < 2 * n_facets; ++i)
            FS_REQUIRE(tri[i] >= 0 && tri[i] < sp->n_nodes_local, "fs_assemble_facet_matrix: edge vertex %d out of range", tri[i]);
        hipStream_t s2 = fs_rt().stream;
        dbuf<int32_t> d_ed;
        dbuf<double> d_h2;
        dbuf<int> d_err2;
        FS_CHECK(d_ed.alloc(2 * n_facets));
        FS_CHECK(d_h2.alloc(n_facets));
        FS_CHECK(d_err2.alloc(1));
        FS_CHECK(d_err2.zero(s2));
        FS_CHECK(d_ed.upload(tri, 2 * n_facets, s2));
        FS_CHECK(d_h2.upload(h, n_facets, s2));
        hipLaunchKernelGGL(k_edge_matrix, dim3(fs_grid_for(4 * n_facets)), dim3(FS_BLOCK), 0, s2, sp->mesh->xyz.p, d_ed.p, n_facets, d_h2.p, sp->n_nodes_owned, sp->slice_ptr.p, sp->sell_col.p, A->val.p, d_err2.p);
        FS_KERNEL_CHECK();
        int h_err2 = 0;
        FS_CHECK(d_err2.download(&h_err2, 1, s2));
        FS_REQUIRE(h_err2 == 0, "fs_assemble_facet_matrix: %d edge vertex pairs are not mesh edges", h_err2);
        return FS_OK;
    }
    for (int64_t i = 0; i < 3 * n_facets; ++i)
        FS_REQUIRE(tri[i] >= 0 && tri[i] < sp->n_nodes_local, "fs_assemble_facet_matrix: facet vertex %d out of range", tri[i]);
    hipStream_t s = fs_rt().stream;
    dbuf<int32_t> d_tri;
    dbuf<double> d_h;
    dbuf<int> d_err;
    FS_CHECK(d_tri.alloc(3 * n_facets));
    FS_CHECK(d_h.alloc(n_facets));
    FS_CHECK(d_err.alloc(1));
    FS_CHECK(d_err.zero(s));
    FS_CHECK(d_tri.upload(tri, 3 * n_facets, s));
    FS_CHECK(d_h.upload(h, n_facets, s));
    hipLaunchKernelGGL(k_facet_matrix, dim3(fs_grid_for(n_facets)), dim3(FS_BLOCK), 0, s, sp->mesh->xyz.p, d_tri.p, n_facets, d_h.p, sp->n_nodes_owned, sp->sell_col.p, sp->slice_ptr.p, A->val.p, d_err.p);
    FS_KERNEL_CHECK();
    int h_err = 0;
    FS_CHECK(d_err.download(&h_err, 1, s));
    FS_REQUIRE(h_err == 0, "fs_assemble_facet_matrix: %d facet vertex pairs are not mesh edges", h_err);
    return FS_OK;
}

extern "C" int fs_apply_dirichlet(fs_matrix_t A, fs_vector_t b, int64_t n, const int32_t* dofs, const double* vals,
                                  int symmetric) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(A || b, "fs_apply_dirichlet: both A and b are null");
    FS_REQUIRE(n == 0 || (dofs && vals), "fs_apply_dirichlet: null dof list");
    FS_REQUIRE(n < (int64_t)INT32_MAX, "fs_apply_dirichlet: list too long");
    if (n == 0) return FS_OK;
    hipStream_t s = fs_rt().stream;
    const int64_t n_dofs = A ? A->space->n_dofs_local : b->d.n;
    for (int64_t i = 0; i < n; ++i)
        FS_REQUIRE(dofs[i] >= 0 && dofs[i] < n_dofs, "fs_apply_dirichlet: dof %d outside [0,%lld)", dofs[i], (long long)n_dofs);
    FS_REQUIRE(!A || !b || b->d.n >= A->space->n_dofs_owned, "fs_apply_dirichlet: b shorter than the owned dofs");
    // scratch: kept on the space when there is one (re-assembly every time step must not hipMalloc)
    dbuf<uint8_t> flag_local;
    dbuf<double> g_local;
    dbuf<int32_t> idx_local;
    dbuf<uint8_t>& flag = A ? A->space->bc_flag : flag_local;
    dbuf<double>& g = A ? A->space->bc_g : g_local;
    dbuf<int32_t>& idx = A ? A->space->bc_idx : idx_local;
    if (flag.n != n_dofs) {
        FS_CHECK(flag.alloc(n_dofs));
        FS_CHECK(g.alloc(n_dofs));
        FS_CHECK(idx.alloc(n_dofs));
    }
    dbuf<int32_t> d_dofs;
    dbuf<double> d_vals;
    FS_CHECK(d_dofs.alloc(n));
    FS_CHECK(d_vals.alloc(n));
    FS_HIP(hipMemcpyAsync(d_dofs.p, dofs, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    FS_HIP(hipMemcpyAsync(d_vals.p, vals, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    FS_CHECK(flag.zero(s));
    FS_HIP(hipMemsetAsync(idx.p, 0xff, (size_t)n_dofs * sizeof(int32_t), s));  // -1
    hipLaunchKernelGGL(k_bc_last_index, dim3(fs_grid_for(n)), dim3(FS_BLOCK), 0, s, d_dofs.p, n, idx.p);
    hipLaunchKernelGGL(k_bc_scatter, dim3(fs_grid_for(n)), dim3(FS_BLOCK), 0, s, d_dofs.p, d_vals.p, n, idx.p, flag.p, g.p);
    FS_KERNEL_CHECK();
    if (!A) {
        hipLaunchKernelGGL(k_bc_vector, dim3(fs_grid_for(n_dofs)), dim3(FS_BLOCK), 0, s, flag.p, g.p, n_dofs, b->d.p);
        FS_KERNEL_CHECK();
        FS_HIP(hipStreamSynchronize(s));
        return FS_OK;
    }
    fs_space_s* sp = A->space;
    const int grid = fs_grid_for(sp->n_slices * 64, FS_BLOCK, 8192);
    double* bp = b ? b->d.p : nullptr;
    if (A->bs == 1)
        hipLaunchKernelGGL(k_dirichlet_sell<1>, dim3(grid), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->sell_col.p, A->val.p, sp->sell_entries, flag.p, g.p, bp, symmetric);
    else if (A->bs == 2)
        hipLaunchKernelGGL(k_dirichlet_sell<2>, dim3(grid), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->sell_col.p, A->val.p, sp->sell_entries, flag.p, g.p, bp, symmetric);
    else if (A->bs == 3)
        hipLaunchKernelGGL(k_dirichlet_sell<3>, dim3(grid), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->sell_col.p, A->val.p, sp->sell_entries, flag.p, g.p, bp, symmetric);
    else
        hipLaunchKernelGGL(k_dirichlet_sell<4>, dim3(grid), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->n_slices, sp->slice_ptr.p, sp->sell_col.p, A->val.p, sp->sell_entries, flag.p, g.p, bp, symmetric);
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

extern "C" int fs_assemble_interior_penalty(fs_matrix_t A, int64_t n_facets, const int32_t* facet_cells, double coefficient) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(A && n_facets >= 0 && (n_facets == 0 || facet_cells), "fs_assemble_interior_penalty: bad arguments");
    fs_space_s* sp = A->space;
    if (A->bs != 1 || (sp->degree != 1 && sp->degree != 2)) {
        fs_set_error("fs_assemble_interior_penalty: built for scalar CG1 / CG2 spaces");
        return FS_ERR_UNSUPPORTED;
    }
    if (n_facets == 0) return FS_OK;
    for (int64_t i = 0; i < 2 * n_facets; ++i)
        FS_REQUIRE(facet_cells[i] >= 0 && facet_cells[i] < sp->mesh->nc, "fs_assemble_interior_penalty: facet %lld names cell %d", (long long)(i / 2), facet_cells[i]);
    hipStream_t s = fs_rt().stream;
    dbuf<int32_t> dfc;
    dbuf<int> d_err;
    FS_CHECK(dfc.alloc(2 * n_facets));
    FS_CHECK(d_err.alloc(1));
    FS_CHECK(d_err.zero(s));
    FS_CHECK(dfc.upload(facet_cells, 2 * n_facets, s));
    if (sp->degree == 2 && sp->mesh->tdim == 2)
        hipLaunchKernelGGL(k_interior_penalty_p2<2>, dim3(fs_grid_for(9 * n_facets, FS_BLOCK, 1 << 16)), dim3(FS_BLOCK), 0, s, n_facets, dfc.p,
                           sp->mesh->cells.p, sp->cell_dofs, sp->mesh->xyz.p, coefficient, sp->n_nodes_owned, sp->slice_ptr.p, sp->sell_col.p, A->val.p, d_err.p);
    else if (sp->degree == 2)
        hipLaunchKernelGGL(k_interior_penalty_p2<3>, dim3(fs_grid_for(14 * n_facets, FS_BLOCK, 1 << 16)), dim3(FS_BLOCK), 0, s, n_facets, dfc.p,
                           sp->mesh->cells.p, sp->cell_dofs, sp->mesh->xyz.p, coefficient, sp->n_nodes_owned, sp->slice_ptr.p, sp->sell_col.p, A->val.p, d_err.p);
    else if (sp->mesh->tdim == 2)
        hipLaunchKernelGGL(k_interior_penalty_tri, dim3(fs_grid_for(4 * n_facets, FS_BLOCK, 1 << 16)), dim3(FS_BLOCK), 0, s, n_facets, dfc.p,
                           sp->mesh->cells.p, sp->mesh->xyz.p, coefficient, sp->n_nodes_owned, sp->slice_ptr.p, sp->sell_col.p, A->val.p, d_err.p);
    else
        hipLaunchKernelGGL(k_interior_penalty, dim3(fs_grid_for(5 * n_facets, FS_BLOCK, 1 << 16)), dim3(FS_BLOCK), 0, s, n_facets, dfc.p,
                           sp->mesh->cells.p, sp->mesh->xyz.p, coefficient, sp->n_nodes_owned, sp->slice_ptr.p, sp->sell_col.p, A->val.p, d_err.p);
    FS_KERNEL_CHECK();
    int h_err = 0;
    FS_CHECK(d_err.download(&h_err, 1, s));
    FS_REQUIRE(h_err == 0, "fs_assemble_interior_penalty: %d entries are missing from the sparsity pattern or the cell pairs share no facet "
               "(create the space with fs_space_create_coupled and the pairs of vertices opposite every interior facet)", h_err);
    return FS_OK;
}

void fs_assemble_preload() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(k_bc_last_index));
    (void)hipGetLastError();
}
