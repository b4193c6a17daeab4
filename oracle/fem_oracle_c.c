/*
 * CPU oracle in C (TEST INFRASTRUCTURE ONLY; see oracle/fem_oracle.py for the
 * pinning statement).  A restatement of the reference's CPU path for the P1 heat
 * problem, organised the way DOLFIN/PETSc organise it:
 *
 *   Assembler::assemble        cell loop: tabulate_tensor (4x4 local matrix from
 *                              |detJ| and the barycentric gradients) followed by
 *                              MatSetValues(ADD_VALUES) = binary search of each column
 *                              in its CSR row          (SolverBase.py:595, 608-612)
 *   assemble_system/DirichletBC symmetric elimination  (SolverBase.py:644)
 *   KSPCG + PCJACOBI           textbook PCG from x0 = 0 (SolverBase.py:663-670)
 *
 * It is used (a) by tests to cross-check the numpy oracle and the HIP path at sizes
 * numpy is too slow for and (b) by bench.py as the `cpu_baseline` leg ("port"),
 * parallelised over all host cores with OpenMP where the reference would use
 * `mpirun -n <cores>`.  Never linked into or called from the product.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* dolfin.BoxMesh ordering, see fem_oracle.box_mesh */
void orc_box_mesh(int64_t nx, int64_t ny, int64_t nz, const double* p0, const double* p1, double* xyz,
                  int32_t* cells) {
    const int64_t px = nx + 1, py = ny + 1;
#pragma omp parallel for schedule(static)
    for (int64_t iz = 0; iz <= nz; ++iz)
        for (int64_t iy = 0; iy <= ny; ++iy)
            for (int64_t ix = 0; ix <= nx; ++ix) {
                const int64_t v = (iz * py + iy) * px + ix;
                xyz[3 * v + 0] = p0[0] + ((double)ix * (p1[0] - p0[0])) / (double)nx;
                xyz[3 * v + 1] = p0[1] + ((double)iy * (p1[1] - p0[1])) / (double)ny;
                xyz[3 * v + 2] = p0[2] + ((double)iz * (p1[2] - p0[2])) / (double)nz;
            }
    static const int T[6][4] = {{0, 1, 3, 7}, {0, 1, 5, 7}, {0, 4, 5, 7}, {0, 2, 3, 7}, {0, 4, 6, 7}, {0, 2, 6, 7}};
#pragma omp parallel for schedule(static)
    for (int64_t iz = 0; iz < nz; ++iz)
        for (int64_t iy = 0; iy < ny; ++iy)
            for (int64_t ix = 0; ix < nx; ++ix) {
                const int64_t h = (iz * ny + iy) * nx + ix;
                int32_t c[8];
                for (int k = 0; k < 8; ++k)
                    c[k] = (int32_t)(((iz + ((k >> 2) & 1)) * py + iy + ((k >> 1) & 1)) * px + ix + (k & 1));
                for (int t = 0; t < 6; ++t)
                    for (int a = 0; a < 4; ++a) cells[(h * 6 + t) * 4 + a] = c[T[t][a]];
            }
}

static int cmp_i32(const void* a, const void* b) {
    const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return (x > y) - (x < y);
}

/* Sparsity pattern of the P1 space (sorted columns).  rowptr[n+1] is filled; returns a
 * malloc'ed colidx through *colidx_out (free with orc_free). */
int64_t orc_csr_pattern(int64_t n, int64_t nc, const int32_t* cells, int32_t* rowptr, int32_t** colidx_out) {
    int64_t* cnt = (int64_t*)calloc((size_t)n + 1, sizeof(int64_t));
    for (int64_t c = 0; c < nc; ++c)
        for (int a = 0; a < 4; ++a) cnt[cells[4 * c + a] + 1] += 4;
    for (int64_t i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
    int32_t* tmp = (int32_t*)malloc((size_t)cnt[n] * sizeof(int32_t));
    int64_t* fill = (int64_t*)malloc((size_t)n * sizeof(int64_t));
    memcpy(fill, cnt, (size_t)n * sizeof(int64_t));
    for (int64_t c = 0; c < nc; ++c)
        for (int a = 0; a < 4; ++a) {
            const int32_t r = cells[4 * c + a];
            for (int b = 0; b < 4; ++b) tmp[fill[r]++] = cells[4 * c + b];
        }
    int64_t* len = (int64_t*)malloc((size_t)n * sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; ++i) {
        int32_t* row = tmp + cnt[i];
        const int64_t m = cnt[i + 1] - cnt[i];
        qsort(row, (size_t)m, sizeof(int32_t), cmp_i32);
        int64_t u = 0;
        for (int64_t k = 0; k < m; ++k)
            if (k == 0 || row[k] != row[k - 1]) row[u++] = row[k];
        len[i] = u;
    }
    rowptr[0] = 0;
    for (int64_t i = 0; i < n; ++i) rowptr[i + 1] = rowptr[i] + (int32_t)len[i];
    const int64_t nnz = rowptr[n];
    int32_t* colidx = (int32_t*)malloc((size_t)nnz * sizeof(int32_t));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) memcpy(colidx + rowptr[i], tmp + cnt[i], (size_t)len[i] * sizeof(int32_t));
    free(tmp);
    free(fill);
    free(len);
    free(cnt);
    *colidx_out = colidx;
    return nnz;
}

void orc_free(void* p) { free(p); }

/* The same for a space with nd dofs per cell (cell_dofs[nc][nd]: CG2 = 10 nodes, vector CG1 = 12 scalar dofs): what
 * DOLFIN's SparsityPatternBuilder gives any dofmap (SolverBase.py:595). */
int64_t orc_csr_pattern_generic(int64_t n, int64_t nc, int nd, const int32_t* cell_dofs, int32_t* rowptr, int32_t** colidx_out) {
    int64_t* cnt = (int64_t*)calloc((size_t)n + 1, sizeof(int64_t));
    for (int64_t c = 0; c < nc; ++c)
        for (int a = 0; a < nd; ++a) cnt[cell_dofs[nd * c + a] + 1] += nd;
    for (int64_t i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
    int32_t* tmp = (int32_t*)malloc((size_t)cnt[n] * sizeof(int32_t));
    int64_t* fill = (int64_t*)malloc((size_t)n * sizeof(int64_t));
    memcpy(fill, cnt, (size_t)n * sizeof(int64_t));
    for (int64_t c = 0; c < nc; ++c)
        for (int a = 0; a < nd; ++a) {
            const int32_t r = cell_dofs[nd * c + a];
            for (int b = 0; b < nd; ++b) tmp[fill[r]++] = cell_dofs[nd * c + b];
        }
    int64_t* len = (int64_t*)malloc((size_t)n * sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; ++i) {
        int32_t* row = tmp + cnt[i];
        const int64_t m = cnt[i + 1] - cnt[i];
        qsort(row, (size_t)m, sizeof(int32_t), cmp_i32);
        int64_t u = 0;
        for (int64_t k = 0; k < m; ++k)
            if (k == 0 || row[k] != row[k - 1]) row[u++] = row[k];
        len[i] = u;
    }
    int64_t total = 0;
    for (int64_t i = 0; i < n; ++i) total += len[i];
    if (total >= (int64_t)INT32_MAX) { free(tmp); free(fill); free(len); free(cnt); return -1; }
    rowptr[0] = 0;
    for (int64_t i = 0; i < n; ++i) rowptr[i + 1] = rowptr[i] + (int32_t)len[i];
    const int64_t nnz = rowptr[n];
    int32_t* colidx = (int32_t*)malloc((size_t)nnz * sizeof(int32_t));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) memcpy(colidx + rowptr[i], tmp + cnt[i], (size_t)len[i] * sizeof(int32_t));
    free(tmp);
    free(fill);
    free(len);
    free(cnt);
    *colidx_out = colidx;
    return nnz;
}

/* barycentric gradients and volume of one tetrahedron */
static double tet_gradients(const double* xyz, const int32_t* v, double g[4][3]) {
    const double* x0 = xyz + 3 * (int64_t)v[0];
    const double* x1 = xyz + 3 * (int64_t)v[1];
    const double* x2 = xyz + 3 * (int64_t)v[2];
    const double* x3 = xyz + 3 * (int64_t)v[3];
    const double e1[3] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]};
    const double e2[3] = {x2[0] - x0[0], x2[1] - x0[1], x2[2] - x0[2]};
    const double e3[3] = {x3[0] - x0[0], x3[1] - x0[1], x3[2] - x0[2]};
    const double c1[3] = {e2[1] * e3[2] - e2[2] * e3[1], e2[2] * e3[0] - e2[0] * e3[2], e2[0] * e3[1] - e2[1] * e3[0]};
    const double c2[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
    const double c3[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double det = e1[0] * c1[0] + e1[1] * c1[1] + e1[2] * c1[2];
    for (int d = 0; d < 3; ++d) {
        g[1][d] = c1[d] / det;
        g[2][d] = c2[d] / det;
        g[3][d] = c3[d] / det;
        g[0][d] = -(g[1][d] + g[2][d] + g[3][d]);
    }
    return fabs(det) / 6.0;
}

static void add_to_row(const int32_t* rowptr, const int32_t* colidx, double* vals, int32_t r, int32_t col, double v) {
    int32_t lo = rowptr[r], hi = rowptr[r + 1];
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (colidx[mid] < col) lo = mid + 1; else hi = mid;
    }
#pragma omp atomic
    vals[lo] += v;
}

/* CG2 stiffness  k inner(grad T, grad q) dx  (ScalarTransportSolver.py:278-281 with fe_degree 2): tabulate_tensor with the
 * 4-point degree-2 rule FFC picks for the quadratic integrand; local dofs = 4 vertices, then the UFC edges
 * e0=(2,3) e1=(1,3) e2=(1,2) e3=(0,3) e4=(0,2) e5=(0,1); grad phi_vertex = (4 l - 1) g, grad phi_edge = 4 (l_i g_j + l_j g_i). */
void orc_assemble_p2(int64_t n, int64_t nc, const double* xyz, const int32_t* cells, const int32_t* cell_dofs, double k,
                     const int32_t* rowptr, const int32_t* colidx, double* vals) {
    static const int EI[6] = {2, 1, 1, 0, 0, 0}, EJ[6] = {3, 3, 2, 3, 2, 1};
    const double qa = 0.5854101966249685, qb = 0.1381966011250105;
    memset(vals, 0, (size_t)rowptr[n] * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < nc; ++c) {
        double g[4][3];
        const double vol = tet_gradients(xyz, cells + 4 * c, g);
        double ke[10][10];
        memset(ke, 0, sizeof(ke));
        for (int q = 0; q < 4; ++q) {
            double l[4] = {qb, qb, qb, qb};
            l[q] = qa;
            double gp[10][3];
            for (int i = 0; i < 4; ++i)
                for (int d = 0; d < 3; ++d) gp[i][d] = (4.0 * l[i] - 1.0) * g[i][d];
            for (int e = 0; e < 6; ++e)
                for (int d = 0; d < 3; ++d) gp[4 + e][d] = 4.0 * (l[EI[e]] * g[EJ[e]][d] + l[EJ[e]] * g[EI[e]][d]);
            for (int a = 0; a < 10; ++a)
                for (int b = 0; b < 10; ++b)
                    ke[a][b] += 0.25 * vol * k * (gp[a][0] * gp[b][0] + gp[a][1] * gp[b][1] + gp[a][2] * gp[b][2]);
        }
        const int32_t* d = cell_dofs + 10 * c;
        for (int a = 0; a < 10; ++a)
            for (int b = 0; b < 10; ++b) add_to_row(rowptr, colidx, vals, d[a], d[b], ke[a][b]);
    }
}

/* Vector CG1 elasticity  inner(sigma(u), eps(v)) dx, sigma = 2 mu eps + lambda tr(eps) I (LinearElasticitySolver.py:152-167):
 * Ke[(a,i),(b,j)] = V (lambda g_a[i] g_b[j] + mu g_a[j] g_b[i] + mu delta_ij g_a . g_b); scalar dof = 3 * vertex + component. */
void orc_assemble_p1_elasticity(int64_t n_dofs, int64_t nc, const double* xyz, const int32_t* cells, double mu, double lambda,
                                const int32_t* rowptr, const int32_t* colidx, double* vals) {
    memset(vals, 0, (size_t)rowptr[n_dofs] * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < nc; ++c) {
        double g[4][3];
        const int32_t* v = cells + 4 * c;
        const double vol = tet_gradients(xyz, v, g);
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                const double gg = g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2];
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) {
                        const double kij = vol * (lambda * g[a][i] * g[b][j] + mu * g[a][j] * g[b][i] + (i == j ? mu * gg : 0.0));
                        add_to_row(rowptr, colidx, vals, 3 * v[a] + i, 3 * v[b] + j, kij);
                    }
            }
    }
}


/* tabulate_tensor of  k * inner(grad T, grad q) * dx  on one P1 tetrahedron */
static void p1_stiffness(const double* xyz, const int32_t* v, double k, double ke[4][4]) {
    const double* x0 = xyz + 3 * (int64_t)v[0];
    const double* x1 = xyz + 3 * (int64_t)v[1];
    const double* x2 = xyz + 3 * (int64_t)v[2];
    const double* x3 = xyz + 3 * (int64_t)v[3];
    const double e1[3] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]};
    const double e2[3] = {x2[0] - x0[0], x2[1] - x0[1], x2[2] - x0[2]};
    const double e3[3] = {x3[0] - x0[0], x3[1] - x0[1], x3[2] - x0[2]};
    const double c1[3] = {e2[1] * e3[2] - e2[2] * e3[1], e2[2] * e3[0] - e2[0] * e3[2], e2[0] * e3[1] - e2[1] * e3[0]};
    const double c2[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
    const double c3[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double det = e1[0] * c1[0] + e1[1] * c1[1] + e1[2] * c1[2];
    double g[4][3];
    for (int d = 0; d < 3; ++d) {
        g[1][d] = c1[d] / det;
        g[2][d] = c2[d] / det;
        g[3][d] = c3[d] / det;
        g[0][d] = -(g[1][d] + g[2][d] + g[3][d]);
    }
    const double w = k * fabs(det) / 6.0;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) ke[a][b] = w * (g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2]);
}

/* Assembler::assemble: cell loop + MatSetValues(ADD) by binary search in the row */
void orc_assemble_p1(int64_t n, int64_t nc, const double* xyz, const int32_t* cells, double k, const double* kcell,
                     const int32_t* rowptr, const int32_t* colidx, double* vals) {
    memset(vals, 0, (size_t)rowptr[n] * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < nc; ++c) {
        const int32_t* v = cells + 4 * c;
        double ke[4][4];
        p1_stiffness(xyz, v, kcell ? kcell[c] : k, ke);
        for (int a = 0; a < 4; ++a) {
            const int32_t r = v[a];
            for (int b = 0; b < 4; ++b) {
                int32_t lo = rowptr[r], hi = rowptr[r + 1];
                while (lo < hi) {
                    const int32_t mid = (lo + hi) >> 1;
                    if (colidx[mid] < v[b]) lo = mid + 1; else hi = mid;
                }
#pragma omp atomic
                vals[lo] += ke[a][b];
            }
        }
    }
}

/* assemble_system-style symmetric Dirichlet elimination; flag/g are full-length arrays */
void orc_apply_dirichlet(int64_t n, const int32_t* rowptr, const int32_t* colidx, double* vals, double* b,
                         const uint8_t* flag, const double* g, int symmetric) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        if (flag[i]) {
            for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) vals[k] = colidx[k] == i ? 1.0 : 0.0;
            b[i] = g[i];
        } else if (symmetric) {
            double acc = 0.0;
            for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
                if (flag[colidx[k]]) {
                    acc += vals[k] * g[colidx[k]];
                    vals[k] = 0.0;
                }
            b[i] -= acc;
        }
    }
}

static void spmv(int64_t n, const int32_t* rowptr, const int32_t* colidx, const double* vals, const double* x,
                 double* y) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) acc += vals[k] * x[colidx[k]];
        y[i] = acc;
    }
}

void orc_spmv(int64_t n, const int32_t* rowptr, const int32_t* colidx, const double* vals, const double* x,
              double* y) {
    spmv(n, rowptr, colidx, vals, x, y);
}

static double dot(int64_t n, const double* a, const double* b) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* KSPCG + PCJACOBI, x0 = 0, stop on ||r||_2 <= rtol ||b||_2.  Returns iterations;
 * hist (may be NULL) receives ||r_k||^2 for k = 0..iterations. */
int orc_pcg_jacobi(int64_t n, const int32_t* rowptr, const int32_t* colidx, const double* vals, const double* b,
                   double* x, double rtol, int maxit, double* hist) {
    double* r = (double*)malloc((size_t)n * sizeof(double));
    double* z = (double*)malloc((size_t)n * sizeof(double));
    double* p = (double*)malloc((size_t)n * sizeof(double));
    double* q = (double*)malloc((size_t)n * sizeof(double));
    double* dinv = (double*)malloc((size_t)n * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double d = 1.0;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
            if (colidx[k] == i) d = vals[k];
        dinv[i] = 1.0 / d;
        x[i] = 0.0;
        r[i] = b[i];
        z[i] = dinv[i] * b[i];
        p[i] = z[i];
    }
    const double bb = dot(n, b, b);
    const double thresh = rtol * rtol * bb;
    double rr = dot(n, r, r);
    if (hist) hist[0] = rr;
    double rz = dot(n, r, z);
    int it = 0;
    while (rr > thresh && it < maxit) {
        spmv(n, rowptr, colidx, vals, p, q);
        const double alpha = rz / dot(n, p, q);
        double rr_new = 0.0, rz_new = 0.0;
#pragma omp parallel for reduction(+ : rr_new, rz_new) schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            x[i] += alpha * p[i];
            const double ri = r[i] - alpha * q[i];
            r[i] = ri;
            const double zi = dinv[i] * ri;
            z[i] = zi;
            rr_new += ri * ri;
            rz_new += ri * zi;
        }
        ++it;
        rr = rr_new;
        if (hist) hist[it] = rr;
        if (rr <= thresh) break;
        const double beta = rz_new / rz;
        rz = rz_new;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    }
    free(r); free(z); free(p); free(q); free(dinv);
    return it;
}

/* The whole config-2 hot path on the host: numeric assembly + Dirichlet + PCG to rtol.
 * Pattern construction is timed separately, as for the GPU path.  Dirichlet planes:
 * axis (0,1,2) = 0 -> t_lo, axis = extent -> t_hi.  times[4] = {mesh, symbolic, assemble+bc, solve}. */
int orc_heat_box_solve(int64_t nx, int64_t ny, int64_t nz, const double* p1, double k, int axis, double t_lo,
                       double t_hi, double rtol, int maxit, double* x_out, double* times, int64_t* nnz_out) {
    const int64_t n = (nx + 1) * (ny + 1) * (nz + 1), nc = 6 * nx * ny * nz;
    const double p0[3] = {0.0, 0.0, 0.0};
    double t0 = now_s();
    double* xyz = (double*)malloc((size_t)n * 3 * sizeof(double));
    int32_t* cells = (int32_t*)malloc((size_t)nc * 4 * sizeof(int32_t));
    orc_box_mesh(nx, ny, nz, p0, p1, xyz, cells);
    double t1 = now_s();
    int32_t* rowptr = (int32_t*)malloc((size_t)(n + 1) * sizeof(int32_t));
    int32_t* colidx = NULL;
    const int64_t nnz = orc_csr_pattern(n, nc, cells, rowptr, &colidx);
    double* vals = (double*)malloc((size_t)nnz * sizeof(double));
    double* b = (double*)calloc((size_t)n, sizeof(double));
    uint8_t* flag = (uint8_t*)calloc((size_t)n, 1);
    double* g = (double*)calloc((size_t)n, sizeof(double));
    for (int64_t i = 0; i < n; ++i) {
        const double c = xyz[3 * i + axis];
        if (c == 0.0) { flag[i] = 1; g[i] = t_lo; }
        else if (c == p1[axis]) { flag[i] = 1; g[i] = t_hi; }
    }
    double t2 = now_s();
    orc_assemble_p1(n, nc, xyz, cells, k, NULL, rowptr, colidx, vals);
    orc_apply_dirichlet(n, rowptr, colidx, vals, b, flag, g, 1);
    double t3 = now_s();
    const int it = orc_pcg_jacobi(n, rowptr, colidx, vals, b, x_out, rtol, maxit, NULL);
    double t4 = now_s();
    if (times) { times[0] = t1 - t0; times[1] = t2 - t1; times[2] = t3 - t2; times[3] = t4 - t3; }
    if (nnz_out) *nnz_out = nnz;
    free(xyz); free(cells); free(rowptr); free(colidx); free(vals); free(b); free(flag); free(g);
    return it;
}
