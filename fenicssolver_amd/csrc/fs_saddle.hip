// Taylor-Hood Navier-Stokes for libfsamd.so (gfx950): assembly of the linearised coupled system and its
// Krylov solve.  Replaces what CoupledNavierStokesSolver hands to DOLFIN/PETSc
// (FenicsSolver/CoupledNavierStokesSolver.py:288-381 forms, :215-245 action/derivative, :492-528 solve,
// SolverBase.py:615-626 NonlinearVariationalSolver):
//   J(w0) w = g(w0)   with   J = 2 nu eps(u):eps(v) + (1/dt) u.v + (grad(u) u0).v [+ (grad(u0) u).v]
//                                 - (p/rho) div v + (q/rho) div u
//                            g = f.v + (1/dt) u_prev.v [+ (grad(u0) u0).v]
// (the Newton step written for the new iterate, so Dirichlet values are applied directly).
//
// Layout: one block of 4 unknowns per P2 node (u_x, u_y, u_z, p), the matrix is a bs = 4 matrix on the CG2
// node pattern (hybrid SELL/DIA like every other operator, so the tuned SpMV applies).  The pressure is P1:
// only vertex nodes carry a pressure; the pressure slot of an edge node is a dummy unknown with a unit row.
//
// Assembly: one thread per (cell, test node a, trial node b): 14-point degree-5 rule (exact for the
// P2*P2*P1 convection integrand), the 4x4 block accumulated in registers, fp64 hardware atomics into the
// 16 value planes through the slot table.  Solve: restarted FGMRES, right preconditioner
//   [A 0; D S]^-1  with  A^-1 ~ Jacobi sweeps,  S^-1 ~ rho^2 ( (1/dt) K_p^-1 + nu M_p^-1 )  (Cahouet-Chabard),
// K_p, M_p = P1 pressure Laplacian / mass matrix solved by the library's own CG.
#include "fs_kernels.h"
#include <hipcub/hipcub.hpp>
#include <chrono>
#include <stdlib.h>
#include <cmath>
#include <vector>

// barycentric points / weights (sum 1) of the 14-point degree-5 rule
__device__ const double NS_QP[14][4] = {
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
__device__ const double NS_QW[14] = {
    0.1126879257180159, 0.1126879257180159, 0.1126879257180159, 0.1126879257180159,
    0.0734930431163620, 0.0734930431163620, 0.0734930431163620, 0.0734930431163620,
    0.0425460207770815, 0.0425460207770815, 0.0425460207770815, 0.0425460207770815, 0.0425460207770815, 0.0425460207770815};
// UFC edge -> vertex pairs of the P2 edge nodes 4..9: e0=(2,3) e1=(1,3) e2=(1,2) e3=(0,3) e4=(0,2) e5=(0,1)
__device__ const int NS_EI[6] = {2, 1, 1, 0, 0, 0};
__device__ const int NS_EJ[6] = {3, 3, 2, 3, 2, 1};

__device__ __forceinline__ double sel4(int i, double a, double b, double c, double d) {
    return i == 0 ? a : (i == 1 ? b : (i == 2 ? c : d));
}

// value and physical gradient of P2 basis function n (runtime index) at barycentric point l
__device__ __forceinline__ void p2_eval(int n, const double l[4], const double gl[4][3], double* phi, double g[3]) {
    if (n < 4) {
        const double li = sel4(n, l[0], l[1], l[2], l[3]);
        const double d = 4.0 * li - 1.0;
        *phi = li * (2.0 * li - 1.0);
#pragma unroll
        for (int k = 0; k < 3; ++k) g[k] = d * sel4(n, gl[0][k], gl[1][k], gl[2][k], gl[3][k]);
    } else {
        const int i = NS_EI[n - 4], j = NS_EJ[n - 4];
        const double li = sel4(i, l[0], l[1], l[2], l[3]), lj = sel4(j, l[0], l[1], l[2], l[3]);
        *phi = 4.0 * li * lj;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            g[k] = 4.0 * (li * sel4(j, gl[0][k], gl[1][k], gl[2][k], gl[3][k]) + lj * sel4(i, gl[0][k], gl[1][k], gl[2][k], gl[3][k]));
    }
}

struct ns_params {
    double nu, inv_rho, inv_dt;
    double f[3];
    double wm[3];      // ALE: the frame (mesh) velocity, subtracted from the ADVECTING velocity only (:321-329)
    int convection, newton;
    int g2;            // G2 streamline term (:334-363): 0 off, 1 delta1 = kappa1 h^2 (Re <= 1), 2 delta1 from |a|, h (and dt)
    double g2_kappa;
    fs_visc_dev V;     // non-Newtonian law (:194-213) evaluated at every quadrature point on the state w0 (fs_common.h); kind 0: off
};

// h = 2 * circumradius of the tetrahedron X (UFL's 2*Circumradius(mesh), :343):
// R = sqrt((aA+bB+cC)(aA+bB-cC)(aA-bB+cC)(-aA+bB+cC)) / (24 V), (a,A) (b,B) (c,C) the opposite edge pairs
__device__ __forceinline__ double ns_cell_h(const double (&X)[4][3], double vol) {
    auto dist = [&](int p, int q) {
        const double dx = X[p][0] - X[q][0], dy = X[p][1] - X[q][1], dz = X[p][2] - X[q][2];
        return sqrt(dx * dx + dy * dy + dz * dz);
    };
    const double aA = dist(0, 1) * dist(2, 3), bB = dist(0, 2) * dist(1, 3), cC = dist(0, 3) * dist(1, 2);
    const double prod = (aA + bB + cC) * (aA + bB - cC) * (aA - bB + cC) * (-aA + bB + cC);
    return 2.0 * sqrt(prod > 0.0 ? prod : 0.0) / (24.0 * vol);
}
// delta1 of the reference's G2 term at a point with advecting velocity a (|a|^2 = U2): kappa1 h^2 for Re <= 1, otherwise
// kappa1/2 * h/|a| (steady) or kappa1/2 / sqrt(1/dt^2 + 1/(|a|^2 h^2)) (transient, as the reference writes it).  The
// term delta1 (a.grad u)(a.grad v) vanishes with |a|: delta1 = 0 where a = 0 (UFL would divide by zero there).
__device__ __forceinline__ double ns_g2_delta(const ns_params& P, double h, double U2) {
    if (P.g2 == 1) return P.g2_kappa * h * h;
    if (U2 <= 0.0) return 0.0;
    if (P.inv_dt != 0.0) return 0.5 * P.g2_kappa / sqrt(P.inv_dt * P.inv_dt + 1.0 / (U2 * h * h));
    return 0.5 * P.g2_kappa * h / sqrt(U2);
}

// thread t = ab*nc + c : block (a, b) of cell c.  val planes [(i*4+j)*plane + slot], g [node*4 + i]
__global__ void __launch_bounds__(FS_BLOCK) k_assemble_ns(const double* __restrict__ xyz, const int32_t* __restrict__ cells,
                                                          const int32_t* __restrict__ cell_dofs,
                                                          int64_t nc, const int32_t* __restrict__ slots,
                                                          const double* __restrict__ w0, const double* __restrict__ wprev,
                                                          ns_params P, double* __restrict__ val, int64_t plane,
                                                          double* __restrict__ g) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nc * 100; t += stride) {
        const int ab = (int)(t / nc);
        const int64_t c = t - (int64_t)ab * nc;
        const int a = ab / 10, b = ab - a * 10;
        const int32_t slot = slots[t];
        if (slot < 0) continue;   // row not owned (single GPU: never)
        int32_t nd[10];
#pragma unroll
        for (int n = 0; n < 10; ++n) nd[n] = cell_dofs[c * 10 + n];
        // geometry: gradients of the barycentric coordinates (vertex ids: node ids differ once ghosts exist)
        double X[4][3];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t vx = cells[c * 4 + v];
            const double2 p01 = reinterpret_cast<const double2*>(xyz)[2 * vx];
            X[v][0] = p01.x; X[v][1] = p01.y; X[v][2] = xyz[4 * vx + 2];
        }
        const double e1[3] = {X[1][0] - X[0][0], X[1][1] - X[0][1], X[1][2] - X[0][2]};
        const double e2[3] = {X[2][0] - X[0][0], X[2][1] - X[0][1], X[2][2] - X[0][2]};
        const double e3[3] = {X[3][0] - X[0][0], X[3][1] - X[0][1], X[3][2] - X[0][2]};
        const double c23[3] = {e2[1] * e3[2] - e2[2] * e3[1], e2[2] * e3[0] - e2[0] * e3[2], e2[0] * e3[1] - e2[1] * e3[0]};
        const double c31[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
        const double c12[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double det = e1[0] * c23[0] + e1[1] * c23[1] + e1[2] * c23[2];
        const double idet = 1.0 / det;
        double gl[4][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gl[1][k] = c23[k] * idet;
            gl[2][k] = c31[k] * idet;
            gl[3][k] = c12[k] * idet;
            gl[0][k] = -(gl[1][k] + gl[2][k] + gl[3][k]);
        }
        const double vol = fabs(det) * (1.0 / 6.0);
        const double hcell = P.g2 ? ns_cell_h(X, vol) : 0.0;
        // state at the cell nodes
        double U0[10][3];
        if (P.convection) {
#pragma unroll
            for (int n = 0; n < 10; ++n) {
                const double2 u01 = reinterpret_cast<const double2*>(w0)[2 * (int64_t)nd[n]];
                U0[n][0] = u01.x; U0[n][1] = u01.y; U0[n][2] = w0[4 * (int64_t)nd[n] + 2];
            }
        }
        double blk[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) blk[i][j] = 0.0;
        double gv[3] = {0.0, 0.0, 0.0};
        const bool do_rhs = b == 0;
        double P0[4] = {0.0, 0.0, 0.0, 0.0}, T0[4] = {0.0, 0.0, 0.0, 0.0};
        if (P.V.kind) {
#pragma unroll
            for (int v = 0; v < 4; ++v) P0[v] = w0[4 * (int64_t)nd[v] + 3];
            if (P.V.kind == 2)
#pragma unroll
                for (int v = 0; v < 4; ++v) T0[v] = P.V.T[nd[v]];
        }
        for (int q = 0; q < 14; ++q) {
            const double l[4] = {NS_QP[q][0], NS_QP[q][1], NS_QP[q][2], NS_QP[q][3]};
            const double wv = NS_QW[q] * vol;
            const double nuq = fs_viscosity(P.V, P.nu, l[0] * P0[0] + l[1] * P0[1] + l[2] * P0[2] + l[3] * P0[3],
                                            l[0] * T0[0] + l[1] * T0[1] + l[2] * T0[2] + l[3] * T0[3]);
            double pa, pb, ga[3], gb[3];
            p2_eval(a, l, gl, &pa, ga);
            p2_eval(b, l, gl, &pb, gb);
            const double gg = ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2];
            double diag = nuq * gg + P.inv_dt * pa * pb;
            double u0[3] = {0.0, 0.0, 0.0}, gu0[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            if (P.convection) {
#pragma unroll
                for (int n = 0; n < 10; ++n) {
                    double pn, gn[3];
                    p2_eval(n, l, gl, &pn, gn);      // n is a compile-time constant here: the selects fold away
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        u0[i] += pn * U0[n][i];
#pragma unroll
                        for (int j = 0; j < 3; ++j) gu0[i][j] += U0[n][i] * gn[j];
                    }
                }
                const double av[3] = {u0[0] - P.wm[0], u0[1] - P.wm[1], u0[2] - P.wm[2]};
                const double agb = av[0] * gb[0] + av[1] * gb[1] + av[2] * gb[2];
                diag += pa * agb;
                if (P.g2) {      // F -= delta1 (a.grad u).(a.grad v) dx, the sign as the reference has it (:360-362)
                    const double aga = av[0] * ga[0] + av[1] * ga[1] + av[2] * ga[2];
                    diag -= ns_g2_delta(P, hcell, av[0] * av[0] + av[1] * av[1] + av[2] * av[2]) * aga * agb;
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                blk[i][i] += wv * diag;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double v = nuq * ga[j] * gb[i];
                    if (P.convection && P.newton) v += pa * pb * gu0[i][j];
                    blk[i][j] += wv * v;
                }
            }
            if (b < 4) {   // pressure trial function psi_b = lambda_b
                const double psi = sel4(b, l[0], l[1], l[2], l[3]);
#pragma unroll
                for (int i = 0; i < 3; ++i) blk[i][3] -= wv * P.inv_rho * psi * ga[i];
            }
            if (a < 4) {   // continuity test function psi_a
                const double psi = sel4(a, l[0], l[1], l[2], l[3]);
#pragma unroll
                for (int j = 0; j < 3; ++j) blk[3][j] += wv * P.inv_rho * psi * gb[j];
            }
            if (do_rhs) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    double v = P.f[i];
                    if (P.convection && P.newton) v += gu0[i][0] * u0[0] + gu0[i][1] * u0[1] + gu0[i][2] * u0[2];
                    gv[i] += wv * pa * v;
                }
                if (wprev && P.inv_dt != 0.0) {
                    double up[3] = {0.0, 0.0, 0.0};
#pragma unroll
                    for (int n = 0; n < 10; ++n) {
                        double pn, gn[3];
                        p2_eval(n, l, gl, &pn, gn);
#pragma unroll
                        for (int i = 0; i < 3; ++i) up[i] += pn * wprev[4 * (int64_t)nd[n] + i];
                    }
#pragma unroll
                    for (int i = 0; i < 3; ++i) gv[i] += wv * P.inv_dt * pa * up[i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (i == 3 && j == 3) continue;
                if (j == 3 && b >= 4) continue;
                if (i == 3 && a >= 4) continue;
                atomicAdd(&val[(int64_t)(i * 4 + j) * plane + slot], blk[i][j]);
            }
        if (do_rhs) {
#pragma unroll
            for (int i = 0; i < 3; ++i) atomicAdd(&g[4 * (int64_t)nd[a] + i], gv[i]);
        }
    }
}

// ---- the same element, one wave per cell ---------------------------------------------------------------------------
// The kernel above (one thread per (cell, a, b), ordered pair-major; FS_NS_ASSEMBLE=pair) recomputes the state at the
// 14 points in each of the 100 threads of a cell.  Here a wave owns a cell: basis functions and state at the quadrature
// points are built once in LDS, lane ab accumulates block (a, b), and all atomics of a cell - and of its neighbours in
// the cell order, which share its rows - are issued together.  Measured on MI355X (configs[4], 477 042 cells): 24.7 ms
// against 22.4 ms for the pair-major kernel - neither the redundant flops nor the locality of the lines is the
// limit, the 544 M device-scope fp64 atomics are (22 G/s: they are resolved behind the per-XCD L2s).  The next
// step is an owner-computes gather (one wave per matrix row, no atomics), as the P1 scalar path already does.
// Occupancy: 4 waves per SIMD (127 VGPRs, 20 B of scratch) 5.9 ms; the compiler's own choice 3 waves (137 VGPRs) 6.9 ms;
// 5 waves (96 VGPRs, 132 B of scratch) 7.2 ms - configs[4], 477 042 cells.
#define NS_WPB (FS_BLOCK / 64)
struct ns_cell_lds {
    double phi[14][10];
    double gphi[14][10][3];
    double u0[14][3];
    double gu0[14][9];
    double up[14][3];
    double U0[10][3];
    double UP[10][3];
    double nuq[14];
    double P0[4], T0[4];
    int32_t nd[10];
    int32_t pad[2];
};
__global__ void __launch_bounds__(FS_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) k_assemble_ns_wave(const double* __restrict__ xyz, const int32_t* __restrict__ cells,
                                                               const int32_t* __restrict__ cell_dofs,
                                                               int64_t nc, int64_t n_rows, const int32_t* __restrict__ slots,
                                                               const double* __restrict__ w0, const double* __restrict__ wprev,
                                                               ns_params P, double* __restrict__ val, int64_t plane,
                                                               double* __restrict__ ebuf, double* __restrict__ g) {
    __shared__ ns_cell_lds S[NS_WPB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ns_cell_lds& L = S[wave];
    const bool has_prev = wprev != nullptr && P.inv_dt != 0.0;
    for (int64_t cbase = (int64_t)blockIdx.x * NS_WPB; cbase < nc; cbase += (int64_t)gridDim.x * NS_WPB) {
        const int64_t c = cbase + wave;
        const bool act = c < nc;
        double gl[4][3], vol = 0.0, hcell = 0.0;
        if (act) {
            // node ids and nodal state -> LDS; geometry in registers (every lane, broadcast loads)
            if (lane < 10) L.nd[lane] = cell_dofs[c * 10 + lane];
            if (lane < 30) {
                const int n = lane / 3, i = lane - 3 * n;
                const int64_t node = cell_dofs[c * 10 + n];
                L.U0[n][i] = P.convection ? w0[4 * node + i] : 0.0;
                L.UP[n][i] = has_prev ? wprev[4 * node + i] : 0.0;
            } else if (lane < 34) {
                L.P0[lane - 30] = P.V.kind ? w0[4 * (int64_t)cell_dofs[c * 10 + (lane - 30)] + 3] : 0.0;
            } else if (lane < 38) {
                L.T0[lane - 34] = P.V.kind == 2 ? P.V.T[cell_dofs[c * 10 + (lane - 34)]] : 0.0;
            }
            double X[4][3];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t vx = cells[c * 4 + v];          // vertex ids: node ids differ once ghosts exist
                const double2 p01 = reinterpret_cast<const double2*>(xyz)[2 * vx];
                X[v][0] = p01.x; X[v][1] = p01.y; X[v][2] = xyz[4 * vx + 2];
            }
            const double e1[3] = {X[1][0] - X[0][0], X[1][1] - X[0][1], X[1][2] - X[0][2]};
            const double e2[3] = {X[2][0] - X[0][0], X[2][1] - X[0][1], X[2][2] - X[0][2]};
            const double e3[3] = {X[3][0] - X[0][0], X[3][1] - X[0][1], X[3][2] - X[0][2]};
            const double c23[3] = {e2[1] * e3[2] - e2[2] * e3[1], e2[2] * e3[0] - e2[0] * e3[2], e2[0] * e3[1] - e2[1] * e3[0]};
            const double c31[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
            const double c12[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            const double det = e1[0] * c23[0] + e1[1] * c23[1] + e1[2] * c23[2];
            const double idet = 1.0 / det;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                gl[1][k] = c23[k] * idet;
                gl[2][k] = c31[k] * idet;
                gl[3][k] = c12[k] * idet;
                gl[0][k] = -(gl[1][k] + gl[2][k] + gl[3][k]);
            }
            vol = fabs(det) * (1.0 / 6.0);
            if (P.g2) hcell = ns_cell_h(X, vol);
            // basis functions and gradients at the 14 points
            for (int item = lane; item < 140; item += 64) {
                const int q = item / 10, n = item - 10 * q;
                const double l[4] = {NS_QP[q][0], NS_QP[q][1], NS_QP[q][2], NS_QP[q][3]};
                double pn, gn[3];
                p2_eval(n, l, gl, &pn, gn);
                L.phi[q][n] = pn;
                L.gphi[q][n][0] = gn[0]; L.gphi[q][n][1] = gn[1]; L.gphi[q][n][2] = gn[2];
            }
        }
        __syncthreads();
        if (act) {
            // state, its gradient and the previous velocity at the 14 points
            if (lane < 14)
                L.nuq[lane] = fs_viscosity(P.V, P.nu, NS_QP[lane][0] * L.P0[0] + NS_QP[lane][1] * L.P0[1] + NS_QP[lane][2] * L.P0[2] + NS_QP[lane][3] * L.P0[3],
                                           NS_QP[lane][0] * L.T0[0] + NS_QP[lane][1] * L.T0[1] + NS_QP[lane][2] * L.T0[2] + NS_QP[lane][3] * L.T0[3]);
            for (int item = lane; item < 210; item += 64) {
                double acc = 0.0;
                if (item < 42) {
                    const int q = item / 3, i = item - 3 * q;
#pragma unroll
                    for (int n = 0; n < 10; ++n) acc += L.phi[q][n] * L.U0[n][i];
                    L.u0[q][i] = acc;
                } else if (item < 168) {
                    const int it = item - 42, q = it / 9, ij = it - 9 * q, i = ij / 3, j = ij - 3 * i;
#pragma unroll
                    for (int n = 0; n < 10; ++n) acc += L.U0[n][i] * L.gphi[q][n][j];
                    L.gu0[q][ij] = acc;
                } else {
                    const int it = item - 168, q = it / 3, i = it - 3 * q;
#pragma unroll
                    for (int n = 0; n < 10; ++n) acc += L.phi[q][n] * L.UP[n][i];
                    L.up[q][i] = acc;
                }
            }
        }
        __syncthreads();
        if (act) {
            for (int ab = lane; ab < 100; ab += 64) {
                const int a = ab / 10, b = ab - 10 * a;
                // two-pass mode needs no slot: blocks of rows this rank does not own are simply never gathered
                const int32_t slot = ebuf ? 0 : slots[(int64_t)ab * nc + c];
                if (slot < 0) continue;   // row not owned
                double blk[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) blk[i][j] = 0.0;
                double gv[3] = {0.0, 0.0, 0.0};
                const bool do_rhs = b == 0;
                const int bsel = b < 4 ? b : 0, asel = a < 4 ? a : 0;
                const double bmask = b < 4 ? 1.0 : 0.0, amask = a < 4 ? 1.0 : 0.0;
                for (int q = 0; q < 14; ++q) {
                    const double wv = NS_QW[q] * vol;
                    const double pa = L.phi[q][a], pb = L.phi[q][b];
                    const double ga[3] = {L.gphi[q][a][0], L.gphi[q][a][1], L.gphi[q][a][2]};
                    const double gb[3] = {L.gphi[q][b][0], L.gphi[q][b][1], L.gphi[q][b][2]};
                    const double u0[3] = {L.u0[q][0], L.u0[q][1], L.u0[q][2]};
                    const double nuq = L.nuq[q];
                    double diag = nuq * (ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2]) + P.inv_dt * pa * pb;
                    if (P.convection) {
                        const double av[3] = {u0[0] - P.wm[0], u0[1] - P.wm[1], u0[2] - P.wm[2]};
                        const double agb = av[0] * gb[0] + av[1] * gb[1] + av[2] * gb[2];
                        diag += pa * agb;
                        if (P.g2) {
                            const double aga = av[0] * ga[0] + av[1] * ga[1] + av[2] * ga[2];
                            diag -= ns_g2_delta(P, hcell, av[0] * av[0] + av[1] * av[1] + av[2] * av[2]) * aga * agb;
                        }
                    }
                    const bool full = P.convection && P.newton;
                    // (the kernel is instruction-bound - counters in DESIGN.md section 3 -: the weight goes into the factors once,
                    // the nine viscous terms are 3 + 9 operations instead of 27, the Newton term 2 + 9)
                    const double wd = wv * diag, wn = wv * nuq;
                    const double wga[3] = {wn * ga[0], wn * ga[1], wn * ga[2]};
                    const double wpp = full ? wv * pa * pb : 0.0;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        blk[i][i] += wd;
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            blk[i][j] = fma(wga[j], gb[i], blk[i][j]);
                            if (full) blk[i][j] = fma(wpp, L.gu0[q][3 * i + j], blk[i][j]);
                        }
                    }
                    // pressure trial function psi_b = lambda_b (b < 4) and continuity test function psi_a (a < 4): branch-free (a
                    // zero factor for the edge nodes) - every lane-divergent branch inside this loop is an exec-mask sequence of
                    // scalar instructions, 14 times per block
                    const double wr = wv * P.inv_rho;
                    const double wpb = wr * (NS_QP[q][bsel] * bmask), wpa = wr * (NS_QP[q][asel] * amask);
#pragma unroll
                    for (int i = 0; i < 3; ++i) blk[i][3] = fma(-wpb, ga[i], blk[i][3]);
#pragma unroll
                    for (int j = 0; j < 3; ++j) blk[3][j] = fma(wpa, gb[j], blk[3][j]);
                    if (do_rhs) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            double v = P.f[i];
                            if (full) v += L.gu0[q][3 * i] * u0[0] + L.gu0[q][3 * i + 1] * u0[1] + L.gu0[q][3 * i + 2] * u0[2];
                            if (has_prev) v += P.inv_dt * L.up[q][i];
                            gv[i] += wv * pa * v;
                        }
                    }
                }
                if (ebuf) {      // element block -> buffer (128 B per lane, contiguous over the wave); summed by k_ns_gather
                    double2* __restrict__ out = reinterpret_cast<double2*>(ebuf + ((int64_t)c * 100 + ab) * 16);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        out[2 * i] = make_double2(blk[i][0], blk[i][1]);
                        out[2 * i + 1] = make_double2(blk[i][2], blk[i][3]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (i == 3 && j == 3) continue;
                            if (j == 3 && b >= 4) continue;
                            if (i == 3 && a >= 4) continue;
                            atomicAdd(&val[(int64_t)(i * 4 + j) * plane + slot], blk[i][j]);
                        }
                }
                if (do_rhs) {
                    const int64_t node = L.nd[a];
                    if (node < n_rows) {      // owned row
#pragma unroll
                        for (int i = 0; i < 3; ++i) atomicAdd(&g[4 * node + i], gv[i]);
                    }
                }
            }
        }
        __syncthreads();
    }
}


// ---- Taylor-Hood on TRIANGLES (the reference's own CFD example is 2-D: examples/test_cfd_solver.py:83) -------------------
// The class is dimension-free upstream (CoupledNavierStokesSolver.py:84-102, 288-381).  Here the 2-D system keeps the block-4
// layout: (u_x, u_y, -, p) per CG2 node with the third slot a dummy unknown (unit row, zero coupling), so the operator product,
// the block preconditioner and FGMRES above serve it unchanged; a 2-D mesh is small, the unused planes cost nothing that matters.
// Element: 6 P2 nodes (3 vertices, then the UFC edges e0=(1,2) e1=(0,2) e2=(0,1)), Radon's 7-point degree-5 rule (exact for
// the P2*P2*P1 convection integrand), one thread per (cell, a, b); the 4x4 element block goes to the element buffer and the
// stored blocks are summed by k_ns_gather in a fixed order (no atomics on the matrix).
__device__ const double NS_TRI_QP[7][3] = {
    {1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0},
    {0.797426985353087, 0.101286507323456, 0.101286507323456},
    {0.101286507323456, 0.797426985353087, 0.101286507323456},
    {0.101286507323456, 0.101286507323456, 0.797426985353087},
    {0.059715871789770, 0.470142064105115, 0.470142064105115},
    {0.470142064105115, 0.059715871789770, 0.470142064105115},
    {0.470142064105115, 0.470142064105115, 0.059715871789770}};
__device__ const double NS_TRI_QW[7] = {0.225, 0.125939180544827, 0.125939180544827, 0.125939180544827,
                                        0.132394152788506, 0.132394152788506, 0.132394152788506};
__device__ const int NS_TRI_EI[3] = {1, 0, 0};
__device__ const int NS_TRI_EJ[3] = {2, 2, 1};

__device__ __forceinline__ double sel3(int i, double a, double b, double c) { return i == 0 ? a : (i == 1 ? b : c); }

// value and physical gradient of P2 basis function n of a triangle at barycentric point l
__device__ __forceinline__ void p2tri_eval(int n, const double l[3], const double gl[3][2], double* phi, double g[2]) {
    if (n < 3) {
        const double li = sel3(n, l[0], l[1], l[2]);
        const double d = 4.0 * li - 1.0;
        *phi = li * (2.0 * li - 1.0);
        g[0] = d * sel3(n, gl[0][0], gl[1][0], gl[2][0]);
        g[1] = d * sel3(n, gl[0][1], gl[1][1], gl[2][1]);
    } else {
        const int i = NS_TRI_EI[n - 3], j = NS_TRI_EJ[n - 3];
        const double li = sel3(i, l[0], l[1], l[2]), lj = sel3(j, l[0], l[1], l[2]);
        *phi = 4.0 * li * lj;
        g[0] = 4.0 * (li * sel3(j, gl[0][0], gl[1][0], gl[2][0]) + lj * sel3(i, gl[0][0], gl[1][0], gl[2][0]));
        g[1] = 4.0 * (li * sel3(j, gl[0][1], gl[1][1], gl[2][1]) + lj * sel3(i, gl[0][1], gl[1][1], gl[2][1]));
    }
}

// gradients of the barycentric coordinates of triangle c, its area and (G2) h = 2 * circumradius = |e0||e1||e2| / (2 area)
__device__ __forceinline__ void ns_tri_geometry(const double* __restrict__ xyz, const int32_t* __restrict__ cells, int64_t c,
                                                double gl[3][2], double* area, double* hcell) {
    double X[3][2];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const int64_t vx = cells[c * 4 + v];
        X[v][0] = xyz[4 * vx]; X[v][1] = xyz[4 * vx + 1];
    }
    const double ax = X[1][0] - X[0][0], ay = X[1][1] - X[0][1], bx = X[2][0] - X[0][0], by = X[2][1] - X[0][1];
    const double det = ax * by - bx * ay;
    const double idet = 1.0 / det;
    gl[1][0] = by * idet;  gl[1][1] = -bx * idet;
    gl[2][0] = -ay * idet; gl[2][1] = ax * idet;
    gl[0][0] = -(gl[1][0] + gl[2][0]); gl[0][1] = -(gl[1][1] + gl[2][1]);
    *area = 0.5 * fabs(det);
    if (hcell) {
        const double cx = X[2][0] - X[1][0], cy = X[2][1] - X[1][1];
        *hcell = sqrt((ax * ax + ay * ay) * (bx * bx + by * by) * (cx * cx + cy * cy)) / (2.0 * *area);
    }
}

__global__ void __launch_bounds__(FS_BLOCK) k_assemble_ns_tri(const double* __restrict__ xyz, const int32_t* __restrict__ cells,
                                                              const int32_t* __restrict__ cell_dofs, int64_t nc, int64_t n_rows,
                                                              const double* __restrict__ w0, const double* __restrict__ wprev,
                                                              ns_params P, double* __restrict__ ebuf, double* __restrict__ g) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool has_prev = wprev != nullptr && P.inv_dt != 0.0;
    for (; t < nc * 36; t += stride) {
        const int64_t c = t / 36;
        const int ab = (int)(t - c * 36);
        const int a = ab / 6, b = ab - a * 6;
        int32_t nd[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) nd[n] = cell_dofs[c * 6 + n];
        double gl[3][2], area, hcell = 0.0;
        ns_tri_geometry(xyz, cells, c, gl, &area, P.g2 ? &hcell : nullptr);
        double U0[6][2], UP[6][2];
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            U0[n][0] = P.convection ? w0[4 * (int64_t)nd[n]] : 0.0;
            U0[n][1] = P.convection ? w0[4 * (int64_t)nd[n] + 1] : 0.0;
            UP[n][0] = has_prev ? wprev[4 * (int64_t)nd[n]] : 0.0;
            UP[n][1] = has_prev ? wprev[4 * (int64_t)nd[n] + 1] : 0.0;
        }
        double P0[3] = {0.0, 0.0, 0.0}, T0[3] = {0.0, 0.0, 0.0};
        if (P.V.kind) {
#pragma unroll
            for (int v = 0; v < 3; ++v) P0[v] = w0[4 * (int64_t)nd[v] + 3];
            if (P.V.kind == 2)
#pragma unroll
                for (int v = 0; v < 3; ++v) T0[v] = P.V.T[nd[v]];
        }
        double blk[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) blk[i][j] = 0.0;
        double gv[2] = {0.0, 0.0};
        const bool do_rhs = b == 0;
        const bool full = P.convection && P.newton;
        for (int q = 0; q < 7; ++q) {
            const double l[3] = {NS_TRI_QP[q][0], NS_TRI_QP[q][1], NS_TRI_QP[q][2]};
            const double wv = NS_TRI_QW[q] * area;
            const double nuq = fs_viscosity(P.V, P.nu, l[0] * P0[0] + l[1] * P0[1] + l[2] * P0[2], l[0] * T0[0] + l[1] * T0[1] + l[2] * T0[2]);
            double pa, pb, ga[2], gb[2];
            p2tri_eval(a, l, gl, &pa, ga);
            p2tri_eval(b, l, gl, &pb, gb);
            double diag = nuq * (ga[0] * gb[0] + ga[1] * gb[1]) + P.inv_dt * pa * pb;
            double u0[2] = {0.0, 0.0}, gu0[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, up[2] = {0.0, 0.0};
#pragma unroll
            for (int n = 0; n < 6; ++n) {
                double pn, gn[2];
                p2tri_eval(n, l, gl, &pn, gn);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    u0[i] += pn * U0[n][i];
                    up[i] += pn * UP[n][i];
                    gu0[i][0] += U0[n][i] * gn[0];
                    gu0[i][1] += U0[n][i] * gn[1];
                }
            }
            if (P.convection) {
                const double av[2] = {u0[0] - P.wm[0], u0[1] - P.wm[1]};
                const double agb = av[0] * gb[0] + av[1] * gb[1];
                diag += pa * agb;
                if (P.g2) {
                    const double aga = av[0] * ga[0] + av[1] * ga[1];
                    diag -= ns_g2_delta(P, hcell, av[0] * av[0] + av[1] * av[1]) * aga * agb;
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                blk[i][i] += wv * diag;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    double v = nuq * ga[j] * gb[i];
                    if (full) v += pa * pb * gu0[i][j];
                    blk[i][j] += wv * v;
                }
            }
            if (b < 3) {   // pressure trial function psi_b = lambda_b
                const double psi = sel3(b, l[0], l[1], l[2]);
                blk[0][3] -= wv * P.inv_rho * psi * ga[0];
                blk[1][3] -= wv * P.inv_rho * psi * ga[1];
            }
            if (a < 3) {   // continuity test function psi_a
                const double psi = sel3(a, l[0], l[1], l[2]);
                blk[3][0] += wv * P.inv_rho * psi * gb[0];
                blk[3][1] += wv * P.inv_rho * psi * gb[1];
            }
            if (do_rhs) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    double v = P.f[i];
                    if (full) v += gu0[i][0] * u0[0] + gu0[i][1] * u0[1];
                    if (has_prev) v += P.inv_dt * up[i];
                    gv[i] += wv * pa * v;
                }
            }
        }
        double2* __restrict__ out = reinterpret_cast<double2*>(ebuf + t * 16);       // source index c*36 + ab = t
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            out[2 * i] = make_double2(blk[i][0], blk[i][1]);
            out[2 * i + 1] = make_double2(blk[i][2], blk[i][3]);
        }
        if (do_rhs && nd[a] < n_rows) {
            atomicAdd(&g[4 * (int64_t)nd[a]], gv[0]);
            atomicAdd(&g[4 * (int64_t)nd[a] + 1], gv[1]);
        }
    }
}

// unit diagonal on the dummy slots of the 2-D layout: u_z of every node, the pressure of edge nodes
__global__ void k_ns_dummy_rows_2d(int64_t nv, int64_t n_nodes, const int64_t* __restrict__ slice_ptr,
                                   const int32_t* __restrict__ sell_col, double* __restrict__ val, int64_t plane) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_nodes; r += stride) {
        const int64_t sp0 = slice_ptr[r >> 6];
        const int width = (int)((slice_ptr[(r >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (r & 63);
        for (int k = 0; k < width; ++k) {
            const int64_t e = base + (int64_t)k * FS_SLICE;
            if (sell_col[e] == (int32_t)r) {
                val[10 * plane + e] = 1.0;
                if (r >= nv) val[15 * plane + e] = 1.0;
                break;
            }
        }
    }
}

// ---- second pass: stored block e = sum of its element blocks (fixed order: deterministic, no atomics) -------------------
__global__ void __launch_bounds__(FS_BLOCK) k_ns_gather(int64_t n_entries, const int32_t* __restrict__ ptr, const int32_t* __restrict__ src,
                                                        const double* __restrict__ ebuf, double* __restrict__ val, int64_t plane) {
    // 8 lanes per stored block: lane w reads the 16 bytes (values 2w, 2w+1) of every element block, so a block is one
    // coalesced 128-byte read
    typedef double v2d __attribute__((ext_vector_type(2)));
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n_entries * 8; t += stride) {
        const int64_t e = t >> 3;
        const int w = (int)(t & 7);
        double a0 = 0.0, a1 = 0.0;
        const int32_t q1 = ptr[e + 1];
        for (int32_t q = ptr[e]; q < q1; ++q) {
            const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(ebuf + (int64_t)src[q] * 16) + w);   // read once
            a0 += v.x;
            a1 += v.y;
        }
        val[(int64_t)(2 * w) * plane + e] = a0;
        val[(int64_t)(2 * w + 1) * plane + e] = a1;
    }
}
// unit diagonal on the dummy pressure slot of edge nodes
__global__ void k_ns_dummy_rows(int64_t nv, int64_t n_nodes, const int64_t* __restrict__ slice_ptr,
                                const int32_t* __restrict__ sell_col, double* __restrict__ val, int64_t plane) {
    int64_t r = nv + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_nodes; r += stride) {
        const int64_t sp0 = slice_ptr[r >> 6];
        const int width = (int)((slice_ptr[(r >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (r & 63);
        for (int k = 0; k < width; ++k) {
            const int64_t e = base + (int64_t)k * FS_SLICE;
            if (sell_col[e] == (int32_t)r) { val[15 * plane + e] = 1.0; break; }
        }
    }
}

// After a periodic fold (fs_matrix_tie_nodes) the dummy pressure slot of a master edge node holds 2 (its own unit
// diagonal plus the slave's); the operator product treats those rows as the identity, so the stored value follows.
void fs_ns_reset_dummy_rows(fs_matrix_s* J, hipStream_t s) {
    fs_space_s* sp = J->space;
    const int64_t nv = sp->mesh->n_owned;
    if (sp->mesh->tdim == 2) {
        hipLaunchKernelGGL(k_ns_dummy_rows_2d, dim3(fs_grid_for(sp->n_nodes_owned)), dim3(FS_BLOCK), 0, s, nv, sp->n_nodes_owned,
                           sp->slice_ptr.p, sp->sell_col.p, J->val.p, sp->sell_entries);
        return;
    }
    if (sp->n_nodes_owned <= nv) return;
    hipLaunchKernelGGL(k_ns_dummy_rows, dim3(fs_grid_for(sp->n_nodes_owned - nv)), dim3(FS_BLOCK), 0, s, nv, sp->n_nodes_owned,
                       sp->slice_ptr.p, sp->sell_col.p, J->val.p, sp->sell_entries);
}

// The law a Taylor-Hood space carries (fs_space_set_viscosity_law), else the (p_ref, exponent) pair of the call
fs_visc_dev fs_space_viscosity(const fs_space_s* sp, double legacy_pref, double legacy_exp) {
    if (sp && sp->visc.kind) return sp->visc;
    fs_visc_dev V;
    if (legacy_pref > 0.0) { V.kind = 1; V.pref = legacy_pref; V.ex = legacy_exp; }
    return V;
}

extern "C" int fs_space_set_viscosity_law(fs_space_t th_space, const fs_viscosity_law* law) {
    FS_REQUIRE(th_space, "fs_space_set_viscosity_law: null space");
    fs_visc_dev V;
    if (law && law->kind) {
        FS_REQUIRE(th_space->degree == 2 && th_space->ncomp == 4, "fs_space_set_viscosity_law: not a Taylor-Hood space");
        FS_REQUIRE((law->kind == 1 || law->kind == 2) && law->pressure_ref > 0.0, "fs_space_set_viscosity_law: kind 1 or 2 with a positive reference pressure");
        V.kind = law->kind;
        V.pref = law->pressure_ref;
        V.ex = law->pressure_exponent;
        if (law->kind == 2) {
            FS_REQUIRE(law->temperature && law->temperature_ref != 0.0 && law->temperature->d.n >= th_space->n_nodes_local,
                       "fs_space_set_viscosity_law: kind 2 needs the CG1 temperature (one value per local node, read at the vertex nodes) and its reference");
            V.cp = law->pressure_coef;
            V.ct = law->temperature_coef;
            V.tref = law->temperature_ref;
            V.T = law->temperature->d.p;
        }
    }
    th_space->visc = V;
    return FS_OK;
}

extern "C" int fs_assemble_navier_stokes(fs_matrix_t J, fs_vector_t g, fs_vector_t w0, fs_vector_t w_prev,
                                         const fs_ns_form* form) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(J && g && form, "fs_assemble_navier_stokes: null pointer");
    fs_space_s* sp = J->space;
    FS_REQUIRE(sp->degree == 2 && sp->ncomp == 4 && J->bs == 4, "fs_assemble_navier_stokes: the matrix must live on a 4-component CG2 space");
    FS_REQUIRE(sp->slots.p, "fs_assemble_navier_stokes: the space has no slot table");
    FS_REQUIRE(!form->convection || w0, "fs_assemble_navier_stokes: convection needs the state w0");
    FS_REQUIRE(g->d.n >= sp->n_dofs_owned && (!w0 || w0->d.n >= sp->n_dofs_local) && (!w_prev || w_prev->d.n >= sp->n_dofs_local),
               "fs_assemble_navier_stokes: vector too short");
    FS_REQUIRE(form->density > 0.0 && form->inv_dt >= 0.0, "fs_assemble_navier_stokes: bad density / time step");
    hipStream_t s = fs_rt().stream;
    fs_mesh_s* m = sp->mesh;
    ns_params P;
    P.nu = form->kinematic_viscosity;
    P.inv_rho = 1.0 / form->density;
    P.inv_dt = form->inv_dt;
    for (int i = 0; i < 3; ++i) P.f[i] = form->body_force[i];
    for (int i = 0; i < 3; ++i) P.wm[i] = form->convection ? form->mesh_velocity[i] : 0.0;
    P.convection = form->convection ? 1 : 0;
    P.newton = form->newton ? 1 : 0;
    FS_REQUIRE(form->g2_mode >= 0 && form->g2_mode <= 2 && (form->g2_mode == 0 || form->convection), "fs_assemble_navier_stokes: bad G2 mode");
    P.g2 = form->g2_mode;
    P.g2_kappa = form->g2_kappa1;
    FS_REQUIRE(form->viscosity_pressure_ref >= 0.0 && (form->viscosity_pressure_ref == 0.0 || w0),
               "fs_assemble_navier_stokes: the pressure-dependent viscosity needs a positive reference pressure and the state w0");
    P.V = fs_space_viscosity(sp, form->viscosity_pressure_ref, form->viscosity_pressure_exponent);
    FS_REQUIRE(P.V.kind == 0 || w0, "fs_assemble_navier_stokes: a non-Newtonian law needs the state w0");
    FS_HIP(hipMemsetAsync(g->d.p, 0, (size_t)sp->n_dofs_owned * sizeof(double), s));
    if (m->tdim == 2) {
        // triangles: element blocks -> buffer -> one sum per stored block (always two-pass), dummy u_z / edge-pressure rows
        if (!sp->gmap_ptr.p) FS_CHECK(fs_space_build_gather_map(sp, s));
        if (sp->elem_buf.n != m->nc * 36 * 16) FS_CHECK(sp->elem_buf.alloc(m->nc * 36 * 16));
        hipLaunchKernelGGL(k_assemble_ns_tri, dim3(fs_grid_for(m->nc * 36, FS_BLOCK, 1 << 16)), dim3(FS_BLOCK), 0, s, m->xyz.p, m->cells.p,
                           sp->cell_dofs, m->nc, sp->n_nodes_owned, w0 ? w0->d.p : nullptr, w_prev ? w_prev->d.p : nullptr, P,
                           sp->elem_buf.p, g->d.p);
        hipLaunchKernelGGL(k_ns_gather, dim3(fs_grid_for(sp->sell_entries * 8, FS_BLOCK, 1 << 18)), dim3(FS_BLOCK), 0, s, sp->sell_entries,
                           sp->gmap_ptr.p, sp->gmap_src.p, sp->elem_buf.p, J->val.p, sp->sell_entries);
        J->taylor_hood = true;
        hipLaunchKernelGGL(k_ns_dummy_rows_2d, dim3(fs_grid_for(sp->n_nodes_owned)), dim3(FS_BLOCK), 0, s, m->n_owned, sp->n_nodes_owned,
                           sp->slice_ptr.p, sp->sell_col.p, J->val.p, sp->sell_entries);
        FS_KERNEL_CHECK();
        FS_HIP(hipStreamSynchronize(s));
        return FS_OK;
    }
    // Element blocks -> buffer -> one sum per stored block: 25 ms with 544 M device-scope fp64 atomics became 7 ms
    // (MI355X, configs[4]) and the matrix is bit-reproducible.  FS_NS_ASSEMBLE=atomic / pair selects the one-pass
    // kernels (no 6 GB element buffer).
    static const char* mode_env = getenv("FS_NS_ASSEMBLE");
    const bool two_pass = !mode_env || (mode_env[0] != 'a' && mode_env[0] != 'p');
    if (two_pass) {
        if (!sp->gmap_ptr.p) FS_CHECK(fs_space_build_gather_map(sp, s));
        if (!sp->elem_buf.p) FS_CHECK(sp->elem_buf.alloc(m->nc * 1600));
    } else {
        FS_CHECK(J->val.zero(s));
    }
    if (!two_pass && mode_env[0] == 'p') {
        const int grid = fs_grid_for(m->nc * 100, FS_BLOCK, 1 << 16);
        hipLaunchKernelGGL(k_assemble_ns, dim3(grid), dim3(FS_BLOCK), 0, s, m->xyz.p, m->cells.p, sp->cell_dofs, m->nc, sp->slots.p,
                           w0 ? w0->d.p : nullptr, w_prev ? w_prev->d.p : nullptr, P, J->val.p, sp->sell_entries, g->d.p);
    } else {
        const int grid = (int)std::min<int64_t>((m->nc + NS_WPB - 1) / NS_WPB, 1 << 16);
        hipLaunchKernelGGL(k_assemble_ns_wave, dim3(std::max(grid, 1)), dim3(FS_BLOCK), 0, s, m->xyz.p, m->cells.p, sp->cell_dofs, m->nc,
                           sp->n_nodes_owned, sp->slots.p,
                           w0 ? w0->d.p : nullptr, w_prev ? w_prev->d.p : nullptr, P, J->val.p, sp->sell_entries,
                           two_pass ? sp->elem_buf.p : nullptr, g->d.p);
        if (two_pass)
            hipLaunchKernelGGL(k_ns_gather, dim3(fs_grid_for(sp->sell_entries * 8, FS_BLOCK, 1 << 18)), dim3(FS_BLOCK), 0, s, sp->sell_entries,
                               sp->gmap_ptr.p, sp->gmap_src.p, sp->elem_buf.p, J->val.p, sp->sell_entries);
    }
    J->taylor_hood = true;      // lets the operator product skip the structurally empty pressure planes
    hipLaunchKernelGGL(k_ns_dummy_rows, dim3(fs_grid_for(sp->n_nodes_owned - m->n_owned)), dim3(FS_BLOCK), 0, s, m->n_owned,
                       sp->n_nodes_owned, sp->slice_ptr.p, sp->sell_col.p, J->val.p, sp->sell_entries);
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

// ---- pressure boundaries: F += inner(p_b n, v) ds - nu inner((grad(u) + grad(u)^T) n, v) ds -----------------------
// (CoupledNavierStokesSolver.py:449-453 pressure Dirichlet, :459-460 pressure 'farfield')
// 6-point degree-4 triangle rule; the facet's P2 nodes as cell-local node indices, by opposite vertex
__device__ const double NS_TQ[6][3] = {{0.108103018168070, 0.445948490915965, 0.445948490915965},
                                       {0.445948490915965, 0.108103018168070, 0.445948490915965},
                                       {0.445948490915965, 0.445948490915965, 0.108103018168070},
                                       {0.816847572980459, 0.091576213509771, 0.091576213509771},
                                       {0.091576213509771, 0.816847572980459, 0.091576213509771},
                                       {0.091576213509771, 0.091576213509771, 0.816847572980459}};
__device__ const double NS_TW[6] = {0.223381589678011, 0.223381589678011, 0.223381589678011,
                                    0.109951743655322, 0.109951743655322, 0.109951743655322};
__device__ const int NS_FACE_NODES[4][6] = {{1, 2, 3, 4, 5, 6}, {0, 2, 3, 4, 7, 8}, {0, 1, 3, 5, 7, 9}, {0, 1, 2, 6, 8, 9}};

// thread t = (facet, facet node al, cell node b)
__global__ void k_ns_pressure_boundary(int64_t nf, const int32_t* __restrict__ facet_cell, const int32_t* __restrict__ facet_opp,
                                       const double* __restrict__ facet_value, double nu0, const double* __restrict__ xyz,
                                       const int32_t* __restrict__ cells, const int32_t* __restrict__ cell_dofs, int64_t nc,
                                       const int32_t* __restrict__ slots,
                                       double* __restrict__ val, int64_t plane, double* __restrict__ g,
                                       const double* __restrict__ w0, fs_visc_dev V, int vstride) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nf * 60; t += stride) {
        const int64_t f = t / 60;
        const int r = (int)(t - f * 60);
        const int al = r / 10, b = r - al * 10;
        const int64_t c = facet_cell[f];
        const int o = facet_opp[f];
        const int a = NS_FACE_NODES[o][al];
        int32_t nd[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) nd[v] = cells[c * 4 + v];     // vertex ids (geometry), not node ids
        double X[4][3];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            X[v][0] = xyz[4 * (int64_t)nd[v]]; X[v][1] = xyz[4 * (int64_t)nd[v] + 1]; X[v][2] = xyz[4 * (int64_t)nd[v] + 2];
        }
        const double e1[3] = {X[1][0] - X[0][0], X[1][1] - X[0][1], X[1][2] - X[0][2]};
        const double e2[3] = {X[2][0] - X[0][0], X[2][1] - X[0][1], X[2][2] - X[0][2]};
        const double e3[3] = {X[3][0] - X[0][0], X[3][1] - X[0][1], X[3][2] - X[0][2]};
        const double c23[3] = {e2[1] * e3[2] - e2[2] * e3[1], e2[2] * e3[0] - e2[0] * e3[2], e2[0] * e3[1] - e2[1] * e3[0]};
        const double c31[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
        const double c12[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double det = e1[0] * c23[0] + e1[1] * c23[1] + e1[2] * c23[2];
        const double idet = 1.0 / det;
        double gl[4][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gl[1][k] = c23[k] * idet;
            gl[2][k] = c31[k] * idet;
            gl[3][k] = c12[k] * idet;
            gl[0][k] = -(gl[1][k] + gl[2][k] + gl[3][k]);
        }
        const double vol = fabs(det) * (1.0 / 6.0);
        double go[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) go[k] = sel4(o, gl[0][k], gl[1][k], gl[2][k], gl[3][k]);
        const double gnorm = sqrt(go[0] * go[0] + go[1] * go[1] + go[2] * go[2]);
        const double n[3] = {-go[0] / gnorm, -go[1] / gnorm, -go[2] / gnorm};   // outward
        const double area = 3.0 * vol * gnorm;
        double blk[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        double gv[3] = {0.0, 0.0, 0.0};
        // boundary pressure: one value per facet, or (vstride 3) its values at the facet's vertices - the P1 interpolant DOLFIN
        // evaluates a degree-1 Expression with - in the order of the cell's local vertices
        double pbv[3] = {0.0, 0.0, 0.0};
        if (facet_value) {
            if (vstride == 3) { pbv[0] = facet_value[3 * f]; pbv[1] = facet_value[3 * f + 1]; pbv[2] = facet_value[3 * f + 2]; }
            else pbv[0] = pbv[1] = pbv[2] = facet_value[f];
        }
        double P0[4] = {0.0, 0.0, 0.0, 0.0}, T0[4] = {0.0, 0.0, 0.0, 0.0};
        if (V.kind) {
#pragma unroll
            for (int v = 0; v < 4; ++v) P0[v] = w0[4 * (int64_t)cell_dofs[c * 10 + v] + 3];
            if (V.kind == 2)
#pragma unroll
                for (int v = 0; v < 4; ++v) T0[v] = V.T[cell_dofs[c * 10 + v]];
        }
        for (int q = 0; q < 6; ++q) {
            double l[4];
            int kk = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) l[v] = (v == o) ? 0.0 : NS_TQ[q][kk++];
            const double wv = NS_TW[q] * area;
            const double pb = NS_TQ[q][0] * pbv[0] + NS_TQ[q][1] * pbv[1] + NS_TQ[q][2] * pbv[2];
            const double nu = fs_viscosity(V, nu0, l[0] * P0[0] + l[1] * P0[1] + l[2] * P0[2] + l[3] * P0[3],
                                           l[0] * T0[0] + l[1] * T0[1] + l[2] * T0[2] + l[3] * T0[3]);
            double pa, pbf, ga[3], gb[3];
            p2_eval(a, l, gl, &pa, ga);
            p2_eval(b, l, gl, &pbf, gb);
            const double gn = gb[0] * n[0] + gb[1] * n[1] + gb[2] * n[2];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) blk[i][j] -= nu * wv * pa * ((i == j ? gn : 0.0) + gb[i] * n[j]);
                gv[i] -= wv * pb * n[i] * pa;
            }
        }
        const int32_t slot = slots[(int64_t)(a * 10 + b) * nc + c];
        if (slot < 0) continue;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) atomicAdd(&val[(int64_t)(i * 4 + j) * plane + slot], blk[i][j]);
        if (b == 0 && facet_value) {
            const int32_t node = cell_dofs[c * 10 + a];
#pragma unroll
            for (int i = 0; i < 3; ++i) atomicAdd(&g[4 * (int64_t)node + i], gv[i]);
        }
    }
}

// the same on boundary EDGES of a triangle mesh: the edge opposite local vertex o carries the P2 nodes (i, j, 3 + o) with
// i < j the other two vertices; 3-point Gauss-Legendre rule (degree 5); thread t = (facet, facet node al, cell node b)
__global__ void k_ns_pressure_boundary_tri(int64_t nf, const int32_t* __restrict__ facet_cell, const int32_t* __restrict__ facet_opp,
                                           const double* __restrict__ facet_value, double nu0, const double* __restrict__ xyz,
                                           const int32_t* __restrict__ cells, const int32_t* __restrict__ cell_dofs, int64_t nc,
                                           const int32_t* __restrict__ slots, double* __restrict__ val, int64_t plane,
                                           double* __restrict__ g, const double* __restrict__ w0, fs_visc_dev V,
                                           int vstride) {
    const double GS[3] = {0.5 - 0.5 * 0.7745966692414834, 0.5, 0.5 + 0.5 * 0.7745966692414834};
    const double GW[3] = {5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0};
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nf * 18; t += stride) {
        const int64_t f = t / 18;
        const int r = (int)(t - f * 18);
        const int al = r / 6, b = r - al * 6;
        const int64_t c = facet_cell[f];
        const int o = facet_opp[f];
        const int vi = o == 0 ? 1 : 0, vj = o == 2 ? 1 : 2;          // the edge's vertices, ascending local index
        const int a = al == 0 ? vi : (al == 1 ? vj : 3 + o);
        double gl[3][2], area;
        ns_tri_geometry(xyz, cells, c, gl, &area, nullptr);
        const double gnorm = sqrt(gl[o][0] * gl[o][0] + gl[o][1] * gl[o][1]);
        const double n[2] = {-gl[o][0] / gnorm, -gl[o][1] / gnorm};     // outward
        const double len = 2.0 * area * gnorm;
        double pbv[2] = {0.0, 0.0};
        if (facet_value) {
            if (vstride == 2) { pbv[0] = facet_value[2 * f]; pbv[1] = facet_value[2 * f + 1]; }
            else pbv[0] = pbv[1] = facet_value[f];
        }
        double P0[3] = {0.0, 0.0, 0.0}, T0[3] = {0.0, 0.0, 0.0};
        if (V.kind) {
#pragma unroll
            for (int v = 0; v < 3; ++v) P0[v] = w0[4 * (int64_t)cell_dofs[c * 6 + v] + 3];
            if (V.kind == 2)
#pragma unroll
                for (int v = 0; v < 3; ++v) T0[v] = V.T[cell_dofs[c * 6 + v]];
        }
        double blk[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
        double gv[2] = {0.0, 0.0};
        for (int q = 0; q < 3; ++q) {
            double l[3] = {0.0, 0.0, 0.0};
            l[vi] = 1.0 - GS[q];
            l[vj] = GS[q];
            const double wv = GW[q] * len;
            const double pb = (1.0 - GS[q]) * pbv[0] + GS[q] * pbv[1];
            const double nu = fs_viscosity(V, nu0, l[0] * P0[0] + l[1] * P0[1] + l[2] * P0[2], l[0] * T0[0] + l[1] * T0[1] + l[2] * T0[2]);
            double pa, pbf, ga[2], gb[2];
            p2tri_eval(a, l, gl, &pa, ga);
            p2tri_eval(b, l, gl, &pbf, gb);
            const double gn = gb[0] * n[0] + gb[1] * n[1];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) blk[i][j] -= nu * wv * pa * ((i == j ? gn : 0.0) + gb[i] * n[j]);
                gv[i] -= wv * pb * n[i] * pa;
            }
        }
        const int32_t slot = slots[(int64_t)(a * 6 + b) * nc + c];
        if (slot < 0) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) atomicAdd(&val[(int64_t)(i * 4 + j) * plane + slot], blk[i][j]);
        if (b == 0 && facet_value) {
            const int32_t node = cell_dofs[c * 6 + a];
            atomicAdd(&g[4 * (int64_t)node], gv[0]);
            atomicAdd(&g[4 * (int64_t)node + 1], gv[1]);
        }
    }
}

extern "C" int fs_assemble_ns_pressure_boundary_nn(fs_matrix_t J, fs_vector_t g, int64_t n_facets, const int32_t* facet_cell,
                                                   const int32_t* facet_opposite, const double* facet_value,
                                                   double kinematic_viscosity, fs_vector_t w0, double nn_pref, double nn_exp,
                                                   int values_per_facet);
extern "C" int fs_assemble_ns_pressure_boundary(fs_matrix_t J, fs_vector_t g, int64_t n_facets, const int32_t* facet_cell,
                                                const int32_t* facet_opposite, const double* facet_value,
                                                double kinematic_viscosity) {
    return fs_assemble_ns_pressure_boundary_nn(J, g, n_facets, facet_cell, facet_opposite, facet_value, kinematic_viscosity, nullptr, 0.0, 0.0, 1);
}
extern "C" int fs_assemble_ns_pressure_boundary_nn(fs_matrix_t J, fs_vector_t g, int64_t n_facets, const int32_t* facet_cell,
                                                   const int32_t* facet_opposite, const double* facet_value,
                                                   double kinematic_viscosity, fs_vector_t w0, double nn_pref, double nn_exp,
                                                   int values_per_facet) {
    FS_CHECK(fs_require_init());
    FS_REQUIRE(J && J->space && J->space->mesh, "fs_assemble_ns_pressure_boundary: null matrix");
    const int fverts = J->space->mesh->tdim;      // vertices of a boundary facet: 3 (triangle) or 2 (edge of a 2-D mesh)
    FS_REQUIRE(values_per_facet == 1 || values_per_facet == fverts, "fs_assemble_ns_pressure_boundary: values_per_facet must be 1 or %d", fverts);
    const fs_visc_dev V = fs_space_viscosity(J->space, nn_pref, nn_exp);
    FS_REQUIRE(nn_pref >= 0.0 && (V.kind == 0 || (w0 && J && w0->d.n >= J->space->n_dofs_local)),
               "fs_assemble_ns_pressure_boundary: the pressure-dependent viscosity needs a positive reference pressure and the state w0");
    FS_REQUIRE(J && g && n_facets >= 0 && (n_facets == 0 || (facet_cell && facet_opposite)), "fs_assemble_ns_pressure_boundary: bad arguments");
    fs_space_s* sp = J->space;
    FS_REQUIRE(sp->degree == 2 && sp->ncomp == 4 && sp->slots.p, "fs_assemble_ns_pressure_boundary: not a Taylor-Hood block matrix");
    if (n_facets == 0) return FS_OK;
    fs_mesh_s* m = sp->mesh;
    for (int64_t i = 0; i < n_facets; ++i)
        FS_REQUIRE(facet_cell[i] >= 0 && facet_cell[i] < m->nc && facet_opposite[i] >= 0 && facet_opposite[i] <= fverts,
                   "fs_assemble_ns_pressure_boundary: facet %lld names cell %d / local vertex %d", (long long)i, facet_cell[i], facet_opposite[i]);
    hipStream_t s = fs_rt().stream;
    dbuf<int32_t> dc, dop;
    dbuf<double> dv;
    FS_CHECK(dc.alloc(n_facets));
    FS_CHECK(dop.alloc(n_facets));
    FS_CHECK(dc.upload(facet_cell, n_facets, s));
    FS_CHECK(dop.upload(facet_opposite, n_facets, s));
    if (facet_value) {
        FS_CHECK(dv.alloc(n_facets * values_per_facet));
        FS_CHECK(dv.upload(facet_value, n_facets * values_per_facet, s));
    }
    if (fverts == 2)
        hipLaunchKernelGGL(k_ns_pressure_boundary_tri, dim3(fs_grid_for(n_facets * 18)), dim3(FS_BLOCK), 0, s, n_facets, dc.p, dop.p,
                           facet_value ? dv.p : (const double*)nullptr, kinematic_viscosity, m->xyz.p, m->cells.p, sp->cell_dofs, m->nc,
                           sp->slots.p, J->val.p, sp->sell_entries, g->d.p, V.kind ? (const double*)w0->d.p : (const double*)nullptr,
                           V, values_per_facet);
    else
    hipLaunchKernelGGL(k_ns_pressure_boundary, dim3(fs_grid_for(n_facets * 60)), dim3(FS_BLOCK), 0, s, n_facets, dc.p, dop.p,
                       facet_value ? dv.p : (const double*)nullptr, kinematic_viscosity, m->xyz.p, m->cells.p, sp->cell_dofs, m->nc,
                       sp->slots.p, J->val.p, sp->sell_entries, g->d.p, V.kind ? (const double*)w0->d.p : (const double*)nullptr,
                       V, values_per_facet);
    FS_KERNEL_CHECK();
    FS_HIP(hipStreamSynchronize(s));
    return FS_OK;
}

// ---- FGMRES with the block-triangular preconditioner ---------------------------------------------------------
__global__ void k_sd_diag(int64_t n_nodes, const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ sell_col,
                          const double* __restrict__ val, int64_t plane, double* __restrict__ dinv) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_nodes; r += stride) {
        const int64_t sp0 = slice_ptr[r >> 6];
        const int width = (int)((slice_ptr[(r >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (r & 63);
        for (int k = 0; k < width; ++k) {
            const int64_t e = base + (int64_t)k * FS_SLICE;
            if (sell_col[e] != (int32_t)r) continue;
            for (int i = 0; i < 4; ++i) {
                const double d = val[(int64_t)(i * 4 + i) * plane + e];
                dinv[r * 4 + i] = d != 0.0 ? 1.0 / d : 0.0;     // pressure rows have no diagonal: 0
            }
            break;
        }
    }
}
// zu = (ADD ? zu : 0) + dinv * (r - t) on velocity components; pressure components of zu are set to 0
template <bool FIRST>
__global__ void k_sd_vel_sweep(int64_t n, const double* __restrict__ dinv, const double* __restrict__ r,
                               const double* __restrict__ t, double* __restrict__ z) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if ((i & 3) == 3) { z[i] = 0.0; continue; }
        z[i] = FIRST ? dinv[i] * r[i] : z[i] + dinv[i] * (r[i] - t[i]);
    }
}
// Chebyshev step on the velocity block (Jacobi-scaled): d = c2 d + c1 dinv (r - t), z += d; FIRST: d = c1 dinv r, z = d.
// Pressure components of z and d stay 0.
template <bool FIRST>
__global__ void k_sd_vel_cheb(int64_t n, const double* __restrict__ dinv, const double* __restrict__ r,
                              const double* __restrict__ t, double* __restrict__ d, double* __restrict__ z, double c1, double c2) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if ((i & 3) == 3) { z[i] = 0.0; d[i] = 0.0; continue; }
        const double v = FIRST ? c1 * dinv[i] * r[i] : c2 * d[i] + c1 * dinv[i] * (r[i] - t[i]);
        d[i] = v;
        z[i] = FIRST ? v : z[i] + v;
    }
}
// v <- dinv * t on velocity components, 0 on pressure components (power iteration of D^-1 A)
__global__ void k_sd_vel_scale(int64_t n, const double* __restrict__ dinv, const double* __restrict__ t, double* __restrict__ v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) v[i] = (i & 3) == 3 ? 0.0 : dinv[i] * t[i];
}
__global__ void k_sd_seed(int64_t n, double* __restrict__ v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t x = (uint32_t)i * 2654435761U + 12345U;
        x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
        v[i] = (i & 3) == 3 ? 0.0 : (double)(x & 0xffffff) / 8388608.0 - 1.0;
    }
}
// rp[v] = r[4v+3] - (D z_u)[v]: only the continuity rows of the block matrix are needed here, i.e. the three value
// planes (3,0..2) of the vertex-node rows (numbered first) - 2 % of the traffic of a full SpMV.  One thread per row.
__global__ void __launch_bounds__(FS_BLOCK) k_sd_pressure_rows(int64_t nv, const int64_t* __restrict__ slice_ptr,
                                                               const int32_t* __restrict__ sell_col, const double* __restrict__ val,
                                                               int64_t plane, const double* __restrict__ z,
                                                               const double* __restrict__ r, double* __restrict__ rp) {
    // One workgroup per slice of 64 vertex rows, lane = row (every load of a wave is one contiguous 512-byte line of
    // the slice), the entries of the up to 65-block-wide rows dealt round-robin to the four waves, partial sums through
    // LDS: a thread per row walks 65 dependent entries on a third of the chip (75 us at configs[4]), 16 lanes per row read
    // 32 bytes per line (65 us).
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t n_sl = (nv + 63) >> 6;
    for (int64_t s = blockIdx.x; s < n_sl; s += gridDim.x) {
        const int64_t v = s * 64 + lane;
        const int64_t sp0 = slice_ptr[s];
        const int width = (int)((slice_ptr[s + 1] - sp0) >> 6);
        const int64_t base = sp0 + lane;
        double acc = 0.0;
        for (int k = w; k < width; k += 4) {
            const int64_t e = base + (int64_t)k * FS_SLICE;
            const int32_t c = sell_col[e];
            if (c < 0) continue;
            const double* zc = z + 4 * (int64_t)c;
            acc += val[12 * plane + e] * zc[0] + val[13 * plane + e] * zc[1] + val[14 * plane + e] * zc[2];
        }
        part[w][lane] = acc;
        __syncthreads();
        if (w == 0 && v < nv) rp[v] = r[4 * v + 3] - ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
        __syncthreads();
    }
}
// rp[v] = r[4v+3] - t[4v+3]
__global__ void k_sd_gather_p(int64_t nv, const double* __restrict__ r, const double* __restrict__ t, double* __restrict__ rp) {
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; v < nv; v += stride) rp[v] = r[4 * v + 3] - t[4 * v + 3];
}
// z[4v+3] = c1*p1[v] + c2*p2[v] for vertices; dummy pressure slots z = r; rows flagged as identity z = r
__global__ void k_sd_scatter_p(int64_t n_nodes, int64_t nv, double c1, const double* __restrict__ p1, double c2,
                               const double* __restrict__ p2, const double* __restrict__ r, const uint8_t* __restrict__ ident,
                               double* __restrict__ z) {
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; v < n_nodes; v += stride) {
        double out;
        if (v >= nv || ident[v]) out = r[4 * v + 3];
        else out = (p1 ? c1 * p1[v] : 0.0) + c2 * p2[v];
        z[4 * v + 3] = out;
    }
}
// pressure rows that are identity rows (Dirichlet pressure): the row has a unit (3,3) diagonal entry
__global__ void k_sd_ident_p(int64_t nv, const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ sell_col,
                             const double* __restrict__ val, int64_t plane, uint8_t* __restrict__ ident) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < nv; r += stride) {
        const int64_t sp0 = slice_ptr[r >> 6];
        const int width = (int)((slice_ptr[(r >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (r & 63);
        uint8_t f = 0;
        for (int k = 0; k < width; ++k) {
            const int64_t e = base + (int64_t)k * FS_SLICE;
            if (sell_col[e] == (int32_t)r) { f = val[15 * plane + e] != 0.0 ? 1 : 0; break; }
        }
        ident[r] = f;
    }
}
__global__ void k_sd_axpy(int64_t n, double a, const double* __restrict__ x, double* __restrict__ y) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] += a * x[i];
}
__global__ void k_sd_scale_to(int64_t n, double a, const double* __restrict__ x, double* __restrict__ y) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = a * x[i];
}
__global__ void k_sd_sub(int64_t n, const double* __restrict__ b, const double* t, double* r) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) r[i] = b[i] - t[i];
}

// Chebyshev step for M x = b with Jacobi: d = c2 d + c1 dinv (b - t), x += d   (FIRST: d = c1 dinv b, x = d)
template <bool FIRST>
__global__ void k_sd_cheb(int64_t n, const double* __restrict__ dinv, const double* __restrict__ b, const double* __restrict__ t,
                          double* __restrict__ d, double* __restrict__ x, double c1, double c2) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const double v = FIRST ? c1 * dinv[i] * b[i] : c2 * d[i] + c1 * dinv[i] * (b[i] - t[i]);
        d[i] = v;
        x[i] = FIRST ? v : x[i] + v;
    }
}
// 1/diag of a scalar SELL matrix
__global__ void k_sd_scalar_dinv(int64_t n_rows, const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ sell_col,
                                 const double* __restrict__ val, double* __restrict__ dinv) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n_rows; r += stride) {
        const int64_t sp0 = slice_ptr[r >> 6];
        const int width = (int)((slice_ptr[(r >> 6) + 1] - sp0) >> 6);
        const int64_t base = sp0 + (r & 63);
        double d = 1.0;
        for (int k = 0; k < width; ++k) {
            const int64_t e = base + (int64_t)k * FS_SLICE;
            if (sell_col[e] == (int32_t)r) { d = val[e]; break; }
        }
        dinv[r] = d != 0.0 ? 1.0 / d : 1.0;
    }
}
// dots[j] partials of w . V_j for j < nvec (one launch; block b writes partial[j*gridDim + b])
// gate (may be null): the launch is a no-op when *gate == 0 - the second Gram-Schmidt pass is enqueued unconditionally
// and the device decides whether it runs, so the host never waits for the first pass.
// Eight basis vectors per sweep over the rows: eight independent load streams per lane and one block reduction for
// the eight sums (a sweep per vector is latency-bound: 20 rows per thread, then a barrier).
#define SD_DOT_GROUP 8
__global__ void __launch_bounds__(FS_BLOCK) k_sd_multi_dot(int64_t n, const double* __restrict__ w, const double* const* __restrict__ V,
                                                           int nvec, int with_self, double* __restrict__ partial,
                                                           const double* __restrict__ gate) {
    __shared__ double lds[SD_DOT_GROUP][FS_BLOCK / 64];
    if (gate && *gate == 0.0) return;
    const int total = nvec + with_self;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int j0 = 0; j0 < total; j0 += SD_DOT_GROUP) {
        const double* vp[SD_DOT_GROUP];
        double acc[SD_DOT_GROUP];
#pragma unroll
        for (int c = 0; c < SD_DOT_GROUP; ++c) {
            vp[c] = j0 + c < nvec ? V[j0 + c] : w;      // the extra "vector" is w itself: ||w||^2
            acc[c] = 0.0;
        }
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            const double wi = w[i];
#pragma unroll
            for (int c = 0; c < SD_DOT_GROUP; ++c) acc[c] += wi * vp[c][i];
        }
#pragma unroll
        for (int c = 0; c < SD_DOT_GROUP; ++c) {
            double t = acc[c];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0) lds[c][wave] = t;
        }
        __syncthreads();
        if (threadIdx.x < SD_DOT_GROUP && j0 + (int)threadIdx.x < total) {
            double t = 0.0;
            for (int q = 0; q < FS_BLOCK / 64; ++q) t += lds[threadIdx.x][q];
            partial[(int64_t)(j0 + threadIdx.x) * gridDim.x + blockIdx.x] = t;
        }
        __syncthreads();
    }
}
// sums[j] = sum_b partial[j*nblk + b]   (one workgroup per j, fixed order)
__global__ void __launch_bounds__(FS_BLOCK) k_sd_multi_sum(const double* __restrict__ partial, int nblk, double* __restrict__ sums,
                                                           const double* __restrict__ gate) {
    __shared__ double lds4[4];
    if (gate && *gate == 0.0) return;
    double acc = 0.0;
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) acc += partial[(int64_t)blockIdx.x * nblk + b];
    const double t = fs_block_sum(acc, lds4);
    if (threadIdx.x == 0) sums[blockIdx.x] = t;
}
// w -= sum_j h[j] V_j
__global__ void k_sd_multi_axpy(int64_t n, double* __restrict__ w, const double* const* __restrict__ V, const double* __restrict__ h,
                                int nvec, const double* __restrict__ gate) {
    if (gate && *gate == 0.0) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        double acc = w[i];
        for (int j = 0; j < nvec; ++j) acc -= h[j] * V[j][i];
        w[i] = acc;
    }
}

// ---- Arnoldi bookkeeping on the device: the Hessenberg column, the Givens rotations and the residual recurrence are
// advanced by single-thread kernels between the vector kernels, so one FGMRES iteration is a fixed launch sequence with
// no host read in it.  ctl = {need second pass, hh^2, 1/hh, |gamma_{k+1}|, hh}.
enum { SD_NEED2 = 0, SD_HH2 = 1, SD_SCALE = 2, SD_RES = 3, SD_HH = 4, SD_CTL = 8 };
// column k of H (stride m) += the projections of this pass; Pythagoras for the new norm; DGKS-type criterion
__global__ void k_sd_hess_pass(const double* __restrict__ hdev, int k, int m, double* __restrict__ H, double* __restrict__ ctl, int pass) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (pass > 0 && ctl[SD_NEED2] == 0.0) return;
    double removed = 0.0;
    for (int j = 0; j <= k; ++j) {
        const double h = hdev[j];
        H[(size_t)j * m + k] = (pass > 0 ? H[(size_t)j * m + k] : 0.0) + h;
        removed += h * h;
    }
    const double before = hdev[k + 1];
    const double hh2 = before - removed;
    ctl[SD_HH2] = hh2;
    if (pass == 0) ctl[SD_NEED2] = hh2 > 0.1 * before ? 0.0 : 1.0;      // eta^2 = 0.1: at most one digit lost to cancellation
}
// host_out (mapped, coherent host memory): {|gamma_{k+1}|, hh, serial}; the serial number is written last, after a
// system-scope fence, and the host spins on it - no runtime call is involved in the hand-over
__global__ void k_sd_givens(int k, int m, double* __restrict__ H, double* __restrict__ cs, double* __restrict__ sn,
                            double* __restrict__ gam, double* __restrict__ ctl, volatile double* host_out, double serial) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double hh2 = ctl[SD_HH2];
    const double hh = hh2 > 0.0 ? sqrt(hh2) : 0.0;
    ctl[SD_HH] = hh;
    ctl[SD_SCALE] = hh > 0.0 ? 1.0 / hh : 0.0;
    H[(size_t)(k + 1) * m + k] = hh;
    for (int j = 0; j < k; ++j) {
        const double a = H[(size_t)j * m + k], c = H[(size_t)(j + 1) * m + k];
        H[(size_t)j * m + k] = cs[j] * a + sn[j] * c;
        H[(size_t)(j + 1) * m + k] = -sn[j] * a + cs[j] * c;
    }
    const double a = H[(size_t)k * m + k], c = H[(size_t)(k + 1) * m + k];
    const double d = sqrt(a * a + c * c);
    cs[k] = d > 0.0 ? a / d : 1.0;
    sn[k] = d > 0.0 ? c / d : 0.0;
    H[(size_t)k * m + k] = d;
    H[(size_t)(k + 1) * m + k] = 0.0;
    gam[k + 1] = -sn[k] * gam[k];
    gam[k] = cs[k] * gam[k];
    ctl[SD_RES] = fabs(gam[k + 1]);
    host_out[0] = ctl[SD_RES];
    host_out[1] = hh;
    __threadfence_system();
    host_out[2] = serial;
}
__global__ void k_sd_cycle_init(int m, double res, double* __restrict__ gam) {
    for (int i = threadIdx.x; i <= m; i += blockDim.x) gam[i] = i == 0 ? res : 0.0;
}
// y = -(H(0:k,0:k))^-1 gamma   (the sign lets k_sd_multi_axpy, which subtracts, add Z y to x)
__global__ void k_sd_backsolve(int k, int m, const double* __restrict__ H, const double* __restrict__ gam, double* __restrict__ y) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = k - 1; i >= 0; --i) {
        double v = -gam[i];
        for (int j = i + 1; j < k; ++j) v -= H[(size_t)i * m + j] * y[j];
        const double d = H[(size_t)i * m + i];
        y[i] = d != 0.0 ? v / d : 0.0;
    }
}
__global__ void k_sd_scale_dev(int64_t n, const double* __restrict__ a, const double* __restrict__ x, double* __restrict__ y) {
    const double f = *a;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = f * x[i];
}

// replicated pressure problem: owned entries <-> the global vector by global vertex id
__global__ void k_sd_to_global(int64_t nvo, const int64_t* __restrict__ gid, const double* __restrict__ local, double* __restrict__ global) {
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; v < nvo; v += stride) global[gid[v]] = local[v];
}
__global__ void k_sd_from_global(int64_t nvo, const int64_t* __restrict__ gid, const double* __restrict__ global, double* __restrict__ local) {
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; v < nvo; v += stride) local[v] = global[gid[v]];
}

static bool g_sd_timing = false, g_sd_sync = false;
static double g_sd_t[4];

struct saddle_ws {
    dbuf<double> partials, sums, dinv, t, r, w, mdinv, md, mt, hdev, cd, gin, gout;
    dbuf<double> pg_in, pg_out;      // the replicated (global) pressure right-hand side / correction on several ranks
    double vel_lmax = 0.0;   // largest eigenvalue of D^-1 A on the velocity block (Chebyshev sweeps)
    dbuf<const double*> vptr, zptr;
    dbuf<double> hs;            // H | cs | sn | gamma | y | ctl  (device-side Arnoldi state)
    double* h_poll = nullptr;   // pinned, mapped, coherent: 2 slots x {|gamma|, hh, serial, -}
    double* d_poll = nullptr;   // the same memory as the device sees it
    double serial = 0.0;
    dbuf<uint8_t> ident;
    fs_vector_s rp, p1, p2;
    std::vector<dbuf<double>*> V, Z;
    ~saddle_ws() {
        if (h_poll) (void)hipHostFree(h_poll);
        for (auto* v : V) delete v;
        for (auto* z : Z) delete z;
    }
};

static int sd_dot(saddle_ws& W, const double* x, const double* y, int64_t n, double* out, hipStream_t s) {
    const int g = fs_grid_for(n, FS_BLOCK, 1024);
    hipLaunchKernelGGL(k_dot_partial, dim3(g), dim3(FS_BLOCK), 0, s, x, y, n, W.partials.p);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(FS_SUM_BLOCK), 0, s, W.partials.p, g, 1, W.sums.p);
    FS_KERNEL_CHECK();
    FS_CHECK(fs_comm_allreduce_dev(W.sums.p, 1, s));      // no-op on one rank
    FS_CHECK(W.sums.download(out, 1, s));
    return FS_OK;
}

// z = P^-1 r
static int sd_precond(fs_matrix_s* J, fs_matrix_s* Kp, fs_amg_s* Kp_amg, fs_matrix_s* Mp, const fs_saddle_opts* o, saddle_ws& W,
                      const double* r, double* z, int* inner_its, hipStream_t s) {
    fs_space_s* sp = J->space;
    const int64_t n = sp->n_dofs_owned, nv = sp->mesh->n_owned;      // owned vertices = pressure unknowns of this rank
    const int g = fs_grid_for(n, FS_BLOCK, 4096);
    const int sweeps = o->velocity_sweeps > 0 ? o->velocity_sweeps : 1;
    if (sweeps == 1) {
        hipLaunchKernelGGL(k_sd_vel_sweep<true>, dim3(g), dim3(FS_BLOCK), 0, s, n, W.dinv.p, r, (const double*)nullptr, z);
    } else {
        // plain Jacobi sweeps are not a convergent iteration on P2 blocks (lambda_max(D^-1 A) > 2, an even number of
        // sweeps is even indefinite): Chebyshev polynomial in D^-1 A on [lambda_max/8, 1.1 lambda_max] instead
        const double up = 1.1 * W.vel_lmax, lo = W.vel_lmax / 8.0;
        const double theta = 0.5 * (up + lo), delta = 0.5 * (up - lo), sigma = theta / delta;
        double rho_c = 1.0 / sigma;
        hipLaunchKernelGGL(k_sd_vel_cheb<true>, dim3(g), dim3(FS_BLOCK), 0, s, n, W.dinv.p, r, (const double*)nullptr, W.cd.p, z, 1.0 / theta, 0.0);
        for (int k = 1; k < sweeps; ++k) {
            FS_CHECK(fs_halo_exchange_dev(sp, z, s));
            FS_CHECK(fs_spmv_dev(J, z, W.t.p, s));
            const double rho_new = 1.0 / (2.0 * sigma - rho_c);
            hipLaunchKernelGGL(k_sd_vel_cheb<false>, dim3(g), dim3(FS_BLOCK), 0, s, n, W.dinv.p, r, W.t.p, W.cd.p, z, 2.0 * rho_new / delta, rho_new * rho_c);
            rho_c = rho_new;
        }
    }
    // the continuity rows couple to the velocities of ghost nodes.  Multi-GPU: the two pressure solves below act on
    // this rank's diagonal block (ghost columns see zeros) - a non-overlapping additive Schwarz step with the AMG
    // V-cycle / Chebyshev iteration as subdomain solver, no communication inside
    FS_CHECK(fs_halo_exchange_dev(sp, z, s));
    hipLaunchKernelGGL(k_sd_pressure_rows, dim3((unsigned)std::min<int64_t>((nv + 63) / 64, 65535)), dim3(FS_BLOCK), 0, s, nv, sp->slice_ptr.p, sp->sell_col.p,
                       J->val.p, sp->sell_entries, z, r, W.rp.d.p);
    FS_KERNEL_CHECK();
    fs_krylov_opts ko;
    memset(&ko, 0, sizeof(ko));
    ko.method = FS_KSP_CG;
    ko.precond = FS_PC_JACOBI;
    ko.rtol = o->inner_rtol > 0.0 ? o->inner_rtol : 1e-2;
    ko.max_iter = 500;
    ko.diagonal_scale = 1;
    ko.norm_type = FS_NORM_UNPRECONDITIONED;
    fs_krylov_stats ks;
    const bool transient = o->inv_dt > 0.0 && Kp;
    if (transient) {
        if (Kp_amg && fs_rt().n_ranks == 1) {          // one V-cycle: a fixed linear operator, spectrally equivalent to Kp^-1
            FS_CHECK(fs_amg_apply_dev(Kp_amg, W.rp.d.p, W.p1.d.p, s));
        } else if (Kp_amg && fs_amg_rows(Kp_amg) != nv) {
            // Several GPUs, hierarchy of the GLOBAL pressure Laplacian held by every rank: the pressure space is small
            // (85 184 unknowns at configs[4]), so the Schur-complement solve is replicated - the owned residuals are
            // summed into the global vector (one all-reduce, 0.7 MB), every rank applies the same V-cycle and keeps
            // its own entries.  Same operator, hence same FGMRES iteration counts, as on one GPU; a block-Jacobi
            // K_p^-1 (rank-local hierarchies) loses the smooth pressure modes that dominate at small dt and
            // FGMRES(60) stagnates at 2e-5 with 2 and 4 ranks.
            const int64_t ng = fs_amg_rows(Kp_amg);
            FS_REQUIRE(sp->mesh->gid.p, "fs_saddle_solve: the mesh carries no global vertex ids (fs_mesh_set_global_ids)");
            if (W.pg_in.n != ng) { FS_CHECK(W.pg_in.alloc(ng)); FS_CHECK(W.pg_out.alloc(ng)); }
            FS_CHECK(W.pg_in.zero(s));
            hipLaunchKernelGGL(k_sd_to_global, dim3(fs_grid_for(nv)), dim3(FS_BLOCK), 0, s, nv, sp->mesh->gid.p, W.rp.d.p, W.pg_in.p);
            FS_CHECK(fs_comm_allreduce_dev(W.pg_in.p, (int)ng, s));
            FS_CHECK(fs_amg_apply_dev(Kp_amg, W.pg_in.p, W.pg_out.p, s));
            hipLaunchKernelGGL(k_sd_from_global, dim3(fs_grid_for(nv)), dim3(FS_BLOCK), 0, s, nv, sp->mesh->gid.p, W.pg_out.p, W.p1.d.p);
        } else if (Kp_amg) {
            // several GPUs with a hierarchy of this rank's diagonal block only: K_p^-1 as a global solve - CG on the
            // distributed operator with the local V-cycles as (additive Schwarz) preconditioner, to 1e-2; FGMRES
            // tolerates the varying operator
            fs_krylov_opts ao = ko;
            ao.rtol = o->inner_rtol > 0.0 ? o->inner_rtol : 1e-2;
            ao.max_iter = 60;
            ao.norm_type = FS_NORM_UNPRECONDITIONED;
            const int rc_in = fs_amg_solve(Kp_amg, &W.rp, &W.p1, &ao, &ks);
            if (rc_in != FS_OK && rc_in != FS_ERR_NUMERIC) return rc_in;
            *inner_its += ks.iterations;
        } else {
            FS_CHECK(fs_krylov_solve(Kp, &W.rp, &W.p1, &ko, &ks));
            *inner_its += ks.iterations;
        }
    }
    // Mp^-1: the Jacobi-scaled P1 mass matrix has its spectrum in [1/2, 5/2] (Wathen 1987), so 5 Chebyshev
    // steps reduce the error by 1e-2 with no reduction and no host synchronisation: a fixed linear operator
    {
        const int gq = fs_grid_for(nv, FS_BLOCK, 2048);
        const double lo = 0.5, up = 2.5, theta = 0.5 * (up + lo), delta = 0.5 * (up - lo), sigma = theta / delta;
        double rho_c = 1.0 / sigma;
        hipLaunchKernelGGL(k_sd_cheb<true>, dim3(gq), dim3(FS_BLOCK), 0, s, nv, W.mdinv.p, W.rp.d.p, (const double*)nullptr, W.md.p, W.p2.d.p, 1.0 / theta, 0.0);
        for (int k = 1; k < 5; ++k) {
            FS_CHECK(fs_spmv_dev(Mp, W.p2.d.p, W.mt.p, s));
            const double rho_new = 1.0 / (2.0 * sigma - rho_c);
            hipLaunchKernelGGL(k_sd_cheb<false>, dim3(gq), dim3(FS_BLOCK), 0, s, nv, W.mdinv.p, W.rp.d.p, W.mt.p, W.md.p, W.p2.d.p, 2.0 * rho_new / delta, rho_new * rho_c);
            rho_c = rho_new;
        }
        (void)ks;
    }
    const double r2 = o->density * o->density;
    hipLaunchKernelGGL(k_sd_scatter_p, dim3(fs_grid_for(sp->n_nodes_owned)), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, nv,
                       r2 * o->inv_dt, transient ? W.p1.d.p : (const double*)nullptr, r2 * o->kinematic_viscosity, W.p2.d.p, r,
                       W.ident.p, z);
    FS_KERNEL_CHECK();
    FS_CHECK(fs_halo_exchange_dev(sp, z, s));      // ghost pressures for the operator product that follows
    return FS_OK;
}

extern "C" int fs_saddle_solve(fs_matrix_t J, fs_matrix_t Kp, fs_amg_t Kp_amg, fs_matrix_t Mp, fs_vector_t b, fs_vector_t x,
                               const fs_saddle_opts* o, fs_krylov_stats* stats) {
    std::lock_guard<std::recursive_mutex> solve_lock(fs_solve_mutex());
    FS_CHECK(fs_require_init());
    FS_REQUIRE(J && Mp && b && x && o && stats, "fs_saddle_solve: null pointer");
    fs_space_s* sp = J->space;
    FS_REQUIRE(J->bs == 4 && sp->degree == 2, "fs_saddle_solve: the operator must be a Taylor-Hood block matrix");
    const bool multi = fs_rt().n_ranks > 1;
    const int64_t n = sp->n_dofs_owned, nl = sp->n_dofs_local, nv = sp->mesh->n_owned;
    FS_REQUIRE(Mp->bs == 1 && Mp->space->n_dofs_owned == nv && (!Kp || (Kp->bs == 1 && Kp->space->n_dofs_owned == nv)),
               "fs_saddle_solve: the pressure operators must live on the CG1 space of the same mesh");
    FS_REQUIRE(b->d.n >= n && x->d.n >= sp->n_dofs_local, "fs_saddle_solve: vector too short");
    hipStream_t s = fs_rt().stream;
    g_sd_timing = getenv("FS_SADDLE_TIMING") != nullptr;
    g_sd_sync = getenv("FS_SADDLE_SYNC") != nullptr;
    g_sd_t[0] = g_sd_t[1] = g_sd_t[2] = g_sd_t[3] = 0.0;
    const int m = o->restart > 0 ? o->restart : 60;
    const int max_iter = o->max_iter > 0 ? o->max_iter : 600;
    memset(stats, 0, sizeof(*stats));
    const auto t0 = std::chrono::steady_clock::now();

    // the workspace (2m+1 Krylov vectors) is kept between calls: a Newton / time loop solves many systems of one size
    static saddle_ws* g_ws = nullptr;
    static int64_t g_ws_n = -1;
    static int g_ws_m = -1;
    const int dot_blocks = 512;
    static int64_t g_ws_nl = -1;
    if (g_ws && (g_ws_n != n || g_ws_m != m || g_ws_nl != nl)) { delete g_ws; g_ws = nullptr; }
    g_ws_nl = nl;
    const bool fresh = g_ws == nullptr;
    if (fresh) { g_ws = new saddle_ws(); g_ws_n = n; g_ws_m = m; }
    saddle_ws& W = *g_ws;
    auto build_ws = [&]() -> int {
    FS_CHECK(W.partials.alloc(std::max<int64_t>(FS_MAX_PARTIAL_BLOCKS, (int64_t)(m + 3) * dot_blocks)));
    FS_CHECK(W.sums.alloc(8));
    FS_CHECK(W.dinv.alloc(n));
    FS_CHECK(W.t.alloc(n));
    FS_CHECK(W.r.alloc(n));
    FS_CHECK(W.w.alloc(n));
    FS_CHECK(W.cd.alloc(n));
    FS_CHECK(W.gin.alloc(n));
    FS_CHECK(W.gout.alloc(nl));
    FS_CHECK(W.gout.zero(s));
    FS_CHECK(W.ident.alloc(nv));
    FS_CHECK(W.mdinv.alloc(nv));
    FS_CHECK(W.md.alloc(nv));
    FS_CHECK(W.mt.alloc(nv));
    FS_CHECK(W.hdev.alloc(m + 3));
    FS_CHECK(W.vptr.alloc(m + 3));
    FS_CHECK(W.zptr.alloc(m + 3));
    FS_CHECK(W.hs.alloc((int64_t)(m + 1) * m + 4 * (int64_t)m + 1 + SD_CTL));
    FS_HIP(hipHostMalloc((void**)&W.h_poll, 8 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    memset(W.h_poll, 0, 8 * sizeof(double));
    FS_HIP(hipHostGetDevicePointer((void**)&W.d_poll, W.h_poll, 0));
    FS_CHECK(W.rp.d.alloc(nv));
    FS_CHECK(W.p1.d.alloc(Mp->space->n_dofs_local));
    FS_CHECK(W.p2.d.alloc(Mp->space->n_dofs_local));
    FS_CHECK(W.p1.d.zero(s));          // ghost entries stay zero: the pressure solves are rank-local blocks
    FS_CHECK(W.p2.d.zero(s));
    for (int k = 0; k <= m; ++k) {
        W.V.push_back(new dbuf<double>());
        FS_CHECK(W.V.back()->alloc(n));
    }
    for (int k = 0; k < m; ++k) {          // Z_k is the input of an operator product: owned + ghost entries
        W.Z.push_back(new dbuf<double>());
        FS_CHECK(W.Z.back()->alloc(nl));
        FS_CHECK(W.Z.back()->zero(s));
    }
    {
        std::vector<const double*> hp(m + 1);
        for (int k = 0; k <= m; ++k) hp[k] = W.V[k]->p;
        FS_HIP(hipMemcpyAsync(W.vptr.p, hp.data(), (size_t)(m + 1) * sizeof(double*), hipMemcpyHostToDevice, s));
        FS_HIP(hipStreamSynchronize(s));
        for (int k = 0; k < m; ++k) hp[k] = W.Z[k]->p;
        FS_HIP(hipMemcpyAsync(W.zptr.p, hp.data(), (size_t)m * sizeof(double*), hipMemcpyHostToDevice, s));
        FS_HIP(hipStreamSynchronize(s));
    }
        return FS_OK;
    };
    if (fresh) {
        const int rc_ws = build_ws();
        if (rc_ws != FS_OK) { delete g_ws; g_ws = nullptr; return rc_ws; }
    }
    {
        fs_space_s* q = Mp->space;
        hipLaunchKernelGGL(k_sd_scalar_dinv, dim3(fs_grid_for(nv)), dim3(FS_BLOCK), 0, s, nv, q->slice_ptr.p, q->sell_col.p, Mp->val.p, W.mdinv.p);
    }
    hipLaunchKernelGGL(k_sd_diag, dim3(fs_grid_for(sp->n_nodes_owned)), dim3(FS_BLOCK), 0, s, sp->n_nodes_owned, sp->slice_ptr.p, sp->sell_col.p, J->val.p, sp->sell_entries, W.dinv.p);
    hipLaunchKernelGGL(k_sd_ident_p, dim3(fs_grid_for(nv)), dim3(FS_BLOCK), 0, s, nv, sp->slice_ptr.p, sp->sell_col.p, J->val.p, sp->sell_entries, W.ident.p);
    FS_KERNEL_CHECK();

    const int g = fs_grid_for(n, FS_BLOCK, 4096);
    const bool transient_kp = o->inv_dt > 0.0 && Kp != nullptr;     // an inner CG on Kp (no hierarchy given) cannot be captured
    if (o->velocity_sweeps > 1) {
        // lambda_max(D^-1 A) of the velocity block: 12 power iterations from a hashed start
        hipLaunchKernelGGL(k_sd_seed, dim3(g), dim3(FS_BLOCK), 0, s, n, W.w.p);
        double lam = 1.0;
        for (int it = 0; it < 12; ++it) {
            double nn = 0.0;
            FS_CHECK(sd_dot(W, W.w.p, W.w.p, n, &nn, s));
            nn = sqrt(nn);
            if (!(nn > 0.0)) break;
            if (it > 0) lam = nn;
            hipLaunchKernelGGL(k_sd_scale_to, dim3(g), dim3(FS_BLOCK), 0, s, n, 1.0 / nn, W.w.p, W.Z[0]->p);
            FS_CHECK(fs_halo_exchange_dev(sp, W.Z[0]->p, s));
            FS_CHECK(fs_spmv_dev(J, W.Z[0]->p, W.t.p, s));
            hipLaunchKernelGGL(k_sd_vel_scale, dim3(g), dim3(FS_BLOCK), 0, s, n, W.dinv.p, W.t.p, W.w.p);
        }
        W.vel_lmax = lam;
    }
    double bb = 0.0;
    FS_CHECK(sd_dot(W, b->d.p, b->d.p, n, &bb, s));
    stats->bnorm = sqrt(bb);
    if (!o->nonzero_guess) FS_HIP(hipMemsetAsync(x->d.p, 0, (size_t)sp->n_dofs_local * sizeof(double), s));
    const double thr = std::max(o->rtol * stats->bnorm, o->atol);
    int it = 0, conv = 0, inner = 0;
    double res = 0.0;
    double* const dH = W.hs.p;
    double* const dcs = dH + (size_t)(m + 1) * m;
    double* const dsn = dcs + m;
    double* const dgam = dsn + m;
    double* const dy = dgam + m + 1;
    double* const dctl = dy + m;
    // The preconditioner is ~55 small launches (AMG V-cycle on the pressure Laplacian, Chebyshev mass solve): a fixed
    // sequence on fixed buffers once its input/output are staged in gin/gout, so it can be captured into a hipGraph
    // and replayed (FS_SADDLE_GRAPH=1).  Measured on MI355X / ROCm 7.2 (round 1): replay and direct launches both take
    // 0.35 ms - the enqueue costs the host only 0.1 ms, it is not launch-bound - so direct launches stay the default.
    hipGraph_t pgraph = nullptr;
    hipGraphExec_t pexec = nullptr;
    bool use_graph = getenv("FS_SADDLE_GRAPH") != nullptr && (!transient_kp || Kp_amg != nullptr) && !multi;
    if (use_graph) {
        FS_HIP(hipStreamSynchronize(s));
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int dummy = 0;
            const int rc_cap = sd_precond(J, Kp, Kp_amg, Mp, o, W, W.gin.p, W.gout.p, &dummy, s);
            const hipError_t e_end = hipStreamEndCapture(s, &pgraph);
            if (rc_cap != FS_OK || e_end != hipSuccess || !pgraph ||
                hipGraphInstantiate(&pexec, pgraph, nullptr, nullptr, 0) != hipSuccess) {
                if (pgraph) (void)hipGraphDestroy(pgraph);
                pgraph = nullptr; pexec = nullptr; use_graph = false;
                (void)hipGetLastError();
            }
        } else {
            use_graph = false;
            (void)hipGetLastError();
        }
    }
    if (g_sd_timing) {      // back-to-back cost of the preconditioner, graph replay vs direct launches
        int dummy = 0;
        (void)hipStreamSynchronize(s);
        auto t0g = std::chrono::steady_clock::now();
        if (use_graph) for (int rep = 0; rep < 20; ++rep) (void)hipGraphLaunch(pexec, s);
        (void)hipStreamSynchronize(s);
        auto t1g = std::chrono::steady_clock::now();
        for (int rep = 0; rep < 20; ++rep) (void)sd_precond(J, Kp, Kp_amg, Mp, o, W, W.gin.p, W.gout.p, &dummy, s);
        (void)hipStreamSynchronize(s);
        auto t2g = std::chrono::steady_clock::now();
        for (int rep = 0; rep < 20; ++rep) (void)fs_spmv_dev(J, W.gin.p, W.w.p, s);
        (void)hipStreamSynchronize(s);
        auto t3g = std::chrono::steady_clock::now();
        (void)fs_spmv_dev(J, W.gin.p, W.w.p, s);
        (void)hipStreamSynchronize(s);
        auto t4g = std::chrono::steady_clock::now();
        fprintf(stderr, "[fs_saddle_solve] block SpMV: %.3f ms each back to back (x20), %.3f ms for a single launch + sync\n",
                std::chrono::duration<double, std::milli>(t3g - t2g).count() / 20, std::chrono::duration<double, std::milli>(t4g - t3g).count());
        fprintf(stderr, "[fs_saddle_solve] preconditioner x20 back to back: graph %.3f ms each, direct %.3f ms each\n",
                std::chrono::duration<double, std::milli>(t1g - t0g).count() / 20, std::chrono::duration<double, std::milli>(t2g - t1g).count() / 20);
    }
    struct graph_guard {
        hipGraph_t& g; hipGraphExec_t& e;
        ~graph_guard() { if (e) (void)hipGraphExecDestroy(e); if (g) (void)hipGraphDestroy(g); }
    } guard{pgraph, pexec};
    while (true) {
        // r = b - J x
        FS_CHECK(fs_halo_exchange_dev(sp, x->d.p, s));
        FS_CHECK(fs_spmv_dev(J, x->d.p, W.t.p, s));
        hipLaunchKernelGGL(k_sd_sub, dim3(g), dim3(FS_BLOCK), 0, s, n, b->d.p, W.t.p, W.r.p);
        double rr = 0.0;
        FS_CHECK(sd_dot(W, W.r.p, W.r.p, n, &rr, s));
        res = sqrt(rr);
        if (!(res == res)) { conv = -1; break; }
        if (res <= thr) { conv = 1; break; }
        if (it >= max_iter) break;
        hipLaunchKernelGGL(k_sd_scale_to, dim3(g), dim3(FS_BLOCK), 0, s, n, 1.0 / res, W.r.p, W.V[0]->p);
        hipLaunchKernelGGL(k_sd_cycle_init, dim3(1), dim3(64), 0, s, m, res, dgam);
        // One FGMRES iteration is a fixed launch sequence: the Hessenberg column, the rotations and the residual
        // recurrence live on the device.  The host runs one iteration ahead of the GPU and reads |gamma| of iteration
        // k-1 (pinned copy + event) after it has enqueued iteration k, so the queue never drains; the price is at
        // most one iteration enqueued beyond convergence, whose column is simply not used in the update.
        int k = 0, kuse = 0, slot = 0, pending = -1, pending_k = -1;
        bool stop = false;
        // FS_SADDLE_TRACE: per-iteration GPU time (events) beside the host's enqueue and wait times, without extra syncs
        static const bool trace = getenv("FS_SADDLE_TRACE") != nullptr;
        std::vector<hipEvent_t> tev;
        std::vector<double> t_enq, t_wait;
        if (trace) {
            tev.resize(m + 1);
            for (auto& e : tev) (void)hipEventCreate(&e);
            (void)hipEventRecord(tev[0], s);
        }
        // hipEventSynchronize is not used for the hand-over: a wait of ~1 ms leaves the runtime's active-wait window and
        // blocks on an interrupt, and that wake-up was measured to take 25-65 ms once or twice per solve (MI355X,
        // ROCm 7.2; FS_SADDLE_TRACE) - the host spins on the serial number the kernel writes instead.
        double expect[2] = {0.0, 0.0};
        auto harvest = [&]() -> int {       // result of the iteration behind `pending`
            volatile double* hp = W.h_poll + 4 * pending;
            const auto t_spin = std::chrono::steady_clock::now();
            for (uint64_t spin = 0; hp[2] != expect[pending]; ++spin) {
                __builtin_ia32_pause();
                if ((spin & 0xfffff) == 0xfffff) {          // every ~10 ms: a failed launch or a hung device must not spin for ever
                    const hipError_t q = hipStreamQuery(s);
                    if (q != hipSuccess && q != hipErrorNotReady) FS_HIP(q);
                    if (q == hipSuccess && hp[2] != expect[pending]) { fs_set_error("fs_saddle_solve: the iteration result never arrived"); return FS_ERR_HIP; }
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_spin).count() > 120.0) { fs_set_error("fs_saddle_solve: device timeout"); return FS_ERR_HIP; }
                }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            const double rk = hp[0], hk = hp[1];
            res = rk;
            kuse = pending_k + 1;
            if (!(rk == rk)) { conv = -1; stop = true; }
            else if (rk <= thr || !(hk > 0.0)) stop = true;
            pending = -1;
            return FS_OK;
        };
        for (; k < m && it + k < max_iter && !stop; ++k) {
            const bool dbg = g_sd_timing || g_sd_sync;
            auto tA = std::chrono::steady_clock::now();
            if (use_graph) {
                FS_HIP(hipMemcpyAsync(W.gin.p, W.V[k]->p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
                FS_HIP(hipGraphLaunch(pexec, s));
                FS_HIP(hipMemcpyAsync(W.Z[k]->p, W.gout.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
            } else {
                FS_CHECK(sd_precond(J, Kp, Kp_amg, Mp, o, W, W.V[k]->p, W.Z[k]->p, &inner, s));
            }
            auto tB = tA;
            if (dbg) { tB = std::chrono::steady_clock::now(); (void)hipStreamSynchronize(s); }
            auto tC = std::chrono::steady_clock::now();
            FS_CHECK(fs_spmv_dev(J, W.Z[k]->p, W.w.p, s));
            if (dbg) (void)hipStreamSynchronize(s);
            auto tD = std::chrono::steady_clock::now();
            // classical Gram-Schmidt with one fused multi-dot launch per pass; the last pointer of the list is w itself,
            // so the pass also returns ||w||^2 before the projection.  The second pass runs only when the projection
            // removed most of w (DGKS-type criterion, decided on the device): orthogonality stays near working precision.
            for (int pass = 0; pass < 2; ++pass) {
                const double* gate = pass ? dctl + SD_NEED2 : nullptr;
                hipLaunchKernelGGL(k_sd_multi_dot, dim3(dot_blocks), dim3(FS_BLOCK), 0, s, n, W.w.p, W.vptr.p, k + 1, 1, W.partials.p, gate);
                hipLaunchKernelGGL(k_sd_multi_sum, dim3(k + 2), dim3(FS_BLOCK), 0, s, W.partials.p, dot_blocks, W.hdev.p, gate);
                // every rank issues the reduction of both passes: whether the second pass counts is decided on the
                // device, from reduced numbers, hence identically on all ranks
                if (multi) FS_CHECK(fs_comm_allreduce_dev(W.hdev.p, k + 2, s));
                hipLaunchKernelGGL(k_sd_hess_pass, dim3(1), dim3(1), 0, s, W.hdev.p, k, m, dH, dctl, pass);
                hipLaunchKernelGGL(k_sd_multi_axpy, dim3(g), dim3(FS_BLOCK), 0, s, n, W.w.p, W.vptr.p, W.hdev.p, k + 1, gate);
            }
            W.serial += 1.0;
            expect[slot] = W.serial;
            hipLaunchKernelGGL(k_sd_givens, dim3(1), dim3(1), 0, s, k, m, dH, dcs, dsn, dgam, dctl, W.d_poll + 4 * slot, W.serial);
            hipLaunchKernelGGL(k_sd_scale_dev, dim3(g), dim3(FS_BLOCK), 0, s, n, dctl + SD_SCALE, W.w.p, W.V[k + 1]->p);
            FS_KERNEL_CHECK();
            if (trace) (void)hipEventRecord(tev[k + 1], s);
            auto tF = std::chrono::steady_clock::now();
            if (g_sd_timing) {
                (void)hipStreamSynchronize(s);
                auto tE = std::chrono::steady_clock::now();
                g_sd_t[0] += std::chrono::duration<double, std::milli>(tB - tA).count();   // enqueue of the preconditioner
                g_sd_t[1] += std::chrono::duration<double, std::milli>(tC - tA).count();   // ... until it has run
                g_sd_t[2] += std::chrono::duration<double, std::milli>(tD - tC).count();   // SpMV
                g_sd_t[3] += std::chrono::duration<double, std::milli>(tE - tD).count();   // orthogonalisation
            }
            if (pending >= 0) FS_CHECK(harvest());
            if (trace) {
                t_enq.push_back(std::chrono::duration<double, std::milli>(tF - tA).count());
                t_wait.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tF).count());
            }
            if (!stop) { pending = slot; pending_k = k; slot ^= 1; }
        }
        if (trace) {
            (void)hipStreamSynchronize(s);
            for (size_t q = 0; q < t_enq.size(); ++q) {
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, tev[q], tev[q + 1]);
                fprintf(stderr, "[fs_saddle_trace] k=%2zu gpu %.3f ms  host enqueue %.3f ms  host wait %.3f ms\n", q, ms, t_enq[q], t_wait[q]);
            }
            for (auto& e : tev) (void)hipEventDestroy(e);
        }
        if (!stop && pending >= 0) FS_CHECK(harvest());
        it += kuse;
        if (conv < 0) break;
        // y = H^-1 gamma ; x += Z y   (the first kuse columns)
        hipLaunchKernelGGL(k_sd_backsolve, dim3(1), dim3(1), 0, s, kuse, m, dH, dgam, dy);
        hipLaunchKernelGGL(k_sd_multi_axpy, dim3(g), dim3(FS_BLOCK), 0, s, n, x->d.p, W.zptr.p, dy, kuse, (const double*)nullptr);
        FS_KERNEL_CHECK();
        if (kuse == 0) break;
    }
    stats->iterations = it;
    stats->converged = conv;
    if (fs_p2p_reduce_enabled()) FS_CHECK(fs_p2p_check(s));      // a peer-to-peer wait timed out: the numbers below mean nothing
    stats->row_classes = 0;
    stats->rel_residual = stats->bnorm > 0.0 ? res / stats->bnorm : 0.0;
    stats->true_rel_residual = stats->rel_residual;
    stats->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    stats->spmv_bytes = sp->nnz_nodes * 16 * 12 + n * 20;
    if (getenv("FS_SADDLE_DEBUG")) fprintf(stderr, "[fs_saddle_solve] %d outer iterations, %d inner CG iterations, %.1f ms%s\n", it, inner, stats->solve_ms, use_graph ? " (preconditioner as hipGraph)" : "");
    if (g_sd_timing) fprintf(stderr, "[fs_saddle_solve] per iteration [ms]: precond enqueue %.3f, precond done %.3f, spmv %.3f, orthogonalisation %.3f\n", g_sd_t[0] / std::max(it, 1), g_sd_t[1] / std::max(it, 1), g_sd_t[2] / std::max(it, 1), g_sd_t[3] / std::max(it, 1));
    if (conv < 0) {
        fs_set_error("fs_saddle_solve: breakdown at iteration %d", it);
        return FS_ERR_NUMERIC;
    }
    return FS_OK;
}

void fs_saddle_preload() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(k_ns_dummy_rows));
    (void)hipGetLastError();
}
