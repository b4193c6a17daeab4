// The row-dictionary product of a P1 (CG1) operator on a uniform box mesh, x staged through LDS windows that MARCH through the
// mesh planes (round 6).  Self-contained (hip runtime only): fs_krylov.hip includes it for the library, tools/probes/
// box_spmv_probe.hip for the stand-alone timing / bit comparison.
//
// Replaces, like k_dict_spmv: PETSc MatMult (+ the three VecDot of the CG iteration) behind
// /root/reference/FenicsSolver/SolverBase.py:663-670 on the operators BASELINE configs[1] builds.
//
// What k_dict_spmv (fs_krylov.hip) left on the table (DESIGN section 3, r05 counters): per row 3.5 vector loads of 16 bytes for x -
// 64 B per row through the texture-address path and the L1 for 8 B of new data -, two thirds of a wave's life in s_waitcnt on a
// chain header -> plan -> eight loads -> multiply -> store per 126-row work item.  0.48 - 0.52 of the HBM peak at 10 M rows.
//
// This kernel uses what a box gives away: rows r = p + b k (p in the mesh plane, b = rows per plane, k the plane), ONE offset list
//   -(a+b+1) -(a+b) | -(b+1) -b | -(a+1) -a | -1 0 1 | a a+1 | b b+1 | a+b a+b+1          (a = rows per mesh line; Kuhn split)
// i.e. plane k-1, k, k+1 at in-plane offsets within +-(a+1).  A workgroup takes a PATCH of L consecutive in-plane rows and marches
// through a chunk of planes; the window x[k b + p0 - H .. k b + p0 + L + H) of every plane is brought into an LDS ring ONCE per
// workgroup (it serves as plane k+1, k, k-1 of three consecutive steps) by a LOADER wave with `global_load_lds_dwordx4` (1 KiB per
// instruction straight into LDS: no registers, no ds_write pass, exact `s_waitcnt vmcnt(N)` counts because the loader issues
// nothing else), two planes ahead of the compute waves.  The compute waves read x from LDS at consecutive addresses (one row per
// lane: conflict-free 8-byte reads), coefficients from a 15-doubles-per-class table in LDS, and touch global memory only for the
// class number (2 B), the dot weight (8 B) and the result (8 B) of a row: 18 B + 8 B x (1 + halo) per row through the vector
// memory path instead of 82.
// Terms in ascending offset order, one fma each: the bits of k_dict_spmv and of the streaming kernels (a position a row has no
// entry at carries a zero coefficient in either kernel; the zero terms of the plan layout's padding are not executed here - they
// add +0 to a sum that started at +0, which changes no bit).
// Everything is linear in the row number: a window simply continues into the neighbouring mesh line / plane where the patch ends
// (those positions have zero coefficients), indices outside [0, n) are clamped into the vector.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

struct box_geom {
    int64_t n;          // rows = columns of the operator
    int64_t b;          // rows per mesh plane
    int32_t a;          // rows per mesh line
    int32_t nz;         // mesh planes (n = b nz)
    int32_t L;          // rows per patch (even)
    int32_t P;          // patches per plane
    int32_t ZC;         // chunks of planes
    int32_t H;          // window positions in front of a patch (even, >= a + 1)
    int32_t slot;       // doubles per window = 128 G
    int32_t G;          // 1 KiB pieces per window
    int32_t dslot;      // doubles per dot-weight slot (multiple of 128)
    int32_t cslot;      // class numbers per slot (multiple of 512)
    int32_t units;      // P ZC
    int32_t upx;        // units per XCD
    int32_t grid;       // workgroups (multiple of 8)
    int32_t S;          // doubles per class row in the dictionary (plan layout)
    uint8_t pos[16];    // plan-layout position of the 15 offsets, ascending
};

constexpr int BOX_TERMS = 15;
// a window value: hipcc pairs neighbouring 8-byte LDS reads into ds_read2_b64 (half the LDS rate of ds_read_b64 per byte on gfx950,
// MI355X_MICROARCH.md, LDS table); a volatile access stays one ds_read_b64
#ifndef BOX_DC_AUX
#define BOX_DC_AUX 0        // cache policy of the read-once streams (dot weights, class numbers): 2 = nt
#endif
#ifdef BOX_PLAIN_LDS_READS
#define BOX_X(p) (*(p))
#else
#define BOX_X(p) (*(const volatile __attribute__((address_space(3))) double*)(p))
#endif

__device__ __forceinline__ void box_wait_vm(int n) {
    // (the count is an immediate of s_waitcnt; n is wave-uniform)
#define BOX_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        BOX_W(0) BOX_W(1) BOX_W(2) BOX_W(3) BOX_W(4) BOX_W(5) BOX_W(6) BOX_W(7) BOX_W(8) BOX_W(9) BOX_W(10) BOX_W(11) BOX_W(12) BOX_W(13) BOX_W(14) BOX_W(15)
        BOX_W(16) BOX_W(17) BOX_W(18) BOX_W(19) BOX_W(20) BOX_W(21) BOX_W(22) BOX_W(23) BOX_W(24) BOX_W(25) BOX_W(26) BOX_W(27) BOX_W(28) BOX_W(29) BOX_W(30) BOX_W(31)
        BOX_W(32) BOX_W(33) BOX_W(34) BOX_W(35) BOX_W(36) BOX_W(37) BOX_W(38) BOX_W(39) BOX_W(40) BOX_W(41) BOX_W(42) BOX_W(43) BOX_W(44) BOX_W(45) BOX_W(46) BOX_W(47)
        BOX_W(48) BOX_W(49) BOX_W(50) BOX_W(51) BOX_W(52) BOX_W(53) BOX_W(54) BOX_W(55) BOX_W(56) BOX_W(57) BOX_W(58) BOX_W(59) BOX_W(60) BOX_W(61) BOX_W(62)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef BOX_W
}

__device__ __forceinline__ void box_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// DOTS as k_dict_spmv: 0 plain product; 1: (r.z, w.z, r.r); 2: (w.r, w.w, r.r); 3: (z.z, w.z, sum r z^2); 4: status word only.
// CW compute waves + NL loader waves (NL = 2: one for the windows, one for dot weights + class numbers); RP rows per compute lane
// and plane (L <= 64 CW RP); D: steps the loaders run ahead (D + 1 slots of window / weights / classes each).
//
// STEP j of a unit (planes k0 .. k1 - 1 of a patch; j = k0 - 1 .. k1) has ONE window in LDS, plane j's, and every lane carries the
// partial sums of its rows in registers: with the window of plane j a row of plane j + 1 gets its terms 0 - 3 (its plane k - 1
// entries), a row of plane j its terms 4 - 10, a row of plane j - 1 its terms 11 - 14 and is finished - the ascending order of the
// streaming kernels, because the planes arrive in ascending order.  The three rows sit at the same place of the plane and read the
// same SEVEN window values.  One resident window instead of three is what lets the loaders run D = 2 steps ahead inside the LDS
// budget of two workgroups per CU (a step's critical path was issue + landing time of one round of loads with D = 1: 2.8 us per
// step, 55 us per product at 10 M rows).
// The loaders bring EVERYTHING a step reads - window, dot weights, class numbers - into LDS; the compute waves issue stores only.
// (First form: class number and dot weight by ordinary loads into registers a step ahead.  hipcc's s_waitcnt insertion made every
// step wait for the loads it had just issued - vmcnt(0) at the register copy, or far too small a count with two register sets.)
template <int DOTS, int CW, int RP, int D, int NL>
__global__ void __launch_bounds__((CW + NL) * 64) k_box_spmv(box_geom g, const uint16_t* __restrict__ cls, const double* __restrict__ dict, int ncls,
                                                             const double* __restrict__ x, double* __restrict__ y,
                                                             const double* __restrict__ rvec, double* __restrict__ partials,
                                                             int* __restrict__ status, int part_base, int part_stride, int bump) {
    const int st0 = DOTS ? status[0] : 0;
    extern __shared__ __attribute__((aligned(16))) double box_lds[];
    __shared__ double red[3][CW + NL];
    constexpr int NS = D + 1;                                      // slots
    constexpr bool WD = DOTS == 1 || DOTS == 2 || DOTS == 3;       // the dot weights are read
    static_assert(D >= 1 && D <= 4 && (NL == 1 || NL == 2), "1 .. 4 steps ahead, one or two loaders");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // LDS: windows | dot weights | class numbers | coefficient table
    double* __restrict__ ring = box_lds;
    double* __restrict__ dring = ring + NS * g.slot;
    uint16_t* __restrict__ cring = reinterpret_cast<uint16_t*>(dring + (WD ? NS * g.dslot : 0));
    double* __restrict__ coef = reinterpret_cast<double*>(cring + NS * g.cslot);
    // class rows: plan layout -> 15 per class (an odd pitch: classes four apart no longer share their banks)
    for (int i = threadIdx.x; i < ncls * BOX_TERMS; i += (CW + NL) * 64) {
        const int c = i / BOX_TERMS, t = i - c * BOX_TERMS;
        coef[i] = dict[(int64_t)c * g.S + g.pos[t]];
    }
    if (DOTS) {
        if (st0 != 0) return;
        if (bump && blockIdx.x == 0 && threadIdx.x == 0) status[2] += 1;
    }
    __syncthreads();
    double d_rz = 0.0, d_wz = 0.0, d_rr = 0.0;
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, ustep = gridDim.x >> 3;
    const int u_end = min((xcd + 1) * g.upx, g.units);
    const int64_t emax = (g.n - 1) & ~(int64_t)1;      // last 16-byte pair of doubles that starts inside a vector
    const int64_t cmax = (g.n - 1) & ~(int64_t)7;      // last 16-byte group of class numbers that starts inside the array
    const int bodd = (int)(g.b & 1);
    const int Gd = WD ? g.dslot >> 7 : 0, Gc = g.cslot >> 9;
    for (int u = xcd * g.upx + j0; u < u_end; u += ustep) {
        const int patch = u % g.P, zc = u / g.P;
        const int k0 = (int)((int64_t)zc * g.nz / g.ZC), k1 = (int)((int64_t)(zc + 1) * g.nz / g.ZC);
        const int64_t p0 = (int64_t)patch * g.L;
        const int Lp = (int)min((int64_t)g.L, g.b - p0);
        const int jb = k0 - 1;                       // first step
        if (wave >= CW) {
            // ---------------- loaders: round r (behind barrier B_r) brings what step r + D reads ----------------
            const bool do_x = NL == 1 || wave == CW, do_dc = NL == 1 || wave == CW + 1;
            const uint32_t lane16 = (uint32_t)lane * 16u;
            auto pieces = [&](const void* src0, int64_t first, int64_t last_ok, int per_lane, int64_t stride, int elem_bytes, void* dst0, int n_pieces, auto aux_tag) {
                constexpr int AUX = decltype(aux_tag)::value;
                // piece c: 1 KiB from element first + c stride (+ per_lane * lane) into dst0 + c KiB; elements outside [0, last_ok] clamped
                const char* s8 = reinterpret_cast<const char*>(src0);
                char* d8 = reinterpret_cast<char*>(dst0);
                if (first >= 0 && first + (int64_t)(n_pieces - 1) * stride + 63 * per_lane <= last_ok) {
                    const char* base = s8 + first * elem_bytes;
                    for (int c = 0; c < n_pieces; ++c)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (int64_t)c * 1024 + lane16),
                                                         (__attribute__((address_space(3))) void*)(d8 + c * 1024), 16, 0, AUX);
                } else {
                    for (int c = 0; c < n_pieces; ++c) {
                        int64_t e = first + c * stride + per_lane * lane;
                        e = e < 0 ? 0 : (e > last_ok ? last_ok : e);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s8 + e * elem_bytes),
                                                         (__attribute__((address_space(3))) void*)(d8 + c * 1024), 16, 0, AUX);
                    }
                }
            };
            auto has_c = [&](int j) { return j + 1 >= k0 && j + 1 < k1; };      // step j reads the classes of plane j + 1
            auto has_d = [&](int j) { return WD && j - 1 >= k0 && j - 1 < k1; }; // ... and the dot weights of plane j - 1
            auto round_size = [&](int r) {
                const int j = r + D;
                if (j > k1) return 0;
                return (do_x ? g.G : 0) + (do_dc ? (has_c(j) ? Gc : 0) + (has_d(j) ? Gd : 0) : 0);
            };
            auto issue_round = [&](int r) {
                const int j = r + D;
                if (j > k1) return;
                const int sl = (j - jb) % NS;
                if (do_x) pieces(x, ((int64_t)j * g.b + p0 - g.H) & ~(int64_t)1, emax, 2, 128, 8, ring + sl * g.slot, g.G, std::integral_constant<int, 0>());
                if (do_dc) {
                    if (has_c(j)) pieces(cls, ((int64_t)(j + 1) * g.b + p0) & ~(int64_t)7, cmax, 8, 512, 2, cring + sl * g.cslot, Gc, std::integral_constant<int, BOX_DC_AUX>());
                    if (has_d(j)) pieces(rvec, ((int64_t)(j - 1) * g.b + p0) & ~(int64_t)1, emax, 2, 128, 8, dring + sl * g.dslot, Gd, std::integral_constant<int, BOX_DC_AUX>());
                }
            };
            for (int r = jb - D; r < jb; ++r) issue_round(r);
            for (int j = jb; j <= k1; ++j) {
                int behind = 0;                      // rounds j - D + 1 .. j - 1 may still be in flight
                for (int r = j - D + 1; r < j; ++r) behind += round_size(r);
                box_wait_vm(behind);
                box_barrier();                       // B_j
                issue_round(j);
            }
            box_barrier();                           // (the unit's last step is done: its slots may be overwritten)
        } else {
            // ---------------- compute waves ----------------
            typedef const __attribute__((address_space(3))) double* lds_cdp;
            typedef const __attribute__((address_space(3))) uint16_t* lds_cup;
            const lds_cdp ring3 = (lds_cdp)ring, dring3 = (lds_cdp)dring;
            const lds_cup cring3 = (lds_cup)cring;
            const int a = g.a;
            double accC[RP], accN[RP], zP[RP];       // sums of the rows of plane j (terms 0 - 10 done after step j) / j + 1 (0 - 3)
            int cC[RP], cP[RP];
#pragma unroll
            for (int i = 0; i < RP; ++i) { accC[i] = accN[i] = zP[i] = 0.0; cC[i] = cP[i] = 0; }
            int sl = 0;
            for (int j = jb; j <= k1; ++j) {
                box_barrier();                       // B_j
                const bool hp = j - 1 >= k0, hc = j >= k0 && j < k1, hn = j + 1 < k1;     // rows of plane j - 1 / j / j + 1 in this unit
                const lds_cdp W = ring3 + sl * g.slot + (bodd & j);
                const lds_cup Cn = cring3 + sl * g.cslot + (int)(((int64_t)(j + 1) * g.b + p0) & 7);
                const lds_cdp Dp = dring3 + sl * g.dslot + (int)(((int64_t)(j - 1) * g.b + p0) & 1);
                sl = sl + 1 == NS ? 0 : sl + 1;
                const int64_t rp = (int64_t)(j - 1) * g.b + p0;
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const int q = (wave * RP + i) * 64 + lane;
                    const bool live = q < Lp;
                    const int cN = (hn && live) ? (int)Cn[q] : 0;
                    const lds_cdp Wq = W + q + g.H;
                    const double v0 = BOX_X(Wq - a - 1), v1 = BOX_X(Wq - a), v2 = BOX_X(Wq - 1), v3 = BOX_X(Wq), v4 = BOX_X(Wq + 1),
                                 v5 = BOX_X(Wq + a), v6 = BOX_X(Wq + a + 1);
                    if (hp) {
                        const double* __restrict__ cf = coef + cP[i] * BOX_TERMS;
                        double acc = accC[i];
                        acc = fma(cf[11], v3, acc);
                        acc = fma(cf[12], v4, acc);
                        acc = fma(cf[13], v5, acc);
                        acc = fma(cf[14], v6, acc);
                        if (live) {
                            y[rp + q] = acc;
                            const double z = zP[i];
                            const double ri = WD ? Dp[q] : 0.0;
                            if (DOTS == 1) { d_rz += ri * z; d_wz += acc * z; d_rr += ri * ri; }
                            else if (DOTS == 2) { d_rz += acc * ri; d_wz += acc * acc; d_rr += ri * ri; }
                            else if (DOTS == 3) { d_rz += z * z; d_wz += acc * z; d_rr += ri * z * z; }
                        }
                    }
                    if (hc) {
                        const double* __restrict__ cf = coef + cC[i] * BOX_TERMS;
                        double acc = accN[i];
                        acc = fma(cf[4], v0, acc);
                        acc = fma(cf[5], v1, acc);
                        acc = fma(cf[6], v2, acc);
                        acc = fma(cf[7], v3, acc);
                        acc = fma(cf[8], v4, acc);
                        acc = fma(cf[9], v5, acc);
                        acc = fma(cf[10], v6, acc);
                        accC[i] = acc;
                        zP[i] = v3;
                    }
                    if (hn) {
                        const double* __restrict__ cf = coef + cN * BOX_TERMS;
                        double acc = 0.0;
                        acc = fma(cf[0], v0, acc);
                        acc = fma(cf[1], v1, acc);
                        acc = fma(cf[2], v2, acc);
                        acc = fma(cf[3], v3, acc);
                        accN[i] = acc;
                    }
                    cP[i] = cC[i];
                    cC[i] = cN;
                }
            }
            box_barrier();
        }
    }
    if (DOTS && DOTS != 4) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            d_rz += __shfl_down(d_rz, off, 64);
            d_wz += __shfl_down(d_wz, off, 64);
            d_rr += __shfl_down(d_rr, off, 64);
        }
        if (lane == 0) { red[0][wave] = d_rz; red[1][wave] = d_wz; red[2][wave] = d_rr; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t0 = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int w = 0; w < CW; ++w) { t0 += red[0][w]; t1 += red[1][w]; t2 += red[2][w]; }
            partials[part_base + blockIdx.x] = t0;
            partials[part_stride + part_base + blockIdx.x] = t1;
            partials[2 * part_stride + part_base + blockIdx.x] = t2;
        }
    }
}

// Host side: is this one-round plan the Kuhn box list, and how to cut the box for a launch.  starts / lens: the 8 runs of the plan
// round (run 0 = the z run).  Returns false when the operator is not of the form (the caller keeps k_dict_spmv).
static inline bool box_recognize(const int32_t* starts, const uint8_t* lens, int n_runs_per_round, int64_t n, int RL, box_geom* out) {
    if (n_runs_per_round != 8 || RL != 3) return false;
    static const int want_len[8] = {0, 2, 2, 2, 3, 2, 2, 2};
    for (int j = 1; j < 8; ++j)
        if ((int)lens[j] != want_len[j]) return false;
    if (starts[4] != -1) return false;
    const int64_t a = starts[5], b = starts[6];
    if (a < 3 || b < 2 * a + 2 || b > (int64_t)1 << 30) return false;
    if (starts[7] != a + b || starts[3] != -(a + 1) || starts[2] != -(b + 1) || starts[1] != -(a + b + 1)) return false;
    if (n % b != 0 || n / b < 2) return false;
    box_geom g = {};
    g.n = n; g.b = b; g.a = (int32_t)a; g.nz = (int32_t)(n / b);
    // ascending offsets -> positions 3 run + t of the plan layout
    const int run_of[BOX_TERMS] = {1, 1, 2, 2, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 7}, t_of[BOX_TERMS] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 0, 1, 0, 1, 0, 1};
    for (int t = 0; t < BOX_TERMS; ++t) g.pos[t] = (uint8_t)(RL * run_of[t] + t_of[t]);
    *out = g;
    return true;
}

// Patches and chunks for compute waves x rows per lane = Lmax rows per patch at most, `slots` workgroups resident on the chip.
static inline void box_cut(box_geom* g, int Lmax, int slots, int passes) {
    const int64_t b = g->b;
    int P = (int)((b + Lmax - 1) / Lmax);
    int L = (int)(((b + P - 1) / P + 1) & ~(int64_t)1);
    g->P = P; g->L = L;
    g->H = (g->a + 2) & ~1;                          // even, >= a + 1
    const int need = L + g->H + g->a + 3;            // last position read: L - 1 + H + a + 1, + 1 for an odd shift
    g->G = (need + 127) / 128;
    g->slot = g->G * 128;
    g->dslot = ((L + 2 + 127) / 128) * 128;          // (+ 1 for an odd first row, rounded to 1 KiB pieces)
    g->cslot = ((L + 8 + 511) / 512) * 512;
    int ZC = (slots * passes) / P;
    if (ZC < 1) ZC = 1;
    if (ZC > g->nz) ZC = g->nz;
    g->ZC = ZC;
    g->units = P * ZC;
    g->upx = (g->units + 7) / 8;
    int per_xcd = (g->upx + passes - 1) / passes;       // workgroups per XCD
    g->grid = 8 * per_xcd;
}
static inline size_t box_lds_bytes(const box_geom& g, int ncls, int D, bool weights) {
    return ((size_t)(D + 1) * g.slot + (weights ? (size_t)(D + 1) * g.dslot : 0) + (size_t)ncls * BOX_TERMS) * sizeof(double) + (size_t)(D + 1) * g.cslot * sizeof(uint16_t);
}
