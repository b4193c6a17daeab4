// The row-dictionary product of a scalar CG2 operator on a uniform Kuhn box mesh in the solver's half-grid LATTICE order
// (fs_lattice.hip: row = X + SX (Y + NY Z)), x staged through LDS windows that MARCH through the lattice planes (round 6).
// Self-contained apart from fs_box.h (the loader / barrier helpers) and the generated stencil tables (fs_cg2_stencil.h).
//
// Replaces, like k_lattice_spmv: PETSc MatMult (+ the three VecDot of the CG iteration) behind
// /root/reference/FenicsSolver/SolverBase.py:663-670 on the operators BASELINE configs[3] builds.
//
// What k_lattice_spmv (fs_krylov_lattice.inc) left on the table: per stored entry one LDS broadcast of (coefficient, offset) and one
// 8-byte LDS read of x - three LDS instructions for two fmas with paired lines -, the window of a 128 x 4 x 4 tile re-read 4.1 x from
// L2, 0.27 - 0.28 of the HBM peak at configs[3].
//
// This kernel uses what the lattice gives away.  Every neighbour of a row lies within +-2 in X, Y and Z, and WHICH of the 125
// positions a row can have an entry at depends only on the parities (X & 1, Y & 1, Z & 1) of the row: eight interior stencils of
// 65 / 27 / 19 entries (fs_cg2_stencil.h, generated from the oracle's mesh; a boundary row holds a subset of its parity's list).
// So the loop structure is COMPILE-TIME - window offsets and coefficient positions are immediates of the LDS reads, nothing but the
// coefficient VALUES is looked up - and along a mesh line the rows of one X parity are of one class away from the ends of the line.
// As in k_box_spmv (fs_box.h) a workgroup takes a PATCH - PY whole lines of a plane - and marches through a chunk of planes with
// ONE resident window per step, the plane jz's: a lane owns the two columns X = 2 l, 2 l + 1 of a line and carries, per column, the
// partial sums of the rows of planes jz - 1 .. jz + 2 in registers; with the window of plane jz the row of plane jz + 2 gets its
// dz = -2 entries, .., the row of plane jz - 2 its dz = +2 entries and is finished - a row's terms in ascending (dz, dy, dx) =
// ascending column order, one fma per stencil position: the bits of k_lattice_spmv, k_dict_spmv and the streaming kernels (a
// position a row has no entry at carries the coefficient +0: it adds +-0 to a sum that started at +0).
// The window comes in as (even X, odd X) pairs: a lane reads the up to 15 pairs around its own with ds_read_b128 (conflict-free at
// one pair per lane), 6 - 15 reads for 36 - 82 fmas per lane and step.
// COEFFICIENTS.  What a (line, step) multiplies with is a STEP LIST: the 36 - 82 coefficients of the five rows in flight of both
// columns, contiguous, in the order the step uses them (window line by window line, so that a lane holds two lines of the window at a
// time, not five).  The lists are made once per dictionary (k_lm_lines .. k_lm_fill: a few hundred distinct ones - the interior of the
// box has four) and read through SCALAR loads: the list of a (line, step) is wave-uniform, a coefficient an SGPR operand of the fmas.
// A wave takes a LINE of the patch - up to two pieces of 64 pairs against one stream of the line's list: a coefficient serves 2 x
// (pieces) fmas; a line of more than 128 pairs takes several waves (27 M rows, lines of 302: 222 us against 341 for the tile product).
// The ENDS of the lines (X < LM_LO, X > SX - 1 - LM_HI: boundary rows, the rows coupled to them, the dummy row that makes a line
// even) have classes of their own - 4 % of the rows at configs[3].  They stay with the code of the tile product (lat_line_ends,
// fs_krylov_lattice.inc: column tiles, lanes along Y), in workgroups of their own at the front of the grid of the same launch
// (k_lat_march, fs_krylov_lattice.inc).
// A LOADER wave brings window and dot weights with global_load_lds_dwordx4, LM_D steps ahead (fs_box.h: exact vmcnt counts).
//
// MEASURED (round 6, MI355X, configs[3], 9.98 M rows; tools/probes/run_latmarch_abl.sh, profiles/r06_p2_latmarch.txt): inside the CG
// iteration 76 - 79 us per product against 114 - 120 for k_lattice_spmv; alone, warm, 61 us without / 79 with the dots (tile product
// 96 / 96).  The loads alone (no arithmetic) take 28 / 43 us, the line waves alone 50 / 63 (seven line waves: 56 / 66), the ends of the
// lines 12: what bounds
// the kernel is the instruction stream of a line wave - 2 700 - 3 800 cycles per step where its fmas take 460 (cycle counters around
// barrier and step).  Steps on the way, all bit-identical:
//  * coefficients through scalar loads from per-class rows (ten rows a step, thirteen s_waitcnt lgkmcnt(0) per line task, a task a
//    half line), end waves in every workgroup with per-lane lists from global memory: 75 us alone / 96 with the dots; the roles of
//    the waves as branches INSIDE the loop over the units had hipcc's s_waitcnt insertion carry the end wave's pending vector loads
//    into the line waves - a vmcnt(0), a wait for the stores of y, in front of every task: roles outside the loop 71 / 84;
//  * step lists staged in LDS by the loader, read as broadcasts: ds_read_b64 pairs up into ds_read2_b64 (half rate), the volatile
//    form is not scheduled ahead of its fma (a round trip per coefficient), 16-byte pairs in chunks streamed two or three ahead:
//    line waves alone 50 - 72 us - an LDS round trip per chunk and no registers to ask further ahead;
//  * end waves (two per workgroup, a lane an end pair, its list from global memory): 44 - 106 us on their own;
//  * a wave per line and eight waves (ten spilled at 168 registers), ends by the column tiles of the tile product: 67 / 84;
//  * the list lane-distributed in two registers, a coefficient handed out with two v_readlane_b32 (4 cycles each, tools/probes/
//    readlane_probe.hip) - no memory round trip for coefficients at all: line waves alone 61 us, no better: not the loads of the
//    coefficients, the whole stream (register moves that rotate the five partial sums, hazards, two waves a SIMD) is what a step costs;
//  * loader three steps ahead: no change; dot weights with the default cache policy instead of nt: the update kernel behind the
//    product 98 -> 117 us;
//  * eleven line waves + the loader (twelve waves, three a SIMD, 168 registers allowed, 153 used) instead of seven + one: fewer steps
//    per wave (30.6 -> 21.4), less window halo (1.57 -> 1.36 x): 80 - 85 -> 76 - 79 us inside the iteration - kept;
//  * the steps two to a loop turn with the parities as template constants (no four-way branch, no registers where branches meet):
//    eight copies of the step, 602 SGPRs spilled, 100 us inside the iteration.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <utility>
#include <type_traits>
#include "fs_box.h"
#include "fs_cg2_stencil.h"

constexpr int LM_LO = 4, LM_HI = 5;     // rows at the two ends of a line that lat_line_ends takes (LT_LO / LT_HI of the tile product)
constexpr int LM_ZPAD = 4;              // planes of padding (the zero row) on either side of the per-line row number tables
constexpr int LM_SL = 84;               // doubles per step list (82 at most: a vertex line's even plane), 42 16-byte pairs
constexpr int LM_LW = 11;               // line waves of a workgroup: patches of up to LM_LW lines
#ifndef LM_DC_AUX
#define LM_DC_AUX 2             // cache policy of the dot weights (read once): 2 = nt (0: the update kernel behind the product 98 -> 117 us)
#endif
constexpr int LM_D = 2;                 // steps the loader runs ahead (LM_D + 1 slots of window and dot weights)
constexpr int LM_WAVES = LM_LW + 1;     // + the loader (twelve waves: three a SIMD, 168 registers a lane)

struct lm_geom {
    int64_t n;                  // rows
    int32_t SX, NY, NZ;         // the lattice (SX even)
    int32_t HP;                 // pairs per line = SX / 2
    int32_t NH;                 // 64-pair pieces of a line
    int32_t WPL;                // waves per line: a wave takes up to two pieces
    int32_t PY;                 // lines per patch
    int32_t NP;                 // patches per plane
    int32_t ZC;                 // chunks of planes
    int32_t slot, G;            // doubles per window slot (a multiple of 128), 1 KiB pieces
    int32_t dslot, Gd;          // ... per dot-weight slot
    int32_t LZ;                 // entries per line of the row number tables = NZ + 2 LM_ZPAD
    int32_t NS2;                // steps a line has = NZ + 4 (window planes -2 .. NZ + 1)
    int32_t NYP;                // lines per step of sl_line = NY + LM_LW (a patch's numbers are read LM_LW at a time)
    int32_t units, upx, grid;
};

typedef double lm_v2d __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) lm_v2d* lm_lds_pairs;

template <class F, int... I>
__device__ __forceinline__ void lm_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>()), ...); }
template <int N, class F>
__device__ __forceinline__ void lm_for(F&& f) { lm_for_impl(f, std::make_integer_sequence<int, N>()); }

// A step list: the coefficients a (line, step) uses, in the order it uses them - by window line dy = -2 .. 2, within a window line
// by age k = 0 .. 4 (the row of plane jz + 2 - k: its dz = k - 2 entries), the even column's entries then the odd column's, a
// slice's entries in stencil order (so every row still gets its terms in ascending column order).  By window line: a lane then holds
// two lines of the window at a time, not five.
struct lm_pos_t { int k, col, s, dy; };
struct lm_variant_t {
    int len;                    // coefficients of the list
    lm_pos_t pos[LM_SL];        // what position p multiplies: age, column, stencil position, window line
    int first[5];               // first position of window line dy = -2 .. 2 (len: none)
    bool used[5][3];            // the window pair (dy, di) is read
};
__host__ __device__ constexpr lm_variant_t lm_make_variant(int py, int jzb) {
    lm_variant_t v = {};
    int p = 0;
    for (int dy = -2; dy <= 2; ++dy) {
        v.first[dy + 2] = -1;
        for (int k = 0; k < 5; ++k)
            for (int c = 0; c < 2; ++c) {
                const int par = c + 2 * py + 4 * ((jzb + k) & 1);
                for (int s = LM_SLICE[par][k]; s < LM_SLICE[par][k + 1]; ++s)
                    if (LM_DY[par][s] == dy) {
                        if (v.first[dy + 2] < 0) v.first[dy + 2] = p;
                        v.pos[p] = lm_pos_t{k, c, s, dy};
                        const int q = c + LM_DX[par][s], hi = q & 1;
                        v.used[dy + 2][(q - hi) / 2 + 1] = true;
                        ++p;
                    }
            }
    }
    v.len = p;
    for (int d = 0; d < 5; ++d)
        if (v.first[d] < 0) v.first[d] = p;
    v.used[2][1] = true;
    return v;
}
constexpr lm_variant_t LM_VAR[2][2] = {{lm_make_variant(0, 0), lm_make_variant(0, 1)}, {lm_make_variant(1, 0), lm_make_variant(1, 1)}};
static_assert(LM_VAR[0][0].len <= LM_SL && LM_VAR[0][1].len <= LM_SL && LM_VAR[1][0].len <= LM_SL && LM_VAR[1][1].len <= LM_SL, "step lists fit");
// a 16-byte pair of the window, read where it is written: the window is STREAMED - line dy + 1 asked for when the fmas of line dy
// begin - with in-order s_waitcnt counts (hipcc hoists plain loads: the whole window at once, 120 registers with two pieces)
__device__ __forceinline__ lm_v2d lm_pair(lm_lds_pairs p, int j) { return *(const volatile __attribute__((address_space(3))) lm_v2d*)(p + j); }

// One step of a line: its RP pieces of 64 pairs (a lane: pair lane of each piece, the two columns of the pair) against ONE stream of
// the line's step list.  PYB: parity of the line, JZB: parity of the window plane.  Wc[i]: the lane's own pair of piece i in the
// window (line + 2, pair + 1 of the slot), hp: pairs per line.  cp[j]: the step list in pairs, the same address in all lanes.
// A[i][k - 1]: the sums of the rows of age k = 1 .. 4 (plane jz + 2 - k), E: even column, O: odd.  out: the finished rows (plane
// jz - 2); cen: the window values at the lane's own position.
template <int PYB, int JZB, int RP>
__device__ __forceinline__ void lm_line_step(const lm_lds_pairs (&Wc)[RP], int hp, const double* __restrict__ cp, double (&AE)[RP][4], double (&AO)[RP][4], double (&outE)[RP], double (&outO)[RP],
                                             double (&cenE)[RP], double (&cenO)[RP]) {
    constexpr lm_variant_t V = LM_VAR[PYB][JZB];
    constexpr int LEN = V.len;
    static_assert((V.first[1] < LEN || !(V.used[2][0] || V.used[2][1] || V.used[2][2])) && (V.first[2] < LEN || !(V.used[3][0] || V.used[3][1] || V.used[3][2])) &&
                  (V.first[3] < LEN || !(V.used[4][0] || V.used[4][1] || V.used[4][2])), "a window line is asked for when the fmas of the line before it begin");
    lm_v2d P[RP][5][3];
    auto window_line = [&](auto DY) {
        constexpr int dy = decltype(DY)::value;
        lm_for<3>([&](auto DI) {
            constexpr int di = decltype(DI)::value - 1;
            if constexpr (V.used[dy + 2][di + 1]) {
#pragma unroll
                for (int i = 0; i < RP; ++i) P[i][dy + 2][di + 1] = lm_pair(Wc[i], dy * hp + di);
            }
        });
    };
    window_line(std::integral_constant<int, -2>());
    window_line(std::integral_constant<int, -1>());
    double rE[RP][5], rO[RP][5];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        rE[i][0] = rO[i][0] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { rE[i][k + 1] = AE[i][k]; rO[i][k + 1] = AO[i][k]; }
    }
    lm_for<LEN>([&](auto QQ) {
        constexpr int p = decltype(QQ)::value;
        constexpr lm_pos_t at = V.pos[p];
        constexpr int k = at.k, par = at.col + 2 * PYB + 4 * ((JZB + k) & 1), dy = at.dy;
        constexpr int q = at.col + LM_DX[par][at.s], hi = q & 1, di = (q - hi) / 2;
        static_assert(LM_DZ[par][at.s] == k - 2 && LM_DY[par][at.s] == dy && di >= -1 && di <= 1, "stencil tables");
        // (window line dy + 1 is asked for when the fmas of line dy begin)
        lm_for<3>([&](auto DD) {
            constexpr int d = decltype(DD)::value;          // 0 .. 2
            if constexpr (p == V.first[d + 1] && V.first[d + 1] < LEN) window_line(std::integral_constant<int, d>());
        });
        const double cf = cp[p];            // (wave-uniform: a scalar load, the coefficient an SGPR operand of the fmas)
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const double xv = hi ? P[i][dy + 2][di + 1].y : P[i][dy + 2][di + 1].x;
            if constexpr (at.col == 0) rE[i][k] = fma(cf, xv, rE[i][k]);
            else rO[i][k] = fma(cf, xv, rO[i][k]);
        }
    });
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        outE[i] = rE[i][4]; outO[i] = rO[i][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { AE[i][k] = rE[i][k]; AO[i][k] = rO[i][k]; }
        cenE[i] = P[i][2][1].x; cenO[i] = P[i][2][1].y;
    }
}


// The interior of the lines (X = LM_LO .. SX - 1 - LM_HI) of the units of workgroup wg of n_wg.  DOTS as k_box_spmv.  A workgroup:
// LM_LW line waves - a wave takes RP <= 2 pieces of 64 pairs of a line of the patch, WPL waves a line - and a loader wave; D: steps the loaders
// run ahead (D + 1 slots each).  SL[list][LM_SL]: the step lists; sl_line[(jz + 2) NYP + Y]: the list of line Y at window plane jz.
// lds: the dynamic LDS (lm_lds_bytes).  The dot sums of a lane are added to d_rz, d_wz, d_rr.
template <int DOTS, int RP, int D>
__device__ __forceinline__ void lm_march(const lm_geom& g, int wg, int n_wg, const double* __restrict__ SL, const int32_t* __restrict__ sl_line,
                                         const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ rvec, double* __restrict__ lds, int dbg,
                                         double& d_rz, double& d_wz, double& d_rr) {
    constexpr int NS = D + 1;
    constexpr bool WD = DOTS == 1 || DOTS == 2 || DOTS == 3;
    static_assert(D >= 1 && D <= 4, "1 .. 4 steps ahead");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // LDS: windows | dot weights
    double* __restrict__ ring = lds;
    double* __restrict__ dring = ring + NS * g.slot;
    const int xcd = wg & 7, j0 = wg >> 3, ustep = n_wg >> 3;
    const int u_end = min((xcd + 1) * g.upx, g.units);
    const int64_t emax = (g.n - 1) & ~(int64_t)1;
    const int SX = g.SX, NY = g.NY, HP = g.HP;
    // (the role of a wave OUTSIDE the loop over the units: with the roles as branches inside it hipcc's s_waitcnt insertion carries one
    // role's pending vector memory operations around the loop into the others - a vmcnt(0), i.e. a wait for the stores of y, in front
    // of every line step)
    auto run = [&](auto ROLE_T) {
    constexpr int ROLE = decltype(ROLE_T)::value;          // 0: loader, 1: line wave, 2: a wave without a line (PY < LM_LW)
    for (int u = xcd * g.upx + j0; u < u_end; u += ustep) {
        const int patch = u % g.NP, zc = u / g.NP;
        const int z0 = (int)((int64_t)zc * g.NZ / g.ZC), z1 = (int)((int64_t)(zc + 1) * g.NZ / g.ZC);
        const int Y0 = patch * g.PY;
        const int steps = z1 - z0 + 4;               // window planes z0 - 2 .. z1 + 1
        if constexpr (ROLE == 0) {
            // ---------------- loaders: round r (behind barrier B_r) brings what step r + D reads ----------------
            const bool do_x = true, do_d = true;
            const uint32_t lane16 = (uint32_t)lane * 16u;
            auto pieces = [&](const double* src0, int64_t first, double* dst0, int n_pieces, auto aux_tag) {
                constexpr int AUX = decltype(aux_tag)::value;
                const char* s8 = reinterpret_cast<const char*>(src0);
                char* d8 = reinterpret_cast<char*>(dst0);
                if (first >= 0 && first + (int64_t)(n_pieces - 1) * 128 + 126 <= emax) {
                    const char* base = s8 + first * 8;
                    for (int c = 0; c < n_pieces; ++c)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (int64_t)c * 1024 + lane16),
                                                         (__attribute__((address_space(3))) void*)(d8 + c * 1024), 16, 0, AUX);
                } else {
                    for (int c = 0; c < n_pieces; ++c) {
                        int64_t e = first + c * 128 + 2 * lane;
                        e = e < 0 ? 0 : (e > emax ? emax : e);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s8 + e * 8),
                                                         (__attribute__((address_space(3))) void*)(d8 + c * 1024), 16, 0, AUX);
                    }
                }
            };
            auto has_d = [&](int st) { return WD && st >= 4 && st < steps; };     // step st finishes the rows of plane z0 + st - 4
            auto round_size = [&](int r) {
                const int st = r + D;
                if (st >= steps || (dbg & 4)) return 0;
                return (do_x ? g.G : 0) + (do_d && has_d(st) ? g.Gd : 0);
            };
            auto issue_round = [&](int r) {
                const int st = r + D;
                if (st >= steps || (dbg & 4)) return;
                const int sl = st % NS;
                const int64_t jz = z0 - 2 + st;
                if (do_x) pieces(x, (jz * NY + Y0 - 2) * SX - 2, ring + sl * g.slot, g.G, std::integral_constant<int, 0>());
                if (do_d && has_d(st)) pieces(rvec, ((jz - 2) * NY + Y0) * SX, dring + sl * g.dslot, g.Gd, std::integral_constant<int, LM_DC_AUX>());
            };
            for (int r = -D; r < 0; ++r) issue_round(r);
            for (int st = 0; st < steps; ++st) {
                int behind = 0;
                for (int r = st - D + 1; r < st; ++r) behind += round_size(r);
                box_wait_vm(behind);
                box_barrier();                       // B_st
                issue_round(st);
            }
            box_barrier();
        } else if constexpr (ROLE == 2) {
            for (int st = 0; st <= steps; ++st) box_barrier();
        } else {
            // ---------------- line waves: wave w = line w of the patch ----------------
            const lm_lds_pairs ring3 = (lm_lds_pairs)ring, dring3 = (lm_lds_pairs)dring;
            double AE[RP][4], AO[RP][4], zE[RP][2], zO[RP][2];
#pragma unroll
            for (int i = 0; i < RP; ++i) {
#pragma unroll
                for (int k = 0; k < 4; ++k) AE[i][k] = AO[i][k] = 0.0;
                zE[i][0] = zE[i][1] = zO[i][0] = zO[i][1] = 0.0;
            }
            const int il = wave / g.WPL, piece0 = (wave - il * g.WPL) * RP, Y = Y0 + il;     // (a line of more than two pieces: several waves)
            const bool live = Y < NY;                // (uniform)
            // (the number of the line's list: asked for a step ahead)
            const int32_t* __restrict__ idp = sl_line + (int64_t)z0 * g.NYP + (live ? Y : NY - 1);       // (step st: window plane z0 - 2 + st, entry jz + 2)
            int id_next = idp[0];
            for (int st = 0; st < steps; ++st) {
                const int id_now = id_next;
                if (st + 1 < steps) id_next = idp[(int64_t)(st + 1) * g.NYP];
                box_barrier();                       // B_st
                if (!live || (dbg & 2)) continue;
                const int sl = st % NS;
                const int jz = z0 - 2 + st;
                const bool fin = st >= 4;            // the rows of plane jz - 2 belong to this unit
                const lm_lds_pairs W = ring3 + sl * (g.slot >> 1) + (il + 2) * HP + 1, Dp = dring3 + sl * (g.dslot >> 1) + il * HP;
                const double* __restrict__ cp = SL + (int64_t)id_now * LM_SL;
                lm_lds_pairs Wc[RP];
                int lq[RP];
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const int lp = (piece0 + i) * 64 + lane;
                    lq[i] = lp < HP ? lp : HP - 1;
                    Wc[i] = W + lq[i];
                }
                double oE[RP], oO[RP], cE[RP], cO[RP];
                const int par = (Y & 1) * 2 + (jz & 1);          // (uniform)
                if (par == 0) lm_line_step<0, 0, RP>(Wc, HP, cp, AE, AO, oE, oO, cE, cO);
                else if (par == 1) lm_line_step<0, 1, RP>(Wc, HP, cp, AE, AO, oE, oO, cE, cO);
                else if (par == 2) lm_line_step<1, 0, RP>(Wc, HP, cp, AE, AO, oE, oO, cE, cO);
                else lm_line_step<1, 1, RP>(Wc, HP, cp, AE, AO, oE, oO, cE, cO);
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    if (fin) {
                        const int lp = (piece0 + i) * 64 + lane, X = 2 * lp;
                        const bool stE = lp < HP && X >= LM_LO && X <= SX - 1 - LM_HI, stO = lp < HP && X + 1 >= LM_LO && X + 1 <= SX - 1 - LM_HI;
                        const int64_t row = ((int64_t)(jz - 2) * NY + Y) * SX + X;
                        lm_v2d ri = lm_v2d{0.0, 0.0};
                        if (WD) ri = Dp[lq[i]];
                        if (stE && stO) *reinterpret_cast<lm_v2d*>(y + row) = lm_v2d{oE[i], oO[i]};
                        else if (stE) y[row] = oE[i];
                        else if (stO) y[row + 1] = oO[i];
                        if (stE) {
                            const double z = zE[i][1], a = oE[i];
                            if (DOTS == 1) { d_rz += ri.x * z; d_wz += a * z; d_rr += ri.x * ri.x; }
                            else if (DOTS == 2) { d_rz += a * ri.x; d_wz += a * a; d_rr += ri.x * ri.x; }
                            else if (DOTS == 3) { d_rz += z * z; d_wz += a * z; d_rr += ri.x * z * z; }
                        }
                        if (stO) {
                            const double z = zO[i][1], a = oO[i];
                            if (DOTS == 1) { d_rz += ri.y * z; d_wz += a * z; d_rr += ri.y * ri.y; }
                            else if (DOTS == 2) { d_rz += a * ri.y; d_wz += a * a; d_rr += ri.y * ri.y; }
                            else if (DOTS == 3) { d_rz += z * z; d_wz += a * z; d_rr += ri.y * z * z; }
                        }
                    }
                    zE[i][1] = zE[i][0]; zE[i][0] = cE[i];
                    zO[i][1] = zO[i][0]; zO[i][0] = cO[i];
                }
            }
            box_barrier();
        }
    }
    };
    if (wave >= LM_LW) run(std::integral_constant<int, 0>());
    else if (wave >= g.PY * g.WPL) run(std::integral_constant<int, 2>());
    else run(std::integral_constant<int, 1>());
}

// Patches and chunks: PY lines per patch, `slots` workgroups resident on the chip.
static inline void lm_cut(lm_geom* g, int64_t SX, int64_t NY, int64_t NZ, int PY, int slots) {
    g->n = SX * NY * NZ;
    g->SX = (int32_t)SX; g->NY = (int32_t)NY; g->NZ = (int32_t)NZ;
    g->HP = (int32_t)(SX / 2);
    g->NH = (g->HP + 63) / 64;
    g->WPL = (g->NH + 1) / 2;
    g->PY = PY;
    g->NP = (int32_t)((NY + PY - 1) / PY);
    const int64_t need = (int64_t)(PY + 4) * SX + 4;            // (a pair in front of the first line, one behind the last)
    g->G = (int32_t)((need + 127) / 128);
    g->slot = g->G * 128;
    g->Gd = (int32_t)(((int64_t)PY * SX + 127) / 128);
    g->dslot = g->Gd * 128;
    g->LZ = (int32_t)NZ + 2 * LM_ZPAD;
    g->NS2 = (int32_t)NZ + 4;
    g->NYP = (int32_t)NY + LM_LW;
    int ZC = slots / g->NP;
    if (ZC < 1) ZC = 1;
    if (ZC > NZ) ZC = (int)NZ;
    g->ZC = ZC;
    g->units = g->NP * ZC;
    g->upx = (g->units + 7) / 8;
    g->grid = 8 * g->upx;
}
static inline size_t lm_lds_bytes(const lm_geom& g, int D, bool weights) {
    return ((size_t)(D + 1) * g.slot + (weights ? (size_t)(D + 1) * g.dslot : 0)) * sizeof(double);
}

// ---- the tables -----------------------------------------------------------------------------------------------------------------
// used[class 8 + parity] = 1 for every (class, parity class) that occurs
__global__ void k_lm_used(int64_t n, int64_t SX, int64_t NY, const uint16_t* __restrict__ cls, int32_t* __restrict__ used) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; r < n; r += stride) {
        const int64_t X = r % SX, q = r / SX, Y = q % NY, Z = q / NY;
        const int key = (int)cls[r] * 8 + (int)((X & 1) + 2 * (Y & 1) + 4 * (Z & 1));
        if (used[key] == 0) used[key] = 1;
    }
}
// coefficient rows: the list of class c (cnt, off, coef: k_lat_table) spread over the stencil positions of parity class p; an entry
// that is no position of that stencil: info[1] += 1
__global__ void k_lm_coef(int ncls, int64_t SX, int64_t NY, const int32_t* __restrict__ used, const int32_t* __restrict__ cnt, const int32_t* __restrict__ off,
                          const double* __restrict__ coef, int list_pitch, double* __restrict__ coefS, int* __restrict__ info) {
    const int key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= ncls * 8 || !used[key]) return;
    const int c = key >> 3, p = key & 7;
    const int64_t plane = SX * NY;
    double* __restrict__ out = coefS + (int64_t)key * LM_CS;
    int bad = 0;
    const int m = cnt[c];
    for (int k = 0; k < m; ++k) {
        const int64_t o = off[(int64_t)c * list_pitch + k];
        const int64_t dz = (o + (o >= 0 ? plane / 2 : -(plane / 2))) / plane;
        const int64_t rem = o - dz * plane;
        const int64_t dy = (rem + (rem >= 0 ? SX / 2 : -(SX / 2))) / SX;
        const int64_t dx = rem - dy * SX;
        int at = -1;
        for (int s = 0; s < LM_CNT[p]; ++s)
            if (LM_DX[p][s] == dx && LM_DY[p][s] == dy && LM_DZ[p][s] == dz) at = s;
        if (at < 0) ++bad;
        else out[at] = coef[(int64_t)c * list_pitch + k];
    }
    if (bad) atomicAdd(&info[1], bad);
}
// the row numbers of the lines: lc[Y LZ + Z + LM_ZPAD] = even column | odd column << 16 of the interior of line (Y, Z) - the class of
// its first interior row of either parity, every interior row compared with it: info[2] += 1 per row of another class -, the zero
// row for Z outside the lattice
__global__ void k_lm_lines(lm_geom g, int zrow, const uint16_t* __restrict__ cls, uint32_t* __restrict__ lc, int* __restrict__ info) {
    const int lane = threadIdx.x & 63;
    const int64_t n_lines = (int64_t)g.NY * g.LZ;
    int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; q < n_lines; q += stride) {
        const int Y = (int)(q / g.LZ), zi = (int)(q - (int64_t)Y * g.LZ), Z = zi - LM_ZPAD;
        const bool in = Z >= 0 && Z < g.NZ;
        const int64_t base = in ? ((int64_t)Z * g.NY + Y) * g.SX : 0;
        const int pyz = 2 * (Y & 1) + 4 * (Z & 1);
        uint32_t rowE = (uint32_t)zrow, rowO = (uint32_t)zrow;
        if (in) {
            // first interior rows: X = LM_LO (even) and LM_LO + 1
            const int cE = cls[base + LM_LO], cO = cls[base + LM_LO + 1];
            rowE = (uint32_t)(cE * 8 + pyz);
            rowO = (uint32_t)(cO * 8 + pyz + 1);
            int bad = 0;
            for (int X = LM_LO + lane; X <= g.SX - 1 - LM_HI; X += 64)
                if ((int)cls[base + X] != ((X & 1) ? cO : cE)) ++bad;
            if (bad) atomicAdd(&info[2], bad);
        }
        if (lane == 0) lc[q] = rowE | (rowO << 16);
    }
}
// A (line, step) TUPLE: number t = Y NS2 + js (js = jz + 2); its key: the variant (Y & 1, jz & 1) and the row number words of planes
// jz - 2 .. jz + 2.
__device__ __forceinline__ const uint32_t* lm_tuple_words(const lm_geom& g, int64_t t, const uint32_t* __restrict__ lc, int* variant) {
    const int js = (int)(t % g.NS2), Y = (int)(t / g.NS2);
    const int jz = js - 2;
    *variant = (Y & 1) * 2 + (jz & 1);
    return lc + (int64_t)Y * g.LZ + (jz - 2 + LM_ZPAD);
}
// rep[t]: the first-come tuple with t's key (tab: open addressing over tuple numbers, -1 = free)
__global__ void k_lm_dedupe(lm_geom g, int64_t n_tuples, const uint32_t* __restrict__ lc, int32_t* __restrict__ tab, uint32_t tab_mask, int32_t* __restrict__ rep) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n_tuples; t += stride) {
        int v;
        const uint32_t* w = lm_tuple_words(g, t, lc, &v);
        uint32_t h = 0x9e3779b9u * (uint32_t)(v + 1);
#pragma unroll
        for (int m = 0; m < 5; ++m) { h ^= w[m] + 0x7f4a7c15u + (h << 6) + (h >> 2); h *= 0x85ebca6bu; h ^= h >> 15; }
        for (uint32_t probe = h & tab_mask;; probe = (probe + 1) & tab_mask) {
            int32_t cur = __hip_atomic_load(&tab[probe], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur < 0) {
                cur = atomicCAS(&tab[probe], -1, (int32_t)t);
                if (cur < 0) { rep[t] = (int32_t)t; break; }
            }
            int v2;
            const uint32_t* w2 = lm_tuple_words(g, cur, lc, &v2);
            if (v2 == v && w2[0] == w[0] && w2[1] == w[1] && w2[2] == w[2] && w2[3] == w[3] && w2[4] == w[4]) { rep[t] = cur; break; }
        }
    }
}
// list numbers for the first-come tuples (info[3]: their count)
__global__ void k_lm_number(int64_t n_tuples, const int32_t* __restrict__ rep, int32_t* __restrict__ num, int* __restrict__ info) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n_tuples; t += stride)
        if (rep[t] == (int32_t)t) num[t] = atomicAdd(&info[3], 1);
}
// sl_line: the list number of every tuple; the lists of the first-come tuples from the coefficient rows
__global__ void k_lm_fill(lm_geom g, int64_t n_tuples, const uint32_t* __restrict__ lc, const int32_t* __restrict__ rep, const int32_t* __restrict__ num,
                          const double* __restrict__ coefS, int32_t* __restrict__ sl_line, double* __restrict__ SL, int n_lists_cap) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n_tuples; t += stride) {
        const int id = num[rep[t]];
        const int js = (int)(t % g.NS2), Y = (int)(t / g.NS2);
        sl_line[(int64_t)js * g.NYP + Y] = id;
        if (rep[t] != (int32_t)t || id >= n_lists_cap) continue;
        int v;
        const uint32_t* w = lm_tuple_words(g, t, lc, &v);
        const int py = v >> 1, jzb = v & 1;
        double* __restrict__ out = SL + (int64_t)id * LM_SL;
        int p = 0;
        for (int dy = -2; dy <= 2; ++dy)        // (the order of lm_make_variant)
            for (int k = 0; k < 5; ++k)
                for (int c = 0; c < 2; ++c) {
                    const int par = c + 2 * py + 4 * ((jzb + k) & 1);
                    const uint32_t ww = w[4 - k];
                    const double* __restrict__ row = coefS + (int64_t)(c ? ww >> 16 : ww & 0xffffu) * LM_CS;
                    for (int s = LM_SLICE[par][k]; s < LM_SLICE[par][k + 1]; ++s)
                        if (LM_DY[par][s] == dy) out[p++] = row[s];
                }
        for (; p < LM_SL; ++p) out[p] = 0.0;
    }
}
