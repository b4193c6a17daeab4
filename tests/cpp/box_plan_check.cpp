// CPU check of the launch planning of k_box_spmv (fenicssolver_amd/csrc/fs_box.h: box_recognize, box_cut, box_lds_bytes) over many box
// shapes - the host logic of the marching-window product, compiled with hipcc (the header holds the kernel too) and run WITHOUT a GPU
// by tests/test_host_api.py.  Prints "ok <cases>" or the first violated invariant.
#include "fs_box.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

static int fail(const char* what, long a, long ny, long nz, int shape) {
    printf("FAILED %s at a=%ld ny=%ld nz=%ld shape=%d\n", what, a, ny, nz, shape);
    return 1;
}

int main() {
    long cases = 0;
    const long as[] = {3, 4, 5, 8, 21, 34, 71, 100, 101, 131, 216, 217, 320, 321, 400, 441, 1000, 2049};
    const long nys[] = {2, 3, 5, 9, 40, 216, 441};
    const long nzs[] = {2, 3, 6, 27, 100, 216, 441};
    for (long a : as) for (long ny : nys) for (long nz : nzs) {
        const int64_t b = (int64_t)a * ny, n = b * nz;
        if (n > (int64_t)2000000000 || b < 2 * a + 2) continue;
        int32_t starts[8] = {0, (int32_t)-(a + b + 1), (int32_t)-(b + 1), (int32_t)-(a + 1), -1, (int32_t)a, (int32_t)b, (int32_t)(a + b)};
        const uint8_t lens[8] = {0, 2, 2, 2, 3, 2, 2, 2};
        box_geom g0;
        if (!box_recognize(starts, lens, 8, n, 3, &g0)) return fail("box_recognize refused a Kuhn box", a, ny, nz, -1);
        if (g0.a != a || g0.b != b || g0.nz != nz) return fail("box_recognize geometry", a, ny, nz, -1);
        for (int t = 0; t < BOX_TERMS; ++t)
            if (g0.pos[t] < 3 || g0.pos[t] >= 24) return fail("coefficient position outside the plan layout", a, ny, nz, -1);
        // a list that is not the Kuhn list must be refused
        int32_t bad[8];
        for (int j = 0; j < 8; ++j) bad[j] = starts[j];
        bad[6] += 1;
        box_geom gb;
        if (box_recognize(bad, lens, 8, n, 3, &gb)) return fail("box_recognize accepted a wrong list", a, ny, nz, -1);
        for (int shape = 0; shape < 2; ++shape) {
            const int cw = shape == 0 ? 6 : 8, rp = shape == 0 ? 2 : 3, Lmax = cw * 64 * rp;
            for (int slots : {256, 512}) {
                box_geom g = g0;
                g.S = 24;
                box_cut(&g, Lmax, slots, 1);
                ++cases;
                if (g.L & 1 || g.L <= 0 || g.L > Lmax + 1) return fail("L", a, ny, nz, shape);
                if ((int64_t)g.P * g.L < b || (int64_t)(g.P - 1) * g.L >= b) return fail("patches do not tile the plane", a, ny, nz, shape);
                if (g.H & 1 || g.H < a + 1) return fail("H", a, ny, nz, shape);
                if (g.slot != 128 * g.G || g.slot < g.L + g.H + a + 3) return fail("window slot too small", a, ny, nz, shape);
                if (g.dslot % 128 || g.dslot < g.L + 2 || g.cslot % 512 || g.cslot < g.L + 8) return fail("weight / class slots", a, ny, nz, shape);
                if (g.ZC < 1 || g.ZC > nz || g.units != g.P * g.ZC) return fail("chunks", a, ny, nz, shape);
                if (g.grid % 8 || (int64_t)(g.grid / 8) * 8 < g.units || g.upx * 8 < g.units) return fail("grid does not cover the units", a, ny, nz, shape);
                // every plane in exactly one chunk
                int64_t covered = 0;
                for (int zc = 0; zc < g.ZC; ++zc) {
                    const int k0 = (int)((int64_t)zc * g.nz / g.ZC), k1 = (int)((int64_t)(zc + 1) * g.nz / g.ZC);
                    if (k1 <= k0) return fail("empty chunk", a, ny, nz, shape);
                    covered += k1 - k0;
                }
                if (covered != nz) return fail("planes covered", a, ny, nz, shape);
                // every unit reachable by the kernel's loop: u = xcd * upx + j, j < grid / 8
                std::vector<char> seen((size_t)g.units, 0);
                for (int blk = 0; blk < g.grid; ++blk) {
                    const int xcd = blk & 7, j0 = blk >> 3, ustep = g.grid >> 3;
                    const int u_end = (xcd + 1) * g.upx < g.units ? (xcd + 1) * g.upx : g.units;
                    for (int u = xcd * g.upx + j0; u < u_end; u += ustep) {
                        if (seen[(size_t)u]) return fail("unit taken twice", a, ny, nz, shape);
                        seen[(size_t)u] = 1;
                    }
                }
                for (char c : seen) if (!c) return fail("unit never taken", a, ny, nz, shape);
                if (box_lds_bytes(g, 88, 2, true) < (size_t)3 * g.slot * 8) return fail("lds bytes", a, ny, nz, shape);
            }
        }
    }
    printf("ok %ld\n", cases);
    return 0;
}
