// Round-6 probe: k_box_spmv (fenicssolver_amd/csrc/fs_box.h - the production kernel, included) on a synthetic Kuhn-box operator,
// against a one-row-per-lane reference with the same fma order (bit comparison) and on the clock:
//   alone (back to back: x stays in the Infinity Cache at 10 M rows) and INSIDE an iteration-like sequence, a 72 B/row update kernel
//   between two products (the state of the caches the CG iteration leaves), each product bracketed by its own events.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fenicssolver_amd/csrc -o /tmp/box_probe tools/probes/box_spmv_probe.hip && /tmp/box_probe [m] [variants]
#include "fs_box.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <algorithm>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k_ref(box_geom g, const uint16_t* __restrict__ cls, const double* __restrict__ dict, const double* __restrict__ x,
                                             double* __restrict__ y, const double* __restrict__ rvec, double* __restrict__ sums) {
    const int64_t off[15] = {-(g.a + g.b + 1), -(g.a + g.b), -(g.b + 1), -g.b, -(g.a + 1), -g.a, -1, 0, 1, g.a, g.a + 1, g.b, g.b + 1, g.a + g.b, g.a + g.b + 1};
    double s0 = 0, s1 = 0, s2 = 0;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < g.n; r += (int64_t)gridDim.x * 256) {
        const double* cf = dict + (int64_t)cls[r] * g.S;
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < 15; ++t) {
            int64_t c = r + off[t];
            c = c < 0 ? 0 : (c > g.n - 1 ? g.n - 1 : c);
            acc = fma(cf[g.pos[t]], x[c], acc);
        }
        y[r] = acc;
        const double z = x[r], ri = rvec[r];
        s0 += z * z; s1 += acc * z; s2 += ri * z * z;
    }
    atomicAdd(&sums[0], s0); atomicAdd(&sums[1], s1); atomicAdd(&sums[2], s2);
}

__global__ void __launch_bounds__(256) k_update(int64_t n, double alpha, double beta, const double* __restrict__ w, double* __restrict__ r,
                                                double* __restrict__ p, double* __restrict__ s, double* __restrict__ xx) {
    const int64_t n2 = n >> 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        v2d rv = reinterpret_cast<v2d*>(r)[i], pv = reinterpret_cast<v2d*>(p)[i], sv = reinterpret_cast<v2d*>(s)[i], xv = reinterpret_cast<v2d*>(xx)[i];
        const v2d wv = __builtin_nontemporal_load(&reinterpret_cast<const v2d*>(w)[i]);
        pv = rv + beta * pv; sv = wv + beta * sv; xv = xv + alpha * pv; rv = rv - alpha * sv;
        __builtin_nontemporal_store(pv, &reinterpret_cast<v2d*>(p)[i]); __builtin_nontemporal_store(sv, &reinterpret_cast<v2d*>(s)[i]);
        __builtin_nontemporal_store(xv, &reinterpret_cast<v2d*>(xx)[i]); reinterpret_cast<v2d*>(r)[i] = rv;
    }
}

// what a plain grid-stride kernel reaches on the same bytes (x, weights: 8 B, classes: 2 B read; 8 B written per row)
__global__ void __launch_bounds__(256) k_stream(int64_t n, const double* __restrict__ x, const double* __restrict__ dv, const uint16_t* __restrict__ cls, double* __restrict__ y) {
    const int64_t n8 = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 c = reinterpret_cast<const uint4*>(cls)[i];
        v2d o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const v2d a = reinterpret_cast<const v2d*>(x)[4 * i + t], b = reinterpret_cast<const v2d*>(dv)[4 * i + t];
            const unsigned w = t == 0 ? c.x : (t == 1 ? c.y : (t == 2 ? c.z : c.w));
            o[t].x = a.x * b.x + (double)(w & 0xffffu); o[t].y = a.y * b.y + (double)(w >> 16);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) reinterpret_cast<v2d*>(y)[4 * i + t] = o[t];
    }
}

struct variant { const char* name; int cw, rp, d, nl; void (*kern)(box_geom, const uint16_t*, const double*, int, const double*, double*, const double*, double*, int*, int, int, int); };

int main(int argc, char** argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 216;
    const char* only = argc > 2 ? argv[2] : "";
    const int64_t n = (int64_t)m * m * m;
    const int a = m;
    const int64_t b = (int64_t)m * m;
    // plan round of the Kuhn list
    int32_t starts[8] = {0, (int32_t)-(a + b + 1), (int32_t)-(b + 1), -(a + 1), -1, a, (int32_t)b, (int32_t)(a + b)};
    uint8_t lens[8] = {0, 2, 2, 2, 3, 2, 2, 2};
    box_geom g0;
    if (!box_recognize(starts, lens, 8, n, 3, &g0)) { printf("not recognised\n"); return 1; }
    g0.S = 24;
    // classes by position: 0, 1, interior, m - 1 per axis -> 64 classes; coefficients zero where the neighbour does not exist
    const int ncls = 64;
    auto cat = [&](int i) { return i == 0 ? 0 : (i == 1 ? 1 : (i == m - 1 ? 3 : 2)); };
    const int dxyz[15][3] = {{-1, -1, -1}, {0, -1, -1}, {-1, 0, -1}, {0, 0, -1}, {-1, -1, 0}, {0, -1, 0}, {-1, 0, 0}, {0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {1, 1, 0}, {0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
    std::vector<double> dict(ncls * 24, 0.0);
    for (int c = 0; c < ncls; ++c) {
        const int ci = c & 3, cj = (c >> 2) & 3, ck = c >> 4;
        for (int t = 0; t < 15; ++t) {
            const int di = dxyz[t][0], dj = dxyz[t][1], dk = dxyz[t][2];
            const bool missing = (ci == 0 && di < 0) || (ci == 3 && di > 0) || (cj == 0 && dj < 0) || (cj == 3 && dj > 0) || (ck == 0 && dk < 0) || (ck == 3 && dk > 0);
            dict[c * 24 + g0.pos[t]] = missing ? 0.0 : (t == 7 ? 1.0 : -0.04 - 0.003 * ((c * 15 + t) % 11));
        }
    }
    std::vector<uint16_t> cls(n + 64, 0);     // (slack: the loader reads whole 16-byte groups)
    std::vector<double> x(n + 2, 0.0), dv(n + 2, 0.0);
    for (int64_t r = 0; r < n; ++r) {
        const int i = r % m, j = (r / m) % m, k = r / b;
        cls[r] = (uint16_t)(cat(i) | (cat(j) << 2) | (cat(k) << 4));
        x[r] = std::sin(0.001 * r) + 0.1;
        dv[r] = 1.0 + 0.001 * (r % 13);
    }
    double *d_x, *d_y, *d_yr, *d_dict, *d_dv, *d_part, *d_p, *d_s, *d_xx, *d_sums; uint16_t* d_cls; int* d_status;
    CK(hipMalloc(&d_x, (n + 2) * 8)); CK(hipMalloc(&d_y, (n + 2) * 8)); CK(hipMalloc(&d_yr, (n + 2) * 8)); CK(hipMalloc(&d_dict, dict.size() * 8)); CK(hipMalloc(&d_dv, (n + 2) * 8));
    CK(hipMalloc(&d_p, (n + 2) * 8)); CK(hipMalloc(&d_s, (n + 2) * 8)); CK(hipMalloc(&d_xx, (n + 2) * 8)); CK(hipMalloc(&d_sums, 64));
    CK(hipMalloc(&d_part, 3 * 8192 * 8)); CK(hipMalloc(&d_cls, cls.size() * 2)); CK(hipMalloc(&d_status, 64));
    CK(hipMemcpy(d_x, x.data(), (n + 2) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_dict, dict.data(), dict.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dv, dv.data(), (n + 2) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_cls, cls.data(), cls.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(d_p, 0, n * 8)); CK(hipMemset(d_s, 0, n * 8)); CK(hipMemset(d_xx, 0, n * 8)); CK(hipMemset(d_status, 0, 64)); CK(hipMemset(d_sums, 0, 64));
    hipLaunchKernelGGL(k_ref, dim3(2048), dim3(256), 0, 0, g0, d_cls, d_dict, d_x, d_yr, d_dv, d_sums);
    CK(hipDeviceSynchronize());
    std::vector<double> yr(n), yb(n);
    double sref[3];
    CK(hipMemcpy(yr.data(), d_yr, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(sref, d_sums, 24, hipMemcpyDeviceToHost));
    printf("m = %d, n = %lld rows, a = %d, b = %lld; reference sums %.15g %.15g %.15g\n", m, (long long)n, a, (long long)b, sref[0], sref[1], sref[2]);

    const variant vs[] = {
        {"cw4_rp4_d2_nl1", 4, 4, 2, 1, k_box_spmv<3, 4, 4, 2, 1>}, {"cw4_rp4_d2_nl2", 4, 4, 2, 2, k_box_spmv<3, 4, 4, 2, 2>},
        {"cw6_rp2_d2_nl2", 6, 2, 2, 2, k_box_spmv<3, 6, 2, 2, 2>}, {"cw6_rp2_d3_nl2", 6, 2, 3, 2, k_box_spmv<3, 6, 2, 3, 2>},
        {"cw6_rp2_d2_nl1", 6, 2, 2, 1, k_box_spmv<3, 6, 2, 2, 1>}, {"cw5_rp2_d2_nl1", 5, 2, 2, 1, k_box_spmv<3, 5, 2, 2, 1>},
        {"cw8_rp2_d2_nl2", 8, 2, 2, 2, k_box_spmv<3, 8, 2, 2, 2>}, {"cw8_rp4_d2_nl2", 8, 4, 2, 2, k_box_spmv<3, 8, 4, 2, 2>},
        {"cw6_rp3_d2_nl2", 6, 3, 2, 2, k_box_spmv<3, 6, 3, 2, 2>}, {"cw8_rp3_d2_nl2", 8, 3, 2, 2, k_box_spmv<3, 8, 3, 2, 2>},
        {"cw4_rp4_d2_nl1_nodots", 4, 4, 2, 1, k_box_spmv<0, 4, 4, 2, 1>},
    };
    const int reps = m > 300 ? 20 : 100;
    std::vector<hipEvent_t> ev(2 * reps + 2);
    for (auto& evt : ev) CK(hipEventCreate(&evt));
    {
        float ms = 0;
        for (int grid : {1024, 2048, 4096}) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, n, d_x, d_dv, d_cls, d_y);
            CK(hipEventRecord(ev[0]));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, n, d_x, d_dv, d_cls, d_y);
            CK(hipEventRecord(ev[1])); CK(hipEventSynchronize(ev[1])); CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
            const double ta = ms * 1e3 / reps;
            double tot = 0;
            for (int i = 0; i < reps; ++i) {
                hipLaunchKernelGGL(k_update, dim3(2048), dim3(256), 0, 0, n, 1e-9, 0.5, d_y, d_x, d_p, d_s, d_xx);
                CK(hipEventRecord(ev[2 + 2 * i]));
                hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, n, d_x, d_dv, d_cls, d_y);
                CK(hipEventRecord(ev[3 + 2 * i]));
            }
            CK(hipDeviceSynchronize());
            for (int i = 0; i < reps; ++i) { CK(hipEventElapsedTime(&ms, ev[2 + 2 * i], ev[3 + 2 * i])); tot += ms * 1e3; }
            CK(hipEventRecord(ev[0]));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_update, dim3(2048), dim3(256), 0, 0, n, 1e-9, 0.5, d_y, d_x, d_p, d_s, d_xx);
            CK(hipEventRecord(ev[1])); CK(hipEventSynchronize(ev[1])); CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
            printf("streaming kernel on the same 26 B/row, grid %4d: alone %7.2f us (%.3f)  in iteration %7.2f us (%.3f);  update kernel (72 B/row) alone %7.2f us (%.3f)\n", grid, ta, 26.0 * n / ta * 1e-6 / 8.0,
                   tot / reps, 26.0 * n / (tot / reps) * 1e-6 / 8.0, ms * 1e3 / reps, 72.0 * n / (ms * 1e3 / reps) * 1e-6 / 8.0);
        }
        CK(hipMemcpy(d_x, x.data(), (n + 2) * 8, hipMemcpyHostToDevice));
    }
    for (const variant& v : vs) {
        if (only[0] && !strstr(v.name, only)) continue;
        for (int passes = 1; passes <= (m > 300 ? 3 : 2); ++passes) {
            box_geom g = g0;
            const int Lmax = v.cw * 64 * v.rp;
            // workgroups per CU by LDS (160 KB) - first cut with one pass to learn the slot size
            box_cut(&g, Lmax, 256, 1);
            size_t lds = box_lds_bytes(g, ncls, v.d, !strstr(v.name, "nodots"));
            int per_cu = (int)std::min<size_t>((160u << 10) / (lds + 512), 2048 / ((v.cw + v.nl) * 64));
            if (per_cu < 1) { printf("%-20s window does not fit LDS (%zu bytes)\n", v.name, lds); break; }
            if (per_cu > 4) per_cu = 4;
            box_cut(&g, Lmax, 256 * per_cu, passes);
            lds = box_lds_bytes(g, ncls, v.d, !strstr(v.name, "nodots"));
            if ((g.G + g.dslot / 128 + g.cslot / 512) * (v.d - 1) > 62) { printf("%-20s too many pieces per round\n", v.name); break; }
            CK(hipFuncSetAttribute((const void*)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            auto launch = [&] { hipLaunchKernelGGL(v.kern, dim3(g.grid), dim3((v.cw + v.nl) * 64), lds, 0, g, d_cls, d_dict, ncls, d_x, d_y, d_dv, d_part, d_status, 0, g.grid, 0); };
            CK(hipMemset(d_y, 0xff, n * 8));
            launch();
            CK(hipDeviceSynchronize());
            CK(hipGetLastError());
            CK(hipMemcpy(yb.data(), d_y, n * 8, hipMemcpyDeviceToHost));
            std::vector<double> part(3 * g.grid);
            CK(hipMemcpy(part.data(), d_part, 3 * g.grid * 8, hipMemcpyDeviceToHost));
            int64_t diff = 0, first = -1;
            for (int64_t r = 0; r < n; ++r) if (memcmp(&yr[r], &yb[r], 8)) { if (first < 0) first = r; ++diff; }
            double sb[3] = {0, 0, 0};
            for (int t = 0; t < 3; ++t) for (int q = 0; q < g.grid; ++q) sb[t] += part[t * g.grid + q];
            // alone
            for (int i = 0; i < 5; ++i) launch();
            CK(hipEventRecord(ev[0]));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(ev[1]));
            CK(hipEventSynchronize(ev[1]));
            float ms = 0; CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
            const double t_alone = ms * 1e3 / reps;
            // between updates
            for (int i = 0; i < reps; ++i) {
                hipLaunchKernelGGL(k_update, dim3(2048), dim3(256), 0, 0, n, 1e-9, 0.5, d_y, d_x, d_p, d_s, d_xx);     // (as in the CG: the product multiplies what the update wrote)
                CK(hipEventRecord(ev[2 + 2 * i]));
                launch();
                CK(hipEventRecord(ev[3 + 2 * i]));
            }
            CK(hipDeviceSynchronize());
            double tot = 0; std::vector<float> ts;
            for (int i = 0; i < reps; ++i) { CK(hipEventElapsedTime(&ms, ev[2 + 2 * i], ev[3 + 2 * i])); ts.push_back(ms * 1e3f); tot += ms * 1e3; }
            std::sort(ts.begin(), ts.end());
            CK(hipMemcpy(d_x, x.data(), (n + 2) * 8, hipMemcpyHostToDevice));
            printf("%-22s passes %d  L %4d P %3d ZC %3d grid %4d lds %6zu (%d/CU) G %2d | alone %7.2f us (%.2f TB/s, %.3f)  in iteration mean %7.2f median %7.2f us (%.3f) | rows differing %lld (first %lld)  sums rel %.1e %.1e %.1e\n",
                   v.name, passes, g.L, g.P, g.ZC, g.grid, lds, per_cu, g.G, t_alone, 26.0 * n / t_alone * 1e-6, 26.0 * n / t_alone * 1e-6 / 8.0,
                   tot / reps, ts[ts.size() / 2], 26.0 * n / ts[ts.size() / 2] * 1e-6 / 8.0, (long long)diff, (long long)first,
                   std::fabs(sb[0] - sref[0]) / std::fabs(sref[0]), std::fabs(sb[1] - sref[1]) / std::fabs(sref[1]), std::fabs(sb[2] - sref[2]) / std::fabs(sref[2]));
            fflush(stdout);
        }
    }
    return 0;
}
