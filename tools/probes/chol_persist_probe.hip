// Stand-alone check + timing of csrc/ba_chol_persist.hpp against a host Cholesky (developer tool):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pycamset_amd/csrc -o tools/probes/chol_persist_probe tools/probes/chol_persist_probe.hip
//   tools/probes/chol_persist_probe [n ...]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <climits>
#include <random>
#include <vector>

#define CP_TRACE 1
#include "ba_chol_persist.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

static int run(int n, int reps, int pipe) {
    std::mt19937_64 rng(n);
    std::normal_distribution<double> nd;
    const int k = n + 5;
    std::vector<double> Gm((size_t)n * k), S((size_t)n * n), rhs(n), x(n);
    for (auto &v : Gm) v = nd(rng);
    std::vector<double> d(n);
    for (auto &v : d) v = std::pow(10.0, std::uniform_real_distribution<double>(-1, 1)(rng));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0;
            for (int q = 0; q < k; ++q) s += Gm[(size_t)i * k + q] * Gm[(size_t)j * k + q];
            if (i == j) s += 1e-3;
            S[(size_t)i * n + j] = s * d[i] * d[j];
            if (j < i) S[(size_t)j * n + i] = std::nan("");
        }
    for (auto &v : rhs) v = nd(rng);
    // host reference: Cholesky + solves
    std::vector<double> L((size_t)n * n, 0.0), y(n), xr(n);
    for (int j = 0; j < n; ++j) {
        double s = S[(size_t)j * n + j];
        for (int q = 0; q < j; ++q) s -= L[(size_t)j * n + q] * L[(size_t)j * n + q];
        const double l = std::sqrt(s);
        L[(size_t)j * n + j] = l;
        for (int i = j + 1; i < n; ++i) {
            double t = S[(size_t)i * n + j];
            for (int q = 0; q < j; ++q) t -= L[(size_t)i * n + q] * L[(size_t)j * n + q];
            L[(size_t)i * n + j] = t / l;
        }
    }
    for (int i = 0; i < n; ++i) { double t = rhs[i]; for (int q = 0; q < i; ++q) t -= L[(size_t)i * n + q] * y[q]; y[i] = t / L[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double t = y[i]; for (int q = i + 1; q < n; ++q) t -= L[(size_t)q * n + i] * xr[q]; xr[i] = t / L[(size_t)i * n + i]; }

    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    if (!pcs::cp_fits(n, cus)) { printf("n = %d does not fit\n", n); return 0; }
    const int64_t nb = (n + 31) / 32;
    double *dS, *dS0, *drhs, *dx, *dwork;
    int32_t *dstatus;
    CK(hipMalloc(&dS, sizeof(double) * n * n)); CK(hipMalloc(&dS0, sizeof(double) * n * n)); CK(hipMalloc(&drhs, sizeof(double) * n));
    CK(hipMalloc(&dx, sizeof(double) * n)); CK(hipMalloc(&dwork, sizeof(double) * pcs::cp_work_doubles(nb))); CK(hipMalloc(&dstatus, 4));
    CK(hipMemcpy(dS0, S.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(drhs, rhs.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    CK(hipMemset(dstatus, 0, 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemcpyAsync(dS, dS0, sizeof(double) * n * n, hipMemcpyDeviceToDevice, s));
        CK(hipEventRecord(e0, s));
        CK(pcs::cp_launch(n, dS, n, drhs, dx, dwork, dstatus, cus, s, 0.05, nullptr));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) { best = ms < best ? ms : best; sum += ms; }
    }
    if (getenv("CP_SHOW_TRACE")) {
        const int64_t T = pcs::cp_tiles(nb);
        const int G = (int)(T < cus ? T : cus);
        int64_t *dtr;
        const size_t tl = (size_t)G * (nb + 1) * 8;
        CK(hipMalloc(&dtr, tl * 8));
        CK(hipMemset(dtr, 0, tl * 8));
        CK(hipMemcpyAsync(dS, dS0, sizeof(double) * n * n, hipMemcpyDeviceToDevice, s));
        CK(pcs::cp_launch(n, dS, n, drhs, dx, dwork, dstatus, cus, s, 0.05, dtr));
        CK(hipStreamSynchronize(s));
        std::vector<int64_t> tr(tl);
        CK(hipMemcpy(tr.data(), dtr, tl * 8, hipMemcpyDeviceToHost));
        int64_t t0 = INT64_MAX;
        for (auto v : tr) if (v > 0 && v < t0) t0 = v;
        printf("  column: [us since first stamp] last wait end | update (max) | factor (max) | publish (max) | published at (max) ; owners\n");
        for (int j = 0; j < nb; ++j) {
            double wend = 0, upd = 0, fac = 0, pub = 0, pat = 0; int owners = 0;
            for (int w = 0; w < G; ++w) {
                const int64_t *q = &tr[((size_t)w * (nb + 1) + j) * 8];
                if (!q[2]) continue;
                ++owners;
                if (!q[4]) continue;   // the diagonal tile's owner: nobody waits for it
                wend = std::fmax(wend, (q[1] - t0) * 0.01); upd = std::fmax(upd, (q[2] - q[1]) * 0.01); fac = std::fmax(fac, (q[3] - q[2]) * 0.01);
                if (q[4]) { pub = std::fmax(pub, (q[4] - q[3]) * 0.01); pat = std::fmax(pat, (q[4] - t0) * 0.01); }
            }
            double bulk_max = 0, bulk_sum = 0, wait_max = 0; int nw = 0;
            for (int w = 0; w < G; ++w) {
                const int64_t *q = &tr[((size_t)w * (nb + 1) + j) * 8];
                if (!q[5]) continue;
                const int64_t from = q[4] ? q[4] : q[1];
                bulk_max = std::fmax(bulk_max, (q[5] - from) * 0.01); bulk_sum += (q[5] - from) * 0.01; ++nw;
                wait_max = std::fmax(wait_max, (q[1] - q[0]) * 0.01);
            }
            {   // the workgroup that publishes this column last, and where its time went (its stamps of this and the previous column)
                int wl = -1; int64_t tl = 0;
                for (int w = 0; w < G; ++w) { const int64_t *q = &tr[((size_t)w * (nb + 1) + j) * 8]; if (q[4] > tl) { tl = q[4]; wl = w; } }
                if (wl >= 0 && getenv("CP_SHOW_LAST")) {
                    const int64_t *q = &tr[((size_t)wl * (nb + 1) + j) * 8];
                    printf("    last: wg %3d  start %7.2f  crit applied %7.2f  factored %7.2f  published %7.2f  bulk done %7.2f", wl, (q[1] - t0) * 0.01, (q[2] - t0) * 0.01, (q[3] - t0) * 0.01, (q[4] - t0) * 0.01, q[5] ? (q[5] - t0) * 0.01 : 0.0);
                    if (j > 0) { const int64_t *p = &tr[((size_t)wl * (nb + 1) + j - 1) * 8]; printf("  | previous column: start %7.2f  bulk done %7.2f", p[1] ? (p[1] - t0) * 0.01 : 0.0, p[5] ? (p[5] - t0) * 0.01 : 0.0); }
                    printf("\n");
                }
            }
            printf("  col %2d: wait end %7.2f | update %5.2f | factor %5.2f | publish %5.2f | at %7.2f ; %d | bulk max %5.2f mean %5.2f | longest wait %5.2f (%d wgs)\n", j, wend, upd, fac, pub, pat, owners, bulk_max, nw ? bulk_sum / nw : 0.0, wait_max, nw);
        }
        printf("  backward: block k: x_{k+1} seen at | x_k stored at (us since first stamp)\n");
        for (int k = nb - 1; k >= 0; --k)
            for (int w = 0; w < G; ++w) {
                const int64_t *q = &tr[((size_t)w * (nb + 1) + k) * 8];
                if (q[7]) printf("  k %2d: seen %7.2f  stored %7.2f\n", k, q[6] ? (q[6] - t0) * 0.01 : 0.0, (q[7] - t0) * 0.01);
            }
        {
            int64_t td[64];
            CK(hipMemcpyFromSymbol(td, HIP_SYMBOL(pcs::cp_tile_dbg), sizeof(td)));
            printf("  workgroup 100, trailing updates with column 3, per tile [us]: wait for operands | park + barrier | products | barrier   (start since the first tile's)\n");
            for (int q = 0; q < 8; ++q) if (td[q * 8 + 4]) printf("   slot %d: start %6.2f | %5.2f | %5.2f | %5.2f | %5.2f\n", q, (td[q * 8] - td[0]) * 0.01, (td[q * 8 + 1] - td[q * 8]) * 0.01, (td[q * 8 + 2] - td[q * 8 + 1]) * 0.01, (td[q * 8 + 3] - td[q * 8 + 2]) * 0.01, (td[q * 8 + 4] - td[q * 8 + 3]) * 0.01);
        }
        {
            int64_t dbg[64];
            CK(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(pcs::cp_dbg), sizeof(dbg)));
            printf("  last workgroup's last panel, [us since wave 0's entry] per wave: entry | earlier columns applied | own columns factored | stored\n");
            for (int w = 0; w < 4; ++w) {
                printf("   wave %d:", w);
                for (int q = 0; q < 4; ++q) printf(" %5.2f", dbg[w * 16 + q] ? (dbg[w * 16 + q] - dbg[0]) * 0.01 : -1.0);
                printf("\n");
            }
        }
        (void)hipFree(dtr);
    }
    int32_t st;
    CK(hipMemcpy(&st, dstatus, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(x.data(), dx, sizeof(double) * n, hipMemcpyDeviceToHost));
    std::vector<double> Ld((size_t)n * n);
    CK(hipMemcpy(Ld.data(), dS, sizeof(double) * n * n, hipMemcpyDeviceToHost));
    double ex = 0, mx = 0, eL = 0;
    for (int i = 0; i < n; ++i) { ex = std::fmax(ex, std::fabs(x[i] - xr[i])); mx = std::fmax(mx, std::fabs(xr[i])); }
    bool upper_ok = true;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (j <= i) eL = std::fmax(eL, std::fabs(Ld[(size_t)i * n + j] - L[(size_t)i * n + j]) / (std::fabs(L[(size_t)i * n + i]) + 1e-300));
            else if (!std::isnan(Ld[(size_t)i * n + j])) upper_ok = false;
        }
    const int64_t T = pcs::cp_tiles(nb);
    const int G = (int)(T < cus ? T : cus);
    printf("pipe %d  n %5d  nb %3d  tiles %5lld  wgs %3d  slots %d  lds %6zu B : status %d  |x - x_ref| / |x| %.2e  |L - L_ref| rel %.2e  upper untouched %d   time mean %.1f us  min %.1f us\n",
           pipe, n, (int)nb, (long long)T, G, (int)((T + G - 1) / G), pcs::cp_lds_bytes((int)((T + G - 1) / G)), st, ex / mx, eL, (int)upper_ok, sum / (reps - 1) * 1e3, best * 1e3);
    fflush(stdout);
    for (void *q : {(void *)dS, (void *)dS0, (void *)drhs, (void *)dx, (void *)dwork, (void *)dstatus}) (void)hipFree(q);
    return 0;
}

int main(int argc, char **argv) {
    std::vector<int> ns;
    for (int i = 1; i < argc; ++i) ns.push_back(atoi(argv[i]));
    if (ns.empty()) ns = {1, 31, 32, 33, 97, 480, 1003, 1680};
    for (int rep = 0; rep < 2; ++rep)
        for (int n : ns)
            for (int pipe = 0; pipe < 1; ++pipe)
                if (run(n, 12, pipe)) return 1;
    return 0;
}
