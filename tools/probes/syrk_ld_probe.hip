// Developer probe: schur_syrk64_kernel (csrc/ba_schur.hpp) with the LDS row stride given at compile time (-DPCS_SYRK64_LD=N).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pycamset_amd/csrc -DPCS_SYRK64_LD=33 -o /tmp/syrk33 tools/probes/syrk_ld_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ba_schur.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv) {
    const int n = 1680, k = 1458, nb = (n + 63) / 64, tiles = nb * (nb + 1) / 2, ksplit = argc > 1 ? atoi(argv[1]) : 2, kchunk = ((k + ksplit - 1) / ksplit + 63) / 64 * 64;
    std::vector<double> V((size_t)n * k);
    for (size_t i = 0; i < V.size(); ++i) V[i] = (double)((i * 2654435761u) % 1000) * 1e-3 - 0.5;
    double *dV, *dS;
    CK(hipMalloc(&dV, V.size() * 8)); CK(hipMalloc(&dS, (size_t)n * n * 8));
    CK(hipMemcpy(dV, V.data(), V.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(dS, 0, (size_t)n * n * 8));
    pcs::SchurSyrkArgs a{dV, dS, nullptr, nullptr, n, k, k, n, ksplit, kchunk, nullptr};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(pcs::schur_syrk64_kernel, dim3(tiles * ksplit), dim3(256), 0, 0, a);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(pcs::schur_syrk64_kernel, dim3(tiles * ksplit), dim3(256), 0, 0, a);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> S((size_t)n * n);
    CK(hipMemcpy(S.data(), dS, S.size() * 8, hipMemcpyDeviceToHost));
    double ref = 0; for (int q = 0; q < k; ++q) ref += V[(size_t)700 * k + q] * V[(size_t)33 * k + q];
    printf("row stride %d doubles, K split %d x %d: %.1f us per launch; S[700][33] / -23 = %.6f, V V' there %.6f\n", PCS_SYRK64_LD, ksplit, kchunk, ms / 20 * 1e3, S[(size_t)700 * n + 33] / -23.0, ref);
    return 0;
}
