// mfma_f64_probe.hip — measures, on the MI355X the build targets, what the FP64 matrix instructions cost and
// checks their operand layouts with exact integer data (the guide gives the 16x16x4 C/D map; the 4x4x4 4-block
// form is not documented there).  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using d4 = __attribute__((ext_vector_type(4))) double;

// `iters` rounds of NACC independent MFMAs (or FMAs) per wave; blockDim = 256 * waves-per-SIMD, one block per CU
template <int KIND, int NACC>
__global__ __launch_bounds__(1024) void rate_kernel(double *out, int iters, long long *cycles) {
    const double a = (double)(threadIdx.x & 15) * 0.5, b = (double)(threadIdx.x >> 4) * 0.25;
    d4 acc[NACC] = {};
    double s[NACC] = {};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            if constexpr (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
            else if constexpr (KIND == 1) s[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s[j], 0, 0, 0);
            else s[j] = __builtin_fma(a, b, s[j]);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double r = 0;
    for (int j = 0; j < NACC; ++j) r += s[j] + acc[j].x + acc[j].y + acc[j].z + acc[j].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// layout check: D = A * B with A[i][k], B[k][j] small integers chosen so that every (i, j) result is unique
__global__ void layout16(const double *A, const double *B, double *D) {   // A 16x4 row-major, B 4x16 row-major
    const int l = threadIdx.x;
    d4 acc = {};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];
}
__global__ void layout4(const double *a_lane, const double *b_lane, double *D) {   // raw per-lane operands
    const int l = threadIdx.x;
    D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_lane[l], b_lane[l], 0.0, 0, 0, 0);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND, int NACC>
static int run_rate(const char *name, double flop_per_inst, int waves_per_simd) {
    int dev = 0; hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
    const int blocks = p.multiProcessorCount, iters = 40000 / NACC, threads = 256 * waves_per_simd;
    double *out; long long *cyc; CK(hipMalloc(&out, sizeof(double) * blocks * threads)); CK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<KIND, NACC><<<blocks, threads>>>(out, 100, cyc);
    CK(hipEventRecord(e0)); rate_kernel<KIND, NACC><<<blocks, threads>>>(out, iters, cyc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double insts = (double)NACC * iters;   // per wave
    printf("%-24s acc %d waves/SIMD %d: %7.3f ms  %7.2f clk/inst/wave  %6.2f clk/inst/SIMD  %6.1f TFLOP/s chip\n", name, NACC, waves_per_simd,
           ms, (double)c / insts, (double)c / insts / waves_per_simd, insts * waves_per_simd * flop_per_inst * blocks * 4 / (ms * 1e-3) / 1e12);
    CK(hipFree(out)); CK(hipFree(cyc));
    return 0;
}

template <int KIND>
static int run_all(const char *name, double flop) {
    for (int w : {1, 2, 4}) {
        if (run_rate<KIND, 1>(name, flop, w)) return 1;
        if (run_rate<KIND, 2>(name, flop, w)) return 1;
        if (run_rate<KIND, 4>(name, flop, w)) return 1;
        if (run_rate<KIND, 8>(name, flop, w)) return 1;
    }
    return 0;
}

int main() {
    if (run_all<0>("v_mfma_f64_16x16x4_f64", 2048.0)) return 1;
    if (run_all<1>("v_mfma_f64_4x4x4_4b_f64", 512.0)) return 1;
    if (run_all<2>("v_fma_f64 (wave64)", 128.0)) return 1;
    // ---- 16x16x4 layout
    std::vector<double> A(64), B(64), D(256), Dd(256);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i + 20 * k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 3 + 7 * j + 1000 * k;
    double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    layout16<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(Dd.data(), dD, 2048, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int col = l & 15, row = (l >> 4) + 4 * r;   // guide: C/D col = lane & 15, row = (lane >> 4) + 4 * reg
        double ref = 0; for (int k = 0; k < 4; ++k) ref += A[row * 4 + k] * B[k * 16 + col];
        bad += ref != Dd[l * 4 + r];
    }
    printf("16x16x4 layout (A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4r col=l&15): %s\n", bad ? "MISMATCH" : "ok");
    // ---- 4x4x4 4-block layout: probe with one-hot operands.  a_lane = value at lane la, b_lane = value at lane lb
    std::vector<double> al(64), bl(64), d(64);
    double *dal, *dbl, *dd; CK(hipMalloc(&dal, 512)); CK(hipMalloc(&dbl, 512)); CK(hipMalloc(&dd, 512));
    printf("4x4x4_4b: for A one-hot at lane la and B all-ones: which D lanes are non-zero\n");
    for (int la : {0, 1, 4, 5, 16, 21, 63}) {
        for (int l = 0; l < 64; ++l) { al[l] = l == la ? 1.0 : 0.0; bl[l] = 100 + l; }
        CK(hipMemcpy(dal, al.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dbl, bl.data(), 512, hipMemcpyHostToDevice));
        layout4<<<1, 64>>>(dal, dbl, dd); CK(hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost));
        printf("  la=%2d:", la);
        for (int l = 0; l < 64; ++l) if (d[l] != 0) printf(" D[%d]=%g", l, d[l]);
        printf("\n");
    }
    return 0;
}
