// mfma_valu_overlap_probe.hip — does an FP64 MFMA leave room for vector FP64 work on gfx950?
// The normal-equations kernel alternates an evaluation phase (v_fma_f64 ...) and a Gram phase (v_mfma_f64_16x16x4_f64)
// and its phase times ADD.  This probe separates the two possible reasons:
//   (a) one wave: a loop of 1 MFMA + K independent v_fma_f64 (or v_fma_f32 / v_add_u32) — if the matrix pipe were a separate
//       unit, K FMAs up to its 64-cycle shadow would be free;
//   (b) two waves on one SIMD, one issuing only MFMAs and the other only FMAs — same question across waves.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap_probe mfma_valu_overlap_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>

using d4 = __attribute__((ext_vector_type(4))) double;

// KIND 0: v_fma_f64, 1: v_fma_f32, 2: v_add_u32  (K per MFMA, 8 independent chains)
template <int KIND, int K, int NMFMA>
__global__ __launch_bounds__(1024) void mixed_kernel(double *out, int iters, long long *cycles) {   // 1024: accumulators stay in VGPRs
    const double a = (double)(threadIdx.x & 15) * 0.5, b = (double)(threadIdx.x >> 4) * 0.25;
    d4 acc0 = {}, acc1 = {};
    double s[8] = {};
    float f[8] = {};
    unsigned u[8] = {};
    const float af = (float)a, bf = (float)b;
    auto valu = [&]() {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if constexpr (KIND == 0) s[j & 7] = __builtin_fma(a, b, s[j & 7]);
            else if constexpr (KIND == 1) f[j & 7] = __builtin_fmaf(af, bf, f[j & 7]);
            else { u[j & 7] += threadIdx.x; asm volatile("" : "+v"(u[j & 7])); }
        }
    };
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {   // per iteration: 2 x (NMFMA MFMA + K vector instructions)
        if constexpr (NMFMA) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        valu();
        if constexpr (NMFMA) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
        valu();
    }
    const long long t1 = __builtin_readcyclecounter();
    double r = acc0.x + acc0.y + acc1.z + acc1.w;
    for (int j = 0; j < 8; ++j) r += s[j] + f[j] + u[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// two waves per SIMD (512 threads): waves 0-3 run `na` MFMAs per iteration, waves 4-7 run `nb` x 8 v_fma_f64 per iteration
__global__ __launch_bounds__(1024) void split_kernel(double *out, int iters, int do_mfma, int do_fma, long long *cycles) {
    const double a = (double)(threadIdx.x & 15) * 0.5, b = (double)(threadIdx.x >> 4) * 0.25;
    const int wave = threadIdx.x >> 6;
    d4 acc[2] = {};
    double s[8] = {};
    const long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (do_mfma)
            for (int i = 0; i < iters; ++i) {
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[1], 0, 0, 0);
            }
    } else {
        if (do_fma)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 32; ++j) s[j & 7] = __builtin_fma(a, b, s[j & 7]);
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    double r = acc[0].x + acc[1].y;
    for (int j = 0; j < 8; ++j) r += s[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cycles[wave] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND, int K, int NMFMA>
static int run_mixed(const char *what) {
    int dev = 0; hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
    const int blocks = p.multiProcessorCount, iters = 20000;
    double *out; long long *cyc; CK(hipMalloc(&out, sizeof(double) * blocks * 256)); CK(hipMalloc(&cyc, 64));
    mixed_kernel<KIND, K, NMFMA><<<blocks, 256>>>(out, 100, cyc);
    mixed_kernel<KIND, K, NMFMA><<<blocks, 256>>>(out, iters, cyc);
    CK(hipDeviceSynchronize());
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("one wave/SIMD: %d MFMA + %2d %-10s : %7.1f clk per (MFMA + vector group)\n", NMFMA, K, what, (double)c / iters / 2);
    CK(hipFree(out)); CK(hipFree(cyc));
    return 0;
}

static int run_split(int do_mfma, int do_fma) {
    int dev = 0; hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
    const int blocks = p.multiProcessorCount, iters = 20000;
    double *out; long long *cyc; CK(hipMalloc(&out, sizeof(double) * blocks * 512)); CK(hipMalloc(&cyc, 64));
    CK(hipMemset(cyc, 0, 64));
    split_kernel<<<blocks, 512>>>(out, 100, do_mfma, do_fma, cyc);
    split_kernel<<<blocks, 512>>>(out, iters, do_mfma, do_fma, cyc);
    CK(hipDeviceSynchronize());
    long long c[8]; CK(hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost));
    printf("two waves/SIMD: MFMA wave %s (2 per iteration), FMA wave %s (32 v_fma_f64 per iteration): MFMA wave %7.1f clk/iter, FMA wave %7.1f clk/iter\n",
           do_mfma ? "on " : "off", do_fma ? "on " : "off", (double)c[0] / iters, (double)c[4] / iters);
    CK(hipFree(out)); CK(hipFree(cyc));
    return 0;
}

int main() {
    if (run_mixed<0, 0, 1>("v_fma_f64")) return 1;
    if (run_mixed<0, 4, 1>("v_fma_f64")) return 1;
    if (run_mixed<0, 8, 1>("v_fma_f64")) return 1;
    if (run_mixed<0, 16, 1>("v_fma_f64")) return 1;
    if (run_mixed<0, 32, 1>("v_fma_f64")) return 1;
    if (run_mixed<0, 16, 0>("v_fma_f64")) return 1;
    if (run_mixed<0, 32, 0>("v_fma_f64")) return 1;
    if (run_mixed<1, 8, 1>("v_fma_f32")) return 1;
    if (run_mixed<1, 16, 1>("v_fma_f32")) return 1;
    if (run_mixed<1, 32, 1>("v_fma_f32")) return 1;
    if (run_mixed<1, 32, 0>("v_fma_f32")) return 1;
    if (run_mixed<2, 8, 1>("v_add_u32")) return 1;
    if (run_mixed<2, 16, 1>("v_add_u32")) return 1;
    if (run_mixed<2, 32, 1>("v_add_u32")) return 1;
    if (run_mixed<2, 32, 0>("v_add_u32")) return 1;
    if (run_split(1, 0)) return 1;
    if (run_split(0, 1)) return 1;
    if (run_split(1, 1)) return 1;
    return 0;
}
