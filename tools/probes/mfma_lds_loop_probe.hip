// mfma_lds_loop_probe.hip — what a wave sustains when every FP64 MFMA takes fresh operands from LDS (the inner loop of
// ba_normal_mfma_kernel): 3 ds_read_b64 + 2 v_mfma_f64_16x16x4_f64 per k-step, in several software-pipelining shapes.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds_loop_probe mfma_lds_loop_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;
constexpr int KS = 1040, NSLOT = 22, STEPS = 32;

template <int SHAPE>
__global__ __launch_bounds__(64, 2) void loop_kernel(double *out, int tiles, long long *cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x;
    for (int s = 0; s < NSLOT; ++s) *reinterpret_cast<double2 *>(lds + s * KS + lane * 16) = make_double2(lane + s, lane - s);
    __syncthreads();
    const int j = lane & 15;
    int off[3];
    off[0] = (j < 8 ? j : j < 14 ? 16 + (j - 8) : 0) * KS + (lane >> 4) * 8;
    off[1] = j * KS + (lane >> 4) * 8;
    off[2] = (j < 8 ? 8 + j : j < 14 ? 16 + (j - 8) : 0) * KS + (lane >> 4) * 8;
    d4 acc0 = {}, acc1 = {};
    auto rd = [&](int w, int s) { return *reinterpret_cast<const double *>(lds + off[w] + s * 32); };
    const long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < tiles; ++t) {
        if constexpr (SHAPE == 0) {          // naive: read, use
            for (int s = 0; s < STEPS; ++s) {
                const double a = rd(0, s), b = rd(1, s), c = rd(2, s);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, c, acc1, 0, 0, 0);
            }
        } else if constexpr (SHAPE == 1) {   // fully unrolled, distance-1 prefetch, compiler waitcnts
            double x[2][3];
            for (int w = 0; w < 3; ++w) x[0][w] = rd(w, 0);
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                if (s + 1 < STEPS) for (int w = 0; w < 3; ++w) x[(s + 1) & 1][w] = rd(w, s + 1);
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s & 1][0], x[s & 1][1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s & 1][2], x[s & 1][2], acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (SHAPE == 2) {   // fully unrolled, distance-2 prefetch
            double x[3][3];
            for (int w = 0; w < 3; ++w) { x[0][w] = rd(w, 0); x[1][w] = rd(w, 1); }
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                if (s + 2 < STEPS) for (int w = 0; w < 3; ++w) x[(s + 2) % 3][w] = rd(w, s + 2);
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s % 3][0], x[s % 3][1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s % 3][2], x[s % 3][2], acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (SHAPE == 3) {   // all 96 operands first, then 64 MFMAs (register-heavy upper bound)
            double x[STEPS][3];
#pragma unroll
            for (int s = 0; s < STEPS; ++s) for (int w = 0; w < 3; ++w) x[s][w] = rd(w, s);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s][0], x[s][1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s][2], x[s][2], acc1, 0, 0, 0);
            }
        } else {                             // no LDS at all: register operands (the pipe's own rate)
            const double a = lane, b = lane * 0.5;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, acc1, 0, 0, 0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double r = 0;
    for (int k = 0; k < 4; ++k) r += acc0[k] + acc1[k];
    out[blockIdx.x * 64 + lane] = r;
    if (lane == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int SHAPE>
static int run(const char *name, int wgs_per_cu) {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * wgs_per_cu, tiles = 200;
    double *out; long long *cyc; CK(hipMalloc(&out, 8 * 64 * blocks)); CK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    loop_kernel<SHAPE><<<blocks, 64, NSLOT * KS>>>(out, 4, cyc);
    CK(hipEventRecord(e0)); loop_kernel<SHAPE><<<blocks, 64, NSLOT * KS>>>(out, tiles, cyc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-44s waves/CU %d: %7.1f cycles per k-step per wave (2 MFMA = 128 pipe cycles), kernel %.3f ms\n", name, wgs_per_cu, (double)c / tiles / STEPS, ms);
    CK(hipFree(out)); CK(hipFree(cyc));
    return 0;
}
int main() {
    for (int w : {1, 4, 7}) {
        if (run<4>("register operands (no LDS)", w)) return 1;
        if (run<0>("naive read-then-use", w)) return 1;
        if (run<1>("unrolled, prefetch distance 1", w)) return 1;
        if (run<2>("unrolled, prefetch distance 2", w)) return 1;
        if (run<3>("all reads first", w)) return 1;
    }
    return 0;
}
