// ba_matfree.hpp — matrix-free products with the bundle-adjustment Jacobian.
//
// What scipy does with the CSR Jacobian the reference hands it (optimisation_handling.py:88-98:
// x_scale='jac' column norms, J^T f gradient, lsmr mat-vecs) needs J only through products.  These
// kernels recompute each detection's 2 x P block from the slabs (FMAs are ~6x under the memory
// budget of the dense kernel) and apply it on the fly, so J is never written to HBM:
//     OP_JV    out[2i..2i+1] = J_i v                       (2N outputs, coalesced)
//     OP_JTU   out += J_i^T u_i                            (n_params accumulators)
//     OP_JTJV  out += J_i^T (J_i v)                        (normal-equation operator, one pass)
//     OP_DIAG  out += diag(J_i^T J_i)                      (column square norms: x_scale='jac', Jacobi)
//     OP_GRAD  out += J_i^T r_i,  cost += |r_i|^2          (gradient + cost at the linearisation point)
// Vectors live in the FULL parameter-string space (afb:777-820 layout); fixed parameters are handled
// by the caller (zeros in v, ignored entries of out).  Traffic: 28 B per detection in (+16 B for
// OP_JTU / out of OP_JV) — the n_params-sized vectors stay in L1/L2.
//
// Accumulation (three levels): a tile of 64 detections in the reference's cam -> image -> key order
// shares its camera and pose, so the 15 camera columns and 6 pose columns are first summed over the
// tile: every lane parks its contributions in a wave-private LDS panel (column-major, stride 65), then
// 63 lanes each add up a third of one column (22 ds_read_b64, no shuffles).  The partial sums (and
// the per-lane contributions of non-uniform tiles and of the 3 point columns) are added into a
// workgroup-private n_params accumulator in LDS (ds_add_f64); at the end each workgroup flushes its
// non-zero accumulators with one global f64 atomic each.  Going straight to global atomics costs
// 480 us at N = 1e6 (488 tiles per camera all hit the same 15 addresses, which serialise at the memory
// side); wave xor-shuffle sums + LDS accumulators 73-93 us; the LDS panel form is the current one.
// Parameter strings too large for LDS (> 150 KiB with the panels) fall back to direct global atomics.
// Atomic order makes the last bits of the sums run-to-run dependent (documented; the tests compare
// with a tolerance).
#pragma once
#include <hip/hip_runtime.h>

#include "ba_device.hpp"

namespace pcs {

constexpr int OP_JV = 0;
constexpr int OP_JTU = 1;
constexpr int OP_JTJV = 2;
constexpr int OP_DIAG = 3;
constexpr int OP_GRAD = 4;

constexpr int RED_COLS = 21;               // camera (15) + pose (6) columns summed per tile
constexpr int RED_STRIDE = 65;             // odd stride: the column sums read conflict-free
constexpr int RED_PANEL = RED_COLS * RED_STRIDE;  // doubles per wave

struct MatfreeArgs {
    DetTable tab;
    const void *cam_slab, *pose_slab, *points;
    const double *vin;   // OP_JV / OP_JTJV: n_params; OP_JTU: 2N
    double *vout;        // OP_JV: 2N; others: n_params (zeroed by the host before the launch)
    double *cost;        // OP_GRAD: 1 accumulator (zeroed by the host)
    int64_t n, n_tiles;
    int64_t extr_off, pose_off, point_off;
    int32_t tiles_per_wg;
    int32_t n_params;
};

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// Add `count` per-detection contributions g[first .. first+count) into out[base(key) + j].
template <int COUNT>
__device__ __forceinline__ void accumulate_group(const double *g, const int key, const int64_t base_off, const int stride,
                                                 const bool valid, const int lane, double *__restrict__ out) {
    const int k0 = __builtin_amdgcn_readfirstlane(key);
    const bool uniform = __all(!valid || key == k0);
    if (uniform) {
        double *dst = out + base_off + (int64_t)k0 * stride;
#pragma unroll
        for (int j = 0; j < COUNT; ++j) {
            const double s = wave_sum(valid ? g[j] : 0.0);
            if (lane == 0) unsafeAtomicAdd(dst + j, s);
        }
    } else if (valid) {
        double *dst = out + base_off + (int64_t)key * stride;
#pragma unroll
        for (int j = 0; j < COUNT; ++j) unsafeAtomicAdd(dst + j, g[j]);
    }
}

template <int CHAIN, int OP, bool LDS_ACC>
__global__ __launch_bounds__(256) void ba_matfree_kernel(const MatfreeArgs a) {
    using T = double;
    extern __shared__ __attribute__((aligned(16))) double lds_acc[];
    if constexpr (LDS_ACC && OP != OP_JV) {
        for (int j = threadIdx.x; j < a.n_params; j += 256) lds_acc[j] = 0.0;
        __syncthreads();
    }
    constexpr int P = chain_P(CHAIN);
    constexpr int P2 = 2 * P;
    using D2 = __attribute__((ext_vector_type(2))) double;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const T *cam_slab = static_cast<const T *>(a.cam_slab);
    const T *pose_slab = static_cast<const T *>(a.pose_slab);
    const T *points = static_cast<const T *>(a.points);
    const int64_t tile0 = (int64_t)blockIdx.x * a.tiles_per_wg;
    const int64_t tile1 = min(tile0 + (int64_t)a.tiles_per_wg, a.n_tiles);
    double cost_acc = 0.0;
    for (int64_t tile = tile0 + wave; tile < tile1; tile += 4) {
        const int64_t i = tile * 64 + lane;
        const bool valid = i < a.n;
        const int64_t ic = valid ? i : a.n - 1;
        int c, im, k;
        load_indices(a.tab, ic, c, im, k);
        const double2v m = load_uv(a.tab, ic);
        if constexpr (OP == OP_JV) {
            // J v WITHOUT J: forward mode (eval_detection_jvp, ba_device.hpp: a third of the arithmetic of the 2 x P block and its product).
            // A tile of the reference's table order lies inside a run of one (camera, image) pair or across ONE run boundary (four tiles in
            // ten on rig-32, runs of 156): the derivative is taken with the first lane's pair and, across a boundary, once more with the
            // last lane's — every lane keeps its own pair's — so that R, t, the intrinsics and v's camera / image entries always come
            // through scalar loads (48 values per pair: they fit the SGPRs, unlike the 87 of a full Jacobian) and no lane ever loads a
            // slab of its own (87 per-lane loads per straddling tile: what the kernel's time went to, whatever was done to the one-pair
            // tiles — DESIGN section 4).  The two 3 x 3 matrices the rotations' derivatives contract to with v (M_e, M_p) are formed once
            // per tile and pair: lane q < 36 holds one entry (three multiply-adds from its own loads of a dR block), read by v_readlane.
            const int c0 = __builtin_amdgcn_readfirstlane(c), im0 = __builtin_amdgcn_readfirstlane(im);
            const int c1 = __builtin_amdgcn_readlane(c, 63), im1 = __builtin_amdgcn_readlane(im, 63);
            const bool in_a = c == c0 && im == im0;
            if (__all(in_a || (c == c1 && im == im1))) {
                const bool two = c0 != c1 || im0 != im1;
                const T X0 = points[3 * k], X1 = points[3 * k + 1], X2 = points[3 * k + 2];
                const double *vin = a.vin;
                const int64_t cEa = a.extr_off + 6 * (int64_t)c0, cPa = a.pose_off + 6 * (int64_t)im0;
                const int64_t cEb = a.extr_off + 6 * (int64_t)c1, cPb = a.pose_off + 6 * (int64_t)im1;
                double mval = 0.0;
                {
                    const int l36 = min(lane, 35), half = l36 >= 18 ? 1 : 0, q18 = l36 - 18 * half;
                    const bool cam_part = q18 < 9;
                    const int q = cam_part ? q18 : q18 - 9;
                    const int cc = half ? c1 : c0, ii = half ? im1 : im0;
                    const T *dr = cam_part ? cam_slab + cc * CAM_STRIDE + CAM_DR : pose_slab + ii * POSE_STRIDE + POSE_DR;
                    const double *vr = cam_part ? vin + (half ? cEb : cEa) : vin + (half ? cPb : cPa);
                    if (CHAIN != CHAIN_FREE || cam_part) mval = dr[q] * vr[0] + dr[9 + q] * vr[1] + dr[18 + q] * vr[2];
                }
                double vX0 = 0.0, vX1 = 0.0, vX2 = 0.0;
                if constexpr (CHAIN != CHAIN_TEMPLATE) {
                    const int64_t cXl = a.point_off + 3 * (int64_t)k;
                    vX0 = vin[cXl]; vX1 = vin[cXl + 1]; vX2 = vin[cXl + 2];
                }
                struct LaneMat {   // M_e, M_p of one pair inside the lanes' values
                    double v;
                    int off;
                    __device__ __forceinline__ double operator[](const int j) const { return readlane_scalar(v, off + j); }
                };
                double du, dv;
                eval_detection_jvp<CHAIN>(ScalarSlab(cam_slab + c0 * CAM_STRIDE), ScalarSlab(pose_slab + im0 * POSE_STRIDE), X0, X1, X2, LaneMat{mval, 0},
                                          ScalarSlab(vin + 9 * (int64_t)c0), ScalarSlab(vin + cEa + 3), ScalarSlab(vin + (CHAIN != CHAIN_FREE ? cPa + 3 : 0)),
                                          vX0, vX1, vX2, du, dv);
                if (two) {
                    double du1, dv1;
                    eval_detection_jvp<CHAIN>(ScalarSlab(cam_slab + c1 * CAM_STRIDE), ScalarSlab(pose_slab + im1 * POSE_STRIDE), X0, X1, X2, LaneMat{mval, 18},
                                              ScalarSlab(vin + 9 * (int64_t)c1), ScalarSlab(vin + cEb + 3), ScalarSlab(vin + (CHAIN != CHAIN_FREE ? cPb + 3 : 0)),
                                              vX0, vX1, vX2, du1, dv1);
                    du = in_a ? du : du1;
                    dv = in_a ? dv : dv1;
                }
                if (valid) {
                    D2 q2;
                    q2.x = du;
                    q2.y = dv;
                    reinterpret_cast<D2 *>(a.vout)[i] = q2;
                }
                continue;
            }
        }
        T u, v;
        T J[P2];
        {
            const T X0 = points[3 * k], X1 = points[3 * k + 1], X2 = points[3 * k + 2];
            const int c0 = __builtin_amdgcn_readfirstlane(c), im0 = __builtin_amdgcn_readfirstlane(im);
            if (__all(c == c0 && im == im0))   // one camera and one image in the tile: slabs through scalar loads (ScalarSlab, ba_device.hpp)
                eval_detection<CHAIN, T, true>(ScalarSlab(cam_slab + c0 * CAM_STRIDE), ScalarSlab(pose_slab + im0 * POSE_STRIDE), X0, X1, X2, u, v, J);
            else
                eval_detection<CHAIN, T, true>(cam_slab + c * CAM_STRIDE, pose_slab + im * POSE_STRIDE, X0, X1, X2, u, v, J);
        }
        // global columns of this detection's P local parameters
        const int64_t cI = 9 * (int64_t)c, cE = a.extr_off + 6 * (int64_t)c;
        const int64_t cP = a.pose_off + 6 * (int64_t)im, cX = a.point_off + 3 * (int64_t)k;
        double w0 = 0.0, w1 = 0.0;
        if constexpr (OP == OP_JV || OP == OP_JTJV) {
            const double *vin = a.vin;
#pragma unroll
            for (int j = 0; j < 9; ++j) { const double x = vin[cI + j]; w0 += (double)J[j] * x; w1 += (double)J[P + j] * x; }
#pragma unroll
            for (int j = 0; j < 6; ++j) { const double x = vin[cE + j]; w0 += (double)J[9 + j] * x; w1 += (double)J[P + 9 + j] * x; }
            if constexpr (CHAIN != CHAIN_FREE) {
#pragma unroll
                for (int j = 0; j < 6; ++j) { const double x = vin[cP + j]; w0 += (double)J[15 + j] * x; w1 += (double)J[P + 15 + j] * x; }
            }
            if constexpr (CHAIN != CHAIN_TEMPLATE) {
                constexpr int o = (CHAIN == CHAIN_SELF) ? 21 : 15;
#pragma unroll
                for (int j = 0; j < 3; ++j) { const double x = vin[cX + j]; w0 += (double)J[o + j] * x; w1 += (double)J[P + o + j] * x; }
            }
        }
        if constexpr (OP == OP_JV) {
            if (valid) {
                D2 q;
                q.x = w0;
                q.y = w1;
                reinterpret_cast<D2 *>(a.vout)[i] = q;
            }
            continue;
        }
        if constexpr (OP == OP_JTU) {
            const D2 q = reinterpret_cast<const D2 *>(a.vin)[ic];
            w0 = q.x;
            w1 = q.y;
        }
        if constexpr (OP == OP_GRAD) {
            w0 = (double)(u - m.x);
            w1 = (double)(v - m.y);
            if (valid) cost_acc += w0 * w0 + w1 * w1;
        }
        double g[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            if constexpr (OP == OP_DIAG) g[j] = (double)J[j] * (double)J[j] + (double)J[P + j] * (double)J[P + j];
            else g[j] = (double)J[j] * w0 + (double)J[P + j] * w1;
        }
        if constexpr (LDS_ACC) {
            // tile sums through the wave-private LDS panel (see the header)
            double *red = lds_acc + ((a.n_params + 1) & ~1) + wave * RED_PANEL;
            // The tile's (camera, image) pairs: the first lane's (A) and the last lane's (B).  A tile of the reference's table order lies
            // inside one run (A = B) or across one boundary, the lanes of A first.  Every lane parks its contributions in the panel; the 63
            // summing lanes split what they add up by pair (rows below n_a: pair A) and send each part to its pair's entries.  (Until round
            // 5 a straddling tile — four in ten on rig-32 — added lane by lane: 21 x 64 LDS atomics on a handful of addresses;
            // SQ_LDS_BANK_CONFLICT was as large as SQ_ACTIVE_INST_LDS, profiles/r05/matfree_jtjv_sq_b.json.)  Other tiles: lane by lane.
            const int ca = __builtin_amdgcn_readfirstlane(c), ima = __builtin_amdgcn_readfirstlane(im);
            const int cb = __builtin_amdgcn_readlane(c, 63), imb = __builtin_amdgcn_readlane(im, 63);
            const bool in_a = c == ca && im == ima;
            const uint64_t mask_a = __ballot(in_a);
            const int n_a = __builtin_popcountll(mask_a);
            const bool two_pair = __all(in_a || (c == cb && im == imb)) && mask_a == (n_a == 64 ? ~0ull : (1ull << n_a) - 1ull);
            constexpr int NRED = (CHAIN != CHAIN_FREE) ? RED_COLS : 15;   // columns summed over the tile: camera (+ pose)
            if (two_pair) {
#pragma unroll
                for (int j = 0; j < NRED; ++j) red[j * RED_STRIDE + lane] = valid ? g[j] : 0.0;
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
                {
                    const int col = lane % RED_COLS, grp = lane / RED_COLS;  // lane 63: grp 3 -> idle
                    if (grp < 3 && col < NRED) {
                        // 22 independent reads first, then the adds: as a plain loop hipcc emits read - wait - add
                        // per row (22 exposed LDS latencies per tile).  Rows past 63 (third group) are clamped and masked.
                        const int r0 = grp * 22;
                        const double *colp = red + col * RED_STRIDE;
                        double vals[22];
#pragma unroll
                        for (int t = 0; t < 22; ++t) vals[t] = colp[min(r0 + t, 63)];
                        asm volatile("" ::: "memory");
                        double sum_a = 0.0, sum_b = 0.0;
                        if (n_a == 64) {   // one pair (wave-uniform): a plain sum
#pragma unroll
                            for (int t = 0; t < 22; ++t) sum_a += (r0 + t < 64) ? vals[t] : 0.0;
                        } else {
#pragma unroll
                            for (int t = 0; t < 22; ++t) {
                                const double x = (r0 + t < 64) ? vals[t] : 0.0;
                                sum_a += (r0 + t < n_a) ? x : 0.0;
                                sum_b += (r0 + t < n_a) ? 0.0 : x;
                            }
                        }
                        const int64_t dst_a = col < 9 ? 9 * (int64_t)ca + col : col < 15 ? a.extr_off + 6 * (int64_t)ca + (col - 9) : a.pose_off + 6 * (int64_t)ima + (col - 15);
                        const int64_t dst_b = col < 9 ? 9 * (int64_t)cb + col : col < 15 ? a.extr_off + 6 * (int64_t)cb + (col - 9) : a.pose_off + 6 * (int64_t)imb + (col - 15);
                        if (r0 < n_a) unsafeAtomicAdd(lds_acc + dst_a, sum_a);
                        if (r0 + 22 > n_a && n_a < 64) unsafeAtomicAdd(lds_acc + dst_b, sum_b);
                    }
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
            } else if (valid) {   // a third pair, or the pairs interleaved (not the reference's table order)
#pragma unroll
                for (int j = 0; j < 9; ++j) unsafeAtomicAdd(lds_acc + cI + j, g[j]);
#pragma unroll
                for (int j = 0; j < 6; ++j) unsafeAtomicAdd(lds_acc + cE + j, g[9 + j]);
                if constexpr (CHAIN != CHAIN_FREE) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) unsafeAtomicAdd(lds_acc + cP + j, g[15 + j]);
                }
            }
            if constexpr (CHAIN != CHAIN_TEMPLATE) {
                constexpr int o = (CHAIN == CHAIN_SELF) ? 21 : 15;
                if (valid) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) unsafeAtomicAdd(lds_acc + cX + j, g[o + j]);
                }
            }
        } else {
            accumulate_group<9>(g, c, 0, 9, valid, lane, a.vout);
            accumulate_group<6>(g + 9, c, a.extr_off, 6, valid, lane, a.vout);
            if constexpr (CHAIN != CHAIN_FREE) accumulate_group<6>(g + 15, im, a.pose_off, 6, valid, lane, a.vout);
            if constexpr (CHAIN != CHAIN_TEMPLATE) {
                constexpr int o = (CHAIN == CHAIN_SELF) ? 21 : 15;
                if (valid) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) unsafeAtomicAdd(a.vout + cX + j, g[o + j]);
                }
            }
        }
    }
    if constexpr (LDS_ACC && OP != OP_JV) {
        __syncthreads();
        for (int j = threadIdx.x; j < a.n_params; j += 256) {
            const double x = lds_acc[j];
            if (x != 0.0) unsafeAtomicAdd(a.vout + j, x);
        }
    }
    if constexpr (OP == OP_GRAD) {
        const double s = wave_sum(cost_acc);
        if (lane == 0) unsafeAtomicAdd(a.cost, s);
    }
}

}  // namespace pcs
