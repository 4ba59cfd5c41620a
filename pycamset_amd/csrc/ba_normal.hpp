// ba_normal.hpp — block-reduced normal equations ("block-reduced J^T J / J^T r, then
// all-reduce of the small result instead of all-gather of J").
//
// One pass over the detections builds, without ever writing J:
//     H    = J^T J   upper triangle of the n_params x n_params matrix (row-major, full parameter-string space)
//     g    = J^T r   n_params
//     cost = r^T r
// This is what a Levenberg-Marquardt step needs from the Jacobian the reference hands to scipy
// (optimisation_handling.py:88-98); with it one LM iteration costs one pass + a small dense solve
// instead of ~150 matrix-free J^T(Jv) products (profiles/r01/lm_rig32_config3_device.log).
//
// Structure.  A detection's 2 x P block touches the 15 columns of its camera, the 6 of its image
// and (self / free chains) the 3 of its key.  In the reference's table order (cam -> image -> key) a run
// of detections shares camera and image, so the "shared" (camera + pose) part of H is a sum of
// X^T X over the run, X = [J_shared | r] being the run's (2 x detections) x NA augmented rows
// (NA = 22, or 16 for the free chain): a small symmetric rank-k update whose NA (NA + 1) / 2 entries are
// the run's contribution to the camera block, the pose block, the camera-pose block, g and the cost.
//   * Every lane evaluates one detection (eval_detection, as the fused kernel does) and parks its two
//     augmented rows in a wave-private LDS image (32 detections per pass, row stride padded to an odd
//     number of 16-byte slots: conflict-free ds_write_b128).
//   * The upper triangle is cut row by row into chunks (p; q0 .. q0+4), one per lane (60 of 64 lanes for
//     NA = 22).  A lane accumulates sum_l X[l][p] X[l][q] over the member lanes: per l one
//     ds_read_b128 for the (u, v) pair of column p and one per q, at immediate offsets — all lanes read
//     the same row l.  (First version: 4 scattered entries per lane, 8 ds_read2_b64 per l: 171 us on
//     rig-32 — LDS-bandwidth-bound at 128 B/clk; this layout needs 6 reads of 16 B per 5 entries.)
//   * Accumulators live in registers across tiles for as long as (cam, image) does not change; on a
//     change (and at the end) they are added to H / g / cost with one global f64 atomic each.  A tile
//     that mixes several (cam, image) pairs is processed pair by pair (wave-uniform member masks), so any
//     table order gives the same result; shuffled tables simply flush more often.
//   * The 3 point columns (self / free chains) differ per lane, so their rows of H (point-point,
//     shared-point) and of g are added per detection with global atomics.
// Atomic order makes the last bits run-to-run dependent (documented; tests compare with a tolerance).
#pragma once
#include <hip/hip_runtime.h>

#include "ba_device.hpp"
#include "ba_matfree.hpp"  // wave_sum

namespace pcs {

struct NormalArgs {
    const int32_t *cam, *img, *key;
    const void *uv;
    const void *cam_slab, *pose_slab, *points;
    double *H;      // n_params x n_params, zeroed by the host; upper triangle written
    double *g;      // n_params, zeroed by the host
    double *cost;   // 1, zeroed by the host
    int64_t n, n_tiles;
    int64_t extr_off, pose_off, point_off;
    int64_t n_params;
    int32_t tiles_per_wave;
    int32_t debug;  // profiling switches: 1 skip the dot loops, 2 skip the flush atomics
};

constexpr int normal_shared_cols(int chain) { return chain == CHAIN_FREE ? 15 : 21; }
// LDS row of one detection: NA 16-byte slots (J[0][p], J[1][p]), slot NS = (r_u, r_v), padded to an odd
// number of slots so that the 8-lane groups of ds_write_b128 land on different banks
constexpr int normal_row(int chain) { return 2 * (normal_shared_cols(chain) + 1) + 2; }  // doubles
constexpr int NORMAL_HALF = 32;
constexpr int NORMAL_TAIL = 16;  // doubles after a wave's image: the dummy columns of short chunks read into it

// Entry ownership: the upper triangle (p <= q < NA) is cut, row by row, into chunks of up to CH
// consecutive q; one chunk per lane.  CH is the smallest chunk length that fits 64 lanes.
constexpr int normal_chunks(int na, int ch) {
    int n = 0;
    for (int p = 0; p < na; ++p) n += (na - p + ch - 1) / ch;
    return n;
}
constexpr int normal_chunk_len(int na) {
    int ch = 1;
    while (normal_chunks(na, ch) > 64) ++ch;
    return ch;
}

// global column of shared local column p for (cam c, image im)
template <int CHAIN>
__device__ __forceinline__ int64_t shared_col(const NormalArgs &a, int p, int c, int im) {
    if (p < 9) return 9 * (int64_t)c + p;
    if (CHAIN == CHAIN_FREE || p < 15) return a.extr_off + 6 * (int64_t)c + (p - 9);
    return a.pose_off + 6 * (int64_t)im + (p - 15);
}

template <int CHAIN, typename T>
__global__ __launch_bounds__(256) void ba_normal_kernel(const NormalArgs a) {
    constexpr int P = chain_P(CHAIN);
    constexpr int P2 = 2 * P;
    constexpr int NS = normal_shared_cols(CHAIN);
    constexpr int NA = NS + 1;                 // shared columns + the residual column
    constexpr int CH = normal_chunk_len(NA);   // 5 (NA = 22: 60 chunks), 3 (NA = 16: 47 chunks)
    constexpr int ROW = normal_row(CHAIN);
    static_assert((ROW / 2) % 2 == 1, "row must be an odd number of 16-byte slots");
    static_assert(2 * CH <= NORMAL_TAIL, "dummy columns must stay inside the tail pad");
    using V2 = __attribute__((ext_vector_type(2))) T;
    using D2 = __attribute__((ext_vector_type(2))) double;

    extern __shared__ __attribute__((aligned(16))) double lds_rows[];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    double *X = lds_rows + wave * (NORMAL_HALF * ROW + NORMAL_TAIL);

    const T *cam_slab = static_cast<const T *>(a.cam_slab);
    const T *pose_slab = static_cast<const T *>(a.pose_slab);
    const T *points = static_cast<const T *>(a.points);
    const V2 *uv = static_cast<const V2 *>(a.uv);

    // this lane's chunk: row ep, columns eq0 .. eq0 + elen - 1
    int ep = 0, eq0 = 0, elen = 0;
    {
        int t = lane;
        for (int p = 0; p < NA; ++p) {
            const int nch = (NA - p + CH - 1) / CH;
            if (t < nch) {
                ep = p;
                eq0 = p + t * CH;
                elen = min(CH, NA - eq0);
                break;
            }
            t -= nch;
        }
    }
    const D2 *xa = reinterpret_cast<const D2 *>(X) + ep;
    const D2 *xb = reinterpret_cast<const D2 *>(X) + eq0;
    double acc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[j] = 0.0;
    int c_run = -1, i_run = -1;  // wave-uniform: the (cam, image) the accumulators belong to
    double cost_acc = 0.0;

    auto flush = [&]() {
        if (c_run < 0) return;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const double s = acc[j];
            acc[j] = 0.0;
            if (j >= elen || s == 0.0 || (a.debug & 2)) continue;
            const int q = eq0 + j;
            if (q == NS) {
                if (ep == NS) cost_acc += s;
                else unsafeAtomicAdd(a.g + shared_col<CHAIN>(a, ep, c_run, i_run), s);
            } else {
                const int64_t gp = shared_col<CHAIN>(a, ep, c_run, i_run), gq = shared_col<CHAIN>(a, q, c_run, i_run);
                unsafeAtomicAdd(a.H + gp * a.n_params + gq, s);
            }
        }
    };

    const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave;
    const int64_t tile0 = wave_id * a.tiles_per_wave;
    const int64_t tile1 = min(tile0 + (int64_t)a.tiles_per_wave, a.n_tiles);
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        const int64_t i = tile * 64 + lane;
        const bool valid = i < a.n;
        const int64_t ic = valid ? i : a.n - 1;
        const int c = a.cam[ic], im = (CHAIN != CHAIN_FREE) ? a.img[ic] : 0, k = a.key[ic];
        const V2 m = uv[ic];
        T u, v;
        T J[P2];
        eval_detection<CHAIN, T, true>(cam_slab + c * CAM_STRIDE, pose_slab + im * POSE_STRIDE, points[3 * k], points[3 * k + 1],
                                       points[3 * k + 2], u, v, J);
        const double r0 = (double)(u - m.x), r1 = (double)(v - m.y);

        if constexpr (CHAIN != CHAIN_TEMPLATE) {
            // point columns: per-detection rows of H and g (upper triangle: shared columns come first)
            if (valid) {
                const int64_t gX = a.point_off + 3 * (int64_t)k;
                double jp0[3], jp1[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) { jp0[t] = (double)J[NS + t]; jp1[t] = (double)J[P + NS + t]; }
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    unsafeAtomicAdd(a.g + gX + t, jp0[t] * r0 + jp1[t] * r1);
#pragma unroll
                    for (int s = t; s < 3; ++s) unsafeAtomicAdd(a.H + (gX + t) * a.n_params + gX + s, jp0[t] * jp0[s] + jp1[t] * jp1[s]);
                }
#pragma unroll
                for (int p = 0; p < NS; ++p) {
                    double *row = a.H + shared_col<CHAIN>(a, p, c, im) * a.n_params + gX;
                    const double j0 = (double)J[p], j1 = (double)J[P + p];
#pragma unroll
                    for (int t = 0; t < 3; ++t) unsafeAtomicAdd(row + t, j0 * jp0[t] + j1 * jp1[t]);
                }
            }
        }

        const uint64_t valid_mask = __ballot(valid);
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            if ((lane >> 5) == h) {
                D2 *dst = reinterpret_cast<D2 *>(X + (lane & 31) * ROW);
#pragma unroll
                for (int p = 0; p < NS; ++p) {
                    D2 w;
                    w.x = (double)J[p];
                    w.y = (double)J[P + p];
                    dst[p] = w;
                }
                D2 w;
                w.x = r0;
                w.y = r1;
                dst[NS] = w;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint32_t rem = (uint32_t)(valid_mask >> (32 * h));
            while (rem) {  // one (cam, image) pair of this half at a time; all conditions are wave-uniform
                const int leader = __builtin_ctz(rem) + 32 * h;
                const int c0 = __builtin_amdgcn_readlane(c, leader), i0 = __builtin_amdgcn_readlane(im, leader);
                const uint32_t member = (uint32_t)(__ballot(valid && c == c0 && im == i0) >> (32 * h)) & rem;
                if (c0 != c_run || i0 != i_run) {
                    flush();
                    c_run = c0;
                    i_run = i0;
                }
                if (a.debug & 1) {
                    acc[0] += 1.0;
                } else if (member == 0xffffffffu) {  // the whole half is one run: straight-line code, reads pipelined
#pragma unroll 8
                    for (int l = 0; l < NORMAL_HALF; ++l) {
                        const D2 p2 = xa[l * (ROW / 2)];
#pragma unroll
                        for (int j = 0; j < CH; ++j) {
                            const D2 q2 = xb[l * (ROW / 2) + j];
                            acc[j] += p2.x * q2.x + p2.y * q2.y;
                        }
                    }
                } else {
#pragma unroll 4
                    for (int l = 0; l < NORMAL_HALF; ++l) {
                        if (member & (1u << l)) {
                            const D2 p2 = xa[l * (ROW / 2)];
#pragma unroll
                            for (int j = 0; j < CH; ++j) {
                                const D2 q2 = xb[l * (ROW / 2) + j];
                                acc[j] += p2.x * q2.x + p2.y * q2.y;
                            }
                        }
                    }
                }
                rem &= ~member;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    flush();
    const double cs = wave_sum(cost_acc);
    if (lane == 0 && cs != 0.0) unsafeAtomicAdd(a.cost, cs);
}

}  // namespace pcs
