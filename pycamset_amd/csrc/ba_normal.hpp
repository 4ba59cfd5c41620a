// ba_normal.hpp — block-reduced normal equations ("block-reduced J^T J / J^T r, then
// all-reduce of the small result instead of all-gather of J").
//
// Built on the GPU without ever writing J:
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
//     augmented rows in a wave-private LDS image (64 rows of 23 16-byte slots — an odd stride, so the
//     ds_write_b128 are conflict-free; 23.6 KB per wave, 2 waves per workgroup, 3 workgroups per CU).
//   * The upper triangle is cut row by row into chunks (p; q0 .. q0+4), one per lane (60 of 64 lanes for
//     NA = 22).  A lane accumulates sum_l X[l][p] X[l][q] over the member lanes: per l one
//     ds_read_b128 for the (u, v) pair of column p and one per q, at immediate offsets — all lanes read
//     the same row l.  (First version: 4 scattered entries per lane, 8 ds_read2_b64 per l: 171 us on
//     rig-32 — LDS-bandwidth-bound at 128 B/clk; this layout needs 6 reads of 16 B per 5 entries.)
//   * Accumulators live in registers across tiles for as long as (cam, image) does not change; on a
//     change (and at the end) they are added to H / g / cost with one global f64 atomic each.  A tile
//     that mixes several (cam, image) pairs is processed pair by pair (wave-uniform member masks), so any
//     table order gives the same result.  For a scattered table (every detection its own run: 18 ms on rig-32)
//     the host hands over a (cam, image)-sorted visiting order instead — the sums do not depend on it.
//   * The 3 point columns (self / free chains) differ per lane; their rows of H (point-point, shared-point) and
//     of g come from ba_normal_point_kernel (end of this file: separate passes over key-sorted visiting
//     orders), or — fallback — from per-detection global atomics in this kernel.
//   * The read loop is software-pipelined by hand (NORMAL_DEPTH rows of operands in flight) and the lane ->
//     chunk assignment comes from a table that keeps LDS slots 16 apart out of the same read group.
// One pass / one kernel for the template chain; up to three kernels for the self chain.
// Atomic order makes the last bits run-to-run dependent (documented; tests compare with a tolerance).
#pragma once
#include <hip/hip_runtime.h>

#include "ba_device.hpp"
#include "ba_matfree.hpp"  // wave_sum

namespace pcs {

struct NormalArgs {
    const int32_t *cam, *img, *key;
    const void *uv;
    const int32_t *order;  // optional: visit the detections in this order (a (cam, image)-sorted permutation of a
                           // scattered table; H, g and the cost do not depend on the row order), or NULL
    const void *cam_slab, *pose_slab, *points;
    double *H;      // n_params x n_params, zeroed by the host; upper triangle written
    double *g;      // n_params, zeroed by the host
    double *cost;   // 1, zeroed by the host
    int64_t n, n_tiles;
    int64_t extr_off, pose_off, point_off;
    int64_t n_params;
    int32_t tiles_per_wave;
    int32_t skip_points;  // 1: the point columns are left to ba_normal_point_kernel
    int32_t debug;  // profiling switches: 1 skip the dot loops, 2 skip the flush atomics, 4 / 8 skip the point-block / shared-point atomics
};

constexpr int normal_shared_cols(int chain) { return chain == CHAIN_FREE ? 15 : 21; }
// LDS row of one detection: NA 16-byte slots (J[0][p], J[1][p]), slot NS = (r_u, r_v), padded to an odd
// number of slots so that the 8-lane groups of ds_write_b128 land on different banks
constexpr int normal_row(int chain) { return 2 * (normal_shared_cols(chain) + 1) + 2; }  // doubles
constexpr int NORMAL_ROWS = 64;  // detections per LDS image (the whole wave tile)
constexpr int NORMAL_WAVES = 2;  // waves per workgroup: 2 x 23.6 KB of LDS -> 3 workgroups per CU
constexpr int NORMAL_DEPTH = 4;  // rows of operands in flight in the dot loop
// doubles after a wave's image: the dummy columns of short chunks and the last DEPTH - 1 prefetches read into it
constexpr int normal_tail(int chain) { return (NORMAL_DEPTH - 1) * normal_row(chain) + 16; }

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

// Lane -> chunk tables from tools/normal_lane_table.py (255 = idle lane).  ds_read_b128 serves a wave in four
// 16-lane groups over 16 slots of 16 B, so two lanes of a group collide when their slots differ by exactly 16;
// handing the chunks out in order leaves 4 such pairs for NA = 22 (SQ_LDS_BANK_CONFLICT = 34 % of the LDS
// cycles); these assignments have none.
// NA = 22, CH = 5: 60 chunks, 0 slot pairs 16 apart left (sequential order: 4)
__device__ constexpr unsigned char NORMAL_P_22[64] = {1, 13, 13, 14, 8, 255, 9, 4, 1, 1, 11, 4, 5, 255, 9, 2, 21, 12, 255, 0, 0, 4, 255, 19, 7, 2, 0, 5, 2, 14, 0, 0, 2, 17, 3, 6, 3, 10, 7, 1, 1, 16, 15, 6, 11, 12, 8, 5, 3, 15, 8, 10, 6, 16, 7, 9, 5, 10, 20, 11, 4, 18, 3, 6};
__device__ constexpr unsigned char NORMAL_Q_22[64] = {11, 18, 13, 19, 8, 255, 14, 4, 21, 1, 11, 14, 20, 255, 9, 7, 21, 12, 255, 0, 20, 19, 255, 19, 7, 17, 5, 5, 2, 14, 15, 10, 12, 17, 8, 11, 18, 20, 17, 6, 16, 21, 20, 16, 21, 17, 18, 10, 13, 15, 13, 15, 21, 16, 12, 19, 15, 10, 20, 16, 9, 18, 3, 6};
// NA = 16, CH = 3: 51 chunks, 0 slot pairs 16 apart left (sequential order: 0)
__device__ constexpr unsigned char NORMAL_P_16[64] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 8, 8, 8, 9, 9, 9, 10, 10, 11, 11, 12, 12, 13, 14, 15, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255};
__device__ constexpr unsigned char NORMAL_Q_16[64] = {0, 3, 6, 9, 12, 15, 1, 4, 7, 10, 13, 2, 5, 8, 11, 14, 3, 6, 9, 12, 15, 4, 7, 10, 13, 5, 8, 11, 14, 6, 9, 12, 15, 7, 10, 13, 8, 11, 14, 9, 12, 15, 10, 13, 11, 14, 12, 15, 13, 14, 15, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255};

// global column of shared local column p for (cam c, image im)
template <int CHAIN>
__device__ __forceinline__ int64_t shared_col(const NormalArgs &a, int p, int c, int im) {
    if (p < 9) return 9 * (int64_t)c + p;
    if (CHAIN == CHAIN_FREE || p < 15) return a.extr_off + 6 * (int64_t)c + (p - 9);
    return a.pose_off + 6 * (int64_t)im + (p - 15);
}

template <int CHAIN, typename T>
__global__ __launch_bounds__(64 * NORMAL_WAVES) void ba_normal_kernel(const NormalArgs a) {
    constexpr int P = chain_P(CHAIN);
    constexpr int P2 = 2 * P;
    constexpr int NS = normal_shared_cols(CHAIN);
    constexpr int NA = NS + 1;                 // shared columns + the residual column
    constexpr int CH = normal_chunk_len(NA);   // 5 (NA = 22: 60 chunks), 3 (NA = 16: 51 chunks)
    constexpr int ROW = normal_row(CHAIN);
    static_assert((ROW / 2) % 2 == 1, "row must be an odd number of 16-byte slots");
    static_assert(2 * CH <= 16 && NORMAL_ROWS % NORMAL_DEPTH == 0, "dummy columns must stay inside the tail pad");
    using V2 = __attribute__((ext_vector_type(2))) T;
    using D2 = __attribute__((ext_vector_type(2))) double;

    extern __shared__ __attribute__((aligned(16))) double lds_rows[];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    double *X = lds_rows + wave * (NORMAL_ROWS * ROW + normal_tail(CHAIN));

    const T *cam_slab = static_cast<const T *>(a.cam_slab);
    const T *pose_slab = static_cast<const T *>(a.pose_slab);
    const T *points = static_cast<const T *>(a.points);
    const V2 *uv = static_cast<const V2 *>(a.uv);

    // this lane's chunk: row ep, columns eq0 .. eq0 + elen - 1
    static_assert(NA == 22 || NA == 16, "lane tables exist for NA = 22 and 16");
    const unsigned char tp = NA == 22 ? NORMAL_P_22[lane] : NORMAL_P_16[lane];
    const unsigned char tq = NA == 22 ? NORMAL_Q_22[lane] : NORMAL_Q_16[lane];
    const int ep = tp == 255 ? 0 : tp, eq0 = tp == 255 ? 0 : tq;
    const int elen = tp == 255 ? 0 : min(CH, NA - eq0);
    const D2 *xa = reinterpret_cast<const D2 *>(X) + ep;
    const D2 *xb = reinterpret_cast<const D2 *>(X) + eq0;
    double acc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[j] = 0.0;
    int c_run = -1, i_run = -1;  // wave-uniform: the (cam, image) the accumulators belong to
    double cost_acc = 0.0;

    // Entries that involve no pose column (camera block, camera part of g, cost) belong to the camera alone:
    // they stay in registers across image changes and are flushed when the camera changes.  The camera
    // block of H otherwise receives one atomic per entry per (cam, image) run — ~260 serialised adds on
    // each of its addresses; this way it is one per wave that touches the camera.
    auto flush = [&](const bool cam_changed) {
        if (c_run < 0) return;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int q = eq0 + j;
            const bool pose_entry = CHAIN != CHAIN_FREE && (ep >= 15 || (q >= 15 && q < NS));
            if (!cam_changed && !pose_entry) continue;
            const double s = acc[j];
            acc[j] = 0.0;
            if (j >= elen || s == 0.0 || (a.debug & 2)) continue;
            if (q == NS) {
                if (ep == NS) cost_acc += s;
                else unsafeAtomicAdd(a.g + shared_col<CHAIN>(a, ep, c_run, i_run), s);
            } else {
                const int64_t gp = shared_col<CHAIN>(a, ep, c_run, i_run), gq = shared_col<CHAIN>(a, q, c_run, i_run);
                unsafeAtomicAdd(a.H + gp * a.n_params + gq, s);
            }
        }
    };

    const int64_t wave_id = (int64_t)blockIdx.x * NORMAL_WAVES + wave;
    const int64_t tile0 = wave_id * a.tiles_per_wave;
    const int64_t tile1 = min(tile0 + (int64_t)a.tiles_per_wave, a.n_tiles);
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        const int64_t i = tile * 64 + lane;
        const bool valid = i < a.n;
        const int64_t is = valid ? i : a.n - 1;
        const int64_t ic = a.order ? a.order[is] : is;
        const int c = a.cam[ic], im = (CHAIN != CHAIN_FREE) ? a.img[ic] : 0, k = a.key[ic];
        const V2 m = uv[ic];
        T u, v;
        T J[P2];
        eval_detection<CHAIN, T, true>(cam_slab + c * CAM_STRIDE, pose_slab + im * POSE_STRIDE, points[3 * k], points[3 * k + 1],
                                       points[3 * k + 2], u, v, J);
        const double r0 = (double)(u - m.x), r1 = (double)(v - m.y);

        if constexpr (CHAIN != CHAIN_TEMPLATE) {
            // point columns: per-detection rows of H and g (upper triangle: shared columns come first).  Fallback
            // only — 54-72 global atomics per detection (2.1-2.7 ms at N = 1e6); the engine normally runs
            // ba_normal_point_kernel over key-sorted visiting orders instead.
            if (valid && !a.skip_points) {
                const int64_t gX = a.point_off + 3 * (int64_t)k;
                double jp0[3], jp1[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) { jp0[t] = (double)J[NS + t]; jp1[t] = (double)J[P + NS + t]; }
                if (!(a.debug & 4)) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        unsafeAtomicAdd(a.g + gX + t, jp0[t] * r0 + jp1[t] * r1);
#pragma unroll
                        for (int s = t; s < 3; ++s) unsafeAtomicAdd(a.H + (gX + t) * a.n_params + gX + s, jp0[t] * jp0[s] + jp1[t] * jp1[s]);
                    }
                }
                if (!(a.debug & 8))
#pragma unroll
                for (int p = 0; p < NS; ++p) {
                    double *row = a.H + shared_col<CHAIN>(a, p, c, im) * a.n_params + gX;
                    const double j0 = (double)J[p], j1 = (double)J[P + p];
#pragma unroll
                    for (int t = 0; t < 3; ++t) unsafeAtomicAdd(row + t, j0 * jp0[t] + j1 * jp1[t]);
                }
            }
        }

        // the tile's 64 augmented rows -> LDS; J is dead after this, so the dot loops below have the
        // registers to keep many ds_read_b128 in flight (with a 32-row image and J alive across two passes
        // hipcc issued one read at a time: 470 cycles per row instead of ~100)
        {
            D2 *dst = reinterpret_cast<D2 *>(X + lane * ROW);
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
        uint64_t rem = __ballot(valid);
        while (rem) {  // one (cam, image) pair of the tile at a time; all conditions are wave-uniform
            const int leader = __builtin_ctzll(rem);
            const int c0 = __builtin_amdgcn_readlane(c, leader), i0 = __builtin_amdgcn_readlane(im, leader);
            const uint64_t member = __ballot(valid && c == c0 && im == i0) & rem;
            if (c0 != c_run || i0 != i_run) {
                flush(c0 != c_run);
                c_run = c0;
                i_run = i0;
            }
            if (a.debug & 1) {
                acc[0] += 1.0;
            } else if (member == ~0ull) {
                // The whole tile is one run (the common case).  Software pipeline: the operands of row
                // l + DEPTH - 1 are requested before row l is consumed; the compiler barrier keeps hipcc from
                // sinking the ds_read_b128 back down to their uses (which it does otherwise: one read, one
                // s_waitcnt 0, three VALU ops — the LDS latency fully exposed).  Rows past the image fall into
                // the tail pad and are never used.
                constexpr int DEPTH = NORMAL_DEPTH;
                D2 pr[DEPTH], qr[DEPTH][CH];
#pragma unroll
                for (int s = 0; s < DEPTH - 1; ++s) {
                    pr[s] = xa[s * (ROW / 2)];
#pragma unroll
                    for (int j = 0; j < CH; ++j) qr[s][j] = xb[s * (ROW / 2) + j];
                }
#pragma unroll 1
                for (int l0 = 0; l0 < NORMAL_ROWS; l0 += DEPTH) {
                    const D2 *xa0 = xa + l0 * (ROW / 2), *xb0 = xb + l0 * (ROW / 2);
#pragma unroll
                    for (int s = 0; s < DEPTH; ++s) {
                        constexpr int AHEAD = DEPTH - 1;
                        const int slot = (s + AHEAD) % DEPTH;
                        pr[slot] = xa0[(s + AHEAD) * (ROW / 2)];
#pragma unroll
                        for (int j = 0; j < CH; ++j) qr[slot][j] = xb0[(s + AHEAD) * (ROW / 2) + j];
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int j = 0; j < CH; ++j) acc[j] += pr[s].x * qr[s][j].x + pr[s].y * qr[s][j].y;
                    }
                }
            } else {
#pragma unroll 4
                for (int l = 0; l < NORMAL_ROWS; ++l) {
                    if (member & (1ull << l)) {
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
    flush(true);
    const double cs = wave_sum(cost_acc);
    if (lane == 0 && cs != 0.0) unsafeAtomicAdd(a.cost, cs);
}

// ---------------------------------------------------------------------------------------------------------------
// Point columns of the self / free chains, pass by pass over key-sorted visiting orders.
//
// A detection's point block couples its key k to its camera (15 columns) and — self chain — to its image (6
// columns).  Summed per detection that is 54-72 global f64 atomics each.  Visiting the detections sorted by
// (cam, key) makes the contributions to one camera-point block E[c,k] (and to D[k], g[k]) a contiguous run of
// lanes (one per image that sees the key, ~64 on rig-32); sorted by (image, key) the same holds for the pose-point
// block F[i,k] (~10 cameras).  Each pass re-evaluates the detection (38 us of arithmetic at N = 1e6), forms its
// products, sums them over the run with a segmented shuffle reduction (6 steps, lanes only add a neighbour that
// carries the same key — runs are contiguous, so that is exact) and the first lane of every run adds the sums
// to H / g with one atomic per entry.
//   WHICH 0: order by (cam, key):    E[c,k] 15 x 3, D[k] 3 x 3 upper, g[k]                      (54 sums)
//   WHICH 1: order by (image, key):  F[i,k] 6 x 3   (self chain only)                          (18 sums)
template <int CHAIN, typename T, int WHICH>
__global__ __launch_bounds__(256) void ba_normal_point_kernel(const NormalArgs a) {
    static_assert(CHAIN != CHAIN_TEMPLATE && (WHICH == 0 || CHAIN == CHAIN_SELF), "no such pass");
    constexpr int P = chain_P(CHAIN);
    constexpr int P2 = 2 * P;
    constexpr int NS = normal_shared_cols(CHAIN);
    constexpr int NV = WHICH == 0 ? 45 + 6 + 3 : 18;
    using V2 = __attribute__((ext_vector_type(2))) T;
    const int lane = threadIdx.x & 63;
    const T *cam_slab = static_cast<const T *>(a.cam_slab);
    const T *pose_slab = static_cast<const T *>(a.pose_slab);
    const T *points = static_cast<const T *>(a.points);
    const V2 *uv = static_cast<const V2 *>(a.uv);
    const int64_t n_waves = (int64_t)gridDim.x * 4, wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t tile = wave_id; tile < a.n_tiles; tile += n_waves) {
        const int64_t i = tile * 64 + lane;
        const bool valid = i < a.n;
        const int64_t ic = a.order[valid ? i : a.n - 1];
        const int c = a.cam[ic], im = (CHAIN != CHAIN_FREE) ? a.img[ic] : 0, k = a.key[ic];
        const V2 m = uv[ic];
        T u, v;
        T J[P2];
        eval_detection<CHAIN, T, true>(cam_slab + c * CAM_STRIDE, pose_slab + im * POSE_STRIDE, points[3 * k], points[3 * k + 1],
                                       points[3 * k + 2], u, v, J);
        double jp0[3], jp1[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) { jp0[t] = (double)J[NS + t]; jp1[t] = (double)J[P + NS + t]; }
        double s[NV];
        if constexpr (WHICH == 0) {
            const double r0 = (double)(u - m.x), r1 = (double)(v - m.y);
#pragma unroll
            for (int p = 0; p < 15; ++p)
#pragma unroll
                for (int t = 0; t < 3; ++t) s[3 * p + t] = (double)J[p] * jp0[t] + (double)J[P + p] * jp1[t];
            int e = 45;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int q = t; q < 3; ++q) s[e++] = jp0[t] * jp0[q] + jp1[t] * jp1[q];
#pragma unroll
            for (int t = 0; t < 3; ++t) s[51 + t] = jp0[t] * r0 + jp1[t] * r1;
        } else {
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int t = 0; t < 3; ++t) s[3 * p + t] = (double)J[15 + p] * jp0[t] + (double)J[P + 15 + p] * jp1[t];
        }
        // run key; invalid lanes get one that matches nothing and contribute zeros
        const int ka = valid ? (WHICH == 0 ? c : im) : -1 - lane, kb = valid ? k : -1;
        if (!valid) {
#pragma unroll
            for (int j = 0; j < NV; ++j) s[j] = 0.0;
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int na = __shfl_down(ka, off), nb = __shfl_down(kb, off);
            const bool same = lane + off < 64 && na == ka && nb == kb;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const double t = __shfl_down(s[j], off);
                s[j] += same ? t : 0.0;
            }
        }
        const int pa = __shfl_up(ka, 1), pb = __shfl_up(kb, 1);
        const bool leader = valid && (lane == 0 || pa != ka || pb != kb);
        if (leader) {
            const int64_t gX = a.point_off + 3 * (int64_t)k;
            if constexpr (WHICH == 0) {
#pragma unroll
                for (int p = 0; p < 15; ++p) {
                    double *row = a.H + shared_col<CHAIN>(a, p, c, im) * a.n_params + gX;
#pragma unroll
                    for (int t = 0; t < 3; ++t) unsafeAtomicAdd(row + t, s[3 * p + t]);
                }
                int e = 45;
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int q = t; q < 3; ++q) unsafeAtomicAdd(a.H + (gX + t) * a.n_params + gX + q, s[e++]);
#pragma unroll
                for (int t = 0; t < 3; ++t) unsafeAtomicAdd(a.g + gX + t, s[51 + t]);
            } else {
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    double *row = a.H + shared_col<CHAIN>(a, 15 + p, c, im) * a.n_params + gX;
#pragma unroll
                    for (int t = 0; t < 3; ++t) unsafeAtomicAdd(row + t, s[3 * p + t]);
                }
            }
        }
    }
}

}  // namespace pcs
