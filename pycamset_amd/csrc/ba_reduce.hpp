// ba_reduce.hpp — the ORDERED second pass of the normal equations (engine option "deterministic", round 5).
//
// ba_normal_mfma_kernel adds a finished run's accumulators to H / g / cost with f64 atomics, in the order the waves happen to
// arrive: two builds of the same system differ in their last bits, and so do the ranks of a sharded solve (the reference's path is
// deterministic for a given thread count: fixed contiguous chunks, abstract_function_blocks.py:281-288, a serial inner loop,
// afb:356-387, and scipy on one thread).  In deterministic mode nothing is added in arrival order:
//
//   1. the build kernel STORES the accumulator tile of every SEGMENT — a maximal stretch of detections inside one run
//      ((cam, image) / (cam, key)) and inside one wave's tile range — raw, 512 bytes per register index, into its own slot of a
//      workspace.  Which segment a flush belongs to is static: the table and the launch geometry fix it (`seg_base[wave]` + the
//      number of flushes the wave has done), so the host builds every index structure below ONCE per table;
//   2. normal_reduce_runs_kernel: one workgroup per GROUP of up to four logical runs of one camera, one wave per run.  Per run it sums the run's segments in table
//      order and sorts the entries by what they still have to meet:
//        RUN  one camera-level and one entity-level column (camera x pose, camera x point): complete — stored to H;
//        ENT  entity-level columns only (pose x pose, pose x r; point x point, point x r): the entity's other cameras are
//             missing — parked in Q[run][72];
//        CAM  camera-level columns only (intrinsics / extrinsics / r): summed over the group's runs in order, parked in G[group][256];
//   3. normal_reduce_final_kernel: one workgroup per camera sums G over the camera's groups in order, one thread per entry; one
//      workgroup sums the cost over all groups in a fixed tree; entity workgroups sum Q over the entity's runs in camera order.
// Every destination has ONE writer (plain stores) and every sum a fixed order: bit-identical results run to run and rank to rank.
// The (image, key) pass of the self chain (ba_normal_imgkey_kernel) keeps its atomics: a run there is at most n_cams <= 64 detections
// long, so it meets at most one tile boundary and every address receives at most TWO contributions onto the zero the prologue wrote —
// and a + b = b + a in IEEE arithmetic.
#pragma once
#include <hip/hip_runtime.h>

#include "ba_normal.hpp"

namespace pcs {

constexpr int RED_NONE = 0, RED_RUN = 1, RED_ENT = 2, RED_CAM = 3;
constexpr int RED_Q = 72;      // doubles per logical run in Q: (oR, oC) of the entity's diagonal block at oR * 8 + oC, its part of g at 64 + o
constexpr int RED_G = 256;     // doubles per group in G: camera-level columns l = 0..14 (intrinsics 0-8, extrinsics 9-14): (la, lb) at la * 16 + lb,
                               // g at 240 + l, the cost at 255
constexpr int RED_RUNS_PER_GROUP = 4;
constexpr int RED_SPLIT_SEGS = 8;   // passes without RUN / ENT entries (free chain, shared pass: one run per camera) cut a long run into pieces of this many segments

// class | position << 2 of accumulator register r of lane `lane` of MFMA m (see the header comment); host-callable
template <int CHAIN, int PASS>
__host__ __device__ __forceinline__ int reduce_descriptor(const int m, const int lane, const int r) {
    constexpr bool HAS_POSE = CHAIN != CHAIN_FREE;
    constexpr int NS = normal_shared_cols(CHAIN);
    constexpr int E = PASS == PASS_SHARED ? 2 : 3;   // the entity group of the pass's runs: pose (shared pass), point (cam, key pass)
    auto col_group = [](const int lc) -> int { return lc == NORMAL_R ? 4 : lc < 9 ? 0 : lc < 15 ? 1 : (HAS_POSE && lc < NS) ? 2 : 3; };
    auto col_offset = [](const int lc) -> int { return lc == NORMAL_R ? 0 : lc < 9 ? lc : lc < 15 ? lc - 9 : (HAS_POSE && lc < NS) ? lc - 15 : lc - NS; };
    int sa, sb;
    if (!entry_kept<CHAIN, PASS>(m, (lane >> 4) + 4 * r, lane & 15, sa, sb)) return RED_NONE;
    const int la = slot_col<CHAIN, PASS>(sa), lb = slot_col<CHAIN, PASS>(sb);
    int gR = col_group(la), oR = col_offset(la), gC = col_group(lb), oC = col_offset(lb);
    if (gR > gC || (gR == gC && oR > oC)) { int t = gR; gR = gC; gC = t; t = oR; oR = oC; oC = t; }   // row <= column; the residual (4) ends up as the column
    auto cam_col = [](const int g, const int o) { return g == 1 ? 9 + o : o; };
    if (gR == 4) return RED_CAM | (255 << 2);                                   // r . r
    if (gC == 4) {                                                              // J^T r of column (gR, oR)
        if (gR <= 1) return RED_CAM | ((240 + cam_col(gR, oR)) << 2);
        return gR == E ? (RED_ENT | ((64 + oR) << 2)) : RED_NONE;
    }
    if (gC <= 1) return RED_CAM | ((cam_col(gR, oR) * 16 + cam_col(gC, oC)) << 2);   // both camera-level (gR <= gC)
    if (gR == E && gC == E) return RED_ENT | ((oR * 8 + oC) << 2);
    if (gR <= 1 && gC == E) return RED_RUN;
    return RED_NONE;   // (not reached: the passes own no other pairs)
}

struct ReduceArgs {
    const double *part;        // n_seg slots of NM * 256 doubles: register (m, r) of lane l at (m * 4 + r) * 64 + l
    double *Q;                 // n_lr x RED_Q
    double *G;                 // n_grp x RED_G
    const int32_t *lr_ptr;     // n_lr + 1: the segments of logical run r are lr_segs[lr_ptr[r] .. lr_ptr[r + 1]), in table order
    const int32_t *lr_segs;
    const int32_t *lr_ka, *lr_kb;   // the run's keys: (cam, image) in the shared pass, (cam, key) in the (cam, key) pass
    const int32_t *grp_ptr;    // n_grp + 1: the logical runs of group g (all of one camera)
    const int32_t *cam_ptr;    // n_cams + 1: the groups of camera c
    const int32_t *ent_ptr;    // n_ent + 1: the logical runs of entity e (image / key) are ent_runs[ent_ptr[e] ..), in camera order
    const int32_t *ent_runs;
    int32_t n_lr, n_grp, n_cams, n_ent;
};

// Step 2 of the header comment.  One workgroup per group, one WAVE per logical run (round 5, second version: the first walked the
// group's runs one after the other — three levels of dependent loads per run, 29 us on rig-32 with 1.6 waves per SIMD):
//   level 1  the run's record (segment range, keys): uniform loads;
//   level 2  the ids of its segments, one per lane (up to 64 in one request);
//   level 3  the segments' slots, four in flight, summed in table order.
// RUN entries go straight to H: their destination is  P + (rb[gR] + oR) ld + cb + oC  with P, ld, cb, rb[] wave-uniform per run and
// gR, oR, oC static per register (entry_descriptor's fields; same address as the flush of ba_normal_mfma_kernel).  ENT entries are
// parked in Q[run]; CAM entries meet the group's other runs in LDS and are summed in run order into G[group].
template <int CHAIN, int PASS>
__global__ __launch_bounds__(256) void normal_reduce_runs_kernel(const NormalArgs a, const ReduceArgs ra) {
    if (a.stop && *a.stop) return;
    constexpr int NM = normal_mfmas(CHAIN, PASS);
    constexpr int E = PASS == PASS_SHARED ? 2 : 3;
    __shared__ double cam_stage[RED_RUNS_PER_GROUP][RED_G];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x;
    const int64_t shift = (a.sel && *a.sel) ? a.alt : 0;
    const int r0 = ra.grp_ptr[g], nr = ra.grp_ptr[g + 1] - r0;
    int ent[NM][4], rd[NM][4];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ent[m][r] = entry_descriptor<CHAIN, PASS>(m, lane, r, a.trail_group);
            rd[m][r] = reduce_descriptor<CHAIN, PASS>(m, lane, r);
        }
    if (wave < nr) {
        const int run = r0 + wave;
        const int k0 = __builtin_amdgcn_readfirstlane(ra.lr_ptr[run]), nseg = __builtin_amdgcn_readfirstlane(ra.lr_ptr[run + 1]) - k0;
        const int ka = __builtin_amdgcn_readfirstlane(ra.lr_ka[run]), kb = __builtin_amdgcn_readfirstlane(ra.lr_kb[run]);
        const int sid_lane = lane < nseg ? ra.lr_segs[k0 + lane] : 0;
        double acc[NM][4];
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][r] = 0.0;
        for (int t = 0; t < nseg; t += 4) {   // four slots in flight; added in table order (a missing one adds +0.0)
            double v[4][NM][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool have = t + u < nseg;   // uniform
                int sid = 0;
                if (have) sid = t + u < 64 ? __builtin_amdgcn_readlane(sid_lane, (t + u) & 63) : __builtin_amdgcn_readfirstlane(ra.lr_segs[k0 + t + u]);
                const double *p = ra.part + (int64_t)sid * (NM * 256) + lane;
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[u][m][r] = have ? p[(m * 4 + r) * 64] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[m][r] += v[u][m][r];
        }
        // the run's destination in H (wave-uniform part)
        const bool ent_trails = a.trail_group == E;
        double *P = (ent_trails ? a.HB : a.H) + shift;
        const int64_t ld = ent_trails ? a.ldB : a.ldA;
        const int64_t cb = ent_trails ? (int64_t)a.tb * kb : (E == 2 ? a.pose_off + 6 * (int64_t)kb : a.point_off + 3 * (int64_t)kb);
        const int64_t rb0 = 9 * (int64_t)ka, rb1 = a.extr_off + 6 * (int64_t)ka;
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = ent[m][r], cls = rd[m][r] & 3, pos = rd[m][r] >> 2;
                if (cls == RED_RUN) {
                    const int gR = (d >> 8) & 3, oR = d & 15, oC = (d >> 4) & 15;
                    P[((gR ? rb1 : rb0) + oR) * ld + cb + oC] = acc[m][r];
                } else if (cls == RED_ENT) {
                    ra.Q[(int64_t)run * RED_Q + pos] = acc[m][r];
                } else if (cls == RED_CAM) {
                    cam_stage[wave][pos] = acc[m][r];
                }
            }
    }
    if (PASS != PASS_SHARED) return;   // the (cam, key) pass owns no camera-level entries (uniform: no barrier is skipped by part of a workgroup)
    __syncthreads();
    {
        const int p = threadIdx.x, la = p >> 4, lb = p & 15;
        if (!((p < 240 && la <= lb && lb < 15) || (p >= 240 && p <= 255))) return;   // positions no register owns
        double s = 0.0;
        for (int w = 0; w < nr; ++w) s += cam_stage[w][p];
        ra.G[(int64_t)g * RED_G + p] = s;
    }
}

// Step 3.  Grid: [n_cams camera workgroups | 1 cost workgroup] (shared pass only) + ceil(n_ent / 3) entity workgroups.
template <int CHAIN, int PASS>
__global__ __launch_bounds__(256) void normal_reduce_final_kernel(const NormalArgs a, const ReduceArgs ra) {
    if (a.stop && *a.stop) return;
    constexpr bool HAS_CAM = PASS == PASS_SHARED;
    constexpr int E = PASS == PASS_SHARED ? 2 : 3;
    constexpr int TBE = E == 2 ? 6 : 3;          // columns of one entity
    const int64_t shift = (a.sel && *a.sel) ? a.alt : 0;
    const int tid = threadIdx.x;
    const int cam_blocks = HAS_CAM ? ra.n_cams + 1 : 0;
    if (HAS_CAM && (int)blockIdx.x < ra.n_cams) {
        const int c = blockIdx.x, p = tid;
        const int g0 = ra.cam_ptr[c], g1 = ra.cam_ptr[c + 1];
        const int la = p >> 4, lb = p & 15;
        const bool is_h = p < 240 && la <= lb && lb < 15, is_g = p >= 240 && p < 255;
        if (g0 == g1 || !(is_h || is_g)) return;          // a camera without detections keeps the prologue's zeros
        double s = 0.0;
        for (int g = g0; g < g1; g += 8) {   // eight loads in flight, added in group order (a missing one adds +0.0)
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = g + u < g1 ? ra.G[(int64_t)(g + u) * RED_G + p] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        auto col = [&](const int l) -> int64_t { return l < 9 ? 9 * (int64_t)c + l : a.extr_off + 6 * (int64_t)c + (l - 9); };
        if (is_h) a.H[shift + col(la) * a.ldA + col(lb)] = s;
        else a.g[shift + col(p - 240)] = s;
        return;
    }
    if (HAS_CAM && (int)blockIdx.x == ra.n_cams) {        // the cost: every group's share, in a fixed tree
        __shared__ double red[256];
        double s = 0.0;
        for (int g = tid; g < ra.n_grp; g += 8 * 256) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = g + 256 * u < ra.n_grp ? ra.G[(int64_t)(g + 256 * u) * RED_G + 255] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        red[tid] = s;
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if (tid < h) red[tid] += red[tid + h];
            __syncthreads();
        }
        if (tid == 0) a.cost[shift] = red[0];
        return;
    }
    const int e = ((int)blockIdx.x - cam_blocks) * 3 + tid / RED_Q, p = tid % RED_Q;
    if (tid >= 3 * RED_Q || e >= ra.n_ent) return;
    const int oR = p >> 3, oC = p & 7;
    const bool is_h = p < 64 && oR <= oC && oC < TBE, is_g = p >= 64 && p < 64 + TBE;
    const int k0 = ra.ent_ptr[e], k1 = ra.ent_ptr[e + 1];
    if (k0 == k1 || !(is_h || is_g)) return;
    double s = 0.0;
    for (int k = k0; k < k1; k += 8) {   // eight loads in flight, added in camera order
        int run[8];
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) run[u] = k + u < k1 ? ra.ent_runs[k + u] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = run[u] >= 0 ? ra.Q[(int64_t)run[u] * RED_Q + p] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    const int64_t base = (E == 2 ? a.pose_off + 6 * (int64_t)e : a.point_off + 3 * (int64_t)e);
    if (is_g) a.g[shift + base + (p - 64)] = s;
    else if (a.trail_group == E) a.HC[shift + (int64_t)a.tb * a.tb * e + (int64_t)oR * a.tb + oC] = s;   // blocked layout, the entity is the trailing group
    else a.H[shift + (base + oR) * (int64_t)a.ldA + base + oC] = s;                                   // a leading group, or the dense layout
}

}  // namespace pcs
