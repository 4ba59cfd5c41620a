// ba_blockgram.hpp — dense normal equations  [J^T J | J^T r | sum r^2]  of a GENERATED chain from its materialised block rows
// (SURVEY 8 row f2 for generated chains; round 5).
//
// The three hand-fused chains build their blocked normal equations while they evaluate (ba_normal.hpp).  A generated chain — any
// composition of the reference's function blocks, user blocks included (afb:290-419 / afb:492-652 generate its loss and Jacobian) —
// has one kernel that writes dense block rows (ba_generic.hpp: 2N x P); until round 5 a Levenberg-Marquardt solve on it went through
// conjugate gradients on the products of ba_blockrow.hpp with the host between every two of them.  This kernel makes the exact step
// available: it contracts the block rows into the dense  A = J^T J  (upper triangle, n_params x n_params — a calibration's few hundred
// to few thousand parameters), g = J^T r and the cost, in the packed layout [A | g | cost] that the Schur / Cholesky step
// (ba_schur.hpp with an empty trailing group, ba_chol_persist.hpp) and the device-steered loop (lm_decide_kernel) already consume.
//
// Which global column a local column stands for is the reference's get_block_param_inds (afb:192-233) as a rule: block b covers local
// columns [col0_b, col0_b + np_b) and its parameters of entity e (camera / image / key by link_b) start at start_b + np_b e.  In the
// reference's table order (camera, then image, then key) a run of detections shares camera AND image, so every column that is not
// linked to the key stands for ONE global column over the whole run: the host cuts the table into SEGMENTS (<= GRAM_SEG detections of
// one (camera, image) pair), one wave contracts a segment on the FP64 matrix cores and flushes (P + 1)^2 / 2 sums:
//   * operands straight from global memory in v_mfma_f64_16x16x4's layout: lane l holds element [row 4 s + l / 16][column 16 c + l % 16]
//     of the segment's rows — four row pieces of 128 bytes per load instruction, no LDS staging; the residual rides along as column P
//     (G[p][P] = g_p, G[P][P] = the cost), so one contraction yields all three outputs;
//   * columns linked to the KEY (free points, per-point user parameters) differ from detection to detection in that order.  A product
//     of two columns is a sum over detections that share the entities of BOTH columns, so every pair of columns has a table order in
//     which its destination is constant over runs: pairs of camera / image columns in the table's own order (runs of one (camera,
//     image) pair: pass 0), pairs that involve a key column and otherwise the camera in (camera, key) order (pass 1), image x key pairs
//     in (image, key) order (pass 2).  The host sorts the detections for passes 1 and 2 (an index per detection; rows are gathered
//     through it) and cuts each order into segments; the kernel is the same in every pass and flushes only the pairs that belong to
//     it.  (First version: key columns per detection with atomics — 60 per detection at P = 20: 2.4 ms of a 2.6 ms build on rig-32.)
//   * the segments of a workgroup (16 waves for P < 32) combine in LDS before they go out: sums meet in f64 atomics on A / g / cost
//     (zeroed by the caller), (P + 1)^2 / 2 per group of segments with the same camera (and image, for entries of image-linked columns).
// On MI355X an FP64 MFMA issues at the rate of the FP64 vector pipe (2 048 flop in 64 cycles); what the matrix cores save here is the
// cross-lane reduction — 64 detections' products land summed in the accumulators.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ba_blockrow.hpp"
#include "ba_device.hpp"

namespace pcs {

constexpr int GRAM_SEG = 128;        // detections per segment at most (256 rows = 64 contraction steps)
constexpr int GRAM_MAX_COLS = 64;    // P + 1 <= 64: four column blocks of 16
constexpr int GRAM_SEG_WORDS = 5;

struct BlockGramArgs {
    DetTable tab;
    const double *J;        // 2N x P dense block rows, u row then v row
    const double *resid;    // N x 2
    const int32_t *seg;     // n_seg x GRAM_SEG_WORDS: first position (in `order`, or the detection itself), count, camera, image, key — -1 where the
                            // entity varies inside the segment (the pass does not hold it)
    const int32_t *order;   // passes 1 and 2: position -> detection; NULL = the table's own order
    int32_t pass;           // 0 (camera, image), 1 (camera, key), 2 (image, key): which pairs of columns this launch flushes
    double *A, *g, *cost;   // n_lead x n_lead (upper triangle written), n_params, 1 — zeroed by the caller
    // Blocked form (a chain whose LAST parameter group is one rigid transform per image or one point per key — ba_schur.hpp's trailing
    // entities): products of two leading columns go to A, leading x trailing to B (n_lead x n_trail, row-major), two trailing columns
    // (always the same entity: a detection has one image and one key) to the entity's tb x tb block of C (upper triangle).
    // Dense form: trail_off = n_lead = n_params, B and C unused.
    double *B, *C;
    int64_t n_params, n_lead, n_trail, trail_off;
    int32_t tb;
    int32_t P, n_seg, n_blocks;
    int32_t blk_col0[BLOCKROW_MAX_BLOCKS], blk_np[BLOCKROW_MAX_BLOCKS], blk_link[BLOCKROW_MAX_BLOCKS];   // link: 0 camera, 1 image, 2 key
    int64_t blk_start[BLOCKROW_MAX_BLOCKS];
    const int32_t *stop;    // optional LM stop word (ba_schur.hpp PCS_STOP_GUARD)
    int32_t debug;          // measurements only (option "gram_debug"): 1 = no flush, 2 = no contraction
    // ORDERED mode (option "deterministic"): the contraction stores every segment's matrix, raw, to ws[segment][entry] instead of
    // flushing it; blockrow_gram_reduce_kernel then adds the segments of a GROUP (consecutive segments that agree in the entities of an
    // entry's two columns) in table order — one thread per (group, entry), plain stores, the same bits on every run.  The passes of this
    // mode are chosen so that every pair of columns has its groups consecutive — DET orders (major, minor) = (camera, image), (key, camera),
    // (image, key); see gram_pass_of_det — or, for the image x image pairs of a chain without key columns, listed (grp[2] + gidx).
    double *ws;
    const int32_t *grp[3];  // [0]: groups by the pass's MAJOR entity, [1]: by (major, minor): pairs (first segment, one past the last);
                            // [2]: groups by the MINOR entity — their segments are not consecutive: pairs (first, one past the last) into `gidx`
    const int32_t *gidx;    // segment ids of the minor-entity groups, group after group, each in table order
    int32_t n_grp[3];
    int32_t det;            // 1: `pass` counts the DET orders
};

using gram_d4 = __attribute__((ext_vector_type(4))) double;

constexpr int gram_blocks(int nb) { return nb * (nb + 1) / 2; }                       // upper column blocks of 16 x 16
constexpr int gram_waves(int nb) { return nb == 1 ? 16 : nb == 2 ? 8 : 4; }         // waves per workgroup: their Gram matrices take 32 / 48 / 48 / 80 KB of LDS — two or three
                                                                                     // workgroups per CU, so that one's flush runs under another's loads (16 waves = 96 KB for nb = 2, one
                                                                                     // workgroup per CU: 182 us on rig-32 with P = 17)
constexpr int gram_unroll(int nb) { return nb <= 2 ? 8 : nb == 3 ? 4 : 2; }          // contraction steps (4 rows each) whose loads are in flight together

// where the product of the global columns gp and gc goes (see BlockGramArgs); a parameter that two blocks share (one group, two
// local columns: afb:160-163) meets itself off the local diagonal: that product belongs to the diagonal entry TWICE
__device__ __forceinline__ void gram_add(const BlockGramArgs &a, const int64_t gp, const int64_t gc, const bool same_local, double val) {
    if (gp == gc && !same_local) val += val;
    const int64_t lo = gp < gc ? gp : gc, hi = gp < gc ? gc : gp;
    if (hi < a.trail_off) {
        unsafeAtomicAdd(a.A + lo * a.n_lead + hi, val);
    } else if (lo < a.trail_off) {
        unsafeAtomicAdd(a.B + lo * a.n_trail + (hi - a.trail_off), val);
    } else {
        const int64_t tl = lo - a.trail_off, th = hi - a.trail_off, e = tl / a.tb;
        unsafeAtomicAdd(a.C + e * a.tb * a.tb + (tl - e * a.tb) * a.tb + (th - e * a.tb), val);
    }
}

// NB = column blocks of 16 covering the P + 1 columns (block rows + the residual column).  One wave = one segment; the WAVES segments
// of a workgroup are consecutive in the table, so they mostly share the camera and often the image: their Gram matrices meet in LDS and
// ONE wave per group of equal (camera) / (camera, image) adds the group's sum to global memory — flushing every segment on its own made
// the atomics the larger part of the kernel (1e6 detections, 32 cameras: 125 us of 230; each camera's 15 x 15 block is hit by all of its
// ~400 segments).
// which pass sums the product of two columns with links lp and lc (0 camera, 1 image, 2 key; -1: the residual column, no entity)
__device__ __forceinline__ int gram_pass_of(const int lp, const int lc) {
    const bool key = lp == 2 || lc == 2, img = lp == 1 || lc == 1;
    return !key ? 0 : !img ? 1 : 2;
}

// ORDERED mode: the pass (0 (camera, image), 2 (key, camera), 3 (image, key) — major entity first; 1 is not used) in which the groups of a
// pair of columns can be walked in a fixed order, and which grouping it needs there: 0 = by the major entity, 1 = by (major, minor),
// 2 = everything (the cost), 3 = by the MINOR entity (segments listed through an index: not consecutive).  `keys`: the chain has
// key-linked columns (then image-only pairs ride in pass 3, where the image is the major entity).
__device__ __forceinline__ void gram_pass_of_det(const int lp, const int lc, const bool keys, int &pass, int &cat) {
    const bool cam = lp == 0 || lc == 0, img = lp == 1 || lc == 1, key = lp == 2 || lc == 2;
    if (!cam && !img && !key) { pass = 0; cat = 2; return; }
    if (key) {
        if (img) { pass = 3; cat = 1; }              // (image, key)
        else { pass = 2; cat = cam ? 1 : 0; }        // (key, camera) / (key, key)
        return;
    }
    if (img && !cam) { pass = keys ? 3 : 0; cat = keys ? 0 : 3; return; }   // (image, image): image-major order, or the (camera, image) order through the list
    pass = 0; cat = img ? 1 : 0;                     // (camera, image) / (camera, camera)
}

template <int NB>
__global__ __launch_bounds__(64 * gram_waves(NB)) void blockrow_gram_kernel(const BlockGramArgs a) {
    if (a.stop && *a.stop) return;
    constexpr int WAVES = gram_waves(NB), NE = gram_blocks(NB) * 256, UNROLL = gram_unroll(NB);
    // per local column: base (global column of entity 0), multiplier (parameters per entity), link
    __shared__ int64_t col_base[GRAM_MAX_COLS];
    __shared__ int32_t col_mul[GRAM_MAX_COLS], col_link[GRAM_MAX_COLS];
    __shared__ int32_t seg_id[3][WAVES];   // camera, image, key of the waves' segments (-1: varies / no segment)
    extern __shared__ double gram_lds[];   // WAVES x NE
    if (threadIdx.x < GRAM_MAX_COLS) {
        const int p = threadIdx.x;
        int64_t base = 0;
        int32_t mul = 0, link = -1;
        for (int b = 0; b < a.n_blocks; ++b)
            if (p >= a.blk_col0[b] && p < a.blk_col0[b] + a.blk_np[b]) { base = a.blk_start[b] + (p - a.blk_col0[b]); mul = a.blk_np[b]; link = a.blk_link[b]; }
        col_base[p] = base; col_mul[p] = mul; col_link[p] = link;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_id = blockIdx.x * WAVES + wave;
    const bool have = s_id < a.n_seg;
    const int32_t *sg = a.seg + (int64_t)GRAM_SEG_WORDS * (have ? s_id : 0);
    const int first = have ? sg[0] : 0, count = have ? sg[1] : 0;
    if (lane < 3) seg_id[lane][wave] = have ? sg[2 + lane] : -1;
    const int P = a.P;
    const int n_rows = 2 * count;
    const int lr = lane >> 4, lc = lane & 15;

    // ---- the contraction: G = [J r]' [J r] over the segment's rows, upper column blocks -------------------------------------------
    gram_d4 acc[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = gram_d4{0.0, 0.0, 0.0, 0.0};
    // row q of the segment = component q & 1 of its detection q >> 1; in the table's order the detections are consecutive, in a sorted
    // order they are looked up (four per load instruction: the lanes of a quarter wave share one)
    // The rows of UNROLL steps first (passes 1 and 2 look the detections up: ONE uniform branch around all of those loads — with the
    // look-up inside every step's fetch the operand loads sat in basic blocks of their own and no longer overlapped: + 76 us on rig-32),
    // then all the operand loads together.
    for (int q0 = 0; q0 < n_rows; q0 += 4 * UNROLL) {
        int64_t rows[UNROLL];
        bool ins[UNROLL];
#pragma unroll
        for (int t = 0; t < UNROLL; ++t) {
            const int q = q0 + 4 * t + lr;
            ins[t] = q < n_rows;
            rows[t] = 2 * ((int64_t)first + (q >> 1)) + (q & 1);
        }
        if (a.order) {
#pragma unroll
            for (int t = 0; t < UNROLL; ++t) {
                const int q = q0 + 4 * t + lr;
                rows[t] = 2 * (int64_t)a.order[(int64_t)first + (ins[t] ? (q >> 1) : 0)] + (q & 1);   // unconditional (see below): lanes past the end look up the first
            }
        }
        // every lane loads — from its element, or from a.J[0] where it has none (masked afterwards): loads under per-lane conditions
        // become branches, and hipcc joins branches with `s_waitcnt vmcnt(0)` — the sixteen loads of a round then went out one by one
        // (+ 100 us on rig-32)
        double v[UNROLL][NB];
#pragma unroll
        for (int t = 0; t < UNROLL; ++t)
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                const int col = 16 * cb + lc;
                const bool use = ins[t] && col <= P;
                const double *src = col < P ? a.J + rows[t] * P + col : a.resid + rows[t];
                v[t][cb] = *(use ? src : a.J);
            }
#pragma unroll
        for (int t = 0; t < UNROLL; ++t)
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) v[t][cb] = (ins[t] && 16 * cb + lc <= P) ? v[t][cb] : 0.0;
        if (a.debug & 2) {
#pragma unroll
            for (int t = 0; t < UNROLL; ++t)
#pragma unroll
                for (int i = 0; i < NB; ++i) acc[i][i][0] += v[t][i];
            continue;
        }
#pragma unroll
        for (int t = 0; t < UNROLL; ++t)
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = i; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[t][i], v[t][j], acc[i][j], 0, 0, 0);
    }
    if (a.debug & 1) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = i; j < NB; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (s == 1.2345e301) a.cost[0] = s;
        return;
    }

    if (a.ws) {   // ORDERED mode: the segment's matrix as it is, to its place in the workspace (coalesced 512-byte stores)
        if (have) {
            double *mine = a.ws + (int64_t)s_id * NE + lane;
            int blk = 0;
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = i; j < NB; ++j) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) mine[blk * 256 + q * 64] = acc[i][j][q];
                    ++blk;
                }
        }
        return;
    }
    // ---- the waves' matrices into LDS: entry e = block x 256 + q x 64 + lane  <->  G[16 i + lane / 16 + 4 q][16 j + lane % 16] -------
    {
        double *mine = gram_lds + wave * NE;
        int blk = 0;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = i; j < NB; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) mine[blk * 256 + q * 64 + lane] = acc[i][j][q];
                ++blk;
            }
    }
    __syncthreads();

    // ---- flush: one (wave, entry) pair per thread and pass; the first wave of a group of equal keys adds the group's sum ---------------
    // (a group = consecutive waves whose segments agree in the entities of BOTH columns of the entry; the sorted orders make equal ones
    // consecutive)
    for (int idx = threadIdx.x; idx < WAVES * NE; idx += 64 * WAVES) {
        const int w = idx / NE, e = idx - w * NE;
        if (seg_id[0][w] < 0 && seg_id[1][w] < 0 && seg_id[2][w] < 0) continue;   // no segment
        const int blk = e >> 8, q = (e >> 6) & 3, ln = e & 63;
        int bi = 0, bj = blk;                      // block index -> (bi, bj), bi <= bj, row-major over the upper triangle
        while (bj >= NB - bi) { bj -= NB - bi; ++bi; }
        bj += bi;
        const int p = 16 * bi + (ln >> 4) + 4 * q, c = 16 * bj + (ln & 15);
        if (p > c || c > P) continue;
        const int lp = p < P ? col_link[p] : -1, lcn = c < P ? col_link[c] : -1;
        if (gram_pass_of(lp, lcn) != a.pass) continue;
        const int ip = lp < 0 ? 0 : seg_id[lp][w], ic = lcn < 0 ? 0 : seg_id[lcn][w];   // the entities of the two columns in wave w's segment
        auto same = [&](const int w2) {
            if (seg_id[0][w2] < 0 && seg_id[1][w2] < 0 && seg_id[2][w2] < 0) return false;
            return (lp < 0 || seg_id[lp][w2] == ip) && (lcn < 0 || seg_id[lcn][w2] == ic);
        };
        if (w > 0 && same(w - 1)) continue;        // not the leader of its group
        double sum = gram_lds[idx];
        for (int w2 = w + 1; w2 < WAVES && same(w2); ++w2) sum += gram_lds[w2 * NE + e];
        if (p == P) {                              // (residual, residual): the cost
            unsafeAtomicAdd(a.cost, sum);
            continue;
        }
        const int64_t gp = col_base[p] + (int64_t)col_mul[p] * ip;
        if (c == P) {
            unsafeAtomicAdd(a.g + gp, sum);
            continue;
        }
        gram_add(a, gp, col_base[c] + (int64_t)col_mul[c] * ic, p == c, sum);
    }
}

// ORDERED mode, second step: one workgroup per group of segments (grid = n_grp[0] + n_grp[1] + n_grp[2] (+ 1 in pass 0: the cost); groups by
// the major entity, by (major, minor), by the minor entity through the list).  The workgroup first lists the entries that belong to its (pass, grouping); then, sixteen entries at a
// time, sixteen SLICES of the group's segments are summed side by side (a slice = a fixed contiguous sixteenth, four interleaved partial
// sums inside it) and joined in slice order — a FIXED tree: the same bits whatever the schedule — and one plain add goes to the
// destination.  Nothing else writes it: every pair of columns belongs to one pass and one grouping, every destination to one pair (chains
// whose blocks share a parameter group are refused in this mode: two local pairs would meet in one destination).  A camera's group is
// hundreds of segments long: walked by one thread per entry (first version) the reductions took 0.9 ms of rig-32's build.
template <int NB>
__global__ __launch_bounds__(256) void blockrow_gram_reduce_kernel(const BlockGramArgs a, const int32_t keys) {
    if (a.stop && *a.stop) return;
    constexpr int NE = gram_blocks(NB) * 256;
    __shared__ int64_t col_base[GRAM_MAX_COLS];
    __shared__ int32_t col_mul[GRAM_MAX_COLS], col_link[GRAM_MAX_COLS];
    __shared__ int16_t list[NE];
    __shared__ int32_t n_list;
    __shared__ double red[16][17];
    const int tid = threadIdx.x;
    if (tid < GRAM_MAX_COLS) {
        const int p = tid;
        int64_t base = 0;
        int32_t mul = 0, link = -1;
        for (int b = 0; b < a.n_blocks; ++b)
            if (p >= a.blk_col0[b] && p < a.blk_col0[b] + a.blk_np[b]) { base = a.blk_start[b] + (p - a.blk_col0[b]); mul = a.blk_np[b]; link = a.blk_link[b]; }
        col_base[p] = base; col_mul[p] = mul; col_link[p] = link;
    }
    if (tid == 0) n_list = 0;
    __syncthreads();
    const int gb = blockIdx.x;
    int cat, lo, hi;
    const int32_t *via = nullptr;   // cat 3: position -> segment
    if (gb < a.n_grp[0]) { cat = 0; lo = a.grp[0][2 * gb]; hi = a.grp[0][2 * gb + 1]; }
    else if (gb < a.n_grp[0] + a.n_grp[1]) { cat = 1; lo = a.grp[1][2 * (gb - a.n_grp[0])]; hi = a.grp[1][2 * (gb - a.n_grp[0]) + 1]; }
    else if (gb < a.n_grp[0] + a.n_grp[1] + a.n_grp[2]) { cat = 3; lo = a.grp[2][2 * (gb - a.n_grp[0] - a.n_grp[1])]; hi = a.grp[2][2 * (gb - a.n_grp[0] - a.n_grp[1]) + 1]; via = a.gidx; }
    else { cat = 2; lo = 0; hi = a.n_seg; }
    const int32_t *sg = a.seg + (int64_t)GRAM_SEG_WORDS * (via ? via[lo] : lo);
    const int ids[3] = {sg[2], sg[3], sg[4]};
    const int P = a.P;
    auto decode = [&](const int e, int &p, int &c) {
        const int blk = e >> 8, q = (e >> 6) & 3, ln = e & 63;
        int bi = 0, bj = blk;
        while (bj >= NB - bi) { bj -= NB - bi; ++bi; }
        bj += bi;
        p = 16 * bi + (ln >> 4) + 4 * q; c = 16 * bj + (ln & 15);
    };
    // the entries of this (pass, grouping): their order in the list is of no consequence (every entry is summed on its own)
    for (int e = tid; e < NE; e += 256) {
        int p, c;
        decode(e, p, c);
        if (p > c || c > P) continue;
        int pass, pc;
        gram_pass_of_det(p < P ? col_link[p] : -1, c < P ? col_link[c] : -1, keys != 0, pass, pc);
        if (pass == a.pass && pc == cat) list[atomicAdd(&n_list, 1)] = (int16_t)e;
    }
    __syncthreads();
    const int n_valid = n_list;
    const int col = tid & 15, slice = tid >> 4;
    const int len = hi - lo, per = (len + 15) / 16;
    const int s_lo = lo + slice * per, s_hi = min(hi, s_lo + per);
    for (int base = 0; base < n_valid; base += 16) {
        const int idx = base + col;
        const int e = idx < n_valid ? list[idx] : 0;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (idx < n_valid && s_lo < s_hi) {
            if (via) {   // listed segments
                int s = s_lo;
                for (; s + 4 <= s_hi; s += 4) {
                    s0 += a.ws[(int64_t)via[s] * NE + e]; s1 += a.ws[(int64_t)via[s + 1] * NE + e];
                    s2 += a.ws[(int64_t)via[s + 2] * NE + e]; s3 += a.ws[(int64_t)via[s + 3] * NE + e];
                }
                for (; s < s_hi; ++s) s0 += a.ws[(int64_t)via[s] * NE + e];
            } else {
                const double *w = a.ws + (int64_t)s_lo * NE + e;
                int s = s_lo;
                for (; s + 4 <= s_hi; s += 4, w += 4 * (int64_t)NE) { s0 += w[0]; s1 += w[NE]; s2 += w[2 * (int64_t)NE]; s3 += w[3 * (int64_t)NE]; }
                for (; s < s_hi; ++s, w += NE) s0 += w[0];
            }
        }
        red[slice][col] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (slice == 0 && idx < n_valid) {
            double sum = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) sum += red[q][col];   // slice order
            int p, c;
            decode(e, p, c);
            if (p == P) {
                *a.cost += sum;
            } else {
                const int64_t gp = col_base[p] + (int64_t)col_mul[p] * ids[col_link[p]];
                if (c == P) {
                    a.g[gp] += sum;
                } else {
                    const int64_t gc = col_base[c] + (int64_t)col_mul[c] * ids[col_link[c]];
                    const int64_t lo_c = gp < gc ? gp : gc, hi_c = gp < gc ? gc : gp;
                    double *dst;
                    if (hi_c < a.trail_off) dst = a.A + lo_c * a.n_lead + hi_c;
                    else if (lo_c < a.trail_off) dst = a.B + lo_c * a.n_trail + (hi_c - a.trail_off);
                    else {
                        const int64_t tl = lo_c - a.trail_off, th = hi_c - a.trail_off, en = tl / a.tb;
                        dst = a.C + en * a.tb * a.tb + (tl - en * a.tb) * a.tb + (th - en * a.tb);
                    }
                    *dst += sum;
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace pcs
