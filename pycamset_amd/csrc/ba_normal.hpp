// ba_normal.hpp — block-reduced normal equations on the FP64 matrix cores ("block-reduced
// J^T J / J^T r, then all-reduce of the small result instead of all-gather of J").
//
// Built on the GPU without ever writing J:
//     H    = J^T J   upper triangle of the n_params x n_params matrix (row-major, full parameter-string space)
//     g    = J^T r   n_params
//     cost = r^T r
// This is what a Levenberg-Marquardt step needs from the Jacobian the reference hands to scipy
// (optimisation_handling.py:88-98); with it one LM iteration costs one pass + a small dense solve
// instead of ~150 matrix-free J^T(Jv) products.
//
// Structure.  A detection's 2 x P block touches the 15 columns of its camera, the 6 of its image and (self / free
// chains) the 3 of its key.  Over a RUN of detections that share two of those, the matching blocks of H are a
// Gram matrix X^T X of the run's (2 x detections) rows — the one dense contraction of this engine, and the only
// place where it uses MFMA (v_mfma_f64_16x16x4_f64: 64 cycles per issue = the FP64 vector rate on gfx950,
// tools/probes/mfma_f64_probe.hip; the gain is operand delivery — one 8-byte LDS read per lane feeds 256 FMAs,
// where the round-1 VALU form read 16 B per 2 FMAs and was LDS-bandwidth-bound: 72 us of LDS array time at N = 1e6).
//
//   PASS_SHARED  table order (cam -> image -> key; any order works, a scattered table is walked through a
//                (cam, image)-sorted permutation): X = [J_cam (15) | J_pose (6) | r], 22 columns (16 for the free
//                chain).  Runs = (cam, image).  Yields the camera, pose and camera-pose blocks, their part of g,
//                and the cost.
//   PASS_CAMKEY  (cam, key)-sorted order: X = [J_cam | J_point (3) | r].  Runs = (cam, key).  Yields the
//                camera-point blocks E[c,k], the point blocks D[k] and the point part of g.
//   PASS_IMGKEY  (image, key)-sorted order (self chain): X = [J_pose | J_point].  Yields the pose-point blocks F[i,k].
//
// Per tile of 64 detections a wave
//   1. evaluates one detection per lane (eval_detection, as the fused kernel does) and writes its two rows
//      column-major into a wave-private LDS image: column slot s holds the 2 x ROWS values of column s as 16-byte
//      (u, v) pairs (ds_write_b128 at consecutive addresses), slot stride = 16 B mod 256 B so that the operand
//      reads below spread over all banks;
//   2. walks the image in k-steps of 4 rows (2 detections): every lane reads ONE double per operand window
//      (ds_read_b64: lane l holds X[4s + (l >> 4)][slot_w(l & 15)]) and issues the step's MFMAs into accumulator
//      tiles that live in registers for as long as the run lasts.  22 columns need the upper triangle of a 3 x 3
//      arrangement of column groups G0 = 0-7, G1 = 8-15, G2 = 16-21; choosing the operands as
//          D_a = (G0,G2)^T (G0,G1)      D_b = (G1,G2)^T (G1,G2)
//      covers all six group pairs with TWO MFMAs per step instead of three (free chain: one; point passes: one);
//   3. at a run boundary (wave-uniform bit mask from a ballot; a step that straddles one is split with masked
//      operands) adds the finished accumulators to H / g / cost with one f64 atomic per entry.  In PASS_SHARED the
//      entries that involve no pose column belong to the camera alone and stay in registers until the camera
//      changes (otherwise ~200 serialised adds per address of a camera block).
// Atomic order makes the last bits run-to-run dependent (documented; tests compare with a tolerance).
#pragma once
#include <hip/hip_runtime.h>

#include "ba_device.hpp"
#include "ba_matfree.hpp"  // wave_sum

namespace pcs {

struct NormalArgs {
    DetTable tab;
    const int32_t *order;  // visit the detections in this order (a sorted permutation), or NULL = table order
    const void *cam_slab, *pose_slab, *points;
    // Output, zeroed by the prologue launch.  Two layouts of J^T J (round 3):
    //   dense    H = n_params x n_params, upper triangle written (ldA = n_params, trail_group = -1);
    //   blocked  the parameter string is split at trail_off into a LEADING part (cameras; + poses for the self chain) and
    //            the TRAILING group whose entities never share a detection (poses of the template chain, points of the
    //            self / free chains; `tb` = 6 or 3 columns each):
    //                H  = A  n_lead x n_lead, upper triangle           (leading x leading, ld = ldA = n_lead)
    //                HB = B  n_lead x n_trail                          (leading x trailing, ld = ldB = n_trail)
    //                HC = C  (n_trail / tb) blocks of tb x tb, upper   (trailing x trailing: block diagonal)
    //            — what a Schur-complement LM step consumes (ba_schur.hpp), and all that is stored: rig-32-self 42 MB
    //            instead of 79 MB, the free chain with 2e4 points 0.23 GB instead of 29 GB.
    double *H;
    double *HB, *HC;
    double *g;      // n_params (parameter-string order in both layouts)
    double *cost;   // 1
    int32_t ldA, ldB, tb;
    int32_t trail_group;   // 2 = pose, 3 = point, -1 = dense layout
    int64_t trail_off;     // first parameter-string column of the trailing group
    int64_t n, n_tiles;
    int64_t extr_off, pose_off, point_off;
    int64_t n_params;
    int32_t tiles_per_wave;
    const int32_t *stop;   // optional device word: non-zero = do nothing (a build queued behind the end of an LM loop, ba_schur.hpp)
    // LM loop with two packed states (ba_schur.hpp SchurArgs::sel): H, HB, HC, g, cost above are where the TRIAL state goes while state 0
    // is current; when *sel != 0 the trial state is the other buffer, `alt` doubles further on (alt may be negative)
    const int32_t *sel;
    int64_t alt;
    // Deterministic mode (csrc/ba_reduce.hpp): instead of adding a finished run's accumulators to H / g / cost with atomics, the kernel
    // stores them raw into slot seg_base[workgroup] + (flushes this wave has done) of `part` — NM * 256 doubles per slot, register (m, r)
    // of lane l at (m * 4 + r) * 64 + l — and an ordered second pass sums the slots.  NULL = atomics.
    double *part;
    const int32_t *seg_base;
    int32_t debug;  // profiling switches (results are wrong while set): 2 no flush atomics, 8 no MFMA phase, 16 no evaluation,
                    // 32 run boundaries ignored, 64 no LDS image writes, 128 flush = clear only; host side: 256 / 512 / 1024 skip the shared /
                    // (cam, key) / (image, key) pass
};

constexpr int PASS_SHARED = 0, PASS_CAMKEY = 1, PASS_IMGKEY = 2;
constexpr int NORMAL_R = 30;                       // local id of the residual column
constexpr int normal_shared_cols(int chain) { return chain == CHAIN_FREE ? 15 : 21; }
// columns of the LDS image per pass
constexpr int normal_slots(int chain, int pass) {
    return pass == PASS_SHARED ? normal_shared_cols(chain) + 1 : pass == PASS_CAMKEY ? 19 : 9;
}
constexpr int normal_mfmas(int chain, int pass) { return (pass == PASS_SHARED && chain != CHAIN_FREE) ? 2 : 1; }
// bytes between column slots: 2 * ROWS doubles + 16 -> slot s starts on bank (4 s) mod 64
constexpr int normal_slot_stride(int rows) { return 2 * rows * 8 + 8; }   // an ODD number of doubles: the 16 slots a quarter wave reads an operand from start on 16 different bank pairs
constexpr int normal_lds_bytes(int chain, int pass, int rows) { return normal_slots(chain, pass) * normal_slot_stride(rows) + 128; }   // + 128: the walk's look-ahead reads (up to three k-steps) past the last one

// local column id (index into a J row; NORMAL_R = residual) stored in slot s of the image
template <int CHAIN, int PASS>
__host__ __device__ __forceinline__ constexpr int slot_col(int s) {
    constexpr int NS = normal_shared_cols(CHAIN);     // = first point column of a J row (self: 21, free: 15)
    if (PASS == PASS_SHARED) return s < NS ? s : NORMAL_R;
    if (PASS == PASS_CAMKEY) return s < 15 ? s : s < 18 ? NS + (s - 15) : NORMAL_R;
    return s < 6 ? 15 + s : NS + (s - 6);
}

// operand window w of a pass: slot read by the lanes with (lane & 15) == j, or -1 (padding: any slot, result unused)
template <int CHAIN, int PASS>
__host__ __device__ __forceinline__ int window_slot(int w, int j) {
    if (PASS == PASS_SHARED) {
        if (CHAIN == CHAIN_FREE) return j;                                   // W0 = slots 0..15
        if (w == 1) return j;                                                // W1 = (G0, G1)
        const int g = w == 0 ? 0 : 8;                                        // W0 = (G0, G2), W2 = (G1, G2)
        return j < 8 ? g + j : j < 14 ? 16 + (j - 8) : -1;
    }
    if (PASS == PASS_CAMKEY) {
        if (w == 0) return j < 3 ? 15 + j : j == 3 ? 18 : j - 4;             // A = [pt | r | cam 0..11]
        return j < 3 ? 15 + j : j < 6 ? 12 + (j - 3) : -1;                   // B = [pt | cam 12..14]
    }
    if (w == 0) return j < 6 ? j : -1;                                       // A = [pose]
    return j < 3 ? 6 + j : -1;                                               // B = [pt]
}
template <int CHAIN, int PASS> constexpr int n_windows() { return (PASS == PASS_SHARED) ? (CHAIN == CHAIN_FREE ? 1 : 3) : 2; }
// MFMA m multiplies window mfma_a(m)^T by window mfma_b(m)
template <int CHAIN, int PASS> __host__ __device__ __forceinline__ constexpr int mfma_a(int m) { return PASS == PASS_SHARED ? (CHAIN == CHAIN_FREE ? 0 : m == 0 ? 0 : 2) : 0; }
template <int CHAIN, int PASS> __host__ __device__ __forceinline__ constexpr int mfma_b(int m) { return PASS == PASS_SHARED ? (CHAIN == CHAIN_FREE ? 0 : m == 0 ? 1 : 2) : 1; }

// Is D_m[i][j] an entry this pass owns (each unordered column pair exactly once)?  sa / sb = its two slots.
template <int CHAIN, int PASS>
__host__ __device__ __forceinline__ bool entry_kept(int m, int i, int j, int &sa, int &sb) {
    sa = window_slot<CHAIN, PASS>(mfma_a<CHAIN, PASS>(m), i);
    sb = window_slot<CHAIN, PASS>(mfma_b<CHAIN, PASS>(m), j);
    if (sa < 0 || sb < 0) return false;
    if (PASS == PASS_SHARED) {
        if (CHAIN == CHAIN_FREE) return i <= j;
        if (m == 0) return i >= 8 || i <= j;          // G0 x (G0 upper | G1), G2 x (G0, G1) in full
        return i <= j && !(i < 8 && j >= 8);          // G1 x G1 upper, G2 x G2 upper (G1 x G2 came from m = 0)
    }
    if (PASS == PASS_CAMKEY) {
        if (i < 3) return j >= 3 || i <= j;           // pt x pt upper, pt x cam 12..14
        return j < 3;                                 // r x pt, cam 0..11 x pt
    }
    return true;                                      // pose x pt
}

// Descriptor of accumulator register r of lane `lane` of MFMA m (see the flush of ba_normal_mfma_kernel): where the entry goes.
//   oR | oC << 4 | row entry << 8 | column entry << 13 | ld entry << 18 | pointer entry << 23 | owned << 27 | pose << 28
// the "entries" being lanes of the look-up table the flush refreshes per run:
//   lanes  0- 3  ldA base[g]              row offset of a leading group inside A
//   lanes  4- 7  ldB base[g]              row offset of a leading group inside B
//   lane   8     tb tb entity             the run's trailing block inside C        lanes 9, 15: zero
//   lanes 12-14  ldA, ldB, tb             row lengths
//   lanes 16-19  base[g] - (g == trail_group ? trail_off : 0)   column offset: leading columns keep their
//                parameter-string index, trailing ones are local to the trailing part
//   lanes 24-27  base[g]                  index into g (parameter-string order)
// (doubles; every entry number fits its 5-bit field).  Pointer entries: 0 A, 1 B, 2 C, 3 g, 4 cost.
// Host-callable: tests/test_host_logic.py decodes these descriptors for sample runs and checks every owned entry against the
// column pair pcs_normal_entry_map reports for it (pcs_normal_descriptors).
template <int CHAIN, int PASS>
__host__ __device__ __forceinline__ int entry_descriptor(const int m, const int lane, const int r, const int tg) {
    constexpr bool HAS_POSE = CHAIN != CHAIN_FREE;
    constexpr int NS = normal_shared_cols(CHAIN);
    auto col_group = [](const int lc) -> int { return lc == NORMAL_R ? 4 : lc < 9 ? 0 : lc < 15 ? 1 : (HAS_POSE && lc < NS) ? 2 : 3; };
    auto col_offset = [](const int lc) -> int { return lc == NORMAL_R ? 0 : lc < 9 ? lc : lc < 15 ? lc - 9 : (HAS_POSE && lc < NS) ? lc - 15 : lc - NS; };
    int sa, sb;
    const bool keep = entry_kept<CHAIN, PASS>(m, (lane >> 4) + 4 * r, lane & 15, sa, sb);
    const int la = slot_col<CHAIN, PASS>(keep ? sa : 0), lb = slot_col<CHAIN, PASS>(keep ? sb : 0);
    int gR = col_group(la), oR = col_offset(la), gC = col_group(lb), oC = col_offset(lb);
    const bool pose = PASS == PASS_SHARED && (gR == 2 || gC == 2);
    if (gR > gC || (gR == gC && oR > oC)) { int t = gR; gR = gC; gC = t; t = oR; oR = oC; oC = t; }   // row <= column; the residual (4) ends up as the column
    int eRow, eCol, eLd, ePtr;
    if (gR == 4) { oR = 0; oC = 0; eRow = 9; eCol = 9; eLd = 15; ePtr = 4; }                       // r . r -> cost
    else if (gC == 4) { oC = oR; oR = 0; eRow = 9; eCol = 24 + gR; eLd = 15; ePtr = 3; }           // J^T r -> g
    else if (gR == tg) { eRow = 8; eCol = 9; eLd = 14; ePtr = 2; }                                 // trailing x trailing -> C
    else if (gC == tg) { eRow = 4 + gR; eCol = 16 + gC; eLd = 13; ePtr = 1; }                      // leading x trailing -> B
    else { eRow = gR; eCol = 16 + gC; eLd = 12; ePtr = 0; }                                        // leading x leading -> A
    return oR | (oC << 4) | (eRow << 8) | (eCol << 13) | (eLd << 18) | (ePtr << 23) | (keep ? 1 << 27 : 0) | (pose ? 1 << 28 : 0);
}

using d4v = __attribute__((ext_vector_type(4))) double;

// Two waves per SIMD: the 22.9 KB image (ROWS = 64) allows 7 one-wave workgroups per CU anyway; the half-tile form (ROWS = 32,
// 11.4 KB) keeps J live across its first MFMA walk and needs 232 VGPRs — held to 168 for a third wave it spills 256 B
// (measured in round 2 with the same outcome: 99 us against 92).  Zero scratch in every instantiation as it stands.
template <int CHAIN, int PASS, int ROWS>
__global__ __launch_bounds__(64, 2) void ba_normal_mfma_kernel(const NormalArgs a) {
    static_assert(ROWS == 64 || ROWS == 32, "image holds a whole or half a tile");
    static_assert(PASS == PASS_SHARED || CHAIN != CHAIN_TEMPLATE, "the template chain has no point columns");
    static_assert(PASS != PASS_IMGKEY || CHAIN == CHAIN_SELF, "only the self chain couples poses and points");
    using T = double;
    constexpr int P = chain_P(CHAIN);
    constexpr int P2 = 2 * P;
    constexpr int NSLOT = normal_slots(CHAIN, PASS);
    constexpr int NW = n_windows<CHAIN, PASS>();
    constexpr int NM = normal_mfmas(CHAIN, PASS);
    constexpr int KS = normal_slot_stride(ROWS);       // bytes
    constexpr int STEPS = ROWS / 2;                    // k-steps (4 rows = 2 detections) per image
    constexpr bool HAS_POSE = CHAIN != CHAIN_FREE;
    using D2 = __attribute__((ext_vector_type(2))) double;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds_image[];
    if (a.stop && *a.stop) return;
    const int lane = threadIdx.x;
    const T *cam_slab = static_cast<const T *>(a.cam_slab);
    const T *pose_slab = static_cast<const T *>(a.pose_slab);
    const T *points = static_cast<const T *>(a.points);

    // ---- per-lane constants: operand addresses and the entries this lane's accumulator registers stand for -------
    int rd_off[NW];   // byte offset of this lane's operand of window w at k-step 0
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int s = window_slot<CHAIN, PASS>(w, lane & 15);
        rd_off[w] = (s < 0 ? 0 : s) * KS + (lane >> 4) * 8;
    }
    // D layout of v_mfma_f64_16x16x4_f64: register r of lane l = D[(l >> 4) + 4 r][l & 15].
    // Where a finished register goes.  Its two columns are (group, offset) pairs — groups in parameter-string order:
    // 0 intrinsics, 1 extrinsics, 2 pose, 3 point, 4 = the residual column — and the destination is
    //     region + 8 (row(gR) + oR ld + col(gC) + oC)          (gR, oR) <= (gC, oC); region / ld by the layout (NormalArgs)
    //     g + 8 (base[g] + o)                                  one of the two is the residual
    //     cost                                                 both are
    // with base[] the first parameter-string column of each group for the current run.  Everything static is folded into one
    // descriptor per register (offsets | which table rows to read | owned | pose) and the run-dependent part is looked up
    // across lanes (ds_bpermute) in a table the flush refreshes with v_writelane.  No selects: hipcc turned the select
    // chains of the first version into ~10 exec-mask branches per register (800 instructions and 80 branches per flush,
    // 12 us of the 92 at N = 1e6).
    // (entry_descriptor above; A, B and C are addressed with 32-bit offsets in doubles: the host checks the sizes)
    const int tg = a.trail_group;
    int ent[NM][4];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) ent[m][r] = entry_descriptor<CHAIN, PASS>(m, lane, r, tg);
    // lanes 0-4: the five output pointers (A, B, C, g, cost)
    const int64_t out_shift = (a.sel && *a.sel) ? 8 * a.alt : 0;   // bytes: which of the two packed states receives this build
    const uint64_t out_ptr = (lane == 1 ? (uint64_t)a.HB : lane == 2 ? (uint64_t)a.HC : lane == 3 ? (uint64_t)a.g : lane == 4 ? (uint64_t)a.cost : (uint64_t)a.H) + (uint64_t)out_shift;
    const int ptr_lo = (int)(uint32_t)out_ptr, ptr_hi = (int)(uint32_t)(out_ptr >> 32);
    int base_tab = lane == 12 ? a.ldA : lane == 13 ? a.ldB : lane == 14 ? a.tb : 0;
    d4v acc[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m] = d4v{0.0, 0.0, 0.0, 0.0};

    // the run the accumulators belong to (wave-uniform): SHARED (cam, image), CAMKEY (cam, key), IMGKEY (image, key)
    int run_a = -1, run_b = -1;

    // Add the finished entries to H / g / cost.  `everything` = false (PASS_SHARED, image changed but not the
    // camera): only entries that involve a pose column are flushed and cleared.
    int64_t seg = a.part ? a.seg_base[blockIdx.x] : 0;   // deterministic mode: the slot of this wave's next flush
    auto flush = [&](const bool everything) {
        if (run_a < 0) return;
        if (a.part) {   // deterministic mode: the whole tile, raw, to its slot (coalesced 512-byte stores); every flush is a complete one
            double *slot = a.part + seg * (NM * 256) + lane;
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    slot[(m * 4 + r) * 64] = acc[m][r];
                    acc[m][r] = 0.0;
                }
            ++seg;
            return;
        }
        if (a.debug & 128) {   // profiling: clear the finished entries, no look-ups, no atomics
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][r] = (everything || (ent[m][r] & (1 << 28))) ? 0.0 : acc[m][r];
            return;
        }
        const int cam = PASS == PASS_IMGKEY ? 0 : run_a;
        const int img = PASS == PASS_SHARED ? run_b : run_a;     // only used where pose columns occur
        const int key = run_b;                                   // only used in the point passes
        // first parameter-string column of each group for this run (wave-uniform) -> the table rows described above
        const int base[4] = {9 * cam, (int)a.extr_off + 6 * cam, (int)a.pose_off + 6 * img, (int)a.point_off + 3 * key};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // the table holds offsets in DOUBLES (round 5; bytes before: a region ended at 4 GiB = 2^29 doubles — now 2^32 doubles = 32 GiB,
            // checked by the host): unsigned 32-bit arithmetic, widened and shifted when the address is formed
            const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane(base[g]);
            const int rowA = (int)((uint32_t)a.ldA * b), rowB = (int)((uint32_t)a.ldB * b);
            const int col = (int)(b - (g == tg ? (uint32_t)a.trail_off : 0u)), gi = (int)b;
            asm("v_writelane_b32 %0, %1, %2" : "+v"(base_tab) : "s"(rowA), "n"(g));
            asm("v_writelane_b32 %0, %1, %2" : "+v"(base_tab) : "s"(rowB), "n"(4 + g));
            asm("v_writelane_b32 %0, %1, %2" : "+v"(base_tab) : "s"(col), "n"(16 + g));
            asm("v_writelane_b32 %0, %1, %2" : "+v"(base_tab) : "s"(gi), "n"(24 + g));
        }
        {
            const int ent_idx = tg == 2 ? img : key;   // entity of the trailing group in this run
            const int rowC = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a.tb * a.tb) * (uint32_t)ent_idx));
            asm("v_writelane_b32 %0, %1, %2" : "+v"(base_tab) : "s"(rowC), "n"(8));
        }
        // all destinations first (5 lane look-ups per register, no branch in between, so their latencies overlap) ...
        uint64_t dst[NM][4];
        int dsc[NM][4];
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int d = ent[m][r];
                asm volatile("" : "+v"(d));   // opaque: nothing of the address arithmetic is to be hoisted out of the tile loop
                const uint32_t ld = (uint32_t)__builtin_amdgcn_ds_bpermute((d >> 16) & (31 << 2), base_tab);
                const uint32_t off = (uint32_t)__builtin_amdgcn_ds_bpermute((d >> 6) & (31 << 2), base_tab) +
                                     (uint32_t)__builtin_amdgcn_ds_bpermute((d >> 11) & (31 << 2), base_tab) +
                                     ((uint32_t)(d & 15) * ld + (uint32_t)((d >> 4) & 15));
                const int psel = (d >> 21) & (7 << 2);
                const uint64_t pb = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(psel, ptr_hi) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(psel, ptr_lo);
                dst[m][r] = pb + ((uint64_t)off << 3);
                dsc[m][r] = d;
            }
        // ... then one predicated atomic per register
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = dsc[m][r];
                const bool now = everything || (d & (1 << 28));
                const double s = acc[m][r];
                acc[m][r] = now ? 0.0 : s;
                if (now && (d & (1 << 27)) && s != 0.0 && !(a.debug & 2))
                    __builtin_amdgcn_global_atomic_fadd_f64(reinterpret_cast<__attribute__((address_space(1))) double *>(dst[m][r]), s);
            }
    };

    const int64_t tile0 = (int64_t)blockIdx.x * a.tiles_per_wave;
    const int64_t tile1 = min(tile0 + (int64_t)a.tiles_per_wave, a.n_tiles);
    // The index words and the measurement of the NEXT tile are requested before this tile's arithmetic and MFMA loop
    // start: that first level of the dependent load chain (detection -> slab / point addresses) is what a wave with one
    // or two neighbours on its SIMD otherwise sits out (SQ_WAIT_ANY was 39 % of the wave cycles, profiles/r02).
    auto det_index = [&](const int64_t tile) -> int64_t {
        const int64_t i = tile * 64 + lane;
        const int64_t is = i < a.n ? i : a.n - 1;
        return a.order ? a.order[is] : is;
    };
    DetWords nxt_w{};
    double2v nxt_m{};
    if (tile0 < tile1) {
        const int64_t ic = det_index(tile0);
        nxt_w = load_words(a.tab, ic);
        nxt_m = load_uv(a.tab, ic);
    }
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        const int64_t i = tile * 64 + lane;
        const bool valid = i < a.n;
        int c, im, k;
        decode_words(a.tab, nxt_w, c, im, k);
        if (!HAS_POSE) im = 0;
        const double2v m = nxt_m;
        if (tile + 1 < tile1) {
            const int64_t icn = det_index(tile + 1);
            nxt_w = load_words(a.tab, icn);
            nxt_m = load_uv(a.tab, icn);
        }
        asm volatile("" ::: "memory");   // keep the requests up here
        T u, v;
        T J[P2];
        if (a.debug & 16) {   // profiling: no evaluation.  The stand-in values hang on THIS tile's measurement: loop-invariant ones
            // ((double)(lane + j), round 2) were hoisted out of the tile loop by hipcc and kept 84 VGPRs alive across it —
            // the reason the kernel sat at 255 VGPRs + scratch
            u = m.x; v = m.y;
#pragma unroll
            for (int j = 0; j < P2; ++j) J[j] = m.x + (double)j;
        } else {
            // Slabs: when the whole tile refers to one camera (image) its slab is fetched with ONE coalesced load and read
            // through v_readlane (LaneSlab) instead of 48 (39) per-lane loads of the same address — with one-wave
            // workgroups and ~2 waves per SIMD the latency of those loads is what this kernel waits for.  Tiles that mix
            // cameras / images (run boundaries) take the per-lane loads.
            const T X0 = points[3 * k], X1 = points[3 * k + 1], X2 = points[3 * k + 2];
            const int c0 = __builtin_amdgcn_readfirstlane(c), im0 = __builtin_amdgcn_readfirstlane(im);
            const bool cam_uni = __all(c == c0), img_uni = !HAS_POSE || __all(im == im0);
            const T *csp = cam_slab + c * CAM_STRIDE, *psp = pose_slab + im * POSE_STRIDE;
            if (PASS == PASS_SHARED && cam_uni && img_uni) {
                const LaneSlab lc{cam_slab[c0 * CAM_STRIDE + min(lane, CAM_STRIDE - 1)]};
                const LaneSlab lp{HAS_POSE ? pose_slab[im0 * POSE_STRIDE + min(lane, POSE_STRIDE - 1)] : 0.0};
                eval_detection<CHAIN, T, true>(lc, lp, X0, X1, X2, u, v, J);
            } else if (PASS == PASS_CAMKEY && cam_uni) {
                const LaneSlab lc{cam_slab[c0 * CAM_STRIDE + min(lane, CAM_STRIDE - 1)]};
                eval_detection<CHAIN, T, true>(lc, psp, X0, X1, X2, u, v, J);
            } else if (PASS == PASS_IMGKEY && img_uni) {
                const LaneSlab lp{pose_slab[im0 * POSE_STRIDE + min(lane, POSE_STRIDE - 1)]};
                eval_detection<CHAIN, T, true>(csp, lp, X0, X1, X2, u, v, J);
            } else {
                eval_detection<CHAIN, T, true>(csp, psp, X0, X1, X2, u, v, J);
            }
        }
        const double r0 = u - m.x, r1 = v - m.y;
        // The requests for the next tile (issued above) have had the whole evaluation to arrive; consume them HERE, before
        // the MFMA phase can issue flush atomics.  vmcnt retires in order on gfx9: waiting for these loads at the top of the
        // next tile would also wait for every atomic issued after them (12 us of the kernel at N = 1e6).
        asm volatile("" ::"v"(nxt_w.w0), "v"(nxt_w.w1), "v"(nxt_w.w2), "v"(nxt_m.x), "v"(nxt_m.y));

        // run boundaries of this tile: bit d set = detection d starts a new run
        const int ka = PASS == PASS_IMGKEY ? im : c, kb = PASS == PASS_SHARED ? im : k;
        const int pa = __shfl_up(ka, 1), pb = __shfl_up(kb, 1);
        const bool starts = valid && (lane == 0 ? (ka != run_a || kb != run_b) : (ka != pa || kb != pb));
        const uint64_t bnd = __ballot(starts);

#pragma unroll
        for (int h = 0; h < 64 / ROWS; ++h) {
            // ---- this pass's detections -> LDS image (lanes past the end of the table write zero rows: they add nothing) -----------
            if (!(a.debug & 64) && (ROWS == 64 || (lane >> 5) == h)) {   // debug 64: profiling, no LDS image
                unsigned char *dst = lds_image + (lane & (ROWS - 1)) * 16;
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
                    const int lc = slot_col<CHAIN, PASS>(s);
                    double *q = reinterpret_cast<double *>(dst + s * KS);   // 8-byte aligned (odd slot stride): one ds_write2_b64
                    q[0] = lc == NORMAL_R ? r0 : J[lc];
                    q[1] = lc == NORMAL_R ? r1 : J[P + lc];
                }
                if (!valid) {   // only the table's last tile has such lanes: their rows are overwritten with zeros
#pragma unroll
                    for (int s = 0; s < NSLOT; ++s) {
                        double *q = reinterpret_cast<double *>(dst + s * KS);
                        q[0] = 0.0;
                        q[1] = 0.0;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            const int d_base = h * ROWS;
            const uint64_t bits = (a.debug & 32) ? 0ull : ROWS == 64 ? bnd : (bnd >> d_base) & 0xffffffffull;   // debug 32: profiling, ignore run boundaries
            auto operand = [&](const int w, const int s) { return *reinterpret_cast<const double *>(lds_image + rd_off[w] + s * 32); };
            auto run_mfmas = [&](const double (&x)[NW]) {
#pragma unroll
                for (int mm = 0; mm < NM; ++mm)
                    acc[mm] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[mfma_a<CHAIN, PASS>(mm)], x[mfma_b<CHAIN, PASS>(mm)], acc[mm], 0, 0, 0);
            };
            auto new_run = [&](const int d) {   // detection d (tile-relative) starts a run
                const int na = __builtin_amdgcn_readlane(ka, d), nb = __builtin_amdgcn_readlane(kb, d);
                flush(PASS != PASS_SHARED || na != run_a);
                run_a = na;
                run_b = nb;
            };
            // k-steps [s0, s1) of the image, all inside the current run: rolled loop, the operands of step s + 1 are
            // requested before the MFMAs of step s are issued (the MFMAs of one step take 64 cycles each — the LDS latency)
            const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds_image;
            auto run_steps = [&](const int s0, const int s1) {
                if (s0 >= s1) return;
                double x[3][NW];
                // one running byte address per operand window: the reads of a trip are immediate offsets from it and the trip costs NW
                // additions.  (The first version rebuilt every address from the step number: 9 v_add per trip + a clamp of the
                // look-ahead step on the scalar side.)  The look-ahead past the image's last step simply reads the next slot's first
                // bytes / the 64-byte pad behind the image, and those values are never used.
                using LdsD = const __attribute__((address_space(3))) double *;
                uint32_t adr[NW];   // LDS byte addresses
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    adr[w] = lds_base + (uint32_t)rd_off[w] + (uint32_t)s0 * 32u;
                    asm volatile("" : "+v"(adr[w]));   // one register per window (otherwise hipcc keeps base and offset apart: an add per read)
                    x[0][w] = *(LdsD)(uintptr_t)adr[w];
                    x[1][w] = *(LdsD)(uintptr_t)(adr[w] + 32u);
                }
                int s = s0;
                // Order per step: wait for THIS step's operands, request those of the step AFTER THE NEXT, issue the MFMAs: two steps
                // of operands are in flight (with one, the walk ran at ~136 cycles per MFMA instead of 64 — LDS latency under seven
                // waves per CU is longer than the two MFMAs of a step; `normal_debug 8`: the walk cost 55 us of the 90).  hipcc
                // places its s_waitcnt directly before the first use of a loaded register and — left alone — sinks the next requests
                // below the MFMAs; `touch` is an empty asm that uses the operands, so the wait lands before the next requests are
                // issued; sched_barrier(0) keeps the three groups in this order.
                auto touch = [&](const double (&v)[NW]) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) asm volatile("" ::"v"(v[w]));
                };
                auto step = [&](const int cur, const int nxt2, const uint32_t off) {   // consume buffer `cur`, refill it... with step + 3
                    touch(x[cur]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int w = 0; w < NW; ++w) x[nxt2][w] = *(LdsD)(uintptr_t)(adr[w] + off);
                    __builtin_amdgcn_sched_barrier(0);
                    run_mfmas(x[cur]);
                    __builtin_amdgcn_sched_barrier(0);
                };
                for (; s + 2 < s1; s += 3) {   // three steps per trip: the operand buffers rotate without register moves
                    step(0, 2, 64u);
                    step(1, 0, 96u);
                    step(2, 1, 128u);
#pragma unroll
                    for (int w = 0; w < NW; ++w) adr[w] += 96u;
                }
                if (s < s1) run_mfmas(x[0]);
                if (s + 1 < s1) run_mfmas(x[1]);
            };
            if (a.debug & 8) {
                // profiling: no MFMA phase
            } else {
                // Walk the image from boundary to boundary.  `bits` >> 2 s0 = boundary flags of the detections not yet
                // consumed; the step that holds the next flagged detection is handled on its own (flush, or split when the
                // run changes between its two detections), everything before it goes through run_steps.
                int s0 = 0;
                while (s0 < STEPS) {
                    const uint64_t rem = bits >> (2 * s0);
                    if (rem == 0) {
                        run_steps(s0, STEPS);
                        break;
                    }
                    const int nb = __builtin_ctzll(rem);      // detections until the next run start
                    const int sb = s0 + (nb >> 1);            // its k-step
                    run_steps(s0, sb);
                    const int d0 = d_base + 2 * sb;
                    const uint32_t bb = (uint32_t)(bnd >> d0) & 3u;
                    double x[NW];
#pragma unroll
                    for (int w = 0; w < NW; ++w) x[w] = operand(w, sb);
                    if (bb & 1u) new_run(d0);
                    if (bb & 2u) {   // the step straddles a boundary: rows of detection d0, flush, rows of d0 + 1
                        double xa[NW], xb[NW];
#pragma unroll
                        for (int w = 0; w < NW; ++w) { xa[w] = lane < 32 ? x[w] : 0.0; xb[w] = lane < 32 ? 0.0 : x[w]; }
                        run_mfmas(xa);
                        new_run(d0 + 1);
                        run_mfmas(xb);
                    } else {
                        run_mfmas(x);
                    }
                    s0 = sb + 1;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    flush(true);
}

// ---------------------------------------------------------------------------------------------
// PASS_IMGKEY (self chain) as a segmented sum: F[i,k] = sum over the detections of (image i, key k) of J_pose^T J_point.
// ---------------------------------------------------------------------------------------------
// An (image, key) run is only as long as the number of cameras that see the feature (2 .. n_cams), so a tile of 64
// detections holds 5 - 30 runs and the boundary walk of ba_normal_mfma_kernel degenerates: every k-step straddles a
// boundary, the accumulator tile holds 18 useful entries of 256, and each run pays its own flush (132 us at
// N = 1.1e6, more than the shared pass).  Two observations replace it:
//
//  (1) Inside a run the pose and the point are fixed, and both column groups are the SAME 2 x 3 block S = A_x R_e
//      (eval_detection) times run constants:  J_pose = S [Q_r | I],  J_point = S R_p,  Q_r[:, a] = (dR_p / dr_a) X.
//      So the run needs only the symmetric 3 x 3 sum  G = sum_d S_d^T S_d  (6 numbers instead of 18), and
//          F[t_c, x] = (G R_p)[c][x]        F[r_a, x] = (Q_r^T G R_p)[a][x]
//      is finished once per run by the lane that owns the run.  The evaluation shrinks with it: no intrinsic, extrinsic
//      or pose-rotation columns are formed per detection.
//  (2) The segmented sum is itself a matrix product, D[run][c] = sum_d M[run][d] P[d][c], with M the 0/1 membership of
//      the tile's detections in (up to 16) runs and P[d][c] the detection's 6 products S_u[a] S_u[b] + S_v[a] S_v[b].
//      P goes through a wave-private LDS image (column-major, 8 bytes per detection); M needs no storage — lane (i, q)
//      of the A operand compares detection 4 s + q with the first and one-past-last detection of run i.  16 k-steps of
//      one v_mfma_f64_16x16x4_f64 per batch of 16 runs, no boundary handling.
// A tile starts a new run (the parts of a run that spans tiles meet in the atomics), so nothing is carried between
// tiles.  Rounding differs from summing the 18 products directly by a few ulp of |Q_r| |G| |R_p| — far inside the
// run-to-run spread the atomics already have.
constexpr int IK_COLS = 6;                                 // (a, b) pairs of the symmetric 3 x 3, a <= b
constexpr int IK_SLOT = 64 * 8 + 16;                       // bytes per product column: 64 detections + pad (slot c starts on bank 4 c)
constexpr int IK_RUNS = IK_COLS * IK_SLOT;                 // run table: 65 x {first detection, image, key, -}
constexpr int IK_GRAM = IK_RUNS + 65 * 16;                 // finished sums: 64 runs x 6 doubles
constexpr int normal_imgkey_lds_bytes() { return IK_GRAM + 64 * IK_COLS * 8; }

__global__ __launch_bounds__(64, 3) void ba_normal_imgkey_kernel(const NormalArgs a) {
    using T = double;
    constexpr int CHAIN = CHAIN_SELF;
    constexpr int P = chain_P(CHAIN);
    constexpr int P2 = 2 * P;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_image[];
    if (a.stop && *a.stop) return;
    const int lane = threadIdx.x;
    const T *cam_slab = static_cast<const T *>(a.cam_slab);
    const T *pose_slab = static_cast<const T *>(a.pose_slab);
    const T *points = static_cast<const T *>(a.points);
    using I4 = __attribute__((ext_vector_type(4))) int;
    using D2 = __attribute__((ext_vector_type(2))) double;
    I4 *runs = reinterpret_cast<I4 *>(lds_image + IK_RUNS);
    const int64_t out_shift = (a.sel && *a.sel) ? a.alt : 0;   // doubles: which of the two packed states receives this build (NormalArgs::sel)

    const int li = lane & 15, lq = lane >> 4;
    // B operand at k-step s: P[4 s + q][j]; lanes whose column does not exist read a valid slot, result unused
    const int rd = min(li, IK_COLS - 1) * IK_SLOT + lq * 8;

    auto det_index = [&](const int64_t tile) -> int64_t {
        const int64_t i = tile * 64 + lane;
        const int64_t is = i < a.n ? i : a.n - 1;
        return a.order ? a.order[is] : is;
    };
    const int64_t tile0 = (int64_t)blockIdx.x * a.tiles_per_wave;
    const int64_t tile1 = min(tile0 + (int64_t)a.tiles_per_wave, a.n_tiles);
    DetWords nxt_w{};
    if (tile0 < tile1) nxt_w = load_words(a.tab, det_index(tile0));
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        const bool valid = tile * 64 + lane < a.n;
        int c, im, k;
        decode_words(a.tab, nxt_w, c, im, k);
        if (tile + 1 < tile1) nxt_w = load_words(a.tab, det_index(tile + 1));
        asm volatile("" ::: "memory");
        // S = A_x R_e = the pose-translation columns of the detection's block (everything else eval_detection forms is dead code here)
        T Su[3], Sv[3];
        const T X0 = points[3 * k], X1 = points[3 * k + 1], X2 = points[3 * k + 2];
        // the image's slab across the lanes (one coalesced load, read back through v_readlane): usable when the tile holds one image
        const int im0 = __builtin_amdgcn_readfirstlane(im);
        const bool img_uni = __all(im == im0);
        const LaneSlab lp{pose_slab[im0 * POSE_STRIDE + min(lane, POSE_STRIDE - 1)]};
        if (a.debug & 16) {
#pragma unroll
            for (int j = 0; j < 3; ++j) { Su[j] = X0 + (double)j; Sv[j] = X1 - (double)j; }
        } else {
            T u, v;
            T J[P2];
            const T *csp = cam_slab + c * CAM_STRIDE;
            if (img_uni) eval_detection<CHAIN, T, true>(csp, lp, X0, X1, X2, u, v, J);
            else eval_detection<CHAIN, T, true>(csp, pose_slab + im * POSE_STRIDE, X0, X1, X2, u, v, J);
#pragma unroll
            for (int j = 0; j < 3; ++j) { Su[j] = J[18 + j]; Sv[j] = J[P + 18 + j]; }
        }
        asm volatile("" ::"v"(nxt_w.w0), "v"(nxt_w.w1), "v"(nxt_w.w2));   // consume the prefetch before any atomic is issued (ba_normal_mfma_kernel)

        // runs of this tile: lane 0 always starts one
        const int pi = __shfl_up(im, 1), pk = __shfl_up(k, 1);
        const bool starts = lane == 0 || (valid && (im != pi || k != pk));
        const uint64_t bnd = __ballot(starts);
        const int n_runs = __popcll(bnd);
        const int rl = __popcll(bnd & ((2ull << lane) - 1ull)) - 1;     // local run of this lane's detection

        // ---- LDS: product columns and the run table ---------------------------------------------------------------------
        if (!(a.debug & 64)) {
            unsigned char *dst = lds_image + lane * 8;
            int cc = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = i; j < 3; ++j, ++cc)
                    *reinterpret_cast<double *>(dst + cc * IK_SLOT) = valid ? Su[i] * Su[j] + Sv[i] * Sv[j] : 0.0;
        }
        runs[lane + 1] = I4{64, 0, 0, 0};                               // sentinel: runs past the last one are empty
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (starts) runs[rl] = I4{lane, im, k, 0};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        const bool owner = lane < n_runs;
        const I4 mine = runs[owner ? lane : 0];   // lane = run: first detection, image, key

        if (!(a.debug & 8)) {
            for (int b0 = 0; b0 < n_runs; b0 += 16) {
                const int lo = runs[b0 + li].x, hi = runs[b0 + li + 1].x;        // detections [lo, hi) form run b0 + li
                double bx[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) bx[s] = *reinterpret_cast<const double *>(lds_image + rd + s * 32);
                d4v acc{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int d = 4 * s + lq;
                    const double member = (d >= lo && d < hi) ? 1.0 : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(member, bx[s], acc, 0, 0, 0);
                }
                // register r of lane (column li, q) = run b0 + q + 4 r
                if (li < IK_COLS) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<double *>(lds_image + IK_GRAM + ((b0 + lq + 4 * r) * IK_COLS + li) * 8) = acc[r];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- lane = run: F = [Q_r | I]^T G R_p -----------------------------------------------------------------------------------
        // The run's point is the one its first detection already holds (a lane shuffle); its pose slab is the tile's LaneSlab
        // when the tile holds one image (no memory access at all), else 36 gathered loads.  Atomics even for a run strictly
        // inside the tile, which owns its block of H outright: plain 8-byte stores into the freshly zeroed H were measured
        // 3 - 4 x slower than the atomics (188 us, 218 us with nontemporal stores, against 57 us for this kernel, profiles/r02/sweeps.md).
        auto finish = [&](const auto ps) {
            const T Xr0 = __shfl(X0, mine.x), Xr1 = __shfl(X1, mine.x), Xr2 = __shfl(X2, mine.x);
            T Rp[9], Qr[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) Rp[j] = ps[POSE_R + j];
#pragma unroll
            for (int aa = 0; aa < 3; ++aa)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
                    Qr[cc * 3 + aa] = ps[POSE_DR + aa * 9 + cc * 3 + 0] * Xr0 + ps[POSE_DR + aa * 9 + cc * 3 + 1] * Xr1 + ps[POSE_DR + aa * 9 + cc * 3 + 2] * Xr2;
            const D2 *gp = reinterpret_cast<const D2 *>(lds_image + IK_GRAM + (owner ? lane : 0) * IK_COLS * 8);
            const D2 g01 = gp[0], g23 = gp[1], g45 = gp[2];
            const double G[3][3] = {{g01.x, g01.y, g23.x}, {g01.y, g23.y, g45.x}, {g23.x, g45.x, g45.y}};
            double GR[3][3];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
#pragma unroll
                for (int x = 0; x < 3; ++x) GR[cc][x] = G[cc][0] * Rp[x] + G[cc][1] * Rp[3 + x] + G[cc][2] * Rp[6 + x];
            const bool live = owner && !(a.debug & 2);
            // pose rows x point columns: inside B when the points are the trailing group (blocked layout), else inside the dense H
            const bool in_b = a.trail_group == 3;
            const int64_t ldh = in_b ? (int64_t)a.ldB : a.n_params;
            double *Hrow = (in_b ? a.HB : a.H) + out_shift + ((int64_t)a.pose_off + 6 * mine.y) * ldh + (in_b ? 0 : a.point_off) + 3 * mine.z;
            auto emit = [&](double *ptr, const double val) {
                if (live && val != 0.0) unsafeAtomicAdd(ptr, val);
            };
#pragma unroll
            for (int aa = 0; aa < 3; ++aa)
#pragma unroll
                for (int x = 0; x < 3; ++x)
                    emit(Hrow + (int64_t)aa * ldh + x, Qr[aa] * GR[0][x] + Qr[3 + aa] * GR[1][x] + Qr[6 + aa] * GR[2][x]);
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
#pragma unroll
                for (int x = 0; x < 3; ++x) emit(Hrow + (int64_t)(3 + cc) * ldh + x, GR[cc][x]);
        };
        if (img_uni) finish(lp);
        else finish(pose_slab + mine.y * POSE_STRIDE);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

}  // namespace pcs
