// ba_chol_persist.hpp — S x = rhs in ONE launch: the dense Cholesky of the reduced camera system (ba_dense_chol.hpp) as a persistent
// kernel whose workgroups hand tiles to each other through HBM instead of through launch boundaries.
//
// Why: the launch-per-block-column form spends 17-22 us per block column, of which the arithmetic on the critical path (factoring a
// 32 x 32 tile) is ~5; the rest is launch ramp, tile loads, the inversion of the diagonal tile and stores (profiles/r03/README.md).
// n = 480 (rig-32) took 283 us, n = 1 680 (rig-32-self) 1.15 ms: 64 % of an LM trial's kernel time.
//
// Form.  32 x 32 tiles of the lower triangle, plus ONE extra block row that carries the right-hand side: the Cholesky factor of
// [[S, b], [b', .]] has y' = (L^-1 b)' as its last row, so the forward substitution is just another row of panel tiles and needs no
// code of its own.  Every tile (i, j) is OWNED by one workgroup for the whole launch and lives in that workgroup's LDS:
//   * left-looking per tile — as soon as block column m is published the owner subtracts L_im L_jm' (FP64 matrix cores);
//   * an off-diagonal owner also keeps a PRIVATE copy of the diagonal tile A_jj and gives it the same updates (the L_jm it needs is
//     the operand it has loaded anyway), so when column j - 1 arrives it can factor the 64 x 32 panel [A_jj; A_ij] at once — one
//     wave, one row per lane, the elimination of the diagonal tile carries the 32 rows below it along in the same instructions.
//     No inverse of the diagonal factor on the critical path, no hand-off of the diagonal tile: ONE hand-off per block column;
//   * a published tile goes, whole, to its own slot of a workspace the host has filled with 0xFF bytes (write-through `sc1` stores,
//     nobody waits for an acknowledgement) and THE DATA IS THE SIGNAL: a consumer reads the slot with `sc1` loads and asks again
//     until no word holds the fill pattern (an all-ones NaN that no arithmetic produces).  No counter, no `s_waitcnt vmcnt(0)` +
//     barrier + atomic on the producer's side, no second round trip on the consumer's: the per-column counter this replaced cost
//     ~2.5 us more per block column (n = 480: 215 -> 157 us; profiles/r04/README.md).  L also goes to S (plain stores: the result);
//   * each workgroup orders its work by urgency: tiles of the NEXT column first (update, factor, publish), the rest of the trailing
//     matrix afterwards, in the shadow of the next column's factorisation.
// Backward substitution L' x = y: distributed over the owners of the diagonal tiles, one wave each.  Owner k keeps
// t_k = y_k - sum_{i > k + 1} L_ik' x_i up to date as the x_i appear (every lane loads the 32 words of x_i itself — uniform
// addresses, so the products need no cross-lane broadcast — and the tile for the next x is already in registers when it arrives).
// The last arrival is the one on the chain, so its work is prepared: W = (L_k+1,k L_kk^-1)' and u = L_kk^-T t_k are ready, and
// x_k = u - W x_k+1 is 32 multiply-adds between "x_k+1 visible" and "x_k stored" (2.3 -> 1.x us per block; the hand-off itself is
// ~1 us).  x is published as 32 data words polled directly, like the tiles.
//
// Safety: every wait has a time limit (CholPersistArgs::timeout_ticks of the 100 MHz wall clock) and watches one abort word;
// a workgroup that gives up sets it and bit 2 (value 4) of *status, and every other workgroup leaves at its next wait — the
// grid always drains.  The host launches at most one workgroup per CU and at most CP_MAX_SLOTS tiles per workgroup; larger systems
// take the launch-per-column path.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ba_dense_chol.hpp"

namespace pcs {

struct CholPersistArgs {
    double *S;              // n x n row-major, row stride ld: lower triangle in, L out (the upper triangle is neither read nor written)
    const double *rhs;      // n
    double *x;              // n: the solution
    // the hand-over workspace, filled with 0xFF bytes by the host (one memset): a word differs from the fill once it is written
    double *tpub;           // nb (nb + 1) / 2 tiles of 32 x 32 (padding included): tile (i, j), j < i <= nb, at i (i - 1) / 2 + j; row nb = the rhs row: y = L^-1 rhs in the tiles' rows 0
    double *xpub;           // nb x 32: x, one block per diagonal owner
    int32_t *flags;         // CP_FLAGS words (= -1 after the fill): the abort word and the "originals are loaded" counter
    int32_t *status;        // |= 2: a pivot was not positive; |= 4: a wait ran out of time (results are not valid)
    int32_t n, ld, nb, slots;   // nb = block columns; slots = tiles per workgroup (LDS is sized for it)
    int64_t timeout_ticks;
    const int32_t *stop;        // optional device word: non-zero = do nothing (a launch queued behind the end of an LM loop, ba_schur.hpp)
#ifdef CP_TRACE
    int64_t *trace;             // developer builds (tools/probes/chol_persist_probe.hip): [workgroup][column][8] wall-clock stamps
#endif
};
#ifdef CP_TRACE
#define CP_STAMP(col, k) do { if (tid == 0 && a.trace) a.trace[((int64_t)wg * (a.nb + 1) + (col)) * 8 + (k)] = (int64_t)wall_clock64(); } while (0)
#else
#define CP_STAMP(col, k) do { } while (0)
#endif

constexpr int CP_ABORT = 0, CP_LOADED = 1, CP_FLAGS = 16;   // flag words (one 64-byte line)
constexpr int CP_LDT = 33;                               // row stride of a resident tile (row-per-lane access: conflict-free)
constexpr int CP_SLOT = 2 * 32 * CP_LDT + 32;            // doubles per slot: the tile, the private diagonal copy (diagonal owner: the inverse), s_k
constexpr int CP_MAX_SLOTS = 8;
constexpr uint64_t CP_FILL = 0xFFFFFFFFFFFFFFFFull;

__host__ __device__ inline int64_t cp_tiles(const int64_t nb) { return nb + nb * (nb + 1) / 2; }   // diagonal + (below-diagonal + rhs row) tiles
__host__ __device__ inline size_t cp_lds_bytes(const int slots) { return sizeof(double) * (2 * 32 * CHOL_LDP + (size_t)slots * CP_SLOT) + 64 * sizeof(int); }

__device__ __forceinline__ double cp_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }        // global_load_dwordx2 sc1
__device__ __forceinline__ void cp_st(double *p, const double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // global_store_dwordx2 sc1
__device__ __forceinline__ int cp_ldi(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The abort word and the clock are looked at every 64th poll of a wait — and at EVERY poll when the time limit is below 10 us (tests set
// 1 us to force the give-up path: the shortest hand-over between two workgroups takes longer than that).
__device__ __forceinline__ bool cp_check_now(const CholPersistArgs &a, const int spins) { return (spins & 63) == 0 || a.timeout_ticks < 1000; }

// lane 0 of the calling wave: spin until flags[word] >= target; false = the launch is being abandoned
__device__ __forceinline__ bool cp_spin(const CholPersistArgs &a, const int word, const int target) {
    if (cp_ldi(a.flags + word) >= target) return true;
    const uint64_t t0 = wall_clock64();
    for (int spins = 1;; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if (cp_ldi(a.flags + word) >= target) return true;
        if (cp_check_now(a, spins) && (cp_ldi(a.flags + CP_ABORT) >= 0 || (int64_t)(wall_clock64() - t0) > a.timeout_ticks)) {
            __hip_atomic_store(a.flags + CP_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(a.status, 4);
            return false;
        }
    }
}
// one wave waits (no barrier); counters start at -1, so "count arrivals" means target - 1
__device__ __forceinline__ bool cp_wait_wave(const CholPersistArgs &a, const int word, const int arrivals, const int lane) {
    int ok = 1;
    if (lane == 0) ok = cp_spin(a, word, arrivals - 1) ? 1 : 0;
    return __builtin_amdgcn_readfirstlane(ok) != 0;
}

// between two polls of a DATA word by one wave: a short sleep; every 64th time the abort word and the clock.  false = abandon the launch
__device__ __forceinline__ bool cp_poll_again(const CholPersistArgs &a, const uint64_t t0, const int spins, const int lane) {
    __builtin_amdgcn_s_sleep(1);
    if (!cp_check_now(a, spins)) return true;
    int give_up = 0;
    if (lane == 0 && (cp_ldi(a.flags + CP_ABORT) >= 0 || (int64_t)(wall_clock64() - t0) > a.timeout_ticks)) {
        __hip_atomic_store(a.flags + CP_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicOr(a.status, 4);
        give_up = 1;
    }
    return __builtin_amdgcn_readfirstlane(give_up) == 0;
}

// entry (gr, gc) of the ORIGINAL matrix, symmetric, with the identity padding of a ragged last block
__device__ __forceinline__ double cp_orig(const CholPersistArgs &a, int gr, int gc) {
    if (gr < gc) { const int t = gr; gr = gc; gc = t; }
    if (gr < a.n) return a.S[(int64_t)gr * a.ld + gc];
    return gr == gc ? 1.0 : 0.0;
}

// tile t of the enumeration -> (i, j): the nb diagonal tiles first (so that they land on different workgroups), then column by
// column the tiles below the diagonal and the column's rhs tile (i = nb)
__device__ __forceinline__ void cp_decode(const int nb, int t, int &i, int &j) {
    if (t < nb) { i = j = t; return; }
    t -= nb;
    for (j = 0; j < nb; ++j) {
        const int cnt = nb - j;   // rows j + 1 .. nb - 1 and the rhs row
        if (t < cnt) { i = j + 1 + t; return; }
        t -= cnt;
    }
    i = -1;
    j = 1 << 30;
}

// this thread's four entries (e = tid + 256 q -> row e >> 5, column e & 31: whole 256-byte rows per half wave) of the PUBLISHED tile
// (i, m), i > m, requested with sc1 loads from the tile's slot; i == nb: the rhs row's tile (y_m' in row 0, zeros below — a tile like
// any other: four loads, no branch).  The caller checks the values against the fill and asks again until none is left (cp_fetch_polled).
__device__ __forceinline__ void cp_fetch(const CholPersistArgs &a, double (&v)[4], const int i, const int m, const int tid) {
    const double *src = a.tpub + ((int64_t)i * (i - 1) / 2 + m) * 1024 + tid;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = cp_ld(src + 256 * q);
}
__device__ __forceinline__ bool cp_is_fill(const double (&v)[4]) {
    bool f = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) f = f || (__builtin_bit_cast(uint64_t, v[q]) == CP_FILL);
    return f;
}
// Fetch the operand tile(s) of an update until no value is the fill any more — the producer's stores are the signal.
// All threads call it; false = the launch is being abandoned.
__device__ __forceinline__ bool cp_fetch_polled(const CholPersistArgs &a, double (&vp)[4], double (&vq)[4], const int i, const int j, const int m, const int tid) {
    const uint64_t t0 = wall_clock64();
    for (int spins = 1;; ++spins) {
        cp_fetch(a, vp, i, m, tid);
        if (i != j) cp_fetch(a, vq, j, m, tid);
        const bool pending = cp_is_fill(vp) || (i != j && cp_is_fill(vq));
        if (!__syncthreads_or(pending)) return true;
        __builtin_amdgcn_s_sleep(1);
        if (cp_check_now(a, spins)) {
            int give_up = 0;
            if (tid == 0 && (cp_ldi(a.flags + CP_ABORT) >= 0 || (int64_t)(wall_clock64() - t0) > a.timeout_ticks)) {
                __hip_atomic_store(a.flags + CP_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(a.status, 4);
                give_up = 1;
            }
            if (__syncthreads_or(give_up)) return false;
        }
    }
}
__device__ __forceinline__ void cp_park(double *dst, const double (&v)[4], const int tid) {   // -> MFMA operand buffer (row stride CHOL_LDP)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q;
        dst[(e >> 5) * CHOL_LDP + (e & 31)] = v[q];
    }
}

// tile (i, j) -= L_im L_jm' and (off-diagonal owners) private A_jj -= L_jm L_jm' from the operand tiles parked in P / Q; wave w owns the
// 16 x 16 quadrant (16 (w >> 1), 16 (w & 1)) of both resident tiles
__device__ __forceinline__ void cp_apply(const double *P, const double *Q, double *Town, double *Td, const bool diag, const int lane, const int wave) {
    const int i0 = 16 * (wave >> 1), j0 = 16 * (wave & 1);
    const int qr = i0 + (lane >> 4), qc = j0 + (lane & 15);
    {
        const chol_d4 u = chol_quadrant_xyT<CHOL_LDP, CHOL_LDP>(P, diag ? P : Q, i0, j0, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) Town[(qr + 4 * r) * CP_LDT + qc] -= u[r];
    }
    if (!diag) {
        const chol_d4 v = chol_quadrant_xyT<CHOL_LDP, CHOL_LDP>(Q, Q, i0, j0, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) Td[(qr + 4 * r) * CP_LDT + qc] -= v[r];
    }
}

// 1 / sqrt(p) on the pivot chain: hardware estimate y (~2^-26) + ONE third-order step, y (1 + h / 2 + 3 h^2 / 8) with h = 1 - p y^2
// (error ~ h^3: full double precision) — four dependent operations where two Newton steps are six.
__device__ __forceinline__ double cp_rsqrt(const double p) {
    const double y = __builtin_amdgcn_rsq(p);
    const double h = __builtin_fma(-(p * y), y, 1.0);
    return __builtin_fma(y * h, __builtin_fma(h, 0.375, 0.5), y);
}

#ifdef CP_TRACE
__device__ int64_t cp_dbg[4 * 16];
#define CP_DBG(k) do { if (lane == 0 && blockIdx.x == gridDim.x - 1) cp_dbg[wave * 16 + (k)] = (int64_t)wall_clock64(); } while (0)   // developer builds: the last workgroup's last panel
__device__ int64_t cp_tile_dbg[8 * 8];   // developer builds: workgroup 100's trailing updates with column 3: five stamps per tile
#define CP_TILE(k) do { if (tid == 0 && blockIdx.x == 100 && m == 3 && !urgent && s < 8) cp_tile_dbg[s * 8 + (k)] = (int64_t)wall_clock64(); } while (0)
#else
#define CP_DBG(k) do { } while (0)
#define CP_TILE(k) do { } while (0)
#endif

// The 64 x 32 panel [D; X] — D = the (private copy of the) diagonal tile, X = the tile below it — factored by the FOUR waves of the
// workgroup: lane r of every wave is row r of the panel (lanes 0-31 D, lanes 32-63 X; the elimination of D carries X along, X ends as
// X L^-T without any inverse), wave w holds the panel's columns 8 w .. 8 w + 7 in registers.  Wave w factors ITS eight columns (pivot
// and column entries from lane j by v_readlane) and parks each finished column in LDS as `Sp[column][row]`; the waves behind it
// subtract that column's rank-1 term from their own columns — row multiplier `Sp[j][lane]`, column multiplier `Sp[j][c]` as a
// broadcast read — COLUMN BY COLUMN AS THEY APPEAR: a finished column is announced by an LDS word (`*seq` = columns finished since
// the kernel started; LDS serves one wave's requests in order, so the word follows the column), and when wave w - 1 stores its
// last column wave w has one rank-1 term left before its own first pivot.  (With a barrier and a block update between the four
// sub-panels the waves behind waited 0.6 us per sub-panel for updates that could have been done while the columns were
// produced: 4.3 -> ~2.9 us per panel, tools/probes/chol_persist_probe.hip's per-wave stamps.)
// `Sp`: 32 x 64 doubles.  `done`: the panels this workgroup has factored so far (every wave counts them itself).
// D == X's tile for the diagonal owner (diag = true): lanes 32-63 shadow lanes 0-31, L goes back to the tile with zeros above the
// diagonal, the reciprocal pivots to `ild` (32 doubles, for the inversion that follows later).
// All four waves call it, between two workgroup barriers of the caller's; returns false in the wave that met a non-positive pivot.
template <int W>   // W = the calling wave: its lanes' indices are compile-time constants
__device__ __forceinline__ bool cp_panel_wave(const double *Dt, double *Xt, const bool diag, double *ild, double *Sp, int *seq, const int base, const int lane) {
    const int r = lane & 31;
    const bool low = lane >= 32;
    const double *src = ((low && !diag) ? Xt : Dt) + r * CP_LDT + 8 * W;
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = src[q];
    bool ok = true;
    double my_il = 0.0;
    const int wave = W;
    CP_DBG(0);
    // the columns of the waves before this one, as they appear: the word and the column are requested together (LDS answers in
    // order, so a column read behind a word that says "stored" is the stored one) — one LDS round trip per column
    for (int j = 0; j < 8 * W; ++j) {
        double mrow, bc[8];
        for (;;) {
            const int seen = __hip_atomic_load(seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // program order only
            mrow = Sp[j * 64 + lane];
#pragma unroll
            for (int q = 0; q < 8; ++q) bc[q] = Sp[j * 64 + 8 * W + q];   // wave-uniform address: broadcast reads
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (seen >= base + j + 1) break;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] -= mrow * bc[q];
    }
    CP_DBG(1);
    // The pivots are one dependent chain (broadcast -> 1/sqrt -> scale -> the NEXT column's update -> broadcast ...): the next pivot's
    // reciprocal root is started as soon as its column has its update; the other columns' updates and the store fill its latency.
    // (Left to itself the scheduler sinks every update to just before its column's pivot: jl dependent multiply-adds on the chain.)
    double p = lane_bcast(v[0], 8 * W);
    double il = cp_rsqrt(p);
#pragma unroll
    for (int jl = 0; jl < 8; ++jl) {
        const int j = 8 * W + jl;   // the pivot's row of D lives in lane j
        ok = ok && (p > 0.0);
        my_il = (r == j) ? il : my_il;
        v[jl] *= il;   // lane j holds p itself: p / sqrt(p)
        if (jl + 1 < 8) {
            v[jl + 1] -= v[jl] * lane_bcast(v[jl], j + 1);   // L[c][j] lives in lane c
            p = lane_bcast(v[jl + 1], j + 1);
            il = cp_rsqrt(p);
        }
        if (W < 3) {
            Sp[j * 64 + lane] = v[jl];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) __hip_atomic_store(seq, base + j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int q = jl + 2; q < 8; ++q) v[q] -= v[jl] * lane_bcast(v[jl], 8 * W + q);
        __builtin_amdgcn_sched_barrier(0);
    }
    CP_DBG(2);
    if (diag) {
        if (!low) {
#pragma unroll
            for (int q = 0; q < 8; ++q) Xt[r * CP_LDT + 8 * W + q] = (8 * W + q <= r) ? v[q] : 0.0;
            if ((r >> 3) == W) ild[r] = my_il;
        }
    } else if (low) {
#pragma unroll
        for (int q = 0; q < 8; ++q) Xt[r * CP_LDT + 8 * W + q] = v[q];
    }
    CP_DBG(3);
    return ok;
}
__device__ __forceinline__ bool cp_panel4(const double *Dt, double *Xt, const bool diag, double *ild, double *Sp, int *seq, const int done,
                                          const int lane, const int wave) {
    const int base = done * 32;
    switch (wave) {
    case 0: return cp_panel_wave<0>(Dt, Xt, diag, ild, Sp, seq, base, lane);
    case 1: return cp_panel_wave<1>(Dt, Xt, diag, ild, Sp, seq, base, lane);
    case 2: return cp_panel_wave<2>(Dt, Xt, diag, ild, Sp, seq, base, lane);
    default: return cp_panel_wave<3>(Dt, Xt, diag, ild, Sp, seq, base, lane);
    }
}

// Li = L^-1 (lower, zeros stored above the diagonal) by forward substitution, one column per lane, column-oriented: once x[m] is known
// every later equation gets its term (31 - m independent FMAs; ba_dense_chol.hpp's factor_and_invert_tile, second half)
__device__ __forceinline__ void cp_invert_diag(const double *D, double *Li, const double *ild, const int lane) {
    if (lane >= 32) return;
    const int c = lane;
    double t[32];
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) t[rr] = (rr == c) ? 1.0 : 0.0;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        const double xm = t[m] * ild[m];   // 0 for m < c
        Li[m * CP_LDT + c] = xm;
#pragma unroll
        for (int rr = m + 1; rr < 32; ++rr) t[rr] -= D[rr * CP_LDT + m] * xm;
    }
}

__global__ __launch_bounds__(256) void chol_persist_kernel(const CholPersistArgs a) {
    extern __shared__ double cp_sm[];
    if (a.stop && *a.stop) return;   // every workgroup reads the same word: all leave, or none
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = a.nb, G = gridDim.x, wg = blockIdx.x;
    double *P = cp_sm, *Q = P + 32 * CHOL_LDP, *slot0 = Q + 32 * CHOL_LDP;
    double *Sp = P;   // the panel's column exchange (32 x 64 doubles) shares the operand buffers P and Q: never in use at the same time
    int *meta = reinterpret_cast<int *>(slot0 + (size_t)a.slots * CP_SLOT);   // [0..7] packed tile of the slot, [32] the panels' column count
    int *seq = meta + 32;
    int panels = 0;
    auto Town = [&](const int s) { return slot0 + (size_t)s * CP_SLOT; };
    auto Td = [&](const int s) { return slot0 + (size_t)s * CP_SLOT + 32 * CP_LDT; };
    auto Sv = [&](const int s) { return slot0 + (size_t)s * CP_SLOT + 2 * 32 * CP_LDT; };

    // The tiles of this workgroup: slot s holds tile wg + s G of the enumeration.  Workgroup w < nb owns the diagonal tile w in slot 0
    // (the host launches at least nb workgroups).  Decoded once by one thread, packed row << 8 | column (0xFFFF = empty): the per-column
    // scans read eight scalars, a slot loop's look-up is one uniform LDS read.
    if (tid < CP_MAX_SLOTS) {
        int i = -1, j = -1;
        if (tid < a.slots) cp_decode(nb, wg + tid * G, i, j);
        meta[tid] = i < 0 ? 0xFFFF : (i << 8 | j);
    }
    if (tid == 0) *seq = 0;
    __syncthreads();
    int pk[CP_MAX_SLOTS], max_j = -1;
#pragma unroll
    for (int s = 0; s < CP_MAX_SLOTS; ++s) {
        pk[s] = __builtin_amdgcn_readfirstlane(meta[s]);
        if (pk[s] != 0xFFFF) max_j = (pk[s] & 255) > max_j ? (pk[s] & 255) : max_j;
    }
    auto tile = [&](const int s, int &i, int &j) {   // one uniform LDS read
        const int v = __builtin_amdgcn_readfirstlane(meta[s]);
        i = v == 0xFFFF ? -1 : v >> 8;
        j = v == 0xFFFF ? (1 << 30) : (v & 255);
    };
    const bool has_diag = wg < nb;   // tile (wg, wg), slot 0

    // ---- the tiles, from the original matrix ---------------------------------------------------------------------------------------------
    for (int s = 0; s < a.slots; ++s) {
        int i, j;
        tile(s, i, j);
        if (i < 0) continue;
        double *To = Town(s), *Tp = Td(s);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, r = e >> 5, c = e & 31;
            double v;
            if (i == nb) v = (r == 0 && j * 32 + c < a.n) ? a.rhs[j * 32 + c] : 0.0;
            else v = cp_orig(a, i * 32 + r, j * 32 + c);
            To[r * CP_LDT + c] = v;
            if (i != j) Tp[r * CP_LDT + c] = cp_orig(a, j * 32 + r, j * 32 + c);
        }
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.flags + CP_LOADED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the originals of S are in LDS: diagonal tiles of S may be overwritten once every workgroup has said so

    // Block column m applied to this workgroup's tiles: fetch (sc1) -> park -> products, tile by tile, TWO tiles' operands in flight
    // (register sets A and B alternate: a fetch is a 1-2 us round trip; park + products + two barriers measure ~1.3 us per tile,
    // 0.43 us of it the sixteen matrix instructions).  With five or six tiles per workgroup — the first columns of n = 1 680 — the
    // trailing update, not the panel chain, sets the pace: ~10.5 us per block column there, 7.3 us later.
    // urgent = true: the tiles of column m + 1 below the diagonal (somebody waits for them); false: every later column AND the diagonal
    // tile of column m + 1 — nobody waits for a diagonal tile (the owners of the tiles below it factor their own copies), its factor
    // and inverse are needed by the backward sweep only.
    auto apply_column = [&](const int m, const bool urgent) -> bool {
        auto next_slot = [&](int s) {
            for (; s < a.slots; ++s) {
                int i, j;
                tile(s, i, j);
                if (i >= 0 && (urgent ? (j == m + 1 && i != j) : (j >= m + 2 || (j == m + 1 && i == j)))) break;
            }
            return s;
        };
        double vpa[4], vqa[4], vpb[4], vqb[4];
        auto request = [&](double (&vp)[4], double (&vq)[4], const int s) {
            if (s >= a.slots) return;
            int i, j;
            tile(s, i, j);
            cp_fetch(a, vp, i, m, tid);
            if (i != j) cp_fetch(a, vq, j, m, tid);
        };
        // consume the set that holds tile s; `sn2` = the tile two further on, requested into the same set once this one is parked
        auto consume = [&](double (&vp)[4], double (&vq)[4], const int s, const int sn2) -> bool {
            int i, j;
            tile(s, i, j);
            CP_TILE(0);
            if (__syncthreads_or(cp_is_fill(vp) || (i != j && cp_is_fill(vq))) && !cp_fetch_polled(a, vp, vq, i, j, m, tid)) return false;
            CP_TILE(1);
            cp_park(P, vp, tid);
            if (i != j) cp_park(Q, vq, tid);
            __syncthreads();
            CP_TILE(2);
            request(vp, vq, sn2);
            cp_apply(P, Q, Town(s), Td(s), i == j, lane, wave);
            CP_TILE(3);
            __syncthreads();
            CP_TILE(4);
            return true;
        };
        int s0 = next_slot(0), s1 = next_slot(s0 + 1);
        request(vpa, vqa, s0);
        request(vpb, vqb, s1);
        while (s0 < a.slots) {
            const int s2 = next_slot(s1 + 1);
            if (!consume(vpa, vqa, s0, s2)) return false;
            if (s1 >= a.slots) break;
            const int s3 = next_slot(s2 + 1);
            if (!consume(vpb, vqb, s1, s3)) return false;
            s0 = s2;
            s1 = s3;
        }
        return true;
    };

    // ---- factorisation: iteration m consumes block column m and produces the tiles of column m + 1 ----------------------------------------
    // One extra round at the end (`last`): this workgroup's own diagonal tile is factored and inverted THERE.  As part of its column's
    // urgent work it made the owner of diagonal tile k late for column k + 1 whenever it also held a tile of that column (23 us
    // iterations: the spikes of 14-22 us per block column in the first columns of n = 1 680).
    for (int m = -1; m <= max_j; ++m) {
        const bool last = m == max_j;
        CP_STAMP(m + 1, 0);
        CP_STAMP(m + 1, 1);
        // urgent: the tiles of column m + 1 below the diagonal
        int ncrit = 0;
#pragma unroll
        for (int s = 0; s < CP_MAX_SLOTS; ++s) ncrit += pk[s] != 0xFFFF && (pk[s] & 255) == m + 1 && (pk[s] >> 8) != m + 1;
        if (last) ncrit = has_diag ? 1 : 0;
        if (ncrit) {
            if (!last && m >= 0 && !apply_column(m, true)) return;
            CP_STAMP(m + 1, 2);
            for (int s = 0; s < a.slots; ++s) {   // the four waves factor one panel together; a workgroup rarely has a second one in a column
                int i, j;
                tile(s, i, j);
                if (last ? (i != j) : (j != m + 1 || i == j)) continue;
                const bool ok = cp_panel4(i == j ? Town(s) : Td(s), Town(s), i == j, Sv(s), Sp, seq, panels++, lane, wave);
                if (!__builtin_amdgcn_readfirstlane((int)__all(ok)) && lane == 0) atomicOr(a.status, 2);
                __syncthreads();
            }
            if (last) {   // the inverse of the diagonal factor, for the backward sweep
                if (wave == 0) cp_invert_diag(Town(0), Td(0), Sv(0), lane);
                __syncthreads();
                break;
            }
            CP_STAMP(m + 1, 3);
            bool published = false;
            for (int s = 0; s < a.slots; ++s) {
                int i, j;
                tile(s, i, j);
                if (j != m + 1 || i == j) continue;   // nobody waits for a diagonal tile: it goes to S at the very end
                published = true;
                const double *To = Town(s);
                // the whole 32 x 32 tile (padding included; the rhs row's tile too: y in row 0) into its slot — write-through, nobody
                // waits for an acknowledgement — and the in-range part into S (plain stores: the result, read by nobody in this launch)
                double v[4];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int e = tid + 256 * qq;
                    v[qq] = To[(e >> 5) * CP_LDT + (e & 31)];
                    cp_st(a.tpub + ((int64_t)i * (i - 1) / 2 + j) * 1024 + e, v[qq]);
                }
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {   // behind the stores somebody is waiting for
                    const int e = tid + 256 * qq, gr = i * 32 + (e >> 5), gc = j * 32 + (e & 31);
                    if (gr < a.n && gc < a.n) a.S[(int64_t)gr * a.ld + gc] = v[qq];
                }
            }
            if (published) CP_STAMP(m + 1, 4);
        }
        if (last) break;
        // the rest of the trailing matrix, in the shadow of the next column's factorisation
        if (m >= 0 && !apply_column(m, false)) return;
        CP_STAMP(m + 1, 5);
    }

    // ---- backward substitution L' x = y: wave 0 of the diagonal tiles' owners ---------------------------------------------------------------
    // (every wave is past its last read of P and Q: apply_column ends with a barrier)
    if (wave != 0 || !has_diag) return;
    const int k = wg, c = lane & 31;
    auto any_fill = [&](const double (&v)[32]) -> bool {
        bool f = false;
#pragma unroll
        for (int r = 0; r < 32; ++r) f = f || (__builtin_bit_cast(uint64_t, v[r]) == CP_FILL);
        return __any(f);
    };
    // the 32 words of x_i into every lane (uniform addresses): a product with them needs no broadcast
    auto load_x = [&](double (&x)[32], const int i) {
#pragma unroll
        for (int r = 0; r < 32; ++r) x[r] = cp_ld(a.xpub + i * 32 + r);
    };
    // column c of the published tile (i, k): what lane c needs for (L_ik' x_i)[c].  Nothing orders this workgroup's arrival here after
    // the OTHER workgroups' tiles of column k (the owner of one of the last diagonal tiles gets here while they are still being
    // factored): the values are checked against the fill where they are used.
    auto load_col = [&](double (&dst)[32], const int i) {
#pragma unroll
        for (int r = 0; r < 32; ++r) dst[r] = cp_ld(a.tpub + ((int64_t)i * (i - 1) / 2 + k) * 1024 + r * 32 + c);
    };
    double t;                                                 // t = y_k - sum_{i > k + 1} L_ik' x_i, lane c holds entry c
    {
        const uint64_t t0 = wall_clock64();
        for (int spins = 1;; ++spins) {
            t = cp_ld(a.tpub + ((int64_t)nb * (nb - 1) / 2 + k) * 1024 + c);   // row 0 of the rhs row's tile of column k
            if (__all(__builtin_bit_cast(uint64_t, t) != CP_FILL)) break;
            if (!cp_poll_again(a, t0, spins, lane)) return;
        }
    }
    const double *Li = Td(0);
    double *Wl = Q;                                           // W' = L_k+1,k L_kk^-1, row r at Wl[32 r ..]: lane c reads its column
    if (k + 1 < nb) {
        // the tile below the diagonal tile, whole, into LDS (P, flat 32 x 32) ...
        double *Lt = P;
        const uint64_t t0 = wall_clock64();
        for (int spins = 1;; ++spins) {
            double v[16];
            bool f = false;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                v[q] = cp_ld(a.tpub + ((int64_t)(k + 1) * k / 2 + k) * 1024 + lane + 64 * q);
                f = f || (__builtin_bit_cast(uint64_t, v[q]) == CP_FILL);
            }
            if (!__any(f)) {
#pragma unroll
                for (int q = 0; q < 16; ++q) Lt[lane + 64 * q] = v[q];
                break;
            }
            if (!cp_poll_again(a, t0, spins, lane)) return;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ... and W'[r][c] = sum_q L[r][q] Li[q][c]: lane c with its column of the inverse in registers, L[r][.] as broadcast reads
        double li[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) li[q] = Li[q * CP_LDT + c];
        for (int r = 0; r < 32; ++r) {
            double w = 0.0;
#pragma unroll
            for (int q = 0; q < 32; ++q) w += Lt[r * 32 + q] * li[q];
            if (lane < 32) Wl[r * 32 + c] = w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    {
        double col[32], x[32];
        if (nb - 1 > k + 1) load_col(col, nb - 1);
        for (int i = nb - 1; i > k + 1; --i) {
            const uint64_t t0 = wall_clock64();
            double acc;
            for (int spins = 1;; ++spins) {
                load_x(x, i);
                acc = 0.0;
#pragma unroll
                for (int r = 0; r < 32; ++r) acc += col[r] * x[r];
                // the fill is a NaN and goes through the sum: only a NaN result is looked at word by word (a genuine NaN passes)
                if (__all(acc == acc) || !(any_fill(x) || any_fill(col))) break;
                if (!cp_poll_again(a, t0, spins, lane)) return;
                load_col(col, i);
            }
            t -= acc;
            if (i - 1 > k + 1) load_col(col, i - 1);   // the next arrival's tile, requested before that x is polled
        }
    }
    {
        double u = 0.0;   // u = L_kk^-T t (the inverse is stored with its zeros)
#pragma unroll
        for (int r = 0; r < 32; ++r) u += Li[r * CP_LDT + c] * lane_bcast(t, r);
        double xk = u;
        if (k + 1 < nb) {
            double w[32], x[32];
#pragma unroll
            for (int r = 0; r < 32; ++r) w[r] = Wl[r * 32 + c];
            const uint64_t t0 = wall_clock64();
            for (int spins = 1;; ++spins) {
                load_x(x, k + 1);
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < 32; ++r) acc += w[r] * x[r];
                xk = u - acc;
                if (__all(xk == xk) || !any_fill(x)) break;
                if (!cp_poll_again(a, t0, spins, lane)) return;
            }
            CP_STAMP(k, 6);
        }
        if (lane < 32) {
            cp_st(a.xpub + k * 32 + c, xk);
            if (k * 32 + c < a.n) a.x[k * 32 + c] = xk;
        }
        CP_STAMP(k, 7);
    }
    // the factor is complete in S once the diagonal tiles are there; their place still holds ORIGINAL entries that a late workgroup may
    // not have copied yet: wait until every workgroup has said it has
    if (!cp_wait_wave(a, CP_LOADED, G, lane)) return;
    {
        const double *To = Town(0);
        for (int e = lane; e < 32 * 32; e += 64) {
            const int r = e >> 5, cc = e & 31;
            const int gr = k * 32 + r, gc = k * 32 + cc;
            if (gr < a.n && cc <= r) a.S[(int64_t)gr * a.ld + gc] = To[r * CP_LDT + cc];
        }
    }
}

}  // namespace pcs

namespace pcs {

// Workspace of the persistent solve inside pcs_dense_spd_solve's d_work: [flags | xpub nb x 32 | tpub nb (nb + 1) / 2 tiles],
// all of it at the fill value when the kernel starts: ONE memset of 0xFF bytes.
inline int64_t cp_flag_doubles() { return CP_FLAGS * 4 / 8; }
inline int64_t cp_work_doubles(const int64_t nb) { return cp_flag_doubles() + nb * 32 + nb * (nb + 1) / 2 * 1024; }

// Can the persistent form take an n x n system on a device with `n_cus` compute units?  (one workgroup per CU, CP_MAX_SLOTS tiles each)
inline bool cp_fits(const int64_t n, const int n_cus) {
    const int64_t nb = (n + 31) / 32;
    return n > 0 && n_cus >= nb && cp_tiles(nb) <= (int64_t)CP_MAX_SLOTS * n_cus;   // every diagonal tile on a workgroup of its own
}

// Enqueue memset (unless `prefilled`) + kernel on `s`.  The caller has checked cp_fits and set the device.
inline hipError_t cp_launch(const int64_t n, double *d_S, const int64_t ld, const double *d_rhs, double *d_x, double *d_work, int32_t *d_status,
                            const int n_cus, hipStream_t s, const double timeout_s = 0.25, int64_t *trace = nullptr, const int32_t *d_stop = nullptr, const bool prefilled = false) {
    const int64_t nb = (n + 31) / 32, T = cp_tiles(nb);
    const int G = (int)(T < n_cus ? T : n_cus);
    const int slots = (int)((T + G - 1) / G);
    CholPersistArgs a{};
    a.S = d_S; a.rhs = d_rhs; a.x = d_x; a.status = d_status;
    a.flags = reinterpret_cast<int32_t *>(d_work);
    a.xpub = d_work + cp_flag_doubles();
    a.tpub = a.xpub + nb * 32;
    a.n = (int32_t)n; a.ld = (int32_t)ld; a.nb = (int32_t)nb; a.slots = slots;
    a.timeout_ticks = (int64_t)(timeout_s * 1.0e8);
    a.stop = d_stop;
#ifdef CP_TRACE
    a.trace = trace;
#else
    (void)trace;
#endif
    const size_t lds = cp_lds_bytes(slots);
    // prefilled: an earlier kernel on `s` has set the workspace to the fill value (schur_trail_lead_kernel in an LM trial)
    hipError_t e = prefilled ? hipSuccess : hipMemsetAsync(d_work, 0xFF, sizeof(double) * (size_t)cp_work_doubles(nb), s);
    if (e != hipSuccess) return e;
    // more than 64 KB of dynamic LDS needs the attribute, and it belongs to the function object of the CURRENT device (the caller has set
    // it): set it before every launch — microseconds, and right for a process that solves on several devices or from several threads
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(chol_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cp_lds_bytes(CP_MAX_SLOTS));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(chol_persist_kernel, dim3((unsigned)G), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace pcs
