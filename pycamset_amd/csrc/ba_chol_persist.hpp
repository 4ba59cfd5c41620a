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
//   * published tiles go to S itself (write-through `sc1` stores, `s_waitcnt vmcnt(0)`, workgroup barrier, one agent-scope atomic
//     add on the column's counter); consumers poll the counter with `sc1` loads and read the tiles with `sc1` loads
//     (MI355X_MICROARCH.md, "Hand-offs measured with sc1 loads in place of the acquire", first row);
//   * each workgroup orders its work by urgency: tiles of the NEXT column first (update, factor, publish), the rest of the trailing
//     matrix afterwards, in the shadow of the next column's factorisation.
// Backward substitution L' x = y: distributed over the owners of the diagonal tiles, one wave each.  Owner k keeps
// s_k = sum_{i > k} L_ik' x_i up to date as the x_i appear (the tile for the next x is already in registers when it arrives),
// x_k = L_kk^-T (y_k - s_k) with the tile's inverse (formed off the critical path, right after the tile was factored), and
// publishes x_k as 32 data words that the consumers poll directly (a word differs from the 0xFF..FF fill once it is written).
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
    double *tpub;           // optional (data-polled hand-over): nb (nb - 1) / 2 tiles of 32 x 32, tile (i, j) at i (i - 1) / 2 + j, filled with 0xFF bytes
    double *ypub;           // nb x 32: y = L^-1 rhs, published per block column (0xFF-filled when tpub is used)
    double *xpub;           // nb x 32: x, published per block (filled with 0xFF bytes by the host)
    int32_t *flags;         // CP_COL + nb words, filled with 0xFF bytes (= -1) by the host: counters start at -1
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

constexpr int CP_ABORT = 0, CP_LOADED = 1, CP_COL = 8;   // flag words
constexpr int CP_LDT = 33;                               // row stride of a resident tile (row-per-lane access: conflict-free)
constexpr int CP_SLOT = 2 * 32 * CP_LDT + 32;            // doubles per slot: the tile, the private diagonal copy (diagonal owner: the inverse), s_k
constexpr int CP_MAX_SLOTS = 8;
constexpr uint64_t CP_FILL = 0xFFFFFFFFFFFFFFFFull;

__host__ __device__ inline int64_t cp_tiles(const int64_t nb) { return nb + nb * (nb + 1) / 2; }   // diagonal + (below-diagonal + rhs row) tiles
__host__ __device__ inline size_t cp_lds_bytes(const int slots) { return sizeof(double) * (2 * 32 * CHOL_LDP + (size_t)slots * CP_SLOT) + 64 * sizeof(int); }

__device__ __forceinline__ double cp_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }        // global_load_dwordx2 sc1
__device__ __forceinline__ void cp_st(double *p, const double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // global_store_dwordx2 sc1
__device__ __forceinline__ int cp_ldi(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// lane 0 of the calling wave: spin until flags[word] >= target; false = the launch is being abandoned
__device__ __forceinline__ bool cp_spin(const CholPersistArgs &a, const int word, const int target) {
    if (cp_ldi(a.flags + word) >= target) return true;
    const uint64_t t0 = wall_clock64();
    for (int spins = 1;; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if (cp_ldi(a.flags + word) >= target) return true;
        if ((spins & 63) == 0 && (cp_ldi(a.flags + CP_ABORT) >= 0 || (int64_t)(wall_clock64() - t0) > a.timeout_ticks)) {
            __hip_atomic_store(a.flags + CP_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(a.status, 4);
            return false;
        }
    }
}
// the whole workgroup waits (two barriers); counters start at -1, so "count arrivals" means target - 1
__device__ __forceinline__ bool cp_wait_wg(const CholPersistArgs &a, const int word, const int arrivals, int *lds_word, const int tid) {
    if (tid == 0) *lds_word = cp_spin(a, word, arrivals - 1) ? 1 : 0;
    __syncthreads();
    const int ok = *lds_word;
    __syncthreads();
    return ok != 0;
}
// one wave waits (no barrier)
__device__ __forceinline__ bool cp_wait_wave(const CholPersistArgs &a, const int word, const int arrivals, const int lane) {
    int ok = 1;
    if (lane == 0) ok = cp_spin(a, word, arrivals - 1) ? 1 : 0;
    return __builtin_amdgcn_readfirstlane(ok) != 0;
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
// (i, m), i > m, requested with sc1 loads; i == nb: the rhs row, y_m' in row 0.  Counter form: from S, after the column's counter
// has been seen.  Data-polled form (a.tpub): from the tile's slot in `tpub`, whole 8 KB tiles — the caller checks the values against
// the 0xFF fill and asks again until none is left (cp_fetch_polled).
__device__ __forceinline__ void cp_fetch(const CholPersistArgs &a, double (&v)[4], const int i, const int m, const int tid) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q, r = e >> 5, c = e & 31;
        v[q] = 0.0;
        if (i == a.nb) {
            if (r == 0) v[q] = cp_ld(a.ypub + m * 32 + c);
        } else if (a.tpub) {
            v[q] = cp_ld(a.tpub + ((int64_t)i * (i - 1) / 2 + m) * 1024 + e);
        } else {
            const int gr = i * 32 + r, gc = m * 32 + c;
            if (gr < a.n) v[q] = cp_ld(a.S + (int64_t)gr * a.ld + gc);   // gc < gr < n
        }
    }
}
__device__ __forceinline__ bool cp_is_fill(const double (&v)[4]) {
    bool f = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) f = f || (__builtin_bit_cast(uint64_t, v[q]) == CP_FILL);
    return f;
}
// Data-polled hand-over: fetch the operand tile(s) of an update until no value is the fill any more — the producer's stores are
// the signal (no counter, no acknowledged write, no second round trip).  All threads call it; false = the launch is being abandoned.
__device__ __forceinline__ bool cp_fetch_polled(const CholPersistArgs &a, double (&vp)[4], double (&vq)[4], const int i, const int j, const int m, const int tid) {
    const uint64_t t0 = wall_clock64();
    for (int spins = 1;; ++spins) {
        cp_fetch(a, vp, i, m, tid);
        if (i != j) cp_fetch(a, vq, j, m, tid);
        const bool pending = cp_is_fill(vp) || (i != j && cp_is_fill(vq));
        if (!__syncthreads_or(pending)) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 63) == 0) {
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

// The 64 x 32 panel [D; X] — D = the (private copy of the) diagonal tile, X = the tile below it — factored by the FOUR waves of the
// workgroup: lane r of every wave is row r of the panel (lanes 0-31 D, lanes 32-63 X; the elimination of D carries X along, X ends as
// X L^-T without any inverse), wave w holds the panel's columns 8 w .. 8 w + 7 in registers.  Block b of eight columns is factored by
// wave b alone (pivot and column entries from lane j by v_readlane), parked in LDS as `Sp[jl][row]`, and the waves behind it
// subtract its eight rank-1 terms from their own columns — row multiplier `Sp[jl][lane]`, column multiplier `Sp[jl][c]` as a
// broadcast read.  Critical path: 4 sub-panels (8 pivots, 28 updates each) + 3 x (barrier + one block update) instead of one
// wave's 32 pivots and 496 updates: 5.6 us -> ~2.6 us per block column, and 16 VGPRs of panel instead of 64.
// `Sp`: 2 x 8 x 64 doubles.  D == X's tile for the diagonal owner (diag = true): lanes 32-63 shadow lanes 0-31, L goes back to the
// tile with zeros above the diagonal, the reciprocal pivots to `ild` (32 doubles, for the inversion that follows later).
// All four waves call it (it contains barriers); returns false in the wave that met a non-positive pivot.
__device__ __forceinline__ bool cp_panel4(const double *Dt, double *Xt, const bool diag, double *ild, double *Sp, const int lane, const int wave) {
    const int r = lane & 31;
    const bool low = lane >= 32;
    const double *src = ((low && !diag) ? Xt : Dt) + r * CP_LDT + 8 * wave;
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = src[q];
    bool ok = true;
    double my_il = 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        double *buf = Sp + (b & 1) * 8 * 64;
        if (wave == b) {
#pragma unroll
            for (int jl = 0; jl < 8; ++jl) {
                const int j = 8 * b + jl;   // the pivot's row of D lives in lane j
                const double p = lane_bcast(v[jl], j);
                ok = ok && (p > 0.0);
                const double il = rsqrt_nr(p);
                my_il = (r == j) ? il : my_il;
                v[jl] *= il;   // lane j holds p itself: p / sqrt(p)
#pragma unroll
                for (int q = jl + 1; q < 8; ++q) v[q] -= v[jl] * lane_bcast(v[jl], 8 * b + q);   // L[c][j] lives in lane c
                buf[jl * 64 + lane] = v[jl];
            }
        }
        if (b < 3) {
            __syncthreads();
            if (wave > b) {
#pragma unroll
                for (int jl = 0; jl < 8; ++jl) {
                    const double mrow = buf[jl * 64 + lane];
                    const double *bc = buf + jl * 64 + 8 * wave;   // wave-uniform address: broadcast reads
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] -= mrow * bc[q];
                }
            }
        }
    }
    if (diag) {
        if (!low) {
#pragma unroll
            for (int q = 0; q < 8; ++q) Xt[r * CP_LDT + 8 * wave + q] = (8 * wave + q <= r) ? v[q] : 0.0;
            if ((r >> 3) == wave) ild[r] = my_il;
        }
    } else if (low) {
#pragma unroll
        for (int q = 0; q < 8; ++q) Xt[r * CP_LDT + 8 * wave + q] = v[q];
    }
    return ok;
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
    double *Sp = P;   // the panel's sub-panel exchange (2 x 8 x 64 doubles) shares the operand buffers: never in use at the same time
    int *meta = reinterpret_cast<int *>(slot0 + (size_t)a.slots * CP_SLOT);   // [0..7] packed tile of the slot, [16] wait word
    int *wword = meta + 16;
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

    // Block column m applied to every slot whose column is in [jlo, jhi]: fetch (sc1) -> park -> products, tile by tile.  (Requesting the
    // next tile's operands before the current tile's products, or all tiles' operands first, changed nothing: the trailing updates
    // hide behind the next column's factorisation either way — profiles/r04/README.md.)
    auto apply_column = [&](const int m, const int jlo, const int jhi) -> bool {
        for (int s = 0; s < a.slots; ++s) {
            int i, j;
            tile(s, i, j);
            if (i < 0 || j < jlo || j > jhi) continue;
            double vp[4], vq[4];
            if (a.tpub) {
                if (!cp_fetch_polled(a, vp, vq, i, j, m, tid)) return false;
            } else {
                cp_fetch(a, vp, i, m, tid);
                if (i != j) cp_fetch(a, vq, j, m, tid);
            }
            cp_park(P, vp, tid);
            if (i != j) cp_park(Q, vq, tid);
            __syncthreads();
            cp_apply(P, Q, Town(s), Td(s), i == j, lane, wave);
            __syncthreads();
        }
        return true;
    };

    // ---- factorisation: iteration m consumes block column m and produces the tiles of column m + 1 ----------------------------------------
    for (int m = -1; m < max_j; ++m) {
        CP_STAMP(m + 1, 0);
        if (m >= 0 && !a.tpub && !cp_wait_wg(a, CP_COL + m, nb - m, wword, tid)) return;   // data-polled form: the tiles themselves are the signal
        CP_STAMP(m + 1, 1);
        // urgent: the tiles of column m + 1
        int ncrit = 0;
#pragma unroll
        for (int s = 0; s < CP_MAX_SLOTS; ++s) ncrit += pk[s] != 0xFFFF && (pk[s] & 255) == m + 1;
        if (ncrit) {
            if (m >= 0 && !apply_column(m, m + 1, m + 1)) return;
            CP_STAMP(m + 1, 2);
            for (int s = 0; s < a.slots; ++s) {   // the four waves factor one panel together; a workgroup rarely has a second one in a column
                int i, j;
                tile(s, i, j);
                if (j != m + 1) continue;
                const bool ok = cp_panel4(i == j ? Town(s) : Td(s), Town(s), i == j, Sv(s), Sp, lane, wave);
                if (!__builtin_amdgcn_readfirstlane((int)__all(ok)) && lane == 0) atomicOr(a.status, 2);
                __syncthreads();
            }
            CP_STAMP(m + 1, 3);
            int published = 0;
            for (int s = 0; s < a.slots; ++s) {
                int i, j;
                tile(s, i, j);
                if (j != m + 1 || i == j) continue;   // nobody waits for a diagonal tile: it goes to S at the very end
                ++published;
                const double *To = Town(s);
                if (i == nb) {
                    if (tid < 32) cp_st(a.ypub + j * 32 + tid, To[tid]);
                } else if (a.tpub) {
                    // data-polled form: the whole 32 x 32 tile (padding included) into its slot — write-through, nobody waits for an
                    // acknowledgement — and the in-range part into S (plain stores: the result, read by nobody in this launch)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int e = tid + 256 * qq, r = e >> 5, c = e & 31;
                        const double v = To[r * CP_LDT + c];
                        cp_st(a.tpub + ((int64_t)i * (i - 1) / 2 + j) * 1024 + e, v);
                        const int gr = i * 32 + r, gc = j * 32 + c;
                        if (gr < a.n && gc < a.n) a.S[(int64_t)gr * a.ld + gc] = v;
                    }
                } else {
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int e = tid + 256 * qq, r = e >> 5, c = e & 31;
                        const int gr = i * 32 + r, gc = j * 32 + c;
                        if (gr < a.n && gc < a.n) cp_st(a.S + (int64_t)gr * a.ld + gc, To[r * CP_LDT + c]);
                    }
                }
            }
            if (published && !a.tpub) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                CP_STAMP(m + 1, 4);
                if (tid == 0) __hip_atomic_fetch_add(a.flags + CP_COL + m + 1, published, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (has_diag && wg == m + 1) {   // the inverse of the fresh diagonal factor (for the backward sweep), now that nobody waits for this workgroup
                if (wave == 0) cp_invert_diag(Town(0), Td(0), Sv(0), lane);
                __syncthreads();
            }
        }
        // the rest of the trailing matrix, in the shadow of the next column's factorisation
        if (m >= 0 && !apply_column(m, m + 2, 1 << 29)) return;
        CP_STAMP(m + 1, 5);
    }

    // ---- backward substitution L' x = y: wave 0 of the diagonal tiles' owners, everything in registers ------------------------------------
    if (wave != 0 || !has_diag) return;
    const int k = wg, c = lane & 31;
    double t;                                                 // t = y_k - sum_{i > k} L_ik' x_i, lane c holds entry c
    if (a.tpub) {                                             // y_k: polled like the tiles (0xFF fill until it is written)
        const uint64_t t0 = wall_clock64();
        for (int spins = 1;; ++spins) {
            t = cp_ld(a.ypub + k * 32 + c);
            if (__all(__builtin_bit_cast(uint64_t, t) != CP_FILL)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 63) == 0) {
                int give_up = 0;
                if (lane == 0 && (cp_ldi(a.flags + CP_ABORT) >= 0 || (int64_t)(wall_clock64() - t0) > a.timeout_ticks)) {
                    __hip_atomic_store(a.flags + CP_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomicOr(a.status, 4);
                    give_up = 1;
                }
                if (__builtin_amdgcn_readfirstlane(give_up)) return;
            }
        }
    } else {
        if (!cp_wait_wave(a, CP_COL + k, nb - k, lane)) return;   // y_k is part of column k (long complete for all but the last blocks)
        t = cp_ld(a.ypub + k * 32 + c);
    }
    auto fetch_col = [&](double (&dst)[32], const int i) {    // column c of tile (i, k): what lane c needs for L_ik' x_i
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int gr = i * 32 + r;
            if (a.tpub) dst[r] = cp_ld(a.tpub + ((int64_t)i * (i - 1) / 2 + k) * 1024 + r * 32 + c);   // may still hold the fill: col_ready() below
            else dst[r] = (gr < a.n) ? cp_ld(a.S + (int64_t)gr * a.ld + k * 32 + c) : 0.0;
        }
    };
    // Data-polled form: nothing orders this workgroup's arrival here after the OTHER workgroups' tiles of column k (the owner of the
    // last columns' diagonal tiles gets here while they are still being factored): the column is asked for again until no fill is left.
    auto col_ready = [&](double (&dst)[32], const int i) -> bool {
        if (!a.tpub) return true;
        const uint64_t t0 = wall_clock64();
        for (int spins = 1;; ++spins) {
            bool f = false;
#pragma unroll
            for (int r = 0; r < 32; ++r) f = f || (__builtin_bit_cast(uint64_t, dst[r]) == CP_FILL);
            if (!__any(f)) return true;
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 63) == 0) {
                int give_up = 0;
                if (lane == 0 && (cp_ldi(a.flags + CP_ABORT) >= 0 || (int64_t)(wall_clock64() - t0) > a.timeout_ticks)) {
                    __hip_atomic_store(a.flags + CP_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomicOr(a.status, 4);
                    give_up = 1;
                }
                if (__builtin_amdgcn_readfirstlane(give_up)) return false;
            }
            fetch_col(dst, i);
        }
    };
    double col[32];
    if (nb - 1 > k) fetch_col(col, nb - 1);
    for (int i = nb - 1; i > k; --i) {
        double xi;   // poll the 32 data words of x_i themselves: a word differs from the fill once it is written
        const uint64_t t0 = wall_clock64();
        for (int spins = 1;; ++spins) {
            xi = cp_ld(a.xpub + i * 32 + c);
            if (__all(__builtin_bit_cast(uint64_t, xi) != CP_FILL)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 63) == 0) {
                int give_up = 0;
                if (lane == 0 && (cp_ldi(a.flags + CP_ABORT) >= 0 || (int64_t)(wall_clock64() - t0) > a.timeout_ticks)) {
                    __hip_atomic_store(a.flags + CP_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomicOr(a.status, 4);
                    give_up = 1;
                }
                if (__builtin_amdgcn_readfirstlane(give_up)) return;
            }
        }
        if (i == k + 1) CP_STAMP(k, 6);
        if (!col_ready(col, i)) return;
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 32; ++r) acc += col[r] * lane_bcast(xi, r);
        t -= acc;
        if (i - 1 > k) fetch_col(col, i - 1);   // the next arrival's tile, requested before that x is polled
    }
    {
        const double *Li = Td(0);
        double xk = 0.0;
#pragma unroll
        for (int r = 0; r < 32; ++r) xk += Li[r * CP_LDT + c] * lane_bcast(t, r);   // x_k = L_kk^-T t (the inverse is stored with its zeros)
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

// Workspace of the persistent solve inside pcs_dense_spd_solve's d_work: [flags (CP_COL + nb ints, padded) | xpub nb x 32 | ypub nb x 32 |
// tpub nb (nb - 1) / 2 tiles]; everything a launch needs at the fill value is one contiguous range: ONE memset of 0xFF bytes
// (flags + xpub in the counter form; all of it in the data-polled form).
inline int64_t cp_flag_doubles(const int64_t nb) { return ((CP_COL + nb) * 4 + 63) / 64 * 8; }   // whole 64-byte lines
inline int64_t cp_work_doubles(const int64_t nb) { return cp_flag_doubles(nb) + 2 * nb * 32 + nb * (nb - 1) / 2 * 1024; }

// Can the persistent form take an n x n system on a device with `n_cus` compute units?  (one workgroup per CU, CP_MAX_SLOTS tiles each)
inline bool cp_fits(const int64_t n, const int n_cus) {
    const int64_t nb = (n + 31) / 32;
    return n > 0 && n_cus >= nb && cp_tiles(nb) <= (int64_t)CP_MAX_SLOTS * n_cus;   // every diagonal tile on a workgroup of its own
}

// Enqueue memset + kernel on `s`.  The caller has checked cp_fits and set the device.
inline hipError_t cp_launch(const int64_t n, double *d_S, const int64_t ld, const double *d_rhs, double *d_x, double *d_work, int32_t *d_status,
                            const int n_cus, hipStream_t s, const double timeout_s = 0.25, int64_t *trace = nullptr, const int32_t *d_stop = nullptr, const bool poll_data = true) {
    const int64_t nb = (n + 31) / 32, T = cp_tiles(nb);
    const int G = (int)(T < n_cus ? T : n_cus);
    const int slots = (int)((T + G - 1) / G);
    CholPersistArgs a{};
    a.S = d_S; a.rhs = d_rhs; a.x = d_x; a.status = d_status;
    a.flags = reinterpret_cast<int32_t *>(d_work);
    a.xpub = d_work + cp_flag_doubles(nb);
    a.ypub = a.xpub + nb * 32;
    a.tpub = poll_data ? a.ypub + nb * 32 : nullptr;
    a.n = (int32_t)n; a.ld = (int32_t)ld; a.nb = (int32_t)nb; a.slots = slots;
    a.timeout_ticks = (int64_t)(timeout_s * 1.0e8);
    a.stop = d_stop;
#ifdef CP_TRACE
    a.trace = trace;
#else
    (void)trace;
#endif
    const size_t lds = cp_lds_bytes(slots);
    hipError_t e = hipMemsetAsync(d_work, 0xFF, sizeof(double) * (size_t)(poll_data ? cp_work_doubles(nb) : cp_flag_doubles(nb) + nb * 32), s);
    if (e != hipSuccess) return e;
    static bool attr_set = false;   // one code object per process: the attribute sticks to the function
    if (!attr_set) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(chol_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cp_lds_bytes(CP_MAX_SLOTS));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(chol_persist_kernel, dim3((unsigned)G), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace pcs
