// ba_dense_chol.hpp — Cholesky factorisation and solve of the reduced camera system S x = rhs (FP64, n = the LEADING size:
// 480 on rig-32, 1 680 on rig-32-self), the dense piece of an LM step that follows ba_schur.hpp.
//
// Why not the library: rocSOLVER's potrf takes 1.15 ms for n = 480 and 4.6 ms for n = 1 680 on MI355X, its potrs another
// 0.45 / 0.6 ms (profiles/r03/lm_profile_*.log) — ten times everything else in the iteration together.  The matrix is small:
// what counts is the length of the dependent chain, not FLOPs.  Blocked right-looking factorisation, NB x NB tiles, ONE
// launch per block column:
//   chol_panel_kernel   (block column 0) every workgroup factors the diagonal tile and inverts the factor — redundantly: 10 k FMAs
//                       are cheaper than a hand-off between workgroups —, workgroup i > 0 turns its tile of the panel into
//                       L_i0 = A_i0 L_00^-T with the inverse (a small GEMM, no sequential substitution);
//   chol_step_kernel    (k = 0, 1, ...) one workgroup per tile (i, j), i >= j > k, of the trailing matrix: A_ij -= L_ik L_jk';
//                       the workgroups of column k + 1 go straight on with that column's panel step (they form the updated
//                       diagonal tile themselves).
// chol_solve_kernel then runs both substitutions in ONE workgroup with the stored inverses of the diagonal tiles: per block a
// partial GEMV over the finished part and a small matrix-vector product — 2 n / NB dependent steps of a few hundred ns.
// Only the lower triangle of S is read and written.  A non-positive pivot sets bit 1 of *status (the LM loop reads it on the
// device and raises the damping).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// Timing builds only (hipcc -DPCS_CHOL_SKIP=1 / 2 / 3): leave out wave 0's factorisation loop / inversion loop of a tile, to see what
// they cost inside a launch (5.2 / 3.4 us of 17, profiles/r03/README.md).  Results are wrong while it is non-zero.
#ifndef PCS_CHOL_SKIP
#define PCS_CHOL_SKIP 0
#endif

namespace pcs {

struct CholArgs {
    double *A;          // n x n, row-major, ld; lower triangle in / L out
    double *linv;       // (n / NB rounded up) x NB x NB: inverses of the diagonal tiles of L (lower triangular, row-major)
    double *ldiag;      // same shape: the diagonal tiles of L.  A launch does NOT write the tile it factors into A — its other
                        // workgroups read that tile of A in the same launch; the NEXT launch copies it there (the last one: chol_solve_kernel)
    int32_t *status;
    int32_t n, ld, k;   // k = block column of this launch
    const double *rhs;  // the forward substitution L y = rhs rides along with the factorisation: the workgroup that owns a
    double *y;          // diagonal tile forms its block of y as soon as the tile's inverse exists (rows of L left of it are final)
    const int32_t *stop;   // optional device word: non-zero = do nothing (a launch queued behind the end of an LM loop, ba_schur.hpp)
};

// tile element (r, c) of block (bi, bj), or the identity outside the matrix (a ragged last block is padded by the identity)
template <int NB>
__device__ __forceinline__ double chol_load(const CholArgs &a, const int bi, const int bj, const int r, const int c) {
    const int gr = bi * NB + r, gc = bj * NB + c;
    if (gr < a.n && gc < a.n) return a.A[(int64_t)gr * a.ld + gc];
    return (gr == gc) ? 1.0 : 0.0;
}

// 1 / sqrt(p) to full double precision without the library's sqrt + divide (~500 cycles of dependent latency per pivot in the
// first version): hardware estimate (v_rsq_f64, ~2^-26) + two Newton steps.
__device__ __forceinline__ double rsqrt_nr(const double p) {
    double y = __builtin_amdgcn_rsq(p);
    y = y * (1.5 - 0.5 * p * y * y);
    y = y * (1.5 - 0.5 * p * y * y);
    return y;
}
__device__ __forceinline__ double lane_bcast(const double v, const int src) {   // v_readlane_b32 x 2: src is a compile-time lane
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), src);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// Factor the lower-triangular tile held in LDS `D` (upper part zero) in place and leave the inverse of the factor in `Li`.
// Wave 0 only, the tile in REGISTERS (lane r = row r; no barriers, no LDS round trips): per column j the pivot and the column
// entries are broadcast with v_readlane (compile-time lanes after unrolling); the inverse by forward substitution with a
// register-resident column and the reciprocal pivots (no divisions).  Returns false (wave-uniform) for a non-positive pivot.
// The caller puts a workgroup barrier before (D complete) and after (D, Li complete).
template <int NB>
__device__ __forceinline__ bool factor_and_invert_tile(double (&D)[NB][NB + 1], double (&Li)[NB][NB + 1], const int tid) {   // NB + 1 = CHOL_LDF
    static_assert(NB == 32, "one row per lane of half a wave");
    bool ok = true;
    if (tid < 64) {
        const int r = tid & (NB - 1);
        double row[NB], ild[NB];
#if PCS_CHOL_SKIP
        for (int c = 0; c < NB; ++c) ild[c] = 1.0;
#endif
#pragma unroll
        for (int c = 0; c < NB; ++c) row[c] = tid < NB ? D[r][c] : (c == r ? 1.0 : 0.0);
#pragma unroll
        for (int j = 0; j < ((PCS_CHOL_SKIP & 1) ? 0 : NB); ++j) {
            const double p = lane_bcast(row[j], j);
            ok = ok && (p > 0.0);
            const double il = rsqrt_nr(p);
            ild[j] = il;
            row[j] = (r == j) ? p * il : row[j] * il;      // l = p / sqrt(p); below the diagonal: scaled (above: unused)
#pragma unroll
            for (int c = j + 1; c < NB; ++c) row[c] -= row[j] * lane_bcast(row[j], c);   // (c, j) lives in lane c
        }
        if (tid < NB) {
#pragma unroll
            for (int c = 0; c < NB; ++c) D[r][c] = c <= r ? row[c] : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (tid < NB) {
            // column c of the inverse: L x = e_c, COLUMN-oriented — once x[m] is known every later equation gets its term, 31 - m
            // independent FMAs.  (Row-oriented, the first version, each x[rr] was a chain of rr dependent FMAs on a SIMD this
            // wave has to itself: 4.3 us per tile; skipping the loop in a timing build showed it.)
            const int c = tid;
            double t[NB];
#pragma unroll
            for (int rr = 0; rr < NB; ++rr) t[rr] = (rr == c) ? 1.0 : 0.0;
#pragma unroll
            for (int m = 0; m < ((PCS_CHOL_SKIP & 2) ? 0 : NB); ++m) {
                const double xm = t[m] * ild[m];   // 0 for m < c
                Li[m][c] = xm;
#pragma unroll
                for (int rr = m + 1; rr < NB; ++rr) t[rr] -= D[rr][m] * xm;
            }
        }
    }
    return ok;
}

// ---- 32 x 32 x 32 products on the FP64 matrix cores ------------------------------------------------------------------------------
// Wave w of the four owns the 16 x 16 quadrant (16 (w >> 1), 16 (w & 1)) of a tile; v_mfma_f64_16x16x4 computes
// D[i][j] += sum_k A[k][i] B[k][j] with lane l supplying A[l >> 4][l & 15] and B[l >> 4][l & 15] per k-step of 4 and
// holding D[(l >> 4) + 4 r][l & 15] in register r.  Here both operands are ROWS of row-major LDS tiles (X Y' products):
// lane l reads P[i0 + (l & 15)][4 s + (l >> 4)] and Q[j0 + (l & 15)][4 s + (l >> 4)].  Eight k-steps, 8 MFMAs and 16
// ds_read_b64 per product and wave — the first version's VALU form read 9 LDS operands per 8 FMAs, ~4 us of a 22 us launch.
using chol_d4 = __attribute__((ext_vector_type(4))) double;
constexpr int CHOL_LDP = 33;   // row stride (doubles) of MFMA operand tiles: odd, so that the 16 rows a quarter wave reads land on 16 different bank pairs (36: 4-way conflicts, syrk64 lost 5 % to them)
constexpr int CHOL_LDF = 33;   // row stride of the tiles the factorisation walks row-per-lane (conflict-free)

template <int LDP_, int LDQ_, int KSTEPS = 8>
__device__ __forceinline__ chol_d4 chol_quadrant_xyT(const double *P, const double *Q, const int i0, const int j0, const int lane) {
    chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
    const double *p = P + (i0 + (lane & 15)) * LDP_ + (lane >> 4);
    const double *q = Q + (j0 + (lane & 15)) * LDQ_ + (lane >> 4);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(p[4 * s], q[4 * s], acc, 0, 0, 0);
    return acc;
}

// First block column: every workgroup factors the diagonal tile A_00 (redundantly: 10 k FMAs are cheaper than a hand-off
// between workgroups); workgroup 0 stores L_00 (into `ldiag`) and its inverse, workgroup i > 0 turns its tile into
// L_i0 = A_i0 L_00^-T (a product with the inverse on the matrix cores, no sequential substitution).
template <int NB>
__global__ __launch_bounds__(256) void chol_panel_kernel(const CholArgs a) {
    static_assert(NB == 32, "four waves, one 16 x 16 quadrant each");
    __shared__ double D[NB][CHOL_LDF];    // diagonal tile -> L_kk (lower)
    __shared__ double Li[NB][CHOL_LDF];   // L_kk^-1 (lower)
    __shared__ double X[NB][CHOL_LDP];    // this workgroup's panel tile
    __shared__ int flag_bad;
    if (a.stop && *a.stop) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.k;
    const int bi = k + blockIdx.x;      // blockIdx.x > 0: a tile below the diagonal
    constexpr int EPT = NB * NB / 256;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = tid + 256 * q, r = e / NB, c = e % NB;
        D[r][c] = c <= r ? chol_load<NB>(a, k, k, r, c) : 0.0;
        X[r][c] = blockIdx.x > 0 ? chol_load<NB>(a, bi, k, r, c) : 0.0;
    }
    if (tid == 0) flag_bad = 0;
    __syncthreads();
    const bool ok = factor_and_invert_tile<NB>(D, Li, tid);
    if (!ok && tid == 0) flag_bad = 1;
    __syncthreads();
    if (blockIdx.x == 0) {
        if (flag_bad && tid == 0) atomicOr(a.status, 2);
        for (int e = tid; e < NB * NB; e += 256) {
            const int r = e / NB, c = e % NB;
            a.ldiag[((int64_t)k * NB + r) * NB + c] = D[r][c];
            a.linv[((int64_t)k * NB + r) * NB + c] = Li[r][c];
        }
        if (tid < NB) {   // y_0 = L_00^-1 b_0
            double t = 0.0;
            for (int m = 0; m <= tid; ++m) t += Li[tid][m] * (k * NB + m < a.n ? a.rhs[k * NB + m] : 0.0);
            a.y[k * NB + tid] = t;
        }
        return;
    }
    const int i0 = 16 * (wave >> 1), j0 = 16 * (wave & 1);
    const chol_d4 l = chol_quadrant_xyT<CHOL_LDP, CHOL_LDF>(&X[0][0], &Li[0][0], i0, j0, lane);   // L_ik = A_ik L_kk^-T: (L^-T)[m][c] = Linv[c][m]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gr = bi * NB + i0 + (lane >> 4) + 4 * r, gc = k * NB + j0 + (lane & 15);
        if (gr < a.n && gc < a.n) a.A[(int64_t)gr * a.ld + gc] = l[r];
    }
}

// One launch per further block column: the trailing update with block column k (one workgroup per tile (i, j), k < j <= i:
// A_ij -= L_ik L_jk') AND the panel step of column k + 1 by the workgroups of that column (j = k + 1):
// each of them also forms the updated diagonal tile A_{k+1,k+1} - L_{k+1,k} L_{k+1,k}' (the one extra tile it needs is the
// L_jk it has loaded anyway), factors and inverts it, and writes its own updated tile as L_{i,k+1} = (A_ij - L_ik L_jk') L^-T
// straight away.  Half the launches of the update + panel pair and one global round trip less per block column.  All three
// products run on the matrix cores (chol_quadrant_xyT).
template <int NB>
__global__ __launch_bounds__(256) void chol_step_kernel(const CholArgs a) {
    static_assert(NB == 32, "four waves, one 16 x 16 quadrant each");
    __shared__ double Lik[NB][CHOL_LDP];   // later: this workgroup's updated tile
    __shared__ double Ljk[NB][CHOL_LDP];
    __shared__ double D[NB][CHOL_LDF];
    __shared__ double Li[NB][CHOL_LDF];
    __shared__ int flag_bad;
    if (a.stop && *a.stop) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.k;
    // workgroups 0 .. m - 1 (m = block rows of the trailing matrix): its first column, the launch's critical path — dispatched
    // first, so that a trailing matrix of more tiles than the chip holds at once (n = 1 680: 1 378) does not leave panel
    // tiles to the second round; the others enumerate the rest of the lower triangle row by row
    const int m = (a.n + NB - 1) / NB - k - 1;
    int di, dj;
    if ((int)blockIdx.x < m) {
        di = blockIdx.x;
        dj = 0;
    } else {
        const int t = blockIdx.x - m;   // -> (di - 1, dj - 1), 0 <= dj - 1 <= di - 1
        int ri = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((ri + 1) * (ri + 2) / 2 <= t) ++ri;
        while (ri * (ri + 1) / 2 > t) --ri;
        di = ri + 1;
        dj = t - ri * (ri + 1) / 2 + 1;
    }
    const int bi = k + 1 + di, bj = k + 1 + dj;
    const bool first_col = dj == 0;
    constexpr int EPT = NB * NB / 256;
    const int i0 = 16 * (wave >> 1), j0 = 16 * (wave & 1);
    // this lane's four entries of its quadrant (the MFMA result layout): rows i0 + (lane >> 4) + 4 r, column j0 + (lane & 15)
    const int qr = i0 + (lane >> 4), qc = j0 + (lane & 15);
    double own[4], dg[4];
    __shared__ double tv[NB];   // di == 0: sum_j L_{bj,j} y_j over the finished block columns j <= k (forward substitution)
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = tid + 256 * q, r = e / NB, c = e % NB;
        Lik[r][c] = chol_load<NB>(a, bi, k, r, c);
        Ljk[r][c] = chol_load<NB>(a, bj, k, r, c);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        own[r] = chol_load<NB>(a, bi, bj, qr + 4 * r, qc);
        dg[r] = (first_col && qc <= qr + 4 * r) ? chol_load<NB>(a, bj, bj, qr + 4 * r, qc) : 0.0;
    }
    if (tid == 0) flag_bad = 0;
    __syncthreads();
    {
        const chol_d4 u = chol_quadrant_xyT<CHOL_LDP, CHOL_LDP>(&Lik[0][0], &Ljk[0][0], i0, j0, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) own[r] -= u[r];
    }
    if (!first_col) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = bi * NB + qr + 4 * r, gc = bj * NB + qc;
            if (gr < a.n && gc < a.n && gc <= gr) a.A[(int64_t)gr * a.ld + gc] = own[r];
        }
        return;
    }
    {   // first column: bj = k + 1, Ljk = L_{k+1,k}; the updated diagonal tile (lower part)
        const chol_d4 u = chol_quadrant_xyT<CHOL_LDP, CHOL_LDP>(&Ljk[0][0], &Ljk[0][0], i0, j0, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) dg[r] = qc <= qr + 4 * r ? dg[r] - u[r] : 0.0;
    }
    __syncthreads();   // everybody is done reading Lik / Ljk
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        D[qr + 4 * r][qc] = dg[r];
        Lik[qr + 4 * r][qc] = own[r];
    }
    __syncthreads();
    if (di == 0 && tid >= 64 && tid < 192) {
        // forward substitution, the part that needs no inverse: t = sum_{j <= k} L_{bj,j} y_j for the 32 rows of this block — by waves
        // 1 and 2 WHILE wave 0 factors the tile (as the first thing of the workgroup it put 2.7 us on the launch's critical path)
        const int r = (tid - 64) >> 2, part = tid & 3;
        const int grow = bj * NB + r;
        double t = 0.0;
        if (grow < a.n) {
            const double *Lr = a.A + (int64_t)grow * a.ld + part * 8, *yy = a.y + part * 8;
            double acc[4] = {0, 0, 0, 0};
            int j = 0;
            for (; j + 4 <= k + 1; j += 4) {   // four block columns (4 x 64 bytes of the row) in flight at a time
                double l[4][8], v[4][8];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int c = 0; c < 8; ++c) { l[q][c] = Lr[(j + q) * NB + c]; v[q][c] = yy[(j + q) * NB + c]; }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[q] += l[q][c] * v[q][c];
            }
            for (; j <= k; ++j)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[j & 3] += Lr[j * NB + c] * yy[j * NB + c];
            t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
        t += __shfl_xor(t, 1);
        t += __shfl_xor(t, 2);
        if (part == 0) tv[r] = (grow < a.n ? a.rhs[grow] : 0.0) - t;   // b - sum: what the inverse is applied to
    }
    if (di == 0 && tid >= 192) {
        // wave 3, meanwhile: complete the factor in place — the diagonal tile of column k (from `ldiag`, written by the previous launch)
        // goes into A now; this launch reads column k only below the diagonal
        for (int e = tid - 192; e < NB * NB; e += 64) {
            const int r = e / NB, c = e % NB;
            const int gr = k * NB + r, gc = k * NB + c;
            if (gr < a.n && c <= r) a.A[(int64_t)gr * a.ld + gc] = a.ldiag[((int64_t)k * NB + r) * NB + c];
        }
    }
    const bool ok = factor_and_invert_tile<NB>(D, Li, tid);
    if (!ok && tid == 0) flag_bad = 1;
    __syncthreads();
    if (di == 0) {   // the diagonal tile's own workgroup
        if (flag_bad && tid == 0) atomicOr(a.status, 2);
        for (int e = tid; e < NB * NB; e += 256) {
            const int r = e / NB, c = e % NB;
            a.ldiag[((int64_t)bj * NB + r) * NB + c] = D[r][c];
            a.linv[((int64_t)bj * NB + r) * NB + c] = Li[r][c];
        }
        if (tid < NB) {   // y_bj = L^-1 (b_bj - sum_j L_{bj,j} y_j)
            double t = 0.0;
#pragma unroll
            for (int m = 0; m < NB; ++m) t += Li[tid][m] * tv[m];   // Li is lower triangular: zeros right of the diagonal
            a.y[bj * NB + tid] = t;
        }
        return;
    }
    const chol_d4 l = chol_quadrant_xyT<CHOL_LDP, CHOL_LDF>(&Lik[0][0], &Li[0][0], i0, j0, lane);   // L_{i,k+1} = (updated tile) L^-T
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gr = bi * NB + qr + 4 * r, gc = bj * NB + qc;
        if (gr < a.n && gc < a.n) a.A[(int64_t)gr * a.ld + gc] = l[r];
    }
}

struct CholSolveArgs {
    double *L;            // factor (lower triangle of A; its diagonal tiles arrive from ldiag here)
    const double *linv;   // inverses of its diagonal tiles
    const double *ldiag;  // the diagonal tiles themselves
    double *y;            // L y = rhs, padded to whole blocks: formed by the factorisation launches (CholArgs::y); the GEMV launches
                          // between the pieces of the backward sweep update it in place
    double *x;            // n
    int32_t n, ld;
    int32_t kb0, kb1;     // this launch solves the blocks kb1 - 1 down to kb0 (at most 16: one unknown per thread)
    const int32_t *stop;
};

// L' x = y for the block range [kb0, kb1) in ONE workgroup of 512 threads, COLUMN-oriented: once a block of the solution is
// known (32 x 32 product with the stored inverse of the diagonal tile, 32 threads), every other thread subtracts that block's
// contribution from the unknown it owns — a 32-term dot product with one panel column, no cross-thread reduction.  Nothing
// that is loaded depends on the solution, so the panel values and the inverse rows of step k - 1 are requested during step k
// (registers): the dependent chain per step is two barriers and ~100 FMAs instead of two global-memory round trips.
// One workgroup has one CU's bandwidth: the whole sweep of n = 1 680 in it (11 MB of L, the first version) took 561 us of a
// 1 566 us solve.  pcs_dense_spd_solve therefore halves the triangle recursively — lower-right half, then chol_gemv_t_kernel
// (many workgroups) takes the solved half out of the rest of y, then the upper-left half — down to pieces of at most 16 blocks.
// The forward sweep is not here at all: it rides along with the factorisation launches.
template <int NB>
__global__ __launch_bounds__(512) void chol_solve_kernel(const CholSolveArgs a) {
    extern __shared__ double sm[];
    if (a.stop && *a.stop) return;
    constexpr int T = 512;
    const int tid = threadIdx.x;
    const int n = a.n;
    const int nblk = (n + NB - 1) / NB;
    const int c0 = a.kb0 * NB;                   // first unknown of the range
    const int nloc = (a.kb1 - a.kb0) * NB;       // <= T
    double *y = sm;            // running right-hand side -> solution of the range (nloc)
    double *blk = y + nloc;    // the block solved in this step (NB)
    for (int i = tid; i < nloc; i += T) y[i] = a.y[c0 + i];
    if (a.kb1 == nblk) {
        for (int e = tid; e < NB * NB; e += T) {   // complete the factor in place: the LAST diagonal tile (the others: chol_step_kernel; the sweep never reads them)
            const int k = nblk - 1, r = e / NB, c = e % NB;
            const int gr = k * NB + r, gc = k * NB + c;
            if (gr < n && c <= r) a.L[(int64_t)gr * a.ld + gc] = a.ldiag[(int64_t)k * NB * NB + e];
        }
    }
    double cur[NB], nxt[NB], lin[NB];
    auto panel_col = [&](const int k, double (&dst)[NB]) {      // column c0 + tid (< k NB) of block row k
        if (c0 + tid < k * NB) {
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const int gr = k * NB + r;
                dst[r] = gr < n ? a.L[(int64_t)gr * a.ld + c0 + tid] : 0.0;
            }
        }
    };
    auto inv_col = [&](const int k) {                           // column tid of Linv_k = row tid of Linv_k'
        if (tid < NB) {
#pragma unroll
            for (int m = 0; m < NB; ++m) lin[m] = a.linv[((int64_t)k * NB + m) * NB + tid];
        }
    };
    inv_col(a.kb1 - 1);
    panel_col(a.kb1 - 1, cur);
    __syncthreads();
    for (int k = a.kb1 - 1; k >= a.kb0; --k) {
        const int kl = (k - a.kb0) * NB;   // offset of block k inside the range
        if (tid < NB) {
            double t = 0.0;
#pragma unroll
            for (int m = 0; m < NB; ++m) t += (m >= tid ? lin[m] : 0.0) * y[kl + m];
            blk[tid] = t;
        }
        __syncthreads();
        if (tid < NB) y[kl + tid] = blk[tid];
        if (k > a.kb0) { inv_col(k - 1); panel_col(k - 1, nxt); }
        if (tid < kl) {   // unknown c0 + tid lies left of block k
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < NB; ++r) s += cur[r] * blk[r];
            y[tid] -= s;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NB; ++c) cur[c] = nxt[c];
    }
    for (int i = tid; i < nloc; i += T)
        if (c0 + i < n) a.x[c0 + i] = y[i];
}

// y[c] -= sum_r L[r][c] x[r] for the columns [c0, c1) and the rows [r0, r1) of the factor (r0 >= c1: a rectangle below the
// diagonal): a solved piece of the backward sweep leaves the rest of y.  64 columns per workgroup, sixteen waves share the rows,
// one writer per entry (deterministic).
__global__ __launch_bounds__(1024) void chol_gemv_t_kernel(const double *__restrict__ L, const int ld, const double *__restrict__ x, double *__restrict__ y,
                                                           const int r0, const int r1, const int c0, const int c1, const int32_t *__restrict__ stop) {
    if (stop && *stop) return;
    const int col = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int j = c0 + blockIdx.x * 64 + col;
    __shared__ double red[16][64];
    double s0 = 0.0, s1 = 0.0;
    if (j < c1) {
        int r = r0 + part;
        for (; r + 16 < r1; r += 32) {   // two independent chains per thread
            s0 += L[(int64_t)r * ld + j] * x[r];
            s1 += L[(int64_t)(r + 16) * ld + j] * x[r + 16];
        }
        if (r < r1) s0 += L[(int64_t)r * ld + j] * x[r];
    }
    red[part][col] = s0 + s1;
    __syncthreads();
    if (part == 0 && j < c1) {
        double s = 0.0;
#pragma unroll
        for (int p = 0; p < 16; ++p) s += red[p][col];
        y[j] -= s;
    }
}

}  // namespace pcs
