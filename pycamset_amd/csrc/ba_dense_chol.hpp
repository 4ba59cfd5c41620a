// ba_dense_chol.hpp — Cholesky factorisation and solve of the reduced camera system S x = rhs (FP64, n = the LEADING size:
// 480 on rig-32, 1 680 on rig-32-self), the dense piece of an LM step that follows ba_schur.hpp.
//
// Why not the library: rocSOLVER's potrf takes 1.15 ms for n = 480 and 4.6 ms for n = 1 680 on MI355X, its potrs another
// 0.45 / 0.6 ms (profiles/r03/lm_profile_*.log) — ten times everything else in the iteration together.  The matrix is small:
// what counts is the length of the dependent chain, not FLOPs.  Blocked right-looking factorisation, NB x NB tiles, two
// launches per block column k:
//   chol_panel_kernel   every workgroup factors the (already updated) diagonal tile A_kk in LDS and inverts the factor
//                       — redundantly: 10 k FMAs are cheaper than a hand-off between workgroups —; workgroup 0 stores
//                       L_kk and L_kk^-1, workgroup i > 0 turns its tile of the panel into L_ik = A_ik L_kk^-T with the inverse
//                       (a small GEMM, no sequential substitution);
//   chol_update_kernel  one workgroup per tile (i, j), i >= j > k, of the trailing matrix: A_ij -= L_ik L_jk'.
// chol_solve_kernel then runs both substitutions in ONE workgroup with the stored inverses of the diagonal tiles: per block a
// partial GEMV over the finished part and a small matrix-vector product — 2 n / NB dependent steps of a few hundred ns.
// Only the lower triangle of S is read and written.  A non-positive pivot sets bit 1 of *status (the LM loop reads it on the
// device and raises the damping).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pcs {

struct CholArgs {
    double *A;          // n x n, row-major, ld; lower triangle in / L out
    double *linv;       // (n / NB rounded up) x NB x NB: inverses of the diagonal tiles of L (lower triangular, row-major)
    int32_t *status;
    int32_t n, ld, k;   // k = block column of this launch
};

// tile element (r, c) of block (bi, bj), or the identity outside the matrix (a ragged last block is padded by the identity)
template <int NB>
__device__ __forceinline__ double chol_load(const CholArgs &a, const int bi, const int bj, const int r, const int c) {
    const int gr = bi * NB + r, gc = bj * NB + c;
    if (gr < a.n && gc < a.n) return a.A[(int64_t)gr * a.ld + gc];
    return (gr == gc) ? 1.0 : 0.0;
}

template <int NB>
__global__ __launch_bounds__(256) void chol_panel_kernel(const CholArgs a) {
    __shared__ double D[NB][NB + 1];    // diagonal tile -> L_kk (lower)
    __shared__ double Li[NB][NB + 1];   // L_kk^-1 (lower)
    __shared__ double X[NB][NB + 1];    // this workgroup's panel tile
    const int tid = threadIdx.x;
    const int k = a.k;
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e / NB, c = e % NB;
        D[r][c] = c <= r ? chol_load<NB>(a, k, k, r, c) : 0.0;
    }
    __syncthreads();
    // Right-looking factorisation of the tile, two barriers per column: (1) column j below the diagonal is scaled by 1 / l —
    // every thread reads the pivot, nobody writes it in this phase —, (2) the trailing part is updated and the pivot's owner
    // stores l.  Every thread owns the same NB * NB / 256 elements throughout; their (row, column) are compile-time shifts.
    constexpr int EPT = NB * NB / 256;
    bool ok = true;
    for (int j = 0; j < NB; ++j) {
        const double p = D[j][j];
        ok = ok && (p > 0.0);
        const double l = sqrt(p), il = 1.0 / l;
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + 256 * q, r = e / NB, c = e % NB;
            if (c == j && r > j) D[r][c] *= il;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + 256 * q, r = e / NB, c = e % NB;
            if (c > j && r >= c) D[r][c] -= D[r][j] * D[c][j];
            else if (r == j && c == j) D[r][c] = l;
        }
        __syncthreads();
    }
    if (!ok && blockIdx.x == 0 && tid == 0) atomicOr(a.status, 2);
    // Inverse of the lower-triangular tile: thread c solves column c by forward substitution, the column in REGISTERS (a first
    // version kept it in LDS: every step waited for its own previous LDS write, 28 us per tile — half of the whole factorisation)
    if (tid < NB) {
        const int c = tid;
        double col[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            double t = (r == c) ? 1.0 : 0.0;
#pragma unroll
            for (int m = 0; m < r; ++m) t -= D[r][m] * col[m];     // col[m] = 0 for m < c
            col[r] = r < c ? 0.0 : t / D[r][r];
        }
#pragma unroll
        for (int r = 0; r < NB; ++r) Li[r][c] = col[r];
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int e = tid; e < NB * NB; e += 256) {
            const int r = e / NB, c = e % NB;
            const int gr = k * NB + r, gc = k * NB + c;
            if (gr < a.n && gc < a.n && c <= r) a.A[(int64_t)gr * a.ld + gc] = D[r][c];
            a.linv[((int64_t)k * NB + r) * NB + c] = Li[r][c];
        }
        return;
    }
    const int bi = k + blockIdx.x;   // a tile below the diagonal: L_ik = A_ik L_kk^-T
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e / NB, c = e % NB;
        X[r][c] = chol_load<NB>(a, bi, k, r, c);
    }
    __syncthreads();
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e / NB, c = e % NB;
        double s = 0.0;
        for (int m = 0; m <= c; ++m) s += X[r][m] * Li[c][m];   // (L^-T)[m][c] = Linv[c][m]
        const int gr = bi * NB + r, gc = k * NB + c;
        if (gr < a.n && gc < a.n) a.A[(int64_t)gr * a.ld + gc] = s;
    }
}

// trailing update: workgroup t handles tile (i, j), k < j <= i, enumerated row by row
template <int NB>
__global__ __launch_bounds__(256) void chol_update_kernel(const CholArgs a) {
    __shared__ double Lik[NB][NB + 1];
    __shared__ double Ljk[NB][NB + 1];
    const int tid = threadIdx.x;
    const int k = a.k;
    // t -> (di, dj) with 0 <= dj <= di: di = floor((sqrt(8 t + 1) - 1) / 2)
    const int t = blockIdx.x;
    int di = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((di + 1) * (di + 2) / 2 <= t) ++di;
    while (di * (di + 1) / 2 > t) --di;
    const int dj = t - di * (di + 1) / 2;
    const int bi = k + 1 + di, bj = k + 1 + dj;
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e / NB, c = e % NB;
        Lik[r][c] = chol_load<NB>(a, bi, k, r, c);
        Ljk[r][c] = chol_load<NB>(a, bj, k, r, c);
    }
    __syncthreads();
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e / NB, c = e % NB;
        const int gr = bi * NB + r, gc = bj * NB + c;
        if (gr >= a.n || gc >= a.n || gc > gr) continue;
        double s = 0.0;
#pragma unroll 8
        for (int m = 0; m < NB; ++m) s += Lik[r][m] * Ljk[c][m];
        a.A[(int64_t)gr * a.ld + gc] -= s;
    }
}

struct CholSolveArgs {
    const double *L;      // factor (lower triangle of A)
    const double *linv;   // inverses of its diagonal tiles
    const double *rhs;
    double *x;            // n
    int32_t n, ld;
};

// L L' x = rhs in one workgroup of 1024 threads.  Block step: t = rhs_k - L[k, done] y[done] (all threads: NB rows x the finished
// columns, reduced through LDS), then y_k = Linv_kk t.  The backward sweep does the same with L' (columns of L).
template <int NB>
__global__ __launch_bounds__(1024) void chol_solve_kernel(const CholSolveArgs a) {
    extern __shared__ double sm[];
    double *y = sm;                       // n (padded to a multiple of NB)
    double *part = y + ((a.n + NB - 1) / NB) * NB;   // NB x G partial sums
    constexpr int G = 1024 / NB;          // threads per row
    const int tid = threadIdx.x;
    const int row = tid / G, lane = tid % G;
    const int nblk = (a.n + NB - 1) / NB;
    for (int i = tid; i < nblk * NB; i += 1024) y[i] = i < a.n ? a.rhs[i] : 0.0;
    __syncthreads();
    // forward: L y = rhs
    for (int k = 0; k < nblk; ++k) {
        const int gr = k * NB + row;
        double s = 0.0;
        if (gr < a.n)
            for (int c = lane; c < k * NB; c += G) s += a.L[(int64_t)gr * a.ld + c] * y[c];
        part[row * G + lane] = s;
        __syncthreads();
        if (tid < NB) {
            double t = y[k * NB + tid];
            for (int g = 0; g < G; ++g) t -= part[tid * G + g];
            part[tid * G] = t;                    // the block's right-hand side
        }
        __syncthreads();
        if (tid < NB) {
            double v = 0.0;
            for (int m = 0; m <= tid; ++m) v += a.linv[((int64_t)k * NB + tid) * NB + m] * part[m * G];
            y[k * NB + tid] = v;
        }
        __syncthreads();
    }
    // backward: L' x = y   (x overwrites y block by block, last block first).  Thread (col, lane): consecutive threads read
    // consecutive columns of a row of L — coalesced, where the forward sweep's mapping would stride by ld
    const int col = tid % NB, rl = tid / NB;
    for (int k = nblk - 1; k >= 0; --k) {
        const int gc = k * NB + col;               // column of L = row of L'
        double s = 0.0;
        if (gc < a.n)
            for (int r = (k + 1) * NB + rl; r < a.n; r += G) s += a.L[(int64_t)r * a.ld + gc] * y[r];
        part[col * G + rl] = s;
        __syncthreads();
        if (tid < NB) {
            double t = y[k * NB + tid];
            for (int g = 0; g < G; ++g) t -= part[tid * G + g];
            part[tid * G] = t;
        }
        __syncthreads();
        if (tid < NB) {
            double v = 0.0;
            for (int m = tid; m < NB; ++m) v += a.linv[((int64_t)k * NB + m) * NB + tid] * part[m * G];   // (Linv')[tid][m] = Linv[m][tid]
            y[k * NB + tid] = v;
        }
        __syncthreads();
    }
    for (int i = tid; i < a.n; i += 1024) a.x[i] = y[i];
}

}  // namespace pcs
