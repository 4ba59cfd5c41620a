// ba_schur.hpp — the block parts of a Levenberg-Marquardt step on the blocked normal equations.
//
// The reference hands J to scipy (optimisation_handling.py:88-98) and lets trf / lsmr work out the step on the host.
// Here J^T J arrives from ba_normal.hpp already split as
//     [ A  B ] [x_l]     [g_l]        A  n_lead x n_lead   leading parameters (cameras; + poses for the self chain)
//     [ B' C ] [x_t] = - [g_t]        C  block diagonal    trailing entities (poses of the template chain, points of the
//                                                          self / free chains) never share a detection
// and the damped system (H + lambda D) x = -g, D = diag(H) (Marquardt), is reduced by the Schur complement of C:
//     C_e + lambda D_e = L_e L_e'                       per trailing entity, tb = 6 or 3, in registers      schur_trail_lead_kernel (trailing part)
//     V = B L^-T   (V_e = B_e L_e^-T)                   one tb-chunk of one row per lane                   schur_v_kernel
//     S = A + lambda D_l - V V',  rhs = -g_l + V u      u_e = L_e^-1 g_e                                    schur_trail_lead_kernel (leading part) + schur_syrk_kernel
//     S x_l = rhs                                       dense Cholesky of the LEADING size only
//     x_e = -L_e^-T (u_e + V_e' x_l)                                                                       schur_back_kernel
// Round 3 ends with no library call in the step: the two products around the dense solve are schur_syrk_kernel / schur_vtx_kernel
// below (FP64 matrix cores), the factorisation is csrc/ba_dense_chol.hpp; rocBLAS / rocSOLVER through torch remain as the A/B path
// of device_solver.py (dense_solver = 'rocsolver').  Fixed parameters (the handlers' masks: th:177-183, sbh:211-218) are not
// squeezed out of the system: their rows and columns are replaced by the identity and their gradient by zero, so their step
// is exactly zero and the block structure survives gauge points with single fixed coordinates (sbh:153-158) unpermuted.
// Everything — lambda included — is read from device memory: the LM driver never has to wait for the host to form a step.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pcs {

struct SchurArgs {
    double *A, *B;              // regions of the packed normal equations (B is masked in place: rows / columns of fixed parameters -> 0)
    const double *C, *g;
    const uint8_t *fixed;       // n_params: 1 = the parameter does not move
    const double *lambda;       // device scalar
    double *linvt;              // n_ent x tb x tb: L_e^-T (upper triangular, row-major)
    double *u;                  // n_trail: L_e^-1 g_e
    double *V;                  // n_lead x n_trail
    double *S;                  // n_lead x n_lead (full symmetric, before the - V V' update)
    double *rhs;                // n_lead: -g_l (the GEMV adds V u)
    double *dvec;               // n_params: the damping diagonal D (0 for fixed parameters)
    double *gm;                 // n_params: g with the fixed entries zeroed
    int32_t *status;            // bit 0: a trailing block was not positive definite
    int64_t n_lead, n_trail, n_ent, trail_off;
    const int32_t *stop;        // optional: a device word; non-zero = the LM loop has ended, this (speculatively queued) launch does nothing
    uint64_t *fill;             // optional (schur_trail_lead_kernel): fill_n words to set to all-ones — the hand-over workspace of the one-launch
    int64_t fill_n;             // Cholesky that follows in the same trial (ba_chol_persist.hpp), instead of a memset launch of its own
    int32_t trail_blocks;       // schur_trail_lead_kernel: the first trail_blocks workgroups take the trailing entities
    int32_t ent_chunks;         // schur_prep_kernel (ba_lm_fused.hpp): its first trail_blocks workgroups = ent_chunks entity chunks x row blocks of V
    // Round 5: an LM loop keeps TWO packed states and a device word that says which one is current (lm_decide_kernel flips it when a
    // trial is accepted — no copy).  A, B, C, g above are the regions of state 0; when *sel != 0 they are `alt` doubles further on.
    const int32_t *sel;
    int64_t alt;
};
// the packed regions of the CURRENT state (see SchurArgs::sel)
__device__ __forceinline__ SchurArgs schur_current(SchurArgs a) {
    if (a.sel && *a.sel) { a.A += a.alt; a.B += a.alt; a.C += a.alt; a.g += a.alt; }
    return a;
}
// Every kernel of an LM trial starts with this: the host queues trial t + 1 before it has read the verdict of trial t
// (pcs_lm_trial), and lm_decide_kernel raises the flag when the loop is over — what was queued behind it then drains as no-ops.
#define PCS_STOP_GUARD(a) do { if ((a).stop && *(a).stop) return; } while (0)

// Entity e: C_e + lambda D_e = L L', L^-T -> `linvt_out` (TB x TB, row-major; global memory or LDS), and — when `publish` — the entity's
// part of dvec / gm / u and the status bit (the fused kernel factors an entity once per block of leading rows and publishes once).
template <int TB>
__device__ __forceinline__ void schur_trail_entity(const SchurArgs &a, const int64_t e, double *linvt_out, const bool publish) {
    const double lam = *a.lambda;
    const double *Ce = a.C + e * TB * TB;
    const int64_t col0 = a.trail_off + e * TB;
    double M[TB][TB], gv[TB];
    bool fx[TB];
#pragma unroll
    for (int i = 0; i < TB; ++i) {
        fx[i] = a.fixed[col0 + i] != 0;
        gv[i] = fx[i] ? 0.0 : a.g[col0 + i];
    }
#pragma unroll
    for (int i = 0; i < TB; ++i)
#pragma unroll
        for (int j = i; j < TB; ++j) {   // the build writes the upper triangle
            double v = Ce[i * TB + j];
            if (fx[i] || fx[j]) v = (i == j) ? 1.0 : 0.0;
            M[i][j] = v;
            M[j][i] = v;
        }
#pragma unroll
    for (int i = 0; i < TB; ++i) {
        const double d = fx[i] ? 0.0 : fmax(M[i][i], 1e-300);
        if (publish) {
            a.dvec[col0 + i] = d;
            a.gm[col0 + i] = gv[i];
        }
        M[i][i] += lam * d;
    }
    // Cholesky M = L L' (lower triangle of M becomes L).  1 / sqrt(s) from the hardware estimate + one third-order step (full double
    // precision, ba_chol_persist.hpp's cp_rsqrt); the library's sqrt and the 27 divisions of the first version were most of this
    // lane's 9 us (one lane per entity: 200 lanes on rig-32 — latency, not throughput)
    bool ok = true;
    double ild[TB];   // 1 / L[j][j]
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        double s = M[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= M[j][k] * M[j][k];
        ok = ok && (s > 0.0);
        double il;
        {
            const double y = __builtin_amdgcn_rsq(s), h = __builtin_fma(-(s * y), y, 1.0);
            il = __builtin_fma(y * h, __builtin_fma(h, 0.375, 0.5), y);
        }
        ild[j] = il;
        const double l = s * il;
        M[j][j] = l;
#pragma unroll
        for (int i = j + 1; i < TB; ++i) {
            double t = M[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= M[i][k] * M[j][k];
            M[i][j] = t * il;
        }
    }
    if (!ok && publish) atomicOr(a.status, 1);
    // Linv = L^-1 (lower), column by column; u = Linv g
    double Li[TB][TB];
#pragma unroll
    for (int c = 0; c < TB; ++c)
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            if (i < c) { Li[i][c] = 0.0; continue; }
            double t = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int k = c; k < i; ++k) t -= M[i][k] * Li[k][c];
            Li[i][c] = t * ild[i];
        }
#pragma unroll
    for (int i = 0; i < TB; ++i) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k <= i; ++k) t += Li[i][k] * gv[k];
        if (publish) a.u[e * TB + i] = t;
    }
#pragma unroll
    for (int i = 0; i < TB; ++i)   // L^-T[i][j] = Linv[j][i]
#pragma unroll
        for (int j = 0; j < TB; ++j) linvt_out[i * TB + j] = Li[j][i];
}
// One lane = one trailing entity (workgroup `block` of the trailing part of schur_trail_lead_kernel).
template <int TB>
__device__ __forceinline__ void schur_trail_body(const SchurArgs &a, const int block) {
    const int64_t e = (int64_t)block * blockDim.x + threadIdx.x;
    if (e < a.n_ent) schur_trail_entity<TB>(a, e, a.linvt + e * TB * TB, true);
}

// One lane = one tb-chunk of one leading row: V[r, e, :] = b L_e^-T with b = the masked B[r, e, :].
template <int TB>
__global__ __launch_bounds__(256) void schur_v_kernel(const SchurArgs a0) {
    PCS_STOP_GUARD(a0);
    const SchurArgs a = schur_current(a0);
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_lead * a.n_ent) return;
    const int64_t r = t / a.n_ent, e = t - r * a.n_ent;
    double *Bp = a.B + r * a.n_trail + e * TB;
    double b[TB];
    bool touched = false;
    const bool row_fixed = a.fixed[r] != 0;
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        b[j] = Bp[j];
        if ((row_fixed || a.fixed[a.trail_off + e * TB + j]) && b[j] != 0.0) { b[j] = 0.0; touched = true; }
    }
    if (touched) {
#pragma unroll
        for (int j = 0; j < TB; ++j) Bp[j] = b[j];
    }
    const double *Lt = a.linvt + e * TB * TB;
    double *Vp = a.V + r * a.n_trail + e * TB;
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i <= j; ++i) s += b[i] * Lt[i * TB + j];
        Vp[j] = s;
    }
}

// S = sym(A) with fixed rows / columns -> identity and the damped diagonal.  One workgroup per 32 x 32 tile of S: the build writes the
// UPPER triangle of A, so a tile below the diagonal is the transpose of A's tile above it — read coalesced into LDS and written
// transposed (one lane per entry with `A[c][r]` for r > c read every second entry with a stride of a row: 20.7 us for the 45 MB of
// n_lead = 1 680).
__device__ __forceinline__ void schur_lead_body(const SchurArgs &a, const int block) {
    __shared__ double T[32][33];
    const int tid = threadIdx.x;
    const int nbt = (int)((a.n_lead + 31) / 32);
    const int bi = block / nbt, bj = block % nbt;      // tile (bi, bj) of S
    const int ti = bi < bj ? bi : bj, tj = bi < bj ? bj : bi;    // the tile of A's upper triangle it comes from
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q, r = e >> 5, c = e & 31;
        const int64_t gr = (int64_t)ti * 32 + r, gc = (int64_t)tj * 32 + c;
        T[r][c] = (gr < a.n_lead && gc < a.n_lead) ? a.A[gr * a.n_lead + gc] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q, r = e >> 5, c = e & 31;
        const int64_t gr = (int64_t)bi * 32 + r, gc = (int64_t)bj * 32 + c;
        if (gr >= a.n_lead || gc >= a.n_lead) continue;
        double v = bi < bj ? T[r][c] : bi > bj ? T[c][r] : (r <= c ? T[r][c] : T[c][r]);
        const bool fr = a.fixed[gr] != 0, fc = a.fixed[gc] != 0;
        if (fr || fc) v = (gr == gc) ? 1.0 : 0.0;
        if (gr == gc) {
            const double d = fr ? 0.0 : fmax(v, 1e-300);
            a.dvec[gr] = d;
            const double g = fr ? 0.0 : a.g[gr];
            a.gm[gr] = g;
            a.rhs[gr] = -g;
            v += *a.lambda * d;
        }
        a.S[gr * a.n_lead + gc] = v;
    }
}

// The two independent parts of the preparation in ONE launch: workgroups [0, trail_blocks) factor the trailing blocks
// (schur_trail_body), the others write S, rhs and the damping diagonal tile by tile (schur_lead_body); every workgroup takes a
// share of the optional fill (the hand-over workspace of the one-launch Cholesky that follows in an LM trial).
template <int TB>
__global__ __launch_bounds__(256) void schur_trail_lead_kernel(const SchurArgs a0) {
    PCS_STOP_GUARD(a0);
    const SchurArgs a = schur_current(a0);
    {
        const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (int64_t i = t; i < a.fill_n; i += (int64_t)gridDim.x * blockDim.x) a.fill[i] = ~0ull;
    }
    if ((int)blockIdx.x < a.trail_blocks) schur_trail_body<TB>(a, (int)blockIdx.x);
    else schur_lead_body(a, (int)blockIdx.x - a.trail_blocks);
}

// ---- S -= V V' (lower tiles) and rhs += V u on the FP64 matrix cores ---------------------------------------------------------------
// rocBLAS picks a 64 x 64 macro tile for the 480 x 480 x 1 200 product of rig-32: 64 workgroups on 256 CUs, 42 us (plus a GEMV
// launch of 6 us).  Only the lower triangle of S is needed (pcs_dense_spd_solve reads nothing else), and a tile's two operands
// are row blocks of the SAME matrix: one workgroup per 32 x 32 tile of the lower triangle (x a split of K when there are few
// tiles: 180 leading x 60 000 trailing of the 2e4-point free chain), four waves = four 16 x 16 quadrants, both row blocks
// staged through LDS 64 columns at a time (coalesced 512-byte row pieces; row stride 65 doubles: conflict-free operand
// reads), 16 v_mfma_f64_16x16x4 per staged chunk and wave, the next chunk in flight meanwhile.  The workgroups of the diagonal
// tiles add their rows' share of V u.  K is split until ~512 workgroups exist (two per CU hide the chunk loads of each other: one
// workgroup per tile and no split took 108 us on rig-32, twice the library call); partial sums meet in f64 atomics — like every
// entry of J'J itself (ba_normal.hpp), so the step's last bits were run-to-run dependent before this kernel.
struct SchurSyrkArgs {
    const double *V;     // n_lead x n_trail, row stride ldv
    double *S;           // n_lead x n_lead, row stride lds: lower triangle updated
    const double *u;     // n_trail (may be null: no rhs update)
    double *rhs;         // n_lead
    int32_t n_lead, n_trail, ldv, lds, ksplit, kchunk;   // kchunk: columns per split (multiple of 64)
    const int32_t *stop;
    // Ordered mode (engine option "deterministic"): with K split, a workgroup does not add its partial tile to S with atomics but stores
    // it to ws[(split * tiles + tile) * TW * TW + row * TW + column] (and its share of V u to ws_rhs[split * n_lead + row]);
    // schur_syrk_reduce_kernel then subtracts the splits in order.  NULL = atomics.
    double *ws, *ws_rhs;
    int32_t tiles;
};
constexpr int SYRK_LD = 65;   // odd: a quarter wave reads 16 rows, 16 different bank pairs (68: 4-way conflicts)
using schur_d4 = __attribute__((ext_vector_type(4))) double;

__global__ __launch_bounds__(256) void schur_syrk_kernel(const SchurSyrkArgs a) {
    PCS_STOP_GUARD(a);
    __shared__ double P[32][SYRK_LD];
    __shared__ double Q[32][SYRK_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x / a.ksplit, kc = blockIdx.x % a.ksplit;
    int bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);   // t -> (bi, bj), 0 <= bj <= bi
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    const int bj = t - bi * (bi + 1) / 2;
    const bool diag = bi == bj;
    const int k_begin = kc * a.kchunk, k_end = min(a.n_trail, k_begin + a.kchunk);
    const int i0 = 16 * (wave >> 1), j0 = 16 * (wave & 1);
    schur_d4 acc = {0.0, 0.0, 0.0, 0.0};
    const double *pq = diag ? &P[0][0] : &Q[0][0];
    const double *pa = &P[0][0] + (i0 + (lane & 15)) * SYRK_LD + (lane >> 4);
    const double *pb = pq + (j0 + (lane & 15)) * SYRK_LD + (lane >> 4);
    double dot = 0.0;   // diagonal tiles: this thread's share of (V u)[row], row = tid / 8
    // the next 64-column chunk travels from global memory into registers while the matrix cores work on the current one
    double pn[8], qn[8];
    auto fetch = [&](const int k0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 256 * q, r = e >> 6, c = e & 63;
            const int gk = k0 + c;
            const int gi = bi * 32 + r, gj = bj * 32 + r;
            pn[q] = (gi < a.n_lead && gk < k_end) ? a.V[(int64_t)gi * a.ldv + gk] : 0.0;
            qn[q] = (!diag && gj < a.n_lead && gk < k_end) ? a.V[(int64_t)gj * a.ldv + gk] : 0.0;
        }
    };
    if (k_begin < k_end) fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += 64) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 256 * q, r = e >> 6, c = e & 63;
            P[r][c] = pn[q];
            if (!diag) Q[r][c] = qn[q];
        }
        __syncthreads();
        if (k0 + 64 < k_end) fetch(k0 + 64);
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * s], pb[4 * s], acc, 0, 0, 0);
        if (diag && a.u) {
            const int r = tid >> 3, part = tid & 7;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int gk = k0 + part * 8 + c;
                dot += P[r][part * 8 + c] * (gk < k_end ? a.u[gk] : 0.0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gi = bi * 32 + i0 + (lane >> 4) + 4 * r, gj = bj * 32 + j0 + (lane & 15);
        if (a.ksplit > 1 && a.ws) {
            a.ws[((int64_t)kc * a.tiles + t) * 1024 + (i0 + (lane >> 4) + 4 * r) * 32 + j0 + (lane & 15)] = acc[r];
        } else if (gi < a.n_lead && gj <= gi) {
            double *dst = a.S + (int64_t)gi * a.lds + gj;
            if (a.ksplit == 1) *dst -= acc[r];
            else unsafeAtomicAdd(dst, -acc[r]);
        }
    }
    if (diag && a.u) {
        dot += __shfl_xor(dot, 1);
        dot += __shfl_xor(dot, 2);
        dot += __shfl_xor(dot, 4);
        const int gi = bi * 32 + (tid >> 3);
        if ((tid & 7) == 0 && gi < a.n_lead) {
            if (a.ksplit == 1) a.rhs[gi] += dot;
            else if (a.ws) a.ws_rhs[(int64_t)kc * a.n_lead + gi] = dot;
            else unsafeAtomicAdd(a.rhs + gi, dot);
        }
    }
}

// The same update on 64 x 64 tiles, for large leading groups (rig-32-self: 1 680 x 1 458).  With 32 x 32 tiles every workgroup
// streams 2 x 32 rows of V for 32 x 32 outputs: 1 431 tiles x 1 458 columns x 512 B = 1.05 GB through L2 / Infinity Cache for a
// 19.6 MB matrix — 175 us, bandwidth-bound at 6 TB/s while the matrix cores idle two thirds of the time (their floor: 60 us).
// A 64 x 64 tile halves the bytes per FMA: four waves = four 32 x 32 quadrants (2 x 2 MFMA tiles each: two operand reads feed
// four v_mfma_f64_16x16x4 per k-step), both row blocks staged 32 columns at a time (row stride 33 doubles: 36 cost 5 % in bank
// conflicts, 40 cost 23 %), the next chunk in flight meanwhile.  Same atomics, same rhs += V u by the diagonal tiles; K is split so
// that the workgroups fill whole rounds of the 2 x 256 resident ones (pcs_engine.hip: 378 tiles x 4 on rig-32-self, 147 -> 125 us).
// Measured and dropped (tools/probes/syrk_ld_probe.hip, profiles/r04/README.md): two chunks in flight (-4 %, +32 VGPRs), 64-column
// chunks (one wave per SIMD: slower), an XCD-aware super-tile order (no change: the operand stream is not what waits).  The matrix
// pipe is 47 % busy over the launch (SQ_VALU_MFMA_BUSY_CYCLES = 64 x SQ_INSTS_MFMA exactly; 58 us of matrix work).
// developer probes (tools/probes/syrk_ld_probe.hip) build the kernel with other chunk widths / row strides
#ifndef PCS_SYRK64_KC
#define PCS_SYRK64_KC 32
#endif
#ifndef PCS_SYRK64_LD
#define PCS_SYRK64_LD (PCS_SYRK64_KC + 1)
#endif
constexpr int SYRK64_KC = PCS_SYRK64_KC, SYRK64_LD = PCS_SYRK64_LD;   // odd row stride: the 16 rows a quarter wave reads land on 16 different bank pairs
__global__ __launch_bounds__(256) void schur_syrk64_kernel(const SchurSyrkArgs a) {
    PCS_STOP_GUARD(a);
    __shared__ double P[64][SYRK64_LD];
    __shared__ double Q[64][SYRK64_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto tri = [](const int t, int &i, int &j) {   // t -> (i, j), 0 <= j <= i
        i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        while (i * (i + 1) / 2 > t) --i;
        j = t - i * (i + 1) / 2;
    };
    int bi, bj;
    const int kc = blockIdx.x % a.ksplit;
    tri(blockIdx.x / a.ksplit, bi, bj);
    const bool diag = bi == bj;
    const int k_begin = kc * a.kchunk, k_end = min(a.n_trail, k_begin + a.kchunk);
    const int i0 = 32 * (wave >> 1), j0 = 32 * (wave & 1);   // this wave's quadrant
    schur_d4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = schur_d4{0.0, 0.0, 0.0, 0.0};
    const double *pq = diag ? &P[0][0] : &Q[0][0];
    const double *pa = &P[0][0] + (i0 + (lane & 15)) * SYRK64_LD + (lane >> 4);
    const double *pb = pq + (j0 + (lane & 15)) * SYRK64_LD + (lane >> 4);
    double dot = 0.0;   // diagonal tiles: this thread's share of (V u)[row], row = tid / 4
    // Chunks of KC columns: 64 rows x KC columns per operand = KC / 4 doubles per thread, the next chunk in flight while one is multiplied.
    constexpr int KC = SYRK64_KC, NQ = KC / 4;
    double pn[NQ], qn[NQ];
    auto fetch = [&](const int k0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tid + 256 * q, r = e / KC, c = e % KC;
            const int gk = k0 + c;
            const int gi = bi * 64 + r, gj = bj * 64 + r;
            pn[q] = (gi < a.n_lead && gk < k_end) ? a.V[(int64_t)gi * a.ldv + gk] : 0.0;
            qn[q] = (!diag && gj < a.n_lead && gk < k_end) ? a.V[(int64_t)gj * a.ldv + gk] : 0.0;
        }
    };
    if (k_begin < k_end) fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += KC) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tid + 256 * q, r = e / KC, c = e % KC;
            P[r][c] = pn[q];
            if (!diag) Q[r][c] = qn[q];
        }
        __syncthreads();
        if (k0 + KC < k_end) fetch(k0 + KC);
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            const double a0 = pa[4 * s], a1 = pa[16 * SYRK64_LD + 4 * s];
            const double b0 = pb[4 * s], b1 = pb[16 * SYRK64_LD + 4 * s];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (diag && a.u) {
            const int r = tid >> 2, part = tid & 3;
#pragma unroll
            for (int c = 0; c < KC / 4; ++c) {
                const int gk = k0 + part * (KC / 4) + c;
                dot += P[r][part * (KC / 4) + c] * (gk < k_end ? a.u[gk] : 0.0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = bi * 64 + i0 + 16 * x + (lane >> 4) + 4 * r, gj = bj * 64 + j0 + 16 * y + (lane & 15);
                if (a.ksplit > 1 && a.ws) {
                    a.ws[((int64_t)kc * a.tiles + blockIdx.x / a.ksplit) * 4096 + (i0 + 16 * x + (lane >> 4) + 4 * r) * 64 + j0 + 16 * y + (lane & 15)] = acc[x][y][r];
                } else if (gi < a.n_lead && gj <= gi) {
                    double *dst = a.S + (int64_t)gi * a.lds + gj;
                    if (a.ksplit == 1) *dst -= acc[x][y][r];
                    else unsafeAtomicAdd(dst, -acc[x][y][r]);
                }
            }
    if (diag && a.u) {
        dot += __shfl_xor(dot, 1);
        dot += __shfl_xor(dot, 2);
        const int gi = bi * 64 + (tid >> 2);
        if ((tid & 3) == 0 && gi < a.n_lead) {
            if (a.ksplit == 1) a.rhs[gi] += dot;
            else if (a.ws) a.ws_rhs[(int64_t)kc * a.n_lead + gi] = dot;
            else unsafeAtomicAdd(a.rhs + gi, dot);
        }
    }
}

// Ordered mode: S -= sum over the K splits of a tile's partial products, rhs += sum of the partial V u, split 0 first.  One thread per
// entry — TW * TW / 256 workgroups per tile of the lower triangle (TW = 32 or 64, the form that produced the partials) —, its loads
// eight in flight, added in split order.
template <int TW>
__global__ __launch_bounds__(256) void schur_syrk_reduce_kernel(const SchurSyrkArgs a) {
    PCS_STOP_GUARD(a);
    constexpr int PARTS = TW * TW / 256;
    const int tid = threadIdx.x, t = blockIdx.x / PARTS, part = blockIdx.x % PARTS;
    int bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);   // t -> (bi, bj), 0 <= bj <= bi
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    const int bj = t - bi * (bi + 1) / 2;
    auto ordered_sum = [&](const double *src, const int64_t stride) {
        double sum = 0.0;
        for (int kc = 0; kc < a.ksplit; kc += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = kc + u < a.ksplit ? src[(int64_t)(kc + u) * stride] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        return sum;
    };
    {
        const int e = part * 256 + tid;
        const int gi = bi * TW + e / TW, gj = bj * TW + e % TW;
        if (gi < a.n_lead && gj <= gi) a.S[(int64_t)gi * a.lds + gj] -= ordered_sum(a.ws + (int64_t)t * (TW * TW) + e, (int64_t)a.tiles * (TW * TW));
    }
    if (bi == bj && part == 0 && a.u && tid < TW) {
        const int gi = bi * TW + tid;
        if (gi < a.n_lead) a.rhs[gi] += ordered_sum(a.ws_rhs + gi, a.n_lead);
    }
}

// w = V' x (n_trail outputs): COLS columns per workgroup, the reads of a row piece coalesced, 1024 / COLS threads share the rows of a
// column (two independent chains each).  COLS = 64 means n_trail / 64 workgroups — 19 on rig-32, 23 on rig-32-self, of 256 CUs: each
// streams ~0.5 MB alone (8.6 / 24.5 us); with 16 columns (whole 128-byte lines per row piece) four times as many CUs take part.
template <int COLS>
__global__ __launch_bounds__(1024) void schur_vtx_kernel(const double *__restrict__ V, const double *__restrict__ x, double *__restrict__ w,
                                                         const int n_lead, const int n_trail, const int ldv, const int32_t *__restrict__ stop) {
    if (stop && *stop) return;
    constexpr int NPART = 1024 / COLS;
    const int col = threadIdx.x % COLS, part = threadIdx.x / COLS;
    const int j = blockIdx.x * COLS + col;
    __shared__ double red[NPART][COLS + 1];
    double s0 = 0.0, s1 = 0.0;
    if (j < n_trail) {
        int r = part;
        for (; r + NPART < n_lead; r += 2 * NPART) {   // two independent chains per thread
            s0 += V[(int64_t)r * ldv + j] * x[r];
            s1 += V[(int64_t)(r + NPART) * ldv + j] * x[r + NPART];
        }
        if (r < n_lead) s0 += V[(int64_t)r * ldv + j] * x[r];
    }
    red[part][col] = s0 + s1;
    __syncthreads();
    for (int half = NPART / 2; half >= 16; half >>= 1) {   // down to 16 partial sums per column
        if (part < half) red[part][col] += red[part + half][col];
        __syncthreads();
    }
    if (part == 0 && j < n_trail) {
        double s = 0.0;
#pragma unroll
        for (int p = 0; p < 16; ++p) s += red[p][col];
        w[j] = s;
    }
}
// few columns per workgroup while that brings more CUs in
inline void launch_schur_vtx(const double *V, const double *x, double *w, const int n_lead, const int n_trail, const int ldv, const int32_t *stop, hipStream_t s) {
    if ((n_trail + 63) / 64 >= 128) hipLaunchKernelGGL(schur_vtx_kernel<64>, dim3((unsigned)((n_trail + 63) / 64)), dim3(1024), 0, s, V, x, w, n_lead, n_trail, ldv, stop);
    else hipLaunchKernelGGL(schur_vtx_kernel<16>, dim3((unsigned)((n_trail + 15) / 16)), dim3(1024), 0, s, V, x, w, n_lead, n_trail, ldv, stop);
}

struct SchurBackArgs {
    const double *linvt, *u, *w, *xl;   // w = V' x_l (n_trail)
    const uint8_t *fixed;
    double *delta;                      // n_params, parameter-string order
    const double *ps_in;                // optional: the current parameter string ...
    double *ps_out;                     // ... and where the trial string ps_in + delta goes
    int64_t n_lead, n_ent, trail_off;
    const int32_t *stop;
    // LM loop with two states (SchurArgs::sel): ps_in / ps_out are the strings of state 0 / state 1; when *sel != 0 they swap roles
    const int32_t *sel;
    // optional: vote[0] (state 0 is current) or vote[vote_alt] (state 1 is current) <- 1.0 when bit 2 of *status is set (the one-launch
    // dense solve gave up), else 0.0.  The word sits behind the TRIAL state's packed buffer, so a sharded loop's all-reduce of that buffer
    // carries it: every rank learns that SOME rank's step is void (lm_decide_kernel) and all of them repeat the trial together.
    double *vote;
    int64_t vote_alt;
    const int32_t *status;
    // optional: zero_n doubles at `zero` (16-byte aligned) are set to 0 on the way — the trial state a dense build sums into right after
    // this kernel (pcs_genchain_lm_trial_build: the two fill launches of a hipMemsetAsync cost a small trial 9 us)
    double *zero;
    int64_t zero_n;
};

template <int TB>
__global__ __launch_bounds__(256) void schur_back_kernel(const SchurBackArgs a0) {
    PCS_STOP_GUARD(a0);
    SchurBackArgs a = a0;
    if (a.zero) {
        using D2 = __attribute__((ext_vector_type(2))) double;
        const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
        D2 *z = reinterpret_cast<D2 *>(a.zero);
        for (int64_t i = t0; i < a.zero_n / 2; i += nt) z[i] = D2{0.0, 0.0};
        if (t0 == 0 && (a.zero_n & 1)) a.zero[a.zero_n - 1] = 0.0;
    }
    const bool flipped = a.sel && *a.sel;
    if (flipped && a.ps_out) { a.ps_in = a0.ps_out; a.ps_out = const_cast<double *>(a0.ps_in); }
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && a.vote) a.vote[flipped ? a.vote_alt : 0] = (a.status && (*a.status & 4)) ? 1.0 : 0.0;
    if (t < a.n_lead) {
        const double d = a.fixed[t] ? 0.0 : a.xl[t];
        a.delta[t] = d;
        if (a.ps_out) a.ps_out[t] = a.ps_in[t] + d;
    }
    if (t >= a.n_ent) return;
    const double *Lt = a.linvt + t * TB * TB;
    double s[TB];
#pragma unroll
    for (int i = 0; i < TB; ++i) s[i] = a.u[t * TB + i] + a.w[t * TB + i];
#pragma unroll
    for (int i = 0; i < TB; ++i) {   // x = -L^-T s: L^-T is upper triangular
        double x = 0.0;
#pragma unroll
        for (int j = i; j < TB; ++j) x += Lt[i * TB + j] * s[j];
        const int64_t col = a.trail_off + t * TB + i;
        const double d = a.fixed[col] ? 0.0 : -x;
        a.delta[col] = d;
        if (a.ps_out) a.ps_out[col] = a.ps_in[col] + d;
    }
}

// The accept / reject decision of one LM trial, on the device (one workgroup): predicted reduction of the damped model,
// actual reduction, gain ratio, the next damping parameter (classic schedule: x 1/3 above 0.75, x 1 above 0.25,
// x 2 below, x 4 on a rejected or failed step — device_solver.lm_solve's host rule), and the numbers the host reads to
// follow the loop.  Nothing else of an iteration ever reaches the host.
// Round 4: with a control block (`ctrl`) the kernel also applies the loop's TERMINATION rules (gtol before the step, ftol / xtol
// after an accepted one, the limit of consecutive rejections, the iteration limit) and raises `*stop_flag`: the host no longer has
// to read a verdict before it may queue the next trial — whatever it queued speculatively starts with PCS_STOP_GUARD.
// Round 5: the loop keeps two states (packed normal equations + parameter string each) and `*sel` names the current one; an
// accepted trial becomes the current state by FLIPPING that word (round 4 copied 6.5 MB on rig-32, 42 MB on rig-32-self, in a launch
// of its own), and the trial's read-back goes to the host's page-locked buffer from here.
//   ctrl[0] stop code (0 = running; 1 gtol, 2 damping exhausted, 3 ftol, 4 xtol, 5 iteration limit, 9 = the dense solve gave up: host must
//           repeat the trial)   [1] consecutive rejections   [2] accepted steps   [3] iteration limit   [4] ftol [5] xtol [6] gtol
//           [7] rejection limit   [8] trials decided so far
//           [9] factor applied to lambda by a rejection BEFORE the first accepted step (0 = the classic 4): a start far from the
//               solution needs orders of magnitude more damping than the default initial value, not five rejections' worth of x 4
//           [10], [11] a gain ratio above ctrl[10] multiplies lambda by ctrl[11] instead of 1/3 (0 = off): an accurate model lets the
//               damping go quickly, so the loop can START with more of it ("damping policy")
struct LmDecideArgs {
    // &packed[s][n_packed - 1] of the two states: [0] sum r^2, [1] the void votes of the ranks (schur_back_kernel, with use_votes),
    // [-n_params .. -1] J^T r.  State *sel (0 without sel) is the current one, the other holds the trial.
    const double *tail[2];
    const double *ps2[2];                   // the two parameter strings, same roles
    int32_t *sel;
    const double *dvec, *gm, *delta;        // n_params each: damping diagonal, masked gradient, step
    const uint8_t *fixed;
    int32_t *status;                        // != 0: the step is invalid (a factorisation failed); cleared here for the next solve
    double *lambda;                         // in: the damping the step was computed with; out: the next one
    double *stats;                          // out[12]: accepted (-1: the trial is void, see below), max |g|, relative cost drop, |step|, |x|, new sum r^2,
                                            // old sum r^2, lambda used, stop code after this trial, trial number (-1: a no-op launch behind a raised flag),
                                            // the current state after this trial (0 / 1), lambda for the next trial
    int64_t n_params;
    double *ctrl;                           // optional control block (see above)
    int32_t *stop_flag;                     // with ctrl: the word PCS_STOP_GUARD reads
    int32_t *accept_flag;                   // with ctrl: 1 when this trial was accepted
    int32_t use_votes;                      // tail[trial][1] > 0: some rank's dense solve gave up — the trial is void on EVERY rank
    int32_t keep_sel;                       // 1: an accepted trial does not flip *sel (the caller copies the trial state over the current one:
                                            // lm_accept_kernel — a sharded loop builds into a FIXED buffer, the one its all-reduce was queued on)
    // optional (with ctrl): when this trial ends the loop, the final state goes to result (mapped page-locked host memory):
    // g[free_idx] | ps[free_idx] | sum r^2 of the state the loop ends in (the trial's if accepted, else the current one)
    const int64_t *free_idx;
    int64_t n_free;
    double *result;
    double *stats_host;                     // optional: the 12 numbers again, in mapped page-locked host memory the host has filled with LM_SENTINEL
};
constexpr int LM_STATS = 12;
// (the host side of the read-back protocol: pycamset_amd/device_solver.py LM_SENTINEL = a quiet NaN with a payload, 0x7FF8DEAD00000001)

__global__ __launch_bounds__(1024) void lm_decide_kernel(const LmDecideArgs a) {
    __shared__ double red[4][1024];
    const int tid = threadIdx.x;
    if (a.ctrl && a.ctrl[0] != 0.0) {   // queued behind the end of the loop: nothing to decide
        if (tid == 0) {
            a.stats[9] = -1.0;
            *a.accept_flag = 0;
            if (a.stats_host) {
#pragma unroll
                for (int i = 0; i < LM_STATS; ++i) a.stats_host[i] = -1.0;     // the host waits for all twelve words
            }
        }
        return;
    }
    const int cur = a.sel ? (*a.sel != 0 ? 1 : 0) : 0, tri = 1 - cur;
    const double *ps = a.ps2[cur];
    // the scalars of the decision, requested now: their latency passes behind the reductions (requested where they are used, thread 0
    // waited for five dependent round trips after the last barrier: 9.5 us per decision on rig-32)
    const double lam = *a.lambda, c_old = a.tail[cur][0], c_new = a.tail[tri][0];
    const int st = *a.status;
    const double votes = a.use_votes ? a.tail[tri][1] : 0.0;
    double cv[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (a.ctrl) {
#pragma unroll
        for (int i = 0; i < 12; ++i) cv[i] = a.ctrl[i];
    }
    double gd = 0.0, dd = 0.0, gmax = 0.0, xx = 0.0, ss = 0.0;
    for (int64_t i = tid; i < a.n_params; i += 1024) {
        const double d = a.delta[i], g = a.gm[i];
        gd += g * d;
        dd += a.dvec[i] * d * d;
        gmax = fmax(gmax, fabs(g));
        ss += d * d;
        if (!a.fixed[i]) xx += ps[i] * ps[i];
    }
    // five reductions (sum, sum, max, sum, sum): inside each wave by lane exchange, across the sixteen waves through LDS — ONE barrier
    // (binary trees through LDS with a barrier per level, until round 5: twenty-two barriers of 1024 threads, ~2.5 us of the 8)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        gd += __shfl_xor(gd, off);
        dd += __shfl_xor(dd, off);
        gmax = fmax(gmax, __shfl_xor(gmax, off));
        xx += __shfl_xor(xx, off);
        ss += __shfl_xor(ss, off);
    }
    if ((tid & 63) == 0) {
        const int w = tid >> 6;
        red[0][w] = gd; red[0][16 + w] = dd; red[0][32 + w] = gmax; red[0][48 + w] = xx; red[0][64 + w] = ss;
    }
    __syncthreads();
    double s_gd = 0.0, s_dd = 0.0, s_gmax = 0.0, s_xx = 0.0, s_ss = 0.0;
    if (tid == 0) {
#pragma unroll
        for (int w = 0; w < 16; ++w) {   // a fixed order: the same bits on every run (deterministic mode relies on it)
            s_gd += red[0][w]; s_dd += red[0][16 + w]; s_gmax = fmax(s_gmax, red[0][32 + w]); s_xx += red[0][48 + w]; s_ss += red[0][64 + w];
        }
    }
    if (tid == 0) {
        const double pred = 0.5 * (lam * s_dd - s_gd);
        const double actual = 0.5 * (c_old - c_new);
        const bool ok = st == 0 && pred == pred && fabs(pred) < 1.0e300;
        const double rho = pred > 0.0 ? actual / pred : -1.0;
        bool acc = ok && c_new == c_new && fabs(c_new) < 1.0e300 && actual > 0.0;
        const double factor = (cv[10] > 0.0 && cv[11] > 0.0 && rho > cv[10]) ? cv[11] : rho > 0.75 ? 1.0 / 3.0 : rho > 0.25 ? 1.0 : 2.0;
        const double rel_drop = actual / (0.5 * c_old), step_norm = sqrt(s_ss), x_norm = sqrt(s_xx);
        double code = 0.0, trial_no = 0.0;
        // the dense solve did not complete (ba_chol_persist.hpp's time limit) — here, or on some rank of a sharded loop: the host repeats the trial
        const bool void_trial = (st & 4) != 0 || votes > 0.0;
        double grow = 4.0;
        if (a.ctrl) {
            double *c = a.ctrl;
            if (cv[2] == 0.0 && cv[9] > 1.0) grow = cv[9];
            if (void_trial) { code = 9.0; acc = false; }
            else if (s_gmax <= cv[6]) { code = 1.0; acc = false; }             // the state BEFORE this step was already stationary: the step is dropped
            else if (acc) {
                c[1] = 0.0;
                c[2] = cv[2] + 1.0;
                if (rel_drop <= cv[4]) code = 3.0;
                else if (step_norm <= cv[5] * (cv[5] + x_norm)) code = 4.0;
                else if (cv[2] + 1.0 >= cv[3]) code = 5.0;
            } else {
                c[1] = cv[1] + 1.0;
                if (cv[1] + 1.0 >= cv[7]) code = 2.0;
            }
            trial_no = cv[8] + 1.0;
            c[8] = trial_no;
            c[0] = code;
            *a.accept_flag = acc ? 1 : 0;
            *a.stop_flag = code != 0.0 ? 1 : 0;
        }
        const double lam_next = (!void_trial && code != 1.0) ? (acc ? fmax(lam * factor, 1e-12) : lam * grow) : lam;
        *a.lambda = lam_next;
        *a.status = 0;
        const int now = (acc && a.sel && !a.keep_sel) ? tri : cur;           // the state the next trial starts from
        if (a.sel) *a.sel = now;
        const double out[LM_STATS] = {void_trial ? -1.0 : acc ? 1.0 : 0.0, s_gmax, rel_drop, step_norm, x_norm, c_new, c_old, lam, code, trial_no, (double)now, lam_next};
#pragma unroll
        for (int i = 0; i < LM_STATS; ++i)
            if (a.ctrl || (i != 8 && i != 9)) a.stats[i] = out[i];
        red[1][0] = code;                    // for the other threads: does the loop end here, and in which state
        red[1][1] = acc ? 1.0 : 0.0;
        red[1][2] = acc ? c_new : c_old;
        // the trial's read-back straight into the host's page-locked buffer (no copy launch, no fence: the host has filled the twelve
        // words with a NaN pattern no arithmetic produces and waits until none is left — the words may land in any order).  Not for the
        // trial that ENDS the loop: its read-back follows the final state (below)
        if (a.ctrl && a.stats_host && (code == 0.0 || !a.result)) {
#pragma unroll
            for (int i = 0; i < LM_STATS; ++i) a.stats_host[i] = out[i];
        }
    }
    if (!a.ctrl) return;
    __syncthreads();
    if (a.result && red[1][0] != 0.0) {      // the loop ends here: gradient, solution and cost of the final state for the host
        const int fin = red[1][1] != 0.0 ? tri : cur;
        const double *g = a.tail[fin] - a.n_params, *pf = a.ps2[fin];
        for (int64_t i = tid; i < a.n_free; i += 1024) {
            const int64_t k = a.free_idx[i];
            a.result[i] = g[k];
            a.result[a.n_free + i] = pf[k];
        }
        __threadfence_system();              // every thread's stores are out before ...
        __syncthreads();
        if (tid == 0) {                      // ... the word the host takes for "the final state is there"
            a.result[2 * a.n_free] = red[1][2];
            __threadfence_system();
        }
    }
    // ... and only then the read-back of the trial that ended the loop: a host that has seen it finds the final state complete
    if (tid == 0 && a.stats_host && a.result && red[1][0] != 0.0) {
#pragma unroll
        for (int i = 0; i < LM_STATS; ++i) a.stats_host[i] = a.stats[i];
    }
}

// Sharded loops only (LmDecideArgs::keep_sel): the build writes into the FIXED buffer its all-reduce was queued on, so an accepted
// trial is copied over the current state: packed[1] -> packed[0], string 1 -> string 0.  Always queued (the host does not know the
// verdict yet); copies only when lm_decide_kernel has set the accept flag.
__global__ __launch_bounds__(256) void lm_accept_kernel(const int32_t *__restrict__ accept_flag, const double *__restrict__ packed_new, double *__restrict__ packed_cur,
                                                        const int64_t n_packed, const double *__restrict__ ps_new, double *__restrict__ ps_cur, const int64_t n_params) {
    if (*accept_flag == 0) return;
    using D2 = __attribute__((ext_vector_type(2))) double;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    const D2 *src = reinterpret_cast<const D2 *>(packed_new);
    D2 *dst = reinterpret_cast<D2 *>(packed_cur);
    for (int64_t i = t; i < n_packed / 2; i += nt) dst[i] = src[i];
    if (t == 0 && (n_packed & 1)) packed_cur[n_packed - 1] = packed_new[n_packed - 1];
    for (int64_t i = t; i < n_params; i += nt) ps_cur[i] = ps_new[i];
}

}  // namespace pcs
