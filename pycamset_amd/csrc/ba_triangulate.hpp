// ba_triangulate.hpp — batched n-view triangulation.
//
// Reference: nb_triangulate_full / nb_triangulate_nviews (compiled_helpers.py:609-663), front end
// CameraSet.multi_cam_triangulate (cameras/camera_set.py:343-402).  Per 3-D point seen by n >= 2
// cameras the reference undistorts every observation (5 fixed-point iterations, ch:409-431), stacks
//     M = [ P_i | 0 .. -x_i .. 0 ]   (3n x (4+n)),  x_i = (u_i, v_i, 1) undistorted pixels
// and returns the right singular vector of the smallest singular value (LAPACK SVD), X[:3] / X[3].
//
// Here a group of G lanes owns one point and never forms M.  The lambda columns of M have disjoint supports, so
// a 3x3 Householder reflector per view (Q_i x_i = alpha_i e_1) triangularises them exactly:
//     Q^T M = [ R  E ]   R: n x 4 (first rows of Q_i P_i),  E = diag(-alpha_i)
//             [ C  0 ]   C: 2n x 4 (other two rows)
// C is folded view by view into a 4x4 upper-triangular R_C with Givens rotations (streaming QR, no
// storage), which gives the triangular factor [[E, R], [0, R_C]] of M up to a column permutation.
// Inverse iteration with that factor (two triangular solves per step, O(n)) converges to the smallest
// right singular vector at the rate (sigma_min / sigma_next)^2 per step; 3-6 steps.  Everything is
// Householder / Givens / triangular solves, i.e. backward stable like the SVD — forming the 4x4 normal
// matrix instead (secular equation) was tried first and loses 1e-8..1e-4 on 2-5 view points, because
// sigma_min^2 ~ 1e-9 sits below the rounding level of entries ~1e6.
// Agreement with the LAPACK SVD: within the SVD's own conditioning bound eps * sigma_1 / gap
// (tests/test_gpu_triangulate.py); 2e-13 relative on 10-30 view points.
#pragma once
#include <hip/hip_runtime.h>

namespace pcs {

constexpr int TRI_CAM_STRIDE = 32;  // P 12 | (10 unused) | fx cx fy cy | k0 k1 p0 p1 k2 | pad

// ch:409-431 nb_undistort: 5 fixed-point iterations of the Brown-Conrady model
__device__ __forceinline__ void undistort5(const double u, const double v, const double *__restrict__ ct, double &uo, double &vo) {
    const double fx = ct[22], cx = ct[23], fy = ct[24], cy = ct[25];
    const double k0 = ct[26], k1 = ct[27], p0 = ct[28], p1 = ct[29], k2 = ct[30];
    const double x0 = (u - cx) / fx, y0 = (v - cy) / fy;
    double x = x0, y = y0;
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const double r2 = x * x + y * y;
        const double k_inv = 1.0 / (1.0 + k0 * r2 + k1 * (r2 * r2) + k2 * (r2 * r2 * r2));
        const double xD = 2.0 * p0 * x * y + p1 * (r2 + 2.0 * (x * x));
        const double yD = p0 * (r2 + 2.0 * (y * y)) + 2.0 * p1 * x * y;
        x = (x0 - xD) * k_inv;
        y = (y0 - yD) * k_inv;
    }
    uo = x * fx + cx;
    vo = y * fy + cy;
}

// First row r (4), scale alpha and optionally the two other rows c0, c1 of Q P for one view, where Q is
// the Householder reflector with Q (u, v, 1)^T = alpha e_1.
__device__ __forceinline__ void view_rows(const double *__restrict__ P, const double u, const double v, double &alpha, double (&r)[4],
                                          double (&c0)[4], double (&c1)[4]) {
    const double nx = sqrt(u * u + v * v + 1.0);
    alpha = (u >= 0.0) ? -nx : nx;  // opposite sign of x_1: no cancellation in v_1
    const double v0 = u - alpha, v1 = v, v2 = 1.0;
    const double f = 2.0 / (v0 * v0 + v1 * v1 + v2 * v2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double t = f * (v0 * P[k] + v1 * P[4 + k] + v2 * P[8 + k]);
        r[k] = P[k] - v0 * t;
        c0[k] = P[4 + k] - v1 * t;
        c1[k] = P[8 + k] - v2 * t;
    }
}

// fold one row into the upper-triangular 4x4 factor (streaming QR update)
__device__ __forceinline__ void givens_insert(double (&R)[4][4], double (&row)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double a = R[k][k], b = row[k];
        if (b != 0.0) {
            const double h = sqrt(a * a + b * b);
            const double c = a / h, s = b / h;
#pragma unroll
            for (int j = k; j < 4; ++j) {
                const double rk = R[k][j], rw = row[j];
                R[k][j] = c * rk + s * rw;
                row[j] = c * rw - s * rk;
            }
        }
    }
}

// G lanes per point (64 / G points per wave): lane g of a group handles views g, g+G, ...
// G = 1 is the plain lane-per-point form; it is latency-bound (97 k points = 1 500 waves, each walking
// ~10 views serially through every pass).  G = 16 has the shortest critical path but executes the 4x4
// solves once per 4 points instead of once per 64, and ends up issue-bound at the same time.  The
// default is in between (profiles/r01/tri_legacy_bench.log).  Every lane folds its views into a private
// 4x4 factor; the G factors are merged pairwise (xor-shuffle tree: the partner's rows are folded in
// with Givens rotations = QR of stacked triangular factors, still backward stable); the 4-vector and
// norm sums of the inverse iteration are group reductions.  Pass 1 caches each view's Householder row
// r_i and alpha_i (scratch, 48 B per observation) so the later passes do no sqrt / divide per view.
template <int G>
__device__ __forceinline__ double group_sum(double x) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, G);
    return x;
}

template <int G>
__global__ __launch_bounds__(256) void triangulate_kernel(const int32_t *__restrict__ cam, const double2 *__restrict__ uv,
                                                          const int64_t *__restrict__ start, const double *__restrict__ cam_tab,
                                                          double4 *__restrict__ scr_r, double2 *__restrict__ scr_al,
                                                          double *__restrict__ pts, int64_t n_pts) {
    const int64_t gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int g = threadIdx.x & (G - 1);
    const bool live = gid < n_pts;  // whole groups are live or dead; dead groups still take part in the shuffles
    const int64_t j = live ? gid : n_pts - 1;
    const int64_t s0 = start[j], s1 = live ? start[j + 1] : s0;
    double R[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int64_t q = s0 + g; q < s1; q += G) {  // pass 1: undistort, Householder rows, fold the C rows into R
        const double *ct = cam_tab + (int64_t)cam[q] * TRI_CAM_STRIDE;
        const double2 m = uv[q];
        double uu, vv, alpha, r[4], c0[4], c1[4];
        undistort5(m.x, m.y, ct, uu, vv);
        view_rows(ct, uu, vv, alpha, r, c0, c1);
        scr_r[q] = make_double4(r[0], r[1], r[2], r[3]);
        scr_al[q] = make_double2(-1.0 / alpha, 0.0);  // (1 / E_i, lambda component)
        givens_insert(R, c0);
        givens_insert(R, c1);
    }
    if constexpr (G > 1) {
        // merge the G private factors: after step `off` every lane holds the factor of its 2*off-lane block
#pragma unroll
        for (int off = 1; off < G; off <<= 1) {
            double Rp[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) Rp[a][b] = (b >= a) ? __shfl_xor(R[a][b], off, G) : 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                double row[4] = {Rp[a][0], Rp[a][1], Rp[a][2], Rp[a][3]};
                givens_insert(R, row);
            }
        }
        // the two partners of a step fold in opposite orders; all lanes adopt lane 0's copy
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = a; b < 4; ++b) R[a][b] = __shfl(R[a][b], 0, G);
    }
    double zX[4] = {0.5, 0.5, 0.5, 0.5};
    double scale = 1.0;  // z_lambda = scale * stored component
    double X0 = 0, X1 = 0, X2 = 0;
    bool done = false;
    for (int it = 0; it < 10; ++it) {
        // forward solve  [E 0; R^T R_C^T] y = z
        double acc[4] = {0, 0, 0, 0};
        for (int64_t q = s0 + g; q < s1; q += G) {
            const double4 r = scr_r[q];
            double2 al = scr_al[q];
            const double yl = (al.y * scale) * al.x;
            if (!done) scr_al[q] = make_double2(al.x, yl);
            acc[0] += r.x * yl; acc[1] += r.y * yl; acc[2] += r.z * yl; acc[3] += r.w * yl;
        }
        if constexpr (G > 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = group_sum<G>(acc[k]);
        }
        double y[4], w[4];
        y[0] = (zX[0] - acc[0]) / R[0][0];
        y[1] = (zX[1] - acc[1] - R[0][1] * y[0]) / R[1][1];
        y[2] = (zX[2] - acc[2] - R[0][2] * y[0] - R[1][2] * y[1]) / R[2][2];
        y[3] = (zX[3] - acc[3] - R[0][3] * y[0] - R[1][3] * y[1] - R[2][3] * y[2]) / R[3][3];
        // back solve  [E R; 0 R_C] w = y
        w[3] = y[3] / R[3][3];
        w[2] = (y[2] - R[2][3] * w[3]) / R[2][2];
        w[1] = (y[1] - R[1][2] * w[2] - R[1][3] * w[3]) / R[1][1];
        w[0] = (y[0] - R[0][1] * w[1] - R[0][2] * w[2] - R[0][3] * w[3]) / R[0][0];
        double part = 0.0;
        for (int64_t q = s0 + g; q < s1; q += G) {
            const double4 r = scr_r[q];
            const double2 al = scr_al[q];
            const double wl = (al.y - (r.x * w[0] + r.y * w[1] + r.z * w[2] + r.w * w[3])) * al.x;
            if (!done) scr_al[q] = make_double2(al.x, wl);
            part += wl * wl;
        }
        if constexpr (G > 1) part = group_sum<G>(part);
        const double nrm2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3] + part;
        const double n0 = w[0] / w[3], n1 = w[1] / w[3], n2 = w[2] / w[3];
        const double change = fmax(fabs(n0 - X0), fmax(fabs(n1 - X1), fabs(n2 - X2)));
        const double size = fmax(fabs(n0), fmax(fabs(n1), fabs(n2)));
        if (!done) {  // a converged group is frozen: its result must not depend on its wave neighbours
            scale = 1.0 / sqrt(nrm2);
#pragma unroll
            for (int k = 0; k < 4; ++k) zX[k] = w[k] * scale;
            X0 = n0; X1 = n1; X2 = n2;
            done = it > 0 && !(change > 1e-14 * size);  // also leaves on NaN
        }
        if (__all(done || !live)) break;  // the shuffles are wave-wide: leave together
    }
    if (live && g == 0) {
        pts[3 * j + 0] = X0;
        pts[3 * j + 1] = X1;
        pts[3 * j + 2] = X2;
    }
}

// ---- round 4: the same algorithm with a group's views ON CHIP and without IEEE divides ---------------------------------------------------
// Where round 3's 106 us went (97 k points, 1e6 observations): not traffic.  A view cost ~25 FP64 divides / square roots (undistort 6,
// reflector 2, two Givens folds 8 x 3), the merge of the G private factors 2 x 4 x 4 x 3 more, every inverse-iteration step 12 —
// each a ~25-35-instruction software sequence: ~9 k VALU instructions per wave of 16 points, issue-bound (24 waves per CU).
//   * every division and square root on the path is a hardware estimate + two Newton steps (tri_rcp / tri_rsq: ~6-9 instructions,
//     <= 1-2 ulp), a Givens rotation needs ONE of them (c = a / h, s = b / h, h = h^2 / h from 1 / sqrt(h^2)), the reciprocals of
//     the 4 x 4 factor's diagonal are formed once per point instead of twice per step;
//   * lane g keeps its views g, g + G, ... in REGISTERS (Householder row r_i, 1 / alpha_i, the lambda component): V views per lane,
//     G V per point (24 at the default G = 4, V = 6: 161 VGPRs, three waves per SIMD); only views beyond that use the 48 B / observation scratch of the first
//     version.  Loops over the register views are unrolled and cut at the wave's largest per-lane count (uniform branch), absent
//     views are all-zero records that change no sum — a point's result still does not depend on its wave neighbours.
__device__ __forceinline__ double tri_rcp(const double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
__device__ __forceinline__ double tri_rsq(const double x) {   // 1 / sqrt(x), x > 0: hardware estimate (~2^-26) + ONE third-order step,
    const double y = __builtin_amdgcn_rsq(x);                   // y (1 + h / 2 + 3 h^2 / 8) with h = 1 - x y^2 (error ~ h^3: full double precision;
    const double h = fma(-(x * y), y, 1.0);                     // ba_chol_persist.hpp's cp_rsqrt) — 6 instructions where two Newton steps took 9
    return fma(y * h, fma(h, 0.375, 0.5), y);
}

__device__ __forceinline__ void undistort5_fast(const double u, const double v, const double *__restrict__ ct, double &uo, double &vo) {
    const double fx = ct[22], cx = ct[23], fy = ct[24], cy = ct[25];
    const double k0 = ct[26], k1 = ct[27], p0 = ct[28], p1 = ct[29], k2 = ct[30];
    const double x0 = (u - cx) * tri_rcp(fx), y0 = (v - cy) * tri_rcp(fy);
    double x = x0, y = y0;
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const double r2 = x * x + y * y;
        const double k_inv = tri_rcp(1.0 + k0 * r2 + k1 * (r2 * r2) + k2 * (r2 * r2 * r2));
        const double xD = 2.0 * p0 * x * y + p1 * (r2 + 2.0 * (x * x));
        const double yD = p0 * (r2 + 2.0 * (y * y)) + 2.0 * p1 * x * y;
        x = (x0 - xD) * k_inv;
        y = (y0 - yD) * k_inv;
    }
    uo = x * fx + cx;
    vo = y * fy + cy;
}

__device__ __forceinline__ void view_rows_fast(const double *__restrict__ P, const double u, const double v, double &inv_alpha, double (&r)[4],
                                               double (&c0)[4], double (&c1)[4]) {
    const double n2 = u * u + v * v + 1.0;
    const double inx = tri_rsq(n2), nx = n2 * inx;
    const double alpha = (u >= 0.0) ? -nx : nx;   // opposite sign of x_1: no cancellation in v_1
    inv_alpha = (u >= 0.0) ? -inx : inx;
    const double v0 = u - alpha, v1 = v, v2 = 1.0;
    const double f = 2.0 * tri_rcp(v0 * v0 + v1 * v1 + v2 * v2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double t = f * (v0 * P[k] + v1 * P[4 + k] + v2 * P[8 + k]);
        r[k] = P[k] - v0 * t;
        c0[k] = P[4 + k] - v1 * t;
        c1[k] = P[8 + k] - v2 * t;
    }
}

// Fold TWO rows into the upper-triangular 4 x 4 factor at once (round 5): per column k a 3-element Householder reflector on
// (R_kk, row0_k, row1_k) — norm, ONE reciprocal square root and one reciprocal, then 7 multiply-adds per remaining column — where two
// Givens folds spent two reciprocal square roots and 8 multiply-adds per column and row: ~110 instead of ~200 instructions per view
// (profiles/r05/tri_sq_counters_a.json: the kernel issues 2 407 vector instructions per wave and is issue-bound to 52 %).  Backward
// stable like the rotations (v_0 = a - alpha with alpha of the opposite sign of a: no cancellation); rows that are zero in a column
// leave it untouched (also keeps 0 / 0 out of an empty factor), so an absent view changes nothing; the factor's diagonal may come out
// negative, which the triangular solves do not mind.
__device__ __forceinline__ void house2_insert(double (&R)[4][4], double (&r0)[4], double (&r1)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double a = R[k][k], b = r0[k], c = r1[k];
        const double s2 = fma(c, c, b * b);
        const bool skip = s2 == 0.0;
        const double n2 = skip ? 1.0 : fma(a, a, s2);
        const double inx = tri_rsq(n2), nx = n2 * inx;
        const double alpha = (a >= 0.0) ? -nx : nx;
        const double v0 = a - alpha;                                  // |v0| = |a| + nx
        const double beta = skip ? 0.0 : tri_rcp(nx * fabs(v0));      // 2 / (v' v) = 1 / (nx |v0|)
        R[k][k] = skip ? a : alpha;
#pragma unroll
        for (int j = k + 1; j < 4; ++j) {
            const double t = beta * fma(c, r1[j], fma(b, r0[j], v0 * R[k][j]));
            R[k][j] = fma(-v0, t, R[k][j]);
            r0[j] = fma(-b, t, r0[j]);
            r1[j] = fma(-c, t, r1[j]);
        }
    }
}

template <int G, int V>
__global__ __launch_bounds__(256) void triangulate_reg_kernel(const int32_t *__restrict__ cam, const double2 *__restrict__ uv,
                                                              const int64_t *__restrict__ start, const double *__restrict__ cam_tab,
                                                              double4 *__restrict__ scr_r, double2 *__restrict__ scr_al,
                                                              double *__restrict__ pts, int64_t n_pts, const int32_t *__restrict__ order) {
    const int64_t gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int g = threadIdx.x & (G - 1);
    const bool live = gid < n_pts;  // whole groups are live or dead; dead groups still take part in the shuffles
    const int64_t jv = live ? gid : n_pts - 1;
    const int64_t j = order ? order[jv] : jv;   // points of equal view counts side by side (tri_order_*_kernel)
    const int64_t s0 = start[j], s1 = live ? start[j + 1] : s0;
    // the wave's largest number of register views per lane: the unrolled loops stop there (uniform)
    int nv = (int)((s1 - s0 - g + G - 1) / G);
    nv = nv < 0 ? 0 : (nv > V ? V : nv);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nv = max(nv, __shfl_xor(nv, off));
    const int nv_w = __builtin_amdgcn_readfirstlane(nv);

    double R[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double rr[V][4], ia[V], lam[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {   // pass 1: undistort, Householder rows, fold the C rows into R
        rr[v][0] = rr[v][1] = rr[v][2] = rr[v][3] = 0.0;
        ia[v] = lam[v] = 0.0;
        if (v >= nv_w) continue;
        const int64_t q = s0 + g + (int64_t)v * G;
        const bool have = q < s1;
        const int64_t qc = have ? q : s0;   // some readable observation for absent views (dead groups: the last point's first)
        const double *ct = cam_tab + (int64_t)cam[qc] * TRI_CAM_STRIDE;
        const double2 m = uv[qc];
        double uu, vv, inv_alpha, r[4], c0[4], c1[4];
        undistort5_fast(m.x, m.y, ct, uu, vv);
        view_rows_fast(ct, uu, vv, inv_alpha, r, c0, c1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rr[v][k] = have ? r[k] : 0.0;
            c0[k] = have ? c0[k] : 0.0;   // a zero row folds as the identity
            c1[k] = have ? c1[k] : 0.0;
        }
        ia[v] = have ? -inv_alpha : 0.0;   // 1 / E_i
        house2_insert(R, c0, c1);
    }
    for (int64_t q = s0 + g + (int64_t)V * G; q < s1; q += G) {   // views beyond the registers: the scratch records of the first version
        const double *ct = cam_tab + (int64_t)cam[q] * TRI_CAM_STRIDE;
        const double2 m = uv[q];
        double uu, vv, inv_alpha, r[4], c0[4], c1[4];
        undistort5_fast(m.x, m.y, ct, uu, vv);
        view_rows_fast(ct, uu, vv, inv_alpha, r, c0, c1);
        scr_r[q] = make_double4(r[0], r[1], r[2], r[3]);
        scr_al[q] = make_double2(-inv_alpha, 0.0);
        house2_insert(R, c0, c1);
    }
    if constexpr (G > 1) {
        // merge the G private factors: after step `off` every lane holds the factor of its 2*off-lane block
#pragma unroll
        for (int off = 1; off < G; off <<= 1) {
            double Rp[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) Rp[a][b] = (b >= a) ? __shfl_xor(R[a][b], off, G) : 0.0;
#pragma unroll
            for (int a = 0; a < 4; a += 2) {   // the partner's rows two at a time
                double row0[4] = {Rp[a][0], Rp[a][1], Rp[a][2], Rp[a][3]};
                double row1[4] = {Rp[a + 1][0], Rp[a + 1][1], Rp[a + 1][2], Rp[a + 1][3]};
                house2_insert(R, row0, row1);
            }
        }
        // the two partners of a step fold in opposite orders; all lanes adopt lane 0's copy
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = a; b < 4; ++b) R[a][b] = __shfl(R[a][b], 0, G);
    }
    const double i0 = tri_rcp(R[0][0]), i1 = tri_rcp(R[1][1]), i2 = tri_rcp(R[2][2]), i3 = tri_rcp(R[3][3]);
    double zX[4] = {0.5, 0.5, 0.5, 0.5};
    double scale = 1.0;  // z_lambda = scale * stored component
    double X0 = 0, X1 = 0, X2 = 0;
    bool done = false;
    for (int it = 0; it < 10; ++it) {
        // forward solve  [E 0; R^T R_C^T] y = z
        double acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (v >= nv_w) break;
            const double yl = (lam[v] * scale) * ia[v];
            lam[v] = done ? lam[v] : yl;
            acc[0] += rr[v][0] * yl; acc[1] += rr[v][1] * yl; acc[2] += rr[v][2] * yl; acc[3] += rr[v][3] * yl;
        }
        for (int64_t q = s0 + g + (int64_t)V * G; q < s1; q += G) {
            const double4 r = scr_r[q];
            double2 al = scr_al[q];
            const double yl = (al.y * scale) * al.x;
            if (!done) scr_al[q] = make_double2(al.x, yl);
            acc[0] += r.x * yl; acc[1] += r.y * yl; acc[2] += r.z * yl; acc[3] += r.w * yl;
        }
        if constexpr (G > 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = group_sum<G>(acc[k]);
        }
        double y[4], w[4];
        y[0] = (zX[0] - acc[0]) * i0;
        y[1] = (zX[1] - acc[1] - R[0][1] * y[0]) * i1;
        y[2] = (zX[2] - acc[2] - R[0][2] * y[0] - R[1][2] * y[1]) * i2;
        y[3] = (zX[3] - acc[3] - R[0][3] * y[0] - R[1][3] * y[1] - R[2][3] * y[2]) * i3;
        // back solve  [E R; 0 R_C] w = y
        w[3] = y[3] * i3;
        w[2] = (y[2] - R[2][3] * w[3]) * i2;
        w[1] = (y[1] - R[1][2] * w[2] - R[1][3] * w[3]) * i1;
        w[0] = (y[0] - R[0][1] * w[1] - R[0][2] * w[2] - R[0][3] * w[3]) * i0;
        double part = 0.0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (v >= nv_w) break;
            const double wl = (lam[v] - (rr[v][0] * w[0] + rr[v][1] * w[1] + rr[v][2] * w[2] + rr[v][3] * w[3])) * ia[v];
            lam[v] = done ? lam[v] : wl;
            part += wl * wl;
        }
        for (int64_t q = s0 + g + (int64_t)V * G; q < s1; q += G) {
            const double4 r = scr_r[q];
            const double2 al = scr_al[q];
            const double wl = (al.y - (r.x * w[0] + r.y * w[1] + r.z * w[2] + r.w * w[3])) * al.x;
            if (!done) scr_al[q] = make_double2(al.x, wl);
            part += wl * wl;
        }
        if constexpr (G > 1) part = group_sum<G>(part);
        const double nrm2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3] + part;
        const double iw3 = tri_rcp(w[3]);
        const double n0 = w[0] * iw3, n1 = w[1] * iw3, n2 = w[2] * iw3;
        const double change = fmax(fabs(n0 - X0), fmax(fabs(n1 - X1), fabs(n2 - X2)));
        const double size = fmax(fabs(n0), fmax(fabs(n1), fabs(n2)));
        if (!done) {  // a converged group is frozen: its result must not depend on its wave neighbours
            scale = tri_rsq(nrm2);
#pragma unroll
            for (int k = 0; k < 4; ++k) zX[k] = w[k] * scale;
            X0 = n0; X1 = n1; X2 = n2;
            done = it > 0 && !(change > 1e-14 * size);  // also leaves on NaN
        }
        if (__all(done || !live)) break;  // the shuffles are wave-wide: leave together
    }
    if (live && g == 0) {
        pts[3 * j + 0] = X0;
        pts[3 * j + 1] = X1;
        pts[3 * j + 2] = X2;
    }
}

// ---- visiting order: points sorted by their number of views (counting sort, 256 buckets) ------------------------------------------------
// A wave of triangulate_reg_kernel runs as many register-view slots as its busiest lane needs; with the points in table order a wave
// mixes 2-view and 22-view points and most lanes idle through most slots (rig-32: mean 3.0 slots needed, 4.1 run).  Walking the points
// in order of their view count makes a wave's groups alike.  Built once per set of observations, on the stream of the first run.
__global__ __launch_bounds__(256) void tri_order_count_kernel(const int64_t *__restrict__ start, const int64_t n_pts, int32_t *__restrict__ hist) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pts) return;
    const int64_t v = start[j + 1] - start[j];
    atomicAdd(hist + (int)(v < 0 ? 0 : (v > 255 ? 255 : v)), 1);
}
__global__ __launch_bounds__(256) void tri_order_scan_kernel(int32_t *__restrict__ hist) {   // one workgroup: hist[256] -> exclusive offsets in hist[256 .. 511]
    __shared__ int32_t sm[256];
    const int t = threadIdx.x;
    sm[t] = hist[t];
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int32_t add = t >= off ? sm[t - off] : 0;
        __syncthreads();
        sm[t] += add;
        __syncthreads();
    }
    hist[256 + t] = sm[t] - hist[t];
}
__global__ __launch_bounds__(256) void tri_order_scatter_kernel(const int64_t *__restrict__ start, const int64_t n_pts, int32_t *__restrict__ hist, int32_t *__restrict__ order) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pts) return;
    const int64_t v = start[j + 1] - start[j];
    const int pos = atomicAdd(hist + 256 + (int)(v < 0 ? 0 : (v > 255 ? 255 : v)), 1);
    order[pos] = (int32_t)j;
}

// ---- the grouping in front of the triangulation, on the device (round 5) ---------------------------------------------------------------------
// CameraSet.multi_cam_triangulate (cameras/camera_set.py:371-378) finds, with two np.unique calls over the (image, key...) columns of the
// detection table, the features seen by more than one camera, keeps their rows in table order and builds start_ind = cumulative counts
// in order of first appearance; nb_triangulate_full then takes CONSECUTIVE rows per feature.  On a table grouped by feature (what
// TargetDetection.get_data returns, and the only kind for which the reference's consecutive slices are one feature each) that is:
//   count[f]  rows per feature (dense feature ids);  head[i] = row i starts a run of its feature;
//   keep[i] = count[feature[i]] >= 2;  position among the kept rows and index of the kept run = an exclusive scan of (keep, keep & head);
//   kept rows scattered in order, start[run] = position of the run's first row.
// The host grouping took 0.37 s for 1e6 observations (NumPy's sort-based unique); this is four launches.  Whether the table IS grouped
// (runs == features present) is checked on the device; the caller falls back to the host grouping — the reference's semantics for
// any table — otherwise.
struct TriGroupArgs {
    const int32_t *cam, *feat;       // n rows: camera index, dense feature id in [0, n_features)
    const double2 *uv;
    int32_t *count;                  // n_features, zeroed
    uint64_t *block_sums;            // one packed (kept rows | kept runs << 32) per 1024-row block; exclusive-scanned in place
    int64_t *totals;                 // [0] kept rows, [1] kept runs, [2] runs in the table, [3] features present; zeroed
    int32_t *cam_out;
    double2 *uv_out;
    int64_t *start;                  // kept runs + 1
    int64_t n, n_features;
    int32_t n_blocks;
};
constexpr int TRI_GROUP_BLOCK = 1024;

__global__ __launch_bounds__(256) void tri_group_count_kernel(const TriGroupArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int32_t f = a.feat[i];
    if ((uint32_t)f >= (uint64_t)a.n_features) return;   // not a feature id (the caller's contract): the row is dropped, nothing is addressed with it
    if (atomicAdd(a.count + f, 1) == 0) atomicAdd(reinterpret_cast<unsigned long long *>(a.totals + 3), 1ull);   // a feature's first row
    if (i == 0 || a.feat[i - 1] != f) atomicAdd(reinterpret_cast<unsigned long long *>(a.totals + 2), 1ull);     // a run's first row
}

__device__ __forceinline__ uint64_t tri_group_flags(const TriGroupArgs &a, const int64_t i) {
    if (i >= a.n) return 0;
    const int32_t f = a.feat[i];
    if ((uint32_t)f >= (uint64_t)a.n_features) return 0;
    const bool keep = a.count[f] >= 2;
    const bool head = i == 0 || a.feat[i - 1] != f;
    return (keep ? 1ull : 0ull) | ((keep && head) ? (1ull << 32) : 0ull);
}
// workgroup-wide inclusive scan of one packed word per thread (1024 threads), through LDS
__device__ __forceinline__ uint64_t tri_group_scan(uint64_t v, uint64_t *sm) {
    const int tid = threadIdx.x;
    sm[tid] = v;
    __syncthreads();
    for (int off = 1; off < TRI_GROUP_BLOCK; off <<= 1) {
        const uint64_t add = tid >= off ? sm[tid - off] : 0;
        __syncthreads();
        sm[tid] += add;
        __syncthreads();
    }
    return sm[tid];
}
__global__ __launch_bounds__(TRI_GROUP_BLOCK) void tri_group_blocksum_kernel(const TriGroupArgs a) {
    __shared__ uint64_t sm[TRI_GROUP_BLOCK];
    const uint64_t inc = tri_group_scan(tri_group_flags(a, (int64_t)blockIdx.x * TRI_GROUP_BLOCK + threadIdx.x), sm);
    if (threadIdx.x == TRI_GROUP_BLOCK - 1) a.block_sums[blockIdx.x] = inc;
}
__global__ __launch_bounds__(TRI_GROUP_BLOCK) void tri_group_scan_sums_kernel(const TriGroupArgs a) {   // one workgroup: exclusive scan of the block sums
    __shared__ uint64_t sm[TRI_GROUP_BLOCK];
    uint64_t carry = 0;
    for (int b0 = 0; b0 < a.n_blocks; b0 += TRI_GROUP_BLOCK) {
        const int b = b0 + threadIdx.x;
        const uint64_t v = b < a.n_blocks ? a.block_sums[b] : 0;
        const uint64_t inc = tri_group_scan(v, sm);
        if (b < a.n_blocks) a.block_sums[b] = carry + inc - v;
        const uint64_t total = sm[TRI_GROUP_BLOCK - 1];
        __syncthreads();
        carry += total;
    }
    if (threadIdx.x == 0) {
        a.totals[0] = (int64_t)(carry & 0xffffffffull);
        a.totals[1] = (int64_t)(carry >> 32);
        a.start[carry >> 32] = (int64_t)(carry & 0xffffffffull);   // start[n_runs] = kept rows
    }
}
__global__ __launch_bounds__(TRI_GROUP_BLOCK) void tri_group_scatter_kernel(const TriGroupArgs a) {
    __shared__ uint64_t sm[TRI_GROUP_BLOCK];
    const int64_t i = (int64_t)blockIdx.x * TRI_GROUP_BLOCK + threadIdx.x;
    const uint64_t fl = tri_group_flags(a, i);
    const uint64_t exc = a.block_sums[blockIdx.x] + tri_group_scan(fl, sm) - fl;
    if (fl & 1ull) {
        const int64_t pos = (int64_t)(exc & 0xffffffffull);
        a.cam_out[pos] = a.cam[i];
        a.uv_out[pos] = a.uv[i];
        if (fl >> 32) a.start[exc >> 32] = pos;
    }
}

}  // namespace pcs
