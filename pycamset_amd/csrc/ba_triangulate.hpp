// ba_triangulate.hpp — batched n-view triangulation.
//
// Reference: nb_triangulate_full / nb_triangulate_nviews (compiled_helpers.py:609-663), front end
// CameraSet.multi_cam_triangulate (cameras/camera_set.py:343-402).  Per 3-D point seen by n >= 2
// cameras the reference undistorts every observation (5 fixed-point iterations, ch:409-431), stacks
//     M = [ P_i | 0 .. -x_i .. 0 ]   (3n x (4+n)),  x_i = (u_i, v_i, 1) undistorted pixels
// and returns the right singular vector of the smallest singular value (LAPACK SVD), X[:3] / X[3].
//
// Here one lane owns one point.  The lambda block of M^T M is diagonal (d_i = |x_i|^2), so that
// singular vector is the solution of a 4x4 nonlinear eigenproblem (secular equation)
//     G(mu) X = mu X,   G(mu) = sum_i [ P_i^T P_i - b_i b_i^T / (d_i - mu) ],   b_i = P_i^T x_i,
// with mu = sigma_min^2.  theta(mu) = smallest eigenvalue of G(mu) (4x4 cyclic Jacobi in registers);
// Newton on theta(mu) - mu with theta'(mu) = -sum_i (b_i . X)^2 / (d_i - mu)^2 converges in 3-5
// steps from mu = 0.  Agreement with the LAPACK SVD: <= 1e-13 relative on the synthetic rigs
// (tests/test_gpu_triangulate.py).
#pragma once
#include <hip/hip_runtime.h>

namespace pcs {

constexpr int TRI_CAM_STRIDE = 32;  // P 12 | PtP upper 10 | fx cx fy cy | k0 k1 p0 p1 k2 | pad

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

// smallest eigenpair of a symmetric 4x4 (upper triangle g[10]: 00 01 02 03 11 12 13 22 23 33), cyclic Jacobi
__device__ __forceinline__ void smallest_eig4(const double (&g)[10], double &lam, double (&vec)[4]) {
    double a[4][4] = {{g[0], g[1], g[2], g[3]}, {g[1], g[4], g[5], g[6]}, {g[2], g[5], g[7], g[8]}, {g[3], g[6], g[8], g[9]}};
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[0][3] * a[0][3] + a[1][2] * a[1][2] + a[1][3] * a[1][3] + a[2][3] * a[2][3];
        const double dia = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2] + a[3][3] * a[3][3];
        if (!(off > 1e-34 * dia)) break;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const double apq = a[p][q];
                if (apq != 0.0) {
                    const double tau = (a[q][q] - a[p][p]) / (2.0 * apq);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {  // A <- A J
                        const double akp = a[k][p], akq = a[k][q];
                        a[k][p] = c * akp - s * akq;
                        a[k][q] = s * akp + c * akq;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {  // A <- J^T A
                        const double apk = a[p][k], aqk = a[q][k];
                        a[p][k] = c * apk - s * aqk;
                        a[q][k] = s * apk + c * aqk;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double vkp = V[k][p], vkq = V[k][q];
                        V[k][p] = c * vkp - s * vkq;
                        V[k][q] = s * vkp + c * vkq;
                    }
                }
            }
        }
    }
    int m = 0;
    lam = a[0][0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (a[k][k] < lam) { lam = a[k][k]; m = k; }
#pragma unroll
    for (int k = 0; k < 4; ++k) vec[k] = (m == 0) ? V[k][0] : (m == 1) ? V[k][1] : (m == 2) ? V[k][2] : V[k][3];
}

// one thread = one point; observations of point j are rows [start[j], start[j+1])
__global__ __launch_bounds__(256) void triangulate_kernel(const int32_t *__restrict__ cam, const double2 *__restrict__ uv,
                                                          const int64_t *__restrict__ start, const double *__restrict__ cam_tab,
                                                          double2 *__restrict__ scratch, double *__restrict__ pts, int64_t n_pts) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pts) return;
    const int64_t s0 = start[j], s1 = start[j + 1];
    double A[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t r = s0; r < s1; ++r) {  // pass 1: undistort once, accumulate sum P^T P
        const double *ct = cam_tab + (int64_t)cam[r] * TRI_CAM_STRIDE;
        const double2 m = uv[r];
        double uu, vv;
        undistort5(m.x, m.y, ct, uu, vv);
        scratch[r] = make_double2(uu, vv);
#pragma unroll
        for (int k = 0; k < 10; ++k) A[k] += ct[12 + k];
    }
    double mu = 0.0, X[4] = {0, 0, 0, 1};
    for (int it = 0; it < 8; ++it) {
        double G[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) G[k] = A[k];
        for (int64_t r = s0; r < s1; ++r) {
            const double *P = cam_tab + (int64_t)cam[r] * TRI_CAM_STRIDE;
            const double2 x = scratch[r];
            const double b0 = P[0] * x.x + P[4] * x.y + P[8], b1 = P[1] * x.x + P[5] * x.y + P[9];
            const double b2 = P[2] * x.x + P[6] * x.y + P[10], b3 = P[3] * x.x + P[7] * x.y + P[11];
            const double w = 1.0 / (x.x * x.x + x.y * x.y + 1.0 - mu);
            G[0] -= b0 * b0 * w; G[1] -= b0 * b1 * w; G[2] -= b0 * b2 * w; G[3] -= b0 * b3 * w;
            G[4] -= b1 * b1 * w; G[5] -= b1 * b2 * w; G[6] -= b1 * b3 * w;
            G[7] -= b2 * b2 * w; G[8] -= b2 * b3 * w; G[9] -= b3 * b3 * w;
        }
        double theta;
        smallest_eig4(G, theta, X);
        double dtheta = 0.0;
        for (int64_t r = s0; r < s1; ++r) {
            const double *P = cam_tab + (int64_t)cam[r] * TRI_CAM_STRIDE;
            const double2 x = scratch[r];
            const double bx = (P[0] * x.x + P[4] * x.y + P[8]) * X[0] + (P[1] * x.x + P[5] * x.y + P[9]) * X[1] +
                              (P[2] * x.x + P[6] * x.y + P[10]) * X[2] + (P[3] * x.x + P[7] * x.y + P[11]) * X[3];
            const double w = 1.0 / (x.x * x.x + x.y * x.y + 1.0 - mu);
            dtheta -= bx * bx * w * w;
        }
        const double mu_new = mu - (theta - mu) / (dtheta - 1.0);
        const bool done = fabs(mu_new - mu) <= 1e-15 * fabs(A[0] + A[4] + A[7] + A[9]);
        mu = mu_new;
        if (done && it > 0) break;
    }
    pts[3 * j + 0] = X[0] / X[3];
    pts[3 * j + 1] = X[1] / X[3];
    pts[3 * j + 2] = X[2] / X[3];
}

}  // namespace pcs
