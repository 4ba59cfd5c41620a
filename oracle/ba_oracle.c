/*
 * ba_oracle.c — CPU restatement of pyCamSet's bundle-adjustment cost / Jacobian hot path.
 *
 * TEST INFRASTRUCTURE.  This file is the *checker* (and the timed `cpu_baseline` "port" in
 * bench.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product path (pycamset_amd + libpcs_hip.so) never links, imports or calls it.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_golden.py) against
 * golden vectors produced by running the reference's own Python sources in the build container
 * (tests/golden/make_golden.py; numba absent -> njit = identity, bodies run as IEEE-754 CPython).
 *
 * The algorithm is kept per-detection exactly like the reference's generated numba code
 * (Rodrigues + dRodrigues re-evaluated for every detection, dense 2xP block per detection);
 * nothing is hoisted.  Citations are file:line under /root/reference/pyCamSet/optimisation/
 *   fbi = function_block_implementations.py   ch = compiled_helpers.py
 *   afb = abstract_function_blocks.py         mm = matmul_map.py
 *
 * Integer powers: the reference writes `z**7`; numba lowers integer powers to multiply chains,
 * CPython calls libm pow().  ipow() below multiplies; define ORC_LIBM_POW to call pow() instead.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_CHAIN_TEMPLATE 0 /* projection + extrinsic3D + template_points      template_handler.py:152 */
#define ORC_CHAIN_SELF 1     /* projection + extrinsic3D + rigidTform3d + free_point  standard_bundle_handler.py:182 */
#define ORC_CHAIN_FREE 2     /* projection + extrinsic3D + free_point           free_point_handler.py:143 */

static inline double ipow(double x, int n) {
#ifdef ORC_LIBM_POW
    return pow(x, (double)n);
#else
    double r = x;
    for (int i = 1; i < n; ++i) r *= x;
    return r;
#endif
}

/* ch:197-235  numba_flat_rodrigues_INPLACE: rotation vector -> row-major 3x3 */
void orc_rodrigues(const double *r, double *R) {
    double theta = sqrt(ipow(r[0], 2) + ipow(r[1], 2) + ipow(r[2], 2)); /* ch:205 */
    if (theta < 1e-10) {                                               /* ch:206-211 */
        for (int i = 0; i < 9; ++i) R[i] = 0.0;
        R[0] = 1.0; R[4] = 1.0; R[8] = 1.0;
        return;
    }
    double scalar = 1.0 / theta;       /* ch:213 */
    double s2 = ipow(scalar, 2);       /* ch:214 */
    double ct = cos(theta);            /* ch:215 */
    double st = sin(theta) * scalar;   /* ch:216 */
    for (int i = 0; i < 3; ++i)        /* ch:219-222 */
        for (int j = i; j < 3; ++j) {
            R[3 * i + j] = r[i] * r[j];
            R[3 * j + i] = r[i] * r[j];
        }
    double f = (1.0 - ct) * s2;        /* ch:225 */
    for (int i = 0; i < 9; ++i) R[i] *= f;
    R[0] += ct; R[4] += ct; R[8] += ct; /* ch:226-228 */
    R[1] -= r[2] * st;                 /* ch:229-234 */
    R[3] += r[2] * st;
    R[2] += r[1] * st;
    R[6] -= r[1] * st;
    R[5] -= r[0] * st;
    R[7] += r[0] * st;
}

/* ch:237-286  numba_rodrigues_jac: out[a*9+k] = d R_flat[k] / d r_a */
void orc_rodrigues_jac(const double *r, double *out) {
    double theta = sqrt(ipow(r[0], 2) + ipow(r[1], 2) + ipow(r[2], 2)); /* ch:244 */
    if (theta < 1e-10) {                                               /* ch:246-254 */
        for (int i = 0; i < 27; ++i) out[i] = 0.0;
        out[5] = -1; out[15] = -1; out[19] = -1;
        out[7] = 1; out[11] = 1; out[21] = 1;
        return;
    }
    double i_theta = (theta == 0) ? 0 : 1 / theta; /* ch:256 */
    double ct = cos(theta);
    double ct_1 = 1 - ct;
    double st = sin(theta);
    double x = r[0] * i_theta, y = r[1] * i_theta, z = r[2] * i_theta; /* ch:262 */
    double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z}; /* ch:264 */
    double r_x[9] = {0, -z, y, z, 0, -x, -y, x, 0};                                   /* ch:265-267 */
    double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double drrt[27] = {x + x, y, z, y, 0, 0, z, 0, 0,                                 /* ch:271-273 */
                       0, x, 0, x, y + y, z, 0, z, 0,
                       0, 0, x, 0, 0, y, x, y, z + z};
    double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,                                  /* ch:275-277 */
                         0, 0, 1, 0, 0, 0, -1, 0, 0,
                         0, -1, 0, 1, 0, 0, 0, 0, 0};
    double xyz[3] = {x, y, z};
    for (int i = 0; i < 3; ++i) {                                                     /* ch:279-286 */
        double ri = xyz[i];
        double a0 = -st * ri;
        double a1 = (st - 2 * ct_1 * i_theta) * ri;
        double a2 = ct_1 * i_theta;
        double a3 = (ct - st * i_theta) * ri;
        double a4 = st * i_theta;
        for (int k = 0; k < 9; ++k)
            out[i * 9 + k] = a0 * eye[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
    }
}

/* ch:288-301  n_e4x4_flat_INPLACE: 6-vector -> [R row-major 9 | t 3] */
void orc_e4x4_flat(const double *p, double *T) {
    orc_rodrigues(p, T);
    T[9] = p[3]; T[10] = p[4]; T[11] = p[5];
}

/* ch:357-370  n_htform_prealloc: out = R * pt + t */
void orc_htform(const double *pt, const double *T, double *out) {
    for (int j = 0; j < 3; ++j)
        out[j] = pt[0] * T[3 * j + 0] + pt[1] * T[3 * j + 1] + pt[2] * T[3 * j + 2];
    out[0] += T[9]; out[1] += T[10]; out[2] += T[11];
}

/* fbi:27-47  projection.compute_fun; params = [fx,px,fy,py,k0,k1,p0,p1,k2] */
void orc_projection_fun(const double *params, const double *inp, double *output) {
    double x = inp[0], y = inp[1], inv_z = 1 / inp[2];       /* fbi:30 */
    double u = (params[0] * x + params[1] * inp[2]) * inv_z; /* fbi:32 */
    double v = (params[2] * y + params[3] * inp[2]) * inv_z; /* fbi:33 */
    const double *k = params + 4;                            /* fbi:34 */
    x = (u - params[1]) / params[0];                         /* fbi:35 */
    y = (v - params[3]) / params[2];
    double r2 = ipow(x, 2) + ipow(y, 2);                     /* fbi:36 */
    double kup = (1 + k[0] * r2 + k[1] * ipow(r2, 2) + k[4] * ipow(r2, 3)); /* fbi:37 */
    double xD = x * kup;                                     /* fbi:39-40 */
    double yD = y * kup;
    xD += 2 * k[2] * x * y + k[3] * (r2 + 2 * ipow(x, 2));   /* fbi:42 */
    yD += k[2] * (r2 + 2 * ipow(y, 2)) + 2 * k[3] * x * y;   /* fbi:43 */
    output[0] = xD * params[0] + params[1];                  /* fbi:45-46 */
    output[1] = yD * params[2] + params[3];
}

/* fbi:50-140  projection.compute_jac: 2x12 row-major,
 * columns [fx,px,fy,py,k0,k1,p0,p1,k2, xw,yw,zw] (fbi:135-137) */
void orc_projection_jac(const double *params, const double *inp, double *output) {
    double f_x = params[0], f_y = params[2];
    double k_0 = params[4], k_1 = params[5], p_0 = params[6], p_1 = params[7], k_2 = params[8];
    double x = inp[0], y = inp[1], z = inp[2];
    double x2 = ipow(x, 2), y2 = ipow(y, 2);
    double s = x2 + y2; /* (x**2 + y**2) */
    double z2 = ipow(z, 2), z3 = ipow(z, 3), z4 = ipow(z, 4), z5 = ipow(z, 5), z6 = ipow(z, 6), z7 = ipow(z, 7), z8 = ipow(z, 8);
    double s2 = ipow(s, 2), s3 = ipow(s, 3);

    double dxdp_x = 1, dxdp_y = 0, dxdf_y = 0;                                               /* fbi:58-67 */
    double dxdf_x = (x * (k_0 * z4 * s + k_1 * z2 * s2 + k_2 * s3 + z6)
                     + z5 * (2 * p_0 * x * y + p_1 * (3 * x2 + y2))) / z7;                   /* fbi:62-65 */
    double dxdk_0 = f_x * x * s / z3;                                                        /* fbi:69 */
    double dxdk_1 = f_x * x * s2 / z5;                                                       /* fbi:71 */
    double dxdk_2 = f_x * x * s3 / z7;                                                       /* fbi:73 */
    double dxdp_0 = 2 * f_x * x * y / z2;                                                    /* fbi:75 */
    double dxdp_1 = f_x * (3 * x2 + y2) / z2;                                                /* fbi:77 */
    double dxdxw = f_x * (k_0 * z4 * s + k_1 * z2 * s2 + k_2 * s3
                          + 2 * x2 * (k_0 * z4 + 2 * k_1 * z2 * s + 3 * k_2 * s2)
                          + z6 + 2 * z5 * (p_0 * y + 3 * p_1 * x)) / z7;                     /* fbi:79-84 */
    double dxdyw = 2 * f_x * (x * y * (k_0 * z4 + 2 * k_1 * z2 * s + 3 * k_2 * s2)
                              + z5 * (p_0 * x + p_1 * y)) / z7;                              /* fbi:86-88 */
    double dxdzw = -f_x * (4 * p_0 * x * y * z5 + 2 * p_1 * z5 * (3 * x2 + y2)
                           + 2 * x * s * (k_0 * z4 + 2 * k_1 * z2 * s + 3 * k_2 * s2)
                           + x * (k_0 * z4 * s + k_1 * z2 * s2 + k_2 * s3 + z6)) / z8;       /* fbi:90-97 */
    double dydp_x = 0, dydp_y = 1, dydf_x = 0;                                               /* fbi:99-103 */
    double dydf_y = (y * (k_0 * z4 * s + k_1 * z2 * s2 + k_2 * s3 + z6)
                     + z5 * (p_0 * (x2 + 3 * y2) + 2 * p_1 * x * y)) / z7;                   /* fbi:105-108 */
    double dydk_0 = f_y * y * s / z3;                                                        /* fbi:110 */
    double dydk_1 = f_y * y * s2 / z5;                                                       /* fbi:112 */
    double dydk_2 = f_y * y * s3 / z7;                                                       /* fbi:114 */
    double dydp_0 = f_y * (x2 + 3 * y2) / z2;                                                /* fbi:116 */
    double dydp_1 = 2 * f_y * x * y / z2;                                                    /* fbi:118 */
    double dydxw = 2 * f_y * (x * y * (k_0 * z4 + 2 * k_1 * z2 * s + 3 * k_2 * s2)
                              + z5 * (p_0 * x + p_1 * y)) / z7;                              /* fbi:120 */
    double dydyw = f_y * (k_0 * z4 * s + k_1 * z2 * s2 + k_2 * s3
                          + 2 * y2 * (k_0 * z4 + 2 * k_1 * z2 * s + 3 * k_2 * s2)
                          + z6 + 2 * z5 * (3 * p_0 * y + p_1 * x)) / z7;                     /* fbi:122-127 */
    double dydzw = -f_y * (2 * p_0 * z5 * (x2 + 3 * y2) + 4 * p_1 * x * y * z5
                           + 2 * y * s * (k_0 * z4 + 2 * k_1 * z2 * s + 3 * k_2 * s2)
                           + y * (k_0 * z4 * s + k_1 * z2 * s2 + k_2 * s3 + z6)) / z8;       /* fbi:129-134 */
    double d[24] = {dxdf_x, dxdp_x, dxdf_y, dxdp_y, dxdk_0, dxdk_1, dxdp_0, dxdp_1, dxdk_2, dxdxw, dxdyw, dxdzw,
                    dydf_x, dydp_x, dydf_y, dydp_y, dydk_0, dydk_1, dydp_0, dydp_1, dydk_2, dydxw, dydyw, dydzw};
    for (int i = 0; i < 24; ++i) output[i] = d[i];                                           /* fbi:135-139 */
}

/* fbi:150-155  rigidTform3d.compute_fun (= extrinsic3D, template_points) */
void orc_rigid_fun(const double *params, const double *inp, double *output) {
    double T[12];
    orc_e4x4_flat(params, T);
    orc_htform(inp, T, output);
}

/* fbi:157-182  rigidTform3d.compute_jac: 3x9 row-major [d/dr(3) | d/dt = I | d/dinp = R] */
void orc_rigid_jac(const double *params, const double *inp, double *output) {
    double memory[27];
    orc_rodrigues_jac(params, memory);   /* fbi:160 */
    for (int i = 0; i < 27; ++i) output[i] = 0;
    for (int op = 0; op < 3; ++op)       /* fbi:163-170 */
        for (int a = 0; a < 3; ++a)
            output[op * 9 + a] = memory[9 * a + op * 3 + 0] * inp[0] + memory[9 * a + op * 3 + 1] * inp[1]
                                 + memory[9 * a + op * 3 + 2] * inp[2];
    output[0 * 9 + 3] = 1; output[1 * 9 + 4] = 1; output[2 * 9 + 5] = 1; /* fbi:172-174 */
    double T[12];
    orc_e4x4_flat(params, T);            /* fbi:177 */
    for (int op = 0; op < 3; ++op)       /* fbi:178-181 */
        for (int c = 0; c < 3; ++c) output[op * 9 + 6 + c] = T[c + 3 * op];
}

/* fbi:194-211  template_points.compute_jac: 3x6 row-major [d/dr(3) | I] */
void orc_template_jac(const double *params, const double *inp, double *output) {
    double memory[27];
    orc_rodrigues_jac(params, memory);
    for (int i = 0; i < 18; ++i) output[i] = 0;
    for (int op = 0; op < 3; ++op)
        for (int a = 0; a < 3; ++a)
            output[op * 6 + a] = memory[9 * a + op * 3 + 0] * inp[0] + memory[9 * a + op * 3 + 1] * inp[1]
                                 + memory[9 * a + op * 3 + 2] * inp[2];
    output[0 * 6 + 3] = 1; output[1 * 6 + 4] = 1; output[2 * 6 + 5] = 1;
}

/* fbi:226-231 / fbi:234-240  free_point */
void orc_free_fun(const double *params, double *output) {
    output[0] = params[0]; output[1] = params[1]; output[2] = params[2];
}
void orc_free_jac(double *output) {
    for (int i = 0; i < 9; ++i) output[i] = 0;
    output[0] = 1; output[4] = 1; output[8] = 1;
}

int orc_chain_P(int chain) { return chain == ORC_CHAIN_TEMPLATE ? 21 : chain == ORC_CHAIN_SELF ? 24 : chain == ORC_CHAIN_FREE ? 18 : -1; }

/*
 * The generated `matflow(output_block, write_data)` (generator mm:147-243): the chain rule
 *   J = J_proj * J_extr * J_pose [* J_point]
 * with each block Jacobian embedded in an identity (mm:182-195), multiplied symbolically with
 * exact 0/1 entries treated as structural (mm:55-87), rows param_len, param_len+1 emitted
 * (mm:234-242) after `write_data[:] = 0` (mm:231).  Sums are written in increasing-k order.
 *
 * A = projection 2x12 at ob[0..24);  E = extrinsic3D 3x9 at ob[24..51);
 * chain T: Q = template_points 3x6 at ob[51..69)
 * chain S: G = rigidTform3d 3x9 at ob[51..78), F = free_point 3x3 at ob[78..87)
 * chain F: F = free_point 3x3 at ob[51..60)
 */
static void orc_matflow(int chain, const double *ob, double *w /* 2 x P row-major */) {
    const int P = orc_chain_P(chain);
    const double *A = ob, *E = ob + 24, *B2 = ob + 51;
    for (int i = 0; i < 2 * P; ++i) w[i] = 0; /* mm:231 */
    for (int i = 0; i < 2; ++i) {
        double *wr = w + i * P;
        const double *Ai = A + 12 * i, *Ax = Ai + 9;
        /* intrinsics: copied; the structural 0 entries stay 0 and the structural 1 is written as 1 */
        for (int j = 0; j < 9; ++j) wr[j] = Ai[j];
        /* extrinsic rotation: (A_x . E_r) */
        for (int a = 0; a < 3; ++a) wr[9 + a] = (Ax[0] * E[0 * 9 + a] + Ax[1] * E[1 * 9 + a] + Ax[2] * E[2 * 9 + a]);
        /* extrinsic translation: E_t = I -> A_x */
        for (int a = 0; a < 3; ++a) wr[12 + a] = Ax[a];
        /* S_c = (A_x . R_e)[c] */
        double S[3];
        for (int c = 0; c < 3; ++c) S[c] = (Ax[0] * E[0 * 9 + 6 + c] + Ax[1] * E[1 * 9 + 6 + c] + Ax[2] * E[2 * 9 + 6 + c]);
        if (chain == ORC_CHAIN_TEMPLATE) {
            const double *Q = B2; /* 3x6 */
            for (int a = 0; a < 3; ++a) wr[15 + a] = (S[0] * Q[0 * 6 + a] + S[1] * Q[1 * 6 + a] + S[2] * Q[2 * 6 + a]);
            for (int a = 0; a < 3; ++a) wr[18 + a] = S[a];
        } else if (chain == ORC_CHAIN_SELF) {
            const double *G = B2; /* 3x9 */
            for (int a = 0; a < 3; ++a) wr[15 + a] = (S[0] * G[0 * 9 + a] + S[1] * G[1 * 9 + a] + S[2] * G[2 * 9 + a]);
            for (int a = 0; a < 3; ++a) wr[18 + a] = S[a];
            /* free_point Jacobian is the identity -> (S . R_p) */
            for (int j = 0; j < 3; ++j) wr[21 + j] = (S[0] * G[0 * 9 + 6 + j] + S[1] * G[1 * 9 + 6 + j] + S[2] * G[2 * 9 + 6 + j]);
        } else {
            for (int j = 0; j < 3; ++j) wr[15 + j] = S[j];
        }
    }
}

/* Global parameter-string offsets (make_param_struct afb:777-820): unique param groups in block
 * order, group start = sum of n_params * count(link type); counts = max index + 1 (afb:793-795). */
typedef struct { int64_t intr, extr, pose, point; } orc_offsets;
static orc_offsets orc_param_offsets(int chain, int64_t C, int64_t I) {
    orc_offsets o;
    o.intr = 0; o.extr = 9 * C; o.pose = 15 * C;
    o.point = (chain == ORC_CHAIN_SELF) ? 15 * C + 6 * I : 15 * C;
    return o;
}

/* One detection through the chain, blocks evaluated last -> first exactly like the generated
 * full_loss / full_jac bodies (afb:246-267, afb:446-460, afb:586-589). */
static void orc_eval_one(int chain, const double *datum, const double *prm, orc_offsets off, const double *tmpl,
                         double *resid /*2 or NULL*/, double *jrow /*2xP or NULL*/) {
    const int64_t c = (int64_t)datum[0], im = (int64_t)datum[1], key = (int64_t)datum[2]; /* afb:214, afb:375 */
    const double *p_intr = prm + off.intr + 9 * c;
    const double *p_extr = prm + off.extr + 6 * c;
    const double *p_pose = prm + off.pose + 6 * im;
    const double *p_pt = prm + off.point + 3 * key;
    double ob[87];
    double inp[3], out[3];
    if (chain == ORC_CHAIN_TEMPLATE) {
        inp[0] = tmpl[3 * key]; inp[1] = tmpl[3 * key + 1]; inp[2] = tmpl[3 * key + 2]; /* afb:374-375 */
        if (jrow) orc_template_jac(p_pose, inp, ob + 51);
        orc_rigid_fun(p_pose, inp, out);
        memcpy(inp, out, sizeof inp);
    } else if (chain == ORC_CHAIN_SELF) {
        if (jrow) orc_free_jac(ob + 78);
        orc_free_fun(p_pt, out);
        memcpy(inp, out, sizeof inp);
        if (jrow) orc_rigid_jac(p_pose, inp, ob + 51);
        orc_rigid_fun(p_pose, inp, out);
        memcpy(inp, out, sizeof inp);
    } else {
        if (jrow) orc_free_jac(ob + 51);
        orc_free_fun(p_pt, out);
        memcpy(inp, out, sizeof inp);
    }
    if (jrow) orc_rigid_jac(p_extr, inp, ob + 24);
    orc_rigid_fun(p_extr, inp, out);
    memcpy(inp, out, sizeof inp);
    if (jrow) {
        orc_projection_jac(p_intr, inp, ob);
        orc_matflow(chain, ob, jrow); /* afb:591-594 */
    }
    if (resid) {
        double uv[2];
        orc_projection_fun(p_intr, inp, uv);
        resid[0] = uv[0] - datum[3]; /* afb:384 */
        resid[1] = uv[1] - datum[4];
    }
}

/* Generated full_loss (afb:350-387): (N,2) = projected - measured. */
int orc_full_loss(int chain, int64_t N, const double *det /*N x 5*/, const double *param_str, int64_t n_cams,
                  int64_t n_imgs, const double *tmpl /*n_keys x 3 or NULL*/, double *losses /*N x 2*/, int threads) {
    if (orc_chain_P(chain) < 0 || (chain == ORC_CHAIN_TEMPLATE && !tmpl)) return -1;
    orc_offsets off = orc_param_offsets(chain, n_cams, n_imgs);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < N; ++i) orc_eval_one(chain, det + 5 * i, param_str, off, tmpl, losses + 2 * i, 0);
    return 0;
}

/* Generated full_jac (afb:552-599) after the [:n_elements] truncation (afb:641): dense (2N, P),
 * per detection the u row then the v row.  `losses` may be NULL. */
int orc_full_jac(int chain, int64_t N, const double *det, const double *param_str, int64_t n_cams, int64_t n_imgs,
                 const double *tmpl, double *dense /*2N x P*/, double *losses /*N x 2 or NULL*/, int threads) {
    const int P = orc_chain_P(chain);
    if (P < 0 || (chain == ORC_CHAIN_TEMPLATE && !tmpl)) return -1;
    orc_offsets off = orc_param_offsets(chain, n_cams, n_imgs);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < N; ++i)
        orc_eval_one(chain, det + 5 * i, param_str, off, tmpl, losses ? losses + 2 * i : 0, dense + 2 * P * i);
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * Legacy residual-only cost: numpy_bundle_adjustment_costfn / bundle_adjustment_costfn
 * (ch:518-549), used by the initial pose selection (template_handler.py:535-592).  Pre-multiplied
 * 3x4 projection matrices, pre-transformed points im_points[image, key], pinhole intrinsics 3x3 and
 * 5 distortion coefficients per camera.
 * ------------------------------------------------------------------------------------------- */
/* ch:443-466 nb_distort_prealloc: pts (pixels) are distorted in place; k = [k0,k1,p0,p1,k2] */
void orc_distort(double *pts, const double *intrinsics /*3x3*/, const double *k /*5*/) {
    double centre_0 = intrinsics[2], centre_1 = intrinsics[5];
    double focal_0 = intrinsics[0], focal_1 = intrinsics[4];
    double x = (pts[0] - centre_0) / focal_0, y = (pts[1] - centre_1) / focal_1; /* ch:455 */
    double r2 = ipow(x, 2) + ipow(y, 2);
    double kup = (1 + k[0] * r2 + k[1] * ipow(r2, 2) + k[4] * ipow(r2, 3));
    double xD = x * kup;
    double yD = y * kup;
    xD += 2 * k[2] * x * y + k[3] * (r2 + 2 * ipow(x, 2));
    yD += k[2] * (r2 + 2 * ipow(y, 2)) + 2 * k[3] * x * y;
    pts[0] = xD * focal_0 + centre_0; /* ch:465-466 */
    pts[1] = yD * focal_1 + centre_1;
}

/* ch:518-547: error[2i], error[2i+1] for every detection row [cam, im, key, u, v] */
int orc_legacy_cost(int64_t N, const double *dct /*N x 5*/, const double *im_points /*I x K x 3*/, int64_t n_keys,
                    const double *proj /*C x 3 x 4*/, const double *intrinsics /*C x 3 x 3*/, const double *dists /*C x 5*/,
                    double *error /*2N*/, int threads) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t idx = 0; idx < N; ++idx) {
        const double *d = dct + 5 * idx;
        const int64_t cam = (int64_t)d[0];                                        /* ch:533 */
        const double *X = im_points + 3 * ((int64_t)d[1] * n_keys + (int64_t)d[2]); /* ch:535-537 */
        const double *P = proj + 12 * cam;
        double puv[3];
        for (int r = 0; r < 3; ++r) puv[r] = P[4 * r] * X[0] + P[4 * r + 1] * X[1] + P[4 * r + 2] * X[2] + P[4 * r + 3] * 1.0; /* ch:538 */
        puv[0] = puv[0] / puv[2];                                                 /* ch:539 */
        puv[1] = puv[1] / puv[2];
        orc_distort(puv, intrinsics + 9 * cam, dists + 5 * cam);                  /* ch:540 */
        error[2 * idx] = puv[0] - d[3];                                           /* ch:541-542 */
        error[2 * idx + 1] = puv[1] - d[4];
    }
    return 0;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
