// ba_device.hpp — per-entity and per-detection device math of the bundle-adjustment hot path.
//
// gfx950 (CDNA4) only.  Everything is templated on the scalar type (double / float).
//
// What the reference evaluates per detection (pyCamSet/optimisation/, fbi =
// function_block_implementations.py, ch = compiled_helpers.py, mm = matmul_map.py):
//     pose  SE3 (fbi:150-155, :188-211)  ->  extrinsic SE3 (fbi:184-185, :157-182)  ->
//     pinhole + Brown-Conrady (fbi:27-140)  ->  chain rule `matflow` (generator mm:147-243)
// including Rodrigues (ch:197-235) and its Jacobian (ch:237-286) three resp. two times per
// detection.  Here the transcendental part is hoisted: `slab_prep` evaluates R, t, dR/dr once
// per camera and once per pose ("slabs"), and the per-detection code is FMAs plus one divide.
#pragma once
#ifndef __HIPCC_RTC__   // a chain compiled by hiprtc (pycamset_amd/chain_compiler.py): the HIP runtime declarations are pre-included, host headers do not exist
#include <hip/hip_runtime.h>

#include <cstdint>
#else
#include "ba_rtc_prelude.hpp"
#endif

namespace pcs {

constexpr int CHAIN_TEMPLATE = 0;
constexpr int CHAIN_SELF = 1;
constexpr int CHAIN_FREE = 2;

__host__ __device__ constexpr int chain_P(int chain) { return chain == CHAIN_TEMPLATE ? 21 : chain == CHAIN_SELF ? 24 : 18; }

// Slab layouts, in scalars.  Strides are multiples of 16 bytes for both dtypes.
//   camera slab : [0,9) intr fx,px,fy,py,k0,k1,p0,p1,k2 | [9,18) R_e row-major | [18,21) t_e |
//                 [21,48) dR_e[a*9 + row*3 + col] = d R_e[row][col] / d r_a
//   pose slab   : [0,9) R_p | [9,12) t_p | [12,39) dR_p | [39] pad
constexpr int CAM_STRIDE = 48;
constexpr int POSE_STRIDE = 40;
constexpr int CAM_R = 9, CAM_T = 18, CAM_DR = 21;
constexpr int POSE_R = 0, POSE_T = 9, POSE_DR = 12;

// The static detection table as every kernel sees it (td:51-55 after return_flattened_keys).
//   * indices: one packed 32-bit word per detection, cam | image | key bit fields (widths from the engine's counts),
//     or — when the three fields need more than 32 bits — three int32 arrays.  12 -> 4 bytes per detection: at 1e7
//     detections the whole table (4 + 16 B) then stays inside the 256 MiB Infinity Cache between LM steps; with
//     28 B per detection it does not, and reads that go to HBM in the middle of the write stream cost the fused
//     kernel a third of its rate (profiles/r02/sweeps.md, "input footprint").
//   * measurements: (u, v) pairs as double (PCS_F64, PCS_MIXED) or float (PCS_F32); arithmetic is always double.
// All branches on the table's form are wave-uniform.
struct DetTable {
    const uint32_t *packed;          // NULL -> use cam / img / key
    const int32_t *cam, *img, *key;
    const void *uv;
    int32_t key_bits, img_bits;      // packed = cam << (img_bits + key_bits) | img << key_bits | key
    int32_t uv_f32;
};

// raw index words of detection i (1 word when packed, 3 otherwise) and their decoding: split so that a kernel can
// request the words of its NEXT tile early and decode them when it gets there
struct DetWords { uint32_t w0, w1, w2; };
__device__ __forceinline__ DetWords load_words(const DetTable &t, const int64_t i) {
    DetWords r;
    if (t.packed) { r.w0 = t.packed[i]; r.w1 = 0; r.w2 = 0; }
    else { r.w0 = (uint32_t)t.cam[i]; r.w1 = (uint32_t)t.img[i]; r.w2 = (uint32_t)t.key[i]; }
    return r;
}
__device__ __forceinline__ void decode_words(const DetTable &t, const DetWords &r, int &c, int &im, int &k) {
    if (t.packed) {
        k = (int)(r.w0 & ((1u << t.key_bits) - 1u));
        im = (int)((r.w0 >> t.key_bits) & ((1u << t.img_bits) - 1u));
        c = (int)(r.w0 >> (t.key_bits + t.img_bits));
    } else {
        c = (int)r.w0; im = (int)r.w1; k = (int)r.w2;
    }
}
__device__ __forceinline__ void load_indices(const DetTable &t, const int64_t i, int &c, int &im, int &k) {
    decode_words(t, load_words(t, i), c, im, k);
}

using double2v = __attribute__((ext_vector_type(2))) double;
__device__ __forceinline__ double2v load_uv(const DetTable &t, const int64_t i) {
    double2v m;
    if (t.uv_f32) {
        const __attribute__((ext_vector_type(2))) float f = static_cast<const __attribute__((ext_vector_type(2))) float *>(t.uv)[i];
        m.x = (double)f.x; m.y = (double)f.y;
    } else {
        m = static_cast<const double2v *>(t.uv)[i];
    }
    return m;
}

// broadcast lane `src`'s value to the whole wave through scalar registers (v_readlane_b32)
__device__ __forceinline__ double readlane_scalar(double v, int src) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), src);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// A parameter slab held ACROSS the lanes of a wave (lane j holds element j, fetched with one coalesced load) and
// read element-wise through v_readlane into scalar registers.  Usable as the slab argument of eval_detection when
// every lane of the wave refers to the same camera (pose): 1 load instruction instead of 48 (39) per-lane loads of
// the same address, and no vector registers tied up by slab values.
struct LaneSlab {
    double v;
    __device__ __forceinline__ double operator[](const int j) const { return readlane_scalar(v, j); }
};

// A parameter slab read with SCALAR loads: the base address is wave-uniform (built from v_readfirstlane values), and a pointer
// in the constant address space makes hipcc emit s_load_dwordx{2..16} through the scalar cache instead of one vector load +
// two v_readlane per value.  The slabs are written by the previous kernel (slab_prep), so they are constant here.  Worth it
// while the values fit the ~100 SGPRs: the residual-only and legacy kernels (33 / 21 values: 9.9 -> 9.0 us at N = 1e6);
// the 87 values of a full Jacobian evaluation do not, and are kept across lanes instead (LaneSlab).
struct ScalarSlab {
    using CP = const __attribute__((address_space(4))) double *;
    CP p;
    __device__ __forceinline__ explicit ScalarSlab(const double *q) : p((CP)(uintptr_t)q) {}
    __device__ __forceinline__ double operator[](const int j) const { return p[j]; }
};

// ---------------------------------------------------------------------------------------------
// Rodrigues rotation and its derivative from a rotation vector (in double whatever the slab dtype), ONE ELEMENT PER LANE
// (round 3; rounds 1-2 computed all 36 in one thread).  q in [0, 36): q < 9 -> R[q], else dR[q - 9]
// (dR[a*9 + row*3 + col] = d R[row][col] / d r_a).  Every operation is written out with explicit roundings
// (no implicit contraction), so the element a lane computes does not depend on which kernel it was inlined
// into: the stand-alone slab_prep_kernel and the evaluation kernels that prepare their own slabs per wave
// (ba_eval_kernel<..., PREP = true>) produce the same bits.  Same formulas and the same theta < 1e-10 branches
// as ch:205-234 / ch:244-286 (pose 0 is exactly zero by default, template_handler.py:134-137, so the branch is live).
struct RotTerms {
    double r0, r1, r2;   // rotation vector
    double it, st, ct;   // 1 / theta, sin(theta), cos(theta)
    bool small;          // theta < 1e-10: R = I, dR = generators
};
__device__ __forceinline__ RotTerms rot_terms(const double r0, const double r1, const double r2) {
#pragma clang fp contract(off)
    RotTerms t;
    t.r0 = r0; t.r1 = r1; t.r2 = r2;
    const double theta = sqrt(__builtin_fma(r2, r2, __builtin_fma(r1, r1, r0 * r0)));
    t.small = theta < 1e-10;
    t.it = 1.0 / (t.small ? 1.0 : theta);
    sincos(theta, &t.st, &t.ct);
    return t;
}
__device__ __forceinline__ double sel3(const int i, const double a, const double b, const double c) { return i == 0 ? a : i == 1 ? b : c; }
// d [r]x[row][col] / d r_a: +1 / -1 / 0 (the generators; ch:246-254 and the drx table of ch:270-277)
__device__ __forceinline__ double skew_gen(const int a, const int row, const int col) {
    // [r]x = {0,-z,y, z,0,-x, -y,x,0}: entry (row, col) = -eps(row, col, m) r_m
    if (row == col || a == row || a == col) return 0.0;
    return ((col - row + 3) % 3 == 1) ? -1.0 : 1.0;   // (0,1),(1,2),(2,0) carry the minus sign
}
__device__ __forceinline__ double rot_element(const RotTerms &t, const int q) {
#pragma clang fp contract(off)
    const bool is_R = q < 9;
    const int e = is_R ? q : q - 9;
    const int a = is_R ? 0 : e / 9;          // derivative axis (dR only)
    const int k = is_R ? e : e - 9 * a;
    const int row = k / 3, col = k - 3 * row;
    const int m = 3 - row - col;             // the axis of the skew entry (row != col)
    const double sgn = (row == col) ? 0.0 : (((col - row + 3) % 3 == 1) ? -1.0 : 1.0);
    if (t.small) {
        if (is_R) return row == col ? 1.0 : 0.0;
        return skew_gen(a, row, col);
    }
    if (is_R) {
        // R = ct I + (1 - ct) / theta^2 r r^T + st / theta [r]x       (un-normalised r, ch:213-234)
        const double f = (1.0 - t.ct) * (t.it * t.it);
        const double s = t.st * t.it;
        const double v = (sel3(row, t.r0, t.r1, t.r2) * sel3(col, t.r0, t.r1, t.r2)) * f;
        if (row == col) return v + t.ct;
        return __builtin_fma(sgn * sel3(m, t.r0, t.r1, t.r2), s, v);
    }
    // dR/dr_a = a0 I + a1 rr^T + a2 d(rr^T)/dr_a + a3 [r]x + a4 d[r]x/dr_a on the unit axis (ch:256-286)
    const double x = t.r0 * t.it, y = t.r1 * t.it, z = t.r2 * t.it;
    const double ri = sel3(a, x, y, z), ar = sel3(row, x, y, z), ac = sel3(col, x, y, z);
    const double ct_1 = 1.0 - t.ct;
    const double a0 = -t.st * ri;
    const double a1 = (t.st - (2.0 * ct_1) * t.it) * ri;
    const double a2 = ct_1 * t.it;
    const double a3 = (t.ct - t.st * t.it) * ri;
    const double a4 = t.st * t.it;
    const double eye = row == col ? 1.0 : 0.0;
    const double rrt = ar * ac;
    const double drrt = (a == row ? ac : 0.0) + (a == col ? ar : 0.0);
    const double rx = row == col ? 0.0 : sgn * sel3(m, x, y, z);
    double v = a0 * eye;
    v = __builtin_fma(a1, rrt, v);
    v = __builtin_fma(a2, drrt, v);
    v = __builtin_fma(a3, rx, v);
    v = __builtin_fma(a4, skew_gen(a, row, col), v);
    return v;
}
// where element q of a rotation goes inside a camera / pose slab
__device__ __forceinline__ int cam_slot_of(const int q) { return q < 9 ? CAM_R + q : CAM_DR + (q - 9); }
__device__ __forceinline__ int pose_slot_of(const int q) { return q < 9 ? POSE_R + q : POSE_DR + (q - 9); }

// The reference's projection multiplies by the focal length and divides by it again before distorting (fbi:32-35:
// u = (fx x + px z) / z, x_n = (u - px) / fx), so a focal length of 0, NaN or +-inf makes BOTH residual rows of that
// camera's detections non-finite, while its Jacobian formulas (fbi:62-134) never form that quotient.  The kernels divide
// x / z directly; to keep the reference's behaviour the slab copy of the principal point (used by the residual only:
// u = xD fx + px, d u / d px = 1 is a constant) is NaN for such a camera.  Pinned by tests/golden/block_*_focal_nonfinite.npz.
__device__ __forceinline__ double principal_or_nan(const double v, const int slot, const double fx, const double fy) {
    const bool ok = fx != 0.0 && fy != 0.0 && (fx - fx) == 0.0 && (fy - fy) == 0.0;   // finite and non-zero
    return ((slot == 1 || slot == 3) && !ok) ? __builtin_nan("") : v;
}

// One detection through the chain.
//   cs : camera slab (CAM_STRIDE scalars), ps : pose slab (POSE_STRIDE scalars, unused for CHAIN_FREE)
//   X  : template point (CHAIN_TEMPLATE) or free 3-D point (CHAIN_SELF / CHAIN_FREE)
// Outputs: projected pixel (u,v) and, when JAC, the dense 2 x P block `J` row-major with the
// reference's column order:
//   T: [A_p(9) | A_x.E_r(3) | A_x(3) | S.Q_r(3) | S(3)]                S = A_x.R_e
//   S: [A_p | A_x.E_r | A_x | S.G_r | S | S.R_p]
//   F: [A_p | A_x.E_r | A_x | S]
// The projection Jacobian is evaluated in normalised coordinates a = x/z, b = y/z; it is the
// same rational function as fbi:62-134 (which carries z**7, z**8 denominators).
template <int CHAIN, typename T, bool JAC, typename SlabPtrC, typename SlabPtrP>
__device__ __forceinline__ void eval_detection(SlabPtrC cs, SlabPtrP ps, const T X0, const T X1, const T X2, T &u, T &v,
                                               T (&J)[2 * chain_P(CHAIN)]) {
    constexpr int P = chain_P(CHAIN);
    T Xw0, Xw1, Xw2;
    T Qr[9];  // Qr[c*3 + a] = (dR_p[a] X)[c]
    if constexpr (CHAIN != CHAIN_FREE) {
        const T r0 = ps[POSE_R + 0], r1 = ps[POSE_R + 1], r2 = ps[POSE_R + 2];
        const T r3 = ps[POSE_R + 3], r4 = ps[POSE_R + 4], r5 = ps[POSE_R + 5];
        const T r6 = ps[POSE_R + 6], r7 = ps[POSE_R + 7], r8 = ps[POSE_R + 8];
        Xw0 = r0 * X0 + r1 * X1 + r2 * X2 + ps[POSE_T + 0];
        Xw1 = r3 * X0 + r4 * X1 + r5 * X2 + ps[POSE_T + 1];
        Xw2 = r6 * X0 + r7 * X1 + r8 * X2 + ps[POSE_T + 2];
        if constexpr (JAC) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    Qr[c * 3 + a] = ps[POSE_DR + a * 9 + c * 3 + 0] * X0 + ps[POSE_DR + a * 9 + c * 3 + 1] * X1 +
                                    ps[POSE_DR + a * 9 + c * 3 + 2] * X2;
        }
    } else {
        Xw0 = X0; Xw1 = X1; Xw2 = X2;
    }
    const T e0 = cs[CAM_R + 0], e1 = cs[CAM_R + 1], e2 = cs[CAM_R + 2];
    const T e3 = cs[CAM_R + 3], e4 = cs[CAM_R + 4], e5 = cs[CAM_R + 5];
    const T e6 = cs[CAM_R + 6], e7 = cs[CAM_R + 7], e8 = cs[CAM_R + 8];
    const T x = e0 * Xw0 + e1 * Xw1 + e2 * Xw2 + cs[CAM_T + 0];
    const T y = e3 * Xw0 + e4 * Xw1 + e5 * Xw2 + cs[CAM_T + 1];
    const T z = e6 * Xw0 + e7 * Xw1 + e8 * Xw2 + cs[CAM_T + 2];

    const T fx = cs[0], px = cs[1], fy = cs[2], py = cs[3];
    const T k0 = cs[4], k1 = cs[5], p0 = cs[6], p1 = cs[7], k2 = cs[8];
    const T iz = T(1) / z;
    const T a = x * iz, b = y * iz;
    const T a2 = a * a, b2 = b * b, ab = a * b;
    const T r2 = a2 + b2;
    const T r4 = r2 * r2;
    const T r6 = r4 * r2;
    const T kup = T(1) + k0 * r2 + k1 * r4 + k2 * r6;                  // fbi:37
    const T xD = a * kup + T(2) * p0 * ab + p1 * (r2 + T(2) * a2);     // fbi:39-42
    const T yD = b * kup + p0 * (r2 + T(2) * b2) + T(2) * p1 * ab;     // fbi:40-43
    u = xD * fx + px;                                                  // fbi:45
    v = yD * fy + py;                                                  // fbi:46
    if constexpr (JAC) {
        const T dk = k0 + T(2) * k1 * r2 + T(3) * k2 * r4;             // d kup / d r2
        // d(u,v)/d(a,b)
        const T ua = fx * (kup + T(2) * a2 * dk + T(2) * p0 * b + T(6) * p1 * a);
        const T cross = T(2) * (ab * dk + p0 * a + p1 * b);
        const T ub = fx * cross;
        const T va = fy * cross;
        const T vb = fy * (kup + T(2) * b2 * dk + T(6) * p0 * b + T(2) * p1 * a);
        // A_x = d(u,v)/d(x,y,z) (fbi:79-97, fbi:120-134)
        T Ax[2][3];
        Ax[0][0] = ua * iz; Ax[0][1] = ub * iz; Ax[0][2] = -(a * ua + b * ub) * iz;
        Ax[1][0] = va * iz; Ax[1][1] = vb * iz; Ax[1][2] = -(a * va + b * vb) * iz;
        // A_p: intrinsics + distortion (fbi:58-77, fbi:99-118); the explicit 0 / 1 entries are stored
        // like the reference's CSR does (mm:231, mm:237-242).
        J[0] = xD;            J[1] = T(1); J[2] = T(0);          J[3] = T(0);
        J[4] = fx * a * r2;   J[5] = fx * a * r4;  J[6] = T(2) * fx * ab;  J[7] = fx * (r2 + T(2) * a2);  J[8] = fx * a * r6;
        J[P + 0] = T(0);      J[P + 1] = T(0);     J[P + 2] = yD;          J[P + 3] = T(1);
        J[P + 4] = fy * b * r2; J[P + 5] = fy * b * r4; J[P + 6] = fy * (r2 + T(2) * b2); J[P + 7] = T(2) * fy * ab; J[P + 8] = fy * b * r6;
        // extrinsic rotation columns: A_x . E_r, E_r[:,a] = dR_e[a] X_w     (fbi:163-170)
        T Er[9];
#pragma unroll
        for (int aa = 0; aa < 3; ++aa)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                Er[c * 3 + aa] = cs[CAM_DR + aa * 9 + c * 3 + 0] * Xw0 + cs[CAM_DR + aa * 9 + c * 3 + 1] * Xw1 +
                                 cs[CAM_DR + aa * 9 + c * 3 + 2] * Xw2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int aa = 0; aa < 3; ++aa) J[i * P + 9 + aa] = Ax[i][0] * Er[0 * 3 + aa] + Ax[i][1] * Er[1 * 3 + aa] + Ax[i][2] * Er[2 * 3 + aa];
#pragma unroll
            for (int aa = 0; aa < 3; ++aa) J[i * P + 12 + aa] = Ax[i][aa];     // E_t = I (fbi:172-174)
        }
        // S = A_x . R_e  (fbi:177-181)
        T S[2][3];
        const T Re[9] = {e0, e1, e2, e3, e4, e5, e6, e7, e8};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) S[i][c] = Ax[i][0] * Re[0 * 3 + c] + Ax[i][1] * Re[1 * 3 + c] + Ax[i][2] * Re[2 * 3 + c];
        if constexpr (CHAIN == CHAIN_FREE) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) J[i * P + 15 + c] = S[i][c];
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int aa = 0; aa < 3; ++aa) J[i * P + 15 + aa] = S[i][0] * Qr[0 * 3 + aa] + S[i][1] * Qr[1 * 3 + aa] + S[i][2] * Qr[2 * 3 + aa];
#pragma unroll
                for (int c = 0; c < 3; ++c) J[i * P + 18 + c] = S[i][c];
            }
            if constexpr (CHAIN == CHAIN_SELF) {  // point columns: S . R_p   (free_point Jacobian = I, fbi:234-240)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        J[i * P + 21 + c] = S[i][0] * ps[POSE_R + 0 * 3 + c] + S[i][1] * ps[POSE_R + 1 * 3 + c] + S[i][2] * ps[POSE_R + 2 * 3 + c];
            }
        }
    }
}

// J_i dv WITHOUT forming J_i: the directional derivative of one detection's projection along dv (forward mode), for a tile whose
// detections share camera and image.  Same chain rule as eval_detection, applied to a vector instead of to the identity:
//   d X_w = M_p X + dv_pt (+ R_p dv_X)         M_p = sum_a dv_pr[a] dR_p[a]   (the caller forms M_e and M_p once per tile)
//   d X_c = M_e X_w + dv_et + R_e d X_w        M_e = sum_a dv_er[a] dR_e[a]
//   d(u, v) = A_x d X_c + A_p dv_intr
// ~110 FP64 operations per detection where the 2 x P block and its product with dv take ~340.  (Round 5; the matrix-free J v is
// issue-bound in the evaluation: DESIGN section 4.)
//   cs / ps : camera / pose slab (R, t and the nine intrinsics are read; the dR blocks are not)
//   M       : M_e (entries 0-8, row-major) and M_p (9-17)
//   vi      : dv of the nine intrinsics; vet / vpt: dv of the extrinsic / pose translation; vX: dv of the point (chains S and F)
template <int CHAIN, typename SlabPtrC, typename SlabPtrP, typename MatPtr, typename VecPtr>
__device__ __forceinline__ void eval_detection_jvp(SlabPtrC cs, SlabPtrP ps, const double X0, const double X1, const double X2, MatPtr M, VecPtr vi,
                                                   VecPtr vet, VecPtr vpt, const double vX0, const double vX1, const double vX2, double &du, double &dv) {
    double Xw0, Xw1, Xw2, dW0, dW1, dW2;
    if constexpr (CHAIN != CHAIN_FREE) {
        const double r0 = ps[POSE_R + 0], r1 = ps[POSE_R + 1], r2 = ps[POSE_R + 2];
        const double r3 = ps[POSE_R + 3], r4 = ps[POSE_R + 4], r5 = ps[POSE_R + 5];
        const double r6 = ps[POSE_R + 6], r7 = ps[POSE_R + 7], r8 = ps[POSE_R + 8];
        Xw0 = r0 * X0 + r1 * X1 + r2 * X2 + ps[POSE_T + 0];
        Xw1 = r3 * X0 + r4 * X1 + r5 * X2 + ps[POSE_T + 1];
        Xw2 = r6 * X0 + r7 * X1 + r8 * X2 + ps[POSE_T + 2];
        dW0 = M[9] * X0 + M[10] * X1 + M[11] * X2 + vpt[0];
        dW1 = M[12] * X0 + M[13] * X1 + M[14] * X2 + vpt[1];
        dW2 = M[15] * X0 + M[16] * X1 + M[17] * X2 + vpt[2];
        if constexpr (CHAIN == CHAIN_SELF) {
            dW0 += r0 * vX0 + r1 * vX1 + r2 * vX2;
            dW1 += r3 * vX0 + r4 * vX1 + r5 * vX2;
            dW2 += r6 * vX0 + r7 * vX1 + r8 * vX2;
        }
    } else {
        Xw0 = X0; Xw1 = X1; Xw2 = X2;
        dW0 = vX0; dW1 = vX1; dW2 = vX2;
    }
    const double e0 = cs[CAM_R + 0], e1 = cs[CAM_R + 1], e2 = cs[CAM_R + 2];
    const double e3 = cs[CAM_R + 3], e4 = cs[CAM_R + 4], e5 = cs[CAM_R + 5];
    const double e6 = cs[CAM_R + 6], e7 = cs[CAM_R + 7], e8 = cs[CAM_R + 8];
    const double x = e0 * Xw0 + e1 * Xw1 + e2 * Xw2 + cs[CAM_T + 0];
    const double y = e3 * Xw0 + e4 * Xw1 + e5 * Xw2 + cs[CAM_T + 1];
    const double z = e6 * Xw0 + e7 * Xw1 + e8 * Xw2 + cs[CAM_T + 2];
    const double dx = M[0] * Xw0 + M[1] * Xw1 + M[2] * Xw2 + vet[0] + e0 * dW0 + e1 * dW1 + e2 * dW2;
    const double dy = M[3] * Xw0 + M[4] * Xw1 + M[5] * Xw2 + vet[1] + e3 * dW0 + e4 * dW1 + e5 * dW2;
    const double dz = M[6] * Xw0 + M[7] * Xw1 + M[8] * Xw2 + vet[2] + e6 * dW0 + e7 * dW1 + e8 * dW2;
    const double fx = cs[0], fy = cs[2];
    const double k0 = cs[4], k1 = cs[5], p0 = cs[6], p1 = cs[7], k2 = cs[8];
    const double iz = 1.0 / z;
    const double a = x * iz, b = y * iz;
    const double a2 = a * a, b2 = b * b, ab = a * b;
    const double r2 = a2 + b2, r4 = r2 * r2, r6 = r4 * r2;
    const double kup = 1.0 + k0 * r2 + k1 * r4 + k2 * r6;
    const double dk = k0 + 2.0 * k1 * r2 + 3.0 * k2 * r4;
    const double ua = fx * (kup + 2.0 * a2 * dk + 2.0 * p0 * b + 6.0 * p1 * a);
    const double cross = 2.0 * (ab * dk + p0 * a + p1 * b);
    const double ub = fx * cross, va = fy * cross;
    const double vb = fy * (kup + 2.0 * b2 * dk + 6.0 * p0 * b + 2.0 * p1 * a);
    const double da = (dx - a * dz) * iz, db = (dy - b * dz) * iz;
    const double xD = a * kup + 2.0 * p0 * ab + p1 * (r2 + 2.0 * a2);
    const double yD = b * kup + p0 * (r2 + 2.0 * b2) + 2.0 * p1 * ab;
    // A_p dv_intr: [xD, 1, 0, 0, fx a r2, fx a r4, 2 fx ab, fx (r2 + 2 a2), fx a r6] and the v row likewise (eval_detection's J[0..8], J[P..P+8])
    du = ua * da + ub * db + xD * vi[0] + vi[1] + fx * (a * (r2 * vi[4] + r4 * vi[5] + r6 * vi[8]) + 2.0 * ab * vi[6] + (r2 + 2.0 * a2) * vi[7]);
    dv = va * da + vb * db + yD * vi[2] + vi[3] + fy * (b * (r2 * vi[4] + r4 * vi[5] + r6 * vi[8]) + (r2 + 2.0 * b2) * vi[6] + 2.0 * ab * vi[7]);
}

}  // namespace pcs
