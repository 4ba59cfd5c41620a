// ba_generic.hpp — the fused residual / Jacobian kernel for ANY chain of the reference's function blocks
// (SURVEY 8a rows a8-a10 as a generator, not as three hand-written outputs of one).
//
// The reference composes blocks with `+` (abstract_function_blocks.py:735-748) and code-generates the loss, the Jacobian
// driver and the chain-rule product `matflow` for the composition (afb:290-419, afb:492-652, matmul_map.py:147-263).
// pycamset_amd/chain_compiler.py does the same for the GPU: it turns a block list
//     projection + T_1 + ... + T_M + source        T_i in {rigidTform3d (per image), extrinsic3D (per camera)},
//                                                  source in {template_points (per image), free_point (per key)}
// into a `ChainSpec` — a struct of constexpr tables, emitted as a ~20-line .hip file that includes this header — and has
// hipcc compile it for gfx950 (--genco).  Everything below is generic over that struct: block maths (the same Rodrigues
// slabs, prepared per parameter group by generic_slab_prep_kernel), the chain rule the reference builds symbolically
//     S_0 = A_x;  columns of T_i = [S_{i-1} E_i | S_{i-1}],  E_i[:, a] = dR_i/dr_a X_in,i;  S_i = S_{i-1} R_i
// (mm:181-243: the product of identity-embedded block Jacobians), and the coalesced store phase of the hand-fused kernels
// (store_jac_tile).  The three chains the reference's handlers build keep their hand-fused kernels (ba_kernels.hpp); a
// generated kernel for one of them computes the same function (tests compare them).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ba_device.hpp"
#include "ba_kernels.hpp"

namespace pcs {

constexpr int LINK_CAM = 0, LINK_IMG = 1, LINK_KEY = 2;      // afb:42-46 key_type
constexpr int SRC_TEMPLATE = 0, SRC_FREE = 1;
constexpr int GENERIC_MAX_GROUPS = 8;

struct GenericArgs {
    DetTable tab;
    const double *prm;                       // parameter string (device)
    const double *tmpl;                      // template points (SRC_TEMPLATE)
    double *slab[GENERIC_MAX_GROUPS];        // per rigid parameter group: count x POSE_STRIDE (R | t | dR/dr | pad)
    int64_t group_off[GENERIC_MAX_GROUPS];   // first parameter-string column of the group
    int32_t group_count[GENERIC_MAX_GROUPS];
    int32_t n_groups;
    int64_t intr_off, point_off;             // projection / free_point groups
    void *resid, *jac, *sink;
    int64_t n, n_tiles;
    int32_t tiles_per_wg;
};

// one thread per slab element over all rigid groups (same element functions as slab_prep_kernel)
__device__ __forceinline__ void generic_slab_prep_body(const GenericArgs &a) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int g = 0; g < a.n_groups; ++g) {
        const int64_t n_el = (int64_t)a.group_count[g] * POSE_STRIDE;
        if (t < n_el) {
            const int64_t e = t / POSE_STRIDE;
            const int slot = (int)(t - e * POSE_STRIDE);
            const double *p6 = a.prm + a.group_off[g] + 6 * e;
            double v;
            if (slot >= POSE_T && slot < POSE_DR) v = p6[3 + slot - POSE_T];
            else if (slot == POSE_STRIDE - 1) v = 0.0;
            else v = rot_element(rot_terms(p6[0], p6[1], p6[2]), slot < POSE_T ? slot - POSE_R : 9 + slot - POSE_DR);
            a.slab[g][t] = v;
            return;
        }
        t -= n_el;
    }
}

// pinhole + Brown-Conrady and its Jacobian in normalised coordinates (same rational function as fbi:27-140; see eval_detection)
template <bool JAC, typename IntrPtr>
__device__ __forceinline__ void project_generic(IntrPtr cs, const double x, const double y, const double z, double &u, double &v, double (&Ap)[18],
                                                double (&Ax)[2][3]) {
    using T = double;
    const T fx = cs[0], px = cs[1], fy = cs[2], py = cs[3];
    const T k0 = cs[4], k1 = cs[5], p0 = cs[6], p1 = cs[7], k2 = cs[8];
    const T iz = T(1) / z;
    const T a = x * iz, b = y * iz;
    const T a2 = a * a, b2 = b * b, ab = a * b;
    const T r2 = a2 + b2;
    const T r4 = r2 * r2;
    const T r6 = r4 * r2;
    const T kup = T(1) + k0 * r2 + k1 * r4 + k2 * r6;
    const T xD = a * kup + T(2) * p0 * ab + p1 * (r2 + T(2) * a2);
    const T yD = b * kup + p0 * (r2 + T(2) * b2) + T(2) * p1 * ab;
    u = xD * fx + px;
    v = yD * fy + py;
    if constexpr (JAC) {
        const T dk = k0 + T(2) * k1 * r2 + T(3) * k2 * r4;
        const T ua = fx * (kup + T(2) * a2 * dk + T(2) * p0 * b + T(6) * p1 * a);
        const T cross = T(2) * (ab * dk + p0 * a + p1 * b);
        const T ub = fx * cross;
        const T va = fy * cross;
        const T vb = fy * (kup + T(2) * b2 * dk + T(6) * p0 * b + T(2) * p1 * a);
        Ax[0][0] = ua * iz; Ax[0][1] = ub * iz; Ax[0][2] = -(a * ua + b * ub) * iz;
        Ax[1][0] = va * iz; Ax[1][1] = vb * iz; Ax[1][2] = -(a * va + b * vb) * iz;
        Ap[0] = xD;           Ap[1] = T(1); Ap[2] = T(0);          Ap[3] = T(0);
        Ap[4] = fx * a * r2;  Ap[5] = fx * a * r4;  Ap[6] = T(2) * fx * ab;  Ap[7] = fx * (r2 + T(2) * a2);  Ap[8] = fx * a * r6;
        Ap[9] = T(0);         Ap[10] = T(0);        Ap[11] = yD;             Ap[12] = T(1);
        Ap[13] = fy * b * r2; Ap[14] = fy * b * r4; Ap[15] = fy * (r2 + T(2) * b2); Ap[16] = T(2) * fy * ab; Ap[17] = fy * b * r6;
    }
}

// intrinsics with the reference's non-finite-focal behaviour (principal_or_nan, ba_device.hpp)
struct IntrRow {
    const double *p;
    __device__ __forceinline__ double operator[](const int j) const { return principal_or_nan(p[j], j, p[0], p[2]); }
};

// One detection through a generated chain.  `slab(i)` returns the slab accessor of transform block i (0 .. M-1, in block
// order) and of the template source (i = M); X = template point or free point.
template <typename Spec, bool JAC, typename SlabOf, typename IntrPtr>
__device__ __forceinline__ void eval_generic(SlabOf slab, IntrPtr intr, const double X0, const double X1, const double X2, double &u, double &v,
                                             double (&J)[2 * Spec::P]) {
    constexpr int M = Spec::M;
    constexpr int P = Spec::P;
    double Xc[3] = {X0, X1, X2};
    double Qs[9];
    if constexpr (Spec::SRC == SRC_TEMPLATE) {   // template_points: the pose of the target (fbi:188-211)
        const auto ps = slab(M);
        const double a0 = Xc[0], a1 = Xc[1], a2 = Xc[2];
        if constexpr (JAC) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) Qs[c * 3 + a] = ps[POSE_DR + a * 9 + c * 3 + 0] * a0 + ps[POSE_DR + a * 9 + c * 3 + 1] * a1 + ps[POSE_DR + a * 9 + c * 3 + 2] * a2;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) Xc[c] = ps[POSE_R + 3 * c + 0] * a0 + ps[POSE_R + 3 * c + 1] * a1 + ps[POSE_R + 3 * c + 2] * a2 + ps[POSE_T + c];
    }
    // transforms, rightmost first (the chain applies its blocks right to left)
    double E[M > 0 ? M : 1][9];
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
        const auto ps = slab(i);
        const double a0 = Xc[0], a1 = Xc[1], a2 = Xc[2];
        if constexpr (JAC) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) E[i][c * 3 + a] = ps[POSE_DR + a * 9 + c * 3 + 0] * a0 + ps[POSE_DR + a * 9 + c * 3 + 1] * a1 + ps[POSE_DR + a * 9 + c * 3 + 2] * a2;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) Xc[c] = ps[POSE_R + 3 * c + 0] * a0 + ps[POSE_R + 3 * c + 1] * a1 + ps[POSE_R + 3 * c + 2] * a2 + ps[POSE_T + c];
    }
    double Ap[18], Ax[2][3];
    project_generic<JAC>(intr, Xc[0], Xc[1], Xc[2], u, v, Ap, Ax);
    if constexpr (JAC) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 9; ++j) J[r * P + j] = Ap[r * 9 + j];
        double S[2][3] = {{Ax[0][0], Ax[0][1], Ax[0][2]}, {Ax[1][0], Ax[1][1], Ax[1][2]}};
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const auto ps = slab(i);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int a = 0; a < 3; ++a) J[r * P + 9 + 6 * i + a] = S[r][0] * E[i][0 * 3 + a] + S[r][1] * E[i][1 * 3 + a] + S[r][2] * E[i][2 * 3 + a];
#pragma unroll
                for (int c = 0; c < 3; ++c) J[r * P + 9 + 6 * i + 3 + c] = S[r][c];
            }
            double Sn[2][3];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Sn[r][c] = S[r][0] * ps[POSE_R + 0 * 3 + c] + S[r][1] * ps[POSE_R + 1 * 3 + c] + S[r][2] * ps[POSE_R + 2 * 3 + c];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) S[r][c] = Sn[r][c];
        }
        constexpr int C0 = 9 + 6 * M;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if constexpr (Spec::SRC == SRC_TEMPLATE) {
#pragma unroll
                for (int a = 0; a < 3; ++a) J[r * P + C0 + a] = S[r][0] * Qs[0 * 3 + a] + S[r][1] * Qs[1 * 3 + a] + S[r][2] * Qs[2 * 3 + a];
#pragma unroll
                for (int c = 0; c < 3; ++c) J[r * P + C0 + 3 + c] = S[r][c];
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) J[r * P + C0 + c] = S[r][c];   // free_point: d X / d point = I (fbi:234-240)
            }
        }
    }
}

// Fused residual + Jacobian for a generated chain: the hand-fused kernel's tile-per-wave structure, scalar-load slabs when the
// tile shares camera and image, transposed non-temporal stores.  MODE as in ba_eval_kernel (1 residual, 2 Jacobian, 3 both).
template <typename Spec, int MODE, typename TO>
__device__ __forceinline__ void generic_eval_body(const GenericArgs &a) {
    constexpr int P = Spec::P;
    constexpr int P2 = 2 * P;
    constexpr bool JAC = (MODE & MODE_JAC) != 0;
    constexpr bool RES = (MODE & MODE_RESID) != 0;
    using O2 = typename Vec2<TO>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6, lane = threadIdx.x & 63;
    constexpr int LROW = lds_row_stride(P2, (int)sizeof(TO));
    TO *tr = reinterpret_cast<TO *>(smem_raw) + wave * (HALF * LROW);
    const int64_t tile0 = (int64_t)blockIdx.x * a.tiles_per_wg;
    const int64_t tile1 = min(tile0 + (int64_t)a.tiles_per_wg, a.n_tiles);
    TO *resid = static_cast<TO *>(a.resid);
    TO *jac = static_cast<TO *>(a.jac);
    const int64_t total_jac = a.n * (int64_t)P2;
    for (int64_t tile = tile0 + wave; tile < tile1; tile += n_waves) {
        const int64_t i = tile * TILE + lane;
        const bool valid = i < a.n;
        const int64_t ic = valid ? i : a.n - 1;
        int c, im, k;
        load_indices(a.tab, ic, c, im, k);
        const double2v m = load_uv(a.tab, ic);
        const double *pt = Spec::SRC == SRC_TEMPLATE ? a.tmpl + 3 * (int64_t)k : a.prm + a.point_off + 3 * (int64_t)k;
        const double X0 = pt[0], X1 = pt[1], X2 = pt[2];
        auto link_index = [&](const int link, const int cc, const int ii) { return link == LINK_CAM ? cc : ii; };
        double u, v;
        double J[P2];
        const int c0 = __builtin_amdgcn_readfirstlane(c), im0 = __builtin_amdgcn_readfirstlane(im);
        if (__all(c == c0 && im == im0)) {   // one camera and one image in the tile: every slab through scalar loads
            auto slab = [&](const int blk) { return ScalarSlab(a.slab[Spec::group(blk)] + (int64_t)link_index(Spec::link(blk), c0, im0) * POSE_STRIDE); };
            eval_generic<Spec, JAC>(slab, IntrRow{a.prm + a.intr_off + 9 * (int64_t)c0}, X0, X1, X2, u, v, J);
        } else {
            auto slab = [&](const int blk) { return static_cast<const double *>(a.slab[Spec::group(blk)]) + (int64_t)link_index(Spec::link(blk), c, im) * POSE_STRIDE; };
            eval_generic<Spec, JAC>(slab, IntrRow{a.prm + a.intr_off + 9 * (int64_t)c}, X0, X1, X2, u, v, J);
        }
        if constexpr (RES) {
            O2 r;
            r.x = (TO)(u - m.x);
            r.y = (TO)(v - m.y);
            __builtin_nontemporal_store(r, valid ? reinterpret_cast<O2 *>(resid) + i : static_cast<O2 *>(a.sink));
        }
        if constexpr (JAC) store_jac_tile<P, TO, true>(J, tr, jac, tile, lane, total_jac);
    }
}

// data[i] = dense[src[i]]: the fixed-parameter mask of a generated chain as a static gather (afb:644-651 on the device)
struct GatherArgs {
    const double *dense;
    const int64_t *src;
    double *data;
    int64_t nnz;
};
__device__ __forceinline__ void generic_gather_body(const GatherArgs &g) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < g.nnz) g.data[i] = g.dense[g.src[i]];
}

// What chain_compiler.py appends after the ChainSpec struct it emits: the entry points of the code object.
#define PCS_GENCHAIN_ENTRY_POINTS(Spec)                                                                                              \
    extern "C" __global__ void pcs_genchain_prep(const pcs::GenericArgs a) { pcs::generic_slab_prep_body(a); }                        \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_1(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 1, double>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_2(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 2, double>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_3(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 3, double>(a); } \
    extern "C" __global__ void pcs_genchain_gather(const pcs::GatherArgs g) { pcs::generic_gather_body(g); }

}  // namespace pcs
