// ba_generic.hpp — the fused residual / Jacobian kernel for ANY chain of the reference's function blocks
// (SURVEY 8a rows a8-a10 as a generator, not as three hand-written outputs of one).
//
// The reference composes blocks with `+` (abstract_function_blocks.py:735-748) and code-generates the loss, the Jacobian
// driver and the chain-rule product `matflow` for the composition (afb:290-419, afb:492-652, matmul_map.py:147-263).
// pycamset_amd/chain_compiler.py does the same for the GPU: it turns a block list
//     projection + T_1 + ... + T_M + source        T_i in {rigidTform3d (per image), extrinsic3D (per camera)},
//                                                  source in {template_points (per image), free_point (per key)}
// into straight-line device code — `struct Chain { P; eval<JAC>(ctx, u, v, J) }`, a .hip file of a few dozen lines that includes
// this header — and has hipcc compile it for gfx950 (--genco).  Round 4: the composition may also contain USER blocks
// (pycamset_amd.function_blocks.device_function_block — the counterpart of the reference's extension point, the
// abstract_function_block ABC, afb:689-775): a block declares its parameter group, num_inp / num_out and two device bodies
// (forward, Jacobian in the reference's `compute_jac` layout num_out x [params | inp]); the bodies are pasted into the
// generated translation unit and chained by the same rule.  This header holds what the generated code calls: the built-in
// blocks' maths (the same Rodrigues slabs, prepared per parameter group by generic_slab_prep_body), the chain rule the
// reference builds symbolically
//     S = d(u, v) / d(output of the block);  columns of block b = S . d out_b / d params_b;  S <- S . d out_b / d inp_b
// (mm:181-243: the product of identity-embedded block Jacobians; for a rigid block [S E | S] and S R), the kernel bodies
// (tile per wave, scalar-load slabs when a tile shares camera and image) and the coalesced store phases of the hand-fused
// kernels (store_jac_tile; compaction at the store like ba_compact_tile_kernel).  The three chains the reference's handlers
// build keep their hand-fused kernels (ba_kernels.hpp); a generated kernel for one of them computes the same function (tests
// compare them).
#pragma once
#ifndef __HIPCC_RTC__   // a chain compiled by hiprtc (pycamset_amd/chain_compiler.py): the HIP runtime declarations are pre-included, host headers do not exist
#include <hip/hip_runtime.h>

#include <cstdint>
#else
#include "ba_rtc_prelude.hpp"
#endif

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
    int64_t user_off[GENERIC_MAX_GROUPS];    // first column of the parameter group of user block u (in block order)
    void *resid, *jac, *sink;
    int64_t n, n_tiles;
    int32_t tiles_per_wg;
    // compaction at the store (pcs_genchain_eval_compact): per detection the kept local columns and the offset of its u row in `jac` (= data)
    const uint64_t *keep;
    const int64_t *row_off;
};

// element `slot` of the slab (R | t | dR/dr | pad) of one 6-parameter transform (same element functions as slab_prep_kernel)
__device__ __forceinline__ double generic_slab_element(const double *p6, const int slot) {
    if (slot >= POSE_T && slot < POSE_DR) return p6[3 + slot - POSE_T];
    if (slot == POSE_STRIDE - 1) return 0.0;
    return rot_element(rot_terms(p6[0], p6[1], p6[2]), slot < POSE_T ? slot - POSE_R : 9 + slot - POSE_DR);
}

// two-launch form: one thread per slab element over all rigid groups
__device__ __forceinline__ void generic_slab_prep_body(const GenericArgs &a) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int g = 0; g < a.n_groups; ++g) {
        const int64_t n_el = (int64_t)a.group_count[g] * POSE_STRIDE;
        if (t < n_el) {
            const int64_t e = t / POSE_STRIDE;
            a.slab[g][t] = generic_slab_element(a.prm + a.group_off[g] + 6 * e, (int)(t - e * POSE_STRIDE));
            return;
        }
        t -= n_el;
    }
}

// One-launch form: the slabs of ONE (camera, image) pair — group g's transform of that camera or image, by the group's link — held
// ACROSS THE LANES of the calling wave: lane l computes elements l, 64 + l, ... of the concatenated slabs (ba_kernels.hpp's
// prep_pair_slab for any set of groups), and a slab entry is a v_readlane broadcast of a compile-time lane (WaveSlab): like the
// scalar loads of the two-launch form the values arrive in scalar registers, and no memory is involved at all.
__device__ __forceinline__ double lane_bcast_f64(const double v, const int src) {   // v_readlane_b32 x 2 -> a scalar register pair
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), src);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
template <typename Spec>
struct LaneSlabs {
    static constexpr int NEL = Spec::N_SLABS * POSE_STRIDE, ROUNDS = (NEL + 63) / 64;
    double el[ROUNDS > 0 ? ROUNDS : 1];
    __device__ __forceinline__ void prepare(const GenericArgs &a, const int c0, const int im0, const int lane) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int e = 64 * r + lane;
            const int g = e < NEL ? e / POSE_STRIDE : 0;   // lanes past the end recompute an element of group 0 (never read)
            const int idx = Spec::slab_link(g) == LINK_CAM ? c0 : im0;
            el[r] = generic_slab_element(a.prm + a.group_off[g] + 6 * (int64_t)idx, e < NEL ? e - g * POSE_STRIDE : 0);
        }
    }
};
template <typename Spec>
struct WaveSlab {
    const LaneSlabs<Spec> &s;
    int base;
    __device__ __forceinline__ double operator[](const int j) const { return lane_bcast_f64(s.el[(base + j) >> 6], (base + j) & 63); }   // compile-time j
};
// a tile across a run boundary: the lanes of its first part take pair A's entry, the others pair B's — both broadcasts with every
// lane active (a cross-lane read from inside a divergent branch may find the source lane's register not kept up: the compiler
// maintains a value only in the lanes that are active where it is used)
template <typename Spec>
struct WaveSlab2 {
    const LaneSlabs<Spec> &sa, &sb;
    bool first;
    int base;
    __device__ __forceinline__ double operator[](const int j) const {
        const double x = lane_bcast_f64(sa.el[(base + j) >> 6], (base + j) & 63), y = lane_bcast_f64(sb.el[(base + j) >> 6], (base + j) & 63);
        return first ? x : y;
    }
};

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

// ---- what generated code calls --------------------------------------------------------------------------------------------------------
// rigid block (rigidTform3d / extrinsic3D / template_points, fbi:143-211): xout = R xin + t;  E[:, a] = dR/dr_a xin (3 x 3, row-major)
template <bool JAC, typename Slab>
__device__ __forceinline__ void rigid_fwd(const Slab ps, const double (&xin)[3], double (&xout)[3], double (&E)[9]) {
    if constexpr (JAC) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) E[c * 3 + a] = ps[POSE_DR + a * 9 + c * 3 + 0] * xin[0] + ps[POSE_DR + a * 9 + c * 3 + 1] * xin[1] + ps[POSE_DR + a * 9 + c * 3 + 2] * xin[2];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) xout[c] = ps[POSE_R + 3 * c + 0] * xin[0] + ps[POSE_R + 3 * c + 1] * xin[1] + ps[POSE_R + 3 * c + 2] * xin[2] + ps[POSE_T + c];
}
// projection as the FIRST block: its parameter columns are A_p, S = A_x
template <int P>
__device__ __forceinline__ void chain_projection(const double (&Ap)[18], const double (&Ax)[2][3], double (&J)[2 * P], double (&S)[2][3]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int j = 0; j < 9; ++j) J[r * P + j] = Ap[r * 9 + j];
#pragma unroll
        for (int c = 0; c < 3; ++c) S[r][c] = Ax[r][c];
    }
}
// rigid block in the middle: columns [S E | S] at COL0, S <- S R
template <int P, int COL0, typename Slab>
__device__ __forceinline__ void chain_rigid(const double (&S)[2][3], const double (&E)[9], const Slab ps, double (&J)[2 * P], double (&Sn)[2][3]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int a = 0; a < 3; ++a) J[r * P + COL0 + a] = S[r][0] * E[0 * 3 + a] + S[r][1] * E[1 * 3 + a] + S[r][2] * E[2 * 3 + a];
#pragma unroll
        for (int c = 0; c < 3; ++c) J[r * P + COL0 + 3 + c] = S[r][c];
#pragma unroll
        for (int c = 0; c < 3; ++c) Sn[r][c] = S[r][0] * ps[POSE_R + 0 * 3 + c] + S[r][1] * ps[POSE_R + 1 * 3 + c] + S[r][2] * ps[POSE_R + 2 * 3 + c];
    }
}
// template_points as the source: columns [S Q | S]
template <int P, int COL0>
__device__ __forceinline__ void chain_template(const double (&S)[2][3], const double (&Q)[9], double (&J)[2 * P]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int a = 0; a < 3; ++a) J[r * P + COL0 + a] = S[r][0] * Q[0 * 3 + a] + S[r][1] * Q[1 * 3 + a] + S[r][2] * Q[2 * 3 + a];
#pragma unroll
        for (int c = 0; c < 3; ++c) J[r * P + COL0 + 3 + c] = S[r][c];
    }
}
// free_point as the source: d X / d point = I (fbi:234-240)
template <int P, int COL0>
__device__ __forceinline__ void chain_free(const double (&S)[2][3], double (&J)[2 * P]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) J[r * P + COL0 + c] = S[r][c];
}
// user block: Jb = its compute_jac output, NOUT x (NP + NIN) row-major, parameter columns first (afb:738-748's layout).
// Columns S Jb[:, :NP] at COL0; Sn = S Jb[:, NP:] (nothing for a source block, NIN = 0).
template <int P, int COL0, int NP, int NIN, int NOUT>
__device__ __forceinline__ void chain_user(const double (&S)[2][NOUT], const double (&Jb)[NOUT * (NP + NIN)], double (&J)[2 * P], double (&Sn)[2][NIN > 0 ? NIN : 1]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            double t = 0.0;
#pragma unroll
            for (int o = 0; o < NOUT; ++o) t += S[r][o] * Jb[o * (NP + NIN) + p];
            J[r * P + COL0 + p] = t;
        }
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            double t = 0.0;
#pragma unroll
            for (int o = 0; o < NOUT; ++o) t += S[r][o] * Jb[o * (NP + NIN) + NP + i];
            Sn[r][i] = t;
        }
    }
}

// What a generated `Chain::eval` sees of one detection: slabs of the rigid groups (scalar loads when the tile shares camera and image),
// intrinsics, the source point, a user block's parameters of this detection's camera / image / key.
template <bool UNIFORM>
struct ChainCtx {
    const GenericArgs &a;
    int c, im, k;   // UNIFORM: c and im are wave-uniform
    __device__ __forceinline__ int index_of(const int link) const { return link == LINK_CAM ? c : link == LINK_IMG ? im : k; }
    __device__ __forceinline__ auto slab(const int g, const int link) const {
        const double *p = a.slab[g] + (int64_t)index_of(link) * POSE_STRIDE;
        if constexpr (UNIFORM) return ScalarSlab(p);
        else return p;
    }
    __device__ __forceinline__ IntrRow intr() const { return IntrRow{a.prm + a.intr_off + 9 * (int64_t)c}; }
    __device__ __forceinline__ const double *point() const { return a.prm + a.point_off + 3 * (int64_t)k; }
    __device__ __forceinline__ const double *tpoint() const { return a.tmpl + 3 * (int64_t)k; }
    __device__ __forceinline__ const double *user(const int u, const int link, const int np) const { return a.prm + a.user_off[u] + (int64_t)np * index_of(link); }
};

// the same for the one-launch form: the slabs of this detection's (camera, image) pair are held across the wave's lanes
template <typename Spec, bool TWO>
struct ChainCtxLanes {
    const GenericArgs &a;
    const LaneSlabs<Spec> &sa, &sb;   // TWO: the pairs of the tile's first and last lane; else sb is not used
    bool first;
    int c, im, k;
    __device__ __forceinline__ int index_of(const int link) const { return link == LINK_CAM ? c : link == LINK_IMG ? im : k; }
    __device__ __forceinline__ auto slab(const int g, const int) const {
        if constexpr (TWO) return WaveSlab2<Spec>{sa, sb, first, g * POSE_STRIDE};
        else return WaveSlab<Spec>{sa, g * POSE_STRIDE};
    }
    __device__ __forceinline__ IntrRow intr() const { return IntrRow{a.prm + a.intr_off + 9 * (int64_t)c}; }
    __device__ __forceinline__ const double *point() const { return a.prm + a.point_off + 3 * (int64_t)k; }
    __device__ __forceinline__ const double *tpoint() const { return a.tmpl + 3 * (int64_t)k; }
    __device__ __forceinline__ const double *user(const int u, const int link, const int np) const { return a.prm + a.user_off[u] + (int64_t)np * index_of(link); }
};

// one detection of a tile through the generated chain.  ONE (the one-launch step, no slab-preparation launch ran; the host picks it
// for tables in which every tile lies inside a run of one (camera, image) pair or across ONE run boundary — the reference's table
// order): the wave prepares the slabs of the first lane's pair and, if the tile has a second one, of the last lane's pair; a tile
// across a boundary is evaluated once, every lane picking its pair's entries (WaveSlab2).  No loop over pairs: a loop carries J
// across its iterations next to the J being computed, +100 VGPRs and one wave per SIMD instead of two.
template <typename Spec, bool JAC, bool ONE>
__device__ __forceinline__ void eval_tile_lane(const GenericArgs &a, const int lane, const int c, const int im, const int k, double &u, double &v, double (&J)[2 * Spec::P]) {
    if constexpr (ONE && Spec::N_SLABS > 0) {
        const int c0 = __builtin_amdgcn_readfirstlane(c), im0 = __builtin_amdgcn_readfirstlane(im);
        const bool first = c == c0 && im == im0;
        LaneSlabs<Spec> sa, sb;
        sa.prepare(a, c0, im0, lane);
        if (__all(first)) {
            Spec::template eval<JAC>(ChainCtxLanes<Spec, false>{a, sa, sa, true, c, im, k}, u, v, J);
        } else {
            sb.prepare(a, __builtin_amdgcn_readlane(c, 63), __builtin_amdgcn_readlane(im, 63), lane);   // tail lanes repeat the last detection
            Spec::template eval<JAC>(ChainCtxLanes<Spec, true>{a, sa, sb, first, c, im, k}, u, v, J);
        }
        return;
    }
    const int c0 = __builtin_amdgcn_readfirstlane(c), im0 = __builtin_amdgcn_readfirstlane(im);
    if (__all(c == c0 && im == im0)) Spec::template eval<JAC>(ChainCtx<true>{a, c0, im0, k}, u, v, J);   // every slab through scalar loads
    else Spec::template eval<JAC>(ChainCtx<false>{a, c, im, k}, u, v, J);
}

// Fused residual + Jacobian for a generated chain: the hand-fused kernel's tile-per-wave structure, scalar-load slabs when the
// tile shares camera and image, transposed non-temporal stores.  MODE as in ba_eval_kernel (1 residual, 2 Jacobian, 3 both);
// TO = the type the outputs are WRITTEN in (arithmetic is FP64 for every dtype, like the hand-fused kernels).
template <typename Spec, int MODE, typename TO, bool ONE = false>
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
        double u, v;
        double J[P2];
        eval_tile_lane<Spec, JAC, ONE>(a, lane, c, im, k, u, v, J);
        if constexpr (RES) {
            O2 r;
            r.x = (TO)(u - m.x);
            r.y = (TO)(v - m.y);
            __builtin_nontemporal_store(r, valid ? reinterpret_cast<O2 *>(resid) + i : static_cast<O2 *>(a.sink));
        }
        if constexpr (JAC) store_jac_tile<P, TO, true>(J, tr, jac, tile, lane, total_jac);
    }
}

// The same evaluation with the fixed-parameter mask applied AT THE STORE (afb:627-651 on the device; round 3 wrote the dense J
// and gathered it): every lane packs the kept entries of its two rows into the wave-private LDS image at its offset inside the
// tile's contiguous range of the CSR data array, then the wave streams the range out — ba_compact_tile_kernel's store phase with
// a 64-bit keep mask (generated chains reach P = 51).  `a.jac` is the data array here.
template <typename Spec, int MODE, typename TO, bool ONE = false>
__device__ __forceinline__ void generic_compact_body(const GenericArgs &a) {
    constexpr int P = Spec::P;
    constexpr int P2 = 2 * P;
    constexpr bool RES = (MODE & MODE_RESID) != 0;
    constexpr int VS = 16 / sizeof(TO);
    constexpr int LINE = 128 / sizeof(TO);           // scalars per 128-byte line
    constexpr int WAVE_LDS = HALF * P2 + LINE + 64;  // packed range + alignment shift + one dummy slot per lane
    using O2 = typename Vec2<TO>::type;
    using V16 = __attribute__((ext_vector_type(VS))) TO;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6, lane = threadIdx.x & 63;
    TO *tr = reinterpret_cast<TO *>(smem_raw) + wave * ((WAVE_LDS + VS - 1) / VS * VS);
    TO *dummy = tr + HALF * P2 + LINE + lane;
    TO *resid = static_cast<TO *>(a.resid);
    TO *data = static_cast<TO *>(a.jac);
    const int64_t tile0 = (int64_t)blockIdx.x * a.tiles_per_wg;
    const int64_t tile1 = min(tile0 + (int64_t)a.tiles_per_wg, a.n_tiles);
    for (int64_t tile = tile0 + wave; tile < tile1; tile += n_waves) {
        const int64_t i = tile * TILE + lane;
        const bool valid = i < a.n;
        const int64_t ic = valid ? i : a.n - 1;
        int c, im, k;
        load_indices(a.tab, ic, c, im, k);
        const double2v m = load_uv(a.tab, ic);
        const uint64_t keep_raw = a.keep[ic];      // requested with the indices: after the evaluation they would cost a memory latency per tile
        const int64_t off_raw = a.row_off[ic];
        asm volatile("" ::: "memory");
        double u, v;
        double J[P2];
        eval_tile_lane<Spec, true, ONE>(a, lane, c, im, k, u, v, J);
        if constexpr (RES) {
            O2 r;
            r.x = (TO)(u - m.x);
            r.y = (TO)(v - m.y);
            __builtin_nontemporal_store(r, valid ? reinterpret_cast<O2 *>(resid) + i : static_cast<O2 *>(a.sink));
        }
        const uint64_t keep = valid ? keep_raw : 0ull;
        const int cnt = __popcll(keep);
        const int64_t off = off_raw + (valid ? 0 : 2 * (int64_t)__popcll(keep_raw));  // tail lanes: end of data
        const int64_t off0 = __shfl(off, 0);                  // first entry of the tile
        const int lo = (int)(off - off0);                     // this detection's offset inside the tile range
        const int mid = __shfl(lo, HALF);                     // pass boundary
        const int end = __shfl(lo + 2 * cnt, TILE - 1);       // tile range length
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int s0 = h ? mid : 0;
            const int len = (h ? end : mid) - s0;
            TO *g0 = data + off0 + s0;                                             // first global element of the pass
            const int mis = (int)((reinterpret_cast<uintptr_t>(g0) / sizeof(TO)) & (LINE - 1));
            const bool mine = (lane >> 5) == h;
            TO *ru = tr + mis + (lo - s0);
            TO *rv = ru + cnt;
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const bool on = mine && ((keep >> j) & 1ull);
                const int pos = __popcll(keep & ((1ull << j) - 1ull));
                *(on ? ru + pos : dummy) = (TO)J[j];
                *(on ? rv + pos : dummy) = (TO)J[P + j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // stream [mis, mis + len) of the LDS image to g0 - mis + [mis, mis + len): whole 16-byte units where the unit lies
            // inside the range, scalars at the ragged ends (the image is shifted by `mis`, so units are line-aligned in memory)
            TO *gbase = g0 - mis;
            const int first = mis / VS, last = (mis + len + VS - 1) / VS;   // units [first, last)
            for (int q = first + lane; q < last; q += 64) {
                const int e0 = q * VS;
                if (e0 >= mis && e0 + VS <= mis + len) {
                    __builtin_nontemporal_store(*reinterpret_cast<const V16 *>(tr + e0), reinterpret_cast<V16 *>(gbase + e0));
                } else {
#pragma unroll
                    for (int e = 0; e < VS; ++e)
                        if (e0 + e >= mis && e0 + e < mis + len) gbase[e0 + e] = tr[e0 + e];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// What chain_compiler.py appends after the ChainSpec struct it emits: the entry points of the code object — every kernel in the
// two-launch form (behind pcs_genchain_prep) and in the one-launch form (`_one`).
#define PCS_GENCHAIN_ENTRY_POINTS(Spec)                                                                                              \
    extern "C" __global__ void pcs_genchain_prep(const pcs::GenericArgs a) { pcs::generic_slab_prep_body(a); }                        \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_1(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 1, double, false>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_2(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 2, double, false>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_3(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 3, double, false>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_1_f32(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 1, float, false>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_2_f32(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 2, float, false>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_3_f32(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 3, float, false>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_compact_2(const pcs::GenericArgs a) { pcs::generic_compact_body<Spec, 2, double, false>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_compact_3(const pcs::GenericArgs a) { pcs::generic_compact_body<Spec, 3, double, false>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_compact_2_f32(const pcs::GenericArgs a) { pcs::generic_compact_body<Spec, 2, float, false>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_compact_3_f32(const pcs::GenericArgs a) { pcs::generic_compact_body<Spec, 3, float, false>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_1_one(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 1, double, true>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_2_one(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 2, double, true>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_3_one(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 3, double, true>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_1_f32_one(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 1, float, true>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_2_f32_one(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 2, float, true>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_eval_3_f32_one(const pcs::GenericArgs a) { pcs::generic_eval_body<Spec, 3, float, true>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_compact_2_one(const pcs::GenericArgs a) { pcs::generic_compact_body<Spec, 2, double, true>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_compact_3_one(const pcs::GenericArgs a) { pcs::generic_compact_body<Spec, 3, double, true>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_compact_2_f32_one(const pcs::GenericArgs a) { pcs::generic_compact_body<Spec, 2, float, true>(a); } \
    extern "C" __global__ __launch_bounds__(256) void pcs_genchain_compact_3_f32_one(const pcs::GenericArgs a) { pcs::generic_compact_body<Spec, 3, float, true>(a); }

}  // namespace pcs
