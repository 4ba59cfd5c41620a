// ba_kernels.hpp — the HIP kernels of the bundle-adjustment engine (gfx950 / CDNA4 only).
//
//   slab_prep_kernel         K0: one thread per slab element -> R, t, dR/dr slabs of every camera / pose (ba_device.hpp);
//                            also copies the 3-D points of chains SELF / FREE out of the parameter string.  Small tables
//                            skip it: ba_eval_kernel<..., PREP = true> prepares the slabs of its tile per wave (one launch per step).
//   ba_eval_kernel           K1-K4: fused residual + dense 2xP Jacobian block per detection.  One lane
//                            owns one detection of a 64-detection tile; slabs + points are read through
//                            L1/L2 or staged in LDS; the Jacobian tile is transposed through LDS so that
//                            every store instruction writes 1 KiB of consecutive addresses.
//   ba_compact_kernel / ba_compact_tile_kernel
//                            same maths, only unfixed columns, in CSR data order.
//   legacy_cost_kernel       pre-bundle residual-only cost.
//   membench_kernel          streaming probes for the roofline comparison.
// Arithmetic and slabs are FP64 in every kernel (the reference's precision, fbi:11); what varies is the type the
// residual / Jacobian are WRITTEN in (TO: double, or float for PCS_F32 / PCS_MIXED) and the type the measurements
// are READ in (DetTable::uv_f32).  An all-float arithmetic path existed in round 1 and was removed: the chain rule
// cancels, and single-precision sums left 5e-3 relative error in Jacobian entries.
// The launch plumbing and the C ABI are in pcs_engine.hip.
#pragma once
#ifndef __HIPCC_RTC__   // a chain compiled by hiprtc (pycamset_amd/chain_compiler.py): the HIP runtime declarations are pre-included, host headers do not exist
#include <hip/hip_runtime.h>

#include <cstdint>
#else
#include "ba_rtc_prelude.hpp"
#endif

#include "ba_device.hpp"

namespace pcs {

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
// clang ext vectors (the nontemporal builtins reject HIP's double2 / float2 structs)
template <typename S> struct Vec2 { using type = __attribute__((ext_vector_type(2))) S; };

constexpr int MODE_RESID = 1;
constexpr int MODE_JAC = 2;

constexpr int VAR_SLAB_LDS = 1;   // stage slabs + points in LDS
constexpr int VAR_TRANSPOSE = 2;  // transpose the Jacobian tile through LDS, coalesced stores
constexpr int VAR_NT = 4;         // non-temporal output stores

constexpr int WG_THREADS = 256;   // largest workgroup; small tables are launched with fewer waves per workgroup
constexpr int WAVES_PER_WG = WG_THREADS / 64;
constexpr int TILE = 64;       // detections per wave tile
constexpr int HALF = 32;       // detections per transpose pass

using T = double;   // arithmetic and slab type of every kernel

// Row stride (scalars) of the wave-private LDS image the fused kernel transposes through.  Each lane
// writes its own 2P-scalar row with 16-byte (f64) / 8-byte (f32) stores, so the stride decides the bank
// pattern: 2P = 42 (template) is conflict-free as it is, but 2P = 48 (self) puts lanes l and l+2 on the
// same banks (96 dwords = 32 mod 64: 8-way conflict) and 2P = 36 (free) lanes l and l+8.  A stride of
// 2 mod 4 scalars (f64) spreads 16 lanes over all 64 banks; f32 keeps strides that are a multiple of
// the 16-byte read unit.
constexpr int lds_row_stride(int p2, int esize) {
    return esize == 8 ? (p2 % 4 == 0 ? p2 + 2 : p2) : (p2 % 16 == 0 ? p2 + 4 : p2);
}

struct EvalArgs {
    DetTable tab;           // indices + measurements (ba_device.hpp)
    const void *cam_slab;   // n_cams x CAM_STRIDE
    const void *pose_slab;  // n_imgs x POSE_STRIDE
    const void *points;     // n_keys x 3 (padded)
    void *resid;            // N x 2
    void *jac;              // 2N x P
    int64_t n;
    int32_t n_cams, n_imgs, n_keys;
    int32_t tiles_per_wg;
    int32_t xcd_remap;      // 1: workgroups that share an XCD (blockIdx % 8) take one contiguous eighth of the tiles
    int64_t n_tiles;
    void *sink;              // 64 B scratch: tail lanes store their (unused) residual here, so the store needs no branch
    // compaction (ba_compact_* only)
    const uint32_t *keep;    // per detection: bit j set = local column j is free
    const int64_t *row_off;  // per detection: offset of its u row in the CSR data array
    // one-launch step (PREP kernels only): the parameter string itself and its group offsets
    const double *prm;
    int64_t extr_off, pose_off, point_off;
};

// Slab of ONE (camera, image) pair prepared by the wave that needs it, in a wave-private LDS strip laid out like the
// global slabs: [0, CAM_STRIDE) camera | [CAM_STRIDE, CAM_STRIDE + POSE_STRIDE) pose.  72 rotation elements on 64 lanes:
// lanes 0-35 take the camera's, lanes 36-63 the pose's first 28, lanes 36-43 also its last 8 (same rot_terms); the 15 plain
// copies (intrinsics, two translations) ride on lanes 0-14.  ~1 sincos + 2 element formulas per lane and 4 L2-resident loads:
// ~0.4 us on a wave's critical path against 4.5 us + a launch gap for slab_prep_kernel — what a table of <= ~4e5
// detections (every real pyCamSet calibration, config 2, an 8-way shard of config 3) spends a third of its step on.
constexpr int PAIR_SLAB = CAM_STRIDE + POSE_STRIDE;
template <int CHAIN>
__device__ __forceinline__ void prep_pair_slab(T *__restrict__ strip, const double *__restrict__ prm, const int c0, const int im0, const int lane,
                                               const int64_t extr_off, const int64_t pose_off) {
    constexpr bool HAS_POSE = CHAIN != CHAIN_FREE;
    const bool on_pose = HAS_POSE && lane >= 36;
    const double *p6 = on_pose ? prm + pose_off + 6 * (int64_t)im0 : prm + extr_off + 6 * (int64_t)c0;
    const double r0 = p6[0], r1 = p6[1], r2 = p6[2];
    // plain copies: lanes 0-8 intrinsics, 9-11 t_e, 12-14 t_p
    const double *cp = lane < 9 ? prm + 9 * (int64_t)c0 + lane : lane < 12 ? prm + extr_off + 6 * (int64_t)c0 + 3 + (lane - 9) : prm + pose_off + 6 * (int64_t)im0 + 3 + (lane - 12);
    const bool copies = lane < (HAS_POSE ? 15 : 12);
    const double cv = principal_or_nan(*(copies ? cp : prm), lane, prm[9 * (int64_t)c0], prm[9 * (int64_t)c0 + 2]);
    const RotTerms t = rot_terms(r0, r1, r2);
    const int q = on_pose ? lane - 36 : min(lane, 35);
    const double e1 = rot_element(t, q);
    if (on_pose) strip[CAM_STRIDE + pose_slot_of(q)] = e1;
    else if (lane < 36) strip[cam_slot_of(q)] = e1;
    if constexpr (HAS_POSE) {
        const double e2 = rot_element(t, min(q + 28, 35));
        if (lane >= 36 && lane < 44) strip[CAM_STRIDE + pose_slot_of(q + 28)] = e2;
    }
    if (copies) strip[lane < 9 ? lane : lane < 12 ? CAM_T + (lane - 9) : CAM_STRIDE + POSE_T + (lane - 12)] = cv;
}

// ---------------------------------------------------------------------------------------------
// K0  slab preparation
// ---------------------------------------------------------------------------------------------
// param_str layout: afb make_param_struct (abstract_function_blocks.py:777-820), see pcs_hip.h.
// One thread per SLAB ELEMENT (round 3; one thread per camera / pose before): thread t of n_threads owns element
// t of [n_cams x CAM_STRIDE | n_imgs x POSE_STRIDE] — a copy of a parameter (intrinsics, translations), or one entry of
// R / dR/dr through rot_terms + rot_element (ba_device.hpp), the same two functions the evaluation kernels use when they
// prepare their slabs themselves (PREP), so both paths hold the same bits — and a strided share of the point copy.
__host__ __device__ inline int64_t slab_prep_threads(int64_t n_cams, int64_t n_imgs, int has_pose) {
    return (int64_t)n_cams * CAM_STRIDE + (has_pose ? (int64_t)n_imgs * POSE_STRIDE : 0);
}
__device__ __forceinline__ void slab_prep_element(const int64_t t, const int64_t n_threads, const double *__restrict__ prm, T *__restrict__ cam_slab,
                                                  T *__restrict__ pose_slab, T *__restrict__ points, int n_cams, int n_imgs, int n_keys,
                                                  int64_t extr_off, int64_t pose_off, int64_t point_off, int has_pose, int copy_points) {
    const int64_t n_cam_el = (int64_t)n_cams * CAM_STRIDE;
    if (t < n_cam_el) {
        const int64_t c = t / CAM_STRIDE;
        const int slot = (int)(t - c * CAM_STRIDE);
        const double *p6 = prm + extr_off + 6 * c;
        T v;
        if (slot < CAM_R) v = principal_or_nan(prm[9 * c + slot], slot, prm[9 * c], prm[9 * c + 2]);
        else if (slot >= CAM_T && slot < CAM_DR) v = p6[3 + slot - CAM_T];
        else v = rot_element(rot_terms(p6[0], p6[1], p6[2]), slot < CAM_T ? slot - CAM_R : 9 + slot - CAM_DR);
        cam_slab[t] = v;
    } else if (has_pose && t < n_cam_el + (int64_t)n_imgs * POSE_STRIDE) {
        const int64_t u = t - n_cam_el;
        const int64_t im = u / POSE_STRIDE;
        const int slot = (int)(u - im * POSE_STRIDE);
        const double *p6 = prm + pose_off + 6 * im;
        T v;
        if (slot >= POSE_T && slot < POSE_DR) v = p6[3 + slot - POSE_T];
        else if (slot == POSE_STRIDE - 1) v = T(0);
        else v = rot_element(rot_terms(p6[0], p6[1], p6[2]), slot < POSE_T ? slot - POSE_R : 9 + slot - POSE_DR);
        pose_slab[u] = v;
    }
    if (copy_points) {
        const int64_t total = (int64_t)n_keys * 3;
        for (int64_t j = t; j < total; j += n_threads) points[j] = prm[point_off + j];
    }
}

__global__ void slab_prep_kernel(const double *__restrict__ prm, T *__restrict__ cam_slab, T *__restrict__ pose_slab,
                                 T *__restrict__ points, int n_cams, int n_imgs, int n_keys, int64_t extr_off,
                                 int64_t pose_off, int64_t point_off, int has_pose, int copy_points) {
    slab_prep_element((int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, prm, cam_slab, pose_slab, points, n_cams, n_imgs,
                      n_keys, extr_off, pose_off, point_off, has_pose, copy_points);
}

// Prologue of a normal-equations build in ONE launch: the first prep_blocks workgroups prepare the slabs, the others zero
// H (n_h doubles, 16-byte aligned), g and the cost — instead of three hipMemsetAsync and a slab_prep launch (12-15 us of
// launch gaps per build).
__global__ void normal_prologue_kernel(const double *__restrict__ prm, T *__restrict__ cam_slab, T *__restrict__ pose_slab,
                                       T *__restrict__ points, int n_cams, int n_imgs, int n_keys, int64_t extr_off,
                                       int64_t pose_off, int64_t point_off, int has_pose, int copy_points, int prep_blocks,
                                       double *__restrict__ Hm, int64_t n_h, double *__restrict__ g, int64_t n_g, double *__restrict__ cost,
                                       const int32_t *__restrict__ stop, const int32_t *__restrict__ sel = nullptr, int64_t alt_prm = 0, int64_t alt_out = 0) {
    if (stop && *stop) return;   // a build queued behind the end of an LM loop (ba_schur.hpp PCS_STOP_GUARD)
    if (sel && *sel) {           // LM loop with two states: the trial state is the other pair (string, packed buffer) — ba_schur.hpp SchurArgs::sel
        prm += alt_prm;
        Hm += alt_out; g += alt_out; cost += alt_out;
    }
    if ((int)blockIdx.x < prep_blocks) {
        slab_prep_element((int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)prep_blocks * blockDim.x, prm, cam_slab, pose_slab, points, n_cams,
                          n_imgs, n_keys, extr_off, pose_off, point_off, has_pose, copy_points);
        return;
    }
    using D2 = typename Vec2<double>::type;
    const int64_t t = (int64_t)(blockIdx.x - prep_blocks) * blockDim.x + threadIdx.x;
    const int64_t nt = (int64_t)(gridDim.x - prep_blocks) * blockDim.x;
    D2 *h2 = reinterpret_cast<D2 *>(Hm);
    for (int64_t i = t; i < n_h / 2; i += nt) __builtin_nontemporal_store(D2{0.0, 0.0}, h2 + i);
    if (t == 0 && (n_h & 1)) Hm[n_h - 1] = 0.0;
    for (int64_t i = t; i < n_g; i += nt) g[i] = 0.0;
    if (t == 0) *cost = 0.0;
}

// The coalesced store phase of a 64-detection tile (VAR_TRANSPOSE): every lane holds the 2P values of its detection in J;
// the wave streams the tile out as 16-byte units at consecutive addresses through its private LDS region `tr`.  Shared by
// ba_eval_kernel and the generated chain kernels (ba_generic.hpp), which therefore write with the same instruction stream.
template <int P, typename TO, bool NT>
__device__ __forceinline__ void store_jac_tile(const T (&J)[2 * P], TO *tr, TO *jac, const int64_t tile, const int lane, const int64_t total_jac) {
    constexpr int P2 = 2 * P;
    constexpr int LROW = lds_row_stride(P2, (int)sizeof(TO));
    using O2 = typename Vec2<TO>::type;
    // Two passes of 32 detections: the active half writes its 2P values row-major into
    // the wave-private LDS region, then all 64 lanes stream the region out in 16-byte
    // units at consecutive addresses.  Same-wave LDS ops execute in order; the
    // wavefront-scope fences only stop the compiler from reordering across them.
    constexpr int VS = 16 / sizeof(TO);  // scalars per 16-byte unit
    using V16 = __attribute__((ext_vector_type(VS))) TO;
    constexpr int UNITS = HALF * P2 / VS;
    static_assert((HALF * P2) % VS == 0, "half tile must be a whole number of 16-byte units");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((lane >> 5) == h) {
            O2 *dst = reinterpret_cast<O2 *>(tr + (lane & 31) * LROW);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                O2 w;
                w.x = (TO)J[2 * j];
                w.y = (TO)J[2 * j + 1];
                dst[j] = w;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int64_t base = (tile * TILE + h * HALF) * (int64_t)P2;  // scalar offset, multiple of VS
        // The padded-row index maths is tile-invariant; left alone, hipcc hoists all of it out of
        // the tile loop and the kernel grows from 164 to 244 VGPRs.  An opaque copy of the lane
        // id keeps it inside (148 VGPRs for the self chain).
        int lane_v = lane;
        if constexpr (LROW != P2) asm volatile("" : "+v"(lane_v));
        auto lds_unit = [&](const int q) {  // unit index inside the (possibly padded) LDS image
            if constexpr (LROW != P2) {
                static_assert(P2 % VS == 0 && LROW % VS == 0, "padded rows must hold whole 16-byte units");
                const int row = q / (P2 / VS);
                return row * (LROW / VS) + (q - row * (P2 / VS));
            } else {
                return q;
            }
        };
        if (base + (int64_t)HALF * P2 <= total_jac) {
            // whole half tile inside the array (every tile but the last): no per-unit range checks, and
            // the LDS read of unit u + 1 is issued before unit u is stored (left alone hipcc pairs every
            // ds_read_b128 with an s_waitcnt 0 right before its store)
            constexpr int NU = (UNITS + 63) / 64;
            V16 w[2];
            w[0] = reinterpret_cast<const V16 *>(tr)[lds_unit(lane_v)];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int qn = (u + 1) * 64 + lane_v;
                if (u + 1 < NU) {
                    if ((u + 2) * 64 <= UNITS || qn < UNITS) w[(u + 1) & 1] = reinterpret_cast<const V16 *>(tr)[lds_unit(qn)];
                }
                asm volatile("" ::: "memory");
                const int q = u * 64 + lane_v;
                if ((u + 1) * 64 <= UNITS || q < UNITS) {
                    V16 *dst = reinterpret_cast<V16 *>(jac + base + (int64_t)q * VS);
                    if constexpr (NT) __builtin_nontemporal_store(w[u & 1], dst);
                    else *dst = w[u & 1];
                }
            }
        } else {
#pragma unroll
            for (int q0 = 0; q0 < UNITS; q0 += 64) {
                const int q = q0 + lane_v;
                if (q < UNITS) {
                    const int64_t e = base + (int64_t)q * VS;
                    const int lq = lds_unit(q);
                    if (e + VS <= total_jac) {
                        const V16 w = reinterpret_cast<const V16 *>(tr)[lq];
                        if constexpr (NT) __builtin_nontemporal_store(w, reinterpret_cast<V16 *>(jac + e));
                        else *reinterpret_cast<V16 *>(jac + e) = w;
                    } else {
                        for (int s = 0; s < VS; ++s)
                            if (e + s < total_jac) jac[e + s] = tr[lq * VS + s];
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---------------------------------------------------------------------------------------------
// K1-K4  fused residual + Jacobian
// ---------------------------------------------------------------------------------------------
// (The float-output kernels are held to 168 VGPRs = three waves per SIMD, which their halved store stream can use — except for
// the self chain, whose 24 columns spilled 36-84 B under that limit: it keeps its registers and runs two waves per SIMD.)
// TO = type of the residual and Jacobian written out.  The workgroup has blockDim.x / 64 waves (4 for large tables,
// fewer for small ones so that the grid still covers every CU several times); wave w takes tiles w, w + waves, ...
// of the workgroup's run of tiles.
// PREP (round 3, "one launch per step"): no slab_prep_kernel ran; the wave prepares the slab of every (camera, image) pair
// of its tile itself (prep_pair_slab) and evaluates the pair's detections from that LDS strip, pair after pair — one
// iteration for a tile inside a run of the reference's table order, two for a tile that straddles a run boundary.  Points
// of chains SELF / FREE come straight from the parameter string.
template <int CHAIN, int MODE, int VARIANT, typename TO, bool PREP = false>
__global__ __launch_bounds__(WG_THREADS, (sizeof(TO) == 4 && (MODE & MODE_JAC) && !PREP && CHAIN != CHAIN_SELF) ? 3 : 1) void ba_eval_kernel(const EvalArgs a) {
    constexpr int P = chain_P(CHAIN);
    constexpr int P2 = 2 * P;
    constexpr bool SLAB_LDS = (VARIANT & VAR_SLAB_LDS) != 0;
    static_assert(!(PREP && SLAB_LDS), "a wave that prepares its own slabs does not stage the global ones");
    constexpr bool TRANSPOSE = (VARIANT & VAR_TRANSPOSE) != 0;
    constexpr bool NT = (VARIANT & VAR_NT) != 0;
    constexpr bool JAC = (MODE & MODE_JAC) != 0;
    constexpr bool RES = (MODE & MODE_RESID) != 0;
    using O2 = typename Vec2<TO>::type;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *smem = reinterpret_cast<T *>(smem_raw);

    const T *cam_slab = static_cast<const T *>(a.cam_slab);
    const T *pose_slab = static_cast<const T *>(a.pose_slab);
    const T *points = static_cast<const T *>(a.points);
    const int n_threads = blockDim.x;
    int lds_used = 0;  // scalars
    if constexpr (SLAB_LDS) {
        const int n_cam_sc = a.n_cams * CAM_STRIDE;
        const int n_pose_sc = (CHAIN != CHAIN_FREE) ? a.n_imgs * POSE_STRIDE : 0;
        const int n_pt_sc = (a.n_keys * 3 + 3) & ~3;  // device buffer is padded to a multiple of 4 scalars
        // 16-byte cooperative copies (all three regions are multiples of 16 bytes)
        constexpr int VS = 16 / sizeof(T);
        using V16 = __attribute__((ext_vector_type(VS))) T;
        const V16 *g0 = reinterpret_cast<const V16 *>(cam_slab);
        V16 *l0 = reinterpret_cast<V16 *>(smem);
        for (int i = threadIdx.x; i < n_cam_sc / VS; i += n_threads) l0[i] = g0[i];
        const V16 *g1 = reinterpret_cast<const V16 *>(pose_slab);
        V16 *l1 = reinterpret_cast<V16 *>(smem + n_cam_sc);
        for (int i = threadIdx.x; i < n_pose_sc / VS; i += n_threads) l1[i] = g1[i];
        const V16 *g2 = reinterpret_cast<const V16 *>(points);
        V16 *l2 = reinterpret_cast<V16 *>(smem + n_cam_sc + n_pose_sc);
        for (int i = threadIdx.x; i < n_pt_sc / VS; i += n_threads) l2[i] = g2[i];
        cam_slab = smem;
        pose_slab = smem + n_cam_sc;
        points = smem + n_cam_sc + n_pose_sc;
        lds_used = n_cam_sc + n_pose_sc + n_pt_sc;
        __syncthreads();
    }
    const int wave = threadIdx.x >> 6;
    const int n_waves = n_threads >> 6;
    const int lane = threadIdx.x & 63;
    T *strip = smem;   // PREP: wave-private pair slab
    if constexpr (PREP) {
        strip = smem + wave * PAIR_SLAB;
        lds_used = n_waves * PAIR_SLAB;
        if constexpr (CHAIN != CHAIN_TEMPLATE) points = a.prm + a.point_off;
    }
    constexpr int LROW = lds_row_stride(P2, (int)sizeof(TO));
    TO *tr = reinterpret_cast<TO *>(smem + lds_used) + wave * (HALF * LROW);  // wave-private transpose region (TRANSPOSE only)

    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD group).  With
    // xcd_remap each group walks one contiguous eighth of the table (bijective remap, any grid size);
    // this is a locality experiment only — the stream has no inter-workgroup reuse beyond the slabs.
    int64_t wg = blockIdx.x;
    if (a.xcd_remap) {
        const int64_t nwg = gridDim.x, q = nwg / 8, r = nwg % 8, x = wg % 8;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + wg / 8;
    }
    const int64_t tile0 = wg * a.tiles_per_wg;
    const int64_t tile1 = min(tile0 + (int64_t)a.tiles_per_wg, a.n_tiles);
    TO *resid = static_cast<TO *>(a.resid);
    TO *jac = static_cast<TO *>(a.jac);
    const int64_t total_jac = a.n * (int64_t)P2;

    for (int64_t tile = tile0 + wave; tile < tile1; tile += n_waves) {
        const int64_t i = tile * TILE + lane;
        const bool valid = i < a.n;
        const int64_t ic = valid ? i : a.n - 1;  // tail lanes recompute the last detection, store nothing
        int c, im, k;
        load_indices(a.tab, ic, c, im, k);
        const double2v m = load_uv(a.tab, ic);
        const T *cs = cam_slab + c * CAM_STRIDE;
        const T *ps = pose_slab + im * POSE_STRIDE;
        const T X0 = points[3 * k], X1 = points[3 * k + 1], X2 = points[3 * k + 2];
        T u, v;
        T J[P2];
        if constexpr (PREP) {
            bool todo = true;   // tail lanes repeat the last detection: they belong to its pair
            while (true) {
                const uint64_t rem = __ballot(todo);
                if (!rem) break;
                const int src = __builtin_ctzll(rem);
                const int c0 = __builtin_amdgcn_readlane(c, src);
                const int im0 = CHAIN != CHAIN_FREE ? __builtin_amdgcn_readlane(im, src) : 0;
                prep_pair_slab<CHAIN>(strip, a.prm, c0, im0, lane, a.extr_off, a.pose_off);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const bool mine = todo && c == c0 && (CHAIN == CHAIN_FREE || im == im0);
                if (mine) eval_detection<CHAIN, T, JAC>(static_cast<const T *>(strip), static_cast<const T *>(strip + CAM_STRIDE), X0, X1, X2, u, v, J);
                todo = todo && !mine;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();   // the next pair overwrites the strip
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        } else if constexpr (!JAC && !SLAB_LDS) {
            // Residual only: 36-44 B of traffic per detection; what the launch is short of is issue slots, not bandwidth (a plain
            // copy of the same bytes takes 6.7 us).  When the tile shares its camera and image (the reference's table order) the
            // 21 + 12 slab scalars come through SCALAR loads (ScalarSlab): 13.6 us with per-lane loads, 10.0 with one coalesced
            // load + v_readlane broadcasts (round 1), 9.0 now; other tiles take the per-lane loads.
            // (The same idea for the fused kernel — staging a tile's slabs in a wave-private LDS strip —
            // was measured and dropped: the second code path costs 12-36 VGPRs and the HBM-bound kernel
            // gains nothing, profiles/r01/sweeps.md.)
            const int c0 = __builtin_amdgcn_readfirstlane(c), im0 = __builtin_amdgcn_readfirstlane(im);
            if (__all(c == c0 && im == im0)) {
                eval_detection<CHAIN, T, false>(ScalarSlab(cam_slab + c0 * CAM_STRIDE), ScalarSlab(pose_slab + im0 * POSE_STRIDE), X0, X1, X2, u, v, J);
            } else {
                eval_detection<CHAIN, T, false>(cs, ps, X0, X1, X2, u, v, J);
            }
        } else if constexpr (JAC && !SLAB_LDS) {
            // Fused kernel, slabs through L1/L2: a tile with one camera and one image (the reference's table order: 156
            // detections per (camera, image) on rig-32) takes its 48 + 39 slab values through SCALAR loads (ScalarSlab) instead of
            // 87 vector loads in which all 64 lanes ask for the same address.  That frees the vector memory path for the store
            // stream: 66.3 -> 64.6 us (chain T), 70.7 -> 65.3 us (chain S), f32 outputs 35.3 -> 33.0 us, rig-128 345 -> 318 us,
            // interleaved on one box (profiles/r02/sweeps.md).  (The round-1 / early round-2 attempts held the slab across lanes
            // and paid 174 v_readlane per tile for it: no gain.)  Other tiles take the per-lane loads; same arithmetic, same bits.
            const int c0 = __builtin_amdgcn_readfirstlane(c), im0 = __builtin_amdgcn_readfirstlane(im);
            const bool first = c == c0 && im == im0;
            if (__all(first)) {
                eval_detection<CHAIN, T, JAC>(ScalarSlab(cam_slab + c0 * CAM_STRIDE), ScalarSlab(pose_slab + im0 * POSE_STRIDE), X0, X1, X2, u, v, J);
            } else {
                // a tile that straddles ONE run boundary (41 % of the tiles on rig-32): its two parts are uniform, each takes its own
                // scalar slabs under its half of the exec mask (chain S 80.5 -> 75.8 us, chain T 68.7 -> 67.9 us on a slow box)
                const int c1 = __builtin_amdgcn_readlane(c, 63), im1 = __builtin_amdgcn_readlane(im, 63);
                if (__all(first || (c == c1 && im == im1))) {
                    if (first) eval_detection<CHAIN, T, JAC>(ScalarSlab(cam_slab + c0 * CAM_STRIDE), ScalarSlab(pose_slab + im0 * POSE_STRIDE), X0, X1, X2, u, v, J);
                    else eval_detection<CHAIN, T, JAC>(ScalarSlab(cam_slab + c1 * CAM_STRIDE), ScalarSlab(pose_slab + im1 * POSE_STRIDE), X0, X1, X2, u, v, J);
                } else {
                    eval_detection<CHAIN, T, JAC>(cs, ps, X0, X1, X2, u, v, J);
                }
            }
        } else {
            eval_detection<CHAIN, T, JAC>(cs, ps, X0, X1, X2, u, v, J);
        }
        if constexpr (RES) {
            // Branch-free: a conditional block here splits the basic block and makes hipcc keep the whole
            // slab + Jacobian live across it (226 VGPRs instead of 150); tail lanes write to a sink.
            O2 r;
            r.x = (TO)(u - m.x);   // afb:384  losses = projected - measured
            r.y = (TO)(v - m.y);
            O2 *rp = valid ? reinterpret_cast<O2 *>(resid) + i : static_cast<O2 *>(a.sink);
            if constexpr (NT) __builtin_nontemporal_store(r, rp);
            else *rp = r;
        }
        if constexpr (JAC) {
            if constexpr (!TRANSPOSE) {
                if (valid) {
                    O2 *row = reinterpret_cast<O2 *>(jac + i * P2);
#pragma unroll
                    for (int j = 0; j < P; ++j) {
                        O2 w;
                        w.x = (TO)J[2 * j];
                        w.y = (TO)J[2 * j + 1];
                        if constexpr (NT) __builtin_nontemporal_store(w, row + j); else row[j] = w;
                    }
                }
            } else {
                store_jac_tile<P, TO, NT>(J, tr, jac, tile, lane, total_jac);
            }
        }
    }
}

// Fixed-parameter compaction (replaces `data[:n_elements][good_mask]`, afb:627-651): every lane
// writes the kept entries of its two rows at the static CSR offsets.  Reads slabs through L1/L2.
// First version, kept as compact_variant = 0 for A/B against the tile kernel (FP64 outputs only).
template <int CHAIN, int MODE>
__global__ __launch_bounds__(WG_THREADS) void ba_compact_kernel(const EvalArgs a) {
    constexpr int P = chain_P(CHAIN);
    constexpr int P2 = 2 * P;
    const T *cam_slab = static_cast<const T *>(a.cam_slab);
    const T *pose_slab = static_cast<const T *>(a.pose_slab);
    const T *points = static_cast<const T *>(a.points);
    T *resid = static_cast<T *>(a.resid);
    T *data = static_cast<T *>(a.jac);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        int c, im, k;
        load_indices(a.tab, i, c, im, k);
        const double2v m = load_uv(a.tab, i);
        T u, v;
        T J[P2];
        eval_detection<CHAIN, T, (MODE & MODE_JAC) != 0>(cam_slab + c * CAM_STRIDE, pose_slab + im * POSE_STRIDE, points[3 * k],
                                                          points[3 * k + 1], points[3 * k + 2], u, v, J);
        if constexpr ((MODE & MODE_RESID) != 0) {
            double2v r;
            r.x = u - m.x;
            r.y = v - m.y;
            reinterpret_cast<double2v *>(resid)[i] = r;
        }
        if constexpr ((MODE & MODE_JAC) != 0) {
            const uint32_t keep = a.keep[i];
            const int cnt = __popc(keep);
            T *ru = data + a.row_off[i];
            T *rv = ru + cnt;
            int o = 0;
#pragma unroll
            for (int j = 0; j < P; ++j) {
                if (keep & (1u << j)) {
                    ru[o] = J[j];
                    rv[o] = J[P + j];
                    ++o;
                }
            }
        }
    }
}

// Coalesced fixed-parameter compaction: tile-per-wave like ba_eval_kernel.  The kept entries of a
// tile form one contiguous range of the CSR data array ([row_off[first], row_off[last] + 2*cnt));
// every lane packs its two rows into the wave-private LDS region at its offset inside that range
// (two passes of 32 detections), then the wave streams the range out at consecutive addresses.
//  * The LDS image is shifted by the offset of the range's first global element inside its 128-byte
//    line, so that the 1 KiB window of every store instruction starts on a line boundary: each
//    instruction writes 8 whole lines (a window that straddles lines leaves one line in eight written
//    by two different non-temporal instructions, which cost 20 us at N = 1e6); only the ragged first /
//    last unit of a pass uses scalar stores.
//  * Packing is branch-free: entry j goes to slot popcount(keep & ((1 << j) - 1)) of its row, or to a
//    per-lane dummy slot when the column is fixed or the lane belongs to the other pass (conditional
//    blocks around the 2P stores would keep the whole Jacobian live in registers, see ba_eval_kernel).
template <int CHAIN, int MODE, typename TO>
__global__ __launch_bounds__(WG_THREADS) void ba_compact_tile_kernel(const EvalArgs a) {
    constexpr int P = chain_P(CHAIN);
    constexpr int P2 = 2 * P;
    constexpr bool JAC = (MODE & MODE_JAC) != 0;
    constexpr int VS = 16 / sizeof(TO);
    constexpr int LINE = 128 / sizeof(TO);           // scalars per 128-byte line
    constexpr int WAVE_LDS = HALF * P2 + LINE + 64;  // packed range + alignment shift + one dummy slot per lane
    using O2 = typename Vec2<TO>::type;
    using V16 = __attribute__((ext_vector_type(VS))) TO;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    TO *tr = reinterpret_cast<TO *>(smem_raw) + wave * ((WAVE_LDS + VS - 1) / VS * VS);
    TO *dummy = tr + HALF * P2 + LINE + lane;
    const T *cam_slab = static_cast<const T *>(a.cam_slab);
    const T *pose_slab = static_cast<const T *>(a.pose_slab);
    const T *points = static_cast<const T *>(a.points);
    TO *resid = static_cast<TO *>(a.resid);
    TO *data = static_cast<TO *>(a.jac);
    const int64_t tile0 = (int64_t)blockIdx.x * a.tiles_per_wg;
    const int64_t tile1 = min(tile0 + (int64_t)a.tiles_per_wg, a.n_tiles);
    for (int64_t tile = tile0 + wave; tile < tile1; tile += WAVES_PER_WG) {
        const int64_t i = tile * TILE + lane;
        const bool valid = i < a.n;
        const int64_t ic = valid ? i : a.n - 1;
        int c, im, k;
        load_indices(a.tab, ic, c, im, k);
        const double2v m = load_uv(a.tab, ic);
        // the mask and the CSR offset are streamed like the indices: requested HERE, with them.  Left where they are used,
        // hipcc issues the two loads after the evaluation and the wave sits out a full memory latency per tile.
        uint32_t keep_raw = 0;
        int64_t off_raw = 0;
        if constexpr (JAC) {
            keep_raw = a.keep[ic];
            off_raw = a.row_off[ic];
        }
        asm volatile("" ::: "memory");
        T u, v;
        T J[P2];
        {
            const T X0 = points[3 * k], X1 = points[3 * k + 1], X2 = points[3 * k + 2];
            const int c0 = __builtin_amdgcn_readfirstlane(c), im0 = __builtin_amdgcn_readfirstlane(im);
            if (__all(c == c0 && im == im0))   // one camera and one image in the tile: slabs through scalar loads (ba_device.hpp)
                eval_detection<CHAIN, T, JAC>(ScalarSlab(cam_slab + c0 * CAM_STRIDE), ScalarSlab(pose_slab + im0 * POSE_STRIDE), X0, X1, X2, u, v, J);
            else
                eval_detection<CHAIN, T, JAC>(cam_slab + c * CAM_STRIDE, pose_slab + im * POSE_STRIDE, X0, X1, X2, u, v, J);
        }
        if constexpr ((MODE & MODE_RESID) != 0) {  // branch-free (see ba_eval_kernel)
            O2 r;
            r.x = (TO)(u - m.x);
            r.y = (TO)(v - m.y);
            __builtin_nontemporal_store(r, valid ? reinterpret_cast<O2 *>(resid) + i : static_cast<O2 *>(a.sink));
        }
        if constexpr (JAC) {
            const uint32_t keep = valid ? keep_raw : 0u;
            const int cnt = __popc(keep);
            const int64_t off = off_raw + (valid ? 0 : 2 * (int64_t)__popc(keep_raw));  // tail lanes: end of data
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
                    const bool on = mine && ((keep >> j) & 1u);
                    const int pos = __popc(keep & ((1u << j) - 1u));
                    *(on ? ru + pos : dummy) = (TO)J[j];
                    *(on ? rv + pos : dummy) = (TO)J[P + j];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                TO *gal = g0 - mis;                                                    // 128-byte aligned
                const int n_units = (mis + len + VS - 1) / VS;
                for (int q = lane; q < n_units; q += 64) {
                    const int e0 = q * VS;
                    if (e0 >= mis && e0 + VS <= mis + len) {
                        __builtin_nontemporal_store(reinterpret_cast<const V16 *>(tr)[q], reinterpret_cast<V16 *>(gal + e0));
                    } else {
#pragma unroll
                        for (int t = 0; t < VS; ++t)
                            if (e0 + t >= mis && e0 + t < mis + len) gal[e0 + t] = tr[e0 + t];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
}

// Legacy residual-only cost (compiled_helpers.py:518-549, used by the initial pose
// selection template_handler.py:535-592): pre-multiplied 3x4 projection matrices and pre-transformed
// points im_points[image, key].  cam_tab row (24 scalars): P row-major 12 | fx cx fy cy | k0 k1 p0 p1 k2 | pad.
// 44 B of traffic per detection: as in the residual-only mode of ba_eval_kernel the per-lane loads of the camera
// table set the pace, so a tile that shares its camera fetches the 21 scalars with scalar loads (ScalarSlab).
constexpr int LEGACY_STRIDE = 24;
__global__ __launch_bounds__(256) void legacy_cost_kernel(const DetTable tab, const T *__restrict__ im_points, const T *__restrict__ cam_tab,
                                                          T *__restrict__ errors, int64_t n, int64_t n_keys, void *sink) {
    const int lane = threadIdx.x & 63;
    const int64_t n_tiles = (n + 63) / 64;
    const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t tile = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; tile < n_tiles; tile += waves) {
        const int64_t i = tile * 64 + lane;
        const bool valid = i < n;
        const int64_t ic = valid ? i : n - 1;
        int c, im, k;
        load_indices(tab, ic, c, im, k);
        const T *X = im_points + 3 * ((int64_t)im * n_keys + k);
        const T X0 = X[0], X1 = X[1], X2 = X[2];
        const double2v m = load_uv(tab, ic);
        T ct[21];
        const int c0 = __builtin_amdgcn_readfirstlane(c);
        if (__all(c == c0)) {
            const ScalarSlab cs(cam_tab + (int64_t)c0 * LEGACY_STRIDE);   // scalar loads (ba_device.hpp)
#pragma unroll
            for (int j = 0; j < 21; ++j) ct[j] = cs[j];
        } else {
#pragma unroll
            for (int j = 0; j < 21; ++j) ct[j] = cam_tab[(int64_t)c * LEGACY_STRIDE + j];
        }
        T p0 = ct[0] * X0 + ct[1] * X1 + ct[2] * X2 + ct[3];        // ch:538  P [X;1]
        T p1 = ct[4] * X0 + ct[5] * X1 + ct[6] * X2 + ct[7];
        const T p2 = ct[8] * X0 + ct[9] * X1 + ct[10] * X2 + ct[11];
        p0 = p0 / p2;                                               // ch:539
        p1 = p1 / p2;
        const T fx = ct[12], cx = ct[13], fy = ct[14], cy = ct[15];
        const T k0 = ct[16], k1 = ct[17], q0 = ct[18], q1 = ct[19], k2 = ct[20];
        const T x = (p0 - cx) / fx, y = (p1 - cy) / fy;             // ch:455
        const T r2 = x * x + y * y;
        const T kup = T(1) + k0 * r2 + k1 * (r2 * r2) + k2 * (r2 * r2 * r2);
        const T xD = x * kup + T(2) * q0 * x * y + q1 * (r2 + T(2) * x * x);
        const T yD = y * kup + q0 * (r2 + T(2) * y * y) + T(2) * q1 * x * y;
        double2v e;
        e.x = (xD * fx + cx) - m.x;                                  // ch:541-542
        e.y = (yD * fy + cy) - m.y;
        __builtin_nontemporal_store(e, valid ? reinterpret_cast<double2v *>(errors) + i : static_cast<double2v *>(sink));
    }
}

// Streaming probes used to measure the box's achievable HBM rate for THIS access shape
// (16 B per lane, 1 KiB per wave-instruction): kind 0 plain fill, 1 non-temporal fill, 2 plain copy,
// 3 non-temporal copy.  Reported next to the 8 TB/s spec figure in DESIGN.md.
// kind 4: non-temporal fill in the fused kernel's shape — every wave owns whole 10 752-byte chunks (a half tile
// of the template chain: 672 units of 16 B, 10.5 store instructions) far apart from the other waves' chunks.
template <int KIND>
__global__ __launch_bounds__(256) void membench_kernel(const double2 *__restrict__ src, double2 *__restrict__ dst, int64_t n16) {
    using V = __attribute__((ext_vector_type(2))) double;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    V *d = reinterpret_cast<V *>(dst);
    const V *s = reinterpret_cast<const V *>(src);
    if constexpr (KIND >= 4) {
        // chunk sizes (16-byte units) of kinds 4..8: the half tile, the whole tile, 1 KiB, 64 KiB, 256 KiB
        constexpr int CHUNK = KIND == 4 ? 672 : KIND == 5 ? 1344 : KIND == 6 ? 64 : KIND == 7 ? 4096 : 16384;
        const int lane = threadIdx.x & 63;
        const int64_t n_chunks = n16 / CHUNK, waves = stride / 64;
        for (int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / 64; c < n_chunks; c += waves) {
            V *o = d + c * CHUNK;
#pragma unroll 4
            for (int q = lane; q < CHUNK; q += 64) {
                V v;
                v.x = (double)c;
                v.y = (double)q;
                __builtin_nontemporal_store(v, o + q);
            }
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        V v;
        if constexpr (KIND >= 2) v = (KIND == 3) ? __builtin_nontemporal_load(s + i) : s[i];
        else { v.x = (double)i; v.y = 1.0; }
        if constexpr (KIND == 1 || KIND == 3) __builtin_nontemporal_store(v, d + i); else d[i] = v;
    }
}

}  // namespace pcs
