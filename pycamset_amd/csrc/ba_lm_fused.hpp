// ba_lm_fused.hpp — the small kernels of an LM trial, merged (round 5).
//
// profiles/r04/lm_trace_rig32.log: of the 273 us of a rig-32 trial, 46 us were six launches of a few microseconds of work each —
// a launch ramp (~3-5 us of dispatch, first loads, tail) around it.  Two fusions take four of them away:
//
//   schur_prep_kernel    = schur_trail_lead_kernel + schur_v_kernel.  V = B L^-T needs the factor of the trailing entity, which a
//                          launch boundary used to deliver.  Here a workgroup owns EPB entities x 64 leading rows of V: its first EPB
//                          lanes factor "their" entities into LDS (a 6 x 6 or 3 x 3 Cholesky in registers: ~2 us of latency, done
//                          redundantly by every row block — in parallel, so it costs nothing but the 2 us), the workgroup then
//                          forms its piece of V.  The row block 0 of an entity chunk publishes the entity-level outputs (L^-T, u,
//                          D, masked g).  The tiles of S = sym(A) + lambda D ride along as before.
//   schur_finish_kernel  = schur_vtx_kernel + schur_back_kernel + normal_prologue_kernel.  A workgroup reduces w = V' x_l for the
//                          columns of a few trailing entities, back-substitutes them (x_e = -L^-T (u_e + w_e)), writes their step
//                          and trial parameters — and prepares what the NEXT kernel (the normal equations at the trial string) needs
//                          from them: the Rodrigues slab of a pose (template chain), the point copy (self / free chain).  Leading
//                          parameters (cameras; poses of the self chain) get their step, trial values and slabs from x_l directly,
//                          in other workgroups; the rest of the grid zeroes the trial state's blocks.  Nothing here waits for
//                          anything inside the launch.
// Same arithmetic, statement for statement, as the kernels they replace (rot_terms / rot_element of ba_device.hpp for the slabs), so a
// trial computes the same bits either way (engine option "fused_trial" = 0 keeps the separate launches for A/B).
#pragma once
#include <hip/hip_runtime.h>

#include "ba_device.hpp"
#include "ba_kernels.hpp"
#include "ba_schur.hpp"

namespace pcs {

constexpr int PREP_RPB = 64;                                               // leading rows per workgroup of the V part
__host__ __device__ constexpr int prep_epb(const int tb) { return tb == 6 ? 16 : 32; }   // entities per workgroup: 96 doubles of a row of B

// grid: [trail_blocks = ent_chunks x row_chunks workgroups for V | tiles of S]; SchurArgs::ent_chunks says how the first part splits
template <int TB>
__global__ __launch_bounds__(256) void schur_prep_kernel(const SchurArgs a0) {
    PCS_STOP_GUARD(a0);
    const SchurArgs a = schur_current(a0);
    {
        const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (int64_t i = t; i < a.fill_n; i += (int64_t)gridDim.x * blockDim.x) a.fill[i] = ~0ull;
    }
    if ((int)blockIdx.x >= a.trail_blocks) {
        schur_lead_body(a, (int)blockIdx.x - a.trail_blocks);
        return;
    }
    constexpr int EPB = prep_epb(TB), LT = TB * TB + 1;                    // odd stride: the lanes of a row read different banks
    __shared__ double Lt[EPB * LT];
    const int tid = threadIdx.x;
    const int ec = (int)blockIdx.x % a.ent_chunks, rc = (int)blockIdx.x / a.ent_chunks;
    const int64_t e0 = (int64_t)ec * EPB;
    const bool publish = rc == 0;
    // this thread's (row, entity) pieces of B, requested BEFORE the factorisation: their latency (and that of the fixed-parameter bytes)
    // passes while the first EPB lanes factor
    constexpr int NQ = PREP_RPB * EPB / 256;
    double bv[NQ][TB];
    int fx[NQ];   // bit j: entry j is masked (its row or its column is a fixed parameter); bit 30: the piece exists
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        const int q = tid + 256 * qi, el = q % EPB;
        const int64_t r = (int64_t)rc * PREP_RPB + q / EPB, e = e0 + el;
        const bool have = r < a.n_lead && e < a.n_ent;
        const double *Bp = a.B + (have ? r * a.n_trail + e * TB : 0);
        const bool row_fixed = have && a.fixed[r] != 0;
        int f = have ? (1 << 30) : 0;
#pragma unroll
        for (int jj = 0; jj < TB; ++jj) {
            bv[qi][jj] = have ? Bp[jj] : 0.0;
            if (have && (row_fixed || a.fixed[a.trail_off + e * TB + jj])) f |= 1 << jj;
        }
        fx[qi] = f;
    }
    if (tid < EPB && e0 + tid < a.n_ent) schur_trail_entity<TB>(a, e0 + tid, Lt + tid * LT, publish);
    __syncthreads();
    if (publish) {
        for (int q = tid; q < EPB * TB * TB; q += 256) {
            const int el = q / (TB * TB), k = q % (TB * TB);
            if (e0 + el < a.n_ent) a.linvt[(e0 + el) * TB * TB + k] = Lt[el * LT + k];
        }
    }
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        if (!(fx[qi] & (1 << 30))) continue;
        const int q = tid + 256 * qi, el = q % EPB;
        const int64_t r = (int64_t)rc * PREP_RPB + q / EPB, e = e0 + el;
        double b[TB];
        bool touched = false;
#pragma unroll
        for (int jj = 0; jj < TB; ++jj) {
            b[jj] = bv[qi][jj];
            if ((fx[qi] >> jj & 1) && b[jj] != 0.0) { b[jj] = 0.0; touched = true; }
        }
        if (touched) {   // B is masked in place (rows / columns of fixed parameters -> 0), as schur_v_kernel does
            double *Bp = a.B + r * a.n_trail + e * TB;
#pragma unroll
            for (int jj = 0; jj < TB; ++jj) Bp[jj] = b[jj];
        }
        const double *L = Lt + el * LT;
        double *Vp = a.V + r * a.n_trail + e * TB;
#pragma unroll
        for (int jj = 0; jj < TB; ++jj) {
            double sum = 0.0;
#pragma unroll
            for (int ii = 0; ii <= jj; ++ii) sum += b[ii] * L[ii * TB + jj];
            Vp[jj] = sum;
        }
    }
}

struct SchurFinishArgs {
    // w = V' x_l
    const double *V, *xl;
    int32_t n_lead, n_trail, ldv;
    // x_e = -L^-T (u_e + w_e), the step and the trial string
    const double *linvt, *u;
    const uint8_t *fixed;
    double *delta;
    const double *ps_in;      // the two parameter strings: state 0 / state 1 (ba_schur.hpp SchurBackArgs)
    double *ps_out;
    int64_t n_ent, trail_off;
    const int32_t *stop, *sel;
    double *vote;
    int64_t vote_alt;
    const int32_t *status;
    // what the normal equations at the trial string need: slabs (and the point copy) from the trial parameters, the trial state's
    // blocks zeroed (normal_prologue_kernel)
    double *cam_slab, *pose_slab, *points;
    int32_t n_cams, n_imgs, n_keys, has_pose, copy_points;
    int64_t extr_off, pose_off, point_off;
    double *Hm;               // [A | B | C] of the trial state while state 0 is current; alt_out doubles further on otherwise
    int64_t n_h;
    double *g;
    int64_t n_g;
    double *cost;
    int64_t alt_out;
    int32_t w_blocks, lead_blocks;   // roles by block index: [w_blocks | lead_blocks | the rest zero]
};
__host__ __device__ constexpr int finish_ecb(const int tb) { return tb == 6 ? 2 : 5; }   // entities per workgroup of the w part: 12 / 15 columns of V

template <int TB>
__global__ __launch_bounds__(1024) void schur_finish_kernel(const SchurFinishArgs a0) {
    PCS_STOP_GUARD(a0);
    SchurFinishArgs a = a0;
    const bool flipped = a.sel && *a.sel;
    if (flipped) { a.ps_in = a0.ps_out; a.ps_out = const_cast<double *>(a0.ps_in); a.Hm += a.alt_out; a.g += a.alt_out; a.cost += a.alt_out; }
    const int tid = threadIdx.x;
    constexpr int ECB = finish_ecb(TB), COLS = ECB * TB;
    static_assert(COLS <= 16, "16 column lanes");
    if ((int)blockIdx.x < a.w_blocks) {
        // ---- w for COLS columns: 64 parts x 16 column lanes, two independent chains per thread (schur_vtx_kernel<16>) ------------------------
        __shared__ double red[64][17];
        __shared__ double pn[ECB][TB];                                      // the entities' trial parameters (for the pose slabs)
        const int col = tid & 15, part = tid >> 4;
        const int64_t e0 = (int64_t)blockIdx.x * ECB;
        const int64_t j = e0 * TB + col;
        double s0 = 0.0, s1 = 0.0;
        if (col < COLS && j < a.n_trail) {
            int r = part;
            for (; r + 64 < a.n_lead; r += 128) {
                s0 += a.V[(int64_t)r * a.ldv + j] * a.xl[r];
                s1 += a.V[(int64_t)(r + 64) * a.ldv + j] * a.xl[r + 64];
            }
            if (r < a.n_lead) s0 += a.V[(int64_t)r * a.ldv + j] * a.xl[r];
        }
        red[part][col] = s0 + s1;
        __syncthreads();
        for (int half = 32; half >= 16; half >>= 1) {
            if (part < half) red[part][col] += red[part + half][col];
            __syncthreads();
        }
        if (part == 0) {
            double s = 0.0;
#pragma unroll
            for (int p = 0; p < 16; ++p) s += red[p][col];
            red[0][col] = s;                                                // (row 0 is read by this thread only in the loop above)
        }
        __syncthreads();
        // ---- back substitution: one lane per entity (schur_back_kernel) ------------------------------------------------------------
        if (tid < ECB && e0 + tid < a.n_ent) {
            const int64_t e = e0 + tid;
            const double *Lt = a.linvt + e * TB * TB;
            double sv[TB];
#pragma unroll
            for (int i = 0; i < TB; ++i) sv[i] = a.u[e * TB + i] + red[0][tid * TB + i];
#pragma unroll
            for (int i = 0; i < TB; ++i) {   // x = -L^-T s: L^-T is upper triangular
                double x = 0.0;
#pragma unroll
                for (int jj = i; jj < TB; ++jj) x += Lt[i * TB + jj] * sv[jj];
                const int64_t c = a.trail_off + e * TB + i;
                const double d = a.fixed[c] ? 0.0 : -x;
                const double pv = a.ps_in[c] + d;
                a.delta[c] = d;
                a.ps_out[c] = pv;
                pn[tid][i] = pv;
                if (TB == 3 && a.copy_points) a.points[3 * e + i] = pv;     // self / free chain: the trailing entities are the points
            }
        }
        if (TB == 6 && a.pose_slab) {                                        // template chain: the trailing entities are the poses — their slabs (a generated chain: none)
            __syncthreads();
            if (tid < ECB * POSE_STRIDE) {
                const int el = tid / POSE_STRIDE, slot = tid % POSE_STRIDE;
                const int64_t im = e0 + el;
                if (im < a.n_ent) {
                    const double *p6 = pn[el];
                    T v;
                    if (slot >= POSE_T && slot < POSE_DR) v = p6[3 + slot - POSE_T];
                    else if (slot == POSE_STRIDE - 1) v = T(0);
                    else v = rot_element(rot_terms(p6[0], p6[1], p6[2]), slot < POSE_T ? slot - POSE_R : 9 + slot - POSE_DR);
                    a.pose_slab[im * POSE_STRIDE + slot] = v;
                }
            }
        }
        return;
    }
    if ((int)blockIdx.x < a.w_blocks + a.lead_blocks) {
        // ---- leading parameters: step, trial values, and the slabs that depend on them alone (cameras; poses of the self chain) ---------
        const int64_t t = (int64_t)((int)blockIdx.x - a.w_blocks) * 1024 + tid;
        if (t == 0 && a.vote) a.vote[flipped ? a.vote_alt : 0] = (a.status && (*a.status & 4)) ? 1.0 : 0.0;
        auto psn = [&](const int64_t i) -> double { return a.ps_in[i] + (a.fixed[i] ? 0.0 : a.xl[i]); };   // i < n_lead
        if (t < a.n_lead) {
            const double d = a.fixed[t] ? 0.0 : a.xl[t];
            a.delta[t] = d;
            a.ps_out[t] = a.ps_in[t] + d;
        }
        const int64_t n_cam_el = (int64_t)a.n_cams * CAM_STRIDE;
        const bool lead_poses = a.has_pose && a.pose_off < a.trail_off;      // self chain
        if (t < n_cam_el) {
            const int64_t c = t / CAM_STRIDE;
            const int slot = (int)(t - c * CAM_STRIDE);
            const int64_t p6 = a.extr_off + 6 * c;
            T v;
            if (slot < CAM_R) v = principal_or_nan(psn(9 * c + slot), slot, psn(9 * c), psn(9 * c + 2));
            else if (slot >= CAM_T && slot < CAM_DR) v = psn(p6 + 3 + slot - CAM_T);
            else v = rot_element(rot_terms(psn(p6), psn(p6 + 1), psn(p6 + 2)), slot < CAM_T ? slot - CAM_R : 9 + slot - CAM_DR);
            a.cam_slab[t] = v;
        } else if (lead_poses && t < n_cam_el + (int64_t)a.n_imgs * POSE_STRIDE) {
            const int64_t u = t - n_cam_el;
            const int64_t im = u / POSE_STRIDE;
            const int slot = (int)(u - im * POSE_STRIDE);
            const int64_t p6 = a.pose_off + 6 * im;
            T v;
            if (slot >= POSE_T && slot < POSE_DR) v = psn(p6 + 3 + slot - POSE_T);
            else if (slot == POSE_STRIDE - 1) v = T(0);
            else v = rot_element(rot_terms(psn(p6), psn(p6 + 1), psn(p6 + 2)), slot < POSE_T ? slot - POSE_R : 9 + slot - POSE_DR);
            a.pose_slab[u] = v;
        }
        return;
    }
    // ---- the trial state's blocks, gradient and cost: zero (normal_prologue_kernel) ----------------------------------------------------------
    using D2 = typename Vec2<double>::type;
    const int first = a.w_blocks + a.lead_blocks;
    const int64_t t = (int64_t)((int)blockIdx.x - first) * 1024 + tid;
    const int64_t nt = (int64_t)((int)gridDim.x - first) * 1024;
    D2 *h2 = reinterpret_cast<D2 *>(a.Hm);
    for (int64_t i = t; i < a.n_h / 2; i += nt) __builtin_nontemporal_store(D2{0.0, 0.0}, h2 + i);
    if (t == 0 && (a.n_h & 1)) a.Hm[a.n_h - 1] = 0.0;
    for (int64_t i = t; i < a.n_g; i += nt) a.g[i] = 0.0;
    if (t == 0) *a.cost = 0.0;
}

}  // namespace pcs
