/*
 * pcs_hip.h — C ABI of the MI355X bundle-adjustment cost/Jacobian engine (libpcs_hip.so).
 *
 * Drop-in boundary for pyCamSet's optimisation hot path.  Every entry point names the
 * reference interface it replaces (paths relative to the reference repo, pyCamSet/optimisation/
 * unless stated: afb = abstract_function_blocks.py, th = template_handler.py,
 * sbh = standard_bundle_handler.py, fph = free_point_handler.py,
 * td = ../calibration_targets/target_detections.py).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types cross the boundary.
 *   - every function returns PCS_OK (0) or a negative pcs_status; pcs_last_error() gives text
 *     for the calling thread's most recent failure.  No exception crosses the ABI.
 *   - the caller owns every buffer it passes; the engine owns its device buffers for the
 *     lifetime of the handle.  One handle = one host thread at a time.
 *   - streams: the *_device entry points take a stream; work a handle queues on different streams is ordered
 *     by the engine (an event recorded after every enqueue is waited for before the shared slabs, the staging
 *     buffer or the masks are touched again), so consecutive calls may use different streams.  What the engine
 *     cannot order is the CALLER's use of the output buffers: synchronise the stream a call was queued on (or
 *     pcs_synchronize) before reading them from another stream or from the host.
 *   - numerical inf/nan (e.g. a point on the camera plane, z = 0) pass through unchanged,
 *     as in the reference's numba code.
 *   - there is NO CPU fallback: without a HIP device pcs_create fails with PCS_ERR_NODEVICE.
 */
#ifndef PCS_HIP_H
#define PCS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcs_engine pcs_engine;

typedef enum {
    PCS_OK = 0,
    PCS_ERR_ARG = -1,      /* null pointer, bad enum, bad size */
    PCS_ERR_HIP = -2,      /* a HIP runtime call failed (text in pcs_last_error) */
    PCS_ERR_STATE = -3,    /* detections / template / structure not set yet */
    PCS_ERR_NODEVICE = -4, /* no usable HIP device */
    PCS_ERR_RANGE = -5     /* a detection indexes outside [0,n_cams) x [0,n_imgs) x [0,n_keys) */
} pcs_status;

/* Function-block chains (the three `op_fun` sums the reference handlers build). */
typedef enum {
    PCS_CHAIN_TEMPLATE = 0, /* projection + extrinsic3D + template_points          th:152   P = 21 */
    PCS_CHAIN_SELF = 1,     /* projection + extrinsic3D + rigidTform3d + free_point sbh:182  P = 24 */
    PCS_CHAIN_FREE = 2      /* projection + extrinsic3D + free_point                fph:143  P = 18 */
} pcs_chain;

/* Arithmetic and parameter slabs are FP64 for every dtype (the reference's precision, fbi:11); the dtype selects the
 * BYTES of the streams:  PCS_F64   measurements double, residual / Jacobian double   (380 B per detection, chain T)
 *                        PCS_F32   measurements float,  residual / Jacobian float    (196 B) — BASELINE config 5
 *                        PCS_MIXED measurements double, residual / Jacobian float    (204 B)
 * A float output is the FP64 result rounded once at the store (<= 1.2e-7 relative); PCS_F32 additionally rounds the
 * measured (u, v) to float on upload (<= 6e-5 px at ~1e3 px).  Round 1's all-float arithmetic (5e-3 relative error
 * from cancellation in the chain rule) is gone.  Device outputs of the *_device entry points are float for PCS_F32
 * and PCS_MIXED. */
typedef enum { PCS_F64 = 0, PCS_F32 = 1, PCS_MIXED = 2 } pcs_dtype;

/* Library / build identification. */
int pcs_version(void);
const char *pcs_last_error(void);
/* Number of HIP devices visible (0 when none); never fails. */
int pcs_device_count(void);

/*
 * Create an engine for one chain on one device.
 * Replaces: optimisation_function.__init__/_prep_for_computation (afb:111-190) +
 *           make_param_struct (afb:777-820).  The reference derives the group counts from the
 *           detection table (max index + 1, afb:793-795); here they are explicit arguments.
 * Parameter-string layout (same as the reference):
 *   intr 9*c+j | extr 9*C + 6*c+j | pose 15*C + 6*i+j (chains T,S) | point (15*C+6*I | 15*C) + 3*k+j (S | F)
 */
int pcs_create(pcs_engine **out, int chain, int dtype, int64_t n_cams, int64_t n_imgs, int64_t n_keys, int device);
int pcs_destroy(pcs_engine *h);

int64_t pcs_n_params(const pcs_engine *h);     /* length of the full parameter string */
int pcs_row_len(const pcs_engine *h);          /* P: dense Jacobian entries per row */
int64_t pcs_n_detections(const pcs_engine *h); /* N */

/*
 * Upload the static detection table.
 * Replaces: the closure state captured by make_full_loss_template / make_full_jac_template
 *           (_reshape_data_for_parallel afb:281-288, get_block_param_inds afb:192-233).
 * pcs_set_detections_table takes the reference's own (N,5) float64 table
 * [cam, im, key, u, v] (td:51-55 after return_flattened_keys td:333-351); indices are cast with
 * (int) like afb:214 / afb:375.  pcs_set_detections takes the split form.
 * Indices are range-checked against the counts given to pcs_create -> PCS_ERR_RANGE.
 */
int pcs_set_detections_table(pcs_engine *h, const double *det5, int64_t n);
int pcs_set_detections(pcs_engine *h, const int32_t *cam, const int32_t *img, const int32_t *key,
                       const double *uv /* n x 2 */, int64_t n);

/*
 * Upload the constant template points (chain TEMPLATE only), (n_keys, 3) float64.
 * Replaces: the `template` argument of the generated loss/jac (afb:352-354, afb:374-375;
 *           th:160, th:174 `target.point_data.reshape((-1, 3))`).
 */
int pcs_set_template(pcs_engine *h, const double *points);

/*
 * One evaluation at a full parameter string (host buffers, synchronous).
 * Replaces: generated full_loss (afb:350-387) and full_jac (afb:552-599) + the [:n_elements]
 *           truncation (afb:641).
 *   resid : (N,2) row-major  = projected - measured            (afb:384) or NULL
 *   jac   : (2N,P) row-major dense block rows, u row then v row (afb:591-594) or NULL
 * Both are float64 on the host whatever the engine dtype (an F32 engine computes and stores
 * float32 on the device and widens on the way out).
 */
int pcs_eval(pcs_engine *h, const double *param_str, double *resid, double *jac);

/*
 * Same evaluation, asynchronous, outputs left in device memory (engine dtype), launched on
 * `stream` (a hipStream_t passed as void*, NULL = the engine's own non-blocking stream; to queue on the
 * process's default (NULL) stream pass hipStreamLegacy, i.e. (hipStream_t)1 — work on the engine stream is
 * NOT ordered against the default stream).
 *   d_resid : device pointer, N*2 elements, or NULL
 *   d_jac   : device pointer, 2N*P elements, or NULL
 * param_str is a host pointer (copied through a pinned staging buffer); use
 * pcs_eval_device_resident when the parameter string already lives on the device.
 */
int pcs_eval_device(pcs_engine *h, const double *param_str, void *d_resid, void *d_jac, void *stream);
int pcs_eval_device_resident(pcs_engine *h, const double *d_param_str, void *d_resid, void *d_jac, void *stream);

/*
 * Static CSR structure for a given fixed-parameter mask.
 * Replaces: make_jac_CSR_columns_row_pointers (afb:465-489).
 *   unfixed : n_params bytes (non-zero = free), NULL = all free
 *   indices : nnz int64 (pass NULL to only query nnz), indptr : 2N+1 int64 (may be NULL)
 */
int pcs_csr_structure(pcs_engine *h, const uint8_t *unfixed, int64_t *indices, int64_t *indptr, int64_t *nnz);
/* Per-detection global column table (N,P) int64.  Replaces: get_block_param_inds(unthreaded) afb:192-233. */
int pcs_block_param_inds(pcs_engine *h, int64_t *out);

/*
 * Fixed-parameter compaction on the device.
 * Replaces: `data[:n_elements][good_mask]` (afb:627-651) — the reference's per-call host
 * boolean-mask copy.  pcs_set_unfixed prepares the static per-row offsets; pcs_eval_compact
 * writes only the unfixed entries, in CSR data order.
 */
int pcs_set_unfixed(pcs_engine *h, const uint8_t *unfixed, int64_t *nnz);
int pcs_eval_compact(pcs_engine *h, const double *param_str, double *resid, double *data /* nnz */);
int pcs_eval_compact_device(pcs_engine *h, const double *param_str, void *d_resid, void *d_data, void *stream);

/*
 * Matrix-free products with the Jacobian at a linearisation point: J is never
 * materialised.  Replaces what scipy does with the reference's CSR Jacobian
 * (optimisation_handling.py:88-98: x_scale='jac' column norms, J^T f, lsmr mat-vecs).
 * pcs_linearize prepares the slabs at `param_str` (so does a pcs_eval* call that launched slab_prep — not a one-launch step,
 * option "fuse_prep": pcs_matfree then returns PCS_ERR_STATE until pcs_linearize has run).
 * pcs_matfree ops (vectors on the host, FULL parameter-string space, float64):
 *   0 JV    in: n_params            out: 2N        out = J in
 *   1 JTU   in: 2N                  out: n_params  out = J^T in
 *   2 JTJV  in: n_params            out: n_params  out = J^T (J in)
 *   3 DIAG  in: NULL                out: n_params  out = diag(J^T J)
 *   4 GRAD  in: NULL                out: n_params  out = J^T r,  *cost = sum r^2 (cost may be NULL)
 * Sums use f64 atomics: their last bits depend on arrival order.
 */
int pcs_linearize(pcs_engine *h, const double *param_str);
int pcs_matfree(pcs_engine *h, int op, const double *in, double *out, double *cost);

/* Block-reduced normal equations at param_str (what a Levenberg-Marquardt step needs
 * from the Jacobian that optimisation_handling.py:88-98 hands to scipy), built in one pass without
 * writing J:  H = J^T J  (n_params x n_params row-major, FULL parameter-string space, only the UPPER
 * triangle incl. the diagonal is written, the rest is zero),  g = J^T r  (n_params),  cost = r^T r.
 * float64 whatever the engine dtype.  The sums use f64 atomics: the last bits depend on arrival order.  The kernels address H
 * with 32-bit offsets in doubles: n_params <= PCS_NORMAL_MAX_PARAMS = 65535 (a 34 GB matrix; 23170 until round 4); both entry points return
 * PCS_ERR_ARG beyond it, before anything is allocated.
 *   pcs_normal_equations         host buffers (engine-owned device scratch, blocking)
 *   pcs_normal_equations_device  device buffers of the caller, queued on `stream` (NULL = engine stream);
 *                                the call zeroes them first.  Observation shards: all-reduce H, g, cost. */
#define PCS_NORMAL_MAX_PARAMS 65535
int pcs_normal_equations(pcs_engine *h, const double *param_str, double *H, double *g, double *cost);
int pcs_normal_equations_device(pcs_engine *h, const double *param_str, double *d_H, double *d_g, double *d_cost, void *stream);

/* The same normal equations in BLOCKED form, for a solver that stays on the device (round 3): the parameter string splits
 * into a leading part (cameras; + poses for the self chain) and the trailing group whose entities never share a detection
 * (poses of the template chain — 6 columns each —, points of the self / free chains — 3 each), and only what is structurally
 * non-zero is stored:
 *     d_packed = [ A (n_lead x n_lead, upper triangle) | B (n_lead x n_trail) | C (n_trail / tb blocks of tb x tb, upper
 *                  triangle) | g (n_params) | cost (1) ]     float64, 16-byte aligned, zeroed by the call.
 * pcs_normal_layout: out5 = {n_lead, n_trail, tb, length of d_packed in doubles, n_params}.  The parameter string is read
 * from DEVICE memory (d_param_str), so an LM loop never stages it through the host.  Limits: each region below 2^32 doubles (32 GiB).
 * Observation shards: all-reduce d_packed.  Consumer: pycamset_amd/device_solver.py (reference consumer of J:
 * optimisation_handling.py:88-98). */
int pcs_normal_layout(const pcs_engine *h, int64_t *out5);
int pcs_normal_blocks_device(pcs_engine *h, const double *d_param_str, double *d_packed, void *stream);
/* The block-shaped parts of one damped step (H + lambda diag(H)) x = -g on the packed form (csrc/ba_schur.hpp; all pointers
 * are device memory, lambda included):
 *   pcs_schur_prepare  per trailing entity C_e + lambda D_e = L L' -> d_linvt (n_ent x tb x tb: L^-T), d_u (n_trail: L^-1 g_e);
 *                      d_V (n_lead x n_trail) = B L^-T;  d_S (n_lead x n_lead) = sym(A) + lambda D;  d_rhs = -g_lead;
 *                      d_dvec (n_params) = D;  d_gm (n_params) = g.  Parameters with d_fixed[i] != 0 (n_params bytes) get identity
 *                      rows / columns and a zero gradient (B is masked in place).  *d_status |= 1 when a trailing block is
 *                      not positive definite.  The caller then forms S -= V V', rhs += V u, solves S x_l = rhs (library
 *                      GEMM / Cholesky) and w = V' x_l, and
 *   pcs_schur_finish   writes the step in parameter-string order: d_delta (n_params) = [x_l | -L^-T (u + w)], 0 where fixed; with
 *                      d_ps_in / d_ps_out (both or neither) also the trial parameter string d_ps_out = d_ps_in + d_delta.
 *   pcs_lm_decide      the accept / reject decision of the trial on the device: predicted reduction 0.5 (lambda d'D d - g'd), actual
 *                      reduction 0.5 (cost_old - cost_new), gain ratio, *d_lambda <- the next damping (x 1/3 | 1 | 2 by the ratio, x 4
 *                      on a rejected or failed step), *d_status <- 0, and d_stats[12] = {accepted (-1: bit 2 of *d_status was set — the dense solve
 *                      did not complete, the trial is void), max |g|, relative cost drop, |step|,
 *                      |x| over the free parameters, new sum r^2, old sum r^2, lambda used, 0, 0, 0, the next lambda} — the one vector the host reads per trial
 *                      ([8] .. [10] are the stop code, the trial number and the current state of the device-steered loop, pcs_lm_trial). */
int pcs_schur_prepare(pcs_engine *h, double *d_packed, const uint8_t *d_fixed, const double *d_lambda, double *d_linvt, double *d_u,
                      double *d_V, double *d_S, double *d_rhs, double *d_dvec, double *d_gm, int32_t *d_status, void *stream);
int pcs_schur_finish(pcs_engine *h, const double *d_linvt, const double *d_u, const double *d_w, const double *d_xlead, const uint8_t *d_fixed,
                     double *d_delta, const double *d_ps_in, double *d_ps_out, void *stream);
int pcs_lm_decide(pcs_engine *h, const double *d_cost_old, const double *d_cost_new, const double *d_dvec, const double *d_gm, const double *d_delta,
                  const double *d_ps, const uint8_t *d_fixed, int32_t *d_status, double *d_lambda, double *d_stats, void *stream);

/* One whole LM trial per call — the device-steered form of the loop optimisation_handling.py:88-98 leaves to scipy.
 * The loop keeps TWO states (packed normal equations + parameter string each); flags[2] names the current one, the other receives the
 * trial.  pcs_lm_trial_build queues, on `stream`: the damped Schur step from the current state at *lambda (pcs_schur_prepare,
 * pcs_schur_syrk, pcs_dense_spd_solve_algo, pcs_schur_vtx, pcs_schur_finish: delta and the trial string = current string + delta) and
 * the blocked normal equations at the trial string into the trial state.  pcs_lm_trial_finish queues the decision (pcs_lm_decide's
 * arithmetic + the loop's termination rules from `ctrl`), which makes an accepted trial the current state by flipping flags[2] (no
 * copy), and writes the 12-double read-back into stats_host (page-locked; may be NULL).  pcs_lm_trial = both.  A loop over observation
 * shards (one process per GPU) queues its all-reduce of the trial state BETWEEN the two halves, on the same stream (RCCL: stream-
 * ordered, no host synchronisation); because that collective is queued on a fixed address, such a loop sets
 * PCS_LM_FIXED_TRIAL_BUFFER: state 0 stays current, trials are built into state 1 and an accepted one is copied over state 0.
 * Every kernel of the sequence first reads flags[0] and does nothing when it is set, so a caller may queue the NEXT trial before it has
 * seen this one's verdict: the GPU never waits for the host between trials.
 *   ctrl (12 doubles, device): [0] stop code, 0 = running (1 gtol reached before the step, 2 `ctrl[7]` consecutive rejections, 3 ftol,
 *        4 xtol, 5 `ctrl[3]` accepted steps, 9 the one-launch dense solve gave up — set spd_algorithm = PCS_SPD_LAUNCHES, clear ctrl[0] and
 *        flags[0] and queue the trial again), [1] consecutive rejections, [2] accepted steps, [3] iteration limit, [4] ftol, [5] xtol,
 *        [6] gtol, [7] rejection limit, [8] trials decided, [9] the factor a rejection applies to lambda before the first accepted step
 *        (0 = 4, as after it), [10], [11] a gain ratio above ctrl[10] multiplies lambda by ctrl[11] instead of 1/3 (0 = off)
 *   stats (12 doubles, device): pcs_lm_decide's eight, [8] the stop code after this trial, [9] the trial's number (ctrl[8]) or -1 for a
 *        launch that found the flag raised (nothing was computed; everything else in `stats` is then stale), [10] the current state
 *        after this trial (0 / 1), [11] lambda for the next trial.
 *   PCS_LM_VOTES: every rank writes "my dense solve gave up" (0 / 1) into the word behind the trial state's packed buffer (hence
 *        pcs_normal_layout's length + 1), the loop's all-reduce sums it with the blocks, and the decision voids the trial on EVERY
 *        rank when the sum is positive — the ranks repeat it together.
 * All pointers are device memory except stats_host / result_host; sizes as for the entry points named above.  After the loop has ended
 * the buffers may only be touched on `stream` (the speculative trial behind the end is still draining). */
#define PCS_LM_FIXED_TRIAL_BUFFER 1
#define PCS_LM_VOTES 2
typedef struct pcs_lm_buffers {
    double *packed[2];                    /* the two states: [A | B | C | g | cost | votes] (pcs_normal_layout's length + 1), 16-byte aligned */
    double *ps[2];                        /* their parameter strings, n_params each */
    int32_t *flags;                       /* 4 words: [0] stop, [1] the last trial was accepted, [2] the current state, [3] unused */
    const uint8_t *fixed;
    double *lambda;
    double *linvt, *u, *V, *S, *rhs, *dvec, *gm;   /* pcs_schur_prepare's outputs */
    int32_t *status;
    double *xlead, *w, *spd_work;         /* n_lead | n_trail | pcs_dense_spd_work_len(n_lead) */
    double *delta;                        /* n_params */
    double *ctrl;
    double *stats;
    double *stats_host;
    int32_t spd_algorithm;                /* PCS_SPD_* */
    int32_t mode;                         /* PCS_LM_* bits */
    /* optional: when the loop ends with this trial (stop code != 0), the final state — gradient and parameters at the n_free indices
     * free_idx (device, int64), then sum r^2 — goes to result_host (page-locked, mapped; 2 n_free + 1 doubles) BEFORE the read-back of
     * the trial: a host that polls stats_host[9] needs no further copy and no synchronisation to return the solution */
    const int64_t *free_idx;
    int64_t n_free;
    double *result_host;
    /* engine option "deterministic": workspace of the ordered S -= V V' (pcs_schur_syrk_work_len(n_lead, n_trail) doubles; may be NULL
     * when that is 0) */
    double *syrk_work;
    int64_t syrk_work_len;
} pcs_lm_buffers;
int pcs_lm_trial_build(pcs_engine *h, const pcs_lm_buffers *b, void *stream);
int pcs_lm_trial_finish(pcs_engine *h, const pcs_lm_buffers *b, void *stream);
int pcs_lm_trial(pcs_engine *h, const pcs_lm_buffers *b, void *stream);

/* The two products of the Schur step around the dense solve, on raw device pointers (float64, row-major):
 *   pcs_schur_syrk   S -= V V' on the LOWER triangle of S (n_lead x n_lead, row stride lds; V n_lead x n_trail, row stride ldv) and,
 *                    when d_u is given, rhs += V u — FP64 matrix cores, one workgroup per 32 x 32 tile and K split
 *                    (csrc/ba_schur.hpp); the partial sums of a split meet in f64 atomics (last bits run-to-run dependent);
 *   pcs_schur_vtx    w = V' x (n_trail outputs).
 * They replace the rocBLAS GEMM / GEMV calls of the reference consumer's step (optimisation_handling.py:88-98) in the device LM
 * loop; queued on `stream` (NULL = the default stream). */
int pcs_schur_syrk(int device, int64_t n_lead, int64_t n_trail, const double *d_V, int64_t ldv, double *d_S, int64_t lds, const double *d_u,
                   double *d_rhs, void *stream);
int pcs_schur_vtx(int device, int64_t n_lead, int64_t n_trail, const double *d_V, int64_t ldv, const double *d_x, double *d_w, void *stream);
/* pcs_schur_syrk with an ORDER: the partial products of a K split are stored to d_work (pcs_schur_syrk_work_len(n_lead, n_trail) doubles;
 * 0 = this size is not split and d_work is not touched) and subtracted split by split by a second kernel — the same bits on every run
 * and on every rank of a sharded solve (the reference's consumer is deterministic: scipy on one thread, optimisation_handling.py:88-98).
 * What engine option "deterministic" makes pcs_lm_trial_build use. */
int64_t pcs_schur_syrk_work_len(int64_t n_lead, int64_t n_trail);
int pcs_schur_syrk_ordered(int device, int64_t n_lead, int64_t n_trail, const double *d_V, int64_t ldv, double *d_S, int64_t lds, const double *d_u,
                           double *d_rhs, double *d_work, int64_t work_doubles, void *stream);

/* S x = rhs for a dense symmetric positive definite S (float64, n x n row-major with row stride ld, LOWER triangle read and
 * overwritten by its Cholesky factor): the reduced system of the Schur step above — what optimisation_handling.py:88-98 leaves to
 * scipy's trf / lsmr on the host.  Two forms of the same blocked factorisation (32 x 32 tiles):
 *   PCS_SPD_ONE_LAUNCH  one persistent launch (csrc/ba_chol_persist.hpp): every tile lives in the LDS of one workgroup for the whole
 *                       launch, block columns are handed over through HBM (write-through stores + one counter per column), the
 *                       right-hand side rides along as one more block row, the backward substitution is a chain of 32-word hand-offs
 *                       between the owners of the diagonal tiles.  n <= 1 984 on a 256-CU part (8 tiles per workgroup);
 *   PCS_SPD_LAUNCHES    one launch per block column + substitution launches (csrc/ba_dense_chol.hpp): any n <= 32 768;
 *   PCS_SPD_AUTO        the first where it fits (environment PCS_CHOL_LAUNCHES=1: always the second).
 * All pointers are device memory; d_work holds pcs_dense_spd_work_len(n) doubles; *d_status |= 2 when a pivot is not positive,
 * |= 4 when the one-launch form gave up waiting (another process holds the compute units: the results are not valid — solve
 * again with PCS_SPD_LAUNCHES); queued on `stream` (NULL = the default stream). */
#define PCS_SPD_AUTO 0
#define PCS_SPD_LAUNCHES 1
#define PCS_SPD_ONE_LAUNCH 2
int64_t pcs_dense_spd_work_len(int64_t n);
int pcs_dense_spd_solve_algo(int device, int64_t n, double *d_S, int64_t ld, const double *d_rhs, double *d_x, double *d_work, int32_t *d_status, void *stream,
                             int algorithm);
int pcs_dense_spd_solve(int device, int64_t n, double *d_S, int64_t ld, const double *d_rhs, double *d_x, double *d_work, int32_t *d_status, void *stream);
/* The same with the one-launch form's time limit spelled out: every wait of a workgroup for a hand-over gives up after timeout_us
 * microseconds (1 .. 60 000 000; the other entry points use 250 000), sets bit 2 of *d_status and lets the whole grid drain.  Inside an LM
 * trial (pcs_lm_trial*) the limit is the engine option "spd_timeout_us".  A caller that sees bit 2 repeats the solve with
 * PCS_SPD_LAUNCHES on the ORIGINAL matrix (the abandoned launch has overwritten part of the lower triangle). */
int pcs_dense_spd_solve_opts(int device, int64_t n, double *d_S, int64_t ld, const double *d_rhs, double *d_x, double *d_work, int32_t *d_status, void *stream,
                             int algorithm, int64_t timeout_us);

/* Which entry of H / g / cost every accumulator register of the normal-equations kernel stands for (host function, no
 * GPU needed): out[m][lane][r][2], m < 2 MFMAs, lane < 64, r < 4 registers = the two local column ids (index into
 * a J row, 30 = the residual column) of D_m[(lane >> 4) + 4 r][lane & 15], or -1, -1 where the register is not owned.
 * pass 0 = shared (camera + pose + residual), 1 = (cam, key) point pass, 2 = (image, key) point pass.  Pass 2 is a
 * segmented sum in matrix form: its registers hold, for 16 runs at a time, the symmetric 3 x 3 sum over the
 * pose-translation columns (local ids 18..20) from which the run's pose-point block is finished; register r of lane l
 * belongs to local run (l >> 4) + 4 r.
 * Diagnostic aid: tests/test_host_logic.py checks that every needed column pair is owned exactly once. */
int pcs_normal_entry_map(int chain, int pass, int32_t *out);
/* The packed destination descriptor of every accumulator register of ba_normal_mfma_kernel (passes 0 and 1; host function, no
 * GPU needed): out[m][lane][r], layout documented at entry_descriptor (csrc/ba_normal.hpp).  trail_group = 2 (pose) / 3 (point)
 * for the blocked layout, -1 for the dense one.  tests/test_host_logic.py decodes them for sample runs exactly like the
 * kernel's flush does and checks every owned entry's address against the column pair pcs_normal_entry_map reports. */
int pcs_normal_descriptors(int chain, int pass, int trail_group, int32_t *out);

/*
 * Generated chains.  The reference composes ANY list of function blocks (afb:735-748) and code-generates the loss / Jacobian /
 * chain rule for it (afb:290-419, afb:492-652, mm:147-263); user-written blocks are its extension point (afb:689-775).
 * pycamset_amd/chain_compiler.py does the same for the GPU: for a composition
 *     [projection | user block with 2 outputs] + {rigidTform3d | extrinsic3D | user block}* + [template_points | free_point | user source]
 * it emits the straight-line evaluation of one detection (the user blocks' device bodies pasted in), has hipcc compile it with
 * csrc/ba_generic.hpp for gfx950 and hands the code object to pcs_genchain_create.
 *   code_object_path   .hsaco with the entry points pcs_genchain_prep / pcs_genchain_eval_{1,2,3}[_f32] / pcs_genchain_compact_{2,3}[_f32]
 *   row_len = P (sum of the blocks' parameter counts, <= 64), uses_template: the source is template_points
 *   rigid parameter groups (one Rodrigues slab each): first parameter-string column and entity count per group;
 *   user_off[u]: first column of the parameter group of user block u; intr_off / point_off: projection / free_point groups
 *   (which group and which index — camera / image / key — every block reads is compiled into the code object)
 *   dtype: PCS_F64 | PCS_F32 | PCS_MIXED as for pcs_create (FP64 arithmetic, float streams)
 * Outputs and layouts as pcs_eval: resid (N, 2), jac (2N, P) dense block rows, u row then v row, elements of the handle's dtype.
 * pcs_genchain_set_unfixed + pcs_genchain_eval_compact[_device] write only the unfixed columns, in CSR data order, at the store
 * (`data[:n][good_mask]`, afb:644-651, never exists as a dense array): keep[i] bit j = local column j of detection i is free,
 * row_off[i] = offset of its u row in the data array.
 */
typedef struct pcs_genchain pcs_genchain;
int pcs_genchain_create(pcs_genchain **out, const char *code_object_path, int row_len, int uses_template, int n_groups, const int64_t *group_off,
                     const int32_t *group_count, int n_user, const int64_t *user_off, int64_t intr_off, int64_t point_off, int64_t n_params, int64_t n_cams,
                     int64_t n_imgs, int64_t n_keys, int dtype, int device);
int pcs_genchain_destroy(pcs_genchain *h);
int pcs_genchain_row_len(const pcs_genchain *h);
int pcs_genchain_set_detections_table(pcs_genchain *h, const double *det5, int64_t n);
int pcs_genchain_set_template(pcs_genchain *h, const double *points);
int pcs_genchain_eval(pcs_genchain *h, const double *param_str, void *resid, void *jac);
int pcs_genchain_eval_device(pcs_genchain *h, const double *d_param_str, void *d_resid, void *d_jac, void *stream);
/* A step of a generated chain is ONE launch (the default, like pcs_eval's fused kernel: every wave prepares the Rodrigues slabs of its
 * tile's (camera, image) pairs itself); on = 0 puts the slab preparation in a launch of its own in front.  Same bits either way. */
int pcs_genchain_set_one_launch(pcs_genchain *h, int on);
int pcs_genchain_set_unfixed(pcs_genchain *h, const uint64_t *keep, const int64_t *row_off, int64_t nnz);
int pcs_genchain_eval_compact(pcs_genchain *h, const double *param_str, void *resid, void *data);
int pcs_genchain_eval_compact_device(pcs_genchain *h, const double *d_param_str, void *d_resid, void *d_data, void *stream);
/* Products with the chain's Jacobian kept on the device — what a solver needs from J (optimisation_handling.py:88-98: column norms,
 * J^T f, mat-vecs), for ANY generated chain: pcs_genchain_linearize writes residual + dense block rows at param_str into the
 * handle's buffers (PCS_F64 chains), pcs_genchain_matfree applies them (csrc/ba_blockrow.hpp) with the op codes and vector shapes
 * of pcs_matfree.  pcs_genchain_set_blocks (once, after create) tells which global column a local column stands for: block b
 * covers local columns [col0_b, col0_b + np_b), its parameters of entity e (link_b: 0 camera, 1 image, 2 key) start at start_b + np_b e. */
int pcs_genchain_set_blocks(pcs_genchain *h, int n_blocks, const int32_t *col0, const int32_t *n_params, const int32_t *link, const int64_t *start);
int pcs_genchain_linearize(pcs_genchain *h, const double *param_str);
int pcs_genchain_matfree(pcs_genchain *h, int op, const double *in, double *out, double *cost);
/* The exact Levenberg-Marquardt step for ANY generated chain (round 5; csrc/ba_blockgram.hpp).  Replaces what scipy's trf does with the
 * CSR Jacobian of a composed chain (optimisation_handling.py:88-98) for chains the three hand-fused engines do not cover — user blocks
 * included.  The packed state is pcs_normal_layout's [A | B | C | g | cost] (pcs_genchain_normal_layout: same out5).  A chain whose LAST
 * parameter group is one rigid transform per image (6) or one point per key (3) has that group as the trailing entities — B and the
 * tb x tb blocks of C are filled, the dense factorisation is of the leading part only (Schur step, like the hand-fused chains); every
 * other chain, and any chain after pcs_genchain_set_option("dense_normal", 1), has every parameter leading: A = J^T J DENSE
 * (n_params x n_params, upper triangle written), out5 = {n_params, 0, 3, n_params^2 + n_params + 1, n_params}.
 * pcs_genchain_normal_blocks_device evaluates the chain at d_param_str (block rows into the handle's buffers) and contracts them on the
 * FP64 matrix cores.  pcs_genchain_lm_trial_build / _finish / pcs_genchain_lm_trial are pcs_lm_trial_build / _finish / pcs_lm_trial for
 * the handle (same pcs_lm_buffers, sized from pcs_genchain_normal_layout like pcs_normal_layout's — in the dense form V, linvt, u and w hold
 * one double each —, same decision, read-back and stop word); mode must hold
 * PCS_LM_FIXED_TRIAL_BUFFER (the generated kernel reads its string from a fixed address: the trial is built at ps[1] into packed[1],
 * an accepted one is copied over state 0).  Limits: FP64 chains, row length <= 63, n_params <= PCS_NORMAL_MAX_PARAMS.
 * Options (pcs_genchain_set_option): "spd_timeout_us" (as pcs_set_option), "timing" (0: no start / stop events around evaluations),
 * "dense_normal" (above; changes the layout: set it before buffers are sized), "deterministic" (1: the ORDERED contraction — every
 * segment's matrix to a workspace, groups of consecutive segments added in table order, the K split of schur_syrk through
 * pcs_lm_buffers.syrk_work: the same bits on every run; refused for chains whose blocks share a parameter group),
 * "gram_debug" (measurements only). */
int pcs_genchain_normal_layout(const pcs_genchain *h, int64_t *out5);
int pcs_genchain_normal_blocks_device(pcs_genchain *h, const double *d_param_str, double *d_packed, void *stream);
int pcs_genchain_lm_trial_build(pcs_genchain *h, const pcs_lm_buffers *b, void *stream);
int pcs_genchain_lm_trial_finish(pcs_genchain *h, const pcs_lm_buffers *b, void *stream);
int pcs_genchain_lm_trial(pcs_genchain *h, const pcs_lm_buffers *b, void *stream);
int pcs_genchain_set_option(pcs_genchain *h, const char *key, int64_t value);
/* The pieces of a trial as calls of their own — pcs_schur_prepare / pcs_schur_finish / pcs_lm_decide for the handle's layout — for a
 * loop the host steers (a sharded solve with a host-staged collective); pcs_schur_syrk, pcs_dense_spd_solve and pcs_schur_vtx take
 * sizes, not handles, and serve both kinds of handle. */
int pcs_genchain_schur_prepare(pcs_genchain *h, double *d_packed, const uint8_t *d_fixed, const double *d_lambda, double *d_linvt, double *d_u,
                               double *d_V, double *d_S, double *d_rhs, double *d_dvec, double *d_gm, int32_t *d_status, void *stream);
int pcs_genchain_schur_finish(pcs_genchain *h, const double *d_linvt, const double *d_u, const double *d_w, const double *d_xlead, const uint8_t *d_fixed,
                              double *d_delta, const double *d_ps_in, double *d_ps_out, void *stream);
int pcs_genchain_lm_decide(pcs_genchain *h, const double *d_cost_old, const double *d_cost_new, const double *d_dvec, const double *d_gm, const double *d_delta,
                           const double *d_ps, const uint8_t *d_fixed, int32_t *d_status, double *d_lambda, double *d_stats, void *stream);
int pcs_genchain_device_buffers(pcs_genchain *h, void **d_resid, void **d_jac);
int pcs_genchain_synchronize(pcs_genchain *h, void *stream);
int pcs_genchain_last_kernel_ms(pcs_genchain *h, float *slab_prep_ms, float *eval_ms);

/*
 * Legacy residual-only cost.
 * Replaces: bundle_adjustment_costfn / numpy_bundle_adjustment_costfn (compiled_helpers.py:518-549),
 * called once per candidate pose set during initialisation (template_handler.py:535-592).
 *   im_points  (n_imgs, n_keys, 3)  target points already transformed by each image's pose
 *   proj       (n_cams, 3, 4)       K [R|t] per camera
 *   intrinsics (n_cams, 3, 3), dists (n_cams, 5) = [k0, k1, p0, p1, k2]
 *   errors     (2N)                 [u0 err, v0 err, u1 err, ...] for the engine's detection table
 */
int pcs_legacy_cost(pcs_engine *h, const double *im_points, const double *proj, const double *intrinsics, const double *dists,
                    double *errors);

/* Block until everything queued on `stream` (NULL = engine stream) has finished. */
int pcs_synchronize(pcs_engine *h, void *stream);
/* Duration of the most recent evaluation's kernels, ms: each kernel's OWN start / stop events (hipExtLaunchKernelGGL on the
 * launch stream), so slab_prep_ms is the slab_prep kernel's duration without the gap to the next launch — the figure
 * rocprofv3 reports.  A one-launch step (option "fuse_prep") has no slab_prep kernel: slab_prep_ms = 0. */
int pcs_last_kernel_ms(pcs_engine *h, float *slab_prep_ms, float *eval_ms);
/* Mean kernel durations over the evaluations kept in the event ring (option "event_ring" = R keeps
 * the last R evaluations; the ring is reset when the option is set).  Used by bench.py for the
 * roofline figure: live HIP-event timing of every launch of the timed region. */
int pcs_kernel_ms_mean(pcs_engine *h, int64_t *count, float *slab_prep_ms, float *eval_ms);
/* The individual durations behind pcs_kernel_ms_mean, oldest first: up to `capacity` (slab_prep, eval) pairs are
 * written, *count = how many.  bench.py takes its median / min / mean from these (the reference's analogue is
 * the sample list of general_utils.benchmark(), utils/general_utils.py:62-104). */
int pcs_kernel_ms_samples(pcs_engine *h, int64_t capacity, float *slab_prep_ms, float *eval_ms, int64_t *count);
/* Tuning knobs ("variant", "wgs_per_cu", "tiles_per_wg", "event_ring", "timing_every", "compact_variant",
 * "fuse_prep" (-1 automatic / 0 / 1: one launch per step — every wave of the evaluation kernel prepares the R, t, dR/dr
 * slabs of its own tile instead of a slab_prep launch in front; automatic = tables in the reference's run order, FP64 outputs
 * at any size, float outputs up to "fuse_prep_max_n" detections; same slab element functions, same bits; the matrix-free
 * operators then need pcs_linearize), "lazy_done_event" (1: the engine's ordering event is recorded when somebody waits for
 * it, not after every step, for work on the engine's own or the default stream; 2: on caller streams too — the caller then keeps
 * the stream alive until it resets the option, which records what is pending; 0: after every enqueue), "timing_every" (0: no start / stop
 * events at all, neither around evaluations nor around normal-equations builds),
 * "matfree_lds", "xcd_remap", "waves_per_wg", "pack_indices", "normal_rows", "normal_imgkey_product",
 * "normal_imgkey_wgs_per_cu", "normal_sort_tables"; "normal_debug" is a bit mask of profiling switches of pcs_normal_equations — phases
 * or whole passes are skipped and the results are wrong while it is non-zero); see DESIGN.md.
 * Unknown keys -> PCS_ERR_ARG. */
int pcs_set_option(pcs_engine *h, const char *key, int64_t value);
/*
 * Batched n-view triangulation.
 * Replaces: nb_triangulate_full (compiled_helpers.py:609-643) = per point nb_undistort (ch:409-431) +
 *           nb_triangulate_nviews (ch:645-663, smallest right singular vector of the 3n x (4+n) DLT
 *           matrix); front end CameraSet.multi_cam_triangulate (cameras/camera_set.py:343-402), which calls it once
 *           per set of frames with the same cameras.
 * A pcs_triangulator owns the camera table, device copies of the observations, the kernel's scratch and (optionally)
 * the output, so repeated calls allocate nothing:
 *   pcs_tri_set_cameras              proj (n_cams,3,4), intrinsics (n_cams,3,3), dists (n_cams,5) = [k0,k1,p0,p1,k2]
 *   pcs_tri_set_observations         host arrays, copied: cam (n_obs) int32, uv (n_obs,2) sorted by point; point j owns
 *                                    rows [start_inds[j], start_inds[j+1]) (n_pts+1 entries, like the reference's);
 *                                    cameras are range-checked -> PCS_ERR_RANGE
 *   pcs_tri_set_observations_device  the same arrays already in device memory (caller-owned, not copied, not checked)
 *   pcs_tri_run                      queue the kernel on `stream` (NULL = the handle's own stream); d_pts = device
 *                                    buffer (n_pts,3) of the caller, or NULL = handle-owned output
 *   pcs_tri_points                   copy the handle-owned output to the host (blocking; PCS_ERR_STATE when the last run
 *                                    wrote to a caller buffer instead)
 * Runs on different streams are ordered by the handle (scratch and output are shared): an event recorded after every run
 * is waited for by the next run, by pcs_tri_points and by the setters, whatever stream the run was queued on.
 * pcs_triangulate is the stateless convenience form (temporary handle: allocations and copies on every call).
 */
typedef struct pcs_triangulator pcs_triangulator;
int pcs_tri_create(pcs_triangulator **out, int device, int64_t n_cams);
int pcs_tri_destroy(pcs_triangulator *t);
int pcs_tri_set_cameras(pcs_triangulator *t, const double *proj, const double *intrinsics, const double *dists);
int pcs_tri_set_observations(pcs_triangulator *t, int64_t n_obs, const int32_t *cam, const double *uv, int64_t n_pts, const int64_t *start_inds);
int pcs_tri_set_observations_device(pcs_triangulator *t, int64_t n_obs, const int32_t *d_cam, const double *d_uv, int64_t n_pts,
                                    const int64_t *d_start_inds);
/* The grouping CameraSet.multi_cam_triangulate does in front of nb_triangulate_full (cameras/camera_set.py:371-378: two np.unique calls over
 * the (image, key...) columns — which features are seen by more than one camera, their rows in table order, start indices in order of
 * first appearance), on the device: n table rows as caller-owned device arrays (camera index, dense feature id in [0, n_features),
 * measurement), grouped by feature like TargetDetection.get_data returns them -> the handle's current observations (pcs_tri_run next).
 * *n_pts / *n_kept: features kept / their rows.  *grouped = 0 (nothing set): a feature's rows are not consecutive — group on the host
 * (the reference's consecutive slices then mix features; the host mirror reproduces that).  One host synchronisation. */
int pcs_tri_group_device(pcs_triangulator *t, int64_t n, const int32_t *d_cam, const int32_t *d_feat, const double *d_uv, int64_t n_features,
                         int64_t *n_pts, int64_t *n_kept, int32_t *grouped, void *stream);
int pcs_tri_run(pcs_triangulator *t, double *d_pts, void *stream);
int pcs_tri_points(pcs_triangulator *t, double *pts);
int pcs_tri_synchronize(pcs_triangulator *t, void *stream);
int pcs_tri_last_kernel_ms(pcs_triangulator *t, float *kernel_ms);
int pcs_triangulate(int device, int64_t n_obs, const int32_t *cam, const double *uv, int64_t n_pts, const int64_t *start_inds,
                    int64_t n_cams, const double *proj, const double *intrinsics, const double *dists, double *pts,
                    float *kernel_ms);

/* Page-locked host memory for outputs: pcs_eval / pcs_eval_compact copy device -> host at PCIe rate
 * into such buffers (a pageable destination is several times slower).  Replaces nothing in the
 * reference (NumPy owns every array there, afb:561); SURVEY 8 f1 "zero-copy hand-off". */
int pcs_host_alloc(void **out, int64_t bytes);
int pcs_host_free(void *p);
/* Streaming probe of the device's achievable HBM rate with this engine's access shape (16 B per lane):
 * kind 0 fill, 1 non-temporal fill, 2 copy, 3 non-temporal copy, 4 non-temporal fill in the fused kernel's
 * shape (every wave writes whole 10 752-byte chunks), 5..8 the same with 21 KiB / 1 KiB / 64 KiB / 256 KiB chunks;
 * `bytes` per launch (per buffer).
 * Measurement aid for DESIGN.md / bench.py --membench; not on the evaluation path. */
int pcs_membench(int device, int kind, int64_t bytes, int iters, int blocks_per_cu, float *mean_ms);
/* Engine-owned device scratch for outputs (engine dtype); valid until the next set_detections. */
int pcs_device_buffers(pcs_engine *h, void **d_resid, void **d_jac);

#ifdef __cplusplus
}
#endif
#endif /* PCS_HIP_H */
