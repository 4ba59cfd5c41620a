// pcs_engine.hip — HIP kernels + C ABI of the MI355X bundle-adjustment cost/Jacobian engine.
//
// gfx950 (CDNA4) only: wave64, 256 CUs in 8 XCDs, 160 KiB LDS per CU, HBM3E.
// The path is an HBM-write-bound stream (380 B/detection for the FP64 template chain, of which
// 352 B are stores) with no dense contraction, so no MFMA there; the one contraction of the engine, the
// block-reduced normal equations, runs on the FP64 matrix cores (ba_normal.hpp).  See DESIGN.md.
//
// Kernels: ba_kernels.hpp (evaluation, compaction, legacy cost), ba_matfree.hpp (J products without J),
// ba_normal.hpp (J^T J / J^T r), ba_triangulate.hpp; device maths: ba_device.hpp.  This file: launch plumbing + the C ABI.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <atomic>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pcs_hip.h"
#include "ba_device.hpp"
#include "ba_kernels.hpp"
#include "ba_matfree.hpp"
#include "ba_normal.hpp"
#include "ba_reduce.hpp"
#include "ba_schur.hpp"
#include "ba_lm_fused.hpp"
#include "ba_dense_chol.hpp"
#include "ba_chol_persist.hpp"
#include "ba_triangulate.hpp"

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
using namespace pcs;

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) return fail(PCS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct pcs_engine {
    int chain = 0, dtype = 0, device = 0, P = 0;
    int64_t n_cams = 0, n_imgs = 0, n_keys = 0, n_params = 0, n = 0;
    int64_t extr_off = 0, pose_off = 0, point_off = 0;
    size_t msize = 8;   // bytes of a measurement scalar on the device (4 for PCS_F32); slabs and arithmetic are always FP64
    size_t osize = 8;   // bytes of the residual / Jacobian type written out (4 for PCS_F32 and PCS_MIXED)
    hipStream_t stream = nullptr;
    // ring of event quadruples: (slab_prep start, slab_prep stop, evaluation start, evaluation stop).  Both kernels carry their OWN
    // start / stop events (hipExtLaunchKernelGGL), so slab_prep's figure is its duration, not duration + the gap to the next
    // launch (round 2 reported start-to-start, 2.5 x what rocprofv3 sees); a one-launch step (PREP) has no slab_prep: 0.
    std::vector<hipEvent_t> ev;
    std::vector<uint8_t> ev_has_prep;   // per quadruple: a slab_prep launch was timed
    int64_t ev_ring = 1;         // quadruples in the ring
    int64_t ev_count = 0;        // evaluations recorded since the ring was (re)created
    bool events_valid = false;
    // Attach start/stop events to the kernel launches of every `timing_every`-th evaluation (0 = never).
    // Timed launches cost ~12 us of extra dispatch gaps per step on MI355X (profiles/r01/step_overhead.log),
    // so bench.py samples every 10th launch of its timed region instead of all of them.
    int64_t timing_every = 1;
    int64_t eval_count = 0;
    // Ordering across streams: `done` is recorded after everything the engine queues; the next piece of work that
    // touches the shared slabs / staging buffer / masks first waits for it — on the host where the host writes
    // (staging buffer, uploads), with hipStreamWaitEvent where another stream takes over.  (Round 1 kept a raw stream
    // handle and synchronised it later: a caller stream may be destroyed by then.)
    hipEvent_t done = nullptr;
    hipStream_t done_stream = nullptr;
    bool have_done = false;
    bool done_pending = false;   // `done` still has to be recorded on done_stream (see flush_done)
    bool lazy_done = true;       // option "lazy_done_event" (A/B switch)
    bool lazy_any_stream = false;   // "lazy_done_event" = 2: also on caller streams (the caller keeps the stream alive until it resets the option)
    // static inputs
    int32_t *d_cam = nullptr, *d_img = nullptr, *d_key = nullptr;  // split index arrays (only when the packed word does not fit)
    uint32_t *d_packed = nullptr;      // cam | image | key bit fields, one word per detection (DetTable, ba_device.hpp)
    int key_bits = 0, img_bits = 0;
    bool pack_indices = true;          // option "pack_indices" (A/B switch; takes effect at the next upload)
    int32_t *d_order = nullptr;  // (cam, image)-sorted visiting order of a scattered table (normal equations), or NULL
    int32_t *d_order_ck = nullptr, *d_order_ik = nullptr;  // (cam, key) / (image, key) orders for the point passes
    // the detection table copied out in the visiting orders (index words; measurements except for the (image, key) pass): the
    // passes then stream their inputs instead of gathering 4 + 16 scattered bytes per detection through the order
    void *d_sorted[3][5] = {};   // per pass (shared — only for a scattered table —, (cam, key), (image, key)): packed, cam, img, key, uv
    int sort_tables = 1;         // option: 0 = walk the original table through the visiting order (A/B)
    bool point_orders_tried = false;
    int normal_rows = 64;        // detections per LDS image of the normal-equations kernel (64 or 32)
    int waves_per_wg = 0;        // fused kernel: waves per workgroup (0 = automatic: fewer for small tables)
    void *d_uv = nullptr;
    std::vector<int32_t> h_cam, h_img, h_key;
    bool have_template = false;
    // per-step
    double *d_param = nullptr;
    double *h_param = nullptr;  // pinned staging
    void *d_cam_slab = nullptr, *d_pose_slab = nullptr, *d_points = nullptr;
    void *d_sink = nullptr;  // 64 B: where tail lanes put their residual
    // scratch outputs for the host-buffer entry points
    void *d_resid = nullptr, *d_jac = nullptr;
    int64_t jac_capacity = 0, resid_capacity = 0;
    // compaction
    uint32_t *d_keep = nullptr;
    int64_t *d_row_off = nullptr;
    int64_t nnz = -1;
    void *d_data = nullptr;
    int64_t data_capacity = 0;
    // matrix-free operators (f2)
    double *d_vin = nullptr, *d_vout = nullptr, *d_cost = nullptr;
    int64_t vin_capacity = 0, vout_capacity = 0;
    // normal equations (f2): scratch of the host-buffer entry point
    double *d_H = nullptr;
    int64_t H_capacity = 0;
    // legacy cost (f3)
    void *d_im_points = nullptr, *d_cam_tab = nullptr;
    int64_t im_points_capacity = 0;
    bool linearized = false;
    int normal_debug = 0;
    int normal_imgkey_wgs_per_cu = 0;   // 0 = default (48)
    int normal_imgkey_product = 1;   // pose-point blocks by ba_normal_imgkey_kernel (0: the boundary-walking pass, kept for A/B)
    bool matfree_lds = true;  // accumulate J^T products in workgroup-private LDS before the global atomics
    // launch geometry
    int n_cu = 256;
    // Launch geometry.  variant < 0 / wgs_per_cu <= 0 = automatic, from the MI355X sweeps in
    // profiles/r01/sweeps.md: transposed + non-temporal stores always; for a table whose 64-detection
    // tiles are (cam, image)-uniform (the reference's cam -> image -> key order) slabs are read
    // through L1/L2 with many small workgroups; for scattered tables slabs are staged in LDS by
    // few long-lived workgroups.
    int variant = -1;
    int64_t wgs_per_cu = 0;
    int compact_variant = 1;     // 1 = tile kernel with coalesced stores, 0 = per-lane stores
    bool xcd_remap = false;      // contiguous eighth of the table per XCD group (experiment; no measured effect)
    double tile_locality = 1.0;  // fraction of 64-detection tiles touching <= 2 distinct (cam, image) pairs
    double tile_segments = 1.0;  // mean number of (cam, image) runs per 64-detection tile (1.0 = every tile inside one run)
    // One-launch step: the evaluation kernel prepares the slabs of its tile per wave instead of a slab_prep launch in front
    // (ba_eval_kernel<..., PREP>).  -1 = automatic: run-ordered tables (few (camera, image) runs per tile); FP64 outputs at
    // any size (measured on MI355X, profiles/r03: rig-32 71.2 -> 65.0 us per step, rig-32-self 82.0 -> 78.7, free 60.1 -> 57.4,
    // an 8-way shard 14.2 -> 12.3, ring-8 12.3 -> 10.3), float outputs and residual-only evaluations only up to
    // fuse_prep_max_n detections (those kernels are issue-bound at large N: rig-32 mixed 35.3 -> 53.4 us, rig-128 f32 294 -> 500 us,
    // residual only at N = 1e6 9 + 4.4 -> 24.6 us with it).
    int fuse_prep = -1;
    int64_t fuse_prep_max_n = 250000;
    int64_t tiles_per_wg = 0;  // 0 = derive from wgs_per_cu
    size_t lds_limit = 160 * 1024;
    // time limit of every wait inside the one-launch dense solve of an LM trial (csrc/ba_chol_persist.hpp); option "spd_timeout_us"
    int64_t spd_timeout_us = 250000;
    // option "deterministic": every sum of the normal equations and of the Schur step is taken in an order the table and the launch
    // geometry fix (csrc/ba_reduce.hpp) instead of with f64 atomics in arrival order: two runs — and the ranks of a sharded loop — compute
    // the same bits
    int deterministic = 0;
    int fused_trial = 1;   // option: pcs_lm_trial_build uses schur_prep_kernel / schur_finish_kernel (csrc/ba_lm_fused.hpp); 0 = the separate launches (A/B)
    // host copies of the visiting orders (shared pass of a scattered table, (cam, key) pass): the deterministic mode's tables are built from them
    std::vector<int32_t> h_order, h_order_ck;
    // static tables + workspaces of the ordered second pass, per MFMA pass (0 shared, 1 (cam, key)); built at the first deterministic
    // build for the launch geometry in use, rebuilt when the table or the geometry changes
    struct DetPass {
        int64_t n = -1;
        int64_t tpw = 0;
        int32_t n_seg = 0, n_lr = 0, n_grp = 0, n_ent = 0, n_waves = 0;
        int32_t *d_idx = nullptr;      // one allocation: seg_base | lr_ptr | lr_segs | lr_ka | lr_kb | grp_ptr | cam_ptr | ent_ptr | ent_runs
        int64_t off[9] = {};
        double *d_work = nullptr;      // one allocation: part | Q | G
        int64_t work_off[3] = {};
    } det[2];
};

// `done` is recorded LAZILY where that is safe (round 3): an event record is a packet of its own between two launches, and a
// small step (config 2, an 8-way shard) paid for one after every evaluation although nothing ever waited for it.  For work
// queued on the engine's own stream or on the default stream (handles that outlive every call) mark_done only notes the
// stream; the record happens when somebody needs to wait — it then covers everything queued on that stream so far, a
// superset.  Work on any other caller stream is recorded at once: that stream may be destroyed before the next call.
static hipError_t flush_done(pcs_engine *h) {
    if (!h->done_pending) return hipSuccess;
    h->done_pending = false;
    return hipEventRecord(h->done, h->done_stream);
}
// everything queued so far has finished (host-side wait)
static hipError_t wait_done_host(pcs_engine *h) {
    if (!h->have_done) return hipSuccess;
    hipError_t e = flush_done(h);
    return e != hipSuccess ? e : hipEventSynchronize(h->done);
}
// work queued on `s` from here on runs after everything queued so far, whatever stream that was on.  Same stream as the
// previous enqueue: stream order already gives that.  Another stream: a device-side wait — except for the legacy
// default-stream handle, on which hipStreamWaitEvent of this runtime faults; the (rare) switch to or from it waits on the host.
static hipError_t order_after_done(pcs_engine *h, hipStream_t s) {
    if (!h->have_done || s == h->done_stream) return hipSuccess;
    hipError_t e = flush_done(h);
    if (e != hipSuccess) return e;
    if (s == hipStreamLegacy || s == nullptr || h->done_stream == hipStreamLegacy) return hipEventSynchronize(h->done);
    return hipStreamWaitEvent(s, h->done, 0);
}
static hipError_t mark_done(pcs_engine *h, hipStream_t s) {
    // (every enqueue calls order_after_done(h, s) first, so work on an earlier stream is already ordered before `s`)
    h->have_done = true;
    h->done_stream = s;   // used as a handle again only when it is the engine's own stream or the default stream
    if (h->lazy_done && (h->lazy_any_stream || s == h->stream || s == hipStreamLegacy)) {
        h->done_pending = true;
        return hipSuccess;
    }
    h->done_pending = false;
    return hipEventRecord(h->done, s);
}

// the event quadruple the next timed piece of work records into (see pcs_engine::ev)
static hipEvent_t *ring_slot(pcs_engine *h, bool has_prep) {
    const int64_t i = h->ev_count % h->ev_ring;
    h->ev_has_prep[i] = has_prep ? 1 : 0;
    return h->ev.data() + 4 * i;
}
// (slab_prep ms — 0 when that quadruple timed no slab_prep launch —, evaluation ms) of quadruple i; waits for it
static int ring_read(pcs_engine *h, int64_t i, float *prep_ms, float *eval_ms) {
    hipEvent_t *ev = h->ev.data() + 4 * i;
    HIPCHK(hipEventSynchronize(ev[3]));
    *prep_ms = 0.f;
    if (h->ev_has_prep[i]) HIPCHK(hipEventElapsedTime(prep_ms, ev[0], ev[1]));
    HIPCHK(hipEventElapsedTime(eval_ms, ev[2], ev[3]));
    return PCS_OK;
}

static int64_t padded_points(int64_t n_keys) { return (n_keys * 3 + 3) & ~(int64_t)3; }

static void free_det_tables(pcs_engine *h) {
    for (auto &d : h->det) {
        if (d.d_idx) (void)hipFree(d.d_idx);
        if (d.d_work) (void)hipFree(d.d_work);
        d = pcs_engine::DetPass{};
    }
}

extern "C" {

int pcs_version(void) { return 100; }
const char *pcs_last_error(void) { return g_err.c_str(); }

// ---- batched triangulation: a handle that owns the camera table, the observation buffers and the
// kernel's scratch, so that repeated calls (CameraSet.multi_cam_triangulate per frame set, cameras/camera_set.py:343-402)
// pay neither allocations nor — with device-resident inputs — copies.
struct pcs_triangulator {
    int device = 0;
    int64_t n_cams = 0;
    bool have_cams = false;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool timed = false;
    double *d_tab = nullptr;
    // handle-owned copies of host inputs (grown on demand)
    int32_t *d_cam = nullptr; double *d_uv = nullptr; int64_t *d_start = nullptr;
    int64_t obs_capacity = 0, uv_capacity = 0, pts_capacity = 0;
    // scratch + default output
    void *d_scr = nullptr, *d_scl = nullptr; double *d_pts = nullptr;
    int64_t scr_capacity = 0, scl_capacity = 0, out_capacity = 0;
    // current problem (device pointers: handle-owned or the caller's)
    const int32_t *cur_cam = nullptr; const double *cur_uv = nullptr; const int64_t *cur_start = nullptr;
    int64_t n_obs = 0, n_pts = -1;
    // the grouping in front of the triangulation (pcs_tri_group_device): per-feature counts, block sums of the scan, totals
    int32_t *d_count = nullptr; uint64_t *d_block_sums = nullptr; int64_t *d_totals = nullptr;
    int64_t count_capacity = 0, block_capacity = 0;
    int32_t *d_order = nullptr, *d_hist = nullptr;   // points by view count (built by the first run of a set of observations)
    int64_t order_capacity = 0;
    bool order_valid = false, sort_points = true;
    int variant = 1; // 1: views in registers + divide-free rotations (round 4; 3: eight instead of six register views per lane); 0: round 3's kernel (PCS_TRI_VARIANT=0)
    int lanes = 4;   // lanes per point: 1, 2, 4, 8 or 16 (profiles/r01/tri_legacy_bench.log: 4 is fastest at 2-22 views)
    // Ordering across streams, as in pcs_engine: `done` is recorded after every run; whatever touches the camera table, the
    // handle-owned observation copies, the scratch or the output next first waits for it — on the host where the host
    // writes or reads, with hipStreamWaitEvent where a run moves to another stream (scratch and output are shared).
    hipEvent_t done = nullptr;
    hipStream_t done_stream = nullptr;
    bool have_done = false;
    bool out_owned = false;   // the last run wrote the handle-owned output (pcs_tri_points has something to return)
};

static hipError_t tri_wait_done_host(pcs_triangulator *t) { return t->have_done ? hipEventSynchronize(t->done) : hipSuccess; }

static int tri_grow(void **buf, int64_t *cap, int64_t need, size_t elem) {
    if (need <= *cap) return PCS_OK;
    if (*buf) HIPCHK(hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
    HIPCHK(hipMalloc(buf, elem * (size_t)need));
    *cap = need;
    return PCS_OK;
}

int pcs_tri_create(pcs_triangulator **out, int device, int64_t n_cams) {
    if (!out || n_cams <= 0) return fail(PCS_ERR_ARG, "pcs_tri_create: bad arguments");
    *out = nullptr;
    const int ndev = pcs_device_count();
    if (ndev <= 0) return fail(PCS_ERR_NODEVICE, "pcs_tri_create: no HIP device visible (no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(PCS_ERR_ARG, "pcs_tri_create: device out of range");
    HIPCHK(hipSetDevice(device));
    pcs_triangulator *t = new pcs_triangulator();
    t->device = device;
    t->n_cams = n_cams;
    const char *lanes_env = getenv("PCS_TRI_LANES");   // A/B switch
    const int lanes = lanes_env ? atoi(lanes_env) : 4;
    t->lanes = (lanes == 1 || lanes == 2 || lanes == 8 || lanes == 16) ? lanes : 4;
    const char *var_env = getenv("PCS_TRI_VARIANT");
    t->variant = var_env ? atoi(var_env) : 1;
    t->sort_points = getenv("PCS_TRI_NO_SORT") == nullptr;   // A/B switch
    hipError_t e = hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&t->e0);
    if (e == hipSuccess) e = hipEventCreate(&t->e1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&t->done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&t->d_tab, sizeof(double) * n_cams * TRI_CAM_STRIDE);
    if (e != hipSuccess) {
        const int rc = fail(PCS_ERR_HIP, "pcs_tri_create: %s", hipGetErrorString(e));
        pcs_tri_destroy(t);
        return rc;
    }
    *out = t;
    return PCS_OK;
}

int pcs_tri_destroy(pcs_triangulator *t) {
    if (!t) return PCS_OK;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    (void)tri_wait_done_host(t);   // a run on a caller stream may still read the tables
    for (void *b : {(void *)t->d_tab, (void *)t->d_cam, (void *)t->d_uv, (void *)t->d_start, t->d_scr, t->d_scl, (void *)t->d_pts, (void *)t->d_order, (void *)t->d_hist,
                    (void *)t->d_count, (void *)t->d_block_sums, (void *)t->d_totals})
        if (b) (void)hipFree(b);
    if (t->e0) (void)hipEventDestroy(t->e0);
    if (t->e1) (void)hipEventDestroy(t->e1);
    if (t->done) (void)hipEventDestroy(t->done);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
    return PCS_OK;
}

int pcs_tri_set_cameras(pcs_triangulator *t, const double *proj, const double *intrinsics, const double *dists) {
    if (!t || !proj || !intrinsics || !dists) return fail(PCS_ERR_ARG, "pcs_tri_set_cameras: bad arguments");
    std::vector<double> tab((size_t)t->n_cams * TRI_CAM_STRIDE, 0.0);
    for (int64_t c = 0; c < t->n_cams; ++c) {
        double *r = tab.data() + c * TRI_CAM_STRIDE;
        for (int k = 0; k < 12; ++k) r[k] = proj[12 * c + k];
        const double *K = intrinsics + 9 * c;
        r[22] = K[0]; r[23] = K[2]; r[24] = K[4]; r[25] = K[5];
        for (int k = 0; k < 5; ++k) r[26 + k] = dists[5 * c + k];
    }
    HIPCHK(hipSetDevice(t->device));
    HIPCHK(tri_wait_done_host(t));   // a run queued on ANY stream may still read the table
    HIPCHK(hipStreamSynchronize(t->stream));
    HIPCHK(hipMemcpy(t->d_tab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
    t->have_cams = true;
    return PCS_OK;
}

int pcs_tri_set_observations(pcs_triangulator *t, int64_t n_obs, const int32_t *cam, const double *uv, int64_t n_pts, const int64_t *start_inds) {
    if (!t || n_obs < 0 || n_pts < 0 || !start_inds || (n_obs > 0 && (!cam || !uv))) return fail(PCS_ERR_ARG, "pcs_tri_set_observations: bad arguments");
    if (start_inds[0] != 0 || start_inds[n_pts] != n_obs) return fail(PCS_ERR_ARG, "pcs_tri_set_observations: start_inds must run from 0 to n_obs");
    for (int64_t j = 0; j < n_pts; ++j)
        if (start_inds[j + 1] < start_inds[j]) return fail(PCS_ERR_ARG, "pcs_tri_set_observations: start_inds must be non-decreasing");
    for (int64_t r = 0; r < n_obs; ++r)
        if (cam[r] < 0 || cam[r] >= t->n_cams) return fail(PCS_ERR_RANGE, "observation %lld has camera %d outside [0,%lld)", (long long)r, cam[r], (long long)t->n_cams);
    HIPCHK(hipSetDevice(t->device));
    HIPCHK(tri_wait_done_host(t));   // a run queued on ANY stream may still read the observation copies
    HIPCHK(hipStreamSynchronize(t->stream));
    t->n_pts = -1;
    int rc = tri_grow((void **)&t->d_cam, &t->obs_capacity, std::max<int64_t>(1, n_obs), sizeof(int32_t));
    if (rc) return rc;
    rc = tri_grow((void **)&t->d_uv, &t->uv_capacity, std::max<int64_t>(1, n_obs), 2 * sizeof(double));
    if (rc) return rc;
    rc = tri_grow((void **)&t->d_start, &t->pts_capacity, n_pts + 1, sizeof(int64_t));
    if (rc) return rc;
    if (n_obs) {
        HIPCHK(hipMemcpyAsync(t->d_cam, cam, sizeof(int32_t) * n_obs, hipMemcpyHostToDevice, t->stream));
        HIPCHK(hipMemcpyAsync(t->d_uv, uv, sizeof(double) * 2 * n_obs, hipMemcpyHostToDevice, t->stream));
    }
    HIPCHK(hipMemcpyAsync(t->d_start, start_inds, sizeof(int64_t) * (n_pts + 1), hipMemcpyHostToDevice, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));   // the caller may reuse its host arrays
    t->cur_cam = t->d_cam; t->cur_uv = t->d_uv; t->cur_start = t->d_start;
    t->n_obs = n_obs; t->n_pts = n_pts;
    t->order_valid = false;
    t->out_owned = false;   // results of an earlier problem are not this problem's
    return PCS_OK;
}

int pcs_tri_set_observations_device(pcs_triangulator *t, int64_t n_obs, const int32_t *d_cam, const double *d_uv, int64_t n_pts, const int64_t *d_start_inds) {
    if (!t || n_obs < 0 || n_pts < 0 || !d_start_inds || (n_obs > 0 && (!d_cam || !d_uv))) return fail(PCS_ERR_ARG, "pcs_tri_set_observations_device: bad arguments");
    t->cur_cam = d_cam; t->cur_uv = d_uv; t->cur_start = d_start_inds;   // caller-owned, not range-checked (stay on the device)
    t->n_obs = n_obs; t->n_pts = n_pts;
    t->order_valid = false;
    t->out_owned = false;
    return PCS_OK;
}

// The grouping CameraSet.multi_cam_triangulate does in front of nb_triangulate_full (cameras/camera_set.py:371-378), on the device
// (csrc/ba_triangulate.hpp, "the grouping in front of the triangulation"): from n table rows (camera, dense feature id, measurement;
// caller-owned device arrays, the table grouped by feature) to the handle's current observations — the rows of features seen by at
// least two cameras, in table order, and their start indices.  One host synchronisation (the counts).  *grouped = 0: the table is
// NOT grouped by feature (a feature's rows are not consecutive): nothing was set, the caller groups on the host.
int pcs_tri_group_device(pcs_triangulator *t, int64_t n, const int32_t *d_cam, const int32_t *d_feat, const double *d_uv, int64_t n_features,
                         int64_t *n_pts, int64_t *n_kept, int32_t *grouped, void *stream) {
    if (!t || n < 0 || n > INT32_MAX || n_features <= 0 || n_features > INT32_MAX || !n_pts || !n_kept || !grouped || (n > 0 && (!d_cam || !d_feat || !d_uv)))
        return fail(PCS_ERR_ARG, "pcs_tri_group_device: bad arguments");
    *n_pts = *n_kept = 0;
    *grouped = 1;
    HIPCHK(hipSetDevice(t->device));
    hipStream_t s = stream ? (hipStream_t)stream : t->stream;
    HIPCHK(tri_wait_done_host(t));   // a run queued on ANY stream may still read the observation copies this call overwrites
    HIPCHK(hipStreamSynchronize(t->stream));
    t->n_pts = -1;
    if (n == 0) {
        int rc0 = tri_grow((void **)&t->d_start, &t->pts_capacity, 1, sizeof(int64_t));
        if (rc0) return rc0;
        HIPCHK(hipMemsetAsync(t->d_start, 0, sizeof(int64_t), s));
        HIPCHK(hipStreamSynchronize(s));
        t->cur_cam = t->d_cam; t->cur_uv = t->d_uv; t->cur_start = t->d_start;
        t->n_obs = 0; t->n_pts = 0; t->order_valid = false; t->out_owned = false;
        return PCS_OK;
    }
    const int64_t n_blocks = (n + TRI_GROUP_BLOCK - 1) / TRI_GROUP_BLOCK;
    int rc = tri_grow((void **)&t->d_cam, &t->obs_capacity, n, sizeof(int32_t));
    if (rc) return rc;
    rc = tri_grow((void **)&t->d_uv, &t->uv_capacity, n, 2 * sizeof(double));
    if (rc) return rc;
    // one entry per kept run + 1.  A GROUPED table has at most n / 2 kept runs, but whether it is grouped is only known afterwards: in a table
    // whose features interleave every row can be the head of a kept run (writes up to start[n]; sized for n / 2 + 2 until this was found
    // by a fault in the full test suite — alone, the overrun stayed inside the allocation)
    rc = tri_grow((void **)&t->d_start, &t->pts_capacity, n + 2, sizeof(int64_t));
    if (rc) return rc;
    rc = tri_grow((void **)&t->d_count, &t->count_capacity, n_features, sizeof(int32_t));
    if (rc) return rc;
    rc = tri_grow((void **)&t->d_block_sums, &t->block_capacity, n_blocks, sizeof(uint64_t));
    if (rc) return rc;
    if (!t->d_totals) HIPCHK(hipMalloc(&t->d_totals, sizeof(int64_t) * 4));
    HIPCHK(hipMemsetAsync(t->d_count, 0, sizeof(int32_t) * n_features, s));
    HIPCHK(hipMemsetAsync(t->d_totals, 0, sizeof(int64_t) * 4, s));
    TriGroupArgs a{d_cam, d_feat, reinterpret_cast<const double2 *>(d_uv), t->d_count, t->d_block_sums, t->d_totals, t->d_cam,
                   reinterpret_cast<double2 *>(t->d_uv), t->d_start, n, n_features, (int32_t)n_blocks};
    hipLaunchKernelGGL(tri_group_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(tri_group_blocksum_kernel, dim3((unsigned)n_blocks), dim3(TRI_GROUP_BLOCK), 0, s, a);
    hipLaunchKernelGGL(tri_group_scan_sums_kernel, dim3(1), dim3(TRI_GROUP_BLOCK), 0, s, a);
    hipLaunchKernelGGL(tri_group_scatter_kernel, dim3((unsigned)n_blocks), dim3(TRI_GROUP_BLOCK), 0, s, a);
    HIPCHK(hipGetLastError());
    int64_t totals[4];
    HIPCHK(hipMemcpyAsync(totals, t->d_totals, sizeof totals, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (totals[2] != totals[3]) {   // more runs than features: some feature's rows are not consecutive
        *grouped = 0;
        return PCS_OK;
    }
    *n_kept = totals[0];
    *n_pts = totals[1];
    t->cur_cam = t->d_cam; t->cur_uv = t->d_uv; t->cur_start = t->d_start;
    t->n_obs = totals[0]; t->n_pts = totals[1];
    t->order_valid = false;
    t->out_owned = false;
    return PCS_OK;
}

int pcs_tri_run(pcs_triangulator *t, double *d_pts, void *stream) {
    if (!t) return fail(PCS_ERR_ARG, "pcs_tri_run: bad arguments");
    if (!t->have_cams) return fail(PCS_ERR_STATE, "pcs_tri_run: cameras not set");
    if (t->n_pts < 0) return fail(PCS_ERR_STATE, "pcs_tri_run: observations not set");
    if (t->n_pts == 0) return PCS_OK;
    HIPCHK(hipSetDevice(t->device));
    hipStream_t s = stream ? (hipStream_t)stream : t->stream;
    const bool need_order = t->variant != 0 && t->sort_points && !t->order_valid && t->n_pts < (1ll << 31);
    const bool grows = t->n_obs > t->scr_capacity || t->n_obs > t->scl_capacity || (!d_pts && t->n_pts > t->out_capacity) || (need_order && t->n_pts > t->order_capacity);
    if (t->have_done) {   // scratch and output are shared between runs: the previous one finishes first
        if (grows || s == hipStreamLegacy || t->done_stream == hipStreamLegacy) HIPCHK(hipEventSynchronize(t->done));   // frees need the host to wait
        else if (s != t->done_stream) HIPCHK(hipStreamWaitEvent(s, t->done, 0));
    }
    int rc = tri_grow(&t->d_scr, &t->scr_capacity, std::max<int64_t>(1, t->n_obs), 4 * sizeof(double));   // Householder row r_i per observation
    if (rc) return rc;
    rc = tri_grow(&t->d_scl, &t->scl_capacity, std::max<int64_t>(1, t->n_obs), 2 * sizeof(double));   // (1 / E_i, lambda_i)
    if (rc) return rc;
    const bool owned = !d_pts;
    if (owned) {
        rc = tri_grow((void **)&t->d_pts, &t->out_capacity, t->n_pts, 3 * sizeof(double));
        if (rc) return rc;
        d_pts = t->d_pts;
    }
    const int lanes = t->lanes;
    const dim3 grid((unsigned)((t->n_pts * lanes + 255) / 256));
    if (need_order) {
        // the visiting order of this set of observations, on the run's own stream (the caller's start_inds may have been produced there)
        rc = tri_grow((void **)&t->d_order, &t->order_capacity, t->n_pts, sizeof(int32_t));
        if (rc) return rc;
        if (!t->d_hist) HIPCHK(hipMalloc(&t->d_hist, sizeof(int32_t) * 512));
        HIPCHK(hipMemsetAsync(t->d_hist, 0, sizeof(int32_t) * 512, s));
        const dim3 pg((unsigned)((t->n_pts + 255) / 256));
        hipLaunchKernelGGL(tri_order_count_kernel, pg, dim3(256), 0, s, t->cur_start, t->n_pts, t->d_hist);
        hipLaunchKernelGGL(tri_order_scan_kernel, dim3(1), dim3(256), 0, s, t->d_hist);
        hipLaunchKernelGGL(tri_order_scatter_kernel, pg, dim3(256), 0, s, t->cur_start, t->n_pts, t->d_hist, t->d_order);
        HIPCHK(hipGetLastError());
        t->order_valid = true;
    }
#define PCS_TRI_LAUNCH(G_)                                                                                                     \
    hipExtLaunchKernelGGL(triangulate_kernel<G_>, grid, dim3(256), 0, s, t->e0, t->e1, 0, t->cur_cam, (const double2 *)t->cur_uv, \
                          t->cur_start, (const double *)t->d_tab, (double4 *)t->d_scr, (double2 *)t->d_scl, d_pts, t->n_pts)
#define PCS_TRI_LAUNCH_REG(G_, V_)                                                                                                     \
    hipExtLaunchKernelGGL((triangulate_reg_kernel<G_, V_>), grid, dim3(256), 0, s, t->e0, t->e1, 0, t->cur_cam, (const double2 *)t->cur_uv, \
                          t->cur_start, (const double *)t->d_tab, (double4 *)t->d_scr, (double2 *)t->d_scl, d_pts, t->n_pts, (const int32_t *)(t->order_valid ? t->d_order : nullptr))
    if (t->variant == 0) {   // round 3's form (views in the global scratch, IEEE divides): kept for A/B (PCS_TRI_VARIANT=0)
        if (lanes == 1) PCS_TRI_LAUNCH(1);
        else if (lanes == 2) PCS_TRI_LAUNCH(2);
        else if (lanes == 8) PCS_TRI_LAUNCH(8);
        else if (lanes == 16) PCS_TRI_LAUNCH(16);
        else PCS_TRI_LAUNCH(4);
    } else {                 // views in registers (6 or 8 per lane; further ones in the scratch), divide-free rotations
        if (lanes == 1) PCS_TRI_LAUNCH_REG(1, 8);
        else if (lanes == 2) PCS_TRI_LAUNCH_REG(2, 8);
        else if (lanes == 8) PCS_TRI_LAUNCH_REG(8, 8);
        else if (lanes == 16) PCS_TRI_LAUNCH_REG(16, 8);
        else if (t->variant == 3) PCS_TRI_LAUNCH_REG(4, 8);
        else PCS_TRI_LAUNCH_REG(4, 6);   // 24 views in registers at 156 VGPRs (three waves per SIMD): 47 us against 52 us for (4, 8); (4, 3) — four waves
                                         // per SIMD, 12 register views, the rest through the scratch records — 53 us, (4, 4) 49 us (profiles/r05/tri_bench_r05.log)
    }
#undef PCS_TRI_LAUNCH_REG
#undef PCS_TRI_LAUNCH
    HIPCHK(hipGetLastError());
    t->timed = true;
    t->out_owned = owned;
    t->have_done = true;
    t->done_stream = s;   // compared only, never used as a handle again
    HIPCHK(hipEventRecord(t->done, s));
    return PCS_OK;
}

int pcs_tri_points(pcs_triangulator *t, double *pts) {
    if (!t || !pts) return fail(PCS_ERR_ARG, "pcs_tri_points: bad arguments");
    if (t->n_pts < 0 || (t->n_pts > 0 && (!t->out_owned || !t->d_pts || t->out_capacity < t->n_pts)))
        return fail(PCS_ERR_STATE, "pcs_tri_points: the last run left no handle-owned result (run with d_pts = NULL first)");
    HIPCHK(hipSetDevice(t->device));
    HIPCHK(tri_wait_done_host(t));   // the run may have been queued on a caller stream
    if (t->n_pts) HIPCHK(hipMemcpyAsync(pts, t->d_pts, sizeof(double) * 3 * t->n_pts, hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    return PCS_OK;
}

int pcs_tri_synchronize(pcs_triangulator *t, void *stream) {
    if (!t) return fail(PCS_ERR_ARG, "pcs_tri_synchronize: bad arguments");
    HIPCHK(hipSetDevice(t->device));
    HIPCHK(hipStreamSynchronize(stream ? (hipStream_t)stream : t->stream));
    return PCS_OK;
}

int pcs_tri_last_kernel_ms(pcs_triangulator *t, float *kernel_ms) {
    if (!t || !kernel_ms) return fail(PCS_ERR_ARG, "pcs_tri_last_kernel_ms: bad arguments");
    if (!t->timed) return fail(PCS_ERR_STATE, "pcs_tri_last_kernel_ms: nothing has run yet");
    HIPCHK(hipEventSynchronize(t->e1));
    HIPCHK(hipEventElapsedTime(kernel_ms, t->e0, t->e1));
    return PCS_OK;
}

// stateless convenience form: one temporary handle per call (allocations + copies every time — use the handle API
// for repeated calls)
int pcs_triangulate(int device, int64_t n_obs, const int32_t *cam, const double *uv, int64_t n_pts, const int64_t *start_inds,
                    int64_t n_cams, const double *proj, const double *intrinsics, const double *dists, double *pts,
                    float *kernel_ms) {
    if (n_obs < 0 || n_pts < 0 || n_cams <= 0 || !start_inds || !proj || !intrinsics || !dists || (n_pts > 0 && !pts) ||
        (n_obs > 0 && (!cam || !uv)))
        return fail(PCS_ERR_ARG, "pcs_triangulate: bad arguments");
    if (n_pts == 0) return PCS_OK;
    pcs_triangulator *t = nullptr;
    int rc = pcs_tri_create(&t, device, n_cams);
    if (rc) return rc;
    rc = pcs_tri_set_cameras(t, proj, intrinsics, dists);
    if (!rc) rc = pcs_tri_set_observations(t, n_obs, cam, uv, n_pts, start_inds);
    if (!rc) rc = pcs_tri_run(t, nullptr, nullptr);
    if (!rc) rc = pcs_tri_points(t, pts);
    if (!rc && kernel_ms) rc = pcs_tri_last_kernel_ms(t, kernel_ms);
    const std::string keep = g_err;   // pcs_tri_destroy must not clobber the message
    pcs_tri_destroy(t);
    g_err = keep;
    return rc;
}

int pcs_host_alloc(void **out, int64_t bytes) {
    if (!out || bytes <= 0) return fail(PCS_ERR_ARG, "pcs_host_alloc: bad arguments");
    *out = nullptr;
    HIPCHK(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
    return PCS_OK;
}

int pcs_host_free(void *p) {
    if (p) HIPCHK(hipHostFree(p));
    return PCS_OK;
}

int pcs_membench(int device, int kind, int64_t bytes, int iters, int blocks_per_cu, float *mean_ms) {
    if (kind < 0 || kind > 8 || bytes < 4096 || iters < 1 || !mean_ms) return fail(PCS_ERR_ARG, "pcs_membench: bad arguments");
    if (device < 0 || device >= pcs_device_count()) return fail(PCS_ERR_NODEVICE, "pcs_membench: device %d not available", device);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    void *src = nullptr, *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (src) (void)hipFree(src);
        if (dst) (void)hipFree(dst);
    };
#define MBCHK(expr)                                                                                              \
    do {                                                                                                         \
        hipError_t _e = (expr);                                                                                  \
        if (_e != hipSuccess) { const int _rc = fail(PCS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); cleanup(); return _rc; } \
    } while (0)
    MBCHK(hipMalloc(&dst, bytes));
    if (kind == 2 || kind == 3) {
        MBCHK(hipMalloc(&src, bytes));
        MBCHK(hipMemset(src, 1, bytes));
    }
    MBCHK(hipEventCreate(&e0));
    MBCHK(hipEventCreate(&e1));
    const int64_t n16 = bytes / 16;
    const dim3 grid((unsigned)std::min<int64_t>((n16 + 255) / 256, (int64_t)prop.multiProcessorCount * std::max(1, blocks_per_cu)));
    auto launch = [&]() {
        switch (kind) {
            case 0: hipLaunchKernelGGL(membench_kernel<0>, grid, dim3(256), 0, nullptr, (const double2 *)src, (double2 *)dst, n16); break;
            case 1: hipLaunchKernelGGL(membench_kernel<1>, grid, dim3(256), 0, nullptr, (const double2 *)src, (double2 *)dst, n16); break;
            case 2: hipLaunchKernelGGL(membench_kernel<2>, grid, dim3(256), 0, nullptr, (const double2 *)src, (double2 *)dst, n16); break;
            case 3: hipLaunchKernelGGL(membench_kernel<3>, grid, dim3(256), 0, nullptr, (const double2 *)src, (double2 *)dst, n16); break;
            case 4: hipLaunchKernelGGL(membench_kernel<4>, grid, dim3(256), 0, nullptr, (const double2 *)src, (double2 *)dst, n16); break;
            case 5: hipLaunchKernelGGL(membench_kernel<5>, grid, dim3(256), 0, nullptr, (const double2 *)src, (double2 *)dst, n16); break;
            case 6: hipLaunchKernelGGL(membench_kernel<6>, grid, dim3(256), 0, nullptr, (const double2 *)src, (double2 *)dst, n16); break;
            case 7: hipLaunchKernelGGL(membench_kernel<7>, grid, dim3(256), 0, nullptr, (const double2 *)src, (double2 *)dst, n16); break;
            default: hipLaunchKernelGGL(membench_kernel<8>, grid, dim3(256), 0, nullptr, (const double2 *)src, (double2 *)dst, n16); break;
        }
    };
    for (int i = 0; i < 3; ++i) launch();
    MBCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) launch();
    MBCHK(hipEventRecord(e1, nullptr));
    MBCHK(hipEventSynchronize(e1));
    float ms = 0;
    MBCHK(hipEventElapsedTime(&ms, e0, e1));
#undef MBCHK
    *mean_ms = ms / iters;
    cleanup();
    return PCS_OK;
}

int pcs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pcs_create(pcs_engine **out, int chain, int dtype, int64_t n_cams, int64_t n_imgs, int64_t n_keys, int device) {
    if (!out) return fail(PCS_ERR_ARG, "pcs_create: out is NULL");
    *out = nullptr;
    if (chain < 0 || chain > 2) return fail(PCS_ERR_ARG, "pcs_create: chain %d not in {0,1,2}", chain);
    if (dtype != PCS_F64 && dtype != PCS_F32 && dtype != PCS_MIXED) return fail(PCS_ERR_ARG, "pcs_create: dtype %d not in {0,1,2}", dtype);
    if (n_cams <= 0 || n_keys <= 0 || (chain != PCS_CHAIN_FREE && n_imgs <= 0))
        return fail(PCS_ERR_ARG, "pcs_create: counts must be positive (cams %lld imgs %lld keys %lld)", (long long)n_cams,
                    (long long)n_imgs, (long long)n_keys);
    if (n_cams > (1 << 24) || n_imgs > (1 << 24) || n_keys > (1 << 26)) return fail(PCS_ERR_ARG, "pcs_create: counts too large");
    int ndev = pcs_device_count();
    if (ndev <= 0) return fail(PCS_ERR_NODEVICE, "pcs_create: no HIP device visible (this engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(PCS_ERR_ARG, "pcs_create: device %d out of range [0,%d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    pcs_engine *h = new pcs_engine();
    h->chain = chain;
    h->dtype = dtype;
    h->device = device;
    h->P = chain_P(chain);
    h->msize = dtype == PCS_F32 ? 4 : 8;
    h->osize = dtype == PCS_F64 ? 8 : 4;
    h->n_cams = n_cams;
    h->n_imgs = chain == PCS_CHAIN_FREE ? 0 : n_imgs;
    h->n_keys = n_keys;
    h->extr_off = 9 * n_cams;
    h->pose_off = 15 * n_cams;
    h->point_off = chain == PCS_CHAIN_SELF ? 15 * n_cams + 6 * n_imgs : 15 * n_cams;
    h->n_params = chain == PCS_CHAIN_TEMPLATE ? 15 * n_cams + 6 * n_imgs
                  : chain == PCS_CHAIN_SELF   ? 15 * n_cams + 6 * n_imgs + 3 * n_keys
                                              : 15 * n_cams + 3 * n_keys;
#define CREATE_CHK(expr)                                                                           \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            int _rc = fail(PCS_ERR_HIP, "%s failed: %s (pcs_create)", #expr, hipGetErrorString(_e)); \
            pcs_destroy(h); /* releases whatever was allocated so far */                           \
            return _rc;                                                                            \
        }                                                                                          \
    } while (0)
    hipDeviceProp_t prop;
    CREATE_CHK(hipGetDeviceProperties(&prop, device));
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->lds_limit = prop.maxSharedMemoryPerMultiProcessor > 0 ? prop.maxSharedMemoryPerMultiProcessor
                   : prop.sharedMemPerBlock > 0            ? prop.sharedMemPerBlock
                                                           : 64 * 1024;
    CREATE_CHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->ev.assign(4, nullptr);
    h->ev_has_prep.assign(1, 0);
    for (auto &e : h->ev) CREATE_CHK(hipEventCreate(&e));
    CREATE_CHK(hipEventCreateWithFlags(&h->done, hipEventDisableTiming));
    CREATE_CHK(hipMalloc(&h->d_param, sizeof(double) * h->n_params));
    CREATE_CHK(hipHostMalloc(reinterpret_cast<void **>(&h->h_param), sizeof(double) * h->n_params, hipHostMallocDefault));
    CREATE_CHK(hipMalloc(&h->d_cam_slab, sizeof(double) * n_cams * CAM_STRIDE));
    CREATE_CHK(hipMalloc(&h->d_pose_slab, sizeof(double) * std::max<int64_t>(1, h->n_imgs) * POSE_STRIDE));
    CREATE_CHK(hipMalloc(&h->d_points, sizeof(double) * padded_points(n_keys)));
    CREATE_CHK(hipMemset(h->d_points, 0, sizeof(double) * padded_points(n_keys)));
    CREATE_CHK(hipMalloc(&h->d_sink, 64));
#undef CREATE_CHK
    *out = h;
    return PCS_OK;
}

int pcs_destroy(pcs_engine *h) {
    if (!h) return PCS_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    void *bufs[] = {h->d_cam, h->d_img, h->d_key, h->d_packed, h->d_order, h->d_order_ck, h->d_order_ik, h->d_uv, h->d_param, h->d_cam_slab, h->d_pose_slab, h->d_points,
                    h->d_resid, h->d_jac, h->d_keep, h->d_row_off, h->d_data, h->d_vin, h->d_vout, h->d_cost, h->d_im_points, h->d_cam_tab, h->d_sink, h->d_H};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (auto &t : h->d_sorted)
        for (void *b : t)
            if (b) (void)hipFree(b);
    free_det_tables(h);
    if (h->h_param) (void)hipHostFree(h->h_param);
    for (auto &e : h->ev)
        if (e) (void)hipEventDestroy(e);
    if (h->done) (void)hipEventDestroy(h->done);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PCS_OK;
}

int64_t pcs_n_params(const pcs_engine *h) { return h ? h->n_params : -1; }
int pcs_row_len(const pcs_engine *h) { return h ? h->P : -1; }
int64_t pcs_n_detections(const pcs_engine *h) { return h ? h->n : -1; }

static int upload_detections(pcs_engine *h, std::vector<int32_t> &cam, std::vector<int32_t> &img, std::vector<int32_t> &key,
                             const double *uv, int64_t n) {
    // range check: an out-of-range index would be an out-of-bounds slab read on the device.  The
    // engine's own tables are only replaced once the new ones are known to be good.
    for (int64_t i = 0; i < n; ++i) {
        const int32_t c = cam[i], im = img[i], k = key[i];
        if (c < 0 || c >= h->n_cams || k < 0 || k >= h->n_keys || (h->chain != PCS_CHAIN_FREE && (im < 0 || im >= h->n_imgs)))
            return fail(PCS_ERR_RANGE, "detection %lld = (cam %d, im %d, key %d) outside (%lld, %lld, %lld)", (long long)i, c, im, k,
                        (long long)h->n_cams, (long long)h->n_imgs, (long long)h->n_keys);
    }
    h->h_cam.swap(cam); h->h_img.swap(img); h->h_key.swap(key);
    h->n = 0;  // stays 0 (= "no detections set") if an allocation or copy below fails
    {   // slab-read locality of the table, per 64-detection tile (drives the automatic variant choice)
        int64_t tiles = 0, good = 0, runs = 0;
        const bool has_img = h->chain != PCS_CHAIN_FREE;
        for (int64_t t0 = 0; t0 < n; t0 += TILE, ++tiles) {
            const int64_t t1 = std::min<int64_t>(t0 + TILE, n);
            int64_t p0 = -1, p1 = -1, prev = -1;
            bool ok = true;
            for (int64_t i = t0; i < t1; ++i) {
                const int64_t pr = ((int64_t)h->h_cam[i] << 32) | (has_img ? (uint32_t)h->h_img[i] : 0u);
                runs += pr != prev;   // an upper bound of the distinct pairs of the tile; exact for run-ordered tables
                prev = pr;
                if (pr == p0 || pr == p1) continue;
                if (p0 < 0) p0 = pr; else if (p1 < 0) p1 = pr; else ok = false;
            }
            good += ok;
        }
        h->tile_locality = tiles ? (double)good / (double)tiles : 1.0;
        h->tile_segments = tiles ? (double)runs / (double)tiles : 1.0;
    }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(wait_done_host(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->point_orders_tried = false;
    for (void **b : {(void **)&h->d_cam, (void **)&h->d_img, (void **)&h->d_key, (void **)&h->d_packed, (void **)&h->d_order, (void **)&h->d_order_ck,
                     (void **)&h->d_order_ik, &h->d_uv, &h->d_resid, &h->d_jac,
                     (void **)&h->d_keep, (void **)&h->d_row_off, &h->d_data}) {
        if (*b) HIPCHK(hipFree(*b));
        *b = nullptr;
    }
    for (auto &t : h->d_sorted)
        for (void *&b : t) {
            if (b) HIPCHK(hipFree(b));
            b = nullptr;
        }
    h->jac_capacity = h->resid_capacity = h->data_capacity = 0;
    h->nnz = -1;
    free_det_tables(h);
    h->h_order.clear();
    h->h_order_ck.clear();
    if (n == 0) return PCS_OK;
    // index word: cam | image | key bit fields when they fit 32 bits (12 -> 4 bytes per detection), else three arrays
    auto bits_for = [](int64_t count) { int b = 0; while (((int64_t)1 << b) < count) ++b; return b; };
    h->key_bits = bits_for(h->n_keys);
    h->img_bits = h->chain == PCS_CHAIN_FREE ? 0 : bits_for(h->n_imgs);   // the free chain has no image column (its values are not range-checked)
    if (h->pack_indices && h->key_bits + h->img_bits + bits_for(h->n_cams) <= 32 && h->key_bits + h->img_bits <= 31) {
        std::vector<uint32_t> w(n);
        for (int64_t i = 0; i < n; ++i)
            w[i] = ((uint32_t)h->h_cam[i] << (h->key_bits + h->img_bits)) | ((h->img_bits ? (uint32_t)h->h_img[i] : 0u) << h->key_bits) | (uint32_t)h->h_key[i];
        HIPCHK(hipMalloc(&h->d_packed, sizeof(uint32_t) * n));
        HIPCHK(hipMemcpy(h->d_packed, w.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    } else {
        HIPCHK(hipMalloc(&h->d_cam, sizeof(int32_t) * n));
        HIPCHK(hipMalloc(&h->d_img, sizeof(int32_t) * n));
        HIPCHK(hipMalloc(&h->d_key, sizeof(int32_t) * n));
        HIPCHK(hipMemcpy(h->d_cam, h->h_cam.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_img, h->h_img.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_key, h->h_key.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMalloc(&h->d_uv, h->msize * 2 * n));
    if (h->msize == 8) {
        HIPCHK(hipMemcpy(h->d_uv, uv, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    } else {
        std::vector<float> f(2 * n);
        for (int64_t i = 0; i < 2 * n; ++i) f[i] = (float)uv[i];
        HIPCHK(hipMemcpy(h->d_uv, f.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice));
    }
    if (h->tile_locality < 0.5 && n <= INT32_MAX) {
        // scattered table: a (cam, image)-sorted visiting order for the kernels whose result does not depend on
        // the row order (ba_normal_kernel keeps its accumulators per (cam, image) run)
        std::vector<int32_t> order(n);
        for (int64_t i = 0; i < n; ++i) order[i] = (int32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
            return h->h_cam[x] != h->h_cam[y] ? h->h_cam[x] < h->h_cam[y] : h->h_img[x] < h->h_img[y];
        });
        HIPCHK(hipMalloc(&h->d_order, sizeof(int32_t) * n));
        HIPCHK(hipMemcpy(h->d_order, order.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
        h->h_order.swap(order);
    }
    h->n = n;
    return PCS_OK;
}

int pcs_set_detections_table(pcs_engine *h, const double *det5, int64_t n) {
    if (!h || (!det5 && n > 0) || n < 0) return fail(PCS_ERR_ARG, "pcs_set_detections_table: bad arguments");
    std::vector<int32_t> cam(n), img(n), key(n);
    std::vector<double> uv(2 * n);
    for (int64_t i = 0; i < n; ++i) {
        for (int j = 0; j < 3; ++j)  // NaN / huge values have no int32 image: refuse instead of casting
            if (!(det5[5 * i + j] > -1.0 && det5[5 * i + j] < 2147483648.0))
                return fail(PCS_ERR_RANGE, "detection %lld: index column %d = %g is not an index", (long long)i, j, det5[5 * i + j]);
        cam[i] = (int32_t)det5[5 * i + 0];  // int() cast like afb:214 / afb:375
        img[i] = (int32_t)det5[5 * i + 1];
        key[i] = (int32_t)det5[5 * i + 2];
        uv[2 * i] = det5[5 * i + 3];
        uv[2 * i + 1] = det5[5 * i + 4];
    }
    return upload_detections(h, cam, img, key, uv.data(), n);
}

int pcs_set_detections(pcs_engine *h, const int32_t *cam, const int32_t *img, const int32_t *key, const double *uv, int64_t n) {
    if (!h || n < 0 || (n > 0 && (!cam || !key || !uv || (!img && h->chain != PCS_CHAIN_FREE))))
        return fail(PCS_ERR_ARG, "pcs_set_detections: bad arguments");
    std::vector<int32_t> c(cam, cam + n), k(key, key + n), im;
    if (img) im.assign(img, img + n); else im.assign(n, 0);
    return upload_detections(h, c, im, k, uv, n);
}

int pcs_set_template(pcs_engine *h, const double *points) {
    if (!h || !points) return fail(PCS_ERR_ARG, "pcs_set_template: bad arguments");
    if (h->chain != PCS_CHAIN_TEMPLATE) return fail(PCS_ERR_ARG, "pcs_set_template: only the template chain has constant points");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(wait_done_host(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->d_points, points, sizeof(double) * 3 * h->n_keys, hipMemcpyHostToDevice));
    h->have_template = true;
    return PCS_OK;
}

int pcs_set_option(pcs_engine *h, const char *key, int64_t value) {
    if (!h || !key) return fail(PCS_ERR_ARG, "pcs_set_option: bad arguments");
    if (!strcmp(key, "variant")) {
        if (value < -1 || value > 7) return fail(PCS_ERR_ARG, "variant must be in [-1,7] (-1 = automatic)");
        h->variant = (int)value;
    } else if (!strcmp(key, "wgs_per_cu")) {
        if (value < 0 || value > 64) return fail(PCS_ERR_ARG, "wgs_per_cu must be in [0,64] (0 = automatic)");
        h->wgs_per_cu = value;
        h->tiles_per_wg = 0;
    } else if (!strcmp(key, "timing_every")) {
        if (value < 0 || value > 1000000) return fail(PCS_ERR_ARG, "timing_every must be in [0,1000000]");
        h->timing_every = value;
        h->eval_count = 0;
    } else if (!strcmp(key, "fuse_prep")) {
        if (value < -1 || value > 1) return fail(PCS_ERR_ARG, "fuse_prep must be -1 (automatic), 0 (slab_prep launch) or 1 (waves prepare their slabs)");
        h->fuse_prep = (int)value;
    } else if (!strcmp(key, "fuse_prep_max_n")) {
        if (value < 0) return fail(PCS_ERR_ARG, "fuse_prep_max_n must be >= 0");
        h->fuse_prep_max_n = value;
    } else if (!strcmp(key, "lazy_done_event")) {
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(flush_done(h));
        if (value < 0 || value > 2) return fail(PCS_ERR_ARG, "lazy_done_event must be 0, 1 or 2");
        h->lazy_done = value != 0;
        h->lazy_any_stream = value == 2;
    } else if (!strcmp(key, "xcd_remap")) {
        h->xcd_remap = value != 0;
    } else if (!strcmp(key, "waves_per_wg")) {
        if (value < 0 || value > WAVES_PER_WG || value == 3) return fail(PCS_ERR_ARG, "waves_per_wg must be 0 (automatic), 1, 2 or 4");
        h->waves_per_wg = (int)value;
    } else if (!strcmp(key, "pack_indices")) {
        h->pack_indices = value != 0;   // takes effect at the next pcs_set_detections*
    } else if (!strcmp(key, "normal_rows")) {
        if (value != 32 && value != 64) return fail(PCS_ERR_ARG, "normal_rows must be 32 or 64");
        h->normal_rows = (int)value;
    } else if (!strcmp(key, "matfree_lds")) {
        h->matfree_lds = value != 0;
    } else if (!strcmp(key, "normal_debug")) {
        h->normal_debug = (int)value;
    } else if (!strcmp(key, "normal_sort_tables")) {
        h->sort_tables = value != 0;
    } else if (!strcmp(key, "normal_imgkey_product")) {
        h->normal_imgkey_product = value != 0;
    } else if (!strcmp(key, "normal_imgkey_wgs_per_cu")) {
        h->normal_imgkey_wgs_per_cu = (int)value;
    } else if (!strcmp(key, "compact_variant")) {
        if (value < 0 || value > 1) return fail(PCS_ERR_ARG, "compact_variant must be 0 or 1");
        h->compact_variant = (int)value;
    } else if (!strcmp(key, "event_ring")) {
        // keep the HIP-event triples of the last `value` evaluations (pcs_kernel_ms_mean averages them)
        if (value < 1 || value > 100000) return fail(PCS_ERR_ARG, "event_ring must be in [1,100000]");
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(wait_done_host(h));
        for (auto &e : h->ev)
            if (e) (void)hipEventDestroy(e);
        h->ev.assign(4 * value, nullptr);
        h->ev_has_prep.assign(value, 0);
        for (auto &e : h->ev) HIPCHK(hipEventCreate(&e));
        h->ev_ring = value;
        h->ev_count = 0;
        h->events_valid = false;
    } else if (!strcmp(key, "fused_trial")) {
        h->fused_trial = value != 0;
    } else if (!strcmp(key, "deterministic")) {
        if (value < 0 || value > 1) return fail(PCS_ERR_ARG, "deterministic must be 0 or 1");
        h->deterministic = (int)value;
    } else if (!strcmp(key, "spd_timeout_us")) {
        // how long a workgroup of the one-launch dense solve waits for a hand-over before it abandons the launch (status bit 2, LM stop
        // code 9: the trial is repeated with the launch-per-column form).  Tests set it to 1 us to force that path.
        if (value < 1 || value > 60000000) return fail(PCS_ERR_ARG, "spd_timeout_us must be in [1, 60000000]");
        h->spd_timeout_us = value;
    } else if (!strcmp(key, "tiles_per_wg")) {
        if (value < 0 || value > (1 << 20)) return fail(PCS_ERR_ARG, "tiles_per_wg out of range");
        h->tiles_per_wg = value;
    } else {
        return fail(PCS_ERR_ARG, "pcs_set_option: unknown key '%s'", key);
    }
    return PCS_OK;
}

}  // extern "C"

// ---- launch plumbing ---------------------------------------------------------------------------
// Kernel timing uses the start/stop events of hipExtLaunchKernelGGL: the timestamps are taken by the
// dispatch itself, without the extra barrier packets that hipEventRecord would put between kernels.
struct EvPair { hipEvent_t start, stop; };

static DetTable det_table(const pcs_engine *h) {
    DetTable t{};
    t.packed = h->d_packed;
    t.cam = h->d_cam; t.img = h->d_img; t.key = h->d_key;
    t.uv = h->d_uv;
    t.key_bits = h->key_bits; t.img_bits = h->img_bits;
    t.uv_f32 = h->msize == 4;
    return t;
}

template <int CHAIN, int MODE, int VARIANT, typename TO, bool PREP = false>
static hipError_t launch_eval_v(const EvalArgs &a, dim3 grid, int threads, size_t lds, hipStream_t s, EvPair ev) {
    auto kern = ba_eval_kernel<CHAIN, MODE, VARIANT, TO, PREP>;
    // per device: largest dynamic-LDS size already enabled for this kernel.  Handles driven from different host threads
    // may get here together: the value is monotone and setting the attribute twice is harmless, so an atomic max is enough
    static std::atomic<size_t> configured[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<size_t> &cfg = configured[dev & 63];
    if (lds > 48 * 1024 && lds > cfg.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        size_t seen = cfg.load(std::memory_order_relaxed);
        while (seen < lds && !cfg.compare_exchange_weak(seen, lds, std::memory_order_release, std::memory_order_relaxed)) {}
    }
    hipExtLaunchKernelGGL(kern, grid, dim3(threads), (std::uint32_t)lds, s, ev.start, ev.stop, 0, a);
    return hipGetLastError();
}

template <int CHAIN, int MODE, typename TO>
static hipError_t launch_eval_m(int variant, bool prep, const EvalArgs &a, dim3 grid, int threads, size_t lds, hipStream_t s, EvPair ev) {
    if (prep) {   // one-launch step: slabs through L1/L2 is the only form it replaces; transposed stores whenever there is a Jacobian
        if constexpr (MODE == MODE_RESID) {
            return (variant & VAR_NT) ? launch_eval_v<CHAIN, MODE, VAR_NT, TO, true>(a, grid, threads, lds, s, ev)
                                      : launch_eval_v<CHAIN, MODE, 0, TO, true>(a, grid, threads, lds, s, ev);
        } else {
            return (variant & VAR_NT) ? launch_eval_v<CHAIN, MODE, VAR_TRANSPOSE | VAR_NT, TO, true>(a, grid, threads, lds, s, ev)
                                      : launch_eval_v<CHAIN, MODE, VAR_TRANSPOSE, TO, true>(a, grid, threads, lds, s, ev);
        }
    }
    switch (variant) {
        case 0: return launch_eval_v<CHAIN, MODE, 0, TO>(a, grid, threads, lds, s, ev);
        case 1: return launch_eval_v<CHAIN, MODE, 1, TO>(a, grid, threads, lds, s, ev);
        case 2: return launch_eval_v<CHAIN, MODE, 2, TO>(a, grid, threads, lds, s, ev);
        case 3: return launch_eval_v<CHAIN, MODE, 3, TO>(a, grid, threads, lds, s, ev);
        case 4: return launch_eval_v<CHAIN, MODE, 4, TO>(a, grid, threads, lds, s, ev);
        case 5: return launch_eval_v<CHAIN, MODE, 5, TO>(a, grid, threads, lds, s, ev);
        case 6: return launch_eval_v<CHAIN, MODE, 6, TO>(a, grid, threads, lds, s, ev);
        default: return launch_eval_v<CHAIN, MODE, 7, TO>(a, grid, threads, lds, s, ev);
    }
}

template <int CHAIN, typename TO>
static hipError_t launch_eval_c(int mode, int variant, bool prep, const EvalArgs &a, dim3 grid, int threads, size_t lds, hipStream_t s, EvPair ev) {
    switch (mode) {
        case MODE_RESID: return launch_eval_m<CHAIN, MODE_RESID, TO>(variant & ~VAR_TRANSPOSE, prep, a, grid, threads, lds, s, ev);
        case MODE_JAC: return launch_eval_m<CHAIN, MODE_JAC, TO>(variant, prep, a, grid, threads, lds, s, ev);
        default: return launch_eval_m<CHAIN, MODE_RESID | MODE_JAC, TO>(variant, prep, a, grid, threads, lds, s, ev);
    }
}

template <typename TO>
static hipError_t launch_eval_t(int chain, int mode, int variant, bool prep, const EvalArgs &a, dim3 grid, int threads, size_t lds, hipStream_t s, EvPair ev) {
    switch (chain) {
        case CHAIN_TEMPLATE: return launch_eval_c<CHAIN_TEMPLATE, TO>(mode, variant, prep, a, grid, threads, lds, s, ev);
        case CHAIN_SELF: return launch_eval_c<CHAIN_SELF, TO>(mode, variant, prep, a, grid, threads, lds, s, ev);
        default: return launch_eval_c<CHAIN_FREE, TO>(mode, variant, prep, a, grid, threads, lds, s, ev);
    }
}

template <int CHAIN, typename TO>
static hipError_t launch_compact_tile_c(int mode, const EvalArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    if (mode == MODE_RESID) hipLaunchKernelGGL((ba_compact_tile_kernel<CHAIN, MODE_RESID, TO>), grid, dim3(WG_THREADS), lds, s, a);
    else if (mode == MODE_JAC) hipLaunchKernelGGL((ba_compact_tile_kernel<CHAIN, MODE_JAC, TO>), grid, dim3(WG_THREADS), lds, s, a);
    else hipLaunchKernelGGL((ba_compact_tile_kernel<CHAIN, MODE_RESID | MODE_JAC, TO>), grid, dim3(WG_THREADS), lds, s, a);
    return hipGetLastError();
}

template <typename TO>
static hipError_t launch_compact_tile_t(int chain, int mode, const EvalArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    switch (chain) {
        case CHAIN_TEMPLATE: return launch_compact_tile_c<CHAIN_TEMPLATE, TO>(mode, a, grid, lds, s);
        case CHAIN_SELF: return launch_compact_tile_c<CHAIN_SELF, TO>(mode, a, grid, lds, s);
        default: return launch_compact_tile_c<CHAIN_FREE, TO>(mode, a, grid, lds, s);
    }
}

template <int CHAIN>
static hipError_t launch_compact_c(int mode, const EvalArgs &a, dim3 grid, hipStream_t s) {
    if (mode == MODE_RESID) hipLaunchKernelGGL((ba_compact_kernel<CHAIN, MODE_RESID>), grid, dim3(WG_THREADS), 0, s, a);
    else if (mode == MODE_JAC) hipLaunchKernelGGL((ba_compact_kernel<CHAIN, MODE_JAC>), grid, dim3(WG_THREADS), 0, s, a);
    else hipLaunchKernelGGL((ba_compact_kernel<CHAIN, MODE_RESID | MODE_JAC>), grid, dim3(WG_THREADS), 0, s, a);
    return hipGetLastError();
}

static hipError_t launch_compact_t(int chain, int mode, const EvalArgs &a, dim3 grid, hipStream_t s) {
    switch (chain) {
        case CHAIN_TEMPLATE: return launch_compact_c<CHAIN_TEMPLATE>(mode, a, grid, s);
        case CHAIN_SELF: return launch_compact_c<CHAIN_SELF>(mode, a, grid, s);
        default: return launch_compact_c<CHAIN_FREE>(mode, a, grid, s);
    }
}

static int launch_slab_prep(pcs_engine *h, const double *d_prm, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr) {
    HIPCHK(order_after_done(h, s));   // the slabs are shared: an evaluation still reading them on another stream goes first
    const int has_pose = h->chain != PCS_CHAIN_FREE;
    const int copy_points = h->chain != PCS_CHAIN_TEMPLATE;
    int64_t threads = slab_prep_threads(h->n_cams, h->n_imgs, has_pose);   // one per slab element
    if (copy_points) threads = std::max<int64_t>(threads, std::min<int64_t>(3 * h->n_keys, 1 << 16));
    const dim3 grid((unsigned)((threads + 63) / 64));
    hipExtLaunchKernelGGL(slab_prep_kernel, grid, dim3(64), 0, s, start, stop, 0, d_prm, (double *)h->d_cam_slab,
                          (double *)h->d_pose_slab, (double *)h->d_points, (int)h->n_cams, (int)h->n_imgs, (int)h->n_keys,
                          h->extr_off, h->pose_off, h->point_off, has_pose, copy_points);
    HIPCHK(hipGetLastError());
    h->linearized = true;
    return PCS_OK;
}

template <int CHAIN, bool LDS_ACC>
static hipError_t launch_matfree_c(int op, const MatfreeArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    if (lds > 48 * 1024) {  // opt in to more than the default dynamic-LDS cap
        const void *fns[] = {(const void *)ba_matfree_kernel<CHAIN, OP_JTU, LDS_ACC>, (const void *)ba_matfree_kernel<CHAIN, OP_JTJV, LDS_ACC>,
                             (const void *)ba_matfree_kernel<CHAIN, OP_DIAG, LDS_ACC>, (const void *)ba_matfree_kernel<CHAIN, OP_GRAD, LDS_ACC>};
        hipError_t e = hipFuncSetAttribute(fns[op - 1], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    switch (op) {
        case OP_JV: hipLaunchKernelGGL((ba_matfree_kernel<CHAIN, OP_JV, false>), grid, dim3(256), 0, s, a); break;
        case OP_JTU: hipLaunchKernelGGL((ba_matfree_kernel<CHAIN, OP_JTU, LDS_ACC>), grid, dim3(256), lds, s, a); break;
        case OP_JTJV: hipLaunchKernelGGL((ba_matfree_kernel<CHAIN, OP_JTJV, LDS_ACC>), grid, dim3(256), lds, s, a); break;
        case OP_DIAG: hipLaunchKernelGGL((ba_matfree_kernel<CHAIN, OP_DIAG, LDS_ACC>), grid, dim3(256), lds, s, a); break;
        default: hipLaunchKernelGGL((ba_matfree_kernel<CHAIN, OP_GRAD, LDS_ACC>), grid, dim3(256), lds, s, a); break;
    }
    return hipGetLastError();
}

static hipError_t launch_matfree_t(int chain, int op, bool lds_acc, const MatfreeArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    if (lds_acc) {
        switch (chain) {
            case CHAIN_TEMPLATE: return launch_matfree_c<CHAIN_TEMPLATE, true>(op, a, grid, lds, s);
            case CHAIN_SELF: return launch_matfree_c<CHAIN_SELF, true>(op, a, grid, lds, s);
            default: return launch_matfree_c<CHAIN_FREE, true>(op, a, grid, lds, s);
        }
    }
    switch (chain) {
        case CHAIN_TEMPLATE: return launch_matfree_c<CHAIN_TEMPLATE, false>(op, a, grid, 0, s);
        case CHAIN_SELF: return launch_matfree_c<CHAIN_SELF, false>(op, a, grid, 0, s);
        default: return launch_matfree_c<CHAIN_FREE, false>(op, a, grid, 0, s);
    }
}

// one pass of the normal-equations kernel (ba_normal.hpp); the workgroup is one wave
template <int CHAIN, int PASS>
static hipError_t launch_normal_p(int rows, const NormalArgs &a, dim3 grid, hipStream_t s) {
    if (rows == 32) hipLaunchKernelGGL((ba_normal_mfma_kernel<CHAIN, PASS, 32>), grid, dim3(64), normal_lds_bytes(CHAIN, PASS, 32), s, a);
    else hipLaunchKernelGGL((ba_normal_mfma_kernel<CHAIN, PASS, 64>), grid, dim3(64), normal_lds_bytes(CHAIN, PASS, 64), s, a);
    return hipGetLastError();
}

static hipError_t launch_normal(int chain, int pass, int rows, const NormalArgs &a, dim3 grid, hipStream_t s) {
    if (pass == PASS_SHARED) {
        switch (chain) {
            case CHAIN_TEMPLATE: return launch_normal_p<CHAIN_TEMPLATE, PASS_SHARED>(rows, a, grid, s);
            case CHAIN_SELF: return launch_normal_p<CHAIN_SELF, PASS_SHARED>(rows, a, grid, s);
            default: return launch_normal_p<CHAIN_FREE, PASS_SHARED>(rows, a, grid, s);
        }
    }
    if (pass == PASS_CAMKEY)
        return chain == CHAIN_SELF ? launch_normal_p<CHAIN_SELF, PASS_CAMKEY>(rows, a, grid, s) : launch_normal_p<CHAIN_FREE, PASS_CAMKEY>(rows, a, grid, s);
    return launch_normal_p<CHAIN_SELF, PASS_IMGKEY>(rows, a, grid, s);
}

// dst[i] = src[order[i]] (one-off, at the first normal-equations call)
template <typename E>
__global__ void gather_rows_kernel(const int32_t *__restrict__ order, const E *__restrict__ src, E *__restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[order[i]];
}
template <typename E>
static hipError_t gather_rows(const int32_t *order, const void *src, void **dst, int64_t n, hipStream_t s) {
    hipError_t e = hipMalloc(dst, sizeof(E) * n);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gather_rows_kernel<E>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, order, static_cast<const E *>(src), static_cast<E *>(*dst), n);
    return hipGetLastError();
}

// (cam, key)- and (image, key)-sorted visiting orders for the point passes of the normal equations: built on the host
// at the first normal-equations call of a self / free engine (two stable sorts, ~0.1 s at 1e6 detections)
static int build_point_orders(pcs_engine *h);
// all or nothing: a build that fails half-way (an allocation, a gather launch) leaves no sorted copy behind, so that the
// next call starts again instead of pairing sorted index words with unsorted measurements
static int ensure_point_orders(pcs_engine *h) {
    if (h->point_orders_tried) return PCS_OK;
    const int rc = build_point_orders(h);
    if (rc == PCS_OK) {
        h->point_orders_tried = true;
        return PCS_OK;
    }
    const std::string keep = g_err;
    (void)hipStreamSynchronize(h->stream);
    for (int32_t **o : {&h->d_order_ck, &h->d_order_ik}) {
        if (*o) (void)hipFree(*o);
        *o = nullptr;
    }
    for (auto &t : h->d_sorted)
        for (void *&b : t) {
            if (b) (void)hipFree(b);
            b = nullptr;
        }
    g_err = keep;
    return rc;
}
static int build_point_orders(pcs_engine *h) {
    const int64_t n = h->n;
    if (n <= 0) return PCS_OK;
    if (h->chain != PCS_CHAIN_TEMPLATE) {
        std::vector<int32_t> order(n);
        for (int pass = 0; pass < (h->chain == PCS_CHAIN_SELF ? 2 : 1); ++pass) {
            const std::vector<int32_t> &major = pass == 0 ? h->h_cam : h->h_img;
            for (int64_t i = 0; i < n; ++i) order[i] = (int32_t)i;
            std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
                return major[x] != major[y] ? major[x] < major[y] : h->h_key[x] < h->h_key[y];
            });
            int32_t **dst = pass == 0 ? &h->d_order_ck : &h->d_order_ik;
            HIPCHK(hipMalloc(dst, sizeof(int32_t) * n));
            HIPCHK(hipMemcpy(*dst, order.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
            if (pass == 0) h->h_order_ck = order;
        }
    }
    // the table in each visiting order: the shared pass of a scattered table, the (cam, key) pass, the (image, key) pass (which
    // reads no measurements)
    const int32_t *orders[3] = {h->d_order, h->d_order_ck, h->d_order_ik};
    using u2 = __attribute__((ext_vector_type(2))) uint32_t;
    using u4 = __attribute__((ext_vector_type(4))) uint32_t;
    for (int pass = 0; pass < 3; ++pass) {
        if (!orders[pass]) continue;
        void **t = h->d_sorted[pass];
        if (h->d_packed) {
            HIPCHK(gather_rows<uint32_t>(orders[pass], h->d_packed, &t[0], n, h->stream));
        } else {
            HIPCHK(gather_rows<uint32_t>(orders[pass], h->d_cam, &t[1], n, h->stream));
            HIPCHK(gather_rows<uint32_t>(orders[pass], h->d_img, &t[2], n, h->stream));
            HIPCHK(gather_rows<uint32_t>(orders[pass], h->d_key, &t[3], n, h->stream));
        }
        if (pass != PASS_IMGKEY) HIPCHK(h->msize == 4 ? gather_rows<u2>(orders[pass], h->d_uv, &t[4], n, h->stream) : gather_rows<u4>(orders[pass], h->d_uv, &t[4], n, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return PCS_OK;
}

// ---- deterministic mode: the static tables of the ordered second pass (csrc/ba_reduce.hpp) ---------------------------------------------
// For MFMA pass `pass` (0 shared, 1 (cam, key)) walked with `tpw` tiles per wave: segments = maximal stretches of detections inside one
// run and one wave; logical runs = the segments of one key pair (one stretch in a sorted table; several when a pair re-appears);
// groups = up to RED_RUNS_PER_GROUP logical runs of one camera; per camera its groups, per entity (image / key) its runs in camera order.
static int ensure_det_tables(pcs_engine *h, int pass, int64_t tpw) {
    pcs_engine::DetPass &D = h->det[pass];
    if (D.n == h->n && D.tpw == tpw && D.d_idx) return PCS_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(wait_done_host(h));
    if (D.d_idx) (void)hipFree(D.d_idx);
    if (D.d_work) (void)hipFree(D.d_work);
    D = pcs_engine::DetPass{};
    const int64_t n = h->n;
    const bool has_pose = h->chain != PCS_CHAIN_FREE;
    const std::vector<int32_t> *order = nullptr;
    if (pass == PASS_SHARED) { if (h->d_order) order = &h->h_order; }
    else order = &h->h_order_ck;
    if (order && (int64_t)order->size() != n) return fail(PCS_ERR_STATE, "deterministic mode: the host copy of a visiting order is missing");
    auto key_of = [&](int64_t d, int32_t &ka, int32_t &kb) {
        const int64_t i = order ? (*order)[d] : d;
        ka = h->h_cam[i];
        kb = pass == PASS_SHARED ? (has_pose ? h->h_img[i] : 0) : h->h_key[i];
    };
    const int64_t per_wave = tpw * TILE;
    const int64_t n_waves = (n + per_wave - 1) / per_wave;
    std::vector<int32_t> seg_base((size_t)n_waves), seg_ka, seg_kb;
    {
        int32_t pa = -1, pb = -1;
        for (int64_t d = 0; d < n; ++d) {
            int32_t ka, kb;
            key_of(d, ka, kb);
            const bool wave_start = d % per_wave == 0;
            if (wave_start) seg_base[(size_t)(d / per_wave)] = (int32_t)seg_ka.size();
            if (wave_start || ka != pa || kb != pb) {
                if (seg_ka.size() >= (size_t)INT32_MAX - 1) return fail(PCS_ERR_ARG, "deterministic mode: too many segments");
                seg_ka.push_back(ka);
                seg_kb.push_back(kb);
            }
            pa = ka;
            pb = kb;
        }
    }
    const int64_t n_seg = (int64_t)seg_ka.size();
    // logical runs: segments sorted by (ka, kb), table order inside one pair
    std::vector<int32_t> lr_segs((size_t)n_seg);
    for (int64_t i = 0; i < n_seg; ++i) lr_segs[(size_t)i] = (int32_t)i;
    std::stable_sort(lr_segs.begin(), lr_segs.end(), [&](int32_t x, int32_t y) {
        return seg_ka[x] != seg_ka[y] ? seg_ka[x] < seg_ka[y] : seg_kb[x] < seg_kb[y];
    });
    // the free chain's shared pass has one run per camera and owns camera-level entries only: a long run may be cut into pieces
    const bool may_split = pass == PASS_SHARED && !has_pose;
    std::vector<int32_t> lr_ptr, lr_ka, lr_kb;
    for (int64_t i = 0; i < n_seg; ++i) {
        const int32_t sgm = lr_segs[(size_t)i];
        const bool fresh = i == 0 || seg_ka[sgm] != lr_ka.back() || seg_kb[sgm] != lr_kb.back() ||
                           (may_split && i - lr_ptr.back() >= RED_SPLIT_SEGS);
        if (fresh) {
            lr_ptr.push_back((int32_t)i);
            lr_ka.push_back(seg_ka[sgm]);
            lr_kb.push_back(seg_kb[sgm]);
        }
    }
    const int64_t n_lr = (int64_t)lr_ka.size();
    lr_ptr.push_back((int32_t)n_seg);
    std::vector<int32_t> grp_ptr, cam_ptr((size_t)h->n_cams + 1, 0);
    for (int64_t r = 0; r < n_lr; ++r) {
        const bool fresh = r == 0 || lr_ka[(size_t)r] != lr_ka[(size_t)r - 1] || r - grp_ptr.back() >= RED_RUNS_PER_GROUP;
        if (fresh) {
            grp_ptr.push_back((int32_t)r);
            ++cam_ptr[(size_t)lr_ka[(size_t)r] + 1];
        }
    }
    const int64_t n_grp = (int64_t)grp_ptr.size();
    grp_ptr.push_back((int32_t)n_lr);
    for (int64_t c = 0; c < h->n_cams; ++c) cam_ptr[(size_t)c + 1] += cam_ptr[(size_t)c];
    const bool has_ent = pass == PASS_CAMKEY || has_pose;
    const int64_t n_ent = !has_ent ? 0 : pass == PASS_SHARED ? h->n_imgs : h->n_keys;
    std::vector<int32_t> ent_ptr((size_t)n_ent + 1, 0), ent_runs(has_ent ? (size_t)n_lr : 0);
    if (has_ent) {
        for (int64_t r = 0; r < n_lr; ++r) ++ent_ptr[(size_t)lr_kb[(size_t)r] + 1];
        for (int64_t e = 0; e < n_ent; ++e) ent_ptr[(size_t)e + 1] += ent_ptr[(size_t)e];
        std::vector<int32_t> fill(ent_ptr.begin(), ent_ptr.end() - 1);
        for (int64_t r = 0; r < n_lr; ++r) ent_runs[(size_t)fill[(size_t)lr_kb[(size_t)r]]++] = (int32_t)r;   // runs are sorted by camera: camera order per entity
    }
    const std::vector<int32_t> *parts[9] = {&seg_base, &lr_ptr, &lr_segs, &lr_ka, &lr_kb, &grp_ptr, &cam_ptr, &ent_ptr, &ent_runs};
    int64_t total = 0;
    for (int i = 0; i < 9; ++i) {
        D.off[i] = total;
        total += ((int64_t)parts[i]->size() + 3) & ~(int64_t)3;
    }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMalloc(&D.d_idx, sizeof(int32_t) * std::max<int64_t>(total, 4)));
    for (int i = 0; i < 9; ++i)
        if (!parts[i]->empty()) HIPCHK(hipMemcpy(D.d_idx + D.off[i], parts[i]->data(), sizeof(int32_t) * parts[i]->size(), hipMemcpyHostToDevice));
    const int nm = (pass == PASS_SHARED && has_pose) ? 2 : 1;
    D.work_off[0] = 0;
    D.work_off[1] = n_seg * nm * 256;
    D.work_off[2] = D.work_off[1] + n_lr * RED_Q;
    HIPCHK(hipMalloc(&D.d_work, sizeof(double) * (size_t)(D.work_off[2] + n_grp * RED_G + 2)));
    D.n = n; D.tpw = tpw;
    D.n_seg = (int32_t)n_seg; D.n_lr = (int32_t)n_lr; D.n_grp = (int32_t)n_grp; D.n_ent = (int32_t)n_ent; D.n_waves = (int32_t)n_waves;
    return PCS_OK;
}

template <int CHAIN, int PASS>
static hipError_t launch_reduce_p(const NormalArgs &a, const ReduceArgs &ra, hipStream_t s) {
    static_assert(RED_RUNS_PER_GROUP == 4, "one wave per run, four waves per workgroup");
    if (ra.n_grp > 0) hipLaunchKernelGGL((normal_reduce_runs_kernel<CHAIN, PASS>), dim3((unsigned)ra.n_grp), dim3(256), 0, s, a, ra);
    const int blocks = (PASS == PASS_SHARED ? ra.n_cams + 1 : 0) + (ra.n_ent + 2) / 3;
    if (blocks > 0) hipLaunchKernelGGL((normal_reduce_final_kernel<CHAIN, PASS>), dim3((unsigned)blocks), dim3(256), 0, s, a, ra);
    return hipGetLastError();
}
static hipError_t launch_reduce(int chain, int pass, const NormalArgs &a, const ReduceArgs &ra, hipStream_t s) {
    if (pass == PASS_SHARED) {
        switch (chain) {
            case CHAIN_TEMPLATE: return launch_reduce_p<CHAIN_TEMPLATE, PASS_SHARED>(a, ra, s);
            case CHAIN_SELF: return launch_reduce_p<CHAIN_SELF, PASS_SHARED>(a, ra, s);
            default: return launch_reduce_p<CHAIN_FREE, PASS_SHARED>(a, ra, s);
        }
    }
    return chain == CHAIN_SELF ? launch_reduce_p<CHAIN_SELF, PASS_CAMKEY>(a, ra, s) : launch_reduce_p<CHAIN_FREE, PASS_CAMKEY>(a, ra, s);
}

// The blocked layout of J^T J (NormalArgs, ba_normal.hpp): where the parameter string splits into leading part and trailing group.
struct BlockLayout {
    int64_t n_lead, n_trail, n_ent, trail_off;
    int tb, tg;
    int64_t a_len() const { return n_lead * n_lead; }
    int64_t b_len() const { return n_lead * n_trail; }
    int64_t c_len() const { return n_ent * tb * tb; }
};
static BlockLayout block_layout(const pcs_engine *h) {
    BlockLayout L{};
    if (h->chain == PCS_CHAIN_TEMPLATE) { L.tg = 2; L.tb = 6; L.trail_off = h->pose_off; L.n_ent = h->n_imgs; }
    else { L.tg = 3; L.tb = 3; L.trail_off = h->point_off; L.n_ent = h->n_keys; }
    L.n_lead = L.trail_off;
    L.n_trail = L.n_ent * L.tb;
    return L;
}

// slab_prep + the normal-equations passes on `s`; d_prm holds the parameter string; outputs are zeroed here.
// blocked: d_H points at the packed [A | B | C] (contiguous), see pcs_normal_blocks_device.
// d_sel (LM loop with two states, ba_schur.hpp SchurArgs::sel): when *d_sel != 0 the string is read alt_prm doubles and the outputs are written
// alt_out doubles further on.
static int enqueue_normal(pcs_engine *h, const double *d_prm, double *d_H, double *d_g, double *d_cost, hipStream_t s, bool blocked = false, const int32_t *d_stop = nullptr,
                          const int32_t *d_sel = nullptr, int64_t alt_prm = 0, int64_t alt_out = 0, bool skip_prologue = false) {
    if (h->n <= 0) return fail(PCS_ERR_STATE, "no detections set");
    if (h->n > INT32_MAX) return fail(PCS_ERR_ARG, "normal equations: tables beyond 2^31 rows are not supported (visiting orders are int32)");
    if (h->chain == PCS_CHAIN_TEMPLATE && !h->have_template) return fail(PCS_ERR_STATE, "template points not set");
    const BlockLayout L = block_layout(h);
    if (!blocked) {
        // the flush addresses H with 32-bit offsets in doubles: n^2 < 2^32 (a 34 GB matrix)
        if (h->n_params > PCS_NORMAL_MAX_PARAMS) return fail(PCS_ERR_ARG, "normal equations: more than %d parameters (dense H beyond 32 GiB) is not supported", PCS_NORMAL_MAX_PARAMS);
    } else {
        const int64_t lim = (int64_t)1 << 32;   // doubles per region: 32-bit offsets in doubles (32 GiB; rounds 3-4: byte offsets, 4 GiB)
        if (L.a_len() >= lim || L.b_len() >= lim || L.c_len() >= lim)
            return fail(PCS_ERR_ARG, "blocked normal equations: a region beyond 32 GiB (leading %lld, trailing %lld columns) is not supported; use pcs_matfree",
                        (long long)L.n_lead, (long long)L.n_trail);
    }
    HIPCHK(hipSetDevice(h->device));
    int rc0 = ensure_point_orders(h);
    if (rc0) return rc0;
    if ((d_stop || d_sel) && (reinterpret_cast<uintptr_t>(d_H) % 16 || alt_out % 2)) return fail(PCS_ERR_ARG, "normal equations behind a stop flag / state selector need 16-byte aligned buffers");
    if (skip_prologue) {   // the caller's previous kernel has prepared the slabs and zeroed the outputs (schur_finish_kernel, csrc/ba_lm_fused.hpp)
        HIPCHK(order_after_done(h, s));
        h->linearized = true;
    } else if (reinterpret_cast<uintptr_t>(d_H) % 16 == 0) {   // slab_prep and the zeroing of the outputs in one launch
        HIPCHK(order_after_done(h, s));
        const int has_pose = h->chain != PCS_CHAIN_FREE, copy_points = h->chain != PCS_CHAIN_TEMPLATE;
        int64_t threads = slab_prep_threads(h->n_cams, h->n_imgs, has_pose);
        if (copy_points) threads = std::max<int64_t>(threads, std::min<int64_t>(3 * h->n_keys, 1 << 16));
        const int prep_blocks = (int)((threads + 63) / 64);
        const int64_t n_h = blocked ? L.a_len() + L.b_len() + L.c_len() : h->n_params * h->n_params;
        const int zero_blocks = (int)std::min<int64_t>((n_h / 2 + 63) / 64 + 1, (int64_t)h->n_cu * 32);
        hipLaunchKernelGGL(normal_prologue_kernel, dim3((unsigned)(prep_blocks + zero_blocks)), dim3(64), 0, s, d_prm, (double *)h->d_cam_slab,
                           (double *)h->d_pose_slab, (double *)h->d_points, (int)h->n_cams, (int)h->n_imgs, (int)h->n_keys, h->extr_off,
                           h->pose_off, h->point_off, has_pose, copy_points, prep_blocks, d_H, n_h, d_g, h->n_params, d_cost, d_stop, d_sel, alt_prm, alt_out);
        HIPCHK(hipGetLastError());
        h->linearized = true;
    } else {
        HIPCHK(hipMemsetAsync(d_H, 0, sizeof(double) * (blocked ? L.a_len() + L.b_len() + L.c_len() : h->n_params * h->n_params), s));
        HIPCHK(hipMemsetAsync(d_g, 0, sizeof(double) * h->n_params, s));
        HIPCHK(hipMemsetAsync(d_cost, 0, sizeof(double), s));
        int rc = launch_slab_prep(h, d_prm, s);
        if (rc) return rc;
    }
    NormalArgs a{};
    a.tab = det_table(h);
    a.cam_slab = h->d_cam_slab; a.pose_slab = h->d_pose_slab; a.points = h->d_points;
    a.H = d_H; a.g = d_g; a.cost = d_cost;
    if (blocked) {
        a.HB = d_H + L.a_len(); a.HC = a.HB + L.b_len();
        a.ldA = (int32_t)L.n_lead; a.ldB = (int32_t)L.n_trail; a.tb = L.tb; a.trail_group = L.tg; a.trail_off = L.trail_off;
    } else {
        a.HB = a.HC = d_H;
        a.ldA = (int32_t)h->n_params; a.ldB = 0; a.tb = 0; a.trail_group = -1; a.trail_off = 0;
    }
    a.n = h->n; a.n_tiles = (h->n + TILE - 1) / TILE;
    a.extr_off = h->extr_off; a.pose_off = h->pose_off; a.point_off = h->point_off;
    a.n_params = h->n_params;
    a.debug = h->normal_debug;
    a.stop = d_stop;
    a.sel = d_sel; a.alt = alt_out;
    // every wave walks a contiguous range of tiles, so its register accumulators survive across tiles.  One-wave
    // workgroups; the LDS image (22.9 KB for 22 columns x 64 rows) allows 7 per CU, and exactly one resident round of
    // waves is fastest (92 us against 105 us with two rounds on rig-32, profiles/r02/sweeps.md).
    // Tables of 1 024 - 1 792 tiles: two tiles per wave instead of one.  A wave flushes everything it holds when it ends, and with
    // one tile per wave every tile pays the camera-level flush into addresses that hundreds of waves share (ring-8, 1 600 tiles,
    // 8 cameras: 35.0 us at one tile per wave, 27.7 at two, 28.4 at three).  Smaller tables keep one tile per wave: there the
    // second tile's latency costs more than the flushes (110 tiles: 14.4 against 20.4 us).
    auto geometry = [&](const int64_t wpc, const int64_t min_tiles = 1) {
        const int64_t target_waves = (int64_t)h->n_cu * wpc;
        const int64_t tpw = std::max<int64_t>(std::min<int64_t>(min_tiles, a.n_tiles), (a.n_tiles + target_waves - 1) / target_waves);
        a.tiles_per_wave = (int32_t)tpw;
        return dim3((unsigned)((a.n_tiles + tpw - 1) / tpw));
    };
    // start / stop events of the passes (pcs_last_kernel_ms) unless timing is switched off: an event record is a packet of its own
    // between two launches, ~5 us each on the stream (the device LM loop builds once per trial and switches them off)
    const bool timed = h->timing_every > 0;
    hipEvent_t *ev = timed ? ring_slot(h, false) : nullptr;
    if (timed) HIPCHK(hipEventRecord(ev[2], s));
    const int n_pass = h->chain == PCS_CHAIN_TEMPLATE ? 1 : h->chain == PCS_CHAIN_SELF ? 3 : 2;
    for (int pass = 0; pass < n_pass; ++pass) {
        if (h->normal_debug & (256 << pass)) continue;   // profiling: time the passes one by one
        a.order = pass == PASS_SHARED ? h->d_order : pass == PASS_CAMKEY ? h->d_order_ck : h->d_order_ik;
        if (pass != PASS_SHARED && !a.order) return fail(PCS_ERR_STATE, "normal equations: key-sorted visiting order missing");
        a.tab = det_table(h);
        if (a.order && h->sort_tables) {   // the pass's own copy of the table, already in visiting order
            void *const *t = h->d_sorted[pass];
            if ((t[0] || t[1]) && (t[4] || pass == PASS_IMGKEY)) {   // index words AND measurements, or neither
                a.tab.packed = static_cast<const uint32_t *>(t[0]);
                a.tab.cam = static_cast<const int32_t *>(t[1]); a.tab.img = static_cast<const int32_t *>(t[2]); a.tab.key = static_cast<const int32_t *>(t[3]);
                if (t[4]) a.tab.uv = t[4];
                a.order = nullptr;
            }
        }
        hipError_t e;
        a.part = nullptr; a.seg_base = nullptr;
        if (pass == PASS_IMGKEY && h->deterministic) {
            // this pass keeps its atomics: a run is at most n_cams long, so with n_cams <= 64 no address gets more than two contributions
            // (csrc/ba_reduce.hpp) — beyond that the order of three could show in the last bit
            if (h->n_cams > 64) return fail(PCS_ERR_ARG, "deterministic mode: the (image, key) pass of the self chain supports at most 64 cameras (%lld)", (long long)h->n_cams);
        }
        if (pass == PASS_IMGKEY && (h->normal_imgkey_product || h->deterministic)) {
            // 7 KB of LDS and 167 VGPRs per wave: 12 resident per CU.  The kernel waits on dependent loads and on its
            // atomics, so short waves (one or two tiles each, 48 per CU) that keep every slot refilled are fastest
            // (59 / 56 / 55 / 59 us for 12 / 24 / 48 / 96 per CU at N = 1e6, profiles/r02/sweeps.md)
            const dim3 grid = geometry(h->normal_imgkey_wgs_per_cu > 0 ? h->normal_imgkey_wgs_per_cu : 48);
            hipLaunchKernelGGL(ba_normal_imgkey_kernel, grid, dim3(64), normal_imgkey_lds_bytes(), s, a);
            e = hipGetLastError();
        } else {
            // the shared pass's image (22.9 KB) allows 7 waves per CU; the (cam, key) pass's (19.8 KB) allows 8 = two per SIMD,
            // 65.5 against 70.0 us on rig-32-self (profiles/r02/sweeps.md)
            const dim3 grid = geometry(h->wgs_per_cu > 0 ? h->wgs_per_cu : pass == PASS_CAMKEY ? 8 : 7, (h->wgs_per_cu > 0 || a.n_tiles < 1024) ? 1 : 2);
            if (h->deterministic) {   // the kernel stores its finished accumulators per segment, an ordered second pass sums them (csrc/ba_reduce.hpp)
                const int rc = ensure_det_tables(h, pass, a.tiles_per_wave);
                if (rc) return rc;
                const pcs_engine::DetPass &D = h->det[pass];
                if ((int64_t)grid.x != D.n_waves) return fail(PCS_ERR_STATE, "deterministic mode: launch geometry and segment table disagree");
                a.part = D.d_work + D.work_off[0];
                a.seg_base = D.d_idx + D.off[0];
                e = launch_normal(h->chain, pass, h->normal_rows, a, grid, s);
                if (e == hipSuccess) {
                    ReduceArgs ra{};
                    ra.part = a.part; ra.Q = D.d_work + D.work_off[1]; ra.G = D.d_work + D.work_off[2];
                    ra.lr_ptr = D.d_idx + D.off[1]; ra.lr_segs = D.d_idx + D.off[2]; ra.lr_ka = D.d_idx + D.off[3]; ra.lr_kb = D.d_idx + D.off[4];
                    ra.grp_ptr = D.d_idx + D.off[5]; ra.cam_ptr = D.d_idx + D.off[6]; ra.ent_ptr = D.d_idx + D.off[7]; ra.ent_runs = D.d_idx + D.off[8];
                    ra.n_lr = D.n_lr; ra.n_grp = D.n_grp; ra.n_cams = (int32_t)h->n_cams; ra.n_ent = D.n_ent;
                    e = launch_reduce(h->chain, pass, a, ra, s);
                }
            } else {
                e = launch_normal(h->chain, pass, h->normal_rows, a, grid, s);
            }
        }
        if (e != hipSuccess) return fail(PCS_ERR_HIP, "normal-equations kernel launch failed: %s", hipGetErrorString(e));
    }
    if (timed) {
        HIPCHK(hipEventRecord(ev[3], s));
        ++h->ev_count;
        h->events_valid = true;
    }
    HIPCHK(mark_done(h, s));
    return PCS_OK;
}

// Queue slab_prep + the evaluation kernel on `s`.  d_prm must already hold the parameter string.
static int enqueue_eval(pcs_engine *h, const double *d_prm, void *d_resid, void *d_out, bool compact, hipStream_t s) {
    if (h->n <= 0) return fail(PCS_ERR_STATE, "no detections set");
    if (h->chain == PCS_CHAIN_TEMPLATE && !h->have_template) return fail(PCS_ERR_STATE, "template points not set");
    const int mode = (d_resid ? MODE_RESID : 0) | (d_out ? MODE_JAC : 0);
    if (!mode) return PCS_OK;
    HIPCHK(hipSetDevice(h->device));
    // One launch per step for small, run-ordered tables (see pcs_engine::fuse_prep): the waves prepare their own slabs.
    const bool slab_lds_forced = h->variant >= 0 && (h->variant & VAR_SLAB_LDS);
    const bool prep = !compact && !slab_lds_forced && !(h->variant >= 0 && (mode & MODE_JAC) && !(h->variant & VAR_TRANSPOSE)) &&
                      (h->fuse_prep > 0 || (h->fuse_prep < 0 && h->tile_locality >= 0.5 && h->tile_segments <= 2.5 &&
                                            ((h->osize == 8 && (mode & MODE_JAC)) || h->n <= h->fuse_prep_max_n)));
    hipEvent_t no_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    const bool timed = h->timing_every > 0 && (h->eval_count++ % h->timing_every) == 0;
    hipEvent_t *ev = timed ? ring_slot(h, !prep) : no_ev;
    if (!prep) {
        int rc0 = launch_slab_prep(h, d_prm, s, ev[0], ev[1]);
        if (rc0) return rc0;
    } else {
        HIPCHK(order_after_done(h, s));   // keeps `done` meaning "everything queued so far", whatever stream it was on
        h->linearized = false;   // the global slabs are not refreshed: the matrix-free operators need pcs_linearize
    }
    EvalArgs a{};
    a.prm = d_prm; a.extr_off = h->extr_off; a.pose_off = h->pose_off; a.point_off = h->point_off;
    a.tab = det_table(h);
    a.cam_slab = h->d_cam_slab; a.pose_slab = h->d_pose_slab; a.points = h->d_points;
    a.resid = d_resid; a.jac = d_out; a.sink = h->d_sink;
    a.xcd_remap = h->xcd_remap ? 1 : 0;
    a.n = h->n; a.n_cams = (int32_t)h->n_cams; a.n_imgs = (int32_t)h->n_imgs; a.n_keys = (int32_t)h->n_keys;
    a.n_tiles = (h->n + TILE - 1) / TILE;
    if (compact) {
        a.keep = h->d_keep; a.row_off = h->d_row_off;
        if (timed) HIPCHK(hipEventRecord(ev[2], s));
        hipError_t e;
        if (h->compact_variant == 0 && h->dtype == PCS_F64) {  // per-lane stores (first version, kept for A/B)
            const int64_t blocks = std::min<int64_t>((h->n + WG_THREADS - 1) / WG_THREADS, (int64_t)h->n_cu * 8);
            e = launch_compact_t(h->chain, mode, a, dim3((unsigned)blocks), s);
        } else {
            const int64_t wpc = h->wgs_per_cu > 0 ? h->wgs_per_cu : 16;
            const int64_t target_wgs = (int64_t)h->n_cu * wpc;
            int64_t tpw = (a.n_tiles + target_wgs - 1) / target_wgs;
            tpw = std::max<int64_t>(WAVES_PER_WG, (tpw + WAVES_PER_WG - 1) / WAVES_PER_WG * WAVES_PER_WG);
            a.tiles_per_wg = (int32_t)tpw;
            const int64_t grid = (a.n_tiles + tpw - 1) / tpw;
            const size_t vs = 16 / h->osize;
            const size_t wave_lds = ((size_t)HALF * 2 * h->P + 128 / h->osize + 64 + vs - 1) / vs * vs;  // scalars, as in the kernel
            const size_t lds = (mode & MODE_JAC) ? h->osize * (size_t)WAVES_PER_WG * wave_lds : 0;
            e = h->osize == 8 ? launch_compact_tile_t<double>(h->chain, mode, a, dim3((unsigned)grid), lds, s)
                              : launch_compact_tile_t<float>(h->chain, mode, a, dim3((unsigned)grid), lds, s);
        }
        if (e != hipSuccess) return fail(PCS_ERR_HIP, "compact kernel launch failed: %s", hipGetErrorString(e));
        if (timed) HIPCHK(hipEventRecord(ev[3], s));
    } else {
        const bool local = h->tile_locality >= 0.5;
        int variant = h->variant >= 0 ? h->variant : (VAR_TRANSPOSE | VAR_NT | (local || prep ? 0 : VAR_SLAB_LDS));
        if (!(mode & MODE_JAC)) variant &= ~VAR_TRANSPOSE;
        // waves per workgroup: 4 for large tables; a small table (config 2: 1 600 tiles on 256 CUs) is cut into
        // one-wave workgroups so that every CU gets several and the tail of the grid stays short
        int waves = h->waves_per_wg;
        if (waves <= 0) waves = (variant & VAR_SLAB_LDS) ? WAVES_PER_WG : a.n_tiles >= (int64_t)h->n_cu * 32 ? WAVES_PER_WG : a.n_tiles >= (int64_t)h->n_cu * 16 ? 2 : 1;
        // LDS budget: slabs + points (+ one wave-private transpose region per wave)
        const size_t slab_bytes = sizeof(double) * (size_t)(h->n_cams * CAM_STRIDE + h->n_imgs * POSE_STRIDE + padded_points(h->n_keys));
        const size_t tr_bytes = (variant & VAR_TRANSPOSE) ? h->osize * (size_t)waves * HALF * lds_row_stride(2 * h->P, (int)h->osize) : 0;
        if ((variant & VAR_SLAB_LDS) && slab_bytes + tr_bytes > h->lds_limit) variant &= ~VAR_SLAB_LDS;  // read slabs through L1/L2
        const size_t lds = ((variant & VAR_SLAB_LDS) ? slab_bytes : 0) + (prep ? sizeof(double) * (size_t)waves * PAIR_SLAB : 0) + tr_bytes;
        int64_t tpw = h->tiles_per_wg;
        if (tpw <= 0 && h->wgs_per_cu <= 0 && !(variant & VAR_SLAB_LDS)) {
            // slabs through L1/L2: one tile per wave, as many workgroups as that takes.  Up to 1e6 detections this
            // is what 16 workgroups per CU give anyway; at 1e7 it beats 10 tiles per wave by 17 % (profiles/r01/sweeps.md)
            tpw = waves;
        } else if (tpw <= 0) {
            const int64_t wpc = h->wgs_per_cu > 0 ? h->wgs_per_cu : 2;  // LDS-staged slabs: few long-lived workgroups
            const int64_t target_wgs = (int64_t)h->n_cu * wpc;
            tpw = (a.n_tiles + target_wgs - 1) / target_wgs;
            tpw = std::max<int64_t>(waves, (tpw + waves - 1) / waves * waves);
        }
        a.tiles_per_wg = (int32_t)tpw;
        const int64_t grid = (a.n_tiles + tpw - 1) / tpw;
        const EvPair evp{ev[2], ev[3]};  // start / stop of the evaluation kernel itself
        hipError_t e = h->osize == 8 ? launch_eval_t<double>(h->chain, mode, variant, prep, a, dim3((unsigned)grid), 64 * waves, lds, s, evp)
                                     : launch_eval_t<float>(h->chain, mode, variant, prep, a, dim3((unsigned)grid), 64 * waves, lds, s, evp);
        if (e != hipSuccess) return fail(PCS_ERR_HIP, "eval kernel launch failed: %s", hipGetErrorString(e));
    }
    if (timed) {
        ++h->ev_count;
        h->events_valid = true;
    }
    HIPCHK(mark_done(h, s));
    return PCS_OK;
}

template <int CHAIN, int PASS>
static void fill_descriptors(int tg, int32_t *out) {
    for (int m = 0; m < 2; ++m)
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) out[(m * 64 + lane) * 4 + r] = m < normal_mfmas(CHAIN, PASS) ? entry_descriptor<CHAIN, PASS>(m, lane, r, tg) : 0;
}

template <int CHAIN, int PASS>
static void fill_entry_map(int32_t *out) {
    for (int m = 0; m < 2; ++m)
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
                int sa = -1, sb = -1;
                const bool keep = m < normal_mfmas(CHAIN, PASS) && entry_kept<CHAIN, PASS>(m, (lane >> 4) + 4 * r, lane & 15, sa, sb);
                int32_t *o = out + ((m * 64 + lane) * 4 + r) * 2;
                o[0] = keep ? slot_col<CHAIN, PASS>(sa) : -1;
                o[1] = keep ? slot_col<CHAIN, PASS>(sb) : -1;
            }
}

static int stage_params(pcs_engine *h, const double *param_str, hipStream_t s) {
    // the pinned staging buffer is reused: wait until the previous copy out of it has been consumed
    HIPCHK(wait_done_host(h));
    memcpy(h->h_param, param_str, sizeof(double) * h->n_params);
    HIPCHK(hipMemcpyAsync(h->d_param, h->h_param, sizeof(double) * h->n_params, hipMemcpyHostToDevice, s));
    HIPCHK(mark_done(h, s));
    return PCS_OK;
}

static int ensure_scratch(pcs_engine *h, bool want_resid, bool want_jac, bool want_data) {
    HIPCHK(hipSetDevice(h->device));
    if (want_resid && h->resid_capacity < 2 * h->n) {
        if (h->d_resid) HIPCHK(hipFree(h->d_resid));
        HIPCHK(hipMalloc(&h->d_resid, sizeof(double) * 2 * h->n));  // doubles: the legacy cost of a mixed engine writes f64
        h->resid_capacity = 2 * h->n;
    }
    if (want_jac && h->jac_capacity < 2 * h->n * h->P) {
        if (h->d_jac) HIPCHK(hipFree(h->d_jac));
        HIPCHK(hipMalloc(&h->d_jac, h->osize * 2 * h->n * h->P));
        h->jac_capacity = 2 * h->n * h->P;
    }
    if (want_data && h->data_capacity < std::max<int64_t>(1, h->nnz)) {
        if (h->d_data) HIPCHK(hipFree(h->d_data));
        HIPCHK(hipMalloc(&h->d_data, h->osize * std::max<int64_t>(1, h->nnz)));
        h->data_capacity = std::max<int64_t>(1, h->nnz);
    }
    return PCS_OK;
}

// device -> host as float64; `elem` = element size on the device (default: the engine's output type)
static int download(pcs_engine *h, double *dst, const void *d_src, int64_t count, hipStream_t s, size_t elem = 0) {
    if ((elem ? elem : h->osize) == 8) {
        HIPCHK(hipMemcpyAsync(dst, d_src, sizeof(double) * count, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    } else {
        std::vector<float> f(count);
        HIPCHK(hipMemcpyAsync(f.data(), d_src, sizeof(float) * count, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (int64_t i = 0; i < count; ++i) dst[i] = (double)f[i];
    }
    return PCS_OK;
}

extern "C" {

int pcs_eval_device_resident(pcs_engine *h, const double *d_param_str, void *d_resid, void *d_jac, void *stream) {
    if (!h || !d_param_str) return fail(PCS_ERR_ARG, "pcs_eval_device_resident: bad arguments");
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    return enqueue_eval(h, d_param_str, d_resid, d_jac, false, s);
}

int pcs_eval_device(pcs_engine *h, const double *param_str, void *d_resid, void *d_jac, void *stream) {
    if (!h || !param_str) return fail(PCS_ERR_ARG, "pcs_eval_device: bad arguments");
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    HIPCHK(hipSetDevice(h->device));
    int rc = stage_params(h, param_str, s);
    if (rc) return rc;
    return enqueue_eval(h, h->d_param, d_resid, d_jac, false, s);
}

int pcs_eval(pcs_engine *h, const double *param_str, double *resid, double *jac) {
    if (!h || !param_str) return fail(PCS_ERR_ARG, "pcs_eval: bad arguments");
    if (h->n <= 0) return fail(PCS_ERR_STATE, "no detections set");
    int rc = ensure_scratch(h, resid != nullptr, jac != nullptr, false);
    if (rc) return rc;
    rc = pcs_eval_device(h, param_str, resid ? h->d_resid : nullptr, jac ? h->d_jac : nullptr, nullptr);
    if (rc) return rc;
    if (resid && (rc = download(h, resid, h->d_resid, 2 * h->n, h->stream))) return rc;
    if (jac && (rc = download(h, jac, h->d_jac, 2 * h->n * h->P, h->stream))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return PCS_OK;
}

int pcs_device_buffers(pcs_engine *h, void **d_resid, void **d_jac) {
    if (!h) return fail(PCS_ERR_ARG, "pcs_device_buffers: bad arguments");
    if (h->n <= 0) return fail(PCS_ERR_STATE, "no detections set");
    int rc = ensure_scratch(h, d_resid != nullptr, d_jac != nullptr, false);
    if (rc) return rc;
    if (d_resid) *d_resid = h->d_resid;
    if (d_jac) *d_jac = h->d_jac;
    return PCS_OK;
}

int pcs_block_param_inds(pcs_engine *h, int64_t *out) {
    if (!h || !out) return fail(PCS_ERR_ARG, "pcs_block_param_inds: bad arguments");
    const int P = h->P;
    for (int64_t i = 0; i < h->n; ++i) {
        int64_t *o = out + i * P;
        const int64_t c = h->h_cam[i], im = h->h_img[i], k = h->h_key[i];
        int j = 0;
        for (int q = 0; q < 9; ++q) o[j++] = 9 * c + q;
        for (int q = 0; q < 6; ++q) o[j++] = h->extr_off + 6 * c + q;
        if (h->chain != PCS_CHAIN_FREE)
            for (int q = 0; q < 6; ++q) o[j++] = h->pose_off + 6 * im + q;
        if (h->chain != PCS_CHAIN_TEMPLATE)
            for (int q = 0; q < 3; ++q) o[j++] = h->point_off + 3 * k + q;
    }
    return PCS_OK;
}

// keep-mask of one detection's local columns under `unfixed`
static inline uint32_t keep_mask(const pcs_engine *h, const uint8_t *unfixed, int64_t i) {
    if (!unfixed) return (1u << h->P) - 1u;
    const int64_t c = h->h_cam[i], im = h->h_img[i], k = h->h_key[i];
    uint32_t m = 0;
    int j = 0;
    for (int q = 0; q < 9; ++q, ++j) m |= (uint32_t)(unfixed[9 * c + q] != 0) << j;
    for (int q = 0; q < 6; ++q, ++j) m |= (uint32_t)(unfixed[h->extr_off + 6 * c + q] != 0) << j;
    if (h->chain != PCS_CHAIN_FREE)
        for (int q = 0; q < 6; ++q, ++j) m |= (uint32_t)(unfixed[h->pose_off + 6 * im + q] != 0) << j;
    if (h->chain != PCS_CHAIN_TEMPLATE)
        for (int q = 0; q < 3; ++q, ++j) m |= (uint32_t)(unfixed[h->point_off + 3 * k + q] != 0) << j;
    return m;
}

int pcs_csr_structure(pcs_engine *h, const uint8_t *unfixed, int64_t *indices, int64_t *indptr, int64_t *nnz_out) {
    if (!h) return fail(PCS_ERR_ARG, "pcs_csr_structure: bad arguments");
    // conversion = [0, cumsum(unfixed)]: full column -> free column (afb:482)
    std::vector<int64_t> conv(h->n_params + 1, 0);
    for (int64_t p = 0; p < h->n_params; ++p) conv[p + 1] = conv[p] + ((!unfixed || unfixed[p]) ? 1 : 0);
    std::vector<int64_t> cols(h->P);
    int64_t pos = 0;
    if (indptr) indptr[0] = 0;
    for (int64_t i = 0; i < h->n; ++i) {
        const int64_t c = h->h_cam[i], im = h->h_img[i], k = h->h_key[i];
        int j = 0;
        for (int q = 0; q < 9; ++q) cols[j++] = 9 * c + q;
        for (int q = 0; q < 6; ++q) cols[j++] = h->extr_off + 6 * c + q;
        if (h->chain != PCS_CHAIN_FREE)
            for (int q = 0; q < 6; ++q) cols[j++] = h->pose_off + 6 * im + q;
        if (h->chain != PCS_CHAIN_TEMPLATE)
            for (int q = 0; q < 3; ++q) cols[j++] = h->point_off + 3 * k + q;
        for (int row = 0; row < 2; ++row) {  // each detection's index row is used for u and for v (afb:475-479)
            for (int q = 0; q < h->P; ++q) {
                if (!unfixed || unfixed[cols[q]]) {
                    if (indices) indices[pos] = conv[cols[q]];
                    ++pos;
                }
            }
            if (indptr) indptr[2 * i + row + 1] = pos;
        }
    }
    if (nnz_out) *nnz_out = pos;
    return PCS_OK;
}

int pcs_set_unfixed(pcs_engine *h, const uint8_t *unfixed, int64_t *nnz_out) {
    if (!h) return fail(PCS_ERR_ARG, "pcs_set_unfixed: bad arguments");
    if (h->n <= 0) return fail(PCS_ERR_STATE, "no detections set");
    std::vector<uint32_t> keep(h->n);
    std::vector<int64_t> off(h->n);
    int64_t pos = 0;
    for (int64_t i = 0; i < h->n; ++i) {
        keep[i] = keep_mask(h, unfixed, i);
        off[i] = pos;
        pos += 2 * __builtin_popcount(keep[i]);
    }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(wait_done_host(h));   // an evaluation on any stream may still read the masks
    HIPCHK(hipStreamSynchronize(h->stream));
    if (!h->d_keep) HIPCHK(hipMalloc(&h->d_keep, sizeof(uint32_t) * h->n));
    if (!h->d_row_off) HIPCHK(hipMalloc(&h->d_row_off, sizeof(int64_t) * h->n));
    HIPCHK(hipMemcpy(h->d_keep, keep.data(), sizeof(uint32_t) * h->n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_row_off, off.data(), sizeof(int64_t) * h->n, hipMemcpyHostToDevice));
    h->nnz = pos;
    if (nnz_out) *nnz_out = pos;
    return PCS_OK;
}

int pcs_eval_compact_device(pcs_engine *h, const double *param_str, void *d_resid, void *d_data, void *stream) {
    if (!h || !param_str) return fail(PCS_ERR_ARG, "pcs_eval_compact_device: bad arguments");
    if (h->nnz < 0) return fail(PCS_ERR_STATE, "pcs_set_unfixed has not been called");
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    HIPCHK(hipSetDevice(h->device));
    int rc = stage_params(h, param_str, s);
    if (rc) return rc;
    return enqueue_eval(h, h->d_param, d_resid, d_data, true, s);
}

int pcs_eval_compact(pcs_engine *h, const double *param_str, double *resid, double *data) {
    if (!h || !param_str) return fail(PCS_ERR_ARG, "pcs_eval_compact: bad arguments");
    if (h->nnz < 0) return fail(PCS_ERR_STATE, "pcs_set_unfixed has not been called");
    int rc = ensure_scratch(h, resid != nullptr, false, data != nullptr);
    if (rc) return rc;
    rc = pcs_eval_compact_device(h, param_str, resid ? h->d_resid : nullptr, data ? h->d_data : nullptr, nullptr);
    if (rc) return rc;
    if (resid && (rc = download(h, resid, h->d_resid, 2 * h->n, h->stream))) return rc;
    if (data && h->nnz > 0 && (rc = download(h, data, h->d_data, h->nnz, h->stream))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return PCS_OK;
}

int pcs_legacy_cost(pcs_engine *h, const double *im_points, const double *proj, const double *intrinsics, const double *dists,
                    double *errors) {
    if (!h || !im_points || !proj || !intrinsics || !dists || !errors) return fail(PCS_ERR_ARG, "pcs_legacy_cost: bad arguments");
    if (h->n <= 0) return fail(PCS_ERR_STATE, "no detections set");
    if (h->chain == PCS_CHAIN_FREE) return fail(PCS_ERR_ARG, "pcs_legacy_cost: needs an engine with images (template or self chain)");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int64_t n_pts = h->n_imgs * h->n_keys * 3;
    if (n_pts > h->im_points_capacity) {
        if (h->d_im_points) HIPCHK(hipFree(h->d_im_points));
        HIPCHK(hipMalloc(&h->d_im_points, sizeof(double) * n_pts));
        h->im_points_capacity = n_pts;
    }
    if (!h->d_cam_tab) HIPCHK(hipMalloc(&h->d_cam_tab, sizeof(double) * h->n_cams * LEGACY_STRIDE));
    int rc = ensure_scratch(h, true, false, false);
    if (rc) return rc;
    std::vector<double> tab((size_t)h->n_cams * LEGACY_STRIDE, 0.0);
    for (int64_t c = 0; c < h->n_cams; ++c) {
        double *t = tab.data() + c * LEGACY_STRIDE;
        for (int j = 0; j < 12; ++j) t[j] = proj[12 * c + j];
        const double *K = intrinsics + 9 * c;
        t[12] = K[0]; t[13] = K[2]; t[14] = K[4]; t[15] = K[5];  // focal_0, centre_0, focal_1, centre_1 (ch:453-454)
        for (int j = 0; j < 5; ++j) t[16 + j] = dists[5 * c + j];
    }
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpyAsync(h->d_im_points, im_points, sizeof(double) * n_pts, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->d_cam_tab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));   // `tab` is a local
    const int64_t n_tiles = (h->n + 63) / 64;
    const dim3 grid((unsigned)std::min<int64_t>((n_tiles + 3) / 4, (int64_t)h->n_cu * 16));
    hipEvent_t *ev = ring_slot(h, false);
    hipExtLaunchKernelGGL(legacy_cost_kernel, grid, dim3(256), 0, s, ev[2], ev[3], 0, det_table(h), (const double *)h->d_im_points,
                          (const double *)h->d_cam_tab, (double *)h->d_resid, h->n, h->n_keys, h->d_sink);
    HIPCHK(hipGetLastError());
    ++h->ev_count;
    h->events_valid = true;
    HIPCHK(mark_done(h, s));
    return download(h, errors, h->d_resid, 2 * h->n, s, sizeof(double));
}

int pcs_linearize(pcs_engine *h, const double *param_str) {
    if (!h || !param_str) return fail(PCS_ERR_ARG, "pcs_linearize: bad arguments");
    if (h->n <= 0) return fail(PCS_ERR_STATE, "no detections set");
    if (h->chain == PCS_CHAIN_TEMPLATE && !h->have_template) return fail(PCS_ERR_STATE, "template points not set");
    HIPCHK(hipSetDevice(h->device));
    int rc = stage_params(h, param_str, h->stream);
    if (rc) return rc;
    rc = launch_slab_prep(h, h->d_param, h->stream);
    if (rc) return rc;
    HIPCHK(mark_done(h, h->stream));
    return PCS_OK;
}

int pcs_matfree(pcs_engine *h, int op, const double *in, double *out, double *cost) {
    if (!h || op < OP_JV || op > OP_GRAD || !out) return fail(PCS_ERR_ARG, "pcs_matfree: bad arguments");
    if ((op == OP_JV || op == OP_JTU || op == OP_JTJV) && !in) return fail(PCS_ERR_ARG, "pcs_matfree: this operator needs an input vector");
    if (!h->linearized) return fail(PCS_ERR_STATE, "pcs_matfree: call pcs_linearize (or an evaluation) first");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    HIPCHK(order_after_done(h, s));  // the slabs may have been prepared on a caller stream
    const int64_t n_in = (op == OP_JTU) ? 2 * h->n : (op == OP_JV || op == OP_JTJV) ? h->n_params : 0;
    const int64_t n_out = (op == OP_JV) ? 2 * h->n : h->n_params;
    if (n_in > h->vin_capacity) {
        if (h->d_vin) HIPCHK(hipFree(h->d_vin));
        HIPCHK(hipMalloc(&h->d_vin, sizeof(double) * n_in));
        h->vin_capacity = n_in;
    }
    if (n_out > h->vout_capacity) {
        if (h->d_vout) HIPCHK(hipFree(h->d_vout));
        HIPCHK(hipMalloc(&h->d_vout, sizeof(double) * n_out));
        h->vout_capacity = n_out;
    }
    if (!h->d_cost) HIPCHK(hipMalloc(&h->d_cost, sizeof(double)));
    if (n_in) HIPCHK(hipMemcpyAsync(h->d_vin, in, sizeof(double) * n_in, hipMemcpyHostToDevice, s));
    if (op != OP_JV) HIPCHK(hipMemsetAsync(h->d_vout, 0, sizeof(double) * n_out, s));
    if (op == OP_GRAD) HIPCHK(hipMemsetAsync(h->d_cost, 0, sizeof(double), s));
    MatfreeArgs a{};
    a.tab = det_table(h);
    a.cam_slab = h->d_cam_slab; a.pose_slab = h->d_pose_slab; a.points = h->d_points;
    a.vin = h->d_vin; a.vout = h->d_vout; a.cost = h->d_cost;
    a.n = h->n; a.n_tiles = (h->n + TILE - 1) / TILE;
    a.extr_off = h->extr_off; a.pose_off = h->pose_off; a.point_off = h->point_off;
    a.n_params = (int32_t)h->n_params;
    // workgroup-private LDS accumulators (+ one reduction panel per wave) when they fit
    const size_t acc_bytes = sizeof(double) * (size_t)((h->n_params + 1) & ~(int64_t)1);
    const size_t lds_need = acc_bytes + sizeof(double) * (size_t)WAVES_PER_WG * RED_PANEL;
    const bool lds_acc = h->matfree_lds && op != OP_JV && lds_need <= 150 * 1024;
    const size_t lds = lds_acc ? lds_need : 0;
    const int64_t wpc = h->wgs_per_cu > 0 ? h->wgs_per_cu : (op == OP_JV ? 8 : 2);
    const int64_t target_wgs = (int64_t)h->n_cu * wpc;
    int64_t tpw = h->tiles_per_wg > 0 ? h->tiles_per_wg : (a.n_tiles + target_wgs - 1) / target_wgs;   // option tiles_per_wg: tests of the tile pipeline
    tpw = std::max<int64_t>(WAVES_PER_WG, (tpw + WAVES_PER_WG - 1) / WAVES_PER_WG * WAVES_PER_WG);
    a.tiles_per_wg = (int32_t)tpw;
    const dim3 grid((unsigned)((a.n_tiles + tpw - 1) / tpw));
    hipEvent_t *ev = ring_slot(h, false);
    HIPCHK(hipEventRecord(ev[2], s));
    hipError_t e = launch_matfree_t(h->chain, op, lds_acc, a, grid, lds, s);
    if (e != hipSuccess) return fail(PCS_ERR_HIP, "matfree kernel launch failed: %s", hipGetErrorString(e));
    HIPCHK(hipEventRecord(ev[3], s));
    ++h->ev_count;
    h->events_valid = true;
    HIPCHK(mark_done(h, s));
    HIPCHK(hipMemcpyAsync(out, h->d_vout, sizeof(double) * n_out, hipMemcpyDeviceToHost, s));
    if (op == OP_GRAD && cost) HIPCHK(hipMemcpyAsync(cost, h->d_cost, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return PCS_OK;
}

int pcs_normal_equations_device(pcs_engine *h, const double *param_str, double *d_H, double *d_g, double *d_cost, void *stream) {
    if (!h || !param_str || !d_H || !d_g || !d_cost) return fail(PCS_ERR_ARG, "pcs_normal_equations_device: bad arguments");
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    HIPCHK(hipSetDevice(h->device));
    int rc = stage_params(h, param_str, s);
    if (rc) return rc;
    return enqueue_normal(h, h->d_param, d_H, d_g, d_cost, s);
}

int pcs_normal_layout(const pcs_engine *h, int64_t *out5) {
    if (!h || !out5) return fail(PCS_ERR_ARG, "pcs_normal_layout: bad arguments");
    const BlockLayout L = block_layout(h);
    out5[0] = L.n_lead; out5[1] = L.n_trail; out5[2] = L.tb;
    out5[3] = L.a_len() + L.b_len() + L.c_len() + h->n_params + 1;
    out5[4] = h->n_params;
    return PCS_OK;
}

int pcs_normal_blocks_device(pcs_engine *h, const double *d_param_str, double *d_packed, void *stream) {
    if (!h || !d_param_str || !d_packed) return fail(PCS_ERR_ARG, "pcs_normal_blocks_device: bad arguments");
    if (reinterpret_cast<uintptr_t>(d_packed) % 16) return fail(PCS_ERR_ARG, "pcs_normal_blocks_device: the packed buffer must be 16-byte aligned");
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    const BlockLayout L = block_layout(h);
    double *d_g = d_packed + L.a_len() + L.b_len() + L.c_len();
    return enqueue_normal(h, d_param_str, d_packed, d_g, d_g + h->n_params, s, true);
}

static int enqueue_schur_prepare(const BlockLayout &L, double *d_packed, const uint8_t *d_fixed, const double *d_lambda, double *d_linvt, double *d_u,
                                 double *d_V, double *d_S, double *d_rhs, double *d_dvec, double *d_gm, int32_t *d_status, hipStream_t s, const int32_t *d_stop,
                                 double *d_fill = nullptr, int64_t fill_n = 0, const int32_t *d_sel = nullptr, int64_t alt = 0) {
    SchurArgs a{};
    a.sel = d_sel; a.alt = alt;
    a.fill = reinterpret_cast<uint64_t *>(d_fill); a.fill_n = d_fill ? fill_n : 0;
    a.A = d_packed; a.B = d_packed + L.a_len(); a.C = a.B + L.b_len(); a.g = a.C + L.c_len();
    a.fixed = d_fixed; a.lambda = d_lambda;
    a.linvt = d_linvt; a.u = d_u; a.V = d_V; a.S = d_S; a.rhs = d_rhs; a.dvec = d_dvec; a.gm = d_gm; a.status = d_status;
    a.n_lead = L.n_lead; a.n_trail = L.n_trail; a.n_ent = L.n_ent; a.trail_off = L.trail_off;
    a.stop = d_stop;
    auto blocks = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
    // trailing blocks and the leading block are independent: one launch (schur_trail_lead_kernel); V needs the trailing factors
    const int64_t nbt = (L.n_lead + 31) / 32, trail_blocks = (L.n_ent + 255) / 256;
    a.trail_blocks = (int32_t)trail_blocks;
    if (trail_blocks + nbt * nbt > 0) {
        const dim3 grid((unsigned)(trail_blocks + nbt * nbt));
        if (L.tb == 6) hipLaunchKernelGGL(schur_trail_lead_kernel<6>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(schur_trail_lead_kernel<3>, grid, dim3(256), 0, s, a);
        HIPCHK(hipGetLastError());
    }
    if (L.n_ent > 0 && L.n_lead > 0) {
        if (L.tb == 6) hipLaunchKernelGGL(schur_v_kernel<6>, blocks(L.n_lead * L.n_ent), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(schur_v_kernel<3>, blocks(L.n_lead * L.n_ent), dim3(256), 0, s, a);
        HIPCHK(hipGetLastError());
    }
    return PCS_OK;
}
static int enqueue_schur_prepare(pcs_engine *h, double *d_packed, const uint8_t *d_fixed, const double *d_lambda, double *d_linvt, double *d_u,
                                 double *d_V, double *d_S, double *d_rhs, double *d_dvec, double *d_gm, int32_t *d_status, hipStream_t s, const int32_t *d_stop,
                                 double *d_fill = nullptr, int64_t fill_n = 0, const int32_t *d_sel = nullptr, int64_t alt = 0) {
    return enqueue_schur_prepare(block_layout(h), d_packed, d_fixed, d_lambda, d_linvt, d_u, d_V, d_S, d_rhs, d_dvec, d_gm, d_status, s, d_stop, d_fill, fill_n, d_sel, alt);
}

int pcs_schur_prepare(pcs_engine *h, double *d_packed, const uint8_t *d_fixed, const double *d_lambda, double *d_linvt, double *d_u,
                      double *d_V, double *d_S, double *d_rhs, double *d_dvec, double *d_gm, int32_t *d_status, void *stream) {
    if (!h || !d_packed || !d_fixed || !d_lambda || !d_linvt || !d_u || !d_V || !d_S || !d_rhs || !d_dvec || !d_gm || !d_status)
        return fail(PCS_ERR_ARG, "pcs_schur_prepare: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    return enqueue_schur_prepare(h, d_packed, d_fixed, d_lambda, d_linvt, d_u, d_V, d_S, d_rhs, d_dvec, d_gm, d_status, stream ? (hipStream_t)stream : h->stream, nullptr);
}

int pcs_lm_decide(pcs_engine *h, const double *d_cost_old, const double *d_cost_new, const double *d_dvec, const double *d_gm, const double *d_delta,
                  const double *d_ps, const uint8_t *d_fixed, int32_t *d_status, double *d_lambda, double *d_stats, void *stream) {
    if (!h || !d_cost_old || !d_cost_new || !d_dvec || !d_gm || !d_delta || !d_ps || !d_fixed || !d_status || !d_lambda || !d_stats)
        return fail(PCS_ERR_ARG, "pcs_lm_decide: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    LmDecideArgs a{};
    a.tail[0] = d_cost_old; a.tail[1] = d_cost_new; a.ps2[0] = a.ps2[1] = d_ps;
    a.dvec = d_dvec; a.gm = d_gm; a.delta = d_delta; a.fixed = d_fixed; a.status = d_status; a.lambda = d_lambda; a.stats = d_stats; a.n_params = h->n_params;
    hipLaunchKernelGGL(lm_decide_kernel, dim3(1), dim3(1024), 0, s, a);
    HIPCHK(hipGetLastError());
    return PCS_OK;
}

static int enqueue_schur_finish(const BlockLayout &L, const double *d_linvt, const double *d_u, const double *d_w, const double *d_xlead, const uint8_t *d_fixed,
                                double *d_delta, const double *d_ps_in, double *d_ps_out, hipStream_t s, const int32_t *d_stop, const int32_t *d_sel = nullptr,
                                double *d_vote = nullptr, int64_t vote_alt = 0, const int32_t *d_status = nullptr, double *d_zero = nullptr, int64_t zero_n = 0,
                                int n_cu = 256) {
    SchurBackArgs a{};
    a.zero = d_zero; a.zero_n = d_zero ? zero_n : 0;
    a.sel = d_sel; a.vote = d_vote; a.vote_alt = vote_alt; a.status = d_status;
    a.linvt = d_linvt; a.u = d_u; a.w = d_w; a.xl = d_xlead; a.fixed = d_fixed; a.delta = d_delta;
    a.ps_in = d_ps_in; a.ps_out = d_ps_out;
    a.n_lead = L.n_lead; a.n_ent = L.n_ent; a.trail_off = L.trail_off;
    a.stop = d_stop;
    const int64_t n = std::max(L.n_lead, L.n_ent);
    // with a buffer to zero on the way: enough workgroups for that as well (8 doubles per thread and pass, at most four workgroups per CU)
    const int64_t zero_blocks = a.zero ? std::min<int64_t>((a.zero_n / 8 + 255) / 256, (int64_t)n_cu * 4) : 0;
    const dim3 grid((unsigned)std::max<int64_t>((n + 255) / 256, zero_blocks));
    if (L.tb == 6) hipLaunchKernelGGL(schur_back_kernel<6>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(schur_back_kernel<3>, grid, dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    return PCS_OK;
}
static int enqueue_schur_finish(pcs_engine *h, const double *d_linvt, const double *d_u, const double *d_w, const double *d_xlead, const uint8_t *d_fixed,
                                double *d_delta, const double *d_ps_in, double *d_ps_out, hipStream_t s, const int32_t *d_stop, const int32_t *d_sel = nullptr,
                                double *d_vote = nullptr, int64_t vote_alt = 0, const int32_t *d_status = nullptr) {
    return enqueue_schur_finish(block_layout(h), d_linvt, d_u, d_w, d_xlead, d_fixed, d_delta, d_ps_in, d_ps_out, s, d_stop, d_sel, d_vote, vote_alt, d_status);
}

int pcs_schur_finish(pcs_engine *h, const double *d_linvt, const double *d_u, const double *d_w, const double *d_xlead, const uint8_t *d_fixed,
                     double *d_delta, const double *d_ps_in, double *d_ps_out, void *stream) {
    if (!h || !d_linvt || !d_u || !d_w || !d_xlead || !d_fixed || !d_delta || ((d_ps_in == nullptr) != (d_ps_out == nullptr)))
        return fail(PCS_ERR_ARG, "pcs_schur_finish: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    return enqueue_schur_finish(h, d_linvt, d_u, d_w, d_xlead, d_fixed, d_delta, d_ps_in, d_ps_out, stream ? (hipStream_t)stream : h->stream, nullptr);
}

int64_t pcs_dense_spd_work_len(int64_t n) {   // launch-per-column form: inverses + diagonal tiles + y; one-launch form: flags + x + y
    if (n <= 0) return -1;
    const int64_t nb = (n + 31) / 32;
    return std::max<int64_t>(2 * nb * 32 * 32 + nb * 32, cp_work_doubles(nb));
}

static int device_cu_count(int device);
// launch geometry of S -= V V' (csrc/ba_schur.hpp): tile width, tiles of the lower triangle, K split.  `ordered` = the partial sums of a
// split go through a workspace and are subtracted in order (deterministic mode) instead of meeting in atomics.
struct SyrkGeometry { bool big; int64_t tw, tiles, ksplit, kchunk; };
static SyrkGeometry syrk_geometry(int64_t n_lead, int64_t n_trail, bool ordered) {
    // 64 x 64 tiles once 32 x 32 ones alone would fill the chip twice over (their operand traffic, not the matrix cores, is the bound then:
    // rig-32-self 175 us -> ~100 us); PCS_SYRK_TILE=32 / 64 forces one form (A/B)
    static const int forced = getenv("PCS_SYRK_TILE") ? atoi(getenv("PCS_SYRK_TILE")) : 0;
    const int64_t nb32 = (n_lead + 31) / 32;
    SyrkGeometry g{};
    g.big = forced == 64 || (forced != 32 && nb32 * (nb32 + 1) / 2 >= 1024);
    g.tw = g.big ? 64 : 32;
    const int64_t nb = (n_lead + g.tw - 1) / g.tw;
    g.tiles = nb * (nb + 1) / 2;
    // split K until ~512 workgroups exist (rig-32: 120 tiles x 5; the 2e4-point free chain: 21 tiles x 25 of 60 000 columns)
    int64_t ksplit = std::min<int64_t>((512 + g.tiles - 1) / g.tiles, (n_trail + 127) / 128);
    ksplit = std::max<int64_t>(1, ksplit);
    int64_t kchunk = ((n_trail + ksplit - 1) / ksplit + 63) / 64 * 64;
    ksplit = (n_trail + kchunk - 1) / kchunk;
    if (g.big) {
        // 64 x 64 tiles run two workgroups per CU: split K so that the workgroups fill whole rounds of the resident ones — the cost of a
        // split = rounds x (columns per workgroup + ~64 columns' worth of ramp and atomics); rig-32-self: 378 tiles x 4 = 2.95 rounds.
        // Ordered mode: every split also writes and re-reads a 32 KB partial tile (rig-32-self: 50 MB per solve for four splits) —
        // priced as 48 more columns per split.  (Priced at 192 the model left rig-32-self unsplit: 378 workgroups of 1 458 columns on
        // 512 slots took 172.7 us against 133.4 us for the four-way split with atomics, profiles/r05/lm_trace_rig32_self_form11.log.)
        int dev = 0;
        const int cus = hipGetDevice(&dev) == hipSuccess && device_cu_count(dev) > 0 ? device_cu_count(dev) : 256;
        const int64_t slots = 2 * (int64_t)cus;
        int64_t best = INT64_MAX;
        for (int64_t ks = 1; ks <= std::max<int64_t>(1, n_trail / 128); ++ks) {
            const int64_t kc = ((n_trail + ks - 1) / ks + 63) / 64 * 64, real = (n_trail + kc - 1) / kc;
            const int64_t cost = (g.tiles * real + slots - 1) / slots * (kc + 64 + ((ordered && real > 1) ? 48 : 0));
            if (cost < best) { best = cost; ksplit = real; kchunk = kc; }
        }
    }
    g.ksplit = ksplit; g.kchunk = kchunk;
    return g;
}
// doubles of the ordered mode's workspace (0: the product is not split, nothing is needed)
static int64_t syrk_work_doubles(int64_t n_lead, int64_t n_trail) {
    const SyrkGeometry g = syrk_geometry(n_lead, n_trail, true);
    return g.ksplit > 1 ? g.ksplit * (g.tiles * g.tw * g.tw + n_lead) : 0;
}

static int enqueue_schur_syrk(int64_t n_lead, int64_t n_trail, const double *d_V, int64_t ldv, double *d_S, int64_t lds, const double *d_u, double *d_rhs,
                              hipStream_t s, const int32_t *d_stop, double *d_ws = nullptr, int64_t ws_doubles = 0) {
    const bool ordered = d_ws != nullptr;
    const SyrkGeometry g = syrk_geometry(n_lead, n_trail, ordered);
    if (ordered && g.ksplit > 1 && ws_doubles < g.ksplit * (g.tiles * g.tw * g.tw + n_lead))
        return fail(PCS_ERR_ARG, "pcs_schur_syrk: the ordered mode needs a workspace of %lld doubles (pcs_schur_syrk_work_len), got %lld",
                    (long long)(g.ksplit * (g.tiles * g.tw * g.tw + n_lead)), (long long)ws_doubles);
    SchurSyrkArgs a{d_V, d_S, d_u, d_rhs, (int32_t)n_lead, (int32_t)n_trail, (int32_t)ldv, (int32_t)lds, (int32_t)g.ksplit, (int32_t)g.kchunk, d_stop};
    a.ws = (ordered && g.ksplit > 1) ? d_ws : nullptr;
    a.ws_rhs = a.ws ? d_ws + g.ksplit * g.tiles * g.tw * g.tw : nullptr;
    a.tiles = (int32_t)g.tiles;
    if (g.big) hipLaunchKernelGGL(schur_syrk64_kernel, dim3((unsigned)(g.tiles * g.ksplit)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(schur_syrk_kernel, dim3((unsigned)(g.tiles * g.ksplit)), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    if (a.ws) {
        if (g.big) hipLaunchKernelGGL(schur_syrk_reduce_kernel<64>, dim3((unsigned)(g.tiles * 16)), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(schur_syrk_reduce_kernel<32>, dim3((unsigned)(g.tiles * 4)), dim3(256), 0, s, a);
        HIPCHK(hipGetLastError());
    }
    return PCS_OK;
}

int64_t pcs_schur_syrk_work_len(int64_t n_lead, int64_t n_trail) {
    if (n_lead <= 0 || n_trail < 0) return -1;
    return syrk_work_doubles(n_lead, n_trail);
}

int pcs_schur_syrk_ordered(int device, int64_t n_lead, int64_t n_trail, const double *d_V, int64_t ldv, double *d_S, int64_t lds, const double *d_u,
                           double *d_rhs, double *d_work, int64_t work_doubles, void *stream) {
    if (n_lead <= 0 || n_lead > (1 << 15) || n_trail < 0 || n_trail > (1ll << 30) || ldv < n_trail || lds < n_lead || !d_S || (n_trail && !d_V) || (d_u && !d_rhs) || !d_work)
        return fail(PCS_ERR_ARG, "pcs_schur_syrk_ordered: bad arguments");
    if (device < 0 || device >= pcs_device_count()) return fail(PCS_ERR_NODEVICE, "pcs_schur_syrk_ordered: device %d not available", device);
    if (n_trail == 0) return PCS_OK;
    HIPCHK(hipSetDevice(device));
    return enqueue_schur_syrk(n_lead, n_trail, d_V, ldv, d_S, lds, d_u, d_rhs, (hipStream_t)stream, nullptr, d_work, work_doubles);
}

int pcs_schur_syrk(int device, int64_t n_lead, int64_t n_trail, const double *d_V, int64_t ldv, double *d_S, int64_t lds, const double *d_u,
                   double *d_rhs, void *stream) {
    if (n_lead <= 0 || n_lead > (1 << 15) || n_trail < 0 || n_trail > (1ll << 30) || ldv < n_trail || lds < n_lead || !d_S || (n_trail && !d_V) || (d_u && !d_rhs))
        return fail(PCS_ERR_ARG, "pcs_schur_syrk: bad arguments");
    if (device < 0 || device >= pcs_device_count()) return fail(PCS_ERR_NODEVICE, "pcs_schur_syrk: device %d not available", device);
    if (n_trail == 0) return PCS_OK;
    HIPCHK(hipSetDevice(device));
    return enqueue_schur_syrk(n_lead, n_trail, d_V, ldv, d_S, lds, d_u, d_rhs, (hipStream_t)stream, nullptr);
}

int pcs_schur_vtx(int device, int64_t n_lead, int64_t n_trail, const double *d_V, int64_t ldv, const double *d_x, double *d_w, void *stream) {
    if (n_lead <= 0 || n_trail < 0 || n_trail > (1ll << 30) || ldv < n_trail || (n_trail && (!d_V || !d_x || !d_w))) return fail(PCS_ERR_ARG, "pcs_schur_vtx: bad arguments");
    if (device < 0 || device >= pcs_device_count()) return fail(PCS_ERR_NODEVICE, "pcs_schur_vtx: device %d not available", device);
    if (n_trail == 0) return PCS_OK;
    HIPCHK(hipSetDevice(device));
    launch_schur_vtx(d_V, d_x, d_w, (int)n_lead, (int)n_trail, (int)ldv, nullptr, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return PCS_OK;
}

static int device_cu_count(int device) {
    static std::atomic<int> cached[64];
    if (device < 0 || device >= 64) return 0;
    int v = cached[device].load();
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) v = 0;
        cached[device].store(v);
    }
    return v;
}

int pcs_dense_spd_solve(int device, int64_t n, double *d_S, int64_t ld, const double *d_rhs, double *d_x, double *d_work, int32_t *d_status, void *stream) {
    return pcs_dense_spd_solve_algo(device, n, d_S, ld, d_rhs, d_x, d_work, d_status, stream, PCS_SPD_AUTO);
}

static int enqueue_dense_spd(int device, int64_t n, double *d_S, int64_t ld, const double *d_rhs, double *d_x, double *d_work, int32_t *d_status, void *stream,
                             int algorithm, const int32_t *d_stop, bool prefilled = false, int64_t timeout_us = 250000);

// does a solve of size n with this algorithm request take the ONE persistent launch (csrc/ba_chol_persist.hpp)?  Wherever its tiles fit
// the chip's LDS: n <= 1 984 on 256 CUs.  (Round 5 also tried ONE workgroup with the whole matrix in its LDS for n <= 160 — nothing to hand
// over —: 110 us at n = 120 against the persistent kernel's 47: fourteen workgroups load, update and factor their tiles side by side, one
// workgroup does it all in sequence.  Dropped.)
static bool dense_spd_is_one_launch(int device, int64_t n, int algorithm) {
    static const bool env_launches = getenv("PCS_CHOL_LAUNCHES") != nullptr;   // A/B switch for whole runs
    return n > 0 && cp_fits(n, device_cu_count(device)) && (algorithm == PCS_SPD_ONE_LAUNCH || (algorithm == PCS_SPD_AUTO && !env_launches));
}

int pcs_dense_spd_solve_algo(int device, int64_t n, double *d_S, int64_t ld, const double *d_rhs, double *d_x, double *d_work, int32_t *d_status, void *stream,
                             int algorithm) {
    return enqueue_dense_spd(device, n, d_S, ld, d_rhs, d_x, d_work, d_status, stream, algorithm, nullptr);
}

int pcs_dense_spd_solve_opts(int device, int64_t n, double *d_S, int64_t ld, const double *d_rhs, double *d_x, double *d_work, int32_t *d_status, void *stream,
                             int algorithm, int64_t timeout_us) {
    if (timeout_us < 1 || timeout_us > 60000000) return fail(PCS_ERR_ARG, "pcs_dense_spd_solve_opts: timeout_us must be in [1, 60000000]");
    return enqueue_dense_spd(device, n, d_S, ld, d_rhs, d_x, d_work, d_status, stream, algorithm, nullptr, false, timeout_us);
}

static int enqueue_dense_spd(int device, int64_t n, double *d_S, int64_t ld, const double *d_rhs, double *d_x, double *d_work, int32_t *d_status, void *stream,
                             int algorithm, const int32_t *d_stop, bool prefilled, int64_t timeout_us) {
    constexpr int NB = 32;
    if (n <= 0 || n > (1 << 15) || ld < n || !d_S || !d_rhs || !d_x || !d_work || !d_status) return fail(PCS_ERR_ARG, "pcs_dense_spd_solve: bad arguments");
    if (algorithm != PCS_SPD_AUTO && algorithm != PCS_SPD_LAUNCHES && algorithm != PCS_SPD_ONE_LAUNCH) return fail(PCS_ERR_ARG, "pcs_dense_spd_solve: unknown algorithm %d", algorithm);
    if (device < 0 || device >= pcs_device_count()) return fail(PCS_ERR_NODEVICE, "pcs_dense_spd_solve: device %d not available", device);
    HIPCHK(hipSetDevice(device));
    if (algorithm == PCS_SPD_ONE_LAUNCH && !cp_fits(n, device_cu_count(device)))
        return fail(PCS_ERR_ARG, "pcs_dense_spd_solve: n = %lld does not fit the one-launch form on %d compute units", (long long)n, device_cu_count(device));
    if (dense_spd_is_one_launch(device, n, algorithm)) {
        HIPCHK(cp_launch(n, d_S, ld, d_rhs, d_x, d_work, d_status, device_cu_count(device), (hipStream_t)stream, 1.0e-6 * (double)timeout_us, nullptr, d_stop, prefilled));
        return PCS_OK;
    }
    hipStream_t s = (hipStream_t)stream;   // NULL = the default stream
    const int nblk = (int)((n + NB - 1) / NB);
    double *d_ldiag = d_work + (int64_t)nblk * NB * NB;
    double *d_y = d_ldiag + (int64_t)nblk * NB * NB;
    CholArgs a{d_S, d_work, d_ldiag, d_status, (int32_t)n, (int32_t)ld, 0, d_rhs, d_y, d_stop};
    a.k = 0;
    hipLaunchKernelGGL(chol_panel_kernel<NB>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    for (int k = 0; k + 1 < nblk; ++k) {   // trailing update with column k + panel step of column k + 1, one launch
        a.k = k;
        const int m = nblk - k - 1;
        hipLaunchKernelGGL(chol_step_kernel<NB>, dim3((unsigned)(m * (m + 1) / 2)), dim3(256), 0, s, a);
    }
    HIPCHK(hipGetLastError());
    // backward sweep L' x = y: pieces of at most 16 blocks in one workgroup each, lower-right first; between two pieces a GEMV over
    // many workgroups takes the solved rows out of the rest of y (csrc/ba_dense_chol.hpp)
    struct Rec {
        static void run(hipStream_t s, const CholSolveArgs &base, int kb0, int kb1) {
            constexpr int NBk = 32;
            if (kb1 - kb0 <= 16) {
                CholSolveArgs b = base;
                b.kb0 = kb0; b.kb1 = kb1;
                const size_t lds = sizeof(double) * ((size_t)(kb1 - kb0) * NBk + NBk);
                hipLaunchKernelGGL(chol_solve_kernel<NBk>, dim3(1), dim3(512), lds, s, b);
                return;
            }
            const int mid = kb0 + (kb1 - kb0 + 1) / 2;
            run(s, base, mid, kb1);
            const int r0 = mid * NBk, r1 = std::min<int>(kb1 * NBk, base.n), c0 = kb0 * NBk, c1 = mid * NBk;
            hipLaunchKernelGGL(chol_gemv_t_kernel, dim3((unsigned)((c1 - c0 + 63) / 64)), dim3(1024), 0, s, (const double *)base.L, (int)base.ld, (const double *)base.x, base.y,
                               r0, r1, c0, c1, base.stop);
            run(s, base, kb0, mid);
        }
    };
    CholSolveArgs b{d_S, d_work, d_ldiag, d_y, d_x, (int32_t)n, (int32_t)ld, 0, nblk, d_stop};
    Rec::run(s, b, 0, nblk);
    HIPCHK(hipGetLastError());
    return PCS_OK;
}

// The fused forms of the trial's small kernels (csrc/ba_lm_fused.hpp).
static int enqueue_schur_prep_fused(const BlockLayout &L, const pcs_lm_buffers *b, hipStream_t s, const int32_t *d_stop, const int32_t *d_sel, int64_t alt,
                                    double *d_fill, int64_t fill_n) {
    SchurArgs a{};
    a.sel = d_sel; a.alt = alt;
    a.fill = reinterpret_cast<uint64_t *>(d_fill); a.fill_n = d_fill ? fill_n : 0;
    a.A = b->packed[0]; a.B = a.A + L.a_len(); a.C = a.B + L.b_len(); a.g = a.C + L.c_len();
    a.fixed = b->fixed; a.lambda = b->lambda;
    a.linvt = b->linvt; a.u = b->u; a.V = b->V; a.S = b->S; a.rhs = b->rhs; a.dvec = b->dvec; a.gm = b->gm; a.status = b->status;
    a.n_lead = L.n_lead; a.n_trail = L.n_trail; a.n_ent = L.n_ent; a.trail_off = L.trail_off;
    a.stop = d_stop;
    const int epb = prep_epb(L.tb);
    const int64_t ent_chunks = (L.n_ent + epb - 1) / epb, row_chunks = std::max<int64_t>(1, (L.n_lead + PREP_RPB - 1) / PREP_RPB);
    const int64_t nbt = (L.n_lead + 31) / 32;
    a.ent_chunks = (int32_t)ent_chunks;
    a.trail_blocks = (int32_t)(ent_chunks * row_chunks);
    const int64_t grid = ent_chunks * row_chunks + nbt * nbt;
    if (grid <= 0) return PCS_OK;
    if (grid > INT32_MAX) return fail(PCS_ERR_ARG, "pcs_lm_trial_build: the system is too large for one launch of the Schur preparation");
    if (L.tb == 6) hipLaunchKernelGGL(schur_prep_kernel<6>, dim3((unsigned)grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(schur_prep_kernel<3>, dim3((unsigned)grid), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    return PCS_OK;
}

// what schur_finish_kernel prepares for the build that follows it, beyond the step itself: the hand-fused engines' slabs and point copy
// (a generated chain has its own preparation: all of it empty)
struct FinishSlabs {
    double *cam_slab = nullptr, *pose_slab = nullptr, *points = nullptr;
    int64_t n_cams = 0, n_imgs = 0, n_keys = 0, extr_off = 0, pose_off = 0, point_off = 0;
    bool has_pose = false, copy_points = false;
};
static int enqueue_schur_finish_fused(const BlockLayout &L, int64_t n_params, int n_cu, const FinishSlabs &fs, const pcs_lm_buffers *b, hipStream_t s, const int32_t *d_stop,
                                      const int32_t *d_sel, int64_t alt_pk, int64_t n_packed) {
    SchurFinishArgs a{};
    a.V = b->V; a.xl = b->xlead; a.n_lead = (int32_t)L.n_lead; a.n_trail = (int32_t)L.n_trail; a.ldv = (int32_t)std::max<int64_t>(1, L.n_trail);
    a.linvt = b->linvt; a.u = b->u; a.fixed = b->fixed; a.delta = b->delta; a.ps_in = b->ps[0]; a.ps_out = b->ps[1];
    a.n_ent = L.n_ent; a.trail_off = L.trail_off;
    a.stop = d_stop; a.sel = d_sel;
    a.vote = (b->mode & PCS_LM_VOTES) ? b->packed[1] + n_packed : nullptr; a.vote_alt = -alt_pk; a.status = b->status;
    a.cam_slab = fs.cam_slab; a.pose_slab = fs.pose_slab; a.points = fs.points;
    a.n_cams = (int32_t)fs.n_cams; a.n_imgs = (int32_t)fs.n_imgs; a.n_keys = (int32_t)fs.n_keys;
    a.has_pose = fs.has_pose; a.copy_points = fs.copy_points;
    a.extr_off = fs.extr_off; a.pose_off = fs.pose_off; a.point_off = fs.point_off;
    a.Hm = b->packed[1]; a.n_h = L.a_len() + L.b_len() + L.c_len(); a.g = a.Hm + a.n_h; a.n_g = n_params; a.cost = a.g + n_params; a.alt_out = -alt_pk;
    const int ecb = finish_ecb(L.tb);
    const int64_t w_blocks = (L.n_ent + ecb - 1) / ecb;
    const bool lead_poses = a.has_pose && fs.pose_off < L.trail_off;
    const int64_t lead_threads = std::max<int64_t>(L.n_lead, fs.n_cams * CAM_STRIDE + (lead_poses ? fs.n_imgs * POSE_STRIDE : 0));
    const int64_t lead_blocks = std::max<int64_t>(1, (lead_threads + 1023) / 1024);
    const int64_t zero_blocks = std::min<int64_t>((a.n_h / 2 + 1023) / 1024 + 1, (int64_t)n_cu * 4);
    a.w_blocks = (int32_t)w_blocks; a.lead_blocks = (int32_t)lead_blocks;
    const int64_t grid = w_blocks + lead_blocks + zero_blocks;
    if (grid > INT32_MAX) return fail(PCS_ERR_ARG, "pcs_lm_trial_build: the system is too large for one launch of the step's completion");
    if (L.tb == 6) hipLaunchKernelGGL(schur_finish_kernel<6>, dim3((unsigned)grid), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(schur_finish_kernel<3>, dim3((unsigned)grid), dim3(1024), 0, s, a);
    HIPCHK(hipGetLastError());
    return PCS_OK;
}
static int enqueue_schur_finish_fused(pcs_engine *h, const pcs_lm_buffers *b, hipStream_t s, const int32_t *d_stop, const int32_t *d_sel, int64_t alt_pk,
                                      int64_t n_packed) {
    FinishSlabs fs;
    fs.cam_slab = (double *)h->d_cam_slab; fs.pose_slab = (double *)h->d_pose_slab; fs.points = (double *)h->d_points;
    fs.n_cams = h->n_cams; fs.n_imgs = h->n_imgs; fs.n_keys = h->n_keys;
    fs.has_pose = h->chain != PCS_CHAIN_FREE; fs.copy_points = h->chain != PCS_CHAIN_TEMPLATE;
    fs.extr_off = h->extr_off; fs.pose_off = h->pose_off; fs.point_off = h->point_off;
    return enqueue_schur_finish_fused(block_layout(h), h->n_params, h->n_cu, fs, b, s, d_stop, d_sel, alt_pk, n_packed);
}

// One whole Levenberg-Marquardt trial in two halves (round 5; pcs_lm_trial = both): BUILD = the damped Schur step from the current state at
// *lambda (+ the trial parameter string) and the normal equations at the trial string into the other state's packed buffer; FINISH = the
// decision INCLUDING the loop's termination rules, the state flip of an accepted trial and the read-back of the twelve numbers the host
// follows the loop with.  A sharded loop puts its all-reduce of the trial state between the two, on the same stream.  Every kernel starts
// with PCS_STOP_GUARD on flags[0], so the host may queue trial t + 1 before it has read the verdict of trial t.
static int lm_check(const void *h, const pcs_lm_buffers *b, const char *who) {
    if (!h || !b || !b->packed[0] || !b->packed[1] || !b->ps[0] || !b->ps[1] || !b->flags || !b->fixed || !b->lambda || !b->linvt || !b->u || !b->V || !b->S || !b->rhs ||
        !b->dvec || !b->gm || !b->status || !b->xlead || !b->w || !b->spd_work || !b->delta || !b->ctrl || !b->stats)
        return fail(PCS_ERR_ARG, "%s: bad arguments", who);
    if (reinterpret_cast<uintptr_t>(b->packed[0]) % 16 || reinterpret_cast<uintptr_t>(b->packed[1]) % 16) return fail(PCS_ERR_ARG, "%s: the packed buffers must be 16-byte aligned", who);
    if (b->mode & ~(PCS_LM_FIXED_TRIAL_BUFFER | PCS_LM_VOTES)) return fail(PCS_ERR_ARG, "%s: unknown mode bits", who);
    return PCS_OK;
}

int pcs_lm_trial_build(pcs_engine *h, const pcs_lm_buffers *b, void *stream) {
    int rc = lm_check(h, b, "pcs_lm_trial_build");
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    const BlockLayout L = block_layout(h);
    const int32_t *stop = b->flags, *sel = b->flags + 2;
    const int64_t alt_pk = b->packed[1] - b->packed[0], alt_ps = b->ps[1] - b->ps[0];   // doubles from state 0 to state 1
    // the one-launch Cholesky wants its hand-over workspace at the fill value: schur_trail_lead_kernel sets it on the way (one launch fewer)
    const bool prefill = L.n_lead > 0 && dense_spd_is_one_launch(h->device, L.n_lead, b->spd_algorithm);
    // fused: the two launches in front of the matrix products as one, the three behind the dense solve as one (csrc/ba_lm_fused.hpp); they
    // need every trailing entity to have leading rows to ride on and the normal equations' prologue to be theirs to replace
    const bool fused = h->fused_trial && L.n_lead > 0 && L.n_ent > 0 && h->n > 0;
    if (fused) rc = enqueue_schur_prep_fused(L, b, s, stop, sel, alt_pk, prefill ? b->spd_work : nullptr, prefill ? cp_work_doubles((L.n_lead + 31) / 32) : 0);
    else rc = enqueue_schur_prepare(h, b->packed[0], b->fixed, b->lambda, b->linvt, b->u, b->V, b->S, b->rhs, b->dvec, b->gm, b->status, s, stop,
                                    prefill ? b->spd_work : nullptr, prefill ? cp_work_doubles((L.n_lead + 31) / 32) : 0, sel, alt_pk);
    if (rc) return rc;
    const int64_t ldv = std::max<int64_t>(1, L.n_trail);
    if (L.n_trail > 0 && L.n_lead > 0) {
        if (h->deterministic && !b->syrk_work && syrk_work_doubles(L.n_lead, L.n_trail) > 0)
            return fail(PCS_ERR_ARG, "pcs_lm_trial_build: deterministic mode needs pcs_lm_buffers.syrk_work (pcs_schur_syrk_work_len doubles)");
        rc = enqueue_schur_syrk(L.n_lead, L.n_trail, b->V, ldv, b->S, L.n_lead, b->u, b->rhs, s, stop, h->deterministic ? b->syrk_work : nullptr, b->syrk_work_len);
        if (rc) return rc;
    }
    const int64_t n_packed = L.a_len() + L.b_len() + L.c_len() + h->n_params + 1;
    const double *w = b->u;
    if (L.n_lead > 0) {
        rc = enqueue_dense_spd(h->device, L.n_lead, b->S, L.n_lead, b->rhs, b->xlead, b->spd_work, b->status, s, b->spd_algorithm, stop, prefill, h->spd_timeout_us);
        if (rc) return rc;
        if (L.n_trail > 0 && !fused) {
            launch_schur_vtx(b->V, b->xlead, b->w, (int)L.n_lead, (int)L.n_trail, (int)ldv, stop, s);
            HIPCHK(hipGetLastError());
            w = b->w;
        }
    }
    double *g_new = b->packed[1] + L.a_len() + L.b_len() + L.c_len();
    if (fused) {
        // w = V' x_l, the back substitution, the step, the trial string, this rank's vote, the slabs at the trial string and the zeroed trial state
        rc = enqueue_schur_finish_fused(h, b, s, stop, sel, alt_pk, n_packed);
        if (rc) return rc;
        return enqueue_normal(h, b->ps[1], b->packed[1], g_new, g_new + h->n_params, s, true, stop, sel, -alt_ps, -alt_pk, true);
    }
    // the step, the trial string (ps[1] while state 0 is current) and this rank's vote behind the TRIAL state's packed buffer
    rc = enqueue_schur_finish(h, b->linvt, b->u, w, b->xlead, b->fixed, b->delta, b->ps[0], b->ps[1], s, stop, sel,
                              (b->mode & PCS_LM_VOTES) ? b->packed[1] + n_packed : nullptr, -alt_pk, b->status);
    if (rc) return rc;
    if (h->n == 0) {
        // a rank whose observation shard is empty (ceil(N / world) rows per rank can leave the last ranks without any) contributes zeros
        // to the all-reduce of the trial state; the vote word behind it stays
        if (!(b->mode & PCS_LM_FIXED_TRIAL_BUFFER)) return fail(PCS_ERR_STATE, "no detections set");
        HIPCHK(hipMemsetAsync(b->packed[1], 0, sizeof(double) * (size_t)n_packed, s));
        return PCS_OK;
    }
    return enqueue_normal(h, b->ps[1], b->packed[1], g_new, g_new + h->n_params, s, true, stop, sel, -alt_ps, -alt_pk);
}

// The second half of a trial for a state of n_packed doubles ([blocks | g | cost]; pcs_engine and pcs_genchain alike).
static int enqueue_lm_finish(int n_cu, int64_t n_params, int64_t n_packed, const pcs_lm_buffers *b, hipStream_t s) {
    const bool fixed_buffer = (b->mode & PCS_LM_FIXED_TRIAL_BUFFER) != 0;
    LmDecideArgs a{};
    a.tail[0] = b->packed[0] + n_packed - 1; a.tail[1] = b->packed[1] + n_packed - 1;
    a.ps2[0] = b->ps[0]; a.ps2[1] = b->ps[1];
    a.sel = b->flags + 2;
    a.dvec = b->dvec; a.gm = b->gm; a.delta = b->delta; a.fixed = b->fixed; a.status = b->status; a.lambda = b->lambda; a.stats = b->stats;
    a.n_params = n_params;
    a.ctrl = b->ctrl; a.stop_flag = b->flags; a.accept_flag = b->flags + 1;
    a.use_votes = (b->mode & PCS_LM_VOTES) ? 1 : 0;
    a.keep_sel = fixed_buffer ? 1 : 0;
    if (b->result_host && b->free_idx && b->n_free > 0) {   // the final state straight into the host's mapped buffer when this trial ends the loop
        double *result_mapped = nullptr;
        if (hipHostGetDevicePointer(reinterpret_cast<void **>(&result_mapped), b->result_host, 0) == hipSuccess && result_mapped) {
            a.free_idx = b->free_idx; a.n_free = b->n_free; a.result = result_mapped;
        } else {
            (void)hipGetLastError();
        }
    }
    // the read-back: lm_decide_kernel writes the twelve numbers straight into the page-locked buffer when the device can address it (no
    // copy launch); a buffer that is not mapped gets an asynchronous copy
    double *stats_mapped = nullptr;
    if (b->stats_host && hipHostGetDevicePointer(reinterpret_cast<void **>(&stats_mapped), b->stats_host, 0) != hipSuccess) {
        (void)hipGetLastError();
        stats_mapped = nullptr;
    }
    a.stats_host = stats_mapped;
    hipLaunchKernelGGL(lm_decide_kernel, dim3(1), dim3(1024), 0, s, a);
    HIPCHK(hipGetLastError());
    if (b->stats_host && !stats_mapped) HIPCHK(hipMemcpyAsync(b->stats_host, b->stats, sizeof(double) * LM_STATS, hipMemcpyDeviceToHost, s));
    if (fixed_buffer) {   // the trial state sits in a fixed buffer (the one a sharded loop's all-reduce was queued on) — an accepted one is copied over the current state
        const int copy_blocks = (int)std::min<int64_t>((n_packed / 2 + 255) / 256 + 1, (int64_t)n_cu * 8);
        hipLaunchKernelGGL(lm_accept_kernel, dim3((unsigned)copy_blocks), dim3(256), 0, s, (const int32_t *)(b->flags + 1), (const double *)b->packed[1], b->packed[0], n_packed,
                           (const double *)b->ps[1], b->ps[0], n_params);
        HIPCHK(hipGetLastError());
    }
    return PCS_OK;
}

int pcs_lm_trial_finish(pcs_engine *h, const pcs_lm_buffers *b, void *stream) {
    int rc = lm_check(h, b, "pcs_lm_trial_finish");
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    const BlockLayout L = block_layout(h);
    return enqueue_lm_finish(h->n_cu, h->n_params, L.a_len() + L.b_len() + L.c_len() + h->n_params + 1, b, stream ? (hipStream_t)stream : h->stream);
}

int pcs_lm_trial(pcs_engine *h, const pcs_lm_buffers *b, void *stream) {
    const int rc = pcs_lm_trial_build(h, b, stream);
    return rc ? rc : pcs_lm_trial_finish(h, b, stream);
}

int pcs_normal_equations(pcs_engine *h, const double *param_str, double *H, double *g, double *cost) {
    if (!h || !param_str || !H || !g || !cost) return fail(PCS_ERR_ARG, "pcs_normal_equations: bad arguments");
    if (h->n <= 0) return fail(PCS_ERR_STATE, "no detections set");
    if (h->n_params > PCS_NORMAL_MAX_PARAMS)  // the same limit as enqueue_normal, checked BEFORE the scratch allocation
        return fail(PCS_ERR_ARG, "pcs_normal_equations: %lld parameters make a dense J^T J of %.1f GB (limit %d parameters: 32-bit offsets); use pcs_matfree",
                    (long long)h->n_params, (double)h->n_params * (double)h->n_params * 8e-9, PCS_NORMAL_MAX_PARAMS);
    HIPCHK(hipSetDevice(h->device));
    const int64_t need = h->n_params * h->n_params + h->n_params + 1;  // H | g | cost in one scratch buffer
    if (h->H_capacity < need) {
        if (h->d_H) HIPCHK(hipFree(h->d_H));
        h->d_H = nullptr;
        h->H_capacity = 0;
        HIPCHK(hipMalloc(&h->d_H, sizeof(double) * need));
        h->H_capacity = need;
    }
    double *d_g = h->d_H + h->n_params * h->n_params, *d_cost = d_g + h->n_params;
    int rc = pcs_normal_equations_device(h, param_str, h->d_H, d_g, d_cost, nullptr);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(H, h->d_H, sizeof(double) * h->n_params * h->n_params, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(g, d_g, sizeof(double) * h->n_params, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(cost, d_cost, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PCS_OK;
}

int pcs_synchronize(pcs_engine *h, void *stream) {
    if (!h) return fail(PCS_ERR_ARG, "pcs_synchronize: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(stream ? (hipStream_t)stream : h->stream));
    return PCS_OK;
}

int pcs_last_kernel_ms(pcs_engine *h, float *slab_prep_ms, float *eval_ms) {
    if (!h) return fail(PCS_ERR_ARG, "pcs_last_kernel_ms: bad arguments");
    if (!h->events_valid) return fail(PCS_ERR_STATE, "no evaluation has been queued yet");
    float a = 0, b = 0;
    int rc = ring_read(h, (h->ev_count - 1) % h->ev_ring, &a, &b);
    if (rc) return rc;
    if (slab_prep_ms) *slab_prep_ms = a;
    if (eval_ms) *eval_ms = b;
    return PCS_OK;
}

int pcs_kernel_ms_mean(pcs_engine *h, int64_t *count, float *slab_prep_ms, float *eval_ms) {
    if (!h) return fail(PCS_ERR_ARG, "pcs_kernel_ms_mean: bad arguments");
    if (!h->events_valid) return fail(PCS_ERR_STATE, "no evaluation has been queued yet");
    const int64_t n = std::min<int64_t>(h->ev_count, h->ev_ring);
    double sa = 0, sb = 0;
    for (int64_t i = 0; i < n; ++i) {
        float a = 0, b = 0;
        int rc = ring_read(h, i, &a, &b);
        if (rc) return rc;
        sa += a;
        sb += b;
    }
    if (count) *count = n;
    if (slab_prep_ms) *slab_prep_ms = (float)(sa / n);
    if (eval_ms) *eval_ms = (float)(sb / n);
    return PCS_OK;
}

int pcs_kernel_ms_samples(pcs_engine *h, int64_t capacity, float *slab_prep_ms, float *eval_ms, int64_t *count) {
    if (!h || capacity < 0 || !count || (capacity > 0 && (!slab_prep_ms || !eval_ms))) return fail(PCS_ERR_ARG, "pcs_kernel_ms_samples: bad arguments");
    if (!h->events_valid) return fail(PCS_ERR_STATE, "no evaluation has been queued yet");
    const int64_t have = std::min<int64_t>(h->ev_count, h->ev_ring), n = std::min<int64_t>(have, capacity);
    const int64_t first = h->ev_count - n;   // the n most recent evaluations, oldest first
    for (int64_t i = 0; i < n; ++i) {
        int rc = ring_read(h, (first + i) % h->ev_ring, slab_prep_ms + i, eval_ms + i);
        if (rc) return rc;
    }
    *count = n;
    return PCS_OK;
}

int pcs_normal_descriptors(int chain, int pass, int trail_group, int32_t *out) {
    if (!out || chain < 0 || chain > 2 || pass < 0 || pass > 1 || (pass == PASS_CAMKEY && chain == CHAIN_TEMPLATE))
        return fail(PCS_ERR_ARG, "pcs_normal_descriptors: bad arguments (passes 0 and 1 of ba_normal_mfma_kernel only)");
    if (pass == PASS_SHARED) {
        if (chain == CHAIN_TEMPLATE) fill_descriptors<CHAIN_TEMPLATE, PASS_SHARED>(trail_group, out);
        else if (chain == CHAIN_SELF) fill_descriptors<CHAIN_SELF, PASS_SHARED>(trail_group, out);
        else fill_descriptors<CHAIN_FREE, PASS_SHARED>(trail_group, out);
    } else {
        if (chain == CHAIN_SELF) fill_descriptors<CHAIN_SELF, PASS_CAMKEY>(trail_group, out);
        else fill_descriptors<CHAIN_FREE, PASS_CAMKEY>(trail_group, out);
    }
    return PCS_OK;
}

int pcs_normal_entry_map(int chain, int pass, int32_t *out) {
    if (!out || chain < 0 || chain > 2 || pass < 0 || pass > 2) return fail(PCS_ERR_ARG, "pcs_normal_entry_map: bad arguments");
    if ((pass != PASS_SHARED && chain == CHAIN_TEMPLATE) || (pass == PASS_IMGKEY && chain != CHAIN_SELF))
        return fail(PCS_ERR_ARG, "pcs_normal_entry_map: chain %d has no pass %d", chain, pass);
    if (pass == PASS_SHARED) {
        if (chain == CHAIN_TEMPLATE) fill_entry_map<CHAIN_TEMPLATE, PASS_SHARED>(out);
        else if (chain == CHAIN_SELF) fill_entry_map<CHAIN_SELF, PASS_SHARED>(out);
        else fill_entry_map<CHAIN_FREE, PASS_SHARED>(out);
    } else if (pass == PASS_CAMKEY) {
        if (chain == CHAIN_SELF) fill_entry_map<CHAIN_SELF, PASS_CAMKEY>(out);
        else fill_entry_map<CHAIN_FREE, PASS_CAMKEY>(out);
    } else {
        // ba_normal_imgkey_kernel: register r of lane l = entry (a, b), a <= b, of the run's 3 x 3 sum G = S^T S over the
        // pose-translation columns 18..20 (column l & 15 in the order 00 01 02 11 12 22), for local run (l >> 4) + 4 r
        static const int ga[IK_COLS] = {0, 0, 0, 1, 1, 2}, gb[IK_COLS] = {0, 1, 2, 1, 2, 2};
        for (int m = 0; m < 2; ++m)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int c = lane & 15;
                    int32_t *o = out + ((m * 64 + lane) * 4 + r) * 2;
                    o[0] = (m == 0 && c < IK_COLS) ? 18 + ga[c] : -1;
                    o[1] = (m == 0 && c < IK_COLS) ? 18 + gb[c] : -1;
                }
    }
    return PCS_OK;
}

}  // extern "C"

#include "pcs_genchain.inc"
