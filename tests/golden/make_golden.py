#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/*.npz from the reference itself.

TEST INFRASTRUCTURE (build container only — /root/reference does not exist on the GPU box and
nothing under tests/ -m gpu, smoke() or bench.py runs this).

What it does: copies /root/reference/pyCamSet into a *temporary* directory (the reference
code-generates ``template_functions/*.py`` beside its own ``__file__``,
abstract_function_blocks.py:298-299, :394; matmul_map.py:250-251, and /root/reference is
read-only), installs the stand-in modules of ``_refstubs.py`` (numba is absent here; njit becomes
the identity so the reference bodies run as IEEE-754 CPython), runs the reference's own
function blocks / ``optimisation_function`` / bundle handlers on seeded synthetic problems from
``pycamset_amd.synthetic`` and stores inputs + outputs.  The temp copy and the files the
reference generates there are deleted on exit; no reference source enters this repo.

Run:  python tests/golden/make_golden.py     (about a minute)
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(HERE))

os.environ.setdefault("MPLBACKEND", "Agg")

from pycamset_amd import synthetic  # noqa: E402


class _DuckCamset:
    """get_names()/get_n_cams() are all the handlers touch (template_handler.py:116-125)."""

    def __init__(self, n):
        self._names = [f"cam_{i}" for i in range(n)]

    def get_names(self):
        return list(self._names)

    def get_n_cams(self):
        return len(self._names)


class _DuckTarget:
    """point_data is all the handlers touch (template_handler.py:120-121, :160-163)."""

    def __init__(self, points):
        self.point_data = np.array(points, dtype=np.float64)[None]  # (1, K, 3) like ChArUco
        self.square_size = 1.0
        self.valid_map = None


def unit_vectors(fb, ch):
    """Known-answer vectors for SURVEY 8a rows a1-a7, including the branch / edge inputs."""
    rng = np.random.default_rng(7)
    rvecs = [
        np.zeros(3), np.array([1e-11, 0.0, 0.0]), np.array([3e-11, -4e-11, 1e-11]),
        np.array([1e-9, 2e-9, -1e-9]), np.array([1e-5, -2e-5, 3e-5]),
        np.array([np.pi, 0.0, 0.0]), np.array([0.0, np.pi - 1e-9, 0.0]),
        np.array([1.2, -2.0, 2.1]), np.array([0.0, 0.0, 1.0]),
    ] + [rng.normal(0, s, 3) for s in (0.01, 0.1, 0.5, 1.0, 2.0, 3.0)]
    rvecs = np.array(rvecs)
    rod = np.empty((len(rvecs), 9))
    rodj = np.empty((len(rvecs), 27))
    for i, r in enumerate(rvecs):
        ch.numba_flat_rodrigues_INPLACE(r.copy(), rod[i])
        ch.numba_rodrigues_jac(r.copy(), rodj[i])

    n = 40
    pose6 = np.concatenate([rng.normal(0, 0.6, (n, 3)), rng.normal(0, 0.2, (n, 3))], axis=1)
    pose6[0, :3] = 0
    pose6[1, :3] = [2e-11, 0, 0]
    pts = rng.normal(0, 0.1, (n, 3))
    e4 = np.empty((n, 12))
    ht = np.empty((n, 3))
    rig_fun = np.empty((n, 3))
    rig_jac = np.empty((n, 27))
    tmp_jac = np.empty((n, 18))
    for i in range(n):
        ch.n_e4x4_flat_INPLACE(pose6[i].copy(), e4[i])
        ch.n_htform_prealloc(pts[i].copy(), e4[i].copy(), out=ht[i])
        mem = np.empty(27)
        fb.rigidTform3d.compute_fun(pose6[i].copy(), pts[i].copy(), rig_fun[i], mem)
        fb.rigidTform3d.compute_jac(pose6[i].copy(), pts[i].copy(), rig_jac[i], np.empty(27))
        out18 = np.empty(18)
        fb.template_points.compute_jac(pose6[i].copy(), pts[i].copy(), out18, np.empty(27))
        tmp_jac[i] = out18

    intr = np.empty((n, 9))
    intr[:, 0] = rng.uniform(900, 1100, n)
    intr[:, 1] = rng.uniform(480, 520, n)
    intr[:, 2] = rng.uniform(900, 1100, n)
    intr[:, 3] = rng.uniform(480, 520, n)
    intr[:, 4] = rng.normal(0, 0.05, n)
    intr[:, 5] = rng.normal(0, 0.01, n)
    intr[:, 6:8] = rng.normal(0, 1e-3, (n, 2))
    intr[:, 8] = rng.normal(0, 1e-3, n)
    xc = np.stack([rng.normal(0, 0.05, n), rng.normal(0, 0.05, n), rng.uniform(0.1, 0.4, n)], axis=1)
    xc[0] = [0.3, -0.2, 0.011]     # small z: large normalised radius
    xc[1] = [0.0, 0.0, 0.25]       # on the axis
    xc[2] = [0.02, 0.01, -0.3]     # behind the camera: formulas still evaluate
    xc[3] = [1.0, 1.0, 1.0]        # all-ones point used by abstract_function_block.test_self (afb:750-775)
    intr[3] = 1.0
    pj_fun = np.empty((n, 2))
    pj_jac = np.empty((n, 24))
    for i in range(n):
        fb.projection.compute_fun(intr[i].copy(), xc[i].copy(), pj_fun[i], np.empty(1))
        fb.projection.compute_jac(intr[i].copy(), xc[i].copy(), pj_jac[i], np.empty(1))
    fp_fun = np.empty(3)
    fp_jac = np.empty(9)
    fb.free_point.compute_fun(np.array([0.1, -0.2, 0.3]), np.empty(0), fp_fun, np.empty(0))
    fb.free_point.compute_jac(np.array([0.1, -0.2, 0.3]), np.empty(0), fp_jac, np.empty(0))
    return dict(rvecs=rvecs, rodrigues=rod, rodrigues_jac=rodj, pose6=pose6, pts=pts, e4x4=e4, htform=ht,
                rigid_fun=rig_fun, rigid_jac=rig_jac, template_jac=tmp_jac,
                intr=intr, xc=xc, proj_fun=pj_fun, proj_jac=pj_jac, free_fun=fp_fun, free_jac=fp_jac)


def chain_blocks(fb, chain):
    if chain == "template":
        return fb.projection() + fb.extrinsic3D() + fb.template_points()
    if chain == "self":
        return fb.projection() + fb.extrinsic3D() + fb.rigidTform3d() + fb.free_point()
    if chain == "free":
        return fb.projection() + fb.extrinsic3D() + fb.free_point()
    raise ValueError(chain)


def chain_slabs(rig, chain):
    if chain == "template":
        return [rig.intr, rig.extr, rig.poses]
    if chain == "self":
        return [rig.intr, rig.extr, rig.poses, rig.points]
    # free-point chain: the points live directly in the world frame; reuse image 0's frame.
    return [rig.intr, rig.extr, rig.points]


def block_level(fb, rig, chain, threads_list=(1, 3), seed=0):
    """optimisation_function API (abstract_function_blocks.py:656-681) on a synthetic rig."""
    rng = np.random.default_rng(seed)
    out = {}
    dd = rig.detections
    template = rig.points if chain == "template" else None
    op = chain_blocks(fb, chain)
    param_str = op.build_param_list(*chain_slabs(rig, chain))
    out["detections"] = dd
    out["param_str"] = param_str
    out["points"] = rig.points
    out["intr"], out["extr"], out["poses"] = rig.intr, rig.extr, rig.poses
    n_par = param_str.shape[0]
    unfixed = rng.random(n_par) > 0.25
    unfixed[:3] = [True, False, True]
    out["unfixed"] = unfixed
    for t in threads_list:
        loss = op.make_full_loss_fn(dd, t)
        out[f"resid_t{t}"] = np.array(loss(param_str, template)) if template is not None else np.array(loss(param_str))
        jac_all = op.make_jacobean(dd, t)
        d, c, rp = jac_all(param_str, template) if template is not None else jac_all(param_str)
        out[f"data_all_t{t}"], out[f"indices_all_t{t}"], out[f"indptr_all_t{t}"] = np.array(d), np.array(c), np.array(rp)
        jac_m = op.make_jacobean(dd, t, unfixed_params=unfixed)
        d, c, rp = jac_m(param_str, template) if template is not None else jac_m(param_str)
        out[f"data_masked_t{t}"], out[f"indices_masked_t{t}"], out[f"indptr_masked_t{t}"] = np.array(d), np.array(c), np.array(rp)
    bpi = op.get_block_param_inds(dd, 1, unthreaded=True)
    out["block_param_inds"] = np.array(bpi).astype(np.int64)
    return out


def handler_level(mods, rig, chain, fixed_cam_ext=True):
    """ParamHandler surface (template_handler.py:157-193, standard_bundle_handler.py:184-226,
    free_point_handler.py:145-186): x -> residual vector, csr_array."""
    th, sbh, fph, TargetDetection = mods
    camset = _DuckCamset(rig.n_cams)
    target = _DuckTarget(rig.points)
    det = TargetDetection(cam_names=camset.get_names(), data=rig.detections.copy())
    fixed = None
    if fixed_cam_ext:
        fixed = {"cam_0": {"ext": rig.extr[0].copy()}, "cam_1": {"int": rig.intr[1].copy()}}
    cls = {"template": th.TemplateBundleHandler, "self": sbh.SelfBundleHandler, "free": fph.FreePointBundleHandler}[chain]
    h = cls(camset, target, det, fixed_params=fixed, options={"verbosity": 0})
    bp = h.bundlePrimitive
    parts = [rig.intr[bp.intr_unfixed].ravel(), rig.extr[bp.extr_unfixed].ravel()]
    if chain != "free":
        parts.append(rig.poses[bp.poses_unfixed].ravel())
    if chain != "template":
        parts.append(rig.points.ravel()[bp.bdpt_unfixed])
    x = np.concatenate(parts)
    loss = h.make_loss_fun(2)
    jac = h.make_loss_jac(2)
    r = np.array(loss(x.copy()))
    J = jac(x.copy())
    out = dict(detections=rig.detections, points=rig.points, x=x, resid=r,
               data=np.array(J.data), indices=np.array(J.indices), indptr=np.array(J.indptr),
               shape=np.array(J.shape), intr0=rig.intr, extr0=rig.extr, poses0=rig.poses,
               intr_unfixed=np.array(bp.intr_unfixed), extr_unfixed=np.array(bp.extr_unfixed))
    if chain != "free":
        out["poses_unfixed"] = np.array(bp.poses_unfixed)
    if chain != "template":
        out["bdpt_unfixed"] = np.array(bp.bdpt_unfixed)
    if fixed_cam_ext:
        out["fixed_ext_cam0"] = rig.extr[0].copy()
        out["fixed_int_cam1"] = rig.intr[1].copy()
    return out


def legacy_cost_vectors(ch, rig):
    """the reference's legacy residual-only cost (compiled_helpers.py:493-549) on inputs
    assembled with the reference's own helpers (template_handler.py:231-237)."""
    C, I, K = rig.n_cams, rig.n_imgs, rig.n_keys
    im_points = np.empty((I, K, 3))
    for i in range(I):
        blank = np.zeros(12)
        ch.n_e4x4_flat_INPLACE(rig.poses[i].copy(), blank)
        ch.n_htform_broadcast_prealloc(rig.points.copy(), blank, im_points[i])
    Kc = np.zeros((C, 3, 3))
    Kc[:, 0, 0], Kc[:, 0, 2], Kc[:, 1, 1], Kc[:, 1, 2], Kc[:, 2, 2] = rig.intr[:, 0], rig.intr[:, 1], rig.intr[:, 2], rig.intr[:, 3], 1.0
    proj = np.empty((C, 3, 4))
    for c in range(C):
        blank = np.zeros(12)
        ch.n_e4x4_flat_INPLACE(rig.extr[c].copy(), blank)
        proj[c] = Kc[c] @ np.concatenate([blank[:9].reshape(3, 3), blank[9:].reshape(3, 1)], axis=1)
    dists = np.ascontiguousarray(rig.intr[:, 4:9])
    err = ch.numpy_bundle_adjustment_costfn(rig.detections, im_points, proj, Kc, dists)
    err_jit = ch.bundle_adjustment_costfn(rig.detections, im_points, proj, Kc, dists)
    n = (rig.n_det // 3) * 3
    err_par = ch.bundle_adj_parrallel_solver(rig.detections[:n].reshape(3, n // 3, 5), im_points, proj, Kc, dists)
    pts = rig.detections[:20, 3:].copy()
    dist_out = np.array([ch.nb_distort(p.copy(), Kc[0], dists[0]) for p in pts])
    return dict(detections=rig.detections, im_points=im_points, proj=proj, intrinsics=Kc, dists=dists,
                errors=np.array(err), errors_njit_alias=np.array(err_jit), errors_parallel=np.array(err_par),
                intr=rig.intr, extr=rig.extr, poses=rig.poses, points=rig.points, distort_in=pts, distort_out=dist_out)


def triangulation_vectors(ch, seed=9):
    """the reference's nb_triangulate_full (compiled_helpers.py:609-663) on observations of a
    synthetic rig, sorted by (image, key, camera) as the reference's front end requires."""
    rig = synthetic.make_rig("tri", 12, 3, synthetic.charuco_points(7, 6.0), seed=seed, visibility=0.55, n_rings=2, noise_px=0.3)
    C, I, K = rig.n_cams, rig.n_imgs, rig.n_keys
    Kc = np.zeros((C, 3, 3))
    it = rig.intr_true
    Kc[:, 0, 0], Kc[:, 0, 2], Kc[:, 1, 1], Kc[:, 1, 2], Kc[:, 2, 2] = it[:, 0], it[:, 1], it[:, 2], it[:, 3], 1.0
    proj = np.empty((C, 3, 4))
    for c in range(C):
        blank = np.zeros(12)
        ch.n_e4x4_flat_INPLACE(rig.extr_true[c].copy(), blank)
        proj[c] = Kc[c] @ np.concatenate([blank[:9].reshape(3, 3), blank[9:].reshape(3, 1)], axis=1)
    dists = np.ascontiguousarray(it[:, 4:9])
    d = rig.detections
    d = d[np.lexsort((d[:, 0], d[:, 2], d[:, 1]))]
    # grouping of CameraSet.multi_cam_triangulate (cameras/camera_set.py:371-378)
    _, inv, count = np.unique(d[:, 1:-2], axis=0, return_inverse=True, return_counts=True)
    rec = d[(count > 1)[inv].squeeze()]
    _, im_index, im_counts = np.unique(rec[:, 1:-2], axis=0, return_index=True, return_counts=True)
    start = np.append(0, np.cumsum(im_counts[np.argsort(im_index)]))
    pts = ch.nb_triangulate_full(rec, proj, start, Kc, dists)
    und = np.array([ch.nb_undistort(r[-2:].copy(), Kc[int(r[0])], dists[int(r[0])]) for r in rec[:40]])
    # ground truth for a sanity bound: pose-transformed template points
    truth = np.empty((I, K, 3))
    for i in range(I):
        blank = np.zeros(12)
        ch.n_e4x4_flat_INPLACE(rig.poses_true[i].copy(), blank)
        ch.n_htform_broadcast_prealloc(rig.points.copy(), blank, truth[i])
    return dict(data=rec, start_inds=start, proj=proj, intrinsics=Kc, dists=dists, points=np.array(pts),
                undistorted_first40=und, unsorted_detections=rig.detections, truth=truth)


class _DuckTargetND:
    """Ccube-shaped target: point_data (6, (n-1)^2, 3) (target_Ccube.py:227-244) — multi-dimensional keys."""

    def __init__(self, points, shape):
        self.point_data = np.array(points, dtype=np.float64).reshape(tuple(shape) + (3,))
        self.square_size = 1.0
        self.valid_map = None


def to_multidim_keys(det5, keydims):
    """(cam, im, flat key, u, v) -> (cam, im, k0, k1, ..., u, v): the table a Ccube detection produces."""
    idx = np.unravel_index(det5[:, 2].astype(np.int64), tuple(keydims))
    return np.concatenate([det5[:, :2]] + [np.asarray(i, dtype=np.float64)[:, None] for i in idx] + [det5[:, 3:]], axis=1)


def flatten_vectors(TargetDetection):
    """TargetDetection.return_flattened_keys (target_detections.py:333-351) run by the reference on
    multi-dimensional key tables."""
    rng = np.random.default_rng(15)
    out = {}
    for tag, dims, n in (("ccube", (6, 81), 400), ("three_dim", (2, 3, 4), 60), ("one_dim", (9,), 30)):
        cols = [rng.integers(0, 3, n), rng.integers(0, 5, n)] + [rng.integers(0, d, n) for d in dims]
        data = np.concatenate([np.stack(cols, 1).astype(np.float64), rng.uniform(0, 1000, (n, 2))], axis=1)
        data[0, 2:-2] = [d - 1 for d in dims]   # the last key of the target
        data[1, 2:-2] = 0
        td = TargetDetection([f"cam_{i}" for i in range(3)], data, max_ims=7)
        flat = td.return_flattened_keys(dims)
        out[f"{tag}_in"], out[f"{tag}_dims"], out[f"{tag}_out"] = data, np.array(dims), np.array(flat.get_data())
        out[f"{tag}_max_ims"] = np.array(flat.max_ims)
    return out


def handler_case(mods, rig, chain, *, keydims=None, fixed=None, options=None, max_ims=0, n_cams=None):
    """One run of the reference's handler closures (th:157-193, sbh:184-226): inputs + residual + csr_array."""
    camset = _DuckCamset(n_cams or rig.n_cams)
    dets = rig.detections if keydims is None else to_multidim_keys(rig.detections, keydims)
    target = _DuckTarget(rig.points) if keydims is None else _DuckTargetND(rig.points, keydims)
    det = mods.TargetDetection(cam_names=camset.get_names(), data=dets.copy(), max_ims=max_ims)
    mods.th.DEFAULT_OPTIONS.update({"fixed_pose": 0})   # the reference mutates its module-level default dict (th:108-110)
    cls = {"template": mods.th.TemplateBundleHandler, "self": mods.sbh.SelfBundleHandler, "free": mods.fph.FreePointBundleHandler}[chain]
    opts = {"verbosity": 0}
    opts.update(options or {})
    h = cls(camset, target, det, fixed_params=fixed, options=opts)
    bp = h.bundlePrimitive
    has_pose = chain != "free"   # the free-point chain has no target poses (fph:47-100)
    C, I = bp.intr.shape[0], bp.poses.shape[0] if has_pose else 0
    intr = np.zeros((C, 9)); extr = np.zeros((C, 6)); poses = np.zeros((I, 6))
    intr[: rig.n_cams], extr[: rig.n_cams] = rig.intr, rig.extr
    if has_pose:
        poses[: rig.n_imgs] = rig.poses
    if C > rig.n_cams:   # cameras / images nobody observed: any finite value
        intr[rig.n_cams:], extr[rig.n_cams:] = rig.intr[:1], rig.extr[:1]
    if I > rig.n_imgs:
        poses[rig.n_imgs:] = 0.01
    parts = [intr[bp.intr_unfixed].ravel(), extr[bp.extr_unfixed].ravel()]
    if has_pose:
        parts.append(poses[bp.poses_unfixed].ravel())
    if chain != "template":
        parts.append(rig.points.ravel()[bp.bdpt_unfixed])
    x = np.concatenate(parts)
    loss, jac = h.make_loss_fun(2), h.make_loss_jac(2)
    r = np.array(loss(x.copy()))
    J = jac(x.copy())
    out = dict(detections=dets, points=rig.points, x=x, resid=r, data=np.array(J.data), indices=np.array(J.indices),
               indptr=np.array(J.indptr), shape=np.array(J.shape), intr_unfixed=np.array(bp.intr_unfixed),
               extr_unfixed=np.array(bp.extr_unfixed), intr_slab=np.array(bp.intr), extr_slab=np.array(bp.extr),
               n_cams=np.array(C), max_ims=np.array(max_ims),
               keydims=np.array(keydims if keydims is not None else (1, rig.n_keys)))
    if has_pose:
        out["poses_unfixed"] = np.array(bp.poses_unfixed)
        out["poses_slab"] = np.array(bp.poses)
    if chain != "template":
        out["bdpt_unfixed"] = np.array(bp.bdpt_unfixed)
    if chain == "self":
        out["visible_feature_mask"] = np.array(h.visible_feature_mask)
    return out


def round2_vectors(mods):
    """Fixtures asked for by the round-1 review: multi-dimensional keys (a15), Ccube-shaped handlers with N ~ 1e3,
    other fixed-pose / fixed-camera settings, tables whose last image / key / camera is unobserved (reference quirk
    ii, abstract_function_blocks.py:793-795), and a ~3000-detection block-level case."""
    np.savez_compressed(HERE / "flatten_keys.npz", **flatten_vectors(mods.TargetDetection))
    print("flatten_keys done")
    ccube = synthetic.make_rig("ccube-1k", 4, 6, synthetic.ccube_points(), seed=21, visibility=0.09)
    print("ccube-1k N =", ccube.n_det)
    both = {"cam_2": {"int": ccube.intr[2].copy(), "ext": ccube.extr[2].copy()}, "cam_0": {"ext": ccube.extr[0].copy()}}
    cases = {
        "handler_template_ccube": dict(chain="template", keydims=(6, 81), fixed=both, options={"fixed_pose": 3}),
        "handler_self_ccube": dict(chain="self", keydims=(6, 81), fixed=both, options={"fixed_pose": 3}),
        # fixed_pose=None: NumPy treats a None index as newaxis, so the reference fixes (and zeroes) EVERY pose (th:134-137)
        "handler_template_fixedpose_none": dict(chain="template", keydims=(6, 81), fixed=None, options={"fixed_pose": None}),
        # the free-point chain (no target poses; images are plain frames of a static scene) on the same table: fph:47-186
        "handler_free_ccube": dict(chain="free", keydims=(6, 81), fixed=both, options={}),
    }
    for name, kw in cases.items():
        res = handler_case(mods, ccube, **kw)
        np.savez_compressed(HERE / f"{name}.npz", **res)
        print(name, "J shape", res["shape"], "nnz", res["data"].shape)
    # quirk ii: trailing entities without detections
    small = synthetic.tiny_rig(seed=3, n_cams=3, n_imgs=5, n_keys=9, visibility=0.9)
    d = small.detections
    no_last_key = d[d[:, 2] != small.n_keys - 1]
    import copy
    rig_k = copy.copy(small); rig_k.detections = no_last_key
    quirk = {
        # last image unobserved, template chain: the unused pose trails the parameter string -> harmless
        "quirk_template_last_image_unobserved": dict(rig=small, chain="template", max_ims=small.n_imgs + 2),
        # last key unobserved, self chain: points trail the string and sbh:160-169 fixes the unseen feature -> harmless
        "quirk_self_last_key_unobserved": dict(rig=rig_k, chain="self"),
        # last image unobserved, self chain: the reference offsets the point block by 6*(max image index + 1) while the
        # handler's pose slab has max_ims rows -> the kernels read point coordinates out of the pose slab (a reference bug)
        "quirk_self_last_image_unobserved": dict(rig=small, chain="self", max_ims=small.n_imgs + 1),
        # last camera unobserved, template chain: extrinsic offset 9*(max cam index + 1) vs a 9*n_cams slab -> same bug class
        "quirk_template_last_cam_unobserved": dict(rig=small, chain="template", n_cams=small.n_cams + 1),
    }
    for name, kw in quirk.items():
        rig = kw.pop("rig")
        res = handler_case(mods, rig, **kw)
        np.savez_compressed(HERE / f"{name}.npz", **res)
        print(name, "J shape", res["shape"], "nnz", res["data"].shape)
    # block level at N ~ 3000 (4 cams, 20 images, 486 keys): residual + unmasked data + both structures
    large = synthetic.make_rig("large", 4, 20, synthetic.ccube_points(), seed=12, visibility=0.078)
    print("large N =", large.n_det)
    for chain in ("template", "self", "free"):
        res = block_level(mods.fb, large, chain, threads_list=(5,))
        del res["data_masked_t5"]    # = data_all[mask]; the masked structure is kept, the values would double the file
        np.savez_compressed(HERE / f"block_{chain}_large.npz", **res)
        print("block", chain, "large nnz", res["data_all_t5"].shape)


EDGE_RVECS = np.array([
    # the rotation vectors of unit_vectors(): every branch and every ill-conditioned region of ch:197-286
    [0.0, 0.0, 0.0], [1e-11, 0.0, 0.0], [3e-11, -4e-11, 1e-11], [1e-9, 2e-9, -1e-9], [1e-5, -2e-5, 3e-5],
    [np.pi, 0.0, 0.0], [0.0, np.pi - 1e-9, 0.0], [1.2, -2.0, 2.1], [0.0, 0.0, 1.0],
    # beyond pi, and the decades between the theta < 1e-10 branch and well-conditioned angles
    [2.5, -2.5, 1.0], [1e-7, -1e-7, 2e-7], [3e-4, 1e-4, -2e-4],
])


def edge_rotation_rig(seed=31):
    """Every camera extrinsic AND every target pose takes one of EDGE_RVECS as its rotation: 12 cameras x 12 images x
    6 points, all visible (N = 864).  Translations keep the camera-frame depth in [0.4, 0.8] whatever the rotation."""
    rng = np.random.default_rng(seed)
    n = EDGE_RVECS.shape[0]
    intr = np.empty((n, 9))
    intr[:, 0], intr[:, 2] = rng.uniform(900, 1100, n), rng.uniform(900, 1100, n)
    intr[:, 1], intr[:, 3] = rng.uniform(480, 520, n), rng.uniform(480, 520, n)
    intr[:, 4], intr[:, 5] = rng.normal(0, 0.05, n), rng.normal(0, 0.01, n)
    intr[:, 6:8], intr[:, 8] = rng.normal(0, 1e-3, (n, 2)), rng.normal(0, 1e-3, n)
    extr = np.concatenate([EDGE_RVECS, np.stack([rng.normal(0, 0.02, n), rng.normal(0, 0.02, n), rng.uniform(0.55, 0.65, n)], 1)], axis=1)
    poses = np.concatenate([EDGE_RVECS[::-1], rng.normal(0, 0.02, (n, 3))], axis=1)   # another pairing than cam i <-> image i
    points = rng.uniform(-0.05, 0.05, (6, 3))
    cam, im, key = np.meshgrid(np.arange(n), np.arange(n), np.arange(6), indexing="ij")
    det = np.stack([cam.ravel(), im.ravel(), key.ravel()], 1).astype(np.float64)
    det = np.concatenate([det, 500.0 + rng.normal(0, 60.0, (det.shape[0], 2))], axis=1)
    return synthetic.SyntheticRig("edge-rot", det, intr, extr, poses, points)


def nonfinite_focal_rig(seed=32):
    """fbi:32-35 multiplies by fx and divides by it again: a focal length of 0, NaN or inf poisons BOTH residual rows
    and every non-constant Jacobian entry of the camera's detections.  Cameras: fx = 0 | fy = NaN | fx = inf | fy = 0 | fine."""
    rig = synthetic.tiny_rig(seed=seed, n_cams=5, n_imgs=3, n_keys=7, visibility=1.0)
    rig.intr[0, 0] = 0.0
    rig.intr[1, 2] = np.nan
    rig.intr[2, 0] = np.inf
    rig.intr[3, 2] = 0.0
    return rig


def generic_chain(fb, names):
    op = getattr(fb, names[0])()
    for n in names[1:]:
        op = op + getattr(fb, n)()
    return op


GENERIC_CHAINS = {
    # chains that are NOT one of the three the reference's handlers build (afb:735-748 composes any of them); the slab a
    # block reads is named by its link type (afb:42-46) and its place among the blocks of that link type
    "proj_rigid_free": ("projection", "rigidTform3d", "free_point"),                       # one camera at the origin, moving scene
    "proj_extr_rigid_template": ("projection", "extrinsic3D", "rigidTform3d", "template_points"),   # two per-image transforms
    "proj_template": ("projection", "template_points"),                                    # a single fixed camera
    "proj_rigid_extr_free": ("projection", "rigidTform3d", "extrinsic3D", "free_point"),   # per-image BEFORE per-camera
}


def generic_block_level(fb, rig, names, seed=5):
    """The reference's code generator on a chain none of its handlers builds: residual, Jacobian (all columns and
    masked), structure, block_param_inds; the slabs each block consumes are stored in block order."""
    rng = np.random.default_rng(seed)
    op = generic_chain(fb, names)
    slabs = []
    seen_img = 0
    for n in names:
        if n == "projection":
            slabs.append(rig.intr)
        elif n == "extrinsic3D":
            slabs.append(rig.extr)
        elif n in ("rigidTform3d", "template_points"):
            # a second per-image group gets its own, different 6-vectors
            slabs.append(rig.poses if seen_img == 0 else np.concatenate([rig.poses[::-1, :3] * 0.5, rig.poses[:, 3:] * -0.3], axis=1))
            seen_img += 1
        elif n == "free_point":
            slabs.append(rig.points)
    template = rig.points if names[-1] == "template_points" else None
    param_str = op.build_param_list(*slabs)
    unfixed = rng.random(param_str.shape[0]) > 0.3
    call = (lambda f: f(param_str, template)) if template is not None else (lambda f: f(param_str))
    out = dict(detections=rig.detections, param_str=param_str, points=rig.points, blocks=np.array(names), unfixed=unfixed)
    for i, sl in enumerate(slabs):
        out[f"slab_{i}"] = np.array(sl)
    out["resid"] = np.array(call(op.make_full_loss_fn(rig.detections, 2)))
    d, c, rp = call(op.make_jacobean(rig.detections, 2))
    out["data_all"], out["indices_all"], out["indptr_all"] = np.array(d), np.array(c), np.array(rp)
    d, c, rp = call(op.make_jacobean(rig.detections, 2, unfixed_params=unfixed))
    out["data_masked"], out["indices_masked"], out["indptr_masked"] = np.array(d), np.array(c), np.array(rp)
    out["block_param_inds"] = np.array(op.get_block_param_inds(rig.detections, 1, unthreaded=True)).astype(np.int64)
    return out


def round3_vectors(mods):
    """Fixtures asked for by the round-2 review: the edge rotations of unit_vectors() as the poses AND extrinsics of a
    block-level problem per chain (so that they go through the HIP slab code, not only the oracle's), non-finite / zero
    focal lengths, and chains that are not one of the handlers' three (for the chain compiler)."""
    edge = edge_rotation_rig()
    for chain in ("template", "self", "free"):
        res = block_level(mods.fb, edge, chain, threads_list=(2,))
        np.savez_compressed(HERE / f"block_{chain}_edge_rot.npz", **res)
        print("block", chain, "edge_rot N", edge.n_det, "nnz", res["data_all_t2"].shape)
    bad = nonfinite_focal_rig()
    for chain in ("template", "self", "free"):
        with np.errstate(all="ignore"):
            res = block_level(mods.fb, bad, chain, threads_list=(1,))
        np.savez_compressed(HERE / f"block_{chain}_focal_nonfinite.npz", **res)
        print("block", chain, "focal_nonfinite: NaN residuals", int(np.isnan(res["resid_t1"]).sum()), "of", res["resid_t1"].size)
    small = synthetic.tiny_rig(seed=33, n_cams=3, n_imgs=5, n_keys=8, visibility=0.85)
    for tag, names in GENERIC_CHAINS.items():
        res = generic_block_level(mods.fb, small, names)
        np.savez_compressed(HERE / f"generic_{tag}.npz", **res)
        print("generic", tag, "P", res["block_param_inds"].shape[1], "nnz", res["data_all"].shape)


USER_CHAINS = {
    # chains with USER blocks (tests/golden/_user_blocks.py, written on the reference's ABC): the reference's extension point
    "user_cam_scale": ("projection", "cam_scale", "extrinsic3D", "template_points"),
    "user_division": ("division_projection", "extrinsic3D", "rigidTform3d", "free_point"),
}
USER_CHAINS_R5 = {
    # round 5: a user block as a TEMPLATED source (template = True on the last block: it receives template[key], afb:374-375)
    "user_board_flex": ("projection", "extrinsic3D", "rigidTform3d", "board_flex"),
}


def user_block_level(mods, ub, rig, names, seed=6):
    """The reference's code generator on a chain that contains user-written blocks: residual, Jacobian (all columns and masked),
    structure, block_param_inds; the slab each block consumes is stored in block order."""
    rng = np.random.default_rng(seed)
    blocks = [getattr(ub, n)() if hasattr(ub, n) else getattr(mods.fb, n)() for n in names]
    op = blocks[0]
    for b in blocks[1:]:
        op = op + b
    slabs = []
    for n in names:
        if n == "projection":
            slabs.append(rig.intr)
        elif n == "division_projection":
            slabs.append(np.concatenate([rig.intr[:, :4], rng.normal(0.0, 0.05, (rig.n_cams, 1))], axis=1))
        elif n == "cam_scale":
            slabs.append(rng.uniform(0.8, 1.2, (rig.n_cams, 1)))
        elif n == "extrinsic3D":
            slabs.append(rig.extr)
        elif n in ("rigidTform3d", "template_points"):
            slabs.append(rig.poses)
        elif n == "free_point":
            slabs.append(rig.points)
        elif n == "board_flex":   # per image [sx, sy, tx, ty, k]: a board that is nearly rigid
            slabs.append(np.concatenate([rng.uniform(0.97, 1.03, (rig.n_imgs, 2)), rng.normal(0.0, 2e-3, (rig.n_imgs, 2)), rng.normal(0.0, 0.5, (rig.n_imgs, 1))], axis=1))
    template = rig.points if getattr(blocks[-1], "template", False) else None
    param_str = op.build_param_list(*slabs)
    unfixed = rng.random(param_str.shape[0]) > 0.3
    call = (lambda f: f(param_str, template)) if template is not None else (lambda f: f(param_str))
    out = dict(detections=rig.detections, param_str=param_str, points=rig.points, blocks=np.array(names), unfixed=unfixed)
    for i, sl in enumerate(slabs):
        out[f"slab_{i}"] = np.array(sl)
    out["resid"] = np.array(call(op.make_full_loss_fn(rig.detections, 2)))
    d, c, rp = call(op.make_jacobean(rig.detections, 2))
    out["data_all"], out["indices_all"], out["indptr_all"] = np.array(d), np.array(c), np.array(rp)
    d, c, rp = call(op.make_jacobean(rig.detections, 2, unfixed_params=unfixed))
    out["data_masked"], out["indices_masked"], out["indptr_masked"] = np.array(d), np.array(c), np.array(rp)
    out["block_param_inds"] = np.array(op.get_block_param_inds(rig.detections, 1, unthreaded=True)).astype(np.int64)
    return out


def round5_vectors(mods):
    """A user block as a TEMPLATED source through the reference's generator (round-4 review, "what's missing" 1)."""
    import importlib

    ub = importlib.import_module("_user_blocks")
    small = synthetic.tiny_rig(seed=35, n_cams=3, n_imgs=5, n_keys=8, visibility=0.85)
    for tag, names in USER_CHAINS_R5.items():
        res = user_block_level(mods, ub, small, names, seed=7)
        np.savez_compressed(HERE / f"{tag}.npz", **res)
        print(tag, "P", res["block_param_inds"].shape[1], "nnz", res["data_all"].shape, "max |resid|", float(np.max(np.abs(res["resid"]))))


def round4_vectors(mods):
    """User-written blocks through the reference's generator (round-3 review: the extension point)."""
    import importlib

    ub = importlib.import_module("_user_blocks")     # tests/golden is on sys.path; the module imports the reference's ABC
    small = synthetic.tiny_rig(seed=34, n_cams=3, n_imgs=5, n_keys=8, visibility=0.85)
    for tag, names in USER_CHAINS.items():
        res = user_block_level(mods, ub, small, names)
        np.savez_compressed(HERE / f"{tag}.npz", **res)
        print(tag, "P", res["block_param_inds"].shape[1], "nnz", res["data_all"].shape, "max |resid|", float(np.max(np.abs(res["resid"]))))


def main():
    import argparse

    import _refload

    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["round1", "round2", "round3", "round4", "round5"], default=None, help="regenerate one group of fixtures only")
    args = ap.parse_args()
    with _refload.reference_modules() as mods:
        ch, fb, th, sbh, fph, TargetDetection = mods.ch, mods.fb, mods.th, mods.sbh, mods.fph, mods.TargetDetection
        if args.only in (None, "round5"):
            round5_vectors(mods)
        if args.only == "round5":
            return
        if args.only in (None, "round4"):
            round4_vectors(mods)
        if args.only == "round4":
            return
        if args.only in (None, "round3"):
            round3_vectors(mods)
        if args.only == "round3":
            return
        if args.only in (None, "round2"):
            round2_vectors(mods)
        if args.only == "round2":
            return
        np.savez_compressed(HERE / "unit_vectors.npz", **unit_vectors(fb, ch))
        print("unit vectors done")

        tiny = synthetic.tiny_rig(seed=0)
        medium = synthetic.make_rig("medium", 4, 20, synthetic.ccube_points(), seed=11, visibility=0.026)
        print("tiny N =", tiny.n_det, " medium N =", medium.n_det)
        for chain in ("template", "self", "free"):
            for tag, rig in (("tiny", tiny), ("medium", medium)):
                res = block_level(fb, rig, chain, threads_list=(1, 3) if tag == "tiny" else (4,))
                np.savez_compressed(HERE / f"block_{chain}_{tag}.npz", **res)
                print("block", chain, tag, "nnz_all", res[[k for k in res if k.startswith('data_all')][0]].shape)
        np.savez_compressed(HERE / "legacy_cost_medium.npz", **legacy_cost_vectors(ch, medium))
        print("legacy cost done")
        np.savez_compressed(HERE / "triangulation.npz", **triangulation_vectors(ch))
        print("triangulation done")
        mods4 = (th, sbh, fph, TargetDetection)
        for chain in ("template", "self", "free"):
            for tag, rig, fx in (("tiny", tiny, True), ("tiny_nofix", tiny, False)):
                th.DEFAULT_OPTIONS.update({"fixed_pose": 0})
                res = handler_level(mods4, rig, chain, fixed_cam_ext=fx)
                np.savez_compressed(HERE / f"handler_{chain}_{tag}.npz", **res)
                print("handler", chain, tag, "J shape", res["shape"], "nnz", res["data"].shape)


if __name__ == "__main__":
    main()
