"""ParamHandler surface of the path — the drop-in boundary.

For the cost/Jacobian path only, this provides what pyCamSet's
  ``TemplateBundleHandler``   optimisation/template_handler.py:80-240          (chain T)
  ``SelfBundleHandler``       optimisation/standard_bundle_handler.py:109-260  (chain S)
  ``FreePointBundleHandler``  optimisation/free_point_handler.py:102-201       (chain F)
and their ``*BundlePrimitive`` helpers (th:32-78, sbh:46-107, fph:47-100) provide: the same constructor
arguments and attribute names, the same free-vector layout
``x = [9 * F_intr | 6 * F_extr | 6 * F_pose | F_pointscalar]`` and the same closures

    loss_fn = handler.make_loss_fun(threads)   # x -> (2N,) float64
    jac_fn  = handler.make_loss_jac(threads)   # x -> scipy.sparse.csr_array (2N, len(x))

so ``scipy.optimize.least_squares(loss_fn, x0, jac=jac_fn, x_scale='jac', ...)``
(optimisation_handling.py:88-98) runs unchanged.  The three reference handlers differ only in which
parameter groups exist, so here ONE table (``_CHAIN_GROUPS``) drives one primitive and one handler core.

``camset`` only needs ``get_names()`` / ``get_n_cams()`` and ``target`` only needs ``point_data`` — the
attributes the reference touches on this path (th:116-129, th:160-163).  Initial-pose estimation (OpenCV
PnP, th:302-346), outlier prompts and CameraSet reconstruction are outside the path: initial parameters
are supplied with ``set_initial_params``.

Differences from the reference, on purpose:
  * The engine lays the parameter string out from the SLAB sizes (n_cams, max_ims, number of target
    points), not from ``max(index) + 1`` of the detections (afb:793-795).  When the last camera / image /
    key has no detection the two rules differ; trailing unused entries (template chain: last images; self
    chain: last keys) give identical results, and where the reference's rule mis-offsets a later group
    (SURVEY 8a quirk ii: last image unobserved in the self chain, last camera unobserved in any chain) this
    package evaluates the slabs the handler actually built.  tests/test_host_logic.py pins both facts against
    fixtures made by the reference.
  * ``options={'fixed_pose': None}``: NumPy reads a ``None`` index as ``newaxis``, so the reference fixes and
    zeroes EVERY pose (th:134-137).  Reproduced as is (same expression).
"""
from __future__ import annotations

from copy import deepcopy

import numpy as np
from scipy.sparse import csr_array

from . import function_blocks as fb
from .detections import TargetDetection  # noqa: F401  (re-exported: the handlers' input type)

DEFAULT_OPTIONS = {  # th:24-31
    "verbosity": 2,
    "fixed_pose": 0,
    "ref_cam": 0,
    "ref_pose": 0,
    "outliers": "ask",
    "max_nfev": 100,
}

# parameter groups in free-vector / parameter-string order: (slab attribute, mask attribute, scalars per free unit)
_GROUP = {
    "intr": ("intr", "intr_unfixed", 9),
    "extr": ("extr", "extr_unfixed", 6),
    "pose": ("poses", "poses_unfixed", 6),
    "bdpt": ("bundle_pts", "bdpt_unfixed", 1),   # points are fixed per scalar (sbh:83-86)
}
_CHAIN_GROUPS = {"template": ("intr", "extr", "pose"), "self": ("intr", "extr", "pose", "bdpt"), "free": ("intr", "extr", "bdpt")}


def fill_flat(src: np.ndarray, dst: np.ndarray, dst_unfixed: np.ndarray) -> None:
    """compiled_helpers.py:155-177: scatter the free rows / scalars of ``src`` into ``dst``."""
    dst[np.asarray(dst_unfixed, dtype=bool)] = src


def _list_dict_to_np_array(d):
    """Lists inside a (nested) ``fixed_params`` dict become arrays, in place — the normalisation the reference applies to
    the dict before reading 'int' / 'ext' entries out of it (utils/general_utils.py:21-30)."""
    pending = [d] if isinstance(d, dict) else []
    while pending:
        level = pending.pop()
        for name in list(level):
            entry = level[name]
            if isinstance(entry, list):
                level[name] = np.asarray(entry)
            elif isinstance(entry, dict):
                pending.append(entry)
    return d


class BundlePrimitive:
    """Free vector ``x`` -> full parameter slabs for one chain (th:32-78, sbh:46-107, fph:47-100).

    Holds persistent full slabs (``intr`` (C,9), ``extr`` (C,6), ``poses`` (I,6), ``bundle_pts`` (3K,)) and one
    boolean "unfixed" mask per slab; ``return_bundle_primitives(x)`` scatters the free entries into them and
    returns the slabs in block order.  ``intr_end`` / ``extr_end`` / ``pose_end`` / ``bdpt_end`` are the
    cumulative ends of the groups inside ``x``, ``free_intr`` … the free unit counts, as in the reference."""

    def __init__(self, chain: str, slabs: dict, unfixed: dict):
        self.chain = chain
        self.groups = _CHAIN_GROUPS[chain]
        for g in self.groups:
            slab_name, mask_name, _ = _GROUP[g]
            slab = slabs[g]
            mask = unfixed.get(g)
            setattr(self, slab_name, slab)
            setattr(self, mask_name, np.ones(slab.shape[0], dtype=bool) if mask is None else mask)
        self.correct_gauge = True
        self.calc_type_inds()

    def calc_type_inds(self):
        end = 0
        for g in self.groups:
            _, mask_name, width = _GROUP[g]
            n_free = int(np.sum(getattr(self, mask_name)))
            end += width * n_free
            setattr(self, f"free_{g}", n_free)
            setattr(self, f"{g}_end", end)
        if "pose" in self.groups:
            self.free_poses = self.free_pose   # both spellings exist in the reference (th:52, sbh:77)

    calc_free_poses = calc_type_inds  # th:50

    def return_bundle_primitives(self, params):
        start, out = 0, []
        for g in self.groups:
            slab_name, mask_name, width = _GROUP[g]
            slab, mask, end = getattr(self, slab_name), getattr(self, mask_name), getattr(self, f"{g}_end")
            part = params[start:end]
            fill_flat(part.reshape((-1, width)) if width > 1 else part, slab, mask)
            out.append(slab.reshape((-1, 3)) if g == "bdpt" else slab)
            start = end
        return tuple(out)


def TemplateBundlePrimitive(poses, extr, intr, poses_unfixed=None, extr_unfixed=None, intr_unfixed=None):  # th:32-47
    return BundlePrimitive("template", {"intr": intr, "extr": extr, "pose": poses},
                           {"intr": intr_unfixed, "extr": extr_unfixed, "pose": poses_unfixed})


def StandardBundlePrimitive(poses, bundle_points, extr, intr, poses_unfixed=None, bundle_points_unfixed=None,
                            extr_unfixed=None, intr_unfixed=None, always_correct_gauge=False):  # sbh:46-75
    return BundlePrimitive("self", {"intr": intr, "extr": extr, "pose": poses, "bdpt": bundle_points},
                           {"intr": intr_unfixed, "extr": extr_unfixed, "pose": poses_unfixed, "bdpt": bundle_points_unfixed})


def FreePointPrimitive(bundle_points, extr, intr, bundle_points_unfixed=None, extr_unfixed=None, intr_unfixed=None):  # fph:47-69
    return BundlePrimitive("free", {"intr": intr, "extr": extr, "bdpt": bundle_points},
                           {"intr": intr_unfixed, "extr": extr_unfixed, "bdpt": bundle_points_unfixed})


def find_not_colinear_pts(points):
    """Indices of three points that span a plane: point 0 and the first pair (i, j), 1 <= i < j in lexicographic order,
    whose triangle with it has a cross product longer than 1e-8 (the gauge points of sbh:30-44, same search order)."""
    pts = np.asarray(points, dtype=np.float64)
    legs = pts[1:] - pts[0]
    for i in range(legs.shape[0]):
        areas = np.linalg.norm(np.cross(legs[i], legs[i + 1:]), axis=-1)
        hit = np.nonzero(areas > 1e-8)[0]
        if hit.size:
            return 0, i + 1, i + 2 + int(hit[0])
    raise ValueError("No set of values that were not colinear were found in the provided data.")


class TemplateBundleHandler:  # th:80-240
    """Target-pose based bundle adjustment against a constant template (chain T).  The subclasses only
    change ``chain`` and the point masks; everything on the path lives here."""

    chain = "template"
    _BLOCKS = {"template": (fb.projection, fb.extrinsic3D, fb.template_points),                  # th:152
               "self": (fb.projection, fb.extrinsic3D, fb.rigidTform3d, fb.free_point),          # sbh:182
               "free": (fb.projection, fb.extrinsic3D, fb.free_point)}                           # fph:143

    def __init__(self, camset, target, detection, fixed_params: dict | None = None, options: dict | None = None,
                 missing_poses: list | None = None, *, dtype: str = "f64", device: int = 0, pinned_ring: int | None = None,
                 counts=None, visible_feature_mask=None):
        """Keyword-only extensions: ``dtype`` / ``device`` / ``pinned_ring`` go to the engine
        (function_blocks.optimisation_function); ``counts`` = (n_cams, n_imgs, n_keys) overrides the slab sizes of
        the parameter-string layout (a rank that holds a shard of the detections must lay out the GLOBAL string);
        ``counts="reference"`` selects the reference's own rule, max index + 1 of the detections (afb:793-795) —
        what a reference handler patched with ``pycamset_amd.function_blocks`` gets, quirk ii included;
        ``visible_feature_mask`` (self chain): which features are seen by ANY rank — the reference derives it from
        the handler's own detections (sbh:160-169), and ranks holding shards must fix the same features."""
        self.problem_opts = dict(DEFAULT_OPTIONS)  # the reference aliases and mutates the module global (th:108-110)
        if options is not None:
            self.problem_opts.update(options)
        self.fixed_params = _list_dict_to_np_array(fixed_params)
        if fixed_params is None:
            self.fixed_params = {}
        self.camset = camset
        self.cam_names = camset.get_names()
        self.detection = deepcopy(detection)
        self.target = target
        self.point_data = deepcopy(target.point_data)
        self.target_point_shape = np.array(target.point_data.shape)
        self.initial_params = None
        self.param_len = None
        self.jac_mask = None
        self.missing_poses = missing_poses

        n_poses = detection.max_ims
        n_cams = camset.get_n_cams()
        slabs = {"intr": np.zeros((n_cams, 9)), "extr": np.zeros((n_cams, 6)), "pose": np.zeros((n_poses, 6))}
        unfixed = {"extr": np.array(["ext" not in self.fixed_params.get(name, {}) for name in self.cam_names]),
                   "intr": np.array(["int" not in self.fixed_params.get(name, {}) for name in self.cam_names]),
                   "pose": np.ones(n_poses, dtype=bool)}
        if "fixed_pose" in self.problem_opts:  # th:134-137 (a None index selects every pose, see module docstring)
            fixed_pose = self.problem_opts["fixed_pose"]
            unfixed["pose"][fixed_pose] = False
            slabs["pose"][fixed_pose, :] = [0, 0, 0, 0, 0, 0]
        if self.chain != "template":
            self.flat_point_data = np.copy(self.point_data.reshape((-1)))
            slabs["bdpt"] = self.flat_point_data
            unfixed["bdpt"] = self.feat_unfixed = self._point_mask(visible_feature_mask)
        if self.chain == "free":
            self.super_primitive = TemplateBundlePrimitive(slabs["pose"], slabs["extr"], slabs["intr"], unfixed["pose"],
                                                           unfixed["extr"], unfixed["intr"])  # fph:131
        self.bundlePrimitive = BundlePrimitive(self.chain, slabs, unfixed)
        self.populate_self_from_fixed_params()
        n_keys = int(np.prod(self.point_data.shape[:-1]))
        self.op_fun = fb.optimisation_function(
            [b() for b in self._BLOCKS[self.chain]], dtype=dtype, device=device, pinned_ring=pinned_ring,
            counts=None if counts == "reference" else counts if counts is not None else (n_cams, n_poses, n_keys))

    def _point_mask(self, visible_feature_mask):
        return None

    # -- the path ------------------------------------------------------------------------------
    def can_make_jac(self):  # th:154-155
        return self.op_fun.can_make_jac()

    def _flat_detections(self) -> np.ndarray:
        return self.detection.return_flattened_keys(self.target.point_data.shape[:-1]).get_data()  # th:162-163

    def _jac_mask(self) -> np.ndarray:  # th:177-183, sbh:211-218, fph:172-178
        bp = self.bundlePrimitive
        return np.concatenate([np.repeat(getattr(bp, _GROUP[g][1]), _GROUP[g][2]) for g in bp.groups], axis=0)

    def _template_arg(self):
        # th:160; the self / free closures call the generated functions without a template (sbh:198, fph:159)
        return self.target.point_data.reshape((-1, 3)) if self.chain == "template" else None

    def make_loss_fun(self, threads=None):  # th:157-170, sbh:184-198, fph:145-159
        obj_data = self._template_arg()
        temp_loss = self.op_fun.make_full_loss_fn(self._flat_detections(), threads)

        def loss_fun(params):
            param_str = self.op_fun.build_param_list(*self.get_bundle_adjustment_inputs(params))
            return temp_loss(param_str, obj_data).flatten()

        return loss_fun

    def make_loss_jac(self, threads=None):  # th:172-193, sbh:200-226, fph:161-186
        obj_data = self._template_arg()
        dd = self._flat_detections()
        temp_loss = self.op_fun.make_jacobean(dd, threads, unfixed_params=self._jac_mask())

        def jac_fn(params):
            param_str = self.op_fun.build_param_list(*self.get_bundle_adjustment_inputs(params))
            d, c, rp = temp_loss(param_str, obj_data)
            return csr_array((d, c, rp), shape=(2 * dd.shape[0], params.shape[0]))

        return jac_fn

    def populate_self_from_fixed_params(self):  # th:204-213
        for idx, cam_name in enumerate(self.cam_names):
            fixed = self.fixed_params.get(cam_name, {})
            if "ext" in fixed:
                self.bundlePrimitive.extr[idx] = fixed["ext"]
            if "int" in fixed:
                self.bundlePrimitive.intr[idx] = fixed["int"]

    def get_bundle_adjustment_inputs(self, x, make_points=False):  # th:215-240
        if make_points:
            raise NotImplementedError("make_points is a visualisation helper outside the cost/Jacobian path")
        return self.bundlePrimitive.return_bundle_primitives(x)

    # -- parameters ----------------------------------------------------------------------------
    def set_initial_params(self, x: np.ndarray):  # th:281-288
        self.initial_params = x

    def get_initial_params(self) -> np.ndarray:  # th:290-300
        if self.initial_params is not None:
            return self.initial_params
        raise NotImplementedError(
            "calc_initial_params needs OpenCV PnP (template_handler.py:302-346), which is outside the "
            "accelerated path: supply a start vector with set_initial_params()")

    def get_detection_data(self, flatten=False) -> np.ndarray:  # th:387-406
        detection = self.detection
        if self.missing_poses is not None and np.any(self.missing_poses):
            detection = self.detection.delete_row(im_num=np.where(self.missing_poses)[0])
        if flatten:
            return detection.return_flattened_keys(self.target_point_shape[:-1]).get_data()
        return detection.get_data()

    def gauge_fixes(self):  # th:417-423
        return None


class SelfBundleHandler(TemplateBundleHandler):  # sbh:109-260
    """Self-calibration: the 3-D target points are free too (chain S), 7-DoF gauge fixed."""

    chain = "self"

    def _point_mask(self, visible_feature_mask):
        pts = self.flat_point_data.reshape((-1, 3))
        self.fixed_inds = find_not_colinear_pts(pts)  # sbh:153-158: 3 + 3 + 1 coordinates fix the 7-DoF gauge
        i0, i1, i2 = self.fixed_inds
        feat_unfixed = np.ones(self.flat_point_data.shape[0], dtype=bool)
        feat_unfixed[3 * i0 : 3 * i0 + 3] = False
        feat_unfixed[3 * i1 : 3 * i1 + 3] = False
        feat_unfixed[3 * i2] = False
        n_points = int(np.prod(self.point_data.shape[:2]))  # sbh:161
        if visible_feature_mask is not None:
            self.visible_feature_mask = np.asarray(visible_feature_mask, dtype=bool)
            if self.visible_feature_mask.shape[0] != n_points:
                raise ValueError("visible_feature_mask must have one entry per target point")
        else:
            self.visible_feature_mask = np.isin(np.arange(n_points), self._flat_detections()[:, 2])  # sbh:166
        feat_unfixed[: 3 * n_points][np.repeat(~self.visible_feature_mask, 3)] = False  # sbh:167-169: unseen features are fixed
        return feat_unfixed


class FreePointBundleHandler(TemplateBundleHandler):  # fph:102-201
    """Classic bundle adjustment of world points without a target pose (chain F)."""

    chain = "free"

    def _point_mask(self, visible_feature_mask):
        return np.ones(self.flat_point_data.shape[0], dtype=bool)  # fph:130



class ChainProblem:
    """The handler protocol for ANY chain of function blocks — what ``device_solver.lm_solve`` and a scipy caller need from a
    handler (template_handler.py:154-240: ``op_fun``, ``make_loss_fun``, ``make_loss_jac``, ``get_bundle_adjustment_inputs``) without
    the target / camera-set model around it.  The three bundle handlers above are tied to the reference's three chains; a chain
    the user composes (generated kernels, user blocks) brings its own parameter groups, so this class takes them as they are:

        op   = projection() + extrinsic3D() + rigidTform3d() + template_points()
        prob = ChainProblem(op, detections, [intr, extr, poses_a, poses_b], template=points,
                            unfixed=[None, None, None, mask_b])            # one array per block, like build_param_list
        res  = lm_solve(prob, prob.x0)                                     # J stays on the device
        res, slabs = run_bundle_adjustment(prob, solver="device")          # the reference's caller (optimisation_handling.py:52-117)
        scipy.optimize.least_squares(prob.make_loss_fun(), prob.x0, jac=prob.make_loss_jac(), x_scale='jac')

    ``slabs``: one array per argument of ``build_param_list`` (afb:669-681), i.e. per parameter GROUP in string order (blocks that
    share a parameter object share one array); ``unfixed``: a boolean array per slab (None = all free).  The free vector ``x`` is the
    concatenation of the slabs' free entries in string order."""

    def __init__(self, op_fun, detections, slabs, *, template=None, unfixed=None, options=None):
        # what optimisation_handling.run_bundle_adjustment reads from a handler (oh:52-117): the reference's option names and defaults
        # (template_handler.py:110-118)
        self.problem_opts = {"verbosity": 0, "max_nfev": 100}
        self.problem_opts.update(options or {})
        self.op_fun = op_fun
        self.det = np.ascontiguousarray(detections, dtype=np.float64)
        self.slabs = [np.array(s, dtype=np.float64) for s in slabs]
        masks = unfixed if unfixed is not None else [None] * len(self.slabs)
        if len(masks) != len(self.slabs):
            raise ValueError("one unfixed mask (or None) per slab")
        self.unfixed = [np.ones(s.shape, dtype=bool) if m is None else np.broadcast_to(np.asarray(m, dtype=bool), s.shape).copy() for s, m in zip(self.slabs, masks)]
        self.template = None if template is None else np.ascontiguousarray(template, dtype=np.float64).reshape(-1, 3)
        self.x0 = np.concatenate([s[m] for s, m in zip(self.slabs, self.unfixed)])

    # -- what run_bundle_adjustment asks of a handler (oh:24-49) ------------------------------------------------------------------------
    def get_initial_params(self):
        return self.x0.copy()

    def can_make_jac(self):
        return True

    # -- what lm_solve asks of a handler ---------------------------------------------------------------------------------------------
    def _flat_detections(self):
        return self.det

    def _template_arg(self):
        return self.template

    def _jac_mask(self):
        return np.concatenate([m.ravel() for m in self.unfixed])

    def get_bundle_adjustment_inputs(self, x):
        x = np.asarray(x, dtype=np.float64)
        out, at = [], 0
        for s, m in zip(self.slabs, self.unfixed):
            k = int(m.sum())
            t = s.copy()
            t[m] = x[at: at + k]
            at += k
            out.append(t)
        if at != x.shape[0]:
            raise ValueError(f"x has {x.shape[0]} entries, the problem has {at} free parameters")
        return out

    # -- the closures scipy takes (th:157-193) -------------------------------------------------------------------------------------
    def _param_str(self, x):
        return self.op_fun.build_param_list(*self.get_bundle_adjustment_inputs(x))

    def make_loss_fun(self, threads=1):
        loss = self.op_fun.make_full_loss_fn(self.det, threads)
        tm = () if self.template is None else (self.template,)
        return lambda x: np.asarray(loss(self._param_str(x), *tm)).reshape(-1)

    def make_loss_jac(self, threads=1):
        from scipy.sparse import csr_array

        jac = self.op_fun.make_jacobean(self.det, threads, unfixed_params=self._jac_mask())
        tm = () if self.template is None else (self.template,)
        shape = (2 * self.det.shape[0], int(self._jac_mask().sum()))

        def jac_fn(x):
            data, indices, indptr = jac(self._param_str(x), *tm)
            return csr_array((data, indices, indptr), shape=shape)

        return jac_fn
