"""ParamHandler surface of the path — the drop-in boundary.

Mirrors, for the cost/Jacobian path only, pyCamSet's
  ``TemplateBundlePrimitive`` / ``TemplateBundleHandler``   optimisation/template_handler.py:32-240
  ``StandardBundlePrimitive`` / ``SelfBundleHandler``       optimisation/standard_bundle_handler.py:46-260
  ``FreePointPrimitive`` / ``FreePointBundleHandler``       optimisation/free_point_handler.py:47-201
with the same constructor arguments, attribute names, free-vector layout
``x = [9*F_intr | 6*F_extr | 6*F_pose | F_pointscalar]`` and the same closures:

    loss_fn = handler.make_loss_fun(threads)   # x -> (2N,) float64
    jac_fn  = handler.make_loss_jac(threads)   # x -> scipy.sparse.csr_array (2N, len(x))

so ``scipy.optimize.least_squares(loss_fn, x0, jac=jac_fn, x_scale='jac', ...)``
(optimisation_handling.py:88-98) runs unchanged.  ``camset`` only needs ``get_names()`` /
``get_n_cams()`` and ``target`` only needs ``point_data`` — the attributes the reference touches on
this path (th:116-129, th:160-163).  Initial-pose estimation (OpenCV PnP, th:302-346), outlier
prompts and CameraSet reconstruction are outside the path: initial parameters are supplied with
``set_initial_params``.
"""
from __future__ import annotations

from copy import deepcopy
from itertools import combinations

import numpy as np
from scipy.sparse import csr_array

from . import function_blocks as fb
from .detections import TargetDetection

DEFAULT_OPTIONS = {  # th:24-31
    "verbosity": 2,
    "fixed_pose": 0,
    "ref_cam": 0,
    "ref_pose": 0,
    "outliers": "ask",
    "max_nfev": 100,
}


def fill_flat(src: np.ndarray, dst: np.ndarray, dst_unfixed: np.ndarray) -> None:
    """compiled_helpers.py:155-177: scatter the free rows / scalars of ``src`` into ``dst``."""
    dst[np.asarray(dst_unfixed, dtype=bool)] = src


def _list_dict_to_np_array(d):  # utils/general_utils.py:21-30
    if isinstance(d, dict):
        for key, val in d.items():
            if isinstance(val, dict):
                _list_dict_to_np_array(val)
            elif isinstance(val, list):
                d[key] = np.array(val)
    return d


class TemplateBundlePrimitive:  # th:32-78
    def __init__(self, poses, extr, intr, poses_unfixed=None, extr_unfixed=None, intr_unfixed=None):
        self.poses = poses
        self.poses_unfixed = poses_unfixed if poses_unfixed is not None else np.ones(poses.shape[0], dtype=bool)
        self.extr = extr
        self.extr_unfixed = extr_unfixed if extr_unfixed is not None else np.ones(extr.shape[0], dtype=bool)
        self.intr = intr
        self.intr_unfixed = intr_unfixed if intr_unfixed is not None else np.ones(intr.shape[0], dtype=bool)
        self.calc_free_poses()

    def calc_free_poses(self):
        self.free_poses = int(np.sum(self.poses_unfixed))
        self.free_extr = int(np.sum(self.extr_unfixed))
        self.free_intr = int(np.sum(self.intr_unfixed))
        self.intr_end = 9 * self.free_intr
        self.extr_end = 6 * self.free_extr + self.intr_end
        self.pose_end = 6 * self.free_poses + self.extr_end

    def return_bundle_primitives(self, params):  # th:63-78
        intr_data = params[: self.intr_end].reshape((self.free_intr, 9))
        extr_data = params[self.intr_end : self.extr_end].reshape((self.free_extr, 6))
        pose_data = params[self.extr_end : self.pose_end].reshape((self.free_poses, 6))
        fill_flat(pose_data, self.poses, self.poses_unfixed)
        fill_flat(extr_data, self.extr, self.extr_unfixed)
        fill_flat(intr_data, self.intr, self.intr_unfixed)
        return self.intr, self.extr, self.poses


class StandardBundlePrimitive:  # sbh:46-107
    def __init__(self, poses, bundle_points, extr, intr, poses_unfixed=None, bundle_points_unfixed=None,
                 extr_unfixed=None, intr_unfixed=None, always_correct_gauge=False):
        self.extr = extr
        self.extr_unfixed = extr_unfixed if extr_unfixed is not None else np.ones(extr.shape[0], dtype=bool)
        self.intr = intr
        self.intr_unfixed = intr_unfixed if intr_unfixed is not None else np.ones(intr.shape[0], dtype=bool)
        self.bundle_pts = bundle_points
        self.bdpt_unfixed = bundle_points_unfixed if bundle_points_unfixed is not None else np.ones(bundle_points.shape[0], dtype=bool)
        self.correct_gauge = True
        self.poses = poses
        self.poses_unfixed = poses_unfixed if poses_unfixed is not None else np.ones(poses.shape[0], dtype=bool)
        self.calc_type_inds()

    def calc_type_inds(self):
        self.free_extr = int(np.sum(self.extr_unfixed))
        self.free_intr = int(np.sum(self.intr_unfixed))
        self.free_pose = int(np.sum(self.poses_unfixed))
        self.free_bdpt = int(np.sum(self.bdpt_unfixed))
        self.intr_end = 9 * self.free_intr
        self.extr_end = 6 * self.free_extr + self.intr_end
        self.pose_end = 6 * self.free_pose + self.extr_end
        self.bdpt_end = 1 * self.free_bdpt + self.pose_end

    def return_bundle_primitives(self, params):  # sbh:88-107
        intr_data = params[: self.intr_end].reshape((self.free_intr, 9))
        extr_data = params[self.intr_end : self.extr_end].reshape((self.free_extr, 6))
        pose_data = params[self.extr_end : self.pose_end].reshape((self.free_pose, 6))
        bdpt_data = params[self.pose_end : self.bdpt_end]
        fill_flat(pose_data, self.poses, self.poses_unfixed)
        fill_flat(extr_data, self.extr, self.extr_unfixed)
        fill_flat(intr_data, self.intr, self.intr_unfixed)
        fill_flat(bdpt_data, self.bundle_pts, self.bdpt_unfixed)
        return self.intr, self.extr, self.poses, self.bundle_pts.reshape((-1, 3))


class FreePointPrimitive:  # fph:47-100
    def __init__(self, bundle_points, extr, intr, bundle_points_unfixed=None, extr_unfixed=None, intr_unfixed=None):
        self.extr = extr
        self.extr_unfixed = extr_unfixed if extr_unfixed is not None else np.ones(extr.shape[0], dtype=bool)
        self.intr = intr
        self.intr_unfixed = intr_unfixed if intr_unfixed is not None else np.ones(intr.shape[0], dtype=bool)
        self.bundle_pts = bundle_points
        self.bdpt_unfixed = bundle_points_unfixed if bundle_points_unfixed is not None else np.ones(bundle_points.shape[0], dtype=bool)
        self.correct_gauge = True
        self.calc_type_inds()

    def calc_type_inds(self):
        self.free_extr = int(np.sum(self.extr_unfixed))
        self.free_intr = int(np.sum(self.intr_unfixed))
        self.free_bdpt = int(np.sum(self.bdpt_unfixed))
        self.intr_end = 9 * self.free_intr
        self.extr_end = 6 * self.free_extr + self.intr_end
        self.bdpt_end = 1 * self.free_bdpt + self.extr_end

    def return_bundle_primitives(self, params):  # fph:84-100
        intr_data = params[: self.intr_end].reshape((self.free_intr, 9))
        extr_data = params[self.intr_end : self.extr_end].reshape((self.free_extr, 6))
        bdpt_data = params[self.extr_end : self.bdpt_end]
        fill_flat(extr_data, self.extr, self.extr_unfixed)
        fill_flat(intr_data, self.intr, self.intr_unfixed)
        fill_flat(bdpt_data, self.bundle_pts, self.bdpt_unfixed)
        return self.intr, self.extr, self.bundle_pts.reshape((-1, 3))


class TemplateBundleHandler:  # th:80-240
    """Target-pose based bundle adjustment against a constant template (chain T)."""

    def __init__(self, camset, target, detection: TargetDetection, fixed_params: dict | None = None,
                 options: dict | None = None, missing_poses: list | None = None, *, dtype: str = "f64", device: int = 0,
                 pinned_ring: int = 0, counts=None):
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
        self._dtype, self._device, self._pinned_ring, self._counts = dtype, device, pinned_ring, counts

        n_poses = detection.max_ims
        n_cams = camset.get_n_cams()
        intr = np.zeros((n_cams, 9))
        extr = np.zeros((n_cams, 6))
        poses = np.zeros((n_poses, 6))
        extr_unfixed = np.array(["ext" not in self.fixed_params.get(cam_name, {}) for cam_name in self.cam_names])
        intr_unfixed = np.array(["int" not in self.fixed_params.get(cam_name, {}) for cam_name in self.cam_names])
        pose_unfixed = np.ones(n_poses, dtype=bool)
        if "fixed_pose" in self.problem_opts and self.problem_opts["fixed_pose"] is not None:  # th:134-137
            fixed_pose = self.problem_opts["fixed_pose"]
            pose_unfixed[fixed_pose] = False
            poses[fixed_pose, :] = [0, 0, 0, 0, 0, 0]
        self.bundlePrimitive = TemplateBundlePrimitive(
            poses, extr, intr, extr_unfixed=extr_unfixed, intr_unfixed=intr_unfixed, poses_unfixed=pose_unfixed)
        self.populate_self_from_fixed_params()
        self.param_len = None
        self.jac_mask = None
        self.missing_poses = missing_poses
        self.op_fun = fb.optimisation_function(
            [fb.projection(), fb.extrinsic3D(), fb.template_points()], dtype=dtype, device=device,
            pinned_ring=pinned_ring, counts=counts)  # th:152

    # -- the path ------------------------------------------------------------------------------
    def can_make_jac(self):  # th:154-155
        return self.op_fun.can_make_jac()

    def _flat_detections(self) -> np.ndarray:
        target_shape = self.target.point_data.shape
        return self.detection.return_flattened_keys(target_shape[:-1]).get_data()  # th:162-163

    def _jac_mask(self) -> np.ndarray:  # th:177-183
        return np.concatenate((
            np.repeat(self.bundlePrimitive.intr_unfixed, 9),
            np.repeat(self.bundlePrimitive.extr_unfixed, 6),
            np.repeat(self.bundlePrimitive.poses_unfixed, 6),
        ), axis=0)

    def _template_arg(self):
        return self.target.point_data.reshape((-1, 3))  # th:160

    def make_loss_fun(self, threads=None):  # th:157-170
        obj_data = self._template_arg()
        dd = self._flat_detections()
        temp_loss = self.op_fun.make_full_loss_fn(dd, threads)

        def loss_fun(params):
            inps = self.get_bundle_adjustment_inputs(params)
            param_str = self.op_fun.build_param_list(*inps)
            return temp_loss(param_str, obj_data).flatten()

        return loss_fun

    def make_loss_jac(self, threads=None):  # th:172-193
        obj_data = self._template_arg()
        dd = self._flat_detections()
        temp_loss = self.op_fun.make_jacobean(dd, threads, unfixed_params=self._jac_mask())

        def jac_fn(params):
            inps = self.get_bundle_adjustment_inputs(params)
            param_str = self.op_fun.build_param_list(*inps)
            d, c, rp = temp_loss(param_str, obj_data)
            return csr_array((d, c, rp), shape=(2 * dd.shape[0], params.shape[0]))

        return jac_fn

    def populate_self_from_fixed_params(self):  # th:204-213
        for idx, cam_name in enumerate(self.cam_names):
            if "ext" in self.fixed_params.get(cam_name, {}):
                self.bundlePrimitive.extr[idx] = self.fixed_params[cam_name]["ext"]
            if "int" in self.fixed_params.get(cam_name, {}):
                self.bundlePrimitive.intr[idx] = self.fixed_params[cam_name]["int"]

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
        dims = self.target_point_shape[:-1]
        detection = self.detection
        if self.missing_poses is not None and np.any(self.missing_poses):
            detection = self.detection.delete_row(im_num=np.where(self.missing_poses)[0])
        if flatten:
            return detection.return_flattened_keys(dims).get_data()
        return detection.get_data()

    def gauge_fixes(self):  # th:417-423
        return None


def find_not_colinear_pts(points):  # sbh:30-44
    ind0 = 0
    for ind1, ind2 in combinations(np.arange(1, points.shape[0]), 2):
        AB = points[ind0] - points[ind1]
        AC = points[ind0] - points[ind2]
        if np.linalg.norm(np.cross(AB, AC)) > 1e-8:
            return ind0, ind1, ind2
    raise ValueError("No set of values that were not colinear were found in the provided data.")


class SelfBundleHandler(TemplateBundleHandler):  # sbh:109-260
    """Self-calibration: the 3-D target points are free too (chain S), 7-DoF gauge fixed."""

    def __init__(self, camset, target, detection, fixed_params=None, options=None, missing_poses=None,
                 *, dtype: str = "f64", device: int = 0, pinned_ring: int = 0, counts=None, visible_feature_mask=None):
        """``visible_feature_mask`` (keyword-only extension): which features are seen by ANY rank.  The
        reference derives it from the handler's own detections (sbh:160-169); a rank that holds only a
        shard must be given the global mask, or the ranks would fix different features."""
        super().__init__(camset, target, detection, fixed_params, options, missing_poses, dtype=dtype, device=device,
                         pinned_ring=pinned_ring, counts=counts)
        self.flat_point_data = np.copy(self.point_data.reshape((-1)))
        self.fixed_inds = find_not_colinear_pts(self.flat_point_data.reshape((-1, 3)))  # sbh:153-158
        i0, i1, i2 = self.fixed_inds
        self.feat_unfixed = np.ones(self.flat_point_data.shape[0], dtype=bool)
        self.feat_unfixed[3 * i0 : 3 * i0 + 3] = False
        self.feat_unfixed[3 * i1 : 3 * i1 + 3] = False
        self.feat_unfixed[3 * i2] = False
        n_points = int(np.prod(self.point_data.shape[:2]))  # sbh:161
        dd = self._flat_detections()[:, 2]
        if visible_feature_mask is not None:
            self.visible_feature_mask = np.asarray(visible_feature_mask, dtype=bool)
            if self.visible_feature_mask.shape[0] != n_points:
                raise ValueError("visible_feature_mask must have one entry per target point")
        else:
            self.visible_feature_mask = np.isin(np.arange(n_points), dd)  # sbh:166
        for idf, vf in enumerate(self.visible_feature_mask):  # sbh:167-169
            if not vf:
                self.feat_unfixed[3 * idf : 3 * idf + 3] = False
        sup = self.bundlePrimitive
        self.bundlePrimitive = StandardBundlePrimitive(
            sup.poses, self.flat_point_data, sup.extr, sup.intr, extr_unfixed=sup.extr_unfixed,
            intr_unfixed=sup.intr_unfixed, poses_unfixed=sup.poses_unfixed, bundle_points_unfixed=self.feat_unfixed)
        self.op_fun = fb.optimisation_function(
            [fb.projection(), fb.extrinsic3D(), fb.rigidTform3d(), fb.free_point()], dtype=dtype, device=device,
            pinned_ring=pinned_ring, counts=counts)  # sbh:182

    def _jac_mask(self):  # sbh:211-218
        return np.concatenate((
            np.repeat(self.bundlePrimitive.intr_unfixed, 9),
            np.repeat(self.bundlePrimitive.extr_unfixed, 6),
            np.repeat(self.bundlePrimitive.poses_unfixed, 6),
            np.repeat(self.bundlePrimitive.bdpt_unfixed, 1),
        ), axis=0)

    def _template_arg(self):
        return None  # sbh:198, sbh:224: the generated functions are called without a template


class FreePointBundleHandler(TemplateBundleHandler):  # fph:102-201
    """Classic bundle adjustment of world points without a target pose (chain F)."""

    def __init__(self, camset, target, detection, fixed_params=None, options=None, missing_poses=None,
                 *, dtype: str = "f64", device: int = 0, pinned_ring: int = 0, counts=None):
        super().__init__(camset, target, detection, fixed_params, options, missing_poses, dtype=dtype, device=device,
                         pinned_ring=pinned_ring, counts=counts)
        self.flat_point_data = np.copy(self.point_data.reshape((-1)))
        self.feat_unfixed = np.ones(self.flat_point_data.shape[0], dtype=bool)
        self.super_primitive = self.bundlePrimitive
        self.bundlePrimitive = FreePointPrimitive(
            self.flat_point_data, self.super_primitive.extr, self.super_primitive.intr,
            extr_unfixed=self.super_primitive.extr_unfixed, intr_unfixed=self.super_primitive.intr_unfixed,
            bundle_points_unfixed=self.feat_unfixed)
        self.op_fun = fb.optimisation_function(
            [fb.projection(), fb.extrinsic3D(), fb.free_point()], dtype=dtype, device=device,
            pinned_ring=pinned_ring, counts=counts)  # fph:143

    def _jac_mask(self):  # fph:172-178
        return np.concatenate((
            np.repeat(self.bundlePrimitive.intr_unfixed, 9),
            np.repeat(self.bundlePrimitive.extr_unfixed, 6),
            np.repeat(self.bundlePrimitive.bdpt_unfixed, 1),
        ), axis=0)

    def _template_arg(self):
        return None
