"""Mirror of the legacy cost entry points and of the batched triangulation
(row f4: ``nb_triangulate_full``, ch:609-643, with the grouping of
``CameraSet.multi_cam_triangulate``, cameras/camera_set.py:371-378) of
pyCamSet/optimisation/compiled_helpers.py, evaluated by the HIP engine:

    bundle_adjustment_costfn(dct, im_points, projection_matrixes, intrinsics, dists) -> (2N,)   ch:518-549
    bundle_adj_parrallel_solver(dct (T, L, 5), ...) -> (T, 2L)                                  ch:493-516

The reference calls them with the detection table on every call (template_handler.py:537-543,
:586-592); the table is uploaded once and cached by content.
"""
from __future__ import annotations

import numpy as np

from .engine import Engine

_cache: dict = {}


def _engine_for(dct: np.ndarray, n_imgs: int, n_keys: int, n_cams: int, device: int) -> Engine:
    dct = np.ascontiguousarray(dct, dtype=np.float64)
    key = (dct.shape, hash(dct.tobytes()), n_imgs, n_keys, n_cams, device)
    if key not in _cache:
        _cache.clear()  # one table at a time, like the reference's single closure
        eng = Engine("template", n_cams, n_imgs, n_keys, device=device)
        eng.set_detections_table(dct)
        _cache[key] = eng
    return _cache[key]


def bundle_adjustment_costfn(dct, im_points, projection_matrixes, intrinsics, dists, device: int = 0) -> np.ndarray:
    im = np.ascontiguousarray(im_points, dtype=np.float64)
    n_imgs = im.shape[0]
    n_keys = int(np.prod(im.shape[1:-1]))
    P = np.ascontiguousarray(projection_matrixes, dtype=np.float64)
    eng = _engine_for(dct, n_imgs, n_keys, P.shape[0], device)
    return eng.legacy_cost(im.reshape(n_imgs, n_keys, 3), P, intrinsics, np.asarray(dists, dtype=np.float64).reshape(P.shape[0], 5))


numpy_bundle_adjustment_costfn = bundle_adjustment_costfn


def bundle_adj_parrallel_solver(dct, im_points, projection_matrixes, intrinsics, dists, device: int = 0) -> np.ndarray:
    dct = np.asarray(dct, dtype=np.float64)
    t, l = dct.shape[0], dct.shape[1]
    flat = bundle_adjustment_costfn(dct.reshape(t * l, dct.shape[2]), im_points, projection_matrixes, intrinsics, dists, device)
    return flat.reshape(t, 2 * l)


last_triangulate_kernel_ms = None


class Triangulator:
    """Owner of one ``pcs_triangulator`` handle (include/pcs_hip.h): the camera table, device copies of the
    observations, the kernel's scratch and the output live on the device across calls, so repeated
    triangulations with the same cameras (``CameraSet.multi_cam_triangulate`` per set of frames,
    cameras/camera_set.py:343-402) allocate nothing and — with device-resident inputs — copy nothing."""

    def __init__(self, n_cams: int, device: int = 0):
        import ctypes

        from . import _capi

        self._capi, self._ct = _capi, ctypes
        self._h = ctypes.c_void_p()
        _capi.check(_capi.lib().pcs_tri_create(ctypes.byref(self._h), int(device), int(n_cams)))
        self.n_cams, self.device, self.n_pts = int(n_cams), int(device), 0
        self._cam_key = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._capi.lib().pcs_tri_destroy(self._h)
            self._h = self._ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_cameras(self, proj, intr, dist):
        P = np.ascontiguousarray(proj, dtype=np.float64)
        K = np.ascontiguousarray(intr, dtype=np.float64)
        D = np.ascontiguousarray(np.asarray(dist, dtype=np.float64).reshape(P.shape[0], -1))
        if P.shape != (self.n_cams, 3, 4) or K.shape != (self.n_cams, 3, 3) or D.shape != (self.n_cams, 5):
            raise ValueError("expected proj (C,3,4), intr (C,3,3), dist (C,5)")
        key = hash(P.tobytes() + K.tobytes() + D.tobytes())
        if key != self._cam_key:   # the same cameras across calls: nothing to upload
            dp = self._ct.POINTER(self._ct.c_double)
            self._capi.check(self._capi.lib().pcs_tri_set_cameras(self._h, P.ctypes.data_as(dp), K.ctypes.data_as(dp), D.ctypes.data_as(dp)))
            self._cam_key = key

    def set_observations(self, cam, uv, start_inds):
        """Host arrays (copied to handle-owned device buffers): cam (n_obs,) int, uv (n_obs, 2), start_inds (n_pts + 1,)."""
        cam = np.ascontiguousarray(cam, dtype=np.int32)
        uv = np.ascontiguousarray(uv, dtype=np.float64)
        start = np.ascontiguousarray(start_inds, dtype=np.int64)
        ct = self._ct
        self._capi.check(self._capi.lib().pcs_tri_set_observations(
            self._h, cam.shape[0], cam.ctypes.data_as(ct.POINTER(ct.c_int32)), uv.ctypes.data_as(ct.POINTER(ct.c_double)),
            start.shape[0] - 1, start.ctypes.data_as(ct.POINTER(ct.c_int64))))
        self.n_pts = start.shape[0] - 1

    def set_observations_device(self, n_obs: int, d_cam: int, d_uv: int, n_pts: int, d_start: int):
        """Raw device addresses (e.g. ``tensor.data_ptr()``) of int32 cam, float64 uv, int64 start_inds: used in place."""
        ct = self._ct
        self._capi.check(self._capi.lib().pcs_tri_set_observations_device(self._h, int(n_obs), ct.c_void_p(d_cam), ct.c_void_p(d_uv),
                                                                          int(n_pts), ct.c_void_p(d_start)))
        self.n_pts = int(n_pts)

    def group_table_device(self, n: int, d_cam: int, d_feat: int, d_uv: int, n_features: int, stream: int | None = None):
        """The grouping of ``multi_cam_triangulate`` (cameras/camera_set.py:371-378) on the device: raw device addresses of int32
        camera indices, int32 dense feature ids (< ``n_features``) and float64 measurements of ``n`` table rows grouped by feature.
        -> (n_pts, n_kept, grouped); with ``grouped`` the handle's current observations are the rows of the features seen by at least
        two cameras (``run`` next); not grouped: nothing was set (group on the host: ``group_reconstructable``)."""
        from .engine import _stream_arg
        ct = self._ct
        n_pts, n_kept, grouped = ct.c_int64(), ct.c_int64(), ct.c_int32()
        self._capi.check(self._capi.lib().pcs_tri_group_device(self._h, int(n), ct.c_void_p(d_cam), ct.c_void_p(d_feat), ct.c_void_p(d_uv), int(n_features),
                                                               ct.byref(n_pts), ct.byref(n_kept), ct.byref(grouped), _stream_arg(stream)))
        if grouped.value:
            self.n_pts = int(n_pts.value)
        return int(n_pts.value), int(n_kept.value), bool(grouped.value)

    def run(self, d_pts: int | None = None, stream: int | None = None):
        """Queue the kernel (asynchronous).  ``d_pts`` = device address of an (n_pts, 3) float64 buffer, or None for the
        handle-owned output (fetch it with ``points()``)."""
        from .engine import _stream_arg
        self._capi.check(self._capi.lib().pcs_tri_run(self._h, self._ct.c_void_p(d_pts or 0), _stream_arg(stream)))

    def synchronize(self, stream: int | None = None):
        from .engine import _stream_arg
        self._capi.check(self._capi.lib().pcs_tri_synchronize(self._h, _stream_arg(stream)))

    def points(self) -> np.ndarray:
        pts = np.empty((max(self.n_pts, 0), 3))
        if self.n_pts > 0:
            self._capi.check(self._capi.lib().pcs_tri_points(self._h, pts.ctypes.data_as(self._ct.POINTER(self._ct.c_double))))
        return pts

    def last_kernel_ms(self) -> float:
        ms = self._ct.c_float(0.0)
        self._capi.check(self._capi.lib().pcs_tri_last_kernel_ms(self._h, self._ct.byref(ms)))
        return float(ms.value)


_tri_cache: dict = {}


def nb_triangulate_full(data, proj, start_inds, intr, dist, device: int = 0) -> np.ndarray:
    """``data`` rows = [cam, ..., u, v] sorted by point; ``start_inds`` (n_pts + 1) like the reference
    (compiled_helpers.py:609-643).  Returns (n_pts, 3).  A ``Triangulator`` per (device, camera count) is kept
    across calls: camera table and buffers are reused."""
    global last_triangulate_kernel_ms
    data = np.asarray(data, dtype=np.float64)
    P = np.asarray(proj, dtype=np.float64)
    start = np.asarray(start_inds, dtype=np.int64)
    if start.shape[0] <= 1:
        return np.empty((0, 3))
    key = (int(device), int(P.shape[0]))
    tri = _tri_cache.get(key)
    if tri is None:
        _tri_cache.clear()
        tri = _tri_cache[key] = Triangulator(P.shape[0], device)
    tri.set_cameras(P, intr, dist)
    tri.set_observations(data[:, 0].astype(np.int32), data[:, -2:], start)
    tri.run()
    pts = tri.points()
    last_triangulate_kernel_ms = tri.last_kernel_ms()
    return pts


def multi_cam_triangulate(data, proj, intr, dist, distort: bool = True, device: int = 0) -> np.ndarray:
    """``CameraSet.multi_cam_triangulate`` for a detection table (cameras/camera_set.py:343-402, the array branch): ``data`` rows =
    [cam, image, key..., u, v] as ``TargetDetection.get_data`` returns them; the features seen by more than one camera are
    triangulated, in order of first appearance.  Grouping (``np.unique`` twice in the reference: 0.37 s for 1e6 rows) AND
    triangulation run on the device; a table whose features are not stored consecutively takes the host grouping
    (``group_reconstructable``: the reference's semantics for any table).  ``distort=False`` zeroes the distortion (camera_set.py:384-385)."""
    global last_triangulate_kernel_ms
    import torch

    table = np.asarray(data, dtype=np.float64)
    P = np.asarray(proj, dtype=np.float64)
    D = np.zeros_like(np.asarray(dist, dtype=np.float64)) if not distort else np.asarray(dist, dtype=np.float64)
    if table.shape[0] == 0:
        return np.empty((0, 3))
    ids = table[:, 1:-2].astype(np.int64)                                   # (image, key...) columns
    dims = ids.max(axis=0) + 1
    n_features = int(np.prod(dims))
    if n_features >= 2 ** 31 or ids.min() < 0:
        rec, start = group_reconstructable(table)
        return nb_triangulate_full(rec, P, start, intr, D, device=device)
    feat = np.ravel_multi_index(ids.T, dims).astype(np.int32)               # a dense id per feature: no sort needed
    key = (int(device), int(P.shape[0]))
    tri = _tri_cache.get(key)
    if tri is None:
        _tri_cache.clear()
        tri = _tri_cache[key] = Triangulator(P.shape[0], device)
    tri.set_cameras(P, intr, D)
    dev = torch.device("cuda", device)
    d_cam = torch.from_numpy(table[:, 0].astype(np.int32)).to(dev)
    d_feat = torch.from_numpy(feat).to(dev)
    d_uv = torch.from_numpy(np.ascontiguousarray(table[:, -2:])).to(dev)
    torch.cuda.synchronize(dev)                                             # the uploads ran on torch's stream, the grouping runs on the handle's
    n_pts, _, grouped = tri.group_table_device(table.shape[0], d_cam.data_ptr(), d_feat.data_ptr(), d_uv.data_ptr(), n_features)
    if not grouped:
        rec, start = group_reconstructable(table)
        return nb_triangulate_full(rec, P, start, intr, D, device=device)
    if n_pts == 0:
        return np.empty((0, 3))
    tri.run()
    pts = tri.points()
    last_triangulate_kernel_ms = tri.last_kernel_ms()
    return pts


def group_reconstructable(data: np.ndarray):
    """The grouping CameraSet.multi_cam_triangulate does before calling nb_triangulate_full
    (cameras/camera_set.py:371-378): keep (image, key) groups seen by more than one camera, in the
    table's order, and return (reconstructable_data, start_ind)."""
    table = np.asarray(data, dtype=np.float64)
    if table.shape[0] == 0:
        return table, np.zeros(1, dtype=np.int64)
    # one id per (image, key...) feature; how many cameras saw it; where it first appears in the table
    _, feature = np.unique(table[:, 1:-2], axis=0, return_inverse=True)
    feature = feature.reshape(-1)
    seen_by = np.bincount(feature)
    keep = seen_by[feature] >= 2                    # one view cannot be triangulated
    kept = table[keep]
    kept_feature = feature[keep]
    first_row = np.full(seen_by.shape[0], kept.shape[0], dtype=np.int64)
    np.minimum.at(first_row, kept_feature, np.arange(kept.shape[0]))
    in_table_order = np.argsort(first_row[seen_by >= 2], kind="stable")
    sizes = seen_by[seen_by >= 2][in_table_order]
    starts = np.zeros(sizes.shape[0] + 1, dtype=np.int64)
    np.cumsum(sizes, out=starts[1:])
    return kept, starts
