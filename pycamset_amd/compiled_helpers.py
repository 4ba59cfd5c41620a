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


def nb_triangulate_full(data, proj, start_inds, intr, dist, device: int = 0) -> np.ndarray:
    """``data`` rows = [cam, ..., u, v] sorted by point; ``start_inds`` (n_pts + 1) like the reference
    (compiled_helpers.py:609-643).  Returns (n_pts, 3)."""
    import ctypes

    from . import _capi

    global last_triangulate_kernel_ms
    data = np.asarray(data, dtype=np.float64)
    cam = np.ascontiguousarray(data[:, 0].astype(np.int32))
    uv = np.ascontiguousarray(data[:, -2:])
    start = np.ascontiguousarray(start_inds, dtype=np.int64)
    P = np.ascontiguousarray(proj, dtype=np.float64)
    K = np.ascontiguousarray(intr, dtype=np.float64)
    D = np.ascontiguousarray(np.asarray(dist, dtype=np.float64).reshape(P.shape[0], -1))
    if P.shape[1:] != (3, 4) or K.shape != (P.shape[0], 3, 3) or D.shape[1] != 5:
        raise ValueError("expected proj (C,3,4), intr (C,3,3), dist (C,5)")
    n_pts = start.shape[0] - 1
    pts = np.empty((max(n_pts, 0), 3))
    ms = ctypes.c_float(0.0)
    dp, ip, lp = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
    _capi.check(_capi.lib().pcs_triangulate(device, cam.shape[0], cam.ctypes.data_as(ip), uv.ctypes.data_as(dp), n_pts,
                                            start.ctypes.data_as(lp), P.shape[0], P.ctypes.data_as(dp), K.ctypes.data_as(dp),
                                            D.ctypes.data_as(dp), pts.ctypes.data_as(dp), ctypes.byref(ms)))
    last_triangulate_kernel_ms = float(ms.value)
    return pts


def group_reconstructable(data: np.ndarray):
    """The grouping CameraSet.multi_cam_triangulate does before calling nb_triangulate_full
    (cameras/camera_set.py:371-378): keep (image, key) groups seen by more than one camera, in the
    table's order, and return (reconstructable_data, start_ind)."""
    data = np.asarray(data, dtype=np.float64)
    _, inv, count = np.unique(data[:, 1:-2], axis=0, return_inverse=True, return_counts=True)
    viable_mask = count > 1
    reconstructable_data = data[viable_mask[inv].squeeze()]
    _, im_index, im_counts = np.unique(reconstructable_data[:, 1:-2], axis=0, return_index=True, return_counts=True)
    start_ind = np.append(0, np.cumsum(im_counts[np.argsort(im_index)]))
    return reconstructable_data, start_ind
