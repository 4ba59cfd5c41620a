"""Mirror of the two legacy cost entry points of pyCamSet/optimisation/compiled_helpers.py
(SURVEY 8 row f3), evaluated by the HIP engine:

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
