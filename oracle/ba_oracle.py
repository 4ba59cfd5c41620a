"""Python face of the CPU oracle (TEST INFRASTRUCTURE — see ba_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package ``pycamset_amd`` never does.

Floating-point work is in ba_oracle.c (ctypes); the integer/index bookkeeping of the reference
(param-string layout, per-detection column table, static CSR structure, fixed-parameter masking)
is restated here in NumPy.  Citations: afb = pyCamSet/optimisation/abstract_function_blocks.py,
th = template_handler.py, sbh = standard_bundle_handler.py, fph = free_point_handler.py.

Parity status: PINNED against tests/golden/*.npz (generated from the reference itself).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
CHAINS = {"template": 0, "self": 1, "free": 2}
CHAIN_P = {"template": 21, "self": 24, "free": 18}

_c_double_p = ctypes.POINTER(ctypes.c_double)
_libs: dict[str, ctypes.CDLL] = {}


def build(force: bool = False) -> None:
    """Compile oracle/ba_oracle.c (gcc) when the shared objects are missing or older than the source."""
    src = (_HERE / "ba_oracle.c").stat().st_mtime
    libs = [_HERE / "libba_oracle.so", _HERE / "libba_oracle_fast.so"]
    if force or any(not p.exists() or p.stat().st_mtime < src for p in libs):
        subprocess.run(["make", "-C", str(_HERE)] + (["-B"] if force else []), check=True, capture_output=True)


def _lib(fast: bool = False) -> ctypes.CDLL:
    name = "libba_oracle_fast.so" if fast else "libba_oracle.so"
    if name not in _libs:
        build()
        lib = ctypes.CDLL(str(_HERE / name))
        i64, dp, ci = ctypes.c_int64, _c_double_p, ctypes.c_int
        lib.orc_full_loss.argtypes = [ci, i64, dp, dp, i64, i64, dp, dp, ci]
        lib.orc_full_loss.restype = ci
        lib.orc_full_jac.argtypes = [ci, i64, dp, dp, i64, i64, dp, dp, dp, ci]
        lib.orc_full_jac.restype = ci
        lib.orc_max_threads.restype = ci
        lib.orc_legacy_cost.argtypes = [i64, dp, dp, i64, dp, dp, dp, dp, ci]
        lib.orc_legacy_cost.restype = ci
        for fn, n in (("orc_rodrigues", 2), ("orc_rodrigues_jac", 2), ("orc_e4x4_flat", 2), ("orc_htform", 3),
                      ("orc_projection_fun", 3), ("orc_projection_jac", 3), ("orc_rigid_fun", 3),
                      ("orc_rigid_jac", 3), ("orc_template_jac", 3)):
            getattr(lib, fn).argtypes = [dp] * n
            getattr(lib, fn).restype = None
        _libs[name] = lib
    return _libs[name]


def _p(a: np.ndarray):
    return a.ctypes.data_as(_c_double_p)


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- unit functions -------------------------------------------------
def call_unit(name: str, out_len: int, *arrays) -> np.ndarray:
    """Call one of the small C functions ``orc_<name>(in..., out)``."""
    ins = [_f64(a) for a in arrays]
    out = np.empty(out_len)
    getattr(_lib(), f"orc_{name}")(*[_p(a) for a in ins], _p(out))
    return out


# ---- param-string layout and static structure (a11, a12) -----------------------------------
def counts_from_detections(det: np.ndarray) -> tuple[int, int, int]:
    """make_param_struct counts: max index + 1 per link type (afb:793-795)."""
    return int(det[:, 0].max()) + 1, int(det[:, 1].max()) + 1, int(det[:, 2].max()) + 1


def param_struct(chain: str, det: np.ndarray, counts=None):
    """(starts, n_params, total) of the unique parameter groups in block order (afb:804-820).
    ``counts`` = (n_cams, n_imgs, n_keys) replaces the reference's max-index+1 rule (afb:793-795)."""
    C, I, K = counts if counts is not None else counts_from_detections(det)
    groups = {"template": [(9, C), (6, C), (6, I)],
              "self": [(9, C), (6, C), (6, I), (3, K)],
              "free": [(9, C), (6, C), (3, K)]}[chain]
    starts, total = [], 0
    for n, cnt in groups:
        starts.append(total)
        total += n * cnt
    return np.array(starts), np.array([g[0] for g in groups]), total


def block_param_inds(chain: str, det: np.ndarray, counts=None) -> np.ndarray:
    """Per-detection global column of every local parameter, (N, P) (afb:211-217, unthreaded)."""
    starts, npar, _ = param_struct(chain, det, counts)
    key_col = {"template": [0, 0, 1], "self": [0, 0, 1, 2], "free": [0, 0, 2]}[chain]
    cols = []
    for s, n, kc in zip(starts, npar, key_col):
        idx = det[:, kc].astype(np.int64)
        cols.append(s + idx[:, None] * n + np.arange(n)[None, :])
    return np.concatenate(cols, axis=1)


def csr_structure(chain: str, det: np.ndarray, unfixed: np.ndarray, counts=None):
    """make_jac_CSR_columns_row_pointers (afb:465-489): static indices / indptr."""
    c = np.repeat(block_param_inds(chain, det, counts), 2, axis=0)
    unfixed = np.asarray(unfixed, dtype=bool)
    conversion = np.concatenate([[0], np.cumsum(unfixed)])
    mask = unfixed[c]
    indices = conversion[c][mask]
    indptr = np.concatenate([[0], np.cumsum(mask.sum(axis=1))])
    return indices.astype(np.int64), indptr.astype(np.int64), mask


def build_param_list(*arrays) -> np.ndarray:
    """afb:669-681: flatten + concatenate in block order."""
    return np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in arrays])


# ---- drivers (a9, a10) ---------------------------------------------------------------------
def full_loss(chain: str, det: np.ndarray, param_str: np.ndarray, template=None, threads: int = 1,
              fast: bool = False, counts=None) -> np.ndarray:
    """``counts`` = (n_cams, n_imgs[, n_keys]) overrides the max-index+1 rule (for evaluating a
    slice of a larger table against the full parameter string)."""
    det, param_str = _f64(det), _f64(param_str)
    C, I = counts[:2] if counts is not None else counts_from_detections(det)[:2]
    t = _f64(template) if template is not None else None
    out = np.empty((det.shape[0], 2))
    rc = _lib(fast).orc_full_loss(CHAINS[chain], det.shape[0], _p(det), _p(param_str), C, I,
                                  _p(t) if t is not None else None, _p(out), threads)
    if rc:
        raise ValueError("orc_full_loss: bad arguments")
    return out


def full_jac_dense(chain: str, det: np.ndarray, param_str: np.ndarray, template=None, threads: int = 1,
                   fast: bool = False, with_resid: bool = False, counts=None):
    """Dense (2N, P) block rows = generated full_jac output after afb:641."""
    det, param_str = _f64(det), _f64(param_str)
    C, I = counts[:2] if counts is not None else counts_from_detections(det)[:2]
    t = _f64(template) if template is not None else None
    P = CHAIN_P[chain]
    dense = np.empty((2 * det.shape[0], P))
    res = np.empty((det.shape[0], 2)) if with_resid else None
    rc = _lib(fast).orc_full_jac(CHAINS[chain], det.shape[0], _p(det), _p(param_str), C, I,
                                 _p(t) if t is not None else None, _p(dense), _p(res) if with_resid else None, threads)
    if rc:
        raise ValueError("orc_full_jac: bad arguments")
    return (dense, res) if with_resid else dense


def jac_csr(chain: str, det: np.ndarray, param_str: np.ndarray, template=None, unfixed=None, threads: int = 1):
    """jac_fn wrapper (afb:627-652): (data, indices, indptr) with fixed columns dropped."""
    dense = full_jac_dense(chain, det, param_str, template, threads)
    if unfixed is None:
        unfixed = np.ones(param_struct(chain, det)[2], dtype=bool)
    indices, indptr, mask = csr_structure(chain, det, unfixed)
    return dense[mask], indices, indptr


# ---- legacy residual-only cost -----------------------------------------------------
def legacy_cost(dct, im_points, projection_matrixes, intrinsics, dists, threads: int = 1, fast: bool = False) -> np.ndarray:
    """numpy_bundle_adjustment_costfn (compiled_helpers.py:518-547): (2N,) errors."""
    dct, im = _f64(dct), _f64(im_points)
    n_keys = int(np.prod(im.shape[1:-1]))
    P, Kc, D = _f64(projection_matrixes), _f64(intrinsics), _f64(dists)
    out = np.empty(2 * dct.shape[0])
    _lib(fast).orc_legacy_cost(dct.shape[0], _p(dct), _p(im), n_keys, _p(P), _p(Kc), _p(D), _p(out), threads)
    return out


def legacy_inputs(intr, extr, poses, points):
    """Build the legacy cost's inputs from the parameter slabs with the oracle's own SE3 helpers:
    im_points (I,K,3) like template_handler.py:231-237, projection matrices K [R|t] (C,3,4),
    intrinsic matrices (C,3,3), distortion (C,5) = [k0,k1,p0,p1,k2]."""
    intr, extr, poses, points = _f64(intr), _f64(extr), _f64(poses), _f64(points)
    C, I, K = intr.shape[0], poses.shape[0], points.shape[0]
    im_points = np.empty((I, K, 3))
    for i in range(I):
        T = call_unit("e4x4_flat", 12, poses[i])
        for k in range(K):
            im_points[i, k] = call_unit("htform", 3, points[k], T)
    Kc = np.zeros((C, 3, 3))
    Kc[:, 0, 0], Kc[:, 0, 2], Kc[:, 1, 1], Kc[:, 1, 2], Kc[:, 2, 2] = intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], 1.0
    proj = np.empty((C, 3, 4))
    for c in range(C):
        T = call_unit("e4x4_flat", 12, extr[c])
        Rt = np.concatenate([T[:9].reshape(3, 3), T[9:].reshape(3, 1)], axis=1)
        proj[c] = Kc[c] @ Rt
    return im_points, proj, Kc, np.ascontiguousarray(intr[:, 4:9])


# ---- batched triangulation -----------------------------------------------------------
def undistort(pts, intrinsics, dist_coef):
    """nb_undistort (compiled_helpers.py:409-431): 5 fixed-point iterations, one point (2,)."""
    centre = intrinsics[:2, -1]
    focal = np.diag(intrinsics)[:2]
    x0, y0 = (pts - centre) / focal
    k = np.reshape(dist_coef, (-1))
    x, y = x0, y0
    for _ in range(5):
        r2 = x ** 2 + y ** 2
        k_inv = 1 / (1 + k[0] * r2 + k[1] * (r2 ** 2) + k[4] * (r2 ** 3))
        xD = 2 * k[2] * x * y + k[3] * (r2 + 2 * (x ** 2))
        yD = k[2] * (r2 + 2 * (y ** 2)) + 2 * k[3] * x * y
        x = (x0 - xD) * k_inv
        y = (y0 - yD) * k_inv
    return np.array([x, y]) * focal + centre


def triangulate_nviews(P, ip):
    """nb_triangulate_nviews (compiled_helpers.py:645-663): DLT matrix + LAPACK SVD."""
    n = len(P)
    M = np.zeros((3 * n, 4 + n))
    for i, (x, p) in enumerate(zip(ip, P)):
        M[3 * i:3 * i + 3, :4] = p
        M[3 * i:3 * i + 3, 4 + i] = -x
    V = np.linalg.svd(M, full_matrices=False)[-1]
    X = V[-1, :4]
    return X[:3] / X[3]


def triangulate_full(data, proj, start_inds, intr, dist):
    """nb_triangulate_full (compiled_helpers.py:609-643)."""
    data = np.asarray(data, dtype=np.float64)
    pts = np.empty((len(start_inds) - 1, 3))
    for idx in range(len(start_inds) - 1):
        s, e = int(start_inds[idx]), int(start_inds[idx + 1])
        uvh = np.empty((e - s, 3))
        Ps = np.empty((e - s, 3, 4))
        for t in range(e - s):
            datum = data[s + t]
            cam = int(datum[0])
            ud = undistort(datum[-2:], intr[cam], dist[cam])
            uvh[t] = [ud[0], ud[1], 1]
            Ps[t] = proj[cam]
        pts[idx] = triangulate_nviews(Ps, uvh)
    return pts


# ---- handler-level x -> slabs (a13, a14) ----------------------------------------------------
def scatter_free(x_part: np.ndarray, full: np.ndarray, unfixed: np.ndarray) -> None:
    """ch.fill_flat (compiled_helpers.py:155-177): write free rows/scalars into the full slab."""
    full[np.asarray(unfixed, dtype=bool)] = x_part


def max_threads() -> int:
    return int(_lib(True).orc_max_threads())


def cpu_count() -> int:
    return os.cpu_count() or 1
