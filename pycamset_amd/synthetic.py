"""Seeded synthetic calibration rigs.

Pure-NumPy *data generators*: they build the detection table ``(N,5) = [cam, im, key, u, v]``
(layout of pyCamSet ``TargetDetection.return_flattened_keys().get_data()``,
calibration_targets/target_detections.py:51-55, :333-351), the parameter slabs and the template
points that the cost/Jacobian path consumes.  Nothing here is on the measured path; the forward
model below exists only to place the synthetic ``u, v`` measurements near the true projection.

Conventions (same as the reference's function blocks, function_block_implementations.py):
  intr  (n_cams, 9)  = [fx, px, fy, py, k0, k1, p0, p1, k2]            (fbi:31, fbi:54)
  extr  (n_cams, 6)  = [rotvec(3), t(3)]  world -> camera               (fbi:184-185)
  poses (n_imgs, 6)  = [rotvec(3), t(3)]  target -> world               (fbi:188-192)
  points (n_keys, 3)                                                     (target.point_data)
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
from scipy.spatial.transform import Rotation

CHAIN_TEMPLATE = "template"   # projection + extrinsic3D + template_points   (template_handler.py:152)
CHAIN_SELF = "self"           # projection + extrinsic3D + rigidTform3d + free_point (standard_bundle_handler.py:182)
CHAIN_FREE = "free"           # projection + extrinsic3D + free_point         (free_point_handler.py:143)


@dataclass
class SyntheticRig:
    name: str
    detections: np.ndarray          # (N,5) float64  [cam, im, key, u, v], ordered cam -> im -> key
    intr: np.ndarray                # (C,9) evaluation point
    extr: np.ndarray                # (C,6)
    poses: np.ndarray               # (I,6)  pose 0 == 0 exactly
    points: np.ndarray              # (K,3)  template / free points
    intr_true: np.ndarray = field(repr=False, default=None)
    extr_true: np.ndarray = field(repr=False, default=None)
    poses_true: np.ndarray = field(repr=False, default=None)
    points_true: np.ndarray = field(repr=False, default=None)

    @property
    def n_cams(self):
        return self.intr.shape[0]

    @property
    def n_imgs(self):
        return self.poses.shape[0]

    @property
    def n_keys(self):
        return self.points.shape[0]

    @property
    def n_det(self):
        return self.detections.shape[0]


# --------------------------------------------------------------------------------------
# target geometries
# --------------------------------------------------------------------------------------
def charuco_points(num_squares: int = 17, square_mm: float = 4.0) -> np.ndarray:
    """Inner chessboard corners of a planar ChArUco board, shape ((n-1)^2, 3), metres.
    Geometry of calibration_targets/target_charuco.py:39-42 (cv2 CharucoBoard corners:
    row-major grid at multiples of the square length, z = 0)."""
    n = num_squares - 1
    s = square_mm / 1000.0
    jj, ii = np.meshgrid(np.arange(1, n + 1), np.arange(1, n + 1), indexing="xy")
    pts = np.stack([jj.ravel() * s, ii.ravel() * s, np.zeros(n * n)], axis=1)
    return pts - pts.mean(axis=0)  # centred so every ring camera sees it near the axis


def ccube_points(n_points: int = 10, length_mm: float = 40.0) -> np.ndarray:
    """6 faces x (n_points-1)^2 corner grid on a cube, shape (6*(n-1)^2, 3), metres —
    the *shape* of Ccube.point_data (calibration_targets/target_Ccube.py:199-206, :227-244)."""
    n = n_points - 1
    L = length_mm / 1000.0
    g = (np.arange(1, n + 1) / n_points - 0.5) * L
    a, b = np.meshgrid(g, g, indexing="ij")
    a, b = a.ravel(), b.ravel()
    h = np.full_like(a, L / 2)
    faces = [
        np.stack([a, b, h], 1), np.stack([a, b, -h], 1),
        np.stack([a, h, b], 1), np.stack([a, -h, b], 1),
        np.stack([h, a, b], 1), np.stack([-h, a, b], 1),
    ]
    return np.concatenate(faces, axis=0)


# --------------------------------------------------------------------------------------
# forward model used only to place measurements
# --------------------------------------------------------------------------------------
def _rot(rv: np.ndarray) -> np.ndarray:
    return Rotation.from_rotvec(rv).as_matrix()


def project_dense(intr, extr, poses, points):
    """uv[c, i, k, :] for every camera / image / key (vectorised NumPy)."""
    Rp = _rot(poses[:, :3])                                   # (I,3,3)
    Xw = np.einsum("iab,kb->ika", Rp, points) + poses[:, None, 3:]          # (I,K,3)
    Re = _rot(extr[:, :3])                                    # (C,3,3)
    Xc = np.einsum("cab,ikb->cika", Re, Xw) + extr[:, None, None, 3:]       # (C,I,K,3)
    fx, px, fy, py = (intr[:, j][:, None, None] for j in range(4))
    k0, k1, p0, p1, k2 = (intr[:, j][:, None, None] for j in range(4, 9))
    x = Xc[..., 0] / Xc[..., 2]
    y = Xc[..., 1] / Xc[..., 2]
    r2 = x * x + y * y
    kup = 1 + k0 * r2 + k1 * r2 ** 2 + k2 * r2 ** 3
    xd = x * kup + 2 * p0 * x * y + p1 * (r2 + 2 * x * x)
    yd = y * kup + p0 * (r2 + 2 * y * y) + 2 * p1 * x * y
    return np.stack([xd * fx + px, yd * fy + py], axis=-1), Xc[..., 2]


def _ring_extrinsics(n_cams: int, rng, n_rings: int = 1, radius: float = 0.2) -> np.ndarray:
    """Cameras on ring(s) looking at the origin: rotvec (0, 2*pi*b/n, 0), t (0, 0, radius)
    as examples/make_camera_ring.py:7-16; further rings are tilted about x."""
    per = n_cams // n_rings
    ext = np.zeros((n_cams, 6))
    c = 0
    for ring in range(n_rings):
        tilt = 0.45 * ring
        m = per if ring < n_rings - 1 else n_cams - c
        for b in range(m):
            R = Rotation.from_rotvec([tilt, 0, 0]) * Rotation.from_rotvec([0, 2 * np.pi * (b + 0.5 * ring) / m, 0])
            ext[c, :3] = R.as_rotvec()
            ext[c, 3:] = [0, 0, radius]
            c += 1
    ext[:, :3] += rng.normal(0, 0.01, (n_cams, 3))
    ext[:, 3:] += rng.normal(0, 0.002, (n_cams, 3))
    return ext


def make_rig(name: str, n_cams: int, n_imgs: int, points: np.ndarray, *, seed: int,
             visibility: float = 1.0, n_rings: int = 1, noise_px: float = 0.3,
             perturb: float = 0.01, order: str = "cam") -> SyntheticRig:
    """Build one seeded rig.  ``order='cam'`` sorts the table cam -> image -> key like
    calibration/camera_calibrator.py:314-317; ``order='im'`` sorts image -> cam -> key
    (used to give each rank of a sharded run a contiguous pose range)."""
    rng = np.random.default_rng(seed)
    K = points.shape[0]
    intr = np.empty((n_cams, 9))
    intr[:, 0] = rng.uniform(900, 1100, n_cams)
    intr[:, 2] = rng.uniform(900, 1100, n_cams)
    intr[:, 1] = 500 + rng.uniform(-20, 20, n_cams)
    intr[:, 3] = 500 + rng.uniform(-20, 20, n_cams)
    intr[:, 4] = rng.normal(0, 0.05, n_cams)
    intr[:, 5] = rng.normal(0, 0.01, n_cams)
    intr[:, 6] = rng.normal(0, 1e-3, n_cams)
    intr[:, 7] = rng.normal(0, 1e-3, n_cams)
    intr[:, 8] = rng.normal(0, 1e-3, n_cams)
    extr = _ring_extrinsics(n_cams, rng, n_rings=n_rings)
    poses = np.concatenate([rng.normal(0, 0.15, (n_imgs, 3)), rng.normal(0, 0.01, (n_imgs, 3))], axis=1)
    poses[0] = 0.0  # fixed_pose = 0 is zeroed and fixed (template_handler.py:134-137)

    if visibility >= 1.0:
        vis = np.ones((n_cams, n_imgs, K), dtype=bool)
    else:
        vis = rng.random((n_cams, n_imgs, K)) < visibility
        vis[-1, -1, -1] = True  # last cam / image / key always observed (SURVEY 8a quirk ii)
        vis[0, 0, 0] = True
    if order == "cam":
        c_idx, i_idx, k_idx = np.nonzero(vis)
    else:
        i_idx, c_idx, k_idx = np.nonzero(np.transpose(vis, (1, 0, 2)))

    # project only what is visible, in chunks, to bound memory at 1e7 detections
    uv = np.empty((c_idx.size, 2))
    Rp = _rot(poses[:, :3])
    Re = _rot(extr[:, :3])
    step = 1 << 20
    for s in range(0, c_idx.size, step):
        sl = slice(s, s + step)
        c, i, k = c_idx[sl], i_idx[sl], k_idx[sl]
        Xw = np.einsum("nab,nb->na", Rp[i], points[k]) + poses[i, 3:]
        Xc = np.einsum("nab,nb->na", Re[c], Xw) + extr[c, 3:]
        x = Xc[:, 0] / Xc[:, 2]
        y = Xc[:, 1] / Xc[:, 2]
        r2 = x * x + y * y
        kup = 1 + intr[c, 4] * r2 + intr[c, 5] * r2 ** 2 + intr[c, 8] * r2 ** 3
        xd = x * kup + 2 * intr[c, 6] * x * y + intr[c, 7] * (r2 + 2 * x * x)
        yd = y * kup + intr[c, 6] * (r2 + 2 * y * y) + 2 * intr[c, 7] * x * y
        uv[sl, 0] = xd * intr[c, 0] + intr[c, 1]
        uv[sl, 1] = yd * intr[c, 2] + intr[c, 3]
    uv += rng.normal(0, noise_px, uv.shape)
    det = np.empty((c_idx.size, 5))
    det[:, 0], det[:, 1], det[:, 2] = c_idx, i_idx, k_idx
    det[:, 3:] = uv

    def jiggle(a):
        return a * (1 + perturb * rng.standard_normal(a.shape))

    rig = SyntheticRig(
        name=name, detections=det,
        intr=jiggle(intr), extr=jiggle(extr), poses=jiggle(poses), points=points.copy(),
        intr_true=intr, extr_true=extr, poses_true=poses, points_true=points.copy(),
    )
    rig.poses[0] = 0.0
    return rig


# --------------------------------------------------------------------------------------
# the BASELINE.json configs
# --------------------------------------------------------------------------------------
def config_rig(number: int, *, scale: float = 1.0, order: str = "cam", n_imgs: int | None = None,
               block: int = 0) -> SyntheticRig:
    """BASELINE.md section 3 configs; ``scale`` < 1 shrinks the visibility (smaller N, same shapes).
    ``block`` > 0 draws an independent rig of the same shape (seed + 1000 * block): rank r of a
    weak-scaling run evaluates block r."""
    if block:
        rig = _config_rig(number, scale, order, n_imgs, 1000 * block)
        rig.name += f"/block{block}"
        return rig
    return _config_rig(number, scale, order, n_imgs, 0)


def _config_rig(number, scale, order, n_imgs, seed_off) -> SyntheticRig:
    if number == 1:   # ccube-plumbing: 3 cams x 24 images, Ccube 486 keys, N ~ 7e3
        return make_rig("ccube-plumbing", 3, n_imgs or 24, ccube_points(), seed=1 + seed_off, visibility=0.2 * scale)
    if number == 2:   # ring-8: planar ChArUco 17x17 -> 256 corners, 50 poses, all visible, N = 102400
        return make_rig("ring-8", 8, n_imgs or 50, charuco_points(17, 4.0), seed=2 + seed_off,
                        visibility=1.0 if scale >= 1 else scale)
    if number in (3, 4):   # rig-32 (headline) / rig-32-self: 32 cams on two rings, Ccube, 200 poses, N ~ 1e6
        return make_rig("rig-32" if number == 3 else "rig-32-self", 32, n_imgs or 200, ccube_points(),
                        seed=number + seed_off, visibility=0.3215 * scale, n_rings=2, order=order)
    if number == 5:   # rig-128: 128 cams, 500 poses, N ~ 1e7
        return make_rig("rig-128", 128, n_imgs or 500, ccube_points(), seed=5 + seed_off, visibility=0.3215 * scale,
                        n_rings=4, order=order)
    raise ValueError(f"unknown config {number}")


def tiny_rig(seed: int = 0, n_cams: int = 3, n_imgs: int = 4, n_keys: int = 8, visibility: float = 0.95) -> SyntheticRig:
    """(3 cams, 4 images, 8 keys) fixture-sized rig."""
    rng = np.random.default_rng(1000 + seed)
    pts = rng.uniform(-0.03, 0.03, (n_keys, 3))
    return make_rig(f"tiny-{seed}", n_cams, n_imgs, pts, seed=seed + 100, visibility=visibility)
