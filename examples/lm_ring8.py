#!/usr/bin/env python3
"""End-to-end use of the drop-in boundary on the ring-8 rig (BASELINE config 2, 102 400 detections):

  (a) scipy.optimize.least_squares driven by the HIP closures — the reference's own call
      (pyCamSet optimisation_handling.py:88-98), Jacobian shipped to the host as CSR every iteration;
  (b) pycamset_amd.device_solver.lm_solve — the Jacobian never leaves the GPU (matrix-free products);
  (c) the same with block-reduced normal equations and a Schur-complement Cholesky step.

    python examples/lm_ring8.py [config] [--device-only] [--self]     # --self: self-calibration chain

Needs an MI355X.  The CPU oracle is not used here.
"""
import sys
import time
from pathlib import Path

import numpy as np
from scipy.optimize import least_squares

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pycamset_amd import handlers, synthetic
from pycamset_amd.detections import TargetDetection
from pycamset_amd.device_solver import lm_solve


class Camset:
    def __init__(self, n):
        self.names = [f"cam_{i}" for i in range(n)]

    def get_names(self):
        return list(self.names)

    def get_n_cams(self):
        return len(self.names)


class Target:
    def __init__(self, pts):
        self.point_data = np.asarray(pts)[None]


def main(config=2, max_nfev=15, device_only=False, self_cal=False):
    rig = synthetic.config_rig(config)
    cs = Camset(rig.n_cams)
    cls = handlers.SelfBundleHandler if self_cal else handlers.TemplateBundleHandler
    h = cls(cs, Target(rig.points), TargetDetection(cs.get_names(), rig.detections),
            fixed_params={"cam_0": {"ext": rig.extr_true[0].copy()}},
            options={"verbosity": 0, "max_nfev": max_nfev}, pinned_ring=3)
    bp = h.bundlePrimitive
    parts = [rig.intr[bp.intr_unfixed].ravel(), rig.extr[bp.extr_unfixed].ravel(), rig.poses[bp.poses_unfixed].ravel()]
    if self_cal:
        parts.append(rig.points.ravel()[bp.bdpt_unfixed])
    x0 = np.concatenate(parts)
    loss_fn, jac_fn = h.make_loss_fun(1), h.make_loss_jac(1)
    e0 = np.mean(np.linalg.norm(loss_fn(x0).reshape(-1, 2), axis=1))
    print(f"{rig.name}: N = {rig.n_det}, free parameters = {x0.size}, initial mean reprojection error {e0:.3f} px")

    if not device_only:
        t0 = time.perf_counter()
        res = least_squares(loss_fn, x0.copy(), jac=jac_fn, x_scale="jac", max_nfev=max_nfev, verbose=0)
        t_scipy = time.perf_counter() - t0
        e1 = np.mean(np.linalg.norm(res.fun.reshape(-1, 2), axis=1))
        print(f"(a) least_squares + HIP closures : {t_scipy:7.2f} s  nfev {res.nfev:3d}  cost {res.cost:.6e}  error {e1:.4f} px")

    t0 = time.perf_counter()
    dev = lm_solve(h, x0.copy(), max_iter=max_nfev, linear_solver="pcg")
    t_dev = time.perf_counter() - t0
    e2 = np.mean(np.linalg.norm(loss_fn(dev.x).reshape(-1, 2), axis=1))
    print(f"(b) device LM (matrix-free J)    : {t_dev:7.2f} s  nfev {dev.nfev:3d}  cost {dev.cost:.6e}  error {e2:.4f} px"
          f"  ({dev.n_jtjv} J^T J v products, status: {dev.message})")

    t0 = time.perf_counter()
    ch = lm_solve(h, x0.copy(), max_iter=max_nfev, linear_solver="cholesky")
    t_ch = time.perf_counter() - t0
    e3 = np.mean(np.linalg.norm(loss_fn(ch.x).reshape(-1, 2), axis=1))
    print(f"(c) device LM (block-reduced J^T J + Cholesky): {t_ch:7.2f} s  nfev {ch.nfev:3d}  cost {ch.cost:.6e}  error {e3:.4f} px"
          f"  ({ch.n_jtjv} factorisations, status: {ch.message})")
    t0 = time.perf_counter()
    ch = lm_solve(h, x0.copy(), max_iter=max_nfev, linear_solver="cholesky")
    print(f"    second run (rocSOLVER warmed up): {time.perf_counter() - t0:7.2f} s")


if __name__ == "__main__":
    nums = [a for a in sys.argv[1:] if not a.startswith("--")]
    main(int(nums[0]) if nums else 2, device_only="--device-only" in sys.argv, self_cal="--self" in sys.argv)
