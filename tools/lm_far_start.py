#!/usr/bin/env python3
"""How the device LM copes with starts FAR from the solution (developer tool, one MI355X) — the table of DESIGN section 4.

The default initial damping (lambda0 = 1e-6, Marquardt scaling) was chosen on starts 1 % / 7 px off the truth; real calibrations start
from PnP poses tens of pixels off.  For every case (rig x chain x start) and every damping policy the script prints the number of
evaluations the device loop needs, its final cost relative to scipy.optimize.least_squares (the reference's solver,
optimisation_handling.py:88-98) on the same HIP closures, and scipy's own evaluation count.

    starts:   near  = the rig's 1 % perturbation;  far / far10 = 5 x / 10 x that (~35 / 70 px);  swap = near with the poses of images 1 and 2 exchanged
    policies: (lambda0, factor applied by a rejection BEFORE the first accepted step, fast decrease on / off = a gain ratio above 0.95
              multiplies lambda by 0.1 instead of 1/3): (1e-5, 1e3, on) the default since round 5; (1e-6, 4, off) round 4's;
              (1e-3, 4, off) round 3's; (1e-5, 4, off) and (1e-4, 1e3, on) for comparison
"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from scipy.optimize import least_squares

from pycamset_amd import handlers, synthetic
from pycamset_amd.detections import TargetDetection
from pycamset_amd.device_solver import lm_solve


class _Camset:
    def __init__(self, n):
        self.names = [f"cam_{i}" for i in range(n)]

    def get_names(self):
        return list(self.names)

    def get_n_cams(self):
        return len(self.names)


class _Target:
    def __init__(self, pts):
        self.point_data = np.asarray(pts)[None]


def make(rig, chain):
    cs = _Camset(rig.n_cams)
    cls = {"template": handlers.TemplateBundleHandler, "self": handlers.SelfBundleHandler}[chain]
    return cls(cs, _Target(rig.points), TargetDetection(cs.get_names(), rig.detections),
               fixed_params={"cam_0": {"ext": rig.extr_true[0].copy()}}, options={"verbosity": 0})


def start_vector(h, rig, chain, kind):
    scale = {"near": 1.0, "far": 5.0, "far10": 10.0, "swap": 1.0}[kind]
    intr = rig.intr_true + scale * (rig.intr - rig.intr_true)
    extr = rig.extr_true + scale * (rig.extr - rig.extr_true)
    poses = rig.poses_true + scale * (rig.poses - rig.poses_true)
    if kind == "swap":
        poses = poses.copy()
        poses[[1, 2]] = poses[[2, 1]]
    pts = rig.points_true + scale * (rig.points - rig.points_true) if hasattr(rig, "points_true") else rig.points
    bp = h.bundlePrimitive
    parts = [intr[bp.intr_unfixed].ravel(), extr[bp.extr_unfixed].ravel(), poses[bp.poses_unfixed].ravel()]
    if chain == "self":
        parts.append(np.asarray(pts).ravel()[bp.bdpt_unfixed])
    return np.concatenate(parts)


POLICIES = [(1e-5, 1e3, True), (1e-6, 4.0, False), (1e-3, 4.0, False), (1e-5, 4.0, False), (1e-4, 1e3, True)]
FAST_ON = None


def main():
    rigs = {"ring-8-small": synthetic.make_rig("ring-8-small", 8, 12, synthetic.charuco_points(9, 8.0), seed=21, visibility=0.8),
            "config-1": synthetic.config_rig(1)}
    if "--big" in sys.argv:
        rigs["ring-8"] = synthetic.config_rig(2)
    import pycamset_amd.device_solver as ds
    global FAST_ON
    FAST_ON = ds.LAM_FAST
    print(f"{'rig':13s} {'chain':9s} {'start':5s} {'px rms':>8s} | scipy nfev cost      | " + " | ".join(f"{l:.0e} x{g:<4.0f} {'fast' if f else 'slow'}" for l, g, f in POLICIES))
    for rname, rig in rigs.items():
        for chain in ("template", "self"):
            h = make(rig, chain)
            loss_fn, jac_fn = h.make_loss_fun(1), h.make_loss_jac(1)
            for kind in ("near", "far", "far10", "swap"):
                x0 = start_vector(h, rig, chain, kind)
                px = float(np.sqrt(np.mean(loss_fn(x0) ** 2)))
                ref = least_squares(loss_fn, x0.copy(), jac=jac_fn, x_scale="jac", max_nfev=100, verbose=0)
                cells = []
                for lam0, grow, fast in POLICIES:
                    ds.LAM_FAST = FAST_ON if fast else (2.0, 1.0)          # a gain ratio never exceeds 2: the rule is off
                    t0 = time.perf_counter()
                    res = lm_solve(h, x0.copy(), max_iter=100, lam0=lam0, lam_grow0=grow)
                    dt = (time.perf_counter() - t0) * 1e3
                    cells.append(f"{res.nfev:3d} ev {res.cost / ref.cost:7.4f} {dt:5.1f}ms")
                print(f"{rname:13s} {chain:9s} {kind:5s} {px:8.2f} | {ref.nfev:5d}      {ref.cost:.4e} | " + " | ".join(cells), flush=True)
    ds.LAM_FAST = FAST_ON




def verbose_case(rname="ring-8-small", chain="template", kind="swap"):
    """python tools/lm_far_start.py --verbose: the trial-by-trial course of one case for two policies (where do the rejections fall?)"""
    rig = synthetic.make_rig("ring-8-small", 8, 12, synthetic.charuco_points(9, 8.0), seed=21, visibility=0.8) if rname == "ring-8-small" else synthetic.config_rig(1)
    h = make(rig, chain)
    h.make_loss_fun(1)
    x0 = start_vector(h, rig, chain, kind)
    for lam0, grow in ((1e-5, 1e3), (1e-6, 4.0)):
        print(f"--- {rname} {chain} {kind}: lam0 {lam0:g}, first-phase growth {grow:g}")
        lm_solve(h, x0.copy(), max_iter=100, lam0=lam0, lam_grow0=grow, verbose=1)


if __name__ == "__main__":
    if "--verbose" in sys.argv:
        verbose_case()
        verbose_case("config-1", "self", "swap")
    else:
        main()
