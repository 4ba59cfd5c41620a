#!/usr/bin/env python3
"""The device LM in its four forms (developer tool, one MI355X): fused / separate small kernels x atomics / deterministic sums.
Per form: the blocked build alone (HIP events), the whole solve (host wall, best of five) and its evaluations — the numbers behind
DESIGN section 4's "slowdown of the ordered mode" and "launches per trial".
    python tools/lm_modes.py --config 3 [--chain template]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from pycamset_amd import handlers, synthetic
from pycamset_amd.detections import TargetDetection
from pycamset_amd.device_solver import BlockedNormalEquations, lm_solve


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--chain", default="template")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--trace", action="store_true", help="one quiet solve per form only (for rocprofv3 --kernel-trace)")
    ap.add_argument("--forms", default="00,01,10,11", help="fused,deterministic pairs")
    ap.add_argument("--stub-collective", action="store_true",
                    help="solve through the SHARDED device-steered loop on one rank: reduce_fn = an in-stream operation that leaves the one rank's sum "
                         "unchanged (what dist.all_reduce over RCCL is to the stream) — the trace then shows build -> collective -> decision without a gap")
    a = ap.parse_args()
    rig = synthetic.config_rig(a.config)
    cs = _Camset(rig.n_cams)
    cls = {"template": handlers.TemplateBundleHandler, "self": handlers.SelfBundleHandler, "free": handlers.FreePointBundleHandler}[a.chain]
    h = cls(cs, _Target(rig.points), TargetDetection(cs.get_names(), rig.detections),
            fixed_params={"cam_0": {"ext": rig.extr_true[0].copy()}}, options={"verbosity": 0, "max_nfev": 30})
    bp = h.bundlePrimitive
    parts = [rig.intr[bp.intr_unfixed].ravel(), rig.extr[bp.extr_unfixed].ravel()]
    if a.chain != "free":
        parts.append(rig.poses[bp.poses_unfixed].ravel())
    if a.chain != "template":
        parts.append(rig.points.ravel()[bp.bdpt_unfixed])
    x0 = np.concatenate(parts)
    h.make_loss_fun(1)
    reduce_fn = None
    if a.stub_collective:
        def reduce_fn(t):
            t.mul_(1.0)
            return t
        reduce_fn.on_device = True
    lm_solve(h, x0.copy(), max_iter=2)
    eng = h.op_fun.engine
    ne = BlockedNormalEquations(eng, h._jac_mask())
    ps = torch.from_numpy(h.op_fun.build_param_list(*h.get_bundle_adjustment_inputs(x0))).cuda()
    print(f"# {rig.name} chain {a.chain}: N = {rig.n_det}, n_lead {ne.n_lead}, n_trail {ne.n_trail}")
    for form in a.forms.split(","):
        fused, det = int(form[0]), int(form[1])
        eng.set_option("fused_trial", fused)
        eng.set_option("deterministic", det)
        if a.trace:
            lm_solve(h, x0.copy(), max_iter=30, reduce_fn=reduce_fn)
            torch.cuda.synchronize()
            continue
        with torch.cuda.stream(ne.stream):
            ne.build(ps, 0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                ne.build(ps, 0)
            e1.record()
            torch.cuda.synchronize()
        build_us = e0.elapsed_time(e1) / a.reps * 1e3
        best, res = 1e9, None
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = lm_solve(h, x0.copy(), max_iter=30, reduce_fn=reduce_fn)
            best = min(best, time.perf_counter() - t0)
        print(f"  fused {fused} deterministic {det}: build {build_us:7.1f} us | lm_solve {best * 1e3:6.2f} ms for {res.nfev} evaluations "
              f"({best / res.nfev * 1e6:6.1f} us each), cost {res.cost:.9e}, {res.message}", flush=True)
    eng.set_option("fused_trial", 1)
    eng.set_option("deterministic", 0)


if __name__ == "__main__":
    main()
