#!/usr/bin/env python3
"""Where one device LM iteration spends its time (developer tool, one MI355X): phases of BlockedNormalEquations.solve and of the
loop in device_solver._lm_solve_blocked, timed with HIP events on the rig of a BASELINE config; then the whole solve.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel table (profiles/r03/lm_rig32_kernel_stats.csv)."""
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
from pycamset_amd.engine import schur_syrk, schur_vtx
from tools.library_solver import library_schur_solve


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
    lm_solve(h, x0.copy(), max_iter=2)     # warm-up (rocSOLVER / rocBLAS start-up)
    eng = h.op_fun.engine
    ne = BlockedNormalEquations(eng, h._jac_mask())
    torch.cuda.set_stream(ne.stream)       # one real stream for torch operations and C-ABI kernels (BlockedNormalEquations.on_stream)
    ps = torch.from_numpy(h.op_fun.build_param_list(*h.get_bundle_adjustment_inputs(x0))).cuda()
    lam = torch.full((1,), 1e-3, dtype=torch.float64, device="cuda")
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    def timed(fn, reps=a.reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3, (time.perf_counter() - t0) / reps * 1e6

    print(f"# {rig.name} chain {a.chain}: N = {rig.n_det}, n_lead {ne.n_lead}, n_trail {ne.n_trail}, tb {ne.tb}   (GPU us per call | host wall us per call)")
    stream = torch.cuda.current_stream().cuda_stream
    V = ne.V[:, : ne.n_trail]
    phases = {
        "build (normal_blocks_device)": lambda: ne.build(ps, 0),
        "schur_prepare (3 kernels)": lambda: ne.eng.schur_prepare(ne.packed[0].data_ptr(), ne.fixed.data_ptr(), lam.data_ptr(), ne.linvt.data_ptr(), ne.u.data_ptr(),
                                                                 ne.V.data_ptr(), ne.S.data_ptr(), ne.rhs.data_ptr(), ne.dvec.data_ptr(), ne.gm.data_ptr(),
                                                                 ne.status.data_ptr(), stream),
        "S -= V V' (addmm)": lambda: ne.S.addmm_(V, V.T, alpha=-1.0),
        "rhs += V u (addmv)": lambda: ne.rhs.addmv_(V, ne.u[: ne.n_trail]),
        "pcs_schur_syrk (S -= V V', rhs += V u)": lambda: schur_syrk(0, ne.n_lead, ne.n_trail, ne.V.data_ptr(), ne.V.shape[1], ne.S.data_ptr(), ne.n_lead,
                                                                    ne.u.data_ptr(), ne.rhs.data_ptr(), stream),
        "pcs_schur_vtx (w = V' x)": lambda: schur_vtx(0, ne.n_lead, ne.n_trail, ne.V.data_ptr(), ne.V.shape[1], ne.xl.data_ptr(), ne.w.data_ptr(), stream),
    }
    for name, fn in phases.items():
        g, w = timed(fn)
        print(f"  {name:34s} {g:9.1f} | {w:9.1f}")
    ne.build(ps, 0)
    library_schur_solve(ne, 0, lam)        # leaves ne.S = the reduced matrix (the HIP solver factors it in place)
    S0 = ne.S.clone()
    S0 = torch.tril(S0) + torch.tril(S0, -1).T
    g, w = timed(lambda: torch.linalg.cholesky_ex(S0))
    print(f"  {'cholesky_ex(S)':34s} {g:9.1f} | {w:9.1f}")
    L, _ = torch.linalg.cholesky_ex(S0)
    rhs = ne.rhs.clone()
    g, w = timed(lambda: torch.cholesky_solve(rhs.unsqueeze(1), L))
    print(f"  {'cholesky_solve':34s} {g:9.1f} | {w:9.1f}")
    g, w = timed(lambda: torch.linalg.solve_triangular(L, rhs.unsqueeze(1), upper=False))
    print(f"  {'solve_triangular (one of two)':34s} {g:9.1f} | {w:9.1f}")
    xl = torch.cholesky_solve(rhs.unsqueeze(1), L).squeeze(1)
    g, w = timed(lambda: torch.mv(V.T, xl))
    print(f"  {'w = V^T x_l (mv)':34s} {g:9.1f} | {w:9.1f}")
    from pycamset_amd.engine import dense_spd_solve
    S1 = S0.clone()

    def hip_solve():
        S1.copy_(S0)
        dense_spd_solve(0, ne.n_lead, S1.data_ptr(), ne.n_lead, rhs.data_ptr(), ne.xl.data_ptr(), ne.chol_work.data_ptr(), ne.status.data_ptr(), stream)

    g, w = timed(hip_solve)
    g0, _ = timed(lambda: S1.copy_(S0))
    print(f"  {'pcs_dense_spd_solve (HIP)':34s} {g - g0:9.1f} | {w:9.1f}   (incl. a {g0:.1f} us copy of S in the host figure)")
    err = float((ne.xl - xl).abs().max() / xl.abs().max())
    print(f"  HIP solve vs rocSOLVER solution: max rel diff {err:.2e}")
    for ds, fn in (("hip", lambda: ne.solve(0, lam)), ("rocsolver", lambda: library_schur_solve(ne, 0, lam))):
        g, w = timed(fn)
        print(f"  {'solve() as a whole, ' + ds:34s} {g:9.1f} | {w:9.1f}")
    # one trial of the loop, section by section (host wall with a synchronize after each section)
    def wall(fn, reps=10):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    ps_new = torch.empty_like(ps)
    stats = torch.zeros(12, dtype=torch.float64, device="cuda")
    ne.build(ps, 1)

    def decide():
        ne.decide(0, 1, ps, lam.clone(), stats)
        return stats.cpu().numpy()

    print(f"  loop sections (host wall us): solve + trial string {wall(lambda: ne.solve(0, lam, ps, ps_new)):.0f}, build {wall(lambda: ne.build(ps_new, 1)):.0f}, "
          f"decision + read-back {wall(decide):.0f}")
    for it in (30,):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = lm_solve(h, x0.copy(), max_iter=it)
        dt = time.perf_counter() - t0
        print(f"  lm_solve: {dt * 1e3:.2f} ms for {res.nfev} evaluations ({dt / res.nfev * 1e3:.3f} ms each), cost {res.cost:.6e}, {res.message}")


if __name__ == "__main__":
    main()
