#!/usr/bin/env python3
"""Levenberg-Marquardt on a GENERATED chain (developer tool, one MI355X): the exact device-steered step of round 5 (dense normal
equations from the block rows, csrc/ba_blockgram.hpp + pcs_genchain_lm_trial) against the conjugate-gradient path it replaces as the
default (products of csrc/ba_blockrow.hpp, the host between every two) and — on small rigs — scipy's trf on the same closures
(optimisation_handling.py:88-98).  The chain is `projection + extrinsic3D + rigidTform3d + board_flex`: the user-written templated
source of tests/helpers.py (a board that bends, one flex model per image) on the rigs of BASELINE's configs.
    python tools/genchain_lm.py --config 1 2 3 [--chain division] [--dense] [--trace]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch

import helpers as H
from pycamset_amd import function_blocks as fb
from pycamset_amd import handlers, synthetic
from pycamset_amd.device_solver import lm_solve


def problem(number: int, which: str):
    """`flex`: projection + extrinsic3D + rigidTform3d + board_flex (a user-written templated source, five parameters per image: the
    DENSE form of the normal equations); `division`: division_projection + extrinsic3D + template_points (a user-written lens model in
    place of the shipped projection; the chain ends in one rigid transform per image: the BLOCKED form, Schur step)."""
    rig = synthetic.config_rig(number)
    ub = H.user_blocks(fb)
    rng = np.random.default_rng(6)
    fix_ext = np.ones((rig.n_cams, 6), dtype=bool)
    fix_ext[0] = False
    det = rig.detections.copy()
    if which == "flex":
        def chain():
            return fb.projection() + fb.extrinsic3D() + fb.rigidTform3d() + ub["board_flex"]()

        flex = np.concatenate([rng.uniform(0.98, 1.02, (rig.n_imgs, 2)), rng.normal(0, 1e-3, (rig.n_imgs, 2)), rng.normal(0, 0.3, (rig.n_imgs, 1))], axis=1)
        truth = [rig.intr_true, rig.extr_true, rig.poses_true, flex]
        free_flex = np.zeros((rig.n_imgs, 5), dtype=bool)
        free_flex[:, 4] = True
        start = [rig.intr_true * (1 + 1e-3 * rng.standard_normal(rig.intr_true.shape)), rig.extr_true + 1e-3 * rng.standard_normal(rig.extr_true.shape),
                 rig.poses_true + 1e-3 * rng.standard_normal(rig.poses_true.shape), flex.copy()]
        start[3][:, 4] = 0.0
        unfixed = [None, fix_ext, None, free_flex]
    elif which == "divfree":   # the lens model AND free points (key-linked trailing entities); the poses hold the frame
        def chain():
            return ub["division_projection"]() + fb.extrinsic3D() + fb.rigidTform3d() + fb.free_point()

        div = np.concatenate([rig.intr_true[:, :4], rng.normal(0, 0.05, (rig.n_cams, 1))], axis=1)
        truth = [div, rig.extr_true, rig.poses_true, rig.points]
        start = [div * (1 + 1e-3 * rng.standard_normal(div.shape)), rig.extr_true + 1e-3 * rng.standard_normal(rig.extr_true.shape),
                 rig.poses_true.copy(), rig.points + rng.normal(0, 2e-4, rig.points.shape)]
        unfixed = [None, None, np.zeros_like(rig.poses_true, dtype=bool), None]
    else:
        def chain():
            return ub["division_projection"]() + fb.extrinsic3D() + fb.template_points()

        div = np.concatenate([rig.intr_true[:, :4], rng.normal(0, 0.05, (rig.n_cams, 1))], axis=1)      # fx, cx, fy, cy, k
        truth = [div, rig.extr_true, rig.poses_true]
        start = [div * (1 + 1e-3 * rng.standard_normal(div.shape)), rig.extr_true + 1e-3 * rng.standard_normal(rig.extr_true.shape),
                 rig.poses_true + 1e-3 * rng.standard_normal(rig.poses_true.shape)]
        unfixed = [None, fix_ext, None]
    if which != "divfree":
        start[1][0] = rig.extr_true[0]
    op = chain()
    tm = (rig.points,) if op.templated else ()
    uv = op.make_full_loss_fn(det, 1)(op.build_param_list(*truth), *tm) + det[:, 3:]
    det[:, 3:] = uv + rng.normal(0, 0.3, uv.shape)
    op.engine.close()
    op = chain()
    prob = handlers.ChainProblem(op, det, start, template=rig.points if op.templated else None, unfixed=unfixed)
    return rig, op, prob


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--trace", action="store_true", help="a warm-up and ONE exact solve per config only (for rocprofv3 --kernel-trace --stats)")
    ap.add_argument("--max-iter", type=int, default=40)
    ap.add_argument("--phases", action="store_true", help="time the build with parts of the contraction switched off")
    ap.add_argument("--no-cg", action="store_true")
    ap.add_argument("--chain", choices=("flex", "division", "divfree"), default="flex")
    ap.add_argument("--two-launch", action="store_true", help="slab preparation as a launch of its own in front of the evaluation")
    ap.add_argument("--deterministic", action="store_true", help="the ordered contraction and step (the same bits on every run)")
    ap.add_argument("--dense", action="store_true", help="the dense form of the normal equations also where the chain has the blocked one")
    a = ap.parse_args()
    for number in a.config:
        rig, op, prob = problem(number, a.chain)
        eng = op._engine_for(prob._flat_detections())
        if a.dense:
            eng.set_option("dense_normal", 1)
        if a.two_launch:
            eng.set_one_launch(False)
        if a.deterministic:
            eng.set_option("deterministic", 1)
        n_free = prob.x0.shape[0]
        lay = eng.normal_layout()
        print(f"config {number}, {eng.chain}: {rig.n_cams} cameras, {rig.n_imgs} images, N = {rig.n_det}, row length {eng.P}, {eng.n_params} parameters "
              f"({n_free} free), normal equations: {lay['n_lead']} leading + {lay['n_trail']} trailing", flush=True)
        lm_solve(prob, prob.x0.copy(), max_iter=a.max_iter)     # warm-up: compiled chain, solver state, page-locked read-back
        torch.cuda.synchronize()
        best, res = np.inf, None
        for _ in range(1 if a.trace else 5):
            t0 = time.perf_counter()
            res = lm_solve(prob, prob.x0.copy(), max_iter=a.max_iter)
            best = min(best, time.perf_counter() - t0)
        print(f"  exact (dense J'J + Cholesky, device-steered): {best * 1e3:8.2f} ms  {res.nfev:3d} evaluations  cost {res.cost:.6e}  {res.message}", flush=True)
        if a.trace:
            continue
        # the dense build alone: evaluation (block rows) + zeroing + contraction, HIP events around ten of them
        lay = eng.normal_layout()
        dev = torch.device("cuda", eng.device)
        ps_dev = torch.from_numpy(np.ascontiguousarray(op.build_param_list(*prob.get_bundle_adjustment_inputs(prob.x0)))[: eng.n_params]).to(dev)
        packed = torch.empty(lay["packed_len"], dtype=torch.float64, device=dev)
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            for _ in range(3):
                eng.normal_blocks_device(ps_dev.data_ptr(), packed.data_ptr(), st.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                eng.normal_blocks_device(ps_dev.data_ptr(), packed.data_ptr(), st.cuda_stream)
            e1.record()
            e1.synchronize()
        print(f"  build (evaluation + zeroing + contraction):    {e0.elapsed_time(e1) * 100:8.1f} us", flush=True)
        if a.phases:   # where the contraction's time goes: the same build without the flush / without the matrix products (results are garbage)
            for dbg, what in ((1, "no flush"), (2, "no contraction"), (3, "loads only")):
                eng.set_option("gram_debug", dbg)
                with torch.cuda.stream(st):
                    e0.record()
                    for _ in range(10):
                        eng.normal_blocks_device(ps_dev.data_ptr(), packed.data_ptr(), st.cuda_stream)
                    e1.record()
                    e1.synchronize()
                print(f"    {what:16s} {e0.elapsed_time(e1) * 100:8.1f} us", flush=True)
            eng.set_option("gram_debug", 0)
        if a.no_cg:
            op.engine.close()
            continue
        t0 = time.perf_counter()
        cg = lm_solve(prob, prob.x0.copy(), max_iter=a.max_iter, linear_solver="pcg")
        t_cg = time.perf_counter() - t0
        print(f"  CG on products (round 4's default):           {t_cg * 1e3:8.2f} ms  {cg.nfev:3d} evaluations, {cg.n_jtjv} products  cost {cg.cost:.6e}", flush=True)
        if rig.n_det <= 200000:
            from scipy.optimize import least_squares
            loss_fn, jac_fn = prob.make_loss_fun(), prob.make_loss_jac()
            t0 = time.perf_counter()
            ref = least_squares(loss_fn, prob.x0.copy(), jac=jac_fn, x_scale="jac", max_nfev=a.max_iter)
            print(f"  scipy trf on the HIP closures:                {(time.perf_counter() - t0) * 1e3:8.2f} ms  {ref.nfev:3d} evaluations  cost {ref.cost:.6e}", flush=True)
        op.engine.close()


if __name__ == "__main__":
    main()
