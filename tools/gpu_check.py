#!/usr/bin/env python3
"""Developer check on a GPU box: every chain / dtype / kernel variant against the CPU oracle, then a
quick timing sweep.  (Not part of the product; the oracle is used here only as the checker.)"""
import sys, time, json
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pycamset_amd import synthetic
from pycamset_amd.engine import Engine
from oracle import ba_oracle as orc


def slabs(rig, chain):
    if chain == "template":
        return [rig.intr, rig.extr, rig.poses]
    if chain == "self":
        return [rig.intr, rig.extr, rig.poses, rig.points]
    return [rig.intr, rig.extr, rig.points]


def relerr(a, b):
    rows = np.max(np.abs(b), axis=1, keepdims=True)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-6 * rows))


def main():
    rig = synthetic.config_rig(1)
    print("rig", rig.name, rig.n_det)
    worst = 0
    for chain in ("template", "self", "free"):
        ps = orc.build_param_list(*slabs(rig, chain))
        tm = rig.points if chain == "template" else None
        ref_r = orc.full_loss(chain, rig.detections, ps, tm)
        ref_j = orc.full_jac_dense(chain, rig.detections, ps, tm)
        for dtype in ("f64", "f32"):
            e = Engine(chain, rig.n_cams, rig.n_imgs, rig.n_keys, dtype=dtype)
            e.set_detections_table(rig.detections)
            if tm is not None:
                e.set_template(tm)
            for variant in range(8):
                e.set_option("variant", variant)
                r, j = e.eval(ps)
                er = np.max(np.abs(r - ref_r))
                ej = relerr(j, ref_j)
                print(f"{chain:9s} {dtype} variant {variant}: resid max abs err {er:.3e}  jac max rel err {ej:.3e}")
                if dtype == "f64":
                    worst = max(worst, ej)
            r_only, _ = e.eval(ps, want_jac=False)
            _, j_only = e.eval(ps, want_resid=False)
            print("   resid-only / jac-only agree:", np.array_equal(r_only, r), np.array_equal(j_only, j))
            unfixed = np.random.default_rng(0).random(ps.shape[0]) > 0.3
            nnz = e.set_unfixed(unfixed)
            _, data = e.eval_compact(ps)
            d_ref, idx, ptr = orc.jac_csr(chain, rig.detections, ps, tm, unfixed=unfixed)
            gi, gp = e.csr_structure(unfixed)
            print("   compact nnz", nnz, "structure equal", np.array_equal(gi, idx), np.array_equal(gp, ptr),
                  "data max abs err", np.max(np.abs(data - d_ref) / (1e-30 + np.max(np.abs(d_ref)))))
            e.close()
    print("worst f64 jac rel err", worst)

    # timing sweep, headline config
    rig = synthetic.config_rig(3)
    print("rig", rig.name, rig.n_det)
    import torch
    chain = "template"
    ps = orc.build_param_list(*slabs(rig, chain))
    e = Engine(chain, rig.n_cams, rig.n_imgs, rig.n_keys)
    e.set_detections_table(rig.detections)
    e.set_template(rig.points)
    N = rig.n_det
    d_r = torch.empty(N * 2, dtype=torch.float64, device="cuda")
    d_j = torch.empty(N * 42, dtype=torch.float64, device="cuda")
    d_p = torch.from_numpy(ps).cuda()
    res = {}
    for variant in range(8):
        for wpc in (1, 2, 4, 8):
            e.set_option("variant", variant)
            e.set_option("wgs_per_cu", wpc)
            for _ in range(3):
                e.eval_device_resident(d_p.data_ptr(), d_r.data_ptr(), d_j.data_ptr())
            e.synchronize()
            ts = []
            for _ in range(10):
                e.eval_device_resident(d_p.data_ptr(), d_r.data_ptr(), d_j.data_ptr())
                e.synchronize()
                ts.append(e.last_kernel_ms())
            k0 = np.median([t[0] for t in ts]); k1 = np.median([t[1] for t in ts])
            gbs = N * 380 / (k1 * 1e-3) / 1e9
            print(f"variant {variant} wgs/cu {wpc}: slab_prep {k0*1e3:7.1f} us  eval {k1*1e3:8.1f} us  {gbs:7.1f} GB/s  ({gbs/8000*100:.1f}% of 8 TB/s)")
            res[f"v{variant}_w{wpc}"] = (k0, k1, gbs)
    j = d_j.cpu().numpy().reshape(-1, 21)
    ref = orc.full_jac_dense(chain, rig.detections[:20000], ps, rig.points, threads=8, fast=True)
    print("headline jac check (first 20000 dets) rel err", relerr(j[:40000], ref))
    Path("gpurun_out").mkdir(exist_ok=True)
    json.dump(res, open("gpurun_out/sweep.json", "w"))


if __name__ == "__main__":
    main()
