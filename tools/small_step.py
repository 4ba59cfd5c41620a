#!/usr/bin/env python3
"""One-launch step vs slab_prep + evaluation, on small tables (developer tool, one MI355X).

For each shard size (config rig cut K ways like bench.py --emulate-world) and each (fuse_prep, variant) pair:
wall-clock step time of back-to-back untimed steps, HIP-event kernel times of extra launches, and a bit-for-bit
comparison of the one-launch outputs with the two-launch ones (same slab element functions -> same bits)."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from bench import BYTES_PER_DET, CONFIG_CHAIN, CONFIG_DTYPE, rank_problem
from pycamset_amd.engine import Engine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--chain", default=None)
    ap.add_argument("--dtype", default=None)
    ap.add_argument("--worlds", default="8,4,2", help="emulated world sizes (rank 0's shard of a K-way split); 1 = the whole rig")
    ap.add_argument("--variants", default="6,2")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--mode", default="both", choices=["both", "resid"])
    ap.add_argument("--tag", default="")
    ap.add_argument("--lazy", default="1", help="comma list of 0/1: lazy done-event recording (option lazy_done_event)")
    a = ap.parse_args()
    chain = a.chain or CONFIG_CHAIN[a.config]
    dtype = a.dtype or CONFIG_DTYPE[a.config]
    dev = torch.device("cuda", 0)
    out = {}
    for world in [int(w) for w in a.worlds.split(",")]:
        prob = rank_problem(a.config, chain, 0, world, "strong")
        rig, det, ps = prob["rig"], prob["det"], prob["param_str"]
        N = det.shape[0]
        eng = Engine(chain, rig.n_cams, rig.n_imgs, rig.n_keys, dtype=dtype)
        eng.set_detections_table(det)
        if chain == "template":
            eng.set_template(rig.points)
        tdt = torch.float64 if dtype == "f64" else torch.float32
        d_r = torch.empty((N, 2), dtype=tdt, device=dev)
        d_j = torch.empty((2 * N, eng.P), dtype=tdt, device=dev)
        d_p = torch.from_numpy(ps).to(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        pj = d_j.data_ptr() if a.mode == "both" else None
        combos = [(f, int(v), int(z)) for z in a.lazy.split(",") for v in a.variants.split(",") for f in (0, 1)]
        wall = {c: [] for c in combos}
        kern = {c: [] for c in combos}
        prep = {c: [] for c in combos}
        ref = {}
        for rnd in range(a.rounds + 1):
            for c in combos:
                eng.set_option("fuse_prep", c[0])
                eng.set_option("variant", c[1])
                eng.set_option("lazy_done_event", c[2])
                eng.set_option("timing_every", 0)
                for _ in range(20):
                    eng.eval_device_resident(d_p.data_ptr(), d_r.data_ptr(), pj, stream)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    eng.eval_device_resident(d_p.data_ptr(), d_r.data_ptr(), pj, stream)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / a.steps
                eng.set_option("timing_every", 1)
                eng.set_option("event_ring", 16)
                for _ in range(16):
                    eng.eval_device_resident(d_p.data_ptr(), d_r.data_ptr(), pj, stream)
                torch.cuda.synchronize()
                p_s, e_s = eng.kernel_ms_samples(16)
                if rnd:
                    wall[c].append(dt)
                    kern[c].append(float(np.median(e_s)))
                    prep[c].append(float(np.median(p_s)))
                else:
                    ref[c] = (d_r.cpu().numpy().copy(), d_j.cpu().numpy().copy() if pj else None)
        bpd = BYTES_PER_DET[(chain, dtype)] if a.mode == "both" else 44
        print(f"# config {a.config} chain {chain} {dtype} world {world}: N = {N} ({N * bpd / 1e6:.1f} MB algorithmic), mode {a.mode}")
        for c in combos:
            w, k, p = np.median(wall[c]) * 1e6, np.median(kern[c]) * 1e3, np.median(prep[c]) * 1e3
            same = ""
            if c[0] == 1:
                r0, j0 = ref[(0, c[1], c[2])]
                r1, j1 = ref[c]
                same = f"  bit-equal to two-launch: resid {bool(np.array_equal(r0, r1))}" + (f" jac {bool(np.array_equal(j0, j1))}" if j0 is not None else "")
            print(f"  fuse_prep {c[0]} variant {c[1]} lazy_done {c[2]}: step {w:7.2f} us (min {np.min(wall[c]) * 1e6:7.2f})  kernel {k:6.2f} us  slab_prep {p:5.2f} us  "
                  f"step rate {N * bpd / (w * 1e-6) / 1e9:7.1f} GB/s = {N * bpd / (w * 1e-6) / 8e12:.3f} of 8 TB/s{same}")
            out[f"w{world}_f{c[0]}_v{c[1]}_z{c[2]}"] = {"N": N, "step_us": w, "step_us_min": float(np.min(wall[c]) * 1e6), "kernel_us": k, "slab_prep_us": p}
        eng.close()
    Path("gpurun_out").mkdir(exist_ok=True)
    json.dump(out, open(f"gpurun_out/small_step_c{a.config}_{chain}_{dtype}_{a.mode}{a.tag}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
