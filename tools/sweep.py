#!/usr/bin/env python3
"""Launch-geometry / variant sweep of the fused kernel on one MI355X (developer tool).
Interleaved rounds in one process (cdna_hip_programming.md rule 24): median and min of the
HIP-event kernel time per (variant, wgs_per_cu)."""
import argparse, json, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pycamset_amd import synthetic
from pycamset_amd.engine import Engine

BPD = {("template", "f64"): 380, ("self", "f64"): 428, ("free", "f64"): 332, ("template", "f32"): 196, ("self", "f32"): 220, ("free", "f32"): 172,
       ("template", "mixed"): 204, ("self", "mixed"): 228, ("free", "mixed"): 180}


def slabs(rig, chain):
    return {"template": [rig.intr, rig.extr, rig.poses], "self": [rig.intr, rig.extr, rig.poses, rig.points],
            "free": [rig.intr, rig.extr, rig.points]}[chain]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--chain", default="template")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--variants", default="6,7")
    ap.add_argument("--wgs", default="1,2,4,8,16")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--n-imgs", type=int, default=None, help="override the number of poses (changes the slab size, N stays ~ the same per pose)")
    ap.add_argument("--scale", type=float, default=1.0, help="visibility scale of the rig")
    ap.add_argument("--shuffle", action="store_true", help="random detection order (worst case for slab locality)")
    ap.add_argument("--mode", default="both", choices=["both", "jac", "resid"])
    ap.add_argument("--tag", default="")
    ap.add_argument("--xcd", default="0", help="comma list of 0/1: XCD-contiguous tile remap")
    ap.add_argument("--waves", default="0", help="comma list of waves per workgroup (0 = automatic, 1, 2, 4)")
    ap.add_argument("--pack", type=int, default=1, help="0: three int32 index arrays instead of the packed 32-bit word")
    a = ap.parse_args()
    rig = synthetic.config_rig(a.config, n_imgs=a.n_imgs, scale=a.scale)
    det = rig.detections
    if a.shuffle:
        det = det[np.random.default_rng(0).permutation(det.shape[0])]
    N = det.shape[0]
    ps = np.concatenate([x.ravel() for x in slabs(rig, a.chain)])
    e = Engine(a.chain, rig.n_cams, rig.n_imgs, rig.n_keys, dtype=a.dtype)
    e.set_option("pack_indices", a.pack)
    e.set_detections_table(det)
    if a.chain == "template":
        e.set_template(rig.points)
    tdt = torch.float64 if a.dtype == "f64" else torch.float32   # device outputs of f32 and mixed engines are float
    d_r = torch.empty(N * 2, dtype=tdt, device="cuda")
    d_j = torch.empty(N * 2 * e.P, dtype=tdt, device="cuda")
    d_p = torch.from_numpy(ps).cuda()
    pr = d_r.data_ptr() if a.mode in ("both", "resid") else None
    pj = d_j.data_ptr() if a.mode in ("both", "jac") else None
    combos = [(int(v), int(w), int(rs), int(x)) for x in a.xcd.split(",") for rs in a.waves.split(",") for v in a.variants.split(",") for w in a.wgs.split(",")]
    times = {c: [] for c in combos}
    for rnd in range(a.rounds + 1):
        for c in combos:
            e.set_option("variant", c[0]); e.set_option("wgs_per_cu", c[1]); e.set_option("waves_per_wg", c[2]); e.set_option("xcd_remap", c[3])
            for _ in range(3):
                e.eval_device_resident(d_p.data_ptr(), pr, pj)
            e.synchronize()
            if rnd:
                times[c].append(e.last_kernel_ms()[1])
    bpd = BPD[(a.chain, a.dtype)] if a.mode == "both" else (44 if a.mode == "resid" else BPD[(a.chain, a.dtype)] - 16)
    out = {}
    slab_kb = (rig.n_cams * 48 + (0 if a.chain == "free" else rig.n_imgs * 40) + 3 * rig.n_keys) * (8 if a.dtype == "f64" else 4) / 1024
    print(f"# config {a.config} chain {a.chain} {a.dtype} pack={a.pack} N={N} imgs={rig.n_imgs} slabs={slab_kb:.0f} KiB shuffle={a.shuffle} mode={a.mode}")
    for c in combos:
        med, mn = float(np.median(times[c])), float(np.min(times[c]))
        gbs = N * bpd / (med * 1e-3) / 1e9
        print(f"xcd {c[3]} waves/wg {c[2]} variant {c[0]} wgs/cu {c[1]:3d}: median {med*1e3:8.1f} us  min {mn*1e3:8.1f} us  {gbs:7.1f} GB/s  {gbs/80:.1f}% of 8 TB/s")
        out[f"x{c[3]}_rs{c[2]}_v{c[0]}_w{c[1]}"] = {"median_us": med * 1e3, "min_us": mn * 1e3, "GBps": gbs}
    Path("gpurun_out").mkdir(exist_ok=True)
    json.dump(out, open(f"gpurun_out/sweep_{a.config}_{a.chain}_{a.dtype}{'_shuf' if a.shuffle else ''}_{a.mode}{a.tag}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
