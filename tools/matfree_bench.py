#!/usr/bin/env python3
"""Time the matrix-free Jacobian products on the headline rig (kernel time by HIP events + host call time)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pycamset_amd import synthetic
from pycamset_amd.engine import Engine

ONLY_OPS = sys.argv[1].split(",") if len(sys.argv) > 1 else None      # e.g. "jv,jtjv": only these products (for counter passes)
WPCS = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (0, 1, 2, 3, 4, 8)
CONFIGS = ((3, "template"),) if len(sys.argv) > 3 and sys.argv[3] == "template" else ((3, "template"), (4, "self"))
for cfg, chain in CONFIGS:
    rig = synthetic.config_rig(cfg)
    sl = [rig.intr, rig.extr, rig.poses] + ([rig.points] if chain == "self" else [])
    ps = np.concatenate([a.ravel() for a in sl])
    e = Engine(chain, rig.n_cams, rig.n_imgs, rig.n_keys)
    e.set_detections_table(rig.detections)
    if chain == "template":
        e.set_template(rig.points)
    e.linearize(ps)
    v = np.random.default_rng(0).standard_normal(ps.shape[0]); u = np.random.default_rng(1).standard_normal(2 * rig.n_det)
    print(f"# {rig.name} chain {chain} N={rig.n_det} n_params={ps.shape[0]}")
    for name, fn in (("jv", lambda: e.jv(v)), ("jtu", lambda: e.jtu(u)), ("jtjv", lambda: e.jtjv(v)), ("diag", e.jtj_diag), ("grad", e.grad)):
        if ONLY_OPS and name not in ONLY_OPS:
            continue
        for wpc in WPCS:
            e.set_option("wgs_per_cu", wpc)
            fn(); fn()
            ks, t0 = [], time.perf_counter()
            for _ in range(10):
                fn(); ks.append(e.last_kernel_ms()[1])
            host = (time.perf_counter() - t0) / 10
            print(f"{name:5s} wgs/cu {wpc:2d}: kernel {np.median(ks)*1e3:7.1f} us   host call {host*1e6:8.1f} us")
    e.close()
