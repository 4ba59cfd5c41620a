#!/usr/bin/env python3
"""Dense vs blocked normal-equations build on the bench rigs (developer tool, one MI355X): kernel time by HIP events, whole
call incl. zeroing + synchronisation, bytes zeroed."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from pycamset_amd import synthetic
from pycamset_amd.engine import Engine

for cfg, chain in ((3, "template"), (4, "self"), (4, "free")):
    rig = synthetic.config_rig(cfg)
    sl = {"template": [rig.intr, rig.extr, rig.poses], "self": [rig.intr, rig.extr, rig.poses, rig.points], "free": [rig.intr, rig.extr, rig.points]}[chain]
    ps = np.concatenate([a.ravel() for a in sl])
    n = ps.shape[0]
    e = Engine(chain, rig.n_cams, rig.n_imgs, rig.n_keys)
    e.set_detections_table(rig.detections)
    if chain == "template":
        e.set_template(rig.points)
    lay = e.normal_layout()
    Hd = torch.empty(n * n + n + 1, dtype=torch.float64, device="cuda")
    pk = torch.empty(lay["packed_len"], dtype=torch.float64, device="cuda")
    d_ps = torch.from_numpy(ps).cuda()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        def dense():
            e.normal_equations_device(ps, Hd.data_ptr(), Hd.data_ptr() + 8 * n * n, Hd.data_ptr() + 8 * (n * n + n), s.cuda_stream)

        def blocked():
            e.normal_blocks_device(d_ps.data_ptr(), pk.data_ptr(), s.cuda_stream)

        print(f"# {rig.name} chain {chain}: N = {rig.n_det}, n_params {n}; dense H {n * n * 8 / 1e6:.1f} MB, blocked [A|B|C] {(lay['packed_len'] - n - 1) * 8 / 1e6:.1f} MB "
              f"(lead {lay['n_lead']}, trail {lay['n_trail']}, tb {lay['tb']})")
        for name, fn in (("dense", dense), ("blocked", blocked)):
            for _ in range(3):
                fn()
            s.synchronize()
            ks, t0 = [], time.perf_counter()
            for _ in range(10):
                fn()
                s.synchronize()
                ks.append(e.last_kernel_ms()[1])
            host = (time.perf_counter() - t0) / 10
            print(f"  {name:8s} passes {np.median(ks) * 1e3:8.1f} us   call incl. prologue (slabs + zeroing) + sync {host * 1e6:8.1f} us")
    e.close()
