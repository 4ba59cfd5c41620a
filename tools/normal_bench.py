#!/usr/bin/env python3
"""Time the block-reduced normal-equations kernel (H = J^T J, g, cost) and the dense solve that follows it."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pycamset_amd import synthetic
from pycamset_amd.engine import Engine

configs = [(3, "template", False), (3, "template", True), (4, "self", False), (4, "free", False)]
if "--only-default" in sys.argv:
    configs = [c for c in configs if not c[2]]
if any(not a.startswith("--") for a in sys.argv[1:]):
    configs = [c for c in configs if c[1] in sys.argv[1:]]
for cfg, chain, shuffle in configs:
    rig = synthetic.config_rig(cfg)
    sl = {"template": [rig.intr, rig.extr, rig.poses], "self": [rig.intr, rig.extr, rig.poses, rig.points], "free": [rig.intr, rig.extr, rig.points]}[chain]
    ps = np.concatenate([a.ravel() for a in sl])
    n = ps.shape[0]
    det = rig.detections
    if shuffle:
        det = det[np.random.default_rng(0).permutation(det.shape[0])]
    e = Engine(chain, rig.n_cams, rig.n_imgs, rig.n_keys)
    e.set_detections_table(det)
    if chain == "template":
        e.set_template(rig.points)
    Hd = torch.empty((n, n), dtype=torch.float64, device="cuda")
    gd = torch.empty(n, dtype=torch.float64, device="cuda")
    cd = torch.empty(1, dtype=torch.float64, device="cuda")
    print(f"# {rig.name} chain {chain} N={det.shape[0]} n_params={n} shuffled={shuffle}  H = {n*n*8/1e6:.1f} MB")
    if "--phases" in sys.argv:   # option normal_debug: 2 = skip the flush atomics (results are wrong)
        for dbg in (2, 0):
            e.set_option("normal_debug", dbg)
            ks = []
            for _ in range(8):
                e.normal_equations_device(ps, Hd.data_ptr(), gd.data_ptr(), cd.data_ptr()); e.synchronize(); ks.append(e.last_kernel_ms()[1])
            print(f"normal_debug {dbg}: kernels {np.median(ks[2:])*1e3:8.1f} us")
        e.set_option("normal_debug", 0)
    for rows in (64, 32):
        e.set_option("normal_rows", rows)
        ks = []
        for _ in range(8):
            e.normal_equations_device(ps, Hd.data_ptr(), gd.data_ptr(), cd.data_ptr()); e.synchronize(); ks.append(e.last_kernel_ms()[1])
        print(f"normal_rows {rows}: kernels {np.median(ks[2:])*1e3:8.1f} us")
    e.set_option("normal_rows", 64)
    for wpc in ((0,) if "--only-default" in sys.argv else (0, 7, 14, 21, 28, 56)):
        e.set_option("wgs_per_cu", wpc)
        for _ in range(2):
            e.normal_equations_device(ps, Hd.data_ptr(), gd.data_ptr(), cd.data_ptr())
        e.synchronize()
        ks, t0 = [], time.perf_counter()
        for _ in range(10):
            e.normal_equations_device(ps, Hd.data_ptr(), gd.data_ptr(), cd.data_ptr())
            e.synchronize()
            ks.append(e.last_kernel_ms()[1])
        host = (time.perf_counter() - t0) / 10
        print(f"normal wgs/cu {wpc:2d}: kernel {np.median(ks)*1e3:8.1f} us   call incl. memsets + sync {host*1e6:8.1f} us")
    if chain == "template" and not shuffle and "--only-default" not in sys.argv:
        # what follows in an LM step: symmetrise, damp, Cholesky, solve (rocSOLVER through torch)
        free = torch.ones(n, dtype=torch.bool, device="cuda"); free[15 * rig.n_cams: 15 * rig.n_cams + 6] = False; free[9 * rig.n_cams: 9 * rig.n_cams + 6] = False
        idx = torch.nonzero(free).ravel()
        torch.cuda.synchronize()
        for rep in range(3):
            t0 = time.perf_counter()
            Hs = torch.triu(Hd) + torch.triu(Hd, 1).T
            Hf = Hs[idx][:, idx]
            d = torch.diagonal(Hf).clone()
            Hf = Hf + torch.diag(1e-3 * d)
            L, info = torch.linalg.cholesky_ex(Hf)
            x = torch.cholesky_solve(-gd[idx].unsqueeze(1), L)
            torch.cuda.synchronize()
            print(f"torch symmetrise + damp + cholesky + solve ({idx.numel()} free): {(time.perf_counter()-t0)*1e3:.2f} ms  info={int(info)}")
        Hh = Hf.cpu().numpy(); gh = gd[idx].cpu().numpy()
        import scipy.linalg as sla
        for rep in range(2):
            t0 = time.perf_counter()
            cf = sla.cho_factor(Hh, lower=True, check_finite=False)
            xh = sla.cho_solve(cf, -gh, check_finite=False)
            print(f"host scipy cho_factor + cho_solve: {(time.perf_counter()-t0)*1e3:.2f} ms   |x_gpu - x_host| / |x| = {np.linalg.norm(x.cpu().numpy().ravel()-xh)/np.linalg.norm(xh):.2e}")
    e.close()
