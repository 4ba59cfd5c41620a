#!/usr/bin/env python3
"""Time the two compaction kernels and the host hand-off (pageable vs pinned) on the headline rig."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pycamset_amd import synthetic
from pycamset_amd.engine import Engine

rig = synthetic.config_rig(3)
ps = np.concatenate([rig.intr.ravel(), rig.extr.ravel(), rig.poses.ravel()])
e = Engine("template", rig.n_cams, rig.n_imgs, rig.n_keys)
e.set_detections_table(rig.detections); e.set_template(rig.points)
N = rig.n_det
mask = np.ones(ps.shape[0], bool)
mask[9 * rig.n_cams: 9 * rig.n_cams + 6] = False      # camera 0 extrinsic fixed
mask[15 * rig.n_cams: 15 * rig.n_cams + 6] = False    # pose 0 fixed
nnz = e.set_unfixed(mask)
d_data = torch.empty(nnz, dtype=torch.float64, device="cuda")
d_r = torch.empty(2 * N, dtype=torch.float64, device="cuda")
print("nnz", nnz, "of dense", 2 * N * 21)
for cv in (0, 1):
    e.set_option("compact_variant", cv)
    for wpc in (0, 4, 16, 32):
        e.set_option("wgs_per_cu", wpc)
        ts = []
        for it in range(12):
            e.eval_compact_device(ps, d_r.data_ptr(), d_data.data_ptr())
            e.synchronize()
            if it >= 2:
                ts.append(e.last_kernel_ms()[1])
        med = float(np.median(ts))
        b = N * (28 + 12 + 16) + nnz * 8
        print(f"compact_variant {cv} wgs/cu {wpc:2d}: {med*1e3:7.1f} us  {b/med/1e6:7.1f} GB/s (algorithmic incl. 12 B/det mask+offset)")
e.set_option("wgs_per_cu", 0)
for ring in (0, 3):
    for name, fn in (("dense eval", lambda: e.eval(ps, want_resid=False, pinned_ring=ring)), ("compact eval", lambda: e.eval_compact(ps, pinned_ring=ring))):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        dt = (time.perf_counter() - t0) / 5
        print(f"{name:13s} pinned_ring={ring}: {dt*1e3:7.2f} ms per call -> {2*N/dt:.3e} rows/s (host arrays)")
