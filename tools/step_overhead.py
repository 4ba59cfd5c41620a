#!/usr/bin/env python3
"""Where does the step time go beyond the kernels?  Stream choice x event timing."""
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
d_r = torch.empty(N * 2, dtype=torch.float64, device="cuda"); d_j = torch.empty(N * 42, dtype=torch.float64, device="cuda")
d_p = torch.from_numpy(ps).cuda()
side = torch.cuda.Stream()
for sname, stream in (("torch default", torch.cuda.current_stream().cuda_stream), ("torch side stream", side.cuda_stream), ("engine stream", None)):
    for timing in (1, 10, 0):
        e.set_option("timing_every", timing)
        for _ in range(20):
            e.eval_device_resident(d_p.data_ptr(), d_r.data_ptr(), d_j.data_ptr(), stream)
        torch.cuda.synchronize(); e.synchronize()
        t0 = time.perf_counter()
        K = 300
        for _ in range(K):
            e.eval_device_resident(d_p.data_ptr(), d_r.data_ptr(), d_j.data_ptr(), stream)
        torch.cuda.synchronize(); e.synchronize()
        dt = (time.perf_counter() - t0) / K
        print(f"{sname:18s} timing_every={timing}: {dt*1e6:7.2f} us/step")
# host-side enqueue cost alone
e.set_option("timing_every", 0)
t0 = time.perf_counter()
for _ in range(300):
    e.eval_device_resident(d_p.data_ptr(), None, None, None)
print("host call overhead (no kernels queued):", (time.perf_counter() - t0) / 300 * 1e6, "us")
