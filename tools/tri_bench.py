#!/usr/bin/env python3
"""Batched triangulation (f4) and legacy cost (f3) at the headline rig's size: kernel times."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import ba_oracle as orc            # developer tool: the oracle times the CPU side only
from pycamset_amd import synthetic
from pycamset_amd import compiled_helpers as hc
from pycamset_amd.engine import Engine

rig = synthetic.config_rig(3)
Kc = np.zeros((rig.n_cams, 3, 3)); it = rig.intr_true
Kc[:, 0, 0], Kc[:, 0, 2], Kc[:, 1, 1], Kc[:, 1, 2], Kc[:, 2, 2] = it[:, 0], it[:, 1], it[:, 2], it[:, 3], 1.0
from scipy.spatial.transform import Rotation
P = np.stack([Kc[c] @ np.concatenate([Rotation.from_rotvec(rig.extr_true[c, :3]).as_matrix(), rig.extr_true[c, 3:, None]], axis=1) for c in range(rig.n_cams)])
D = np.ascontiguousarray(it[:, 4:9])
d = rig.detections
d = d[np.lexsort((d[:, 0], d[:, 2], d[:, 1]))]
t0 = time.perf_counter(); rec, start = hc.group_reconstructable(d); tg = time.perf_counter() - t0
n_pts = len(start) - 1
print(f"triangulation: {rec.shape[0]} observations, {n_pts} points, views/point {np.diff(start).min()}..{np.diff(start).max()} (host grouping {tg:.2f} s)")
for _ in range(3):
    t0 = time.perf_counter(); pts = hc.nb_triangulate_full(rec, P, start, Kc, D); wall = time.perf_counter() - t0
    print(f"  HIP kernel {hc.last_triangulate_kernel_ms*1e3:9.1f} us   ({n_pts/(hc.last_triangulate_kernel_ms*1e-3):.3e} points/s, {rec.shape[0]*20/hc.last_triangulate_kernel_ms/1e6:.1f} GB/s of 20 B/obs)   call wall {wall*1e3:.1f} ms")
sel = np.arange(0, n_pts, 200)
rows = np.concatenate([np.arange(start[j], start[j + 1]) for j in sel]); sst = np.append(0, np.cumsum(np.diff(start)[sel]))
t0 = time.perf_counter(); ref = orc.triangulate_full(rec[rows], P, sst, Kc, D); tc = time.perf_counter() - t0
print(f"  NumPy/LAPACK oracle: {tc/len(sel)*1e6:.1f} us per point (1 core) -> {len(sel)/tc:.3e} points/s; max rel diff {np.max(np.linalg.norm(pts[sel]-ref,axis=1)/np.linalg.norm(ref,axis=1)):.2e}")

im, Pj, Kj, Dj = orc.legacy_inputs(rig.intr, rig.extr, rig.poses, rig.points)
eng = Engine("template", rig.n_cams, rig.n_imgs, rig.n_keys); eng.set_detections_table(rig.detections); eng.set_template(rig.points)
for _ in range(3):
    t0 = time.perf_counter(); err = eng.legacy_cost(im, Pj, Kj, Dj); wall = time.perf_counter() - t0
    k = eng.last_kernel_ms()[1]
    print(f"legacy cost: kernel {k*1e3:7.1f} us  {rig.n_det*68/k/1e6:.1f} GB/s (28 B in + 24 B point gather + 16 B out per detection)  call wall {wall*1e3:.1f} ms")
t0 = time.perf_counter(); orc.legacy_cost(rig.detections, im, Pj, Kj, Dj, threads=16, fast=True); print(f"  oracle (16 threads): {(time.perf_counter()-t0)*1e3:.1f} ms")
