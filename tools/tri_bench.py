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
import ctypes
import torch
from pycamset_amd import _capi
cam32 = np.ascontiguousarray(rec[:, 0].astype(np.int32)); uv = np.ascontiguousarray(rec[:, -2:]); st = np.ascontiguousarray(start, dtype=np.int64)
dp, ip, lp = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
pts = np.empty((n_pts, 3)); ms = ctypes.c_float()
for _ in range(3):   # (a) the stateless C entry point: a temporary handle, 7 allocations + 4 H2D copies + 1 D2H per call
    t0 = time.perf_counter()
    _capi.check(_capi.lib().pcs_triangulate(0, cam32.shape[0], cam32.ctypes.data_as(ip), uv.ctypes.data_as(dp), n_pts, st.ctypes.data_as(lp), rig.n_cams,
                                            P.ctypes.data_as(dp), Kc.ctypes.data_as(dp), D.ctypes.data_as(dp), pts.ctypes.data_as(dp), ctypes.byref(ms)))
    print(f"  stateless pcs_triangulate:            kernel {ms.value*1e3:8.1f} us   call {1e3*(time.perf_counter()-t0):8.2f} ms")
for _ in range(3):   # (b) the mirror of nb_triangulate_full: cached handle, host inputs copied, result copied back
    t0 = time.perf_counter(); pts = hc.nb_triangulate_full(rec, P, start, Kc, D); wall = time.perf_counter() - t0
    print(f"  handle, host in / host out:           kernel {hc.last_triangulate_kernel_ms*1e3:8.1f} us   call {wall*1e3:8.2f} ms   ({n_pts/(hc.last_triangulate_kernel_ms*1e-3):.3e} points/s)")
tri = hc.Triangulator(rig.n_cams); tri.set_cameras(P, Kc, D)
d_cam, d_uv, d_st = torch.from_numpy(cam32).cuda(), torch.from_numpy(uv).cuda(), torch.from_numpy(st).cuda()
d_pts = torch.empty((n_pts, 3), dtype=torch.float64, device="cuda")
tri.set_observations_device(cam32.shape[0], d_cam.data_ptr(), d_uv.data_ptr(), n_pts, d_st.data_ptr())
for _ in range(2):
    tri.run(d_pts.data_ptr()); tri.synchronize()
calls = []
for _ in range(20):  # (c) everything resident: the call is enqueue + synchronize
    t0 = time.perf_counter(); tri.run(d_pts.data_ptr()); tri.synchronize(); calls.append(time.perf_counter() - t0)
k = tri.last_kernel_ms()
print(f"  handle, device in / device out:       kernel {k*1e3:8.1f} us   call {np.median(calls)*1e6:8.1f} us (median of 20) = {np.median(calls)*1e3/k:.2f} x kernel")
assert np.array_equal(d_pts.cpu().numpy(), pts), "resident and host paths agree bit for bit"
for _ in range(3):   # (d) the whole front end: grouping (device: pcs_tri_group_device) + triangulation from the detection table
    t0 = time.perf_counter(); pts_fe = hc.multi_cam_triangulate(d, P, Kc, D); wall = time.perf_counter() - t0
    print(f"  multi_cam_triangulate (table in, points out; grouping on the device): call {wall*1e3:8.2f} ms   (host grouping alone: {tg*1e3:.0f} ms)")
assert np.array_equal(pts_fe, pts), "device grouping and host grouping give the same points"
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
