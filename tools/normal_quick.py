#!/usr/bin/env python3
"""Quick timing of the normal-equations build on rig-32 (developer tool): rows x wgs_per_cu grid for one chain."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pycamset_amd import synthetic
from pycamset_amd.engine import Engine
CONFIG = None
for arg in [x for x in sys.argv if x.startswith("--config=")]: sys.argv.remove(arg); CONFIG = int(arg[9:])
chain = sys.argv[1] if len(sys.argv) > 1 else "template"
rig = synthetic.config_rig(CONFIG if CONFIG else (3 if chain == "template" else 4))
sl = {"template": [rig.intr, rig.extr, rig.poses], "self": [rig.intr, rig.extr, rig.poses, rig.points], "free": [rig.intr, rig.extr, rig.points]}[chain]
ps = np.concatenate([a.ravel() for a in sl]); n = ps.shape[0]
e = Engine(chain, rig.n_cams, rig.n_imgs, rig.n_keys); e.set_detections_table(rig.detections)
if chain == "template": e.set_template(rig.points)
Hd = torch.empty((n, n), dtype=torch.float64, device="cuda"); gd = torch.empty(n, dtype=torch.float64, device="cuda"); cd = torch.empty(1, dtype=torch.float64, device="cuda")
for arg in [x for x in sys.argv if x.startswith("--ikw=")]: sys.argv.remove(arg); e.set_option("normal_imgkey_wgs_per_cu", int(arg[6:]))
for arg in [x for x in sys.argv if x.startswith("--sort=")]: sys.argv.remove(arg); e.set_option("normal_sort_tables", int(arg[7:]))
if "--walk" in sys.argv: sys.argv.remove("--walk"); e.set_option("normal_imgkey_product", 0)
import os
if os.environ.get("PCS_NORMAL_DET", "0") == "1": e.set_option("deterministic", 1)   # the ordered build (csrc/ba_reduce.hpp)
dbgs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
for dbg in dbgs:
  e.set_option("normal_debug", dbg)
  print("normal_debug", dbg)
  for rows in ([int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else (64,)):
    for wpc in [int(x) for x in (sys.argv[3].split(',') if len(sys.argv) > 3 else ['0'])]:
        e.set_option("normal_rows", rows); e.set_option("wgs_per_cu", wpc)
        ks = []
        for _ in range(9):
            e.normal_equations_device(ps, Hd.data_ptr(), gd.data_ptr(), cd.data_ptr()); e.synchronize(); ks.append(e.last_kernel_ms()[1])
        print(f"{chain} rows {rows} wgs/cu {wpc:2d}: median {np.median(ks[2:])*1e3:7.1f} us  min {np.min(ks[2:])*1e3:7.1f} us")
