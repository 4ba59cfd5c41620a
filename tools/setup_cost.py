#!/usr/bin/env python3
"""Where the one-off setup time of a bundle adjustment goes (rig-32, 1e6 detections): table upload, CSR structure,
closures — the parts that run once per problem before the first kernel."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "examples"))
import torch
from pycamset_amd import handlers, synthetic
from pycamset_amd.detections import TargetDetection
from pycamset_amd.engine import Engine
from lm_ring8 import Camset, Target

torch.zeros(1, device="cuda")
t = time.perf_counter
t0 = t(); rig = synthetic.config_rig(3); print(f"synthetic rig                      {t()-t0:7.3f} s  (N = {rig.n_det})")
t0 = t(); e = Engine("template", rig.n_cams, rig.n_imgs, rig.n_keys); print(f"Engine()                           {t()-t0:7.3f} s")
t0 = t(); e.set_detections_table(rig.detections); print(f"set_detections_table               {t()-t0:7.3f} s")
t0 = t(); e.set_template(rig.points); print(f"set_template                       {t()-t0:7.3f} s")
mask = np.ones(e.n_params, bool); mask[15 * rig.n_cams: 15 * rig.n_cams + 6] = False
t0 = t(); ind, ptr = e.csr_structure(mask); print(f"csr_structure (nnz {ind.shape[0]})      {t()-t0:7.3f} s")
t0 = t(); e.set_unfixed(mask); print(f"set_unfixed                        {t()-t0:7.3f} s")
e.close()
cs = Camset(rig.n_cams)
t0 = t(); det = TargetDetection(cs.get_names(), rig.detections); print(f"TargetDetection                    {t()-t0:7.3f} s")
t0 = t()
h = handlers.TemplateBundleHandler(cs, Target(rig.points), det, fixed_params={"cam_0": {"ext": rig.extr_true[0].copy()}}, options={"verbosity": 0})
print(f"TemplateBundleHandler()            {t()-t0:7.3f} s")
t0 = t(); loss = h.make_loss_fun(1); print(f"make_loss_fun                      {t()-t0:7.3f} s")
t0 = t(); jac = h.make_loss_jac(1); print(f"make_loss_jac                      {t()-t0:7.3f} s")
bp = h.bundlePrimitive
x0 = np.concatenate([rig.intr[bp.intr_unfixed].ravel(), rig.extr[bp.extr_unfixed].ravel(), rig.poses[bp.poses_unfixed].ravel()])
t0 = t(); r = loss(x0); print(f"first loss(x)                      {t()-t0:7.3f} s")
t0 = t(); r = loss(x0); print(f"second loss(x)                     {t()-t0:7.3f} s")
t0 = t(); J = jac(x0); print(f"first jac(x)                       {t()-t0:7.3f} s")
t0 = t(); J = jac(x0); print(f"second jac(x)                      {t()-t0:7.3f} s")
