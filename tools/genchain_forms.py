"""Step time of a generated chain (projection + extrinsic3D + template_points) in its two launch forms, 50 steps back to back:
one launch (every wave prepares its tile's slabs) against slab preparation + evaluation.  usage: python tools/genchain_forms.py"""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from pycamset_amd import function_blocks as fb, synthetic
from pycamset_amd.chain_compiler import ChainEngine
from oracle import ba_oracle as orc
for cfg, take in ((1, None), (2, None), (3, 125000), (3, 250000), (3, 500000), (3, None)):
    rig = synthetic.config_rig(cfg)
    det = rig.detections if take is None else rig.detections[:take]
    ps = orc.build_param_list(rig.intr, rig.extr, rig.poses)
    eng = ChainEngine([fb.projection(), fb.extrinsic3D(), fb.template_points()], rig.n_cams, rig.n_imgs, rig.n_keys)
    eng.set_detections_table(det); eng.set_template(rig.points)
    N = len(det)
    d_p = torch.from_numpy(ps).cuda()
    r = torch.empty((N, 2), dtype=torch.float64, device="cuda"); j = torch.empty((2 * N, 21), dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    for rep in range(3):
        for one in (True, False):
            eng.set_one_launch(one)
            for _ in range(5): eng.eval_device(d_p.data_ptr(), r.data_ptr(), j.data_ptr(), s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): eng.eval_device(d_p.data_ptr(), r.data_ptr(), j.data_ptr(), s)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(one, []).append(e0.elapsed_time(e1) / 50 * 1e3)
    print(f"N = {N:8d}: step (50 back to back, us) one launch {min(res[True]):7.1f}   two launches {min(res[False]):7.1f}")
    eng.close()
