#!/usr/bin/env python3
"""Measure the box's achievable HBM rates with the engine's access shape (pcs_membench)."""
import ctypes, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pycamset_amd import _capi

lib = _capi.lib()
res = {}
kinds = ((0, "fill"), (1, "fill_nt"), (2, "copy"), (3, "copy_nt"), (4, "fill_nt_chunks"), (5, "fill_nt_21k"), (6, "fill_nt_1k"),
         (7, "fill_nt_64k"), (8, "fill_nt_256k"))
if len(sys.argv) > 1:
    kinds = [k for k in kinds if k[1] in sys.argv[1:]]
for kind, name in kinds:
    for mb in (352, 1024, 4096):
        for bpc in ((4, 8, 16) if kind < 4 else (4, 12, 16, 64)):
            ms = ctypes.c_float()
            _capi.check(lib.pcs_membench(0, kind, mb * 1000 * 1000, 20, bpc, ctypes.byref(ms)))
            moved = mb * 1e6 * (2 if kind in (2, 3) else 1)
            gbs = moved / (ms.value * 1e-3) / 1e9
            res[f"{name}_{mb}MB_bpc{bpc}"] = gbs
            print(f"{name:14s} {mb:5d} MB  blocks/CU {bpc:2d}: {ms.value*1e3:8.1f} us  {gbs:7.1f} GB/s")
Path("gpurun_out").mkdir(exist_ok=True)
json.dump(res, open("gpurun_out/membench.json", "w"), indent=1)
