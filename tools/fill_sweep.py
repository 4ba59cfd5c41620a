#!/usr/bin/env python3
"""Sweep the chunked non-temporal fill probes (pcs_membench kinds 4..8) over workgroups per CU, three rounds each."""
import ctypes, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pycamset_amd import _capi

lib = _capi.lib()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 352
names = {1: "linear", 4: "10.5k", 5: "21k", 6: "1k", 7: "64k"}
bpcs = (2, 3, 4, 6, 8, 10, 12, 14, 16, 20, 24, 32, 48, 64)
res = {k: {b: [] for b in bpcs} for k in names}
for rnd in range(3):
    for kind in names:
        for bpc in bpcs:
            ms = ctypes.c_float()
            _capi.check(lib.pcs_membench(0, kind, mb * 1000 * 1000, 20, bpc, ctypes.byref(ms)))
            res[kind][bpc].append(ms.value * 1e3)
print(f"# non-temporal fill of {mb} MB, us (median of 3 rounds x 20 launches) per workgroups/CU")
print("chunk   " + " ".join(f"{b:6d}" for b in bpcs))
for kind, nm in names.items():
    print(f"{nm:7s} " + " ".join(f"{np.median(res[kind][b]):6.1f}" for b in bpcs))
