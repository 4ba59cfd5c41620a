#!/usr/bin/env python3
"""S -= V V' (csrc/ba_schur.hpp) on its own: time per call for the leading / trailing sizes of the reference rigs.
usage: python tools/syrk_bench.py [n_lead n_trail] [--reps R]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pycamset_amd.engine import schur_syrk

args = [a for a in sys.argv[1:] if not a.startswith("--")]
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 30
sizes = [(int(args[0]), int(args[1]))] if len(args) >= 2 else [(480, 1200), (1680, 1458)]
for n_lead, n_trail in sizes:
    g = torch.Generator(device="cuda").manual_seed(1)
    V = torch.randn((n_lead, n_trail), dtype=torch.float64, device="cuda", generator=g)
    S = torch.zeros((n_lead, n_lead), dtype=torch.float64, device="cuda")
    u = torch.randn(n_trail, dtype=torch.float64, device="cuda", generator=g)
    rhs = torch.zeros(n_lead, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        schur_syrk(0, n_lead, n_trail, V.data_ptr(), n_trail, S.data_ptr(), n_lead, u.data_ptr(), rhs.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        schur_syrk(0, n_lead, n_trail, V.data_ptr(), n_trail, S.data_ptr(), n_lead, u.data_ptr(), rhs.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    flop = n_lead * (n_lead + 1) * n_trail   # lower triangle: n (n + 1) / 2 entries x 2 K
    print(f"syrk n_lead {n_lead} n_trail {n_trail}: {us:.1f} us per call, {flop / us * 1e-6:.1f} TFLOP/s of the lower triangle (FP64 matrix peak 78.6)")
