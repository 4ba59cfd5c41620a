"""In-tree build of the HIP extension: ``python -m pycamset_amd.build``.

One translation unit, gfx950 only; hipcc cross-compiles without a GPU.  The result,
``pycamset_amd/libpcs_hip.so``, is git-ignored but travels with gpurun snapshots.
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
SRC = PKG / "csrc" / "pcs_engine.hip"
DEPS = [SRC, PKG / "csrc" / "ba_device.hpp", PKG / "csrc" / "ba_rtc_prelude.hpp", PKG / "csrc" / "ba_kernels.hpp", PKG / "csrc" / "ba_matfree.hpp", PKG / "csrc" / "ba_normal.hpp", PKG / "csrc" / "ba_schur.hpp", PKG / "csrc" / "ba_dense_chol.hpp", PKG / "csrc" / "ba_chol_persist.hpp", PKG / "csrc" / "ba_generic.hpp", PKG / "csrc" / "ba_blockrow.hpp", PKG / "csrc" / "ba_blockgram.hpp", PKG / "csrc" / "pcs_genchain.inc",
        PKG / "csrc" / "ba_triangulate.hpp", PKG.parent / "include" / "pcs_hip.h"]
OUT = PKG / "libpcs_hip.so"


def needs_build() -> bool:
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    return any(d.stat().st_mtime > t for d in DEPS)


def build(force: bool = False, verbose: bool = False, resource_log: str | None = None) -> Path:
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", str(OUT), str(SRC)]
    if resource_log:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if resource_log:
        Path(resource_log).write_text(proc.stderr)
    if proc.returncode != 0:
        sys.stderr.write(proc.stderr)
        raise RuntimeError("hipcc failed building libpcs_hip.so")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(OUT)
