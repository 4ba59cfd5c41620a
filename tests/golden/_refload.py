"""Load the reference's own Python modules for the hot path (build container only).

TEST INFRASTRUCTURE.  /root/reference is read-only and the reference code-generates
``template_functions/*.py`` beside its own ``__file__`` (abstract_function_blocks.py:298-299, :394;
matmul_map.py:250-251), so the package is copied into a *temporary* directory, the stand-in modules of
``_refstubs.py`` are installed (numba -> identity decorators: the unmodified bodies run as IEEE-754
CPython) and the modules are imported from there.  The copy — and whatever the reference generates in
it — is removed when the context exits; no reference source ever lands in this repo, and nothing on the
GPU box uses this file (there is no /root/reference there: ``available()`` is False).
"""
from __future__ import annotations

import contextlib
import os
import shutil
import sys
import tempfile
from pathlib import Path
from types import SimpleNamespace

HERE = Path(__file__).resolve().parent
REFERENCE = Path(os.environ.get("PCS_REFERENCE", "/root/reference"))


def available() -> bool:
    return (REFERENCE / "pyCamSet" / "optimisation" / "template_handler.py").is_file()


@contextlib.contextmanager
def reference_modules():
    """-> namespace(ch, fb, afb, th, sbh, fph, TargetDetection) of the reference, imported from a temp copy."""
    if not available():
        raise RuntimeError(f"reference not found at {REFERENCE}")
    os.environ.setdefault("MPLBACKEND", "Agg")
    if str(HERE) not in sys.path:
        sys.path.insert(0, str(HERE))
    import _refstubs

    tmp = Path(tempfile.mkdtemp(prefix="pcs_ref_"))
    added = str(tmp)
    before = set(sys.modules)
    try:
        shutil.copytree(REFERENCE / "pyCamSet", tmp / "pyCamSet")
        for p in (tmp / "pyCamSet").rglob("*"):
            os.chmod(p, 0o755 if p.is_dir() else 0o644)
        os.chmod(tmp / "pyCamSet", 0o755)
        _refstubs.install()
        sys.path.insert(0, added)
        import pyCamSet.optimisation.abstract_function_blocks as afb
        import pyCamSet.optimisation.compiled_helpers as ch
        import pyCamSet.optimisation.free_point_handler as fph
        import pyCamSet.optimisation.function_block_implementations as fb
        import pyCamSet.optimisation.standard_bundle_handler as sbh
        import pyCamSet.optimisation.template_handler as th
        from pyCamSet.calibration_targets import TargetDetection

        yield SimpleNamespace(ch=ch, fb=fb, afb=afb, th=th, sbh=sbh, fph=fph, TargetDetection=TargetDetection, root=tmp)
    finally:
        if added in sys.path:
            sys.path.remove(added)
        for name in set(sys.modules) - before:  # forget the temp copy (and the files it generated)
            if name == "pyCamSet" or name.startswith("pyCamSet."):
                del sys.modules[name]
        shutil.rmtree(tmp, ignore_errors=True)
