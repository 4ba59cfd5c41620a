"""ctypes binding of the C ABI in include/pcs_hip.h (libpcs_hip.so, built in-tree by
``__graft_entry__.build()`` / ``python -m pycamset_amd.build``).

There is no fallback: if the shared library is missing or no HIP device is visible the
product path raises (PcsError / OSError) — it never routes through NumPy or the CPU oracle.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_uint64, c_void_p
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libpcs_hip.so"

PCS_OK, PCS_ERR_ARG, PCS_ERR_HIP, PCS_ERR_STATE, PCS_ERR_NODEVICE, PCS_ERR_RANGE = 0, -1, -2, -3, -4, -5
CHAIN_IDS = {"template": 0, "self": 1, "free": 2}
CHAIN_P = {"template": 21, "self": 24, "free": 18}
DTYPE_IDS = {"f64": 0, "f32": 1, "mixed": 2}   # mixed: FP64 arithmetic, FP32 residual / Jacobian bytes

# every symbol include/pcs_hip.h declares: name -> (restype, argtypes)
_P = c_void_p
class LmBuffers(ctypes.Structure):
    """include/pcs_hip.h pcs_lm_buffers: the device buffers of one LM trial (pcs_lm_trial_build / pcs_lm_trial_finish)."""
    _fields_ = [("packed", c_void_p * 2), ("ps", c_void_p * 2)] + [
        (n, c_void_p) for n in ("flags", "fixed", "lam", "linvt", "u", "V", "S", "rhs", "dvec", "gm", "status", "xlead", "w", "spd_work", "delta", "ctrl", "stats",
                                "stats_host")] + [
        ("spd_algorithm", ctypes.c_int32), ("mode", ctypes.c_int32), ("free_idx", c_void_p), ("n_free", ctypes.c_int64), ("result_host", c_void_p),
        ("syrk_work", c_void_p), ("syrk_work_len", ctypes.c_int64)]


LM_FIXED_TRIAL_BUFFER, LM_VOTES = 1, 2   # include/pcs_hip.h PCS_LM_*
LM_STATS = 12                            # doubles in a trial's read-back (and in the control block)


SYMBOLS = {
    "pcs_version": (c_int, []),
    "pcs_last_error": (c_char_p, []),
    "pcs_device_count": (c_int, []),
    "pcs_create": (c_int, [POINTER(_P), c_int, c_int, c_int64, c_int64, c_int64, c_int]),
    "pcs_destroy": (c_int, [_P]),
    "pcs_n_params": (c_int64, [_P]),
    "pcs_row_len": (c_int, [_P]),
    "pcs_n_detections": (c_int64, [_P]),
    "pcs_set_detections_table": (c_int, [_P, POINTER(c_double), c_int64]),
    "pcs_set_detections": (c_int, [_P, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_double), c_int64]),
    "pcs_set_template": (c_int, [_P, POINTER(c_double)]),
    "pcs_eval": (c_int, [_P, POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "pcs_eval_device": (c_int, [_P, POINTER(c_double), _P, _P, _P]),
    "pcs_eval_device_resident": (c_int, [_P, _P, _P, _P, _P]),
    "pcs_csr_structure": (c_int, [_P, POINTER(c_uint8), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "pcs_block_param_inds": (c_int, [_P, POINTER(c_int64)]),
    "pcs_set_unfixed": (c_int, [_P, POINTER(c_uint8), POINTER(c_int64)]),
    "pcs_eval_compact": (c_int, [_P, POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "pcs_eval_compact_device": (c_int, [_P, POINTER(c_double), _P, _P, _P]),
    "pcs_legacy_cost": (c_int, [_P, POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "pcs_linearize": (c_int, [_P, POINTER(c_double)]),
    "pcs_matfree": (c_int, [_P, c_int, POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "pcs_normal_equations": (c_int, [_P, POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "pcs_normal_equations_device": (c_int, [_P, POINTER(c_double), c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcs_normal_layout": (c_int, [_P, POINTER(c_int64)]),
    "pcs_normal_blocks_device": (c_int, [_P, _P, _P, _P]),
    "pcs_schur_prepare": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pcs_schur_finish": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pcs_lm_decide": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pcs_lm_trial": (c_int, [_P, _P, _P]),
    "pcs_lm_trial_build": (c_int, [_P, _P, _P]),
    "pcs_lm_trial_finish": (c_int, [_P, _P, _P]),
    "pcs_schur_syrk": (c_int, [c_int, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, _P, _P]),
    "pcs_schur_vtx": (c_int, [c_int, c_int64, c_int64, _P, c_int64, _P, _P, _P]),
    "pcs_schur_syrk_work_len": (c_int64, [c_int64, c_int64]),
    "pcs_schur_syrk_ordered": (c_int, [c_int, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, _P, _P, c_int64, _P]),
    "pcs_dense_spd_work_len": (c_int64, [c_int64]),
    "pcs_dense_spd_solve": (c_int, [c_int, c_int64, _P, c_int64, _P, _P, _P, _P, _P]),
    "pcs_dense_spd_solve_algo": (c_int, [c_int, c_int64, _P, c_int64, _P, _P, _P, _P, _P, c_int]),
    "pcs_dense_spd_solve_opts": (c_int, [c_int, c_int64, _P, c_int64, _P, _P, _P, _P, _P, c_int, c_int64]),
    "pcs_normal_descriptors": (c_int, [c_int, c_int, c_int, POINTER(c_int32)]),
    "pcs_genchain_create": (c_int, [POINTER(_P), c_char_p, c_int, c_int, c_int, POINTER(c_int64), POINTER(c_int32), c_int, POINTER(c_int64), c_int64, c_int64,
                                    c_int64, c_int64, c_int64, c_int64, c_int, c_int]),
    "pcs_genchain_destroy": (c_int, [_P]),
    "pcs_genchain_row_len": (c_int, [_P]),
    "pcs_genchain_set_detections_table": (c_int, [_P, POINTER(c_double), c_int64]),
    "pcs_genchain_set_template": (c_int, [_P, POINTER(c_double)]),
    "pcs_genchain_eval": (c_int, [_P, POINTER(c_double), _P, _P]),
    "pcs_genchain_eval_device": (c_int, [_P, _P, _P, _P, _P]),
    "pcs_genchain_set_one_launch": (c_int, [_P, c_int]),
    "pcs_genchain_set_unfixed": (c_int, [_P, POINTER(c_uint64), POINTER(c_int64), c_int64]),
    "pcs_genchain_eval_compact": (c_int, [_P, POINTER(c_double), _P, _P]),
    "pcs_genchain_eval_compact_device": (c_int, [_P, _P, _P, _P, _P]),
    "pcs_genchain_set_blocks": (c_int, [_P, c_int, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int64)]),
    "pcs_genchain_linearize": (c_int, [_P, POINTER(c_double)]),
    "pcs_genchain_matfree": (c_int, [_P, c_int, POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "pcs_genchain_normal_layout": (c_int, [_P, POINTER(c_int64)]),
    "pcs_genchain_normal_blocks_device": (c_int, [_P, _P, _P, _P]),
    "pcs_genchain_lm_trial": (c_int, [_P, _P, _P]),
    "pcs_genchain_lm_trial_build": (c_int, [_P, _P, _P]),
    "pcs_genchain_lm_trial_finish": (c_int, [_P, _P, _P]),
    "pcs_genchain_set_option": (c_int, [_P, c_char_p, c_int64]),
    "pcs_genchain_schur_prepare": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pcs_genchain_schur_finish": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pcs_genchain_lm_decide": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pcs_genchain_device_buffers": (c_int, [_P, POINTER(_P), POINTER(_P)]),
    "pcs_genchain_synchronize": (c_int, [_P, _P]),
    "pcs_genchain_last_kernel_ms": (c_int, [_P, POINTER(c_float), POINTER(c_float)]),
    "pcs_synchronize": (c_int, [_P, _P]),
    "pcs_last_kernel_ms": (c_int, [_P, POINTER(c_float), POINTER(c_float)]),
    "pcs_kernel_ms_mean": (c_int, [_P, POINTER(c_int64), POINTER(c_float), POINTER(c_float)]),
    "pcs_kernel_ms_samples": (c_int, [_P, c_int64, POINTER(c_float), POINTER(c_float), POINTER(c_int64)]),
    "pcs_normal_entry_map": (c_int, [c_int, c_int, POINTER(c_int32)]),
    "pcs_triangulate": (c_int, [c_int, c_int64, POINTER(c_int32), POINTER(c_double), c_int64, POINTER(c_int64), c_int64,
                                POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_float)]),
    "pcs_tri_create": (c_int, [POINTER(_P), c_int, c_int64]),
    "pcs_tri_destroy": (c_int, [_P]),
    "pcs_tri_set_cameras": (c_int, [_P, POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "pcs_tri_set_observations": (c_int, [_P, c_int64, POINTER(c_int32), POINTER(c_double), c_int64, POINTER(c_int64)]),
    "pcs_tri_set_observations_device": (c_int, [_P, c_int64, _P, _P, c_int64, _P]),
    "pcs_tri_group_device": (c_int, [_P, c_int64, _P, _P, _P, c_int64, POINTER(c_int64), POINTER(c_int64), POINTER(c_int32), _P]),
    "pcs_tri_run": (c_int, [_P, _P, _P]),
    "pcs_tri_points": (c_int, [_P, POINTER(c_double)]),
    "pcs_tri_synchronize": (c_int, [_P, _P]),
    "pcs_tri_last_kernel_ms": (c_int, [_P, POINTER(c_float)]),
    "pcs_host_alloc": (c_int, [POINTER(_P), c_int64]),
    "pcs_host_free": (c_int, [_P]),
    "pcs_membench": (c_int, [c_int, c_int, c_int64, c_int, c_int, POINTER(c_float)]),
    "pcs_set_option": (c_int, [_P, c_char_p, c_int64]),
    "pcs_device_buffers": (c_int, [_P, POINTER(_P), POINTER(_P)]),
}


class PcsError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"pcs error {code}: {message}")
        self.code = code


_lib = None


def lib() -> ctypes.CDLL:
    """Load libpcs_hip.so (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise OSError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                f"(python -c 'import __graft_entry__ as g; g.build()' or python -m pycamset_amd.build). "
                f"pycamset_amd has no CPU fallback."
            )
        # PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so under the same
        # SONAMEs as /opt/rocm's.  Two HIP runtimes in one process cannot both own the GPU, so the
        # runtime torch ships must be the one already loaded when libpcs_hip.so resolves its
        # DT_NEEDED entries: import torch first (set PCS_NO_TORCH=1 for a torch-free process).
        if os.environ.get("PCS_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        handle = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code: int) -> None:
    if code != PCS_OK:
        raise PcsError(code, (lib().pcs_last_error() or b"").decode("utf-8", "replace"))
