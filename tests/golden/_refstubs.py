"""Stand-in modules that let the *unmodified* pyCamSet sources be imported as plain Python.

TEST INFRASTRUCTURE ONLY (used by tests/golden/make_golden.py in the build container).

pyCamSet's hot path is numba ``@njit`` Python.  numba (and cv2, pyvista, ...) are not
installed in the build image and there is no network, so this file provides:

* ``numba``: ``njit``/``jit`` are identity decorators -> every function body runs as ordinary
  IEEE-754 CPython.  ``prange`` is ``range``.  The few type objects the sources touch at
  import time are inert placeholders.
* permissive placeholder modules for cv2 / pyvista / natsort / coloredlogs / blosc / uniplot,
  which the package ``__init__`` chain imports but the optimisation path never calls.

One reference quirk needs emulation: ``matmul_map.create_optimisable_compute_flow``
(matmul_map.py:185-192) calls ``template_points.compute_jac`` with a zero-length ``inp``
(``num_inp = 0``, function_block_implementations.py:190).  Under numba that is an unchecked
out-of-bounds read of arbitrary non-zero memory; the structure discovery then only tests
``== 0`` / ``== 1`` (matmul_map.py:25-30).  The ``njit`` stub therefore pads a too-short
``inp`` with random values for functions declared with the 4-array signature string.
"""
from __future__ import annotations

import functools
import sys
import types

import numpy as np


class _Anything:
    """Inert object: callable, subscriptable, attribute-able; int()/index() give 0."""

    def __init__(self, name="anything"):
        self._name = name

    def __call__(self, *a, **k):
        return _Anything(self._name + "()")

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Anything(self._name + "." + item)

    def __getitem__(self, item):
        return _Anything(self._name + "[]")

    def __int__(self):
        return 0

    def __index__(self):
        return 0

    def __iter__(self):
        return iter(())

    def __repr__(self):
        return f"<stub {self._name}>"


class _PermissiveModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        if item.startswith("DICT_") or item.isupper():
            return 0
        return _Anything(f"{self.__name__}.{item}")


_rng = np.random.default_rng(12345)


def _njit(*args, **kwargs):
    """Identity decorator in all three spellings used by the reference:
    ``@njit``, ``@njit(cache=True, ...)``, ``@njit("sig", cache=True)`` and ``njit(lambda)``."""
    if len(args) == 1 and callable(args[0]):  # @njit, njit(lambda), njit(fn, fastmath=True, cache=True)
        return args[0]
    signature = args[0] if args and isinstance(args[0], str) else None

    def deco(fn):
        if signature is None:
            return fn

        @functools.wraps(fn)  # sets __wrapped__ so inspect.unwrap() finds the source
        def wrapper(params=None, inp=None, output=None, memory=None, **kw):
            params = kw.pop("params", params)
            inp = kw.pop("inp", inp)
            output = kw.pop("output", output)
            memory = kw.pop("memory", memory)
            if inp is not None and len(inp) < 3:
                padded = _rng.random(3) + 0.1
                padded[: len(inp)] = inp
                inp = padded
            return fn(params, inp, output, memory)

        return wrapper

    return deco


def install():
    """Register the stand-in modules in ``sys.modules`` (idempotent)."""
    if "numba" in sys.modules and getattr(sys.modules["numba"], "_pcs_stub", False):
        return
    numba = types.ModuleType("numba")
    numba._pcs_stub = True
    numba.njit = _njit
    numba.jit = _njit
    numba.prange = range
    numba.gdb_init = lambda *a, **k: None
    numba.int64 = int
    numba.void = _Anything("numba.void")
    numba.float64 = _Anything("numba.float64")
    nb_types = types.ModuleType("numba.types")
    nb_types.FunctionType = _Anything("numba.types.FunctionType")
    nb_types.Array = _Anything("numba.types.Array")
    nb_types.UniTuple = _Anything("numba.types.UniTuple")
    nb_types.int_ = int
    nb_typed = types.ModuleType("numba.typed")
    nb_typed.List = list
    numba.types = nb_types
    numba.typed = nb_typed
    sys.modules["numba"] = numba
    sys.modules["numba.types"] = nb_types
    sys.modules["numba.typed"] = nb_typed

    for name in ("cv2", "cv2.aruco", "pyvista", "natsort", "coloredlogs", "blosc", "uniplot",
                 "pyvistaqt", "sklearn", "sklearn.decomposition"):
        if name in sys.modules:
            continue
        sys.modules[name] = _PermissiveModule(name)
    sys.modules["cv2"].aruco = sys.modules["cv2.aruco"]
    sys.modules["natsort"].natsorted = sorted
