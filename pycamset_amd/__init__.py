"""pycamset_amd — MI355X-native bundle-adjustment cost/Jacobian engine behind pyCamSet's
ParamHandler / optimisation_function surface.

Importing the package does not touch the GPU; the HIP extension (libpcs_hip.so) is loaded on
first use and there is no CPU fallback.
"""
from .detections import TargetDetection  # noqa: F401

__all__ = ["TargetDetection", "Engine", "handlers", "function_blocks", "compiled_helpers", "device_solver", "sharding", "synthetic"]
__version__ = "0.1.0"


def __getattr__(name):
    if name == "Engine":
        from .engine import Engine
        return Engine
    if name in ("handlers", "function_blocks", "synthetic", "engine", "optimisation_handling", "sharding", "device_solver",
                "compiled_helpers", "detections"):
        import importlib
        return importlib.import_module(f".{name}", __name__)
    raise AttributeError(name)
