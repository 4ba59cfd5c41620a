"""The caller of the path, kept in Python like the reference
(pyCamSet optimisation/optimisation_handling.py:24-117).

``make_optimisation_function`` returns the handler's two closures plus the start vector;
``run_bundle_adjustment`` feeds them to ``scipy.optimize.least_squares`` with the reference's
keyword arguments (jac = the CSR closure, ``x_scale='jac'``, ``max_nfev`` and ``verbose`` from the
handler's options, oh:88-98).  The reference finishes by rebuilding a ``CameraSet`` (oh:109); that
data model is outside the accelerated path, so the full parameter slabs at the solution are
returned instead.  ``solver='device'`` swaps scipy for ``device_solver.lm_solve`` (the Jacobian never
leaves the GPU).
"""
from __future__ import annotations

import logging
import time

import numpy as np
from scipy.optimize import least_squares

log = logging.getLogger(__name__)


def mean_reprojection_error(residuals: np.ndarray) -> float:
    """Mean Euclidean pixel error of a flat residual vector [u0, v0, u1, v1, ...] (the figure the
    reference logs before and after the solve, oh:66-70, oh:101-103)."""
    uv = np.asarray(residuals, dtype=np.float64).reshape(-1, 2)
    return float(np.hypot(uv[:, 0], uv[:, 1]).mean())


def make_optimisation_function(param_handler, threads: int = 1):
    """(loss_fn, jac_fn | None, x0) — oh:24-49."""
    x0 = param_handler.get_initial_params()
    loss_fn = param_handler.make_loss_fun(threads)
    jac_fn = param_handler.make_loss_jac(threads) if param_handler.can_make_jac() else None
    return loss_fn, jac_fn, x0


def run_bundle_adjustment(param_handler, threads: int = 1, solver: str = "scipy", linear_solver: str = "auto"):
    """Solve the handler's problem; returns (result, parameter slabs at the solution) — oh:52-117.
    ``solver='device'`` runs device_solver.lm_solve with ``linear_solver`` in {'auto', 'cholesky', 'pcg'}."""
    loss_fn, jac_fn, x0 = make_optimisation_function(param_handler, threads)
    opts = param_handler.problem_opts
    start_error = mean_reprojection_error(loss_fn(x0))
    log.info("%d parameters, %d residuals, start error %.2f px", len(x0), 2 * param_handler._flat_detections().shape[0], start_error)
    if not np.isfinite(start_error) or start_error > 150:  # the reference's sanity threshold (oh:80-83)
        log.critical("start error is very high or not finite: check the initial parameters")
    tic = time.perf_counter()
    if solver == "device":
        from .device_solver import lm_solve

        result = lm_solve(param_handler, x0, max_iter=opts["max_nfev"], linear_solver=linear_solver)
        end_error = mean_reprojection_error(loss_fn(result.x))
    else:
        result = least_squares(loss_fn, x0, jac=jac_fn if jac_fn is not None else "2-point", x_scale="jac",
                               max_nfev=opts["max_nfev"], verbose=opts["verbosity"])
        end_error = mean_reprojection_error(result.fun)
    log.info("solved in %.2f s, final error %.2f px", time.perf_counter() - tic, end_error)
    if end_error > 5:  # oh:105-107
        log.critical("remaining error is very large: check the result")
    slabs = tuple(np.array(a) for a in param_handler.get_bundle_adjustment_inputs(result.x))
    return result, slabs
