"""Caller of the path: pyCamSet optimisation/optimisation_handling.py:24-117, kept in Python.

``run_bundle_adjustment`` hands the handler's closures to ``scipy.optimize.least_squares`` with the
same keyword arguments as the reference (oh:88-98).  The reference then rebuilds a CameraSet
(oh:109); that data model is outside the path, so the full parameter slabs are returned instead.
"""
from __future__ import annotations

import logging
import time

import numpy as np
from scipy.optimize import least_squares


def make_optimisation_function(param_handler, threads: int = 1):  # oh:24-49
    init_params = param_handler.get_initial_params()
    bundle_loss_fun = param_handler.make_loss_fun(threads)
    bundle_loss_jac = param_handler.make_loss_jac(threads) if param_handler.can_make_jac() else None
    return bundle_loss_fun, bundle_loss_jac, init_params


def run_bundle_adjustment(param_handler, threads: int = 1):  # oh:52-117
    loss_fn, bundle_jac, init_params = make_optimisation_function(param_handler, threads)
    init_err = loss_fn(init_params)
    init_euclid = np.mean(np.linalg.norm(np.reshape(init_err, (-1, 2)), axis=1))
    logging.info(f"found {len(init_params):.2e} parameters")
    logging.info(f"found {len(init_err):.2e} control points")
    logging.info(f"Initial Euclidean error: {init_euclid:.2f} px")
    if (init_euclid > 150) or np.isnan(init_euclid):
        logging.critical("Found worryingly high/NaN initial error: check that the initial parametisation is sensible")
    start = time.time()
    optimisation = least_squares(
        loss_fn,
        init_params,
        verbose=param_handler.problem_opts["verbosity"],
        jac=bundle_jac if bundle_jac is not None else "2-point",
        max_nfev=param_handler.problem_opts["max_nfev"],
        x_scale="jac",
    )
    end = time.time()
    final_euclid = np.mean(np.linalg.norm(np.reshape(optimisation.fun, (-1, 2)), axis=1))
    logging.info(f"Final Euclidean error: {final_euclid:.2f} px")
    logging.info(f"Optimisation took {end - start: .2f} seconds.")
    if final_euclid > 5:
        logging.critical("Remaining error is very large: please check the output results")
    slabs = tuple(np.array(a) for a in param_handler.get_bundle_adjustment_inputs(optimisation.x))
    return optimisation, slabs
