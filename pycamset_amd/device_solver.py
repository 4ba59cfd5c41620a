"""Levenberg-Marquardt with the Jacobian kept on the device.

The reference hands a 2N x n_free CSR matrix to ``scipy.optimize.least_squares``
(optimisation_handling.py:88-98), which scales its columns (x_scale='jac'), forms J^T f and runs
lsmr mat-vecs on the host.  Here J only exists as matrix-free products on the GPU
(csrc/ba_matfree.hpp): per LM iteration the host sees n_params-sized vectors only, so neither the
PCIe copy of J (335 MB at N = 1e6) nor — when sharded — an all-gather of J is needed: ranks
all-reduce J^T J v, diag(J^T J) and J^T r (a few KB).

    op  = JacobianOperator(engine, unfixed_mask)          # products in the FREE parameter space
    res = lm_solve(handler, x0)                           # drop-in for run_bundle_adjustment's solve

``as_linear_operator()`` exposes J as a scipy LinearOperator (matvec / rmatvec) for callers that
want scipy's own solvers without materialising J.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


class JacobianOperator:
    """J(x) restricted to the free parameters, at the engine's current linearisation point.

    ``reduce_fn`` (optional) sums an n-vector across ranks (sharded detections): it is applied to every
    parameter-space result and to the cost.
    """

    def __init__(self, engine, unfixed=None, reduce_fn=None):
        self.eng = engine
        mask = np.ones(engine.n_params, dtype=bool) if unfixed is None else np.asarray(unfixed, dtype=bool)
        if mask.shape[0] != engine.n_params:
            raise ValueError("mask must have one entry per parameter")
        self.free = np.flatnonzero(mask)
        self.n_free = self.free.shape[0]
        self.reduce_fn = reduce_fn
        self._full = np.zeros(engine.n_params)

    def _expand(self, v_free):
        self._full[:] = 0.0
        self._full[self.free] = v_free
        return self._full

    def _red(self, v):
        return self.reduce_fn(v) if self.reduce_fn is not None else v

    def linearize(self, param_str):
        self.eng.linearize(param_str)

    def jv(self, v_free):          # (2N_local,) — stays per rank
        return self.eng.jv(self._expand(v_free))

    def jtu(self, u):
        return self._red(self.eng.jtu(u)[self.free])

    def jtjv(self, v_free):
        return self._red(self.eng.jtjv(self._expand(v_free))[self.free])

    def diag(self):
        return self._red(self.eng.jtj_diag()[self.free])

    def grad(self):
        g, c = self.eng.grad()
        if self.reduce_fn is not None:
            packed = self.reduce_fn(np.concatenate([g[self.free], [c]]))
            return packed[:-1], float(packed[-1])
        return g[self.free], c

    def as_linear_operator(self):
        from scipy.sparse.linalg import LinearOperator

        return LinearOperator((2 * self.eng.n, self.n_free), matvec=self.jv, rmatvec=self.jtu, dtype=np.float64)


def pcg(apply_a, b, m_inv, tol: float, max_iter: int):
    """Jacobi-preconditioned conjugate gradients for the SPD damped normal equations."""
    x = np.zeros_like(b)
    r = b.copy()
    z = m_inv * r
    p = z.copy()
    rz = float(r @ z)
    b_norm = float(np.sqrt(b @ (m_inv * b))) or 1.0
    it = 0
    for it in range(1, max_iter + 1):
        ap = apply_a(p)
        pap = float(p @ ap)
        if pap <= 0:
            break
        alpha = rz / pap
        x += alpha * p
        r -= alpha * ap
        z = m_inv * r
        rz_new = float(r @ z)
        if np.sqrt(max(rz_new, 0.0)) <= tol * b_norm:
            break
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, it


@dataclass
class DeviceLMResult:
    x: np.ndarray
    cost: float                 # 0.5 * sum r^2, like scipy's OptimizeResult.cost
    grad: np.ndarray
    optimality: float
    nit: int
    nfev: int
    n_jtjv: int
    status: int
    message: str
    history: list = field(default_factory=list)


def lm_solve(handler, x0, *, max_iter: int = 50, ftol: float = 1e-8, xtol: float = 1e-8, gtol: float = 1e-8,
             cg_tol: float = 1e-3, cg_max_iter: int = 200, lam0: float = 1e-3, reduce_fn=None, verbose: int = 0,
             operator: JacobianOperator | None = None) -> DeviceLMResult:
    """Levenberg-Marquardt (Marquardt scaling D = diag(J^T J), damped normal equations solved by
    Jacobi-PCG on matrix-free J^T J products) for a pycamset_amd handler.  Every quantity that
    depends on the detections is computed by the HIP engine; the host only does n_free-vector algebra.
    ``operator`` replaces the engine-backed JacobianOperator (used by the CPU tests of this driver)."""
    op_fun = handler.op_fun
    if operator is not None:
        op = operator
    else:
        dd = handler._flat_detections()
        eng = op_fun._engine_for(dd)
        op_fun._bind_template(eng, handler._template_arg())
        op = JacobianOperator(eng, handler._jac_mask(), reduce_fn=reduce_fn)

    def param_str(x):
        return op_fun.build_param_list(*handler.get_bundle_adjustment_inputs(x))

    x = np.array(x0, dtype=np.float64)
    op.linearize(param_str(x))
    g, sumsq = op.grad()
    d = op.diag()
    nfev, n_jtjv, lam = 1, 0, lam0
    history = [0.5 * sumsq]
    status, message = 0, "maximum number of iterations reached"
    it = 0
    for it in range(1, max_iter + 1):
        gnorm = float(np.max(np.abs(g)))
        if gnorm <= gtol:
            status, message = 1, "gtol reached"
            break
        accepted = False
        for _ in range(12):  # damping retries
            dd_ = np.maximum(d, 1e-300)
            delta, k = pcg(lambda v: op.jtjv(v) + lam * dd_ * v, -g, 1.0 / ((1.0 + lam) * dd_), cg_tol, cg_max_iter)
            n_jtjv += k
            x_new = x + delta
            op.linearize(param_str(x_new))
            g_new, sumsq_new = op.grad()
            nfev += 1
            pred = 0.5 * float(-(g @ delta) + lam * (delta @ (dd_ * delta)))
            actual = 0.5 * (sumsq - sumsq_new)
            rho = actual / pred if pred > 0 else -1.0
            if verbose:
                print(f"  it {it}: lam {lam:.2e} cg {k} cost {0.5 * sumsq:.6e} -> {0.5 * sumsq_new:.6e} rho {rho:.3f}")
            if np.isfinite(sumsq_new) and actual > 0:
                accepted = True
                step_norm, x_norm = float(np.linalg.norm(delta)), float(np.linalg.norm(x))
                rel_drop = actual / (0.5 * sumsq)
                x, g, sumsq = x_new, g_new, sumsq_new
                d = op.diag()
                lam = max(lam * (1.0 / 3.0 if rho > 0.75 else 1.0 if rho > 0.25 else 2.0), 1e-12)
                history.append(0.5 * sumsq)
                break
            lam *= 4.0
        if not accepted:
            op.linearize(param_str(x))  # the slabs hold the rejected trial point
            status, message = 2, "no further decrease (damping exhausted)"
            break
        if rel_drop <= ftol:
            status, message = 3, "ftol reached"
            break
        if step_norm <= xtol * (xtol + x_norm):
            status, message = 4, "xtol reached"
            break
    return DeviceLMResult(x=x, cost=0.5 * sumsq, grad=g, optimality=float(np.max(np.abs(g))), nit=it, nfev=nfev,
                          n_jtjv=n_jtjv, status=status, message=message, history=history)
