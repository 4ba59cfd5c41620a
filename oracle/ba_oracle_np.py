"""NumPy-vectorised twin of the CPU oracle (TEST INFRASTRUCTURE — see ba_oracle.c header).

A third, independently written statement of the same path, vectorised over the detections: easier
to read than the per-detection C code and a cross-check of it (tests/test_oracle_golden.py compares
both with the reference's golden vectors).  Same algorithm as the reference — Rodrigues and its
Jacobian are evaluated per *detection* (after gathering the per-camera / per-image parameters),
nothing is hoisted.  Only tests/ and bench.py's cpu_baseline leg may import this module.

Citations: fbi = pyCamSet/optimisation/function_block_implementations.py,
ch = compiled_helpers.py, mm = matmul_map.py, afb = abstract_function_blocks.py.
"""
from __future__ import annotations

import numpy as np

CHAIN_P = {"template": 21, "self": 24, "free": 18}


def rodrigues(r: np.ndarray) -> np.ndarray:
    """(n,3) rotation vectors -> (n,3,3) matrices; ch:197-235 with the theta < 1e-10 branch."""
    theta = np.sqrt(np.sum(r * r, axis=1))
    small = theta < 1e-10
    th = np.where(small, 1.0, theta)
    ct, st = np.cos(th), np.sin(th) / th
    rr = r[:, :, None] * r[:, None, :]
    R = rr * ((1 - ct) / th ** 2)[:, None, None]
    idx = np.arange(3)
    R[:, idx, idx] += ct[:, None]
    R[:, 0, 1] -= r[:, 2] * st; R[:, 1, 0] += r[:, 2] * st
    R[:, 0, 2] += r[:, 1] * st; R[:, 2, 0] -= r[:, 1] * st
    R[:, 1, 2] -= r[:, 0] * st; R[:, 2, 1] += r[:, 0] * st
    R[small] = np.eye(3, dtype=r.dtype)
    return R


def rodrigues_jac(r: np.ndarray) -> np.ndarray:
    """(n,3) -> (n,3,3,3): out[n, a] = d R / d r_a; ch:237-286 (port of OpenCV's formula)."""
    n = r.shape[0]
    theta = np.sqrt(np.sum(r * r, axis=1))
    small = theta < 1e-10
    th = np.where(small, 1.0, theta)
    it = 1.0 / th
    ct, st = np.cos(th), np.sin(th)
    ct_1 = 1 - ct
    ax = r * it[:, None]
    x, y, z = ax[:, 0], ax[:, 1], ax[:, 2]
    zero = np.zeros(n, dtype=r.dtype)
    rrt = ax[:, :, None] * ax[:, None, :]
    r_x = np.stack([np.stack([zero, -z, y], 1), np.stack([z, zero, -x], 1), np.stack([-y, x, zero], 1)], 1)
    eye = np.eye(3, dtype=r.dtype)[None]
    # d(rr^T)/d axis_a and d[r]x/d axis_a
    drrt = np.zeros((n, 3, 3, 3), dtype=r.dtype)
    for a in range(3):
        drrt[:, a, a, :] += ax
        drrt[:, a, :, a] += ax
    drx = np.zeros((3, 3, 3))
    drx[0, 1, 2], drx[0, 2, 1] = -1, 1
    drx[1, 0, 2], drx[1, 2, 0] = 1, -1
    drx[2, 0, 1], drx[2, 1, 0] = -1, 1
    out = np.empty((n, 3, 3, 3), dtype=r.dtype)
    for a in range(3):
        ri = ax[:, a]
        a0 = -st * ri
        a1 = (st - 2 * ct_1 * it) * ri
        a2 = ct_1 * it
        a3 = (ct - st * it) * ri
        a4 = st * it
        out[:, a] = (a0[:, None, None] * eye + a1[:, None, None] * rrt + a2[:, None, None] * drrt[:, a]
                     + a3[:, None, None] * r_x + a4[:, None, None] * drx[a][None])
    gen = np.zeros((3, 3, 3))  # theta < 1e-10: the generators (ch:246-254)
    gen[0, 1, 2], gen[0, 2, 1] = -1, 1
    gen[1, 0, 2], gen[1, 2, 0] = 1, -1
    gen[2, 0, 1], gen[2, 1, 0] = -1, 1
    out[small] = gen
    return out


def _projection(intr, X):
    """fbi:27-47 and fbi:50-140 for (n,9) parameters and (n,3) camera-frame points.
    Returns uv (n,2), A_p (n,2,9), A_x (n,2,3); the Jacobian is written like the reference's sympy
    output, with the z**k denominators."""
    fx, px, fy, py, k0, k1, p0, p1, k2 = intr.T
    x, y, z = X.T
    u = (fx * x + px * z) / z
    v = (fy * y + py * z) / z
    xn, yn = (u - px) / fx, (v - py) / fy
    r2 = xn ** 2 + yn ** 2
    kup = 1 + k0 * r2 + k1 * r2 ** 2 + k2 * r2 ** 3
    xD = xn * kup + 2 * p0 * xn * yn + p1 * (r2 + 2 * xn ** 2)
    yD = yn * kup + p0 * (r2 + 2 * yn ** 2) + 2 * p1 * xn * yn
    uv = np.stack([xD * fx + px, yD * fy + py], 1)
    s = x ** 2 + y ** 2
    rad = k0 * z ** 4 * s + k1 * z ** 2 * s ** 2 + k2 * s ** 3 + z ** 6
    drad = k0 * z ** 4 + 2 * k1 * z ** 2 * s + 3 * k2 * s ** 2
    one, zero = np.ones_like(x), np.zeros_like(x)
    row_u = [(x * rad + z ** 5 * (2 * p0 * x * y + p1 * (3 * x ** 2 + y ** 2))) / z ** 7, one, zero, zero,
             fx * x * s / z ** 3, fx * x * s ** 2 / z ** 5, 2 * fx * x * y / z ** 2, fx * (3 * x ** 2 + y ** 2) / z ** 2,
             fx * x * s ** 3 / z ** 7]
    row_v = [zero, zero, (y * rad + z ** 5 * (p0 * (x ** 2 + 3 * y ** 2) + 2 * p1 * x * y)) / z ** 7, one,
             fy * y * s / z ** 3, fy * y * s ** 2 / z ** 5, fy * (x ** 2 + 3 * y ** 2) / z ** 2, 2 * fy * x * y / z ** 2,
             fy * y * s ** 3 / z ** 7]
    A_p = np.stack([np.stack(row_u, 1), np.stack(row_v, 1)], 1)
    cross = x * y * drad + z ** 5 * (p0 * x + p1 * y)
    dxdx = fx * (rad + 2 * x ** 2 * drad + 2 * z ** 5 * (p0 * y + 3 * p1 * x)) / z ** 7
    dxdy = 2 * fx * cross / z ** 7
    dxdz = -fx * (4 * p0 * x * y * z ** 5 + 2 * p1 * z ** 5 * (3 * x ** 2 + y ** 2) + 2 * x * s * drad + x * rad) / z ** 8
    dydx = 2 * fy * cross / z ** 7
    dydy = fy * (rad + 2 * y ** 2 * drad + 2 * z ** 5 * (3 * p0 * y + p1 * x)) / z ** 7
    dydz = -fy * (2 * p0 * z ** 5 * (x ** 2 + 3 * y ** 2) + 4 * p1 * x * y * z ** 5 + 2 * y * s * drad + y * rad) / z ** 8
    A_x = np.stack([np.stack([dxdx, dxdy, dxdz], 1), np.stack([dydx, dydy, dydz], 1)], 1)
    return uv, A_p, A_x


def evaluate(chain: str, det: np.ndarray, param_str: np.ndarray, template=None, counts=None, want_jac: bool = True, dtype=np.float64):
    """-> (resid (N,2), dense (2N,P) | None) with the layout of the generated full_loss / full_jac
    (afb:350-387, afb:552-599 + afb:641).  ``dtype=np.longdouble`` evaluates the same formulas in x87 extended precision
    (64-bit mantissa): the yardstick tests use to tell the kernel's rounding from the float64 oracle's own."""
    det = np.asarray(det, dtype=np.float64)
    param_str = np.asarray(param_str, dtype=dtype)
    c, im, k = det[:, 0].astype(np.int64), det[:, 1].astype(np.int64), det[:, 2].astype(np.int64)
    C, I = counts[:2] if counts is not None else (int(c.max()) + 1, int(im.max()) + 1)
    intr = param_str[: 9 * C].reshape(C, 9)[c]
    extr = param_str[9 * C: 15 * C].reshape(C, 6)[c]
    if chain == "free":
        X = param_str[15 * C:].reshape(-1, 3)[k]
        Xw = X
    else:
        pose = param_str[15 * C: 15 * C + 6 * I].reshape(I, 6)[im]
        X = np.asarray(template, dtype=dtype)[k] if chain == "template" else param_str[15 * C + 6 * I:].reshape(-1, 3)[k]
        Rp = rodrigues(pose[:, :3])
        Xw = np.einsum("nij,nj->ni", Rp, X) + pose[:, 3:]          # fbi:150-155
    Re = rodrigues(extr[:, :3])
    Xc = np.einsum("nij,nj->ni", Re, Xw) + extr[:, 3:]
    uv, A_p, A_x = _projection(intr, Xc)
    resid = uv - det[:, 3:].astype(dtype)                                          # afb:384
    if not want_jac:
        return resid, None
    dRe = rodrigues_jac(extr[:, :3])
    E_r = np.einsum("naij,nj->nia", dRe, Xw)                         # fbi:163-170: column a = dR_e[a] X_w
    S = np.einsum("nri,nij->nrj", A_x, Re)                           # A_x . R_e
    blocks = [A_p, np.einsum("nri,nia->nra", A_x, E_r), A_x]
    if chain != "free":
        dRp = rodrigues_jac(pose[:, :3])
        Q_r = np.einsum("naij,nj->nia", dRp, X)
        blocks += [np.einsum("nri,nia->nra", S, Q_r), S]
        if chain == "self":
            blocks.append(np.einsum("nri,nij->nrj", S, Rp))          # free_point Jacobian = I
    else:
        blocks.append(S)
    J = np.concatenate(blocks, axis=2)                               # (N, 2, P): the chain rule of mm:147-243
    return resid, J.reshape(-1, J.shape[2])
