"""Shared helpers for the test-suite (tolerance rules, slab selection)."""
import numpy as np

# Product tolerance (BASELINE.json north_star: 1e-10 relative FP64):
#   Jacobian:  |a - b| <= 1e-10 * max(|ref|, 1e-6 * ||row||_inf)
#   residual:  |a - b| <= 1e-10 * max(|ref|, 1e-3 * max(|u|,|v|))   (a residual is the difference of
#              two ~500 px coordinates; its relative accuracy is meaningful on the projection)
JAC_RTOL = 1e-10
ROW_FLOOR = 1e-6
RES_RTOL = 1e-10
# FP32 engine (config 5): FP32 measurements in, FP32 residual / Jacobian out, FP64 arithmetic on FP64 slabs in between
# (round 1's all-float arithmetic lost 5e-3 to cancellation in the chain rule and was removed).  Jacobian: one rounding
# to float at the store; residual: the measurement itself is rounded to float on upload (half an ulp of ~1e3 px = 6e-5 px).
F32_JAC_RTOL = 1.2e-7
F32_RES_ATOL = 2e-4  # pixels
# mixed engine: FP64 arithmetic, one rounding to FP32 at the store (2^-24 = 6e-8 relative per value)
MIXED_JAC_RTOL = 1.2e-7
MIXED_RES_RTOL = 1.2e-7


def chain_slabs(rig, chain):
    if chain == "template":
        return [rig.intr, rig.extr, rig.poses]
    if chain == "self":
        return [rig.intr, rig.extr, rig.poses, rig.points]
    return [rig.intr, rig.extr, rig.points]


def jac_rel_err(a, ref, rows=None):
    a, ref = np.asarray(a), np.asarray(ref)
    if rows is None:
        rows = np.max(np.abs(ref), axis=1, keepdims=True)
    return float(np.max(np.abs(a - ref) / np.maximum(np.abs(ref), ROW_FLOOR * rows))) if ref.size else 0.0


def resid_rel_err(r, ref, uv):
    scale = np.maximum(np.abs(ref), 1e-3 * np.max(np.abs(uv), axis=1, keepdims=True))
    return float(np.max(np.abs(r - ref) / scale)) if ref.size else 0.0


def assert_jac_close(a, ref, rtol=JAC_RTOL, rows=None):
    err = jac_rel_err(a, ref, rows)
    assert err <= rtol, f"Jacobian max rel err {err:.3e} > {rtol:.1e}"


def assert_resid_close(r, ref, uv, rtol=RES_RTOL):
    err = resid_rel_err(r, ref, uv)
    assert err <= rtol, f"residual max rel err {err:.3e} > {rtol:.1e}"


def jac_error_report(a, ref, P):
    """Where a Jacobian (rows x P) differs from a reference, beyond the one floored figure the gate uses:
    floored      max |a - ref| / max(|ref|, 1e-6 ||row||_inf)            (the product tolerance's rule)
    plain        max |a - ref| / |ref| over ALL non-zero reference entries (nothing excused by the floor)
    col_*        the block column (0 .. P - 1) the worst entry of either kind sits in
    under_floor  share of the non-zero reference entries smaller than 1e-6 of their row's largest"""
    a, ref = np.asarray(a, dtype=np.longdouble).reshape(-1, P), np.asarray(ref, dtype=np.longdouble).reshape(-1, P)
    rows = np.max(np.abs(ref), axis=1, keepdims=True)
    d = np.abs(a - ref)
    fl = d / np.maximum(np.abs(ref), ROW_FLOOR * rows)
    nz = ref != 0
    pl = np.where(nz, d / np.where(nz, np.abs(ref), 1), 0)
    return {"floored": float(fl.max()), "col_floored": int(np.argmax(fl.max(axis=0))), "plain": float(pl.max()), "col_plain": int(np.argmax(pl.max(axis=0))),
            "under_floor": float(np.mean((np.abs(ref) < ROW_FLOOR * rows)[nz]))}


def describe(rep):
    return (f"floored {rep['floored']:.2e} (column {rep['col_floored']}), plain {rep['plain']:.2e} (column {rep['col_plain']}), "
            f"{100 * rep['under_floor']:.1f} % of the non-zero entries under the row floor")


def user_blocks(fb):
    """The user blocks of tests/golden/_user_blocks.py (written there on the reference's ABC) as device code."""

    class cam_scale(fb.device_function_block):
        num_inp, num_out = 3, 3
        params = fb.param_type(fb.key_type.PER_CAM, 1)
        device_fun = "for (int i = 0; i < 3; ++i) out[i] = params[0] * inp[i];"
        device_jac = """for (int o = 0; o < 3; ++o) {
            out[o * 4 + 0] = inp[o];
            for (int i = 0; i < 3; ++i) out[o * 4 + 1 + i] = (o == i) ? params[0] : 0.0;
        }"""

    class division_projection(fb.device_function_block):
        num_inp, num_out = 3, 2
        params = fb.param_type(fb.key_type.PER_CAM, 5)
        device_fun = """const double iz = 1.0 / inp[2], x = inp[0] * iz, y = inp[1] * iz;
        const double d = 1.0 / (1.0 + params[4] * (x * x + y * y));
        out[0] = params[0] * x * d + params[1];
        out[1] = params[2] * y * d + params[3];"""
        device_jac = """const double iz = 1.0 / inp[2], x = inp[0] * iz, y = inp[1] * iz, r2 = x * x + y * y;
        const double d = 1.0 / (1.0 + params[4] * r2), d2 = d * d;
        for (int q = 0; q < 16; ++q) out[q] = 0.0;
        out[0] = x * d; out[1] = 1.0; out[4] = -params[0] * x * r2 * d2;
        const double ux = params[0] * (d - 2.0 * params[4] * x * x * d2), uy = -2.0 * params[0] * params[4] * x * y * d2;
        out[5] = ux * iz; out[6] = uy * iz; out[7] = -(x * ux + y * uy) * iz;
        out[8 + 2] = y * d; out[8 + 3] = 1.0; out[8 + 4] = -params[2] * y * r2 * d2;
        const double vx = -2.0 * params[2] * params[4] * x * y * d2, vy = params[2] * (d - 2.0 * params[4] * y * y * d2);
        out[8 + 5] = vx * iz; out[8 + 6] = vy * iz; out[8 + 7] = -(x * vx + y * vy) * iz;"""

    class board_flex(fb.device_function_block):
        # a TEMPLATED source: template = True on the last block -> `inp` is the detection's template point (afb:374-375)
        template = True
        num_inp, num_out, array_memory = 0, 3, 0
        params = fb.param_type(fb.key_type.PER_IMG, 5)
        device_fun = """out[0] = params[0] * inp[0] + params[2];
        out[1] = params[1] * inp[1] + params[3];
        out[2] = inp[2] + params[4] * (inp[0] * inp[0] + inp[1] * inp[1]);"""
        device_jac = """for (int q = 0; q < 15; ++q) out[q] = 0.0;
        out[0] = inp[0]; out[2] = 1.0;
        out[5 + 1] = inp[1]; out[5 + 3] = 1.0;
        out[10 + 4] = inp[0] * inp[0] + inp[1] * inp[1];"""

    return {"cam_scale": cam_scale, "division_projection": division_projection, "board_flex": board_flex}
