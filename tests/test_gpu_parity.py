"""Parity tests proper (MI355X): every call goes through the C ABI (libpcs_hip.so) and is checked
against the CPU oracle, the golden fixtures generated from the reference, or a size-independent
property.  Tolerances are stated in tests/helpers.py."""
import numpy as np
import pytest

from oracle import ba_oracle as orc
from pycamset_amd import _capi, synthetic
from tests import helpers as H

pytestmark = pytest.mark.gpu

CHAINS = ["template", "self", "free"]


@pytest.fixture(scope="module")
def Engine():
    from pycamset_amd.engine import Engine as E
    assert _capi.lib().pcs_device_count() > 0, "no HIP device: the GPU tests must run on an MI355X"
    return E


def make_engine(Engine, rig, chain, dtype="f64", det=None):
    e = Engine(chain, rig.n_cams, rig.n_imgs, rig.n_keys, dtype=dtype)
    e.set_detections_table(rig.detections if det is None else det)
    if chain == "template":
        e.set_template(rig.points)
    return e


def oracle_eval(rig, chain, det=None, threads=8):
    det = rig.detections if det is None else det
    ps = orc.build_param_list(*H.chain_slabs(rig, chain))
    tm = rig.points if chain == "template" else None
    counts = (rig.n_cams, rig.n_imgs, rig.n_keys)
    j, r = orc.full_jac_dense(chain, det, ps, tm, threads=threads, fast=False, with_resid=True, counts=counts)
    return ps, r, j


# ---- golden fixtures (reference outputs) -------------------------------------------------------
@pytest.mark.parametrize("chain", CHAINS)
@pytest.mark.parametrize("tag", ["tiny", "medium"])
def test_hip_against_reference_goldens(Engine, golden_dir, chain, tag):
    g = np.load(golden_dir / f"block_{chain}_{tag}.npz")
    t = "t1" if tag == "tiny" else "t4"
    det, ps = g["detections"], g["param_str"]
    C, I, K = orc.counts_from_detections(det)
    e = Engine(chain, C, I, K)
    e.set_detections_table(det)
    if chain == "template":
        e.set_template(g["points"])
    r, j = e.eval(ps)
    H.assert_resid_close(r, g[f"resid_{t}"], det[:, 3:])
    H.assert_jac_close(j, g[f"data_all_{t}"].reshape(j.shape))
    idx, ptr = e.csr_structure(None)
    assert np.array_equal(idx, g[f"indices_all_{t}"]) and np.array_equal(ptr, g[f"indptr_all_{t}"])
    assert np.array_equal(e.block_param_inds(), g["block_param_inds"])
    # fixed-parameter compaction on the device == data[:n][good_mask] (afb:627-651)
    unfixed = g["unfixed"]
    idx, ptr = e.csr_structure(unfixed)
    assert np.array_equal(idx, g[f"indices_masked_{t}"]) and np.array_equal(ptr, g[f"indptr_masked_{t}"])
    assert e.set_unfixed(unfixed) == g[f"data_masked_{t}"].shape[0]
    rc, data = e.eval_compact(ps, want_resid=True)
    assert np.array_equal(rc, r)
    _, _, m = orc.csr_structure(chain, det, unfixed)
    rows = np.broadcast_to(np.max(np.abs(j), axis=1, keepdims=True), j.shape)[m]
    err = np.max(np.abs(data - g[f"data_masked_{t}"]) / np.maximum(np.abs(g[f"data_masked_{t}"]), H.ROW_FLOOR * rows))
    assert err <= H.JAC_RTOL
    assert np.array_equal(data, j[m])  # same kernel maths, only the store pattern differs
    e.close()


# ---- round 3: the reference's edge rotations THROUGH the HIP slab code ---------------------------------
def _unfloored(j, ref):
    """(worst plain relative error over entries the row floor does not cover, fraction of entries under the floor,
    worst plain relative error over ALL non-zero entries)."""
    rows = np.max(np.abs(ref), axis=1, keepdims=True)
    nz = ref != 0
    under = nz & (np.abs(ref) < H.ROW_FLOOR * rows)
    rel = np.zeros_like(ref)
    rel[nz] = np.abs(j - ref)[nz] / np.abs(ref)[nz]
    above = nz & ~under
    return (float(rel[above].max()) if above.any() else 0.0, float(under.sum() / max(1, nz.sum())), float(rel[nz].max()) if nz.any() else 0.0)


@pytest.mark.parametrize("chain", CHAINS)
@pytest.mark.parametrize("fuse_prep", [0, 1])
def test_edge_rotations_through_the_hip_slab_code(Engine, golden_dir, chain, fuse_prep, capsys):
    """block_*_edge_rot.npz: every camera extrinsic and every target pose of the problem is one of the edge rotation
    vectors of unit_vectors.npz (theta = 0, 1e-11, 5e-11, 2e-9, 4e-5, pi, pi - 1e-9, ~pi, > pi, 2e-7, 4e-4), evaluated by the
    reference's make_full_loss_fn / make_jacobean (afb:656-667).  Both forms of the step — slab_prep launch (0) and the
    waves preparing their own slabs (1) — must match at the product tolerance."""
    g = np.load(golden_dir / f"block_{chain}_edge_rot.npz")
    det, ps = g["detections"], g["param_str"]
    C, I, K = orc.counts_from_detections(det)
    e = Engine(chain, C, I, K)
    e.set_detections_table(det)
    if chain == "template":
        e.set_template(g["points"])
    e.set_option("fuse_prep", fuse_prep)
    r, j = e.eval(ps)
    ref_j = g["data_all_t2"].reshape(j.shape)
    H.assert_resid_close(r, g["resid_t2"], det[:, 3:])
    H.assert_jac_close(j, ref_j)
    worst_above, frac_under, worst_all = _unfloored(j, ref_j)
    with capsys.disabled():
        print(f"\n[edge rotations, chain {chain}, fuse_prep {fuse_prep}] floored err {H.jac_rel_err(j, ref_j):.2e}; plain relative error: "
              f"{worst_above:.2e} over entries above the 1e-6 row floor, {worst_all:.2e} over all non-zero entries; "
              f"{100 * frac_under:.3f} % of the non-zero entries sit under the floor")
    assert worst_above <= H.JAC_RTOL
    # masked (device compaction) against the reference's masked data
    unfixed = g["unfixed"]
    assert e.set_unfixed(unfixed) == g["data_masked_t2"].shape[0]
    _, data = e.eval_compact(ps)
    _, _, m = orc.csr_structure(chain, det, unfixed)
    rows = np.broadcast_to(np.max(np.abs(ref_j), axis=1, keepdims=True), ref_j.shape)[m]
    assert np.max(np.abs(data - g["data_masked_t2"]) / np.maximum(np.abs(g["data_masked_t2"]), H.ROW_FLOOR * rows)) <= H.JAC_RTOL
    e.close()


@pytest.mark.parametrize("chain", CHAINS)
@pytest.mark.parametrize("fuse_prep", [0, 1])
def test_nonfinite_focal_lengths_poison_both_residual_rows_like_the_reference(Engine, golden_dir, chain, fuse_prep):
    """fbi:32-35 (u = (fx x + px z) / z, x_n = (u - px) / fx): fx or fy in {0, NaN, inf} makes BOTH residual rows of the
    camera's detections non-finite in the reference; its Jacobian formulas never form that quotient.  Pinned behaviour of the
    engine: the same rows are non-finite (the slab's principal point is NaN for such a camera, ba_device.hpp), every other
    residual and every Jacobian entry the reference leaves finite agrees at the product tolerance."""
    g = np.load(golden_dir / f"block_{chain}_focal_nonfinite.npz")
    det, ps = g["detections"], g["param_str"]
    C, I, K = orc.counts_from_detections(det)
    e = Engine(chain, C, I, K)
    e.set_detections_table(det)
    if chain == "template":
        e.set_template(g["points"])
    e.set_option("fuse_prep", fuse_prep)
    r, j = e.eval(ps)
    e.close()
    ref_r, ref_j = g["resid_t1"].reshape(r.shape), g["data_all_t1"].reshape(j.shape)
    assert np.array_equal(np.isfinite(r), np.isfinite(ref_r))
    assert not np.isfinite(r[det[:, 0] < 4]).any() and np.isfinite(r[det[:, 0] == 4]).all()
    ok = np.isfinite(ref_r).all(axis=1)
    H.assert_resid_close(r[ok], ref_r[ok], det[ok, 3:])
    # Jacobian: rows of cameras with a ZERO focal length are finite in the reference and must agree; where the reference is
    # non-finite (fx = inf / NaN enter its products) the engine must be non-finite too
    fin = np.isfinite(ref_j)
    assert np.array_equal(np.isfinite(j), fin)
    rows = np.max(np.abs(np.where(fin, ref_j, 0.0)), axis=1, keepdims=True)
    err = np.abs(np.where(fin, j - ref_j, 0.0)) / np.maximum(np.abs(np.where(fin, ref_j, 0.0)), np.maximum(H.ROW_FLOOR * rows, 1e-300))
    assert err.max() <= H.JAC_RTOL


@pytest.mark.parametrize("chain", CHAINS)
@pytest.mark.parametrize("dtype", ["f64", "mixed"])
def test_one_launch_step_is_bit_identical_to_slab_prep_plus_evaluation(Engine, chain, dtype):
    """fuse_prep = 1: every wave prepares the slab of each (camera, image) pair of its tile itself (rot_terms + rot_element,
    the functions slab_prep_kernel is made of) — residual, Jacobian and residual-only mode must not differ in one bit from
    the two-launch step, on a run-ordered table (tiles inside one run and tiles straddling a boundary) and on a shuffled
    one (many pairs per tile)."""
    rig = synthetic.config_rig(1)
    ps = orc.build_param_list(*H.chain_slabs(rig, chain))
    rng = np.random.default_rng(5)
    for det in (rig.detections, rig.detections[rng.permutation(rig.n_det)][:1500]):
        e = make_engine(Engine, rig, chain, dtype=dtype, det=det)
        outs = {}
        for f in (0, 1):
            e.set_option("fuse_prep", f)
            outs[f] = e.eval(ps) + (e.eval(ps, want_jac=False)[0],)
        for a, b in zip(outs[0], outs[1]):
            assert np.array_equal(a, b)
        # a second parameter string: the one-launch step must not see anything of the first (no stale slabs)
        ps2 = ps * (1.0 + 1e-3 * rng.standard_normal(ps.shape[0]))
        e.set_option("fuse_prep", 1)
        r1, j1 = e.eval(ps2)
        e.set_option("fuse_prep", 0)
        r0, j0 = e.eval(ps2)
        assert np.array_equal(r0, r1) and np.array_equal(j0, j1) and not np.array_equal(j0, outs[0][1])
        e.close()


# ---- oracle on seeded synthetic rigs: every chain, every kernel variant ------------------------
@pytest.mark.parametrize("chain", CHAINS)
def test_all_kernel_variants_agree_with_oracle(Engine, chain):
    rig = synthetic.config_rig(1)
    ps, ref_r, ref_j = oracle_eval(rig, chain)
    e = make_engine(Engine, rig, chain)
    outs = []
    for variant in range(8):
        e.set_option("variant", variant)
        r, j = e.eval(ps)
        H.assert_resid_close(r, ref_r, rig.detections[:, 3:])
        H.assert_jac_close(j, ref_j)
        outs.append((r, j))
    for r, j in outs[1:]:  # LDS staging / transposed stores / non-temporal stores never change a bit
        assert np.array_equal(r, outs[0][0]) and np.array_equal(j, outs[0][1])
    for wpc in (1, 3, 16):
        e.set_option("wgs_per_cu", wpc)
        r, j = e.eval(ps)
        assert np.array_equal(j, outs[0][1])
    e.set_option("tiles_per_wg", 4)
    assert np.array_equal(e.eval(ps)[1], outs[0][1])
    r_only, none = e.eval(ps, want_jac=False)
    none2, j_only = e.eval(ps, want_resid=False)
    assert none is None and none2 is None
    assert np.array_equal(r_only, outs[0][0]) and np.array_equal(j_only, outs[0][1])
    e.close()


@pytest.mark.parametrize("chain", CHAINS)
def test_fp32_engine(Engine, chain):
    rig = synthetic.config_rig(1)
    ps, ref_r, ref_j = oracle_eval(rig, chain)
    e = make_engine(Engine, rig, chain, dtype="f32")
    for variant in (0, 3, 6, 7):
        e.set_option("variant", variant)
        r, j = e.eval(ps)
        assert np.max(np.abs(r - ref_r)) <= H.F32_RES_ATOL
        assert H.jac_rel_err(j, ref_j) <= H.F32_JAC_RTOL
    e.close()


@pytest.mark.parametrize("chain", CHAINS)
def test_mixed_engine_has_fp32_bytes_and_fp64_accuracy(Engine, chain):
    """dtype='mixed': FP64 arithmetic on FP64 slabs and measurements, residual / Jacobian stored as FP32.  Every
    value must be the correctly rounded float of the FP64 result up to the FP64 kernel's own 1e-10 — i.e. within
    1.2e-7 relative — on the host path, in the device buffers (float32)
    and through the compaction kernel."""
    import torch
    rig = synthetic.config_rig(1)
    ps, ref_r, ref_j = oracle_eval(rig, chain)
    e = make_engine(Engine, rig, chain, dtype="mixed")
    assert e.np_dtype == np.float32
    rows = np.max(np.abs(ref_j), axis=1, keepdims=True)
    for variant in (0, 3, 6, 7):
        e.set_option("variant", variant)
        r, j = e.eval(ps)
        assert np.array_equal(j, j.astype(np.float32)), "values are float32-representable"
        assert H.jac_rel_err(j, ref_j) <= H.MIXED_JAC_RTOL
        assert np.max(np.abs(r - ref_r) / np.maximum(np.abs(ref_r), 1e-3)) <= H.MIXED_RES_RTOL
    e.set_option("variant", -1)
    d_r = torch.zeros((rig.n_det, 2), dtype=torch.float32, device="cuda")
    d_j = torch.zeros((2 * rig.n_det, e.P), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()                      # the fills run on torch's stream, the evaluation on the engine's own
    e.eval_device(ps, d_r.data_ptr(), d_j.data_ptr())
    e.synchronize()
    assert np.array_equal(d_j.cpu().numpy().astype(np.float64), j) and np.array_equal(d_r.cpu().numpy().astype(np.float64), r)
    mask = np.random.default_rng(2).random(ps.shape[0]) > 0.3
    _, _, m = orc.csr_structure(chain, rig.detections, mask)
    e.set_unfixed(mask)
    r2, data = e.eval_compact(ps, want_resid=True)
    assert np.array_equal(data, j[m]) and np.array_equal(r2, r)
    # products that never leave the GPU are plain FP64 on a mixed engine
    e.linearize(ps)
    g, cost = e.grad()
    assert abs(cost - float(np.sum(ref_r ** 2))) <= 1e-10 * float(np.sum(ref_r ** 2))
    e.close()


# ---- ragged / edge inputs ----------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 63, 64, 65, 127, 129, 255, 257, 1000])
def test_ragged_sizes(Engine, n):
    rig = synthetic.config_rig(1)
    det = rig.detections[:n].copy()
    det[-1, :3] = [rig.n_cams - 1, rig.n_imgs - 1, rig.n_keys - 1]
    for chain in CHAINS:
        ps, ref_r, ref_j = oracle_eval(rig, chain, det)
        e = make_engine(Engine, rig, chain, det=det)
        for variant in (0, 7):
            e.set_option("variant", variant)
            r, j = e.eval(ps)
            assert r.shape == (n, 2) and j.shape == (2 * n, e.P)
            H.assert_resid_close(r, ref_r, det[:, 3:])
            H.assert_jac_close(j, ref_j)
        e.close()


def test_error_paths(Engine):
    rig = synthetic.tiny_rig()
    e = Engine("template", rig.n_cams, rig.n_imgs, rig.n_keys)
    ps = orc.build_param_list(rig.intr, rig.extr, rig.poses)
    with pytest.raises(_capi.PcsError) as ex:   # nothing uploaded yet
        e.eval(ps)
    assert ex.value.code == _capi.PCS_ERR_STATE
    e.set_detections_table(rig.detections)
    with pytest.raises(_capi.PcsError) as ex:   # template missing
        e.eval(ps)
    assert ex.value.code == _capi.PCS_ERR_STATE
    bad = rig.detections.copy()
    bad[3, 0] = rig.n_cams                      # camera index out of range -> refused on the host
    with pytest.raises(_capi.PcsError) as ex:
        e.set_detections_table(bad)
    assert ex.value.code == _capi.PCS_ERR_RANGE
    bad = rig.detections.copy()
    bad[0, 2] = -1
    with pytest.raises(_capi.PcsError):
        e.set_detections_table(bad)
    bad[0, 2] = np.nan                          # not an index at all
    with pytest.raises(_capi.PcsError) as ex:
        e.set_detections_table(bad)
    assert ex.value.code == _capi.PCS_ERR_RANGE
    # a refused table leaves the previous one in place, host-side index tables included
    e.set_template(rig.points)
    r, j = e.eval(ps)
    _, ref_r, ref_j = oracle_eval(rig, "template")
    H.assert_resid_close(r, ref_r, rig.detections[:, 3:])
    H.assert_jac_close(j, ref_j)
    ind, ptr = e.csr_structure(None)
    assert ptr[-1] == 2 * rig.detections.shape[0] * e.P and ind.max() < e.n_params
    with pytest.raises(ValueError):
        e.eval(ps[:-1])
    with pytest.raises(_capi.PcsError):
        e.set_option("no_such_option", 1)
    with pytest.raises(_capi.PcsError):          # empty table: nothing to evaluate
        e.set_detections_table(np.zeros((0, 5)))
        e.eval(ps)
    e.close()
    f = Engine("free", rig.n_cams, 0, rig.n_keys)
    with pytest.raises(_capi.PcsError):
        f.set_template(rig.points)
    f.close()


def test_zero_pose_and_nonfinite_passthrough(Engine):
    """theta < 1e-10 Rodrigues branches (pose 0 is exactly zero by default) and IEEE inf/nan for a
    point on the camera plane, like the reference's numba code (no trap, no clamping)."""
    rig = synthetic.config_rig(1)
    assert np.all(rig.poses[0] == 0)
    sel = rig.detections[:, 1] == 0
    assert sel.sum() > 50
    det = rig.detections[sel].copy()
    det[-1, :3] = [rig.n_cams - 1, 0, rig.n_keys - 1]
    rig1 = synthetic.SyntheticRig("p0", det, rig.intr, rig.extr, rig.poses[:1].copy(), rig.points)
    rig1.poses[0, :3] = [3e-11, -2e-11, 1e-11]  # inside the small-angle branch, not exactly zero
    for chain in ("template", "self"):
        ps, ref_r, ref_j = oracle_eval(rig1, chain)
        e = make_engine(Engine, rig1, chain)
        r, j = e.eval(ps)
        H.assert_resid_close(r, ref_r, det[:, 3:])
        H.assert_jac_close(j, ref_j)
        e.close()
    # put one point exactly on a camera plane: z = 0 -> inf / nan in that detection only
    rig2 = synthetic.tiny_rig(seed=5)
    rig2.extr[:, :3] = 0
    rig2.extr[:, 3:] = 0
    rig2.points[0] = [0.01, 0.02, 0.0]
    ps = orc.build_param_list(rig2.intr, rig2.extr, rig2.points)
    e = make_engine(Engine, rig2, "free")
    r, j = e.eval(ps)
    ref_j, ref_r = orc.full_jac_dense("free", rig2.detections, ps, None, with_resid=True)
    hit = rig2.detections[:, 2] == 0
    assert hit.any() and not np.isfinite(r[hit]).all() and not np.isfinite(ref_r[hit]).all()
    assert np.isfinite(r[~hit]).all()
    assert np.array_equal(np.isfinite(r), np.isfinite(ref_r))
    H.assert_resid_close(r[~hit], ref_r[~hit], rig2.detections[~hit, 3:])
    e.close()


# ---- full-size configs: sampled oracle comparison + size-independent properties ------------------
@pytest.mark.parametrize("number,chain", [(2, "template"), (3, "template"), (4, "self")])
def test_full_size_config(Engine, number, chain, capsys):
    rig = synthetic.config_rig(number)
    N = rig.n_det
    assert N > 1e5
    ps = orc.build_param_list(*H.chain_slabs(rig, chain))
    e = make_engine(Engine, rig, chain)
    r, j = e.eval(ps)
    assert np.isfinite(r).all() and np.isfinite(j).all()
    # (1) oracle on a strided sample plus the head and the ragged tail of the table
    idx = np.unique(np.concatenate([np.arange(0, N, 37), np.arange(200), np.arange(N - 200, N)]))
    tm = rig.points if chain == "template" else None
    counts = (rig.n_cams, rig.n_imgs, rig.n_keys)
    ref_j, ref_r = orc.full_jac_dense(chain, rig.detections[idx], ps, tm, threads=8, with_resid=True, counts=counts)
    P = e.P
    H.assert_resid_close(r[idx], ref_r, rig.detections[idx, 3:])
    H.assert_jac_close(j.reshape(N, 2 * P)[idx].reshape(-1, P), ref_j)
    # (1b) whose rounding is it?  The same formulas in x87 extended precision (64-bit mantissa) on a sub-sample: the kernel and
    # the float64 oracle are each held against that yardstick.  The oracle restates the reference's own arithmetic (z**7 / z**8
    # denominators, per-detection Rodrigues), which loses ~3e-11 on the point columns of the self chain; the kernel's normalised
    # form stays well inside — the 1e-10 gate above is spent on the ORACLE's rounding, not on the kernel's.
    from oracle import ba_oracle_np as onp
    sub = idx[:: max(1, idx.shape[0] // 4000)]
    _, ext_j = onp.evaluate(chain, rig.detections[sub], ps, tm, counts=counts, dtype=np.longdouble)
    pos = np.searchsorted(idx, sub)
    hip_sub = j.reshape(N, 2 * P)[sub].reshape(-1, P)
    orc_sub = ref_j.reshape(-1, 2 * P)[pos].reshape(-1, P)
    rep_hip, rep_orc, rep_gate = H.jac_error_report(hip_sub, ext_j, P), H.jac_error_report(orc_sub, ext_j, P), H.jac_error_report(hip_sub, orc_sub, P)
    with capsys.disabled():
        print(f"\n[config {number}, chain {chain}, {sub.shape[0]} detections] kernel vs extended precision: {H.describe(rep_hip)}"
              f"\n    float64 oracle vs extended precision: {H.describe(rep_orc)}\n    kernel vs oracle (the gate): {H.describe(rep_gate)}")
    assert rep_hip["floored"] <= 0.2 * H.JAC_RTOL, rep_hip      # a margin of 5 against the yardstick (measured 6e-12 .. 9e-12: the conditioning of the entries themselves)
    # (2) permutation equivariance: evaluating a shuffled table gives the shuffled rows, bit for bit
    perm = np.random.default_rng(0).permutation(N)
    e2 = make_engine(Engine, rig, chain, det=rig.detections[perm])
    r2, j2 = e2.eval(ps)
    assert np.array_equal(r2, r[perm])
    assert np.array_equal(j2.reshape(N, 2 * P), j.reshape(N, 2 * P)[perm])
    e2.close()
    # (3) the Jacobian is the derivative of the residual: central difference along a random direction
    rng = np.random.default_rng(1)
    dx = rng.standard_normal(ps.shape[0]) * np.maximum(np.abs(ps), 1e-3)
    # pose 0 is exactly zero: its rotation sits inside the theta < 1e-10 Rodrigues branch, where the
    # reference's function is locally constant in r while its Jacobian is the generator; the handlers
    # keep that pose fixed (template_handler.py:134-137), so the direction leaves it untouched.
    dx[15 * rig.n_cams: 15 * rig.n_cams + 6] = 0
    dx /= np.linalg.norm(dx)
    h = 1e-6
    rp, _ = e.eval(ps + h * dx, want_jac=False)
    rm, _ = e.eval(ps - h * dx, want_jac=False)
    fd = ((rp - rm) / (2 * h)).reshape(-1)
    cols = e.block_param_inds()
    jv = np.einsum("nrp,np->nr", j.reshape(N, 2, P), dx[cols]).reshape(-1)
    assert np.max(np.abs(fd - jv)) <= 2e-5 * max(1.0, np.max(np.abs(jv)))
    # (4) sharding: evaluating two contiguous halves and concatenating == the full evaluation
    half = N // 2
    ea = make_engine(Engine, rig, chain, det=rig.detections[:half])
    eb = make_engine(Engine, rig, chain, det=rig.detections[half:])
    ja = ea.eval(ps, want_resid=False)[1]
    jb = eb.eval(ps, want_resid=False)[1]
    assert np.array_equal(np.concatenate([ja, jb]), j)
    for x in (e, ea, eb):
        x.close()


def test_device_resident_outputs_and_kernel_timer(Engine):
    import torch
    rig = synthetic.config_rig(2)
    ps = orc.build_param_list(rig.intr, rig.extr, rig.poses)
    e = make_engine(Engine, rig, "template")
    r_host, j_host = e.eval(ps)
    d_r = torch.zeros((rig.n_det, 2), dtype=torch.float64, device="cuda")
    d_j = torch.zeros((2 * rig.n_det, 21), dtype=torch.float64, device="cuda")
    d_p = torch.from_numpy(ps).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    e.eval_device_resident(d_p.data_ptr(), d_r.data_ptr(), d_j.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_r.cpu().numpy(), r_host) and np.array_equal(d_j.cpu().numpy(), j_host)
    d_j.zero_()
    e.eval_device(ps, None, d_j.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_j.cpu().numpy(), j_host)
    prep_ms, eval_ms = e.last_kernel_ms()
    assert 0 < eval_ms < 50 and 0 <= prep_ms < 50
    e.close()


# ---- fixed-parameter compaction --------------------------------------------------------
@pytest.mark.parametrize("chain", CHAINS)
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_compaction_kernels_match_masked_dense(Engine, chain, dtype):
    """Both compaction kernels (per-lane stores, tile-coalesced stores) must reproduce
    ``dense[:n][good_mask]`` (afb:627-651) bit for bit, for scattered per-scalar masks, for
    entity-wise masks like the handlers', for all-fixed and all-free rows, on ragged sizes."""
    rig = synthetic.config_rig(1)
    rng = np.random.default_rng(5)
    ps = orc.build_param_list(*H.chain_slabs(rig, chain))
    n_par = ps.shape[0]
    masks = {
        "scattered": rng.random(n_par) > 0.4,
        "all_free": np.ones(n_par, bool),
        "all_fixed": np.zeros(n_par, bool),
        "entity": np.concatenate([np.repeat(rng.random(rig.n_cams) > 0.3, 9), np.repeat(rng.random(rig.n_cams) > 0.3, 6),
                                  np.ones(n_par - 15 * rig.n_cams, bool)]),
    }
    masks["entity"][15 * rig.n_cams: 15 * rig.n_cams + 6] = chain == "free"  # pose 0 fixed (th:134-137)
    for n in (rig.n_det, 1000, 65, 33, 1):
        det = rig.detections[:n].copy()
        det[-1, :3] = [rig.n_cams - 1, rig.n_imgs - 1, rig.n_keys - 1]
        e = make_engine(Engine, rig, chain, dtype=dtype, det=det)
        r_ref, j = e.eval(ps)
        for name, mask in masks.items():
            idx, ptr, m = orc.csr_structure(chain, det, mask)
            assert e.set_unfixed(mask) == int(m.sum())
            gi, gp = e.csr_structure(mask)
            assert np.array_equal(gi, idx) and np.array_equal(gp, ptr)
            for cv in (0, 1):
                e.set_option("compact_variant", cv)
                r, data = e.eval_compact(ps, want_resid=True)
                # the arithmetic is FP64 in every kernel and rounded once at the store: bit-equal for both dtypes
                assert np.array_equal(data, j[m]), (name, n, cv, dtype)
                assert np.array_equal(r, r_ref)
        e.close()


def test_pinned_output_ring(Engine):
    from pycamset_amd.engine import pinned_empty
    a = pinned_empty((5, 3))
    a[:] = 7.0
    assert a.shape == (5, 3) and a.dtype == np.float64 and float(a.sum()) == 105.0
    del a
    rig = synthetic.config_rig(2)
    ps = orc.build_param_list(rig.intr, rig.extr, rig.poses)
    e = make_engine(Engine, rig, "template")
    r0, j0 = e.eval(ps)
    # reference semantics (afb:561: a fresh array per call): an array that is still held is never handed out again ...
    outs = [e.eval(ps, pinned_ring=2) for _ in range(4)]
    addrs = {o[1].ctypes.data for o in outs}
    assert len(addrs) == 4
    for r, j in outs:
        assert np.array_equal(r, r0) and np.array_equal(j, j0)
    kept = outs[0][1]
    view = outs[1][1].reshape(-1)[:100]      # a VIEW keeps its buffer out of circulation too (csr_array holds one)
    kept_addr, view_addr = kept.ctypes.data, outs[1][1].ctypes.data
    snapshot = kept.copy()
    del outs, r, j
    later = []
    for _ in range(6):
        jn = e.eval(ps * 1.001, pinned_ring=2)[1]
        later.append(jn.ctypes.data)
        del jn
    assert kept_addr not in later and view_addr not in later
    assert np.array_equal(kept, snapshot) and np.array_equal(view, snapshot.reshape(-1)[:100])
    # ... and a buffer nobody holds any more is: a loop that drops each Jacobian before asking for the next one (scipy's
    # least_squares) cycles through the ring's two page-locked buffers
    assert len(set(later[2:])) <= 2
    del kept, view
    again = []
    for _ in range(6):
        jn = e.eval(ps, pinned_ring=2)[1]
        again.append(jn.ctypes.data)
        del jn
    assert len(set(again[2:])) <= 2
    # holders that are not NumPy arrays keep the lease too (round 5: leases, not reference counts): a torch tensor sharing the
    # memory, a memoryview, a csr_array built on a reshape
    import torch
    from scipy.sparse import csr_array as _csr
    j1 = e.eval(ps, pinned_ring=2)[1]
    a1, t1 = j1.ctypes.data, torch.from_numpy(j1)
    j2 = e.eval(ps, pinned_ring=2)[1]
    a2, m2 = j2.ctypes.data, memoryview(j2)
    j3 = e.eval(ps, pinned_ring=2)[1]
    a3 = j3.ctypes.data
    idx3, ptr3 = e.csr_structure()                # the whole data array, like the handlers' jac_fn (a csr_array built on a SMALL slice of a
    c3 = _csr((j3.reshape(-1), idx3, ptr3), shape=(2 * rig.n_det, ps.shape[0]))   # large array copies it: scipy prunes such views)
    del j1, j2, j3
    held = []
    for _ in range(5):
        jn = e.eval(ps * 1.002, pinned_ring=2)[1]
        held.append(jn.ctypes.data)
        del jn
    assert not ({a1, a2, a3} & set(held))
    assert np.array_equal(t1.numpy(), j0) and np.array_equal(np.asarray(m2), j0) and np.array_equal(c3.data, j0.reshape(-1))
    assert c3.data.ctypes.data == a3              # the csr_array wraps the page-locked block itself
    del t1, m2, c3
    # the compacted data array goes through the same ring
    nnz = e.set_unfixed(np.arange(ps.shape[0]) % 3 != 0)
    _, d0 = e.eval_compact(ps)
    ds = [e.eval_compact(ps, pinned_ring=3)[1] for _ in range(3)]
    assert all(d.shape == (nnz,) and np.array_equal(d, d0) for d in ds) and len({d.ctypes.data for d in ds}) == 3
    e.close()
    assert np.array_equal(ds[0], d0)  # page-locked outputs outlive the engine


def test_slabs_too_large_for_lds_fall_back_to_l2(Engine):
    """Config 5 shape (128 cameras x 500 poses): the FP64 slabs (49 KB + 160 KB) exceed the 160 KiB
    LDS, so a forced LDS-staging variant must fall back to reading them through L1/L2 — same bits."""
    rig = synthetic.config_rig(5, scale=0.02)
    assert (rig.n_cams, rig.n_imgs) == (128, 500) and rig.n_det > 1e5
    ps = orc.build_param_list(rig.intr, rig.extr, rig.poses)
    idx = np.unique(np.concatenate([np.arange(0, rig.n_det, 53), np.arange(rig.n_det - 100, rig.n_det)]))
    ref_j, ref_r = orc.full_jac_dense("template", rig.detections[idx], ps, rig.points, threads=8, with_resid=True,
                                      counts=(rig.n_cams, rig.n_imgs, rig.n_keys))
    shuffled = rig.detections[np.random.default_rng(2).permutation(rig.n_det)]
    for dtype in ("f64", "f32"):
        e = make_engine(Engine, rig, "template", dtype=dtype)
        outs = []
        for variant in (6, 7, -1):
            e.set_option("variant", variant)
            r, j = e.eval(ps)
            outs.append(j)
            if dtype == "f64":
                H.assert_resid_close(r[idx], ref_r, rig.detections[idx, 3:])
                H.assert_jac_close(j.reshape(rig.n_det, 42)[idx].reshape(-1, 21), ref_j)
            else:
                assert H.jac_rel_err(j.reshape(rig.n_det, 42)[idx].reshape(-1, 21), ref_j) <= H.F32_JAC_RTOL
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
        e.close()
        # scattered table: the automatic choice stages slabs in LDS when they fit (f32) or falls back (f64)
        e = make_engine(Engine, rig, "template", dtype=dtype, det=shuffled)
        r, j = e.eval(ps)
        inv = np.empty(rig.n_det, dtype=np.int64)
        perm = np.random.default_rng(2).permutation(rig.n_det)
        inv[perm] = np.arange(rig.n_det)
        assert np.array_equal(j.reshape(rig.n_det, 42)[inv], outs[0].reshape(rig.n_det, 42))
        e.close()


def test_randomised_shapes_orders_and_masks(Engine):
    """40 seeded random problems: odd counts (1 camera, 1 image, keys not a multiple of 4), random row
    order, random kernel variant / geometry, random fixed-parameter masks — HIP vs oracle."""
    rng = np.random.default_rng(2024)
    for trial in range(40):
        chain = CHAINS[trial % 3]
        n_cams, n_imgs, n_keys = int(rng.integers(1, 6)), int(rng.integers(1, 8)), int(rng.integers(1, 40))
        rig = synthetic.tiny_rig(seed=trial, n_cams=n_cams, n_imgs=n_imgs, n_keys=n_keys, visibility=float(rng.uniform(0.4, 1.0)))
        det = rig.detections
        if trial % 2:
            det = det[rng.permutation(det.shape[0])]
        if trial % 5 == 0:   # repeat rows: N is then not tied to the rig's size
            det = np.concatenate([det] * int(rng.integers(2, 9)))
        ps = orc.build_param_list(*H.chain_slabs(rig, chain))
        tm = rig.points if chain == "template" else None
        counts = (rig.n_cams, rig.n_imgs, rig.n_keys)
        ref_j, ref_r = orc.full_jac_dense(chain, det, ps, tm, with_resid=True, counts=counts)
        e = make_engine(Engine, rig, chain, det=det)
        e.set_option("variant", int(rng.integers(-1, 8)))
        e.set_option("wgs_per_cu", int(rng.integers(0, 20)))
        e.set_option("waves_per_wg", int(rng.choice([0, 1, 2, 4])))
        e.set_option("pack_indices", int(rng.integers(0, 2)))     # takes effect at the next upload
        e.set_option("xcd_remap", int(rng.integers(0, 2)))
        r, j = e.eval(ps)
        H.assert_resid_close(r, ref_r, det[:, 3:])
        H.assert_jac_close(j, ref_j)
        mask = rng.random(ps.shape[0]) > rng.uniform(0, 0.9)
        cols = e.block_param_inds()
        keep = np.repeat(mask[cols], 2, axis=0)
        assert e.set_unfixed(mask) == int(keep.sum())
        _, data = e.eval_compact(ps)
        assert np.array_equal(data, j[keep]), trial
        # matrix-free products against the dense block rows
        v = rng.standard_normal(ps.shape[0])
        e.linearize(ps)
        jv = np.einsum("nrp,np->nr", j.reshape(-1, 2, e.P), v[cols]).reshape(-1)
        assert np.max(np.abs(e.jv(v) - jv)) <= 1e-10 * max(1.0, np.max(np.abs(jv)))
        jtjv = np.zeros(ps.shape[0])
        np.add.at(jtjv, np.repeat(cols, 2, axis=0).reshape(-1), (j * jv[:, None]).reshape(-1))
        assert np.max(np.abs(e.jtjv(v) - jtjv)) <= 1e-10 * max(1.0, np.max(np.abs(jtjv)))
        e.close()


@pytest.mark.parametrize("chain", CHAINS)
def test_extreme_inputs_propagate_like_ieee_python(Engine, chain):
    """Points behind / on the camera plane, huge distortion, NaN and inf parameters: the kernel must
    produce non-finite values exactly where the CPU oracle does (no clamping, no trap) and agree
    within tolerance everywhere else."""
    rig = synthetic.tiny_rig(seed=9, n_cams=3, n_imgs=4, n_keys=12, visibility=1.0)
    rng = np.random.default_rng(9)
    cases = []
    a = synthetic.SyntheticRig("behind", rig.detections, rig.intr.copy(), rig.extr.copy(), rig.poses.copy(), rig.points.copy())
    a.extr[1, 3:] = [0, 0, -0.2]                       # camera 1 looks away: z < 0
    cases.append(a)
    b = synthetic.SyntheticRig("distorted", rig.detections, rig.intr.copy(), rig.extr.copy(), rig.poses.copy(), rig.points.copy())
    b.intr[:, 4:] = rng.normal(0, 50.0, (rig.n_cams, 5))   # absurd distortion
    cases.append(b)
    c = synthetic.SyntheticRig("nan", rig.detections, rig.intr.copy(), rig.extr.copy(), rig.poses.copy(), rig.points.copy())
    c.intr[0, 5] = np.nan                              # k1 of camera 0
    c.extr[2, 4] = np.nan                              # t_y of camera 2
    c.poses[2, 1] = np.inf                             # rotation of image 2
    # (non-finite or zero FOCAL LENGTHS are the one documented difference: the reference multiplies by fx
    #  and divides by it again, fbi:32-35, so fx = 0 / NaN poisons both residual rows there)
    cases.append(c)
    d = synthetic.SyntheticRig("onplane", rig.detections, rig.intr.copy(), np.zeros_like(rig.extr), np.zeros_like(rig.poses), rig.points.copy())
    d.points[:, 2] = 0.0                               # every point on every camera plane: z = 0
    cases.append(d)
    for case in cases:
        ps = orc.build_param_list(*H.chain_slabs(case, chain))
        tm = case.points if chain == "template" else None
        with np.errstate(all="ignore"):
            ref_j, ref_r = orc.full_jac_dense(chain, case.detections, ps, tm, with_resid=True,
                                              counts=(case.n_cams, case.n_imgs, case.n_keys))
        e = make_engine(Engine, case, chain)
        r, j = e.eval(ps)
        e.close()
        assert np.array_equal(np.isnan(r), np.isnan(ref_r)), case.name
        fin = np.isfinite(ref_r) & np.isfinite(r)
        assert np.array_equal(np.isfinite(r), np.isfinite(ref_r)) or case.name in ("onplane", "nan"), case.name
        if fin.any():
            scale = np.maximum(np.abs(ref_r[fin]), 1e-3 * np.max(np.abs(case.detections[:, 3:])))
            assert np.max(np.abs(r[fin] - ref_r[fin]) / scale) <= 1e-9, case.name
        finj = np.isfinite(ref_j) & np.isfinite(j)
        rows_ok = np.all(np.isfinite(ref_j), axis=1) & np.all(np.isfinite(j), axis=1)
        if rows_ok.any():
            H.assert_jac_close(j[rows_ok], ref_j[rows_ok], rtol=1e-9)
        # a row that is finite in the reference must be finite here and vice versa, except at z = 0 / NaN
        # inputs where inf - inf and 0 * inf orderings legitimately differ between the two formulations
        if case.name in ("behind", "distorted"):
            assert np.array_equal(np.isfinite(j), np.isfinite(ref_j)), case.name


# ---- randomized shapes through every evaluation entry point ------------------------------------------------
@pytest.mark.parametrize("seed", range(6))
def test_random_shapes_through_every_entry_point(Engine, seed):
    """Seeded random rig shapes (1-7 cameras, 1-9 images, 1-40 keys, random visibility, random row order,
    random fixed-parameter mask): dense evaluation, CSR compaction, matrix-free products and the block-reduced
    normal equations must all agree with the oracle's Jacobian of the same table."""
    from scipy.sparse import csr_array
    rng = np.random.default_rng(1234 + seed)
    n_cams, n_imgs, n_keys = int(rng.integers(1, 8)), int(rng.integers(1, 10)), int(rng.integers(1, 41))
    pts = rng.uniform(-0.04, 0.04, (n_keys, 3))
    rig = synthetic.make_rig(f"rand-{seed}", n_cams, n_imgs, pts, seed=500 + seed, visibility=float(rng.uniform(0.3, 1.0)))
    det = rig.detections
    if seed % 2:
        det = det[rng.permutation(det.shape[0])]
    n = det.shape[0]
    for chain in CHAINS:
        ps, ref_r, ref_j = oracle_eval(rig, chain, det)
        npar = ps.shape[0]
        e = make_engine(Engine, rig, chain, det=det)
        r, j = e.eval(ps)
        H.assert_resid_close(r, ref_r, det[:, 3:])
        H.assert_jac_close(j, ref_j)
        # compaction under a random mask (at least one free parameter)
        mask = rng.random(npar) < 0.7
        mask[int(rng.integers(npar))] = True
        idx, ptr, _ = orc.csr_structure(chain, det, mask)
        gi, gp = e.csr_structure(mask)
        assert np.array_equal(gp, ptr) and np.array_equal(gi, idx)
        e.set_unfixed(mask)
        _, data = e.eval_compact(ps)
        full_idx, full_ptr, _ = orc.csr_structure(chain, det, np.ones(npar, bool))
        keep = mask[full_idx]
        ref_data = ref_j.reshape(-1)[keep]
        assert data.shape == ref_data.shape
        scale = np.maximum(np.abs(ref_data), 1e-6 * np.repeat(np.max(np.abs(ref_j), axis=1), ref_j.shape[1])[keep])
        assert np.max(np.abs(data - ref_data) / scale) <= H.JAC_RTOL
        # products and normal equations
        J = csr_array((ref_j.reshape(-1), full_idx, full_ptr), shape=(2 * n, npar))
        v = rng.standard_normal(npar)
        e.linearize(ps)
        ref = J.T @ (J @ v)
        assert np.max(np.abs(e.jtjv(v) - ref)) <= 1e-10 * max(np.max(np.abs(ref)), 1e-300)
        Hm, g, cost = e.normal_equations(ps)
        H_ref = (J.T @ J).toarray()
        sc = np.sqrt(np.outer(np.diag(H_ref), np.diag(H_ref)))
        assert np.max(np.abs(Hm - H_ref) / np.where(sc > 0, sc, 1.0)) <= 1e-10
        g_ref = J.T @ ref_r.reshape(-1)
        assert np.max(np.abs(g - g_ref)) <= 1e-10 * max(np.max(np.abs(g_ref)), 1e-300)
        assert abs(cost - float(np.sum(ref_r ** 2))) <= 1e-10 * float(np.sum(ref_r ** 2))
        e.close()


def test_engine_lifecycle_releases_device_memory(Engine):
    """Create / use / destroy engines repeatedly (every scratch buffer exercised: dense + compact outputs,
    matrix-free vectors, normal equations, legacy cost tables, the sorted visiting order of a scattered table);
    the device memory in use must return to where it started, and two live engines must not disturb each other."""
    import torch
    rig = synthetic.config_rig(1)
    shuffled = rig.detections[np.random.default_rng(0).permutation(rig.n_det)]
    ps_t, ref_r, ref_j = oracle_eval(rig, "template")

    def exercise(det):
        e = make_engine(Engine, rig, "template", det=det)
        e.eval(ps_t)
        mask = np.ones(e.n_params, bool)
        mask[:9] = False
        e.set_unfixed(mask)
        e.eval_compact(ps_t, want_resid=True)
        e.linearize(ps_t)
        e.jtjv(np.ones(e.n_params))
        e.normal_equations(ps_t)
        return e

    torch.cuda.synchronize()
    exercise(rig.detections).close()          # first use: one-off allocations of the runtime itself
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for it in range(20):
        exercise(shuffled if it % 2 else rig.detections).close()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, f"{(free0 - free1) / 2**20:.1f} MiB not returned after 20 create/destroy cycles"
    # two engines alive at once, different chains, interleaved calls
    a = make_engine(Engine, rig, "template")
    b = make_engine(Engine, rig, "self")
    ps_s, ref_rs, ref_js = oracle_eval(rig, "self")
    for _ in range(3):
        ra, ja = a.eval(ps_t)
        rb, jb = b.eval(ps_s)
        H.assert_jac_close(ja, ref_j)
        H.assert_jac_close(jb, ref_js)
    # a table of another size on a live engine: capacities are re-derived
    half = rig.detections[: rig.n_det // 2].copy()
    half[-1, :3] = [rig.n_cams - 1, rig.n_imgs - 1, rig.n_keys - 1]
    a.set_detections_table(half)
    a.set_template(rig.points)
    _, r_half, j_half = oracle_eval(rig, "template", half)
    r2, j2 = a.eval(ps_t)
    H.assert_resid_close(r2, r_half, half[:, 3:])
    H.assert_jac_close(j2, j_half)
    a.set_detections_table(rig.detections)
    r3, j3 = a.eval(ps_t)
    H.assert_jac_close(j3, ref_j)
    a.close()
    b.close()


def test_scattered_table_whose_slabs_exceed_lds_falls_back_to_l2(Engine):
    """128 cameras x 500 images: 216 KB of FP64 slabs, more than the 160 KiB of LDS.  A shuffled table asks for
    LDS-staged slabs (tile locality < 0.5); the launch must fall back to reading them through L1/L2 and still
    match the oracle (dense, compaction, normal equations through the sorted visiting order)."""
    from scipy.sparse import csr_array
    rig = synthetic.config_rig(5, scale=0.002)        # ~6e4 detections, full-size slabs
    assert rig.n_cams == 128 and rig.n_imgs == 500 and 2e4 < rig.n_det < 2e5
    det = rig.detections[np.random.default_rng(3).permutation(rig.n_det)][:30000].copy()
    det[-1, :3] = [rig.n_cams - 1, rig.n_imgs - 1, rig.n_keys - 1]
    ps, ref_r, ref_j = oracle_eval(rig, "template", det)
    e = make_engine(Engine, rig, "template", det=det)
    for variant in (-1, 7, 3):                       # automatic choice and explicit requests for LDS slabs
        e.set_option("variant", variant)
        r, j = e.eval(ps)
        H.assert_resid_close(r, ref_r, det[:, 3:])
        H.assert_jac_close(j, ref_j)
    e.set_option("variant", -1)
    idx, ptr, _ = orc.csr_structure("template", det, np.ones(ps.shape[0], bool))
    J = csr_array((ref_j.reshape(-1), idx, ptr), shape=(2 * det.shape[0], ps.shape[0]))
    Hm, g, cost = e.normal_equations(ps)
    H_ref = (J.T @ J).toarray()
    sc = np.sqrt(np.outer(np.diag(H_ref), np.diag(H_ref)))
    assert np.max(np.abs(Hm - H_ref) / np.where(sc > 0, sc, 1.0)) <= 1e-10
    e.close()


def test_soa_upload_engine_buffers_and_timing_options(Engine):
    """The entry points the other tests do not touch: pcs_set_detections (separate index / measurement arrays),
    pcs_device_buffers (engine-owned output scratch), the HIP-event ring behind pcs_kernel_ms_mean and the
    timing_every option."""
    import torch
    rig = synthetic.config_rig(1)
    det = rig.detections
    ps, ref_r, ref_j = oracle_eval(rig, "template")
    e = Engine("template", rig.n_cams, rig.n_imgs, rig.n_keys)
    e.set_detections(det[:, 0].astype(np.int32), det[:, 1].astype(np.int32), det[:, 2].astype(np.int32), det[:, 3:])
    e.set_template(rig.points)
    r, j = e.eval(ps)
    H.assert_resid_close(r, ref_r, det[:, 3:])
    H.assert_jac_close(j, ref_j)
    with pytest.raises(ValueError):
        e.set_detections(det[:, 0].astype(np.int32), det[:5, 1].astype(np.int32), det[:, 2].astype(np.int32), det[:, 3:])
    # engine-owned device scratch: the pointers pcs_eval itself writes to
    d_r, d_j = e.device_buffers()
    assert d_r and d_j
    e.eval_device(ps * (1 + 1e-4), d_r, d_j)      # overwrite with another point ...
    e.synchronize()
    r2, j2 = e.eval(ps)                           # ... and the host path still returns the right one
    assert np.array_equal(r2, r) and np.array_equal(j2, j)
    # event ring: the last R timed evaluations are averaged; timing_every = k times every k-th evaluation only
    e.set_option("event_ring", 4)
    e.set_option("timing_every", 1)
    for _ in range(6):
        e.eval_device(ps, d_r, d_j)
    e.synchronize()
    cnt, prep_ms, eval_ms = e.kernel_ms_mean()
    assert cnt == 4 and 0 < eval_ms < 5 and prep_ms == 0.0        # a small table: ONE launch per step, no slab_prep kernel
    e.set_option("fuse_prep", 0)                                  # slab_prep + evaluation: both kernels carry their own events
    for _ in range(4):
        e.eval_device(ps, d_r, d_j)
    e.synchronize()
    cnt, prep_ms, eval_ms = e.kernel_ms_mean()
    assert cnt == 4 and 0 < eval_ms < 5 and 0 < prep_ms < 5
    e.set_option("fuse_prep", -1)
    e.set_option("event_ring", 8)
    e.set_option("timing_every", 3)
    for _ in range(7):                            # evaluations 0, 3, 6 are timed
        e.eval_device(ps, d_r, d_j)
    e.synchronize()
    cnt, _, eval_ms2 = e.kernel_ms_mean()
    assert cnt == 3 and 0 < eval_ms2 < 5
    e.set_option("timing_every", 0)               # never: the ring keeps what it has
    e.eval_device(ps, d_r, d_j)
    e.synchronize()
    assert e.kernel_ms_mean()[0] == 3
    with pytest.raises(_capi.PcsError):
        e.set_option("event_ring", 0)
    e.close()


def test_config5_jacobian_streamed_to_host(Engine):
    """BASELINE config 5's mode ("block-sparse Jacobian streamed to host"): pycamset_amd.host_stream.JacobianHostStreamer, the
    class behind bench.py --stream-to-host.  Three consecutive steps with three different parameter strings, TWO device
    buffers and THREE page-locked host buffers, no host synchronisation in between: every host buffer must hold exactly
    the Jacobian of its own step — in particular the first one, whose device buffer the third step reuses while copy 1
    (~1 ms) may still be reading it: the step has to wait for that copy (`free` event), not overwrite it."""
    import torch
    from pycamset_amd.host_stream import JacobianHostStreamer
    rig = synthetic.config_rig(5, scale=0.02)
    assert rig.n_det > 1e5
    e = make_engine(Engine, rig, "template", dtype="f32")
    ps = orc.build_param_list(rig.intr, rig.extr, rig.poses)
    rng = np.random.default_rng(8)
    strings = [ps * (1.0 + 1e-3 * rng.standard_normal(ps.shape[0])) for _ in range(3)]
    d_ps = [torch.from_numpy(p).cuda() for p in strings]
    N = rig.n_det
    d_r = torch.empty((N, 2), dtype=torch.float32, device="cuda")
    # reference Jacobians, one at a time, each into its own device buffer
    refs = []
    for dp in d_ps:
        dj = torch.empty((2 * N, e.P), dtype=torch.float32, device="cuda")
        e.eval_device_resident(dp.data_ptr(), d_r.data_ptr(), dj.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        refs.append(dj.cpu().numpy())
    assert not np.array_equal(refs[0], refs[1]) and not np.array_equal(refs[0], refs[2])
    st = JacobianHostStreamer(e, N, n_device_buffers=2, n_host_buffers=3)
    for h in st.host:
        h.fill_(-7.0)
    slots = [st.step(dp.data_ptr(), d_r.data_ptr()) for dp in d_ps]          # queued back to back
    assert slots == [0, 1, 2]
    for k, ref in zip(slots, refs):
        got = st.wait(k).numpy()
        assert np.array_equal(got, ref), f"host buffer {k} does not hold its step's Jacobian"
    # a host buffer handed out by wait() belongs to its consumer: a step that would copy into it is refused until release()
    with pytest.raises(RuntimeError, match="still with its consumer"):
        st.step(d_ps[0].data_ptr(), d_r.data_ptr())
    kept = st.host[0].numpy().copy()
    for k in slots:
        st.release(k)
    # a second round through the same ring: host buffers are reused once their consumer is done with them
    slots2 = [st.step(dp.data_ptr(), d_r.data_ptr()) for dp in reversed(d_ps)]
    for k, ref in zip(slots2, reversed(refs)):
        assert np.array_equal(st.wait(k).numpy(), ref)
    assert np.array_equal(kept, refs[0])
    torch.cuda.synchronize()
    e.close()


@pytest.mark.parametrize("dtype", ["f32", "mixed"])
def test_config5_at_full_size(Engine, dtype):
    """BASELINE config 5 at its stated size (128 cameras x 500 poses x 486 keys, ~1e7 detections, FP32 bytes) — outputs stay
    on the device (1.7 GB of Jacobian).  Checks: (1) oracle on a strided sample plus head and ragged tail, at the dtype's
    tolerance; (2) permutation equivariance, bit for bit; (3) two contiguous shards concatenate to the full evaluation,
    bit for bit (the sharded 8-GPU form of the config is exactly this, per rank)."""
    import torch
    rig = synthetic.config_rig(5)
    N, P = rig.n_det, 21
    assert N > 9.5e6 and (rig.n_cams, rig.n_imgs, rig.n_keys) == (128, 500, 486)
    ps = orc.build_param_list(rig.intr, rig.extr, rig.poses)

    def run(det):
        e = make_engine(Engine, rig, "template", dtype=dtype, det=det)
        n = det.shape[0]
        d_r = torch.empty((n, 2), dtype=torch.float32, device="cuda")
        d_j = torch.empty((n, 2 * P), dtype=torch.float32, device="cuda")
        e.eval_device(ps, d_r.data_ptr(), d_j.data_ptr())
        e.synchronize()
        e.close()
        return d_r, d_j

    d_r, d_j = run(rig.detections)
    assert bool(torch.isfinite(d_j).all()) and bool(torch.isfinite(d_r).all())
    idx = np.unique(np.concatenate([np.arange(0, N, 997), np.arange(300), np.arange(N - 300, N)]))
    ref_j, ref_r = orc.full_jac_dense("template", rig.detections[idx], ps, rig.points, threads=8, with_resid=True,
                                      counts=(rig.n_cams, rig.n_imgs, rig.n_keys))
    t_idx = torch.from_numpy(idx).cuda()
    j = d_j[t_idx].cpu().numpy().astype(np.float64).reshape(-1, P)
    r = d_r[t_idx].cpu().numpy().astype(np.float64)
    assert H.jac_rel_err(j, ref_j) <= (H.F32_JAC_RTOL if dtype == "f32" else H.MIXED_JAC_RTOL)
    if dtype == "f32":
        assert np.max(np.abs(r - ref_r)) <= H.F32_RES_ATOL
    else:
        assert np.max(np.abs(r - ref_r) / np.maximum(np.abs(ref_r), 1e-3)) <= H.MIXED_RES_RTOL
    perm = np.random.default_rng(5).permutation(N)
    p_r, p_j = run(rig.detections[perm])
    t_perm = torch.from_numpy(perm).cuda()
    assert bool(torch.equal(p_j, d_j[t_perm])) and bool(torch.equal(p_r, d_r[t_perm]))
    del p_r, p_j
    cut = (N // 2 // 64) * 64 + 17            # not a tile boundary
    a_r, a_j = run(rig.detections[:cut])
    b_r, b_j = run(rig.detections[cut:])
    assert bool(torch.equal(torch.cat([a_j, b_j]), d_j)) and bool(torch.equal(torch.cat([a_r, b_r]), d_r))
