"""Host logic of the drop-in boundary (no GPU): free-vector <-> slab scatter, fixed-parameter masks,
key flattening, synthetic rig layout.  The handler goldens come from the reference's own handlers
(tests/golden/make_golden.py); the arithmetic here is done by the CPU oracle."""
from pathlib import Path

import numpy as np
import pytest

from oracle import ba_oracle as orc
from pycamset_amd import handlers, synthetic
from pycamset_amd.detections import TargetDetection
from tests import helpers as H
from tests.test_oracle_golden import assert_close


class DuckCamset:
    def __init__(self, n):
        self._names = [f"cam_{i}" for i in range(n)]

    def get_names(self):
        return list(self._names)

    def get_n_cams(self):
        return len(self._names)


class DuckTarget:
    def __init__(self, points):
        self.point_data = np.array(points, dtype=np.float64)[None]


HANDLERS = {"template": handlers.TemplateBundleHandler, "self": handlers.SelfBundleHandler,
            "free": handlers.FreePointBundleHandler}


def make_handler(g, chain, fixed):
    n_cams = g["intr0"].shape[0]
    det = TargetDetection([f"cam_{i}" for i in range(n_cams)], g["detections"])
    fp = None
    if fixed:
        fp = {"cam_0": {"ext": g["fixed_ext_cam0"].copy()}, "cam_1": {"int": g["fixed_int_cam1"].copy()}}
    return HANDLERS[chain](DuckCamset(n_cams), DuckTarget(g["points"]), det, fixed_params=fp, options={"verbosity": 0})


@pytest.mark.parametrize("chain", ["template", "self", "free"])
@pytest.mark.parametrize("tag,fixed", [("tiny", True), ("tiny_nofix", False)])
def test_handler_host_logic_against_reference_goldens(golden_dir, chain, tag, fixed):
    g = np.load(golden_dir / f"handler_{chain}_{tag}.npz")
    h = make_handler(g, chain, fixed)
    bp = h.bundlePrimitive
    assert np.array_equal(bp.intr_unfixed, g["intr_unfixed"]) and np.array_equal(bp.extr_unfixed, g["extr_unfixed"])
    if chain != "free":
        assert np.array_equal(bp.poses_unfixed, g["poses_unfixed"])
        assert not bp.poses_unfixed[0]  # fixed_pose = 0 (template_handler.py:134-137)
    if chain != "template":
        assert np.array_equal(bp.bdpt_unfixed, g["bdpt_unfixed"])
    x = g["x"]
    assert x.shape[0] == g["shape"][1]
    slabs = h.get_bundle_adjustment_inputs(x.copy())
    param_str = h.op_fun.build_param_list(*slabs)
    mask = h._jac_mask()
    assert mask.shape[0] == param_str.shape[0] and int(mask.sum()) == x.shape[0]
    det = h._flat_detections()
    tmpl = h._template_arg()
    # structure (integer work) is exact; values go through the oracle
    idx, ptr, m = orc.csr_structure(chain, det, mask)
    assert np.array_equal(idx, g["indices"]) and np.array_equal(ptr, g["indptr"])
    res = orc.full_loss(chain, det, param_str, tmpl).flatten()
    assert_close(res, g["resid"], rtol=1e-11)
    dense = orc.full_jac_dense(chain, det, param_str, tmpl)
    rows = np.broadcast_to(np.max(np.abs(dense), axis=1, keepdims=True), dense.shape)
    assert_close(dense[m], g["data"], rows=rows[m])


class DuckTargetND:
    """Ccube-shaped target: point_data (6, 81, 3) (target_Ccube.py:227-244)."""

    def __init__(self, points, keydims):
        self.point_data = np.array(points, dtype=np.float64).reshape(tuple(int(k) for k in keydims) + (3,))


# round-2 fixtures (tests/golden/make_golden.py round2_vectors): name -> (chain, options)
R2_CASES = {
    "handler_template_ccube": ("template", {"fixed_pose": 3}),
    "handler_self_ccube": ("self", {"fixed_pose": 3}),
    "handler_template_fixedpose_none": ("template", {"fixed_pose": None}),
    "handler_free_ccube": ("free", {}),
    "quirk_template_last_image_unobserved": ("template", {}),
    "quirk_self_last_key_unobserved": ("self", {}),
    "quirk_self_last_image_unobserved": ("self", {}),
    "quirk_template_last_cam_unobserved": ("template", {}),
}
R2_SAME_AS_REFERENCE = [n for n in R2_CASES if n not in ("quirk_self_last_image_unobserved", "quirk_template_last_cam_unobserved")]


def make_handler_r2(g, name, **kw):
    """Rebuild the handler of a round-2 fixture from the inputs stored beside the reference's outputs."""
    chain, options = R2_CASES[name]
    n_cams = int(g["n_cams"])
    names = [f"cam_{i}" for i in range(n_cams)]
    fixed = {}
    for c in range(n_cams):
        if not g["intr_unfixed"][c]:
            fixed.setdefault(names[c], {})["int"] = g["intr_slab"][c].copy()
        if not g["extr_unfixed"][c]:
            fixed.setdefault(names[c], {})["ext"] = g["extr_slab"][c].copy()
    det = TargetDetection(names, g["detections"], max_ims=int(g["max_ims"]))
    keydims = tuple(int(k) for k in g["keydims"])
    target = DuckTargetND(g["points"], keydims)
    opts = {"verbosity": 0}
    opts.update(options)
    return HANDLERS[chain](DuckCamset(n_cams), target, det, fixed_params=fixed or None, options=opts, **kw), chain


def test_return_flattened_keys_against_the_reference(golden_dir):
    """TargetDetection.return_flattened_keys (target_detections.py:333-351), fixtures made by the
    reference on 2-, 3- and 1-dimensional key tables."""
    g = np.load(golden_dir / "flatten_keys.npz")
    for tag in ("ccube", "three_dim", "one_dim"):
        td = TargetDetection(["cam_0", "cam_1", "cam_2"], g[f"{tag}_in"], max_ims=7)
        flat = td.return_flattened_keys(tuple(g[f"{tag}_dims"]))
        assert np.array_equal(flat.get_data(), g[f"{tag}_out"]), tag       # integer-valued keys: bit-exact
        assert flat.max_ims == int(g[f"{tag}_max_ims"]) == 7
    # the flattened key indexes point_data.reshape(-1, 3) (th:160-163)
    dims = tuple(g["ccube_dims"])
    pts = np.arange(int(np.prod(dims)) * 3, dtype=np.float64).reshape(dims + (3,))
    a, b = g["ccube_in"], g["ccube_out"]
    assert np.array_equal(pts.reshape(-1, 3)[b[:, 2].astype(int)], pts[a[:, 2].astype(int), a[:, 3].astype(int)])


@pytest.mark.parametrize("name", R2_SAME_AS_REFERENCE)
def test_ccube_shaped_and_edge_handlers_against_reference_goldens(golden_dir, name):
    """Handler host logic (masks, x -> slabs, CSR structure) on Ccube-shaped targets (6, 81, 3) with multi-dimensional
    keys, fixed_pose = 3 / None, a camera with 'int' AND 'ext' fixed, and trailing unobserved images / keys.
    Values go through the CPU oracle."""
    g = np.load(golden_dir / f"{name}.npz")
    h, chain = make_handler_r2(g, name)
    bp = h.bundlePrimitive
    assert np.array_equal(bp.intr_unfixed, g["intr_unfixed"]) and np.array_equal(bp.extr_unfixed, g["extr_unfixed"])
    if chain != "free":
        assert np.array_equal(bp.poses_unfixed, g["poses_unfixed"])
    if chain != "template":
        assert np.array_equal(bp.bdpt_unfixed, g["bdpt_unfixed"])
    if chain == "self":
        assert np.array_equal(h.visible_feature_mask, g["visible_feature_mask"])
    if name == "handler_template_fixedpose_none":
        assert not bp.poses_unfixed.any()          # a None index fixes every pose (th:134-137)
    x = g["x"]
    slabs = h.get_bundle_adjustment_inputs(x.copy())
    assert np.array_equal(slabs[0], g["intr_slab"]) and np.array_equal(slabs[1], g["extr_slab"])
    if chain != "free":
        assert np.array_equal(slabs[2], g["poses_slab"])
    param_str = h.op_fun.build_param_list(*slabs)
    mask = h._jac_mask()
    assert int(mask.sum()) == x.shape[0] == g["shape"][1]
    det = h._flat_detections()
    assert det.shape[1] == 5
    counts = h.op_fun.counts
    n_imgs = bp.poses.shape[0] if chain != "free" else counts[1]   # the free chain has no pose group: any image count lays out the same string
    assert counts == (bp.intr.shape[0], n_imgs, int(np.prod(g["keydims"])))
    assert orc.param_struct(chain, det, counts)[2] == param_str.shape[0] == mask.shape[0]
    idx, ptr, m = orc.csr_structure(chain, det, mask, counts)
    assert np.array_equal(idx, g["indices"]) and np.array_equal(ptr, g["indptr"])
    tmpl = h._template_arg()
    res = orc.full_loss(chain, det, param_str, tmpl, counts=counts).flatten()
    assert_close(res, g["resid"], rtol=1e-11)
    dense = orc.full_jac_dense(chain, det, param_str, tmpl, counts=counts)
    rows = np.broadcast_to(np.max(np.abs(dense), axis=1, keepdims=True), dense.shape)
    assert_close(dense[m], g["data"], rows=rows[m])


@pytest.mark.parametrize("name", ["quirk_self_last_image_unobserved", "quirk_template_last_cam_unobserved"])
def test_reference_quirk_ii_is_understood_and_not_reproduced(golden_dir, name):
    """abstract_function_blocks.py:793-795 sizes the parameter groups from max(index) + 1 of the detections while the
    handlers size their slabs from n_cams / max_ims / point_data (th:124-129).  With a trailing unobserved camera (any
    chain) or image (self chain) the reference therefore reads later groups at the wrong offset.  (1) The fixture IS
    what that rule gives: the oracle, laid out with max(index) + 1 and fed the handler's longer string, reproduces it.
    (2) This package lays the string out from the slab sizes instead and evaluates the slabs the handler built."""
    g = np.load(golden_dir / f"{name}.npz")
    h, chain = make_handler_r2(g, name)
    x = g["x"]
    param_str = h.op_fun.build_param_list(*h.get_bundle_adjustment_inputs(x.copy()))
    mask, det, tmpl = h._jac_mask(), h._flat_detections(), h._template_arg()
    ref_counts = orc.counts_from_detections(det)
    own_counts = h.op_fun.counts
    assert ref_counts != own_counts and orc.param_struct(chain, det, own_counts)[2] == param_str.shape[0]
    # (1) the reference's rule on the handler's string
    n_ref = orc.param_struct(chain, det, ref_counts)[2]
    assert n_ref < param_str.shape[0]
    res = orc.full_loss(chain, det, param_str[:n_ref], tmpl, counts=ref_counts).flatten()
    assert_close(res, g["resid"], rtol=1e-11)
    idx, ptr, m = orc.csr_structure(chain, det, mask, ref_counts)     # columns by the detections' rule, mask by the handler's
    assert np.array_equal(idx, g["indices"]) and np.array_equal(ptr, g["indptr"])
    dense = orc.full_jac_dense(chain, det, param_str[:n_ref], tmpl, counts=ref_counts)
    rows = np.broadcast_to(np.max(np.abs(dense), axis=1, keepdims=True), dense.shape)
    assert_close(dense[m], g["data"], rows=rows[m])
    # (2) the slab-size layout gives a different (the intended) function: residuals stay at the noise level
    own = orc.full_loss(chain, det, param_str, tmpl, counts=own_counts).flatten()
    assert np.median(np.abs(own)) < 20 and np.max(np.abs(own - g["resid"])) > 1.0
    idx2, ptr2, _ = orc.csr_structure(chain, det, mask, own_counts)
    assert idx2.max() < x.shape[0] and ptr2[-1] == idx2.shape[0]


def test_fill_flat_scatter_and_layout():
    rng = np.random.default_rng(0)
    full = np.zeros((5, 6))
    unf = np.array([True, False, True, True, False])
    src = rng.random((3, 6))
    handlers.fill_flat(src, full, unf)
    assert np.array_equal(full[unf], src) and np.all(full[~unf] == 0)
    flat = np.zeros(7)
    handlers.fill_flat(np.array([1.0, 2.0]), flat, np.array([0, 1, 0, 0, 1, 0, 0], bool))
    assert flat.tolist() == [0, 1, 0, 0, 2, 0, 0]


def test_return_flattened_keys_matches_row_major_reshape():
    # Ccube-style keys (face, index) flatten like point_data.reshape(-1, 3) (target_detections.py:333-351)
    data = np.array([[0, 0, 2, 5, 10.0, 20.0], [1, 3, 5, 80, 1.0, 2.0], [1, 3, 0, 0, 3.0, 4.0]])
    td = TargetDetection(["a", "b"], data)
    flat = td.return_flattened_keys((6, 81)).get_data()
    assert flat.shape == (3, 5)
    assert flat[:, 2].tolist() == [2 * 81 + 5, 5 * 81 + 80, 0]
    assert np.array_equal(flat[:, 3:], data[:, 4:])
    assert td.max_ims == 4
    td5 = TargetDetection(["a", "b"], flat)
    assert td5.return_flattened_keys((6, 81)) is td5
    with pytest.raises(ValueError):
        TargetDetection(["a", "a"], flat)


def test_unsupported_chain_raises():
    """Compositions outside `projection + {rigidTform3d | extrinsic3D}* + (template_points | free_point)` have no kernel and
    no interpreter: they raise when used (a + b + c builds partial chains first, afb:735-748, so not at construction)."""
    from pycamset_amd import function_blocks as fb
    for op in (fb.projection() + fb.rigidTform3d(),                                   # no point source
               fb.rigidTform3d() + fb.free_point(),                                   # no projection
               fb.projection() + fb.free_point() + fb.rigidTform3d(),                 # source not last
               fb.projection() + fb.projection() + fb.free_point()):                  # a 3 -> 2 block in the middle
        with pytest.raises(NotImplementedError):
            op.make_full_loss_fn(np.zeros((1, 5)), 1)
    assert (fb.projection() + fb.extrinsic3D() + fb.template_points()).chain == "template"
    assert (fb.projection() + fb.rigidTform3d() + fb.free_point()).chain == "generated"


@pytest.mark.parametrize("tag", ["proj_rigid_free", "proj_extr_rigid_template", "proj_template", "proj_rigid_extr_free"])
def test_chain_compiler_layout_and_structure_match_the_reference_generator(golden_dir, tag):
    """Chains that are NOT one of the handlers' three, run through the reference's own code generator (make_golden.py
    generic_block_level): the parameter-string layout (make_param_struct, afb:777-820), get_block_param_inds (afb:192-233)
    and the CSR structure (afb:465-489) of pycamset_amd.chain_compiler must reproduce them bit for bit — integer work, no GPU.
    The emitted translation unit must also compile for gfx950 (hipcc cross-compiles here)."""
    from pycamset_amd import function_blocks as fb
    from pycamset_amd import chain_compiler as cc
    g = np.load(golden_dir / f"generic_{tag}.npz")
    names = [str(n) for n in g["blocks"]]
    spec = cc.ChainSpec.from_blocks([getattr(fb, n)() for n in names])
    det = g["detections"]
    C, I, K = (int(det[:, j].max()) + 1 for j in range(3))
    lay = spec.layout(C, I, K)
    assert lay["n_params"] == g["param_str"].shape[0]
    cols = cc.block_param_inds(spec, lay, det[:, :3].astype(np.int64))
    assert cols.shape[1] == spec.P and np.array_equal(cols, g["block_param_inds"])
    idx, ptr, keep, off = cc.csr_structure_of(cols, lay["n_params"], None)
    assert np.array_equal(idx, g["indices_all"]) and np.array_equal(ptr, g["indptr_all"])
    assert np.all(keep == (1 << spec.P) - 1) and np.array_equal(off, 2 * spec.P * np.arange(det.shape[0]))
    idx, ptr, keep, off = cc.csr_structure_of(cols, lay["n_params"], g["unfixed"])
    assert np.array_equal(idx, g["indices_masked"]) and np.array_equal(ptr, g["indptr_masked"])
    # the per-detection tables the device packs with (csrc/ba_generic.hpp generic_compact_body) describe data[:n][good_mask] (afb:644-651)
    _check_compaction_tables(g["data_all"], g["data_masked"], keep, off, spec.P)
    src_text = cc.emit_source(spec)
    assert f"static constexpr int P = {spec.P};" in src_text and "PCS_GENCHAIN_ENTRY_POINTS" in src_text
    obj = cc.compile_chain(spec)
    assert obj.exists() and obj.stat().st_size > 10_000
    # blocks of one class share ONE parameter group, like the reference's object-identity rule (afb:160-163)
    twice = cc.ChainSpec.from_blocks([fb.projection(), fb.rigidTform3d(), fb.rigidTform3d(), fb.free_point()])
    assert twice.n_rigid_groups == 1 and [b.slab for b in twice.blocks] == [None, 0, 0, None] and twice.layout(2, 3, 4)["n_params"] == 18 + 18 + 12


def _check_compaction_tables(data_all, data_masked, keep, off, P):
    """Pack the dense block rows exactly like the device does — detection i's kept u entries at off[i], its v entries behind them."""
    dense = np.asarray(data_all).reshape(-1, 2, P)
    out = np.full(data_masked.shape[0], np.nan)
    for i in range(dense.shape[0]):
        sel = [j for j in range(P) if (int(keep[i]) >> j) & 1]
        out[int(off[i]): int(off[i]) + len(sel)] = dense[i, 0, sel]
        out[int(off[i]) + len(sel): int(off[i]) + 2 * len(sel)] = dense[i, 1, sel]
    assert np.array_equal(out, data_masked)


@pytest.mark.parametrize("tag", ["user_cam_scale", "user_division"])
def test_user_blocks_layout_structure_and_code_generation(golden_dir, tag):
    """The reference's extension point (afb:689-775): chains that contain USER-written blocks, run through the reference's own
    code generator (make_golden.py --only round4; the blocks are tests/golden/_user_blocks.py).  The GPU counterpart declares
    the same blocks as device code (function_blocks.device_function_block): parameter-string layout, block_param_inds and both
    CSR structures must reproduce the reference bit for bit, and the generated translation unit must compile for gfx950."""
    from pycamset_amd import function_blocks as fb
    from pycamset_amd import chain_compiler as cc
    g = np.load(golden_dir / f"{tag}.npz")
    ub = H.user_blocks(fb)
    names = [str(n) for n in g["blocks"]]
    blocks = [ub[n]() if n in ub else getattr(fb, n)() for n in names]
    spec = cc.ChainSpec.from_blocks(blocks)
    assert [b.kind for b in spec.blocks].count("user") == 1
    det = g["detections"]
    C, I, K = (int(det[:, j].max()) + 1 for j in range(3))
    lay = spec.layout(C, I, K)
    assert lay["n_params"] == g["param_str"].shape[0] and spec.P == g["block_param_inds"].shape[1]
    cols = cc.block_param_inds(spec, lay, det[:, :3].astype(np.int64))
    assert np.array_equal(cols, g["block_param_inds"])
    idx, ptr, _, _ = cc.csr_structure_of(cols, lay["n_params"], None)
    assert np.array_equal(idx, g["indices_all"]) and np.array_equal(ptr, g["indptr_all"])
    idx, ptr, keep, off = cc.csr_structure_of(cols, lay["n_params"], g["unfixed"])
    assert np.array_equal(idx, g["indices_masked"]) and np.array_equal(ptr, g["indptr_masked"])
    _check_compaction_tables(g["data_all"], g["data_masked"], keep, off, spec.P)
    text = cc.emit_source(spec)
    assert "namespace user" in text and "pcs::chain_user<" in text
    # what the one-launch kernels need to prepare the slabs themselves: one slab per rigid parameter group, and whose transform it holds
    n_rigid = sum(1 for gr in spec.groups if gr["kind"] == "rigid")
    assert f"static constexpr int N_SLABS = {n_rigid};" in text and "slab_link(const int g)" in text
    obj = cc.compile_chain(spec)
    assert obj.stat().st_size > 10_000
    blob = obj.read_bytes()
    for entry in (b"pcs_genchain_prep", b"pcs_genchain_eval_3", b"pcs_genchain_eval_3_one", b"pcs_genchain_compact_2_f32", b"pcs_genchain_compact_2_f32_one"):
        assert entry in blob, entry   # both launch forms of every kernel are in the code object
    # what the composition rules refuse: a block that is neither shipped nor a device block; neighbours that do not fit
    class plain(fb.abstract_function_block):
        num_inp, num_out = 3, 3
        params = fb.param_type(fb.key_type.PER_CAM, 1)
    for bad in ([fb.projection(), plain(), fb.free_point()], [fb.projection(), ub["division_projection"](), fb.free_point()],
                [ub["cam_scale"](), fb.extrinsic3D(), fb.free_point()]):
        with pytest.raises(NotImplementedError):
            cc.ChainSpec.from_blocks(bad)


def test_initial_params_must_be_supplied():
    rig = synthetic.tiny_rig()
    det = TargetDetection([f"cam_{i}" for i in range(rig.n_cams)], rig.detections)
    h = handlers.TemplateBundleHandler(DuckCamset(rig.n_cams), DuckTarget(rig.points), det)
    with pytest.raises(NotImplementedError):
        h.get_initial_params()
    h.set_initial_params(np.ones(3))
    assert h.get_initial_params().shape == (3,)


@pytest.mark.parametrize("number,expect", [(1, (3, 24, 486)), (2, (8, 50, 256))])
def test_synthetic_config_shapes(number, expect):
    rig = synthetic.config_rig(number)
    assert (rig.n_cams, rig.n_imgs, rig.n_keys) == expect
    d = rig.detections
    if number == 2:
        assert d.shape == (102400, 5)
    # ordered cam -> image -> key (camera_calibrator.py:314-317) and covering the last cam / image / key
    order = np.lexsort((d[:, 2], d[:, 1], d[:, 0]))
    assert np.array_equal(order, np.arange(d.shape[0]))
    assert orc.counts_from_detections(d) == expect
    assert np.all(rig.poses[0] == 0)
    # measurements sit within a few pixels of the projection at the evaluation point
    r = orc.full_loss("template", d, orc.build_param_list(rig.intr, rig.extr, rig.poses), rig.points, threads=4, fast=True)
    assert np.isfinite(r).all() and np.median(np.abs(r)) < 50


def test_synthetic_headline_scaled_down():
    rig = synthetic.config_rig(3, scale=0.01)
    assert (rig.n_cams, rig.n_imgs, rig.n_keys) == (32, 200, 486)
    assert 5000 < rig.n_det < 20000
    rig_im = synthetic.config_rig(3, scale=0.01, order="im")
    assert rig_im.n_det == rig.n_det
    assert np.all(np.diff(rig_im.detections[:, 1]) >= 0)


def test_pcg_solves_spd_system():
    from pycamset_amd.device_solver import pcg
    rng = np.random.default_rng(0)
    B = rng.standard_normal((40, 25))
    A = B.T @ B + 0.1 * np.eye(25)
    b = rng.standard_normal(25)
    x, its = pcg(lambda v: A @ v, b, 1.0 / np.diag(A), 1e-12, 200)
    assert its < 100 and np.max(np.abs(A @ x - b)) <= 1e-8 * np.max(np.abs(b))


def test_device_lm_driver_logic_on_cpu_operator():
    """The LM driver (damping, acceptance, stopping) with a CPU operator built from the oracle's
    Jacobian injected in place of the HIP engine: it must reduce the cost monotonically to the
    noise floor, like scipy on the same closures."""
    from scipy.optimize import least_squares
    from scipy.sparse import csr_array
    from pycamset_amd.device_solver import JacobianOperator, lm_solve

    rig = synthetic.make_rig("ring-4", 4, 6, synthetic.charuco_points(7, 8.0), seed=31, visibility=0.9)
    names = [f"cam_{i}" for i in range(rig.n_cams)]
    h = handlers.TemplateBundleHandler(DuckCamset(rig.n_cams), DuckTarget(rig.points), TargetDetection(names, rig.detections),
                                       fixed_params={"cam_0": {"ext": rig.extr_true[0].copy()}}, options={"verbosity": 0})
    bp = h.bundlePrimitive
    x0 = np.concatenate([rig.intr[bp.intr_unfixed].ravel(), rig.extr[bp.extr_unfixed].ravel(), rig.poses[bp.poses_unfixed].ravel()])
    det, mask = h._flat_detections(), h._jac_mask()
    counts = orc.counts_from_detections(det)

    class CpuEngine:
        n, n_params = det.shape[0], mask.shape[0]

        def linearize(self, ps):
            dense, r = orc.full_jac_dense("template", det, ps, rig.points, with_resid=True, counts=counts)
            idx, ptr, _ = orc.csr_structure("template", det, np.ones(mask.shape[0], bool))
            self.J = csr_array((dense.reshape(-1), idx, ptr), shape=(2 * det.shape[0], mask.shape[0]))
            self.r = r.reshape(-1)

        def jv(self, v):
            return self.J @ v

        def jtu(self, u):
            return self.J.T @ u

        def jtjv(self, v):
            return self.J.T @ (self.J @ v)

        def jtj_diag(self):
            return np.asarray(self.J.multiply(self.J).sum(axis=0)).ravel()

        def grad(self):
            return self.J.T @ self.r, float(self.r @ self.r)

    op = JacobianOperator(CpuEngine(), mask)
    res = lm_solve(h, x0.copy(), max_iter=25, operator=op)
    assert res.history == sorted(res.history, reverse=True) and res.nit >= 2

    def loss(x):
        return orc.full_loss("template", det, orc.build_param_list(*h.get_bundle_adjustment_inputs(x)), rig.points).reshape(-1)

    err = np.mean(np.linalg.norm(loss(res.x).reshape(-1, 2), axis=1))
    assert abs(0.5 * np.sum(loss(res.x) ** 2) - res.cost) <= 1e-9 * res.cost
    assert err < 0.6 and res.cost < 0.01 * res.history[0]
    L = op.as_linear_operator()
    assert L.shape == (2 * det.shape[0], x0.shape[0])

    # the same driver with the Cholesky step on explicit normal equations (CPU stand-in for NormalEquations)
    import torch
    from tools.library_solver import cholesky_step

    class CpuNormal:
        free = np.flatnonzero(mask)

        def build(self, ps):
            e = CpuEngine()
            e.linearize(ps)
            Jf = e.J[:, self.free]
            return torch.from_numpy((Jf.T @ Jf).toarray()), torch.from_numpy(Jf.T @ e.r), float(e.r @ e.r)

        solve = staticmethod(cholesky_step)

    # the Schur-complement form of the step equals the dense solve
    from tools.library_solver import schur_cholesky_step, trailing_block_structure
    n_cams, n_imgs, n_keys = counts
    st = trailing_block_structure("template", n_cams, n_imgs, n_keys, mask)
    assert st[:2] == (int(mask[: 15 * n_cams].sum()), 6) and st[2] is None
    Hs, gs, _ = CpuNormal().build(orc.build_param_list(*h.get_bundle_adjustment_inputs(x0)))
    dd = torch.diagonal(Hs).clone()
    for lam in (1e-6, 1e-2, 10.0):
        a, b = cholesky_step(Hs, gs, lam, dd), schur_cholesky_step(Hs, gs, lam, dd, st[0], st[1])
        assert float((a - b).abs().max()) <= 1e-9 * float(a.abs().max())
    # a gauge-style mask that fixes single point coordinates: the partly fixed point joins the leading group
    m2 = np.ones(15 * 3 + 6 * 4 + 3 * 8, bool)
    assert trailing_block_structure("self", 3, 4, 8, m2)[:2] == (15 * 3 + 6 * 4, 3)
    m2[-2] = False                                       # y of the last point fixed
    n_lead, blk, perm = trailing_block_structure("self", 3, 4, 8, m2)
    assert (n_lead, blk) == (15 * 3 + 6 * 4 + 2, 3) and sorted(perm) == list(range(m2.sum()))
    assert list(perm[:69]) == list(range(69)) and list(perm[69:71]) == [90, 91] and list(perm[71:]) == list(range(69, 90))
    assert trailing_block_structure("free", 3, 0, 8, np.ones(15 * 3 + 24, bool))[:2] == (45, 3)
    assert trailing_block_structure("template", 3, 4, 8, np.r_[np.ones(45, bool), np.zeros(24, bool)]) is None

    res2 = lm_solve(h, x0.copy(), max_iter=25, operator=CpuNormal(), linear_solver="cholesky")
    assert res2.history == sorted(res2.history, reverse=True)
    assert abs(res2.cost - res.cost) <= 1e-6 * res.cost and res2.nfev <= res.nfev + 2
    # the initial damping: None = LAM0_EXACT = 1e-5 with the exact step (round 5: one evaluation slower than round 4's 1e-6 from a near
    # start, up to twice as fast from a far one), 1e-3 with PCG; from any value the loop gets to the same cost (this
    # small, weakly determined rig creeps along a flat valley either way: 17-19 evaluations)
    from pycamset_amd.device_solver import LAM0_EXACT
    res3 = lm_solve(h, x0.copy(), max_iter=25, operator=CpuNormal(), linear_solver="cholesky", lam0=1e-3)
    assert abs(res3.cost - res2.cost) <= 1e-6 * res2.cost and abs(res3.nfev - res2.nfev) <= 6
    res4 = lm_solve(h, x0.copy(), max_iter=25, operator=CpuNormal(), linear_solver="cholesky", lam0=LAM0_EXACT)
    assert LAM0_EXACT == 1e-5 and res4.nfev == res2.nfev and res4.history == res2.history
    res5 = lm_solve(h, x0.copy(), max_iter=25, operator=op, lam0=1e-3)
    assert res5.nfev == res.nfev and res5.history == res.history
    with pytest.raises(ValueError):
        lm_solve(h, x0.copy(), operator=op, linear_solver="qr")


def test_normal_equation_entry_maps_own_every_pair_exactly_once():
    """csrc/ba_normal.hpp hard-codes which entry of H / g / cost every MFMA accumulator register stands for (operand
    windows + the keep rule).  Per chain and pass, the owned registers must cover every needed column pair exactly
    once — pcs_normal_entry_map is a host function, so this runs without a GPU."""
    from ctypes import POINTER, c_int32
    from pycamset_amd import _capi
    lib = _capi.lib()
    R = 30

    def owned(chain, p):
        out = np.full((2, 64, 4, 2), -7, dtype=np.int32)
        _capi.check(lib.pcs_normal_entry_map(_capi.CHAIN_IDS[chain], p, out.ctypes.data_as(POINTER(c_int32))))
        pairs = [tuple(sorted(x)) for x in out.reshape(-1, 2).tolist() if x[0] >= 0]
        assert all((a < 0) == (b < 0) for a, b in out.reshape(-1, 2).tolist())
        return pairs

    def upper(cols):
        return sorted((a, b) for i, a in enumerate(cols) for b in cols[i:])

    for chain, n_shared in (("template", 21), ("self", 21), ("free", 15)):
        cols = list(range(n_shared)) + [R]
        assert sorted(owned(chain, 0)) == upper(cols), chain                     # shared pass: whole upper triangle
    for chain, pt0 in (("self", 21), ("free", 15)):
        pts, cam = [pt0, pt0 + 1, pt0 + 2], list(range(15))
        need = sorted([tuple(sorted((c, q))) for c in cam for q in pts] + upper(pts) + [(q, R) for q in pts])
        assert sorted(owned(chain, 1)) == need, chain                            # E[c,k], D[k], g[k]
    # F[i,k] (ba_normal_imgkey_kernel): the accumulators hold, for 16 runs at a time, the symmetric 3 x 3 sum over the
    # pose-translation columns 18..20 that the pose-point block is finished from; register r of lane l belongs to local
    # run (l >> 4) + 4 r, and every run must own each of the 6 entries exactly once
    out = np.full((2, 64, 4, 2), -7, dtype=np.int32)
    _capi.check(lib.pcs_normal_entry_map(_capi.CHAIN_IDS["self"], 2, out.ctypes.data_as(POINTER(c_int32))))
    per_run = {}
    for m in range(2):
        for lane in range(64):
            for r in range(4):
                if out[m, lane, r, 0] >= 0:
                    per_run.setdefault((lane >> 4) + 4 * r, []).append(tuple(out[m, lane, r].tolist()))
    assert sorted(per_run) == list(range(16))
    for pairs in per_run.values():
        assert sorted(pairs) == upper([18, 19, 20])
    for chain, p in (("template", 1), ("template", 2), ("free", 2)):
        assert lib.pcs_normal_entry_map(_capi.CHAIN_IDS[chain], p, np.zeros(1024, np.int32).ctypes.data_as(POINTER(c_int32))) == _capi.PCS_ERR_ARG


def test_normal_equation_flush_descriptors_address_the_right_entries():
    """Round 3: the flush of ba_normal_mfma_kernel finds every destination from a packed per-register descriptor and a
    64-lane table it refreshes per run (entry_descriptor, csrc/ba_normal.hpp) — for the dense layout and for the blocked
    one ([A | B | C]: leading x leading, leading x trailing, block-diagonal trailing group).  Both are host functions here:
    the test rebuilds the table for sample runs, decodes the descriptors exactly like the kernel does (two row / column
    look-ups, the row length, the pointer entry) and checks every owned register against the column pair
    pcs_normal_entry_map reports for it — inside its region, at the right offset."""
    from ctypes import POINTER, c_int32
    from pycamset_amd import _capi
    lib = _capi.lib()
    R = 30
    C, I, K = 5, 7, 11
    for chain in ("template", "self", "free"):
        has_pose = chain != "free"
        n_shared = 21 if has_pose else 15
        extr_off, pose_off = 9 * C, 15 * C
        point_off = 15 * C + 6 * I if chain == "self" else 15 * C
        n_params = 15 * C + (6 * I if has_pose else 0) + (0 if chain == "template" else 3 * K)
        for p in ((0,) if chain == "template" else (0, 1)):
            emap = np.full((2, 64, 4, 2), -7, dtype=np.int32)
            _capi.check(lib.pcs_normal_entry_map(_capi.CHAIN_IDS[chain], p, emap.ctypes.data_as(POINTER(c_int32))))
            for blocked in (False, True):
                tg = (2 if chain == "template" else 3) if blocked else -1
                trail_off = (pose_off if chain == "template" else point_off) if blocked else 0
                tb = (6 if chain == "template" else 3) if blocked else 0
                n_lead = trail_off if blocked else n_params
                n_trail = n_params - n_lead
                ldA, ldB = n_lead, n_trail
                desc = np.zeros((2, 64, 4), dtype=np.int32)
                _capi.check(lib.pcs_normal_descriptors(_capi.CHAIN_IDS[chain], p, tg, desc.ctypes.data_as(POINTER(c_int32))))
                for cam, img, key in ((0, 0, 0), (C - 1, I - 1, K - 1), (2, 3, 4)):
                    base = [9 * cam, extr_off + 6 * cam, pose_off + 6 * img, point_off + 3 * key]
                    tab = np.zeros(64, dtype=np.int64)
                    tab[12], tab[13], tab[14] = ldA, ldB, tb
                    for g in range(4):          # offsets in doubles (round 5; bytes until round 4)
                        tab[g] = ldA * base[g]
                        tab[4 + g] = ldB * base[g]
                        tab[16 + g] = base[g] - (trail_off if g == tg else 0)
                        tab[24 + g] = base[g]
                    tab[8] = tb * tb * (img if tg == 2 else key)

                    def glob(lc):
                        if lc < 9:
                            return base[0] + lc
                        if lc < 15:
                            return base[1] + lc - 9
                        if has_pose and lc < n_shared:
                            return base[2] + lc - 15
                        return base[3] + lc - n_shared

                    seen = set()
                    for m in range(2):
                        for lane in range(64):
                            for r in range(4):
                                la, lb = (int(v) for v in emap[m, lane, r])
                                d = int(desc[m, lane, r])
                                assert ((d >> 27) & 1) == (la >= 0)
                                if la < 0:
                                    continue
                                o_r, o_c = d & 15, (d >> 4) & 15
                                e_row, e_col, e_ld, e_ptr = (d >> 8) & 31, (d >> 13) & 31, (d >> 18) & 31, (d >> 23) & 7
                                off = tab[e_row] + tab[e_col] + o_r * tab[e_ld] + o_c
                                if la == R and lb == R:
                                    want = (4, 0)
                                elif R in (la, lb):
                                    want = (3, glob(lb if la == R else la))
                                else:
                                    ga, gb = sorted((glob(la), glob(lb)))
                                    if gb < n_lead:
                                        want = (0, ga * ldA + gb)
                                    elif ga < n_lead:
                                        want = (1, ga * ldB + gb - n_lead)
                                    else:
                                        e = (ga - n_lead) // tb
                                        assert (gb - n_lead) // tb == e
                                        want = (2, e * tb * tb + ((ga - n_lead) % tb) * tb + (gb - n_lead) % tb)
                                assert (e_ptr, off) == want, (chain, p, blocked, (cam, img, key), m, lane, r, (la, lb), (e_ptr, off), want)
                                limit = [n_lead * n_lead, n_lead * n_trail, (n_trail // tb) * tb * tb if tb else 0, n_params, 1][e_ptr]
                                assert 0 <= off < limit
                                assert (e_ptr, off) not in seen
                                seen.add((e_ptr, off))
                                # entries that involve a pose column are the ones flushed when only the image changes
                                if p == 0:
                                    assert ((d >> 28) & 1) == int(has_pose and any(15 <= x < n_shared for x in (la, lb)))


def test_output_ring_leases_follow_every_kind_of_holder(monkeypatch):
    """Round 5: a block of the page-locked output ring is handed out again only when its LEASE has ended — a
    weakref.finalize on the root array NumPy hangs every derived view on — not when a reference count looks idle.  Holders:
    the array, a slice of a reshape, a csr_array built on it, torch.from_numpy, a memoryview.  (The pinned allocation itself
    needs a GPU; a stand-in block of ordinary memory exercises the same logic.)"""
    import ctypes
    import gc

    import torch
    from scipy.sparse import csr_array

    import pycamset_amd.engine as E

    class FakeBlock:
        def __init__(self, nbytes):
            self.mem = (ctypes.c_char * nbytes)()
            self.ptr = ctypes.c_void_p(ctypes.addressof(self.mem))
            self.nbytes = nbytes

    monkeypatch.setattr(E, "_PinnedBlock", FakeBlock)

    class Owner:
        _rings = {}

    def out(ring=2):
        return E.Engine._out(Owner, "jac", (4, 3), ring)

    a = out()
    a[:] = 1.0
    b = out()
    assert a.ctypes.data != b.ctypes.data
    c = out()                                   # slot 0 is still leased: a new block, never a's
    assert c.ctypes.data not in (a.ctypes.data, b.ctypes.data)
    addr_a = a.ctypes.data
    v = a.reshape(-1)[:5]                       # a view of a view keeps the lease
    del a
    x = out(); ax = x.ctypes.data
    m = csr_array((x.reshape(-1), np.arange(12) % 3, np.arange(0, 13, 3)), shape=(4, 3))
    del x
    t = out(); at = t.ctypes.data
    tt = torch.from_numpy(t)
    del t
    y = out(); ay = y.ctypes.data
    mv = memoryview(y)
    del y
    later = []
    for _ in range(6):
        z = out()
        z[:] = 9.0
        later.append(z.ctypes.data)
        del z
    assert not ({addr_a, ax, at, ay} & set(later))
    assert v[0] == 1.0 and len(set(later[2:])) <= 2          # held blocks untouched; a drop-before-next loop cycles through two blocks
    del m, tt, mv, v, b, c
    gc.collect()
    again = []
    for _ in range(6):
        z = out()
        again.append(z.ctypes.data)
        del z
    assert len(set(again[2:])) <= 2
    assert out(0).flags.owndata or out(0).base is None       # ring 0: plain pageable memory
