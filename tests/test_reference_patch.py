"""The INTEGRATION.md "two-line patch", applied to the reference's REAL handlers (build container only).

The reference's ``TemplateBundleHandler`` / ``SelfBundleHandler`` / ``FreePointBundleHandler`` are imported from a
temporary copy of /root/reference (tests/golden/_refload.py), their ``self.op_fun`` (template_handler.py:152,
standard_bundle_handler.py:182, free_point_handler.py:143) is replaced by the adapter of
``pycamset_amd.function_blocks`` and the reference's own ``make_loss_fun`` / ``make_loss_jac`` (th:157-193) are driven:

  1. with the real engine: every attribute the reference touches exists on the adapter and the only failure in this
     GPU-less container is ``PcsError(PCS_ERR_NODEVICE)`` out of ``pcs_create`` (there is no CPU fallback to hide it);
  2. with a test-only stand-in for ``Engine`` (arithmetic by the CPU oracle — test infrastructure, never the product):
     the closures the REFERENCE builds around the adapter return what the unpatched reference returns.

Skipped where /root/reference does not exist (the GPU box).
"""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
import _refload  # noqa: E402

from oracle import ba_oracle as orc  # noqa: E402
from pycamset_amd import _capi, synthetic  # noqa: E402
from pycamset_amd import function_blocks as hip_fb  # noqa: E402
from tests.test_host_logic import DuckCamset, DuckTargetND  # noqa: E402
from tests.test_oracle_golden import assert_close  # noqa: E402

pytestmark = pytest.mark.skipif(not _refload.available(), reason="needs /root/reference (build container only)")


def hip_chain(chain):
    if chain == "template":
        return hip_fb.projection() + hip_fb.extrinsic3D() + hip_fb.template_points()
    if chain == "self":
        return hip_fb.projection() + hip_fb.extrinsic3D() + hip_fb.rigidTform3d() + hip_fb.free_point()
    return hip_fb.projection() + hip_fb.extrinsic3D() + hip_fb.free_point()


class OracleEngine:
    """TEST-ONLY stand-in for pycamset_amd.engine.Engine (same methods the adapter calls), CPU oracle inside."""

    def __init__(self, chain, n_cams, n_imgs, n_keys, *, dtype="f64", device=0):
        self.chain, self.counts = chain, (int(n_cams), int(n_imgs), int(n_keys))
        self.P = orc.CHAIN_P[chain]
        self.n_keys = int(n_keys)
        self.mask_key = self.nnz = self.template = None

    def set_detections_table(self, det):
        self.det = np.array(det, dtype=np.float64)
        self.n = self.det.shape[0]
        self.n_params = int(orc.param_struct(self.chain, self.det, self.counts)[2])

    def set_template(self, t):
        self.template = np.array(t, dtype=np.float64).reshape(-1, 3)

    def eval(self, ps, want_resid=True, want_jac=True, pinned_ring=0):
        assert ps.shape[0] == self.n_params
        r = orc.full_loss(self.chain, self.det, ps, self.template, counts=self.counts) if want_resid else None
        j = orc.full_jac_dense(self.chain, self.det, ps, self.template, counts=self.counts) if want_jac else None
        return r, j

    def csr_structure(self, unfixed):
        return orc.csr_structure(self.chain, self.det, unfixed, self.counts)[:2]

    def set_unfixed(self, unfixed):
        self.mask = np.asarray(unfixed, dtype=bool)
        self.mask_key = hash(self.mask.tobytes())
        self.keep = orc.csr_structure(self.chain, self.det, self.mask, self.counts)[2]
        self.nnz = int(self.keep.sum())
        return self.nnz

    def eval_compact(self, ps, want_resid=False, pinned_ring=0):
        return None, self.eval(ps, False, True)[1][self.keep]

    def block_param_inds(self):
        return orc.block_param_inds(self.chain, self.det, self.counts)


def _problem(mods, chain, rig, keydims, max_ims=0):
    from tests.golden.make_golden import to_multidim_keys  # the fixture generator's own helper
    names = [f"cam_{i}" for i in range(rig.n_cams)]
    det = mods.TargetDetection(cam_names=names, data=to_multidim_keys(rig.detections, keydims), max_ims=max_ims)
    fixed = {"cam_1": {"int": rig.intr[1].copy(), "ext": rig.extr[1].copy()}}
    mods.th.DEFAULT_OPTIONS.update({"fixed_pose": 0})
    cls = {"template": mods.th.TemplateBundleHandler, "self": mods.sbh.SelfBundleHandler, "free": mods.fph.FreePointBundleHandler}[chain]
    h = cls(DuckCamset(rig.n_cams), DuckTargetND(rig.points, keydims), det, fixed_params=fixed, options={"verbosity": 0})
    bp = h.bundlePrimitive
    poses = np.zeros((det.max_ims, 6))
    poses[: rig.n_imgs] = rig.poses
    parts = [rig.intr[bp.intr_unfixed].ravel(), rig.extr[bp.extr_unfixed].ravel()]
    if chain != "free":
        parts.append(poses[bp.poses_unfixed].ravel())
    if chain != "template":
        parts.append(rig.points.ravel()[bp.bdpt_unfixed])
    return h, np.concatenate(parts)


@pytest.fixture(scope="module")
def mods():
    with _refload.reference_modules() as m:
        yield m


@pytest.mark.parametrize("chain", ["template", "self", "free"])
def test_patched_reference_handler_reaches_pcs_create(mods, chain):
    rig = synthetic.make_rig("patch", 3, 4, synthetic.ccube_points(4, 40.0), seed=5, visibility=0.7)   # 6 x 9 keys
    h, x = _problem(mods, chain, rig, (6, 9))
    h.op_fun = hip_chain(chain)                      # <- the patch
    assert h.can_make_jac() is True                  # th:154-155
    if _capi.lib().pcs_device_count() > 0:
        pytest.skip("a GPU is visible: the closure tests in test_gpu_dropin.py cover the live path")
    for make in (h.make_loss_fun, h.make_loss_jac):
        with pytest.raises(_capi.PcsError) as err:
            make(4)
        assert err.value.code == _capi.PCS_ERR_NODEVICE, "every step up to pcs_create must succeed on the host"


@pytest.mark.parametrize("chain,max_ims", [("template", 0), ("self", 0), ("free", 0), ("template", 6), ("self", 6)])
def test_patched_reference_handler_returns_what_the_reference_returns(mods, chain, max_ims, monkeypatch):
    """max_ims = 6 > 4 images: the handler's pose slab (and mask) is longer than the string the detections index —
    the reference reads only the leading entries (afb:363) and, in the self chain, mis-offsets the point block
    (quirk ii); the patched handler must do exactly the same."""
    rig = synthetic.make_rig("patch", 3, 4, synthetic.ccube_points(4, 40.0), seed=5, visibility=0.7)
    h_ref, x = _problem(mods, chain, rig, (6, 9), max_ims)
    r_ref = np.array(h_ref.make_loss_fun(3)(x.copy()))
    J_ref = h_ref.make_loss_jac(3)(x.copy())
    h, x2 = _problem(mods, chain, rig, (6, 9), max_ims)
    assert np.array_equal(x, x2)
    monkeypatch.setattr(hip_fb, "Engine", OracleEngine)      # test-only: no GPU in the build container
    h.op_fun = hip_chain(chain)                               # <- the patch
    r = h.make_loss_fun(3)(x.copy())
    J = h.make_loss_jac(3)(x.copy())
    assert r.shape == r_ref.shape and J.shape == J_ref.shape
    assert np.array_equal(J.indices, J_ref.indices) and np.array_equal(J.indptr, J_ref.indptr)
    assert_close(r, r_ref, rtol=1e-11, rows=1e3 * np.repeat(np.max(np.abs(rig.detections[:, 3:]), axis=1), 2))
    dense_rows = np.repeat(np.max(np.abs(J_ref).toarray(), axis=1), np.diff(J_ref.indptr))
    assert_close(J.data, J_ref.data, rows=dense_rows)
