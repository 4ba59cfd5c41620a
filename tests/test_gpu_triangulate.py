"""SURVEY f4 on the GPU: batched n-view triangulation through the C ABI against the golden vectors
(reference output) and the NumPy/LAPACK oracle.

Tolerance: |X - X_ref| <= tol * |X_ref| with tol = max(1e-10, 1e3 * eps * sigma_1 / (sigma_{m-1} - sigma_m))
per point: the smallest right singular vector is only determined to eps * sigma_1 / gap by ANY backward
stable algorithm (the reference's LAPACK SVD included), and two-view points near the baseline have
gaps of 1e-4.  Well-conditioned points (10-30 views) agree to ~2e-13."""
import numpy as np
import pytest

from oracle import ba_oracle as orc
from pycamset_amd import _capi, synthetic
from pycamset_amd import compiled_helpers as hip_ch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return float(np.max(np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)))


def svd_tolerances(rec, start, P, K, D):
    """Per-point conditioning bound of the reference's own SVD."""
    tol = np.empty(len(start) - 1)
    for j in range(len(start) - 1):
        rows = rec[start[j]:start[j + 1]]
        n = len(rows)
        M = np.zeros((3 * n, 4 + n))
        for i, r in enumerate(rows):
            c = int(r[0])
            M[3 * i:3 * i + 3, :4] = P[c]
            M[3 * i:3 * i + 2, 4 + i] = -orc.undistort(r[-2:], K[c], D[c])
            M[3 * i + 2, 4 + i] = -1.0
        S = np.linalg.svd(M, compute_uv=False)
        tol[j] = max(1e-10, 1e3 * np.finfo(float).eps * S[0] / (S[-2] - S[-1]))
    return tol


def assert_points_close(pts, ref, tol):
    err = np.linalg.norm(pts - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert np.all(err <= tol), (float(np.max(err / tol)), float(err.max()))


def test_triangulation_matches_reference_goldens(golden_dir):
    g = np.load(golden_dir / "triangulation.npz")
    pts = hip_ch.nb_triangulate_full(g["data"], g["proj"], g["start_inds"], g["intrinsics"], g["dists"])
    assert pts.shape == g["points"].shape
    assert rel_err(pts, g["points"]) <= 1e-10
    assert hip_ch.last_triangulate_kernel_ms > 0


def rig_inputs(rig):
    im, P, K, D = orc.legacy_inputs(rig.intr_true, rig.extr_true, rig.poses_true, rig.points)
    d = rig.detections
    d = d[np.lexsort((d[:, 0], d[:, 2], d[:, 1]))]
    rec, start = hip_ch.group_reconstructable(d)
    return rec, start, P, K, D, im


@pytest.mark.parametrize("n_cams,vis", [(2, 1.0), (5, 0.7), (32, 0.5), (32, 1.0)])
def test_triangulation_matches_svd_oracle(n_cams, vis):
    rig = synthetic.make_rig("tri", n_cams, 4, synthetic.ccube_points(6, 30.0), seed=40 + n_cams, visibility=vis,
                             n_rings=2 if n_cams >= 4 else 1)
    rec, start, P, K, D, im = rig_inputs(rig)
    views = np.diff(start)
    assert views.min() >= 2 and views.max() <= n_cams
    pts = hip_ch.nb_triangulate_full(rec, P, start, K, D)
    ref = orc.triangulate_full(rec, P, start, K, D)
    tol = svd_tolerances(rec, start, P, K, D)
    assert_points_close(pts, ref, tol)
    if n_cams >= 32:
        assert tol.max() == 1e-10 and rel_err(pts, ref) <= 1e-11
    first = rec[start[:-1]]
    truth = im[first[:, 1].astype(int), first[:, 2].astype(int)]
    assert np.median(np.linalg.norm(pts - truth, axis=1)) < 5e-4      # 0.3 px noise at 0.2 m
    # distortion switched off (multi_cam_triangulate(distort=False), camera_set.py:384-385)
    pts0 = hip_ch.nb_triangulate_full(rec, P, start, K, np.zeros_like(D))
    assert_points_close(pts0, orc.triangulate_full(rec, P, start, K, np.zeros_like(D)), 10 * tol)


def test_triangulation_at_scale_and_errors():
    rig = synthetic.make_rig("tri-big", 32, 40, synthetic.ccube_points(), seed=77, visibility=0.3215, n_rings=2)
    rec, start, P, K, D, im = rig_inputs(rig)
    n_pts = start.shape[0] - 1
    assert n_pts > 15000 and rec.shape[0] > 1.5e5
    pts = hip_ch.nb_triangulate_full(rec, P, start, K, D)
    ms = hip_ch.last_triangulate_kernel_ms
    sel = np.arange(0, n_pts, 41)
    sub_rows = np.concatenate([np.arange(start[j], start[j + 1]) for j in sel])
    sub_start = np.append(0, np.cumsum(np.diff(start)[sel]))
    ref = orc.triangulate_full(rec[sub_rows], P, sub_start, K, D)
    assert rel_err(pts[sel], ref) <= 1e-10
    assert np.isfinite(pts).all() and 0 < ms < 50
    # permuting the points permutes the output (each lane owns one point)
    perm = np.random.default_rng(0).permutation(n_pts)
    rows = np.concatenate([np.arange(start[j], start[j + 1]) for j in perm])
    pstart = np.append(0, np.cumsum(np.diff(start)[perm]))
    assert np.array_equal(hip_ch.nb_triangulate_full(rec[rows], P, pstart, K, D), pts[perm])
    # argument checks
    with pytest.raises(_capi.PcsError):
        hip_ch.nb_triangulate_full(rec, P, start[:-1], K, D)           # start_inds does not end at n_obs
    bad = rec.copy()
    bad[5, 0] = 99
    with pytest.raises(_capi.PcsError) as e:
        hip_ch.nb_triangulate_full(bad, P, start, K, D)
    assert e.value.code == _capi.PCS_ERR_RANGE
    assert hip_ch.nb_triangulate_full(rec[:0], P, np.array([0]), K, D).shape == (0, 3)


def test_triangulator_handle_resident_inputs_and_repeated_calls():
    """The handle API (pcs_tri_*): device-resident inputs / outputs give the bits of the host path, the handle survives
    problems of different sizes and camera sets, and misuse fails with a status code instead of reading garbage."""
    import torch
    rig = synthetic.make_rig("tri-h", 6, 5, synthetic.charuco_points(7, 6.0), seed=4, visibility=0.7, n_rings=2)
    rec, start, P, K, D, _ = rig_inputs(rig)
    ref = hip_ch.nb_triangulate_full(rec, P, start, K, D)
    tri = hip_ch.Triangulator(rig.n_cams)
    with pytest.raises(_capi.PcsError) as ex:          # nothing set yet
        tri.run()
    assert ex.value.code == _capi.PCS_ERR_STATE
    tri.set_cameras(P, K, D)
    with pytest.raises(_capi.PcsError):
        tri.run()                                      # observations missing
    cam = rec[:, 0].astype(np.int32)
    tri.set_observations(cam, rec[:, -2:], start)
    tri.run()
    assert np.array_equal(tri.points(), ref)
    # device-resident: caller-owned tensors in, caller-owned tensor out, on torch's current stream
    d_cam, d_uv = torch.from_numpy(cam).cuda(), torch.from_numpy(np.ascontiguousarray(rec[:, -2:])).cuda()
    d_st = torch.from_numpy(np.ascontiguousarray(start, dtype=np.int64)).cuda()
    d_pts = torch.zeros((len(start) - 1, 3), dtype=torch.float64, device="cuda")
    tri.set_observations_device(cam.shape[0], d_cam.data_ptr(), d_uv.data_ptr(), len(start) - 1, d_st.data_ptr())
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        tri.run(d_pts.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_pts.cpu().numpy(), ref) and tri.last_kernel_ms() > 0
    # a smaller problem on the same handle (buffers are reused), then the first one again
    half = len(start) // 2
    tri.set_observations(cam[: start[half]], rec[: start[half], -2:], start[: half + 1])
    tri.run()
    assert np.array_equal(tri.points(), ref[:half])
    tri.set_observations(cam, rec[:, -2:], start)
    tri.run()
    assert np.array_equal(tri.points(), ref)
    bad = cam.copy()
    bad[3] = rig.n_cams
    with pytest.raises(_capi.PcsError) as ex:
        tri.set_observations(bad, rec[:, -2:], start)
    assert ex.value.code == _capi.PCS_ERR_RANGE
    with pytest.raises(ValueError):
        tri.set_cameras(P[:-1], K[:-1], D[:-1])
    tri.close()


def test_triangulator_orders_runs_across_streams():
    """Round 3 (advisor finding): the handle records an event after every run.  A run queued on a CALLER stream with the
    handle-owned output followed by points() — which copies on the handle's stream — must return that run's points; points()
    after a run that wrote to a caller buffer must refuse (PCS_ERR_STATE) instead of returning an older result; setters called
    while a run may still be in flight on another stream wait for it."""
    import torch
    rig = synthetic.make_rig("tri-s", 12, 30, synthetic.ccube_points(), seed=9, visibility=0.4, n_rings=2)
    rec, start, P, K, D, _ = rig_inputs(rig)
    ref = hip_ch.nb_triangulate_full(rec, P, start, K, D)
    cam = rec[:, 0].astype(np.int32)
    tri = hip_ch.Triangulator(rig.n_cams)
    tri.set_cameras(P, K, D)
    tri.set_observations(cam, rec[:, -2:], start)
    side = torch.cuda.Stream()
    for _ in range(5):
        tri.run(None, side.cuda_stream)                 # handle-owned output, caller stream, no synchronisation by the caller
        assert np.array_equal(tri.points(), ref)
    d_pts = torch.zeros((len(start) - 1, 3), dtype=torch.float64, device="cuda")
    tri.run(d_pts.data_ptr(), side.cuda_stream)         # the result goes to the caller's buffer ...
    with pytest.raises(_capi.PcsError) as ex:
        tri.points()                                    # ... so there is no handle-owned result of THIS run
    assert ex.value.code == _capi.PCS_ERR_STATE
    side.synchronize()
    assert np.array_equal(d_pts.cpu().numpy(), ref)
    # a new problem right behind a run on the side stream: the setter waits for that run before it overwrites the inputs
    half = len(start) // 2
    tri.run(None, side.cuda_stream)
    tri.set_observations(cam[: start[half]], rec[: start[half], -2:], start[: half + 1])
    with pytest.raises(_capi.PcsError):
        tri.points()                                    # results of the earlier problem are not this problem's
    tri.run(None, side.cuda_stream)
    assert np.array_equal(tri.points(), ref[:half])
    tri.close()



@pytest.mark.parametrize("n_cams,vis", [(5, 0.45), (32, 0.3215)])
def test_multi_cam_triangulate_groups_on_the_device(n_cams, vis):
    """The front end of row f4 (cameras/camera_set.py:343-402, the array branch): which features are seen by more than one camera, their
    rows in table order, start indices in order of first appearance — np.unique twice in the reference, four small launches here
    (pcs_tri_group_device) — and then the triangulation.  Against the host grouping (group_reconstructable, itself pinned to the
    reference's in tests/test_oracle_golden.py) followed by nb_triangulate_full: the same points bit for bit, single-view features
    dropped, features with NO detection skipped, an empty table, and a table that is not grouped by feature (host fallback)."""
    import torch
    rig = synthetic.make_rig("tri-group", n_cams, 6, synthetic.ccube_points(6, 30.0), seed=70 + n_cams, visibility=vis, n_rings=2 if n_cams >= 4 else 1)
    _, P, K, D = orc.legacy_inputs(rig.intr_true, rig.extr_true, rig.poses_true, rig.points)
    d = rig.detections
    d = d[np.lexsort((d[:, 0], d[:, 2], d[:, 1]))]                       # grouped by (image, key), cameras inside: TargetDetection.get_data's order
    rec, start = hip_ch.group_reconstructable(d)
    counts = np.unique(d[:, 1:3], axis=0, return_counts=True)[1]
    assert start.shape[0] - 1 == int((counts >= 2).sum())
    if n_cams == 5:
        assert (counts == 1).any()                                       # some features are seen once and must be dropped
    ref = hip_ch.nb_triangulate_full(rec, P, start, K, D)
    pts = hip_ch.multi_cam_triangulate(d, P, K, D)
    assert pts.shape == ref.shape and np.array_equal(pts, ref)
    # the pieces: counts, kept rows and start indices of the device grouping
    tri = hip_ch.Triangulator(n_cams)
    tri.set_cameras(P, K, D)
    feat = (d[:, 1].astype(np.int64) * rig.n_keys + d[:, 2].astype(np.int64)).astype(np.int32)
    d_cam, d_feat, d_uv = (torch.from_numpy(x).cuda() for x in (d[:, 0].astype(np.int32), feat, np.ascontiguousarray(d[:, 3:])))
    torch.cuda.synchronize()
    n_pts, n_kept, grouped = tri.group_table_device(d.shape[0], d_cam.data_ptr(), d_feat.data_ptr(), d_uv.data_ptr(), rig.n_imgs * rig.n_keys)
    assert grouped and (n_pts, n_kept) == (start.shape[0] - 1, rec.shape[0])
    tri.run()
    assert np.array_equal(tri.points(), ref)
    # distortion off, like multi_cam_triangulate(distort=False)
    assert np.array_equal(hip_ch.multi_cam_triangulate(d, P, K, D, distort=False), hip_ch.nb_triangulate_full(rec, P, start, K, np.zeros_like(D)))
    # a table that is NOT grouped by feature (cam -> image -> key order): the device says so, the host grouping takes over — and that is
    # the reference's own (consecutive-slice) semantics for such a table
    dd = rig.detections                                                   # cam-major
    feat2 = (dd[:, 1].astype(np.int64) * rig.n_keys + dd[:, 2].astype(np.int64)).astype(np.int32)
    t_cam, t_feat, t_uv = (torch.from_numpy(x).cuda() for x in (dd[:, 0].astype(np.int32), feat2, np.ascontiguousarray(dd[:, 3:])))
    torch.cuda.synchronize()
    assert tri.group_table_device(dd.shape[0], t_cam.data_ptr(), t_feat.data_ptr(), t_uv.data_ptr(), rig.n_imgs * rig.n_keys)[2] is False
    rec2, start2 = hip_ch.group_reconstructable(dd)
    assert np.array_equal(hip_ch.multi_cam_triangulate(dd, P, K, D), hip_ch.nb_triangulate_full(rec2, P, start2, K, D))
    # nothing to reconstruct
    assert hip_ch.multi_cam_triangulate(d[:0], P, K, D).shape == (0, 3)
    single = d[np.concatenate([[True], np.any(d[1:, 1:3] != d[:-1, 1:3], axis=1)])]      # one row per feature: no feature is seen twice
    assert hip_ch.multi_cam_triangulate(single, P, K, D).shape == (0, 3)
    tri.close()
