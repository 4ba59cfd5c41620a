"""World-size-2 (and 3) gloo test of the observation sharding: contiguous equal chunks padded by
cyclic repetition, all-gather, padding dropped — the gathered blocks must equal the single-process
evaluation bit for bit.  The CPU oracle stands in for the per-rank kernel (CPU box, no GPU)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ba_oracle as orc
        from pycamset_amd import sharding, synthetic

        rig = synthetic.tiny_rig(seed=3, n_cams=3, n_imgs=5, n_keys=9)  # N not a multiple of 2 or 3
        det = rig.detections
        counts = orc.counts_from_detections(det)
        ps = orc.build_param_list(rig.intr, rig.extr, rig.poses)
        shard = sharding.padded_shard(det, rank, world)

        def local_eval(param_str, want_resid=True, want_jac=True):
            j, r = orc.full_jac_dense("template", shard, param_str, rig.points, with_resid=True, counts=counts)
            return (torch.from_numpy(r) if want_resid else None), (torch.from_numpy(j) if want_jac else None)

        ev = sharding.ShardedEvaluator(det.shape[0], 21, local_eval)
        assert ev.per == -(-det.shape[0] // world) and shard.shape[0] == ev.per
        r, j = ev.eval_gathered(ps)
        ref_j, ref_r = orc.full_jac_dense("template", det, ps, rig.points, with_resid=True)
        assert r.shape == (det.shape[0], 2) and j.shape == (2 * det.shape[0], 21)
        assert np.array_equal(r.numpy(), ref_r) and np.array_equal(j.numpy(), ref_j)
        r_only, none = ev.eval_gathered(ps, want_jac=False)
        assert none is None and np.array_equal(r_only.numpy(), ref_r)
        Path(out_dir, f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_allgather_matches_single_process(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _worker_bench_path(rank, world, port, out_dir):
    """bench.py's own strong-scaling code path (rank_problem -> per-rank evaluation -> gather_blocks -> strip_padding)
    with the CPU oracle standing in for the kernel: the gathered blocks equal the single-process output."""
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from oracle import ba_oracle as orc

        prob = bench.rank_problem(1, "template", rank, world, "strong", scale=0.5)     # ccube-plumbing, N ~ 3.5e3
        rig, shard, ps = prob["rig"], prob["det"], prob["param_str"]
        n_total, per = prob["n_total"], prob["per"]
        assert n_total == rig.n_det and shard.shape[0] == per == -(-n_total // world)
        assert prob["n_real"] == max(0, min(n_total, (rank + 1) * per) - rank * per)
        counts = (rig.n_cams, rig.n_imgs, rig.n_keys)
        j, r = orc.full_jac_dense("template", shard, ps, prob["template"], with_resid=True, counts=counts)
        g_r = bench.strip_padding(bench.gather_blocks(dist, torch.from_numpy(r)), n_total)
        g_j = bench.strip_padding(bench.gather_blocks(dist, torch.from_numpy(j)), n_total, rows_per_det=2)
        g_r2 = bench.strip_padding(bench.gather_blocks(dist, torch.from_numpy(r), via_host=True), n_total)
        ref_j, ref_r = orc.full_jac_dense("template", rig.detections, ps, rig.points, with_resid=True, counts=counts)
        assert np.array_equal(g_r.numpy(), ref_r) and np.array_equal(g_j.numpy(), ref_j) and np.array_equal(g_r2.numpy(), ref_r)
        # every rank holds the SAME parameter string and the rows of all ranks add up to the table
        t = torch.tensor([float(prob["n_real"]), float(np.sum(ps))], dtype=torch.float64)
        dist.all_reduce(t)
        assert int(t[0]) == n_total and abs(float(t[1]) - world * float(np.sum(ps))) <= 1e-9 * abs(float(t[1]))
        w = bench.rank_problem(1, "template", rank, world, "weak", scale=0.5)
        assert w["n_total"] is None and w["det"].shape[0] == w["n_real"] and (rank == 0 or not np.array_equal(w["det"][:50], rig.detections[:50]))
        Path(out_dir, f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])   # 8: the rank count of the driver's scaling run (host logic + gloo only here)
def test_bench_strong_scaling_path_gathers_the_single_process_output(tmp_path, world):
    mp.spawn(_worker_bench_path, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_padded_shard_rule():
    from pycamset_amd import sharding

    det = np.arange(35, dtype=np.float64).reshape(7, 5)
    assert sharding.shard_rows(7, 2) == 4
    s0, s1 = sharding.padded_shard(det, 0, 2), sharding.padded_shard(det, 1, 2)
    assert np.array_equal(s0, det[:4])
    assert np.array_equal(s1, np.concatenate([det[4:], det[:1]]))  # cyclic repetition like np.resize (afb:281-288)
    assert np.array_equal(np.concatenate([s0, s1])[:7], det)


class _OracleEngine:
    """CPU stand-in with the Engine's matrix-free interface, built from the oracle's Jacobian of one
    shard (tests only): lets the sharded normal-equation algebra run under gloo without a GPU."""

    def __init__(self, chain, det, counts, template):
        from oracle import ba_oracle as orc
        self.orc, self.chain, self.det, self.counts, self.template = orc, chain, det, counts, template
        self.n = det.shape[0]
        self.n_params = 15 * counts[0] + 6 * counts[1]

    def linearize(self, ps):
        from scipy.sparse import csr_array
        orc = self.orc
        dense, r = orc.full_jac_dense(self.chain, self.det, ps, self.template, with_resid=True, counts=self.counts)
        cols = np.repeat(self._cols(), 2, axis=0)
        ptr = np.arange(0, dense.size + 1, dense.shape[1])
        self.J = csr_array((dense.reshape(-1), cols.reshape(-1), ptr), shape=(2 * self.n, self.n_params))
        self.r = r.reshape(-1)

    def _cols(self):
        C = self.counts[0]
        c, i = self.det[:, 0].astype(np.int64), self.det[:, 1].astype(np.int64)
        return np.concatenate([9 * c[:, None] + np.arange(9), 9 * C + 6 * c[:, None] + np.arange(6),
                               15 * C + 6 * i[:, None] + np.arange(6)], axis=1)

    def jtjv(self, v):
        return self.J.T @ (self.J @ v)

    def jtj_diag(self):
        return np.asarray(self.J.multiply(self.J).sum(axis=0)).ravel()

    def grad(self):
        return self.J.T @ self.r, float(self.r @ self.r)


def _worker_normal_eq(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ba_oracle as orc
        from pycamset_amd import sharding, synthetic
        from pycamset_amd.device_solver import JacobianOperator

        rig = synthetic.tiny_rig(seed=4, n_cams=3, n_imgs=5, n_keys=9)
        det = rig.detections
        counts = orc.counts_from_detections(det)
        ps = orc.build_param_list(rig.intr, rig.extr, rig.poses)
        per = sharding.shard_rows(det.shape[0], world)
        mine = det[rank * per: (rank + 1) * per]           # unpadded shard: sums must not double count
        mask = np.ones(ps.shape[0], bool)
        mask[15 * counts[0]: 15 * counts[0] + 6] = False   # pose 0 fixed
        op = JacobianOperator(_OracleEngine("template", mine, counts, rig.points), mask, reduce_fn=sharding.allreduce_sum_fn())
        op.linearize(ps)
        full = JacobianOperator(_OracleEngine("template", det, counts, rig.points), mask)
        full.linearize(ps)
        v = np.random.default_rng(0).standard_normal(op.n_free)
        for a, b in ((op.jtjv(v), full.jtjv(v)), (op.diag(), full.diag()), (op.grad()[0], full.grad()[0])):
            assert np.max(np.abs(a - b)) <= 1e-12 * np.max(np.abs(b))
        assert abs(op.grad()[1] - full.grad()[1]) <= 1e-12 * full.grad()[1]

        # block-reduced form: every rank's (J^T J, J^T r, cost) packed into ONE all-reduce, both flavours of reduce_fn
        import torch
        from tools.library_solver import reduce_normal_equations, schur_cholesky_step, cholesky_step, trailing_block_structure
        free = np.flatnonzero(mask)
        Jm, Jf = op.eng.J[:, free], full.eng.J[:, free]
        H_full, g_full, c_full = (Jf.T @ Jf).toarray(), Jf.T @ full.eng.r, float(full.eng.r @ full.eng.r)
        for fn in (sharding.allreduce_sum_fn(), sharding.allreduce_sum_tensor_fn()):
            U = torch.from_numpy(np.triu((Jm.T @ Jm).toarray()))         # the kernel writes the upper triangle
            Ur, gr, cr = reduce_normal_equations(U, torch.from_numpy(Jm.T @ op.eng.r), torch.tensor([float(op.eng.r @ op.eng.r)], dtype=torch.float64), fn)
            Hs = torch.triu(Ur) + torch.triu(Ur, 1).T
            assert np.max(np.abs(Hs.numpy() - H_full)) <= 1e-12 * np.max(np.abs(H_full))
            assert np.max(np.abs(gr.numpy() - g_full)) <= 1e-12 * np.max(np.abs(g_full)) and abs(float(cr) - c_full) <= 1e-12 * c_full
        n_lead, blk, perm = trailing_block_structure("template", *counts, mask)
        assert perm is None and blk == 6
        dd = torch.diagonal(Hs).clone()
        a, b = cholesky_step(Hs, gr, 1e-3, dd), schur_cholesky_step(Hs, gr, 1e-3, dd, n_lead, blk)
        assert float((a - b).abs().max()) <= 1e-8 * float(a.abs().max())
        Path(out_dir, f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


def test_sharded_normal_equations_allreduce(tmp_path):
    """ranks all-reduce J^T J v / diag / J^T r (parameter-sized) instead of gathering J."""
    world = 2
    mp.spawn(_worker_normal_eq, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _worker_empty_shard(rank, world, port, out_dir):
    """N = 9 detections on 4 ranks: ceil(9 / 4) = 3 rows per rank leaves rank 3 with none.  Its operator must contribute
    zeros and still enter every all-reduce (the other ranks used to block forever)."""
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ba_oracle as orc
        from pycamset_amd import sharding, synthetic
        from pycamset_amd.device_solver import JacobianOperator

        rig = synthetic.tiny_rig(seed=4, n_cams=3, n_imgs=5, n_keys=9)
        det = rig.detections[:9].copy()
        det[-1, :3] = [rig.n_cams - 1, rig.n_imgs - 1, rig.n_keys - 1]
        counts = (rig.n_cams, rig.n_imgs, rig.n_keys)
        ps = orc.build_param_list(rig.intr, rig.extr, rig.poses)
        per = sharding.shard_rows(det.shape[0], world)
        mine = det[rank * per: (rank + 1) * per]
        assert (mine.shape[0] == 0) == (rank == 3)
        mask = np.ones(ps.shape[0], bool)
        op = JacobianOperator(_OracleEngine("template", mine, counts, rig.points), mask, reduce_fn=sharding.allreduce_sum_fn())
        op.linearize(ps)
        full = JacobianOperator(_OracleEngine("template", det, counts, rig.points), mask)
        full.linearize(ps)
        v = np.random.default_rng(1).standard_normal(op.n_free)
        assert np.allclose(op.jtjv(v), full.jtjv(v), rtol=1e-12, atol=1e-9) and np.allclose(op.diag(), full.diag(), rtol=1e-12, atol=1e-12)
        g, c = op.grad()
        gf, cf = full.grad()
        assert np.allclose(g, gf, rtol=1e-12, atol=1e-9) and abs(c - cf) <= 1e-12 * cf
        Path(out_dir, f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


def test_empty_shard_contributes_zeros_and_does_not_deadlock(tmp_path):
    world = 4
    mp.spawn(_worker_empty_shard, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _capture_self_launch(capfd, n, argv):
    import bench

    rc = bench.self_launch(n, argv, script=REPO / "tests" / "_bench_stub.py")
    out, err = capfd.readouterr()
    return rc, out, err


def test_bench_launches_its_own_ranks(capfd):
    """`python bench.py --gpus N` with no launcher around it: the parent starts N children through
    torch.distributed.run, relays exactly ONE JSON line (rank 0's) on stdout and sends everything else to stderr.  Driven
    here with a stub per-rank body (tests/_bench_stub.py: gloo rendezvous + all-reduce, no GPU)."""
    import json

    rc, out, err = _capture_self_launch(capfd, 3, ["--gpus", "3", "--steps", "5"])
    assert rc == 0, err
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, (out, err[-3000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 3 and line["sum"] == 6.0 and line["argv"] == ["--gpus", "3", "--steps", "5"]
    assert "noise from rank 0" in err and "noise from rank 2" in err


def test_bench_self_launch_reports_a_failed_rank(capfd):
    rc, out, err = _capture_self_launch(capfd, 2, ["--fail-rank", "1"])
    assert rc != 0


def test_bench_main_becomes_the_launcher_only_without_world_size(monkeypatch):
    """main(): --gpus N > 1 without WORLD_SIZE -> self_launch (and its exit code); with a WORLD_SIZE that disagrees -> an error,
    never a second launcher."""
    import bench

    calls = []
    monkeypatch.setattr(bench, "self_launch", lambda n, argv, **kw: calls.append((n, list(argv))) or 7)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7 and calls == [(4, ["--gpus", "4", "--steps", "3"])]
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE" in str(e.value.code) and len(calls) == 1
