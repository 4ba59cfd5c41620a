"""Observation sharding across the GPUs of one node.

One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI on ROCm, "gloo" on
CPU for the tests).  Detections are independent given the parameter string, so the table is split
into contiguous ranges of ``ceil(N / G)`` rows — the same equal-chunk rule as the reference's
thread split (abstract_function_blocks.py:281-288): the last chunk is padded by cyclic repetition
(``np.resize``) so every rank holds the same count, and the padding is dropped after the gather
(afb:385, afb:641).  Computing needs no exchange.  The only collective is the optional all-gather
of the residual / Jacobian blocks, which is link-bound (44 MB per GPU at N = 1e6, P = 21) and is
therefore timed separately from the kernel in bench.py.
"""
from __future__ import annotations

import numpy as np


def shard_rows(n: int, world: int) -> int:
    """Rows per rank: ceil(n / world) (afb:286)."""
    return -(-n // world)


def padded_shard(det: np.ndarray, rank: int, world: int) -> np.ndarray:
    """Rank ``rank``'s contiguous block of the table padded to equal length by cyclic repetition.
    For GATHERING blocks only (the padding is dropped after the gather): sums over shards — J^T J, J^T r, the cost —
    must use the unpadded ranges ``det[rank * per : (rank + 1) * per]`` (possibly empty for the last ranks), or the
    repeated rows are counted twice."""
    per = shard_rows(det.shape[0], world)
    idx = (np.arange(rank * per, (rank + 1) * per)) % det.shape[0]
    return np.ascontiguousarray(det[idx])


class ShardedEvaluator:
    """Evaluate one rank's shard and (optionally) all-gather the blocks.

    ``local_eval(param_str, want_resid, want_jac) -> (resid_tensor (per,2) | None, jac_tensor (2*per,P) | None)``
    produces this rank's blocks as torch tensors on the rank's device.  ``make_engine_eval`` builds
    that callable on top of the HIP engine; tests inject a CPU stand-in to exercise the
    partition / gather logic under gloo.
    """

    def __init__(self, n_total: int, row_len: int, local_eval, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_total = n_total
        self.per = shard_rows(n_total, self.world)
        self.P = row_len
        self.local_eval = local_eval
        self._gr = self._gj = None

    def eval_local(self, param_str, want_resid=True, want_jac=True):
        return self.local_eval(param_str, want_resid, want_jac)

    def all_gather(self, resid, jac):
        """-> (resid (N,2), jac (2N,P)) on every rank, padding removed.  Uses preallocated outputs."""
        import torch

        out_r = out_j = None
        if resid is not None:
            if self._gr is None:
                self._gr = torch.empty((self.world * self.per, 2), dtype=resid.dtype, device=resid.device)
            if self.world > 1:
                self.dist.all_gather_into_tensor(self._gr, resid.contiguous(), group=self.group)
            else:
                self._gr.copy_(resid)
            out_r = self._gr[: self.n_total]
        if jac is not None:
            if self._gj is None:
                self._gj = torch.empty((self.world * 2 * self.per, self.P), dtype=jac.dtype, device=jac.device)
            if self.world > 1:
                self.dist.all_gather_into_tensor(self._gj, jac.contiguous(), group=self.group)
            else:
                self._gj.copy_(jac)
            out_j = self._gj[: 2 * self.n_total]
        return out_r, out_j

    def eval_gathered(self, param_str, want_resid=True, want_jac=True):
        r, j = self.eval_local(param_str, want_resid, want_jac)
        return self.all_gather(r, j)


def make_engine_eval(chain: str, det_full: np.ndarray, counts, template=None, *, dtype: str = "f64", device: int = 0,
                     rank: int = 0, world: int = 1):
    """Build ``local_eval`` for ShardedEvaluator on top of the HIP engine: the rank's padded shard is
    uploaded once; every call launches the fused kernel on torch's current stream and writes
    straight into torch tensors (no copy)."""
    import torch

    from .engine import Engine

    shard = padded_shard(det_full, rank, world)
    eng = Engine(chain, *counts, dtype=dtype, device=device)
    eng.set_detections_table(shard)
    if template is not None:
        eng.set_template(template)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    dev = torch.device("cuda", device)
    per = shard.shape[0]
    d_r = torch.empty((per, 2), dtype=tdt, device=dev)
    d_j = torch.empty((2 * per, eng.P), dtype=tdt, device=dev)

    def local_eval(param_str, want_resid=True, want_jac=True):
        stream = torch.cuda.current_stream(dev).cuda_stream
        eng.eval_device(param_str, d_r.data_ptr() if want_resid else None, d_j.data_ptr() if want_jac else None, stream)
        return (d_r if want_resid else None), (d_j if want_jac else None)

    local_eval.engine = eng
    return local_eval


def allreduce_sum_fn(group=None, device=None):
    """``reduce_fn`` for ``device_solver.JacobianOperator``: sum an n-vector (J^T J v, diag, J^T r,
    cost — a few KB) over the ranks.  This replaces the all-gather of J: with the
    Jacobian kept on each GPU only parameter-sized vectors cross xGMI.  ``device`` = a torch CUDA
    device for the RCCL backend; None keeps the tensor on the host (gloo)."""
    import torch
    import torch.distributed as dist

    def reduce_fn(vec: np.ndarray) -> np.ndarray:
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return vec
        t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64))
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()

    return reduce_fn


def allreduce_sum_tensor_fn(group=None):
    """``reduce_fn`` for ``device_solver.NormalEquations`` on the RCCL backend: sums the packed
    [J^T J, J^T r, cost] CUDA tensor across the ranks in place, without a host copy."""
    import torch.distributed as dist

    def reduce_fn(t):
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t

    reduce_fn.on_device = True
    return reduce_fn
