"""The LIBRARY-backed forms of the LM step (rocBLAS / rocSOLVER through torch) — measurement and test tooling, not product.

They lived in pycamset_amd/device_solver.py until round 4 "for A/B"; the product solves every step with its own HIP kernels
(csrc/ba_schur.hpp, csrc/ba_chol_persist.hpp, csrc/ba_dense_chol.hpp), so the vendor-library variants moved here:
  * ``NormalEquations``: dense H = J^T J from pcs_normal_equations_device as torch tensors + a torch Cholesky step;
  * ``cholesky_step`` / ``schur_cholesky_step``: the damped step on explicit torch matrices (CPU tensors in the tests of the
    host-side LM driver, CUDA tensors = rocSOLVER in the A/B scripts);
  * ``library_schur_solve``: BlockedNormalEquations.solve with the two products and the dense solve done by the libraries
    (what ``dense_solver='rocsolver'`` selected);
  * ``trailing_block_structure`` / ``reduce_normal_equations``: their helpers.
tools/lm_profile.py times them against the HIP kernels; tests/test_host_logic.py and tests/test_sharding_gloo.py use the step
functions on CPU tensors as an independent statement of the algebra.
"""
from __future__ import annotations

import numpy as np


class NormalEquations:
    """H = J^T J, g = J^T r and sum r^2 restricted to the free parameters, built by one pass of the
    block-reduced kernel (csrc/ba_normal.hpp) and kept on the GPU as torch tensors.

    ``reduce_fn`` (optional, sharded detections) sums a float64 vector across ranks; it receives the packed
    [H_ff, g_f, cost] — the "all-reduce of the small result instead of an all-gather of J" of SURVEY 8 f2.
    A callable with attribute ``on_device = True`` is handed the CUDA tensor itself (RCCL), otherwise a
    NumPy copy."""

    def __init__(self, engine, unfixed=None, reduce_fn=None):
        import torch

        self.torch = torch
        self.eng = engine
        mask = np.ones(engine.n_params, dtype=bool) if unfixed is None else np.asarray(unfixed, dtype=bool)
        if mask.shape[0] != engine.n_params:
            raise ValueError("mask must have one entry per parameter")
        self.free = np.flatnonzero(mask)
        self.n_free = self.free.shape[0]
        self.reduce_fn = reduce_fn
        dev = torch.device("cuda", engine.device)
        n = engine.n_params
        self._H = torch.empty((n, n), dtype=torch.float64, device=dev)
        self._g = torch.empty(n, dtype=torch.float64, device=dev)
        self._c = torch.empty(1, dtype=torch.float64, device=dev)
        self._idx = torch.from_numpy(self.free).to(dev)
        self._all_free = self.n_free == n
        self.schur = trailing_block_structure(engine.chain, engine.n_cams, engine.n_imgs, engine.n_keys, mask)
        self._perm = self._inv_perm = None

    def build(self, param_str):
        """-> (H_ff (n_free, n_free) symmetric CUDA tensor, g_f CUDA tensor, sum r^2 float)."""
        torch = self.torch
        with torch.cuda.device(self._H.device):
            stream = torch.cuda.current_stream().cuda_stream
            if self.eng.n == 0:   # empty shard: zeros, but the all-reduce below still happens (see JacobianOperator)
                self._H.zero_(); self._g.zero_(); self._c.zero_()
            else:
                self.eng.normal_equations_device(param_str, self._H.data_ptr(), self._g.data_ptr(), self._c.data_ptr(), stream)
            U = self._H if self._all_free else self._H[self._idx][:, self._idx]
            g = self._g if self._all_free else self._g[self._idx]
            U, g, c = reduce_normal_equations(U, g, self._c, self.reduce_fn)
            Hs = torch.triu(U) + torch.triu(U, 1).T   # the kernel writes the upper triangle only
            return Hs, g.clone(), float(c.item())

    def solve(self, Hs, g, lam, d):
        """delta of (H + lam diag(d)) delta = -g; None if the damped matrix is not positive definite."""
        if self.schur is None:
            return cholesky_step(Hs, g, lam, d)
        n_lead, block, perm = self.schur
        if perm is None:
            return schur_cholesky_step(Hs, g, lam, d, n_lead, block)
        if self._perm is None:
            self._perm = self.torch.from_numpy(perm).to(Hs.device)
            self._inv_perm = self.torch.argsort(self._perm)
        p = self._perm
        delta = schur_cholesky_step(Hs[p][:, p], g[p], lam, d[p], n_lead, block)
        return None if delta is None else delta[self._inv_perm]


def reduce_normal_equations(U, g, c, reduce_fn):
    """Sum one rank's (J^T J, J^T r, cost) torch tensors over the ranks with ONE collective on the packed
    buffer.  ``reduce_fn`` with ``on_device = True`` receives the tensor itself (RCCL on a CUDA tensor, see
    sharding.allreduce_sum_tensor_fn), otherwise a NumPy copy (sharding.allreduce_sum_fn, gloo)."""
    if reduce_fn is None:
        return U, g, c
    import torch

    m = g.shape[0]
    packed = torch.cat([U.reshape(-1), g, c.reshape(1)])
    if getattr(reduce_fn, "on_device", False):
        packed = reduce_fn(packed)
    else:
        packed = torch.from_numpy(reduce_fn(packed.cpu().numpy())).to(U.device)
    return packed[: m * m].view(m, m), packed[m * m: m * m + m], packed[-1:]


def cholesky_step(Hs, g, lam, d):
    """Damped normal equations by Cholesky on torch tensors (rocSOLVER on a CUDA tensor)."""
    import torch

    L, info = torch.linalg.cholesky_ex(Hs + torch.diag(lam * d))
    if int(info.item()) != 0:
        return None
    return torch.cholesky_solve(-g.unsqueeze(1), L).squeeze(1)


def schur_cholesky_step(Hs, g, lam, d, n_lead: int, block: int):
    """The same step through the Schur complement of the trailing parameter group.

    The free parameters are ordered [leading | trailing]; the trailing group (the per-image poses of the
    template chain, the per-key points of the self / free chains) consists of ``block``-sized sets that
    never share a detection, so its part of J^T J is block diagonal:  H = [[A, B], [B^T, C]],
    C = diag(C_1 .. C_m).  Eliminating it leaves a dense system of the leading size only
    (480 instead of 1 680 unknowns on rig-32):
        (A_d - B C_d^-1 B^T) x_a = -g_a + B C_d^-1 g_c,      x_c = -C_d^-1 (g_c + B^T x_a)
    with A_d, C_d the damped blocks.  Returns None when a factorisation fails."""
    import torch

    n = Hs.shape[0]
    m = (n - n_lead) // block
    if m == 0 or n_lead == 0:
        return cholesky_step(Hs, g, lam, d)
    A = Hs[:n_lead, :n_lead] + torch.diag(lam * d[:n_lead])
    B = Hs[:n_lead, n_lead:]                                           # (n_lead, m * block)
    Ct = Hs[n_lead:, n_lead:].reshape(m, block, m, block)
    C = Ct.diagonal(dim1=0, dim2=2).permute(2, 0, 1)                  # (m, block, block) diagonal blocks
    C = C + torch.diag_embed(lam * d[n_lead:].reshape(m, block))
    Lc, info_c = torch.linalg.cholesky_ex(C)
    if int(info_c.max().item()) != 0:
        return None
    Cinv = torch.cholesky_inverse(Lc)
    BCinv = torch.einsum("amk,mkl->aml", B.reshape(n_lead, m, block), Cinv).reshape(n_lead, m * block)
    S = A - BCinv @ B.T
    rhs = -g[:n_lead] + BCinv @ g[n_lead:]
    Ls, info_s = torch.linalg.cholesky_ex(S)
    if int(info_s.item()) != 0:
        return None
    xa = torch.cholesky_solve(rhs.unsqueeze(1), Ls).squeeze(1)
    t = (g[n_lead:] + B.T @ xa).reshape(m, block)
    xc = -torch.einsum("mkl,ml->mk", Cinv, t).reshape(-1)
    return torch.cat([xa, xc])


def trailing_block_structure(chain: str, n_cams: int, n_imgs: int, n_keys: int, mask):
    """(n_lead, block, perm) for schur_cholesky_step, or None if there is nothing to eliminate.
    template: poses (6 per image) trail the cameras; self / free: points (3 per key) trail everything
    else.  Sets that are only partly free (single point coordinates fixed by the self-calibration gauge,
    sbh:153-158) are moved to the leading group: ``perm`` is that reordering of the free-parameter vector
    (None when it is the identity)."""
    mask = np.asarray(mask, dtype=bool)
    block = 6 if chain == "template" else 3
    start = 15 * n_cams if chain in ("template", "free") else 15 * n_cams + 6 * n_imgs
    sets = mask[start:].reshape(-1, block)
    whole = sets.all(axis=1)
    if not whole.any() or not (mask[:start].any() or (sets.any(axis=1) & ~whole).any()):
        return None
    free_pos = np.cumsum(mask) - 1                       # full index -> position in the free vector
    trailing = np.repeat(whole, block)
    full_idx = np.arange(mask.shape[0])
    lead_full = np.concatenate([full_idx[:start][mask[:start]], (full_idx[start:])[mask[start:] & ~trailing]])
    trail_full = (full_idx[start:])[trailing]
    perm = free_pos[np.concatenate([lead_full, trail_full])]
    n_lead = lead_full.shape[0]
    return n_lead, block, (None if np.array_equal(perm, np.arange(perm.shape[0])) else perm)


def library_schur_solve(ne, slot: int, lam, ps=None, ps_out=None):
    """``BlockedNormalEquations.solve`` with rocBLAS GEMM / GEMV and rocSOLVER potrf / potrs (through torch) in place of
    pcs_schur_syrk / pcs_dense_spd_solve / pcs_schur_vtx — the A/B partner of the HIP step (1.6 ms against 0.13 ms for the dense
    solve at n = 480).  Leaves ``ne.S`` = the reduced matrix (the HIP solver factors it in place)."""
    torch = ne.torch
    with ne.on_stream() as stream:
        ne.eng.schur_prepare(ne.packed[slot].data_ptr(), ne.fixed.data_ptr(), lam.data_ptr(), ne.linvt.data_ptr(), ne.u.data_ptr(), ne.V.data_ptr(),
                             ne.S.data_ptr(), ne.rhs.data_ptr(), ne.dvec.data_ptr(), ne.gm.data_ptr(), ne.status.data_ptr(), stream)
        if ne.n_trail:
            V = ne.V[:, : ne.n_trail]
            ne.S.addmm_(V, V.T, alpha=-1.0)
            ne.rhs.addmv_(V, ne.u[: ne.n_trail])
        L, info = torch.linalg.cholesky_ex(ne.S)   # `info` stays on the device
        xl = torch.cholesky_solve(ne.rhs.unsqueeze(1), L).squeeze(1)
        ne.status.bitwise_or_((info != 0).to(torch.int32) * 2)
        w = torch.mv(ne.V[:, : ne.n_trail].T, xl) if ne.n_trail else ne.u
        ne.eng.schur_finish(ne.linvt.data_ptr(), ne.u.data_ptr(), w.data_ptr(), xl.data_ptr(), ne.fixed.data_ptr(), ne.delta.data_ptr(),
                            ps.data_ptr() if ps is not None else 0, ps_out.data_ptr() if ps is not None else 0, stream)
    return ne.delta
