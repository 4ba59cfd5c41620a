"""Levenberg-Marquardt with the Jacobian kept on the device.

The reference hands a 2N x n_free CSR matrix to ``scipy.optimize.least_squares``
(optimisation_handling.py:88-98), which scales its columns (x_scale='jac'), forms J^T f and runs
lsmr mat-vecs on the host.  Here J only exists as matrix-free products on the GPU
(csrc/ba_matfree.hpp): per LM iteration the host sees n_params-sized vectors only, so neither the
PCIe copy of J (335 MB at N = 1e6) nor — when sharded — an all-gather of J is needed: ranks
all-reduce J^T J v, diag(J^T J) and J^T r (a few KB).

    op  = JacobianOperator(engine, unfixed_mask)          # products in the FREE parameter space
    res = lm_solve(handler, x0)                           # drop-in for run_bundle_adjustment's solve
    res = lm_solve(handler, x0, linear_solver="cholesky") # block-reduced J^T J per step + dense Cholesky

Two ways to get the LM step from the GPU: ``"pcg"`` — Jacobi-preconditioned CG on matrix-free
J^T (J v) products (~150 passes over the detections per step, parameter-sized traffic only); and
``"cholesky"`` — one pass of csrc/ba_normal.hpp builds J^T J, J^T r and the cost in BLOCKED form
([A | B | C]: leading x leading, leading x trailing, block-diagonal trailing group) and the damped system is
reduced by the Schur complement of the trailing group (csrc/ba_schur.hpp: the block parts and both products on the FP64
matrix cores; csrc/ba_chol_persist.hpp: the dense Cholesky of the leading size in one persistent launch) — HIP kernels only; the
vendor-library variants kept for A/B until round 4 live in tools/library_solver.py.  That loop is device-resident: parameter
string, step, damping, gain ratio, the accept / reject decision AND the termination rules live in HBM, the host reads ONE small
vector per trial to follow it.  The sharded form all-reduces the packed [A | B | C | g | cost | votes] once per trial — on the
solver's stream between the two halves of a trial when the collective is RCCL (no host synchronisation), through the host for gloo.

``as_linear_operator()`` exposes J as a scipy LinearOperator (matvec / rmatvec) for callers that
want scipy's own solvers without materialising J.
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass, field

import time

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

    # A rank whose shard is EMPTY (ceil(N / world) rows per rank can leave the last ranks without any: N = 9, world = 4)
    # contributes zeros and still takes part in every collective — raising there would leave the other ranks blocked
    # in their all-reduce.
    @property
    def _empty(self) -> bool:
        return self.eng.n == 0

    def linearize(self, param_str):
        if not self._empty:
            self.eng.linearize(param_str)

    def jv(self, v_free):          # (2N_local,) — stays per rank
        return np.zeros(0) if self._empty else self.eng.jv(self._expand(v_free))

    def jtu(self, u):
        return self._red(np.zeros(self.n_free) if self._empty else self.eng.jtu(u)[self.free])

    def jtjv(self, v_free):
        return self._red(np.zeros(self.n_free) if self._empty else self.eng.jtjv(self._expand(v_free))[self.free])

    def diag(self):
        return self._red(np.zeros(self.n_free) if self._empty else self.eng.jtj_diag()[self.free])

    def grad(self):
        g, c = (np.zeros(self.eng.n_params), 0.0) if self._empty else self.eng.grad()
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


LAM0_EXACT = 1e-5           # initial damping of the exact (Cholesky) step; 1e-3 for the inexact PCG step (lm_solve's docstring has the why)
LAM_GROW0 = 1e3             # factor a rejected trial applies to lambda before the first accepted step (lm_solve)
LAM_FAST = (0.95, 0.1)      # a gain ratio above 0.95 multiplies lambda by 0.1 instead of 1/3: a model THAT accurate lets the damping go quickly
LEAD_LIMIT = 16384          # leading parameters up to which lm_solve(linear_solver="auto") takes the Schur / Cholesky step (S: 2 GB)
BLOCKED_BYTES_LIMIT = 96e9  # and total bytes of the two packed buffers + V + S (a third of the part's 288 GB)


REGION_LIMIT = 2 ** 32     # doubles per region of the packed buffer: the build addresses A, B and C with 32-bit offsets in doubles (pcs_engine.hip enqueue_normal; 2^29 until round 4)


def blocked_fits(engine) -> bool:
    """Can ``linear_solver='auto'`` take the blocked normal equations + Schur / Cholesky step on this engine?  False for leading
    groups beyond LEAD_LIMIT, for buffers beyond BLOCKED_BYTES_LIMIT and for any single region A / B / C of 2^32 doubles or more (the
    build would refuse it).  A generated chain has the DENSE form of the same state (every parameter leading, csrc/ba_blockgram.hpp):
    the same limits apply to its n_params, plus the contraction's own (FP64 block rows of at most 63 columns)."""
    if not hasattr(engine, "normal_layout"):
        return False
    if hasattr(engine, "dense_lm_supported") and not engine.dense_lm_supported():
        return False
    lay = engine.normal_layout()
    n_lead, n_trail, tb = lay["n_lead"], lay["n_trail"], lay["tb"]
    if max(n_lead * n_lead, n_lead * n_trail, n_trail * tb) >= REGION_LIMIT:
        return False
    return n_lead <= LEAD_LIMIT and 8.0 * (2 * lay["packed_len"] + n_lead * n_trail + n_lead ** 2) <= BLOCKED_BYTES_LIMIT


class BlockedNormalEquations:
    """J^T J in blocked form + the Schur-complement step, all on the device (engine.normal_blocks_device,
    engine.schur_prepare / schur_finish; include/pcs_hip.h).  Two packed states (current, trial): in the device-steered loop a
    device word says which is which and an accepted trial flips it (include/pcs_hip.h pcs_lm_buffers)."""

    def __init__(self, engine, unfixed=None, reduce_fn=None):
        import torch

        self.torch, self.eng, self.reduce_fn = torch, engine, reduce_fn
        lay = engine.normal_layout()
        self.n_lead, self.n_trail, self.tb, self.n_params = lay["n_lead"], lay["n_trail"], lay["tb"], lay["n_params"]
        self.n_ent = self.n_trail // self.tb
        self.n_packed = lay["packed_len"]            # [A | B | C | g | cost]; one more word behind it: the ranks' void votes
        mask = np.ones(self.n_params, dtype=bool) if unfixed is None else np.asarray(unfixed, dtype=bool)
        if mask.shape[0] != self.n_params:
            raise ValueError("mask must have one entry per parameter")
        self.free = np.flatnonzero(mask)
        self.n_free = self.free.shape[0]
        dev = self.dev = torch.device("cuda", engine.device)
        f64 = dict(dtype=torch.float64, device=dev)
        n_alloc = self.n_packed + 1 + ((self.n_packed + 1) & 1)     # an even number of doubles: both states 16-byte aligned in any allocator
        self.packed = [torch.zeros(n_alloc, **f64)[: self.n_packed + 1] for _ in range(2)]
        self.fixed = torch.from_numpy((~mask).astype(np.uint8)).to(dev)
        self.free_idx = torch.from_numpy(self.free).to(dev)
        self.linvt = torch.empty(max(1, self.n_ent * self.tb * self.tb), **f64)
        self.u = torch.empty(max(1, self.n_trail), **f64)
        self.V = torch.empty((self.n_lead, max(1, self.n_trail)), **f64)
        self.S = torch.empty((self.n_lead, self.n_lead), **f64)
        self.rhs = torch.empty(self.n_lead, **f64)
        self.dvec = torch.empty(self.n_params, **f64)
        self.gm = torch.empty(self.n_params, **f64)
        self.delta = torch.empty(self.n_params, **f64)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.stream = torch.cuda.Stream(device=dev)
        from .engine import dense_spd_work_len

        self.xl = torch.empty(self.n_lead, **f64)
        self.w = torch.empty(max(1, self.n_trail), **f64)
        self.chol_work = torch.empty(dense_spd_work_len(self.n_lead), **f64)
        from .engine import schur_syrk_work_len

        # the ordered S -= V V' of the engine's deterministic mode parks the partial products of its K split here (0: not split)
        self.syrk_work_len = schur_syrk_work_len(self.n_lead, self.n_trail) if self.n_trail else 0
        self.syrk_work = torch.empty(max(1, self.syrk_work_len), **f64)
        self._cons = None   # sharded host-steered loop only: the ranks' consensus step (n_params + 2)
        self.spd_algorithm = "auto"   # 'launches' after a one-launch solve ran out of time (another process held the compute units)

    def cost(self, slot):
        return self.packed[slot][self.n_packed - 1]

    @contextlib.contextmanager
    def on_stream(self):
        """Kernels launched through the C ABI and the few torch operations around them (copies, the collective of a sharded loop)
        must land on ONE stream handle.  torch's default stream is the NULL handle, which the C ABI can only name as
        ``hipStreamLegacy`` — two spellings the runtime is not guaranteed to order against each other in both directions.  So
        whenever the caller is on the default stream the work moves to a stream of this object (ordered after what the caller
        queued, and the caller's later work after ours); a caller that already runs on a real stream keeps it."""
        torch = self.torch
        cur = torch.cuda.current_stream(self.dev)
        if cur.cuda_stream != 0:
            yield cur.cuda_stream
            return
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            yield self.stream.cuda_stream
        cur.wait_stream(self.stream)

    def reduce(self, buf):
        """Sum ``buf`` (a packed state incl. its vote word) over the ranks, in place: stream-ordered on the device when the
        callable says ``on_device`` (RCCL), through the host otherwise (gloo)."""
        if self.reduce_fn is None:
            return
        if getattr(self.reduce_fn, "on_device", False):
            self.reduce_fn(buf)
        else:
            buf.copy_(self.torch.from_numpy(np.asarray(self.reduce_fn(buf.cpu().numpy()), dtype=np.float64)))

    def build(self, ps, slot: int):
        """packed[slot] <- [A | B | C | g | cost] at the device parameter string ``ps`` (+ the sum over the ranks)."""
        buf = self.packed[slot]
        with self.on_stream() as stream:
            if self.eng.n == 0:   # empty shard: zeros, but the all-reduce below still happens (see JacobianOperator)
                buf[: self.n_packed].zero_()
            else:
                self.eng.normal_blocks_device(ps.data_ptr(), buf.data_ptr(), stream)
            self.reduce(buf)

    def solve(self, slot: int, lam, ps=None, ps_out=None):
        """Enqueue the damped step (H + lam diag(H)) delta = -g for the state in packed[slot]: ``self.delta`` (n_params,
        parameter-string order, 0 where fixed) and — given the current parameter string ``ps`` — the trial string
        ``ps_out = ps + delta``.  Device work only; ``self.status`` becomes non-zero when a factorisation fails."""
        with self.on_stream() as stream:
            self.eng.schur_prepare(self.packed[slot].data_ptr(), self.fixed.data_ptr(), lam.data_ptr(), self.linvt.data_ptr(), self.u.data_ptr(),
                                   self.V.data_ptr(), self.S.data_ptr(), self.rhs.data_ptr(), self.dvec.data_ptr(), self.gm.data_ptr(),
                                   self.status.data_ptr(), stream)
            ldv = self.V.shape[1]
            # S = A + lam D - V V' (lower triangle), rhs = -g_l + V u: MFMA kernel of csrc/ba_schur.hpp; S x_l = rhs: blocked
            # Cholesky + substitutions (csrc/ba_chol_persist.hpp / ba_dense_chol.hpp); w = V' x_l
            from .engine import dense_spd_solve, schur_syrk, schur_vtx

            xl = self.xl
            if self.n_trail:
                det = bool(self.eng.option("deterministic", 0))
                schur_syrk(self.eng.device, self.n_lead, self.n_trail, self.V.data_ptr(), ldv, self.S.data_ptr(), self.n_lead,
                           self.u.data_ptr(), self.rhs.data_ptr(), stream, work=self.syrk_work.data_ptr() if det else None, work_len=self.syrk_work_len)
            dense_spd_solve(self.eng.device, self.n_lead, self.S.data_ptr(), self.n_lead, self.rhs.data_ptr(), xl.data_ptr(),
                            self.chol_work.data_ptr(), self.status.data_ptr(), stream, algorithm=self.spd_algorithm,
                            timeout_us=self.eng.option("spd_timeout_us", None))
            if self.n_trail:
                schur_vtx(self.eng.device, self.n_lead, self.n_trail, self.V.data_ptr(), ldv, xl.data_ptr(), self.w.data_ptr(), stream)
                w = self.w
            else:
                w = self.u
            self.eng.schur_finish(self.linvt.data_ptr(), self.u.data_ptr(), w.data_ptr(), xl.data_ptr(), self.fixed.data_ptr(), self.delta.data_ptr(),
                                  ps.data_ptr() if ps is not None else 0, ps_out.data_ptr() if ps is not None else 0, stream)
        return self.delta

    def _consensus_step(self, ps, ps_out):
        """Sharded host-steered loop WITHOUT the engine's deterministic mode: every rank has solved the SAME all-reduced system, but
        schur_syrk_kernel sums its K splits with f64 atomics in arrival order (csrc/ba_schur.hpp), so the steps agree to the last
        bits only — enough for one rank to meet xtol, or to take the other branch of the gain-ratio rule, while its peers enter the
        next all-reduce (a hang).  The ranks therefore adopt ONE step: the mean of theirs, from one more all-reduce of
        n_params + 1 doubles (the extra entry counts the ranks; an all-reduce delivers the same bits to every rank).  With
        ``set_option('deterministic', 1)`` the steps are bit-identical and this collective is not needed."""
        torch = self.torch
        if self._cons is None:
            self._cons = torch.empty(self.n_params + 1, dtype=torch.float64, device=self.dev)
        buf = self._cons
        buf[:-1].copy_(self.delta)
        buf[-1] = 1.0
        self.reduce(buf)
        torch.div(buf[:-1], buf[-1], out=self.delta)
        torch.add(ps, self.delta, out=ps_out)

    def decide(self, cur: int, new: int, ps, lam, stats):
        """The accept / reject decision of the trial state packed[new] against packed[cur] on the device (pcs_lm_decide):
        updates ``lam`` in place, clears ``status`` and fills ``stats`` (12 doubles) — what the host reads once per trial."""
        last = 8 * (self.n_packed - 1)
        with self.on_stream() as stream:
            self.eng.lm_decide(self.packed[cur].data_ptr() + last, self.packed[new].data_ptr() + last, self.dvec.data_ptr(), self.gm.data_ptr(),
                               self.delta.data_ptr(), ps.data_ptr(), self.fixed.data_ptr(), self.status.data_ptr(), lam.data_ptr(), stats.data_ptr(),
                               stream)

    def predicted_reduction(self, lam):
        """(0.5 (lam d'D d - g'd), step is valid) of the last ``solve`` as device tensors (tests; the loop uses ``decide``)."""
        torch = self.torch
        pred = 0.5 * (lam[0] * torch.dot(self.dvec, self.delta * self.delta) - torch.dot(self.gm, self.delta))
        return pred, (self.status[0] == 0) & torch.isfinite(pred)

    def gradient(self, slot: int, lam=None):
        """masked J^T r of the state in packed[slot] (free entries), as NumPy — one read-back, used once at the end.  g sits in the
        packed buffer right behind the blocks ([A | B | C | g | cost]); no solve is needed to read it."""
        g0 = self.n_packed - 1 - self.n_params
        return self.packed[slot][g0: g0 + self.n_params][self.free_idx].cpu().numpy()


def _lm_solve_blocked(ne: BlockedNormalEquations, ps0: np.ndarray, *, max_iter, ftol, xtol, gtol, lam0, lam_grow0, verbose):
    """The device-resident loop behind ``lm_solve(..., linear_solver='cholesky')``.  Per trial step the host enqueues
    step (+ trial parameter string) -> build -> decision and then reads ONE small vector
        [accepted, max |g| before the step, relative cost drop, |step|, |x|, new sum r^2, old sum r^2, lambda used, ...]
    to follow the loop; x, lambda, the gain ratio and both states stay in HBM."""
    # For the duration of the loop the engine records neither the start / stop events of every build (pcs_last_kernel_ms) nor its
    # ordering event after every enqueue on this solver's stream (the stream lives as long as `ne`; "lazy_done_event" = 2 records
    # when somebody waits): each record is a packet between two launches, three of them cost a trial ~17 us (rocprofv3 trace).
    eng = ne.eng
    saved = (eng.option("timing_every", 1), eng.option("lazy_done_event", 1))
    eng.set_option("timing_every", 0)
    eng.set_option("lazy_done_event", 2)
    try:
        # The device steers the loop wherever the collective (if any) is stream-ordered: one GPU, or RCCL on the solver's stream.
        # A host-staged collective (gloo) needs the host between build and decision anyway: host-steered loop.
        # A sharded loop runs in the engine's deterministic mode — every rank computes the same bits from the all-reduced blocks, so
        # no rank can take another branch than its peers — unless the engine cannot (the (image, key) pass of the self chain with more
        # than 64 cameras keeps its atomics, csrc/ba_reduce.hpp): then the ranks adopt one consensus step per trial, host-steered.
        sharded = ne.reduce_fn is not None
        # (generated chains: the ordered contraction of csrc/ba_blockgram.hpp, unless two blocks share a parameter group)
        det_ok = not (eng.chain == "self" and eng.n_cams > DET_SELF_CAM_LIMIT) and getattr(eng, "deterministic_supported", lambda: True)()
        device_steered = not sharded or (getattr(ne.reduce_fn, "on_device", False) and det_ok)
        loop = _lm_loop_device if device_steered else _lm_loop_blocked
        saved_det = eng.option("deterministic", 0)
        if sharded and det_ok and not saved_det:
            eng.set_option("deterministic", 1)
        try:
            return loop(ne, ps0, max_iter=max_iter, ftol=ftol, xtol=xtol, gtol=gtol, lam0=lam0, lam_grow0=lam_grow0, verbose=verbose)
        finally:
            if sharded and det_ok and not saved_det:
                eng.set_option("deterministic", 0)
    finally:
        eng.set_option("lazy_done_event", saved[1])   # flushes the pending record while the stream exists
        eng.set_option("timing_every", saved[0])


STOP_MESSAGES = {0: "maximum number of iterations reached", 1: "gtol reached", 2: "no further decrease (damping exhausted)", 3: "ftol reached", 4: "xtol reached",
                 5: "maximum number of iterations reached"}
STOP_STATUS = {0: 0, 1: 1, 2: 2, 3: 3, 4: 4, 5: 0}
DET_SELF_CAM_LIMIT = 64          # the self chain's (image, key) pass is order-deterministic up to this many cameras (csrc/ba_reduce.hpp)
REJECTION_LIMIT = 12      # consecutive rejected trials before the loop gives up ("damping exhausted")
LM_SENTINEL = np.uint64(0x7FF8DEAD00000001)   # what a read-back slot holds until the device has written it (csrc/ba_schur.hpp lm_decide_kernel)


def _lm_loop_device(ne: BlockedNormalEquations, ps0: np.ndarray, *, max_iter, ftol, xtol, gtol, lam0, lam_grow0, verbose):
    """The loop the DEVICE steers: one trial = the step, the build at the trial string, the decision WITH the termination rules and
    the state flip of an accepted trial, a 12-double read-back into page-locked memory — and every kernel of it starts by reading a
    stop word.  The host therefore queues trial t + 1 BEFORE it reads the verdict of trial t: the GPU never idles between trials
    (38-53 us per trial on rig-32 in the host-steered form, profiles/r04/lm_trace_rig32.log), and when the loop ends the one
    speculative trial behind it drains as empty launches.

    Sharded over ranks with a stream-ordered collective (RCCL; ``reduce_fn.on_device``): the same loop with the all-reduce of the
    trial state queued between the two halves of a trial (pcs_lm_trial_build / pcs_lm_trial_finish) — no host synchronisation
    either.  The collective is queued on a fixed address, so the trial is always built into packed[1] and an accepted one is copied
    over packed[0] (PCS_LM_FIXED_TRIAL_BUFFER); the ranks decide on identical all-reduced blocks with order-deterministic kernels
    (`_lm_solve_blocked` switches the engine's deterministic mode on for a sharded loop), so every rank walks the same path without a consensus
    collective, and a dense solve that gives up on ONE rank voids the trial on all of them (PCS_LM_VOTES)."""
    from ._capi import LM_FIXED_TRIAL_BUFFER, LM_STATS, LM_VOTES, LmBuffers
    from .engine import SPD_ALGORITHMS

    torch = ne.torch
    dev = ne.dev
    sharded = ne.reduce_fn is not None
    eng = ne.eng
    with torch.cuda.device(dev), torch.cuda.stream(ne.stream):
        stream = ne.stream.cuda_stream
        ps = [torch.from_numpy(np.ascontiguousarray(ps0, dtype=np.float64)).to(dev), None]
        ps[1] = torch.empty_like(ps[0])
        lam = torch.full((1,), float(lam0), dtype=torch.float64, device=dev)
        ctrl = torch.tensor([0.0, 0.0, 0.0, float(max_iter), ftol, xtol, gtol, float(REJECTION_LIMIT), 0.0, float(lam_grow0), LAM_FAST[0], LAM_FAST[1]],
                            dtype=torch.float64, device=dev)
        flags = torch.zeros(4, dtype=torch.int32, device=dev)          # [stop, accepted, current state, -]
        stats_dev = torch.zeros(LM_STATS, dtype=torch.float64, device=dev)
        ring = 4
        # page-locked, allocated once per solver state, not per solve (hipHostMalloc costs ~0.2 ms) — and not shared between states: a
        # speculative trial one solve leaves behind still writes its read-back while the next solve may already run
        stats_host = ne.__dict__.get("_stats_host")
        if stats_host is None:
            stats_host = ne._stats_host = [torch.zeros(LM_STATS, dtype=torch.float64).pin_memory() for _ in range(ring)]
        # the final state (gradient | solution | sum r^2) is written into this page-locked buffer by the trial that ends the loop, before
        # that trial's read-back: the host returns without a copy of its own and without waiting for the speculative trial to drain
        n_free = int(ne.free_idx.numel())
        result_host = ne.__dict__.get("_result_host")
        if result_host is None or result_host.numel() != 2 * n_free + 1:
            result_host = ne._result_host = torch.zeros(2 * n_free + 1, dtype=torch.float64).pin_memory()
        result_view = result_host.numpy()
        result_view[-1] = np.nan
        pending = ne.__dict__.pop("_drain_event", None)
        if pending is not None:
            pending.synchronize()              # the speculative trial the previous solve left behind has written its (void) read-back
        ne.build(ps[0], 0)
        history = []                       # the first entry — the starting cost — comes with the first trial's read-back (no sync of its own)

        def buffers(k):
            b = LmBuffers()
            b.packed[0], b.packed[1] = ne.packed[0].data_ptr(), ne.packed[1].data_ptr()
            b.ps[0], b.ps[1] = ps[0].data_ptr(), ps[1].data_ptr()
            b.flags, b.fixed, b.lam = flags.data_ptr(), ne.fixed.data_ptr(), lam.data_ptr()
            b.linvt, b.u, b.V, b.S, b.rhs, b.dvec, b.gm = (t.data_ptr() for t in (ne.linvt, ne.u, ne.V, ne.S, ne.rhs, ne.dvec, ne.gm))
            b.status, b.xlead, b.w, b.spd_work = ne.status.data_ptr(), ne.xl.data_ptr(), ne.w.data_ptr(), ne.chol_work.data_ptr()
            b.delta, b.ctrl = ne.delta.data_ptr(), ctrl.data_ptr()
            b.stats, b.stats_host = stats_dev.data_ptr(), stats_host[k % ring].data_ptr()
            b.spd_algorithm = SPD_ALGORITHMS[ne.spd_algorithm]
            # a fixed trial buffer: where a collective is queued on it (sharded) or the build reads its string from a fixed address (generated chains)
            b.mode = (LM_FIXED_TRIAL_BUFFER | LM_VOTES) if sharded else LM_FIXED_TRIAL_BUFFER if getattr(eng, "lm_fixed_trial_buffer", False) else 0
            b.free_idx, b.n_free, b.result_host = ne.free_idx.data_ptr(), n_free, result_host.data_ptr()
            b.syrk_work, b.syrk_work_len = ne.syrk_work.data_ptr(), ne.syrk_work_len
            return b

        # The read-back of trial k lands in page-locked memory the device writes directly (lm_decide_kernel; all twelve words are -1 for
        # a launch that found the stop flag raised).  The host waits for THOSE words instead of an event — an event record between two
        # trials cost the GPU 5.6 us of idling per trial (the one gap in profiles/r04/lm_trace_*.log) — and the device needs no fence
        # between them: the host fills the slot with a NaN pattern no arithmetic produces and waits until none of it is left.
        views = [t.numpy() for t in stats_host]
        raw = [v.view(np.uint64) for v in views]

        def enqueue(k):
            raw[k % ring][:] = LM_SENTINEL
            b = buffers(k)
            if sharded:   # the trial state is all-reduced between the two halves, on this stream: nothing waits for the host
                eng.lm_trial_build(b, stream)
                ne.reduce(ne.packed[1])
                eng.lm_trial_finish(b, stream)
            else:
                eng.lm_trial(b, stream)

        def wait_for(k):
            v, u = views[k % ring], raw[k % ring]
            t_end = time.perf_counter() + 30.0
            spins = 0
            while (u == LM_SENTINEL).any():
                spins += 1
                if spins & 1023 == 0:
                    if time.perf_counter() > t_end:
                        raise RuntimeError("device LM loop: no read-back within 30 s")
                    time.sleep(0)
            return v.copy()

        code, nfev, n_lin, it = 0, 1, 0, 0
        limit = REJECTION_LIMIT * max_iter + 16          # every accepted step is preceded by fewer than REJECTION_LIMIT rejections
        queued = read = 0
        if max_iter > 0:
            enqueue(queued)
            queued += 1
        else:
            code = 5
        while code == 0 and read < limit:
            if queued < limit:                 # speculate: the next trial goes out before this one's verdict is read
                enqueue(queued)
                queued += 1
            st = wait_for(read)
            read += 1
            if st[9] < 0:                       # a launch that found the flag raised: nothing happened
                if read >= queued:
                    break
                continue
            n_lin += 1
            nfev += 1
            if not history:
                history.append(0.5 * float(st[6]))
            if verbose:
                print(f"  trial {int(st[9])}: lam {st[7]:.2e} cost {0.5 * st[6]:.6e} -> {0.5 * st[5]:.6e} accepted {bool(st[0] > 0)} stop {int(st[8])}")
            if st[0] > 0:
                it += 1
                history.append(0.5 * float(st[5]))
            code = int(st[8])
            if code == 9:   # the one-launch dense solve gave up waiting (on this rank or on a peer): repeat the trial with the launch-per-column form
                torch.cuda.current_stream().synchronize()      # whatever was queued behind it has drained as no-ops
                ne.spd_algorithm = "launches"
                ctrl[0] = 0.0
                flags[:2].zero_()                               # stop and accept; the current-state word stays
                nfev -= 1
                n_lin -= 1
                code = 0
                read = queued                                   # forget the drained launches
                if queued < limit:
                    enqueue(queued)
                    queued += 1
        if read < queued:                           # the speculative trial behind the end drains on its own (empty launches) ...
            ne._drain_event = torch.cuda.Event()    # ... and the next solve on this state waits for that before it reuses the read-back ring
            ne._drain_event.record()
        if code not in (0, 9) and max_iter > 0 and not np.isnan(result_view[-1]):
            out = result_view.copy()           # written by the trial that raised the stop code (lm_decide_kernel), complete before its read-back
        else:
            # gradient, solution and cost in ONE read-back (g and the cost sit behind the blocks of the packed state: [A | B | C | g | cost])
            torch.cuda.current_stream().synchronize()
            cur = int(flags[2].item())
            g0 = ne.n_packed - 1 - ne.n_params
            out = torch.cat([ne.packed[cur][g0: g0 + ne.n_params][ne.free_idx], ps[cur][ne.free_idx], ne.packed[cur][ne.n_packed - 1: ne.n_packed]]).cpu().numpy()
        g, x, cost = out[:n_free].copy(), out[n_free: 2 * n_free].copy(), 0.5 * float(out[-1])
        if not history:
            history.append(cost)
    return DeviceLMResult(x=x, cost=cost, grad=g, optimality=float(np.max(np.abs(g))) if g.size else 0.0, nit=it, nfev=nfev,
                          n_jtjv=n_lin, status=STOP_STATUS.get(code, 0), message=STOP_MESSAGES.get(code, f"stopped ({code})"), history=history)


def _gain_ratio(ne: BlockedNormalEquations, stats) -> float:
    """actual / predicted reduction of the trial `stats` describes (host-steered loop; the device-steered one has it in the kernel)."""
    torch = ne.torch
    lam_used = float(stats[7])
    pred = 0.5 * float((lam_used * torch.dot(ne.dvec, ne.delta * ne.delta) - torch.dot(ne.gm, ne.delta)).item())
    return 0.5 * (float(stats[6]) - float(stats[5])) / pred if pred > 0 else -1.0


def _lm_loop_blocked(ne: BlockedNormalEquations, ps0: np.ndarray, *, max_iter, ftol, xtol, gtol, lam0, lam_grow0, verbose):
    """The host-steered loop: what a sharded solve takes when its collective goes through the host (gloo).  Same rules as the
    device-steered loop (the decision itself is pcs_lm_decide on the device).  Every rank decides on the same all-reduced blocks;
    whether a rank's one-launch dense solve gave up is all-reduced with them (the word behind the packed state), so the ranks repeat
    a void trial TOGETHER.  `_lm_solve_blocked` runs the loop in the engine's deterministic mode, where the ranks compute identical
    steps; only where that mode is not available (self chain beyond 64 cameras) do they adopt one consensus step per trial."""
    torch = ne.torch
    dev = ne.dev
    deterministic = bool(ne.eng.option("deterministic", 0))
    with torch.cuda.device(dev), torch.cuda.stream(ne.stream):     # one real stream for torch operations and C-ABI kernels alike
        ps = torch.from_numpy(np.ascontiguousarray(ps0, dtype=np.float64)).to(dev)
        ps_new = torch.empty_like(ps)
        lam = torch.full((1,), float(lam0), dtype=torch.float64, device=dev)
        stats_dev = torch.zeros(12, dtype=torch.float64, device=dev)
        stats_host = torch.zeros(12, dtype=torch.float64).pin_memory()
        verdict = torch.cuda.Event()
        cur, new = 0, 1
        ne.build(ps, cur)
        sumsq = float(ne.cost(cur).item())
        history = [0.5 * sumsq]
        nfev, n_lin = 1, 0
        status, message = 0, "maximum number of iterations reached"
        it = 0
        any_accepted = False
        for it in range(1, max_iter + 1):
            accepted = False
            stop = False
            retry = 0
            while retry < REJECTION_LIMIT:   # damping retries
                ne.solve(cur, lam, ps, ps_new)
                if ne.reduce_fn is not None:
                    if not deterministic:
                        ne._consensus_step(ps, ps_new)
                    ne.packed[new][ne.n_packed] = (ne.status[0] & 4).to(torch.float64)     # this rank's vote: "my dense solve gave up"
                n_lin += 1
                ne.build(ps_new, new)                                                        # the all-reduce sums the votes with the blocks
                if ne.reduce_fn is not None:
                    ne.status.bitwise_or_((ne.packed[new][ne.n_packed: ne.n_packed + 1] > 0).to(torch.int32) * 4)
                nfev += 1
                ne.decide(cur, new, ps, lam, stats_dev)
                stats_host.copy_(stats_dev, non_blocking=True)   # the ONE read-back of the trial: 96 bytes into page-locked memory
                verdict.record()
                verdict.synchronize()
                stats = stats_host.numpy().copy()
                if stats[0] < 0:   # the one-launch dense solve gave up waiting on SOME rank (status bit 2): nothing of this trial is valid —
                    ne.spd_algorithm = "launches"   # every rank repeats it with the launch-per-column form (the decision left the damping alone)
                    nfev -= 1
                    n_lin -= 1
                    continue
                retry += 1
                gmax = float(stats[1])
                if verbose:
                    print(f"  it {it}: lam {stats[7]:.2e} cost {0.5 * stats[6]:.6e} -> {0.5 * stats[5]:.6e} accepted {bool(stats[0])}")
                if gmax <= gtol:   # the state BEFORE this step was already stationary: the step is dropped
                    status, message, stop = 1, "gtol reached", True
                    break
                if stats[0] > 0:
                    accepted = any_accepted = True
                    # pcs_lm_decide applied the classic 1/3; a gain ratio above LAM_FAST[0] earns LAM_FAST[1] (the device-steered loop's rule)
                    if _gain_ratio(ne, stats) > LAM_FAST[0]:
                        lam.fill_(max(float(stats[7]) * LAM_FAST[1], 1e-12))
                    ps, ps_new = ps_new, ps
                    cur, new = new, cur
                    history.append(0.5 * float(stats[5]))
                    rel_drop, step_norm, x_norm = float(stats[2]), float(stats[3]), float(stats[4])
                    break
                if not any_accepted and lam_grow0 > 1.0:   # a rejection before the first accepted step: the device rule multiplied by 4
                    lam.mul_(lam_grow0 / 4.0)
            if stop:
                break
            if not accepted:
                status, message = 2, "no further decrease (damping exhausted)"
                break
            if rel_drop <= ftol:
                status, message = 3, "ftol reached"
                break
            if step_norm <= xtol * (xtol + x_norm):
                status, message = 4, "xtol reached"
                break
        g = ne.gradient(cur, lam)
        x = ps[ne.free_idx].cpu().numpy()
        cost = 0.5 * float(ne.cost(cur).item())
    return DeviceLMResult(x=x, cost=cost, grad=g, optimality=float(np.max(np.abs(g))) if g.size else 0.0, nit=it, nfev=nfev,
                          n_jtjv=n_lin, status=status, message=message, history=history)


@dataclass
class DeviceLMResult:
    x: np.ndarray
    cost: float                 # 0.5 * sum r^2, like scipy's OptimizeResult.cost
    grad: np.ndarray
    optimality: float
    nit: int
    nfev: int
    n_jtjv: int                 # matrix-free J^T J v products ("pcg") / factorisations ("cholesky")
    status: int
    message: str
    history: list = field(default_factory=list)


class _PcgStep:
    """LM linear algebra from matrix-free products (JacobianOperator)."""

    def __init__(self, op, cg_tol, cg_max_iter):
        self.op, self.cg_tol, self.cg_max_iter = op, cg_tol, cg_max_iter

    def evaluate(self, ps, need_scale=True):
        self.op.linearize(ps)
        g, sumsq = self.op.grad()
        return {"ps": ps, "g": g, "sumsq": sumsq, "d": None}

    def scale(self, st):
        if st["d"] is None:
            st["d"] = np.maximum(self.op.diag(), 1e-300)   # refers to the current linearisation = st
        return st["d"]

    def restore(self, st):
        self.op.linearize(st["ps"])   # the slabs hold a rejected trial point

    def solve(self, st, lam):
        dd = self.scale(st)
        delta, k = pcg(lambda v: self.op.jtjv(v) + lam * dd * v, -st["g"], 1.0 / ((1.0 + lam) * dd), self.cg_tol, self.cg_max_iter)
        return delta, k, 0.5 * float(-(st["g"] @ delta) + lam * (delta @ (dd * delta)))


class _CholeskyStep:
    """LM linear algebra from the block-reduced normal equations (NormalEquations)."""

    def __init__(self, ne):
        self.ne = ne

    def evaluate(self, ps, need_scale=True):
        Hs, g, sumsq = self.ne.build(ps)
        return {"ps": ps, "H": Hs, "g_dev": g, "g": g.cpu().numpy(), "sumsq": sumsq, "d": None}

    def scale(self, st):
        if st["d"] is None:
            import torch

            st["d"] = torch.clamp(torch.diagonal(st["H"]).clone(), min=1e-300)
        return st["d"]

    def restore(self, st):
        pass   # the state carries H: nothing refers to the engine's slabs

    def solve(self, st, lam):
        dd = self.scale(st)
        delta = self.ne.solve(st["H"], st["g_dev"], lam, dd)
        if delta is None:
            return None, 1, 0.0
        pred = 0.5 * float((-(st["g_dev"] @ delta) + lam * (delta @ (dd * delta))).item())
        return delta.cpu().numpy(), 1, pred


def lm_solve(handler, x0, *, max_iter: int = 50, ftol: float = 1e-8, xtol: float = 1e-8, gtol: float = 1e-8,
             cg_tol: float = 1e-3, cg_max_iter: int = 200, lam0: float | None = None, lam_grow0: float | None = None, reduce_fn=None, verbose: int = 0,
             operator=None, linear_solver: str = "auto") -> DeviceLMResult:
    """Levenberg-Marquardt (Marquardt scaling D = diag(J^T J)) for a pycamset_amd handler.  The damped
    normal equations are solved by Jacobi-PCG on matrix-free J^T J products (``linear_solver='pcg'``) or
    by a Cholesky factorisation of the block-reduced J^T J (``'cholesky'``).  Every quantity that depends
    on the detections is computed by the HIP engine.  ``operator`` replaces the engine-backed
    JacobianOperator (used by the CPU tests of this driver).

    ``lam0``: the initial damping (Nielsen's tau: lambda multiplies D).  ``None`` = LAM0_EXACT = 1e-5 with the exact (Cholesky)
    step, 1e-3 with PCG.  The reference's solver — scipy ``least_squares(method='trf')``, optimisation_handling.py:88-98 — starts with
    the plain Gauss-Newton step whenever that lies inside its first trust region, which it does from a calibration's starting values;
    round 3's 1e-3 with the update lambda <- lambda / 3 needs nine accepted steps on rig-32 to get the damping out of the way.  The
    policy since round 5 (DESIGN section 4 has the table: near starts, 5 x / 10 x farther starts, two poses swapped, chains T and S):
    start at 1e-5 and let an ACCURATE model shed damping fast — a gain ratio above LAM_FAST[0] = 0.95 multiplies lambda by 0.1, above
    0.75 by 1/3, above 0.25 by 1, below by 2; every rejected trial multiplies it by 4 (host and device rule alike; at most
    REJECTION_LIMIT = 12 in a row), a rejection BEFORE the first accepted step by ``lam_grow0`` (default LAM_GROW0 = 1e3).  1e-6 (round
    4) is one evaluation faster from a near start and up to twice as slow from a far one: a nearly undamped step from 70 px away
    overshoots into a region where five to eight trials are rejected in a row.  ``max_iter <= 0`` evaluates the start and returns it."""
    if linear_solver not in ("auto", "pcg", "cholesky"):
        raise ValueError("linear_solver must be 'auto', 'pcg' or 'cholesky'")
    op_fun = handler.op_fun
    lam0_exact, lam0_pcg = (LAM0_EXACT, 1e-3) if lam0 is None else (float(lam0), float(lam0))
    grow0 = LAM_GROW0 if lam_grow0 is None else float(lam_grow0)
    if operator is None:
        dd = handler._flat_detections()
        eng = op_fun._engine_for(dd)
        op_fun._bind_template(eng, handler._template_arg())
        if linear_solver == "auto":   # blocked J^T J while its regions fit comfortably; beyond that matrix-free CG
            linear_solver = "cholesky" if blocked_fits(eng) else "pcg"
        if linear_solver == "cholesky":
            # the solver's device workspace (two packed states, V, S, a stream) lives with the engine: a second solve on the same
            # table and mask — the usual case: a calibration re-run with other start values or tolerances — allocates nothing
            mask = np.asarray(handler._jac_mask(), dtype=bool)
            # (... and layout: a generated chain changes it with set_option("dense_normal", ...) — buffers sized for the other form would be overrun)
            key = (hash(mask.tobytes()), id(reduce_fn), tuple(sorted(eng.normal_layout().items())))
            cache = eng.__dict__.setdefault("_blocked_solvers", {})
            ne = cache.get(key)
            if ne is None:
                for old in cache.values():   # a speculative trial of the solver state being dropped may still be draining: its buffers go back to the allocator after that
                    old.stream.synchronize()
                cache.clear()
                ne = cache[key] = BlockedNormalEquations(eng, mask, reduce_fn=reduce_fn)
            ne.spd_algorithm = "auto"
            ps0 = op_fun.build_param_list(*handler.get_bundle_adjustment_inputs(np.array(x0, dtype=np.float64)))
            return _lm_solve_blocked(ne, ps0, max_iter=max_iter, ftol=ftol, xtol=xtol, gtol=gtol, lam0=lam0_exact, lam_grow0=grow0, verbose=verbose)
        operator = JacobianOperator(eng, handler._jac_mask(), reduce_fn=reduce_fn)
    elif linear_solver == "auto":
        linear_solver = "cholesky" if hasattr(operator, "build") else "pcg"
    step = _PcgStep(operator, cg_tol, cg_max_iter) if linear_solver == "pcg" else _CholeskyStep(operator)

    def param_str(x):
        return op_fun.build_param_list(*handler.get_bundle_adjustment_inputs(x))

    x = np.array(x0, dtype=np.float64)
    st = step.evaluate(param_str(x))
    nfev, n_lin, lam = 1, 0, (lam0_pcg if linear_solver == "pcg" else lam0_exact)
    history = [0.5 * st["sumsq"]]
    status, message = 0, "maximum number of iterations reached"
    it = 0
    any_accepted = False
    for it in range(1, max_iter + 1):
        if float(np.max(np.abs(st["g"]))) <= gtol:
            status, message = 1, "gtol reached"
            break
        accepted = False
        for retry in range(REJECTION_LIMIT):  # damping retries
            if retry:
                step.restore(st)
            delta, k, pred = step.solve(st, lam)
            n_lin += k
            if delta is None:   # damped matrix not positive definite: more damping
                lam *= 4.0 if any_accepted else max(grow0, 4.0)
                continue
            x_new = x + delta
            st_new = step.evaluate(param_str(x_new))
            nfev += 1
            actual = 0.5 * (st["sumsq"] - st_new["sumsq"])
            rho = actual / pred if pred > 0 else -1.0
            if verbose:
                print(f"  it {it}: lam {lam:.2e} lin {k} cost {0.5 * st['sumsq']:.6e} -> {0.5 * st_new['sumsq']:.6e} rho {rho:.3f}")
            if np.isfinite(st_new["sumsq"]) and actual > 0:
                accepted = any_accepted = True
                step_norm, x_norm = float(np.linalg.norm(delta)), float(np.linalg.norm(x))
                rel_drop = actual / (0.5 * st["sumsq"])
                x, st = x_new, st_new
                fast = linear_solver != "pcg" and rho > LAM_FAST[0]    # exact steps only: less damping costs an inexact (CG) step iterations
                lam = max(lam * (LAM_FAST[1] if fast else 1.0 / 3.0 if rho > 0.75 else 1.0 if rho > 0.25 else 2.0), 1e-12)
                history.append(0.5 * st["sumsq"])
                break
            lam *= 4.0 if any_accepted else max(grow0, 4.0)
        if not accepted:
            step.restore(st)
            status, message = 2, "no further decrease (damping exhausted)"
            break
        if rel_drop <= ftol:
            status, message = 3, "ftol reached"
            break
        if step_norm <= xtol * (xtol + x_norm):
            status, message = 4, "xtol reached"
            break
    return DeviceLMResult(x=x, cost=0.5 * st["sumsq"], grad=st["g"], optimality=float(np.max(np.abs(st["g"]))), nit=it, nfev=nfev,
                          n_jtjv=n_lin, status=status, message=message, history=history)
