"""``Engine`` — thin Python owner of one ``pcs_engine`` handle (include/pcs_hip.h).

NumPy in / NumPy out for the drop-in closures; raw device pointers (e.g. ``tensor.data_ptr()``)
for callers that keep the residual / Jacobian in HBM (bench.py, the sharded multi-GPU path).
All arithmetic happens in the HIP kernels behind the C ABI; nothing here computes.
"""
from __future__ import annotations

import ctypes
import weakref
from ctypes import POINTER, byref, c_double, c_float, c_int32, c_int64, c_uint8, c_void_p

import numpy as np

from . import _capi
from ._capi import CHAIN_IDS, CHAIN_P, DTYPE_IDS, check, lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(POINTER(c_double))


def _f64c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


class _PinnedBlock:
    """Owner of one page-locked host allocation; NumPy views keep it alive through ``base``."""

    def __init__(self, nbytes: int):
        self.ptr = c_void_p()
        check(lib().pcs_host_alloc(byref(self.ptr), int(nbytes)))
        self.nbytes = nbytes

    def __del__(self):
        try:
            if self.ptr:
                lib().pcs_host_free(self.ptr)
                self.ptr = c_void_p()
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64) -> np.ndarray:
    """``np.empty`` in page-locked host memory (freed when the last view dies)."""
    shape = tuple(int(x) for x in np.atleast_1d(shape))
    n = int(np.prod(shape)) if shape else 1
    nbytes = max(1, n) * np.dtype(dtype).itemsize
    block = _PinnedBlock(nbytes)
    buf = (ctypes.c_char * nbytes).from_address(block.ptr.value)
    buf._pcs_owner = block  # ctypes object keeps the owner; the ndarray keeps the ctypes object
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


class _PinnedSlot:
    """One page-locked block of an output ring and the LEASE on it (round 5; rounds 3-4 counted references with
    ``sys.getrefcount``).  ``lease()`` hands the block out as a fresh NumPy array; NumPy hangs every view derived from that array —
    ``a.reshape(-1)``, slices, a ``csr_array`` built on one, ``torch.from_numpy(a)``, a ``memoryview(a)`` — on ONE root array (the
    ``base`` chain collapses to it), and a ``weakref.finalize`` on that root ends the lease when the last of them has gone.  The ring
    keeps no reference to the root, so nothing about frames, temporaries or interpreter versions enters the decision; a holder that
    keeps only a raw address (``a.ctypes.data`` handed to a C library or queued on a stream) must keep the array too, as with any
    NumPy buffer — or call ``release`` semantics explicitly by dropping it when the consumer is done."""

    def __init__(self, shape, dtype=np.float64):
        self.shape = tuple(int(x) for x in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.count = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = max(1, self.count) * self.dtype.itemsize
        self.block = _PinnedBlock(self.nbytes)
        self.leased = False

    def _end(self):
        self.leased = False

    def lease(self) -> np.ndarray:
        buf = (ctypes.c_char * self.nbytes).from_address(self.block.ptr.value)
        buf._pcs_owner = self.block          # the memory outlives the ring (and the engine) for as long as somebody holds the array
        root = np.frombuffer(buf, dtype=self.dtype, count=self.count)
        self.leased = True
        weakref.finalize(root, self._end)
        return root.reshape(self.shape)      # a view: its base is `root`, like every view the caller derives from it


def _stream_arg(stream):
    """HIP stream handle for the C ABI.  ``None`` = the engine's own (non-blocking) stream.  An integer is
    a ``hipStream_t``; 0 — what ``torch.cuda.current_stream().cuda_stream`` returns for torch's default
    stream — means the process's default (NULL) stream, which the ABI spells ``hipStreamLegacy`` because a
    NULL argument already selects the engine stream."""
    if stream is None:
        return c_void_p(0)
    return c_void_p(1 if stream == 0 else stream)


class Engine:
    """One chain ('template' | 'self' | 'free') on one device.

    Counts follow the reference's ``make_param_struct`` (abstract_function_blocks.py:793-795):
    callers that mirror the reference pass ``max index + 1`` of the detection table.
    """

    def __init__(self, chain: str, n_cams: int, n_imgs: int, n_keys: int, *, dtype: str = "f64", device: int = 0):
        if chain not in CHAIN_IDS:
            raise ValueError(f"chain must be one of {sorted(CHAIN_IDS)}")
        if dtype not in DTYPE_IDS:
            raise ValueError("dtype must be 'f64', 'f32' or 'mixed'")
        self.chain, self.dtype, self.device = chain, dtype, device
        self.n_cams, self.n_imgs, self.n_keys = int(n_cams), (0 if chain == "free" else int(n_imgs)), int(n_keys)
        self.P = CHAIN_P[chain]
        self._h = c_void_p()
        check(lib().pcs_create(byref(self._h), CHAIN_IDS[chain], DTYPE_IDS[dtype], int(n_cams), int(n_imgs), int(n_keys), int(device)))
        self.n_params = int(lib().pcs_n_params(self._h))
        self.n = 0
        self.nnz = None
        self.mask_key = None
        self._rings = {}
        self.np_dtype = np.float64 if dtype == "f64" else np.float32   # element type of the DEVICE outputs

    # -- lifetime -----------------------------------------------------------------------------
    def close(self):
        self._rings = {}
        self.__dict__.pop("_blocked_solvers", None)   # device_solver's cached workspace refers to this engine
        if getattr(self, "_h", None) is not None and self._h:
            lib().pcs_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- static inputs ------------------------------------------------------------------------
    def set_detections_table(self, det5: np.ndarray):
        """(N,5) float64 [cam, im, key, u, v] — the reference's flattened detection table."""
        det5 = _f64c(det5)
        if det5.ndim != 2 or det5.shape[1] != 5:
            raise ValueError("detections must have shape (N, 5)")
        check(lib().pcs_set_detections_table(self._h, _dp(det5), det5.shape[0]))
        self.n = det5.shape[0]
        self.nnz = None

    def set_detections(self, cam, img, key, uv):
        cam = np.ascontiguousarray(cam, dtype=np.int32)
        img = np.ascontiguousarray(img, dtype=np.int32)
        key = np.ascontiguousarray(key, dtype=np.int32)
        uv = _f64c(uv)
        n = cam.shape[0]
        if not (img.shape[0] == key.shape[0] == n and uv.shape == (n, 2)):
            raise ValueError("cam/img/key must have length N and uv shape (N,2)")
        i32 = POINTER(c_int32)
        check(lib().pcs_set_detections(self._h, cam.ctypes.data_as(i32), img.ctypes.data_as(i32), key.ctypes.data_as(i32), _dp(uv), n))
        self.n = n
        self.nnz = None

    def set_template(self, points: np.ndarray):
        points = _f64c(points).reshape(-1, 3)
        if points.shape[0] < self.n_keys:
            raise ValueError(f"template has {points.shape[0]} points, the engine indexes {self.n_keys} keys")
        check(lib().pcs_set_template(self._h, _dp(points)))

    def set_option(self, key: str, value: int):
        check(lib().pcs_set_option(self._h, key.encode(), int(value)))
        self.__dict__.setdefault("_options", {})[key] = int(value)

    def option(self, key: str, default: int) -> int:
        """The value last given to ``set_option(key, ...)`` through this object, else ``default`` (the library's own default)."""
        return self.__dict__.get("_options", {}).get(key, default)

    # -- evaluation: host buffers -------------------------------------------------------------
    def _check_params(self, param_str) -> np.ndarray:
        p = _f64c(param_str).ravel()
        if p.shape[0] != self.n_params:
            raise ValueError(f"parameter string has {p.shape[0]} entries, engine expects {self.n_params}")
        return p

    def _out(self, name: str, shape, pinned_ring: int):
        """Output array: fresh pageable memory, or the next block of a ring of ``pinned_ring`` page-locked blocks.  A block is only
        handed out again once its LEASE has ended (``_PinnedSlot``): every array handed out stays valid for as long as it — or any
        view NumPy derives from it: ``j.reshape(-1)``, slices, a ``scipy.sparse.csr_array`` built on it, ``torch.from_numpy(j)``, a
        ``memoryview(j)`` — is held (tests/test_gpu_parity.py covers these): reference semantics either way, like the reference's
        fresh array per call (afb:561).  What a lease cannot see is a raw address taken from the array (``j.ctypes.data``, a pointer
        handed to a C library or queued on a stream): keep the array itself alive for as long as such a pointer is in use."""
        if pinned_ring <= 0:
            return np.empty(shape)
        ring = self._rings.setdefault((name, tuple(shape), pinned_ring), {"slots": [], "count": 0})
        idx = ring["count"] % pinned_ring
        ring["count"] += 1
        if idx >= len(ring["slots"]):
            ring["slots"].append(_PinnedSlot(shape))
        elif ring["slots"][idx].leased:
            # Somebody still holds this block: a kept Jacobian must never change under its owner.  The slot gets a new page-locked
            # block; the old one lives on with its holder and is freed with it.
            ring["slots"][idx] = _PinnedSlot(shape)
        return ring["slots"][idx].lease()

    def eval(self, param_str, want_resid: bool = True, want_jac: bool = True, pinned_ring: int = 0):
        """-> (resid (N,2) | None, jac (2N,P) | None), float64 NumPy."""
        p = self._check_params(param_str)
        r = self._out("resid", (self.n, 2), pinned_ring) if want_resid else None
        j = self._out("jac", (2 * self.n, self.P), pinned_ring) if want_jac else None
        check(lib().pcs_eval(self._h, _dp(p), _dp(r) if want_resid else None, _dp(j) if want_jac else None))
        return r, j

    def set_unfixed(self, unfixed) -> int:
        m = None if unfixed is None else np.ascontiguousarray(unfixed, dtype=np.uint8)
        if m is not None and m.shape[0] != self.n_params:
            raise ValueError("unfixed mask must have one entry per parameter")
        nnz = c_int64()
        check(lib().pcs_set_unfixed(self._h, m.ctypes.data_as(POINTER(c_uint8)) if m is not None else None, byref(nnz)))
        self.nnz = int(nnz.value)
        self.mask_key = None if m is None else hash(m.astype(bool).tobytes())
        return self.nnz

    def eval_compact(self, param_str, want_resid: bool = False, pinned_ring: int = 0):
        """-> (resid | None, data (nnz,)) with the fixed columns removed on the device."""
        if self.nnz is None:
            raise RuntimeError("call set_unfixed() first")
        p = self._check_params(param_str)
        r = self._out("resid", (self.n, 2), pinned_ring) if want_resid else None
        d = self._out("data", (self.nnz,), pinned_ring)
        check(lib().pcs_eval_compact(self._h, _dp(p), _dp(r) if want_resid else None, _dp(d)))
        return r, d

    # -- legacy residual-only cost (compiled_helpers.py:518-549) ----------------------------------------
    def legacy_cost(self, im_points, projection_matrixes, intrinsics, dists) -> np.ndarray:
        im = _f64c(im_points)
        P, K, D = _f64c(projection_matrixes), _f64c(intrinsics), _f64c(dists)
        if im.size != 3 * im.shape[0] * (im.size // (3 * im.shape[0])) or im.shape[-1] != 3:
            raise ValueError("im_points must have shape (n_imgs, ..., 3)")
        n_cams = P.shape[0]
        if P.shape[1:] != (3, 4) or K.shape != (n_cams, 3, 3) or D.reshape(n_cams, -1).shape[1] != 5:
            raise ValueError("expected proj (C,3,4), intrinsics (C,3,3), dists (C,5)")
        out = np.empty(2 * self.n)
        check(lib().pcs_legacy_cost(self._h, _dp(im), _dp(P), _dp(K), _dp(D), _dp(out)))
        return out

    # -- matrix-free Jacobian products (J never materialised) -------------------------------------
    OP_JV, OP_JTU, OP_JTJV, OP_DIAG, OP_GRAD = 0, 1, 2, 3, 4

    def linearize(self, param_str):
        """Prepare the slabs at ``param_str``; the products below refer to this point."""
        check(lib().pcs_linearize(self._h, _dp(self._check_params(param_str))))

    def _matfree(self, op: int, vin, n_out: int, want_cost: bool = False):
        out = np.empty(n_out)
        cost = c_double(0.0)
        vin_p = _dp(vin) if vin is not None else None
        check(lib().pcs_matfree(self._h, op, vin_p, _dp(out), byref(cost) if want_cost else None))
        return (out, float(cost.value)) if want_cost else out

    def jv(self, v) -> np.ndarray:
        """J v, shape (2N,); ``v`` in the full parameter-string space."""
        return self._matfree(self.OP_JV, self._check_params(v), 2 * self.n)

    def jtu(self, u) -> np.ndarray:
        u = _f64c(u).ravel()
        if u.shape[0] != 2 * self.n:
            raise ValueError("u must have 2N entries")
        return self._matfree(self.OP_JTU, u, self.n_params)

    def jtjv(self, v) -> np.ndarray:
        """J^T (J v) in one pass over the detections."""
        return self._matfree(self.OP_JTJV, self._check_params(v), self.n_params)

    def jtj_diag(self) -> np.ndarray:
        return self._matfree(self.OP_DIAG, None, self.n_params)

    def grad(self) -> tuple[np.ndarray, float]:
        """(J^T r, sum r^2) at the linearisation point."""
        return self._matfree(self.OP_GRAD, None, self.n_params, want_cost=True)

    # -- block-reduced normal equations (J never materialised) --------------------------------------
    def normal_equations(self, param_str, symmetric: bool = True):
        """(H = J^T J (n_params, n_params), g = J^T r, cost = r^T r) at ``param_str``, built on the GPU in
        one pass.  The kernel writes the upper triangle; ``symmetric`` mirrors it on the host."""
        p = self._check_params(param_str)
        H = np.empty((self.n_params, self.n_params))
        g = np.empty(self.n_params)
        cost = c_double(0.0)
        check(lib().pcs_normal_equations(self._h, _dp(p), _dp(H), _dp(g), byref(cost)))
        if symmetric:
            H = H + np.triu(H, 1).T
        return H, g, float(cost.value)

    def normal_equations_device(self, param_str, d_H: int, d_g: int, d_cost: int, stream: int | None = None):
        """Asynchronous; raw device addresses of float64 buffers (n_params^2, n_params, 1), zeroed by the call.
        Only the upper triangle of H is written."""
        p = self._check_params(param_str)
        check(lib().pcs_normal_equations_device(self._h, _dp(p), c_void_p(d_H), c_void_p(d_g), c_void_p(d_cost), _stream_arg(stream)))

    # -- blocked normal equations + the block parts of a damped step (device pointers only) -------------
    def normal_layout(self) -> dict:
        """{n_lead, n_trail, tb, packed_len, n_params}: see ``pcs_normal_blocks_device`` (include/pcs_hip.h)."""
        out = (c_int64 * 5)()
        check(lib().pcs_normal_layout(self._h, out))
        return dict(n_lead=int(out[0]), n_trail=int(out[1]), tb=int(out[2]), packed_len=int(out[3]), n_params=int(out[4]))

    def normal_blocks_device(self, d_param_str: int, d_packed: int, stream: int | None = None):
        """[A | B | C | g | cost] at the DEVICE-resident parameter string; asynchronous, zeroes the buffer first."""
        check(lib().pcs_normal_blocks_device(self._h, c_void_p(d_param_str), c_void_p(d_packed), _stream_arg(stream)))

    def schur_prepare(self, d_packed, d_fixed, d_lambda, d_linvt, d_u, d_V, d_S, d_rhs, d_dvec, d_gm, d_status, stream=None):
        check(lib().pcs_schur_prepare(self._h, *(c_void_p(p) for p in (d_packed, d_fixed, d_lambda, d_linvt, d_u, d_V, d_S, d_rhs, d_dvec, d_gm, d_status)),
                                      _stream_arg(stream)))

    def schur_finish(self, d_linvt, d_u, d_w, d_xlead, d_fixed, d_delta, d_ps_in=0, d_ps_out=0, stream=None):
        check(lib().pcs_schur_finish(self._h, *(c_void_p(p) for p in (d_linvt, d_u, d_w, d_xlead, d_fixed, d_delta, d_ps_in, d_ps_out)), _stream_arg(stream)))

    def lm_decide(self, d_cost_old, d_cost_new, d_dvec, d_gm, d_delta, d_ps, d_fixed, d_status, d_lambda, d_stats, stream=None):
        check(lib().pcs_lm_decide(self._h, *(c_void_p(p) for p in (d_cost_old, d_cost_new, d_dvec, d_gm, d_delta, d_ps, d_fixed, d_status, d_lambda, d_stats)),
                                  _stream_arg(stream)))

    def lm_trial(self, buffers, stream=None):
        """One whole LM trial (step, build at the trial string, decision with the termination rules and the state flip, read-back) in
        one call; ``buffers`` = a filled ``_capi.LmBuffers`` (include/pcs_hip.h pcs_lm_buffers)."""
        check(lib().pcs_lm_trial(self._h, ctypes.byref(buffers), _stream_arg(stream)))

    def lm_trial_build(self, buffers, stream=None):
        """First half of a trial: the damped step and the normal equations at the trial string (a sharded loop all-reduces the
        trial state after this, on the same stream)."""
        check(lib().pcs_lm_trial_build(self._h, ctypes.byref(buffers), _stream_arg(stream)))

    def lm_trial_finish(self, buffers, stream=None):
        """Second half: decision, termination rules, state flip (or copy, PCS_LM_FIXED_TRIAL_BUFFER), read-back."""
        check(lib().pcs_lm_trial_finish(self._h, ctypes.byref(buffers), _stream_arg(stream)))

    # -- evaluation: device buffers -----------------------------------------------------------
    def eval_device(self, param_str, d_resid: int | None, d_jac: int | None, stream: int | None = None):
        """Asynchronous; ``d_resid`` / ``d_jac`` are raw device addresses in the engine dtype."""
        p = self._check_params(param_str)
        check(lib().pcs_eval_device(self._h, _dp(p), c_void_p(d_resid or 0), c_void_p(d_jac or 0), _stream_arg(stream)))

    def eval_device_resident(self, d_param_str: int, d_resid: int | None, d_jac: int | None, stream: int | None = None):
        """Parameter string already in HBM (float64, n_params entries)."""
        check(lib().pcs_eval_device_resident(self._h, c_void_p(d_param_str), c_void_p(d_resid or 0), c_void_p(d_jac or 0), _stream_arg(stream)))

    def eval_compact_device(self, param_str, d_resid: int | None, d_data: int | None, stream: int | None = None):
        p = self._check_params(param_str)
        check(lib().pcs_eval_compact_device(self._h, _dp(p), c_void_p(d_resid or 0), c_void_p(d_data or 0), _stream_arg(stream)))

    def synchronize(self, stream: int | None = None):
        check(lib().pcs_synchronize(self._h, _stream_arg(stream)))

    def last_kernel_ms(self) -> tuple[float, float]:
        """(slab_prep ms, eval kernel ms) of the most recent evaluation, from HIP events on its stream."""
        a, b = c_float(), c_float()
        check(lib().pcs_last_kernel_ms(self._h, byref(a), byref(b)))
        return float(a.value), float(b.value)

    def kernel_ms_mean(self) -> tuple[int, float, float]:
        """(count, mean slab_prep ms, mean eval ms) over the evaluations kept in the event ring
        (``set_option('event_ring', R)``)."""
        n, a, b = c_int64(), c_float(), c_float()
        check(lib().pcs_kernel_ms_mean(self._h, byref(n), byref(a), byref(b)))
        return int(n.value), float(a.value), float(b.value)

    def kernel_ms_samples(self, capacity: int = 4096) -> tuple[np.ndarray, np.ndarray]:
        """(slab_prep ms, eval ms) of each evaluation kept in the event ring, oldest first."""
        a, b = np.empty(capacity, dtype=np.float32), np.empty(capacity, dtype=np.float32)
        n = c_int64()
        fp = POINTER(c_float)
        check(lib().pcs_kernel_ms_samples(self._h, capacity, a.ctypes.data_as(fp), b.ctypes.data_as(fp), byref(n)))
        return a[: n.value].astype(np.float64), b[: n.value].astype(np.float64)

    def device_buffers(self) -> tuple[int, int]:
        r, j = c_void_p(), c_void_p()
        check(lib().pcs_device_buffers(self._h, byref(r), byref(j)))
        return int(r.value), int(j.value)

    # -- static structure ---------------------------------------------------------------------
    def csr_structure(self, unfixed=None):
        """(indices int64 (nnz,), indptr int64 (2N+1,)) — abstract_function_blocks.py:465-489."""
        m = None if unfixed is None else np.ascontiguousarray(unfixed, dtype=np.uint8)
        mp = m.ctypes.data_as(POINTER(c_uint8)) if m is not None else None
        nnz = c_int64()
        check(lib().pcs_csr_structure(self._h, mp, None, None, byref(nnz)))
        indices = np.empty(nnz.value, dtype=np.int64)
        indptr = np.empty(2 * self.n + 1, dtype=np.int64)
        i64 = POINTER(c_int64)
        check(lib().pcs_csr_structure(self._h, mp, indices.ctypes.data_as(i64), indptr.ctypes.data_as(i64), byref(nnz)))
        return indices, indptr

    def block_param_inds(self) -> np.ndarray:
        out = np.empty((self.n, self.P), dtype=np.int64)
        check(lib().pcs_block_param_inds(self._h, out.ctypes.data_as(POINTER(c_int64))))
        return out


SPD_ALGORITHMS = {"auto": 0, "launches": 1, "one_launch": 2}   # include/pcs_hip.h PCS_SPD_*


def dense_spd_solve(device: int, n: int, d_S: int, ld: int, d_rhs: int, d_x: int, d_work: int, d_status: int, stream: int | None = None,
                    algorithm: str = "auto", timeout_us: int | None = None):
    """S x = rhs on the device (raw float64 device addresses, the lower triangle of S becomes its Cholesky factor).
    ``algorithm``: 'one_launch' = the persistent kernel of csrc/ba_chol_persist.hpp, 'launches' = one launch per block column
    (csrc/ba_dense_chol.hpp), 'auto' = the first where it fits.  ``d_work``: ``dense_spd_work_len(n)`` doubles; status bit 1
    (value 2): a pivot was not positive, bit 2 (value 4): the one-launch form gave up waiting (``timeout_us`` per wait, default
    250 000: the launch drains, S is partly overwritten — solve again from the original matrix with 'launches').  ``stream=None``
    queues on the default stream."""
    s = c_void_p(0) if stream is None else _stream_arg(stream)
    if timeout_us is not None:
        check(lib().pcs_dense_spd_solve_opts(int(device), int(n), c_void_p(d_S), int(ld), c_void_p(d_rhs), c_void_p(d_x), c_void_p(d_work), c_void_p(d_status), s,
                                             SPD_ALGORITHMS[algorithm], int(timeout_us)))
        return
    check(lib().pcs_dense_spd_solve_algo(int(device), int(n), c_void_p(d_S), int(ld), c_void_p(d_rhs), c_void_p(d_x), c_void_p(d_work), c_void_p(d_status), s,
                                         SPD_ALGORITHMS[algorithm]))


def schur_syrk(device: int, n_lead: int, n_trail: int, d_V: int, ldv: int, d_S: int, lds: int, d_u: int | None, d_rhs: int | None, stream: int | None = None,
               work: int | None = None, work_len: int = 0):
    """S -= V V' (lower triangle) and, with ``d_u``, rhs += V u on the device (csrc/ba_schur.hpp; raw float64 device addresses).
    ``work`` (a device buffer of ``schur_syrk_work_len`` doubles): the ordered form — the partial sums of the K split are subtracted
    in a fixed order instead of meeting in atomics."""
    s = c_void_p(0) if stream is None else _stream_arg(stream)
    if work:
        check(lib().pcs_schur_syrk_ordered(int(device), int(n_lead), int(n_trail), c_void_p(d_V), int(ldv), c_void_p(d_S), int(lds), c_void_p(d_u or 0), c_void_p(d_rhs or 0),
                                           c_void_p(work), int(work_len), s))
        return
    check(lib().pcs_schur_syrk(int(device), int(n_lead), int(n_trail), c_void_p(d_V), int(ldv), c_void_p(d_S), int(lds), c_void_p(d_u or 0), c_void_p(d_rhs or 0), s))


def schur_syrk_work_len(n_lead: int, n_trail: int) -> int:
    """Doubles of the workspace ``schur_syrk(..., work=...)`` needs for the ordered (deterministic) product; 0 = not split."""
    return int(lib().pcs_schur_syrk_work_len(int(n_lead), int(n_trail)))


def schur_vtx(device: int, n_lead: int, n_trail: int, d_V: int, ldv: int, d_x: int, d_w: int, stream: int | None = None):
    """w = V' x on the device (csrc/ba_schur.hpp)."""
    s = c_void_p(0) if stream is None else _stream_arg(stream)
    check(lib().pcs_schur_vtx(int(device), int(n_lead), int(n_trail), c_void_p(d_V), int(ldv), c_void_p(d_x), c_void_p(d_w), s))


def dense_spd_work_len(n: int) -> int:
    return int(lib().pcs_dense_spd_work_len(int(n)))


def device_count() -> int:
    return int(lib().pcs_device_count())
