"""Host-side mirror of pyCamSet's operator API for the bundle-adjustment hot path.

Same names, argument meaning and return layouts as
``pyCamSet/optimisation/abstract_function_blocks.py`` (afb) and
``pyCamSet/optimisation/function_block_implementations.py`` (fbi), so that a pyCamSet
``ParamHandler`` can swap its ``self.op_fun`` for one built from these blocks:

    op_fun = projection() + extrinsic3D() + template_points()          # template_handler.py:152
    loss = op_fun.make_full_loss_fn(detections, threads)                # afb:656-658
    jac  = op_fun.make_jacobean(detections, threads, unfixed_params)    # afb:661-667
    param_str = op_fun.build_param_list(intr, extr, poses)              # afb:669-681
    loss(param_str, template) -> (N, 2)
    jac(param_str, template)  -> (data, indices, indptr)

The blocks here are *declarations*: the arithmetic of every block (and the chain rule the
reference code-generates with matmul_map.py) lives in the fused HIP kernels behind the C ABI.  The three
chains the reference's handlers build run on hand-fused kernels (csrc/ba_kernels.hpp); every other valid
composition of the shipped blocks AND of user blocks (``device_function_block`` below: the reference's extension point,
afb:689-775, with the two bodies given as device code) is compiled on first use into a fused kernel of its own
(pycamset_amd/chain_compiler.py -> csrc/ba_generic.hpp, hipcc --genco) — the counterpart of the reference's code generator
(afb:424-463, afb:492-652, mm:147-263).  A block that is neither shipped nor a device_function_block raises, loudly: there
is no interpreter fallback.
"""
from __future__ import annotations

from enum import IntEnum

import numpy as np

from .engine import Engine


class key_type(IntEnum):  # afb:42-46
    PER_CAM = 0
    PER_IMG = 1
    PER_KEY = 2
    SINGLE = 3


class param_type:  # afb:50-70
    def __init__(self, link_type: key_type, n_params: int, mod_function=None) -> None:
        self.link_type = link_type
        self.n_params = n_params
        self.sparse_param = False
        self.mod_function = mod_function


class abstract_function_block:  # afb:689-748
    array_memory: int = 0
    template = False
    num_inp: int = 0
    num_out: int = 0
    params: param_type = None

    def __add__(self, other):
        if isinstance(other, abstract_function_block):
            return optimisation_function([self, other])
        if isinstance(other, optimisation_function):
            return optimisation_function([self] + other.function_blocks)
        raise ValueError(f"could not combine function block with {other}")

    def __radd__(self, other):
        if isinstance(other, abstract_function_block):
            return optimisation_function([other, self])
        if isinstance(other, optimisation_function):
            return optimisation_function(other.function_blocks + [self])
        raise ValueError(f"could not combine function block with {other}")


class device_function_block(abstract_function_block):
    """A USER block for generated chains — the GPU counterpart of subclassing the reference's ``abstract_function_block``
    (afb:689-775: ``num_inp``, ``num_out``, ``params``, ``compute_fun``, ``compute_jac``).  Declare the same three attributes
    and give the two bodies as HIP device code:

        class cam_scale(device_function_block):          # one isotropic scale per camera, between projection and extrinsic
            num_inp, num_out = 3, 3
            params = param_type(key_type.PER_CAM, 1)
            device_fun = "for (int i = 0; i < 3; ++i) out[i] = params[0] * inp[i];"
            device_jac = '''for (int o = 0; o < 3; ++o) {
                                out[o * 4 + 0] = inp[o];                                   // d out / d params
                                for (int i = 0; i < 3; ++i) out[o * 4 + 1 + i] = (o == i) ? params[0] : 0.0;   // d out / d inp
                            }'''

        op_fun = projection() + cam_scale() + extrinsic3D() + template_points()

    ``device_fun`` is the body of ``void fun(const double *params, const double *inp, double *out)`` (``out[num_out]``);
    ``device_jac`` the body of ``void jac(const double *params, const double *inp, double *out)`` with ``out`` laid out like
    ``compute_jac``'s output: ``num_out`` rows of ``n_params + num_inp`` entries, parameter columns first (afb:738-748).
    ``params`` points at this block's parameters of the detection's camera / image / key (``params.link_type``) inside the
    parameter string; ``inp`` is the next block's output (NULL for a source, ``num_inp = 0``).  Everything is FP64; the bodies are
    pasted into the chain's translation unit (pycamset_amd/chain_compiler.py) next to the built-in blocks and chained by the
    same rule S <- S . d out / d inp.  A first block must have ``num_out = 2`` (the pixel), neighbours must agree
    (``num_inp`` of a block = ``num_out`` of the next), the last block is a source.  As in the reference, blocks that share
    one ``param_type`` OBJECT share one parameter group (afb:160-163).

    A TEMPLATED source (round 5) — ``template = True`` on the LAST block, ``num_inp = 0`` — receives the detection's template point
    as ``inp`` (three doubles), exactly like the reference's ``inp[:3] = t_data[int(datum[2])]`` (afb:138, afb:374-375, afb:582;
    shipped example: ``template_points``, fbi:188-211): a per-image similarity or a board-flex model of the calibration target is
    written like this::

        class board_flex(device_function_block):            # out = [sx X + tx, sy Y + ty, Z + k (X^2 + Y^2)], one set per image
            template = True
            num_inp, num_out, array_memory = 0, 3, 0
            params = param_type(key_type.PER_IMG, 5)
            device_fun = "out[0] = params[0] * inp[0] + params[2]; out[1] = params[1] * inp[1] + params[3];" \
                         " out[2] = inp[2] + params[4] * (inp[0] * inp[0] + inp[1] * inp[1]);"
            device_jac = "for (int i = 0; i < 15; ++i) out[i] = 0.0; out[0] = inp[0]; out[2] = 1.0; out[6] = inp[1]; out[8] = 1.0;" \
                         " out[14] = inp[0] * inp[0] + inp[1] * inp[1];"

        op_fun = projection() + extrinsic3D() + rigidTform3d() + board_flex()     # loss_fn(param_str, template)"""
    device_fun: str = ""
    device_jac: str = ""


class projection(abstract_function_block):  # fbi:21-140
    num_inp, num_out, array_memory = 3, 2, 1
    params = param_type(key_type.PER_CAM, 9)


class rigidTform3d(abstract_function_block):  # fbi:143-182
    num_inp, num_out, array_memory = 3, 3, 27
    params = param_type(key_type.PER_IMG, 6)


class extrinsic3D(rigidTform3d):  # fbi:184-185
    params = param_type(key_type.PER_CAM, 6)


class template_points(rigidTform3d):  # fbi:188-211
    template = True
    num_inp, num_out = 0, 3
    params = param_type(key_type.PER_IMG, 6)


class free_point(abstract_function_block):  # fbi:216-240
    num_inp, num_out, array_memory = 0, 3, 0
    params = param_type(key_type.PER_KEY, 3)


_CHAINS = {
    (projection, extrinsic3D, template_points): "template",
    (projection, extrinsic3D, rigidTform3d, free_point): "self",
    (projection, extrinsic3D, free_point): "free",
}


def _counts(detections: np.ndarray) -> tuple[int, int, int]:
    """make_param_struct's group counts: max index + 1 per link type (afb:793-795)."""
    if detections.shape[0] == 0:
        raise ValueError("empty detection table")
    return (int(np.max(detections[:, 0])) + 1, int(np.max(detections[:, 1])) + 1, int(np.max(detections[:, 2])) + 1)


class optimisation_function:  # afb:111-685
    """A chain of function blocks evaluated by the MI355X engine."""

    DEFAULT_PINNED_RING = 2

    def __init__(self, function_blocks, *, dtype: str = "f64", device: int = 0, pinned_ring: int | None = None, counts=None) -> None:
        """``pinned_ring`` = R > 0 returns the Jacobian ``data`` array from a ring of R page-locked host buffers
        (PCIe-rate device -> host copy, wrapped by ``csr_array`` without a copy: a 335 MB Jacobian arrives in ~10 ms instead of
        ~34 ms into pageable memory).  A ring buffer is handed out again only once nobody references it or any view
        of it any more (``Engine._out``): an array — or a ``csr_array`` built on it — that the caller keeps stays intact for as
        long as it is kept, exactly like the fresh array the reference returns per call (afb:561); the slot simply gets a new
        page-locked buffer.  A loop that drops each Jacobian before asking for the next (scipy's ``least_squares``,
        optimisation_handling.py:88-98) cycles through R buffers without allocating.  The default (None) is R = 2;
        0 = always a fresh pageable NumPy array."""
        self.pinned_ring = self.DEFAULT_PINNED_RING if pinned_ring is None else int(pinned_ring)
        # (n_cams, n_imgs, n_keys): slab sizes of the parameter string.  None = the reference's own rule, max
        # index + 1 of the detection table (afb:793-795).  The handlers pass their slab sizes (a trailing camera /
        # image / key without detections must still have its place in the string), a rank that holds only a
        # shard of the detections passes the GLOBAL sizes.  Counts smaller than what the table needs are raised
        # to max index + 1.
        self.counts = None if counts is None else tuple(int(c) for c in counts)
        self.function_blocks = list(function_blocks)
        self.n_blocks = len(self.function_blocks)
        self.dtype, self.device = dtype, device
        self.templated = self.function_blocks[-1].template
        self.n_params = np.array([b.params.n_params for b in self.function_blocks])
        self.param_line_length = int(self.n_params.sum())  # P
        self._engine: Engine | None = None
        self._engine_key = None

    @property
    def chain(self) -> str:
        """Which fused kernel evaluates this chain: 'template' / 'self' / 'free' (hand-fused) or 'generated'.
        `a + b + c` builds intermediate partial chains (afb:735-748), so validity is checked on use, not on construction
        (an invalid composition raises NotImplementedError from the chain compiler)."""
        kinds = tuple(type(b) for b in self.function_blocks)
        if kinds in _CHAINS:
            return _CHAINS[kinds]
        from .chain_compiler import ChainSpec

        ChainSpec.from_blocks(self.function_blocks)   # raises for compositions outside the compilable family
        return "generated"

    # -- engine sharing between the loss and the Jacobian closures ---------------------------
    def _engine_for(self, detections: np.ndarray) -> Engine:
        det = np.ascontiguousarray(detections, dtype=np.float64)
        if det.ndim != 2 or det.shape[1] != 5:
            raise ValueError("detections must be the flattened (N, 5) table [cam, im, key, u, v]")
        # Which table the engine holds.  Hashing 40 MB per call (1e6 detections) cost lm_solve 20 ms of its 28: the SAME array
        # OBJECT with the same sampled rows is taken as the same table.  "Same object" is a weak reference that is still alive
        # and still refers to the array passed in — an id() alone could belong to a new array that reuses a freed address.  In-place
        # edits of rows outside the sample are NOT seen, on purpose: the reference's closures capture their tables when they are
        # made (afb:406-419) and later edits never reach them either; pass a new array for a new table.  Anything else is hashed in full.
        import weakref

        sample = det[:: max(1, det.shape[0] // 256)]
        quick = (det.shape, det.ctypes.data, hash(sample.tobytes()))
        held = getattr(self, "_engine_ref", None)
        same_object = held is not None and held() is detections
        if self._engine is not None and same_object and quick == getattr(self, "_engine_quick", None):
            return self._engine
        try:
            self._engine_ref = weakref.ref(detections)
        except TypeError:                     # not weak-referenceable (a list, a tuple): never take the fast path
            self._engine_ref = None
        key = (det.shape, hash(det.tobytes()))
        self._engine_quick = quick
        if self._engine is None or key != self._engine_key:
            if det.shape[0] == 0 and self.counts is not None:
                C, I, K = self.counts    # an empty shard of a sharded table: the global layout, no detections
            else:
                C, I, K = _counts(det)
                if self.counts is not None:
                    C, I, K = (max(a, b) for a, b in zip(self.counts, (C, I, K)))
            if self.chain == "generated":
                from .chain_compiler import ChainEngine

                eng = ChainEngine(self.function_blocks, C, I, K, device=self.device, dtype=self.dtype)
            else:
                eng = Engine(self.chain, C, I, K, dtype=self.dtype, device=self.device)
            eng.set_detections_table(det)
            self._engine, self._engine_key = eng, key
        return self._engine

    def _bind_template(self, eng: Engine, template) -> None:
        if not self.templated:
            return
        if template is None:
            raise ValueError("the template chain needs the template points (target.point_data.reshape(-1, 3))")
        t = np.ascontiguousarray(template, dtype=np.float64).reshape(-1, 3)
        key = hash(t.tobytes())
        if key != getattr(eng, "_bound_template_key", None):   # per engine: closures built on different tables own different engines
            eng.set_template(t)
            eng._bound_template_key = key

    @staticmethod
    def _leading(param, eng: Engine) -> np.ndarray:
        """The generated reference functions gather ``inp_params[block_param_inds]`` (afb:363, afb:570): entries of
        the parameter string beyond the last indexed one are never read.  A string LONGER than the engine's layout is
        therefore cut to its leading ``n_params`` entries — with ``counts=None`` (layout from max index + 1 of the
        detections, afb:793-795) that is bit for bit what the reference evaluates, including its mis-offset groups
        when a trailing camera / image has no detection (SURVEY 8a quirk ii).  A shorter string is an error."""
        p = np.asarray(param, dtype=np.float64).reshape(-1)
        return p[: eng.n_params] if p.shape[0] > eng.n_params else p

    # -- reference API -------------------------------------------------------------------------
    def can_make_jac(self) -> bool:  # afb:683-684
        return True

    def build_param_list(self, *args) -> np.ndarray:  # afb:669-681
        return np.concatenate([np.asarray(a, dtype=np.float64).flatten() for a in args], axis=0)

    def make_full_loss_fn(self, detections, threads=None):  # afb:656-658 -> afb:290-419
        """``threads`` is accepted for signature compatibility and ignored (the GPU grid replaces
        the reference's prange chunks; its np.resize padding never reaches the output, afb:385)."""
        eng = self._engine_for(detections)

        def loss_fn(param, template=None):
            self._bind_template(eng, template)
            r, _ = eng.eval(self._leading(param, eng), want_resid=True, want_jac=False)
            return r

        return loss_fn

    def make_jacobean(self, detections, threads=None, unfixed_params=None):  # afb:661-667 -> afb:492-652
        eng = self._engine_for(detections)
        if unfixed_params is None:
            unfixed = np.ones(eng.n_params, dtype=bool)
        else:
            # A mask longer than the string (a handler whose slabs have trailing entries no detection indexes) is
            # cut like the string itself, see _leading(): `conversion` (afb:482) only ever looks up indexed columns.
            unfixed = np.asarray(unfixed_params, dtype=bool)[: eng.n_params]
            if unfixed.shape[0] != eng.n_params:
                raise ValueError(f"unfixed_params has {unfixed.shape[0]} entries, expected at least {eng.n_params}")
        indices, indptr = eng.csr_structure(unfixed)  # static, built once (afb:619)
        all_free = bool(np.all(unfixed))
        mask_key = hash(unfixed.tobytes())
        if not all_free:
            eng.set_unfixed(unfixed)

        def jac_fn(param, template=None):
            self._bind_template(eng, template)
            if all_free:  # afb:633-642
                _, j = eng.eval(self._leading(param, eng), want_resid=False, want_jac=True, pinned_ring=self.pinned_ring)
                return j.reshape(-1), indices, indptr
            if eng.mask_key != mask_key:  # another closure re-bound the engine's mask
                eng.set_unfixed(unfixed)
            _, data = eng.eval_compact(self._leading(param, eng), pinned_ring=self.pinned_ring)  # afb:644-651, masked on the device
            return data, indices, indptr

        return jac_fn

    def get_block_param_inds(self, detections, threads=None, unthreaded=True) -> np.ndarray:  # afb:192-233
        return self._engine_for(detections).block_param_inds()

    def make_jac_CSR_columns_row_pointers(self, detections, threads, unfixed_params):  # afb:465-489
        eng = self._engine_for(detections)
        return eng.csr_structure(np.asarray(unfixed_params, dtype=bool)[: eng.n_params])

    @property
    def engine(self) -> Engine | None:
        return self._engine
