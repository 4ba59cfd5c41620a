"""Chains as data: a fused HIP kernel for ANY composition of the reference's function blocks.

The reference's ``optimisation_function`` accepts any list of blocks (``a + b + c``, abstract_function_blocks.py:735-748)
and code-generates the loss, the Jacobian driver and the chain rule for it (afb:290-419, afb:492-652,
matmul_map.py:147-263); user blocks are its documented extension point.  The three chains its handlers build have
hand-fused kernels here (csrc/ba_kernels.hpp).  Every other valid composition of the five known blocks

    projection + T_1 + ... + T_M + source      T_i in {rigidTform3d (per image), extrinsic3D (per camera)}
                                               source in {template_points (per image), free_point (per key)}

goes through this module: ``ChainSpec`` turns the block list into the tables of csrc/ba_generic.hpp (which rigid parameter
group and which index each block reads), ``emit_source`` writes the ~15-line translation unit that instantiates the
generic kernels for it, ``compile_chain`` has hipcc build a gfx950 code object (cached by content hash under
``pycamset_amd/_chains/``; hipcc cross-compiles without a GPU, so ``__graft_entry__.build()`` pre-builds the chains the
tests use), and ``ChainEngine`` drives it through the C ABI (``pcs_genchain_*``, include/pcs_hip.h) with the subset of
``Engine``'s interface the operator API needs.  There is no interpreter and no CPU fallback: a composition outside the
family raises.

Parameter layout = the reference's ``make_param_struct`` (afb:777-820): unique parameter objects in block order — two
blocks of the same class share their class-level ``params`` object (afb:160-163) and therefore ONE group; each group
takes ``n_params x count(link type)`` consecutive columns of the parameter string.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
from ctypes import POINTER, byref, c_double, c_float, c_int32, c_int64, c_void_p
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

from . import _capi
from ._capi import check, lib

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
CACHE = PKG / "_chains"
LINK_CAM, LINK_IMG, LINK_KEY = 0, 1, 2          # afb:42-46 key_type
SRC_TEMPLATE, SRC_FREE = 0, 1
MAX_TRANSFORMS, MAX_GROUPS = 6, 8


@dataclass
class ChainSpec:
    """What csrc/ba_generic.hpp needs to know about a composition."""
    names: tuple                       # block class names, in block order
    n_transforms: int                  # M
    src_kind: int                      # SRC_TEMPLATE / SRC_FREE
    block_group: tuple                 # rigid-group id of transform block i (0..M-1) and, for a template source, of the source (entry M)
    block_link: tuple                  # LINK_CAM / LINK_IMG of the same blocks
    # parameter groups in string order: (kind, link, n_params) with kind in {"intr", "rigid", "point"}; rigid groups carry their slab id
    groups: list = field(default_factory=list)
    group_of_block: tuple = ()         # for EVERY block (projection and free_point included): index into `groups`

    @property
    def P(self) -> int:
        return 9 + 6 * self.n_transforms + (6 if self.src_kind == SRC_TEMPLATE else 3)

    @property
    def n_rigid_groups(self) -> int:
        return sum(1 for g in self.groups if g["kind"] == "rigid")

    @classmethod
    def from_blocks(cls, function_blocks) -> "ChainSpec":
        names = tuple(type(b).__name__ for b in function_blocks)

        def bad(why):
            return NotImplementedError(
                f"chain '{' + '.join(names)}' cannot be compiled: {why}.  Supported: projection + any number (<= {MAX_TRANSFORMS}) of "
                "rigidTform3d / extrinsic3D + template_points | free_point")

        if len(names) < 2 or names[0] != "projection":
            raise bad("the first block must be `projection`")
        if names[-1] not in ("template_points", "free_point"):
            raise bad("the last block must produce the 3-D point (`template_points` or `free_point`)")
        mids = names[1:-1]
        if any(n not in ("rigidTform3d", "extrinsic3D") for n in mids):
            raise bad("only rigid transforms may stand between the projection and the point source")
        if len(mids) > MAX_TRANSFORMS:
            raise bad("too many transforms")
        groups, seen, block_group, block_link, group_of_block = [], {}, [], [], []
        rigid_id = 0
        for b in function_blocks:
            key = id(b.params)                      # the reference tells groups apart by object identity (afb:160-163)
            name = type(b).__name__
            if key not in seen:
                kind = "intr" if name == "projection" else "point" if name == "free_point" else "rigid"
                g = dict(kind=kind, link=int(b.params.link_type), n_params=int(b.params.n_params), slab=None)
                if kind == "rigid":
                    if g["link"] not in (LINK_CAM, LINK_IMG) or g["n_params"] != 6:
                        raise bad(f"block {name} does not carry a 6-parameter per-camera / per-image transform")
                    g["slab"] = rigid_id
                    rigid_id += 1
                seen[key] = len(groups)
                groups.append(g)
            g = groups[seen[key]]
            group_of_block.append(seen[key])
            if name in ("rigidTform3d", "extrinsic3D", "template_points"):
                block_group.append(g["slab"])
                block_link.append(g["link"])
        if rigid_id > MAX_GROUPS:
            raise bad("too many rigid parameter groups")
        src_kind = SRC_TEMPLATE if names[-1] == "template_points" else SRC_FREE
        return cls(names=names, n_transforms=len(mids), src_kind=src_kind, block_group=tuple(block_group), block_link=tuple(block_link), groups=groups,
                   group_of_block=tuple(group_of_block))

    # -- the reference's parameter-string layout for given entity counts (afb:793-818) --------------------
    def layout(self, n_cams: int, n_imgs: int, n_keys: int) -> dict:
        count = {LINK_CAM: n_cams, LINK_IMG: n_imgs, LINK_KEY: n_keys}
        off, starts = 0, []
        for g in self.groups:
            starts.append(off)
            off += g["n_params"] * count[g["link"]]
        rigid = [(starts[i], count[g["link"]]) for i, g in enumerate(self.groups) if g["kind"] == "rigid"]
        intr = [starts[i] for i, g in enumerate(self.groups) if g["kind"] == "intr"]
        point = [starts[i] for i, g in enumerate(self.groups) if g["kind"] == "point"]
        return dict(n_params=off, starts=starts, rigid_off=[r[0] for r in rigid], rigid_count=[r[1] for r in rigid],
                    intr_off=intr[0], point_off=point[0] if point else 0)


def emit_source(spec: ChainSpec) -> str:
    """The translation unit of one chain: a ChainSpec struct + the entry points of csrc/ba_generic.hpp."""
    link_name = {LINK_CAM: "pcs::LINK_CAM", LINK_IMG: "pcs::LINK_IMG"}
    groups = list(spec.block_group) or [-1]
    links = [link_name[l] for l in spec.block_link] or ["-1"]
    src = "pcs::SRC_TEMPLATE" if spec.src_kind == SRC_TEMPLATE else "pcs::SRC_FREE"
    return (
        f"// generated by pycamset_amd/chain_compiler.py for: {' + '.join(spec.names)}\n"
        '#include "ba_generic.hpp"\n'
        "struct Chain {\n"
        f"    static constexpr int M = {spec.n_transforms}, SRC = {src}, P = {spec.P};\n"
        f"    __host__ __device__ static constexpr int group(int b) {{ constexpr int t[] = {{{', '.join(str(g) for g in groups)}}}; return t[b]; }}\n"
        f"    __host__ __device__ static constexpr int link(int b) {{ constexpr int t[] = {{{', '.join(links)}}}; return t[b]; }}\n"
        "};\n"
        "PCS_GENCHAIN_ENTRY_POINTS(Chain)\n"
    )


def _header_digest() -> str:
    h = hashlib.sha256()
    for name in ("ba_generic.hpp", "ba_kernels.hpp", "ba_device.hpp"):
        h.update((CSRC / name).read_bytes())
    return h.hexdigest()


def code_object_path(spec: ChainSpec) -> Path:
    key = hashlib.sha256((emit_source(spec) + _header_digest()).encode()).hexdigest()[:20]
    return CACHE / f"chain_{'_'.join(n[:4] for n in spec.names)}_{key}.hsaco"


def compile_chain(spec: ChainSpec, verbose: bool = False) -> Path:
    """gfx950 code object of the chain, built by hipcc unless an up-to-date one is cached."""
    out = code_object_path(spec)
    if out.exists():
        return out
    CACHE.mkdir(exist_ok=True)
    src = out.with_suffix(".hip")
    src.write_text(emit_source(spec))
    cmd = [os.environ.get("HIPCC", "hipcc"), "--genco", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{CSRC}", str(src), "-o", str(out)]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0 or not out.exists():
        raise RuntimeError(f"hipcc failed compiling the chain {' + '.join(spec.names)}:\n{proc.stderr[-2000:]}")
    return out


def block_param_inds(spec: ChainSpec, lay: dict, det_idx: np.ndarray) -> np.ndarray:
    """(N, P) global parameter indices of every detection's block row, in block order (afb:192-233).
    ``det_idx`` = the integer (cam, image, key) columns of the detection table."""
    idx = {LINK_CAM: det_idx[:, 0], LINK_IMG: det_idx[:, 1], LINK_KEY: det_idx[:, 2]}
    cols = []
    for g in spec.group_of_block:     # blocks that share a parameter group repeat its columns (the reference gathers per block)
        grp = spec.groups[g]
        cols.append(lay["starts"][g] + grp["n_params"] * idx[grp["link"]][:, None] + np.arange(grp["n_params"])[None, :])
    return np.ascontiguousarray(np.concatenate(cols, axis=1), dtype=np.int64)


def csr_structure_of(cols: np.ndarray, n_params: int, unfixed=None):
    """(indices, indptr, dense positions of the kept entries) of the CSR Jacobian with the fixed columns removed
    (afb:465-489); each detection's index row serves its u row and its v row (afb:475-479)."""
    mask = np.ones(n_params, dtype=bool) if unfixed is None else np.asarray(unfixed, dtype=bool)
    if mask.shape[0] != n_params:
        raise ValueError("unfixed mask must have one entry per parameter")
    conv = np.concatenate([[0], np.cumsum(mask)])
    rows2 = np.repeat(cols, 2, axis=0)
    keep2 = mask[rows2]
    indices = conv[rows2[keep2]].astype(np.int64)
    indptr = np.concatenate([[0], np.cumsum(keep2.sum(axis=1))]).astype(np.int64)
    return indices, indptr, np.ascontiguousarray(np.flatnonzero(keep2.ravel()), dtype=np.int64)


class ChainEngine:
    """One generated chain on one device — the ``Engine`` interface the operator API uses (NumPy in / NumPy out)."""

    dtype = "f64"

    def __init__(self, function_blocks, n_cams: int, n_imgs: int, n_keys: int, *, device: int = 0):
        self.spec = ChainSpec.from_blocks(function_blocks)
        self.chain = " + ".join(self.spec.names)
        self.n_cams, self.n_imgs, self.n_keys, self.device = int(n_cams), int(n_imgs), int(n_keys), int(device)
        self.P = self.spec.P
        self.lay = self.spec.layout(self.n_cams, self.n_imgs, self.n_keys)
        self.n_params = self.lay["n_params"]
        path = compile_chain(self.spec)
        ng = self.spec.n_rigid_groups
        off = (c_int64 * max(1, ng))(*self.lay["rigid_off"])
        cnt = (c_int32 * max(1, ng))(*self.lay["rigid_count"])
        self._h = c_void_p()
        check(lib().pcs_genchain_create(byref(self._h), str(path).encode(), self.spec.n_transforms, self.spec.src_kind, ng, off, cnt,
                                        self.lay["intr_off"], self.lay["point_off"], self.n_params, self.n_cams, self.n_imgs, self.n_keys, self.device))
        self.n = 0
        self.nnz = None
        self.mask_key = None
        self._det = None
        self._rings = {}

    def close(self):
        self._rings = {}
        if getattr(self, "_h", None) is not None and self._h:
            lib().pcs_genchain_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- static inputs ---------------------------------------------------------------------------------------------------
    def set_detections_table(self, det5: np.ndarray):
        det5 = np.ascontiguousarray(det5, dtype=np.float64)
        if det5.ndim != 2 or det5.shape[1] != 5:
            raise ValueError("detections must have shape (N, 5)")
        check(lib().pcs_genchain_set_detections_table(self._h, det5.ctypes.data_as(POINTER(c_double)), det5.shape[0]))
        self.n = det5.shape[0]
        self._det = det5[:, :3].astype(np.int64)
        self.nnz = None

    def set_template(self, points: np.ndarray):
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        if pts.shape[0] < self.n_keys:
            raise ValueError(f"template has {pts.shape[0]} points, the chain indexes {self.n_keys} keys")
        check(lib().pcs_genchain_set_template(self._h, pts.ctypes.data_as(POINTER(c_double))))

    # -- evaluation ------------------------------------------------------------------------------------------------------
    def _check_params(self, param_str) -> np.ndarray:
        p = np.ascontiguousarray(param_str, dtype=np.float64).ravel()
        if p.shape[0] != self.n_params:
            raise ValueError(f"parameter string has {p.shape[0]} entries, the chain expects {self.n_params}")
        return p

    def eval(self, param_str, want_resid: bool = True, want_jac: bool = True, pinned_ring: int = 0):
        p = self._check_params(param_str)
        r = np.empty((self.n, 2)) if want_resid else None
        j = np.empty((2 * self.n, self.P)) if want_jac else None
        dp = POINTER(c_double)
        check(lib().pcs_genchain_eval(self._h, p.ctypes.data_as(dp), r.ctypes.data_as(dp) if want_resid else None, j.ctypes.data_as(dp) if want_jac else None))
        return r, j

    def eval_device(self, d_param_str: int, d_resid: int | None, d_jac: int | None, stream: int | None = None):
        from .engine import _stream_arg

        check(lib().pcs_genchain_eval_device(self._h, c_void_p(d_param_str), c_void_p(d_resid or 0), c_void_p(d_jac or 0), _stream_arg(stream)))

    def device_buffers(self):
        r, j = c_void_p(), c_void_p()
        check(lib().pcs_genchain_device_buffers(self._h, byref(r), byref(j)))
        return int(r.value), int(j.value)

    def synchronize(self, stream: int | None = None):
        from .engine import _stream_arg

        check(lib().pcs_genchain_synchronize(self._h, _stream_arg(stream)))

    def last_kernel_ms(self):
        a, b = c_float(), c_float()
        check(lib().pcs_genchain_last_kernel_ms(self._h, byref(a), byref(b)))
        return float(a.value), float(b.value)

    # -- static structure (integer work on the host, like the reference's afb:192-233, afb:465-489) -------------------------
    def block_param_inds(self) -> np.ndarray:
        if self._det is None:
            raise RuntimeError("no detections set")
        return block_param_inds(self.spec, self.lay, self._det)

    def csr_structure(self, unfixed=None):
        indices, indptr, _ = csr_structure_of(self.block_param_inds(), self.n_params, unfixed)
        return indices, indptr

    def set_unfixed(self, unfixed) -> int:
        _, _, src = csr_structure_of(self.block_param_inds(), self.n_params, unfixed)
        check(lib().pcs_genchain_set_gather(self._h, src.ctypes.data_as(POINTER(c_int64)), src.shape[0]))
        self.nnz = int(src.shape[0])
        self.mask_key = None if unfixed is None else hash(np.asarray(unfixed, dtype=bool).tobytes())
        return self.nnz

    def eval_compact(self, param_str, want_resid: bool = False, pinned_ring: int = 0):
        if self.nnz is None:
            raise RuntimeError("call set_unfixed() first")
        p = self._check_params(param_str)
        r = np.empty((self.n, 2)) if want_resid else None
        d = np.empty(self.nnz)
        dp = POINTER(c_double)
        check(lib().pcs_genchain_eval_compact(self._h, p.ctypes.data_as(dp), r.ctypes.data_as(dp) if want_resid else None, d.ctypes.data_as(dp)))
        return r, d
