"""Chains as data: a fused HIP kernel for ANY composition of the reference's function blocks.

The reference's ``optimisation_function`` accepts any list of blocks (``a + b + c``, abstract_function_blocks.py:735-748)
and code-generates the loss, the Jacobian driver and the chain rule for it (afb:290-419, afb:492-652,
matmul_map.py:147-263); user blocks are its documented extension point (afb:689-775).  The three chains its handlers build
have hand-fused kernels here (csrc/ba_kernels.hpp).  Every other valid composition

    first + middle* + source     first  in {projection, user block with num_out = 2}
                                 middle in {rigidTform3d (per image), extrinsic3D (per camera), user block}
                                 source in {template_points (per image), free_point (per key), user block with num_inp = 0}

goes through this module: ``ChainSpec`` validates the block list and lays out the parameter string, ``emit_source`` writes
the translation unit — the user blocks' device bodies (``function_blocks.device_function_block``) and the straight-line
evaluation of one detection: blocks right to left, then the chain rule left to right, calling the built-in blocks' helpers of
csrc/ba_generic.hpp —, ``compile_chain`` has hipcc build a gfx950 code object (cached by content hash under
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
import uuid
from ctypes import POINTER, byref, c_double, c_float, c_int32, c_int64, c_uint64, c_void_p
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
MAX_GROUPS = 8


BUILTIN_KINDS = {"projection": "projection", "rigidTform3d": "rigid", "extrinsic3D": "rigid", "template_points": "template_points", "free_point": "free_point"}
MAX_BLOCKS, MAX_USER = 10, 8
_LINK_CPP = {LINK_CAM: "pcs::LINK_CAM", LINK_IMG: "pcs::LINK_IMG", LINK_KEY: "pcs::LINK_KEY"}


@dataclass
class BlockInfo:
    """One block of a composition as the code generator sees it."""
    name: str            # class name
    kind: str            # 'projection' | 'rigid' | 'template_points' | 'free_point' | 'user'
    link: int            # LINK_CAM / LINK_IMG / LINK_KEY of its parameter group
    n_params: int
    nin: int
    nout: int
    group: int           # index into ChainSpec.groups
    slab: int | None = None    # rigid blocks: which Rodrigues slab
    uidx: int | None = None    # user blocks: ordinal among the user blocks
    device_fun: str = ""
    device_jac: str = ""
    cpp_name: str = ""
    templated: bool = False    # user SOURCE with ``template = True``: its ``inp`` is the detection's template point (afb:138, afb:374-375)


@dataclass
class ChainSpec:
    """What the code generator and the host side need to know about a composition."""
    blocks: list
    # parameter groups in string order: dict(kind in {'intr', 'rigid', 'point', 'user'}, link, n_params, slab)
    groups: list = field(default_factory=list)

    @property
    def names(self) -> tuple:
        return tuple(b.name for b in self.blocks)

    @property
    def group_of_block(self) -> tuple:
        return tuple(b.group for b in self.blocks)

    @property
    def P(self) -> int:
        return sum(b.n_params for b in self.blocks)

    @property
    def n_rigid_groups(self) -> int:
        return sum(1 for g in self.groups if g["kind"] == "rigid")

    @property
    def uses_template(self) -> bool:
        # the reference feeds template[key] to whatever block sits last when that block says template = True (afb:138, afb:374-375, afb:582)
        return self.blocks[-1].kind == "template_points" or self.blocks[-1].templated

    @property
    def user_blocks(self) -> list:
        return [b for b in self.blocks if b.kind == "user"]

    @classmethod
    def from_blocks(cls, function_blocks) -> "ChainSpec":
        names = tuple(type(b).__name__ for b in function_blocks)

        def bad(why):
            return NotImplementedError(
                f"chain '{' + '.join(names)}' cannot be compiled: {why}.  A chain is: [projection | a user block with num_out = 2] + any "
                "rigidTform3d / extrinsic3D / user blocks + [template_points | free_point | a user block with num_inp = 0], each block's "
                "num_inp equal to the next block's num_out (pycamset_amd.function_blocks.device_function_block declares a user block)")

        if len(names) < 2:
            raise bad("a chain needs at least two blocks")
        if len(names) > MAX_BLOCKS:
            raise bad("too many blocks")
        groups, seen, blocks = [], {}, []
        rigid_id = user_id = 0
        for pos, b in enumerate(function_blocks):
            name = type(b).__name__
            is_user = bool(getattr(b, "device_fun", None)) and bool(getattr(b, "device_jac", None))
            if not is_user and name not in BUILTIN_KINDS:
                raise bad(f"block {name} is neither one of the five shipped blocks nor a device_function_block (it has no device_fun / device_jac)")
            kind = "user" if is_user else BUILTIN_KINDS[name]
            key = id(b.params)                      # the reference tells groups apart by object identity (afb:160-163)
            gkind = {"projection": "intr", "rigid": "rigid", "template_points": "rigid", "free_point": "point", "user": "user"}[kind]
            link, npar = int(b.params.link_type), int(b.params.n_params)
            if link not in (LINK_CAM, LINK_IMG, LINK_KEY):
                raise bad(f"block {name}: parameters must be per camera, per image or per key")
            if key not in seen:
                g = dict(kind=gkind, link=link, n_params=npar, slab=None)
                if gkind == "rigid":
                    if link not in (LINK_CAM, LINK_IMG) or npar != 6:
                        raise bad(f"block {name} does not carry a 6-parameter per-camera / per-image transform")
                    g["slab"] = rigid_id
                    rigid_id += 1
                seen[key] = len(groups)
                groups.append(g)
            g = groups[seen[key]]
            if g["kind"] != gkind:
                raise bad(f"block {name} shares its parameter object with a block of another kind")
            info = BlockInfo(name=name, kind=kind, link=link, n_params=npar, nin=int(b.num_inp), nout=int(b.num_out), group=seen[key], slab=g["slab"])
            if kind == "user":
                if npar < 1 or info.nout < 1 or info.nin < 0:
                    raise bad(f"user block {name}: n_params >= 1, num_out >= 1, num_inp >= 0 expected")
                info.uidx, info.device_fun, info.device_jac = user_id, str(b.device_fun), str(b.device_jac)
                info.cpp_name = f"{''.join(ch if ch.isalnum() else '_' for ch in name)}_{user_id}"
                info.templated = bool(getattr(b, "template", False))
                if info.templated and (pos != len(names) - 1 or info.nin != 0):
                    raise bad(f"user block {name} says template = True: only the LAST block receives the template point, and it is a source (num_inp = 0) "
                              "— its `inp` then holds the three coordinates of the detection's template point (afb:374-375)")
                user_id += 1
            elif kind == "template_points":
                info.nin = 0
            blocks.append(info)
        if rigid_id > MAX_GROUPS:
            raise bad("too many rigid parameter groups")
        if user_id > MAX_USER:
            raise bad("too many user blocks")
        # shape of the chain: out of block i + 1 is the input of block i (afb:375-383); the first block yields the pixel
        if blocks[0].kind not in ("projection", "user") or blocks[0].nout != 2:
            raise bad("the first block must produce the two pixel coordinates (`projection` or a user block with num_out = 2)")
        if blocks[-1].nin != 0:
            raise bad("the last block must be a source (num_inp = 0: `template_points`, `free_point` or a user block)")
        for i, b in enumerate(blocks):
            if b.kind == "projection" and i != 0:
                raise bad("`projection` can only be the first block")
            if b.kind in ("template_points", "free_point") and i != len(blocks) - 1:
                raise bad(f"`{b.name}` can only be the last block")
            if b.kind == "rigid" and (i == 0 or i == len(blocks) - 1):
                raise bad(f"`{b.name}` needs a block on either side")
            if i + 1 < len(blocks) and b.nin != blocks[i + 1].nout:
                raise bad(f"block {i} ({b.name}) takes {b.nin} inputs but block {i + 1} ({blocks[i + 1].name}) produces {blocks[i + 1].nout}")
        return cls(blocks=blocks, groups=groups)

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
                    intr_off=intr[0] if intr else 0, point_off=point[0] if point else 0, user_off=[starts[b.group] for b in self.user_blocks])


def emit_source(spec: ChainSpec) -> str:
    """The translation unit of one chain: the user blocks' device bodies, `struct Chain` with the straight-line evaluation of one
    detection (forward right to left, chain rule left to right — what the reference writes into template_functions/*.py,
    afb:350-387, afb:552-599, mm:218-243) and the entry points of csrc/ba_generic.hpp."""
    B, P = spec.blocks, spec.P
    nb = len(B)
    out = [f"// generated by pycamset_amd/chain_compiler.py for: {' + '.join(spec.names)}", '#include "ba_generic.hpp"']
    for b in spec.user_blocks:
        out += ["namespace user {",
                f"struct {b.cpp_name} {{   // user block {b.uidx}: {b.name}",
                f"    static constexpr int NP = {b.n_params}, NIN = {b.nin}, NOUT = {b.nout};",
                "    // out[NOUT]",
                "    __device__ static __forceinline__ void fun(const double *params, const double *inp, double *out) {",
                b.device_fun,
                "    }",
                "    // out[NOUT x (NP + NIN)], row-major, parameter columns first (compute_jac's layout, afb:738-748)",
                "    __device__ static __forceinline__ void jac(const double *params, const double *inp, double *out) {",
                b.device_jac,
                "    }",
                "};",
                "}  // namespace user"]
    fwd, rule = [], []
    col = 0
    cols = []
    for b in B:
        cols.append(col)
        col += b.n_params
    # forward: the source first (afb:375-383 walks the blocks in reverse)
    for i in reversed(range(nb)):
        b = B[i]
        x_in = f"x{i + 1}"
        if b.kind == "free_point":
            fwd.append(f"double x{i}[3]; {{ const double *pp = c.point(); x{i}[0] = pp[0]; x{i}[1] = pp[1]; x{i}[2] = pp[2]; }}")
        elif b.kind == "template_points":
            fwd.append(f"double x{i}[3], E{i}[9]; {{ const double *tp = c.tpoint(); const double X[3] = {{tp[0], tp[1], tp[2]}}; "
                       f"pcs::rigid_fwd<JAC>(c.slab({b.slab}, {_LINK_CPP[b.link]}), X, x{i}, E{i}); }}")
        elif b.kind == "rigid":
            fwd.append(f"double x{i}[3], E{i}[9]; pcs::rigid_fwd<JAC>(c.slab({b.slab}, {_LINK_CPP[b.link]}), {x_in}, x{i}, E{i});")
        elif b.kind == "user":
            inp = x_in if b.nin > 0 else "nullptr"
            if b.templated:   # a templated source: inp = template[key] (three doubles), like `inp[:3] = t_data[int(datum[2])]` (afb:374-375)
                fwd.append(f"double xt{i}[3]; {{ const double *tp = c.tpoint(); xt{i}[0] = tp[0]; xt{i}[1] = tp[1]; xt{i}[2] = tp[2]; }}")
                inp = f"xt{i}"
            fwd.append(f"double x{i}[{b.nout}]; const double *p{i} = c.user({b.uidx}, {_LINK_CPP[b.link]}, {b.n_params}); user::{b.cpp_name}::fun(p{i}, {inp}, x{i});")
        elif b.kind == "projection":
            fwd.append(f"double Ap[18], Ax[2][3]; pcs::project_generic<JAC>(c.intr(), {x_in}[0], {x_in}[1], {x_in}[2], u, v, Ap, Ax);")
    if B[0].kind == "user":
        fwd.append("u = x0[0]; v = x0[1];")
    # chain rule, first block to source: S{i} = d(u, v) / d(input of block i)
    prev = None
    for i, b in enumerate(B):
        nin = max(b.nin, 1)
        if b.kind == "projection":
            rule.append(f"double S{i}[2][3]; pcs::chain_projection<P>(Ap, Ax, J, S{i});")
        elif b.kind == "user":
            if prev is None:
                rule.append("const double Sid[2][2] = {{1.0, 0.0}, {0.0, 1.0}};")
                prev = "Sid"
            inp = f"x{i + 1}" if b.nin > 0 else (f"xt{i}" if b.templated else "nullptr")
            rule.append(f"double Jb{i}[{b.nout * (b.n_params + b.nin)}]; user::{b.cpp_name}::jac(p{i}, {inp}, Jb{i});")
            rule.append(f"double S{i}[2][{nin}]; pcs::chain_user<P, {cols[i]}, {b.n_params}, {b.nin}, {b.nout}>({prev}, Jb{i}, J, S{i});")
        elif b.kind == "rigid":
            rule.append(f"double S{i}[2][3]; pcs::chain_rigid<P, {cols[i]}>({prev}, E{i}, c.slab({b.slab}, {_LINK_CPP[b.link]}), J, S{i});")
        elif b.kind == "template_points":
            rule.append(f"pcs::chain_template<P, {cols[i]}>({prev}, E{i}, J);")
        elif b.kind == "free_point":
            rule.append(f"pcs::chain_free<P, {cols[i]}>({prev}, J);")
        prev = f"S{i}"
    slab_links = [g["link"] for g in spec.groups if g["kind"] == "rigid"]
    link_cases = " ".join(f"g == {i} ? {_LINK_CPP[l]} :" for i, l in enumerate(slab_links))
    out += ["struct Chain {",
            f"    static constexpr int P = {P};",
            f"    static constexpr int N_SLABS = {len(slab_links)};   // Rodrigues slabs (one per rigid parameter group) and whose transform each holds",
            f"    __device__ static constexpr int slab_link(const int g) {{ return {link_cases} pcs::LINK_CAM; }}",
            "    template <bool JAC, typename Ctx>",
            "    __device__ static __forceinline__ void eval(const Ctx &c, double &u, double &v, double (&J)[2 * P]) {"]
    out += ["        " + ln for ln in fwd]
    out += ["        if constexpr (JAC) {"] + ["            " + ln for ln in rule] + ["        }", "    }", "};", "PCS_GENCHAIN_ENTRY_POINTS(Chain)", ""]
    return "\n".join(out)


# Multiply-adds are fused where the SOURCE has them in one expression, not wherever the optimiser finds a product next to a sum
# (hipcc's default, `fast`): one detection's values then do not depend on which kernel the chain was inlined into — the dense
# kernel, the one that packs at the store and both launch forms write the same bits (with `fast`, 6 % of the entries of a user
# chain's template columns differed in the last two bits between the dense and the packing kernel).
FP_CONTRACT = "-ffp-contract=on"


def _header_digest() -> str:
    h = hashlib.sha256()
    for name in ("ba_generic.hpp", "ba_kernels.hpp", "ba_device.hpp", "ba_rtc_prelude.hpp"):
        h.update((CSRC / name).read_bytes())
    return h.hexdigest()


def code_object_path(spec: ChainSpec) -> Path:
    key = hashlib.sha256((emit_source(spec) + _header_digest() + FP_CONTRACT).encode()).hexdigest()[:20]
    return CACHE / f"chain_{'_'.join(''.join(ch for ch in n if ch.isalnum())[:4] for n in spec.names)}_{key}.hsaco"


def _hiprtc():
    """libhiprtc through ctypes, or None: the HIP runtime-compilation library is part of a runtime-only ROCm install, hipcc is not."""
    import ctypes

    for name in ("libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    return None


def _compile_with_hiprtc(src_text: str, out: Path) -> str | None:
    """Compile the translation unit with hiprtc (no hipcc, no GPU needed); returns None on success, the log otherwise."""
    import ctypes

    rtc = _hiprtc()
    if rtc is None:
        return "libhiprtc.so not found"
    prog = ctypes.c_void_p()
    rtc.hiprtcCreateProgram.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rtc.hiprtcCompileProgram.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]
    if rtc.hiprtcCreateProgram(ctypes.byref(prog), src_text.encode(), b"chain.hip", 0, None, None) != 0:
        return "hiprtcCreateProgram failed"
    try:
        opts = [b"--offload-arch=gfx950", b"-O3", b"-std=c++17", FP_CONTRACT.encode(), f"-I{CSRC}".encode()]
        rc = rtc.hiprtcCompileProgram(prog, len(opts), (ctypes.c_char_p * len(opts))(*opts))
        n = ctypes.c_size_t()
        if rc != 0:
            rtc.hiprtcGetProgramLogSize(prog, ctypes.byref(n))
            log = ctypes.create_string_buffer(n.value + 1)
            rtc.hiprtcGetProgramLog(prog, log)
            return log.value.decode(errors="replace")[-2000:] or f"hiprtcCompileProgram returned {rc}"
        if rtc.hiprtcGetCodeSize(prog, ctypes.byref(n)) != 0 or n.value == 0:
            return "hiprtcGetCodeSize failed"
        code = ctypes.create_string_buffer(n.value)
        if rtc.hiprtcGetCode(prog, code) != 0:
            return "hiprtcGetCode failed"
        _publish(out, code.raw[: n.value])
        return None
    finally:
        rtc.hiprtcDestroyProgram(ctypes.byref(prog))


def _publish(out: Path, data: bytes | None = None, tmp: Path | None = None) -> None:
    """Put a finished code object under its cache name in ONE step: several ranks (torchrun, `bench.py --gpus N`) may compile the
    same chain at first use, and `compile_chain` returns as soon as the name exists — so a name must never point at a file somebody
    is still writing.  The bytes go to a private name in the same directory first; `os.replace` is atomic."""
    if tmp is None:
        tmp = out.with_name(f".{out.name}.{os.getpid()}.{uuid.uuid4().hex[:8]}.tmp")
        tmp.write_bytes(data)
    os.replace(tmp, out)


def compile_chain(spec: ChainSpec, verbose: bool = False) -> Path:
    """gfx950 code object of the chain unless an up-to-date one is cached.  Compiler: hiprtc (runtime compilation: works on a box
    without hipcc — a runtime-only ROCm install — and without a GPU), hipcc --genco when hiprtc is missing or refuses the unit;
    ``PCS_CHAIN_COMPILER=hipcc|hiprtc`` forces one."""
    out = code_object_path(spec)
    if out.exists():
        return out
    CACHE.mkdir(exist_ok=True)
    unique = f"{os.getpid()}.{uuid.uuid4().hex[:8]}"
    src = out.with_name(f".{out.stem}.{unique}.hip")      # private to this process until the object is published
    text = emit_source(spec)
    src.write_text(text)
    try:
        which = os.environ.get("PCS_CHAIN_COMPILER", "auto")
        log = None
        if which in ("auto", "hiprtc"):
            log = _compile_with_hiprtc(text, out)
            if log is None:
                src.replace(out.with_suffix(".hip"))
                if verbose:
                    print(f"hiprtc: {out}", flush=True)
                return out
            if which == "hiprtc":
                raise RuntimeError(f"hiprtc failed compiling the chain {' + '.join(spec.names)}:\n{log}")
        tmp = out.with_name(f".{out.name}.{unique}.tmp")
        cmd = [os.environ.get("HIPCC", "hipcc"), "--genco", "--offload-arch=gfx950", "-O3", "-std=c++17", FP_CONTRACT, f"-I{CSRC}", str(src), "-o", str(tmp)]
        if verbose:
            print(" ".join(cmd), flush=True)
        try:
            proc = subprocess.run(cmd, capture_output=True, text=True)
        except FileNotFoundError as e:
            raise RuntimeError(f"no compiler for the chain {' + '.join(spec.names)}: hiprtc said `{log}`, and hipcc is not installed ({e})") from None
        if proc.returncode != 0 or not tmp.exists():
            raise RuntimeError(f"hipcc failed compiling the chain {' + '.join(spec.names)}:\n{proc.stderr[-2000:]}")
        _publish(out, tmp=tmp)
        src.replace(out.with_suffix(".hip"))
        return out
    finally:
        for leftover in (src, out.with_name(f".{out.name}.{unique}.tmp")):   # a failed compile leaves nothing behind
            if leftover.exists():
                leftover.unlink()


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
    """(indices, indptr, keep, row_off) of the CSR Jacobian with the fixed columns removed (afb:465-489); each detection's index
    row serves its u row and its v row (afb:475-479).  ``keep`` (N,) uint64: bit j = local column j of the detection is free;
    ``row_off`` (N,) int64: offset of its u row in the data array — what the device needs to write the data array directly."""
    mask = np.ones(n_params, dtype=bool) if unfixed is None else np.asarray(unfixed, dtype=bool)
    if mask.shape[0] != n_params:
        raise ValueError("unfixed mask must have one entry per parameter")
    if cols.shape[1] > 64:
        raise NotImplementedError("rows longer than 64 parameters do not fit the 64-bit keep mask")
    conv = np.concatenate([[0], np.cumsum(mask)])
    kept = mask[cols]                                   # (N, P)
    rows2 = np.repeat(cols, 2, axis=0)
    keep2 = np.repeat(kept, 2, axis=0)
    indices = conv[rows2[keep2]].astype(np.int64)
    cnt = kept.sum(axis=1).astype(np.int64)
    indptr = np.concatenate([[0], np.cumsum(np.repeat(cnt, 2))]).astype(np.int64)
    keep = (kept.astype(np.uint64) << np.arange(cols.shape[1], dtype=np.uint64)[None, :]).sum(axis=1).astype(np.uint64)
    row_off = np.concatenate([[0], np.cumsum(2 * cnt)[:-1]]).astype(np.int64) if cols.shape[0] else np.zeros(0, dtype=np.int64)
    return indices, indptr, np.ascontiguousarray(keep), np.ascontiguousarray(row_off)


class ChainEngine:
    """One generated chain on one device — the ``Engine`` interface the operator API uses (NumPy in / NumPy out)."""

    def __init__(self, function_blocks, n_cams: int, n_imgs: int, n_keys: int, *, device: int = 0, dtype: str = "f64"):
        """``dtype`` like Engine's: 'f64', or FP64 arithmetic with float outputs — 'mixed' (double measurements) / 'f32' (float
        measurements too)."""
        if dtype not in ("f64", "f32", "mixed"):
            raise ValueError("dtype must be 'f64', 'f32' or 'mixed'")
        self.dtype = dtype
        self.spec = ChainSpec.from_blocks(function_blocks)
        self.chain = " + ".join(self.spec.names)
        self.n_cams, self.n_imgs, self.n_keys, self.device = int(n_cams), int(n_imgs), int(n_keys), int(device)
        self.P = self.spec.P
        self.lay = self.spec.layout(self.n_cams, self.n_imgs, self.n_keys)
        self.n_params = self.lay["n_params"]
        path = compile_chain(self.spec)
        ng, nu = self.spec.n_rigid_groups, len(self.spec.user_blocks)
        off = (c_int64 * max(1, ng))(*self.lay["rigid_off"])
        cnt = (c_int32 * max(1, ng))(*self.lay["rigid_count"])
        uoff = (c_int64 * max(1, nu))(*self.lay["user_off"])
        self._h = c_void_p()
        check(lib().pcs_genchain_create(byref(self._h), str(path).encode(), self.P, int(self.spec.uses_template), ng, off, cnt, nu, uoff,
                                        self.lay["intr_off"], self.lay["point_off"], self.n_params, self.n_cams, self.n_imgs, self.n_keys,
                                        {"f64": 0, "f32": 1, "mixed": 2}[self.dtype], self.device))
        # which global column a local column stands for (pcs_genchain_matfree)
        nb = len(self.spec.blocks)
        col0 = np.concatenate([[0], np.cumsum([b.n_params for b in self.spec.blocks])[:-1]]).astype(np.int32)
        check(lib().pcs_genchain_set_blocks(self._h, nb, (c_int32 * nb)(*col0), (c_int32 * nb)(*[b.n_params for b in self.spec.blocks]),
                                            (c_int32 * nb)(*[b.link for b in self.spec.blocks]), (c_int64 * nb)(*[self.lay["starts"][b.group] for b in self.spec.blocks])))
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

    @property
    def out_dtype(self):
        return np.float64 if self.dtype == "f64" else np.float32

    def eval(self, param_str, want_resid: bool = True, want_jac: bool = True, pinned_ring: int = 0):
        p = self._check_params(param_str)
        r = np.empty((self.n, 2), dtype=self.out_dtype) if want_resid else None
        j = np.empty((2 * self.n, self.P), dtype=self.out_dtype) if want_jac else None
        check(lib().pcs_genchain_eval(self._h, p.ctypes.data_as(POINTER(c_double)), c_void_p(r.ctypes.data if want_resid else 0), c_void_p(j.ctypes.data if want_jac else 0)))
        return r, j

    def eval_device(self, d_param_str: int, d_resid: int | None, d_jac: int | None, stream: int | None = None):
        from .engine import _stream_arg

        check(lib().pcs_genchain_eval_device(self._h, c_void_p(d_param_str), c_void_p(d_resid or 0), c_void_p(d_jac or 0), _stream_arg(stream)))

    def eval_compact_device(self, d_param_str: int, d_resid: int | None, d_data: int, stream: int | None = None):
        """Only the unfixed columns (set_unfixed), in CSR data order, at the device-resident parameter string."""
        from .engine import _stream_arg

        check(lib().pcs_genchain_eval_compact_device(self._h, c_void_p(d_param_str), c_void_p(d_resid or 0), c_void_p(d_data), _stream_arg(stream)))

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

    # -- products with the Jacobian kept on the device (the interface device_solver.JacobianOperator drives; csrc/ba_blockrow.hpp) ------
    OP_JV, OP_JTU, OP_JTJV, OP_DIAG, OP_GRAD = 0, 1, 2, 3, 4

    def linearize(self, param_str):
        """Residual + dense block rows at ``param_str`` into the handle's device buffers; the products below refer to this point."""
        check(lib().pcs_genchain_linearize(self._h, self._check_params(param_str).ctypes.data_as(POINTER(c_double))))

    def _matfree(self, op: int, vin, n_out: int, want_cost: bool = False):
        out = np.empty(n_out)
        cost = c_double(0.0)
        dp = POINTER(c_double)
        check(lib().pcs_genchain_matfree(self._h, op, vin.ctypes.data_as(dp) if vin is not None else None, out.ctypes.data_as(dp), byref(cost) if want_cost else None))
        return (out, float(cost.value)) if want_cost else out

    def jv(self, v) -> np.ndarray:
        return self._matfree(self.OP_JV, self._check_params(v), 2 * self.n)

    def jtu(self, u) -> np.ndarray:
        u = np.ascontiguousarray(u, dtype=np.float64).ravel()
        if u.shape[0] != 2 * self.n:
            raise ValueError("u must have 2N entries")
        return self._matfree(self.OP_JTU, u, self.n_params)

    def jtjv(self, v) -> np.ndarray:
        return self._matfree(self.OP_JTJV, self._check_params(v), self.n_params)

    def jtj_diag(self) -> np.ndarray:
        return self._matfree(self.OP_DIAG, None, self.n_params)

    def grad(self) -> tuple[np.ndarray, float]:
        """(J^T r, sum r^2) at the linearisation point."""
        return self._matfree(self.OP_GRAD, None, self.n_params, want_cost=True)

    # -- the exact LM step: dense normal equations + the device-steered trial (csrc/ba_blockgram.hpp; device_solver.BlockedNormalEquations) ----
    lm_fixed_trial_buffer = True   # the generated kernel reads its string from a fixed address: trials are built at ps[1] into packed[1]
    DENSE_OPTIONS = ("spd_timeout_us", "timing", "gram_debug", "dense_normal")

    def deterministic_supported(self) -> bool:
        """The ordered sums write every destination from ONE pair of local columns: no two blocks may share a parameter group."""
        return len(set(self.spec.group_of_block)) == len(self.spec.blocks)

    def dense_lm_supported(self) -> bool:
        """Does the contraction of csrc/ba_blockgram.hpp take this chain?  FP64 block rows of at most 63 columns, n_params <= 65 535."""
        return self.dtype == "f64" and self.P + 1 <= 64 and self.n_params <= 65535

    def normal_layout(self) -> dict:
        """{n_lead, n_trail, tb, packed_len, n_params} of the packed state [A | B | C | g | cost] (include/pcs_hip.h): the chain's LAST
        parameter group as trailing entities when it is one rigid transform per image or one point per key (Schur step on the leading
        part), else — or with ``set_option("dense_normal", 1)`` — every parameter leading and A dense."""
        out = (c_int64 * 5)()
        check(lib().pcs_genchain_normal_layout(self._h, out))
        return dict(n_lead=int(out[0]), n_trail=int(out[1]), tb=int(out[2]), packed_len=int(out[3]), n_params=int(out[4]))

    def normal_blocks_device(self, d_param_str: int, d_packed: int, stream: int | None = None):
        """[J^T J | J^T r | sum r^2] at the DEVICE-resident parameter string; asynchronous, zeroes the buffer first."""
        from .engine import _stream_arg

        check(lib().pcs_genchain_normal_blocks_device(self._h, c_void_p(d_param_str), c_void_p(d_packed), _stream_arg(stream)))

    def lm_trial(self, buffers, stream=None):
        from .engine import _stream_arg

        check(lib().pcs_genchain_lm_trial(self._h, byref(buffers), _stream_arg(stream)))

    def lm_trial_build(self, buffers, stream=None):
        from .engine import _stream_arg

        check(lib().pcs_genchain_lm_trial_build(self._h, byref(buffers), _stream_arg(stream)))

    def lm_trial_finish(self, buffers, stream=None):
        from .engine import _stream_arg

        check(lib().pcs_genchain_lm_trial_finish(self._h, byref(buffers), _stream_arg(stream)))

    def schur_prepare(self, d_packed, d_fixed, d_lambda, d_linvt, d_u, d_V, d_S, d_rhs, d_dvec, d_gm, d_status, stream=None):
        from .engine import _stream_arg

        check(lib().pcs_genchain_schur_prepare(self._h, *(c_void_p(p) for p in (d_packed, d_fixed, d_lambda, d_linvt, d_u, d_V, d_S, d_rhs, d_dvec, d_gm, d_status)),
                                               _stream_arg(stream)))

    def schur_finish(self, d_linvt, d_u, d_w, d_xlead, d_fixed, d_delta, d_ps_in=0, d_ps_out=0, stream=None):
        from .engine import _stream_arg

        check(lib().pcs_genchain_schur_finish(self._h, *(c_void_p(p) for p in (d_linvt, d_u, d_w, d_xlead, d_fixed, d_delta, d_ps_in, d_ps_out)), _stream_arg(stream)))

    def lm_decide(self, d_cost_old, d_cost_new, d_dvec, d_gm, d_delta, d_ps, d_fixed, d_status, d_lambda, d_stats, stream=None):
        from .engine import _stream_arg

        check(lib().pcs_genchain_lm_decide(self._h, *(c_void_p(p) for p in (d_cost_old, d_cost_new, d_dvec, d_gm, d_delta, d_ps, d_fixed, d_status, d_lambda, d_stats)),
                                           _stream_arg(stream)))

    def set_option(self, key: str, value: int):
        """Engine's option interface as far as the LM driver uses it: "spd_timeout_us", "timing" and "deterministic" (the ORDERED
        contraction of csrc/ba_blockgram.hpp: every sum in a fixed order, the same bits on every run) reach the handle; "timing_every"
        maps to "timing"; "lazy_done_event" has nothing to switch here."""
        value = int(value)
        if key == "deterministic":
            if value and not self.deterministic_supported():
                raise NotImplementedError("deterministic mode: blocks that share a parameter group are not supported in the ordered sums (csrc/ba_blockgram.hpp)")
            check(lib().pcs_genchain_set_option(self._h, b"deterministic", value))
        elif key == "timing_every":
            check(lib().pcs_genchain_set_option(self._h, b"timing", int(value != 0)))
        elif key in self.DENSE_OPTIONS:
            check(lib().pcs_genchain_set_option(self._h, key.encode(), value))
        elif key != "lazy_done_event":
            raise ValueError(f"unknown option '{key}' for a generated chain")
        self.__dict__.setdefault("_options", {})[key] = value

    def option(self, key: str, default):
        return self.__dict__.get("_options", {}).get(key, default)

    # -- static structure (integer work on the host, like the reference's afb:192-233, afb:465-489) -------------------------
    def block_param_inds(self) -> np.ndarray:
        if self._det is None:
            raise RuntimeError("no detections set")
        return block_param_inds(self.spec, self.lay, self._det)

    def csr_structure(self, unfixed=None):
        indices, indptr, _, _ = csr_structure_of(self.block_param_inds(), self.n_params, unfixed)
        return indices, indptr

    def set_one_launch(self, on: bool) -> None:
        """A step is one launch by default (every wave prepares the Rodrigues slabs of its tile itself, like Engine's fused kernel);
        ``False`` runs the slab preparation as a launch of its own in front of the evaluation.  The outputs hold the same bits."""
        check(lib().pcs_genchain_set_one_launch(self._h, int(bool(on))))

    def set_unfixed(self, unfixed) -> int:
        _, indptr, keep, row_off = csr_structure_of(self.block_param_inds(), self.n_params, unfixed)
        nnz = int(indptr[-1])
        check(lib().pcs_genchain_set_unfixed(self._h, keep.ctypes.data_as(POINTER(c_uint64)), row_off.ctypes.data_as(POINTER(c_int64)), nnz))
        self.nnz = nnz
        self.mask_key = None if unfixed is None else hash(np.asarray(unfixed, dtype=bool).tobytes())
        return self.nnz

    def eval_compact(self, param_str, want_resid: bool = False, pinned_ring: int = 0):
        if self.nnz is None:
            raise RuntimeError("call set_unfixed() first")
        p = self._check_params(param_str)
        r = np.empty((self.n, 2), dtype=self.out_dtype) if want_resid else None
        d = np.empty(self.nnz, dtype=self.out_dtype)
        check(lib().pcs_genchain_eval_compact(self._h, p.ctypes.data_as(POINTER(c_double)), c_void_p(r.ctypes.data if want_resid else 0), c_void_p(d.ctypes.data)))
        return r, d
