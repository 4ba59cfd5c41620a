#!/usr/bin/env python3
"""bench.py — residual + Jacobian rows/s of the bundle-adjustment hot path on N MI355X.

    python bench.py --gpus N --steps K --warmup W

A *step* is one LM-step evaluation of the hot path over the detection table, with the parameter string and the
detection table already resident in HBM and the residual / dense Jacobian blocks left in HBM.  For FP64 outputs on a
table in the reference's run order (every BASELINE config, the 1e6-detection headline included) that is ONE launch:
the waves of the fused residual/Jacobian kernel (K1) prepare the slabs of their own tiles (the same element functions
slab_prep, K0, is made of).  Float outputs above 2.5e5 detections and shuffled tables take K0 + K1 (two launches).

Workload: BASELINE.json configs[2], "rig-32" — 32 cameras, Ccube target (486 keys), 200 poses, ~1.0e6
detections, template chain, FP64; `--config 4 / 5` select the self-calibration and the
1e7-detection FP32 rigs.

N > 1 (one process per GPU, RCCL backend): either under a launcher (`python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`) or plainly as `python bench.py
--gpus N ...` — without WORLD_SIZE in the environment the process starts that launcher itself as a child
(`self_launch`, before torch or HIP are touched) and relays rank 0's line.  `--scaling strong` (default) —
every rank evaluates its contiguous ceil(N/G)-row shard of the ONE rig the config names (the reference's
equal-chunk rule incl. cyclic padding, abstract_function_blocks.py:281-288; `pycamset_amd.sharding.padded_shard`),
all ranks hold the same parameter string, and the timed region contains no collective: the path partitions by
observation.  `value` = rows of the whole rig x steps / max-over-ranks time.  Two more step times are
measured outside that region and reported under "multi_gpu": step + RCCL all-gather of the residual / Jacobian
blocks (what a host-side consumer of J needs; xGMI-link-bound) and normal-equations build + ONE all-reduce of the
packed [J^T J, J^T r, cost] (what a device-side LM step needs).  `--scaling weak` gives every rank an independent
rig of the same shape instead (seed + 1000 r).

Kernel time for the roofline figure: after the wall-clock region, >= 20 extra launches carry HIP start/stop
events (hipExtLaunchKernelGGL on the launch stream); mean / median / min are reported, the mean is what
`roofline.achieved` uses.  The timed region itself runs without events, so they cannot perturb `value`.

Rank 0 prints ONE JSON line (schema: task contract + "roofline" + "cpu_baseline").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

BYTES_PER_DET = {("template", "f64"): 380, ("self", "f64"): 428, ("free", "f64"): 332,
                 ("template", "f32"): 196, ("self", "f32"): 220, ("free", "f32"): 172,
                 # mixed (FP64 arithmetic and measurements, FP32 outputs): 28 B in like f64, outputs like f32
                 ("template", "mixed"): 204, ("self", "mixed"): 228, ("free", "mixed"): 180}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
CONFIG_CHAIN = {1: "template", 2: "template", 3: "template", 4: "self", 5: "template"}
CONFIG_DTYPE = {1: "f64", 2: "f64", 3: "f64", 4: "f64", 5: "f32"}
KERNEL_SAMPLES = 24     # event-timed launches after the wall-clock region (>= 20)


def chain_slabs(rig, chain):
    if chain == "template":
        return [rig.intr, rig.extr, rig.poses]
    if chain == "self":
        return [rig.intr, rig.extr, rig.poses, rig.points]
    return [rig.intr, rig.extr, rig.points]


# ---------------------------------------------------------------------------------------------------------------
# the per-rank problem (importable without a GPU: tests/test_sharding_gloo.py drives exactly this code)
# ---------------------------------------------------------------------------------------------------------------
def rank_problem(config: int, chain: str, rank: int, world: int, scaling: str = "strong", scale: float = 1.0):
    """What rank `rank` of `world` evaluates.  -> dict(rig, det, n_real, n_total, per, param_str, template)
    strong: the rank's `padded_shard` of the one config rig (rows beyond the table are cyclic repeats, dropped
            again by `strip_padding`); weak: an independent rig of the same shape per rank."""
    from pycamset_amd import sharding, synthetic

    if scaling == "strong":
        rig = synthetic.config_rig(config, scale=scale)
        n_total = rig.n_det
        per = sharding.shard_rows(n_total, world)
        det = sharding.padded_shard(rig.detections, rank, world)
        n_real = max(0, min(n_total, (rank + 1) * per) - rank * per)
    elif scaling == "weak":
        rig = synthetic.config_rig(config, scale=scale, block=rank)
        det, n_real, per, n_total = rig.detections, rig.n_det, rig.n_det, None   # n_total: sum over ranks (all-reduced)
    else:
        raise ValueError("scaling must be 'strong' or 'weak'")
    ps = np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in chain_slabs(rig, chain)])
    return dict(rig=rig, det=det, n_real=n_real, n_total=n_total, per=per, param_str=ps,
                template=rig.points if chain == "template" else None)


def gather_blocks(dist, local, out=None, via_host: bool = False):
    """All-gather equal-sized blocks (rows x cols tensors) from every rank into one (world * rows, cols) tensor.
    RCCL: `all_gather_into_tensor` on the device.  via_host (the gloo rehearsal on one GPU): staged through the host."""
    import torch

    world = dist.get_world_size()
    if via_host:
        loc = local.detach().cpu().contiguous()
        buf = torch.empty((world * loc.shape[0],) + tuple(loc.shape[1:]), dtype=loc.dtype)
        dist.all_gather_into_tensor(buf, loc)
        if out is None:
            return buf.to(local.device)
        out.copy_(buf)
        return out
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def strip_padding(gathered, n_total: int, rows_per_det: int = 1):
    """Contiguous shards in rank order: the real rows are the leading n_total (x rows_per_det) of the gathered
    block, everything after them is the cyclic padding of the last rank(s) (afb:385, afb:641)."""
    return gathered[: n_total * rows_per_det]


def usable_cpus() -> int:
    """Host threads this process may actually use: min(affinity mask, cgroup CPU quota, cpu_count).
    (A GPU box hands each job a CPU share well below os.cpu_count().)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(rig, chain, param_str, budget_s: float):
    """Time the CPU oracle (the repo's C restatement of the reference's per-detection algorithm,
    -O3 AVX2/FMA + OpenMP) on the host cores, on the same workload.  kind = "port"."""
    from oracle import ba_oracle as orc

    orc.build()
    threads = int(os.environ.get("PCS_CPU_THREADS", "0")) or usable_cpus()
    tm = rig.points if chain == "template" else None
    counts = (rig.n_cams, rig.n_imgs, rig.n_keys)
    n_sample = min(rig.n_det, 1_000_000)
    det = rig.detections[:n_sample]
    orc.full_jac_dense(chain, det[:50_000], param_str, tm, threads=threads, fast=True, with_resid=True, counts=counts)  # warm
    passes, t0 = 0, time.perf_counter()
    while True:
        orc.full_jac_dense(chain, det, param_str, tm, threads=threads, fast=True, with_resid=True, counts=counts)
        passes += 1
        el = time.perf_counter() - t0
        if el >= 0.6 * budget_s or passes >= 200:
            break
    out = {
        "value": 2.0 * n_sample * passes / el,
        "unit": "rows/s",
        "cores": threads,
        "kind": "port",
        "host_cpu_count": os.cpu_count(),
        "sample": f"{passes} full residual+Jacobian passes over the first {n_sample} detections of the same rig "
                  f"({el:.1f} s, oracle/libba_oracle_fast.so, OpenMP static schedule)",
    }
    # the reference's own default thread count, min(max(1, cpu_count() - 2), 20) (camera_calibrator.py:57-58)
    ref_threads = min(max(1, threads - 2), 20)
    t0, p2 = time.perf_counter(), 0
    while time.perf_counter() - t0 < min(3.0, budget_s / 4):
        orc.full_jac_dense(chain, det, param_str, tm, threads=ref_threads, fast=True, with_resid=True, counts=counts)
        p2 += 1
    out["reference_default_threads"] = {"threads": ref_threads, "value": 2.0 * n_sample * p2 / (time.perf_counter() - t0)}
    # NumPy-vectorised twin, 1 core, on a 2e5-detection slice (readability cross-check)
    from oracle import ba_oracle_np as onp

    n_np = min(n_sample, 200_000)
    t0 = time.perf_counter()
    onp.evaluate(chain, det[:n_np], param_str, tm, counts=counts)
    out["numpy_vectorised_1core"] = {"value": 2.0 * n_np / (time.perf_counter() - t0), "detections": n_np}
    return out


def lm_end_to_end(config: int, with_scipy: bool, scipy_nfev: int = 8):
    """End-to-end solve of the config's calibration problem through the drop-in boundary (outside the timed region,
    never part of `value`):
      device  pycamset_amd.optimisation_handling.run_bundle_adjustment(handler, solver='device') — block-reduced normal
              equations + Schur / Cholesky on the GPU, J never leaves HBM (row f2);
      scipy   the reference's own call, scipy.optimize.least_squares(loss_fn, x0, jac=jac_fn, x_scale='jac')
              (optimisation_handling.py:88-98) driven by the HIP closures with the default page-locked output ring —
              bounded by `scipy_nfev` evaluations because it is PCIe- and lsmr-bound (only with --lm-compare)."""
    from scipy.optimize import least_squares

    from pycamset_amd import handlers, synthetic
    from pycamset_amd.detections import TargetDetection
    from pycamset_amd.device_solver import lm_solve

    class _Camset:
        def __init__(self, n):
            self.names = [f"cam_{i}" for i in range(n)]

        def get_names(self):
            return list(self.names)

        def get_n_cams(self):
            return len(self.names)

    class _Target:
        def __init__(self, pts):
            self.point_data = np.asarray(pts)[None]

    rig = synthetic.config_rig(config)
    cs = _Camset(rig.n_cams)
    h = handlers.TemplateBundleHandler(cs, _Target(rig.points), TargetDetection(cs.get_names(), rig.detections),
                                       fixed_params={"cam_0": {"ext": rig.extr_true[0].copy()}}, options={"verbosity": 0, "max_nfev": 30})
    bp = h.bundlePrimitive
    x0 = np.concatenate([rig.intr[bp.intr_unfixed].ravel(), rig.extr[bp.extr_unfixed].ravel(), rig.poses[bp.poses_unfixed].ravel()])
    loss_fn = h.make_loss_fun(1)
    err = lambda r: float(np.mean(np.linalg.norm(np.asarray(r).reshape(-1, 2), axis=1)))  # noqa: E731
    out = {"workload": f"{rig.name}: {rig.n_det} detections, {x0.size} free parameters", "start_error_px": err(loss_fn(x0))}
    lm_solve(h, x0.copy(), max_iter=2, linear_solver="cholesky")   # first-call set-up (solver workspace, visiting orders, page-locked read-back), once per engine
    t0 = time.perf_counter()
    dev = lm_solve(h, x0.copy(), max_iter=30, linear_solver="cholesky")
    out["device_lm"] = {"seconds": time.perf_counter() - t0, "iterations": dev.nit, "nfev": dev.nfev, "cost": dev.cost,
                        "final_error_px": err(loss_fn(dev.x)), "status": dev.message}
    if with_scipy:
        jac_fn = h.make_loss_jac(1)
        t0 = time.perf_counter()
        res = least_squares(loss_fn, x0.copy(), jac=jac_fn, x_scale="jac", max_nfev=scipy_nfev, verbose=0)
        out["scipy_hip_closures"] = {"seconds": time.perf_counter() - t0, "nfev": int(res.nfev), "njev": int(res.njev), "cost": float(res.cost),
                                     "final_error_px": err(res.fun), "max_nfev": scipy_nfev,
                                     "note": "least_squares(..., x_scale='jac') -> trf + lsmr on the host; every Jacobian crosses PCIe as CSR"}
    return out


def pmc_traffic(workload_key: str):
    """(HBM bytes per launch, where they come from): the committed rocprofv3 PMC summary of this same command —
    FETCH_SIZE and WRITE_SIZE need passes of their own (MI355X_MICROARCH.md), so they cannot be taken inside this run."""
    f = REPO / "profiles" / "pmc_traffic.json"
    if f.exists():
        try:
            e = json.loads(f.read_text()).get(workload_key)
            if e:
                return e.get("hbm_bytes_per_launch"), f"profiles/pmc_traffic.json['{workload_key}'] <- {e.get('source')}"
        except Exception:
            pass
    return None, None


def self_launch(n_ranks: int, argv, script=None, python=None, extra_env=None) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILDREN
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py argv`),
    pass rank 0's JSON line (every stdout line that parses as a JSON object) through to this process's stdout, everything else
    to stderr, and return the launcher's exit code (non-zero if any rank failed).  The parent never imports torch and never
    touches HIP: the children are fresh processes, not a re-exec of one that has initialised the GPU."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.update(extra_env or {})
    cmd = [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script or Path(__file__).resolve()), *argv]
    print("[bench] self-launch: " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for raw in proc.stdout:
        txt = raw.rstrip("\n")
        # rank 0's JSON object, also when another rank's (or a library's) unterminated output shares its line: the ranks write to ONE pipe
        is_line = False
        at = txt.find('{"')
        if at >= 0:
            try:
                is_line = isinstance(json.loads(txt[at:]), dict)
            except ValueError:
                pass
        if is_line and at > 0:
            print(txt[:at], file=sys.stderr, flush=True)
            txt = txt[at:]
        print(txt, file=sys.stdout if is_line else sys.stderr, flush=True)
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.md config number (3 = headline rig-32)")
    ap.add_argument("--chain", default=None)
    ap.add_argument("--dtype", default=None)
    ap.add_argument("--scale", type=float, default=1.0, help="visibility scale (<1: smaller N, for quick checks)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: 'strong' shards the ONE rig of the config over the ranks (default), 'weak' gives every rank its own rig")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="one-GPU projection of strong scaling: evaluate only rank 0's shard of a K-way split of the rig "
                         "(value then counts that shard's rows only; reported under config.emulated_world)")
    ap.add_argument("--collective", default="none", choices=["none", "allgather"],
                    help="'allgather' puts the RCCL all-gather of residual+Jacobian blocks inside the timed step")
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--wgs-per-cu", type=int, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--device-warmup-ms", type=float, default=500.0,
                    help="plain streaming fills before the W warm-up steps, so that the timed region sees the GPU's steady-state clocks (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allgather-probe", action="store_true")
    ap.add_argument("--no-normal-probe", action="store_true")
    ap.add_argument("--stream-to-host", action="store_true",
                    help="config-5 mode: every step also copies the Jacobian to page-locked host memory on a side "
                         "stream (double buffered); the step rate is then PCIe-bound and reported as such")
    ap.add_argument("--lm-compare", action="store_true",
                    help="also solve configs 2 and 3 end to end with scipy least_squares on the HIP closures (tens of seconds); "
                         "the device LM solve of the bench's own rig is always reported under 'lm_end_to_end'")
    ap.add_argument("--host-path", action="store_true",
                    help="also time the host-buffer boundary (pcs_eval: H2D params + kernels + D2H of residual and Jacobian); "
                         "reported under 'host_boundary', never in 'value'")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # no launcher around us: become one (before torch / HIP are touched in this process)
            raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")

    import torch
    import torch.distributed as dist

    from pycamset_amd.engine import Engine

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # PCS_BENCH_BACKEND=gloo lets the N > 1 code path be rehearsed on a one-GPU box (ranks share the card)
    backend = os.environ.get("PCS_BENCH_BACKEND", "nccl")
    via_host = backend != "nccl"
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    chain = args.chain or CONFIG_CHAIN[args.config]
    dtype = args.dtype or CONFIG_DTYPE[args.config]
    prob = rank_problem(args.config, chain, rank, world, args.scaling, args.scale)
    if args.emulate_world > 1 and world == 1:
        prob = rank_problem(args.config, chain, 0, args.emulate_world, "strong", args.scale)
        prob["n_total"] = prob["n_real"]          # only this shard is evaluated here
    rig, det, ps = prob["rig"], prob["det"], prob["param_str"]
    N = det.shape[0]            # rows this rank evaluates (incl. cyclic padding of the last shard)

    eng = Engine(chain, rig.n_cams, rig.n_imgs, rig.n_keys, dtype=dtype, device=local_rank)
    eng.set_detections_table(det)
    if chain == "template":
        eng.set_template(rig.points)
    if args.variant is not None:
        eng.set_option("variant", args.variant)
    if args.wgs_per_cu is not None:
        eng.set_option("wgs_per_cu", args.wgs_per_cu)
    P = eng.P
    tdt = torch.float64 if dtype == "f64" else torch.float32
    d_r = torch.empty((N, 2), dtype=tdt, device=dev)
    d_j = torch.empty((2 * N, P), dtype=tdt, device=dev)
    d_p = torch.from_numpy(ps).to(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def reduce_scalar(x: float, op):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=op)
        return float(t.item())

    # equal block sizes for the all-gather (strong: already equal; weak: pad to the largest rank)
    per = int(reduce_scalar(float(N), dist.ReduceOp.MAX)) if world > 1 else N
    n_total = prob["n_total"] if prob["n_total"] is not None else int(reduce_scalar(float(prob["n_real"]), dist.ReduceOp.SUM))
    want_gather = world > 1 and (args.collective == "allgather" or not args.no_allgather_probe)
    if want_gather:
        send_r = d_r if per == N else torch.zeros((per, 2), dtype=tdt, device=dev)
        send_j = d_j if per == N else torch.zeros((2 * per, P), dtype=tdt, device=dev)
        g_r = torch.empty((world * per, 2), dtype=tdt, device="cpu" if via_host else dev)
        g_j = torch.empty((world * 2 * per, P), dtype=tdt, device="cpu" if via_host else dev)

        def gather():
            if per != N:
                send_r[:N].copy_(d_r)
                send_j[: 2 * N].copy_(d_j)
            gather_blocks(dist, send_r, g_r, via_host)
            gather_blocks(dist, send_j, g_j, via_host)

    use_gather_in_step = args.collective == "allgather" and world > 1

    streamer = None
    if args.stream_to_host:
        # two device buffers + two page-locked host buffers: the copy of step i overlaps the kernel of step i + 1
        from pycamset_amd.host_stream import JacobianHostStreamer

        streamer = JacobianHostStreamer(eng, N, device=local_rank, first_device_buffer=d_j)

    def step():
        if streamer is not None:
            streamer.step(d_p.data_ptr(), d_r.data_ptr())
            return
        eng.eval_device_resident(d_p.data_ptr(), d_r.data_ptr(), d_j.data_ptr(), stream)
        if use_gather_in_step:
            gather()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    region_gpu_ms = [None]   # GPU time of the last timed(..., bracket=True) region per call: two events on the launch stream

    def timed(fn, reps, bracket=False):
        """max-over-ranks seconds per call of `fn`, bracketed by barrier + synchronize on both sides.  ``bracket``: also two HIP
        events on the launch stream around the calls (this rank's GPU time per call, gaps between launches included)."""
        fence()
        e0 = e1 = None
        if bracket:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if bracket:
            e0.record()
        for _ in range(reps):
            fn()
        if bracket:
            e1.record()
        torch.cuda.synchronize(dev)
        if bracket:
            region_gpu_ms[0] = e0.elapsed_time(e1) / reps
        if world > 1:
            dist.barrier()
        return reduce_scalar(time.perf_counter() - t0, dist.ReduceOp.MAX) / reps

    # ---- device warm-up (not steps): a fresh process finds the GPU in its idle power state, and the W warm-up steps of a
    # 65 us kernel (a millisecond or two) are over long before its memory and fabric clocks have ramped — measured on MI355X:
    # the same step 75 us right after start-up, 65 us 20 ms later, 63.5 us in a process that has been busy for seconds
    # (profiles/r03/README.md).  A fixed stretch of plain non-temporal fills brings the part to its steady state first.
    if args.device_warmup_ms > 0:
        from pycamset_amd import _capi
        import ctypes

        ms = ctypes.c_float()
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) * 1e3 < args.device_warmup_ms:
            # half of the stretch: streaming fills (clocks); the other half: residual-only evaluations through the engine's own
            # launch path on the bench stream (the runtime grows its signal / kernel-argument pools on the first long burst of
            # launches — inside the timed region that cost 7 us per step at K = 200)
            _capi.check(_capi.lib().pcs_membench(local_rank, 1, 256 << 20, 100, 8, ctypes.byref(ms)))
            for _ in range(512):
                eng.eval_device_resident(d_p.data_ptr(), d_r.data_ptr(), None, stream)
            torch.cuda.synchronize(dev)

    # the part's sustained write rate TODAY, in the fused kernel's store shape (non-temporal 16-byte stores, whole chunks per wave),
    # back to back like the steps: boxes of this pool differ by 20 % in it (4.5-5.6 TB/s), and it is the bound of this path
    nt_fill_GBps = None
    try:
        from pycamset_amd import _capi
        import ctypes

        ms_f = ctypes.c_float()
        fill_bytes = max(1 << 20, (int(N) * int(2 * P + 2) * (8 if dtype == "f64" else 4)) // 4096 * 4096)
        _capi.check(_capi.lib().pcs_membench(local_rank, 4, fill_bytes, 50, 8, ctypes.byref(ms_f)))
        if ms_f.value > 0:
            nt_fill_GBps = fill_bytes / (ms_f.value * 1e-3) / 1e9
        torch.cuda.synchronize(dev)
    except Exception as e:  # a probe: its failure must not cost the bench line
        print(f"[bench] membench probe failed: {e}", file=sys.stderr)

    # ---- the timed region: K steps, no HIP events attached to the launches ---------------------------------------
    eng.set_option("timing_every", 0)
    for _ in range(args.warmup):
        step()
    sec_per_step = timed(step, args.steps, bracket=True)
    elapsed = sec_per_step * args.steps

    # ---- kernel durations: >= 20 extra launches with start/stop events, outside the wall-clock region ------------
    eng.set_option("timing_every", 1)
    eng.set_option("event_ring", KERNEL_SAMPLES)
    for _ in range(2 * KERNEL_SAMPLES):   # the first batch only switches the queue to timed dispatches (its launches run 5-10 % long); the ring keeps the second
        eng.eval_device_resident(d_p.data_ptr(), d_r.data_ptr(), d_j.data_ptr(), stream)
    torch.cuda.synchronize(dev)
    prep_s, eval_s = eng.kernel_ms_samples(KERNEL_SAMPLES)
    eng.set_option("timing_every", 0)
    iso_ms, prep_ms = float(np.mean(eval_s)), float(np.mean(prep_s))
    # The kernel's average duration OVER THE TIMED REGION: the GPU time between two events that bracket the K steps on the launch
    # stream, per step, minus slab_prep's own duration where a step has that second launch.  It contains the ~1-3 us between
    # consecutive launches — an upper bound of the kernel's duration, and the figure that agrees with `rocprofv3 --stats` of
    # this command.  The launches that carry their own start / stop events (`kernel_ms_isolated*`, after the region) follow a
    # timed dispatch's ~12 us gap, find the memory system drained and run 4-9 % faster: reported, not used for `frac`.
    in_region = region_gpu_ms[0] is not None and streamer is None and not use_gather_in_step
    eval_ms = max(region_gpu_ms[0] - prep_ms, 0.0) if in_region else iso_ms

    # ---- multi-GPU extras, outside the timed region --------------------------------------------------------------
    multi = None
    if world > 1:
        multi = {"kernel_only_step_ms": sec_per_step * 1e3, "shard_rows_per_gpu": N, "collective_backend": backend}
    if want_gather and not use_gather_in_step:
        try:  # a probe: its failure must not cost the bench line
            def step_gather():
                eng.eval_device_resident(d_p.data_ptr(), d_r.data_ptr(), d_j.data_ptr(), stream)
                gather()

            for _ in range(2):
                step_gather()
            reps = max(3, min(10, args.steps))
            t_sg = timed(step_gather, reps)
            t_g = timed(gather, reps)
            sent = (2 * per + 2 * per * P) * d_r.element_size()
            multi["step_plus_allgather_ms"] = t_sg * 1e3
            multi["allgather"] = {"ms": t_g * 1e3, "bytes_sent_per_gpu": sent, "bytes_received_per_gpu": sent * (world - 1),
                                  "recv_GBps_per_gpu": sent * (world - 1) / t_g / 1e9,
                                  "note": ("gloo rehearsal: staged through the host" if via_host else
                                           "RCCL all_gather_into_tensor of residual + Jacobian blocks (equal counts, afb:281-288 padding)")}
        except Exception as exc:  # noqa: BLE001
            multi["allgather"] = {"error": f"{type(exc).__name__}: {exc}"}

    # what a solver needs instead of the gathered J — the block-reduced normal equations per rank and
    # ONE all-reduce of the packed [J^T J, J^T r, cost]
    normal_info = None
    from pycamset_amd.device_solver import blocked_fits

    if not args.no_normal_probe and blocked_fits(eng):
        try:
            npar = eng.n_params
            lay = eng.normal_layout()
            # the reduction must count every row of the rig ONCE: the last ranks' shards are padded by cyclic repeats
            # (sharding.padded_shard), so the probe runs on the rank's REAL rows (possibly none: zeros enter the all-reduce)
            eng_n, n_rows_n = eng, N
            if prob["n_real"] != N:
                n_rows_n = prob["n_real"]
                eng_n = None
                if n_rows_n > 0:
                    eng_n = Engine(chain, rig.n_cams, rig.n_imgs, rig.n_keys, dtype=dtype, device=local_rank)
                    eng_n.set_detections_table(det[:n_rows_n])
                    if chain == "template":
                        eng_n.set_template(rig.points)
            # blocked form [A | B | C | g | cost] (include/pcs_hip.h): leading x leading, leading x trailing, block-diagonal trailing group
            packed = torch.empty(lay["packed_len"], dtype=torch.float64, device=dev)
            def build():
                if eng_n is None:
                    packed.zero_()
                else:
                    eng_n.normal_blocks_device(d_p.data_ptr(), packed.data_ptr(), stream)

            k_ms = [0.0]
            if eng_n is not None:
                eng_n.set_option("timing_every", 1)
                eng_n.set_option("event_ring", 8)
                for _ in range(2):
                    build()
                torch.cuda.synchronize(dev)
                k_ms = []
                for _ in range(5):
                    build()
                    torch.cuda.synchronize(dev)
                    k_ms.append(eng_n.last_kernel_ms()[1])
                eng_n.set_option("timing_every", 0)

            def build_reduce():
                build()
                if via_host:
                    t = packed.cpu()
                    dist.all_reduce(t)
                    packed.copy_(t)
                else:
                    dist.all_reduce(packed)

            normal_info = {"n_params": npar, "kernel_ms": float(np.median(k_ms)), "build_call_ms": timed(build, 5) * 1e3,
                           "bytes": packed.numel() * 8, "rows_this_rank": 2 * n_rows_n,
                           "layout": {k: lay[k] for k in ("n_lead", "n_trail", "tb")},
                           "note": "ba_normal_mfma_kernel (+ point passes for the self / free chains): J^T J in blocked form [A | B | C], g, cost; "
                                   "J never written; parameter string resident in HBM"}
            if world > 1:
                for _ in range(2):
                    build_reduce()
                multi["step_plus_normal_allreduce_ms"] = timed(build_reduce, 5) * 1e3
            if eng_n is not None and eng_n is not eng:
                eng_n.close()
        except Exception as exc:  # noqa: BLE001
            normal_info = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        bpd = BYTES_PER_DET[(chain, dtype)]
        achieved = N * bpd / (eval_ms * 1e-3) / 1e9
        name = rig.name.split("/")[0]
        traffic, traffic_source = pmc_traffic(f"{name}/{chain}/{dtype}") if (world == 1 and args.scale == 1.0 and args.emulate_world <= 1) else (None, None)
        in_bytes = 20 if dtype == "f32" else 28
        shard_txt = (f"{n_total} detections" if world == 1 else
                     f"{n_total} detections sharded into contiguous blocks of {per} per GPU" if args.scaling == "strong" else
                     f"{n_total} detections = one independent rig of {N} per GPU")
        line = {
            "metric": "residual+Jacobian rows/sec",
            "value": 2.0 * n_total * args.steps / elapsed,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{name} (BASELINE config {args.config}): {rig.n_cams} cams x {rig.n_imgs} poses x {rig.n_keys} keys, "
                            f"{shard_txt}, chain {chain}, "
                            + ("ONE launch per step (fused residual/Jacobian kernel whose waves prepare their slabs)" if prep_ms == 0.0
                               else "slab_prep + fused residual/Jacobian kernel per step"),
                "detections_total": n_total,
                "detections_per_gpu": N,
                "rows_per_step": 2.0 * n_total,
                "row_len_P": P,
                "collective_in_step": args.collective if world > 1 else "none",
                "device_warmup_ms": args.device_warmup_ms,
                "jacobian_streamed_to_host": bool(args.stream_to_host),
                "parallelism": f"obs-shard x{world}",
                **({"emulated_world": args.emulate_world} if args.emulate_world > 1 and world == 1 else {}),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": "ba_eval_kernel",
                "kernel_ms": eval_ms,
                "kernel_ms_isolated": iso_ms,
                "frac_isolated": N * bpd / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,   # what rounds 1-2 reported as `frac`
                "kernel_ms_isolated_median": float(np.median(eval_s)),
                "kernel_ms_isolated_min": float(np.min(eval_s)),
                # each kernel's own start/stop events: the figures rocprofv3 reports.  0 = a one-launch step (the waves of the
                # evaluation kernel prepare their own slabs; tables of <= 4e5 detections in run order, option "fuse_prep")
                "slab_prep_ms": prep_ms,
                "one_launch_step": bool(prep_ms == 0.0),
                "step_kernel_sum_ms": eval_ms + prep_ms,    # <= ms_per_step: GPU time per step of the region (the rest: first dispatch + final synchronisation)
                "launches_timed": int(eval_s.shape[0]),
                "timing": ("kernel_ms: two HIP events on the launch stream around the K timed steps, per step" + (" minus slab_prep_ms" if prep_ms > 0 else "")
                           + " (launch-to-launch gaps included: an upper bound)" if in_region else "kernel_ms = kernel_ms_isolated (the step of this mode contains copies / collectives)")
                          + "; kernel_ms_isolated*, slab_prep_ms: start/stop events of hipExtLaunchKernelGGL, every kernel its own pair, on extra launches after the region",
                "units_per_launch": N,
                "algorithmic_bytes_per_detection": bpd,
                "frac_of_measured_copy_peak_6290": achieved / 6290.0,
                # written bytes only, against a bare non-temporal fill of the same chunk shape on the same part
                # (pcs_membench kind 4: 5.5-5.6 TB/s, profiles/r01/sweeps.md) — the write stream is the bound
                "written_GBps": N * (bpd - in_bytes) / (eval_ms * 1e-3) / 1e9,
                "frac_of_measured_nt_fill_5600": N * (bpd - in_bytes) / (eval_ms * 1e-3) / 1e9 / 5600.0,
                # the same fill measured in THIS run on THIS box right before the timed region (50 back-to-back launches of the
                # output's size): the ceiling the write stream of the step is held against
                "nt_fill_GBps_this_run": nt_fill_GBps,
                "frac_of_nt_fill_this_run": (N * (bpd - in_bytes) / (eval_ms * 1e-3) / 1e9 / nt_fill_GBps) if nt_fill_GBps else None,
            },
        }
        if args.stream_to_host:
            jb = d_j.numel() * d_j.element_size()
            line["host_stream"] = {"bytes_per_step_per_gpu": jb, "d2h_GBps_per_gpu": jb * args.steps / elapsed / 1e9,
                                   "note": "PCIe-bound: the Jacobian of every step is copied to page-locked host memory "
                                           "(double buffered, overlapped with the next kernel)"}
        if multi:
            line["multi_gpu"] = multi
        if normal_info:
            line["normal_equations"] = normal_info
        if args.host_path:
            eng.eval(ps)  # allocate scratch, warm
            reps, h0 = 3, time.perf_counter()
            for _ in range(reps):
                eng.eval(ps)
            hs = (time.perf_counter() - h0) / reps
            line["host_boundary"] = {"ms_per_call": hs * 1e3, "rows_per_s": 2.0 * N / hs,
                                     "d2h_GBps": (2 * N * (P + 1)) * 8 / hs / 1e9,
                                     "note": "Engine.eval: pageable NumPy outputs, PCIe D2H of the dense Jacobian dominates"}
        if world == 1 and args.scale == 1.0 and args.emulate_world <= 1 and (args.lm_compare or (args.config in (2, 3) and not args.no_normal_probe)):
            try:  # a probe: its failure must not cost the bench line
                cfgs = (2, 3) if args.lm_compare else (args.config,)
                line["lm_end_to_end"] = {f"config_{c}": lm_end_to_end(c, with_scipy=args.lm_compare, scipy_nfev=8 if c == 2 else 4) for c in cfgs}
            except Exception as exc:  # noqa: BLE001
                line["lm_end_to_end"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(rig, chain, ps, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
