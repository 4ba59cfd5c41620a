"""Streaming the dense Jacobian of every step to page-locked host memory (BASELINE config 5: "block-sparse Jacobian
streamed to host for LM normal-equations").

The reference returns J as a host array from every call (abstract_function_blocks.py:561, :641-651); at 1e7 detections
that is 1.7 GB per step and PCIe-bound.  Here the copy of step i runs on a side stream while the kernel of step i + 1
computes into the other device buffer:

    compute stream   wait free[b] -> evaluate into dev[b] -> record done[b]
    copy stream      wait done[b] -> dev[b] -> host[k] (page-locked) -> record free[b], ready[k]

``free[b]`` guards the device buffer (a later step must not overwrite what a copy is still reading), ``ready[k]`` tells
the host consumer that host buffer ``k`` holds its step.  A HOST buffer is handed out by ``wait(k)`` and belongs to the
consumer until it gives it back with ``release(k)``: ``step()`` refuses to queue a copy into a buffer that is still out
(a Jacobian somebody is reading must never change under its reader — the guarantee ``Engine``'s page-locked ring gives).
A consumer that never calls ``wait`` (bench.py --stream-to-host measures the copy rate only) is never in the way.
bench.py --stream-to-host and tests/test_gpu_parity.py use this class.
"""
from __future__ import annotations


class JacobianHostStreamer:
    def __init__(self, engine, n_rows: int, *, device: int = 0, n_device_buffers: int = 2, n_host_buffers: int = 2, first_device_buffer=None):
        import torch

        self.torch, self.eng = torch, engine
        self.dev = torch.device("cuda", device)
        tdt = torch.float64 if engine.dtype == "f64" else torch.float32
        shape = (2 * n_rows, engine.P)
        self.dev_bufs = [first_device_buffer if (first_device_buffer is not None and i == 0) else torch.empty(shape, dtype=tdt, device=self.dev)
                         for i in range(n_device_buffers)]
        self.host = [torch.empty(shape, dtype=tdt, pin_memory=True) for _ in range(n_host_buffers)]
        self.copy_stream = torch.cuda.Stream(self.dev)
        self.done = [torch.cuda.Event() for _ in range(n_device_buffers)]    # kernel finished writing dev_bufs[b]
        self.free = [torch.cuda.Event() for _ in range(n_device_buffers)]    # copy out of dev_bufs[b] finished
        self.ready = [torch.cuda.Event() for _ in range(n_host_buffers)]     # host[k] holds its step
        for e in self.free:
            e.record(self.copy_stream)
        self.step_no = 0
        self.out = [False] * n_host_buffers     # host[k] was handed to the consumer by wait(k) and has not been released
        # torch's default stream has the NULL handle, which the C ABI can only name as hipStreamLegacy — two spellings the runtime
        # does not order against each other in both directions (BlockedNormalEquations.on_stream): a caller on the default
        # stream gets its evaluations on a stream of this object, ordered after / before the caller's work by events
        self.compute_stream = torch.cuda.Stream(self.dev)

    def step(self, d_param_str: int, d_resid: int | None) -> int:
        """Queue one evaluation at the device-resident parameter string and the copy of its Jacobian; returns the index of
        the host buffer that will hold it (``wait(k)``).  Nothing here blocks the host."""
        torch = self.torch
        b = self.step_no % len(self.dev_bufs)
        k = self.step_no % len(self.host)
        if self.out[k]:
            raise RuntimeError(f"host buffer {k} is still with its consumer: call release({k}) before the step that reuses it "
                               f"(or build the streamer with more than {len(self.host)} host buffers)")
        self.step_no += 1
        cur = torch.cuda.current_stream(self.dev)
        compute = cur
        if cur.cuda_stream == 0:
            compute = self.compute_stream
            compute.wait_stream(cur)
        compute.wait_event(self.free[b])
        self.eng.eval_device_resident(d_param_str, d_resid, self.dev_bufs[b].data_ptr(), compute.cuda_stream)
        self.done[b].record(compute)
        if compute is not cur:
            cur.wait_stream(compute)      # later work of the caller (the next parameter update) follows the evaluation
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.done[b])
            self.host[k].copy_(self.dev_bufs[b], non_blocking=True)
            self.free[b].record(self.copy_stream)
            self.ready[k].record(self.copy_stream)
        return k

    def wait(self, k: int):
        """Block until host buffer ``k`` holds the Jacobian of the step that was routed to it; returns the tensor, which stays
        untouched until ``release(k)``."""
        self.ready[k].synchronize()
        self.out[k] = True
        return self.host[k]

    def release(self, k: int):
        """The consumer is done with host buffer ``k``: a later step may copy into it again."""
        self.out[k] = False
