"""HIP-graph replay of a train step: the host leaves the launch loop.

The reference dispatches every TF op of ``train_step`` eagerly from Python (model.py:168 -- the ``@tf.function`` is commented
out); this build's eager path does the same with ~1 250 ctypes launches per cycle step (32 ms of host time for a 40 ms step).
``StepProgram`` records one execution of the step body into HIP graphs (``torch.cuda.CUDAGraph`` = hipGraph on ROCm: stream
capture of the launches libsggan.so makes on the capturing stream) and afterwards replays them: a step then costs the host a
handful of ``hipGraphLaunch`` calls.

What cannot live inside a captured graph stays on the host BETWEEN graph segments, in program order: the RCCL gradient
all-reduces and their waits (collectives are launched eagerly so that nothing depends on RCCL's capture support; the
all-reduce of one network still overlaps the next segment's kernels because it runs on RCCL's own stream).  Every buffer a
segment touches is allocated from one private pool during capture and keeps its address; step inputs are copied into static
buffers before a replay; Adam reads its step number from a device counter (``sgg_adam_iter``).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi as A


class StepProgram:
    def __init__(self, device):
        self.device = torch.device(device)
        self.items = []            # ("graph", torch.cuda.CUDAGraph) | ("host", callable)
        self.capturing = False
        self._pool = None
        self._g = None
        self.n_graphs = 0
        self.keep = []             # tensors allocated OUTSIDE the capture pool whose addresses the recorded launches carry
                                   # (the shared scratch workspace, per-network scratch vectors): alive as long as the program

    # ---- recording -------------------------------------------------------------------------------------------------
    def _open(self):
        self._g = torch.cuda.CUDAGraph()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()     # one private pool shared by every segment of the step
        # thread_local: only THIS thread's unsafe calls invalidate the capture -- RCCL's watchdog thread polls its events
        # with HIP calls of its own while a process group is alive, and must not abort a recording
        self._g.capture_begin(pool=self._pool, capture_error_mode="thread_local")

    def _close(self):
        self._g.capture_end()
        self.items.append(("graph", self._g))
        self.n_graphs += 1
        self._g = None

    def _segment_is_empty(self):
        """No launch has been captured into the open segment yet (hipStreamGetCaptureInfo through the C ABI)."""
        n = C.c_int(0)
        A.check(A.lib().sgg_stream_capture_nodes(C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), C.byref(n)),
                "stream_capture_nodes")
        return n.value == 0

    def record(self, body):
        """Capture ``body()`` (kernel launches on the current stream + ``host()`` cut points).  Kernels are NOT executed."""
        assert not self.items, "a StepProgram records once"
        # Cyclic garbage is collected NOW and the collector stays off while the stream is capturing: an earlier model's recorded
        # program (hipGraphExec objects, events, its private pool) is cyclic garbage, and a collection that happens to run inside
        # the capture destroys it with HIP calls this thread may not make while capturing -- the process aborts in a destructor
        # (seen once in the test suite: "Fatal Python error: Aborted ... Garbage-collecting" under test_gpu_dp.py).
        # torch.cuda.graph() collects before its capture for the same reason.
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            torch.cuda.synchronize(self.device)
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                self.capturing = True
                try:
                    self._open()
                    out = body()
                    self._close()
                except BaseException:
                    self.capturing = False
                    if self._g is not None:
                        try:
                            self._g.capture_end()
                        except Exception:
                            pass
                        self._g = None
                    self.items = []
                    raise
                self.capturing = False
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
        finally:
            if gc_was_on:
                gc.enable()
        return out

    def host(self, fn):
        """A host-side action in program order.  While recording: closes the current graph segment, stores ``fn`` and
        opens the next segment (``fn`` is not called); a host action that follows another with no launch in between joins its cut.
        Otherwise calls it."""
        if not self.capturing:
            return fn()
        if self._segment_is_empty():
            self.items.append(("host", fn))          # consecutive host actions share one cut: no empty graph in between
            return None
        self._close()
        self.items.append(("host", fn))
        self._open()
        return None

    # ---- replay ----------------------------------------------------------------------------------------------------
    def replay(self):
        for kind, x in self.items:
            if kind == "graph":
                x.replay()
            else:
                x()
