"""Device context: one process, one MI355X, one HIP stream.

PyTorch is plumbing here -- it owns device memory (caching allocator) and the stream, and
`torch.distributed` (RCCL) carries the multi-GPU exchange.  All compute goes through the C ABI
(`uq_amd._lib`); tensors cross it as raw device pointers.
"""
import contextlib
import ctypes as C

import numpy as np

from . import _lib
from ._lib import call, UqHipError


class Context:
    """Binds a `uq_ctx` to a torch device and a dedicated torch stream (made current).

    All Contexts of a process on one device share that ONE stream: the kernels behind the C ABI and the torch operations of the
    caller (allocations, comparisons, copies) are ordered by being on the same stream, and a second Context that made a stream of
    its own current would silently take that ordering away from the first (its kernels would race with torch operations issued
    afterwards).  Work that is meant to run beside it takes a SideContext."""

    _streams = {}        # device index -> the torch stream every Context on that device uses

    def __init__(self, device=0):
        import torch
        self.torch = torch
        _lib.load()
        if not torch.cuda.is_available():
            raise UqHipError('no HIP device visible: the uQ hot path runs on MI355X only (no CPU fallback)')
        self.device = torch.device('cuda', device)
        torch.cuda.set_device(self.device)
        idx = self.device.index or 0
        if idx not in Context._streams:
            Context._streams[idx] = torch.cuda.Stream(device=self.device)
        self.stream = Context._streams[idx]
        torch.cuda.set_stream(self.stream)
        h = C.c_void_p()
        call('uq_ctx_create', int(device), C.c_void_p(self.stream.cuda_stream), C.byref(h))
        self.h = h
        import os
        if os.environ.get('UQ_MSD_MIN_ROWS'):              # hunts: every row sort's round 0 as the MSD partition (1) or never (-1); default 2^18 rows
            call('uq_sort_config', self.h, int(os.environ['UQ_MSD_MIN_ROWS']), None, 0)

    def close(self):
        if self.h:
            call('uq_ctx_destroy', self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory helpers (torch tensors as typed views of HBM)
    def empty(self, n, dtype=None):
        t = self.torch
        return t.empty(int(n), dtype=dtype or t.uint8, device=self.device)

    def zeros(self, n, dtype=None):
        t = self.torch
        return t.zeros(int(n), dtype=dtype or t.uint8, device=self.device)

    def to_device(self, array):
        """numpy array (any dtype) -> device tensor of bytes-compatible dtype."""
        t = self.torch
        a = np.ascontiguousarray(array)
        if not a.flags.writeable:
            a = a.copy()
        if a.dtype in (np.uint16, np.uint32, np.uint64):
            signed = {2: np.int16, 4: np.int32, 8: np.int64}[a.dtype.itemsize]
            return t.from_numpy(a.view(signed)).to(self.device)
        return t.from_numpy(a).to(self.device)

    def bytes_to_device(self, data):
        a = np.frombuffer(data, dtype=np.uint8)
        return self.torch.from_numpy(a.copy() if not a.flags.writeable else a).to(self.device)

    def to_numpy(self, tensor, dtype=None, shape=None):
        a = tensor.detach().cpu().numpy()
        if dtype is not None:
            a = a.view(dtype)
        if shape is not None:
            a = a.reshape(shape)
        return a

    def sync(self):
        call('uq_ctx_sync', self.h)

    def scope(self):
        """Context manager under which torch allocations belong to this context's stream (the main context: already current)."""
        return contextlib.nullcontext()

    def adopt(self, tensor):
        """Declare that `tensor` (allocated elsewhere) is about to be used on this context's stream (the main context: nothing to do)."""
        return tensor

    @staticmethod
    def ptr(tensor):
        return C.c_void_p(tensor.data_ptr()) if tensor is not None else C.c_void_p(0)


class SideContext:
    """A second `uq_ctx` on the same device with a private stream of its own: small independent work (the encoder's guess from the
    head of the file) runs and synchronises there while the main context's stream is busy with the census.

    Allocation discipline (torch's caching allocator hands a freed block back to the stream it was allocated on, without waiting
    for any other stream): everything a SideContext computes with is allocated under `scope()` -- i.e. with the side stream as
    torch's current stream, so the blocks belong to it -- and a tensor that comes from the main stream is handed over with
    `adopt()` (`Tensor.record_stream`: the allocator will not reuse its block before the side stream's pending work has finished).
    An adopted input must already be COMPLETE on the main stream (the side stream does not wait for the main one -- that is the
    point of it); `ops.head_guess` follows both rules."""

    def __init__(self, ctx):
        self.torch, self.device, self.main = ctx.torch, ctx.device, ctx
        self.stream = ctx.torch.cuda.Stream(device=ctx.device)
        h = C.c_void_p()
        call('uq_ctx_create', int(ctx.device.index or 0), C.c_void_p(self.stream.cuda_stream), C.byref(h))
        self.h = h

    def scope(self):
        return self.torch.cuda.stream(self.stream)

    def adopt(self, tensor):
        tensor.record_stream(self.stream)
        return tensor

    def to_numpy(self, tensor, dtype=None, shape=None):
        with self.scope():
            return self.main.to_numpy(tensor, dtype, shape)

    def sync(self):
        call('uq_ctx_sync', self.h)

    def close(self):
        if self.h:
            call('uq_ctx_destroy', self.h)
            self.h = None

    def __del__(self):
        try: self.close()
        except Exception: pass
