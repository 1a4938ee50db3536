"""Host I/O around the hot path: files <-> HBM through pinned staging buffers.

The reference reads the FASTQ line by line (uq.py:371-425) and numpy.save()s each table (uq.py:263-274);
here a file moves in 16 MiB chunks: worker threads `readinto` / `pwrite` pinned buffers (the GIL is
released inside those calls) while the PCIe copy of the neighbouring chunk is in flight on the
context's stream.  Nothing is computed here -- bytes only.
"""
import fcntl
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

CHUNK = int(os.environ.get('UQ_IO_CHUNK_MB', '16')) << 20       # 16 buffers of 16 MiB: 0.53 s per 10 M reads end to end against 0.60 with 8 x 32 MiB
NBUF = int(os.environ.get('UQ_IO_NBUF', '16'))
WRITERS = int(os.environ.get('UQ_IO_WRITERS', '1'))      # pwrite()s in flight at a time (0: as many as buffers).  One: page allocation in tmpfs / the page cache does
                                                         # not scale with writers -- 1.57 GB of members took 0.44 s with eight, 0.30 s with one (the copies from HBM still run ahead)


class Staging:
    """NBUF pinned buffers of CHUNK bytes, allocated on first use and kept for the session."""

    def __init__(self, ctx, chunk=CHUNK, nbuf=NBUF):
        self.ctx, self.chunk, self.nbuf = ctx, chunk, nbuf
        self._pin = None
        self._pool = None

    def _buffers(self):
        if self._pin is None:
            t = self.ctx.torch
            self._pin = [t.empty(self.chunk, dtype=t.uint8, pin_memory=True) for _ in range(self.nbuf)]
            self._np = [p.numpy() for p in self._pin]
            self._ev = [t.cuda.Event() for _ in range(self.nbuf)]
            self._pool = ThreadPoolExecutor(max_workers=self.nbuf)
        return self._pin, self._np, self._ev

    # ------------------------------------------------------------------ file -> HBM
    def file_to_device(self, path, offset=0, size=None, out=None, on_chunk=None):
        """Bytes [offset, offset + size) of `path` as a uint8 device tensor (or into `out`).  on_chunk(d, lo, n): called right
        after the copy of bytes [lo, lo + n) has been queued on the stream -- work it queues there runs on that chunk while
        the next ones are still being read and copied (row f2: the newline census of the encoder)."""
        ctx = self.ctx
        if size is None:
            size = os.path.getsize(path) - offset
        d = out if out is not None else ctx.empty(size)
        if size == 0:
            return d
        pin, pnp, ev = self._buffers()
        fd = os.open(path, os.O_RDONLY)
        try:
            nchunks = (size + self.chunk - 1) // self.chunk

            def read(j):
                lo = j * self.chunk
                n = min(self.chunk, size - lo)
                k = j % self.nbuf
                mv = memoryview(pnp[k])[:n]
                got = 0
                while got < n:
                    r = os.preadv(fd, [mv[got:]], offset + lo + got)
                    if r <= 0: raise IOError('short read from %s' % path)
                    got += r
                return n

            futs = {}
            for j in range(min(self.nbuf, nchunks)):
                futs[j] = self._pool.submit(read, j)
            for j in range(nchunks):
                n = futs.pop(j).result()
                k = j % self.nbuf
                lo = j * self.chunk
                d[lo:lo + n].copy_(pin[k][:n], non_blocking=True)
                ev[k].record()
                if on_chunk is not None: on_chunk(d, lo, n)
                nxt = j + self.nbuf
                if nxt < nchunks:
                    # the buffer is free for the next read once its copy has left the host
                    def chained(jj=nxt, e=ev[k]):
                        e.synchronize()
                        return read(jj)
                    futs[nxt] = self._pool.submit(chained)
            for e in ev: e.synchronize()
        finally:
            os.close(fd)
        return d

    # ------------------------------------------------------------------ HBM -> file
    def device_to_fd(self, tensor, fd, offset):
        """Writes the bytes of a device tensor to `fd` at `offset` (pwrite).  Returns the byte count."""
        t = self.ctx.torch
        src = tensor.contiguous().view(t.uint8).reshape(-1)
        size = src.numel()
        if size == 0:
            return 0
        pin, pnp, ev = self._buffers()
        nchunks = (size + self.chunk - 1) // self.chunk
        writes = [None] * self.nbuf

        gate = threading.BoundedSemaphore(WRITERS) if WRITERS > 0 else None

        def write(k, n, pos):
            mv = memoryview(pnp[k])[:n]
            done = 0
            if gate is not None: gate.acquire()
            try:
                while done < n:
                    done += os.pwrite(fd, mv[done:], pos + done)
            finally:
                if gate is not None: gate.release()

        for j in range(nchunks):
            k = j % self.nbuf
            if writes[k] is not None:
                writes[k].result()                      # buffer k is on disk: reuse it
            lo = j * self.chunk
            n = min(self.chunk, size - lo)
            pin[k][:n].copy_(src[lo:lo + n], non_blocking=True)
            ev[k].record()

            def job(k=k, n=n, pos=offset + lo, e=ev[k]):
                e.synchronize()
                write(k, n, pos)
            writes[k] = self._pool.submit(job)
        for w in writes:
            if w is not None: w.result()
        return size

    def device_to_stream(self, tensor, fileobj):
        """Same, for file objects without a descriptor (BytesIO, pipes): sequential write()."""
        try:
            fd = fileobj.fileno()
            seekable = fileobj.seekable()
            # pwrite() on an O_APPEND descriptor ignores the offset (Linux): `--decode >> out.fastq` would get its
            # 32 MiB chunks in completion order.  Appending streams take the sequential path.
            if fcntl.fcntl(fd, fcntl.F_GETFL) & os.O_APPEND: seekable = False
        except Exception:
            fd, seekable = None, False
        if fd is not None and seekable:
            fileobj.flush()
            pos = fileobj.tell()
            n = self.device_to_fd(tensor, fd, pos)
            fileobj.seek(pos + n)
            return n
        t = self.ctx.torch
        src = tensor.contiguous().view(t.uint8).reshape(-1)
        size = src.numel()
        pin, pnp, ev = self._buffers() if size else (None, None, None)
        issued = {}
        nchunks = (size + self.chunk - 1) // self.chunk
        for j in range(nchunks + 1):
            if j < nchunks:
                k = j % 2
                lo = j * self.chunk
                n = min(self.chunk, size - lo)
                pin[k][:n].copy_(src[lo:lo + n], non_blocking=True)
                ev[k].record()
                issued[j] = (k, n)
            if j >= 1:
                k, n = issued.pop(j - 1)
                ev[k].synchronize()
                fileobj.write(memoryview(pnp[k])[:n])
        return size

    def to_numpy(self, tensor, dtype=None):
        """Device tensor -> fresh numpy array, through the pinned buffers (large arrays only pay off)."""
        t = self.ctx.torch
        src = tensor.contiguous().view(t.uint8).reshape(-1)
        size = src.numel()
        out = np.empty(size, dtype=np.uint8)
        if size < (4 << 20):
            out[:] = src.cpu().numpy()
        else:
            pin, pnp, ev = self._buffers()
            nchunks = (size + self.chunk - 1) // self.chunk
            for j in range(nchunks + 1):
                if j < nchunks:
                    k = j % 2
                    lo = j * self.chunk
                    n = min(self.chunk, size - lo)
                    pin[k][:n].copy_(src[lo:lo + n], non_blocking=True)
                    ev[k].record()
                if j >= 1:
                    kk = (j - 1) % 2
                    lo = (j - 1) * self.chunk
                    n = min(self.chunk, size - lo)
                    ev[kk].synchronize()
                    out[lo:lo + n] = pnp[kk][:n]
        return out.view(dtype) if dtype is not None else out
