"""numpy / oracle stand-ins for the device layer under `Session.load_tables` / `decode_text`, so that the sharded
decoder's own logic (row / key slices, offsets from the all-gathered text sizes, writes in place) runs on the CPU
with several gloo ranks.  Test infrastructure only."""
import io
import os

import numpy as np
import torch

import uq_oracle as O


class FakeCtx:
    torch = torch
    device = 'cpu'

    @staticmethod
    def to_numpy(tensor, dtype=None, shape=None):
        a = tensor.numpy()
        return a.view(dtype) if dtype is not None else a


class FakeIO:
    def file_to_device(self, path, offset=0, size=None, out=None):
        with open(path, 'rb') as f:
            f.seek(offset)
            data = f.read() if size is None else f.read(size)
        return torch.frombuffer(bytearray(data), dtype=torch.uint8) if data else torch.empty(0, dtype=torch.uint8)

    def device_to_fd(self, tensor, fd, offset):
        os.pwrite(fd, tensor.numpy().tobytes(), offset)


def _npy(a):
    f = io.BytesIO(); np.save(f, a); return f.getvalue()


class FakeOps:
    @staticmethod
    def gather_rows(ctx, table, nrows, cols, index):
        t = table.numpy().reshape(nrows, cols)
        idx = index.numpy().astype(np.int64)
        if index.dtype != torch.uint8: idx &= (1 << (8 * index.element_size())) - 1      # the key's bytes are unsigned
        return torch.from_numpy(np.ascontiguousarray(t[idx]).reshape(-1))

    @staticmethod
    def check_index_range(ctx, index, limit):
        idx = index.numpy().astype(np.int64)
        if index.dtype != torch.uint8: idx &= (1 << (8 * index.element_size())) - 1
        bad = np.flatnonzero(idx >= limit)
        return int(bad[0]) if len(bad) else None

    @staticmethod
    def unpattern(ctx, payload, rows, cols, pattern):
        k = int(pattern[0])
        shape = (rows, cols) if k % 2 == 0 else (cols, rows)
        a = payload.numpy().reshape(shape, order='F' if pattern.endswith('.2') else 'C')
        return torch.from_numpy(np.ascontiguousarray(np.rot90(a, -k)).reshape(-1))

    @staticmethod
    def decode_fastq(ctx, config, column_tensors, dna, qual, nreads):
        members = {'DNA.raw': _npy(dna.numpy().reshape(nreads, -1)), 'QUAL.raw': _npy(qual.numpy().reshape(nreads, -1))}
        for i, (c, cc) in enumerate(zip(column_tensors, config['QNAME_columns'])):
            members['QNAME_%d.raw' % (i + 1)] = _npy(np.frombuffer(c.numpy().tobytes(), dtype=np.dtype(cc['dtype'])))
        text = O.decode(dict(config, pattern=['0.1', '0.1']), members).encode('latin-1')
        return torch.frombuffer(bytearray(text), dtype=torch.uint8), None


class FakeLoadOps:
    """numpy stand-ins for what `ShardedSession.load` calls (newline census, record index, statistics)."""

    @staticmethod
    def count_lines(ctx, buf):
        return int((buf.numpy() == 10).sum())

    @staticmethod
    def index_lines(ctx, buf, nlines):
        pos = np.flatnonzero(buf.numpy() == 10)[:nlines] + 1
        return torch.from_numpy(np.concatenate([[0], pos]).astype(np.int64))

    @staticmethod
    def stats_new(ctx):
        return torch.zeros(8, dtype=torch.uint8)

    @staticmethod
    def stats_accumulate(ctx, st, buf, ls, first, n):
        pass
