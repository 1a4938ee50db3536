"""Sharded encode: one process per GPU, every rank encodes a contiguous range of reads and all ranks write
one `.uQ` file together (BASELINE configs[3] is this with `--sort QUAL --raw DNA QUAL QNAME`).

    python -m torch.distributed.run --nproc-per-node N -m uq_amd.dist_encode -i reads.fastq [uq flags]
    python -m torch.distributed.run --nproc-per-node N -m uq_amd.dist_encode --decode -i reads.uQ -o reads.fastq

The result is byte-for-byte the file the single-GPU CLI writes (members and config; tar mtimes aside):
  load      each rank streams its byte range of the file (+ slack) to HBM; an all-gather of the number of line
            starts per range tells every rank where its first record begins
  pass 1    local `uq_stats`, all-reduced (one collective) -> identical decisions everywhere
  QNAME     the device QNAME passes over shards (uq_amd.qname_device with a `Shard`)
  pass 3    local pack
  tables    `--sort`: sample sort over the ranks (dist.global_sort_rows: all-to-all(v) of rows by key range; equal
            rows share a rank unless their value is heavier than a rank's share -- then it is dealt over several
            ranks by file position and every consumer that counts groups stitches them at the rank boundaries:
            `_unique` here, `qname_device._distinct_counts`); the other tables follow with dist.dist_gather_rows;
            unique + key: head flags of the sorted shard, exclusive offset of the group counts, keys returned to
            file order with dist.dist_scatter_rows
  write     every member's global size is known after an all-gather of the shard sizes; ranks `pwrite` their
            pieces into place (pinned staging), rank 0 writes the tar / .npy headers and config.json

Only the exchange steps use the process group (RCCL over xGMI: rows move once, by key range; everything else is
a few integers).  With the gloo backend device tensors are staged through the host, which is how the CPU box
rehearses this path with several ranks sharing one GPU (tests/test_gpu_dist.py).
"""
import json
import os
import sys
import tarfile
import time

import numpy as np

from . import dist as uqdist
from .uq import Session, UqError, build_parser, error, npy_header, pattern_header, validate_args

# pattern id -> (column-major?, rows flipped?, columns flipped?)   (csrc/pattern.hip)
PATTERN_FLAGS = {'0.1': (0, 0, 0), '0.2': (1, 0, 0), '1.1': (1, 0, 1), '1.2': (0, 0, 1),
                 '2.1': (0, 1, 1), '2.2': (1, 1, 1), '3.1': (1, 1, 0), '3.2': (0, 1, 0)}
SLACK = 4 << 20          # bytes read past a rank's range so that its last record is complete


class ShardedSession(Session):
    """`Session` whose tables are shards; members are (header, total payload bytes, [(offset, device bytes)])."""

    def __init__(self, args, ctx=None, out=sys.stdout, group=None):
        super().__init__(args, ctx=ctx, out=out)
        self.be = uqdist.HipRows(self.ctx)
        self.group = group
        self.dist, self.rank, self.world = uqdist._world()
        if self.rank != 0: args.quiet = True

    # ------------------------------------------------------------------ errors are agreed on
    def agree(self, message=None):
        """Collective: every rank calls this after a step that can fail on ONE rank only (I/O, a record straddling the
        slack, an uncoded symbol); `message` = this rank's error text or None.  If any rank failed, EVERY rank raises the
        same UqError -- the text of the lowest failing rank -- instead of leaving its peers in the next collective."""
        sh = uqdist.Shard(self.be, 0, 0, self.group)
        flags = sh.gather_ints(0 if message is None else 1)
        if any(flags):
            msgs = sh.gather_bytes((message or '').encode('utf-8', 'replace'))
            error(msgs[flags.index(1)].decode('utf-8', 'replace'))

    def guarded(self, fn, what):
        """Run a rank-local step; whatever it raises is agreed on (see `agree`).  Returns fn()'s value."""
        res, msg = None, None
        try:
            res = fn()
        except UqError as e:
            msg = str(e)
        except Exception as e:                       # I/O, HIP / ABI errors (UqHipError is a RuntimeError), anything else rank-local: the
            msg = 'ERROR: rank %d failed while %s: %s: %s' % (self.rank, what, type(e).__name__, e)    # peers must not wait in the next collective
        self.agree(msg)
        return res

    # ------------------------------------------------------------------ load
    def load(self, path):
        ops, ctx, t = self.ops, self.ctx, self.ctx.torch
        size = os.path.getsize(path)
        if size == 0: error('ERROR: empty input')
        self.path, self._host = path, None
        lo, hi = size * self.rank // self.world, size * (self.rank + 1) // self.world
        a = max(lo - 1, 0)                         # a newline at lo - 1 makes `lo` a line start
        b = min(size, hi + SLACK)
        chunk = self.guarded(lambda: self.io.file_to_device(path, a, b - a), 'reading ' + path)
        # line starts inside [lo, hi): one per newline in [lo - 1, hi - 1), plus the start of the file
        span = max(0, (hi - 1) - a)
        mine = (ops.count_lines(ctx, chunk[:span]) if span else 0) + (1 if self.rank == 0 else 0)
        sh = uqdist.Shard(self.be, 0, 0, self.group)
        per_rank = sh.gather_ints(mine)
        # A file that does not end in '\n' has one line start more than it has lines (`wc -l` counts newlines, uq.py:85):
        # the unterminated tail is nobody's line -- the rank that owns its start drops it, as a GLOBAL decision (the last
        # rank sees the last byte; every rank applies the same correction), so that the count below is the single-GPU
        # CLI's and the reference's, and so is the message.
        ends_nl = sh.gather_ints(int(chunk[-1]) == 10 if (self.rank == self.world - 1 and chunk.numel()) else 0)[self.world - 1]
        if not ends_nl:
            owner = max(r for r in range(self.world) if per_rank[r] > 0)
            per_rank[owner] -= 1
            if owner == self.rank: mine -= 1
        total_lines = sum(per_rank)
        if total_lines % 4 != 0:
            error('ERROR: The FASTQ file provided contains' + str(total_lines) + 'rows, which is not divisible by 4!')
        if total_lines == 0: error('ERROR: empty input')
        first_line = sum(per_rank[:self.rank])                     # file-wide number of my first line start
        j0 = (-first_line) % 4                                     # my first record starts at my j0-th line start
        n = max(0, -(-(mine - j0) // 4))
        counts = sh.gather_ints(n)
        self.total = n
        self.total_reads = sum(counts)
        self.read_offset = sum(counts[:self.rank])
        self.shard_starts = [sum(counts[:r]) for r in range(self.world + 1)]
        self.shard = uqdist.Shard(self.be, self.read_offset, self.total_reads, self.group)
        nl = ops.count_lines(ctx, chunk)
        ls = ops.index_lines(ctx, chunk, nl)                       # ls[k] = chunk offset after the k-th newline, ls[0] = 0
        k0 = j0 + (0 if self.rank == 0 else 1)
        self.agree('ERROR: a record of more than %d bytes straddles a shard boundary (rank %d)' % (SLACK, self.rank)
                   if (n and k0 + 4 * n > nl) else None)
        if n:
            ends = ctx.to_numpy(ls[k0:k0 + 4 * n + 1:4 * n], np.uint64)
            start, end = int(ends[0]), int(ends[1])
            del ls
            # the shard's own records through the single-GPU step (Session.load_device: census -> [rank 0: QNAME layout guess] -> ONE broadcast of the guess
            # -> pack + statistics + QNAME fields in one kernel, no record index; `d_ls` is expanded when a fallback asks for it)
            self.load_device(chunk[start:end])
            if self.total != n: error('ERROR: rank %d counted %d reads in a shard of %d' % (self.rank, self.total, n))
        else:
            self.d_buf, self.d_ls = chunk[:0], t.zeros(1, dtype=t.int64, device=ctx.device)
            self._spec, self._fq, self._guess_shared = None, None, False
            self.d_stats = ops.stats_new(ctx)
        if self.fused_qname_enabled() and not self._guess_shared:
            # a rank without reads, or whose head gave no guess of its own, still takes part in the broadcast (its analyse_qname says "not usable")
            self.share_qname_guess(ops.FusedQname(ctx, 16), dummy=True)

    # the fused QNAME pass over shards: rank 0's guess is every rank's (qname_device.broadcast_guess), exactly once per load
    def owns_qname_guess(self):
        return self.rank == 0

    def share_qname_guess(self, fq, dummy=False):
        from . import qname_device
        if dummy and self.rank == 0: fq.q.zero_()                 # rank 0 itself has no guess: ok = 0, every rank stands down
        qname_device.broadcast_guess(self.ctx, fq, self.shard)
        self._guess_shared = True

    # ------------------------------------------------------------------ analysis seams
    def fetch_stats(self):
        hs = uqdist.allreduce_stats(self.ctx, self.d_stats, self.read_offset)
        if hs.incomplete:                        # the speculative pass of SOME rank could not count everything (the flag is summed): all redo the plain pass
            self._spec = None
            self.d_stats = self.ops.stats_new(self.ctx)
            if self.total: self.ops.stats_accumulate(self.ctx, self.d_stats, self.d_buf, self.d_ls, 0, self.total)
            hs = uqdist.allreduce_stats(self.ctx, self.d_stats, self.read_offset)
        return hs

    def starts_with_at(self):
        flag = 1 if (self.rank != 0 or (self.total and int(self.d_buf[0]) == ord('@'))) else 0
        return min(self.shard.reduce([flag], 'min')) == 1

    def first_seen(self):
        fs = self.ops.first_occurrence(self.ctx, self.d_buf, self.d_ls, 0, self.total, index_base=self.read_offset) if self.total \
            else np.full(256, np.iinfo(np.uint64).max, dtype=np.uint64)
        big = (1 << 63) - 1
        red = self.shard.reduce([min(int(v), big) for v in fs], 'min')
        return np.array([np.iinfo(np.uint64).max if v == big else v for v in red], dtype=np.uint64)

    def reads_in_file(self):
        return self.total_reads

    def analyse_qname(self):
        from . import qname, qname_device
        res = None
        if self.fused_qname_enabled():           # collective: every rank enters, a rank whose own pass did not hold says so inside
            self.qname_path = 'fused'
            fq, self._fq = getattr(self, '_fq', None), None
            res = qname_device.analyse_fused_sharded(self.ctx, fq, self.total, self.shard, usable=fq is not None or self.total == 0)
        if res is None:
            self.qname_path = 'device'
            res = qname_device.analyse_device(self.ctx, self.d_buf, self.d_ls, self.total, self.shard)
        if res is None:
            raise qname.QnameError('ERROR: these QNAMEs need the sequential host passes (see DESIGN.md 2), which the sharded '
                                   'encoder does not run; encode this file on one GPU')
        return res

    # ------------------------------------------------------------------ members as pieces
    def _put(self, name, header, total_bytes, pieces):
        self.members[name] = (header, int(total_bytes), [(int(o), p) for o, p in pieces if p.numel()])

    def write_pattern_shard(self, table, first_row, rows_total, filename):
        """write_pattern (uq.py:257-270) for rows [first_row, first_row + n) of a `rows_total`-row table."""
        args = self.args
        if args.pattern is None: pattern = '0.1'; args.pattern = ['0.1', '0.1']
        elif filename.startswith('DNA'): pattern = args.pattern[0]
        elif filename.startswith('QUAL'): pattern = args.pattern[1]
        else: error('ERROR: This should never happen!')
        t, n, cols = table
        colmajor, fr, _ = PATTERN_FLAGS[pattern]
        header = pattern_header(rows_total, cols, pattern)
        if n == 0:
            return self._put(filename, header, rows_total * cols, [])
        payload = t[:n * cols] if pattern == '0.1' else self.ops.pattern(self.ctx, t, n, cols, pattern)   # the shard's block, same flips applied locally
        at = (rows_total - first_row - n) if fr else first_row             # where the block's rows sit in the (flipped) row order
        if not colmajor:
            pieces = [(at * cols, payload)]
        else:                                                              # column k of the payload = rows_total bytes; mine are n of them
            pieces = [(k * rows_total + at, payload[k * n:(k + 1) * n]) for k in range(cols)]
        self._put(filename, header, rows_total * cols, pieces)

    def write_out_shard(self, tensor, first, total, filename, dtype):
        """write_out (uq.py:272-274) for entries [first, first + len) of a 1-D member of `total` entries."""
        isz = np.dtype(dtype).itemsize
        self._put(filename, npy_header((total,), False, dtype), total * isz,
                  [(first * isz, tensor.contiguous().view(self.ctx.torch.uint8).reshape(-1))])

    # ------------------------------------------------------------------ table builds over shards
    def _sorted(self, t, n, cols, want='sorted'):
        gs = uqdist.global_sort_rows(self.be, t, n, cols, self.read_offset, self.group, total_rows=self.total_reads,
                                      rows_of_ranks=[self.shard_starts[r + 1] - self.shard_starts[r] for r in range(self.world)], want=want)
        return gs, {'gidx': gs['gidx'], 'offset': gs['offset'], 'rows': gs['rows']}

    def _unique(self, gs, cols):
        """A rank's range of the global sort (want='unique') -> (unique rows, count, first global group id, total groups, group id per sorted row).
        A group may run on from the rank in front (dist.global_sort_rows deals a tie group heavier than a rank's share over several
        ranks): the ranks exchange their first and last rows, and a rank whose first row equals the last row in front of it drops
        that row from its unique table and starts its ids one lower."""
        ops, ctx, t = self.ops, self.ctx, self.ctx.torch
        m = gs['rows']
        if m:
            skey, nu, uniq = gs['group'], gs['ngroups'], gs['unique']                # the sort's own head flags; only the distinct rows were moved
            edge = bytes(ctx.to_numpy(gs['edge']).tobytes())
        else:
            skey, uniq, nu, edge = t.empty(0, dtype=t.int32, device=ctx.device), ctx.empty(0), 0, b''
        edges = self.shard.gather_bytes(edge)                       # first + last row of every rank (empty: no rows)
        per_rank = self.shard.gather_ints(nu)
        cont, last = [], None                                       # cont[r]: rank r's first group is the one the rows in front ended with
        for r in range(self.world):
            cont.append(1 if (edges[r] and last is not None and edges[r][:cols] == last) else 0)
            if edges[r]: last = edges[r][cols:]
        eff = [per_rank[r] - cont[r] for r in range(self.world)]
        g0, nu_total = sum(eff[:self.rank]), sum(eff)
        mine = cont[self.rank]
        shift = g0 - mine                                           # (group ids are u32 bit patterns in an int32 tensor)
        return uniq[mine * cols:], nu - mine, g0, nu_total, self.be.index_affine(skey, shift, 4) if m else skey

    def _key_member(self, gs, order, gids, sort_order, isz, name):
        """The key member: group ids in sorted order (this table is sorted on), in file order, or in another table's order."""
        ops, ctx, t = self.ops, self.ctx, self.ctx.torch
        N = self.total_reads
        if sort_order is False:
            k, first = gids, order['offset']
        else:
            in_file_order = uqdist.dist_scatter_rows(self.be, gids.view(t.uint8), 4, self.shard_starts, gs['gidx'], self.group).view(t.int32)
            if sort_order is None:
                k, first = in_file_order, self.read_offset
            else:
                k = uqdist.dist_gather_rows(self.be, in_file_order.view(t.uint8), self.total, 4, self.shard_starts, sort_order['gidx'], self.group).view(t.int32)
                first = sort_order['offset']
        self.write_out_shard(ops.narrow(ctx, k, isz), first, N, name, self._npdtype(isz))

    def encode_dna_qual(self, sort_order, table_name, raw, test):
        """uq.py:765-805 over shards.  sort_order: None = no sort, False = compute and return, dict = apply."""
        if test: error('ERROR: --test is not available in the sharded encoder')
        ops = self.ops
        t, n, cols = self.tables[table_name]
        N = self.total_reads
        if raw:
            if sort_order is None:
                table, first = (t, n, cols), self.read_offset
            elif sort_order is False:
                gs, sort_order = self._sorted(t, n, cols)                                        # uq.py:773-777
                table, first = (gs['table'], gs['rows'], cols), gs['offset']
            else:
                g = uqdist.dist_gather_rows(self.be, t, n, cols, self.shard_starts, sort_order['gidx'], self.group)
                table, first = (g, sort_order['rows'], cols), sort_order['offset']
            self.write_pattern_shard(table, first, N, table_name + '.raw')
        else:
            gs, order = self._sorted(t, n, cols, want='unique')                                  # uq.py:784-789
            uniq, nu, g0, nu_total, gids = self._unique(gs, cols)
            isz = ops.key_itemsize(nu_total - 1)                                                 # uq.py:790
            self._key_member(gs, order, gids, sort_order, isz, table_name + '.key')
            if sort_order is False: sort_order = order                                           # stable order == argsort(key), uq.py:796
            self.write_pattern_shard((uniq, nu, cols), g0, nu_total, table_name)
        return sort_order

    def encode_qname(self, sort_order, raw, test):
        """uq.py:808-851 over shards."""
        if test: error('ERROR: --test is not available in the sharded encoder')
        ops, ctx, columns, t = self.ops, self.ctx, self.columns, self.ctx.torch
        cols_d = self.tables['QNAME']
        n, N = self.total, self.total_reads
        common = max(c.element_size() for c in cols_d)
        ncols = len(cols_d)
        width = ncols * common
        if raw:
            if sort_order is False:
                rows = ops.stack_columns(ctx, cols_d, common)                                    # uq.py:814-816
                gs, sort_order = self._sorted(rows, n, width)
                for idx, column in enumerate(columns):
                    isz = np.dtype(column['dtype']).itemsize
                    c = ops.unstack_column(ctx, gs['table'], gs['rows'], ncols, common, idx, isz)
                    self.write_out_shard(c, gs['offset'], N, column['name'] + '.raw', np.dtype(column['dtype']))
                return sort_order
            for idx, column in enumerate(columns):
                c, first = cols_d[idx], self.read_offset
                if sort_order is not None:
                    c = uqdist.dist_gather_rows(self.be, c.view(t.uint8), n, c.element_size(), self.shard_starts, sort_order['gidx'],
                                                self.group).view(c.dtype)
                    first = sort_order['offset']
                self.write_out_shard(c, first, N, column['name'] + '.raw', np.dtype(column['dtype']))
        else:
            rows = ops.stack_columns(ctx, cols_d, common)                                        # uq.py:828-830
            gs, order = self._sorted(rows, n, width, want='unique')
            uniq, nu, g0, nu_total, gids = self._unique(gs, width)
            isz = ops.key_itemsize(nu_total - 1)                                                 # uq.py:832
            self._key_member(gs, order, gids, sort_order, isz, 'QNAME.key')
            if sort_order is False: sort_order = order
            for idx, column in enumerate(columns):                                               # uq.py:845-847
                col = ops.unstack_column(ctx, uniq, nu, ncols, common, idx, np.dtype(column['dtype']).itemsize)
                self.write_out_shard(col, g0, nu_total, column['name'], np.dtype(column['dtype']))
        return sort_order

    def _encode(self, variable):
        if self.total == 0:                                        # a rank without reads still owns (empty) tables
            e = self.ctx.empty(0)
            self.agree(None)
            return ((e, 0, self.d['dna_bytes_per_row']), (e, 0, self.d['quality_bytes_per_row']))
        return self.guarded(lambda: Session._encode(self, variable), 'packing')     # an uncoded symbol is one rank's finding

    # ------------------------------------------------------------------ container
    def write_container(self, path):
        """uq.py:897-913 with every rank writing its pieces: the tar layout follows from the member sizes alone."""
        args = self.args
        cfg = dict(self.config)
        cfg['sort'] = args.sort if isinstance(args.sort, str) else [None]
        cfg['raw'] = sorted(args.raw, key=str) if args.raw else [None]
        cfg['pattern'] = list(args.pattern) if args.pattern else None
        self.config = cfg
        blob = json.dumps(cfg, indent=4, sort_keys=True).encode()

        def order(name):
            base = name.split('.')[0]
            if base.startswith('QNAME_'): return (3, int(base[6:]), name)
            return ({'DNA': 0, 'QUAL': 1, 'QNAME': 2}.get(base, 4), 0, name)

        names = sorted(self.members, key=order)
        layout, pos = [], 0                         # (name, header bytes, tar header offset, payload offset, size)
        for name, header, size in [('config.json', blob, 0)] + [(k, self.members[k][0], self.members[k][1]) for k in names]:
            layout.append((name, header, pos, pos + tarfile.BLOCKSIZE + len(header), len(header) + size))
            pos += tarfile.BLOCKSIZE + len(header) + size
            pos += -pos % tarfile.BLOCKSIZE
        end = pos + 2 * tarfile.BLOCKSIZE
        end += -end % tarfile.RECORDSIZE
        tmp = path + '.part'
        mtime = self.shard.reduce([int(time.time())], 'min')[0]
        def create():
            if self.rank != 0: return None
            f = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            os.ftruncate(f, end)                                   # sparse zeros: tar padding and end blocks are already there
            for name, header, hpos, _, size in layout:
                ti = tarfile.TarInfo(name); ti.size = size; ti.mtime = mtime
                os.pwrite(f, ti.tobuf(tarfile.DEFAULT_FORMAT, tarfile.ENCODING, 'surrogateescape') + header, hpos)
            return f
        fd = self.guarded(create, 'creating ' + tmp)               # also the barrier: the file exists at its final size

        def put():
            f = fd if self.rank == 0 else os.open(tmp, os.O_WRONLY)
            try:
                for name, _, _, ppos, _ in layout[1:]:
                    for off, piece in self.members[name][2]:
                        self.io.device_to_fd(piece, f, ppos + off)
            finally:
                os.close(f)
        try:
            self.guarded(put, 'writing ' + tmp)                    # also the barrier: all pieces are in place
        except UqError:
            if self.rank == 0 and os.path.exists(tmp): os.remove(tmp)
            raise
        if self.rank == 0:
            os.replace(tmp, path)


    # ------------------------------------------------------------------ decode
    def decode_sharded(self, out_path):
        """uq.py:926-1058 over the ranks: rank r decodes reads [n r / W, n (r + 1) / W) of the stored order -- key slices
        and raw row slices come from the file, the tables the keys index are loaded by every rank -- and writes its
        text in place; the offsets are an all-gather of the text sizes.  The result is the single-GPU decoder's output."""
        def my_text():
            # everything here is rank-local (a damaged key entry or a row without sentinel shows up on the rank that owns
            # that slice only): `guarded` turns it into the same UqError on every rank
            members, config = self.open_container()
            if not self.device_text_possible(config):
                error('ERROR: this QNAME layout decodes on one GPU only (python -m uq_amd.uq --decode)')
            if 'DNA.raw' in members: n = self.member_rows(members, 'DNA.raw', (config['pattern'] or ['0.1', '0.1'])[0])
            else: n = self.member_rows(members, 'DNA.key')
            lo, hi = n * self.rank // self.world, n * (self.rank + 1) // self.world
            DNA, QUAL, d_cols = self.load_tables(members, config, rows=(lo, hi))
            if hi > lo: return self.decode_text(config, DNA, QUAL, d_cols), lo, n
            return self.ctx.torch.empty(0, dtype=self.ctx.torch.uint8, device=self.ctx.device), lo, n
        text, lo, n = self.guarded(my_text, 'decoding ' + str(self.args.input))
        shard = uqdist.Shard(self.be, lo, n, self.group)
        sizes = shard.gather_ints(int(text.numel()))
        offset, total = sum(sizes[:self.rank]), sum(sizes)
        tmp = out_path + '.part'

        def create():
            if self.rank != 0: return None
            f = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            os.ftruncate(f, total)
            return f
        fd = self.guarded(create, 'creating ' + tmp)               # also the barrier: the file exists at its final size

        def put():
            f = fd if self.rank == 0 else os.open(tmp, os.O_WRONLY)
            try:
                if text.numel(): self.io.device_to_fd(text, f, offset)
            finally:
                os.close(f)
        try:
            self.guarded(put, 'writing ' + tmp)                    # also the barrier: all pieces are in place
        except UqError:
            if self.rank == 0 and os.path.exists(tmp): os.remove(tmp)
            raise
        if self.rank == 0:
            os.replace(tmp, out_path)

    def member_rows(self, members, name, pattern='0.1'):
        """Rows of a 2-D member / elements of a 1-D one, from its .npy header."""
        import io as _io
        offset, size = members[name]
        with open(self.tar_path, 'rb') as fh:
            fh.seek(offset)
            f = _io.BytesIO(fh.read(min(size, 65536)))
        version = np.lib.format.read_magic(f)
        shape, _, _ = np.lib.format.read_array_header_1_0(f) if version == (1, 0) else np.lib.format.read_array_header_2_0(f)
        if len(shape) == 1: return int(shape[0])
        return int(shape[0] if int(pattern[0]) % 2 == 0 else shape[1])


def main(argv=None):
    import torch
    import torch.distributed as dist
    args = build_parser().parse_args(argv)
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('UQ_DIST_BACKEND', 'nccl')
    ngpu = torch.cuda.device_count()
    device = local_rank % max(ngpu, 1)                             # gloo rehearsal: ranks may share a card
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', device))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    code, session = 0, None
    t_ready = time.perf_counter()                                  # interpreter, torch and the process group are up
    try:
        args.device = device
        validate_args(args)
        session = ShardedSession(args)
        if args.decode:
            if not args.output: error('ERROR: the sharded decoder writes a file: give it -o reads.fastq')
            session.decode_sharded(args.output)
        else:
            session.encode()
    except UqError as e:
        if rank == 0: print(e)
        code = 1
    finally:
        if os.environ.get('UQ_TIMING') and rank == 0:
            print(json.dumps({'uq_timing': 'dist_encode', 'world': world, 'work_s': round(time.perf_counter() - t_ready, 3),
                              'load': getattr(session, 'load_path', None), 'qname': getattr(session, 'qname_path', None)}),
                  file=sys.stderr, flush=True)
        dist.destroy_process_group()
    return code


if __name__ == '__main__':
    sys.exit(main())
