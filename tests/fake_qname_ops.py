"""numpy stand-ins for the device entry points uq_amd.qname_device calls (uq_qname_layout, uq_qname_tokenise,
uq_unique_rows, uq_prefix_distinct, uq_narrow, uq_encode_int), following the contracts in include/uqhip.h.
TEST INFRASTRUCTURE ONLY: lets the CPU suite exercise the host-side decision logic of qname_device (the
closed form of the reference's sequential QNAME loop) against the oracle without a GPU.  The kernels
themselves are checked on the GPU by tests/test_gpu_qname.py."""
import types

import numpy as np
import torch

NONE = (1 << 64) - 1


class FakeCtx:
    torch = torch
    device = torch.device('cpu')

    @staticmethod
    def to_numpy(t, dtype=None, shape=None):
        a = t.detach().cpu().numpy()
        if dtype is not None: a = a.view(dtype)
        return a

    @staticmethod
    def empty(n, dtype=None):
        return torch.empty(int(n), dtype=dtype or torch.uint8)


def gather_rows(ctx, table, table_rows, cols, index, n_out=None, out=None):
    t = table.numpy().view(np.uint8).reshape(table_rows, cols)
    idx = index.numpy().astype(np.int64) & 0xFFFFFFFF if index.dtype == torch.int32 else index.numpy().astype(np.int64)
    return torch.from_numpy(np.ascontiguousarray(t[idx]).reshape(-1))


def lower_bound_rows(ctx, sorted_table, rows, cols, probes, nprobes):
    import bisect
    keys = [bytes(r) for r in sorted_table.numpy().view(np.uint8).reshape(rows, cols)]
    return torch.tensor([bisect.bisect_left(keys, bytes(p)) for p in probes.numpy().view(np.uint8).reshape(nprobes, cols)], dtype=torch.int64)


def _names(buf, ls, n):
    b = buf.numpy().tobytes()
    o = ls.numpy().view(np.uint64)
    return [b[int(o[4 * i]):int(o[4 * i + 1]) - 1] for i in range(n)]


def qname_layout(ctx, buf, ls, n, line1, read_index_base=0):
    names = _names(buf, ls, n)
    chars = []
    for c in line1:
        if c not in chars: chars.append(c)
    if len(chars) > 64:
        from uq_amd._lib import UqHipError
        raise UqHipError('more than 64 distinct characters')
    r = types.SimpleNamespace(min_lcp=len(line1), min_lcs=len(line1), flags=0, nch=len(chars), ch=chars + [0] * (64 - len(chars)),
                              entry=[NONE] * 64, lastviol=[0] * 64)
    lastpos = [line1.rindex(bytes([c])) for c in chars]
    cnt1 = [line1.count(bytes([c])) for c in chars]
    for li in range(1 if read_index_base == 0 else 0, n):
        q = names[li]
        i = read_index_base + li
        if len(q) > 255:
            r.flags |= 2; continue
        m = min(len(q), len(line1))
        lcp = 0
        while lcp < m and q[lcp] == line1[lcp]: lcp += 1
        lcs = 0
        while lcs < m and q[len(q) - 1 - lcs] == line1[len(line1) - 1 - lcs]: lcs += 1
        if (lcp == len(q) or lcs == len(q)) and len(q) < len(line1): r.flags |= 1
        r.min_lcp = min(r.min_lcp, lcp); r.min_lcs = min(r.min_lcs, lcs)
        for k, c in enumerate(chars):
            if lcp <= lastpos[k] and r.entry[k] == NONE: r.entry[k] = i
            if q.count(bytes([c])) != cnt1[k]: r.lastviol[k] = i
    return r


def qname_tokenise(ctx, buf, ls, n, plen, slen, separators):
    names = _names(buf, ls, n)
    ncols = len(separators) + 1
    vals = np.zeros((ncols, n), dtype=np.int64)
    strs = np.zeros((ncols, n, 8), dtype=np.uint8)
    r = types.SimpleNamespace(first_nonint=[NONE] * 32, vmin=[2 ** 63 - 1] * 32, vmax=[-2 ** 63] * 32, any_long=[0] * 32, flags=0)
    sepset = set(separators)
    for i, q in enumerate(names):
        if len(q) < plen + slen:
            r.flags |= 8; continue
        mid = q[plen:len(q) - slen]
        found = bytes(b for b in mid if b in sepset)
        if found != separators:
            r.flags |= 1
        fields, cur = [], b''
        for b in mid:
            if b in sepset and len(fields) < ncols - 1:
                fields.append(cur); cur = b''
            elif b not in sepset:
                cur += bytes([b])
        fields.append(cur)
        fields += [b''] * (ncols - len(fields))
        for c, f in enumerate(fields[:ncols]):
            if any(b == 0 or b >= 0x80 for b in f): r.any_long[c] |= 1
            if any(b == 32 or 9 <= b <= 13 for b in f): r.flags |= 2
            body = f[1:] if f[:1] in (b'+', b'-') else f
            isint = len(body) > 0 and all(48 <= b <= 57 for b in body)
            if isint and len(body) > 18:
                r.flags |= 4
            if isint and len(body) <= 18 and (f[:1] == b'+' or (len(body) > 1 and body[:1] == b'0') or (f[:1] == b'-' and int(body) == 0)):
                r.any_long[c] |= 4
            key = f[:8].ljust(8, b'\0')
            if len(f) > 8:
                if isint and body is f and not (len(f) > 1 and f[:1] == b'0') and len(body) <= 18:
                    key = ((1 << 63) | int(f)).to_bytes(8, 'big'); r.any_long[c] |= 2
                else:
                    r.any_long[c] |= 1
            strs[c, i] = np.frombuffer(key, dtype=np.uint8)
            if isint and len(body) <= 18:
                v = int(f)
                vals[c, i] = v
                r.vmin[c] = min(r.vmin[c], v); r.vmax[c] = max(r.vmax[c], v)
            elif not isint:
                r.first_nonint[c] = min(r.first_nonint[c], i)
    return ([torch.from_numpy(vals[c].copy()) for c in range(ncols)],
            [torch.from_numpy(strs[c].copy().view(np.int64).ravel()) for c in range(ncols)], r)


def unique_rows(ctx, table, rows, cols, want_key=True, want_sorted_key=True, want_unique=True):
    t = table.numpy().view(np.uint8).reshape(rows, cols)
    perm = np.lexsort(t.T[::-1]).astype(np.int32) if rows else np.zeros(0, np.int32)     # stable, most significant byte first
    s = t[perm]
    head = np.ones(rows, dtype=bool)
    if rows > 1: head[1:] = np.any(s[1:] != s[:-1], axis=1)
    skey = (np.cumsum(head) - 1).astype(np.int32)
    key = np.empty(rows, dtype=np.int32); key[perm] = skey
    uniq = s[head]
    return (torch.from_numpy(perm), torch.from_numpy(key), torch.from_numpy(skey), torch.from_numpy(uniq.ravel().copy()), int(head.sum()))


def prefix_distinct(ctx, perm, skey, n, thresholds):
    p, s = perm.numpy(), skey.numpy()
    head = np.ones(n, dtype=bool)
    if n > 1: head[1:] = s[1:] != s[:-1]
    firsts = p[head]
    return [int((firsts <= T).sum()) for T in thresholds]


_DT = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}
_VIEW = {1: np.uint8, 2: np.int16, 4: np.int32, 8: np.int64}


def int_prefix_distinct(ctx, val, n, vmin, value_range, thresholds, index_base=0):
    v = val.numpy()[:n]
    first = {}
    for i, x in enumerate(v.tolist()):
        if vmin <= x < vmin + value_range and x not in first: first[x] = index_base + i
    f = np.array(sorted(first.values()), dtype=np.uint64)
    return [int((f <= T).sum()) for T in thresholds]


def narrow(ctx, key, itemsize):
    return torch.from_numpy(key.numpy().astype(_DT[itemsize]).view(_VIEW[itemsize]))


def encode_int(ctx, val, sub, itemsize):
    return torch.from_numpy((val.numpy() - sub).astype(np.uint64).astype(_DT[itemsize]).view(_VIEW[itemsize]))
