// stats.hip -- pass-1 statistics on the device.
// Replaces the per-base dict increment `static_qualities[base][qual] += 1` and the length range /
// sanity checks of uq.py:366-375, 382, 388, 415-425.
//
// Workgroups are persistent (they keep a private LDS count table for the whole launch) and walk
// over tiles of R consecutive records.  A tile is one contiguous byte span: it is copied to LDS with
// 16-byte coalesced loads (the next tile's span bounds are requested one iteration ahead), then each
// half-wave (32 lanes) owns one record at a time and a lane takes 4 consecutive (base, quality) pairs
// per step (a 150-base read is one step of 38 lanes).  Counts go to LDS with fire-and-forget ds_add_u32,
// in three tiers per group of 8 positions:
//   1. all eight bases exactly A/C/G/T and qualities inside the window: an 8-fold replicated table
//      [4 bases][64 qualities][8 copies] (scattered LDS adds are bank-conflict bound: -0.14 ms on 10M x 150);
//   2. bases in [bbase, bbase+32), qualities in [qbase, qbase+64): one general table, slot = byte - corner;
//   3. anything else: a global atomic on the full 256 x 256 table -- every byte value is counted exactly.
// `bbase` / `qbase` are speed hints the host takes from the first record.  Tables are flushed to the u64
// global table once per workgroup.  A tile whose span exceeds the staging buffer (very long reads) is counted
// straight from HBM.
// Algorithmic HBM bytes: the record bytes once + 32 B of line offsets per record.
#include "common.h"
#include "swar.h"

namespace {
constexpr int ST_THREADS = 256;
constexpr uint32_t ST_NB = 32, ST_NQ = 64;  // general LDS table: 32 base bytes from bbase (64: upper case, 96: lower case) x 64 qualities from qbase
constexpr int ST_COPIES = 8;               // replication of the A/C/G/T table (bank-conflict control)
static_assert(4 * ST_NQ == 256, "one lane per A/C/G/T bin in the flush");
constexpr int ST_NV = 5;                  // 16-byte loads per lane per tile
constexpr uint32_t ST_CAP = ST_NV * 256 * 16;   // staging bytes per tile (20 KiB)
constexpr uint32_t ST_RMAX = 64;           // records per tile, upper bound (4 * ST_RMAX + 1 <= 2 * ST_THREADS)

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint32_t load_tail(const uint8_t* buf, uint64_t pos, uint64_t nbytes) {
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i)
        if (pos + i < nbytes) v |= (uint32_t)buf[pos + i] << (8 * i);
    return v;
}
// 4 consecutive bytes at LDS byte offset `o` (any alignment).
__device__ __forceinline__ uint32_t lds_load4(const uint8_t* base, uint32_t o) {
    const uint32_t* p = (const uint32_t*)(base + (o & ~3u));
    return __builtin_amdgcn_alignbyte(p[1], p[0], o & 3u);
}

// `win` = bbase << 8 | qbase: the corners of the LDS table's base / quality windows
__device__ __forceinline__ void count_pair(uint32_t b, uint32_t c, uint32_t win, uint32_t* hist, uq_stats* st) {
    const uint32_t sb = b - (win >> 8), sq = c - (win & 255u);
    if (sb < ST_NB && sq < ST_NQ) atomicAdd(&hist[sb * ST_NQ + sq], 1u);
    else atomicAdd((unsigned long long*)&st->counts[b * 256 + c], 1ull);
}
__device__ __forceinline__ void count_quad(uint32_t vb, uint32_t vq, uint32_t cnt, uint32_t win, uint32_t* hist, uq_stats* st) {
    count_pair(vb & 255u, vq & 255u, win, hist, st);
    if (cnt > 1) count_pair((vb >> 8) & 255u, (vq >> 8) & 255u, win, hist, st);
    if (cnt > 2) count_pair((vb >> 16) & 255u, (vq >> 16) & 255u, win, hist, st);
    if (cnt > 3) count_pair(vb >> 24, vq >> 24, win, hist, st);
}

struct Acc {
    uint32_t lmin = 0xFFFFFFFFu, lmax = 0, rmax = 0;
    uint64_t bad_plus = UQ_NONE, bad_len = UQ_NONE;
    __device__ __forceinline__ void record(uint64_t gr, bool plus_ok, uint32_t L, uint32_t Lq, uint32_t rb) {
        if (!plus_ok) bad_plus = bad_plus < gr ? bad_plus : gr;
        if (L != Lq) bad_len = bad_len < gr ? bad_len : gr;
        lmin = L < lmin ? L : lmin;
        lmax = L > lmax ? L : lmax;
        rmax = rb > rmax ? rb : rmax;
    }
};

__global__ __launch_bounds__(ST_THREADS) void stats_kernel(const uint8_t* __restrict__ buf, const uint64_t* __restrict__ ls, uint64_t first,
                                                           uint64_t n, uq_stats* __restrict__ st) {
    __shared__ uint32_t hist[ST_NB * ST_NQ];
    __shared__ uint32_t fast[4 * ST_NQ * ST_COPIES];          // [A C T G][quality slot][copy]
    __shared__ __align__(16) uint8_t stage[ST_CAP + 32];
    __shared__ uint32_t meta[4 * ST_RMAX + 4];
    __shared__ uint32_t cfg[3];
    const uint32_t tid = threadIdx.x, lane = lane_id();
    for (int i = threadIdx.x; i < (int)(ST_NB * ST_NQ); i += ST_THREADS) hist[i] = 0;
    for (int i = threadIdx.x; i < (int)(4 * ST_NQ * ST_COPIES); i += ST_THREADS) fast[i] = 0;
    // Launch geometry is decided here, not on the host (no device -> host round trip before the launch):
    // wave 0 places the table windows from the first record (speed hints: any byte is still counted exactly)
    // and sizes the tile from the shard's average record length so that a typical span fits the staging buffer.
    const uint64_t nbytes = ls[4 * (first + n)];              // end of the shard: bound of the unaligned tail loads
    if (tid < 64) {
        const uint64_t* l0 = ls + 4 * first;
        const uint64_t s = l0[1], e1 = l0[2], q = l0[3], e = l0[4];
        uint64_t len = e > q + 1 ? e - q - 1 : 0;
        if (len > 4096) len = 4096;
        uint32_t mn = 255;
        for (uint64_t j = lane; j < len; j += 64) { const uint32_t c = buf[q + j]; mn = c < mn ? c : mn; }
        mn = wave_min(mn);
        if (lane == 0) {
            uint32_t qb = 33, bb = 64;
            if (len) { if (mn >= 64) qb = 59; else if (mn < 33) qb = 0; }   // Phred+64 era files: window [59, 123)
            if (e1 > s + 1 && buf[s] >= 96) bb = 96;                           // lower-case reads
            const uint64_t avg = (nbytes - l0[0]) / n + 1;
            uint64_t R = (ST_CAP - 64) / (avg + avg / 8 + 1);
            R = R > ST_RMAX ? ST_RMAX : (R < 1 ? 1 : R);
            uint32_t P = ST_THREADS / (uint32_t)R;
            P = P > 16 ? 16 : (P < 1 ? 1 : P);
            cfg[0] = (bb << 8) | qb; cfg[1] = (uint32_t)R; cfg[2] = P;
        }
    }
    __syncthreads();
    // wave-uniform values: keep them in SGPRs (an LDS read lands in a VGPR and would drag all tile arithmetic onto the VALU)
    const uint32_t win = __builtin_amdgcn_readfirstlane(cfg[0]), R = __builtin_amdgcn_readfirstlane(cfg[1]),
                   P = __builtin_amdgcn_readfirstlane(cfg[2]);
    const uint32_t qbase = win & 255u, bbase = win >> 8;       // bbase is 64 or 96 (the XOR test below needs a multiple of 32)

    const uint32_t hl = lane & 31;
    const uint32_t hw = (tid >> 6) * 2 + (lane >> 5);       // half-wave index inside the workgroup, 0..7
    constexpr uint32_t NHW = 2 * (ST_THREADS / 64);
    Acc acc;
    uint32_t fill_fast = 0, fill_hist = 0;      // pairs (window base + 1, quality slot 0) counted for bytes beyond a read's end
    const uint64_t ntiles = (n + R - 1) / R;
    const uint32_t rr = tid / P, pp = tid - rr * P;   // this lane counts groups pp, pp + P, ... of read rr of every tile
    const uint32_t q_addlo = 0x01010101u * (0x80u - qbase), q_addhi = 0x01010101u * (0x80u - qbase - ST_NQ);
    const uint32_t b_xor = 0x01010101u * bbase;

    // Software pipeline over this workgroup's tiles t, t + S, t + 2S, ... (S = gridDim.x):
    //   span bounds are requested two tiles ahead, the tile's bytes + line offsets one tile ahead (they
    //   stay in registers, in flight, while the current tile is counted out of LDS).
    const uint64_t S = gridDim.x;
    struct Bounds { uint64_t g0, g1; };
    struct Regs { uint4 v[ST_NV]; uint64_t m0, m1; uint64_t g0; uint32_t skew, nvec, Rt; bool exists, staged; };
    auto load_bounds = [&](uint64_t tt) {
        Bounds b{0, 0};
        if (tt < ntiles) {
            const uint32_t Rn = (uint32_t)((n - tt * R) < R ? (n - tt * R) : R);
            b.g0 = ls[4 * (first + tt * R)]; b.g1 = ls[4 * (first + tt * R) + 4 * Rn];
        }
        return b;
    };
    auto issue = [&](uint64_t tt, Bounds b) {
        Regs x;
        x.exists = tt < ntiles; x.staged = false; x.m0 = x.m1 = 0; x.g0 = b.g0; x.skew = 0; x.nvec = 0; x.Rt = 0;
        for (int u = 0; u < ST_NV; ++u) x.v[u] = make_uint4(0, 0, 0, 0);
        if (!x.exists) return x;
        x.Rt = (uint32_t)((n - tt * R) < R ? (n - tt * R) : R);
        const uint64_t a0 = ((uint64_t)(uintptr_t)buf + b.g0) & ~uint64_t(15);
        x.skew = (uint32_t)(((uint64_t)(uintptr_t)buf + b.g0) - a0);
        const uint64_t span = b.g1 - b.g0 + x.skew;
        x.staged = span + 16 <= ST_CAP;
        if (!x.staged) return x;
        x.nvec = (uint32_t)((span + 15) >> 4);
        const uint4* src = (const uint4*)(buf + ((int64_t)b.g0 - (int64_t)x.skew));      // (global_load, not flat_load: see pack.hip)
#pragma unroll
        for (int u = 0; u < ST_NV; ++u) { const uint32_t i = u * ST_THREADS + tid; if (i < x.nvec) x.v[u] = src[i]; }
        const uint64_t* lsp = ls + 4 * (first + tt * R);
        if (tid <= 4 * x.Rt) x.m0 = lsp[tid];
        if (tid + ST_THREADS <= 4 * x.Rt) x.m1 = lsp[tid + ST_THREADS];
        return x;
    };

    uint64_t t = blockIdx.x;
    Bounds b_next = load_bounds(t + S);
    Regs cur = issue(t, load_bounds(t));
    for (; t < ntiles; t += S) {
        const uint64_t r0 = t * R;
        const uint32_t Rt = cur.Rt;
        const Bounds b_nn = load_bounds(t + 2 * S);
        if (cur.staged) {
            if (tid <= 4 * Rt) meta[tid] = (uint32_t)(cur.m0 - cur.g0) + cur.skew;
            if (tid + ST_THREADS <= 4 * Rt) meta[tid + ST_THREADS] = (uint32_t)(cur.m1 - cur.g0) + cur.skew;
#pragma unroll
            for (int u = 0; u < ST_NV; ++u) { const uint32_t i = u * ST_THREADS + tid; if (i < cur.nvec) ((uint4*)stage)[i] = cur.v[u]; }
        }
        __syncthreads();
        const bool staged = cur.staged;
        cur = issue(t + S, b_next);          // next tile's loads fly while this one is counted
        b_next = b_nn;
        if (staged) {
            // P lanes per read: lane pp == 0 does the per-record checks, every lane counts groups of 8 positions
            if (rr < Rt) {
                const uint32_t r = rr;
                const uint32_t p0 = meta[4 * r], s = meta[4 * r + 1], e1 = meta[4 * r + 2], q = meta[4 * r + 3], e2 = meta[4 * r + 4];
                const uint32_t L = e1 - s - 1, Lq = e2 - q - 1;
                if (pp == 0) acc.record(r0 + r, stage[e1] == '+', L, Lq, e2 - p0);
                const uint32_t Lc = L < Lq ? L : Lq;
                for (uint32_t j = 8 * pp; j < Lc; j += 8 * P) {
                    uint32_t b_lo, b_hi, q_lo, q_hi;
                    lds_window8(stage, (int32_t)(s + j), b_lo, b_hi);
                    lds_window8(stage, (int32_t)(q + j), q_lo, q_hi);
                    const uint32_t cnt = Lc - j;
                    // The read's last group holds cnt < 8 pairs.  One lane in ceil(L / 8) holds it, i.e. every wave: sent down
                    // the per-byte path it made every wave run both forms.  Instead its missing bytes become (window base + 1,
                    // quality slot 0) -- in range, so the group takes the same tier as its neighbours -- and the
                    // pairs counted too many are taken off that one bin when the tables are flushed (fill_fast / fill_hist).
                    const uint32_t nb = cnt < 8 ? cnt : 8u;
                    {
                        const uint32_t nlo = nb < 4 ? nb : 4u, nhi = nb - nlo;
                        const uint32_t keep_lo = nlo == 4 ? 0xFFFFFFFFu : (1u << (8 * nlo)) - 1u, keep_hi = nhi == 4 ? 0xFFFFFFFFu : (1u << (8 * nhi)) - 1u;
                        const uint32_t bfill = 0x01010101u * (bbase + 1), qfill = 0x01010101u * qbase;
                        b_lo = bfi(keep_lo, b_lo, bfill); b_hi = bfi(keep_hi, b_hi, bfill);
                        q_lo = bfi(keep_lo, q_lo, qfill); q_hi = bfi(keep_hi, q_hi, qfill);
                    }
                    // byte - window base, valid when the top bits vanish: bases bbase..bbase+31, qualities qbase..qbase+63
                    const uint32_t sb0 = b_lo ^ b_xor, sb1 = b_hi ^ b_xor;
                    const uint32_t u0 = q_lo + q_addlo, u1 = q_hi + q_addlo;
                    const uint32_t bad = ((sb0 | sb1) & 0xE0E0E0E0u) |
                                         ((q_lo | (q_lo + q_addhi) | ~u0 | q_hi | (q_hi + q_addhi) | ~u1) & 0x80808080u);
                    // tier 1: all eight bases are exactly A/C/G/T -> bin byte = code << 6 | quality slot, counted in the
                    // ST_COPIES-fold replicated table (copy = lane % 8: the bank is (quality & 3) * 8 + lane % 8, so the
                    // four lanes of a half-wave that share a copy collide only on equal low quality bits -- mostly
                    // <= 2-way, which ds_add_u32 absorbs at no cost; the single-copy table is ~3.5-way on random data)
                    const uint32_t c0 = (b_lo >> 1) & 0x03030303u, c1 = (b_hi >> 1) & 0x03030303u;
                    const bool acgt = __builtin_amdgcn_perm(0u, 0x47544341u, c0) == b_lo && __builtin_amdgcn_perm(0u, 0x47544341u, c1) == b_hi;
                    if (bad == 0 && acgt) {
                        const uint32_t bin0 = (c0 << 6) | (u0 & 0x7F7F7F7Fu), bin1 = (c1 << 6) | (u1 & 0x7F7F7F7Fu);
                        uint8_t* hb = (uint8_t*)fast + ((lane & (ST_COPIES - 1)) << 2);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            atomicAdd((uint32_t*)(hb + (((bin0 >> (8 * k)) & 0xFFu) << 5)), 1u);
                            atomicAdd((uint32_t*)(hb + (((bin1 >> (8 * k)) & 0xFFu) << 5)), 1u);
                        }
                        fill_fast += 8u - nb;
                    } else if (bad == 0) {
                        const uint32_t sq0 = (u0 & 0x7F7F7F7Fu) << 2, sq1 = (u1 & 0x7F7F7F7Fu) << 2;   // quality slot * 4 per byte
                        uint8_t* hb = (uint8_t*)hist;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            atomicAdd((uint32_t*)(hb + ((((sb0 >> (8 * k)) & 0xFFu) << 8) | ((sq0 >> (8 * k)) & 0xFFu))), 1u);
                            atomicAdd((uint32_t*)(hb + ((((sb1 >> (8 * k)) & 0xFFu) << 8) | ((sq1 >> (8 * k)) & 0xFFu))), 1u);
                        }
                        fill_hist += 8u - nb;
                    } else {
                        count_quad(b_lo, q_lo, cnt, win, hist, st);
                        if (cnt > 4) count_quad(b_hi, q_hi, cnt - 4, win, hist, st);
                    }
                }
            }
        } else {
            // oversize tile: straight from HBM, one record per half-wave
            const uint64_t* lsp = ls + 4 * (first + r0);
            for (uint32_t r = hw; r < Rt; r += NHW) {
                const uint64_t* p = lsp + 4 * r;
                const uint64_t p0 = p[0], s = p[1], e1 = p[2], q = p[3], e2 = p[4];
                const uint32_t L = (uint32_t)(e1 - s - 1), Lq = (uint32_t)(e2 - q - 1);
                if (hl == 0) acc.record(r0 + r, buf[e1] == '+', L, Lq, (uint32_t)(e2 - p0));
                const uint32_t Lc = L < Lq ? L : Lq;
                for (uint32_t j = hl * 4; j < Lc; j += 128) {
                    const uint32_t vb = load_u32_unaligned(buf + s + j);   // s + j + 3 <= e1 + 1 < nbytes always
                    const uint32_t vq = (q + j + 4 <= nbytes) ? load_u32_unaligned(buf + q + j) : load_tail(buf, q + j, nbytes);
                    count_quad(vb, vq, Lc - j, win, hist, st);
                }
            }
        }
        __syncthreads();
    }
    if (fill_fast) atomicSub(&fast[lane & (ST_COPIES - 1)], fill_fast);          // bin 0 = ('A', slot 0); any copy: the flush sums them mod 2^32
    if (fill_hist) atomicSub(&hist[1 * ST_NQ], fill_hist);                                                               // base bbase + 1, slot 0
    __syncthreads();
    for (int i = threadIdx.x; i < (int)(ST_NB * ST_NQ); i += ST_THREADS) {
        const uint32_t v = hist[i];
        if (v) atomicAdd((unsigned long long*)&st->counts[(bbase + i / ST_NQ) * 256 + qbase + (i % ST_NQ)], (unsigned long long)v);
    }
    {
        // the replicated A/C/T/G table: one lane per bin sums its copies
        const uint32_t bin = threadIdx.x;                   // 4 * ST_NQ == ST_THREADS
        uint32_t v = 0;
#pragma unroll
        for (int c = 0; c < ST_COPIES; ++c) v += fast[bin * ST_COPIES + ((c + bin) & (ST_COPIES - 1))];
        const uint32_t base = (0x47544341u >> (8 * (bin >> 6))) & 0xFFu;
        if (v) atomicAdd((unsigned long long*)&st->counts[base * 256 + qbase + (bin & 63u)], (unsigned long long)v);
    }
    const uint32_t lmin = wave_min(acc.lmin), lmax = wave_max(acc.lmax), rmax = wave_max(acc.rmax);
    const uint64_t bad_plus = wave_min(acc.bad_plus), bad_len = wave_min(acc.bad_len);
    if (lane == 0) {
        if (lmin != 0xFFFFFFFFu) atomicMin(&st->len_min, lmin);
        atomicMax(&st->len_max, lmax);
        atomicMax(&st->max_record_bytes, rmax);
        if (bad_plus != UQ_NONE) atomicMin((unsigned long long*)&st->bad_plus, (unsigned long long)(first + bad_plus));
        if (bad_len != UQ_NONE) atomicMin((unsigned long long*)&st->bad_len, (unsigned long long)(first + bad_len));
    }
}

// ---- multi-GPU: a uq_stats as ONE summable buffer.  words [0, 65536) = counts; then 6 words per rank:
// the rank's bad_plus / bad_len (made file-wide, sign bit flipped so that signed order == unsigned order), len_min,
// len_max, max_record_bytes, the `incomplete` flag in its own slots and zeros in the others -- after an all-reduce SUM every rank holds
// everybody's scalars and folds them with MIN / MAX.
constexpr long long ST_SIGN = (long long)0x8000000000000000ull;

__global__ void stats_export_kernel(const uq_stats* __restrict__ st, uint32_t rank, uint32_t world, uint64_t read_offset, long long* __restrict__ buf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 65536) { buf[i] = (long long)st->counts[i]; return; }
    const uint32_t j = i - 65536;
    if (j >= world * 6) return;
    long long v = 0;
    if (j / 6 == rank) {
        switch (j % 6) {
            case 0: v = (long long)(st->bad_plus == UQ_NONE ? UQ_NONE : st->bad_plus + read_offset) ^ ST_SIGN; break;
            case 1: v = (long long)(st->bad_len == UQ_NONE ? UQ_NONE : st->bad_len + read_offset) ^ ST_SIGN; break;
            case 2: v = st->len_min; break;
            case 3: v = st->len_max; break;
            case 4: v = st->max_record_bytes; break;
            default: v = st->reserved; break;
        }
    }
    buf[i] = v;
}

__global__ void stats_import_kernel(const long long* __restrict__ buf, uint32_t world, uq_stats* __restrict__ st) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 65536) st->counts[i] = (uint64_t)buf[i];
    if (i == 0) {
        long long bp = 0x7FFFFFFFFFFFFFFFll, bl = bp, lmin = bp, lmax = 0, rmax = 0, inc = 0;
        for (uint32_t r = 0; r < world; ++r) {
            const long long* s = buf + 65536 + 6 * r;
            bp = s[0] < bp ? s[0] : bp; bl = s[1] < bl ? s[1] : bl; lmin = s[2] < lmin ? s[2] : lmin;
            lmax = s[3] > lmax ? s[3] : lmax; rmax = s[4] > rmax ? s[4] : rmax; inc = s[5] > inc ? s[5] : inc;
        }
        st->bad_plus = (uint64_t)(bp ^ ST_SIGN); st->bad_len = (uint64_t)(bl ^ ST_SIGN);
        st->len_min = (uint32_t)lmin; st->len_max = (uint32_t)lmax; st->max_record_bytes = (uint32_t)rmax; st->reserved = (uint32_t)inc;
    }
}

// ---- the statistics as a short list: a file uses a few hundred of the 65 536 (base, quality) counters, and reading 512 KiB back
// (pageable: staged) costs more than the decisions that follow
constexpr uint32_t SC_CAP = UQ_STATS_COMPACT_CAP;
typedef uq_stats_compact StatsCompact;
__global__ void stats_compact_kernel(const uq_stats* __restrict__ st, StatsCompact* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        out->len_min = st->len_min; out->len_max = st->len_max; out->max_record_bytes = st->max_record_bytes; out->reserved = st->reserved;
        out->bad_plus = st->bad_plus; out->bad_len = st->bad_len;
    }
    if (i >= 65536) return;
    const uint64_t c = st->counts[i];
    if (c) {
        const uint32_t k = atomicAdd(&out->n, 1u);
        if (k < SC_CAP) { out->key[k] = i; out->count[k] = c; }
    }
}

__global__ void stats_init_kernel(uq_stats* st) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 65536) st->counts[i] = 0;
    if (i == 0) {
        st->bad_plus = UQ_NONE; st->bad_len = UQ_NONE;
        st->len_min = 0xFFFFFFFFu; st->len_max = 0; st->max_record_bytes = 0; st->reserved = 0;
    }
}

// First occurrence of every base byte (ordering of N-trick candidates, uq.py:480 under pypy/py3).
__global__ __launch_bounds__(256) void first_occurrence_kernel(const uint8_t* __restrict__ buf,
                                                               const uint64_t* __restrict__ ls, uint64_t first,
                                                               uint64_t n, uint64_t index_base, uint64_t* __restrict__ out) {
    __shared__ unsigned long long seen[256];
    seen[threadIdx.x] = UQ_NONE;
    __syncthreads();
    const uint64_t gw = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6));
    const uint64_t GW = (uint64_t)gridDim.x * 4;
    const uint32_t lane = lane_id();
    for (uint64_t r = gw; r < n; r += GW) {
        const uint64_t* p = ls + 4 * (first + r);
        const uint64_t s = p[1], e1 = p[2];
        const uint32_t L = (uint32_t)(e1 - s - 1);
        for (uint32_t j = lane; j < L; j += 64) {
            uint32_t b = buf[s + j];
            unsigned long long key = ((index_base + r) << 20) | (j & 0xFFFFFu);
            if (key < seen[b]) atomicMin(&seen[b], key);
        }
    }
    __syncthreads();
    if (seen[threadIdx.x] != UQ_NONE) atomicMin((unsigned long long*)&out[threadIdx.x], seen[threadIdx.x]);
}
}  // namespace

// the non-zero counters as a list (n entries, any order) + the scalars, in the context's pinned staging; n > SC_CAP: list cut short
static int stats_fetch_compact(uq_ctx* ctx, const uq_stats* d_stats, const StatsCompact** out) {
    static_assert(sizeof(StatsCompact) <= 63000 && sizeof(StatsCompact) % 4 == 0, "the compact form must fit the context's pinned staging buffer");
    void* scr;
    UQ_TRY(uq_scratch(ctx, sizeof(StatsCompact), &scr));
    UQ_CHECK_HIP(hipMemsetAsync(scr, 0, 8, ctx->stream));
    stats_compact_kernel<<<65536 / 256, 256, 0, ctx->stream>>>(d_stats, (StatsCompact*)scr);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned, scr, sizeof(StatsCompact)));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *out = (const StatsCompact*)ctx->h_pinned;
    return 0;
}

extern "C" int uq_stats_fetch(uq_ctx* ctx, const uq_stats* d_stats, uq_stats* h_stats) {
    UQ_REQUIRE(ctx && d_stats && h_stats, "uq_stats_fetch: null argument");
    const StatsCompact* c;
    UQ_TRY(stats_fetch_compact(ctx, d_stats, &c));
    if (c->n > SC_CAP) {                                   // an unusually rich file: the whole table
        UQ_CHECK_HIP(hipMemcpyAsync(h_stats, d_stats, sizeof(uq_stats), hipMemcpyDeviceToHost, ctx->stream));
        UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        return 0;
    }
    memset(h_stats->counts, 0, sizeof(h_stats->counts));
    for (uint32_t k = 0; k < c->n; ++k) h_stats->counts[c->key[k]] = c->count[k];
    h_stats->bad_plus = c->bad_plus; h_stats->bad_len = c->bad_len; h_stats->len_min = c->len_min; h_stats->len_max = c->len_max;
    h_stats->max_record_bytes = c->max_record_bytes; h_stats->reserved = c->reserved;
    return 0;
}

extern "C" int uq_stats_fetch_compact(uq_ctx* ctx, const uq_stats* d_stats, uq_stats_compact* h_out) {
    UQ_REQUIRE(ctx && d_stats && h_out, "uq_stats_fetch_compact: null argument");
    const StatsCompact* c;
    UQ_TRY(stats_fetch_compact(ctx, d_stats, &c));
    const uint32_t n = c->n < SC_CAP ? c->n : SC_CAP;
    memcpy(h_out, c, offsetof(StatsCompact, key));
    memcpy(h_out->key, c->key, n * sizeof(uint32_t));
    memcpy(h_out->count, c->count, n * sizeof(uint64_t));
    return 0;
}

extern "C" int uq_stats_init(uq_ctx* ctx, uq_stats* d_stats) {
    UQ_REQUIRE(ctx && d_stats, "uq_stats_init: null argument");
    stats_init_kernel<<<256, 256, 0, ctx->stream>>>(d_stats);
    UQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int uq_stats_export(uq_ctx* ctx, const uq_stats* d_stats, uint32_t rank, uint32_t world, uint64_t read_offset, int64_t* d_words) {
    UQ_REQUIRE(ctx && d_stats && d_words && world >= 1 && rank < world, "uq_stats_export: bad argument");
    const uint32_t n = 65536 + 6 * world;
    stats_export_kernel<<<(n + 255) / 256, 256, 0, ctx->stream>>>(d_stats, rank, world, read_offset, (long long*)d_words);
    UQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int uq_stats_import(uq_ctx* ctx, const int64_t* d_words, uint32_t world, uq_stats* d_stats) {
    UQ_REQUIRE(ctx && d_stats && d_words && world >= 1, "uq_stats_import: bad argument");
    stats_import_kernel<<<256, 256, 0, ctx->stream>>>((const long long*)d_words, world, d_stats);
    UQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int uq_stats_accumulate(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start,
                                   uint64_t first_read, uint64_t nreads, uq_stats* d_stats) {
    UQ_REQUIRE(ctx && d_buf && d_line_start && d_stats, "uq_stats_accumulate: null argument");
    if (nreads == 0) return 0;
    // No host round trip: tile size and table windows are derived on the device (kernel prologue).  The grid
    // only needs an upper bound of the tile count (tiles hold >= 1 record); surplus workgroups exit at once.
    const uint32_t blocks = (uint32_t)(nreads < UQ_NUM_CU * 4 ? nreads : UQ_NUM_CU * 4);
    stats_kernel<<<blocks, ST_THREADS, 0, ctx->stream>>>(d_buf, d_line_start, first_read, nreads, d_stats);
    UQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int uq_first_occurrence(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start,
                                   uint64_t first_read, uint64_t nreads, uint64_t read_index_base, uint64_t* d_first) {
    UQ_REQUIRE(ctx && d_buf && d_line_start && d_first, "uq_first_occurrence: null argument");
    if (nreads == 0) return 0;
    uint32_t blocks = (uint32_t)((nreads + 3) / 4);
    if (blocks > UQ_NUM_CU * 8) blocks = UQ_NUM_CU * 8;
    first_occurrence_kernel<<<blocks, 256, 0, ctx->stream>>>(d_buf, d_line_start, first_read, nreads, read_index_base, d_first);
    UQ_LAUNCH_CHECK();
    return 0;
}
