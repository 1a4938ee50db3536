// msd.hip -- round 0 of the row sort as an MSD radix partition finished in LDS (round 4; replaces the LSD passes of radix.hip for
// tables whose keys spread: numpy.argsort(table.view('V<C>'), axis=0), uq.py:773-777, 786-798).
//
// A table of n rows needs about log2(n) bits of sorting, not 32 or 64.  The rows' heads (their first eight bytes behind the z leading
// bits every row shares, as a big-endian number; the top 32 bits of that for tables that do not crowd on them) are partitioned
// MSD-first on T = log2(n / 64) bits in two or three levels of at most eight bits, then every run of neighbouring buckets that fits
// a workgroup's LDS is put in its final order there:
//   extract   rows -> keys, the level-1 histogram, AND / OR over all heads (the check of z)                    reads the row heads once
//   level l   per-parent histograms of the next digit (count: reads the keys), exclusive scan = the buckets' first slots, then the
//             scatter: a tile of one parent bucket, ONE returning LDS atomic per key = its rank among the tile's keys of that digit,
//             one returning global atomic per (tile, digit) on the bucket's cursor = where the tile's run goes, the tile regrouped in
//             LDS and stored in runs.  The order INSIDE a bucket is whatever the atomics gave -- no workgroup waits for another
//             (no look-back, no dispatch-order assumption), and nothing has to be stable, because ...
//   finish    ... the pairs (key, row number) of a chunk (neighbouring buckets, < 2048 pairs) are sorted in LDS by (key, row number):
//             binned by the key's position in the chunk's key range, ranked inside the bin by counting.  Row number as the tie-break
//             IS the stable order (uq.py's argsort under the Q17 rule), whatever the levels did.  The head flags (key differs from the
//             key in front) fall out of the same counting loop.
//   ties      rows wider than the key: the positions where a group of equal keys starts are listed per chunk by the finishing kernel;
//             msd_ties_kernel sorts each group by whole rows, a sub-wave of four or eight lanes per pair of rows (every lane fetches
//             sixteen bytes of each row), and sets the final flags (1 = a new row value, 2 = equal to the row in front).
// (tools/variants.sh msd.hip <name> -DMSD_ABL=2 builds the finishing kernel without its rank loop: the ablation DESIGN.md section 11 quotes.)
// Tables whose buckets come out heavier than a chunk (few distinct heads: QNAME columns, a read copied a million times) are
// reported back (*status = 1) before the last scatter and take the LSD passes as before.
#include "radix.h"

namespace {
constexpr int MT = 256;
constexpr uint32_t MSD_G = 1024;            // chunk c = the buckets that start in slots [c G, (c + 1) G): fewer than G + the largest bucket pairs
constexpr uint32_t MSD_CAP = 2048;          // pairs a finishing workgroup holds in LDS
constexpr uint32_t MSD_NB = 2048;           // bins of the in-LDS step
constexpr uint32_t MSD_MAXCHILD = 1024;     // widest digit: ten bits
constexpr int MSD_MAXLEVELS = 4;

template <typename K> struct MsdGeom { static constexpr int ITEMS = sizeof(K) == 4 ? 16 : 12; static constexpr int TILE = MT * ITEMS; };

__device__ __forceinline__ uint64_t msd_chunk0(const uint8_t* __restrict__ row, uint32_t C) {
    uint64_t v;
    if (C >= 8) { __builtin_memcpy(&v, row, 8); return __builtin_bswap64(v); }
    v = 0;
    for (uint32_t i = 0; i < 8; ++i) v = (v << 8) | (i < C ? row[i] : 0);
    return v;
}

template <typename K>
__device__ __forceinline__ K msd_min(K a, K b) { return a < b ? a : b; }
template <typename K>
__device__ __forceinline__ K msd_max(K a, K b) { return a > b ? a : b; }

// rows -> keys[i] = the top 8 sizeof(K) bits of (head << z); ghist[d] += keys whose top digit is d; andor[0 .. 1] = AND / OR over the heads
template <typename K>
__global__ __launch_bounds__(MT) void msd_extract_kernel(const uint8_t* __restrict__ table, uint32_t C, uint64_t n, uint32_t z, K* __restrict__ keys,
                                                         uint32_t shift, uint32_t nchild, uint32_t* __restrict__ ghist, unsigned long long* __restrict__ andor) {
    constexpr int ITEMS = MsdGeom<K>::ITEMS, TILE = MsdGeom<K>::TILE;
    __shared__ uint32_t h[MSD_MAXCHILD];
    __shared__ unsigned long long sa[MT / 64], so[MT / 64];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < nchild; i += MT) h[i] = 0;
    __syncthreads();
    unsigned long long a = ~0ull, o = 0ull;
    const uint64_t ntiles = (n + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t base = tile * TILE;
        auto take = [&](uint64_t idx, uint64_t c) {
            a &= c; o |= c;
            const K k = (K)((c << z) >> (64 - 8 * sizeof(K)));
            keys[idx] = k;
            atomicAdd(&h[(uint32_t)(k >> shift)], 1u);
        };
        if (C >= 8) {
            // all of the lane's row heads are requested before the first is used (unconditional loads with a clamped row number: a load behind `idx < n` is
            // followed by its own wait -- one head in flight per lane), and the row width is tested once for all of them
            uint64_t c64[ITEMS];
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const uint64_t idx = base + (uint64_t)i * MT + tid;
                uint64_t v; __builtin_memcpy(&v, table + (idx < n ? idx : n - 1) * C, 8);
                c64[i] = __builtin_bswap64(v);
            }
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) { const uint64_t idx = base + (uint64_t)i * MT + tid; if (idx < n) take(idx, c64[i]); }
        } else {
            // rows of fewer than eight bytes (up to eight byte loads a row): one row at a time -- sixteen rows' byte loads at once cost the kernel 252 VGPRs
#pragma unroll 1
            for (int i = 0; i < ITEMS; ++i) { const uint64_t idx = base + (uint64_t)i * MT + tid; if (idx < n) take(idx, msd_chunk0(table + idx * C, C)); }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a &= __shfl_xor(a, d, 64); o |= __shfl_xor(o, d, 64); }
    if ((tid & 63) == 0) { sa[tid >> 6] = a; so[tid >> 6] = o; }
    __syncthreads();
    for (uint32_t i = tid; i < nchild; i += MT)
        if (h[i]) atomicAdd(&ghist[i], h[i]);
    if (tid == 0) {
        for (int w = 1; w < MT / 64; ++w) { a &= sa[w]; o |= so[w]; }
        atomicAnd(andor, a); atomicOr(andor + 1, o);
    }
}

// tiles per parent bucket (entry np: 0, so that the exclusive scan over np + 1 entries ends in the total)
__global__ __launch_bounds__(MT) void msd_parent_tiles_kernel(const uint32_t* __restrict__ parent_end, uint32_t np, uint32_t tile, uint32_t* __restrict__ ntiles) {
    const uint32_t p = blockIdx.x * MT + threadIdx.x;
    if (p > np) return;
    if (p == np) { ntiles[p] = 0; return; }
    const uint32_t lo = p ? parent_end[p - 1] : 0u;
    ntiles[p] = (parent_end[p] - lo + tile - 1) / tile;
}

// the parent bucket tile t belongs to: the largest p with tile_prefix[p] <= t (empty parents in front of it are skipped by construction)
__device__ __forceinline__ uint32_t msd_find_parent(const uint32_t* __restrict__ tile_prefix, uint32_t np, uint32_t t) {
    uint32_t lo = 0, hi = np;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tile_prefix[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// hist[p * nchild + d] = keys of parent bucket p whose digit is d.  A workgroup takes a run of consecutive tiles: one search, then it walks.
template <typename K>
__global__ __launch_bounds__(MT) void msd_count_kernel(const K* __restrict__ keys, const uint32_t* __restrict__ parent_end, const uint32_t* __restrict__ tile_prefix,
                                                       uint32_t np, uint32_t shift, uint32_t nchild, uint32_t* __restrict__ hist) {
    constexpr int ITEMS = MsdGeom<K>::ITEMS, TILE = MsdGeom<K>::TILE;
    __shared__ uint32_t h[MSD_MAXCHILD];
    const uint32_t tid = threadIdx.x;
    const uint32_t ntiles = tile_prefix[np];
    const uint32_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    uint32_t t = blockIdx.x * per;
    const uint32_t tend = t + per < ntiles ? t + per : ntiles;
    if (per == 0 || t >= tend) return;
    uint32_t p = msd_find_parent(tile_prefix, np, t);
    for (uint32_t i = tid; i < nchild; i += MT) h[i] = 0;
    __syncthreads();
    uint32_t cur = p;
    const uint32_t mask = nchild - 1;
    for (; t < tend; ++t) {
        while (p + 1 < np && tile_prefix[p + 1] <= t) ++p;
        if (p != cur) {
            __syncthreads();
            for (uint32_t i = tid; i < nchild; i += MT) { const uint32_t c = h[i]; if (c) atomicAdd(&hist[(uint64_t)cur * nchild + i], c); h[i] = 0; }
            __syncthreads();
            cur = p;
        }
        const uint32_t pend = parent_end[p];
        const uint32_t lo = (p ? parent_end[p - 1] : 0u) + (t - tile_prefix[p]) * TILE;
        const uint32_t m = pend - lo < (uint32_t)TILE ? pend - lo : (uint32_t)TILE;
        K kv[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { const uint32_t q = i * MT + tid; kv[i] = keys[lo + (q < m ? q : 0u)]; }      // all of the lane's keys requested before the first is counted
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t q = i * MT + tid;
            if (q < m) atomicAdd(&h[(uint32_t)(kv[i] >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < nchild; i += MT) { const uint32_t c = h[i]; if (c) atomicAdd(&hist[(uint64_t)cur * nchild + i], c); }
}

__global__ __launch_bounds__(MT) void msd_max_kernel(const uint32_t* __restrict__ hist, uint64_t nb, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * MT + threadIdx.x; i < nb; i += (uint64_t)gridDim.x * MT) m = hist[i] > m ? hist[i] : m;
    m = wave_max(m);
    if (lane_id() == 0 && m) atomicMax(out, m);
}

// One level of the partition.  cursors[p * nchild + d] holds the next free slot of bucket (p, d) (the scanned histogram before the
// launch, the buckets' ends after it).  FIRST: the keys as msd_extract_kernel left them, row number = position, one parent.
template <typename K, bool FIRST>
__global__ __launch_bounds__(MT) void msd_scatter_kernel(const K* __restrict__ keys_in, const uint32_t* __restrict__ idx_in, K* __restrict__ keys_out,
                                                         uint32_t* __restrict__ idx_out, uint32_t n, const uint32_t* __restrict__ parent_end,
                                                         const uint32_t* __restrict__ tile_prefix, uint32_t np, uint32_t shift, uint32_t nchild,
                                                         uint32_t* __restrict__ cursors) {
    constexpr int ITEMS = MsdGeom<K>::ITEMS, TILE = MsdGeom<K>::TILE;
    __shared__ K s_keys[TILE];
    __shared__ uint32_t s_idx[TILE];
    __shared__ uint32_t s_h[MSD_MAXCHILD];       // the tile's count per digit, then the digit's first slot in the tile
    __shared__ uint32_t s_g[MSD_MAXCHILD];       // the digit's run in the output: first slot - first slot in the tile
    __shared__ uint32_t s_scan[MT / 64 + 1];
    const uint32_t tid = threadIdx.x;
    const uint32_t ntiles = FIRST ? (n + TILE - 1) / TILE : tile_prefix[np];
    const uint32_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    uint32_t t = blockIdx.x * per;
    const uint32_t tend = t + per < ntiles ? t + per : ntiles;
    if (per == 0 || t >= tend) return;
    uint32_t p = FIRST ? 0u : msd_find_parent(tile_prefix, np, t);
    const uint32_t mask = nchild - 1;
    const uint32_t R = nchild > (uint32_t)MT ? nchild / MT : 1u;          // digits per lane in the scan of the counts (nchild is a power of two)
    for (; t < tend; ++t) {
        uint32_t lo, m;
        if (FIRST) { lo = t * TILE; m = n - lo < (uint32_t)TILE ? n - lo : (uint32_t)TILE; }
        else {
            while (p + 1 < np && tile_prefix[p + 1] <= t) ++p;
            const uint32_t pend = parent_end[p];
            lo = (p ? parent_end[p - 1] : 0u) + (t - tile_prefix[p]) * TILE;
            m = pend - lo < (uint32_t)TILE ? pend - lo : (uint32_t)TILE;
        }
        for (uint32_t i = tid; i < nchild; i += MT) s_h[i] = 0;
        __syncthreads();
        K key[ITEMS];
        uint32_t val[ITEMS], rk[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t q = i * MT + tid;
            const uint32_t qc = q < m ? q : 0u;                     // (unconditional loads: all in flight at once)
            key[i] = keys_in[lo + qc];
            val[i] = FIRST ? lo + q : idx_in[lo + qc];
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t q = i * MT + tid;
            rk[i] = q < m ? atomicAdd(&s_h[(uint32_t)(key[i] >> shift) & mask], 1u) : 0u;
        }
        __syncthreads();
        uint32_t cnt[4], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) { const uint32_t b = tid * R + j; cnt[j] = (j < R && b < nchild) ? s_h[b] : 0u; sum += cnt[j]; }
        uint32_t total;
        uint32_t start = block_exclusive_sum<uint32_t, MT / 64>(sum, s_scan, total);
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t b = tid * R + j;
            if (j < R && b < nchild) {
                const uint32_t g = cnt[j] ? atomicAdd(&cursors[(uint64_t)p * nchild + b], cnt[j]) : 0u;
                s_h[b] = start; s_g[b] = g - start;
                start += cnt[j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t q = i * MT + tid;
            if (q < m) {
                const uint32_t pos = s_h[(uint32_t)(key[i] >> shift) & mask] + rk[i];
                s_keys[pos] = key[i]; s_idx[pos] = val[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t q = i * MT + tid;
            if (q < m) {
                const K k = s_keys[q];
                const uint32_t out = s_g[(uint32_t)(k >> shift) & mask] + q;
                keys_out[out] = k; idx_out[out] = s_idx[q];
            }
        }
        __syncthreads();
    }
}

// bound[c] = the first bucket boundary at or behind slot c G (bound[0] = 0, bound[nchunks] = n): chunk c = pairs [bound[c], bound[c + 1])
__global__ __launch_bounds__(MT) void msd_chunk_bounds_kernel(const uint32_t* __restrict__ bucket_end, uint64_t nb, uint32_t n, uint32_t nchunks,
                                                              uint32_t* __restrict__ bound) {
    const uint32_t c = blockIdx.x * MT + threadIdx.x;
    if (c > nchunks) return;
    if (c == 0) { bound[0] = 0; return; }
    if (c == nchunks) { bound[c] = n; return; }
    const uint32_t target = c * MSD_G;
    uint64_t lo = 0, hi = nb;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (bucket_end[mid] < target) lo = mid + 1; else hi = mid;
    }
    bound[c] = lo < nb ? bucket_end[lo] : n;
}

// memcmp order of two rows of C bytes, taken by a sub-wave of eight lanes (t = lane & 7, sub = lane & 56): every lane fetches sixteen bytes of
// each row -- a row of 113 bytes is one request of eight neighbouring lanes instead of fifteen dependent 8-byte loads of one --, the lowest
// lane whose piece differs decides (LPG = 4 lanes a sub-wave for rows of up to 64 bytes: twice the groups in flight).  The piece that would hang over the row's end is the sixteen bytes that END the row (the overlap is
// compared twice: if it differs, so does the piece in front, which wins).  Returns the same value (< 0, 0, > 0) in all eight lanes; the
// eight lanes must be in the same control flow.
template <int LPG = 8>
__device__ __forceinline__ int msd_row_cmp8(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint32_t C, uint32_t t, uint32_t sub) {
    for (uint32_t base = 0; base < C; base += 16 * LPG) {
        const uint32_t off = base + 16 * t;
        int c = 0;
        if (C >= 16) {
            if (off < C) {
                const uint32_t o = off + 16 <= C ? off : C - 16;
                uint64_t x0, x1, y0, y1;
                __builtin_memcpy(&x0, a + o, 8); __builtin_memcpy(&x1, a + o + 8, 8);
                __builtin_memcpy(&y0, b + o, 8); __builtin_memcpy(&y1, b + o + 8, 8);
                if (x0 != y0) c = __builtin_bswap64(x0) < __builtin_bswap64(y0) ? -1 : 1;
                else if (x1 != y1) c = __builtin_bswap64(x1) < __builtin_bswap64(y1) ? -1 : 1;
            }
        } else if (t == 0) {
            for (uint32_t i = 0; i < C && c == 0; ++i) c = a[i] == b[i] ? 0 : (a[i] < b[i] ? -1 : 1);
        }
        const uint32_t mine = (uint32_t)(__ballot(c != 0) >> sub) & ((1u << LPG) - 1u);
        if (mine) return __shfl(c, (int)(sub + (uint32_t)__ffs((int)mine) - 1u), 64);
    }
    return 0;
}

constexpr uint32_t MSD_IDX = 0x3FFFFFFFu;        // a row number (n < 2^30) under the two flag bits of an entry of the sorted chunk
constexpr uint32_t MSD_SEG_MAX = 32;             // tie groups of more rows are left open (flag 0) for the caller's refinement rounds

// A chunk in its final order by (key, row number): perm[j] = row number, heads[j] = 1 when the key at j differs from the key at j - 1 (or
// j = 0); else 2 (not ROWS: the key is the whole row, an equal key is a duplicate, final) or 0 (ROWS: the rows go on behind the key -- the
// positions where such a tie group starts are appended to tie_list for msd_ties_kernel).
template <typename K, bool ROWS>
__global__ __launch_bounds__(MT) void msd_finish_kernel(const K* __restrict__ keys, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ bound,
                                                        uint32_t nchunks, uint32_t* __restrict__ perm, uint8_t* __restrict__ heads,
                                                        uint32_t* __restrict__ flags_out /* [0] overflow */, uint32_t* __restrict__ tie_list,
                                                        uint32_t* __restrict__ tie_count) {
    constexpr int FI = MSD_CAP / MT;
    constexpr int BI = MSD_NB / MT;
    __shared__ K s_k[MSD_CAP];
    __shared__ uint32_t s_i[MSD_CAP];             // row numbers by bin, then the sorted chunk: row number | flag << 30
    __shared__ uint32_t s_bin[MSD_NB + 1];        // bins, then the list of the chunk's tie groups (their first positions)
    __shared__ K s_mn[MT / 64], s_mx[MT / 64];
    __shared__ uint32_t s_scan[MT / 64 + 1];
    __shared__ uint32_t s_nlead;
    const uint32_t tid = threadIdx.x;
    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const uint32_t lo = bound[c], hi = bound[c + 1];
        const uint32_t m = hi - lo;
        if (hi <= lo || m > MSD_CAP) {
            if (tid == 0) { if (m > MSD_CAP && hi > lo) flags_out[0] = 1u; if (ROWS) tie_count[c] = 0; }
            continue;
        }
        K k[FI];
        uint32_t v[FI], r[FI];
        K mn = ~(K)0, mx = 0;
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const uint32_t q = i * MT + tid;
            k[i] = q < m ? keys[lo + q] : (K)0;
            v[i] = q < m ? idx[lo + q] : 0u;
            if (q < m) { mn = msd_min(mn, k[i]); mx = msd_max(mx, k[i]); }
        }
        mn = wave_min(mn); mx = wave_max(mx);
#pragma unroll
        for (int j = 0; j < BI; ++j) s_bin[j * MT + tid] = 0;
        if ((tid & 63) == 0) { s_mn[tid >> 6] = mn; s_mx[tid >> 6] = mx; }
        if (tid == 0) s_nlead = 0;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < MT / 64; ++w) { mn = msd_min(mn, s_mn[w]); mx = msd_max(mx, s_mx[w]); }
        const K range = mx - mn;
        const uint32_t bits = range ? 64u - (uint32_t)__clzll((unsigned long long)range) : 0u;
        const uint32_t sh = bits > 11u ? bits - 11u : 0u;                    // (k - mn) >> sh < 2048 = MSD_NB
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const uint32_t q = i * MT + tid;
            r[i] = q < m ? atomicAdd(&s_bin[(uint32_t)((k[i] - mn) >> sh)], 1u) : 0u;
        }
        __syncthreads();
        uint32_t cnt[BI], sum = 0;
#pragma unroll
        for (int j = 0; j < BI; ++j) { cnt[j] = s_bin[tid * BI + j]; sum += cnt[j]; }
        uint32_t total;
        uint32_t start = block_exclusive_sum<uint32_t, MT / 64>(sum, s_scan, total);
#pragma unroll
        for (int j = 0; j < BI; ++j) { s_bin[tid * BI + j] = start; start += cnt[j]; }
        if (tid == 0) s_bin[MSD_NB] = m;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const uint32_t q = i * MT + tid;
            if (q < m) {
                const uint32_t pos = s_bin[(uint32_t)((k[i] - mn) >> sh)] + r[i];
                s_k[pos] = k[i]; s_i[pos] = v[i];
            }
        }
        __syncthreads();
        // the place of the pair at slot q among the pairs of its bin, by (key, row number); its flag from the largest key below it
        uint32_t place[FI], entry[FI];
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const uint32_t q = i * MT + tid;
            place[i] = 0xFFFFFFFFu; entry[i] = 0;
            if (q < m) {
                const K kk = s_k[q];
                const uint32_t ii = s_i[q];
                const uint32_t b = (uint32_t)((kk - mn) >> sh);
                const uint32_t blo = s_bin[b], bhi = s_bin[b + 1];
                uint32_t below = 0;
                bool has = false;
                K pk = 0;
#if defined(MSD_ABL) && MSD_ABL >= 2
                if (false) {
#else
                if (bhi - blo > 1) {
#endif
                    for (uint32_t j = blo; j < bhi; ++j) {
                        const K kj = s_k[j];
                        const uint32_t ij = s_i[j];
                        const bool less = kj < kk || (kj == kk && ij < ii);
                        if (less) { ++below; pk = has ? msd_max(pk, kj) : kj; has = true; }
                    }
                }
                place[i] = blo + below;
#if defined(MSD_ABL) && MSD_ABL >= 2
                place[i] = q;
#endif
                const uint32_t flag = (!has || pk != kk) ? 1u : (ROWS ? 0u : 2u);
                entry[i] = ii | (flag << 30);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < FI; ++i)
            if (place[i] != 0xFFFFFFFFu) s_i[place[i]] = entry[i];
        __syncthreads();
        if (ROWS) {
            // the chunk's tie groups (an entry that is a head in front of one that is open) go on the table's list: msd_ties_kernel sorts them
            // by whole rows with every lane of the device on row fetches (inside this kernel, between its barriers, they cost twice the rest)
            uint32_t* s_lead = s_bin;
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const uint32_t q = i * MT + tid;
                if (q + 1 < m && (s_i[q] >> 30) == 1u && (s_i[q + 1] >> 30) == 0u) s_lead[atomicAdd(&s_nlead, 1u)] = lo + q;
            }
            __syncthreads();
            const uint32_t nlead = s_nlead;                       // (at most m / 2 <= 1024: the chunk's own slots of the list)
            for (uint32_t g = tid; g < nlead; g += MT) tie_list[(uint64_t)c * (MSD_CAP / 2) + g] = s_lead[g];
            if (tid == 0) tie_count[c] = nlead;
        }
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const uint32_t q = i * MT + tid;
            if (q < m) { const uint32_t e = s_i[q]; perm[lo + q] = e & MSD_IDX; heads[lo + q] = (uint8_t)(e >> 30); }
        }
        __syncthreads();
    }
}

// The groups that tie on the key, by whole rows: nearly all are SHORT (duplicated reads, chance collisions of a 32-bit prefix).  A sub-wave of
// eight lanes per group (msd_row_cmp8), two groups at a time: stable insertion sort of the group's slice of perm -- a row moves only past
// strictly greater ones and the slice starts in row order --, then heads[j] = 1 (a new row value) or 2 (equal to the row in front, final).
// Groups of more than 32 rows keep their 0 flags for the caller's refinement rounds (*leftover is raised).  A group's first position keeps
// its flag 1 throughout: the sub-wave of the group in front may be looking for its end.
template <int LPG>
__global__ __launch_bounds__(MT) void msd_ties_kernel(const uint8_t* __restrict__ table, uint32_t C, uint32_t n, uint32_t* __restrict__ perm,
                                                      uint8_t* __restrict__ heads, const uint32_t* __restrict__ tie_list,
                                                      const uint32_t* __restrict__ tie_count, uint32_t nchunks, uint32_t* __restrict__ leftover) {
    constexpr uint32_t LM = (1u << LPG) - 1u;                 // a sub-wave's lanes in a ballot, shifted down
    constexpr uint32_t CB8 = 8;                               // chunks a workgroup takes at a time: their lists, one behind the other, keep every sub-wave busy
    const uint32_t lane = threadIdx.x & 63u, t = lane & (LPG - 1), sub = lane & ~(uint32_t)(LPG - 1);
    constexpr int U = 2;
    // (finish left the lists per chunk: no counter that every workgroup of the device would queue on)
    for (uint32_t c0 = blockIdx.x * CB8; c0 < nchunks; c0 += gridDim.x * CB8) {
    uint32_t pre[CB8 + 1];
    pre[0] = 0;
#pragma unroll
    for (uint32_t k = 0; k < CB8; ++k) pre[k + 1] = pre[k] + (c0 + k < nchunks ? tie_count[c0 + k] : 0u);
    const uint32_t ngroups = pre[CB8];
    for (uint32_t g0 = (threadIdx.x / LPG) * U; g0 < ngroups; g0 += (MT / LPG) * U) {
        uint32_t jj[U], kk[U], xa[U], xb[U];
        uint64_t a0[U], a1[U], b0[U], b1[U];
        const uint32_t off = 16 * t;
        const bool piece = C >= 16 && off < C;
        const uint32_t o = piece ? (off + 16 <= C ? off : C - 16) : 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t g = g0 + u;
            jj[u] = 0; kk[u] = 0; xa[u] = xb[u] = 0; a0[u] = a1[u] = b0[u] = b1[u] = 0;
            if (g < ngroups) {
                uint32_t kc = 0;
#pragma unroll
                for (uint32_t k = 1; k < CB8; ++k) kc += g >= pre[k] ? 1u : 0u;
                const uint32_t j = tie_list[(uint64_t)(c0 + kc) * (MSD_CAP / 2) + (g - pre[kc])];
                jj[u] = j;
                // the group's length: lane t looks at position j + 1 + t; the first position that is not open ends the group
                uint32_t len = 1;
                for (uint32_t basep = j + 1; ; basep += LPG) {
                    const uint32_t pos = basep + t;
                    const bool open = pos < n && heads[pos] == 0;
                    const uint32_t closed = (uint32_t)(__ballot(!open) >> sub) & LM;
                    if (closed) { len += (uint32_t)__ffs((int)closed) - 1u; break; }
                    len += LPG;
                    if (len > MSD_SEG_MAX) break;
                }
                kk[u] = len;
                if (len == 2) {
                    xa[u] = perm[j]; xb[u] = perm[j + 1];
                    if (piece) {
                        const uint8_t* ra = table + (uint64_t)xa[u] * C + o;
                        const uint8_t* rb = table + (uint64_t)xb[u] * C + o;
                        __builtin_memcpy(&a0[u], ra, 8); __builtin_memcpy(&a1[u], ra + 8, 8);
                        __builtin_memcpy(&b0[u], rb, 8); __builtin_memcpy(&b1[u], rb + 8, 8);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t j = jj[u], len = kk[u];
            if (len < 2) continue;
            if (len > MSD_SEG_MAX) { if (t == 0) *leftover = 1u; continue; }
            if (len == 2) {                                   // nearly every group: one comparison settles the order and the flag
                int cm;                                       // memcmp(row xb, row xa)
                if (C >= 16 && C <= 16 * LPG) {
                    int c = 0;
                    if (b0[u] != a0[u]) c = __builtin_bswap64(b0[u]) < __builtin_bswap64(a0[u]) ? -1 : 1;
                    else if (b1[u] != a1[u]) c = __builtin_bswap64(b1[u]) < __builtin_bswap64(a1[u]) ? -1 : 1;
                    const uint32_t mine = (uint32_t)(__ballot(c != 0) >> sub) & LM;
                    cm = mine ? __shfl(c, (int)(sub + (uint32_t)__ffs((int)mine) - 1u), 64) : 0;
                } else cm = msd_row_cmp8<LPG>(table + (uint64_t)xb[u] * C, table + (uint64_t)xa[u] * C, C, t, sub);
                if (t == 0) {
                    if (cm < 0) { perm[j] = xb[u]; perm[j + 1] = xa[u]; }
                    heads[j + 1] = cm == 0 ? (uint8_t)2 : (uint8_t)1;
                }
                continue;
            }
            // a longer group: its row numbers in the sub-wave's registers (lane t holds entries t, t + LPG, t + 2 LPG, ...)
            constexpr int EW = (MSD_SEG_MAX + LPG - 1) / LPG;
            uint32_t e[EW];
#pragma unroll
            for (int w = 0; w < EW; ++w) e[w] = (t + LPG * w) < len ? perm[j + t + LPG * w] : 0u;
            auto entry = [&](uint32_t i) -> uint32_t {      // entry i of the group, the same value in all lanes of the sub-wave
                const uint32_t w = i / LPG;
                uint32_t v = 0;
#pragma unroll
                for (int q = 0; q < EW; ++q) v = w == (uint32_t)q ? e[q] : v;
                return __shfl(v, (int)(sub + (i & (LPG - 1))), 64);
            };
            auto put = [&](uint32_t i, uint32_t x) {
                const uint32_t w = i / LPG;
                if (t == (i & (LPG - 1))) {
#pragma unroll
                    for (int q = 0; q < EW; ++q) e[q] = w == (uint32_t)q ? x : e[q];
                }
            };
            for (uint32_t i = 1; i < len; ++i) {
                const uint32_t x = entry(i);
                uint32_t tt = i;
                while (tt > 0) {
                    const uint32_t y = entry(tt - 1);
                    if (msd_row_cmp8<LPG>(table + (uint64_t)x * C, table + (uint64_t)y * C, C, t, sub) >= 0) break;
                    put(tt, y);
                    --tt;
                }
                put(tt, x);
            }
            uint32_t prev = entry(0);
            uint32_t dupmask = 0;                             // bit i: entry i equals entry i - 1
            for (uint32_t i = 1; i < len; ++i) {
                const uint32_t x = entry(i);
                if (msd_row_cmp8<LPG>(table + (uint64_t)x * C, table + (uint64_t)prev * C, C, t, sub) == 0) dupmask |= 1u << i;
                prev = x;
            }
#pragma unroll
            for (int w = 0; w < EW; ++w) {
                const uint32_t i = t + LPG * w;
                if (i < len) { perm[j + i] = e[w]; if (i > 0) heads[j + i] = ((dupmask >> i) & 1u) ? (uint8_t)2 : (uint8_t)1; }
            }
        }
    }
    }
}

int ceil_log2_u64(uint64_t v) { int b = 0; while ((uint64_t(1) << b) < v) ++b; return b; }

// the digits of the levels for n rows: T bits in all so that a bucket holds 32 .. 64 rows on average -- a sixteenth of what a chunk may
// hold: QUAL rows (6-bit symbols of 41 values) fill a fifth of the prefixes, and the buckets need not be full for the passes to run
// well (a level's runs are tile / 2^bits pairs whatever T is) --, at most eight bits a level (uq_sort_config overrides: tuning and tests)
int msd_plan(const uq_ctx* ctx, uint64_t n, int keybits, int* bits) {
    if (ctx->msd_levels > 0) {
        int tot = 0;
        for (int l = 0; l < ctx->msd_levels; ++l) { bits[l] = ctx->msd_bits[l]; tot += bits[l]; }
        return tot <= keybits && tot <= 24 ? ctx->msd_levels : 0;
    }
    int T = ceil_log2_u64((n + 63) / 64);
    if (T < 8) T = 8;
    if (T > 24) T = 24;
    if (T > keybits) return 0;
    const int L = (T + 7) / 8;
    bits[0] = 8;
    int rest = T - 8;
    for (int l = 1; l < L; ++l) { bits[l] = (rest + (L - 1 - l)) / (L - l); rest -= bits[l]; }
    return L;
}
}  // namespace

size_t msd_ws_bytes(uint64_t n) {
    // the histograms of the levels (the last one the largest: up to 2^24 buckets), the tile tables, the chunk table, a few words
    const uint64_t nbmax = uint64_t(1) << 24;
    uint64_t nb = 256;
    while (nb < nbmax && nb * 32 < n) nb <<= 1;
    nb <<= 1;                                            // (a plan from the environment may ask for more buckets than the default)
    if (nb > nbmax) nb = nbmax;
    return (size_t)(nb * 2 + 4096) * 4 * 2 + (size_t)(n / MSD_G + 16) * 4 * 2 + 4096;
}

template <typename K>
static int msd_round0_impl(uq_ctx* ctx, const uint8_t* table, uint32_t C, uint64_t n, uint32_t z, void* keysA, void* keysB, uint32_t* idxA, uint32_t* idxB,
                           uint32_t* perm, uint8_t* heads, void* ws, size_t ws_bytes, int* status, uint64_t* h_andor, bool* settled) {
    *status = 1;
    *settled = false;
    int bits[MSD_MAXLEVELS];
    const int keybits = 8 * (int)sizeof(K);
    const int L = msd_plan(ctx, n, keybits, bits);
    if (L == 0) return 0;
    hipStream_t s = ctx->stream;
    // ---- workspace
    uint64_t nb[MSD_MAXLEVELS];
    size_t need = 4096;
    { uint64_t b = 1; for (int l = 0; l < L; ++l) { b <<= bits[l]; nb[l] = b; need += (size_t)(b + 64) * 4 + (size_t)(b + 64 + 1) * 4; } }
    const uint32_t nchunks = (uint32_t)((n + MSD_G - 1) / MSD_G);
    need += (size_t)(nchunks + 16) * 4 * 2;
    if (need > ws_bytes) return 0;
    uint8_t* w = (uint8_t*)ws;
    unsigned long long* andor = (unsigned long long*)w;            // [0] AND  [1] OR
    uint32_t* misc = (uint32_t*)(w + 16);                          // [0] largest bucket of the last level  [1] a chunk beyond the LDS (cannot happen)  [2] a tie group left open  [3] tie groups listed
    size_t off = 4096;
    uint32_t* hist[MSD_MAXLEVELS];
    uint32_t* tilep[MSD_MAXLEVELS];
    for (int l = 0; l < L; ++l) { hist[l] = (uint32_t*)(w + off); off += (size_t)(nb[l] + 64) * 4; }
    for (int l = 1; l < L; ++l) { tilep[l] = (uint32_t*)(w + off); off += (size_t)(nb[l - 1] + 64 + 1) * 4; }
    uint32_t* bound = (uint32_t*)(w + off);
    uint32_t* tie_cnt = bound + nchunks + 8;
    UQ_CHECK_HIP(hipMemsetAsync(w, 0, 4096, s));
    UQ_CHECK_HIP(hipMemsetAsync(andor, 0xFF, 8, s));
    for (int l = 0; l < L; ++l) UQ_CHECK_HIP(hipMemsetAsync(hist[l], 0, nb[l] * 4, s));
    constexpr int TILE = MsdGeom<K>::TILE;
    const uint32_t nt1 = (uint32_t)((n + TILE - 1) / TILE);
    const uint32_t grid = UQ_NUM_CU * 6;
    K* kin = (K*)keysA; K* kout = (K*)keysB;
    uint32_t* vin = idxA; uint32_t* vout = idxB;
    int shift = keybits - bits[0];
    msd_extract_kernel<K><<<nt1 < grid ? nt1 : grid, MT, 0, s>>>(table, C, n, z, kin, (uint32_t)shift, (uint32_t)nb[0], hist[0], andor);
    UQ_LAUNCH_CHECK();
    for (int l = 0; l < L; ++l) {
        const uint32_t nchild = 1u << bits[l];
        if (l > 0) {
            shift -= bits[l];
            const uint32_t np = (uint32_t)nb[l - 1];
            msd_parent_tiles_kernel<<<(np + 1 + MT - 1) / MT, MT, 0, s>>>(hist[l - 1], np, TILE, tilep[l]);
            UQ_LAUNCH_CHECK();
            UQ_TRY(uq_scan_exclusive_u32(ctx, tilep[l], tilep[l], (uint64_t)np + 1, nullptr));
            msd_count_kernel<K><<<grid, MT, 0, s>>>(kin, hist[l - 1], tilep[l], np, (uint32_t)shift, nchild, hist[l]);
            UQ_LAUNCH_CHECK();
        }
        if (l == L - 1) {
            // the largest bucket decides whether the chunks fit the LDS; the heads' AND / OR whether z held: one wait, before the last scatter
            msd_max_kernel<<<256, MT, 0, s>>>(hist[l], nb[l], misc);
            UQ_LAUNCH_CHECK();
            UQ_TRY(uq_read_back(ctx, ctx->h_pinned, w, 32));
            UQ_CHECK_HIP(hipStreamSynchronize(s));
            h_andor[0] = ctx->h_pinned[0]; h_andor[1] = ctx->h_pinned[1];
            const uint32_t largest = (uint32_t)ctx->h_pinned[2];
            const uint64_t same = ~(h_andor[0] ^ h_andor[1]);
            uint32_t zt = 0;
            while (zt < 64 && ((same >> (63 - zt)) & 1)) ++zt;
            if (zt < z) { *status = 2; return 0; }                   // the sample agreed on more bits than the table: the caller runs it again
            if (largest > MSD_G) return 0;                           // heavy buckets: not this table
        }
        UQ_TRY(uq_scan_exclusive_u32(ctx, hist[l], hist[l], nb[l], nullptr));
        if (l == 0) msd_scatter_kernel<K, true><<<nt1 < grid ? nt1 : grid, MT, 0, s>>>(kin, nullptr, kout, vout, (uint32_t)n, nullptr, nullptr, 1, (uint32_t)shift, nchild, hist[0]);
        else msd_scatter_kernel<K, false><<<grid, MT, 0, s>>>(kin, vin, kout, vout, (uint32_t)n, hist[l - 1], tilep[l], (uint32_t)nb[l - 1], (uint32_t)shift, nchild, hist[l]);
        UQ_LAUNCH_CHECK();
        K* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    msd_chunk_bounds_kernel<<<(nchunks + 1 + MT - 1) / MT, MT, 0, s>>>(hist[L - 1], nb[L - 1], (uint32_t)n, nchunks, bound);
    UQ_LAUNCH_CHECK();
    const bool rows = C > 8;                                                 // the rows go on behind the head (32-bit keys are for such tables only)
    const uint32_t fgrid = nchunks < UQ_NUM_CU * 16 ? nchunks : UQ_NUM_CU * 16;
    uint32_t* tie_list = vout;                                               // (the ping-pong buffer the last scatter read from: n / 2 groups at most)
    if (rows) {
        msd_finish_kernel<K, true><<<fgrid, MT, 0, s>>>(kin, vin, bound, nchunks, perm, heads, misc + 1, tie_list, tie_cnt);
        UQ_LAUNCH_CHECK();
        const uint32_t tgrid = (nchunks + 7) / 8 < UQ_NUM_CU * 8 ? (nchunks + 7) / 8 : UQ_NUM_CU * 8;
        if (C <= 64) msd_ties_kernel<4><<<tgrid, MT, 0, s>>>(table, C, (uint32_t)n, perm, heads, tie_list, tie_cnt, nchunks, misc + 2);
        else msd_ties_kernel<8><<<tgrid, MT, 0, s>>>(table, C, (uint32_t)n, perm, heads, tie_list, tie_cnt, nchunks, misc + 2);
    } else msd_finish_kernel<K, false><<<fgrid, MT, 0, s>>>(kin, vin, bound, nchunks, perm, heads, misc + 1, nullptr, nullptr);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned, misc, 16));
    UQ_CHECK_HIP(hipStreamSynchronize(s));
    const uint32_t* hm = (const uint32_t*)ctx->h_pinned;
    UQ_REQUIRE(hm[1] == 0, "row sort: a chunk beyond the finishing workgroup's LDS (the bucket bound was checked: a defect)");
    *settled = hm[2] == 0;
    *status = 0;
    return 0;
}

int msd_round0(uq_ctx* ctx, const uint8_t* table, uint32_t C, uint64_t n, uint32_t z, int key64, void* keysA, void* keysB, uint32_t* idxA, uint32_t* idxB,
               uint32_t* perm, uint8_t* heads, void* ws, size_t ws_bytes, int* status, uint64_t* h_andor, bool* settled) {
    if (key64) return msd_round0_impl<uint64_t>(ctx, table, C, n, z, keysA, keysB, idxA, idxB, perm, heads, ws, ws_bytes, status, h_andor, settled);
    return msd_round0_impl<uint32_t>(ctx, table, C, n, z, keysA, keysB, idxA, idxB, perm, heads, ws, ws_bytes, status, h_andor, settled);
}
