// sort.hip -- row sort and unique/inverse on the device.
// Replaces numpy.argsort(table.view('V<C>'), axis=0) (uq.py:773-775), numpy.unique(rows,
// return_inverse=True) (uq.py:784-789, 828-830) and numpy.argsort(key) (uq.py:796, 833).
//
// Keys are whole rows of C bytes (38 .. 227 B and more) compared like memcmp.  An LSD radix sort over
// every byte would need C passes; instead the sort is MSD by 8-byte chunks with LSD radix inside:
//   round 0   sort all rows by chunk 0 (their first 8 bytes as a big-endian u64), stable -- or, for wide rows whose chunk 0
//             is rich in its leading bits, by the 32 bits behind chunk 0's constant leading bits as u32 keys (four passes of
//             20 B a pair instead of eight of 32 B); the groups that tie on those -- chance collisions, duplicated reads -- are
//             short: a lane per group sorts its slice of the order by whole rows (segment_sort_kernel) and flags duplicates as
//             final; only groups of more than 32 rows go on to round 1, which sorts (tie segment, rest of chunk 0) as one key;
//   round k   only rows that still tie with a neighbour ("active") are touched: they are compacted,
//             sorted by chunk k, then stably regrouped by their tie-segment id, and written back into
//             the slots their segment occupies.  Distinct rows drop out as soon as a chunk separates
//             them, so random reads finish after round 0 and only true duplicates visit every chunk.
// Ties are broken by input order (stable) -- the canonical order of SURVEY.md Q17.
// The per-row "first of its group" flags that fall out of the rounds give unique / inverse for free:
// key = inclusive scan of the flags, and the stable sort order IS argsort(key, stable) (uq.py:796).
#include <algorithm>
#include <stdlib.h>
#include "radix.h"

namespace {
constexpr int ST = 256;

__device__ __forceinline__ uint64_t load_chunk_be(const uint8_t* row, uint32_t C, uint32_t k) {
    const uint32_t o = 8 * k;
    uint64_t v;
    if (o + 8 <= C) {
        __builtin_memcpy(&v, row + o, 8);
        return __builtin_bswap64(v);
    }
    v = 0;
    for (uint32_t i = 0; i < 8; ++i) v = (v << 8) | (o + i < C ? row[o + i] : 0);
    return v;
}

// chunk 0 of every row (the 64-bit round 0) and the rows' numbers
__global__ __launch_bounds__(ST) void extract_all_kernel(const uint8_t* __restrict__ table, uint64_t n, uint32_t C, uint64_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals) {
    const uint64_t i = (uint64_t)blockIdx.x * ST + threadIdx.x;
    if (i < n) { keys[i] = load_chunk_be(table + i * C, C, 0); vals[i] = (uint32_t)i; }
}
// chunk 0 of `samples` rows spread evenly over the table, straight into the pinned staging: the host guesses from them how many leading bits
// every row agrees on and whether the rows crowd on few prefixes
__global__ __launch_bounds__(ST) void sample_chunks_kernel(const uint8_t* __restrict__ table, uint64_t n, uint32_t C, uint64_t samples, uint64_t* __restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * ST + threadIdx.x;
    if (j < samples) out[j] = load_chunk_be(table + (samples > 1 ? j * (n - 1) / (samples - 1) : 0) * C, C, 0);
}
__global__ void iota_kernel(uint32_t* __restrict__ v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * ST + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}
__global__ void heads_first_kernel(const uint64_t* __restrict__ keys, uint64_t n, uint8_t* __restrict__ heads) {
    uint64_t j = (uint64_t)blockIdx.x * ST + threadIdx.x;
    if (j >= n) return;
    heads[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1 : 0;
}

// Do the rows that still tie with their predecessor (heads[j] == 0: equal through byte `from` - 1) differ anywhere in
// bytes [from, C)?  Ties in real tables are overwhelmingly whole-row duplicates: when no pair differs the order is
// final (stable = file order inside a group) and the remaining rounds are skipped.
__global__ void tail_differs_kernel(const uint8_t* __restrict__ table, uint32_t C, uint32_t from, const uint32_t* __restrict__ perm,
                                    const uint8_t* __restrict__ heads, uint64_t n, uint32_t* __restrict__ flag) {
    uint64_t j = (uint64_t)blockIdx.x * ST + threadIdx.x;
    if (j == 0 || j >= n || heads[j]) return;
    const uint8_t* a = table + (uint64_t)perm[j] * C;
    const uint8_t* b = table + (uint64_t)perm[j - 1] * C;
    bool diff = false;
    uint32_t i = from;
    for (; i + 8 <= C; i += 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, a + i, 8); __builtin_memcpy(&y, b + i, 8);
        diff |= x != y;
    }
    for (; i < C; ++i) diff |= a[i] != b[i];
    if (diff) *flag = 1u;
}

// ---- after the 32-bit round 0: the groups of rows that tie on the prefix are nearly all SHORT -- pairs that collide by chance
// (n^2 / 2^33 of them) and duplicated reads -- so the lane at a group's first position sorts it on the spot: insertion sort of
// its slice of `perm` by whole rows (memcmp order, stable: a row moves only past strictly greater ones), then the head flags
// of the slice: 1 = a new row value, HEAD_DUP = equal to the row before AND final (the refinement rounds below leave the group
// alone; the consumers count flag 1 only).  Groups beyond SEG_MAX rows keep their 0 flags and go through the radix rounds.
constexpr uint32_t SEG_MAX = 32;
constexpr uint8_t HEAD_DUP = 2;
// memcmp order of two rows.  No early exit: the loads of the next words do not wait for the comparison of these (a row is one or
// two cache lines, fetched whole anyway), four words of each row are in flight at a time.
__device__ __forceinline__ int row_cmp(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint32_t C) {
    int res = 0;
    uint32_t i = 0;
#pragma unroll 4
    for (; i + 8 <= C; i += 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, a + i, 8); __builtin_memcpy(&y, b + i, 8);
        if (res == 0 && x != y) res = __builtin_bswap64(x) < __builtin_bswap64(y) ? -1 : 1;
    }
    if (i < C) {                                       // the last C % 8 bytes: the eight bytes that END the row, the overlap compared twice
        uint64_t x, y;
        if (C >= 8) { __builtin_memcpy(&x, a + C - 8, 8); __builtin_memcpy(&y, b + C - 8, 8); }
        else { x = y = 0; for (uint32_t t = 0; t < C; ++t) { x |= (uint64_t)a[t] << (8 * t); y |= (uint64_t)b[t] << (8 * t); } }
        if (res == 0 && x != y) res = __builtin_bswap64(x) < __builtin_bswap64(y) ? -1 : 1;
    }
    return res;
}
// The groups' first positions are compacted first (segment_starts_kernel, below the counting helpers): a lane per GROUP, every
// lane of a wave busy -- a lane per POSITION left 60 of 64 lanes idle while the others chased pointers: 2.4 ms instead of 0.2.
// `heads` is read, `flags` (a copy of it) is written: a lane must not mistake a flag its neighbour has just set for a group's first row.
// *leftover is raised when a group was too long and keeps its 0 flags.
__global__ __launch_bounds__(ST) void segment_sort_kernel(const uint8_t* __restrict__ table, uint32_t C, uint32_t* __restrict__ perm,
                                                          const uint8_t* __restrict__ heads, uint8_t* __restrict__ flags, uint64_t n,
                                                          const uint32_t* __restrict__ seg_start, const unsigned long long* __restrict__ totals,
                                                          uint32_t* __restrict__ leftover) {
    const uint64_t nseg = totals[0] >> 32;                        // {rows in open groups, open groups} as scanned
    for (uint64_t sidx = (uint64_t)blockIdx.x * ST + threadIdx.x; sidx < nseg; sidx += (uint64_t)gridDim.x * ST) {
        const uint64_t j = seg_start[sidx];
        uint32_t k = 1;
        while (j + k < n && heads[j + k] == 0 && k <= SEG_MAX) ++k;
        if (k > SEG_MAX) { *leftover = 1u; continue; }
        uint32_t* p = perm + j;
        if (k == 2) {                                     // nearly every group: one comparison settles the order and the flag
            const uint32_t x0 = p[0], x1 = p[1];
            const int c = row_cmp(table + (uint64_t)x1 * C, table + (uint64_t)x0 * C, C);
            if (c < 0) { p[0] = x1; p[1] = x0; }
            flags[j + 1] = c == 0 ? HEAD_DUP : (uint8_t)1;
            continue;
        }
        for (uint32_t i = 1; i < k; ++i) {
            const uint32_t x = p[i];
            const uint8_t* rx = table + (uint64_t)x * C;
            uint32_t t = i;
            while (t > 0 && row_cmp(rx, table + (uint64_t)p[t - 1] * C, C) < 0) { p[t] = p[t - 1]; --t; }
            p[t] = x;
        }
        for (uint32_t i = 1; i < k; ++i)
            flags[j + i] = row_cmp(table + (uint64_t)p[i] * C, table + (uint64_t)p[i - 1] * C, C) == 0 ? HEAD_DUP : (uint8_t)1;
    }
}

// ---- compaction of the rows that still tie, without arrays of n flags and their scans: per block of CB sorted positions the number of rows
// that still tie and of tie segments that start there (counts[2 b], [2 b + 1]); after a scan of those few counters the
// second kernel recomputes the flags of its block, ranks them inside the block and writes the compacted lists.
constexpr int CB = 1024;                     // positions per workgroup of 256: four per lane
// flags: 1 = first row of a group, 0 = ties with the row before on everything looked at so far, HEAD_DUP = equal to the row before, final
__device__ __forceinline__ void active_of(const uint8_t* __restrict__ heads, uint64_t n, uint64_t j, bool& act, bool& seg) {
    act = seg = false;
    if (j >= n) return;
    const uint8_t f = heads[j];
    const bool next_open = (j + 1 < n) && heads[j + 1] == 0;
    seg = f == 1 && next_open;                       // first row of a group that is still open
    act = seg || f == 0;
}
__global__ __launch_bounds__(ST) void active_count_kernel(const uint8_t* __restrict__ heads, uint64_t n, uint32_t* __restrict__ counts) {
    __shared__ uint32_t sa[ST / 64], sh[ST / 64];
    const uint64_t j0 = (uint64_t)blockIdx.x * CB + (uint64_t)threadIdx.x * 4;
    uint32_t a = 0, h = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { bool act, seg; active_of(heads, n, j0 + i, act, seg); a += act; h += seg; }
    a = wave_sum(a); h = wave_sum(h);
    if (lane_id() == 0) { sa[threadIdx.x >> 6] = a; sh[threadIdx.x >> 6] = h; }
    __syncthreads();
    if (threadIdx.x == 0) { counts[2 * (uint64_t)blockIdx.x] = sa[0] + sa[1] + sa[2] + sa[3]; counts[2 * (uint64_t)blockIdx.x + 1] = sh[0] + sh[1] + sh[2] + sh[3]; }
}
// after the 32-bit round 0, in one pass over the sorted keys: the group-head flags (two copies: segment_sort_kernel reads one and
// writes the other) and the per-block counters of active_count_kernel
template <typename K>
__global__ __launch_bounds__(ST) void heads32_count_kernel(const K* __restrict__ keys, uint64_t n, uint8_t* __restrict__ heads,
                                                           uint8_t* __restrict__ flags, uint32_t* __restrict__ counts) {
    __shared__ uint32_t sa[ST / 64], sh[ST / 64];
    const uint64_t j0 = (uint64_t)blockIdx.x * CB + (uint64_t)threadIdx.x * 4;
    K k[6];                                          // keys j0 - 1 .. j0 + 4
#pragma unroll
    for (int i = 0; i < 6; ++i) { const uint64_t j = j0 + i; k[i] = (j >= 1 && j - 1 < n) ? keys[j - 1] : (K)0; }
    uint32_t a = 0, h = 0, packed = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint64_t j = j0 + i;
        if (j >= n) break;
        const bool hd = j == 0 || k[i + 1] != k[i];
        const bool next_open = j + 1 < n && k[i + 2] == k[i + 1];
        const bool seg = hd && next_open;
        a += (seg || !hd) ? 1u : 0u; h += seg ? 1u : 0u;
        packed |= (hd ? 1u : 0u) << (8 * i);
    }
    if (j0 + 4 <= n) { *(uint32_t*)(heads + j0) = packed; *(uint32_t*)(flags + j0) = packed; }      // (j0 is a multiple of 4, the arrays 16-byte aligned)
    else for (int i = 0; i < 4 && j0 + i < n; ++i) { heads[j0 + i] = (uint8_t)(packed >> (8 * i)); flags[j0 + i] = (uint8_t)(packed >> (8 * i)); }
    a = wave_sum(a); h = wave_sum(h);
    if (lane_id() == 0) { sa[threadIdx.x >> 6] = a; sh[threadIdx.x >> 6] = h; }
    __syncthreads();
    if (threadIdx.x == 0) { counts[2 * (uint64_t)blockIdx.x] = sa[0] + sa[1] + sa[2] + sa[3]; counts[2 * (uint64_t)blockIdx.x + 1] = sh[0] + sh[1] + sh[2] + sh[3]; }
}

// first positions of the open groups, in order (offs as below)
__global__ __launch_bounds__(ST) void segment_starts_kernel(const uint8_t* __restrict__ heads, const unsigned long long* __restrict__ offs, uint64_t n,
                                                            uint32_t* __restrict__ seg_start) {
    __shared__ uint32_t lds[ST / 64 + 1];
    const uint64_t j0 = (uint64_t)blockIdx.x * CB + (uint64_t)threadIdx.x * 4;
    bool seg[4];
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { bool act; active_of(heads, n, j0 + i, act, seg[i]); mine += seg[i]; }
    uint32_t total;
    uint32_t ex = block_exclusive_sum<uint32_t, ST / 64>(mine, lds, total) + (uint32_t)(offs[blockIdx.x] >> 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) if (seg[i]) seg_start[ex++] = (uint32_t)(j0 + i);
}
// offs = exclusive scan of counts (pairs interleaved: scanned as u64 = {actives, segments}, both below 2^32)
__global__ __launch_bounds__(ST) void compact_active2_kernel(const uint8_t* __restrict__ heads, const unsigned long long* __restrict__ offs,
                                                             const uint32_t* __restrict__ perm, uint64_t n, const uint8_t* __restrict__ table,
                                                             uint32_t C, uint32_t k, uint32_t fold_z, uint32_t* __restrict__ pos, uint32_t* __restrict__ aval,
                                                             uint32_t* __restrict__ sid, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    __shared__ unsigned long long lds[ST / 64 + 1];
    const uint64_t j0 = (uint64_t)blockIdx.x * CB + (uint64_t)threadIdx.x * 4;
    bool act[4], seg[4];
    unsigned long long mine = 0;             // low word: actives, high word: segment heads
#pragma unroll
    for (int i = 0; i < 4; ++i) { active_of(heads, n, j0 + i, act[i], seg[i]); mine += (unsigned long long)act[i] + ((unsigned long long)seg[i] << 32); }
    unsigned long long total;
    unsigned long long ex = block_exclusive_sum<unsigned long long, ST / 64>(mine, lds, total) + offs[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (act[i]) {
            const uint32_t u = (uint32_t)ex;
            const uint32_t segs_before = (uint32_t)(ex >> 32);
            const uint32_t row = perm[j0 + i];
            pos[u] = (uint32_t)(j0 + i);
            aval[u] = row;
            const uint32_t sg = segs_before + (seg[i] ? 1u : 0u) - 1u;
            sid[u] = sg;
            const uint64_t chunk = load_chunk_be(table + (uint64_t)row * C, C, k);
            // fold_z (after a 32-bit round 0): the rows of a segment agree on the 32 bits behind chunk 0's z leading ones, so what
            // is left of the chunk fits the low word and (segment, rest) is ONE sort key -- no separate sort by chunk value
            keys[u] = fold_z == 0xFFFFFFFFu ? chunk : ((uint64_t)sg << 32) | (uint32_t)(chunk << fold_z);
            vals[u] = u;
        }
        ex += (unsigned long long)act[i] + ((unsigned long long)seg[i] << 32);
    }
}

// d[t] = chunk value differs from the previous active element (in chunk order)
__global__ void diff_flags_kernel(const uint64_t* __restrict__ keys, uint64_t m, uint32_t* __restrict__ d) {
    uint64_t t = (uint64_t)blockIdx.x * ST + threadIdx.x;
    if (t >= m) return;
    d[t] = (t == 0 || keys[t] != keys[t - 1]) ? 1u : 0u;
}

// keys2[t] = sid[V1[t]] << 32 | rank1[t]   (rank1 = exclusive scan of d + d: monotone label of the chunk value)
__global__ void compose_kernel(const uint32_t* __restrict__ v1, const uint32_t* __restrict__ sid, const uint32_t* __restrict__ d,
                               const uint32_t* __restrict__ dscan, uint64_t m, uint64_t* __restrict__ keys2) {
    uint64_t t = (uint64_t)blockIdx.x * ST + threadIdx.x;
    if (t >= m) return;
    keys2[t] = ((uint64_t)sid[v1[t]] << 32) | (uint64_t)(dscan[t] + d[t]);
}

// write the refined order back into the slots of the active segments and refresh their head flags
__global__ void writeback_kernel(const uint64_t* __restrict__ keys2, const uint32_t* __restrict__ v2, const uint32_t* __restrict__ pos,
                                 const uint32_t* __restrict__ aval, uint64_t m, uint32_t* __restrict__ perm, uint8_t* __restrict__ heads) {
    uint64_t t = (uint64_t)blockIdx.x * ST + threadIdx.x;
    if (t >= m) return;
    const uint32_t j = pos[t];
    perm[j] = aval[v2[t]];
    heads[j] = (t == 0 || keys2[t] != keys2[t - 1]) ? 1 : 0;
}

inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + ST - 1) / ST); }

int bits_for(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

struct Core {
    uint8_t* heads;
    void* extra;
};

// Sorts the rows; leaves the stable order in d_perm and the group-head flags in `heads`.
int sort_rows_core(uq_ctx* ctx, const uint8_t* table, uint64_t n, uint32_t C, uint32_t* d_perm, size_t extra_bytes, Core* out) {
    UQ_REQUIRE(n < (uint64_t(1) << 32), "row sort: more than 2^32-1 rows per GPU");
    ScratchPlan plan;
    const size_t o_keysA = plan.add(n * 8), o_keysB = plan.add(n * 8);
    const size_t o_valsA = plan.add(n * 4), o_valsB = plan.add(n * 4);
    const size_t o_heads = plan.add(n + 16), o_heads2 = plan.add(n + 16);
    const size_t o_a = plan.add(n * 4), o_h = plan.add(n * 4);
    const size_t o_apos = plan.add(n * 4);
    const size_t o_pos = plan.add(n * 4), o_aval = plan.add(n * 4), o_sid = plan.add(n * 4);
    const size_t o_tot = plan.add(64);
    const size_t o_rws = plan.add(radix_ws_bytes(n));
    const size_t o_msd = plan.add(msd_ws_bytes(n));
    const size_t o_extra = plan.add(extra_bytes + 256);
    void* base;
    UQ_TRY(uq_scratch(ctx, plan.off, &base));
    uint8_t* b = (uint8_t*)base;
    uint64_t* keysA = (uint64_t*)(b + o_keysA); uint64_t* keysB = (uint64_t*)(b + o_keysB);
    uint32_t* valsA = (uint32_t*)(b + o_valsA); uint32_t* valsB = (uint32_t*)(b + o_valsB);
    uint8_t* heads = b + o_heads;
    uint32_t* fa = (uint32_t*)(b + o_a); uint32_t* fh = (uint32_t*)(b + o_h);
    uint32_t* apos = (uint32_t*)(b + o_apos);
    uint32_t* pos = (uint32_t*)(b + o_pos); uint32_t* aval = (uint32_t*)(b + o_aval); uint32_t* sid = (uint32_t*)(b + o_sid);
    uint64_t* tot = (uint64_t*)(b + o_tot);
    void* rws = b + o_rws;
    out->heads = heads;
    out->extra = b + o_extra;
    if (n == 0) return 0;
    hipStream_t s = ctx->stream;

    // ---- round 0: all rows by (a prefix of) chunk 0
    bool mode32 = C > 8 && n >= (1u << 16);
    // A 64-bit LSD sort moves 32 B per pair and pass, eight passes.  Rows wider than a chunk go to refinement rounds anyway
    // when they tie, so round 0 may as well sort on FEWER bits: the 32 bits of chunk 0 behind its constant leading bits
    // (a 2-bit DNA row of 150 bases starts with four zero bits) as u32 keys -- 20 B per pair and pass, four passes -- and
    // leave the few rows that collide on them (n^2 / 2^33 pairs for random reads) to the first refinement round together
    // with the true duplicates.  Tables whose rows crowd on few prefixes (the digit census says so) keep the 64-bit sort.
    // The number z of constant leading bits is guessed from 4096 rows spread over the table; the pass that makes the keys checks
    // it on every row (AND / OR over all chunks) and runs again in the rare case that the sample agreed on more bits than the table.
    auto leading_same = [](uint64_t a, uint64_t o) { const uint64_t same = ~(a ^ o); uint32_t z = 0; while (z < 64 && ((same >> (63 - z)) & 1)) ++z; return z; };
    uint32_t z = 0;
    auto collisions = [](const uint32_t* hist, double total, double rows) {      // expected share of rows that tie on the 32-bit prefix if its bytes were independent: n * prod_p sum_d (h[p][d] / n)^2
        double coll = rows;
        for (int p = 0; p < 4; ++p) {
            double c2 = 0;
            for (int d = 0; d < 256; ++d) { const double q = (double)hist[p * 256 + d] / total; c2 += q * q; }
            coll *= c2;
        }
        return coll;
    };
    // the short groups that tie on round 0's key (chance collisions, duplicated reads): a lane per group sorts its slice of the order by whole
    // rows and sets the final flags (segment_sort_kernel).  bcnt0 = active_count_kernel's / heads32_count_kernel's counters of `heads`.
    // Returns through *settled whether every group was short enough (then the order is final).
    uint8_t* flags = b + o_heads2;
    const uint64_t ncb0 = (n + CB - 1) / CB;
    uint32_t* bcnt0 = apos;
    auto settle_short_groups = [&](bool* settled) -> int {
        UQ_TRY(uq_scan_exclusive_u64(ctx, (const uint64_t*)bcnt0, (uint64_t*)bcnt0, ncb0, tot));
        segment_starts_kernel<<<(uint32_t)ncb0, ST, 0, s>>>(heads, (const unsigned long long*)bcnt0, n, pos);
        UQ_LAUNCH_CHECK();
        UQ_CHECK_HIP(hipMemsetAsync(tot + 3, 0, 8, s));
        segment_sort_kernel<<<UQ_NUM_CU * 8, ST, 0, s>>>(table, C, d_perm, heads, flags, n, pos, (const unsigned long long*)tot, (uint32_t*)(tot + 3));
        UQ_LAUNCH_CHECK();
        heads = flags; out->heads = flags;
        UQ_TRY(uq_read_back(ctx, ctx->h_pinned, tot + 3, 8));
        UQ_CHECK_HIP(hipStreamSynchronize(s));
        *settled = (uint32_t)ctx->h_pinned[0] == 0;
        return 0;
    };
    const uint64_t msd_min = ctx->msd_min_rows > 0 ? (uint64_t)ctx->msd_min_rows : (uint64_t(1) << 18);
    bool sampled = false, msd_ok = ctx->msd_min_rows >= 0 && n >= msd_min && n >= 2 && n < (uint64_t(1) << 30);
    if (mode32 || msd_ok) {
        const uint64_t samples = 4096;                                           // (spread over the table; fewer rows than that: some twice)
        sample_chunks_kernel<<<(uint32_t)((samples + ST - 1) / ST), ST, 0, s>>>(table, n, C, samples, (uint64_t*)ctx->d_pinned);
        UQ_LAUNCH_CHECK();
        UQ_CHECK_HIP(hipStreamSynchronize(s));
        sampled = true;
        uint64_t a = ~0ull, o = 0;
        for (uint64_t j = 0; j < samples; ++j) { a &= ctx->h_pinned[j]; o |= ctx->h_pinned[j]; }
        z = leading_same(a, o);
        mode32 = mode32 && z <= 32;
        if (mode32) {
            // crowded tables (a QUAL table of 200 M rows on its first 32 bits) go to the 64-bit round 0 without the pass below: the sample's own digit
            // census overestimates the collisions of spread-out keys by a quarter (a sample of 4096 adds 1 / 4096 to each sum of squares), far from the threshold
            static thread_local uint32_t s_hist[4 * 256];
            memset(s_hist, 0, sizeof(s_hist));
            for (uint64_t j = 0; j < samples; ++j) {
                const uint32_t k = (uint32_t)((ctx->h_pinned[j] << z) >> 32);
                ++s_hist[k & 255u]; ++s_hist[256 + ((k >> 8) & 255u)]; ++s_hist[512 + ((k >> 16) & 255u)]; ++s_hist[768 + (k >> 24)];
            }
            if (collisions(s_hist, (double)samples, (double)n) > 0.6) mode32 = false;
        }
        if (msd_ok) {
            // the MSD partition wants heads that spread: a value of the top 20 bits behind z that takes eight of the 4096 sampled rows holds
            // n / 512 rows -- heavier than a finishing chunk whatever the levels (QNAME columns, a read copied a million times)
            msd_ok = z < 40;
            if (msd_ok) {
                static thread_local uint32_t top[4096];
                for (uint64_t j = 0; j < samples; ++j) top[j] = (uint32_t)((ctx->h_pinned[j] << z) >> 44);
                std::sort(top, top + samples);
                uint32_t run = 1, worst = 1;
                for (uint64_t j = 1; j < samples; ++j) { run = top[j] == top[j - 1] ? run + 1 : 1; worst = run > worst ? run : worst; }
                if (worst >= 8 && (double)worst * (double)n / (double)samples > 1024.0) msd_ok = false;
            }
        }
    }
    int alt = 0;
    bool round0_done = false;
    if (msd_ok) {
        // MSD partition + LDS finish (msd.hip): keys of 32 bits where the LSD round 0 would take them (mode32), else the whole head
        int status = 1;
        bool settled = false;
        for (int attempt = 0; attempt < 3; ++attempt) {
            uint64_t h_andor[2];
            UQ_TRY(msd_round0(ctx, table, C, n, z, mode32 ? 0 : 1, keysA, keysB, valsA, valsB, d_perm, heads, b + o_msd, msd_ws_bytes(n), &status, h_andor, &settled));
            if (status != 2) break;
            z = leading_same(h_andor[0], h_andor[1]);
        }
        if (status == 0) {
            round0_done = true;
            ++ctx->n_msd_rounds;
            if (settled) return 0;                                               // every tie group was short: the order and the flags are final
        } else if (status == 2) mode32 = false;                                  // (a table that keeps contradicting its sample: the plain 64-bit passes)
    }
    if (!round0_done) ++ctx->n_lsd_rounds;
    if (mode32 && !round0_done) {
        uint32_t* k32a = (uint32_t*)keysB;                                      // keysB's n * 8 bytes hold both u32 key buffers
        uint32_t* k32b = k32a + n;
        static thread_local uint32_t h_hist[4 * 256];
        // keys, values (d_perm is the sort's value buffer: an even number of passes ends there), digit census and the first pass's
        // per-tile counts in one pass over the rows' first eight bytes
        for (;;) {
            uint64_t h_andor[2];
            UQ_TRY(radix_rows_prefix_census32(ctx, table, C, n, z, k32a, d_perm, rws, h_hist, h_andor));
            const uint32_t zt = leading_same(h_andor[0], h_andor[1]);
            if (zt >= z) break;
            z = zt;                                                             // the sample agreed on more bits than the table does: once more
        }
        const double coll = collisions(h_hist, (double)n, (double)n);
        if (coll > 0.3) mode32 = false;
        else {
            UQ_TRY(radix_sort_pairs32(ctx, k32a, d_perm, k32b, valsB, n, 0, 32, rws, &alt, h_hist, 1));
            if (alt) UQ_CHECK_HIP(hipMemcpyAsync(d_perm, valsB, n * 4, hipMemcpyDeviceToDevice, s));
            // the groups that tie on the prefix are short (colliding pairs, duplicates): a lane sorts each by whole rows
            heads32_count_kernel<uint32_t><<<(uint32_t)ncb0, ST, 0, s>>>(alt ? k32b : k32a, n, heads, flags, bcnt0);
            UQ_LAUNCH_CHECK();
            bool settled;
            UQ_TRY(settle_short_groups(&settled));
            if (settled) return 0;        // every group settled: the order is final
        }
    }
    if (!mode32 && !round0_done) {
        extract_all_kernel<<<blocks_for(n), ST, 0, s>>>(table, n, C, keysA, valsA);
        UQ_LAUNCH_CHECK();
        UQ_TRY(radix_sort_pairs(ctx, keysA, valsA, keysB, valsB, n, 0, 64, rws, &alt));
        const uint64_t* K = alt ? keysB : keysA;
        const uint32_t* V = alt ? valsB : valsA;
        UQ_CHECK_HIP(hipMemcpyAsync(d_perm, V, n * 4, hipMemcpyDeviceToDevice, s));
        if (C <= 8) {
            heads_first_kernel<<<blocks_for(n), ST, 0, s>>>(K, n, heads);      // the chunk IS the row: ties are duplicates, nothing to refine
            UQ_LAUNCH_CHECK();
        } else {
            // rows that tie on all 64 bits of chunk 0 are nearly all whole-row duplicates in SHORT groups (a duplicated read, the odd
            // pair that collides by chance): as after the 32-bit round 0, a lane sorts each group by whole rows on the spot and sets
            // its final flags -- one pair that differs somewhere behind byte 8 used to send every tied row (a tenth of a 200 M row
            // QUAL table) through a radix refinement round of fifteen more passes
            heads32_count_kernel<uint64_t><<<(uint32_t)ncb0, ST, 0, s>>>(K, n, heads, flags, bcnt0);
            UQ_LAUNCH_CHECK();
            bool settled;
            UQ_TRY(settle_short_groups(&settled));
            if (settled) return 0;        // every group settled: the order is final
        }
    }
    (void)sampled;

    // ---- refinement rounds (after a 32-bit round 0 the first one looks at the whole of chunk 0 again)
    const uint32_t nchunks = (C + 7) / 8;
    for (uint32_t k = mode32 ? 0 : 1; k < nchunks; ++k) {
        // exact duplicates need no further rounds: stop as soon as no tying pair differs in what is left of the rows.
        // That test, the flags of the rows still tying and their two scans are queued together: ONE synchronisation per round.
        if (mode32 && k == 0) UQ_CHECK_HIP(hipMemsetAsync(tot + 2, 0xFF, 8, s));        // rows that collide on 32 bits: they do differ, no need to look
        else {
            UQ_CHECK_HIP(hipMemsetAsync(tot + 2, 0, 8, s));
            tail_differs_kernel<<<blocks_for(n), ST, 0, s>>>(table, C, 8 * k, d_perm, heads, n, (uint32_t*)(tot + 2));
            UQ_LAUNCH_CHECK();
        }
        const uint64_t ncb = (n + CB - 1) / CB;
        uint32_t* bcnt = apos;                                                  // 2 counters per block of CB positions, then their scan
        active_count_kernel<<<(uint32_t)ncb, ST, 0, s>>>(heads, n, bcnt);
        UQ_LAUNCH_CHECK();
        UQ_TRY(uq_scan_exclusive_u64(ctx, (const uint64_t*)bcnt, (uint64_t*)bcnt, ncb, tot));      // {actives, segments} packed in one u64
        UQ_TRY(uq_read_back(ctx, ctx->h_pinned, tot, 24));
        UQ_CHECK_HIP(hipStreamSynchronize(s));
        const uint64_t m = ctx->h_pinned[0] & 0xFFFFFFFFull, nseg = ctx->h_pinned[0] >> 32;
        if ((uint32_t)ctx->h_pinned[2] == 0 || m == 0) break;
        const bool fold = mode32 && k == 0;
        compact_active2_kernel<<<(uint32_t)ncb, ST, 0, s>>>(heads, (const unsigned long long*)bcnt, d_perm, n, table, C, k, fold ? z : 0xFFFFFFFFu, pos, aval, sid,
                                                           keysA, valsA);
        UQ_LAUNCH_CHECK();
        const int sbits0 = bits_for(nseg > 0 ? nseg - 1 : 0);
        if (fold) {
            int alt2 = 0;
            UQ_TRY(radix_sort_pairs(ctx, keysA, valsA, keysB, valsB, m, 0, 32 + (sbits0 ? sbits0 : 1), rws, &alt2));
            writeback_kernel<<<blocks_for(m), ST, 0, s>>>(alt2 ? keysB : keysA, alt2 ? valsB : valsA, pos, aval, m, d_perm, heads);
            UQ_LAUNCH_CHECK();
            continue;
        }
        // sort #1: active rows by the value of chunk k
        UQ_TRY(radix_sort_pairs(ctx, keysA, valsA, keysB, valsB, m, 0, 64, rws, &alt));
        const uint64_t* K1 = alt ? keysB : keysA;
        uint32_t* V1 = alt ? valsB : valsA;
        uint64_t* Kfree = alt ? keysA : keysB;       // the key buffer not holding K1
        uint32_t* Vfree = alt ? valsA : valsB;
        diff_flags_kernel<<<blocks_for(m), ST, 0, s>>>(K1, m, fa);
        UQ_LAUNCH_CHECK();
        UQ_TRY(uq_scan_exclusive_u32(ctx, fa, fh, m, nullptr));
        compose_kernel<<<blocks_for(m), ST, 0, s>>>(V1, sid, fa, fh, m, Kfree);
        UQ_LAUNCH_CHECK();
        // sort #2: stable regroup by tie-segment id (high word); K1's buffer is dead and serves as ping-pong
        int alt2 = 0;
        const int sbits = bits_for(nseg > 0 ? nseg - 1 : 0);
        UQ_TRY(radix_sort_pairs(ctx, Kfree, V1, (uint64_t*)K1, Vfree, m, 32, 32 + (sbits ? sbits : 1), rws, &alt2));
        const uint64_t* K2 = alt2 ? K1 : Kfree;
        const uint32_t* V2 = alt2 ? Vfree : V1;
        writeback_kernel<<<blocks_for(m), ST, 0, s>>>(K2, V2, pos, aval, m, d_perm, heads);
        UQ_LAUNCH_CHECK();
    }
    return 0;
}

// ---- unique / inverse from the sorted order + head flags: heads per block of CB positions, then ranks inside the block
__global__ __launch_bounds__(ST) void heads_count_kernel(const uint8_t* __restrict__ heads, uint64_t n, uint32_t* __restrict__ counts) {
    __shared__ uint32_t sc[ST / 64];
    const uint64_t j0 = (uint64_t)blockIdx.x * CB + (uint64_t)threadIdx.x * 4;
    uint32_t c = 0;
    if (j0 + 4 <= n) {            // the lane's four flags in one aligned load (j0 is a multiple of four)
        const uint32_t w = *(const uint32_t*)(heads + j0) ^ 0x01010101u;             // a zero byte = a flag that is 1
        c = (uint32_t)__popc(~(((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w) & 0x80808080u);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) c += (j0 + i < n && heads[j0 + i] == 1) ? 1u : 0u;
    }
    c = wave_sum(c);
    if (lane_id() == 0) sc[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = sc[0] + sc[1] + sc[2] + sc[3];
}
__global__ __launch_bounds__(ST) void keys_from_groups2_kernel(const uint8_t* __restrict__ heads, const uint32_t* __restrict__ boffs,
                                                               const uint32_t* __restrict__ perm, uint64_t n, uint32_t* __restrict__ key,
                                                               uint32_t* __restrict__ sorted_key, uint32_t* __restrict__ uidx) {
    __shared__ uint32_t lds[ST / 64 + 1];
    const uint64_t j0 = (uint64_t)blockIdx.x * CB + (uint64_t)threadIdx.x * 4;
    bool hd[4];
    uint32_t c = 0;
    uint32_t pv[4] = {0, 0, 0, 0};
    if (j0 + 4 <= n) {            // the four flags in one aligned load, the four permutation entries in one (unconditional: all in flight before the scan's barrier)
        const uint32_t w = *(const uint32_t*)(heads + j0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { hd[i] = ((w >> (8 * i)) & 0xFFu) == 1u; c += hd[i]; }
        if (perm) { const uint4 q = *(const uint4*)(perm + j0); pv[0] = q.x; pv[1] = q.y; pv[2] = q.z; pv[3] = q.w; }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { hd[i] = j0 + i < n && heads[j0 + i] == 1; c += hd[i]; if (perm && j0 + i < n) pv[i] = perm[j0 + i]; }
    }
    uint32_t total;
    uint32_t g = block_exclusive_sum<uint32_t, ST / 64>(c, lds, total) + boffs[blockIdx.x];       // heads in front of position j0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint64_t j = j0 + i;
        if (j >= n) break;
        g += hd[i];                              // inclusive count of heads
        const uint32_t row = perm ? pv[i] : (uint32_t)j;            // (no perm: the table itself is in sorted order)
        if (key) key[row] = g - 1;
        if (sorted_key) sorted_key[j] = g - 1;
        if (uidx && hd[i]) uidx[g - 1] = row;
    }
}

// heads of a table that IS sorted: 1 = the row differs from the one in front of it (or is the first), 2 = a duplicate of it.  A lane
// per row; neighbouring rows differ within their first bytes unless they are equal, so the common case is one 8-byte compare.
__global__ __launch_bounds__(ST) void adjacent_heads_kernel(const uint8_t* __restrict__ table, uint64_t n, uint32_t C, uint8_t* __restrict__ heads) {
    const uint64_t j = (uint64_t)blockIdx.x * ST + threadIdx.x;
    if (j >= n) return;
    if (j == 0) { heads[0] = 1; return; }
    const uint8_t* a = table + j * C;
    const uint8_t* b = a - C;
    bool same = true;
    uint32_t i = 0;
    for (; i + 8 <= C; i += 8) {
        unsigned long long x, y;
        __builtin_memcpy(&x, a + i, 8); __builtin_memcpy(&y, b + i, 8);
        if (x != y) { same = false; break; }
    }
    if (same) for (; i < C; ++i) if (a[i] != b[i]) { same = false; break; }
    heads[j] = same ? 2 : 1;
}

// where a new group starts in sorted order: first[group[j]] = j, or the row at j (perm[j]) when the table itself has not been moved
__global__ __launch_bounds__(ST) void group_firsts_kernel(const uint32_t* __restrict__ group, const uint32_t* __restrict__ perm, uint64_t n,
                                                          uint32_t* __restrict__ first) {
    const uint64_t j = (uint64_t)blockIdx.x * ST + threadIdx.x;
    if (j >= n) return;
    const uint32_t g = group[j];
    if (j == 0 || group[j - 1] != g) first[g] = perm ? perm[j] : (uint32_t)j;
}

__global__ void lower_bound_rows_kernel(const uint8_t* __restrict__ table, uint64_t rows, uint32_t C, const uint8_t* __restrict__ probes,
                                        uint64_t nprobes, uint64_t* __restrict__ pos) {
    uint64_t k = (uint64_t)blockIdx.x * ST + threadIdx.x;
    if (k >= nprobes) return;
    const uint8_t* p = probes + k * C;
    uint64_t lo = 0, hi = rows;
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        const uint8_t* r = table + mid * C;
        int cmp = 0;
        for (uint32_t i = 0; i < C; ++i) {
            if (r[i] != p[i]) { cmp = r[i] < p[i] ? -1 : 1; break; }
        }
        if (cmp < 0) lo = mid + 1; else hi = mid;
    }
    pos[k] = lo;
}

template <typename T>
__global__ void narrow_kernel(const uint32_t* __restrict__ in, uint64_t n, T* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * ST + threadIdx.x;
    if (i < n) out[i] = (T)in[i];
}

__global__ void stack_column_kernel(const uint8_t* __restrict__ col, int itemsize, uint64_t n, int ncols, int cidx, int common,
                                    uint8_t* __restrict__ rows) {
    uint64_t i = (uint64_t)blockIdx.x * ST + threadIdx.x;
    if (i >= n) return;
    uint64_t v = 0;
    for (int b = 0; b < itemsize; ++b) v |= (uint64_t)col[i * itemsize + b] << (8 * b);   // little-endian column
    uint8_t* dst = rows + (i * ncols + cidx) * common;
    for (int b = 0; b < common; ++b) dst[b] = (uint8_t)(v >> (8 * (common - 1 - b)));   // big-endian field
}

__global__ void unstack_column_kernel(const uint8_t* __restrict__ rows, uint64_t n, int ncols, int common, int cidx, int out_itemsize,
                                      uint8_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * ST + threadIdx.x;
    if (i >= n) return;
    const uint8_t* src = rows + (i * ncols + cidx) * common;
    uint64_t v = 0;
    for (int b = 0; b < common; ++b) v = (v << 8) | src[b];
    for (int b = 0; b < out_itemsize; ++b) out[i * out_itemsize + b] = (uint8_t)(v >> (8 * b));
}
}  // namespace

int uq_gather_rows_internal(uq_ctx* ctx, const uint8_t* d_table, uint64_t table_rows, uint32_t cols, const void* d_index,
                            int index_itemsize, uint64_t n_out, uint8_t* d_out);

extern "C" int uq_sort_config(uq_ctx* ctx, int64_t msd_min_rows, const int* h_level_bits, int nlevels) {
    UQ_REQUIRE(ctx && nlevels >= 0 && nlevels <= 4 && (nlevels == 0 || h_level_bits), "uq_sort_config: bad argument");
    int tot = 0;
    for (int l = 0; l < nlevels; ++l) { UQ_REQUIRE(h_level_bits[l] >= 1 && h_level_bits[l] <= 10, "uq_sort_config: a level takes 1 .. 10 bits"); tot += h_level_bits[l]; }
    UQ_REQUIRE(tot <= 24, "uq_sort_config: at most 24 bits over the levels");
    ctx->msd_min_rows = msd_min_rows;
    ctx->msd_levels = nlevels;
    for (int l = 0; l < nlevels; ++l) ctx->msd_bits[l] = h_level_bits[l];
    return 0;
}

extern "C" int uq_sort_counters(uq_ctx* ctx, uint64_t* h_msd_rounds, uint64_t* h_lsd_rounds) {
    UQ_REQUIRE(ctx && h_msd_rounds && h_lsd_rounds, "uq_sort_counters: null argument");
    *h_msd_rounds = ctx->n_msd_rounds; *h_lsd_rounds = ctx->n_lsd_rounds;
    return 0;
}

extern "C" int uq_argsort_rows(uq_ctx* ctx, const uint8_t* d_table, uint64_t rows, uint32_t cols, uint32_t* d_perm) {
    UQ_REQUIRE(ctx && (rows == 0 || (d_table && d_perm)), "uq_argsort_rows: null argument");
    UQ_REQUIRE(cols >= 1, "uq_argsort_rows: rows need at least one byte");
    Core c;
    return sort_rows_core(ctx, d_table, rows, cols, d_perm, 0, &c);
}

extern "C" int uq_unique_rows(uq_ctx* ctx, const uint8_t* d_table, uint64_t rows, uint32_t cols, uint32_t* d_perm,
                              uint32_t* d_key, uint32_t* d_sorted_key, uint8_t* d_unique, uint64_t* h_nunique) {
    UQ_REQUIRE(ctx && h_nunique && (rows == 0 || (d_table && d_perm)), "uq_unique_rows: null argument");
    UQ_REQUIRE(cols >= 1, "uq_unique_rows: rows need at least one byte");
    *h_nunique = 0;
    if (rows == 0) return 0;
    Core c;
    const size_t extra = rows * 4 * 3 + 1024;
    UQ_TRY(sort_rows_core(ctx, d_table, rows, cols, d_perm, extra, &c));
    uint32_t* f = (uint32_t*)c.extra;
    uint32_t* gscan = f + rows;
    uint32_t* uidx = gscan + rows;
    uint64_t* tot = (uint64_t*)(uidx + rows);
    tot = (uint64_t*)(((uintptr_t)tot + 7) & ~uintptr_t(7));
    const uint64_t ncb = (rows + CB - 1) / CB;
    heads_count_kernel<<<(uint32_t)ncb, ST, 0, ctx->stream>>>(c.heads, rows, f);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_scan_exclusive_u32(ctx, f, gscan, ncb, tot));
    keys_from_groups2_kernel<<<(uint32_t)ncb, ST, 0, ctx->stream>>>(c.heads, gscan, d_perm, rows, d_key, d_sorted_key, d_unique ? uidx : nullptr);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned, tot, 8));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *h_nunique = ctx->h_pinned[0];
    if (d_unique) UQ_TRY(uq_gather_rows_internal(ctx, d_table, rows, cols, uidx, 4, *h_nunique, d_unique));
    return 0;
}

// unique + group ids of a table that is already in memcmp order (a shard that came out of the global sort): no sort, one
// streaming comparison of neighbours.  d_group[j] = rank of row j's value (dense, from 0); d_unique (may be null) = the distinct rows.
extern "C" int uq_unique_sorted_rows(uq_ctx* ctx, const uint8_t* d_sorted_table, uint64_t rows, uint32_t cols, uint32_t* d_group,
                                     uint8_t* d_unique, uint64_t* h_nunique) {
    UQ_REQUIRE(ctx && h_nunique && cols >= 1 && (rows == 0 || (d_sorted_table && d_group)), "uq_unique_sorted_rows: null argument");
    UQ_REQUIRE(rows < (uint64_t(1) << 32), "uq_unique_sorted_rows: more than 2^32-1 rows");
    *h_nunique = 0;
    if (rows == 0) return 0;
    const uint64_t ncb = (rows + CB - 1) / CB;
    ScratchPlan sp;
    const size_t o_heads = sp.add(rows), o_f = sp.add(ncb * 4), o_scan = sp.add(ncb * 4), o_uidx = sp.add(d_unique ? rows * 4 : 4), o_tot = sp.add(16);
    void* scr;
    UQ_TRY(uq_scratch(ctx, sp.off, &scr));
    uint8_t* base = (uint8_t*)scr;
    uint8_t* heads = base + o_heads;
    uint32_t* f = (uint32_t*)(base + o_f);
    uint32_t* gscan = (uint32_t*)(base + o_scan);
    uint32_t* uidx = (uint32_t*)(base + o_uidx);
    uint64_t* tot = (uint64_t*)(base + o_tot);
    adjacent_heads_kernel<<<blocks_for(rows), ST, 0, ctx->stream>>>(d_sorted_table, rows, cols, heads);
    UQ_LAUNCH_CHECK();
    heads_count_kernel<<<(uint32_t)ncb, ST, 0, ctx->stream>>>(heads, rows, f);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_scan_exclusive_u32(ctx, f, gscan, ncb, tot));
    keys_from_groups2_kernel<<<(uint32_t)ncb, ST, 0, ctx->stream>>>(heads, gscan, nullptr, rows, nullptr, d_group, d_unique ? uidx : nullptr);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned, tot, 8));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *h_nunique = ctx->h_pinned[0];
    if (d_unique) UQ_TRY(uq_gather_rows_internal(ctx, d_sorted_table, rows, cols, uidx, 4, *h_nunique, d_unique));
    return 0;
}

// The distinct rows of a table whose sort has left the order (d_perm) and the group ids of the sorted positions (d_group: dense ranks of
// the row values, uq_unique_rows' d_sorted_key): d_unique[g] = the first row of group g in sorted order.  d_perm = NULL: the table has
// been moved into sorted order already.  No row is compared -- the sort has said where the values change -- and only the distinct
// rows are moved: a keyed table never needs its duplicates in sorted order.
extern "C" int uq_unique_rows_of_groups(uq_ctx* ctx, const uint8_t* d_table, uint64_t rows, uint32_t cols, const uint32_t* d_perm, const uint32_t* d_group,
                                        uint64_t nunique, uint8_t* d_unique) {
    UQ_REQUIRE(ctx && cols >= 1 && nunique <= rows, "uq_unique_rows_of_groups: bad argument");
    if (rows == 0 || nunique == 0) return 0;
    UQ_REQUIRE(d_table && d_group && d_unique && rows < (uint64_t(1) << 32), "uq_unique_rows_of_groups: null buffer or more than 2^32-1 rows");
    void* scr;
    UQ_TRY(uq_scratch(ctx, nunique * 4 + 256, &scr));
    uint32_t* first = (uint32_t*)scr;
    group_firsts_kernel<<<blocks_for(rows), ST, 0, ctx->stream>>>(d_group, d_perm, rows, first);
    UQ_LAUNCH_CHECK();
    return uq_gather_rows_internal(ctx, d_table, rows, cols, first, 4, nunique, d_unique);
}

extern "C" int uq_lower_bound_rows(uq_ctx* ctx, const uint8_t* d_sorted_table, uint64_t rows, uint32_t cols,
                                   const uint8_t* d_probes, uint64_t nprobes, uint64_t* d_pos) {
    UQ_REQUIRE(ctx && cols >= 1, "uq_lower_bound_rows: bad argument");
    if (nprobes == 0) return 0;
    UQ_REQUIRE(d_probes && d_pos && (rows == 0 || d_sorted_table), "uq_lower_bound_rows: null buffer");
    lower_bound_rows_kernel<<<blocks_for(nprobes), ST, 0, ctx->stream>>>(d_sorted_table, rows, cols, d_probes, nprobes, d_pos);
    UQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int uq_key_itemsize(uint64_t max_key) {
    // numpy.min_scalar_type of a non-negative integer (uq.py:790): u1 / u2 / u4 / u8
    if (max_key <= 0xFFull) return 1;
    if (max_key <= 0xFFFFull) return 2;
    if (max_key <= 0xFFFFFFFFull) return 4;
    return 8;
}

extern "C" int uq_narrow(uq_ctx* ctx, const uint32_t* d_key, uint64_t n, int itemsize, void* d_out) {
    UQ_REQUIRE(ctx && (n == 0 || (d_key && d_out)), "uq_narrow: null argument");
    if (n == 0) return 0;
    switch (itemsize) {
        case 1: narrow_kernel<uint8_t><<<blocks_for(n), ST, 0, ctx->stream>>>(d_key, n, (uint8_t*)d_out); break;
        case 2: narrow_kernel<uint16_t><<<blocks_for(n), ST, 0, ctx->stream>>>(d_key, n, (uint16_t*)d_out); break;
        case 4: narrow_kernel<uint32_t><<<blocks_for(n), ST, 0, ctx->stream>>>(d_key, n, (uint32_t*)d_out); break;
        case 8: narrow_kernel<uint64_t><<<blocks_for(n), ST, 0, ctx->stream>>>(d_key, n, (uint64_t*)d_out); break;
        default: UQ_REQUIRE(false, "uq_narrow: itemsize %d not in {1,2,4,8}", itemsize);
    }
    UQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int uq_stack_columns(uq_ctx* ctx, const void* const* h_d_cols, const int* h_itemsize, int ncols, uint64_t n,
                                int common_itemsize, uint8_t* d_rows) {
    UQ_REQUIRE(ctx && h_d_cols && h_itemsize && ncols >= 1, "uq_stack_columns: bad argument");
    UQ_REQUIRE(common_itemsize == 1 || common_itemsize == 2 || common_itemsize == 4 || common_itemsize == 8, "uq_stack_columns: bad common itemsize");
    if (n == 0) return 0;
    for (int c = 0; c < ncols; ++c) {
        UQ_REQUIRE(h_itemsize[c] >= 1 && h_itemsize[c] <= common_itemsize, "uq_stack_columns: column %d wider than the common dtype", c);
        stack_column_kernel<<<blocks_for(n), ST, 0, ctx->stream>>>((const uint8_t*)h_d_cols[c], h_itemsize[c], n, ncols, c, common_itemsize, d_rows);
        UQ_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int uq_unstack_column(uq_ctx* ctx, const uint8_t* d_rows, uint64_t n, int ncols, int common_itemsize, int col,
                                 int out_itemsize, void* d_out) {
    UQ_REQUIRE(ctx && col >= 0 && col < ncols, "uq_unstack_column: bad argument");
    if (n == 0) return 0;
    unstack_column_kernel<<<blocks_for(n), ST, 0, ctx->stream>>>(d_rows, n, ncols, common_itemsize, col, out_itemsize, (uint8_t*)d_out);
    UQ_LAUNCH_CHECK();
    return 0;
}
