// route.hip -- where rows go in the multi-GPU table builds (SURVEY.md 8 row e): the index arithmetic around the exchanges of
// uq_amd/dist.py, as kernels behind the C ABI (the reference has no counterpart: it is one process; the arithmetic stands in for the
// implicit "everything is in one numpy array" of uq.py:767-851).
//   uq_partition_rows      destination rank of every row of an UNSORTED shard, from W-1 splitter rows (sample sort without a local
//                          pre-sort); a value that several splitters share -- a heavy tie group -- is spread over those ranks by file
//                          position, which keeps the global order stable
//   uq_owner_of_rows       owner rank of a file-wide row number (shards are contiguous record ranges)
//   uq_index_affine        out[j] = in[j] + add, between 32- and 64-bit index arrays (local order <-> file-wide row numbers)
//   uq_invert_permutation  inv[perm[j] - base] = j
//   uq_scatter_rows        out[index[j] - base] = values[j]: rows to their places in ONE random pass (instead of inverse + gather)
//   uq_partition_order     the stable partition of a shard's positions by destination rank (what goes in front of an all-to-all)
#include "common.h"

namespace {
constexpr int RT = 256;

__device__ __forceinline__ unsigned long long prefix_key(const uint8_t* r, uint32_t C) {
    unsigned long long k = 0;
    if (C >= 8) { __builtin_memcpy(&k, r, 8); return __builtin_bswap64(k); }      // (global loads need no alignment)
    for (uint32_t i = 0; i < C; ++i) k |= (unsigned long long)r[i] << (56 - 8 * i);
    return k;
}

__device__ __forceinline__ int row_cmp(const uint8_t* a, const uint8_t* b, uint32_t C) {
    for (uint32_t i = 0; i < C; ++i)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}

// lb = #{splitters < row}, ub = #{splitters <= row}; e = ub - lb splitters equal the row.  e <= 1: dest = lb (rows equal to one
// splitter go down with it).  e >= 2: ranks lb .. ub - 1 would otherwise be (lb: everything up to the value, the others: nothing):
// the value's rows are dealt to them by file position, dest = lb + floor(file_index * e / total) -- monotone in the file index, so
// the concatenation of the ranks is still the stable order.
__global__ __launch_bounds__(RT) void partition_rows_kernel(const uint8_t* __restrict__ table, uint64_t rows, uint32_t C, const uint8_t* __restrict__ split,
                                                            uint32_t nsplit, uint64_t index_base, uint64_t total, uint8_t* __restrict__ dest) {
    __shared__ unsigned long long skey[256];
    if (threadIdx.x < 256) skey[threadIdx.x] = threadIdx.x < nsplit ? prefix_key(split + (size_t)threadIdx.x * C, C) : ~0ull;
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * RT + threadIdx.x;
    if (r >= rows) return;
    const uint8_t* row = table + r * C;
    const unsigned long long key = prefix_key(row, C);
    uint32_t lo = 0, hi = nsplit;                       // first splitter whose prefix >= key
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (skey[mid] < key) lo = mid + 1; else hi = mid; }
    uint32_t lb = lo;
    hi = nsplit;                                        // first splitter whose prefix > key
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (skey[mid] <= key) lo = mid + 1; else hi = mid; }
    uint32_t ub = lo;
    if (ub > lb && C > 8) {                             // splitters that share the row's first eight bytes: whole rows decide
        uint32_t a = lb, b = ub;
        while (a < b) { const uint32_t mid = (a + b) >> 1; if (row_cmp(split + (size_t)mid * C, row, C) < 0) a = mid + 1; else b = mid; }
        const uint32_t lb2 = a;
        b = ub;
        while (a < b) { const uint32_t mid = (a + b) >> 1; if (row_cmp(split + (size_t)mid * C, row, C) <= 0) a = mid + 1; else b = mid; }
        lb = lb2; ub = a;
    }
    uint32_t d = lb;
    const uint32_t e = ub - lb;
    if (e >= 2 && total) d = lb + (uint32_t)(((index_base + r) * e) / total);
    dest[r] = (uint8_t)d;
}

struct Starts { long long at[257]; };                   // by value in the kernel-argument segment: nothing of the caller's to keep alive
__global__ __launch_bounds__(RT) void owner_kernel(const long long* __restrict__ gidx, uint64_t n, Starts starts, uint32_t world,
                                                   uint8_t* __restrict__ owner) {
    __shared__ long long s[257];
    for (uint32_t i = threadIdx.x; i <= world; i += RT) s[i] = starts.at[i];
    __syncthreads();
    const uint64_t j = (uint64_t)blockIdx.x * RT + threadIdx.x;
    if (j >= n) return;
    const long long g = gidx[j];
    uint32_t lo = 0, hi = world;                        // the last rank whose start <= g (of several empty shards at one start: the one behind them)
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s[mid] <= g) lo = mid; else hi = mid; }
    owner[j] = (uint8_t)lo;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(RT) void affine_kernel(const TI* __restrict__ in, uint64_t n, long long add, TO* __restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * RT + threadIdx.x;
    if (j < n) out[j] = (TO)((long long)in[j] + add);
}

template <typename TI>
__global__ __launch_bounds__(RT) void invert_kernel(const TI* __restrict__ perm, uint64_t n, long long base, uint32_t* __restrict__ inv, unsigned long long* __restrict__ bad) {
    const uint64_t j = (uint64_t)blockIdx.x * RT + threadIdx.x;
    if (j >= n) return;
    const unsigned long long at = (unsigned long long)((long long)perm[j] - base);
    if (at < n) inv[at] = (uint32_t)j; else atomicMin(bad, (unsigned long long)j);
}

// out[(perm[j] - base) * C ..] = values[j * C ..]: rows to their places (the inverse of a gather, without the inverse permutation).
// A wave moves 64 / lanes-per-row rows at a time, a lane a dword (or the row's tail bytes) of its row.
template <typename TI>
__global__ __launch_bounds__(RT) void scatter_rows_kernel(const uint8_t* __restrict__ values, uint64_t n, uint32_t C, const TI* __restrict__ perm, long long base,
                                                          uint64_t out_rows, uint8_t* __restrict__ out, unsigned long long* __restrict__ bad) {
    const uint32_t lpr = C <= 4 ? 1u : (C <= 8 ? 2u : (C <= 16 ? 4u : (C <= 64 ? 16u : 64u)));     // lanes per row
    const uint64_t slot = ((uint64_t)blockIdx.x * RT + threadIdx.x) / lpr;
    const uint32_t part = threadIdx.x % lpr;
    if (slot >= n) return;
    const unsigned long long at = (unsigned long long)((long long)perm[slot] - base);
    if (at >= out_rows) { if (part == 0) atomicMin(bad, (unsigned long long)slot); return; }
    const uint8_t* src = values + slot * C;
    uint8_t* dst = out + at * C;
    for (uint32_t b = part * 4; b < C; b += lpr * 4) {
        if (b + 4 <= C) { uint32_t w; __builtin_memcpy(&w, src + b, 4); __builtin_memcpy(dst + b, &w, 4); }
        else for (uint32_t k = b; k < C; ++k) dst[k] = src[k];
    }
}

// ---- stable partition of positions 0 .. n - 1 by a one-byte destination (at most 16 destinations: the ranks of a node): order = the
// positions grouped by destination, each group in ascending order; counts[d] = size of group d.  What the exchanges need in front of
// an all-to-all (rows by destination rank, requests by owner rank) without a sort: one histogram pass over the bytes, the scan of the
// per-tile counts, one ranking pass -- a position's rank among its tile's positions of the same destination comes from one ballot per
// destination and popcounts (wave64), the waves' counts through LDS.
constexpr int PT = 4096, PI = PT / RT;              // positions per tile, per lane (position of item i of lane l of wave w: i * 256 + w * 64 + l)
constexpr uint32_t PMAXD = 16;
__global__ __launch_bounds__(RT) void partition_count_kernel(const uint8_t* __restrict__ dest, uint64_t n, uint32_t ndest, uint32_t ntiles,
                                                             uint32_t* __restrict__ tile_counts, unsigned long long* __restrict__ bad) {
    __shared__ uint32_t h[PMAXD];
    if (threadIdx.x < PMAXD) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * PT;
    uint32_t dv[PI];          // (unconditional loads, all in flight at once: a load behind `p < n` is followed by its own wait)
#pragma unroll
    for (int i = 0; i < PI; ++i) { const uint64_t p = base + (uint64_t)i * RT + threadIdx.x; dv[i] = dest[p < n ? p : n - 1]; }
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const uint64_t p = base + (uint64_t)i * RT + threadIdx.x;
        if (p < n) { const uint32_t d = dv[i]; if (d < ndest) atomicAdd(&h[d], 1u); else atomicMin(bad, (unsigned long long)p); }
    }
    __syncthreads();
    if (threadIdx.x < ndest) tile_counts[(uint64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}
__global__ __launch_bounds__(RT) void partition_rank_kernel(const uint8_t* __restrict__ dest, uint64_t n, uint32_t ndest, uint32_t ntiles,
                                                            const uint32_t* __restrict__ tile_offsets, uint32_t* __restrict__ order) {
    __shared__ uint32_t s_cnt[PI * (RT / 64) * PMAXD];          // [item][wave][destination]: count, then the exclusive prefix in position order
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * PT;
    const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    uint32_t d[PI], below[PI];
#pragma unroll
    for (int i = 0; i < PI; ++i) { const uint64_t p = base + (uint64_t)i * RT + tid; d[i] = dest[p < n ? p : n - 1]; }
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const uint64_t p = base + (uint64_t)i * RT + tid;
        if (p >= n) d[i] = 0xFFFFFFFFu;
        below[i] = 0;
        for (uint32_t k = 0; k < ndest; ++k) {
            const uint64_t b = __ballot(d[i] == k);
            if (d[i] == k) below[i] = (uint32_t)__popcll(b & lt);
            if (lane == 0) s_cnt[(i * (RT / 64) + w) * PMAXD + k] = (uint32_t)__popcll(b);
        }
    }
    __syncthreads();
    if (tid < ndest) {                                          // one lane per destination: prefix over the 64 (item, wave) cells, from the tile's first slot
        uint32_t run = tile_offsets[(uint64_t)tid * ntiles + blockIdx.x];
        for (uint32_t c = 0; c < PI * (RT / 64); ++c) { const uint32_t v = s_cnt[c * PMAXD + tid]; s_cnt[c * PMAXD + tid] = run; run += v; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const uint64_t p = base + (uint64_t)i * RT + tid;
        if (p < n && d[i] < ndest) order[s_cnt[(i * (RT / 64) + w) * PMAXD + d[i]] + below[i]] = (uint32_t)p;
    }
}
__global__ void partition_totals_kernel(const uint32_t* __restrict__ tile_offsets, const uint32_t* __restrict__ tile_counts_last, uint32_t ndest, uint32_t ntiles,
                                        uint64_t n, unsigned long long* __restrict__ counts) {
    const uint32_t k = threadIdx.x;                             // counts[k] = first slot of destination k + 1 (or n) - first slot of k
    if (k >= ndest) return;
    const unsigned long long lo = tile_offsets[(uint64_t)k * ntiles];
    const unsigned long long hi = k + 1 < ndest ? tile_offsets[(uint64_t)(k + 1) * ntiles] : n;
    counts[k] = hi - lo;
}

uint32_t blocks_of(uint64_t n) { return (uint32_t)((n + RT - 1) / RT); }
}  // namespace

// d_order[0 .. n) = the positions 0 .. n - 1 grouped by d_dest[position] (ascending), every group in ascending order; d_counts[k] (device,
// uint64) = size of group k.  ndest <= 16.  *h_bad = UQ_NONE or a position whose destination is >= ndest.  Nothing here waits for the
// device unless h_bad is given.
extern "C" int uq_partition_order(uq_ctx* ctx, const uint8_t* d_dest, uint64_t n, uint32_t ndest, uint32_t* d_order, uint64_t* d_counts, uint64_t* h_bad) {
    UQ_REQUIRE(ctx && ndest >= 1 && ndest <= PMAXD && d_counts, "uq_partition_order: 1 .. 16 destinations");
    UQ_REQUIRE(n < (uint64_t(1) << 32), "uq_partition_order: more than 2^32-1 positions");
    if (h_bad) *h_bad = UQ_NONE;
    if (n == 0) { UQ_CHECK_HIP(hipMemsetAsync(d_counts, 0, ndest * 8, ctx->stream)); return 0; }
    UQ_REQUIRE(d_dest && d_order, "uq_partition_order: null buffer");
    const uint32_t ntiles = (uint32_t)((n + PT - 1) / PT);
    void* scr;
    UQ_TRY(uq_scratch(ctx, (size_t)ndest * ntiles * 4 + 512, &scr));
    unsigned long long* bad = (unsigned long long*)scr;
    uint32_t* tc = (uint32_t*)((uint8_t*)scr + 256);
    UQ_CHECK_HIP(hipMemsetAsync(bad, 0xFF, 8, ctx->stream));
    partition_count_kernel<<<ntiles, RT, 0, ctx->stream>>>(d_dest, n, ndest, ntiles, tc, bad);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_scan_exclusive_u32(ctx, tc, tc, (uint64_t)ndest * ntiles, nullptr));
    partition_totals_kernel<<<1, PMAXD, 0, ctx->stream>>>(tc, nullptr, ndest, ntiles, n, (unsigned long long*)d_counts);
    UQ_LAUNCH_CHECK();
    partition_rank_kernel<<<ntiles, RT, 0, ctx->stream>>>(d_dest, n, ndest, ntiles, tc, d_order);
    UQ_LAUNCH_CHECK();
    if (h_bad) {
        UQ_TRY(uq_read_back(ctx, ctx->h_pinned, bad, 8));
        UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        *h_bad = ctx->h_pinned[0];
    }
    return 0;
}

// d_out[(d_index[j] - base) * cols ...] = d_values[j * cols ...] for j < n; every target row must lie in [0, out_rows).  *h_bad = UQ_NONE or
// the lowest j that points outside.  Synchronises.
extern "C" int uq_scatter_rows(uq_ctx* ctx, const uint8_t* d_values, uint64_t n, uint32_t cols, const void* d_index, int index_itemsize, int64_t base,
                               uint64_t out_rows, uint8_t* d_out, uint64_t* h_bad) {
    UQ_REQUIRE(ctx && h_bad && cols >= 1 && (index_itemsize == 4 || index_itemsize == 8), "uq_scatter_rows: bad argument");
    *h_bad = UQ_NONE;
    if (n == 0) return 0;
    UQ_REQUIRE(d_values && d_index && d_out, "uq_scatter_rows: null buffer");
    void* scr;
    UQ_TRY(uq_scratch(ctx, 256, &scr));
    UQ_CHECK_HIP(hipMemsetAsync(scr, 0xFF, 8, ctx->stream));
    const uint32_t lpr = cols <= 4 ? 1u : (cols <= 8 ? 2u : (cols <= 16 ? 4u : (cols <= 64 ? 16u : 64u)));
    const uint32_t blocks = blocks_of(n * lpr);
    if (index_itemsize == 4) scatter_rows_kernel<uint32_t><<<blocks, RT, 0, ctx->stream>>>(d_values, n, cols, (const uint32_t*)d_index, base, out_rows, d_out, (unsigned long long*)scr);
    else scatter_rows_kernel<long long><<<blocks, RT, 0, ctx->stream>>>(d_values, n, cols, (const long long*)d_index, base, out_rows, d_out, (unsigned long long*)scr);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned, scr, 8));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *h_bad = ctx->h_pinned[0];
    return 0;
}

extern "C" int uq_partition_rows(uq_ctx* ctx, const uint8_t* d_splitters, uint32_t nsplit, uint32_t cols, const uint8_t* d_table, uint64_t rows,
                                 uint64_t row_index_base, uint64_t total_rows, uint8_t* d_dest) {
    UQ_REQUIRE(ctx && cols >= 1 && nsplit <= 255, "uq_partition_rows: bad argument (at most 255 splitters)");
    if (rows == 0) return 0;
    UQ_REQUIRE(d_table && d_dest && (nsplit == 0 || d_splitters), "uq_partition_rows: null buffer");
    partition_rows_kernel<<<blocks_of(rows), RT, 0, ctx->stream>>>(d_table, rows, cols, d_splitters, nsplit, row_index_base, total_rows, d_dest);
    UQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int uq_owner_of_rows(uq_ctx* ctx, const int64_t* d_row_index, uint64_t n, const int64_t* h_shard_starts, uint32_t world, uint8_t* d_owner) {
    UQ_REQUIRE(ctx && h_shard_starts && world >= 1 && world <= 256, "uq_owner_of_rows: bad argument (1..256 ranks)");
    if (n == 0) return 0;
    UQ_REQUIRE(d_row_index && d_owner, "uq_owner_of_rows: null buffer");
    Starts st;
    memset(&st, 0, sizeof(st));
    for (uint32_t i = 0; i <= world; ++i) st.at[i] = h_shard_starts[i];
    owner_kernel<<<blocks_of(n), RT, 0, ctx->stream>>>((const long long*)d_row_index, n, st, world, d_owner);
    UQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int uq_index_affine(uq_ctx* ctx, const void* d_in, int in_itemsize, uint64_t n, int64_t add, void* d_out, int out_itemsize) {
    UQ_REQUIRE(ctx && (in_itemsize == 4 || in_itemsize == 8) && (out_itemsize == 4 || out_itemsize == 8), "uq_index_affine: item sizes must be 4 or 8");
    if (n == 0) return 0;
    UQ_REQUIRE(d_in && d_out, "uq_index_affine: null buffer");
    if (in_itemsize == 4 && out_itemsize == 8) affine_kernel<uint32_t, long long><<<blocks_of(n), RT, 0, ctx->stream>>>((const uint32_t*)d_in, n, add, (long long*)d_out);
    else if (in_itemsize == 8 && out_itemsize == 4) affine_kernel<long long, uint32_t><<<blocks_of(n), RT, 0, ctx->stream>>>((const long long*)d_in, n, add, (uint32_t*)d_out);
    else if (in_itemsize == 4) affine_kernel<uint32_t, uint32_t><<<blocks_of(n), RT, 0, ctx->stream>>>((const uint32_t*)d_in, n, add, (uint32_t*)d_out);
    else affine_kernel<long long, long long><<<blocks_of(n), RT, 0, ctx->stream>>>((const long long*)d_in, n, add, (long long*)d_out);
    UQ_LAUNCH_CHECK();
    return 0;
}

// *h_bad = UQ_NONE, or the lowest j whose perm[j] - base lies outside [0, n) (then d_inv is incomplete).  Synchronises.
extern "C" int uq_invert_permutation(uq_ctx* ctx, const void* d_perm, int perm_itemsize, uint64_t n, int64_t base, uint32_t* d_inv, uint64_t* h_bad) {
    UQ_REQUIRE(ctx && h_bad && (perm_itemsize == 4 || perm_itemsize == 8), "uq_invert_permutation: bad argument");
    *h_bad = UQ_NONE;
    if (n == 0) return 0;
    UQ_REQUIRE(d_perm && d_inv && n < (uint64_t(1) << 32), "uq_invert_permutation: null buffer or more than 2^32-1 entries");
    void* scr;
    UQ_TRY(uq_scratch(ctx, 256, &scr));
    UQ_CHECK_HIP(hipMemsetAsync(scr, 0xFF, 8, ctx->stream));
    if (perm_itemsize == 4) invert_kernel<uint32_t><<<blocks_of(n), RT, 0, ctx->stream>>>((const uint32_t*)d_perm, n, base, d_inv, (unsigned long long*)scr);
    else invert_kernel<long long><<<blocks_of(n), RT, 0, ctx->stream>>>((const long long*)d_perm, n, base, d_inv, (unsigned long long*)scr);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned, scr, 8));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *h_bad = ctx->h_pinned[0];
    return 0;
}
