// radix.hip -- stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass, wave64-native.
//
// Per pass: (1) per-tile digit histogram (LDS atomics) -> digit-major table [256][tiles];
// (2) exclusive scan of that table (scan.hip: kernel boundaries only, no inter-workgroup hand-off);
// (3) scatter: each wave ranks its 768 keys 64 at a time with eight __ballot()s per key (the wave64
// form of match-any: lanes holding the same digit find each other, popcount below the lane = stable
// rank), the tile is regrouped by digit in LDS and written out in runs of equal digit, so consecutive
// lanes store to consecutive addresses.
// Passes whose digit is constant over all keys are skipped (one 8-digit census up front).
// HBM traffic per executed pass: 8 B read (count) + 12 B read + 12 B written (scatter) per pair.
#include "radix.h"

namespace {
constexpr int RS_THREADS = 256;
// pairs per thread: 12 with u64 keys = 3072 pairs per workgroup (36 KiB of LDS: 4 workgroups per CU; 16 items = 48 KiB = 3 per CU
// was 8 % slower); u32 keys are 8 bytes a pair, so 16 items fit the same 36 KiB
template <typename K> struct RsGeom { static constexpr int ITEMS = sizeof(K) == 4 ? 16 : 12; static constexpr int TILE = RS_THREADS * ITEMS;
                                      static constexpr int WAVE_KEYS = 64 * ITEMS; static constexpr int DIGITS = sizeof(K); };

template <typename K>
__device__ __forceinline__ uint32_t digit_of(K k, int shift) { return (uint32_t)(k >> shift) & 255u; }

// Census of all digit positions at once: ghist[p][d].
template <typename K>
__global__ __launch_bounds__(RS_THREADS) void rs_census_kernel(const K* __restrict__ keys, uint64_t n,
                                                               uint32_t* __restrict__ ghist) {
    constexpr int ND = RsGeom<K>::DIGITS;
    __shared__ uint32_t h[ND * 256];
    for (int i = threadIdx.x; i < ND * 256; i += RS_THREADS) h[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * RS_THREADS;
    for (uint64_t i = (uint64_t)blockIdx.x * RS_THREADS + threadIdx.x; i < n; i += stride) {
        const K k = keys[i];
#pragma unroll
        for (int p = 0; p < ND; ++p) atomicAdd(&h[p * 256 + digit_of(k, 8 * p)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ND * 256; i += RS_THREADS)
        if (h[i]) atomicAdd(&ghist[i], h[i]);
}

// What the row sort does before its 32-bit round 0, in one pass over the 64-bit chunk values: keys[i] = the 32 bits behind the z
// leading bits, vals[i] = i, the census of all four digits of the new keys, and the per-tile histograms of their lowest digit --
// what the first pass's rs_count would read the keys again for.  Workgroups walk tiles of the sort's own size.
__global__ __launch_bounds__(RS_THREADS) void rs_prefix_census_kernel(const uint64_t* __restrict__ keys64, uint64_t n, uint32_t z, uint32_t* __restrict__ keys,
                                                                      uint32_t* __restrict__ vals, uint32_t nb, uint32_t* __restrict__ ghist,
                                                                      uint32_t* __restrict__ block_hist) {
    constexpr int ITEMS = RsGeom<uint32_t>::ITEMS, TILE = RsGeom<uint32_t>::TILE;
    __shared__ uint32_t h[4 * 256];          // this workgroup's census; digit 0 is added a tile at a time from h0
    __shared__ uint32_t h0[256];             // digit 0 of the tile in hand (bin tid belongs to lane tid outside the counting phase)
    const uint32_t tid = threadIdx.x;
    for (int i = tid; i < 4 * 256; i += RS_THREADS) h[i] = 0;
    for (uint32_t tile = blockIdx.x; tile < nb; tile += gridDim.x) {
        h0[tid] = 0;
        __syncthreads();
        const uint64_t base = (uint64_t)tile * TILE;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint64_t idx = base + (uint64_t)i * RS_THREADS + tid;
            if (idx < n) {
                const uint32_t k = (uint32_t)((keys64[idx] << z) >> 32);
                keys[idx] = k; vals[idx] = (uint32_t)idx;
                atomicAdd(&h0[k & 255u], 1u);
                atomicAdd(&h[256 + ((k >> 8) & 255u)], 1u); atomicAdd(&h[512 + ((k >> 16) & 255u)], 1u); atomicAdd(&h[768 + (k >> 24)], 1u);
            }
        }
        __syncthreads();
        const uint32_t c = h0[tid];
        block_hist[(uint64_t)tid * nb + tile] = c;
        h[tid] += c;
    }
    __syncthreads();
    for (int i = tid; i < 4 * 256; i += RS_THREADS)
        if (h[i]) atomicAdd(&ghist[i], h[i]);
}

// The same straight from the TABLE: the 64-bit chunk values (the first eight bytes of every row, as a big-endian number) are never written out.
// z comes from a sample of the rows (sort.hip); the AND / OR over every row's chunk, left in andor[0 .. 1], tells the caller whether the
// sample's z holds (every row agrees on its z leading bits) -- if not, it runs this again with the true one.
__global__ __launch_bounds__(RS_THREADS) void rs_rows_prefix_census_kernel(const uint8_t* __restrict__ table, uint32_t C, uint64_t n, uint32_t z,
                                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t nb,
                                                                           uint32_t* __restrict__ ghist, uint32_t* __restrict__ block_hist,
                                                                           unsigned long long* __restrict__ andor) {
    constexpr int ITEMS = RsGeom<uint32_t>::ITEMS, TILE = RsGeom<uint32_t>::TILE;
    __shared__ uint32_t h[4 * 256];
    __shared__ uint32_t h0[256];
    __shared__ unsigned long long sa[RS_THREADS / 64], so[RS_THREADS / 64];
    const uint32_t tid = threadIdx.x;
    for (int i = tid; i < 4 * 256; i += RS_THREADS) h[i] = 0;
    unsigned long long a = ~0ull, o = 0ull;
    for (uint32_t tile = blockIdx.x; tile < nb; tile += gridDim.x) {
        h0[tid] = 0;
        __syncthreads();
        const uint64_t base = (uint64_t)tile * TILE;
        uint64_t c64[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {                  // (rows of at least eight bytes: the caller's 32-bit round 0 is for wider tables)
            const uint64_t idx = base + (uint64_t)i * RS_THREADS + tid;
            { uint64_t v; __builtin_memcpy(&v, table + (idx < n ? idx : n - 1) * C, 8); c64[i] = __builtin_bswap64(v); }      // (unconditional: see rs_count_kernel)
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint64_t idx = base + (uint64_t)i * RS_THREADS + tid;
            if (idx < n) {
                a &= c64[i]; o |= c64[i];
                const uint32_t k = (uint32_t)((c64[i] << z) >> 32);
                keys[idx] = k; vals[idx] = (uint32_t)idx;
                atomicAdd(&h0[k & 255u], 1u);
                atomicAdd(&h[256 + ((k >> 8) & 255u)], 1u); atomicAdd(&h[512 + ((k >> 16) & 255u)], 1u); atomicAdd(&h[768 + (k >> 24)], 1u);
            }
        }
        __syncthreads();
        const uint32_t c = h0[tid];
        block_hist[(uint64_t)tid * nb + tile] = c;
        h[tid] += c;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a &= __shfl_xor(a, d, 64); o |= __shfl_xor(o, d, 64); }
    if ((tid & 63) == 0) { sa[tid >> 6] = a; so[tid >> 6] = o; }
    __syncthreads();
    for (int i = tid; i < 4 * 256; i += RS_THREADS)
        if (h[i]) atomicAdd(&ghist[i], h[i]);
    if (tid == 0) {
        for (int w = 1; w < RS_THREADS / 64; ++w) { a &= sa[w]; o |= so[w]; }
        atomicAnd(andor, a); atomicOr(andor + 1, o);
    }
}

template <typename K>
__global__ __launch_bounds__(RS_THREADS) void rs_count_kernel(const K* __restrict__ keys, uint64_t n, int shift,
                                                              uint32_t nb, uint32_t* __restrict__ block_hist) {
    constexpr int RS_ITEMS = RsGeom<K>::ITEMS, RS_TILE = RsGeom<K>::TILE;
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
    K kv[RS_ITEMS];           // all of the lane's keys are requested before the first is counted (a load behind `idx < n` is followed by its own wait)
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) { const uint64_t idx = base + (uint64_t)i * RS_THREADS + threadIdx.x; kv[i] = keys[idx < n ? idx : n - 1]; }
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        uint64_t idx = base + (uint64_t)i * RS_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&h[digit_of(kv[i], shift)], 1u);
    }
    __syncthreads();
    block_hist[(uint64_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

template <typename K>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                K* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                uint64_t n, int shift, uint32_t nb,
                                                                const uint32_t* __restrict__ block_off) {
    constexpr int RS_ITEMS = RsGeom<K>::ITEMS, RS_TILE = RsGeom<K>::TILE, RS_WAVE_KEYS = RsGeom<K>::WAVE_KEYS;
    __shared__ K s_keys[RS_TILE];
    __shared__ uint32_t s_vals[RS_TILE];
    __shared__ uint32_t s_cnt[4 * 256];      // per-wave digit counts, then per-wave exclusive bases
    __shared__ uint32_t s_start[256];        // first tile-local slot of each digit
    __shared__ uint32_t s_goff[256];         // first global slot of each digit for this tile
    __shared__ uint32_t s_scan[RS_THREADS / 64 + 1];

    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t tile0 = (uint64_t)blockIdx.x * RS_TILE;
    const uint32_t ntile = (uint32_t)((n - tile0) < RS_TILE ? (n - tile0) : RS_TILE);
    for (int i = tid; i < 4 * 256; i += RS_THREADS) s_cnt[i] = 0;
    s_goff[tid] = block_off[(uint64_t)tid * nb + blockIdx.x];
    __syncthreads();

    K key[RS_ITEMS];
    uint32_t val[RS_ITEMS];
    uint32_t rank[RS_ITEMS];
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {                   // (unconditional loads, all in flight at once; slots beyond the tile read its first pair and are never used)
        const uint32_t loc = w * RS_WAVE_KEYS + i * 64 + lane, lc = loc < ntile ? loc : 0u;
        key[i] = keys_in[tile0 + lc];
        val[i] = vals_in[tile0 + lc];
    }
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const uint32_t loc = w * RS_WAVE_KEYS + i * 64 + lane;
        const bool valid = loc < ntile;
        const uint32_t d = digit_of(key[i], shift);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        // lanes with `valid` false hold an arbitrary peers mask; they never touch LDS below
        const uint32_t leader = (uint32_t)(__ffsll((unsigned long long)peers) - 1);
        uint32_t prev = 0;
        if (valid && lane == leader) {
            prev = s_cnt[w * 256 + d];
            s_cnt[w * 256 + d] = prev + (uint32_t)__popcll(peers);
        }
        prev = __shfl(prev, valid ? leader : lane, 64);
        rank[i] = prev + (uint32_t)__popcll(peers & lt_mask);
    }
    __syncthreads();
    // digit `tid`: exclusive bases over the four waves, tile total
    uint32_t tot = 0;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
        uint32_t c = s_cnt[ww * 256 + tid];
        s_cnt[ww * 256 + tid] = tot;
        tot += c;
    }
    uint32_t all;
    uint32_t start = block_exclusive_sum<uint32_t, RS_THREADS / 64>(tot, s_scan, all);
    s_start[tid] = start;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const uint32_t loc = w * RS_WAVE_KEYS + i * 64 + lane;
        if (loc < ntile) {
            const uint32_t d = digit_of(key[i], shift);
            const uint32_t p = s_start[d] + s_cnt[w * 256 + d] + rank[i];
            s_keys[p] = key[i];
            s_vals[p] = val[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const uint32_t p = i * RS_THREADS + tid;
        if (p < ntile) {
            const K k = s_keys[p];
            const uint32_t d = digit_of(k, shift);
            const uint32_t g = s_goff[d] + (p - s_start[d]);
            keys_out[g] = k;
            vals_out[g] = s_vals[p];
        }
    }
}
}  // namespace

size_t radix_ws_bytes(uint64_t n) {
    uint64_t nb = (n + RsGeom<uint64_t>::TILE - 1) / RsGeom<uint64_t>::TILE;      // the smaller tile: more workgroups
    return (size_t)(256 * nb + 64) * 4 + 8 * 256 * 4 + 256;
}

template <typename K>
static int radix_sort_impl(uq_ctx* ctx, K* keys, uint32_t* vals, K* keys_alt, uint32_t* vals_alt,
                           uint64_t n, int begin_bit, int end_bit, void* ws, int* in_alt, const uint32_t* h_hist_in, bool digit0_counted = false) {
    constexpr int ND = RsGeom<K>::DIGITS, TILE = RsGeom<K>::TILE;
    *in_alt = 0;
    if (n <= 1 || end_bit <= begin_bit) return 0;
    UQ_REQUIRE(n < (uint64_t(1) << 32), "radix sort: more than 2^32-1 pairs");
    UQ_REQUIRE(begin_bit % 8 == 0 && begin_bit >= 0 && end_bit <= 8 * ND, "radix sort: bad bit range");
    const uint32_t nb = (uint32_t)((n + TILE - 1) / TILE);
    uint32_t* ghist = (uint32_t*)ws;
    uint32_t* block_hist = ghist + 8 * 256 + 64;
    static thread_local uint32_t h_hist[8 * 256];
    if (h_hist_in) memcpy(h_hist, h_hist_in, ND * 256 * 4);          // the caller has taken the census already
    else {
        UQ_CHECK_HIP(hipMemsetAsync(ghist, 0, ND * 256 * 4, ctx->stream));
        uint32_t cb = nb < 2048 ? nb : 2048;
        rs_census_kernel<K><<<cb, RS_THREADS, 0, ctx->stream>>>(keys, n, ghist);
        UQ_LAUNCH_CHECK();
        UQ_TRY(uq_read_back(ctx, ctx->h_pinned + 4096, ghist, ND * 256 * 4));          // (pageable h_hist: through the pinned staging)
        UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        memcpy(h_hist, ctx->h_pinned + 4096, ND * 256 * 4);
    }
    K* kin = keys; uint32_t* vin = vals; K* kout = keys_alt; uint32_t* vout = vals_alt;
    for (int p = begin_bit / 8; p * 8 < end_bit; ++p) {
        bool trivial = false;
        for (int d = 0; d < 256; ++d)
            if (h_hist[p * 256 + d] == n) { trivial = true; break; }
        if (trivial) continue;
        if (!(digit0_counted && p == 0)) {               // (digit 0's per-tile counts came with the census: radix_prefix_census32)
            rs_count_kernel<K><<<nb, RS_THREADS, 0, ctx->stream>>>(kin, n, 8 * p, nb, block_hist);
            UQ_LAUNCH_CHECK();
        }
        UQ_TRY(uq_scan_exclusive_u32(ctx, block_hist, block_hist, (uint64_t)256 * nb, nullptr));
        rs_scatter_kernel<K><<<nb, RS_THREADS, 0, ctx->stream>>>(kin, vin, kout, vout, n, 8 * p, nb, block_hist);
        UQ_LAUNCH_CHECK();
        K* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
        *in_alt ^= 1;
    }
    return 0;
}

int radix_sort_pairs(uq_ctx* ctx, uint64_t* keys, uint32_t* vals, uint64_t* keys_alt, uint32_t* vals_alt,
                     uint64_t n, int begin_bit, int end_bit, void* ws, int* in_alt) {
    return radix_sort_impl<uint64_t>(ctx, keys, vals, keys_alt, vals_alt, n, begin_bit, end_bit, ws, in_alt, nullptr);
}

int radix_sort_pairs32(uq_ctx* ctx, uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt,
                       uint64_t n, int begin_bit, int end_bit, void* ws, int* in_alt, const uint32_t* h_hist, int digit0_counted) {
    return radix_sort_impl<uint32_t>(ctx, keys, vals, keys_alt, vals_alt, n, begin_bit, end_bit, ws, in_alt, h_hist, digit0_counted != 0 && begin_bit == 0);
}

int radix_prefix_census32(uq_ctx* ctx, const uint64_t* keys64, uint64_t n, uint32_t z, uint32_t* keys, uint32_t* vals, void* ws, uint32_t* h_hist) {
    uint32_t* ghist = (uint32_t*)ws;
    uint32_t* block_hist = ghist + 8 * 256 + 64;                      // where radix_sort_impl keeps its per-tile counts
    UQ_CHECK_HIP(hipMemsetAsync(ghist, 0, 4 * 256 * 4, ctx->stream));
    const uint32_t nb = (uint32_t)((n + RsGeom<uint32_t>::TILE - 1) / RsGeom<uint32_t>::TILE);
    rs_prefix_census_kernel<<<nb < 2048 ? (nb ? nb : 1) : 2048, RS_THREADS, 0, ctx->stream>>>(keys64, n, z, keys, vals, nb, ghist, block_hist);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned + 4096, ghist, 4 * 256 * 4));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(h_hist, ctx->h_pinned + 4096, 4 * 256 * 4);
    return 0;
}

int radix_rows_prefix_census32(uq_ctx* ctx, const uint8_t* table, uint32_t C, uint64_t n, uint32_t z, uint32_t* keys, uint32_t* vals, void* ws,
                               uint32_t* h_hist, uint64_t* h_andor) {
    uint32_t* ghist = (uint32_t*)ws;
    uint32_t* block_hist = ghist + 8 * 256 + 64;
    unsigned long long* andor = (unsigned long long*)(ghist + 8 * 256);        // the 64 spare words between the census and the per-tile counts
    UQ_CHECK_HIP(hipMemsetAsync(ghist, 0, 4 * 256 * 4, ctx->stream));
    UQ_CHECK_HIP(hipMemsetAsync(andor, 0xFF, 8, ctx->stream));
    UQ_CHECK_HIP(hipMemsetAsync(andor + 1, 0, 8, ctx->stream));
    const uint32_t nb = (uint32_t)((n + RsGeom<uint32_t>::TILE - 1) / RsGeom<uint32_t>::TILE);
    rs_rows_prefix_census_kernel<<<nb < 2048 ? (nb ? nb : 1) : 2048, RS_THREADS, 0, ctx->stream>>>(table, C, n, z, keys, vals, nb, ghist, block_hist, andor);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned + 4096, ghist, 4 * 256 * 4));
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned + 4096 + 512, andor, 16));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(h_hist, ctx->h_pinned + 4096, 4 * 256 * 4);
    h_andor[0] = ctx->h_pinned[4096 + 512]; h_andor[1] = ctx->h_pinned[4096 + 513];
    return 0;
}
