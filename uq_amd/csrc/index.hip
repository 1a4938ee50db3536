// index.hip -- record index of a FASTQ byte stream resident in HBM.
// Replaces `wc -l` (uq.py:85-87) and the `next(f)` line iteration every pass of the reference
// repeats (uq.py:132-137, 205-210, 378-385, 563-569): one newline census, one scan, one scatter.
//
// Census: a workgroup of 4 waves owns a 16 KiB tile; wave w owns the 4 KiB chunk w of it and reads it
// as 4 x (64 lanes x 16 B) fully coalesced wave-loads.  Newlines are found with an exact SWAR
// zero-byte test on (word ^ 0x0A0A0A0A).  Besides the per-tile count the census leaves the newline
// BITMAP of the stream (1 bit per byte, one u16 per 16-byte vector, coalesced 128 B per wave-load) in a
// context-owned buffer, so the scatter pass reads nbytes / 8 instead of the stream again: a lane takes
// one u64 of bitmap (64 stream bytes), ranks its newlines with a block prefix sum and writes them.
// HBM traffic: nbytes read + nbytes/8 written, then nbytes/8 read + 8 B per line written
// (algorithmic: nbytes + 8 B/line).
//
// uq_count_lines runs the LIST form of the census instead: the tile's newlines are ranked in the census itself (two packed
// wave scans) and their 14-bit in-tile offsets go, in stream order, to the tile's slot of IDX_LIST_CAP u16 entries; the
// index pass then only expands the lists (2 B read + 8 B written per line).  The bitmap's 2 x nbytes/8 of traffic
// becomes 2 x 2 B per line: 0.86 -> 0.7x ms for census + index at 10 M x 150 bp.  A tile with more newlines than its slot
// holds (lines shorter than 16 bytes on average) raises a flag and uq_index_lines falls back to the bitmap form.
#include "common.h"
#include "swar.h"

namespace {
constexpr int IDX_THREADS = 256;
constexpr int IDX_LOADS = 4;                                  // 16-byte loads per lane
constexpr uint64_t IDX_WAVE_BYTES = 64 * 16 * IDX_LOADS;      // 4096
constexpr uint64_t IDX_TILE = IDX_WAVE_BYTES * (IDX_THREADS / 64);  // 16384
constexpr uint64_t IDX_TILE_VECS = IDX_TILE / 16;             // 1024 vectors = 256 bitmap words of 64 bits
constexpr uint32_t IDX_LIST_CAP = 1024;                       // newline offsets a tile's list slot holds

// `abuf` = buf rounded down to 16 bytes, `mis` = buf - abuf.  Stream position of abuf[x] is x - mis.
// bitmap[vi] = newline mask of vector vi (0 beyond the stream; every vector of every tile is written).
__global__ __launch_bounds__(IDX_THREADS) void count_newlines_kernel(const uint4* __restrict__ abuf, uint32_t mis,
                                                                      uint64_t nbytes, uint64_t nvec,
                                                                      uint32_t* __restrict__ partials, uint16_t* __restrict__ bitmap) {
    __shared__ uint32_t lds[IDX_THREADS / 64];
    const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
    const uint64_t v0 = ((uint64_t)blockIdx.x * IDX_TILE + (uint64_t)w * IDX_WAVE_BYTES) / 16 + lane;
    uint32_t c = 0;
    uint4 q[IDX_LOADS];           // (all requested before the first is looked at: see census_list_kernel)
#pragma unroll
    for (int it = 0; it < IDX_LOADS; ++it) { const uint64_t vi = v0 + (uint64_t)it * 64; q[it] = abuf[vi < nvec ? vi : nvec - 1]; }
#pragma unroll
    for (int it = 0; it < IDX_LOADS; ++it) {
        uint64_t vi = v0 + (uint64_t)it * 64;
        int64_t p = (int64_t)(vi * 16) - (int64_t)mis;
        const uint32_t m = vi < nvec ? nl_mask16(q[it]) & valid_mask16(p, nbytes) : 0u;
        bitmap[vi] = (uint16_t)m;
        c += __popc(m);
    }
    c = wave_sum(c);
    if (lane == 0) lds[w] = c;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// The list form.  Stream order inside a tile = (wave, load, lane, bit); the counts of a lane's four vectors are scanned two
// to a word (16-bit fields: a wave's sum stays below 2^16).
#ifndef CENSUS_TPW
#define CENSUS_TPW 2        // (same box: one tile a workgroup 2.46 ms per bench step, two 2.42, four 2.47)
#endif
// TPW tiles per workgroup, one behind the other: the next tile's vectors are requested before the current one's are looked at.
template <int TPW>
__global__ __launch_bounds__(IDX_THREADS) void census_list_kernel(const uint4* __restrict__ abuf, uint32_t mis, uint64_t nbytes, uint64_t nvec,
                                                                   uint32_t* __restrict__ partials, uint16_t* __restrict__ list,
                                                                   uint32_t* __restrict__ overflow, uint64_t tile0, uint64_t tile_end) {
    __shared__ uint32_t lds[2][IDX_THREADS / 64];               // (by the parity of the tile's turn: one barrier a tile)
    const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
    uint64_t tile = tile0 + (uint64_t)blockIdx.x * TPW;         // tile0: the chunked census (uq_count_lines_chunk) walks the buffer in pieces
    // all of the lane's vectors are requested before the first is looked at (a load inside the `vi < nvec` branch was followed by its own wait: one
    // vector in flight per lane); beyond the stream the last vector is read again and masked out
    auto request = [&](uint64_t tl, uint4 (&q)[IDX_LOADS]) {
        const uint64_t v0 = (tl * IDX_TILE + (uint64_t)w * IDX_WAVE_BYTES) / 16 + lane;
#pragma unroll
        for (int it = 0; it < IDX_LOADS; ++it) { const uint64_t vi = v0 + (uint64_t)it * 64; q[it] = abuf[vi < nvec ? vi : nvec - 1]; }
    };
    uint4 q[IDX_LOADS];
    request(tile, q);
    bool over = false;
#pragma unroll
    for (int turn = 0; turn < TPW; ++turn, ++tile) {
        if (tile >= tile_end) break;
        uint4 qn[IDX_LOADS];
        if (turn + 1 < TPW && tile + 1 < tile_end) request(tile + 1, qn);
        const uint64_t v0 = (tile * IDX_TILE + (uint64_t)w * IDX_WAVE_BYTES) / 16 + lane;
        uint32_t m[IDX_LOADS];
#pragma unroll
        for (int it = 0; it < IDX_LOADS; ++it) {
            const uint64_t vi = v0 + (uint64_t)it * 64;
            const int64_t p = (int64_t)(vi * 16) - (int64_t)mis;
            m[it] = vi < nvec ? nl_mask16(q[it]) & valid_mask16(p, nbytes) : 0u;
        }
        const uint32_t c0 = __popc(m[0]), c1 = __popc(m[1]), c2 = __popc(m[2]), c3 = __popc(m[3]);
        const uint32_t i01 = wave_inclusive_sum(c0 | (c1 << 16)), i23 = wave_inclusive_sum(c2 | (c3 << 16));
        const uint32_t t01 = __shfl(i01, 63, 64), t23 = __shfl(i23, 63, 64);
        const uint32_t T0 = t01 & 0xFFFFu, T1 = t01 >> 16, T2 = t23 & 0xFFFFu, T3 = t23 >> 16;
        uint32_t* l = lds[turn & 1];
        if (lane == 0) l[w] = T0 + T1 + T2 + T3;
        __syncthreads();
        uint32_t base = 0, total = 0;
#pragma unroll
        for (uint32_t i = 0; i < IDX_THREADS / 64; ++i) { const uint32_t x = l[i]; if (i < w) base += x; total += x; }
        if (threadIdx.x == 0) partials[tile] = total;
        const uint32_t ex[IDX_LOADS] = {base + (i01 & 0xFFFFu) - c0, base + T0 + (i01 >> 16) - c1, base + T0 + T1 + (i23 & 0xFFFFu) - c2,
                                        base + T0 + T1 + T2 + (i23 >> 16) - c3};
        uint16_t* slot = list + tile * IDX_LIST_CAP;
#pragma unroll
        for (int it = 0; it < IDX_LOADS; ++it) {
            uint32_t mm = m[it], k = ex[it];
            const uint32_t pos0 = ((w * IDX_LOADS + it) * 64 + lane) * 16;
            while (mm) {
                const uint32_t b = (uint32_t)__ffs((int)mm) - 1u;
                mm &= mm - 1;
                if (k < IDX_LIST_CAP) slot[k] = (uint16_t)(pos0 + b); else over = true;
                ++k;
            }
        }
        if (turn + 1 < TPW) {
#pragma unroll
            for (int it = 0; it < IDX_LOADS; ++it) q[it] = qn[it];
        }
    }
    if (over) atomicOr(overflow, 1u);
}

// A wave per tile: line_start[rank + 1] = stream position of the byte after the newline.
// d_async (the queued form, uq_index_lines_async): the line count is read from d_async[0] -- the census's scan, queued in front,
// left it there -- and line_start holds `cap` + 1 entries: lines beyond are dropped and d_async[1] is raised.
constexpr uint32_t EXP_TILES = 4;            // census tiles per wave: a tile's ~200 offsets alone are too little work for a wave's set-up
// d_over: the census's overflow word (a tile held more newlines than its list slot).  The queued form runs before the host has
// seen it: a tile whose count exceeds its slot is skipped (its list is not there) and d_async[1] is raised, so that every later
// consumer of this index (uq_pack_stats_async) stands down; the plain form is only called when the host has read the word as 0.
__global__ __launch_bounds__(IDX_THREADS) void expand_list_kernel(const uint16_t* __restrict__ list, uint32_t mis, const uint32_t* __restrict__ offsets,
                                                                   uint64_t nb, uint64_t nlines, uint64_t* __restrict__ line_start,
                                                                   unsigned long long* __restrict__ d_async, uint64_t cap,
                                                                   const uint32_t* __restrict__ d_over) {
    const uint32_t lane = lane_id();
    const uint64_t tile0 = ((uint64_t)blockIdx.x * (IDX_THREADS / 64) + (threadIdx.x >> 6)) * EXP_TILES;
    if (tile0 >= nb) return;
    if (d_async) nlines = d_async[0];
    if (tile0 == 0 && lane == 0) line_start[0] = 0;
    if (d_async && *d_over) { if (lane == 0) d_async[1] = 1ull; return; }      // some tile's list is incomplete: no index at all
    // the tiles' first ranks (and the rank behind the last one) once, a lane each
    const uint64_t tl = tile0 + lane;
    const uint64_t mine = lane <= EXP_TILES ? (tl < nb ? (uint64_t)offsets[tl] : nlines) : 0;
    for (uint32_t k = 0; k < EXP_TILES && tile0 + k < nb; ++k) {
        const uint64_t tile = tile0 + k;
        const uint64_t P = __shfl(mine, k, 64), end = __shfl(mine, k + 1, 64);
        uint32_t cnt = (uint32_t)(end - P);
        const uint16_t* slot = list + tile * IDX_LIST_CAP;
        const int64_t p0 = (int64_t)(tile * IDX_TILE) - (int64_t)mis + 1;
        if (cnt > IDX_LIST_CAP) {                                              // never with a census whose overflow word is 0; keeps the reads inside the slot
            if (d_async && lane == 0) d_async[1] = 1ull;
            cnt = IDX_LIST_CAP;
        }
        if (d_async && end > cap) {
            if (lane == 0) d_async[1] = 1ull;
            for (uint32_t j = lane; j < cnt; j += 64) if (P + j + 1 <= cap) line_start[P + j + 1] = (uint64_t)(p0 + slot[j]);
            continue;
        }
        for (uint32_t j = lane; j < cnt; j += 64) line_start[P + j + 1] = (uint64_t)(p0 + slot[j]);
    }
}

// One workgroup per census tile: 256 lanes x one 64-bit bitmap word (= 64 stream bytes) each.
__global__ __launch_bounds__(IDX_THREADS) void scatter_newlines_kernel(const uint64_t* __restrict__ bitmap, uint32_t mis,
                                                                        const uint32_t* __restrict__ offsets,
                                                                        uint64_t nlines, uint64_t* __restrict__ line_start) {
    __shared__ uint32_t lds[IDX_THREADS / 64 + 1];
    const uint64_t wi = (uint64_t)blockIdx.x * IDX_THREADS + threadIdx.x;
    uint64_t m = bitmap[wi];
    const uint32_t c = (uint32_t)__popcll(m);
    uint32_t total;
    const uint32_t ex = block_exclusive_sum<uint32_t, IDX_THREADS / 64>(c, lds, total);
    if (blockIdx.x == 0 && threadIdx.x == 0) line_start[0] = 0;
    uint64_t k = (uint64_t)offsets[blockIdx.x] + ex;
    const int64_t p = (int64_t)(wi * 64) - (int64_t)mis;
    while (m) {
        const int b = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
        if (++k <= nlines) line_start[k] = (uint64_t)(p + b + 1);
    }
}

int census_buffers(uq_ctx* ctx, uint64_t nb) {
    if (nb + 1 > ctx->idx_partials_cap) {
        UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->idx_partials) UQ_CHECK_HIP(hipFree(ctx->idx_partials));
        if (ctx->idx_bitmap) UQ_CHECK_HIP(hipFree(ctx->idx_bitmap));
        ctx->idx_partials = nullptr; ctx->idx_bitmap = nullptr; ctx->idx_partials_cap = 0;
        UQ_CHECK_HIP(hipMalloc((void**)&ctx->idx_partials, (nb + 1) * sizeof(uint32_t)));
        // one buffer serves both forms: nb * 1024 u16 = the bitmap of nb tiles = nb list slots of IDX_LIST_CAP entries
        static_assert(IDX_TILE_VECS == IDX_LIST_CAP, "bitmap words and list entries per tile share one allocation");
        UQ_CHECK_HIP(hipMalloc((void**)&ctx->idx_bitmap, nb * IDX_TILE_VECS * sizeof(uint16_t) + 16));
        ctx->idx_partials_cap = nb + 1;
    }
    return 0;
}

int run_count(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t* nblocks_out, bool list_form) {
    const uint32_t mis = (uint32_t)((uintptr_t)d_buf & 15);
    const uint64_t nvec = (nbytes + mis + 15) / 16;
    const uint64_t nb = (nvec * 16 + IDX_TILE - 1) / IDX_TILE;
    UQ_REQUIRE(nb <= 0x7fffffffu, "uq_count_lines: buffer too large for one launch");
    ctx->async_buf = nullptr;                              // the lists of a queued census (if any) are about to be overwritten: its index-free consumers must refuse
    UQ_TRY(census_buffers(ctx, nb));
    if (list_form) {
        uint32_t* d_over = (uint32_t*)(ctx->idx_bitmap + nb * IDX_TILE_VECS);          // the 16 spare bytes behind the slots
        UQ_CHECK_HIP(hipMemsetAsync(d_over, 0, 4, ctx->stream));
        census_list_kernel<CENSUS_TPW><<<(uint32_t)((nb + CENSUS_TPW - 1) / CENSUS_TPW), IDX_THREADS, 0, ctx->stream>>>((const uint4*)(d_buf - mis), mis, nbytes, nvec, ctx->idx_partials,
                                                                         ctx->idx_bitmap, d_over, 0, nb);
    } else {
        count_newlines_kernel<<<(uint32_t)nb, IDX_THREADS, 0, ctx->stream>>>((const uint4*)(d_buf - mis), mis, nbytes, nvec, ctx->idx_partials,
                                                                            ctx->idx_bitmap);
    }
    UQ_LAUNCH_CHECK();
    *nblocks_out = nb;
    return 0;
}
}  // namespace

static int finish_count(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t nb, uint64_t* h_nlines);

extern "C" int uq_count_lines(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t* h_nlines) {
    UQ_REQUIRE(ctx && h_nlines, "uq_count_lines: null argument");
    ctx->idx_buf = nullptr;
    if (nbytes == 0) { *h_nlines = 0; return 0; }
    UQ_REQUIRE(d_buf, "uq_count_lines: null buffer");
    uint64_t nb;
    UQ_TRY(run_count(ctx, d_buf, nbytes, &nb, true));
    // exclusive scan of the per-tile counts right away: its total is the census, and uq_index_lines
    // reuses the scanned offsets (and the lists) for the same buffer
    return finish_count(ctx, d_buf, nbytes, nb, h_nlines);
}

// ---- the census in pieces: a file arrives in HBM chunk by chunk; every chunk's newlines are counted and
// listed while the next one is still crossing PCIe, so that only the scan of the per-tile counts is left when the last byte lands.
static int finish_count(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t nb, uint64_t* h_nlines) {
    void* scr;
    UQ_TRY(uq_scratch(ctx, 256, &scr));
    UQ_TRY(uq_scan_exclusive_u32(ctx, ctx->idx_partials, ctx->idx_partials, nb, (uint64_t*)scr));
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned, scr, 8));
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned + 1, ctx->idx_bitmap + nb * IDX_TILE_VECS, 4));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *h_nlines = ctx->h_pinned[0];
    // the lists are only good when every tile's newlines fitted its slot; otherwise uq_index_lines runs the bitmap form
    if ((uint32_t)ctx->h_pinned[1] == 0) { ctx->idx_buf = d_buf; ctx->idx_nbytes = nbytes; ctx->idx_nlines = *h_nlines; }
    return 0;
}

extern "C" int uq_count_lines_begin(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes) {
    UQ_REQUIRE(ctx && (d_buf || nbytes == 0), "uq_count_lines_begin: null argument");
    ctx->idx_buf = nullptr;
    ctx->async_buf = nullptr;
    if (nbytes == 0) return 0;
    const uint32_t mis = (uint32_t)((uintptr_t)d_buf & 15);
    const uint64_t nb = (((nbytes + mis + 15) / 16) * 16 + IDX_TILE - 1) / IDX_TILE;
    UQ_REQUIRE(nb <= 0x7fffffffu, "uq_count_lines_begin: buffer too large");
    UQ_TRY(census_buffers(ctx, nb));
    UQ_CHECK_HIP(hipMemsetAsync(ctx->idx_bitmap + nb * IDX_TILE_VECS, 0, 4, ctx->stream));
    return 0;
}

extern "C" int uq_count_lines_chunk(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t first_byte, uint64_t chunk_bytes) {
    UQ_REQUIRE(ctx && d_buf, "uq_count_lines_chunk: null argument");
    if (chunk_bytes == 0) return 0;
    const uint32_t mis = (uint32_t)((uintptr_t)d_buf & 15);
    const uint64_t nvec = (nbytes + mis + 15) / 16;
    const uint64_t nb = (nvec * 16 + IDX_TILE - 1) / IDX_TILE;
    UQ_REQUIRE(first_byte + chunk_bytes <= nbytes, "uq_count_lines_chunk: chunk beyond the buffer");
    // chunks are whole census tiles (16 KiB of the 16-byte-aligned address space), except that the last one ends with the buffer
    UQ_REQUIRE((first_byte + mis) % IDX_TILE == 0 || first_byte == 0, "uq_count_lines_chunk: a chunk must start on a 16 KiB tile boundary");
    UQ_REQUIRE((first_byte + chunk_bytes + mis) % IDX_TILE == 0 || first_byte + chunk_bytes == nbytes,
               "uq_count_lines_chunk: a chunk must end on a 16 KiB tile boundary or with the buffer");
    const uint64_t t0 = (first_byte + mis) / IDX_TILE, t1 = first_byte + chunk_bytes == nbytes ? nb : (first_byte + chunk_bytes + mis) / IDX_TILE;
    if (t1 <= t0) return 0;
    census_list_kernel<CENSUS_TPW><<<(uint32_t)((t1 - t0 + CENSUS_TPW - 1) / CENSUS_TPW), IDX_THREADS, 0, ctx->stream>>>((const uint4*)(d_buf - mis), mis, nbytes, nvec, ctx->idx_partials,
                                                                            ctx->idx_bitmap, (uint32_t*)(ctx->idx_bitmap + nb * IDX_TILE_VECS), t0, t1);
    UQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int uq_count_lines_end(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t* h_nlines) {
    UQ_REQUIRE(ctx && h_nlines, "uq_count_lines_end: null argument");
    if (nbytes == 0) { *h_nlines = 0; return 0; }
    const uint32_t mis = (uint32_t)((uintptr_t)d_buf & 15);
    const uint64_t nb = (((nbytes + mis + 15) / 16) * 16 + IDX_TILE - 1) / IDX_TILE;
    return finish_count(ctx, d_buf, nbytes, nb, h_nlines);
}

// ---- the census and the record index without a host round trip between them: the closing scan is queued behind the chunks and
// leaves the line count ON THE DEVICE (ctx->d_async); uq_index_lines_async and uq_pack_stats_async, queued behind it, read it
// there; the host learns it from uq_count_lines_wait, after everything else of the step has been queued.  (A step that waits for
// the count before it can size and queue the index and the pack kernel leaves the device idle for two launch latencies and a
// synchronisation: ~0.14 ms of a 2.4 ms step.)
extern "C" int uq_count_lines_end_async(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes) {
    UQ_REQUIRE(ctx && (d_buf || nbytes == 0), "uq_count_lines_end_async: null argument");
    ctx->idx_buf = nullptr; ctx->async_buf = nullptr; ctx->async_read = false;
    UQ_CHECK_HIP(hipMemsetAsync(ctx->d_async, 0, 64, ctx->stream));
    if (nbytes == 0) { ctx->h_pinned[8000] = 0; ctx->h_pinned[8001] = 0; ctx->h_pinned[8002] = 0; ctx->async_buf = d_buf; ctx->async_nbytes = 0; return 0; }
    const uint32_t mis = (uint32_t)((uintptr_t)d_buf & 15);
    const uint64_t nb = (((nbytes + mis + 15) / 16) * 16 + IDX_TILE - 1) / IDX_TILE;
    UQ_TRY(uq_scan_exclusive_u32(ctx, ctx->idx_partials, ctx->idx_partials, nb, (uint64_t*)ctx->d_async));
    ctx->async_buf = d_buf; ctx->async_nbytes = nbytes;
    return 0;
}

extern "C" int uq_index_lines_async(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t capacity_lines, uint64_t* d_line_start) {
    UQ_REQUIRE(ctx && d_line_start, "uq_index_lines_async: null argument");
    UQ_REQUIRE(ctx->async_buf == d_buf && ctx->async_nbytes == nbytes, "uq_index_lines_async: not the buffer of the last uq_count_lines_end_async");
    UQ_REQUIRE(capacity_lines < (uint64_t(1) << 32), "uq_index_lines_async: more than 2^32-1 lines in one shard");
    if (nbytes == 0) { UQ_CHECK_HIP(hipMemsetAsync(d_line_start, 0, 8, ctx->stream)); return 0; }
    const uint32_t mis = (uint32_t)((uintptr_t)d_buf & 15);
    const uint64_t nb = (((nbytes + mis + 15) / 16) * 16 + IDX_TILE - 1) / IDX_TILE;
    expand_list_kernel<<<(uint32_t)((nb + EXP_TILES * (IDX_THREADS / 64) - 1) / (EXP_TILES * (IDX_THREADS / 64))), IDX_THREADS, 0, ctx->stream>>>(ctx->idx_bitmap, mis, ctx->idx_partials,
                                                                                                                 nb, 0, d_line_start, ctx->d_async, capacity_lines,
                                                                                                                 (const uint32_t*)(ctx->idx_bitmap + nb * IDX_TILE_VECS));
    UQ_LAUNCH_CHECK();
    return 0;
}

// what uq_count_lines_wait hands out, sent on its way to the host (uq_pack_stats_async queues this behind its kernel: nothing small
// stands between the index and the pack kernel)
int uq_async_read_back(uq_ctx* ctx) {
    if (!ctx->async_buf || ctx->async_nbytes == 0 || ctx->async_read) return 0;
    const uint32_t mis = (uint32_t)((uintptr_t)ctx->async_buf & 15);
    const uint64_t nb = (((ctx->async_nbytes + mis + 15) / 16) * 16 + IDX_TILE - 1) / IDX_TILE;
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned + 8000, ctx->d_async, 16));
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned + 8002, ctx->idx_bitmap + nb * IDX_TILE_VECS, 4));
    ctx->async_read = true;
    return 0;
}

// *h_ok = 0: a tile held more newlines than its list (the index above is not usable: uq_count_lines + uq_index_lines take the bitmap
// form) or more lines than the capacity given to uq_index_lines_async.
extern "C" int uq_count_lines_wait(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t* h_nlines, int* h_ok) {
    UQ_REQUIRE(ctx && h_nlines && h_ok, "uq_count_lines_wait: null argument");
    UQ_REQUIRE(ctx->async_buf == d_buf && ctx->async_nbytes == nbytes, "uq_count_lines_wait: not the buffer of the last uq_count_lines_end_async");
    *h_nlines = 0; *h_ok = 1;
    if (nbytes == 0) { ctx->async_buf = nullptr; return 0; }
    UQ_TRY(uq_async_read_back(ctx));
    ctx->async_read = false;
    ctx->async_buf = nullptr;
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *h_nlines = ctx->h_pinned[8000];
    *h_ok = ctx->h_pinned[8001] == 0 && (uint32_t)ctx->h_pinned[8002] == 0;
    return 0;
}

extern "C" int uq_index_lines(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t nlines, uint64_t* d_line_start) {
    UQ_REQUIRE(ctx && d_line_start, "uq_index_lines: null argument");
    UQ_REQUIRE(nlines < (uint64_t(1) << 32), "uq_index_lines: more than 2^32-1 lines in one shard");
    if (nbytes == 0 || nlines == 0) {
        UQ_CHECK_HIP(hipMemsetAsync(d_line_start, 0, 8, ctx->stream));
        return 0;
    }
    uint64_t nb;
    const uint32_t mis = (uint32_t)((uintptr_t)d_buf & 15);
    const uint64_t nvec = (nbytes + mis + 15) / 16;
    if (ctx->idx_buf == d_buf && ctx->idx_nbytes == nbytes) {
        UQ_REQUIRE(ctx->idx_nlines == nlines, "uq_index_lines: nlines %llu does not match the census %llu",
                   (unsigned long long)nlines, (unsigned long long)ctx->idx_nlines);
        nb = (nvec * 16 + IDX_TILE - 1) / IDX_TILE;
        ctx->idx_buf = nullptr;
        expand_list_kernel<<<(uint32_t)((nb + EXP_TILES * (IDX_THREADS / 64) - 1) / (EXP_TILES * (IDX_THREADS / 64))), IDX_THREADS, 0, ctx->stream>>>(ctx->idx_bitmap, mis, ctx->idx_partials,
                                                                                                                     nb, nlines, d_line_start, nullptr, 0, nullptr);
        UQ_LAUNCH_CHECK();
        return 0;
    }
    UQ_TRY(run_count(ctx, d_buf, nbytes, &nb, false));
    UQ_TRY(uq_scan_exclusive_u32(ctx, ctx->idx_partials, ctx->idx_partials, nb, nullptr));
    ctx->idx_buf = nullptr;
    scatter_newlines_kernel<<<(uint32_t)nb, IDX_THREADS, 0, ctx->stream>>>((const uint64_t*)ctx->idx_bitmap, mis, ctx->idx_partials, nlines,
                                                                           d_line_start);
    UQ_LAUNCH_CHECK();
    return 0;
}
