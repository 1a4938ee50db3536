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
#include "common.h"
#include "swar.h"

namespace {
constexpr int IDX_THREADS = 256;
constexpr int IDX_LOADS = 4;                                  // 16-byte loads per lane
constexpr uint64_t IDX_WAVE_BYTES = 64 * 16 * IDX_LOADS;      // 4096
constexpr uint64_t IDX_TILE = IDX_WAVE_BYTES * (IDX_THREADS / 64);  // 16384
constexpr uint64_t IDX_TILE_VECS = IDX_TILE / 16;             // 1024 vectors = 256 bitmap words of 64 bits

// `abuf` = buf rounded down to 16 bytes, `mis` = buf - abuf.  Stream position of abuf[x] is x - mis.
// bitmap[vi] = newline mask of vector vi (0 beyond the stream; every vector of every tile is written).
__global__ __launch_bounds__(IDX_THREADS) void count_newlines_kernel(const uint4* __restrict__ abuf, uint32_t mis,
                                                                      uint64_t nbytes, uint64_t nvec,
                                                                      uint32_t* __restrict__ partials, uint16_t* __restrict__ bitmap) {
    __shared__ uint32_t lds[IDX_THREADS / 64];
    const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
    const uint64_t v0 = ((uint64_t)blockIdx.x * IDX_TILE + (uint64_t)w * IDX_WAVE_BYTES) / 16 + lane;
    uint32_t c = 0;
#pragma unroll
    for (int it = 0; it < IDX_LOADS; ++it) {
        uint64_t vi = v0 + (uint64_t)it * 64;
        uint32_t m = 0;
        if (vi < nvec) {
            uint4 q = abuf[vi];
            int64_t p = (int64_t)(vi * 16) - (int64_t)mis;
            m = nl_mask16(q) & valid_mask16(p, nbytes);
        }
        bitmap[vi] = (uint16_t)m;
        c += __popc(m);
    }
    c = wave_sum(c);
    if (lane == 0) lds[w] = c;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
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

int run_count(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t* nblocks_out) {
    const uint32_t mis = (uint32_t)((uintptr_t)d_buf & 15);
    const uint64_t nvec = (nbytes + mis + 15) / 16;
    const uint64_t nb = (nvec * 16 + IDX_TILE - 1) / IDX_TILE;
    UQ_REQUIRE(nb <= 0x7fffffffu, "uq_count_lines: buffer too large for one launch");
    if (nb + 1 > ctx->idx_partials_cap) {
        UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->idx_partials) UQ_CHECK_HIP(hipFree(ctx->idx_partials));
        if (ctx->idx_bitmap) UQ_CHECK_HIP(hipFree(ctx->idx_bitmap));
        ctx->idx_partials = nullptr; ctx->idx_bitmap = nullptr; ctx->idx_partials_cap = 0;
        UQ_CHECK_HIP(hipMalloc((void**)&ctx->idx_partials, (nb + 1) * sizeof(uint32_t)));
        UQ_CHECK_HIP(hipMalloc((void**)&ctx->idx_bitmap, nb * IDX_TILE_VECS * sizeof(uint16_t)));
        ctx->idx_partials_cap = nb + 1;
    }
    count_newlines_kernel<<<(uint32_t)nb, IDX_THREADS, 0, ctx->stream>>>((const uint4*)(d_buf - mis), mis, nbytes, nvec, ctx->idx_partials,
                                                                        ctx->idx_bitmap);
    UQ_LAUNCH_CHECK();
    *nblocks_out = nb;
    return 0;
}
}  // namespace

// For fused.hip: make sure ctx->idx_partials holds the exclusive scan of the per-tile newline counts
// of this buffer (reusing the census of uq_count_lines when `have_scanned`).
int uq_index_run_census(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t* nblocks_out, bool have_scanned) {
    const uint32_t mis = (uint32_t)((uintptr_t)d_buf & 15);
    const uint64_t nvec = (nbytes + mis + 15) / 16;
    if (have_scanned) { *nblocks_out = (nvec * 16 + IDX_TILE - 1) / IDX_TILE; return 0; }
    UQ_TRY(run_count(ctx, d_buf, nbytes, nblocks_out));
    UQ_TRY(uq_scan_exclusive_u32(ctx, ctx->idx_partials, ctx->idx_partials, *nblocks_out, nullptr));
    return 0;
}

extern "C" int uq_count_lines(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t* h_nlines) {
    UQ_REQUIRE(ctx && h_nlines, "uq_count_lines: null argument");
    ctx->idx_buf = nullptr;
    if (nbytes == 0) { *h_nlines = 0; return 0; }
    UQ_REQUIRE(d_buf, "uq_count_lines: null buffer");
    uint64_t nb;
    UQ_TRY(run_count(ctx, d_buf, nbytes, &nb));
    // exclusive scan of the per-tile counts right away: its total is the census, and uq_index_lines
    // reuses the scanned offsets (and the bitmap) for the same buffer
    void* scr;
    UQ_TRY(uq_scratch(ctx, 256, &scr));
    UQ_TRY(uq_scan_exclusive_u32(ctx, ctx->idx_partials, ctx->idx_partials, nb, (uint64_t*)scr));
    UQ_CHECK_HIP(hipMemcpyAsync(ctx->h_pinned, scr, 8, hipMemcpyDeviceToHost, ctx->stream));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *h_nlines = ctx->h_pinned[0];
    ctx->idx_buf = d_buf; ctx->idx_nbytes = nbytes; ctx->idx_nlines = *h_nlines;
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
    } else {
        UQ_TRY(run_count(ctx, d_buf, nbytes, &nb));
        UQ_TRY(uq_scan_exclusive_u32(ctx, ctx->idx_partials, ctx->idx_partials, nb, nullptr));
    }
    ctx->idx_buf = nullptr;
    scatter_newlines_kernel<<<(uint32_t)nb, IDX_THREADS, 0, ctx->stream>>>((const uint64_t*)ctx->idx_bitmap, mis, ctx->idx_partials, nlines,
                                                                           d_line_start);
    UQ_LAUNCH_CHECK();
    return 0;
}
