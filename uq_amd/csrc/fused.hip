// fused.hip -- record index + pass-1 statistics in ONE pass over the FASTQ stream (rows "index" + a1).
// Replaces uq_index_lines followed by uq_stats_accumulate (uq.py:85, 366-375, 382, 388, 415-425): the
// stream is read once instead of twice.  The newline census (uq_count_lines) must have run on the same
// buffer: its scanned per-tile counts give every tile the global number of its first line.
//
// Persistent workgroups walk 16 KiB byte tiles (the census' tiles) with a register-prefetch pipeline; a
// tile is staged in LDS together with a 4 KiB forward halo.  Each lane then owns 80 consecutive staged
// bytes: exact SWAR newline masks, one block scan (main-part and total counts packed in one word), and
// the sorted list of newline offsets of tile + halo lands in LDS.  From that list
//   * the tile's own newlines are written to line_start (their global ranks are base + local rank);
//   * every record whose QNAME line ends in the tile is counted: its SEQ / '+' / QUAL lines are the
//     next three entries of the list, so P lanes per record histogram (base, quality) pairs straight
//     out of LDS with the lookup-free windowed table of stats.hip.
// A record that runs past the halo (reads longer than ~2 kbp) is counted from HBM by one lane group;
// a tile with more than FZ_NLMAX newlines raises `overflow` and the host falls back to the two-pass
// form.  max_record_bytes is filled by a small follow-up kernel over line_start.
// Algorithmic HBM bytes: nbytes read + 8 B per line written.
#include "common.h"
#include "swar.h"

namespace {
constexpr int FZ_THREADS = 256;
constexpr int FZ_NV = 5;                               // 16-byte chunks per lane: 4 main + 1 halo per 256 lanes
constexpr uint32_t FZ_MAIN = 16384, FZ_STAGE = FZ_NV * FZ_THREADS * 16;   // 20480
constexpr uint32_t FZ_NLMAX = 2048;
constexpr uint32_t FZ_BBASE = 64, FZ_NB = 64, FZ_NQ = 64;
constexpr uint32_t FZ_P = 5;                           // lanes per record
constexpr uint32_t FZ_RPB = FZ_THREADS / FZ_P;         // records per batch

__device__ __forceinline__ void fz_count_pair(uint32_t b, uint32_t c, uint32_t qbase, uint32_t* hist, uq_stats* st) {
    const uint32_t sb = b - FZ_BBASE, sq = c - qbase;
    if (sb < FZ_NB && sq < FZ_NQ) atomicAdd(&hist[sb * FZ_NQ + sq], 1u);
    else atomicAdd((unsigned long long*)&st->counts[b * 256 + c], 1ull);
}

__global__ __launch_bounds__(FZ_THREADS) void index_stats_kernel(const uint4* __restrict__ abuf, uint32_t mis, uint64_t nbytes,
                                                                  uint64_t nvec, uint64_t ntiles, const uint32_t* __restrict__ offsets,
                                                                  uint64_t nlines, uint64_t* __restrict__ line_start, uint32_t qbase,
                                                                  uq_stats* __restrict__ st, uint32_t* __restrict__ overflow) {
    __shared__ uint32_t hist[FZ_NB * FZ_NQ];
    __shared__ __align__(16) uint8_t stage[FZ_STAGE + 32];
    __shared__ uint16_t nlpos[FZ_NLMAX + 8];
    __shared__ uint32_t s_scan[FZ_THREADS / 64 + 1];
    const uint32_t tid = threadIdx.x, lane = lane_id();
    for (int i = tid; i < (int)(FZ_NB * FZ_NQ); i += FZ_THREADS) hist[i] = 0;
    const uint8_t* bytes = (const uint8_t*)abuf + mis;          // stream position 0
    const uint32_t q_addlo = 0x01010101u * (0x80u - qbase), q_addhi = 0x01010101u * (0x80u - qbase - FZ_NQ);
    uint32_t lmin = 0xFFFFFFFFu, lmax = 0;
    uint64_t bad_plus = UQ_NONE, bad_len = UQ_NONE;
    const uint64_t S = gridDim.x;
    __syncthreads();

    struct Regs { uint4 v[FZ_NV]; uint32_t base; };
    auto issue = [&](uint64_t tt) {
        Regs x;
        x.base = 0;
#pragma unroll
        for (int u = 0; u < FZ_NV; ++u) x.v[u] = make_uint4(0, 0, 0, 0);
        if (tt >= ntiles) return x;
        const uint64_t v0 = tt * (FZ_MAIN / 16);
#pragma unroll
        for (int u = 0; u < FZ_NV; ++u) { const uint64_t vi = v0 + (uint64_t)u * FZ_THREADS + tid; if (vi < nvec) x.v[u] = abuf[vi]; }
        x.base = offsets[tt];
        return x;
    };

    uint64_t t = blockIdx.x;
    Regs cur = issue(t);
    for (; t < ntiles; t += S) {
        const uint64_t v0 = t * (FZ_MAIN / 16);                 // first chunk of the tile (absolute chunk index)
        const int64_t pos0 = (int64_t)(v0 * 16) - (int64_t)mis; // stream position of staged byte 0
        const uint32_t G0 = cur.base;                           // global index of the tile's first newline
        // ---- A: registers -> LDS
#pragma unroll
        for (int u = 0; u < FZ_NV; ++u) ((uint4*)stage)[u * FZ_THREADS + tid] = cur.v[u];
        __syncthreads();
        cur = issue(t + S);                                     // next tile's loads fly during B..D
        // ---- B: lane = chunks 5*tid .. 5*tid+4 (80 consecutive bytes): newline masks, block scan, offset list
        uint32_t m[FZ_NV];
        uint32_t cnt_all = 0, cnt_main = 0;
#pragma unroll
        for (int u = 0; u < FZ_NV; ++u) {
            const uint32_t c = tid * FZ_NV + u;
            const uint4 q = ((const uint4*)stage)[c];
            const uint64_t vi = v0 + c;
            uint32_t mm = 0;
            if (vi < nvec) mm = nl_mask16(q) & valid_mask16(pos0 + (int64_t)c * 16, nbytes);
            m[u] = mm;
            const uint32_t k = __popc(mm);
            cnt_all += k;
            if (c < FZ_MAIN / 16) cnt_main += k;
        }
        uint32_t tot;
        const uint32_t ex = block_exclusive_sum<uint32_t, FZ_THREADS / 64>(cnt_all | (cnt_main << 16), s_scan, tot);
        const uint32_t ntot = tot & 0xFFFFu, nmain = tot >> 16;
        if (ntot > FZ_NLMAX) {                                  // pathological tile (e.g. blank lines): two-pass fallback
            if (tid == 0) atomicExch(overflow, 1u);
            __syncthreads();
            continue;
        }
        {
            uint32_t k = ex & 0xFFFFu;
#pragma unroll
            for (int u = 0; u < FZ_NV; ++u) {
                uint32_t mm = m[u];
                const uint32_t cb = (tid * FZ_NV + u) * 16;
                while (mm) {
                    const int b = __ffs(mm) - 1;
                    mm &= mm - 1;
                    nlpos[k++] = (uint16_t)(cb + b);
                }
            }
        }
        __syncthreads();
        // ---- C: this tile's newlines -> line_start
        if (t == 0 && tid == 0) line_start[0] = 0;
        for (uint32_t j = tid; j < nmain; j += FZ_THREADS) {
            const uint64_t k = (uint64_t)G0 + j + 1;
            if (k <= nlines) line_start[k] = (uint64_t)(pos0 + nlpos[j] + 1);
        }
        // ---- D: records whose QNAME line ends in this tile: newline j with (G0 + j) % 4 == 0
        const uint32_t j0 = (4u - (G0 & 3u)) & 3u;
        const uint32_t nrec = nmain > j0 ? (nmain - j0 + 3) / 4 : 0;
        uint32_t rr, pp;
        rr = tid / FZ_P; pp = tid - rr * FZ_P;
        for (uint32_t rb = 0; rb < nrec; rb += FZ_RPB) {
            const uint32_t ri = rb + rr;
            if (rr >= FZ_RPB || ri >= nrec) continue;
            const uint32_t j = j0 + 4 * ri;
            const uint64_t grec = ((uint64_t)G0 + j) >> 2;      // record index in the buffer
            if ((uint64_t)G0 + j + 3 >= nlines) continue;       // truncated last record: the census check reports it
            if (j + 3 < ntot) {
                const uint32_t s = nlpos[j] + 1u, e1 = nlpos[j + 1], q = nlpos[j + 2] + 1u, e2 = nlpos[j + 3];
                const uint32_t L = e1 - s, Lq = e2 - q;
                if (pp == 0) {
                    if (stage[e1 + 1] != '+') bad_plus = bad_plus < grec ? bad_plus : grec;
                    if (L != Lq) bad_len = bad_len < grec ? bad_len : grec;
                    lmin = L < lmin ? L : lmin;
                    lmax = L > lmax ? L : lmax;
                }
                const uint32_t Lc = L < Lq ? L : Lq;
                for (uint32_t jj = 8 * pp; jj < Lc; jj += 8 * FZ_P) {
                    uint32_t b_lo, b_hi, q_lo, q_hi;
                    lds_window8(stage, (int32_t)(s + jj), b_lo, b_hi);
                    lds_window8(stage, (int32_t)(q + jj), q_lo, q_hi);
                    const uint32_t cnt = Lc - jj;
                    const uint32_t sb0 = b_lo ^ 0x40404040u, sb1 = b_hi ^ 0x40404040u;
                    const uint32_t u0 = q_lo + q_addlo, u1 = q_hi + q_addlo;
                    const uint32_t bad = ((sb0 | sb1) & 0xC0C0C0C0u) |
                                         ((q_lo | (q_lo + q_addhi) | ~u0 | q_hi | (q_hi + q_addhi) | ~u1) & 0x80808080u);
                    if (cnt >= 8 && bad == 0) {
                        const uint32_t sq0 = (u0 & 0x7F7F7F7Fu) << 2, sq1 = (u1 & 0x7F7F7F7Fu) << 2;
                        uint8_t* hb = (uint8_t*)hist;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            atomicAdd((uint32_t*)(hb + ((((sb0 >> (8 * k)) & 0xFFu) << 8) | ((sq0 >> (8 * k)) & 0xFFu))), 1u);
                            atomicAdd((uint32_t*)(hb + ((((sb1 >> (8 * k)) & 0xFFu) << 8) | ((sq1 >> (8 * k)) & 0xFFu))), 1u);
                        }
                    } else {
                        const uint32_t w[4] = {b_lo, b_hi, q_lo, q_hi};
                        for (uint32_t k = 0; k < 8 && k < cnt; ++k)
                            fz_count_pair((w[k >> 2] >> (8 * (k & 3))) & 255u, (w[2 + (k >> 2)] >> (8 * (k & 3))) & 255u, qbase, hist, st);
                    }
                }
            } else {
                // the record runs past the staged halo: find its lines in HBM (one lane group, byte loads)
                const uint64_t s = (uint64_t)(pos0 + nlpos[j] + 1);
                uint64_t e[3] = {0, 0, 0};
                int found = 0;
                if (pp == 0)
                    for (uint64_t p = s; p < nbytes && found < 3; ++p)
                        if (bytes[p] == '\n') e[found++] = p;
                if (pp == 0 && found == 3) {
                    const uint64_t e1 = e[0], q = e[1] + 1, e2 = e[2];
                    const uint32_t L = (uint32_t)(e1 - s), Lq = (uint32_t)(e2 - q);
                    if (bytes[e1 + 1] != '+') bad_plus = bad_plus < grec ? bad_plus : grec;
                    if (L != Lq) bad_len = bad_len < grec ? bad_len : grec;
                    lmin = L < lmin ? L : lmin;
                    lmax = L > lmax ? L : lmax;
                    const uint32_t Lc = L < Lq ? L : Lq;
                    for (uint32_t jj = 0; jj < Lc; ++jj) fz_count_pair(bytes[s + jj], bytes[q + jj], qbase, hist, st);
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    for (int i = tid; i < (int)(FZ_NB * FZ_NQ); i += FZ_THREADS) {
        const uint32_t v = hist[i];
        if (v) atomicAdd((unsigned long long*)&st->counts[(FZ_BBASE + i / FZ_NQ) * 256 + qbase + (i % FZ_NQ)], (unsigned long long)v);
    }
    lmin = wave_min(lmin); lmax = wave_max(lmax);
    bad_plus = wave_min(bad_plus); bad_len = wave_min(bad_len);
    if (lane == 0) {
        if (lmin != 0xFFFFFFFFu) atomicMin(&st->len_min, lmin);
        atomicMax(&st->len_max, lmax);
        if (bad_plus != UQ_NONE) atomicMin((unsigned long long*)&st->bad_plus, (unsigned long long)bad_plus);
        if (bad_len != UQ_NONE) atomicMin((unsigned long long*)&st->bad_len, (unsigned long long)bad_len);
    }
}

__global__ void record_span_max_kernel(const uint64_t* __restrict__ ls, uint64_t nreads, uq_stats* __restrict__ st) {
    uint32_t mx = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(ls[4 * r + 4] - ls[4 * r]);
        mx = b > mx ? b : mx;
    }
    mx = wave_max(mx);
    if (lane_id() == 0 && mx) atomicMax(&st->max_record_bytes, mx);
}
}  // namespace

int uq_index_run_census(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t* nblocks_out, bool scanned);

extern "C" int uq_index_stats(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t nlines, uint64_t* d_line_start,
                              uq_stats* d_stats, int* h_fused) {
    UQ_REQUIRE(ctx && d_line_start && d_stats && h_fused, "uq_index_stats: null argument");
    UQ_REQUIRE(nlines < (uint64_t(1) << 32), "uq_index_stats: more than 2^32-1 lines in one shard");
    *h_fused = 0;
    if (nbytes == 0 || nlines == 0 || nlines % 4 != 0) return 0;      // caller uses the two-pass entry points
    UQ_REQUIRE(d_buf, "uq_index_stats: null buffer");
    uint64_t nb = 0;
    const bool cached = ctx->idx_buf == d_buf && ctx->idx_nbytes == nbytes && ctx->idx_nlines == nlines;
    UQ_TRY(uq_index_run_census(ctx, d_buf, nbytes, &nb, cached));
    ctx->idx_buf = nullptr;
    const uint32_t mis = (uint32_t)((uintptr_t)d_buf & 15);
    const uint64_t nvec = (nbytes + mis + 15) / 16;
    // the quality window of the LDS table: peek at the first record (its 4th line starts after the 3rd newline)
    uint32_t qbase = 33;
    {
        uint8_t* h = (uint8_t*)ctx->h_pinned;
        const uint64_t n = nbytes < 16384 ? nbytes : 16384;
        UQ_CHECK_HIP(hipMemcpyAsync(h, d_buf, n, hipMemcpyDeviceToHost, ctx->stream));
        UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        uint64_t p = 0; int nl = 0;
        while (p < n && nl < 3) { if (h[p] == '\n') ++nl; ++p; }
        uint32_t mn = 255;
        for (; p < n && h[p] != '\n'; ++p) if (h[p] < mn) mn = h[p];
        if (mn != 255) { if (mn >= 64) qbase = 59; else if (mn < 33) qbase = 0; }
    }
    void* scr;
    UQ_TRY(uq_scratch(ctx, 256, &scr));
    uint32_t* d_over = (uint32_t*)scr;
    UQ_CHECK_HIP(hipMemsetAsync(d_over, 0, 4, ctx->stream));
    const uint32_t blocks = (uint32_t)(nb < (uint64_t)UQ_NUM_CU * 3 ? nb : (uint64_t)UQ_NUM_CU * 3);
    index_stats_kernel<<<blocks, FZ_THREADS, 0, ctx->stream>>>((const uint4*)(d_buf - mis), mis, nbytes, nvec, nb, ctx->idx_partials,
                                                               nlines, d_line_start, qbase, d_stats, d_over);
    UQ_LAUNCH_CHECK();
    record_span_max_kernel<<<UQ_NUM_CU * 2, 256, 0, ctx->stream>>>(d_line_start, nlines / 4, d_stats);
    UQ_LAUNCH_CHECK();
    UQ_CHECK_HIP(hipMemcpyAsync(ctx->h_pinned, d_over, 4, hipMemcpyDeviceToHost, ctx->stream));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *h_fused = ((uint32_t*)ctx->h_pinned)[0] == 0 ? 1 : 0;
    return 0;
}
