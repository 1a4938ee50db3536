// emit.hip -- FASTQ text assembled on the device.
// Replaces the decoder's per-read Python: the exec-compiled `convert_qname` (uq.py:1010-1026) and the
// four `print`s per read (uq.py:1042-1045 / 1055-1058).  Inputs are what the decode kernels already
// left in HBM: fixed-pitch sequence / quality text + lengths (uq_unpack) and the QNAME columns.
//   pass 1  one lane per read: bytes of its record (prefix + fields + separators + suffix + 2 L + 6)
//   scan    exclusive prefix sum -> record offsets, total size
//   pass 2  emit_tile_kernel: a workgroup assembles the text of R consecutive records in LDS (decimal digits of
//           integer columns are produced on the fly; mapping columns copy from a flattened string table) and
//           stores the span with aligned 16-byte vectors (11.1 ms -> see DESIGN.md for 10 M reads; the first
//           version wrote each record from one wave straight to HBM and is kept for oversize tiles)
// Algorithmic HBM bytes per read: 2 L + column bytes read, record bytes written.
#include "common.h"

namespace {
constexpr int EM_MAXCOLS = 32;

struct EmitGeom {
    uint8_t prefix[256];
    uint8_t suffix[256];
    uint8_t seps[EM_MAXCOLS];
    uint32_t prefix_len, suffix_len, ncols, dna_max;
    const void* col[EM_MAXCOLS];          // device column arrays (little-endian unsigned, itemsize bytes)
    uint32_t itemsize[EM_MAXCOLS];
    int64_t add[EM_MAXCOLS];              // value added to an integer column ('min' when offset, uq.py:1019)
    const uint8_t* map_chars[EM_MAXCOLS]; // mapping columns: flattened strings; null for integer columns
    const uint32_t* map_offs[EM_MAXCOLS]; // and their offsets [nmap + 1]
};

__device__ __forceinline__ uint64_t load_col(const void* p, uint32_t itemsize, uint64_t r) {
    switch (itemsize) {
        case 1: return ((const uint8_t*)p)[r];
        case 2: return ((const uint16_t*)p)[r];
        case 4: return ((const uint32_t*)p)[r];
        default: return ((const uint64_t*)p)[r];
    }
}
__device__ __forceinline__ uint32_t ndigits_u64(uint64_t v) {
    if (!(v >> 32)) {                    // the usual case: nine compares, no division
        const uint32_t w = (uint32_t)v;
        return 1u + (w >= 10u) + (w >= 100u) + (w >= 1000u) + (w >= 10000u) + (w >= 100000u) + (w >= 1000000u) + (w >= 10000000u) +
               (w >= 100000000u) + (w >= 1000000000u);
    }
    uint32_t n = 1;
    while (v >= 10) { v /= 10; ++n; }
    return n;
}
// text length of field c of read r; for integers also returns the magnitude and sign
__device__ __forceinline__ uint32_t field_from_raw(const EmitGeom& g, uint32_t c, uint64_t raw, uint64_t& mag, bool& neg, uint32_t& moff) {
    if (g.map_chars[c]) {
        moff = g.map_offs[c][raw];
        mag = 0; neg = false;
        return g.map_offs[c][raw + 1] - moff;
    }
    const int64_t v = (int64_t)raw + g.add[c];      // str(row[i] + min): columns narrower than 64 bit never wrap here
    neg = v < 0 && g.itemsize[c] < 8;               // a uint64 column without offset prints as unsigned
    mag = neg ? (uint64_t)(-v) : (uint64_t)v;
    if (g.itemsize[c] == 8 && g.add[c] == 0) { mag = raw; neg = false; }
    moff = 0;
    return ndigits_u64(mag) + (neg ? 1u : 0u);
}
__device__ __forceinline__ uint32_t field_len(const EmitGeom& g, uint32_t c, uint64_t r, uint64_t& mag, bool& neg, uint32_t& moff) {
    return field_from_raw(g, c, load_col(g.col[c], g.itemsize[c], r), mag, neg, moff);
}

__global__ void emit_sizes_kernel(EmitGeom g, const uint32_t* __restrict__ len, uint64_t n, uint64_t* __restrict__ sizes) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    uint64_t s = g.prefix_len + g.suffix_len + (g.ncols ? g.ncols - 1 : 0) + 2ull * len[r] + 5;   // 4 '\n' and the '+'
    for (uint32_t c = 0; c < g.ncols; ++c) {
        uint64_t mag; bool neg; uint32_t moff;
        s += field_len(g, c, r, mag, neg, moff);
    }
    sizes[r] = s;
}

// One wave writes one record straight to HBM: the fallback for tiles that do not fit the LDS buffer of
// emit_tile_kernel (very long reads).
__device__ void emit_record_direct(const EmitGeom& g, const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual,
                                   const uint32_t* __restrict__ len, const uint64_t* __restrict__ offsets, uint8_t* __restrict__ out,
                                   uint64_t r, uint32_t lane) {
    {
        uint8_t* o = out + offsets[r];
        const uint32_t L = len[r];
        // QNAME line: lane 0..ncols-1 render one field each (fields are short), lanes copy prefix / suffix
        uint32_t pos = g.prefix_len;
        for (uint32_t i = lane; i < g.prefix_len; i += 64) o[i] = g.prefix[i];
        for (uint32_t c = 0; c < g.ncols; ++c) {
            uint64_t mag; bool neg; uint32_t moff;
            const uint32_t fl = field_len(g, c, r, mag, neg, moff);     // wave-uniform (same r)
            if (g.map_chars[c]) {
                for (uint32_t i = lane; i < fl; i += 64) o[pos + i] = g.map_chars[c][moff + i];
            } else if (lane == 0) {
                uint32_t k = pos + fl;
                do { o[--k] = (uint8_t)('0' + mag % 10); mag /= 10; } while (mag);
                if (neg) o[--k] = '-';
            }
            pos += fl;
            if (c + 1 < g.ncols) { if (lane == 0) o[pos] = g.seps[c]; ++pos; }
        }
        for (uint32_t i = lane; i < g.suffix_len; i += 64) o[pos + i] = g.suffix[i];
        pos += g.suffix_len;
        if (lane == 0) o[pos] = '\n';
        ++pos;
        const uint8_t* s = seq + r * g.dna_max;
        const uint8_t* q = qual + r * g.dna_max;
        for (uint32_t i = lane; i < L; i += 64) { o[pos + i] = s[i]; o[pos + L + 3 + i] = q[i]; }
        if (lane == 0) { o[pos + L] = '\n'; o[pos + L + 1] = '+'; o[pos + L + 2] = '\n'; o[pos + 2 * L + 3] = '\n'; }
    }
}

// ---- the tile kernel: a workgroup assembles the text of R consecutive records in LDS (their bytes are one
// contiguous span of the output) and stores it with 16-byte coalesced vectors.
//   1  one lane per (record, QNAME field): text length of the field -> LDS; after a barrier the lane sums the
//      lengths before its field and renders it in place (decimal digits, or a copy from the string table);
//      prefix / suffix / separators / newlines by the same lanes
//   2  P lanes per record copy SEQ and QUAL: a lane owns destination-aligned dwords of the record's text and
//      fetches the matching 4 source bytes with one (unaligned) global load -- the fixed-pitch source rows and
//      the variable text offsets never share an alignment
//   3  the LDS image is stored to HBM: it mirrors the destination's 16-byte phase, so all interior stores are
//      aligned uint4
constexpr int EM_THREADS = 256;
constexpr uint32_t EM_CAP = 24 * 1024;
constexpr uint32_t EM_RMAX = 64;

__device__ __forceinline__ uint32_t load_u32_any(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

__global__ __launch_bounds__(EM_THREADS) void emit_tile_kernel(EmitGeom g, const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual,
                                                               const uint32_t* __restrict__ len, uint64_t n, const uint64_t* __restrict__ offsets,
                                                               uint8_t* __restrict__ out, uint32_t R, uint32_t P) {
    __shared__ __align__(16) uint8_t tile[EM_CAP + 32];
    __shared__ uint32_t rec_off[EM_RMAX + 1];            // record start inside `tile`
    __shared__ uint32_t qend[EM_RMAX];                   // end of the QNAME text (position of its '\n') relative to the record
    __shared__ uint16_t flen[EM_RMAX * EM_MAXCOLS];
    __shared__ unsigned long long s_off[EM_RMAX + 1];   // record offsets of the tile, as fetched
    __shared__ uint32_t s_len[EM_RMAX];
    const uint32_t tid = threadIdx.x, lane = lane_id();
    const uint64_t ntiles = (n + R - 1) / R;
    const uint32_t ncols = g.ncols;
    // Per-record metadata of the NEXT tile (offset, length, this lane's column value) is requested one tile ahead and
    // waits in registers: without it every tile paid three dependent global-memory latencies before the first byte.
    const bool one_item = R * ncols <= EM_THREADS;       // lane == (record, field) item; else the fields reload their values
    struct Pre { unsigned long long off; uint64_t raw; uint32_t L; };
    auto fetch = [&](uint64_t tt) {
        Pre x; x.off = 0; x.raw = 0; x.L = 0;
        if (tt >= ntiles) return x;
        const uint64_t r0 = tt * R;
        const uint32_t Rt = (uint32_t)((n - r0) < R ? (n - r0) : R);
        if (tid <= Rt) x.off = offsets[r0 + tid];
        if (tid < Rt) x.L = len[r0 + tid];
        if (one_item && tid < Rt * ncols) { const uint32_t i = tid / ncols, c = tid - i * ncols; x.raw = load_col(g.col[c], g.itemsize[c], r0 + i); }
        return x;
    };
    Pre nx = fetch(blockIdx.x);
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint64_t r0 = t * R;
        const uint32_t Rt = (uint32_t)((n - r0) < R ? (n - r0) : R);
        const Pre cur = nx;
        if (tid <= Rt) s_off[tid] = cur.off;
        if (tid < Rt) s_len[tid] = cur.L;
        __syncthreads();
        nx = fetch(t + gridDim.x);
        const uint64_t o0 = s_off[0], o1 = s_off[Rt];
        const uint32_t skew = (uint32_t)((uintptr_t)(out + o0) & 15);
        const uint64_t span = o1 - o0;
        if (span + skew > EM_CAP) {                      // does not fit: wave per record, straight to HBM
            for (uint32_t i = tid >> 6; i < Rt; i += EM_THREADS / 64) emit_record_direct(g, seq, qual, len, offsets, out, r0 + i, lane);
            __syncthreads();
            continue;
        }
        if (tid <= Rt) rec_off[tid] = (uint32_t)(cur.off - o0) + skew;
        // ---- 1a: field lengths
        uint64_t my_mag = 0; bool my_neg = false; uint32_t my_moff = 0, my_fl = 0;
        if (one_item) {
            if (tid < Rt * ncols) {
                const uint32_t i = tid / ncols, c = tid - i * ncols;
                my_fl = field_from_raw(g, c, cur.raw, my_mag, my_neg, my_moff);
                flen[i * EM_MAXCOLS + c] = (uint16_t)my_fl;
            }
        } else {
            for (uint32_t idx = tid; idx < Rt * ncols; idx += EM_THREADS) {
                const uint32_t i = idx / ncols, c = idx - i * ncols;
                uint64_t mag; bool neg; uint32_t moff;
                flen[i * EM_MAXCOLS + c] = (uint16_t)field_len(g, c, r0 + i, mag, neg, moff);
            }
        }
        __syncthreads();
        // ---- 1b: render the fields, the separators and (lane of the last field) the suffix + '\n'
        for (uint32_t idx = tid; idx < Rt * ncols; idx += EM_THREADS) {
            const uint32_t i = idx / ncols, c = idx - i * ncols;
            uint32_t pos = g.prefix_len;
            for (uint32_t k = 0; k < c; ++k) pos += flen[i * EM_MAXCOLS + k] + 1u;
            uint8_t* o = tile + rec_off[i] + pos;
            uint64_t mag = my_mag; bool neg = my_neg; uint32_t moff = my_moff;
            const uint32_t fl = one_item ? my_fl : field_len(g, c, r0 + i, mag, neg, moff);
            if (g.map_chars[c]) {
                for (uint32_t k = 0; k < fl; ++k) o[k] = g.map_chars[c][moff + k];
            } else {
                uint32_t k = fl;
                if (mag >> 32) { do { o[--k] = (uint8_t)('0' + mag % 10); mag /= 10; } while (mag); }
                else { uint32_t m = (uint32_t)mag; do { const uint32_t q = m / 10; o[--k] = (uint8_t)('0' + (m - q * 10)); m = q; } while (m); }
                if (neg) o[--k] = '-';
            }
            if (c + 1 < ncols) o[fl] = g.seps[c];
            else {
                for (uint32_t k = 0; k < g.suffix_len; ++k) o[fl + k] = g.suffix[k];
                o[fl + g.suffix_len] = '\n';
                qend[i] = pos + fl + g.suffix_len;
            }
        }
        for (uint32_t idx = tid; idx < Rt * g.prefix_len; idx += EM_THREADS) {
            const uint32_t i = idx / g.prefix_len, k = idx - i * g.prefix_len;
            tile[rec_off[i] + k] = g.prefix[k];
        }
        if (ncols == 0) {                                 // no columns: QNAME = prefix + suffix
            for (uint32_t i = tid; i < Rt; i += EM_THREADS) {
                uint8_t* o = tile + rec_off[i] + g.prefix_len;
                for (uint32_t k = 0; k < g.suffix_len; ++k) o[k] = g.suffix[k];
                o[g.suffix_len] = '\n';
                qend[i] = g.prefix_len + g.suffix_len;
            }
        }
        __syncthreads();
        // ---- 2: SEQ and QUAL text, P lanes per record
        {
            const uint32_t i = tid / P, p = tid - i * P;
            if (i < Rt) {
                const uint64_t r = r0 + i;
                const uint32_t L = s_len[i];
                const uint32_t ds = rec_off[i] + qend[i] + 1;           // first SEQ byte in the tile
                const uint32_t dq = ds + L + 3;                         // first QUAL byte
                if (p == 0) { tile[ds + L] = '\n'; tile[ds + L + 1] = '+'; tile[ds + L + 2] = '\n'; tile[dq + L] = '\n'; }
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    const uint8_t* src = (which ? qual : seq) + r * g.dna_max;
                    const uint32_t d = which ? dq : ds;
                    const uint32_t a = d & 3u;                          // text byte k sits at dword (d >> 2) + (k + a) / 4
                    const uint32_t nw = (a + L + 3) >> 2;
                    for (uint32_t wb = p; wb < nw; wb += 4 * P) {           // four loads in flight per lane before the first LDS store
                        uint32_t v[4];
                        bool full[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t w = wb + u * P;
                            const int32_t k0 = (int32_t)(4 * w) - (int32_t)a;   // source index of the dword's first byte
                            full[u] = w < nw && k0 >= 0 && (uint32_t)k0 + 4 <= L;
                            v[u] = full[u] ? load_u32_any(src + k0) : 0u;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t w = wb + u * P;
                            if (full[u]) *(uint32_t*)(tile + (d & ~3u) + 4 * w) = v[u];
                            else if (w < nw) {
                                const int32_t k0 = (int32_t)(4 * w) - (int32_t)a;
                                for (int b = 0; b < 4; ++b) {
                                    const int32_t k = k0 + b;
                                    if (k >= 0 && (uint32_t)k < L) tile[(d & ~3u) + 4 * w + b] = src[k];
                                }
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- 3: LDS image -> HBM (tile[skew ...] is out[o0 ...]; interior vectors are 16-byte aligned on both sides)
        {
            uint8_t* dst = out + o0 - skew;                            // 16-byte aligned
            const uint32_t endb = skew + (uint32_t)span;
            const uint32_t v0 = skew ? 1u : 0u, v1 = endb >> 4;         // full vectors [v0, v1)
            for (uint32_t v = v0 + tid; v < v1; v += EM_THREADS) ((uint4*)dst)[v] = ((const uint4*)tile)[v];
            if (skew) for (uint32_t b = skew + tid; b < 16 && b < endb; b += EM_THREADS) dst[b] = tile[b];
            if (v1 >= v0) for (uint32_t b = (v1 << 4) + tid; b < endb; b += EM_THREADS) dst[b] = tile[b];
        }
        __syncthreads();
    }
}
}  // namespace

extern "C" int uq_emit_fastq(uq_ctx* ctx, const uq_emit_params* hp, const void* const* h_d_cols, const uint8_t* const* h_d_map_chars,
                             const uint32_t* const* h_d_map_offs, const uint8_t* d_seq, const uint8_t* d_qual, const uint32_t* d_len,
                             uint64_t nreads, uint64_t* d_offsets, uint8_t* d_out, uint64_t capacity, uint64_t* h_total) {
    UQ_REQUIRE(ctx && hp && d_offsets && h_total, "uq_emit_fastq: null argument");
    UQ_REQUIRE(hp->ncols >= 0 && hp->ncols <= EM_MAXCOLS && hp->prefix_len >= 0 && hp->prefix_len <= 256 && hp->suffix_len >= 0 && hp->suffix_len <= 256,
               "uq_emit_fastq: QNAME layout out of range (<= 32 columns, prefix / suffix <= 256 bytes)");
    *h_total = 0;
    if (nreads == 0) return 0;
    UQ_REQUIRE(d_seq && d_qual && d_len && (hp->ncols == 0 || h_d_cols), "uq_emit_fastq: null buffer");
    EmitGeom g;
    memset(&g, 0, sizeof(g));
    memcpy(g.prefix, hp->prefix, 256); memcpy(g.suffix, hp->suffix, 256); memcpy(g.seps, hp->separators, EM_MAXCOLS);
    g.prefix_len = hp->prefix_len; g.suffix_len = hp->suffix_len; g.ncols = hp->ncols; g.dna_max = hp->dna_max;
    for (int c = 0; c < hp->ncols; ++c) {
        g.col[c] = h_d_cols[c]; g.itemsize[c] = hp->itemsize[c]; g.add[c] = hp->add[c];
        UQ_REQUIRE(g.itemsize[c] == 1 || g.itemsize[c] == 2 || g.itemsize[c] == 4 || g.itemsize[c] == 8, "uq_emit_fastq: bad column itemsize");
        g.map_chars[c] = h_d_map_chars ? h_d_map_chars[c] : nullptr;
        g.map_offs[c] = h_d_map_offs ? h_d_map_offs[c] : nullptr;
        UQ_REQUIRE((g.map_chars[c] == nullptr) == (g.map_offs[c] == nullptr), "uq_emit_fastq: mapping column needs both string tables");
    }
    emit_sizes_kernel<<<(uint32_t)((nreads + 255) / 256), 256, 0, ctx->stream>>>(g, d_len, nreads, d_offsets);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_scan_exclusive_u64(ctx, d_offsets, d_offsets, nreads, d_offsets + nreads));
    UQ_CHECK_HIP(hipMemcpyAsync(ctx->h_pinned, d_offsets + nreads, 8, hipMemcpyDeviceToHost, ctx->stream));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *h_total = ctx->h_pinned[0];
    if (!d_out) return 0;                       // size query
    UQ_REQUIRE(capacity >= *h_total, "uq_emit_fastq: output buffer too small (%llu < %llu)", (unsigned long long)capacity, (unsigned long long)*h_total);
    // tile = R records sized from the average record so that a typical tile fits the LDS image
    const uint64_t avg = *h_total / nreads + 1;
    uint64_t R = (EM_CAP - 64) / (avg + avg / 8 + 1);
    if (R > EM_RMAX) R = EM_RMAX;
    if (R < 1) R = 1;
    const uint32_t P = EM_THREADS / (uint32_t)R;
    const uint64_t tiles = (nreads + R - 1) / R;
    const uint64_t blocks = tiles < (uint64_t)UQ_NUM_CU * 6 ? tiles : (uint64_t)UQ_NUM_CU * 6;
    emit_tile_kernel<<<(uint32_t)blocks, EM_THREADS, 0, ctx->stream>>>(g, d_seq, d_qual, d_len, nreads, d_offsets, d_out, (uint32_t)R, P);
    UQ_LAUNCH_CHECK();
    return 0;
}
