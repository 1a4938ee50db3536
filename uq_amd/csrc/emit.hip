// emit.hip -- FASTQ text assembled on the device.
// Replaces the decoder's per-read Python: the exec-compiled `convert_qname` (uq.py:1010-1026) and the
// four `print`s per read (uq.py:1042-1045 / 1055-1058).  Inputs are what the decode kernels already
// left in HBM: fixed-pitch sequence / quality text + lengths (uq_unpack) and the QNAME columns.
//   pass 1  one lane per read: bytes of its record (prefix + fields + separators + suffix + 2 L + 6)
//   scan    exclusive prefix sum -> record offsets, total size
//   pass 2  one wave per read: lanes write the record's bytes (decimal digits of integer columns are
//           produced on the fly; mapping columns copy from a flattened string table)
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
    uint32_t n = 1;
    while (v >= 10) { v /= 10; ++n; }
    return n;
}
// text length of field c of read r; for integers also returns the magnitude and sign
__device__ __forceinline__ uint32_t field_len(const EmitGeom& g, uint32_t c, uint64_t r, uint64_t& mag, bool& neg, uint32_t& moff) {
    const uint64_t raw = load_col(g.col[c], g.itemsize[c], r);
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

__global__ __launch_bounds__(256) void emit_write_kernel(EmitGeom g, const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual,
                                                         const uint32_t* __restrict__ len, uint64_t n, const uint64_t* __restrict__ offsets,
                                                         uint8_t* __restrict__ out) {
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t GW = (uint64_t)gridDim.x * 4;
    const uint32_t lane = lane_id();
    for (uint64_t r = gw; r < n; r += GW) {
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
    uint64_t blocks = (nreads + 3) / 4;
    if (blocks > UQ_NUM_CU * 16) blocks = UQ_NUM_CU * 16;
    emit_write_kernel<<<(uint32_t)blocks, 256, 0, ctx->stream>>>(g, d_seq, d_qual, d_len, nreads, d_offsets, d_out);
    UQ_LAUNCH_CHECK();
    return 0;
}
