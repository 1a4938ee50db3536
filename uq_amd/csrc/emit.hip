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
#include "swar.h"
#include "codes.h"
#include <type_traits>

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

// The column / string-table pointers reach the kernels inside a struct (kept in LDS): generic pointers, whose loads would be FLAT -- counted in lgkmcnt as
// well, returning out of order, so that the next wait of any kind becomes `vmcnt(0) lgkmcnt(0)`.  They point to device memory: said so, the loads are global.
template <typename T> using gptr = const __attribute__((address_space(1))) T*;
template <typename T> __device__ __forceinline__ gptr<T> as_global(const void* p) { return (gptr<T>)(uintptr_t)p; }
__device__ __forceinline__ uint64_t load_col(const void* p, uint32_t itemsize, uint64_t r) {
    switch (itemsize) {
        case 1: return as_global<uint8_t>(p)[r];
        case 2: return as_global<uint16_t>(p)[r];
        case 4: return as_global<uint32_t>(p)[r];
        default: return as_global<uint64_t>(p)[r];
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
        // add[c] of a mapping column = the number of its strings (0: not given): a stored code beyond the table reads its last
        // string instead of whatever lies behind the offsets (callers that want the error ask uq_check_index_range first)
        const uint64_t nmap = (uint64_t)g.add[c];
        if (nmap && raw >= nmap) raw = nmap - 1;
        const gptr<uint32_t> mo = as_global<uint32_t>(g.map_offs[c]);
        moff = mo[raw];
        mag = 0; neg = false;
        return mo[raw + 1] - moff;
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
// The tile kernels request the next tile's rows right behind their first barrier and want them in flight through the whole of phase B.  vmcnt counts every
// vector-memory operation in issue order, and the compiler merges "a load is pending" over all paths: a load in a COLD branch of phase B (a column value that
// was not prefetched, a mapping column's characters) left every later reuse of its registers behind an unconditional `s_waitcnt vmcnt(0)` -- which, at run
// time, waited for the rows just requested.  The cold loads therefore live in functions of their own: a call waits for everything on both sides, but only
// where it is executed.
struct FieldCold { uint64_t mag; uint32_t fl, moff, neg; };          // (returned in registers: out-parameters by address would put the hot path's copies on the stack)
__device__ __noinline__ FieldCold field_len_cold(const EmitGeom* g, uint32_t c, uint64_t r) {
    FieldCold f; bool neg;
    f.fl = field_len(*g, c, r, f.mag, neg, f.moff); f.neg = neg ? 1u : 0u;
    return f;
}
__device__ __noinline__ void copy_chars_cold(uint8_t* o, const uint8_t* src, uint32_t n) {      // (o: an LDS address, passed as a generic pointer)
    const gptr<uint8_t> gs = as_global<uint8_t>(src);
    for (uint32_t k = 0; k < n; ++k) o[k] = gs[k];
}

__global__ void emit_sizes_kernel(EmitGeom g, const uint32_t* __restrict__ len, uint64_t n, uint64_t* __restrict__ sizes) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    uint64_t s = g.prefix_len + g.suffix_len + (g.ncols ? g.ncols - 1 : 0) + 2ull * (len ? len[r] : g.dna_max) + 5;   // 4 '\n' and the '+'
    for (uint32_t c = 0; c < g.ncols; ++c) {
        uint64_t mag; bool neg; uint32_t moff;
        s += field_len(g, c, r, mag, neg, moff);
    }
    sizes[r] = s;
}

// One wave writes one record straight to HBM: the fallback for tiles that do not fit the LDS buffer of
// emit_tile_kernel (very long reads).  emit_qname_direct writes the QNAME line and returns its length.
__device__ uint32_t emit_qname_direct(const EmitGeom& g, uint8_t* __restrict__ o, uint64_t r, uint32_t lane) {
    // lane 0 renders the integer fields (they are short), lanes copy prefix / suffix / mapping strings
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
    return pos + 1;
}

__device__ void emit_record_direct(const EmitGeom& g, const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual,
                                   const uint32_t* __restrict__ len, const uint64_t* __restrict__ offsets, uint8_t* __restrict__ out,
                                   uint64_t r, uint32_t lane) {
    uint8_t* o = out + offsets[r];
    const uint32_t L = len[r];
    const uint32_t pos = emit_qname_direct(g, o, r, lane);
    const uint8_t* s = seq + r * g.dna_max;
    const uint8_t* q = qual + r * g.dna_max;
    for (uint32_t i = lane; i < L; i += 64) { o[pos + i] = s[i]; o[pos + L + 3 + i] = q[i]; }
    if (lane == 0) { o[pos + L] = '\n'; o[pos + L + 1] = '+'; o[pos + L + 2] = '\n'; o[pos + 2 * L + 3] = '\n'; }
}

// ---- the tile kernel: a workgroup assembles the text of R consecutive records in LDS (their bytes are one
// contiguous span of the output) and stores it with 16-byte coalesced vectors.
//   1  one lane per (record, QNAME field): text length of the field -> LDS; after a barrier the lane sums the
//      lengths before its field and renders it in place (decimal digits, or a copy from the string table);
//      prefix / suffix / separators / newlines by the same lanes
//   2  SEQ and QUAL text.  Two sources:
//      text form   (uq_emit_fastq)    P lanes per record copy from the fixed-pitch character arrays uq_unpack left: a
//                  lane owns destination-aligned dwords and fetches the matching 4 source bytes with one unaligned load
//      packed form (uq_decode_fastq)  the records' packed DNA / QUAL rows -- prefetched into registers one tile ahead,
//                  like the metadata -- are decoded straight into the image, a lane per 8 symbols (the arithmetic of
//                  unpack_pipe_kernel).  The 2 x dna_max bytes per read of intermediate text never exist.
//   3  the LDS image is stored to HBM: it mirrors the destination's 16-byte phase, so all interior stores are
//      aligned uint4
constexpr int EM_THREADS = 256;
constexpr uint32_t EM_RMAX = 64;
constexpr int DE_K = 5;                  // chunks per lane of the fixed-length instances
#ifndef DS_K_
#define DS_K_ 2
#endif
#ifndef DS_OCC
#define DS_OCC 4
#endif
constexpr int DS_K = DS_K_;              // groups per lane of decode_stream_kernel's instances
constexpr int DE_NVD = 2, DE_NVQ = 3;    // 16-byte vectors per lane of packed rows in flight: DNA rows <= 8 KiB, QUAL rows <= 12 KiB per tile
constexpr uint32_t EM_BUDGET_TEXT = 19 * 1024, EM_BUDGET_PACKED = 36 * 1024;   // dynamic LDS per workgroup (the registers allow four workgroups per CU)

struct TileGeom {
    uint32_t R, P, cap;                                          // records per tile, lanes per record (text form), bytes of the text image
    uint32_t o_off, o_len, o_flen, o_ind, o_inq;                 // LDS byte offsets behind the image
    uint32_t magicP;                                             // magic_u32(prefix_len)
    uint32_t bd, bq, Cd, Cq, G, magicG;                          // packed form: row geometry, G = 8-symbol groups per read
    uint32_t variable, NC, magicNC, o_cum, RS;                   // fixed length: NC = lanes per line (the chunks a line can touch, or pieces of K chunks), RS = reads a workgroup-step covers (EM_THREADS / NC)
    uint32_t K;                                                  // chunks per lane (emit_tile_kernel<.., K>)
    FastAlphabet fa;
};
struct NoLut {};

__device__ __forceinline__ uint32_t load_u32_any(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// ---- packed form, fast alphabet: eight symbols per lane without per-symbol work
// byte k of the result = bits (7 - 2k .. 6 - 2k) of x: the four 2-bit base codes of one row byte, in text order
__device__ __forceinline__ uint32_t spread2(uint32_t x) { return ((x >> 6) | (x << 4) | (x << 14) | (x << 24)) & 0x03030303u; }
// 2-bit alphabets with the N-trick: where the quality code (one per byte, below 128) is the N-trick's, the base's selector becomes 4 + code -- and
// base_tab_hi, unused by four bases, holds the N character four times (fast_alphabet): the N replaces the base inside the v_perm that looks the
// bases up (five instructions per four characters; comparing, widening the flags to byte masks and blending took ten)
__device__ __forceinline__ uint32_t n_selectors(uint32_t codes, uint32_t q, uint32_t n_code4) { return codes | ((~((q ^ n_code4) + 0x7F7F7F7Fu) & 0x80808080u) >> 5); }
// ---- LDS access on gfx950 (tools/ldsbench.hip, profiles/r02_l_ldsbench.txt): a wave-instruction at its natural alignment
// takes 3 - 7 cycles; the same instruction at a misaligned address takes 40 (reads) or 128 (writes) -- the lanes go one
// by one.  So the decoder works from the DESTINATION: a lane produces one 8-byte-aligned chunk of the text image, and
// pulls the bits of its eight symbols out of the packed rows through aligned dwords.
// (32-bit multiplies and 64-bit shifts by a variable run at a quarter of the rate of the other integer instructions: the
// index arithmetic below uses the 24-bit multiply, v_perm and v_alignbit instead)
// selector of v_perm_b32 that takes the four bytes at + 3 .. at (in this order: a big-endian number) out of the two aligned dwords around `at`
__device__ __forceinline__ uint32_t be_selector(uint32_t at) { return 0x00010203u + __builtin_amdgcn_perm(0u, at & 3u, 0u); }   // byte 0 of (at & 3) in all four
// big-endian number of the four bytes tile[at .. at + 3] (any alignment), from the two aligned dwords around them
__device__ __forceinline__ uint32_t be32_at(const uint8_t* tile, uint32_t at) {
    const uint32_t* q = (const uint32_t*)(tile + (at & ~3u));
    return __builtin_amdgcn_perm(q[1], q[0], be_selector(at));
}
// 2-bit codes of the symbols t0 + 7 .. t0 (t = index from the END of the read; text order = descending t) of the row whose
// last byte is tile[end]: byte k of (lo, hi) = the code of text byte k of the chunk.  t0 may be negative or reach beyond the
// read at the chunks on a line's edges: those bytes are never stored, and the addresses stay inside the rows' carve.
__device__ __forceinline__ void dna_codes8(const uint8_t* tile, uint32_t end, int32_t t0, uint32_t& lo, uint32_t& hi) {
    const int32_t b0 = 2 * t0;
    const uint32_t w = be32_at(tile, (uint32_t)((int32_t)end - 3 - (b0 >> 3))) >> (b0 & 7);
    lo = spread2((w >> 8) & 0xFFu); hi = spread2(w & 0xFFu);
}
// the same for BQ-bit quality codes
template <int BQ>
__device__ __forceinline__ void qual_codes8(const uint8_t* tile, uint32_t end, int32_t t0, uint32_t& qlo, uint32_t& qhi) {
    const int32_t b0 = __mul24(BQ, t0);
    const uint32_t at = (uint32_t)((int32_t)end - 7 - (b0 >> 3));
    const uint32_t* q = (const uint32_t*)(tile + (at & ~3u));
    const uint32_t sel = be_selector(at);
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    const uint32_t whi = __builtin_amdgcn_perm(d1, d0, sel), wlo = __builtin_amdgcn_perm(d2, d1, sel), sh = (uint32_t)b0 & 7u;
    const uint64_t V = ((uint64_t)(whi >> sh) << 32) | __builtin_amdgcn_alignbit(whi, wlo, sh);       // (whi : wlo) >> sh
    const uint32_t up = (uint32_t)(V >> (4 * BQ)) & ((1u << (4 * BQ)) - 1u), dn = (uint32_t)V & ((1u << (4 * BQ)) - 1u);       // 4 BQ <= 28 bits each
    constexpr uint32_t M = (1u << BQ) - 1u;
    qlo = ((up >> (3 * BQ)) & M) | (((up >> (2 * BQ)) & M) << 8) | (((up >> BQ) & M) << 16) | ((up & M) << 24);
    qhi = ((dn >> (3 * BQ)) & M) | (((dn >> (2 * BQ)) & M) << 8) | (((dn >> BQ) & M) << 16) | ((dn & M) << 24);
}
__device__ __forceinline__ void qual_codes8(uint32_t bq, const uint8_t* tile, uint32_t end, int32_t t0, uint32_t& qlo, uint32_t& qhi) {
    switch (bq) {
        case 1: qual_codes8<1>(tile, end, t0, qlo, qhi); break;
        case 2: qual_codes8<2>(tile, end, t0, qlo, qhi); break;
        case 3: qual_codes8<3>(tile, end, t0, qlo, qhi); break;
        case 4: qual_codes8<4>(tile, end, t0, qlo, qhi); break;
        case 5: qual_codes8<5>(tile, end, t0, qlo, qhi); break;
        case 6: qual_codes8<6>(tile, end, t0, qlo, qhi); break;
        default: qual_codes8<7>(tile, end, t0, qlo, qhi); break;
    }
}
// ---- K chunks (8 K symbols) per lane: one window, one selector and one shift for all of them.  A lane that owns eight characters pays
// ~ 25 instructions of addressing / alignment per line on top of ~ 35 of field extraction; with K = 5 the first part is paid once per
// forty characters (emit_tile_kernel<.., K>: 10 M x 150 bp 1.50 -> see DESIGN 10).
// B-bit codes of the symbols t0 + 8 K - 1 .. t0 of the row whose last byte is tile[end], as K chunks in TEXT order: byte k of (lo[c], hi[c]) =
// the code of text byte k of chunk c.  Reads up to 4 * ND + 3 bytes in front of the window's last byte: inside the rows' carve (plan_tile's slack).
// ALIGNED: t0 is a multiple of 8 (the window ends on a byte: no shift).
template <int B, int K, bool ALIGNED = false>
__device__ __forceinline__ void codes_piece(const uint8_t* tile, uint32_t end, int32_t t0, uint32_t (&lo)[K], uint32_t (&hi)[K]) {
    constexpr int TB = 8 * K * B, NB = TB / 8 + 1, ND = (NB + 3) / 4, NV = (TB + 31) / 32;
    static_assert(NV <= ND && 4 * B <= 28, "codes_piece: window");
    const int32_t b0 = __mul24(B, t0);
    const uint32_t a0 = (uint32_t)((int32_t)end - (4 * ND - 1) - (b0 >> 3));
    const uint32_t* q = (const uint32_t*)(tile + (a0 & ~3u));
    const uint32_t sel = be_selector(a0), sh = (uint32_t)b0 & 7u;
    uint32_t d[ND + 1], W[ND], V[NV];
#pragma unroll
    for (int k = 0; k <= ND; ++k) d[k] = q[k];
#pragma unroll
    for (int k = 0; k < ND; ++k) W[k] = __builtin_amdgcn_perm(d[k + 1], d[k], sel);          // big-endian dwords, most significant first
#pragma unroll
    for (int i = 0; i < NV; ++i) V[i] = ALIGNED ? W[ND - 1 - i] : ND - 2 - i >= 0 ? __builtin_amdgcn_alignbit(W[ND - 2 - i >= 0 ? ND - 2 - i : 0], W[ND - 1 - i], sh) : W[ND - 1 - i] >> sh;
    constexpr uint32_t M = (1u << B) - 1u;
#pragma unroll
    for (int f = 0; f < 2 * K; ++f) {                      // four symbols t0 + 4 f .. + 3: bits 4 B f .. of V
        const int idx = (4 * B * f) / 32, s = (4 * B * f) % 32;
        const uint32_t x = s == 0 ? V[idx] : (idx + 1 < NV ? __builtin_amdgcn_alignbit(V[idx + 1 < NV ? idx + 1 : idx], V[idx], s) : V[idx] >> s);
        // the four B-bit fields into bytes in two steps (pairs into 16-bit halves, then each pair apart), then the bytes reversed: text order is
        // most significant first
        constexpr uint32_t M2 = (1u << (2 * B)) - 1u;
        const uint32_t t = (x & M2) | ((x << (16 - 2 * B)) & (M2 << 16));
        const uint32_t y = (t & (M * 0x00010001u)) | ((t << (8 - B)) & (M * 0x01000100u));
        const uint32_t o = __builtin_amdgcn_perm(0u, y, 0x00010203u);
        if (f & 1) lo[K - 1 - f / 2] = o; else hi[K - 1 - f / 2] = o;
    }
}
// the same for 2-bit base codes (spread2 is cheaper than four field extractions)
template <int K, bool ALIGNED = false>
__device__ __forceinline__ void dna_piece(const uint8_t* tile, uint32_t end, int32_t t0, uint32_t (&lo)[K], uint32_t (&hi)[K]) {
    constexpr int TB = 16 * K, NB = TB / 8 + 1, ND = (NB + 3) / 4, NV = (TB + 31) / 32;
    const int32_t b0 = 2 * t0;
    const uint32_t a0 = (uint32_t)((int32_t)end - (4 * ND - 1) - (b0 >> 3));
    const uint32_t* q = (const uint32_t*)(tile + (a0 & ~3u));
    const uint32_t sel = be_selector(a0), sh = (uint32_t)b0 & 7u;
    uint32_t d[ND + 1], W[ND], V[NV];
#pragma unroll
    for (int k = 0; k <= ND; ++k) d[k] = q[k];
#pragma unroll
    for (int k = 0; k < ND; ++k) W[k] = __builtin_amdgcn_perm(d[k + 1], d[k], sel);
#pragma unroll
    for (int i = 0; i < NV; ++i) V[i] = ALIGNED ? W[ND - 1 - i] : ND - 2 - i >= 0 ? __builtin_amdgcn_alignbit(W[ND - 2 - i >= 0 ? ND - 2 - i : 0], W[ND - 1 - i], sh) : W[ND - 1 - i] >> sh;
#pragma unroll
    for (int c = 0; c < K; ++c) {
        const int bit = 16 * (K - 1 - c);
        const uint32_t w = V[bit / 32] >> (bit % 32);
        lo[c] = spread2((w >> 8) & 0xFFu); hi[c] = spread2(w & 0xFFu);
    }
}
// one symbol through the tables: t = its index from the END of the read (rows are right-aligned)
__device__ __forceinline__ void decode_symbol(const uint8_t* drow, const uint8_t* qrow, const TileGeom& tg, uint32_t t, const uint8_t* l_base,
                                              const uint8_t* l_qual, const uint8_t* l_qn, uint8_t& cb, uint8_t& cc) {
    const uint32_t bitd = t * tg.bd, bitq = t * tg.bq;
    const uint32_t bd0 = tg.Cd - 1 - (bitd >> 3), bq0 = tg.Cq - 1 - (bitq >> 3);
    const uint32_t vd = drow[bd0] | (bd0 ? (uint32_t)drow[bd0 - 1] << 8 : 0u);
    const uint32_t vq = qrow[bq0] | (bq0 ? (uint32_t)qrow[bq0 - 1] << 8 : 0u);
    const uint32_t cd = (vd >> (bitd & 7)) & ((1u << tg.bd) - 1), cq = (vq >> (bitq & 7)) & ((1u << tg.bq) - 1);
    const uint8_t nb = l_qn[cq];
    cb = nb ? nb : l_base[cd];
    cc = l_qual[cq];
}

// packed form of emit_record_direct: symbols straight from the HBM rows (unpack_long_kernel's arithmetic)
__device__ void decode_record_direct(const EmitGeom& g, const TileGeom& tg, const uint8_t* l_base, const uint8_t* l_qual, const uint8_t* l_qn,
                                     const uint8_t* __restrict__ dna, const uint8_t* __restrict__ qual, uint32_t L,
                                     const uint64_t* __restrict__ offsets, uint8_t* __restrict__ out, uint64_t r, uint32_t lane) {
    uint8_t* o = out + offsets[r];
    const uint32_t pos = emit_qname_direct(g, o, r, lane);
    const uint8_t* drow = dna + r * tg.Cd;
    const uint8_t* qrow = qual + r * tg.Cq;
    for (uint32_t j = lane; j < L; j += 64) {
        uint8_t cb, cc;
        decode_symbol(drow, qrow, tg, L - 1 - j, l_base, l_qual, l_qn, cb, cc);
        o[pos + j] = cb;
        o[pos + L + 3 + j] = cc;
    }
    if (lane == 0) { o[pos + L] = '\n'; o[pos + L + 1] = '+'; o[pos + L + 2] = '\n'; o[pos + 2 * L + 3] = '\n'; }
}

// BQ / HASN: the packed form's lookup-free path with the quality width and the N-trick known at compile time (fixed-length tables
// whose chunk loop has the per-lane mapping: uq_decode_fastq picks the instance) -- no scalar dispatch per chunk, constant shifts:
// 1.59 -> 1.54 ms for 10 M x 150 bp at 6 bits.  BQ = 0: everything at run time.
template <bool PACKED, int BQ = 0, bool HASN = false, int K = 1>
__global__ __launch_bounds__(EM_THREADS, 4) void emit_tile_kernel(EmitGeom g, TileGeom tg, std::conditional_t<PACKED, UnpackLut, NoLut> lut,
                                                               const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual,
                                                               const uint32_t* __restrict__ len, uint64_t n, const uint64_t* __restrict__ offsets,
                                                               uint8_t* __restrict__ out) {
    extern __shared__ __align__(16) uint8_t tile[];                                  // [cap + 32] text image, then:
    unsigned long long* s_off = (unsigned long long*)(tile + tg.o_off);              // [R + 1] record offsets of the tile, as fetched
    uint32_t* s_len = (uint32_t*)(tile + tg.o_len);                                  // [R]
    uint16_t* flen = (uint16_t*)(tile + tg.o_flen);                                  // [R][ncols]
    uint32_t* cum = (uint32_t*)(tile + tg.o_cum);                                    // [R + 1] packed form: chunk items before record i
    __shared__ uint8_t l_tab[PACKED ? 768 : 4];
    const uint8_t* l_base = l_tab; const uint8_t* l_qual = l_tab + 256; const uint8_t* l_qn = l_tab + 512;
    // the QNAME layout (prefix, suffix, separators, the per-column pointers and offsets) is indexed per lane: an LDS copy
    // answers in a tenth of the time of the kernel-argument segment
    __shared__ EmitGeom sg;
    const uint32_t tid = threadIdx.x, lane = lane_id();
    for (uint32_t i = tid; i < sizeof(EmitGeom) / 4; i += EM_THREADS) ((uint32_t*)&sg)[i] = ((const uint32_t*)&g)[i];
    if constexpr (PACKED)
        for (uint32_t i = tid; i < 256; i += EM_THREADS) { l_tab[i] = lut.base_char[i]; l_tab[256 + i] = lut.qual_char[i]; l_tab[512 + i] = lut.qual_n_base[i]; }
    __syncthreads();
    const uint32_t R = tg.R, P = tg.P, cap = tg.cap;
    const uint64_t ntiles = (n + R - 1) / R;
    const uint32_t ncols = g.ncols;
    // Per-record metadata of the NEXT tile (offset, length, this lane's column value) -- and in the packed form its rows --
    // is requested one tile ahead and waits in registers: without it every tile paid three dependent global-memory
    // latencies before the first byte.
    const bool one_item = R * ncols <= EM_THREADS;       // lane == (record, field) item; else the fields reload their values
    struct Pre { unsigned long long off; uint64_t raw; uint32_t L; };
    auto fetch = [&](uint64_t tt) {
        Pre x; x.off = 0; x.raw = 0; x.L = 0;
        if (tt >= ntiles) return x;
        const uint64_t r0 = tt * R;
        const uint32_t Rt = (uint32_t)((n - r0) < R ? (n - r0) : R);
        if (tid <= Rt) x.off = offsets[r0 + tid];
        if (tid < Rt) { x.L = g.dna_max; if (len) x.L = len[r0 + tid]; }      // (not `len ? len[i] : g.dna_max`: the compiler selects between the two ADDRESSES -- kernel argument or table -- and the load becomes flat)
        if (one_item && tid < Rt * ncols) { const uint32_t i = tid / ncols, c = tid - i * ncols; x.raw = load_col(sg.col[c], sg.itemsize[c], r0 + i); }
        return x;
    };
    struct Rows { uint4 d[DE_NVD], q[DE_NVQ]; uint32_t skd, skq, nvd, nvq; };
    auto fetch_rows = [&](uint64_t tt) {
        Rows x;
        x.skd = x.skq = x.nvd = x.nvq = 0;
#pragma unroll
        for (int u = 0; u < DE_NVD; ++u) x.d[u] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < DE_NVQ; ++u) x.q[u] = make_uint4(0, 0, 0, 0);
        if (!PACKED || cap == 0 || tt >= ntiles) return x;
        const uint32_t Rt = (uint32_t)((n - tt * R) < R ? (n - tt * R) : R);
        const uint64_t ad = (uint64_t)(uintptr_t)(seq + tt * R * tg.Cd), aq = (uint64_t)(uintptr_t)(qual + tt * R * tg.Cq);
        x.skd = (uint32_t)(ad & 15); x.skq = (uint32_t)(aq & 15);
        x.nvd = (x.skd + Rt * tg.Cd + 15) >> 4; x.nvq = (x.skq + Rt * tg.Cq + 15) >> 4;
        const uint4* sd = (const uint4*)(seq + ((int64_t)(tt * R * tg.Cd) - (int64_t)x.skd));         // (global_load, not flat_load: see pack.hip)
        const uint4* sq = (const uint4*)(qual + ((int64_t)(tt * R * tg.Cq) - (int64_t)x.skq));
#pragma unroll
        for (int u = 0; u < DE_NVD; ++u) { const uint32_t i = u * EM_THREADS + tid; if (i < x.nvd) x.d[u] = sd[i]; }
#pragma unroll
        for (int u = 0; u < DE_NVQ; ++u) { const uint32_t i = u * EM_THREADS + tid; if (i < x.nvq) x.q[u] = sq[i]; }
        return x;
    };
    uint32_t my_slot = 0, my_j = 0;                        // packed form, fixed length: this lane's place in a step (see stage 2)
    if (PACKED) fast_divmod(tid, tg.NC, tg.magicNC, my_slot, my_j);
    Pre nx = fetch(blockIdx.x);
    Rows nr = fetch_rows(blockIdx.x);
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint64_t r0 = t * R;
        const uint32_t Rt = (uint32_t)((n - r0) < R ? (n - r0) : R);
        const Pre cur = nx;
        if (tid <= Rt) s_off[tid] = cur.off;
        if (tid < Rt) s_len[tid] = cur.L;
        uint32_t skd = 0, skq = 0;
        if constexpr (PACKED) {
            if (tid < 64) {                               // wave 0 holds every offset and length of the tile (R <= 63 here / 64)
                if constexpr (BQ != 0) {
                    // the compile-time instances (fixed lengths, per-lane chunk mapping) need no running count of chunk items:
                    // the array holds the records' image offsets as 32-bit numbers instead (ro() below: one LDS dword, no 64-bit
                    // arithmetic per chunk)
                    const unsigned long long first = __shfl(cur.off, 0, 64);
                    if (tid <= Rt) cum[tid] = (uint32_t)(cur.off - first) + (uint32_t)((uintptr_t)(out + first) & 15);
                } else {
                    uint32_t v = tid < Rt ? (cur.L + 14u) >> 3 : 0u;       // running count of chunk items: the most chunks a line of L characters can touch
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(v, d, 64); if (lane >= (uint32_t)d) v += o; }
                    if (tid < R) cum[tid + 1] = v;            // R + 1 entries: lanes beyond the tile have nowhere to write
                    if (tid == 0) cum[0] = 0;
                }
            }
            skd = nr.skd; skq = nr.skq;
#pragma unroll
            for (int u = 0; u < DE_NVD; ++u) { const uint32_t i = u * EM_THREADS + tid; if (i < nr.nvd) ((uint4*)(tile + tg.o_ind))[i] = nr.d[u]; }
#pragma unroll
            for (int u = 0; u < DE_NVQ; ++u) { const uint32_t i = u * EM_THREADS + tid; if (i < nr.nvq) ((uint4*)(tile + tg.o_inq))[i] = nr.q[u]; }
        }
        // ---- 1a: field lengths (from the prefetched column values)
        uint64_t my_mag = 0; bool my_neg = false; uint32_t my_moff = 0, my_fl = 0;
        if (one_item) {
            if (tid < Rt * ncols) {
                const uint32_t i = tid / ncols, c = tid - i * ncols;
                my_fl = field_from_raw(sg, c, cur.raw, my_mag, my_neg, my_moff);
                flen[i * ncols + c] = (uint16_t)my_fl;
            }
        } else {
            for (uint32_t idx = tid; idx < Rt * ncols; idx += EM_THREADS) {
                const uint32_t i = idx / ncols, c = idx - i * ncols;
                uint64_t mag; bool neg; uint32_t moff;
                flen[i * ncols + c] = (uint16_t)field_len(sg, c, r0 + i, mag, neg, moff);
            }
        }
        __syncthreads();
        nx = fetch(t + gridDim.x);
        if constexpr (PACKED) nr = fetch_rows(t + gridDim.x);
        const uint64_t o0 = s_off[0], o1 = s_off[Rt];
        const uint32_t skew = (uint32_t)((uintptr_t)(out + o0) & 15);
        const uint64_t span = o1 - o0;
        if (span + skew > cap) {                          // does not fit the image: direct_tiles_kernel writes this tile
            __syncthreads();
            continue;
        }
        // record i starts at image byte ro(i); its QUAL line ends right before ro(i + 1), so its SEQ text starts at
        // ro(i + 1) - 2 L - 4 whatever the QNAME was: the three parts below need no barrier between them
        auto ro = [&](uint32_t i) { if constexpr (PACKED && BQ != 0) return cum[i]; else return (uint32_t)(s_off[i] - o0) + skew; };
        // ---- 1b: render the fields, the separators and (lane of the last field) the suffix + '\n'
        for (uint32_t idx = tid; idx < Rt * ncols; idx += EM_THREADS) {
            const uint32_t i = idx / ncols, c = idx - i * ncols;
            uint32_t pos = g.prefix_len;
            for (uint32_t k = 0; k < c; ++k) pos += flen[i * ncols + k] + 1u;
            uint8_t* o = tile + ro(i) + pos;
            uint64_t mag = my_mag; bool neg = my_neg; uint32_t moff = my_moff;
            uint32_t fl = my_fl;
            if (!one_item) { const FieldCold f = field_len_cold(&sg, c, r0 + i); fl = f.fl; mag = f.mag; neg = f.neg != 0; moff = f.moff; }
            if (sg.map_chars[c]) {
                copy_chars_cold(o, sg.map_chars[c] + moff, fl);
            } else {
                uint32_t k = fl;
                if (mag >> 32) { do { o[--k] = (uint8_t)('0' + mag % 10); mag /= 10; } while (mag); }
                else { uint32_t m = (uint32_t)mag; do { const uint32_t q = m / 10; o[--k] = (uint8_t)('0' + (m - q * 10)); m = q; } while (m); }
                if (neg) o[--k] = '-';
            }
            if (c + 1 < ncols) o[fl] = sg.seps[c];
            else {
                for (uint32_t k = 0; k < g.suffix_len; ++k) o[fl + k] = sg.suffix[k];
                o[fl + g.suffix_len] = '\n';
            }
        }
        if (one_item && ncols) {                          // the prefix: shared out among the record's field lanes
            if (tid < Rt * ncols) {
                const uint32_t i = tid / ncols, c = tid - i * ncols;
                uint8_t* o = tile + ro(i);
                for (uint32_t k = c; k < g.prefix_len; k += ncols) o[k] = sg.prefix[k];
            }
        } else {
            for (uint32_t idx = tid; idx < Rt * g.prefix_len; idx += EM_THREADS) {
                uint32_t i, k;
                fast_divmod(idx, g.prefix_len, tg.magicP, i, k);
                tile[ro(i) + k] = sg.prefix[k];
            }
        }
        if (ncols == 0) {                                 // no columns: QNAME = prefix + suffix
            for (uint32_t i = tid; i < Rt; i += EM_THREADS) {
                uint8_t* o = tile + ro(i) + g.prefix_len;
                for (uint32_t k = 0; k < g.suffix_len; ++k) o[k] = sg.suffix[k];
                o[g.suffix_len] = '\n';
            }
        }
        // ---- 2: SEQ and QUAL text
        if constexpr (PACKED) {
            const uint8_t* in_d = tile + tg.o_ind + skd;
            const uint8_t* in_q = tile + tg.o_inq + skq;
            const FastAlphabet& fa = tg.fa;
            if (fa.fast) {
                // the lookup-free alphabet, from the destination: a line of L characters that starts at image byte d lies in
                // the aligned 8-byte chunks j = 0 .. (d % 8 + L - 1) / 8 behind d - d % 8; chunk j holds the characters
                // p0 = 8 j - d % 8 .. p0 + 7.  A chunk wholly inside the line is one aligned LDS store of the main loop; the
                // (at most two) others of a line are left to the edge pass below.
                const uint32_t od = tg.o_ind + skd + tg.Cd - 1, oq = tg.o_inq + skq + tg.Cq - 1;      // last byte of row 0
                auto seq8 = [&](uint32_t r, uint32_t L, int32_t p0) {        // characters p0 .. p0 + 7 of read r's SEQ line
                    const int32_t t0 = (int32_t)L - 8 - p0;
                    uint32_t clo, chi;
                    if (fa.bd == 2) dna_codes8(tile, od + __umul24(r, tg.Cd), t0, clo, chi);
                    else qual_codes8<3>(tile, od + __umul24(r, tg.Cd), t0, clo, chi);       // 3-bit bases: rows laid out like quality rows
                    uint32_t blo = __builtin_amdgcn_perm(fa.base_tab_hi, fa.base_tab, clo), bhi = __builtin_amdgcn_perm(fa.base_tab_hi, fa.base_tab, chi);
                    if (BQ ? HASN : fa.has_n != 0) {
                        uint32_t qlo, qhi;
                        if constexpr (BQ != 0) qual_codes8<BQ>(tile, oq + __umul24(r, tg.Cq), t0, qlo, qhi);
                        else qual_codes8(tg.bq, tile, oq + __umul24(r, tg.Cq), t0, qlo, qhi);
                        const uint32_t mlo = ~nonzero_bytes(qlo ^ fa.n_code4), mhi = ~nonzero_bytes(qhi ^ fa.n_code4);
                        blo = bfi(mlo, fa.n_char4, blo); bhi = bfi(mhi, fa.n_char4, bhi);
                    }
                    return ((uint64_t)bhi << 32) | blo;
                };
                auto qual8 = [&](uint32_t r, uint32_t L, int32_t p0) {       // ... of its QUAL line
                    uint32_t qlo, qhi;
                    if constexpr (BQ != 0) qual_codes8<BQ>(tile, oq + __umul24(r, tg.Cq), (int32_t)L - 8 - p0, qlo, qhi);
                    else qual_codes8(tg.bq, tile, oq + __umul24(r, tg.Cq), (int32_t)L - 8 - p0, qlo, qhi);
                    // code + qmin; a code beyond the alphabet decodes to the tables' 0
                    const uint32_t olo = nonzero_bytes((qlo + fa.q_over) & 0x80808080u), ohi = nonzero_bytes((qhi + fa.q_over) & 0x80808080u);
                    qlo = (qlo + fa.qmin4) & ~olo; qhi = (qhi + fa.qmin4) & ~ohi;
                    return ((uint64_t)qhi << 32) | qlo;
                };
                auto chunks = [&](uint32_t r, uint32_t j, uint32_t L) {      // chunk j of both lines of read r, where whole
                    const uint32_t e = ro(r + 1), ds = e - 2 * L - 4, dq = e - L - 1;
                    const int32_t ps = (int32_t)(8 * j) - (int32_t)(ds & 7u), pq = (int32_t)(8 * j) - (int32_t)(dq & 7u);
                    if (ps >= 0 && ps + 8 <= (int32_t)L) *(uint64_t*)(tile + (ds & ~7u) + 8 * j) = seq8(r, L, ps);
                    if (pq >= 0 && pq + 8 <= (int32_t)L) *(uint64_t*)(tile + (dq & ~7u) + 8 * j) = qual8(r, L, pq);
                };
                // K chunks of both lines of read r (K > 1: my_j counts pieces of K chunks)
                auto pieces = [&](uint32_t r, uint32_t m, uint32_t L) {
                    if constexpr (K > 1 && BQ != 0) {
                        const uint32_t e = ro(r + 1), ds = e - 2 * L - 4, dq = e - L - 1;
                        const int32_t ps = (int32_t)(8 * K * m) - (int32_t)(ds & 7u), pq = (int32_t)(8 * K * m) - (int32_t)(dq & 7u);
                        const uint32_t endd = od + __umul24(r, tg.Cd), endq = oq + __umul24(r, tg.Cq);
                        uint32_t lo[K], hi[K];
                        if (ps + 8 * K > 0 && ps < (int32_t)L) {
                            const int32_t t0 = (int32_t)L - 8 * K - ps;
                            if (fa.bd == 2) dna_piece<K>(tile, endd, t0, lo, hi); else codes_piece<3, K>(tile, endd, t0, lo, hi);
                            uint32_t nlo[K], nhi[K];
                            if constexpr (HASN) codes_piece<BQ, K>(tile, endq, t0, nlo, nhi);
                            uint8_t* base = tile + (ds & ~7u) + 8 * K * m;
#pragma unroll
                            for (int c = 0; c < K; ++c) {
                                if (HASN && fa.bd == 2) { lo[c] = n_selectors(lo[c], nlo[c], fa.n_code4); hi[c] = n_selectors(hi[c], nhi[c], fa.n_code4); }
                                uint32_t blo = __builtin_amdgcn_perm(fa.base_tab_hi, fa.base_tab, lo[c]), bhi = __builtin_amdgcn_perm(fa.base_tab_hi, fa.base_tab, hi[c]);
                                if (HASN && fa.bd != 2) {
                                    const uint32_t mlo = ~nonzero_bytes(nlo[c] ^ fa.n_code4), mhi = ~nonzero_bytes(nhi[c] ^ fa.n_code4);
                                    blo = bfi(mlo, fa.n_char4, blo); bhi = bfi(mhi, fa.n_char4, bhi);
                                }
                                const int32_t p = ps + 8 * c;
                                if (p >= 0 && p + 8 <= (int32_t)L) *(uint64_t*)(base + 8 * c) = ((uint64_t)bhi << 32) | blo;
                            }
                        }
                        if (pq + 8 * K > 0 && pq < (int32_t)L) {
                            codes_piece<BQ, K>(tile, endq, (int32_t)L - 8 * K - pq, lo, hi);
                            uint8_t* base = tile + (dq & ~7u) + 8 * K * m;
                            uint32_t over = 0;                                     // bit 7 of a byte: a code beyond the alphabet in one of the chunks stored
#pragma unroll
                            for (int c = 0; c < K; ++c) {
                                const int32_t p = pq + 8 * c;
                                if (p >= 0 && p + 8 <= (int32_t)L) {
                                    over |= (lo[c] + fa.q_over) | (hi[c] + fa.q_over);
                                    *(uint64_t*)(base + 8 * c) = ((uint64_t)(hi[c] + fa.qmin4) << 32) | (lo[c] + fa.qmin4);
                                }
                            }
                            if (over & 0x80808080u) {                              // (no encoder writes such a table) those codes decode to 0, as through the tables
#pragma unroll
                                for (int c = 0; c < K; ++c) {
                                    uint32_t qlo = lo[c], qhi = hi[c];
                                    uint32_t olo = (qlo + fa.q_over) & 0x80808080u, ohi = (qhi + fa.q_over) & 0x80808080u;
                                    olo |= olo - (olo >> 7); ohi |= ohi - (ohi >> 7);
                                    qlo = (qlo + fa.qmin4) & ~olo; qhi = (qhi + fa.qmin4) & ~ohi;
                                    const int32_t p = pq + 8 * c;
                                    if (p >= 0 && p + 8 <= (int32_t)L) *(uint64_t*)(base + 8 * c) = ((uint64_t)qhi << 32) | qlo;
                                }
                            }
                        }
                    }
                };
                if (K > 1) {
                    if (my_slot < tg.RS) for (uint32_t r = my_slot; r < Rt; r += tg.RS) pieces(r, my_j, g.dna_max);
                } else
                if (BQ == 0 && (tg.variable || tg.RS == 0)) {  // (RS = 0: fixed-length reads of more chunks than the workgroup has lanes)
                    // flat over the tile: item = (read, chunk) in the order of cum[] (the running count of chunks, left by wave 0
                    // before the first barrier)
                    const uint32_t total = cum[Rt];
                    for (uint32_t item = tid; item < total; item += EM_THREADS) {
                        uint32_t r = 0, hi = Rt;                             // largest r with cum[r] <= item
#pragma unroll
                        for (int it = 0; it < 6; ++it) { const uint32_t mid = (r + hi) >> 1; if (cum[mid] <= item) r = mid; else hi = mid; }
                        chunks(r, item - cum[r], s_len[r]);
                    }
                } else if (my_slot < tg.RS) {
                    // fixed length: a lane keeps its chunk number for the whole kernel (my_slot, my_j = tid / NC, tid % NC) and
                    // walks over the reads my_slot, my_slot + RS, ...
                    for (uint32_t r = my_slot; r < Rt; r += tg.RS) chunks(r, my_j, g.dna_max);
                }
                // the edge pass: wave 0 / 1 = first / last chunk of the SEQ lines, wave 2 / 3 = of the QUAL lines, a lane per read
                // (R <= 64): the chunk's bytes inside the line one by one, where the main loop left it out.  Wave 0 also writes
                // the separators.
                static_assert(EM_THREADS == 4 * 64 && EM_RMAX <= 64, "the edge pass is one step of four waves");
                do {
                    const uint32_t i = lane, w = tid >> 6;
                    if (i >= Rt) continue;
                    const uint32_t L = s_len[i], e = ro(i + 1);
                    const uint32_t d = w < 2 ? e - 2 * L - 4 : e - L - 1;
                    if (w == 0) { tile[d + L] = '\n'; tile[d + L + 1] = '+'; tile[d + L + 2] = '\n'; tile[d + 2 * L + 3] = '\n'; }
                    if (L == 0) continue;
                    const uint32_t a = d & 7u, j = (w & 1u) ? (a + L - 1) >> 3 : 0u;
                    if ((w & 1u) && j == 0) continue;                       // a one-chunk line: the head lane has it
                    const int32_t p0 = (int32_t)(8 * j) - (int32_t)a;
                    const int32_t k0 = p0 < 0 ? -p0 : 0, k1 = (int32_t)L - p0 < 8 ? (int32_t)L - p0 : 8;
                    if (k0 == 0 && k1 == 8) continue;                        // whole: stored by the main loop
                    const uint64_t v = w < 2 ? seq8(i, L, p0) : qual8(i, L, p0);
                    uint8_t* c = tile + (d & ~7u) + 8 * j;
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (k >= k0 && k < k1) c[k] = (uint8_t)(v >> (8 * k));
                } while (false);
            } else {
                // any other alphabet: a lane per 8 symbols, every symbol through the tables
                for (uint32_t i = tid; i < Rt; i += EM_THREADS) {
                    const uint32_t L = s_len[i];
                    uint8_t* ts = tile + ro(i + 1) - 2 * L - 4;
                    ts[L] = '\n'; ts[L + 1] = '+'; ts[L + 2] = '\n'; ts[2 * L + 3] = '\n';
                }
                const uint32_t items = Rt * tg.G;
                const uint32_t md = (1u << tg.bd) - 1;
                const uint64_t mq = (1ull << tg.bq) - 1;
                for (uint32_t idx = tid; idx < items; idx += EM_THREADS) {
                    uint32_t r, gg;
                    fast_divmod(idx, tg.G, tg.magicG, r, gg);
                    const uint32_t L = s_len[r];
                    if (8 * gg >= L) continue;
                    uint8_t* ts = tile + ro(r + 1) - 2 * L - 4;             // first SEQ byte in the tile
                    uint8_t* tq = ts + L + 3;                               // first QUAL byte
                    const uint64_t vd = group_bits(in_d + r * tg.Cd, tg.Cd, tg.bd, gg);
                    const uint64_t vq = group_bits(in_q + r * tg.Cq, tg.Cq, tg.bq, gg);
#pragma unroll
                    for (uint32_t i = 0; i < 8; ++i) {
                        const uint32_t tt = 8 * gg + i;
                        if (tt < L) {
                            const uint32_t cd = (uint32_t)(vd >> (tg.bd * i)) & md, cq = (uint32_t)((vq >> (tg.bq * i)) & mq);
                            const uint8_t nb = l_qn[cq];
                            ts[L - 1 - tt] = nb ? nb : l_base[cd];
                            tq[L - 1 - tt] = l_qual[cq];
                        }
                    }
                }
            }
        } else {
            // P lanes per record
            const uint32_t i = tid / P, p = tid - i * P;
            if (i < Rt) {
                const uint64_t r = r0 + i;
                const uint32_t L = s_len[i];
                const uint32_t ds = ro(i + 1) - 2 * L - 4;              // first SEQ byte in the tile
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
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the next tile's rows, requested a whole phase ago -- with nothing pending, none of the store loops below is entered behind a wait (decode_stream_kernel tells the story)
        // ---- 3: LDS image -> HBM (tile[skew ...] is out[o0 ...]; interior vectors are 16-byte aligned on both sides)
        {
            uint8_t* dst = out + o0 - skew;                            // 16-byte aligned
            const uint32_t endb = skew + (uint32_t)span;
            const uint32_t v0 = skew ? 1u : 0u, v1 = endb >> 4;         // full vectors [v0, v1)
            for (uint32_t v = v0 + tid; v < v1; v += EM_THREADS) ((uint4*)dst)[v] = ((const uint4*)tile)[v];
            if (skew) for (uint32_t b = skew + tid; b < 16 && b < endb; b += EM_THREADS) dst[b] = tile[b];
            if (v1 >= v0) for (uint32_t b = (v1 << 4) + tid; b < endb; b += EM_THREADS) dst[b] = tile[b];
        }
        // no barrier here: what the next tile writes before ITS first barrier (offsets, lengths, field lengths, rows) is not read
        // by stage 3, and every lane that gets there has passed the barrier above, behind the last readers of those arrays
    }
}

// ---- the packed form with the lookup-free alphabet: SEQ and QUAL text go from the rows STRAIGHT to HBM.
// Global stores, unlike LDS ones, take any alignment at nearly full speed (tools/storebench.hip, profiles/r02_l_storebench.txt:
// 8 bytes a lane at an odd address 4.25 TB/s, aligned 4.7), and the L2 puts the pieces of a cache line together before it
// leaves for HBM.  So the decoder can work from the SOURCE -- group wg of a read = the eight symbols its row bytes
// Cd - 2 wg - 2 .. and Cq - bq (wg + 1) .. hold, byte-aligned in the row, no bit shifting -- and store each group's eight
// SEQ and eight QUAL characters where they belong; a lane per read adds the L % 8 characters at the front of the two lines
// and the separators.  Only the QNAME lines, rendered a digit at a time, still go through LDS: a staging area of 8-byte
// aligned pieces, copied out eight bytes a lane.  The text image of emit_tile_kernel, its alignment arithmetic and its copy
// to HBM are gone; what is paid instead is HBM write traffic (pieces of a cache line that leave the L2 before the others
// arrive: + 20 % measured), so uq_decode_fastq takes this kernel where the image's index arithmetic is dear: variable lengths.
// Per tile: [A] metadata + rows (prefetched a tile ahead) -> LDS, field lengths | barrier | QNAME -> staging, groups -> HBM,
// front pieces -> HBM | barrier | staging -> HBM.  The small per-tile arrays are double-buffered by tile parity, so the
// next tile's [A] needs no third barrier.
struct StreamGeom {
    uint32_t R, qcap, set;                                       // reads per tile, bytes of QNAME staging, bytes of one set of the per-tile arrays
    uint32_t o_set, o_ind, o_inq;                                // LDS offsets: the two sets, the rows
    uint32_t s_off, s_len, s_flen, s_cum, s_qst;                 // offsets inside a set
    uint32_t bq, Cd, Cq, variable, Gf, magicGf, RS;              // Gf = whole groups of a fixed-length read, RS = reads per workgroup step
    FastAlphabet fa;
};

// the eight characters of group t0 / 8 (t0 = index from the end of the read of the group's LAST character); BQ / HASN as in emit_tile_kernel
template <int BQ, bool HASN>
__device__ __forceinline__ void group_text(const uint8_t* tile, const StreamGeom& sg2, uint32_t endd, uint32_t endq, int32_t t0, uint64_t& vb, uint64_t& vq) {
    const FastAlphabet& fa = sg2.fa;
    uint32_t clo, chi, qlo, qhi;
    if (fa.bd == 2) dna_codes8(tile, endd, t0, clo, chi);
    else qual_codes8<3>(tile, endd, t0, clo, chi);                   // 3-bit bases: rows laid out like quality rows
    if constexpr (BQ != 0) qual_codes8<BQ>(tile, endq, t0, qlo, qhi);
    else qual_codes8(sg2.bq, tile, endq, t0, qlo, qhi);
    uint32_t blo = __builtin_amdgcn_perm(fa.base_tab_hi, fa.base_tab, clo), bhi = __builtin_amdgcn_perm(fa.base_tab_hi, fa.base_tab, chi);
    if (BQ ? HASN : fa.has_n != 0) {
        const uint32_t mlo = ~nonzero_bytes(qlo ^ fa.n_code4), mhi = ~nonzero_bytes(qhi ^ fa.n_code4);
        blo = bfi(mlo, fa.n_char4, blo); bhi = bfi(mhi, fa.n_char4, bhi);
    }
    // code + qmin; a code beyond the alphabet decodes to the tables' 0
    const uint32_t olo = nonzero_bytes((qlo + fa.q_over) & 0x80808080u), ohi = nonzero_bytes((qhi + fa.q_over) & 0x80808080u);
    qlo = (qlo + fa.qmin4) & ~olo; qhi = (qhi + fa.qmin4) & ~ohi;
    vb = ((uint64_t)bhi << 32) | blo; vq = ((uint64_t)qhi << 32) | qlo;
}
// the low nb (< 8) bytes of v to p, any alignment
__device__ __forceinline__ void store_low_bytes(uint8_t* p, uint64_t v, uint32_t nb) {
    if (nb & 4u) { const uint32_t w = (uint32_t)v; __builtin_memcpy(p, &w, 4); v >>= 32; p += 4; }
    if (nb & 2u) { const uint16_t w = (uint16_t)v; __builtin_memcpy(p, &w, 2); v >>= 16; p += 2; }
    if (nb & 1u) *p = (uint8_t)v;
}

template <int BQ = 0, bool HASN = false, int K = 1>
__global__ __launch_bounds__(EM_THREADS, DS_OCC) void decode_stream_kernel(EmitGeom g, StreamGeom tg, const uint8_t* __restrict__ dna, const uint8_t* __restrict__ qual,
                                                                   const uint32_t* __restrict__ len, uint64_t n, const uint64_t* __restrict__ offsets,
                                                                   uint8_t* __restrict__ out) {
    extern __shared__ __align__(16) uint8_t tile[];                                  // [qcap] QNAME staging, the two sets of per-tile arrays, the rows
    __shared__ EmitGeom sg;                                                          // (see emit_tile_kernel)
    const uint32_t tid = threadIdx.x, lane = lane_id();
    for (uint32_t i = tid; i < sizeof(EmitGeom) / 4; i += EM_THREADS) ((uint32_t*)&sg)[i] = ((const uint32_t*)&g)[i];
    __syncthreads();
    const uint32_t R = tg.R, ncols = g.ncols;
    const uint64_t ntiles = (n + R - 1) / R;
    const bool one_item = R * ncols <= EM_THREADS;       // lane == (record, field) item; else the fields reload their values
    struct Pre { unsigned long long off; uint64_t raw; uint32_t L; };
    auto fetch = [&](uint64_t tt) {
        Pre x; x.off = 0; x.raw = 0; x.L = 0;
        if (tt >= ntiles) return x;
        const uint64_t r0 = tt * R;
        const uint32_t Rt = (uint32_t)((n - r0) < R ? (n - r0) : R);
        if (tid <= Rt) x.off = offsets[r0 + tid];
        if (tid < Rt) x.L = len ? len[r0 + tid] : g.dna_max;
        if (one_item && tid < Rt * ncols) { const uint32_t i = tid / ncols, c = tid - i * ncols; x.raw = load_col(sg.col[c], sg.itemsize[c], r0 + i); }
        return x;
    };
    struct Rows { uint4 d[DE_NVD], q[DE_NVQ]; uint32_t skd, skq, nvd, nvq; };
    auto fetch_rows = [&](uint64_t tt) {
        Rows x;
        x.skd = x.skq = x.nvd = x.nvq = 0;
#pragma unroll
        for (int u = 0; u < DE_NVD; ++u) x.d[u] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < DE_NVQ; ++u) x.q[u] = make_uint4(0, 0, 0, 0);
        if (tt >= ntiles) return x;
        const uint32_t Rt = (uint32_t)((n - tt * R) < R ? (n - tt * R) : R);
        const uint64_t ad = (uint64_t)(uintptr_t)(dna + tt * R * tg.Cd), aq = (uint64_t)(uintptr_t)(qual + tt * R * tg.Cq);
        x.skd = (uint32_t)(ad & 15); x.skq = (uint32_t)(aq & 15);
        x.nvd = (x.skd + Rt * tg.Cd + 15) >> 4; x.nvq = (x.skq + Rt * tg.Cq + 15) >> 4;
        const uint4* sd = (const uint4*)(dna + ((int64_t)(tt * R * tg.Cd) - (int64_t)x.skd));         // (global_load, not flat_load: see pack.hip)
        const uint4* sq = (const uint4*)(qual + ((int64_t)(tt * R * tg.Cq) - (int64_t)x.skq));
#pragma unroll
        for (int u = 0; u < DE_NVD; ++u) { const uint32_t i = u * EM_THREADS + tid; if (i < x.nvd) x.d[u] = sd[i]; }
#pragma unroll
        for (int u = 0; u < DE_NVQ; ++u) { const uint32_t i = u * EM_THREADS + tid; if (i < x.nvq) x.q[u] = sq[i]; }
        return x;
    };
    uint32_t my_slot = 0, my_wg = 0;                       // fixed length: this lane's place in a step of the group loop
    if (tg.Gf) { fast_divmod(tid, tg.Gf, tg.magicGf, my_slot, my_wg); my_wg = tg.Gf - 1 - my_wg; }      // addresses rise with the lane
    Pre nx = fetch(blockIdx.x);
    Rows nr = fetch_rows(blockIdx.x);
    uint32_t par = 0;
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x, par ^= 1u) {
        uint8_t* set = tile + tg.o_set + par * tg.set;
        unsigned long long* s_off = (unsigned long long*)(set + tg.s_off);           // [R + 1] record offsets of the tile
        uint32_t* s_len = (uint32_t*)(set + tg.s_len);                               // [R]
        uint16_t* flen = (uint16_t*)(set + tg.s_flen);                               // [R][ncols]
        uint32_t* cum = (uint32_t*)(set + tg.s_cum);                                 // [R + 1] whole groups before read i
        uint32_t* qst = (uint32_t*)(set + tg.s_qst);                                 // [R + 1] staging offset of read i's QNAME line (8-byte aligned pieces)
        const uint64_t r0 = t * R;
        const uint32_t Rt = (uint32_t)((n - r0) < R ? (n - r0) : R);
        const Pre cur = nx;
        // ---- A
        if (tid <= Rt) s_off[tid] = cur.off;
        if (tid < Rt) s_len[tid] = cur.L;
        if (tid < 64) {                                   // wave 0 holds every offset and length of the tile (R <= 63): two running sums
            const uint32_t lo1 = __shfl_down((uint32_t)cur.off, 1, 64);               // record sizes fit 32 bits
            const uint32_t size = lo1 - (uint32_t)cur.off;
            uint32_t v = tid < Rt ? ((cur.L >> 3) + (K - 1)) / K : 0u;                // items: pieces of K whole groups (the one at the line's front may hold fewer)
            uint32_t q = tid < Rt ? (size - 2u * cur.L - 4u + 7u) & ~7u : 0u;        // the QNAME line with its '\n', rounded up
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t ov = __shfl_up(v, d, 64), oq = __shfl_up(q, d, 64);
                if (lane >= (uint32_t)d) { v += ov; q += oq; }
            }
            if (tid < R) { cum[tid + 1] = v; qst[tid + 1] = q; }
            if (tid == 0) { cum[0] = 0; qst[0] = 0; }
        }
        const uint32_t skd = nr.skd, skq = nr.skq;
#pragma unroll
        for (int u = 0; u < DE_NVD; ++u) { const uint32_t i = u * EM_THREADS + tid; if (i < nr.nvd) ((uint4*)(tile + tg.o_ind))[i] = nr.d[u]; }
#pragma unroll
        for (int u = 0; u < DE_NVQ; ++u) { const uint32_t i = u * EM_THREADS + tid; if (i < nr.nvq) ((uint4*)(tile + tg.o_inq))[i] = nr.q[u]; }
        uint64_t my_mag = 0; bool my_neg = false; uint32_t my_moff = 0, my_fl = 0;
        if (one_item) {
            if (tid < Rt * ncols) {
                const uint32_t i = tid / ncols, c = tid - i * ncols;
                my_fl = field_from_raw(sg, c, cur.raw, my_mag, my_neg, my_moff);
                flen[i * ncols + c] = (uint16_t)my_fl;
            }
        } else {
            for (uint32_t idx = tid; idx < Rt * ncols; idx += EM_THREADS) {
                const uint32_t i = idx / ncols, c = idx - i * ncols;
                uint64_t mag; bool neg; uint32_t moff;
                flen[i * ncols + c] = (uint16_t)field_len(sg, c, r0 + i, mag, neg, moff);
            }
        }
        __syncthreads();
        nx = fetch(t + gridDim.x);
        nr = fetch_rows(t + gridDim.x);
        const bool staged = qst[Rt] <= tg.qcap;           // else (QNAME lines far beyond what the staging was sized for): a wave per line, straight out
        // ---- QNAME lines -> staging: render the fields, the separators and (lane of the last field) the suffix + '\n'
#ifdef DS_ABL_NOQNAME
        if (false) {
#else
        if (staged) {
#endif
            for (uint32_t idx = tid; idx < Rt * ncols; idx += EM_THREADS) {
                const uint32_t i = idx / ncols, c = idx - i * ncols;
                uint32_t pos = g.prefix_len;
                for (uint32_t k = 0; k < c; ++k) pos += flen[i * ncols + k] + 1u;
                uint8_t* o = tile + qst[i] + pos;
                uint64_t mag = my_mag; bool neg = my_neg; uint32_t moff = my_moff;
                // (inline here, unlike emit_tile_kernel: this kernel sits at its 128 registers, and calls -- their live registers saved around them -- made it spill;
                // its one explicit wait below covers what these loads leave pending)
                const uint32_t fl = one_item ? my_fl : field_len(sg, c, r0 + i, mag, neg, moff);
                if (sg.map_chars[c]) {
                    const gptr<uint8_t> mc = as_global<uint8_t>(sg.map_chars[c]);
                    for (uint32_t k = 0; k < fl; ++k) o[k] = mc[moff + k];
                } else {
                    uint32_t k = fl;
                    if (mag >> 32) { do { o[--k] = (uint8_t)('0' + mag % 10); mag /= 10; } while (mag); }
                    else { uint32_t m = (uint32_t)mag; do { const uint32_t q = m / 10; o[--k] = (uint8_t)('0' + (m - q * 10)); m = q; } while (m); }
                    if (neg) o[--k] = '-';
                }
                if (c + 1 < ncols) o[fl] = sg.seps[c];
                else {
                    for (uint32_t k = 0; k < g.suffix_len; ++k) o[fl + k] = sg.suffix[k];
                    o[fl + g.suffix_len] = '\n';
                }
            }
            if (one_item && ncols) {                      // the prefix: shared out among the record's field lanes
                if (tid < Rt * ncols) {
                    const uint32_t i = tid / ncols, c = tid - i * ncols;
                    uint8_t* o = tile + qst[i];
                    for (uint32_t k = c; k < g.prefix_len; k += ncols) o[k] = sg.prefix[k];
                }
            } else {
                for (uint32_t i = tid; i < Rt; i += EM_THREADS) {
                    uint8_t* o = tile + qst[i];
                    for (uint32_t k = 0; k < g.prefix_len; ++k) o[k] = sg.prefix[k];
                    if (ncols == 0) {                     // no columns: QNAME = prefix + suffix
                        for (uint32_t k = 0; k < g.suffix_len; ++k) o[g.prefix_len + k] = sg.suffix[k];
                        o[g.prefix_len + g.suffix_len] = '\n';
                    }
                }
            }
        }
        // The next tile's rows and metadata were requested before the QNAME lines were rendered: wait for them HERE, before the tile's first store.  With a load
        // pending, the compiler puts a `s_waitcnt vmcnt(0)` in front of every loop that stores (its preheader rule for targets whose stores share the loads'
        // counter): in front of the group loop it waited for the rows just requested, in front of the front pieces for all of the groups' stores to be
        // acknowledged, in front of the QNAME copy for the front pieces', in front of the next tile's phase A for the copy's -- four round trips a tile, which
        // is what the ablations saw (the kernel's time did not depend on where its stores went).  Behind this wait nothing is pending and no store waits.
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0); expcnt / lgkmcnt left alone
        // ---- whole groups -> HBM
        const uint32_t od = tg.o_ind + skd + tg.Cd - 1, oq = tg.o_inq + skq + tg.Cq - 1;      // last byte of row 0
        if (tg.variable || tg.RS == 0) {                  // (RS = 0: fixed-length reads of more groups than the workgroup has lanes)
            // flat over the tile: item = (read, group) in the order of cum[]
#ifdef DS_ABL_NOGROUPS
            const uint32_t total = 0;
#else
            const uint32_t total = cum[Rt];
#endif
            for (uint32_t item = tid; item < total; item += EM_THREADS) {
                uint32_t r = 0, hi = Rt;                                 // largest r with cum[r] <= item
#pragma unroll
                for (int it = 0; it < 6; ++it) { const uint32_t mid = (r + hi) >> 1; if (cum[mid] <= item) r = mid; else hi = mid; }
                const uint32_t wg = cum[r + 1] - 1 - item, L = s_len[r];     // addresses rise with the lane
                if constexpr (K > 1 && BQ != 0) {
                    // K groups a lane: one window per row, 8 K contiguous characters of each line (the stores of a full piece are 16 bytes wide)
                    const uint32_t have = (L >> 3) - K * wg, nv = have < (uint32_t)K ? have : (uint32_t)K;       // groups of the piece inside the read
                    const uint32_t endd = od + __umul24(r, tg.Cd), endq = oq + __umul24(r, tg.Cq);
                    const FastAlphabet& fa = tg.fa;
                    uint32_t blo[K], bhi[K], qlo[K], qhi[K];
                    if (fa.bd == 2) dna_piece<K, true>(tile, endd, (int32_t)(8 * K * wg), blo, bhi); else codes_piece<3, K, true>(tile, endd, (int32_t)(8 * K * wg), blo, bhi);
                    codes_piece<BQ, K, true>(tile, endq, (int32_t)(8 * K * wg), qlo, qhi);
                    uint32_t over = 0;
#pragma unroll
                    for (int c = 0; c < K; ++c) {
                        if (HASN && fa.bd == 2) { blo[c] = n_selectors(blo[c], qlo[c], fa.n_code4); bhi[c] = n_selectors(bhi[c], qhi[c], fa.n_code4); }
                        blo[c] = __builtin_amdgcn_perm(fa.base_tab_hi, fa.base_tab, blo[c]); bhi[c] = __builtin_amdgcn_perm(fa.base_tab_hi, fa.base_tab, bhi[c]);
                        if (HASN && fa.bd != 2) {
                            const uint32_t mlo = ~nonzero_bytes(qlo[c] ^ fa.n_code4), mhi = ~nonzero_bytes(qhi[c] ^ fa.n_code4);
                            blo[c] = bfi(mlo, fa.n_char4, blo[c]); bhi[c] = bfi(mhi, fa.n_char4, bhi[c]);
                        }
                        over |= (qlo[c] + fa.q_over) | (qhi[c] + fa.q_over);
                    }
                    if (over & 0x80808080u) {                                // a code beyond the alphabet (no encoder writes one; the bits in front of a short piece may look like one) decodes to 0
#pragma unroll
                        for (int c = 0; c < K; ++c) {
                            uint32_t olo = (qlo[c] + fa.q_over) & 0x80808080u, ohi = (qhi[c] + fa.q_over) & 0x80808080u;
                            olo |= olo - (olo >> 7); ohi |= ohi - (ohi >> 7);
                            qlo[c] = (qlo[c] + fa.qmin4) & ~olo; qhi[c] = (qhi[c] + fa.qmin4) & ~ohi;
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < K; ++c) { qlo[c] += fa.qmin4; qhi[c] += fa.qmin4; }
                    }
#ifdef DS_ABL_SMALLOUT
                    uint8_t* ts = out + 4096u + (blockIdx.x & 255u) * 65536u + tid * 128u - L;      // (ablation: every store lands in a few MiB that stay in the L2s)
#else
                    uint8_t* ts = out + s_off[r + 1] - (L + 8 * K * (wg + 1) + 4);      // chunk c of the piece: SEQ at ts + 8 c, QUAL at ts + L + 3 + 8 c
#endif
#ifdef DS_ABL_NOFULL
                    if (nv == (uint32_t)K) { if ((blo[0] ^ qlo[0]) == 0x12345678u && bhi[K - 1] == qhi[K - 1]) ts[0] = 1; } else
#endif
#ifdef DS_ABL_NOPART
                    if (nv != (uint32_t)K) { if ((blo[1] ^ qlo[1]) == 0x12345678u && bhi[K - 1] == qhi[K - 1]) ts[0] = 1; } else
#endif
                    if (nv == (uint32_t)K) {
                        uint32_t vb[2 * K], vq[2 * K];
#pragma unroll
                        for (int c = 0; c < K; ++c) { vb[2 * c] = blo[c]; vb[2 * c + 1] = bhi[c]; vq[2 * c] = qlo[c]; vq[2 * c + 1] = qhi[c]; }
                        __builtin_memcpy(ts, vb, 8 * K);
                        __builtin_memcpy(ts + L + 3, vq, 8 * K);
                    } else {
#pragma unroll
                        for (int c = 1; c < K; ++c) {
                            if ((uint32_t)c >= (uint32_t)K - nv) {
                                const uint64_t b = ((uint64_t)bhi[c] << 32) | blo[c], q = ((uint64_t)qhi[c] << 32) | qlo[c];
                                __builtin_memcpy(ts + 8 * c, &b, 8);
                                __builtin_memcpy(ts + L + 3 + 8 * c, &q, 8);
                            }
                        }
                    }
                    continue;
                }
                uint64_t vb, vq;
                group_text<BQ, HASN>(tile, tg, od + __umul24(r, tg.Cd), oq + __umul24(r, tg.Cq), (int32_t)(8 * wg), vb, vq);
                uint8_t* ts = out + s_off[r + 1] - (L + 8 * wg + 12);    // line start = record end - 2 L - 4, the group at + L - 8 wg - 8
                __builtin_memcpy(ts, &vb, 8);
                __builtin_memcpy(ts + L + 3, &vq, 8);
            }
        } else if (my_slot < tg.RS) {
            // fixed length: a lane keeps its group for the whole kernel (my_slot, my_wg = tid / Gf, tid % Gf) and walks over the reads
            // my_slot, my_slot + RS, ...
            const uint32_t L = g.dna_max, back = L + 8 * my_wg + 12;
            uint32_t ed = od + my_slot * tg.Cd, eq = oq + my_slot * tg.Cq;
            for (uint32_t r = my_slot; r < Rt; r += tg.RS, ed += tg.RS * tg.Cd, eq += tg.RS * tg.Cq) {
                uint64_t vb, vq;
                group_text<BQ, HASN>(tile, tg, ed, eq, (int32_t)(8 * my_wg), vb, vq);
                uint8_t* ts = out + s_off[r + 1] - back;
                __builtin_memcpy(ts, &vb, 8);
                __builtin_memcpy(ts + L + 3, &vq, 8);
            }
        }
        // ---- a lane per read: the L % 8 characters at the front of the two lines (their row bytes are read like a group's: what
        // lies before them decodes to characters that are not stored), and the separators
#ifdef DS_ABL_NOFRONT
        if (false)
#endif
        for (uint32_t i = tid; i < Rt; i += EM_THREADS) {
            const uint32_t L = s_len[i], nsym = L & 7u;
            uint8_t* ts = out + s_off[i + 1] - (2 * L + 4);
            uint8_t* tq = ts + L + 3;
            if (nsym) {
                uint64_t vb, vq;
                group_text<BQ, HASN>(tile, tg, od + __umul24(i, tg.Cd), oq + __umul24(i, tg.Cq), (int32_t)(L & ~7u), vb, vq);
                const uint32_t drop = 8u * (8u - nsym);                  // characters 0 .. nsym - 1 = the last nsym bytes of the eight
                store_low_bytes(ts, vb >> drop, nsym);
                store_low_bytes(tq, vq >> drop, nsym);
            }
            const uint16_t nl_plus = (uint16_t)('\n' | ('+' << 8));
            __builtin_memcpy(ts + L, &nl_plus, 2);
            ts[L + 2] = '\n'; tq[L] = '\n';
        }
        __syncthreads();
        // ---- QNAME lines: staging -> HBM, eight lanes per line, eight bytes a lane (aligned in LDS, wherever they fall in HBM)
#ifdef DS_ABL_NOQNAME
        if (false) {
#else
        if (staged) {
#endif
            for (uint32_t idx = tid; idx < Rt * 8u; idx += EM_THREADS) {
                const uint32_t i = idx >> 3, L = s_len[i];
                const uint64_t o = s_off[i];
                const uint32_t q = (uint32_t)(s_off[i + 1] - o) - 2 * L - 4;         // the line with its '\n'
                const uint8_t* src = tile + qst[i];
                for (uint32_t c = 8 * (idx & 7u); c < q; c += 64) {
                    const uint64_t v = *(const uint64_t*)(src + c);
                    if (c + 8 <= q) __builtin_memcpy(out + o + c, &v, 8);
                    else store_low_bytes(out + o + c, v, q - c);
                }
            }
        } else {
#ifndef DS_ABL_NOQNAME
            for (uint32_t i = tid >> 6; i < Rt; i += EM_THREADS / 64) emit_qname_direct(g, out + s_off[i], r0 + i, lane);
#endif
        }
    }
}

// The tiles emit_tile_kernel skips (their text does not fit the LDS image: reads of tens of kbp, or a tile far above the
// average the image was sized for): a wave per record, straight to HBM.  A separate kernel so that this cold code does
// not weigh on the tile kernel's registers; on ordinary inputs it reads two offsets per tile and writes nothing.
template <bool PACKED>
__global__ __launch_bounds__(EM_THREADS) void direct_tiles_kernel(EmitGeom g, TileGeom tg, std::conditional_t<PACKED, UnpackLut, NoLut> lut,
                                                                  const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual,
                                                                  const uint32_t* __restrict__ len, uint64_t n, const uint64_t* __restrict__ offsets,
                                                                  uint8_t* __restrict__ out) {
    __shared__ uint8_t l_tab[PACKED ? 768 : 4];
    const uint8_t* l_base = l_tab; const uint8_t* l_qual = l_tab + 256; const uint8_t* l_qn = l_tab + 512;
    const uint32_t tid = threadIdx.x, lane = lane_id();
    if constexpr (PACKED)
        for (uint32_t i = tid; i < 256; i += EM_THREADS) { l_tab[i] = lut.base_char[i]; l_tab[256 + i] = lut.qual_char[i]; l_tab[512 + i] = lut.qual_n_base[i]; }
    __syncthreads();
    const uint64_t R = tg.R, ntiles = (n + R - 1) / R;
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint64_t r0 = t * R;
        const uint32_t Rt = (uint32_t)((n - r0) < R ? (n - r0) : R);
        const uint64_t o0 = offsets[r0], o1 = offsets[r0 + Rt];
        const uint32_t skew = (uint32_t)((uintptr_t)(out + o0) & 15);
        if (o1 - o0 + skew <= tg.cap) continue;            // emit_tile_kernel's condition, negated
        for (uint32_t i = tid >> 6; i < Rt; i += EM_THREADS / 64) {
            if constexpr (PACKED) decode_record_direct(g, tg, l_base, l_qual, l_qn, seq, qual, len ? len[r0 + i] : g.dna_max, offsets, out, r0 + i, lane);
            else emit_record_direct(g, seq, qual, len, offsets, out, r0 + i, lane);
        }
    }
}

// read lengths of variable-length DNA rows (the sentinel's position).  W lanes scan one row a dword each per step
// (unaligned loads: rows start anywhere), the first non-zero byte is the minimum over the group.
template <int W>
__global__ __launch_bounds__(256) void row_lengths_kernel(const uint8_t* __restrict__ dna, uint64_t n, uint32_t Cd, uint32_t bd, uint32_t dmax,
                                                          uint32_t* __restrict__ len, unsigned long long* __restrict__ bad) {
    const uint64_t r = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / W;
    const uint32_t w = threadIdx.x % W;
    if (r >= n) return;                                   // whole groups leave together (256 % W == 0)
    const uint8_t* row = dna + r * Cd;
    const uint32_t nd = Cd >> 2;
    uint32_t first = Cd;
    for (uint32_t d0 = 0; d0 < nd && first == Cd; d0 += W) {      // `first` is the same in all lanes of the group
        const uint32_t d = d0 + w;
        const uint32_t v = d < nd ? load_u32_any(row + 4 * d) : 0u;
        uint32_t cand = v ? 4 * d + ((uint32_t)__builtin_ctz(v) >> 3) : Cd;
#pragma unroll
        for (int s = 1; s < W; s <<= 1) { const uint32_t o = __shfl_xor(cand, s, 64); cand = o < cand ? o : cand; }
        first = cand;
    }
    if (first == Cd) {                                     // the last Cd % 4 bytes
        uint32_t cand = Cd;
        for (uint32_t k = 4 * nd + w; k < Cd; k += W) if (row[k]) { cand = k; break; }
#pragma unroll
        for (int s = 1; s < W; s <<= 1) { const uint32_t o = __shfl_xor(cand, s, 64); cand = o < cand ? o : cand; }
        first = cand;
    }
    uint32_t L;
    const bool ok = row_length(row, Cd, first, bd, dmax, L);
    if (w == 0) {
        len[r] = L;
        if (!ok) atomicMin(bad, (unsigned long long)r);
    }
}

int make_emit_geom(uq_ctx* ctx, const char* who, const uq_emit_params* hp, const void* const* h_d_cols, const uint8_t* const* h_d_map_chars,
                   const uint32_t* const* h_d_map_offs, EmitGeom& g) {
    UQ_REQUIRE(hp->ncols >= 0 && hp->ncols <= EM_MAXCOLS && hp->prefix_len >= 0 && hp->prefix_len <= 256 && hp->suffix_len >= 0 && hp->suffix_len <= 256,
               "%s: QNAME layout out of range (<= 32 columns, prefix / suffix <= 256 bytes)", who);
    UQ_REQUIRE(hp->ncols == 0 || h_d_cols, "%s: null column table", who);
    memset(&g, 0, sizeof(g));
    memcpy(g.prefix, hp->prefix, 256); memcpy(g.suffix, hp->suffix, 256); memcpy(g.seps, hp->separators, EM_MAXCOLS);
    g.prefix_len = hp->prefix_len; g.suffix_len = hp->suffix_len; g.ncols = hp->ncols; g.dna_max = hp->dna_max;
    for (int c = 0; c < hp->ncols; ++c) {
        g.col[c] = h_d_cols[c]; g.itemsize[c] = hp->itemsize[c]; g.add[c] = hp->add[c];
        UQ_REQUIRE(g.itemsize[c] == 1 || g.itemsize[c] == 2 || g.itemsize[c] == 4 || g.itemsize[c] == 8, "%s: bad column itemsize", who);
        g.map_chars[c] = h_d_map_chars ? h_d_map_chars[c] : nullptr;
        g.map_offs[c] = h_d_map_offs ? h_d_map_offs[c] : nullptr;
        UQ_REQUIRE((g.map_chars[c] == nullptr) == (g.map_offs[c] == nullptr), "%s: mapping column needs both string tables", who);
    }
    return 0;
}

// record sizes -> offsets (exclusive scan); the total (and *d_bad, when given) come back with one synchronisation
int emit_offsets(uq_ctx* ctx, const EmitGeom& g, const uint32_t* d_len, uint64_t nreads, uint64_t* d_offsets, uint64_t* h_total,
                 const uint64_t* d_bad, uint64_t* h_bad) {
    emit_sizes_kernel<<<(uint32_t)((nreads + 255) / 256), 256, 0, ctx->stream>>>(g, d_len, nreads, d_offsets);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_scan_exclusive_u64(ctx, d_offsets, d_offsets, nreads, d_offsets + nreads));
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned, d_offsets + nreads, 8));
    if (d_bad) UQ_TRY(uq_read_back(ctx, ctx->h_pinned + 1, d_bad, 8));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *h_total = ctx->h_pinned[0];
    if (d_bad && h_bad) *h_bad = ctx->h_pinned[1];
    return 0;
}

// LDS carve for a tile of records of `avg` text bytes; packed != 0 adds the staged rows.  R = 0: nothing fits (direct path only).
size_t plan_tile(TileGeom& tg, uint64_t avg, const EmitGeom& g, bool packed) {
    const uint32_t ncols = g.ncols;
    tg.magicP = g.prefix_len ? magic_u32(g.prefix_len) : 0;
    const uint64_t text = avg + avg / 8 + 1;
    const uint64_t per = text + 8 + 4 + 4 + 2ull * ncols + (packed ? tg.Cd + tg.Cq : 0);
    uint64_t R = ((packed ? EM_BUDGET_PACKED : EM_BUDGET_TEXT) - 256) / per;
    if (R > EM_RMAX) R = EM_RMAX;
    if (packed && R > EM_RMAX - 1) R = EM_RMAX - 1;           // wave 0 holds the tile's R + 1 record offsets
    if (packed) while (R > 0 && (R * tg.Cd + 48 > DE_NVD * EM_THREADS * 16u || R * tg.Cq + 48 > DE_NVQ * EM_THREADS * 16u)) --R;
    if (packed && tg.fa.fast && !tg.variable && tg.RS && R > tg.RS) R -= R % tg.RS;      // whole steps of the fixed-length chunk loop
    const bool fits = R >= 1;
    if (!fits) R = 1;
    tg.R = (uint32_t)R;
    tg.P = EM_THREADS / tg.R;
    tg.cap = fits ? (uint32_t)(R * text + 64) & ~15u : 0u;
    uint32_t off = fits ? tg.cap + 32 : 32;
    auto carve = [&](uint32_t bytes) { uint32_t o = off; off += (bytes + 15) & ~15u; return o; };
    tg.o_off = carve((tg.R + 1) * 8); tg.o_len = carve(tg.R * 4); tg.o_cum = carve((tg.R + 2) * 4);
    tg.o_flen = carve(tg.R * (ncols ? ncols : 1) * 2);
    tg.o_ind = tg.o_inq = 0;
    if (packed && fits) { tg.o_ind = carve(tg.R * tg.Cd + 64); tg.o_inq = carve(tg.R * tg.Cq + 64); }      // (the windows of a line's edge pieces reach up to 40 bytes beyond the last row)
    return off;
}

// LDS carve of decode_stream_kernel.  R = 0: the rows of a single read do not fit (emit_tile_kernel / the direct path take over).
size_t plan_stream(StreamGeom& sg, const EmitGeom& g, const uq_unpack_params* up, const FastAlphabet& fa) {
    memset(&sg, 0, sizeof(sg));
    sg.fa = fa; sg.bq = up->bits_per_quality; sg.Cd = up->dna_bytes_per_row; sg.Cq = up->quality_bytes_per_row; sg.variable = up->variable ? 1 : 0;
    sg.Gf = sg.variable ? 0 : up->dna_max / 8;
    sg.magicGf = sg.Gf ? magic_u32(sg.Gf) : 0;
    sg.RS = sg.Gf ? EM_THREADS / sg.Gf : 0;
    // a QNAME line's share of the staging area: what its fields can take at most (integers by their width, 16 characters for a
    // mapping string); a tile that needs more writes its QNAME lines a wave per line
    uint32_t qp = g.prefix_len + g.suffix_len + 1;
    for (uint32_t c = 0; c < g.ncols; ++c) qp += 1 + (g.map_chars[c] ? 16u : g.itemsize[c] == 1 ? 4u : g.itemsize[c] == 2 ? 6u : g.itemsize[c] == 4 ? 11u : 20u);
    qp = (qp + 7) & ~7u;
    const uint32_t ncols = g.ncols ? g.ncols : 1;
    auto a16 = [](uint32_t b) { return (b + 15) & ~15u; };
    const uint64_t per = qp + 2ull * (8 + 4 + 4 + 4 + 2 * ncols) + sg.Cd + sg.Cq;
    uint64_t R = ((DS_OCC == 4 ? EM_BUDGET_PACKED : 160u * 1024u / DS_OCC - 4096u) - 1024) / per;
    if (R > 63) R = 63;                                      // wave 0 scans the tile's offsets and lengths, one more offset than reads
    while (R > 0 && (R * sg.Cd + 48 > DE_NVD * EM_THREADS * 16u || R * sg.Cq + 48 > DE_NVQ * EM_THREADS * 16u)) --R;
    if (sg.RS && R > sg.RS) R -= R % sg.RS;                  // whole steps of the fixed-length group loop
    if (R == 0) return 0;
    sg.R = (uint32_t)R;
    sg.qcap = a16(sg.R * qp);
    uint32_t off = 0;
    sg.s_off = off; off += a16((sg.R + 1) * 8);
    sg.s_len = off; off += a16(sg.R * 4);
    sg.s_cum = off; off += a16((sg.R + 2) * 4);
    sg.s_qst = off; off += a16((sg.R + 2) * 4);
    sg.s_flen = off; off += a16(sg.R * ncols * 2);
    sg.set = off;
    sg.o_set = sg.qcap;
    sg.o_ind = sg.o_set + 2 * sg.set;
    sg.o_inq = sg.o_ind + a16(sg.R * sg.Cd + 32);
    return sg.o_inq + a16(sg.R * sg.Cq + 32);
}

uint32_t tile_blocks(uint64_t tiles, size_t lds, uint64_t by_registers) {
    uint64_t per_cu = (160u * 1024u) / (lds + 3072);       // + the static tables
    if (per_cu > by_registers) per_cu = by_registers;      // workgroups per CU the kernel's registers allow
    if (per_cu < 1) per_cu = 1;
    return (uint32_t)(tiles < UQ_NUM_CU * per_cu ? tiles : UQ_NUM_CU * per_cu);
}
uint32_t direct_blocks(uint64_t tiles) { return (uint32_t)(tiles < UQ_NUM_CU * 8ull ? tiles : UQ_NUM_CU * 8ull); }
}  // namespace

extern "C" int uq_emit_fastq(uq_ctx* ctx, const uq_emit_params* hp, const void* const* h_d_cols, const uint8_t* const* h_d_map_chars,
                             const uint32_t* const* h_d_map_offs, const uint8_t* d_seq, const uint8_t* d_qual, const uint32_t* d_len,
                             uint64_t nreads, uint64_t* d_offsets, uint8_t* d_out, uint64_t capacity, uint64_t* h_total) {
    UQ_REQUIRE(ctx && hp && d_offsets && h_total, "uq_emit_fastq: null argument");
    *h_total = 0;
    EmitGeom g;
    UQ_TRY(make_emit_geom(ctx, "uq_emit_fastq", hp, h_d_cols, h_d_map_chars, h_d_map_offs, g));
    if (nreads == 0) return 0;
    UQ_REQUIRE(d_seq && d_qual && d_len, "uq_emit_fastq: null buffer");
    UQ_TRY(emit_offsets(ctx, g, d_len, nreads, d_offsets, h_total, nullptr, nullptr));
    if (!d_out) return 0;                       // size query
    UQ_REQUIRE(capacity >= *h_total, "uq_emit_fastq: output buffer too small (%llu < %llu)", (unsigned long long)capacity, (unsigned long long)*h_total);
    // tile = R records sized from the average record so that a typical tile fits the LDS image
    TileGeom tg;
    memset(&tg, 0, sizeof(tg));
    const size_t lds = plan_tile(tg, *h_total / nreads + 1, g, false);
    const uint64_t tiles = (nreads + tg.R - 1) / tg.R;
    emit_tile_kernel<false><<<tile_blocks(tiles, lds, 5), EM_THREADS, lds, ctx->stream>>>(g, tg, NoLut{}, d_seq, d_qual, d_len, nreads, d_offsets, d_out);
    UQ_LAUNCH_CHECK();
    direct_tiles_kernel<false><<<direct_blocks(tiles), EM_THREADS, 0, ctx->stream>>>(g, tg, NoLut{}, d_seq, d_qual, d_len, nreads, d_offsets, d_out);
    UQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int uq_decode_fastq(uq_ctx* ctx, const uq_emit_params* hp, const uq_unpack_params* up, const void* const* h_d_cols,
                               const uint8_t* const* h_d_map_chars, const uint32_t* const* h_d_map_offs, const uint8_t* d_dna,
                               const uint8_t* d_qual, uint64_t nreads, uint32_t* d_len, uint64_t* d_offsets, uint64_t* d_bad,
                               uint8_t* d_out, uint64_t capacity, uint64_t* h_total, uint64_t* h_bad) {
    UQ_REQUIRE(ctx && hp && up && d_offsets && d_bad && h_total && h_bad, "uq_decode_fastq: null argument");
    *h_total = 0; *h_bad = ~0ull;
    EmitGeom g;
    UQ_TRY(make_emit_geom(ctx, "uq_decode_fastq", hp, h_d_cols, h_d_map_chars, h_d_map_offs, g));
    UQ_REQUIRE(up->bits_per_base >= 1 && up->bits_per_base <= 8 && up->bits_per_quality >= 1 && up->bits_per_quality <= 8,
               "uq_decode_fastq: bits per symbol must be 1..8");
    UQ_REQUIRE(up->dna_max >= 1 && up->dna_max == hp->dna_max, "uq_decode_fastq: dna_max must be positive and the same in both parameter blocks");
    TileGeom tg;
    memset(&tg, 0, sizeof(tg));
    tg.bd = up->bits_per_base; tg.bq = up->bits_per_quality; tg.Cd = up->dna_bytes_per_row; tg.Cq = up->quality_bytes_per_row;
    const uint32_t variable = up->variable ? 1 : 0, Lv = up->dna_max + variable;
    UQ_REQUIRE(tg.Cd == (tg.bd * Lv + 7) / 8 && tg.Cq == (tg.bq * Lv + 7) / 8, "uq_decode_fastq: row bytes do not match the geometry");
    tg.G = (up->dna_max + 7) / 8;
    tg.magicG = magic_u32(tg.G);
    tg.fa = fast_alphabet(up);
    tg.variable = variable; tg.NC = (up->dna_max + 14) / 8; tg.K = 1;
    // pieces of DE_K chunks per lane where a line has enough of them to keep the lanes busy (fixed lengths of 66 characters and more)
    if (tg.fa.fast && !variable && tg.bq >= 2 && tg.bq <= 6 && tg.NC >= 2 * DE_K) { tg.K = DE_K; tg.NC = (tg.NC + DE_K - 1) / DE_K; }
    tg.magicNC = magic_u32(tg.NC);
    tg.RS = EM_THREADS / tg.NC;
    if (nreads == 0) return 0;
    UQ_REQUIRE(d_dna && d_qual && (d_len || !variable), "uq_decode_fastq: null buffer");
    const uint32_t* lens = variable ? d_len : nullptr;       // fixed-length tables: every read is dna_max long
    if (!d_out) {                                            // first call: lengths, record offsets, total size
        UQ_CHECK_HIP(hipMemsetAsync(d_bad, 0xFF, 8, ctx->stream));
        if (variable) {
            if (tg.Cd > 512) row_lengths_kernel<64><<<(uint32_t)((nreads + 3) / 4), 256, 0, ctx->stream>>>(d_dna, nreads, tg.Cd, tg.bd, up->dna_max, d_len, (unsigned long long*)d_bad);
            else row_lengths_kernel<8><<<(uint32_t)((nreads + 31) / 32), 256, 0, ctx->stream>>>(d_dna, nreads, tg.Cd, tg.bd, up->dna_max, d_len, (unsigned long long*)d_bad);
            UQ_LAUNCH_CHECK();
        }
        UQ_TRY(emit_offsets(ctx, g, lens, nreads, d_offsets, h_total, d_bad, h_bad));
        return 0;
    }
    // second call: d_len / d_offsets hold what the first one left; the total is its last offset
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned, d_offsets + nreads, 8));
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned + 1, d_bad, 8));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *h_total = ctx->h_pinned[0]; *h_bad = ctx->h_pinned[1];
    UQ_REQUIRE(capacity >= *h_total, "uq_decode_fastq: output buffer too small (%llu < %llu)", (unsigned long long)capacity, (unsigned long long)*h_total);
    // the lookup-free alphabet, variable lengths: groups straight to HBM (10 M reads of 36 - 301 bp: 2.85 ms against 3.7 through the text
    // image, whose chunk arithmetic then pays a search per chunk; fixed-length reads are faster through the image, 1.90 against
    // 2.03 ms for 10 M x 150 bp: its stores are whole cache lines, the stream's pieces cost 20 % more HBM write traffic)
    if (tg.fa.fast && variable) {
        StreamGeom sg;
        const size_t lds = plan_stream(sg, g, up, tg.fa);
        if (lds) {
            const uint64_t tiles = (nreads + sg.R - 1) / sg.R;
            const uint32_t tb = tile_blocks(tiles, lds, DS_OCC);
#define UQ_STREAM_CASE(Q) \
            case Q: if (sg.fa.has_n) decode_stream_kernel<Q, true, DS_K><<<tb, EM_THREADS, lds, ctx->stream>>>(g, sg, d_dna, d_qual, lens, nreads, d_offsets, d_out); \
                    else decode_stream_kernel<Q, false, DS_K><<<tb, EM_THREADS, lds, ctx->stream>>>(g, sg, d_dna, d_qual, lens, nreads, d_offsets, d_out); \
                    break;
            switch (sg.bq) {
                UQ_STREAM_CASE(2) UQ_STREAM_CASE(3) UQ_STREAM_CASE(4) UQ_STREAM_CASE(5) UQ_STREAM_CASE(6)
                default: decode_stream_kernel<><<<tb, EM_THREADS, lds, ctx->stream>>>(g, sg, d_dna, d_qual, lens, nreads, d_offsets, d_out);
            }
#undef UQ_STREAM_CASE
            UQ_LAUNCH_CHECK();
            return 0;
        }
    }
    UnpackLut lut;
    memcpy(lut.base_char, up->base_char, 256); memcpy(lut.qual_char, up->qual_char, 256); memcpy(lut.qual_n_base, up->qual_n_base, 256);
    const size_t lds = plan_tile(tg, *h_total / nreads + 1, g, true);
    const uint64_t tiles = (nreads + tg.R - 1) / tg.R;
    const uint32_t tb = tile_blocks(tiles, lds, 4);
    const bool special = tg.fa.fast && !variable && tg.RS != 0 && tg.cap != 0;      // the instances with compile-time quality width
#define UQ_EMIT_CASE(Q) \
    case Q: if (tg.K > 1) { \
                if (tg.fa.has_n) emit_tile_kernel<true, Q, true, DE_K><<<tb, EM_THREADS, lds, ctx->stream>>>(g, tg, lut, d_dna, d_qual, lens, nreads, d_offsets, d_out); \
                else emit_tile_kernel<true, Q, false, DE_K><<<tb, EM_THREADS, lds, ctx->stream>>>(g, tg, lut, d_dna, d_qual, lens, nreads, d_offsets, d_out); \
            } else if (tg.fa.has_n) emit_tile_kernel<true, Q, true><<<tb, EM_THREADS, lds, ctx->stream>>>(g, tg, lut, d_dna, d_qual, lens, nreads, d_offsets, d_out); \
            else emit_tile_kernel<true, Q, false><<<tb, EM_THREADS, lds, ctx->stream>>>(g, tg, lut, d_dna, d_qual, lens, nreads, d_offsets, d_out); \
            break;
    switch (special ? tg.bq : 0u) {
        UQ_EMIT_CASE(2) UQ_EMIT_CASE(3) UQ_EMIT_CASE(4) UQ_EMIT_CASE(5) UQ_EMIT_CASE(6)
        default: emit_tile_kernel<true><<<tb, EM_THREADS, lds, ctx->stream>>>(g, tg, lut, d_dna, d_qual, lens, nreads, d_offsets, d_out);
    }
#undef UQ_EMIT_CASE
    UQ_LAUNCH_CHECK();
    direct_tiles_kernel<true><<<direct_blocks(tiles), EM_THREADS, 0, ctx->stream>>>(g, tg, lut, d_dna, d_qual, lens, nreads, d_offsets, d_out);
    UQ_LAUNCH_CHECK();
    return 0;
}
