// pack.hip -- the per-read DNA / QUAL bit packers.
// Replaces `encoder_fixed` (uq.py:108-182) and `encoder_variable` (uq.py:188-254).
//
// Row semantics: value = sum_j code(read[j]) << (b * (L-1-j)), plus, for
// variable-length input, a sentinel 1 << (b * L); stored big-endian and right-aligned in C bytes.
// A base byte that is not in `bases` takes DNA code 0 and the quality code N_qual[byte] (uq.py:151-153).
//
// Kernel `pack_tile_kernel<BD, BQ, NTRICK>` (the hot one): one workgroup packs a tile of R consecutive
// reads; bits per symbol are template constants so every shift is an immediate.
//   A  the tile's records are one contiguous byte span of the FASTQ stream: it is copied to LDS with
//      fully coalesced 16-byte loads, all issued before the first LDS write (the QNAME and '+' lines
//      ride along; they share cache lines with the payload).
//   B  a thread owns 8 consecutive symbols of one read, for BOTH streams: it pulls the 8 base and 8
//      quality characters out of LDS (two aligned ds_read_b64 each, funnel-shifted), maps them to codes
//      through LUTs held in LDS (the N-trick needs the base to choose the quality code), and Horner-
//      packs them with v_lshl_or: 8 symbols of b bits = exactly b whole output bytes, so no thread
//      ever shares a byte with another -- no atomics, no ballots.  The sentinel is code 1 at index L.
//   C  the packed tile is a contiguous span of the output table: stored with 16-byte coalesced stores.
// Algorithmic HBM bytes per read: record bytes read + C_dna + C_qual written (+32 B of line offsets).
//
// Kernel `pack_carry_kernel`: the Q9 corner (an N quality code equal to 2^b, uq.py:493-494) makes
// the reference's `+=` carry into the neighbouring symbol; that case is packed by a plain
// thread-per-read big-integer addition so the bytes still match.
#include "common.h"
#include "swar.h"
#include "lines.h"
#include "histo.h"

namespace {
constexpr int PK_THREADS = 256;

struct PackLut {
    int16_t dna_code[256];
    int16_t qual_code[256];
    int16_t n_qual[256];
};

struct PackGeom {
    uint32_t R;            // reads per tile (sized from the average record)
    uint32_t Rs;           // reads per piece when a tile's records do not fit the stage (sized from the longest record)
    uint32_t Cd, Cq;       // bytes per row
    uint32_t G;            // 8-symbol groups per row = ceil((dna_max + variable) / 8)
    uint32_t variable;
    uint32_t stage_bytes;  // size of the staging region (multiple of 16)
    uint32_t out_bytes;    // size of the output tiles region (multiple of 16)
    uint32_t magicG;       // for division by G
    uint32_t dna_max;      // longest read the geometry was sized for
    uint32_t fill_d, fill_q;  // the characters with code 0 (bases[0], qualities[0]) replicated in 4 bytes
    uint32_t P, magicP;    // lanes per read in phase B, and the magic for tid / P
    // fast path (bases == "ACGT", qualities a contiguous ASCII range below 128, at most one N-trick base)
    uint32_t q_addlo, q_addhi;   // (0x80 - qmin) and (0x80 - qmin - nq), replicated in 4 bytes
    uint32_t n_char, n_code;     // the N-trick base and its quality code, replicated in 4 bytes
    uint32_t n_qchar;            // the quality CHARACTER that code stands for (the one the N-trick base occurs with), replicated
    // 3-bit bases on the fast path: (character >> h_shift) & 7 is different for every base of the alphabet (found by the host);
    // i2c = code of each such index, c2c = character of each code, eight bytes each (what v_perm_b32 selects from)
    uint32_t h_shift, i2c_lo, i2c_hi, c2c_lo, c2c_hi;
};

// four ACGT characters -> four 2-bit codes (A0 C1 G2 T3), one per byte
__device__ __forceinline__ uint32_t acgt_codes(uint32_t w) { return ((w ^ (w >> 1)) >> 1) & 0x03030303u; }
// the characters those codes stand for (v_perm_b32 picks bytes of "ACGT")
__device__ __forceinline__ uint32_t acgt_chars(uint32_t codes) { return __builtin_amdgcn_perm(0u, 0x54474341u, codes); }
// four B-bit codes (one per byte, first character in byte 0) -> 4*B bits, first character most significant
// (the bytes hold codes below 2^B.  v_dot4_u32_u8 weighs the four bytes in one full-rate instruction: a 32-bit multiply is quarter rate, and the
// shift-and-mask form is eight instructions per four 6-bit codes)
template <int B>
__device__ __forceinline__ uint32_t pack4(uint32_t x) {
    if (B == 8) return __builtin_amdgcn_perm(0u, x, 0x00010203u);
#ifndef PK_OLD_PACK4
    if (B <= 2) return __builtin_amdgcn_udot4(x, (1u << (3 * B)) | (1u << (2 * B + 8)) | (1u << (B + 16)) | (1u << 24), 0u, false);
    {
        const uint32_t hi = __builtin_amdgcn_udot4(x, (1u << B) | (1u << 8), 0u, false);
        return __builtin_amdgcn_udot4(x, (1u << (B + 16)) | (1u << 24), hi << (2 * B), false);        // (the low pair added onto the shifted high pair)
    }
#endif
    if (B == 2) return (x * 0x40100401u) >> 24;
    const uint32_t c0 = x & ((1u << B) - 1), c1 = (x >> 8) & ((1u << B) - 1), c2 = (x >> 16) & ((1u << B) - 1), c3 = x >> 24;
    return (((((c0 << B) | c1) << B) | c2) << B) | c3;
}

// byte i (from the least significant) of a group's B bytes = a0 << 4 B | a1, the two halves of four codes each: with an even B the bytes never
// straddle the halves -- no 64-bit shifts, and the byte stores take bits 0-7 or (d16_hi) 16-23 of a register as they are
template <int B>
__device__ __forceinline__ uint32_t group_byte(uint32_t a0, uint32_t a1, int i) {
#ifndef PK_OLD_STORE
    if constexpr ((4 * B) % 8 == 0) return i < B / 2 ? a1 >> (8 * i) : a0 >> (8 * (i - B / 2));
    else
#endif
    return (uint32_t)((((uint64_t)a0 << (4 * B)) | a1) >> (8 * i));
}

// LDS tile (16-byte aligned) -> global span, with the widest stores the destination alignment allows.
// CLEAR: every byte is zeroed right after the lane has read it (the variable-length dealing leaves the empty upper parts of the rows to that).
template <bool CLEAR = false>
__device__ __noinline__ void store_tile_any(uint8_t* gdst, uint8_t* lsrc, uint32_t nb, uint32_t tid) {
    const uint32_t mis = (uint32_t)(uintptr_t)gdst;
    if ((mis & 15) == 0) {
        const uint32_t nv = nb >> 4;
        for (uint32_t i = tid; i < nv; i += 256) { ((uint4*)gdst)[i] = ((const uint4*)lsrc)[i]; if (CLEAR) ((uint4*)lsrc)[i] = make_uint4(0, 0, 0, 0); }
        for (uint32_t i = (nv << 4) + tid; i < nb; i += 256) { gdst[i] = lsrc[i]; if (CLEAR) lsrc[i] = 0; }
    } else if ((mis & 7) == 0) {
        const uint32_t nv = nb >> 3;
        for (uint32_t i = tid; i < nv; i += 256) { ((uint2*)gdst)[i] = ((const uint2*)lsrc)[i]; if (CLEAR) ((uint2*)lsrc)[i] = make_uint2(0, 0); }
        for (uint32_t i = (nv << 3) + tid; i < nb; i += 256) { gdst[i] = lsrc[i]; if (CLEAR) lsrc[i] = 0; }
    } else if ((mis & 3) == 0) {
        const uint32_t nv = nb >> 2;
        for (uint32_t i = tid; i < nv; i += 256) { ((uint32_t*)gdst)[i] = ((const uint32_t*)lsrc)[i]; if (CLEAR) ((uint32_t*)lsrc)[i] = 0; }
        for (uint32_t i = (nv << 2) + tid; i < nb; i += 256) { gdst[i] = lsrc[i]; if (CLEAR) lsrc[i] = 0; }
    } else {
        for (uint32_t i = tid; i < nb; i += 256) { gdst[i] = lsrc[i]; if (CLEAR) lsrc[i] = 0; }
    }
}
// The usual case inline -- R is a multiple of 16 wherever that wastes little, so a tile's rows start 16-byte aligned: a call costs two full
// waits (the callee waits for everything in flight on entry -- the next tile's bytes -- and for its own stores before it returns).
template <bool CLEAR = false>
__device__ __forceinline__ void store_tile(uint8_t* gdst, uint8_t* lsrc, uint32_t nb, uint32_t tid) {
    if ((((uint32_t)(uintptr_t)gdst) & 15u) == 0 && (nb & 15u) == 0) {
#pragma unroll 1
        for (uint32_t i = tid; i < (nb >> 4); i += 256) { ((uint4*)gdst)[i] = ((const uint4*)lsrc)[i]; if (CLEAR) ((uint4*)lsrc)[i] = make_uint4(0, 0, 0, 0); }
    } else store_tile_any<CLEAR>(gdst, lsrc, nb, tid);
}

// ---- the QNAME phase of the fused pack + statistics kernel (uq_pack_stats_qname; SURVEY.md 8 row f1 inside rows a3 / a4).
// The tile's QNAME lines are in the stage already; the lanes of ONE wave take a read each, check the line against the guessed
// layout (qname_fused.hip: prefix = line1[:plen], suffix = line1[l1len - slen:], ordered separators) and parse its fields
// (uq.py:560-565 split, 724-726 int()): value of field c of read `row` -> vals[c * pitch + row].  Whatever does not conform
// raises a flag (include/uqhip.h lists them); the host then runs the exact passes of qname_dev.hip instead.
struct QnLds {
    uint8_t line1[256 + 16];        // zero padded: windows may read past the line
    uint8_t inset[256];
    uint8_t seps[32];
    uint32_t vmin[UQ_QF_MAXC], vmax[UQ_QF_MAXC];
    uint32_t flags, on, plen, slen, nsep, l1len;
    uint32_t sepset;                // the (at most four) distinct separator characters, one per byte (unused bytes repeat the first)
    uint32_t seps_lo, seps_hi;      // seps[0..7] as two dwords
    uint32_t pow10[10];
};

// high bit of every byte of w that equals the byte replicated in c4
__device__ __forceinline__ uint32_t eq_bytes(uint32_t w, uint32_t c4) {
    const uint32_t x = w ^ c4;
    return ~((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x)) & 0x80808080u;
}
// high bit of every byte of w that is NOT an ASCII digit
__device__ __forceinline__ uint32_t nondigit_bytes(uint32_t w) {
    const uint32_t lo7 = w & 0x7F7F7F7Fu;
    return (~(lo7 + 0x50505050u) | (lo7 + 0x46464646u) | w) & 0x80808080u;
}
// 0xFF in bytes k < n of a dword (n >= 4: all)
__device__ __forceinline__ uint32_t low_bytes(uint32_t n) { return n >= 4 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu << (8 * n)); }

// 12 consecutive bytes at LDS byte offset `o` (any alignment, may be slightly negative) as three dwords
__device__ __forceinline__ void lds_window12(const uint8_t* base, int32_t o, uint32_t& w0, uint32_t& w1, uint32_t& w2) {
    const uint32_t* p = (const uint32_t*)(base + (o & ~3));
    const uint32_t a = p[0], b = p[1], c = p[2], d = p[3];
    const uint32_t sh = (uint32_t)o & 3u;
    w0 = __builtin_amdgcn_alignbyte(b, a, sh);
    w1 = __builtin_amdgcn_alignbyte(c, b, sh);
    w2 = __builtin_amdgcn_alignbyte(d, c, sh);
}

// Two steps, both without a branch per byte.  (1) The middle of the line is scanned eight bytes at a time: separators and non-digits
// are found byte-parallel, the separators' positions and characters are collected (a short loop per window, one turn per
// separator).  (2) Field by field -- every lane of the wave closes field c at the same time, so the column stores are coalesced --
// the twelve bytes that end where the field ends are fetched, the field's digits kept, the value built with nine multiply-adds.
// (A loop over the bytes that closed a field wherever a lane met a separator diverged at nearly every byte: ~1 200 instructions
// per tile for the wave, the long pole of the tile; this form: ~500.)
__device__ __forceinline__ void qname_tile(const uint8_t* stage, const uint32_t* meta, uint32_t q, QnLds* s, uint32_t* __restrict__ vals, uint64_t pitch,
                                           uint64_t row) {
    const uint32_t qs = meta[4 * q], ql = meta[4 * q + 1] - qs - 1;
    const uint32_t plen = s->plen, slen = s->slen, nsep = s->nsep, l1len = s->l1len;
    uint32_t f = 0;
    if (ql > 255) f = 2;
    else if (ql < plen + slen) f = 8;
    else {
        // prefix and suffix: eight bytes at a time against line 1's (the line-1 side is the same address for every lane: a broadcast)
        for (uint32_t j = 0; j < plen; j += 8) {
            uint32_t a0, a1;
            lds_window8(stage, (int32_t)(qs + j), a0, a1);
            const uint32_t l0 = *(const uint32_t*)(s->line1 + j), l1 = *(const uint32_t*)(s->line1 + j + 4);
            const uint32_t left = plen - j;
            if (((a0 ^ l0) & low_bytes(left)) | ((a1 ^ l1) & (left > 4 ? low_bytes(left - 4) : 0u))) f |= 32u;
        }
        for (uint32_t j = 0; j < slen; j += 8) {
            uint32_t a0, a1, l0, l1;
            lds_window8(stage, (int32_t)(qs + ql - slen + j), a0, a1);
            lds_window8(s->line1, (int32_t)(l1len - slen + j), l0, l1);
            const uint32_t left = slen - j;
            if (((a0 ^ l0) & low_bytes(left)) | ((a1 ^ l1) & (left > 4 ? low_bytes(left - 4) : 0u))) f |= 64u;
        }
        // (1) separators of the middle [plen, end): positions (relative to the line) and characters, eight of each at most
        const uint32_t end = ql - slen;
        const uint32_t sepset = s->sepset;
        const uint32_t c0 = 0x01010101u * (sepset & 0xFFu), c1 = 0x01010101u * ((sepset >> 8) & 0xFFu), c2 = 0x01010101u * ((sepset >> 16) & 0xFFu),
                       c3 = 0x01010101u * (sepset >> 24);
        unsigned long long epos = 0, eseq = 0;
        uint32_t found = 0, differs = 0, bad = 0;
        for (uint32_t w = plen; w < ql; w += 8) {
            uint32_t a0, a1, l0, l1;
            lds_window8(stage, (int32_t)(qs + w), a0, a1);
            lds_window8(s->line1, (int32_t)w, l0, l1);
            const uint32_t inl = ql - w;                                   // bytes of the line in this window (>= 1)
            differs |= ((a0 ^ l0) & low_bytes(inl)) | ((a1 ^ l1) & (inl > 4 ? low_bytes(inl - 4) : 0u));   // (equal to line 1 so far?)
            if (w >= end) continue;
            const uint32_t inm = end - w;                                  // ... of the middle
            const uint32_t v0 = low_bytes(inm), v1 = inm > 4 ? low_bytes(inm - 4) : 0u;
            const uint32_t sm0 = (eq_bytes(a0, c0) | eq_bytes(a0, c1) | eq_bytes(a0, c2) | eq_bytes(a0, c3)) & v0;
            const uint32_t sm1 = (eq_bytes(a1, c0) | eq_bytes(a1, c1) | eq_bytes(a1, c2) | eq_bytes(a1, c3)) & v1;
            bad |= (nondigit_bytes(a0) & ~sm0 & v0) | (nondigit_bytes(a1) & ~sm1 & v1);
            unsigned long long m = (unsigned long long)sm0 | ((unsigned long long)sm1 << 32);
            const unsigned long long bytes = (unsigned long long)a0 | ((unsigned long long)a1 << 32);
            while (m) {
                const uint32_t k = ((uint32_t)__ffsll((long long)m) - 1u) >> 3;      // byte of the window
                m &= m - 1;
                if (found < 8) {
                    epos |= (unsigned long long)(w + k) << (8 * found);
                    eseq |= ((bytes >> (8 * k)) & 0xFFull) << (8 * found);
                }
                ++found;
            }
        }
        if (bad) f |= 16u;
        const unsigned long long want = (unsigned long long)s->seps_lo | ((unsigned long long)s->seps_hi << 32);
        if (found != nsep || eseq != want) f |= 1u;
        // (2) the fields: field c = [start, stop) with stop = the c-th separator (or the end of the middle)
        uint32_t start = plen;
        for (uint32_t c = 0; c <= nsep && c < UQ_QF_MAXC; ++c) {
            uint32_t stop = c == nsep ? end : (uint32_t)(epos >> (8 * c)) & 0xFFu;
            if (stop < start || stop > end) stop = start;                  // (only with bit0 up: keeps the window inside the line)
            const uint32_t nd = stop - start;
            uint32_t w0, w1, w2;
            lds_window12(stage, (int32_t)(qs + stop) - 12, w0, w1, w2);    // bytes [stop - 12, stop)
            const uint32_t keep = nd > 9 ? 9u : nd;                       // digits wanted: the last `keep` of the twelve bytes
            // per dword: digits of the bytes that belong to the field, zero in the others (leading zeros add nothing below)
            const uint32_t k2 = low_bytes(keep >= 4 ? 0u : 4u - keep), k1 = keep <= 4 ? 0xFFFFFFFFu : (keep >= 8 ? 0u : low_bytes(8u - keep)),
                           k0 = keep <= 8 ? 0xFFFFFFFFu : low_bytes(12u - keep);
            const uint32_t d0 = w0 & 0x0F0F0F0Fu & ~k0, d1 = w1 & 0x0F0F0F0Fu & ~k1, d2 = w2 & 0x0F0F0F0Fu & ~k2;    // ('0'..'9' = 0x30..0x39)
            uint32_t mag = d0 >> 24;                                       // byte 3: the ninth digit from the end
            mag = mag * 10 + (d1 & 0xFFu); mag = mag * 10 + ((d1 >> 8) & 0xFFu); mag = mag * 10 + ((d1 >> 16) & 0xFFu); mag = mag * 10 + (d1 >> 24);
            mag = mag * 10 + (d2 & 0xFFu); mag = mag * 10 + ((d2 >> 8) & 0xFFu); mag = mag * 10 + ((d2 >> 16) & 0xFFu); mag = mag * 10 + (d2 >> 24);
            if (nd > 9) f |= 4u;
            else if (nd == 0 || (nd > 1 && mag < s->pow10[nd - 1])) f |= 16u;        // empty, or a leading zero ('007')
            vals[c * pitch + row] = mag;
            if (mag < s->vmin[c]) atomicMin(&s->vmin[c], mag);
            if (mag > s->vmax[c]) atomicMax(&s->vmax[c], mag);
            start = stop + 1;
        }
        if (ql < l1len && !differs && !(f & 32u)) f |= 128u;                      // a proper prefix of line 1
        if (ql < l1len && s->line1[l1len - ql] == s->line1[0]) f |= 128u;         // could be a proper suffix of it: the exact pass decides
    }
    if (f) atomicOr(&s->flags, f);
}

constexpr int PK_NV = 5;   // 16-byte loads per lane per tile: a tile spans at most PK_NV * 256 * 16 = 20 KiB
constexpr int PK_NV_STATS = 4;   // the fused pack + statistics kernel keeps a few KiB of LDS for its count table: 16 KiB tiles, same occupancy
// Its counts are taken on the CODES the lookup-free conversion has just produced (bin = base code << 6 | quality code: two
// instructions per four pairs instead of re-deriving bins from the characters), into an LDS table (3-bit bases: four copies);
// code 0 fill pairs of a read's partial top group and the N-trick positions (counted as their substitute there) are taken off
// again at the flush, the N-trick base's own pairs go to a 128-entry table by quality character.  A symbol outside the guessed
// alphabets is not counted at all: the statistics are flagged incomplete (the guess is wrong then anyway).
// The table is a STATIC LDS array (address 0 of the workgroup's LDS, known to the compiler), four copies chosen by the lane.  (Tried, round 4: ONE
// copy for the 2-bit bases -- a counter's address is then `byte of the bin word << 2` plus an immediate, one SDWA shift per atomic, 10 VALU
// instructions per group of eight pairs instead of 20; the lanes that meet in a counter cost more than that saves, 1.66 -> 1.71 ms: PK_ONE_COPY.)
#ifdef PK_ONE_COPY
constexpr int pks_copies(int bd, bool four_wg) { return bd == 3 ? 4 : 1; }
#else
// (the forms built for four workgroups per CU -- QNAME phase or N-trick -- have the LDS for eight copies of the 256 bins: 1.56 -> 1.54 ms, two copies 1.60;
// the others would drop from four workgroups per CU to three)
constexpr int pks_copies(int bd, bool four_wg) { return bd == 2 && four_wg ? 8 : 4; }
#endif
constexpr uint32_t pks_bins(int bd) { return bd == 3 ? 512u : 256u; }     // bin = base code << 6 | quality code (3-bit bases: nine bits)
constexpr uint32_t pks_words(int bd, bool four_wg) { return pks_bins(bd) * pks_copies(bd, four_wg) + 128; }

// The queued forms take their line starts from the census's lists instead of an expanded index (lines.h): a record per pack tile (TileRec): the start of
// the tile's first record (entry `ntiles`: the end of the last record), where the newline in front of it sits (census tile,
// slot) and how many newlines that tile and its two successors hold -- left by tile_origin_kernel, one binary search per pack tile.  With
// them a lane finds its list entry by arithmetic: ONE load per line start, nothing that waits for another load inside the prefetch (a walk
// with dependent loads there stalls on the tile's bytes requested just before it: 1.62 -> 1.94 ms).
// start of the tile's first record; T = census tile of the newline in front of it, A = its slot + 1 | newlines of T << 16, B = newlines of T + 1 | of T + 2 << 16
struct TileRec { uint64_t start; uint32_t T, A, B, pad; };
struct PackLists {
    CensusView cv;                             // cv.list == nullptr: the expanded index (`ls`) is used
    const uint32_t* over;                      // the census's overflow word: a tile held more newlines than its list (the lists are not usable)
};

__global__ __launch_bounds__(256) void tile_origin_kernel(CensusView cv, const unsigned long long* __restrict__ d_async, const uint32_t* __restrict__ over, uint64_t first,
                                                          uint64_t n, uint64_t R, uint64_t entries, TileRec* __restrict__ rec) {
    const uint64_t tt = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (tt >= entries) return;
    uint64_t nlines = 4 * (first + n);
    if (d_async) {                                  // the queued census: what it counted (pack_tile_kernel clips the same way)
        nlines = d_async[0];
        const uint64_t have = nlines / 4;
        if (have < n) n = have;
        if (d_async[1] || *over) n = 0;
    }
    uint64_t r = tt * R;
    if (r > n) r = n;
    uint32_t T, kk;
    cv_locate(cv, (int64_t)(4 * (first + r)) - 1, T, kk);
    bool ok;
    TileRec out;
    out.start = n ? cv_line_start(cv, nlines, T, kk, 0, ok) : 0;
    uint32_t c[3];
#pragma unroll
    for (uint32_t k = 0; k < 3; ++k) {
        const uint64_t t = (uint64_t)T + k;
        c[k] = t < cv.nb ? (uint32_t)((t + 1 < cv.nb ? (uint64_t)cv.offs[t + 1] : nlines) - cv.offs[t]) : 0u;
        if (c[k] > CV_LIST_CAP) c[k] = CV_LIST_CAP;          // (an overflowed census: the kernels stand down anyway)
    }
    out.T = T; out.A = kk | (c[0] << 16); out.B = c[1] | (c[2] << 16); out.pad = 0;
    rec[tt] = out;
}

// LDS carve (dynamic): [16 B guard][stage][out_d | out_q][meta u32 x (4R+4)][luts 3 x 512 B]
// Workgroups are persistent: each walks tiles b, b + S, b + 2S, ... with a software pipeline -- the
// next tile's bytes and line offsets are loaded into registers (in flight) while the current tile is
// packed out of LDS; span bounds are requested two tiles ahead.
// STATS: the same pass also produces the pass-1 statistics (uq_stats) of the shard -- the speculative one-pass
// encode: the caller packs with GUESSED decisions while counting, then checks the guess against the counts
// (uq_pack_stats).  `st->reserved` is raised when the counts are incomplete (a record longer than the guess
// allowed for, or malformed): the caller then runs the plain statistics pass.
// QN (uq_pack_stats_qname): the last wave of the workgroup takes the tile's QNAME lines (qname_tile) while the other three share the
// packing -- two kinds of work side by side between the tile's two barriers.  (One wave doing both while three wait at the barrier
// was the long pole of every tile: 2.10 ms against 1.38 ms without the QNAME phase; a fifth wave for the QNAME lines leaves the CU
// with workgroups of five waves, of which it places three at a time where four of four waves fit: 2.0 - 2.5 ms.)  Built for four
// workgroups per CU: the parser's registers come on top of a tile in flight.
// VAR (variable-length tables, the queued fused forms): the lanes of a read visit only the groups that hold a symbol or the sentinel; the
// rows' empty upper parts (rows are sized for the longest read) are never stored -- the copy-out of the tile before has cleared the LDS image.
template <int BD, int BQ, bool NTRICK, bool FAST, bool STATS, bool QN, bool LISTS = false, bool VAR = false>
__global__ __launch_bounds__(PK_THREADS, (STATS && (NTRICK || QN)) ? 4 : 5) void pack_tile_kernel(const uint8_t* __restrict__ buf,
                                                               const uint64_t* __restrict__ ls, uint64_t first,
                                                               uint64_t n, PackLut lut, PackGeom g,
                                                               uint8_t* __restrict__ dna, uint8_t* __restrict__ qual,
                                                               unsigned long long* __restrict__ bad,
                                                               uq_stats* __restrict__ st, const unsigned long long* __restrict__ d_async,
                                                               uq_qname_fused* __restrict__ qf, uint32_t* __restrict__ qvals, uint64_t qpitch, PackLists pl,
                                                               const TileRec* __restrict__ trec) {      // trec: [pack tiles + 1] with pl (a restrict parameter of its own: the records are read through the scalar cache, like `ls`)
    constexpr int NV = STATS ? PK_NV_STATS : PK_NV;
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* stage = smem + 16;                       // reads of up to 8 bytes below offset 0 stay in bounds
    uint8_t* out_d = stage + g.stage_bytes;
    uint32_t* meta = (uint32_t*)(out_d + g.out_bytes);
    int16_t* l_dna = (int16_t*)(meta + 4 * g.R + 4);
    int16_t* l_qual = l_dna + 256;
    int16_t* l_nq = l_qual + 256;

    const uint32_t tid = threadIdx.x;
    l_dna[tid] = lut.dna_code[tid];
    l_qual[tid] = lut.qual_code[tid];
    if (NTRICK) l_nq[tid] = lut.n_qual[tid];
    RecordAcc acc;
    bool incomplete = false;
    constexpr bool FOUR_WG = STATS && (NTRICK || QN);
    constexpr int PKS_COPIES = pks_copies(BD, FOUR_WG);
    __shared__ uint32_t cnt_tab[STATS ? pks_words(BD, FOUR_WG) : 1];      // [bins][PKS_COPIES], then [128] N-trick base by quality character
    uint32_t fill_pairs = 0, n_pairs = 0;                        // pairs counted in bin 0 / bin n_code that were fills / N-trick positions
    constexpr uint32_t PKS_BINS = pks_bins(BD), PKS_WORDS = pks_words(BD, FOUR_WG);
    if (STATS) for (uint32_t i = tid; i < PKS_WORDS; i += PK_THREADS) cnt_tab[i] = 0;       // the first tile's barrier orders it
    if (VAR) for (uint32_t i = tid; i < (g.out_bytes >> 4); i += PK_THREADS) ((uint4*)out_d)[i] = make_uint4(0, 0, 0, 0);       // (likewise; every copy-out clears what it reads)
    QnLds* qn = (QnLds*)(l_nq + 256);                            // the QNAME phase's state (uq_pack_stats_qname only)
    bool qn_on = false;
    if (STATS && QN) qn_on = qf->ok != 0;                        // the guess kernels in front may have declined: the lines are left alone
    if (STATS && QN) {
        qn->line1[tid] = qf->line1[tid]; qn->inset[tid] = qf->inset[tid];
        if (tid < 16) qn->line1[256 + tid] = 0;
        if (tid < 32) qn->seps[tid] = qf->seps[tid];
        if (tid < UQ_QF_MAXC) { qn->vmin[tid] = 0xFFFFFFFFu; qn->vmax[tid] = 0; }
        if (tid == 0) {
            qn->flags = 0; qn->plen = qf->plen; qn->slen = qf->slen; qn->nsep = qf->nsep; qn->l1len = qf->l1len;
            uint32_t set = 0x01010101u * qf->seps[0], nset = 1, lo = 0, hi = 0;       // (the guess kernel allows at most four distinct separators)
            for (uint32_t k = 0; k < qf->nsep && k < 8; ++k) {
                const uint32_t c = qf->seps[k];
                if (k < 4) lo |= c << (8 * k); else hi |= c << (8 * (k - 4));
                bool seen = false;
                for (uint32_t m = 0; m < nset; ++m) seen = seen || ((set >> (8 * m)) & 0xFFu) == c;
                if (!seen && nset < 4) { set = (set & ~(0xFFu << (8 * nset))) | (c << (8 * nset)); ++nset; }
            }
            qn->sepset = set; qn->seps_lo = lo; qn->seps_hi = hi;
            uint32_t pw = 1;
            for (uint32_t k = 0; k < 10; ++k) { qn->pow10[k] = pw; pw *= 10; }
        }
    }
    uint64_t nl_total = 4 * (first + n);              // (list form: closes the last census tile's list)
    constexpr bool lists = STATS && LISTS;            // (instances of their own: a run-time choice between `ls` and `trec` turns the tile bounds' scalar loads into vector
                                                      // loads that wait for the tile's stores -- and spills the kernels built for five workgroups per CU)
    if (d_async) {                                    // the queued form (uq_pack_stats_async): the census in front left the line count on the device;
        nl_total = d_async[0];
        const uint64_t have = d_async[0] / 4;         // `n` is what the tables hold
        if (have > n) incomplete = true; else n = have;
        if (d_async[1] || (lists && *pl.over)) { incomplete = true; n = 0; }   // the queued index in front gave up (list or capacity overflow): its entries are not line starts
    }

    const uint64_t R = g.R;
    const uint64_t ntiles = (n + R - 1) / R;
    const uint64_t S = gridDim.x;
    struct Bounds { uint64_t g0, g1; uint32_t pT, pA, pB; };       // pT, pA, pB: PackLists::place
    struct Regs { uint4 v[NV]; uint64_t m0; uint64_t g0; uint32_t skew, nvec, Rt, mrel, mraw; bool ok; };      // (lists: line start - g0 + skew = mrel + mraw)
    // The bounds of a tile two steps ahead are requested by VECTOR loads, a dword (or an index entry) a lane, and put together with v_readlane when the tile's
    // bytes are requested a step later.  (As scalar loads -- which the compiler makes of a uniform address -- they were followed by `s_waitcnt lgkmcnt(0)` on the
    // spot: the kernel is short of SGPRs, the loaded values went straight to VGPR lanes, and every tile began with a round trip to the L2.  A vector load waits
    // behind phase C's `vmcnt(0)`, a whole tile later.)
    struct RawBounds { uint32_t lo, hi; };
    auto load_raw = [&](uint64_t tt) {
        RawBounds x{0u, 0u};
        if (tt < ntiles) {
            const uint32_t ln = lane_id();
            if constexpr (lists) {                // lanes 0 .. 7: the six dwords of the tile's record and the two of the next record's start
                x.lo = ((const uint32_t*)(trec + tt))[ln < 8u ? ln : 0u];
            } else {
                const uint32_t Rn = (uint32_t)((n - tt * R) < R ? (n - tt * R) : R);
                const uint64_t v = ls[4 * (first + tt * R) + (ln == 1u ? 4 * Rn : 0u)];           // lane 0: the tile's start, lane 1: its end
                x.lo = (uint32_t)v; x.hi = (uint32_t)(v >> 32);
            }
        }
        return x;
    };
    auto bounds_of = [&](uint64_t tt, const RawBounds& x) {
        auto lane = [](uint32_t v, int k) { return (uint32_t)__builtin_amdgcn_readlane((int)v, k); };       // (the builtin returns int: without the cast a set bit 31 -- an offset beyond 2 GiB -- would be sign-extended into the upper half)
        Bounds b{0, 0, 0, 0, 0};
        if (tt < ntiles) {
            if constexpr (lists) {
                b.g0 = (uint64_t)lane(x.lo, 0) | ((uint64_t)lane(x.lo, 1) << 32);
                b.pT = lane(x.lo, 2); b.pA = lane(x.lo, 3); b.pB = lane(x.lo, 4);
                b.g1 = (uint64_t)lane(x.lo, 6) | ((uint64_t)lane(x.lo, 7) << 32);
            } else {
                b.g0 = (uint64_t)lane(x.lo, 0) | ((uint64_t)lane(x.hi, 0) << 32);
                b.g1 = (uint64_t)lane(x.lo, 1) | ((uint64_t)lane(x.hi, 1) << 32);
            }
        }
        return b;
    };
    auto load_bounds = [&](uint64_t tt) {
        Bounds b{0, 0, 0, 0, 0};
        if (tt < ntiles) {
            const uint32_t Rn = (uint32_t)((n - tt * R) < R ? (n - tt * R) : R);
            if constexpr (lists) {                // (one round trip: the two records side by side)
                const TileRec r0 = trec[tt];
                b.g0 = r0.start; b.pT = r0.T; b.pA = r0.A; b.pB = r0.B; b.g1 = trec[tt + 1].start;
            }
            else { b.g0 = ls[4 * (first + tt * R)]; b.g1 = ls[4 * (first + tt * R) + 4 * Rn]; }
        }
        return b;
    };
    auto tile_reads = [&](uint64_t tt) -> uint32_t { return tt < ntiles ? (uint32_t)((n - tt * R) < R ? (n - tt * R) : R) : 0u; };
    // request the bytes and line offsets of reads [rfirst, rfirst + cnt) (their span is b); not ok = the span does not fit the stage
    auto issue = [&](uint64_t rfirst, uint32_t cnt, Bounds b) {
        Regs x;
        x.ok = false; x.m0 = 0; x.g0 = b.g0; x.skew = 0; x.nvec = 0; x.Rt = 0; x.mrel = 0; x.mraw = 0;
#pragma unroll
        for (int u = 0; u < NV; ++u) x.v[u] = make_uint4(0, 0, 0, 0);
        if (cnt == 0) return x;
        x.Rt = cnt;
        const uint64_t a0 = ((uint64_t)(uintptr_t)buf + b.g0) & ~uint64_t(15);   // absolute, 16-aligned
        x.skew = (uint32_t)(((uint64_t)(uintptr_t)buf + b.g0) - a0);
        const uint64_t span = b.g1 - b.g0 + x.skew;
        x.nvec = (uint32_t)((span + 15) >> 4);
        x.ok = span + 32 <= g.stage_bytes && x.nvec <= (uint32_t)(NV * PK_THREADS);
        if constexpr (lists) {                   // the tile's 4 Rt + 1 newlines within the three census tiles the place describes?
            const uint32_t kk = b.pA & 0xFFFFu, avail = (b.pA >> 16) + (b.pB & 0xFFFFu) + (b.pB >> 16);
            if (kk + 4 * cnt > avail) x.ok = false;         // (lines of many KiB: the tile goes piece by piece, each with its own search)
        }
        if (!x.ok) return x;                          // tiles are sized from the AVERAGE record: a long-winded one is split below
        const uint4* src = (const uint4*)(buf + ((int64_t)b.g0 - (int64_t)x.skew));      // (pointer arithmetic on the kernel's own argument, not an integer cast back to a pointer: the loads are global_load, not flat_load -- a flat access counts in lgkmcnt as well and returns out of order, so every wait for LDS data also waited for the tile in flight)
#pragma unroll
        for (int u = 0; u < NV; ++u) { const uint32_t i = u * PK_THREADS + tid; if (i < x.nvec) x.v[u] = src[i]; }
        if (tid <= 4 * x.Rt) {
            if constexpr (lists) {
                // the newline in front of line start `tid` of the tile: slot kk - 1 + tid of census tile T, counted on through T + 1, T + 2
                uint32_t idx = (b.pA & 0xFFFFu) + tid, t = b.pT;          // (slot + 1)
                const uint32_t c0 = b.pA >> 16, c1 = b.pB & 0xFFFFu;
                if (idx > c0) { idx -= c0; ++t; if (idx > c1) { idx -= c1; ++t; } }
                // the list entry stays as loaded until the next tile's phase A adds it up: arithmetic on it HERE would wait for the load -- and
                // with it for the tile's bytes requested above
                x.mrel = x.skew - (uint32_t)b.g0 + (idx == 0 ? 0u : (uint32_t)t * (uint32_t)CV_TILE - pl.cv.mis + 1u);      // (modulo 2^32: a tile spans less)
                x.mraw = idx == 0 ? 0u : (uint32_t)pl.cv.list[(uint64_t)t * CV_LIST_CAP + (idx - 1)];
            } else x.m0 = ls[4 * (first + rfirst) + tid];
        }
        return x;
    };

    uint32_t rr, pp;                                  // this lane packs groups pp, pp + P, ... of read rr of every tile
    fast_divmod(tid, g.P, g.magicP, rr, pp);
    uint32_t badr = 0xFFFFFFFFu;
    uint64_t bad_tile = UQ_NONE;
    // one tile: registers -> LDS (A), `between()` (the caller's prefetch of the next tile), pack (B), store (C)
    auto do_tile = [&](const uint64_t r0, const Regs& cur, auto&& between) {
        const uint32_t Rt = cur.Rt;
        uint8_t* out_q = out_d + ((Rt * g.Cd + 15) & ~15u);
        // ---- A: registers -> LDS
        if (cur.ok) {
            if (tid <= 4 * Rt) meta[tid] = lists ? cur.mrel + cur.mraw : (uint32_t)(cur.m0 - cur.g0) + cur.skew;
#pragma unroll
            for (int u = 0; u < NV; ++u) { const uint32_t i = u * PK_THREADS + tid; if (i < cur.nvec) ((uint4*)stage)[i] = cur.v[u]; }
        } else {
            if (tid == 0) bad_tile = bad_tile < r0 ? bad_tile : r0;
            incomplete = true;
            if (STATS && QN && qn_on && tid == 0) atomicOr(&qf->flags, 256u);
        }
        __syncthreads();
        const bool ok = cur.ok;
        between();
        // the QNAME lines of the tile: the last wave, a lane per read (its lanes have no part in phase B: P is sized for three waves)
#ifndef PK_ABL_NOQN
        if (STATS && QN && qn_on && ok && tid >= PK_THREADS - 64 && tid - (PK_THREADS - 64) < Rt)
            qname_tile(stage, meta, tid - (PK_THREADS - 64), qn, qvals, qpitch, r0 + (tid - (PK_THREADS - 64)));
#endif
        if (ok) {
            // ---- B: P lanes per read; a lane owns groups of 8 consecutive symbols, both streams:
            //         characters -> codes -> bits (8 symbols of b bits = b whole bytes)
            // one group of eight symbols of read r (length L, SEQ / QUAL lines at so / qo of the stage, the rows' last bytes at orow_d / orow_q):
            // characters -> codes -> bits, counted (STATS), stored into the LDS image of the two tables
            // (wd / wq: the dword-aligned LDS offsets of the group's two windows, shd / shq: the windows' byte phases -- the same for every group of a
            // read, since the windows move by eight bytes: one subtraction per window and group instead of add, two ANDs, add)
            auto pack_group = [&](const uint32_t r, const uint32_t L, const int32_t wd, const uint32_t shd, const int32_t wq, const uint32_t shq, uint8_t* orow_d,
                                  uint8_t* orow_q, const uint32_t gg) {
                // symbols t = 8gg .. 8gg+7 (t counts from the END of the read) = characters j = L-1-t, i.e. the
                // 8 bytes ending at position L - 8gg; byte k of the window is character j = j0 + k
                const int32_t j0 = (int32_t)L - 8 * (int32_t)gg - 8;
                uint32_t ad0 = 0, ad1 = 0, aq0 = 0, aq1 = 0;      // 4 symbols each; *0 = the more significant half
                if (j0 > -8) {      // else: the whole group lies above the read (rows are sized for the longest one): zeros
                uint32_t b_lo, b_hi, q_lo, q_hi;
                lds_window8_at(stage, wd, shd, b_lo, b_hi);
                lds_window8_at(stage, wq, shq, q_lo, q_hi);
                if (j0 < 0) {   // window reaches above the first base: those symbols are zero (fill with code-0 characters)
                    uint32_t mlo, mhi;
                    window_masks((uint32_t)(-j0), mlo, mhi);
                    b_lo = bfi(mlo, b_lo, g.fill_d); b_hi = bfi(mhi, b_hi, g.fill_d);
                    q_lo = bfi(mlo, q_lo, g.fill_q); q_hi = bfi(mhi, q_hi, g.fill_q);
                }
                bool generic = !FAST;
                if (FAST) {
                    uint32_t c0, c1, e0, e1;                     // codes, and (non-zero byte) = not a base of the alphabet
                    if constexpr (BD == 2) {
                        c0 = acgt_codes(b_lo); c1 = acgt_codes(b_hi);
                        e0 = acgt_chars(c0) ^ b_lo; e1 = acgt_chars(c1) ^ b_hi;
                    } else {                                     // up to eight bases told apart by three bits of their characters
                        c0 = __builtin_amdgcn_perm(g.i2c_hi, g.i2c_lo, (b_lo >> g.h_shift) & 0x07070707u);
                        c1 = __builtin_amdgcn_perm(g.i2c_hi, g.i2c_lo, (b_hi >> g.h_shift) & 0x07070707u);
                        e0 = __builtin_amdgcn_perm(g.c2c_hi, g.c2c_lo, c0) ^ b_lo; e1 = __builtin_amdgcn_perm(g.c2c_hi, g.c2c_lo, c1) ^ b_hi;
                    }
                    uint32_t u0 = q_lo + g.q_addlo, u1 = q_hi + g.q_addlo;
                    uint32_t bq0 = (q_lo | (q_lo + g.q_addhi) | ~u0) & 0x80808080u;
                    uint32_t bq1 = (q_hi | (q_hi + g.q_addhi) | ~u1) & 0x80808080u;
                    uint32_t x0 = u0 & 0x7F7F7F7Fu, x1 = u1 & 0x7F7F7F7Fu;
                    if (e0 | e1) {
                        if (NTRICK && !((q_lo | q_hi) & 0x80808080u)) {
                            const uint32_t m0 = nonzero_bytes(e0), m1 = nonzero_bytes(e1);
                            if (((b_lo ^ g.n_char) & m0) | ((b_hi ^ g.n_char) & m1)) generic = true;   // not the N-trick base
                            c0 &= ~m0; c1 &= ~m1;
                            x0 = bfi(m0, g.n_code, x0); x1 = bfi(m1, g.n_code, x1);
                            bq0 &= ~m0; bq1 &= ~m1;
                        } else generic = true;
                    }
                    if (bq0 | bq1) generic = true;
                    ad0 = pack4<BD>(c0); ad1 = pack4<BD>(c1);
                    aq0 = pack4<BQ>(x0); aq1 = pack4<BQ>(x1);
                    if (STATS) {
                        if (generic) incomplete = true;          // a symbol outside the guess: not counted here
#ifdef PK_ABL_NOCOUNT
                        else if (false) {
#else
                        else {
#endif
                            uint8_t* hb = (uint8_t*)cnt_tab + ((lane_id() & (PKS_COPIES - 1)) << 2);
                            if constexpr (BD == 2) {
                                const uint32_t bin0 = (c0 << 6) | x0, bin1 = (c1 << 6) | x1;    // BQ <= 6 and two bits of base: a bin per byte
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    if constexpr (PKS_COPIES == 1) {
                                        atomicAdd(&cnt_tab[(bin0 >> (8 * k)) & 0xFFu], 1u);
                                        atomicAdd(&cnt_tab[(bin1 >> (8 * k)) & 0xFFu], 1u);
                                    } else {
                                        atomicAdd((uint32_t*)(hb + (((bin0 >> (8 * k)) & 0xFFu) * (4 * PKS_COPIES))), 1u);
                                        atomicAdd((uint32_t*)(hb + (((bin1 >> (8 * k)) & 0xFFu) * (4 * PKS_COPIES))), 1u);
                                    }
                                }
                            } else {                                                            // three bits of base: nine bits of bin, pair by pair
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const uint32_t b0 = (((c0 >> (8 * k)) & 7u) << 6) | ((x0 >> (8 * k)) & 63u), b1 = (((c1 >> (8 * k)) & 7u) << 6) | ((x1 >> (8 * k)) & 63u);
                                    atomicAdd((uint32_t*)(hb + b0 * (4 * PKS_COPIES)), 1u);
                                    atomicAdd((uint32_t*)(hb + b1 * (4 * PKS_COPIES)), 1u);
                                }
                            }
                            if (j0 < 0) fill_pairs += (uint32_t)(-j0);
                            if (NTRICK && (e0 | e1)) {
                                // the N-trick base's own pairs: the guess says it occurs with ONE quality character (that is what
                                // makes it an N-trick base); positions that bear it are counted in a register, byte-parallel, and a
                                // position with any other quality makes the guess wrong: incomplete, the exact pass decides
                                const uint32_t m0 = nonzero_bytes(e0), m1 = nonzero_bytes(e1);
                                const uint32_t ne0 = nonzero_bytes(q_lo ^ g.n_qchar), ne1 = nonzero_bytes(q_hi ^ g.n_qchar);
                                n_pairs += (uint32_t)__popc(m0 & 0x01010101u) + (uint32_t)__popc(m1 & 0x01010101u);
                                if ((m0 & ne0) | (m1 & ne1)) incomplete = true;
                            }
                        }
                    }
                }
                if (generic) {
                    const uint32_t cbw[2] = {b_lo, b_hi};
                    const uint32_t cqw[2] = {q_lo, q_hi};
                    uint32_t ad[2] = {0, 0}, aq[2] = {0, 0};
                    int32_t orall = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const uint32_t cb = (cbw[k >> 2] >> (8 * (k & 3))) & 255u;
                        const uint32_t cc = (cqw[k >> 2] >> (8 * (k & 3))) & 255u;
                        int32_t dc = l_dna[cb];
                        int32_t qc = l_qual[cc];
                        if (NTRICK) {
                            if (dc < 0) { dc = 0; qc = l_nq[cb]; }
                        }
                        orall |= dc | qc;
                        ad[k >> 2] = (ad[k >> 2] << BD) | (uint32_t)dc;
                        aq[k >> 2] = (aq[k >> 2] << BQ) | (uint32_t)qc;
                    }
                    if (orall < 0) {            // a character without a code: report, keep the row deterministic
                        badr = badr < (uint32_t)(r0 + r) ? badr : (uint32_t)(r0 + r);
                        ad[0] = ad[1] = aq[0] = aq[1] = 0;
                    }
                    ad0 = ad[0]; ad1 = ad[1]; aq0 = aq[0]; aq1 = aq[1];
                }
                }
                if (g.variable) {                                  // sentinel = code 1 at symbol index t = L
                    const int32_t i = (int32_t)L - 8 * (int32_t)gg;
                    if (i >= 0 && i < 4) { ad1 |= 1u << (BD * i); aq1 |= 1u << (BQ * i); }
                    else if (i >= 4 && i < 8) { ad0 |= 1u << (BD * (i - 4)); aq0 |= 1u << (BQ * (i - 4)); }
                }
                // byte index counts from the row's LAST byte; only the top group can stick out of the row
                uint8_t* od = orow_d - BD * gg;
                uint8_t* oq = orow_q - BQ * gg;
#ifdef PK_ABL_NOSTORE
                if ((ad0 ^ aq0) == 0x12345678u && (ad1 ^ aq1) == 0x9ABCDEF0u) od[0] = 1;      // (keeps the values alive)
                else if (false)
#endif
                if (gg + 1 < g.G) {
                    // byte stores: unaligned ds_write_b32 / b16 pieces are accepted by gfx950 but slower (1.11 -> 1.27 ms)
#pragma unroll
                    for (int i = 0; i < BD; ++i) od[-i] = (uint8_t)group_byte<BD>(ad0, ad1, i);
#pragma unroll
                    for (int i = 0; i < BQ; ++i) oq[-i] = (uint8_t)group_byte<BQ>(aq0, aq1, i);
                } else {
                    const uint32_t nd = g.Cd - BD * gg, nq = g.Cq - BQ * gg;     // bytes left in the row from here up
#pragma unroll
                    for (int i = 0; i < BD; ++i)
                        if ((uint32_t)i < nd) od[-i] = (uint8_t)group_byte<BD>(ad0, ad1, i);
#pragma unroll
                    for (int i = 0; i < BQ; ++i)
                        if ((uint32_t)i < nq) oq[-i] = (uint8_t)group_byte<BQ>(aq0, aq1, i);
                }
            };
            if (rr < Rt) {
                const uint32_t r = rr;
                const uint32_t so = meta[4 * r + 1];
                uint32_t L = meta[4 * r + 2] - so - 1;
                const uint32_t qo = meta[4 * r + 3];
                if (STATS) {
                    const uint32_t Lq = meta[4 * r + 4] - qo - 1;
#ifndef PK_ABL_NOREC
                    if (pp == 0) acc.record(r0 + r, stage[meta[4 * r + 2]] == '+', L, Lq, meta[4 * r + 4] - meta[4 * r]);
#endif
                    if (L > g.dna_max || Lq != L) incomplete = true;        // symbols this kernel does not visit
                }
                if (L > g.dna_max || meta[4 * r + 4] - qo - 1 != L) { badr = badr < (uint32_t)(r0 + r) ? badr : (uint32_t)(r0 + r); L = 0; }
                uint8_t* orow_d = out_d + r * g.Cd + (g.Cd - 1);
                uint8_t* orow_q = out_q + r * g.Cq + (g.Cq - 1);
                // VAR: only the groups that hold a symbol or the sentinel -- the rows' empty upper parts are zero already: every copy-out clears what
                // it reads (rows are sized for the LONGEST read: 38 group slots for 21.5 groups on average at 36 - 301 bp)
                const uint32_t gend = VAR ? (((L + 8) >> 3) < g.G ? ((L + 8) >> 3) : g.G) : g.G;
#ifndef PK_ABL_NOGROUP
                int32_t wd = (int32_t)(so + L) - 8 * (int32_t)pp - 8, wq = (int32_t)(qo + L) - 8 * (int32_t)pp - 8;      // so + j0, qo + j0 of group pp
                const uint32_t shd = (uint32_t)wd & 3u, shq = (uint32_t)wq & 3u;
                wd &= ~3; wq &= ~3;
                const int32_t wstep = 8 * (int32_t)g.P;
                for (uint32_t gg = pp; gg < gend; gg += g.P, wd -= wstep, wq -= wstep) pack_group(r, L, wd, shd, wq, shq, orow_d, orow_q, gg);
#endif
            }
        }
        __syncthreads();
#ifndef PK_NO_EARLY_WAIT
        // The next tile's bytes were requested a whole phase B ago: wait for them HERE, before this tile's stores are issued.  vmcnt counts loads and
        // stores together, in issue order: the wait the compiler would put in front of the next tile's phase A also waits for the stores below to be
        // acknowledged (1 - 1.5 us under load, every tile); behind this one it has nothing left to wait for.
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt left alone
#endif
        if (ok) {
            // ---- C: coalesced stores of the two packed tiles (widest vector the tile's byte offset allows)
            if constexpr (STATS && (NTRICK || QN)) {            // (the forms built for four workgroups per CU have the registers for the inline loop)
                store_tile<VAR>(dna + r0 * g.Cd, out_d, Rt * g.Cd, tid);
                store_tile<VAR>(qual + r0 * g.Cq, out_q, Rt * g.Cq, tid);
            } else {
                store_tile_any<VAR>(dna + r0 * g.Cd, out_d, Rt * g.Cd, tid);
                store_tile_any<VAR>(qual + r0 * g.Cq, out_q, Rt * g.Cq, tid);
            }
        }
    };

    uint64_t t = blockIdx.x;
    // (only the forms built for four workgroups per CU take the vector loads: at the 96 registers of the others they spill -- 27 dwords in the indexed
    // pack + statistics kernel, 1.40 -> 1.50 ms -- and those keep the scalar loads, requested at the top of the tile before)
    constexpr bool VBOUNDS = STATS && (NTRICK || QN);
    using NextBounds = std::conditional_t<VBOUNDS, RawBounds, Bounds>;
    auto fetch_next = [&](uint64_t tt) -> NextBounds { if constexpr (VBOUNDS) return load_raw(tt); else return load_bounds(tt); };
    auto next_bounds = [&](uint64_t tt, const NextBounds& x) -> Bounds { if constexpr (VBOUNDS) return bounds_of(tt, x); else return x; };
    NextBounds b_next = fetch_next(t + S);
    Regs cur = issue(t * R, tile_reads(t), load_bounds(t));
#ifndef PK_NO_EARLY_WAIT
    __builtin_amdgcn_s_waitcnt(0x0F70);          // (every path into a tile's phase A has waited for the tile's bytes: see phase C)
#endif
    for (; t < ntiles; t += S) {
        NextBounds b_nn{};
        if constexpr (!VBOUNDS) b_nn = fetch_next(t + 2 * S);
        if (cur.ok) {
            const Regs now = cur;
            do_tile(t * R, now, [&] {
                cur = issue((t + S) * R, tile_reads(t + S), next_bounds(t + S, b_next));
                if constexpr (VBOUNDS) b_next = fetch_next(t + 2 * S); else b_next = b_nn;
            });
        } else {
            // the tile's records add up to more than the stage holds (R comes from the average record length): pack it in
            // pieces of g.Rs reads, which always fit, one after the other; then pick the pipeline up again
            const uint32_t Rt = cur.Rt;
            for (uint32_t sub = 0; sub < Rt; sub += g.Rs) {
                const uint32_t cnt = Rt - sub < g.Rs ? Rt - sub : g.Rs;
                const uint64_t rf = t * R + sub;
                Bounds sb{0, 0, 0, 0, 0};
                if constexpr (lists) {                  // (rare: a search per piece, and the counts of its three census tiles)
                    uint32_t T, kk; bool found;
                    cv_locate(pl.cv, (int64_t)(4 * (first + rf)) - 1, T, kk);
                    sb.g0 = cv_line_start(pl.cv, nl_total, T, kk, 0, found); sb.g1 = cv_line_start(pl.cv, nl_total, T, kk, 4 * cnt, found);
                    uint32_t c[3];
#pragma unroll
                    for (uint32_t k = 0; k < 3; ++k) {
                        const uint64_t tq = (uint64_t)T + k;
                        c[k] = tq < pl.cv.nb ? (uint32_t)((tq + 1 < pl.cv.nb ? (uint64_t)pl.cv.offs[tq + 1] : nl_total) - pl.cv.offs[tq]) : 0u;
                        if (c[k] > CV_LIST_CAP) c[k] = CV_LIST_CAP;
                    }
                    sb.pT = T; sb.pA = kk | (c[0] << 16); sb.pB = c[1] | (c[2] << 16);
                } else { sb.g0 = ls[4 * (first + rf)]; sb.g1 = ls[4 * (first + rf) + 4 * cnt]; }
                const Regs piece = issue(rf, cnt, sb);
                do_tile(rf, piece, [] {});
                __syncthreads();                      // the next piece overwrites the stage and the out tile
            }
            cur = issue((t + S) * R, tile_reads(t + S), next_bounds(t + S, b_next));
            if constexpr (VBOUNDS) b_next = fetch_next(t + 2 * S); else b_next = b_nn;
#ifndef PK_NO_EARLY_WAIT
            __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
        }
    }
    if (badr != 0xFFFFFFFFu) atomicMin(bad, (unsigned long long)badr);
    if (bad_tile != UQ_NONE) atomicMin(bad, (unsigned long long)bad_tile);
    if (STATS) {
        // fills were counted as (code 0, code 0), N-trick positions as (code 0, n_code): take them off their bins (any copy: the
        // flush sums the copies modulo 2^32)
        if (fill_pairs) atomicSub(&cnt_tab[lane_id() & (PKS_COPIES - 1)], fill_pairs);
        if (NTRICK && n_pairs) {
            atomicSub(&cnt_tab[((g.n_code & 0xFFu) * PKS_COPIES) + (lane_id() & (PKS_COPIES - 1))], n_pairs);
            atomicAdd(&cnt_tab[PKS_BINS * PKS_COPIES], n_pairs);       // ... and onto the N-trick base's own counter
        }
        __syncthreads();
        const uint32_t qmin = 0x80u - (g.q_addlo & 0xFFu);
        for (uint32_t bin = tid; bin < PKS_BINS; bin += PK_THREADS) {
            uint32_t v = 0;
#pragma unroll
            for (int c = 0; c < PKS_COPIES; ++c) v += cnt_tab[bin * PKS_COPIES + c];
            const uint32_t code = bin >> 6;
            const uint32_t base = BD == 2 ? (0x54474341u >> (8 * code)) & 0xFFu                           // "ACGT"[code]
                                          : ((code < 4 ? g.c2c_lo >> (8 * code) : g.c2c_hi >> (8 * (code - 4))) & 0xFFu);
            if (v) atomicAdd((unsigned long long*)&st->counts[base * 256 + qmin + (bin & 63u)], (unsigned long long)v);
        }
        if (NTRICK && tid == 0) {
            const uint32_t v = cnt_tab[PKS_BINS * PKS_COPIES];
            if (v) atomicAdd((unsigned long long*)&st->counts[(g.n_char & 0xFFu) * 256 + (g.n_qchar & 0xFFu)], (unsigned long long)v);
        }
        acc.flush(st, first);
        if (incomplete) st->reserved = 1;
        if (QN) {                                            // (the barrier above also closes the last tile's QNAME phase)
            if (qn_on && tid < UQ_QF_MAXC && qn->vmin[tid] <= qn->vmax[tid]) { atomicMin(&qf->vmin[tid], qn->vmin[tid]); atomicMax(&qf->vmax[tid], qn->vmax[tid]); }
            if (qn_on && tid == 0 && qn->flags) atomicOr(&qf->flags, qn->flags);
            if (blockIdx.x == 0 && tid == 0) qf->nreads = n;
        }
    }
}

// Exact big-integer form: thread per read, byte-serial addition with carry, straight from HBM.
__global__ __launch_bounds__(256) void pack_carry_kernel(const uint8_t* __restrict__ buf, const uint64_t* __restrict__ ls,
                                                         uint64_t first, uint64_t n, PackLut lut, uint32_t bd, uint32_t bq,
                                                         uint32_t Cd, uint32_t Cq, uint32_t variable,
                                                         uint8_t* __restrict__ dna, uint8_t* __restrict__ qual,
                                                         unsigned long long* __restrict__ bad) {
    __shared__ int16_t l_dna[256], l_qual[256], l_nq[256];
    l_dna[threadIdx.x] = lut.dna_code[threadIdx.x];
    l_qual[threadIdx.x] = lut.qual_code[threadIdx.x];
    l_nq[threadIdx.x] = lut.n_qual[threadIdx.x];
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint64_t* p = ls + 4 * (first + r);
    const uint64_t s = p[1], q = p[3];
    const uint32_t L = (uint32_t)(p[2] - s - 1);
    uint8_t* drow = dna + r * Cd;
    uint8_t* qrow = qual + r * Cq;
    // accumulate from the last base (least significant) upward, as uq.py:147-167 does
    uint32_t td = 0, tq = 0, bdn = 0, bqn = 0;
    int32_t pd = (int32_t)Cd - 1, pq = (int32_t)Cq - 1;
    for (uint32_t t = 0; t < L; ++t) {
        uint32_t cb = buf[s + L - 1 - t], cc = buf[q + L - 1 - t];
        int dc = l_dna[cb], qc = l_qual[cc];
        if (dc < 0) { dc = 0; qc = l_nq[cb]; }
        if (qc < 0) { atomicMin(bad, (unsigned long long)r); qc = 0; }
        td += (uint32_t)dc << bdn; tq += (uint32_t)qc << bqn;
        bdn += bd; bqn += bq;
        while (bdn > 8) { bdn -= 8; if (pd >= 0) drow[pd] = (uint8_t)td; td >>= 8; --pd; }
        while (bqn > 8) { bqn -= 8; if (pq >= 0) qrow[pq] = (uint8_t)tq; tq >>= 8; --pq; }
    }
    td += variable << bdn; tq += variable << bqn;
    while (pd >= 0) { drow[pd] = (uint8_t)td; td >>= 8; --pd; }
    while (pq >= 0) { qrow[pq] = (uint8_t)tq; tq >>= 8; --pq; }
}

typedef void (*PackKernel)(const uint8_t*, const uint64_t*, uint64_t, uint64_t, PackLut, PackGeom, uint8_t*, uint8_t*, unsigned long long*,
                           uq_stats*, const unsigned long long*, uq_qname_fused*, uint32_t*, uint64_t, PackLists, const TileRec*);

template <int BD, int BQ>
PackKernel pick_nt(bool ntrick, bool fast) {
    if ((BD == 2 || BD == 3) && fast) return ntrick ? pack_tile_kernel<(BD == 3 ? 3 : 2), BQ, true, true, false, false> : pack_tile_kernel<(BD == 3 ? 3 : 2), BQ, false, true, false, false>;
    return ntrick ? pack_tile_kernel<BD, BQ, true, false, false, false> : pack_tile_kernel<BD, BQ, false, false, false, false>;
}

// the fused pack + statistics kernels exist for the lookup-free path only (2-bit A/C/G/T, contiguous qualities)
template <int BD, bool QN, bool LISTS = false, bool VAR = false>
PackKernel pick_stats_kernel_q(int bq, bool ntrick) {
#define UQ_PS(B) case B: return ntrick ? pack_tile_kernel<BD, B, true, true, true, QN, LISTS, VAR> : pack_tile_kernel<BD, B, false, true, true, QN, LISTS, VAR>;
    switch (bq) { UQ_PS(1) UQ_PS(2) UQ_PS(3) UQ_PS(4) UQ_PS(5) default: return ntrick ? pack_tile_kernel<BD, 6, true, true, true, QN, LISTS, VAR> : pack_tile_kernel<BD, 6, false, true, true, QN, LISTS, VAR>; }
#undef UQ_PS
}
PackKernel pick_stats_kernel(int bd, int bq, bool ntrick, bool qn, bool lists, bool var) {
    if (var && lists) {                           // variable lengths in the queued forms: the dealing of phase B (VAR)
        if (qn) return bd == 3 ? pick_stats_kernel_q<3, true, true, true>(bq, ntrick) : pick_stats_kernel_q<2, true, true, true>(bq, ntrick);
        return bd == 3 ? pick_stats_kernel_q<3, false, true, true>(bq, ntrick) : pick_stats_kernel_q<2, false, true, true>(bq, ntrick);
    }
    if (qn && lists) return bd == 3 ? pick_stats_kernel_q<3, true, true>(bq, ntrick) : pick_stats_kernel_q<2, true, true>(bq, ntrick);
    if (lists) return bd == 3 ? pick_stats_kernel_q<3, false, true>(bq, ntrick) : pick_stats_kernel_q<2, false, true>(bq, ntrick);
    if (bd == 3) return qn ? pick_stats_kernel_q<3, true>(bq, ntrick) : pick_stats_kernel_q<3, false>(bq, ntrick);
    return qn ? pick_stats_kernel_q<2, true>(bq, ntrick) : pick_stats_kernel_q<2, false>(bq, ntrick);
}

template <int BD>
PackKernel pick_bq(int bq, bool ntrick, bool fast) {
    switch (bq) {
        case 1: return pick_nt<BD, 1>(ntrick, fast); case 2: return pick_nt<BD, 2>(ntrick, fast); case 3: return pick_nt<BD, 3>(ntrick, fast);
        case 4: return pick_nt<BD, 4>(ntrick, fast); case 5: return pick_nt<BD, 5>(ntrick, fast); case 6: return pick_nt<BD, 6>(ntrick, fast);
        case 7: return pick_nt<BD, 7>(ntrick, fast); default: return pick_nt<BD, 8>(ntrick, fast);
    }
}

PackKernel pick_kernel(int bd, int bq, bool ntrick, bool fast) {
    switch (bd) {
        case 1: return pick_bq<1>(bq, ntrick, fast); case 2: return pick_bq<2>(bq, ntrick, fast); case 3: return pick_bq<3>(bq, ntrick, fast);
        case 4: return pick_bq<4>(bq, ntrick, fast); case 5: return pick_bq<5>(bq, ntrick, fast); case 6: return pick_bq<6>(bq, ntrick, fast);
        case 7: return pick_bq<7>(bq, ntrick, fast); default: return pick_bq<8>(bq, ntrick, fast);
    }
}
}  // namespace

// d_stats == nullptr: plain pack.  Otherwise the fused pack + statistics kernel, if this geometry has one
// (*h_fused = 1); if not, nothing is launched and *h_fused = 0.
static int pack_impl(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t first_read,
                     uint64_t nreads, const uq_pack_params* hp, uint8_t* d_dna, uint8_t* d_qual, uint64_t* d_bad,
                     uq_stats* d_stats, int* h_fused, const unsigned long long* d_async = nullptr, uq_qname_fused* d_q = nullptr,
                     uint32_t* d_vals = nullptr, uint64_t vals_pitch = 0) {
    // d_line_start == nullptr (queued forms only): the line starts come from the lists of the census queued in front (lines.h)
    const bool use_lists = d_line_start == nullptr && d_async != nullptr && d_stats != nullptr;
    UQ_REQUIRE(ctx && d_buf && (d_line_start || use_lists) && hp && d_dna && d_qual && d_bad, "uq_pack: null argument");
    UQ_REQUIRE(!use_lists || (ctx->async_buf == d_buf && ctx->async_nbytes > 0), "uq_pack_stats_async: no line index given and no census of this buffer queued in front");
    if (h_fused) *h_fused = 0;
    UQ_REQUIRE(hp->bits_per_base >= 1 && hp->bits_per_base <= 8 && hp->bits_per_quality >= 1 && hp->bits_per_quality <= 8,
               "uq_pack: bits per symbol must be 1..8");
    const uint32_t bd = hp->bits_per_base, bq = hp->bits_per_quality;
    const uint32_t Cd = hp->dna_bytes_per_row, Cq = hp->quality_bytes_per_row;
    const uint32_t Lv = hp->dna_max + (hp->variable ? 1 : 0);
    UQ_REQUIRE(Cd == (bd * Lv + 7) / 8 && Cq == (bq * Lv + 7) / 8,
               "uq_pack: row bytes (%u, %u) do not match ceil(bits * (dna_max + variable) / 8)", Cd, Cq);
    UQ_CHECK_HIP(hipMemsetAsync(d_bad, 0xFF, 8, ctx->stream));
    if (nreads == 0) return 0;

    PackLut lut;
    int max_d = 0, max_q = 0;
    bool ntrick = false;
    int fill_d = -1, fill_q = -1;
    for (int i = 0; i < 256; ++i) {
        lut.dna_code[i] = hp->dna_code[i];
        lut.qual_code[i] = hp->qual_code[i];
        lut.n_qual[i] = (int16_t)(hp->n_qual[i] > 32767 ? 32767 : hp->n_qual[i]);
        if (hp->dna_code[i] > max_d) max_d = hp->dna_code[i];
        if (hp->qual_code[i] > max_q) max_q = hp->qual_code[i];
        if (hp->dna_code[i] < 0 && hp->n_qual[i] >= 0) { ntrick = true; if (hp->n_qual[i] > max_q) max_q = hp->n_qual[i]; }
        if (hp->dna_code[i] == 0 && fill_d < 0) fill_d = i;
        if (hp->qual_code[i] == 0 && fill_q < 0) fill_q = i;
    }
    UQ_REQUIRE(fill_d >= 0 && fill_q >= 0, "uq_pack: the alphabets need a symbol with code 0");
    UQ_REQUIRE(max_d < (1 << bd), "uq_pack: a DNA code does not fit %u bits", bd);
    const bool carry = max_q >= (1 << bq);   // Q9: N quality code == 2^b (or beyond)
    // A tile is at most PK_NV * 256 * 16 B = 20 KiB of FASTQ (what one workgroup keeps in flight in registers).  Records
    // that do not fit (reads beyond ~10 kbp: long-read platforms) take the exact thread-per-read kernel too: slow, never wrong.
    const uint32_t stage_cap = (d_stats ? PK_NV_STATS : PK_NV) * PK_THREADS * 16;
    const uint32_t rec = (uint32_t)hp->max_record_bytes;
    UQ_REQUIRE(rec >= 4, "uq_pack: max_record_bytes not set (take it from uq_stats)");

    if (carry || rec + 64 > stage_cap) {
        if (d_stats) return 0;                   // no fused form of the exact kernel
        uint32_t blocks = (uint32_t)((nreads + 255) / 256);
        pack_carry_kernel<<<blocks, 256, 0, ctx->stream>>>(d_buf, d_line_start, first_read, nreads, lut, bd, bq, Cd, Cq,
                                                           hp->variable ? 1u : 0u, d_dna, d_qual, (unsigned long long*)d_bad);
        UQ_LAUNCH_CHECK();
        return 0;
    }

    PackGeom g;
    g.Cd = Cd; g.Cq = Cq; g.variable = hp->variable ? 1 : 0;
    g.G = (Lv + 7) / 8;
    g.dna_max = (uint32_t)hp->dna_max;
    g.magicG = magic_u32(g.G);
    g.fill_d = 0x01010101u * (uint32_t)fill_d;
    g.fill_q = 0x01010101u * (uint32_t)fill_q;
    // Reads per tile.  Rs (from the LONGEST record) always fits the stage; R comes from the AVERAGE record (+15 %) when the
    // caller knows it -- variable-length files fill the stage instead of leaving room for 48 longest reads -- and a tile
    // whose records add up to more than the stage is packed in pieces of Rs reads.
    uint32_t Rs = (stage_cap - 64) / rec;
    if (Rs > (PK_THREADS - 1) / 4) Rs = (PK_THREADS - 1) / 4;   // 4R + 1 line offsets, one per lane
    if (Rs == 0) Rs = 1;
    uint32_t R = Rs;
    const uint32_t avg = (uint32_t)hp->avg_record_bytes;        // 0 = unknown
    if (avg >= 4 && avg < rec) {
        R = (stage_cap - 64) / (avg + avg * 15 / 100 + 1);
        if (R > (PK_THREADS - 1) / 4) R = (PK_THREADS - 1) / 4;
        if (R < Rs) R = Rs;
    }
    // ... but the rows of a tile are sized for the LONGEST read: a file of short reads with a few long ones among them must not ask for
    // R rows of the long kind in LDS (a request beyond the CU's 160 KiB ended the call with an error instead of a slower tile)
    {
        const uint32_t out_cap = stage_cap * 3 / 4, fit = out_cap / (Cd + Cq);
        if (R > fit) R = fit > Rs ? fit : Rs;
    }
    // R a multiple of 16 keeps every tile's output offset 16-byte aligned (uint4 stores); when that would waste
    // more than ~15 % of the tile, settle for a multiple of 8 or 4 (8- / 4-byte stores)
    if (R >= 4) {
        const uint32_t r16 = R & ~15u, r8 = R & ~7u, r4 = R & ~3u;
        if (r16 * 100 >= R * 85) R = r16; else if (r8 * 100 >= R * 85) R = r8; else R = r4;
    }
    if (R == 0) R = 1;
    if (Rs > R) Rs = R;
    g.R = R; g.Rs = Rs;
    g.stage_bytes = stage_cap + 32;
    g.out_bytes = (((R * Cd + 15) & ~15u) + R * Cq + 15) & ~15u;
    // phase B: P lanes per read
    uint32_t P = (d_q ? PK_THREADS - 64 : PK_THREADS) / R;        // (QN: the last wave parses QNAME lines instead)
    if (P > g.G) P = g.G;
    if (P < 1) P = 1;
    g.P = P; g.magicP = magic_u32(P);
    // fast path: bases == "ACGT" (2 bits) or up to eight bases that three bits of their characters tell apart (3 bits), qualities one
    // contiguous ASCII range below 128, at most one N-trick base
    bool fast = bd == 2 && hp->dna_code['A'] == 0 && hp->dna_code['C'] == 1 && hp->dna_code['G'] == 2 && hp->dna_code['T'] == 3;
    int nbases = 0, nq = 0, qmin = 256, qmaxc = -1, ntrick_bases = 0, nchar = 0;
    for (int i = 0; i < 256; ++i) {
        if (hp->dna_code[i] >= 0) ++nbases;
        if (hp->qual_code[i] >= 0) { ++nq; if (i < qmin) qmin = i; if (i > qmaxc) qmaxc = i; }
        if (hp->dna_code[i] < 0 && hp->n_qual[i] >= 0) { ++ntrick_bases; nchar = i; }
    }
    g.h_shift = g.i2c_lo = g.i2c_hi = g.c2c_lo = g.c2c_hi = 0;
    bool fast3 = false;
    if (bd == 3 && nbases >= 1 && nbases <= 8) {
        for (uint32_t sh = 0; sh <= 4 && !fast3; ++sh) {
            uint8_t i2c[8], c2c[8]; bool used[8] = {false, false, false, false, false, false, false, false};
            memset(i2c, 0, 8); memset(c2c, 0, 8);
            bool ok = true;
            for (int ch = 1; ch < 128 && ok; ++ch) {
                const int code = hp->dna_code[ch];
                if (code < 0) continue;
                const uint32_t idx = ((uint32_t)ch >> sh) & 7u;
                if (used[idx] || code > 7) ok = false;
                else { used[idx] = true; i2c[idx] = (uint8_t)code; c2c[code] = (uint8_t)ch; }
            }
            for (int ch = 128; ch < 256 && ok; ++ch) if (hp->dna_code[ch] >= 0) ok = false;      // (the byte masks work on 7-bit characters)
            // an index no base owns must not pass for a base: it maps to code 0, whose character then differs from the input --
            // unless the input IS that character with another index, which cannot be (one index per character)
            if (ok) {
                fast3 = true; g.h_shift = sh;
                memcpy(&g.i2c_lo, i2c, 4); memcpy(&g.i2c_hi, i2c + 4, 4); memcpy(&g.c2c_lo, c2c, 4); memcpy(&g.c2c_hi, c2c + 4, 4);
            }
        }
    }
    fast = ((fast && nbases == 4) || fast3) && nq >= 1 && qmaxc - qmin + 1 == nq && qmaxc < 128 && ntrick_bases <= 1 &&
           (ntrick_bases == 0 || (hp->n_qual[nchar] < 128 && nchar < 128));
    if (fast)
        for (int i = qmin; i <= qmaxc; ++i) fast = fast && hp->qual_code[i] == i - qmin;
    g.q_addlo = g.q_addhi = g.n_char = g.n_code = g.n_qchar = 0;
    if (fast) {
        g.q_addlo = 0x01010101u * (uint32_t)(0x80 - qmin);
        g.q_addhi = 0x01010101u * (uint32_t)(0x80 - qmin - nq);
        if (ntrick_bases == 1) {
            g.n_char = 0x01010101u * (uint32_t)nchar; g.n_code = 0x01010101u * (uint32_t)hp->n_qual[nchar];
            g.n_qchar = 0x01010101u * (uint32_t)(qmin + hp->n_qual[nchar]);       // (a code beyond the alphabet -- Q9 -- has no fused kernel: below)
        }
    }
    if (d_stats && ntrick_bases == 1 && hp->n_qual[nchar] >= nq) return 0;    // the N-trick code is no quality of the alphabet (Q9): exact kernels only
    if (d_stats && (!fast || (bd != 2 && bd != 3) || bq > 6)) return 0;  // the fused kernels exist for the lookup-free paths (2- / 3-bit bases, <= 64 contiguous qualities) only
    const bool var_deal = d_stats && use_lists && hp->variable;
    const size_t lds_static = d_stats ? pks_words((int)bd, ntrick || d_q != nullptr) * 4 : 4;        // the kernels' count table
    const size_t lds = 16 + (size_t)g.stage_bytes + g.out_bytes + (4 * R + 4) * 4 + 3 * 512 + (d_stats ? sizeof(QnLds) : 0);
    UQ_REQUIRE(lds + lds_static <= 160 * 1024, "uq_pack: tile needs %zu bytes of LDS", lds + lds_static);
    const uint64_t tiles = (nreads + R - 1) / R;
    PackKernel k = d_stats ? pick_stats_kernel((int)bd, (int)bq, ntrick, d_q != nullptr, use_lists, var_deal) : pick_kernel((int)bd, (int)bq, ntrick, fast);
    if (lds > 48 * 1024) UQ_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // persistent workgroups: exactly as many as are resident at once (LDS and registers both limit that: a grid sized from the LDS
    // alone would leave the kernels built for four waves per SIMD with a second, quarter-full round of workgroups)
    uint32_t per_cu = (uint32_t)((160 * 1024) / (lds + lds_static));
    {
        // (computed from the kernel's register count: hipOccupancyMaxActiveBlocksPerMultiprocessor is one workgroup per CU high for
        // kernels with 97-112 SGPRs on this ROCm -- MI355X_MICROARCH.md, Correctness boundaries -- which these are)
        int cached_regs = 0;
        for (int i = 0; i < ctx->kreg_n; ++i) if (ctx->kreg_key[i] == (const void*)k) cached_regs = ctx->kreg_val[i];
        if (cached_regs == 0) {
            hipFuncAttributes fa;
            UQ_CHECK_HIP(hipFuncGetAttributes(&fa, (const void*)k));
            cached_regs = fa.numRegs > 0 ? fa.numRegs : 128;
            const int slot = ctx->kreg_n < 8 ? ctx->kreg_n++ : 7;                  // (a context alternates between two or three instances)
            ctx->kreg_key[slot] = (const void*)k; ctx->kreg_val[slot] = cached_regs;
        }
        const uint32_t alloc = ((uint32_t)(cached_regs > 0 ? cached_regs : 128) + 7) & ~7u;          // VGPRs are handed out in eights
        uint32_t waves_per_simd = 512 / alloc; if (waves_per_simd > 8) waves_per_simd = 8; if (waves_per_simd < 1) waves_per_simd = 1;
        const uint32_t waves_per_wg = PK_THREADS / 64;
        const uint32_t by_regs = 4 * waves_per_simd / waves_per_wg;
        if (by_regs >= 1 && by_regs < per_cu) per_cu = by_regs;
    }
    if (per_cu > 6) per_cu = 6;
    if (per_cu < 1) per_cu = 1;
    const uint64_t blocks = tiles < (uint64_t)UQ_NUM_CU * per_cu ? tiles : (uint64_t)UQ_NUM_CU * per_cu;
    PackLists pl;
    memset(&pl, 0, sizeof(pl));
    const TileRec* trec = nullptr;
    if (use_lists) {
        const uint32_t mis = (uint32_t)((uintptr_t)d_buf & 15);
        pl.cv.list = ctx->idx_bitmap; pl.cv.offs = ctx->idx_partials; pl.cv.mis = mis;
        pl.cv.nb = (((ctx->async_nbytes + mis + 15) / 16) * 16 + CV_TILE - 1) / CV_TILE;
        const uint64_t entries = tiles + 1;
        void* scr;
        UQ_TRY(uq_scratch(ctx, entries * sizeof(TileRec) + 64, &scr));
        TileRec* rec = (TileRec*)scr;
        pl.over = (const uint32_t*)(ctx->idx_bitmap + pl.cv.nb * CV_LIST_CAP);        // index.hip: the 16 spare bytes behind the slots
        tile_origin_kernel<<<(uint32_t)((entries + 255) / 256), 256, 0, ctx->stream>>>(pl.cv, d_async, pl.over, first_read, nreads, R, entries, rec);
        UQ_LAUNCH_CHECK();
        trec = rec;
    }
    k<<<(uint32_t)blocks, PK_THREADS, lds, ctx->stream>>>(d_buf, d_line_start, first_read, nreads, lut, g, d_dna, d_qual,
                                                         (unsigned long long*)d_bad, d_stats, d_async, d_q, d_vals, vals_pitch, pl, trec);
    UQ_LAUNCH_CHECK();
    if (h_fused) *h_fused = 1;
    return 0;
}

extern "C" int uq_pack(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t first_read,
                       uint64_t nreads, const uq_pack_params* hp, uint8_t* d_dna, uint8_t* d_qual, uint64_t* d_bad) {
    return pack_impl(ctx, d_buf, d_line_start, first_read, nreads, hp, d_dna, d_qual, d_bad, nullptr, nullptr);
}

extern "C" int uq_pack_stats(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t first_read,
                             uint64_t nreads, const uq_pack_params* h_guess, uint8_t* d_dna, uint8_t* d_qual, uint64_t* d_bad,
                             uq_stats* d_stats, int* h_fused) {
    UQ_REQUIRE(d_stats && h_fused, "uq_pack_stats: null argument");
    return pack_impl(ctx, d_buf, d_line_start, first_read, nreads, h_guess, d_dna, d_qual, d_bad, d_stats, h_fused);
}

// uq_pack_stats behind a census still in flight (uq_count_lines_end_async + uq_index_lines_async): the number of reads is taken on the
// device; `capacity_reads` is what d_dna / d_qual hold (more reads than that: the statistics come back flagged incomplete).
extern "C" int uq_pack_stats_async(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t capacity_reads,
                                   const uq_pack_params* h_guess, uint8_t* d_dna, uint8_t* d_qual, uint64_t* d_bad,
                                   uq_stats* d_stats, int* h_fused) {
    UQ_REQUIRE(d_stats && h_fused, "uq_pack_stats_async: null argument");
    UQ_REQUIRE(ctx && ctx->async_buf == d_buf, "uq_pack_stats_async: not the buffer of the last uq_count_lines_end_async");
    UQ_TRY(pack_impl(ctx, d_buf, d_line_start, 0, capacity_reads, h_guess, d_dna, d_qual, d_bad, d_stats, h_fused, ctx->d_async));
    return uq_async_read_back(ctx);
}

// uq_pack_stats / uq_pack_stats_async with the QNAME phase (qname_fused.hip made the guess in *d_q): d_vals holds UQ_QF_MAXC columns
// of vals_pitch uint32 each.  *h_fused = 0: this geometry has no fused kernel, nothing ran (d_q->nreads stays 0).
extern "C" int uq_pack_stats_qname(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t first_read, uint64_t nreads,
                                   const uq_pack_params* h_guess, uint8_t* d_dna, uint8_t* d_qual, uint64_t* d_bad, uq_stats* d_stats,
                                   uq_qname_fused* d_q, uint32_t* d_vals, uint64_t vals_pitch, int* h_fused) {
    UQ_REQUIRE(d_stats && h_fused && d_q && d_vals && vals_pitch >= nreads, "uq_pack_stats_qname: null argument or columns shorter than the reads");
    return pack_impl(ctx, d_buf, d_line_start, first_read, nreads, h_guess, d_dna, d_qual, d_bad, d_stats, h_fused, nullptr, d_q, d_vals, vals_pitch);
}

extern "C" int uq_pack_stats_qname_async(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t capacity_reads,
                                         const uq_pack_params* h_guess, uint8_t* d_dna, uint8_t* d_qual, uint64_t* d_bad, uq_stats* d_stats,
                                         uq_qname_fused* d_q, uint32_t* d_vals, uint64_t vals_pitch, int* h_fused) {
    UQ_REQUIRE(d_stats && h_fused && d_q && d_vals && vals_pitch >= capacity_reads, "uq_pack_stats_qname_async: null argument or columns shorter than the capacity");
    UQ_REQUIRE(ctx && ctx->async_buf == d_buf, "uq_pack_stats_qname_async: not the buffer of the last uq_count_lines_end_async");
    UQ_TRY(pack_impl(ctx, d_buf, d_line_start, 0, capacity_reads, h_guess, d_dna, d_qual, d_bad, d_stats, h_fused, ctx->d_async, d_q, d_vals, vals_pitch));
    return uq_async_read_back(ctx);
}
