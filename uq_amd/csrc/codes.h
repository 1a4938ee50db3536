// codes.h -- code -> character decoding shared by the unpack kernels (unpack.hip) and the fused decode (emit.hip).
// The tables restate the decoder's character maps (uq.py:1031-1054): base code -> base, quality code -> quality
// character, and the N-trick's quality code -> base that replaces the decoded one.
#pragma once
#include "common.h"

struct UnpackLut { uint8_t base_char[256], qual_char[256], qual_n_base[256]; };

// The lookup-free path: 2- or 3-bit bases (base_tab / base_tab_hi = the characters of codes 0..3 / 4..7, what v_perm_b32 selects
// from; a code beyond the alphabet decodes to 0, as through the tables), quality character = code + qmin for the nq real codes, at
// most one quality code that stands for an N-trick base.  Anything else decodes through the tables.
struct FastAlphabet { uint32_t fast, base_tab, qmin4, q_over, has_n, n_code4, n_char4, base_tab_hi, bd; };

static inline FastAlphabet fast_alphabet(const uq_unpack_params* hp) {
    FastAlphabet a;
    memset(&a, 0, sizeof(a));
    const int bd = hp->bits_per_base, bq = hp->bits_per_quality;
    int nq = 0;
    while (nq < (1 << bq) && hp->qual_char[nq] != 0) ++nq;
    bool ok = (bd == 2 || bd == 3) && bq <= 7 && nq >= 1;
    for (int c = 0; ok && c < nq; ++c) ok = hp->qual_char[c] == hp->qual_char[0] + c;
    for (int c = nq; ok && c < (1 << bq); ++c) ok = hp->qual_char[c] == 0 && hp->qual_n_base[c] == 0;
    int ncodes = 0, ncode = 0;
    for (int c = 0; ok && c < nq; ++c) if (hp->qual_n_base[c]) { ++ncodes; ncode = c; }
    ok = ok && ncodes <= 1;
    if (bd == 2) for (int c = 0; ok && c < 4; ++c) ok = hp->base_char[c] != 0;       // (the 2-bit kernels of unpack.hip take all four for granted)
    if (!ok) return a;
    a.fast = 1; a.bd = (uint32_t)bd;
    a.base_tab = (uint32_t)hp->base_char[0] | ((uint32_t)hp->base_char[1] << 8) | ((uint32_t)hp->base_char[2] << 16) | ((uint32_t)hp->base_char[3] << 24);
    if (bd == 3) a.base_tab_hi = (uint32_t)hp->base_char[4] | ((uint32_t)hp->base_char[5] << 8) | ((uint32_t)hp->base_char[6] << 16) | ((uint32_t)hp->base_char[7] << 24);
    a.qmin4 = 0x01010101u * hp->qual_char[0];
    a.q_over = 0x01010101u * (uint32_t)(0x80 - nq);
    a.has_n = (uint32_t)ncodes; a.n_code4 = 0x01010101u * (uint32_t)ncode; a.n_char4 = 0x01010101u * hp->qual_n_base[ncode];
    if (bd == 2 && ncodes) a.base_tab_hi = a.n_char4;        // selectors 4 .. 7 (no base of a 2-bit alphabet): the N-trick's character (emit.hip: n_selectors)
    return a;
}

// bytes i = b*g .. b*g + b - 1 counted from the row's LAST byte, little-endian into a u64: the b bytes that hold
// symbols 8 g .. 8 g + 7 (symbol 0 is the read's last character)
__device__ __forceinline__ uint64_t group_bits(const uint8_t* row, uint32_t C, uint32_t b, uint32_t g) {
    uint64_t v = 0;
    for (uint32_t i = 0; i < b; ++i) {
        uint32_t bi = b * g + i;
        if (bi < C) v |= (uint64_t)row[C - 1 - bi] << (8 * i);
    }
    return v;
}

// read length of a variable-length DNA row: the sentinel is the only 1 above the payload.
// `first` = index of the row's first non-zero byte (C when there is none).  Returns false for a row that no encoder writes.
__device__ __forceinline__ bool row_length(const uint8_t* row, uint32_t C, uint32_t first, uint32_t bd, uint32_t dmax, uint32_t& L) {
    if (first >= C) { L = 0; return false; }
    const uint32_t hb = 8 * (C - 1 - first) + (31 - __clz((uint32_t)row[first]));
    L = hb / bd;
    if (L * bd != hb || L > dmax) { L = L > dmax ? dmax : L; return false; }
    return true;
}
