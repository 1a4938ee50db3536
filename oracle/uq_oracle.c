/* uq_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the byte-level loops of the reference `uq.py` (JohnLonginotto/uq), used to
 * check the HIP path at sizes the Python restatement (oracle/uq_oracle.py) is too slow for, and as the
 * optional C "port" leg of bench.py's cpu_baseline.  Nothing under uq_amd/ links or loads it.
 * Parity status: pinned through oracle/uq_oracle.py -- tests/test_oracle_c.py checks these loops
 * against that restatement, which is itself checked against the reference's own outputs
 * (tests/golden/, see oracle/uq_oracle.py header).
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC -> oracle/liboracle.so)
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/* uq.py:85 (`wc -l`) + the next(f) iteration: line_start[k] = offset of line k, line_start[nlines] = end.
 * Returns the number of lines; writes at most cap+1 entries. */
uint64_t uqo_index_lines(const uint8_t* buf, uint64_t nbytes, uint64_t* line_start, uint64_t cap) {
    uint64_t k = 0;
    if (line_start && cap + 1 > 0) line_start[0] = 0;
    for (uint64_t p = 0; p < nbytes; ++p)
        if (buf[p] == '\n') {
            ++k;
            if (line_start && k <= cap) line_start[k] = p + 1;
        }
    return k;
}

/* uq.py:366-375, 382, 388, 415-425: counts[base*256+qual], length range, first bad records,
 * first_seen[base] = (read << 20 | pos) of the first occurrence. */
void uqo_stats(const uint8_t* buf, const uint64_t* ls, uint64_t first, uint64_t n, uint64_t* counts /*65536*/,
               uint64_t* first_seen /*256*/, uint32_t* len_min, uint32_t* len_max, uint32_t* max_record_bytes,
               uint64_t* bad_plus, uint64_t* bad_len) {
    for (uint64_t r = 0; r < n; ++r) {
        const uint64_t* p = ls + 4 * (first + r);
        uint64_t s = p[1], e1 = p[2], q = p[3], e2 = p[4];
        uint32_t L = (uint32_t)(e1 - s - 1), Lq = (uint32_t)(e2 - q - 1);
        if (buf[e1] != '+' && first + r < *bad_plus) *bad_plus = first + r;
        if (L != Lq && first + r < *bad_len) *bad_len = first + r;
        if (L < *len_min) *len_min = L;
        if (L > *len_max) *len_max = L;
        if ((uint32_t)(e2 - p[0]) > *max_record_bytes) *max_record_bytes = (uint32_t)(e2 - p[0]);
        uint32_t Lc = L < Lq ? L : Lq;
        for (uint32_t j = 0; j < Lc; ++j) {
            uint8_t b = buf[s + j], c = buf[q + j];
            counts[(uint32_t)b * 256 + c] += 1;
            uint64_t key = ((first + r) << 20) | (j & 0xFFFFFu);
            if (first_seen && key < first_seen[b]) first_seen[b] = key;
        }
    }
}

/* uq.py:132-175 (encoder_fixed) / 205-247 (encoder_variable): the per-base loop on the reversed read,
 * `+=` accumulation, `while bits_done > 8` flush, final byte(s) with the sentinel; carries propagate
 * upward (Q7 rule), bytes never written are zero (Q8).  Returns the first read with a symbol that has
 * no code, or UINT64_MAX. */
uint64_t uqo_pack(const uint8_t* buf, const uint64_t* ls, uint64_t first, uint64_t n, const int16_t* dna_code,
                  const int16_t* qual_code, const int32_t* n_qual, int bits_per_base, int bits_per_quality,
                  int variable, uint32_t Cd, uint32_t Cq, uint8_t* dna, uint8_t* qual) {
    uint64_t bad = UINT64_MAX;
    memset(dna, 0, (size_t)n * Cd);
    memset(qual, 0, (size_t)n * Cq);
    for (uint64_t r = 0; r < n; ++r) {
        const uint64_t* p = ls + 4 * (first + r);
        uint64_t s = p[1], q = p[3];
        uint32_t L = (uint32_t)(p[2] - s - 1);
        uint8_t* drow = dna + r * Cd;
        uint8_t* qrow = qual + r * Cq;
        uint64_t td = 0, tq = 0;
        int dbits = 0, qbits = 0;
        int64_t pd = (int64_t)Cd - 1, pq = (int64_t)Cq - 1;
        for (uint32_t t = 0; t < L; ++t) {
            uint8_t cb = buf[s + L - 1 - t], cc = buf[q + L - 1 - t];
            int dc = dna_code[cb], qc = qual_code[cc];
            if (dc < 0) { dc = 0; qc = n_qual[cb]; }
            if (qc < 0) { if (r < bad) bad = r; qc = 0; }
            td += (uint64_t)dc << dbits;
            tq += (uint64_t)qc << qbits;
            dbits += bits_per_base;
            qbits += bits_per_quality;
            while (dbits > 8) { dbits -= 8; if (pd >= 0) drow[pd] = (uint8_t)td; td >>= 8; --pd; }
            while (qbits > 8) { qbits -= 8; if (pq >= 0) qrow[pq] = (uint8_t)tq; tq >>= 8; --pq; }
        }
        td += (uint64_t)(variable ? 1 : 0) << dbits;
        tq += (uint64_t)(variable ? 1 : 0) << qbits;
        while (td && pd >= 0) { drow[pd] = (uint8_t)td; td >>= 8; --pd; }
        while (tq && pq >= 0) { qrow[pq] = (uint8_t)tq; tq >>= 8; --pq; }
    }
    return bad;
}

/* uq.py:1002-1007 + 1031-1054 on codes: row -> symbols MSB first -> characters; N restore through
 * qual_n_base; variable length strips everything up to and including the first code-1 DNA symbol.
 * seq / qtxt are [n][dna_max], len[r] = read length.  Returns first row without sentinel or UINT64_MAX. */
uint64_t uqo_unpack(const uint8_t* dna, const uint8_t* qual, uint64_t n, const uint8_t* base_char,
                    const uint8_t* qual_char, const uint8_t* qual_n_base, int bits_per_base, int bits_per_quality,
                    int variable, uint32_t Cd, uint32_t Cq, uint32_t dna_max, uint8_t* seq, uint8_t* qtxt, uint32_t* len) {
    uint64_t bad = UINT64_MAX;
    uint32_t Lv = dna_max + (variable ? 1 : 0);
    for (uint64_t r = 0; r < n; ++r) {
        const uint8_t* drow = dna + r * Cd;
        const uint8_t* qrow = qual + r * Cq;
        uint32_t start = 0;
        uint32_t out = 0;
        int found = !variable;
        for (uint32_t k = 0; k < Lv; ++k) {   /* symbol k counted from the MSB end */
            uint32_t t = Lv - 1 - k;          /* index from the LSB end */
            uint32_t cd = 0, cq = 0;
            for (int i = 0; i < bits_per_base; ++i) {
                uint64_t bit = (uint64_t)t * bits_per_base + i;
                cd |= ((drow[Cd - 1 - bit / 8] >> (bit % 8)) & 1u) << i;
            }
            for (int i = 0; i < bits_per_quality; ++i) {
                uint64_t bit = (uint64_t)t * bits_per_quality + i;
                cq |= ((qrow[Cq - 1 - bit / 8] >> (bit % 8)) & 1u) << i;
            }
            if (!found) {
                if (cd == 1) { found = 1; start = k + 1; }
                continue;
            }
            uint8_t b = qual_n_base[cq] ? qual_n_base[cq] : base_char[cd];
            seq[r * dna_max + out] = b;
            qtxt[r * dna_max + out] = qual_char[cq];
            ++out;
        }
        (void)start;
        if (!found && r < bad) bad = r;
        len[r] = out;
        for (uint32_t k = out; k < dna_max; ++k) { seq[r * dna_max + k] = 0; qtxt[r * dna_max + k] = 0; }
    }
    return bad;
}

/* uq.py:263-270: payload bytes of numpy.save(rot90(T, k) in C or F order) -- SURVEY.md A.4 closed forms. */
void uqo_pattern(const uint8_t* T, uint64_t R, uint32_t Cc, int pattern_id, uint8_t* out) {
    uint64_t C = Cc, total = R * C;
    int k = pattern_id >> 1, f = pattern_id & 1;
    for (uint64_t r = 0; r < R; ++r)
        for (uint64_t c = 0; c < C; ++c) {
            uint8_t v = T[r * C + c];
            uint64_t pos;
            switch (k * 2 + f) {
                case 0: pos = r * C + c; break;                       /* 0.1 */
                case 1: pos = c * R + r; break;                       /* 0.2 */
                case 2: pos = (C - 1 - c) * R + r; break;             /* 1.1 */
                case 3: pos = r * C + (C - 1 - c); break;             /* 1.2 */
                case 4: pos = total - 1 - (r * C + c); break;         /* 2.1 */
                case 5: pos = total - 1 - (c * R + r); break;         /* 2.2 */
                case 6: pos = total - 1 - ((C - 1 - c) * R + r); break; /* 3.1 */
                default: pos = total - 1 - (r * C + (C - 1 - c)); break; /* 3.2 */
            }
            out[pos] = v;
        }
}
