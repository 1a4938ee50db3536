// encode.hip -- the encode hot path in ONE pass over the FASTQ stream (SURVEY.md 8 rows index + a1 + a3 / a4).
// Replaces, in a single read of the file, what the reference does in three: `wc -l` + line iteration (uq.py:85,
// 132-137), the pass-1 histogram and checks (uq.py:366-388, 415-425) and `encoder_fixed` / `encoder_variable`
// (uq.py:108-254).  The multi-pass form (index.hip -> stats.hip -> pack.hip) reads the 340 B of a record three times;
// none of those kernels can go faster than the memory fabric lets it, so the way to a faster step is to
// read the stream once.
//
// What stood in the way, and how this kernel gets round it:
//   * records are found from newlines, and a record's number needs the count of ALL newlines in front of it.  A
//     workgroup owns a 16 KiB BYTE tile (+ 1 KiB halo), counts its newlines in registers (the census of index.hip),
//     publishes the count in a status word and obtains the count of everything in front of the tile by decoupled
//     look-back over the status words of the preceding tiles (one 8-byte {flag, value} granule per tile, relaxed
//     agent-scope atomics: a granule needs no ordering).  Tiles are dealt round-robin to a grid that is resident as a
//     whole; every spin is bounded and a stuck look-back aborts the launch (the caller then runs the multi-pass form).
//   * the pack needs the decisions (alphabets, N-trick, bit widths), which need the statistics of the whole file.  The
//     kernel packs with GUESSED decisions (the caller's: a sample of this file, the previous file) while it counts the
//     (base, quality) pairs it holds in registers anyway; the caller derives the real decisions from the counts and
//     keeps the tables iff they equal the guess.  Nothing is assumed: a symbol or a read length outside the guess
//     raises `mismatch`, a record the tile cannot see whole raises `incomplete`, and the caller falls back.
//   * a tile owns the records whose FIRST line starts in it (the line starts = the bytes after its newlines; tile 0
//     also owns the start of the stream); their tails reach into the halo.  The tile's line starts, in stream order,
//     ARE the `meta` array pack.hip reads from the global index, so phase B below is pack_tile_kernel's lookup-free path.
// Per tile: registers -> census (SWAR newline masks, two wave scans) -> aggregate published -> line-start list + bytes to
// LDS, next tile's loads issued -> look-back -> record index written (8 B per line) -> P lanes per read pack 8 symbols of
// both streams per step into the LDS row images -> images stored with aligned 16-byte vectors (the image is skewed to the
// destination's alignment: a tile's first row is wherever the previous tile's last one ended).
// Algorithmic HBM bytes per read: record bytes read once (+ 6 % halo) + C_dna + C_qual + 32 B of index written.
#include "common.h"
#include "swar.h"
#include "histo.h"
#include <stdlib.h>

namespace {
constexpr int EN_THREADS = 256;
constexpr int EN_NV = 4;                                   // 16-byte loads per lane per tile
constexpr uint32_t EN_TV = EN_NV * EN_THREADS;             // vectors per tile (1024)
constexpr uint32_t EN_TILE = EN_TV * 16;                   // 16 KiB
constexpr uint32_t EN_HV = 64;                             // halo vectors: the last wave loads one more vector per lane
constexpr uint32_t EN_HALO = EN_HV * 16;                   // 1 KiB
constexpr uint32_t EN_ECAP = 1536;                         // line starts a tile (+ halo) may hold
constexpr uint32_t EN_SPIN_MAX = 1u << 22;                 // look-back polls before the launch is given up
constexpr unsigned long long EN_VALUE = (1ull << 62) - 1;

struct EncGeom {
    uint64_t nbytes, nvec, ntiles, max_reads;
    uint32_t mis;                  // buf - abuf (abuf = buf rounded down to 16 bytes)
    uint32_t Cd, Cq, G, variable, dna_max;
    uint32_t fill_d, fill_q;       // the characters with code 0, replicated in 4 bytes
    uint32_t P, magicP;            // lanes per read in phase B
    uint32_t rmax;                 // reads per batch: what the row images hold
    uint32_t img_d, img_q;         // LDS bytes of the two row images (16-byte multiples, room for the skew)
    uint32_t q_addlo, q_addhi, n_char, n_code;      // lookup-free conversion, as pack.hip
    uint32_t debug;                // timing experiments (UQ_ENC_DEBUG; results are wrong): bit 0 = no look-back, bit 1 = no publish either
};

struct EncCtl {
    unsigned long long nlines;
    uint32_t abort, incomplete, mismatch, index_overflow, row_overflow, pad;
};

__device__ __forceinline__ uint32_t acgt_codes(uint32_t w) { return ((w ^ (w >> 1)) >> 1) & 0x03030303u; }
__device__ __forceinline__ uint32_t acgt_chars(uint32_t codes) { return __builtin_amdgcn_perm(0u, 0x54474341u, codes); }
template <int B>
__device__ __forceinline__ uint32_t pack4(uint32_t x) {
    if (B == 8) return __builtin_amdgcn_perm(0u, x, 0x00010203u);
    if (B == 2) return (x * 0x40100401u) >> 24;
    const uint32_t c0 = x & ((1u << B) - 1), c1 = (x >> 8) & ((1u << B) - 1), c2 = (x >> 16) & ((1u << B) - 1), c3 = x >> 24;
    return (((((c0 << B) | c1) << B) | c2) << B) | c3;
}

// LDS image -> global span: image byte `skew + k` is destination byte k; skew = destination address & 15, so whole
// 16-byte vectors of the image are whole aligned vectors of the destination.  Edge bytes go one by one.
__device__ __noinline__ void store_image(uint8_t* gdst, const uint8_t* img, uint32_t skew, uint32_t nb, uint32_t tid) {
    const uint32_t end = skew + nb;
    const uint32_t v0 = (skew + 15) >> 4, v1 = end >> 4;            // full vectors [v0, v1)
    uint8_t* g0 = gdst - skew;                                      // 16-byte aligned
    if (v1 > v0) {
        for (uint32_t i = v0 + tid; i < v1; i += EN_THREADS) ((uint4*)g0)[i] = ((const uint4*)img)[i];
        const uint32_t head = (v0 << 4) - skew, tail = end - (v1 << 4);
        if (tid < head) gdst[tid] = img[skew + tid];
        if (tid >= 64 && tid - 64 < tail) g0[(v1 << 4) + tid - 64] = img[(v1 << 4) + tid - 64];
    } else {
        for (uint32_t i = tid; i < nb; i += EN_THREADS) gdst[i] = img[skew + i];
    }
}

// The number of newlines in front of tile t: decoupled look-back on TWO levels.  One level (every tile sums the counts of
// the tiles in front of it back to the nearest tile that already knows its own answer) moves 64 tiles per memory round
// trip: 207 000 tiles of a 3.4 GB shard at ~1.2 us a hop is slower than the rest of the kernel (measured: +1.2 ms).  So
// tiles are taken in groups of 64:
//   A[t]   tile t's own count            flag 1 << 62 | count          written once, by the tile
//   GS[g]  group g's running sum         arrived << 40 | sum           every tile adds 1 << 40 | count; complete at 64 arrivals
//   GP[g]  count up to and including g   flag 1 << 62 | value          written by the group's last arriver after ITS look-back
// A tile adds up (1) the counts of the tiles in front of it inside its group (one wave-load of A) and (2) the groups in
// front of its group, newest first, until one has its GP (one wave-load each of GP and GS covers 64 groups = 4096 tiles).
// Every word is one self-contained 8-byte granule read and written with relaxed agent-scope atomics; every spin is bounded.
constexpr uint32_t EN_GROUP = 64;
constexpr uint32_t EN_GWIN = 16;                          // groups asked about per round trip (a resident grid spans 12)
constexpr unsigned long long EN_SUM40 = (1ull << 40) - 1;
constexpr uint32_t EN_GSTRIDE = 16;                       // u64 words per group record: GS and GP of a group share a 128-byte line of
                                                          // their own -- 768 tiles arriving at counters that sit in ONE line serialise
                                                          // at that line's memory channel (measured: 37 us per round of the grid)
struct LookBack { unsigned long long *A, *GR; };         // GR[g * EN_GSTRIDE + 0] = GS[g], + 1 = GP[g]

__device__ __forceinline__ bool lb_spin(uint32_t& spins, EncCtl* ctl) {
    ++spins;
    if (spins > EN_SPIN_MAX || ((spins & 255u) == 0 && __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return true;
    __builtin_amdgcn_s_sleep(1);
    return false;
}

// thread 0, as soon as the tile's count is known: publish it, join the group.  Returns GS[g] as it was before.
__device__ __forceinline__ unsigned long long lb_arrive(const LookBack& lb, uint64_t t, uint32_t count) {
    __hip_atomic_store(&lb.A[t], (1ull << 62) | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __hip_atomic_fetch_add(&lb.GR[(t / EN_GROUP) * EN_GSTRIDE], (1ull << 40) | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave 0; `old` = lb_arrive's return value in lane 0
__device__ __forceinline__ uint64_t lookback(const LookBack& lb, EncCtl* __restrict__ ctl, uint64_t t, uint64_t ntiles, uint32_t count,
                                             unsigned long long old, uint32_t lane, bool& dead) {
    const uint64_t g = t / EN_GROUP;
    const uint32_t j = (uint32_t)(t % EN_GROUP);
    const uint64_t left = ntiles - g * EN_GROUP;
    const uint32_t n_g = left < EN_GROUP ? (uint32_t)left : EN_GROUP;
    uint32_t spins = 0;
    dead = false;
    uint64_t own = 0, exg = 0;
    int64_t pos = (int64_t)g - 1;
    bool need_own = j != 0, need_groups = pos >= 0;
    // all three wave-loads of a round are in flight together; in the steady state one round answers both questions
    while (need_own || need_groups) {
        const int64_t idx = pos - (int64_t)lane;
        unsigned long long a = 1ull << 62, gp = 1ull << 62, gs = 0;
        if (need_own && lane < j) a = __hip_atomic_load(&lb.A[g * EN_GROUP + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (need_groups && idx >= 0 && lane < EN_GWIN) {
            gp = __hip_atomic_load(&lb.GR[idx * EN_GSTRIDE + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gs = __hip_atomic_load(&lb.GR[idx * EN_GSTRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool progress = false;
        if (need_own && !__ballot((a >> 62) == 0)) {                       // (1) the tiles in front, inside the group
            own = wave_sum<uint64_t>(lane < j ? (uint64_t)(a & EN_VALUE) : 0ull);
            need_own = false; progress = true;
        }
        if (need_groups) {                                                 // (2) the groups in front, 64 per round trip
            uint32_t flag = 2;
            uint64_t value = 0;
            if (lane >= EN_GWIN) flag = 3;                                     // not asked
            else if (idx >= 0) {
                if (gp >> 62) value = gp & EN_VALUE;
                else if ((uint32_t)(gs >> 40) == EN_GROUP) { flag = 1; value = gs & EN_SUM40; }   // groups in front of another are full ones
                else flag = 0;
            }
            const unsigned long long bp = __ballot(flag == 2), bi = __ballot(flag == 0);
            const uint32_t fp = bp ? (uint32_t)__ffsll((long long)bp) - 1u : 64u, fi = bi ? (uint32_t)__ffsll((long long)bi) - 1u : 64u;
            if (fi >= fp || fi == 64) {                                    // no incomplete group this side of the nearest GP
                exg += wave_sum<uint64_t>(lane <= (fp < 64 ? fp : EN_GWIN - 1) && lane < EN_GWIN ? value : 0ull);
                if (fp < 64) need_groups = false; else { pos -= EN_GWIN; need_groups = pos >= 0; }
                progress = true;
            }
        }
        if (!progress && lb_spin(spins, ctl)) { dead = true; break; }
    }
    old = __shfl(old, 0, 64);
    const bool last = (uint32_t)(old >> 40) == n_g - 1;                    // this tile completes its group
    const uint64_t total_g = (old & EN_SUM40) + count;
    if (lane == 0) {
        if (dead) __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (last) __hip_atomic_store(&lb.GR[g * EN_GSTRIDE + 1], (1ull << 62) | (exg + total_g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return exg + own;
}

// LDS carve (dynamic): [16 B guard][stage: tile + halo + 32][DNA image][QUAL image][E: line starts][misc 64 B][count tables]
template <int BQ, bool NTRICK, bool STATS>
__global__ __launch_bounds__(EN_THREADS, STATS ? 3 : 4) void encode_tile_kernel(const uint4* __restrict__ abuf, EncGeom g,
                                                                               LookBack lb, EncCtl* __restrict__ ctl,
                                                                               uint64_t* __restrict__ line_start, uint8_t* __restrict__ dna,
                                                                               uint8_t* __restrict__ qual, uq_stats* __restrict__ st, uint32_t win) {
    constexpr int BD = 2;
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* stage = smem + 16;
    uint8_t* img_d = stage + EN_TILE + EN_HALO + 32;
    uint8_t* img_q = img_d + g.img_d;
    uint32_t* E = (uint32_t*)(img_q + g.img_q);
    uint32_t* misc = E + EN_ECAP;                  // [0..3] wave totals, [4] halo total, [5] look-back lo, [6] hi, [7] dead
    const uint32_t tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    Histo hz;
    RecordAcc acc;
    if (STATS) hz.init(misc + 16, st, win);        // tables zeroed; the first tile's barriers order it
    bool incomplete = false, mismatch = false, idx_over = false, row_over = false;

    const uint64_t S = gridDim.x;
    uint32_t rr, pp;                               // this lane packs groups pp, pp + P, ... of read rr of a batch
    fast_divmod(tid, g.P, g.magicP, rr, pp);
    // vector `u` of this lane inside a tile: wave w owns the 4 KiB chunk w (stream order = wave, load, lane, byte); the last
    // wave also holds the halo, so the newline ranks of the halo continue the tile's
    const uint32_t vl0 = w * (EN_NV * 64) + lane;
    uint4 v[EN_NV + 1];
    auto issue = [&](uint64_t t) {
#pragma unroll
        for (int u = 0; u <= EN_NV; ++u) v[u] = make_uint4(0, 0, 0, 0);
        if (t >= g.ntiles) return;
        const uint64_t base = t * EN_TV;
#pragma unroll
        for (int u = 0; u < EN_NV; ++u) { const uint64_t vi = base + vl0 + u * 64; if (vi < g.nvec) v[u] = abuf[vi]; }
        if (w == EN_THREADS / 64 - 1) { const uint64_t vi = base + EN_TV + lane; if (vi < g.nvec) v[EN_NV] = abuf[vi]; }
    };

    uint64_t t = blockIdx.x;
    issue(t);
    for (; t < g.ntiles; t += S) {
        // ---- census of the tile's bytes, in registers (index.hip's list form)
        const int64_t pos0 = (int64_t)(t * EN_TILE) - (int64_t)g.mis;          // stream position of stage byte 0
        uint32_t m[EN_NV + 1];
#pragma unroll
        for (int u = 0; u < EN_NV; ++u) m[u] = nl_mask16(v[u]) & valid_mask16(pos0 + (int64_t)(vl0 + u * 64) * 16, g.nbytes);
        m[EN_NV] = w == EN_THREADS / 64 - 1 ? nl_mask16(v[EN_NV]) & valid_mask16(pos0 + (int64_t)(EN_TV + lane) * 16, g.nbytes) : 0u;
        const uint32_t c0 = __popc(m[0]), c1 = __popc(m[1]), c2 = __popc(m[2]), c3 = __popc(m[3]), c4 = __popc(m[4]);
        const uint32_t i01 = wave_inclusive_sum(c0 | (c1 << 16)), i23 = wave_inclusive_sum(c2 | (c3 << 16));
        const uint32_t i4 = wave_inclusive_sum(c4);
        const uint32_t t01 = __shfl(i01, 63, 64), t23 = __shfl(i23, 63, 64), t4 = __shfl(i4, 63, 64);
        const uint32_t T0 = t01 & 0xFFFFu, T1 = t01 >> 16, T2 = t23 & 0xFFFFu, T3 = t23 >> 16;
        if (lane == 0) { misc[w] = T0 + T1 + T2 + T3; if (w == EN_THREADS / 64 - 1) misc[4] = t4; }
        __syncthreads();                                                       // B1
        uint32_t base = 0, count = 0;
#pragma unroll
        for (uint32_t i = 0; i < EN_THREADS / 64; ++i) { const uint32_t x = misc[i]; if (i < w) base += x; count += x; }
        const uint32_t nhalo = misc[4];
        unsigned long long arrived = 0;
        if (tid == 0 && !(g.debug & 2)) arrived = lb_arrive(lb, t, count);      // published as early as possible
        // ---- line starts (stage offsets of the bytes after the newlines, stream order) and the bytes themselves -> LDS
        const uint32_t shift = t == 0 ? 1u : 0u;                               // tile 0: entry 0 = the start of the stream
        const uint32_t ex[EN_NV + 1] = {base + (i01 & 0xFFFFu) - c0, base + T0 + (i01 >> 16) - c1, base + T0 + T1 + (i23 & 0xFFFFu) - c2,
                                        base + T0 + T1 + T2 + (i23 >> 16) - c3, count + i4 - c4};
#pragma unroll
        for (int u = 0; u <= EN_NV; ++u) {
            uint32_t mm = m[u], k = ex[u] + shift;
            const uint32_t o = (u < EN_NV ? vl0 + u * 64 : EN_TV + lane) * 16 + 1;
            while (mm) {
                const uint32_t b = (uint32_t)__ffs((int)mm) - 1u;
                mm &= mm - 1;
                if (k < EN_ECAP) E[k] = o + b; else idx_over = true;
                ++k;
            }
        }
        if (shift && tid == 0) E[0] = g.mis;
#pragma unroll
        for (int u = 0; u < EN_NV; ++u) ((uint4*)stage)[vl0 + u * 64] = v[u];
        if (w == EN_THREADS / 64 - 1) ((uint4*)stage)[EN_TV + lane] = v[EN_NV];
        if (w != 0) issue(t + S);                                              // the next tile's bytes are in flight from here on
        // ---- how many newlines lie in front of this tile.  Wave 0 asks BEFORE it issues any other vector memory operation of
        //      this iteration: an atomic load waits (vmcnt) for every older load of its wave, and behind the next tile's 16 KiB
        //      each poll took an HBM round trip under load instead of a trip to the L2 / fabric (measured: +1.2 ms per launch)
        if (w == 0) {
            bool dead = false;
            const uint64_t excl = (g.debug & 1) ? t * 192 : lookback(lb, ctl, t, g.ntiles, count, arrived, lane, dead);
            if (lane == 0) { misc[5] = (uint32_t)excl; misc[6] = (uint32_t)(excl >> 32); misc[7] = dead ? 1u : 0u; }
        }
        if (w == 0) issue(t + S);
        __syncthreads();                                                       // B2
        if (misc[7]) break;                                                    // the launch is given up (workgroup-uniform)
        const uint64_t Gp = ((uint64_t)misc[6] << 32) | misc[5];
        const uint32_t nown = count + shift;                                   // line starts this tile owns
        const uint32_t nent = nown + nhalo < EN_ECAP ? nown + nhalo : EN_ECAP; // ... and sees
        const uint64_t line0 = shift ? 0ull : Gp + 1;                          // file-wide number of the line that starts at E[0]
        // ---- record index: line_start[line0 + x] = stream position of E[x]
        for (uint32_t x = tid; x < nown && x < EN_ECAP; x += EN_THREADS) {
            const uint64_t li = line0 + x;
            if (li <= 4 * g.max_reads) line_start[li] = (uint64_t)(pos0 + (int64_t)E[x]); else row_over = true;
        }
        if (t == g.ntiles - 1 && tid == 0) ctl->nlines = Gp + count;
        // ---- the records that start in this tile: entries x0, x0 + 4, ... (file-wide line number divisible by 4)
        const uint32_t x0 = (uint32_t)((0ull - line0) & 3ull);
        uint32_t Rt = nown > x0 ? (nown - x0 + 3) >> 2 : 0u;
        const uint64_t r0 = (line0 + x0) >> 2;
        const uint32_t whole = nent >= x0 + 5 ? (nent - 5 - x0) / 4 + 1 : 0u;  // records whose five line starts are all in sight
        if (whole < Rt) {
            // the tail of a record is out of sight: beyond the halo (the caller's guess of the longest record was too
            // small: multi-pass form), or beyond the end of the stream (an unterminated last line: nobody's record)
            if ((uint64_t)(pos0 + (int64_t)(EN_TILE + EN_HALO)) < g.nbytes) incomplete = true;
            Rt = whole;
        }
        if (r0 + Rt > g.max_reads) { row_over = true; Rt = r0 < g.max_reads ? (uint32_t)(g.max_reads - r0) : 0u; }
        for (uint32_t qb = 0; qb < Rt; qb += g.rmax) {
            const uint32_t Rc = Rt - qb < g.rmax ? Rt - qb : g.rmax;
            const uint32_t* meta = E + x0 + 4 * qb;
            uint8_t* gd = dna + (r0 + qb) * g.Cd;
            uint8_t* gq = qual + (r0 + qb) * g.Cq;
            const uint32_t skew_d = (uint32_t)((uintptr_t)gd & 15), skew_q = (uint32_t)((uintptr_t)gq & 15);
            if (rr < Rc) {
                // ---- B: P lanes per read; a lane owns groups of 8 consecutive symbols of both streams (pack.hip, lookup-free path)
                const uint32_t r = rr;
                const uint32_t so = meta[4 * r + 1];
                uint32_t L = meta[4 * r + 2] - so - 1;
                const uint32_t qo = meta[4 * r + 3];
                const uint32_t Lq = meta[4 * r + 4] - qo - 1;
                if (STATS && pp == 0) acc.record(r0 + qb + r, stage[meta[4 * r + 2]] == '+', L, Lq, meta[4 * r + 4] - meta[4 * r]);
                if (L > g.dna_max || Lq != L) { incomplete = true; mismatch = true; L = 0; }    // symbols this kernel does not visit
                uint8_t* orow_d = img_d + skew_d + r * g.Cd + (g.Cd - 1);
                uint8_t* orow_q = img_q + skew_q + r * g.Cq + (g.Cq - 1);
                for (uint32_t gg = pp; gg < g.G; gg += g.P) {
                    const int32_t j0 = (int32_t)L - 8 * (int32_t)gg - 8;
                    uint64_t vd = 0, vq = 0;
                    if (j0 > -8) {
                        uint32_t b_lo, b_hi, q_lo, q_hi;
                        lds_window8(stage, (int32_t)so + j0, b_lo, b_hi);
                        lds_window8(stage, (int32_t)qo + j0, q_lo, q_hi);
                        if (STATS) hz.group8(b_lo, b_hi, q_lo, q_hi, j0 < 0 ? (uint32_t)(-j0) : 0u, lane);
                        if (j0 < 0) {       // the window reaches above the first base: code-0 characters there
                            uint32_t mlo, mhi;
                            window_masks((uint32_t)(-j0), mlo, mhi);
                            b_lo = bfi(mlo, b_lo, g.fill_d); b_hi = bfi(mhi, b_hi, g.fill_d);
                            q_lo = bfi(mlo, q_lo, g.fill_q); q_hi = bfi(mhi, q_hi, g.fill_q);
                        }
                        uint32_t cd0 = acgt_codes(b_lo), cd1 = acgt_codes(b_hi);
                        const uint32_t e0 = acgt_chars(cd0) ^ b_lo, e1 = acgt_chars(cd1) ^ b_hi;      // non-zero byte = not ACGT
                        const uint32_t u0 = q_lo + g.q_addlo, u1 = q_hi + g.q_addlo;
                        uint32_t bq0 = (q_lo | (q_lo + g.q_addhi) | ~u0) & 0x80808080u;
                        uint32_t bq1 = (q_hi | (q_hi + g.q_addhi) | ~u1) & 0x80808080u;
                        uint32_t x0q = u0 & 0x7F7F7F7Fu, x1q = u1 & 0x7F7F7F7Fu;
                        if (e0 | e1) {
                            if (NTRICK && !((q_lo | q_hi) & 0x80808080u)) {
                                const uint32_t m0 = nonzero_bytes(e0), m1 = nonzero_bytes(e1);
                                if (((b_lo ^ g.n_char) & m0) | ((b_hi ^ g.n_char) & m1)) mismatch = true;    // not the N-trick base
                                cd0 &= ~m0; cd1 &= ~m1;
                                x0q = bfi(m0, g.n_code, x0q); x1q = bfi(m1, g.n_code, x1q);
                                bq0 &= ~m0; bq1 &= ~m1;
                            } else mismatch = true;
                        }
                        if (bq0 | bq1) mismatch = true;                         // a quality outside the guessed range
                        vd = ((uint64_t)pack4<BD>(cd0) << (4 * BD)) | pack4<BD>(cd1);
                        vq = ((uint64_t)pack4<BQ>(x0q) << (4 * BQ)) | pack4<BQ>(x1q);
                    }
                    if (g.variable) {                                          // sentinel = code 1 at symbol index L
                        const int32_t i = (int32_t)L - 8 * (int32_t)gg;
                        if (i >= 0 && i < 8) { vd |= 1ull << (BD * i); vq |= 1ull << (BQ * i); }
                    }
                    uint8_t* od = orow_d - BD * gg;
                    uint8_t* oq = orow_q - BQ * gg;
                    if (gg + 1 < g.G) {
#pragma unroll
                        for (int i = 0; i < BD; ++i) od[-i] = (uint8_t)(vd >> (8 * i));
#pragma unroll
                        for (int i = 0; i < BQ; ++i) oq[-i] = (uint8_t)(vq >> (8 * i));
                    } else {
                        const uint32_t nd = g.Cd - BD * gg, nq = g.Cq - BQ * gg;     // bytes left in the row from here up
#pragma unroll
                        for (int i = 0; i < BD; ++i)
                            if ((uint32_t)i < nd) od[-i] = (uint8_t)(vd >> (8 * i));
#pragma unroll
                        for (int i = 0; i < BQ; ++i)
                            if ((uint32_t)i < nq) oq[-i] = (uint8_t)(vq >> (8 * i));
                    }
                }
            }
            __syncthreads();                                                   // B3
            // ---- C: the two row images -> the tables
            store_image(gd, img_d, skew_d, Rc * g.Cd, tid);
            store_image(gq, img_q, skew_q, Rc * g.Cq, tid);
            if (qb + g.rmax < Rt) __syncthreads();                             // the next batch overwrites the images
        }
    }
    if (incomplete) ctl->incomplete = 1;
    if (mismatch) ctl->mismatch = 1;
    if (idx_over) ctl->index_overflow = 1;
    if (row_over) ctl->row_overflow = 1;
    if (STATS) {
        __syncthreads();
        hz.flush();
        acc.flush(st, 0);
    }
}

typedef void (*EncKernel)(const uint4*, EncGeom, LookBack, EncCtl*, uint64_t*, uint8_t*, uint8_t*, uq_stats*, uint32_t);

template <bool STATS>
EncKernel pick_enc(int bq, bool ntrick) {
#define UQ_EN(B) case B: return ntrick ? encode_tile_kernel<B, true, STATS> : encode_tile_kernel<B, false, STATS>;
    switch (bq) { UQ_EN(1) UQ_EN(2) UQ_EN(3) UQ_EN(4) UQ_EN(5) UQ_EN(6) UQ_EN(7) default: return ntrick ? encode_tile_kernel<8, true, STATS> : encode_tile_kernel<8, false, STATS>; }
#undef UQ_EN
}
}  // namespace

extern "C" int uq_encode_stream(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, const uq_pack_params* hp, uint64_t max_reads,
                                uint64_t* d_line_start, uint8_t* d_dna, uint8_t* d_qual, uq_stats* d_stats, uq_encode_result* h_out) {
    UQ_REQUIRE(ctx && hp && h_out, "uq_encode_stream: null argument");
    memset(h_out, 0, sizeof(*h_out));
    if (nbytes == 0) return 0;
    UQ_REQUIRE(d_buf && d_line_start && d_dna && d_qual, "uq_encode_stream: null buffer");
    const uint32_t bd = hp->bits_per_base, bq = hp->bits_per_quality;
    if (bd != 2 || bq < 1 || bq > 8 || max_reads == 0) return 0;
    const uint32_t Cd = hp->dna_bytes_per_row, Cq = hp->quality_bytes_per_row;
    const uint32_t Lv = hp->dna_max + (hp->variable ? 1 : 0);
    UQ_REQUIRE(Cd == (bd * Lv + 7) / 8 && Cq == (bq * Lv + 7) / 8,
               "uq_encode_stream: row bytes (%u, %u) do not match ceil(bits * (dna_max + variable) / 8)", Cd, Cq);
    // the lookup-free path only (pack.hip's FAST): bases "ACGT", qualities one contiguous ASCII range below 128, at most one
    // N-trick base whose quality code fits; anything else has no one-pass kernel (*h_out stays zero: nothing launched)
    bool fast = hp->dna_code['A'] == 0 && hp->dna_code['C'] == 1 && hp->dna_code['G'] == 2 && hp->dna_code['T'] == 3;
    int nbases = 0, nq = 0, qmin = 256, qmaxc = -1, ntrick_bases = 0, nchar = 0, max_q = 0, fill_q = -1;
    for (int i = 0; i < 256; ++i) {
        if (hp->dna_code[i] >= 0) ++nbases;
        if (hp->qual_code[i] >= 0) { ++nq; if (i < qmin) qmin = i; if (i > qmaxc) qmaxc = i; if (hp->qual_code[i] > max_q) max_q = hp->qual_code[i]; }
        if (hp->qual_code[i] == 0 && fill_q < 0) fill_q = i;
        if (hp->dna_code[i] < 0 && hp->n_qual[i] >= 0) { ++ntrick_bases; nchar = i; if (hp->n_qual[i] > max_q) max_q = hp->n_qual[i]; }
    }
    fast = fast && nbases == 4 && nq >= 1 && qmaxc - qmin + 1 == nq && qmaxc < 128 && ntrick_bases <= 1 && fill_q >= 0 &&
           (ntrick_bases == 0 || hp->n_qual[nchar] < 128) && max_q < (1 << bq);      // max_q >= 2^b: the Q9 carry, exact kernel only
    if (fast)
        for (int i = qmin; i <= qmaxc; ++i) fast = fast && hp->qual_code[i] == i - qmin;
    const uint32_t rec = (uint32_t)hp->max_record_bytes;
    if (!fast || rec < 4 || rec > EN_HALO - 16) return 0;           // a record that starts on a tile's last byte must end inside the halo

    EncGeom g;
    memset(&g, 0, sizeof(g));
    g.mis = (uint32_t)((uintptr_t)d_buf & 15);
    g.nbytes = nbytes; g.nvec = (nbytes + g.mis + 15) / 16; g.ntiles = (g.nvec + EN_TV - 1) / EN_TV; g.max_reads = max_reads;
    g.Cd = Cd; g.Cq = Cq; g.variable = hp->variable ? 1 : 0; g.G = (Lv + 7) / 8; g.dna_max = (uint32_t)hp->dna_max;
    g.fill_d = 0x01010101u * (uint32_t)'A'; g.fill_q = 0x01010101u * (uint32_t)fill_q;
    g.q_addlo = 0x01010101u * (uint32_t)(0x80 - qmin); g.q_addhi = 0x01010101u * (uint32_t)(0x80 - qmin - nq);
    if (ntrick_bases == 1) { g.n_char = 0x01010101u * (uint32_t)nchar; g.n_code = 0x01010101u * (uint32_t)hp->n_qual[nchar]; }
    // reads per batch: what a tile typically owns (+ slack); a tile that owns more packs them in several batches
    const uint32_t avg = hp->avg_record_bytes >= 4 && hp->avg_record_bytes < (int32_t)rec ? (uint32_t)hp->avg_record_bytes : rec;
    uint32_t rmax = EN_TILE / avg + 2 + (g.variable ? EN_TILE / avg / 8 : 0);
    const bool stats = d_stats != nullptr;
    const size_t fixed = 16 + EN_TILE + EN_HALO + 32 + EN_ECAP * 4 + 64 + (stats ? HZ_WORDS * 4 : 0);
    while (rmax > 1 && fixed + (size_t)rmax * (Cd + Cq) + 96 > 64 * 1024) --rmax;         // keep at least two workgroups per CU
    if (rmax > EN_THREADS) rmax = EN_THREADS;
    g.rmax = rmax;
    g.img_d = (rmax * Cd + 15 + 16) & ~15u; g.img_q = (rmax * Cq + 15 + 16) & ~15u;
    uint32_t P = EN_THREADS / rmax;
    if (P > g.G) P = g.G;
    if (P < 1) P = 1;
    g.P = P; g.magicP = magic_u32(P);
    if (const char* dbg = getenv("UQ_ENC_DEBUG")) g.debug = (uint32_t)atoi(dbg);
    const size_t lds = fixed + g.img_d + g.img_q;
    UQ_REQUIRE(lds <= 160 * 1024, "uq_encode_stream: tile needs %zu bytes of LDS", lds);

    EncKernel k = stats ? pick_enc<true>((int)bq, ntrick_bases == 1) : pick_enc<false>((int)bq, ntrick_bases == 1);
    if (lds > 48 * 1024) UQ_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // The look-back needs every workgroup of the grid resident at once (a waiting workgroup must be able to count on the
    // ones in front of it running): grid = what the runtime says fits, bounded by the LDS carve -- and every spin is bounded.
    int per_cu = 0;
    UQ_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k, EN_THREADS, lds));
    const int by_lds = (int)((160 * 1024) / (lds + 256));
    if (per_cu > by_lds) per_cu = by_lds;
    if (per_cu > (stats ? 3 : 4)) per_cu = stats ? 3 : 4;
    if (per_cu < 1) return 0;
    const uint64_t blocks = g.ntiles < (uint64_t)UQ_NUM_CU * per_cu ? g.ntiles : (uint64_t)UQ_NUM_CU * per_cu;

    void* ws;
    const uint64_t ngroups = (g.ntiles + EN_GROUP - 1) / EN_GROUP;
    const size_t ws_bytes = 256 + (((g.ntiles + 15) & ~uint64_t(15)) + ngroups * EN_GSTRIDE) * 8;
    UQ_TRY(uq_scratch(ctx, ws_bytes, &ws));
    EncCtl* d_ctl = (EncCtl*)ws;
    LookBack lb;
    lb.A = (unsigned long long*)((uint8_t*)ws + 256); lb.GR = lb.A + ((g.ntiles + 15) & ~uint64_t(15));
    UQ_CHECK_HIP(hipMemsetAsync(ws, 0, ws_bytes, ctx->stream));
    const uint32_t qbase = qmin >= 64 ? 59u : (qmin < 33 ? 0u : 33u);
    k<<<(uint32_t)blocks, EN_THREADS, lds, ctx->stream>>>((const uint4*)(d_buf - g.mis), g, lb, d_ctl, d_line_start, d_dna, d_qual, d_stats,
                                                         (64u << 8) | qbase);
    UQ_LAUNCH_CHECK();
    UQ_CHECK_HIP(hipMemcpyAsync(ctx->h_pinned, d_ctl, sizeof(EncCtl), hipMemcpyDeviceToHost, ctx->stream));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    const EncCtl c = *(const EncCtl*)ctx->h_pinned;
    h_out->launched = 1;
    h_out->nlines = c.nlines;
    h_out->index_ok = !c.abort && !c.index_overflow && c.nlines <= 4 * max_reads;
    const bool whole = h_out->index_ok && !c.row_overflow && c.nlines % 4 == 0;
    h_out->stats_ok = stats && whole && !c.incomplete;
    h_out->tables_ok = whole && !c.incomplete && !c.mismatch;
    return 0;
}
