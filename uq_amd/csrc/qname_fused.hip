// qname_fused.hip -- what surrounds the QNAME phase of the pack kernel (pack.hip, `qname_tile`): the layout GUESS made on the
// device in front of it and the distinct-value counts behind it (SURVEY.md 8 row f1 fused into rows a3 / a4).
//
// The reference infers prefix, suffix and separators in one sequential loop over all QNAME lines (uq.py:394-444), splits every line
// at the separators (uq.py:555-565) and int()s the fields (uq.py:717-736).  qname_dev.hip does that as two traversals of the QNAME
// lines (layout reductions, then the tokeniser) with a host decision in between.  Here the layout is guessed from a SAMPLE:
//   qname_sample_kernel   builds line 1's character table in every workgroup and runs qname_dev.hip's layout reductions (min lcp /
//                         lcs with line 1, entry / last violation per character of line 1) over a stratified sample of the reads;
//   qname_guess_kernel    one lane: uq.py:428-444 on those numbers -> prefix / suffix lengths, ordered separators -> uq_qname_fused;
// the pack kernel then verifies the guess on EVERY read while it tokenises (so nothing is assumed: see include/uqhip.h for why a
// clean pass proves the layout is the reference's), and
//   qf_first_seen_kernel / qf_count_kernel   count, per column of uint32 values, the distinct values among reads [0, T] at the
//                         checkpoints of uq.py:586-602 (first occurrence per value by atomic minima; no sort).
// Nothing here waits for the host: the read count is taken from the queued census (ctx->d_async) or from the structure itself.
#include "common.h"
#include "lines.h"

namespace {
constexpr int QF_THREADS = 256;
constexpr int QF_MAXCH = 64;
constexpr uint32_t QF_SAMPLES = 4096;
constexpr uint32_t QF_ROW = 64, QF_STRIDE = QF_ROW + 4;     // per-lane staging row (as qname_dev.hip)
constexpr uint32_t QF_FT = 1u << 20;                        // widest value range whose distinct values are counted here (a bitmap in LDS)
constexpr uint32_t QF_SMALL = 4096;                         // ranges counted over the whole column, through private LDS tables
constexpr size_t QF_PINNED_AT = 5200;                        // uq_qname_fused's place in ctx->h_pinned (uint64 index; common.h lists the others)
static_assert(sizeof(uq_qname_fused) % 4 == 0 && QF_PINNED_AT * 8 + sizeof(uq_qname_fused) <= 8000 * 8, "the structure's slot in the pinned staging");
constexpr uint64_t QF_PREFIX = 1ull << 21;                  // checkpoints a wide-range column is judged on (qname_device.INT_PREFIX)

struct DevLine1 {
    uint8_t text[256];
    uint8_t slot[256];              // character -> candidate slot, 0xFF = not in line 1
    uint8_t ch[QF_MAXCH];
    uint16_t cnt[QF_MAXCH];         // occurrences in line 1
    uint16_t lastpos[QF_MAXCH];     // last position in line 1
    uint32_t len, nch, bad, pad;    // bad: no usable line 1 (empty buffer, not '@...', longer than 255 bytes, more than 64 distinct characters)
};

__device__ __forceinline__ void stage_line(uint8_t* row, const uint8_t* q, uint32_t ql, const uint8_t* buf_end) {
    for (uint32_t c = 0; c < ql; c += 16) {
        if (q + c + 16 <= buf_end) {
            uint4 v;
            __builtin_memcpy(&v, q + c, 16);
            uint32_t* d = (uint32_t*)(row + c);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        } else {
            for (uint32_t b = c; b < ql; ++b) row[b] = q[b];
        }
    }
}

// n: reads in the buffer (d_async: taken from the queued census).  `lay` was initialised by the host side of the call.
// One wave per workgroup, QF_SPW samples per wave: the work per sample is a chain of dependent wave-wide operations, so the kernel's
// time is one wave's -- 4096 samples over 256 CUs as 16 per wave, not over 16 CUs as 256 per workgroup.
constexpr uint32_t QF_SPW = 8;
__global__ __launch_bounds__(64) void qname_sample_kernel(const uint8_t* __restrict__ buf, const uint64_t* __restrict__ ls, uint64_t n,
                                                                   const unsigned long long* __restrict__ d_async, uq_qname_layout_result* __restrict__ lay,
                                                                   DevLine1* __restrict__ l1_out, uint32_t* __restrict__ step_out, CensusView cv, uint64_t nbytes,
                                                                   const uint32_t* __restrict__ over) {
    __shared__ __align__(16) uint8_t stage[64 * QF_STRIDE];
    __shared__ DevLine1 l1;
    __shared__ unsigned long long s_entry[QF_MAXCH], s_viol[QF_MAXCH];
    __shared__ uint32_t s_lcp, s_lcs, s_flags;
    const uint32_t tid = threadIdx.x;
    // cv.list: no expanded index (`ls` is null) -- the line starts come from the lists of the census queued in front (lines.h), and the
    // sample is stratified by POSITION: the first record that starts in a census tile picked from every stratum of tiles
    const bool lists = cv.list != nullptr;
    uint64_t nlines = 4 * n;
    if (d_async) { nlines = d_async[0]; n = (d_async[1] || (lists && *over)) ? 0 : d_async[0] / 4; }
    // ---- line 1 and its character table (every workgroup builds its own copy: 255 bytes, one lane)
    uint32_t len = 0;
    if (n) { bool ok; const uint64_t e = lists ? cv_line_start(cv, nlines, 0, 0, 1, ok) : ls[1]; len = e >= 1 && e <= 256 ? (uint32_t)(e - 1) : 0xFFFFFFFFu; }
    for (uint32_t x = tid; x < 256; x += 64) { l1.text[x] = (len != 0xFFFFFFFFu && x < len) ? buf[x] : 0; l1.slot[x] = 0xFF; }
    if (tid < QF_MAXCH) { l1.ch[tid] = 0; l1.cnt[tid] = 0; l1.lastpos[tid] = 0; s_entry[tid] = UQ_NONE; s_viol[tid] = 0; }
    if (tid == 0) { s_lcp = 0xFFFFFFFFu; s_lcs = 0xFFFFFFFFu; s_flags = 0; }
    __syncthreads();
    const bool bad0 = len == 0 || len == 0xFFFFFFFFu || l1.text[0] != '@';
    if (!bad0 && len <= 64) {
        // the character table of a line of at most 64 bytes, a lane per position: lane p learns the set of positions that hold its
        // character (one ballot per position), from which its count, its last position and -- for a first occurrence -- its slot follow
        const uint32_t lane0 = lane_id();
        const uint32_t mych = lane0 < len ? (uint32_t)l1.text[lane0] : 0x100u;
        unsigned long long same = 0;
        for (uint32_t j = 0; j < len; ++j) {
            const unsigned long long bm = __ballot(mych == (uint32_t)__builtin_amdgcn_readlane((int)mych, (int)j));
            if (lane0 == j) same = bm;
        }
        const bool first = lane0 < len && (uint32_t)__builtin_ctzll(same | (1ull << 63)) == lane0;
        const unsigned long long fm = __ballot(first);
        if (first) {
            const uint32_t sl = (uint32_t)__popcll(fm & ((1ull << lane0) - 1));
            l1.slot[mych] = (uint8_t)sl; l1.ch[sl] = (uint8_t)mych;
            l1.cnt[sl] = (uint16_t)__popcll(same); l1.lastpos[sl] = (uint16_t)(63 - __builtin_clzll(same));
        }
        if (lane0 == 0) { l1.len = len; l1.nch = (uint32_t)__popcll(fm); l1.bad = 0; l1.pad = 0; }
    } else if (tid == 0) {
        uint32_t nch = 0, bad = bad0 ? 1u : 0u;
        if (!bad)
            for (uint32_t p = 0; p < len; ++p) {
                const uint8_t c = l1.text[p];
                if (l1.slot[c] == 0xFF) {
                    if (nch == QF_MAXCH) { bad = 1; break; }
                    l1.slot[c] = (uint8_t)nch; l1.ch[nch] = c; ++nch;
                }
                l1.cnt[l1.slot[c]] += 1; l1.lastpos[l1.slot[c]] = (uint16_t)p;
            }
        l1.len = bad ? 0 : len; l1.nch = nch; l1.bad = bad; l1.pad = 0;
    }
    __syncthreads();
    const uint32_t step = (uint32_t)(n / QF_SAMPLES > 1 ? n / QF_SAMPLES : 1);
    if (blockIdx.x == 0) {
        for (uint32_t i = tid; i < sizeof(DevLine1) / 4; i += 64) ((uint32_t*)l1_out)[i] = ((const uint32_t*)&l1)[i];
        if (tid == 0) *step_out = step;
    }
    if (l1.bad) return;
    const uint8_t* buf_end = lists ? buf + nbytes : buf + ls[4 * n];
    const uint64_t ks = lists ? (uint64_t)QF_SAMPLES : 0;           // sample k: a record out of a census tile of the k-th of ks strata of the stream
    uint32_t my_lcp = 0xFFFFFFFFu, my_lcs = 0xFFFFFFFFu;         // (wave-uniform below: every lane holds the same value)
    const uint32_t lane = lane_id();
    const uint32_t my_ch = lane < l1.nch ? l1.ch[lane] : 0x100u;   // lane c speaks for candidate character c of line 1
    const uint32_t my_cnt = lane < l1.nch ? l1.cnt[lane] : 0u, my_last = lane < l1.nch ? l1.lastpos[lane] : 0u;
    // sample = one record out of every `step` consecutive ones, at a pseudo-random place in its stratum (a fixed stride would
    // alias with periodic names: interleaved mates, lanes that cycle); record 0 is line 1 itself.  Two steps per round of 64 samples
    // a wave: (1) every lane fetches ITS sample's line into its LDS row (64 index and line loads in flight at once); (2) the wave goes
    // through the 64 lines one by one with a lane per BYTE: common prefix / suffix with line 1 and the per-character counts are a
    // few ballots each, no loop over the bytes (a lane per sample walking its line byte by byte took 72 us for 4096 samples).
    for (uint64_t k0 = (uint64_t)blockIdx.x * QF_SPW; lists ? k0 < ks : k0 * step < n; k0 += (uint64_t)gridDim.x * QF_SPW) {
        const uint64_t k = k0 + lane;
        uint64_t h = (k + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
        h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
        uint64_t i = lists ? 0 : k * step + h % step;
        uint64_t q0 = 0, q1 = 0;
        if (lists) {
            if (lane < QF_SPW && k < ks) {
                const uint64_t T = (((k << 16) | (h & 0xFFFFu)) * cv.nb) / (ks << 16);
                const uint64_t o = cv.offs[T], c = (T + 1 < cv.nb ? (uint64_t)cv.offs[T + 1] : nlines) - o;
                uint32_t slot = (3u - (uint32_t)o) & 3u;                // the newlines of the tile that close a record: rank = 3 mod 4
                if (c > slot) slot += 4u * (uint32_t)((h >> 16) % ((c - slot + 3) / 4));       // ... one of them (a tile without any: the next one behind it)
                const uint64_t rank = o + slot;
                bool ok0, ok1;
                q0 = cv_line_start(cv, nlines, (uint32_t)T, slot + 1, 0, ok0); q1 = cv_line_start(cv, nlines, (uint32_t)T, slot + 1, 1, ok1);
                if (ok0 && ok1) i = (rank + 1) / 4;
            }
        } else if (lane < QF_SPW && i && i < n) { q0 = ls[4 * i]; q1 = ls[4 * i + 1]; }
        if (lane >= QF_SPW || i >= n) i = 0;                            // (0 = no sample in this lane)
        const uint8_t* q = buf;
        uint32_t ql = 0;
        uint8_t* row = stage + tid * QF_STRIDE;
        if (i) {
            q = buf + q0;
            ql = (uint32_t)(q1 - q0 - 1);
            if (ql <= QF_ROW) stage_line(row, q, ql, buf_end);
        }
        for (uint32_t sidx = 0; sidx < QF_SPW; ++sidx) {
            const uint64_t si = __shfl(i, sidx, 64);
            if (si == 0) continue;                                      // (uniform)
            const uint32_t sql = (uint32_t)__shfl(ql, sidx, 64);
            if (sql > 255) { if (lane == 0) atomicOr(&s_flags, 2u); continue; }
            const uint8_t* sq = (const uint8_t*)__shfl((unsigned long long)(uintptr_t)q, sidx, 64);
            const uint8_t* srow = stage + sidx * QF_STRIDE;
            const uint32_t m = sql < l1.len ? sql : l1.len;
            uint32_t lcp = m, last_bad = 0xFFFFFFFFu, have = 0;         // last_bad: highest position that differs from line 1 counted from the ends
            bool lcp_open = true;
            for (uint32_t base = 0; base < sql; base += 64) {
                const uint32_t pos = base + lane;
                const bool valid = pos < sql;
                const uint32_t b = valid ? (sql <= QF_ROW ? srow[pos] : sq[pos]) : 0x100u;
                const unsigned long long pm = __ballot(valid && pos < m && b != l1.text[pos]);
                if (lcp_open && pm) { lcp = base + (uint32_t)__builtin_ctzll(pm); lcp_open = false; }
                const uint32_t t = sql - 1 - pos;                       // distance from the line's end
                const unsigned long long sm = __ballot(valid && t < m && b != l1.text[l1.len - 1 - t]);
                if (sm) last_bad = base + 63u - (uint32_t)__builtin_clzll(sm);
                for (uint32_t c = 0; c < l1.nch; ++c) {
                    const unsigned long long cm = __ballot(b == (uint32_t)l1.ch[c]);
                    if (lane == c) have += (uint32_t)__popcll(cm);
                }
            }
            const uint32_t lcs = last_bad == 0xFFFFFFFFu ? m : sql - 1 - last_bad;
            if (lane == 0 && ((lcp == sql && sql < l1.len) || (lcs == sql && sql < l1.len))) atomicOr(&s_flags, 1u);
            if (lcp < my_lcp) my_lcp = lcp;
            if (lcs < my_lcs) my_lcs = lcs;
            if (lane < l1.nch) {
                if (lcp <= my_last && (unsigned long long)si < s_entry[lane]) atomicMin(&s_entry[lane], (unsigned long long)si);
                if (have != my_cnt && (unsigned long long)si > s_viol[lane]) atomicMax(&s_viol[lane], (unsigned long long)si);
            }
        }
    }
    (void)my_ch;
    if (lane_id() == 0) { atomicMin(&s_lcp, my_lcp); atomicMin(&s_lcs, my_lcs); }
    __syncthreads();
    if (tid < l1.nch) {
        if (s_entry[tid] != UQ_NONE) atomicMin((unsigned long long*)&lay->entry[tid], s_entry[tid]);
        if (s_viol[tid]) atomicMax((unsigned long long*)&lay->lastviol[tid], s_viol[tid]);
    }
    if (tid == 0) {
        atomicMin(&lay->min_lcp, s_lcp); atomicMin(&lay->min_lcs, s_lcs);
        if (s_flags) atomicOr(&lay->flags, s_flags);
    }
}

__global__ void layout_init_kernel(uq_qname_layout_result* lay) {
    const uint32_t t = threadIdx.x;
    if (t < QF_MAXCH) { lay->entry[t] = UQ_NONE; lay->lastviol[t] = 0; lay->ch[t] = 0; }
    if (t == 0) { lay->min_lcp = 0xFFFFFFFFu; lay->min_lcs = 0xFFFFFFFFu; lay->flags = 0; lay->nch = 0; }
}

__device__ bool regex_special(uint8_t c) {
    const char* s = ".^$*+?{}[]\\|()-";
    for (int i = 0; s[i]; ++i) if ((uint8_t)s[i] == c) return true;
    return false;
}

// uq.py:428-444 on the sample's reductions -> the guess.  One wave.
__global__ void qname_guess_kernel(const DevLine1* __restrict__ g_l1, const uq_qname_layout_result* __restrict__ g_lay, const uint32_t* __restrict__ step,
                                   uq_qname_fused* __restrict__ q) {
    // the inputs are copied to LDS first: lane 0's loops below are serial, and a dependent global load costs ten LDS reads
    __shared__ DevLine1 s_l1;
    __shared__ uq_qname_layout_result s_lay;
    for (uint32_t i = threadIdx.x; i < sizeof(DevLine1) / 4; i += 64) ((uint32_t*)&s_l1)[i] = ((const uint32_t*)g_l1)[i];
    for (uint32_t i = threadIdx.x; i < sizeof(uq_qname_layout_result) / 4; i += 64) ((uint32_t*)&s_lay)[i] = ((const uint32_t*)g_lay)[i];
    __syncthreads();
    const DevLine1* l1 = &s_l1;
    const uq_qname_layout_result* lay = &s_lay;
    const uint32_t t = threadIdx.x;                           // one wave
    for (uint32_t i = t; i < 256; i += 64) { q->line1[i] = i < l1->len ? l1->text[i] : 0; q->inset[i] = 0; }
    if (t < 32) q->seps[t] = 0;
    if (t < UQ_QF_MAXC) { q->vmin[t] = 0xFFFFFFFFu; q->vmax[t] = 0; q->undetermined[t] = 0; }
    for (uint32_t i = t; i < UQ_QF_MAXC * UQ_QF_MAXT; i += 64) (&q->counts[0][0])[i] = 0;
    if (t < UQ_QF_MAXT) q->thresholds[t] = 0;
    if (t == 0) { q->ok = 0; q->plen = q->slen = q->nsep = q->l1len = 0; q->flags = 0; q->sample_step = *step; q->nth = 0; q->nreads = 0; }
    if (l1->bad || lay->flags) return;                        // (uniform)
    const uint32_t len = l1->len;
    // min_lcp / min_lcs start at line 1's length: with no sampled read (a file of one record) plen + slen > len declines below
    const uint32_t plen = lay->min_lcp < len ? lay->min_lcp : len, slen = lay->min_lcs < len ? lay->min_lcs : len;
    if (plen == 0 || plen + slen > len) return;               // every QNAME starts with '@': an empty prefix is no FASTQ the fused pass takes
    // lane k judges character k of line 1 (uq.py:428-431): it survived the loop and occurs in the middle of line 1
    __shared__ uint8_t s_sep[QF_MAXCH];
    bool mine = false, special = false;
    if (t < l1->nch && lay->entry[t] != UQ_NONE && lay->lastviol[t] < lay->entry[t]) {
        const uint8_t c = l1->ch[t];
        int mid = 0, ord = 0;                                 // l1[plen:].count(c) - suffix.count(c); its occurrences in the slice the ORDER is read from
        for (uint32_t p = plen; p < len - slen; ++p) { mid += l1->text[p] == c; ord += (p + 1 < len - slen) && l1->text[p] == c; }
        mine = mid != 0;
        special = mine && regex_special(c);                   // '[seps]+' and '(.*)'.join(seps) are regexes in the reference: the host's `re` path
        // The pack kernel holds every read to the ORDERED separators, which the reference reads from l1[plen : len - 1 - slen] (Q14: one
        // character short) -- while its per-read rule counts each separator over the whole middle.  A separator whose occurrences the slice
        // does not all show (it is line 1's last character before the suffix: `@q_0=1`, `@q_1=4` make '1' one) is in the reference's set but
        // not in the order: what the reference then does depends on the LAST read (uq.py:438-444), which no per-read check reproduces.  A
        // digit as separator is left to the exact path as well (the fields are decimal numbers here).
        if (mine && (ord != mid || (c >= '0' && c <= '9'))) special = true;
    }
    s_sep[t] = mine ? 1 : 0;
    const uint32_t nsepch = (uint32_t)__popcll(__ballot(mine));
    if (__ballot(special) || nsepch == 0 || nsepch > 4) return;      // (no separator: the reference refuses such files, the exact path words the error;
                                                                     //  more than four: the pack kernel tests a window against four characters)
    __syncthreads();
    if (t != 0) return;
    // separators = the separator characters of l1[plen : len - 1 - slen] in order (Q14: the slice drops one more character)
    uint32_t nsep = 0;
    const uint32_t end = len - slen >= 1 ? len - slen - 1 : 0;
    for (uint32_t p = plen; p < end; ++p) {
        const uint8_t c = l1->text[p];
        if (l1->slot[c] != 0xFF && s_sep[l1->slot[c]]) {
            if (nsep == UQ_QF_MAXC - 1) return;
            q->seps[nsep++] = c;
        }
    }
    if (nsep == 0) return;
    __builtin_amdgcn_s_waitcnt(0);                            // (the other lanes' zeroes of inset[] are stores of this same wave: in order)
    for (uint32_t k = 0; k < l1->nch; ++k) if (s_sep[k]) q->inset[l1->ch[k]] = 1;
    q->plen = plen; q->slen = slen; q->nsep = nsep; q->l1len = len;
    q->ok = 1;
}

// ---- distinct values per checkpoint, queued behind the pack kernel
struct Checkpoints { uint64_t t[UQ_QF_MAXT]; uint32_t nth, nhead; };
__device__ __forceinline__ Checkpoints checkpoints(uint64_t n) {
    Checkpoints c; c.nth = 0; c.nhead = 0;
    if (n == 0) return c;
    for (uint64_t t = 10000; t <= n - 1 && c.nth < UQ_QF_MAXT - 1; t *= 2) c.t[c.nth++] = t;
    if (c.nth == 0 || c.t[c.nth - 1] != n - 1) c.t[c.nth++] = n - 1;
    for (uint32_t k = 0; k < c.nth; ++k) if (c.t[k] < QF_PREFIX) c.nhead = k + 1;
    return c;
}

// first[col][v - vmin] = lowest read holding value v, for columns whose range is at most QF_SMALL: over all reads, through a private
// LDS table per workgroup (a column of four lanes would otherwise be ten million atomics on four addresses).
__global__ __launch_bounds__(QF_THREADS) void qf_first_seen_kernel(uq_qname_fused* __restrict__ q, const uint32_t* __restrict__ vals, uint64_t pitch,
                                                                    uint32_t* __restrict__ first) {
    __shared__ uint32_t s_first[QF_SMALL];
    const uint32_t col = blockIdx.y;
    if (!q->ok || q->flags || col > q->nsep) return;
    const uint64_t n = q->nreads;
    const uint32_t vmin = q->vmin[col], vmax = q->vmax[col];
    if (n == 0 || vmin > vmax) return;
    const uint64_t range = (uint64_t)vmax - vmin + 1;
    if (range > QF_SMALL) return;                                       // qf_wide_kernel's
    const uint32_t* v = vals + col * pitch;
    uint32_t* f = first + (size_t)col * QF_SMALL;
    for (uint32_t i = threadIdx.x; i < (uint32_t)range; i += QF_THREADS) s_first[i] = 0xFFFFFFFFu;
    __syncthreads();
    // a workgroup takes a CONTIGUOUS slice of the reads (the first lane to see a value in a slice usually settles it), a lane four
    // consecutive values per step, two steps requested before the first is looked at
    const uint64_t per = (((n + gridDim.x - 1) / gridDim.x) + 3) & ~uint64_t(3);
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    auto see = [&](uint32_t val, uint64_t i) {
        const uint32_t slot = val - vmin;
        if (slot < range && (uint32_t)i < s_first[slot]) atomicMin(&s_first[slot], (uint32_t)i);
    };
    const bool vec = (((uintptr_t)v) & 15) == 0;             // column c starts at c * pitch values: 16-byte aligned when the pitch is a multiple of 4
    uint64_t i = lo + 4 * (uint64_t)threadIdx.x;
    if (vec) {
        for (; i + 4 * QF_THREADS + 4 <= hi; i += 8 * QF_THREADS) {
            const uint4 a = *(const uint4*)(v + i), b = *(const uint4*)(v + i + 4 * QF_THREADS);
            see(a.x, i); see(a.y, i + 1); see(a.z, i + 2); see(a.w, i + 3);
            const uint64_t j = i + 4 * QF_THREADS;
            see(b.x, j); see(b.y, j + 1); see(b.z, j + 2); see(b.w, j + 3);
        }
    }
    for (; i < hi; i += 4 * QF_THREADS)
        for (uint32_t k = 0; k < 4 && i + k < hi; ++k) see(v[i + k], i + k);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (uint32_t)range; i += QF_THREADS)
        if (s_first[i] != 0xFFFFFFFFu && s_first[i] < f[i]) atomicMin(&f[i], s_first[i]);
}

// Wide ranges (QF_SMALL < range <= QF_FT): the rule `distinct among reads [0, T] > T / 10` is evaluated checkpoint by checkpoint,
// T = 10 000, 20 000, 40 000 (those below 2^21 among the first QF_WIDE_CP), by ONE workgroup with the value set as a bitmap in LDS:
// flow-cell coordinates fire at the first one.  counts[col][k] is written up to the checkpoint that fires (the host needs no more:
// the column becomes `integers`), or for all of them when that is every checkpoint of the file; otherwise undetermined[col] = 1 and
// the host sorts the column.  (A first-occurrence table over the first two million reads, as uq_int_prefix_distinct keeps it, is
// two million contended global atomics: 0.1 ms for a decision the first 10 001 reads already give.)
constexpr uint32_t QF_WIDE_CP = 3;
__global__ __launch_bounds__(QF_THREADS) void qf_wide_kernel(uq_qname_fused* __restrict__ q, const uint32_t* __restrict__ vals, uint64_t pitch) {
    extern __shared__ uint32_t bits[];                                  // QF_FT / 32 words
    __shared__ uint32_t s_count;
    const uint32_t col = blockIdx.x;
    if (!q->ok || q->flags || col > q->nsep) return;
    const uint64_t n = q->nreads;
    const uint32_t vmin = q->vmin[col], vmax = q->vmax[col];
    if (n == 0 || vmin > vmax) return;
    const uint64_t range = (uint64_t)vmax - vmin + 1;
    if (range <= QF_SMALL) return;
    if (range > QF_FT) { if (threadIdx.x == 0) q->undetermined[col] = 1; return; }
    const Checkpoints cp = checkpoints(n);
    const uint32_t* v = vals + col * pitch;
    for (uint32_t i = threadIdx.x; i < (uint32_t)((range + 31) / 32); i += QF_THREADS) bits[i] = 0;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    uint64_t done = 0;                                                  // reads [0, done) are in the set
    bool fired = false;
    uint32_t k = 0;
    for (; k < cp.nth && k < QF_WIDE_CP && cp.t[k] < QF_PREFIX; ++k) {
        uint32_t fresh = 0;
        auto add = [&](uint32_t val) {
            const uint32_t slot = val - vmin;
            if (slot >= range) return;
            const uint32_t bit = 1u << (slot & 31);
            if (!(atomicOr(&bits[slot >> 5], bit) & bit)) ++fresh;
        };
        uint64_t i = done + threadIdx.x;
        for (; i + 7 * QF_THREADS <= cp.t[k]; i += 8 * QF_THREADS) {       // eight loads in flight, then their eight updates
            uint32_t x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = v[i + (uint64_t)u * QF_THREADS];
#pragma unroll
            for (int u = 0; u < 8; ++u) add(x[u]);
        }
        for (; i <= cp.t[k]; i += QF_THREADS) add(v[i]);
        done = cp.t[k] + 1;
        fresh = wave_sum(fresh);
        if (lane_id() == 0 && fresh) atomicAdd(&s_count, fresh);
        __syncthreads();
        const uint32_t cnt = s_count;
        if (threadIdx.x == 0) q->counts[col][k] = cnt;
        if (cnt > cp.t[k] / 10) { fired = true; break; }
        __syncthreads();
    }
    if (!fired && k < cp.nth && threadIdx.x == 0) q->undetermined[col] = 1;      // checkpoints left that were not evaluated
}

__global__ __launch_bounds__(QF_THREADS) void qf_count_kernel(uq_qname_fused* __restrict__ q, const uint32_t* __restrict__ first) {
    __shared__ uint32_t s_cnt[UQ_QF_MAXT];
    const uint32_t col = blockIdx.y;
    if (!q->ok || q->flags || col > q->nsep) return;
    const uint64_t n = q->nreads;
    const Checkpoints cp = checkpoints(n);
    if (col == 0 && blockIdx.x == 0 && threadIdx.x == 0) { q->nth = cp.nth; for (uint32_t k = 0; k < cp.nth; ++k) q->thresholds[k] = cp.t[k]; }
    const uint32_t vmin = q->vmin[col], vmax = q->vmax[col];
    if (n == 0 || vmin > vmax) return;
    const uint64_t range = (uint64_t)vmax - vmin + 1;
    if (range > QF_SMALL) return;
    const uint32_t nth = cp.nth;
    if (threadIdx.x < UQ_QF_MAXT) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t* f = first + (size_t)col * QF_SMALL;
    for (uint64_t j = (uint64_t)blockIdx.x * QF_THREADS + threadIdx.x; j < range; j += (uint64_t)gridDim.x * QF_THREADS) {
        const uint32_t at = f[j];
        if (at == 0xFFFFFFFFu) continue;
        for (uint32_t k = 0; k < nth; ++k)
            if (at <= cp.t[k]) atomicAdd(&s_cnt[k], 1u);
    }
    __syncthreads();
    if (threadIdx.x < nth && s_cnt[threadIdx.x]) atomicAdd((unsigned long long*)&q->counts[col][threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

template <typename T>
__global__ void encode_u32_kernel(const uint32_t* __restrict__ val, uint64_t n, uint32_t sub, T* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * QF_THREADS + threadIdx.x;
    if (i < n) out[i] = (T)(val[i] - sub);
}

int guess_impl(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t nreads, const unsigned long long* d_async, uq_qname_fused* d_q) {
    // d_line_start == nullptr (the queued form only): the census queued in front serves as the index
    const bool use_lists = d_line_start == nullptr && d_async != nullptr;
    UQ_REQUIRE(ctx && d_buf && (d_line_start || use_lists) && d_q, "uq_qname_guess: null argument");
    UQ_REQUIRE(!use_lists || ctx->async_nbytes > 0, "uq_qname_guess_async: no line index given and no census of this buffer queued in front");
    CensusView cv;
    memset(&cv, 0, sizeof(cv));
    const uint32_t* over = nullptr;
    if (use_lists) {
        cv.mis = (uint32_t)((uintptr_t)d_buf & 15);
        cv.list = ctx->idx_bitmap; cv.offs = ctx->idx_partials;
        cv.nb = (((ctx->async_nbytes + cv.mis + 15) / 16) * 16 + CV_TILE - 1) / CV_TILE;
        over = (const uint32_t*)(ctx->idx_bitmap + cv.nb * CV_LIST_CAP);
    }
    ScratchPlan sp;
    const size_t o_lay = sp.add(sizeof(uq_qname_layout_result)), o_l1 = sp.add(sizeof(DevLine1)), o_step = sp.add(16);
    void* scr;
    UQ_TRY(uq_scratch(ctx, sp.off, &scr));
    uint8_t* base = (uint8_t*)scr;
    uq_qname_layout_result* d_lay = (uq_qname_layout_result*)(base + o_lay);
    // initial values of the reductions: min_lcp = min_lcs = 0xFFFFFFFF (clamped to line 1's length by the guess kernel), entry = none
    layout_init_kernel<<<1, 256, 0, ctx->stream>>>(d_lay);             // (one launch instead of three fills)
    UQ_LAUNCH_CHECK();
    const uint32_t grid = QF_SAMPLES / QF_SPW;
    qname_sample_kernel<<<grid, 64, 0, ctx->stream>>>(d_buf, d_line_start, nreads, d_async, d_lay, (DevLine1*)(base + o_l1), (uint32_t*)(base + o_step), cv,
                                                      use_lists ? ctx->async_nbytes : 0, over);
    UQ_LAUNCH_CHECK();
    qname_guess_kernel<<<1, 64, 0, ctx->stream>>>((const DevLine1*)(base + o_l1), d_lay, (const uint32_t*)(base + o_step), d_q);
    UQ_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int uq_qname_guess(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t nreads, uq_qname_fused* d_q) {
    return guess_impl(ctx, d_buf, d_line_start, nreads, nullptr, d_q);
}

extern "C" int uq_qname_guess_async(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uq_qname_fused* d_q) {
    UQ_REQUIRE(ctx && ctx->async_buf == d_buf, "uq_qname_guess_async: not the buffer of the last uq_count_lines_end_async");
    return guess_impl(ctx, d_buf, d_line_start, 0, ctx->d_async, d_q);
}

extern "C" int uq_qname_fused_finish(uq_ctx* ctx, uq_qname_fused* d_q, const uint32_t* d_vals, uint64_t vals_pitch) {
    UQ_REQUIRE(ctx && d_q && d_vals, "uq_qname_fused_finish: null argument");
    void* scr;
    const size_t bytes = (size_t)UQ_QF_MAXC * QF_SMALL * sizeof(uint32_t);
    UQ_TRY(uq_scratch(ctx, bytes, &scr));
    UQ_CHECK_HIP(hipMemsetAsync(scr, 0xFF, bytes, ctx->stream));
    qf_first_seen_kernel<<<dim3(UQ_NUM_CU * 4, UQ_QF_MAXC), QF_THREADS, 0, ctx->stream>>>(d_q, d_vals, vals_pitch, (uint32_t*)scr);
    UQ_LAUNCH_CHECK();
    qf_count_kernel<<<dim3(4, UQ_QF_MAXC), QF_THREADS, 0, ctx->stream>>>(d_q, (const uint32_t*)scr);
    UQ_LAUNCH_CHECK();
    if (!ctx->qf_attr_set) { UQ_CHECK_HIP(hipFuncSetAttribute((const void*)qf_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(QF_FT / 8))); ctx->qf_attr_set = true; }
    qf_wide_kernel<<<UQ_QF_MAXC, QF_THREADS, QF_FT / 8, ctx->stream>>>(d_q, d_vals, vals_pitch);
    UQ_LAUNCH_CHECK();
    // the structure is sent on its way to the host right away (a region of the pinned staging that nothing else uses):
    // uq_qname_fused_fetch then only waits -- a read-back queued by the fetch itself would start a host round trip later
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned + QF_PINNED_AT, d_q, sizeof(uq_qname_fused)));
    ctx->qf_sent = d_q;
    return 0;
}

// ---- the same first occurrences over SHARDS (one rank per GPU): the read numbers are file-wide (read_offset + local number) and the value
// ranges are the whole file's (the caller has combined vmin / vmax over the ranks), so that a MIN over the ranks' tables is the file's
// table and `distinct among reads [0, T]` follows for every checkpoint T of the file.  int64 entries (a collective's MIN takes them as they
// are), INT64_MAX = the value does not occur in this shard.
struct QfRanges { uint32_t vmin[UQ_QF_MAXC]; uint32_t range[UQ_QF_MAXC]; };          // range 0: not a small-range column, skipped
namespace {
__global__ void qf_fill_i64_kernel(long long* __restrict__ p, uint64_t n, long long v) {
    const uint64_t i = (uint64_t)blockIdx.x * QF_THREADS + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ __launch_bounds__(QF_THREADS) void qf_first_seen_global_kernel(const uint32_t* __restrict__ vals, uint64_t pitch, uint64_t n, uint64_t read_offset,
                                                                           QfRanges rg, long long* __restrict__ first) {
    __shared__ uint32_t s_first[QF_SMALL];
    const uint32_t col = blockIdx.y;
    const uint32_t vmin = rg.vmin[col], range = rg.range[col];
    if (range == 0 || range > QF_SMALL || n == 0) return;
    const uint32_t* v = vals + col * pitch;
    long long* f = first + (size_t)col * QF_SMALL;
    for (uint32_t i = threadIdx.x; i < range; i += QF_THREADS) s_first[i] = 0xFFFFFFFFu;
    __syncthreads();
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;          // a workgroup takes a contiguous slice of the shard's reads
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += QF_THREADS) {
        const uint32_t slot = v[i] - vmin;
        if (slot < range && (uint32_t)i < s_first[slot]) atomicMin(&s_first[slot], (uint32_t)i);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < range; i += QF_THREADS)
        if (s_first[i] != 0xFFFFFFFFu) atomicMin(&f[i], (long long)(read_offset + s_first[i]));
}
}  // namespace

extern "C" int uq_qname_fused_first_seen(uq_ctx* ctx, const uint32_t* d_vals, uint64_t vals_pitch, uint64_t nreads, uint64_t read_offset,
                                         const uint32_t* h_vmin, const uint32_t* h_range, int ncols, int64_t* d_first) {
    UQ_REQUIRE(ctx && h_vmin && h_range && d_first && ncols >= 1 && ncols <= UQ_QF_MAXC, "uq_qname_fused_first_seen: bad argument");
    UQ_REQUIRE(nreads < (uint64_t(1) << 32) && (nreads == 0 || d_vals), "uq_qname_fused_first_seen: null buffer or more than 2^32-1 reads in a shard");
    QfRanges rg;
    memset(&rg, 0, sizeof(rg));
    for (int c = 0; c < ncols; ++c) { UQ_REQUIRE(h_range[c] <= QF_SMALL, "uq_qname_fused_first_seen: a range beyond %u", QF_SMALL); rg.vmin[c] = h_vmin[c]; rg.range[c] = h_range[c]; }
    const uint64_t cells = (uint64_t)UQ_QF_MAXC * QF_SMALL;
    qf_fill_i64_kernel<<<(uint32_t)((cells + QF_THREADS - 1) / QF_THREADS), QF_THREADS, 0, ctx->stream>>>((long long*)d_first, cells, 0x7FFFFFFFFFFFFFFFll);
    UQ_LAUNCH_CHECK();
    if (nreads) {
        qf_first_seen_global_kernel<<<dim3(UQ_NUM_CU * 4, (uint32_t)ncols), QF_THREADS, 0, ctx->stream>>>(d_vals, vals_pitch, nreads, read_offset, rg, (long long*)d_first);
        UQ_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int uq_qname_fused_fetch(uq_ctx* ctx, const uq_qname_fused* d_q, uq_qname_fused* h_out) {
    UQ_REQUIRE(ctx && d_q && h_out, "uq_qname_fused_fetch: null argument");
    if (ctx->qf_sent != d_q) UQ_TRY(uq_read_back(ctx, ctx->h_pinned + QF_PINNED_AT, d_q, sizeof(uq_qname_fused)));    // (no finish in front)
    ctx->qf_sent = nullptr;
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(h_out, ctx->h_pinned + QF_PINNED_AT, sizeof(uq_qname_fused));
    return 0;
}

// the columns of a fused pass in ONE launch: column c = d_vals + c * pitch -> h_outs[c] (itemsize h_itemsize[c], minus h_sub[c])
struct EncCols { const uint32_t* in[UQ_QF_MAXC]; void* out[UQ_QF_MAXC]; uint32_t sub[UQ_QF_MAXC]; uint32_t isz[UQ_QF_MAXC]; };
__global__ __launch_bounds__(QF_THREADS) void encode_cols_kernel(EncCols e, uint64_t n) {
    const uint32_t c = blockIdx.y;
    const uint32_t* __restrict__ in = e.in[c];
    const uint32_t sub = e.sub[c], isz = e.isz[c];
    // four values per lane: 16-byte loads; the stores are 4 / 8 / 16 bytes wide
    const uint64_t i = ((uint64_t)blockIdx.x * QF_THREADS + threadIdx.x) * 4;
    if (i >= n) return;
    uint32_t v[4];
    if (i + 4 <= n && ((uintptr_t)in & 15) == 0) { const uint4 a = *(const uint4*)(in + i); v[0] = a.x - sub; v[1] = a.y - sub; v[2] = a.z - sub; v[3] = a.w - sub; }
    else for (uint32_t k = 0; k < 4; ++k) v[k] = i + k < n ? in[i + k] - sub : 0;
    const uint32_t m = (uint32_t)(n - i < 4 ? n - i : 4);
    if (isz == 1) {
        uint8_t* o = (uint8_t*)e.out[c] + i;
        if (m == 4 && ((uintptr_t)o & 3) == 0) *(uint32_t*)o = (v[0] & 0xFFu) | ((v[1] & 0xFFu) << 8) | ((v[2] & 0xFFu) << 16) | (v[3] << 24);
        else for (uint32_t k = 0; k < m; ++k) o[k] = (uint8_t)v[k];
    } else if (isz == 2) {
        uint16_t* o = (uint16_t*)e.out[c] + i;
        if (m == 4 && ((uintptr_t)o & 7) == 0) *(uint2*)o = make_uint2((v[0] & 0xFFFFu) | (v[1] << 16), (v[2] & 0xFFFFu) | (v[3] << 16));
        else for (uint32_t k = 0; k < m; ++k) o[k] = (uint16_t)v[k];
    } else if (isz == 4) {
        uint32_t* o = (uint32_t*)e.out[c] + i;
        for (uint32_t k = 0; k < m; ++k) o[k] = v[k];
    } else {
        uint64_t* o = (uint64_t*)e.out[c] + i;
        for (uint32_t k = 0; k < m; ++k) o[k] = v[k];
    }
}

extern "C" int uq_encode_u32_columns(uq_ctx* ctx, const uint32_t* d_vals, uint64_t vals_pitch, uint64_t n, int ncols, const uint32_t* h_sub,
                                     const int* h_itemsize, void* const* h_d_outs) {
    UQ_REQUIRE(ctx && (n == 0 || (d_vals && h_sub && h_itemsize && h_d_outs)) && ncols >= 0 && ncols <= UQ_QF_MAXC, "uq_encode_u32_columns: bad argument");
    if (n == 0 || ncols == 0) return 0;
    EncCols e;
    memset(&e, 0, sizeof(e));
    for (int c = 0; c < ncols; ++c) {
        UQ_REQUIRE(h_itemsize[c] == 1 || h_itemsize[c] == 2 || h_itemsize[c] == 4 || h_itemsize[c] == 8, "uq_encode_u32_columns: itemsize %d not in {1,2,4,8}", h_itemsize[c]);
        e.in[c] = d_vals + (size_t)c * vals_pitch; e.out[c] = h_d_outs[c]; e.sub[c] = h_sub[c]; e.isz[c] = (uint32_t)h_itemsize[c];
    }
    encode_cols_kernel<<<dim3((uint32_t)((n + 4 * QF_THREADS - 1) / (4 * QF_THREADS)), (uint32_t)ncols), QF_THREADS, 0, ctx->stream>>>(e, n);
    UQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int uq_encode_u32(uq_ctx* ctx, const uint32_t* d_val, uint64_t n, uint32_t sub, int itemsize, void* d_out) {
    UQ_REQUIRE(ctx && (n == 0 || (d_val && d_out)), "uq_encode_u32: null argument");
    if (n == 0) return 0;
    const uint32_t blocks = (uint32_t)((n + QF_THREADS - 1) / QF_THREADS);
    switch (itemsize) {
        case 1: encode_u32_kernel<uint8_t><<<blocks, QF_THREADS, 0, ctx->stream>>>(d_val, n, sub, (uint8_t*)d_out); break;
        case 2: encode_u32_kernel<uint16_t><<<blocks, QF_THREADS, 0, ctx->stream>>>(d_val, n, sub, (uint16_t*)d_out); break;
        case 4: encode_u32_kernel<uint32_t><<<blocks, QF_THREADS, 0, ctx->stream>>>(d_val, n, sub, (uint32_t*)d_out); break;
        case 8: encode_u32_kernel<uint64_t><<<blocks, QF_THREADS, 0, ctx->stream>>>(d_val, n, sub, (uint64_t*)d_out); break;
        default: UQ_REQUIRE(false, "uq_encode_u32: itemsize %d not in {1,2,4,8}", itemsize);
    }
    UQ_LAUNCH_CHECK();
    return 0;
}
