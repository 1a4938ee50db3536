// qname_dev.hip -- the heavy parts of the QNAME passes on the device (device form).
//
// The reference's pass 1 (uq.py:394-413) looks sequential, but its result has a closed form:
//   prefix  = longest common prefix of all QNAMEs = line1[: min_i lcp(line1, q_i)]      (suffix likewise)
//   a character c of line1 becomes a separator candidate at the first record e_c whose lcp with line1
//   is <= the last position of c in line1 (that is when the shrinking prefix first sheds a c), and it
//   survives iff no record i >= e_c has count_c(q_i) != count_c(line1)  -- because q_i and line1 share
//   the current prefix, the reference's test `qname[len(prefix):].count(c) == separators[c]` is exactly
//   that equality.  So one pass of reductions (min lcp, min lcs, e_c, last violating record per c)
//   replaces the loop; the host finishes with uq.py:428-444 on a handful of numbers.
// Passes 2 and 4 (uq.py:555-678, 717-736) split every QNAME at the separators and parse the fields;
// `qname_tokenise_kernel` does that for all reads at once, leaving per column an int64 value array and
// an 8-byte big-endian string key array in HBM, plus the reductions the typing rules need.  The typing
// decisions themselves (a few numbers per column) stay on the host, fed by `uq_prefix_distinct`.
// Anything outside the exactly reproducible subset raises a flag and the host-native path takes over.
#include "common.h"

namespace {
constexpr int QN_THREADS = 256;
constexpr int QN_MAXCH = 64;
constexpr int QN_MAXCOLS = 32;

struct Line1 {
    uint8_t text[256];
    uint32_t len;
    uint32_t nch;                   // distinct characters of line1 (candidates)
    uint8_t ch[QN_MAXCH];
    uint16_t cnt[QN_MAXCH];         // occurrences in line1
    uint16_t lastpos[QN_MAXCH];     // last position in line1
    uint8_t slot[256];              // character -> candidate slot, 0xFF = not in line1
};

struct LayoutOut {                  // device + host mirror
    uint32_t min_lcp, min_lcs;
    uint32_t flags;                 // bit0: a QNAME is a proper prefix / suffix of line1 (reference IndexError territory); bit1: QNAME > 255 bytes
    uint32_t nch;
    unsigned long long entry[QN_MAXCH];     // e_c: first record (>= 1) with lcp <= lastpos[c]; UQ_NONE = never
    unsigned long long lastviol[QN_MAXCH];  // last record with count mismatch (0 = none; record 0 is line1 itself)
    uint8_t ch[QN_MAXCH];
};

// A lane's QNAME line is fetched ONCE, with 16-byte (unaligned) global loads, into the lane's private LDS row; all
// the byte-wise work then runs out of LDS.  (Byte loads straight from HBM made both kernels ~8 ms per 10 M reads:
// 64 lanes x 64 different cache lines per load thrash the 32 KiB L1, so every byte came from L2 again.)
constexpr uint32_t QN_ROW = 64;                        // staged bytes per lane; longer lines are read from HBM directly
constexpr uint32_t QN_STRIDE = QN_ROW + 4;             // 17 dwords: consecutive lanes hit different banks

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

__global__ __launch_bounds__(QN_THREADS) void qname_layout_kernel(const uint8_t* __restrict__ buf, const uint64_t* __restrict__ ls,
                                                                   uint64_t n, uint64_t index_base, uint32_t start, Line1 l1,
                                                                   LayoutOut* __restrict__ out) {
    __shared__ uint8_t cnt[QN_MAXCH * QN_THREADS];      // per-lane character counters, slot-major (no bank conflicts)
    __shared__ __align__(16) uint8_t stage[QN_THREADS * QN_STRIDE];
    __shared__ uint8_t s_slot[256];
    __shared__ uint8_t s_text[256];
    __shared__ unsigned long long s_entry[QN_MAXCH], s_viol[QN_MAXCH];
    __shared__ uint32_t s_lcp, s_lcs, s_flags;
    const uint32_t tid = threadIdx.x;
    s_slot[tid] = l1.slot[tid]; s_text[tid] = l1.text[tid];
    if (tid < QN_MAXCH) { s_entry[tid] = UQ_NONE; s_viol[tid] = 0; }
    if (tid == 0) { s_lcp = 0xFFFFFFFFu; s_lcs = 0xFFFFFFFFu; s_flags = 0; }
    for (uint32_t k = 0; k < l1.nch; ++k) cnt[k * QN_THREADS + tid] = 0;
    __syncthreads();
    const uint8_t* buf_end = buf + ls[4 * n];
    uint32_t my_lcp = 0xFFFFFFFFu, my_lcs = 0xFFFFFFFFu;
    const uint64_t stride = (uint64_t)gridDim.x * QN_THREADS;
    for (uint64_t li = (uint64_t)blockIdx.x * QN_THREADS + tid + start; li < n; li += stride) {
        const uint64_t i = index_base + li;           // record number in the whole file (shards: SURVEY.md 8e)
        const uint8_t* q = buf + ls[4 * li];
        const uint32_t ql = (uint32_t)(ls[4 * li + 1] - ls[4 * li] - 1);
        if (ql > 255) { atomicOr(&s_flags, 2u); continue; }
        const uint8_t* row = stage + tid * QN_STRIDE;
        const bool staged = ql <= QN_ROW;
        if (staged) stage_line(stage + tid * QN_STRIDE, q, ql, buf_end);
        auto at = [&](uint32_t j) -> uint32_t { return staged ? row[j] : q[j]; };
        const uint32_t m = ql < l1.len ? ql : l1.len;
        uint32_t lcp = 0;
        while (lcp < m && at(lcp) == s_text[lcp]) ++lcp;
        uint32_t lcs = 0;
        while (lcs < m && at(ql - 1 - lcs) == s_text[l1.len - 1 - lcs]) ++lcs;
        if ((lcp == ql && ql < l1.len) || (lcs == ql && ql < l1.len)) atomicOr(&s_flags, 1u);
        if (lcp < my_lcp) my_lcp = lcp;
        if (lcs < my_lcs) my_lcs = lcs;
        for (uint32_t j = 0; j < ql; ++j) {
            const uint32_t sl = s_slot[at(j)];
            if (sl != 0xFFu) cnt[sl * QN_THREADS + tid] += 1;      // saturation impossible: QNAME lines are < 256 bytes here (checked by the host)
        }
        for (uint32_t k = 0; k < l1.nch; ++k) {
            const uint32_t c = cnt[k * QN_THREADS + tid];
            cnt[k * QN_THREADS + tid] = 0;
            // the lanes of a wave hold increasing record numbers: one lane per wave (the first / last that qualifies)
            // speaks for all, and only if it can still change the table -- same-address LDS atomics from 64 lanes serialise
            const unsigned long long em = __ballot(lcp <= l1.lastpos[k]), vm = __ballot(c != l1.cnt[k]);
            const uint32_t lane = lane_id();
            if (em && lane == (uint32_t)__builtin_ctzll(em) && (unsigned long long)i < s_entry[k]) atomicMin(&s_entry[k], (unsigned long long)i);
            if (vm && lane == 63u - (uint32_t)__builtin_clzll(vm) && (unsigned long long)i > s_viol[k]) atomicMax(&s_viol[k], (unsigned long long)i);
        }
    }
    my_lcp = wave_min(my_lcp); my_lcs = wave_min(my_lcs);
    if (lane_id() == 0) { atomicMin(&s_lcp, my_lcp); atomicMin(&s_lcs, my_lcs); }
    __syncthreads();
    if (tid < l1.nch) {
        if (s_entry[tid] != UQ_NONE) atomicMin(&out->entry[tid], s_entry[tid]);
        if (s_viol[tid]) atomicMax(&out->lastviol[tid], s_viol[tid]);
    }
    if (tid == 0) {
        atomicMin(&out->min_lcp, s_lcp); atomicMin(&out->min_lcs, s_lcs);
        if (s_flags) atomicOr(&out->flags, s_flags);
    }
}

struct Split {
    uint32_t plen, slen, nsep;
    uint8_t seps[QN_MAXCOLS];
    uint8_t inset[256];
};

struct ColsOut {                     // per-column reductions (device + host mirror)
    unsigned long long first_nonint[QN_MAXCOLS];   // smallest record whose field is not a plain decimal integer; UQ_NONE = none
    long long vmin[QN_MAXCOLS], vmax[QN_MAXCOLS];  // over integer fields
    uint32_t any_long[QN_MAXCOLS];                 // a field longer than 8 bytes (or holding a NUL)
    uint32_t flags;                                // bit0 separators out of order / wrong count, bit1 whitespace in a field, bit2 > 18 digits, bit3 QNAME shorter than prefix+suffix
    uint32_t pad;
};

__global__ __launch_bounds__(QN_THREADS) void qname_tokenise_kernel(const uint8_t* __restrict__ buf, const uint64_t* __restrict__ ls, uint64_t n,
                                                                    Split sp, long long* const* __restrict__ vals,
                                                                    unsigned long long* const* __restrict__ strs, ColsOut* __restrict__ out) {
    __shared__ unsigned long long s_first[QN_MAXCOLS];
    __shared__ long long s_min[QN_MAXCOLS], s_max[QN_MAXCOLS];
    __shared__ uint32_t s_long[QN_MAXCOLS];
    __shared__ uint32_t s_flags;
    __shared__ uint8_t s_inset[256];
    __shared__ __align__(16) uint8_t stage[QN_THREADS * QN_STRIDE];
    const uint32_t tid = threadIdx.x, lane = lane_id();
    const uint32_t ncols = sp.nsep + 1;
    s_inset[tid] = sp.inset[tid];
    if (tid < QN_MAXCOLS) { s_first[tid] = UQ_NONE; s_min[tid] = 0x7FFFFFFFFFFFFFFFll; s_max[tid] = -0x7FFFFFFFFFFFFFFFll - 1; s_long[tid] = 0; }
    if (tid == 0) s_flags = 0;
    __syncthreads();
    const uint8_t* buf_end = buf + ls[4 * n];
    const uint64_t stride = (uint64_t)gridDim.x * QN_THREADS;
    for (uint64_t i = (uint64_t)blockIdx.x * QN_THREADS + tid; i < n; i += stride) {
        const uint8_t* q = buf + ls[4 * i];
        const uint32_t ql = (uint32_t)(ls[4 * i + 1] - ls[4 * i] - 1);
        uint32_t flags = 0;
        if (ql < sp.plen + sp.slen) { atomicOr(&s_flags, 8u); continue; }
        const uint8_t* row = stage + tid * QN_STRIDE;
        const bool staged = ql <= QN_ROW;
        if (staged) stage_line(stage + tid * QN_STRIDE, q, ql, buf_end);
        auto at = [&](uint32_t j) -> uint32_t { return staged ? row[j] : q[j]; };
        uint32_t pos = sp.plen;
        const uint32_t end = ql - sp.slen;
        for (uint32_t c = 0; c < ncols; ++c) {
            // field c runs to the next separator-set character (which must be separators[c]) or to `end`
            uint32_t e = pos;
            while (e < end && !s_inset[at(e)]) ++e;
            if (c < sp.nsep) { if (e >= end || at(e) != sp.seps[c]) flags |= 1u; }
            else if (e != end) flags |= 1u;
            const uint32_t fl = e - pos;
            // parse: [+-]digits, nothing else (Python would also accept surrounding whitespace: flagged instead)
            unsigned long long key = 0;
            bool odd = false, ws = false;            // odd: NUL or non-ASCII byte (the 8-byte key cannot carry it)
            for (uint32_t k = 0; k < fl; ++k) {
                const uint32_t b = at(pos + k);
                if (k < 8) key |= (unsigned long long)b << (56 - 8 * k);
                odd |= (b == 0 || b >= 0x80);
                ws |= (b == ' ' || (b >= 9 && b <= 13));
            }
            uint32_t k = 0;
            bool neg = false;
            const uint32_t b0 = fl ? at(pos) : 0u;
            if (b0 == '+' || b0 == '-') { neg = b0 == '-'; k = 1; }
            const bool sign = k != 0;
            bool isint = fl > k;
            unsigned long long mag = 0;
            const uint32_t nd = fl - k;
            const bool lead0 = nd > 1 && at(pos + k) == '0';
            for (; k < fl; ++k) {
                const uint32_t b = at(pos + k);
                if (b < '0' || b > '9') { isint = false; break; }
                mag = mag * 10 + (b - '0');
            }
            if (isint && nd > 18) { flags |= 4u; mag = 0; }
            if (ws) flags |= 2u;
            const long long v = neg ? -(long long)mag : (long long)mag;
            uint32_t lng = odd ? 1u : 0u;
            if (isint && (b0 == '+' || lead0 || (neg && mag == 0))) lng |= 4u;     // '+7', '007', '-0': text is not THE decimal of the value
            if (fl > 8) {
                if (isint && !sign && !lead0) { key = 0x8000000000000000ull | mag; lng |= 2u; }   // text <-> value is a bijection here
                else lng |= 1u;
            }
            vals[c][i] = isint ? v : 0;
            strs[c][i] = __builtin_bswap64(key);     // memory order = text order (rows for uq_unique_rows)
            if (lng && (s_long[c] & lng) != lng) atomicOr(&s_long[c], lng);
            // same-address LDS atomics from 64 lanes serialise: lanes that cannot change the table stay out
            if (isint) { if (v < s_min[c]) atomicMin(&s_min[c], v); if (v > s_max[c]) atomicMax(&s_max[c], v); }
            else if ((unsigned long long)i < s_first[c]) atomicMin(&s_first[c], (unsigned long long)i);
            pos = e + 1;
        }
        if (flags) atomicOr(&s_flags, flags);
    }
    __syncthreads();
    if (tid < ncols) {
        if (s_first[tid] != UQ_NONE) atomicMin(&out->first_nonint[tid], s_first[tid]);
        atomicMin(&out->vmin[tid], s_min[tid]); atomicMax(&out->vmax[tid], s_max[tid]);
        if (s_long[tid]) atomicOr(&out->any_long[tid], s_long[tid]);
    }
    if (tid == 0 && s_flags) atomicOr(&out->flags, s_flags);
}

// number of groups (runs of equal sorted keys) whose FIRST member in file order has index <= T_k
// the checkpoints travel by value in the kernel-argument segment: nothing of the caller's has to stay alive, no copy, no wait
struct Thresholds { unsigned long long t[64]; };

template <typename IDX>
__global__ __launch_bounds__(QN_THREADS) void prefix_distinct_kernel(const IDX* __restrict__ perm, const uint32_t* __restrict__ skey, uint64_t n,
                                                                     Thresholds thv, int nth,
                                                                     unsigned long long* __restrict__ counts) {
    const unsigned long long* th = thv.t;
    __shared__ uint32_t s_cnt[64];
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * QN_THREADS;
    for (uint64_t j = (uint64_t)blockIdx.x * QN_THREADS + threadIdx.x; j < n; j += stride) {
        if (j == 0 || skey[j] != skey[j - 1]) {
            const unsigned long long f = perm[j];        // stable sort: the head of a run is its first occurrence
            for (int k = 0; k < nth; ++k)
                if (f <= th[k]) atomicAdd(&s_cnt[k], 1u);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nth && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

// ---- distinct counts of an INTEGER column without sorting it (uq_int_prefix_distinct).
// first[v - vmin] = lowest read index (file-wide) holding value v, over reads [0, n): small ranges through a private LDS table
// per workgroup (a column of four lanes or sixty-four tiles would otherwise be ten million atomics on a handful of
// addresses), wide ranges straight to the global table (contention falls with the range).
template <bool PRIVATE>
__global__ __launch_bounds__(QN_THREADS) void first_seen_kernel(const long long* __restrict__ val, uint64_t n, long long vmin, uint32_t range,
                                                               uint64_t index_base, unsigned long long* __restrict__ first) {
    extern __shared__ unsigned long long s_first[];
    if (PRIVATE) {
        for (uint32_t i = threadIdx.x; i < range; i += QN_THREADS) s_first[i] = UQ_NONE;
        __syncthreads();
    }
    // a workgroup takes a CONTIGUOUS slice of the reads: the first lane to see a value in a slice usually settles it
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += QN_THREADS) {
        const unsigned long long slot = (unsigned long long)(val[i] - vmin);
        if (slot >= range) continue;                         // not in [vmin, vmax]: the caller's bounds were wrong (never with uq_qname_tokenise's)
        const unsigned long long gi = index_base + i;
        if (PRIVATE) { if (gi < s_first[slot]) atomicMin(&s_first[slot], gi); }
        else if (gi < first[slot]) atomicMin(&first[slot], gi);
    }
    if (PRIVATE) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < range; i += QN_THREADS)
            if (s_first[i] != UQ_NONE && s_first[i] < first[i]) atomicMin(&first[i], s_first[i]);
    }
}

// counts[k] = number of values whose first occurrence is <= T_k
__global__ __launch_bounds__(QN_THREADS) void first_seen_count_kernel(const unsigned long long* __restrict__ first, uint32_t range,
                                                                      Thresholds thv, int nth,
                                                                      unsigned long long* __restrict__ counts) {
    const unsigned long long* th = thv.t;
    __shared__ uint32_t s_cnt[64];
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t j = (uint64_t)blockIdx.x * QN_THREADS + threadIdx.x; j < range; j += (uint64_t)gridDim.x * QN_THREADS) {
        const unsigned long long f = first[j];
        if (f == UQ_NONE) continue;
        for (int k = 0; k < nth; ++k)
            if (f <= th[k]) atomicAdd(&s_cnt[k], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < nth && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

template <typename T>
__global__ void encode_int_kernel(const long long* __restrict__ val, uint64_t n, long long sub, T* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * QN_THREADS + threadIdx.x;
    if (i < n) out[i] = (T)(unsigned long long)(val[i] - sub);
}

uint32_t grid_for(uint64_t n) {
    uint64_t b = (n + QN_THREADS - 1) / QN_THREADS;
    return (uint32_t)(b < (uint64_t)UQ_NUM_CU * 8 ? (b ? b : 1) : (uint64_t)UQ_NUM_CU * 8);
}
}  // namespace

extern "C" int uq_qname_layout(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t nreads, uint64_t read_index_base,
                               const uint8_t* h_line1, uint32_t line1_len, uq_qname_layout_result* h_out) {
    UQ_REQUIRE(ctx && h_line1 && h_out && (nreads == 0 || (d_buf && d_line_start)), "uq_qname_layout: null argument");
    UQ_REQUIRE(line1_len >= 1 && line1_len <= 255, "uq_qname_layout: first QNAME line must be 1..255 bytes");
    static_assert(sizeof(uq_qname_layout_result) == sizeof(LayoutOut), "layout result mirrors differ");
    Line1 l1;
    memset(&l1, 0, sizeof(l1));
    memset(l1.slot, 0xFF, 256);
    memcpy(l1.text, h_line1, line1_len);
    l1.len = line1_len;
    for (uint32_t p = 0; p < line1_len; ++p) {
        const uint8_t c = h_line1[p];
        if (l1.slot[c] == 0xFF) {
            UQ_REQUIRE(l1.nch < QN_MAXCH, "uq_qname_layout: more than 64 distinct characters in the first QNAME");
            l1.slot[c] = (uint8_t)l1.nch; l1.ch[l1.nch] = c; ++l1.nch;
        }
        l1.cnt[l1.slot[c]] += 1; l1.lastpos[l1.slot[c]] = (uint16_t)p;
    }
    void* scr;
    UQ_TRY(uq_scratch(ctx, sizeof(LayoutOut) + 256, &scr));
    LayoutOut init;
    memset(&init, 0, sizeof(init));
    init.min_lcp = line1_len; init.min_lcs = line1_len;
    for (int k = 0; k < QN_MAXCH; ++k) init.entry[k] = UQ_NONE;
    UQ_CHECK_HIP(hipMemcpyAsync(scr, &init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    const uint32_t start = read_index_base == 0 ? 1u : 0u;      // read 0 of the file is line 1 itself
    if (nreads > start) {
        qname_layout_kernel<<<grid_for(nreads - start), QN_THREADS, 0, ctx->stream>>>(d_buf, d_line_start, nreads, read_index_base, start, l1,
                                                                                     (LayoutOut*)scr);
        UQ_LAUNCH_CHECK();
    }
    static_assert(sizeof(LayoutOut) % 4 == 0 && sizeof(LayoutOut) <= 16384, "read back through the pinned staging");
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned, scr, sizeof(LayoutOut)));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(h_out, ctx->h_pinned, sizeof(LayoutOut));
    for (uint32_t k = 0; k < l1.nch; ++k) h_out->ch[k] = l1.ch[k];
    h_out->nch = l1.nch;
    return 0;
}

extern "C" int uq_qname_tokenise(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t nreads, uint32_t prefix_len,
                                 uint32_t suffix_len, const uint8_t* h_separators, uint32_t nsep, int64_t* const* h_d_vals,
                                 uint64_t* const* h_d_strs, uq_qname_cols_result* h_out) {
    UQ_REQUIRE(ctx && h_separators && h_d_vals && h_d_strs && h_out && (nreads == 0 || (d_buf && d_line_start)), "uq_qname_tokenise: null argument");
    UQ_REQUIRE(nsep >= 1 && nsep < QN_MAXCOLS, "uq_qname_tokenise: 1..31 separators supported");
    static_assert(sizeof(uq_qname_cols_result) == sizeof(ColsOut), "column result mirrors differ");
    Split sp;
    memset(&sp, 0, sizeof(sp));
    sp.plen = prefix_len; sp.slen = suffix_len; sp.nsep = nsep;
    for (uint32_t k = 0; k < nsep; ++k) { sp.seps[k] = h_separators[k]; sp.inset[h_separators[k]] = 1; }
    const uint32_t ncols = nsep + 1;
    void* scr;
    const size_t ptr_bytes = 2 * QN_MAXCOLS * sizeof(void*);
    UQ_TRY(uq_scratch(ctx, sizeof(ColsOut) + ptr_bytes + 512, &scr));
    ColsOut init;
    memset(&init, 0, sizeof(init));
    for (int c = 0; c < QN_MAXCOLS; ++c) { init.first_nonint[c] = UQ_NONE; init.vmin[c] = 0x7FFFFFFFFFFFFFFFll; init.vmax[c] = -0x7FFFFFFFFFFFFFFFll - 1; }
    uint8_t* base = (uint8_t*)scr;
    void* h_ptrs[2 * QN_MAXCOLS] = {nullptr};
    for (uint32_t c = 0; c < ncols; ++c) { h_ptrs[c] = h_d_vals[c]; h_ptrs[QN_MAXCOLS + c] = h_d_strs[c]; }
    UQ_CHECK_HIP(hipMemcpyAsync(base, &init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    UQ_CHECK_HIP(hipMemcpyAsync(base + ((sizeof(ColsOut) + 255) & ~size_t(255)), h_ptrs, ptr_bytes, hipMemcpyHostToDevice, ctx->stream));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));       // h_ptrs / init live on this stack frame
    long long* const* d_vals = (long long* const*)(base + ((sizeof(ColsOut) + 255) & ~size_t(255)));
    unsigned long long* const* d_strs = (unsigned long long* const*)(d_vals + QN_MAXCOLS);
    if (nreads) {
        qname_tokenise_kernel<<<grid_for(nreads), QN_THREADS, 0, ctx->stream>>>(d_buf, d_line_start, nreads, sp, d_vals, d_strs, (ColsOut*)base);
        UQ_LAUNCH_CHECK();
    }
    static_assert(sizeof(ColsOut) % 4 == 0 && sizeof(ColsOut) <= 16384, "read back through the pinned staging");
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned, base, sizeof(ColsOut)));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(h_out, ctx->h_pinned, sizeof(ColsOut));
    return 0;
}

extern "C" int uq_prefix_distinct(uq_ctx* ctx, const void* d_perm, int perm_itemsize, const uint32_t* d_sorted_key, uint64_t n,
                                  const uint64_t* h_thresholds, int nthresholds, uint64_t* h_counts) {
    UQ_REQUIRE(ctx && h_thresholds && h_counts && nthresholds >= 1 && nthresholds <= 64, "uq_prefix_distinct: bad argument");
    UQ_REQUIRE(perm_itemsize == 4 || perm_itemsize == 8, "uq_prefix_distinct: perm_itemsize must be 4 or 8");
    for (int k = 0; k < nthresholds; ++k) h_counts[k] = 0;
    if (n == 0) return 0;
    UQ_REQUIRE(d_perm && d_sorted_key, "uq_prefix_distinct: null buffer");
    void* scr;
    UQ_TRY(uq_scratch(ctx, 2048, &scr));
    unsigned long long* d_cnt = (unsigned long long*)scr + 64;
    Thresholds d_th;
    memset(&d_th, 0, sizeof(d_th));
    for (int k = 0; k < nthresholds; ++k) d_th.t[k] = h_thresholds[k];
    UQ_CHECK_HIP(hipMemsetAsync(d_cnt, 0, 64 * 8, ctx->stream));
    if (perm_itemsize == 4)
        prefix_distinct_kernel<uint32_t><<<grid_for(n), QN_THREADS, 0, ctx->stream>>>((const uint32_t*)d_perm, d_sorted_key, n, d_th, nthresholds, d_cnt);
    else
        prefix_distinct_kernel<unsigned long long><<<grid_for(n), QN_THREADS, 0, ctx->stream>>>((const unsigned long long*)d_perm, d_sorted_key, n, d_th,
                                                                                               nthresholds, d_cnt);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned, d_cnt, (size_t)nthresholds * 8));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(h_counts, ctx->h_pinned, (size_t)nthresholds * 8);
    return 0;
}

extern "C" int uq_int_prefix_distinct(uq_ctx* ctx, const int64_t* d_val, uint64_t n, int64_t vmin, uint64_t range, uint64_t read_index_base,
                                      const uint64_t* h_thresholds, int nthresholds, uint64_t* h_counts) {
    UQ_REQUIRE(ctx && h_thresholds && h_counts && nthresholds >= 1 && nthresholds <= 64, "uq_int_prefix_distinct: bad argument");
    UQ_REQUIRE(range >= 1 && range <= (uint64_t(1) << 26), "uq_int_prefix_distinct: value range %llu not in 1 .. 2^26", (unsigned long long)range);
    for (int k = 0; k < nthresholds; ++k) h_counts[k] = 0;
    if (n == 0) return 0;
    UQ_REQUIRE(d_val, "uq_int_prefix_distinct: null buffer");
    void* scr;
    UQ_TRY(uq_scratch(ctx, 2048 + range * 8, &scr));
    unsigned long long* d_cnt = (unsigned long long*)scr + 64;
    unsigned long long* d_first = (unsigned long long*)scr + 256;
    Thresholds d_th;
    memset(&d_th, 0, sizeof(d_th));
    for (int k = 0; k < nthresholds; ++k) d_th.t[k] = h_thresholds[k];
    UQ_CHECK_HIP(hipMemsetAsync(d_cnt, 0, 64 * 8, ctx->stream));
    UQ_CHECK_HIP(hipMemsetAsync(d_first, 0xFF, range * 8, ctx->stream));
    const uint32_t grid = grid_for(n);
    if (range <= 4096) first_seen_kernel<true><<<grid, QN_THREADS, range * 8, ctx->stream>>>((const long long*)d_val, n, vmin, (uint32_t)range, read_index_base, d_first);
    else first_seen_kernel<false><<<grid, QN_THREADS, 0, ctx->stream>>>((const long long*)d_val, n, vmin, (uint32_t)range, read_index_base, d_first);
    UQ_LAUNCH_CHECK();
    first_seen_count_kernel<<<grid_for(range), QN_THREADS, 0, ctx->stream>>>(d_first, (uint32_t)range, d_th, nthresholds, d_cnt);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned, d_cnt, (size_t)nthresholds * 8));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(h_counts, ctx->h_pinned, (size_t)nthresholds * 8);
    return 0;
}

extern "C" int uq_encode_int(uq_ctx* ctx, const int64_t* d_val, uint64_t n, int64_t sub, int itemsize, void* d_out) {
    UQ_REQUIRE(ctx && (n == 0 || (d_val && d_out)), "uq_encode_int: null argument");
    if (n == 0) return 0;
    const uint32_t blocks = (uint32_t)((n + QN_THREADS - 1) / QN_THREADS);
    switch (itemsize) {
        case 1: encode_int_kernel<uint8_t><<<blocks, QN_THREADS, 0, ctx->stream>>>((const long long*)d_val, n, sub, (uint8_t*)d_out); break;
        case 2: encode_int_kernel<uint16_t><<<blocks, QN_THREADS, 0, ctx->stream>>>((const long long*)d_val, n, sub, (uint16_t*)d_out); break;
        case 4: encode_int_kernel<uint32_t><<<blocks, QN_THREADS, 0, ctx->stream>>>((const long long*)d_val, n, sub, (uint32_t*)d_out); break;
        case 8: encode_int_kernel<uint64_t><<<blocks, QN_THREADS, 0, ctx->stream>>>((const long long*)d_val, n, sub, (uint64_t*)d_out); break;
        default: UQ_REQUIRE(false, "uq_encode_int: itemsize %d not in {1,2,4,8}", itemsize);
    }
    UQ_LAUNCH_CHECK();
    return 0;
}
