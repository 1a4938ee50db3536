// stats.hip -- pass-1 statistics on the device.
// Replaces the per-base dict increment `static_qualities[base][qual] += 1` and the length range /
// sanity checks of uq.py:366-375, 382, 388, 415-425.
//
// Each half-wave (32 lanes) owns one record at a time; a lane takes 4 consecutive (base, quality)
// pairs per step with two unaligned dword loads, so a 150-base read is one step of 38 lanes.  Counts
// go to a workgroup-private LDS table [32 base slots][256 qualities] of u32 (ds_add_u32); base bytes
// outside the 31 common nucleotide letters use global atomics on the full 256 x 256 table, so any
// byte is counted exactly.  Tables are flushed to the u64 global table once per workgroup.
// HBM traffic: the record bytes once + 32 B of line offsets per record.
#include "common.h"

namespace {
constexpr int ST_THREADS = 256;
constexpr int ST_SLOTS = 32;              // slot 31 = "other"
constexpr uint32_t ST_OTHER = 31;

struct SlotLut { uint8_t slot[256]; uint8_t byte_of[ST_SLOTS]; };

SlotLut make_slot_lut() {
    SlotLut l;
    memset(l.slot, ST_OTHER, sizeof(l.slot));
    memset(l.byte_of, 0, sizeof(l.byte_of));
    const char* common = "ACGTNacgtnRYKMSWBDHVU.-*Xryksw=";  // 31 letters seen in sequence lines
    for (int i = 0; common[i] && i < 31; ++i) { l.slot[(uint8_t)common[i]] = (uint8_t)i; l.byte_of[i] = (uint8_t)common[i]; }
    return l;
}

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint32_t load_tail(const uint8_t* buf, uint64_t pos, uint64_t nbytes) {
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i)
        if (pos + i < nbytes) v |= (uint32_t)buf[pos + i] << (8 * i);
    return v;
}

__global__ __launch_bounds__(ST_THREADS) void stats_kernel(const uint8_t* __restrict__ buf, uint64_t nbytes,
                                                           const uint64_t* __restrict__ ls, uint64_t first,
                                                           uint64_t n, SlotLut lut, uq_stats* __restrict__ st) {
    __shared__ uint32_t hist[ST_SLOTS * 256];
    __shared__ uint8_t slot_of[256];
    for (int i = threadIdx.x; i < ST_SLOTS * 256; i += ST_THREADS) hist[i] = 0;
    if (threadIdx.x < 256) slot_of[threadIdx.x] = lut.slot[threadIdx.x];
    __syncthreads();

    const uint32_t lane = lane_id();
    const uint32_t hl = lane & 31;
    const uint64_t gid = ((uint64_t)blockIdx.x * (ST_THREADS / 64) + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const uint64_t G = (uint64_t)gridDim.x * (ST_THREADS / 64) * 2;
    uint32_t lmin = 0xFFFFFFFFu, lmax = 0, rmax = 0;
    uint64_t bad_plus = UQ_NONE, bad_len = UQ_NONE;

    for (uint64_t r = gid; r < n; r += G) {
        const uint64_t* p = ls + 4 * (first + r);
        const uint64_t p0 = p[0], s = p[1], e1 = p[2], q = p[3], e2 = p[4];
        const uint32_t L = (uint32_t)(e1 - s - 1), Lq = (uint32_t)(e2 - q - 1);
        if (hl == 0) {
            if (buf[e1] != '+') bad_plus = bad_plus < r ? bad_plus : r;
            if (L != Lq) bad_len = bad_len < r ? bad_len : r;
            lmin = L < lmin ? L : lmin;
            lmax = L > lmax ? L : lmax;
            uint32_t rb = (uint32_t)(e2 - p0);
            rmax = rb > rmax ? rb : rmax;
        }
        const uint32_t Lc = L < Lq ? L : Lq;
        for (uint32_t j = hl * 4; j < Lc; j += 128) {
            uint32_t vb = load_u32_unaligned(buf + s + j);   // s + j + 3 <= e1 + 1 < nbytes always
            uint32_t vq = (q + j + 4 <= nbytes) ? load_u32_unaligned(buf + q + j) : load_tail(buf, q + j, nbytes);
            uint32_t cnt = Lc - j < 4 ? Lc - j : 4;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if (k < cnt) {
                    uint32_t b = (vb >> (8 * k)) & 255u, c = (vq >> (8 * k)) & 255u;
                    uint32_t sl = slot_of[b];
                    if (sl != ST_OTHER) atomicAdd(&hist[sl * 256 + c], 1u);
                    else atomicAdd((unsigned long long*)&st->counts[b * 256 + c], 1ull);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (ST_SLOTS - 1) * 256; i += ST_THREADS) {
        uint32_t v = hist[i];
        if (v) atomicAdd((unsigned long long*)&st->counts[(uint32_t)lut.byte_of[i >> 8] * 256 + (i & 255)], (unsigned long long)v);
    }
    lmin = wave_min(lmin); lmax = wave_max(lmax); rmax = wave_max(rmax);
    bad_plus = wave_min(bad_plus); bad_len = wave_min(bad_len);
    if (lane == 0) {
        if (lmin != 0xFFFFFFFFu) atomicMin(&st->len_min, lmin);
        atomicMax(&st->len_max, lmax);
        atomicMax(&st->max_record_bytes, rmax);
        if (bad_plus != UQ_NONE) atomicMin((unsigned long long*)&st->bad_plus, (unsigned long long)(first + bad_plus));
        if (bad_len != UQ_NONE) atomicMin((unsigned long long*)&st->bad_len, (unsigned long long)(first + bad_len));
    }
}

__global__ void stats_init_kernel(uq_stats* st) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 65536) st->counts[i] = 0;
    if (i == 0) {
        st->bad_plus = UQ_NONE; st->bad_len = UQ_NONE;
        st->len_min = 0xFFFFFFFFu; st->len_max = 0; st->max_record_bytes = 0; st->reserved = 0;
    }
}

// First occurrence of every base byte (ordering of N-trick candidates, uq.py:480 under pypy/py3).
__global__ __launch_bounds__(256) void first_occurrence_kernel(const uint8_t* __restrict__ buf,
                                                               const uint64_t* __restrict__ ls, uint64_t first,
                                                               uint64_t n, uint64_t index_base, uint64_t* __restrict__ out) {
    __shared__ unsigned long long seen[256];
    seen[threadIdx.x] = UQ_NONE;
    __syncthreads();
    const uint64_t gw = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6));
    const uint64_t GW = (uint64_t)gridDim.x * 4;
    const uint32_t lane = lane_id();
    for (uint64_t r = gw; r < n; r += GW) {
        const uint64_t* p = ls + 4 * (first + r);
        const uint64_t s = p[1], e1 = p[2];
        const uint32_t L = (uint32_t)(e1 - s - 1);
        for (uint32_t j = lane; j < L; j += 64) {
            uint32_t b = buf[s + j];
            unsigned long long key = ((index_base + r) << 20) | (j & 0xFFFFFu);
            if (key < seen[b]) atomicMin(&seen[b], key);
        }
    }
    __syncthreads();
    if (seen[threadIdx.x] != UQ_NONE) atomicMin((unsigned long long*)&out[threadIdx.x], seen[threadIdx.x]);
}
}  // namespace

extern "C" int uq_stats_init(uq_ctx* ctx, uq_stats* d_stats) {
    UQ_REQUIRE(ctx && d_stats, "uq_stats_init: null argument");
    stats_init_kernel<<<256, 256, 0, ctx->stream>>>(d_stats);
    UQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int uq_stats_accumulate(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start,
                                   uint64_t first_read, uint64_t nreads, uq_stats* d_stats) {
    UQ_REQUIRE(ctx && d_buf && d_line_start && d_stats, "uq_stats_accumulate: null argument");
    if (nreads == 0) return 0;
    // The kernel needs the end of the buffer to guard its last unaligned load: read it from the index.
    uint64_t* tmp = ctx->h_pinned;
    UQ_CHECK_HIP(hipMemcpyAsync(tmp, d_line_start + 4 * (first_read + nreads), 8, hipMemcpyDeviceToHost, ctx->stream));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    const uint64_t nbytes = tmp[0];
    static const SlotLut lut = make_slot_lut();
    uint64_t groups = (nreads + 1) / 2;
    uint32_t blocks = (uint32_t)((groups + 3) / 4);
    if (blocks > UQ_NUM_CU * 4) blocks = UQ_NUM_CU * 4;
    if (blocks == 0) blocks = 1;
    stats_kernel<<<blocks, ST_THREADS, 0, ctx->stream>>>(d_buf, nbytes, d_line_start, first_read, nreads, lut, d_stats);
    UQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int uq_first_occurrence(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start,
                                   uint64_t first_read, uint64_t nreads, uint64_t read_index_base, uint64_t* d_first) {
    UQ_REQUIRE(ctx && d_buf && d_line_start && d_first, "uq_first_occurrence: null argument");
    if (nreads == 0) return 0;
    uint32_t blocks = (uint32_t)((nreads + 3) / 4);
    if (blocks > UQ_NUM_CU * 8) blocks = UQ_NUM_CU * 8;
    first_occurrence_kernel<<<blocks, 256, 0, ctx->stream>>>(d_buf, d_line_start, first_read, nreads, read_index_base, d_first);
    UQ_LAUNCH_CHECK();
    return 0;
}
