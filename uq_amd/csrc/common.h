// common.h -- context, error plumbing and wave/block primitives shared by the kernels of libuqhip.so.
// gfx950 only: wavefront = 64 lanes, 256 CUs in 8 XCDs, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/uqhip.h"

#define UQ_WAVE 64
#define UQ_NUM_CU 256

struct uq_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    void* scratch;          // grow-only scratch pool
    size_t scratch_bytes;
    hipEvent_t ev0, ev1;
    // cache of the newline counts of the last uq_count_lines call (reused by uq_index_lines)
    const uint8_t* idx_buf; uint64_t idx_nbytes; uint64_t idx_nlines;
    uint32_t* idx_partials; size_t idx_partials_cap;
    uint16_t* idx_bitmap;   // newline bitmap of that buffer: one u16 per 16-byte vector (index.hip)
    uint64_t* h_pinned;     // small pinned host staging (64 KiB).  Who uses which part (uint64 index): [0, 3200) whatever the running call reads
                            // back (scalars, uq_stats_compact, QNAME reductions); [4096, 5120) the radix sort's digit census; [8000, 8003) the
                            // queued census's line count and flags, which must survive the calls queued behind it (uq_count_lines_wait); [5200, 5520) the
                            // fused QNAME pass's structure (uq_qname_fused_finish -> uq_qname_fused_fetch)
    uint64_t* d_pinned;     // the same memory as the device sees it (uq_read_back)
    void* scan_ws; size_t scan_ws_bytes;   // partial sums of the hierarchical scans
    // the queued form of the census (uq_count_lines_end_async): the line count stays on the device for the kernels queued behind it
    unsigned long long* d_async;           // [0] line count  [1] line_start capacity exceeded
    const uint8_t* async_buf; uint64_t async_nbytes; bool async_read;
    const void* qf_sent;                   // the uq_qname_fused whose read-back uq_qname_fused_finish has queued ([5200, 5520) of h_pinned)
    // the row sort's round 0 (uq_sort_config / uq_sort_counters): tables of msd_min_rows rows and more take the MSD partition (0 = the default
    // threshold, < 0 = never); msd_levels > 0: the digits of its levels instead of the automatic plan; how many sorts went which way
    long long msd_min_rows; int msd_levels; int msd_bits[4];
    unsigned long long n_msd_rounds, n_lsd_rounds;
    // per-context caches of what used to be function statics (a second context on another device, or on another thread, must not share them):
    // register counts of the pack kernels by kernel pointer (the persistent grid is sized from them), and whether this context's device has had
    // qf_wide_kernel's dynamic LDS limit raised
    const void* kreg_key[8]; int kreg_val[8]; int kreg_n;
    bool qf_attr_set;
};

void uq_set_error(const char* fmt, ...);

#define UQ_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            uq_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

#define UQ_REQUIRE(cond, ...)                                                           \
    do {                                                                                \
        if (!(cond)) { uq_set_error(__VA_ARGS__); return 2; }                           \
    } while (0)

#define UQ_TRY(expr)                                                                    \
    do { int _r = (expr); if (_r) return _r; } while (0)

#define UQ_LAUNCH_CHECK() UQ_CHECK_HIP(hipGetLastError())

// Small results for the host: a kernel stores `bytes` (a multiple of 4) from d_src into the context's pinned staging at h_dst (which
// must lie inside ctx->h_pinned), queued on the context's stream -- no copy command: under a kernel that fills the device a blit / DMA
// copy of a few bytes was seen to wait 0.2 - 0.6 ms for its turn, an ordinary one-workgroup kernel is dispatched at once.
int uq_read_back(uq_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int uq_async_read_back(uq_ctx* ctx);        // index.hip: the queued census's results (line count, overflow flags) -> pinned staging

// Scratch: returns a pointer into the context's pool, growing it if needed.  A grow synchronises the
// stream first (the old pool may still be in use by queued kernels).
int uq_scratch(uq_ctx* ctx, size_t bytes, void** out);

// A tiny bump allocator over the scratch pool for functions that need several temporaries.
struct ScratchPlan {
    size_t off = 0;
    size_t add(size_t bytes) { size_t o = off; off += (bytes + 255) & ~size_t(255); return o; }
};

// Division by a runtime constant d for small dividends (tile-local indices): q = umulhi(k, magic) is
// off by at most one in either direction; fast_divmod repairs it.
static inline uint32_t magic_u32(uint32_t d) {
    uint64_t m = ((uint64_t(1) << 32) + d - 1) / d;
    return m > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)m;
}

static inline uint32_t div_up_u64(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// Exclusive prefix sums on device arrays (scan.hip).  In place is allowed.  n up to 2^32.
int uq_scan_exclusive_u32(uq_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, uint64_t n, uint64_t* d_total /*device, may be null*/);
int uq_scan_exclusive_u64(uq_ctx* ctx, const uint64_t* d_in, uint64_t* d_out, uint64_t n, uint64_t* d_total);

#ifdef __HIPCC__
__device__ __forceinline__ void fast_divmod(uint32_t k, uint32_t d, uint32_t magic, uint32_t& q, uint32_t& r) {
    q = __umulhi(k, magic);
    r = k - q * d;
    if ((int32_t)r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
}

// ---- wave-level primitives (64 lanes)
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

template <typename T>
__device__ __forceinline__ T wave_inclusive_sum(T v) {
    const uint32_t lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { T o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { T o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}

// Block-wide exclusive sum for blocks of NW waves.  `lds` needs NW+1 entries.  Returns the exclusive
// prefix of `v` over the block in thread order and the block total in `total`.
template <typename T, int NW>
__device__ __forceinline__ T block_exclusive_sum(T v, T* lds, T& total) {
    const uint32_t lane = lane_id();
    const uint32_t w = threadIdx.x >> 6;
    T inc = wave_inclusive_sum(v);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        T x = lds[i];
        if ((uint32_t)i < w) base += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}
#endif
