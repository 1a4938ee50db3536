// scan.hip -- hierarchical exclusive prefix sum (reduce -> scan partials -> scan tiles).
// No inter-workgroup hand-off inside a launch: every dependency is a kernel boundary, so the
// result cannot depend on dispatch order or XCD placement.
#include "common.h"

namespace {
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048

template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const T* __restrict__ in, uint64_t n, T* __restrict__ partials) {
    __shared__ T lds[SCAN_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    T s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        uint64_t k = base + (uint64_t)i * SCAN_THREADS + threadIdx.x;
        if (k < n) s += in[k];
    }
    s = wave_sum(s);
    if (lane_id() == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        T t = 0;
        for (int i = 0; i < SCAN_THREADS / 64; ++i) t += lds[i];
        partials[blockIdx.x] = t;
    }
}

// Each thread owns SCAN_ITEMS consecutive elements (blocked arrangement).
template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_kernel(const T* __restrict__ in, T* __restrict__ out, uint64_t n,
                                                                const T* __restrict__ offsets, T* __restrict__ total) {
    __shared__ T lds[SCAN_THREADS / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
    T s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? in[base + i] : T(0);
        s += v[i];
    }
    T tot;
    T ex = block_exclusive_sum<T, SCAN_THREADS / 64>(s, lds, tot);
    T off = offsets ? offsets[blockIdx.x] : T(0);
    T run = off + ex;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = off + tot;
}

template <typename T>
__global__ void widen_total_kernel(const T* t, uint64_t* out) { *out = (uint64_t)*t; }

template <typename T>
int scan_level(uq_ctx* ctx, const T* d_in, T* d_out, uint64_t n, T* d_total, T* ws) {
    if (n == 0) {
        if (d_total) UQ_CHECK_HIP(hipMemsetAsync(d_total, 0, sizeof(T), ctx->stream));
        return 0;
    }
    uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    UQ_REQUIRE(nb <= 0x7fffffffu, "scan: too many tiles");
    if (nb == 1) {
        scan_tile_kernel<T><<<1, SCAN_THREADS, 0, ctx->stream>>>(d_in, d_out, n, nullptr, d_total);
        UQ_LAUNCH_CHECK();
        return 0;
    }
    T* partials = ws;
    scan_reduce_kernel<T><<<(uint32_t)nb, SCAN_THREADS, 0, ctx->stream>>>(d_in, n, partials);
    UQ_LAUNCH_CHECK();
    UQ_TRY(scan_level<T>(ctx, partials, partials, nb, nullptr, ws + ((nb + 63) & ~uint64_t(63))));
    scan_tile_kernel<T><<<(uint32_t)nb, SCAN_THREADS, 0, ctx->stream>>>(d_in, d_out, n, partials, d_total);
    UQ_LAUNCH_CHECK();
    return 0;
}

size_t scan_ws_elems(uint64_t n) {
    size_t tot = 64;
    while (n > SCAN_TILE) {
        n = (n + SCAN_TILE - 1) / SCAN_TILE;
        tot += (n + 63) & ~uint64_t(63);
    }
    return tot;
}

// The scans keep their partials in a private allocation of the context so that callers can use the
// scratch pool freely.
int get_ws(uq_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->scan_ws_bytes) {
        UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->scan_ws) UQ_CHECK_HIP(hipFree(ctx->scan_ws));
        ctx->scan_ws = nullptr; ctx->scan_ws_bytes = 0;
        size_t want = (bytes + 65535) & ~size_t(65535);
        UQ_CHECK_HIP(hipMalloc(&ctx->scan_ws, want));
        ctx->scan_ws_bytes = want;
    }
    *out = ctx->scan_ws;
    return 0;
}
}  // namespace

int uq_scan_exclusive_u32(uq_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, uint64_t n, uint64_t* d_total) {
    void* ws;
    UQ_TRY(get_ws(ctx, (scan_ws_elems(n) + 2) * sizeof(uint64_t), &ws));
    uint32_t* t32 = (uint32_t*)ws;
    UQ_TRY(scan_level<uint32_t>(ctx, d_in, d_out, n, d_total ? t32 : nullptr, (uint32_t*)ws + 64));
    if (d_total) {
        widen_total_kernel<uint32_t><<<1, 1, 0, ctx->stream>>>(t32, d_total);
        UQ_LAUNCH_CHECK();
    }
    return 0;
}

int uq_scan_exclusive_u64(uq_ctx* ctx, const uint64_t* d_in, uint64_t* d_out, uint64_t n, uint64_t* d_total) {
    void* ws;
    UQ_TRY(get_ws(ctx, (scan_ws_elems(n) + 2) * sizeof(uint64_t), &ws));
    UQ_TRY(scan_level<uint64_t>(ctx, d_in, d_out, n, d_total, (uint64_t*)ws + 64));
    return 0;
}
