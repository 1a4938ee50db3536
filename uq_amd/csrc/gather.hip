// gather.hip -- out[j] = table[index[j]] for rows of C bytes.
// Replaces table[sort_order] (uq.py:777), key[sort_order] (798), columns_data[idx][sort_order] (822)
// and, on decode, table[key] (953, 957, 973).
//
// A workgroup produces TR consecutive OUTPUT rows (one contiguous span -> 16-byte coalesced stores).
// Each source row is fetched by a group of lanes as the aligned dwords that cover it (rows start at
// arbitrary byte offsets) into an LDS slot with its 0..3 byte skew noted; the emit pass reads bytes
// back in output order.  Small items (1/2/4/8-byte keys and QNAME columns) take a direct typed path.
// Algorithmic HBM bytes per output row: 2 * C + index itemsize.
#include "common.h"
#include "tile_io.h"

namespace {
constexpr int GT = TIO_THREADS;

__device__ __forceinline__ uint64_t load_index(const void* idx, int itemsize, uint64_t j) {
    switch (itemsize) {
        case 1: return ((const uint8_t*)idx)[j];
        case 2: return ((const uint16_t*)idx)[j];
        case 4: return ((const uint32_t*)idx)[j];
        default: return ((const uint64_t*)idx)[j];
    }
}

template <typename T>
__global__ void gather_items_kernel(const T* __restrict__ table, uint64_t table_rows, const void* __restrict__ idx, int itemsize,
                                    uint64_t n, T* __restrict__ out) {
    uint64_t j = (uint64_t)blockIdx.x * GT + threadIdx.x;
    if (j >= n) return;
    uint64_t r = load_index(idx, itemsize, j);
    if (r >= table_rows) r = 0;
    out[j] = table[r];
}

struct GatherGeom {
    uint64_t table_rows, n_out;
    uint32_t C, TR, P;       // P = LDS pitch per row in dwords
    uint32_t G;              // lanes per row group (power of two <= 64)
    uint32_t magicC;
};

struct GatherFn {
    const uint8_t* lds; const uint8_t* skew; const GatherGeom& g;
    __device__ __forceinline__ void divmod(uint32_t k, uint32_t& q, uint32_t& rem) const {
        fast_divmod(k, g.C, g.magicC, q, rem);
    }
    __device__ __forceinline__ uint8_t at(uint32_t i, uint32_t c) const { return lds[i * (g.P * 4) + skew[i] + c]; }
    __device__ __forceinline__ uint8_t byte(uint32_t k) const { uint32_t i, c; divmod(k, i, c); return at(i, c); }
    __device__ __forceinline__ void operator()(uint32_t k0, uint32_t* w) const {
        uint32_t i, c; divmod(k0, i, c);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                v |= (uint32_t)at(i, c) << (8 * b);
                if (++c == g.C) { c = 0; ++i; }
            }
            w[d] = v;
        }
    }
};

// IS = the index's itemsize (an instance each: with the width chosen per load, every index load sat in a branch of its own, followed by its own wait)
template <int IS>
__device__ __forceinline__ uint64_t load_index_t(const void* idx, uint64_t j) {
    if (IS == 1) return ((const uint8_t*)idx)[j];
    if (IS == 2) return ((const uint16_t*)idx)[j];
    if (IS == 4) return ((const uint32_t*)idx)[j];
    return ((const uint64_t*)idx)[j];
}
template <int IS>
__global__ __launch_bounds__(GT) void gather_rows_kernel(const uint8_t* __restrict__ table, const void* __restrict__ idx,
                                                         GatherGeom g, uint8_t* __restrict__ out) {
    extern __shared__ __align__(16) uint8_t smem[];   // [TR][P dwords] + skew[TR]
    uint8_t* skew = smem + (size_t)g.TR * g.P * 4;
    uint32_t* lds32 = (uint32_t*)smem;
    const uint64_t j0 = (uint64_t)blockIdx.x * g.TR;
    const uint32_t n = (uint32_t)((g.n_out - j0) < g.TR ? (g.n_out - j0) : g.TR);
    const uint32_t gl = threadIdx.x & (g.G - 1);          // lane inside the row group
    const uint32_t groups = GT / g.G;
    if (((g.C + 6) >> 2) <= g.G) {
        // a row is at most one dword per lane of its group: four rows per group in flight (index -> row -> LDS is a
        // chain of dependent latencies; one row at a time left the kernel latency-bound)
        constexpr int U = 4;
        for (uint32_t i0 = threadIdx.x / g.G; i0 < n; i0 += U * groups) {
            uint64_t r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = i0 + u * groups;
                r[u] = load_index_t<IS>(idx, j0 + (i < n ? i : n - 1));       // (unconditional: a load behind `i < n` is followed by its own wait -- the four would go one by one)
                if (r[u] >= g.table_rows) r[u] = 0;
            }
            uint32_t v[U], sk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint8_t* s = table + r[u] * g.C;
                sk[u] = (uint32_t)((uintptr_t)s & 3);
                const uint32_t nd = (sk[u] + g.C + 3) >> 2;
                v[u] = ((const uint32_t*)(s - sk[u]))[gl < nd ? gl : 0u];           // (likewise; lanes beyond the row read its first dword and store nothing)
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = i0 + u * groups;
                if (i < n) {
                    if (gl < ((sk[u] + g.C + 3) >> 2)) lds32[i * g.P + gl] = v[u];
                    if (gl == 0) skew[i] = (uint8_t)sk[u];
                }
            }
        }
    } else {
        for (uint32_t i = threadIdx.x / g.G; i < n; i += groups) {
            uint64_t r = load_index_t<IS>(idx, j0 + i);
            if (r >= g.table_rows) r = 0;
            const uint8_t* s = table + r * g.C;
            const uint32_t sk = (uint32_t)((uintptr_t)s & 3);
            const uint32_t* s32 = (const uint32_t*)(s - sk);
            const uint32_t nd = (sk + g.C + 3) >> 2;
            for (uint32_t d = gl; d < nd; d += g.G) lds32[i * g.P + d] = s32[d];
            if (gl == 0) skew[i] = (uint8_t)sk;
        }
    }
    __syncthreads();
    GatherFn fn{smem, skew, g};
    emit_span(out + j0 * g.C, n * g.C, fn);
}
}  // namespace

int uq_gather_rows_internal(uq_ctx* ctx, const uint8_t* d_table, uint64_t table_rows, uint32_t cols, const void* d_index,
                            int index_itemsize, uint64_t n_out, uint8_t* d_out) {
    UQ_REQUIRE(ctx, "null context");
    UQ_REQUIRE(index_itemsize == 1 || index_itemsize == 2 || index_itemsize == 4 || index_itemsize == 8,
               "uq_gather_rows: index itemsize %d not in {1,2,4,8}", index_itemsize);
    UQ_REQUIRE(cols >= 1, "uq_gather_rows: rows need at least one byte");
    if (n_out == 0) return 0;
    UQ_REQUIRE(d_table && d_index && d_out && table_rows > 0, "uq_gather_rows: null / empty table");
    const uint32_t blocks1 = (uint32_t)((n_out + GT - 1) / GT);
    const bool aligned = (((uintptr_t)d_table | (uintptr_t)d_out) & (cols - 1)) == 0;
    if (aligned && cols == 1) { gather_items_kernel<uint8_t><<<blocks1, GT, 0, ctx->stream>>>((const uint8_t*)d_table, table_rows, d_index, index_itemsize, n_out, (uint8_t*)d_out); UQ_LAUNCH_CHECK(); return 0; }
    if (aligned && cols == 2) { gather_items_kernel<uint16_t><<<blocks1, GT, 0, ctx->stream>>>((const uint16_t*)d_table, table_rows, d_index, index_itemsize, n_out, (uint16_t*)d_out); UQ_LAUNCH_CHECK(); return 0; }
    if (aligned && cols == 4) { gather_items_kernel<uint32_t><<<blocks1, GT, 0, ctx->stream>>>((const uint32_t*)d_table, table_rows, d_index, index_itemsize, n_out, (uint32_t*)d_out); UQ_LAUNCH_CHECK(); return 0; }
    if (aligned && cols == 8) { gather_items_kernel<uint64_t><<<blocks1, GT, 0, ctx->stream>>>((const uint64_t*)d_table, table_rows, d_index, index_itemsize, n_out, (uint64_t*)d_out); UQ_LAUNCH_CHECK(); return 0; }

    GatherGeom g;
    g.table_rows = table_rows; g.n_out = n_out; g.C = cols;
    g.magicC = magic_u32(cols);
    const uint32_t nd_max = (cols + 6) >> 2;
    g.P = nd_max | 1;                         // odd pitch: consecutive rows start on different banks
    uint32_t G = 1;
    while (G < nd_max && G < 64) G <<= 1;
    g.G = G;
    UQ_REQUIRE((size_t)g.P * 4 + 1 <= 150 * 1024, "uq_gather_rows: %u-byte rows do not fit one LDS tile", cols);
    // 16 KiB tiles: ~9 workgroups per CU hide the index -> row -> LDS latency chain (measured on 10M x 38 B / 113 B rows:
    // 48 KiB 0.60 / 1.34 ms, 32 KiB 0.50 / 1.03, 16 KiB 0.36 / 0.81, 8 KiB 0.37 / 0.88); wide rows keep >= 16 per tile
    const uint32_t pitch = g.P * 4 + 1;
    uint32_t TR = (16 * 1024) / pitch;
    if (TR < 16) TR = (48 * 1024) / pitch < 16 ? (48 * 1024) / pitch : 16;
    if (TR >= 16) TR &= ~15u;
    if (TR == 0) TR = 1;
    if (TR > 1024) TR = 1024;
    g.TR = TR;
    const size_t lds = (size_t)TR * g.P * 4 + TR + 16;
    const uint64_t tiles = (n_out + TR - 1) / TR;
    UQ_REQUIRE(tiles <= 0x7fffffffu, "uq_gather_rows: too many tiles");
    auto k = index_itemsize == 1 ? gather_rows_kernel<1> : index_itemsize == 2 ? gather_rows_kernel<2> : index_itemsize == 4 ? gather_rows_kernel<4> : gather_rows_kernel<8>;
    if (lds > 48 * 1024) UQ_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k<<<(uint32_t)tiles, GT, lds, ctx->stream>>>(d_table, d_index, g, d_out);
    UQ_LAUNCH_CHECK();
    return 0;
}

namespace {
// first position whose index does not address the table (the reference's numpy raises IndexError there, uq.py:953-973)
__global__ __launch_bounds__(256) void index_range_kernel(const void* __restrict__ idx, int itemsize, uint64_t n, uint64_t limit,
                                                          unsigned long long* __restrict__ first_bad) {
    uint64_t bad = UQ_NONE;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (load_index(idx, itemsize, i) >= limit) { bad = i; break; }          // a lane's positions increase: its first is its lowest
    bad = wave_min(bad);
    if (lane_id() == 0 && bad != UQ_NONE) atomicMin(first_bad, (unsigned long long)bad);
}
}  // namespace

extern "C" int uq_check_index_range(uq_ctx* ctx, const void* d_index, int index_itemsize, uint64_t n, uint64_t limit, uint64_t* h_first_bad) {
    UQ_REQUIRE(ctx && h_first_bad, "uq_check_index_range: null argument");
    UQ_REQUIRE(index_itemsize == 1 || index_itemsize == 2 || index_itemsize == 4 || index_itemsize == 8,
               "uq_check_index_range: index itemsize %d not in {1,2,4,8}", index_itemsize);
    *h_first_bad = UQ_NONE;
    if (n == 0) return 0;
    UQ_REQUIRE(d_index, "uq_check_index_range: null index array");
    void* ws;
    UQ_TRY(uq_scratch(ctx, 8, &ws));
    UQ_CHECK_HIP(hipMemsetAsync(ws, 0xFF, 8, ctx->stream));
    const uint64_t want = (n + 255) / 256;
    index_range_kernel<<<(uint32_t)(want < UQ_NUM_CU * 8ull ? want : UQ_NUM_CU * 8ull), 256, 0, ctx->stream>>>(d_index, index_itemsize, n, limit,
                                                                                                               (unsigned long long*)ws);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_read_back(ctx, ctx->h_pinned, ws, 8));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *h_first_bad = ctx->h_pinned[0];
    return 0;
}

extern "C" int uq_gather_rows(uq_ctx* ctx, const uint8_t* d_table, uint64_t table_rows, uint32_t cols, const void* d_index,
                              int index_itemsize, uint64_t n_out, uint8_t* d_out) {
    return uq_gather_rows_internal(ctx, d_table, table_rows, cols, d_index, index_itemsize, n_out, d_out);
}
