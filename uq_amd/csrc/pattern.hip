// pattern.hip -- the eight --pattern byte layouts and their inverses.
// Replaces numpy.rot90 + ascontiguousarray / asfortranarray + the payload write of numpy.save
// (uq.py:263-270) and numpy.load + rot90(-k) on decode (uq.py:943-945).
//
// With T[r][c] an R x C byte table, ro(r) = r or R-1-r and co(c) = c or C-1-c, the eight payloads are
//     row-major     out[ro(r) * C + co(c)] = T[r][c]     0.1 (-,-)  1.2 (-,c)  3.2 (r,-)  2.1 (r,c)
//     column-major  out[co(c) * R + ro(r)] = T[r][c]     0.2 (-,-)  1.1 (-,c)  3.1 (r,-)  2.2 (r,c)
// (byte streams verified against numpy).  ro/co are involutions, so the row-major
// kernel is its own inverse; the column-major family has a forward and an inverse kernel.
//
// All three kernels move a tile of TR table rows through LDS: the row-major side of a tile is one
// contiguous span (16-byte coalesced loads/stores), the column-major side is C runs of TR bytes
// (dword loads/stores behind the run's first aligned address).
// Algorithmic HBM bytes: 2 * R * C.
#include "common.h"
#include "tile_io.h"

namespace {
constexpr int PT_THREADS = TIO_THREADS;

struct PatGeom {
    uint64_t R;
    uint32_t C;
    uint32_t TR;      // rows per tile
    uint32_t TRp;     // (unused)
    uint32_t fr, fc;  // flip rows / flip columns
    uint32_t magicC;  // ceil(2^32 / C)
};

__device__ __forceinline__ uint32_t load_u32_any(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

__device__ __forceinline__ void divmod_c(uint32_t k, const PatGeom& g, uint32_t& q, uint32_t& rem) {
    fast_divmod(k, g.C, g.magicC, q, rem);
}

// ---------------------------------------------------------------- row-major family (self-inverse)
struct RowMajorFn {
    const uint8_t* lds; uint32_t skew, n; const PatGeom& g;
    __device__ __forceinline__ uint8_t at(uint32_t i, uint32_t c) const {
        uint32_t row = g.fr ? n - 1 - i : i;
        uint32_t col = g.fc ? g.C - 1 - c : c;
        return lds[skew + row * g.C + col];
    }
    __device__ __forceinline__ uint8_t byte(uint32_t k) const {
        uint32_t i, c; divmod_c(k, g, i, c);
        return at(i, c);
    }
    __device__ __forceinline__ void operator()(uint32_t k0, uint32_t* w) const {
        uint32_t i, c; divmod_c(k0, g, i, c);
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

__global__ __launch_bounds__(PT_THREADS) void pattern_rm_kernel(const uint8_t* __restrict__ T, PatGeom g, uint8_t* __restrict__ out) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint64_t r0 = (uint64_t)blockIdx.x * g.TR;
    const uint32_t n = (uint32_t)((g.R - r0) < g.TR ? (g.R - r0) : g.TR);
    const uint32_t len = n * g.C;
    const uint32_t skew = stage_span(T + r0 * g.C, len, smem);
    __syncthreads();
    const uint64_t ro0 = g.fr ? g.R - r0 - n : r0;
    RowMajorFn fn{smem, skew, n, g};
    emit_span(out + ro0 * g.C, len, fn);
}

// ---------------------------------------------------------------- column-major family
// The tile lies in LDS the way it lies in the TABLE (row-major, at the 16-byte phase of its address): that side moves with
// 16-byte vectors, conflict-free.  The column runs of the payload are put together (forward) or taken apart (inverse) a dword
// a lane -- four byte accesses at LDS stride C, and a wave's lanes 4 C bytes = C dwords apart: all 64 banks when C is odd (the
// 113-byte quality rows), 32 when C = 2 mod 4 (38-byte DNA rows).  (Before: the LDS image was column-major and the row-major
// side scattered / gathered single bytes whose lanes were 16 column pitches apart -- four banks for 64 lanes; the counters
// showed more than half of the LDS time as bank conflicts, profiles/r02_n_pattern_pmc.txt.)
__global__ __launch_bounds__(PT_THREADS) void pattern_cm_kernel(const uint8_t* __restrict__ T, PatGeom g, uint8_t* __restrict__ out) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint64_t r0 = (uint64_t)blockIdx.x * g.TR;
    const uint32_t n = (uint32_t)((g.R - r0) < g.TR ? (g.R - r0) : g.TR);
    const uint32_t skew = stage_span(T + r0 * g.C, n * g.C, smem);
    __syncthreads();
    const uint64_t ro0 = g.fr ? g.R - r0 - n : r0;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // run position p (0 .. n - 1) holds table row p, or n - 1 - p under a row flip: LDS byte first + p * step
    const int32_t step = g.fr ? -(int32_t)g.C : (int32_t)g.C;
    // (global accesses, unlike LDS ones, take any alignment at nearly full speed -- tools/storebench.hip -- so a run is moved as
    // n / 4 dwords from its first byte, wherever that falls, and n % 4 bytes)
    for (uint32_t c = w; c < g.C; c += PT_THREADS / 64) {
        const uint32_t cc = g.fc ? g.C - 1 - c : c;
        uint8_t* s = out + (uint64_t)cc * g.R + ro0;
        const uint8_t* first = smem + skew + c + (g.fr ? (n - 1) * g.C : 0u);
        const uint32_t nd = n >> 2;
        for (uint32_t d = lane; d < nd; d += 64) {
            const uint8_t* q = first + (int32_t)(4 * d) * step;
            const uint32_t v = (uint32_t)q[0] | ((uint32_t)q[step] << 8) | ((uint32_t)q[2 * step] << 16) | ((uint32_t)q[3 * step] << 24);
            __builtin_memcpy(s + 4 * d, &v, 4);
        }
        if (lane < (n & 3u)) s[4 * nd + lane] = first[(int32_t)(4 * nd + lane) * step];
    }
}

__global__ __launch_bounds__(PT_THREADS) void unpattern_cm_kernel(const uint8_t* __restrict__ P, PatGeom g, uint8_t* __restrict__ T) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint64_t r0 = (uint64_t)blockIdx.x * g.TR;
    const uint32_t n = (uint32_t)((g.R - r0) < g.TR ? (g.R - r0) : g.TR);
    const uint64_t ro0 = g.fr ? g.R - r0 - n : r0;
    uint8_t* dst = T + r0 * g.C;
    const uint32_t skew = (uint32_t)((uintptr_t)dst & 15), len = n * g.C;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int32_t step = g.fr ? -(int32_t)g.C : (int32_t)g.C;
    // a wave takes the columns w, w + 4, ...; the dwords of PU runs are requested before the first is taken apart (a run is
    // two cache lines somewhere in the payload: one request at a time leaves the wave waiting out each latency)
    constexpr int PU = 4;
    auto scatter = [&](uint32_t c, uint32_t d, uint32_t v) {            // the run's positions 4 d .. 4 d + 3
        uint8_t* q = smem + skew + c + (g.fr ? (n - 1) * g.C : 0u) + (int32_t)(4 * d) * step;
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j * step] = (uint8_t)(v >> (8 * j));
    };
    const uint32_t nd = n >> 2;
    for (uint32_t c0 = w; c0 < g.C; c0 += PU * (PT_THREADS / 64)) {
        const uint8_t* s[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const uint32_t c = c0 + u * (PT_THREADS / 64);
            const uint32_t cc = g.fc ? g.C - 1 - c : c;
            s[u] = c < g.C ? P + (uint64_t)cc * g.R + ro0 : nullptr;
        }
        for (uint32_t d = lane; d < nd; d += 64) {
            uint32_t v[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) v[u] = s[u] ? load_u32_any(s[u] + 4 * d) : 0u;
#pragma unroll
            for (int u = 0; u < PU; ++u) if (s[u]) scatter(c0 + u * (PT_THREADS / 64), d, v[u]);
        }
        if (lane < (n & 3u)) {
#pragma unroll
            for (int u = 0; u < PU; ++u)
                if (s[u]) smem[skew + c0 + u * (PT_THREADS / 64) + (g.fr ? (n - 1) * g.C : 0u) + (int32_t)(4 * nd + lane) * step] = s[u][4 * nd + lane];
        }
    }
    __syncthreads();
    // the image mirrors the 16-byte phase of the destination: whole vectors inside the span, bytes at its two ends
    uint8_t* a0 = dst - skew;
    const uint32_t nvec = (skew + len + 15) >> 4;
    for (uint32_t v = threadIdx.x; v < nvec; v += PT_THREADS) {
        const uint32_t b0 = v * 16;
        if (b0 >= skew && b0 + 16 <= skew + len) ((uint4*)a0)[v] = ((const uint4*)smem)[v];
        else for (uint32_t b = b0 < skew ? skew : b0; b < b0 + 16 && b < skew + len; ++b) a0[b] = smem[b];
    }
}

// Rows too wide for an LDS tile (long-read tables): one byte per lane, indexed directly.  Lanes walk the
// TABLE in row-major order, so the table side is coalesced and the payload side is not.
__global__ __launch_bounds__(PT_THREADS) void pattern_wide_kernel(const uint8_t* __restrict__ in, uint64_t R, uint32_t C, uint32_t colmajor,
                                                                  uint32_t fr, uint32_t fc, uint32_t inverse, uint8_t* __restrict__ out) {
    const uint64_t total = R * C;
    for (uint64_t k = (uint64_t)blockIdx.x * PT_THREADS + threadIdx.x; k < total; k += (uint64_t)gridDim.x * PT_THREADS) {
        const uint64_t r = k / C;
        const uint32_t c = (uint32_t)(k - r * C);
        const uint64_t ro = fr ? R - 1 - r : r;
        const uint64_t co = fc ? C - 1 - c : c;
        const uint64_t p = colmajor ? co * R + ro : ro * C + co;       // payload position of T[r][c]
        if (inverse) out[k] = in[p]; else out[p] = in[k];
    }
}

int launch(uq_ctx* ctx, const uint8_t* in, uint64_t rows, uint32_t cols, int pattern_id, uint8_t* out, bool inverse) {
    UQ_REQUIRE(ctx, "null context");
    UQ_REQUIRE(pattern_id >= 0 && pattern_id < 8, "pattern id %d out of range", pattern_id);
    UQ_REQUIRE(cols >= 1, "pattern: table needs at least one column");
    UQ_REQUIRE(rows * (uint64_t)cols == 0 || (in && out), "pattern: null buffer");
    if (rows == 0) return 0;
    const int k = pattern_id >> 1, f = pattern_id & 1;
    // (k, order) -> family and flips, SURVEY.md A.4
    bool colmajor; uint32_t fr, fc;
    switch (k * 2 + f) {
        case 0: colmajor = false; fr = 0; fc = 0; break;   // 0.1
        case 1: colmajor = true;  fr = 0; fc = 0; break;   // 0.2
        case 2: colmajor = true;  fr = 0; fc = 1; break;   // 1.1
        case 3: colmajor = false; fr = 0; fc = 1; break;   // 1.2
        case 4: colmajor = false; fr = 1; fc = 1; break;   // 2.1
        case 5: colmajor = true;  fr = 1; fc = 1; break;   // 2.2
        case 6: colmajor = true;  fr = 1; fc = 0; break;   // 3.1
        default: colmajor = false; fr = 1; fc = 0; break;  // 3.2
    }
    if (!colmajor && !fr && !fc) {
        UQ_CHECK_HIP(hipMemcpyAsync(out, in, rows * cols, hipMemcpyDeviceToDevice, ctx->stream));
        return 0;
    }
    PatGeom g;
    g.R = rows; g.C = cols; g.fr = fr; g.fc = fc;
    g.magicC = magic_u32(cols);
    if (cols > 8192) {                          // fewer than ~8 rows would fit a tile: take the direct kernel
        const uint64_t total = rows * cols;
        const uint64_t nb = (total + PT_THREADS - 1) / PT_THREADS;
        const uint32_t blocks = (uint32_t)(nb < (uint64_t)UQ_NUM_CU * 16 ? nb : (uint64_t)UQ_NUM_CU * 16);
        pattern_wide_kernel<<<blocks, PT_THREADS, 0, ctx->stream>>>(in, rows, cols, colmajor ? 1u : 0u, fr, fc, inverse ? 1u : 0u, out);
        UQ_LAUNCH_CHECK();
        return 0;
    }
    // LDS per tile: small enough for many workgroups per CU (these kernels do not prefetch: occupancy hides the latency).
    // Measured on a 10 M x 113 B table: 60 KiB tiles 1.15 / 0.75 ms (column-major / row-major), 32 KiB 0.71 / 0.51, 16 KiB 0.83 / 0.47.
    // Wide rows keep the larger budget so that a tile still holds a few rows.
    uint32_t budget = (colmajor ? 32u : 16u) * 1024u;
    if (budget / cols < 8) budget = 60 * 1024;
    size_t lds;
    if (!colmajor) {
        uint32_t TR = budget / cols;
        if (TR >= 16) TR &= ~15u;
        if (TR == 0) TR = 1;
        if (TR > 4096) TR = 4096;
        g.TR = TR; g.TRp = 0;
        lds = (size_t)TR * cols + 32;
    } else {
        uint32_t TR = budget / cols;
        if (TR >= 64) TR &= ~63u; else if (TR >= 4) TR &= ~3u;
        if (TR == 0) TR = 1;
        if (TR > 2048) TR = 2048;
        g.TR = TR; g.TRp = 0;
        lds = (size_t)TR * cols + 48;
    }
    UQ_REQUIRE(lds <= 160 * 1024, "pattern: tile needs %zu bytes of LDS", lds);
    const uint64_t tiles = (rows + g.TR - 1) / g.TR;
    UQ_REQUIRE(tiles <= 0x7fffffffu, "pattern: too many tiles");
    const void* fn = !colmajor ? (const void*)pattern_rm_kernel : (inverse ? (const void*)unpattern_cm_kernel : (const void*)pattern_cm_kernel);
    if (lds > 48 * 1024) UQ_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (!colmajor) pattern_rm_kernel<<<(uint32_t)tiles, PT_THREADS, lds, ctx->stream>>>(in, g, out);
    else if (!inverse) pattern_cm_kernel<<<(uint32_t)tiles, PT_THREADS, lds, ctx->stream>>>(in, g, out);
    else unpattern_cm_kernel<<<(uint32_t)tiles, PT_THREADS, lds, ctx->stream>>>(in, g, out);
    UQ_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int uq_pattern(uq_ctx* ctx, const uint8_t* d_table, uint64_t rows, uint32_t cols, int pattern_id, uint8_t* d_payload) {
    return launch(ctx, d_table, rows, cols, pattern_id, d_payload, false);
}

extern "C" int uq_unpattern(uq_ctx* ctx, const uint8_t* d_payload, uint64_t rows, uint32_t cols, int pattern_id, uint8_t* d_table) {
    return launch(ctx, d_payload, rows, cols, pattern_id, d_table, true);
}
