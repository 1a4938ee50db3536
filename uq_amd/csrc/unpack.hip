// unpack.hip -- packed rows back to characters.
// Replaces the decoder's split_bits + per-symbol character map, N restore and sentinel strip
// (uq.py:1002-1007, 1031-1054).  Works on CODES: for variable-length data the read
// length is the position of the row's highest set bit divided by bits-per-base (the sentinel is the
// only 1 above the payload), not a search for a character.
//
// A workgroup stages the two contiguous row spans of R reads in LDS (16-byte coalesced loads); a
// thread owns 8 consecutive symbols of one read (b_dna + b_qual whole bytes), maps codes through LUTs
// in LDS and writes characters into fixed-pitch text tiles that leave with 16-byte coalesced stores.
// Algorithmic HBM bytes per read: C_dna + C_qual read + 2 * dna_max (+4) written.
#include "common.h"
#include "tile_io.h"

namespace {
constexpr int UT = TIO_THREADS;

struct UnpackLut { uint8_t base_char[256], qual_char[256], qual_n_base[256]; };
struct UnpackGeom {
    uint32_t R, bd, bq, Cd, Cq, dmax, variable, G;   // G = 8-symbol groups per read
    uint32_t in_d, in_q, out_s, out_q, lens;         // LDS byte offsets
    uint32_t magicG;
};

struct LinearFn {
    const uint8_t* lds;
    __device__ __forceinline__ uint8_t byte(uint32_t k) const { return lds[k]; }
    __device__ __forceinline__ void operator()(uint32_t k0, uint32_t* w) const {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= (uint32_t)lds[k0 + 4 * d + b] << (8 * b);
            w[d] = v;
        }
    }
};

__device__ __forceinline__ uint64_t group_bits(const uint8_t* row, uint32_t C, uint32_t b, uint32_t g) {
    // bytes i = b*g .. b*g + b - 1 counted from the row's LAST byte, little-endian into a u64
    uint64_t v = 0;
    for (uint32_t i = 0; i < b; ++i) {
        uint32_t bi = b * g + i;
        if (bi < C) v |= (uint64_t)row[C - 1 - bi] << (8 * i);
    }
    return v;
}

__global__ __launch_bounds__(UT) void unpack_kernel(const uint8_t* __restrict__ dna, const uint8_t* __restrict__ qual, uint64_t n,
                                                    UnpackLut lut, UnpackGeom g, uint8_t* __restrict__ seq, uint8_t* __restrict__ qtxt,
                                                    uint32_t* __restrict__ len, unsigned long long* __restrict__ bad) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ uint8_t l_base[256], l_qual[256], l_qn[256];
    const uint32_t tid = threadIdx.x;
    l_base[tid] = lut.base_char[tid]; l_qual[tid] = lut.qual_char[tid]; l_qn[tid] = lut.qual_n_base[tid];
    const uint64_t r0 = (uint64_t)blockIdx.x * g.R;
    const uint32_t Rt = (uint32_t)((n - r0) < g.R ? (n - r0) : g.R);
    const uint32_t skd = stage_span(dna + r0 * g.Cd, Rt * g.Cd, smem + g.in_d);
    const uint32_t skq = stage_span(qual + r0 * g.Cq, Rt * g.Cq, smem + g.in_q);
    uint32_t* lens = (uint32_t*)(smem + g.lens);
    uint8_t* o_s = smem + g.out_s;
    uint8_t* o_q = smem + g.out_q;
    for (uint32_t i = tid; i < (Rt * g.dmax + 3) / 4; i += UT) { ((uint32_t*)o_s)[i] = 0; ((uint32_t*)o_q)[i] = 0; }
    __syncthreads();
    const uint8_t* in_d = smem + g.in_d + skd;
    const uint8_t* in_q = smem + g.in_q + skq;
    // read lengths
    for (uint32_t r = tid; r < Rt; r += UT) {
        uint32_t L = g.dmax;
        if (g.variable) {
            const uint8_t* row = in_d + r * g.Cd;
            uint32_t k = 0;
            while (k < g.Cd && row[k] == 0) ++k;
            if (k == g.Cd) { L = 0; atomicMin(bad, (unsigned long long)(r0 + r)); }
            else {
                uint32_t hb = 8 * (g.Cd - 1 - k) + (31 - __clz((uint32_t)row[k]));
                L = hb / g.bd;
                if (L * g.bd != hb || L > g.dmax) { atomicMin(bad, (unsigned long long)(r0 + r)); L = L > g.dmax ? g.dmax : L; }
            }
        }
        lens[r] = L;
        len[r0 + r] = L;
    }
    __syncthreads();
    const uint32_t items = Rt * g.G;
    const uint64_t md = (1ull << g.bd) - 1, mq = (1ull << g.bq) - 1;
    for (uint32_t idx = tid; idx < items; idx += UT) {
        uint32_t r, gg;
        fast_divmod(idx, g.G, g.magicG, r, gg);
        const uint32_t L = lens[r];
        if (8 * gg >= L) continue;
        const uint64_t vd = group_bits(in_d + r * g.Cd, g.Cd, g.bd, gg);
        const uint64_t vq = group_bits(in_q + r * g.Cq, g.Cq, g.bq, gg);
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) {
            const uint32_t t = 8 * gg + i;
            if (t < L) {
                const uint32_t cd = (uint32_t)((vd >> (g.bd * i)) & md), cq = (uint32_t)((vq >> (g.bq * i)) & mq);
                const uint8_t nb = l_qn[cq];
                const uint32_t j = L - 1 - t;
                o_s[r * g.dmax + j] = nb ? nb : l_base[cd];
                o_q[r * g.dmax + j] = l_qual[cq];
            }
        }
    }
    __syncthreads();
    LinearFn fs{o_s}, fq{o_q};
    emit_span(seq + r0 * g.dmax, Rt * g.dmax, fs);
    emit_span(qtxt + r0 * g.dmax, Rt * g.dmax, fq);
}

// Rows too long for one LDS tile (reads beyond ~50 kbp): one wave per read, symbols straight from / to HBM.
__global__ __launch_bounds__(UT) void unpack_long_kernel(const uint8_t* __restrict__ dna, const uint8_t* __restrict__ qual, uint64_t n,
                                                         UnpackLut lut, UnpackGeom g, uint8_t* __restrict__ seq, uint8_t* __restrict__ qtxt,
                                                         uint32_t* __restrict__ len, unsigned long long* __restrict__ bad) {
    __shared__ uint8_t l_base[256], l_qual[256], l_qn[256];
    const uint32_t tid = threadIdx.x, lane = lane_id();
    l_base[tid] = lut.base_char[tid]; l_qual[tid] = lut.qual_char[tid]; l_qn[tid] = lut.qual_n_base[tid];
    __syncthreads();
    const uint32_t md = (1u << g.bd) - 1, mq = (1u << g.bq) - 1;
    for (uint64_t r = (uint64_t)blockIdx.x * (UT / 64) + (tid >> 6); r < n; r += (uint64_t)gridDim.x * (UT / 64)) {
        const uint8_t* drow = dna + r * g.Cd;
        const uint8_t* qrow = qual + r * g.Cq;
        uint32_t L = g.dmax;
        if (g.variable) {
            // first non-zero byte of the row: lanes scan strided, the lowest hit wins
            uint32_t first = 0xFFFFFFFFu;
            for (uint32_t k = lane; k < g.Cd && first == 0xFFFFFFFFu; k += 64) if (drow[k]) first = k;
            first = wave_min(first);
            bool ok = first != 0xFFFFFFFFu;
            if (ok) {
                const uint32_t hb = 8 * (g.Cd - 1 - first) + (31 - __clz((uint32_t)drow[first]));
                L = hb / g.bd;
                if (L * g.bd != hb || L > g.dmax) { ok = false; L = L > g.dmax ? g.dmax : L; }
            } else L = 0;
            if (!ok && lane == 0) atomicMin(bad, (unsigned long long)r);
        }
        if (lane == 0) len[r] = L;
        uint8_t* so = seq + r * g.dmax;
        uint8_t* qo = qtxt + r * g.dmax;
        for (uint32_t j = lane; j < g.dmax; j += 64) {
            uint8_t cb = 0, cc = 0;
            if (j < L) {
                const uint32_t t = L - 1 - j;                        // symbol index from the end of the read
                const uint32_t bitd = t * g.bd, bitq = t * g.bq;
                const uint32_t bd0 = g.Cd - 1 - (bitd >> 3), bq0 = g.Cq - 1 - (bitq >> 3);
                const uint32_t vd = drow[bd0] | (bd0 ? (uint32_t)drow[bd0 - 1] << 8 : 0u);
                const uint32_t vq = qrow[bq0] | (bq0 ? (uint32_t)qrow[bq0 - 1] << 8 : 0u);
                const uint32_t cd = (vd >> (bitd & 7)) & md, cq = (vq >> (bitq & 7)) & mq;
                const uint8_t nb = l_qn[cq];
                cb = nb ? nb : l_base[cd];
                cc = l_qual[cq];
            }
            so[j] = cb; qo[j] = cc;
        }
    }
}
}  // namespace

extern "C" int uq_unpack(uq_ctx* ctx, const uint8_t* d_dna, const uint8_t* d_qual, uint64_t nreads, const uq_unpack_params* hp,
                         uint8_t* d_seq, uint8_t* d_qualtxt, uint32_t* d_len, uint64_t* d_bad) {
    UQ_REQUIRE(ctx && hp && d_bad, "uq_unpack: null argument");
    UQ_CHECK_HIP(hipMemsetAsync(d_bad, 0xFF, 8, ctx->stream));
    if (nreads == 0) return 0;
    UQ_REQUIRE(d_dna && d_qual && d_seq && d_qualtxt && d_len, "uq_unpack: null buffer");
    UQ_REQUIRE(hp->bits_per_base >= 1 && hp->bits_per_base <= 8 && hp->bits_per_quality >= 1 && hp->bits_per_quality <= 8,
               "uq_unpack: bits per symbol must be 1..8");
    UQ_REQUIRE(hp->dna_max >= 1, "uq_unpack: dna_max must be positive");
    UnpackGeom g;
    g.bd = hp->bits_per_base; g.bq = hp->bits_per_quality; g.Cd = hp->dna_bytes_per_row; g.Cq = hp->quality_bytes_per_row;
    g.dmax = hp->dna_max; g.variable = hp->variable ? 1 : 0;
    const uint32_t Lv = g.dmax + g.variable;
    UQ_REQUIRE(g.Cd == (g.bd * Lv + 7) / 8 && g.Cq == (g.bq * Lv + 7) / 8, "uq_unpack: row bytes do not match the geometry");
    g.G = (g.dmax + 7) / 8;
    g.magicG = magic_u32(g.G);
    const uint32_t per_read = g.Cd + g.Cq + 2 * g.dmax + 4;
    UnpackLut lut;
    memcpy(lut.base_char, hp->base_char, 256); memcpy(lut.qual_char, hp->qual_char, 256); memcpy(lut.qual_n_base, hp->qual_n_base, 256);
    if (per_read + 256 > 150 * 1024) {            // rows beyond one LDS tile: wave per read
        g.R = 0; g.in_d = g.in_q = g.out_s = g.out_q = g.lens = 0;
        const uint64_t waves = (nreads + 3) / 4;
        const uint32_t blocks = (uint32_t)(waves < (uint64_t)UQ_NUM_CU * 8 ? waves : (uint64_t)UQ_NUM_CU * 8);
        unpack_long_kernel<<<blocks, UT, 0, ctx->stream>>>(d_dna, d_qual, nreads, lut, g, d_seq, d_qualtxt, d_len, (unsigned long long*)d_bad);
        UQ_LAUNCH_CHECK();
        return 0;
    }
    uint32_t R = (48 * 1024 - 256) / per_read;
    if (R >= 16) R &= ~15u;
    if (R == 0) R = 1;
    if (R > 512) R = 512;
    g.R = R;
    uint32_t off = 0;
    auto carve = [&](uint32_t bytes) { uint32_t o = off; off += (bytes + 15) & ~15u; return o; };
    g.in_d = carve(R * g.Cd + 32); g.in_q = carve(R * g.Cq + 32);
    g.out_s = carve(R * g.dmax + 16); g.out_q = carve(R * g.dmax + 16);
    g.lens = carve(R * 4);
    const size_t lds = off;
    const uint64_t tiles = (nreads + R - 1) / R;
    UQ_REQUIRE(tiles <= 0x7fffffffu, "uq_unpack: too many tiles");
    if (lds > 48 * 1024) UQ_CHECK_HIP(hipFuncSetAttribute((const void*)unpack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unpack_kernel<<<(uint32_t)tiles, UT, lds, ctx->stream>>>(d_dna, d_qual, nreads, lut, g, d_seq, d_qualtxt, d_len, (unsigned long long*)d_bad);
    UQ_LAUNCH_CHECK();
    return 0;
}
