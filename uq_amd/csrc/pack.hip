// pack.hip -- the per-read DNA / QUAL bit packers.
// Replaces `encoder_fixed` (uq.py:108-182) and `encoder_variable` (uq.py:188-254).
//
// Row semantics: value = sum_j code(read[j]) << (b * (L-1-j)), plus, for
// variable-length input, a sentinel 1 << (b * L); stored big-endian and right-aligned in C bytes.
// A base byte that is not in `bases` takes DNA code 0 and the quality code N_qual[byte] (uq.py:151-153).
//
// Kernel `pack_tile_kernel` (the hot one): one workgroup packs a tile of R consecutive reads.
//   A  the tile's records are one contiguous byte span of the FASTQ stream: it is copied to LDS with
//      fully coalesced 16-byte loads (the QNAME and '+' lines ride along; they share cache lines).
//   B  every (read, position) becomes a code byte through LUTs held in LDS; codes are written in
//      REVERSED order (index t = L-1-j), the sentinel is simply code 1 at t = L, zeros above.
//   C  a thread owns 8 consecutive codes of one read (one aligned ds_read_b64): 8*b bits = exactly b
//      whole output bytes, so no thread ever shares a byte with another -- no atomics, no ballots.
//   D  the packed tile is a contiguous span of the output table: stored with 16-byte coalesced stores.
// Algorithmic HBM bytes per read: record bytes read + C_dna + C_qual written (+32 B of line offsets).
//
// Kernel `pack_carry_kernel`: the Q9 corner (an N quality code equal to 2^b, uq.py:493-494) makes
// the reference's `+=` carry into the neighbouring symbol; that case is packed by a plain
// thread-per-read big-integer addition so the bytes still match.
#include "common.h"

namespace {
constexpr int PK_THREADS = 256;

struct PackLut {
    int16_t dna_code[256];
    int16_t qual_code[256];
    int16_t n_qual[256];
};

struct PackGeom {
    uint32_t R;            // reads per tile
    uint32_t Lp;           // code bytes per read in LDS (multiple of 8)
    uint32_t bd, bq;       // bits per base / quality
    uint32_t Cd, Cq;       // bytes per row
    uint32_t Gd, Gq;       // 8-code groups per row
    uint32_t variable;
    uint32_t stage_bytes;  // size of the staging region (multiple of 16)
    uint32_t magicG;       // ceil(2^32 / (Gd + Gq))
    uint32_t dna_max;      // longest read the geometry was sized for
};

// LDS carve (dynamic): [stage | out_d out_q alias stage] [codes_d] [codes_q] [meta u32 x (4R+1)] [luts]
__global__ __launch_bounds__(PK_THREADS) void pack_tile_kernel(const uint8_t* __restrict__ buf,
                                                               const uint64_t* __restrict__ ls, uint64_t first,
                                                               uint64_t n, PackLut lut, PackGeom g,
                                                               uint8_t* __restrict__ dna, uint8_t* __restrict__ qual,
                                                               unsigned long long* __restrict__ bad) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* stage = smem;
    uint8_t* codes_d = stage + g.stage_bytes;
    uint8_t* codes_q = codes_d + g.R * g.Lp;
    uint32_t* meta = (uint32_t*)(codes_q + g.R * g.Lp);
    int16_t* l_dna = (int16_t*)(meta + 4 * g.R + 4);
    int16_t* l_qual = l_dna + 256;
    int16_t* l_nq = l_qual + 256;

    const uint32_t tid = threadIdx.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * g.R;
    const uint32_t Rt = (uint32_t)((n - r0) < g.R ? (n - r0) : g.R);   // reads in this tile

    l_dna[tid] = lut.dna_code[tid];
    l_qual[tid] = lut.qual_code[tid];
    l_nq[tid] = lut.n_qual[tid];

    // ---- A: stage the span
    const uint64_t* lsp = ls + 4 * (first + r0);
    const uint64_t g0 = lsp[0];
    const uint64_t g1 = lsp[4 * Rt];
    const uint64_t a0 = ((uint64_t)(uintptr_t)buf + g0) & ~uint64_t(15);   // absolute, 16-aligned
    const uint32_t skew = (uint32_t)(((uint64_t)(uintptr_t)buf + g0) - a0);
    const uint32_t nvec = (uint32_t)((g1 - g0 + skew + 15) >> 4);
    if ((uint64_t)nvec * 16 > g.stage_bytes) {   // a record longer than the caller's max_record_bytes
        if (tid == 0) atomicMin(bad, (unsigned long long)r0);
        return;
    }
    const uint4* src = (const uint4*)(uintptr_t)a0;
    uint4* dst = (uint4*)stage;
    for (uint32_t i = tid; i < nvec; i += PK_THREADS) dst[i] = src[i];
    for (uint32_t i = tid; i <= 4 * Rt; i += PK_THREADS) meta[i] = (uint32_t)(lsp[i] - g0) + skew;
    __syncthreads();

    // ---- B: characters -> codes (reversed), sentinel, zero padding
    const uint32_t w = tid >> 6, lane = tid & 63;
    uint32_t badr = 0xFFFFFFFFu;
    for (uint32_t r = w; r < Rt; r += PK_THREADS / 64) {
        const uint32_t so = meta[4 * r + 1];
        uint32_t L = meta[4 * r + 2] - so - 1;
        const uint32_t qo = meta[4 * r + 3];
        if (L > g.dna_max || meta[4 * r + 4] - qo - 1 != L) { badr = r; L = 0; }
        uint8_t* cd = codes_d + r * g.Lp;
        uint8_t* cq = codes_q + r * g.Lp;
        for (uint32_t j = lane; j < g.Lp; j += 64) {
            if (j < L) {
                uint32_t cb = stage[so + j], cc = stage[qo + j];
                int dc = l_dna[cb], qc = l_qual[cc];
                if (dc < 0) { dc = 0; qc = l_nq[cb]; }
                if (qc < 0) { badr = r; qc = 0; }
                uint32_t t = L - 1 - j;
                cd[t] = (uint8_t)dc;
                cq[t] = (uint8_t)qc;
            } else {
                uint8_t v = (uint8_t)((j == L) ? g.variable : 0);
                cd[j] = v;
                cq[j] = v;
            }
        }
    }
    if (badr != 0xFFFFFFFFu) atomicMin(bad, (unsigned long long)(r0 + badr));
    __syncthreads();

    // ---- C: 8 codes -> b bytes.  Output tiles alias the staging region (no longer needed).
    uint8_t* out_d = stage;
    uint8_t* out_q = stage + ((Rt * g.Cd + 15) & ~15u);
    const uint32_t Gt = g.Gd + g.Gq;
    const uint32_t items = Rt * Gt;
    for (uint32_t idx = tid; idx < items; idx += PK_THREADS) {
        uint32_t r, gg;
        fast_divmod(idx, Gt, g.magicG, r, gg);
        const bool isq = gg >= g.Gd;
        if (isq) gg -= g.Gd;
        const uint32_t b = isq ? g.bq : g.bd, C = isq ? g.Cq : g.Cd;
        const uint8_t* cp = (isq ? codes_q : codes_d) + r * g.Lp + 8 * gg;
        const uint64_t c8 = *(const uint64_t*)cp;
        uint64_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v |= ((c8 >> (8 * i)) & 0xFFull) << (b * i);
        uint8_t* orow = (isq ? out_q : out_d) + r * C;
        const uint32_t i0 = b * gg;                       // byte index counted from the row's LAST byte
        for (uint32_t i = 0; i < b; ++i) {
            uint32_t bi = i0 + i;
            if (bi < C) orow[C - 1 - bi] = (uint8_t)(v >> (8 * i));
        }
    }
    __syncthreads();

    // ---- D: coalesced stores of the two packed tiles
    {
        const uint64_t nb = (uint64_t)Rt * g.Cd;
        uint8_t* gdst = dna + r0 * g.Cd;
        if ((((uintptr_t)gdst) & 15) == 0) {
            const uint32_t nv = (uint32_t)(nb >> 4);
            for (uint32_t i = tid; i < nv; i += PK_THREADS) ((uint4*)gdst)[i] = ((const uint4*)out_d)[i];
            for (uint32_t i = (nv << 4) + tid; i < nb; i += PK_THREADS) gdst[i] = out_d[i];
        } else {
            for (uint32_t i = tid; i < nb; i += PK_THREADS) gdst[i] = out_d[i];
        }
    }
    {
        const uint64_t nb = (uint64_t)Rt * g.Cq;
        uint8_t* gdst = qual + r0 * g.Cq;
        if ((((uintptr_t)gdst) & 15) == 0) {
            const uint32_t nv = (uint32_t)(nb >> 4);
            for (uint32_t i = tid; i < nv; i += PK_THREADS) ((uint4*)gdst)[i] = ((const uint4*)out_q)[i];
            for (uint32_t i = (nv << 4) + tid; i < nb; i += PK_THREADS) gdst[i] = out_q[i];
        } else {
            for (uint32_t i = tid; i < nb; i += PK_THREADS) gdst[i] = out_q[i];
        }
    }
}

// Exact big-integer form: thread per read, byte-serial addition with carry, straight from HBM.
__global__ __launch_bounds__(256) void pack_carry_kernel(const uint8_t* __restrict__ buf, const uint64_t* __restrict__ ls,
                                                         uint64_t first, uint64_t n, PackLut lut, uint32_t bd, uint32_t bq,
                                                         uint32_t Cd, uint32_t Cq, uint32_t variable,
                                                         uint8_t* __restrict__ dna, uint8_t* __restrict__ qual,
                                                         unsigned long long* __restrict__ bad) {
    __shared__ int16_t l_dna[256], l_qual[256], l_nq[256];
    l_dna[threadIdx.x] = lut.dna_code[threadIdx.x];
    l_qual[threadIdx.x] = lut.qual_code[threadIdx.x];
    l_nq[threadIdx.x] = lut.n_qual[threadIdx.x];
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint64_t* p = ls + 4 * (first + r);
    const uint64_t s = p[1], q = p[3];
    const uint32_t L = (uint32_t)(p[2] - s - 1);
    uint8_t* drow = dna + r * Cd;
    uint8_t* qrow = qual + r * Cq;
    // accumulate from the last base (least significant) upward, as uq.py:147-167 does
    uint32_t td = 0, tq = 0, bdn = 0, bqn = 0;
    int32_t pd = (int32_t)Cd - 1, pq = (int32_t)Cq - 1;
    for (uint32_t t = 0; t < L; ++t) {
        uint32_t cb = buf[s + L - 1 - t], cc = buf[q + L - 1 - t];
        int dc = l_dna[cb], qc = l_qual[cc];
        if (dc < 0) { dc = 0; qc = l_nq[cb]; }
        if (qc < 0) { atomicMin(bad, (unsigned long long)r); qc = 0; }
        td += (uint32_t)dc << bdn; tq += (uint32_t)qc << bqn;
        bdn += bd; bqn += bq;
        while (bdn > 8) { bdn -= 8; if (pd >= 0) drow[pd] = (uint8_t)td; td >>= 8; --pd; }
        while (bqn > 8) { bqn -= 8; if (pq >= 0) qrow[pq] = (uint8_t)tq; tq >>= 8; --pq; }
    }
    td += variable << bdn; tq += variable << bqn;
    while (pd >= 0) { drow[pd] = (uint8_t)td; td >>= 8; --pd; }
    while (pq >= 0) { qrow[pq] = (uint8_t)tq; tq >>= 8; --pq; }
}
}  // namespace

extern "C" int uq_pack(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t first_read,
                       uint64_t nreads, const uq_pack_params* hp, uint8_t* d_dna, uint8_t* d_qual, uint64_t* d_bad) {
    UQ_REQUIRE(ctx && d_buf && d_line_start && hp && d_dna && d_qual && d_bad, "uq_pack: null argument");
    UQ_REQUIRE(hp->bits_per_base >= 1 && hp->bits_per_base <= 8 && hp->bits_per_quality >= 1 && hp->bits_per_quality <= 8,
               "uq_pack: bits per symbol must be 1..8");
    const uint32_t bd = hp->bits_per_base, bq = hp->bits_per_quality;
    const uint32_t Cd = hp->dna_bytes_per_row, Cq = hp->quality_bytes_per_row;
    const uint32_t Lv = hp->dna_max + (hp->variable ? 1 : 0);
    UQ_REQUIRE(Cd == (bd * Lv + 7) / 8 && Cq == (bq * Lv + 7) / 8,
               "uq_pack: row bytes (%u, %u) do not match ceil(bits * (dna_max + variable) / 8)", Cd, Cq);
    UQ_CHECK_HIP(hipMemsetAsync(d_bad, 0xFF, 8, ctx->stream));
    if (nreads == 0) return 0;

    PackLut lut;
    int max_d = 0, max_q = 0;
    for (int i = 0; i < 256; ++i) {
        lut.dna_code[i] = hp->dna_code[i];
        lut.qual_code[i] = hp->qual_code[i];
        lut.n_qual[i] = (int16_t)(hp->n_qual[i] > 32767 ? 32767 : hp->n_qual[i]);
        if (hp->dna_code[i] > max_d) max_d = hp->dna_code[i];
        if (hp->qual_code[i] > max_q) max_q = hp->qual_code[i];
        if (hp->dna_code[i] < 0 && hp->n_qual[i] > max_q) max_q = hp->n_qual[i];
    }
    UQ_REQUIRE(max_d < (1 << bd), "uq_pack: a DNA code does not fit %u bits", bd);
    const bool carry = max_q >= (1 << bq);   // Q9: N quality code == 2^b (or beyond)

    if (carry) {
        uint32_t blocks = (uint32_t)((nreads + 255) / 256);
        pack_carry_kernel<<<blocks, 256, 0, ctx->stream>>>(d_buf, d_line_start, first_read, nreads, lut, bd, bq, Cd, Cq,
                                                           hp->variable ? 1u : 0u, d_dna, d_qual, (unsigned long long*)d_bad);
        UQ_LAUNCH_CHECK();
        return 0;
    }

    PackGeom g;
    g.bd = bd; g.bq = bq; g.Cd = Cd; g.Cq = Cq; g.variable = hp->variable ? 1 : 0;
    g.Gd = (Cd + bd - 1) / bd; g.Gq = (Cq + bq - 1) / bq;
    g.Lp = 8 * (g.Gd > g.Gq ? g.Gd : g.Gq);
    if (g.Lp < ((Lv + 7) & ~7u)) g.Lp = (Lv + 7) & ~7u;
    g.dna_max = (uint32_t)hp->dna_max;
    g.magicG = magic_u32(g.Gd + g.Gq);
    const uint32_t rec = (uint32_t)hp->max_record_bytes;
    UQ_REQUIRE(rec >= 4, "uq_pack: max_record_bytes not set (take it from uq_stats)");
    // LDS budget: aim for ~3 workgroups per CU.
    const uint32_t budget = 48 * 1024;
    const uint32_t fixed = 3 * 512 + 64;
    const uint32_t per_read = rec + 2 * g.Lp + 16;
    UQ_REQUIRE(per_read + fixed + 64 <= 150 * 1024, "uq_pack: a %u-byte record does not fit one LDS tile", rec);
    uint32_t R = (budget - fixed - 64) / per_read;
    if (R >= 16) R &= ~15u;            // keeps every tile's output offset 16-byte aligned
    if (R > 256) R = 256;
    if (R == 0) R = 1;
    g.R = R;
    uint32_t stage = R * rec + 32;
    uint32_t outb = ((R * Cd + 15) & ~15u) + R * Cq + 16;
    if (outb > stage) stage = outb;
    g.stage_bytes = (stage + 15) & ~15u;
    const size_t lds = (size_t)g.stage_bytes + 2 * (size_t)R * g.Lp + (4 * R + 4) * 4 + 3 * 512;
    UQ_REQUIRE(lds <= 160 * 1024, "uq_pack: tile needs %zu bytes of LDS", lds);
    const uint64_t tiles = (nreads + R - 1) / R;
    UQ_REQUIRE(tiles <= 0x7fffffffu, "uq_pack: too many tiles");
    if (lds > 48 * 1024)
        UQ_CHECK_HIP(hipFuncSetAttribute((const void*)pack_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    pack_tile_kernel<<<(uint32_t)tiles, PK_THREADS, lds, ctx->stream>>>(d_buf, d_line_start, first_read, nreads, lut, g, d_dna,
                                                                        d_qual, (unsigned long long*)d_bad);
    UQ_LAUNCH_CHECK();
    return 0;
}
