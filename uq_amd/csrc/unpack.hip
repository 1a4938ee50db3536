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
#include "swar.h"
#include "tile_io.h"
#include "codes.h"

namespace {
constexpr int UT = TIO_THREADS;

struct UnpackGeom {
    uint32_t R, bd, bq, Cd, Cq, dmax, variable, G;   // G = 8-symbol groups per read
    uint32_t in_d, in_q, out_s, out_q, lens;         // LDS byte offsets
    uint32_t magicG;
    uint32_t P;                                      // lanes per read (pipelined kernel)
    // lookup-free path: 2-bit bases (base_tab = their four characters), quality character = code + qmin for the nq real
    // codes, at most one quality code that stands for the N-trick base
    uint32_t fast, base_tab, qmin4, q_over, has_n, n_code4, n_char4;
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

// ---- the hot form: persistent workgroups, the next tile's rows in flight in registers while this one is decoded
// (the structure of pack_tile_kernel, mirrored).  Tile = R reads with R * (C_dna + C_qual + 2 dna_max) <= ~31 KiB, so five
// workgroups share a CU.  The lookup-free path covers 2-bit bases with qualities = code + qmin and at most one N code;
// anything else decodes through the LUTs, with the same results.
constexpr int UP_NVD = 2, UP_NVQ = 4;       // 16-byte vectors per lane: DNA rows (<= 8 KiB per tile), QUAL rows (<= 16 KiB)

__global__ __launch_bounds__(UT) void unpack_pipe_kernel(const uint8_t* __restrict__ dna, const uint8_t* __restrict__ qual, uint64_t n,
                                                         UnpackLut lut, UnpackGeom g, uint8_t* __restrict__ seq, uint8_t* __restrict__ qtxt,
                                                         uint32_t* __restrict__ len, unsigned long long* __restrict__ bad) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ uint8_t l_base[256], l_qual[256], l_qn[256];
    const uint32_t tid = threadIdx.x;
    l_base[tid] = lut.base_char[tid]; l_qual[tid] = lut.qual_char[tid]; l_qn[tid] = lut.qual_n_base[tid];
    uint32_t* lens = (uint32_t*)(smem + g.lens);
    uint8_t* o_s = smem + g.out_s;
    uint8_t* o_q = smem + g.out_q;
    const uint64_t R = g.R, ntiles = (n + R - 1) / R, S = gridDim.x;
    struct Regs { uint4 d[UP_NVD], q[UP_NVQ]; uint32_t skd, skq, nvd, nvq, Rt; };
    auto issue = [&](uint64_t tt) {
        Regs x;
        x.skd = x.skq = x.nvd = x.nvq = x.Rt = 0;
#pragma unroll
        for (int u = 0; u < UP_NVD; ++u) x.d[u] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < UP_NVQ; ++u) x.q[u] = make_uint4(0, 0, 0, 0);
        if (tt >= ntiles) return x;
        x.Rt = (uint32_t)((n - tt * R) < R ? (n - tt * R) : R);
        const uint64_t ad = (uint64_t)(uintptr_t)(dna + tt * R * g.Cd), aq = (uint64_t)(uintptr_t)(qual + tt * R * g.Cq);
        x.skd = (uint32_t)(ad & 15); x.skq = (uint32_t)(aq & 15);
        x.nvd = (x.skd + x.Rt * g.Cd + 15) >> 4; x.nvq = (x.skq + x.Rt * g.Cq + 15) >> 4;
        const uint4* sd = (const uint4*)(dna + ((int64_t)(tt * R * g.Cd) - (int64_t)x.skd));          // (global_load, not flat_load: see pack.hip)
        const uint4* sq = (const uint4*)(qual + ((int64_t)(tt * R * g.Cq) - (int64_t)x.skq));
#pragma unroll
        for (int u = 0; u < UP_NVD; ++u) { const uint32_t i = u * UT + tid; if (i < x.nvd) x.d[u] = sd[i]; }
#pragma unroll
        for (int u = 0; u < UP_NVQ; ++u) { const uint32_t i = u * UT + tid; if (i < x.nvq) x.q[u] = sq[i]; }
        return x;
    };
    const uint32_t P = g.P;
    const uint32_t rr = tid / P, pp = tid - rr * P;
    const uint32_t md = (1u << g.bd) - 1;
    const uint64_t mq = (1ull << g.bq) - 1;
    uint64_t t = blockIdx.x;
    Regs cur = issue(t);
    for (; t < ntiles; t += S) {
        const uint64_t r0 = t * R;
        const uint32_t Rt = cur.Rt;
        // ---- A: registers -> LDS
#pragma unroll
        for (int u = 0; u < UP_NVD; ++u) { const uint32_t i = u * UT + tid; if (i < cur.nvd) ((uint4*)(smem + g.in_d))[i] = cur.d[u]; }
#pragma unroll
        for (int u = 0; u < UP_NVQ; ++u) { const uint32_t i = u * UT + tid; if (i < cur.nvq) ((uint4*)(smem + g.in_q))[i] = cur.q[u]; }
        const uint8_t* in_d = smem + g.in_d + cur.skd;
        const uint8_t* in_q = smem + g.in_q + cur.skq;
        __syncthreads();
        cur = issue(t + S);
        if (g.variable) {
            // rows shorter than dna_max leave zeros behind them; read lengths = highest set bit of the DNA row
            for (uint32_t i = tid; i < (Rt * g.dmax + 3) / 4; i += UT) { ((uint32_t*)o_s)[i] = 0; ((uint32_t*)o_q)[i] = 0; }
            for (uint32_t r = tid; r < Rt; r += UT) {
                const uint8_t* row = in_d + r * g.Cd;
                uint32_t k = 0, L;
                while (k < g.Cd && row[k] == 0) ++k;
                if (k == g.Cd) { L = 0; atomicMin(bad, (unsigned long long)(r0 + r)); }
                else {
                    const uint32_t hb = 8 * (g.Cd - 1 - k) + (31 - __clz((uint32_t)row[k]));
                    L = hb / g.bd;
                    if (L * g.bd != hb || L > g.dmax) { atomicMin(bad, (unsigned long long)(r0 + r)); L = L > g.dmax ? g.dmax : L; }
                }
                lens[r] = L;
                len[r0 + r] = L;
            }
            __syncthreads();
        } else {
            for (uint32_t r = tid; r < Rt; r += UT) len[r0 + r] = g.dmax;
        }
        // ---- B: P lanes per read, 8 symbols per step
        if (rr < Rt) {
            const uint32_t r = rr;
            const uint32_t L = g.variable ? lens[r] : g.dmax;
            const uint8_t* drow = in_d + r * g.Cd;
            const uint8_t* qrow = in_q + r * g.Cq;
            uint8_t* ts = o_s + r * g.dmax;
            uint8_t* tq = o_q + r * g.dmax;
            for (uint32_t gg = pp; 8 * gg < L; gg += P) {
                const uint64_t vd = group_bits(drow, g.Cd, g.bd, gg);
                const uint64_t vq = group_bits(qrow, g.Cq, g.bq, gg);
                if (g.fast && 8 * gg + 8 <= L) {
                    // byte k of the two dwords is the character at position L - 8 gg - 8 + k = symbol t = 8 gg + 7 - k
                    const uint32_t v = (uint32_t)vd;
                    const uint32_t clo = ((v >> 14) & 3u) | (((v >> 12) & 3u) << 8) | (((v >> 10) & 3u) << 16) | (((v >> 8) & 3u) << 24);
                    const uint32_t chi = ((v >> 6) & 3u) | (((v >> 4) & 3u) << 8) | (((v >> 2) & 3u) << 16) | ((v & 3u) << 24);
                    uint32_t qlo = 0, qhi = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        qlo |= (uint32_t)((vq >> (g.bq * (7 - k))) & mq) << (8 * k);
                        qhi |= (uint32_t)((vq >> (g.bq * (3 - k))) & mq) << (8 * k);
                    }
                    if ((((qlo + g.q_over) | (qhi + g.q_over)) & 0x80808080u) == 0) {      // all eight are real quality codes
                        uint32_t blo = __builtin_amdgcn_perm(0u, g.base_tab, clo), bhi = __builtin_amdgcn_perm(0u, g.base_tab, chi);
                        if (g.has_n) {
                            const uint32_t mlo = ~nonzero_bytes(qlo ^ g.n_code4), mhi = ~nonzero_bytes(qhi ^ g.n_code4);
                            blo = bfi(mlo, g.n_char4, blo); bhi = bfi(mhi, g.n_char4, bhi);
                        }
                        qlo += g.qmin4; qhi += g.qmin4;
                        uint8_t* ps = ts + (L - 8 * gg - 8);
                        uint8_t* pq = tq + (L - 8 * gg - 8);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            ps[k] = (uint8_t)(blo >> (8 * k)); ps[4 + k] = (uint8_t)(bhi >> (8 * k));
                            pq[k] = (uint8_t)(qlo >> (8 * k)); pq[4 + k] = (uint8_t)(qhi >> (8 * k));
                        }
                        continue;
                    }
                }
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i) {
                    const uint32_t tt = 8 * gg + i;
                    if (tt < L) {
                        const uint32_t cd = (uint32_t)(vd >> (g.bd * i)) & md, cq = (uint32_t)((vq >> (g.bq * i)) & mq);
                        const uint8_t nb = l_qn[cq];
                        ts[L - 1 - tt] = nb ? nb : l_base[cd];
                        tq[L - 1 - tt] = l_qual[cq];
                    }
                }
            }
        }
        __syncthreads();
        // ---- C: text tiles -> HBM (a tile of R = 16 k reads starts 16-byte aligned)
        uint8_t* ds = seq + r0 * g.dmax;
        uint8_t* dq = qtxt + r0 * g.dmax;
        const uint32_t nb = Rt * g.dmax;
        if ((((uintptr_t)ds | (uintptr_t)dq) & 15) == 0) {
            const uint32_t nv = nb >> 4;
            for (uint32_t i = tid; i < nv; i += UT) { ((uint4*)ds)[i] = ((const uint4*)o_s)[i]; ((uint4*)dq)[i] = ((const uint4*)o_q)[i]; }
            for (uint32_t i = (nv << 4) + tid; i < nb; i += UT) { ds[i] = o_s[i]; dq[i] = o_q[i]; }
        } else {
            for (uint32_t i = tid; i < nb; i += UT) { ds[i] = o_s[i]; dq[i] = o_q[i]; }
        }
    }
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
    uint32_t off = 0;
    auto carve = [&](uint32_t bytes) { uint32_t o = off; off += (bytes + 15) & ~15u; return o; };
    {
        // is this alphabet one the lookup-free path decodes?  (anything else: the table path, same results)
        const FastAlphabet a = fast_alphabet(hp);
        g.fast = a.fast && a.bd == 2; g.base_tab = a.base_tab; g.qmin4 = a.qmin4; g.q_over = a.q_over; g.has_n = a.has_n; g.n_code4 = a.n_code4; g.n_char4 = a.n_char4;
    }
    // the pipelined kernel: tiles of R reads within ~31 KiB of LDS (five workgroups per CU) and the register budget
    uint32_t Rp = (31 * 1024) / per_read;
    if (Rp >= 16) Rp &= ~15u;
    if (Rp > UT) Rp = UT;
    if (Rp >= 16 && Rp * g.Cd + 48 <= UP_NVD * UT * 16 && Rp * g.Cq + 48 <= UP_NVQ * UT * 16) {
        g.R = Rp;
        g.P = UT / Rp;
        if (g.P > g.G) g.P = g.G;
        g.in_d = carve(Rp * g.Cd + 32); g.in_q = carve(Rp * g.Cq + 32);
        g.out_s = carve(Rp * g.dmax + 16); g.out_q = carve(Rp * g.dmax + 16);
        g.lens = carve(Rp * 4);
        const size_t lds = off;
        const uint64_t tiles = (nreads + Rp - 1) / Rp;
        uint32_t per_cu = (uint32_t)((156 * 1024) / (lds + 768));
        if (per_cu > 6) per_cu = 6;
        if (per_cu < 1) per_cu = 1;
        const uint32_t blocks = (uint32_t)(tiles < (uint64_t)UQ_NUM_CU * per_cu ? tiles : (uint64_t)UQ_NUM_CU * per_cu);
        unpack_pipe_kernel<<<blocks, UT, lds, ctx->stream>>>(d_dna, d_qual, nreads, lut, g, d_seq, d_qualtxt, d_len, (unsigned long long*)d_bad);
        UQ_LAUNCH_CHECK();
        return 0;
    }
    uint32_t R = (48 * 1024 - 256) / per_read;
    if (R >= 16) R &= ~15u;
    if (R == 0) R = 1;
    if (R > 512) R = 512;
    g.R = R; g.P = 1;
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
