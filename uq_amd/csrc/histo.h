// histo.h -- the (base, quality) pair counter of stats.hip as a reusable piece, for kernels that already hold the
// eight base and eight quality characters of a group in registers (the fused pack + statistics kernel).
// Same three tiers and the same tables as stats_kernel (see stats.hip for the reasoning and the measurements):
//   fast  [A C T G][64 qualities][8 copies]   exact A/C/G/T groups inside the quality window
//   hist  [32 base bytes][64 qualities]       anything inside the two windows
//   global atomics on uq_stats.counts          every other byte pair
#pragma once
#include "common.h"

constexpr uint32_t HZ_NB = 32, HZ_NQ = 64;
constexpr int HZ_COPIES = 8;
constexpr uint32_t HZ_WORDS = HZ_NB * HZ_NQ + 4 * HZ_NQ * HZ_COPIES;     // u32 words of LDS: hist then fast

struct Histo {
    uint32_t* hist;            // LDS [HZ_NB * HZ_NQ]
    uint32_t* fast;            // LDS [4 * HZ_NQ * HZ_COPIES]
    uq_stats* st;
    uint32_t qbase, bbase, q_addlo, q_addhi, b_xor;

    __device__ __forceinline__ void init(uint32_t* lds, uq_stats* stats, uint32_t win) {
        hist = lds; fast = lds + HZ_NB * HZ_NQ; st = stats;
        qbase = win & 255u; bbase = win >> 8;
        q_addlo = 0x01010101u * (0x80u - qbase); q_addhi = 0x01010101u * (0x80u - qbase - HZ_NQ);
        b_xor = 0x01010101u * bbase;
        for (uint32_t i = threadIdx.x; i < HZ_WORDS; i += blockDim.x) lds[i] = 0;
    }
    __device__ __forceinline__ void pair(uint32_t b, uint32_t c) const {
        const uint32_t sb = b - bbase, sq = c - qbase;
        if (sb < HZ_NB && sq < HZ_NQ) atomicAdd(&hist[sb * HZ_NQ + sq], 1u);
        else atomicAdd((unsigned long long*)&st->counts[b * 256 + c], 1ull);
    }
    // bytes k >= from of the 8-byte windows (b_lo | b_hi << 32, q likewise) are real (base, quality) pairs
    __device__ __forceinline__ void group8(uint32_t b_lo, uint32_t b_hi, uint32_t q_lo, uint32_t q_hi, uint32_t from, uint32_t lane) const {
        if (from == 0) {
            const uint32_t sb0 = b_lo ^ b_xor, sb1 = b_hi ^ b_xor;
            const uint32_t u0 = q_lo + q_addlo, u1 = q_hi + q_addlo;
            const uint32_t bad = ((sb0 | sb1) & 0xE0E0E0E0u) |
                                 ((q_lo | (q_lo + q_addhi) | ~u0 | q_hi | (q_hi + q_addhi) | ~u1) & 0x80808080u);
            const uint32_t c0 = (b_lo >> 1) & 0x03030303u, c1 = (b_hi >> 1) & 0x03030303u;
            const bool acgt = __builtin_amdgcn_perm(0u, 0x47544341u, c0) == b_lo && __builtin_amdgcn_perm(0u, 0x47544341u, c1) == b_hi;
            if (bad == 0 && acgt) {
                const uint32_t bin0 = (c0 << 6) | (u0 & 0x7F7F7F7Fu), bin1 = (c1 << 6) | (u1 & 0x7F7F7F7Fu);
                uint8_t* hb = (uint8_t*)fast + ((lane & (HZ_COPIES - 1)) << 2);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    atomicAdd((uint32_t*)(hb + (((bin0 >> (8 * k)) & 0xFFu) << 5)), 1u);
                    atomicAdd((uint32_t*)(hb + (((bin1 >> (8 * k)) & 0xFFu) << 5)), 1u);
                }
                return;
            }
            if (bad == 0) {
                const uint32_t sq0 = (u0 & 0x7F7F7F7Fu) << 2, sq1 = (u1 & 0x7F7F7F7Fu) << 2;
                uint8_t* hb = (uint8_t*)hist;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    atomicAdd((uint32_t*)(hb + ((((sb0 >> (8 * k)) & 0xFFu) << 8) | ((sq0 >> (8 * k)) & 0xFFu))), 1u);
                    atomicAdd((uint32_t*)(hb + ((((sb1 >> (8 * k)) & 0xFFu) << 8) | ((sq1 >> (8 * k)) & 0xFFu))), 1u);
                }
                return;
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            if (k >= from) {
                const uint32_t b = ((k < 4 ? b_lo : b_hi) >> (8 * (k & 3))) & 255u, c = ((k < 4 ? q_lo : q_hi) >> (8 * (k & 3))) & 255u;
                pair(b, c);
            }
        }
    }
    // after a barrier: LDS tables -> the u64 global table
    __device__ __forceinline__ void flush() const {
        for (uint32_t i = threadIdx.x; i < HZ_NB * HZ_NQ; i += blockDim.x) {
            const uint32_t v = hist[i];
            if (v) atomicAdd((unsigned long long*)&st->counts[(bbase + i / HZ_NQ) * 256 + qbase + (i % HZ_NQ)], (unsigned long long)v);
        }
        for (uint32_t bin = threadIdx.x; bin < 4 * HZ_NQ; bin += blockDim.x) {
            uint32_t v = 0;
#pragma unroll
            for (int c = 0; c < HZ_COPIES; ++c) v += fast[bin * HZ_COPIES + ((c + bin) & (HZ_COPIES - 1))];
            const uint32_t base = (0x47544341u >> (8 * (bin >> 6))) & 0xFFu;
            if (v) atomicAdd((unsigned long long*)&st->counts[base * 256 + qbase + (bin & 63u)], (unsigned long long)v);
        }
    }
};

struct RecordAcc {
    uint32_t lmin = 0xFFFFFFFFu, lmax = 0, rmax = 0;
    uint64_t bad_plus = UQ_NONE, bad_len = UQ_NONE;
    __device__ __forceinline__ void record(uint64_t gr, bool plus_ok, uint32_t L, uint32_t Lq, uint32_t rb) {
        if (!plus_ok) bad_plus = bad_plus < gr ? bad_plus : gr;
        if (L != Lq) bad_len = bad_len < gr ? bad_len : gr;
        lmin = L < lmin ? L : lmin;
        lmax = L > lmax ? L : lmax;
        rmax = rb > rmax ? rb : rmax;
    }
    __device__ __forceinline__ void flush(uq_stats* st, uint64_t first) {
        const uint32_t mn = wave_min(lmin), mx = wave_max(lmax), rm = wave_max(rmax);
        const uint64_t bp = wave_min(bad_plus), bl = wave_min(bad_len);
        if (lane_id() == 0) {
            if (mn != 0xFFFFFFFFu) atomicMin(&st->len_min, mn);
            atomicMax(&st->len_max, mx);
            atomicMax(&st->max_record_bytes, rm);
            if (bp != UQ_NONE) atomicMin((unsigned long long*)&st->bad_plus, (unsigned long long)(first + bp));
            if (bl != UQ_NONE) atomicMin((unsigned long long*)&st->bad_len, (unsigned long long)(first + bl));
        }
    }
};
