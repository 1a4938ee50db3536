// swar.h -- byte-parallel (SIMD-within-a-register) helpers shared by the pack and stats kernels.
#pragma once
#include "common.h"

// High bit of each byte of the result is set iff that byte of v equals '\n' (exact, no carries between bytes).
__device__ __forceinline__ uint32_t nl_bits(uint32_t v) {
    v ^= 0x0A0A0A0Au;
    uint32_t t = (v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | v | 0x7F7F7F7Fu);
}
// 16-bit mask (bit i = byte i of the 16-byte vector is '\n').
__device__ __forceinline__ uint32_t nl_mask16(uint4 q) {
    auto nib = [](uint32_t w) { return (((nl_bits(w) >> 7) & 0x01010101u) * 0x01020408u >> 24) & 0xFu; };
    return nib(q.x) | (nib(q.y) << 4) | (nib(q.z) << 8) | (nib(q.w) << 12);
}
// Bytes of the vector at stream position p .. p+15 that fall inside [0, nbytes) (p may be negative).
__device__ __forceinline__ uint32_t valid_mask16(int64_t p, uint64_t nbytes) {
    if (p >= 0 && (uint64_t)p + 16 <= nbytes) return 0xFFFFu;
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int64_t q = p + i;
        if (q >= 0 && (uint64_t)q < nbytes) m |= 1u << i;
    }
    return m;
}

// 8 consecutive bytes at LDS byte offset `o` (any alignment, may be slightly negative) as two dwords:
// three aligned ds_read_b32 + two v_alignbyte.  (gfx950 accepts an unaligned ds_read_b64 here, but it is
// much slower: pack 1.11 -> 1.62 ms, stats 0.93 -> 1.24 ms on 10M x 150 bp -- measured, reverted.)
__device__ __forceinline__ void lds_window8(const uint8_t* base, int32_t o, uint32_t& lo, uint32_t& hi) {
    const uint32_t* p = (const uint32_t*)(base + (o & ~3));
    const uint32_t a = p[0], b = p[1], c = p[2];
    const uint32_t sh = (uint32_t)o & 3u;
    lo = __builtin_amdgcn_alignbyte(b, a, sh);
    hi = __builtin_amdgcn_alignbyte(c, b, sh);
}
// the same with the offset already split into its dword-aligned part and its byte phase (loops whose windows move by multiples of four bytes)
__device__ __forceinline__ void lds_window8_at(const uint8_t* base, int32_t aligned, uint32_t sh, uint32_t& lo, uint32_t& hi) {
    const uint32_t* p = (const uint32_t*)(base + aligned);
    const uint32_t a = p[0], b = p[1], c = p[2];
    lo = __builtin_amdgcn_alignbyte(b, a, sh);
    hi = __builtin_amdgcn_alignbyte(c, b, sh);
}
// bytes k < nbad of the 8-byte window are above the first base: byte masks of the bytes to KEEP
__device__ __forceinline__ void window_masks(uint32_t nbad, uint32_t& mlo, uint32_t& mhi) {
    mlo = nbad >= 4 ? 0u : (0xFFFFFFFFu << (8 * nbad));
    mhi = nbad >= 8 ? 0u : (nbad > 4 ? (0xFFFFFFFFu << (8 * (nbad - 4))) : 0xFFFFFFFFu);
}
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }
// 0xFF in every byte of x that is non-zero
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x) {
    const uint32_t nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
    return (nz - (nz >> 7)) | nz;
}
