// tile_io.h -- moving byte spans between HBM and LDS with 16-byte coalesced accesses, for kernels whose
// tiles start at arbitrary byte offsets (row widths like 38 or 113 bytes are not multiples of 16).
#pragma once
#include "common.h"

constexpr int TIO_THREADS = 256;
#define PT_THREADS TIO_THREADS

// Load the byte span [src, src + len) into lds (16-byte aligned chunks); byte i lands at lds[skew + i].
__device__ __forceinline__ uint32_t stage_span(const uint8_t* src, uint32_t len, uint8_t* lds) {
    const uint64_t a0 = (uint64_t)(uintptr_t)src & ~uint64_t(15);
    const uint32_t skew = (uint32_t)((uint64_t)(uintptr_t)src - a0);
    const uint32_t nvec = (skew + len + 15) >> 4;
    const uint4* s4 = (const uint4*)(src - skew);         // (keeps the pointer's address space: global_load, not flat_load)
    uint4* d4 = (uint4*)lds;
    for (uint32_t i = threadIdx.x; i < nvec; i += PT_THREADS) d4[i] = s4[i];
    return skew;
}

// Emit the byte span [dst, dst + len) where byte k = f(k): full 16-byte chunks as one vector store.
template <typename F>
__device__ __forceinline__ void emit_span(uint8_t* dst, uint32_t len, F f) {
    const uint64_t a0 = (uint64_t)(uintptr_t)dst & ~uint64_t(15);
    const uint32_t skew = (uint32_t)((uint64_t)(uintptr_t)dst - a0);
    const uint32_t nvec = (skew + len + 15) >> 4;
    for (uint32_t q = threadIdx.x; q < nvec; q += PT_THREADS) {
        const int32_t k0 = (int32_t)(q * 16) - (int32_t)skew;
        if (k0 >= 0 && (uint32_t)k0 + 16 <= len) {
            uint32_t w[4];
            f((uint32_t)k0, w);
            *(uint4*)(dst + k0) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int b = 0; b < 16; ++b) {
                int32_t k = k0 + b;
                if (k >= 0 && (uint32_t)k < len) dst[k] = f.byte((uint32_t)k);
            }
        }
    }
}

#undef PT_THREADS
