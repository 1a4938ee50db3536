// radix.h -- internal interface of the device LSD radix sort (radix.hip) used by sort.hip.
#pragma once
#include "common.h"

// Stable LSD radix sort of (u64 key, u32 value) pairs on bits [begin_bit, end_bit) of the key, 8 bits
// per pass; passes whose digit is the same for every key are skipped.  `keys`/`vals` hold the input,
// `keys_alt`/`vals_alt` are same-sized ping-pong buffers.  On return *in_alt tells which pair of
// buffers holds the sorted result (0 = keys/vals, 1 = keys_alt/vals_alt).
// `ws` must provide radix_ws_bytes(n) bytes of device scratch.
size_t radix_ws_bytes(uint64_t n);
int radix_sort_pairs(uq_ctx* ctx, uint64_t* keys, uint32_t* vals, uint64_t* keys_alt, uint32_t* vals_alt,
                     uint64_t n, int begin_bit, int end_bit, void* ws, int* in_alt);
// The same with 32-bit keys: 8 bytes a pair instead of 12, four digit positions instead of eight.
// h_hist: the digit census of the keys (h_hist[p * 256 + d] = keys whose byte p is d) when the caller has taken it already
// (radix_prefix_census32), else NULL.
// digit0_counted: the per-tile counts of the lowest digit are in `ws` already (radix_prefix_census32 left them there).
int radix_sort_pairs32(uq_ctx* ctx, uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt,
                       uint64_t n, int begin_bit, int end_bit, void* ws, int* in_alt, const uint32_t* h_hist, int digit0_counted);
// keys[i] = the 32 bits of keys64[i] behind its z leading bits, vals[i] = i, h_hist = the census of the new keys,
// and the first pass's per-tile counts left in `ws`: one pass over keys64.
int radix_prefix_census32(uq_ctx* ctx, const uint64_t* keys64, uint64_t n, uint32_t z, uint32_t* keys, uint32_t* vals, void* ws, uint32_t* h_hist);
// The same from the table itself (rows of C >= 8 bytes, chunk = their first eight bytes as a big-endian number; the 64-bit values are never
// written out), with z from a sample: h_andor[0 .. 1] = AND / OR over every row's chunk, so that the caller can tell whether z holds.
int radix_rows_prefix_census32(uq_ctx* ctx, const uint8_t* table, uint32_t C, uint64_t n, uint32_t z, uint32_t* keys, uint32_t* vals, void* ws,
                               uint32_t* h_hist, uint64_t* h_andor);
