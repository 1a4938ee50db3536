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

// msd.hip -- round 0 of the row sort as an MSD partition finished in LDS.  Sorts the rows by key = the top 32 (key64 = 0) or all 64 bits of
// (head << z), head = a row's first eight bytes as a big-endian number (rows of fewer bytes: zero-filled), ties in row order: d_perm = the
// order, heads[j] = 1 when the key at position j differs from the one in front of it, else 0.  keysA / keysB: n * 8 bytes each, idxA / idxB:
// n * 4 bytes each, ws: msd_ws_bytes(n).  *status: 0 = done; 1 = not for this table (buckets heavier than a workgroup's LDS: few distinct
// heads) -- nothing usable was written, the caller takes the LSD passes; 2 = the table's rows share fewer than z leading bits
// (h_andor = AND / OR over all heads says how many): call again with that z.
// Rows wider than the key: the groups that tie on the key are sorted by whole rows inside the finishing kernel (up to 32 rows a group) and
// heads[j] is 1 = a new row value / 2 = equal to the row in front, final / 0 = ties with it on the key, left open (*settled = false: the
// caller's refinement rounds take over).  Two host waits inside (the bucket bound; the open-group flag).
size_t msd_ws_bytes(uint64_t n);
int msd_round0(uq_ctx* ctx, const uint8_t* table, uint32_t C, uint64_t n, uint32_t z, int key64, void* keysA, void* keysB, uint32_t* idxA, uint32_t* idxB,
               uint32_t* perm, uint8_t* heads, void* ws, size_t ws_bytes, int* status, uint64_t* h_andor, bool* settled);
