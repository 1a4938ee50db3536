/* uqhip.h -- C ABI of libuqhip.so: the MI355X (gfx950) implementation of the uQ encode/decode
 * hot path.  This is the drop-in boundary: every entry point replaces one numpy / per-base-loop
 * site of the reference (JohnLonginotto/uq, `uq.py`), cited as `uq.py:<lines>` on each.
 *
 * Conventions
 *   - Every function returns 0 on success, non-zero on failure; the message is `uq_last_error()`
 *     (thread-local).  Nothing throws across the ABI.
 *   - `uq_ctx` binds one HIP device and one stream.  It is NOT thread-safe: the caller serialises,
 *     as the single-threaded reference does.  Calls enqueue on the context's stream and are
 *     synchronous at return only where they hand a value back to the host (`h_*` out-parameters).
 *   - `d_*` pointers are device memory (from `uq_dev_alloc`, hipMalloc, or a torch tensor's
 *     data_ptr()); `h_*` are host.  Buffers are caller-owned.  Scratch memory is owned by the
 *     context and grows on demand (`uq_ctx_reserve` pre-sizes it).
 *   - Tables are row-major `uint8[rows][cols]`, as the reference's numpy arrays (uq.py:178-181).
 *   - Row permutations and keys are `uint32` on the device (< 2^32 rows per GPU); the host mirror
 *     widens them where numpy would give int64.
 */
#ifndef UQHIP_H
#define UQHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UQ_ABI_VERSION 1
#define UQ_NONE UINT64_MAX

typedef struct uq_ctx uq_ctx;

/* ---- context, memory, errors.  Replaces cffi -> libc malloc/free (uq.py:111-126, 712-713). */
const char* uq_last_error(void);
int uq_abi_version(void);
int uq_device_count(int* h_count);
/* `stream` = an existing hipStream_t (e.g. torch's current stream) or NULL for a private one. */
int uq_ctx_create(int device, void* stream, uq_ctx** out);
int uq_ctx_destroy(uq_ctx* ctx);
int uq_ctx_reserve(uq_ctx* ctx, size_t scratch_bytes);
int uq_ctx_sync(uq_ctx* ctx);
int uq_dev_alloc(uq_ctx* ctx, size_t bytes, void** d_out);
int uq_dev_free(uq_ctx* ctx, void* d_ptr);
int uq_h2d(uq_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int uq_d2h(uq_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
/* Test aid: overwrite the LDS of every CU with `pattern`-derived words (LDS survives between launches, which hides
 * reads of LDS a kernel never wrote).  No effect on results. */
int uq_debug_scribble_lds(uq_ctx* ctx, uint32_t pattern);
int uq_memset(uq_ctx* ctx, void* d_dst, int value, size_t bytes);
/* Device-side timing on the context's stream (hipEvents), for bench.py's roofline leg. */
int uq_timer_start(uq_ctx* ctx);
int uq_timer_stop(uq_ctx* ctx, float* h_ms);

/* ---- record index.  Replaces `wc -l` (uq.py:85) and the `next(f)` line iteration (uq.py:132-137).
 * uq_count_lines: number of '\n' in d_buf[0, nbytes).
 * uq_index_lines: d_line_start[k] = offset of the first byte of line k, k in [0, nlines];
 *                 d_line_start[nlines] = offset one past the last '\n'.  nlines from uq_count_lines. */
int uq_count_lines(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t* h_nlines);
int uq_index_lines(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t nlines, uint64_t* d_line_start);
/* uq_count_lines in pieces, for a buffer that is still being filled (a file streaming into HBM chunk by chunk, SURVEY.md 8
 * row f2): _begin once, _chunk for every byte range [first_byte, first_byte + chunk_bytes) as it lands -- ranges are whole
 * 16 KiB tiles of the 16-byte-aligned address space (any multiple of 16 KiB from an aligned base), the last one ends with the
 * buffer; every byte is covered exactly once, in any order -- and _end = the line count, exactly what uq_count_lines returns for
 * the whole buffer (uq_index_lines then reuses the census the same way). */
int uq_count_lines_begin(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes);
int uq_count_lines_chunk(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t first_byte, uint64_t chunk_bytes);
int uq_count_lines_end(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t* h_nlines);
/* The same census, the record index and the fused pack + statistics pass QUEUED one behind the other, without the host round trip that
 * `wc -l` (uq.py:85) puts between counting and reading: _end_async queues the closing scan and leaves the line count on the device,
 * uq_index_lines_async (d_line_start holds capacity_lines + 1 entries) and uq_pack_stats_async (below) read it there, and
 * uq_count_lines_wait hands it to the host once everything has been queued.  *h_ok = 0: more lines than capacity_lines, or a 16 KiB
 * tile with more newlines than its list holds -- the index (and what was packed from it) is not usable, take uq_count_lines +
 * uq_index_lines.  Between _end_async and _wait the context must not start another census. */
int uq_count_lines_end_async(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes);
int uq_index_lines_async(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t capacity_lines, uint64_t* d_line_start);
int uq_count_lines_wait(uq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t* h_nlines, int* h_ok);

/* ---- a1: pass-1 statistics.  Replaces uq.py:366-375, 382, 388, 415-425.
 * counts[base * 256 + qual] over every (base, quality) pair of reads [0, nreads); DNA length range;
 * the first record (if any) whose third line does not start with '+' or whose SEQ/QUAL lengths
 * differ.  The struct lives in device memory; the caller zero-initialises it with uq_stats_init
 * (so that several calls / shards accumulate) and copies it back with uq_d2h. */
typedef struct uq_stats {
    uint64_t counts[256 * 256];
    uint64_t bad_plus;          /* smallest read index with a bad '+' line, or UQ_NONE   (uq.py:382) */
    uint64_t bad_len;           /* smallest read index with len(SEQ) != len(QUAL)        (uq.py:388) */
    uint32_t len_min, len_max;  /* uq.py:416-417 */
    uint32_t max_record_bytes;  /* longest record (4 lines), sizing hint for the pack tiles */
    uint32_t reserved;          /* uq_pack_stats only: non-zero = the counts above are incomplete, run uq_stats_accumulate */
} uq_stats;
int uq_stats_init(uq_ctx* ctx, uq_stats* d_stats);
/* The device struct copied to the host (synchronous).  Same bytes as uq_d2h of the whole struct, but a file touches a few hundred
 * of the 65 536 counters: they travel as a short list through the context's pinned buffer instead of 512 KiB to pageable memory. */
int uq_stats_fetch(uq_ctx* ctx, const uq_stats* d_stats, uq_stats* h_stats);
/* The same statistics as the list of their non-zero counters (a file uses a few hundred of the 65 536): key = base * 256 + quality, in
 * no particular order.  n > UQ_STATS_COMPACT_CAP: the list is cut short, take uq_stats_fetch.  What the host decisions of
 * uq.py:448-545 need, without 512 KiB crossing PCIe and being cleared and copied on the host for every shard. */
#define UQ_STATS_COMPACT_CAP 2048u
typedef struct uq_stats_compact {
    uint32_t n, len_min, len_max, max_record_bytes, reserved, pad;
    uint64_t bad_plus, bad_len;
    uint32_t key[UQ_STATS_COMPACT_CAP];
    uint64_t count[UQ_STATS_COMPACT_CAP];
} uq_stats_compact;
int uq_stats_fetch_compact(uq_ctx* ctx, const uq_stats* d_stats, uq_stats_compact* h_out);
int uq_stats_accumulate(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start,
                        uint64_t first_read, uint64_t nreads, uq_stats* d_stats);
/* Multi-GPU: a uq_stats as ONE summable buffer of 65536 + 6 * world int64 words.  uq_stats_export writes the
 * counts and, in this rank's six slots, bad_plus / bad_len (made file-wide by adding read_offset; sign bit flipped so that
 * signed order == unsigned order), len_min, len_max, max_record_bytes and the `reserved` flag; the other ranks' slots are 0.
 * After an all-reduce SUM of the buffer (RCCL), uq_stats_import folds everybody's slots with MIN / MAX back into d_stats:
 * one collective per statistics exchange. */
int uq_stats_export(uq_ctx* ctx, const uq_stats* d_stats, uint32_t rank, uint32_t world, uint64_t read_offset, int64_t* d_words);
int uq_stats_import(uq_ctx* ctx, const int64_t* d_words, uint32_t world, uq_stats* d_stats);
/* First occurrence of each base byte: d_first[b] = min over pairs of (read_index << 20 | position),
 * or UQ_NONE.  Only needed to order N-trick candidates as the reference's dict does (uq.py:480). */
int uq_first_occurrence(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start,
                        uint64_t first_read, uint64_t nreads, uint64_t read_index_base, uint64_t* d_first /*[256], pre-set to UQ_NONE*/);

/* ---- a3 / a4: the per-read packers.  Replaces `encoder_fixed` (uq.py:108-182) and
 * `encoder_variable` (uq.py:188-254).  Row = sum_j code(read[j]) << (bits * (L-1-j)) [+ 1 << bits*L
 * when `variable`], big-endian, right-aligned in `*_bytes_per_row` bytes, high bytes zero. */
typedef struct uq_pack_params {
    int16_t dna_code[256];      /* bases.index(byte), or -1 when the byte is not in `bases` (uq.py:149-152) */
    int16_t qual_code[256];     /* qualities.index(byte), or -1 */
    int32_t n_qual[256];        /* N_qual[byte] for bytes with dna_code == -1, else -1 (uq.py:153) */
    int32_t bits_per_base, bits_per_quality;
    int32_t variable;           /* 0 = encoder_fixed, 1 = encoder_variable */
    int32_t dna_bytes_per_row, quality_bytes_per_row;
    int32_t max_record_bytes;   /* from uq_stats (tile sizing) */
    int32_t dna_max;            /* longest read */
    int32_t avg_record_bytes;   /* optional: mean record bytes (tile sizing for variable-length files), 0 = unknown */
} uq_pack_params;
/* Packs reads [first_read, first_read + nreads) into d_dna / d_qual rows [0, nreads).
 * *d_bad (device, 8 bytes, pre-set to UQ_NONE by the callee) = smallest local read index holding a
 * symbol with no code, else UQ_NONE. */
int uq_pack(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t first_read,
            uint64_t nreads, const uq_pack_params* h_params, uint8_t* d_dna, uint8_t* d_qual, uint64_t* d_bad);
/* a1 + a3/a4 in ONE pass over the stream: packs with GUESSED parameters (h_guess: e.g. the decisions of the previous
 * file / shard, or of a sample of this one) and accumulates the pass-1 statistics of the same reads into d_stats
 * (initialised by uq_stats_init) while the characters are in registers.  The caller then derives the real decisions
 * from d_stats (uq.py:448-545) and keeps the tables iff they equal the guess; otherwise it calls uq_pack with the
 * real ones -- nothing is assumed, the stream is just read once instead of twice when the guess holds.
 * *h_fused = 0: this geometry has no fused kernel (anything but 2-bit A/C/G/T with one contiguous quality range, the
 * Q9 carry case, records beyond one tile) and NOTHING was launched: run uq_stats_accumulate + uq_pack.
 * d_stats->reserved != 0 after the call: the counts are incomplete (a read longer than h_guess->dna_max, a record
 * longer than h_guess->max_record_bytes, SEQ / QUAL lengths differ, a symbol outside the guessed alphabets -- the kernel counts
 * on the codes the conversion produces): re-initialise and run uq_stats_accumulate.  No fused kernel either for more than 64
 * quality symbols (a count bin per byte). */
int uq_pack_stats(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t first_read,
                  uint64_t nreads, const uq_pack_params* h_guess, uint8_t* d_dna, uint8_t* d_qual, uint64_t* d_bad,
                  uq_stats* d_stats, int* h_fused);
/* uq_pack_stats of ALL reads of the buffer behind uq_count_lines_end_async + uq_index_lines_async (above): the number of reads
 * (lines / 4) is taken on the device; d_dna / d_qual hold capacity_reads rows -- more reads than that raise d_stats->reserved. */
int uq_pack_stats_async(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t capacity_reads,
                        const uq_pack_params* h_guess, uint8_t* d_dna, uint8_t* d_qual, uint64_t* d_bad,
                        uq_stats* d_stats, int* h_fused);

/* ---- a9 / a11: the eight --pattern byte layouts.  Replaces numpy.rot90 + ascontiguousarray /
 * asfortranarray + the payload write of numpy.save (uq.py:263-270) and, inverse, numpy.load +
 * rot90(-k) (uq.py:943-945).  pattern_id = 2*k + (order == '.2'), k = rotations (so '0.1'=0,
 * '0.2'=1, '1.1'=2, '1.2'=3, '2.1'=4, '2.2'=5, '3.1'=6, '3.2'=7).  d_payload is the rows*cols
 * payload bytes in file order; the .npy header (shape, fortran_order) is host business. */
int uq_pattern(uq_ctx* ctx, const uint8_t* d_table, uint64_t rows, uint32_t cols, int pattern_id, uint8_t* d_payload);
int uq_unpattern(uq_ctx* ctx, const uint8_t* d_payload, uint64_t rows, uint32_t cols, int pattern_id, uint8_t* d_table);

/* ---- a5: stable argsort of rows in memcmp order.  Replaces table.view('V<C>') +
 * numpy.argsort(axis=0) (uq.py:773-775).  d_perm[j] = index of the j-th smallest row. */
int uq_argsort_rows(uq_ctx* ctx, const uint8_t* d_table, uint64_t rows, uint32_t cols, uint32_t* d_perm);
/* How the row sorts of this context run their round 0 (tuning and tests; the result is the same order either way).  Tables of
 * msd_min_rows rows and more whose heads spread take the MSD partition finished in LDS (csrc/msd.hip), the others the LSD passes
 * (csrc/radix.hip): 0 = the built-in threshold (2^18), < 0 = never.  h_level_bits[nlevels] (1 .. 10 bits a level, at most 24 in all;
 * nlevels = 0: chosen from the row count) = the digits of the partition's levels.
 * uq_sort_counters: how many sorts of this context finished round 0 the one way and the other. */
int uq_sort_config(uq_ctx* ctx, int64_t msd_min_rows, const int* h_level_bits, int nlevels);
int uq_sort_counters(uq_ctx* ctx, uint64_t* h_msd_rounds, uint64_t* h_lsd_rounds);

/* ---- multi-GPU --sort: d_pos[k] = index of the first row of the memcmp-SORTED table that is
 * >= probe row k (numpy.searchsorted(side='left') on void rows).  Splits a locally sorted shard at the
 * splitter rows of the sample sort before the all-to-all. */
int uq_lower_bound_rows(uq_ctx* ctx, const uint8_t* d_sorted_table, uint64_t rows, uint32_t cols,
                        const uint8_t* d_probes, uint64_t nprobes, uint64_t* d_pos);

/* ---- a5 / a6 / a7 / a11: out[j] = table[index[j]].  Replaces table[sort_order] (uq.py:777),
 * key[sort_order] (798), columns_data[idx][sort_order] (822), table[key] on decode (953, 957, 973).
 * index_itemsize in {1,2,4,8} (keys are stored narrowed, uq.py:790). */
int uq_gather_rows(uq_ctx* ctx, const uint8_t* d_table, uint64_t table_rows, uint32_t cols,
                   const void* d_index, int index_itemsize, uint64_t n_out, uint8_t* d_out);

/* Decode-side validation of a stored key / mapping column before it indexes a table (a .uQ file is untrusted input):
 * *h_first_bad = lowest position j with index[j] >= limit, or UQ_NONE.  The reference's numpy raises IndexError on
 * such a file (uq.py:953, 957, 973 `table[key]`; 1016 `column['map'][row[i]]`); uq_gather_rows itself reads row 0 for an
 * out-of-range index (never out of bounds), so callers that want the error check first. */
int uq_check_index_range(uq_ctx* ctx, const void* d_index, int index_itemsize, uint64_t n, uint64_t limit, uint64_t* h_first_bad);

/* ---- a6: unique rows + inverse.  Replaces numpy.unique(rows as void, return_inverse=True)
 * (uq.py:784-789).  Outputs: d_perm = stable sort order of the rows (== argsort(key, stable),
 * uq.py:796), d_key[i] = rank of row i among the distinct rows, d_sorted_key[j] = d_key[d_perm[j]]
 * (uq.py:798), d_unique = the distinct rows in memcmp order (capacity rows*cols; pass NULL to skip),
 * *h_nunique = their number.  Any of d_key / d_sorted_key / d_unique may be NULL. */
int uq_unique_rows(uq_ctx* ctx, const uint8_t* d_table, uint64_t rows, uint32_t cols,
                   uint32_t* d_perm, uint32_t* d_key, uint32_t* d_sorted_key, uint8_t* d_unique, uint64_t* h_nunique);

/* ---- a6: key narrowing.  Replaces key.astype(numpy.min_scalar_type(max(key))) (uq.py:790, 832).
 * uq_key_itemsize: 1/2/4/8 for the largest key value nunique-1.  uq_narrow: u32 -> u8/u16/u32/u64. */
int uq_key_itemsize(uint64_t max_key);
int uq_narrow(uq_ctx* ctx, const uint32_t* d_key, uint64_t n, int itemsize, void* d_out);

/* ---- a7: QNAME column tables.  Replaces numpy.dstack(cols)[0] + structured view (uq.py:814-815,
 * 828-829): rows of ncols fields, each widened to `common_itemsize` and stored BIG-endian so that
 * memcmp order == field-by-field numeric order; then the row functions above apply.
 * uq_unstack_column converts one field of such rows back to a little-endian column of out_itemsize
 * (uq.py:846-847 `[:,idx].astype(dtype)`). */
int uq_stack_columns(uq_ctx* ctx, const void* const* h_d_cols, const int* h_itemsize, int ncols,
                     uint64_t n, int common_itemsize, uint8_t* d_rows);
int uq_unstack_column(uq_ctx* ctx, const uint8_t* d_rows, uint64_t n, int ncols, int common_itemsize,
                      int col, int out_itemsize, void* d_out);

/* ---- a12: unpack.  Replaces split_bits + the per-symbol char map, N restore and sentinel strip of
 * the decoder (uq.py:1002-1007, 1031-1054).  Outputs fixed-pitch text: d_seq / d_qualtxt are
 * uint8[nreads][dna_max], left-aligned, d_len[r] = read length. */
typedef struct uq_unpack_params {
    uint8_t base_char[256];     /* bases[code] */
    uint8_t qual_char[256];     /* qualities[code] (0 where undefined) */
    uint8_t qual_n_base[256];   /* qual_N[code] = the base an N-trick quality code restores, else 0 (uq.py:999, 1036) */
    int32_t bits_per_base, bits_per_quality;
    int32_t variable;
    int32_t dna_bytes_per_row, quality_bytes_per_row;
    int32_t dna_max;
} uq_unpack_params;
int uq_unpack(uq_ctx* ctx, const uint8_t* d_dna, const uint8_t* d_qual, uint64_t nreads,
              const uq_unpack_params* h_params, uint8_t* d_seq, uint8_t* d_qualtxt, uint32_t* d_len,
              uint64_t* d_bad /* smallest row with no sentinel (variable) or UQ_NONE */);

/* ---- f1: the QNAME passes, native on the HOST (no GPU involved).  Replaces the per-line Python of
 * uq.py:348-352, 394-444 (prefix / suffix / separator inference), 555-678 (column typing) and 717-736
 * (column encoding) with the same sequential semantics.  h_buf / h_line_start are HOST copies of the
 * FASTQ bytes and of the record index.  *h_status: 0 = done (*out valid); 1 = outside what the native
 * code reproduces exactly (regex-special separators, integers beyond int64, separators out of order):
 * use the Python implementation; 2 = the reference refuses this input too (uq_last_error() has its message).
 * uq_qname_json: {"prefix","suffix","separators","columns":[...]} as config.json stores them (uq.py:692-695).
 * uq_qname_column: column `col` as little-endian unsigned integers of the column's dtype, nreads entries. */
typedef struct uq_qname uq_qname;
int uq_qname_analyse(const uint8_t* h_buf, const uint64_t* h_line_start, uint64_t nreads, uq_qname** out, int* h_status);
int uq_qname_json(const uq_qname* q, const char** h_json);
int uq_qname_column(const uq_qname* q, int col, void* h_out, uint64_t capacity_bytes);
int uq_qname_free(uq_qname* q);

/* ---- f1 on the DEVICE: the same passes as reductions + one tokenising kernel over the FASTQ already in HBM.
 * uq_qname_layout (uq.py:348-352, 394-413): per character c of the first QNAME (candidate slot k, ch[k]):
 *   entry[k]    = first record whose common prefix with line 1 is <= the last position of c in line 1
 *                 (the record at which the reference's shrinking prefix sheds a c and starts counting it; UQ_NONE = never)
 *   lastviol[k] = last record whose count of c differs from line 1's (0 = none)
 *   c survives the reference's loop  <=>  entry[k] != UQ_NONE and lastviol[k] < entry[k].
 *   min_lcp / min_lcs = lengths of the common prefix / suffix.  flags bit0: some QNAME is a proper prefix or
 *   suffix of line 1 (the reference may raise IndexError there), bit1: a QNAME longer than 255 bytes -- in
 *   both cases use uq_qname_analyse.  Sharded input: h_line1 is the first QNAME of the WHOLE file,
 *   read_index_base the file-wide number of this shard's first read (0: the shard starts with line 1 itself, which
 *   is skipped); entry / lastviol are file-wide read numbers, so ranks combine results with MIN / MAX.
 * uq_qname_tokenise (uq.py:555-565, 717-736): splits QNAME[prefix_len : len - suffix_len] of every read at the
 *   ordered separators and writes, per column c, h_d_vals[c][read] = int(field) (0 if not an integer) and
 *   h_d_strs[c][read] = the field as 8 bytes in text order, zero padded (fields longer than 8 bytes that are
 *   canonical non-negative integers: 0x80 | value, big-endian).  Reductions per column: first_nonint (UQ_NONE if
 *   every field is [+-]digits), vmin / vmax over the integer fields, any_long bit0 = a field that does not fit
 *   the 8-byte key (long and not a canonical integer, or holding NUL / non-ASCII), bit1 = long canonical integer
 *   seen, bit2 = an integer field that is not the canonical decimal of its value ('+7', '007', '-0').  flags: bit0 separators missing / out of order / extra, bit1 whitespace inside a field (Python's int()
 *   strips it), bit2 more than 18 digits, bit3 QNAME shorter than prefix + suffix -- any flag: use uq_qname_analyse.
 * uq_prefix_distinct (uq.py:609-625, the `len(map) > entries_read / 10` checkpoints): from a STABLE argsort
 *   (d_perm) and the group ids in sorted order (d_sorted_key; both from uq_unique_rows), the number of distinct
 *   rows among rows [0, T] for each threshold T.  d_perm holds, per sorted position, the file-order index of that
 *   row as uint32 (perm_itemsize 4: the local argsort) or uint64 (8: file-wide indices after a distributed sort).
 * uq_int_prefix_distinct: the same counts for a column of INTEGER fields whose text is the canonical decimal of its value
 *   (then distinct strings = distinct values) without sorting it: first[v] = lowest read holding value v in [vmin, vmin +
 *   range) via atomic minima (a private LDS table per workgroup for small ranges), then h_counts[k] = #{v: first[v] <= T_k}.
 *   d_val = uq_qname_tokenise's value array; indices are file-wide (read_index_base + i).  any_long bit 2 from
 *   uq_qname_tokenise (a non-canonical integer such as '+7', '007', '-0' was seen) rules this shortcut out.
 * uq_encode_int (uq.py:724-733): d_out[i] = (unsigned itemsize)(d_val[i] - sub). */
typedef struct uq_qname_layout_result {
    uint32_t min_lcp, min_lcs;
    uint32_t flags;
    uint32_t nch;
    uint64_t entry[64];
    uint64_t lastviol[64];
    uint8_t ch[64];
} uq_qname_layout_result;
typedef struct uq_qname_cols_result {
    uint64_t first_nonint[32];
    int64_t vmin[32], vmax[32];
    uint32_t any_long[32];
    uint32_t flags;
    uint32_t reserved;
} uq_qname_cols_result;
int uq_qname_layout(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t nreads, uint64_t read_index_base,
                    const uint8_t* h_line1, uint32_t line1_len, uq_qname_layout_result* h_out);
int uq_qname_tokenise(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t nreads, uint32_t prefix_len,
                      uint32_t suffix_len, const uint8_t* h_separators, uint32_t nsep, int64_t* const* h_d_vals,
                      uint64_t* const* h_d_strs, uq_qname_cols_result* h_out);
int uq_prefix_distinct(uq_ctx* ctx, const void* d_perm, int perm_itemsize, const uint32_t* d_sorted_key, uint64_t n,
                       const uint64_t* h_thresholds, int nthresholds, uint64_t* h_counts);
int uq_int_prefix_distinct(uq_ctx* ctx, const int64_t* d_val, uint64_t n, int64_t vmin, uint64_t range, uint64_t read_index_base,
                           const uint64_t* h_thresholds, int nthresholds, uint64_t* h_counts);
int uq_encode_int(uq_ctx* ctx, const int64_t* d_val, uint64_t n, int64_t sub, int itemsize, void* d_out);

/* ---- e: row routing of the multi-GPU table builds (the reference is one process, these stand in for "all rows are in one
 * numpy array" of uq.py:767-851).  uq_partition_rows: d_dest[r] = rank that row r of an UNSORTED shard goes to in a sample sort, from
 * nsplit = world - 1 splitter rows in ascending order: the number of splitters below the row; a value that e >= 2 splitters share (a
 * tie group heavier than a rank's share) is dealt to those e ranks by file position, d_dest = lb + floor((row_index_base + r) * e /
 * total_rows), which keeps the concatenation of the ranks the stable order.  uq_owner_of_rows: d_owner[j] = rank whose record range
 * [h_shard_starts[k], h_shard_starts[k + 1]) holds file-wide row d_row_index[j].  uq_index_affine: d_out[j] = d_in[j] + add between
 * uint32 / int64 index arrays.  uq_invert_permutation: d_inv[d_perm[j] - base] = j; *h_bad = UQ_NONE or the lowest j that points
 * outside [0, n). */
/* unique + group ids (uq.py:786) of a shard that is ALREADY in memcmp order -- what the global sort hands a rank: no sort, one comparison
 * of neighbouring rows.  d_group[j] = dense rank of row j's value (from 0), d_unique (may be NULL) = the distinct rows in order. */
int uq_unique_sorted_rows(uq_ctx* ctx, const uint8_t* d_sorted_table, uint64_t rows, uint32_t cols, uint32_t* d_group, uint8_t* d_unique,
                          uint64_t* h_nunique);
/* The distinct rows of a table from what its sort left: the order (d_perm, uq_argsort_rows / uq_unique_rows) and the group ids of the
 * sorted positions (d_group = uq_unique_rows' d_sorted_key): d_unique[g] = the first row of group g, g < nunique.  d_perm = NULL: the
 * table has been moved into sorted order already.  Replaces the table half of numpy.unique (uq.py:786); only the distinct rows move. */
int uq_unique_rows_of_groups(uq_ctx* ctx, const uint8_t* d_table, uint64_t rows, uint32_t cols, const uint32_t* d_perm, const uint32_t* d_group,
                             uint64_t nunique, uint8_t* d_unique);
/* d_order = the positions 0 .. n - 1 grouped by d_dest[position] (ndest <= 16 destinations, ascending), each group in ascending order
 * (a stable partition); d_counts[k] (DEVICE, uint64) = size of group k.  What goes in front of an all-to-all: rows by destination rank,
 * requests by owner rank.  *h_bad (may be NULL: then nothing waits for the device) = UQ_NONE or a position whose destination is >= ndest. */
int uq_partition_order(uq_ctx* ctx, const uint8_t* d_dest, uint64_t n, uint32_t ndest, uint32_t* d_order, uint64_t* d_counts, uint64_t* h_bad);
int uq_partition_rows(uq_ctx* ctx, const uint8_t* d_splitters, uint32_t nsplit, uint32_t cols, const uint8_t* d_table, uint64_t rows,
                      uint64_t row_index_base, uint64_t total_rows, uint8_t* d_dest);
int uq_owner_of_rows(uq_ctx* ctx, const int64_t* d_row_index, uint64_t n, const int64_t* h_shard_starts, uint32_t world, uint8_t* d_owner);
int uq_index_affine(uq_ctx* ctx, const void* d_in, int in_itemsize, uint64_t n, int64_t add, void* d_out, int out_itemsize);
int uq_invert_permutation(uq_ctx* ctx, const void* d_perm, int perm_itemsize, uint64_t n, int64_t base, uint32_t* d_inv, uint64_t* h_bad);
/* d_out[(d_index[j] - base) * cols ...] = d_values[j * cols ...]: rows to their places (keys back to file order, uq.py:786's inverse) in one
 * random pass; *h_bad = UQ_NONE or the lowest j whose target lies outside [0, out_rows). */
int uq_scatter_rows(uq_ctx* ctx, const uint8_t* d_values, uint64_t n, uint32_t cols, const void* d_index, int index_itemsize, int64_t base,
                    uint64_t out_rows, uint8_t* d_out, uint64_t* h_bad);

/* ---- f1 INSIDE a3 / a4: the QNAME passes in the pack kernel's read of the stream (uq.py:394-444 layout inference, 555-565 field
 * split, 717-736 int() of the fields).  The pack kernel already holds every record's QNAME line in LDS; with a uq_qname_fused it
 * also splits each line at the separators, parses the fields and writes them as uint32 columns -- against a layout GUESSED on
 * the device from a sample of the file's own reads, and VERIFIED for every read in the same pass:
 *   uq_qname_guess[_async]  line 1 of the file and the layout reductions of uq_qname_layout (min lcp / lcs, entry / lastviol per
 *       character of line 1) over a stratified pseudo-random sample of ~4096 reads -> prefix / suffix lengths and the ordered separators, left in
 *       *d_q (ok = 1), or ok = 0 when the guess declines (no separator, regex metacharacters, > UQ_QF_MAXC columns, a first line
 *       beyond 255 bytes, a digit among the separators, a separator that occurs in line 1 outside the slice the reference reads the ORDER
 *       from -- uq.py:433-436 drops the last character before the suffix, its per-read rule (410-413) counts over the whole middle: such a
 *       character is in the reference's set but not in its order, and what happens then hangs on the LAST read alone -- ...): the pack
 *       kernel then leaves the QNAME lines alone and the caller runs uq_qname_layout / _tokenise.
 *   uq_pack_stats_qname[_async]  = uq_pack_stats[_async] + per read: the line starts with line1[:plen] and ends with
 *       line1[l1len - slen:], its middle holds exactly the separators in order, every field is the canonical decimal of a value
 *       below 10^9 (digits only, no sign, no leading zero), the line is no proper prefix / suffix of line 1; d_vals[c * pitch + r]
 *       = value of field c of read r.  Anything else raises d_q->flags (bit0 separators, bit1 line > 255 bytes, bit2 > 9 digits,
 *       bit3 line shorter than prefix + suffix, bit4 field not a canonical decimal, bit5 / bit6 prefix / suffix mismatch, bit7
 *       proper prefix / suffix of line 1 (or possibly so), bit8 a tile was not visited).  With flags == 0 and nreads == the number
 *       of reads the layout IS the reference's: the sample attains the minima (so min lcp / lcs over the file are plen / slen), no
 *       guessed separator is ever violated, and every other character of line 1 was seen violated at or after its entry in the
 *       sample (which bounds the file's entry from above and its last violation from below).
 *   uq_qname_fused_finish  queued behind it: per column the number of distinct values among reads [0, T] at the checkpoints
 *       T = 10000 * 2^k <= nreads - 1 and T = nreads - 1 (uq.py:586-602, 634-638; canonical decimals: distinct strings = distinct
 *       values).  Value ranges <= 4096: all checkpoints, from first occurrences per value.  Wider ranges (up to 2^20): checkpoint by
 *       checkpoint over the first three, counts[c][k] filled up to the first k with counts[c][k] > T_k / 10 (the reference demotes
 *       the column to `integers` there and later counts do not matter) or for all of them when the file has no more;
 *       undetermined[c] = 1: not decided here (range beyond 2^20, or no checkpoint among the first three fires): sort the column.
 *   uq_qname_fused_fetch  the structure on the host (synchronises).  uq_encode_u32: d_out[i] = (unsigned itemsize)(d_val[i] - sub).
 * Exact by construction; the caller falls back to uq_qname_layout / uq_qname_tokenise on any flag. */
#define UQ_QF_MAXC 8
#define UQ_QF_MAXT 24
typedef struct uq_qname_fused {
    uint32_t ok, plen, slen, nsep, l1len;
    uint32_t flags;                         /* raised by the pack kernel */
    uint32_t sample_step, nth;
    uint8_t line1[256];
    uint8_t seps[32];
    uint8_t inset[256];
    uint32_t vmin[UQ_QF_MAXC], vmax[UQ_QF_MAXC];
    uint32_t undetermined[UQ_QF_MAXC];
    uint64_t nreads;                        /* reads the pack kernel visited */
    uint64_t thresholds[UQ_QF_MAXT];
    uint64_t counts[UQ_QF_MAXC][UQ_QF_MAXT];
} uq_qname_fused;
int uq_qname_guess(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t nreads, uq_qname_fused* d_q);
/* The two queued forms below -- and uq_pack_stats_async -- accept d_line_start == NULL: no record index is expanded at all (uq_index_lines_async is left out of the
 * step: 8 B per line written and read again).  The line starts then come from the newline lists the census queued in front left in the
 * context (uq_count_lines_end_async for the same d_buf): the pack kernel finds each tile's place in them through one record per tile
 * (a binary search over the per-tile counts, done by a small kernel in front), the QNAME sample is stratified by position in the stream
 * instead of by read number.  Same tables, statistics and field values (tests/test_gpu_pack.py, tests/test_gpu_qname.py run both forms);
 * a census whose lists overflowed (uq_count_lines_wait's *h_ok = 0) makes both stand down as the indexed forms do. */
int uq_qname_guess_async(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uq_qname_fused* d_q);
int uq_pack_stats_qname(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t first_read, uint64_t nreads,
                        const uq_pack_params* h_guess, uint8_t* d_dna, uint8_t* d_qual, uint64_t* d_bad, uq_stats* d_stats,
                        uq_qname_fused* d_q, uint32_t* d_vals, uint64_t vals_pitch, int* h_fused);
int uq_pack_stats_qname_async(uq_ctx* ctx, const uint8_t* d_buf, const uint64_t* d_line_start, uint64_t capacity_reads,
                              const uq_pack_params* h_guess, uint8_t* d_dna, uint8_t* d_qual, uint64_t* d_bad, uq_stats* d_stats,
                              uq_qname_fused* d_q, uint32_t* d_vals, uint64_t vals_pitch, int* h_fused);
int uq_qname_fused_finish(uq_ctx* ctx, uq_qname_fused* d_q, const uint32_t* d_vals, uint64_t vals_pitch);
int uq_qname_fused_fetch(uq_ctx* ctx, const uq_qname_fused* d_q, uq_qname_fused* h_out);
/* The fused pass over SHARDS (one rank per GPU, uq_amd.qname_device.analyse_fused_sharded): rank 0's guess is broadcast, every rank's
 * pack kernel verifies and parses its own reads, flags / vmin / vmax are combined over the ranks, and the distinct counts of the
 * small-range columns (h_range[c] <= 4096 values from h_vmin[c]; 0: skip the column) come from first-occurrence tables with FILE-WIDE
 * read numbers: d_first[c * 4096 + v - h_vmin[c]] = read_offset + the shard's lowest read holding v, INT64_MAX where v does not occur;
 * int64 so that a collective's MIN over the ranks is the file's table.  Nothing waits for the device. */
int uq_qname_fused_first_seen(uq_ctx* ctx, const uint32_t* d_vals, uint64_t vals_pitch, uint64_t nreads, uint64_t read_offset,
                              const uint32_t* h_vmin, const uint32_t* h_range, int ncols, int64_t* d_first);
int uq_encode_u32(uq_ctx* ctx, const uint32_t* d_val, uint64_t n, uint32_t sub, int itemsize, void* d_out);
int uq_encode_u32_columns(uq_ctx* ctx, const uint32_t* d_vals, uint64_t vals_pitch, uint64_t n, int ncols, const uint32_t* h_sub,
                          const int* h_itemsize, void* const* h_d_outs);       /* uq_encode_u32 of the first ncols columns of a fused pass, one launch */

/* ---- f3: FASTQ text assembled on the device.  Replaces the decoder's exec-compiled convert_qname
 * (uq.py:1010-1026) and its four prints per read (uq.py:1042-1045).  Inputs: the fixed-pitch text and
 * lengths uq_unpack produced, and the QNAME columns (device arrays, one value per read, little-endian
 * unsigned of itemsize[c] bytes).  Integer columns print str(value + add[c]) (add = 'min' when the
 * column has an offset); mapping columns copy string number `value` of a flattened table
 * (h_d_map_chars[c], h_d_map_offs[c][nmap + 1]; both NULL for integer columns) and pass nmap in add[c] (0 = not
 * given): a value >= nmap then reads the last string, never beyond the table -- uq_check_index_range tells the caller.
 * Record = prefix + fields joined by separators[c] + suffix '\n' SEQ '\n' '+' '\n' QUAL '\n'.
 * d_offsets: uint64[nreads + 1] workspace (record offsets on return).  With d_out == NULL only the total
 * size is computed (*h_total). */
typedef struct uq_emit_params {
    uint8_t prefix[256];
    uint8_t suffix[256];
    uint8_t separators[32];
    int32_t prefix_len, suffix_len, ncols, dna_max;
    int32_t itemsize[32];
    int64_t add[32];
} uq_emit_params;
int uq_emit_fastq(uq_ctx* ctx, const uq_emit_params* h_params, const void* const* h_d_cols, const uint8_t* const* h_d_map_chars,
                  const uint32_t* const* h_d_map_offs, const uint8_t* d_seq, const uint8_t* d_qual, const uint32_t* d_len,
                  uint64_t nreads, uint64_t* d_offsets, uint8_t* d_out, uint64_t capacity, uint64_t* h_total);

/* ---- a12 + f3 in one pass: packed DNA / QUAL rows (+ QNAME columns) -> FASTQ text.  The decoder's split_bits,
 * character maps, N restore, sentinel strip (uq.py:1002-1007, 1031-1054), convert_qname (1010-1026) and prints
 * (1042-1045) without the 2 x dna_max bytes per read of intermediate text uq_unpack + uq_emit_fastq pass through HBM.
 * Same bytes as those two calls.  Protocol: call once with d_out == NULL -- read lengths (variable-length tables:
 * d_len, uint32[nreads]; may be NULL for fixed-length tables), record offsets (d_offsets, uint64[nreads + 1]) and
 * *h_total are computed, *h_bad = lowest row without a valid length sentinel (UINT64_MAX: none; also left in
 * d_bad[0]) -- then again with the output buffer and the SAME d_len / d_offsets / d_bad, which are only read. */
int uq_decode_fastq(uq_ctx* ctx, const uq_emit_params* h_emit, const uq_unpack_params* h_unpack, const void* const* h_d_cols,
                    const uint8_t* const* h_d_map_chars, const uint32_t* const* h_d_map_offs, const uint8_t* d_dna,
                    const uint8_t* d_qual, uint64_t nreads, uint32_t* d_len, uint64_t* d_offsets, uint64_t* d_bad,
                    uint8_t* d_out, uint64_t capacity, uint64_t* h_total, uint64_t* h_bad);

/* ---- synthetic FASTQ ("synth-v1"): workload generation for tests and bench.py.
 * Byte-identical to uq_amd/synth.py.  uq_synth_size: total bytes of reads [first, first+n).
 * uq_synth_fastq: writes them to d_out (capacity >= that size). */
typedef struct uq_synth_spec {
    uint64_t seed;
    int32_t len_lo, len_hi;
    int32_t n_rate;             /* percent of bases that are 'N' */
    int32_t n_qual_exclusive;   /* 1: N <-> '!' exclusively; 0: N always '#', shared */
    int32_t dup;                /* 0 none, 1 DNA, 2 QUAL, 3 both */
    int32_t dup_templates;
    int32_t skip_len_mod4;
    int32_t reserved;
} uq_synth_spec;
int uq_synth_size(uq_ctx* ctx, const uq_synth_spec* h_spec, uint64_t first, uint64_t n, uint64_t* h_bytes);
int uq_synth_fastq(uq_ctx* ctx, const uq_synth_spec* h_spec, uint64_t first, uint64_t n, uint8_t* d_out, uint64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* UQHIP_H */
