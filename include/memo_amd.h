/*
 * memo_amd.h -- C ABI of the MI355X-native MEMO windowed k-mer query path.
 *
 * The reference (StephenHwang/MEMO) is pure Python and has no FFI/plugin layer;
 * the seam this library replaces is function-level, inside
 * /root/reference/src/memo_query.py:main (lines 89-105):
 *
 *     genome_mems_arr = filter_pq(in_file, record, qs, qe + k)           # :100
 *     mem_arr, rec    = memo_init(genome_mems_arr, k, qs, qe, N, memb)   # :103
 *     rec             = memo_query(mem_arr, rec, memb)                   # :104
 *     print_res(rec, out_file, memb)                                     # :105
 *
 * Every entry point below names the reference lines it stands in for.  Plain C
 * types only: pointers, sizes, an opaque handle.  No torch / C++ types, no
 * exceptions across the boundary.  All functions return MEMO_OK (0) or a
 * negative code; memo_last_error() gives the message (thread-local).
 *
 * Row semantics: an index row (start s, end e, annot a) marks the
 * k-mers starting at p, max(e-(k-1), qs) <= p < min(s, qe), as ABSENT for order / genome a.
 * Rows must be sorted by start (they are in every index dap_to_bed.py writes);
 * memo_index_finalize() checks that and sorts on the device when it does not hold.  Index
 * rows have e >= s (dap_to_bed.py:93-98), which bounds a row's reach to k-1 positions; rows
 * with e < s are legal input to memo_query.py and give the same results here, through a
 * separate pass (they are set aside at finalize and applied after each sweep).
 *
 * Result encodings
 *   conservation  uint16 out[L], L = qe - qs:  smallest order a of any row
 *                 covering p, num_docs if none  (= np.argmax(rec, axis=1), :70)
 *   membership    uint32 out[L * W], W = ceil(num_docs / 32): genome g of
 *                 position p is bit (g & 31) of word p*W + (g >> 5); 1 = k-mer
 *                 present (= rec[p, g], :51 / :68); bits >= num_docs are 0
 */
#ifndef MEMO_AMD_H
#define MEMO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MEMO_OK 0
#define MEMO_EINVAL (-1)    /* bad argument; also: a row that writes has an annot outside the
                               result columns -- the reference's IndexError (NumPy) / UB (Numba) */
#define MEMO_EHIP (-2)      /* HIP runtime error */
#define MEMO_ENOTREADY (-3) /* index not finalized */
#define MEMO_EUNSORTED (-4) /* rows not sorted by start and sorting was disabled */
#define MEMO_ELONGROW (-5)  /* more than 2^22 rows have end < start.  dap_to_bed.py:93-98 never emits
                               such a row; a few are accepted and applied by a side pass */

typedef struct memo_index memo_index_t; /* one chromosome's rows, resident in HBM */

typedef struct memo_index_info {
    uint64_t rows;          /* m */
    int64_t min_start;      /* valid after finalize */
    int64_t max_start;
    int32_t device;
    int32_t bucket_shift;   /* bucket b starts at pivot position b << bucket_shift */
    uint64_t buckets;
    int32_t was_sorted;     /* 1 if the rows arrived start-sorted */
    int32_t finalized;
    uint64_t device_bytes;  /* HBM held by this index (columns + padding + bucket table) */
    int32_t packed_format;  /* 0 = none, 4 = 4 B/row, 6 = 6 B/row (memo_index_pack) */
    int32_t has_wide;       /* 1 while the three int64 columns are resident */
} memo_index_info_t;

const char *memo_last_error(void);
int memo_device_count(void);
const char *memo_version(void);

/* ---- index lifecycle ---------------------------------------------------------------
 * Stands in for the arrays filter_pq returns (memo_query.py:28-36) and memo_init
 * re-types (:45), but kept as three int64 columns (exactly what the Parquet file
 * stores, parquet_compress_bed.py:21-26) and kept RESIDENT so that many windows can be
 * queried against one upload.  The library owns the device memory; the caller owns
 * every host buffer and nothing is retained after a call returns. */
int memo_index_create(uint64_t rows, int32_t device, memo_index_t **out);
/* copy host columns (pageable or pinned) into the index; blocking */
int memo_index_upload(memo_index_t *ix, const int64_t *start, const int64_t *end,
                      const int64_t *annot, uint64_t rows);
/* streaming form of the same: copy `rows` host rows to row_offset .. row_offset + rows, so that
 * a region slice can be uploaded row group by row group while the next one is decoded; create
 * the index with an upper bound and memo_index_truncate() it to the rows actually written */
int memo_index_upload_rows(memo_index_t *ix, uint64_t row_offset, const int64_t *start,
                           const int64_t *end, const int64_t *annot, uint64_t rows);
int memo_index_truncate(memo_index_t *ix, uint64_t rows);
/* device pointers of the three columns, for callers that fill them on the device
 * (synthetic generator, another kernel); each holds `rows` int64 */
int memo_index_columns(memo_index_t *ix, int64_t **d_start, int64_t **d_end, int64_t **d_annot);
/* check start-sortedness and end >= start, sort by start on the device if needed
 * (allow_sort != 0), build the start-bucket table.  bucket_shift <= 0 picks the default. */
int memo_index_finalize(memo_index_t *ix, int32_t bucket_shift, int32_t allow_sort);
/* Build the query-time row format: one 32-bit word per row
 *     start mod 2^16 | min(end - start, 255) << 16 | annot << 24
 * (4 B/row; when some annot > 255 the annot moves to a separate uint16 column, 6 B/row).
 * Exact for every query with k <= 256: inside a tile's row slice starts span far less than 2^16
 * positions, and an overlap of >= 255 never writes when k - 1 <= 255.  Queries then read the packed
 * rows (6x / 4x fewer bytes); k > 256 keeps using the int64 columns.  keep_wide == 0 frees the int64
 * columns (an HPRC-scale shard is 37 GB packed against 225 GB as int64); such an index answers
 * k <= 256 only and cannot be re-uploaded.  Needs 0 <= annot <= 65535 on every row. */
int memo_index_pack(memo_index_t *ix, int32_t keep_wide);
int memo_index_get_info(const memo_index_t *ix, memo_index_info_t *info);
void memo_index_destroy(memo_index_t *ix);

/* ---- the hot path: memo_init + memo_query + reduction (memo_query.py:42-63, :70) ----
 * d_out is a DEVICE pointer (16-byte aligned) in the index's device; the launch is
 * asynchronous on `stream` (a hipStream_t, NULL = default stream).  The window may be
 * any [qs, qe); rows outside (qs, qe + k) are ignored exactly as filter_pq ignores them
 * (:25-27 with :100).  An annot outside the result columns on a row that writes sets a
 * sticky device flag that memo_query_check() reports as MEMO_EINVAL. */
int memo_query_conservation_dev(memo_index_t *ix, int64_t qs, int64_t qe, int32_t k,
                                int32_t num_docs, uint16_t *d_out, void *stream);
int memo_query_membership_dev(memo_index_t *ix, int64_t qs, int64_t qe, int32_t k,
                              int32_t num_docs, uint32_t *d_out, void *stream);
/* same values as memo_query_conservation_dev, one byte per position; needs num_docs <= 255.
 * Halves the bytes each rank sends in the multi-GPU gather (memo_amd/shard.py). */
int memo_query_conservation_u8_dev(memo_index_t *ix, int64_t qs, int64_t qe, int32_t k,
                                   int32_t num_docs, uint8_t *d_out, void *stream);
/* synchronise `stream`, return and clear the sticky error of earlier queries */
int memo_query_check(memo_index_t *ix, void *stream);

/* ---- one-shot host form: the drop-in for memo_query.py:103-104 + :70 ------------------
 * Host pointers in and out; uploads, finalizes, queries, downloads, frees.  `device`
 * is a HIP device ordinal. */
int memo_conservation(const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows,
                      int64_t qs, int64_t qe, int32_t k, int32_t num_docs, uint16_t *out,
                      int32_t device);
int memo_membership(const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows,
                    int64_t qs, int64_t qe, int32_t k, int32_t num_docs, uint32_t *out_bits,
                    int32_t device);

/* ---- raw device buffers, for hosts that do not bring a device allocator (the CLI) ------ */
int memo_dev_malloc(int32_t device, size_t bytes, void **out);
int memo_dev_free(int32_t device, void *p);
/* copy host -> device / device -> host on `stream` and wait for it */
int memo_dev_upload(int32_t device, void *dev, const void *host, size_t bytes, void *stream);
int memo_dev_download(int32_t device, void *host, const void *dev, size_t bytes, void *stream);

/* ---- print_res (memo_query.py:65-71), host side ----------------------------------------
 * Byte-identical text: conservation = decimal + '\n' per position (a single '\n' when
 * L == 0); membership = num_docs '0'/'1' separated by ' ' per line.  Return the number
 * of bytes the text needs; it is written only if it fits in cap. */
size_t memo_emit_conservation(const uint16_t *vec, int64_t L, char *buf, size_t cap);
size_t memo_emit_membership(const uint32_t *bits, int64_t L, int32_t num_docs, char *buf, size_t cap);

/* ---- `memo view` binning: the per-bin histogram of plot_conservation.py:52-56 ------------------
 * d_vec: conservation result on `device` (L values); edges: nbins + 1 HOST values, the reference's
 * list(map(int, np.linspace(0, L, nbins + 1))); counts: HOST array [nbins][num_docs + 1] of uint64,
 * counts[b][v] = number of positions p in [edges[b], edges[b+1]) with d_vec[p] == v.
 * Blocking (synchronises `stream`). */
int memo_bin_conservation_dev(const uint16_t *d_vec, int64_t L, const int64_t *edges, int32_t nbins,
                              int32_t num_docs, uint64_t *counts, int32_t device, void *stream);

/* ---- index-row construction: dap_to_bed.py:55-134 (--mem [--order] [--overlap]) -------------
 * A DAP (src/index.sh:83) has one row per pivot position: the matching statistic of every
 * non-pivot genome at that position.  Rows go in as a HOST int32 matrix [positions][columns],
 * consecutive positions starting at 0, in as many pushes as the caller likes (state carries over);
 * each push produces, on `device`, the (record, start, end, annot) rows the reference would print
 * for those positions, in its order.  rec_begin: nrec + 1 cumulative record offsets of the pivot
 * (from its .fai).  memo_dap_fetch copies the rows of the last push; memo_dap_finish returns the
 * chr-end rows of a DAP that stops inside a record (at most `columns` rows). */
typedef struct memo_dap memo_dap_t;
int memo_dap_create(int32_t columns, const int64_t *rec_begin, int32_t nrec, int32_t sort_order,
                    int32_t overlaps, int32_t device, memo_dap_t **out);
int memo_dap_push(memo_dap_t *h, const int32_t *lcp, int64_t positions, uint64_t *out_rows);
int memo_dap_fetch(memo_dap_t *h, int32_t *rec, int64_t *start, int64_t *end, int32_t *annot);
int memo_dap_finish(memo_dap_t *h, int32_t *rec, int64_t *start, int64_t *end, int32_t *annot,
                    uint64_t *out_rows);
void memo_dap_destroy(memo_dap_t *h);
/* host-side parser for the DAP text (whitespace-separated decimal integers), multi-threaded.
 * Returns how many integers the text holds (they are written only when cap is enough), -1 on a
 * malformed character. */
int64_t memo_parse_ints(const char *text, size_t len, int64_t *out, size_t cap);
/* BED text of such rows: "name\tstart\tend\tannot\n" (dap_to_bed.py:105,109).  names: nrec
 * NUL-terminated strings back to back.  Returns the bytes needed; writes only if they fit. */
size_t memo_emit_bed(const int32_t *rec, const int64_t *start, const int64_t *end, const int32_t *annot,
                     uint64_t rows, const char *names, int32_t nrec, char *buf, size_t cap);

/* ---- transport coding of uint8 conservation results (multi-GPU gather) ---------------------
 * One nibble per position; values >= 15 travel in an exception list of `cap` slots.  Lossless.
 * wire size = memo_transport_bytes(n, cap); pack and unpack are asynchronous on `stream`.
 * memo_transport_exceptions() tells (synchronising `stream`) how many exceptions the sender found:
 * more than cap means this slice has to travel as plain bytes instead. */
size_t memo_transport_bytes(int64_t n, uint32_t cap);
int memo_transport_pack_dev(const uint8_t *d_vec, int64_t n, uint32_t cap, void *d_wire, int32_t device,
                            void *stream);
int memo_transport_unpack_dev(const void *d_wire, int64_t n, uint8_t *d_vec, int32_t device, void *stream);
int memo_transport_exceptions(const void *d_wire, int32_t device, void *stream, uint32_t *found, uint32_t *cap);

/* Second, denser coding for the same purpose (memo_transport.hip): 2 bits per position (values 1..3;
 * 0 = escape) + one nibble per escape, allocated exactly per 32768 positions from a B region of
 * `b_capacity` bytes (a multiple of 4), + the same exception list (nibble 15: values > 17).  Buffers
 * 16-byte aligned.  _stats (synchronising `stream`) returns the exceptions found and their capacity,
 * the B bytes taken and the B capacity: the slice is complete iff neither exceeds.  Pack once with
 * generous capacities to learn what a workload needs (at most n / 2 + 4 * ceil(n / 32768) bytes of B). */
size_t memo_transport_dense_bytes(int64_t n, uint32_t b_capacity, uint32_t cap);
int memo_transport_dense_pack_dev(const uint8_t *d_vec, int64_t n, uint32_t b_capacity, uint32_t cap,
                                  void *d_wire, int32_t device, void *stream);
int memo_transport_dense_unpack_dev(const void *d_wire, int64_t n, uint32_t b_capacity, uint32_t cap,
                                    uint8_t *d_vec, int32_t device, void *stream);
int memo_transport_dense_stats(const void *d_wire, int32_t device, void *stream, uint32_t *found, uint32_t *cap,
                               uint32_t *b_taken, uint32_t *b_capacity);

/* ---- synthetic pangenome index (BASELINE.json configs 2-5) -------------------
 * Fills rows [0, rows) of the index with global rows row_begin + i of the generator
 *   start = 1 + floor(i * den / num), end = start + mix(seed, 2i) % 60,
 *   annot = 1 + mix(seed, 2i+1) % (num_docs - 1)
 * on the device (no host copy).  Same generator as oracle_synth_rows. */
int memo_synth_fill(memo_index_t *ix, uint64_t row_begin, uint64_t num, uint64_t den,
                    int32_t num_docs, uint64_t seed);

/* ---- tuning knobs (process-wide; also read once from the environment: MEMO_TILE_W,
 * MEMO_WAVES, MEMO_MEMB_ALGO).  0 = let the library choose.  tile_w: positions per tile
 * (256..4096); waves: 1 or 4 waves share a tile (8: the unclipped conservation kernel only, else
 * the library's choice); membership_algo: 1 = direct scatter,
 * 2 = doubling, 3 = runs (bit planes per genome + register transpose), 4 = the same planes without
 * clipping and with the result staged through LDS (packed rows, <= 512 genomes; else 3).
 * Results never depend on these. */
int memo_set_tuning(int32_t tile_w, int32_t waves, int32_t membership_algo);
/* 0 = queries read the packed rows when the index has them (default); 1 = always the int64
 * columns (also MEMO_ROWS=wide).  For A/B measurements; results are identical. */
int memo_set_row_source(int32_t source);
/* 0 = library's choice, 1 = one workgroup per tile, 2 = persistent workgroups (as many as stay
 * resident; each walks a run of tiles and looks the next one up under the current one's work).
 * Also MEMO_PERSIST.  Results are identical. */
int memo_set_persistent(int32_t mode);
/* conservation scatter with packed rows: 0 = library's choice, 1 = clip every interval to the tile,
 * 2 = unclipped into level arrays with a halo (fewer instructions and registers per row; only when
 * every annot of the index is inside the result matrix and workgroups are not persistent, else 1 is
 * used anyway).  tile_w of memo_set_tuning is then the size of a level array, halo included.
 * Also MEMO_SCATTER.  Results are identical. */
int memo_set_scatter(int32_t mode);

/* profiling aid: one pass that reads the three columns exactly once (24 B/row) with the
 * sweep's access shape, to calibrate the FETCH_SIZE counter on a known byte count */
int memo_debug_stream_rows(memo_index_t *ix, void *stream);
/* profiling aid for -DMEMO_STAMPS builds of the conservation sweep: a device buffer of 8 uint64
 * per workgroup that receives the cycles wave 0 spent in each phase (NULL = off; ignored by the
 * product library, which contains no stamp) */
int memo_debug_set_stamp_buffer(uint64_t *d_buffer);

#ifdef __cplusplus
}
#endif
#endif /* MEMO_AMD_H */
