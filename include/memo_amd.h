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
 *
 * Threads and streams.  A memo_index_t is used by ONE host thread at a time: a query updates the handle (which kernel
 * family answered, the k-class views and tile tables it builds on the way, the sticky status word) without a lock.
 * Different indexes -- and builders, one-shot calls, transfers -- are independent and may run on different threads;
 * the process-wide pieces (pinned staging rings, the worker pool, memo_last_error) are locked or thread-local.  The
 * *_dev forms enqueue on the caller's stream and return; one thread may put queries of one index on several streams:
 * whatever a query builds for later ones (views, tile tables) is complete on the device before the call returns, and
 * nothing a queued sweep reads is freed before the device has drained: what a query takes out of service (a view past the
 * budget) waits on the index's retire list for the next memo_query_check -- queries themselves never wait for the DEVICE; the
 * query that builds something (a view, the placed copy of a view, the query order of rows that came in start order: each
 * built beside what is in use, never over it) waits for ITS stream once; memo_index_prepare moves all of that out of the query
 * path.
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
#define MEMO_EUNPACKABLE (-6) /* memo_builder_*: these rows cannot take the packed way in (unsorted, negative
                               start, annot outside [0, 4095]); use memo_index_upload + _finalize */

typedef struct memo_index memo_index_t; /* one chromosome's rows, resident in HBM */

/* Versioned: the CALLER sets struct_bytes = sizeof(memo_index_info_t) as it compiled it (a binder built against an older
 * header has a shorter struct); memo_index_get_info_v5 writes no more than that many bytes -- whole leading fields of the layout
 * below, which only ever grows at its end (a size that ends inside a field is rounded down to the field's start) -- and puts
 * the number of bytes it wrote back into struct_bytes, its own layout version into `version`.  struct_bytes below 16 (never
 * set) is MEMO_EINVAL.  The entry point carries the version of the FIRST layout that began with struct_bytes in its name:
 * rounds 1-3 had no such field (every round grew the struct: a round-2 binder would have had its stack overwritten by round
 * 3's library), and a binder that still passes one of those structs must fail when it looks the symbol up, not read its own
 * garbage as a size -- so `memo_index_get_info` is a macro for C callers and not an exported name. */
#define MEMO_INDEX_INFO_VERSION 5
typedef struct memo_index_info {
    uint32_t struct_bytes;  /* in: sizeof of the caller's struct; out: bytes written */
    uint32_t version;       /* out: MEMO_INDEX_INFO_VERSION of the library */
    uint64_t rows;          /* m */
    int64_t min_start;      /* valid after finalize */
    int64_t max_start;
    int32_t device;
    int32_t bucket_shift;   /* bucket b starts at pivot position b << bucket_shift */
    uint64_t buckets;
    int32_t was_sorted;     /* 1 if the rows arrived start-sorted */
    int32_t finalized;
    uint64_t device_bytes;  /* HBM held by this index: rows in every resident format, bucket tables, and side_bytes */
    int32_t packed_format;  /* 0 = none; 4, 12 = 4 B/row (8- / 12-bit annot); 6 = 6 B/row (memo_index_pack) */
    int32_t has_wide;       /* 1 while the three int64 columns are resident */
    float pack_ms;          /* device time of the last memo_index_pack: annot census + packing kernel + the ordering of the rows
                               inside their buckets, HIP events on its stream (the narrowing pass, timed apart) */
    int32_t dense_rows;     /* 1 while the dense rows of memo_index_pack_dense are resident */
    uint64_t long_rows;     /* rows with end < start, kept aside (see above) */
    uint64_t max_annot;     /* largest annot of the packed rows (valid when packed_format != 0) */
    int64_t bucket_base;    /* the bucket table starts at this bucket (a region slice of memo_index_import_packed) */
    int32_t last_sweep;     /* which kernel family answered the last query on this index: 0 none yet; conservation:
                               1 clipped scatter into doubling level arrays, 2 unclipped doubling, 3 unclipped
                               radix-4, 4 unclipped mixed (1, 4, 16, then doubling), 5 dense rows -- the library picks
                               2 / 3 / 4 per query from k and the overlap lengths of the rows it sampled when the
                               packed rows were made; membership: 6 bit planes on the dense rows, 7 any other */
    int32_t last_variant;   /* of the dense-row sweep: 0 every wave worked its tile out, 2 the tile's rows from the index's tile table, 3 the same
                               on a k-class view of SIX rows per group (last_view_rows_per_group) */
    uint64_t dense_row_count; /* rows the dense rows hold (0: none resident): fewer than `rows` when the rows that can never
                               write at k <= 64 (overlap >= 63, or end < start) were left out of them -- they are when more
                               than a tenth of the rows are such rows (none of the synthetic index, 40 % of one built from
                               sequences) -- the dense rows then have their own numbering and bucket table */
    uint64_t last_rows_read;  /* rows of the row source the last sweep read: the index's rows, the dense rows, or a k-class VIEW
                               of the dense rows (conservation, k - 1 <= 32: classes of two, caps 2, 4 ... 32) or of the 4-byte
                               words (any query on formats 4 / 12 with k - 1 <= 128: caps 2 ... 32 by 2, ... 64 by 8, ... 128 by
                               16) that leaves out the rows whose overlap is the class's cap or more -- none of them can write at
                               a k - 1 up to that cap; memo_query.py:49 drops them per query, the view once per index and class,
                               when that spares a fifth of the rows and the device has room for it (see memo_index_prepare) */
    float last_view_ms;       /* device time of building that view, when the last sweep was the one that built it (else 0) */
    int32_t row_order;        /* order of the 4-byte rows inside a start bucket: 0 by start (as packed), 1 / 2 dealt round-robin
                               over the bucket's starts in chunks of four (2: the rows of a start by overlap mod 32: the library's
                               order), 3 dealt over annot mod 32 (A/B library only) -- the order
                               never changes a result, it spreads a wave's LDS atomics (memo_amd/csrc/memo_interleave.hip) */
    uint64_t side_bytes;      /* of device_bytes: what queries built on the side -- k-class views, tile tables, and buffers taken
                               out of service that wait for the device to drain (freed by the next memo_query_check) */
    int32_t views_resident;   /* k-class views held now */
    int32_t tile_tables_resident;
    uint64_t view_builds;     /* k-class views built over the index's lifetime: a service can watch it for thrashing */
    int32_t last_level_arrays; /* last_sweep == 4: level arrays the sweep allocated per tile -- only those some row of the index can
                               write to at this k (which overlaps occur in the index is known exactly since its rows were packed) */
    int32_t last_view_placed; /* the view the last sweep read has its rows placed inside their groups against LDS bank conflicts (a second
                               pass over the rows, decided like the first: MEMO_OPT_VIEW_PLACES) */
    uint64_t view_placings;   /* dense views (re)built with their rows placed, over the index's lifetime */
    int32_t last_view_rows_per_group; /* rows per 16-byte group of the dense rows the last sweep read: 5, or 6 (a view whose groups carry
                               their bucket: 2.67 B per row; last_rows_read then counts the places a bucket leaves empty too) */
    int32_t reserved;
} memo_index_info_t;

const char *memo_last_error(void);
int memo_device_count(void);
const char *memo_version(void);
/* Host threads the library runs flat out (the pool that narrows host rows for memo_conservation / memo_membership /
 * memo_builder_* and stages transfers; the text emitters): the CPUs the process may run on, cut to its cgroup's CFS
 * bandwidth quota -- a container granted 16 CPUs' worth of time per period on a 256-CPU host gets 16 threads, because
 * threads beyond the quota only bring the period's freeze forward -- at most 32; MEMO_HOST_THREADS overrides.  The
 * two inputs come back through the pointers (either may be NULL; quota 0 = none).  The reference's path is
 * single-threaded (a bare @jit, memo_query.py:57): nothing to mirror, this is about being a good tenant. */
int memo_host_threads(int32_t *cpus_allowed, double *cgroup_quota_cpus);

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
/* Build the query-time row format: one 32-bit word per row, laid out by the largest annot of the index
 *     annot <= 255    start mod 2^16 | min(end - start, 255) << 16 | annot << 24              format 4, 4 B/row
 *     annot <= 4095   min(end - start, 255) | (start mod 2^12) << 8 | annot << 20            format 12, 4 B/row
 *     else            the first word with annot 0 + the annot in a separate uint16 column    format 6, 6 B/row
 * Exact for every query with k <= 256: inside a tile's row slice starts span far less than 2^12
 * positions, and an overlap of >= 255 never writes when k - 1 <= 255.  Queries then read the packed
 * rows (6x / 4x fewer bytes); k > 256 keeps using the int64 columns.  keep_wide == 0 frees the int64
 * columns (an HPRC-scale shard is 37 GB packed against 225 GB as int64); such an index answers
 * k <= 256 only and cannot be re-uploaded.  Needs 0 <= annot <= 65535 on every row. */
int memo_index_pack(memo_index_t *ix, int32_t keep_wide);
/* A denser copy of the packed rows: 24 bits per row, five rows per 16 bytes (3.2 B per row),
 *     (start mod 2^10) << 6 | min(end - start, 63)   +   annot (8 bits; a ninth in the group's spare byte)
 * Exact for k <= 64 on the unclipped conservation sweep (level arrays of <= 1024 cells), and its fastest source:
 * conservation queries read these rows wherever they are resident and can answer (BASELINE config 3, launches back
 * to back: 0.324 ms against 0.374 on the 4-byte rows at k = 31; config 5, 500 genomes: 0.27 against 0.33).  Up to 255
 * genomes both result widths; 256 .. 511 genomes (round 4) uint16 results -- membership queries and k > 64 of such an
 * index read the 4-byte rows.  Membership
 * queries of up to 255 genomes read the dense rows when the index holds no 4-byte rows (4 % slower than on the 4-byte
 * rows).  Needs memo_index_pack first and every annot <= 511.  keep_packed == 0 frees the 4-byte rows: such an index
 * holds 3.2 B per row and answers only what the dense rows (or, if still resident, the int64 columns) can. */
int memo_index_pack_dense(memo_index_t *ix, int32_t keep_packed);
int memo_index_get_info_v5(const memo_index_t *ix, memo_index_info_t *info);  /* set info->struct_bytes first (see the struct) */
#define memo_index_get_info memo_index_get_info_v5
/* Options of one index (queries never change a result with them).
 *   MEMO_OPT_VIEWS            1 (default): queries may build k-class views of the rows (memo_index_info_t.last_rows_read);
 *                             0: never -- resident views are dropped (the call waits for the device), every sweep reads all
 *                             the rows of its format
 *   MEMO_OPT_VIEW_BUDGET_PCT  the views of ONE row source (the dense rows; the 4-byte words) together stay within this many
 *                             percent of that row source's own bytes; past it the least recently used view is dropped (its
 *                             class is then rebuilt only after four times as many queries as the last time).  Default 200:
 *                             a long-lived index that holds both row sources can grow to three times their bytes.  0 .. 1600.
 *                             (Tile tables are not under this budget: up to 64 of them per index, 32 B per tile of the chromosome
 *                             each -- 3.4 MB for 10^8 positions -- least recently used out first.)
 *   MEMO_OPT_BUILD_COST_PCT   when a query builds a view (or brings the 4-byte rows into the order its KIND of query reads fastest -- rows that
 *                             came in start order, or rows ordered for conservation under membership queries and the reverse): every query
 *                             of a k class that runs without its view adds what the view would have saved it (the rows of its window
 *                             the view leaves out x what a sweep pays per row); the view is built by the query that finds the sum has
 *                             reached this many percent of what the pass is estimated to cost (calibrated, then measured by the
 *                             index's own last pass).  100 (the default) is the ski-rental rule: never more than twice what knowing
 *                             the future would have cost; ten whole-chromosome queries build nothing, two hundred build one view,
 *                             ten thousand 10-kbp windows build nothing.  0: the first query of a class builds (what a test wants);
 *                             0 .. 100000.  memo_index_prepare builds at once whatever this says.
 *   MEMO_OPT_VIEW_ROWS        rows per 16-byte group of the views of the dense rows: 0 (default) the library's choice -- 6 (groups that
 *                             carry their bucket: 2.67 B per row) where they apply and the view holds enough rows per bucket, else 5;
 *                             5 or 6: that kind wherever it applies.  Results never depend on it.
 *   MEMO_OPT_VIEW_PLACES      1 (default): a dense view is built a second time, the places of its rows inside their 16-byte groups chosen
 *                             against LDS bank conflicts (2-4 % of a sweep; the pass costs twice the plain one), once the class's
 *                             queries have lost to the plain view what that costs (MEMO_OPT_BUILD_COST_PCT applies) -- or at once by
 *                             memo_index_prepare; 0: never (views keep their rows in the order they come in). */
#define MEMO_OPT_VIEWS 1
#define MEMO_OPT_VIEW_BUDGET_PCT 2
#define MEMO_OPT_BUILD_COST_PCT 3
#define MEMO_OPT_VIEW_ROWS 4
#define MEMO_OPT_VIEW_PLACES 5
/* Returns the option's PREVIOUS value (>= 0) -- what a caller that changes an option for one pass puts back -- or a negative code. */
int memo_index_set_option(memo_index_t *ix, int32_t option, int64_t value);
/* Build NOW what queries of one kind would otherwise build on the way: the k-class view of the rows such a query reads (else
 * built by the query that finds it has become worth its pass -- MEMO_OPT_BUILD_COST_PCT -- on the caller's stream, with a wait
 * for it; the places of its rows by a later one: MEMO_OPT_VIEW_PLACES) and the tile table of the table-driven
 * sweep (else built by the first query that needs it).  The stand-in for what memo_init does per query (memo_query.py:45-49:
 * recentre, shadow-cast, drop the rows that cannot write) done once for every later query of this k class.  window_hint: the
 * length of the windows to come (0 = the whole chromosome; it decides tile shapes only).  Blocking; returns the device bytes
 * the call took in *bytes_taken (may be NULL; 0 when everything was there, or when a view would not pay or does not fit).
 * A host that will sweep one k over many windows calls this once; one that asks a single question need not. */
int memo_index_prepare(memo_index_t *ix, int32_t k, int32_t num_docs, int32_t membership, int64_t window_hint, void *stream,
                       uint64_t *bytes_taken);
/* A packed index to host memory and back: what the CLI's sidecar cache (memo_amd/cache.py) stores next to
 * the Parquet file, so that a repeat query uploads packed rows from the page cache instead of decoding ZSTD
 * pages.  _export copies the packed rows (rows x uint32; rows x uint16 more when packed_format == 6), the
 * bucket table (info.buckets x int64) and the rows with end < start (3 x info.long_rows int64: starts, ends,
 * annots) into caller buffers.  _import builds a finalized, packed index from such arrays -- or from a SLICE
 * of them: rows [row_base, row_base + rows) and, as `boff`, the buckets - 1 table entries of buckets
 * [bucket_base, bucket_base + buckets - 1) AS THEY ARE in the exported table (absolute row numbers: the first
 * must equal row_base, the last must not exceed row_base + rows); the library rebases them and appends the entry
 * pinned to `rows`.  A whole exported index: row_base 0, bucket_base 0, buckets = info.buckets.  The word layout follows from the arguments as it does in the
 * packers: pa != NULL: format 6; else max_annot > 255: format 12; else format 4.  Host memory may be pageable
 * (a memory-mapped file): it goes through the pinned ring. */
int memo_index_export_packed(memo_index_t *ix, uint32_t *pk, uint16_t *pa, int64_t *boff, int64_t *long_rows);
/* The same for the DENSE rows (five per 16 bytes; memo_index_pack_dense, or a memo_builder_create_rows(...,
 * MEMO_ROWS_DENSE) index): _export copies ceil(rows / 5) groups, the bucket table and the rows with end < start;
 * _import builds a finalized index that holds the dense rows only (3.2 B per row: conservation, k <= 64, <= 511
 * genomes -- memo_dense_rows_can_answer) from such arrays, or from a slice of them: `groups` points at the group
 * that holds row `row_base` (a multiple of 5), `rows` counts from row_base, and the first table entry may lie up to
 * 4 rows behind row_base (the bucket's first row sits inside that group).  So that `memo query` reads the
 * benchmarked row format straight from its sidecar cache. */
int memo_index_export_dense(memo_index_t *ix, void *groups, int64_t *boff, int64_t *long_rows);
/* The resident k-class VIEW of the dense rows that conservation queries with this k read (memo_index_info_t.last_rows_read;
 * memo_index_prepare builds it): the rows whose overlap is below *cap -- what memo_init keeps per query at any k - 1 <= cap
 * (memo_query.py:49) -- as groups of 5 rows (PackedRows3, the view's rows back to back: what memo_index_import_dense takes, so the
 * CLI's cache stores the view the device built instead of rebuilding it on the host) or of 6 rows that carry their bucket (groups end at
 * bucket boundaries, a place a bucket leaves empty holds a copy of one of its rows; memo_amd/csrc/memo_view.hip).  *rows: the
 * view's rows (0: no such view is resident), *group_count: its 16-byte groups, boff: info.buckets entries in units of the view's
 * (rows_per_group 6: padded) row numbers.  groups == boff == NULL: the sizes only. */
int memo_index_export_view(memo_index_t *ix, int32_t k, int32_t rows_per_group, void *groups, int64_t *boff, uint64_t *rows,
                           uint64_t *group_count, int32_t *cap);
int memo_index_import_dense(uint64_t rows, int32_t device, int32_t bucket_shift, int64_t bucket_base,
                            const void *groups, const int64_t *boff, uint64_t buckets, int64_t row_base,
                            int64_t min_start, int64_t max_start, uint64_t max_annot, const int64_t *long_rows,
                            uint64_t n_long, memo_index_t **out);
/* 1 when an index that holds ONLY the dense rows answers this query (the unclipped conservation sweep on them:
 * conservation, 2 <= k <= 64, num_docs <= 511, every annot <= num_docs, at least one row per pivot position between
 * min_start and max_start), else 0.  Pure host arithmetic; the one rule the one-shot form, memo_amd/memo_query.py and
 * the cache use to choose the row format BEFORE they build or import an index. */
int memo_dense_rows_can_answer(uint64_t rows, int64_t min_start, int64_t max_start, uint64_t max_annot, int32_t k,
                               int32_t num_docs, int32_t membership);
int memo_index_import_packed(uint64_t rows, int32_t device, int32_t bucket_shift, int64_t bucket_base,
                             const uint32_t *pk, const uint16_t *pa, const int64_t *boff, uint64_t buckets,
                             int64_t row_base, int64_t min_start, int64_t max_start, uint64_t max_annot,
                             const int64_t *long_rows, uint64_t n_long, memo_index_t **out);
void memo_index_destroy(memo_index_t *ix);

/* ---- packed upload: the fast way in for host rows ------------------------------------------
 * Stands in for the same arrays (memo_query.py:28-36, :45) when they arrive start-sorted, as every
 * region slice of an index does.  Rows are pushed in as many start-ordered pieces as the caller
 * likes (a Parquet row group at a time); worker threads narrow the three int64 columns to the packed
 * query format into a ring of PINNED buffers, and each piece crosses PCIe with hipMemcpyAsync while
 * the next one is being packed: 4-6 B per row on the link instead of 24, from pinned memory.  The same
 * pass validates the rows and builds the start-bucket table, so memo_builder_finish() returns an index
 * that is finalized and packed (int64 columns never reach the GPU; it answers k <= 256).
 * Rows that cannot be packed into one word (unsorted, negative start, annot outside [0, 4095]) make _push
 * return MEMO_EUNPACKABLE: start over with memo_index_upload (+ _finalize, _pack: the 6-byte format).
 * Nothing of the caller's memory is referenced after _push returns.  One builder per thread.
 * memo_builder_create_rows(..., MEMO_ROWS_DENSE) narrows the rows to the DENSE format instead (five rows per 16
 * bytes: 3.2 B per row on the link and in HBM, the fastest source of the conservation sweep) for callers that know
 * their queries fit it (memo_dense_rows_can_answer); a row with an annot > 511 makes _push return MEMO_EUNPACKABLE
 * there too: start over with MEMO_ROWS_PACKED. */
typedef struct memo_builder memo_builder_t;
#define MEMO_ROWS_PACKED 0 /* one 32-bit word per row (formats 4 / 12): every query with k <= 256 */
#define MEMO_ROWS_DENSE 1  /* five rows per 16 bytes: conservation, k <= 64, <= 511 genomes (membership: <= 255) */
int memo_builder_create(uint64_t max_rows, int32_t device, int32_t bucket_shift, memo_builder_t **out);
int memo_builder_create_rows(uint64_t max_rows, int32_t device, int32_t bucket_shift, int32_t row_format,
                             memo_builder_t **out);
int memo_builder_push(memo_builder_t *b, const int64_t *start, const int64_t *end, const int64_t *annot,
                      uint64_t rows);
/* the same from ROWS: rows3 = [rows][3] row-major (start, end, annot of a row side by side) -- the array filter_pq builds
 * (memo_query.py:28-36: uint64; the same bits) -- read as it lies, no transposed copy on the host */
int memo_builder_push_rows(memo_builder_t *b, const int64_t *rows3, uint64_t rows);
/* hands the index over (destroy the builder afterwards; it cannot be used again) */
int memo_builder_finish(memo_builder_t *b, memo_index_t **out);
void memo_builder_destroy(memo_builder_t *b);

/* ---- the hot path: memo_init + memo_query + reduction (memo_query.py:42-63, :70) ----
 * d_out is a DEVICE pointer (16-byte aligned: checked) in the index's device; the launch is
 * asynchronous on `stream` (a hipStream_t, NULL = default stream).  The window may be
 * any [qs, qe), at the same speed wherever it starts (round 4: before, a start that is not a multiple of four ran
 * up to 60 % slower); rows outside (qs, qe + k) are ignored exactly as filter_pq ignores them
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
 * Host pointers in and out; uploads, queries, downloads, frees.  `device` is a HIP device ordinal.
 * For k <= 256 and packable rows (see memo_builder_*) the rows take the packed, pinned way in -- as dense rows
 * (3.2 B per row, the benchmarked kernel) when memo_dense_rows_can_answer says so, else as 4-byte words -- and the
 * result comes back through the same pinned ring.  Anything else is uploaded as int64 columns, validated (and
 * sorted if need be) on the device. */
int memo_conservation(const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows,
                      int64_t qs, int64_t qe, int32_t k, int32_t num_docs, uint16_t *out,
                      int32_t device);
int memo_membership(const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows,
                    int64_t qs, int64_t qe, int32_t k, int32_t num_docs, uint32_t *out_bits,
                    int32_t device);
/* The same two, taking the reference's array AS IT IS: memo_init's first argument, filter_pq's result (memo_query.py:28-36, :100,
 * :103) -- uint64 / int64 [rows][3] row-major, start / end / annot of a row side by side.  No argsort, no three contiguous copies
 * in the caller (at BASELINE config 3 those cost a NumPy host a minute; this call ~50 ms): the host packer reads the rows as they
 * lie (AVX-512: 48 consecutive qwords taken apart by two-source permutes).  Rows that cannot be packed (unsorted, wild annots,
 * k > 256) are transposed into columns inside the call and take the int64 way in, as above. */
int memo_conservation_rows(const int64_t *rows3, uint64_t rows, int64_t qs, int64_t qe, int32_t k, int32_t num_docs, uint16_t *out,
                           int32_t device);
int memo_membership_rows(const int64_t *rows3, uint64_t rows, int64_t qs, int64_t qe, int32_t k, int32_t num_docs, uint32_t *out_bits,
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

/* ---- synthetic pangenome index (BASELINE.json configs 2-5) -------------------
 * Fills rows [0, rows) of the index with global rows row_begin + i of the generator
 *   start = 1 + floor(i * den / num), end = start + mix(seed, 2i) % 60,
 *   annot = 1 + mix(seed, 2i+1) % (num_docs - 1)
 * on the device (no host copy).  Same generator as oracle_synth_rows. */
int memo_synth_fill(memo_index_t *ix, uint64_t row_begin, uint64_t num, uint64_t den,
                    int32_t num_docs, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* MEMO_AMD_H */
