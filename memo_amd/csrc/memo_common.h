// memo_common.h -- shared by the translation units of libmemo_amd.so (not part of the public ABI).
#ifndef MEMO_COMMON_H
#define MEMO_COMMON_H

#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "memo_amd.h"
#include "memo_amd_dap.h"
#include "memo_amd_multi.h"
#include "memo_amd_transport.h"

namespace memo {

// error plumbing: every C-ABI function returns fail(code, ...) on error; the message is what
// memo_last_error() hands out (thread-local, defined in memo_index.hip)
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// device -> pageable host memory through the pinned ring of memo_hostpack.hip (DMA of piece i+1 under
// the worker threads' copy of piece i); work already queued on `producer` finishes first
int download_pipelined(int device, void *host, const void *dev, size_t bytes, hipStream_t producer);
int upload_pipelined(int device, void *dev, const void *host, size_t bytes);
double pinned_alloc_ms_total();  // memo_hostcore.cpp: time this process has spent allocating pinned staging slots (MEMO_TIMING)

struct DeviceGuard {  // the caller (e.g. torch) keeps its own notion of the current device
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

constexpr uint64_t kPadRows = 4096;           // sentinel rows behind the last real row
constexpr int64_t kSentinel = INT64_MAX / 4;  // start/end of a padding row: clips to "empty"
constexpr int kDefaultBucketShift = 5;        // 32 pivot positions per bucket
constexpr int64_t kCoordLimit = (int64_t)1 << 61;
constexpr int kStatusBadAnnot = 1;            // sticky device flag: the reference's IndexError case
constexpr int kStatusHugeSlice = 2;           // sticky device flag: >= 2^32 rows reach one tile
constexpr int kStatusExecNarrow = 4;          // -DMEMO_EXEC_CHECK builds: a branch-free row block was entered with lanes disabled

}  // namespace memo

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t err__ = (expr);                                                             \
        if (err__ != hipSuccess)                                                               \
            return memo::fail(MEMO_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(err__),    \
                              __FILE__, __LINE__);                                             \
    } while (0)

// Kernel-shape choices of one index.  All zero in the product (= the library chooses per query);
// only libmemo_amd_ab.so (memo_debug.hip, include/memo_amd_debug.h) can set them, for A/B timing and
// for the tests that walk every tile shape.  Per index, so two threads on two indexes never share
// mutable state.  Results never depend on these.
struct memo_tuning {
    int tile_w = 0;      // positions per tile (256..4096); unclipped kernels: cells per level array
    int waves = 0;       // waves per tile: 1, 4 (8: unclipped conservation only)
    int memb_algo = 0;   // membership: 2 = doubling, 3 = runs, 4 = planes
    int force_wide = 0;  // 1 = read the int64 columns even when packed rows exist
    int scatter = 0;     // conservation, packed rows: 1 = clipped, 2 = unclipped + halo
    int persistent = 0;  // dense rows: 0 = the library's choice, 5 = the table-driven kernel (sweep_conservation_halo3t_kernel),
                         //   1 = every wave works its tile out by itself (sweep_conservation_halo3_kernel)
                         //   (the name is round 3's: 2 / 3 / 4 were persistent workgroups -- profiles/r03_persistent_sweep.txt: 17 - 50 % slower)
    int no_views = 0;    // dense rows: 1 = never read a k-class view (A/B)
    int no_all_write = 0;  // dense rows: 1 = the row blocks keep their "this row writes" test on a view of exactly the writing rows (A/B)
    int force_packed = 0;  // 1 = read the 4-byte rows even when the dense rows are resident and could answer (they are
                           //     the faster source for the conservation sweep: profiles/r02_dense_rows_ab.txt)
    int row_order = 0;     // order of the 4-byte rows inside a bucket (memo_interleave.hip): 0 = the library's (kRowOrderDefault),
                           //     1 = start order as packed, 2 = chunks of four dealt over the starts, 3 = the same with the rows of a
                           //     start ordered by overlap mod 32 (the conservation order), 4 = dealt over annot mod 32 (membership)
};

// one chromosome's rows, resident in HBM (the opaque memo_index_t of the ABI)
struct memo_index {
    int device = 0;
    memo_tuning tune;
    uint64_t rows = 0;
    uint64_t padded = 0;
    int64_t *s = nullptr, *e = nullptr, *o = nullptr;
    int64_t *boff = nullptr;  // boff[b] = first row with start >= (b << bshift); boff[nb-1] == rows
    uint64_t nb = 0;
    int bshift = 0;
    int64_t bbase = 0;        // boff[0] belongs to bucket `bbase` (a region slice imported from the CLI's cache; else 0)
    int64_t min_s = 0, max_s = -1;
    int finalized = 0;
    int was_sorted = 0;
    // packed rows (memo_index_pack): one word per row, layout by the largest annot (PackedRows, memo_sweep.h)
    uint32_t *pk = nullptr;
    uint16_t *pa = nullptr;    // format 6 only: 16-bit annot per row (the word's top byte is 0)
    int packed_fmt = 0;        // 0 = none, 4 = word with 8-bit annot, 12 = word with 12-bit annot and 12-bit start,
                               //   6 = word + 16-bit annot column
    uint64_t packed_rows = 0;  // rows the pk (pa) allocation holds (reused by the next memo_index_pack)
    int row_order = 0;         // how the 4-byte rows are ordered inside a bucket: 0 = by start (as packed), 1 / 2 = interleave_words modes
    // Rows that came in through memo_builder_* or memo_index_import_packed are in start order (or in whatever order their
    // file holds): an index that answers one query -- the one-shot forms, `memo query` -- should not pay a pass over its rows
    // for an order that spares a fraction of one sweep.  They are ordered once the queries that read them have lost to the
    // start order what the ordering pass costs (order_due, memo_view.hip: the same ski-rental rule as the views), by
    // memo_index_prepare, or by memo_index_pack on the finished index.
    int order_pending = 0;
    double order_lost_ns = 0;    // what the queries so far would have saved on rows in the query order (estimate)
    int order_backoff = 1;       // (x 4 after an ordering that found no room for its second copy)
    double order_ns_per_row = 0;  // measured by the last ordering pass (0: the calibrated constant)
    float pack_ms = 0.f;       // device time of the last memo_index_pack (census + packing kernel)
    uint32_t *p3 = nullptr;    // dense rows (memo_index_pack_dense): 16 bytes per 5 rows; annot <= 255 only
    // The dense rows may be FEWER than the index's rows: a row whose 6-bit length field is saturated (overlap >= 63, or
    // end < start) can never write at k <= 64 -- all the dense rows answer -- so when more than a tenth of the rows are
    // such rows they are left out (dense_compact, memo_index.hip: 40 % of the rows of an index built from sequences,
    // profiles/r03_realistic_index*.json; none of the synthetic one).  The dense stream then has its own row numbers and
    // its own bucket table; boff3 == nullptr: the dense rows are the index's rows, numbered alike (rows3 == rows).
    int64_t *boff3 = nullptr;
    uint64_t rows3 = 0, padded3 = 0;
    // k-class views of the dense rows (dense_rows_for, memo_view.hip): the rows whose overlap is below 2 / 4 / ... / 32 -- all a
    // query with k - 1 <= 2 / 4 / ... / 32 can be touched by -- with their own bucket table; built by memo_index_prepare, or by the
    // query that finds that its class's queries have by now paid more for the rows a view would have spared them than the view costs
    // (view_due: ski rental), kept within the views' budget (memo_index_set_option: MEMO_OPT_VIEW_BUDGET_PCT)
    struct DenseView {
        int cap = 0, state = 0;  // state: 0 not looked at yet, 1 built, 2 not worth it (it would spare less than a fifth)
        double lost_ns = 0;      // what this class's queries since it was last looked at would have saved with the view (estimate)
        int backoff = 1;         // the view is due when lost_ns reaches backoff x its estimated cost: x 4 after every eviction or failed
                                 //   allocation (a service that cycles through more classes than the budget holds must not rebuild all the time)
        int seen = 0;            // queries of the class since it was last looked at ...
        int ask_after = 0;       // ... of which this many must pass before it is looked at again: 0, then 16, 64 ... after evictions (what
                                 //   keeps the back-off alive when MEMO_OPT_BUILD_COST_PCT is 0 and every cost is nothing)
        int placed = 0;          // dense views: the rows' places inside their groups were chosen against LDS bank conflicts (memo_view.hip):
        double unplaced_ns = 0;  //   a second pass, decided like the first -- what the class's queries on the view as it is have lost to that
        uint32_t *p3 = nullptr;
        int64_t *boff = nullptr;
        uint64_t rows = 0, padded = 0;
        uint64_t bytes = 0;      // of the rows' allocation (set where a view is installed: a six-row view is (groups + 64) x 16 B, not
                                 //   what its padded row count makes of five-row groups -- ADVICE r05: the budget charged it a fifth too much)
        float build_ms = 0.f;
        uint64_t stamp = 0;      // last use (view_clock): past the views' budget the least recently used one goes
    };
    int views_on = 1;             // memo_index_set_option(MEMO_OPT_VIEWS)
    int view_budget_pct = 200;    // memo_index_set_option(MEMO_OPT_VIEW_BUDGET_PCT)
    int build_cost_pct = 100;     // memo_index_set_option(MEMO_OPT_BUILD_COST_PCT): a view / the ordering is due at this share of its cost
    int view_rows = 0;            // memo_index_set_option(MEMO_OPT_VIEW_ROWS): 0 the library's choice, 5 / 6 rows per group of a dense view
    int view_places = 1;          // memo_index_set_option(MEMO_OPT_VIEW_PLACES): 0 the rows of a dense view keep the order they come in
    uint64_t view_placings = 0;   // dense views rebuilt with their rows placed (memo_index_info_t.view_placings)
    double view_ns_per_row[3] = {0, 0, 0};  // measured by the last view build ([0] dense rows, [1] 4-byte words, [2] dense rows with places; 0: the calibrated constant)
    uint64_t view_clock = 0;
    uint64_t view_builds = 0;     // views built over the index's lifetime (memo_index_info_t.view_builds)
    DenseView views[16];          // classes of two: overlaps below 2, 4, 6 ... 32
    DenseView views6[16];         // the same classes as groups of SIX rows that carry their bucket (memo_view.hip; what the table-driven sweep reads where it can)
    DenseView pviews[24];         // the same for the 4-byte words (caps 2 .. 32 by 2, .. 64 by 8, .. 128 by 16; `p3` holds words there): packed_rows_for
    uint64_t last_rows_read = 0;  // rows of the row source the last sweep read (info.last_rows_read)
    float last_view_ms = 0.f;     // device time of the view build, when the last sweep's view was built by it (else 0)
    int last_view_placed = 0, last_view_rpg = 5;  // (memo_index_info_t: of the dense rows the last sweep read)
    uint64_t max_annot = 0;    // largest annot of the packed rows
    // Sampled histogram of the packed rows' overlap field (min(end - start, 255)): with it the length n = k - 1 -
    // overlap of a row's interval is known in distribution for any k, which is what the choice between the level
    // arrays of the unclipped conservation sweeps turns on (memo_sweep_cons.hip, pick_levels).  Filled wherever
    // 4- / 6-byte rows come into being: memo_index_pack, memo_builder_finish, memo_index_import_packed.
    uint32_t len_hist[256] = {0};
    uint64_t len_hist_rows = 0;  // rows sampled (0: no histogram)
    // ... and which overlap values occur among ALL the rows (bit v of len_seen: some row has min(end - start, 255) == v), exact:
    // with it the lengths n = k - 1 - overlap that can occur at a k are known, hence which level arrays a sweep can skip
    uint32_t len_seen[8] = {0};
    int len_seen_exact = 0;
    // Tile tables of the table-driven dense-row sweep (memo_sweep_cons3t.hip): per (tile width, k) the row slice of every
    // tile of the chromosome, 32 bytes per tile, built by the first query that needs one and kept (least recently used of
    // four replaced); dropped with the dense rows.
    struct TileTable {
        const void *rows_of = nullptr;  // the dense rows (or view) the table was made for
        int w = 0, km1 = 0;
        void *d = nullptr;
        int64_t n = 0;
        uint64_t stamp = 0;
    };
    std::vector<TileTable> ttabs;  // one per (row source, tile width, k) in use, at most kMaxTileTables (least recently used out first)
    uint64_t ttab_clock = 0;
    // Device buffers a query took out of service (an evicted view, a tile table past the limit) while sweeps queued on any of
    // the caller's streams may still read them: they wait here -- no synchronisation on the query path -- and are freed by
    // the next call that has the device drained anyway (memo_query_check, memo_index_prepare, memo_index_pack*,
    // memo_index_destroy), or, past kRetiredLimit of the index's own bytes, by a query that then does wait for the device.
    struct Retired {
        void *p = nullptr;
        uint64_t bytes = 0;
    };
    std::vector<Retired> retired;
    uint64_t retired_bytes = 0;
    int last_sweep = 0;          // level arrays of the last conservation sweep (memo_index_info_t.last_sweep)
    int last_arrays = 0;         // mixed level arrays (last_sweep 4): how many of them the sweep's level plan allocated
    int last_variant = 0;        // ... 2 table-driven on five-row groups (memo_sweep_cons3t.hip), 3 on a view of six rows per group; 0 no table
    int has_wide = 1;          // the three int64 columns are still resident
    // rows with end < start (never written by the reference's index builder, but legal input to
    // memo_query.py): copied aside at finalize and applied by long_rows_kernel after each sweep
    int64_t *ls = nullptr, *le = nullptr, *lo = nullptr;
    uint64_t n_long = 0;
    // memo_multi.hip (resident form) sweeps a SUB-window of the caller's window on this index.  A row with end < start
    // can reach any distance left of its start, so it has to pass the reference's filter (memo_query.py:25-27) on the
    // WHOLE window, not on the sub-window: while whole_set, the long-row kernels filter by [whole_qs, whole_qe).
    int64_t whole_qs = 0, whole_qe = 0;
    int whole_set = 0;
    int *d_status = nullptr;   // sticky flags set by the sweep kernels
    uint64_t *d_scratch = nullptr;  // finalize(): [0] unsorted pairs, [1] rows with end < start
};

namespace memo {
void drop_dense(memo_index *ix);       // frees the dense rows, their bucket table and the tile tables
int dense_compact(memo_index *ix);     // memo_index.hip: leave the rows that can never write out of the dense rows (see boff3)
// ... or a k-class view of them (memo_view.hip); window: the query's length (what a view would save this query decides when it is built);
// allow_six: the caller reads views of six rows per group too (*rpg says which kind it got: 5 or 6); account = false: a query's
// SECOND call (its six-row view found no tile table): hands back a five-row view that exists, adds nothing to the class's ledgers
// and builds nothing -- the query has been counted once already
int dense_rows_for(memo_index *ix, int km1, int64_t window, hipStream_t st, uint32_t **p3, int64_t **boff, uint64_t *rows,
                   int *view_cap = nullptr, bool allow_six = false, int *rpg = nullptr, bool account = true);
extern thread_local int g_six_views;  // (AB library, memo_debug_six_views: -1 the library's choice, 0 five rows per group always, 1 six wherever they apply)
constexpr int kNoRoom = 1;  // (internal) the device has no memory for a view / tile table: run without it
constexpr size_t kMaxTileTables = 64;
void retire(memo_index *ix, void *p, uint64_t bytes);  // memo_index.hip: out of service now, freed once the device has drained
void flush_retired(memo_index *ix);                    // ... which the caller guarantees (it synchronised the device)
int builder_why(const memo_builder_t *b);  // memo_hostpack.hip: which rows a builder refused with MEMO_EUNPACKABLE (BlockResult::bad bits)
extern thread_local bool g_dense_keep_all;  // (AB library, memo_debug_dense_keep_all: dense_compact keeps every row)
extern thread_local int g_one_shot_way;     // (AB library, memo_debug_one_shot_way: 1 = int64 columns, 2 = 4-byte words)
extern thread_local bool g_prepare_only;  // memo_index_prepare: the query path builds what it would build and launches nothing
// ... and hands this as the result pointer: a launch site that missed g_prepare_only refuses it instead of writing to it (launch_tiles,
// launch_halo3t, the fill and long-row launches: memo_sweep.hip: refuse_plan_pointer)
static void *const kNeverWritten = reinterpret_cast<void *>(uintptr_t(4096));
int refuse_plan_pointer(const void *d_out);
extern thread_local bool g_side_alloc_fails;  // (AB library, memo_debug_fail_side_allocations: every side_alloc fails -- the test of kNoRoom)
hipError_t side_alloc(void **p, size_t bytes);
void drop_dense_views(memo_index *ix);
int packed_rows_for(memo_index *ix, int km1, int64_t window, bool membership, hipStream_t st, uint32_t **pk, int64_t **boff, uint64_t *rows);  // k-class view of the words (+ their order)
void drop_packed_views(memo_index *ix);
inline uint64_t dense_view_bytes(uint64_t padded, int rows_per_group) {  // what dense_view_build allocates for a view's rows
    return (rows_per_group == 6 ? padded / 6 + 64 : (padded + 4) / 5 + 64) * 16;
}
inline uint64_t dense_groups_for(uint64_t padded) { return (padded + 4) / 5 + 64; }  // (+ one wave-load of slack: a wave reads its 64 groups whole)
void drop_tile_tables(memo_index *ix);  // memo_sweep_cons3t.hip: the tables derive from the dense rows and the bucket table
// memo_interleave.hip: reorder the 4-byte rows inside every bucket (mode 0: start order, 1: chunks of four dealt round-robin
// over the bucket's starts, 2: the same with the rows of a start ordered by overlap mod 32), in place, queued on st
int interleave_words(uint32_t *words, const int64_t *boff, uint64_t nb, int bshift, int fmt, int mode, hipStream_t st, uint64_t *scratch);  // scratch: ix->d_scratch
extern thread_local int g_view_colouring;  // 1: the dense rows' k-class views may get their rows' places inside a group chosen against bank conflicts (memo_debug_view_colouring of the AB library turns it off)
constexpr int kRowOrderDefault = 2;  // interleave_words mode the product applies wherever 4-byte rows come into being
int order_words_now(memo_index *ix, int mode);  // memo_index.hip: waits for the device, orders ix->pk in place, waits again
inline int row_order_mode(const memo_index *ix) { return ix->tune.row_order ? ix->tune.row_order - 1 : kRowOrderDefault; }
extern thread_local int g_last_one_shot_sweep;  // which kernel family answered this thread's last one-shot call
}

// fills ix->len_hist from the resident 4- / 6-byte rows (a few thousand 1024-row blocks, evenly spread); NULL stream, synchronous
int memo_len_census(memo_index *ix);

#endif  // MEMO_COMMON_H
