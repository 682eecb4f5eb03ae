/*
 * memo_amd_debug.h -- A/B switches and profiling aids.  NOT part of the product ABI: these are
 * exported only by libmemo_amd_ab.so (the product objects + memo_amd/csrc/memo_debug.hip), which
 * tests/, tests/fuzz_gpu.py, tools/ab.py and the PMC calibration pass load instead of
 * libmemo_amd.so.  Results never depend on any of them; they replace nothing in the reference.
 */
#ifndef MEMO_AMD_DEBUG_H
#define MEMO_AMD_DEBUG_H

#include "memo_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel-shape choices of ONE index (0 = let the library choose, which is all the product does).
 * tile_w: positions per tile (256..4096; unclipped conservation: cells per level array, halo included);
 * waves: 1 or 4 waves share a tile (8: the unclipped conservation kernel only);
 * membership_algo: 2 = doubling, 3 = runs (clipped bit planes per genome + register transpose),
 *   4 = planes (unclipped, result staged through LDS; packed rows, <= 512 genomes; else 3);
 * row_source: 0 = the library's choice (conservation: the dense rows where they are resident and can answer,
 *   else the 4- / 6-byte rows, else the int64 columns), 1 = the int64 columns even when packed rows exist,
 *   2 = same as 0 (kept for older scripts), 3 = the 4- / 6-byte rows even where the dense rows could answer,
 *   5 = the dense rows with one workgroup per tile, every wave working its tile out (round 2's kernel: what a negative window start
 *   or a device without room for a tile table gets), 8 = the dense rows with one workgroup per tile that reads its row slice from the
 *   index's tile table (memo_sweep_cons3t.hip); 5 reads all the dense rows,
 *   0 and 8 the k-class view of them where one pays (memo_index_info_t.last_rows_read), 9 = as 0 without views, 10 = as 5
 *   with views, 13 = as 0 with the "this row writes" test kept in the row blocks even where the
 *   view holds exactly the rows that write (cap = k - 1; round 4);
 * scatter (conservation, packed rows): 1 = clip every interval to the tile, 2 = unclipped into doubling
 *   level arrays with a halo, 3 = unclipped into radix-4 level arrays, 4 = unclipped into mixed level arrays
 *   (blocks of 1, 4, 16, then doubling; k - 1 >= 16) with every array a k - 1 of that size can need (rounds 2-3), 5 = the mixed
 *   arrays with the library's level plan (only the arrays some row of the index can write to) -- 2 .. 5 only when every annot of
 *   the index is inside the result matrix, else 1. */
int memo_debug_set_tuning(memo_index_t *ix, int32_t tile_w, int32_t waves, int32_t membership_algo,
                          int32_t row_source, int32_t scatter);
/* Order of the 4-byte rows inside a start bucket (memo_amd/csrc/memo_interleave.hip): 0 = the library's choice (3 for conservation, 4 for membership: whichever kind of query pays for the pass), 1 = start order (as the packers write them), 2 = chunks of four rows dealt round-robin
 * over the bucket's starts, 3 = the same with the rows of a start ordered by overlap mod 32, 4 = chunks dealt over annot mod 32,
 * the rows of a class by their end (the order made for the membership planes: 6-7 % on a sequence-built index, level on config 4 -- profiles/r05_large_k.txt).  Applied at once to resident 4-byte rows (their k-class views are dropped) and by every
 * later memo_index_pack of this index.  Results never depend on it. */
int memo_debug_row_order(memo_index_t *ix, int32_t order);
/* 1 = this index's sweeps read all the rows of their format even where a k-class view is resident (views already built stay:
 * bench.py times the same index with and without); 0 = back to the library's choice.  (The product switch is
 * memo_index_set_option(MEMO_OPT_VIEWS), which also drops the views.) */
int memo_debug_no_views(memo_index_t *ix, int32_t on);
/* this THREAD's later builds of a dense k-class view: 0 = the rows keep the order they come in, whoever asks; 1 (the default) = the place
 * of a row inside its 16-byte group may be chosen against LDS bank conflicts (memo_view.hip: view_place_bucket; MEMO_OPT_VIEW_PLACES of
 * the index decides).  Results never depend on it. */
int memo_debug_view_colouring(int32_t on);
/* this THREAD's later conservation queries on dense rows: which kind of k-class view they build and read where views of SIX rows per
 * group apply (k - 1 <= 31, up to 255 genomes, buckets of 32 positions; memo_view.hip; info.last_variant 3): 1 = six wherever they apply,
 * 0 = five always, -1 (the default) = the library's choice (MEMO_OPT_VIEW_ROWS of the index, else six where the view holds enough rows
 * per bucket for the padding of every bucket to whole groups not to matter). */
int memo_debug_six_views(int32_t on);
/* this THREAD's later calls: every device allocation for a view or a tile table fails (the test of the no-memory path) */
int memo_debug_fail_side_allocations(int32_t on);
/* this thread's later memo_index_pack_dense / dense builders keep the rows that can never write at k <= 64 in the dense rows */
int memo_debug_dense_keep_all(int32_t on);
/* this thread's later one-shot calls (memo_conservation / memo_membership): 0 = the library's way in, 1 = int64 columns
 * uploaded as they are, 2 = 4-byte words even where the dense rows could answer */
int memo_debug_one_shot_way(int32_t way);
/* memo_index_info_t.last_sweep of the index this thread's last memo_conservation / memo_membership call built and
 * swept (the one-shot forms destroy their index before they return): which kernel family -- hence which row format --
 * answered it */
int memo_debug_last_one_shot_sweep(void);
/* one pass that reads the three int64 columns exactly once (24 B/row) with the sweep's access
 * shape, to calibrate the FETCH_SIZE counter on a known byte count */
int memo_debug_stream_rows(memo_index_t *ix, void *stream);
/* -DMEMO_STAMPS builds of the conservation sweep: a device buffer of 8 uint64 per workgroup that
 * receives the cycles wave 0 spent in each phase (NULL = off) */
int memo_debug_set_stamp_buffer(uint64_t *d_buffer);

#ifdef __cplusplus
}
#endif
#endif /* MEMO_AMD_DEBUG_H */
