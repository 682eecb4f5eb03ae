/*
 * memo_amd_multi.h -- several GPUs of one node from ONE process (8b's proposed
 * `devices, n_devices`).  New: the reference is single-process.  The query window is cut into contiguous
 * sub-windows; sub-window [a, b) needs exactly the rows a < start < b + k (the reference's own filter,
 * memo_query.py:25-27 with :100, applied to the sub-window), every GPU runs the single-GPU sweep
 * unchanged, the disjoint result slices are delivered to one place.  Results are bit-identical to the
 * single-GPU calls.  Part of the C ABI of libmemo_amd.so (conventions: memo_amd.h).
 */
#ifndef MEMO_AMD_MULTI_H
#define MEMO_AMD_MULTI_H

#include "memo_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The one partition rule (memo_amd/shard.py calls it too): cuts[0] = qs <= cuts[1] <= ... <= cuts[parts] = qe.
 * Part 0 gets `first_weight` shares of the window, every other part one share (1.0 = equal parts; less
 * when the root also gathers and decodes; 0 = the root only gathers).  Part lengths are multiples of
 * `align` positions (rounded up; 8 keeps uint16 slices 16-byte aligned), the last non-empty part takes
 * what is left, parts past the end are empty.  Pure host arithmetic: needs no GPU. */
int memo_split_window(int64_t qs, int64_t qe, int32_t parts, int32_t align, double first_weight, int64_t *cuts);

/* Host form: memo_conservation / memo_membership (the drop-in for memo_query.py:103-104 + :70) with the
 * window split over `devices`.  Rows must be start-sorted for the split (else, or when the window is
 * shorter than 8 positions per device, the call runs on devices[0] alone).  One host thread per device;
 * each device's slice returns over its own PCIe link into its part of `out` -- no GPU-to-GPU traffic. */
int memo_conservation_multi(const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows,
                            int64_t qs, int64_t qe, int32_t k, int32_t num_docs, uint16_t *out,
                            const int32_t *devices, int32_t n_devices);
int memo_membership_multi(const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows,
                          int64_t qs, int64_t qe, int32_t k, int32_t num_docs, uint32_t *out_bits,
                          const int32_t *devices, int32_t n_devices);

/* Resident form: shards[g] is an index resident on ITS device that holds (at least) the rows sub-window g
 * of memo_split_window(qs, qe, n_shards, 8, root_weight) needs -- a replica of the chromosome on every
 * GPU always does (2 GB of packed rows for 5 * 10^8 rows), a position-sharded index does for windows
 * inside its cuts when it also holds every row of the window with end < start (those reach any distance left
 * of their start; they are filtered by the whole window [qs, qe), not by the sub-window).  Every device sweeps its sub-window on a stream of its own; hipMemcpyPeerAsync
 * delivers the slice into d_out on `root_device` (over xGMI every peer has its own link to the root);
 * `root_stream` waits for all slices, so the result is complete in its order.  The root's own slice is
 * swept straight into d_out.  Check every shard with memo_query_check afterwards. */
int memo_query_conservation_multi_dev(memo_index_t *const *shards, int32_t n_shards, int64_t qs, int64_t qe,
                                      int32_t k, int32_t num_docs, uint16_t *d_out, int32_t root_device,
                                      void *root_stream, double root_weight);
int memo_query_membership_multi_dev(memo_index_t *const *shards, int32_t n_shards, int64_t qs, int64_t qe,
                                    int32_t k, int32_t num_docs, uint32_t *d_out, int32_t root_device,
                                    void *root_stream, double root_weight);

#ifdef __cplusplus
}
#endif
#endif /* MEMO_AMD_MULTI_H */
