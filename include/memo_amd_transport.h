/*
 * memo_amd_transport.h -- lossless transport codings of conservation slices (uint8; the runs coding also uint16) for the multi-GPU gather
 * (new: the reference is single-process).
 * Part of the C ABI of libmemo_amd.so (see memo_amd.h for conventions: plain C types, 0 or a negative
 * code, memo_last_error()).
 */
#ifndef MEMO_AMD_TRANSPORT_H
#define MEMO_AMD_TRANSPORT_H

#include "memo_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

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

/* Third coding, "runs": one bit per position (1 = the value differs from the one before; the first position of
 * every 32768 is marked always) + one byte per marked position, allocated exactly per 32768 positions from a B
 * region of `b_capacity` bytes (a multiple of 4).  A conservation value changes at one position in ten on BASELINE
 * config 3 (k = 31): 1.8 bits per position, no escapes, no exception list.  Buffers 16-byte aligned.  _stats
 * (synchronising `stream`) returns the B bytes taken and the B capacity: the slice is complete iff taken <=
 * capacity.  Pack once with a generous capacity to learn what a workload needs (at most n + 4 * ceil(n / 32768)). */
size_t memo_transport_runs_bytes(int64_t n, uint32_t b_capacity);
int memo_transport_runs_pack_dev(const uint8_t *d_vec, int64_t n, uint32_t b_capacity, void *d_wire, int32_t device,
                                 void *stream);
int memo_transport_runs_unpack_dev(const void *d_wire, int64_t n, uint32_t b_capacity, uint8_t *d_vec, int32_t device,
                                   void *stream);
int memo_transport_runs_stats(const void *d_wire, int32_t device, void *stream, uint32_t *b_taken, uint32_t *b_capacity);
/* The runs coding of uint16 results (more than 255 genomes: BASELINE config 5): the same streams with TWO bytes per marked
 * position in the B region (at most 2 n + 4 * ceil(n / 32768) bytes of it); memo_transport_runs_bytes and _stats serve both.
 * A config-5 slice (500 genomes, 2^25 positions, k = 31) is 67 MB as plain uint16 and ~9 MB coded. */
int memo_transport_runs16_pack_dev(const uint16_t *d_vec, int64_t n, uint32_t b_capacity, void *d_wire, int32_t device,
                                   void *stream);
/* The slices of one gather step -- count wires of the same length n and capacity, as rank 0 receives them from its peers --
 * decoded by ONE launch into count result vectors (value_bytes 1: uint8 as memo_transport_runs_unpack_dev, 2: uint16 as
 * ..._runs16_...): a slice is too short to fill the device, and count launches one after the other cost count launch-and-drain
 * times on the rank every peer waits for.  d_wires / d_vecs: HOST arrays of count device pointers. */
int memo_transport_runs_unpack_many_dev(const void *const *d_wires, void *const *d_vecs, int32_t count, int64_t n, uint32_t b_capacity,
                                        int32_t value_bytes, int32_t device, void *stream);
int memo_transport_runs16_unpack_dev(const void *d_wire, int64_t n, uint32_t b_capacity, uint16_t *d_vec, int32_t device,
                                     void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MEMO_AMD_TRANSPORT_H */
