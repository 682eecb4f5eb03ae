/*
 * memo_amd_dap.h -- index-row construction (the dap_to_bed.py step of `memo index`), off the query path.
 * Part of the C ABI of libmemo_amd.so (see memo_amd.h for conventions: plain C types, 0 or a negative
 * code, memo_last_error()).
 */
#ifndef MEMO_AMD_DAP_H
#define MEMO_AMD_DAP_H

#include "memo_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

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


#ifdef __cplusplus
}
#endif
#endif /* MEMO_AMD_DAP_H */
