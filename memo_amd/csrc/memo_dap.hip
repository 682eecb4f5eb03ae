// memo_dap.hip -- index-row construction on the GPU: document-array-profile rows -> MEMs /
// MEM-overlap rows, the (chr, start, end, annot) rows a MEMO index is made of.
//
// Counterpart of /root/reference/src/dap_to_bed.py:55-134 (class print_dap_as_mem_bed; --mem with
// optional --order and --overlap -- the two forms src/index.sh:88-102 uses, plus plain --mem).  The
// reference walks the DAP row by row in Python; here a chunk of rows is a [positions][columns] int32
// matrix in HBM and every column is an independent scan:
//
//   sort_rows_kernel      --order: each row sorted descending (bitonic network in LDS)      :85-91
//   segment_summary_kernel / carry_kernel / emit_kernel
//                         column c has a MEM starting at row p iff p opens its record or
//                         lcp[p-1][c] <= lcp[p][c] (:119-125).  What a MEM needs from the past is
//                         only the END of the previous MEM of its column, so the column is cut into
//                         segments: each reports its last MEM end, a short serial pass chains the
//                         segments, and the segments are replayed with their carry-in to produce
//                         per (row, column): the row to print or nothing                   :93-109
//                         and, after the last row of a record, the chr-end pseudo-MEM (len, 2 len)
//                         pushed through the same rule                              :126-128, :133
//   count_kernel + exclusive scan + write_kernel
//                         compaction into (record, start, end, annot) in the reference's print
//                         order: by row, then column, chr-end rows after their record's last row.
//
// State (previous row, previous MEM end per column) is carried across chunks, so a DAP of any
// length streams through in pieces.  Off the query hot path; the exclusive scan is rocPRIM's.
#include <cstring>  // rocprim's texture iterator calls memset without including it

#include <rocprim/rocprim.hpp>

#include <new>
#include <utility>
#include <vector>

#include "memo_common.h"

using namespace memo;

namespace {

constexpr int kNone = -1;   // "no MEM end yet" / "nothing to print"
constexpr int kSeg = 256;   // rows per scan segment

__global__ void locate_kernel(const int64_t *rec_begin, int nrec, int64_t g0, int64_t npos, int32_t *rec,
                              int32_t *rel, int32_t *len) {
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= npos) return;
    const int64_t g = g0 + p;
    int lo = 0, hi = nrec;  // last record with rec_begin <= g
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rec_begin[mid] <= g) lo = mid; else hi = mid;
    }
    rec[p] = lo;
    rel[p] = (int32_t)(g - rec_begin[lo]);
    len[p] = (int32_t)(rec_begin[lo + 1] - rec_begin[lo]);
}

// one workgroup per row; P2 = columns rounded up to a power of two, padding sorts to the end
__global__ void sort_rows_kernel(int32_t *M, int64_t npos, int C, int P2) {
    extern __shared__ int32_t v[];
    const int64_t p = blockIdx.x;
    int32_t *row = M + p * C;
    for (int i = threadIdx.x; i < P2; i += blockDim.x) v[i] = i < C ? row[i] : INT32_MIN;
    __syncthreads();
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < P2 / 2; t += blockDim.x) {
                const int i = 2 * t - (t & (j - 1));  // lower index of the pair, bit j clear
                const int l = i + j;
                const bool desc = (i & k) == 0;       // this run sorts descending
                const int32_t a = v[i], b = v[l];
                if (desc ? a < b : a > b) {
                    v[i] = b;
                    v[l] = a;
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < C; i += blockDim.x) row[i] = v[i];
}

struct ScanArgs {
    const int32_t *M;          // [npos][C], sorted already when --order
    const int32_t *carry_row;  // [C] last row of the previous chunk
    const int32_t *rel, *len;  // per row: position inside its record, record length
    int64_t npos;
    int C;
    int overlap;
};

__device__ __forceinline__ int32_t lcp_above(const ScanArgs &A, int64_t p, int c) {
    return p == 0 ? A.carry_row[c] : A.M[(p - 1) * A.C + c];
}

// pass 1: end of the last MEM that starts inside the segment, per column
__global__ void segment_summary_kernel(const ScanArgs A, int32_t *seg_last) {
    const int64_t p0 = blockIdx.x * (int64_t)kSeg, p1 = p0 + kSeg < A.npos ? p0 + kSeg : A.npos;
    for (int c = threadIdx.x; c < A.C; c += blockDim.x) {
        int32_t above = lcp_above(A, p0, c), last = kNone;
        for (int64_t p = p0; p < p1; ++p) {
            const int32_t m = A.M[p * A.C + c];
            if (A.rel[p] == 0 || above <= m) last = A.rel[p] + m;
            above = m;
        }
        seg_last[blockIdx.x * (int64_t)A.C + c] = last;
    }
}

// pass 2: chain the segments; prev_end carries over to the next chunk, so does the last row
__global__ void carry_kernel(const ScanArgs A, const int32_t *seg_last, int64_t nseg, int32_t *carry_in,
                             int32_t *prev_end, int32_t *carry_row_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= A.C) return;
    int32_t carry = prev_end[c];
    for (int64_t s = 0; s < nseg; ++s) {
        carry_in[s * A.C + c] = carry;
        const int32_t l = seg_last[s * A.C + c];
        if (l != kNone) carry = l;
    }
    prev_end[c] = carry;
    carry_row_out[c] = A.M[(A.npos - 1) * A.C + c];
}

// pass 3: E[2p][c] = end of the row printed for (row p, column c) or kNone; E[2p+1][c] = the same
// for the chr-end pseudo-MEM when p is the last row of its record
__global__ void emit_kernel(const ScanArgs A, const int32_t *carry_in, int32_t *E) {
    const int64_t p0 = blockIdx.x * (int64_t)kSeg, p1 = p0 + kSeg < A.npos ? p0 + kSeg : A.npos;
    for (int c = threadIdx.x; c < A.C; c += blockDim.x) {
        int32_t above = lcp_above(A, p0, c), carry = carry_in[blockIdx.x * (int64_t)A.C + c];
        for (int64_t p = p0; p < p1; ++p) {
            const int32_t m = A.M[p * A.C + c], rel = A.rel[p], L = A.len[p];
            const bool opener = rel == 0;
            int32_t e0 = kNone, e1 = kNone;
            if (opener || above <= m) {  // a MEM starts here: (rel, rel + m)
                const int32_t cur = rel + m;
                if (!A.overlap) {
                    e0 = cur;
                } else if (!opener && carry != kNone) {  // overlap with the previous MEM of the column
                    const int32_t e = carry < cur ? carry : cur;
                    if (e >= rel) e0 = e;
                }
                carry = cur;
            }
            if (rel == L - 1) {  // chr end: the pseudo-MEM (L, 2L) through the same printer
                if (!A.overlap) {
                    e1 = 2 * L;
                } else if (carry != kNone) {
                    const int32_t e = carry < 2 * L ? carry : 2 * L;
                    if (e >= L) e1 = e;
                }
            }
            above = m;
            E[(2 * p) * A.C + c] = e0;
            E[(2 * p + 1) * A.C + c] = e1;
        }
    }
}

// one wave per slot (2 per row): how many rows it prints
__global__ void count_kernel(const int32_t *E, int64_t nslots, int C, uint64_t *counts) {
    const int64_t slot = blockIdx.x * (int64_t)(blockDim.x / 64) + threadIdx.x / 64;
    if (slot >= nslots) return;
    const int lane = threadIdx.x & 63;
    unsigned n = 0;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        n += __popcll(__ballot(c < C && E[slot * C + c] != kNone));
    }
    if (lane == 0) counts[slot] = n;
}

__global__ void write_kernel(const int32_t *E, int64_t nslots, int C, const uint64_t *offsets,
                             const int32_t *rec, const int32_t *rel, const int32_t *len, int32_t *o_rec,
                             int64_t *o_start, int64_t *o_end, int32_t *o_annot) {
    const int64_t slot = blockIdx.x * (int64_t)(blockDim.x / 64) + threadIdx.x / 64;
    if (slot >= nslots) return;
    const int lane = threadIdx.x & 63;
    const int64_t p = slot >> 1;
    const int64_t start = (slot & 1) ? len[p] : rel[p];
    uint64_t at = offsets[slot];
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        const int32_t e = c < C ? E[slot * C + c] : kNone;
        const unsigned long long mask = __ballot(e != kNone);
        if (e != kNone) {
            const uint64_t k = at + __popcll(mask & ((1ull << lane) - 1));
            o_rec[k] = rec[p];
            o_start[k] = start;
            o_end[k] = e;
            o_annot[k] = c + 1;  // annots are 1-based (dap_to_bed.py:113,122)
        }
        at += __popcll(mask);
    }
}

template <typename T>
struct DevBuf {  // device buffer that only ever grows
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t need) {
        if (need <= cap) return MEMO_OK;
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
        HIP_TRY(hipMalloc(&p, need * sizeof(T)));
        cap = need;
        return MEMO_OK;
    }
};

}  // namespace

struct memo_dap {
    int device = 0, C = 0, nrec = 0, order = 0, overlap = 0;
    int64_t g = 0;      // global position of the next row
    int64_t total = 0;  // rec_begin[nrec]
    DevBuf<int64_t> rec_begin;
    std::vector<int64_t> h_rec_begin;
    DevBuf<int32_t> carry_row, carry_row_next, prev_end;
    // per-chunk buffers
    DevBuf<int32_t> M, E, rec, rel, len, seg_last, carry_in;
    DevBuf<uint64_t> counts, offsets;
    DevBuf<char> scan_tmp;
    // rows of the last push
    DevBuf<int32_t> o_rec, o_annot;
    DevBuf<int64_t> o_start, o_end;
    uint64_t n_out = 0;
};

extern "C" {

void memo_dap_destroy(memo_dap_t *h) {
    if (!h) return;
    DeviceGuard guard(h->device);
    void *bufs[] = {h->rec_begin.p, h->carry_row.p, h->carry_row_next.p, h->prev_end.p, h->M.p, h->E.p, h->rec.p,
                    h->rel.p, h->len.p, h->seg_last.p, h->carry_in.p, h->counts.p, h->offsets.p, h->scan_tmp.p,
                    h->o_rec.p, h->o_annot.p, h->o_start.p, h->o_end.p};
    for (void *b : bufs) (void)hipFree(b);
    delete h;
}

int memo_dap_create(int32_t columns, const int64_t *rec_begin, int32_t nrec, int32_t sort_order,
                    int32_t overlaps, int32_t device, memo_dap_t **out) {
    if (!out) return fail(MEMO_EINVAL, "out is NULL");
    *out = nullptr;
    if (columns < 1 || columns > 4096) return fail(MEMO_EINVAL, "columns must be in [1, 4096], got %d", columns);
    if (nrec < 1 || !rec_begin) return fail(MEMO_EINVAL, "need at least one record");
    for (int r = 0; r < nrec; ++r) {
        const int64_t L = rec_begin[r + 1] - rec_begin[r];
        if (L < 1 || L >= ((int64_t)1 << 30)) return fail(MEMO_EINVAL, "record %d has length %lld (need 1 .. 2^30-1)", r, (long long)L);
    }
    if (rec_begin[0] != 0) return fail(MEMO_EINVAL, "rec_begin[0] must be 0");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(MEMO_EHIP, "cannot select HIP device %d", device);
    memo_dap *h = new (std::nothrow) memo_dap();
    if (!h) return fail(MEMO_EHIP, "out of host memory");
    h->device = device;
    h->C = columns;
    h->nrec = nrec;
    h->order = sort_order != 0;
    h->overlap = overlaps != 0;
    h->h_rec_begin.assign(rec_begin, rec_begin + nrec + 1);
    h->total = rec_begin[nrec];
    int rc = h->rec_begin.ensure((size_t)nrec + 1);
    if (!rc) rc = h->carry_row.ensure((size_t)columns);
    if (!rc) rc = h->carry_row_next.ensure((size_t)columns);
    if (!rc) rc = h->prev_end.ensure((size_t)columns);
    hipError_t err = hipSuccess;
    if (!rc) err = hipMemcpy(h->rec_begin.p, rec_begin, (size_t)(nrec + 1) * sizeof(int64_t), hipMemcpyHostToDevice);
    if (!rc && err == hipSuccess) err = hipMemset(h->carry_row.p, 0, (size_t)columns * sizeof(int32_t));
    if (!rc && err == hipSuccess) err = hipMemset(h->prev_end.p, 0xFF, (size_t)columns * sizeof(int32_t));  // kNone
    if (rc || err != hipSuccess) {
        memo_dap_destroy(h);
        return rc ? rc : fail(MEMO_EHIP, "memo_dap_create: %s", hipGetErrorString(err));
    }
    *out = h;
    return MEMO_OK;
}

int memo_dap_push(memo_dap_t *h, const int32_t *lcp, int64_t positions, uint64_t *out_rows) {
    if (!h || !out_rows) return fail(MEMO_EINVAL, "NULL argument");
    *out_rows = 0;
    h->n_out = 0;
    if (positions <= 0) return MEMO_OK;
    if (!lcp) return fail(MEMO_EINVAL, "lcp is NULL");
    if (h->g + positions > h->total)
        return fail(MEMO_EINVAL, "DAP has more rows than the .fai has positions (%lld > %lld)",
                    (long long)(h->g + positions), (long long)h->total);
    DeviceGuard guard(h->device);
    hipStream_t st = nullptr;
    const int C = h->C;
    const int64_t npos = positions, nseg = (npos + kSeg - 1) / kSeg, nslots = 2 * npos;
    int rc;
    if ((rc = h->M.ensure((size_t)npos * C)) || (rc = h->E.ensure((size_t)nslots * C)) ||
        (rc = h->rec.ensure((size_t)npos)) || (rc = h->rel.ensure((size_t)npos)) || (rc = h->len.ensure((size_t)npos)) ||
        (rc = h->seg_last.ensure((size_t)nseg * C)) || (rc = h->carry_in.ensure((size_t)nseg * C)) ||
        (rc = h->counts.ensure((size_t)nslots + 1)) || (rc = h->offsets.ensure((size_t)nslots + 1)))
        return rc;
    HIP_TRY(hipMemcpyAsync(h->M.p, lcp, (size_t)npos * C * sizeof(int32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(locate_kernel, dim3((unsigned)((npos + 255) / 256)), dim3(256), 0, st, h->rec_begin.p, h->nrec,
                       h->g, npos, h->rec.p, h->rel.p, h->len.p);
    if (h->order && C > 1) {
        int p2 = 1;
        while (p2 < C) p2 <<= 1;
        const int threads = p2 / 2 < 64 ? 64 : (p2 / 2 > 1024 ? 1024 : p2 / 2);
        hipLaunchKernelGGL(sort_rows_kernel, dim3((unsigned)npos), dim3(threads), (size_t)p2 * sizeof(int32_t), st,
                           h->M.p, npos, C, p2);
    }
    const ScanArgs A{h->M.p, h->carry_row.p, h->rel.p, h->len.p, npos, C, h->overlap};
    const int threads = C <= 64 ? 64 : (C <= 128 ? 128 : 256);
    hipLaunchKernelGGL(segment_summary_kernel, dim3((unsigned)nseg), dim3(threads), 0, st, A, h->seg_last.p);
    hipLaunchKernelGGL(carry_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, st, A, h->seg_last.p, nseg,
                       h->carry_in.p, h->prev_end.p, h->carry_row_next.p);
    hipLaunchKernelGGL(emit_kernel, dim3((unsigned)nseg), dim3(threads), 0, st, A, h->carry_in.p, h->E.p);
    hipLaunchKernelGGL(count_kernel, dim3((unsigned)((nslots + 3) / 4)), dim3(256), 0, st, h->E.p, nslots, C,
                       h->counts.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(h->counts.p + nslots, 0, sizeof(uint64_t), st));  // scanned too: offsets[nslots] = total
    size_t tmp = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, h->counts.p, h->offsets.p, (uint64_t)0, (size_t)nslots + 1,
                                    rocprim::plus<uint64_t>(), st));
    if ((rc = h->scan_tmp.ensure(tmp ? tmp : 16))) return rc;
    HIP_TRY(rocprim::exclusive_scan(h->scan_tmp.p, tmp, h->counts.p, h->offsets.p, (uint64_t)0, (size_t)nslots + 1,
                                    rocprim::plus<uint64_t>(), st));
    uint64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, h->offsets.p + nslots, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if ((rc = h->o_rec.ensure((size_t)total)) || (rc = h->o_annot.ensure((size_t)total)) ||
        (rc = h->o_start.ensure((size_t)total)) || (rc = h->o_end.ensure((size_t)total)))
        return rc;
    if (total)
        hipLaunchKernelGGL(write_kernel, dim3((unsigned)((nslots + 3) / 4)), dim3(256), 0, st, h->E.p, nslots, C,
                           h->offsets.p, h->rec.p, h->rel.p, h->len.p, h->o_rec.p, h->o_start.p, h->o_end.p,
                           h->o_annot.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    std::swap(h->carry_row, h->carry_row_next);
    h->g += npos;
    h->n_out = total;
    *out_rows = total;
    return MEMO_OK;
}

int memo_dap_fetch(memo_dap_t *h, int32_t *rec, int64_t *start, int64_t *end, int32_t *annot) {
    if (!h) return fail(MEMO_EINVAL, "handle is NULL");
    if (!h->n_out) return MEMO_OK;
    if (!rec || !start || !end || !annot) return fail(MEMO_EINVAL, "output pointer is NULL");
    DeviceGuard guard(h->device);
    HIP_TRY(hipMemcpy(rec, h->o_rec.p, h->n_out * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(start, h->o_start.p, h->n_out * sizeof(int64_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(end, h->o_end.p, h->n_out * sizeof(int64_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(annot, h->o_annot.p, h->n_out * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MEMO_OK;
}

// End of the DAP.  If it stopped inside a record (the reference still prints that record's chr-end
// rows, dap_to_bed.py:133-134), they are produced here: at most `columns` rows, straight into the
// caller's arrays.  Returns their number in out_rows.
int memo_dap_finish(memo_dap_t *h, int32_t *rec, int64_t *start, int64_t *end, int32_t *annot, uint64_t *out_rows) {
    if (!h || !out_rows) return fail(MEMO_EINVAL, "NULL argument");
    *out_rows = 0;
    if (h->g == 0) return MEMO_OK;  // empty DAP: the reference fails on an unbound name there
    int r = 0;                      // record of the last row
    while (r + 1 < h->nrec && h->h_rec_begin[r + 1] <= h->g - 1) ++r;
    if (h->h_rec_begin[r + 1] == h->g) return MEMO_OK;  // the last record was complete: already printed
    if (!rec || !start || !end || !annot) return fail(MEMO_EINVAL, "output pointer is NULL");
    DeviceGuard guard(h->device);
    std::vector<int32_t> pe(h->C);
    HIP_TRY(hipMemcpy(pe.data(), h->prev_end.p, (size_t)h->C * sizeof(int32_t), hipMemcpyDeviceToHost));
    const int64_t L = h->h_rec_begin[r + 1] - h->h_rec_begin[r];
    uint64_t n = 0;
    for (int c = 0; c < h->C; ++c) {
        int64_t e;
        if (!h->overlap) {
            e = 2 * L;
        } else {
            if (pe[c] == kNone) continue;
            e = pe[c] < 2 * L ? pe[c] : 2 * L;
            if (e < L) continue;
        }
        rec[n] = r;
        start[n] = L;
        end[n] = e;
        annot[n] = c + 1;
        ++n;
    }
    *out_rows = n;
    return MEMO_OK;
}

}  // extern "C"
