// memo_sweep.hip -- the hot path's launch plumbing: tile geometry, tuning state, query checks.
// The kernels are in memo_sweep_cons.hip (conservation) and memo_sweep_memb.hip (membership);
// what follows describes all three.
//
// Replaces /root/reference/src/memo_query.py:42-63 and :70 (memo_init + memo_query + the argmax
// reduction of print_res).  DESIGN.md section 3 has the algorithm; in short:
//
//   * a workgroup of one or four 64-lane waves owns one TILE of W consecutive pivot positions
//     and never talks to another workgroup;
//   * the rows that can touch the tile are a contiguous slice of the start-sorted index, found
//     with two loads from a bucket table built once per index; they are streamed from HBM with
//     16-byte-per-lane loads, either as the three int64 columns (WideRows) or as the packed
//     4/6-byte rows of memo_index_pack (PackedRows);
//   * conservation: each row's interval is covered by two power-of-two blocks (one ds_min_u32
//     each into the level-log2 array); the levels are then folded top-down so that level 0 holds
//     min(order) per position.  Packed rows: no clipping -- the level arrays carry a halo wide
//     enough for any row of the slice (sweep_conservation_halo_kernel); int64 rows, sparse or
//     checked indexes: intervals clipped to the tile (sweep_conservation_kernel);
//   * membership: per-genome bit planes (a row is one run of bits) + an in-register 32 x 32 bit
//     transpose -- unclipped, with the result staged through LDS for whole-line stores
//     (sweep_membership_planes_kernel), or clipped (..._runs_kernel); or the doubling scheme on
//     bit cells (int64 rows);
//   * min / or are idempotent and commutative: overlapping blocks and the arrival order of the
//     atomics cannot change a bit of the result.
//
// Integer work only: no MFMA.  Bound: HBM bandwidth -- all three default kernels run at 5.7-6.2 TB/s
// of algorithmic traffic on a good device, where an in-order sweep of this part reaches 6.0-6.3.
#include <map>
#include <utility>

#include "memo_sweep.h"

namespace memo {

#ifdef MEMO_STAMPS
unsigned long long *g_stamp_buffer = nullptr;  // diagnostic builds: 8 words per workgroup
#endif

// tiles are aligned in pivot coordinates: tile 0 starts at floor(qs / w) * w  (w: any multiple of
// the bucket width).  One workgroup per tile.
int launch_tiles(SweepKernel kernel, SweepArgs &A, int w, int threads, size_t lds, hipStream_t st) {
    int64_t q = A.qs / w;
    if (A.qs % w < 0) --q;  // floor
    A.tile0 = q * w;
    A.ntiles = ((A.qe - A.tile0) + w - 1) / w;
    A.tiles_per_xcd = (A.ntiles + 7) / 8;
    if (lds > 160 * 1024) return fail(MEMO_EINVAL, "tile needs %zu bytes of LDS (> 160 KiB)", lds);
    if (lds > 64 * 1024) {  // opt in to large dynamic LDS once per (thread, device, kernel, size)
        thread_local std::map<std::pair<const void *, int>, size_t> granted;
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        size_t &have = granted[{reinterpret_cast<const void *>(kernel), dev}];
        if (have < lds) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            have = lds;
        }
    }
    if (A.tiles_per_xcd * 8 * threads >= ((int64_t)1 << 32))
        return fail(MEMO_EINVAL, "window too long for one launch at tile width %d", w);
    if (g_prepare_only) return MEMO_OK;  // memo_index_prepare: everything a query builds on the way exists now; nothing is launched
    if (int rc = refuse_plan_pointer(A.out)) return rc;
    hipLaunchKernelGGL(kernel, dim3((unsigned)(A.tiles_per_xcd * 8)), dim3(threads), lds, st, A);
    HIP_TRY(hipGetLastError());
    return MEMO_OK;
}

// which row source a query reads: packed when the index has it and k - 1 <= 255, else the int64 columns.
// fmt: 0 = int64 columns, 4 / 6 = packed words (+ 16-bit order column), 3 = only the dense rows are left
// (memo_index_pack_dense dropped the words): k - 1 <= 63, and only kernels that read PackedRows3
thread_local bool g_prepare_only = false;
int refuse_plan_pointer(const void *d_out) {  // (a launch must never see memo_index_prepare's stand-in for a result)
    return d_out == kNeverWritten ? fail(MEMO_EHIP, "internal: a sweep was about to be launched into memo_index_prepare's placeholder result") : MEMO_OK;
}
thread_local bool g_side_alloc_fails = false;
thread_local int g_view_colouring = 1;
thread_local int g_six_views = -1;

int pick_rows(const memo_index *ix, int32_t k, int &fmt) {
    fmt = 0;
    if (ix->packed_fmt && k - 1 <= 255 && !(ix->tune.force_wide && ix->has_wide)) fmt = ix->pk ? ix->packed_fmt : 0;
    if (!fmt && ix->p3 && k - 1 <= 63 && !(ix->tune.force_wide && ix->has_wide)) fmt = 3;
    if (!fmt && !ix->has_wide)
        return fail(MEMO_EINVAL, "k = %d needs the int64 columns, which this index dropped when it was packed", k);
    return MEMO_OK;
}

int check_query_args(const memo_index *ix, int64_t qs, int64_t qe, int32_t k, int32_t num_docs,
                     const void *d_out) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (!ix->finalized) return fail(MEMO_ENOTREADY, "index not finalized");
    if (num_docs < 1 || num_docs > 65534)
        return fail(MEMO_EINVAL, "num_docs must be in [1, 65534], got %d", num_docs);
    if (qs <= -kCoordLimit || qe >= kCoordLimit || qs >= kCoordLimit || qe <= -kCoordLimit)
        return fail(MEMO_EINVAL, "window coordinates out of range");
    if (k >= (1 << 30) || k <= -(1 << 30)) return fail(MEMO_EINVAL, "k out of range");
    if (qe < qs)  // np.zeros([true_len, ...]) with true_len < 0 (memo_query.py:51,53)
        return fail(MEMO_EINVAL, "ValueError: negative dimensions are not allowed (window end < start)");
    if (qe > qs && !d_out) return fail(MEMO_EINVAL, "output pointer is NULL");
    if (qe > qs && ((uintptr_t)d_out & 15)) return fail(MEMO_EINVAL, "output must be 16-byte aligned");
    if (qe - qs > ((int64_t)1 << 40)) return fail(MEMO_EINVAL, "window longer than 2^40");
    return MEMO_OK;
}

void fill_args(const memo_index *ix, SweepArgs &A, int64_t qs, int64_t qe, int32_t k, void *d_out) {
    A.s = ix->s;
    A.e = ix->e;
    A.o = ix->o;
    A.pk = ix->pk;
    A.p3 = ix->p3;
    A.pa = ix->pa;
    A.boff = ix->boff;
    A.nb = (int64_t)ix->nb;
    A.bbase = ix->bbase;
    A.qs = qs;
    A.qe = qe;
    A.out = d_out;
    A.status = ix->d_status;
#ifdef MEMO_STAMPS
    A.stamps = g_stamp_buffer;
#else
    A.stamps = nullptr;
#endif
    A.bshift = ix->bshift;
    A.km1 = k - 1;
}


}  // namespace memo

using namespace memo;

extern "C" {

int memo_query_check(memo_index_t *ix, void *stream) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    DeviceGuard guard(ix->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int flags = 0;
    HIP_TRY(hipMemcpyAsync(&flags, ix->d_status, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (!ix->retired.empty()) {  // buffers earlier queries took out of service: the caller is waiting anyway, so wait for
        HIP_TRY(hipDeviceSynchronize());  // whatever its other streams still run on this index, and free them
        flush_retired(ix);
    }
    if (flags) {
        HIP_TRY(hipMemsetAsync(ix->d_status, 0, sizeof(int), st));
        HIP_TRY(hipStreamSynchronize(st));
        if (flags & kStatusHugeSlice)
            return fail(MEMO_EINVAL, "more than 2^32 index rows can reach one tile of the window: unsupported");
        if (flags & kStatusExecNarrow)
            return fail(MEMO_EINVAL, "EXEC check: a branch-free row block was entered with lanes disabled (its s_mov_b64 exec, -1 "
                                     "would have enabled them): a compiler change broke the blocks' invariant");
        if (flags & kStatusBadAnnot)
            return fail(MEMO_EINVAL,
                        "a row that covers the window has an order/genome column outside the "
                        "result matrix (num_docs too small?) -- the reference raises IndexError here");
        return fail(MEMO_EINVAL, "device status 0x%x", flags);
    }
    return MEMO_OK;
}

}  // extern "C"
