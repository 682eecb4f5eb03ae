// memo_hostpack.hip -- the fast way in for HOST rows, device side: the HIP implementation of the seam the host core
// is written against (memo_hostcore.h: pinned memory, copy stream, events), the builder's create / finish (device
// allocations, the index it hands over) and the export / import of packed and dense rows (the CLI's sidecar cache).
// The threaded host code -- worker pool, pinned ring, the row packers, the push loop -- is memo_hostcore.cpp, which
// knows nothing of HIP and therefore also runs under the CPU sanitizers (tests/test_host_sanitizers.py).
#include <new>

#include "memo_common.h"
#include "memo_cpus.h"
#include "memo_hostcore.h"

using namespace memo;

namespace {

// memo_index_import_packed: the uploaded slice of an absolute bucket table -> this index's table: entries rebased
// to the slice's first row, one more entry pinned to its row count
__global__ void rebase_table_kernel(int64_t *boff, uint64_t buckets, int64_t row_base, int64_t rows) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i + 1 < buckets) boff[i] -= row_base;
    else if (i + 1 == buckets) boff[i] = rows;
}

// rows uploaded in format 4 (8-bit annots) when the first annot > 255 shows up: rewrite them in format 12
// (start16 | len8 << 16 | annot8 << 24  ->  len8 | start12 << 8 | annot12 << 20)
__global__ void widen_annot_kernel(uint32_t *pk, uint64_t rows) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < rows;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t w = pk[i];
        pk[i] = ((w >> 16) & 0xFFu) | ((w & 0xFFFu) << 8) | ((w >> 24) << 20);
    }
}

int hip_rc(hipError_t err, const char *what) {
    return err == hipSuccess ? MEMO_OK : fail(MEMO_EHIP, "%s: %s", what, hipGetErrorString(err));
}

}  // namespace

// ---- the device seam of memo_hostcore.h, on HIP ------------------------------------------------------------
namespace memo {
namespace hp {
int set_device(int device, int *previous) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) cur = -1;
    if (previous) *previous = cur == device ? -1 : cur;
    return cur == device ? MEMO_OK : hip_rc(hipSetDevice(device), "hipSetDevice");
}
int pinned_alloc(void **p, size_t bytes) { return hip_rc(hipHostMalloc(p, bytes, hipHostMallocDefault), "pinned staging buffer"); }
void pinned_free(void *p) { (void)hipHostFree(p); }
int stream_create(void **stream) {
    return hip_rc(hipStreamCreateWithFlags(reinterpret_cast<hipStream_t *>(stream), hipStreamNonBlocking), "hipStreamCreate");
}
void stream_destroy(void *stream) { (void)hipStreamDestroy(static_cast<hipStream_t>(stream)); }
int stream_sync(void *stream) { return hip_rc(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "hipStreamSynchronize"); }
int event_create(void **event) {
    return hip_rc(hipEventCreateWithFlags(reinterpret_cast<hipEvent_t *>(event), hipEventDisableTiming), "hipEventCreate");
}
void event_destroy(void *event) { (void)hipEventDestroy(static_cast<hipEvent_t>(event)); }
int event_record(void *event, void *stream) {
    return hip_rc(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)), "hipEventRecord");
}
int event_sync(void *event) { return hip_rc(hipEventSynchronize(static_cast<hipEvent_t>(event)), "hipEventSynchronize"); }
int event_query(void *event, int *done) {
    const hipError_t err = hipEventQuery(static_cast<hipEvent_t>(event));
    *done = err == hipSuccess;
    return err == hipSuccess || err == hipErrorNotReady ? MEMO_OK : hip_rc(err, "hipEventQuery");
}
int copy_h2d_async(void *dev, const void *host, size_t bytes, void *stream) {
    return hip_rc(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)), "hipMemcpyAsync H2D");
}
int copy_d2h_async(void *host, const void *dev, size_t bytes, void *stream) {
    return hip_rc(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)), "hipMemcpyAsync D2H");
}
int copy_h2d(void *dev, const void *host, size_t bytes) { return hip_rc(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice), "hipMemcpy H2D"); }
int widen_annots(uint32_t *d_pk, uint64_t rows, void *stream) {
    hipLaunchKernelGGL(widen_annot_kernel, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), d_pk, rows);
    return hip_rc(hipGetLastError(), "widen_annot_kernel");
}
}  // namespace hp

// device -> pageable host memory through the pinned ring; work already queued on `producer` finishes first
int download_pipelined(int device, void *host, const void *dev, size_t bytes, hipStream_t producer) {
    {
        DeviceGuard guard(device);
        HIP_TRY(hipStreamSynchronize(producer));
    }
    return download_pipelined_core(device, host, dev, bytes);
}
int upload_pipelined(int device, void *dev, const void *host, size_t bytes) { return upload_pipelined_core(device, dev, host, bytes); }
}  // namespace memo

// ------------------------------------------------------------------------------------------
// the builder
// ------------------------------------------------------------------------------------------

namespace memo {
int builder_why(const memo_builder_t *b) { return b ? b->why : 0; }
}  // namespace memo

extern "C" {

int memo_host_threads(int32_t *cpus_allowed_out, double *cgroup_quota_cpus) {
    if (cpus_allowed_out) *cpus_allowed_out = cpus_allowed();
    if (cgroup_quota_cpus) *cgroup_quota_cpus = cgroup_cpu_quota();
    return host_threads_default();
}

int memo_builder_create_rows(uint64_t max_rows, int32_t device, int32_t bucket_shift, int32_t row_format,
                             memo_builder_t **out) {
    if (!out) return fail(MEMO_EINVAL, "out is NULL");
    *out = nullptr;
    if (max_rows > ((uint64_t)1 << 40)) return fail(MEMO_EINVAL, "too many rows");
    if (row_format != MEMO_ROWS_PACKED && row_format != MEMO_ROWS_DENSE)
        return fail(MEMO_EINVAL, "row_format must be MEMO_ROWS_PACKED (0) or MEMO_ROWS_DENSE (1)");
    if (bucket_shift <= 0) bucket_shift = kDefaultBucketShift;
    if (bucket_shift > 8) return fail(MEMO_EINVAL, "bucket_shift must be <= 8 (tile width 256)");
    const int ndev = memo_device_count();
    if (device < 0 || device >= ndev)
        return fail(MEMO_EHIP, "HIP device %d not available (%d visible)", device, ndev);
    DeviceGuard guard(device);
    if (!guard.ok) return fail(MEMO_EHIP, "cannot select HIP device %d", device);
    memo_builder *b = new (std::nothrow) memo_builder();
    if (!b) return fail(MEMO_EHIP, "out of host memory");
    b->device = device;
    b->bshift = bucket_shift;
    b->cap = max_rows;
    b->dense = row_format == MEMO_ROWS_DENSE;
    b->padded = ((max_rows + 15) & ~(uint64_t)15) + kPadRows;
    b->d_groups = b->dense ? dense_groups_for(b->padded) : 0;
    const size_t bytes = b->dense ? (size_t)b->d_groups * 16 : (size_t)b->padded * sizeof(uint32_t);
    hipError_t err = hipMalloc(&b->d_pk, bytes);
    if (err != hipSuccess) {
        delete b;
        return fail(MEMO_EHIP, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(err));
    }
    int rc = acquire_ring(device, &b->ring);
    if (rc) {
        (void)hipFree(b->d_pk);
        delete b;
        return rc;
    }
    *out = b;
    return MEMO_OK;
}

int memo_builder_create(uint64_t max_rows, int32_t device, int32_t bucket_shift, memo_builder_t **out) {
    return memo_builder_create_rows(max_rows, device, bucket_shift, MEMO_ROWS_PACKED, out);
}

void memo_builder_destroy(memo_builder_t *b) {
    if (!b) return;
    DeviceGuard guard(b->device);
    release_ring(b->ring);
    (void)hipFree(b->d_pk);
    delete b;
}

int memo_builder_push(memo_builder_t *b, const int64_t *start, const int64_t *end, const int64_t *annot,
                      uint64_t rows) {
    if (!b) return fail(MEMO_EINVAL, "builder is NULL");
    if (b->failed) return fail(b->failed, "the builder already failed");
    if (!rows) return MEMO_OK;
    if (!start || !end || !annot) return fail(MEMO_EINVAL, "column pointer is NULL");
    if (rows > b->cap - b->rows)
        return fail(MEMO_EINVAL, "%llu more rows do not fit a builder of %llu", (unsigned long long)rows,
                    (unsigned long long)b->cap);
    DeviceGuard guard(b->device);
    return builder_push_core(b, start, end, annot, rows);  // (every error in there marks the builder failed)
}

int memo_builder_push_rows(memo_builder_t *b, const int64_t *rows3, uint64_t rows) {
    if (!b) return fail(MEMO_EINVAL, "builder is NULL");
    if (b->failed) return fail(b->failed, "the builder already failed");
    if (!rows) return MEMO_OK;
    if (!rows3) return fail(MEMO_EINVAL, "rows pointer is NULL");
    if (rows > b->cap - b->rows)
        return fail(MEMO_EINVAL, "%llu more rows do not fit a builder of %llu", (unsigned long long)rows,
                    (unsigned long long)b->cap);
    DeviceGuard guard(b->device);
    return builder_push_core(b, rows3, rows3 + 1, rows3 + 2, rows, 3);
}

int memo_builder_finish(memo_builder_t *b, memo_index_t **out) {
    if (!b || !out) return fail(MEMO_EINVAL, "NULL argument");
    *out = nullptr;
    if (b->failed) return fail(b->failed, "the builder already failed");
    DeviceGuard guard(b->device);
    if (int rc = builder_flush_core(b)) return rc;
    memo_index *ix = new (std::nothrow) memo_index();
    if (!ix) return fail(MEMO_EHIP, "out of host memory");
    ix->device = b->device;
    ix->rows = b->rows;
    ix->padded = b->padded;
    ix->packed_rows = b->dense ? 0 : b->padded;
    ix->has_wide = 0;
    ix->was_sorted = 1;
    ix->bshift = b->bshift;
    ix->min_s = b->any ? b->first_start : 0;
    ix->max_s = b->any ? b->last_start : -1;
    ix->max_annot = b->max_annot;
    // buckets 0 .. ceil((max_s + 1) / width), plus one pinned to `rows` (as bucket_table_kernel builds them)
    const int64_t top = ix->max_s < 0 ? 0 : ix->max_s;
    const uint64_t nb = (uint64_t)((top >> b->bshift) + 3);
    if (b->boff.size() < nb) b->boff.resize(nb);
    for (int64_t q = b->last_bucket + 1; q < (int64_t)nb; ++q) b->boff[(size_t)q] = (int64_t)b->rows;
    int rc = MEMO_OK;
    hipStream_t st = static_cast<hipStream_t>(b->ring->stream);
    do {
        hipError_t err = hipMalloc(&ix->boff, nb * sizeof(int64_t));
        if (err == hipSuccess) err = hipMalloc(&ix->d_status, 64);
        if (err == hipSuccess) err = hipMalloc(&ix->d_scratch, 64);
        if (err == hipSuccess) err = hipMemsetAsync(ix->d_status, 0, 64, st);
        if (err == hipSuccess)
            err = hipMemcpyAsync(ix->boff, b->boff.data(), nb * sizeof(int64_t), hipMemcpyHostToDevice, st);
        // the rows behind the last one are read (never used) by whole-wave loads: keep them defined
        if (err == hipSuccess) {
            if (b->dense)
                err = hipMemsetAsync(b->d_pk + 4 * b->groups_sent, 0, (size_t)(b->d_groups - b->groups_sent) * 16, st);
            else
                err = hipMemsetAsync(b->d_pk + b->rows, 0, (b->padded - b->rows) * 4, st);
        }

        const uint64_t n_long = b->long_rows.size() / 3;
        std::vector<int64_t> cols;
        if (err == hipSuccess && n_long) {
            cols.resize(3 * n_long);
            for (uint64_t i = 0; i < n_long; ++i) {
                cols[i] = b->long_rows[3 * i];
                cols[n_long + i] = b->long_rows[3 * i + 1];
                cols[2 * n_long + i] = b->long_rows[3 * i + 2];
            }
            err = hipMalloc(&ix->ls, n_long * 8);
            if (err == hipSuccess) err = hipMalloc(&ix->le, n_long * 8);
            if (err == hipSuccess) err = hipMalloc(&ix->lo, n_long * 8);
            if (err == hipSuccess) err = hipMemcpyAsync(ix->ls, cols.data(), n_long * 8, hipMemcpyHostToDevice, st);
            if (err == hipSuccess) err = hipMemcpyAsync(ix->le, cols.data() + n_long, n_long * 8, hipMemcpyHostToDevice, st);
            if (err == hipSuccess) err = hipMemcpyAsync(ix->lo, cols.data() + 2 * n_long, n_long * 8, hipMemcpyHostToDevice, st);
            ix->n_long = n_long;
        }
        if (err == hipSuccess) err = hipStreamSynchronize(st);
        if (err != hipSuccess) {
            rc = fail(MEMO_EHIP, "finishing the packed index: %s", hipGetErrorString(err));
            break;
        }
    } while (0);
    if (rc) {
        memo_index_destroy(ix);
        return rc;
    }
    for (int s = 0; s < PinnedRing::kSlots; ++s) b->ring->in_flight[s] = false;
    ix->nb = nb;
    if (b->dense)
        ix->p3 = b->d_pk;
    else
        ix->pk = b->d_pk;
    b->d_pk = nullptr;  // the index owns them now
    ix->packed_fmt = b->dense ? 4 : b->fmt;  // (dense rows: what an index looks like after memo_index_pack_dense(ix, 0))
    ix->order_pending = b->dense ? 0 : 1;    // (start order now; the query order once the queries have lost to it what the pass costs: memo_view.hip, keep_row_order)
    ix->finalized = 1;
    b->failed = MEMO_EINVAL;  // a builder finishes once
    if (b->dense) {  // rows that can never write at k <= 64 leave the dense rows when they are many (memo_common.h: boff3)
        ix->rows3 = ix->rows;
        ix->padded3 = ix->padded;
        if (int rc3 = dense_compact(ix)) {
            memo_index_destroy(ix);
            return rc3;
        }
    }
    if (int rc2 = memo_len_census(ix)) {
        memo_index_destroy(ix);
        return rc2;
    }
    *out = ix;
    return MEMO_OK;
}

// ---- a packed index to host memory and back (the CLI's sidecar cache, memo_amd/cache.py) ----------------
int memo_index_export_packed(memo_index_t *ix, uint32_t *pk, uint16_t *pa, int64_t *boff, int64_t *long_rows) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (!ix->finalized || !ix->packed_fmt || !ix->pk) return fail(MEMO_ENOTREADY, "the index has no packed rows (memo_index_pack)");
    if ((ix->rows && !pk) || (ix->packed_fmt == 6 && ix->rows && !pa) || !boff || (ix->n_long && !long_rows))
        return fail(MEMO_EINVAL, "output pointer is NULL");
    int rc;
    if ((rc = download_pipelined(ix->device, pk, ix->pk, ix->rows * 4, nullptr))) return rc;
    if (ix->packed_fmt == 6 && (rc = download_pipelined(ix->device, pa, ix->pa, ix->rows * 2, nullptr))) return rc;
    if ((rc = download_pipelined(ix->device, boff, ix->boff, ix->nb * 8, nullptr))) return rc;
    if (ix->n_long) {
        DeviceGuard guard(ix->device);
        HIP_TRY(hipMemcpy(long_rows, ix->ls, ix->n_long * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(long_rows + ix->n_long, ix->le, ix->n_long * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(long_rows + 2 * ix->n_long, ix->lo, ix->n_long * 8, hipMemcpyDeviceToHost));
    }
    return MEMO_OK;
}

// export of the dense rows of an index that holds them: ceil(rows / 5) groups of 16 bytes, the bucket table
// (info.buckets x int64) and the rows with end < start
int memo_index_export_dense(memo_index_t *ix, void *groups, int64_t *boff, int64_t *long_rows) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (!ix->finalized || !ix->p3) return fail(MEMO_ENOTREADY, "the index has no dense rows (memo_index_pack_dense)");
    if ((ix->rows && !groups) || !boff || (ix->n_long && !long_rows)) return fail(MEMO_EINVAL, "output pointer is NULL");
    int rc;
    const uint64_t drows = ix->boff3 ? ix->rows3 : ix->rows;  // (info.dense_row_count)
    if ((rc = download_pipelined(ix->device, groups, ix->p3, (size_t)((drows + 4) / 5) * 16, nullptr))) return rc;
    if ((rc = download_pipelined(ix->device, boff, ix->boff3 ? ix->boff3 : ix->boff, ix->nb * 8, nullptr))) return rc;
    if (ix->n_long) {
        DeviceGuard guard(ix->device);
        HIP_TRY(hipMemcpy(long_rows, ix->ls, ix->n_long * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(long_rows + ix->n_long, ix->le, ix->n_long * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(long_rows + 2 * ix->n_long, ix->lo, ix->n_long * 8, hipMemcpyDeviceToHost));
    }
    return MEMO_OK;
}

}  // extern "C"

// pk (+ pa) = 4- / 6-byte rows, or dense = 16-byte groups of five rows (then row_base is a multiple of 5 and the first
// table entry may lie up to 4 rows behind it: the slice starts with the group that holds the bucket's first row)
static int import_rows(uint64_t rows, int32_t device, int32_t bucket_shift, int64_t bucket_base, const uint32_t *pk,
                       const uint16_t *pa, const void *dense, const int64_t *boff, uint64_t buckets, int64_t row_base,
                       int64_t min_start, int64_t max_start, uint64_t max_annot, const int64_t *long_rows, uint64_t n_long,
                       memo_index_t **out) {
    if (!out) return fail(MEMO_EINVAL, "out is NULL");
    *out = nullptr;
    if (rows > ((uint64_t)1 << 40) || bucket_shift < 1 || bucket_shift > 8 || bucket_base < 0 || buckets < 2 ||
        (rows && !pk && !dense) || !boff || (n_long && !long_rows) || n_long > kMaxLongRows || max_annot > 65535)
        return fail(MEMO_EINVAL, "bad packed-index arguments");
    if (dense && (max_annot > 511 || row_base % 5 || row_base < 0))  // (256 .. 511: the ninth bit in the group's spare byte, memo_index.hip: pack3_rows_kernel)
        return fail(MEMO_EINVAL, "dense rows need every annot <= 511 and a row base that is a multiple of 5");
    // boff: buckets - 1 entries of the (absolute) table the rows were cut from; the index's table is those minus
    // row_base, plus one entry pinned to `rows`
    const int64_t lead = boff[0] - row_base;  // rows of the slice in front of its first bucket (dense rows: up to 4)
    if (lead < 0 || lead > (dense ? 4 : 0) || boff[buckets - 2] - row_base > (int64_t)rows || boff[buckets - 2] < row_base)
        return fail(MEMO_EINVAL, "bucket table does not match the rows (first %lld, last %lld, row base %lld, rows %llu)",
                    (long long)boff[0], (long long)boff[buckets - 2], (long long)row_base, (unsigned long long)rows);
    const int ndev = memo_device_count();
    if (device < 0 || device >= ndev) return fail(MEMO_EHIP, "HIP device %d not available (%d visible)", device, ndev);
    DeviceGuard guard(device);
    if (!guard.ok) return fail(MEMO_EHIP, "cannot select HIP device %d", device);
    memo_index *ix = new (std::nothrow) memo_index();
    if (!ix) return fail(MEMO_EHIP, "out of host memory");
    ix->device = device;
    ix->rows = rows;
    ix->padded = ((rows + 15) & ~(uint64_t)15) + kPadRows;
    ix->packed_rows = dense ? 0 : ix->padded;
    ix->has_wide = 0;
    ix->was_sorted = 1;
    ix->bshift = bucket_shift;
    ix->bbase = bucket_base;
    ix->nb = buckets;
    ix->min_s = min_start;
    ix->max_s = max_start;
    ix->max_annot = max_annot;
    const uint64_t groups = dense_groups_for(ix->padded), used = (rows + 4) / 5;
    if (dense) {
        ix->rows3 = rows;
        ix->padded3 = ix->padded;
    }
    int rc = MEMO_OK;
    do {
        hipError_t err = dense ? hipMalloc(&ix->p3, groups * 16) : hipMalloc(&ix->pk, ix->padded * 4);
        if (err == hipSuccess && pa) err = hipMalloc(&ix->pa, ix->padded * 2);
        if (err == hipSuccess) err = hipMalloc(&ix->boff, buckets * 8);
        if (err == hipSuccess) err = hipMalloc(&ix->d_status, 64);
        if (err == hipSuccess) err = hipMalloc(&ix->d_scratch, 64);
        if (err == hipSuccess) err = hipMemset(ix->d_status, 0, 64);
        if (err == hipSuccess)
            err = dense ? hipMemset(ix->p3 + 4 * used, 0, (groups - used) * 16) : hipMemset(ix->pk + rows, 0, (ix->padded - rows) * 4);
        if (err == hipSuccess && pa) err = hipMemset(ix->pa + rows, 0, (ix->padded - rows) * 2);
        if (err == hipSuccess && n_long) {
            err = hipMalloc(&ix->ls, n_long * 8);
            if (err == hipSuccess) err = hipMalloc(&ix->le, n_long * 8);
            if (err == hipSuccess) err = hipMalloc(&ix->lo, n_long * 8);
            if (err == hipSuccess) err = hipMemcpy(ix->ls, long_rows, n_long * 8, hipMemcpyHostToDevice);
            if (err == hipSuccess) err = hipMemcpy(ix->le, long_rows + n_long, n_long * 8, hipMemcpyHostToDevice);
            if (err == hipSuccess) err = hipMemcpy(ix->lo, long_rows + 2 * n_long, n_long * 8, hipMemcpyHostToDevice);
            ix->n_long = n_long;
        }
        if (err != hipSuccess) {
            rc = fail(MEMO_EHIP, "importing a packed index: %s", hipGetErrorString(err));
            break;
        }
        if (dense) {
            if ((rc = upload_pipelined(device, ix->p3, dense, used * 16))) break;
        } else {
            if ((rc = upload_pipelined(device, ix->pk, pk, rows * 4))) break;
            if (pa && (rc = upload_pipelined(device, ix->pa, pa, rows * 2))) break;
        }
        if ((rc = upload_pipelined(device, ix->boff, boff, (buckets - 1) * 8))) break;
        hipLaunchKernelGGL(rebase_table_kernel, dim3((unsigned)((buckets + 255) / 256)), dim3(256), 0, nullptr, ix->boff,
                           buckets, row_base, (int64_t)rows);
        err = hipGetLastError();
        if (err == hipSuccess) err = hipStreamSynchronize(nullptr);
        if (err != hipSuccess) rc = fail(MEMO_EHIP, "importing a packed index: %s", hipGetErrorString(err));
    } while (0);
    if (rc) {
        memo_index_destroy(ix);
        return rc;
    }
    ix->packed_fmt = pa ? 6 : (max_annot > 255 ? 12 : 4);  // the rule both packers follow: by the largest annot
    ix->order_pending = (dense || pa) ? 0 : 1;  // (whatever order the file holds; the query order when it has become worth its pass: idempotent)
    ix->finalized = 1;
    if ((rc = memo_len_census(ix))) {
        memo_index_destroy(ix);
        return rc;
    }
    *out = ix;
    return MEMO_OK;
}

extern "C" {

int memo_index_import_packed(uint64_t rows, int32_t device, int32_t bucket_shift, int64_t bucket_base,
                             const uint32_t *pk, const uint16_t *pa, const int64_t *boff, uint64_t buckets,
                             int64_t row_base, int64_t min_start, int64_t max_start, uint64_t max_annot,
                             const int64_t *long_rows, uint64_t n_long, memo_index_t **out) {
    if (rows && !pk) return fail(MEMO_EINVAL, "bad packed-index arguments");
    return import_rows(rows, device, bucket_shift, bucket_base, pk, pa, nullptr, boff, buckets, row_base, min_start,
                       max_start, max_annot, long_rows, n_long, out);
}

int memo_index_import_dense(uint64_t rows, int32_t device, int32_t bucket_shift, int64_t bucket_base,
                            const void *groups, const int64_t *boff, uint64_t buckets, int64_t row_base,
                            int64_t min_start, int64_t max_start, uint64_t max_annot, const int64_t *long_rows,
                            uint64_t n_long, memo_index_t **out) {
    static const uint32_t none[4] = {0, 0, 0, 0};
    if (rows && !groups) return fail(MEMO_EINVAL, "bad packed-index arguments");
    return import_rows(rows, device, bucket_shift, bucket_base, nullptr, nullptr, groups ? groups : none, boff, buckets,
                       row_base, min_start, max_start, max_annot, long_rows, n_long, out);
}

// One rule for "the dense rows alone can answer this query" (memo_sweep_cons.hip, query_conservation: the unclipped
// sweep on PackedRows3), for callers that choose the row format before they build or import an index: conservation,
// 2 <= k <= 64, at most 511 genomes (above 255: uint16 results, the nine-bit form of the table-driven kernel), every annot inside
// the result matrix, at least one row per position.
int memo_dense_rows_can_answer(uint64_t rows, int64_t min_start, int64_t max_start, uint64_t max_annot, int32_t k,
                               int32_t num_docs, int32_t membership) {
    if (membership || k < 2 || k - 1 > 63 || num_docs < 1 || num_docs > 511 || max_annot > (uint64_t)num_docs || !rows)
        return 0;
    const double span = (double)max_start - (double)min_start + 1.0;  // (as query_conservation judges "dense enough")
    return (double)rows >= span ? 1 : 0;
}

}  // extern "C"
