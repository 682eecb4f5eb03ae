// memo_hostpack.hip -- the fast way in for HOST rows: the drop-in seam hands over three int64 columns
// (what filter_pq returns, /root/reference/src/memo_query.py:28-36, re-typed at :45); this file narrows
// them to the packed query format ON THE HOST, with a pool of worker threads, into a ring of pinned
// buffers, and sends each chunk to the GPU with hipMemcpyAsync while the next one is being packed.
// PCIe then carries 4 (6) bytes per row instead of 24, from pinned memory instead of pageable, and the
// sweep behind the seam reads PackedRows (memo_sweep.h) like a resident query does.
//
// The same pass does what memo_index_finalize does on the device for int64 uploads: start-sortedness,
// coordinate range, the rows with end < start (set aside for long_rows_*_kernel), the largest annot, and
// the start-bucket table -- built from the sorted starts as they stream by, no search.
//
// Rows that cannot be packed into one word (unsorted, negative start, annot outside [0, 4095], coordinates
// beyond +-2^61) make the builder return MEMO_EUNPACKABLE; the caller then takes the int64 path
// (memo_index_upload + memo_index_finalize + memo_index_pack), which sorts on the device, knows the 6-byte
// format for larger annots and handles every legal input.
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "memo_common.h"

using namespace memo;

namespace memo {

// ------------------------------------------------------------------------------------------
// worker threads: one process-wide pool, created on first use, never joined (the library may be
// unloaded at exit with the threads parked on their condition variable)
// ------------------------------------------------------------------------------------------
class HostPool {
public:
    static HostPool &get() {
        static HostPool *p = new HostPool();  // leaked on purpose
        return *p;
    }
    int threads() const { return (int)workers_.size() + 1; }

    // f(task) for task in [0, n); the caller works too.  One job at a time (jobs from different
    // threads queue on run_mutex_).
    void run(int n, const std::function<void(int)> &f) {
        if (n <= 0) return;
        if (n == 1 || workers_.empty()) {
            for (int i = 0; i < n; ++i) f(i);
            return;
        }
        std::lock_guard<std::mutex> serial(run_mutex_);
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &f;
            n_ = n;
            next_.store(0, std::memory_order_relaxed);
            busy_ = (int)workers_.size();
            ++generation_;
        }
        cv_work_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return busy_ == 0; });
        job_ = nullptr;
    }

private:
    HostPool() {
        unsigned hw = std::thread::hardware_concurrency();
        if (hw == 0) hw = 1;
        unsigned want = hw < 32 ? hw : 32;  // profiles/r02_oneshot_host_threads.txt: 32 threads pack fastest; 64 and up lose a third
        if (const char *v = getenv("MEMO_HOST_THREADS")) {
            const int n = atoi(v);
            if (n > 0) want = (unsigned)n;
        }
        for (unsigned i = 1; i < want; ++i) {
            workers_.emplace_back([this] { loop(); });
            workers_.back().detach();
        }
    }
    void work() {
        for (;;) {
            const int i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_) break;
            (*job_)(i);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_work_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
            }
            work();
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--busy_ == 0) cv_done_.notify_one();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, run_mutex_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(int)> *job_ = nullptr;
    int n_ = 0, busy_ = 0;
    std::atomic<int> next_{0};
    uint64_t generation_ = 0;
};

// ------------------------------------------------------------------------------------------
// pinned staging ring: kSlots buffers of kSlotBytes, a copy stream, one event per slot.  Rings are
// cached per device (hipHostMalloc of 3 x 24 MiB costs milliseconds) and handed out to one user at a
// time; a second concurrent user on the same device gets a ring of its own.
// ------------------------------------------------------------------------------------------
struct PinnedRing {
    static constexpr int kSlots = 3;
    static constexpr size_t kSlotBytes = (size_t)24 << 20;  // 4 Mi rows x (4 + 2) B
    int device = -1;
    char *slot[kSlots] = {nullptr, nullptr, nullptr};
    hipEvent_t done[kSlots] = {nullptr, nullptr, nullptr};
    bool in_flight[kSlots] = {false, false, false};
    hipStream_t stream = nullptr;
    int next = 0;

    int wait(int s) {  // the slot's last copy has left (or arrived in) the buffer
        if (in_flight[s]) {
            HIP_TRY(hipEventSynchronize(done[s]));
            in_flight[s] = false;
        }
        return MEMO_OK;
    }
    int mark(int s) {
        HIP_TRY(hipEventRecord(done[s], stream));
        in_flight[s] = true;
        return MEMO_OK;
    }
    int drain() {
        HIP_TRY(hipStreamSynchronize(stream));
        for (int s = 0; s < kSlots; ++s) in_flight[s] = false;
        return MEMO_OK;
    }
};

namespace {
std::mutex g_ring_mutex;
std::vector<PinnedRing *> g_idle_rings;
}  // namespace

int acquire_ring(int device, PinnedRing **out) {
    *out = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_ring_mutex);
        for (size_t i = 0; i < g_idle_rings.size(); ++i)
            if (g_idle_rings[i]->device == device) {
                *out = g_idle_rings[i];
                g_idle_rings.erase(g_idle_rings.begin() + (long)i);
                return MEMO_OK;
            }
    }
    PinnedRing *r = new (std::nothrow) PinnedRing();
    if (!r) return fail(MEMO_EHIP, "out of host memory");
    r->device = device;
    hipError_t err = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
    for (int s = 0; s < PinnedRing::kSlots && err == hipSuccess; ++s) {
        err = hipHostMalloc(reinterpret_cast<void **>(&r->slot[s]), PinnedRing::kSlotBytes, hipHostMallocDefault);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&r->done[s], hipEventDisableTiming);
    }
    if (err != hipSuccess) {
        for (int s = 0; s < PinnedRing::kSlots; ++s) {
            if (r->slot[s]) (void)hipHostFree(r->slot[s]);
            if (r->done[s]) (void)hipEventDestroy(r->done[s]);
        }
        if (r->stream) (void)hipStreamDestroy(r->stream);
        delete r;
        return fail(MEMO_EHIP, "pinned staging ring: %s", hipGetErrorString(err));
    }
    *out = r;
    return MEMO_OK;
}

void release_ring(PinnedRing *r) {
    if (!r) return;
    (void)hipStreamSynchronize(r->stream);
    for (int s = 0; s < PinnedRing::kSlots; ++s) r->in_flight[s] = false;
    std::lock_guard<std::mutex> lk(g_ring_mutex);
    g_idle_rings.push_back(r);  // kept for the next builder / download on this device
}

// device -> pageable host memory through the ring: the DMA of piece i+1 runs while the worker threads
// copy piece i out of its pinned slot.  `stream_done`: work on this stream must finish first.
int download_pipelined(int device, void *host, const void *dev, size_t bytes, hipStream_t producer) {
    DeviceGuard guard(device);
    if (!bytes) {
        HIP_TRY(hipStreamSynchronize(producer));
        return MEMO_OK;
    }
    if (bytes < ((size_t)4 << 20)) {
        HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, producer));
        HIP_TRY(hipStreamSynchronize(producer));
        return MEMO_OK;
    }
    HIP_TRY(hipStreamSynchronize(producer));
    PinnedRing *ring = nullptr;
    int rc = acquire_ring(device, &ring);
    if (rc) return rc;
    const size_t piece = PinnedRing::kSlotBytes;
    const size_t n = (bytes + piece - 1) / piece;
    HostPool &pool = HostPool::get();
    auto size_of = [&](size_t i) { return i + 1 < n ? piece : bytes - i * piece; };
    auto issue = [&](size_t i) -> int {
        const int s = (int)(i % PinnedRing::kSlots);
        HIP_TRY(hipMemcpyAsync(ring->slot[s], static_cast<const char *>(dev) + i * piece, size_of(i),
                               hipMemcpyDeviceToHost, ring->stream));
        return ring->mark(s);
    };
    for (size_t i = 0; i < n && i < (size_t)PinnedRing::kSlots - 1 && rc == MEMO_OK; ++i) rc = issue(i);
    for (size_t i = 0; i < n && rc == MEMO_OK; ++i) {
        const int s = (int)(i % PinnedRing::kSlots);
        if ((rc = ring->wait(s))) break;
        if (i + PinnedRing::kSlots - 1 < n && (rc = issue(i + PinnedRing::kSlots - 1))) break;
        const size_t sz = size_of(i);
        char *dst = static_cast<char *>(host) + i * piece;
        const char *src = ring->slot[s];
        const int tasks = (int)((sz + ((size_t)1 << 20) - 1) >> 20);
        pool.run(tasks, [&](int t) {
            const size_t b = (size_t)t << 20, e = b + ((size_t)1 << 20) < sz ? b + ((size_t)1 << 20) : sz;
            memcpy(dst + b, src + b, e - b);
        });
    }
    release_ring(ring);
    return rc;
}

// pageable host memory (a memory-mapped cache file, a NumPy array) -> device through the ring: the worker
// threads copy piece i + 1 into a pinned slot while piece i crosses PCIe
int upload_pipelined(int device, void *dev, const void *host, size_t bytes) {
    DeviceGuard guard(device);
    if (!bytes) return MEMO_OK;
    if (bytes < ((size_t)1 << 20)) {
        HIP_TRY(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
        return MEMO_OK;
    }
    PinnedRing *ring = nullptr;
    int rc = acquire_ring(device, &ring);
    if (rc) return rc;
    const size_t piece = PinnedRing::kSlotBytes / 2;  // 12 MiB pieces: the first one leaves early
    const size_t n = (bytes + piece - 1) / piece;
    HostPool &pool = HostPool::get();
    for (size_t i = 0; i < n && rc == MEMO_OK; ++i) {
        const int s = (int)(i % PinnedRing::kSlots);
        if ((rc = ring->wait(s))) break;
        const size_t sz = i + 1 < n ? piece : bytes - i * piece;
        const char *src = static_cast<const char *>(host) + i * piece;
        char *dst = ring->slot[s];
        const int tasks = (int)((sz + ((size_t)1 << 20) - 1) >> 20);
        pool.run(tasks, [&](int t) {
            const size_t b = (size_t)t << 20, e = b + ((size_t)1 << 20) < sz ? b + ((size_t)1 << 20) : sz;
            memcpy(dst + b, src + b, e - b);
        });
        hipError_t err = hipMemcpyAsync(static_cast<char *>(dev) + i * piece, dst, sz, hipMemcpyHostToDevice, ring->stream);
        if (err != hipSuccess) {
            rc = fail(MEMO_EHIP, "hipMemcpyAsync H2D: %s", hipGetErrorString(err));
            break;
        }
        rc = ring->mark(s);
    }
    release_ring(ring);  // synchronises the copy stream
    return rc;
}

}  // namespace memo

// ------------------------------------------------------------------------------------------
// the builder
// ------------------------------------------------------------------------------------------
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

constexpr uint64_t kChunkRows = PinnedRing::kSlotBytes / 4;  // rows per pinned slot
constexpr uint64_t kBlockRows = 1 << 16;                     // rows per worker task
constexpr uint64_t kMaxLongRows = (uint64_t)1 << 22;

struct BlockResult {  // what one worker task found in its rows
    uint64_t max_annot = 0;
    int bad = 0;  // 1 unsorted, 2 negative start, 4 annot outside [0, 4095], 8 wild coordinate
    int wide_annot = 0;
    std::vector<int64_t> long_rows;  // (start, end, annot) triples with end < start
};

}  // namespace

struct memo_builder {
    int device = 0;
    int bshift = kDefaultBucketShift;
    uint64_t cap = 0, padded = 0, rows = 0;
    uint32_t *d_pk = nullptr;
    int fmt = 4;  // 4 until the first annot > 255 arrives, then 12 (PackedRows, memo_sweep.h)
    bool any = false;
    int64_t first_start = 0, last_start = 0;
    int64_t last_bucket = -1;       // bucket of the last row seen; boff[0 .. last_bucket] are final
    std::vector<int64_t> boff;
    std::vector<int64_t> long_rows;
    uint64_t max_annot = 0;
    PinnedRing *ring = nullptr;
    int failed = 0;
};

namespace {

// rows [i0, i1) of this push -> words (format 4 or 12) at the same offsets of the slot; the buckets these
// rows open get their first row number
void pack_block(const memo_builder *b, const int64_t *start, const int64_t *end, const int64_t *annot,
                uint64_t i0, uint64_t i1, uint64_t global0, int64_t prev_start, int64_t prev_bucket,
                int64_t *boff, int64_t boff_size, uint32_t *pk, int fmt, BlockResult &res) {
    const int shift = b->bshift;
    uint64_t top = 0;
    int bad = 0, wide = 0;
    int64_t ps = prev_start, pb = prev_bucket;
    for (uint64_t i = i0; i < i1; ++i) {
        const int64_t s = start[i], e = end[i], a = annot[i];
        bad |= (s < ps) ? 1 : 0;
        bad |= (s < 0) ? 2 : 0;
        bad |= ((uint64_t)a > 4095u) ? 4 : 0;
        bad |= (s >= kCoordLimit || e <= -kCoordLimit || e >= kCoordLimit) ? 8 : 0;
        ps = s;
        const int64_t len = e - s;
        if (len < 0) {
            res.long_rows.push_back(s);
            res.long_rows.push_back(e);
            res.long_rows.push_back(a);
        }
        // end < start (handled by long_rows_*_kernel) packs as "never writes", like len >= 255
        const uint32_t l8 = (uint64_t)len > 255u ? 255u : (uint32_t)len;
        const uint32_t a12 = (uint32_t)a & 0xFFFu;
        top = a12 > top ? a12 : top;
        wide |= a12 > 255u;
        pk[i] = fmt == 12 ? l8 | (((uint32_t)s & 0xFFFu) << 8) | (a12 << 20) : ((uint32_t)s & 0xFFFFu) | (l8 << 16) | (a12 << 24);
        const int64_t bk = s >> shift;
        if (bk != pb) {  // first row of its bucket(s): boff[b] = lower_bound(start, b << shift)
            if (bk > pb && !bad && bk < boff_size)
                for (int64_t q = pb + 1; q <= bk; ++q) boff[q] = (int64_t)(global0 + i);
            pb = bk;
        }
    }
    res.max_annot = top;
    res.bad = bad;
    res.wide_annot = wide;
}

int builder_fail(memo_builder *b, int code, const char *what) {
    b->failed = code;
    return fail(code, "%s", what);
}

}  // namespace

extern "C" {

int memo_builder_create(uint64_t max_rows, int32_t device, int32_t bucket_shift, memo_builder_t **out) {
    if (!out) return fail(MEMO_EINVAL, "out is NULL");
    *out = nullptr;
    if (max_rows > ((uint64_t)1 << 40)) return fail(MEMO_EINVAL, "too many rows");
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
    b->padded = ((max_rows + 15) & ~(uint64_t)15) + kPadRows;
    const size_t bytes = (size_t)b->padded * sizeof(uint32_t);
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
    HostPool &pool = HostPool::get();
    PinnedRing *ring = b->ring;
    // the bucket table has to reach the last row's bucket before the workers fill it
    const int64_t s_last = start[rows - 1];
    if (s_last < 0 || s_last >= kCoordLimit || (b->any && start[0] < b->last_start))
        return builder_fail(b, MEMO_EUNPACKABLE, "rows are unsorted or have a start outside [0, 2^61): not packable");
    const int64_t need = (s_last >> b->bshift) + 3;
    if ((int64_t)b->boff.size() < need) {
        if ((uint64_t)need > ((uint64_t)1 << 34)) return builder_fail(b, MEMO_EUNPACKABLE, "bucket table too large");
        b->boff.resize((size_t)need + (size_t)need / 4);
    }
    for (uint64_t c0 = 0; c0 < rows; c0 += kChunkRows) {
        const uint64_t cn = rows - c0 < kChunkRows ? rows - c0 : kChunkRows;
        const int s = ring->next;
        ring->next = (s + 1) % PinnedRing::kSlots;
        int rc = ring->wait(s);
        if (rc) return rc;
        uint32_t *pk = reinterpret_cast<uint32_t *>(ring->slot[s]);
        const int tasks = (int)((cn + kBlockRows - 1) / kBlockRows);
        std::vector<BlockResult> res((size_t)tasks);
        for (int pass = 0; pass < 2; ++pass) {  // a second pass only when this chunk is the first with an annot > 255
            const int fmt = b->fmt;
            pool.run(tasks, [&](int t) {
                const uint64_t i0 = c0 + (uint64_t)t * kBlockRows;
                const uint64_t i1 = i0 + kBlockRows < c0 + cn ? i0 + kBlockRows : c0 + cn;
                const bool first = i0 == 0;
                const int64_t prev_start = first ? (b->any ? b->last_start : INT64_MIN) : start[i0 - 1];
                const int64_t prev_bucket = first ? b->last_bucket : (start[i0 - 1] >> b->bshift);
                res[(size_t)t].long_rows.clear();
                pack_block(b, start, end, annot, i0, i1, b->rows, prev_start, prev_bucket, b->boff.data(),
                           (int64_t)b->boff.size(), pk - c0, fmt, res[(size_t)t]);
            });
            int bad = 0, wide = 0;
            for (const BlockResult &r : res) {
                bad |= r.bad;
                wide |= r.wide_annot;
            }
            if (bad)
                return builder_fail(b, MEMO_EUNPACKABLE,
                                    bad & 1   ? "rows are not sorted by start: not packable on the host"
                                    : bad & 2 ? "rows with a negative start cannot be packed"
                                    : bad & 4 ? "rows with an annot outside [0, 4095] do not fit the one-word formats"
                                              : "rows have coordinates beyond +-2^61");
            if (wide && fmt == 4) {  // switch the index to 12-bit annots: rewrite what is on the device, redo this chunk
                HIP_TRY(hipStreamSynchronize(ring->stream));
                if (b->rows + c0) {
                    hipLaunchKernelGGL(widen_annot_kernel, dim3(2048), dim3(256), 0, ring->stream, b->d_pk, b->rows + c0);
                    HIP_TRY(hipGetLastError());
                }
                b->fmt = 12;
                continue;
            }
            break;
        }
        for (BlockResult &r : res) {
            if (r.max_annot > b->max_annot) b->max_annot = r.max_annot;
            if (!r.long_rows.empty()) {
                b->long_rows.insert(b->long_rows.end(), r.long_rows.begin(), r.long_rows.end());
                if (b->long_rows.size() / 3 > kMaxLongRows)
                    return builder_fail(b, MEMO_ELONGROW, "more than 2^22 rows have end < start: not a MEMO overlap index");
            }
        }
        HIP_TRY(hipMemcpyAsync(b->d_pk + b->rows + c0, pk, cn * 4, hipMemcpyHostToDevice, ring->stream));
        if ((rc = ring->mark(s))) return rc;
    }
    if (!b->any) b->first_start = start[0];
    b->any = true;
    b->last_start = s_last;
    b->last_bucket = s_last >> b->bshift;
    b->rows += rows;
    return MEMO_OK;
}

int memo_builder_finish(memo_builder_t *b, memo_index_t **out) {
    if (!b || !out) return fail(MEMO_EINVAL, "NULL argument");
    *out = nullptr;
    if (b->failed) return fail(b->failed, "the builder already failed");
    DeviceGuard guard(b->device);
    memo_index *ix = new (std::nothrow) memo_index();
    if (!ix) return fail(MEMO_EHIP, "out of host memory");
    ix->device = b->device;
    ix->rows = b->rows;
    ix->padded = b->padded;
    ix->packed_rows = b->padded;
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
    hipStream_t st = b->ring->stream;
    do {
        hipError_t err = hipMalloc(&ix->boff, nb * sizeof(int64_t));
        if (err == hipSuccess) err = hipMalloc(&ix->d_status, 64);
        if (err == hipSuccess) err = hipMalloc(&ix->d_scratch, 64);
        if (err == hipSuccess) err = hipMemsetAsync(ix->d_status, 0, 64, st);
        if (err == hipSuccess)
            err = hipMemcpyAsync(ix->boff, b->boff.data(), nb * sizeof(int64_t), hipMemcpyHostToDevice, st);
        // the rows behind the last one are read (never used) by whole-wave loads: keep them defined
        if (err == hipSuccess) err = hipMemsetAsync(b->d_pk + b->rows, 0, (b->padded - b->rows) * 4, st);

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
    ix->pk = b->d_pk;
    b->d_pk = nullptr;  // the index owns them now
    ix->packed_fmt = b->fmt;
    ix->finalized = 1;
    b->failed = MEMO_EINVAL;  // a builder finishes once
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

int memo_index_import_packed(uint64_t rows, int32_t device, int32_t bucket_shift, int64_t bucket_base,
                             const uint32_t *pk, const uint16_t *pa, const int64_t *boff, uint64_t buckets,
                             int64_t row_base, int64_t min_start, int64_t max_start, uint64_t max_annot,
                             const int64_t *long_rows, uint64_t n_long, memo_index_t **out) {
    if (!out) return fail(MEMO_EINVAL, "out is NULL");
    *out = nullptr;
    if (rows > ((uint64_t)1 << 40) || bucket_shift < 1 || bucket_shift > 8 || bucket_base < 0 || buckets < 2 ||
        (rows && !pk) || !boff || (n_long && !long_rows) || n_long > kMaxLongRows || max_annot > 65535)
        return fail(MEMO_EINVAL, "bad packed-index arguments");
    // boff: buckets - 1 entries of the (absolute) table the rows were cut from; the index's table is those minus
    // row_base, plus one entry pinned to `rows`
    if (boff[0] != row_base || boff[buckets - 2] - row_base > (int64_t)rows || boff[buckets - 2] < row_base)
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
    ix->packed_rows = ix->padded;
    ix->has_wide = 0;
    ix->was_sorted = 1;
    ix->bshift = bucket_shift;
    ix->bbase = bucket_base;
    ix->nb = buckets;
    ix->min_s = min_start;
    ix->max_s = max_start;
    ix->max_annot = max_annot;
    int rc = MEMO_OK;
    do {
        hipError_t err = hipMalloc(&ix->pk, ix->padded * 4);
        if (err == hipSuccess && pa) err = hipMalloc(&ix->pa, ix->padded * 2);
        if (err == hipSuccess) err = hipMalloc(&ix->boff, buckets * 8);
        if (err == hipSuccess) err = hipMalloc(&ix->d_status, 64);
        if (err == hipSuccess) err = hipMalloc(&ix->d_scratch, 64);
        if (err == hipSuccess) err = hipMemset(ix->d_status, 0, 64);
        if (err == hipSuccess) err = hipMemset(ix->pk + rows, 0, (ix->padded - rows) * 4);
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
        if ((rc = upload_pipelined(device, ix->pk, pk, rows * 4))) break;
        if (pa && (rc = upload_pipelined(device, ix->pa, pa, rows * 2))) break;
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
    ix->finalized = 1;
    if ((rc = memo_len_census(ix))) {
        memo_index_destroy(ix);
        return rc;
    }
    *out = ix;
    return MEMO_OK;
}

}  // extern "C"
