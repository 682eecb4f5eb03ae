// memo_hostcore.h -- the host side of the fast way in (worker pool, pinned staging ring, row packers, the
// builder's push loop), written against a small device seam (memo::hp) instead of the HIP runtime: the product
// links memo_hostpack.hip, which implements the seam with HIP; tests/host_stub.cpp implements it with malloc and
// memcpy so that the only multi-threaded host code of the library runs under -fsanitize=thread and
// -fsanitize=address,undefined on a machine without a GPU (tests/test_host_sanitizers.py).  Not part of the ABI.
#ifndef MEMO_HOSTCORE_H
#define MEMO_HOSTCORE_H

#include <cstddef>
#include <cstdint>
#include <vector>

#include "memo_amd.h"

namespace memo {

// error plumbing (memo_index.hip; the stub brings its own)
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

namespace hp {  // ---- the device seam: every function returns MEMO_OK or fail(MEMO_EHIP, ...) -----------------
int set_device(int device, int *previous);             // previous < 0: nothing to restore
int pinned_alloc(void **p, size_t bytes);
void pinned_free(void *p);
int stream_create(void **stream);                      // non-blocking stream
void stream_destroy(void *stream);
int stream_sync(void *stream);
int event_create(void **event);
void event_destroy(void *event);
int event_record(void *event, void *stream);
int event_sync(void *event);
int event_query(void *event, int *done);               // *done = 1 when everything queued before the record has finished
int copy_h2d_async(void *dev, const void *host, size_t bytes, void *stream);
int copy_d2h_async(void *host, const void *dev, size_t bytes, void *stream);
int copy_h2d(void *dev, const void *host, size_t bytes);       // blocking; pageable source allowed
// rows already on the device in format 4 -> format 12 (the first annot > 255 arrived late), queued on `stream`
int widen_annots(uint32_t *d_pk, uint64_t rows, void *stream);
}  // namespace hp

constexpr int kHostBucketShiftDefault = 5;
constexpr int64_t kHostCoordLimit = (int64_t)1 << 61;
constexpr uint64_t kMaxLongRows = (uint64_t)1 << 22;

// ---- worker threads: one process-wide pool, created on FIRST USE by a call that needs it -----------------
int host_threads_default();  // min(CPUs allowed, cgroup CFS quota, 32), or MEMO_HOST_THREADS (memo_cpus.h)
class HostPool {
public:
    static HostPool &get();
    int threads() const;
    // f(ctx, task) for task in [0, n); the caller works too.  One job at a time (jobs from different threads queue).
    void run(int n, void (*f)(void *, int), void *ctx);
    template <typename F>
    void run(int n, F &&f) {
        run(n, [](void *c, int t) { (*static_cast<F *>(c))(t); }, &f);
    }
    // the same in three steps, for a caller with work of its own while the job runs (the builder's push loop issues the
    // copies): begin wakes the pool on the n tasks, help makes the caller take tasks until none is left, end waits
    void begin(int n, void (*f)(void *, int), void *ctx);
    void help();
    void end();

private:
    HostPool();
    struct Impl;
    Impl *impl_;
};

// ---- pinned staging ring: slots allocated on first use, a copy stream, one event per slot ----------------
struct PinnedRing {
    static constexpr int kSlots = 4;
    static constexpr size_t kSlotBytes = (size_t)24 << 20;
    int device = -1;
    char *slot[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    void *done[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool in_flight[kSlots] = {false, false, false, false};
    void *stream = nullptr;
    int next = 0;

    int buffer(int s, char **out);  // the slot's pinned buffer, allocated now if this is its first use
    int wait(int s);                // the slot's last copy has left (or arrived in) the buffer
    int poll(int s, bool *idle);    // the same without blocking: *idle = the slot can be written now
    int mark(int s);                // an asynchronous copy of the slot was just queued on `stream`
    int drain();
};
int acquire_ring(int device, PinnedRing **out);
void release_ring(PinnedRing *r);
double pinned_alloc_ms_total();  // time this process has spent allocating pinned slots (MEMO_TIMING)
bool ring_cached(int device);  // an idle ring with at least one allocated slot exists for this device

int download_pipelined_core(int device, void *host, const void *dev, size_t bytes);
int upload_pipelined_core(int device, void *dev, const void *host, size_t bytes);

// ---- the builder ------------------------------------------------------------------------------------------
struct BlockResult {  // what one worker task found in its rows
    uint64_t max_annot = 0;
    int bad = 0;  // 1 unsorted, 2 negative start, 4 annot outside [0, 4095], 8 wild coordinate, 16 annot > 511 (dense rows)
    int wide_annot = 0;  // some annot > 255 (the one-word rows go to format 12)
    int over511 = 0;     // some annot > 511 (no dense rows: nine annot bits per row, memo_index.hip: pack3_rows_kernel)
    std::vector<int64_t> long_rows;  // (start, end, annot) triples with end < start
};

}  // namespace memo

struct memo_builder {
    int device = 0;
    int bshift = memo::kHostBucketShiftDefault;
    uint64_t cap = 0, padded = 0, rows = 0;
    int dense = 0;              // MEMO_ROWS_DENSE: five 24-bit rows per 16 bytes (PackedRows3, memo_sweep.h)
    uint32_t *d_pk = nullptr;   // 4-byte words (format 4 / 12), or the dense groups when `dense`
    uint64_t d_groups = 0;      // dense: groups the allocation holds
    int fmt = 4;                // 4 until the first annot > 255 arrives, then 12 (PackedRows, memo_sweep.h)
    bool any = false;
    int64_t first_start = 0, last_start = 0;
    int64_t last_bucket = -1;   // bucket of the last row seen; boff[0 .. last_bucket] are final
    std::vector<int64_t> boff;
    std::vector<int64_t> long_rows;
    uint64_t max_annot = 0;
    uint32_t len_hist[256] = {0};  // dense: overlap lengths of a sample of the rows (memo_index.len_hist)
    uint64_t len_hist_rows = 0;
    // dense: the rows of the last, incomplete group (they wait for the next push or for finish)
    uint32_t carry_b[5] = {0, 0, 0, 0, 0}, carry_a[5] = {0, 0, 0, 0, 0};
    int carry_n = 0;
    uint64_t groups_sent = 0;   // dense: whole groups already on their way to the device
    memo::PinnedRing *ring = nullptr;
    int failed = 0;
    int why = 0;                // MEMO_EUNPACKABLE: BlockResult::bad bits of the rows that could not be packed
};

namespace memo {
// the threaded part of memo_builder_push / memo_builder_finish (arguments already validated)
// stride 1: three columns; 3: rows -- start / end / annot of a row side by side (start = the [M, 3] array, end = start + 1,
// annot = start + 2): filter_pq's own array (memo_query.py:28-36)
int builder_push_core(memo_builder *b, const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows, int stride = 1);
int builder_flush_core(memo_builder *b);  // dense: the incomplete last group, padded with rows that never write
int builder_fail(memo_builder *b, int code, const char *what);
}  // namespace memo

#endif  // MEMO_HOSTCORE_H
