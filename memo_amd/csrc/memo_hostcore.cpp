// memo_hostcore.cpp -- the host side of the fast way in for HOST rows, free of the HIP runtime (memo_hostcore.h).
//
// The drop-in seam hands over three int64 columns (what filter_pq returns, /root/reference/src/memo_query.py:28-36,
// re-typed at :45).  A pool of worker threads narrows them ON THE HOST to the row format the sweep reads -- one
// 32-bit word per row (PackedRows) or five 24-bit rows per 16 bytes (PackedRows3, the format of the benchmarked
// kernel; memo_sweep.h) -- into a ring of pinned buffers, and each chunk crosses PCIe asynchronously while the next
// one is being packed: 4 or 3.2 bytes per row on the link instead of 24.  The same pass does what memo_index_finalize
// does on the device for int64 uploads: start-sortedness, coordinate range, the rows with end < start (set aside for
// long_rows_*_kernel), the largest annot, and the start-bucket table -- built from the sorted starts as they stream
// by, no search.
//
// Rows that cannot be packed (unsorted, negative start, annot outside [0, 4095] -- outside [0, 511] for the dense
// rows --, coordinates beyond +-2^61) make the builder return MEMO_EUNPACKABLE; the caller then takes the next way in
// (dense -> 4-byte words -> memo_index_upload + memo_index_finalize + memo_index_pack, which sorts on the device, knows
// the 6-byte format for larger annots and handles every legal input).
#include "memo_hostcore.h"

#include "memo_cpus.h"

#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace memo {

// ------------------------------------------------------------------------------------------
// worker threads: one process-wide pool, created on first use, never joined (the library may be
// unloaded at exit with the threads parked on their condition variable)
// ------------------------------------------------------------------------------------------
struct HostPool::Impl {
    std::vector<std::thread> workers;
    std::mutex m, run_mutex;
    std::condition_variable cv_work, cv_done;
    void (*job)(void *, int) = nullptr;
    void *ctx = nullptr;
    int n = 0, busy = 0;
    std::atomic<int> next{0};
    uint64_t generation = 0;

    void work(void (*f)(void *, int), void *c, int count) {
        for (;;) {
            const int i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= count) break;
            f(c, i);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            void (*f)(void *, int);
            void *c;
            int count;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return generation != seen; });
                seen = generation;
                f = job;
                c = ctx;
                count = n;
            }
            work(f, c, count);
            {
                std::lock_guard<std::mutex> lk(m);
                if (--busy == 0) cv_done.notify_one();
            }
        }
    }
};

HostPool &HostPool::get() {
    static HostPool *p = new HostPool();  // leaked on purpose
    return *p;
}

// How many threads: the CPUs this process may run on, cut to its cgroup's CFS quota (memo_cpus.h), at most 32 -- the
// packer is bound by DRAM from ~16-24 threads on (tools/host_probe.cpp on the pool's 2 x EPYC 9575F: three int64 columns
// stream at 300 GB/s with 8 threads, 470 with 32).  Round 2's finding "64 threads and up lose a third" was the quota: the
// boxes grant 16 CPUs' worth of time per 100 ms, threads beyond that bring the period's freeze forward
// (profiles/r06_oneshot.txt).
int host_threads_default() {
    int want = cpu_budget();
    if (want > 32) want = 32;
    if (const char *v = getenv("MEMO_HOST_THREADS")) {
        const int n = atoi(v);
        if (n > 0) want = n;
    }
    return want < 1 ? 1 : want;
}

HostPool::HostPool() : impl_(new Impl()) {
    const int want = host_threads_default();
    for (int i = 1; i < want; ++i) {
        impl_->workers.emplace_back([this] { impl_->loop(); });
        impl_->workers.back().detach();
    }
}

int HostPool::threads() const { return (int)impl_->workers.size() + 1; }

// begin / help / end: the job's tasks are taken by the pool's threads from begin on; the caller may do something else
// in between (the builder's push loop issues the copies), take tasks itself (help), and must call end.  One job at a
// time: begin blocks while another thread's job runs.
void HostPool::begin(int n, void (*f)(void *, int), void *ctx) {
    Impl &I = *impl_;
    I.run_mutex.lock();
    {
        std::lock_guard<std::mutex> lk(I.m);
        I.job = f;
        I.ctx = ctx;
        I.n = n;
        I.next.store(0, std::memory_order_relaxed);
        I.busy = (int)I.workers.size();
        ++I.generation;
    }
    I.cv_work.notify_all();
}

void HostPool::help() {
    Impl &I = *impl_;
    I.work(I.job, I.ctx, I.n);
}

void HostPool::end() {
    Impl &I = *impl_;
    {
        std::unique_lock<std::mutex> lk(I.m);
        I.cv_done.wait(lk, [&] { return I.busy == 0; });
        I.job = nullptr;
    }
    I.run_mutex.unlock();
}

void HostPool::run(int n, void (*f)(void *, int), void *ctx) {
    if (n <= 0) return;
    Impl &I = *impl_;
    if (n == 1 || I.workers.empty()) {
        for (int i = 0; i < n; ++i) f(ctx, i);
        return;
    }
    begin(n, f, ctx);
    help();
    end();
}

// ------------------------------------------------------------------------------------------
// pinned staging ring.  Rings are cached per device and handed out to one user at a time; a second concurrent
// user on the same device gets a ring of its own.  A slot's pinned buffer is allocated when the slot is first
// used (hipHostMalloc of 24 MiB costs milliseconds: a call that moves one small piece pays for one slot, a call
// that moves none -- a cache hit of `memo query` on a small window -- for nothing).
// ------------------------------------------------------------------------------------------
std::atomic<uint64_t> g_pinned_alloc_ns{0};  // time spent allocating pinned slots, process-wide (MEMO_TIMING reports it)
double pinned_alloc_ms_total() { return (double)g_pinned_alloc_ns.load(std::memory_order_relaxed) * 1e-6; }

int PinnedRing::buffer(int s, char **out) {
    if (!slot[s]) {
        void *p = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        int rc = hp::pinned_alloc(&p, kSlotBytes);
        g_pinned_alloc_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(),
                                    std::memory_order_relaxed);
        if (rc) return rc;
        if (!done[s] && (rc = hp::event_create(&done[s]))) {
            hp::pinned_free(p);
            return rc;
        }
        slot[s] = static_cast<char *>(p);
    }
    *out = slot[s];
    return MEMO_OK;
}

int PinnedRing::wait(int s) {
    if (in_flight[s]) {
        int rc = hp::event_sync(done[s]);
        if (rc) return rc;
        in_flight[s] = false;
    }
    return MEMO_OK;
}

int PinnedRing::poll(int s, bool *idle) {
    if (in_flight[s]) {
        int done_now = 0;
        int rc = hp::event_query(done[s], &done_now);
        if (rc) return rc;
        if (done_now) in_flight[s] = false;
    }
    *idle = !in_flight[s];
    return MEMO_OK;
}

int PinnedRing::mark(int s) {
    int rc = hp::event_record(done[s], stream);
    if (rc) return rc;
    in_flight[s] = true;
    return MEMO_OK;
}

int PinnedRing::drain() {
    int rc = hp::stream_sync(stream);
    for (int s = 0; s < kSlots; ++s) in_flight[s] = false;
    return rc;
}

namespace {
std::mutex g_ring_mutex;
std::vector<PinnedRing *> g_idle_rings;
}  // namespace

bool ring_cached(int device) {
    std::lock_guard<std::mutex> lk(g_ring_mutex);
    for (PinnedRing *r : g_idle_rings)
        if (r->device == device && r->slot[0]) return true;
    return false;
}

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
    int rc = hp::stream_create(&r->stream);
    if (rc) {
        delete r;
        return rc;
    }
    *out = r;
    return MEMO_OK;
}

void release_ring(PinnedRing *r) {
    if (!r) return;
    (void)r->drain();
    r->next = 0;
    std::lock_guard<std::mutex> lk(g_ring_mutex);
    g_idle_rings.push_back(r);  // kept for the next builder / transfer on this device
}

namespace {
struct DeviceScope {  // the caller keeps its own notion of the current device
    int prev = -1;
    int rc;
    explicit DeviceScope(int dev) { rc = hp::set_device(dev, &prev); }
    ~DeviceScope() {
        if (prev >= 0) (void)hp::set_device(prev, nullptr);
    }
};

void copy_tasks(HostPool &pool, char *dst, const char *src, size_t sz) {
    const int tasks = (int)((sz + ((size_t)1 << 20) - 1) >> 20);
    pool.run(tasks, [&](int t) {
        const size_t b = (size_t)t << 20, e = b + ((size_t)1 << 20) < sz ? b + ((size_t)1 << 20) : sz;
        memcpy(dst + b, src + b, e - b);
    });
}
}  // namespace

// device -> pageable host memory through the ring: the DMA of piece i+1 runs while the worker threads
// copy piece i out of its pinned slot.  (Whatever produced `dev` has finished: the caller synchronised.)
int download_pipelined_core(int device, void *host, const void *dev, size_t bytes) {
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    if (!bytes) return MEMO_OK;
    if (bytes < ((size_t)4 << 20)) {
        int rc = hp::copy_d2h_async(host, dev, bytes, nullptr);
        return rc ? rc : hp::stream_sync(nullptr);
    }
    PinnedRing *ring = nullptr;
    int rc = acquire_ring(device, &ring);
    if (rc) return rc;
    const size_t piece = PinnedRing::kSlotBytes;
    const size_t n = (bytes + piece - 1) / piece;
    HostPool &pool = HostPool::get();
    auto size_of = [&](size_t i) { return i + 1 < n ? piece : bytes - i * piece; };
    auto issue = [&](size_t i) -> int {
        const int s = (int)(i % PinnedRing::kSlots);
        char *buf = nullptr;
        int r = ring->buffer(s, &buf);
        if (r) return r;
        if ((r = hp::copy_d2h_async(buf, static_cast<const char *>(dev) + i * piece, size_of(i), ring->stream))) return r;
        return ring->mark(s);
    };
    for (size_t i = 0; i < n && i < (size_t)PinnedRing::kSlots - 1 && rc == MEMO_OK; ++i) rc = issue(i);
    for (size_t i = 0; i < n && rc == MEMO_OK; ++i) {
        const int s = (int)(i % PinnedRing::kSlots);
        if ((rc = ring->wait(s))) break;
        if (i + PinnedRing::kSlots - 1 < n && (rc = issue(i + PinnedRing::kSlots - 1))) break;
        copy_tasks(pool, static_cast<char *>(host) + i * piece, ring->slot[s], size_of(i));
    }
    release_ring(ring);
    return rc;
}

// pageable host memory (a memory-mapped cache file, a NumPy array) -> device through the ring: the worker
// threads copy piece i + 1 into a pinned slot while piece i crosses PCIe.  A process that has no ring yet and
// moves less than MEMO_COLD_RING_MB (default 48 MiB) takes the runtime's own pageable copy instead: setting the ring up
// (worker threads, a stream, two or three pinned buffers) costs more than it saves on a transfer that short.
int upload_pipelined_core(int device, void *dev, const void *host, size_t bytes) {
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    if (!bytes) return MEMO_OK;
    static const size_t cold_limit = [] {
        const char *v = getenv("MEMO_COLD_RING_MB");
        return (size_t)(v && atoi(v) >= 0 ? atoi(v) : 48) << 20;
    }();
    if (bytes < ((size_t)1 << 20) || (bytes < cold_limit && !ring_cached(device))) return hp::copy_h2d(dev, host, bytes);
    PinnedRing *ring = nullptr;
    int rc = acquire_ring(device, &ring);
    if (rc) return rc;
    const size_t piece = PinnedRing::kSlotBytes / 2;  // 12 MiB pieces: the first one leaves early
    const size_t n = (bytes + piece - 1) / piece;
    HostPool &pool = HostPool::get();
    for (size_t i = 0; i < n && rc == MEMO_OK; ++i) {
        const int s = (int)(i % PinnedRing::kSlots);
        char *dst = nullptr;
        if ((rc = ring->buffer(s, &dst))) break;
        if ((rc = ring->wait(s))) break;
        const size_t sz = i + 1 < n ? piece : bytes - i * piece;
        copy_tasks(pool, dst, static_cast<const char *>(host) + i * piece, sz);
        if ((rc = hp::copy_h2d_async(static_cast<char *>(dev) + i * piece, dst, sz, ring->stream))) break;
        rc = ring->mark(s);
    }
    release_ring(ring);  // synchronises the copy stream
    return rc;
}

// ------------------------------------------------------------------------------------------
// row packers
// ------------------------------------------------------------------------------------------
namespace {

constexpr uint64_t kChunkRows = PinnedRing::kSlotBytes / 4;   // 4-byte words per pinned slot
constexpr uint64_t kBlockRows = 1 << 14;                      // rows per worker task (384 tasks per pinned slot)
constexpr uint64_t kBlockGroups = 3264;                       // dense: groups per worker task (16 320 rows = 51 pieces = 204 units of 80 rows)
constexpr uint64_t kChunkGroups = 321 * kBlockGroups;         // dense: groups per pinned slot (16.0 MiB; one more may lead them)
static_assert(kChunkRows % kBlockRows == 0 && kChunkGroups * 16 + 16 <= PinnedRing::kSlotBytes, "a slot holds whole tasks");

struct PackArgs {
    const int64_t *start, *end, *annot;
    int shift;
    uint64_t global0;   // global row number of local row 0
    int64_t *boff;
    int64_t boff_size;
    // 1: three columns (what the ABI's column forms hand over); 3: ROWS -- filter_pq's own [M, 3] array, row-major
    // (memo_query.py:28-36): start = the array, end = start + 1, annot = start + 2, a row every three elements
    int stride = 1;
    int64_t s(uint64_t i) const { return start[i * (uint64_t)stride]; }
    int64_t e(uint64_t i) const { return end[i * (uint64_t)stride]; }
    int64_t a(uint64_t i) const { return annot[i * (uint64_t)stride]; }
};

// One row: the checks of memo_index_finalize, the bucket table, the rows with end < start.  Returns the row's fields
// through s / len / a12.  `ps` / `pb`: start and bucket of the row before.
struct RowScan {
    uint64_t top = 0;
    int bad = 0, wide = 0;
    int64_t ps, pb;
    RowScan(int64_t prev_start, int64_t prev_bucket) : ps(prev_start), pb(prev_bucket) {}
    inline void row(const PackArgs &A, uint64_t i, BlockResult &res, int64_t &s_out, int64_t &len_out, uint32_t &a12_out) {
        const int64_t s = A.s(i), e = A.e(i), a = A.a(i);
        bad |= (s < ps) ? 1 : 0;
        bad |= (s < 0) ? 2 : 0;
        bad |= ((uint64_t)a > 4095u) ? 4 : 0;
        bad |= (s >= kHostCoordLimit || e <= -kHostCoordLimit || e >= kHostCoordLimit) ? 8 : 0;
        ps = s;
        const int64_t len = (int64_t)((uint64_t)e - (uint64_t)s);
        if (e < s) {
            res.long_rows.push_back(s);
            res.long_rows.push_back(e);
            res.long_rows.push_back(a);
        }
        const uint32_t a12 = (uint32_t)a & 0xFFFu;
        top = a12 > top ? a12 : top;
        wide |= a12 > 255u;
        const int64_t bk = s >> A.shift;
        if (bk != pb) {  // first row of its bucket(s): boff[b] = lower_bound(start, b << shift)
            if (bk > pb && !bad && bk < A.boff_size)  // (q >= 0: the row before may belong to another task and be negative)
                for (int64_t q = pb < -1 ? 0 : pb + 1; q <= bk; ++q) A.boff[q] = (int64_t)(A.global0 + i);
            pb = bk;
        }
        s_out = s;
        len_out = e < s ? -1 : len;
        a12_out = a12;
    }
    void finish(BlockResult &res) const {
        res.max_annot = top;
        res.bad = bad;
        res.wide_annot = wide;
        res.over511 = top > 511u ? 1 : 0;
    }
};

// ------------------------------------------------------------------------------------------
// The packers proper.  RowScan above is the row-at-a-time statement of what a packer does (and what carries the few
// rows between pushes); the blocks of a push go through scan_piece: the same checks and fields for kPiece rows at a
// time, branch-free and flag-accumulating so that the compiler vectorises it (4 rows per AVX2 operation where the CPU
// has it: one thread packs ~2.5x the rows per second of the row-at-a-time loop), then the rare events -- a new bucket, a
// row with end < start -- from the marks the pass left.  Results are the same to the bit (tests/host_stub.cpp compares
// every packed row and bucket entry with its own restatement; GPU: test_dense_builder_equals_device_packing).
// ------------------------------------------------------------------------------------------
constexpr int kPiece = 320;  // rows per piece: 64 dense groups

struct PieceFlags {
    uint64_t unsorted = 0, sign = 0, annot_or = 0, coord = 0, any_long = 0, any_bucket = 0;
};

// FMT: 3 = dense field B (start mod 2^10 << 6 | min(len, 63)) + annot byte, 4 / 12 = the one-word formats
template <int FMT>
static inline __attribute__((always_inline)) void scan_piece_body(const int64_t *__restrict S, const int64_t *__restrict E,
                                                                   const int64_t *__restrict An, int n, int64_t prev_s, int shift,
                                                                   uint32_t *__restrict W, uint32_t *__restrict A8,
                                                                   uint8_t *__restrict mark, PieceFlags &f) {
    uint64_t unsorted = 0, sign = 0, annot_or = 0, coord = 0, any_long = 0, any_bucket = 0;
    // row 0 against the row before the piece; the others against their neighbour in the array
    {
        const int64_t s = S[0];
        unsorted |= (uint64_t)(s < prev_s);
        const uint8_t m = (uint8_t)((s >> shift) != (prev_s >> shift));
        mark[0] = m;
        any_bucket |= m;
    }
    for (int i = 1; i < n; ++i) {
        const int64_t s = S[i], p = S[i - 1];
        unsorted |= (uint64_t)(s < p);
        const uint8_t m = (uint8_t)((s >> shift) != (p >> shift));
        mark[i] = m;
        any_bucket |= m;
    }
    for (int i = 0; i < n; ++i) {
        const int64_t s = S[i], e = E[i], a = An[i];
        sign |= (uint64_t)s;
        annot_or |= (uint64_t)a;
        coord |= (uint64_t)(s >= kHostCoordLimit) | (uint64_t)(e <= -kHostCoordLimit) | (uint64_t)(e >= kHostCoordLimit);
        const uint64_t lng = (uint64_t)(e < s);
        any_long |= lng;
        mark[i] |= (uint8_t)(lng << 1);
        const uint64_t len = (uint64_t)e - (uint64_t)s;  // (end < start: huge, saturates to "never writes")
        if (FMT == 3) {
            const uint32_t l6 = len > 63u ? 63u : (uint32_t)len;
            W[i] = (((uint32_t)s & 1023u) << 6) | l6;
            A8[i] = (uint32_t)a & 0x1FFu;  // (nine bits: the ninth goes to the group's spare byte, dense_group)
        } else {
            const uint32_t l8 = len > 255u ? 255u : (uint32_t)len;
            const uint32_t a12 = (uint32_t)a & 0xFFFu;
            A8[i] = a12;
            W[i] = FMT == 12 ? l8 | (((uint32_t)s & 0xFFFu) << 8) | (a12 << 20) : ((uint32_t)s & 0xFFFFu) | (l8 << 16) | (a12 << 24);
        }
    }
    f.unsorted |= unsorted;
    f.sign |= sign;
    f.annot_or |= annot_or;
    f.coord |= coord;
    f.any_long |= any_long;
    f.any_bucket |= any_bucket;
}

typedef void (*ScanPieceFn)(const int64_t *, const int64_t *, const int64_t *, int, int64_t, int, uint32_t *, uint32_t *, uint8_t *,
                            PieceFlags &);

template <int FMT>
static void scan_piece_base(const int64_t *S, const int64_t *E, const int64_t *An, int n, int64_t prev_s, int shift, uint32_t *W,
                            uint32_t *A8, uint8_t *mark, PieceFlags &f) {
    scan_piece_body<FMT>(S, E, An, n, prev_s, shift, W, A8, mark, f);
}

#if defined(__x86_64__)
template <int FMT>
__attribute__((target("avx2"))) static void scan_piece_avx2(const int64_t *S, const int64_t *E, const int64_t *An, int n,
                                                            int64_t prev_s, int shift, uint32_t *W, uint32_t *A8, uint8_t *mark,
                                                            PieceFlags &f) {
    scan_piece_body<FMT>(S, E, An, n, prev_s, shift, W, A8, mark, f);
}
#endif

template <int FMT>
static ScanPieceFn scan_piece_for() {
#if defined(__x86_64__)
    static const bool avx2 = [] {
#ifdef MEMO_HOST_TEST_KNOBS  // (tests/test_host_sanitizers.py builds this file with it: every instance runs under the sanitizers)
        if (const char *v = getenv("MEMO_HOST_SIMD")) {
            if (atoi(v) == 0) return false;
        }
#endif
        return __builtin_cpu_supports("avx2") != 0;
    }();
    if (avx2) return scan_piece_avx2<FMT>;
#endif
    return scan_piece_base<FMT>;
}

// One block of a push: rows [i0, i1) through scan_piece, piece by piece.  emit(piece_row0, n, W, A8): the piece's packed
// fields, in order.  The block's checks, largest annot, bucket entries and long rows go where RowScan puts them.
template <int FMT, typename Emit>
static void scan_block(const PackArgs &A, uint64_t i0, uint64_t i1, int64_t prev_start, int64_t prev_bucket, BlockResult &res,
                       Emit emit) {
    const ScanPieceFn scan = scan_piece_for<FMT>();
    alignas(64) uint32_t W[kPiece], A8[kPiece];
    alignas(64) uint8_t mark[kPiece];
    PieceFlags f;
    uint32_t top = 0;
    int64_t ps = prev_start, pb = prev_bucket;
    int bad = 0;
    for (uint64_t i = i0; i < i1; i += kPiece) {
        const int n = (int)(i1 - i < (uint64_t)kPiece ? i1 - i : (uint64_t)kPiece);
        f.any_long = f.any_bucket = 0;
        if (A.stride == 1) {
            scan(A.start + i, A.end + i, A.annot + i, n, ps, A.shift, W, A8, mark, f);
        } else {  // rows: the piece's three columns first (7.5 KiB on the stack), then the same pass
            alignas(64) int64_t tS[kPiece], tE[kPiece], tA[kPiece];
            for (int j = 0; j < n; ++j) tS[j] = A.s(i + (uint64_t)j), tE[j] = A.e(i + (uint64_t)j), tA[j] = A.a(i + (uint64_t)j);
            scan(tS, tE, tA, n, ps, A.shift, W, A8, mark, f);
        }
        bad |= f.unsorted ? 1 : 0;
        bad |= (f.sign >> 63) ? 2 : 0;
        bad |= f.annot_or > 4095u ? 4 : 0;
        bad |= f.coord ? 8 : 0;
        for (int j = 0; j < n; ++j) top = A8[j] > top ? A8[j] : top;
        if (f.any_bucket | f.any_long) {
            for (int j = 0; j < n; ++j) {
                if (!mark[j]) continue;
                const int64_t s = A.s(i + (uint64_t)j);
                if (mark[j] & 2) {
                    res.long_rows.push_back(s);
                    res.long_rows.push_back(A.e(i + (uint64_t)j));
                    res.long_rows.push_back(A.a(i + (uint64_t)j));
                }
                const int64_t bk = s >> A.shift;
                if (bk != pb) {  // first row of its bucket(s): boff[b] = lower_bound(start, b << shift)
                    // (`bad` as RowScan has it at this row: set by any earlier piece, or by this one -- a piece that is
                    // bad anywhere fails the builder, what it wrote to the table is never read)
                    if (bk > pb && !bad && bk < A.boff_size)
                        for (int64_t q = pb < -1 ? 0 : pb + 1; q <= bk; ++q) A.boff[q] = (int64_t)(A.global0 + i + (uint64_t)j);
                    pb = bk;
                }
            }
        }
        ps = A.s(i + (uint64_t)n - 1);
        emit(i, n, W, A8);
    }
    res.max_annot = top;
    res.bad = bad;
    res.wide_annot = f.annot_or > 255u ? 1 : 0;
    res.over511 = f.annot_or > 511u ? 1 : 0;
}

// the dense row: B = (start mod 2^10) << 6 | min(end - start, 63); end < start packs as "never writes" (k - 1 <= 63)
inline uint32_t dense_b(int64_t s, int64_t len) {
    return (((uint32_t)s & 1023u) << 6) | ((uint64_t)len > 63u ? 63u : (uint32_t)len);
}

// five rows -> one 16-byte group (PackedRows3, memo_sweep.h): dword j = B_j | X_j << 16 | A_j << 24, the fifth row in
// the spare bytes X
// (annots of nine bits -- indexes of 256 .. 511 genomes --: the ninth bit of row i at bit 16 + i of the last dword, the byte no
// row used; memo_index.hip: pack3_rows_kernel builds the same on the device)
inline void dense_group(const uint32_t *B, const uint32_t *Aa, uint32_t *out) {
    const uint32_t hi = ((Aa[0] >> 8) & 1u) | (((Aa[1] >> 8) & 1u) << 1) | (((Aa[2] >> 8) & 1u) << 2) | (((Aa[3] >> 8) & 1u) << 3) |
                        (((Aa[4] >> 8) & 1u) << 4);
    out[0] = B[0] | ((B[4] & 0xFFu) << 16) | ((Aa[0] & 0xFFu) << 24);
    out[1] = B[1] | ((B[4] >> 8) << 16) | ((Aa[1] & 0xFFu) << 24);
    out[2] = B[2] | ((Aa[4] & 0xFFu) << 16) | ((Aa[2] & 0xFFu) << 24);
    out[3] = B[3] | (hi << 16) | ((Aa[3] & 0xFFu) << 24);
}

// ------------------------------------------------------------------------------------------
// The hand-written packers (AVX-512 F / BW / DQ / VL / VBMI + BMI2: Zen 4 and later, Ice Lake and later).  The GPU boxes
// of this pool grant the process 16 CPUs' worth of time per 100 ms (memo_cpus.h), so what the seam pays for a row is
// CPU-seconds, and the compiler's rendering of scan_piece -- 64-bit unsigned minima and narrowing emulated in AVX2, the
// groups assembled by scalar code -- costs ~3.6 ns per row there: 1.8 CPU-seconds for config 3's 5 * 10^8 rows, more
// than a period's quota (profiles/r06_oneshot.txt).  Here 16 rows are one step: six loads, the checks as mask compares,
// the fields narrowed by one two-source permute each and merged by ternary logic; 80 rows (five steps) become 16 groups
// by four byte permutes (vpermi2b) and leave as four 64-byte stores -- non-temporal where the slot is aligned, the
// pinned buffer is read by the DMA engine, not by a CPU.  The nine-bit annots' spare byte comes from the rows' mask bits
// through pdep.  Same bits as scan_block + dense_group (tests/host_stub.cpp runs every instance against its
// restatement; MEMO_HOST_SIMD chooses one in the sanitizer builds).  Blocks whose row count is not a multiple of 80 (16)
// finish through scan_block.
// ------------------------------------------------------------------------------------------
#if defined(__x86_64__)
#define MEMO_T512 __attribute__((target("avx512f,avx512bw,avx512dq,avx512vl,avx512vbmi,bmi2")))

static int host_simd_level() {  // 0 plain, 1 AVX2 (compiler-vectorised scan_piece), 2 the AVX-512 packers
    static const int level = [] {
        int best = 0;
        if (__builtin_cpu_supports("avx2")) best = 1;
        if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") &&
            __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512vbmi") && __builtin_cpu_supports("bmi2"))
            best = 2;
#ifdef MEMO_HOST_TEST_KNOBS  // (tests/test_host_sanitizers.py builds this file with it: every instance runs under the sanitizers)
        if (const char *v = getenv("MEMO_HOST_SIMD")) {
            const int want = atoi(v);
            if (want >= 0 && want < best) best = want;
        }
#endif
        return best;
    }();
    return level;
}

struct Fast512 {  // what a block's fast part leaves for its tail and its result
    int64_t ps, pb;        // start and bucket of the last row done
    unsigned uns = 0, neg = 0, coord = 0;
    uint64_t annot_or = 0;
    uint32_t top = 0;
};

// the rare rows of a 16-row step: first row of a bucket, end < start (as scan_block treats them)
static inline void events16(const PackArgs &A, uint64_t row0, unsigned kb, unsigned kl, bool bad, int64_t &pb, BlockResult &res) {
    unsigned m = kb | kl;
    while (m) {
        const int j = __builtin_ctz(m);
        m &= m - 1;
        const uint64_t i = row0 + (uint64_t)j;
        const int64_t s = A.s(i);
        if ((kl >> j) & 1u) {
            res.long_rows.push_back(s);
            res.long_rows.push_back(A.e(i));
            res.long_rows.push_back(A.a(i));
        }
        const int64_t bk = s >> A.shift;
        if (bk != pb) {
            if (bk > pb && !bad && bk < A.boff_size)
                for (int64_t q = pb < -1 ? 0 : pb + 1; q <= bk; ++q) A.boff[q] = (int64_t)(A.global0 + i);
            pb = bk;
        }
    }
}

// Eight ROWS (24 consecutive qwords: start at qword 3 j, end at 3 j + 1, annot at 3 j + 2 of row j) -> their starts, ends, annots.
// A field's eight values sit in eight DIFFERENT lanes of the three vectors (3 j + c mod 8 runs through all of them), so two masked
// blends collect them into one vector and one permute puts them in row order: six blends + three permutes per eight rows (twelve
// two-source permutes at first: the rows form was 20 % behind the columns).
MEMO_T512 static inline void rows8(const int64_t *q, __m512i &sv, __m512i &ev, __m512i &av) {
    const __m512i s_ix = _mm512_set_epi64(5, 2, 7, 4, 1, 6, 3, 0), e_ix = _mm512_set_epi64(6, 3, 0, 5, 2, 7, 4, 1),
                  a_ix = _mm512_set_epi64(7, 4, 1, 6, 3, 0, 5, 2);
    const __m512i q0 = _mm512_loadu_si512(q), q1 = _mm512_loadu_si512(q + 8), q2 = _mm512_loadu_si512(q + 16);
    // starts: q0 lanes 0 3 6, q1 lanes 1 4 7, q2 lanes 2 5;  ends: q0 1 4 7, q1 2 5, q2 0 3 6;  annots: q0 2 5, q1 0 3 6, q2 1 4 7
    const __m512i sb = _mm512_mask_blend_epi64(0x24, _mm512_mask_blend_epi64(0x92, q0, q1), q2);
    const __m512i eb = _mm512_mask_blend_epi64(0x49, _mm512_mask_blend_epi64(0x24, q0, q1), q2);
    const __m512i ab = _mm512_mask_blend_epi64(0x92, _mm512_mask_blend_epi64(0x49, q0, q1), q2);
    sv = _mm512_permutexvar_epi64(s_ix, sb);
    ev = _mm512_permutexvar_epi64(e_ix, eb);
    av = _mm512_permutexvar_epi64(a_ix, ab);
}

// sixteen rows from row i on as two vectors each of starts, ends, annots: three columns (six loads), or ROWS -- 48 consecutive
// qwords, start / end / annot of a row side by side -- taken apart by blends and permutes
template <bool ROWS>
MEMO_T512 static inline void load16(const PackArgs &A, uint64_t i, __m512i &s0, __m512i &s1, __m512i &e0, __m512i &e1, __m512i &a0,
                                    __m512i &a1) {
    if (!ROWS) {
        s0 = _mm512_loadu_si512(A.start + i), s1 = _mm512_loadu_si512(A.start + i + 8);
        e0 = _mm512_loadu_si512(A.end + i), e1 = _mm512_loadu_si512(A.end + i + 8);
        a0 = _mm512_loadu_si512(A.annot + i), a1 = _mm512_loadu_si512(A.annot + i + 8);
        return;
    }
    rows8(A.start + 3 * i, s0, e0, a0);
    rows8(A.start + 3 * i + 24, s1, e1, a1);
}

struct Perm512 {  // byte permutes of the dense groups: output vector j (four groups) from row vectors j and j + 1
    alignas(64) uint8_t idx[4][64];
    alignas(64) uint8_t hidx[4][64];
    Perm512() {
        for (int j = 0; j < 4; ++j)
            for (int t = 0; t < 4; ++t) {
                const int r0 = 20 * j + 5 * t;  // first row of the group (row r, byte c sits at 4 r + c of the unit's 320 bytes)
                auto at = [&](int r, int c) { return (uint8_t)(4 * r + c - 64 * j); };
                uint8_t *o = idx[j] + 16 * t;
                o[0] = at(r0, 0), o[1] = at(r0, 1), o[2] = at(r0 + 4, 0), o[3] = at(r0, 2);
                o[4] = at(r0 + 1, 0), o[5] = at(r0 + 1, 1), o[6] = at(r0 + 4, 1), o[7] = at(r0 + 1, 2);
                o[8] = at(r0 + 2, 0), o[9] = at(r0 + 2, 1), o[10] = at(r0 + 4, 2), o[11] = at(r0 + 2, 2);
                o[12] = at(r0 + 3, 0), o[13] = at(r0 + 3, 1), o[14] = 0, o[15] = at(r0 + 3, 2);
                for (int c = 0; c < 16; ++c) hidx[j][16 * t + c] = 0;
                hidx[j][16 * t + 14] = (uint8_t)(4 * j + t);  // the group's spare byte: byte 4 j + t of the sixteen
            }
    }
};

// rows [i0, i0 + 80 units) -> 16 units groups at out; returns the state after the last row
template <bool ROWS>
MEMO_T512 static void dense_units_512(const PackArgs &A, uint64_t i0, uint64_t units, int64_t prev_start, int64_t prev_bucket,
                                      uint32_t *out, BlockResult &res, Fast512 &st) {
    static const Perm512 P;
    const __m512i idx0 = _mm512_load_si512(P.idx[0]), idx1 = _mm512_load_si512(P.idx[1]), idx2 = _mm512_load_si512(P.idx[2]),
                  idx3 = _mm512_load_si512(P.idx[3]);
    const __mmask64 keep = ~0x4000400040004000ull, spare = 0x4000400040004000ull;  // (bytes 14, 30, 46, 62: the spare bytes)
    const __m512i lim = _mm512_set1_epi64(kHostCoordLimit), limm1 = _mm512_set1_epi64(kHostCoordLimit - 1),
                  lim2 = _mm512_set1_epi64(2 * kHostCoordLimit - 2), c63 = _mm512_set1_epi64(63),
                  thr = _mm512_set1_epi64((int64_t)1 << A.shift);
    const __m512i even = _mm512_set_epi32(30, 28, 26, 24, 22, 20, 18, 16, 14, 12, 10, 8, 6, 4, 2, 0);
    const __m512i cB = _mm512_set1_epi32(0xFFC0), cA = _mm512_set1_epi32(0xFF0000), c256 = _mm512_set1_epi32(256),
                  c1FF = _mm512_set1_epi32(0x1FF);
    const bool nt = ((uintptr_t)out & 63u) == 0;
    __m512i sprev = _mm512_set1_epi64(prev_start), acc_a = _mm512_setzero_si512(), top = _mm512_setzero_si512();
    unsigned uns = 0, neg = 0, coord = 0;
    int64_t pb = prev_bucket;
    for (uint64_t u = 0; u < units; ++u) {
        __m512i R[5];
        unsigned k9[5];
        for (int v = 0; v < 5; ++v) {
            const uint64_t i = i0 + 80 * u + 16 * (uint64_t)v;
            __m512i s0, s1, e0, e1, a0, a1;
            load16<ROWS>(A, i, s0, s1, e0, e1, a0, a1);
            const __m512i p0 = _mm512_alignr_epi64(s0, sprev, 7), p1 = _mm512_alignr_epi64(s1, s0, 7);
            sprev = s1;
            uns |= (unsigned)_mm512_cmplt_epi64_mask(s0, p0) | (unsigned)_mm512_cmplt_epi64_mask(s1, p1);
            const unsigned kb = (unsigned)_mm512_cmpge_epu64_mask(_mm512_xor_si512(s0, p0), thr) |
                                ((unsigned)_mm512_cmpge_epu64_mask(_mm512_xor_si512(s1, p1), thr) << 8);
            const unsigned kl = (unsigned)_mm512_cmplt_epi64_mask(e0, s0) | ((unsigned)_mm512_cmplt_epi64_mask(e1, s1) << 8);
            neg |= (unsigned)_mm512_movepi64_mask(s0) | (unsigned)_mm512_movepi64_mask(s1);
            coord |= (unsigned)_mm512_cmpge_epi64_mask(s0, lim) | (unsigned)_mm512_cmpge_epi64_mask(s1, lim) |
                     (unsigned)_mm512_cmpgt_epu64_mask(_mm512_add_epi64(e0, limm1), lim2) |
                     (unsigned)_mm512_cmpgt_epu64_mask(_mm512_add_epi64(e1, limm1), lim2);
            acc_a = _mm512_ternarylogic_epi64(acc_a, a0, a1, 0xFE);
            const __m512i l0 = _mm512_min_epu64(_mm512_sub_epi64(e0, s0), c63), l1 = _mm512_min_epu64(_mm512_sub_epi64(e1, s1), c63);
            const __m512i s32 = _mm512_permutex2var_epi32(s0, even, s1), a32 = _mm512_permutex2var_epi32(a0, even, a1),
                          l32 = _mm512_permutex2var_epi32(l0, even, l1);
            const __m512i w = _mm512_ternarylogic_epi32(_mm512_slli_epi32(s32, 6), cB, l32, 0xEA);   // (s << 6 & 0xFFC0) | l
            R[v] = _mm512_ternarylogic_epi32(_mm512_slli_epi32(a32, 16), cA, w, 0xEA);               // (a << 16 & 0xFF0000) | w
            k9[v] = (unsigned)_mm512_test_epi32_mask(a32, c256);
            top = _mm512_max_epu32(top, _mm512_and_si512(a32, c1FF));
            if (kb | kl) events16(A, i, kb, kl, (uns | neg | coord) != 0, pb, res);
        }
        __m512i g0 = _mm512_maskz_permutex2var_epi8(keep, R[0], idx0, R[1]), g1 = _mm512_maskz_permutex2var_epi8(keep, R[1], idx1, R[2]),
                g2 = _mm512_maskz_permutex2var_epi8(keep, R[2], idx2, R[3]), g3 = _mm512_maskz_permutex2var_epi8(keep, R[3], idx3, R[4]);
        if (k9[0] | k9[1] | k9[2] | k9[3] | k9[4]) {  // ninth annot bits: bit r of the unit's 80-bit string belongs to row r
            const uint64_t lo = (uint64_t)k9[0] | ((uint64_t)k9[1] << 16) | ((uint64_t)k9[2] << 32) | ((uint64_t)k9[3] << 48);
            const uint64_t h0 = _pdep_u64(lo, 0x1F1F1F1F1F1F1F1Full);                                   // groups 0 .. 7: bits 0 .. 39
            const uint64_t h1 = _pdep_u64((lo >> 40) | ((uint64_t)k9[4] << 24), 0x1F1F1F1F1F1F1F1Full);  // groups 8 .. 15: bits 40 .. 79
            const __m512i H = _mm512_castsi128_si512(_mm_set_epi64x((long long)h1, (long long)h0));
            g0 = _mm512_or_si512(g0, _mm512_maskz_permutexvar_epi8(spare, _mm512_load_si512(P.hidx[0]), H));
            g1 = _mm512_or_si512(g1, _mm512_maskz_permutexvar_epi8(spare, _mm512_load_si512(P.hidx[1]), H));
            g2 = _mm512_or_si512(g2, _mm512_maskz_permutexvar_epi8(spare, _mm512_load_si512(P.hidx[2]), H));
            g3 = _mm512_or_si512(g3, _mm512_maskz_permutexvar_epi8(spare, _mm512_load_si512(P.hidx[3]), H));
        }
        uint32_t *o = out + 64 * u;
        if (nt) {
            _mm512_stream_si512((__m512i *)o, g0), _mm512_stream_si512((__m512i *)(o + 16), g1);
            _mm512_stream_si512((__m512i *)(o + 32), g2), _mm512_stream_si512((__m512i *)(o + 48), g3);
        } else {
            _mm512_storeu_si512(o, g0), _mm512_storeu_si512(o + 16, g1), _mm512_storeu_si512(o + 32, g2), _mm512_storeu_si512(o + 48, g3);
        }
    }
    if (nt) _mm_sfence();  // (the stores are weakly ordered: they are in memory before the task reports itself done)
    st.ps = units ? A.s(i0 + 80 * units - 1) : prev_start;
    st.pb = pb;
    st.uns = uns, st.neg = neg, st.coord = coord;
    st.annot_or = (uint64_t)_mm512_reduce_or_epi64(acc_a);
    st.top = _mm512_reduce_max_epu32(top);
}

// rows [i0, i0 + 16 steps) -> one word each (format 4 or 12) at pk[i0 ...]
template <int FMT, bool ROWS>
MEMO_T512 static void word_steps_512(const PackArgs &A, uint64_t i0, uint64_t steps, int64_t prev_start, int64_t prev_bucket,
                                     uint32_t *pk, BlockResult &res, Fast512 &st) {
    const __m512i lim = _mm512_set1_epi64(kHostCoordLimit), limm1 = _mm512_set1_epi64(kHostCoordLimit - 1),
                  lim2 = _mm512_set1_epi64(2 * kHostCoordLimit - 2), c255 = _mm512_set1_epi64(255),
                  thr = _mm512_set1_epi64((int64_t)1 << A.shift);
    const __m512i even = _mm512_set_epi32(30, 28, 26, 24, 22, 20, 18, 16, 14, 12, 10, 8, 6, 4, 2, 0);
    const __m512i cFFFF = _mm512_set1_epi32(0xFFFF), cFFF = _mm512_set1_epi32(0xFFF), cFFF00 = _mm512_set1_epi32(0xFFF00);
    __m512i sprev = _mm512_set1_epi64(prev_start), acc_a = _mm512_setzero_si512(), top = _mm512_setzero_si512();
    unsigned uns = 0, neg = 0, coord = 0;
    int64_t pb = prev_bucket;
    bool any_nt = false;
    for (uint64_t v = 0; v < steps; ++v) {
        const uint64_t i = i0 + 16 * v;
        __m512i s0, s1, e0, e1, a0, a1;
        load16<ROWS>(A, i, s0, s1, e0, e1, a0, a1);
        const __m512i p0 = _mm512_alignr_epi64(s0, sprev, 7), p1 = _mm512_alignr_epi64(s1, s0, 7);
        sprev = s1;
        uns |= (unsigned)_mm512_cmplt_epi64_mask(s0, p0) | (unsigned)_mm512_cmplt_epi64_mask(s1, p1);
        const unsigned kb = (unsigned)_mm512_cmpge_epu64_mask(_mm512_xor_si512(s0, p0), thr) |
                            ((unsigned)_mm512_cmpge_epu64_mask(_mm512_xor_si512(s1, p1), thr) << 8);
        const unsigned kl = (unsigned)_mm512_cmplt_epi64_mask(e0, s0) | ((unsigned)_mm512_cmplt_epi64_mask(e1, s1) << 8);
        neg |= (unsigned)_mm512_movepi64_mask(s0) | (unsigned)_mm512_movepi64_mask(s1);
        coord |= (unsigned)_mm512_cmpge_epi64_mask(s0, lim) | (unsigned)_mm512_cmpge_epi64_mask(s1, lim) |
                 (unsigned)_mm512_cmpgt_epu64_mask(_mm512_add_epi64(e0, limm1), lim2) |
                 (unsigned)_mm512_cmpgt_epu64_mask(_mm512_add_epi64(e1, limm1), lim2);
        acc_a = _mm512_ternarylogic_epi64(acc_a, a0, a1, 0xFE);
        const __m512i l0 = _mm512_min_epu64(_mm512_sub_epi64(e0, s0), c255), l1 = _mm512_min_epu64(_mm512_sub_epi64(e1, s1), c255);
        const __m512i s32 = _mm512_permutex2var_epi32(s0, even, s1), l32 = _mm512_permutex2var_epi32(l0, even, l1);
        const __m512i a12 = _mm512_and_si512(_mm512_permutex2var_epi32(a0, even, a1), cFFF);
        __m512i w;
        if (FMT == 12)  // len | (start mod 2^12) << 8 | annot << 20
            w = _mm512_or_si512(_mm512_ternarylogic_epi32(_mm512_slli_epi32(s32, 8), cFFF00, l32, 0xEA), _mm512_slli_epi32(a12, 20));
        else            // start mod 2^16 | len << 16 | annot << 24
            w = _mm512_or_si512(_mm512_ternarylogic_epi32(s32, cFFFF, _mm512_slli_epi32(l32, 16), 0xEA), _mm512_slli_epi32(a12, 24));
        top = _mm512_max_epu32(top, a12);
        uint32_t *o = pk + i;
        if (((uintptr_t)o & 63u) == 0) {
            _mm512_stream_si512((__m512i *)o, w);
            any_nt = true;
        } else {
            _mm512_storeu_si512(o, w);
        }
        if (kb | kl) events16(A, i, kb, kl, (uns | neg | coord) != 0, pb, res);
    }
    if (any_nt) _mm_sfence();
    st.ps = steps ? A.s(i0 + 16 * steps - 1) : prev_start;
    st.pb = pb;
    st.uns = uns, st.neg = neg, st.coord = coord;
    st.annot_or = (uint64_t)_mm512_reduce_or_epi64(acc_a);
    st.top = _mm512_reduce_max_epu32(top);
}

// a block's result from its fast part (st) and the tail scan_block did (res already holds the tail's)
static void fold_fast(const Fast512 &st, bool had_tail, BlockResult &res) {
    int bad = (st.uns ? 1 : 0) | (st.neg ? 2 : 0) | (st.annot_or > 4095u ? 4 : 0) | (st.coord ? 8 : 0);
    if (!had_tail) {
        res.max_annot = 0;
        res.bad = 0;
        res.wide_annot = res.over511 = 0;
    }
    res.bad |= bad;
    res.max_annot = st.top > res.max_annot ? st.top : res.max_annot;
    res.wide_annot |= st.annot_or > 255u ? 1 : 0;
    res.over511 |= st.annot_or > 511u ? 1 : 0;
}
#else
static int host_simd_level() { return 0; }
#endif

// rows [i0, i1) -> words (format 4 or 12), pk[i] for row i.  end < start (handled by long_rows_*_kernel) packs as
// "never writes", like len >= 255.
void pack_words(const PackArgs &A, uint64_t i0, uint64_t i1, int64_t prev_start, int64_t prev_bucket, uint32_t *pk,
                int fmt, BlockResult &res) {
    auto emit = [&](uint64_t at, int n, const uint32_t *W, const uint32_t *) { memcpy(pk + at, W, (size_t)n * 4); };
#if defined(__x86_64__)
    if (host_simd_level() >= 2 && i1 - i0 >= 16) {
        const uint64_t steps = (i1 - i0) / 16, mid = i0 + 16 * steps;
        Fast512 st;
        std::vector<int64_t> head_long;
        if (fmt == 12)
            A.stride == 1 ? word_steps_512<12, false>(A, i0, steps, prev_start, prev_bucket, pk, res, st)
                          : word_steps_512<12, true>(A, i0, steps, prev_start, prev_bucket, pk, res, st);
        else
            A.stride == 1 ? word_steps_512<4, false>(A, i0, steps, prev_start, prev_bucket, pk, res, st)
                          : word_steps_512<4, true>(A, i0, steps, prev_start, prev_bucket, pk, res, st);
        if (mid < i1) {
            head_long.swap(res.long_rows);  // (scan_block sets the other fields; the long rows of the fast part stay in front)
            if (fmt == 12)
                scan_block<12>(A, mid, i1, st.ps, st.pb, res, emit);
            else
                scan_block<4>(A, mid, i1, st.ps, st.pb, res, emit);
            head_long.insert(head_long.end(), res.long_rows.begin(), res.long_rows.end());
            res.long_rows.swap(head_long);
        }
        fold_fast(st, mid < i1, res);
        return;
    }
#endif
    if (fmt == 12)
        scan_block<12>(A, i0, i1, prev_start, prev_bucket, res, emit);
    else
        scan_block<4>(A, i0, i1, prev_start, prev_bucket, res, emit);
}

// rows [i0, i0 + 5 * groups) -> groups at out (4 dwords each)
void pack_dense(const PackArgs &A, uint64_t i0, uint64_t groups, int64_t prev_start, int64_t prev_bucket, uint32_t *out,
                BlockResult &res) {
    static_assert(kPiece % 5 == 0, "a piece is whole groups");
    auto tail = [&](uint64_t r0, uint64_t g, int64_t ps, int64_t pb, uint32_t *o0) {
        scan_block<3>(A, r0, r0 + 5 * g, ps, pb, res, [&](uint64_t at, int n, const uint32_t *W, const uint32_t *A8) {
            uint32_t *o = o0 + 4 * ((at - r0) / 5);
            for (int j = 0; j + 5 <= n; j += 5, o += 4) dense_group(W + j, A8 + j, o);
        });
    };
#if defined(__x86_64__)
    if (host_simd_level() >= 2 && groups >= 16) {
        const uint64_t units = groups / 16, mid = i0 + 80 * units;
        Fast512 st;
        std::vector<int64_t> head_long;
        A.stride == 1 ? dense_units_512<false>(A, i0, units, prev_start, prev_bucket, out, res, st)
                      : dense_units_512<true>(A, i0, units, prev_start, prev_bucket, out, res, st);
        if (groups > 16 * units) {
            head_long.swap(res.long_rows);
            tail(mid, groups - 16 * units, st.ps, st.pb, out + 64 * units);
            head_long.insert(head_long.end(), res.long_rows.begin(), res.long_rows.end());
            res.long_rows.swap(head_long);
        }
        fold_fast(st, groups > 16 * units, res);
        if (res.over511) res.bad |= 16;
        return;
    }
#endif
    tail(i0, groups, prev_start, prev_bucket, out);
    if (res.over511) res.bad |= 16;
}

const char *bad_message(int bad) {
    return bad & 1    ? "rows are not sorted by start: not packable on the host"
           : bad & 2  ? "rows with a negative start cannot be packed"
           : bad & 4  ? "rows with an annot outside [0, 4095] do not fit the one-word formats"
           : bad & 8  ? "rows have coordinates beyond +-2^61"
                      : "rows with an annot above 511 do not fit the dense rows: take the 4-byte rows";
}

int merge_results(memo_builder *b, std::vector<BlockResult> &res) {
    for (BlockResult &r : res) {
        if (r.max_annot > b->max_annot) b->max_annot = r.max_annot;
        if (!r.long_rows.empty()) {
            b->long_rows.insert(b->long_rows.end(), r.long_rows.begin(), r.long_rows.end());
            r.long_rows.clear();
            if (b->long_rows.size() / 3 > kMaxLongRows)
                return builder_fail(b, MEMO_ELONGROW, "more than 2^22 rows have end < start: not a MEMO overlap index");
        }
    }
    return MEMO_OK;
}

// ------------------------------------------------------------------------------------------
// One push as ONE job of the pool.  The rows are cut into chunks (what a pinned slot holds) of blocks (a worker's task).
// Workers take blocks in order from one counter and pack them into the chunk's slot as soon as the slot is free; the
// CALLER's thread issues the copies -- chunk c leaves the moment its last block reports itself done -- watches the copy
// events without blocking, frees slots, and packs blocks itself whenever neither has anything for it.  No fork-join per
// chunk (rounds 2-5: 67 of them per call on config 3, every one a wake-up of the pool and a wait for its slowest
// thread), no HIP call off the caller's thread, and packing never waits for PCIe unless all four slots are full.
//
// pack(block, chunk, slot buffer, result): rows of the block -> the slot; blocks_of(chunk); issue(chunk, slot buffer).
// ------------------------------------------------------------------------------------------
struct PushPipe {
    memo_builder *b;
    PinnedRing *ring;
    uint64_t nblocks = 0, per_chunk = 1, nchunks = 0;
    int base_slot = 0;
    std::atomic<uint64_t> next{0};
    std::atomic<uint64_t> free_upto{0};  // chunks below this number may be packed: their slot is theirs
    std::atomic<int> stop{0};            // a block refused its rows, or a HIP call failed: nobody takes another block
    std::vector<std::atomic<uint32_t>> done;  // per chunk: blocks packed
    std::vector<BlockResult> res;             // per block
    char *buf[PinnedRing::kSlots] = {nullptr, nullptr, nullptr, nullptr};

    PushPipe(memo_builder *b_, uint64_t nblocks_, uint64_t per_chunk_)
        : b(b_), ring(b_->ring), nblocks(nblocks_), per_chunk(per_chunk_), nchunks((nblocks_ + per_chunk_ - 1) / per_chunk_),
          base_slot(b_->ring->next), done((nblocks_ + per_chunk_ - 1) / per_chunk_), res(nblocks_) {
        for (auto &d : done) d.store(0, std::memory_order_relaxed);
    }
    int slot_of(uint64_t chunk) const { return (int)((chunk + (uint64_t)base_slot) % PinnedRing::kSlots); }
    uint64_t blocks_of(uint64_t chunk) const { return chunk + 1 < nchunks ? per_chunk : nblocks - chunk * per_chunk; }

    template <typename Pack>
    void pack_one(uint64_t blk, Pack &pack) {
        const uint64_t c = blk / per_chunk;
        pack(blk, c, buf[slot_of(c)], res[blk]);
        if (res[blk].bad) stop.store(1, std::memory_order_relaxed);
        done[c].fetch_add(1, std::memory_order_release);
    }

    template <typename Pack>
    void worker(Pack &pack) {
        for (;;) {
            if (stop.load(std::memory_order_relaxed)) return;
            const uint64_t blk = next.fetch_add(1, std::memory_order_relaxed);
            if (blk >= nblocks) return;
            const uint64_t c = blk / per_chunk;
            for (unsigned spins = 0; free_upto.load(std::memory_order_acquire) <= c; ++spins) {  // the slot is still on its way out
                if (stop.load(std::memory_order_relaxed)) {  // (a claimed block must still be counted: the issuer may wait for its chunk)
                    done[c].fetch_add(1, std::memory_order_release);
                    return;
                }
                if (spins < 256) {
#if defined(__x86_64__)
                    _mm_pause();
#endif
                } else {
                    std::this_thread::sleep_for(std::chrono::microseconds(20));  // (spinning spends the cgroup's CPU quota)
                }
            }
            if (stop.load(std::memory_order_relaxed)) {  // (the issuer gave up and let everybody through: the slot may not be ours)
                done[c].fetch_add(1, std::memory_order_release);
                return;
            }
            pack_one(blk, pack);
        }
    }

    // the caller's loop; returns a HIP failure code or MEMO_OK (refused rows are in res[].bad)
    template <typename Pack, typename Issue>
    int drive(Pack &pack, Issue &issue) {
        HostPool &pool = HostPool::get();
        auto task = [&](int) { worker(pack); };
        const bool threaded = nblocks > 2 && pool.threads() > 1;
        if (threaded) pool.begin(pool.threads() - 1, [](void *c, int t) { (*static_cast<decltype(task) *>(c))(t); }, &task);
        int rc = MEMO_OK;
        uint64_t opened = 0, issued = 0, landed = 0;  // chunks whose slot is open / whose copy is queued / whose copy has finished
        while (issued < nchunks && !rc) {
            if (stop.load(std::memory_order_relaxed)) break;
            bool progress = false;
            while (opened < nchunks && opened < landed + PinnedRing::kSlots) {  // open the next slot
                const int s = slot_of(opened);
                if ((rc = ring->buffer(s, &buf[s]))) break;
                if (opened < (uint64_t)PinnedRing::kSlots && (rc = ring->wait(s))) break;  // (an earlier push's copy)
                free_upto.store(++opened, std::memory_order_release);
                progress = true;
            }
            if (rc) break;
            if (done[issued].load(std::memory_order_acquire) == blocks_of(issued)) {  // the next chunk is whole: send it
                const int s = slot_of(issued);
                if ((rc = issue(issued, buf[s])) || (rc = ring->mark(s))) break;
                ++issued;
                continue;
            }
            if (landed < issued) {  // has the oldest copy finished?
                bool idle = false;
                if ((rc = ring->poll(slot_of(landed), &idle))) break;
                if (idle) {
                    ++landed;
                    continue;
                }
            }
            // nothing to issue, nothing to free: pack a block (exactly the one looked at: a later one might wait for a slot only this thread opens)
            uint64_t blk = next.load(std::memory_order_relaxed);
            if (blk < nblocks && blk / per_chunk < opened) {
                if (next.compare_exchange_strong(blk, blk + 1, std::memory_order_relaxed)) pack_one(blk, pack);
                continue;
            }
            if (!progress) {
#if defined(__x86_64__)
                _mm_pause();
#endif
            }
        }
        if (rc || stop.load(std::memory_order_relaxed)) {
            stop.store(1, std::memory_order_relaxed);
            free_upto.store(nchunks, std::memory_order_release);  // (nobody waits for a slot any more; workers see `stop` first)
        }
        if (threaded) pool.end();
        ring->next = slot_of(issued);
        return rc;
    }
};

int refuse(memo_builder *b, int bad) {
    b->why = bad;
    return builder_fail(b, MEMO_EUNPACKABLE, bad_message(bad));
}

// a sample of the annots decides the format the packing starts in (a wrong guess costs a restart of the push, below)
bool sample_has_wide_annot(const PackArgs &A, uint64_t rows) {
    const uint64_t step = rows / 2048 + 1;
    for (uint64_t i = 0; i < rows; i += step)
        if ((uint64_t)A.a(i) > 255u) return true;
    return (uint64_t)A.a(rows - 1) > 255u;
}

int push_words(memo_builder *b, const PackArgs &A, uint64_t rows) {
    if (b->fmt == 4 && sample_has_wide_annot(A, rows)) {  // switch before anything of this push is on its way
        int rc = hp::stream_sync(b->ring->stream);
        if (!rc && b->rows) rc = hp::widen_annots(b->d_pk, b->rows, b->ring->stream);
        if (rc) return builder_fail(b, rc, "rewriting the rows with 12-bit annots failed");
        b->fmt = 12;
    }
    for (int pass = 0; pass < 2; ++pass) {  // a second pass only when this push is the first with an annot > 255 and the sample missed it
        const int fmt = b->fmt;
        const uint64_t nblocks = (rows + kBlockRows - 1) / kBlockRows;
        PushPipe pipe(b, nblocks, kChunkRows / kBlockRows);
        auto pack = [&](uint64_t blk, uint64_t c, char *buf, BlockResult &r) {
            const uint64_t i0 = blk * kBlockRows, i1 = i0 + kBlockRows < rows ? i0 + kBlockRows : rows;
            const bool first = i0 == 0;
            const int64_t prev_start = first ? (b->any ? b->last_start : INT64_MIN) : A.s(i0 - 1);
            const int64_t prev_bucket = first ? b->last_bucket : (A.s(i0 - 1) >> b->bshift);
            pack_words(A, i0, i1, prev_start, prev_bucket, reinterpret_cast<uint32_t *>(buf) - c * kChunkRows, fmt, r);
            if (r.wide_annot && fmt == 4) pipe.stop.store(1, std::memory_order_relaxed);
        };
        auto issue = [&](uint64_t c, char *buf) {
            const uint64_t c0 = c * kChunkRows, cn = rows - c0 < kChunkRows ? rows - c0 : kChunkRows;
            return hp::copy_h2d_async(b->d_pk + b->rows + c0, buf, cn * 4, b->ring->stream);
        };
        int rc = pipe.drive(pack, issue);
        if (rc) return builder_fail(b, rc, "copy to the device failed");
        int bad = 0, wide = 0;
        for (const BlockResult &r : pipe.res) {
            bad |= r.bad;
            wide |= r.wide_annot;
        }
        if (bad) return refuse(b, bad);
        if (wide && fmt == 4) {  // switch the index to 12-bit annots: rewrite what is on the device, redo this push
            if ((rc = hp::stream_sync(b->ring->stream))) return builder_fail(b, rc, "copy stream failed");
            for (int s = 0; s < PinnedRing::kSlots; ++s) b->ring->in_flight[s] = false;
            if (b->rows && (rc = hp::widen_annots(b->d_pk, b->rows, b->ring->stream)))
                return builder_fail(b, rc, "rewriting the rows with 12-bit annots failed");
            b->fmt = 12;
            continue;
        }
        return merge_results(b, pipe.res);
    }
    return MEMO_OK;
}

int carry_rows(memo_builder *b, const PackArgs &A, uint64_t i0, uint64_t i1) {
    if (i0 >= i1) return MEMO_OK;
    const bool first = i0 == 0;
    RowScan scan(first ? (b->any ? b->last_start : INT64_MIN) : A.s(i0 - 1),
                 first ? b->last_bucket : (A.s(i0 - 1) >> b->bshift));
    std::vector<BlockResult> res(1);
    for (uint64_t i = i0; i < i1; ++i) {
        int64_t s, len;
        uint32_t a12;
        scan.row(A, i, res[0], s, len, a12);
        b->carry_b[b->carry_n] = dense_b(s, len);
        b->carry_a[b->carry_n] = a12 & 0x1FFu;
        ++b->carry_n;
    }
    scan.finish(res[0]);
    if (scan.top > 511u) res[0].bad |= 16;
    if (res[0].bad) return refuse(b, res[0].bad);
    return merge_results(b, res);
}

int push_dense(memo_builder *b, const PackArgs &A, uint64_t rows) {
    PinnedRing *ring = b->ring;
    // rows that complete the group the last push left open
    const uint64_t head = b->carry_n ? ((uint64_t)(5 - b->carry_n) < rows ? (uint64_t)(5 - b->carry_n) : rows) : 0;
    int rc = carry_rows(b, A, 0, head);
    if (rc) return rc;
    const bool lead = b->carry_n == 5;  // a completed group leads the first chunk
    const uint64_t groups = (rows - head) / 5, tail0 = head + 5 * groups;
    if (groups || lead) {
        if (b->groups_sent + groups + (lead ? 1 : 0) > b->d_groups) return builder_fail(b, MEMO_EINVAL, "more rows than the builder was made for");
        const uint64_t nblocks = groups ? (groups + kBlockGroups - 1) / kBlockGroups : 1;  // (only the leading group: one empty block)
        PushPipe pipe(b, nblocks, kChunkGroups / kBlockGroups);
        const uint64_t pos = lead ? 1 : 0;  // the first chunk's groups sit one group into their slot
        auto pack = [&](uint64_t blk, uint64_t c, char *buf, BlockResult &r) {
            const uint64_t ga = blk * kBlockGroups, gb = ga + kBlockGroups < groups ? ga + kBlockGroups : groups;
            if (ga >= gb) return;
            const uint64_t i0 = head + 5 * ga;
            const bool first = i0 == 0;
            const int64_t prev_start = first ? (b->any ? b->last_start : INT64_MIN) : A.s(i0 - 1);
            const int64_t prev_bucket = first ? b->last_bucket : (A.s(i0 - 1) >> b->bshift);
            uint32_t *out = reinterpret_cast<uint32_t *>(buf) + 4 * ((c == 0 ? pos : 0) + ga - c * kChunkGroups);
            pack_dense(A, i0, gb - ga, prev_start, prev_bucket, out, r);
        };
        uint64_t sent = b->groups_sent;
        auto issue = [&](uint64_t c, char *buf) {
            const uint64_t g0 = c * kChunkGroups, gn = groups - g0 < kChunkGroups ? groups - g0 : kChunkGroups;
            uint64_t send = gn;
            if (c == 0 && lead) {
                dense_group(b->carry_b, b->carry_a, reinterpret_cast<uint32_t *>(buf));
                b->carry_n = 0;
                send += 1;
            }
            const int r = hp::copy_h2d_async(b->d_pk + 4 * sent, buf, send * 16, ring->stream);
            sent += send;
            return r;
        };
        rc = pipe.drive(pack, issue);
        if (rc) return builder_fail(b, rc, "copy to the device failed");
        int bad = 0;
        for (const BlockResult &r : pipe.res) bad |= r.bad;
        if (bad) return refuse(b, bad);
        if ((rc = merge_results(b, pipe.res))) return rc;
        b->groups_sent = sent;
    }
    return carry_rows(b, A, tail0, rows);  // the rows of the last, incomplete group wait for the next push
}

}  // namespace

int builder_fail(memo_builder *b, int code, const char *what) {
    b->failed = code;
    return fail(code, "%s", what);
}

int builder_push_core(memo_builder *b, const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows, int stride) {
    // the bucket table has to reach the last row's bucket before the workers fill it
    const int64_t s_last = start[(rows - 1) * (uint64_t)stride];
    if (s_last < 0 || s_last >= kHostCoordLimit || (b->any && start[0] < b->last_start))
    {
        b->why = 1;
        return builder_fail(b, MEMO_EUNPACKABLE, "rows are unsorted or have a start outside [0, 2^61): not packable");
    }
    const int64_t need = (s_last >> b->bshift) + 3;
    if ((int64_t)b->boff.size() < need) {
        if ((uint64_t)need > ((uint64_t)1 << 34)) return builder_fail(b, MEMO_EUNPACKABLE, "bucket table too large");
        b->boff.resize((size_t)need + (size_t)need / 4);
    }
    PackArgs A{start, end, annot, b->bshift, b->rows, b->boff.data(), (int64_t)b->boff.size(), stride};
    const int rc = b->dense ? push_dense(b, A, rows) : push_words(b, A, rows);
    if (rc) return rc;
    if (!b->any) b->first_start = start[0];
    b->any = true;
    b->last_start = s_last;
    b->last_bucket = s_last >> b->bshift;
    b->rows += rows;
    return MEMO_OK;
}

// dense rows: the last group, when it is incomplete, goes out padded with rows that lie behind the index (row numbers
// >= rows: the sweep masks them by number)
int builder_flush_core(memo_builder *b) {
    if (!b->dense || !b->carry_n) return MEMO_OK;
    PinnedRing *ring = b->ring;
    const int s = ring->next;
    ring->next = (s + 1) % PinnedRing::kSlots;
    char *buf = nullptr;
    int rc = ring->buffer(s, &buf);
    if (!rc) rc = ring->wait(s);
    if (rc) return builder_fail(b, rc, "pinned staging ring failed");
    for (int j = b->carry_n; j < 5; ++j) b->carry_b[j] = b->carry_a[j] = 0;
    dense_group(b->carry_b, b->carry_a, reinterpret_cast<uint32_t *>(buf));
    b->carry_n = 0;
    if (b->groups_sent + 1 > b->d_groups) return builder_fail(b, MEMO_EINVAL, "more rows than the builder was made for");
    if ((rc = hp::copy_h2d_async(b->d_pk + 4 * b->groups_sent, buf, 16, ring->stream)) || (rc = ring->mark(s)))
        return builder_fail(b, rc, "copy to the device failed");
    b->groups_sent += 1;
    return MEMO_OK;
}

}  // namespace memo
