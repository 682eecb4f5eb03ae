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

#include <atomic>
#include <climits>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>

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

HostPool::HostPool() : impl_(new Impl()) {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    unsigned want = hw < 32 ? hw : 32;  // profiles/r02_oneshot_host_threads.txt: 32 threads pack fastest; 64 and up lose a third
    if (const char *v = getenv("MEMO_HOST_THREADS")) {
        const int n = atoi(v);
        if (n > 0) want = (unsigned)n;
    }
    for (unsigned i = 1; i < want; ++i) {
        impl_->workers.emplace_back([this] { impl_->loop(); });
        impl_->workers.back().detach();
    }
}

int HostPool::threads() const { return (int)impl_->workers.size() + 1; }

void HostPool::run(int n, void (*f)(void *, int), void *ctx) {
    if (n <= 0) return;
    Impl &I = *impl_;
    if (n == 1 || I.workers.empty()) {
        for (int i = 0; i < n; ++i) f(ctx, i);
        return;
    }
    std::lock_guard<std::mutex> serial(I.run_mutex);
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
    I.work(f, ctx, n);
    std::unique_lock<std::mutex> lk(I.m);
    I.cv_done.wait(lk, [&] { return I.busy == 0; });
    I.job = nullptr;
}

// ------------------------------------------------------------------------------------------
// pinned staging ring.  Rings are cached per device and handed out to one user at a time; a second concurrent
// user on the same device gets a ring of its own.  A slot's pinned buffer is allocated when the slot is first
// used (hipHostMalloc of 24 MiB costs milliseconds: a call that moves one small piece pays for one slot, a call
// that moves none -- a cache hit of `memo query` on a small window -- for nothing).
// ------------------------------------------------------------------------------------------
int PinnedRing::buffer(int s, char **out) {
    if (!slot[s]) {
        void *p = nullptr;
        int rc = hp::pinned_alloc(&p, kSlotBytes);
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
constexpr uint64_t kBlockRows = 1 << 14;                      // rows per worker task (256 tasks per pinned slot: an even share for 32 workers)
constexpr uint64_t kBlockGroups = 3264;                       // dense: groups per worker task (16 320 rows = 51 pieces)
constexpr uint64_t kChunkGroups = (uint64_t)1 << 20;          // dense: groups per pinned slot (16 MiB; one more may lead them)

struct PackArgs {
    const int64_t *start, *end, *annot;
    int shift;
    uint64_t global0;   // global row number of local row 0
    int64_t *boff;
    int64_t boff_size;
};

// One row: the checks of memo_index_finalize, the bucket table, the rows with end < start.  Returns the row's fields
// through s / len / a12.  `ps` / `pb`: start and bucket of the row before.
struct RowScan {
    uint64_t top = 0;
    int bad = 0, wide = 0;
    int64_t ps, pb;
    RowScan(int64_t prev_start, int64_t prev_bucket) : ps(prev_start), pb(prev_bucket) {}
    inline void row(const PackArgs &A, uint64_t i, BlockResult &res, int64_t &s_out, int64_t &len_out, uint32_t &a12_out) {
        const int64_t s = A.start[i], e = A.end[i], a = A.annot[i];
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
            if (bk > pb && !bad && bk < A.boff_size)
                for (int64_t q = pb + 1; q <= bk; ++q) A.boff[q] = (int64_t)(A.global0 + i);
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
#ifdef MEMO_HOST_TEST_KNOBS  // (tests/test_host_sanitizers.py builds this file with it: both instances run under the sanitizers)
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
        scan(A.start + i, A.end + i, A.annot + i, n, ps, A.shift, W, A8, mark, f);
        bad |= f.unsorted ? 1 : 0;
        bad |= (f.sign >> 63) ? 2 : 0;
        bad |= f.annot_or > 4095u ? 4 : 0;
        bad |= f.coord ? 8 : 0;
        for (int j = 0; j < n; ++j) top = A8[j] > top ? A8[j] : top;
        if (f.any_bucket | f.any_long) {
            for (int j = 0; j < n; ++j) {
                if (!mark[j]) continue;
                const int64_t s = A.start[i + (uint64_t)j];
                if (mark[j] & 2) {
                    res.long_rows.push_back(s);
                    res.long_rows.push_back(A.end[i + (uint64_t)j]);
                    res.long_rows.push_back(A.annot[i + (uint64_t)j]);
                }
                const int64_t bk = s >> A.shift;
                if (bk != pb) {  // first row of its bucket(s): boff[b] = lower_bound(start, b << shift)
                    // (`bad` as RowScan has it at this row: set by any earlier piece, or by this one -- a piece that is
                    // bad anywhere fails the builder, what it wrote to the table is never read)
                    if (bk > pb && !bad && bk < A.boff_size)
                        for (int64_t q = pb + 1; q <= bk; ++q) A.boff[q] = (int64_t)(A.global0 + i + (uint64_t)j);
                    pb = bk;
                }
            }
        }
        ps = A.start[i + (uint64_t)n - 1];
        emit(i, n, W, A8);
    }
    res.max_annot = top;
    res.bad = bad;
    res.wide_annot = f.annot_or > 255u ? 1 : 0;
    res.over511 = f.annot_or > 511u ? 1 : 0;
}

// rows [i0, i1) -> words (format 4 or 12), pk[i] for row i.  end < start (handled by long_rows_*_kernel) packs as
// "never writes", like len >= 255.
void pack_words(const PackArgs &A, uint64_t i0, uint64_t i1, int64_t prev_start, int64_t prev_bucket, uint32_t *pk,
                int fmt, BlockResult &res) {
    auto emit = [&](uint64_t at, int n, const uint32_t *W, const uint32_t *) { memcpy(pk + at, W, (size_t)n * 4); };
    if (fmt == 12)
        scan_block<12>(A, i0, i1, prev_start, prev_bucket, res, emit);
    else
        scan_block<4>(A, i0, i1, prev_start, prev_bucket, res, emit);
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

// rows [i0, i0 + 5 * groups) -> groups at out (4 dwords each)
void pack_dense(const PackArgs &A, uint64_t i0, uint64_t groups, int64_t prev_start, int64_t prev_bucket, uint32_t *out,
                BlockResult &res) {
    static_assert(kPiece % 5 == 0, "a piece is whole groups");
    scan_block<3>(A, i0, i0 + 5 * groups, prev_start, prev_bucket, res, [&](uint64_t at, int n, const uint32_t *W, const uint32_t *A8) {
        uint32_t *o = out + 4 * ((at - i0) / 5);
        for (int j = 0; j + 5 <= n; j += 5, o += 4) dense_group(W + j, A8 + j, o);
    });
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

int push_words(memo_builder *b, const PackArgs &A, uint64_t rows) {
    HostPool &pool = HostPool::get();
    PinnedRing *ring = b->ring;
    const int64_t *start = A.start;
    for (uint64_t c0 = 0; c0 < rows; c0 += kChunkRows) {
        const uint64_t cn = rows - c0 < kChunkRows ? rows - c0 : kChunkRows;
        const int s = ring->next;
        ring->next = (s + 1) % PinnedRing::kSlots;
        char *buf = nullptr;
        int rc = ring->buffer(s, &buf);
        if (!rc) rc = ring->wait(s);
        if (rc) return builder_fail(b, rc, "pinned staging ring failed");
        uint32_t *pk = reinterpret_cast<uint32_t *>(buf);
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
                pack_words(A, i0, i1, prev_start, prev_bucket, pk - c0, fmt, res[(size_t)t]);
            });
            int bad = 0, wide = 0;
            for (const BlockResult &r : res) {
                bad |= r.bad;
                wide |= r.wide_annot;
            }
            if (bad) { b->why = bad; return builder_fail(b, MEMO_EUNPACKABLE, bad_message(bad)); }
            if (wide && fmt == 4) {  // switch the index to 12-bit annots: rewrite what is on the device, redo this chunk
                if ((rc = hp::stream_sync(ring->stream))) return builder_fail(b, rc, "copy stream failed");
                if (b->rows + c0 && (rc = hp::widen_annots(b->d_pk, b->rows + c0, ring->stream)))
                    return builder_fail(b, rc, "rewriting the rows with 12-bit annots failed");
                b->fmt = 12;
                continue;
            }
            break;
        }
        if ((rc = merge_results(b, res))) return rc;
        if ((rc = hp::copy_h2d_async(b->d_pk + b->rows + c0, pk, cn * 4, ring->stream)) || (rc = ring->mark(s)))
            return builder_fail(b, rc, "copy to the device failed");
    }
    return MEMO_OK;
}

// serial rows [i0, i1) of a push into the carried group (dense rows)
int carry_rows(memo_builder *b, const PackArgs &A, uint64_t i0, uint64_t i1) {
    if (i0 >= i1) return MEMO_OK;
    const bool first = i0 == 0;
    RowScan scan(first ? (b->any ? b->last_start : INT64_MIN) : A.start[i0 - 1],
                 first ? b->last_bucket : (A.start[i0 - 1] >> b->bshift));
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
    if (res[0].bad) { b->why = res[0].bad; return builder_fail(b, MEMO_EUNPACKABLE, bad_message(res[0].bad)); }
    return merge_results(b, res);
}

int push_dense(memo_builder *b, const PackArgs &A, uint64_t rows) {
    HostPool &pool = HostPool::get();
    PinnedRing *ring = b->ring;
    const int64_t *start = A.start;
    // rows that complete the group the last push left open
    const uint64_t head = b->carry_n ? ((uint64_t)(5 - b->carry_n) < rows ? (uint64_t)(5 - b->carry_n) : rows) : 0;
    int rc = carry_rows(b, A, 0, head);
    if (rc) return rc;
    bool lead = b->carry_n == 5;  // a completed group waits to lead the next chunk
    const uint64_t groups = (rows - head) / 5, tail0 = head + 5 * groups;
    for (uint64_t g0 = 0; g0 < groups || lead; g0 += kChunkGroups) {
        const uint64_t gn = groups - g0 < kChunkGroups ? groups - g0 : kChunkGroups;
        const int s = ring->next;
        ring->next = (s + 1) % PinnedRing::kSlots;
        char *buf = nullptr;
        rc = ring->buffer(s, &buf);
        if (!rc) rc = ring->wait(s);
        if (rc) return builder_fail(b, rc, "pinned staging ring failed");
        uint32_t *out = reinterpret_cast<uint32_t *>(buf);
        uint64_t pos = 0;
        if (lead) {
            dense_group(b->carry_b, b->carry_a, out);
            b->carry_n = 0;
            lead = false;
            pos = 1;
        }
        const int tasks = (int)((gn + kBlockGroups - 1) / kBlockGroups);
        std::vector<BlockResult> res((size_t)tasks);
        pool.run(tasks, [&](int t) {
            const uint64_t ga = g0 + (uint64_t)t * kBlockGroups;
            const uint64_t gb = ga + kBlockGroups < g0 + gn ? ga + kBlockGroups : g0 + gn;
            const uint64_t i0 = head + 5 * ga;
            const bool first = i0 == 0;
            const int64_t prev_start = first ? (b->any ? b->last_start : INT64_MIN) : start[i0 - 1];
            const int64_t prev_bucket = first ? b->last_bucket : (start[i0 - 1] >> b->bshift);
            pack_dense(A, i0, gb - ga, prev_start, prev_bucket, out + 4 * (pos + ga - g0), res[(size_t)t]);
        });
        int bad = 0;
        for (const BlockResult &r : res) bad |= r.bad;
        if (bad) { b->why = bad; return builder_fail(b, MEMO_EUNPACKABLE, bad_message(bad)); }
        if ((rc = merge_results(b, res))) return rc;
        const uint64_t send = pos + gn;
        if (b->groups_sent + send > b->d_groups) return builder_fail(b, MEMO_EINVAL, "more rows than the builder was made for");
        if ((rc = hp::copy_h2d_async(b->d_pk + 4 * b->groups_sent, out, send * 16, ring->stream)) || (rc = ring->mark(s)))
            return builder_fail(b, rc, "copy to the device failed");
        b->groups_sent += send;
        if (gn == 0) break;  // (only the leading group went)
    }
    return carry_rows(b, A, tail0, rows);  // the rows of the last, incomplete group wait for the next push
}

}  // namespace

int builder_fail(memo_builder *b, int code, const char *what) {
    b->failed = code;
    return fail(code, "%s", what);
}

int builder_push_core(memo_builder *b, const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows) {
    // the bucket table has to reach the last row's bucket before the workers fill it
    const int64_t s_last = start[rows - 1];
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
    PackArgs A{start, end, annot, b->bshift, b->rows, b->boff.data(), (int64_t)b->boff.size()};
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
