// tests/host_stub.cpp -- the host core of the packed way in (memo_amd/csrc/memo_hostcore.cpp: worker pool, pinned
// ring, row packers, the builder's push loop) on a machine without a GPU: the device seam memo::hp is implemented
// here with malloc and memcpy ("device memory" is host memory; asynchronous copies run on a copier thread per stream,
// so that ring slots really are in flight while the workers pack the next one), and main() drives it the way the
// library does -- two builders on two threads at once, ragged pieces, both row formats, the late switch to 12-bit
// annots, refusals -- comparing every packed row with a restatement of the format written here.
// Built and run by tests/test_host_sanitizers.py under -fsanitize=thread and -fsanitize=address,undefined.
// Test infrastructure: not linked into the product.
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../memo_amd/csrc/memo_hostcore.h"

namespace memo {

static thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace hp {
// a "stream": a thread that runs queued copies in order; an "event": a flag set when the stream reaches it
struct Stream {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false, busy = false;
    std::thread th;
    Stream() : th([this] { loop(); }) {}
    ~Stream() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
    void loop() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
                busy = true;
            }
            f();
            {
                std::lock_guard<std::mutex> lk(m);
                busy = false;
            }
            cv.notify_all();
        }
    }
    void push(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> lk(m);
            q.push_back(std::move(f));
        }
        cv.notify_all();
    }
    void sync() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return q.empty() && !busy; });
    }
};
struct Event {
    std::mutex m;
    std::condition_variable cv;
    uint64_t recorded = 0, reached = 0;
};
static Stream g_null_stream;

int set_device(int, int *previous) {
    if (previous) *previous = -1;
    return MEMO_OK;
}
int pinned_alloc(void **p, size_t bytes) {
    *p = malloc(bytes);
    return *p ? MEMO_OK : fail(MEMO_EHIP, "out of memory");
}
void pinned_free(void *p) { free(p); }
int stream_create(void **stream) {
    *stream = new Stream();
    return MEMO_OK;
}
void stream_destroy(void *stream) { delete static_cast<Stream *>(stream); }
static Stream *S(void *s) { return s ? static_cast<Stream *>(s) : &g_null_stream; }
int stream_sync(void *stream) {
    S(stream)->sync();
    return MEMO_OK;
}
int event_create(void **event) {
    *event = new Event();
    return MEMO_OK;
}
void event_destroy(void *event) { delete static_cast<Event *>(event); }
int event_record(void *event, void *stream) {
    Event *e = static_cast<Event *>(event);
    uint64_t id;
    {
        std::lock_guard<std::mutex> lk(e->m);
        id = ++e->recorded;
    }
    S(stream)->push([e, id] {
        {
            std::lock_guard<std::mutex> lk(e->m);
            e->reached = id;
        }
        e->cv.notify_all();
    });
    return MEMO_OK;
}
int event_sync(void *event) {
    Event *e = static_cast<Event *>(event);
    std::unique_lock<std::mutex> lk(e->m);
    const uint64_t want = e->recorded;
    e->cv.wait(lk, [&] { return e->reached >= want; });
    return MEMO_OK;
}
int event_query(void *event, int *done) {
    Event *e = static_cast<Event *>(event);
    std::lock_guard<std::mutex> lk(e->m);
    *done = e->reached >= e->recorded;
    return MEMO_OK;
}
int copy_h2d_async(void *dev, const void *host, size_t bytes, void *stream) {
    S(stream)->push([=] { memcpy(dev, host, bytes); });
    return MEMO_OK;
}
int copy_d2h_async(void *host, const void *dev, size_t bytes, void *stream) {
    S(stream)->push([=] { memcpy(host, dev, bytes); });
    return MEMO_OK;
}
int copy_h2d(void *dev, const void *host, size_t bytes) {
    memcpy(dev, host, bytes);
    return MEMO_OK;
}
int widen_annots(uint32_t *pk, uint64_t rows, void *stream) {
    S(stream)->push([=] {
        for (uint64_t i = 0; i < rows; ++i) {
            const uint32_t w = pk[i];
            pk[i] = ((w >> 16) & 0xFFu) | ((w & 0xFFFu) << 8) | ((w >> 24) << 20);
        }
    });
    return MEMO_OK;
}
}  // namespace hp
}  // namespace memo

using namespace memo;

#define CHECK(cond)                                                                    \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            fprintf(stderr, "%s:%d: CHECK failed: %s (%s)\n", __FILE__, __LINE__, #cond, g_err.c_str()); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

struct Rows {
    std::vector<int64_t> s, e, a;
};

static Rows make_rows(uint64_t n, uint64_t seed, int max_annot, int late_wide_at = -1, int long_every = 0) {
    std::mt19937_64 rng(seed);
    Rows r;
    r.s.resize(n), r.e.resize(n), r.a.resize(n);
    int64_t pos = (int64_t)(rng() % 1000);
    for (uint64_t i = 0; i < n; ++i) {
        pos += (int64_t)(rng() % 3 == 0);
        r.s[i] = pos;
        r.e[i] = pos + (int64_t)(rng() % 300);
        if (long_every && i % (uint64_t)long_every == 7) r.e[i] = pos - 1 - (int64_t)(rng() % 1000);
        r.a[i] = 1 + (int64_t)(rng() % (uint64_t)max_annot);
        if (late_wide_at >= 0 && i >= (uint64_t)late_wide_at && rng() % 50 == 0) r.a[i] = 256 + (int64_t)(rng() % 3000);
    }
    return r;
}

static memo_builder *new_builder(uint64_t cap, bool dense) {
    memo_builder *b = new memo_builder();
    b->cap = cap;
    b->dense = dense;
    b->padded = ((cap + 15) & ~(uint64_t)15) + 4096;
    b->d_groups = dense ? (b->padded + 4) / 5 + 64 : 0;
    const size_t bytes = dense ? (size_t)b->d_groups * 16 : (size_t)b->padded * 4;
    b->d_pk = static_cast<uint32_t *>(calloc(bytes, 1));
    CHECK(acquire_ring(0, &b->ring) == MEMO_OK);
    return b;
}

static void free_builder(memo_builder *b) {
    release_ring(b->ring);
    free(b->d_pk);
    delete b;
}

// the format restated: what row i of the index must look like on the "device"
static void verify(const memo_builder *b, const Rows &r, uint64_t n) {
    std::vector<int64_t> boff;
    const int shift = b->bshift;
    uint64_t n_long = 0, top = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const int64_t s = r.s[i], e = r.e[i], a = r.a[i];
        const uint64_t len = e < s ? ~0ull : (uint64_t)(e - s);
        n_long += e < s;
        top = (uint64_t)a > top ? (uint64_t)a : top;
        if (b->dense) {
            const uint32_t *g = b->d_pk + 4 * (i / 5);
            const int j = (int)(i % 5);
            const uint32_t B = (((uint32_t)s & 1023u) << 6) | (len > 63 ? 63u : (uint32_t)len);
            uint32_t gotB, gotA;
            if (j < 4) {
                gotB = g[j] & 0xFFFFu;
                gotA = g[j] >> 24;
            } else {
                gotB = ((g[0] >> 16) & 0xFFu) | (((g[1] >> 16) & 0xFFu) << 8);
                gotA = (g[2] >> 16) & 0xFFu;
            }
            gotA |= ((g[3] >> (16 + j)) & 1u) << 8;  // the ninth annot bit: bit 16 + j of the group's last dword
            CHECK(gotB == B && gotA == (uint32_t)a);
            CHECK(((g[3] >> 21) & 7u) == 0);  // (the byte's other three bits stay clear)
        } else {
            const uint32_t l8 = len > 255 ? 255u : (uint32_t)len;
            const uint32_t want = b->fmt == 12 ? l8 | (((uint32_t)s & 0xFFFu) << 8) | ((uint32_t)a << 20)
                                               : ((uint32_t)s & 0xFFFFu) | (l8 << 16) | ((uint32_t)a << 24);
            CHECK(b->d_pk[i] == want);
        }
        // bucket table: boff[q] = first row with start >= q << shift
        const int64_t bk = s >> shift;
        while ((int64_t)boff.size() <= bk) boff.push_back((int64_t)i);
    }
    CHECK(b->rows == n && b->long_rows.size() == 3 * n_long && b->max_annot == top);
    for (size_t q = 0; q < boff.size(); ++q) CHECK(b->boff[q] == boff[q]);
}

// the rows as filter_pq hands them over: [M, 3] row-major (stride 3 of builder_push_core)
static std::vector<int64_t> interleaved(const Rows &r) {
    std::vector<int64_t> v(3 * r.s.size());
    for (size_t i = 0; i < r.s.size(); ++i) v[3 * i] = r.s[i], v[3 * i + 1] = r.e[i], v[3 * i + 2] = r.a[i];
    return v;
}

static void run_builder(bool dense, uint64_t n, uint64_t seed, int late_wide_at, int long_every, bool as_rows = false) {
    Rows r = make_rows(n, seed, dense ? (seed % 2 ? 511 : 255) : 200, dense ? -1 : late_wide_at, long_every);   // (dense rows: annots of up to nine bits)
    const std::vector<int64_t> rows = as_rows ? interleaved(r) : std::vector<int64_t>();
    memo_builder *b = new_builder(n, dense);
    std::mt19937_64 rng(seed ^ 0x55);
    uint64_t at = 0;
    while (at < n) {  // ragged pieces: 1 .. 300 000 rows, often not a multiple of five
        uint64_t piece = 1 + rng() % (rng() % 4 == 0 ? 300000 : 23);
        if (piece > n - at) piece = n - at;
        if (as_rows)
            CHECK(builder_push_core(b, rows.data() + 3 * at, rows.data() + 3 * at + 1, rows.data() + 3 * at + 2, piece, 3) == MEMO_OK);
        else
            CHECK(builder_push_core(b, r.s.data() + at, r.e.data() + at, r.a.data() + at, piece) == MEMO_OK);
        at += piece;
    }
    CHECK(builder_flush_core(b) == MEMO_OK);
    CHECK(b->ring->drain() == MEMO_OK);
    if (!dense && late_wide_at >= 0 && (uint64_t)late_wide_at < n) CHECK(b->fmt == 12);
    if (dense) CHECK(b->groups_sent == (n + 4) / 5 && b->carry_n == 0);
    verify(b, r, n);
    free_builder(b);
}

int main() {
    // two builders at a time, on two threads (they share the worker pool and the ring cache)
    for (int round = 0; round < 3; ++round) {
        std::thread t1([&] { run_builder(false, 700001 + 13 * round, 1 + round, round == 1 ? 400000 : -1, round == 2 ? 1000 : 0); });
        std::thread t2([&] { run_builder(true, 612347 + 7 * round, 11 + round, -1, round == 2 ? 777 : 0); });
        t1.join();
        t2.join();
    }
    // the same from ROWS ([M, 3] row-major, what filter_pq returns): both formats, the late switch to 12-bit annots, rows with end < start
    run_builder(false, 700014, 2, 400000, 0, true);
    run_builder(false, 500009, 3, -1, 1000, true);
    run_builder(true, 612361, 13, -1, 777, true);
    run_builder(true, 612354, 12, -1, 0, true);
    // one push of many chunks: more chunks than the ring has slots, so slots are reused while workers pack ahead
    // (dense: 1 047 744 groups = 5.2 M rows per chunk; words: 6.3 M rows), then the same rows with a row out of order
    // deep inside -- every worker has to stop, the caller has to come back
    {
        const uint64_t n = 27000007;
        Rows r;
        r.s.resize(n), r.e.resize(n), r.a.resize(n);
        uint64_t x = 0x9E3779B97F4A7C15ull;
        int64_t pos = 17;
        for (uint64_t i = 0; i < n; ++i) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            pos += (int64_t)((x >> 33) % 3 == 0);
            r.s[i] = pos;
            r.e[i] = i % 100003 == 5 ? pos - 2 : pos + (int64_t)((x >> 40) % 90);
            r.a[i] = 1 + (int64_t)((x >> 50) % 300);
        }
        for (int dense = 0; dense < 2; ++dense) {
            memo_builder *b = new_builder(n, dense);
            CHECK(builder_push_core(b, r.s.data(), r.e.data(), r.a.data(), 3) == MEMO_OK);   // (a ragged start: carried rows, a leading group)
            if (dense) {  // (the rest as ROWS: the many-chunk push with the interleaved loads)
                const std::vector<int64_t> rows = interleaved(r);
                CHECK(builder_push_core(b, rows.data() + 9, rows.data() + 10, rows.data() + 11, n - 3, 3) == MEMO_OK);
            } else
            CHECK(builder_push_core(b, r.s.data() + 3, r.e.data() + 3, r.a.data() + 3, n - 3) == MEMO_OK);
            CHECK(builder_flush_core(b) == MEMO_OK);
            CHECK(b->ring->drain() == MEMO_OK);
            if (!dense) CHECK(b->fmt == 12);
            verify(b, r, n);
            free_builder(b);
        }
        r.s[n - 1000] = 3;
        for (int dense = 0; dense < 2; ++dense) {
            memo_builder *b = new_builder(n, dense);
            CHECK(builder_push_core(b, r.s.data(), r.e.data(), r.a.data(), n) == MEMO_EUNPACKABLE && b->failed == MEMO_EUNPACKABLE && (b->why & 1));
            free_builder(b);
        }
    }
    // one annot > 255 that the sample of the annots misses: the push finds it while packing and starts over in format 12,
    // the rows of the push before it are rewritten on the "device"
    {
        Rows r = make_rows(300000, 9, 200);
        r.a[277777] = 300;
        memo_builder *b = new_builder(300000, false);
        CHECK(builder_push_core(b, r.s.data(), r.e.data(), r.a.data(), 100000) == MEMO_OK && b->fmt == 4);
        CHECK(builder_push_core(b, r.s.data() + 100000, r.e.data() + 100000, r.a.data() + 100000, 200000) == MEMO_OK && b->fmt == 12);
        CHECK(b->ring->drain() == MEMO_OK);
        verify(b, r, 300000);
        free_builder(b);
    }
    run_builder(true, 4, 5, -1, 0);      // less than one group
    run_builder(true, 5, 6, -1, 0);
    run_builder(false, 1, 7, -1, 0);
    // refusals mark the builder failed: unsorted rows, a negative start, an annot the format cannot hold
    {
        Rows r = make_rows(1000, 3, 100);
        std::swap(r.s[500], r.s[10]);
        for (int dense = 0; dense < 2; ++dense) {
            memo_builder *b = new_builder(1000, dense);
            CHECK(builder_push_core(b, r.s.data(), r.e.data(), r.a.data(), 1000) == MEMO_EUNPACKABLE && b->failed == MEMO_EUNPACKABLE);
            free_builder(b);
        }
        Rows w = make_rows(1000, 4, 100);
        w.a[999] = 512;
        memo_builder *b = new_builder(1000, true);
        CHECK(builder_push_core(b, w.s.data(), w.e.data(), w.a.data(), 1000) == MEMO_EUNPACKABLE);
        free_builder(b);
        w.a[999] = 5000;
        b = new_builder(1000, false);
        CHECK(builder_push_core(b, w.s.data(), w.e.data(), w.a.data(), 1000) == MEMO_EUNPACKABLE);
        free_builder(b);
    }
    // the pipelined transfers, both directions, sizes around the piece boundaries
    for (size_t bytes : {(size_t)0, (size_t)5, (size_t)3 << 20, ((size_t)24 << 20) + 17, ((size_t)80 << 20) + 1}) {
        std::vector<unsigned char> src(bytes), dev(bytes + 1, 0xEE), back(bytes);
        for (size_t i = 0; i < bytes; ++i) src[i] = (unsigned char)(i * 131 + (i >> 12));
        setenv("MEMO_COLD_RING_MB", "0", 1);
        CHECK(upload_pipelined_core(0, dev.data(), src.data(), bytes) == MEMO_OK);
        CHECK(dev[bytes] == 0xEE && (!bytes || memcmp(dev.data(), src.data(), bytes) == 0));
        CHECK(download_pipelined_core(0, back.data(), dev.data(), bytes) == MEMO_OK);
        CHECK(!bytes || memcmp(back.data(), src.data(), bytes) == 0);
    }
    printf("hostcore ok (%d pool threads)\n", HostPool::get().threads());
    return 0;
}
