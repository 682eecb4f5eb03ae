// memo_multi.hip -- several GPUs from ONE process, behind the C ABI (the proposed
// `devices, n_devices` arguments).  The reference is single-process and single-threaded; what shards is
// the query window: position p depends only on rows with p < start <= p + k - 1, so [qs, qe) is cut
// into contiguous sub-windows, sub-window [a, b) needs exactly the rows a < start < b + k -- the
// reference's own filter (/root/reference/src/memo_query.py:25-27 with :100) applied to the sub-window
// -- and every GPU runs the single-GPU sweep unchanged.  No exchange during the sweep; the result
// slices are disjoint and go straight to where the caller wants them:
//   * host form (memo_conservation_multi / memo_membership_multi): each device's slice comes back
//     over its own PCIe link into its part of the caller's buffer -- no GPU-to-GPU traffic at all;
//   * resident form (memo_query_*_multi_dev): each device sweeps into a slice buffer of its own and
//     hipMemcpyPeerAsync delivers it into the root device's result (xGMI: every peer has its own link
//     to the root, the copies run concurrently); the root's stream waits for all of them.
// memo_split_window() is the one partition rule; memo_amd/shard.py (one process per GPU over RCCL)
// calls the same function.
#include <string>
#include <thread>
#include <vector>

#include "memo_common.h"

using namespace memo;

extern "C" {

int memo_split_window(int64_t qs, int64_t qe, int32_t parts, int32_t align, double first_weight, int64_t *cuts) {
    if (parts < 1 || !cuts) return fail(MEMO_EINVAL, "bad partition arguments");
    if (align < 1) align = 1;
    if (!(first_weight >= 0.0) || first_weight > 1e6) return fail(MEMO_EINVAL, "first_weight must be in [0, 1e6]");
    const int64_t L = qe > qs ? qe - qs : 0;
    // part 0 takes first_weight shares, every other part one share; lengths are multiples of `align`
    // (rounded up), the tail takes what is left, parts past the end are empty
    const double shares = first_weight + (double)(parts - 1);
    auto round_up = [&](double x) {
        int64_t v = (int64_t)x;
        if ((double)v < x) ++v;
        return (v + align - 1) / align * align;
    };
    const int64_t per = shares > 0 ? round_up((double)L / shares) : 0;
    const int64_t first = parts == 1 ? L : (shares > 0 ? round_up((double)L * first_weight / shares) : 0);
    int64_t at = qs;
    cuts[0] = qs;
    for (int32_t g = 0; g < parts; ++g) {
        const int64_t len = g == 0 ? first : per;
        at = at + len < qs + L ? at + len : qs + L;
        cuts[g + 1] = at;
    }
    cuts[parts] = qs + L;  // rounding never loses the tail
    for (int32_t g = parts; g > 0; --g)
        if (cuts[g - 1] > cuts[g]) cuts[g - 1] = cuts[g];
    return MEMO_OK;
}

}  // extern "C"

namespace {

// first index i in [0, n) with v[i] >= key (v sorted)
uint64_t lower_bound64(const int64_t *v, uint64_t n, int64_t key) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (v[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// rows start-sorted, and none with end < start?  Such a row (legal input to memo_query.py, never written by the
// reference's index builder) shades [end - (k-1), start), any distance left of its start: the rule "sub-window
// [a, b) needs the rows a < start < b + k" does not hold for it, so the split is not taken.
bool splittable(const int64_t *s, const int64_t *e, uint64_t n) {
    if (n < 1) return true;
    const unsigned nt = n > (1u << 22) ? 16u : 1u;
    std::vector<int> bad(nt, 0);
    std::vector<std::thread> th;
    const uint64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([=, &bad] {
            const uint64_t b = (uint64_t)t * per, en = b + per < n ? b + per : n;
            int x = 0;
            for (uint64_t i = b; i < en; ++i) x |= (i > 0 && s[i - 1] > s[i]) | (e[i] < s[i]);
            bad[t] = x;
        });
    for (auto &x : th) x.join();
    for (int x : bad)
        if (x) return false;
    return true;
}

int host_multi(const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows, int64_t qs,
               int64_t qe, int32_t k, int32_t num_docs, void *out, const int32_t *devices, int32_t n_devices,
               bool membership) {
    if (n_devices < 1 || !devices) return fail(MEMO_EINVAL, "no devices given");
    if (n_devices > 64) return fail(MEMO_EINVAL, "at most 64 devices");
    if (rows && (!start || !end || !annot)) return fail(MEMO_EINVAL, "column pointer is NULL");
    auto single = [&](int32_t dev) {
        return membership ? memo_membership(start, end, annot, rows, qs, qe, k, num_docs, (uint32_t *)out, dev)
                          : memo_conservation(start, end, annot, rows, qs, qe, k, num_docs, (uint16_t *)out, dev);
    };
    // one device, nothing to cut, rows the binary searches below cannot be trusted on, or rows with end < start
    // (which reach further than k - 1 positions): the single-GPU path (which sorts on the device, applies such rows
    // across the whole window and reproduces the reference's errors)
    if (n_devices == 1 || qe - qs < 8 * (int64_t)n_devices || k >= (1 << 30) || k <= -(1 << 30) || num_docs < 1 ||
        !splittable(start, end, rows))
        return single(devices[0]);
    std::vector<int64_t> cuts((size_t)n_devices + 1);
    int rc = memo_split_window(qs, qe, n_devices, 8, 1.0, cuts.data());
    if (rc) return rc;
    const size_t stride = membership ? (size_t)((num_docs + 31) / 32) * 4 : 2;  // bytes per position
    std::vector<int> codes((size_t)n_devices, MEMO_OK);
    std::vector<std::string> messages((size_t)n_devices);
    std::vector<std::thread> th;
    for (int32_t g = 0; g < n_devices; ++g)
        th.emplace_back([&, g] {
            const int64_t a = cuts[(size_t)g], b = cuts[(size_t)g + 1];
            if (b <= a) return;
            // rows this sub-window sees: a < start < b + k
            const int64_t hi_key = k > 0 ? b + k : b;
            const uint64_t i0 = lower_bound64(start, rows, a + 1), i1 = lower_bound64(start, rows, hi_key);
            const uint64_t n = i1 > i0 ? i1 - i0 : 0;
            char *dst = static_cast<char *>(out) + (size_t)(a - qs) * stride;
            const int r = membership
                              ? memo_membership(start + i0, end + i0, annot + i0, n, a, b, k, num_docs, (uint32_t *)dst, devices[g])
                              : memo_conservation(start + i0, end + i0, annot + i0, n, a, b, k, num_docs, (uint16_t *)dst, devices[g]);
            if (r) {
                codes[(size_t)g] = r;
                messages[(size_t)g] = memo_last_error();  // thread-local: carry it to the caller's thread
            }
        });
    for (auto &x : th) x.join();
    for (int32_t g = 0; g < n_devices; ++g)
        if (codes[(size_t)g]) return fail(codes[(size_t)g], "%s", messages[(size_t)g].c_str());
    return MEMO_OK;
}

// per-(thread, device) slice buffers and streams of the resident form, grown on demand and kept
struct PeerLane {
    int device = -1;
    void *buf = nullptr;
    size_t cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
};

int dev_multi(memo_index_t *const *shards, int32_t n, int64_t qs, int64_t qe, int32_t k, int32_t num_docs,
              void *d_out, int32_t root_device, void *root_stream, double root_weight, bool membership) {
    if (n < 1 || !shards) return fail(MEMO_EINVAL, "no shards given");
    if (n > 64) return fail(MEMO_EINVAL, "at most 64 shards");
    for (int32_t g = 0; g < n; ++g)
        if (!shards[g]) return fail(MEMO_EINVAL, "shard %d is NULL", g);
    if (qe < qs) return fail(MEMO_EINVAL, "ValueError: negative dimensions are not allowed (window end < start)");
    if (qe == qs) return MEMO_OK;
    if (!d_out || ((uintptr_t)d_out & 15)) return fail(MEMO_EINVAL, "output must be a 16-byte aligned device pointer");
    std::vector<int64_t> cuts((size_t)n + 1);
    int rc = memo_split_window(qs, qe, n, 8, root_weight, cuts.data());
    if (rc) return rc;
    const size_t stride = membership ? (size_t)((num_docs + 31) / 32) * 4 : 2;
    thread_local std::vector<PeerLane> lanes;
    if (lanes.size() < (size_t)n) lanes.resize((size_t)n);
    hipStream_t rs = static_cast<hipStream_t>(root_stream);
    // whatever the root's stream still has to do with d_out (readers of the previous result) comes first:
    // the peers' copies into it wait for this point of root_stream
    thread_local hipEvent_t root_ready[64] = {};
    hipEvent_t ready = nullptr;
    if (root_device >= 0 && root_device < 64) {
        DeviceGuard root(root_device);
        if (!root.ok) return fail(MEMO_EHIP, "cannot select HIP device %d", root_device);
        if (!root_ready[root_device]) HIP_TRY(hipEventCreateWithFlags(&root_ready[root_device], hipEventDisableTiming));
        ready = root_ready[root_device];
        HIP_TRY(hipEventRecord(ready, rs));
    } else {
        return fail(MEMO_EINVAL, "root device %d out of range", root_device);
    }
    for (int32_t g = 0; g < n; ++g) {
        const int64_t a = cuts[(size_t)g], b = cuts[(size_t)g + 1];
        if (b <= a) continue;
        const int dev = shards[g]->device;
        char *dst = static_cast<char *>(d_out) + (size_t)(a - qs) * stride;
        // rows with end < start pass the reference's filter on the WHOLE window (they can reach any distance left of
        // their start, memo_common.h: whole_set); every shard holds the ones its sub-window can see as long as it
        // holds the rows of the whole window with end < start (a replica does)
        struct WholeWindow {
            memo_index_t *ix;
            WholeWindow(memo_index_t *i, int64_t s, int64_t e) : ix(i) { ix->whole_qs = s; ix->whole_qe = e; ix->whole_set = 1; }
            ~WholeWindow() { ix->whole_set = 0; }
        } whole(shards[g], qs, qe);
        if (g == 0 && dev == root_device) {  // the root's own slice is swept straight into the result, on the root's stream
            rc = membership ? memo_query_membership_dev(shards[g], a, b, k, num_docs, (uint32_t *)dst, root_stream)
                            : memo_query_conservation_dev(shards[g], a, b, k, num_docs, (uint16_t *)dst, root_stream);
            if (rc) return rc;
            continue;
        }
        DeviceGuard guard(dev);
        if (!guard.ok) return fail(MEMO_EHIP, "cannot select HIP device %d", dev);
        PeerLane &ln = lanes[(size_t)g];
        const size_t bytes = (size_t)(b - a) * stride;
        if (ln.device != dev || ln.cap < bytes) {
            if (ln.buf) {
                DeviceGuard old(ln.device);
                (void)hipFree(ln.buf);
                ln.buf = nullptr;
            }
            if (!ln.stream || ln.device != dev) {
                if (ln.stream) {
                    DeviceGuard old(ln.device);
                    (void)hipStreamDestroy(ln.stream);
                    (void)hipEventDestroy(ln.done);
                }
                HIP_TRY(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
            }
            HIP_TRY(hipMalloc(&ln.buf, bytes));
            ln.cap = bytes;
            ln.device = dev;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, dev, root_device) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(root_device, 0);
            (void)hipGetLastError();  // "already enabled" is fine
        }
        rc = membership ? memo_query_membership_dev(shards[g], a, b, k, num_docs, (uint32_t *)ln.buf, ln.stream)
                        : memo_query_conservation_dev(shards[g], a, b, k, num_docs, (uint16_t *)ln.buf, ln.stream);
        if (rc) return rc;
        HIP_TRY(hipStreamWaitEvent(ln.stream, ready, 0));
        HIP_TRY(hipMemcpyPeerAsync(dst, root_device, ln.buf, dev, bytes, ln.stream));
        HIP_TRY(hipEventRecord(ln.done, ln.stream));
        {
            DeviceGuard root(root_device);
            HIP_TRY(hipStreamWaitEvent(rs, ln.done, 0));  // the result is complete in root_stream's order
        }
    }
    return MEMO_OK;
}

}  // namespace

extern "C" {

int memo_conservation_multi(const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows,
                            int64_t qs, int64_t qe, int32_t k, int32_t num_docs, uint16_t *out,
                            const int32_t *devices, int32_t n_devices) {
    return host_multi(start, end, annot, rows, qs, qe, k, num_docs, out, devices, n_devices, false);
}

int memo_membership_multi(const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows,
                          int64_t qs, int64_t qe, int32_t k, int32_t num_docs, uint32_t *out_bits,
                          const int32_t *devices, int32_t n_devices) {
    return host_multi(start, end, annot, rows, qs, qe, k, num_docs, out_bits, devices, n_devices, true);
}

int memo_query_conservation_multi_dev(memo_index_t *const *shards, int32_t n_shards, int64_t qs, int64_t qe,
                                      int32_t k, int32_t num_docs, uint16_t *d_out, int32_t root_device,
                                      void *root_stream, double root_weight) {
    return dev_multi(shards, n_shards, qs, qe, k, num_docs, d_out, root_device, root_stream, root_weight, false);
}

int memo_query_membership_multi_dev(memo_index_t *const *shards, int32_t n_shards, int64_t qs, int64_t qe,
                                    int32_t k, int32_t num_docs, uint32_t *d_out, int32_t root_device,
                                    void *root_stream, double root_weight) {
    return dev_multi(shards, n_shards, qs, qe, k, num_docs, d_out, root_device, root_stream, root_weight, true);
}

}  // extern "C"
