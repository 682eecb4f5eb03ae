// memo_emit.cpp -- print_res (/root/reference/src/memo_query.py:65-71) as host C++.
//
// Byte-identical text:  conservation = print(*vec, sep='\n')  (decimal + '\n' per position, a lone
// '\n' for an empty vector);  membership = np.savetxt(rec.astype('byte'), fmt='%i', delimiter=' ').
// No device code here; part of libmemo_amd.so so that one library serves the whole seam.
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

#include "memo_amd.h"

// ---- print_res (memo_query.py:65-71) -----------------------------------------------------
// Host-side formatters, split over threads: the text of config 3 is ~290 MB (conservation) and
// 20 GB (membership); a single core formatting it would dwarf the 0.5-2 ms sweep.
namespace {

unsigned emit_threads(int64_t items, int64_t min_per_thread) {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    if (hw > 64) hw = 64;
    if (const char *v = getenv("MEMO_EMIT_THREADS")) {
        const int n = atoi(v);
        if (n > 0) hw = (unsigned)n;
    }
    const int64_t want = items / min_per_thread + 1;
    return (unsigned)(want < (int64_t)hw ? want : (int64_t)hw);
}

template <typename F>
void parallel_chunks(int64_t n, unsigned nt, F f) {  // f(chunk index, begin, end)
    if (nt <= 1) {
        f(0u, (int64_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(nt);
    const int64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const int64_t b = (int64_t)t * per, e = b + per < n ? b + per : n;
        th.emplace_back([=] { if (b < e) f(t, b, e); });
    }
    for (auto &x : th) x.join();
}

inline int dec_len(unsigned v) { return v < 10 ? 1 : v < 100 ? 2 : v < 1000 ? 3 : v < 10000 ? 4 : 5; }

}  // namespace

extern "C" {

size_t memo_emit_conservation(const uint16_t *vec, int64_t L, char *buf, size_t cap) {
    if (L <= 0) {  // print(*[], sep='\n') still writes the newline
        if (cap >= 1 && buf) buf[0] = '\n';
        return 1;
    }
    const unsigned nt = emit_threads(L, 1 << 20);
    std::vector<size_t> bytes(nt + 1, 0);
    parallel_chunks(L, nt, [&](unsigned t, int64_t b, int64_t e) {
        size_t n = 0;
        for (int64_t i = b; i < e; ++i) n += (size_t)dec_len(vec[i]) + 1;
        bytes[t + 1] = n;
    });
    for (unsigned t = 0; t < nt; ++t) bytes[t + 1] += bytes[t];
    const size_t need = bytes[nt];
    if (need > cap || !buf) return need;
    parallel_chunks(L, nt, [&](unsigned t, int64_t b, int64_t e) {
        char *p = buf + bytes[t];
        for (int64_t i = b; i < e; ++i) {
            unsigned v = vec[i];
            const int n = dec_len(v);
            for (int d = n - 1; d >= 0; --d) { p[d] = (char)('0' + v % 10); v /= 10; }
            p[n] = '\n';
            p += n + 1;
        }
    });
    return need;
}

size_t memo_emit_membership(const uint32_t *bits, int64_t L, int32_t num_docs, char *buf, size_t cap) {
    if (L <= 0) return 0;
    const size_t per_line = num_docs > 0 ? (size_t)2 * num_docs : 1;
    const size_t need = per_line * (size_t)L;
    if (need > cap || !buf) return need;
    const int nw = (num_docs + 31) / 32;
    parallel_chunks(L, emit_threads(L * (int64_t)per_line, 1 << 22), [&](unsigned, int64_t b, int64_t e) {
        char *p = buf + (size_t)b * per_line;
        for (int64_t i = b; i < e; ++i) {
            const uint32_t *row = bits + i * nw;
            for (int g = 0; g < num_docs; ++g) {
                *p++ = (char)('0' + ((row[g >> 5] >> (g & 31)) & 1u));
                *p++ = ' ';
            }
            if (num_docs > 0) p[-1] = '\n'; else *p++ = '\n';
        }
    });
    return need;
}

}  // extern "C"
