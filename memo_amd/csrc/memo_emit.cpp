// memo_emit.cpp -- print_res (/root/reference/src/memo_query.py:65-71) as host C++.
//
// Byte-identical text:  conservation = print(*vec, sep='\n')  (decimal + '\n' per position, a lone
// '\n' for an empty vector);  membership = np.savetxt(rec.astype('byte'), fmt='%i', delimiter=' ').
// No device code here; part of libmemo_amd.so so that one library serves the whole seam.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "memo_amd.h"
#include "memo_cpus.h"
#include "memo_amd_dap.h"

// ---- print_res (memo_query.py:65-71) -----------------------------------------------------
// Host-side formatters, split over threads: the text of config 3 is ~290 MB (conservation) and
// 20 GB (membership); a single core formatting it would dwarf the 0.5-2 ms sweep.
namespace {

unsigned emit_threads(int64_t items, int64_t min_per_thread) {
    unsigned hw = (unsigned)memo::cpu_budget();  // allowed CPUs cut to the cgroup's CFS quota (memo_cpus.h)
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

// The DAP text of index.sh:83 is whitespace-separated decimal integers; parsing it in Python costs
// ~40x the GPU work on it.  Threads take line-aligned pieces of the buffer, count their numbers,
// then parse them to their final offsets.  Returns the number of integers found (parsed only when
// they fit in cap); -1 on a character that is neither a digit, a sign nor whitespace.
int64_t memo_parse_ints(const char *text, size_t len, int64_t *out, size_t cap) {
    if (!len) return 0;
    const unsigned nt = emit_threads((int64_t)len, 1 << 22);
    std::vector<size_t> cut(nt + 1, len);
    cut[0] = 0;
    for (unsigned t = 1; t < nt; ++t) {  // move each cut forward to just behind a newline
        size_t c = len / nt * t;
        while (c < len && text[c - 1] != '\n') ++c;
        cut[t] = c;
    }
    std::vector<int64_t> count(nt + 1, 0);
    std::vector<int> bad(nt, 0);
    auto scan = [&](unsigned t, int64_t *dst) {
        int64_t n = 0;
        const char *p = text + cut[t], *e = text + cut[t + 1];
        while (p < e) {
            const char ch = *p;
            if (ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r') { ++p; continue; }
            bool neg = false;
            if (ch == '-' || ch == '+') { neg = ch == '-'; ++p; }
            if (p >= e || *p < '0' || *p > '9') { bad[t] = 1; return n; }
            int64_t v = 0;
            while (p < e && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
            if (dst) dst[n] = neg ? -v : v;
            ++n;
        }
        return n;
    };
    parallel_chunks((int64_t)nt, nt, [&](unsigned, int64_t b, int64_t e) {
        for (int64_t t = b; t < e; ++t) count[t + 1] = scan((unsigned)t, nullptr);
    });
    for (unsigned t = 0; t < nt; ++t) {
        if (bad[t]) return -1;
        count[t + 1] += count[t];
    }
    const int64_t total = count[nt];
    if ((size_t)total > cap || !out) return total;
    parallel_chunks((int64_t)nt, nt, [&](unsigned, int64_t b, int64_t e) {
        for (int64_t t = b; t < e; ++t) scan((unsigned)t, out + count[t]);
    });
    return total;
}

// BED rows of the index builder: print('\t'.join(map(str, [header, start, end, annot])))
// (dap_to_bed.py:105,109)
size_t memo_emit_bed(const int32_t *rec, const int64_t *start, const int64_t *end, const int32_t *annot,
                     uint64_t rows, const char *names, int32_t nrec, char *buf, size_t cap) {
    if (!rows) return 0;
    std::vector<const char *> name(nrec);
    std::vector<size_t> nlen(nrec);
    const char *q = names;
    for (int r = 0; r < nrec; ++r) {
        name[r] = q;
        nlen[r] = strlen(q);
        q += nlen[r] + 1;
    }
    auto digits = [](long long v) {
        size_t n = v < 0 ? 2 : 1;
        for (unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v; u >= 10; u /= 10) ++n;
        return n;
    };
    auto put = [](char *p, long long v) {
        char tmp[24];
        int n = 0;
        unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
        do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) *p++ = '-';
        while (n) *p++ = tmp[--n];
        return p;
    };
    const unsigned nt = emit_threads((int64_t)rows, 1 << 18);
    std::vector<size_t> bytes(nt + 1, 0);
    parallel_chunks((int64_t)rows, nt, [&](unsigned t, int64_t b, int64_t e) {
        size_t n = 0;
        for (int64_t i = b; i < e; ++i) n += nlen[rec[i]] + digits(start[i]) + digits(end[i]) + digits(annot[i]) + 4;
        bytes[t + 1] = n;
    });
    for (unsigned t = 0; t < nt; ++t) bytes[t + 1] += bytes[t];
    const size_t need = bytes[nt];
    if (need > cap || !buf) return need;
    parallel_chunks((int64_t)rows, nt, [&](unsigned t, int64_t b, int64_t e) {
        char *p = buf + bytes[t];
        for (int64_t i = b; i < e; ++i) {
            memcpy(p, name[rec[i]], nlen[rec[i]]);
            p += nlen[rec[i]];
            *p++ = '\t';
            p = put(p, start[i]);
            *p++ = '\t';
            p = put(p, end[i]);
            *p++ = '\t';
            p = put(p, annot[i]);
            *p++ = '\n';
        }
    });
    return need;
}

}  // extern "C"
