// tools/ms_sam.cpp -- matching statistics of a pivot against genomes, by suffix automaton (tools/realistic_index.py).
//
// MS_g[i] = length of the longest prefix of pivot[i:] that occurs in text_g, where text_g is what index.sh builds for
// genome g (/root/reference/src/index.sh:63-65): the genome's records and their reverse complements.  One row of
// MS values per pivot position is the DAP `memo index` feeds to dap_to_bed.py (index.sh:83).
//
// A suffix automaton of the REVERSED text recognises the reversed substrings; streaming the reversed pivot through
// it gives, for every position, the longest substring ENDING there in reversed coordinates = STARTING there in
// forward coordinates.  Linear time, ~28 bytes per automaton state, <= 2 |text| states.  Genomes are processed by a
// pool of threads (MS_THREADS, default 8); every thread owns one automaton at a time.
//
//   ms_sam PIVOT.bin OUT.i32 GENOME1.bin GENOME2.bin ...
// *.bin: one byte per base (0..3 = ACGT, 4 = record separator); OUT: int32 [positions][genomes], row-major.
// Development tool (not part of the product, not linked into the library).
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct Sam {
    struct State {
        int32_t next[5];
        int32_t link, len;
    };
    std::vector<State> st;
    int last = 0;
    explicit Sam(size_t n) {
        st.reserve(2 * n + 2);
        st.push_back(State{{-1, -1, -1, -1, -1}, -1, 0});
    }
    void extend(int c) {
        const int cur = (int)st.size();
        st.push_back(State{{-1, -1, -1, -1, -1}, -1, st[last].len + 1});
        int p = last;
        while (p != -1 && st[p].next[c] == -1) {
            st[p].next[c] = cur;
            p = st[p].link;
        }
        if (p == -1) {
            st[cur].link = 0;
        } else {
            const int q = st[p].next[c];
            if (st[p].len + 1 == st[q].len) {
                st[cur].link = q;
            } else {
                const int clone = (int)st.size();
                st.push_back(st[q]);
                st[clone].len = st[p].len + 1;
                while (p != -1 && st[p].next[c] == q) {
                    st[p].next[c] = clone;
                    p = st[p].link;
                }
                st[q].link = st[cur].link = clone;
            }
        }
        last = cur;
    }
};

static std::vector<uint8_t> slurp(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path);
        exit(2);
    }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> v((size_t)n);
    if (n && fread(v.data(), 1, (size_t)n, f) != (size_t)n) exit(2);
    fclose(f);
    return v;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: ms_sam PIVOT.bin OUT.i32 GENOME.bin...\n");
        return 2;
    }
    const std::vector<uint8_t> pivot = slurp(argv[1]);
    const int ng = argc - 3;
    const size_t L = pivot.size();
    std::vector<int32_t> out(L * (size_t)ng);
    std::atomic<int> next{0};
    int nthreads = getenv("MS_THREADS") ? atoi(getenv("MS_THREADS")) : 8;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > ng) nthreads = ng;
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&] {
            for (;;) {
                const int g = next.fetch_add(1);
                if (g >= ng) break;
                const std::vector<uint8_t> text = slurp(argv[3 + g]);
                Sam sam(text.size());
                for (size_t i = text.size(); i-- > 0;) sam.extend(text[i] > 4 ? 4 : text[i]);  // the reversed text
                int v = 0, l = 0;
                for (size_t i = L; i-- > 0;) {  // the reversed pivot
                    const int c = pivot[i];
                    if (c >= 4) {  // a record boundary of the pivot never matches
                        v = 0;
                        l = 0;
                    } else {
                        while (v != 0 && sam.st[v].next[c] == -1) {
                            v = sam.st[v].link;
                            l = sam.st[v].len;
                        }
                        if (sam.st[v].next[c] != -1) {
                            v = sam.st[v].next[c];
                            ++l;
                        } else {
                            l = 0;
                        }
                    }
                    out[i * (size_t)ng + (size_t)g] = l;
                }
            }
        });
    for (auto &x : th) x.join();
    FILE *f = fopen(argv[2], "wb");
    if (!f || fwrite(out.data(), sizeof(int32_t), out.size(), f) != out.size()) {
        fprintf(stderr, "cannot write %s\n", argv[2]);
        return 2;
    }
    fclose(f);
    return 0;
}
