// tools/hostpack_bench.cpp -- the host packer alone (memo_hostcore.cpp + the device seam of tests/host_stub.cpp: "device
// memory" is host memory, copies are memcpy on a copier thread): rows per second by thread count, with and without the
// vectorised row pass.  Development tool:  g++ -O3 -std=c++17 -pthread -I include tools/hostpack_bench.cpp
// memo_amd/csrc/memo_hostcore.cpp -o /tmp/hostpack_bench && MEMO_HOST_THREADS=32 /tmp/hostpack_bench 200000000 1
#define main stub_main
#include "../tests/host_stub.cpp"
#undef main
#include <chrono>
int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 100000000ull;
    const int dense = argc > 2 ? atoi(argv[2]) : 1;
    std::vector<int64_t> s(n), e(n), a(n);
    uint64_t x = 88172645463325252ull;
    for (uint64_t i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        s[i] = 1 + (int64_t)(i / 5);
        e[i] = s[i] + (int64_t)(x % 60);
        a[i] = 1 + (int64_t)((x >> 20) % 99);
    }
    for (int rep = 0; rep < 4; ++rep) {
        memo_builder *b = new_builder(n, dense);
        auto t0 = std::chrono::steady_clock::now();
        int rc = builder_push_core(b, s.data(), e.data(), a.data(), n);
        rc |= builder_flush_core(b);
        rc |= b->ring->drain();
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("rc %d  %.1f ms  %.2f Grows/s  %.1f GB/s of int64 columns (%d threads, %s)\n", rc, dt * 1e3, n / dt * 1e-9,
               24.0 * n / dt * 1e-9, HostPool::get().threads(), dense ? "dense rows" : "4-byte words");
        free_builder(b);
    }
}
