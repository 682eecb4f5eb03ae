// tools/host_probe.cpp -- what the GPU box's HOST can deliver to the packer of the one-shot seam (memo_hostcore.cpp):
// topology, cgroup CPU quota, where pages land, streaming-read bandwidth of three int64 columns by thread count, with
// the columns first-touched by ONE thread (what a NumPy caller gives) and by the reading threads themselves.
// Development tool:  g++ -O3 -mavx2 -std=c++17 -pthread tools/host_probe.cpp -o /tmp/host_probe && /tmp/host_probe 200000000
#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void cat(const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) { printf("%s: (absent)\n", path); return; }
    char buf[4096];
    size_t n = fread(buf, 1, sizeof buf - 1, f);
    buf[n] = 0;
    fclose(f);
    printf("%s: %s%s", path, buf, n && buf[n - 1] == '\n' ? "" : "\n");
}

static void nodes_of(const void *p, size_t bytes, const char *what) {
    const size_t page = 4096, step = bytes / 64;
    int hist[16] = {0}, other = 0;
    for (int i = 0; i < 64; ++i) {
        void *q = (void *)(((uintptr_t)p + i * step) & ~(page - 1));
        int st = -1;
        if (syscall(SYS_move_pages, 0, 1ul, &q, nullptr, &st, 0) != 0) { other++; continue; }
        if (st >= 0 && st < 16) hist[st]++; else other++;
    }
    printf("  pages of %s by node:", what);
    for (int i = 0; i < 16; ++i) if (hist[i]) printf(" node%d=%d", i, hist[i]);
    printf(" other=%d (of 64 samples)\n", other);
}

static int64_t *big(size_t n) {
    void *p = mmap(nullptr, n * 8, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); exit(1); }
    madvise(p, n * 8, MADV_HUGEPAGE);
    return (int64_t *)p;
}

template <typename F>
static void par(int T, F f) {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([=] { f(t); });
    for (auto &x : th) x.join();
}

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 200000000ull;
    printf("== topology\n");
    printf("hardware_concurrency %u, sched_getaffinity ", std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) printf("%d cpus\n", CPU_COUNT(&set)); else printf("?\n");
    cat("/sys/fs/cgroup/cpu.max");
    cat("/sys/fs/cgroup/cpu/cpu.cfs_quota_us");
    cat("/sys/fs/cgroup/cpuset.cpus.effective");
    cat("/sys/fs/cgroup/cpuset.mems.effective");
    cat("/sys/fs/cgroup/memory.max");
    cat("/sys/devices/system/node/online");
    for (int i = 0; i < 8; ++i) {
        char p[128];
        snprintf(p, sizeof p, "/sys/devices/system/node/node%d/cpulist", i);
        if (access(p, R_OK) == 0) cat(p);
    }
    cat("/sys/kernel/mm/transparent_hugepage/enabled");
    cat("/sys/kernel/mm/transparent_hugepage/defrag");
    cat("/proc/sys/kernel/numa_balancing");
    for (int mode = 0; mode < 2; ++mode) {
        int64_t *s = big(n), *e = big(n), *a = big(n);
        const int TT = 64;
        double t0 = now();
        if (mode == 0) {
            memset(s, 1, n * 8); memset(e, 2, n * 8); memset(a, 3, n * 8);
        } else {
            par(TT, [&](int t) {
                size_t b = n * t / TT, en = n * (t + 1) / TT;
                memset(s + b, 1, (en - b) * 8); memset(e + b, 2, (en - b) * 8); memset(a + b, 3, (en - b) * 8);
            });
        }
        printf("== columns of %zu rows (%.1f GB) first-touched by %s: %.0f ms\n", n, 24e-9 * n, mode ? "64 threads" : "one thread",
               (now() - t0) * 1e3);
        nodes_of(s, n * 8, "start");
        nodes_of(a, n * 8, "annot");
        for (int T : {4, 8, 16, 24, 32, 48, 64, 96, 128, 192, 256}) {
            if ((unsigned)T > std::thread::hardware_concurrency()) break;
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                std::vector<uint64_t> acc((size_t)T * 8);
                // dynamic 64 Ki-row blocks from one counter, like the packer's tasks
                std::atomic<size_t> next{0};
                const size_t blk = 1 << 16, nb = (n + blk - 1) / blk;
                double t1 = now();
                par(T, [&](int t) {
                    uint64_t x = 0;
                    for (;;) {
                        size_t b = next.fetch_add(1, std::memory_order_relaxed);
                        if (b >= nb) break;
                        size_t i0 = b * blk, i1 = i0 + blk < n ? i0 + blk : n;
                        for (size_t i = i0; i < i1; ++i) x += (uint64_t)(s[i] ^ e[i] ^ a[i]);
                    }
                    acc[(size_t)t * 8] = x;
                });
                double dt = now() - t1;
                if (dt < best) best = dt;
            }
            printf("  read T=%3d: %.1f ms  %.1f GB/s\n", T, best * 1e3, 24e-9 * n / best);
            fflush(stdout);
        }
        munmap(s, n * 8); munmap(e, n * 8); munmap(a, n * 8);
    }
}
