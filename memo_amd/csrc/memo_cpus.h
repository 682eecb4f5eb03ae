// memo_cpus.h -- how many host threads this process may usefully run: the CPUs it is allowed on (affinity mask), cut
// to its cgroup's CFS bandwidth quota.  Header-only (memo_hostcore.cpp and memo_emit.cpp, which is also built alone).
//
// Why the quota matters (profiles/r06_oneshot.txt): the GPU boxes of this pool show 256 CPUs and give the container
// `cpu.max = 1600000 100000` -- 1.6 CPU-seconds per 100 ms period.  A pool of 32 busy threads spends that in 50 ms and
// the whole cgroup is then frozen until the period ends; 64 threads freeze after 25 ms.  That, not NUMA and not the
// fork-joins, is why "64 threads and up lose a third" (profiles/r02_oneshot_host_threads.txt): threads beyond the
// quota do not add throughput, they bring the freeze forward.  Under a quota the currency is CPU-seconds per row.
#ifndef MEMO_CPUS_H
#define MEMO_CPUS_H

#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

namespace memo {

// CPUs the calling thread may run on (affinity mask; cpusets show up here), at least 1
inline int cpus_allowed() {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int n = CPU_COUNT(&set);
        if (n > 0) return n;
    }
    const unsigned hw = std::thread::hardware_concurrency();
    return hw ? (int)hw : 1;
}

// the smallest CFS quota, in CPUs, on the way from this process's cgroup to the root; 0 = none found
inline double cgroup_cpu_quota() {
    double best = 0;
    auto take = [&](double q) {
        if (q > 0 && (best == 0 || q < best)) best = q;
    };
    auto read_v2 = [&](const std::string &dir) {  // cgroup v2: "max 100000" or "1600000 100000"
        FILE *f = fopen((dir + "/cpu.max").c_str(), "r");
        if (!f) return;
        char q[64] = {0};
        long long period = 0;
        if (fscanf(f, "%63s %lld", q, &period) == 2 && period > 0 && strcmp(q, "max") != 0) take(atof(q) / (double)period);
        fclose(f);
    };
    auto read_v1 = [&](const std::string &dir) {  // cgroup v1: cpu.cfs_quota_us (-1 = none) / cpu.cfs_period_us
        FILE *fq = fopen((dir + "/cpu.cfs_quota_us").c_str(), "r"), *fp = fopen((dir + "/cpu.cfs_period_us").c_str(), "r");
        long long quota = -1, period = 0;
        if (fq && fp && fscanf(fq, "%lld", &quota) == 1 && fscanf(fp, "%lld", &period) == 1 && quota > 0 && period > 0)
            take((double)quota / (double)period);
        if (fq) fclose(fq);
        if (fp) fclose(fp);
    };
    // this process's cgroup path(s): "0::/a/b" (v2), "4:cpu,cpuacct:/a/b" (v1)
    std::string v2_path = "/", v1_path = "/";
    if (FILE *f = fopen("/proc/self/cgroup", "r")) {
        char line[1024];
        while (fgets(line, sizeof line, f)) {
            char *c1 = strchr(line, ':');
            char *c2 = c1 ? strchr(c1 + 1, ':') : nullptr;
            if (!c2) continue;
            std::string ctrl(c1 + 1, c2), path(c2 + 1);
            while (!path.empty() && (path.back() == '\n' || path.back() == '\r')) path.pop_back();
            if (ctrl.empty()) v2_path = path;
            else if (ctrl.find("cpu") != std::string::npos && ctrl.find("cpuset") == std::string::npos) v1_path = path;
        }
        fclose(f);
    }
    for (std::string p = v2_path;;) {  // every level up to the mount's root
        read_v2("/sys/fs/cgroup" + (p == "/" ? std::string() : p));
        if (p == "/" || p.empty()) break;
        const size_t cut = p.rfind('/');
        p = cut == 0 || cut == std::string::npos ? "/" : p.substr(0, cut);
    }
    for (std::string p = v1_path;;) {
        read_v1("/sys/fs/cgroup/cpu" + (p == "/" ? std::string() : p));
        read_v1("/sys/fs/cgroup/cpu,cpuacct" + (p == "/" ? std::string() : p));
        if (p == "/" || p.empty()) break;
        const size_t cut = p.rfind('/');
        p = cut == 0 || cut == std::string::npos ? "/" : p.substr(0, cut);
    }
    return best;
}

// threads worth running flat out: allowed CPUs, no more than the quota pays for (rounded down, at least 1)
inline int cpu_budget() {
    static const int budget = [] {
        int n = cpus_allowed();
        const double q = cgroup_cpu_quota();
        if (q > 0 && q < (double)n) n = q < 1.0 ? 1 : (int)q;
        return n;
    }();
    return budget;
}

}  // namespace memo

#endif  // MEMO_CPUS_H
