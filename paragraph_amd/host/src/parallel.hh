// Internal: run fn(i) for i in [0, n) on up to `threads` host threads (dynamic chunks of `grain` indexes); the first
// exception is rethrown on the caller.
#pragma once
#include <algorithm>
#include <atomic>
#include <exception>
#include <thread>
#include <vector>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sched.h>
#include <string>

namespace pghost
{
// CPUs this process can really run on at once: the hardware threads, its affinity mask and -- in a container -- the CPU
// bandwidth of its cgroup (v2: cpu.max, v1: cpu.cfs_quota_us / cpu.cfs_period_us: a box may show 256 CPUs and allow 16).  The command lines default their host threads to it.
inline int usableCpus()
{
    int n = (int)std::thread::hardware_concurrency();
    if (n <= 0)
        n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0)
        n = std::min(n, (int)CPU_COUNT(&set));
    bool quota_seen = false;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r"))  // cgroup v2
    {
        char quota[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && strcmp(quota, "max") != 0)
        {
            const long q = atol(quota);
            if (q > 0)
            {
                n = std::min(n, (int)std::max(1L, (q + period / 2) / period));
                quota_seen = true;
            }
        }
        fclose(f);
    }
    if (!quota_seen)  // cgroup v1: cpu.cfs_quota_us / cpu.cfs_period_us of the cpu controller (-1 = no limit)
    {
        auto read_long = [](const char* path, long& v) {
            FILE* f = fopen(path, "r");
            if (!f)
                return false;
            const bool ok = fscanf(f, "%ld", &v) == 1;
            fclose(f);
            return ok;
        };
        long q = -1, period = 0;
        for (const char* dir : { "/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct" })
        {
            const std::string base(dir);
            if (read_long((base + "/cpu.cfs_quota_us").c_str(), q) && read_long((base + "/cpu.cfs_period_us").c_str(), period))
                break;
            q = -1;
        }
        if (q > 0 && period > 0)
            n = std::min(n, (int)std::max(1L, (q + period / 2) / period));
    }
    return std::max(1, n);
}

template <typename Fn> void parallelFor(size_t n, int threads, Fn fn, size_t grain = 1)
{
    grain = std::max<size_t>(grain, 1);
    const size_t chunks = (n + grain - 1) / grain;
    const size_t workers = std::max<size_t>(1, std::min<size_t>((size_t)std::max(threads, 1), chunks));
    if (workers == 1)
    {
        for (size_t i = 0; i < n; ++i)
            fn(i);
        return;
    }
    std::atomic<size_t> next(0);
    std::exception_ptr failure;
    std::atomic<bool> failed(false);
    auto work = [&] {
        for (;;)
        {
            const size_t c = next.fetch_add(1);
            if (c >= chunks || failed.load())
                return;
            try
            {
                for (size_t i = c * grain, e = std::min(n, i + grain); i < e; ++i)
                    fn(i);
            }
            catch (...)
            {
                if (!failed.exchange(true))
                    failure = std::current_exception();
                return;
            }
        }
    };
    std::vector<std::thread> pool;
    for (size_t w = 1; w < workers; ++w)
        pool.emplace_back(work);
    work();
    for (auto& t : pool)
        t.join();
    if (failure)
        std::rethrow_exception(failure);
}
}  // namespace pghost
