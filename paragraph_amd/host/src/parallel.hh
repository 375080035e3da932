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

namespace pghost
{
// CPUs this process can really run on at once: the hardware threads, its affinity mask and -- in a container -- the CPU
// bandwidth of its cgroup (cpu.max: a box may show 256 CPUs and allow 16).  The command lines default their host threads to it.
inline int usableCpus()
{
    int n = (int)std::thread::hardware_concurrency();
    if (n <= 0)
        n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0)
        n = std::min(n, (int)CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r"))
    {
        char quota[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && strcmp(quota, "max") != 0)
        {
            const long q = atol(quota);
            if (q > 0)
                n = std::min(n, (int)std::max(1L, (q + period / 2) / period));
        }
        fclose(f);
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
