// Internal: run fn(i) for i in [0, n) on up to `threads` host threads (dynamic chunks of `grain` indexes); the first
// exception is rethrown on the caller.
#pragma once
#include <algorithm>
#include <atomic>
#include <exception>
#include <thread>
#include <vector>

namespace pghost
{
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
