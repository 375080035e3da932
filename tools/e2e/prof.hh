// Poor man's sampling profiler for the e2e probe (no perf on the GPU box): ITIMER_PROF -> SIGPROF lands on a running thread, the
// handler keeps the top frames; at exit the samples are written as "library+offset symbol" lines (dladdr), leaf first, one sample
// per line.  PG_E2E_PROF=<file> switches it on.  Build the libraries with -g and resolve static functions here with addr2line.
#pragma once
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/time.h>

namespace e2eprof
{
enum { kDepth = 18, kMax = 400000 };
static void* g_frames[kMax][kDepth];
static unsigned char g_n[kMax];
static std::atomic<unsigned> g_next(0);
static std::atomic<bool> g_on(false);

static void handler(int)
{
    if (!g_on.load(std::memory_order_relaxed))
        return;
    const unsigned i = g_next.fetch_add(1, std::memory_order_relaxed);
    if (i >= kMax)
        return;
    const int n = backtrace(g_frames[i], kDepth);
    g_n[i] = (unsigned char)(n < 0 ? 0 : n);
}

inline void start(int hz = 2000)
{
    void* warm[4];
    backtrace(warm, 4);  // first call loads libgcc: not inside a signal handler
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = handler;
    sa.sa_flags = SA_RESTART;
    sigaction(SIGPROF, &sa, nullptr);
    struct itimerval it;
    it.it_interval.tv_sec = 0;
    it.it_interval.tv_usec = 1000000 / hz;
    it.it_value = it.it_interval;
    setitimer(ITIMER_PROF, &it, nullptr);
}
inline void enable(bool on) { g_on.store(on); }
inline void dump(const char* path)
{
    struct itimerval off;
    memset(&off, 0, sizeof off);
    setitimer(ITIMER_PROF, &off, nullptr);
    FILE* f = fopen(path, "w");
    if (!f)
        return;
    const unsigned n = g_next.load() < (unsigned)kMax ? g_next.load() : (unsigned)kMax;
    for (unsigned i = 0; i < n; ++i)
    {
        for (int d = 2; d < g_n[i]; ++d)  // skip the handler and the signal trampoline
        {
            Dl_info info;
            if (dladdr(g_frames[i][d], &info) && info.dli_fname)
            {
                const char* base = strrchr(info.dli_fname, '/');
                fprintf(f, "%s+0x%lx %s;", base ? base + 1 : info.dli_fname, (unsigned long)((char*)g_frames[i][d] - (char*)info.dli_fbase),
                        info.dli_sname ? info.dli_sname : "?");
            }
            else
                fprintf(f, "?+%p ?;", g_frames[i][d]);
        }
        fputc('\n', f);
    }
    fclose(f);
}
}  // namespace e2eprof
