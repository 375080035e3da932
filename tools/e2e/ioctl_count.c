// LD_PRELOAD shim: counts ioctl calls by request number and the CPU time spent in them (thread CPU clock), printed at exit.
// A measuring stick for the host side of the workflow: which KFD calls the HIP runtime makes per batch (tools/gpu/r03_ae.sh).
//   gcc -O2 -shared -fPIC -o ioctl_count.so ioctl_count.c -ldl ; LD_PRELOAD=./ioctl_count.so <program>
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdarg.h>
#include <stdatomic.h>
#include <stdio.h>
#include <time.h>

static int (*real_ioctl)(int, unsigned long, ...);
static _Atomic unsigned long calls[256], cpu_ns[256], wall_ns[256];

static unsigned long now(clockid_t c)
{
    struct timespec t;
    clock_gettime(c, &t);
    return (unsigned long)t.tv_sec * 1000000000ul + (unsigned long)t.tv_nsec;
}

int ioctl(int fd, unsigned long req, ...)
{
    va_list ap;
    va_start(ap, req);
    void* arg = va_arg(ap, void*);
    va_end(ap);
    if (!real_ioctl)
        real_ioctl = (int (*)(int, unsigned long, ...))dlsym(RTLD_NEXT, "ioctl");
    const unsigned long c0 = now(CLOCK_THREAD_CPUTIME_ID), w0 = now(CLOCK_MONOTONIC);
    const int r = real_ioctl(fd, req, arg);
    const unsigned nr = (unsigned)(req & 0xFFu);
    atomic_fetch_add(&calls[nr], 1ul);
    atomic_fetch_add(&cpu_ns[nr], now(CLOCK_THREAD_CPUTIME_ID) - c0);
    atomic_fetch_add(&wall_ns[nr], now(CLOCK_MONOTONIC) - w0);
    return r;
}

__attribute__((destructor)) static void dump(void)
{
    for (unsigned nr = 0; nr < 256; ++nr)
        if (calls[nr])
            fprintf(stderr, "ioctl nr 0x%02x: %lu calls, %.3f s cpu, %.3f s wall\n", nr, (unsigned long)calls[nr], cpu_ns[nr] * 1e-9, wall_ns[nr] * 1e-9);
}
