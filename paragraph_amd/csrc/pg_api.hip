// pg_api.hip -- host side of the C ABI declared in include/paragraph_amd.h.
//
// Host responsibilities (everything here is plumbing; all alignment arithmetic runs in the HIP
// kernels of pg_fill.hip / pg_trace.hip):
//   * graph upload: builds, for every graph and both graph directions, the linear column layout
//     (node order = topological id order, as GraphAlignerImpl::initializeGraph does,
//     GraphAligner.cpp:110-167; reversed graph as graphtools::reverseGraph,
//     GT!/src/graphcore/GraphOperations.cpp:38-60)
//   * batch upload: buckets reads by (rows-per-lane variant, graph), packs them 4 per wavefront,
//     plans chunks so that the traceback workspace of a chunk fits the configured HBM budget
//   * batch align: per chunk one fill launch (forward + reversed graph wavefronts) and one
//     pick/traceback launch on the ctx stream
// There is no CPU fallback: without a HIP device every entry point that needs one fails.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <chrono>
#include <map>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <sys/prctl.h>
#include <time.h>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/paragraph_amd.h"
#include "pg_device.h"
#include "pg_general.h"
#include "pg_internal.h"
#include "pg_kernels.h"
#include "pg_kmerindex.h"

pg_status pg_fail(pg_ctx* ctx, pg_status st, const std::string& msg)
{
    static std::mutex err_mutex;  // uploads / downloads of one batch may overlap the stage calls of another (paragraph_amd.h)
    if (ctx)
    {
        std::lock_guard<std::mutex> lock(err_mutex);
        ctx->err = msg;
    }
    return st;
}
static pg_status fail(pg_ctx* ctx, pg_status st, const std::string& msg) { return pg_fail(ctx, st, msg); }
static void recycle_sync_events(pg_ctx* ctx);
static void recycle_done_sync_events(pg_ctx* ctx);


// ---------------------------------------------------------------------------------------------------
// cache of idle device blocks (pg_internal.h)
// ---------------------------------------------------------------------------------------------------
namespace
{
struct DevCache
{
    std::mutex m;
    std::unordered_map<size_t, std::vector<void*>> idle;  // by class size
    std::unordered_map<void*, size_t> class_of;           // every block that came from pg_dev_alloc
    size_t idle_bytes = 0;
};
const size_t PG_DEV_CACHE_MAX_IDLE = 8ull << 30;
DevCache& dev_cache()
{
    static DevCache* caches = new DevCache[64];  // one per device ordinal; never torn down (like the HIP runtime itself)
    int dev = 0;
    (void)hipGetDevice(&dev);
    return caches[dev & 63];
}
size_t class_size(size_t bytes)
{
    size_t c = 256;
    while (c < bytes)
        c <<= 1;
    if (c >= 2048)
    {
        const size_t step = c / 16;  // eight classes between c / 2 and c
        c = (bytes + step - 1) / step * step;
    }
    return c;
}
}  // namespace

hipError_t pg_dev_alloc(void** p, size_t bytes)
{
    DevCache& dc = dev_cache();
    const size_t c = class_size(std::max<size_t>(bytes, 1));
    {
        std::lock_guard<std::mutex> lock(dc.m);
        auto it = dc.idle.find(c);
        if (it != dc.idle.end() && !it->second.empty())
        {
            *p = it->second.back();
            it->second.pop_back();
            dc.idle_bytes -= c;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, c);
    if (e != hipSuccess)
    {
        (void)hipGetLastError();
        pg_dev_cache_release();  // idle blocks may be what is in the way
        e = hipMalloc(p, c);
    }
    if (e == hipSuccess)
    {
        std::lock_guard<std::mutex> lock(dc.m);
        dc.class_of[*p] = c;
    }
    return e;
}

hipError_t pg_dev_free(void* p)
{
    if (!p)
        return hipSuccess;
    DevCache& dc = dev_cache();
    // A full cache makes room by letting its LARGEST idle blocks go, not by refusing the block in hand: what fills it are the blocks
    // of large batches that are over (a million-read batch leaves gigabytes behind), what a workflow cycles through per batch are
    // kilobytes to megabytes -- with the cache closed to them every one would go through hipFree (a device synchronisation) and
    // hipMalloc.
    std::vector<void*> victims;
    bool cached = false;
    {
        std::lock_guard<std::mutex> lock(dc.m);
        auto it = dc.class_of.find(p);
        if (it != dc.class_of.end() && it->second <= PG_DEV_CACHE_MAX_IDLE / 2)
        {
            const size_t c = it->second;
            while (dc.idle_bytes + c > PG_DEV_CACHE_MAX_IDLE)
            {
                size_t big = 0;
                for (auto& kv : dc.idle)
                    if (!kv.second.empty() && kv.first > big)
                        big = kv.first;
                if (big <= c)
                    break;  // (nothing larger than this block left to let go)
                std::vector<void*>& v = dc.idle[big];
                victims.push_back(v.back());
                dc.class_of.erase(v.back());
                v.pop_back();
                dc.idle_bytes -= big;
            }
            if (dc.idle_bytes + c <= PG_DEV_CACHE_MAX_IDLE)
            {
                dc.idle[c].push_back(p);
                dc.idle_bytes += c;
                cached = true;
            }
        }
        if (!cached && it != dc.class_of.end())
            dc.class_of.erase(it);
    }
    hipError_t e = hipSuccess;
    for (void* v : victims)
    {
        const hipError_t ev = hipFree(v);
        if (e == hipSuccess)
            e = ev;
    }
    if (!cached)
    {
        const hipError_t ep = hipFree(p);
        if (e == hipSuccess)
            e = ep;
    }
    return e;
}

void pg_dev_cache_release()
{
    DevCache& dc = dev_cache();
    std::vector<void*> blocks;
    {
        std::lock_guard<std::mutex> lock(dc.m);
        for (auto& kv : dc.idle)
        {
            for (void* p : kv.second)
            {
                blocks.push_back(p);
                dc.class_of.erase(p);
            }
            kv.second.clear();
        }
        dc.idle_bytes = 0;
    }
    for (void* p : blocks)
        (void)hipFree(p);
}

extern "C" const char* pg_strerror(pg_status st)
{
    switch (st)
    {
    case PG_OK: return "ok";
    case PG_ERR_INVALID: return "invalid argument";
    case PG_ERR_NO_DEVICE: return "no usable HIP device";
    case PG_ERR_HIP: return "HIP runtime error";
    case PG_ERR_UNSUPPORTED: return "outside the supported envelope";
    case PG_ERR_NOMEM: return "out of memory";
    case PG_ERR_OVERFLOW: return "output buffer too small";
    default: return "unknown status";
    }
}

extern "C" const char* pg_last_error(const pg_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }

// ---- page-locked staging blocks, cached by power-of-two size class ------------------------------------------------
namespace
{
std::mutex g_pinned_mutex;
std::unordered_map<size_t, std::vector<void*>> g_pinned_idle;
}  // namespace

hipError_t pg_pinned_get(size_t bytes, void** p, size_t* cap)
{
    size_t c = 65536;
    while (c < bytes)
        c <<= 1;
    *cap = c;
    {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        auto& idle = g_pinned_idle[c];
        if (!idle.empty())
        {
            *p = idle.back();
            idle.pop_back();
            return hipSuccess;
        }
    }
    return hipHostMalloc(p, c, hipHostMallocPortable);
}

void pg_pinned_put(void* p, size_t cap)
{
    if (!p)
        return;
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    auto& idle = g_pinned_idle[cap];
    if (idle.size() < 64)
        idle.push_back(p);
    else
        (void)hipHostFree(p);
}

hipError_t PgStagedUpload::commit_async(hipStream_t stream, void** device_block, void** staging, size_t* staging_cap)
{
    *device_block = nullptr;
    *staging = nullptr;
    *staging_cap = 0;
    size_t total = 0;
    for (Item const& it : items)
        total += (std::max<size_t>(it.bytes, 1) + 255) & ~(size_t)255;
    void* host = nullptr;
    size_t cap = 0;
    hipError_t e = pg_pinned_get(total, &host, &cap);
    if (e != hipSuccess)
        return e;
    void* dev = nullptr;
    e = pg_dev_alloc(&dev, total);
    if (e == hipSuccess)
    {
        size_t at = 0;
        for (Item const& it : items)
        {
            if (it.bytes)
                memcpy((char*)host + at, it.src, it.bytes);
            *it.dst = (char*)dev + at;
            at += (std::max<size_t>(it.bytes, 1) + 255) & ~(size_t)255;
        }
        e = hipMemcpyAsync(dev, host, total, hipMemcpyHostToDevice, stream);
    }
    if (e != hipSuccess)
    {
        pg_pinned_put(host, cap);
        (void)pg_dev_free(dev);
        for (Item const& it : items)
            *it.dst = nullptr;
        return e;
    }
    *device_block = dev;
    *staging = host;
    *staging_cap = cap;
    return hipSuccess;
}

hipError_t PgStagedUpload::commit(hipStream_t stream, void** device_block)
{
    *device_block = nullptr;
    size_t total = 0;
    for (Item const& it : items)
        total += (std::max<size_t>(it.bytes, 1) + 255) & ~(size_t)255;
    void* host = nullptr;
    size_t cap = 0;
    hipError_t e = pg_pinned_get(total, &host, &cap);
    if (e != hipSuccess)
        return e;
    void* dev = nullptr;
    e = pg_dev_alloc(&dev, total);
    if (e == hipSuccess)
    {
        size_t at = 0;
        for (Item const& it : items)
        {
            if (it.bytes)
                memcpy((char*)host + at, it.src, it.bytes);
            *it.dst = (char*)dev + at;
            at += (std::max<size_t>(it.bytes, 1) + 255) & ~(size_t)255;
        }
        e = hipMemcpyAsync(dev, host, total, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess)
        {
            int device = 0;
            (void)hipGetDevice(&device);
            e = pg_stream_wait(device, stream);
        }
    }
    pg_pinned_put(host, cap);
    if (e != hipSuccess)
    {
        (void)pg_dev_free(dev);
        for (Item const& it : items)
            *it.dst = nullptr;
        return e;
    }
    *device_block = dev;
    return hipSuccess;
}

extern "C" pg_status pg_device_prefer_blocking_waits(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
        return PG_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess)
        return PG_ERR_HIP;
    (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);  // refused once the device is in use: then nothing changes
    (void)hipGetLastError();
    return PG_OK;
}

extern "C" pg_status pg_ctx_create(int device, pg_ctx** out)
{
    if (!out)
        return PG_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
        return PG_ERR_NO_DEVICE;
    pg_ctx* ctx = new (std::nothrow) pg_ctx();
    if (!ctx)
        return PG_ERR_NOMEM;
    ctx->device = device;
    // The three streams must sit on three HARDWARE queues: kernels of streams that share one run one after the other, and the
    // design rests on the traceback + count of chunk n running under the fill of chunk n + 1.  The runtime hands its (by
    // default four) hardware queues of a priority level to streams round-robin over every stream of the process -- with
    // torch's, RCCL's and other contexts' streams around, two of ours ended up on one queue (fill 9.6 ms + traceback 1.0 ms in
    // series, profiles/r03_*).  Queues are pooled per priority level, so the count and copy streams are created one level up:
    // there they share a pool with nobody but each other.  (PG_STREAM_PRIORITY=0 creates all three at the default level.)
    int prio_least = 0, prio_greatest = 0;
    if (hipSetDevice(device) == hipSuccess)
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    // reads of 251..512 bases: 32 lanes per read (two wavefronts per work item); PG_WIDE16=1 = the 16-lane kernels, for A/B timing
    ctx->wide32 = getenv("PG_WIDE16") == nullptr;
    {
        const char* le = getenv("PG_LEAN");  // (the default of pg_ctx_set_lean: on; PG_LEAN=0 = the plain four fills)
        ctx->lean = !(le && le[0] == '0');
        if (const char* mp = getenv("PG_LEAN_MIN_CELLS"))
            ctx->lean_min_cells_default = (uint64_t)atof(mp);
        ctx->lean_min_cells = (le && le[0] == '2') ? 0u : ctx->lean_min_cells_default;
        if (const char* lf = getenv("PG_LEAN_FUSED"))
            ctx->lean_fused = lf[0] != '0';
    }
    {
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0)
            ctx->max_lds_per_block = (uint64_t)lds;
    }
    const char* pe = getenv("PG_STREAM_PRIORITY");
    const int side_prio = (pe && pe[0] == '0') ? 0 : prio_greatest;
    // (the second fill stream right after the first: the runtime deals its hardware queues round-robin, so the two land on
    // different ones)
    const bool fills_low = getenv("PG_FILLS_LOW") != nullptr;  // (A/B: both fill streams one level DOWN, a pool nobody else uses)
    if (hipSetDevice(device) != hipSuccess
        || (fills_low ? hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_least) : hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess
        || ((getenv("PG_FILL2_LOW") || fills_low) ? hipStreamCreateWithPriority(&ctx->stream_fill2, hipStreamNonBlocking, prio_least)  // (A/B: a pool of its own)
                                                  : hipStreamCreateWithFlags(&ctx->stream_fill2, hipStreamNonBlocking)) != hipSuccess
        || hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, side_prio) != hipSuccess
        || hipStreamCreateWithPriority(&ctx->stream_seed, hipStreamNonBlocking, side_prio) != hipSuccess
        || hipStreamCreateWithPriority(&ctx->stream_copy, hipStreamNonBlocking, side_prio) != hipSuccess)
    {
        delete ctx;
        return PG_ERR_HIP;
    }
    ctx->side_priority = side_prio;
    if (const char* ss = getenv("PG_SEED_STREAMS"))  // (A/B; the streams beyond the first are made by the first path stage)
        ctx->seed_streams = std::min(4, std::max(1, atoi(ss)));
    if (const char* fs = getenv("PG_FILL_STREAMS"))  // A/B timing: overrides pg_ctx_set_fill_streams' default of this context
        ctx->fill_streams = fs[0] == '2' ? 2 : 1;
    *out = ctx;
    return PG_OK;
}

extern "C" pg_status pg_ctx_set_lean(pg_ctx* ctx, int on)
{
    if (!ctx)
        return PG_ERR_INVALID;
    ctx->lean = on != 0;
    if (on == 1)
        ctx->lean_min_cells = ctx->lean_min_cells_default;
    else if (on >= 2)
        ctx->lean_min_cells = 0;  // every chunk of byte-variant reads (tests: small batches through the lean kernels)
    return PG_OK;
}

extern "C" pg_status pg_ctx_set_fill_streams(pg_ctx* ctx, int n)
{
    if (!ctx || (n != 1 && n != 2))
        return PG_ERR_INVALID;
    if (getenv("PG_FILL_STREAMS"))
        return PG_OK;  // the environment decides (A/B runs)
    if (n == ctx->fill_streams)
        return PG_OK;
    // the regions are re-cut: nothing may be in flight on them
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_fill2));
    if (ctx->stream_lean)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_lean));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
    for (auto& e : ctx->region_free)
        e = nullptr;
    ctx->fill_streams = n;
    ++ctx->plan_epoch;  // chunks cut for the old region size must not meet the new regions (pg_batch_align checks)
    return PG_OK;
}

extern "C" void pg_ctx_destroy(pg_ctx* ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream_fill2)
        (void)hipStreamSynchronize(ctx->stream_fill2);
    if (ctx->stream_lean)
        (void)hipStreamSynchronize(ctx->stream_lean);
    if (ctx->stream_seed)
        (void)hipStreamSynchronize(ctx->stream_seed);
    for (hipStream_t s : ctx->stream_seed_more)
        if (s)
            (void)hipStreamSynchronize(s);
    if (ctx->stream2)
        (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream_copy)
        (void)hipStreamSynchronize(ctx->stream_copy);
    recycle_sync_events(ctx);
    for (auto e : ctx->sync_event_pool)
        (void)hipEventDestroy(e);
    for (auto& e : ctx->events)
    {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    for (auto e : ctx->event_pool)
        (void)hipEventDestroy(e);
    if (ctx->workspace)
        (void)hipFree(ctx->workspace);
    if (ctx->gen_ws)
        (void)hipFree(ctx->gen_ws);
    for (void* p : ctx->klib_scratch)
        (void)pg_dev_free(p);
    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream_fill2)
        (void)hipStreamDestroy(ctx->stream_fill2);
    if (ctx->stream_lean)
        (void)hipStreamDestroy(ctx->stream_lean);
    if (ctx->stream_seed)
        (void)hipStreamDestroy(ctx->stream_seed);
    for (hipStream_t s : ctx->stream_seed_more)
        if (s)
            (void)hipStreamDestroy(s);
    if (ctx->stream2)
        (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream_copy)
        (void)hipStreamDestroy(ctx->stream_copy);
    pg_dev_cache_release();
    delete ctx;
}

extern "C" pg_status pg_ctx_set_workspace_bytes(pg_ctx* ctx, uint64_t bytes)
{
    if (!ctx || bytes < (64ull << 20))
        return PG_ERR_INVALID;
    ctx->ws_limit = bytes;
    return PG_OK;
}

extern "C" pg_status pg_ctx_sync(pg_ctx* ctx)
{
    if (!ctx)
        return PG_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_copy));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_fill2));
    if (ctx->stream_lean)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_lean));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_seed));
    for (hipStream_t s : ctx->stream_seed_more)
        if (s)
            HIP_TRY(ctx, hipStreamSynchronize(s));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
    recycle_sync_events(ctx);
    return PG_OK;
}

// Pinned (page-locked) host staging: with it every hipMemcpyAsync of the upload / download calls is a real DMA that runs
// beside the kernels; from pageable memory the runtime stages through its own bounce buffers on the calling thread.
extern "C" pg_status pg_host_alloc(pg_ctx* ctx, size_t bytes, void** out)
{
    if (!ctx || !out)
        return PG_ERR_INVALID;
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocPortable));
    return PG_OK;
}

extern "C" void pg_host_free(pg_ctx* ctx, void* p)
{
    if (!p)
        return;
    if (ctx)
        (void)hipSetDevice(ctx->device);
    (void)hipHostFree(p);
}

extern "C" pg_status pg_host_register(pg_ctx* ctx, void* p, size_t bytes)
{
    if (!ctx || !p || !bytes)
        return PG_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipHostRegister(p, bytes, hipHostRegisterPortable));
    return PG_OK;
}

extern "C" pg_status pg_host_unregister(pg_ctx* ctx, void* p)
{
    if (!ctx || !p)
        return PG_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipHostUnregister(p));
    return PG_OK;
}

extern "C" pg_status pg_counts_zero(pg_ctx* ctx, uint32_t* d_counts, uint64_t n_counters)
{
    if (!ctx || (!d_counts && n_counters))
        return PG_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n_counters)
        HIP_TRY(ctx, hipMemsetAsync(d_counts, 0, n_counters * sizeof(uint32_t), ctx->stream2));  // the stream the count path runs on
    return PG_OK;
}

// Blocks the host until everything queued on the compute streams so far is done (the copy stream is left alone): the
// ordering point between pg_batch_count into a caller-owned table and the caller's all-reduce of that table.
extern "C" pg_status pg_ctx_sync_compute(pg_ctx* ctx)
{
    if (!ctx)
        return PG_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_fill2));
    if (ctx->stream_lean)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_lean));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_seed));
    for (hipStream_t s : ctx->stream_seed_more)
        if (s)
            HIP_TRY(ctx, hipStreamSynchronize(s));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
    return PG_OK;
}

// Non-blocking ordering between the count stream and a stream of the caller (the all-reduce of the counter table).
extern "C" pg_status pg_ctx_native_stream(pg_ctx* ctx, int which, void** out)
{
    if (!ctx || !out || which < PG_STREAM_FILL || which > PG_STREAM_COPY)
        return PG_ERR_INVALID;
    *out = which == PG_STREAM_FILL ? (void*)ctx->stream : which == PG_STREAM_COUNT ? (void*)ctx->stream2 : (void*)ctx->stream_copy;
    return PG_OK;
}

extern "C" pg_status pg_ctx_count_record(pg_ctx* ctx, void* native_event)
{
    if (!ctx || !native_event)
        return PG_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord((hipEvent_t)native_event, ctx->stream2));
    return PG_OK;
}

extern "C" pg_status pg_ctx_count_wait(pg_ctx* ctx, void* native_event)
{
    if (!ctx || !native_event)
        return PG_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, (hipEvent_t)native_event, 0));
    return PG_OK;
}

// A stage waits for the batch's inputs and for the batch's previous stage (which may have run on the other compute stream).
hipError_t pg_stage_begin_on(pg_ctx* ctx, pg_batch* b, hipStream_t s)
{
    (void)ctx;
    hipError_t e = hipSuccess;
    if (b->upload_recorded)
        e = hipStreamWaitEvent(s, b->ev_upload, 0);
    if (e == hipSuccess && b->busy_recorded)
        e = hipStreamWaitEvent(s, b->ev_busy, 0);
    return e;
}

hipError_t pg_stage_end_on(pg_ctx* ctx, pg_batch* b, hipStream_t s)
{
    if (const pg_graphs* G = b->graphs)
    {
        const int w = s == ctx->stream ? 0 : ctx->is_seed_stream(s) ? 2 : 1;
        hipError_t e = hipSuccess;
        if (!G->ev_use[w])
            e = hipEventCreateWithFlags(&G->ev_use[w], pg_wait_event_flags());
        // ONE event per slot, several seed streams: re-recording it on another stream would drop what it said about the first.
        // The stream recorded now first waits for the event's previous recording, so the new one covers every earlier use.
        if (e == hipSuccess && G->use_recorded[w] && G->use_stream[w] != s)
            e = hipStreamWaitEvent(s, G->ev_use[w], 0);
        if (e == hipSuccess)
            e = hipEventRecord(G->ev_use[w], s);
        if (e != hipSuccess)
            return e;
        G->use_recorded[w] = true;
        G->use_stream[w] = s;
    }
    if (!b->ev_busy)
        return hipSuccess;
    b->busy_recorded = true;
    return hipEventRecord(b->ev_busy, s);
}

hipError_t pg_stage_begin(pg_ctx* ctx, pg_batch* b) { return pg_stage_begin_on(ctx, b, ctx->stream); }
hipError_t pg_stage_end(pg_ctx* ctx, pg_batch* b) { return pg_stage_end_on(ctx, b, ctx->stream); }

bool pg_api_timing = getenv("PG_API_TIMING") != nullptr;
uint64_t pg_now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
namespace
{
struct ApiTimes
{
    std::mutex m;
    std::unordered_map<const char*, std::pair<uint64_t, uint64_t>> by_site;  // call text (a literal: its address is the key) -> (ns, calls)
    ~ApiTimes()
    {
        if (by_site.empty())
            return;
        std::map<std::string, std::pair<uint64_t, uint64_t>> merged;
        for (auto const& kv : by_site)
        {
            merged[kv.first].first += kv.second.first;
            merged[kv.first].second += kv.second.second;
        }
        std::vector<std::pair<std::string, std::pair<uint64_t, uint64_t>>> rows(merged.begin(), merged.end());
        std::sort(rows.begin(), rows.end(), [](auto const& a, auto const& b) { return a.second.first > b.second.first; });
        fprintf(stderr, "[PG_API_TIMING] host time per call site (ms total, calls, us per call)\n");
        for (auto const& r : rows)
            fprintf(stderr, "[PG_API_TIMING] %10.2f %8llu %9.1f  %s\n", r.second.first / 1e6, (unsigned long long)r.second.second,
                    r.second.first / 1e3 / (double)std::max<uint64_t>(1, r.second.second), r.first.c_str());
    }
};
ApiTimes& apiTimes()
{
    static ApiTimes t;
    return t;
}
}  // namespace
void pg_api_time_add(const char* what, uint64_t t0_ns)
{
    const uint64_t dt = pg_now_ns() - t0_ns;
    ApiTimes& t = apiTimes();
    std::lock_guard<std::mutex> lock(t.m);
    auto& e = t.by_site[what];
    e.first += dt;
    e.second += 1;
}

// ---- one thread waits for the device on behalf of all callers -----------------------------------------------------------------
// A host thread blocked in hipEventSynchronize sleeps in the kernel driver until an interrupt arrives -- ANY completion interrupt
// of the process: every one of them wakes every sleeper, which looks at its own signal and goes back to sleep.  With 24 lanes of a
// workflow each waiting for its batch that was 35 000 wait ioctls per pass of 52 batches (profiles/r05_e2e_ioctls.txt), a fifth of
// the job's CPU together with the runtime's side of them.  Here the callers sleep on a condition variable of their own and ONE
// thread looks at the pending events (hipEventQuery reads the completion signal in host memory: no system call) every 50 us.
// PG_WAITER=0: every caller waits in the runtime as before (A/B runs).
namespace
{
struct WaitService
{
    struct Req
    {
        hipEvent_t ev;
        int device;
        bool done = false;
        hipError_t err = hipSuccess;
        std::condition_variable cv;
    };
    std::mutex m;
    std::condition_variable cv_new;
    std::vector<Req*> pending;
    bool started = false;
    void run()
    {
        int device = -1;
        std::unique_lock<std::mutex> lock(m);
        for (;;)
        {
            while (pending.empty())
                cv_new.wait(lock);
            for (size_t i = 0; i < pending.size();)
            {
                Req* r = pending[i];
                if (r->device != device && hipSetDevice(r->device) == hipSuccess)
                    device = r->device;
                const hipError_t e = hipEventQuery(r->ev);
                if (e == hipErrorNotReady)
                {
                    ++i;
                    continue;
                }
                r->err = e;
                r->done = true;
                r->cv.notify_one();
                pending[i] = pending.back();
                pending.pop_back();
            }
            if (!pending.empty())
            {
                lock.unlock();
                struct timespec ts = { 0, 50 * 1000 };
                nanosleep(&ts, nullptr);
                lock.lock();
            }
        }
    }
};
WaitService& waitService()
{
    static WaitService* w = new WaitService();  // (never destroyed: its thread outlives main's statics)
    return *w;
}
}  // namespace

hipError_t pg_event_wait(int device, hipEvent_t ev)
{
    static const bool off = [] {
        const char* e = getenv("PG_WAITER");
        return (e && e[0] == '0') || getenv("PG_SPIN_WAITS") != nullptr;
    }();
    if (off)
        return hipEventSynchronize(ev);
    const hipError_t q = hipEventQuery(ev);
    if (q != hipErrorNotReady)
        return q;
    WaitService& w = waitService();
    WaitService::Req r;
    r.ev = ev;
    r.device = device;
    std::unique_lock<std::mutex> lock(w.m);
    if (!w.started)
    {
        w.started = true;
        std::thread([&w] {
            prctl(PR_SET_TIMERSLACK, 1UL, 0UL, 0UL, 0UL);  // (a 50 us sleep that lasts 50 us)
            w.run();
        }).detach();
    }
    w.pending.push_back(&r);
    w.cv_new.notify_one();
    r.cv.wait(lock, [&r] { return r.done; });
    return r.err;
}

// hipStreamSynchronize through the wait service: an event from a small pool is recorded on the stream and waited for
hipError_t pg_stream_wait(int device, hipStream_t s)
{
    static std::mutex m;
    static std::vector<std::pair<int, hipEvent_t>> idle;
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> lock(m);
        for (size_t i = 0; i < idle.size(); ++i)
            if (idle[i].first == device)
            {
                ev = idle[i].second;
                idle[i] = idle.back();
                idle.pop_back();
                break;
            }
    }
    hipError_t e = hipSuccess;
    if (!ev)
        e = hipEventCreateWithFlags(&ev, pg_wait_event_flags());
    if (e != hipSuccess)
        return hipStreamSynchronize(s);
    e = hipEventRecord(ev, s);
    if (e == hipSuccess)
        e = pg_event_wait(device, ev);
    std::lock_guard<std::mutex> lock(m);
    idle.emplace_back(device, ev);
    return e;
}

unsigned pg_wait_event_flags()
{
    static const bool spin = getenv("PG_SPIN_WAITS") != nullptr;
    return hipEventDisableTiming | (spin ? 0u : (unsigned)hipEventBlockingSync);
}

hipError_t pg_wait_stream(pg_batch* b, hipStream_t s)
{
    if (!b || !b->ev_host)
        return hipStreamSynchronize(s);
    const hipError_t e = hipEventRecord(b->ev_host, s);
    return e != hipSuccess ? e : pg_event_wait(b->device, b->ev_host);
}

hipError_t pg_batch_wait(pg_ctx* ctx, pg_batch* b)
{
    (void)ctx;
    hipError_t e = hipSuccess;
    if (b->upload_recorded)
        e = pg_event_wait(b->device, b->ev_upload);
    if (e == hipSuccess && b->busy_recorded)
        e = pg_event_wait(b->device, b->ev_busy);
    return e;
}

static pg_status drain_events(pg_ctx* ctx)
{
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_fill2));
    if (ctx->stream_lean)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_lean));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
    for (auto& e : ctx->events)
    {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, e.a, e.b));
        if (e.kind == 0)
        {
            ctx->acc.fill_ms += ms;
            ctx->acc.fill_launches++;
        }
        else if (e.kind == 4)
        {
            ctx->acc.fill_ms += ms;
            ctx->acc.fill_launches++;
            ctx->acc.lean_fused_ms += ms;
            ctx->acc.lean_fused_launches++;
        }
        else if (e.kind == 2)
        {
            ctx->acc.fill_ms += ms;
            ctx->acc.fill_launches++;
            ctx->acc.lean_rev_ms += ms;
            ctx->acc.lean_rev_launches++;
        }
        else if (e.kind == 3)
        {
            ctx->acc.fill_ms += ms;
            ctx->acc.lean_fwd_ms += ms;
            ctx->acc.lean_fwd_launches++;
        }
        else
        {
            ctx->acc.trace_ms += ms;
            ctx->acc.trace_launches++;
        }
        ctx->event_pool.push_back(e.a);
        ctx->event_pool.push_back(e.b);
    }
    ctx->events.clear();
    return PG_OK;
}

extern "C" pg_status pg_ctx_timing_enable(pg_ctx* ctx, int enable)
{
    if (!ctx)
        return PG_ERR_INVALID;
    ctx->timing = enable != 0;
    return PG_OK;
}

extern "C" pg_status pg_ctx_timing_reset(pg_ctx* ctx)
{
    if (!ctx)
        return PG_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pg_status st = drain_events(ctx);
    if (st != PG_OK)
        return st;
    ctx->acc = pg_timing{};
    return PG_OK;
}

extern "C" pg_status pg_ctx_timing_get(pg_ctx* ctx, pg_timing* out)
{
    if (!ctx || !out)
        return PG_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pg_status st = drain_events(ctx);
    if (st != PG_OK)
        return st;
    *out = ctx->acc;
    return PG_OK;
}

// ordering-only events (no timing) for the two-stream chunk pipeline; recycled whenever the ctx is synchronised
static hipError_t get_sync_event(pg_ctx* ctx, hipEvent_t* ev)
{
    if (!ev)
        return hipSuccess;
    if (!ctx->sync_event_pool.empty())
    {
        *ev = ctx->sync_event_pool.back();
        ctx->sync_event_pool.pop_back();
        return hipSuccess;
    }
    return hipEventCreateWithFlags(ev, hipEventDisableTiming);
}
static void recycle_sync_events(pg_ctx* ctx)
{
    for (auto e : ctx->sync_events_in_flight)
        ctx->sync_event_pool.push_back(e);
    ctx->sync_events_in_flight.clear();
    for (auto& e : ctx->region_free)
        e = nullptr;  // every traceback is over (the caller synchronised the compute streams)
}
// A workflow never calls pg_ctx_sync: ordering events whose work is over (they complete in the order they were recorded
// per stream, so the scan stops at the first one still pending) go back to the pool at the start of every pg_batch_align.
static void recycle_done_sync_events(pg_ctx* ctx)
{
    size_t done = 0;
    while (done < ctx->sync_events_in_flight.size() && hipEventQuery(ctx->sync_events_in_flight[done]) == hipSuccess)
    {
        const hipEvent_t e = ctx->sync_events_in_flight[done++];
        for (auto& rf : ctx->region_free)
            if (rf == e)
                rf = nullptr;  // that traceback is over: the region needs no wait (and the event gets a new job)
        ctx->sync_event_pool.push_back(e);
    }
    if (done)
        ctx->sync_events_in_flight.erase(ctx->sync_events_in_flight.begin(), ctx->sync_events_in_flight.begin() + (std::ptrdiff_t)done);
    (void)hipGetLastError();  // hipErrorNotReady of the first pending event is not an error
}

static hipError_t get_event(pg_ctx* ctx, hipEvent_t* ev)
{
    if (!ctx->event_pool.empty())
    {
        *ev = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return hipSuccess;
    }
    return hipEventCreate(ev);
}

// ---------------------------------------------------------------------------------------------------
// graphs
// ---------------------------------------------------------------------------------------------------
static char up(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }
static uint32_t nt_code_host(char c)
{
    switch (c)
    {
    case 'A':
    case 'U':
        return 0;
    case 'C':
        return 1;
    case 'G':
        return 2;
    case 'T':
        return 3;
    default:
        return 4;
    }
}

extern "C" pg_status pg_graphs_upload(
    pg_ctx* ctx, uint32_t n_graphs, const uint32_t* node_off, const uint32_t* seq_off, const char* seq,
    const uint32_t* pred_off, const uint32_t* pred, pg_graphs** out)
{
    if (!ctx || !out || !node_off || !seq_off || !seq || !pred_off || (n_graphs == 0))
        return fail(ctx, PG_ERR_INVALID, "pg_graphs_upload: null argument or no graphs");
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    std::vector<PgGraphDev> gdev(n_graphs);
    std::vector<PgNode> nodes;
    std::vector<uint32_t> preds;
    std::vector<uint32_t> colmeta;
    std::vector<char> seqchars;
    std::vector<HostGraph> host(n_graphs);
    {
        // (sizes are known up front: one word per column and direction + the idle tail, one character per column).  The offsets
        // have not been validated yet: every size is clamped, and a reservation that fails is only a missed optimisation --
        // nothing may throw out of an extern "C" function (the per-graph checks below return PG_ERR_INVALID for bad offsets)
        const uint32_t total_nodes = node_off[n_graphs];
        const uint64_t total_cols = seq_off[total_nodes] >= seq_off[0] ? seq_off[total_nodes] - seq_off[0] : 0;
        const uint64_t total_preds = pred ? pred_off[total_nodes] : 0;
        if (total_cols < (1ull << 30) && total_nodes < (1u << 28) && total_preds < (1ull << 30) && node_off[0] <= total_nodes)
        {
            try
            {
                colmeta.reserve(2 * total_cols + 2ull * n_graphs * PG_META_PAD);
                seqchars.reserve(total_cols);
                nodes.reserve(2ull * total_nodes);
                preds.reserve(2ull * total_preds + 1);
            }
            catch (std::exception const&)
            {
            }
        }
    }
    std::vector<std::vector<uint32_t>> succ;  // (reused from graph to graph)
    std::vector<uint32_t> ps;

    for (uint32_t g = 0; g < n_graphs; ++g)
    {
        const uint32_t nb = node_off[g], ne = node_off[g + 1];
        if (ne <= nb)
            return fail(ctx, PG_ERR_INVALID, "graph without nodes");
        const uint32_t n = ne - nb;
        if (n > PG_MAX_NODES)
            return fail(ctx, PG_ERR_UNSUPPORTED, "graph with more than 65535 nodes");
        uint64_t total = 0;
        for (uint32_t i = 0; i < n; ++i)
        {
            const uint32_t len = seq_off[nb + i + 1] - seq_off[nb + i];
            if (seq_off[nb + i + 1] <= seq_off[nb + i])
                return fail(ctx, PG_ERR_INVALID, "empty node sequence");
            if (len >= (1u << 24))  // pg_general.hip's per-node maximum key holds the in-node column in 24 bits
                return fail(ctx, PG_ERR_UNSUPPORTED, "node of 2^24 columns or more");
            total += len;
            for (uint32_t k = pred_off[nb + i]; k < pred_off[nb + i + 1]; ++k)
            {
                if (!pred || pred[k] >= i)
                    return fail(ctx, PG_ERR_INVALID, "edge breaks topological order");
                if (k > pred_off[nb + i] && pred[k] <= pred[k - 1])
                    return fail(ctx, PG_ERR_INVALID, "predecessors must be ascending and unique");
            }
        }
        // beyond 65 519 columns the packed kernels' 16-bit column fields end: the graph's reads take the general path
        // (pg_general.h), which reads nodes / predecessors / characters only -- no column words are built for it
        // ... and beyond 4 095 nodes the 12-bit node field of the column words ends
        const bool wide = total > 65535 - PG_GROUP_LANES || n > PG_MAX_PACKED_NODES;
        if (total > 0x7FFFFFFFull)
            return fail(ctx, PG_ERR_UNSUPPORTED, "graph longer than 2^31 columns");
        host[g].n_nodes = n;
        host[g].ncols = (uint32_t)total;
        host[g].general_only = wide;

        // successors (forward ids)
        if (succ.size() < n)
            succ.resize(n);
        for (uint32_t i = 0; i < n; ++i)
            succ[i].clear();
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t k = pred_off[nb + i]; k < pred_off[nb + i + 1]; ++k)
                succ[pred[k]].push_back(i);

        gdev[g].seq_off = (uint32_t)seqchars.size();
        gdev[g].pad = 0;
        for (int dir = 0; dir < 2; ++dir)
        {
            PgGraphDir& gd = gdev[g].dir[dir];
            gd.meta_off = (uint32_t)colmeta.size();
            const size_t dir_meta_begin = colmeta.size();
            gd.ncols = (uint32_t)total;
            gd.node_off = (uint32_t)nodes.size();
            gd.n_nodes = n;
            uint32_t col = 0;
            for (uint32_t id = 0; id < n; ++id)
            {
                const uint32_t src = dir ? n - 1 - id : id;
                const uint32_t s0 = seq_off[nb + src], len = seq_off[nb + src + 1] - s0;
                PgNode nd;
                nd.col_start = col;
                nd.len = len;
                nd.pred_off = (uint32_t)preds.size();
                bool has_succ = false, has_far_succ = false;
                if (!dir)
                {
                    for (uint32_t k = pred_off[nb + id]; k < pred_off[nb + id + 1]; ++k)
                        preds.push_back(pred[k]);
                    has_succ = !succ[id].empty();
                    for (uint32_t s : succ[id])
                        has_far_succ |= (s != id + 1);
                }
                else
                {
                    // predecessors in the reversed graph = successors of src in the original, mapped
                    // s -> n-1-s, ascending
                    ps.clear();
                    for (uint32_t s : succ[src])
                        ps.push_back(n - 1 - s);
                    std::sort(ps.begin(), ps.end());
                    for (uint32_t p : ps)
                        preds.push_back(p);
                    // successors in the reversed graph = predecessors of src in the original
                    for (uint32_t k = pred_off[nb + src]; k < pred_off[nb + src + 1]; ++k)
                    {
                        has_succ = true;
                        has_far_succ |= ((n - 1 - pred[k]) != id + 1);
                    }
                }
                nd.n_pred = (uint32_t)preds.size() - nd.pred_off;
                nodes.push_back(nd);
                // forward direction keeps every seed (the traceback reads them); the reversed
                // direction only needs seeds that a non-adjacent successor will load
                const bool save = dir == 0 ? has_succ : has_far_succ;
                // predecessor summary of the node, carried by its first column's meta word (no table loads on the device
                // in the common cases): adjacent predecessor?, and none / exactly one far predecessor with id < 128 / general
                uint32_t pred_bits = 0;
                {
                    uint32_t n_far = 0, far = 0;
                    for (uint32_t k = nd.pred_off; k < nd.pred_off + nd.n_pred; ++k)
                    {
                        if (preds[k] + 1 == id)
                            pred_bits |= PG_META_PRED_ADJ;
                        else
                        {
                            ++n_far;
                            far = preds[k];
                        }
                    }
                    if (n_far == 1 && far < 128)
                        pred_bits |= PG_META_PRED_ONE | (far << PG_META_PRED_SHIFT);
                    else if (n_far != 0)
                        pred_bits |= PG_META_PRED_MANY;
                }
                for (uint32_t c = 0; c < len; ++c)
                {
                    const char ch = up(dir ? seq[s0 + len - 1 - c] : seq[s0 + c]);
                    if (!dir)
                        seqchars.push_back(ch);
                    if (wide)
                        continue;
                    uint32_t m = nt_code_host(ch) | (id << 8);
                    if (c == 0)
                        m |= PG_META_FIRST | pred_bits;
                    if (c == len - 1)
                        m |= PG_META_LAST | PG_META_LAST_HI | (save ? PG_META_SAVE : 0u);
                    colmeta.push_back(m);
                }
                col += len;
            }
            if (wide)
                continue;
            // PG_META_RARE: node boundary, or code 4 in this column or in the next one of the layout
            for (size_t i = dir_meta_begin; i < colmeta.size(); ++i)
            {
                const uint32_t next_code = i + 1 < colmeta.size() ? PG_META_CODE(colmeta[i + 1]) : 4u;
                if ((colmeta[i] & (PG_META_FIRST | PG_META_LAST)) || PG_META_CODE(colmeta[i]) >= 4u || next_code >= 4u)
                    colmeta[i] |= PG_META_RARE;
            }
            for (int c = 0; c < PG_META_PAD; ++c)
                colmeta.push_back(PG_META_IDLE);
        }
    }
    if (preds.empty())
        preds.push_back(0);

    pg_graphs* G = new (std::nothrow) pg_graphs();
    if (!G)
        return PG_ERR_NOMEM;
    G->n_graphs = n_graphs;
    G->host.swap(host);
    {
        // host CSR copies in the caller's indexing (the count path reports counters in that indexing)
        const uint32_t total_nodes = node_off[n_graphs];
        G->h_node_off.assign(node_off, node_off + n_graphs + 1);
        G->h_pred_off.assign(pred_off, pred_off + total_nodes + 1);
        G->h_pred.assign(pred, pred + (pred ? pred_off[total_nodes] : 0));
        G->h_nodeseq_off.assign(seq_off, seq_off + total_nodes + 1);
        G->h_seq_raw.assign(seq, seq + seq_off[total_nodes]);
        G->h_node_len.resize(total_nodes);
        for (uint32_t i = 0; i < total_nodes; ++i)
            G->h_node_len[i] = seq_off[i + 1] - seq_off[i];
    }
    // one staging block, one copy, one device block for the five tables -- on the copy stream: a graph set can be prepared
    // while another thread's batch occupies the compute stream
    PgStagedUpload up;
    up.add(gdev, &G->d_graphs);
    up.add(nodes, &G->d_nodes);
    up.add(preds, &G->d_preds);
    up.add(colmeta, &G->d_colmeta);
    up.add(seqchars, &G->d_seqchars);
    hipError_t e = up.commit(ctx->stream_copy, &G->d_layout_block);
    if (e != hipSuccess)
    {
        pg_graphs_destroy(ctx, G);
        return fail(ctx, PG_ERR_HIP, std::string("graph upload: ") + hipGetErrorString(e));
    }
    *out = G;
    return PG_OK;
}

extern "C" void pg_graphs_destroy(pg_ctx* ctx, pg_graphs* G)
{
    if (!G)
        return;
    if (ctx)
        (void)hipSetDevice(ctx->device);
    // only the stages that used THIS graph set have to be over (other lanes' batches keep the streams busy all the time)
    for (int w = 0; w < 3; ++w)
    {
        if (G->use_recorded[w])
            (void)hipEventSynchronize(G->ev_use[w]);
        if (G->ev_use[w])
            (void)hipEventDestroy(G->ev_use[w]);
    }
    (void)pg_dev_free(G->d_layout_block);  // d_graphs .. d_seqchars live in it
    (void)pg_dev_free(G->d_count_block);   // d_cnt_graphs .. d_in_mask
    pg_path_index_free(G->path_index);
    pg_path_index_free(G->filter_index);
    pg_kmer_index_free(G->kmer_index);
    pg_klib_index_free(G->klib_index);
    delete G;
}

// ---------------------------------------------------------------------------------------------------
// batches
// ---------------------------------------------------------------------------------------------------
extern "C" pg_status pg_batch_create(pg_ctx* ctx, pg_batch** out)
{
    if (!ctx || !out)
        return PG_ERR_INVALID;
    *out = new (std::nothrow) pg_batch();
    if (!*out)
        return PG_ERR_NOMEM;
    if (hipSetDevice(ctx->device) != hipSuccess || hipEventCreateWithFlags(&(*out)->ev_upload, pg_wait_event_flags()) != hipSuccess
        || hipEventCreateWithFlags(&(*out)->ev_busy, pg_wait_event_flags()) != hipSuccess
        || hipEventCreateWithFlags(&(*out)->ev_host, pg_wait_event_flags()) != hipSuccess)
    {
        delete *out;
        *out = nullptr;
        return PG_ERR_HIP;
    }
    (*out)->device = ctx->device;
    return PG_OK;
}

static void batch_free_device(pg_batch* b)
{
    for (void* p : b->parked_blocks)
        (void)pg_dev_free(p);
    b->parked_blocks.clear();
    (void)pg_dev_free(b->d_base_off);
    (void)pg_dev_free(b->d_bases);
    (void)pg_dev_free(b->d_bases_rc);
    b->d_bases_rc = nullptr;
    b->cap_bases_rc = 0;
    (void)pg_dev_free(b->d_items);
    (void)pg_dev_free(b->d_fillsum);
    (void)pg_dev_free(b->d_gen_reads);
    (void)pg_dev_free(b->d_gen_fsum);
    b->d_gen_reads = nullptr;
    b->d_gen_fsum = nullptr;
    b->cap_gen = 0;
    (void)pg_dev_free(b->d_results);
    (void)pg_dev_free(b->d_ops);
    (void)pg_dev_free(b->d_ops_counter);
    (void)pg_dev_free(b->d_graph_of_read);
    (void)pg_dev_free(b->d_path_flags);
    b->d_path_flags = nullptr;
    (void)pg_dev_free(b->d_active);
    b->d_active = nullptr;
    (void)pg_dev_free(b->d_inst);
    (void)pg_dev_free(b->d_lean_extra);
    (void)pg_dev_free(b->d_yloc);
    (void)pg_dev_free(b->d_lean_ucount);
    (void)pg_dev_free(b->d_lean_ulist);
    b->d_inst = nullptr;
    b->d_lean_extra = b->d_yloc = b->d_lean_ucount = b->d_lean_ulist = nullptr;
    b->cap_lean_pairs = b->cap_lean_reads = 0;
    b->has_active = false;
    (void)pg_dev_free(b->d_group_of_read2);
    (void)pg_dev_free(b->d_group_base);
    (void)pg_dev_free(b->d_group_count);
    (void)pg_dev_free(b->d_active_list);
    (void)pg_dev_free(b->d_segments);
    b->d_group_of_read2 = b->d_group_base = b->d_group_count = b->d_active_list = nullptr;
    b->d_segments = nullptr;
    b->cap_groups = b->cap_cascade_reads = b->cap_segments = 0;
    b->cascade_uploaded = false;
    (void)pg_dev_free(b->d_support);
    (void)pg_dev_free(b->d_label_ext);
    b->d_label_ext = nullptr;
    b->cap_label_ext = 0;
    (void)pg_dev_free(b->d_path);
    (void)pg_dev_free(b->d_path_counter);
    (void)pg_dev_free(b->d_frag_off);
    (void)pg_dev_free(b->d_frag_reads);
    (void)pg_dev_free(b->d_is_rev);
    (void)pg_dev_free(b->d_counts);
    b->d_graph_of_read = nullptr;
    b->d_support = nullptr;
    b->d_path = nullptr;
    b->d_path_counter = nullptr;
    b->d_frag_off = nullptr;
    b->d_frag_reads = nullptr;
    b->d_is_rev = nullptr;
    b->d_counts = nullptr;
    b->cap_count_reads = b->cap_frags = 0;
    b->cap_counts = 0;
    b->d_base_off = nullptr;
    b->d_bases = nullptr;
    b->d_items = nullptr;
    b->d_fillsum = nullptr;
    b->d_results = nullptr;
    b->d_ops = nullptr;
    b->d_ops_counter = nullptr;
    b->cap_reads = b->cap_bases = b->cap_items = 0;
    b->ops_cap = 0;
}

extern "C" void pg_batch_destroy(pg_ctx* ctx, pg_batch* b)
{
    if (!b)
        return;
    if (ctx)
    {
        // only this batch's own work has to be over: other batches may be on the device right now
        (void)hipSetDevice(ctx->device);
        (void)pg_batch_wait(ctx, b);
        (void)hipStreamSynchronize(ctx->stream_copy);
    }
    batch_free_device(b);
    if (b->h_counters)
        (void)hipHostFree(b->h_counters);
    if (b->ev_upload)
        (void)hipEventDestroy(b->ev_upload);
    if (b->ev_host)
        (void)hipEventDestroy(b->ev_host);
    if (b->ev_cascade)
        (void)hipEventDestroy(b->ev_cascade);
    if (b->ev_busy)
        (void)hipEventDestroy(b->ev_busy);
    delete b;
}

static inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

// workspace bytes of one item pair (its regions are proportional to its pipeline steps, i.e. to its work)
struct PairNeed
{
    uint64_t nsteps, trace_bytes, seed_bytes, need;
};
static PairNeed pair_need_of(int C, const HostGraph& hg)
{
    PairNeed n;
    // (the wide variants' sweeps run 32 lanes per read: 16 steps more; sized for that whichever kernel set runs)
    n.nsteps = pg_fill_steps_lanes(hg.ncols, pg_var_wide(C) ? PG_WIDE_LANES : PG_GROUP_LANES);
    n.trace_bytes = align_up(n.nsteps * 64 * pg_trace_lane_bytes(C), 256);
    n.seed_bytes = pg_seed_region_bytes(C, hg.n_nodes) + pg_key_region_bytes(hg.n_nodes);
    // + the traceback's CIGAR scratch of the pair's four reads (reversed-graph item's trace_off, which has no trace)
    const uint64_t ops_bytes = align_up((uint64_t)PG_GROUPS * pg_ops_cap(C) * sizeof(uint32_t), 256);
    n.need = n.trace_bytes + 2 * n.seed_bytes + ops_bytes;
    return n;
}

// Builds the wavefront work items + chunk plan for the reads with active[i] != 0 (all reads when active is
// NULL) and uploads them.  Items/fill summaries never need more room than the all-reads plan.
static pg_status plan_items(pg_ctx* ctx, pg_batch* b, const uint8_t* active, hipStream_t cs)
{
    const pg_graphs* G = b->graphs;
    const uint32_t n_reads = b->n_reads;
    const uint32_t* base_off = b->h_base_off.data();
    const uint32_t* graph_of_read = b->h_graph_of_read.data();
    b->chunks.clear();
    b->gen_idx.clear();
    b->n_pairs = 0;
    // ---- bucket reads by (variant, graph) -----------------------------------------------------------
    struct Key
    {
        uint32_t c, graph, idx;
    };
    std::vector<Key> keys;
    keys.reserve(n_reads);
    uint64_t gen_total = 0, gen_largest = 0;
    for (uint32_t i = 0; i < n_reads; ++i)
    {
        const uint32_t L = base_off[i + 1] - base_off[i];
        if (L == 0 || (active && !active[i]))
            continue;
        if (L > PG_MAX_READ_LEN || G->host[graph_of_read[i]].general_only)
        {
            // outside the packed kernels' envelope: the general path (pg_general.h).  Whether its matrices fit the workspace
            // budget is decided HERE, before anything of the batch is queued: pg_batch_align must not find out after the
            // packed chunks' kernels are in flight (the caller drops the site and recycles its device blocks on this error)
            const HostGraph& hg = G->host[graph_of_read[i]];
            // the general fill keeps two rolling columns + the query codes of its read in LDS: 5 bytes per base (pg_general.hip)
            if ((uint64_t)5 * L + 4 > ctx->max_lds_per_block)
                return fail(ctx, PG_ERR_UNSUPPORTED, "read too long for the general path's LDS columns on this device");
            const uint64_t gen_need = pg_gen_read_bytes(L, hg.ncols, hg.n_nodes);
            if (gen_need > ctx->ws_limit)
                return fail(ctx, PG_ERR_UNSUPPORTED, "workspace limit too small for one read of the general path (read length x graph columns)");
            gen_total += gen_need;
            gen_largest = std::max(gen_largest, gen_need);
            b->gen_idx.push_back(i);
            continue;
        }
        keys.push_back(Key{ (uint32_t)pg_variant_of(L), graph_of_read[i], i });
    }
    // ONE budget (pg_ctx_set_workspace_bytes) for both paths: the general path's share -- all of its reads when they fit 8 GiB or
    // a quarter of the budget, else groups of that size, and never less than its largest read -- comes out of what the packed
    // chunks may take, so the context's device memory stays within the budget and a batch that cannot fit is refused HERE.
    b->gen_reserve = b->gen_idx.empty() ? 0 : std::min<uint64_t>(gen_total, std::max<uint64_t>(gen_largest, std::min<uint64_t>(8ull << 30, ctx->ws_limit / 4)));
    const uint64_t packed_limit = ctx->ws_limit - b->gen_reserve;
    const auto key_less = [](const Key& x, const Key& y) { return x.c != y.c ? x.c < y.c : x.graph < y.graph; };
    if (!std::is_sorted(keys.begin(), keys.end(), key_less))  // reads of one length, site after site, arrive in order
        std::stable_sort(keys.begin(), keys.end(), key_less);
    b->plan_stale = false;
    b->device_plan = false;
    b->plan_epoch = ctx->plan_epoch;
    if (!active)
    {
        // the (variant, graph) runs of the whole batch: what a plan made from the device's per-run counts of ACTIVE reads starts
        // from (pg_batch_retire_mapped / pg_batch_ensure_plan)
        b->groups.clear();
        b->h_group_of_read.assign(n_reads, PG_NONE);
        b->has_general_reads = !b->gen_idx.empty();
        b->cascade_uploaded = false;
        b->full_plan_ready = false;
        for (size_t p0 = 0; p0 < keys.size();)
        {
            size_t q0 = p0;
            PgReadGroup g{ keys[p0].c, keys[p0].graph, (uint32_t)p0, 0, 0 };
            while (q0 < keys.size() && keys[q0].c == g.C && keys[q0].graph == g.graph)
            {
                b->h_group_of_read[keys[q0].idx] = (uint32_t)b->groups.size();
                g.sum_len += base_off[keys[q0].idx + 1] - base_off[keys[q0].idx];
                ++q0;
            }
            g.n_reads = (uint32_t)(q0 - p0);
            b->groups.push_back(g);
            p0 = q0;
        }
    }

    // ---- work items (pairs: forward graph, reversed graph) + chunk plan -----------------------------
    std::vector<PgWorkItem> items;
    items.reserve(keys.size() / 2 + 16);
    Chunk cur{};
    bool open = false;
    b->max_ws = 0;
    std::vector<uint32_t> pair_steps;  // pipeline steps of every item pair (its work)
    std::vector<uint32_t> order;
    std::vector<PgWorkItem> sorted;
    auto close_chunk = [&]() {
        if (open)
        {
            cur.pair_end = (uint32_t)(items.size() / 2);
            // Longest first: a launch ends with a tail in which the chip drains, and the tail is as long as the last wavefront to
            // start lives -- a 3 000-column graph's lives five times as long as a 600-column one's.  Pairs of one graph have the
            // same length and stay together (their graph's tables stay hot); regions keep the offsets they were given.
            const uint32_t pb = cur.pair_begin, pe = cur.pair_end;
            bool mixed = false;
            for (uint32_t i = pb + 1; i < pe && !mixed; ++i)
                mixed = pair_steps[i] != pair_steps[pb];
            if (mixed)
            {
                order.resize(pe - pb);
                for (uint32_t i = 0; i < pe - pb; ++i)
                    order[i] = pb + i;
                std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return pair_steps[x] > pair_steps[y]; });
                sorted.resize(2 * (size_t)(pe - pb));
                for (uint32_t i = 0; i < pe - pb; ++i)
                {
                    sorted[2 * (size_t)i] = items[2 * (size_t)order[i]];
                    sorted[2 * (size_t)i + 1] = items[2 * (size_t)order[i] + 1];
                }
                std::copy(sorted.begin(), sorted.end(), items.begin() + 2 * (size_t)pb);
            }
            b->chunks.push_back(cur);
            b->max_ws = std::max(b->max_ws, cur.ws_bytes);
            open = false;
        }
    };
    auto pair_need = [&](int C, const HostGraph& hg) { return pair_need_of(C, hg); };
    // EQUAL chunks: every fill launch ends with a tail in which the chip drains, and a chunk that holds what was left over pays
    // that tail on a fraction of the work (1 M config-2 reads at 128 GiB went out as 2 full chunks and a small one: launches of
    // 20.8 / 20.8 / 10.1 ms, profiles/r03_kernel_stats.csv).  The bytes of each variant's run are counted first and cut into
    // the smallest number of chunks that fit half the budget, all of (nearly) the same size.
    std::vector<uint64_t> chunk_target(PG_VAR_WIDE + 33, 0);
    {
        std::vector<uint64_t> total(chunk_target.size(), 0), largest(chunk_target.size(), 0);
        size_t p0 = 0;
        while (p0 < keys.size())
        {
            size_t q0 = p0;
            while (q0 < keys.size() && q0 - p0 < PG_GROUPS && keys[q0].c == keys[p0].c && keys[q0].graph == keys[p0].graph)
                ++q0;
            const uint64_t need = pair_need((int)keys[p0].c, G->host[keys[p0].graph]).need;
            total[keys[p0].c] += need;
            largest[keys[p0].c] = std::max(largest[keys[p0].c], need);
            p0 = q0;
        }
        const uint64_t cap = packed_limit / ctx->regions();
        for (size_t c = 0; c < total.size(); ++c)
        {
            if (!total[c])
                continue;
            if (largest[c] >= cap)
            {
                chunk_target[c] = cap;  // (a pair beyond the budget is refused below)
                continue;
            }
            // a chunk closes BEFORE the pair that would exceed its target, so the target carries one pair of slack:
            // total / n + largest <= cap
            const uint64_t room = cap - largest[c];
            const uint64_t n_chunks = std::max<uint64_t>(1, (total[c] + room - 1) / room);
            chunk_target[c] = std::min(cap, (total[c] + n_chunks - 1) / n_chunks + largest[c]);
        }
    }
    size_t p = 0;
    while (p < keys.size())
    {
        size_t q = p;
        while (q < keys.size() && q - p < PG_GROUPS && keys[q].c == keys[p].c && keys[q].graph == keys[p].graph)
            ++q;
        const int C = (int)keys[p].c;
        const HostGraph& hg = G->host[keys[p].graph];
        const PairNeed pn = pair_need(C, hg);
        const uint64_t nsteps = pn.nsteps, trace_bytes = pn.trace_bytes, seed_bytes = pn.seed_bytes, need = pn.need;
        if (need > packed_limit / ctx->regions())
            return fail(ctx, PG_ERR_UNSUPPORTED, b->gen_reserve ? "workspace limit too small for one wavefront of this graph beside the batch's general-path reads"
                                                               : "workspace limit too small for one wavefront of this graph");
        if (open && (cur.C != C || cur.ws_bytes + need > chunk_target[C]))
            close_chunk();
        if (!open)
        {
            cur = Chunk{};
            cur.C = C;
            cur.pair_begin = (uint32_t)(items.size() / 2);
            open = true;
        }
        PgWorkItem fw{}, rv{};
        fw.graph = rv.graph = keys[p].graph;
        fw.dir = 0;
        rv.dir = 1;
        for (int gI = 0; gI < PG_GROUPS; ++gI)
        {
            const uint32_t r = (p + gI < q) ? keys[p + gI].idx : PG_NONE;
            fw.read[gI] = rv.read[gI] = r;
            if (r != PG_NONE)
            {
                const uint64_t L = base_off[r + 1] - base_off[r];
                cur.fills += 4;
                cur.cells += 4 * L * hg.ncols;
            }
        }
        fw.trace_off = cur.ws_bytes;
        fw.seed_off = cur.ws_bytes + trace_bytes;
        rv.trace_off = cur.ws_bytes + trace_bytes + 2 * seed_bytes;
        rv.seed_off = cur.ws_bytes + trace_bytes + seed_bytes;
        cur.ws_bytes += need;
        cur.trace_bytes += nsteps * 64 * pg_trace_lane_bytes(C);
        cur.max_nodes = std::max(cur.max_nodes, hg.n_nodes);
        items.push_back(fw);
        items.push_back(rv);
        pair_steps.push_back((uint32_t)nsteps);
        p = q;
    }
    close_chunk();
    b->n_pairs = (uint32_t)(items.size() / 2);
    if (items.size() > b->cap_items)
    {
        HIP_TRY(ctx, pg_batch_wait(ctx, b));  // kernels of an earlier use of this batch may still read them
        (void)pg_dev_free(b->d_items);
        (void)pg_dev_free(b->d_fillsum);
        b->cap_items = std::max<size_t>(items.size(), 2);
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_items, b->cap_items * sizeof(PgWorkItem)));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_fillsum, b->cap_items * PG_GROUPS * 2 * sizeof(PgFillSummary)));
    }
    if (!items.empty())
        HIP_TRY(ctx, hipMemcpyAsync(b->d_items, items.data(), items.size() * sizeof(PgWorkItem), hipMemcpyHostToDevice, cs));
    HIP_TRY(ctx, pg_wait_stream(b, cs));  // `items` goes out of scope
    return PG_OK;
}

// Workspace / CIGAR scratch owned by the ctx and shared by its batches: grown when a batch that needs more reaches
// pg_batch_align (never while planning, so that a batch can be uploaded while another one is on the device).
static pg_status ensure_ctx_workspace(pg_ctx* ctx, const pg_batch* b)
{
    // Two (three with two fill streams) regions, used in turn by the chunks of all batches (ctx->chunk_seq): the traceback of a
    // chunk overlaps the fill of the next chunk -- of the same batch or of the next batch.
    const uint64_t need = (uint64_t)ctx->regions() * b->max_ws;
    if (need > ctx->ws_cap)
    {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_fill2));
        if (ctx->stream_lean)
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_lean));
    if (ctx->stream_lean)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_lean));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
        if (ctx->workspace)
            HIP_TRY(ctx, hipFree(ctx->workspace));
        ctx->workspace = nullptr;
        ctx->ws_cap = 0;
        for (auto& e : ctx->region_free)
            e = nullptr;  // the compute streams are idle: nothing reads the old regions
        // batches of one workflow differ by a few percent: one eighth of headroom spares the next, slightly larger one a
        // second multi-GiB allocation (each costs up to a second)
        uint64_t want = std::min<uint64_t>(need + need / 8, std::max<uint64_t>(ctx->ws_limit - std::min(ctx->ws_limit, b->gen_reserve), need));
        want = align_up(want, 512);
        if (hipMalloc((void**)&ctx->workspace, want) != hipSuccess)
        {
            (void)hipGetLastError();
            pg_dev_cache_release();  // idle blocks of dropped graph sets / batches may be what is in the way
            want = align_up(need, 512);
            HIP_TRY(ctx, hipMalloc((void**)&ctx->workspace, want));
        }
        ctx->ws_cap = want;
    }
    return PG_OK;
}

// The general path of pg_batch_align: the reads plan_items set aside (longer than PG_MAX_READ_LEN, or on a graph of more than
// 65 519 columns) go through pg_general.hip's one-thread-per-fill kernels on the second stream, behind the chunk pipeline, in
// groups whose H matrices fit the workspace budget.  Slow on purpose -- it exists so that such a read, and its site, stays in
// the run (the reference has no such bounds: gssw.c:527-786, GraphAligner.cpp:110-167).
static pg_status run_general(pg_ctx* ctx, pg_batch* b, uint32_t flags)
{
    const pg_graphs* G = b->graphs;
    const size_t n = b->gen_idx.size();
    if (n > b->cap_gen)
    {
        HIP_TRY(ctx, pg_batch_wait(ctx, b));
        (void)pg_dev_free(b->d_gen_reads);
        (void)pg_dev_free(b->d_gen_fsum);
        b->d_gen_reads = nullptr;
        b->d_gen_fsum = nullptr;
        b->cap_gen = n;
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_gen_reads, n * sizeof(PgGenRead)));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_gen_fsum, n * 4 * sizeof(PgFillSummary)));
    }
    // h_gen_reads is this batch's: its earlier use ended with the batch's last stage (the copy out of it was queued before that
    // stage's end event).  The general workspace is shared by the batches of the context, but only ever touched by launches on
    // the second stream, which run in order -- no host wait for the stream itself.
    HIP_TRY(ctx, pg_batch_wait(ctx, b));
    b->h_gen_reads.assign(n, PgGenRead{});
    // The general workspace is the share of the context's budget plan_items set aside for it (gen_reserve: the packed chunks were
    // cut to what is left, so both together stay within pg_ctx_set_workspace_bytes); groups are cut at that share.
    const uint64_t gen_limit = std::max<uint64_t>(b->gen_reserve, 1);
    // A launch's dynamic LDS is sized by its longest read (5 bytes per base, every block of the launch): reads are grouped by
    // length class, so that 150-base reads on a wide graph do not run with the two-blocks-per-CU occupancy of a 16 000-base one
    auto length_class = [&](uint32_t r) {
        const uint32_t L = b->h_base_off[r + 1] - b->h_base_off[r];
        uint32_t c = 0;
        while ((512u << c) < L)
            ++c;
        return c;  // <= 512, <= 1024, ... <= 16 384 bases
    };
    std::stable_sort(b->gen_idx.begin(), b->gen_idx.end(), [&](uint32_t x, uint32_t y) { return length_class(x) < length_class(y); });
    std::vector<std::pair<size_t, size_t>> groups;
    uint64_t cur = 0, largest = 0;
    size_t begin = 0;
    for (size_t i = 0; i < n; ++i)
    {
        const uint32_t r = b->gen_idx[i], g = b->h_graph_of_read[r];
        const uint64_t L = b->h_base_off[r + 1] - b->h_base_off[r];
        const HostGraph& hg = G->host[g];
        const uint64_t need = pg_gen_read_bytes(L, hg.ncols, hg.n_nodes);
        if (need > ctx->ws_limit)
            return fail(ctx, PG_ERR_UNSUPPORTED, "workspace limit too small for one read of the general path (read length x graph columns)");
        if (i > begin && (cur + need > gen_limit || length_class(r) != length_class(b->gen_idx[i - 1])))
        {
            groups.emplace_back(begin, i);
            begin = i;
            cur = 0;
        }
        PgGenRead& gr = b->h_gen_reads[i];
        gr.read = r;
        gr.graph = g;
        gr.h_off = cur;
        gr.seed_off = gr.h_off + pg_gen_align8(2 * (uint64_t)hg.ncols * L * 2);
        gr.col_off = gr.seed_off + pg_gen_align8(4 * 2 * (uint64_t)hg.n_nodes * L * 2);
        gr.node_off = gr.col_off + pg_gen_align8(4 * 2 * L * 2);
        gr.ops_off = gr.node_off + pg_gen_align8(4 * (uint64_t)hg.n_nodes * PG_GEN_NODE_BYTES);
        cur += need;
        largest = std::max(largest, cur);
    }
    groups.emplace_back(begin, n);
    if (largest > ctx->gen_ws_cap)
    {
        if (ctx->gen_ws)
            HIP_TRY(ctx, hipFree(ctx->gen_ws));
        ctx->gen_ws = nullptr;
        ctx->gen_ws_cap = 0;
        const uint64_t want = (largest + 511) & ~(uint64_t)511;
        if (hipMalloc((void**)&ctx->gen_ws, want) != hipSuccess)
        {
            (void)hipGetLastError();
            pg_dev_cache_release();
            HIP_TRY(ctx, hipMalloc((void**)&ctx->gen_ws, want));
        }
        ctx->gen_ws_cap = want;
    }
    HIP_TRY(ctx, hipMemcpyAsync(b->d_gen_reads, b->h_gen_reads.data(), n * sizeof(PgGenRead), hipMemcpyHostToDevice, ctx->stream2));
    for (auto const& grp : groups)
    {
        PgGenArgs ga{};
        ga.reads = b->d_gen_reads + grp.first;
        ga.n = (uint32_t)(grp.second - grp.first);
        ga.flags = flags;
        ga.max_len = 0;
        for (size_t i = grp.first; i < grp.second; ++i)
        {
            const uint32_t r = b->gen_idx[i];
            ga.max_len = std::max<uint32_t>(ga.max_len, b->h_base_off[r + 1] - b->h_base_off[r]);
        }
        ga.graphs = G->d_graphs;
        ga.nodes = G->d_nodes;
        ga.preds = G->d_preds;
        ga.seqchars = G->d_seqchars;
        ga.base_off = b->d_base_off;
        ga.bases = b->d_bases;
        ga.ws = ctx->gen_ws;
        ga.fsum = b->d_gen_fsum + grp.first * 4;
        ga.results = b->d_results;
        ga.ops = b->d_ops;
        ga.ops_counter = b->d_ops_counter;
        HIP_TRY(ctx, pg_launch_general(ga, ctx->stream2));
    }
    return PG_OK;
}

extern "C" pg_status pg_batch_upload(
    pg_ctx* ctx, pg_batch* b, const pg_graphs* G, uint32_t n_reads, const uint32_t* graph_of_read,
    const uint32_t* base_off, const char* bases)
{
    if (!ctx || !b || !G || (n_reads && (!graph_of_read || !base_off || !bases)))
        return fail(ctx, PG_ERR_INVALID, "pg_batch_upload: null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // Uploads run on the copy stream: the upload of the NEXT batch overlaps the kernels of the current one.  Only an
    // earlier use of THIS batch object has to be finished before its buffers are overwritten.
    hipStream_t cs = ctx->stream_copy;
    if (b->busy_recorded)
        HIP_TRY(ctx, hipStreamWaitEvent(cs, b->ev_busy, 0));
    if (!b->parked_blocks.empty())  // what the stages of the last use outgrew (the batch is idle by now: the wait returns at once)
    {
        HIP_TRY(ctx, pg_batch_wait(ctx, b));
        for (void* p : b->parked_blocks)
            (void)pg_dev_free(p);
        b->parked_blocks.clear();
    }
    b->graphs = G;
    b->n_reads = n_reads;
    b->has_skipped = false;
    b->seed_chain = false;
    b->fragments_set = false;
    b->has_active = false;
    b->h_counters_valid = false;
    b->label_ext_words = 0;  // the label sets of an earlier count belong to the earlier reads
    b->label_ext_reads = 0;
    b->host_template.clear();  // built only when a read is skipped (rare)
    uint64_t ops_total = 0;
    for (uint32_t i = 0; i < n_reads; ++i)
    {
        if (base_off[i + 1] < base_off[i])
            return fail(ctx, PG_ERR_INVALID, "base_off must be non-decreasing");
        const uint32_t L = base_off[i + 1] - base_off[i];
        if (graph_of_read[i] >= G->n_graphs)
            return fail(ctx, PG_ERR_INVALID, "graph_of_read out of range");
        if (L > PG_GEN_MAX_READ_LEN)
            return fail(ctx, PG_ERR_UNSUPPORTED, "read longer than 16000 bp");
        if (L == 0)
        {
            // grm::sequentialAlignReads skips reads without bases (Align.cpp:74-77)
            if (!b->has_skipped)
                b->host_template.assign(n_reads, pg_result{});
            b->host_template[i].status = 1;
            b->has_skipped = true;
            continue;
        }
        ops_total += (L > PG_MAX_READ_LEN || G->host[graph_of_read[i]].general_only) ? pg_gen_ops_cap(L) : pg_ops_cap(pg_variant_of(L));
    }
    b->h_graph_of_read.assign(graph_of_read, graph_of_read + n_reads);
    b->h_base_off.assign(base_off, base_off + (n_reads ? n_reads + 1 : 0));
    if (!n_reads)
        b->h_base_off.assign(1, 0);

    // ---- device buffers ----------------------------------------------------------------------------
    const size_t n_bases = n_reads ? base_off[n_reads] : 0;
    if (n_reads + 1 > b->cap_reads || n_bases > b->cap_bases || ops_total > b->ops_cap)
    {
        HIP_TRY(ctx, pg_batch_wait(ctx, b));
        HIP_TRY(ctx, hipStreamSynchronize(cs));
        batch_free_device(b);
        b->cap_reads = n_reads + 1;
        b->cap_bases = std::max<size_t>(n_bases, 1);
        b->ops_cap = std::max<uint64_t>(ops_total, 1);
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_base_off, b->cap_reads * sizeof(uint32_t)));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_bases, b->cap_bases));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_results, std::max<size_t>(n_reads, 1) * sizeof(pg_result)));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_ops, b->ops_cap * sizeof(pg_op)));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_ops_counter, sizeof(unsigned long long)));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_graph_of_read, b->cap_reads * sizeof(uint32_t)));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_path_flags, b->cap_reads));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_active, b->cap_reads));
    }
    if (n_reads)
    {
        HIP_TRY(ctx, hipMemcpyAsync(b->d_base_off, base_off, (n_reads + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
        HIP_TRY(ctx, hipMemcpyAsync(b->d_graph_of_read, graph_of_read, n_reads * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
        if (n_bases)
            HIP_TRY(ctx, hipMemcpyAsync(b->d_bases, bases, n_bases, hipMemcpyHostToDevice, cs));
        // the template is all zero except for reads the device never sees (empty reads: status 1)
        if (b->has_skipped)
            HIP_TRY(ctx, hipMemcpyAsync(b->d_results, b->host_template.data(), n_reads * sizeof(pg_result), hipMemcpyHostToDevice, cs));
        else
            HIP_TRY(ctx, hipMemsetAsync(b->d_results, 0, n_reads * sizeof(pg_result), cs));
        HIP_TRY(ctx, hipMemsetAsync(b->d_path_flags, 0, n_reads, cs));
    }
    HIP_TRY(ctx, hipMemsetAsync(b->d_ops_counter, 0, sizeof(unsigned long long), cs));
    b->ops_counter_fresh = true;  // (the first stage behind this upload need not zero it again)
    b->cascade_recorded = false;
    const pg_status st = plan_items(ctx, b, nullptr, cs);
    if (st != PG_OK)
        return st;
    HIP_TRY(ctx, hipEventRecord(b->ev_upload, cs));
    b->upload_recorded = true;
    return PG_OK;
}

extern "C" pg_status pg_batch_set_active(pg_ctx* ctx, pg_batch* b, const uint8_t* active)
{
    if (!ctx || !b || !b->graphs)
        return fail(ctx, PG_ERR_INVALID, "pg_batch_set_active: batch not uploaded");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!active)
    {
        // back to every read: nothing to send, and the work items are re-made where the next stage that needs them runs
        // (pg_batch_ensure_plan: from the upload-time groups, on the device)
        b->has_active = false;
        b->plan_stale = true;
        return PG_OK;
    }
    HIP_TRY(ctx, pg_batch_wait(ctx, b));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    b->has_active = true;
    if (b->n_reads)
        HIP_TRY(ctx, hipMemcpyAsync(b->d_active, active, b->n_reads, hipMemcpyHostToDevice, ctx->stream));
    return plan_items(ctx, b, active, ctx->stream);
}

// ----------------------------------------------------------------------------------------------------------------------
// Device-resident hand-over between the stages of the cascade (CompositeAligner::alignRead, src/c++/lib/grm/
// CompositeAligner.cpp:78-176: a read the stage mapped and the filter accepted is done, everything else goes on).
// The decision is made where the flags are -- on the device -- and the next stage's work follows it without ANYTHING
// crossing to the host, not even a count:
//   * once per upload the host cuts the batch's FULL plan (every read active) into (group, chunk) segments -- a group is a run of
//     reads of one (read-length class, graph) -- and uploads them: O(groups), no device information needed;
//   * pg_batch_retire_mapped: active[i] &= !((stage flag & 1) && count-path status == MAPPED); the reads still active are listed
//     group by group (one atomic per (wavefront, group)); the work items are re-written over the full plan's pair slots: the
//     first ceil(count / 4) pairs of a group hold its active reads, the rest are EMPTY (read[0] == PG_NONE) -- the fill, klib
//     and traceback wavefronts of an empty pair return at once.  The launches keep the full plan's grids and chunk cuts (an empty
//     workgroup costs a dispatch, not a sweep).
// ----------------------------------------------------------------------------------------------------------------------
namespace
{
// (had_mask == 0: the batch has no mask yet = every read is active -- the kernel writes the first one instead of a memset before it;
//  group_count: the per-group counters of the list kernel behind this one, zeroed here instead of by a memset between the two.  Every
//  dispatch of a seed chain waits for a wavefront slot beside the fills: the chain is as long as it has dispatches.)
__global__ void pg_retire_kernel(uint32_t n, const uint8_t* __restrict__ stage_flags, const pg_read_support* __restrict__ sup, uint8_t* active,
                                 uint32_t had_mask, uint32_t* group_count, uint32_t n_groups)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_groups)
        group_count[i] = 0;
    if (i < n)
    {
        const bool was = had_mask ? active[i] != 0 : true;
        active[i] = (was && !((stage_flags[i] & 1u) && sup[i].status == 1)) ? 1 : 0;
    }
}

// pg_batch_retire_exact_matches: the reads whose gssw record the path stage's match forces (include/paragraph_amd.h)
__global__ void pg_retire_exact_kernel(uint32_t n, const uint8_t* __restrict__ stage_flags, pg_result* results, const uint32_t* __restrict__ base_off,
                                       const char* __restrict__ bases, uint8_t* active, uint32_t had_mask, uint32_t* group_count, uint32_t n_groups)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_groups)
        group_count[i] = 0;
    if (i >= n)
        return;
    const bool was = had_mask ? active[i] != 0 : true;
    bool forced = false;
    const uint8_t f = stage_flags[i];
    if (was && (f & 1u))
    {
        const pg_result r = results[i];
        const uint32_t off = base_off[i], L = base_off[i + 1] - off;
        forced = r.is_unique != 0 && L <= 250u && (r.returned_reverse == 0 || (f & PG_PATH_FLAG_FWD_ABSENT) != 0u);
        for (uint32_t c = 0; forced && c < L; ++c)
        {
            const uint32_t ch = (uint8_t)bases[off + c] & 0xDFu;  // (upper case)
            forced = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
        }
        if (forced)
        {
            // the record alignRead would have written: that of the gssw stage (the hosts treat PathAligner's differently:
            // PathAligner.cpp:121-129 leaves the qualities as they are, GraphAligner.cpp:372-376 reverses them)
            results[i].status = (uint16_t)(r.status & ~PG_STATUS_PATH_ALIGNER);
        }
    }
    active[i] = (was && !forced) ? 1 : 0;
}

__global__ void pg_group_list_kernel(
    uint32_t n, const uint8_t* __restrict__ active, const uint32_t* __restrict__ group_of_read, const uint32_t* __restrict__ group_base,
    uint32_t* group_count, uint32_t* list)
{
    // One atomic per (wavefront, group), not per read: reads arrive site after site, so the lanes of a wavefront mostly share
    // ONE group -- a million reads of one graph were a million atomics on one address (1.8 ms per 200 k reads,
    // profiles/r05_stage_counters.json); the lanes of a group take consecutive slots behind the leader's base.
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t g = PG_NONE;
    if (i < n && (!active || active[i]))  // (no mask: every read is active)
        g = group_of_read[i];
    const unsigned lane = threadIdx.x & 63u;
    unsigned long long todo = __ballot(g != PG_NONE);
    while (todo)
    {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t g0 = (uint32_t)__shfl((int)g, leader, 64);
        const unsigned long long same = __ballot(g == g0) & todo;
        uint32_t base = 0;
        if ((int)lane == leader)
            base = atomicAdd(&group_count[g0], (uint32_t)__popcll(same));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if ((same >> lane) & 1ull)
            list[group_base[g0] + base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = i;
        todo &= ~same;
    }
}

// pair slot p of the full plan: the group's active reads 4k .. 4k + 3 (k = the pair's rank in its group), PG_NONE beyond the
// group's count of active reads; workspace offsets as the full plan cut them
__global__ void pg_build_items_kernel(
    uint32_t n_pairs, uint32_t n_segments, const PgPlanSegment* __restrict__ seg, const uint32_t* __restrict__ list,
    const uint32_t* __restrict__ group_count, PgWorkItem* items)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs)
        return;
    uint32_t lo = 0, hi = n_segments;  // last segment whose pair_begin <= p
    while (hi - lo > 1)
    {
        const uint32_t mid = (lo + hi) >> 1;
        if (seg[mid].pair_begin <= p)
            lo = mid;
        else
            hi = mid;
    }
    const PgPlanSegment s = seg[lo];
    const uint32_t k = s.first_pair + (p - s.pair_begin);
    const uint32_t count = group_count[s.group];
    PgWorkItem fw{}, rv{};
    fw.graph = rv.graph = s.graph;
    fw.dir = 0;
    rv.dir = 1;
    for (int j = 0; j < PG_GROUPS; ++j)
    {
        const uint32_t slot = 4u * k + (uint32_t)j;
        fw.read[j] = rv.read[j] = slot < count ? list[s.list_base + slot] : PG_NONE;
    }
    const uint64_t base = s.ws_base + (uint64_t)(p - s.pair_begin) * s.need;
    fw.trace_off = base;
    fw.seed_off = base + s.trace_bytes;
    rv.trace_off = base + s.trace_bytes + 2 * s.seed_bytes;
    rv.seed_off = base + s.trace_bytes + s.seed_bytes;
    items[2 * (size_t)p] = fw;
    items[2 * (size_t)p + 1] = rv;
}

// ---- the lean gssw stage: pick behind the reversed-graph fills ---------------------------------------------------------------------
// init: the instance items of a chunk's pair slots -- graph and workspace regions of the forward work item in the same slot (an
// instance item of a run always takes a slot of that run: same graph, same region sizes), no instances yet
__global__ void pg_lean_init_kernel(PgLeanBuildArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_pairs)
        return;
    const uint32_t p = a.pair_begin + i;
    const PgWorkItem fw = a.items[2 * (size_t)p];
    PgInstItem it;
    it.graph = fw.graph;
    it.pad = 0;
    for (int h = 0; h < 2; ++h)
        for (int g = 0; g < PG_GROUPS; ++g)
            it.inst[h][g] = PG_NONE;
    it.trace_off = fw.trace_off;
    it.seed_off = fw.seed_off;
    a.inst[p] = it;
    a.extra[p] = 0;
    if (i == 0)
        *a.ucount = 0;
}

// build: per read of the pair, X = the strand whose reversed-graph fill scored higher (the forward strand on a tie) goes to the instance
// item (run start + rank / 2), half rank & 1, the read's own group; the other strand Y gets a forward fill too where the reversed-graph
// fills already say "X is not unique, Y may be" (pg_trace.hip, LEAN) -- in the slots behind the run's X items, eight to an item.
__global__ void pg_lean_build_kernel(PgLeanBuildArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_pairs)
        return;
    const uint32_t p = a.pair_begin + i;
    uint32_t lo = 0, hi = a.n_segments;  // last segment whose pair_begin <= p
    while (hi - lo > 1)
    {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.segments[mid].pair_begin <= p)
            lo = mid;
        else
            hi = mid;
    }
    const PgPlanSegment s = a.segments[lo];
    const uint32_t rs = s.pair_begin;
    const uint32_t count = a.group_count[s.group];
    // pairs of this run that hold reads (a group's active reads come first)
    uint32_t ne = 0;
    if (count > 4u * s.first_pair)
    {
        ne = (count - 4u * s.first_pair + 3u) / 4u;
        ne = ne < s.n_pairs ? ne : s.n_pairs;
    }
    const PgWorkItem fw = a.items[2 * (size_t)p];
    const uint32_t q = rs + (p - rs) / 2u, h = (p - rs) & 1u;
    for (int g = 0; g < PG_GROUPS; ++g)
    {
        const uint32_t ridx = fw.read[g];
        if (ridx == PG_NONE)
            continue;
        const PgFillSummary* fsR = a.fillsum + ((size_t)(2 * p + 1) * PG_GROUPS + g) * 2;
        const int SA = fsR[0].score, SB = fsR[1].score;
        const int X = SA >= SB ? 0 : 1;
        const int mXr = fsR[X].multi, mYr = fsR[1 - X].multi;
        a.inst[q].inst[h][g] = ridx | (X ? PG_INST_RC : 0u);
        uint32_t yl = PG_NONE;
        if (mXr && !mYr)
        {
            // the run's instance slots, eight per pair slot: X of (pair rank r, group g) has 8 (r / 2) + 4 (r & 1) + g, i.e. the first
            // 4 ne; the others follow in the order they ask (an odd ne leaves them the second half of the last X item) -- at most
            // one per read, so they always fit; slots are handed out in order, so an item is empty iff its (half 0, group 0) is
            const uint32_t e = atomicAdd(&a.extra[rs], 1u);
            const uint32_t t = 4u * ne + e;
            const uint32_t qq = rs + t / 8u, within = t & 7u;
            a.inst[qq].inst[within >> 2][within & 3u] = ridx | (X ? 0u : PG_INST_RC);
            yl = (qq << 3) | ((within & 3u) << 1) | (within >> 2);
        }
        a.yloc[ridx] = yl;
    }
}
}  // namespace

hipError_t pg_launch_lean_build(const PgLeanBuildArgs& args, hipStream_t stream)
{
    if (!args.n_pairs)
        return hipSuccess;
    hipLaunchKernelGGL(pg_lean_init_kernel, dim3((args.n_pairs + 255) / 256), dim3(256), 0, stream, args);
    hipLaunchKernelGGL(pg_lean_build_kernel, dim3((args.n_pairs + 255) / 256), dim3(256), 0, stream, args);
    return hipGetLastError();
}

// The batch's full plan (every read active) as (group, chunk) segments on the device, and b->chunks = its chunks: made once per
// upload, on the host, from the groups' sizes alone -- plan_items' rule: equal chunks per variant, longest graph first in a chunk.
static pg_status cascade_full_plan(pg_ctx* ctx, pg_batch* b, hipStream_t stream)
{
    if (b->full_plan_ready)
        return PG_OK;
    const pg_graphs* G = b->graphs;
    const size_t n_groups = b->groups.size();
    const uint64_t packed_limit = ctx->ws_limit - b->gen_reserve;
    const uint64_t cap = packed_limit / ctx->regions();
    std::vector<uint64_t> chunk_target(PG_VAR_WIDE + 33, 0);
    {
        std::vector<uint64_t> total(chunk_target.size(), 0), largest(chunk_target.size(), 0);
        for (size_t g = 0; g < n_groups; ++g)
        {
            const uint64_t pairs = (b->groups[g].n_reads + PG_GROUPS - 1) / PG_GROUPS;
            const uint64_t need = pair_need_of((int)b->groups[g].C, G->host[b->groups[g].graph]).need;
            total[b->groups[g].C] += pairs * need;
            largest[b->groups[g].C] = std::max(largest[b->groups[g].C], need);
        }
        for (size_t c = 0; c < total.size(); ++c)
        {
            if (!total[c])
                continue;
            if (largest[c] >= cap)
            {
                chunk_target[c] = cap;
                continue;
            }
            const uint64_t room = cap - largest[c];
            const uint64_t n_chunks = std::max<uint64_t>(1, (total[c] + room - 1) / room);
            chunk_target[c] = std::min(cap, (total[c] + n_chunks - 1) / n_chunks + largest[c]);
        }
    }
    std::vector<PgPlanSegment>& segs = b->h_segments;
    segs.clear();
    b->full_chunks.clear();
    std::vector<uint64_t> seg_steps;
    Chunk cur{};
    bool open = false;
    size_t chunk_seg0 = 0;
    uint64_t max_ws = 0;
    uint32_t next_pair = 0;
    auto close_chunk = [&]() {
        if (!open)
            return;
        // longest graph first (what runs in a launch's tail is short); regions keep the offsets they were given
        std::vector<size_t> order(segs.size() - chunk_seg0);
        for (size_t i = 0; i < order.size(); ++i)
            order[i] = chunk_seg0 + i;
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return seg_steps[x] > seg_steps[y]; });
        std::vector<PgPlanSegment> sorted;
        std::vector<uint64_t> sorted_steps;
        uint32_t at = cur.pair_begin;
        for (size_t i : order)
        {
            PgPlanSegment sg = segs[i];
            sg.pair_begin = at;
            at += sg.n_pairs;
            sorted.push_back(sg);
            sorted_steps.push_back(seg_steps[i]);
        }
        std::copy(sorted.begin(), sorted.end(), segs.begin() + (std::ptrdiff_t)chunk_seg0);
        std::copy(sorted_steps.begin(), sorted_steps.end(), seg_steps.begin() + (std::ptrdiff_t)chunk_seg0);
        cur.pair_end = at;
        b->full_chunks.push_back(cur);
        max_ws = std::max(max_ws, cur.ws_bytes);
        open = false;
    };
    for (size_t g = 0; g < n_groups; ++g)
    {
        const PgReadGroup& grp = b->groups[g];
        uint32_t pairs_left = (grp.n_reads + PG_GROUPS - 1) / PG_GROUPS, first_pair = 0;
        const int C = (int)grp.C;
        const HostGraph& hg = G->host[grp.graph];
        const PairNeed pn = pair_need_of(C, hg);
        if (pn.need > cap)
            return fail(ctx, PG_ERR_UNSUPPORTED, "workspace limit too small for one wavefront of this graph");
        const uint64_t mean_len = grp.n_reads ? grp.sum_len / grp.n_reads : 0;
        while (pairs_left)
        {
            if (open && (cur.C != C || cur.ws_bytes + pn.need > chunk_target[C]))
                close_chunk();
            if (!open)
            {
                cur = Chunk{};
                cur.C = C;
                cur.pair_begin = next_pair;
                chunk_seg0 = segs.size();
                open = true;
            }
            const uint64_t fit = std::max<uint64_t>(1, (chunk_target[C] - cur.ws_bytes) / pn.need);
            const uint32_t take = (uint32_t)std::min<uint64_t>(pairs_left, fit);
            PgPlanSegment sg{};
            sg.pair_begin = next_pair;  // (re-assigned when the chunk closes)
            sg.n_pairs = take;
            sg.first_pair = first_pair;
            sg.graph = grp.graph;
            sg.list_base = grp.list_base;
            sg.group = (uint32_t)g;
            sg.ws_base = cur.ws_bytes;
            sg.need = pn.need;
            sg.trace_bytes = pn.trace_bytes;
            sg.seed_bytes = pn.seed_bytes;
            segs.push_back(sg);
            seg_steps.push_back(pn.nsteps);
            const uint64_t reads_here = std::min<uint64_t>((uint64_t)take * PG_GROUPS, grp.n_reads - (uint64_t)first_pair * PG_GROUPS);
            cur.ws_bytes += (uint64_t)take * pn.need;
            cur.trace_bytes += (uint64_t)take * pn.nsteps * 64 * pg_trace_lane_bytes(C);
            cur.max_nodes = std::max(cur.max_nodes, hg.n_nodes);
            cur.fills += 4 * reads_here;  // (of the full plan: what a launch is sized for, not what is left active; timing figures only)
            cur.cells += 4 * reads_here * mean_len * hg.ncols;
            next_pair += take;
            first_pair += take;
            pairs_left -= take;
        }
    }
    close_chunk();
    b->full_pairs = next_pair;
    b->full_max_ws = max_ws;
    if (segs.size() > b->cap_segments)
    {
        b->park(b->d_segments);  // (no wait under the caller's device lock: pg_internal.h, parked_blocks)
        b->d_segments = nullptr;
        b->cap_segments = segs.size() + segs.size() / 4 + 16;
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_segments, b->cap_segments * sizeof(PgPlanSegment)));
    }
    // (h_segments is the batch's: it stays until the next upload, the copy may run on)
    if (!segs.empty())
        HIP_TRY(ctx, hipMemcpyAsync(b->d_segments, segs.data(), segs.size() * sizeof(PgPlanSegment), hipMemcpyHostToDevice, stream));
    b->full_plan_ready = true;
    return PG_OK;
}

// Lists of the active reads + the work items over the full plan's slots, all on `stream` (d_active: NULL = every read).
// the cascade's device tables (groups, lists, plan segments) for this upload: allocated / uploaded / cut on first use
static pg_status cascade_prepare(pg_ctx* ctx, pg_batch* b, hipStream_t stream)
{
    const uint32_t n = b->n_reads;
    const size_t n_groups = b->groups.size();
    if (!n || !n_groups)
        return PG_OK;
    if (n_groups > b->cap_groups || n > b->cap_cascade_reads)
    {
        b->park(b->d_group_of_read2);
        b->park(b->d_group_base);
        b->park(b->d_group_count);
        b->park(b->d_active_list);
        b->d_group_of_read2 = b->d_group_base = b->d_group_count = b->d_active_list = nullptr;
        b->cap_groups = n_groups;
        b->cap_cascade_reads = n;
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_group_of_read2, (size_t)n * sizeof(uint32_t)));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_active_list, (size_t)n * sizeof(uint32_t)));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_group_base, n_groups * sizeof(uint32_t)));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_group_count, n_groups * sizeof(uint32_t)));
        b->cascade_uploaded = false;
    }
    if (!b->cascade_uploaded)
    {
        b->h_group_base.resize(n_groups);
        for (size_t g = 0; g < n_groups; ++g)
            b->h_group_base[g] = b->groups[g].list_base;
        HIP_TRY(ctx, hipMemcpyAsync(b->d_group_of_read2, b->h_group_of_read.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        HIP_TRY(ctx, hipMemcpyAsync(b->d_group_base, b->h_group_base.data(), n_groups * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        b->cascade_uploaded = true;
    }
    return cascade_full_plan(ctx, b, stream);
}

pg_status pg_cascade_prepare_early(pg_ctx* ctx, pg_batch* b)
{
    if (b->has_general_reads || b->cascade_recorded || !b->n_reads || b->groups.empty())
        return PG_OK;
    const pg_status ps = cascade_prepare(ctx, b, ctx->stream_copy);
    if (ps != PG_OK)
        return ps;
    if (!b->ev_cascade)
        HIP_TRY(ctx, hipEventCreateWithFlags(&b->ev_cascade, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(b->ev_cascade, ctx->stream_copy));
    b->cascade_recorded = true;
    return PG_OK;
}

static pg_status cascade_rebuild_items(pg_ctx* ctx, pg_batch* b, hipStream_t stream, const uint8_t* d_active, bool group_counts_zeroed = false)
{
    const uint32_t n = b->n_reads;
    const size_t n_groups = b->groups.size();
    b->chunks.clear();
    b->n_pairs = 0;
    if (!n || !n_groups)
    {
        b->plan_stale = false;  // nothing the packed kernels could run
        return PG_OK;
    }
    const pg_status fp = cascade_prepare(ctx, b, stream);
    if (fp != PG_OK)
        return fp;
    if (b->cascade_recorded)
        HIP_TRY(ctx, hipStreamWaitEvent(stream, b->ev_cascade, 0));
    if (!group_counts_zeroed)
        HIP_TRY(ctx, hipMemsetAsync(b->d_group_count, 0, n_groups * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(pg_group_list_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, d_active, b->d_group_of_read2, b->d_group_base,
                       b->d_group_count, b->d_active_list);
    HIP_TRY(ctx, hipGetLastError());
    if (b->full_pairs)
    {
        hipLaunchKernelGGL(pg_build_items_kernel, dim3((b->full_pairs + 255) / 256), dim3(256), 0, stream, b->full_pairs, (uint32_t)b->h_segments.size(),
                           b->d_segments, b->d_active_list, b->d_group_count, b->d_items);
        HIP_TRY(ctx, hipGetLastError());
    }
    b->chunks = b->full_chunks;
    b->n_pairs = b->full_pairs;
    b->max_ws = b->full_max_ws;
    b->plan_stale = false;
    b->device_plan = true;
    return PG_OK;
}

static pg_status retire_reads(pg_ctx* ctx, pg_batch* b, bool exact);

extern "C" pg_status pg_batch_retire_mapped(pg_ctx* ctx, pg_batch* b)
{
    if (!ctx || !b || !b->graphs || !b->d_support || !b->d_path_flags)
        return fail(ctx, PG_ERR_INVALID, "pg_batch_retire_mapped: a seed stage and pg_batch_count must have run");
    return retire_reads(ctx, b, false);
}

extern "C" pg_status pg_batch_retire_exact_matches(pg_ctx* ctx, pg_batch* b)
{
    if (!ctx || !b || !b->graphs || !b->d_path_flags || !b->d_results || !b->seed_chain)
        return fail(ctx, PG_ERR_INVALID, "pg_batch_retire_exact_matches: pg_batch_path_align must have run last");
    return retire_reads(ctx, b, true);
}

static pg_status retire_reads(pg_ctx* ctx, pg_batch* b, bool exact)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // behind the count pass, on its stream (the seed stream behind a path stage): the next stage waits for the batch's event as
    // always
    hipStream_t cs = b->seed_chain ? b->seed_stream : ctx->stream2;
    HIP_TRY(ctx, pg_stage_begin_on(ctx, b, cs));
    // every way out from here on records the end of the stage on its stream (an error return that skipped it would let the
    // caller free blocks the kernels queued so far still read); the mask only counts as written once the kernel that writes it
    // is queued
    struct StageEnd
    {
        pg_ctx* c;
        pg_batch* b;
        hipStream_t s;
        bool done;
        ~StageEnd()
        {
            if (!done)
                (void)pg_stage_end_on(c, b, s);
        }
    } stage_end{ ctx, b, cs, false };
    const uint32_t had_mask = b->has_active ? 1u : 0u;
    if (!b->n_reads)
    {
        b->has_active = true;
        b->plan_stale = true;
    }
    if (b->n_reads)
    {
        // (a batch with general-path reads plans on the host from the downloaded flags: pg_batch_ensure_plan)
        const bool rebuild = !b->has_general_reads;
        uint32_t n_groups = 0;
        if (rebuild)
        {
            const pg_status ps = cascade_prepare(ctx, b, cs);
            if (ps != PG_OK)
                return ps;
            n_groups = (uint32_t)b->groups.size();
        }
        const uint32_t threads = std::max(b->n_reads, n_groups);
        if (exact)
            hipLaunchKernelGGL(pg_retire_exact_kernel, dim3((threads + 255) / 256), dim3(256), 0, cs, b->n_reads, b->d_path_flags, b->d_results,
                               b->d_base_off, b->d_bases, b->d_active, had_mask, n_groups ? b->d_group_count : nullptr, n_groups);
        else
            hipLaunchKernelGGL(pg_retire_kernel, dim3((threads + 255) / 256), dim3(256), 0, cs, b->n_reads, b->d_path_flags, b->d_support, b->d_active,
                               had_mask, n_groups ? b->d_group_count : nullptr, n_groups);
        HIP_TRY(ctx, hipGetLastError());
        b->has_active = true;
        b->plan_stale = true;
        if (rebuild)
        {
            const pg_status rs = cascade_rebuild_items(ctx, b, cs, b->d_active, true);
            if (rs != PG_OK)
                return rs;
        }
    }
    stage_end.done = true;
    HIP_TRY(ctx, pg_stage_end_on(ctx, b, cs));
    return PG_OK;
}

pg_status pg_batch_ensure_plan(pg_ctx* ctx, pg_batch* b, hipStream_t stream)
{
    if (!b->plan_stale)
        return PG_OK;
    if (b->has_general_reads)
    {
        // reads of the general path are planned one by one on the host: this (rare) batch fetches the flags and plans from them
        if (!b->has_active)
            return plan_items(ctx, b, nullptr, stream);
        std::vector<uint8_t> active(b->n_reads);
        HIP_TRY(ctx, hipMemcpyAsync(active.data(), b->d_active, b->n_reads, hipMemcpyDeviceToHost, stream));
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        return plan_items(ctx, b, active.data(), stream);
    }
    // pg_batch_set_active(NULL) after a hand-over: every read again, on the stage's own stream
    b->gen_idx.clear();
    return cascade_rebuild_items(ctx, b, stream, b->has_active ? b->d_active : nullptr);
}

extern "C" pg_status pg_batch_align(pg_ctx* ctx, pg_batch* b, uint32_t flags)
{
    PG_TIMED("pg_batch_align (whole call)");
    if (!ctx || !b || !b->graphs)
        return fail(ctx, PG_ERR_INVALID, "pg_batch_align: batch not uploaded");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        PG_TIMED("recycle_done_sync_events");
        recycle_done_sync_events(ctx);
    }
    const pg_graphs* G = b->graphs;
    if (b->plan_epoch != ctx->plan_epoch)
        return fail(ctx, PG_ERR_INVALID, "pg_batch_align: the batch was uploaded before pg_ctx_set_fill_streams changed the number of "
                                         "workspace regions (its chunks were cut for the old ones): upload it again");
    {
        const auto t0 = std::chrono::steady_clock::now();
        const uint64_t cap0 = ctx->ws_cap;
        const pg_status ws = ensure_ctx_workspace(ctx, b);
        if (ws != PG_OK)
            return ws;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ctx->ws_cap != cap0 && std::getenv("PG_BATCH_TIMING"))
            fprintf(stderr, "[pg_batch_align] workspace %.2f -> %.2f GiB (%zu chunk(s) of up to %.2f GiB) in %.1f ms\n",
                    cap0 / 1073741824.0, ctx->ws_cap / 1073741824.0, b->chunks.size(), b->max_ws / 1073741824.0, ms);
    }
    b->h_counters_valid = false;
    b->seed_chain = false;
    HIP_TRY(ctx, pg_stage_begin(ctx, b));
    // Every way out of this call from here on records the end of the stage: pg_graphs_destroy / pg_dev_free / the next stage of
    // this batch trust those events.  An error return that skipped it (a failed launch, the general path) would let the caller
    // hand the batch's and the graph set's device blocks to another lane while the kernels queued so far still read them.
    // The stage ends on the second stream, which is made to wait for whatever the main stream has been given.
    struct StageEnd
    {
        pg_ctx* c;
        pg_batch* b;
        bool done;
        ~StageEnd()
        {
            if (done)
                return;
            hipEvent_t e;
            for (hipStream_t fs : { c->stream, c->stream_fill2, c->stream_lean })
                if (fs && get_sync_event(c, &e) == hipSuccess)
                {
                    if (hipEventRecord(e, fs) == hipSuccess)
                        (void)hipStreamWaitEvent(c->stream2, e, 0);
                    c->sync_events_in_flight.push_back(e);
                }
            (void)pg_stage_end_on(c, b, c->stream2);
        }
    } stage_end{ ctx, b, false };
    {
        // (inside the guarded region: a plan re-made on the device has queued kernels by the time a later step of it fails)
        const pg_status ps = pg_batch_ensure_plan(ctx, b, ctx->stream);  // work items follow a device-side hand-over (no-op otherwise)
        if (ps != PG_OK)
            return ps;
        const pg_status ws2 = ensure_ctx_workspace(ctx, b);  // (a re-made plan never needs more than the upload-time one; cheap)
        if (ws2 != PG_OK)
            return ws2;
    }
    // (the lean stage's decision and its preparations -- the work items re-made as the full plan's slots, which can change the chunks and
    //  grow the workspace -- BEFORE the regions are cut and the side streams are told to wait for the main one: a fill on the second fill
    //  stream must not start before the list and item kernels queued here are done)
    const bool revg_early = (flags & PG_AF_REVERSE_GRAPH) != 0;
    const uint64_t lean_min_cells = ctx->lean_min_cells;
    const auto lean_chunk_ok = [&](const Chunk& ch) { return !pg_var_wide(ch.C) && ch.cells >= lean_min_cells; };
    bool lean = ctx->lean && (flags & PG_AF_CIGAR) && (flags & PG_AF_BOTH_STRANDS) && revg_early && !b->has_general_reads && b->gen_idx.empty()
        && b->n_reads && !b->groups.empty();
    if (lean)
    {
        lean = false;
        for (const Chunk& ch : b->chunks)
            lean = lean || lean_chunk_ok(ch);
    }
    if (lean)
    {
        if (!b->device_plan)
        {
            const pg_status rp = cascade_rebuild_items(ctx, b, ctx->stream, b->has_active ? b->d_active : nullptr);
            if (rp != PG_OK)
                return rp;
            const pg_status ws3 = ensure_ctx_workspace(ctx, b);
            if (ws3 != PG_OK)
                return ws3;
        }
        if (b->full_pairs > b->cap_lean_pairs || b->n_reads > b->cap_lean_reads)
        {
            b->park(b->d_inst);
            b->park(b->d_lean_extra);
            b->park(b->d_yloc);
            b->park(b->d_lean_ucount);
            b->park(b->d_lean_ulist);
            b->d_inst = nullptr;
            b->d_lean_extra = b->d_yloc = b->d_lean_ucount = b->d_lean_ulist = nullptr;
            b->cap_lean_pairs = b->full_pairs + b->full_pairs / 8 + 16;
            b->cap_lean_reads = (size_t)b->n_reads + b->n_reads / 8 + 16;
            HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_inst, b->cap_lean_pairs * sizeof(PgInstItem)));
            HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_lean_extra, b->cap_lean_pairs * sizeof(uint32_t)));
            HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_yloc, b->cap_lean_reads * sizeof(uint32_t)));
            HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_lean_ucount, b->cap_lean_pairs * sizeof(uint32_t)));
            HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_lean_ulist, 4 * b->cap_lean_pairs * sizeof(uint32_t)));
        }
    }
    if ((!(flags & PG_AF_KEEP_RESULTS) || (flags == PG_AF_ALL)) && !b->ops_counter_fresh)
        HIP_TRY(ctx, hipMemsetAsync(b->d_ops_counter, 0, sizeof(unsigned long long), ctx->stream));
    b->ops_counter_fresh = false;
    const bool revg = (flags & PG_AF_REVERSE_GRAPH) != 0;
    // Chunk pipeline on two streams: fill(c) runs on `stream`, pick + traceback(c) on `stream2`; chunk number ctx->chunk_seq
    // (counted over all batches of the ctx) uses workspace half (chunk_seq & 1), so the latency-bound traceback of a chunk
    // overlaps the VALU-bound fill of the next one -- also across batches: nothing of this call makes the main stream wait
    // for a traceback except the one that still reads the half about to be overwritten.
    const unsigned regions = ctx->regions();
    const uint64_t half = (ctx->ws_cap / regions) & ~(uint64_t)255;  // keeps the 256-byte alignment of the trace rows
    {
        // the trace stream (and the second fill stream) must see everything queued on the main stream so far (memsets, uploads,
        // seed stages, the waits of pg_stage_begin)
        hipEvent_t e0;
        HIP_TRY(ctx, get_sync_event(ctx, &e0));
        HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, e0, 0));
        if (ctx->fill_streams == 2)
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream_fill2, e0, 0));
        ctx->sync_events_in_flight.push_back(e0);
    }
    // one chunk: its fills on the fill stream, pick + traceback behind them on the second stream.  lean_chunk: the lean stage's three
    // launches (reversed-graph fills, pick, forward-graph fills of the instances) instead of the one launch of all four fills
    auto run_chunk = [&](const Chunk& ch, const bool lean_chunk) -> pg_status {
        const uint32_t n_pairs = ch.pair_end - ch.pair_begin;
        const unsigned h = (unsigned)(ctx->chunk_seq % regions);
        // with two fill streams consecutive chunks' fills are on different streams: the later one's wavefronts fill the slots the
        // earlier one leaves while it drains
        const hipStream_t fill_stream = (ctx->fill_streams == 2 && (ctx->chunk_seq & 1u)) ? ctx->stream_fill2 : ctx->stream;
        uint8_t* ws = ctx->workspace + h * half;
        if (ctx->region_free[h])
            HIP_TRY(ctx, hipStreamWaitEvent(fill_stream, ctx->region_free[h], 0));  // the traceback that read this region is done
        PgFillArgs fa{};
        fa.items = b->d_items;
        fa.item_begin = 2 * ch.pair_begin;
        fa.graphs = G->d_graphs;
        fa.nodes = G->d_nodes;
        fa.preds = G->d_preds;
        fa.colmeta = G->d_colmeta;
        fa.base_off = b->d_base_off;
        fa.bases = b->d_bases;
        fa.workspace = ws;
        fa.fillsum = b->d_fillsum;
        hipStream_t done_stream = fill_stream;  // the stream the chunk's last fill launch is on
        EventPair ev{};
        if (ctx->timing)
        {
            HIP_TRY(ctx, get_event(ctx, &ev.a));
            HIP_TRY(ctx, get_event(ctx, &ev.b));
            ev.kind = 0;
            HIP_TRY(ctx, hipEventRecord(ev.a, fill_stream));
        }
        if (lean_chunk && ctx->lean_fused)
        {
            // ONE launch: a wavefront takes two pairs through their reversed-graph fills, the pick and the forward-graph fills (pg_fill.hip)
            fa.inst = b->d_inst;
            // the chunk's runs (segments are in the order of their pair slots)
            uint32_t seg_begin = 0, n_seg = 0;
            {
                const std::vector<PgPlanSegment>& sv = b->h_segments;
                const auto lb = std::lower_bound(sv.begin(), sv.end(), ch.pair_begin, [](const PgPlanSegment& x, uint32_t v) { return x.pair_begin < v; });
                const auto ub = std::lower_bound(sv.begin(), sv.end(), ch.pair_end, [](const PgPlanSegment& x, uint32_t v) { return x.pair_begin < v; });
                seg_begin = (uint32_t)(lb - sv.begin());
                n_seg = (uint32_t)(ub - lb);
            }
            HIP_TRY(ctx, hipMemsetAsync(b->d_lean_ucount + ch.pair_begin, 0, sizeof(uint32_t), fill_stream));
            // (every instance item of the chunk EMPTY -- entry (0, 0) = PG_NONE: the kernel writes the slots of the couples that hold
            // reads, the second forward launch walks them all)
            HIP_TRY(ctx, hipMemsetAsync(b->d_inst + ch.pair_begin, 0xFF, (size_t)n_pairs * sizeof(PgInstItem), fill_stream));
            HIP_TRY(ctx, pg_launch_fill_lean_fused(ch.C, fa, b->d_segments, (uint32_t)b->h_segments.size(), b->d_group_count, b->d_inst, b->d_yloc,
                                                   b->d_lean_ucount + ch.pair_begin, b->d_lean_ulist + 4 * (size_t)ch.pair_begin, seg_begin, n_seg, n_pairs,
                                                   fill_stream));
            ev.kind = 4;
        }
        else if (lean_chunk)
        {
            fa.inst = b->d_inst;
            HIP_TRY(ctx, pg_launch_fill_lean(ch.C, fa, n_pairs, 2, fill_stream));
            if (ctx->timing)
            {
                ev.kind = 2;
                HIP_TRY(ctx, hipEventRecord(ev.b, fill_stream));
                ctx->events.push_back(ev);
            }
            PgLeanBuildArgs la{};
            la.pair_begin = ch.pair_begin;
            la.n_pairs = n_pairs;
            la.items = b->d_items;
            la.fillsum = b->d_fillsum;
            la.segments = b->d_segments;
            la.n_segments = (uint32_t)b->h_segments.size();
            la.group_count = b->d_group_count;
            la.inst = b->d_inst;
            la.extra = b->d_lean_extra;
            la.yloc = b->d_yloc;
            la.ucount = b->d_lean_ucount + ch.pair_begin;
            // The pick and the forward launch go to a stream of their own: the next chunk's reversed-graph fills on the fill stream
            // do not wait for them.
            // (one priority level up, like the second stream: the forward launch then takes the slots a draining reversed-graph launch
            // frees before the next chunk's reversed-graph fills do, its traceback starts that much earlier, and the two kinds of
            // wavefront -- one writes the H trace, the other nothing -- share the machine instead of taking turns at the memory:
            // 38.9 -> 36.4 ms per million reads, profiles/r06_lean_streams_ab.jsonl.  Made by the first lean stage: an idle
            // high-priority stream is not free, pg_internal.h.)
            if (!ctx->stream_lean && !getenv("PG_LEAN_ONE_STREAM"))
                HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->stream_lean, hipStreamNonBlocking, ctx->side_priority));
            if (ctx->stream_lean)
            {
                hipEvent_t rev_done;
                HIP_TRY(ctx, get_sync_event(ctx, &rev_done));
                HIP_TRY(ctx, hipEventRecord(rev_done, fill_stream));
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream_lean, rev_done, 0));
                ctx->sync_events_in_flight.push_back(rev_done);
                done_stream = ctx->stream_lean;
            }
            if (ctx->timing)
            {
                HIP_TRY(ctx, get_event(ctx, &ev.a));
                HIP_TRY(ctx, get_event(ctx, &ev.b));
                ev.kind = 3;
                HIP_TRY(ctx, hipEventRecord(ev.a, done_stream));
            }
            HIP_TRY(ctx, pg_launch_lean_build(la, done_stream));
            HIP_TRY(ctx, pg_launch_fill_lean(ch.C, fa, n_pairs, 3, done_stream));
        }
        else
            HIP_TRY(ctx, pg_launch_fill(ch.C, fa, n_pairs, revg, ctx->wide32, fill_stream));
        if (ctx->timing)
        {
            HIP_TRY(ctx, hipEventRecord(ev.b, done_stream));
            ctx->events.push_back(ev);
            // (lean: two reversed-graph fills and one forward-graph fill per read -- the fourth fills of the reads that need them, a
            // few per cent, are not counted; the forward-graph fills of half the wavefronts write the trace)
            ctx->acc.fills += lean_chunk ? ch.fills / 4 * 3 : revg ? ch.fills : ch.fills / 2;
            ctx->acc.cells += lean_chunk ? ch.cells / 4 * 3 : revg ? ch.cells : ch.cells / 2;
            ctx->acc.trace_bytes += lean_chunk ? ch.trace_bytes / 2 : ch.trace_bytes;
        }
        hipEvent_t fill_done;
        HIP_TRY(ctx, get_sync_event(ctx, &fill_done));
        HIP_TRY(ctx, hipEventRecord(fill_done, done_stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, fill_done, 0));
        ctx->sync_events_in_flight.push_back(fill_done);
        PgTraceArgs ta{};
        ta.items = b->d_items;
        ta.pair_begin = ch.pair_begin;
        ta.n_pairs = n_pairs;
        ta.C = ch.C;
        ta.wide32 = ctx->wide32 ? 1u : 0u;
        ta.flags = flags;
        ta.graphs = G->d_graphs;
        ta.nodes = G->d_nodes;
        ta.preds = G->d_preds;
        ta.seqchars = G->d_seqchars;
        ta.base_off = b->d_base_off;
        ta.bases = b->d_bases;
        ta.workspace = ws;
        ta.workspace_rw = ws;
        ta.fillsum = b->d_fillsum;
        ta.results = b->d_results;
        ta.ops = b->d_ops;
        ta.ops_counter = b->d_ops_counter;
        if (ctx->timing)
        {
            HIP_TRY(ctx, get_event(ctx, &ev.a));
            HIP_TRY(ctx, get_event(ctx, &ev.b));
            ev.kind = 1;
            HIP_TRY(ctx, hipEventRecord(ev.a, ctx->stream2));
        }
        if (lean_chunk)
        {
            ta.inst = b->d_inst;
            ta.segments = b->d_segments;
            ta.n_segments = (uint32_t)b->h_segments.size();
            ta.yloc = b->d_yloc;
            ta.inst_rw = b->d_inst;
            ta.yloc_rw = b->d_yloc;
            ta.extra = b->d_lean_extra;
            ta.group_count = b->d_group_count;
            ta.ucount = b->d_lean_ucount + ch.pair_begin;           // (one counter and 4 list entries per pair slot: a chunk's are its own)
            ta.ulist = b->d_lean_ulist + 4 * (size_t)ch.pair_begin;
            ta.fused = ctx->lean_fused ? 1u : 0u;
            HIP_TRY(ctx, pg_launch_trace_lean(ta, ctx->stream2));
            // the reads whose record needs the fourth fill after all (X not unique through its own forward fill, a per cent or two):
            // that fill -- the instances the first look queued -- and a second look at them, here on the second stream, under the
            // next chunk's fills
            static const bool skip_second = getenv("PG_LEAN_TIMING_SKIP_SECOND") != nullptr;  // (timing probes only: the listed reads keep no record)
            if (!skip_second)
            {
                HIP_TRY(ctx, pg_launch_fill_lean(ch.C, fa, n_pairs, 4, ctx->stream2));
                HIP_TRY(ctx, pg_launch_trace_lean2(ta, ctx->stream2));
            }
        }
        else
            HIP_TRY(ctx, pg_launch_trace(ta, ctx->stream2));
        if (ctx->timing)
        {
            HIP_TRY(ctx, hipEventRecord(ev.b, ctx->stream2));
            ctx->events.push_back(ev);
        }
        hipEvent_t td;
        HIP_TRY(ctx, get_sync_event(ctx, &td));
        HIP_TRY(ctx, hipEventRecord(td, ctx->stream2));
        ctx->region_free[h] = td;
        ctx->sync_events_in_flight.push_back(td);
        ++ctx->chunk_seq;
        return PG_OK;
    };
    // ---- the lean gssw stage (pg_ctx_set_lean) -------------------------------------------------------------------------------------
    // alignRead(AF_ALL) asks for four fills per read and reads one record off them (GraphAligner.cpp:340-401).  A fill's best score is the
    // same on the graph and on the reversed graph, so the reversed-graph fills of both strands (scores + their two multi flags, no
    // trace) already say which strand X scores higher, and the record only ever needs the forward-graph fill of the OTHER strand when
    // X turns out not to be unique while that strand still may be.  Per chunk, ONE launch (pg_fill_lean_fused_kernel, pg_fill.hip): a
    // wavefront takes two work-item pairs of a run through their reversed-graph sweeps, the pick, and the forward-graph sweep of their
    // eight X strands (one per (16-lane group, register half): an instance item); the other strands a record still needs -- known from
    // the reversed-graph fills, or found by X's own forward fill -- it queues for a second, small forward launch, which follows the
    // traceback's first look on the SECOND stream together with a second look at those reads, under the next chunk's fills.  (ctx->
    // lean_fused off: the stage's first form, three launches per chunk -- reversed-graph fills of every work item, a pick kernel that
    // packs the instances, the forward-graph fills of the instance items on a stream of their own.)
    // Same records as the plain stage field by field, except multi_mask's bit of a forward fill that did not run
    // (PG_MULTI_OTHER_FWD_SKIPPED says so).  Byte variants (reads <= 250 bases); other chunks run the plain stage.
    // A chunk takes the lean route when its launches are long: the one launch of all four fills becomes two dependent ones of a half
    // and a quarter of its wavefronts, every launch ends with a tail of one wavefront's lifetime, and a launch that does not fill the
    // 4 096 wavefront slots a few times over lasts that lifetime whatever it holds.  A workflow batch's chunk (19 200 reads of 150
    // bases on 600-column graphs: 7 G cell updates, a 1 ms launch) is no faster lean than plain and the BAM -> genotypes job lost a
    // tenth with it -- a third with a path stage in front, whose seed chains wait behind the forward launch's stream
    // (profiles/r06_lean_e2e_ab.jsonl); the headline's chunks (150 G cell updates) and those of 250-base reads on kilobase nodes
    // (120 G in 6 000 pairs) gain a fifth and a tenth.  The bound: 30 G cell updates of the plain stage, a launch of some 5 ms
    // (pg_ctx_set_lean(ctx, 2) / PG_LEAN_MIN_CELLS: other bounds).
    if (lean)
    {
        for (const Chunk& ch : b->chunks)
        {
            const pg_status cs = run_chunk(ch, lean_chunk_ok(ch));
            if (cs != PG_OK)
                return cs;
        }
    }
    else
        for (const Chunk& ch : b->chunks)
        {
            const pg_status cs = run_chunk(ch, false);
            if (cs != PG_OK)
                return cs;
        }
    if (!b->gen_idx.empty())
    {
        const pg_status gs = run_general(ctx, b, flags);
        if (gs != PG_OK)
            return gs;
    }
    // the batch is busy until its last traceback is over; later stages of THIS batch wait for that (pg_stage_begin*), other
    // batches' fills do not
    stage_end.done = true;
    HIP_TRY(ctx, pg_stage_end_on(ctx, b, ctx->stream2));
    return PG_OK;
}

extern "C" pg_status pg_batch_ops_count(pg_ctx* ctx, pg_batch* b, uint64_t* n_ops)
{
    if (!ctx || !b || !n_ops)
        return PG_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long v = 0;
    if (b->d_ops_counter)
    {
        HIP_TRY(ctx, pg_batch_wait(ctx, b));
        HIP_TRY(ctx, hipMemcpyAsync(&v, b->d_ops_counter, sizeof v, hipMemcpyDeviceToHost, ctx->stream_copy));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_copy));
    }
    *n_ops = v;
    return PG_OK;
}

extern "C" pg_status pg_batch_download(
    pg_ctx* ctx, pg_batch* b, pg_result* results, pg_op* ops, uint64_t ops_cap, uint64_t* n_ops)
{
    if (!ctx || !b || (b->n_reads && !results))
        return fail(ctx, PG_ERR_INVALID, "pg_batch_download: null argument");
    uint64_t cnt = 0;
    pg_status st = pg_batch_ops_count(ctx, b, &cnt);
    if (st != PG_OK)
        return st;
    if (n_ops)
        *n_ops = cnt;
    if (b->n_reads)
        HIP_TRY(ctx, hipMemcpyAsync(results, b->d_results, b->n_reads * sizeof(pg_result), hipMemcpyDeviceToHost, ctx->stream_copy));
    if (ops && cnt)
    {
        if (cnt > ops_cap)
        {
            HIP_TRY(ctx, pg_stream_wait(ctx->device, ctx->stream_copy));
            return fail(ctx, PG_ERR_OVERFLOW, "ops buffer too small");
        }
        HIP_TRY(ctx, hipMemcpyAsync(ops, b->d_ops, cnt * sizeof(pg_op), hipMemcpyDeviceToHost, ctx->stream_copy));
    }
    return pg_path_index_check(ctx, b->graphs);  // (waits for the copy stream; a path index built on the device reports here)
}

extern "C" pg_status pg_align_batch(
    pg_ctx* ctx, const pg_graphs* graphs, uint32_t n_reads, const uint32_t* graph_of_read, const uint32_t* base_off,
    const char* bases, uint32_t flags, pg_result* results, pg_op* ops, uint64_t ops_cap, uint64_t* n_ops)
{
    pg_batch* b = nullptr;
    pg_status st = pg_batch_create(ctx, &b);
    if (st != PG_OK)
        return st;
    st = pg_batch_upload(ctx, b, graphs, n_reads, graph_of_read, base_off, bases);
    if (st == PG_OK)
        st = pg_batch_align(ctx, b, flags);
    if (st == PG_OK)
        st = pg_batch_download(ctx, b, results, ops, ops_cap, n_ops);
    pg_batch_destroy(ctx, b);
    return st;
}

extern "C" size_t pg_render_cigar(const pg_result* r, const pg_op* ops, char* buf, size_t cap)
{
    static const char OPC[] = "MXNIDS";
    size_t len = 0;
    char tmp[32];
    auto put = [&](const char* s, size_t n) {
        for (size_t i = 0; i < n; ++i, ++len)
            if (buf && len + 1 < cap)
                buf[len] = s[i];
    };
    uint32_t cur = 0xFFFFFFFFu;
    for (uint32_t e = 0; r && ops && e < r->n_ops; ++e)
    {
        const pg_op o = ops[r->ops_off + e];
        const uint32_t node = PG_OP_NODE(o), code = PG_OP_CODE(o);
        if (node != cur)
        {
            if (cur != 0xFFFFFFFFu)
                put("]", 1);
            int k = snprintf(tmp, sizeof tmp, "%u[", node);
            put(tmp, (size_t)k);
            cur = node;
        }
        if (code <= PG_OPC_S)
        {
            // pieces of one run (same node, same op: a run beyond PG_OP_MAX_LEN) print as one element
            uint32_t run = PG_OP_LEN(o);
            while (e + 1 < r->n_ops && PG_OP_NODE(ops[r->ops_off + e + 1]) == node && PG_OP_CODE(ops[r->ops_off + e + 1]) == code)
                run += PG_OP_LEN(ops[r->ops_off + ++e]);
            int k = snprintf(tmp, sizeof tmp, "%u%c", run, OPC[code]);
            put(tmp, (size_t)k);
        }
    }
    if (cur != 0xFFFFFFFFu)
        put("]", 1);
    if (buf && cap)
        buf[len < cap ? len : cap - 1] = 0;
    return len;
}

extern "C" pg_status pg_render_cigars(const pg_result* results, uint64_t n, const pg_op* ops, char* buf, size_t stride)
{
    if ((n && (!results || !buf)) || stride < 2)
        return PG_ERR_INVALID;
    pg_status st = PG_OK;
    for (uint64_t i = 0; i < n; ++i)
    {
        char* slot = buf + i * stride;
        const size_t len = pg_render_cigar(&results[i], ops, slot, stride);
        if (len >= stride)
            st = PG_ERR_OVERFLOW;
        else
            memset(slot + len, 0, stride - len);
    }
    return st;
}
