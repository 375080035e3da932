// pg_internal.h -- host-side object definitions shared by the C-ABI translation units (internal).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/paragraph_amd.h"
#include "pg_device.h"
#include "pg_general.h"
#include "pg_kernels.h"

struct HostGraph
{
    uint32_t n_nodes;
    uint32_t ncols;
    bool general_only;  // longer than the packed kernels' 65 519 columns: every read on it takes the general path (pg_general.h)
};

struct Chunk
{
    int C;
    uint32_t pair_begin, pair_end;
    uint64_t ws_bytes;
    uint32_t max_nodes;
    uint64_t fills, cells, trace_bytes;
};

// Device-resident cascade hand-over (pg_batch_retire_mapped): a run of reads of one (variant, graph) in the sorted order of the
// upload-time plan, and a piece of such a run inside one chunk of a plan made from the device's per-run counts.
struct PgReadGroup
{
    uint32_t C, graph;
    uint32_t list_base;  // first slot of the group in the device's active-read list (room for all its reads)
    uint32_t n_reads;    // reads of the group in the batch
    uint64_t sum_len;    // their bases (for the cell counts of the timing figures)
};
struct EventPair
{
    hipEvent_t a, b;
    int kind;  // 0 fill, 1 trace
};

struct pg_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // pick + traceback of chunk i overlaps the fill of chunk i + 1
    hipStream_t stream_copy = nullptr;  // uploads of the NEXT batch / downloads of the PREVIOUS one overlap the kernels
    // The klib stage's scratch (candidate records, work list, CIGAR slots, direction bytes): the context's, like the fill's workspace
    // -- the stage's kernels run one batch after the other on the main stream and nothing of it outlives the stage's last kernel.  (As
    // blocks of every batch's own index they were a quarter of a gigabyte allocated and freed per batch: 5 ms per hipMalloc, the
    // workflow with the klib stage at 5 k sites/s.)
    void* klib_scratch[4] = { nullptr, nullptr, nullptr, nullptr };
    size_t klib_scratch_bytes[4] = { 0, 0, 0, 0 };
    std::vector<hipEvent_t> sync_event_pool, sync_events_in_flight;
    // The workspace is `regions` equal regions used in turn by the chunks of ALL batches in the order they are aligned
    // (chunk_seq): the fill of a chunk only waits for the traceback that last read its region (region_free), so the traceback +
    // count of one batch (second stream) run under the fill of the next batch.  fill_streams = 1: two regions, fills on the
    // main stream one after the other.  fill_streams = 2 (pg_ctx_set_fill_streams): THREE regions and the fills alternate
    // between the main stream and stream_fill2 -- the next chunk's wavefronts take the slots a draining launch leaves (every
    // launch ends with a tail of one wavefront lifetime, 10 - 15 % of a 2 ms launch), while the traceback of the chunk before
    // still has its region to itself.
    uint64_t chunk_seq = 0;
    hipStream_t stream_fill2 = nullptr;
    hipStream_t stream_lean = nullptr;  // the lean stage's pick + forward launch (side priority; made by the first lean stage)
    // The path stage and the hand-over chain behind it (its count pass, the retire kernel, the lists and counts of the next plan)
    // run on a stream of their own, one priority level up like the second stream: on the main stream they would wait behind the
    // fills of every batch queued before them, on the second stream behind tracebacks that wait for those fills.
    hipStream_t stream_seed = nullptr;
    // Seed streams beyond the first, dealt to the batches' path stages in turn (each batch's seed chain stays on one).  A path stage
    // beside the fills is latency: it waits for wavefront slots the fills hold (0.6 ms) and then walks chains of dependent loads
    // (0.7 ms for the 50 wavefronts of a workflow batch with the device to itself, 1.5 ms beside a fill), and on ONE stream the
    // batches of all the lanes queue behind each other for it -- 68 x 2 ms per pass of the 10 000-site job.  Two streams halve that
    // (profiles/r05_e2e_seed_streams_ab.jsonl: 61-66k -> 71-72k sites/s); four share hardware queues with the count stream and
    // lose again.  Made by the first path stage, not with the context: one more high-priority stream in a process that never
    // runs a path stage cost the gssw-only workflow 5 %.
    hipStream_t stream_seed_more[3] = { nullptr, nullptr, nullptr };
    int seed_streams = 2;
    int side_priority = 0;
    unsigned seed_turn = 0;
    bool is_seed_stream(hipStream_t s) const
    {
        return s == stream_seed || (s && (s == stream_seed_more[0] || s == stream_seed_more[1] || s == stream_seed_more[2]));
    }
    int fill_streams = 1;
    uint32_t plan_epoch = 0;  // counts the changes of the region count: a batch planned under another one is refused (pg_batch_align)
    unsigned regions() const { return fill_streams == 2 ? 3u : 2u; }
    hipEvent_t region_free[3] = { nullptr, nullptr, nullptr };
    uint64_t ws_limit = 8ull << 30;
    uint64_t max_lds_per_block = 64 * 1024;  // hipDeviceAttributeMaxSharedMemoryPerBlock (gfx950: 160 KB)
    uint8_t* workspace = nullptr;
    uint64_t ws_cap = 0;
    // workspace of the general path (reads / graphs beyond the packed kernels' envelope); used on stream2 only
    uint8_t* gen_ws = nullptr;
    uint64_t gen_ws_cap = 0;
    bool wide32 = true;  // wide variants (reads of 251..512 bases) with 32 lanes per read
    // The lean gssw stage (pg_ctx_set_lean; on unless PG_LEAN=0): alignRead(AF_ALL) with three fills per read where four are not
    // needed (pg_batch_align)
    bool lean = true;
    bool lean_fused = true;  // the lean stage in one launch per chunk (pg_fill_lean_fused_kernel); PG_LEAN_FUSED=0: the three-launch form
    uint64_t lean_min_cells_default = 30000000000ull;  // a chunk goes lean from this many cell updates (of its four fills per read) on: pg_batch_align
    uint64_t lean_min_cells = 30000000000ull;
    bool timing = false;
    std::vector<EventPair> events;
    std::vector<hipEvent_t> event_pool;
    pg_timing acc{};
    std::string err;
};

struct pg_path_index;
void pg_path_index_free(pg_path_index* ix);
struct pg_kmer_index;
void pg_kmer_index_free(pg_kmer_index* ix);
struct pg_klib_index;
void pg_klib_index_free(pg_klib_index* ix);

struct pg_graphs
{
    uint32_t n_graphs = 0;
    // last stage queued on this graph set, per compute stream (main, second): pg_graphs_destroy waits for these two events
    // only -- not for whatever other batches have queued on the streams
    mutable hipEvent_t ev_use[3] = { nullptr, nullptr, nullptr };  // (main stream, second stream, seed stream)
    mutable bool use_recorded[3] = { false, false, false };
    mutable hipStream_t use_stream[3] = { nullptr, nullptr, nullptr };  // the stream each event was last recorded on (the seed slot is shared by several)
    std::vector<HostGraph> host;
    void* d_layout_block = nullptr;  // one allocation behind d_graphs .. d_seqchars (PgStagedUpload)
    void* d_count_block = nullptr;   // one allocation behind d_cnt_graphs .. d_in_mask
    PgGraphDev* d_graphs = nullptr;
    PgNode* d_nodes = nullptr;
    uint32_t* d_preds = nullptr;
    uint32_t* d_colmeta = nullptr;
    char* d_seqchars = nullptr;
    // ---- count path (pg_count.hip): host CSR copies + device tables in the caller's CSR indexing
    std::vector<uint32_t> h_node_off;  // n_graphs + 1
    std::vector<uint32_t> h_pred_off;  // total_nodes + 1
    std::vector<uint32_t> h_pred;
    std::vector<uint32_t> h_node_len;
    std::vector<uint32_t> h_nodeseq_off;  // total_nodes + 1, into h_seq_raw
    std::string h_seq_raw;            // node sequences exactly as given (the path stage compares raw characters)
    pg_path_index* path_index = nullptr;
    pg_path_index* filter_index = nullptr;  // KmerFilter (count path)
    pg_kmer_index* kmer_index = nullptr;
    pg_klib_index* klib_index = nullptr;
    std::vector<uint32_t> h_n_labels;  // per graph
    std::vector<uint64_t> h_seq_off;   // n_graphs + 1 (dense sequence-set slots)
    bool labels_set = false;
    PgCountGraph* d_cnt_graphs = nullptr;
    uint32_t* d_cnt_pred_off = nullptr;
    uint32_t* d_cnt_pred = nullptr;
    uint32_t* d_cnt_node_len = nullptr;
    uint32_t label_words = 1;          // 64-bit words per label set (set-wide; pg_graphs_set_labels_wide)
    uint32_t frag_lds_counters = 0;    // LDS counters a pg_fragment_kernel block needs for any ONE graph of the set (capped)
    uint64_t* d_label_mask = nullptr;  // per predecessor entry x label_words
    uint64_t* d_out_mask = nullptr;    // per node x label_words: labels on outgoing edges
    uint64_t* d_in_mask = nullptr;     // per node x label_words: labels on incoming edges
};

struct pg_batch
{
    const pg_graphs* graphs = nullptr;
    uint32_t n_reads = 0;
    uint32_t n_pairs = 0;
    uint32_t* d_base_off = nullptr;
    char* d_bases = nullptr;
    char* d_bases_rc = nullptr;  // the reads reverse-complemented, at the same offsets (written by the path kernel, each thread its own read)
    size_t cap_bases_rc = 0;
    PgWorkItem* d_items = nullptr;
    PgFillSummary* d_fillsum = nullptr;
    pg_result* d_results = nullptr;
    pg_op* d_ops = nullptr;
    uint64_t ops_cap = 0;
    unsigned long long* d_ops_counter = nullptr;
    std::vector<Chunk> chunks;
    // reads of the current plan that take the general path, and their device-side records
    std::vector<uint32_t> gen_idx;
    std::vector<PgGenRead> h_gen_reads;  // host copy of the records in flight (the upload reads it asynchronously)
    PgGenRead* d_gen_reads = nullptr;
    PgFillSummary* d_gen_fsum = nullptr;
    size_t cap_gen = 0;
    uint64_t max_ws = 0;
    uint64_t gen_reserve = 0;  // the general path's share of the context's workspace budget (plan_items)
    size_t cap_reads = 0, cap_bases = 0, cap_items = 0;
    std::vector<pg_result> host_template;  // status for reads the device never sees (empty reads)
    bool has_skipped = false;
    // ---- count path
    std::vector<uint32_t> h_graph_of_read;
    std::vector<uint32_t> h_base_off;
    uint8_t* d_path_flags = nullptr;  // per read: bit0 mapped by the last seed stage, bit1 anchored, bit2 BAD_ALIGN
    uint8_t* d_active = nullptr;      // per read: 0 = skipped by the stage kernels (NULL semantics via has_active)
    bool has_active = false;
    // ---- device-resident hand-over between cascade stages (pg_batch_retire_mapped -> pg_batch_ensure_plan)
    std::vector<PgReadGroup> groups;          // the (variant, graph) runs of the upload-time plan
    std::vector<uint32_t> h_group_of_read;    // per read: its group, PG_NONE for empty reads and reads of the general path
    bool has_general_reads = false;           // some read of the batch takes the general path (then the host re-plans from the flags)
    bool plan_stale = false;                  // d_active changed on the device since the work items were made
    bool device_plan = false;                 // the work items are the full plan's slots (cascade_rebuild_items), not plan_items' own
    // ---- the lean gssw stage's tables (pg_batch_align): instance items, extra-instance counters per pair slot, where a read's
    //      other strand was filled, the reads its pick leaves to the plain pass
    PgInstItem* d_inst = nullptr;
    uint32_t* d_lean_extra = nullptr;
    uint32_t* d_yloc = nullptr;
    uint32_t* d_lean_ucount = nullptr;  // per pair slot (a chunk uses the one at its first pair): reads listed for the second look
    uint32_t* d_lean_ulist = nullptr;   // four entries per pair slot: (pair << 2 | read of the pair)
    size_t cap_lean_pairs = 0, cap_lean_reads = 0;
    uint32_t plan_epoch = 0;                  // pg_ctx::plan_epoch when the upload-time plan was cut
    hipStream_t seed_stream = nullptr;        // the seed stream this batch's path stage ran on
    bool seed_chain = false;                  // the batch's last stage ran on the seed stream (pg_batch_path_align): its count pass and
                                              // hand-over follow it there
    bool cascade_uploaded = false;
    uint32_t* d_group_of_read2 = nullptr;     // [n_reads] group per read
    uint32_t* d_group_base = nullptr;         // [n_groups] list_base per group
    uint32_t* d_group_count = nullptr;        // [n_groups] active reads per group (made by pg_group_list_kernel)
    uint32_t* d_active_list = nullptr;        // [n_reads] active reads, group after group
    PgPlanSegment* d_segments = nullptr;
    size_t cap_groups = 0, cap_cascade_reads = 0, cap_segments = 0;
    std::vector<uint32_t> h_group_base;       // (kept: the upload of the group tables reads it asynchronously)
    bool full_plan_ready = false;             // the full plan's segments are on the device, full_chunks / full_pairs hold it
    std::vector<Chunk> full_chunks;
    uint32_t full_pairs = 0;
    uint64_t full_max_ws = 0;
    std::vector<PgPlanSegment> h_segments;
    uint32_t* d_graph_of_read = nullptr;
    pg_read_support* d_support = nullptr;
    uint64_t* d_label_ext = nullptr;  // [n_reads][label_words - 1]: the label sets' words beyond pg_read_support.label_mask
    size_t cap_label_ext = 0;
    uint32_t label_ext_words = 0;     // words - 1 of the last pg_batch_count
    uint32_t label_ext_reads = 0;     // ... and the reads it counted (a later upload must not be read with this stride)
    uint32_t* d_path = nullptr;
    unsigned long long* d_path_counter = nullptr;
    uint32_t* d_frag_off = nullptr;
    uint32_t* d_frag_reads = nullptr;
    uint8_t* d_is_rev = nullptr;
    uint32_t* d_counts = nullptr;  // owned table (when the caller passes none)
    size_t cap_count_reads = 0, cap_frags = 0;
    uint64_t cap_counts = 0;
    uint32_t n_frags = 0;
    // {CIGAR elements, path entries} of the batch as of its last pg_batch_count, copied into page-locked host memory by the count
    // stream itself (pg_batch_result_sizes reads them after the batch's event: no copy + wait of their own)
    unsigned long long* h_counters = nullptr;
    unsigned long long* d_h_counters = nullptr;  // the same page-locked words as the device sees them (the count pass writes them itself)
    bool ops_counter_fresh = false;              // zeroed by pg_batch_upload and not used since
    bool h_counters_valid = false;
    bool counts_owned_valid = false;
    bool fragments_set = false;
    // ---- batch pipelining: uploads run on ctx->stream_copy, kernels on ctx->stream
    hipEvent_t ev_upload = nullptr;  // recorded on stream_copy when the batch's inputs are resident
    hipEvent_t ev_busy = nullptr;    // recorded on stream after the last stage queued for this batch
    // Device blocks a STAGE call outgrew (count table, label sets, the cascade's lists): a stage runs under the caller's device lock
    // with the batch's earlier stages still on the device, so it must not wait for them before it frees -- it parks the old block
    // here and takes a new one; pg_batch_upload (the batch is idle then) and pg_batch_destroy free what is parked.  (The wait that
    // used to stand there held the lock of a 24-lane workflow for the length of the batch's fills: 27 % of its batches, 87 ms of a
    // 145 ms pass, PG_API_TIMING.)
    std::vector<void*> parked_blocks;
    void park(void* p)
    {
        if (p)
            parked_blocks.push_back(p);
    }
    int device = 0;                  // the context's device (the wait service asks for events of several devices)
    hipEvent_t ev_host = nullptr;    // what the host waits on where it used to synchronise a stream (pg_wait_stream)
    hipEvent_t ev_cascade = nullptr; // the cascade's tables of this upload are on the device (pg_cascade_prepare_early: copy stream)
    bool cascade_recorded = false;
    bool upload_recorded = false, busy_recorded = false;
};

// Every function that queues work on a batch's buffers on ctx->stream calls pg_stage_begin first (the main stream waits for
// the batch's upload) and pg_stage_end last (marks the batch busy until here).  pg_batch_wait blocks the host until the
// batch's queued stages and uploads are complete -- without waiting for other batches' work.
hipError_t pg_stage_begin(pg_ctx* ctx, pg_batch* b);
hipError_t pg_stage_end(pg_ctx* ctx, pg_batch* b);
// the same for a stage that runs on the second stream (traceback, count path): never touches the main stream, so the
// fills of other batches queued there are not held up
hipError_t pg_stage_begin_on(pg_ctx* ctx, pg_batch* b, hipStream_t s);
hipError_t pg_stage_end_on(pg_ctx* ctx, pg_batch* b, hipStream_t s);
hipError_t pg_batch_wait(pg_ctx* ctx, pg_batch* b);
// Host waits.  A lane waits for its batch for milliseconds while the other lanes need its CPU: the events a host thread waits on
// are made with hipEventBlockingSync (the thread sleeps until the completion interrupt) unless PG_SPIN_WAITS is set -- the
// device-wide hipDeviceScheduleBlockingSync (pg_device_prefer_blocking_waits) is refused once the device is in use, which it
// always is in a process that imported torch before it came here.  pg_wait_stream: hipStreamSynchronize by the same rule.
unsigned pg_wait_event_flags();
hipError_t pg_wait_stream(pg_batch* b, hipStream_t s);
// the calling thread sleeps until the event is complete; ONE service thread polls the pending events of all callers (pg_api.hip)
hipError_t pg_event_wait(int device, hipEvent_t ev);
hipError_t pg_stream_wait(int device, hipStream_t s);  // hipStreamSynchronize by the same route
// The tables the device-side hand-over needs (group of every read, list bases, plan segments) go up on the COPY stream while the
// batch's first seed stage runs, not on the seed stream between its count pass and the hand-over kernels.
pg_status pg_cascade_prepare_early(pg_ctx* ctx, pg_batch* b);
// the work items follow d_active: pg_batch_retire_mapped re-writes them on the device; this is for the cases it leaves (a batch with
// general-path reads, pg_batch_set_active(NULL) after a hand-over) -- called by the stages that run work items, on `stream`
pg_status pg_batch_ensure_plan(pg_ctx* ctx, pg_batch* b, hipStream_t stream);


// Device memory of graph sets and batches comes from a per-device cache of idle blocks (size classes of an eighth of an
// octave): hipFree synchronises the whole device and hipMalloc takes a driver round trip, and a workflow creates and drops
// a dozen small tables per batch.  A block must be idle on the device when it is handed back (callers wait for the owning
// batch / the compute stream first); the cache keeps at most PG_DEV_CACHE_MAX_IDLE bytes and releases the rest.
hipError_t pg_dev_alloc(void** p, size_t bytes);
hipError_t pg_dev_free(void* p);
void pg_dev_cache_release();  // hipFree of every idle block of the current device

// Several small tables -> ONE device block through ONE page-locked staging block and ONE copy.  A graph set is a dozen tables of
// a few KB each; from pageable vectors every one of them was its own blocking hipMemcpyAsync (150 us of host time apiece,
// 6 888 calls per two passes of the 10 000-site workflow).  add() notes a table and where its device pointer goes; commit()
// packs them 256-byte aligned, uploads on `stream`, sets the pointers and hands back the device block (pg_dev_free it when
// the tables go); it waits for the copy, so the staging block returns to its cache before commit() does.
struct PgStagedUpload
{
    struct Item
    {
        const void* src;
        size_t bytes;
        void** dst;
    };
    std::vector<Item> items;
    template <typename T> void add(const std::vector<T>& v, T** dst) { items.push_back(Item{ v.data(), v.size() * sizeof(T), (void**)dst }); }
    void add_raw(const void* src, size_t bytes, void** dst) { items.push_back(Item{ src, bytes, dst }); }
    hipError_t commit(hipStream_t stream, void** device_block);
    // the same without the wait: the caller queues more work behind the copy, waits once, and then hands the page-locked staging
    // block back (pg_pinned_put(*staging, *staging_cap))
    hipError_t commit_async(hipStream_t stream, void** device_block, void** staging, size_t* staging_cap);
};
// page-locked host blocks by size class (hipHostMalloc takes milliseconds)
hipError_t pg_pinned_get(size_t bytes, void** p, size_t* cap);
void pg_pinned_put(void* p, size_t cap);

pg_status pg_fail(pg_ctx* ctx, pg_status st, const std::string& msg);

// PG_API_TIMING=1: host time of every runtime call made through HIP_TRY (and of the scopes marked PG_TIMED), summed per call site
// text and printed when the process ends -- what a lane's device section costs the host, call by call
extern bool pg_api_timing;
uint64_t pg_now_ns();
void pg_api_time_add(const char* what, uint64_t t0_ns);
struct PgTimedScope
{
    const char* what;
    uint64_t t0;
    explicit PgTimedScope(const char* w) : what(w), t0(pg_api_timing ? pg_now_ns() : 0) {}
    ~PgTimedScope()
    {
        if (pg_api_timing)
            pg_api_time_add(what, t0);
    }
};
#define PG_TIMED(label) PgTimedScope pg_timed_scope__(label)

#define HIP_TRY(ctx, call)                                                                                     \
    do                                                                                                         \
    {                                                                                                          \
        const uint64_t t0__ = pg_api_timing ? pg_now_ns() : 0;                                                 \
        hipError_t e__ = (call);                                                                               \
        if (pg_api_timing)                                                                                     \
            pg_api_time_add(#call, t0__);                                                                      \
        if (e__ != hipSuccess)                                                                                 \
            return pg_fail(ctx, e__ == hipErrorOutOfMemory ? PG_ERR_NOMEM : PG_ERR_HIP,                          \
                           std::string(#call) + ": " + hipGetErrorString(e__));                                \
    } while (0)
