/*
 * paragraph_amd.h -- C ABI of the MI355X-native graph-realignment core.
 *
 * This is the drop-in boundary for the read -> variant-graph realignment path of Illumina/paragraph:
 * everything the reference does between "a graph and a vector of reads" and "per-read graph
 * alignments (+ node/edge support counts)" is behind these entry points.  Plain C, no exceptions,
 * integer status codes, caller-owned buffers, no torch/HIP types in any signature.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * checkout, GT! = inside external/graph-tools.tar.gz):
 *
 *   pg_graphs_upload     grm::GraphAligner::setGraph            src/c++/lib/grm/GraphAligner.cpp:277-285
 *                        (+ GraphAlignerImpl::initializeGraph   GraphAligner.cpp:110-167,
 *                           graphtools::reverseGraph            GT!/src/graphcore/GraphOperations.cpp:38-60)
 *   pg_batch_upload      the read vector handed to grm::alignReads   src/c++/include/grm/Align.hh:49-52
 *   pg_batch_align       grm::alignReads -> CompositeAligner::alignRead (gssw stage)
 *                        -> GraphAligner::alignRead             src/c++/lib/grm/Align.cpp:114-156,
 *                                                               CompositeAligner.cpp:152-175, GraphAligner.cpp:308-404
 *                        i.e. gssw_graph_fill + gssw_graph_trace_back x4 per read
 *                                                               external/gssw/gssw.c:4033-4044, 3539-3560
 *   pg_batch_download    the graph_* fields written into common::Read  src/c++/include/common/Read.hh:40-264
 *   pg_align_batch       one-shot convenience = upload + align + download
 *   pg_render_cigar      GraphAlignerImpl::extractCigar         GraphAligner.cpp:88-108
 *   pg_graphs_set_labels graphtools::Graph::addLabelToEdge as grm::graphFromJson fills it   src/c++/lib/grm/GraphInput.cpp:126-156
 *   pg_graphs_build_path_index / pg_batch_path_align   grm::PathAligner::{setGraph,alignRead}   src/c++/lib/grm/PathAligner.cpp:70-164
 *   pg_graphs_build_kmer_index / pg_batch_kmer_align   grm::KmerAligner<16>::{setGraph,alignRead}   src/c++/lib/grm/KmerAligner.cpp:305-538
 *   pg_graphs_build_klib_index / pg_batch_klib_align   grm::KlibAligner::{setGraph,alignRead}     src/c++/lib/grm/KlibAligner.cpp:186-442
 *                        (+ common::KlibAlignment::update = ksw_align + ksw_global   src/c++/lib/common/Klib.cpp:144-164, external/klib/ksw.c)
 *   pg_graphs_build_filter_index   readfilters::KmerFilter's graphtools::KmerIndex   src/c++/lib/paragraph/readfilters/KmerFilter.cpp:52-76
 *   pg_batch_set_active  the `status != MAPPED` hand-over between cascade stages   src/c++/lib/grm/CompositeAligner.cpp:78-176
 *   pg_batch_set_fragments   Read::fragment_id / is_reverse_strand of the input reads   src/c++/include/common/Read.hh:40-120
 *   pg_batch_count       read filters (NonUniq, BadAlign) applied by CompositeAligner::alignRead
 *                                                               src/c++/lib/paragraph/ReadFilter.cpp:43-90, readfilters/{NonUniq,BadAlign}.hh,
 *                                                               src/c++/lib/grm/CompositeAligner.cpp:152-175
 *                        + paragraph::disambiguateReads with the production node/edge filters
 *                                                               src/c++/lib/paragraph/Disambiguation.cpp:82-142, 212-296
 *                        + paragraph::countReads (fragments, node/edge/sequence counts)
 *                                                               src/c++/lib/paragraph/ReadCounting.cpp:52-127, 225-244,
 *                                                               src/c++/lib/common/Fragment.cpp:34-69, 141-181
 *
 * Threading: one pg_ctx per device; calls on one ctx must be serialised by the caller (same rule
 * as one CompositeAligner instance per worker in the reference, Align.cpp:107-110).  Exception:
 * pg_graphs_upload, pg_graphs_set_labels and the pg_graphs_build_*_index calls only build up a new
 * graph set (host work + copy stream) and may run while another thread is inside a pg_batch_* call
 * on the same ctx with OTHER graph sets -- a graph set can be prepared for the next batch while the
 * current one is on the device.  Likewise pg_batch_create / _upload / _set_fragments, the _download* calls and
 * pg_batch_destroy touch only their own batch and the copy stream: they may overlap the stage calls
 * (pg_batch_path_align / _kmer_align / _klib_align / _align / _count / _set_active, which share the ctx workspace and
 * stay serialised) of OTHER batches -- batch k+1 goes up and batch k-1 comes down while batch k computes.
 * The stage calls only QUEUE work: none of them waits for the device, whatever the batch object held before (a device block a
 * stage outgrows is kept until the batch's next upload), so the lock a caller holds around them is held for microseconds.
 * pg_last_error is then whichever call failed last.
 * Scoring is fixed as in the reference: match +1, mismatch -4, gap open 6, gap extend 1,
 * N / non-ACGTU = 0 (GraphAligner.cpp:229-233, gssw.c:4188-4220).
 */
#ifndef PARAGRAPH_AMD_H
#define PARAGRAPH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t pg_status;
enum
{
    PG_OK = 0,
    PG_ERR_INVALID = 1,     /* bad argument (NULL, non-topological edge, empty node, ...) */
    PG_ERR_NO_DEVICE = 2,   /* no usable HIP device */
    PG_ERR_HIP = 3,         /* a HIP runtime call failed (see pg_last_error) */
    PG_ERR_UNSUPPORTED = 4, /* outside the supported envelope (read > 16 000 bp, > 65 535 nodes, > 256 labels, ...) */
    PG_ERR_NOMEM = 5,
    PG_ERR_OVERFLOW = 6     /* an output buffer supplied by the caller is too small */
};

/* GraphAligner alignment flags, src/c++/include/grm/GraphAligner.hh:45-54 */
enum
{
    PG_AF_CIGAR = 1u,
    PG_AF_BOTH_STRANDS = 2u,
    PG_AF_REVERSE_GRAPH = 4u,
    PG_AF_ALL = 0xFFFFFFFFu,
    /* library extension (ignored when flags == PG_AF_ALL): keep the results / ops already produced for reads
     * that are not active in this call (stage 2 of the cascade after pg_batch_path_align) */
    PG_AF_KEEP_RESULTS = 0x100u
};

enum
{
    PG_MAX_READ_LEN = 512, /* the packed kernels: <= 250 gssw's byte mode (never overflows, gssw.c:380); 251..512 its 16-bit
                              word mode.  Longer reads (up to 16 000 bases) take the general device path, as do graphs of more
                              than PG_MAX_PACKED_NODES nodes or 65 519 columns: same results, far slower */
    PG_MAX_PACKED_NODES = 4095,
    PG_MAX_NODES = 65535   /* node ids are 16-bit fields of pg_op and of the count path's entries */
};

typedef struct pg_ctx pg_ctx;
typedef struct pg_graphs pg_graphs; /* a set of variant graphs resident on the device */
typedef struct pg_batch pg_batch;   /* device-resident reads + results of one batch */

/* Per-read result: the graph_* fields GraphAligner::alignRead writes (GraphAligner.cpp:358-401). */
typedef struct pg_result
{
    int32_t graph_pos;        /* Read::graph_pos: 0-based start in the first node of the alignment */
    int16_t score;            /* Read::graph_alignment_score */
    uint8_t mapq;             /* Read::graph_mapq: 60 if unique else 0 */
    uint8_t is_unique;        /* Read::is_graph_alignment_unique */
    uint8_t returned_reverse; /* 1: the reverse complement of the input bases is what was aligned
                                 (caller flips bases/quals and is_graph_reverse_strand, GraphAligner.cpp:358-378) */
    uint8_t multi_mask;       /* bit0 fwd-graph/fwd-strand, bit1 fwd-graph/rc, bit2 rev-graph/fwd, bit3 rev-graph/rc:
                                 alignsEndAtMultNodes of that fill (GraphAligner.cpp:170-212) */
    uint16_t n_ops;           /* number of pg_op entries of this read */
    uint32_t ops_off;         /* first entry in the ops array */
    int16_t strand_score[2];  /* best score of the forward / reverse-complement strand fill */
    uint16_t clipped;         /* soft-clipped bases (both ends) of the chosen alignment: what BadAlign needs */
    uint16_t status;          /* low byte: 0 ok; 1 = degenerate (score 0, empty CIGAR: the reference's behaviour is
                                 undefined downstream); 2 = internal traceback inconsistency.
                                 PG_STATUS_PATH_ALIGNER is or-ed in when the PathAligner stage produced the record */
} pg_result;
/* multi_mask bit 4 (lean pass, PG_AF_LEAN): the forward-graph fill of the strand that was NOT returned did not run -- its bit (0 or 1)
 * reads 0; nothing of the reference's Read depends on it (GraphAligner.cpp:340-356: the returned strand was unique, or the other
 * strand's reversed-graph fill already made it non-unique) */
#define PG_MULTI_OTHER_FWD_SKIPPED 0x10u
#define PG_STATUS_PATH_ALIGNER 0x100u
#define PG_STATUS_KMER_ALIGNER 0x200u
#define PG_STATUS_KLIB_ALIGNER 0x400u

/* One run-length CIGAR element inside a node: node id (16 bits) | op (4 bits) | length (12 bits).  A run longer than
 * PG_OP_MAX_LEN (only reads beyond 4 095 bases can have one) comes as several consecutive elements of the same node and op:
 * add their lengths (pg_render_cigar does). */
typedef uint32_t pg_op;
#define PG_OP_MAX_LEN 0xFFFu
#define PG_OP_MAKE(node, code, len) (((uint32_t)(node) << 16) | ((uint32_t)(code) << 12) | ((uint32_t)(len) & PG_OP_MAX_LEN))
#define PG_OP_NODE(x) ((uint32_t)(x) >> 16)
#define PG_OP_CODE(x) (((uint32_t)(x) >> 12) & 0xFu)
#define PG_OP_LEN(x) ((uint32_t)(x) & PG_OP_MAX_LEN)
/* op codes -> characters "MXNIDS"; code 6 = "node visited with an empty CIGAR" marker (length 0) */
enum
{
    PG_OPC_M = 0,
    PG_OPC_X = 1,
    PG_OPC_N = 2,
    PG_OPC_I = 3,
    PG_OPC_D = 4,
    PG_OPC_S = 5,
    PG_OPC_EMPTY = 6
};

/* Timing of the device work of the last pg_batch_align calls (HIP events on the ctx stream). */
typedef struct pg_timing
{
    double fill_ms;   /* accumulated duration of the DP fill kernel launches */
    double trace_ms;  /* accumulated duration of the pick + traceback kernel launches */
    uint64_t fill_launches;
    uint64_t trace_launches;
    uint64_t fills;       /* (read, strand, graph direction) fills performed */
    uint64_t cells;       /* DP cell updates performed (useful cells, no padding) */
    uint64_t trace_bytes; /* bytes of traceback state written by the fill kernel */
    /* the lean gssw stage's two fill launches per chunk (both are also in fill_ms; fill_launches counts the chunk once).  The forward
     * launch runs on a stream of its own beside the next chunk's reversed-graph launch: the two durations overlap in time, their sum
     * can exceed the wall clock */
    double lean_rev_ms;   /* reversed-graph fills of both strands */
    double lean_fwd_ms;   /* pick + forward-graph fills of the instance items */
    uint64_t lean_rev_launches;
    uint64_t lean_fwd_launches;
    double lean_fused_ms;          /* the lean stage as ONE launch per chunk (the default form; also in fill_ms / fill_launches) */
    uint64_t lean_fused_launches;
} pg_timing;

/* Host threads that wait for `device` (pg_batch_wait, downloads, pg_ctx_sync) sleep instead of spinning:
 * hipSetDeviceFlags(hipDeviceScheduleBlockingSync).  Has an effect only before the device's first use in the process, so
 * call it before pg_ctx_create; harmless (PG_OK) later.  For workflows whose host threads fill the CPUs they may use: a
 * spinning wait takes a CPU from another lane's read extraction (the reference has no device to wait for; its threads
 * block in htslib and in each other's mutexes, src/c++/lib/grmpy/Workflow.cpp:225-231). */
pg_status pg_device_prefer_blocking_waits(int device);
pg_status pg_ctx_create(int device, pg_ctx** out);
void pg_ctx_destroy(pg_ctx* ctx);
const char* pg_strerror(pg_status st);
const char* pg_last_error(const pg_ctx* ctx);
/* bytes of HBM the ctx may use for traceback state per chunk (default 8 GiB) */
pg_status pg_ctx_set_workspace_bytes(pg_ctx* ctx, uint64_t bytes);
/* The LEAN gssw stage (ON by default; pg_ctx_set_lean(ctx, 0), or PG_LEAN=0 in the environment for every new context, gives the plain
 * four fills per read and all four multi_mask bits): pg_batch_align with
 * CIGAR + both strands + reversed graph -- GraphAligner::alignRead(AF_ALL), GraphAligner.cpp:308-404 -- computes the record from THREE
 * fills per read where the fourth cannot change it: the reversed-graph fills of both strands first (a fill's best score is the same
 * on the graph and on the reversed graph), then the forward-graph fill of the higher-scoring strand X, and that of the other strand
 * only where X is not unique and the other strand still may be (GraphAligner.cpp:340-356) -- known from the reversed-graph fills, or
 * found by X's own forward fill (those reads get the fourth fill in a second, small launch of the same call).  Every field of the reference's Read is what the plain
 * stage writes; of pg_result, multi_mask's bit of a forward fill that did not run reads 0 and PG_MULTI_OTHER_FWD_SKIPPED is set.
 * Reads of up to 250 bases; longer ones (and batches with general-path reads) run the plain stage -- and so do chunks (the pieces
 * of a batch that share the workspace) whose four fills per read come to fewer than 30 G cell updates (PG_LEAN_MIN_CELLS; 50 000
 * reads of 150 bases on 1 kb of graph): two dependent launches of a half and a quarter of the plain launch's wavefronts only pay
 * when each still lasts milliseconds.  on = 0: off; 1: on from that size; 2: on for every chunk (PG_LEAN=2; what the tests use).
 * tests/test_gpu_lean.py. */
pg_status pg_ctx_set_lean(pg_ctx* ctx, int on);

/* 1 (default): the fills of a batch's chunks run one after the other on the main stream, over two workspace regions.
 * 2: three regions, the fills alternate between two streams, so that the next chunk's wavefronts take the slots a draining
 * launch leaves -- for workflows whose launches are short (a 192-site batch fills for ~2 ms, 10 - 15 % of it the tail in
 * which the chip drains; the reference has no such grain: one site per call, lib/grmpy/AlignSamples.cpp:115-172).  Call it
 * before batches are uploaded (the chunk plan is cut to the region size); it drains the compute streams.  Results do not
 * depend on it.  The environment variable PG_FILL_STREAMS=1|2 overrides it (A/B timing). */
pg_status pg_ctx_set_fill_streams(pg_ctx* ctx, int n);
pg_status pg_ctx_sync(pg_ctx* ctx);
/* waits for the compute streams only (stage calls queued so far); uploads / downloads of other batches keep running */
pg_status pg_ctx_sync_compute(pg_ctx* ctx);
pg_status pg_ctx_timing_enable(pg_ctx* ctx, int enable);
pg_status pg_ctx_timing_reset(pg_ctx* ctx);
pg_status pg_ctx_timing_get(pg_ctx* ctx, pg_timing* out); /* synchronises */

/* ---------------------------------------------------------------------------------------------------
 * Pinned host staging.  The reference packs a site's reads on worker threads (src/c++/lib/grmpy/AlignSamples.cpp:115-172,
 * src/c++/lib/common/ReadExtraction.cpp:38-219) into std::vector<p_Read>; the batched form of that hand-over is a set
 * of flat arrays, and when those live in page-locked memory every copy of pg_batch_upload / pg_batch_set_fragments /
 * pg_batch_download* is a DMA on the copy stream that overlaps another batch's kernels (double buffering: two sets of
 * pinned arrays, two pg_batch objects).  Pageable arrays still work -- the runtime then stages through bounce buffers on
 * the calling thread.  pg_host_alloc'ed memory is visible to every device (portable).
 * ------------------------------------------------------------------------------------------------- */
pg_status pg_host_alloc(pg_ctx* ctx, size_t bytes, void** out);
void pg_host_free(pg_ctx* ctx, void* p);
/* page-locks / releases memory the caller allocated itself (e.g. the storage of a std::vector that is kept alive) */
pg_status pg_host_register(pg_ctx* ctx, void* p, size_t bytes);
pg_status pg_host_unregister(pg_ctx* ctx, void* p);

/*
 * Upload n_graphs variant graphs.  Graph g owns nodes [node_off[g], node_off[g+1]); node ids inside a
 * graph are 0-based in topological order (every edge goes from a lower to a higher id, as
 * graphtools::Graph::addEdge enforces).  CSR over ALL nodes of all graphs:
 *   seq_off[n_total+1], seq   : node sequences (any case; upper-cased by the library as initializeGraph does)
 *   pred_off[n_total+1], pred : predecessors of each node as graph-local ids, ascending
 * The reversed graph (node i -> n-1-i, sequences reversed) is derived by the library.
 */
pg_status pg_graphs_upload(
    pg_ctx* ctx, uint32_t n_graphs, const uint32_t* node_off, const uint32_t* seq_off, const char* seq,
    const uint32_t* pred_off, const uint32_t* pred, pg_graphs** out);
void pg_graphs_destroy(pg_ctx* ctx, pg_graphs* graphs);

/* Batch = reads resident on the device.  graph_of_read[i] indexes the graph set. */
pg_status pg_batch_create(pg_ctx* ctx, pg_batch** out);
void pg_batch_destroy(pg_ctx* ctx, pg_batch* batch);
pg_status pg_batch_upload(
    pg_ctx* ctx, pg_batch* batch, const pg_graphs* graphs, uint32_t n_reads, const uint32_t* graph_of_read,
    const uint32_t* base_off /* n_reads+1 */, const char* bases);
/* Runs the device path (fill + pick + traceback) asynchronously on the ctx stream. */
pg_status pg_batch_align(pg_ctx* ctx, pg_batch* batch, uint32_t flags);
/* Number of pg_op entries produced by the last pg_batch_align (synchronises). */
pg_status pg_batch_ops_count(pg_ctx* ctx, pg_batch* batch, uint64_t* n_ops);
/* Copies results (n_reads entries) and ops (up to ops_cap entries) to host memory; synchronises. */
pg_status pg_batch_download(
    pg_ctx* ctx, pg_batch* batch, pg_result* results, pg_op* ops, uint64_t ops_cap, uint64_t* n_ops);

/* One-shot: upload reads, align, download. */
pg_status pg_align_batch(
    pg_ctx* ctx, const pg_graphs* graphs, uint32_t n_reads, const uint32_t* graph_of_read, const uint32_t* base_off,
    const char* bases, uint32_t flags, pg_result* results, pg_op* ops, uint64_t ops_cap, uint64_t* n_ops);

/* ---------------------------------------------------------------------------------------------------
 * Count path: read filters -> node/edge/sequence support per read -> per-fragment union -> per-site
 * counters {count, :READS, :FWD, :REV} (ReadCounting.cpp:52-69).
 * ------------------------------------------------------------------------------------------------- */
typedef struct pg_count_params
{
    uint32_t remove_nonuniq;      /* paragraph --bad-align-nonuniq (default 1; grmpy keeps that default, Parameters.hh:141) */
    uint32_t use_support_filters; /* 1: production nodefilter/edgefilter (Disambiguation.cpp:212-296); 0: none
                                     (what the reference's unit tests call disambiguateReads with) */
    double bad_align_frac;        /* --bad-align-frac, default 0.8 */
    uint32_t use_kmer_filter;     /* 1: KmerFilter after BadAlign (--bad-align-uniq-kmer-len != 0, default off); needs
                                     pg_graphs_build_filter_index */
    uint32_t reserved;
} pg_count_params;

/* Per-read outcome of the count path. */
typedef struct pg_read_support
{
    uint64_t label_mask; /* Read::graph_sequences_supported as a bit set over the graph's labels: labels 0..63 (the further
                            words of a graph set with more labels: pg_batch_download_label_ext) */
    uint32_t path_off;   /* first entry in the path array */
    uint16_t n_path;     /* nodes on the read's path */
    uint8_t status;      /* 0 not aligned / skipped, 1 MAPPED, 2 BAD_ALIGN (filtered), 3 invalid alignment */
    uint8_t filter;      /* 0 none, 1 nonuniq, 2 bad_align, 3 kmer_tooshort, 4 kmer_uncov (KmerFilter.cpp:88-139) */
} pg_read_support;
/* path entry: node id | (node supported) << 30 | (edge from the previous path node supported) << 31 */
#define PG_PATH_NODE(x) ((uint32_t)(x) & 0xFFFFu)
#define PG_PATH_NODE_OK(x) (((uint32_t)(x) >> 30) & 1u)
#define PG_PATH_EDGE_OK(x) (((uint32_t)(x) >> 31) & 1u)

enum
{
    PG_MAX_LABELS = 256,        /* labels per graph (bit set of up to PG_LABEL_WORDS 64-bit words) */
    PG_LABEL_WORDS = 4,
    PG_MAX_SEQ_TABLE_LABELS = 8 /* graphs with more labels get no dense sequence-set table (slots = 0) */
};

/* Layout of the uint32 counter table of a graph set: [nodes*4][edges*4][seq slots*4][tallies*4].
 * nodes/edges are indexed like the node / predecessor CSR given to pg_graphs_upload (edge k = k-th
 * predecessor entry); graph g owns seq slots [seq_off[g], seq_off[g] + 2^n_labels(g)) indexed by label
 * bit set; tallies per graph = {aligned, mapped, bad_align, nonuniq}. */
typedef struct pg_count_layout
{
    uint64_t n_counters; /* total uint32 entries */
    uint64_t node_base, edge_base, seq_base, tally_base; /* in uint32 entries */
    uint64_t n_nodes, n_edges, n_seq_slots, n_graphs;
} pg_count_layout;

/* label_mask_of_pred[k] = bit set of the labels on the edge (pred[k] -> its node), same indexing as the
 * `pred` array of pg_graphs_upload; n_labels[g] = number of labels of graph g (<= 64). */
pg_status pg_graphs_set_labels(
    pg_ctx* ctx, pg_graphs* graphs, const uint64_t* label_mask_of_pred, const uint32_t* n_labels);
/* The same for graph sets with more than 64 labels on a graph (the reference's PathFamily has no bound,
 * src/c++/lib/paragraph/ReadCounting.cpp:96-127): label_words_of_pred[k * words + w] = word w of the bit set of edge k,
 * words = 1 .. PG_LABEL_WORDS for the WHOLE set, n_labels[g] <= 64 * words.  pg_read_support.label_mask then holds word 0 of
 * a read's set and pg_batch_download_label_ext the others. */
pg_status pg_graphs_set_labels_wide(
    pg_ctx* ctx, pg_graphs* graphs, const uint64_t* label_words_of_pred, uint32_t words, const uint32_t* n_labels);
/* words per label set of the graph set (1 unless pg_graphs_set_labels_wide was given more) */
pg_status pg_graphs_label_words(const pg_graphs* graphs, uint32_t* words);
pg_status pg_graphs_count_layout(const pg_graphs* graphs, pg_count_layout* out);
/* seq_off[g] (n_graphs + 1 entries) of the layout above */
pg_status pg_graphs_seq_offsets(const pg_graphs* graphs, uint64_t* seq_off);

/* KmerFilter's graphtools::KmerIndex (src/c++/lib/paragraph/readfilters/KmerFilter.cpp:52-76): kmer_len > 0 = that
 * length on every graph; kmer_len < 0 = per graph the smallest length in 10..63 at which every node and edge is
 * overlapped by at least -kmer_len unique k-mers (graphtools::findMinCoveringKmerLength; PG_ERR_UNSUPPORTED when
 * there is none).  kmer_len_of_graph (n_graphs entries, may be NULL) receives the lengths used. */
pg_status pg_graphs_build_filter_index(pg_ctx* ctx, pg_graphs* graphs, int32_t kmer_len, uint32_t* kmer_len_of_graph);

/* Fragment membership of the uploaded reads (call once after pg_batch_upload):
 * fragment_of_read: fragment id per read (mates share an id; ids are local to the read's graph);
 * is_reverse_strand: Read::is_reverse_strand() of the input (BAM flag), may be NULL (all forward). */
pg_status pg_batch_set_fragments(
    pg_ctx* ctx, pg_batch* batch, const uint32_t* fragment_of_read, const uint8_t* is_reverse_strand);
/* Runs the count path on the device for the batch's last pg_batch_align results (asynchronous on the ctx
 * stream).  d_counts: DEVICE pointer to pg_count_layout.n_counters uint32 the kernels ADD into (caller zeroes
 * it; e.g. a torch tensor that is then all-reduced with RCCL), or NULL to use a table owned by the batch
 * (zeroed per call). */
pg_status pg_batch_count(pg_ctx* ctx, pg_batch* batch, const pg_count_params* params, uint32_t* d_counts);
/* Zeroes a caller-owned counter table ON THE CTX STREAM, i.e. ordered against the pg_batch_count calls before and after
 * it (a memset on any other stream is not).  Asynchronous. */
pg_status pg_counts_zero(pg_ctx* ctx, uint32_t* d_counts, uint64_t n_counters);
/* ---- Stream-ordered hand-over of the counter table to the caller's collective (the path's ONE all-reduce).
 * The reference's threads meet in a mutex around the merged counts (src/c++/lib/grmpy/Workflow.cpp:225-231); across GPUs the
 * meeting point is an RCCL all-reduce that the CALLER issues on a stream of its own.  These three calls order that stream
 * against the library's count stream WITHOUT blocking the host (pg_ctx_sync_compute is the blocking form):
 *   pg_ctx_count_record(ctx, ev)  hipEventRecord(ev, count stream): everything pg_counts_zero / pg_batch_count queued so far;
 *                                 the caller's stream waits for ev before it reduces the table
 *   pg_ctx_count_wait(ctx, ev)    hipStreamWaitEvent(count stream, ev): later pg_counts_zero / pg_batch_count calls start after
 *                                 ev -- recorded by the caller behind its reduce -- so a table is not zeroed under a reduce.
 *                                 With two tables taking turns step n + 1 never waits for reduce n.
 *   pg_ctx_native_stream          the stream itself, for callers that bring their own event type (torch.cuda.ExternalStream).
 * native_event / *out are the runtime's handles (hipEvent_t / hipStream_t) passed as plain pointers: no HIP type in the ABI. */
enum
{
    PG_STREAM_FILL = 0,  /* graph fills + seed stages */
    PG_STREAM_COUNT = 1, /* pick + traceback, count path, pg_counts_zero */
    PG_STREAM_COPY = 2   /* uploads / downloads */
};
pg_status pg_ctx_native_stream(pg_ctx* ctx, int which, void** out);
pg_status pg_ctx_count_record(pg_ctx* ctx, void* native_event);
pg_status pg_ctx_count_wait(pg_ctx* ctx, void* native_event);
/* Everything a workflow reads back after pg_batch_align + pg_batch_count, with ONE wait for the batch and ONE for the copies
 * (pg_batch_ops_count + pg_batch_download + two pg_batch_download_counts wait eight times):
 *   pg_batch_result_sizes   the number of CIGAR elements and path entries of the batch (pg_batch_count sends both to
 *                           page-locked host memory on its own stream; this call waits for the batch and reads them)
 *   pg_batch_download_all   results, CIGAR elements, count table (NULL when it lives in caller memory), supports, path entries
 * The batch's last stage must be pg_batch_count.  (What they replace: the graph_* fields and support lists written into
 * common::Read, src/c++/include/common/Read.hh:40-264, and countReads' tables, lib/paragraph/ReadCounting.cpp:52-127.) */
pg_status pg_batch_result_sizes(pg_ctx* ctx, pg_batch* batch, uint64_t* n_ops, uint64_t* n_path);
pg_status pg_batch_download_all(
    pg_ctx* ctx, pg_batch* batch, pg_result* results, pg_op* ops, uint64_t ops_cap, uint32_t* counts, pg_read_support* supports,
    uint32_t* path, uint64_t path_cap);
/* Copies the count table (if the batch owns it; pass NULL otherwise), per-read supports and path entries
 * (capacity path_cap entries; *n_path receives the number used) to host memory; synchronises. */
pg_status pg_batch_download_counts(
    pg_ctx* ctx, pg_batch* batch, uint32_t* counts, pg_read_support* supports, uint32_t* path, uint64_t path_cap,
    uint64_t* n_path);
/* Words 1 .. words-1 of every read's label set, label_ext[r * (words - 1) + (w - 1)] (word 0 is pg_read_support.label_mask);
 * cap_words = capacity of label_ext in 64-bit words (needs n_reads * (words - 1)).  Nothing to copy when words == 1. */
pg_status pg_batch_download_label_ext(pg_ctx* ctx, pg_batch* batch, uint64_t* label_ext, uint64_t cap_words);

/* ---------------------------------------------------------------------------------------------------
 * Exact path matching stage (grm::PathAligner, --path-sequence-matching; default ON in `paragraph`, OFF in grmpy)
 * ------------------------------------------------------------------------------------------------- */
/* KmerIndex of every graph (all length-k paths; the reference uses k = 32, PathAligner.hh) */
pg_status pg_graphs_build_path_index(pg_ctx* ctx, pg_graphs* graphs, uint32_t kmer_len);
/* PathAligner::alignRead for every read of the batch: reads whose whole length matches a path exactly get their
 * pg_result (status has PG_STATUS_PATH_ALIGNER set; is_graph_reverse_strand = returned_reverse, NOT xor-ed with the
 * BAM strand, PathAligner.cpp:121-129) and ops.  Resets the batch's results/ops.  Asynchronous. */
pg_status pg_batch_path_align(pg_ctx* ctx, pg_batch* batch);
/* flags[i] of the LAST seed stage run: bit0 = MAPPED, bit1 = anchored (path stage: a unique k-mer was found),
 * bit2 = BAD_ALIGN (k-mer stage), bit3 (path stage) = PG_PATH_FLAG_FWD_ABSENT: some k-mer of the read as given is in no
 * path of the graph; synchronises */
pg_status pg_batch_download_path_flags(pg_ctx* ctx, pg_batch* batch, uint8_t* flags);
#define PG_PATH_FLAG_FWD_ABSENT 8u
/* An EXACT shortcut for the gssw stage (GraphAligner::alignRead, src/c++/lib/grm/GraphAligner.cpp:308-404), for callers
 * that run gssw WITHOUT the PathAligner stage (grmpy's default cascade): after pg_batch_path_align, a read leaves the
 * following pg_batch_align -- with the record alignRead(AF_ALL) itself would have written -- when that record is forced:
 *   - the path stage found exactly ONE full-length exact match over both strands (a k-mer with one path in the graph,
 *     extended without a choice at any node end, PathOperations.cpp:117-271): every walk of the graph that spells the
 *     read holds that k-mer's one path and leaves it the same way, so one cell of one fill reaches score L = the read's
 *     length, all four multi-node flags of that strand are clear, and the traceback from that cell is the walk, all 'M';
 *   - the match is on the strand as given -- then alignRead returns it whatever the other strand scores
 *     (GraphAligner.cpp:340-356: unique beats non-unique, equal uniqueness goes to the better score, ties to the
 *     forward strand) -- or it is on the reverse complement AND some k-mer of the read as given is in no path of the
 *     graph (PG_PATH_FLAG_FWD_ABSENT), i.e. the forward strand scores below L;
 *   - the read holds A / C / G / T only (an N scores 0 in gssw where the path stage compares characters) and is at most
 *     250 bases long (beyond that gssw redoes the fill in its word mode, whose multi-node test reads bytes).
 * Such a read's record loses PG_STATUS_PATH_ALIGNER (it IS the gssw record: graph_pos, score = L, mapq 60, unique,
 * returned_reverse, CIGAR); multi_mask and strand_score only hold what is known (the other strand's fills never ran).
 * Every other read stays active and is aligned by pg_batch_align(flags | PG_AF_KEEP_RESULTS) as if the path stage had not
 * run.  On the device, queued behind the path stage, no wait; the work items of pg_batch_align are re-written like
 * pg_batch_retire_mapped does.  tests/test_gpu_exact.py compares every field with a plain pg_batch_align. */
pg_status pg_batch_retire_exact_matches(pg_ctx* ctx, pg_batch* batch);
/* ---------------------------------------------------------------------------------------------------
 * k-mer seed stage (grm::KmerAligner<16>, --kmer-sequence-matching; default OFF in both CLIs)
 * ------------------------------------------------------------------------------------------------- */
/* Paths of every graph (grm::pathsFromJson, GraphInput.cpp:163-196): graph g owns paths [path_off[g], path_off[g+1]),
 * path p = node ids path_nodes[path_node_off[p] .. path_node_off[p+1]) (whole nodes, first to last).  Builds the
 * per-path sorted (k-mer, position) tables; kmer_len <= 16 (the reference instantiates 16; its unit test 10). */
pg_status pg_graphs_build_kmer_index(
    pg_ctx* ctx, pg_graphs* graphs, uint32_t kmer_len, const uint32_t* path_off, const uint32_t* path_node_off,
    const uint32_t* path_nodes);
/* KmerAligner::alignRead for every ACTIVE read: results get PG_STATUS_KMER_ALIGNER; stage flags (see
 * pg_batch_download_path_flags): bit0 MAPPED, bit2 BAD_ALIGN (several equally good, different alignments).
 * flags: PG_AF_KEEP_RESULTS keeps earlier stages' results/ops.  Asynchronous. */
pg_status pg_batch_kmer_align(pg_ctx* ctx, pg_batch* batch, uint32_t flags);

/* ---------------------------------------------------------------------------------------------------
 * klib (ksw) stage (grm::KlibAligner, --klib-sequence-matching; default OFF in both CLIs)
 *   replaces KlibAligner::{setGraph,alignRead}  src/c++/lib/grm/KlibAligner.cpp:186-205, 388-442 and, underneath,
 *   common::KlibAlignment::update (src/c++/lib/common/Klib.cpp:144-164) = ksw_align(KSW_XSTART) + ksw_global
 *   (external/klib/ksw.c:223-355, 457-531) with match 1, mismatch -4, gap open 5 (+1 for its first base), extend 1.
 * ------------------------------------------------------------------------------------------------- */
/* Paths as for pg_graphs_build_kmer_index (<= 126 per graph, whole nodes, no empty node). */
pg_status pg_graphs_build_klib_index(
    pg_ctx* ctx, pg_graphs* graphs, const uint32_t* path_off, const uint32_t* path_node_off, const uint32_t* path_nodes);
/* KlibAligner::alignRead for every ACTIVE read (reads <= 512 bases): per path and strand one local alignment with start
 * recovery and a global re-alignment of the local window; best score wins, an equally good candidate with a different
 * (CIGAR, position) makes the read BAD_ALIGN.  results get PG_STATUS_KLIB_ALIGNER, score = number of matched bases
 * (KlibAligner.cpp:207-308), strand_score[] = the ksw score; stage flags as for the k-mer stage (bit0 MAPPED, bit2
 * BAD_ALIGN).  flags: PG_AF_KEEP_RESULTS keeps earlier stages' results/ops.  Asynchronous. */
pg_status pg_batch_klib_align(pg_ctx* ctx, pg_batch* batch, uint32_t flags);
/* Device-side overflow word of the klib stage since the last call (0 = none; bit0 the batch's op buffer was too small
 * for a read's CIGAR, bit1 a path CIGAR exceeded 2 * L + 4 runs: those reads stay unmapped).  Synchronises, clears. */
pg_status pg_graphs_klib_error(pg_ctx* ctx, pg_graphs* graphs, uint32_t* error);

/* Which kernels the last pg_batch_klib_align on this graph set ran: 1 = the packed two-strand kernels (every path
 * of the set at least as long as the longest read), 0 = the general ones.  Both give KlibAligner's results
 * (src/c++/lib/grm/KlibAligner.cpp:388-442); the tests use this to know which of the two they have checked. */
pg_status pg_graphs_klib_last_kernels(pg_ctx* ctx, pg_graphs* graphs, uint32_t* packed);

/* Restricts the following stage calls (pg_batch_path_align / pg_batch_kmer_align / pg_batch_klib_align / pg_batch_align) to reads with active[i] != 0 (NULL = every read): the next
 * stage of the cascade runs only on reads the previous stage left unmapped / filtered. */
pg_status pg_batch_set_active(pg_ctx* ctx, pg_batch* batch, const uint8_t* active);
/* The same hand-over decided where the flags are -- on the device (CompositeAligner::alignRead, CompositeAligner.cpp:78-150: a
 * read the stage mapped and the filter chain accepted is done, the others go on to the next stage).  After a seed stage and
 * its pg_batch_count: active[i] &= !((stage flag & 1) && count-path status == MAPPED), one thread per read, queued behind the
 * count pass; the wavefront work items of the stages behind it (pg_batch_klib_align, pg_batch_align) are re-written on the
 * device over the batch's full plan -- a group's active reads first, the pair slots left over EMPTY (their wavefronts return at
 * once).  Nothing crosses to the host, no call waits.  Asynchronous.  Results equal those of pg_batch_set_active with the
 * mask a host loop would build. */
pg_status pg_batch_retire_mapped(pg_ctx* ctx, pg_batch* batch);

/* Renders "<node>[<len><op>...]..." for one read into buf (NUL-terminated); returns the string length
 * (which may be >= cap, in which case the output was truncated). Host-only helper. */
size_t pg_render_cigar(const pg_result* r, const pg_op* ops, char* buf, size_t cap);
/* pg_render_cigar for n results into fixed slots of `stride` bytes (slot i at buf + i * stride, NUL-padded to the end of
 * the slot); PG_ERR_OVERFLOW when a string did not fit its slot (that slot holds the truncated string). Host-only helper. */
pg_status pg_render_cigars(const pg_result* results, uint64_t n, const pg_op* ops, char* buf, size_t stride);

#ifdef __cplusplus
}
#endif
#endif /* PARAGRAPH_AMD_H */
