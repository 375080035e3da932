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
 *
 * Threading: one pg_ctx per device; calls on one ctx must be serialised by the caller (same rule
 * as one CompositeAligner instance per worker in the reference, Align.cpp:107-110).
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
    PG_ERR_UNSUPPORTED = 4, /* outside the supported envelope (read > 250 bp, > 4095 nodes, ...) */
    PG_ERR_NOMEM = 5,
    PG_ERR_OVERFLOW = 6     /* an output buffer supplied by the caller is too small */
};

/* GraphAligner alignment flags, src/c++/include/grm/GraphAligner.hh:45-54 */
enum
{
    PG_AF_CIGAR = 1u,
    PG_AF_BOTH_STRANDS = 2u,
    PG_AF_REVERSE_GRAPH = 4u,
    PG_AF_ALL = 0xFFFFFFFFu
};

enum
{
    PG_MAX_READ_LEN = 250, /* byte-mode gssw never overflows up to here (gssw.c:380) */
    PG_MAX_NODES = 4095
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
    uint16_t status;          /* 0 ok; 1 = degenerate (score 0, empty CIGAR: the reference's behaviour is
                                 undefined downstream); 2 = internal traceback inconsistency */
} pg_result;

/* One run-length CIGAR element inside a node: node id (12 bits) | op (4 bits) | length (16 bits). */
typedef uint32_t pg_op;
#define PG_OP_NODE(x) ((uint32_t)(x) >> 20)
#define PG_OP_CODE(x) (((uint32_t)(x) >> 16) & 0xFu)
#define PG_OP_LEN(x) ((uint32_t)(x) & 0xFFFFu)
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
} pg_timing;

pg_status pg_ctx_create(int device, pg_ctx** out);
void pg_ctx_destroy(pg_ctx* ctx);
const char* pg_strerror(pg_status st);
const char* pg_last_error(const pg_ctx* ctx);
/* bytes of HBM the ctx may use for traceback state per chunk (default 8 GiB) */
pg_status pg_ctx_set_workspace_bytes(pg_ctx* ctx, uint64_t bytes);
pg_status pg_ctx_sync(pg_ctx* ctx);
pg_status pg_ctx_timing_enable(pg_ctx* ctx, int enable);
pg_status pg_ctx_timing_reset(pg_ctx* ctx);
pg_status pg_ctx_timing_get(pg_ctx* ctx, pg_timing* out); /* synchronises */

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

/* Renders "<node>[<len><op>...]..." for one read into buf (NUL-terminated); returns the string length
 * (which may be >= cap, in which case the output was truncated). Host-only helper. */
size_t pg_render_cigar(const pg_result* r, const pg_op* ops, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* PARAGRAPH_AMD_H */
