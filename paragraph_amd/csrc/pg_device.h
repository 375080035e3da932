// pg_device.h -- device-side data layout shared by the HIP kernels and the C-ABI host code.
//
// HBM layout (see DESIGN.md "Data layout"):
//   graph set   : PgGraphDev[n_graphs], PgNode[], preds[], colmeta[] (one u32 per graph column, both
//                 graph directions), seqchars[] (upper-cased node characters, forward direction)
//   batch       : raw read bases + offsets, PgWorkItem[] (one per wavefront), PgFillSummary[],
//                 pg_result[], pg_op[]
//   workspace   : per wavefront a "trace" region (H bytes of every DP cell of the two forward-graph
//                 fills, stored by pipeline step so each step is one contiguous coalesced store) and a
//                 "seed" region (last-column H / next-column E of every node)
#pragma once
#include <stdint.h>

#define PG_GROUPS 4        // reads per wavefront
#define PG_GROUP_LANES 16  // lanes per read
#define PG_NONE 0xFFFFFFFFu

// column meta word: bits 0-2 nt code (0-3 ACGT, 4 other), bit3 FIRST column of node, bit4 LAST column,
// bit5 SAVE (store the node's last column as a seed), bits 8.. node id
#define PG_META_CODE(m) ((m)&7u)
#define PG_META_FIRST 8u
#define PG_META_LAST 16u
#define PG_META_SAVE 32u
#define PG_META_NODE(m) (((m) >> 8) & 0xFFFu)
// first column of a node only: summary of its predecessors (bits 20..31)
#define PG_META_PRED_ADJ (1u << 20)   // the node directly before it in the layout is a predecessor (its state is still in registers)
#define PG_META_PRED_ONE (1u << 21)   // exactly one other predecessor, id (< 128) in bits 23..29
#define PG_META_PRED_MANY (2u << 21)  // anything else: read the predecessor table
#define PG_META_PRED_SHIFT 23
// bit 31, every column: something other than the plain recurrence happens in this step -- a node boundary (FIRST / LAST), or this
// column or the NEXT one of the layout carries code 4 (the profile rows of the next column are fetched one step ahead).  One
// sign test per step instead of assembling the condition from two words.
#define PG_META_RARE 0x80000000u
// bit 30: a copy of LAST.  RARE is always set on a LAST column, so a LAST column is the only kind of word that is >= 0xC0000000
// as an unsigned number: the test after the column is one compare.
#define PG_META_LAST_HI 0x40000000u
#define PG_META_IDLE (4u | PG_META_RARE)  // code 4 (scores 0 against everything)
#define PG_META_PAD 160   // idle words appended to every direction's column array (64-wide block prefetch)

// pipeline steps of one sweep: the 16 lanes of a read are skewed by one column each (ncols + 15 steps); rounded up to an
// even count because the step loop is unrolled twice with ping-pong register naming (the extra step runs on an idle column)
static inline __host__ __device__ uint32_t pg_fill_steps(uint32_t ncols) { return (ncols + PG_GROUP_LANES) & ~1u; }
// ... and of a sweep whose reads take `lanes` lanes each.  The WIDE variants (reads of 251..512 bases) run 32 lanes per read:
// a work item's four reads are swept by TWO wavefronts (reads 0-1 and 2-3, PG_WIDE_HALVES), each with half the rows per lane --
// half the LDS profile and far fewer registers per wavefront than four reads x 16 lanes x up to 32 rows.
#define PG_WIDE_LANES 32
#define PG_WIDE_HALVES 2
static inline __host__ __device__ uint32_t pg_fill_steps_lanes(uint32_t ncols, uint32_t lanes) { return (ncols + lanes) & ~1u; }

// The fill kernel keeps scores in a frame that moves by one per pipeline step (pg_fill.hip): the H trace holds
// (score + PG_TAU0 + (step & 255)) & 0xFF per cell, step = column + lane of the row.
#define PG_TAU0 8u

#define PG_GAP_OPEN 6
#define PG_GAP_EXT 1
#define PG_PAD_SCORE (-300)        // byte variants (scores <= 250)
#define PG_PAD_SCORE_WIDE (-1000)  // wide variants (scores <= 512)

struct PgGraphDir
{
    uint32_t meta_off;  // into colmeta[]; ncols + PG_META_PAD entries (tail = PG_META_IDLE)
    uint32_t ncols;     // total node length
    uint32_t node_off;  // into nodes[]
    uint32_t n_nodes;
};

struct PgGraphDev
{
    PgGraphDir dir[2];  // [0] forward graph, [1] reversed graph
    uint32_t seq_off;   // into seqchars[] (forward direction, by column)
    uint32_t pad;
};

struct PgNode
{
    uint32_t col_start;  // first column of the node in its direction's linear layout
    uint32_t len;
    uint32_t pred_off;  // into preds[] (ids local to the graph direction, ascending)
    uint32_t n_pred;
};

// One wavefront of work: up to 4 reads of the same graph, one graph direction, both strands.
struct PgWorkItem
{
    uint32_t graph;
    uint32_t dir;
    uint32_t read[PG_GROUPS];
    uint64_t trace_off;  // byte offset into the workspace (dir 0 only)
    uint64_t seed_off;   // byte offset into the workspace
};

// One wavefront of the LEAN forward pass (pg_batch_align with PG_AF_LEAN): eight (read, strand) instances of one graph, one per
// (16-lane group, 16-bit half).  An entry is the read's index, bit 31 set for its reverse complement; PG_NONE = empty.
#define PG_INST_RC 0x80000000u
#define PG_YLOC_PENDING 0x80000000u  // yloc entry: the fill is queued for the chunk's second forward launch (the traceback's first look passes the read by)
struct alignas(128) PgInstItem  // (a cache line of its own: the fused kernel's wavefront writes its items and reads them back at once)
{
    uint32_t graph;
    uint32_t pad;
    uint32_t inst[2][PG_GROUPS];  // [half][group]
    uint64_t trace_off;
    uint64_t seed_off;
};

// A piece of a (variant, graph) run of reads inside one chunk of a batch's full plan (pg_api.hip, cascade_full_plan).
struct PgPlanSegment
{
    uint32_t pair_begin, n_pairs;  // pair slots [pair_begin, pair_begin + n_pairs) of the batch's full plan
    uint32_t first_pair;           // ... are pairs first_pair.. of the group (four reads per pair, active reads first)
    uint32_t graph, list_base, group;
    uint64_t ws_base, need, trace_bytes, seed_bytes;
};

// Written by the fill kernel per (work item, group, strand).
struct PgFillSummary
{
    int32_t score;     // best local score of the fill (gssw max_node->score1)
    int32_t max_node;  // first node (topological) holding it
    // (the packed fill, pg_fill.hip; the general path, pg_general.h, fills in the end cell itself: ref_end, read_end = its row)
    int32_t ref_end;   // -1 (the node-local column of the end cell is the traceback kernel's: end_col - the node's first column)
    int32_t read_end;  // first row of the fill lane that holds the score in that column; the row itself (read_end1, the smallest
                       // read index in the column) is found by the traceback kernel among that lane's rows
    int32_t end_col;   // graph-global column of the score's first occurrence (ref_end1), -1 if score 0 or a reversed-graph fill
    int32_t multi;     // alignsEndAtMultNodes
    int32_t pad[2];
};

// Count path: per-graph bases into the caller-indexed tables.
struct PgCountGraph
{
    uint32_t node_base;  // first node of the graph in the set-wide node numbering
    uint32_t n_nodes;
    uint32_t n_labels;
    uint32_t pad;
    uint64_t seq_base;  // first dense sequence-set slot (valid if n_labels <= PG_MAX_SEQ_TABLE_LABELS)
};

// Kernel variants.  A variant code V is the rows-per-lane count C for the byte variants (reads <= 250 bp: one byte
// of H per cell, C = 2 * ceil(L / 32) in 2..16) and 64 + C for the WIDE variants (reads of 251..512 bp, gssw's
// 16-bit "word mode", gssw.c:527-786: two bytes of H per cell, C = 4 * ceil(L / 64) in 16..32).
#define PG_VAR_WIDE 64
static inline __host__ __device__ int pg_var_c(int V) { return V & 63; }
static inline __host__ __device__ bool pg_var_wide(int V) { return V >= PG_VAR_WIDE; }
static inline __host__ __device__ int pg_variant_of(uint32_t L)
{
    return L <= 250u ? (int)(2u * ((L + 31u) / 32u)) : PG_VAR_WIDE + (int)(4u * ((L + 63u) / 64u));
}
static inline __host__ __device__ uint32_t pg_rows(int V) { return (uint32_t)(PG_GROUP_LANES * pg_var_c(V)); }
// bytes of H trace one lane writes per pipeline step: C rows x 2 strands, one byte per cell in every variant (the wide
// variants store H mod 256: neighbouring cells differ by less than 128, which lets the traceback carry exact scores along)
static inline __host__ __device__ uint32_t pg_trace_lane_bytes(int V) { return (uint32_t)(2 * pg_var_c(V)); }
// bytes of seed (H, next-column E of both strands) one lane keeps per node
static inline __host__ __device__ uint32_t pg_seed_lane_bytes(int V) { return (uint32_t)((pg_var_wide(V) ? 8 : 4) * pg_var_c(V)); }
// seed region of one work item: [node][lane][seed dwords], 256-byte aligned; behind it the per-node maxima keys
// [node][4 reads][2 strands] as 64-bit slots
static inline __host__ __device__ uint64_t pg_seed_region_bytes(int V, uint32_t n_nodes)
{
    return ((uint64_t)n_nodes * 64u * pg_seed_lane_bytes(V) + 255u) & ~(uint64_t)255u;
}
static inline __host__ __device__ uint64_t pg_key_region_bytes(uint32_t n_nodes) { return ((uint64_t)n_nodes * 64u + 255u) & ~(uint64_t)255u; }
static inline __host__ __device__ uint32_t pg_ops_cap(int V) { return pg_rows(V) + 32u; }
