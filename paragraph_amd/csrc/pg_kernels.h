// pg_kernels.h -- kernel argument blocks + launch wrappers (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/paragraph_amd.h"
#include "pg_device.h"

struct PgFillArgs
{
    const PgWorkItem* items;
    uint32_t item_begin;
    uint32_t both_dirs;  // 1: forward-graph and reversed-graph workgroups alternate in runs of eight (pg_fill_kernel); 0: forward only;
                         // 2: reversed only (the lean pass's first launch)
    const PgInstItem* inst;  // lean forward pass: instance item of pair p = inst[item_begin / 2 + p] (its summaries take the slots of work item item_begin + 2 p)
    uint32_t n_pairs;    // work-item pairs of this launch
    const PgGraphDev* graphs;
    const PgNode* nodes;
    const uint32_t* preds;
    const uint32_t* colmeta;
    const uint32_t* base_off;
    const char* bases;
    uint8_t* workspace;
    PgFillSummary* fillsum;  // [(item * 4 + group) * 2 + strand]
};

struct PgTraceArgs
{
    const PgWorkItem* items;
    uint32_t pair_begin;  // first (fwd,rev) item pair of this launch; fwd item = 2*pair, rev = 2*pair+1
    uint32_t n_pairs;
    int C;           // variant code of the chunk
    uint32_t wide32; // wide variants: the fill ran 32 lanes per read (two wavefronts per work item, pg_fill.hip)
    uint32_t flags;  // PG_AF_*
    const PgGraphDev* graphs;
    const PgNode* nodes;
    const uint32_t* preds;
    const char* seqchars;
    const uint32_t* base_off;
    const char* bases;
    const uint8_t* workspace;
    uint8_t* workspace_rw;  // the same block: the CIGAR scratch of a work-item pair sits behind its seed regions
    const PgFillSummary* fillsum;
    pg_result* results;     // [read]
    pg_op* ops;             // compact output
    unsigned long long* ops_counter;
    // ---- the lean pass (pg_launch_trace_lean): the forward fills are instance items made by pg_lean_build_kernel
    const PgInstItem* inst;
    const struct PgPlanSegment* segments;  // the full plan's (group, chunk) runs: the instance item of (pair, read) follows from its run
    uint32_t n_segments;
    const uint32_t* yloc;                  // per read: where the forward fill of its OTHER strand is (PG_NONE: not run)
    // first look only: a read whose record needs that fill after all gets it queued (an instance behind its run's others) and is listed
    PgInstItem* inst_rw;
    uint32_t* yloc_rw;
    uint32_t* extra;                       // per pair slot: the run's count of such instances (pg_lean_build_kernel started it)
    const uint32_t* group_count;
    uint32_t* ucount;                      // the chunk's list of reads for the second look, and its length
    uint32_t* ulist;
    uint32_t fused;                        // the forward fills were made by pg_fill_lean_fused_kernel: X of (pair, read) sits in the leader pair's slot, half = the pair's parity in its run
};

// The lean pass's pick: one thread per work-item pair of a chunk, behind its reversed-graph fills (pg_api.hip)
struct PgLeanBuildArgs
{
    uint32_t pair_begin, n_pairs;
    const PgWorkItem* items;
    const PgFillSummary* fillsum;
    const struct PgPlanSegment* segments;
    uint32_t n_segments;
    const uint32_t* group_count;
    PgInstItem* inst;
    uint32_t* extra;  // per pair slot: instances beyond the first per read that the run starting there has been given
    uint32_t* yloc;
    uint32_t* ucount; // the chunk's second-look counter (zeroed here)
};
hipError_t pg_launch_lean_build(const PgLeanBuildArgs& args, hipStream_t stream);
hipError_t pg_launch_trace_lean(const PgTraceArgs& args, hipStream_t stream);
hipError_t pg_launch_trace_lean2(const PgTraceArgs& args, hipStream_t stream);  // the reads the first look listed

hipError_t pg_launch_fill(int V, const PgFillArgs& args, uint32_t n_pairs, bool revg, bool wide32, hipStream_t stream);
// the lean pass (byte variants only): mode 2 = the reversed-graph fills of the work items, mode 3 = forward-graph fills of args.inst,
// mode 4 = of those instance items only that the traceback's first look marked (PgInstItem::pad)
hipError_t pg_launch_fill_lean(int V, const PgFillArgs& args, uint32_t n_pairs, int mode, hipStream_t stream);
// the lean stage in one launch: reversed-graph fills, pick and forward-graph fills of two pairs per wavefront (pg_fill.hip)
hipError_t pg_launch_fill_lean_fused(int V, const PgFillArgs& args, const PgPlanSegment* segments, uint32_t n_segments, const uint32_t* group_count,
                                     PgInstItem* inst, uint32_t* yloc, uint32_t* ucount, uint32_t* ulist, uint32_t seg_begin, uint32_t n_seg,
                                     uint32_t n_pairs, hipStream_t stream);
hipError_t pg_launch_trace(const PgTraceArgs& args, hipStream_t stream);
