// pg_klib.h -- data shared by the klib stage's kernels (pg_klib.hip: general kernels + C ABI; pg_klib_packed.hip: the
// packed two-strand sweeps).  Internal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/paragraph_amd.h"
#include "pg_device.h"

constexpr uint32_t PG_KLIB_CIG_SMALL = 24;  // entries of a candidate's own CIGAR slot in the packed finish kernel (KlibArgs::cig_small)
constexpr int PG_KLIB_MAX_PATHS = 30;        // candidate heaps of paths + 2 entries in registers (select / pick kernels)
constexpr int PG_KLIB_MAX_PATHS_WIDE = 126;  // graphs with more paths: the same kernels with 128-entry heaps (scratch memory)

struct LPathDev
{
    uint32_t seq_off;  // into pathseq[] / pathcode[]
    uint32_t len;
    uint32_t start_off;  // into starts[] (pairs: start position, node id)
    uint32_t n_nodes;
    uint32_t meta_off;  // into pathmeta[]: one word per column (its code) + PG_META_PAD idle words
    uint32_t pad[3];
};
struct LGraphDev
{
    uint32_t path_off;
    uint32_t n_paths;
};

struct KlibItem
{  // result of one KlibAlignment::update()
    int32_t score, tb, te, qb, qe;
    uint32_t n_cigar;   // entries of the ksw_global CIGAR (len<<4 | op, op 0 M / 1 I / 2 D)
    uint32_t cig_begin; // first entry, as an index into cigars[]
    uint32_t valid;     // 1 = candidate (te >= tb), 0 = none
};

struct KlibArgs
{
    uint32_t n_reads;
    uint32_t max_paths;  // items per read = 2 * max_paths
    uint32_t len_limit;  // reads longer than this are not this stage's: they fall through to the graph aligner
    const uint32_t* base_off;
    const char* bases;
    const uint32_t* graph_of_read;
    const LGraphDev* graphs;
    const LPathDev* paths;
    const char* pathseq;
    const uint8_t* pathcode;
    const uint32_t* pathmeta;
    const uint32_t* starts;
    const uint8_t* active;
    KlibItem* items;
    uint32_t* cigars;    // general kernels: [n_items][cig_cap]; packed kernels: [n_work][cig_small] + a pool of [ovf_cap][cig_cap]
    uint32_t cig_cap;
    // packed kernels: a candidate's CIGAR goes into its own short slot (cig_small entries: nine in ten have <= 5) and moves to a
    // full-size slot of the pool (entries from ovf_base on, handed out through ovf_count) when it outgrows it; an exhausted pool
    // raises error bit 1 like an overflowing slot
    uint32_t cig_small;
    uint32_t ovf_cap;
    uint32_t ovf_base;
    uint32_t* ovf_count;
    uint8_t* z;          // [gridDim.x][z_bytes]
    uint64_t z_bytes;
    // packed kernels
    const PgWorkItem* work;  // the batch's wavefront work items (forward-graph member of each pair)
    uint32_t pair_begin;
    uint32_t* worklist;      // items whose start cell and CIGAR the pick needs
    uint32_t* work_count;
    uint32_t n_work;
    // pick kernel
    pg_result* results;
    pg_op* ops;
    unsigned long long* ops_counter;
    uint64_t ops_cap;
    uint8_t* flags;
    uint32_t* error;  // bit0 = ops overflow, bit1 = cigar slot overflow, bit2 = internal (window / start cell)
};

__device__ __forceinline__ uint32_t klib_comp_raw(uint32_t c)
{
    switch (c)
    {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return 'N';
    }
}
__device__ __forceinline__ int klib_code(uint32_t c)
{  // KlibImpl.hh translation_matrix: A/a/U/u 0, C/c 1, G/g 2, T/t 3, others 4 (index & 0x7f)
    switch (c & 0x7f)
    {
    case 'A': case 'a': case 'U': case 'u': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

// packed kernels (pg_klib_packed.hip)
hipError_t pg_klib_launch_local(int C, const KlibArgs& a, uint32_t n_pairs, hipStream_t stream);
hipError_t pg_klib_launch_finish(int C, const KlibArgs& a, uint32_t grid, hipStream_t stream);
uint64_t pg_klib_finish_z_bytes(int C);
