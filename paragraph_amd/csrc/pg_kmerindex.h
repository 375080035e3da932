// pg_kmerindex.h -- graphtools::KmerIndex on the device: shared by the exact path matching stage (pg_path.hip) and the
// KmerFilter of the count path (pg_count.hip).
//   graphtools::KmerIndex (construction, numPaths, getPaths, numUniqueKmersOverlappingNode / Edge)
//       GT!/src/graphalign/KmerIndex.cpp:76-135, 209-260
// One open-addressing hash table per graph: key = 64-bit polynomial hash of the k raw characters, value = (number of
// paths with that sequence, the FIRST such path in the reference's enumeration order).  k may differ per graph (the
// KmerFilter's auto-detected length is per graph).
#ifndef PG_KMERINDEX_H
#define PG_KMERINDEX_H
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "pg_internal.h"

constexpr uint64_t PG_HASH_B = 0x9E3779B97F4A7C15ull | 1ull;

struct PathGraphDev
{
    uint32_t node_base;  // set-wide node numbering (caller CSR)
    uint32_t n_nodes;
    uint64_t tab_off;    // into the entry table
    uint32_t tab_mask;   // capacity - 1 (capacity is a power of two), 0xFFFFFFFF = graph has no k-mers
    uint32_t k;          // k-mer length of this graph's table
    uint64_t pow_k1;     // PG_HASH_B^(k-1)
    // Presence filter: one bit per (hash >> 32) & filt_mask, set for every k-mer of the graph, 64 bits per k-mer (a k-mer that is
    // not in the graph meets a clear bit 98 times in 100; one that is never does).  The path stage's scan asks it before the
    // table: in the table "the slot is not empty" is true for a quarter to a half of all hashes.
    uint32_t filt_off;   // first 32-bit word in the set's filter array
    uint32_t filt_mask;  // bits - 1 (a power of two)
};

struct KmerEntry
{
    uint64_t hash;  // 0 = empty
    uint32_t count;
    uint32_t start_pos;
    uint32_t end_pos;
    uint32_t n_nodes;
    uint32_t pool_off;  // node ids (graph-local) of the first path with this sequence
    uint32_t pad;
};

struct pg_path_index
{
    uint32_t k = 0;  // common k (0 when the graphs differ)
    uint64_t pow_k1 = 1;
    PathGraphDev* d_graphs = nullptr;
    KmerEntry* d_table = nullptr;
    uint32_t* d_pool = nullptr;
    uint32_t* d_node_off = nullptr;
    char* d_raw = nullptr;
    uint32_t* d_succ_off = nullptr;
    uint32_t* d_succ = nullptr;
    uint8_t* d_node_uniq = nullptr;  // per set-wide node: numUniqueKmersOverlappingNode(node) > 0
    uint32_t* d_filter = nullptr;    // the graphs' presence filters (PathGraphDev::filt_off)
    void* d_block = nullptr;         // device-built index: d_graphs, d_node_off, d_raw, d_succ_off, d_succ live in this ONE block
    uint32_t* d_error = nullptr;     // device-built index: bit0 = two k-mers of a graph share a hash (the set must be refused), bit1 = internal
    void* staging = nullptr;         // its page-locked upload block, handed back when the index is freed (the build does not wait)
    size_t staging_cap = 0;
    hipEvent_t ev_built = nullptr;   // behind the build's last kernel (on the seed stream of the first path stage that uses the index)
    // device-built index: the small tables go up on the copy stream (ev_tables behind them); the two build launches wait for the set's
    // FIRST path stage and run on its seed stream in front of the path kernel (pg_path_index_ensure_built) -- on the copy stream they
    // stood in front of every lane's uploads
    hipEvent_t ev_tables = nullptr;
    bool build_pending = false;
    hipStream_t built_on = nullptr;
    uint64_t build_table_entries = 0, build_filter_words = 0;
    uint32_t build_n_chars = 0, build_n_total = 0;
    uint32_t* d_graph_of_node = nullptr;  // (inside d_block)
    uint32_t* d_pool_next = nullptr;      // (inside d_block)
    std::vector<uint32_t> h_k;       // per graph
};

// what the error word of a device-built path index says (pg_path_index::d_error)
const char* pg_path_index_error_text(uint32_t word);
// fetches that word for a batch whose stages are complete (synchronises the copy stream): PG_OK, or the build's failure
pg_status pg_path_index_check(pg_ctx* ctx, const pg_graphs* G);
// a device-built index: queues its build on `stream` if no stage has yet, otherwise makes `stream` wait for the build
hipError_t pg_path_index_ensure_built(pg_path_index* ix, hipStream_t stream);

// the tables pg_build_kmer_index makes on the host before they go up
struct PgKmerIndexHost
{
    std::vector<uint32_t> succ_off, succ, pool, h_k;
    std::vector<PathGraphDev> gd;
    std::vector<KmerEntry> table;
    std::vector<uint8_t> node_uniq;
    std::vector<uint32_t> filter;
};
const char* pg_build_kmer_index_host(const pg_graphs* G, const std::vector<int32_t>& k_per_graph, PgKmerIndexHost& tables, bool want_node_uniq);  // NULL = fine

// k_per_graph[g] > 0: that length; < 0: graphtools::findMinCoveringKmerLength(graph, -k, -k)
// (GT!/src/graphalign/KmerIndexOperations.cpp:77-113; fails with PG_ERR_UNSUPPORTED when no length 10..63 covers).
// want_node_uniq: also mark the nodes that unique k-mers overlap (the KmerFilter reads that; the path stage does not)
pg_status pg_build_kmer_index(pg_ctx* ctx, pg_graphs* G, const std::vector<int32_t>& k_per_graph, pg_path_index** out, bool want_node_uniq);

// Looks the k characters q(pos..pos+k-1) up; true only for a k-mer with EXACTLY ONE path (characters verified, so a
// 64-bit hash collision cannot fake a hit).  Q: callable int -> uint32_t raw character.
template <typename Q>
__device__ __forceinline__ bool pg_kmer_lookup_unique(
    const PathGraphDev& g, const KmerEntry* __restrict__ table, const uint32_t* __restrict__ pool, const uint32_t* __restrict__ node_off,
    const char* __restrict__ raw, uint64_t h, int pos, Q q, KmerEntry& out)
{
    if (g.tab_mask == 0xFFFFFFFFu)
        return false;
    if (h == 0)
        h = 1;
    uint32_t slot = (uint32_t)(h >> 20) & g.tab_mask;
    for (;;)
    {
        const KmerEntry e = table[g.tab_off + slot];
        if (e.hash == 0)
            return false;
        if (e.hash == h)
        {
            if (e.count != 1)
                return false;
            uint32_t ni = 0, node = pool[e.pool_off], p = e.start_pos;
            for (uint32_t c = 0; c < g.k; ++c)
            {
                if (p >= node_off[g.node_base + node + 1] - node_off[g.node_base + node])
                {
                    ++ni;
                    node = pool[e.pool_off + ni];
                    p = 0;
                }
                if ((uint32_t)(uint8_t)raw[node_off[g.node_base + node] + p] != q(pos + (int)c))
                    return false;
                ++p;
            }
            out = e;
            return true;
        }
        slot = (slot + 1) & g.tab_mask;
    }
}
#endif
