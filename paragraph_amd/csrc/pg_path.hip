// pg_path.hip -- exact path matching stage of the aligner cascade on the device.
//
// Replaces
//   grm::PathAligner::{setGraph,alignRead}            src/c++/lib/grm/PathAligner.cpp:70-164
//   graphtools::KmerIndex (construction + numPaths)   GT!/src/graphalign/KmerIndex.cpp:76-116, 209-223
//   graphtools::extendPathMatching                    GT!/src/graphcore/PathOperations.cpp:117-271
//   projectAlignmentOntoGraph for an all-match alignment + GraphAlignment::generateCigar
//                                                     GT!/src/graphalign/GraphAlignmentOperations.cpp:130-164
//
// Host: enumerates every length-k path of every graph (depth-first over successors in ascending id, as
// extendPathEnd does), groups them by sequence and builds one open-addressing hash table per graph:
// key = 64-bit polynomial hash of the k raw characters, value = (number of paths with that sequence, first path).
// Device: one thread per read; both strands; rolling hash over the read; for k-mers with exactly one path the
// reference's greedy exact extension (right, then left; at a node end the neighbour with the UNIQUE longest
// common prefix over the shortest neighbour's length) is replayed on the raw node sequences, eight characters
// at a time (the reverse strand on a reverse-complemented copy of the reads made once per upload).  A read is MAPPED when a match covers the whole read; more than one such match makes it
// non-unique (MAPQ 0).  HBM-bound byte work: no LDS staging is needed (a read touches <= 2L graph bytes).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/paragraph_amd.h"
#include "pg_device.h"
#include "pg_internal.h"
#include "pg_kmerindex.h"

namespace
{
constexpr uint64_t HASH_B = PG_HASH_B;

struct PathArgs
{
    uint32_t n_reads;
    uint32_t k;
    uint64_t pow_k1;  // HASH_B^(k-1)
    const uint32_t* base_off;
    const char* bases;
    char* bases_rc;  // the reads as the reverse strand sees them, at the same offsets: written by the kernel
    const uint32_t* graph_of_read;
    const PathGraphDev* graphs;
    const KmerEntry* table;
    const uint32_t* pool;
    const uint32_t* node_off;  // raw char offsets per (set-wide) node, n_total + 1
    const char* raw;
    const uint32_t* succ_off;  // per set-wide node
    const uint32_t* succ;
    const uint32_t* pred_off;
    const uint32_t* pred;
    pg_result* results;
    pg_op* ops;
    unsigned long long* ops_counter;
    uint8_t* flags;
    const uint8_t* active;  // nullptr = every read
};

__device__ __forceinline__ uint32_t comp_raw(uint32_t c)
{  // GT!/src/graphutils/SequenceOperations.cpp:66-81 (case-sensitive: anything else becomes 'N')
    switch (c)
    {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return 'N';
    }
}

// Eight characters at a time: the reference's character loops (PathOperations.cpp:128-136, 150-159, 202-210, 225-235) are runs of
// equal characters of two byte strings, forwards or backwards.  One thread owns one read and the 64 threads of a wavefront are at
// different places of their reads, so a wavefront pays for the LONGEST loop any of its threads is in at every step: with one
// character per trip the stage spent 430 000 VALU instructions per wavefront (profiles/r05_stage_counters.json: 2.8 ms, the
// lifetime of one wavefront however few there are).  Wide loads only where eight characters remain on both sides: nothing is read
// outside the two strings.
__device__ __forceinline__ uint64_t load8(const char* p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// number of leading characters x[0..n) and y[0..n) have in common
__device__ __forceinline__ uint32_t common_prefix(const char* x, const char* y, uint32_t n)
{
    uint32_t i = 0;
    while (i + 16 <= n)  // (four loads in flight: the loop is a chain of round trips, one per trip)
    {
        const uint64_t d0 = load8(x + i) ^ load8(y + i), d1 = load8(x + i + 8) ^ load8(y + i + 8);
        if (d0)
            return i + ((uint32_t)__builtin_ctzll(d0) >> 3);
        if (d1)
            return i + 8 + ((uint32_t)__builtin_ctzll(d1) >> 3);
        i += 16;
    }
    while (i + 8 <= n)
    {
        const uint64_t d = load8(x + i) ^ load8(y + i);
        if (d)
            return i + ((uint32_t)__builtin_ctzll(d) >> 3);
        i += 8;
    }
    while (i < n && x[i] == y[i])
        ++i;
    return i;
}

// number of trailing characters the strings ENDING at xe and ye (exclusive) have in common, at most n
__device__ __forceinline__ uint32_t common_suffix(const char* xe, const char* ye, uint32_t n)
{
    uint32_t i = 0;
    while (i + 16 <= n)
    {
        const uint64_t d0 = load8(xe - i - 8) ^ load8(ye - i - 8), d1 = load8(xe - i - 16) ^ load8(ye - i - 16);
        if (d0)
            return i + ((uint32_t)__builtin_clzll(d0) >> 3);
        if (d1)
            return i + 8 + ((uint32_t)__builtin_clzll(d1) >> 3);
        i += 16;
    }
    while (i + 8 <= n)
    {
        const uint64_t d = load8(xe - i - 8) ^ load8(ye - i - 8);
        if (d)
            return i + ((uint32_t)__builtin_clzll(d) >> 3);
        i += 8;
    }
    while (i < n && xe[-1 - (int)i] == ye[-1 - (int)i])
        ++i;
    return i;
}

struct Walker
{
    const PathArgs& a;
    const PathGraphDev g;
    const char* qp;  // the read as this strand sees it (the reverse strand: pg_revcomp_kernel's copy)
    int L;

    __device__ uint32_t q(int j) const { return (uint8_t)qp[j]; }
    __device__ uint32_t nlen(uint32_t node) const { return a.node_off[g.node_base + node + 1] - a.node_off[g.node_base + node]; }
    __device__ const char* nptr(uint32_t node) const { return a.raw + a.node_off[g.node_base + node]; }

    // Extends the seed path `e` anchored at read position qpos (extendPathMatching).  Outputs the final path as
    // (first node, start_pos, last node, end_pos, length, #nodes prepended, #nodes appended) and the new qpos.
    // With REC, prepended / appended node ids are written to rec_l[0..] (closest first) / rec_r[0..].
    template <bool REC>
    __device__ void extend(const KmerEntry& e, int& qpos, uint32_t& first_node, uint32_t& start_pos, uint32_t& end_pos,
                           int& length, uint32_t& n_left, uint32_t& n_right, uint32_t* rec_l, uint32_t* rec_r) const
    {
        // ---- extendPathEndMatching (PathOperations.cpp:117-189)
        uint32_t node = a.pool[e.pool_off + e.n_nodes - 1];
        uint32_t pos_in_node = e.end_pos + 1;
        int pos_in_query = qpos + (int)a.k;
        n_right = 0;
        bool moved = true;
        while (moved)
        {
            moved = false;
            const uint32_t len = nlen(node);
            if (pos_in_query < L && pos_in_node < len)
            {
                const uint32_t m = common_prefix(qp + pos_in_query, nptr(node) + pos_in_node, min((uint32_t)(L - pos_in_query), len - pos_in_node));
                moved = m != 0;
                pos_in_node += m;
                pos_in_query += (int)m;
            }
            if (pos_in_node >= len)
            {
                const uint32_t sb = a.succ_off[g.node_base + node], se = a.succ_off[g.node_base + node + 1];
                uint32_t min_size = 0xFFFFFFFFu;
                for (uint32_t s = sb; s < se; ++s)
                    min_size = min(min_size, nlen(a.succ[s]));
                uint32_t n_longest = 0, longest = 0, cur = 0;
                for (uint32_t s = sb; s < se; ++s)
                {
                    const uint32_t sn = a.succ[s];
                    const uint32_t p = common_prefix(nptr(sn), qp + pos_in_query, min(min_size, (uint32_t)(L - pos_in_query)));
                    if (p > longest)
                    {
                        longest = p;
                        cur = sn;
                        n_longest = 1;
                    }
                    else if (p == longest)
                        ++n_longest;
                }
                if (longest == 0 || n_longest != 1)
                    break;
                if (REC)
                    rec_r[n_right] = cur;
                ++n_right;
                pos_in_query += (int)longest;
                pos_in_node = longest;
                node = cur;
                moved = true;
            }
        }
        end_pos = pos_in_node - 1;
        const int end_query = pos_in_query;
        // ---- extendPathStartMatching (PathOperations.cpp:191-266)
        node = a.pool[e.pool_off];
        pos_in_node = e.start_pos;
        pos_in_query = qpos;
        n_left = 0;
        moved = true;
        while (moved)
        {
            moved = false;
            if (pos_in_query > 0 && pos_in_node > 0)
            {
                const uint32_t m = common_suffix(qp + pos_in_query, nptr(node) + pos_in_node, min((uint32_t)pos_in_query, pos_in_node));
                moved = m != 0;
                pos_in_node -= m;
                pos_in_query -= (int)m;
            }
            if (pos_in_node == 0)
            {
                const uint32_t pb = a.pred_off[g.node_base + node], pe = a.pred_off[g.node_base + node + 1];
                uint32_t min_size = 0xFFFFFFFFu;
                for (uint32_t s = pb; s < pe; ++s)
                    min_size = min(min_size, nlen(a.pred[s]));
                uint32_t n_longest = 0, longest = 0, cur = 0;
                for (uint32_t s = pb; s < pe; ++s)
                {
                    const uint32_t pn = a.pred[s];
                    const uint32_t ml = common_suffix(nptr(pn) + nlen(pn), qp + pos_in_query, min(min_size, (uint32_t)pos_in_query));
                    if (ml > longest)
                    {
                        longest = ml;
                        cur = pn;
                        n_longest = 1;
                    }
                    else if (ml == longest)
                        ++n_longest;
                }
                if (longest == 0 || n_longest != 1)
                    break;
                if (REC)
                    rec_l[n_left] = cur;
                ++n_left;
                pos_in_query -= (int)longest;
                node = cur;
                pos_in_node = nlen(node) - longest;
                moved = true;
            }
        }
        first_node = node;
        start_pos = pos_in_node;
        qpos = pos_in_query;
        length = end_query - pos_in_query;
    }

    // hash of q[p .. p + k)
    __device__ uint64_t hash_at(int p) const
    {
        uint64_t h = 0;
        uint32_t c = 0;
        for (; c + 8 <= a.k; c += 8)
        {
            const uint64_t v = load8(qp + p + (int)c);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                h = h * HASH_B + ((v >> (8 * i)) & 0xFFu) + 1;
        }
        for (; c < a.k; ++c)
            h = h * HASH_B + (uint64_t)q(p + (int)c) + 1;
        return h;
    }

    // the hash stored in the slot the k-mer with hash h goes to first: 0 = empty, the k-mer is in no path of the graph
    __device__ uint64_t first_slot_hash(uint64_t h) const
    {
        if (h == 0)
            h = 1;
        return a.table[g.tab_off + ((uint32_t)(h >> 20) & g.tab_mask)].hash;
    }

    // pg_kmer_lookup_unique (pg_kmerindex.h) with the k characters compared node by node in runs
    __device__ bool lookup(uint64_t h, int pos, KmerEntry& out) const
    {
        if (g.tab_mask == 0xFFFFFFFFu)
            return false;
        if (h == 0)
            h = 1;
        uint32_t slot = (uint32_t)(h >> 20) & g.tab_mask;
        for (;;)
        {
            const KmerEntry* ep = a.table + g.tab_off + slot;
            const uint64_t eh = ep->hash;
            if (eh == 0)
                return false;
            if (eh == h)
            {
                const KmerEntry e = *ep;
                if (e.count != 1)
                    return false;
                uint32_t ni = 0, p = e.start_pos, c = 0;
                for (;;)
                {
                    const uint32_t node = a.pool[e.pool_off + ni];
                    const uint32_t seg = min(g.k - c, nlen(node) - p);
                    if (common_prefix(nptr(node) + p, qp + pos + (int)c, seg) != seg)
                        return false;
                    c += seg;
                    if (c >= g.k)
                        break;
                    if (++ni >= e.n_nodes)
                        return false;
                    p = 0;
                }
                out = e;
                return true;
            }
            slot = (slot + 1) & g.tab_mask;
        }
    }
};

// The read as the reverse strand sees it, written at the read's own offset of the second buffer by the thread that goes on to read
// it (a kernel of its own for this waited for a free slot beside the fills like any kernel does: 0.6 ms, profiles/r05_e2e_modes.json).
__device__ __forceinline__ void reverse_complement(const char* __restrict__ src, char* __restrict__ dst, int L)
{
    int j = 0;
    for (; j + 8 <= L; j += 8)
    {
        const uint64_t v = load8(src + L - 8 - j);
        uint64_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            o |= (uint64_t)comp_raw((uint32_t)(v >> (8 * (7 - i))) & 0xFFu) << (8 * i);
        __builtin_memcpy(dst + j, &o, 8);
    }
    for (; j < L; ++j)
        dst[j] = (char)comp_raw((uint8_t)src[L - 1 - j]);
}

__global__ __launch_bounds__(64) void pg_path_kernel(PathArgs a)
{
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= a.n_reads)
        return;
    if (a.active && !a.active[r])
    {
        a.flags[r] = 0;  // (every read's flag is written: no memset in front of the kernel)
        return;
    }
    const uint32_t off = a.base_off[r];
    const int L = (int)(a.base_off[r + 1] - off);
    uint8_t flags = 0;
    if (L < (int)a.k || L == 0)
    {
        a.flags[r] = 0;
        return;
    }
    const PathGraphDev g = a.graphs[a.graph_of_read[r]];
    int n_full = 0, n_matches = 0;
    int first_strand = 0, first_pos = 0;
    const int last = L - (int)a.k;  // the last position a k-mer starts at
    if (g.tab_mask != 0xFFFFFFFFu)
        reverse_complement(a.bases + off, a.bases_rc + off, L);
    for (int strand = 0; strand < 2 && g.tab_mask != 0xFFFFFFFFu; ++strand)
    {
        Walker w{ a, g, (strand == 0 ? a.bases : a.bases_rc) + off, L };
        uint64_t h = w.hash_at(0);
        int pos = 0;
        // PathAligner.cpp:92-106 walks the read one position at a time and nearly every position is a miss (the slot its hash goes
        // to is empty).  Each step was a chain of dependent loads -- two characters for the rolling hash, then the slot -- and the
        // 64 threads of the wavefront walk in lockstep.  Eight positions per trip: their characters in two loads, their eight slots
        // probed together; the first one that is not empty gets the full lookup, the ones before it are misses, the ones behind it
        // are probed again from wherever the walk goes next.  The scan is a loop of its own, so that the threads meet at the lookup
        // each with a candidate: a wavefront pays for an extension (20 - 60 dependent loads) once per candidate of its busiest
        // thread, not once per scan step in which any of its threads had one.
        for (;;)
        {
            int at = -1;
            uint64_t h_first = 0, h_after = 0;
            while (pos <= last)
            {
                const int nb = min(8, last - pos + 1);
                uint64_t qo = 0, qi = 0;  // characters leaving / entering the window at pos, pos + 1, ...
                if (pos + 8 <= last)
                {
                    qo = load8(w.qp + pos);
                    qi = load8(w.qp + pos + (int)a.k);
                }
                else
                    for (int i = 0; i + 1 < nb; ++i)
                    {
                        qo |= (uint64_t)w.q(pos + i) << (8 * i);
                        qi |= (uint64_t)w.q(pos + (int)a.k + i) << (8 * i);
                    }
                uint64_t hs[9];
                hs[0] = h;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    hs[i + 1] = (hs[i] - (((qo >> (8 * i)) & 0xFFu) + 1) * a.pow_k1) * HASH_B + ((qi >> (8 * i)) & 0xFFu) + 1;
                uint64_t eh[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    eh[i] = w.first_slot_hash(hs[i < nb ? i : 0]);  // (every load unconditional: eight in flight, not eight round trips)
                int first = nb;
                h_after = hs[8];  // (all eight miss: nb == 8 and the next window's hash comes from all eight characters of the loads)
#pragma unroll
                for (int i = 7; i >= 0; --i)
                    if (i < nb && eh[i] != 0)
                    {
                        first = i;
                        h_first = hs[i];
                        h_after = hs[i + 1];
                    }
                if (first < nb)
                {
                    at = pos + first;
                    break;
                }
                pos += nb;
                h = h_after;
            }
            if (at < 0)
                break;
            KmerEntry e;
            int next = at + 1;
            if (w.lookup(h_first, at, e))
            {
                int qpos = at;
                uint32_t fn, sp, ep, nl, nr;
                int len;
                w.extend<false>(e, qpos, fn, sp, ep, len, nl, nr, nullptr, nullptr);
                ++n_matches;
                if (len == L)
                {
                    if (n_full == 0)
                    {
                        first_strand = strand;
                        first_pos = at;
                    }
                    ++n_full;
                }
                next = qpos + len + 1;  // PathAligner.cpp:104 + the loop increment
            }
            if (next > last)
                break;
            h = next == at + 1 ? h_after : w.hash_at(next);
            pos = next;
        }
    }
    if (n_matches)
        flags |= 2;
    if (n_full == 0)
    {
        a.flags[r] = flags;
        return;
    }
    // ---- second pass over the first full-length match: count nodes, allocate ops, record, emit ----------
    Walker w{ a, g, (first_strand == 0 ? a.bases : a.bases_rc) + off, L };
    KmerEntry e;
    w.lookup(w.hash_at(first_pos), first_pos, e);
    int qpos = first_pos;
    uint32_t fn, sp, ep, nl, nr;
    int len;
    w.extend<false>(e, qpos, fn, sp, ep, len, nl, nr, nullptr, nullptr);
    const uint32_t n_nodes = nl + e.n_nodes + nr;
    // one element per node -- except that a match run longer than an element holds (PG_OP_MAX_LEN: only reads beyond 4 095 bases
    // have one) goes out in pieces: at most L / PG_OP_MAX_LEN more elements, reserved up front
    const uint32_t extra = (uint32_t)L > PG_OP_MAX_LEN ? (uint32_t)L / PG_OP_MAX_LEN : 0u;
    const unsigned long long base = atomicAdd(a.ops_counter, (unsigned long long)(n_nodes + extra));
    pg_op* ops = a.ops + base;
    uint32_t n_ops = 0;
    // prepended nodes are produced closest-first: write them at nl-1-j; seed nodes at nl+i; appended at nl+n_seed+j
    {
        // record into the ops area itself (node ids first -- behind the reserved extra elements, so that the op words written
        // from the front never overtake the ids still to be read -- converted to op words below)
        uint32_t* ids = (uint32_t*)ops + extra;
        qpos = first_pos;
        w.extend<true>(e, qpos, fn, sp, ep, len, nl, nr, ids, ids + nl + e.n_nodes);
        for (uint32_t i = 0, j = nl ? nl - 1 : 0; i < j; ++i, --j)
        {
            const uint32_t t = ids[i];
            ids[i] = ids[j];
            ids[j] = t;
        }
        for (uint32_t i = 0; i < e.n_nodes; ++i)
            ids[nl + i] = a.pool[e.pool_off + i];
        for (uint32_t i = 0; i < n_nodes; ++i)
        {
            const uint32_t node = ids[i];
            const uint32_t lo = i == 0 ? sp : 0u;
            const uint32_t hi = i == n_nodes - 1 ? ep : w.nlen(node) - 1;
            uint32_t left = hi - lo + 1;
            do
            {
                const uint32_t piece = left > PG_OP_MAX_LEN ? PG_OP_MAX_LEN : left;
                ops[n_ops++] = PG_OP_MAKE(node, PG_OPC_M, piece);
                left -= piece;
            } while (left);
        }
    }
    pg_result res;
    res.graph_pos = (int32_t)sp;
    res.score = (int16_t)L;
    res.mapq = n_full == 1 ? 60 : 0;
    res.is_unique = n_full == 1 ? 1 : 0;
    res.returned_reverse = first_strand ? 1 : 0;
    res.multi_mask = 0;
    res.n_ops = (uint16_t)n_ops;
    res.ops_off = (uint32_t)base;
    res.strand_score[0] = first_strand ? -1 : (int16_t)L;
    res.strand_score[1] = first_strand ? (int16_t)L : -1;
    res.clipped = 0;
    res.status = PG_STATUS_PATH_ALIGNER;
    a.results[r] = res;
    a.flags[r] = flags | 1;
}

uint64_t hash_str(const char* s, uint32_t k)
{
    uint64_t h = 0;
    for (uint32_t c = 0; c < k; ++c)
        h = h * HASH_B + (uint64_t)(uint8_t)s[c] + 1;
    return h ? h : 1;
}
}  // namespace

void pg_path_index_free(pg_path_index* ix)
{
    if (!ix)
        return;
    (void)pg_dev_free(ix->d_graphs);
    (void)pg_dev_free(ix->d_table);
    (void)pg_dev_free(ix->d_pool);
    (void)pg_dev_free(ix->d_node_off);
    (void)pg_dev_free(ix->d_raw);
    (void)pg_dev_free(ix->d_succ_off);
    (void)pg_dev_free(ix->d_succ);
    (void)pg_dev_free(ix->d_node_uniq);
    delete ix;
}

template <typename T> static hipError_t up(const std::vector<T>& v, T** d, hipStream_t s)
{
    hipError_t e = pg_dev_alloc((void**)d, std::max<size_t>(v.size(), 1) * sizeof(T));
    if (e != hipSuccess || v.empty())
        return e;
    return hipMemcpyAsync(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
}

namespace
{
// One length-k path of a graph, as the enumeration meets it: hash of its characters, where it starts and ends, its nodes
// (in occ_pool).  No strings, no maps: a site's graph has ~700 of these, and a std::unordered_map<std::string, {count, path}> cost
// three allocations for each of them -- 100+ us of host time per site, more than everything else `paragraph`'s default cascade
// adds to a batch (profiles/r05_e2e_phases.json).
struct KmerOcc
{
    uint64_t hash;
    uint32_t start, end, pool_off, n_nodes;
};

struct KmerEnumerator
{
    const pg_graphs* G;
    const std::vector<uint32_t>& succ_off;
    const std::vector<uint32_t>& succ;
    uint32_t nb, k;
    std::vector<KmerOcc>& occ;
    std::vector<uint32_t>& occ_pool;
    std::vector<uint32_t> nl;

    const char* node_chars(uint32_t node) const { return G->h_seq_raw.data() + G->h_nodeseq_off[nb + node]; }
    static uint64_t extend(uint64_t h, const char* s, uint32_t n)
    {
        for (uint32_t c = 0; c < n; ++c)
            h = h * HASH_B + (uint64_t)(uint8_t)s[c] + 1;
        return h;
    }
    void leaf(uint64_t h, uint32_t start, uint32_t end)
    {
        occ.push_back(KmerOcc{ h ? h : 1, start, end, (uint32_t)occ_pool.size(), (uint32_t)nl.size() });
        occ_pool.insert(occ_pool.end(), nl.begin(), nl.end());
    }
    // depth-first = extendPathEnd (PathOperations.cpp:70-101): `have` characters hashed into h so far, now at `pos` of nl.back()
    void go(uint64_t h, uint32_t have, uint32_t start, uint32_t pos)
    {
        const uint32_t node = nl.back();
        const uint32_t room = G->h_node_len[nb + node] - pos, need = k - have;
        if (need <= room)
        {
            leaf(extend(h, node_chars(node) + pos, need), start, pos + need - 1);
            return;
        }
        h = extend(h, node_chars(node) + pos, room);
        for (uint32_t s = succ_off[nb + node]; s < succ_off[nb + node + 1]; ++s)
        {
            nl.push_back(succ[s]);
            go(h, have + room, start, 0);
            nl.pop_back();
        }
    }
    // every length-k path of the graph in the reference's order (KmerIndex.cpp:76-99: node by node, position by position).  The
    // k-mers that lie inside one node -- most of them -- roll: h' = (h - (c_out + 1) B^(k-1)) B + c_in + 1.
    void run(uint32_t n_nodes)
    {
        uint64_t pow_k1 = 1;
        for (uint32_t i = 1; i < k; ++i)
            pow_k1 *= HASH_B;
        for (uint32_t node = 0; node < n_nodes; ++node)
        {
            const uint32_t len = G->h_node_len[nb + node];
            const char* s = node_chars(node);
            uint64_t h = 0;
            bool rolling = false;
            for (uint32_t pos = 0; pos < len; ++pos)
            {
                nl.assign(1, node);
                if (pos + k <= len)
                {
                    if (!rolling)
                    {
                        h = extend(0, s + pos, k);
                        rolling = true;
                    }
                    else
                        h = (h - ((uint64_t)(uint8_t)s[pos - 1] + 1) * pow_k1) * HASH_B + (uint64_t)(uint8_t)s[pos + k - 1] + 1;
                    leaf(h, pos, pos + k - 1);
                }
                else
                    go(0, 0, pos, pos);
            }
        }
    }
};

// the k characters of an occurrence (for telling a repeated sequence from a 64-bit hash collision)
void occ_chars(const pg_graphs* G, uint32_t nb, const KmerOcc& o, const std::vector<uint32_t>& occ_pool, uint32_t k, char* out)
{
    uint32_t got = 0;
    for (uint32_t i = 0; i < o.n_nodes && got < k; ++i)
    {
        const uint32_t node = occ_pool[o.pool_off + i];
        const uint32_t from = i == 0 ? o.start : 0, len = G->h_node_len[nb + node];
        const uint32_t take = std::min(k - got, len - from);
        memcpy(out + got, G->h_seq_raw.data() + G->h_nodeseq_off[nb + node] + from, take);
        got += take;
    }
}

// The graph's table, filled in enumeration order: the first path with a sequence owns the entry (graphtools::KmerIndex keeps
// every path; only the first and the count are ever read), later ones count up.  false = two different sequences with one hash.
bool build_table(const pg_graphs* G, uint32_t nb, uint32_t k, const std::vector<KmerOcc>& occ, const std::vector<uint32_t>& occ_pool,
                 std::vector<KmerEntry>& table, size_t tab_off, uint32_t cap, std::vector<uint32_t>& pool, std::vector<uint32_t>& first_occ)
{
    first_occ.assign(cap, 0xFFFFFFFFu);
    std::vector<char> a(k), b(k);
    for (uint32_t oi = 0; oi < (uint32_t)occ.size(); ++oi)
    {
        const KmerOcc& o = occ[oi];
        uint32_t slot = (uint32_t)(o.hash >> 20) & (cap - 1);
        for (;;)
        {
            KmerEntry& e = table[tab_off + slot];
            if (e.hash == 0)
            {
                e.hash = o.hash;
                e.count = 1;
                e.start_pos = o.start;
                e.end_pos = o.end;
                e.n_nodes = o.n_nodes;
                e.pool_off = (uint32_t)pool.size();
                pool.insert(pool.end(), occ_pool.begin() + o.pool_off, occ_pool.begin() + o.pool_off + o.n_nodes);
                first_occ[slot] = oi;
                break;
            }
            if (e.hash == o.hash)
            {
                occ_chars(G, nb, occ[first_occ[slot]], occ_pool, k, a.data());
                occ_chars(G, nb, o, occ_pool, k, b.data());
                if (memcmp(a.data(), b.data(), k) != 0)
                    return false;
                ++e.count;
                break;
            }
            slot = (slot + 1) & (cap - 1);
        }
    }
    return true;
}

// KmerIndex::Impl::updateKmerCounts (KmerIndex.cpp:118-141): unique k-mers per node / per edge, from the graph's table
void unique_counts(const std::vector<KmerEntry>& table, size_t tab_off, uint32_t cap, const std::vector<uint32_t>& pool, uint32_t n_nodes,
                   std::vector<uint32_t>& node_cnt, std::unordered_map<uint64_t, uint32_t>* edge_cnt)
{
    node_cnt.assign(n_nodes, 0);
    if (edge_cnt)
        edge_cnt->clear();
    for (uint32_t s = 0; s < cap; ++s)
    {
        const KmerEntry& e = table[tab_off + s];
        if (e.hash == 0 || e.count != 1)
            continue;
        for (uint32_t i = 0; i < e.n_nodes; ++i)
        {
            node_cnt[pool[e.pool_off + i]] += 1;
            if (i && edge_cnt)
                (*edge_cnt)[((uint64_t)pool[e.pool_off + i - 1] << 32) | pool[e.pool_off + i]] += 1;
        }
    }
}
}  // namespace

// The host half of pg_build_kmer_index (no device call: tools/ubench/kmer_index_host.cpp times it on a CPU)
const char* pg_build_kmer_index_host(const pg_graphs* G, const std::vector<int32_t>& k_per_graph, PgKmerIndexHost& t, bool want_node_uniq)
{
    std::vector<uint32_t>& succ_off = t.succ_off;
    std::vector<uint32_t>& succ = t.succ;
    std::vector<PathGraphDev>& gd = t.gd;
    std::vector<KmerEntry>& table = t.table;
    std::vector<uint32_t>& pool = t.pool;
    std::vector<uint8_t>& node_uniq = t.node_uniq;
    std::vector<uint32_t>& h_k = t.h_k;
    const uint32_t n_total = (uint32_t)G->h_node_len.size();
    // successor CSR (set-wide node numbering, graph-local ids as values, ascending): counting sort over the predecessor CSR --
    // a node's successors come out ascending because the nodes are visited in ascending order
    succ_off.assign(n_total + 1, 0);
    succ.assign(G->h_pred.size(), 0);
    for (uint32_t g = 0; g < G->n_graphs; ++g)
    {
        const uint32_t nb = G->h_node_off[g], ne = G->h_node_off[g + 1];
        for (uint32_t node = nb; node < ne; ++node)
            for (uint32_t q = G->h_pred_off[node]; q < G->h_pred_off[node + 1]; ++q)
                ++succ_off[nb + G->h_pred[q] + 1];
    }
    for (uint32_t i = 0; i < n_total; ++i)
        succ_off[i + 1] += succ_off[i];
    {
        std::vector<uint32_t> fill(succ_off.begin(), succ_off.end() - 1);
        for (uint32_t g = 0; g < G->n_graphs; ++g)
        {
            const uint32_t nb = G->h_node_off[g], ne = G->h_node_off[g + 1];
            for (uint32_t node = nb; node < ne; ++node)
                for (uint32_t q = G->h_pred_off[node]; q < G->h_pred_off[node + 1]; ++q)
                    succ[fill[nb + G->h_pred[q]]++] = node - nb;
        }
    }
    gd.assign(G->n_graphs, PathGraphDev{});
    table.clear();
    pool.clear();
    node_uniq.assign(n_total, 0);
    h_k.assign(G->n_graphs, 0);
    std::vector<KmerOcc> occ;
    std::vector<uint32_t> occ_pool, first_occ;
    for (uint32_t g = 0; g < G->n_graphs; ++g)
    {
        const uint32_t nb = G->h_node_off[g], ne = G->h_node_off[g + 1];
        std::vector<uint32_t> node_cnt;
        std::unordered_map<uint64_t, uint32_t> edge_cnt;
        const size_t tab_off = table.size(), pool_mark = pool.size();
        uint32_t cap = 0;
        // the table of graph g for k-mers of length kk, appended to `table` / `pool` (dropped again when the length is only tried)
        auto make = [&](uint32_t kk, bool want_edges) -> int {
            table.resize(tab_off);
            pool.resize(pool_mark);
            occ.clear();
            occ_pool.clear();
            KmerEnumerator en{ G, succ_off, succ, nb, kk, occ, occ_pool, {} };
            en.run(ne - nb);
            cap = 4;
            while (cap < 2 * occ.size())
                cap *= 2;
            if (occ.empty())
            {
                cap = 0;
                node_cnt.assign(ne - nb, 0);
                edge_cnt.clear();
                return 0;
            }
            table.resize(tab_off + cap, KmerEntry{});
            if (!build_table(G, nb, kk, occ, occ_pool, table, tab_off, cap, pool, first_occ))
                return -1;
            if (want_node_uniq || want_edges)
                unique_counts(table, tab_off, cap, pool, ne - nb, node_cnt, want_edges ? &edge_cnt : nullptr);
            else
                node_cnt.assign(ne - nb, 0);  // (the path stage never asks which nodes have unique k-mers: that is the KmerFilter's)
            return 0;
        };
        int32_t k = k_per_graph[g];
        if (k > 0)
        {
            if (make((uint32_t)k, false) != 0)
                return "64-bit k-mer hash collision inside one graph";
        }
        else
        {
            // findMinCoveringKmerLength(graph, -k, -k): the first length 10..63 at which every node and every edge is
            // overlapped by at least -k unique k-mers
            const uint32_t need = (uint32_t)(-k);
            k = -1;
            for (int32_t kk = 10; kk < 64 && k < 0; ++kk)
            {
                if (make((uint32_t)kk, true) != 0)
                    return "64-bit k-mer hash collision inside one graph";
                bool any_below = false;
                for (uint32_t node = 0; node < ne - nb && !any_below; ++node)
                {
                    if (node_cnt[node] < need)
                        any_below = true;
                    for (uint32_t s = succ_off[nb + node]; s < succ_off[nb + node + 1] && !any_below; ++s)
                    {
                        auto it = edge_cnt.find(((uint64_t)node << 32) | succ[s]);
                        if ((it == edge_cnt.end() ? 0u : it->second) < need)
                            any_below = true;
                    }
                }
                if (!any_below)
                    k = kk;
            }
            if (k < 0)
                return "no k-mer length in 10..63 covers every node and edge with unique k-mers";
        }
        h_k[g] = (uint32_t)k;
        for (uint32_t node = 0; node < ne - nb; ++node)
            node_uniq[nb + node] = node_cnt[node] > 0;
        gd[g].node_base = nb;
        gd[g].n_nodes = ne - nb;
        gd[g].k = (uint32_t)k;
        gd[g].pow_k1 = 1;
        for (int32_t i = 1; i < k; ++i)
            gd[g].pow_k1 *= HASH_B;
        gd[g].tab_off = tab_off;
        gd[g].tab_mask = cap ? cap - 1 : 0xFFFFFFFFu;
    }
    return nullptr;
}

pg_status pg_build_kmer_index(pg_ctx* ctx, pg_graphs* G, const std::vector<int32_t>& k_per_graph, pg_path_index** out, bool want_node_uniq)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // The tables are made in vectors this thread keeps from call to call: a batch's entry table is ~10 MB, and as a fresh vector
    // it cost 58 us per graph in page faults and regrowth alone -- more than enumerating and hashing the k-mers (12 us) and
    // filling the table (8 us) together (tools/ubench/kmer_index_host.cpp).  They are uploaded before this call returns.
    static thread_local PgKmerIndexHost t;
    if (const char* why = pg_build_kmer_index_host(G, k_per_graph, t, want_node_uniq))
        return pg_fail(ctx, PG_ERR_UNSUPPORTED, why);
    std::vector<uint32_t>& succ_off = t.succ_off;
    std::vector<uint32_t>& succ = t.succ;
    std::vector<PathGraphDev>& gd = t.gd;
    std::vector<KmerEntry>& table = t.table;
    std::vector<uint32_t>& pool = t.pool;
    std::vector<uint8_t>& node_uniq = t.node_uniq;
    std::vector<uint32_t>& h_k = t.h_k;
    const std::string& raw = G->h_seq_raw;
    const std::vector<uint32_t>& noff = G->h_nodeseq_off;
    pg_path_index* ix = new pg_path_index();
    ix->h_k = h_k;
    ix->k = 0;
    if (G->n_graphs)
    {
        ix->k = h_k[0];
        for (uint32_t g = 1; g < G->n_graphs; ++g)
            if (h_k[g] != h_k[0])
                ix->k = 0;
    }
    ix->pow_k1 = 1;
    for (uint32_t i = 1; i < ix->k; ++i)
        ix->pow_k1 *= HASH_B;
    std::vector<uint32_t> node_off(noff.begin(), noff.end());
    std::vector<char> rawv(raw.begin(), raw.end());
    hipError_t e = up(gd, &ix->d_graphs, ctx->stream_copy);
    if (e == hipSuccess) e = up(table, &ix->d_table, ctx->stream_copy);
    if (e == hipSuccess) e = up(pool, &ix->d_pool, ctx->stream_copy);
    if (e == hipSuccess) e = up(node_off, &ix->d_node_off, ctx->stream_copy);
    if (e == hipSuccess) e = up(rawv, &ix->d_raw, ctx->stream_copy);
    if (e == hipSuccess) e = up(succ_off, &ix->d_succ_off, ctx->stream_copy);
    if (e == hipSuccess) e = up(succ, &ix->d_succ, ctx->stream_copy);
    if (e == hipSuccess) e = up(node_uniq, &ix->d_node_uniq, ctx->stream_copy);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream_copy);
    if (e != hipSuccess)
    {
        pg_path_index_free(ix);
        return pg_fail(ctx, PG_ERR_HIP, std::string("k-mer index upload: ") + hipGetErrorString(e));
    }
    *out = ix;
    return PG_OK;
}

extern "C" pg_status pg_graphs_build_path_index(pg_ctx* ctx, pg_graphs* G, uint32_t kmer_len)
{
    if (!ctx || !G || kmer_len == 0 || kmer_len > 250)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_graphs_build_path_index: bad argument");
    pg_path_index* ix = nullptr;
    const pg_status st = pg_build_kmer_index(ctx, G, std::vector<int32_t>(G->n_graphs, (int32_t)kmer_len), &ix, false);
    if (st != PG_OK)
        return st;
    ix->k = kmer_len;
    ix->pow_k1 = 1;
    for (uint32_t i = 1; i < kmer_len; ++i)
        ix->pow_k1 *= HASH_B;
    pg_path_index_free(G->path_index);
    G->path_index = ix;
    // the extension walks predecessors too: reuse / create the caller-indexed predecessor tables
    if (!G->d_cnt_pred_off)
    {
        HIP_TRY(ctx, up(G->h_pred_off, &G->d_cnt_pred_off, ctx->stream_copy));
        HIP_TRY(ctx, up(G->h_pred, &G->d_cnt_pred, ctx->stream_copy));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_copy));
    }
    return PG_OK;
}

/* KmerFilter's index (KmerFilter.cpp:52-76): kmer_len > 0 for every graph, or < 0 = auto-detected per graph */
extern "C" pg_status pg_graphs_build_filter_index(pg_ctx* ctx, pg_graphs* G, int32_t kmer_len, uint32_t* kmer_len_of_graph)
{
    if (!ctx || !G || kmer_len == 0 || kmer_len > 250)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_graphs_build_filter_index: bad argument");
    pg_path_index* ix = nullptr;
    const pg_status st = pg_build_kmer_index(ctx, G, std::vector<int32_t>(G->n_graphs, kmer_len), &ix, true);
    if (st != PG_OK)
        return st;
    if (kmer_len_of_graph)
        for (uint32_t g = 0; g < G->n_graphs; ++g)
            kmer_len_of_graph[g] = ix->h_k[g];
    pg_path_index_free(G->filter_index);
    G->filter_index = ix;
    return PG_OK;
}

extern "C" pg_status pg_batch_path_align(pg_ctx* ctx, pg_batch* b)
{
    if (!ctx || !b || !b->graphs)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_path_align: batch not uploaded");
    const pg_graphs* G = b->graphs;
    if (!G->path_index)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_path_align: call pg_graphs_build_path_index first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const pg_path_index* ix = G->path_index;
    b->h_counters_valid = false;
    // The path stage runs on the SEED stream (one priority level up, like the second stream), not behind the fills queued on the
    // main one nor behind the tracebacks that wait for them on the second: it is 0.2 ms of work per batch whose outcome -- after
    // the count pass and the hand-over, which follow it onto this stream -- decides what the batch's fills are.
    const unsigned turn = ctx->seed_streams > 1 ? ctx->seed_turn++ % (unsigned)ctx->seed_streams : 0u;
    if (turn && !ctx->stream_seed_more[turn - 1])
        HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->stream_seed_more[turn - 1], hipStreamNonBlocking, ctx->side_priority));
    const hipStream_t ps = turn ? ctx->stream_seed_more[turn - 1] : ctx->stream_seed;
    b->seed_stream = ps;
    b->seed_chain = true;
    {
        const pg_status cp = pg_cascade_prepare_early(ctx, b);  // (the hand-over's tables go up beside the kernel below)
        if (cp != PG_OK)
            return cp;
    }
    HIP_TRY(ctx, pg_stage_begin_on(ctx, b, ps));
    // (no memsets in front of the kernel: it writes the flag of every read, and the counter is still zero from the upload when this is
    // the batch's first stage -- every dispatch of a seed chain waits for a wavefront slot beside the fills)
    if (!b->ops_counter_fresh)
        HIP_TRY(ctx, hipMemsetAsync(b->d_ops_counter, 0, sizeof(unsigned long long), ps));
    b->ops_counter_fresh = false;
    if (b->cap_bases_rc < b->cap_bases)  // (the reverse strand's copy of the reads: each thread of the kernel writes its own)
    {
        b->park(b->d_bases_rc);  // (a stage call never waits for the batch: pg_internal.h, parked_blocks)
        b->d_bases_rc = nullptr;
        b->cap_bases_rc = 0;
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_bases_rc, b->cap_bases));
        b->cap_bases_rc = b->cap_bases;
    }
    PathArgs a{};
    a.n_reads = b->n_reads;
    a.bases_rc = b->d_bases_rc;
    a.k = ix->k;
    a.pow_k1 = ix->pow_k1;
    a.base_off = b->d_base_off;
    a.bases = b->d_bases;
    a.graph_of_read = b->d_graph_of_read;
    a.graphs = ix->d_graphs;
    a.table = ix->d_table;
    a.pool = ix->d_pool;
    a.node_off = ix->d_node_off;
    a.raw = ix->d_raw;
    a.succ_off = ix->d_succ_off;
    a.succ = ix->d_succ;
    a.pred_off = G->d_cnt_pred_off;
    a.pred = G->d_cnt_pred;
    a.results = b->d_results;
    a.ops = b->d_ops;
    a.ops_counter = b->d_ops_counter;
    a.flags = b->d_path_flags;
    a.active = b->has_active ? b->d_active : nullptr;
    if (b->n_reads)
    {
        hipLaunchKernelGGL(pg_path_kernel, dim3((b->n_reads + 63) / 64), dim3(64), 0, ps, a);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, pg_stage_end_on(ctx, b, ps));
    return PG_OK;
}

extern "C" pg_status pg_batch_download_path_flags(pg_ctx* ctx, pg_batch* b, uint8_t* flags)
{
    if (!ctx || !b || (b->n_reads && !flags))
        return PG_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, pg_batch_wait(ctx, b));
    if (b->n_reads)
        HIP_TRY(ctx, hipMemcpyAsync(flags, b->d_path_flags, b->n_reads, hipMemcpyDeviceToHost, ctx->stream_copy));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_copy));
    return PG_OK;
}
