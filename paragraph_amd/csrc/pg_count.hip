// pg_count.hip -- the count path on the device: read filters, per-read node/edge/sequence support,
// per-fragment union and per-site counters.
//
// Replaces (per site, per read, with string CIGAR re-parsing and std::set/std::map on the host)
//   NonUniq / BadAlign filters as applied by CompositeAligner::alignRead
//        src/c++/lib/paragraph/readfilters/NonUniq.hh:48-52, BadAlign.hh:62-73,
//        src/c++/lib/paragraph/ReadFilter.cpp:43-90, src/c++/lib/grm/CompositeAligner.cpp:152-175
//   decodeGraphAlignment + Alignment counters   GT!/src/graphalign/GraphAlignmentOperations.cpp:67-127,
//                                               GT!/src/graphalign/LinearAlignment.cpp:49-131
//   nodefilter / edgefilter                     src/c++/lib/paragraph/Disambiguation.cpp:212-296
//   disambiguateReads + PathFamily::containsPath   Disambiguation.cpp:82-142, GT!/src/graphcore/PathFamily.cpp:89-108
//   readsToFragments / Fragment::addRead        src/c++/lib/common/Fragment.cpp:34-69, 141-181
//   countNodes / countEdges / countPathFamilies src/c++/lib/paragraph/ReadCounting.cpp:52-127
//
// Two kernels, both pure integer/byte work on data that is already in HBM (pg_result + pg_op from the
// traceback kernel): one thread per read (support), one thread per fragment (union + atomic counters).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/paragraph_amd.h"
#include "pg_device.h"
#include "pg_internal.h"
#include "pg_kmerindex.h"

namespace
{
struct CountArgs
{
    uint32_t n_reads;
    pg_count_params prm;
    const pg_result* results;
    const pg_op* ops;
    const uint32_t* base_off;
    const uint32_t* graph_of_read;
    const uint8_t* is_rev;
    const PgCountGraph* graphs;
    const uint32_t* pred_off;   // set-wide node numbering
    const uint32_t* pred;
    const uint32_t* node_len;
    const uint64_t* label_mask;  // per predecessor entry x label_words
    const uint64_t* out_mask;    // per node x label_words
    const uint64_t* in_mask;
    uint32_t label_words;        // 64-bit words per label set (1 unless a graph of the set has more than 64 labels)
    uint32_t frag_lds_counters;  // dwords of dynamic LDS a pg_fragment_kernel block has
    uint64_t* label_ext;         // [read][label_words - 1]: words 1.. of the reads' sets (word 0 is in pg_read_support)
    pg_read_support* support;
    uint32_t* path;
    unsigned long long* path_counter;
    // fragments
    uint32_t n_frags;
    const uint32_t* frag_off;
    const uint32_t* frag_reads;
    uint32_t* counts;
    pg_count_layout lay;
    // KmerFilter (prm.use_kmer_filter)
    const char* bases;
    const PathGraphDev* kf_graphs;
    const KmerEntry* kf_table;
    const uint32_t* kf_pool;
    const uint32_t* kf_node_off;
    const char* kf_raw;
    const uint8_t* kf_node_uniq;
};

__device__ __forceinline__ uint32_t kf_comp(uint32_t c)
{  // GT!/src/graphutils/SequenceOperations.cpp:66-81
    switch (c)
    {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return 'N';
    }
}

// readfilters::KmerFilter::filterRead (src/c++/lib/paragraph/readfilters/KmerFilter.cpp:78-139) on the read as the
// aligner left it (bases reverse-complemented when the alignment is to the reverse strand).  Returns 0 = keep,
// 3 = kmer_tooshort, 4 = kmer_uncov.  The read's unique k-mers are looked up once per alignment node that has unique
// k-mers (no per-thread node set: a read crosses a handful of nodes).
__device__ uint32_t kmer_filter(const CountArgs& a, const pg_result& res, uint32_t r, uint32_t graph, int L)
{
    const PathGraphDev g = a.kf_graphs[graph];
    const int k = (int)g.k;
    const pg_op* ops = a.ops + res.ops_off;
    // numClipped of the first / last node alignment (all of that node's soft clips, on either side)
    const uint32_t first = PG_OP_NODE(ops[0]), last = PG_OP_NODE(ops[res.n_ops - 1]);
    int sc_left = 0, sc_right = 0;
    for (uint32_t e = 0; e < res.n_ops && PG_OP_NODE(ops[e]) == first; ++e)
        if (PG_OP_CODE(ops[e]) == PG_OPC_S)
            sc_left += (int)PG_OP_LEN(ops[e]);
    for (uint32_t e = res.n_ops; e > 0 && PG_OP_NODE(ops[e - 1]) == last; --e)
        if (PG_OP_CODE(ops[e - 1]) == PG_OPC_S)
            sc_right += (int)PG_OP_LEN(ops[e - 1]);
    if (L - sc_left - sc_right < k)
        return 3;
    const char* b = a.bases + a.base_off[r];
    const bool rev = res.returned_reverse != 0;
    auto q = [&](int j) -> uint32_t { return rev ? kf_comp((uint8_t)b[L - 1 - j]) : (uint32_t)(uint8_t)b[j]; };
    const int p_first = sc_left, p_last = L - sc_right - k;  // inclusive
    // scan(node): does a unique read k-mer exist whose path contains `node` (node < 0: any unique k-mer at all)
    auto scan = [&](int node) -> bool {
        uint64_t h = 0;
        for (int c = 0; c < k; ++c)
            h = h * PG_HASH_B + (uint64_t)q(p_first + c) + 1;
        for (int pos = p_first;; ++pos)
        {
            KmerEntry e;
            if (pg_kmer_lookup_unique(g, a.kf_table, a.kf_pool, a.kf_node_off, a.kf_raw, h, pos, q, e))
            {
                if (node < 0)
                    return true;
                for (uint32_t i = 0; i < e.n_nodes; ++i)
                    if (a.kf_pool[e.pool_off + i] == (uint32_t)node)
                        return true;
            }
            if (pos == p_last)
                return false;
            h = (h - ((uint64_t)q(pos) + 1) * g.pow_k1) * PG_HASH_B + (uint64_t)q(pos + k) + 1;
        }
    };
    // with no unique k-mer in the read the reference reports "kmer_uncov" even when no node needs covering
    if (!scan(-1))
        return 4;
    uint32_t cur = 0xFFFFFFFFu;
    for (uint32_t e = 0; e < res.n_ops; ++e)
    {
        const uint32_t nd = PG_OP_NODE(ops[e]);
        if (nd == cur)
            continue;
        cur = nd;
        if (a.kf_node_uniq[g.node_base + nd] && !scan((int)nd))
            return 4;
    }
    return 0;
}

struct NodeAln
{
    uint32_t node;
    uint32_t m, x, n, ins, del, s;
    __device__ uint32_t rlen() const { return m + x + n + del; }
    __device__ uint32_t qlen() const { return m + x + n + ins + s; }
};

__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

__device__ bool node_ok(const NodeAln& a, uint32_t node_len, uint32_t L, bool use)
{  // Disambiguation.cpp:212-242
    if (!use)
        return true;
    const bool is_short = node_len < L / 2;
    const uint32_t nonmatch = a.x + a.s;
    const uint32_t indel = a.ins + a.del;
    if (is_short && (nonmatch > 0 || indel > 0))
        return false;
    return nonmatch + indel <= L / 2;
}

__device__ bool edge_ok(const NodeAln& p, const NodeAln& c, uint32_t len1, uint32_t len2, uint32_t L, bool use)
{  // Disambiguation.cpp:244-296
    if (!use)
        return true;
    const uint32_t mno = L / 10 + 1;
    bool st = p.m >= umin(p.rlen(), mno) && c.m >= umin(c.rlen(), mno);
    if (st)
        st = (p.qlen() < p.rlen() * 2) && (c.qlen() < c.rlen() * 2);
    if (st)
        st = ((int32_t)p.m >= (int32_t)umin(len1, mno)) && ((int32_t)c.m >= (int32_t)umin(len2, mno));
    return st;
}

__global__ __launch_bounds__(64) void pg_support_kernel(CountArgs a)
{
    const uint32_t r0 = blockIdx.x * 64u + threadIdx.x;
    // out-of-range lanes shadow the last read with every side effect disabled, so that the wave-wide
    // ballots below always see all 64 lanes
    const bool live = r0 < a.n_reads;
    const uint32_t r = live ? r0 : a.n_reads - 1;
    pg_result res = a.results[r];
    if (!live)
        res.status = 0xFFFF;
    const uint32_t L = a.base_off[r + 1] - a.base_off[r];
    const PgCountGraph cg = a.graphs[a.graph_of_read[r]];
    uint32_t* tally = a.counts + a.lay.tally_base + 4ull * a.graph_of_read[r];
    pg_read_support sup;
    sup.label_mask = 0;
    sup.path_off = 0;
    sup.n_path = 0;
    sup.status = 0;
    sup.filter = 0;
    // tallies: one atomic per wavefront when all its (active) lanes belong to the same graph (the usual case)
    auto tally_add = [&](int which, bool pred) {
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.graph_of_read[r]);
        const bool uniform = __all(a.graph_of_read[r] == g0);
        if (uniform)
        {
            const unsigned long long m = __ballot(pred);
            if (m != 0 && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m))
                atomicAdd(&tally[which], (uint32_t)__builtin_popcountll(m));
        }
        else if (pred)
            atomicAdd(&tally[which], 1u);
    };
    const bool aligned = !(L == 0 || (res.status & 0xFFu) != 0 || res.n_ops == 0);
    tally_add(0, aligned);
    if (!aligned)
    {
        // skipped (Align.cpp:74-77) or degenerate all-zero alignment (no CIGAR): never counted
        if (L != 0 && (res.status & 0xFFu) == 2)
            sup.status = 3;
        if (live)
            a.support[r] = sup;
        return;
    }
    // ---- CompositeAligner.cpp:156-172: MAPPED, then the filter chain may turn it into BAD_ALIGN
    sup.status = 1;
    if (a.prm.remove_nonuniq && !res.is_unique)
    {
        sup.status = 2;
        sup.filter = 1;
    }
    else
    {
        // BadAlign.hh:62-73: queryLength of the whole alignment = read length
        const double thr = round(a.prm.bad_align_frac * (double)L);
        if ((double)(L - res.clipped) < thr)
        {
            sup.status = 2;
            sup.filter = 2;
        }
    }
    // decodeGraphAlignment would throw on an invalid path (Path.cpp:86-190): start inside the first node
    const uint32_t first_node = PG_OP_NODE(a.ops[res.ops_off]);
    if (res.graph_pos < 0 || (uint32_t)res.graph_pos >= a.node_len[cg.node_base + first_node])
        sup.status = 3;
    if (sup.status == 1 && a.prm.use_kmer_filter)
    {
        const uint32_t kf = kmer_filter(a, res, r, a.graph_of_read[r], (int)L);
        if (kf)
        {
            sup.status = 2;
            sup.filter = (uint8_t)kf;
        }
    }
    tally_add(3, sup.status == 2 && sup.filter == 1);
    tally_add(2, sup.status == 2 && sup.filter == 2);
    tally_add(1, sup.status == 1);
    if (sup.status != 1)
    {
        a.support[r] = sup;
        return;
    }

    // ---- walk the CIGAR node by node (streaming: previous + current node alignment) -----------------
    uint32_t n_path = 0;
    {
        uint32_t cur = 0xFFFFFFFFu;
        for (uint32_t e = 0; e < res.n_ops; ++e)
        {
            const uint32_t nd = PG_OP_NODE(a.ops[res.ops_off + e]);
            if (nd != cur)
            {
                ++n_path;
                cur = nd;
            }
        }
    }
    const uint32_t poff = (uint32_t)atomicAdd(a.path_counter, (unsigned long long)n_path);
    const bool use = a.prm.use_support_filters != 0;
    NodeAln prev{}, cur{};
    bool have_prev = false, have_cur = false;
    uint32_t k = 0;
    // label sets of up to PG_LABEL_WORDS words (W = 1 on all but graph sets with more than 64 labels on a graph)
    const uint32_t W = a.label_words;
    uint64_t overlapped[PG_LABEL_WORDS] = {}, matched[PG_LABEL_WORDS] = {}, failed[PG_LABEL_WORDS] = {};
    auto finish_node = [&]() {
        // edge (prev -> cur) and node cur
        uint32_t entry = cur.node;
        const uint32_t gn = cg.node_base + cur.node;
        if (have_prev)
        {
            // predecessor entry of the edge prev.node -> cur.node
            uint32_t eq = 0xFFFFFFFFu;
            for (uint32_t q = a.pred_off[gn]; q < a.pred_off[gn + 1]; ++q)
                if (a.pred[q] == prev.node)
                    eq = q;
            const bool eok = edge_ok(prev, cur, a.node_len[cg.node_base + prev.node], a.node_len[gn], L, use);
            for (uint32_t w = 0; w < W; ++w)
            {
                const uint64_t emask = eq != 0xFFFFFFFFu ? a.label_mask[(size_t)eq * W + w] : 0ull;
                const uint64_t touch = a.out_mask[(size_t)(cg.node_base + prev.node) * W + w] | a.in_mask[(size_t)gn * W + w];
                matched[w] |= emask;
                failed[w] |= (~emask) & touch;
                if (eok)
                    overlapped[w] |= emask;
            }
            if (eok)
                entry |= 1u << 31;
        }
        if (node_ok(cur, a.node_len[gn], L, use))
            entry |= 1u << 30;
        a.path[poff + k] = entry;
        ++k;
    };
    for (uint32_t e = 0; e < res.n_ops; ++e)
    {
        const pg_op o = a.ops[res.ops_off + e];
        const uint32_t nd = PG_OP_NODE(o), code = PG_OP_CODE(o), len = PG_OP_LEN(o);
        if (!have_cur || nd != cur.node)
        {
            if (have_cur)
            {
                finish_node();
                prev = cur;
                have_prev = true;
            }
            cur = NodeAln{};
            cur.node = nd;
            have_cur = true;
        }
        switch (code)
        {
        case PG_OPC_M: cur.m += len; break;
        case PG_OPC_X: cur.x += len; break;
        case PG_OPC_N: cur.n += len; break;
        case PG_OPC_I: cur.ins += len; break;
        case PG_OPC_D: cur.del += len; break;
        case PG_OPC_S: cur.s += len; break;
        default: break;
        }
    }
    if (have_cur)
        finish_node();
    // PathFamily::containsPath for every label overlapped by a supported edge
    sup.label_mask = overlapped[0] & matched[0] & ~failed[0];
    for (uint32_t w = 1; w < W; ++w)
        a.label_ext[(size_t)r * (W - 1) + (w - 1)] = overlapped[w] & matched[w] & ~failed[w];
    sup.path_off = poff;
    sup.n_path = (uint16_t)n_path;
    a.support[r] = sup;
}

constexpr int FRAG_BLOCK = 256;
constexpr uint32_t FRAG_LDS_COUNTERS = 4096;  // at most: a graph that needs more counts with global atomics

// One thread per fragment.  Fragments are sorted by graph, so a block usually works on ONE graph: its
// counters are then accumulated in LDS and flushed with one global atomic per touched counter per block
// (a single hot site would otherwise serialise millions of atomics on a handful of addresses).
// The LDS is dynamic, sized by the launcher to what the largest graph of the set needs (a 3-node deletion graph: 48 counters):
// this kernel runs on the second stream under the next chunk's fill, whose 16 wavefronts per CU hold all 160 KB of LDS but for
// what a retiring one frees -- a block that asked for a fixed 16 KB waited for two of them to retire on the same CU while the
// dispatcher kept handing the freed 10 KB to the next fill wavefront (8 ms under the fill for 62 us of work).
__global__ __launch_bounds__(FRAG_BLOCK) void pg_fragment_kernel(CountArgs a)
{
    extern __shared__ uint32_t lcnt[];
    const uint32_t f = blockIdx.x * FRAG_BLOCK + threadIdx.x;
    const uint32_t f_first = blockIdx.x * FRAG_BLOCK;
    const uint32_t f_last = min(f_first + FRAG_BLOCK, a.n_frags) - 1;
    const uint32_t g_first = a.graph_of_read[a.frag_reads[a.frag_off[f_first]]];
    const uint32_t g_last = a.graph_of_read[a.frag_reads[a.frag_off[f_last]]];
    const PgCountGraph bg = a.graphs[g_first];
    const uint32_t e_base = a.pred_off[bg.node_base];
    const uint32_t n_edges_g = a.pred_off[bg.node_base + bg.n_nodes] - e_base;
    const uint32_t n_seq_g = bg.n_labels <= PG_MAX_SEQ_TABLE_LABELS ? (1u << bg.n_labels) : 0u;
    const uint32_t l_edge = 4 * bg.n_nodes, l_seq = l_edge + 4 * n_edges_g, l_total = l_seq + 4 * n_seq_g;
    const bool use_lds = g_first == g_last && l_total <= a.frag_lds_counters;
    if (use_lds)
    {
        for (uint32_t i = threadIdx.x; i < l_total; i += FRAG_BLOCK)
            lcnt[i] = 0;
    }
    __syncthreads();

    // Pass 1: the fragment's read tallies and label set.  Pass 2: every node / edge supported by at least one of its reads gets
    // {1, reads, fwd, rev} once (Fragment::addRead unions the supports, Fragment.cpp:141-181): an element of read q counts
    // unless an earlier read of the fragment supports it too.  A path visits its nodes in ascending id order, so "does read
    // q2 support node x" is a short scan -- no per-thread set, hence no limit on the nodes a fragment may touch.
    uint32_t n = 0, fwd = 0, rev = 0;
    uint64_t labels = 0;
    uint32_t graph = g_first;
    uint32_t b = 0, e = 0;
    if (f < a.n_frags)
    {
        b = a.frag_off[f];
        e = a.frag_off[f + 1];
        for (uint32_t q = b; q < e; ++q)
        {
            const uint32_t r = a.frag_reads[q];
            const pg_read_support sup = a.support[r];
            if (sup.status != 1)
                continue;  // only MAPPED reads survive alignReads (Align.cpp:81-84,155)
            graph = a.graph_of_read[r];
            ++n;
            const bool read_rev = a.is_rev[r] != 0;
            const pg_result rr = a.results[r];
            // gssw stage: is_reverse_strand() != return_reverse (GraphAligner.cpp:358-359); path stage: the match's
            // own strand (PathAligner.cpp:121-129)
            const bool graph_rev = (rr.status & PG_STATUS_PATH_ALIGNER) ? rr.returned_reverse != 0
                                                                        : read_rev != (rr.returned_reverse != 0);
            if (graph_rev)
                ++rev;
            else
                ++fwd;
            labels |= sup.label_mask;
        }
    }
    if (n != 0)
    {
        const PgCountGraph cg = a.graphs[graph];
        auto add = [&](uint32_t* c) {  // ReadCounting.cpp:52-69
            atomicAdd(&c[0], 1u);
            atomicAdd(&c[1], n);
            atomicAdd(&c[2], fwd);
            atomicAdd(&c[3], rev);
        };
        // does an earlier MAPPED read of the fragment (positions [b, q)) support node `nd` / the edge pnode -> nd ?
        auto seen_before = [&](uint32_t q, uint32_t nd, bool want_edge, uint32_t pnode) -> bool {
            for (uint32_t q2 = b; q2 < q; ++q2)
            {
                const pg_read_support s2 = a.support[a.frag_reads[q2]];
                if (s2.status != 1)
                    continue;
                uint32_t prev = 0;
                for (uint32_t k = 0; k < s2.n_path; ++k)
                {
                    const uint32_t en = a.path[s2.path_off + k];
                    const uint32_t x = PG_PATH_NODE(en);
                    if (x == nd)
                    {
                        if (!want_edge ? PG_PATH_NODE_OK(en) != 0 : (k > 0 && PG_PATH_EDGE_OK(en) && prev == pnode))
                            return true;
                        break;
                    }
                    if (x > nd)
                        break;
                    prev = x;
                }
            }
            return false;
        };
        for (uint32_t q = b; q < e; ++q)
        {
            const pg_read_support sup = a.support[a.frag_reads[q]];
            if (sup.status != 1)
                continue;
            uint32_t pnode = 0;
            for (uint32_t k = 0; k < sup.n_path; ++k)
            {
                const uint32_t en = a.path[sup.path_off + k];
                const uint32_t nd = PG_PATH_NODE(en);
                if (PG_PATH_NODE_OK(en) && !seen_before(q, nd, false, 0))
                    add(use_lds ? lcnt + 4 * nd : a.counts + a.lay.node_base + 4ull * (cg.node_base + nd));
                if (k > 0 && PG_PATH_EDGE_OK(en))
                {
                    const uint32_t gn = cg.node_base + nd;
                    uint32_t eidx = 0xFFFFFFFFu;
                    for (uint32_t p = a.pred_off[gn]; p < a.pred_off[gn + 1]; ++p)
                        if (a.pred[p] == pnode)
                            eidx = p;
                    if (eidx != 0xFFFFFFFFu && !seen_before(q, nd, true, pnode))
                        add(use_lds ? lcnt + l_edge + 4 * (eidx - e_base) : a.counts + a.lay.edge_base + 4ull * eidx);
                }
                pnode = nd;
            }
        }
        if (labels != 0 && cg.n_labels <= PG_MAX_SEQ_TABLE_LABELS)
            add(use_lds ? lcnt + l_seq + 4 * (uint32_t)labels : a.counts + a.lay.seq_base + 4ull * (cg.seq_base + labels));
    }
    __syncthreads();
    if (use_lds)
    {
        for (uint32_t i = threadIdx.x; i < l_total; i += FRAG_BLOCK)
        {
            const uint32_t v = lcnt[i];
            if (v == 0)
                continue;
            uint32_t* dst;
            if (i < l_edge)
                dst = a.counts + a.lay.node_base + 4ull * bg.node_base + i;
            else if (i < l_seq)
                dst = a.counts + a.lay.edge_base + 4ull * e_base + (i - l_edge);
            else
                dst = a.counts + a.lay.seq_base + 4ull * bg.seq_base + (i - l_seq);
            atomicAdd(dst, v);
        }
    }
}

}  // namespace

static void layout_of(const pg_graphs* G, pg_count_layout* lay)
{
    lay->n_nodes = G->h_pred_off.size() - 1;
    lay->n_edges = G->h_pred.size();
    lay->n_seq_slots = G->h_seq_off.empty() ? 0 : G->h_seq_off.back();
    lay->n_graphs = G->n_graphs;
    lay->node_base = 0;
    lay->edge_base = 4 * lay->n_nodes;
    lay->seq_base = lay->edge_base + 4 * lay->n_edges;
    lay->tally_base = lay->seq_base + 4 * lay->n_seq_slots;
    lay->n_counters = lay->tally_base + 4 * lay->n_graphs;
}

extern "C" pg_status pg_graphs_set_labels_wide(
    pg_ctx* ctx, pg_graphs* G, const uint64_t* label_words_of_pred, uint32_t words, const uint32_t* n_labels)
{
    if (!ctx || !G)
        return PG_ERR_INVALID;
    if (words < 1 || words > PG_LABEL_WORDS)
        return pg_fail(ctx, PG_ERR_UNSUPPORTED, "label sets of more than 256 labels (4 words)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n_nodes = G->h_pred_off.size() - 1, n_pred = G->h_pred.size(), W = words;
    std::vector<uint64_t> lm(n_pred * W, 0), outm(n_nodes * W, 0), inm(n_nodes * W, 0);
    G->h_n_labels.assign(G->n_graphs, 0);
    G->h_seq_off.assign(G->n_graphs + 1, 0);
    std::vector<PgCountGraph> cg(G->n_graphs);
    uint32_t frag_need = 0;
    for (uint32_t g = 0; g < G->n_graphs; ++g)
    {
        const uint32_t nl = n_labels ? n_labels[g] : 0;
        if (nl > 64u * words)
            return pg_fail(ctx, PG_ERR_UNSUPPORTED, words == 1 ? "more than 64 labels on a graph (pg_graphs_set_labels_wide takes up to 256)"
                                                                 : "more labels on a graph than the label words hold");
        G->h_n_labels[g] = nl;
        const uint32_t nb = G->h_node_off[g], ne = G->h_node_off[g + 1];
        for (uint32_t node = nb; node < ne; ++node)
            for (uint32_t q = G->h_pred_off[node]; q < G->h_pred_off[node + 1]; ++q)
                for (uint32_t w = 0; w < words; ++w)
                {
                    const uint64_t m = label_words_of_pred ? label_words_of_pred[(size_t)q * W + w] : 0;
                    // bits of this word that name a label of the graph
                    const uint64_t valid = nl >= 64u * (w + 1) ? ~0ull : (nl > 64u * w ? ((1ull << (nl - 64u * w)) - 1) : 0ull);
                    if (m & ~valid)
                        return pg_fail(ctx, PG_ERR_INVALID, "label bit outside n_labels");
                    lm[(size_t)q * W + w] = m;
                    inm[(size_t)node * W + w] |= m;
                    outm[(size_t)(nb + G->h_pred[q]) * W + w] |= m;
                }
        {
            // what a pg_fragment_kernel block counting this graph in LDS needs: {count, reads, fwd, rev} per node, edge and sequence set
            const uint64_t need = 4ull * (ne - nb) + 4ull * (G->h_pred_off[ne] - G->h_pred_off[nb]) + 4ull * (nl <= PG_MAX_SEQ_TABLE_LABELS ? (1ull << nl) : 0ull);
            if (need <= FRAG_LDS_COUNTERS)
                frag_need = std::max<uint32_t>(frag_need, (uint32_t)need);
        }
        cg[g].node_base = nb;
        cg[g].n_nodes = ne - nb;
        cg[g].n_labels = nl;
        cg[g].pad = 0;
        cg[g].seq_base = G->h_seq_off[g];
        G->h_seq_off[g + 1] = G->h_seq_off[g] + (nl <= PG_MAX_SEQ_TABLE_LABELS ? (1ull << nl) : 0);
    }
    if (G->d_cnt_graphs)  // labels set again: kernels of an earlier batch may still read the old tables
    {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
    }
    (void)pg_dev_free(G->d_count_block);
    G->d_count_block = nullptr;
    G->label_words = words;
    G->frag_lds_counters = (frag_need + 63u) & ~63u;
    PgStagedUpload up;  // seven tables, one copy
    up.add(cg, &G->d_cnt_graphs);
    up.add(G->h_pred_off, &G->d_cnt_pred_off);
    up.add(G->h_pred, &G->d_cnt_pred);
    up.add(G->h_node_len, &G->d_cnt_node_len);
    up.add(lm, &G->d_label_mask);
    up.add(outm, &G->d_out_mask);
    up.add(inm, &G->d_in_mask);
    HIP_TRY(ctx, up.commit(ctx->stream_copy, &G->d_count_block));
    G->labels_set = true;
    return PG_OK;
}

extern "C" pg_status pg_graphs_set_labels(
    pg_ctx* ctx, pg_graphs* G, const uint64_t* label_mask_of_pred, const uint32_t* n_labels)
{
    return pg_graphs_set_labels_wide(ctx, G, label_mask_of_pred, 1, n_labels);
}

extern "C" pg_status pg_graphs_label_words(const pg_graphs* G, uint32_t* words)
{
    if (!G || !words || !G->labels_set)
        return PG_ERR_INVALID;
    *words = G->label_words;
    return PG_OK;
}

extern "C" pg_status pg_graphs_count_layout(const pg_graphs* G, pg_count_layout* out)
{
    if (!G || !out || !G->labels_set)
        return PG_ERR_INVALID;
    layout_of(G, out);
    return PG_OK;
}

extern "C" pg_status pg_graphs_seq_offsets(const pg_graphs* G, uint64_t* seq_off)
{
    if (!G || !seq_off || !G->labels_set)
        return PG_ERR_INVALID;
    std::copy(G->h_seq_off.begin(), G->h_seq_off.end(), seq_off);
    return PG_OK;
}

extern "C" pg_status pg_batch_set_fragments(
    pg_ctx* ctx, pg_batch* b, const uint32_t* fragment_of_read, const uint8_t* is_reverse_strand)
{
    if (!ctx || !b || !b->graphs || (b->n_reads && !fragment_of_read))
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_set_fragments: null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n = b->n_reads;
    // fragments: CSR over (graph, fragment id), reads in input order inside a fragment
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    const std::vector<uint32_t>& gor = b->h_graph_of_read;
    if (std::is_sorted(gor.begin(), gor.end()))
    {
        // reads arrive site after site: order each site's own range (short, cache-resident) by fragment id
        for (uint32_t lo = 0; lo < n;)
        {
            uint32_t hi = lo + 1;
            while (hi < n && gor[hi] == gor[lo])
                ++hi;
            if (!std::is_sorted(fragment_of_read + lo, fragment_of_read + hi))
                std::stable_sort(order.begin() + lo, order.begin() + hi,
                                 [&](uint32_t x, uint32_t y) { return fragment_of_read[x] < fragment_of_read[y]; });
            lo = hi;
        }
    }
    else
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            return gor[x] != gor[y] ? gor[x] < gor[y] : fragment_of_read[x] < fragment_of_read[y];
        });
    std::vector<uint32_t> frag_off;
    frag_off.reserve(n / 2 + 2);
    for (uint32_t i = 0; i < n; ++i)
        if (i == 0 || gor[order[i]] != gor[order[i - 1]] || fragment_of_read[order[i]] != fragment_of_read[order[i - 1]])
            frag_off.push_back(i);
    const uint32_t n_frags = (uint32_t)frag_off.size();
    frag_off.push_back(n);
    b->n_frags = n_frags;
    if (n > b->cap_count_reads || !b->d_support)  // also for a batch without reads: the later calls expect the buffers
    {
        HIP_TRY(ctx, pg_batch_wait(ctx, b));  // an earlier use of this batch may still run
        (void)pg_dev_free(b->d_support);
        (void)pg_dev_free(b->d_frag_reads);
        (void)pg_dev_free(b->d_is_rev);
        (void)pg_dev_free(b->d_path);
        b->cap_count_reads = n;
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_support, std::max<size_t>(n, 1) * sizeof(pg_read_support)));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_frag_reads, std::max<size_t>(n, 1) * sizeof(uint32_t)));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_is_rev, std::max<size_t>(n, 1)));
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_path, std::max<uint64_t>(b->ops_cap, 1) * sizeof(uint32_t)));
    }
    if (!b->d_path_counter)
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_path_counter, sizeof(unsigned long long)));
    if (n_frags + 1 > b->cap_frags)
    {
        HIP_TRY(ctx, pg_batch_wait(ctx, b));
        (void)pg_dev_free(b->d_frag_off);
        b->cap_frags = n_frags + 1;
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_frag_off, b->cap_frags * sizeof(uint32_t)));
    }
    if (n)
    {
        HIP_TRY(ctx, hipMemcpyAsync(b->d_frag_reads, order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream_copy));
        HIP_TRY(ctx, hipMemcpyAsync(b->d_frag_off, frag_off.data(), frag_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream_copy));
        if (is_reverse_strand)
            HIP_TRY(ctx, hipMemcpyAsync(b->d_is_rev, is_reverse_strand, n, hipMemcpyHostToDevice, ctx->stream_copy));
        else
            HIP_TRY(ctx, hipMemsetAsync(b->d_is_rev, 0, n, ctx->stream_copy));
    }
    HIP_TRY(ctx, pg_wait_stream(b, ctx->stream_copy));  // `order` / `frag_off` are host temporaries
    if (b->ev_upload)
    {
        HIP_TRY(ctx, hipEventRecord(b->ev_upload, ctx->stream_copy));
        b->upload_recorded = true;
    }
    b->fragments_set = true;
    return PG_OK;
}

namespace
{
__global__ void pg_count_zero_kernel(uint32_t* counts, uint32_t n, unsigned long long* path_counter)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        counts[i] = 0;
    if (i == 0)
        *path_counter = 0;
}

__global__ void pg_publish_counters_kernel(const unsigned long long* ops_counter, const unsigned long long* path_counter, const uint32_t* index_error,
                                           unsigned long long* host_words)
{
    host_words[0] = *ops_counter;
    host_words[1] = *path_counter;
    host_words[2] = index_error ? *index_error : 0u;  // the error word of a path index built on the device (pg_path.hip)
    __threadfence_system();
}
}  // namespace

extern "C" pg_status pg_batch_count(pg_ctx* ctx, pg_batch* b, const pg_count_params* params, uint32_t* d_counts)
{
    PG_TIMED("pg_batch_count (whole call)");
    if (!ctx || !b || !b->graphs || !params)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_count: null argument");
    const pg_graphs* G = b->graphs;
    if (!G->labels_set)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_count: call pg_graphs_set_labels first");
    if (!b->fragments_set)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_count: call pg_batch_set_fragments first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the count path runs behind the traceback on the second compute stream: the main stream stays free for the next batch's fill.
    // The count pass behind a path stage (the filter chain of the cascade's first hand-over) follows that stage onto the seed
    // stream -- when it counts into the batch's own table: a caller's table is ordered against the second stream
    // (pg_ctx_count_record / pg_ctx_count_wait).
    hipStream_t cs = (b->seed_chain && !d_counts) ? b->seed_stream : ctx->stream2;
    if (cs == ctx->stream2)
        b->seed_chain = false;
    HIP_TRY(ctx, pg_stage_begin_on(ctx, b, cs));
    const uint32_t n = b->n_reads;
    pg_count_layout lay;
    layout_of(G, &lay);
    uint32_t* counts = d_counts;
    b->counts_owned_valid = false;
    if (!counts)
    {
        if (lay.n_counters > b->cap_counts)
        {
            b->park(b->d_counts);  // (no wait: pg_internal.h, parked_blocks)
            b->d_counts = nullptr;
            b->cap_counts = lay.n_counters;
            HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_counts, lay.n_counters * sizeof(uint32_t)));
        }
        counts = b->d_counts;
        b->counts_owned_valid = true;
    }
    {
        // the batch's own table and the path-entry counter zeroed by ONE dispatch (a caller's table is the caller's to zero)
        const uint32_t nz = b->counts_owned_valid ? (uint32_t)lay.n_counters : 0u;
        hipLaunchKernelGGL(pg_count_zero_kernel, dim3(std::max(1u, (nz + 255u) / 256u)), dim3(256), 0, cs, nz ? counts : nullptr, nz, b->d_path_counter);
        HIP_TRY(ctx, hipGetLastError());
    }
    CountArgs a{};
    a.n_reads = n;
    a.prm = *params;
    a.results = b->d_results;
    a.ops = b->d_ops;
    a.base_off = b->d_base_off;
    a.graph_of_read = b->d_graph_of_read;
    a.is_rev = b->d_is_rev;
    a.graphs = G->d_cnt_graphs;
    a.pred_off = G->d_cnt_pred_off;
    a.pred = G->d_cnt_pred;
    a.node_len = G->d_cnt_node_len;
    a.label_mask = G->d_label_mask;
    a.out_mask = G->d_out_mask;
    a.in_mask = G->d_in_mask;
    a.label_words = G->label_words;
    a.frag_lds_counters = G->frag_lds_counters;
    b->label_ext_words = G->label_words - 1;
    b->label_ext_reads = n;
    if (b->label_ext_words)
    {
        const size_t need = std::max<size_t>(n, 1) * b->label_ext_words;
        if (need > b->cap_label_ext)
        {
            b->park(b->d_label_ext);
            b->d_label_ext = nullptr;
            b->cap_label_ext = 0;
            HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_label_ext, need * sizeof(uint64_t)));
            b->cap_label_ext = need;
        }
        HIP_TRY(ctx, hipMemsetAsync(b->d_label_ext, 0, need * sizeof(uint64_t), cs));  // reads that are not MAPPED write nothing
    }
    a.label_ext = b->d_label_ext;
    a.support = b->d_support;
    a.path = b->d_path;
    a.path_counter = b->d_path_counter;
    a.n_frags = b->n_frags;
    a.frag_off = b->d_frag_off;
    a.frag_reads = b->d_frag_reads;
    a.counts = counts;
    a.lay = lay;
    if (params->use_kmer_filter)
    {
        const pg_path_index* fx = G->filter_index;
        if (!fx)
            return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_count: use_kmer_filter needs pg_graphs_build_filter_index");
        a.bases = b->d_bases;
        a.kf_graphs = fx->d_graphs;
        a.kf_table = fx->d_table;
        a.kf_pool = fx->d_pool;
        a.kf_node_off = fx->d_node_off;
        a.kf_raw = fx->d_raw;
        a.kf_node_uniq = fx->d_node_uniq;
    }
    if (n)
    {
        PG_TIMED("launch pg_support_kernel + pg_fragment_kernel");
        hipLaunchKernelGGL(pg_support_kernel, dim3((n + 63) / 64), dim3(64), 0, cs, a);
        HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(pg_fragment_kernel, dim3((b->n_frags + FRAG_BLOCK - 1) / FRAG_BLOCK), dim3(FRAG_BLOCK),
                           (size_t)a.frag_lds_counters * sizeof(uint32_t), cs, a);
        HIP_TRY(ctx, hipGetLastError());
    }
    // the two sizes a caller needs before it can fetch the records, sent to page-locked host memory by this stream: they are there
    // when the batch's event is (pg_batch_result_sizes)
    if (!b->h_counters)
    {
        void* p = nullptr;
        HIP_TRY(ctx, hipHostMalloc(&p, 3 * sizeof(unsigned long long), hipHostMallocPortable | hipHostMallocMapped));
        b->h_counters = (unsigned long long*)p;
        void* dp = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&dp, p, 0));
        b->d_h_counters = (unsigned long long*)dp;
    }
    // (one single-thread dispatch writes both words into the page-locked block: two 8-byte copies were two dispatches)
    hipLaunchKernelGGL(pg_publish_counters_kernel, dim3(1), dim3(1), 0, cs, b->d_ops_counter, b->d_path_counter,
                       b->graphs->path_index && !b->graphs->path_index->build_pending ? b->graphs->path_index->d_error : nullptr, b->d_h_counters);
    HIP_TRY(ctx, hipGetLastError());
    b->h_counters_valid = true;
    HIP_TRY(ctx, pg_stage_end_on(ctx, b, cs));
    return PG_OK;
}

extern "C" pg_status pg_batch_result_sizes(pg_ctx* ctx, pg_batch* b, uint64_t* n_ops, uint64_t* n_path)
{
    if (!ctx || !b || !b->graphs || !b->d_support || !b->h_counters_valid)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_result_sizes: the batch's last stage must be pg_batch_count");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, pg_batch_wait(ctx, b));
    if (b->h_counters[2])
        return pg_fail(ctx, b->h_counters[2] & 1u ? PG_ERR_UNSUPPORTED : PG_ERR_HIP, pg_path_index_error_text((uint32_t)b->h_counters[2]));
    if (n_ops)
        *n_ops = b->h_counters[0];
    if (n_path)
        *n_path = b->h_counters[1];
    return PG_OK;
}

extern "C" pg_status pg_batch_download_all(
    pg_ctx* ctx, pg_batch* b, pg_result* results, pg_op* ops, uint64_t ops_cap, uint32_t* counts, pg_read_support* supports,
    uint32_t* path, uint64_t path_cap)
{
    if (!ctx || !b || !b->graphs || !b->d_support || !b->h_counters_valid)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_download_all: the batch's last stage must be pg_batch_count");
    if (b->n_reads && (!results || !supports))
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_download_all: null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, pg_batch_wait(ctx, b));
    if (b->h_counters[2])
        return pg_fail(ctx, b->h_counters[2] & 1u ? PG_ERR_UNSUPPORTED : PG_ERR_HIP, pg_path_index_error_text((uint32_t)b->h_counters[2]));
    const uint64_t n_ops = b->h_counters[0], n_path = b->h_counters[1];
    if ((n_ops && (!ops || n_ops > ops_cap)) || (n_path && (!path || n_path > path_cap)))
        return pg_fail(ctx, PG_ERR_OVERFLOW, "pg_batch_download_all: ops / path buffer too small (pg_batch_result_sizes gives the sizes)");
    hipStream_t cs = ctx->stream_copy;
    if (b->n_reads)
    {
        HIP_TRY(ctx, hipMemcpyAsync(results, b->d_results, b->n_reads * sizeof(pg_result), hipMemcpyDeviceToHost, cs));
        HIP_TRY(ctx, hipMemcpyAsync(supports, b->d_support, b->n_reads * sizeof(pg_read_support), hipMemcpyDeviceToHost, cs));
    }
    if (n_ops)
        HIP_TRY(ctx, hipMemcpyAsync(ops, b->d_ops, n_ops * sizeof(pg_op), hipMemcpyDeviceToHost, cs));
    if (n_path)
        HIP_TRY(ctx, hipMemcpyAsync(path, b->d_path, n_path * sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
    if (counts)
    {
        if (!b->counts_owned_valid)
        {
            HIP_TRY(ctx, hipStreamSynchronize(cs));
            return pg_fail(ctx, PG_ERR_INVALID, "the count table lives in caller memory (d_counts was given)");
        }
        pg_count_layout lay;
        layout_of(b->graphs, &lay);
        HIP_TRY(ctx, hipMemcpyAsync(counts, b->d_counts, lay.n_counters * sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
    }
    HIP_TRY(ctx, pg_wait_stream(b, cs));
    return PG_OK;
}

extern "C" pg_status pg_batch_download_label_ext(pg_ctx* ctx, pg_batch* b, uint64_t* label_ext, uint64_t cap_words)
{
    if (!ctx || !b || !b->graphs || !b->d_support)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_download_label_ext: pg_batch_count has not run");
    // the sets on the device are those of the last pg_batch_count: a batch uploaded again since (other reads, another graph set)
    // has none until it is counted again
    if (b->label_ext_reads != b->n_reads || b->label_ext_words != b->graphs->label_words - 1)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_download_label_ext: the batch was uploaded again after its last pg_batch_count");
    const uint64_t need = (uint64_t)b->n_reads * b->label_ext_words;
    if (need == 0)
        return PG_OK;
    if (need > b->cap_label_ext)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_download_label_ext: no label sets of this size on the device");
    if (!label_ext || cap_words < need)
        return pg_fail(ctx, PG_ERR_OVERFLOW, "label_ext buffer too small (n_reads x (words - 1))");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, pg_batch_wait(ctx, b));
    HIP_TRY(ctx, hipMemcpyAsync(label_ext, b->d_label_ext, need * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream_copy));
    HIP_TRY(ctx, pg_stream_wait(ctx->device, ctx->stream_copy));
    return PG_OK;
}

extern "C" pg_status pg_batch_download_counts(
    pg_ctx* ctx, pg_batch* b, uint32_t* counts, pg_read_support* supports, uint32_t* path, uint64_t path_cap,
    uint64_t* n_path)
{
    if (!ctx || !b || !b->graphs || !b->d_support)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_download_counts: pg_batch_count has not run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, pg_batch_wait(ctx, b));
    unsigned long long np = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&np, b->d_path_counter, sizeof np, hipMemcpyDeviceToHost, ctx->stream_copy));
    HIP_TRY(ctx, pg_stream_wait(ctx->device, ctx->stream_copy));
    if (n_path)
        *n_path = np;
    if (counts)
    {
        if (!b->counts_owned_valid)
            return pg_fail(ctx, PG_ERR_INVALID, "the count table lives in caller memory (d_counts was given)");
        pg_count_layout lay;
        layout_of(b->graphs, &lay);
        HIP_TRY(ctx, hipMemcpyAsync(counts, b->d_counts, lay.n_counters * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream_copy));
    }
    if (supports && b->n_reads)
        HIP_TRY(ctx, hipMemcpyAsync(supports, b->d_support, b->n_reads * sizeof(pg_read_support), hipMemcpyDeviceToHost, ctx->stream_copy));
    if (path && np)
    {
        if (np > path_cap)
        {
            HIP_TRY(ctx, pg_stream_wait(ctx->device, ctx->stream_copy));
            return pg_fail(ctx, PG_ERR_OVERFLOW, "path buffer too small");
        }
        HIP_TRY(ctx, hipMemcpyAsync(path, b->d_path, np * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream_copy));
    }
    HIP_TRY(ctx, pg_stream_wait(ctx->device, ctx->stream_copy));
    return PG_OK;
}
