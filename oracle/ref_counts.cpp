/*
 * oracle/ref_counts.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Count-path checker built on the REFERENCE's own graph-tools library (unpacked from
 * /root/reference/external/graph-tools.tar.gz into a temporary directory by oracle/Makefile and
 * compiled as-is): graphtools::Graph, Path (validity rules), PathFamily::containsPath,
 * decodeGraphAlignment, Alignment/Operation counters are the reference's code.  What is restated here
 * is only the glue that cannot be compiled in this image (it lives in translation units that pull
 * Boost/jsoncpp/spdlog):
 *
 *   read filters NonUniq / BadAlign + chain     src/c++/lib/paragraph/readfilters/NonUniq.hh:48-52,
 *                                               BadAlign.hh:62-73, lib/paragraph/ReadFilter.cpp:43-90
 *   CompositeAligner filter step (gssw stage)   src/c++/lib/grm/CompositeAligner.cpp:152-175
 *   nodefilter / edgefilter lambdas             src/c++/lib/paragraph/Disambiguation.cpp:212-296
 *   disambiguateReads                           src/c++/lib/paragraph/Disambiguation.cpp:82-142
 *   Fragment::addRead / readsToFragments        src/c++/lib/common/Fragment.cpp:34-69, 141-181
 *   countNodes / countEdges / countPathFamilies src/c++/lib/paragraph/ReadCounting.cpp:52-127
 *
 * Interface limits (checker only): <= 64 nodes, <= 64 edges, <= 256 labels per graph (label sets: 4 words of 64 bits).
 */
#include <limits>
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "graphcore/GraphCoordinates.hh"
#include "graphalign/GraphAlignment.hh"
#include "graphalign/GraphAlignmentOperations.hh"
#include "graphcore/Graph.hh"
#include "graphcore/Path.hh"
#include "graphcore/PathFamily.hh"

using graphtools::Graph;
using graphtools::GraphAlignment;
using graphtools::NodeId;
using graphtools::decodeGraphAlignment;

struct pgrefc_graph
{
    Graph graph;
    std::vector<std::pair<NodeId, NodeId>> edges;
    std::map<std::pair<NodeId, NodeId>, uint32_t> edge_index;
    std::vector<std::string> labels;
    std::map<std::string, uint32_t> label_index;
};

struct pgrefc_params
{
    int32_t remove_nonuniq;  // paragraph: --bad-align-nonuniq (default true); grmpy: false
    double bad_align_frac;   // 0.8
    int32_t use_support_filters;  // 1: production nodefilter/edgefilter lambdas, 0: none (unit-test mode)
};

extern "C" pgrefc_graph* pgrefc_graph_create(
    uint32_t n_nodes, const uint32_t* seq_off, const char* seq, uint32_t n_edges, const uint32_t* from,
    const uint32_t* to, const uint32_t* label_off, const uint32_t* label_ids, uint32_t n_labels,
    const char* const* label_names)
{
    if (n_nodes > 64 || n_edges > 64 || n_labels > 256)
        return nullptr;
    auto* g = new pgrefc_graph{ Graph(n_nodes, false), {}, {}, {}, {} };  // expansion off as graphFromJson (GraphInput.cpp:62)
    for (uint32_t i = 0; i < n_nodes; ++i)
    {
        g->graph.setNodeName(i, "n" + std::to_string(i));
        g->graph.setNodeSeq(i, std::string(seq + seq_off[i], seq + seq_off[i + 1]));
    }
    for (uint32_t l = 0; l < n_labels; ++l)
    {
        g->labels.emplace_back(label_names[l]);
        g->label_index[label_names[l]] = l;
    }
    try
    {
        for (uint32_t e = 0; e < n_edges; ++e)
        {
            g->graph.addEdge(from[e], to[e]);
            g->edges.emplace_back(from[e], to[e]);
            g->edge_index[{ from[e], to[e] }] = e;
            for (uint32_t k = label_off[e]; k < label_off[e + 1]; ++k)
                g->graph.addLabelToEdge(from[e], to[e], g->labels[label_ids[k]]);
        }
    }
    catch (std::exception const&)
    {
        delete g;
        return nullptr;
    }
    return g;
}

extern "C" void pgrefc_graph_destroy(pgrefc_graph* g) { delete g; }

// Graph length of a fragment of two graph-mapped reads exactly as common::Fragment::addRead computes it
// (src/c++/lib/common/Fragment.cpp:70-95) on the reference's own graphtools::GraphCoordinates (compiled from the tarball):
// canonical (start, end) of both mappings into span[0..3], the fragment length (all ones = no path) into *length.
extern "C" int pgrefc_pair_length(
    pgrefc_graph* g, int32_t pos1, const char* cigar1, int32_t pos2, const char* cigar2, uint64_t* span, uint64_t* length)
{
    try
    {
        graphtools::GraphCoordinates coordinates(&g->graph);
        const GraphAlignment m1 = decodeGraphAlignment(pos1, cigar1, &g->graph), m2 = decodeGraphAlignment(pos2, cigar2, &g->graph);
        const std::pair<uint64_t, uint64_t> p1 = coordinates.canonicalStartAndEnd(m1.path()), p2 = coordinates.canonicalStartAndEnd(m2.path());
        span[0] = p1.first, span[1] = p1.second, span[2] = p2.first, span[3] = p2.second;
        const uint64_t d1 = coordinates.distance(p1.second, p2.first), d2 = coordinates.distance(p2.second, p1.first);
        const uint64_t distance = std::min(d1, d2);
        *length = distance == std::numeric_limits<uint64_t>::max() ? (uint64_t)-1 : m1.queryLength() + m2.queryLength() + distance;
        return 0;
    }
    catch (std::exception const&)
    {
        return 1;
    }
}

namespace
{
struct ReadRec
{
    int32_t pos;
    std::string cigar;
    size_t read_len;
    bool unique, graph_reverse;
    uint32_t fragment;
    int status;  // 0 not aligned, 1 MAPPED, 2 BAD_ALIGN
    std::set<NodeId> nodes;
    std::set<std::pair<NodeId, NodeId>> edges;
    std::set<std::string> labels;
};

// Disambiguation.cpp:212-242
bool node_filter(const Graph& graph, const ReadRec& read, NodeId node_id)
{
    try
    {
        GraphAlignment alignment = decodeGraphAlignment(read.pos, read.cigar, &graph);
        const bool is_short_node = graph.nodeSeq(node_id).size() < read.read_len / 2;
        int32_t index = 0;
        for (const auto& node_alignment : alignment)
        {
            if (node_id == alignment.getNodeIdByIndex(index))
            {
                const size_t nonmatch = node_alignment.numMismatched() + node_alignment.numClipped();
                const size_t indel = node_alignment.numInserted() + node_alignment.numDeleted();
                if (is_short_node && (nonmatch > 0 || indel > 0))
                    return false;
                return nonmatch + indel <= read.read_len / 2;
            }
            ++index;
        }
    }
    catch (std::exception const&)
    {
    }
    return false;
}

// Disambiguation.cpp:244-296
bool edge_filter(const Graph& graph, const ReadRec& read, NodeId node_id1, NodeId node_id2)
{
    try
    {
        GraphAlignment alignment = decodeGraphAlignment(read.pos, read.cigar, &graph);
        const graphtools::Alignment* previous_alignment = nullptr;
        auto previous_node_id = static_cast<NodeId>(-1);
        int32_t index = 0;
        for (const auto& node_alignment : alignment)
        {
            const auto node_alignment_node_id = alignment.getNodeIdByIndex(index);
            if (previous_alignment != nullptr && previous_node_id == node_id1 && node_alignment_node_id == node_id2)
            {
                auto const min_node_overlap = static_cast<int32_t>(read.read_len / 10 + 1);
                bool status
                    = (previous_alignment->numMatched()
                           >= (unsigned)std::min(previous_alignment->referenceLength(), (uint32_t)min_node_overlap)
                       && node_alignment.numMatched()
                           >= (unsigned)std::min(node_alignment.referenceLength(), (uint32_t)min_node_overlap));
                if (status)
                    status = (previous_alignment->queryLength() < previous_alignment->referenceLength() * 2)
                        && (node_alignment.queryLength() < node_alignment.referenceLength() * 2);
                if (status)
                {
                    const auto node1_length = static_cast<int32_t>(graph.nodeSeq(node_id1).size());
                    const auto node2_length = static_cast<int32_t>(graph.nodeSeq(node_id2).size());
                    status = ((int32_t)previous_alignment->numMatched() >= std::min(node1_length, min_node_overlap))
                        && ((int32_t)node_alignment.numMatched() >= std::min(node2_length, min_node_overlap));
                }
                return status;
            }
            previous_alignment = &node_alignment;
            previous_node_id = node_alignment_node_id;
            ++index;
        }
    }
    catch (std::exception const&)
    {
    }
    return false;
}

struct Frag
{
    uint64_t n_reads = 0, n_fwd = 0, n_rev = 0;
    std::set<NodeId> nodes;
    std::set<std::pair<NodeId, NodeId>> edges;
    std::set<std::string> labels;
};

void add_count(uint64_t* c, const Frag& f)
{  // ReadCounting.cpp:52-69
    c[0] += 1;
    c[1] += f.n_reads;
    c[2] += f.n_fwd;
    c[3] += f.n_rev;
}
}  // namespace

/*
 * Runs filter -> disambiguate -> fragments -> counts for the reads of ONE site.
 * per-read outputs: status (0 skipped, 1 MAPPED, 2 BAD_ALIGN), node/edge/label support masks
 * site outputs: node_counts[n_nodes*4], edge_counts[n_edges*4] = {count, READS, FWD, REV};
 *               seq table: n_seq entries {label mask, 4 counters}
 * returns 0, or -1 if decodeGraphAlignment threw where the reference would propagate the exception.
 */
extern "C" int pgrefc_count_site(
    pgrefc_graph* g, uint32_t n_reads, const int32_t* pos, const uint32_t* cigar_off, const char* cigars,
    const uint8_t* aligned, const uint8_t* unique, const uint8_t* graph_reverse, const uint32_t* read_len,
    const uint32_t* fragment, const pgrefc_params* prm, uint8_t* out_status, uint64_t* out_nodes, uint64_t* out_edges,
    uint64_t* out_labels, uint64_t* node_counts, uint64_t* edge_counts, uint32_t* n_seq, uint64_t* seq_masks,
    uint64_t* seq_counts, uint32_t seq_cap)
{
    Graph& graph = g->graph;
    std::vector<ReadRec> reads(n_reads);
    int rc = 0;
    for (uint32_t i = 0; i < n_reads; ++i)
    {
        ReadRec& r = reads[i];
        r.pos = pos[i];
        r.cigar.assign(cigars + cigar_off[i], cigars + cigar_off[i + 1]);
        r.read_len = read_len[i];
        r.unique = unique[i] != 0;
        r.graph_reverse = graph_reverse[i] != 0;
        r.fragment = fragment[i];
        r.status = 0;
        if (!aligned[i])
            continue;
        r.status = 1;  // CompositeAligner.cpp:156
        // filter chain, ReadFilter.cpp:43-90
        bool bad = false;
        if (prm->remove_nonuniq && !r.unique)
            bad = true;
        if (!bad)
        {
            try
            {
                const GraphAlignment mapping = decodeGraphAlignment(r.pos, r.cigar, &graph);
                size_t query_clipped = 0;
                for (auto const& aln : mapping)
                    query_clipped += aln.numClipped();
                const auto query_aligned = mapping.queryLength() - query_clipped;
                bad = query_aligned < round(prm->bad_align_frac * (mapping.queryLength()));
            }
            catch (std::exception const&)
            {
                rc = -1;
                bad = true;
            }
        }
        if (bad)
            r.status = 2;
    }
    // disambiguateReads (only MAPPED reads survive alignReads, Align.cpp:81-84,155)
    for (auto& r : reads)
    {
        if (r.status != 1)
            continue;
        bool has_previous = false;
        NodeId pnode = 0;
        std::set<std::string> overlapped_pfams;
        try
        {
            GraphAlignment gm = decodeGraphAlignment(r.pos, r.cigar, &graph);
            auto const& path = gm.path();
            for (auto node = path.begin(); node != path.end(); ++node)
            {
                if (has_previous && (!prm->use_support_filters || edge_filter(graph, r, pnode, *node)))
                {
                    r.edges.emplace(pnode, *node);
                    for (const auto& s : graph.edgeLabels(pnode, *node))
                        overlapped_pfams.insert(s);
                }
                has_previous = true;
                pnode = *node;
                if (!prm->use_support_filters || node_filter(graph, r, *node))
                    r.nodes.emplace(*node);
            }
            for (auto const& label : overlapped_pfams)
            {
                graphtools::PathFamily pfam(&graph, label);
                if (pfam.containsPath(path))
                    r.labels.insert(label);
            }
        }
        catch (std::exception const&)
        {
            rc = -1;
        }
    }
    for (uint32_t i = 0; i < n_reads; ++i)
    {
        const ReadRec& r = reads[i];
        out_status[i] = (uint8_t)r.status;
        uint64_t nm = 0, em = 0;
        for (auto n : r.nodes)
            nm |= 1ull << n;
        for (auto const& e : r.edges)
            em |= 1ull << g->edge_index.at(e);
        uint64_t* lm = out_labels + 4 * (size_t)i;  // four words per read
        lm[0] = lm[1] = lm[2] = lm[3] = 0;
        for (auto const& l : r.labels)
        {
            const uint32_t b = g->label_index.at(l);
            lm[b >> 6] |= 1ull << (b & 63);
        }
        out_nodes[i] = nm;
        out_edges[i] = em;
    }
    // readsToFragments over the surviving reads, in read order (Fragment.cpp:165-181)
    std::list<Frag> frags;
    std::unordered_map<uint32_t, Frag*> fmap;
    for (auto const& r : reads)
    {
        if (r.status != 1)
            continue;
        auto it = fmap.find(r.fragment);
        if (it == fmap.end())
        {
            frags.emplace_back();
            it = fmap.emplace(r.fragment, &frags.back()).first;
        }
        Frag& f = *it->second;
        ++f.n_reads;
        if (r.graph_reverse)
            ++f.n_rev;
        else
            ++f.n_fwd;
        f.nodes.insert(r.nodes.begin(), r.nodes.end());
        f.edges.insert(r.edges.begin(), r.edges.end());
        f.labels.insert(r.labels.begin(), r.labels.end());
    }
    std::fill(node_counts, node_counts + 4 * graph.numNodes(), 0);
    std::fill(edge_counts, edge_counts + 4 * g->edges.size(), 0);
    std::map<std::array<uint64_t, 4>, std::array<uint64_t, 4>> seqs;  // label set (four words) -> counters
    for (auto const& f : frags)
    {
        for (auto n : f.nodes)
            add_count(node_counts + 4 * n, f);
        for (auto const& e : f.edges)
            add_count(edge_counts + 4 * g->edge_index.at(e), f);
        if (!f.labels.empty())
        {
            std::array<uint64_t, 4> lm{ { 0, 0, 0, 0 } };
            for (auto const& l : f.labels)
            {
                const uint32_t b = g->label_index.at(l);
                lm[b >> 6] |= 1ull << (b & 63);
            }
            auto& c = seqs[lm];
            add_count(c.data(), f);
        }
    }
    *n_seq = 0;
    for (auto const& kv : seqs)
    {
        if (*n_seq >= seq_cap)
            break;
        std::memcpy(seq_masks + 4 * (*n_seq), kv.first.data(), 4 * sizeof(uint64_t));
        std::memcpy(seq_counts + 4 * (*n_seq), kv.second.data(), 4 * sizeof(uint64_t));
        ++*n_seq;
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------------------
 * PathAligner checker (src/c++/lib/grm/PathAligner.cpp:70-164).  The reference's own
 * graphtools::extendPath / extendPathMatching / projectAlignmentOntoGraph / Alignment / generateCigar run
 * as compiled from the tarball; restated are (a) the KmerIndex constructor (GT!/src/graphalign/KmerIndex.cpp:
 * 76-116 -- that file includes a Boost header and cannot be compiled here): every length-k path from every
 * node position via extendPath, keyed by its sequence, and (b) PathAligner::alignRead's control flow.
 * ---------------------------------------------------------------------------------------------------- */
#include "graphcore/PathOperations.hh"

namespace
{
std::string rc_graphtools(std::string s)
{  // GT!/src/graphutils/SequenceOperations.cpp:66-88
    for (char& c : s)
        c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
    std::reverse(s.begin(), s.end());
    return s;
}
}  // namespace

struct pgrefc_path_result
{
    int32_t status;  /* 0 unmapped, 1 MAPPED */
    int32_t graph_pos, score, mapq, unique, is_graph_reverse, anchored;
    int32_t cigar_len;
};

extern "C" int pgrefc_path_align(
    pgrefc_graph* g, int32_t kmer_len, uint32_t n_reads, const uint32_t* base_off, const char* bases,
    pgrefc_path_result* out, char* cigars, int cigar_stride)
{
    using graphtools::Path;
    const Graph& graph = g->graph;
    std::unordered_map<std::string, std::list<Path>> index;
    for (NodeId node_id = 0; node_id != graph.numNodes(); ++node_id)
    {
        const std::string node_seq = graph.nodeSeq(node_id);
        std::vector<NodeId> node_list{ node_id };
        for (size_t pos = 0; pos != node_seq.length(); ++pos)
        {
            Path path(&graph, static_cast<int32_t>(pos), node_list, static_cast<int32_t>(pos));
            for (const Path& kp : graphtools::extendPath(path, 0, kmer_len - 1))
                index[kp.seq()].push_back(kp);
        }
    }
    struct ExactMatch
    {
        size_t qpos;
        Path path;
        bool isReverse;
    };
    for (uint32_t r = 0; r < n_reads; ++r)
    {
        pgrefc_path_result& o = out[r];
        std::memset(&o, 0, sizeof o);
        if (cigars)
            cigars[(size_t)r * cigar_stride] = 0;
        const std::string read_fwd(bases + base_off[r], bases + base_off[r + 1]);
        const size_t read_length = read_fwd.size();
        if (read_length < (size_t)kmer_len)
            continue;
        std::list<ExactMatch> matches;
        for (int strand = 0; strand < 2; ++strand)
        {
            const bool is_reverse_strand = strand != 0;
            const std::string read_bases = is_reverse_strand ? rc_graphtools(read_fwd) : read_fwd;
            for (size_t pos = 0; pos + kmer_len <= read_bases.size(); ++pos)
            {
                const std::string kmer = read_bases.substr(pos, kmer_len);
                auto it = index.find(kmer);
                if (it != index.end() && it->second.size() == 1)
                {
                    size_t qpos = pos;
                    const auto extended = graphtools::extendPathMatching(it->second.front(), read_bases, qpos);
                    matches.push_back(ExactMatch{ qpos, extended, is_reverse_strand });
                    pos = matches.back().qpos + matches.back().path.length();
                }
            }
        }
        o.anchored = matches.empty() ? 0 : 1;
        const ExactMatch* first = nullptr;
        int n_full = 0;
        for (auto const& m : matches)
            if (m.path.length() == read_length)
            {
                if (!first)
                    first = &m;
                ++n_full;
            }
        if (!first)
            continue;
        std::string cigar;
        if (first->qpos > 0)
            cigar += std::to_string(first->qpos) + "S";
        cigar += std::to_string(first->path.length()) + "M";
        if (first->qpos + first->path.length() < read_length)
            cigar += std::to_string(read_length - first->qpos - first->path.length()) + "S";
        graphtools::Alignment linearAlignment(0, cigar);
        graphtools::GraphAlignment graphAlignment = graphtools::projectAlignmentOntoGraph(linearAlignment, first->path);
        const std::string gc = graphAlignment.generateCigar();
        o.status = 1;
        o.graph_pos = first->path.startPosition();
        o.score = (int32_t)first->path.length();
        o.is_graph_reverse = first->isReverse ? 1 : 0;
        o.unique = n_full == 1 ? 1 : 0;
        o.mapq = n_full == 1 ? 60 : 0;
        o.cigar_len = (int32_t)gc.size();
        if (cigars && (int)gc.size() < cigar_stride)
            std::memcpy(cigars + (size_t)r * cigar_stride, gc.c_str(), gc.size() + 1);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------
 * KmerFilter checker (src/c++/lib/paragraph/readfilters/KmerFilter.cpp:52-139).  The reference's own
 * graphtools::decodeGraphAlignment / Alignment::numClipped / extendPath run as compiled from the tarball; restated
 * are the KmerIndex (constructor + updateKmerCounts, GT!/src/graphalign/KmerIndex.cpp:76-141 -- Boost include),
 * findMinCoveringKmerLength (GT!/src/graphalign/KmerIndexOperations.cpp:77-113) and filterRead's control flow.
 * Returns the k-mer length used (-1: auto-detection found none).
 * ---------------------------------------------------------------------------------------------------- */
#include <unordered_set>

namespace
{
struct RestatedKmerIndex
{
    std::unordered_map<std::string, std::list<graphtools::Path>> map;
    std::unordered_map<NodeId, size_t> node_counts;
    std::map<std::pair<NodeId, NodeId>, size_t> edge_counts;
    RestatedKmerIndex(const Graph& graph, int32_t k)
    {
        for (NodeId node_id = 0; node_id != graph.numNodes(); ++node_id)
        {
            const std::string node_seq = graph.nodeSeq(node_id);
            std::vector<NodeId> node_list{ node_id };
            for (size_t pos = 0; pos != node_seq.length(); ++pos)
            {
                graphtools::Path path(&graph, static_cast<int32_t>(pos), node_list, static_cast<int32_t>(pos));
                for (const graphtools::Path& kp : graphtools::extendPath(path, 0, k - 1))
                    map[kp.seq()].push_back(kp);
            }
        }
        for (auto const& kv : map)
        {
            if (kv.second.size() != 1)
                continue;
            bool has_previous = false;
            NodeId previous = 0;
            for (auto const& n : kv.second.front().nodeIds())
            {
                node_counts[n] += 1;
                if (has_previous)
                    edge_counts[std::make_pair(previous, n)] += 1;
                has_previous = true;
                previous = n;
            }
        }
    }
    size_t nodeCount(NodeId n) const
    {
        auto it = node_counts.find(n);
        return it == node_counts.end() ? 0 : it->second;
    }
    size_t edgeCount(NodeId a, NodeId b) const
    {
        auto it = edge_counts.find(std::make_pair(a, b));
        return it == edge_counts.end() ? 0 : it->second;
    }
};
}  // namespace

extern "C" int pgrefc_kmer_filter(
    pgrefc_graph* g, int32_t kmer_len, uint32_t n_reads, const int32_t* pos, const uint32_t* cigar_off, const char* cigars,
    const uint32_t* base_off, const char* bases, uint8_t* out_filtered, char* msgs, int msg_stride)
{
    const Graph& graph = g->graph;
    if (kmer_len < 0)
    {
        const size_t need = (size_t)(-kmer_len);
        kmer_len = -1;
        for (int32_t k = 10; k < 64 && kmer_len < 0; ++k)
        {
            RestatedKmerIndex index(graph, k);
            bool any_below = false;
            for (NodeId node_id = 0; node_id != graph.numNodes() && !any_below; ++node_id)
            {
                if (index.nodeCount(node_id) < need)
                    any_below = true;
                for (const auto succ : graph.successors(node_id))
                    if (!any_below && index.edgeCount(node_id, succ) < need)
                        any_below = true;
            }
            if (!any_below)
                kmer_len = k;
        }
        if (kmer_len < 0)
            return -1;
    }
    RestatedKmerIndex index(graph, kmer_len);
    for (uint32_t r = 0; r < n_reads; ++r)
    {
        const std::string cigar(cigars + cigar_off[r], cigars + cigar_off[r + 1]);
        const std::string rb(bases + base_off[r], bases + base_off[r + 1]);
        std::pair<bool, std::string> result{ false, "" };
        try
        {
            const graphtools::GraphAlignment alignment = graphtools::decodeGraphAlignment(pos[r], cigar, &graph);
            const auto sc_left = alignment[0].numClipped();
            const auto sc_right = alignment[alignment.size() - 1].numClipped();
            if ((signed)(rb.size() - sc_left - sc_right) < kmer_len)
                result = { true, "kmer_tooshort" };
            else
            {
                std::unordered_set<std::string> kmers;
                for (size_t p = sc_left; p <= rb.size() - sc_right - kmer_len; ++p)
                    kmers.insert(rb.substr(p, (size_t)kmer_len));
                std::unordered_set<NodeId> nodes_not_covered;
                std::list<NodeId> nodes_supported;
                for (int32_t ni = 0; ni != (int32_t)alignment.size(); ++ni)
                {
                    const NodeId node_id = (NodeId)alignment.getNodeIdByIndex(ni);
                    if (index.nodeCount(node_id) > 0)
                    {
                        nodes_not_covered.insert(node_id);
                        nodes_supported.push_back(node_id);
                    }
                }
                bool pass = false;
                for (auto const& kmer : kmers)
                {
                    auto it = index.map.find(kmer);
                    if (it != index.map.end() && it->second.size() == 1)
                    {
                        for (const auto& node_id : it->second.front().nodeIds())
                        {
                            nodes_not_covered.erase(node_id);
                            if (nodes_not_covered.empty())
                            {
                                pass = true;
                                break;
                            }
                        }
                    }
                    if (pass)
                        break;
                }
                if (!pass)
                {
                    std::string msg = "kmer_uncov";
                    for (auto const& node : nodes_supported)
                        if (nodes_not_covered.count(node) != 0)
                            msg += "_" + std::to_string(node);
                    result = { true, msg };
                }
            }
        }
        catch (std::exception const&)
        {
            result = { true, "kmer_nomapping" };
        }
        out_filtered[r] = result.first ? 1 : 0;
        if (msgs)
        {
            std::strncpy(msgs + (size_t)r * msg_stride, result.second.c_str(), (size_t)msg_stride - 1);
            msgs[(size_t)r * msg_stride + msg_stride - 1] = 0;
        }
    }
    return kmer_len;
}
