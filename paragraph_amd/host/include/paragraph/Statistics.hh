// Post-hoc numbers of the count document: "fragment_statistics" (alignmentStats, lib/paragraph/ReadCounting.cpp:129-223, over
// common::Fragment lengths, lib/common/Fragment.cpp:33-152) and "alignment_statistics" (summarizeAlignments,
// lib/paragraph/GraphSummaryStatistics.cpp:47-185 with AlignmentStatistics.cpp:41-144).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "common/Json.hh"
#include "common/Read.hh"
#include "graphcore/Graph.hh"

namespace paragraph
{
// per-node pieces of a "<id>[<ops>]..." graph CIGAR
struct NodeAlignment
{
    graphtools::NodeId node = 0;
    uint32_t matched = 0, mismatched = 0, clipped = 0, inserted = 0, deleted = 0, missing = 0;
    uint32_t referenceLength() const { return matched + mismatched + deleted + missing; }
    uint32_t queryLength() const { return matched + mismatched + inserted + clipped + missing; }
};
// throws std::runtime_error on malformed text or node ids outside the graph
std::vector<NodeAlignment> decodeGraphCigar(std::string const& graph_cigar, graphtools::Graph const& graph);

// The running estimators the fragment statistics are defined by (boost::accumulators 1.6x: tag::mean = sum / n,
// tag::variance = the iterative update, tag::median = the P-square estimator of Jain & Chlamtac 1985, markers seeded
// with the first five samples; fewer than five samples report the third slot of the unsorted seed array).
class RunningStats
{
public:
    void add(double x);
    double mean() const;      // NaN when empty (written as null)
    double variance() const;  // 0 when fewer than two samples
    double median() const;    // 0 when empty
    size_t count() const { return n_; }

private:
    size_t n_ = 0;
    double sum_ = 0, imm_mean_ = 0, var_ = 0;
    double heights_[5] = { 0, 0, 0, 0, 0 }, actual_[5] = { 1, 2, 3, 4, 5 }, desired_[5] = { 1, 2, 3, 4, 5 };
};

// A set of sequence labels of one graph (bit i = SiteReadViews::label_names[i]).  The reference keeps label NAMES in std::sets
// and has no bound on their number (src/c++/lib/paragraph/ReadCounting.cpp:96-127); the device keeps bit sets of up to
// PG_LABEL_WORDS = 4 words (include/paragraph_amd.h), i.e. up to 256 labels on a graph.
struct LabelSet
{
    static constexpr unsigned WORDS = 4;
    uint64_t w[WORDS] = { 0, 0, 0, 0 };
    LabelSet() = default;
    LabelSet(uint64_t w0) { w[0] = w0; }  // (implicit: the ABI's first word)
    void set(size_t b) { w[b >> 6] |= 1ull << (b & 63); }
    bool test(size_t b) const { return b < 64 * WORDS && ((w[b >> 6] >> (b & 63)) & 1) != 0; }
    bool any() const { return (w[0] | w[1] | w[2] | w[3]) != 0; }
    LabelSet& operator|=(LabelSet const& o)
    {
        for (unsigned i = 0; i < WORDS; ++i)
            w[i] |= o.w[i];
        return *this;
    }
    bool operator==(LabelSet const& o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
    bool operator<(LabelSet const& o) const
    {
        for (unsigned i = WORDS; i-- > 0;)
            if (w[i] != o.w[i])
                return w[i] < o.w[i];
        return false;
    }
};

// What the statistics need of one read that survived alignment + filters, independent of where it is kept (a common::Read
// with its graph CIGAR string, or the flat device results of a packed batch)
struct MappedReadView
{
    uint32_t fragment = 0;  // dense id of the read's fragment within the site, numbered by first appearance
    uint32_t read_length = 0;
    int32_t chrom_id = -1, pos = -1, mate_chrom_id = -1, mate_pos = -1;
    int32_t graph_pos = 0, graph_alignment_score = 0;
    bool is_mapped = false, is_mate_mapped = false, is_reverse_strand = false, is_mate_reverse_strand = false;
    bool is_graph_mapped = false, is_graph_reverse_strand = false;
    uint32_t pieces_off = 0, n_pieces = 0;  // node alignments in SiteReadViews::pieces
    LabelSet sequences;                     // supported sequence labels, bit i = SiteReadViews::label_names[i]
    uint32_t support_off = 0, n_support = 0;  // PG_PATH-coded path entries in SiteReadViews::support (packed batches only)
};

struct SiteReadViews
{
    std::vector<MappedReadView> reads;
    std::vector<NodeAlignment> pieces;
    std::vector<uint32_t> support;
    std::vector<std::string> label_names;  // the graph's edge labels, sorted
    uint32_t n_fragments = 0;
};
// views of reads held as objects (fragments numbered by fragment_id, graph CIGARs decoded)
SiteReadViews viewsOfReads(graphtools::Graph const& graph, std::vector<common::Read const*> const& reads);

common::Json fragmentStatistics(graphtools::Graph const& graph, SiteReadViews const& views);
common::Json alignmentStatistics(graphtools::Graph const& graph, SiteReadViews const& views);
inline common::Json fragmentStatistics(graphtools::Graph const& graph, std::vector<common::Read const*> const& reads)
{
    return fragmentStatistics(graph, viewsOfReads(graph, reads));
}
inline common::Json alignmentStatistics(graphtools::Graph const& graph, std::vector<common::Read const*> const& reads)
{
    return alignmentStatistics(graph, viewsOfReads(graph, reads));
}
}  // namespace paragraph
