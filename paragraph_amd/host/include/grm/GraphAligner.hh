// grm::GraphAligner over the MI355X device library: same interface as the reference's gssw wrapper
// (src/c++/include/grm/GraphAligner.hh:36-88).  setGraph uploads the graph (both directions are derived on the
// device side), alignRead / align run ONE read through pg_align_batch -- correct but launch-bound; use
// grm::alignReads (Align.hh) or DeviceBatch for throughput.
#pragma once
#include <memory>
#include <string>

#include "common/Read.hh"
#include "graphcore/Graph.hh"

namespace grm
{
class GraphAligner
{
public:
    GraphAligner();
    virtual ~GraphAligner();
    GraphAligner(GraphAligner&& rhs) noexcept;
    GraphAligner& operator=(GraphAligner&& rhs) noexcept;

    void setGraph(graphtools::Graph const* g);
    std::string align(const std::string& read, int& mapq, int& position, int& score) const;

    static const unsigned int AF_CIGAR = 0x01;
    static const unsigned int AF_BOTH_STRANDS = 0x02;
    static const unsigned int AF_REVERSE_GRAPH = 0x04;
    static const unsigned int AF_ALL = (unsigned int)-1;

    void alignRead(common::Read& read, unsigned int alignment_flags = AF_ALL) const;
    // batched form: every non-empty read is aligned and its graph_* fields are set; status is not touched
    void alignReads(std::vector<common::Read*> const& reads, unsigned int alignment_flags = AF_ALL) const;

private:
    struct GraphAlignerImpl;
    std::unique_ptr<GraphAlignerImpl> _impl;
};
}  // namespace grm
