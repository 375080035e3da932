// gssw stage front end with the reference's call surface (grm::GraphAligner, src/c++/include/grm/GraphAligner.hh:36-88) on
// top of the device library.  setGraph uploads the graph (both orientations are derived on the device); alignRead sends
// ONE read through a device batch -- correct but launch-bound, meant for drop-in use and tests; alignReads is the same
// per-read operation for a whole vector in one batch (what grm::alignReads and the SiteBatcher build on).
#pragma once
#include <memory>
#include <string>

#include "grm/Types.hh"

namespace grm
{
class GraphAligner
{
public:
    // what alignRead computes: the CIGAR, the reverse-complement strand too, the reversed-graph fills for uniqueness
    enum AlignmentFlag : unsigned int
    {
        AF_CIGAR = 1u << 0,
        AF_BOTH_STRANDS = 1u << 1,
        AF_REVERSE_GRAPH = 1u << 2,
        AF_ALL = ~0u
    };

    GraphAligner();
    GraphAligner(GraphAligner&& other) noexcept;
    GraphAligner& operator=(GraphAligner&& other) noexcept;
    virtual ~GraphAligner();

    void setGraph(GraphPtr graph);
    // every non-empty read gets its graph_* fields; the mapping status is left to the caller (the cascade sets it)
    void alignReads(ReadPtrs const& reads, unsigned int alignment_flags = AF_ALL) const;
    void alignRead(common::Read& read, unsigned int alignment_flags = AF_ALL) const;
    // bare form: returns the graph CIGAR of `bases` (forward strand, forward graph) and its mapq / start / score
    std::string align(const std::string& bases, int& mapq, int& position, int& score) const;

private:
    struct GraphAlignerImpl;
    std::unique_ptr<GraphAlignerImpl> _impl;
};
}  // namespace grm
