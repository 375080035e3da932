// --validate-alignments: the aligner wrapper that compares every alignment with the path a simulated read was drawn from
// (grm::ValidationAligner, src/c++/include/grm/ValidationAligner.hh:44-74, lib/grm/ValidationAligner.cpp:41-125).  The path
// id is the part of the fragment id before the first '_' (Path::encode() of the simulated path); an alignment "supports" it
// when the node sequence of its CIGAR occurs in the path's node sequence.  Statistics only: the reads come out as the
// wrapped aligner leaves them.  The counters are process-wide like the original's static members.
//
// Two details of the original are kept as they are: getNodes() copies the characters outside the brackets one by one, so a
// two-digit node id "12" turns into "1->2" while the path side has "12"; and the counters are never reset.
#pragma once
#include <list>
#include <string>
#include <unordered_map>
#include <vector>

#include "grm/CompositeAligner.hh"

namespace grm
{
template <typename AlignerT> class ValidationAligner : private AlignerT
{
public:
    ValidationAligner(AlignerT&& aligner, const graphtools::Graph* graph, std::list<graphtools::Path> const& paths);
    virtual ~ValidationAligner() {}

    using AlignerT::setGraph;
    void alignRead(common::Read& read, ReadFilter filter);
    // batched form: one device launch per stage, then the same bookkeeping read by read
    void alignReads(std::vector<common::Read*> const& reads, ReadFilter filter);
    const AlignerT& base() const { return *this; }
    static unsigned mismapped();
    static unsigned repeats();
    static unsigned aligned();
    static unsigned total();

private:
    void account(common::Read& read);
    std::unordered_map<std::string, std::string> pathNodes_;
    static std::string getNodes(const std::string& cigar);
    static std::string getSimulatedPathId(common::Read& read);
};

extern template class ValidationAligner<CompositeAligner>;
}  // namespace grm
