// Exact-match stage front end (grm::PathAligner, src/c++/include/grm/PathAligner.hh): whole reads are matched base for base
// along graph walks, anchored on k-mers that occur on exactly one walk (k = 32 unless told otherwise).
#pragma once
#include <cstdint>
#include <memory>

#include "grm/Types.hh"

namespace grm
{
class PathAligner : public StageTally
{
public:
    explicit PathAligner(int32_t kmer_size = 32);
    PathAligner(PathAligner&&) noexcept;
    PathAligner& operator=(PathAligner&&) noexcept;
    virtual ~PathAligner();

    // the JSON paths are accepted for interface parity; the index is built from the graph itself
    void setGraph(GraphPtr graph, PathList const& paths);
    // a read that matches a walk end to end gets its graph_* fields and MAPPED (PathAligner.cpp:75-164); others are untouched
    void alignReads(ReadPtrs const& reads);
    void alignRead(common::Read& read);
    unsigned anchored() const { return anchored_; }  // reads with at least one unique k-mer hit

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
    unsigned anchored_ = 0;
};
}  // namespace grm
