// grm::PathAligner over the device library (src/c++/include/grm/PathAligner.hh): exact matching of whole reads
// along graph paths, anchored on k-mers that occur on exactly one path (k = 32 by default).
#pragma once
#include <list>
#include <memory>
#include <vector>

#include "common/Read.hh"
#include "graphcore/Graph.hh"

namespace grm
{
class PathAligner
{
public:
    explicit PathAligner(int32_t kmer_size = 32);
    virtual ~PathAligner();
    PathAligner(PathAligner&& rhs) noexcept;
    PathAligner& operator=(PathAligner&& rhs) noexcept;

    void setGraph(graphtools::Graph const* g, std::list<graphtools::Path> const& paths);
    // Sets the graph_* fields and MAPPED status when the read matches a path end to end (PathAligner.cpp:75-164)
    void alignRead(common::Read& read);
    void alignReads(std::vector<common::Read*> const& reads);

    unsigned attempted() const { return attempted_; }
    unsigned mapped() const { return mapped_; }
    unsigned anchored() const { return anchored_; }

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
    unsigned attempted_ = 0, mapped_ = 0, anchored_ = 0;
};
}  // namespace grm
