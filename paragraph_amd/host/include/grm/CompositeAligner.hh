// grm::CompositeAligner (src/c++/include/grm/CompositeAligner.hh:44-91): the aligner cascade
// path -> kmer -> klib -> gssw with a filter after each stage; every stage runs on the device.
#pragma once
#include <list>
#include <vector>

#include "grm/Filter.hh"
#include "grm/GraphAligner.hh"
#include "grm/KlibAligner.hh"
#include "grm/KmerAligner.hh"
#include "grm/PathAligner.hh"

namespace grm
{
class CompositeAligner
{
public:
    CompositeAligner(
        bool pathMatching, bool graphMatching, bool klibMatching, bool kmerMatching,
        unsigned grapAlignmentflags = GraphAligner::AF_ALL);
    virtual ~CompositeAligner();
    CompositeAligner(CompositeAligner&& rhs) noexcept;
    CompositeAligner& operator=(CompositeAligner&& rhs) noexcept = delete;

    void setGraph(graphtools::Graph const* graph, std::list<graphtools::Path> const& paths);
    void alignRead(common::Read& read, ReadFilter filter);
    // batched cascade over many reads (one device launch); same per-read semantics and counters
    void alignReads(std::vector<common::Read*> const& reads, ReadFilter filter);

    unsigned attempted() const { return attempted_; }
    unsigned filtered() const { return filtered_; }
    unsigned mappedKlib() const { return mappedKlib_; }
    unsigned mappedPath() const { return mappedPath_; }
    unsigned anchoredPath() const { return anchoredPath_; }
    unsigned mappedKmers() const { return mappedKmers_; }
    unsigned mappedSw() const { return mappedSw_; }

private:
    const bool pathMatching_, graphMatching_, klibMatching_, kmerMatching_;
    const unsigned int grapAlignmentflags_;
    PathAligner pathAligner_;
    GraphAligner graphAligner_;
    KmerAligner<16> kmerAligner_;
    KlibAligner klibAligner_;
    unsigned attempted_ = 0, filtered_ = 0, mappedKlib_ = 0, mappedPath_ = 0, anchoredPath_ = 0, mappedKmers_ = 0,
             mappedSw_ = 0;
};
}  // namespace grm
