// The aligner cascade of the reference (grm::CompositeAligner, src/c++/include/grm/CompositeAligner.hh:44-91) on top of the
// device library: exact path matching, then k-mer seeds, then ksw per path, then gssw -- each stage only sees the reads the
// previous ones left unmapped or that the caller's filter rejected.  Every stage runs on the GPU.
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
    struct Tally
    {
        unsigned attempted = 0, filtered = 0, path = 0, anchored = 0, kmers = 0, klib = 0, sw = 0;
    };

public:
    // which stages run; `grapAlignmentflags` are GraphAligner's AF_* bits for the gssw stage
    CompositeAligner(bool pathMatching, bool graphMatching, bool klibMatching, bool kmerMatching, unsigned grapAlignmentflags = GraphAligner::AF_ALL);
    CompositeAligner(CompositeAligner&&) noexcept;
    CompositeAligner& operator=(CompositeAligner&&) noexcept = delete;
    virtual ~CompositeAligner();

    // the graph must outlive the aligner; `paths` feed the k-mer and klib stages
    void setGraph(graphtools::Graph const* graph, std::list<graphtools::Path> const& paths);

    // one read through the cascade (the reference's entry point) ...
    void alignRead(common::Read& read, ReadFilter filter);
    // ... or many at once: one device launch per stage, same per-read outcome and counters
    void alignReads(std::vector<common::Read*> const& reads, ReadFilter filter);

    // per-stage statistics, as logged by grm::alignReads
    unsigned attempted() const { return tally_.attempted; }
    unsigned mappedPath() const { return tally_.path; }
    unsigned anchoredPath() const { return tally_.anchored; }
    unsigned mappedKmers() const { return tally_.kmers; }
    unsigned mappedKlib() const { return tally_.klib; }
    unsigned mappedSw() const { return tally_.sw; }
    unsigned filtered() const { return tally_.filtered; }

private:
    struct Stages
    {
        bool path, graph, klib, kmer;
    };
    const Stages on_;
    const unsigned gssw_flags_;
    PathAligner path_stage_;
    KmerAligner<16> kmer_stage_;
    KlibAligner klib_stage_;
    GraphAligner gssw_stage_;
    Tally tally_;
};
}  // namespace grm
