// grm::KmerAligner<KMER_LENGTH> over the device library (src/c++/include/grm/KmerAligner.hh): ungapped alignment
// along the JSON paths seeded by exact k-mers, at most 2 mismatches; a second equally good but different
// alignment makes the read BAD_ALIGN.
#pragma once
#include <list>
#include <memory>
#include <vector>

#include "common/Read.hh"
#include "graphcore/Graph.hh"

namespace grm
{
class KmerAlignerBase
{
public:
    explicit KmerAlignerBase(unsigned kmer_length);
    virtual ~KmerAlignerBase();
    KmerAlignerBase(KmerAlignerBase&& rhs) noexcept;
    KmerAlignerBase& operator=(KmerAlignerBase&& rhs) noexcept;
    void setGraph(graphtools::Graph const* g, std::list<graphtools::Path> const& paths);
    void alignRead(common::Read& read);
    void alignReads(std::vector<common::Read*> const& reads);
    unsigned attempted() const { return attempted_; }
    unsigned mapped() const { return mapped_; }

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
    unsigned attempted_ = 0, mapped_ = 0;
};

template <unsigned KMER_LENGTH> class KmerAligner : public KmerAlignerBase
{
public:
    KmerAligner() : KmerAlignerBase(KMER_LENGTH) {}
};
}  // namespace grm
