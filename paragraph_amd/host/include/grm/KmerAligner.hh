// k-mer seed stage front end (grm::KmerAligner<KMER_LENGTH>, src/c++/include/grm/KmerAligner.hh): ungapped placement along
// the JSON paths seeded by exact k-mers, at most 2 mismatches; a second, equally good but different placement makes the
// read BAD_ALIGN.  The template only fixes k; the work is in the base.
#pragma once
#include <memory>

#include "grm/Types.hh"

namespace grm
{
class KmerAlignerBase : public StageTally
{
public:
    explicit KmerAlignerBase(unsigned kmer_length);  // <= 16
    KmerAlignerBase(KmerAlignerBase&&) noexcept;
    KmerAlignerBase& operator=(KmerAlignerBase&&) noexcept;
    virtual ~KmerAlignerBase();

    void setGraph(GraphPtr graph, PathList const& paths);
    void alignReads(ReadPtrs const& reads);  // one device batch
    void alignRead(common::Read& read);

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

template <unsigned KMER_LENGTH> class KmerAligner : public KmerAlignerBase
{
public:
    KmerAligner() : KmerAlignerBase(KMER_LENGTH) {}
};
}  // namespace grm
