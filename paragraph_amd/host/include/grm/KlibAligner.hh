// ksw stage front end (grm::KlibAligner, src/c++/include/grm/KlibAligner.hh:43-82): the read and its reverse complement are
// aligned locally (klib ksw: match 1, mismatch -4, gap 5 + 1, extend 1) against every JSON path; the best score wins, an
// equally good but different alignment makes the read BAD_ALIGN.
#pragma once
#include <memory>

#include "grm/Types.hh"

namespace grm
{
class KlibAligner : public StageTally
{
public:
    KlibAligner();
    KlibAligner(KlibAligner&&) noexcept;
    KlibAligner& operator=(KlibAligner&&) noexcept;
    virtual ~KlibAligner();

    void setGraph(GraphPtr graph, PathList const& paths);  // <= 30 paths, whole nodes
    void alignReads(ReadPtrs const& reads);                // one device batch; reads <= 512 bases
    void alignRead(common::Read& read);

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};
}  // namespace grm
