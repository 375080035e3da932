// grm::KlibAligner over the device library (src/c++/include/grm/KlibAligner.hh:43-82): local alignment (klib ksw,
// match 1 / mismatch -4 / gap 5+1 / extend 1) of the read and its reverse complement against every JSON path;
// the best score wins, an equally good but different alignment makes the read BAD_ALIGN.
#pragma once
#include <list>
#include <memory>
#include <vector>

#include "common/Read.hh"
#include "graphcore/Graph.hh"

namespace grm
{
class KlibAligner
{
public:
    KlibAligner();
    virtual ~KlibAligner();
    KlibAligner(KlibAligner&& rhs) noexcept;
    KlibAligner& operator=(KlibAligner&& rhs) noexcept;
    void setGraph(graphtools::Graph const* g, std::list<graphtools::Path> const& paths);
    void alignRead(common::Read& read);
    // batched form (one device launch); same per-read semantics
    void alignReads(std::vector<common::Read*> const& reads);
    unsigned attempted() const { return attempted_; }
    unsigned mapped() const { return mapped_; }

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
    unsigned attempted_ = 0, mapped_ = 0;
};
}  // namespace grm
