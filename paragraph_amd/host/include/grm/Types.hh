// Short names shared by the aligner front ends of this directory.
#pragma once
#include <list>
#include <vector>

#include "common/Read.hh"
#include "graphcore/Graph.hh"

namespace grm
{
using GraphPtr = const graphtools::Graph*;
using PathList = std::list<graphtools::Path>;
using ReadPtrs = std::vector<common::Read*>;

// Bookkeeping every seed-stage front end shares: how many reads it was given and how many it mapped.
class StageTally
{
public:
    unsigned attempted() const { return attempted_; }
    unsigned mapped() const { return mapped_; }

protected:
    unsigned attempted_ = 0, mapped_ = 0;
};
}  // namespace grm
