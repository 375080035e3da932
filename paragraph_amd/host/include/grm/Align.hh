// grm::alignReads (src/c++/include/grm/Align.hh:49-52): aligns the reads of one site and keeps only the MAPPED
// ones.  The reference cuts the read vector into `threads` chunks with one CompositeAligner each
// (Align.cpp:114-156); here the whole vector is ONE device batch and `threads` is ignored, so the surviving
// reads come back in input order (= the reference's 1-thread order).
#pragma once
#include <cstdint>
#include <list>
#include <vector>

#include "common/Read.hh"
#include "graphcore/Graph.hh"
#include "grm/Filter.hh"

namespace grm
{
void alignReads(
    const graphtools::Graph* graph, std::list<graphtools::Path> const& paths, std::vector<common::p_Read>& reads,
    ReadFilter const& filter, bool path_sequence_matching, bool graph_sequence_matching, bool klib_sequence_matching,
    bool kmer_sequence_matching, bool validate_alignments, uint32_t threads = 1);
}
