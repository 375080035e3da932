// The per-site entry of the realignment path (grm::alignReads, src/c++/include/grm/Align.hh:49-52): run the cascade over the
// reads of one site and keep only the MAPPED ones.  Where the reference cuts the vector into `threads` chunks with one
// CompositeAligner each (Align.cpp:114-156), the whole vector is ONE device batch here; `threads` is accepted and ignored,
// so survivors come back in input order (the reference's single-thread order).  `validate_alignments` is accepted too.
#pragma once
#include <cstdint>

#include "grm/Filter.hh"
#include "grm/Types.hh"

namespace grm
{
void alignReads(
    GraphPtr graph, PathList const& paths, std::vector<common::p_Read>& reads, ReadFilter const& filter,
    bool path_sequence_matching, bool graph_sequence_matching, bool klib_sequence_matching, bool kmer_sequence_matching,
    bool validate_alignments, uint32_t threads = 1);
}
