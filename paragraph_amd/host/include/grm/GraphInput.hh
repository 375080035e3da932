// Graph descriptions (share/schema/graph_schema.json) -> graphtools::Graph / paths (grm::graphFromJson / pathsFromJson,
// src/c++/include/grm/GraphInput.hh:36-52, lib/grm/GraphInput.cpp:44-197).
#pragma once
#include <list>
#include <string>

#include "common/Fasta.hh"
#include "common/Json.hh"
#include "graphcore/Graph.hh"

namespace grm
{
// Node ids follow the order of "nodes".  A first / last node called source / sink (any case) gets the sequence "X"; other
// nodes take "sequence", or the reference interval(s) named by "reference" (all listed intervals must spell the same
// bases).  Node-level "sequences" label every edge in or out of the node.  `in` may wrap everything in a "graph" member.
// Throws std::runtime_error where the original asserts (no nodes, duplicate names, node without sequence / reference, edges
// not an array, unknown edge ends).
graphtools::Graph graphFromJson(common::Json const& in, std::string const& reference, bool store_ref_sequence = true);
// the same with an already opened reference (shared between threads when many graphs are loaded)
graphtools::Graph graphFromJson(common::Json const& in, common::FastaFile const& reference, bool store_ref_sequence = true);
// every path runs from offset 0 of its first node to the last base of its last node
std::list<graphtools::Path> pathsFromJson(graphtools::Graph const* graph, common::Json const& in_paths);
}  // namespace grm
