// genotyping::BreakpointStatistics / createBreakpointMap (src/c++/include/genotyping/BreakpointStatistics.hh, BreakpointFinder.hh;
// lib/genotyping/BreakpointStatistics.cpp:45-176, BreakpointFinder.cpp:49-76): every node with more than one successor
// ("<node>_") or predecessor ("_<node>") is a breakpoint; its edges carry the alleles (edge labels), alleles with identical edge
// sets are merged into one canonical allele.  addCounts takes read_counts_by_edge as a map (the reference reads the JSON key).
#pragma once
#include <cstdint>
#include <list>
#include <map>
#include <string>
#include <vector>

#include "graphcore/Graph.hh"

namespace genotyping
{
class BreakpointStatistics
{
public:
    BreakpointStatistics(graphtools::Graph const& graph, graphtools::NodeId node_id, bool forward);
    void addCounts(std::map<std::string, int32_t> const& read_counts_by_edge);
    int32_t getCount(std::string const& edge_or_allele_name) const;
    std::vector<std::string> const& edgeNames() const { return edge_names; }
    std::vector<std::string> const& canonicalAlleleNames() const { return canonical_allele_names; }
    std::list<std::string> const& allAlleleNames() const { return all_allele_names; }
    std::string const& getCanonicalAlleleName(std::string const& allele) const { return allele_name_to_canonical_allele_name.at(allele); }

private:
    std::vector<std::string> edge_names;
    std::map<std::string, size_t> edge_name_to_index;
    std::list<std::string> all_allele_names;
    std::vector<std::string> canonical_allele_names;
    std::map<std::string, std::vector<size_t>> edgename_to_alleles;
    std::map<std::string, size_t> allele_name_to_index;
    std::map<std::string, std::string> allele_name_to_canonical_allele_name;
    std::vector<int32_t> edge_counts, allele_counts;
};

typedef std::map<std::string, BreakpointStatistics> BreakpointMap;
BreakpointMap createBreakpointMap(graphtools::Graph const& graph);
}  // namespace genotyping
