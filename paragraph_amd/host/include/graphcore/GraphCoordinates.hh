// One linear coordinate per graph base, nodes laid end to end in id order, plus shortest walking distances between nodes
// (graphtools::GraphCoordinates, GT!/include/graphcore/GraphCoordinates.hh, GT!/src/graphcore/GraphCoordinates.cpp).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "graphcore/Graph.hh"

namespace graphtools
{
class GraphCoordinates
{
public:
    static constexpr uint64_t kNoPath = (uint64_t)-1;
    explicit GraphCoordinates(Graph const* graph);
    uint64_t canonicalPos(NodeId node, uint64_t offset = 0) const { return starts_.at(node) + offset; }
    // (first base, end offset in the last node) of a walk, ordered; an end offset of 0 leaves the end "unknown" (all ones)
    std::pair<uint64_t, uint64_t> canonicalStartAndEnd(Path const& path) const;
    void nodeAndOffset(uint64_t canonical_pos, NodeId& node, uint64_t& offset) const;
    // bases to walk from the smaller to the larger position, kNoPath when the later node cannot be reached
    uint64_t distance(uint64_t pos1, uint64_t pos2) const;
    Graph const& getGraph() const { return *graph_; }

private:
    Graph const* graph_;
    std::vector<uint64_t> starts_;
    std::map<std::pair<NodeId, NodeId>, uint64_t> end_to_start_;  // bases strictly between two non-adjacent nodes
};
}  // namespace graphtools
