// Minimal sequence-graph model for the host side of the MI355X realignment core.
// Mirrors the parts of graphtools::Graph the hot path reads (GT!/include/graphcore/Graph.hh:56-101):
// nodes with name + sequence, edges in topological order with labels, adjacency kept sorted.
#pragma once
#include <cstdint>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace graphtools
{
using NodeId = uint32_t;
using NodeIdPair = std::pair<NodeId, NodeId>;

class Graph
{
public:
    explicit Graph(size_t num_nodes = 0, bool is_sequence_expansion_required = true)
        : names_(num_nodes), seqs_(num_nodes), succ_(num_nodes), pred_(num_nodes), expand_(is_sequence_expansion_required)
    {
    }
    size_t numNodes() const { return seqs_.size(); }
    size_t numEdges() const { return labels_.size(); }
    bool isSequenceExpansionRequired() const { return expand_; }
    const std::string& nodeName(NodeId n) const { return names_.at(n); }
    void setNodeName(NodeId n, const std::string& name) { names_.at(n) = name; }
    const std::string& nodeSeq(NodeId n) const { return seqs_.at(n); }
    // Degenerate (IUPAC) symbols would need node expansion (GraphAligner.cpp:125-133); production graphs from
    // grm::graphFromJson never expand (GraphInput.cpp:62), and this implementation refuses them.
    void setNodeSeq(NodeId n, const std::string& seq) { seqs_.at(n) = seq; }
    void addEdge(NodeId from, NodeId to)
    {
        check(from);
        check(to);
        if (hasEdge(from, to))
            throw std::logic_error("Graph already contains edge (" + std::to_string(from) + " ," + std::to_string(to) + ")");
        if (from > to)
            throw std::logic_error("Edge (" + std::to_string(from) + " ," + std::to_string(to) + ") breaks topological order");
        labels_[{ from, to }];
        succ_[from].insert(to);
        pred_[to].insert(from);
    }
    bool hasEdge(NodeId from, NodeId to) const { return labels_.count({ from, to }) != 0; }
    void addLabelToEdge(NodeId from, NodeId to, const std::string& label)
    {
        auto it = labels_.find({ from, to });
        if (it == labels_.end())
            throw std::logic_error("There is no edge between " + std::to_string(from) + " and " + std::to_string(to));
        it->second.insert(label);
    }
    const std::set<std::string>& edgeLabels(NodeId from, NodeId to) const { return labels_.at({ from, to }); }
    std::set<std::string> allLabels() const
    {
        std::set<std::string> all;
        for (auto const& e : labels_)
            all.insert(e.second.begin(), e.second.end());
        return all;
    }
    const std::set<NodeId>& successors(NodeId n) const { return succ_.at(n); }
    const std::set<NodeId>& predecessors(NodeId n) const { return pred_.at(n); }

private:
    void check(NodeId n) const
    {
        if (n >= seqs_.size())
            throw std::logic_error("Node with id " + std::to_string(n) + " does not exist");
    }
    std::vector<std::string> names_, seqs_;
    std::vector<std::set<NodeId>> succ_, pred_;
    std::map<NodeIdPair, std::set<std::string>> labels_;
    bool expand_;
};

// Paths are only consumed by the path/k-mer/ksw seed aligners; the gssw cascade ignores them.
struct Path
{
    const Graph* graph = nullptr;
    int32_t start_position = 0;
    std::vector<NodeId> nodes;
    int32_t end_position = 0;
    std::vector<NodeId> const& nodeIds() const { return nodes; }
    // "(first@start)-(mid)-...-(last@end)"; a one-node path is "(n@start)-(n@end)" (GT!/src/graphcore/Path.cpp:152-179)
    std::string encode() const
    {
        std::string out;
        for (size_t i = 0; i < nodes.size(); ++i)
        {
            const std::string name = std::to_string(nodes[i]);
            std::string piece;
            if (i == 0)
                piece = "(" + name + "@" + std::to_string(start_position) + ")";
            if (i + 1 == nodes.size())
                piece += "-(" + name + "@" + std::to_string(end_position) + ")";
            if (i != 0 && i + 1 != nodes.size())
                piece = "-(" + name + ")";
            out += piece;
        }
        return out;
    }
};
}  // namespace graphtools
