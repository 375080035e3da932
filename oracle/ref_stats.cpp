/* ref_stats.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Harness around the REFERENCE's own paragraph::summarizeAlignments (src/c++/lib/paragraph/GraphSummaryStatistics.cpp) and
 * AlignmentStatistics.cpp, compiled from where they lie together with the graph-tools tarball and the vendored jsoncpp
 * (oracle/Makefile, target `ref`): the "alignment_statistics" block of a count document for a graph and a list of aligned
 * reads, as JSON text.  Only the glue below is written here: building the graphtools::Graph and the common::Read objects from
 * flat arrays.  Nothing under paragraph_amd/ links or calls this.
 */
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "common/Read.hh"
#include "graphcore/Graph.hh"
#include "json/json.h"
#include "paragraph/GraphSummaryStatistics.hh"

/* nodes: n_nodes names / sequences; edges: from / to + labels (label_off into label_names);
 * reads: graph_pos, CIGAR, graph-reverse flag, score, supported sequence names (seq_off into seq_names), all MAPPED.
 * Writes the JSON text (NUL-terminated) into out (capacity cap); returns its length, or -1 on an exception. */
extern "C" long pgrefs_alignment_statistics(
    uint32_t n_nodes, const char* const* node_names, const char* const* node_seqs, uint32_t n_edges, const uint32_t* from,
    const uint32_t* to, const uint32_t* label_off, const char* const* label_names, uint32_t n_reads, const int32_t* pos,
    const char* const* cigars, const uint8_t* reverse, const int32_t* score, const uint32_t* seq_off, const char* const* seq_names,
    char* out, size_t cap)
{
    try
    {
        graphtools::Graph graph(n_nodes, false);  // expansion off as graphFromJson (GraphInput.cpp:62)
        for (uint32_t i = 0; i < n_nodes; ++i)
        {
            graph.setNodeName(i, node_names[i]);
            graph.setNodeSeq(i, node_seqs[i]);
        }
        for (uint32_t e = 0; e < n_edges; ++e)
        {
            graph.addEdge(from[e], to[e]);
            for (uint32_t k = label_off[e]; k < label_off[e + 1]; ++k)
                graph.addLabelToEdge(from[e], to[e], label_names[k]);
        }
        common::ReadBuffer reads;
        for (uint32_t r = 0; r < n_reads; ++r)
        {
            std::unique_ptr<common::Read> read(new common::Read("f" + std::to_string(r), "A", "#"));
            read->set_graph_mapping_status(common::Read::MAPPED);
            read->set_graph_pos(pos[r]);
            read->set_graph_cigar(cigars[r]);
            read->set_is_graph_reverse_strand(reverse[r] != 0);
            read->set_graph_alignment_score(score[r]);
            for (uint32_t k = seq_off[r]; k < seq_off[r + 1]; ++k)
                read->add_graph_sequences_supported(seq_names[k]);
            reads.emplace_back(std::move(read));
        }
        Json::Value output(Json::objectValue);
        paragraph::summarizeAlignments(graph, reads, output);
        Json::StreamWriterBuilder builder;
        builder["indentation"] = "";
        builder["precision"] = 17;
        const std::string text = Json::writeString(builder, output["alignment_statistics"]);
        if (text.size() + 1 > cap)
            return -1;
        memcpy(out, text.c_str(), text.size() + 1);
        return (long)text.size();
    }
    catch (std::exception const&)
    {
        return -1;
    }
}
