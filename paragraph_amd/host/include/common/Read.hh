// The record the aligners mutate: inputs (fragment id, bases, quals, strand, linear-genome placement as the BAM had it)
// and the graph_* outputs (fields of common::Read, src/c++/include/common/Read.hh:40-264).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "common/Json.hh"

namespace common
{
class Read
{
public:
    enum GraphMappingStatus
    {
        UNMAPPED = 0,
        MAPPED = 1,
        BAD_ALIGN = 2
    };
    Read() = default;
    Read(std::string fragment_id, std::string bases, std::string quals)
        : fragment_id_(std::move(fragment_id)), bases_(std::move(bases)), quals_(std::move(quals))
    {
    }
    void setCoreInfo(const std::string& fragment_id, const std::string& bases, const std::string& quals)
    {
        fragment_id_ = fragment_id;
        bases_ = bases;
        quals_ = quals;
    }
    const std::string& fragment_id() const { return fragment_id_; }
    void set_fragment_id(const std::string& v) { fragment_id_ = v; }
    const std::string& bases() const { return bases_; }
    void set_bases(const std::string& v) { bases_ = v; }
    const std::string& quals() const { return quals_; }
    void set_quals(const std::string& v) { quals_ = v; }
    bool is_reverse_strand() const { return is_reverse_strand_; }
    void set_is_reverse_strand(bool v) { is_reverse_strand_ = v; }
    bool is_first_mate() const { return is_first_mate_; }
    void set_is_first_mate(bool v) { is_first_mate_ = v; }
    // a default-constructed read is "not there" (ReadPair slots, getAlignedMate misses)
    bool is_initialized() const { return !bases_.empty(); }

    // placement on the linear reference, straight from the BAM record (BamReader.cpp:82-107)
    int32_t chrom_id() const { return chrom_id_; }
    void set_chrom_id(int32_t v) { chrom_id_ = v; }
    int32_t pos() const { return pos_; }
    void set_pos(int32_t v) { pos_ = v; }
    uint8_t mapq() const { return mapq_; }
    void set_mapq(uint8_t v) { mapq_ = v; }
    bool is_mapped() const { return is_mapped_; }
    void set_is_mapped(bool v) { is_mapped_ = v; }
    bool is_mate_mapped() const { return is_mate_mapped_; }
    void set_is_mate_mapped(bool v) { is_mate_mapped_ = v; }
    bool is_mate_reverse_strand() const { return is_mate_reverse_strand_; }
    void set_is_mate_reverse_strand(bool v) { is_mate_reverse_strand_ = v; }
    int32_t mate_chrom_id() const { return mate_chrom_id_; }
    void set_mate_chrom_id(int32_t v) { mate_chrom_id_ = v; }
    int32_t mate_pos() const { return mate_pos_; }
    void set_mate_pos(int32_t v) { mate_pos_ = v; }

    int32_t graph_pos() const { return graph_pos_; }
    void set_graph_pos(int32_t v) { graph_pos_ = v; }
    const std::string& graph_cigar() const { return graph_cigar_; }
    void set_graph_cigar(const std::string& v) { graph_cigar_ = v; }
    int32_t graph_mapq() const { return graph_mapq_; }
    void set_graph_mapq(int32_t v) { graph_mapq_ = v; }
    int32_t graph_alignment_score() const { return graph_alignment_score_; }
    void set_graph_alignment_score(int32_t v) { graph_alignment_score_ = v; }
    bool is_graph_alignment_unique() const { return is_graph_alignment_unique_; }
    void set_is_graph_alignment_unique(bool v) { is_graph_alignment_unique_ = v; }
    bool is_graph_reverse_strand() const { return is_graph_reverse_strand_; }
    void set_is_graph_reverse_strand(bool v) { is_graph_reverse_strand_ = v; }
    GraphMappingStatus graph_mapping_status() const { return status_; }
    void set_graph_mapping_status(GraphMappingStatus v) { status_ = v; }

    const std::vector<std::string>& graph_nodes_supported() const { return nodes_; }
    const std::vector<std::string>& graph_edges_supported() const { return edges_; }
    const std::vector<std::string>& graph_sequences_supported() const { return sequences_; }
    void clear_graph_nodes_supported() { nodes_.clear(); }
    void clear_graph_edges_supported() { edges_.clear(); }
    void clear_graph_sequences_supported() { sequences_.clear(); }
    void add_graph_nodes_supported(const std::string& v) { nodes_.push_back(v); }
    void add_graph_edges_supported(const std::string& v) { edges_.push_back(v); }
    void add_graph_sequences_supported(const std::string& v) { sequences_.push_back(v); }

    // identity of the sequenced read; graph_* results do not take part (Read.hh:134-141)
    bool operator==(const Read& o) const
    {
        return fragment_id_ == o.fragment_id_ && bases_ == o.bases_ && quals_ == o.quals_ && chrom_id_ == o.chrom_id_ && pos_ == o.pos_
            && mapq_ == o.mapq_ && is_mapped_ == o.is_mapped_ && is_first_mate_ == o.is_first_mate_
            && is_mate_mapped_ == o.is_mate_mapped_ && mate_chrom_id_ == o.mate_chrom_id_ && mate_pos_ == o.mate_pos_;
    }
    // the "alignments" entries of the output document: zero / false / empty fields are left out (Read.hh:142-233)
    Json toJson() const;

private:
    std::string fragment_id_, bases_, quals_;
    bool is_reverse_strand_ = false, is_first_mate_ = true;
    int32_t chrom_id_ = -1, pos_ = -1, mate_chrom_id_ = -1, mate_pos_ = -1;
    uint8_t mapq_ = 0;
    bool is_mapped_ = false, is_mate_mapped_ = false, is_mate_reverse_strand_ = false;
    int32_t graph_pos_ = 0, graph_mapq_ = 0, graph_alignment_score_ = 0;
    std::string graph_cigar_;
    bool is_graph_alignment_unique_ = false, is_graph_reverse_strand_ = false;
    GraphMappingStatus status_ = UNMAPPED;
    std::vector<std::string> nodes_, edges_, sequences_;
};
typedef std::unique_ptr<Read> p_Read;
typedef std::vector<p_Read> ReadBuffer;
}  // namespace common
