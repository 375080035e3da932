// Many-site version of the `paragraph` / `grmpy` per-site loop.
//
// The reference handles one (graph, sample) pair at a time: paragraph::Parameters::load -> common::extractReads ->
// paragraph::alignAndDisambiguate (lib/paragraph/Disambiguation.cpp:152-361), driven per sample and graph by
// grmpy::alignSingleSample (lib/grmpy/AlignSamples.cpp:115-172) and grmpy::Workflow::alignSamples (Workflow.cpp:108-146),
// then grmpy::countAndGenotype (CountAndGenotype.cpp:46-88) per graph.  Here read extraction for ALL pairs runs first on
// host threads (one BamReader per thread), thousands of pairs go through ONE SiteBatcher::run() on the device at a time,
// and the count documents and genotypes are assembled from the batch results while the next batch's reads are extracted.
#pragma once
#include <list>
#include <memory>
#include <functional>
#include <string>
#include <vector>

#include "common/Fasta.hh"
#include "common/Json.hh"
#include "common/Read.hh"
#include "common/ReadReader.hh"
#include "common/Region.hh"
#include "genotyping/SampleInfo.hh"
#include "graphcore/Graph.hh"
#include "paragraph/PackedReads.hh"

namespace paragraph
{
// seconds spent in the workflow's phases, summed over lanes (so they can add up to more than the wall clock), filled in
// when a Parameters object points at one
struct Timings
{
    double load_graphs = 0, extract_reads = 0, device_batch = 0, documents = 0, genotypes = 0, release = 0;
    size_t sites = 0, reads = 0, batches = 0, lanes = 1;
};

// paragraph::Parameters (include/paragraph/Parameters.hh:44-126): what to compute and emit for a site
struct Parameters
{
    enum output_options
    {
        ALIGNMENTS = 0x01,
        FILTERED_ALIGNMENTS = 0x02,  // the reads the filter chain rejected, with the filter's message under "error", + the filter tallies
        NODE_READ_COUNTS = 0x08,
        EDGE_READ_COUNTS = 0x10,
        PATH_READ_COUNTS = 0x20,
        DETAILED_READ_COUNTS = 0x40,
        // VARIANTS 0x04, PATH_COVERAGE 0x80, NODE_COVERAGE 0x100, HAPLOTYPES 0x200 are not produced
    };
    int max_reads = 10000;
    float bad_align_frac = 0.8f;
    int output_options_ = NODE_READ_COUNTS | EDGE_READ_COUNTS | PATH_READ_COUNTS;  // the `paragraph` tool's default
    bool path_sequence_matching = true;    // `paragraph` default; grmpy turns it off
    bool graph_sequence_matching = true;
    bool kmer_sequence_matching = false;   // the optional seed stages between the path stage and gssw
    bool klib_sequence_matching = false;
    bool remove_nonuniq_reads = true;
    bool exact_match_shortcut = false;     // BatchParameters::exact_match_shortcut (only without the seed stages)
    int kmer_len = 0;
    int threads = 1;  // host threads for read extraction and document assembly
    int device = 0;   // slot of the device list (paragraph::setDevices / PG_DEVICES) the batch runs on
    Timings* timings = nullptr;
    // false: a count document holds only what the site's OWN analysis produced (statistics + the enabled tables), not a copy of
    // the graph description.  grmpy::genotypeGraphs runs that way: its count documents never leave the process -- the genotyper
    // takes the edge table and the statistics from them and the description from the loaded graph (as the reference's
    // countAndGenotype does from the graph file, lib/grmpy/CountAndGenotype.cpp:46-88).
    bool description_in_document = true;
    // `paragraph --validate-alignments`: grm::ValidationAligner's bookkeeping for simulated reads (fragment ids "<path id>_...");
    // read objects instead of packed reads, the sites need their paths
    bool validate_alignments = false;
    bool output_enabled(output_options o) const { return (output_options_ & o) != 0; }
};

// A loaded graph description (Parameters::load, lib/paragraph/Parameters.cpp:39-90): the document with a "graph" wrapper
// flattened, its target regions, "max_reads" override and the longest explicit node sequence.
struct GraphDescription
{
    // `opened_reference`: an already opened FASTA of reference_path to share between loads (may be null)
    static GraphDescription load(
        std::string const& graph_path, std::string const& reference_path, std::string const& override_target_regions = "",
        common::FastaFile const* opened_reference = nullptr);
    static GraphDescription fromJson(
        common::Json root, std::string const& reference_path, std::string const& override_target_regions = "",
        common::FastaFile const* opened_reference = nullptr);
    common::Json description;
    std::string reference_path;
    std::list<common::Region> target_regions;
    size_t longest_alt_insertion = 0;
    int64_t max_reads = -1;  // < 0: not given
    std::shared_ptr<graphtools::Graph> graph;
    std::list<graphtools::Path> paths;
};

struct SiteInput
{
    GraphDescription const* description = nullptr;
    common::ReadBuffer* reads = nullptr;  // in: extracted reads; out: the reads the document was built from
};

struct PackedSiteInput
{
    GraphDescription const* description = nullptr;
    PackedSite const* reads = nullptr;  // from extractPacked; not modified
};

// One device batch over all sites; returns one count document per site: the description plus "reference",
// "read_counts_by_node" / "_by_edge" / "_by_sequence", "fragment_statistics", "alignment_statistics" and optionally "alignments".
std::vector<common::Json> alignAndDisambiguateBatch(Parameters const& parameters, std::vector<SiteInput> const& sites);
// the same documents (minus "alignments", which this form cannot give) from reads kept in the packed form
std::vector<common::Json> alignAndDisambiguateBatch(Parameters const& parameters, std::vector<PackedSiteInput> const& sites);
// The `paragraph` tool's job (lib/paragraph/Workflow.cpp:77-233): every graph against the given BAM(s).  One BAM: one count
// document per graph with "bam" = its path.  Several BAMs: their reads are pooled per graph ("joint inputs") and "bam" lists
// them.  All graphs go through the device in batches of `sites_per_batch`; target_regions ("chr:a-b,chr:c-d") overrides the
// graphs' own.  Documents come back in the order of graph_paths.
std::vector<common::Json> countGraphs(
    Parameters const& parameters, std::vector<std::string> const& graph_paths, std::string const& reference_path,
    std::vector<std::string> const& bam_paths, std::vector<std::string> const& bam_index_paths = {},
    std::string const& target_regions = "", size_t sites_per_batch = 0 /* 0 = 192 */);
// single-site convenience with the reference's shape
common::Json alignAndDisambiguate(Parameters const& parameters, GraphDescription const& description, common::ReadBuffer& all_reads);
}  // namespace paragraph

namespace grmpy
{
// grmpy::Parameters (include/grmpy/Parameters.hh:30-74)
struct Parameters
{
    int threads = 1;
    int max_reads = 10000;
    float bad_align_frac = 0.8f;
    bool path_sequence_matching = false;
    bool graph_sequence_matching = true;
    bool klib_sequence_matching = false;
    bool kmer_sequence_matching = false;
    int bad_align_uniq_kmer_len = 0;
    // gssw-only cascade (the default here): reads whose alignRead record their one exact full-length match forces skip their
    // fills (paragraph::BatchParameters::exact_match_shortcut); the documents are the same either way
    bool exact_match_shortcut = false;
    bool output_alignments = false;  // keep "alignments" in the per-sample documents (the original writes them to a folder)
    // grmpy -A / --alignment-output-folder (lib/grmpy/AlignSamples.cpp:57-109, 120-162): when this names an existing directory,
    // every (sample, graph) pair's count document WITH the per-read records -- the reads the filters rejected included, and the
    // filter tallies -- is written there as <sample>-<graph ID>-<target regions>.json.gz; as in the original the folder switches
    // the filtered records and tallies on for the documents the genotyper reads too.  (Of paragraph::Parameters::ALL the keys
    // this build produces: no "variants" / "node_coverage" / "path_coverage" / "phasing".)
    std::string alignment_output_folder;
    // keep the reads of a site as flat arrays instead of common::Read objects (several times less host work); switched off
    // automatically when output_alignments needs the per-read records
    bool packed_reads = true;
    size_t sites_per_batch = 0;      // (graph, sample) pairs per device batch; bounds host memory and sets the pipeline grain.  0 = 192
                                     // (10 000 sites, 16 threads: gssw only 128 78.5 k sites/s, 192 79 k, 256 79 k, 384 70.5 k; with the
                                     // path stage 192 81 k, 256 79 k, 384 73 - 76 k -- profiles/r05_e2e_lanes_ab2.jsonl,
                                     // r05_e2e_path_batch_ab.jsonl.  While every batch's seed chain carried its table uploads the
                                     // path cascade wanted 384: fewer, longer chains.)
    int lanes = 0;                   // chunks in flight: each lane carries one chunk through all stages with threads / lanes
                                     // workers; 0 = one lane per four threads, at most eight per device, at least one per device
    std::vector<int> devices;        // HIP ordinals to spread the lanes over (lane l -> devices[l % n]); empty = the list of
                                     // paragraph::setDevices / PG_DEVICES / PG_DEVICE / {0}
    paragraph::Timings* timings = nullptr;
    // When set, genotypeGraphs writes every genotype document as JSON text into (*genotype_text)[graph] (resized to the graph
    // list) right where it is made -- on the lane that made it, beside the other lanes' device batches -- and the returned
    // documents are left empty: a caller that only writes the documents out (the command line, pgw_genotype_graphs) then does
    // not serialise 40 MB on one thread after the last lane has finished.
    std::vector<std::string>* genotype_text = nullptr;
    int genotype_text_indent = -1;   // Json::dump's indent: < 0 one line
    // called by a lane when the texts of graphs [first, last) are in *genotype_text (any order of ranges, from any lane's thread):
    // a caller that writes the documents out in order can do so while the other lanes work (pgw_genotype_graphs does)
    std::function<void(size_t first, size_t last)> genotype_text_ready;
};

// extraction + alignment + counting of ONE sample against ONE graph; stores the count document in the sample
void alignSingleSample(
    Parameters const& parameters, std::string const& graph_path, std::string const& reference_path, common::ReadReader& reader,
    genotyping::SampleInfo& sample);
// genotypes of one graph from the samples' count documents (graph_path "" = take the graph from the first sample's document)
common::Json countAndGenotype(
    std::string const& graph_path, std::string const& reference_path, std::string const& genotyping_parameter_path,
    genotyping::Samples const& samples);
// every graph x every sample, `sites_per_batch` pairs per device batch, `lanes` batches in flight (their host stages
// overlap, the device stage is serialised); returns one genotype document per graph, in the order given
std::vector<common::Json> genotypeGraphs(
    Parameters const& parameters, std::vector<std::string> const& graph_paths, std::string const& reference_path,
    genotyping::Samples const& samples, std::string const& genotyping_parameter_path);
// the [first, last) graph ranges genotypeGraphs cuts its work into: all lanes start at once, so the first round is staggered
// (lane k takes (k+1)/lanes of a chunk: the device gets work while the other lanes still prepare) and the last chunks shrink
// to a quarter of a chunk at least, so that the lanes finish together; plain equal chunks when there is one lane or no
// more chunks than lanes
std::vector<std::pair<size_t, size_t>> chunkSchedule(size_t n_graphs, size_t graphs_per_chunk, size_t lanes);
}  // namespace grmpy
