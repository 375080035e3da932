// Site-batching replacement of the per-site align -> filter -> disambiguate -> count steps of
// paragraph::alignAndDisambiguate (src/c++/lib/paragraph/Disambiguation.cpp:152-361).
//
// The reference handles ONE site (~130 reads) per call; a GPU needs >= 1e5 reads in flight, so sites are
// accumulated with addSite() and processed by ONE run(): graphs -> pg_graphs_upload (+labels), all reads ->
// pg_batch_upload, pg_batch_align, pg_batch_count, then results are fanned back into the common::Read objects
// (graph_* fields, mapping status after the NonUniq/BadAlign filters, nodes/edges/sequences supported) and into
// per-site count tables keyed like read_counts_by_node / _by_edge / _by_sequence (ReadCounting.cpp:52-127).
#pragma once
#include <array>
#include <cstdint>
#include <list>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common/Read.hh"
#include "graphcore/Graph.hh"
#include "paragraph/PackedReads.hh"
#include "paragraph/Statistics.hh"

namespace paragraph
{
struct CountEntry
{
    uint64_t count = 0, reads = 0, fwd = 0, rev = 0;  // element, :READS, :FWD, :REV
    bool operator==(CountEntry const& o) const { return count == o.count && reads == o.reads && fwd == o.fwd && rev == o.rev; }
};

struct SiteCounts
{
    std::map<std::string, CountEntry> by_node;      // node name
    std::map<std::string, CountEntry> by_edge;      // "<from>_<to>"
    std::map<std::string, CountEntry> by_sequence;  // sorted, comma-joined label set ("total" of countPathFamilies)
    uint64_t aligned = 0, mapped = 0, bad_align = 0, nonuniq = 0;
};

// read_counts_by_edge as genotyping::BreakpointStatistics::addCounts reads it (BreakpointStatistics.cpp:108-143): fragment
// counts keyed "<from>_<to>"
inline std::map<std::string, int32_t> readCountsByEdge(SiteCounts const& c)
{
    std::map<std::string, int32_t> out;
    for (auto const& kv : c.by_edge)
        out[kv.first] = (int32_t)kv.second.count;
    return out;
}

struct BatchParameters
{
    bool remove_nonuniq_reads = true;  // paragraph --bad-align-nonuniq
    double bad_align_frac = 0.8;       // --bad-align-frac
    bool use_support_filters = true;   // production nodefilter / edgefilter
    int32_t kmer_len = 0;              // --bad-align-uniq-kmer-len: 0 = no KmerFilter, < 0 = auto-detect per graph
    // --path-sequence-matching (default ON in `paragraph`, OFF in grmpy): exact 32-mer-anchored path matching first; reads
    // it leaves unmapped -- or that the filter chain rejects -- go on to the gssw stage (CompositeAligner.cpp:78-103, 152)
    bool path_sequence_matching = false;
    // --kmer-sequence-matching / --klib-sequence-matching (default OFF in both tools): the k-mer seed stage (k = 16) and the
    // ksw stage between the path stage and gssw, each on the reads the stages before left unmapped or filtered
    // (CompositeAligner.cpp:105-150).  Both need the sites' paths (addSite's `paths`).
    bool kmer_sequence_matching = false;
    bool klib_sequence_matching = false;
    // Without the path stage: an EXACT shortcut in front of gssw (pg_batch_retire_exact_matches, include/paragraph_amd.h).  The
    // device's path kernel runs all the same, but only a read whose alignRead(AF_ALL) record its one full-length exact match
    // forces keeps that record and skips its four fills; every other read is aligned as if the kernel had not run.  Counts,
    // documents and genotypes are those of the plain gssw cascade (tests/test_gpu_exact.py, test_gpu_workflow.py); what it costs
    // is the path index per graph set, what it saves is the fills of the reads without a sequencing error.
    bool exact_match_shortcut = false;
    // which of SiteCounts' keyed tables to fill (the edge table always is): a caller that only genotypes needs neither
    bool node_counts = true, sequence_counts = true;
    // grm::ValidationAligner's bookkeeping for simulated reads (lib/grm/ValidationAligner.cpp:59-125) read by read after the
    // batch: total / MAPPED / MAPPED off the simulated path / non-unique BAD_ALIGN; object sites with paths only.  The totals
    // are the process-wide ones grm::ValidationAligner<CompositeAligner>::total() ... report (validationLogLines()).
    bool validate_alignments = false;
    // object sites: keep the reads the filter chain rejected (with the rejecting filter's name) for filtered() instead of
    // dropping them -- what Disambiguation.cpp:177-199 appends to "alignments" under FILTERED_ALIGNMENTS
    bool keep_filtered = false;
    unsigned alignment_flags = (unsigned)-1;
    int threads = 1;  // host threads for packing the reads and fanning the results back into them
    int device = 0;   // slot of the device list (setDevices / PG_DEVICES) this batch runs on
};

// The HIP devices the host library drives, one context each: `ordinals` as hipSetDevice takes them; the same ordinal may
// appear more than once (two contexts on one GPU -- how a 1-GPU box exercises the multi-device paths).  Default: the
// environment's PG_DEVICES ("0,1,2,3" or "all"), else PG_DEVICE, else device 0.  Must be called before the first device
// call (the list is fixed once a context exists; repeating the same list is fine).
void setDevices(std::vector<int> const& ordinals);
size_t deviceCount();
// page-locked staging memory obtained so far (the flat arrays of SiteBatcher::run live there)
size_t pinnedStagingBytes();
// CPUs this process may use at once (hardware threads, affinity mask, cgroup CPU bandwidth): the command lines' default for
// their host threads (the reference defaults to std::thread::hardware_concurrency(), grmpy.cpp:66)
int usableCpus();
// the [VALIDATION] lines logAlignerStats writes (lib/grm/Align.cpp:42-55) from the counters accumulated so far
std::vector<std::string> validationLogLines();

class SiteBatcher
{
public:
    SiteBatcher();
    ~SiteBatcher();
    // graph and reads must outlive run(); returns the site index
    // `paths` (grm::pathsFromJson) are only read by the k-mer and klib stages; they must outlive run() too
    size_t addSite(const graphtools::Graph* graph, std::vector<common::p_Read>* reads, std::list<graphtools::Path> const* paths = nullptr);
    // the packed form: the reads are only read; what the statistics need of the MAPPED ones is available through views()
    // after run().  One batch holds sites of one form only.
    size_t addSite(const graphtools::Graph* graph, PackedSite const* reads, std::list<graphtools::Path> const* paths = nullptr);
    // Aligns and counts every site added so far.  Afterwards each site's read vector holds only the MAPPED
    // reads (as grm::alignReads leaves it, Align.cpp:155) with their supports filled in.
    void run(BatchParameters const& parameters = BatchParameters());
    // run() in two halves, for a caller that has other work while the device has this batch: submit() packs and uploads the sites
    // and queues every device stage, then returns (nothing waits for the device); collect() waits for the batch, fetches its
    // records and makes views / reads and the site tables.  submit() returns false -- with nothing of the attempt left -- when the
    // batch holds a site outside the device's envelope: call run(), which isolates that site.  The sites (graphs, reads, paths)
    // must stay alive and untouched until collect() returns.
    bool submit(BatchParameters const& parameters = BatchParameters());
    void collect();
    size_t numSites() const;
    SiteCounts const& counts(size_t site) const;
    SiteReadViews const& views(size_t site) const;  // packed sites only
    // "" or why the device path could not take this site (outside its envelope: a read beyond PG_MAX_READ_LEN, more than
    // PG_MAX_NODES nodes, ...).  Such a site has empty counts and no MAPPED reads; the other sites of the batch are unaffected.
    std::string const& error(size_t site) const;
    // object sites run with keep_filtered: the reads of this site the filter chain rejected (graph_* fields of the rejected
    // alignment, status BAD_ALIGN) and the filter's message ("nonuniq", "bad_align", "kmer_tooshort", "kmer_uncov"), in input
    // order.  The caller may move the reads out.
    std::vector<std::pair<common::p_Read, std::string>>& filtered(size_t site);

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};
}  // namespace paragraph
