// paragraph::alignAndDisambiguateBatch / grmpy::genotypeGraphs -- see include/paragraph/Workflow.hh.
#include "paragraph/Workflow.hh"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <mutex>
#include <set>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <unordered_map>

#include "common/BamReader.hh"
#include "common/ReadExtraction.hh"
#include "genotyping/BreakpointStatistics.hh"
#include "genotyping/GraphBreakpointGenotyper.hh"
#include "genotyping/PopulationStatistics.hh"
#include "grm/GraphInput.hh"
#include "paragraph/SiteBatcher.hh"
#include "paragraph/Statistics.hh"
#include "parallel.hh"
#include <sys/stat.h>
#include <zlib.h>

using common::Json;

namespace
{
double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

using pghost::parallelFor;
}  // namespace

namespace paragraph
{
GraphDescription GraphDescription::load(
    std::string const& graph_path, std::string const& reference_path, std::string const& override_target_regions,
    common::FastaFile const* opened_reference)
{
    return fromJson(Json::parseFile(graph_path), reference_path, override_target_regions, opened_reference);
}

GraphDescription GraphDescription::fromJson(
    Json root, std::string const& reference_path, std::string const& override_target_regions, common::FastaFile const* opened_reference)
{
    GraphDescription d;
    d.reference_path = reference_path;
    if (root.isMember("graph"))
    {
        const Json inner = root["graph"];
        for (auto const& kv : inner.members())
            root[kv.first] = kv.second;
        root.removeMember("graph");
    }
    if (!override_target_regions.empty())
    {
        std::stringstream ss(override_target_regions);
        std::string piece;
        while (std::getline(ss, piece, ','))  // stringutil::split's default separators: space and comma
        {
            std::stringstream words(piece);
            std::string word;
            while (words >> word)
                d.target_regions.emplace_back(word);
        }
    }
    else
    {
        if (!root["target_regions"].isArray())
            throw std::runtime_error("Graph description is missing \"target_regions\" key.");
        for (Json const& r : root["target_regions"].elements())
            d.target_regions.emplace_back(r.asString());
    }
    if (root.isMember("max_reads"))
        d.max_reads = (int64_t)root["max_reads"].asUInt64();
    for (Json const& node : root["nodes"].elements())
        if (node.isMember("sequence"))
            d.longest_alt_insertion = std::max(d.longest_alt_insertion, node["sequence"].asString().size());
    d.graph = std::make_shared<graphtools::Graph>(
        opened_reference ? grm::graphFromJson(root, *opened_reference) : grm::graphFromJson(root, reference_path));
    d.paths = grm::pathsFromJson(d.graph.get(), root["paths"]);
    d.description = std::move(root);
    return d;
}

namespace
{
Json countsToJson(std::map<std::string, CountEntry> const& table)
{
    Json out = Json::object();
    for (auto const& kv : table)
    {
        out[kv.first] = kv.second.count;
        out[kv.first + ":READS"] = kv.second.reads;
        out[kv.first + ":FWD"] = kv.second.fwd;
        out[kv.first + ":REV"] = kv.second.rev;
    }
    return out;
}

// per-fragment support sets, needed for the per-family node / edge breakdown (countPathFamilies with `detailed`,
// ReadCounting.cpp:96-127); the family totals themselves come from the device tables
struct FragmentSupport
{
    uint64_t reads = 0, fwd = 0, rev = 0;
    std::set<std::string> nodes, edges, sequences;
};

void addDetailedCounts(Json& by_sequence, common::ReadBuffer const& reads)
{
    std::vector<std::string> order;
    std::unordered_map<std::string, FragmentSupport> fragments;
    for (auto const& read : reads)
    {
        auto it = fragments.find(read->fragment_id());
        if (it == fragments.end())
        {
            it = fragments.emplace(read->fragment_id(), FragmentSupport()).first;
            order.push_back(read->fragment_id());
        }
        FragmentSupport& f = it->second;
        ++f.reads;
        if (read->graph_mapping_status() == common::Read::MAPPED)
            ++(read->is_graph_reverse_strand() ? f.rev : f.fwd);
        f.nodes.insert(read->graph_nodes_supported().begin(), read->graph_nodes_supported().end());
        f.edges.insert(read->graph_edges_supported().begin(), read->graph_edges_supported().end());
        f.sequences.insert(read->graph_sequences_supported().begin(), read->graph_sequences_supported().end());
    }
    auto bump = [](Json& table, std::string const& key, FragmentSupport const& f) {
        table[key] = table[key].asUInt64() + 1;
        table[key + ":READS"] = table[key + ":READS"].asUInt64() + f.reads;
        table[key + ":FWD"] = table[key + ":FWD"].asUInt64() + f.fwd;
        table[key + ":REV"] = table[key + ":REV"].asUInt64() + f.rev;
    };
    for (auto const& id : order)
    {
        FragmentSupport const& f = fragments[id];
        if (f.sequences.empty())
            continue;
        std::string family;
        for (auto const& s : f.sequences)
            family += (family.empty() ? "" : ",") + s;
        Json& table = by_sequence[family];
        for (auto const& n : f.nodes)
            bump(table, n, f);
        for (auto const& e : f.edges)
            bump(table, e, f);
    }
}
}  // namespace

namespace
{
// the per-family node / edge breakdown from the views of a packed site (same rule as addDetailedCounts)
void addDetailedCounts(Json& by_sequence, graphtools::Graph const& graph, SiteReadViews const& views)
{
    // Support sets are collected by id (a node id / a (from, to) pair); names are made once per distinct element when the
    // tables are written.  Ids stand in for names only as long as names are unambiguous: two edges spelling the same
    // "<from>_<to>" would be ONE element of the original's name sets, so such graphs take the name-keyed path.
    struct Support
    {
        uint64_t reads = 0, fwd = 0, rev = 0;
        LabelSet sequences;
        std::vector<uint32_t> nodes;
        std::vector<uint64_t> edges;  // from << 32 | to
    };
    struct Counter
    {
        uint64_t count = 0, reads = 0, fwd = 0, rev = 0;
    };
    std::vector<uint32_t> order;
    std::unordered_map<uint32_t, Support> fragments;
    bool ambiguous_names = false;
    {
        std::set<std::string> names;
        for (graphtools::NodeId n = 0; n < graph.numNodes() && !ambiguous_names; ++n)
        {
            ambiguous_names = !names.insert(graph.nodeName(n)).second;
            for (graphtools::NodeId s : graph.successors(n))
                if (!names.insert(graph.nodeName(n) + "_" + graph.nodeName(s)).second)
                    ambiguous_names = true;
        }
    }
    for (MappedReadView const& read : views.reads)
    {
        auto it = fragments.find(read.fragment);
        if (it == fragments.end())
        {
            it = fragments.emplace(read.fragment, Support()).first;
            order.push_back(read.fragment);
        }
        Support& f = it->second;
        ++f.reads;
        ++(read.is_graph_reverse_strand ? f.rev : f.fwd);
        f.sequences |= read.sequences;
        uint32_t prev = 0;
        for (uint32_t k = 0; k < read.n_support; ++k)
        {
            const uint32_t entry = views.support[read.support_off + k], node = entry & 0xFFFFu;  // PG_PATH_NODE (include/paragraph_amd.h)
            if ((entry >> 30) & 1u)
                f.nodes.push_back(node);
            if (k > 0 && (entry >> 31))
                f.edges.push_back((uint64_t)prev << 32 | node);
            prev = node;
        }
    }
    auto add = [](Counter& c, Support const& f) {
        ++c.count;
        c.reads += f.reads;
        c.fwd += f.fwd;
        c.rev += f.rev;
    };
    std::map<LabelSet, std::pair<std::map<uint32_t, Counter>, std::map<uint64_t, Counter>>> by_family;  // by label bit set
    std::map<LabelSet, std::map<std::string, Counter>> by_family_named;                                    // ambiguous names
    for (uint32_t id : order)
    {
        Support& f = fragments[id];
        if (!f.sequences.any())
            continue;
        std::sort(f.nodes.begin(), f.nodes.end());
        f.nodes.erase(std::unique(f.nodes.begin(), f.nodes.end()), f.nodes.end());
        std::sort(f.edges.begin(), f.edges.end());
        f.edges.erase(std::unique(f.edges.begin(), f.edges.end()), f.edges.end());
        if (ambiguous_names)
        {
            std::set<std::string> elements;
            for (uint32_t n : f.nodes)
                elements.insert(graph.nodeName(n));
            for (uint64_t e : f.edges)
                elements.insert(graph.nodeName((uint32_t)(e >> 32)) + "_" + graph.nodeName((uint32_t)e));
            for (auto const& name : elements)
                add(by_family_named[f.sequences][name], f);
            continue;
        }
        auto& tables = by_family[f.sequences];
        for (uint32_t n : f.nodes)
            add(tables.first[n], f);
        for (uint64_t e : f.edges)
            add(tables.second[e], f);
    }
    auto family_name = [&](LabelSet const& mask) {
        std::string family;  // label_names are sorted, so this is the sorted join
        for (size_t b = 0; b < views.label_names.size(); ++b)
            if (mask.test(b))
                family += (family.empty() ? "" : ",") + views.label_names[b];
        return family;
    };
    auto write = [](Json& table, std::string const& key, Counter const& c) {
        table[key] = c.count;
        table[key + ":READS"] = c.reads;
        table[key + ":FWD"] = c.fwd;
        table[key + ":REV"] = c.rev;
    };
    for (auto const& fam : by_family)
    {
        Json& table = by_sequence[family_name(fam.first)];
        for (auto const& kv : fam.second.first)
            write(table, graph.nodeName(kv.first), kv.second);
        for (auto const& kv : fam.second.second)
            write(table, graph.nodeName((uint32_t)(kv.first >> 32)) + "_" + graph.nodeName((uint32_t)kv.first), kv.second);
    }
    for (auto const& fam : by_family_named)
    {
        Json& table = by_sequence[family_name(fam.first)];
        for (auto const& kv : fam.second)
            write(table, kv.first, kv.second);
    }
}

// one site's count document; `reads` is null for packed sites (then `views` is the batcher's); `filtered`: the reads the filter
// chain rejected, with the filter's message (object sites under FILTERED_ALIGNMENTS: moved behind `reads`' own, as
// alignAndDisambiguate leaves its read buffer, Disambiguation.cpp:348-358)
Json countDocument(
    Parameters const& parameters, GraphDescription const& d, SiteCounts const& counts, SiteReadViews const& views, size_t reads_in,
    common::ReadBuffer* reads, std::vector<std::pair<common::p_Read, std::string>>* filtered = nullptr)
{
    Json out = parameters.description_in_document ? d.description : Json::object();
    if (parameters.description_in_document)
        out["reference"] = d.reference_path;
    out["fragment_statistics"] = fragmentStatistics(*d.graph, views);
    if (parameters.output_enabled(Parameters::NODE_READ_COUNTS))
        out["read_counts_by_node"] = countsToJson(counts.by_node);
    if (parameters.output_enabled(Parameters::EDGE_READ_COUNTS))
        out["read_counts_by_edge"] = countsToJson(counts.by_edge);
    if (parameters.output_enabled(Parameters::PATH_READ_COUNTS))
    {
        Json families = Json::object();
        for (auto const& kv : counts.by_sequence)
        {
            Json total = Json::object();
            total["total"] = kv.second.count;
            total["total:READS"] = kv.second.reads;
            total["total:FWD"] = kv.second.fwd;
            total["total:REV"] = kv.second.rev;
            families[kv.first] = std::move(total);
        }
        if (parameters.output_enabled(Parameters::DETAILED_READ_COUNTS))
        {
            if (reads)
                addDetailedCounts(families, *reads);
            else
                addDetailedCounts(families, *d.graph, views);
        }
        out["read_counts_by_sequence"] = std::move(families);
    }
    Json stats = alignmentStatistics(*d.graph, views);
    // the filter tallies only exist when filtered alignments are asked for (Disambiguation.cpp:177-199, 333-346)
    const bool tally = parameters.output_enabled(Parameters::FILTERED_ALIGNMENTS);
    stats["bad_alignment_pct"] = (tally && reads_in) ? (double)counts.bad_align / (double)reads_in : 0.0;
    if (tally && counts.bad_align)
        stats["read_filter_bad_align"] = counts.bad_align;
    if (tally && counts.nonuniq)
        stats["read_filter_nonuniq"] = counts.nonuniq;
    out["alignment_statistics"] = std::move(stats);
    if (reads && (parameters.output_enabled(Parameters::ALIGNMENTS) || parameters.output_enabled(Parameters::FILTERED_ALIGNMENTS)))
    {
        // the rejected reads first, as records of the rejected alignment with the filter's message (Disambiguation.cpp:183-203:
        // there they are appended while the aligner's threads run, in whatever order those get to them; here in input order),
        // then the reads that were kept (Disambiguation.cpp:348-356).  One deviation: with a seed stage in front of gssw the
        // original also emits a read the filter rejected after an EARLIER stage and that a later stage then mapped -- twice, once
        // per outcome; here a read has the record of its final outcome only.  "kmer_uncov" comes without the node ids the
        // original appends to it.
        Json alignments = Json::array();
        if (filtered && parameters.output_enabled(Parameters::FILTERED_ALIGNMENTS))
            for (auto const& rf : *filtered)
            {
                Json r = rf.first->toJson();
                r["error"] = rf.second;
                alignments.append(std::move(r));
            }
        if (parameters.output_enabled(Parameters::ALIGNMENTS))
            for (auto const& r : *reads)
                alignments.append(r->toJson());
        out["alignments"] = std::move(alignments);
        if (filtered && parameters.output_enabled(Parameters::FILTERED_ALIGNMENTS))
        {
            common::ReadBuffer all;
            all.reserve(filtered->size() + reads->size());
            for (auto& rf : *filtered)
                all.emplace_back(std::move(rf.first));
            if (parameters.output_enabled(Parameters::ALIGNMENTS))
                for (auto& r : *reads)
                    all.emplace_back(std::move(r));
            reads->swap(all);
            filtered->clear();
        }
    }
    return out;
}

// A site the device path could not take (SiteBatcher::error): its document keeps the graph description, has empty counts and
// says why under "error"; one line on stderr.  The other sites of the run are unaffected.
void noteSiteError(Json& document, GraphDescription const& d, std::string const& error)
{
    if (error.empty())
        return;
    document["error"] = error;
    std::string id = d.description.isMember("ID") ? d.description["ID"].asString() : std::string();
    fprintf(stderr, "[paragraph_amd] WARNING: graph %s skipped (no counts, no genotype): %s\n", id.empty() ? "<no ID>" : id.c_str(), error.c_str());
}

BatchParameters batchParameters(Parameters const& parameters)
{
    if (!parameters.graph_sequence_matching)
        throw std::runtime_error("alignAndDisambiguateBatch: the gssw stage cannot be switched off in the batched workflow");
    BatchParameters bp;
    bp.remove_nonuniq_reads = parameters.remove_nonuniq_reads;
    bp.bad_align_frac = parameters.bad_align_frac;
    bp.kmer_len = parameters.kmer_len;
    bp.path_sequence_matching = parameters.path_sequence_matching;
    bp.exact_match_shortcut = parameters.exact_match_shortcut;
    bp.kmer_sequence_matching = parameters.kmer_sequence_matching;
    bp.klib_sequence_matching = parameters.klib_sequence_matching;
    bp.threads = parameters.threads;
    bp.device = parameters.device;
    bp.validate_alignments = parameters.validate_alignments;
    bp.keep_filtered = parameters.output_enabled(Parameters::FILTERED_ALIGNMENTS);  // (object sites only: packed ones keep no records)
    bp.node_counts = parameters.output_enabled(Parameters::NODE_READ_COUNTS);
    bp.sequence_counts = parameters.output_enabled(Parameters::PATH_READ_COUNTS);
    return bp;
}
}  // namespace

std::vector<Json> alignAndDisambiguateBatch(Parameters const& parameters, std::vector<SiteInput> const& sites)
{
    const BatchParameters bp = batchParameters(parameters);
    SiteBatcher batcher;
    std::vector<size_t> reads_in(sites.size());
    for (size_t s = 0; s < sites.size(); ++s)
    {
        if (!sites[s].description || !sites[s].reads)
            throw std::runtime_error("alignAndDisambiguateBatch: site without description or reads");
        reads_in[s] = sites[s].reads->size();
        batcher.addSite(sites[s].description->graph.get(), sites[s].reads, &sites[s].description->paths);
    }
    const double t_batch = now();
    if (!sites.empty())
        batcher.run(bp);
    const double t_documents = now();
    std::vector<Json> documents(sites.size());
    parallelFor(sites.size(), parameters.threads, [&](size_t s) {
        GraphDescription const& d = *sites[s].description;
        std::vector<common::Read const*> view;
        view.reserve(sites[s].reads->size());
        for (auto const& r : *sites[s].reads)
            view.push_back(r.get());
        documents[s] = countDocument(parameters, d, batcher.counts(s), viewsOfReads(*d.graph, view), reads_in[s], sites[s].reads, &batcher.filtered(s));
        noteSiteError(documents[s], d, batcher.error(s));
    });
    if (parameters.timings)
    {
        parameters.timings->device_batch += t_documents - t_batch;
        parameters.timings->documents += now() - t_documents;
        parameters.timings->sites += sites.size();
        for (size_t n : reads_in)
            parameters.timings->reads += n;
    }
    return documents;
}

std::vector<Json> alignAndDisambiguateBatch(Parameters const& parameters, std::vector<PackedSiteInput> const& sites)
{
    if (parameters.output_enabled(Parameters::ALIGNMENTS))
        throw std::runtime_error("alignAndDisambiguateBatch: packed sites keep no per-read records; use the object form for \"alignments\"");
    const BatchParameters bp = batchParameters(parameters);
    SiteBatcher batcher;
    for (size_t s = 0; s < sites.size(); ++s)
    {
        if (!sites[s].description || !sites[s].reads)
            throw std::runtime_error("alignAndDisambiguateBatch: site without description or reads");
        batcher.addSite(sites[s].description->graph.get(), sites[s].reads, &sites[s].description->paths);
    }
    const double t_batch = now();
    if (!sites.empty())
        batcher.run(bp);
    const double t_documents = now();
    std::vector<Json> documents(sites.size());
    parallelFor(sites.size(), parameters.threads, [&](size_t s) {
        documents[s] = countDocument(parameters, *sites[s].description, batcher.counts(s), batcher.views(s), sites[s].reads->size(), nullptr);
        noteSiteError(documents[s], *sites[s].description, batcher.error(s));
    });
    if (parameters.timings)
    {
        parameters.timings->device_batch += t_documents - t_batch;
        parameters.timings->documents += now() - t_documents;
        parameters.timings->sites += sites.size();
        for (auto const& site : sites)
            parameters.timings->reads += site.reads->size();
    }
    return documents;
}

// alignAndDisambiguateBatch (packed form) in two halves: beginPackedBatch packs, uploads and queues the batch on the device and
// returns; finishPackedBatch waits for it and writes the documents.  Between the two the caller does other work (a lane of
// grmpy::genotypeGraphs prepares its next chunk).
struct PackedBatchInFlight
{
    Parameters parameters;
    std::vector<PackedSiteInput> sites;
    SiteBatcher batcher;
    bool submitted = false;
    double begin_s = 0;
};

std::unique_ptr<PackedBatchInFlight> beginPackedBatch(Parameters const& parameters, std::vector<PackedSiteInput> sites)
{
    if (parameters.output_enabled(Parameters::ALIGNMENTS))
        throw std::runtime_error("alignAndDisambiguateBatch: packed sites keep no per-read records; use the object form for \"alignments\"");
    std::unique_ptr<PackedBatchInFlight> f(new PackedBatchInFlight);
    f->parameters = parameters;
    f->sites = std::move(sites);
    for (auto const& site : f->sites)
    {
        if (!site.description || !site.reads)
            throw std::runtime_error("alignAndDisambiguateBatch: site without description or reads");
        f->batcher.addSite(site.description->graph.get(), site.reads, &site.description->paths);
    }
    const double t0 = now();
    f->submitted = f->sites.empty() || f->batcher.submit(batchParameters(parameters));
    f->begin_s = now() - t0;
    return f;
}

std::vector<Json> finishPackedBatch(PackedBatchInFlight& f)
{
    const double t_batch = now();
    if (!f.sites.empty())
    {
        if (f.submitted)
            f.batcher.collect();
        else
            f.batcher.run(batchParameters(f.parameters));  // a site outside the envelope: run() isolates it
    }
    const double t_documents = now();
    std::vector<Json> documents(f.sites.size());
    parallelFor(f.sites.size(), f.parameters.threads, [&](size_t s) {
        documents[s] = countDocument(f.parameters, *f.sites[s].description, f.batcher.counts(s), f.batcher.views(s), f.sites[s].reads->size(), nullptr);
        noteSiteError(documents[s], *f.sites[s].description, f.batcher.error(s));
    });
    if (f.parameters.timings)
    {
        f.parameters.timings->device_batch += f.begin_s + (t_documents - t_batch);
        f.parameters.timings->documents += now() - t_documents;
        f.parameters.timings->sites += f.sites.size();
        for (auto const& site : f.sites)
            f.parameters.timings->reads += site.reads->size();
    }
    return documents;
}

std::vector<Json> countGraphs(
    Parameters const& parameters, std::vector<std::string> const& graph_paths, std::string const& reference_path,
    std::vector<std::string> const& bam_paths, std::vector<std::string> const& bam_index_paths, std::string const& target_regions,
    size_t sites_per_batch)
{
    if (bam_paths.empty())
        throw std::runtime_error("ERROR: BAM file is missing.");
    if (!bam_index_paths.empty() && bam_index_paths.size() != bam_paths.size())
        throw std::runtime_error("ERROR: the number of BAM index files differs from the number of BAM files");
    auto index_of = [&](size_t b) { return bam_index_paths.empty() ? std::string() : bam_index_paths[b]; };
    const bool packed = bam_paths.size() == 1 && !parameters.output_enabled(Parameters::ALIGNMENTS)
        && !parameters.output_enabled(Parameters::FILTERED_ALIGNMENTS) && !parameters.validate_alignments;
    const common::FastaFile fasta(reference_path);
    std::vector<std::unique_ptr<common::BamReader>> keep_alive;  // also shares the parsed header / index with the workers
    for (size_t b = 0; b < bam_paths.size(); ++b)
        keep_alive.emplace_back(new common::BamReader(bam_paths[b], index_of(b), reference_path));
    Json bam_value = bam_paths.size() == 1 ? Json(bam_paths[0]) : Json::array();
    if (bam_paths.size() != 1)
        for (auto const& b : bam_paths)
            bam_value.append(b);

    std::vector<Json> documents(graph_paths.size());
    if (sites_per_batch == 0)
        sites_per_batch = 192;
    // one chunk of graphs: load + extract, ONE device batch, documents -- with `lane.threads` workers
    auto processChunk = [&](size_t g0, Parameters const& lane) {
        const size_t n_here = std::min(sites_per_batch, graph_paths.size() - g0);
        std::vector<GraphDescription> graphs(n_here);
        std::vector<PackedSite> packed_reads(packed ? n_here : 0);
        std::vector<common::ReadBuffer> object_reads(packed ? 0 : n_here);
        const size_t kRun = 8;  // neighbouring graphs per grab, see prepareChunk
        parallelFor((n_here + kRun - 1) / kRun, lane.threads, [&](size_t run) {
            std::vector<std::unique_ptr<common::BamReader>> readers;
            for (size_t g = run * kRun; g < std::min(n_here, (run + 1) * kRun); ++g)
            {
                graphs[g] = GraphDescription::load(graph_paths[g0 + g], reference_path, target_regions, &fasta);
                const int max_reads = graphs[g].max_reads >= 0 ? (int)graphs[g].max_reads : parameters.max_reads;
                for (size_t b = 0; b < bam_paths.size(); ++b)
                {
                    if (readers.size() <= b)
                        readers.emplace_back(new common::BamReader(bam_paths[b], index_of(b), reference_path));
                    if (packed)
                        extractPacked(*readers[b], graphs[g].target_regions, max_reads, (unsigned)graphs[g].longest_alt_insertion, packed_reads[g]);
                    else
                        common::extractReads(
                            *readers[b], graphs[g].target_regions, max_reads, (unsigned)graphs[g].longest_alt_insertion, object_reads[g]);
                }
            }
        });
        std::vector<Json> batch;
        if (packed)
        {
            std::vector<PackedSiteInput> sites(n_here);
            for (size_t g = 0; g < n_here; ++g)
            {
                sites[g].description = &graphs[g];
                sites[g].reads = &packed_reads[g];
            }
            batch = alignAndDisambiguateBatch(lane, sites);
        }
        else
        {
            std::vector<SiteInput> sites(n_here);
            for (size_t g = 0; g < n_here; ++g)
            {
                sites[g].description = &graphs[g];
                sites[g].reads = &object_reads[g];
            }
            batch = alignAndDisambiguateBatch(lane, sites);
        }
        for (size_t g = 0; g < n_here; ++g)
        {
            batch[g]["bam"] = bam_value;
            documents[g0 + g] = std::move(batch[g]);
        }
    };
    // Several chunks: lanes as in grmpy::genotypeGraphs -- one chunk is on the device while others extract reads or write
    // documents (one host thread per lane, 1.5 lanes per thread, see grmpy::genotypeGraphs).
    // With several devices (paragraph::setDevices / PG_DEVICES) lane l works on device l % devices, at least one lane each.
    const size_t n_chunks = (graph_paths.size() + sites_per_batch - 1) / sites_per_batch;
    const size_t n_devices = paragraph::deviceCount();
    const size_t lanes_by_threads = (size_t)std::min<size_t>(32 * n_devices, (size_t)std::max(1, parameters.threads + parameters.threads / 2));
    const size_t lanes = std::max<size_t>(1, std::min<size_t>(n_chunks, std::max(lanes_by_threads, n_devices)));
    if (lanes == 1)
    {
        for (size_t c = 0; c < n_chunks; ++c)
            processChunk(c * sites_per_batch, parameters);
        return documents;
    }
    std::atomic<size_t> next_chunk(0);
    std::atomic<bool> failed(false);
    std::exception_ptr failure;
    std::mutex timings_mutex;
    std::atomic<int> lane_ids(0);
    auto lane_body = [&] {
        Timings mine;
        Parameters lane = parameters;
        lane.threads = std::max(1, parameters.threads / (int)lanes);
        lane.timings = parameters.timings ? &mine : nullptr;
        lane.device = (int)((size_t)lane_ids.fetch_add(1) % n_devices);
        try
        {
            for (size_t c = next_chunk.fetch_add(1); c < n_chunks && !failed.load(); c = next_chunk.fetch_add(1))
                processChunk(c * sites_per_batch, lane);
        }
        catch (...)
        {
            if (!failed.exchange(true))
                failure = std::current_exception();
        }
        if (parameters.timings)
        {
            std::lock_guard<std::mutex> lock(timings_mutex);
            parameters.timings->device_batch += mine.device_batch;
            parameters.timings->documents += mine.documents;
            parameters.timings->sites += mine.sites;
            parameters.timings->reads += mine.reads;
        }
    };
    std::vector<std::thread> pool;
    for (size_t l = 1; l < lanes; ++l)
        pool.emplace_back(lane_body);
    lane_body();
    for (auto& t : pool)
        t.join();
    if (failure)
        std::rethrow_exception(failure);
    return documents;
}

Json alignAndDisambiguate(Parameters const& parameters, GraphDescription const& description, common::ReadBuffer& all_reads)
{
    std::vector<SiteInput> one(1);
    one[0].description = &description;
    one[0].reads = &all_reads;
    return alignAndDisambiguateBatch(parameters, one)[0];
}
}  // namespace paragraph

namespace grmpy
{
namespace
{
// -A given and the folder is there (AlignSamples.cpp:120-121: a missing folder silently means "no")
bool writesAlignments(Parameters const& p)
{
    struct stat st;
    return !p.alignment_output_folder.empty() && stat(p.alignment_output_folder.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

paragraph::Parameters siteParameters(Parameters const& p)
{
    paragraph::Parameters sp;
    sp.max_reads = p.max_reads;
    sp.bad_align_frac = p.bad_align_frac;
    sp.path_sequence_matching = p.path_sequence_matching;
    sp.exact_match_shortcut = p.exact_match_shortcut;
    sp.graph_sequence_matching = p.graph_sequence_matching;
    sp.kmer_sequence_matching = p.kmer_sequence_matching;
    sp.klib_sequence_matching = p.klib_sequence_matching;
    sp.kmer_len = p.bad_align_uniq_kmer_len;
    sp.threads = p.threads;
    sp.timings = p.timings;
    sp.output_options_ = paragraph::Parameters::NODE_READ_COUNTS | paragraph::Parameters::EDGE_READ_COUNTS
        | paragraph::Parameters::PATH_READ_COUNTS | paragraph::Parameters::DETAILED_READ_COUNTS;
    if (p.output_alignments)
        sp.output_options_ |= paragraph::Parameters::ALIGNMENTS;
    if (writesAlignments(p))
        sp.output_options_ |= paragraph::Parameters::ALIGNMENTS | paragraph::Parameters::FILTERED_ALIGNMENTS;
    return sp;
}

// writeAlignments (AlignSamples.cpp:57-109): the sample's whole count document, gzip-compressed, named after sample, graph and
// target regions with everything outside [A-Za-z0-9.-] turned into '_'
void writeAlignmentFile(
    Parameters const& parameters, Json& doc, paragraph::GraphDescription const& d, std::string const& reference_path,
    genotyping::SampleInfo const& sample)
{
    doc["sample"] = sample.sample_name();
    doc["reference"] = reference_path;
    auto safe = [](std::string text) {
        for (char& c : text)
            if (!(std::isalnum((unsigned char)c) || c == '.' || c == '-'))
                c = '_';
        return text;
    };
    std::string regions;
    for (common::Region const& r : d.target_regions)
        regions += (regions.empty() ? "" : "_") + std::string(r);
    std::string graph_id = "00000000-0000-0000-0000-000000000000";  // (the original: a default-constructed boost uuid)
    if (d.description.isMember("ID"))
        graph_id = d.description["ID"].asString();
    else if (d.description.isMember("model_name"))
        graph_id = d.description["model_name"].asString();
    const std::string path
        = parameters.alignment_output_folder + "/" + safe(sample.sample_name()) + "-" + safe(graph_id) + "-" + safe(regions) + ".json.gz";
    const std::string text = doc.dump(1) + "\n";
    gzFile f = gzopen(path.c_str(), "wb");
    if (!f)
        throw std::runtime_error("ERROR: Failed to open output file '" + path + "'. Error: '" + std::strerror(errno) + "'");
    const int n = gzwrite(f, text.data(), (unsigned)text.size());
    const int rc = gzclose(f);
    if (n != (int)text.size() || rc != Z_OK)
        throw std::runtime_error("ERROR: Failed to write output file '" + path + "'");
}

// what alignSingleSample keeps of a count document (AlignSamples.cpp:160-171)
void finishSampleDocument(Json& doc, std::string const& bam, bool keep_alignments)
{
    doc["bam"] = bam;
    if (!keep_alignments)
        doc.removeMember("alignments");
}

Json genotypeToJson(genotyping::Genotype const& g, std::vector<std::string> const& allele_names)
{
    Json out = Json::object();
    out["GT"] = g.toString(&allele_names);
    if (!g.gl.empty())
    {
        Json gl = Json::object();
        for (size_t i = 0; i < g.gl.size(); ++i)
        {
            std::string name;
            for (size_t j = 0; j < g.gl_name[i].size(); ++j)
                name += (j ? "/" : "") + allele_names[g.gl_name[i][j]];
            gl[name] = g.gl[i];
        }
        out["GL"] = std::move(gl);
    }
    if (g.gq != -1)
        out["GQ"] = g.gq;
    if (!g.allele_fractions.empty())
    {
        Json fractions = Json::object();
        for (size_t a = 0; a < g.allele_fractions.size() && a < allele_names.size(); ++a)
            fractions[allele_names[a]] = g.allele_fractions[a];
        out["allele_fractions"] = std::move(fractions);
    }
    if (!g.filters.empty())
    {
        Json filters = Json::array();
        for (auto const& f : g.filters)
            filters.append(f);
        out["filters"] = std::move(filters);
    }
    if (!g.gt.empty())
    {
        out["num_reads"] = g.num_reads;
        if (g.coverage_test_pvalue != -1)
            out["coverage_test_pvalue"] = g.coverage_test_pvalue;
    }
    return out;
}

std::map<std::string, int32_t> edgeCounts(Json const& doc)
{
    if (!doc.isMember("read_counts_by_edge"))
        throw std::runtime_error("Cannot find key read_counts_by_edge in JSON");
    std::map<std::string, int32_t> out;
    for (auto const& kv : doc["read_counts_by_edge"].members())
        if (kv.second.isNumber())
            out[kv.first] = (int32_t)kv.second.asInt64();
    return out;
}

// GraphGenotyper::addAlignment + getGenotypes (lib/genotyping/GraphGenotyper.cpp:88-337)
Json genotypeDocument(
    graphtools::Graph const& graph, Json const& root, std::string const& genotyping_parameter_path,
    std::vector<genotyping::SampleInfo const*> const& samples, std::vector<Json const*> const& documents,
    std::vector<Json*> const* consumed = nullptr /* the same documents, when the caller drops them afterwards: their
                                                     alignment_statistics are moved into the result instead of copied */)
{
    std::vector<std::string> regions;
    for (Json const& r : root["target_regions"].elements())
        regions.push_back(r.asString());
    const auto ploidies = genotyping::GraphBreakpointGenotyper::ploidiesForTargetRegions(regions);
    genotyping::GraphBreakpointGenotyper genotyper(ploidies.first, ploidies.second);
    genotyper.reset(&graph);
    if (!genotyping_parameter_path.empty())
        genotyper.parameters().setFromJson(Json::parseFile(genotyping_parameter_path));

    Json result = Json::object();
    genotyping::BreakpointMap const& breakpoints = genotyper.breakpointsOfGraph();
    for (size_t i = 0; i < samples.size(); ++i)
    {
        genotyping::SampleInfo const& sample = *samples[i];
        Json const& counted = *documents[i];
        // a count document normally repeats the graph description (the reference's do); the batched workflow's do not, the
        // description then comes from the loaded graph itself
        Json const& doc = counted.isMember("nodes") ? counted : root;
        genotyper.addSample(
            sample.sample_name(), edgeCounts(counted), sample.autosome_depth(), (int)sample.read_length(), sample.depth_sd(), sample.sex());
        if (doc.isMember("eventinfo"))
        {
            if (result.isMember("eventinfo") && result["eventinfo"] != doc["eventinfo"])
                throw std::runtime_error("Samples disagree on eventinfo");
            result["eventinfo"] = doc["eventinfo"];
        }
        if (!result.isMember("graphinfo"))
        {
            Json info = Json::object();
            if (doc.isMember("ID"))
                info["ID"] = doc["ID"];
            else if (doc.isMember("vcf_records"))
            {
                std::string ids;
                for (Json const& rec : doc["vcf_records"].elements())
                    if (rec.isMember("id"))
                        ids += (ids.empty() ? "" : ",") + rec["id"].asString();
                info["ID"] = ids;
            }
            Json bp_info = Json::array();
            for (auto const& bp : breakpoints)
            {
                Json entry = Json::object();
                entry["name"] = bp.first;
                entry["mapped_alleles"] = Json::object();
                for (auto const& allele : bp.second.allAlleleNames())
                {
                    auto const& canonical = bp.second.getCanonicalAlleleName(allele);
                    if (canonical != allele)
                        entry["mapped_alleles"][allele] = canonical;
                }
                bp_info.append(std::move(entry));
            }
            result["breakpointinfo"] = std::move(bp_info);
            info["target_regions"] = doc["target_regions"];
            info["sequencenames"] = doc["sequencenames"];
            info["nodes"] = Json::array();
            for (Json const& n : doc["nodes"].elements())
            {
                Json node = Json::object();
                node["name"] = n["name"];
                if (n.isMember("sequences"))
                    node["sequences"] = n["sequences"];
                info["nodes"].append(std::move(node));
            }
            info["edges"] = Json::array();
            for (Json const& e : doc["edges"].elements())
            {
                Json edge = Json::object();
                edge["name"] = e["from"].asString() + "_" + e["to"].asString();
                if (e.isMember("sequences"))
                    edge["sequences"] = e["sequences"];
                info["edges"].append(std::move(edge));
            }
            result["graphinfo"] = std::move(info);
        }
        Json per_sample = consumed ? std::move((*(*consumed)[i])["alignment_statistics"]) : Json(counted["alignment_statistics"]);
        for (auto const& kv : counted["fragment_statistics"].members())
            if (kv.first != "linear_histogram" && kv.first != "graph_histogram")
                per_sample[kv.first] = kv.second;
        result["samples"][sample.sample_name()] = std::move(per_sample);
    }
    genotyper.runGenotyping();
    auto const& allele_names = genotyper.alleleNames();
    std::map<std::string, genotyping::GenotypeSet> by_breakpoint;  // "" = the combined genotype
    for (size_t i = 0; i < samples.size(); ++i)
    {
        const std::string name = samples[i]->sample_name();
        Json& entry = result["samples"][name];
        entry["breakpoints"] = Json::object();
        by_breakpoint[""].add(allele_names, genotyper.getGenotype(name, ""));
        for (auto const& bp : breakpoints)
        {
            by_breakpoint[bp.first].add(allele_names, genotyper.getGenotype(name, bp.first));
            Json bj = Json::object();
            bj["gt"] = genotypeToJson(genotyper.getGenotype(name, bp.first), allele_names);
            Json counts = Json::object();
            counts["edges"] = Json::object();
            counts["alleles"] = Json::object();
            for (auto const& edge : bp.second.edgeNames())
                counts["edges"][edge] = genotyper.getCount(i, bp.first, edge);
            for (auto const& allele : bp.second.canonicalAlleleNames())
                counts["alleles"][allele] = genotyper.getCount(i, bp.first, allele);
            bj["counts"] = std::move(counts);
            entry["breakpoints"][bp.first] = std::move(bj);
        }
        entry["gt"] = genotypeToJson(genotyper.getGenotype(name, ""), allele_names);
    }
    if (samples.size() > 1)
    {
        Json pop = genotyping::PopulationStatistics(by_breakpoint[""]).toJson();
        for (auto const& kv : by_breakpoint)
            if (!kv.first.empty())
                pop["breakpoints"][kv.first] = genotyping::PopulationStatistics(kv.second).toJson();
        result["population"] = std::move(pop);
    }
    return result;
}
}  // namespace

void alignSingleSample(
    Parameters const& parameters, std::string const& graph_path, std::string const& reference_path, common::ReadReader& reader,
    genotyping::SampleInfo& sample)
{
    const paragraph::GraphDescription d = paragraph::GraphDescription::load(graph_path, reference_path);
    common::ReadBuffer reads;
    common::extractReads(reader, d.target_regions, parameters.max_reads, (unsigned)d.longest_alt_insertion, reads);
    Json doc = paragraph::alignAndDisambiguate(siteParameters(parameters), d, reads);
    doc["bam"] = sample.filename();
    if (writesAlignments(parameters))
        writeAlignmentFile(parameters, doc, d, reference_path, sample);
    finishSampleDocument(doc, sample.filename(), parameters.output_alignments);
    sample.set_alignment_data(doc);
}

Json countAndGenotype(
    std::string const& graph_path, std::string const& reference_path, std::string const& genotyping_parameter_path,
    genotyping::Samples const& samples)
{
    if (samples.empty())
        throw std::runtime_error("countAndGenotype: no samples");
    const Json root = graph_path.empty() ? samples.front().get_alignment_data() : Json::parseFile(graph_path);
    const graphtools::Graph graph = grm::graphFromJson(root, reference_path, true);
    std::vector<genotyping::SampleInfo const*> sample_ptrs;
    std::vector<Json const*> docs;
    for (auto const& s : samples)
    {
        sample_ptrs.push_back(&s);
        docs.push_back(&s.get_alignment_data());
    }
    Json const& flat = root.isMember("graph") ? root["graph"] : root;
    return genotypeDocument(graph, flat, genotyping_parameter_path, sample_ptrs, docs);
}

namespace
{
// graphs [g0, g1) x all samples, loaded and with their reads extracted: the unit that moves through the pipeline
struct Chunk
{
    size_t g0 = 0, g1 = 0;
    std::vector<paragraph::GraphDescription> graphs;
    std::vector<common::ReadBuffer> reads;       // [(g - g0) * n_samples + s], object form
    std::vector<paragraph::PackedSite> packed;   // same indexing, packed form (one of the two is filled)
    double load_s = 0, extract_s = 0;
};

std::unique_ptr<Chunk> prepareChunk(
    Parameters const& parameters, std::vector<std::string> const& graph_paths, std::string const& reference_path,
    genotyping::Samples const& samples, size_t g0, size_t g1, int threads, common::FastaFile const* fasta)
{
    std::unique_ptr<Chunk> chunk(new Chunk);
    chunk->g0 = g0;
    chunk->g1 = g1;
    const size_t n_graphs = g1 - g0, n_samples = samples.size();
    chunk->graphs.resize(n_graphs);
    const double t_load = now();
    parallelFor(n_graphs, threads, [&](size_t g) { chunk->graphs[g] = paragraph::GraphDescription::load(graph_paths[g0 + g], reference_path, "", fasta); });
    const double t_extract = now();
    chunk->load_s = t_extract - t_load;

    // tasks in sample-major order so a worker mostly stays on one BAM; every worker owns its readers
    const bool packed = parameters.packed_reads && !parameters.output_alignments && !writesAlignments(parameters);
    if (packed)
        chunk->packed.resize(n_graphs * n_samples);
    else
        chunk->reads.resize(n_graphs * n_samples);
    const size_t n_tasks = n_graphs * n_samples;
    const size_t workers = std::max<size_t>(1, std::min<size_t>((size_t)std::max(threads, 1), n_tasks));
    std::atomic<size_t> next(0);
    std::exception_ptr failure;
    std::atomic<bool> failed(false);
    auto work = [&] {
        std::map<size_t, std::unique_ptr<common::BamReader>> readers;
        try
        {
            // a worker takes runs of neighbouring graphs: sites that are close on the genome share BGZF blocks, which the
            // reader keeps inflated
            const size_t kRun = std::max<size_t>(8, std::min<size_t>(32, n_tasks / (workers * 4)));
            for (;;)
            {
                const size_t first = next.fetch_add(kRun);
                if (first >= n_tasks || failed.load())
                    return;
                for (size_t task = first; task < std::min(n_tasks, first + kRun); ++task)
                {
                    const size_t s = task / n_graphs, g = task % n_graphs;
                    auto& reader = readers[s];
                    if (!reader)
                        reader.reset(new common::BamReader(samples[s].filename(), samples[s].index_filename(), reference_path));
                    paragraph::GraphDescription const& d = chunk->graphs[g];
                    const int max_reads = d.max_reads >= 0 ? (int)d.max_reads : parameters.max_reads;
                    if (packed)
                        paragraph::extractPacked(*reader, d.target_regions, max_reads, (unsigned)d.longest_alt_insertion, chunk->packed[g * n_samples + s]);
                    else
                        common::extractReads(*reader, d.target_regions, max_reads, (unsigned)d.longest_alt_insertion, chunk->reads[g * n_samples + s]);
                }
            }
        }
        catch (...)
        {
            if (!failed.exchange(true))
                failure = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    for (size_t w = 1; w < workers; ++w)
        pool.emplace_back(work);
    work();
    for (auto& t : pool)
        t.join();
    if (failure)
        std::rethrow_exception(failure);
    chunk->extract_s = now() - t_extract;
    return chunk;
}
}  // namespace

std::vector<Json> genotypeGraphs(
    Parameters const& parameters, std::vector<std::string> const& graph_paths, std::string const& reference_path,
    genotyping::Samples const& samples, std::string const& genotyping_parameter_path)
{
    const size_t n_graphs = graph_paths.size(), n_samples = samples.size();
    std::vector<Json> genotypes(n_graphs);
    if (parameters.genotype_text)
        parameters.genotype_text->assign(n_graphs, std::string());
    if (n_graphs == 0 || n_samples == 0)
        return genotypes;
    const size_t pairs_per_batch = parameters.sites_per_batch ? parameters.sites_per_batch : 192;
    const size_t per_batch = std::max<size_t>(1, pairs_per_batch / n_samples);
    const size_t n_even_chunks = (n_graphs + per_batch - 1) / per_batch;
    // Lanes: each lane takes the next chunk and carries it through every stage (load + extract, device batch, documents,
    // genotypes) with its share of the host threads.  The device part of SiteBatcher::run() is serialised by the device
    // mutex; everything else of different chunks overlaps -- one lane extracts while another is on the device and a third
    // writes documents.
    // Several devices (parameters.devices, else paragraph::setDevices / PG_DEVICES): lane l works on device l % devices --
    // chunks of sites are independent, so the devices never exchange anything (the reference's thread-per-(sample, graph)
    // parallelism, grmpy/Workflow.cpp:225-231, with a GPU behind every lane); at least one lane per device.
    if (!parameters.devices.empty())
        paragraph::setDevices(parameters.devices);
    const size_t n_devices = paragraph::deviceCount();
    // One host thread per lane, and half as many lanes again as threads: a lane that does its own extraction, documents and
    // releases stays inside one allocator arena and starts no helper threads, and it sleeps while its batch is on the device
    // (pg_device_prefer_blocking_waits), about a quarter of its time.  Measured on the 16 CPUs the GPU box allows, 10 000
    // sites, batches of 128: 16 lanes 57 k sites/s, 24 lanes 62 k, 32 lanes 60 k; the earlier 8 lanes x 4 threads: 31 k.
    const int lanes_default
        = std::max((int)n_devices, std::min(32 * (int)n_devices, std::max(1, parameters.threads + parameters.threads / 2)));
    const int lanes_wanted = parameters.lanes > 0 ? parameters.lanes : lanes_default;
    const size_t lanes = std::max<size_t>(1, std::min<size_t>((size_t)lanes_wanted, n_even_chunks));
    const std::vector<std::pair<size_t, size_t>> chunk_ranges = chunkSchedule(n_graphs, per_batch, lanes);
    const size_t n_chunks = chunk_ranges.size();
    const int lane_threads = std::max(1, parameters.threads / (int)lanes);
    paragraph::Timings lane_timings_total;
    std::mutex timings_mutex;
    const common::FastaFile fasta(reference_path);  // one handle for all graph loads (positional reads)
    // one reader per sample kept open for the whole call: fails early on unreadable inputs and keeps the parsed header /
    // index of every BAM alive, so the workers' own readers share it instead of parsing it again per chunk
    std::vector<std::unique_ptr<common::BamReader>> keep_alive;
    for (auto const& sample : samples)
        keep_alive.emplace_back(new common::BamReader(sample.filename(), sample.index_filename(), reference_path));

    std::atomic<size_t> next_chunk(0);
    std::exception_ptr failure;
    std::atomic<bool> failed(false);
    // PG_WORKFLOW_TRACE=<file>: one line per (lane, chunk, phase) with start / end seconds since the call began
    const char* trace_path = std::getenv("PG_WORKFLOW_TRACE");
    const double t_call = now();
    std::vector<std::string> trace;
    std::atomic<int> lane_ids(0);
    auto lane = [&] {
        const int lane_id = lane_ids.fetch_add(1);
        std::vector<std::string> my_trace;
        double t_mark = now();
        auto phase = [&](size_t chunk_index, const char* what) {
            if (!trace_path)
                return;
            const double t = now();
            char line[160];
            snprintf(line, sizeof line, "%d\t%zu\t%s\t%.6f\t%.6f", lane_id, chunk_index, what, t_mark - t_call, t - t_call);
            my_trace.push_back(line);
            t_mark = t;
        };
        paragraph::Timings mine;
        paragraph::Parameters site_parameters = siteParameters(parameters);
        const bool write_alignments = writesAlignments(parameters);
        if (!parameters.output_alignments && !write_alignments)
        {
            // the count documents of this workflow are read by the genotyper and dropped: only the table it reads, no copy of
            // the description (node / sequence tables and the per-family breakdown are what `paragraph` writes, not grmpy)
            site_parameters.output_options_ = paragraph::Parameters::EDGE_READ_COUNTS;
            site_parameters.description_in_document = false;
        }
        site_parameters.threads = lane_threads;
        site_parameters.timings = parameters.timings ? &mine : nullptr;
        site_parameters.device = (int)((size_t)lane_id % n_devices);
        try
        {
            // A lane keeps TWO chunks going: while one is on the device it prepares the next (graph loading + read extraction),
            // queues it, and only then waits for the first one's records and writes its documents and genotypes.  The device
            // always has this lane's next batch behind the current one, and the lane's thread is busy instead of asleep --
            // as many lanes as host threads are then enough (no oversubscribed CPUs, no descheduled lock holders).
            struct InFlight
            {
                size_t c = 0, g0 = 0, n_here = 0;
                std::unique_ptr<Chunk> chunk;
                std::unique_ptr<paragraph::PackedBatchInFlight> batch;
            };
            auto finish = [&](InFlight& f) {
                if (!f.chunk)
                    return;
                std::vector<Json> documents;
                if (f.batch)
                    documents = paragraph::finishPackedBatch(*f.batch);
                else
                {
                    std::vector<paragraph::SiteInput> sites(f.n_here * n_samples);
                    for (size_t i = 0; i < sites.size(); ++i)
                    {
                        sites[i].description = &f.chunk->graphs[i / n_samples];
                        sites[i].reads = &f.chunk->reads[i];
                    }
                    documents = paragraph::alignAndDisambiguateBatch(site_parameters, sites);
                }
                for (size_t i = 0; i < documents.size(); ++i)
                {
                    if (write_alignments)
                    {
                        documents[i]["bam"] = samples[i % n_samples].filename();
                        writeAlignmentFile(parameters, documents[i], f.chunk->graphs[i / n_samples], reference_path, samples[i % n_samples]);
                    }
                    finishSampleDocument(documents[i], samples[i % n_samples].filename(), parameters.output_alignments);
                }
                phase(f.c, "batch+documents");

                const double t_genotype = now();
                const size_t g0 = f.g0;
                Chunk* chunk = f.chunk.get();
                parallelFor(f.n_here, lane_threads, [&](size_t g) {
                    std::vector<genotyping::SampleInfo const*> sample_ptrs;
                    std::vector<Json const*> docs;
                    std::vector<Json*> dropped_after;  // the count documents end with this chunk
                    for (size_t s = 0; s < n_samples; ++s)
                    {
                        sample_ptrs.push_back(&samples[s]);
                        docs.push_back(&documents[g * n_samples + s]);
                        dropped_after.push_back(&documents[g * n_samples + s]);
                    }
                    genotypes[g0 + g] = genotypeDocument(
                        *chunk->graphs[g].graph, chunk->graphs[g].description, genotyping_parameter_path, sample_ptrs, docs, &dropped_after);
                    for (Json const* doc : docs)  // a graph the device path could not take: genotyped from no counts, and says so
                        if (doc->isMember("error") && !genotypes[g0 + g].isMember("error"))
                            genotypes[g0 + g]["error"] = (*doc)["error"];
                    if (parameters.genotype_text)
                    {
                        (*parameters.genotype_text)[g0 + g] = genotypes[g0 + g].dump(parameters.genotype_text_indent);
                        genotypes[g0 + g] = Json();
                    }
                });
                if (parameters.genotype_text && parameters.genotype_text_ready)
                    parameters.genotype_text_ready(g0, g0 + f.n_here);
                phase(f.c, "genotypes");
                const double t_release = now();
                // the lane frees what it allocated itself (a helper thread doing it met the lanes in the allocator's arena locks)
                mine.load_graphs += f.chunk->load_s;
                mine.extract_reads += f.chunk->extract_s;
                std::vector<Json>().swap(documents);
                f.batch.reset();
                f.chunk.reset();
                phase(f.c, "release");
                mine.genotypes += t_release - t_genotype;
                mine.release += now() - t_release;
                mine.batches += 1;
            };
            InFlight current;
            for (;;)
            {
                const size_t c = next_chunk.fetch_add(1);
                if (c >= n_chunks || failed.load())
                    break;
                InFlight next;
                next.c = c;
                next.g0 = chunk_ranges[c].first;
                next.n_here = chunk_ranges[c].second - next.g0;
                t_mark = now();
                next.chunk = prepareChunk(parameters, graph_paths, reference_path, samples, next.g0, chunk_ranges[c].second, lane_threads, &fasta);
                phase(c, "prepare");
                if (!next.chunk->packed.empty())
                {
                    std::vector<paragraph::PackedSiteInput> sites(next.n_here * n_samples);
                    for (size_t i = 0; i < sites.size(); ++i)
                    {
                        sites[i].description = &next.chunk->graphs[i / n_samples];
                        sites[i].reads = &next.chunk->packed[i];
                    }
                    next.batch = paragraph::beginPackedBatch(site_parameters, std::move(sites));  // queued: the device works on
                    phase(c, "submit");
                }
                finish(current);  // the chunk before: on the device while this one was prepared
                current = std::move(next);
            }
            finish(current);
        }
        catch (...)
        {
            if (!failed.exchange(true))
                failure = std::current_exception();
        }
        std::lock_guard<std::mutex> lock(timings_mutex);
        trace.insert(trace.end(), my_trace.begin(), my_trace.end());
        lane_timings_total.load_graphs += mine.load_graphs;
        lane_timings_total.extract_reads += mine.extract_reads;
        lane_timings_total.device_batch += mine.device_batch;
        lane_timings_total.documents += mine.documents;
        lane_timings_total.genotypes += mine.genotypes;
        lane_timings_total.release += mine.release;
        lane_timings_total.sites += mine.sites;
        lane_timings_total.reads += mine.reads;
        lane_timings_total.batches += mine.batches;
    };
    const double t_lanes = now();
    std::vector<std::thread> pool;
    for (size_t l = 1; l < lanes; ++l)
        pool.emplace_back(lane);
    lane();
    for (auto& t : pool)
        t.join();
    if (failure)
        std::rethrow_exception(failure);
    if (trace_path)
    {
        std::ofstream out(trace_path, std::ios::app);
        out << "# lanes=" << lanes << " threads/lane=" << lane_threads << " chunks=" << n_chunks << " setup_s=" << (t_lanes - t_call) << " total_s=" << (now() - t_call) << "\n";
        for (auto const& line : trace)
            out << line << "\n";
    }
    if (parameters.timings)
    {
        *parameters.timings = lane_timings_total;
        parameters.timings->lanes = lanes;
    }
    return genotypes;
}
}  // namespace grmpy

// ----------------------------------------------------------------------------------------------------------------------
// C entry point (include/paragraph_workflow.h)
// ----------------------------------------------------------------------------------------------------------------------
#include <cstring>
#include <fstream>

#include "../../../include/paragraph_workflow.h"

extern "C" int pgw_genotype_graphs(
    const char* reference_fasta, const char* manifest, const char* const* graph_paths, size_t n_graphs,
    const char* genotyping_parameters, const char* options_json, const char* output_path, char* error, size_t error_cap)
{
    auto report = [&](std::string const& what) {
        if (error && error_cap)
        {
            strncpy(error, what.c_str(), error_cap - 1);
            error[error_cap - 1] = '\0';
        }
        return 1;
    };
    try
    {
        if (!reference_fasta || !manifest || !output_path || (n_graphs && !graph_paths))
            return report("pgw_genotype_graphs: null argument");
        grmpy::Parameters parameters;
        if (options_json && *options_json)
        {
            const Json options = Json::parse(options_json);
            if (!options.isObject())
                return report("pgw_genotype_graphs: options_json must be a JSON object");
            for (auto const& kv : options.members())
            {
                if (kv.first == "threads")
                    parameters.threads = (int)kv.second.asInt64();
                else if (kv.first == "lanes")
                    parameters.lanes = (int)kv.second.asInt64();
                else if (kv.first == "sites_per_batch")
                    parameters.sites_per_batch = (size_t)kv.second.asUInt64();
                else if (kv.first == "max_reads")
                    parameters.max_reads = (int)kv.second.asInt64();
                else if (kv.first == "bad_align_frac")
                    parameters.bad_align_frac = (float)kv.second.asDouble();
                else if (kv.first == "path_sequence_matching")
                    parameters.path_sequence_matching = kv.second.asBool();
                else if (kv.first == "exact_match_shortcut")
                    parameters.exact_match_shortcut = kv.second.asBool();
                else if (kv.first == "kmer_sequence_matching")
                    parameters.kmer_sequence_matching = kv.second.asBool();
                else if (kv.first == "klib_sequence_matching")
                    parameters.klib_sequence_matching = kv.second.asBool();
                else if (kv.first == "bad_align_uniq_kmer_len")
                    parameters.bad_align_uniq_kmer_len = (int)kv.second.asInt64();
                else if (kv.first == "packed_reads")
                    parameters.packed_reads = kv.second.asBool();
                else if (kv.first == "devices")
                {
                    if (!kv.second.isArray())
                        return report("pgw_genotype_graphs: \"devices\" must be an array of device ordinals");
                    for (size_t d = 0; d < kv.second.size(); ++d)
                        parameters.devices.push_back((int)kv.second[d].asInt64());
                }
                else
                    return report("pgw_genotype_graphs: unknown option " + kv.first);
            }
        }
        std::vector<std::string> graphs(graph_paths, graph_paths + n_graphs);
        const genotyping::Samples samples = genotyping::loadManifest(manifest);
        // Every document is serialised by the lane that made it (Parameters::genotype_text) and written out -- in graph order -- by
        // whichever lane completes the next stretch of the order, while the other lanes work: joining 10 000 texts into one string
        // and writing it after the last lane had finished was 40 ms of one thread at the end of a 135 ms job.
        std::vector<std::string> text;
        parameters.genotype_text = &text;
        FILE* out = fopen(output_path, "wb");
        if (!out)
            return report(std::string("cannot write ") + output_path);
        std::vector<char> buffer(1 << 20);
        setvbuf(out, buffer.data(), _IOFBF, buffer.size());
        struct Ordered
        {
            std::mutex m;
            std::vector<char> ready;
            size_t next = 0;
            bool ok = true;
        } ordered;
        ordered.ready.assign(n_graphs, 0);
        ordered.ok = fputc('[', out) != EOF;
        auto flush = [&](size_t first, size_t last) {
            std::lock_guard<std::mutex> lock(ordered.m);
            for (size_t g = first; g < last && g < ordered.ready.size(); ++g)
                ordered.ready[g] = 1;
            while (ordered.next < text.size() && ordered.ready[ordered.next])
            {
                std::string& t = text[ordered.next];
                const char* sep = ordered.next ? ",\n" : "\n";
                // (a graph nothing was written for -- no samples -- is `null`, like Json::dump() of the empty document: never a hole)
                ordered.ok = ordered.ok && fputs(sep, out) != EOF
                    && (t.empty() ? fputs("null", out) != EOF : fwrite(t.data(), 1, t.size(), out) == t.size());
                ++ordered.next;  // (the text is freed with the others at the end: freed here it would go back to ANOTHER lane's arena)
            }
        };
        parameters.genotype_text_ready = flush;
        try
        {
            grmpy::genotypeGraphs(parameters, graphs, reference_fasta, samples, genotyping_parameters ? genotyping_parameters : "");
        }
        catch (...)
        {
            fclose(out);
            remove(output_path);  // (half a document array is worse than none)
            throw;
        }
        flush(0, n_graphs);  // (whatever a path without lanes left: a single chunk, no graphs at all)
        ordered.ok = ordered.ok && fputs("\n]\n", out) != EOF;
        if (fclose(out) != 0 || !ordered.ok)
            return report(std::string("error while writing ") + output_path);
        return 0;
    }
    catch (std::exception const& e)
    {
        return report(e.what());
    }
    catch (...)
    {
        return report("unknown error");
    }
}

