// Host-side documents around the realignment core: graph descriptions in, manifests in, graph coordinates and the
// post-hoc statistics of the count document.  CPU only; headers cite the reference interfaces.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstring>
#include <fstream>
#include <limits>
#include <list>
#include <memory>
#include <set>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

#include "common/Fasta.hh"
#include "genotyping/SampleInfo.hh"
#include "graphcore/GraphCoordinates.hh"
#include "grm/GraphInput.hh"
#include "paragraph/Statistics.hh"

using common::Json;
using graphtools::Graph;
using graphtools::NodeId;

namespace
{
[[noreturn]] void fail(std::string const& msg) { throw std::runtime_error(msg); }

std::string upper(std::string s)
{
    std::transform(s.begin(), s.end(), s.begin(), ::toupper);
    return s;
}

bool isFile(std::string const& p)
{
    std::ifstream f(p);
    return f.good();
}
}  // namespace

// ----------------------------------------------------------------------------------------------------------------------
namespace grm
{
namespace
{
Graph buildGraph(Json const& in, std::string const& reference, common::FastaFile const* opened, bool store_ref_sequence);
}

Graph graphFromJson(Json const& in, std::string const& reference, bool store_ref_sequence)
{
    return buildGraph(in, reference, nullptr, store_ref_sequence);
}

Graph graphFromJson(Json const& in, common::FastaFile const& reference, bool store_ref_sequence)
{
    return buildGraph(in, reference.getFilename(), &reference, store_ref_sequence);
}

namespace
{
Graph buildGraph(Json const& in, std::string const& reference, common::FastaFile const* opened, bool store_ref_sequence)
{
    Json const& spec = in.isMember("graph") ? in["graph"] : in;
    if (!spec["nodes"].isArray())
        fail("Graph description has no \"nodes\" array");
    if (!spec["edges"].isNull() && !spec["edges"].isArray())
        fail("Graph description: \"edges\" must be an array");
    Json::Elements const& nodes = spec["nodes"].elements();
    std::unique_ptr<common::FastaFile> own;  // opened on first use: graphs with explicit sequences need no reference
    auto ref_bases = [&](std::string const& where) {
        if (opened)
            return opened->query(where);
        if (!own)
            own.reset(new common::FastaFile(reference));
        return own->query(where);
    };

    Graph graph(nodes.size(), false);
    std::map<std::string, NodeId> id_of;
    for (NodeId i = 0; i < nodes.size(); ++i)
    {
        Json const& node = nodes[i];
        const std::string name = node.isMember("name") ? node["name"].asString() : "node-" + std::to_string(i + 1);
        if (!id_of.emplace(name, i).second)
            fail("Graph description: duplicate node name " + name);
        graph.setNodeName(i, name);
        const std::string uc = upper(name);
        const bool terminal = (i == 0 || i + 1 == nodes.size()) && (uc == "SOURCE" || uc == "SINK");
        if (terminal)
        {
            graph.setNodeSeq(i, "X");
            continue;
        }
        if (node.isMember("sequence"))
        {
            graph.setNodeSeq(i, node["sequence"].asString());
            continue;
        }
        if (!node.isMember("reference"))
            fail("Graph description: node " + name + " has neither \"sequence\" nor \"reference\"");
        std::string bases;
        if (node["reference"].isString())
            bases = ref_bases(node["reference"].asString());
        else
        {
            if (!node["reference"].isArray())
                fail("Graph description: \"reference\" of node " + name + " must be a string or an array of strings");
            for (Json const& where : node["reference"].elements())
            {
                const std::string here = ref_bases(where.asString());
                if (!bases.empty() && bases != here)
                    fail("Graph description: reference intervals of node " + name + " differ in sequence");
                bases = here;
            }
        }
        if (store_ref_sequence)
        {
            if (bases.empty())
                fail("Graph description: empty reference sequence for node " + name);
            graph.setNodeSeq(i, bases);
        }
    }
    auto node_id = [&](Json const& name) {
        auto it = id_of.find(name.asString());
        if (it == id_of.end())
            fail("Graph description: edge refers to unknown node " + name.asString());
        return it->second;
    };
    for (Json const& edge : spec["edges"].elements())
    {
        const NodeId from = node_id(edge["from"]), to = node_id(edge["to"]);
        graph.addEdge(from, to);
        for (Json const& label : edge["sequences"].elements())
            graph.addLabelToEdge(from, to, label.asString());
    }
    for (NodeId i = 0; i < nodes.size(); ++i)
    {
        for (Json const& label : nodes[i]["sequences"].elements())
        {
            for (NodeId p : graph.predecessors(i))
                graph.addLabelToEdge(p, i, label.asString());
            for (NodeId s : graph.successors(i))
                graph.addLabelToEdge(i, s, label.asString());
        }
    }
    return graph;
}
}  // namespace

std::list<graphtools::Path> pathsFromJson(Graph const* graph, Json const& in_paths)
{
    std::list<graphtools::Path> paths;
    if (in_paths.isNull())
        return paths;
    if (!in_paths.isArray())
        fail("Graph description: \"paths\" must be an array");
    std::unordered_map<std::string, NodeId> id_of;
    for (NodeId n = 0; n != graph->numNodes(); ++n)
        id_of[graph->nodeName(n)] = n;
    for (Json const& path : in_paths.elements())
    {
        if (!path["nodes"].isArray() || path["nodes"].size() == 0)
            fail("Graph description: path without nodes");
        graphtools::Path out;
        out.graph = graph;
        for (Json const& name : path["nodes"].elements())
        {
            auto it = id_of.find(name.asString());
            if (it == id_of.end())
                fail("Graph description: path refers to unknown node " + name.asString());
            out.nodes.push_back(it->second);
        }
        out.start_position = 0;
        out.end_position = (int32_t)graph->nodeSeq(out.nodes.back()).size() - 1;
        paths.push_back(out);
    }
    return paths;
}
}  // namespace grm

// ----------------------------------------------------------------------------------------------------------------------
namespace genotyping
{
void SampleInfo::set_autosome_depth(double v)
{
    autosome_depth_ = v;
    if (depth_sd_ == 0)
        depth_sd_ = std::sqrt(autosome_depth_ * 5);
}

void SampleInfo::set_sex(std::string sex_string)
{
    const char c = sex_string.empty() ? '\0' : (char)tolower((unsigned char)sex_string[0]);
    if (c == 'm')
        sex_ = Sex::MALE;
    else if (c == 'f')
        sex_ = Sex::FEMALE;
    else if (c == 'u')
        sex_ = Sex::UNKNOWN;
    else
        fail("illegal sex string: " + sex_string);
}

namespace
{
// stringutil::split: any of the separator characters ends a field, empty fields are dropped
std::vector<std::string> fields(std::string const& line, const char* seps)
{
    std::vector<std::string> out;
    std::string cur;
    for (char c : line)
    {
        if (strchr(seps, c))
        {
            if (!cur.empty())
                out.push_back(cur);
            cur.clear();
        }
        else
            cur += c;
    }
    if (!cur.empty())
        out.push_back(cur);
    return out;
}

std::string parentOf(std::string const& path)
{
    const size_t slash = path.rfind('/');
    return slash == std::string::npos ? std::string(".") : path.substr(0, slash);
}

double toDouble(std::string const& s, double fallback)
{
    try
    {
        return std::stod(s);
    }
    catch (std::exception const&)
    {
        return fallback;
    }
}
}  // namespace

Samples loadManifest(std::string const& filename)
{
    std::ifstream in(filename);
    if (!in.is_open())
        fail("Unable to open manifest: " + filename);
    Samples samples;
    std::vector<std::string> header;
    std::map<std::string, size_t> column;
    auto has = [&](const char* name) { return column.count(name) != 0; };
    std::string line;
    while (std::getline(in, line))
    {
        line.erase(std::remove(line.begin(), line.end(), '#'), line.end());
        line.erase(std::remove(line.begin(), line.end(), '\r'), line.end());
        if (line.empty())
            continue;
        if (header.empty())
        {
            header = fields(line, "\t,");
            static const std::set<std::string> legal = { "id",    "path",        "index_path", "paragraph",      "idxdepth",
                                                         "depth", "read length", "sex",        "depth variance", "depth sd" };
            for (size_t j = 0; j < header.size(); ++j)
            {
                std::transform(header[j].begin(), header[j].end(), header[j].begin(), ::tolower);
                if (!legal.count(header[j]))
                    fail("Unknown column " + header[j] + " in manifest");
                column[header[j]] = j;
            }
            for (const char* required : { "id", "path" })
                if (!has(required))
                    fail(std::string("Required header ") + required + " not present in manifest");
            if (!(has("idxdepth") || (has("depth") && has("read length"))))
                fail("Manifest header must either specify index depth locations or depth and read length.");
            continue;
        }
        std::vector<std::string> tokens = fields(line, "\t,");
        tokens.resize(header.size(), "");
        SampleInfo sample;
        sample.set_sample_name(tokens[column["id"]]);
        auto locate = [&](std::string const& given, bool must_exist) {
            if (given.compare(0, 5, "s3://") == 0 || given.compare(0, 7, "http://") == 0 || given.compare(0, 8, "https://") == 0)
                return given;
            if (isFile(given))
                return given;
            const std::string beside = parentOf(filename) + "/" + given;
            if (isFile(beside))
                return beside;
            if (must_exist)
                fail("Sample " + sample.sample_name() + ": File not found: " + beside);
            return given;
        };
        sample.set_filename(locate(tokens[column["path"]], true));
        if (has("index_path"))
            sample.set_index_filename(locate(tokens[column["index_path"]], true));

        double depth = -1;
        int read_length = -1;
        if (has("depth") && has("read length"))
        {
            try
            {
                depth = std::stod(tokens[column["depth"]]);
                read_length = std::stoi(tokens[column["read length"]]);
            }
            catch (std::exception const&)
            {
            }
        }
        if ((depth < 0 || read_length < 0) && has("idxdepth"))
        {
            try
            {
                const Json idx = Json::parseFile(locate(tokens[column["idxdepth"]], false));
                if (read_length < 0 && idx.isMember("read_length"))
                    read_length = (int)idx["read_length"].asInt64();
                if (depth < 0 && idx.isMember("autosome") && idx["autosome"].isMember("depth"))
                    depth = idx["autosome"]["depth"].asDouble();
            }
            catch (std::exception const&)
            {
                // an unreadable idxdepth document only matters if nothing else gave depth / read length (checked next)
            }
        }
        if (depth <= 0 || read_length <= 0)
            fail("No depth / read length estimate for sample " + sample.sample_name());
        sample.set_autosome_depth(depth);
        sample.set_read_length((unsigned)read_length);
        if (has("depth sd"))
        {
            const double sd = toDouble(tokens[column["depth sd"]], 0);
            if (sd <= 0)
                fail("Depth sd is not positive in sample " + sample.sample_name());
            sample.set_depth_sd(sd);
        }
        else if (has("depth variance"))
        {
            const double variance = toDouble(tokens[column["depth variance"]], 0);
            if (variance <= 0)
                fail("Depth variance is not positive in sample " + sample.sample_name());
            sample.set_depth_sd(std::sqrt(variance));
        }
        if (has("sex"))
            sample.set_sex(tokens[column["sex"]]);
        if (has("paragraph"))
        {
            const std::string doc = locate(tokens[column["paragraph"]], false);
            if (isFile(doc))
                sample.set_alignment_data(Json::parseFile(doc));
        }
        samples.push_back(sample);
    }
    return samples;
}
}  // namespace genotyping

// ----------------------------------------------------------------------------------------------------------------------
namespace graphtools
{
GraphCoordinates::GraphCoordinates(Graph const* graph) : graph_(graph)
{
    const NodeId n_nodes = (NodeId)graph->numNodes();
    uint64_t at = 0;
    starts_.resize(n_nodes);
    for (NodeId node = 0; node < n_nodes; ++node)
    {
        starts_[node] = at;
        at += std::max<size_t>(1, graph->nodeSeq(node).size());
        // nodes come in topological order, so every predecessor's distances are final by now
        for (NodeId from = 0; from < n_nodes; ++from)
        {
            if (from == node || graph->hasEdge(from, node))
                continue;
            uint64_t best = kNoPath;
            for (NodeId pred : graph->predecessors(node))
            {
                auto known = end_to_start_.find({ from, pred });
                if (known != end_to_start_.end())
                    best = std::min<uint64_t>(best, known->second + graph->nodeSeq(pred).size());
                else if (graph->hasEdge(from, pred))
                    best = std::min<uint64_t>(best, graph->nodeSeq(pred).size());
            }
            if (best != kNoPath)
                end_to_start_[{ from, node }] = best;
        }
    }
}

std::pair<uint64_t, uint64_t> GraphCoordinates::canonicalStartAndEnd(Path const& path) const
{
    std::pair<uint64_t, uint64_t> span(kNoPath, kNoPath);
    span.first = canonicalPos(path.nodes.front(), (uint64_t)path.start_position);
    if (path.end_position > 0)
        span.second = canonicalPos(path.nodes.back(), (uint64_t)path.end_position);
    if (span.first > span.second)
        std::swap(span.first, span.second);
    return span;
}

void GraphCoordinates::nodeAndOffset(uint64_t canonical_pos, NodeId& node, uint64_t& offset) const
{
    // the node with the last start <= canonical_pos; positions past the end stay on the last node
    auto after = std::upper_bound(starts_.begin(), starts_.end(), canonical_pos);
    node = after == starts_.begin() ? 0 : (NodeId)(after - starts_.begin() - 1);
    offset = canonical_pos - starts_[node];
}

uint64_t GraphCoordinates::distance(uint64_t pos1, uint64_t pos2) const
{
    if (pos1 == pos2)
        return 0;
    if (pos2 < pos1)
        std::swap(pos1, pos2);
    NodeId n1, n2;
    uint64_t o1, o2;
    nodeAndOffset(pos1, n1, o1);
    nodeAndOffset(pos2, n2, o2);
    if (n1 == n2)
        return pos2 - pos1;
    const uint64_t rest_of_n1 = graph_->nodeSeq(n1).size() - o1;
    if (graph_->hasEdge(n1, n2))
        return rest_of_n1 + o2;
    auto between = end_to_start_.find({ n1, n2 });
    return between == end_to_start_.end() ? kNoPath : rest_of_n1 + o2 + between->second;
}
}  // namespace graphtools

// ----------------------------------------------------------------------------------------------------------------------
namespace paragraph
{
std::vector<NodeAlignment> decodeGraphCigar(std::string const& text, Graph const& graph)
{
    std::vector<NodeAlignment> out;
    size_t i = 0;
    auto number = [&]() {
        if (i >= text.size() || !isdigit((unsigned char)text[i]))
            fail("Malformed graph CIGAR: " + text);
        uint64_t v = 0;
        while (i < text.size() && isdigit((unsigned char)text[i]))
            v = v * 10 + (uint64_t)(text[i++] - '0');
        return v;
    };
    while (i < text.size())
    {
        NodeAlignment na;
        const uint64_t id = number();
        if (id >= graph.numNodes() || i >= text.size() || text[i] != '[')
            fail("Malformed graph CIGAR: " + text);
        na.node = (NodeId)id;
        ++i;
        while (i < text.size() && text[i] != ']')
        {
            const uint32_t len = (uint32_t)number();
            if (i >= text.size())
                fail("Malformed graph CIGAR: " + text);
            switch (text[i++])
            {
            case 'M': na.matched += len; break;
            case 'X': na.mismatched += len; break;
            case 'S': na.clipped += len; break;
            case 'I': na.inserted += len; break;
            case 'D': na.deleted += len; break;
            case 'N': na.missing += len; break;
            default: fail("Malformed graph CIGAR: " + text);
            }
        }
        if (i >= text.size())
            fail("Malformed graph CIGAR: " + text);
        ++i;  // ']'
        out.push_back(na);
    }
    if (out.empty())
        fail("Empty graph CIGAR");
    return out;
}

void RunningStats::add(double x)
{
    ++n_;
    sum_ += x;
    imm_mean_ += (x - imm_mean_) / (double)n_;
    if (n_ > 1)
    {
        const double d = x - imm_mean_;
        var_ = var_ * (double)(n_ - 1) / (double)n_ + d * d / (double)(n_ - 1);
    }
    if (n_ <= 5)
    {
        heights_[n_ - 1] = x;
        if (n_ == 5)
            std::sort(heights_, heights_ + 5);
        return;
    }
    static const double kIncrement[5] = { 0, 0.25, 0.5, 0.75, 1 };
    size_t cell;
    if (x < heights_[0])
    {
        heights_[0] = x;
        cell = 1;
    }
    else if (x >= heights_[4])
    {
        heights_[4] = x;
        cell = 4;
    }
    else
        cell = (size_t)(std::upper_bound(heights_, heights_ + 5, x) - heights_);
    for (size_t i = cell; i < 5; ++i)
        actual_[i] += 1;
    for (size_t i = 0; i < 5; ++i)
        desired_[i] += kIncrement[i];
    for (size_t i = 1; i <= 3; ++i)
    {
        const double d = desired_[i] - actual_[i];
        const double dp = actual_[i + 1] - actual_[i], dm = actual_[i - 1] - actual_[i];
        const double hp = (heights_[i + 1] - heights_[i]) / dp, hm = (heights_[i - 1] - heights_[i]) / dm;
        if ((d >= 1 && dp > 1) || (d <= -1 && dm < -1))
        {
            const double sign = d > 0 ? 1 : -1;
            const double parabolic = heights_[i] + sign / (dp - dm) * ((sign - dm) * hp + (dp - sign) * hm);
            if (heights_[i - 1] < parabolic && parabolic < heights_[i + 1])
                heights_[i] = parabolic;
            else if (d > 0)
                heights_[i] += hp;
            else
                heights_[i] -= hm;
            actual_[i] += sign;
        }
    }
}

double RunningStats::mean() const { return n_ ? sum_ / (double)n_ : std::numeric_limits<double>::quiet_NaN(); }
double RunningStats::variance() const { return var_; }
double RunningStats::median() const { return heights_[2]; }

SiteReadViews viewsOfReads(Graph const& graph, std::vector<common::Read const*> const& reads)
{
    SiteReadViews v;
    const std::set<std::string> labels = graph.allLabels();
    v.label_names.assign(labels.begin(), labels.end());
    std::unordered_map<std::string, uint32_t> fragment_of;
    v.reads.reserve(reads.size());
    for (common::Read const* read : reads)
    {
        MappedReadView m;
        m.fragment = fragment_of.emplace(read->fragment_id(), (uint32_t)fragment_of.size()).first->second;
        m.read_length = (uint32_t)read->bases().size();
        m.chrom_id = read->chrom_id();
        m.pos = read->pos();
        m.mate_chrom_id = read->mate_chrom_id();
        m.mate_pos = read->mate_pos();
        m.is_mapped = read->is_mapped();
        m.is_mate_mapped = read->is_mate_mapped();
        m.is_reverse_strand = read->is_reverse_strand();
        m.is_mate_reverse_strand = read->is_mate_reverse_strand();
        m.is_graph_mapped = read->graph_mapping_status() == common::Read::MAPPED;
        m.is_graph_reverse_strand = read->is_graph_reverse_strand();
        m.graph_pos = read->graph_pos();
        m.graph_alignment_score = read->graph_alignment_score();
        if (m.is_graph_mapped)
        {
            const auto pieces = decodeGraphCigar(read->graph_cigar(), graph);
            m.pieces_off = (uint32_t)v.pieces.size();
            m.n_pieces = (uint32_t)pieces.size();
            v.pieces.insert(v.pieces.end(), pieces.begin(), pieces.end());
        }
        for (auto const& name : read->graph_sequences_supported())
        {
            const auto it = std::lower_bound(v.label_names.begin(), v.label_names.end(), name);
            if (it != v.label_names.end() && *it == name)
                m.sequences.set((size_t)(it - v.label_names.begin()));
        }
        v.reads.push_back(m);
    }
    v.n_fragments = (uint32_t)fragment_of.size();
    return v;
}

namespace
{
const uint64_t kNoLength = std::numeric_limits<uint64_t>::max();

// (a fragment is one or two reads: their spans and lengths live in the shape itself -- a list and a vector per fragment were two
// or three allocations for each of a site's hundred fragments; only a third graph-mapped read spills into `more`)
struct FragmentShape
{
    unsigned n_reads = 0, n_spans = 0;
    uint64_t bam_length = kNoLength, graph_length = kNoLength;
    std::pair<uint64_t, uint64_t> span[2];
    uint64_t length[2] = { 0, 0 };
    std::vector<std::pair<uint64_t, uint64_t>> more;  // every span, once there are three or more
};

void addToFragment(FragmentShape& f, graphtools::GraphCoordinates const& coords, SiteReadViews const& views, MappedReadView const& read)
{
    ++f.n_reads;
    const bool proper = read.is_mapped && read.is_mate_mapped && read.is_reverse_strand != read.is_mate_reverse_strand
        && read.mate_chrom_id == read.chrom_id;
    f.bam_length = (!proper || f.n_reads > 2) ? kNoLength : (uint64_t)std::abs(read.mate_pos - read.pos) + read.read_length;
    if (!read.is_graph_mapped || read.n_pieces == 0)
        return;
    const NodeAlignment* nodes = &views.pieces[read.pieces_off];
    graphtools::Path walk;
    walk.graph = &coords.getGraph();
    walk.start_position = read.graph_pos;
    uint64_t query_length = 0;
    for (uint32_t k = 0; k < read.n_pieces; ++k)
    {
        walk.nodes.push_back(nodes[k].node);
        query_length += nodes[k].queryLength();
    }
    // the LAST aligned base of the last node (inclusive), as decodeGraphAlignment builds the alignment's path
    // (GT!/src/graphalign/GraphAlignmentOperations.cpp:103-104); canonicalStartAndEnd then treats an end of 0 as "unknown"
    walk.end_position = (int32_t)nodes[read.n_pieces - 1].referenceLength() + (read.n_pieces == 1 ? read.graph_pos : 0) - 1;
    const std::pair<uint64_t, uint64_t> this_span = coords.canonicalStartAndEnd(walk);
    if (f.n_spans < 2)
    {
        f.span[f.n_spans] = this_span;
        f.length[f.n_spans] = query_length;
    }
    else
    {
        if (f.more.empty())
            f.more.assign(f.span, f.span + 2);
        f.more.push_back(this_span);
    }
    ++f.n_spans;
    if (f.n_spans == 1)
        f.graph_length = f.length[0];
    else if (f.n_spans == 2)
    {
        const uint64_t gap = std::min(coords.distance(f.span[0].second, f.span[1].first), coords.distance(f.span[1].second, f.span[0].first));
        f.graph_length = gap == kNoLength ? kNoLength : f.length[0] + f.length[1] + gap;
    }
    else
    {
        // three or more graph-mapped reads in one fragment: each step adds the start-to-start distance, twice from the
        // second read on (kept as the original computes it); the spans stay sorted from step to step, as the original's list does
        std::stable_sort(f.more.begin(), f.more.end(),
                         [](std::pair<uint64_t, uint64_t> const& a, std::pair<uint64_t, uint64_t> const& b) { return a.first < b.first; });
        uint64_t previous = 0, length = 0;
        bool has_previous = false;
        for (auto const& span : f.more)
        {
            const uint64_t step = coords.distance(previous, span.first);
            if (step == kNoLength)
            {
                length = kNoLength;
                break;
            }
            length += has_previous ? 2 * step : step;
            previous = span.first;
            has_previous = true;
        }
        f.graph_length = length;
    }
}
}  // namespace

Json fragmentStatistics(Graph const& graph, SiteReadViews const& views)
{
    graphtools::GraphCoordinates coords(&graph);
    // fragments in order of first appearance (the order the running estimators see them in)
    std::vector<FragmentShape> fragments;
    std::vector<uint32_t> slot_of(views.n_fragments, (uint32_t)-1);
    for (MappedReadView const& read : views.reads)
    {
        if (read.fragment >= slot_of.size())
            slot_of.resize(read.fragment + 1, (uint32_t)-1);
        if (slot_of[read.fragment] == (uint32_t)-1)
        {
            slot_of[read.fragment] = (uint32_t)fragments.size();
            fragments.emplace_back();
        }
        addToFragment(fragments[slot_of[read.fragment]], coords, views, read);
    }
    RunningStats linear, on_graph;
    uint64_t bad_linear = 0, bad_graph = 0, single = 0, paired = 0, multi = 0;
    for (FragmentShape const& f : fragments)
    {
        if (f.bam_length == kNoLength)
            ++bad_linear;
        else if (f.n_reads >= 2)
            linear.add((double)f.bam_length);
        if (f.graph_length == kNoLength)
            ++bad_graph;
        else if (f.n_reads >= 2)
            on_graph.add((double)f.graph_length);
        ++(f.n_reads == 1 ? single : f.n_reads == 2 ? paired : multi);
    }
    Json stats = Json::object();
    stats["mean_linear"] = linear.mean();
    stats["mean_graph"] = on_graph.mean();
    stats["median_linear"] = linear.median();
    stats["median_graph"] = on_graph.median();
    stats["variance_linear"] = linear.variance();
    stats["variance_graph"] = on_graph.variance();
    stats["single_read"] = single;
    stats["paired_read"] = paired;
    stats["multi_read"] = multi;
    stats["problematic_linear"] = bad_linear;
    stats["problematic_graph"] = bad_graph;
    return stats;
}

namespace
{
struct Tally
{
    explicit Tally(size_t length_ = 0) : length(length_) {}
    size_t length;
    uint64_t match = 0, mismatch = 0, gap = 0, clip = 0;
    int fwd = 0, rev = 0;
    void bases(NodeAlignment const& a, bool with_clips)
    {
        match += a.matched;
        mismatch += a.mismatched;
        gap += a.inserted + a.deleted;
        if (with_clips)
            clip += a.clipped;
    }
    void strand(bool reverse) { ++(reverse ? rev : fwd); }
    void merge(Tally const& o)  // two edges whose "<from>_<to>" names coincide share one entry in the document
    {
        match += o.match;
        mismatch += o.mismatch;
        gap += o.gap;
        clip += o.clip;
        fwd += o.fwd;
        rev += o.rev;
    }
    Json toJson() const
    {
        Json out = Json::object();
        const double aligned = (double)(match + mismatch + gap);
        out["num_fwd_reads"] = fwd;
        out["num_rev_reads"] = rev;
        out["mismatch_rate"] = (double)mismatch / aligned;
        out["gap_rate"] = (double)gap / aligned;
        out["clip_rate"] = (double)clip / aligned;
        if (length > 0)
            out["match_base_depth"] = (double)match / (double)length;
        out["contig_length"] = (int)length;
        return out;
    }
};
}  // namespace

Json alignmentStatistics(Graph const& graph, SiteReadViews const& views)
{
    const NodeId n_nodes = (NodeId)graph.numNodes();
    // an allele's length: the nodes that carry its label on an edge in AND an edge out (labels as bits of views.label_names,
    // which holds every label of the graph in sorted order: no string sets per node)
    std::vector<size_t> allele_length(views.label_names.size(), 0);
    {
        auto bits_of = [&](std::set<std::string> const& labels, LabelSet& into) {
            for (auto const& label : labels)
            {
                const auto it = std::lower_bound(views.label_names.begin(), views.label_names.end(), label);
                if (it != views.label_names.end() && *it == label)
                    into.set((size_t)(it - views.label_names.begin()));
            }
        };
        for (NodeId node = 0; node < n_nodes; ++node)
        {
            LabelSet in, out;
            for (NodeId p : graph.predecessors(node))
                bits_of(graph.edgeLabels(p, node), in);
            for (NodeId s : graph.successors(node))
                bits_of(graph.edgeLabels(node, s), out);
            for (size_t b = 0; b < views.label_names.size(); ++b)
                if (in.test(b) && out.test(b))
                    allele_length[b] += graph.nodeSeq(node).size();
        }
    }
    const bool terminals = n_nodes && (graph.nodeName(0) == "source" || graph.nodeName(n_nodes - 1) == "sink");
    // tallies are kept by id while the reads go by (names only when the document is written)
    std::vector<Tally> node_tally(n_nodes);
    std::vector<bool> node_seen(n_nodes, false);
    std::map<std::pair<NodeId, NodeId>, Tally> edge_tally;
    std::vector<Tally> allele_tally(views.label_names.size());
    std::vector<bool> allele_seen(views.label_names.size(), false);
    std::vector<int> allele_score_by_label(views.label_names.size(), 0);
    for (MappedReadView const& read : views.reads)
    {
        if (!read.is_graph_mapped)
            continue;
        const NodeAlignment* pieces = read.n_pieces ? &views.pieces[read.pieces_off] : nullptr;
        const bool reverse = read.is_graph_reverse_strand;
        for (size_t k = 0; k < read.n_pieces; ++k)
        {
            const NodeId node = pieces[k].node;
            const bool terminal = terminals && (node == 0 || node == n_nodes - 1);
            if (!node_seen[node])
            {
                node_seen[node] = true;
                node_tally[node] = Tally(graph.nodeSeq(node).size());
            }
            node_tally[node].bases(pieces[k], !terminal);
            node_tally[node].strand(reverse);
            if (k == 0)
                continue;
            const NodeId prev = pieces[k - 1].node;
            auto it = edge_tally.find({ prev, node });
            if (it == edge_tally.end())
                it = edge_tally.emplace(std::make_pair(prev, node), Tally(graph.nodeSeq(prev).size() + graph.nodeSeq(node).size())).first;
            // clips on the `from` side count only when the `to` node is node 1 of a graph with terminals; on the `to` side
            // only when `to` is itself a terminal (the original's argument order, GraphSummaryStatistics.cpp:131-135)
            it->second.bases(pieces[k - 1], terminals && node == 1);
            it->second.bases(pieces[k], terminal);
            it->second.strand(reverse);
        }
        for (size_t b = 0; b < views.label_names.size(); ++b)
        {
            if (!read.sequences.test(b))
                continue;
            if (!allele_seen[b])
            {
                allele_seen[b] = true;
                allele_tally[b] = Tally(allele_length[b]);
            }
            for (size_t k = 0; k < read.n_pieces; ++k)
                allele_tally[b].bases(pieces[k], !(terminals && (pieces[k].node == 0 || pieces[k].node == n_nodes - 1)));
            allele_tally[b].strand(reverse);
            allele_score_by_label[b] += read.graph_alignment_score;
        }
    }
    std::map<std::string, Tally> node_stats, edge_stats, allele_stats;
    std::map<std::string, int> allele_score;
    for (NodeId node = 0; node < n_nodes; ++node)
        if (node_seen[node])
            node_stats.emplace(graph.nodeName(node), node_tally[node]);
    for (auto const& kv : edge_tally)
    {
        auto placed = edge_stats.emplace(graph.nodeName(kv.first.first) + "_" + graph.nodeName(kv.first.second), kv.second);
        if (!placed.second)
            placed.first->second.merge(kv.second);
    }
    for (size_t b = 0; b < views.label_names.size(); ++b)
        if (allele_seen[b])
        {
            allele_stats.emplace(views.label_names[b], allele_tally[b]);
            allele_score[views.label_names[b]] = allele_score_by_label[b];
        }
    Json out = Json::object();
    out["nodes"] = Json::object();
    out["edges"] = Json::object();
    out["alleles"] = Json::object();
    for (auto const& kv : node_stats)
        out["nodes"][kv.first] = kv.second.toJson();
    for (auto const& kv : edge_stats)
        out["edges"][kv.first] = kv.second.toJson();
    for (auto const& kv : allele_stats)
    {
        Json entry = kv.second.toJson();
        const int n = kv.second.fwd + kv.second.rev;
        entry["avr_score"] = n == 0 ? 0.0 : (double)allele_score[kv.first] / n;
        out["alleles"][kv.first] = std::move(entry);
    }
    return out;
}
}  // namespace paragraph

// ----------------------------------------------------------------------------------------------------------------------
// chunk schedule of grmpy::genotypeGraphs (no device calls: lives here so that the CPU test programs link it)
// ----------------------------------------------------------------------------------------------------------------------
#include "paragraph/Workflow.hh"

namespace grmpy
{
std::vector<std::pair<size_t, size_t>> chunkSchedule(size_t n_graphs, size_t graphs_per_chunk, size_t lanes)
{
    std::vector<std::pair<size_t, size_t>> ranges;
    const size_t per_chunk = std::max<size_t>(1, graphs_per_chunk);
    lanes = std::max<size_t>(1, lanes);
    const size_t n_even_chunks = (n_graphs + per_chunk - 1) / per_chunk;
    const bool shaped = lanes > 1 && n_even_chunks > lanes;
    size_t g = 0;
    for (size_t k = 0; g < n_graphs; ++k)
    {
        size_t size = per_chunk;
        if (shaped && k < lanes)
            size = std::max<size_t>(1, per_chunk * (k + 1) / lanes);
        else if (shaped)
            size = std::min(per_chunk, std::max<size_t>(std::max<size_t>(1, per_chunk / 4), (n_graphs - g) / lanes));
        size = std::min(size, n_graphs - g);
        ranges.emplace_back(g, g + size);
        g += size;
    }
    return ranges;
}
}  // namespace grmpy
