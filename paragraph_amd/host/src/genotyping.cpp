// Host-side genotyping from the per-site edge counters: the consumer of the device count
// table.  Re-implements, with the reference's names and semantics,
//   lib/genotyping/Genotype.cpp, GenotypeSet.cpp, GenotypingParameters.cpp:37-84, 198-280, BreakpointGenotyper.cpp:41-255,
//   BreakpointStatistics.cpp:45-176, BreakpointFinder.cpp:49-76, CombinedGenotype.cpp:45-265,
//   GraphGenotyper.cpp:64-86, 378-421, GraphBreakpointGenotyper.cpp:42-115, PopulationStatistics.cpp:37-327,
//   lib/grmpy/CountAndGenotype.cpp:55-70.
// boost::math's poisson pdf / cdf and normal cdf are written out with lgamma / erfc.  No device code: a few hundred
// floating-point operations per site.
#include <algorithm>
#include <cassert>
#include <cmath>
#include <functional>
#include <limits>
#include <numeric>
#include <set>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

#include "genotyping/BreakpointGenotyper.hh"
#include "genotyping/BreakpointStatistics.hh"
#include "genotyping/CombinedGenotype.hh"
#include "genotyping/Genotype.hh"
#include "genotyping/GenotypingParameters.hh"
#include "genotyping/GraphBreakpointGenotyper.hh"
#include "genotyping/PopulationStatistics.hh"

using std::string;
using std::vector;

namespace genotyping
{
namespace
{
[[noreturn]] void error(const string& msg) { throw std::runtime_error(msg); }

template <typename It, typename F> string joinWith(It b, It e, const char* sep, F f)
{
    string out;
    for (It it = b; it != e; ++it)
    {
        if (it != b)
            out += sep;
        out += f(*it);
    }
    return out;
}

// log of boost::math::pdf(poisson_distribution<>(mean), k)
double logPoissonPdf(double mean, int32_t k)
{
    if (!(mean > 0))
        throw std::domain_error("Poisson mean must be > 0");
    if (k < 0)
        throw std::domain_error("Poisson count must be >= 0");
    int sign = 0;  // lgamma_r: plain lgamma writes the process-wide `signgam`, and genotyping runs on several threads
    return k * std::log(mean) - mean - lgamma_r((double)k + 1.0, &sign);
}

// boost::math::cdf(poisson_distribution<>(mean), k) = Q(k + 1, mean)
double poissonCdf(double mean, int32_t k)
{
    double sum = 0;
    for (int32_t i = 0; i <= k; ++i)
        sum += std::exp(logPoissonPdf(mean, i));
    return sum > 1 ? 1 : sum;
}

// boost::math::cdf(normal_distribution<>(mean, sd), x)
double normalCdf(double mean, double sd, double x)
{
    if (!(sd > 0))
        throw std::domain_error("normal distribution needs sd > 0");
    return 0.5 * std::erfc(-(x - mean) / (sd * std::sqrt(2.0)));
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------- Genotype
string Genotype::toString(const vector<string>* p_allele_names) const
{
    if (gt.empty())
        return ".";
    if (!p_allele_names)
        return joinWith(gt.begin(), gt.end(), "/", [](uint64_t g) { return std::to_string(g); });
    return joinWith(gt.begin(), gt.end(), "/", [&](uint64_t g) { return (*p_allele_names)[g]; });
}

string Genotype::filterString() const
{
    return joinWith(filters.begin(), filters.end(), ";", [](const string& s) { return s; });
}

void Genotype::relabel(vector<uint64_t> const& new_labels)
{
    auto remap = [&](GenotypeVector& v) {
        for (uint64_t& allele : v)
            allele = new_labels.at(allele);
        std::sort(v.begin(), v.end());
    };
    remap(gt);
    for (GenotypeVector& name : gl_name)
        remap(name);
    vector<double> moved(new_labels.size(), 0.0);
    for (size_t old = 0; old < allele_fractions.size(); ++old)
        moved[new_labels.at(old)] = allele_fractions[old];
    allele_fractions.swap(moved);
}

size_t GenotypeSet::add(vector<string> const& allele_names, Genotype const& gt)
{
    // index of every incoming allele in the merged list (appending the ones not seen before)
    vector<uint64_t> to_merged;
    to_merged.reserve(allele_names.size());
    for (const string& name : allele_names)
    {
        size_t at = 0;
        while (at < merged_allele_names.size() && merged_allele_names[at] != name)
            ++at;
        if (at == merged_allele_names.size())
            merged_allele_names.push_back(name);
        to_merged.push_back(at);
    }
    genotypes.push_back(gt);
    genotypes.back().relabel(to_merged);
    return genotypes.size() - 1;
}

// ------------------------------------------------------------------------------------------------- GenotypingParameters
GenotypingParameters::GenotypingParameters(const vector<string>& _allele_names, unsigned int ploidy)
    : ploidy_(ploidy), num_alleles((unsigned int)_allele_names.size()), coverage_test_cutoff(0.02, 0.0001), min_pass_gq(10),
      allele_names(_allele_names), min_overlap_bases(16), reference_allele("REF"), reference_allele_error_rate(0.05),
      other_allele_error_rate(0.05), other_het_haplotype_fraction(0.5), other_genotype_fraction(1), use_poisson_depth(false)
{
    setPossibleGenotypes();
}

// All unordered genotypes (non-decreasing allele tuples) of the given ploidy, in the order the likelihood loop visits them:
// colexicographic -- 0/0, 0/1, 1/1, 0/2, 1/2, 2/2, ... (the first best likelihood wins a tie, so the order matters).
static void enumerateGenotypes(unsigned slots_left, unsigned max_allele, GenotypeVector& tail, vector<GenotypeVector>& out)
{
    for (unsigned allele = 0; allele <= max_allele; ++allele)
    {
        tail.insert(tail.begin(), allele);
        if (slots_left == 1)
            out.push_back(tail);
        else
            enumerateGenotypes(slots_left - 1, allele, tail, out);
        tail.erase(tail.begin());
    }
}

void GenotypingParameters::setPossibleGenotypes()
{
    possible_genotypes.clear();
    if (num_alleles == 0 || ploidy_ == 0)
        return;
    GenotypeVector tail;
    enumerateGenotypes(ploidy_, num_alleles - 1, tail, possible_genotypes);
}

vector<int> GenotypingParameters::alleleNameConversionIndex(const vector<string>& names) const
{
    vector<int> conversion_index;
    for (auto const& key : names)
    {
        auto it = std::find(allele_names.begin(), allele_names.end(), key);
        conversion_index.push_back(it != allele_names.end() ? (int)(it - allele_names.begin()) : -1);
    }
    return conversion_index;
}

void GenotypingParameters::setAlleleErrorRates(const vector<string>& names, const vector<double>& values)
{
    const auto conversion_index = alleleNameConversionIndex(names);
    if (std::all_of(conversion_index.begin(), conversion_index.end(), [](int x) { return x == -1; }))
        return;  // "None of the allele names ... match those in graph": parameters are not used
    allele_error_rates.assign(num_alleles, other_allele_error_rate);
    auto it_ref = std::find(allele_names.begin(), allele_names.end(), reference_allele);
    if (it_ref != allele_names.end())
        allele_error_rates[(size_t)(it_ref - allele_names.begin())] = reference_allele_error_rate;
    for (size_t index = 0; index < values.size() && index < conversion_index.size(); ++index)
        if (conversion_index[index] != -1)
            allele_error_rates[(size_t)conversion_index[index]] = values[index];
}

void GenotypingParameters::setHetHaplotypeFractions(const vector<string>& names, const vector<double>& values)
{
    const auto conversion_index = alleleNameConversionIndex(names);
    if (std::all_of(conversion_index.begin(), conversion_index.end(), [](int x) { return x == -1; }))
        return;
    het_haplotype_fractions.assign(num_alleles, other_het_haplotype_fraction);
    for (size_t index = 0; index < values.size() && index < conversion_index.size(); ++index)
        if (conversion_index[index] != -1)
            het_haplotype_fractions[(size_t)conversion_index[index]] = values[index];
}

void GenotypingParameters::setGenotypeFractions(const vector<string>& names, const std::map<string, double>& fractions)
{
    const auto conversion_index = alleleNameConversionIndex(names);
    if (std::all_of(conversion_index.begin(), conversion_index.end(), [](int x) { return x == -1; }))
        return;
    for (auto const& kv : fractions)
    {
        // genotypeVectorFromString (Genotype.cpp:44-58) only takes the numbers that are FOLLOWED by a '/': "0/1" parses as
        // {0}, "0/1/" as {0, 1}.  Kept as is -- a key only takes effect when it yields `ploidy` alleles.
        GenotypeVector gv;
        size_t prev_pos = 0, current_pos = kv.first.find('/');
        while (current_pos != string::npos)
        {
            gv.push_back((uint64_t)std::stoi(kv.first.substr(prev_pos, current_pos - prev_pos)));
            prev_pos = current_pos + 1;
            current_pos = kv.first.find('/', prev_pos);
        }
        if (gv.empty())
            error("Error: Empty or illegal genotype in parameter JSON: " + kv.first);
        GenotypeVector new_gt;
        for (auto g : gv)
        {
            if (g >= conversion_index.size() || conversion_index[g] == -1)
                break;
            new_gt.push_back((uint64_t)conversion_index[g]);
        }
        if (new_gt.size() == ploidy_)
            genotype_fractions[new_gt] = kv.second;
    }
    for (auto& gt : possible_genotypes)
        if (genotype_fractions.find(gt) == genotype_fractions.end())
            genotype_fractions[gt] = other_genotype_fraction;
}

void GenotypingParameters::setFromJson(common::Json const& doc)
{
    if (doc.isMember("min_overlap_bases"))
        min_overlap_bases = (unsigned)doc["min_overlap_bases"].asUInt64();
    if (doc.isMember("reference_allele"))
        reference_allele = doc["reference_allele"].asString();
    if (doc.isMember("reference_allele_error_rate"))
        reference_allele_error_rate = doc["reference_allele_error_rate"].asDouble();
    if (doc.isMember("other_allele_error_rate"))
        other_allele_error_rate = doc["other_allele_error_rate"].asDouble();
    if (doc.isMember("other_genotype_fraction"))
        other_genotype_fraction = doc["other_genotype_fraction"].asDouble();
    if (doc.isMember("ploidy"))
        ploidy_ = (unsigned)doc["ploidy"].asInt64();
    // "het_haplotype_fraction" (singular) only acts when its text form starts with '[', which no JSON number does
    if (doc.isMember("coverage_test_cutoff"))
    {
        common::Json const& cut = doc["coverage_test_cutoff"];
        if (!cut.isArray() || cut.size() != 2)
            error("Error: coverage_test_cutoff needs to be a list of 2 values: lower end, upper end.");
        coverage_test_cutoff.first = cut[(size_t)0].asDouble();
        coverage_test_cutoff.first = cut[(size_t)1].asDouble();
    }
    const bool per_allele = doc.isMember("allele_error_rates") || doc.isMember("het_haplotype_fractions") || doc.isMember("genotype_fractions");
    if (per_allele)
    {
        if (!doc.isMember("allele_names"))
            error("Error: with allele_error_rates/het_haplotype_fractions/genotype_fractions specified in JSON, allele_names must be "
                  "specified as well.");
        vector<string> names;
        for (auto const& n : doc["allele_names"].elements())
            names.push_back(n.asString());
        auto reals = [](common::Json const& arr) {
            vector<double> out;
            for (auto const& v : arr.elements())
                out.push_back(v.asDouble());
            return out;
        };
        if (doc.isMember("allele_error_rates"))
            setAlleleErrorRates(names, reals(doc["allele_error_rates"]));
        if (doc.isMember("het_haplotype_fractions"))
            setHetHaplotypeFractions(names, reals(doc["het_haplotype_fractions"]));
        if (doc.isMember("genotype_fractions"))
        {
            std::map<string, double> fractions;
            for (auto const& kv : doc["genotype_fractions"].members())
                fractions[kv.first] = kv.second.asDouble();
            setGenotypeFractions(names, fractions);
        }
    }
    if (doc.isMember("use_poisson_depth"))
    {
        common::Json const& flag = doc["use_poisson_depth"];
        if (flag.isString() && flag.asString() == "true")
            use_poisson_depth = true;
        else if (flag.isString() && flag.asString() == "false")
            use_poisson_depth = false;
        else
            error("In genotyping parameter JSON use_poisson_depth only allows true or false.");
    }
}

// --------------------------------------------------------------------------------------------------- BreakpointGenotyper
BreakpointGenotyper::BreakpointGenotyper(std::unique_ptr<GenotypingParameters> const& param)
    : n_alleles_(param->numAlleles()), ploidy_(param->ploidy()), coverage_test_cutoff_(param->coverageTestCutoff()),
      min_pass_gq_(param->minPassGQ()), min_overlap_bases_(param->minOverlapBases()), possible_genotypes(param->possibleGenotypes())
{
    if (param->alleleErrorRates().empty())
        allele_error_rate_.push_back(param->otherAlleleErrorRate());
    else
        allele_error_rate_ = param->alleleErrorRates();
    if (param->hetHaplotypeFractions().empty())
        haplotype_read_fraction_.push_back(param->otherHetHaplotypeFraction());
    else
        haplotype_read_fraction_ = param->hetHaplotypeFractions();
    if (!param->genotypeFractions().empty())
    {
        genotype_prior_ = param->genotypeFractions();
        for (auto& phi : genotype_prior_)
        {
            if (phi.first.size() < ploidy_)
                error("Error: genotype and ploidy does not match.");
            if (phi.second < 0 || phi.second > 1)
                error("Error: genotype prior should be between 0~1.");
            phi.second = std::log(phi.second);
        }
    }
}

// One breakpoint, one sample.  Model: the reads supporting allele a are Poisson with mean lambda * (copies of a in the genotype)
// * het fraction, or lambda * error rate when the genotype has no copy of a; lambda = depth scaled by the fraction of read
// start positions that give at least min_overlap_bases across the breakpoint.
Genotype BreakpointGenotyper::genotype(const BreakpointGenotyperParameter& param, const vector<int32_t>& read_counts_per_allele) const
{
    if (read_counts_per_allele.size() != n_alleles_)
        error("Error: number of read counts and alleles mismatches. " + std::to_string(read_counts_per_allele.size()) + " != "
              + std::to_string(n_alleles_) + ".");
    Genotype call;
    const double lambda = param.read_depth * ((param.read_length - (int32_t)min_overlap_bases_) / (double)param.read_length);
    int32_t n_reads = 0;
    for (int32_t c : read_counts_per_allele)
        n_reads += c;
    if (n_reads == 0)
    {
        call.filters.insert("NO_READS");
        return call;
    }
    call.num_reads = n_reads;

    // likelihood of every candidate genotype; the first maximum is the call
    double top = -std::numeric_limits<double>::max(), likelihood_mass = 0;
    for (const GenotypeVector& candidate : possible_genotypes)
    {
        const double gl = genotypeLikelihood(lambda, candidate, read_counts_per_allele);
        call.gl_name.push_back(candidate);
        call.gl.push_back(gl);
        likelihood_mass += std::exp(gl);
        if (gl > top)
        {
            top = gl;
            call.gt = candidate;
        }
    }
    // GQ = phred of the posterior error under a flat prior over the candidates, truncated, capped at 100
    const double p_wrong = 1.0 - std::exp(top) / likelihood_mass;
    call.gq = 100;
    if (p_wrong != 0)
    {
        const double lg = std::log10(p_wrong);
        if (lg >= -10)
            call.gq = (int)(-10 * lg);
    }
    if (call.gq < min_pass_gq_)
        call.filters.insert("GQ");

    for (int32_t c : read_counts_per_allele)
        call.allele_fractions.push_back((double)c / n_reads);

    // two-sided depth test: is the total read count plausible for this depth?
    double tail = param.use_poisson_depth ? poissonCdf(lambda, n_reads) : normalCdf(lambda, param.depth_sd, n_reads);
    const bool upper = tail > 0.5;
    if (upper)
        tail = 1 - tail;
    if (tail < (upper ? coverage_test_cutoff_.first : coverage_test_cutoff_.second))
        call.filters.insert("BP_DEPTH");
    call.coverage_test_pvalue = tail;
    return call;
}

double BreakpointGenotyper::genotypeLikelihood(double lambda, const GenotypeVector& gv, const vector<int32_t>& read_counts) const
{
    auto prior = genotype_prior_.find(gv);
    double gl = prior == genotype_prior_.end() ? 0.0 : prior->second;
    for (unsigned int al = 0; al < n_alleles_; ++al)
    {
        const int copies = (int)std::count(gv.begin(), gv.end(), (uint64_t)al);
        const double rate = copies == 0 ? (allele_error_rate_.size() == 1 ? allele_error_rate_[0] : allele_error_rate_[al])
                                        : copies * (haplotype_read_fraction_.size() == 1 ? haplotype_read_fraction_[0] : haplotype_read_fraction_[al]);
        const double lp = logPoissonPdf(lambda * rate, read_counts[al]);
        // the reference takes log(pdf): a pdf that underflows to zero (or an infinite sum) ends the evaluation
        if (std::exp(lp) == 0 || std::isinf(gl + lp))
            return -std::numeric_limits<double>::max();
        gl += lp;
    }
    return gl;
}

// -------------------------------------------------------------------------------------------------- BreakpointStatistics
// A breakpoint = one node and the edges leaving it (forward) or entering it (backward).  Alleles are the labels on those
// edges; two alleles that use exactly the same edges cannot be told apart here and are merged into one "canonical" allele
// (named REF when REF is among them, otherwise after the alphabetically first member).
BreakpointStatistics::BreakpointStatistics(graphtools::Graph const& graph, graphtools::NodeId node_id, bool forward)
{
    const string& here = graph.nodeName(node_id);
    std::map<string, std::set<string>> edges_of_allele;  // ordered by allele name
    for (graphtools::NodeId other : (forward ? graph.successors(node_id) : graph.predecessors(node_id)))
    {
        const string& there = graph.nodeName(other);
        const string edge = forward ? here + "_" + there : there + "_" + here;
        edge_name_to_index[edge] = edge_names.size();
        edge_names.push_back(edge);
        for (const string& label : (forward ? graph.edgeLabels(node_id, other) : graph.edgeLabels(other, node_id)))
        {
            edges_of_allele[label].insert(edge);
            if (std::find(all_allele_names.begin(), all_allele_names.end(), label) == all_allele_names.end())
                all_allele_names.push_back(label);
        }
    }
    // group alleles by their edge set; groups come out ordered by the ";"-joined edge names
    std::map<string, vector<string>> groups;
    for (auto const& ae : edges_of_allele)
    {
        string key;
        for (const string& e : ae.second)
            key += (key.empty() ? "" : ";") + e;
        groups[key].push_back(ae.first);
    }
    for (auto const& grp : groups)
    {
        const vector<string>& members = grp.second;
        const bool holds_ref = std::find(members.begin(), members.end(), "REF") != members.end();
        const string name = holds_ref ? "REF" : members.front();
        const size_t index = canonical_allele_names.size();
        canonical_allele_names.push_back(name);
        for (const string& e : edges_of_allele[name])
            edgename_to_alleles[e].push_back(index);
        for (const string& m : members)
        {
            allele_name_to_index[m] = index;
            allele_name_to_canonical_allele_name[m] = name;
        }
    }
}

void BreakpointStatistics::addCounts(std::map<string, int32_t> const& read_counts_by_edge)
{
    for (size_t e = 0; e < edge_names.size(); ++e)
    {
        auto hit = read_counts_by_edge.find(edge_names[e]);
        if (hit == read_counts_by_edge.end() || hit->second == 0)
            continue;
        // (the count vectors stay empty until the first non-zero count arrives, as in the reference)
        if (edge_counts.empty())
            edge_counts.assign(edge_names.size(), 0);
        edge_counts[e] += hit->second;
        for (size_t allele : edgename_to_alleles[edge_names[e]])
        {
            if (allele_counts.empty())
                allele_counts.assign(canonical_allele_names.size(), 0);
            allele_counts[allele] += hit->second;
        }
    }
}

int32_t BreakpointStatistics::getCount(string const& name) const
{
    const auto as_edge = edge_name_to_index.find(name);
    const auto as_allele = allele_name_to_index.find(name);
    const bool is_edge = as_edge != edge_name_to_index.end(), is_allele = as_allele != allele_name_to_index.end();
    if (is_edge && is_allele)
        error("Allele / sequence name " + name + " is ambiguous with an edge name.");
    if (is_edge)
        return as_edge->second < edge_counts.size() ? edge_counts[as_edge->second] : 0;
    if (is_allele)
        return as_allele->second < allele_counts.size() ? allele_counts[as_allele->second] : 0;
    return 0;  // not every allele of a complex site is seen at every breakpoint
}

BreakpointMap createBreakpointMap(graphtools::Graph const& graph)
{
    BreakpointMap out;
    const size_t n = graph.numNodes();
    if (n == 0)
        return out;
    // the artificial "source" / "sink" nodes of converted graphs are not breakpoints
    const bool framed = graph.nodeName(0) == "source" && graph.nodeName((graphtools::NodeId)(n - 1)) == "sink";
    for (size_t i = 0; i < n; ++i)
    {
        if (framed && (i == 0 || i + 1 == n))
            continue;
        const graphtools::NodeId node = (graphtools::NodeId)i;
        if (graph.successors(node).size() > 1)
            out.emplace(graph.nodeName(node) + "_", BreakpointStatistics(graph, node, true));
        if (graph.predecessors(node).size() > 1)
            out.emplace("_" + graph.nodeName(node), BreakpointStatistics(graph, node, false));
    }
    return out;
}

// ------------------------------------------------------------------------------------------------------ CombinedGenotype
namespace
{
GenotypeVector sortedCopy(GenotypeVector v)
{
    std::sort(v.begin(), v.end());
    return v;
}

bool usable(const Genotype& bp, bool pass_only) { return !bp.gt.empty() && !(pass_only && !bp.filters.empty()); }
}  // namespace

size_t countUniqGenotypes(GenotypeSet const& genotypes, bool pass_only)
{
    std::set<GenotypeVector> distinct;
    for (const Genotype& bp : genotypes)
        if (usable(bp, pass_only))
            distinct.insert(sortedCopy(bp.gt));
    return distinct.size();
}

// Site genotype from the breakpoint genotypes: when the (passing, else all) breakpoints agree, report that genotype with the
// pooled evidence; when they disagree, genotype the mean allele counts again and flag CONFLICT.
Genotype combinedGenotype(GenotypeSet const& genotypes, const BreakpointGenotyperParameter* b_param, const BreakpointGenotyper* p_genotyper)
{
    Genotype site;
    bool pass_only = true;
    size_t distinct = countUniqGenotypes(genotypes, true);
    if (distinct == 0)
    {
        pass_only = false;
        distinct = countUniqGenotypes(genotypes, false);
    }
    if (distinct == 0)
        site.filters.insert("NO_VALID_GT");
    else if (distinct == 1)
        site = reportConsensusGenotypes(genotypes, pass_only);
    else
        site = genotypeByTotalCounts(genotypes, pass_only, p_genotyper, b_param);
    if (site.filters.empty())
        site.filters.insert("PASS");
    return site;
}

Genotype reportConsensusGenotypes(GenotypeSet const& genotypes, bool pass_only)
{
    Genotype site;
    // best likelihood seen for every (sorted) genotype; the reference keeps them in an unordered_map keyed by "a|b", so the
    // order of the resulting GL list is unspecified there -- here it is sorted by genotype
    std::map<GenotypeVector, double> best_gl;
    bool any = false;
    int lowest_gq = 0;
    for (const Genotype& bp : genotypes)
    {
        if (bp.gt.empty())
        {
            site.filters.insert("BP_NO_GT");
            continue;
        }
        if (pass_only && !bp.filters.empty())
        {
            site.filters.insert(bp.filters.begin(), bp.filters.end());
            continue;
        }
        if (site.gt.empty())
            site.gt = sortedCopy(bp.gt);
        lowest_gq = any ? std::min(lowest_gq, bp.gq) : bp.gq;
        any = true;
        site.num_reads += bp.num_reads;
        if (site.allele_fractions.size() < bp.allele_fractions.size())
            site.allele_fractions.resize(bp.allele_fractions.size(), 0.0);
        for (size_t al = 0; al < bp.allele_fractions.size(); ++al)
            site.allele_fractions[al] += bp.num_reads * bp.allele_fractions[al];  // read-weighted mean, divided below
        for (size_t i = 0; i < bp.gl.size(); ++i)
        {
            const GenotypeVector key = sortedCopy(bp.gl_name[i]);
            auto slot = best_gl.find(key);
            if (slot == best_gl.end())
                best_gl.emplace(key, bp.gl[i]);
            else if (bp.gl[i] > slot->second)
                slot->second = bp.gl[i];
        }
    }
    for (double& af : site.allele_fractions)
        af /= site.num_reads;
    for (auto const& kv : best_gl)
    {
        site.gl_name.push_back(kv.first);
        site.gl.push_back(kv.second);
    }
    site.gq = any ? lowest_gq : 0;
    return site;
}

Genotype genotypeByTotalCounts(
    GenotypeSet const& genotypes, bool use_pass_only, const BreakpointGenotyper* p_genotyper, const BreakpointGenotyperParameter* b_param)
{
    if (!p_genotyper || !b_param || !(b_param->read_depth > 0) || b_param->read_length <= 0)
        error("genotypeByTotalCounts needs a genotyper and positive depth / read length");
    std::set<string> flags{ "CONFLICT" };
    vector<int> total;
    int n_used = 0;
    for (const Genotype& bp : genotypes)
    {
        if (use_pass_only && !bp.filters.empty())
        {
            flags.insert(bp.filters.begin(), bp.filters.end());
            continue;
        }
        if (bp.num_reads == 0)
        {
            flags.insert("BP_NO_GT");
            continue;
        }
        if (total.empty())
            total.assign(bp.allele_fractions.size(), 0);
        for (size_t al = 0; al < bp.allele_fractions.size(); ++al)
            total.at(al) += (int)std::round(bp.allele_fractions[al] * bp.num_reads);  // back to per-allele read counts
        ++n_used;
    }
    vector<int32_t> mean_counts;
    for (int t : total)
        mean_counts.push_back((int32_t)std::round((double)t / n_used));
    Genotype site = p_genotyper->genotype(*b_param, mean_counts);
    site.filters = flags;
    return site;
}

// ------------------------------------------------------------------------------------------------ GraphBreakpointGenotyper
std::pair<unsigned, unsigned> GraphBreakpointGenotyper::ploidiesForTargetRegions(vector<string> const& target_regions)
{
    unsigned male_ploidy = 2, female_ploidy = 2;
    for (auto const& t_region : target_regions)
    {
        const string chrom = t_region.substr(0, t_region.find(':'));
        if (chrom == "chrX" || chrom == "X")
            male_ploidy = 1;
        else if (chrom == "chrY" || chrom == "Y")
        {
            male_ploidy = 1;
            female_ploidy = 1;
        }
    }
    return { male_ploidy, female_ploidy };
}

void GraphBreakpointGenotyper::reset(graphtools::Graph const* g)
{
    graph = g;
    allelenames.clear();
    samplenames.clear();
    breakpointnames.clear();
    breakpoint_maps.clear();
    depths.clear();
    depth_sds.clear();
    sexes.clear();
    graph_genotypes.clear();
    // built once per graph: a sample's map is a copy of it (the reference walks the graph again for every sample,
    // GraphBreakpointGenotyper.cpp:62-75 -- the result is the same map)
    breakpoints_of_graph_ = createBreakpointMap(*graph);
    BreakpointMap const& bp_map = breakpoints_of_graph_;
    std::set<string> allele_names;
    for (const auto& bp : bp_map)
    {
        breakpointnames.push_back(bp.first);
        for (auto const& an : bp.second.canonicalAlleleNames())
            allele_names.insert(an);
    }
    allelenames.assign(allele_names.begin(), allele_names.end());
    p_genotype_parameter.reset(new GenotypingParameters(allelenames, female_ploidy_));
    p_male_genotype_parameter.reset(new GenotypingParameters(allelenames, male_ploidy_));
}

void GraphBreakpointGenotyper::addSample(
    string const& sample_name, std::map<string, int32_t> const& read_counts_by_edge, double autosome_depth, int read_length, double depth_sd,
    Sex sex)
{
    if (!graph)
        error("GraphBreakpointGenotyper::reset has not been called");
    samplenames.push_back(sample_name);
    breakpoint_maps.push_back(breakpoints_of_graph_);
    for (auto& breakpoint : breakpoint_maps.back())
        breakpoint.second.addCounts(read_counts_by_edge);
    depths.emplace_back(autosome_depth, read_length);
    depth_sds.emplace_back(depth_sd);
    sexes.emplace_back(sex);
}

unsigned int GraphBreakpointGenotyper::samplePloidy(size_t sample_index) const
{
    return sexes[sample_index] == Sex::MALE ? male_ploidy_ : female_ploidy_;  // unknown is treated as female
}

int32_t GraphBreakpointGenotyper::getCount(size_t sample_index, string const& breakpoint, string const& edge_or_allele_name) const
{
    return breakpoint_maps.at(sample_index).at(breakpoint).getCount(edge_or_allele_name);
}

Genotype GraphBreakpointGenotyper::getGenotype(string const& sample_name, string const& breakpoint_name) const
{
    auto gt_it = graph_genotypes.find(std::make_pair(sample_name, breakpoint_name));
    return gt_it == graph_genotypes.end() ? Genotype() : gt_it->second;
}

void GraphBreakpointGenotyper::runGenotyping()
{
    BreakpointGenotyper genotyper(p_genotype_parameter);
    BreakpointGenotyper male_genotyper(p_male_genotype_parameter);
    for (const auto& breakpointname : breakpointnames)
    {
        for (size_t sample_index = 0; sample_index < samplenames.size(); ++sample_index)
        {
            auto const& depth_readlength = depths[sample_index];
            vector<int32_t> counts;
            for (const auto& e : allelenames)
                counts.push_back(getCount(sample_index, breakpointname, e));
            const unsigned sample_ploidy = samplePloidy(sample_index);
            const double expected_depth = depth_readlength.first * ((double)sample_ploidy / female_ploidy_);
            const BreakpointGenotyperParameter b_param(
                expected_depth, depth_readlength.second, depth_sds[sample_index], p_genotype_parameter->usePoissonDepth());
            const Genotype gt = sample_ploidy == male_ploidy_ ? male_genotyper.genotype(b_param, counts) : genotyper.genotype(b_param, counts);
            graph_genotypes[std::make_pair(samplenames[sample_index], breakpointname)] = gt;
        }
    }
    for (size_t sample_index = 0; sample_index < samplenames.size(); ++sample_index)
    {
        GenotypeSet all_breakpoint_gts;
        for (const auto& breakpointname : breakpointnames)
            all_breakpoint_gts.add(allelenames, getGenotype(samplenames[sample_index], breakpointname));
        auto const& depth_readlength = depths[sample_index];
        const BreakpointGenotyperParameter b_param(
            depth_readlength.first, depth_readlength.second, depth_sds[sample_index], p_genotype_parameter->usePoissonDepth());
        graph_genotypes[std::make_pair(samplenames[sample_index], string(""))] = combinedGenotype(all_breakpoint_gts, &b_param, &genotyper);
    }
}
// --------------------------------------------------------------------------------------------------- PopulationStatistics
PopulationStatistics::PopulationStatistics(GenotypeSet const& genotypes) : n_samples_((int)genotypes.size())
{
    for (Genotype const& g : genotypes)
    {
        if (g.gt.empty())
            continue;
        ++n_called_;
        ++genotype_count_[g.gt];
        for (uint64_t allele : g.gt)
        {
            if (allele_count_.size() <= allele)
                allele_count_.resize(allele + 1, 0);
            ++allele_count_[allele];
        }
    }
}

common::Json PopulationStatistics::toJson() const
{
    common::Json out = common::Json::object();
    out["hwe"] = getChisqPvalue();
    if (needFisherExactHWE())
        out["hwe_fisher"] = getFisherExactPvalue();
    else
        out["hwe_fisher"] = "";
    out["call_rate"] = getCallrate();
    out["allele_frequencies"] = common::Json::array();
    for (double f : getAlleleFrequencies())
        out["allele_frequencies"].append(f);
    return out;
}

double PopulationStatistics::getChisqPvalue() const
{
    const double n = n_called_;
    double chisq = 0;
    for (auto const& gc : genotype_count_)
    {
        if (gc.first.size() != 2)
            continue;
        const uint64_t h1 = gc.first[0], h2 = gc.first[1];
        if (allele_count_[h1] == 0 || allele_count_[h2] == 0)
            continue;
        const double f1 = (double)allele_count_[h1] / n / 2, f2 = (double)allele_count_[h2] / n / 2;
        const double expected = h1 == h2 ? f1 * f1 * n : 2 * f1 * f2 * n;
        const double diff = expected - gc.second;
        chisq += diff * diff / expected;
    }
    // 1 - cdf of chi-square with one degree of freedom
    return std::erfc(std::sqrt(chisq / 2));
}

bool PopulationStatistics::needFisherExactHWE() const
{
    const auto observed = std::count_if(allele_count_.begin(), allele_count_.end(), [](uint32_t a) { return a > 0; });
    if (observed != 2)
        return false;
    if (n_called_ <= 30)
        return true;
    for (auto const& gc : genotype_count_)
        if (gc.second > 0 && gc.second <= 20)
            return true;
    const double maf = (double)allele_count_[minNonZeroAlleleIndex()] / 2 / n_called_;
    return maf * maf * n_called_ <= 20;
}

double PopulationStatistics::getFisherExactPvalue() const
{
    const size_t minor = minNonZeroAlleleIndex();
    const auto major_it = std::max_element(allele_count_.begin(), allele_count_.end());
    const int minor_count = (int)allele_count_[minor], major_count = (int)*major_it;
    GenotypeVector het = { (uint64_t)(major_it - allele_count_.begin()), (uint64_t)minor };
    std::sort(het.begin(), het.end());
    int observed_het = 0;
    for (auto const& gc : genotype_count_)
        if (gc.first.size() == 2 && gc.first[0] == het[0] && gc.first[1] == het[1])
        {
            observed_het = gc.second;
            break;
        }
    const double n = n_called_;
    const int expected_het = (int)std::round(2 * ((double)minor_count / n / 2) * ((double)major_count / n / 2) * n);

    // probabilities of every heterozygote count with the parity of the expectation, relative to the expectation's own;
    // walked upwards, then downwards, by the recurrence of the exact test
    vector<double> scaled = { 1 };
    double observed_scaled = -1;
    int rare_hom = (minor_count - expected_het) / 2, common_hom = n_called_ - rare_hom - expected_het;
    double prev = 1;
    for (int hets = expected_het + 2; hets <= minor_count; hets += 2)
    {
        const int below = hets - 2;
        prev = prev * (4 * rare_hom * common_hom) / ((below + 2) * (below + 1));
        scaled.push_back(prev);
        --rare_hom;
        --common_hom;
        if (observed_scaled == -1 && hets == observed_het)
            observed_scaled = prev;
    }
    rare_hom = (minor_count - expected_het) / 2;
    common_hom = n_called_ - rare_hom - expected_het;
    prev = 1;
    for (int hets = expected_het - 2; hets >= 0; hets -= 2)
    {
        const int above = hets + 2;
        prev = prev / 4 * above / (rare_hom + 1) * (above - 1) / (common_hom + 1);
        scaled.push_back(prev);
        ++rare_hom;
        ++common_hom;
        if (observed_scaled == -1 && hets == observed_het)
            observed_scaled = prev;
    }
    double at_most_observed = 0;
    for (double s : scaled)
        if (s <= observed_scaled)
            at_most_observed += s;
    return at_most_observed / std::accumulate(scaled.begin(), scaled.end(), 0.0);
}

vector<double> PopulationStatistics::getAlleleFrequencies() const
{
    const uint32_t sum = std::accumulate(allele_count_.begin(), allele_count_.end(), (uint32_t)0);
    vector<double> out;
    for (uint32_t ac : allele_count_)
        out.push_back(sum > 0 ? (double)ac / sum : 0.0);
    return out;
}

size_t PopulationStatistics::minNonZeroAlleleIndex() const
{
    auto pick = std::min_element(allele_count_.begin(), allele_count_.end());
    if (*pick > 0)
        return (size_t)(pick - allele_count_.begin());
    pick = std::max_element(allele_count_.begin(), allele_count_.end());
    if (*pick == 0)
        return 0;
    for (auto it = allele_count_.begin(); it != allele_count_.end(); ++it)
        if (*it < *pick)
            pick = it;
    return (size_t)(pick - allele_count_.begin());
}
}  // namespace genotyping
