// Host-side genotyping from the per-site edge counters: the consumer of the device count
// table.  Re-implements, with the reference's names and semantics,
//   lib/genotyping/Genotype.cpp, GenotypeSet.cpp, GenotypingParameters.cpp:37-84, 198-280, BreakpointGenotyper.cpp:41-255,
//   BreakpointStatistics.cpp:45-176, BreakpointFinder.cpp:49-76, CombinedGenotype.cpp:45-265,
//   GraphGenotyper.cpp:64-86, 378-421, GraphBreakpointGenotyper.cpp:42-115, lib/grmpy/CountAndGenotype.cpp:55-70.
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
    return k * std::log(mean) - mean - std::lgamma((double)k + 1.0);
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
    for (auto& g : gt)
        g = new_labels.at(g);
    std::sort(gt.begin(), gt.end());
    for (auto& l : gl_name)
    {
        for (auto& g : l)
            g = new_labels.at(g);
        std::sort(l.begin(), l.end());
    }
    vector<double> new_allele_fractions(new_labels.size(), 0.0);
    for (size_t g = 0; g < allele_fractions.size(); ++g)
        new_allele_fractions[new_labels.at(g)] = allele_fractions[g];
    allele_fractions = new_allele_fractions;
}

size_t GenotypeSet::add(vector<string> const& allele_names, Genotype const& gt)
{
    genotypes.push_back(gt);
    Genotype& remapped_gt = genotypes.back();
    vector<uint64_t> gt_remapping(allele_names.size());
    size_t j = 0;
    for (auto const& a : allele_names)
    {
        auto a_it = std::find(merged_allele_names.begin(), merged_allele_names.end(), a);
        if (a_it == merged_allele_names.end())
        {
            gt_remapping[j] = merged_allele_names.size();
            merged_allele_names.push_back(a);
        }
        else
            gt_remapping[j] = (uint64_t)(a_it - merged_allele_names.begin());
        ++j;
    }
    remapped_gt.relabel(gt_remapping);
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

void GenotypingParameters::setPossibleGenotypes()
{
    vector<GenotypeVector> gts;
    if (num_alleles)
    {
        const std::function<void(unsigned int, unsigned int, vector<uint64_t>)> makeGenotypes
            = [&makeGenotypes, &gts](unsigned int p, unsigned int n, vector<uint64_t> suffix) {
                  for (unsigned int a = 0; a <= n; ++a)
                  {
                      auto new_suffix = suffix;
                      new_suffix.insert(new_suffix.begin(), a);
                      if (p == 1)
                          gts.push_back(new_suffix);
                      else if (p > 1)
                          makeGenotypes(p - 1, a, new_suffix);
                  }
              };
        makeGenotypes(ploidy_, num_alleles - 1, {});
    }
    possible_genotypes = std::move(gts);
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

Genotype BreakpointGenotyper::genotype(const BreakpointGenotyperParameter& param, const vector<int32_t>& read_counts_per_allele) const
{
    if (read_counts_per_allele.size() != n_alleles_)
        error("Error: number of read counts and alleles mismatches. " + std::to_string(read_counts_per_allele.size()) + " != "
              + std::to_string(n_alleles_) + ".");
    Genotype result;
    // adjusted depth: reads must overlap the breakpoint by min_overlap_bases
    const double multiplier = (param.read_length - (int32_t)min_overlap_bases_) / (double)param.read_length;
    const double lambda = param.read_depth * multiplier;
    const int32_t total_num_reads = std::accumulate(read_counts_per_allele.begin(), read_counts_per_allele.end(), 0);
    if (total_num_reads == 0)
    {
        result.filters.insert("NO_READS");
        return result;
    }
    result.num_reads = total_num_reads;

    double best_gl = -std::numeric_limits<double>::max();
    for (const auto& igt : possible_genotypes)
    {
        const double gl = genotypeLikelihood(lambda, igt, read_counts_per_allele);
        result.gl_name.push_back(igt);
        result.gl.push_back(gl);
        if (gl > best_gl)
        {
            best_gl = gl;
            result.gt = igt;
        }
    }

    double sum_gl = 0;
    for (auto l : result.gl)
        sum_gl += std::exp(l);
    const double pr_gt_error = 1.0 - std::exp(best_gl) / sum_gl;
    if (pr_gt_error == 0)
        result.gq = 100;
    else
    {
        const double gq_log10 = std::log10(pr_gt_error);
        result.gq = gq_log10 < -10 ? 100 : (int)(-10 * gq_log10);
    }
    if (result.gq < min_pass_gq_)
        result.filters.insert("GQ");

    result.allele_fractions.assign(n_alleles_, 0.0);
    for (unsigned int al = 0; al < n_alleles_; ++al)
        result.allele_fractions[al] = ((double)read_counts_per_allele[al]) / total_num_reads;

    double coverage_test_pvalue = param.use_poisson_depth ? poissonCdf(lambda, total_num_reads)
                                                          : normalCdf(lambda, param.depth_sd, total_num_reads);
    if (coverage_test_pvalue > 0.5)
    {
        coverage_test_pvalue = 1 - coverage_test_pvalue;
        if (coverage_test_pvalue < coverage_test_cutoff_.first)
            result.filters.insert("BP_DEPTH");
    }
    else if (coverage_test_pvalue < coverage_test_cutoff_.second)
        result.filters.insert("BP_DEPTH");
    result.coverage_test_pvalue = coverage_test_pvalue;
    return result;
}

double BreakpointGenotyper::genotypeLikelihood(double lambda, const GenotypeVector& gv, const vector<int32_t>& read_counts) const
{
    auto it = genotype_prior_.find(gv);
    const double log_phi = it == genotype_prior_.end() ? 0 : it->second;
    vector<int> allele_ploidy(n_alleles_, 0);
    for (unsigned int al = 0; al < n_alleles_; ++al)
        for (const auto g : gv)
            if (al == g)
                ++allele_ploidy[al];
    double gl = log_phi;
    for (unsigned int al = 0; al < n_alleles_; ++al)
    {
        double mean;
        if (allele_ploidy[al] == 0)  // no copies -> all reads supporting this allele are errors
            mean = lambda * (allele_error_rate_.size() == 1 ? allele_error_rate_[0] : allele_error_rate_[al]);
        else
            mean = lambda * allele_ploidy[al] * (haplotype_read_fraction_.size() == 1 ? haplotype_read_fraction_[0] : haplotype_read_fraction_[al]);
        const double lp = logPoissonPdf(mean, read_counts[al]);
        if (std::exp(lp) == 0)  // the reference takes log(pdf): an underflowing pdf ends the sum
            return -std::numeric_limits<double>::max();
        gl += lp;
        if (std::isinf(gl))
            return -std::numeric_limits<double>::max();
    }
    return gl;
}

// -------------------------------------------------------------------------------------------------- BreakpointStatistics
BreakpointStatistics::BreakpointStatistics(graphtools::Graph const& graph, graphtools::NodeId node_id, bool forward)
{
    const auto& node_name = graph.nodeName(node_id);
    const auto allele_nodes = forward ? graph.successors(node_id) : graph.predecessors(node_id);
    std::map<string, std::set<string>> allele_edge_sets;
    for (auto const& an : allele_nodes)
    {
        const auto& an_name = graph.nodeName(an);
        const string edge_name = forward ? (node_name + "_" + an_name) : (an_name + "_" + node_name);
        edge_names.push_back(edge_name);
        edge_name_to_index[edge_name] = edge_names.size() - 1;
        const auto& edge_labels = forward ? graph.edgeLabels(node_id, an) : graph.edgeLabels(an, node_id);
        for (const auto& allele_name : edge_labels)
        {
            allele_edge_sets[allele_name].insert(edge_name);
            if (std::find(all_allele_names.begin(), all_allele_names.end(), allele_name) == all_allele_names.end())
                all_allele_names.push_back(allele_name);
        }
    }
    // canonical alleles: alleles with the same edge set are one equivalence class, named REF if it contains REF, else by
    // its first member
    std::map<string, std::list<string>> canonical_allele_to_allele;
    for (const auto& allele : allele_edge_sets)
        canonical_allele_to_allele[joinWith(allele.second.begin(), allele.second.end(), ";", [](const string& s) { return s; })]
            .push_back(allele.first);
    for (const auto& canonical_allele : canonical_allele_to_allele)
    {
        const bool has_ref = std::find(canonical_allele.second.begin(), canonical_allele.second.end(), "REF") != canonical_allele.second.end();
        const string canonical_allele_name = has_ref ? string("REF") : canonical_allele.second.front();
        canonical_allele_names.push_back(canonical_allele_name);
        const size_t this_allele_index = canonical_allele_names.size() - 1;
        for (const auto& edge : allele_edge_sets[canonical_allele_name])
            edgename_to_alleles[edge].push_back(this_allele_index);
        for (auto const& noncanonical_allele : canonical_allele.second)
        {
            allele_name_to_index[noncanonical_allele] = this_allele_index;
            allele_name_to_canonical_allele_name[noncanonical_allele] = canonical_allele_name;
        }
    }
}

void BreakpointStatistics::addCounts(std::map<string, int32_t> const& read_counts_by_edge)
{
    for (auto const& edge_name : edge_names)
    {
        const size_t e_index = edge_name_to_index.at(edge_name);
        auto c_it = read_counts_by_edge.find(edge_name);
        const int this_edge_count = c_it == read_counts_by_edge.end() ? 0 : c_it->second;
        if (this_edge_count == 0)
            continue;
        if (edge_counts.size() <= e_index)
            edge_counts.resize(edge_names.size(), 0);
        edge_counts[e_index] += this_edge_count;
        for (const auto& allele : edgename_to_alleles[edge_name])
        {
            if (allele_counts.size() <= allele)
                allele_counts.resize(canonical_allele_names.size(), 0);
            allele_counts[allele] += this_edge_count;
        }
    }
}

int32_t BreakpointStatistics::getCount(string const& edge_or_allele_name) const
{
    const auto e_it = edge_name_to_index.find(edge_or_allele_name);
    const auto a_it = allele_name_to_index.find(edge_or_allele_name);
    if (e_it != edge_name_to_index.end() && a_it != allele_name_to_index.end())
        error("Allele / sequence name " + edge_or_allele_name + " is ambiguous with an edge name.");
    if (e_it != edge_name_to_index.end())
        return e_it->second >= edge_counts.size() ? 0 : edge_counts[e_it->second];
    if (a_it != allele_name_to_index.end())
        return a_it->second >= allele_counts.size() ? 0 : allele_counts[a_it->second];
    return 0;  // unknown edge or allele: not every allele is seen at every breakpoint of a complex site
}

BreakpointMap createBreakpointMap(graphtools::Graph const& wgraph)
{
    BreakpointMap breakpoint_map;
    if (wgraph.numNodes() == 0)
        return breakpoint_map;
    const graphtools::NodeId source_node = 0, sink_node = (graphtools::NodeId)(wgraph.numNodes() - 1);
    const bool has_source_and_sink = wgraph.nodeName(source_node) == "source" && wgraph.nodeName(sink_node) == "sink";
    for (graphtools::NodeId node = source_node; node <= sink_node; ++node)
    {
        if (has_source_and_sink && (node == source_node || node == sink_node))
            continue;
        const string& node_name = wgraph.nodeName(node);
        if (wgraph.successors(node).size() > 1)
            breakpoint_map.emplace(node_name + "_", BreakpointStatistics(wgraph, node, true));
        if (wgraph.predecessors(node).size() > 1)
            breakpoint_map.emplace(string("_") + node_name, BreakpointStatistics(wgraph, node, false));
    }
    return breakpoint_map;
}

// ------------------------------------------------------------------------------------------------------ CombinedGenotype
Genotype combinedGenotype(GenotypeSet const& genotypes, const BreakpointGenotyperParameter* b_param, const BreakpointGenotyper* p_genotyper)
{
    Genotype result;
    const size_t num_pass_genotypes = countUniqGenotypes(genotypes, true);
    if (num_pass_genotypes == 0)
    {
        const auto num_fail_genotypes = (int)countUniqGenotypes(genotypes, false);
        if (num_fail_genotypes == 0)
            result.filters.insert("NO_VALID_GT");
        else if (num_fail_genotypes == 1)
            result = reportConsensusGenotypes(genotypes, false);
        else
            result = genotypeByTotalCounts(genotypes, false, p_genotyper, b_param);
    }
    else if (num_pass_genotypes == 1)
        result = reportConsensusGenotypes(genotypes, true);
    else
        result = genotypeByTotalCounts(genotypes, true, p_genotyper, b_param);
    if (result.filters.empty())
        result.filters.insert("PASS");
    return result;
}

size_t countUniqGenotypes(GenotypeSet const& genotypes, bool pass_only)
{
    std::set<GenotypeVector> voted_gts;
    for (auto& bp : genotypes)
    {
        if (bp.gt.empty())
            continue;
        if (pass_only && !bp.filters.empty())
            continue;
        GenotypeVector sorted_gt = bp.gt;
        std::sort(sorted_gt.begin(), sorted_gt.end());
        voted_gts.insert(sorted_gt);
    }
    return voted_gts.size();
}

Genotype reportConsensusGenotypes(GenotypeSet const& genotypes, bool pass_only)
{
    Genotype result;
    // the reference keys an unordered_map by the "a|b" string of the sorted genotype; the order of the resulting GL list
    // is that container's iteration order there and sorted-by-key here (GL lookups are by name)
    std::map<string, std::pair<GenotypeVector, double>> GLs;
    result.num_reads = 0;
    vector<int> gqs;
    for (auto& bp : genotypes)
    {
        if (bp.gt.empty())
        {
            result.filters.insert("BP_NO_GT");
            continue;
        }
        if (pass_only && !bp.filters.empty())
        {
            result.filters.insert(bp.filters.begin(), bp.filters.end());
            continue;
        }
        if (result.gt.empty())
        {
            GenotypeVector sorted_bp = bp.gt;
            std::sort(sorted_bp.begin(), sorted_bp.end());
            result.gt = sorted_bp;
        }
        result.num_reads += bp.num_reads;
        if (!result.gt.empty())
            gqs.emplace_back(bp.gq);
        if (bp.allele_fractions.size() > result.allele_fractions.size())
            result.allele_fractions.resize(bp.allele_fractions.size(), 0);
        for (size_t i = 0; i < bp.allele_fractions.size(); ++i)
            result.allele_fractions[i] += bp.num_reads * bp.allele_fractions[i];
        for (size_t i = 0; i < bp.gl.size(); ++i)
        {
            auto sorted_gl_name = bp.gl_name[i];
            std::sort(sorted_gl_name.begin(), sorted_gl_name.end());
            const string key = joinWith(sorted_gl_name.begin(), sorted_gl_name.end(), "|", [](uint64_t g) { return std::to_string(g); });
            auto gl_it = GLs.find(key);
            if (gl_it == GLs.end())
                GLs.emplace(key, std::make_pair(sorted_gl_name, bp.gl[i]));
            else
                gl_it->second.second = std::max(gl_it->second.second, bp.gl[i]);
        }
    }
    for (auto& af : result.allele_fractions)
        af /= result.num_reads;
    for (auto const& gl : GLs)
    {
        result.gl.push_back(gl.second.second);
        result.gl_name.push_back(gl.second.first);
    }
    result.gq = gqs.empty() ? 0 : *std::min_element(gqs.begin(), gqs.end());
    return result;
}

Genotype genotypeByTotalCounts(
    GenotypeSet const& genotypes, bool use_pass_only, const BreakpointGenotyper* p_genotyper, const BreakpointGenotyperParameter* b_param)
{
    if (!p_genotyper || !b_param || !(b_param->read_depth > 0) || b_param->read_length <= 0)
        error("genotypeByTotalCounts needs a genotyper and positive depth / read length");
    std::set<string> filters;
    filters.insert("CONFLICT");
    vector<int> sum_counts;
    int num_bp = 0;
    for (auto const& bp : genotypes)
    {
        if (use_pass_only && !bp.filters.empty())
        {
            filters.insert(bp.filters.begin(), bp.filters.end());
            continue;
        }
        if (bp.num_reads == 0)
        {
            filters.insert("BP_NO_GT");
            continue;
        }
        if (sum_counts.empty())
            sum_counts.resize(bp.allele_fractions.size(), 0);
        size_t allele_index = 0;
        for (auto& af : bp.allele_fractions)
        {
            sum_counts.at(allele_index) += (int)std::round(af * bp.num_reads);
            allele_index++;
        }
        num_bp++;
    }
    for (auto& s : sum_counts)
        s = (int)std::round((double)s / num_bp);
    Genotype result = p_genotyper->genotype(*b_param, vector<int32_t>(sum_counts.begin(), sum_counts.end()));
    result.filters = filters;
    return result;
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
    const auto bp_map = createBreakpointMap(*graph);
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
    breakpoint_maps.push_back(createBreakpointMap(*graph));
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
}  // namespace genotyping
