// genotyping::GraphBreakpointGenotyper (src/c++/include/genotyping/GraphGenotyper.hh, GraphBreakpointGenotyper.hh;
// lib/genotyping/GraphGenotyper.cpp:64-120, 378-421, GraphBreakpointGenotyper.cpp:42-115; lib/grmpy/CountAndGenotype.cpp:46-88):
// per sample the edge counts of one site -> breakpoint genotypes -> the combined site genotype.  Samples are added with their
// read_counts_by_edge (what paragraph::SiteBatcher / the device count table delivers) instead of the alignment JSON.
#pragma once
#include <list>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "genotyping/BreakpointStatistics.hh"
#include "genotyping/CombinedGenotype.hh"
#include "genotyping/GenotypingParameters.hh"
#include "graphcore/Graph.hh"

namespace genotyping
{
enum class Sex { UNKNOWN = 0, MALE = 1, FEMALE = 2 };  // SampleInfo::Sex

class GraphBreakpointGenotyper
{
public:
    explicit GraphBreakpointGenotyper(unsigned int male_ploidy = 2, unsigned int female_ploidy = 2)
        : male_ploidy_(male_ploidy), female_ploidy_(female_ploidy)
    {
    }
    // ploidies as countAndGenotype derives them from the target regions: chrX / X -> male 1; chrY / Y -> both 1
    static std::pair<unsigned, unsigned> ploidiesForTargetRegions(std::vector<std::string> const& target_regions);
    void reset(graphtools::Graph const* graph);
    // default parameters; the returned object can be adjusted before samples are genotyped (female / autosomal one)
    GenotypingParameters& parameters() { return *p_genotype_parameter; }
    void addSample(
        std::string const& sample_name, std::map<std::string, int32_t> const& read_counts_by_edge, double autosome_depth, int read_length,
        double depth_sd = 0, Sex sex = Sex::UNKNOWN);
    void runGenotyping();
    Genotype getGenotype(std::string const& sample_name, std::string const& breakpoint_name) const;  // "" = combined
    int32_t getCount(size_t sample_index, std::string const& breakpoint, std::string const& edge_or_allele_name) const;
    std::vector<std::string> const& alleleNames() const { return allelenames; }
    std::vector<std::string> const& sampleNames() const { return samplenames; }
    std::list<std::string> const& breakpointNames() const { return breakpointnames; }
    // the graph's breakpoints without counts, as reset() built them (every sample starts from a copy)
    BreakpointMap const& breakpointsOfGraph() const { return breakpoints_of_graph_; }

private:
    unsigned int samplePloidy(size_t sample_index) const;
    unsigned int male_ploidy_, female_ploidy_;
    const graphtools::Graph* graph = nullptr;
    std::vector<std::string> allelenames, samplenames;
    std::list<std::string> breakpointnames;
    BreakpointMap breakpoints_of_graph_;
    std::vector<BreakpointMap> breakpoint_maps;
    std::vector<std::pair<double, int>> depths;
    std::vector<double> depth_sds;
    std::vector<Sex> sexes;
    std::map<std::pair<std::string, std::string>, Genotype> graph_genotypes;
    std::unique_ptr<GenotypingParameters> p_genotype_parameter, p_male_genotype_parameter;
};
}  // namespace genotyping
