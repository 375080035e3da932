// genotyping::BreakpointGenotyper (src/c++/include/genotyping/BreakpointGenotyper.hh, lib/genotyping/BreakpointGenotyper.cpp:41-255):
// Poisson model of the per-allele read counts at one breakpoint -> GT, GL, GQ, allele fractions, depth test, filters.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <vector>

#include "genotyping/Genotype.hh"
#include "genotyping/GenotypingParameters.hh"

namespace genotyping
{
struct BreakpointGenotyperParameter
{
    BreakpointGenotyperParameter(double read_depth_, int32_t read_length_, double depth_sd_, bool use_poisson_depth_)
        : read_depth(read_depth_), read_length(read_length_), depth_sd(depth_sd_), use_poisson_depth(use_poisson_depth_)
    {
    }
    double read_depth;
    int32_t read_length;
    double depth_sd;
    bool use_poisson_depth;
};

class BreakpointGenotyper
{
public:
    explicit BreakpointGenotyper(std::unique_ptr<GenotypingParameters> const& param);
    // throws std::runtime_error when the number of counts differs from the number of alleles
    Genotype genotype(const BreakpointGenotyperParameter& param, const std::vector<int32_t>& read_counts_per_allele) const;

private:
    double genotypeLikelihood(double lambda, const GenotypeVector& gv, const std::vector<int32_t>& read_counts) const;
    unsigned int n_alleles_, ploidy_;
    std::pair<double, double> coverage_test_cutoff_;
    int min_pass_gq_;
    unsigned int min_overlap_bases_;
    std::vector<double> allele_error_rate_, haplotype_read_fraction_;
    std::map<GenotypeVector, double> genotype_prior_;
    std::vector<GenotypeVector> possible_genotypes;
};
}  // namespace genotyping
