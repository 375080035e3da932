// genotyping::GenotypingParameters (src/c++/include/genotyping/GenotypingParameters.hh, lib/genotyping/GenotypingParameters.cpp:37-84).
// The reference reads overrides from a JSON file (setFromJson, :86-196); here the same overrides are plain setters with the
// same allele-name remapping.
#pragma once
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "common/Json.hh"
#include "genotyping/Genotype.hh"

namespace genotyping
{
class GenotypingParameters
{
public:
    explicit GenotypingParameters(const std::vector<std::string>& allele_names, unsigned int ploidy = 2);

    unsigned int ploidy() const { return ploidy_; }
    unsigned int numAlleles() const { return num_alleles; }
    unsigned int minOverlapBases() const { return min_overlap_bases; }
    int minPassGQ() const { return min_pass_gq; }
    std::pair<double, double> coverageTestCutoff() const { return coverage_test_cutoff; }
    bool usePoissonDepth() const { return use_poisson_depth; }
    double otherAlleleErrorRate() const { return other_allele_error_rate; }
    double otherHetHaplotypeFraction() const { return other_het_haplotype_fraction; }
    const std::vector<double>& alleleErrorRates() const { return allele_error_rates; }
    const std::vector<double>& hetHaplotypeFractions() const { return het_haplotype_fractions; }
    const std::map<GenotypeVector, double>& genotypeFractions() const { return genotype_fractions; }
    const std::vector<GenotypeVector>& possibleGenotypes() const { return possible_genotypes; }

    // the genotyping-parameter document of grmpy -G (GenotypingParameters::setFromJson, GenotypingParameters.cpp:83-187).
    // Quirks kept: both values of "coverage_test_cutoff" land in the LOWER cutoff, "use_poisson_depth" must be the STRING
    // "true" / "false", "ploidy" does not re-enumerate the possible genotypes, "min_pass_gq" is not read at all.
    void setFromJson(common::Json const& param_json);
    // setFromJson's keys
    void setMinOverlapBases(unsigned int v) { min_overlap_bases = v; }
    void setReferenceAllele(const std::string& v) { reference_allele = v; }
    void setReferenceAlleleErrorRate(double v) { reference_allele_error_rate = v; }
    void setOtherAlleleErrorRate(double v) { other_allele_error_rate = v; }
    void setOtherHetHaplotypeFraction(double v) { other_het_haplotype_fraction = v; }
    void setUsePoissonDepth(bool v) { use_poisson_depth = v; }
    // "allele_names" + "allele_error_rates" / "het_haplotype_fractions" / "genotype_fractions": values are given in the order
    // of `names`; names the graph does not have are ignored (GenotypingParameters.cpp:198-280)
    void setAlleleErrorRates(const std::vector<std::string>& names, const std::vector<double>& values);
    void setHetHaplotypeFractions(const std::vector<std::string>& names, const std::vector<double>& values);
    void setGenotypeFractions(const std::vector<std::string>& names, const std::map<std::string, double>& fractions);

private:
    std::vector<int> alleleNameConversionIndex(const std::vector<std::string>& names) const;
    void setPossibleGenotypes();
    unsigned int ploidy_, num_alleles;
    std::pair<double, double> coverage_test_cutoff;
    int min_pass_gq;
    std::vector<std::string> allele_names;
    unsigned int min_overlap_bases;
    std::string reference_allele;
    double reference_allele_error_rate, other_allele_error_rate, other_het_haplotype_fraction, other_genotype_fraction;
    bool use_poisson_depth;
    std::vector<double> allele_error_rates, het_haplotype_fractions;
    std::map<GenotypeVector, double> genotype_fractions;
    std::vector<GenotypeVector> possible_genotypes;
};
}  // namespace genotyping
