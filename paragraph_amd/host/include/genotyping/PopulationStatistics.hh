// Per-breakpoint population summary of a genotype document's "population" block: Hardy-Weinberg p-values (chi-square with one
// degree of freedom; Wigginton et al. 2005 exact test when two alleles are observed and counts are small), call rate and
// allele frequencies (genotyping::PopulationStatistics, src/c++/include/genotyping/PopulationStatistics.hh:37-88,
// lib/genotyping/PopulationStatistics.cpp:37-327; expectations of src/c++/test/test_popstats.cpp:26-95).
#pragma once
#include <cstdint>
#include <map>
#include <vector>

#include "common/Json.hh"
#include "genotyping/Genotype.hh"

namespace genotyping
{
class PopulationStatistics
{
public:
    explicit PopulationStatistics(GenotypeSet const& genotypes);

    // fraction of samples with a call; per-allele share of all called alleles
    double getCallrate() const { return (double)n_called_ / n_samples_; }
    std::vector<double> getAlleleFrequencies() const;
    std::vector<uint32_t> const& alleleCounts() const { return allele_count_; }
    // Hardy-Weinberg: the chi-square p-value always; the exact one where needFisherExactHWE() says it is called for
    double getChisqPvalue() const;
    bool needFisherExactHWE() const;
    double getFisherExactPvalue() const;
    // { "hwe", "hwe_fisher" ("" when the exact test does not apply), "call_rate", "allele_frequencies" }
    common::Json toJson() const;

private:
    // index of the rarest allele; when some allele has count 0 the scan starts from the commonest one and takes the LAST
    // strictly smaller entry it meets, zeros included (kept as in the original)
    size_t minNonZeroAlleleIndex() const;
    int n_samples_ = 0, n_called_ = 0;
    std::vector<uint32_t> allele_count_;
    std::map<GenotypeVector, int> genotype_count_;
};
}  // namespace genotyping
