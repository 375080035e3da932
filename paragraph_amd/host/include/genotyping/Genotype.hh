// genotyping::Genotype / GenotypeSet (src/c++/include/genotyping/Genotype.hh:40-121, GenotypeSet.hh; lib/genotyping/Genotype.cpp,
// GenotypeSet.cpp) without the JSON writer.  Host code: genotyping consumes the per-site edge counters the device produced.
#pragma once
#include <cstdint>
#include <set>
#include <string>
#include <vector>

namespace genotyping
{
typedef std::vector<uint64_t> GenotypeVector;

struct Genotype
{
    Genotype() = default;
    explicit Genotype(GenotypeVector rhs) : gt(std::move(rhs)) {}
    GenotypeVector gt;                     // most likely genotype
    std::vector<GenotypeVector> gl_name;   // log-likelihood for each possible genotype
    std::vector<double> gl;
    int gq = -1;                           // -10 log10(1 - gl / sum(gl)), capped at 100
    std::vector<double> allele_fractions;  // fractions of edge count
    double coverage_test_pvalue = -1;      // tail probability of coverage (all alleles)
    int num_reads = 0;
    std::set<std::string> filters;

    void relabel(std::vector<uint64_t> const& new_labels);
    std::string toString(const std::vector<std::string>* p_allele_names = nullptr) const;
    std::string filterString() const;
    explicit operator std::string() const { return toString(); }
};

class GenotypeSet
{
public:
    // remaps the genotype's allele indexes onto the merged allele list; returns the sample index
    size_t add(std::vector<std::string> const& allele_names, Genotype const& gt);
    std::vector<std::string> const& getAlleleNames() const { return merged_allele_names; }
    Genotype const& operator[](size_t i) const { return genotypes[i]; }
    size_t size() const { return genotypes.size(); }
    bool empty() const { return genotypes.empty(); }
    std::vector<Genotype>::const_iterator begin() const { return genotypes.begin(); }
    std::vector<Genotype>::const_iterator end() const { return genotypes.end(); }

private:
    std::vector<Genotype> genotypes;
    std::vector<std::string> merged_allele_names;
};
}  // namespace genotyping
