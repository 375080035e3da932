// genotyping::combinedGenotype (src/c++/include/genotyping/CombinedGenotype.hh, lib/genotyping/CombinedGenotype.cpp:45-265): the
// site genotype from its breakpoint genotypes -- consensus of the passing ones, else re-genotyping on the mean counts (CONFLICT).
#pragma once
#include "genotyping/BreakpointGenotyper.hh"
#include "genotyping/Genotype.hh"

namespace genotyping
{
Genotype combinedGenotype(
    GenotypeSet const& genotypes, const BreakpointGenotyperParameter* b_param = nullptr, const BreakpointGenotyper* p_genotyper = nullptr);
size_t countUniqGenotypes(GenotypeSet const& genotypes, bool pass_only);
Genotype reportConsensusGenotypes(GenotypeSet const& genotypes, bool pass_only);
Genotype genotypeByTotalCounts(
    GenotypeSet const& genotypes, bool use_pass_only, const BreakpointGenotyper* p_genotyper, const BreakpointGenotyperParameter* b_param);
}  // namespace genotyping
