// From the genotypes of a site's breakpoints to the genotype of the site (genotyping::combinedGenotype and helpers,
// src/c++/include/genotyping/CombinedGenotype.hh, lib/genotyping/CombinedGenotype.cpp:45-265): when the passing breakpoint
// genotypes agree that is the call; when they conflict the mean counts are genotyped again and the call carries CONFLICT.
#pragma once
#include <cstddef>

#include "genotyping/BreakpointGenotyper.hh"
#include "genotyping/Genotype.hh"

namespace genotyping
{
// how many different genotypes the set holds (only PASS ones when pass_only)
size_t countUniqGenotypes(GenotypeSet const& breakpoint_genotypes, bool pass_only);
// the one genotype all (passing) members share, with summed reads, averaged fractions and merged filters
Genotype reportConsensusGenotypes(GenotypeSet const& breakpoint_genotypes, bool pass_only);
// re-genotyping of the members' mean allele counts
Genotype genotypeByTotalCounts(
    GenotypeSet const& breakpoint_genotypes, bool use_pass_only, const BreakpointGenotyper* genotyper,
    const BreakpointGenotyperParameter* sample_parameters);
// the site call; without a genotyper / parameters a conflict cannot be resolved and is reported as a no-call with CONFLICT
Genotype combinedGenotype(
    GenotypeSet const& breakpoint_genotypes, const BreakpointGenotyperParameter* sample_parameters = nullptr,
    const BreakpointGenotyper* genotyper = nullptr);
}  // namespace genotyping
