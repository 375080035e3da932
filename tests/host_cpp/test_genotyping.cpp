// CPU-only test program of the host genotyping module (paragraph_amd/host/src/genotyping.cpp).  Re-types the expectations of
//   src/c++/test/test_breakpoint_genotyper.cpp:30-82, test_combined_genotype.cpp:33-159, test_genotype.cpp:28-47,
//   test_genotyping_parameter.cpp:25-49
// With "--dump <seed> <n>" it prints genotyping results for random count vectors as JSON lines (cross-checked against a
// scipy-based restatement by tests/test_genotyping_cpu.py).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "genotyping/BreakpointGenotyper.hh"
#include "genotyping/BreakpointStatistics.hh"
#include "genotyping/CombinedGenotype.hh"
#include "genotyping/GraphBreakpointGenotyper.hh"
#include "genotyping/PopulationStatistics.hh"

using namespace genotyping;
using std::string;
using std::vector;

static int failures = 0;
#define EXPECT_EQ(a, b)                                                                                              \
    do                                                                                                               \
    {                                                                                                                \
        auto va = (a);                                                                                               \
        auto vb = (b);                                                                                               \
        if (!(va == vb))                                                                                             \
        {                                                                                                            \
            std::cerr << __FILE__ << ":" << __LINE__ << ": " #a " != " #b " (" << va << " vs " << vb << ")\n";       \
            ++failures;                                                                                              \
        }                                                                                                            \
    } while (0)
#define EXPECT_NEAR_REL(a, b, rel)                                                                                   \
    do                                                                                                               \
    {                                                                                                                \
        double va = (a), vb = (b);                                                                                   \
        if (std::fabs(va - vb) > (rel) * std::fabs(vb))                                                              \
        {                                                                                                            \
            std::cerr << __FILE__ << ":" << __LINE__ << ": " #a " = " << va << " not within " << (rel) << " of " << vb << "\n"; \
            ++failures;                                                                                              \
        }                                                                                                            \
    } while (0)
#define EXPECT_THROW_ANY(stmt)                                                                                       \
    do                                                                                                               \
    {                                                                                                                \
        bool threw = false;                                                                                          \
        try                                                                                                          \
        {                                                                                                            \
            stmt;                                                                                                    \
        }                                                                                                            \
        catch (...)                                                                                                  \
        {                                                                                                            \
            threw = true;                                                                                            \
        }                                                                                                            \
        if (!threw)                                                                                                  \
        {                                                                                                            \
            std::cerr << __FILE__ << ":" << __LINE__ << ": expected an exception\n";                                \
            ++failures;                                                                                              \
        }                                                                                                            \
    } while (0)

static void testBreakpointGenotyper()
{
    const vector<string> alleles = { "REF", "ALT" };
    {
        auto param = std::unique_ptr<GenotypingParameters>(new GenotypingParameters(alleles, 2));
        BreakpointGenotyper genotyper(param);
        const BreakpointGenotyperParameter b_param(40.0, 100, std::sqrt(40.0 * 5), false);
        EXPECT_THROW_ANY(genotyper.genotype(b_param, {}));
        EXPECT_THROW_ANY(genotyper.genotype(b_param, { 10 }));
    }
    auto param = std::unique_ptr<GenotypingParameters>(new GenotypingParameters(alleles, 2));
    BreakpointGenotyper genotyper(param);
    const BreakpointGenotyperParameter b_param(40.0, 100, 20, false);
    EXPECT_EQ(string("0/0"), (string)genotyper.genotype(b_param, { 20, 0 }));
    EXPECT_EQ(string("0/1"), (string)genotyper.genotype(b_param, { 20, 20 }));
    EXPECT_EQ(string("1/1"), (string)genotyper.genotype(b_param, { 0, 20 }));
    auto haploid_param = std::unique_ptr<GenotypingParameters>(new GenotypingParameters(alleles, 1));
    BreakpointGenotyper haploid_genotyper(haploid_param);
    EXPECT_EQ(string("1"), (string)haploid_genotyper.genotype(b_param, { 0, 20 }));
    EXPECT_NEAR_REL(genotyper.genotype(b_param, { 0, 20 }).coverage_test_pvalue, 0.24825223, 1e-6);
    const BreakpointGenotyperParameter b_poisson_param(40.0, 100, 20, true);
    EXPECT_NEAR_REL(genotyper.genotype(b_poisson_param, { 0, 20 }).coverage_test_pvalue, 0.0080560343, 1e-6);
    const vector<string> alleles2 = { "REF", "ALT1", "ALT2", "ALT3", "ALT4" };
    auto param2 = std::unique_ptr<GenotypingParameters>(new GenotypingParameters(alleles2, 2));
    BreakpointGenotyper genotyper_q(param2);
    EXPECT_EQ(string("1/3"), (string)genotyper_q.genotype(b_param, { 1, 20, 2, 20, 2 }));
    EXPECT_EQ(string("NO_READS"), genotyper.genotype(b_param, { 0, 0 }).filterString());
}

static void testCombinedGenotype()
{
    const vector<string> alleles{ "REF", "ALT" };
    {  // SimplePass
        Genotype gt1;
        gt1.gt = { 1, 1 };
        gt1.gl_name = { { 0, 0 }, { 0, 1 }, { 1, 1 } };
        gt1.gl = { -10, -10, -0.1 };
        GenotypeSet gs;
        for (size_t i = 0; i < 2; i++)
            gs.add(alleles, gt1);
        const Genotype combined_genotype = combinedGenotype(gs);
        EXPECT_EQ(string("1/1"), combined_genotype.toString());
        EXPECT_EQ(string("ALT/ALT"), combined_genotype.toString(&alleles));
    }
    {  // GenotypeUnphasedMatch
        Genotype gt1;
        gt1.gt = { 0, 1 };
        gt1.gl_name = { { 0, 0 }, { 0, 1 }, { 1, 1 } };
        gt1.gl = { -10, -0.1, -10 };
        gt1.gq = 20;
        Genotype gt2;
        gt2.gt = { 1, 0 };
        gt2.gl_name = { { 1, 0 }, { 1, 1 }, { 0, 0 } };
        gt2.gl = { -0.1, -10, -10 };
        gt2.gq = 30;
        GenotypeSet gs;
        gs.add(alleles, gt1);
        gs.add(alleles, gt2);
        const Genotype combined_genotype = combinedGenotype(gs);
        EXPECT_EQ(string("0/1"), combined_genotype.toString());
        EXPECT_EQ(string("PASS"), combined_genotype.filterString());
        EXPECT_EQ(20, combined_genotype.gq);
    }
    {  // GenotypeConflictNoConsensus
        Genotype gt1;
        gt1.gt = { 0, 1 };
        gt1.num_reads = 10;
        gt1.allele_fractions = { 0.5, 0.5 };
        Genotype gt2;
        gt2.gt = { 1, 1 };
        gt2.num_reads = 10;
        gt2.allele_fractions = { 0, 1 };
        GenotypeSet gs;
        gs.add(alleles, gt1);
        gs.add(alleles, gt2);
        auto param = std::unique_ptr<GenotypingParameters>(new GenotypingParameters(alleles, 2));
        BreakpointGenotyper genotyper(param);
        const BreakpointGenotyperParameter b_param(10.0, 100, 50, false);
        const Genotype combined_genotype = combinedGenotype(gs, &b_param, &genotyper);
        EXPECT_EQ(string("0/1"), combined_genotype.toString());
        EXPECT_EQ(string("CONFLICT"), combined_genotype.filterString());
        EXPECT_EQ(8, combined_genotype.gq);
        auto haploid_param = std::unique_ptr<GenotypingParameters>(new GenotypingParameters(alleles, 1));
        BreakpointGenotyper haploid_genotyper(haploid_param);
        Genotype gtX1;
        gtX1.gt = { 0 };
        gtX1.num_reads = 10;
        gtX1.allele_fractions = { 1, 0 };
        Genotype gtX2;
        gtX2.gt = { 1 };
        gtX2.num_reads = 2;
        gtX2.allele_fractions = { 0, 1 };
        GenotypeSet gsX;
        gsX.add(alleles, gtX1);
        gsX.add(alleles, gtX2);
        EXPECT_EQ(string("0"), combinedGenotype(gsX, &b_param, &haploid_genotyper).toString());
    }
    {  // GenotypeMissing
        Genotype gt1;
        Genotype gt2;
        gt2.gt = { 0, 1 };
        gt2.gl_name = { { 0, 1 }, { 1, 1 }, { 0, 0 } };
        gt2.gl = { -1, -10, -10 };
        gt2.gq = 36;
        GenotypeSet gs;
        gs.add(alleles, gt1);
        gs.add(alleles, gt2);
        const Genotype combined_genotype = combinedGenotype(gs);
        EXPECT_EQ(string("0/1"), combined_genotype.toString());
        EXPECT_EQ(string("BP_NO_GT"), combined_genotype.filterString());
        EXPECT_EQ(36, combined_genotype.gq);
    }
}

static void testGenotypeAndParameters()
{
    Genotype variant;
    variant.gt = { 0, 1 };
    variant.gl_name = { { 0, 0 }, { 0, 1 }, { 1, 1 } };
    variant.relabel({ 1, 3 });
    EXPECT_EQ(string("1/3"), variant.toString());
    EXPECT_EQ(variant.gl_name[0][0], 1ull);
    EXPECT_EQ(variant.gl_name[1][1], 3ull);
    EXPECT_EQ(variant.gl_name[2][0], 3ull);
    const vector<string> alleles = { "REF", "ALT1", "ALT2" };
    GenotypingParameters param(alleles);
    param.setAlleleErrorRates({ "ALT1", "REF", "ALT2" }, { 0.1, 0.04, 0.1 });
    param.setHetHaplotypeFractions({ "ALT1", "REF", "ALT2" }, { 0.33, 0.33, 0.33 });
    EXPECT_EQ((int)param.possibleGenotypes().size(), 6);
    EXPECT_EQ(param.alleleErrorRates()[0], 0.04);
    EXPECT_EQ(param.alleleErrorRates()[1], 0.1);
    EXPECT_EQ(param.alleleErrorRates()[2], 0.1);
    // genotype order of setPossibleGenotypes: 0/0, 0/1, 1/1, 0/2, 1/2, 2/2
    EXPECT_EQ(Genotype(param.possibleGenotypes()[1]).toString(), string("0/1"));
    EXPECT_EQ(Genotype(param.possibleGenotypes()[3]).toString(), string("0/2"));
}

// deletion site as graph_templates/shortdeletion builds it: source -> LF -> {MID -> RF | RF} -> sink, edges labelled REF / ALT
static graphtools::Graph deletionGraph()
{
    graphtools::Graph g(5, false);
    const char* names[] = { "source", "LF", "MID", "RF", "sink" };
    const char* seqs[] = { "X", "ACGTACGTAC", "TTTTT", "GGGGGCCCCC", "X" };
    for (int i = 0; i < 5; ++i)
    {
        g.setNodeName(i, names[i]);
        g.setNodeSeq(i, seqs[i]);
    }
    g.addEdge(0, 1);
    g.addEdge(1, 2);
    g.addEdge(1, 3);
    g.addEdge(2, 3);
    g.addEdge(3, 4);
    g.addLabelToEdge(1, 2, "REF");
    g.addLabelToEdge(2, 3, "REF");
    g.addLabelToEdge(1, 3, "ALT");
    return g;
}

static void testGraphBreakpointGenotyper()
{
    graphtools::Graph g = deletionGraph();
    BreakpointMap bm = createBreakpointMap(g);
    EXPECT_EQ(bm.size(), (size_t)2);
    EXPECT_EQ(bm.count("LF_"), (size_t)1);
    EXPECT_EQ(bm.count("_RF"), (size_t)1);
    BreakpointStatistics& bs = bm.at("LF_");
    bs.addCounts({ { "LF_MID", 12 }, { "LF_RF", 7 }, { "MID_RF", 11 } });
    EXPECT_EQ(bs.getCount("REF"), 12);
    EXPECT_EQ(bs.getCount("ALT"), 7);
    EXPECT_EQ(bs.getCount("LF_RF"), 7);
    EXPECT_EQ(bs.getCount("nope"), 0);

    GraphBreakpointGenotyper gg;
    gg.reset(&g);
    EXPECT_EQ(gg.alleleNames().size(), (size_t)2);
    EXPECT_EQ(gg.alleleNames()[0], string("ALT"));  // sorted set of canonical allele names
    gg.addSample("het", { { "LF_MID", 15 }, { "LF_RF", 14 }, { "MID_RF", 16 } }, 30.0, 150, 8.0);
    gg.addSample("homalt", { { "LF_RF", 29 } }, 30.0, 150, 8.0);
    gg.addSample("homref", { { "LF_MID", 27 }, { "MID_RF", 30 } }, 30.0, 150, 8.0);
    gg.addSample("conflict", { { "LF_MID", 14 }, { "LF_RF", 15 }, { "MID_RF", 30 } }, 30.0, 150, 8.0);
    gg.addSample("empty", {}, 30.0, 150, 8.0);
    gg.runGenotyping();
    const vector<string>& an = gg.alleleNames();
    EXPECT_EQ(gg.getGenotype("het", "").toString(&an), string("ALT/REF"));
    EXPECT_EQ(gg.getGenotype("het", "").filterString(), string("PASS"));
    EXPECT_EQ(gg.getGenotype("homalt", "").toString(&an), string("ALT/ALT"));
    EXPECT_EQ(gg.getGenotype("homref", "").toString(&an), string("REF/REF"));
    EXPECT_EQ(gg.getGenotype("homref", "LF_").toString(&an), string("REF/REF"));
    EXPECT_EQ(gg.getGenotype("conflict", "LF_").toString(&an), string("ALT/REF"));
    EXPECT_EQ(gg.getGenotype("conflict", "_RF").toString(&an), string("ALT/REF"));
    EXPECT_EQ(gg.getGenotype("empty", "").filterString(), string("NO_VALID_GT"));
    EXPECT_EQ(gg.getGenotype("empty", "").toString(), string("."));
    auto pl = GraphBreakpointGenotyper::ploidiesForTargetRegions({ "chrX:100-200" });
    EXPECT_EQ(pl.first, 1u);
    EXPECT_EQ(pl.second, 2u);
}

static int dump(unsigned seed, int n)
{
    std::mt19937_64 rng(seed);
    std::printf("[\n");
    for (int it = 0; it < n; ++it)
    {
        const int n_alleles = 2 + (int)(rng() % 3);
        const unsigned ploidy = 1 + (unsigned)(rng() % 2);
        vector<string> alleles;
        for (int a = 0; a < n_alleles; ++a)
            alleles.push_back(a == 0 ? "REF" : "ALT" + std::to_string(a));
        auto param = std::unique_ptr<GenotypingParameters>(new GenotypingParameters(alleles, ploidy));
        BreakpointGenotyper genotyper(param);
        const double depth = 5 + (double)(rng() % 600) / 10.0;
        const int read_length = 100 + (int)(rng() % 151);
        const double sd = 1 + (double)(rng() % 300) / 10.0;
        const bool poisson = (rng() % 2) != 0;
        vector<int32_t> counts;
        for (int a = 0; a < n_alleles; ++a)
            counts.push_back((rng() % 3) == 0 ? 0 : (int32_t)(rng() % (unsigned)(2 * depth + 2)));
        const BreakpointGenotyperParameter bp(depth, read_length, sd, poisson);
        const Genotype g = genotyper.genotype(bp, counts);
        std::printf("{\"ploidy\": %u, \"depth\": %.17g, \"read_length\": %d, \"sd\": %.17g, \"poisson\": %s, \"counts\": [", ploidy, depth,
                    read_length, sd, poisson ? "true" : "false");
        for (size_t a = 0; a < counts.size(); ++a)
            std::printf("%s%d", a ? ", " : "", counts[a]);
        std::printf("], \"gt\": \"%s\", \"gq\": %d, \"filters\": \"%s\", \"pvalue\": %.17g, \"gl\": [", g.toString().c_str(), g.gq,
                    g.filterString().c_str(), g.coverage_test_pvalue);
        for (size_t i = 0; i < g.gl.size(); ++i)
            std::printf("%s%.17g", i ? ", " : "", g.gl[i]);
        std::printf("]}%s\n", it + 1 < n ? "," : "");
    }
    std::printf("]\n");
    return 0;
}

// src/c++/test/test_popstats.cpp:26-95
static void testPopulationStatistics()
{
    GenotypeSet pop;
    const Genotype ref({ 0, 0 }), het({ 0, 1 }), alt({ 1, 1 });
    const vector<string> alleles = { "REF", "ALT" };
    for (int i = 0; i < 83; ++i)
        pop.add(alleles, ref);
    EXPECT_EQ(PopulationStatistics(pop).getChisqPvalue(), 1.0);
    EXPECT_EQ(PopulationStatistics(pop).needFisherExactHWE(), false);
    for (int i = 0; i < 13; ++i)
        pop.add(alleles, het);
    for (int i = 0; i < 4; ++i)
        pop.add(alleles, alt);
    const PopulationStatistics ps1(pop);
    EXPECT_NEAR_REL(ps1.getChisqPvalue(), 0.0020474148859159769, 1e-12);
    EXPECT_NEAR_REL(ps1.getFisherExactPvalue(), 0.010293433548874801, 1e-13);
    EXPECT_EQ(ps1.needFisherExactHWE(), true);
    EXPECT_EQ(ps1.getCallrate(), 1.0);
    EXPECT_NEAR_REL(ps1.getAlleleFrequencies()[1], 21.0 / 200.0, 1e-15);
    const common::Json doc = ps1.toJson();
    EXPECT_NEAR_REL(doc["hwe_fisher"].asDouble(), 0.010293433548874801, 1e-13);
    EXPECT_EQ(doc["allele_frequencies"].size(), (size_t)2);

    const Genotype e02({ 0, 2 }), e12({ 1, 2 }), e22({ 2, 2 });
    const vector<string> multi = { "REF", "ALT1", "ALT2" };
    GenotypeSet pop2;
    const std::pair<const Genotype*, int> groups[] = { { &ref, 24 }, { &het, 31 }, { &alt, 10 }, { &e02, 19 }, { &e12, 11 }, { &e22, 5 } };
    for (auto const& g : groups)
        for (int i = 0; i < g.second; ++i)
            pop2.add(multi, *g.first);
    const PopulationStatistics ps2(pop2);
    EXPECT_NEAR_REL(ps2.getChisqPvalue(), 0.50000945615245529, 1e-12);
    EXPECT_EQ(ps2.needFisherExactHWE(), false);
    EXPECT_EQ(ps2.toJson()["hwe_fisher"].asString(), string(""));
    // no-calls lower the call rate and are left out of every count
    pop2.add({}, Genotype());
    EXPECT_NEAR_REL(PopulationStatistics(pop2).getCallrate(), 100.0 / 101.0, 1e-15);
    EXPECT_NEAR_REL(PopulationStatistics(pop2).getChisqPvalue(), 0.50000945615245529, 1e-12);
}

int main(int argc, char** argv)
{
    if (argc == 4 && string(argv[1]) == "--dump")
        return dump((unsigned)std::atoi(argv[2]), std::atoi(argv[3]));
    try
    {
        testBreakpointGenotyper();
        testCombinedGenotype();
        testGenotypeAndParameters();
        testGraphBreakpointGenotyper();
        testPopulationStatistics();
    }
    catch (std::exception const& e)
    {
        std::cerr << "exception: " << e.what() << "\n";
        return 2;
    }
    if (failures)
    {
        std::cerr << failures << " check(s) failed\n";
        return 1;
    }
    std::cout << "genotyping tests passed\n";
    return 0;
}
