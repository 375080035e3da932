// Host-side probe without a device: graph loading + packed read extraction of every site of a data set on ONE thread, CPU
// seconds per phase, optional SIGPROF profile (PG_E2E_PROF=<file>).  Not a product program -- a measuring stick for the part of
// the workflow that has to feed the GPUs (src/c++/lib/common/ReadExtraction.cpp:38-219, lib/grmpy/AlignSamples.cpp:115-172).
//   host_probe <reference.fa> <reads.bam> <graphs.txt> [passes]
#include <chrono>
#include <fstream>
#include <iostream>
#include <sys/resource.h>

#include "prof.hh"
#include "common/BamReader.hh"
#include "common/Fasta.hh"
#include "common/Json.hh"
#include "common/Region.hh"
#include "grm/GraphInput.hh"
#include "paragraph/PackedReads.hh"

namespace
{
// what paragraph::GraphDescription::load does (host/src/workflow.cpp), without the device-side half of the library
struct Description
{
    std::list<common::Region> target_regions;
    size_t longest_alt_insertion = 0;
    std::shared_ptr<graphtools::Graph> graph;
    common::Json description;
};
Description loadDescription(std::string const& path, common::FastaFile const& fasta)
{
    Description d;
    common::Json root = common::Json::parseFile(path);
    for (common::Json const& r : root["target_regions"].elements())
        d.target_regions.emplace_back(r.asString());
    for (common::Json const& node : root["nodes"].elements())
        if (node.isMember("sequence"))
            d.longest_alt_insertion = std::max(d.longest_alt_insertion, node["sequence"].asString().size());
    d.graph = std::make_shared<graphtools::Graph>(grm::graphFromJson(root, fasta));
    d.description = std::move(root);
    return d;
}
}  // namespace

static double cpuNow()
{
    struct rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    return (double)ru.ru_utime.tv_sec + 1e-6 * ru.ru_utime.tv_usec + (double)ru.ru_stime.tv_sec + 1e-6 * ru.ru_stime.tv_usec;
}

int main(int argc, char** argv)
{
    if (argc < 4)
        return 2;
    std::vector<std::string> graphs;
    std::ifstream list(argv[3]);
    for (std::string line; std::getline(list, line);)
        if (!line.empty())
            graphs.push_back(line);
    const int passes = argc > 4 ? std::atoi(argv[4]) : 3;
    const char* prof = std::getenv("PG_E2E_PROF");
    const char* only = std::getenv("PG_PROBE_ONLY");  // "load" | "extract"
    if (prof)
        e2eprof::start();
    common::FastaFile fasta(argv[1]);
    for (int p = 0; p < passes; ++p)
    {
        if (prof)
            e2eprof::enable(p > 0);
        const double c0 = cpuNow();
        std::vector<Description> d(graphs.size());
        for (size_t g = 0; g < graphs.size(); ++g)
            d[g] = loadDescription(graphs[g], fasta);
        const double c1 = cpuNow();
        size_t reads = 0, bases = 0;
        if (!only || std::string(only) != "load")
        {
            common::BamReader reader(argv[2], "", argv[1]);
            std::vector<paragraph::PackedSite> sites(graphs.size());
            for (size_t g = 0; g < graphs.size(); ++g)
            {
                paragraph::extractPacked(reader, d[g].target_regions, 10000, (unsigned)d[g].longest_alt_insertion, sites[g]);
                reads += sites[g].size();
                bases += sites[g].bases.size();
            }
        }
        const double c2 = cpuNow();
        printf("{\"pass\": %d, \"sites\": %zu, \"reads\": %zu, \"bases\": %zu, \"load_graphs_cpu_s\": %.3f, \"extract_cpu_s\": %.3f, "
               "\"load_us_per_site\": %.1f, \"extract_us_per_site\": %.1f}\n",
               p, graphs.size(), reads, bases, c1 - c0, c2 - c1, 1e6 * (c1 - c0) / graphs.size(), 1e6 * (c2 - c1) / graphs.size());
    }
    if (prof)
    {
        e2eprof::enable(false);
        e2eprof::dump(prof);
    }
    return 0;
}
