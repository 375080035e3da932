// End-to-end probe: every graph of a list x every sample of a manifest through grmpy::genotypeGraphs (one device batch),
// with the wall-clock split by phase.  Not a product CLI -- a measuring stick for the host side of the workflow.
//   grmpy_batch <reference.fa> <manifest.txt> <graphs.txt> <threads> [genotypes.json] [sites_per_batch] [lanes] [packed 0|1]
#include <chrono>
#include <sys/resource.h>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "prof.hh"
#include "paragraph/SiteBatcher.hh"
#include "paragraph/Workflow.hh"

int main(int argc, char** argv)
{
    if (argc < 5)
    {
        std::cerr << "usage: grmpy_batch <reference.fa> <manifest.txt> <graphs.txt> <threads> [genotypes.json]\n";
        return 2;
    }
    try
    {
        std::vector<std::string> graphs;
        std::ifstream list(argv[3]);
        for (std::string line; std::getline(list, line);)
            if (!line.empty())
                graphs.push_back(line);
        genotyping::Samples samples = genotyping::loadManifest(argv[2]);
        grmpy::Parameters parameters;
        parameters.threads = std::atoi(argv[4]);
        if (argc > 6)
            if (std::atoll(argv[6]) > 0)  // 0: the library's default
                parameters.sites_per_batch = (size_t)std::atoll(argv[6]);
        if (argc > 7)
            parameters.lanes = std::atoi(argv[7]);
        if (argc > 8)
            parameters.packed_reads = std::atoi(argv[8]) != 0;
        common::Json runs = common::Json::array();
        std::vector<common::Json> genotypes;
        const int reps = std::getenv("PG_E2E_REPS") ? std::atoi(std::getenv("PG_E2E_REPS")) : 2;
        const char* prof_path = std::getenv("PG_E2E_PROF");
        if (prof_path)
            e2eprof::start();
        for (int rep = 0; rep < reps; ++rep)  // the first pass pays device start-up and cold file cache
        {
            paragraph::Timings t;
            parameters.timings = &t;
            std::vector<common::Json>().swap(genotypes);  // the previous pass's documents are not part of this one
            if (prof_path)
                e2eprof::enable(rep > 0);
            struct rusage ru0;
            getrusage(RUSAGE_SELF, &ru0);
            const auto t0 = std::chrono::steady_clock::now();
            genotypes = grmpy::genotypeGraphs(parameters, graphs, argv[1], samples, "");
            const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (prof_path)
                e2eprof::enable(false);
            common::Json run = common::Json::object();
            run["total_s"] = total;
            {
                struct rusage ru1;
                getrusage(RUSAGE_SELF, &ru1);
                auto sec = [](timeval const& a, timeval const& b) { return (double)(b.tv_sec - a.tv_sec) + 1e-6 * (double)(b.tv_usec - a.tv_usec); };
                run["cpu_user_s"] = sec(ru0.ru_utime, ru1.ru_utime);
                run["cpu_sys_s"] = sec(ru0.ru_stime, ru1.ru_stime);
                run["minor_faults"] = (uint64_t)(ru1.ru_minflt - ru0.ru_minflt);
            }
            run["load_graphs_s"] = t.load_graphs;
            run["extract_reads_s"] = t.extract_reads;
            run["device_batch_s"] = t.device_batch;
            run["documents_s"] = t.documents;
            run["genotypes_s"] = t.genotypes;
            run["release_s"] = t.release;
            run["batches"] = (uint64_t)t.batches;
            run["lanes"] = (uint64_t)t.lanes;
            run["sites"] = (uint64_t)t.sites;
            run["reads"] = (uint64_t)t.reads;
            run["sites_per_s"] = (double)t.sites / total;
            run["reads_per_s"] = (double)t.reads / total;
            run["pinned_staging_bytes"] = (uint64_t)paragraph::pinnedStagingBytes();
            runs.append(run);
        }
        if (prof_path)
        {
            e2eprof::enable(false);
            e2eprof::dump(prof_path);
        }
        common::Json out = common::Json::object();
        out["threads"] = parameters.threads;
        out["packed_reads"] = parameters.packed_reads;
        out["graphs"] = (uint64_t)graphs.size();
        out["samples"] = (uint64_t)samples.size();
        out["runs"] = runs;
        std::cout << out.dump() << "\n";
        if (argc > 5)
        {
            common::Json all = common::Json::array();
            for (auto const& g : genotypes)
            {
                common::Json brief = common::Json::object();
                brief["ID"] = g["graphinfo"]["ID"];
                for (auto const& kv : g["samples"].members())
                    brief[kv.first] = kv.second["gt"]["GT"];
                all.append(brief);
            }
            std::ofstream(argv[5]) << all.dump() << "\n";
            // and every document in full, for comparing runs (packed against object form) byte for byte
            common::Json full = common::Json::array();
            for (auto const& g : genotypes)
                full.append(g);
            std::ofstream(std::string(argv[5]) + ".full") << full.dump() << "\n";
        }
    }
    catch (std::exception const& e)
    {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
