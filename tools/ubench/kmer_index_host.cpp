// Times the host half of pg_build_kmer_index (no device): N synthetic e2e-like graphs (source X, LF 150, REF <= 300, INS <= 120, RF 150,
// sink X), k = 32.  Build: hipcc -O3 -std=c++17 -I paragraph_amd/csrc -o /tmp/kmer_index_host tools/ubench/kmer_index_host.cpp -L paragraph_amd -lparagraph_amd -Wl,-rpath,$PWD/paragraph_amd
#include <chrono>
#include <cstdio>
#include <random>

#include "pg_internal.h"
#include "pg_kmerindex.h"

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 192;
    std::mt19937_64 rng(7);
    pg_graphs G;
    G.n_graphs = n;
    G.h_node_off.push_back(0);
    G.h_pred_off.push_back(0);
    G.h_nodeseq_off.push_back(0);
    auto seq = [&](uint32_t len) {
        std::string s(len, 'A');
        for (auto& c : s)
            c = "ACGT"[rng() & 3];
        return s;
    };
    for (uint32_t g = 0; g < n; ++g)
    {
        // nodes: 0 source, 1 LF, 2 REF, 3 INS, 4 RF, 5 sink; edges 0-1, 1-2, 1-3, 2-4, 3-4, 4-5
        const std::string nodes[6] = { "X", seq(150), seq(20 + rng() % 280), seq(10 + rng() % 110), seq(150), "X" };
        const std::vector<std::vector<uint32_t>> preds = { {}, { 0 }, { 1 }, { 1 }, { 2, 3 }, { 4 } };
        for (int i = 0; i < 6; ++i)
        {
            G.h_seq_raw += nodes[i];
            G.h_nodeseq_off.push_back((uint32_t)G.h_seq_raw.size());
            G.h_node_len.push_back((uint32_t)nodes[i].size());
            for (uint32_t p : preds[i])
                G.h_pred.push_back(p);
            G.h_pred_off.push_back((uint32_t)G.h_pred.size());
        }
        G.h_node_off.push_back((uint32_t)G.h_node_len.size());
    }
    const std::vector<int32_t> k(n, 32);
    double best = 1e9;
    size_t entries = 0;
    const bool fresh = argc > 2 && atoi(argv[2]) != 0;  // 1: fresh tables every call (page faults + regrowth), 0: the thread's own, kept
    PgKmerIndexHost kept;
    for (int rep = 0; rep < 20; ++rep)
    {
        PgKmerIndexHost fresh_tables;
        PgKmerIndexHost& t = fresh ? fresh_tables : kept;
        const auto t0 = std::chrono::steady_clock::now();
        const char* why = pg_build_kmer_index_host(&G, k, t, false);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (why)
        {
            printf("failed: %s\n", why);
            return 1;
        }
        best = std::min(best, s);
        entries = t.table.size();
    }
    // (a checksum of everything the device gets: two builds of the library that print the same one made the same tables)
    uint64_t sum = 0;
    for (auto const& e : kept.table)
        sum = sum * 1000003u + e.hash + e.count * 7u + e.start_pos * 11u + e.end_pos * 13u + e.n_nodes * 17u + e.pool_off * 19u;
    for (uint32_t v : kept.pool)
        sum = sum * 1000003u + v;
    for (uint32_t v : kept.filter)
        sum = sum * 1000003u + v;
    printf("{\"graphs\": %u, \"k\": 32, \"fresh_tables\": %s, \"us_per_graph\": %.2f, \"table_entries_per_graph\": %.0f, \"checksum\": \"%016llx\"}\n", n,
           fresh ? "true" : "false", best / n * 1e6, (double)entries / n, (unsigned long long)sum);
    return 0;
}
