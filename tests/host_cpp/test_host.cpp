// C++ tests of the host-side mirror (grm::alignReads / GraphAligner / CompositeAligner / SiteBatcher) on a real
// GPU.  The fixtures are the reference's own unit tests, re-typed against the same class and method names:
//   ParagraphTest.Aligns       src/c++/test/test_paragraph_parts.cpp:46-159
//   DisambiguationTest         src/c++/test/test_disambiguation.cpp:44-105
// Exit code 0 = all checks passed.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <list>
#include <random>
#include <thread>
#include <string>
#include <vector>

#include "grm/Align.hh"
#include "grm/CompositeAligner.hh"
#include "grm/GraphAligner.hh"
#include "genotyping/GraphBreakpointGenotyper.hh"
#include "grm/ValidationAligner.hh"
#include "paragraph/SiteBatcher.hh"
#include "paragraph/Statistics.hh"

using namespace common;
using namespace grm;
using namespace graphtools;

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
#define EXPECT_TRUE(a) EXPECT_EQ(bool(a), true)

static std::string join(std::vector<std::string> const& v)
{
    std::string s;
    for (auto const& x : v)
        s += (s.empty() ? "" : ",") + x;
    return s;
}

static Graph alignsGraph()
{
    Graph graph{ 4 };
    graph.setNodeName(0, "LF");
    graph.setNodeSeq(0, "AAAAAAAAAAA");
    graph.setNodeName(1, "P1");
    graph.setNodeSeq(1, "TTTTTTTT");
    graph.setNodeName(2, "Q1");
    graph.setNodeSeq(2, "GGGGGGGG");
    graph.setNodeName(3, "RF");
    graph.setNodeSeq(3, "AAAAAAAAAAA");
    graph.addEdge(0, 1);
    graph.addEdge(0, 2);
    graph.addEdge(0, 3);
    graph.addEdge(1, 3);
    graph.addEdge(2, 3);
    graph.addLabelToEdge(0, 1, "P");
    graph.addLabelToEdge(1, 3, "P");
    graph.addLabelToEdge(0, 2, "Q");
    graph.addLabelToEdge(2, 3, "Q");
    graph.addLabelToEdge(0, 3, "D");
    return graph;
}

static std::vector<p_Read> alignsReads()
{
    const char* raw[6][3] = { { "f1", "AAAAAAAATTTTCTTTAAAAAAAA", "########################" },
                              { "f2", "TTTTTTAAAGAAAATTTTTTT", "#####################" },
                              { "f3", "AAAAAGCGGGGGGAAAAAA", "###################" },
                              { "f4", "AAAAGCGGGGGGAAAAAA", "##################" },
                              { "f5", "TTTTTTCCCCCCGCTTTTT", "###################" },
                              { "f6", "AAAAAAAAAAAAAAAAAAA", "###################" } };
    std::vector<p_Read> reads;
    for (auto& r : raw)
        reads.emplace_back(new Read(r[0], r[1], r[2]));
    return reads;
}

struct Expect
{
    const char* bases;
    int pos;
    const char* cigar;
    int mapq, score;
    bool reverse;
    const char *nodes, *edges, *seqs;
};
static const Expect kAligns[6] = {
    { "AAAAAAAATTTTCTTTAAAAAAAA", 3, "0[8M]1[4M1X3M]3[8M]", 60, 19, false, "LF,P1,RF", "LF_P1,P1_RF", "P" },
    { "AAAAAAATTTTCTTTAAAAAA", 4, "0[7M]1[4M1X3M]3[6M]", 60, 16, true, "LF,P1,RF", "LF_P1,P1_RF", "P" },
    { "AAAAAGCGGGGGGAAAAAA", 6, "0[5M]2[1M1X6M]3[6M]", 60, 14, false, "LF,Q1,RF", "LF_Q1,Q1_RF", "Q" },
    { "AAAAGCGGGGGGAAAAAA", 7, "0[4M]2[1M1X6M]3[6M]", 60, 13, false, "LF,Q1,RF", "LF_Q1,Q1_RF", "Q" },
    { "AAAAAGCGGGGGGAAAAAA", 6, "0[5M]2[1M1X6M]3[6M]", 60, 14, true, "LF,Q1,RF", "LF_Q1,Q1_RF", "Q" },
    { "AAAAAAAAAAAAAAAAAAA", 0, "0[11M]3[8M]", 60, 19, false, "LF,RF", "LF_RF", "D" },
};

static void testAlignReads()
{
    Graph graph = alignsGraph();
    auto reads = alignsReads();
    std::list<Path> paths;
    grm::alignReads(&graph, paths, reads, nullptr, false, true, false, false, false);
    EXPECT_EQ(reads.size(), size_t(6));
    for (size_t i = 0; i < reads.size() && i < 6; ++i)
    {
        Read const& r = *reads[i];
        EXPECT_EQ(r.bases(), std::string(kAligns[i].bases));  // reverse-strand reads come back reverse-complemented
        EXPECT_EQ(r.graph_pos(), kAligns[i].pos);
        EXPECT_EQ(r.graph_cigar(), std::string(kAligns[i].cigar));
        EXPECT_EQ(r.graph_mapq(), kAligns[i].mapq);
        EXPECT_EQ(r.graph_alignment_score(), kAligns[i].score);
        EXPECT_TRUE(r.is_graph_alignment_unique());
        EXPECT_EQ(r.is_graph_reverse_strand(), kAligns[i].reverse);
        EXPECT_EQ(int(r.graph_mapping_status()), int(Read::MAPPED));
    }
}

// grm::alignReads(validate_alignments = true): the ValidationAligner bookkeeping (lib/grm/ValidationAligner.cpp:59-88) --
// reads named "<Path::encode() of the simulated path>_<n>", an alignment whose node sequence is not part of that path is
// "mismapped"; the reads themselves come out as without validation
static void testValidationAligner()
{
    Graph graph = alignsGraph();
    std::list<graphtools::Path> paths;
    graphtools::Path p, q;
    p.graph = q.graph = &graph;
    p.nodes = { 0, 1, 3 };
    q.nodes = { 0, 2, 3 };
    p.start_position = q.start_position = 0;
    p.end_position = q.end_position = 10;
    paths.push_back(p);
    paths.push_back(q);
    EXPECT_EQ(p.encode(), std::string("(0@0)-(1)-(3@10)"));
    graphtools::Path single;
    single.nodes = { 2 };
    single.start_position = 1;
    single.end_position = 5;
    EXPECT_EQ(single.encode(), std::string("(2@1)-(2@5)"));
    using VA = grm::ValidationAligner<grm::CompositeAligner>;
    const unsigned total0 = VA::total(), aligned0 = VA::aligned(), mis0 = VA::mismapped(), rep0 = VA::repeats();
    std::vector<p_Read> reads;
    reads.emplace_back(new Read(p.encode() + "_0", kAligns[0].bases, std::string(strlen(kAligns[0].bases), '#')));  // from P, aligns to P
    reads.emplace_back(new Read(p.encode() + "_1", kAligns[2].bases, std::string(strlen(kAligns[2].bases), '#')));  // named P, aligns to Q
    reads.emplace_back(new Read(q.encode() + "_2", kAligns[3].bases, std::string(strlen(kAligns[3].bases), '#')));  // from Q, aligns to Q
    reads.emplace_back(new Read(q.encode() + "_3", kAligns[5].bases, std::string(strlen(kAligns[5].bases), '#')));  // 0[..]3[..]: not a piece of Q
    std::vector<p_Read> plain;
    for (auto const& r : reads)
        plain.emplace_back(new Read(*r));
    grm::alignReads(&graph, paths, reads, grm::ReadFilter(), false, true, false, false, true, 1);
    grm::alignReads(&graph, paths, plain, grm::ReadFilter(), false, true, false, false, false, 1);
    EXPECT_EQ(VA::total() - total0, 4u);
    EXPECT_EQ(VA::aligned() - aligned0, 4u);
    EXPECT_EQ(VA::mismapped() - mis0, 2u);
    EXPECT_EQ(VA::repeats() - rep0, 0u);
    EXPECT_EQ(reads.size(), plain.size());
    for (size_t i = 0; i < reads.size() && i < plain.size(); ++i)
        EXPECT_EQ(reads[i]->graph_cigar(), plain[i]->graph_cigar());
}

static void testGraphAlignerAlign()
{
    Graph graph = alignsGraph();
    GraphAligner aligner;
    aligner.setGraph(&graph);
    int mapq = -1, pos = -1, score = -1;
    const std::string cigar = aligner.align("AAAAAAAATTTTCTTTAAAAAAAA", mapq, pos, score);
    EXPECT_EQ(cigar, std::string("0[8M]1[4M1X3M]3[8M]"));
    EXPECT_EQ(mapq, 60);
    EXPECT_EQ(pos, 3);
    EXPECT_EQ(score, 19);
}

static void testCompositeAlignerFilter()
{
    Graph graph = alignsGraph();
    CompositeAligner aligner(false, true, false, false);
    std::list<Path> paths;
    aligner.setGraph(&graph, paths);
    Read read("f", "AAAAAAAATTTTCTTTAAAAAAAA", "########################");
    aligner.alignRead(read, [](Read& r) { return r.graph_alignment_score() > 10; });
    EXPECT_EQ(int(read.graph_mapping_status()), int(Read::BAD_ALIGN));
    EXPECT_EQ(aligner.attempted(), 1u);
    EXPECT_EQ(aligner.filtered(), 1u);
    EXPECT_EQ(aligner.mappedSw(), 0u);
}

static void testSiteBatcher()
{
    // site 0: ParagraphTest graph; site 1: DisambiguationTest graph
    Graph g0 = alignsGraph();
    auto r0 = alignsReads();
    Graph g1(5);
    const char* names[5] = { "LF", "R1", "R2", "A1", "RF" };
    const char* seqs[5] = { "AAAAAAAAAA", "TTTTTTTTTT", "TTTTTTTTTT", "GGGGGGGGGG", "AAAAAAAAAA" };
    for (NodeId n = 0; n < 5; ++n)
    {
        g1.setNodeName(n, names[n]);
        g1.setNodeSeq(n, seqs[n]);
    }
    g1.addEdge(0, 1);
    g1.addEdge(0, 4);
    g1.addEdge(1, 2);
    g1.addEdge(1, 3);
    g1.addEdge(2, 4);
    g1.addEdge(3, 4);
    g1.addLabelToEdge(0, 1, "R");
    g1.addLabelToEdge(1, 2, "R");
    g1.addLabelToEdge(2, 4, "R");
    g1.addLabelToEdge(0, 4, "D");
    std::vector<p_Read> r1;
    r1.emplace_back(new Read("f0", "AAAAAAAAAATTTTTTTTTTTTTTTTTTTTAAAAAAAAAA", "AAAAAAAAAATTTTTTTTTTTTTTTTTTTTAAAAAAAAAA"));
    r1.emplace_back(new Read("f1", "AAAAAAAAAATTTTTTTTTTT", "AAAAAAAAAATTTTTTTTTTT"));
    r1.emplace_back(new Read("f2", "AAAAAAAAAATTTTTTTTTTGGGGGGGGGGAAAAAAAAAA", "AAAAAAAAAATTTTTTTTTTGGGGGGGGGGAAAAAAAAAA"));
    r1.emplace_back(new Read("f3", "AAAAAAAAAAAAAAAAAAAA", "AAAAAAAAAAAAAAAAAAAA"));

    paragraph::SiteBatcher batcher;
    batcher.addSite(&g0, &r0);
    batcher.addSite(&g1, &r1);
    paragraph::BatchParameters prm;
    prm.remove_nonuniq_reads = false;
    prm.use_support_filters = false;  // what the reference's unit tests call disambiguateReads with
    batcher.run(prm);
    EXPECT_EQ(r0.size(), size_t(6));
    for (size_t i = 0; i < r0.size() && i < 6; ++i)
    {
        EXPECT_EQ(r0[i]->graph_cigar(), std::string(kAligns[i].cigar));
        EXPECT_EQ(join(r0[i]->graph_nodes_supported()), std::string(kAligns[i].nodes));
        EXPECT_EQ(join(r0[i]->graph_edges_supported()), std::string(kAligns[i].edges));
        EXPECT_EQ(join(r0[i]->graph_sequences_supported()), std::string(kAligns[i].seqs));
    }
    EXPECT_EQ(r1.size(), size_t(4));
    const char* want1[4] = { "R", "R", "", "D" };  // test_disambiguation.cpp:97-105
    for (size_t i = 0; i < r1.size() && i < 4; ++i)
        EXPECT_EQ(join(r1[i]->graph_sequences_supported()), std::string(want1[i]));
    auto const& c0 = batcher.counts(0);
    EXPECT_EQ(c0.by_sequence.at("P").count, uint64_t(2));
    EXPECT_EQ(c0.by_sequence.at("Q").count, uint64_t(3));
    EXPECT_EQ(c0.by_sequence.at("D").count, uint64_t(1));
    EXPECT_EQ(c0.by_node.at("LF").count, uint64_t(6));
    EXPECT_EQ(c0.by_node.at("LF").reads, uint64_t(6));
    EXPECT_EQ(c0.by_node.at("LF").fwd, uint64_t(4));
    EXPECT_EQ(c0.by_node.at("LF").rev, uint64_t(2));
    EXPECT_EQ(c0.by_edge.at("LF_RF").count, uint64_t(1));
    EXPECT_EQ(c0.mapped, uint64_t(6));
    auto const& c1 = batcher.counts(1);
    EXPECT_EQ(c1.by_sequence.at("R").count, uint64_t(2));
    EXPECT_EQ(c1.by_edge.at("LF_RF").count, uint64_t(1));
}

// More than 8 sequence labels on a graph (the device keeps no dense sequence-set table then: the family totals are summed on
// the host from the per-read label sets), and a fragment that supports 60 nodes / 59 edges (no per-fragment limit).
static void testManyLabelsAndLongPaths()
{
    Graph g0 = alignsGraph();
    for (int extra = 2; extra <= 5; ++extra)
    {
        const std::string p = "P" + std::to_string(extra), q = "Q" + std::to_string(extra);
        g0.addLabelToEdge(0, 1, p);
        g0.addLabelToEdge(1, 3, p);
        g0.addLabelToEdge(0, 2, q);
        g0.addLabelToEdge(2, 3, q);
    }
    auto r0 = alignsReads();  // 11 labels: D, P, P2..P5, Q, Q2..Q5
    // a chain of 60 three-base nodes; one fragment of two reads that together cover all of it
    const int kNodes = 60;
    Graph g1((size_t)kNodes);
    std::string whole;
    unsigned state = 12345;
    for (int n = 0; n < kNodes; ++n)
    {
        std::string seq;
        for (int c = 0; c < 3; ++c)
        {
            state = state * 1103515245u + 12345u;
            seq += "ACGT"[(state >> 16) & 3];
        }
        g1.setNodeName((NodeId)n, "n" + std::to_string(n));
        g1.setNodeSeq((NodeId)n, seq);
        whole += seq;
        if (n)
        {
            g1.addEdge((NodeId)(n - 1), (NodeId)n);
            g1.addLabelToEdge((NodeId)(n - 1), (NodeId)n, "CHAIN");
        }
    }
    std::vector<p_Read> r1;
    r1.emplace_back(new Read("pair", whole.substr(0, 120), std::string(120, '#')));
    r1.emplace_back(new Read("pair", whole.substr(60), std::string(whole.size() - 60, '#')));
    r1[1]->set_is_first_mate(false);

    paragraph::SiteBatcher batcher;
    batcher.addSite(&g0, &r0);
    batcher.addSite(&g1, &r1);
    paragraph::BatchParameters prm;
    prm.remove_nonuniq_reads = false;
    prm.use_support_filters = false;
    batcher.run(prm);
    auto const& c0 = batcher.counts(0);
    EXPECT_EQ(c0.by_sequence.size(), size_t(3));
    EXPECT_EQ(c0.by_sequence.at("P,P2,P3,P4,P5").count, uint64_t(2));
    EXPECT_EQ(c0.by_sequence.at("Q,Q2,Q3,Q4,Q5").count, uint64_t(3));
    EXPECT_EQ(c0.by_sequence.at("Q,Q2,Q3,Q4,Q5").fwd + c0.by_sequence.at("Q,Q2,Q3,Q4,Q5").rev, uint64_t(3));
    EXPECT_EQ(c0.by_sequence.at("D").count, uint64_t(1));
    EXPECT_EQ(join(r0[0]->graph_sequences_supported()), std::string("P,P2,P3,P4,P5"));
    auto const& c1 = batcher.counts(1);
    EXPECT_EQ(r1.size(), size_t(2));
    EXPECT_EQ(c1.by_node.size(), size_t(kNodes));
    EXPECT_EQ(c1.by_edge.size(), size_t(kNodes - 1));
    for (auto const& kv : c1.by_node)
    {
        EXPECT_EQ(kv.second.count, uint64_t(1));  // ONE fragment, however many of its reads cover the node
        EXPECT_EQ(kv.second.reads, uint64_t(2));
    }
    EXPECT_EQ(c1.by_sequence.at("CHAIN").count, uint64_t(1));
}

static Graph deletionGraph(const char* l, const char* d, const char* r)
{
    // graphtools::makeDeletionGraph (GT!/src/graphcore/GraphBuilders.cpp): left -> {deletion, right}, deletion -> right
    Graph g(3);
    g.setNodeSeq(0, l);
    g.setNodeSeq(1, d);
    g.setNodeSeq(2, r);
    g.addEdge(0, 1);
    g.addEdge(0, 2);
    g.addEdge(1, 2);
    return g;
}

// PathAligner.Aligns_ExactMatch / Aligns_ExactMatchLongMEM / Aligns_MultipleMatches, src/c++/test/test_pathaligner.cpp:37-144
static void testPathAligner()
{
    Graph g = deletionGraph("AAAAAAAAA", "CCCC", "GGGGGGGGG");
    std::list<Path> paths;
    PathAligner aligner(16);
    aligner.setGraph(&g, paths);
    struct Case
    {
        const char* bases;
        bool bam_reverse;
        int pos;
        const char* cigar;
        int score;
        bool reverse;
    } cases[] = { { "AAAAAAAAGGGGGGGG", false, 1, "0[8M]2[8M]", 16, false },
                  { "CCCCCCCCTTTTTTTT", false, 1, "0[8M]2[8M]", 16, true },
                  { "AAAAAAAACCCCGGGG", false, 1, "0[8M]1[4M]2[4M]", 16, false },
                  { "CCCCGGGGTTTTTTTT", true, 1, "0[8M]1[4M]2[4M]", 16, true },
                  { "AAAAAAAAGGGGGGGGG", false, 1, "0[8M]2[9M]", 17, false },
                  { "CCCCCCCCCTTTTTTTTT", false, 0, "0[9M]2[9M]", 18, true } };
    for (auto const& c : cases)
    {
        Read read;
        read.setCoreInfo("f1", c.bases, "################");
        read.set_is_reverse_strand(c.bam_reverse);
        aligner.alignRead(read);
        EXPECT_EQ(int(read.graph_mapping_status()), int(Read::MAPPED));
        EXPECT_EQ(read.graph_pos(), c.pos);
        EXPECT_EQ(read.graph_cigar(), std::string(c.cigar));
        EXPECT_EQ(read.graph_alignment_score(), c.score);
        EXPECT_EQ(read.is_graph_reverse_strand(), c.reverse);
    }
    Graph g2 = deletionGraph("GGGGGGGGGGGG", "CCCCCCCCCCCCCCCC", "GGGGGGGGGGGGGTGGG");
    PathAligner aligner2(16);
    aligner2.setGraph(&g2, paths);
    Read read;
    read.setCoreInfo("f1", "CCCCCCCCCCCCGGGGGGGGGGGG", "#####################################");
    aligner2.alignRead(read);
    EXPECT_EQ(int(read.graph_mapping_status()), int(Read::MAPPED));
    EXPECT_EQ(read.graph_pos(), 4);
    EXPECT_EQ(read.graph_cigar(), std::string("1[12M]2[12M]"));
    EXPECT_EQ(read.graph_alignment_score(), 24);
    EXPECT_EQ(read.is_graph_reverse_strand(), false);
    EXPECT_EQ(read.is_graph_alignment_unique(), false);
    EXPECT_EQ(read.graph_mapq(), 0);
    // cascade: the exact read goes through the path stage, the mismatching one through gssw
    Graph g3 = alignsGraph();
    CompositeAligner comp(true, true, false, false);
    comp.setGraph(&g3, paths);
    Read exact("e", "AAAAAAAAAAATTTTTTTTAAAAAAAAAAA", ""), inexact("i", "AAAAAAAATTTTCTTTAAAAAAAA", "");
    std::vector<Read*> both{ &exact, &inexact };
    comp.alignReads(both, nullptr);
    EXPECT_EQ(comp.mappedPath(), 0u);  // 30 bp read is shorter than the 32-mer index: falls through to gssw
    EXPECT_EQ(comp.mappedSw(), 2u);
    EXPECT_EQ(exact.graph_cigar(), std::string("0[11M]1[8M]3[11M]"));
    EXPECT_EQ(inexact.graph_cigar(), std::string("0[8M]1[4M1X3M]3[8M]"));
}

// KmerAlignerTest.Aligns, src/c++/test/test_kmeraligner.cpp:44-193 (KmerAligner<10>, paths P / Q / D)
static void testKmerAligner()
{
    Graph graph = alignsGraph();
    std::list<Path> paths;
    for (auto const& nodes : std::vector<std::vector<NodeId>>{ { 0, 1, 3 }, { 0, 2, 3 }, { 0, 3 } })
    {
        Path p;
        p.graph = &graph;
        p.nodes = nodes;
        p.end_position = (int32_t)graph.nodeSeq(nodes.back()).size() - 1;
        paths.push_back(p);
    }
    KmerAligner<10> aligner;
    aligner.setGraph(&graph, paths);
    struct Case
    {
        const char* bases;
        int status, pos;
        const char* cigar;
        int score;
        bool reverse;
    } cases[] = { { "AAAAAAAATTTTTTTTAAAAAAAA", Read::MAPPED, 3, "0[8M]1[8M]3[8M]", 24, false },
                  { "TTTTTTAAAAAAAATTTTTTT", Read::MAPPED, 4, "0[7M]1[8M]3[6M]", 21, true },
                  { "AAAAAGGGGGGGGAAAAAA", Read::MAPPED, 6, "0[5M]2[8M]3[6M]", 19, false },
                  { "AAAAGGGGGGGGAAAAAA", Read::MAPPED, 7, "0[4M]2[8M]3[6M]", 18, false },
                  { "TTTTTTCCCCCCCCTTTTT", Read::MAPPED, 6, "0[5M]2[8M]3[6M]", 19, true },
                  { "AAAAAAAAAAAAAAAAAAA", Read::BAD_ALIGN, 0, "0[11M]3[8M]", 19, false } };
    for (auto const& c : cases)
    {
        Read read("f", c.bases, "");
        aligner.alignRead(read);
        EXPECT_EQ(int(read.graph_mapping_status()), c.status);
        EXPECT_EQ(read.graph_pos(), c.pos);
        EXPECT_EQ(read.graph_cigar(), std::string(c.cigar));
        EXPECT_EQ(read.graph_alignment_score(), c.score);
        EXPECT_EQ(read.is_graph_reverse_strand(), c.reverse);
        EXPECT_EQ(read.graph_mapq(), c.status == Read::MAPPED ? 60 : 0);
    }
}

// KlibAlignerTest.Aligns, src/c++/test/test_klibaligner.cpp:44-193, and the klib stage of the cascade
static void testKlibAligner()
{
    Graph graph = alignsGraph();
    std::list<Path> paths;
    for (auto const& nodes : std::vector<std::vector<NodeId>>{ { 0, 1, 3 }, { 0, 2, 3 }, { 0, 3 } })
    {
        Path p;
        p.graph = &graph;
        p.nodes = nodes;
        p.end_position = (int32_t)graph.nodeSeq(nodes.back()).size() - 1;
        paths.push_back(p);
    }
    KlibAligner aligner;
    aligner.setGraph(&graph, paths);
    struct Case
    {
        const char* bases;
        const char* bases_after;
        int pos;
        const char* cigar;
        int score;
        bool reverse;
    } cases[] = { { "AAAAAAAATTTTTTTTAAAAAAAA", "AAAAAAAATTTTTTTTAAAAAAAA", 3, "0[8M]1[8M]3[8M]", 24, false },
                  { "TTTTTTAAAAAAAATTTTTTT", "AAAAAAATTTTTTTTAAAAAA", 4, "0[7M]1[8M]3[6M]", 21, true },
                  { "AAAAAGGGGGGGGAAAAAA", "AAAAAGGGGGGGGAAAAAA", 6, "0[5M]2[8M]3[6M]", 19, false },
                  { "AAAAGGGGGGGGAAAAAA", "AAAAGGGGGGGGAAAAAA", 7, "0[4M]2[8M]3[6M]", 18, false },
                  { "TTTTTTCCCCCCCCTTTTT", "AAAAAGGGGGGGGAAAAAA", 6, "0[5M]2[8M]3[6M]", 19, true },
                  { "TTTTTTCCCCCCCCGGGGG", "CCCCCGGGGGGGGAAAAAA", 0, "2[5S8M]3[6M]", 14, true },
                  { "GGGGGGCCCCCCCCTTTTT", "AAAAAGGGGGGGGCCCCCC", 6, "0[5M]2[8M6S]", 13, true } };
    for (auto const& c : cases)
    {
        Read read("f", c.bases, "###");
        aligner.alignRead(read);
        EXPECT_EQ(int(read.graph_mapping_status()), int(Read::MAPPED));
        EXPECT_EQ(read.bases(), std::string(c.bases_after));
        EXPECT_EQ(read.quals(), std::string("###"));  // KlibAligner.cpp:328 replaces the bases only
        EXPECT_EQ(read.graph_pos(), c.pos);
        EXPECT_EQ(read.graph_cigar(), std::string(c.cigar));
        EXPECT_EQ(read.graph_alignment_score(), c.score);
        EXPECT_EQ(read.is_graph_reverse_strand(), c.reverse);
        EXPECT_EQ(read.graph_mapq(), 60);
        EXPECT_EQ(read.is_graph_alignment_unique(), true);
    }
    EXPECT_EQ(aligner.attempted(), 7u);
    EXPECT_EQ(aligner.mapped(), 7u);
    // cascade: klib only, then klib -> gssw with a filter that rejects everything klib maps
    CompositeAligner klibOnly(false, false, true, false);
    klibOnly.setGraph(&graph, paths);
    Read a("a", "AAAAAGGGGGGGGAAAAAA", ""), b("b", "TTTTTTCCCCCCCCGGGGG", "");
    std::vector<Read*> two{ &a, &b };
    klibOnly.alignReads(two, nullptr);
    EXPECT_EQ(klibOnly.mappedKlib(), 2u);
    EXPECT_EQ(klibOnly.mappedSw(), 0u);
    EXPECT_EQ(b.graph_cigar(), std::string("2[5S8M]3[6M]"));
    CompositeAligner both(false, true, true, false);
    both.setGraph(&graph, paths);
    Read c2("c", "AAAAAGGGGGGGGAAAAAA", "");
    unsigned calls = 0;
    both.alignRead(c2, [&](Read& r) { return ++calls == 1; });
    EXPECT_EQ(both.mappedKlib(), 0u);
    EXPECT_EQ(both.mappedSw(), 1u);
    EXPECT_EQ(both.filtered(), 0u);
    EXPECT_EQ(c2.graph_cigar(), std::string("0[5M]2[8M]3[6M]"));
}

static std::string revComp(std::string s)
{
    for (char& c : s)
        c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
    std::reverse(s.begin(), s.end());
    return s;
}

// paragraph's default cascade in the batcher (path -> gssw) must give every read what CompositeAligner gives it
static void testSiteBatcherPathStage()
{
    std::mt19937_64 rng(5);
    auto rnd = [&](size_t n) {
        std::string s(n, 'A');
        for (auto& c : s)
            c = "ACGT"[rng() % 4];
        return s;
    };
    Graph g(3, false);
    const std::string seqs[3] = { rnd(120), rnd(40), rnd(120) };
    const char* names[3] = { "LF", "MID", "RF" };
    for (NodeId n = 0; n < 3; ++n)
    {
        g.setNodeName(n, names[n]);
        g.setNodeSeq(n, seqs[n]);
    }
    g.addEdge(0, 1);
    g.addEdge(0, 2);
    g.addEdge(1, 2);
    g.addLabelToEdge(0, 1, "REF");
    g.addLabelToEdge(1, 2, "REF");
    g.addLabelToEdge(0, 2, "ALT");
    const std::string hap[2] = { seqs[0] + seqs[1] + seqs[2], seqs[0] + seqs[2] };
    std::vector<p_Read> a, b;
    for (int i = 0; i < 60; ++i)
    {
        const std::string& h = hap[i & 1];
        std::string r = h.substr(rng() % (h.size() - 70), 70);
        if (i % 3 == 0)
            r[rng() % r.size()] = "ACGT"[rng() % 4];  // not an exact match any more: falls through to gssw
        if (i % 4 == 0)
            r = revComp(r);
        a.emplace_back(new Read("f" + std::to_string(i), r, std::string(r.size(), '#')));
        b.emplace_back(new Read("f" + std::to_string(i), r, std::string(r.size(), '#')));
    }
    paragraph::SiteBatcher batcher;
    batcher.addSite(&g, &a);
    paragraph::BatchParameters prm;
    prm.path_sequence_matching = true;
    batcher.run(prm);
    std::list<Path> no_paths;
    CompositeAligner comp(true, true, false, false);
    comp.setGraph(&g, no_paths);
    std::vector<Read*> ptrs;
    for (auto& r : b)
        ptrs.push_back(r.get());
    comp.alignReads(ptrs, nullptr);
    EXPECT_TRUE(comp.mappedPath() > 10u);
    EXPECT_TRUE(comp.mappedSw() > 10u);
    size_t k = 0;
    for (auto& r : b)
    {
        // the batcher drops nothing here (no BAD_ALIGN expected for these reads) and keeps the input order
        if (k >= a.size())
            break;
        EXPECT_EQ(a[k]->fragment_id(), r->fragment_id());
        EXPECT_EQ(a[k]->graph_cigar(), r->graph_cigar());
        EXPECT_EQ(a[k]->graph_pos(), r->graph_pos());
        EXPECT_EQ(a[k]->graph_mapq(), r->graph_mapq());
        EXPECT_EQ(a[k]->is_graph_reverse_strand(), r->is_graph_reverse_strand());
        EXPECT_EQ(a[k]->bases(), r->bases());
        ++k;
    }
    EXPECT_EQ(k, b.size());
}

// align + count on the device, genotype on the host (lib/grmpy/CountAndGenotype.cpp:46-88): a 60 bp deletion, three
// samples simulated as REF/REF, REF/ALT and ALT/ALT at ~30x
static void testSiteToGenotype()
{
    std::mt19937_64 rng(11);
    auto rnd = [&](size_t n) {
        std::string s(n, 'A');
        for (auto& c : s)
            c = "ACGT"[rng() % 4];
        return s;
    };
    Graph g(5, false);
    const char* names[5] = { "source", "LF", "MID", "RF", "sink" };
    const std::string seqs[5] = { "X", rnd(200), rnd(60), rnd(200), "X" };
    for (NodeId n = 0; n < 5; ++n)
    {
        g.setNodeName(n, names[n]);
        g.setNodeSeq(n, seqs[n]);
    }
    g.addEdge(0, 1);
    g.addEdge(1, 2);
    g.addEdge(1, 3);
    g.addEdge(2, 3);
    g.addEdge(3, 4);
    g.addLabelToEdge(1, 2, "REF");
    g.addLabelToEdge(2, 3, "REF");
    g.addLabelToEdge(1, 3, "ALT");
    const std::string hap[2] = { seqs[1] + seqs[2] + seqs[3], seqs[1] + seqs[3] };
    const int alt_copies[3] = { 0, 1, 2 };
    const char* want[3] = { "REF/REF", "ALT/REF", "ALT/ALT" };
    genotyping::GraphBreakpointGenotyper genotyper;
    genotyper.reset(&g);
    std::vector<std::vector<p_Read>> reads(3);
    paragraph::SiteBatcher batcher;
    const int L = 100;
    for (int s = 0; s < 3; ++s)
    {
        for (int copy = 0; copy < 2; ++copy)
        {
            const std::string& h = hap[copy < alt_copies[s] ? 1 : 0];
            const int n_reads = (int)(15.0 * h.size() / L);  // 15x per haplotype
            for (int i = 0; i < n_reads; ++i)
            {
                const size_t st = rng() % (h.size() - L + 1);
                std::string r = h.substr(st, L);
                if (rng() % 50 == 0)
                    r[rng() % L] = "ACGT"[rng() % 4];
                reads[s].emplace_back(new Read("s" + std::to_string(s) + "c" + std::to_string(copy) + "r" + std::to_string(i), r, std::string(L, '#')));
            }
        }
        batcher.addSite(&g, &reads[s]);
    }
    batcher.run();
    for (int s = 0; s < 3; ++s)
        genotyper.addSample("sample" + std::to_string(s), paragraph::readCountsByEdge(batcher.counts(s)), 30.0, L, 8.0);
    genotyper.runGenotyping();
    const std::vector<std::string>& an = genotyper.alleleNames();
    for (int s = 0; s < 3; ++s)
    {
        const genotyping::Genotype gt = genotyper.getGenotype("sample" + std::to_string(s), "");
        EXPECT_EQ(gt.toString(&an), std::string(want[s]));
        EXPECT_EQ(gt.filterString(), std::string("PASS"));
    }
}

// the whole cascade (path -> k-mer -> klib -> gssw, filters between the stages) batched on the device == the per-read
// cascade of CompositeAligner with the NonUniq + BadAlign chain as its filter
static void testSiteBatcherFullCascade()
{
    std::mt19937_64 rng(17);
    auto rnd = [&](size_t n) {
        std::string s(n, 'A');
        for (auto& c : s)
            c = "ACGT"[rng() % 4];
        return s;
    };
    Graph g(4, false);
    const std::string seqs[4] = { rnd(150), rnd(60), rnd(35), rnd(150) };
    const char* names[4] = { "LF", "REF", "ALT", "RF" };
    for (NodeId n = 0; n < 4; ++n)
    {
        g.setNodeName(n, names[n]);
        g.setNodeSeq(n, seqs[n]);
    }
    g.addEdge(0, 1);
    g.addEdge(0, 2);
    g.addEdge(1, 3);
    g.addEdge(2, 3);
    g.addLabelToEdge(0, 1, "REF");
    g.addLabelToEdge(1, 3, "REF");
    g.addLabelToEdge(0, 2, "ALT");
    g.addLabelToEdge(2, 3, "ALT");
    std::list<Path> paths;
    for (NodeId mid : { 1u, 2u })
    {
        Path p;
        p.graph = &g;
        p.nodes = { 0, mid, 3 };
        p.start_position = 0;
        p.end_position = (int32_t)seqs[3].size() - 1;
        paths.push_back(p);
    }
    const std::string hap[2] = { seqs[0] + seqs[1] + seqs[3], seqs[0] + seqs[2] + seqs[3] };
    std::vector<p_Read> a, b;
    for (int i = 0; i < 160; ++i)
    {
        const std::string& h = hap[i & 1];
        std::string r = h.substr(rng() % (h.size() - 100), 100);
        switch (i % 5)
        {
        case 1:  // one substitution: k-mer stage
            r[10 + rng() % 80] = "ACGT"[rng() % 4];
            break;
        case 2:  // a small deletion or insertion: klib stage
            if (rng() & 1)
                r.erase(30 + rng() % 40, 1 + rng() % 3);
            else
                r.insert(30 + rng() % 40, rnd(1 + rng() % 3));
            break;
        case 3:  // noisy: a substitution every ~12 bases
            for (size_t k = rng() % 12; k < r.size(); k += 8 + rng() % 8)
                r[k] = "ACGT"[rng() % 4];
            break;
        case 4:  // half of the read is foreign: clipped, BadAlign rejects it whatever stage finds it
            r = r.substr(0, 45) + rnd(55);
            break;
        default: break;  // exact: path stage
        }
        if (i % 3 == 0)
            r = revComp(r);
        a.emplace_back(new Read("f" + std::to_string(i), r, std::string(r.size(), '#')));
        b.emplace_back(new Read("f" + std::to_string(i), r, std::string(r.size(), '#')));
    }
    paragraph::SiteBatcher batcher;
    batcher.addSite(&g, &a, &paths);
    paragraph::BatchParameters prm;
    prm.path_sequence_matching = true;
    prm.kmer_sequence_matching = true;
    prm.klib_sequence_matching = true;
    batcher.run(prm);

    CompositeAligner comp(true, true, true, true);
    comp.setGraph(&g, paths);
    std::vector<Read*> ptrs;
    for (auto& r : b)
        ptrs.push_back(r.get());
    comp.alignReads(ptrs, [&](Read& read) {  // createReadFilter(graph, nonuniq = true, 0.8, kmer_len = 0)
        if (!read.is_graph_alignment_unique())
            return true;
        size_t clipped = 0, query = 0;
        for (auto const& piece : paragraph::decodeGraphCigar(read.graph_cigar(), g))
        {
            clipped += piece.clipped;
            query += piece.queryLength();
        }
        return (double)(query - clipped) < std::round(0.8 * (double)query);
    });
    EXPECT_TRUE(comp.mappedPath() > 10u);
    EXPECT_TRUE(comp.mappedKmers() > 10u);
    EXPECT_TRUE(comp.mappedKlib() > 5u);
    EXPECT_TRUE(comp.filtered() > 5u);  // the half-foreign reads: rejected after klib, again after gssw (klib maps the rest)
    // the batcher keeps the MAPPED reads, in input order
    size_t k = 0;
    for (auto& r : b)
    {
        if (r->graph_mapping_status() != Read::MAPPED)
            continue;
        EXPECT_TRUE(k < a.size());
        if (k >= a.size())
            break;
        EXPECT_EQ(a[k]->fragment_id(), r->fragment_id());
        EXPECT_EQ(a[k]->graph_cigar(), r->graph_cigar());
        EXPECT_EQ(a[k]->graph_pos(), r->graph_pos());
        EXPECT_EQ(a[k]->graph_mapq(), r->graph_mapq());
        EXPECT_EQ(a[k]->graph_alignment_score(), r->graph_alignment_score());
        EXPECT_EQ(a[k]->is_graph_reverse_strand(), r->is_graph_reverse_strand());
        EXPECT_EQ(a[k]->bases(), r->bases());
        EXPECT_EQ(a[k]->quals(), r->quals());
        ++k;
    }
    EXPECT_EQ(k, a.size());
    EXPECT_EQ((uint64_t)(b.size() - k), batcher.counts(0).bad_align + batcher.counts(0).nonuniq);
}

// Batches of different threads overlap on the device (uploads / downloads beside another batch's kernels, pooled batch
// objects, one shared workspace): every batcher must reproduce what it produced alone.
static void testConcurrentBatchers()
{
    const int n_batchers = 6, n_rounds = 4;
    struct Job
    {
        std::vector<std::unique_ptr<Graph>> graphs;
        std::vector<std::vector<p_Read>> reads;
    };
    std::vector<Job> jobs(n_batchers);
    std::mt19937_64 rng(4711);
    auto rnd = [&](size_t n) {
        std::string s(n, 'A');
        for (auto& c : s)
            c = "ACGT"[rng() % 4];
        return s;
    };
    for (int j = 0; j < n_batchers; ++j)
    {
        const int n_sites = 3 + j * 5;
        for (int k = 0; k < n_sites; ++k)
        {
            const std::string l = rnd(120 + rng() % 80), d = rnd(20 + rng() % 200), r = rnd(120 + rng() % 80);
            jobs[j].graphs.emplace_back(new Graph(deletionGraph(l.c_str(), d.c_str(), r.c_str())));
            const std::string hap[2] = { l + d + r, l + r };
            std::vector<p_Read> reads;
            const int n_reads = 40 + (int)(rng() % 120);
            const size_t len = (j == 2) ? 250 : (j == 4 ? 300 : 100);  // byte, long-byte and 16-bit kernels side by side
            for (int i = 0; i < n_reads; ++i)
            {
                const std::string& h = hap[rng() & 1];
                if (h.size() <= len)
                    continue;
                std::string q = h.substr(rng() % (h.size() - len), len);
                for (size_t e = rng() % 60; e < q.size(); e += 20 + rng() % 60)
                    q[e] = "ACGT"[rng() % 4];
                if (rng() & 1)
                    q = revComp(q);
                reads.emplace_back(new Read("f" + std::to_string(i / 2), q, std::string(q.size(), '#')));
            }
            jobs[j].reads.push_back(std::move(reads));
        }
    }
    auto snapshot = [&](int j) {
        // a fresh copy of the reads each time: run() rewrites them
        std::vector<std::vector<p_Read>> mine(jobs[j].reads.size());
        for (size_t k = 0; k < mine.size(); ++k)
            for (auto const& r : jobs[j].reads[k])
                mine[k].emplace_back(new Read(r->fragment_id(), r->bases(), r->quals()));
        paragraph::SiteBatcher batcher;
        for (size_t k = 0; k < mine.size(); ++k)
            batcher.addSite(jobs[j].graphs[k].get(), &mine[k]);
        paragraph::BatchParameters prm;
        prm.path_sequence_matching = (j % 2) == 1;
        batcher.run(prm);
        std::string out;
        for (size_t k = 0; k < mine.size(); ++k)
        {
            auto const& c = batcher.counts(k);
            out += "|" + std::to_string(c.aligned) + "," + std::to_string(c.mapped) + "," + std::to_string(c.bad_align) + ","
                + std::to_string(c.nonuniq);
            for (auto const& kv : c.by_edge)
                out += ";" + kv.first + "=" + std::to_string(kv.second.count) + "/" + std::to_string(kv.second.fwd) + "/"
                    + std::to_string(kv.second.rev);
            for (auto const& kv : c.by_sequence)
                out += ";" + kv.first + "=" + std::to_string(kv.second.count);
            for (auto const& r : mine[k])
                out += ":" + r->graph_cigar() + "@" + std::to_string(r->graph_pos());
        }
        return out;
    };
    std::vector<std::string> alone(n_batchers);
    for (int j = 0; j < n_batchers; ++j)
        alone[j] = snapshot(j);
    EXPECT_TRUE(alone[0].size() > 100);
    for (int round = 0; round < n_rounds; ++round)
    {
        std::vector<std::string> together(n_batchers);
        std::vector<std::string> errors(n_batchers);
        std::vector<std::thread> threads;
        for (int j = 0; j < n_batchers; ++j)
            threads.emplace_back([&, j] {
                try
                {
                    together[j] = snapshot(j);
                }
                catch (std::exception const& e)
                {
                    errors[j] = e.what();
                }
            });
        for (auto& t : threads)
            t.join();
        for (int j = 0; j < n_batchers; ++j)
        {
            EXPECT_EQ(errors[j], std::string());
            EXPECT_TRUE(together[j] == alone[j]);
        }
    }
}

int main()
{
    try
    {
        testSiteToGenotype();
        testManyLabelsAndLongPaths();
        testSiteBatcherPathStage();
        testSiteBatcherFullCascade();
        testConcurrentBatchers();
        testKlibAligner();
        testKmerAligner();
        testPathAligner();
        testAlignReads();
        testValidationAligner();
        testGraphAlignerAlign();
        testCompositeAlignerFilter();
        testSiteBatcher();
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
    std::cout << "host_cpp: all checks passed\n";
    return 0;
}
