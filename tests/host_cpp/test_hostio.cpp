// CPU-only checks of the host input layer (JSON, coordinates, FASTA, BAM/BAI, read extraction, graph descriptions,
// manifests, graph coordinates, statistics).  Expectations come from the reference's own unit tests and data files:
//   src/c++/test/test_stringutil.cpp:86-118, test_readextraction.cpp:105-199, test_graph_input.cpp:46-155,
//   GT!/tests/GraphCoordinatesTest.cpp:72-140, share/test-data/multiparagraph/reads.sam (text of reads.bam).
// Usage: test_hostio <tests/golden/sites directory>
#include <unistd.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <random>
#include <sstream>

#include "common/BamReader.hh"
#include "common/Fasta.hh"
#include "common/Json.hh"
#include "common/ReadExtraction.hh"
#include "common/Region.hh"
#include "genotyping/GenotypingParameters.hh"
#include "genotyping/SampleInfo.hh"
#include "graphcore/GraphCoordinates.hh"
#include "grm/GraphInput.hh"
#include "paragraph/PackedReads.hh"
#include "paragraph/Statistics.hh"
#include "paragraph/Workflow.hh"

#include <zlib.h>
#include "../../paragraph_amd/host/src/inflate.hh"

using namespace common;

static int g_failures = 0;
#define CHECK(cond)                                                                  \
    do                                                                               \
    {                                                                                \
        if (!(cond))                                                                 \
        {                                                                            \
            std::cerr << __FILE__ << ":" << __LINE__ << ": CHECK failed: " #cond "\n"; \
            ++g_failures;                                                            \
        }                                                                            \
    } while (0)
#define CHECK_THROWS(expr)                                                           \
    do                                                                               \
    {                                                                                \
        bool threw = false;                                                          \
        try                                                                          \
        {                                                                            \
            (void)(expr);                                                            \
        }                                                                            \
        catch (std::exception const&)                                                \
        {                                                                            \
            threw = true;                                                            \
        }                                                                            \
        if (!threw)                                                                  \
        {                                                                            \
            std::cerr << __FILE__ << ":" << __LINE__ << ": expected an exception: " #expr "\n"; \
            ++g_failures;                                                            \
        }                                                                            \
    } while (0)

// what the writer's numbers must equal: printf's "%.{p}g" for the smallest p in 15, 16, 17 that reads back, ".0" after a bare integer
static std::string printfReal(double d)
{
    char buf[40];
    for (int prec = 15; prec <= 17; ++prec)
    {
        snprintf(buf, sizeof buf, "%.*g", prec, d);
        if (strtod(buf, nullptr) == d)
            break;
    }
    std::string out = buf;
    if (!strpbrk(buf, ".eEn"))
        out += ".0";
    return out;
}

static void testJsonNumbers()
{
    // the writer makes its digits with std::to_chars and lays them out by %g's rule: every kind of double against the printf loop
    uint64_t state = 88172645463325252ull;
    auto next = [&]() {
        state ^= state << 13;
        state ^= state >> 7;
        state ^= state << 17;
        return state;
    };
    size_t bad = 0, n = 0;
    auto check = [&](double d) {
        if (std::isnan(d) || std::isinf(d))
            return;
        ++n;
        if (Json(d).dump() != printfReal(d) && ++bad < 5)
            std::cerr << "number " << printfReal(d) << " written as " << Json(d).dump() << "\n";
    };
    for (int i = 0; i < 300000; ++i)
    {
        const uint64_t u = next();
        double d;
        memcpy(&d, &u, 8);
        check(d);  // any bit pattern: every magnitude, subnormals included
        const double f = std::ldexp((double)(next() >> 11), -53);
        check(f);
        check(-1000.0 * f);
        check(f * 1e-7);
        check(std::log(f + 1e-300));  // (log-likelihoods)
        check((double)(int64_t)(next() % 2000000000000ull) / (double)(1 + next() % 100000));
    }
    for (int e = -320; e <= 308; ++e)
    {
        const double d = std::pow(10.0, e);
        for (int k = -3; k <= 3; ++k)
        {
            double x = d;
            for (int s = 0; s < std::abs(k); ++s)
                x = std::nextafter(x, k < 0 ? 0.0 : INFINITY);
            check(x);
            check(-x);
        }
        check(d * 5);
        check(d * 9.999999999999999);
    }
    for (int64_t v = -20000; v <= 20000; ++v)
    {
        check((double)v);
        check(v / 8.0);
        check(v / 1000.0);
    }
    for (double v : { 0.0, -0.0, 1e15, 1e16, 1e17, 123456789012345678.0, 99999999999999.9, 999999999999999.0, 9999999999999998.0, 0.0001, 0.00001,
                      0.00012345, 5e-324, 1.7976931348623157e308, 100000.0, 0.1, 1.0 / 3.0 })
        check(v);
    CHECK(bad == 0 && n > 1500000);
}

static void testJson()
{
    const Json v = Json::parse(R"({"b": [1, -2, 3.5, 1e3, true, false, null], "a": {"s": "x\"\\\né😀", "big": 18446744073709551615}})");
    CHECK(v.isObject() && v["b"].isArray() && v["b"].size() == 7);
    CHECK(v["b"][0].asInt64() == 1 && v["b"][1].asInt64() == -2 && v["b"][2].asDouble() == 3.5 && v["b"][3].asDouble() == 1000.0);
    CHECK(v["b"][4].asBool() && !v["b"][5].asBool() && v["b"][6].isNull());
    CHECK(v["a"]["s"].asString() == "x\"\\\n\xc3\xa9\xf0\x9f\x98\x80");
    CHECK(v["a"]["big"].asUInt64() == 18446744073709551615ull);
    CHECK(v["missing"].isNull() && !v.isMember("missing") && v.isMember("a"));
    CHECK(v.getMemberNames() == (std::vector<std::string>{ "a", "b" }));  // sorted like Json::Value
    CHECK(Json::parse(v.dump()) == v);
    CHECK(Json::parse(v.dump(4)) == v);
    Json w = Json::object();
    w["nan"] = std::nan("");
    w["x"] = 0.1;
    w["i"] = 3.0;
    CHECK(w.dump() == R"({"i":3.0,"nan":null,"x":0.1})");
    Json arr;
    arr.append(1);
    arr.append("two");
    CHECK(arr.isArray() && arr.size() == 2 && arr.dump() == "[1,\"two\"]");
    CHECK(Json(1) == Json(1.0) && Json(1) != Json(2) && Json((uint64_t)5) == Json(5));
    CHECK(Json(std::nan("")) == Json(std::nan("")) && Json(std::nan("")) != Json(0.0));
    CHECK_THROWS(Json::parse("{\"a\": }"));
    CHECK_THROWS(Json::parse("[1, 2"));
    CHECK_THROWS(Json::parse("{} x"));
    CHECK_THROWS(Json::parse("\"abc"));
    CHECK_THROWS(v["b"].asString());
    CHECK_THROWS(Json::parseFile("/nonexistent/file.json"));
    // members: found without building a key, inserted with the key's own storage when it is handed over; a reference to a member
    // stays good while other members are added (documents are built that way)
    {
        Json o;
        Json& first = o["a_key_that_is_longer_than_the_small_string_buffer"];
        first = 1;
        std::string key = "another_key_that_is_longer_than_the_small_string_buffer";
        const char* storage = key.data();
        Json& second = o[std::move(key)];
        second = 2;
        CHECK(o.members().find("another_key_that_is_longer_than_the_small_string_buffer")->first.data() == storage);
        for (int i = 0; i < 100; ++i)
            o["k" + std::to_string(i)] = i;
        CHECK(first == Json(1) && second == Json(2) && o.size() == 102);
        CHECK(o.isMember("k42") && o.isMember(std::string("k42")) && !o.isMember("k420"));
        CHECK(o["k42"] == Json(42) && static_cast<Json const&>(o)["missing"].isNull() && o.size() == 102);
        o.removeMember("k42");
        o.removeMember("not there");
        CHECK(!o.isMember("k42") && o.size() == 101);
        Json moved = std::move(o["k7"]);
        CHECK(moved == Json(7) && o["k7"].isNull());
        CHECK_THROWS(Json(3)["x"]);
    }
}

static void testCoordinates()
{
    std::string chr;
    int64_t start = -1, end = -1;
    parsePos("chr1", chr, start, end);
    CHECK(chr == "chr1" && start == -1 && end == -1);
    parsePos("chr1:1,000", chr, start, end);
    CHECK(chr == "chr1" && start == 999 && end == -1);
    parsePos("chr1:1,000-2000", chr, start, end);
    CHECK(chr == "chr1" && start == 999 && end == 1999);
    CHECK(formatPos("chr2", 9, 19) == "chr2:10-20" && formatPos("chr2", 9) == "chr2:10" && formatPos("chr2") == "chr2");
    const Region r("chrX:850-1149");
    CHECK(r.chrom == "chrX" && r.start == 849 && r.end == 1148 && r.length() == 300);
    CHECK((std::string)r.getExtendedRegion(999) == "chrX:1-2148");
    CHECK((std::string)r.getExtendedRegion(100) == "chrX:750-1249");
    CHECK((std::string)r.getLeftFlank(10) == "chrX:839-849" && (std::string)r.getRightFlank(10) == "chrX:1150-1160");
}

static void testFasta(std::string const& dir)
{
    FastaFile fa(dir + "/multiparagraph/dummy.fa");
    CHECK(fa.contigSize("chr") == 440 && fa.contigSize("") == 440);
    CHECK(fa.query("chr:1-40") == std::string(40, 'A'));
    CHECK(fa.query("chr:41-80") == std::string(40, 'C'));
    CHECK(fa.query("chr:39-42") == "AACC");
    CHECK(fa.query("chr:81-231").size() == 151);
    CHECK(fa.query("chr:431-600") == std::string(10, 'C'));  // clipped at the contig end
    CHECK(fa.query("chr:441-450").empty() && fa.query("chr", 10, 5).empty());
    CHECK(fa.query("chr", -5, 2) == "AAA");
    CHECK_THROWS(fa.query("nochr:1-2"));
    // no .fai next to this one: indexed by scanning
    FastaFile plain(dir + "/basic/dummy.fa");
    CHECK(plain.getContigNames().size() == 2);
    CHECK(plain.query("SimpleDeletion:1-25") == "CGTCGACGTCGAACGATCGTCAGTA");
    CHECK(plain.query("SimpleDeletion:19-22") == "GTCA");
    // a lower-case / IUPAC base comes back as upper case / N
    {
        const std::string path = "/tmp/pg_hostio_test.fa";
        std::ofstream(path) << ">c1 description\nacgtRYn\nACG\n>c2\nTT\n";
        FastaFile mixed(path);
        CHECK(mixed.query("c1:1-10") == "ACGTNNNACG" && mixed.query("c2:1-2") == "TT" && mixed.contigSize("c1") == 10);
        std::remove(path.c_str());
    }
}

struct SamLine
{
    std::string name, rname, rnext, seq, qual;
    int flag = 0, pos = 0, mapq = 0, pnext = 0;
};

static std::vector<SamLine> readSam(std::string const& path)
{
    std::vector<SamLine> out;
    std::ifstream in(path);
    std::string line;
    while (std::getline(in, line))
    {
        if (line.empty() || line[0] == '@')
            continue;
        std::stringstream ss(line);
        SamLine s;
        std::string cigar, tlen;
        ss >> s.name >> s.flag >> s.rname >> s.pos >> s.mapq >> cigar >> s.rnext >> s.pnext >> tlen >> s.seq >> s.qual;
        out.push_back(s);
    }
    return out;
}

static void testBamAgainstSam(std::string const& dir)
{
    const auto sam = readSam(dir + "/multiparagraph/reads.sam");
    CHECK(sam.size() == 6);
    BamReader reader(dir + "/multiparagraph/reads.bam", "", dir + "/multiparagraph/dummy.fa");
    CHECK(reader.contigNames() == std::vector<std::string>{ "chr" } && reader.contigLengths()[0] == 160);
    CHECK(reader.headerText().find("@SQ\tSN:chr\tLN:160") != std::string::npos);
    for (const char* region : { "chr", "chr:1-160", "chr:40", "chr:40-40", "chr:89-200" })
    {
        reader.setRegion(region);
        Read r;
        size_t n = 0;
        while (reader.getAlign(r))
        {
            CHECK(n < sam.size());
            SamLine const& s = sam[std::min(n, sam.size() - 1)];
            CHECK(r.fragment_id() == s.name && r.bases() == s.seq && r.quals() == s.qual);
            CHECK(r.pos() == s.pos - 1 && r.mapq() == s.mapq && r.chrom_id() == 0);
            CHECK(r.is_first_mate() == ((s.flag & 0x40) != 0) && r.is_mapped() == !(s.flag & 4) && r.is_mate_mapped() == !(s.flag & 8));
            CHECK(r.is_reverse_strand() == ((s.flag & 0x10) != 0) && r.is_mate_reverse_strand() == ((s.flag & 0x20) != 0));
            CHECK(r.mate_chrom_id() == (s.rnext == "=" ? 0 : -1) && r.mate_pos() == s.pnext - 1);
            ++n;
        }
        CHECK(n == sam.size());
        CHECK(!reader.getAlign(r));  // stays exhausted
    }
    // reads span 40..89 (1-based): windows before / after see nothing
    Read r;
    reader.setRegion("chr:1-39");
    CHECK(!reader.getAlign(r));
    reader.setRegion("chr:90-160");
    CHECK(!reader.getAlign(r));
    CHECK_THROWS(reader.setRegion("nochr:1-10"));
    CHECK_THROWS(BamReader(dir + "/multiparagraph/missing.bam", "", ""));
}

// a BGZF block whose stored CRC-32 (or payload) does not match is refused: the block check runs on the carry-less-multiply
// path where the CPU has it, zlib's otherwise -- both must catch a flipped bit anywhere in a 64 KiB block
static void testBamCorruption(std::string const& dir)
{
    const std::string src = dir + "/chrX/chrX_graph_typing.bam";
    std::ifstream in(src, std::ios::binary);
    std::string data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    CHECK(data.size() > 1000);
    auto scan = [&](std::string const& path) {
        BamReader reader(path, src + ".bai", dir + "/chrX/chrX_graph_typing.fa");
        reader.setRegion("chrX");
        Read r;
        size_t n = 0;
        while (reader.getAlign(r))
            ++n;
        return n;
    };
    const size_t n_reads = scan(src);
    CHECK(n_reads > 100);
    // second block of the file: [header 12 + xlen][deflate data][crc32][isize]
    auto block_size = [&](size_t at) { return (size_t)((unsigned char)data[at + 16] | ((unsigned char)data[at + 17] << 8)) + 1; };
    const size_t b0 = block_size(0), b1 = block_size(b0);
    CHECK(b0 + b1 < data.size());
    const std::string tmp = "/tmp/pg_test_corrupt_" + std::to_string((long)getpid()) + ".bam";
    for (size_t where : { b0 + b1 - 8 /* stored CRC */, b0 + b1 - 12 /* last payload bytes */ })
    {
        std::string bad = data;
        bad[where] = (char)(bad[where] ^ 0x10);
        std::ofstream(tmp, std::ios::binary) << bad;
        CHECK_THROWS(scan(tmp));
    }
    std::ofstream(tmp, std::ios::binary) << data;
    CHECK(scan(tmp) == n_reads);
    std::remove(tmp.c_str());
}

// region queries through the index must equal a filter over the full scan, in the same order
static void testBamIndexConsistency(std::string const& dir)
{
    BamReader reader(dir + "/chrX/chrX_graph_typing.bam", "", dir + "/chrX/chrX_graph_typing.fa");
    CHECK(reader.contigNames().size() == 25);
    const size_t tid = (size_t)(std::find(reader.contigNames().begin(), reader.contigNames().end(), "chrX") - reader.contigNames().begin());
    CHECK(tid < reader.contigNames().size());
    std::vector<Read> all;
    reader.setRegion("chrX");
    Read r;
    while (reader.getAlign(r))
        all.push_back(r);
    CHECK(all.size() > 50);
    for (size_t i = 1; i < all.size(); ++i)
        CHECK(all[i - 1].pos() <= all[i].pos());
    std::mt19937 rng(5);
    const int64_t len = 10000;  // the reads of this file sit in the first 10 kbp of chrX
    size_t nonempty = 0;
    for (int iter = 0; iter < 200; ++iter)
    {
        const int64_t beg = (int64_t)(rng() % (uint64_t)len), span = 1 + (int64_t)(rng() % 600);
        reader.setRegion(formatPos("chrX", beg, beg + span - 1));
        std::vector<Read> got;
        while (reader.getAlign(r))
            got.push_back(r);
        // every returned record starts before the window end; every full-scan record starting inside is returned
        size_t k = 0;
        for (auto const& g : got)
        {
            CHECK(g.pos() < beg + span);
            while (k < all.size() && !(all[k] == g))
                ++k;
            CHECK(k < all.size());  // subsequence of the full scan, same order
        }
        for (auto const& a : all)
        {
            if (a.pos() >= beg && a.pos() < beg + span)
                CHECK(std::find(got.begin(), got.end(), a) != got.end());
            if (a.pos() + (int64_t)a.bases().size() + 600 < beg || a.pos() >= beg + span)
                CHECK(std::find(got.begin(), got.end(), a) == got.end());
        }
        nonempty += !got.empty();
    }
    CHECK(nonempty > 5);
    // mates: for paired records whose mate is in the file, getAlignedMate finds the other end
    size_t found = 0, tried = 0;
    for (auto const& a : all)
    {
        if (!a.is_mate_mapped() || a.mate_chrom_id() != a.chrom_id() || tried >= 40)
            continue;
        ++tried;
        Read mate;
        if (reader.getAlignedMate(a, mate))
        {
            ++found;
            CHECK(mate.fragment_id() == a.fragment_id() && mate.is_first_mate() != a.is_first_mate() && mate.pos() == a.mate_pos());
        }
    }
    CHECK(tried > 0 && found > 0);
}

class ScriptedReader : public ReadReader
{
public:
    std::deque<Read> aligns;
    std::map<std::string, Read> mates;
    std::vector<std::string> regions, mate_calls;
    void setRegion(const std::string& region) override { regions.push_back(region); }
    bool getAlign(Read& align) override
    {
        if (aligns.empty())
            return false;
        align = aligns.front();
        aligns.pop_front();
        return true;
    }
    bool getAlignedMate(const Read& read, Read& mate) override
    {
        mate_calls.push_back(read.fragment_id());
        auto it = mates.find(read.fragment_id());
        if (it == mates.end())
            return false;
        mate = it->second;
        return true;
    }
};

static Read makeRead(const char* id, const char* bases, bool first, int chrom, int pos, int mchrom = -1, int mpos = -1)
{
    Read r;
    r.setCoreInfo(id, bases, std::string(strlen(bases), '#'));
    r.set_is_first_mate(first);
    r.set_chrom_id(chrom);
    r.set_pos(pos);
    r.set_mate_chrom_id(mchrom);
    r.set_mate_pos(mpos);
    return r;
}

static void testExtraction()
{
    const Read read1 = makeRead("Fragment_1", "AAAA", true, 1, 100), read2 = makeRead("Fragment_2", "AAAA", true, 1, 100);
    {  // ExtractsAllReadsFromReader
        ScriptedReader reader;
        reader.aligns = { read1, read2 };
        ReadPairs pairs;
        const int mean_len = extractMappedReadsFromRegion(pairs, 10, reader, Region("1", 0, 1800));
        std::vector<Read> got;
        pairs.getReads(got);
        CHECK(got == (std::vector<Read>{ read1, read2 }) && mean_len == 4 && pairs.num_reads() == 2);
    }
    {  // ExtractsMaxAllowedReadsFromReader
        ScriptedReader reader;
        reader.aligns = { read1, read2 };
        ReadPairs pairs;
        extractMappedReadsFromRegion(pairs, 1, reader, Region("1", 0, 1800));
        std::vector<Read> got;
        pairs.getReads(got);
        CHECK(got == std::vector<Read>{ read1 } && reader.aligns.size() == 1);
    }
    {  // RecoversAnomalousMates
        const Read a = makeRead("Fragment_1", "AAAA", true, 1, 100, 1, 1600), b = makeRead("Fragment_2", "CCCC", true, 3, 500, 3, 800),
                   c = makeRead("Fragment_3", "AAAA", false, 5, 500, 3, 500);
        Read mate_a, mate_c;
        mate_a.setCoreInfo("Fragment_1", "TTTT", "####");
        mate_a.set_is_first_mate(false);
        mate_c.setCoreInfo("Fragment_3", "GGGG", "####");
        mate_c.set_is_first_mate(true);
        ScriptedReader reader;
        reader.mates["Fragment_1"] = mate_a;
        reader.mates["Fragment_3"] = mate_c;
        ReadPairs pairs;
        pairs.add(a);
        pairs.add(b);
        pairs.add(c);
        recoverMissingMates(reader, pairs);
        std::vector<Read> got;
        pairs.getReads(got);
        CHECK(got == (std::vector<Read>{ a, mate_a, b, mate_c, c }));
        CHECK(reader.mate_calls == (std::vector<std::string>{ "Fragment_1", "Fragment_3" }));  // the nearby mate of b is not looked up
        CHECK(pairs.num_reads() == 5 && pairs["Fragment_1"].second_mate() == mate_a);
        CHECK_THROWS(pairs["nope"]);
    }
    {  // isReadOrItsMateInRegion
        Read r = read1;
        CHECK(!isReadOrItsMateInRegion(r, Region("1", 0, 50)));
        CHECK(isReadOrItsMateInRegion(r, Region("1", 101, 103)));
        CHECK(!isReadOrItsMateInRegion(r, Region("1", 110, 200)));
        r.set_mate_chrom_id(1);
        r.set_mate_pos(1600);
        CHECK(isReadOrItsMateInRegion(r, Region("1", 1550, 1650)));
    }
    {  // extractReadsFromRegion: the scan window is region +- 3 x fragment length; mates only when reads are short enough
        const Read far = makeRead("F", "ACGTACGTAC", true, 0, 1000, 0, 9000);
        Read mate = makeRead("F", "TTTTTTTTTT", false, 0, 9000, 0, 1000);
        for (unsigned longest_insertion : { 0u, 5u, 6u })
        {
            ScriptedReader reader;
            reader.aligns = { far };
            reader.mates["F"] = mate;
            std::vector<p_Read> out;
            const auto n = extractReadsFromRegion(out, 100, reader, Region("chr", 990, 1100), longest_insertion, 333);
            CHECK(reader.regions == std::vector<std::string>{ "chr:1-2100" });
            const bool expect_mate = 10 <= longest_insertion * 2;
            CHECK(n.first == 1 && n.second == (expect_mate ? 1 : 0) && out.size() == (expect_mate ? 2u : 1u));
        }
        // a replaced mate slot does not count twice
        ReadPairs pairs;
        pairs.add(far);
        pairs.add(far);
        CHECK(pairs.num_reads() == 1);
    }
}

static void testGraphInput(std::string const& dir)
{
    const std::string fa = dir + "/basic/dummy.fa";
    auto load = [&](const char* name, bool store = true) { return grm::graphFromJson(Json::parseFile(dir + "/basic/" + name), fa, store); };
    {
        const auto g = load("del-with-edges-nodes.json");
        CHECK(g.numEdges() == 5 && g.numNodes() == 5);
        for (graphtools::NodeId n = 0; n < g.numNodes(); ++n)
            CHECK(!g.nodeSeq(n).empty());
        CHECK(g.nodeSeq(0) == "X" && g.nodeSeq(4) == "X" && g.nodeName(0) == "source");
        CHECK(g.nodeSeq(1) == "CGTCGACGTCGAACGATCGTCAGTACGACTACGTCGACAT");
        CHECK(g.edgeLabels(1, 2).count("REF") == 1 && g.edgeLabels(1, 3).count("ALT") == 1);
        const Json doc = Json::parseFile(dir + "/basic/del-with-edges-nodes.json");
        const auto paths = grm::pathsFromJson(&g, doc["paths"]);
        CHECK(paths.size() == 2 && paths.front().nodes.size() == 5 && paths.front().start_position == 0 && paths.front().end_position == 0);
    }
    {
        const auto g = load("del-with-edges-nodes.json", false);
        for (graphtools::NodeId n = 0; n < g.numNodes(); ++n)
            if (g.nodeName(n) != "source" && g.nodeName(n) != "sink")
                CHECK(g.nodeSeq(n).empty());
    }
    {
        const auto g = load("del-with-nodes-only.json");
        CHECK(g.numEdges() == 0 && g.numNodes() == 3);
    }
    {
        const auto g = load("del-with-ref-node-array.json");
        CHECK(g.numEdges() == 0 && g.numNodes() == 4);
        for (graphtools::NodeId n = 0; n < g.numNodes(); ++n)
            CHECK(!g.nodeSeq(n).empty());
    }
    CHECK_THROWS(load("del-with-no-ref-or-seq-node-key.json"));
    CHECK_THROWS(load("del-with-edges-only.json"));
    CHECK_THROWS(load("del-with-bad-edges-value.json"));
    CHECK_THROWS(load("del-with-bad-node-seq-ids.json"));
    CHECK_THROWS(load("del-with-duplicate-node-names.json"));
    // the chrX swap graph: explicit N-runs on source / sink become "X", the rest comes from the FASTA
    {
        const auto g = grm::graphFromJson(Json::parseFile(dir + "/chrX/chrX_graph_typing.2sample.json"), dir + "/chrX/chrX_graph_typing.fa");
        CHECK(g.numNodes() == 6 && g.numEdges() == 7 && g.nodeSeq(0) == "X" && g.nodeSeq(5) == "X");
        CHECK(g.nodeSeq(2).size() == 150 && g.nodeSeq(3).size() == 149 && g.nodeSeq(4).size() == 150 && g.nodeSeq(1).size() == 150);
        // the node-level "sequences": ["REF"] of LF and RF labels every edge in and out of them
        CHECK(g.edgeLabels(2, 3) == (std::set<std::string>{ "ALT", "REF" }) && g.edgeLabels(2, 4) == std::set<std::string>{ "REF" });
        CHECK(g.edgeLabels(0, 2) == std::set<std::string>{ "REF" } && g.edgeLabels(0, 1).empty() && g.edgeLabels(4, 5).empty());
    }
}

static void testGraphCoordinates()
{
    graphtools::Graph graph(4);
    const char* names[] = { "LF", "P1", "Q1", "RF" };
    const char* seqs[] = { "AAAAAAAAAAA", "TTTTTT", "GGGGGGGG", "AAAAAAAAAAA" };
    for (graphtools::NodeId n = 0; n < 4; ++n)
    {
        graph.setNodeName(n, names[n]);
        graph.setNodeSeq(n, seqs[n]);
    }
    graph.addEdge(0, 1);
    graph.addEdge(0, 2);
    graph.addEdge(1, 3);
    graph.addEdge(2, 3);
    graphtools::GraphCoordinates c(&graph);
    CHECK(c.canonicalPos(0, 6) == 6 && c.canonicalPos(1, 4) == 15 && c.canonicalPos(2, 3) == 20 && c.canonicalPos(3, 2) == 27);
    const uint64_t starts[] = { 0, 11, 17, 25 };
    for (graphtools::NodeId n = 0; n < 4; ++n)
        for (uint64_t j = 0; j < strlen(seqs[n]); ++j)
        {
            graphtools::NodeId node;
            uint64_t off;
            c.nodeAndOffset(starts[n] + j, node, off);
            CHECK(node == n && off == j);
        }
    CHECK(c.distance(10, 5) == 5 && c.distance(5, 10) == 5);
    CHECK(c.distance(14, 6) == 8 && c.distance(20, 6) == 8);
    CHECK(c.distance(2, 11 + 6 + 8 + 4) == 9 + 6 + 4);  // LF -> RF goes through the shorter P1
    CHECK(c.distance(12, 18) == graphtools::GraphCoordinates::kNoPath);  // P1 and Q1 are alternatives
    graphtools::Path p;
    p.graph = &graph;
    p.nodes = { 0, 1 };
    p.start_position = 3;
    p.end_position = 4;
    CHECK(c.canonicalStartAndEnd(p) == (std::pair<uint64_t, uint64_t>(3, 15)));
}

static void testStatistics()
{
    using paragraph::RunningStats;
    RunningStats empty;
    CHECK(std::isnan(empty.mean()) && empty.variance() == 0 && empty.median() == 0);
    RunningStats few;
    few.add(7);
    few.add(3);
    CHECK(few.mean() == 5 && few.median() == 0);  // fewer than five samples: the third seed slot, still unset
    CHECK(std::fabs(few.variance() - 4.0) < 1e-12);  // iterative update: 0 * 1/2 + (3 - 5)^2 / 1
    RunningStats many;
    for (int i = 1; i <= 1001; ++i)
        many.add((double)((i * 37) % 1001));
    CHECK(std::fabs(many.mean() - 500.0) < 1e-9);
    CHECK(std::fabs(many.median() - 500.0) < 25.0);
    CHECK(std::fabs(many.variance() - (1001.0 * 1001.0 - 1.0) / 12.0) / 83500.0 < 0.01);

    graphtools::Graph g(4);
    const char* names[] = { "source", "A", "B", "sink" };
    const char* seqs[] = { "X", "ACGTACGTAC", "TTTTTTTTTTTTTTTTTTTT", "X" };
    for (graphtools::NodeId n = 0; n < 4; ++n)
    {
        g.setNodeName(n, names[n]);
        g.setNodeSeq(n, seqs[n]);
    }
    g.addEdge(0, 1);
    g.addEdge(1, 2);
    g.addEdge(2, 3);
    g.addLabelToEdge(0, 1, "REF");
    g.addLabelToEdge(1, 2, "REF");
    g.addLabelToEdge(2, 3, "REF");
    const auto pieces = paragraph::decodeGraphCigar("1[2S3M1X4M]2[5M1I2M1D3M4S]", g);
    CHECK(pieces.size() == 2 && pieces[0].node == 1 && pieces[0].matched == 7 && pieces[0].mismatched == 1 && pieces[0].clipped == 2);
    CHECK(pieces[1].referenceLength() == 11 && pieces[1].queryLength() == 15 && pieces[0].queryLength() == 10);
    CHECK_THROWS(paragraph::decodeGraphCigar("9[3M]", g));
    CHECK_THROWS(paragraph::decodeGraphCigar("1[3M", g));
    CHECK_THROWS(paragraph::decodeGraphCigar("1[3Q]", g));

    Read r1("f1", std::string(25, 'A'), std::string(25, '#')), r2("f1", std::string(10, 'T'), std::string(10, '#'));
    r1.set_graph_mapping_status(Read::MAPPED);
    r1.set_graph_pos(0);
    r1.set_graph_cigar("1[2S3M1X4M]2[5M1I2M1D3M4S]");
    r1.set_graph_alignment_score(11);
    r1.add_graph_sequences_supported("REF");
    r1.set_is_mapped(true);
    r1.set_is_mate_mapped(true);
    r1.set_chrom_id(0);
    r1.set_mate_chrom_id(0);
    r1.set_pos(100);
    r1.set_mate_pos(300);
    r1.set_is_mate_reverse_strand(true);
    r2.set_is_first_mate(false);
    r2.set_graph_mapping_status(Read::MAPPED);
    r2.set_graph_pos(12);
    r2.set_graph_cigar("2[8M]");
    r2.set_is_graph_reverse_strand(true);
    r2.set_is_mapped(true);
    r2.set_is_mate_mapped(true);
    r2.set_chrom_id(0);
    r2.set_mate_chrom_id(0);
    r2.set_pos(300);
    r2.set_mate_pos(100);
    r2.set_is_reverse_strand(true);
    Read lone("f2", "ACGT", "####");
    lone.set_graph_mapping_status(Read::MAPPED);
    lone.set_graph_pos(1);
    lone.set_graph_cigar("1[4M]");
    const std::vector<Read const*> reads = { &r1, &r2, &lone };
    const Json fs = paragraph::fragmentStatistics(g, reads);
    CHECK(fs["paired_read"].asUInt64() == 1 && fs["single_read"].asUInt64() == 1 && fs["multi_read"].asUInt64() == 0);
    CHECK(fs["problematic_linear"].asUInt64() == 1 && fs["problematic_graph"].asUInt64() == 0);
    CHECK(fs["mean_linear"].asDouble() == 210.0);  // |300 - 100| + 10 bases of the read added last
    // r1 spans canonical 1+0 .. 11+10 (its LAST aligned base: decodeGraphAlignment's path end is inclusive,
    // GT!/src/graphalign/GraphAlignmentOperations.cpp:103), r2 starts at 11+12: distance 21 -> 23 is 2 -> 25 + 8 + 2
    // (the arithmetic itself is compared with the reference's compiled GraphCoordinates in tests/test_counts_oracle.py)
    CHECK(fs["mean_graph"].asDouble() == 35.0);
    CHECK(fs["median_graph"].asDouble() == 0.0 && fs["variance_graph"].asDouble() == 0.0);
    const Json as = paragraph::alignmentStatistics(g, reads);
    CHECK(as["nodes"]["A"]["num_fwd_reads"].asInt64() == 2 && as["nodes"]["B"]["num_rev_reads"].asInt64() == 1);
    CHECK(as["nodes"]["A"]["contig_length"].asInt64() == 10 && as["edges"]["A_B"]["contig_length"].asInt64() == 30);
    CHECK(std::fabs(as["nodes"]["A"]["mismatch_rate"].asDouble() - 1.0 / 12.0) < 1e-12);  // (7 + 4) M + 1 X
    CHECK(std::fabs(as["nodes"]["A"]["clip_rate"].asDouble() - 2.0 / 12.0) < 1e-12);
    CHECK(as["edges"]["A_B"]["clip_rate"].asDouble() == 0.0);  // clips on an edge only count next to the terminals
    CHECK(as["alleles"]["REF"]["avr_score"].asDouble() == 11.0 && as["alleles"]["REF"]["contig_length"].asInt64() == 30);
}

static void testManifestAndParameters(std::string const& dir)
{
    const std::string path = "/tmp/pg_hostio_manifest.txt";
    {
        std::ofstream out(path);
        out << "#id\tpath\tdepth\tread length\tdepth sd\tsex\n";
        out << "SAMPLE1\t" << dir << "/chrX/chrX_graph_typing.bam\t44.2\t150\t20\tmale\n\n";
        out << "SAMPLE2\t" << dir << "/chrX/chrX_graph_typing.bam\t30\t100\t5.5\tFemale\n";
    }
    const auto samples = genotyping::loadManifest(path);
    CHECK(samples.size() == 2 && samples[0].sample_name() == "SAMPLE1" && samples[0].autosome_depth() == 44.2);
    CHECK(samples[0].read_length() == 150 && samples[0].depth_sd() == 20 && samples[0].sex() == genotyping::Sex::MALE);
    CHECK(samples[1].sex() == genotyping::Sex::FEMALE && samples[1].depth_sd() == 5.5 && samples[1].index_filename().empty());
    {
        std::ofstream out(path);
        out << "id,path,depth,read length\nS," << dir << "/chrX/chrX_graph_typing.bam,20,100\n";
    }
    const auto plain = genotyping::loadManifest(path);
    CHECK(plain.size() == 1 && plain[0].depth_sd() == std::sqrt(100.0) && plain[0].sex() == genotyping::Sex::UNKNOWN);
    {
        std::ofstream out(path);
        out << "id\tpath\tcolour\nS\tx\tred\n";
    }
    CHECK_THROWS(genotyping::loadManifest(path));
    {
        std::ofstream out(path);
        out << "id\tpath\tdepth\tread length\nS\t/no/such.bam\t20\t100\n";
    }
    CHECK_THROWS(genotyping::loadManifest(path));
    {
        std::ofstream out(path);
        out << "id\tpath\tdepth\n";
    }
    CHECK_THROWS(genotyping::loadManifest(path));
    std::remove(path.c_str());

    genotyping::GenotypingParameters p({ "ALT", "REF" }, 2);
    p.setFromJson(Json::parseFile(dir + "/chrX/param.json"));
    CHECK(p.minOverlapBases() == 16 && p.ploidy() == 2 && p.otherAlleleErrorRate() == 0.05);
    CHECK(p.alleleErrorRates() == (std::vector<double>{ 0.04, 0.04 }));       // given as REF, ALT -> stored as ALT, REF
    CHECK(p.hetHaplotypeFractions() == (std::vector<double>{ 0.45, 0.5 }));
    CHECK(p.coverageTestCutoff().first == 0.0001);  // both values land in the lower cutoff
    CHECK_THROWS(p.setFromJson(Json::parse(R"({"use_poisson_depth": true})")));
    p.setFromJson(Json::parse(R"({"use_poisson_depth": "true"})"));
    CHECK(p.usePoissonDepth());
    CHECK_THROWS(p.setFromJson(Json::parse(R"({"allele_error_rates": [0.1]})")));
}

// extractPacked must keep exactly the reads extractReads keeps, in the same order, with the same fragment grouping
static void comparePackedWithObjects(paragraph::PackedSite const& packed, std::vector<p_Read> const& objects, const char* what)
{
    CHECK(packed.size() == objects.size());
    if (packed.size() != objects.size())
    {
        std::cerr << what << ": " << packed.size() << " packed vs " << objects.size() << " objects\n";
        return;
    }
    std::map<std::string, uint32_t> fragment_of;
    for (size_t k = 0; k < objects.size(); ++k)
    {
        Read const& r = *objects[k];
        const uint32_t begin = k ? packed.base_end[k - 1] : 0;
        CHECK(packed.bases.substr(begin, packed.base_end[k] - begin) == r.bases());
        CHECK(packed.fragment[k] == fragment_of.emplace(r.fragment_id(), (uint32_t)fragment_of.size()).first->second);
        const uint8_t f = packed.flags[k];
        CHECK(((f & paragraph::PackedSite::REVERSE) != 0) == r.is_reverse_strand() && ((f & paragraph::PackedSite::FIRST_MATE) != 0) == r.is_first_mate());
        CHECK(((f & paragraph::PackedSite::MAPPED) != 0) == r.is_mapped() && ((f & paragraph::PackedSite::MATE_MAPPED) != 0) == r.is_mate_mapped());
        CHECK(((f & paragraph::PackedSite::MATE_REVERSE) != 0) == r.is_mate_reverse_strand());
        CHECK(packed.chrom_id[k] == r.chrom_id() && packed.pos[k] == r.pos() && packed.mate_chrom_id[k] == r.mate_chrom_id()
              && packed.mate_pos[k] == r.mate_pos());
    }
}

static void testPackedExtraction(std::string const& dir)
{
    // real data: both target regions of the chrX swap graph (reads near both are extracted twice), with and without a cap
    for (int max_reads : { 10000, 57, 1 })
        for (unsigned longest_insertion : { 0u, 10u, 200u })
        {
            BamReader reader(dir + "/chrX/chrX_graph_typing.bam", "", "");
            const std::list<Region> regions = { Region("chrX:850-1149"), Region("chrX:8667-8965"), Region("chrX:900-1000") };
            std::vector<p_Read> objects;
            extractReads(reader, regions, max_reads, longest_insertion, objects);
            paragraph::PackedSite packed;
            paragraph::extractPacked(reader, regions, max_reads, longest_insertion, packed);
            CHECK(!objects.empty());
            comparePackedWithObjects(packed, objects, "chrX");
        }
    // scripted: replaced mate slots, an empty-sequence record, far mates to recover
    auto script = [](ScriptedReader& reader) {
        Read a1 = makeRead("A", "ACGTACGTAC", true, 0, 1000, 0, 1200), a1_again = makeRead("A", "TTTTTTTTTT", true, 0, 1001, 0, 1200);
        Read a2 = makeRead("A", "GGGGGGGGGG", false, 0, 1200, 0, 1000), empty = makeRead("E", "", true, 0, 1010, 0, 1300);
        Read far = makeRead("F", "CCCCCCCCCC", true, 0, 1020, 0, 9000), far2 = makeRead("G", "CCCCCCCCAA", false, 0, 1030, 3, 50);
        far.set_is_mate_mapped(true);
        far2.set_is_mate_mapped(true);
        reader.aligns = { a1, a2, empty, far, a1_again, far2 };
        reader.mates["F"] = makeRead("F", "AAAAAAAAAA", false, 0, 9000, 0, 1020);
        reader.mates["G"] = makeRead("G", "AAAAAAAATT", true, 3, 50, 0, 1030);
    };
    for (int max_reads : { 100, 3 })
    {
        ScriptedReader r1, r2;
        script(r1);
        script(r2);
        std::vector<p_Read> objects;
        extractReads(r1, { Region("chr", 990, 1100) }, max_reads, 50, objects);
        paragraph::PackedSite packed;
        paragraph::extractPacked(r2, { Region("chr", 990, 1100) }, max_reads, 50, packed);
        comparePackedWithObjects(packed, objects, "scripted");
        CHECK(r1.mate_calls == r2.mate_calls && r1.regions == r2.regions);
        if (max_reads == 100)
            CHECK(objects.size() == 6 && r1.mate_calls.size() == 2);  // A x2 (later first mate wins), F x2, G x2; E is not a read
    }
    paragraph::PackedSite not_empty;
    not_empty.fragment.push_back(0);
    ScriptedReader r;
    CHECK_THROWS(paragraph::extractPacked(r, { Region("chr", 1, 2) }, 10, 0, not_empty));
}

static void testChunkSchedule()
{
    for (size_t n : { 0u, 1u, 7u, 512u, 513u, 4096u, 10000u })
        for (size_t per : { 1u, 2u, 64u, 512u })
            for (size_t lanes : { 1u, 3u, 8u })
            {
                const auto ranges = grmpy::chunkSchedule(n, per, lanes);
                size_t at = 0;
                for (auto const& r : ranges)
                {
                    CHECK(r.first == at && r.second > r.first && r.second - r.first <= per);
                    at = r.second;
                }
                CHECK(at == n);
                const size_t even = (n + per - 1) / per;
                if (lanes == 1 || even <= lanes)
                    CHECK(ranges.size() == even);
                else
                {
                    // staggered first round, then full chunks, then a tail of at least a quarter chunk (but the very last)
                    for (size_t k = 0; k + 1 < std::min(lanes, ranges.size()); ++k)
                        CHECK(ranges[k].second - ranges[k].first <= ranges[k + 1].second - ranges[k + 1].first);
                    for (size_t k = lanes; k + 1 < ranges.size(); ++k)
                        CHECK(ranges[k].second - ranges[k].first >= std::max<size_t>(1, per / 4));
                }
            }
    const auto ten_k = grmpy::chunkSchedule(10000, 512, 8);
    CHECK(ten_k[0].second == 64 && ten_k[7].second - ten_k[7].first == 512);
}

// The BGZF block decoder (host/src/inflate.hh) against zlib: streams of every block type (stored, fixed, dynamic), sizes from
// empty to 64 KiB, texts from one repeated byte over BAM-like records to noise, several compression levels and strategies,
// concatenated blocks (Z_FULL_FLUSH) -- and corrupted / truncated streams, which must be refused, never overrun a buffer.
static void testInflate()
{
    std::mt19937_64 rng(20260927);
    auto deflateRaw = [](std::vector<unsigned char> const& text, int level, int strategy, bool flush_blocks) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        CHECK(deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy) == Z_OK);
        std::vector<unsigned char> out(deflateBound(&zs, (uLong)text.size()) + 64 + text.size() / 100);
        zs.next_out = out.data();
        zs.avail_out = (uInt)out.size();
        size_t at = 0;
        while (flush_blocks && at + 5000 < text.size())
        {
            zs.next_in = const_cast<unsigned char*>(text.data() + at);
            zs.avail_in = 5000;
            CHECK(deflate(&zs, Z_FULL_FLUSH) == Z_OK);
            at += 5000;
        }
        zs.next_in = const_cast<unsigned char*>(text.data() + at);
        zs.avail_in = (uInt)(text.size() - at);
        CHECK(deflate(&zs, Z_FINISH) == Z_STREAM_END);
        out.resize(zs.total_out);
        deflateEnd(&zs);
        return out;
    };
    auto makeText = [&](int kind, size_t n) {
        std::vector<unsigned char> t(n);
        switch (kind)
        {
        case 0:  // one byte
            std::fill(t.begin(), t.end(), (unsigned char)'A');
            break;
        case 1:  // noise
            for (auto& c : t)
                c = (unsigned char)rng();
            break;
        case 2:  // BAM-like: short binary headers, names, 4-bit bases, qualities from a small alphabet
            for (size_t i = 0; i < n; ++i)
            {
                const size_t r = i % 280;
                t[i] = r < 36 ? (unsigned char)((i / 280) >> (r % 3)) : r < 50 ? (unsigned char)("s123_f4567890"[r % 13]) : r < 125
                        ? (unsigned char)(((rng() & 3) == 0 ? 1 : (rng() & 3) == 1 ? 2 : (rng() & 1) ? 4 : 8) * 17)
                        : (unsigned char)(30 + rng() % 12);
            }
            break;
        case 3:  // short period (offsets < 8: the overlapping-copy path)
            for (size_t i = 0; i < n; ++i)
                t[i] = (unsigned char)("ACGTTGA"[i % (1 + (i / 997) % 7)]);
            break;
        default:  // text with long-range repeats
            for (size_t i = 0; i < n; ++i)
                t[i] = i >= 3000 && (rng() % 8) ? t[i - 3000 + (i % 7 == 0)] : (unsigned char)("ACGTN"[rng() % 5]);
        }
        return t;
    };
    size_t streams = 0, kinds_seen[3] = { 0, 0, 0 };
    for (size_t n : { (size_t)0, (size_t)1, (size_t)2, (size_t)7, (size_t)63, (size_t)300, (size_t)301, (size_t)4096, (size_t)20000, (size_t)65280, (size_t)65536 })
        for (int kind = 0; kind < 5; ++kind)
            for (int cfg = 0; cfg < 6; ++cfg)
            {
                static const int levels[6] = { 0, 1, 6, 9, 6, 6 };
                static const int strategies[6] = { Z_DEFAULT_STRATEGY, Z_DEFAULT_STRATEGY, Z_DEFAULT_STRATEGY, Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY };
                const std::vector<unsigned char> text = makeText(kind, n);
                std::vector<unsigned char> comp = deflateRaw(text, levels[cfg], strategies[cfg], kind == 4 && cfg == 2);
                if (!comp.empty())
                    ++kinds_seen[(comp[0] >> 1) & 3 ? ((comp[0] >> 1) & 3) - 0 > 2 ? 0 : (comp[0] >> 1) & 3 : 0];
                const size_t clen = comp.size();
                comp.resize(clen + 8, 0xAB);
                std::vector<unsigned char> out(n + 16, 0xCD);
                const int rc = pginflate::inflateBlock(comp.data(), clen, out.data(), n);
                CHECK(rc == pginflate::kOk);
                CHECK(n == 0 || memcmp(out.data(), text.data(), n) == 0);
                ++streams;
                if (n == 0)
                    continue;
                // wrong expected size: one byte short / one byte long must be reported, not written past
                std::vector<unsigned char> small(n - 1 + 16, 0xCD);
                CHECK(pginflate::inflateBlock(comp.data(), clen, small.data(), n - 1) != pginflate::kOk);
                for (size_t k = n - 1 + 16; k-- > n - 1 + 0 && k >= n - 1 + 16;)
                    CHECK(small[k] == 0xCD);
                std::vector<unsigned char> large(n + 1 + 16, 0xCD);
                CHECK(pginflate::inflateBlock(comp.data(), clen, large.data(), n + 1) != pginflate::kOk);
                // truncated and bit-flipped streams: any answer but a crash / an overrun; "ok" only with the right bytes
                for (int trial = 0; trial < 6; ++trial)
                {
                    std::vector<unsigned char> bad(comp.begin(), comp.begin() + (std::ptrdiff_t)clen);
                    size_t blen = clen;
                    if (trial < 2)
                        blen = clen > 1 ? (size_t)(rng() % clen) : 0;
                    else
                        bad[rng() % clen] ^= (unsigned char)(1u << (rng() % 8));
                    bad.resize(blen);
                    bad.resize(blen + 8, 0xEE);
                    std::vector<unsigned char> o2(n + 16, 0xCD);
                    const int r2 = pginflate::inflateBlock(bad.data(), blen, o2.data(), n);
                    // the reference decoder on the same input
                    std::vector<unsigned char> oz(n + 1);
                    z_stream zi;
                    memset(&zi, 0, sizeof zi);
                    inflateInit2(&zi, -15);
                    zi.next_in = bad.data();
                    zi.avail_in = (uInt)blen;
                    zi.next_out = oz.data();
                    zi.avail_out = (uInt)(n + 1);
                    const int rz = inflate(&zi, Z_FINISH);
                    const bool z_ok = rz == Z_STREAM_END && zi.total_out == n && zi.avail_in == 0;
                    inflateEnd(&zi);
                    if (z_ok)
                        CHECK(r2 == pginflate::kOk && memcmp(o2.data(), oz.data(), n) == 0);
                    else if (r2 == pginflate::kOk)  // zlib also insists that no input is left over; otherwise the bytes must agree
                        CHECK(rz == Z_STREAM_END && zi.total_out == n && memcmp(o2.data(), oz.data(), n) == 0);
                    for (size_t k = n; k < n + 16; ++k)
                        CHECK(o2[k] == 0xCD);
                }
            }
    CHECK(streams == 11 * 5 * 6);
}

int main(int argc, char** argv)
{
    if (argc == 4 && std::string(argv[1]) == "--dump-bam")
    {
        // <bam> <region>: one line per primary record, for the independent Python decoder in tests/test_hostio_cpu.py
        try
        {
        BamReader reader(argv[2], "", "");
        reader.setRegion(argv[3]);
        Read r;
        while (reader.getAlign(r))
            std::cout << r.fragment_id() << "\t" << r.chrom_id() << "\t" << r.pos() << "\t" << (int)r.mapq() << "\t" << r.is_mapped()
                      << r.is_first_mate() << r.is_mate_mapped() << r.is_reverse_strand() << r.is_mate_reverse_strand() << "\t"
                      << r.mate_chrom_id() << "\t" << r.mate_pos() << "\t" << r.bases() << "\t" << r.quals() << "\n";
        }
        catch (std::exception const& e)
        {
            std::cerr << "error: " << e.what() << "\n";
            return 1;
        }
        return 0;
    }
    if (argc == 4 && std::string(argv[1]) == "--load-graph")
    {
        // <graph.json> <reference.fa>: parse + build, for robustness checks on damaged descriptions (exit 1 = clean refusal)
        try
        {
            const Json doc = Json::parseFile(argv[2]);
            const auto g = grm::graphFromJson(doc, argv[3]);
            const auto paths = grm::pathsFromJson(&g, (doc.isMember("graph") ? doc["graph"] : doc)["paths"]);
            std::cout << g.numNodes() << " nodes, " << g.numEdges() << " edges, " << paths.size() << " paths\n";
        }
        catch (std::exception const& e)
        {
            std::cerr << "error: " << e.what() << "\n";
            return 1;
        }
        return 0;
    }
    if (argc == 4 && std::string(argv[1]) == "--alignment-statistics")
    {
        // <graph.json> <reference.fa>; stdin: "pos cigar reverse(0|1) score seq1,seq2|-" per MAPPED read -> the
        // "alignment_statistics" object as paragraph::alignmentStatistics builds it, for the comparison with the reference's
        // own summarizeAlignments in tests/test_counts_oracle.py
        try
        {
            const Json doc = Json::parseFile(argv[2]);
            const auto g = grm::graphFromJson(doc, argv[3]);
            std::deque<Read> store;
            std::string line;
            while (std::getline(std::cin, line))
            {
                std::istringstream in(line);
                int pos = 0, reverse = 0, score = 0;
                std::string cigar, seqs;
                if (!(in >> pos >> cigar >> reverse >> score >> seqs))
                    continue;
                store.emplace_back("f" + std::to_string(store.size()), "A", "#");
                Read& r = store.back();
                r.set_graph_mapping_status(Read::MAPPED);
                r.set_graph_pos(pos);
                r.set_graph_cigar(cigar);
                r.set_is_graph_reverse_strand(reverse != 0);
                r.set_graph_alignment_score(score);
                if (seqs != "-")
                {
                    std::stringstream ss(seqs);
                    std::string one;
                    while (std::getline(ss, one, ','))
                        r.add_graph_sequences_supported(one);
                }
            }
            std::vector<Read const*> reads;
            for (Read const& r : store)
                reads.push_back(&r);
            std::cout << paragraph::alignmentStatistics(g, reads).dump() << "\n";
        }
        catch (std::exception const& e)
        {
            std::cerr << "error: " << e.what() << "\n";
            return 1;
        }
        return 0;
    }
    if (argc == 4 && std::string(argv[1]) == "--pair-lengths")
    {
        // <graph.json> <reference.fa>; stdin: "pos1 cigar1 pos2 cigar2" per line -> graph length of the two-read fragment as
        // paragraph::fragmentStatistics sees it ("-" = no path), for the comparison with the reference's Fragment /
        // GraphCoordinates in tests/test_counts_oracle.py
        try
        {
            const Json doc = Json::parseFile(argv[2]);
            const auto g = grm::graphFromJson(doc, argv[3]);
            std::string line;
            while (std::getline(std::cin, line))
            {
                std::istringstream in(line);
                int pos1 = 0, pos2 = 0;
                std::string c1, c2;
                if (!(in >> pos1 >> c1 >> pos2 >> c2))
                    continue;
                Read r1("f", "A", "#"), r2("f", "A", "#");
                r1.set_graph_mapping_status(Read::MAPPED);
                r2.set_graph_mapping_status(Read::MAPPED);
                r1.set_graph_pos(pos1);
                r1.set_graph_cigar(c1);
                r2.set_graph_pos(pos2);
                r2.set_graph_cigar(c2);
                const std::vector<Read const*> reads = { &r1, &r2 };
                const Json fs = paragraph::fragmentStatistics(g, reads);
                if (fs["problematic_graph"].asUInt64())
                    std::cout << "-\n";
                else
                {
                    char text[64];
                    snprintf(text, sizeof text, "%.17g", fs["mean_graph"].asDouble());
                    std::cout << text << "\n";
                }
            }
        }
        catch (std::exception const& e)
        {
            std::cerr << "error: " << e.what() << "\n";
            return 1;
        }
        return 0;
    }
    if (argc < 2)
    {
        std::cerr << "usage: test_hostio <tests/golden/sites> | --dump-bam <bam> <region> | --load-graph <json> <fasta>\n";
        return 2;
    }
    const std::string dir = argv[1];
    const std::pair<const char*, std::function<void()>> tests[] = {
        { "json", testJson },
        { "json numbers", testJsonNumbers },
        { "inflate", testInflate },
        { "coordinates", testCoordinates },
        { "fasta", [&] { testFasta(dir); } },
        { "bam-vs-sam", [&] { testBamAgainstSam(dir); } },
        { "bam-index", [&] { testBamIndexConsistency(dir); } },
        { "bam-corruption", [&] { testBamCorruption(dir); } },
        { "extraction", testExtraction },
        { "packed-extraction", [&] { testPackedExtraction(dir); } },
        { "chunk-schedule", [&] { testChunkSchedule(); } },
        { "graph-input", [&] { testGraphInput(dir); } },
        { "graph-coordinates", testGraphCoordinates },
        { "statistics", testStatistics },
        { "manifest-parameters", [&] { testManifestAndParameters(dir); } },
    };
    for (auto const& t : tests)
    {
        const int before = g_failures;
        try
        {
            t.second();
        }
        catch (std::exception const& e)
        {
            std::cerr << t.first << ": unexpected exception: " << e.what() << "\n";
            ++g_failures;
        }
        std::cout << (g_failures == before ? "ok   " : "FAIL ") << t.first << "\n";
    }
    std::cout << (g_failures ? "FAILED" : "ALL OK") << "\n";
    return g_failures ? 1 : 0;
}
