// GPU end-to-end checks of the host workflow (BAM -> read extraction -> batched realignment -> count documents ->
// genotypes) against the reference's own expected outputs:
//   share/test-data/multiparagraph/expected.json   (`paragraph` count documents, src/python/test/test_multiparagraph.py:76-105)
//   Grmpy.GenotypesSingleSwap                      (src/c++/test-blackbox/test_grm.cpp:33-65)
// Usage: test_workflow <tests/golden/sites directory>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>

#include "common/BamReader.hh"
#include "common/ReadExtraction.hh"
#include "paragraph/Workflow.hh"

using common::Json;

static int failures = 0;
#define CHECK(cond)                                                                  \
    do                                                                               \
    {                                                                                \
        if (!(cond))                                                                 \
        {                                                                            \
            std::cerr << __FILE__ << ":" << __LINE__ << ": CHECK failed: " #cond "\n"; \
            ++failures;                                                              \
        }                                                                            \
    } while (0)

// expected documents went through Python's json module: NaN statistics are null there and here
static bool sameValue(Json const& want, Json const& got)
{
    if (want.isNull() && got.kind() == Json::REAL)
        return std::isnan(got.asDouble());
    if (want.isNumber() && got.isNumber())
        return std::fabs(want.asDouble() - got.asDouble()) <= 1e-9 * std::max(1.0, std::fabs(want.asDouble()));
    return want == got;
}

static void compareObject(std::string const& what, Json const& want, Json const& got)
{
    for (auto const& kv : want.members())
    {
        if (!got.isMember(kv.first))
        {
            std::cerr << what << ": missing key " << kv.first << "\n";
            ++failures;
        }
        else if (kv.second.isObject())
            compareObject(what + "/" + kv.first, kv.second, got[kv.first]);
        else if (!sameValue(kv.second, got[kv.first]))
        {
            std::cerr << what << "/" << kv.first << ": expected " << kv.second.dump() << " got " << got[kv.first].dump() << "\n";
            ++failures;
        }
    }
    for (auto const& kv : got.members())
        if (!want.isMember(kv.first))
        {
            std::cerr << what << ": unexpected key " << kv.first << " = " << kv.second.dump() << "\n";
            ++failures;
        }
}

static void testMultiparagraph(std::string const& dir)
{
    const std::string base = dir + "/multiparagraph/";
    const Json expected = Json::parseFile(base + "expected.json");
    CHECK(expected.isArray() && expected.size() == 5);
    static const char* kOutputs[] = { "fragment_statistics", "read_counts_by_edge", "read_counts_by_node", "read_counts_by_sequence" };
    std::vector<paragraph::GraphDescription> graphs;
    std::vector<common::ReadBuffer> reads(expected.size());
    for (size_t i = 0; i < expected.size(); ++i)
    {
        Json spec = expected[i]["graph"];
        for (const char* key : kOutputs)
            spec.removeMember(key);
        graphs.push_back(paragraph::GraphDescription::fromJson(spec, base + "dummy.fa"));
        common::extractReads(
            base + "reads.bam", "", base + "dummy.fa", graphs.back().target_regions, 10000, (unsigned)graphs.back().longest_alt_insertion,
            reads[i]);
        CHECK(reads[i].size() >= 4);  // one pass per target region: a read near two regions is extracted twice, as in the original
    }
    std::vector<paragraph::SiteInput> sites(expected.size());
    for (size_t i = 0; i < expected.size(); ++i)
    {
        sites[i].description = &graphs[i];
        sites[i].reads = &reads[i];
    }
    paragraph::Parameters parameters;  // the `paragraph` tool's defaults: exact path matching first, then gssw
    parameters.threads = 3;
    const std::vector<Json> documents = paragraph::alignAndDisambiguateBatch(parameters, sites);
    for (size_t i = 0; i < expected.size(); ++i)
    {
        const std::string what = "multiparagraph[" + expected[i]["desc"].asString() + "]";
        for (const char* key : kOutputs)
        {
            CHECK(documents[i].isMember(key));
            compareObject(what + "/" + key, expected[i]["graph"][key], documents[i][key]);
        }
        for (const char* key : { "nodes", "edges", "paths", "sequencenames", "target_regions" })
            CHECK(documents[i][key] == expected[i]["graph"][key]);
        CHECK(documents[i]["reference"].asString() == base + "dummy.fa" && documents[i].isMember("alignment_statistics"));
        CHECK(!documents[i].isMember("alignments"));
    }
    // the packed form (no common::Read objects) yields the same documents, with and without the path stage, incl. the
    // per-family node / edge breakdown
    for (int pass = 0; pass < 3; ++pass)
    {
        paragraph::Parameters pp = parameters;
        pp.path_sequence_matching = pass != 1;
        pp.kmer_sequence_matching = pp.klib_sequence_matching = pass == 2;  // the whole cascade
        pp.output_options_ |= paragraph::Parameters::DETAILED_READ_COUNTS | paragraph::Parameters::FILTERED_ALIGNMENTS;
        std::vector<paragraph::PackedSite> packed(expected.size());
        std::vector<common::ReadBuffer> objects(expected.size());
        std::vector<paragraph::PackedSiteInput> packed_sites(expected.size());
        std::vector<paragraph::SiteInput> object_sites(expected.size());
        for (size_t i = 0; i < expected.size(); ++i)
        {
            common::BamReader reader(base + "reads.bam", "", "");
            paragraph::extractPacked(reader, graphs[i].target_regions, 10000, (unsigned)graphs[i].longest_alt_insertion, packed[i]);
            common::extractReads(reader, graphs[i].target_regions, 10000, (unsigned)graphs[i].longest_alt_insertion, objects[i]);
            CHECK(packed[i].size() == objects[i].size());
            for (size_t k = 0; k < packed[i].size() && k < objects[i].size(); ++k)
                CHECK(packed[i].readLength(k) == objects[i][k]->bases().size() && packed[i].pos[k] == objects[i][k]->pos());
            packed_sites[i].description = &graphs[i];
            packed_sites[i].reads = &packed[i];
            object_sites[i].description = &graphs[i];
            object_sites[i].reads = &objects[i];
        }
        const std::vector<Json> from_packed = paragraph::alignAndDisambiguateBatch(pp, packed_sites);
        std::vector<Json> from_objects = paragraph::alignAndDisambiguateBatch(pp, object_sites);
        for (size_t i = 0; i < expected.size(); ++i)
        {
            // FILTERED_ALIGNMENTS: the object form also lists the reads the filter chain rejected, each with the filter's message
            // (Disambiguation.cpp:183-203) -- as many of each kind as the tallies say; the packed form keeps no records
            {
                const Json rejected = from_objects[i]["alignments"];
                from_objects[i].removeMember("alignments");
                std::map<std::string, uint64_t> by_filter;
                for (Json const& r : rejected.elements())
                {
                    CHECK(r.isMember("error") && r["graphMappingStatus"].asString() == "BAD_ALIGN" && !r.isMember("graphNodesSupported"));
                    ++by_filter["read_filter_" + r["error"].asString()];
                }
                Json const& st = from_objects[i]["alignment_statistics"];
                for (const char* key : { "read_filter_bad_align", "read_filter_nonuniq" })
                    CHECK(by_filter[key] == (st.isMember(key) ? (uint64_t)st[key].asUInt64() : 0));
                CHECK(objects[i].size() >= rejected.size());  // (the read buffer holds the rejected reads too, as the original's does)
            }
            CHECK(from_packed[i] == from_objects[i]);
            if (from_packed[i] != from_objects[i])
                compareObject("packed-vs-objects[" + std::to_string(i) + "]", from_objects[i], from_packed[i]);
            if (pass < 2)  // expected.json is a path + gssw result; the seed stages place a few more reads
                compareObject("packed/edges", expected[i]["graph"]["read_counts_by_edge"], from_packed[i]["read_counts_by_edge"]);
        }
    }
    // gssw only (grmpy's default cascade) reaches the same tables on these reads
    for (auto& r : reads)
        r.clear();
    for (size_t i = 0; i < expected.size(); ++i)
        common::extractReads(base + "reads.bam", "", "", graphs[i].target_regions, 10000, 0, reads[i]);
    parameters.path_sequence_matching = false;
    parameters.output_options_ |= paragraph::Parameters::ALIGNMENTS;
    const std::vector<Json> gssw_only = paragraph::alignAndDisambiguateBatch(parameters, sites);
    for (size_t i = 0; i < expected.size(); ++i)
    {
        compareObject("gssw-only/edges", expected[i]["graph"]["read_counts_by_edge"], gssw_only[i]["read_counts_by_edge"]);
        CHECK(gssw_only[i]["alignments"].isArray() && gssw_only[i]["alignments"].size() == reads[i].size());
        for (Json const& a : gssw_only[i]["alignments"].elements())
            CHECK(a["graphMappingStatus"].asString() == "MAPPED" && a.isMember("graphCigar") && a.isMember("bases"));
    }
}

static void testGenotypesSingleSwap(std::string const& dir)
{
    const std::string base = dir + "/chrX/";
    const std::string manifest = "/tmp/pg_workflow_manifest.txt";
    {
        std::ofstream out(manifest);
        out << "#id\tpath\tdepth\tread length\tdepth sd\tsex\n";
        out << "SAMPLE1\t" << base << "chrX_graph_typing.bam\t44.2\t150\t20\tmale\n";
        out << "SAMPLE2\t" << base << "chrX_graph_typing.bam\t44.2\t150\t20\tfemale\n";
    }
    genotyping::Samples samples = genotyping::loadManifest(manifest);
    std::remove(manifest.c_str());
    const std::string graph = base + "chrX_graph_typing.2sample.json", fasta = base + "chrX_graph_typing.fa", gparams = base + "param.json";

    // the reference's shape: one sample at a time, then the genotyper
    grmpy::Parameters parameters;
    for (auto& sample : samples)
    {
        common::BamReader reader(sample.filename(), sample.index_filename(), fasta);
        grmpy::alignSingleSample(parameters, graph, fasta, reader, sample);
        CHECK(sample.get_alignment_data().isMember("read_counts_by_edge") && !sample.get_alignment_data().isMember("alignments"));
        CHECK(sample.get_alignment_data()["bam"].asString() == sample.filename());
    }
    const Json one_by_one = grmpy::countAndGenotype(graph, fasta, gparams, samples);
    CHECK(one_by_one["samples"]["SAMPLE1"]["gt"]["GT"].asString() == "REF");
    CHECK(one_by_one["samples"]["SAMPLE2"]["gt"]["GT"].asString() == "REF/REF");
    CHECK(one_by_one["graphinfo"]["ID"].asString() == "chrX_graph_typing" && one_by_one["breakpointinfo"].isArray());
    CHECK(one_by_one["samples"]["SAMPLE1"]["breakpoints"].size() >= 2);
    CHECK(one_by_one["samples"]["SAMPLE1"]["gt"]["num_reads"].asInt64() > 50);
    CHECK(Json::parse(one_by_one.dump(2)) == one_by_one);
    // two samples -> a "population" block: both call REF, so one observed allele, no exact test, call rate 1
    CHECK(one_by_one["population"]["call_rate"].asDouble() == 1.0 && one_by_one["population"]["hwe_fisher"].asString().empty());
    CHECK(one_by_one["population"]["breakpoints"].size() == one_by_one["samples"]["SAMPLE1"]["breakpoints"].size());

    // the batched shape: both samples (x the same graph listed twice) in one device batch, same documents
    parameters.threads = 4;
    const std::vector<Json> batched = grmpy::genotypeGraphs(parameters, { graph, graph }, fasta, samples, gparams);
    CHECK(batched.size() == 2 && batched[0] == one_by_one && batched[1] == one_by_one);  // packed reads (the default) == object form
    parameters.packed_reads = false;
    parameters.lanes = 2;
    parameters.sites_per_batch = 2;
    const std::vector<Json> from_objects = grmpy::genotypeGraphs(parameters, { graph, graph, graph }, fasta, samples, gparams);
    CHECK(from_objects.size() == 3 && from_objects[0] == one_by_one && from_objects[2] == one_by_one);
    // many small chunks on three lanes: the staggered first round and the shrinking last chunks change nothing
    parameters.packed_reads = true;
    parameters.lanes = 3;
    parameters.threads = 6;
    parameters.sites_per_batch = 4;  // two graphs x two samples per chunk
    const std::vector<Json> many = grmpy::genotypeGraphs(parameters, std::vector<std::string>(25, graph), fasta, samples, gparams);
    CHECK(many.size() == 25);
    for (Json const& document : many)
        CHECK(document == one_by_one);

    // paragraph::countGraphs: one chunk after the other == chunks on lanes
    paragraph::Parameters count_parameters;
    count_parameters.threads = 8;
    const std::string bam = base + "chrX_graph_typing.bam";
    const std::vector<Json> one_chunk = paragraph::countGraphs(count_parameters, std::vector<std::string>(7, graph), fasta, { bam });
    const std::vector<Json> on_lanes = paragraph::countGraphs(count_parameters, std::vector<std::string>(7, graph), fasta, { bam }, {}, "", 2);
    CHECK(one_chunk.size() == 7 && on_lanes.size() == 7);
    for (size_t i = 0; i < on_lanes.size() && i < one_chunk.size(); ++i)
        CHECK(on_lanes[i] == one_chunk[0] && one_chunk[i] == one_chunk[0] && on_lanes[i]["read_counts_by_edge"].size() > 0);
    std::cout << one_by_one["samples"]["SAMPLE2"]["gt"].dump() << "\n";
}

int main(int argc, char** argv)
{
    if (argc < 2)
    {
        std::cerr << "usage: test_workflow <tests/golden/sites>\n";
        return 2;
    }
    try
    {
        testMultiparagraph(argv[1]);
        testGenotypesSingleSwap(argv[1]);
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
    std::cout << "workflow: all checks passed\n";
    return 0;
}
