// `paragraph` with the reference's command line (src/c++/main/paragraph.cpp:83-290) over the batched MI355X workflow: every
// graph against the given BAM(s), one count document per graph (an array when several graphs are given) to -o / stdout, or
// one file per graph under -O; multiparagraph.py can call it in place of the original.
// Not carried over (outputs this build does not compute): --output-variants, --output-path-coverage,
// --output-node-coverage, --output-read-haplotypes and therefore --output-everything; --validate-alignments, --variant-min-*
// and --log-* are accepted and ignored.
#include <algorithm>

#include "cli_common.hh"
#include "paragraph/SiteBatcher.hh"
#include "paragraph/Workflow.hh"

namespace
{
const char* kUsage = "paragraph -r <reference> -g <graph(s)> -b <input bam(s)> [optional arguments]\n"
                     "  -b, --bam FILE...                 input BAM file(s); several are pooled per graph\n"
                     "      --bam-index FILE...           their indexes (default: next to each BAM)\n"
                     "  -g, --graph-spec FILE...          JSON file(s) describing the graph(s)\n"
                     "  -r, --reference FILE              reference genome FASTA\n"
                     "  -o, --output-file FILE            output file; stdout if '-' or neither -o nor -O is given\n"
                     "  -O, --output-folder DIR           one output file per graph, named like the graph file\n"
                     "  -z, --gzip-output [BOOL]\n"
                     "  -T, --target-regions LIST         chr1:1-20,chr2:2-40 -- overrides the graphs' target regions\n"
                     "  -M, --max-reads-per-event N       (10000)\n"
                     "      --bad-align-frac F (0.8)   --bad-align-nonuniq BOOL (true)   --bad-align-uniq-kmer-len N (0)\n"
                     "      --path-sequence-matching BOOL (true)    --graph-sequence-matching BOOL (true)\n"
                     "      --klib-sequence-matching BOOL (false)   --kmer-sequence-matching BOOL (false)\n"
                     "      --output-detailed-read-counts [BOOL]    -a, --output-alignments [BOOL]\n"
                     "  -A, --output-filtered-alignments [BOOL]     (filter tallies; filtered reads are not re-emitted)\n"
                     "      --threads N                   host threads (the CPUs this process may use)\n"
                     "      --devices LIST                GPUs to spread the site batches over: 0,1,2,3 or 'all' (default: PG_DEVICES, else 0)\n"
                     "      --response-file FILE\n";
}

int main(int argc, char** argv)
{
    try
    {
        cli::Arguments args(cli::expandArguments(argc, argv));
        paragraph::Parameters parameters;
        parameters.threads = paragraph::usableCpus();
        std::string reference, output_file, output_folder, target_regions;
        std::vector<std::string> graphs, bams, bam_indexes;
        bool gzip = false;
        auto output_flag = [&](paragraph::Parameters::output_options bit, bool on) {
            if (on)
                parameters.output_options_ |= bit;
            else
                parameters.output_options_ &= ~bit;
        };
        while (args.next())
        {
            if (args.is("-h", "--help"))
            {
                std::cout << kUsage;
                return 0;
            }
            else if (args.is("-b", "--bam"))
                args.values(bams);
            else if (args.is(nullptr, "--bam-index"))
                args.values(bam_indexes);
            else if (args.is("-g", "--graph-spec"))
                args.values(graphs);
            else if (args.is("-r", "--reference"))
                reference = args.value();
            else if (args.is("-o", "--output-file"))
                output_file = args.value();
            else if (args.is("-O", "--output-folder"))
                output_folder = args.value();
            else if (args.is("-T", "--target-regions"))
                target_regions = args.value();
            else if (args.is("-z", "--gzip-output"))
                gzip = args.optionalBool();
            else if (args.is("-M", "--max-reads-per-event"))
                parameters.max_reads = std::stoi(args.value());
            else if (args.is(nullptr, "--bad-align-frac"))
                parameters.bad_align_frac = std::stof(args.value());
            else if (args.is(nullptr, "--bad-align-nonuniq"))
                parameters.remove_nonuniq_reads = args.boolValue();
            else if (args.is(nullptr, "--bad-align-uniq-kmer-len"))
                parameters.kmer_len = std::stoi(args.value());
            else if (args.is(nullptr, "--path-sequence-matching"))
                parameters.path_sequence_matching = args.boolValue();
            else if (args.is(nullptr, "--graph-sequence-matching"))
                parameters.graph_sequence_matching = args.boolValue();
            else if (args.is(nullptr, "--klib-sequence-matching"))
                parameters.klib_sequence_matching = args.boolValue();
            else if (args.is(nullptr, "--kmer-sequence-matching"))
                parameters.kmer_sequence_matching = args.boolValue();
            else if (args.is(nullptr, "--output-detailed-read-counts"))
                output_flag(paragraph::Parameters::DETAILED_READ_COUNTS, args.optionalBool());
            else if (args.is("-a", "--output-alignments"))
                output_flag(paragraph::Parameters::ALIGNMENTS, args.optionalBool());
            else if (args.is("-A", "--output-filtered-alignments"))
                output_flag(paragraph::Parameters::FILTERED_ALIGNMENTS, args.optionalBool());
            else if (args.is(nullptr, "--threads"))
                parameters.threads = std::max(1, std::stoi(args.value()));
            else if (args.is(nullptr, "--devices"))
                paragraph::setDevices(cli::deviceList(args.value()));
            else if (args.is(nullptr, "--validate-alignments"))
            {
                // simulated-read bookkeeping (grm::ValidationAligner, lib/grm/ValidationAligner.cpp:59-125): read objects instead
                // of packed reads, the [VALIDATION] lines of logAlignerStats (lib/grm/Align.cpp:42-55) on stderr at the end
                parameters.validate_alignments = args.optionalBool();
            }
            else if (args.is(nullptr, "--progress"))
                (void)args.optionalBool();
            else if (args.is(nullptr, "--variant-min-reads") || args.is(nullptr, "--variant-min-frac") || args.is(nullptr, "--log-level")
                     || args.is(nullptr, "--log-file") || args.is(nullptr, "--log-async"))
                (void)args.value();
            else if (args.is("-v", "--output-variants") || args.is(nullptr, "--output-path-coverage") || args.is(nullptr, "--output-node-coverage")
                     || args.is(nullptr, "--output-read-haplotypes") || args.is("-E", "--output-everything"))
            {
                if (args.optionalBool())
                    throw std::runtime_error("option '" + args.name() + "' is not available in this build (variants / coverage / haplotypes are not computed)");
            }
            else
                throw std::runtime_error("unrecognised option '" + args.name() + "'");
        }
        if (bams.empty())
            throw std::runtime_error("ERROR: BAM file is missing.");
        if (graphs.empty())
            throw std::runtime_error("ERROR: File with variant specification is missing.");
        if (reference.empty())
            throw std::runtime_error("ERROR: Reference genome is missing.");

        const std::vector<common::Json> documents = paragraph::countGraphs(parameters, graphs, reference, bams, bam_indexes, target_regions);
        if (parameters.validate_alignments)
            for (std::string const& line : paragraph::validationLogLines())
                std::cerr << line << "\n";

        if (!output_folder.empty())
            for (size_t g = 0; g < graphs.size(); ++g)
                cli::writeOutput(output_folder + "/" + cli::baseName(graphs[g]) + (gzip ? ".gz" : ""), documents[g].dump(1) + "\n", gzip);
        if (!output_file.empty() || output_folder.empty())
        {
            std::string text;
            if (graphs.size() > 1)
                text += "[";
            for (size_t g = 0; g < documents.size(); ++g)
                text += (g ? "," : "") + documents[g].dump(1);
            text += graphs.size() > 1 ? "]\n" : "\n";
            cli::writeOutput(output_file, text, gzip);
        }
        return 0;
    }
    catch (std::exception const& e)
    {
        std::cerr << e.what() << "\n";
        return 1;
    }
}
