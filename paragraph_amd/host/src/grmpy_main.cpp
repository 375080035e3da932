// `grmpy` with the reference's command line (src/c++/main/grmpy.cpp:60-200) over the batched MI355X workflow: existing
// drivers -- e.g. src/python/bin/multigrmpy.py --grmpy <this binary> -- keep working unchanged.  All it does is parse the
// options and call grmpy::genotypeGraphs (paragraph/Workflow.hh); output as the original writes it: one document, or a
// JSON array when several graphs are given, to -o (gzip with -z) or one file per graph under -O.
// -A / --alignment-output-folder writes every (sample, graph) pair's count document with its per-read records into the folder
// (lib/grmpy/AlignSamples.cpp:57-109).  Not carried over: --infer-read-haplotypes (phasing is outside this build); --log-* and
// --progress are accepted and ignored.
#include <algorithm>

#include "cli_common.hh"
#include "paragraph/SiteBatcher.hh"
#include "paragraph/Workflow.hh"

namespace
{
const char* kUsage = "grmpy -r <reference> -g <graphs> -m <manifest> [optional arguments]\n"
                     "  -r, --reference FILE              reference genome FASTA\n"
                     "  -g, --graph-spec FILE...          JSON file(s) describing the graph(s)\n"
                     "  -m, --manifest FILE               manifest of samples (id, path, depth, read length, ...)\n"
                     "  -G, --genotyping-parameters FILE  JSON file with genotyping model parameters\n"
                     "  -o, --output-file FILE            output file; stdout if omitted or '-'\n"
                     "  -O, --output-folder DIR           one output file per graph, named like the graph file\n"
                     "  -z, --gzip-output [BOOL]          gzip-compress the output\n"
                     "  -A, --alignment-output-folder DIR per (sample, graph): count document + per-read alignments, <DIR>/<sample>-<graph>-<regions>.json.gz\n"
                     "  -M, --max-reads-per-event N       (10000)\n"
                     "      --bad-align-frac F            (0.8)\n"
                     "      --path-sequence-matching BOOL (false)   --graph-sequence-matching BOOL (true)\n"
                     "      --klib-sequence-matching BOOL (false)   --kmer-sequence-matching BOOL (false)\n"
                     "      --bad-align-uniq-kmer-len N   (0)\n"
                     "      --exact-match-shortcut [BOOL] (not in the original) reads with one exact full-length match skip their fills; same output\n"
                     "  -t, --sample-threads N            host threads (the CPUs this process may use)\n"
                     "      --devices LIST                GPUs to spread the site batches over: 0,1,2,3 or 'all' (default: PG_DEVICES, else 0)\n"
                     "      --response-file FILE          read further options from FILE\n";
}

int main(int argc, char** argv)
{
    try
    {
        cli::Arguments args(cli::expandArguments(argc, argv));
        grmpy::Parameters parameters;
        parameters.threads = paragraph::usableCpus();
        std::string reference, manifest, genotyping_parameters, output_file, output_folder;
        std::vector<std::string> graphs;
        bool gzip = false;
        while (args.next())
        {
            if (args.is("-h", "--help"))
            {
                std::cout << kUsage;
                return 0;
            }
            else if (args.is("-r", "--reference"))
                reference = args.value();
            else if (args.is("-m", "--manifest"))
                manifest = args.value();
            else if (args.is("-G", "--genotyping-parameters"))
                genotyping_parameters = args.value();
            else if (args.is("-o", "--output-file"))
                output_file = args.value();
            else if (args.is("-O", "--output-folder"))
                output_folder = args.value();
            else if (args.is("-g", "--graph-spec"))
                args.values(graphs);
            else if (args.is("-M", "--max-reads-per-event"))
                parameters.max_reads = std::stoi(args.value());
            else if (args.is(nullptr, "--bad-align-frac"))
                parameters.bad_align_frac = std::stof(args.value());
            else if (args.is(nullptr, "--path-sequence-matching"))
                parameters.path_sequence_matching = args.boolValue();
            else if (args.is(nullptr, "--graph-sequence-matching"))
                parameters.graph_sequence_matching = args.boolValue();
            else if (args.is(nullptr, "--klib-sequence-matching"))
                parameters.klib_sequence_matching = args.boolValue();
            else if (args.is(nullptr, "--kmer-sequence-matching"))
                parameters.kmer_sequence_matching = args.boolValue();
            else if (args.is(nullptr, "--exact-match-shortcut"))
                parameters.exact_match_shortcut = args.optionalBool();
            else if (args.is(nullptr, "--bad-align-uniq-kmer-len"))
                parameters.bad_align_uniq_kmer_len = std::stoi(args.value());
            else if (args.is("-t", "--sample-threads"))
                parameters.threads = std::max(1, std::stoi(args.value()));
            else if (args.is(nullptr, "--devices"))
                parameters.devices = cli::deviceList(args.value());
            else if (args.is("-z", "--gzip-output"))
                gzip = args.optionalBool();
            else if (args.is(nullptr, "--progress"))
                (void)args.optionalBool();
            else if (args.is(nullptr, "--log-level") || args.is(nullptr, "--log-file") || args.is(nullptr, "--log-async"))
                (void)args.value();
            else if (args.is("-A", "--alignment-output-folder"))
                parameters.alignment_output_folder = args.value();
            else if (args.is(nullptr, "--infer-read-haplotypes"))
                throw std::runtime_error("option '" + args.name() + "' is not available in this build (no phasing output)");
            else
                throw std::runtime_error("unrecognised option '" + args.name() + "'");
        }
        if (reference.empty())
            throw std::runtime_error("Reference genome is missing.");
        if (graphs.empty())
            throw std::runtime_error("Graph spec is missing.");
        if (manifest.empty())
            throw std::runtime_error("Manifest file is missing.");

        const genotyping::Samples samples = genotyping::loadManifest(manifest);
        std::vector<std::string> documents;  // serialised by the lanes that made them (Parameters::genotype_text)
        parameters.genotype_text = &documents;
        parameters.genotype_text_indent = 1;
        grmpy::genotypeGraphs(parameters, graphs, reference, samples, genotyping_parameters);

        for (std::string& d : documents)
            if (d.empty())
                d = "null";  // (a graph nothing was written for: the empty document's text, never a hole between the commas)
        if (!output_folder.empty())
            for (size_t g = 0; g < graphs.size(); ++g)
                cli::writeOutput(output_folder + "/" + cli::baseName(graphs[g]) + (gzip ? ".gz" : ""), documents[g] + "\n", gzip);
        if (!output_file.empty() || output_folder.empty())
        {
            std::string text;
            if (graphs.size() > 1)
                text += "[";
            for (size_t g = 0; g < documents.size(); ++g)
                text += (g ? "," : "") + documents[g];
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
