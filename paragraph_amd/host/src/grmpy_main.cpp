// `grmpy` with the reference's command line (src/c++/main/grmpy.cpp:60-200) over the batched MI355X workflow: existing
// drivers -- e.g. src/python/bin/multigrmpy.py --grmpy <this binary> -- keep working unchanged.  All it does is parse the
// options and call grmpy::genotypeGraphs (paragraph/Workflow.hh); output as the original writes it: one document, or a
// JSON array when several graphs are given, to -o (gzip with -z) or one file per graph under -O.
// Not carried over: --alignment-output-folder / --infer-read-haplotypes (per-read outputs the batched workflow does not
// produce); --log-* and --progress are accepted and ignored.
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

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
                     "  -M, --max-reads-per-event N       (10000)\n"
                     "      --bad-align-frac F            (0.8)\n"
                     "      --path-sequence-matching BOOL (false)   --graph-sequence-matching BOOL (true)\n"
                     "      --klib-sequence-matching BOOL (false)   --kmer-sequence-matching BOOL (false)\n"
                     "      --bad-align-uniq-kmer-len N   (0)\n"
                     "  -t, --sample-threads N            host threads (1)\n"
                     "      --response-file FILE          read further options from FILE\n";

std::vector<std::string> splitShell(std::string const& text)
{
    std::vector<std::string> out;
    std::string cur;
    bool in_token = false;
    char quote = 0;
    for (size_t i = 0; i < text.size(); ++i)
    {
        const char c = text[i];
        if (quote)
        {
            if (c == quote)
                quote = 0;
            else if (c == '\\' && quote == '"' && i + 1 < text.size())
                cur += text[++i];
            else
                cur += c;
        }
        else if (c == '\'' || c == '"')
        {
            quote = c;
            in_token = true;
        }
        else if (c == '\\' && i + 1 < text.size())
        {
            cur += text[++i];
            in_token = true;
        }
        else if (isspace((unsigned char)c))
        {
            if (in_token)
                out.push_back(cur);
            cur.clear();
            in_token = false;
        }
        else
        {
            cur += c;
            in_token = true;
        }
    }
    if (quote)
        throw std::runtime_error("unterminated quote in response file");
    if (in_token)
        out.push_back(cur);
    return out;
}

bool toBool(std::string v, std::string const& option)
{
    for (auto& c : v)
        c = (char)tolower((unsigned char)c);
    if (v == "1" || v == "true" || v == "yes" || v == "on")
        return true;
    if (v == "0" || v == "false" || v == "no" || v == "off")
        return false;
    throw std::runtime_error("the argument ('" + v + "') for option '" + option + "' is invalid");
}

void writeOutput(std::string const& path, std::string const& text, bool gzip)
{
    if (path.empty() || path == "-")
    {
        std::cout << text;
        return;
    }
    if (gzip)
    {
        gzFile f = gzopen(path.c_str(), "wb");
        if (!f)
            throw std::runtime_error("ERROR: Failed to open output file '" + path + "'");
        const int n = gzwrite(f, text.data(), (unsigned)text.size());
        const int rc = gzclose(f);
        if (n != (int)text.size() || rc != Z_OK)
            throw std::runtime_error("ERROR: Failed to write output file '" + path + "'");
        return;
    }
    std::ofstream f(path, std::ios::binary);
    if (!f.good())
        throw std::runtime_error("ERROR: Failed to open output file '" + path + "'");
    f << text;
}
}  // namespace

int main(int argc, char** argv)
{
    try
    {
        std::vector<std::string> args(argv + 1, argv + argc);
        // --response-file[=]FILE splices the file's words in place
        for (size_t i = 0; i < args.size();)
        {
            std::string file;
            size_t used = 0;
            if (args[i].compare(0, 16, "--response-file=") == 0)
            {
                file = args[i].substr(16);
                used = 1;
            }
            else if (args[i] == "--response-file" && i + 1 < args.size())
            {
                file = args[i + 1];
                used = 2;
            }
            if (!used)
            {
                ++i;
                continue;
            }
            std::ifstream in(file);
            if (!in.good())
                throw std::runtime_error("cannot open response file " + file);
            std::stringstream ss;
            ss << in.rdbuf();
            const auto words = splitShell(ss.str());
            args.erase(args.begin() + (std::ptrdiff_t)i, args.begin() + (std::ptrdiff_t)(i + used));
            args.insert(args.begin() + (std::ptrdiff_t)i, words.begin(), words.end());
        }

        grmpy::Parameters parameters;
        std::string reference, manifest, genotyping_parameters, output_file, output_folder;
        std::vector<std::string> graphs;
        bool gzip = false;
        for (size_t i = 0; i < args.size(); ++i)
        {
            std::string name = args[i], value;
            bool has_value = false;
            if (name.compare(0, 2, "--") == 0)
            {
                const size_t eq = name.find('=');
                if (eq != std::string::npos)
                {
                    value = name.substr(eq + 1);
                    name = name.substr(0, eq);
                    has_value = true;
                }
            }
            auto next = [&]() -> std::string {
                if (has_value)
                    return value;
                if (i + 1 >= args.size())
                    throw std::runtime_error("the required argument for option '" + name + "' is missing");
                return args[++i];
            };
            auto optionalBool = [&]() {  // implicit_value(true): a following word is taken only if it reads as a bool
                if (has_value)
                    return toBool(value, name);
                if (i + 1 < args.size() && !args[i + 1].empty() && args[i + 1][0] != '-')
                {
                    try
                    {
                        const bool b = toBool(args[i + 1], name);
                        ++i;
                        return b;
                    }
                    catch (std::exception const&)
                    {
                    }
                }
                return true;
            };
            if (name == "-h" || name == "--help")
            {
                std::cout << kUsage;
                return 0;
            }
            else if (name == "-r" || name == "--reference")
                reference = next();
            else if (name == "-m" || name == "--manifest")
                manifest = next();
            else if (name == "-G" || name == "--genotyping-parameters")
                genotyping_parameters = next();
            else if (name == "-o" || name == "--output-file")
                output_file = next();
            else if (name == "-O" || name == "--output-folder")
                output_folder = next();
            else if (name == "-g" || name == "--graph-spec")
            {
                if (has_value)
                    graphs.push_back(value);
                while (i + 1 < args.size() && !(args[i + 1].size() > 1 && args[i + 1][0] == '-'))
                    graphs.push_back(args[++i]);
            }
            else if (name == "-M" || name == "--max-reads-per-event")
                parameters.max_reads = std::stoi(next());
            else if (name == "--bad-align-frac")
                parameters.bad_align_frac = std::stof(next());
            else if (name == "--path-sequence-matching")
                parameters.path_sequence_matching = toBool(next(), name);
            else if (name == "--graph-sequence-matching")
                parameters.graph_sequence_matching = toBool(next(), name);
            else if (name == "--klib-sequence-matching")
                parameters.klib_sequence_matching = toBool(next(), name);
            else if (name == "--kmer-sequence-matching")
                parameters.kmer_sequence_matching = toBool(next(), name);
            else if (name == "--bad-align-uniq-kmer-len")
                parameters.bad_align_uniq_kmer_len = std::stoi(next());
            else if (name == "-t" || name == "--sample-threads")
                parameters.threads = std::max(1, std::stoi(next()));
            else if (name == "-z" || name == "--gzip-output")
                gzip = optionalBool();
            else if (name == "--progress")
                (void)optionalBool();
            else if (name == "--log-level" || name == "--log-file" || name == "--log-async")
                (void)next();
            else if (name == "-A" || name == "--alignment-output-folder" || name == "--infer-read-haplotypes")
                throw std::runtime_error("option '" + name + "' is not available in this build (no per-read outputs)");
            else
                throw std::runtime_error("unrecognised option '" + name + "'");
        }
        if (reference.empty())
            throw std::runtime_error("Reference genome is missing.");
        if (graphs.empty())
            throw std::runtime_error("Graph spec is missing.");
        if (manifest.empty())
            throw std::runtime_error("Manifest file is missing.");

        const genotyping::Samples samples = genotyping::loadManifest(manifest);
        const std::vector<common::Json> genotypes = grmpy::genotypeGraphs(parameters, graphs, reference, samples, genotyping_parameters);

        if (!output_folder.empty())
        {
            for (size_t g = 0; g < graphs.size(); ++g)
            {
                const size_t slash = graphs[g].rfind('/');
                const std::string base = slash == std::string::npos ? graphs[g] : graphs[g].substr(slash + 1);
                writeOutput(output_folder + "/" + base + (gzip ? ".gz" : ""), genotypes[g].dump(1) + "\n", gzip);
            }
        }
        if (!output_file.empty() || output_folder.empty())
        {
            std::string text;
            if (graphs.size() > 1)
                text += "[";
            for (size_t g = 0; g < genotypes.size(); ++g)
                text += (g ? "," : "") + genotypes[g].dump(1);
            text += graphs.size() > 1 ? "]\n" : "\n";
            writeOutput(output_file, text, gzip);
        }
        return 0;
    }
    catch (std::exception const& e)
    {
        std::cerr << e.what() << "\n";
        return 1;
    }
}
