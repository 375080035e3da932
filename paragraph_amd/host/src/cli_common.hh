// Shared by the two command-line front ends (grmpy_main.cpp, paragraph_main.cpp): response files, boost-style bool
// values, option walking, plain / gzip output.
#pragma once
#include <zlib.h>

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace cli
{
// --devices 0,1,2,3 | all  ("all" is resolved by the host library through PG_DEVICES: an empty list means "its default")
inline std::vector<int> deviceList(std::string const& value)
{
    std::vector<int> out;
    if (value == "all")
    {
        setenv("PG_DEVICES", "all", 1);
        return out;
    }
    std::stringstream ss(value);
    std::string item;
    while (std::getline(ss, item, ','))
        if (!item.empty())
        {
            size_t used = 0;
            int ordinal = -1;
            try
            {
                ordinal = std::stoi(item, &used);
            }
            catch (std::exception const&)
            {
                used = 0;
            }
            if (used != item.size() || ordinal < 0)
                throw std::runtime_error("--devices: '" + item + "' is not a device ordinal (expected e.g. 0,1,2,3 or 'all')");
            out.push_back(ordinal);
        }
    if (out.empty())
        throw std::runtime_error("--devices expects a comma-separated list of device ordinals or 'all'");
    return out;
}

inline std::vector<std::string> splitShell(std::string const& text)
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

inline bool toBool(std::string v, std::string const& option)
{
    for (auto& c : v)
        c = (char)tolower((unsigned char)c);
    if (v == "1" || v == "true" || v == "yes" || v == "on")
        return true;
    if (v == "0" || v == "false" || v == "no" || v == "off")
        return false;
    throw std::runtime_error("the argument ('" + v + "') for option '" + option + "' is invalid");
}

inline void writeOutput(std::string const& path, std::string const& text, bool gzip)
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

// argv with every --response-file[=]FILE replaced by the words of that file
inline std::vector<std::string> expandArguments(int argc, char** argv)
{
    std::vector<std::string> args(argv + 1, argv + argc);
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
    return args;
}

// Walks "--name value", "--name=value", "-n value" and multi-value options the way boost::program_options reads them.
class Arguments
{
public:
    explicit Arguments(std::vector<std::string> words) : words_(std::move(words)) {}
    bool next()
    {
        if (++at_ >= (std::ptrdiff_t)words_.size())
            return false;
        name_ = words_[(size_t)at_];
        has_value_ = false;
        if (name_.compare(0, 2, "--") == 0)
        {
            const size_t eq = name_.find('=');
            if (eq != std::string::npos)
            {
                value_ = name_.substr(eq + 1);
                name_ = name_.substr(0, eq);
                has_value_ = true;
            }
        }
        return true;
    }
    std::string const& name() const { return name_; }
    bool is(const char* short_name, const char* long_name) const { return (short_name && name_ == short_name) || name_ == long_name; }
    std::string value()
    {
        if (has_value_)
            return value_;
        if (at_ + 1 >= (std::ptrdiff_t)words_.size())
            throw std::runtime_error("the required argument for option '" + name_ + "' is missing");
        return words_[(size_t)++at_];
    }
    bool boolValue() { return toBool(value(), name_); }
    // implicit_value(true): a following word is taken only when it reads as a bool
    bool optionalBool()
    {
        if (has_value_)
            return toBool(value_, name_);
        if (at_ + 1 < (std::ptrdiff_t)words_.size() && !words_[(size_t)at_ + 1].empty() && words_[(size_t)at_ + 1][0] != '-')
        {
            try
            {
                const bool b = toBool(words_[(size_t)at_ + 1], name_);
                ++at_;
                return b;
            }
            catch (std::exception const&)
            {
            }
        }
        return true;
    }
    // multitoken(): every following word up to the next option
    void values(std::vector<std::string>& out)
    {
        if (has_value_)
            out.push_back(value_);
        while (at_ + 1 < (std::ptrdiff_t)words_.size() && !(words_[(size_t)at_ + 1].size() > 1 && words_[(size_t)at_ + 1][0] == '-'))
            out.push_back(words_[(size_t)++at_]);
    }

private:
    std::vector<std::string> words_;
    std::ptrdiff_t at_ = -1;
    std::string name_, value_;
    bool has_value_ = false;
};

inline std::string baseName(std::string const& path)
{
    const size_t slash = path.rfind('/');
    return slash == std::string::npos ? path : path.substr(slash + 1);
}
}  // namespace cli
