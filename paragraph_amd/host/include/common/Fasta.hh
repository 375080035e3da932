// Indexed FASTA access for graph nodes given as reference intervals (common::FastaFile, src/c++/include/common/Fasta.hh:40-70).
// Uses the samtools .fai next to the file (scans the FASTA when there is none); sequence comes back upper-cased with everything but ACGT turned into N
// (lib/common/Fasta.cpp:435-462).  Queries are positional reads: one object may be shared by many threads.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace common
{
class FastaFile
{
public:
    explicit FastaFile(std::string const& path);
    ~FastaFile();
    FastaFile(FastaFile const&) = delete;
    FastaFile& operator=(FastaFile const&) = delete;
    std::string const& getFilename() const;
    std::string query(std::string const& location) const;  // "chr:start-end", 1-based inclusive
    std::string query(std::string const& chrom, int64_t start, int64_t end) const;  // 0-based inclusive; clipped to the contig
    size_t contigSize(std::string const& contig) const;
    std::vector<std::string> getContigNames() const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};
}  // namespace common
