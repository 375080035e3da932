// Indexed BAM access without htslib (role of common::BamReader, src/c++/include/common/BamReader.hh:63-100): BGZF blocks
// are inflated with zlib and the .bai bin / linear index is walked directly (SAM/BAM specification v1, sections 4 and 5).
// Region iteration yields what sam_itr_querys / sam_itr_next would, in the same order: file order, records whose
// [pos, end) touches the query; getAlign additionally drops secondary (0x100) and supplementary (0x800) records.
// Header and index are parsed once per process and shared between readers of the same file; each reader keeps its own
// file handle and a small ring of inflated blocks, so one reader per thread is the intended use.
// CRAM is not supported; the FASTA argument is only checked for existence, like the original does.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "common/ReadReader.hh"

namespace common
{
// the SAM flag bits this reader looks at
namespace samflag
{
constexpr uint16_t kUnmapped = 0x4, kMateUnmapped = 0x8, kReverse = 0x10, kMateReverse = 0x20, kFirstInPair = 0x40;
constexpr uint16_t kSecondary = 0x100, kSupplementary = 0x800;
}  // namespace samflag

class BamReader final : public ReadReader
{
public:
    // index_path "" = <path>.bai, else <path without .bam>.bai; reference "" skips the FASTA existence checks
    BamReader(const std::string& path, const std::string& index_path, const std::string& reference);
    BamReader(BamReader&&) noexcept;
    BamReader& operator=(BamReader&&) noexcept;
    ~BamReader() override;

    void setRegion(const std::string& region_text) override;
    bool getAlign(Read& record) override;
    bool getAlignLean(LeanAlign& record) override;
    bool getAlignedMate(const Read& read, Read& mate) override;

    // from the BAM header
    std::vector<std::string> const& contigNames() const;
    std::vector<int64_t> const& contigLengths() const;
    std::string const& headerText() const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};
}  // namespace common
