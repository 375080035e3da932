// Indexed BAM access (common::BamReader, src/c++/include/common/BamReader.hh:63-100) without htslib: BGZF blocks are inflated
// with zlib and the .bai bin / linear index is walked directly (SAM/BAM specification v1, sections 4 and 5).  Region
// iteration yields the same records in the same order as sam_itr_querys / sam_itr_next: file order, records whose
// [pos, end) touches the query, secondary (0x100) and supplementary (0x800) ones dropped by getAlign.
// CRAM is not supported (the reference argument is only checked for existence like the original does).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "common/ReadReader.hh"

namespace common
{
class BamReader : public ReadReader
{
public:
    enum
    {
        kSupplementaryAlign = 0x800,
        kSecondaryAlign = 0x100,
        kIsMapped = 0x0004,
        kIsFirstMate = 0x0040,
        kIsMateMapped = 0x0008
    };
    // index_path "" = <path>.bai, then <path minus .bam>.bai; reference "" skips the FASTA existence checks
    BamReader(const std::string& path, const std::string& index_path, const std::string& reference);
    ~BamReader() override;
    BamReader(BamReader&&) noexcept;
    BamReader& operator=(BamReader&&) noexcept;

    void setRegion(const std::string& region_encoding) override;  // "chr", "chr:beg" or "chr:beg-end", 1-based inclusive
    bool getAlign(Read& align) override;
    bool getAlignedMate(const Read& read, Read& mate) override;

    std::vector<std::string> const& contigNames() const;
    std::vector<int64_t> const& contigLengths() const;
    std::string const& headerText() const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};
}  // namespace common
