// Genomic interval [start, end], 0-based inclusive, written "chr:start+1-end+1" (common::Region, src/c++/include/common/Region.hh:32-80;
// coordinate text per stringutil::formatPos / parsePos, include/common/StringUtil.hh:116-173).
#pragma once
#include <cstdint>
#include <string>

namespace common
{
// "chr1:1,000-2000" -> ("chr1", 999, 1999); parts that are absent leave their output untouched
void parsePos(std::string const& text, std::string& chrom, int64_t& start, int64_t& end);
std::string formatPos(std::string const& chrom, int64_t start = -1, int64_t end = -1);

struct Region
{
    Region() = default;
    Region(std::string chrom_, int64_t start_, int64_t end_) : chrom(std::move(chrom_)), start(start_), end(end_) {}
    explicit Region(std::string const& text) { parsePos(text, chrom, start, end); }
    operator std::string() const { return formatPos(chrom, start, end); }

    // the region grown by `flank` on both sides, clamped at the contig start
    Region getExtendedRegion(int64_t flank) const { return Region(chrom, start > flank ? start - flank : 0, end + flank); }
    Region getLeftFlank(int64_t flank) const { return Region(chrom, (start - 1) > flank ? start - flank - 1 : 0, start - 1); }
    Region getRightFlank(int64_t flank) const { return Region(chrom, end + 1, end + 1 + flank); }
    int64_t length() const { return end + 1 - start; }

    std::string chrom;
    int64_t start = -1, end = -1;
};
}  // namespace common
