// The reads of one site in the flat form pg_batch_upload takes, filled straight from the BAM records: no common::Read
// object, name or quality string is kept per read.  Used by the batched workflow when per-read "alignments" are not part
// of the output (grmpy's default); the object form stays available for callers that want the reads back.
#pragma once
#include <cstdint>
#include <list>
#include <string>
#include <vector>

#include "common/ReadReader.hh"
#include "common/Region.hh"

namespace paragraph
{
struct PackedSite
{
    enum Flag : uint8_t
    {
        REVERSE = 1,       // is_reverse_strand
        FIRST_MATE = 2,
        MAPPED = 4,
        MATE_MAPPED = 8,
        MATE_REVERSE = 16
    };
    std::string bases;               // all reads back to back
    std::vector<uint32_t> base_end;  // end offset of read i in `bases`
    std::vector<uint32_t> fragment;  // dense fragment id (same fragment_id text = same id), numbered by first appearance
    std::vector<uint8_t> flags;
    std::vector<int32_t> chrom_id, pos, mate_chrom_id, mate_pos;
    size_t size() const { return base_end.size(); }
    uint32_t readLength(size_t i) const { return base_end[i] - (i ? base_end[i - 1] : 0); }
    void clear();
};

// common::extractReads (lib/common/ReadExtraction.cpp:38-219) into the packed form: same scan window, same in-region test,
// same max_num_reads cap, same mate recovery, same order (per target region: by fragment id, first mate before second).
void extractPacked(
    common::ReadReader& reader, std::list<common::Region> const& target_regions, int max_num_reads, unsigned longest_alt_insertion,
    PackedSite& site, int avr_fragment_length = 333);
}  // namespace paragraph
