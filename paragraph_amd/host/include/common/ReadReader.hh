// Where the reads of one sample come from, as the extraction code sees it (common::ReadReader,
// src/c++/include/common/ReadReader.hh:28-38): position on a region, hand out its records one by one, look up a mate.
// common::BamReader is the file-backed implementation; tests script their own.
#pragma once
#include <cstdint>
#include <string>

#include "common/Read.hh"

namespace common
{
// One alignment record as the packed extraction (paragraph::extractPacked) needs it: no strings are built.  `name` and the
// base storage point into the reader and stay valid until its next call.
struct LeanAlign
{
    const char* name = nullptr;
    uint32_t name_len = 0;
    uint32_t n_bases = 0;
    const unsigned char* packed_bases = nullptr;  // BAM's 4-bit codes, two per byte (high nibble first), or
    const char* text_bases = nullptr;             // plain characters (readers that only have getAlign)
    int32_t chrom_id = -1, pos = -1, mate_chrom_id = -1, mate_pos = -1;
    bool is_mapped = false, is_first_mate = false, is_mate_mapped = false, is_reverse_strand = false, is_mate_reverse_strand = false;
    void appendBasesTo(std::string& out) const;
};

class ReadReader
{
public:
    // "chr", "chr:begin" or "chr:begin-end" (1-based, inclusive); restarts iteration
    virtual void setRegion(const std::string& region_text) = 0;
    // fills `record` with the next primary alignment overlapping the region; false once there is none left
    virtual bool getAlign(Read& record) = 0;
    // looks where `read` says its mate is (or, for an unmapped mate, where the read itself is) for the other end of the
    // fragment; true when found.  `mate` may be overwritten with other records on the way
    virtual bool getAlignedMate(const Read& read, Read& mate) = 0;
    // the records getAlign hands out, without building a Read (the default goes through getAlign)
    virtual bool getAlignLean(LeanAlign& record);
    virtual ~ReadReader() = default;

private:
    Read lean_scratch_;
};
}  // namespace common
