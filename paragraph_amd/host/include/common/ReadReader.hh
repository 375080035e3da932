// Where the reads of one sample come from, as the extraction code sees it (common::ReadReader,
// src/c++/include/common/ReadReader.hh:28-38): position on a region, hand out its records one by one, look up a mate.
// common::BamReader is the file-backed implementation; tests script their own.
#pragma once
#include <string>

#include "common/Read.hh"

namespace common
{
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
    virtual ~ReadReader() = default;
};
}  // namespace common
