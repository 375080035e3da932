// Source of reads for one sample (common::ReadReader, src/c++/include/common/ReadReader.hh:28-38).
#pragma once
#include <string>

#include "common/Read.hh"

namespace common
{
class ReadReader
{
public:
    virtual ~ReadReader() = default;
    virtual void setRegion(const std::string& region_encoding) = 0;
    virtual bool getAlign(Read& align) = 0;                           // next primary record of the region; false when exhausted
    virtual bool getAlignedMate(const Read& read, Read& mate) = 0;   // looks the mate up where the record says it is
};
}  // namespace common
