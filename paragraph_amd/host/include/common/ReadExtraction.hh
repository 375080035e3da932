// Pulling the reads of one site out of a sample: the policy of common::extractReads and its helpers
// (src/c++/include/common/ReadExtraction.hh:33-92, lib/common/ReadExtraction.cpp:38-219), over any ReadReader.
//
//   scan window   = target region grown by 3 x avr_fragment_length on both sides
//   kept          = reads that touch the region themselves, or whose mate (same contig, assumed as long as the read) does
//   cap           = max_num_reads per region (counting filled mate slots)
//   mate recovery = only when the cap was not hit and the mean read length is <= 2 x longest_alt_insertion: fragments with
//                   one mate whose partner lies >= 1000 bp away or on another contig get that partner looked up
//   order         = per region by fragment id, first mate before second; regions in the order given
#pragma once
#include <list>
#include <string>
#include <utility>

#include "common/ReadPairs.hh"
#include "common/ReadReader.hh"
#include "common/Region.hh"

namespace common
{
using RegionList = std::list<Region>;

// lowest level: is the read, or the place its mate is reported at, inside `region` (>= 1 base)?
bool isReadOrItsMateInRegion(Read& read, const Region& region);
// scans the reader's current region into `pairs` until it is exhausted or `pairs` holds max_num_reads; returns the mean
// length of the non-empty reads that went by
int extractMappedReadsFromRegion(ReadPairs& pairs, int max_num_reads, ReadReader& reader, const Region& region);
void recoverMissingMates(ReadReader& reader, ReadPairs& pairs);
// one target region, appended to `out`; returns <reads kept by the scan, mates recovered afterwards>
std::pair<int, int> extractReadsFromRegion(
    ReadBuffer& out, int max_num_reads, ReadReader& reader, const Region& region, unsigned longest_alt_insertion,
    int avr_fragment_length);
// all target regions of a site
void extractReads(
    ReadReader& reader, RegionList const& target_regions, int max_num_reads, unsigned longest_alt_insertion, ReadBuffer& out,
    int avr_fragment_length = 333);
// the same on a BAM opened for the occasion (bam_index_path "" = next to the BAM)
void extractReads(
    const std::string& bam_path, const std::string& bam_index_path, const std::string& reference_path, RegionList const& target_regions,
    int max_num_reads, unsigned longest_alt_insertion, ReadBuffer& out, int avr_fragment_length = 333);
}  // namespace common
