// Pulling the reads of one site out of a sample (src/c++/include/common/ReadExtraction.hh:33-92, lib/common/ReadExtraction.cpp).
#pragma once
#include <list>
#include <utility>
#include <vector>

#include "common/ReadPairs.hh"
#include "common/ReadReader.hh"
#include "common/Region.hh"

namespace common
{
// every target region in turn; reads are appended to all_reads
void extractReads(
    ReadReader& reader, std::list<Region> const& target_regions, int max_num_reads, unsigned longest_alt_insertion,
    std::vector<p_Read>& all_reads, int avr_fragment_length = 333);
// opens path (index_path may be empty = next to the BAM) and runs the above
void extractReads(
    const std::string& bam_path, const std::string& bam_index_path, const std::string& reference_path,
    std::list<Region> const& target_regions, int max_num_reads, unsigned longest_alt_insertion, std::vector<p_Read>& all_reads,
    int avr_fragment_length = 333);
// one region: scan region +- 3 x avr_fragment_length, keep reads that (or whose mates) touch the region; when reads are at
// most twice the longest inserted sequence, also fetch far-away mates.  Returns <kept from the scan, recovered mates>.
std::pair<int, int> extractReadsFromRegion(
    std::vector<p_Read>& all_reads, int max_num_reads, ReadReader& reader, const Region& region, unsigned longest_alt_insertion,
    int avr_fragment_length);
// returns the mean length of the non-empty reads seen
int extractMappedReadsFromRegion(ReadPairs& read_pairs, int max_num_reads, ReadReader& reader, const Region& region);
bool isReadOrItsMateInRegion(Read& read, const Region& region);
void recoverMissingMates(ReadReader& reader, ReadPairs& read_pairs);
}  // namespace common
