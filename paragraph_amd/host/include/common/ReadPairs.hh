// Reads of a region grouped by fragment id, first / second mate slots (common::ReadPair / ReadPairs,
// src/c++/include/common/ReadPair.hh:27-39, ReadPairs.hh:33-62).  Iteration and getReads() go in fragment-id order.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common/Read.hh"

namespace common
{
class ReadPair
{
public:
    const Read& first_mate() const { return first_; }
    const Read& second_mate() const { return second_; }
    int numInitialized() const { return (int)first_.is_initialized() + (int)second_.is_initialized(); }
    void add(const Read& read) { (read.is_first_mate() ? first_ : second_) = read; }
    Read& first_mate() { return first_; }
    Read& second_mate() { return second_; }

private:
    Read first_, second_;
};

class ReadPairs
{
public:
    typedef std::map<std::string, ReadPair>::const_iterator const_iterator;
    const_iterator begin() const { return pairs_.begin(); }
    const_iterator end() const { return pairs_.end(); }
    // a later record for the same mate slot replaces the earlier one without changing num_reads()
    void add(const Read& read);
    const ReadPair& operator[](const std::string& fragment_id) const;  // throws for unknown fragments
    int num_reads() const { return num_reads_; }
    void getReads(std::vector<Read>& reads) const;
    void getReads(std::vector<p_Read>& reads) const;
    // like getReads, but moves the reads out (this container is left empty): no second copy of every read's strings
    void takeReads(std::vector<p_Read>& reads);
    void clear();

private:
    std::map<std::string, ReadPair> pairs_;
    int num_reads_ = 0;
};
}  // namespace common
