// One line of a grmpy manifest (genotyping::SampleInfo / loadManifest, src/c++/include/genotyping/SampleInfo.hh:40-110,
// lib/genotyping/SampleInfo.cpp:62-260).
#pragma once
#include <string>
#include <vector>

#include "common/Json.hh"
#include "genotyping/GraphBreakpointGenotyper.hh"

namespace genotyping
{
class SampleInfo
{
public:
    std::string const& sample_name() const { return sample_name_; }
    void set_sample_name(std::string const& v) { sample_name_ = v; }
    std::string const& filename() const { return filename_; }
    void set_filename(std::string const& v) { filename_ = v; }
    std::string const& index_filename() const { return index_filename_; }
    void set_index_filename(std::string const& v) { index_filename_ = v; }
    double autosome_depth() const { return autosome_depth_; }
    // without an explicit sd the depth sd defaults to sqrt(5 x depth)
    void set_autosome_depth(double v);
    double depth_sd() const { return depth_sd_; }
    void set_depth_sd(double v) { depth_sd_ = v; }
    unsigned read_length() const { return read_length_; }
    void set_read_length(unsigned v) { read_length_ = v; }
    Sex sex() const { return sex_; }
    void set_sex(std::string sex_string);  // m... / f... / u..., any case; anything else throws
    // per-graph count documents ("paragraph" column, or filled in by the workflow)
    common::Json const& get_alignment_data() const { return alignment_data_; }
    void set_alignment_data(common::Json v) { alignment_data_ = std::move(v); }

private:
    std::string sample_name_, filename_, index_filename_;
    double autosome_depth_ = 0, depth_sd_ = 0;
    unsigned read_length_ = 0;
    Sex sex_ = Sex::UNKNOWN;
    common::Json alignment_data_;
};
typedef std::vector<SampleInfo> Samples;

// Tab- or comma-separated table; '#' anywhere is dropped; columns (any case): id, path, index_path, paragraph, idxdepth,
// depth, read length, sex, depth variance, depth sd.  id + path and either idxdepth or depth + read length are required.
// Relative files are looked up next to the manifest when they are not found as given.
Samples loadManifest(std::string const& filename);
}  // namespace genotyping
