// Read filter callback: returns true when the alignment should be rejected (src/c++/include/grm/Filter.hh:36).
#pragma once
#include <functional>

#include "common/Read.hh"

namespace grm
{
typedef std::function<bool(common::Read&)> ReadFilter;
}
