// The callback a cascade stage asks whether to reject the alignment it just produced: true = reject, the read goes on to
// the next stage (role of grm::ReadFilter, src/c++/include/grm/Filter.hh:36).
#pragma once
#include <functional>

#include "common/Read.hh"

namespace grm
{
using ReadFilter = std::function<bool(common::Read& aligned_read)>;
}
