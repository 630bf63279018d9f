// Force-included when compiling the reference's C++ operators (where they lie
// under /root/reference) against the Arrow 25 headers that ship with this
// image's pyarrow wheel.  The reference pins Arrow 3.0 (setup.py:33); between
// 3.0 and 25 two namespaces it uses were *renamed* (not removed):
//   arrow::BitUtil            -> arrow::bit_util
//   arrow::util::string_view  -> std::string_view
// These aliases only restore the old spellings.  No function, type or header
// is implemented here; every symbol resolves to the real Arrow 25 library.
#pragma once
#include <cassert>
#include <cstring>
#include <string_view>
#include <arrow/util/bit_util.h>
namespace arrow {
namespace BitUtil = ::arrow::bit_util;
namespace util { using string_view = std::string_view; }
}  // namespace arrow
