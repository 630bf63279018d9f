#include "vnm_common.hpp"
using namespace vnm;
extern "C" {
int vnm_sort_indices(int, const vnm_dcol*, const int*, int64_t, int64_t, int64_t*, void*) { return set_error("vnm_sort_indices: not implemented yet"); }
int vnm_take(const vnm_dcol*, const int64_t*, int64_t, void*, uint8_t*, void*) { return set_error("vnm_take: not implemented yet"); }
}
