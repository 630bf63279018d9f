#include "vnm_common.hpp"
using namespace vnm;
extern "C" {
int vnm_project(int, const vnm_expr_ins*, int, const vnm_dcol*, int64_t, void*, int*, void*) { return set_error("vnm_project: not implemented yet"); }
}
