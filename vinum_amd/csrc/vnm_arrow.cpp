#include "vnm_common.hpp"
using namespace vnm;
extern "C" {
vnm_agg_op* vnm_agg_op_create(int, int, const char**, int, const char**, int, const int*, const char**, const char**) { set_error("not implemented yet"); return nullptr; }
int vnm_agg_op_next(vnm_agg_op*, struct ArrowArray*, struct ArrowSchema*) { return set_error("not implemented yet"); }
int vnm_agg_op_result(vnm_agg_op*, struct ArrowArray*, struct ArrowSchema*) { return set_error("not implemented yet"); }
void vnm_agg_op_destroy(vnm_agg_op*) {}
vnm_sort_op* vnm_sort_op_create(int, const char**, const int*) { set_error("not implemented yet"); return nullptr; }
int vnm_sort_op_next(vnm_sort_op*, struct ArrowArray*, struct ArrowSchema*) { return set_error("not implemented yet"); }
int vnm_sort_op_sorted(vnm_sort_op*, int64_t, struct ArrowArray*, struct ArrowSchema*) { return set_error("not implemented yet"); }
void vnm_sort_op_destroy(vnm_sort_op*) {}
}
