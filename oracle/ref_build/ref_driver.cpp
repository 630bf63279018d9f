// C-ABI driver around the *real* reference operators (compiled from
// /root/reference, never copied).  TEST INFRASTRUCTURE ONLY: used to pin the
// oracle restatement, to generate tests/golden/*.npz, and as the
// "kind": "reference" CPU baseline of bench.py.  Nothing under vinum_amd/
// links or loads this.
//
// Reference classes driven (all under /root/reference/vinum_cpp/src):
//   operators/aggregate/single_numerical_hash_aggregate.{h,cpp}
//   operators/aggregate/multi_numerical_hash_aggregate.{h,cpp}
//   operators/aggregate/one_group_aggregate.{h,cpp}
//   operators/sort/sort.{h,cpp}
// The call sequence mirrors vinum/core/vinum_lib.cpp:54-142 (next/result,
// next/sorted); batches cross as Arrow C Data Interface structs instead of
// pyarrow handles.
#include <arrow/api.h>
#include <arrow/c/bridge.h>
#include <arrow/compute/api.h>
#include <arrow/compute/initialize.h>

#include <memory>
#include <string>
#include <vector>

#include "operators/aggregate/multi_numerical_hash_aggregate.h"
#include "operators/aggregate/one_group_aggregate.h"
#include "operators/aggregate/single_numerical_hash_aggregate.h"
#include "operators/sort/sort.h"

namespace agg = vinum::operators::aggregate;
namespace srt = vinum::operators::sort;

namespace {
thread_local std::string g_err;

struct AggHandle {
    std::unique_ptr<agg::BaseAggregate> op;
};
struct SortHandle {
    std::unique_ptr<srt::Sort> op;
};

agg::AggFuncType func_from_int(int f) {
    // ints follow the pybind enum order of vinum/core/vinum_lib.cpp:25-32
    switch (f) {
        case 0: return agg::AggFuncType::COUNT_STAR;
        case 1: return agg::AggFuncType::COUNT;
        case 2: return agg::AggFuncType::MIN;
        case 3: return agg::AggFuncType::MAX;
        case 4: return agg::AggFuncType::SUM;
        case 5: return agg::AggFuncType::AVG;
    }
    throw std::runtime_error("bad func id");
}

std::vector<std::string> strs(int n, const char** p) {
    std::vector<std::string> v;
    for (int i = 0; i < n; i++) v.emplace_back(p[i]);
    return v;
}
}  // namespace

extern "C" {

const char* ref_last_error() { return g_err.c_str(); }

// kind: 0 = OneGroupAggregate, 1 = SingleNumericalHashAggregate, 2 = MultiNumericalHashAggregate
void* ref_agg_create(int kind, int n_groupby, const char** groupby_cols, int n_aggcols,
                     const char** agg_cols, int n_funcs, const int* func_types,
                     const char** in_cols, const char** out_cols) {
    try {
        std::vector<agg::AggFuncDef> defs;
        for (int i = 0; i < n_funcs; i++) {
            defs.push_back(agg::AggFuncDef{func_from_int(func_types[i]), in_cols[i], out_cols[i]});
        }
        auto h = new AggHandle();
        auto gb = strs(n_groupby, groupby_cols);
        auto ac = strs(n_aggcols, agg_cols);
        if (kind == 0) {
            h->op = std::make_unique<agg::OneGroupAggregate>(defs);
        } else if (kind == 1) {
            h->op = std::make_unique<agg::SingleNumericalHashAggregate>(gb, ac, defs);
        } else {
            h->op = std::make_unique<agg::MultiNumericalHashAggregate>(gb, ac, defs);
        }
        return h;
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}

int ref_agg_next(void* hv, struct ArrowArray* arr, struct ArrowSchema* schema) {
    try {
        auto h = static_cast<AggHandle*>(hv);
        auto res = arrow::ImportRecordBatch(arr, schema);
        if (!res.ok()) { g_err = res.status().ToString(); return 1; }
        h->op->Next(res.ValueOrDie());
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return 1;
    }
}

int ref_agg_result(void* hv, struct ArrowArray* out, struct ArrowSchema* out_schema) {
    try {
        auto h = static_cast<AggHandle*>(hv);
        auto batch = h->op->Result();
        auto st = arrow::ExportRecordBatch(*batch, out, out_schema);
        if (!st.ok()) { g_err = st.ToString(); return 1; }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return 1;
    }
}

void ref_agg_destroy(void* hv) { delete static_cast<AggHandle*>(hv); }

void* ref_sort_create(int n, const char** cols, const int* orders /*0 asc 1 desc*/) {
    try {
        static bool inited = false;
        if (!inited) {
            auto st = arrow::compute::Initialize();
            if (!st.ok()) { g_err = st.ToString(); return nullptr; }
            inited = true;
        }
        std::vector<srt::SortOrder> ord;
        for (int i = 0; i < n; i++) ord.push_back(orders[i] ? srt::SortOrder::DESC : srt::SortOrder::ASC);
        auto h = new SortHandle();
        h->op = std::make_unique<srt::Sort>(strs(n, cols), ord);
        return h;
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}

int ref_sort_next(void* hv, struct ArrowArray* arr, struct ArrowSchema* schema) {
    try {
        auto h = static_cast<SortHandle*>(hv);
        auto res = arrow::ImportRecordBatch(arr, schema);
        if (!res.ok()) { g_err = res.status().ToString(); return 1; }
        h->op->Next(res.ValueOrDie());
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return 1;
    }
}

int ref_sort_sorted(void* hv, struct ArrowArray* out, struct ArrowSchema* out_schema) {
    try {
        auto h = static_cast<SortHandle*>(hv);
        auto batch = h->op->Sorted();
        auto st = arrow::ExportRecordBatch(*batch, out, out_schema);
        if (!st.ok()) { g_err = st.ToString(); return 1; }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return 1;
    }
}

void ref_sort_destroy(void* hv) { delete static_cast<SortHandle*>(hv); }

}  // extern "C"
