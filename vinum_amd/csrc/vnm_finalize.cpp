// Host side of the aggregate: lowering of (function, input type) onto accumulator words, and the
// finalisation of result columns from those words.  Pure host code (unit-tested without a GPU).
#include <cstdlib>

#include "vnm_agg.hpp"

namespace vnm {

typedef __int128 i128;
typedef unsigned __int128 u128;

int build_plan(int kind, int n_keys, const int* key_types, int n_funcs, const int* funcs, const int* in_types,
               const int* in_flags, const int* func_col_id, AggPlan* plan, FuncOut* outs) {
    if (n_keys < 0 || n_keys > AGG_MAX_KEYS) return set_error("aggregate: at most %d group-by columns", AGG_MAX_KEYS);
    if (n_funcs < 0 || n_funcs > AGG_MAX_FUNCS) return set_error("aggregate: at most %d aggregate functions", AGG_MAX_FUNCS);
    if (kind == VNM_ONE_GROUP && n_keys != 0) return set_error("OneGroupAggregate takes no group-by columns");
    if (kind == VNM_SINGLE_NUMERICAL && n_keys != 1) return set_error("SingleNumericalHashAggregate takes exactly one group-by column");
    if (kind == VNM_MULTI_NUMERICAL && n_keys < 1) return set_error("MultiNumericalHashAggregate needs group-by columns");
    memset(plan, 0, sizeof(*plan));
    plan->kind = kind;
    plan->n_keys = n_keys;
    plan->kw = n_keys ? n_keys + 1 : 0;
    for (int i = 0; i < n_keys; i++) {
        if (key_types[i] < VNM_I8 || key_types[i] > VNM_F64) return set_error("Unsupported data type for aggregation column.");
        plan->key_types[i] = key_types[i];
    }
    // distinct input columns
    int col_of[AGG_MAX_FUNCS];
    int ncols = 0;
    int col_ids[AGG_MAX_FUNCS];
    for (int i = 0; i < n_funcs; i++) {
        col_of[i] = -1;
        if (funcs[i] == VNM_COUNT_STAR) continue;
        int id = func_col_id ? func_col_id[i] : (1000 + i);
        if (id < 0) id = -1000 - i;   // a negative id means "no sharing": this function's column is its own
        for (int c = 0; c < ncols; c++)
            if (col_ids[c] == id) col_of[i] = c;
        if (col_of[i] < 0) {
            if (ncols >= AGG_MAX_COLS) return set_error("aggregate: at most %d distinct input columns", AGG_MAX_COLS);
            col_ids[ncols] = id;
            col_of[i] = ncols++;
        }
    }
    plan->n_cols = ncols;
    int w_rows = -1;
    int w_valid[AGG_MAX_COLS], w_sum[AGG_MAX_COLS], w_hi[AGG_MAX_COLS], w_min[AGG_MAX_COLS], w_max[AGG_MAX_COLS];
    for (int c = 0; c < AGG_MAX_COLS; c++) w_valid[c] = w_sum[c] = w_hi[c] = w_min[c] = w_max[c] = -1;
    int nw = 0, nops = 0;
    auto new_word = [&](int mk) -> int {
        if (nw >= AGG_MAX_WORDS) return -1;
        plan->merge[nw] = mk;
        return nw++;
    };
    auto add_op = [&](int k, int col, int word) -> bool {
        if (nops >= AGG_MAX_OPS) return false;
        plan->ops[nops++] = AccOp{k, col, word};
        return true;
    };
    const char* too_many = "aggregate: too many accumulators for one operator (split the SELECT list)";
    for (int i = 0; i < n_funcs; i++) {
        int f = funcs[i], t = in_types[i], c = col_of[i];
        FuncOut& o = outs[i];
        o = FuncOut{f, t, in_flags ? in_flags[i] : 0, -1, -1, -1};
        if (f == VNM_COUNT_STAR) {
            if (w_rows < 0) {
                w_rows = new_word(M_ADD_U64);
                if (w_rows < 0 || !add_op(A_COUNT_ROWS, -1, w_rows)) return set_error(too_many);
            }
            o.w_a = w_rows;
            continue;
        }
        if (t < VNM_I8 || t > VNM_F64) {
            switch (f) {
                case VNM_MIN: case VNM_MAX: return set_error("Column data type is not supported by min()/max().");
                case VNM_SUM: return set_error("Column data type is not supported by sum().");
                case VNM_AVG: return set_error("Column data type is not supported by avg().");
                default: return set_error("Unsupported data type for aggregation column.");
            }
        }
        if (w_valid[c] < 0) {
            w_valid[c] = new_word(M_ADD_U64);
            if (w_valid[c] < 0 || !add_op(A_COUNT_VALID, c, w_valid[c])) return set_error(too_many);
        }
        o.w_valid = w_valid[c];
        if (f == VNM_COUNT) {
            o.w_a = w_valid[c];
        } else if (f == VNM_SUM || f == VNM_AVG) {
            if (w_sum[c] < 0) {
                if (type_is_float(t)) {
                    // compensated sum: (hi, lo) word pair, see M_ADD_F64C.  VNM_AGG_PLAIN_FSUM=1 keeps the single
                    // float64 word (order-dependent rounding; measurement aid only)
                    static const bool plain = getenv("VNM_AGG_PLAIN_FSUM") != nullptr;
                    w_sum[c] = new_word(plain ? M_ADD_F64 : M_ADD_F64C);
                    if (w_sum[c] < 0 || (!plain && new_word(M_ADD_F64) < 0) || !add_op(A_SUM_F64, c, w_sum[c])) return set_error(too_many);
                } else if (t == VNM_I64 || t == VNM_U64) {
                    w_sum[c] = new_word(M_ADD_U64);
                    w_hi[c] = new_word(M_ADD_U64);
                    if (w_sum[c] < 0 || w_hi[c] < 0 || !add_op(A_SUM_LO32, c, w_sum[c]) ||
                        !add_op(t == VNM_I64 ? A_SUM_HI32S : A_SUM_HI32U, c, w_hi[c]))
                        return set_error(too_many);
                } else {
                    w_sum[c] = new_word(M_ADD_U64);
                    if (w_sum[c] < 0 || !add_op(A_SUM_I64, c, w_sum[c])) return set_error(too_many);
                }
            }
            o.w_a = w_sum[c];
            o.w_b = w_hi[c];
            if (type_is_float(t) && plan->merge[w_sum[c]] == M_ADD_F64C) o.w_b = w_sum[c] + 1;
        } else if (f == VNM_MIN) {
            if (w_min[c] < 0) {
                w_min[c] = new_word(M_MIN_U64);
                if (w_min[c] < 0 || !add_op(A_MIN, c, w_min[c])) return set_error(too_many);
            }
            o.w_a = w_min[c];
        } else if (f == VNM_MAX) {
            if (w_max[c] < 0) {
                w_max[c] = new_word(M_MAX_U64);
                if (w_max[c] < 0 || !add_op(A_MAX, c, w_max[c])) return set_error(too_many);
            }
            o.w_a = w_max[c];
        } else {
            return set_error("Unrecognized Aggregate function type.");
        }
    }
    if (nw == 0) {  // keys only (SELECT k ... GROUP BY k): keep one dummy row counter so every group has state
        w_rows = new_word(M_ADD_U64);
        add_op(A_COUNT_ROWS, -1, w_rows);
    }
    plan->n_words = nw;
    plan->n_ops = nops;
    return 0;
}

// ---- 128-bit helpers restating the reference's hugeint conversions ---------------------------------
// Hugeint::TryCast<double>  vinum_cpp/src/common/huge_int.cpp:395-406
static double hugeint_to_double(i128 x) {
    uint64_t lower = (uint64_t)(u128)x;
    int64_t upper = (int64_t)(x >> 64);
    if (upper == -1) return -(double)(UINT64_MAX - lower) - 1;
    return (double)lower + (double)upper * (double)UINT64_MAX;
}
// hugeint_try_cast_integer  huge_int.cpp:334-355 (INT64_MIN itself does not fit: strict '>')
static bool fits_i64(i128 x) {
    uint64_t lower = (uint64_t)(u128)x;
    int64_t upper = (int64_t)(x >> 64);
    if (upper == 0) return lower <= (uint64_t)INT64_MAX;
    if (upper == -1) return lower > UINT64_MAX - (uint64_t)INT64_MAX;
    return false;
}
static bool fits_u64(i128 x) { return (int64_t)(x >> 64) == 0; }

// float64 sum of a group: hi (+ lo of the compensated pair, see M_ADD_F64C)
static inline double fsum_word(const FuncOut& fo, const uint64_t* const* words, int64_t r) {
    double hi, lo = 0.0;
    memcpy(&hi, &words[fo.w_a][r], 8);
    if (fo.w_b >= 0) memcpy(&lo, &words[fo.w_b][r], 8);
    return fsum2(hi, lo);
}

static inline i128 sum128(const FuncOut& fo, const uint64_t* const* words, int64_t r) {
    uint64_t lo = words[fo.w_a][r];
    uint64_t hi = words[fo.w_b][r];
    if (fo.in_type == VNM_I64) return ((i128)(int64_t)hi << 32) + (i128)(u128)lo;
    return (i128)(((u128)hi << 32) + (u128)lo);
}

int finalize_func(const FuncOut& fo, int64_t n, const uint64_t* const* words, void* cells16, uint8_t* valid,
                  int* out_kind) {
    uint8_t* cells = (uint8_t*)cells16;
    memset(cells, 0, (size_t)n * 16);
    const int t = fo.in_type;
    auto put8 = [&](int64_t r, const void* p) { memcpy(cells + r * 16, p, 8); };
    switch (fo.func) {
        case VNM_COUNT_STAR:
        case VNM_COUNT:
            for (int64_t r = 0; r < n; r++) { put8(r, &words[fo.w_a][r]); valid[r] = 1; }
            *out_kind = VNM_OUT_U64;
            return 0;
        case VNM_MIN:
        case VNM_MAX:
            for (int64_t r = 0; r < n; r++) {
                valid[r] = words[fo.w_valid][r] > 0;
                if (!valid[r]) continue;
                uint64_t e = words[fo.w_a][r];
                if (type_is_float(t)) { double d = dec_f64(e); put8(r, &d); }
                else if (type_is_unsigned(t)) { put8(r, &e); }
                else { int64_t v = dec_i64(e); put8(r, &v); }
            }
            *out_kind = type_is_float(t) ? VNM_OUT_F64 : (type_is_unsigned(t) ? VNM_OUT_U64 : VNM_OUT_I64);
            return 0;
        case VNM_SUM: {
            if (t == VNM_I64 || t == VNM_U64) {
                for (int64_t r = 0; r < n; r++)
                    if (words[fo.w_valid][r] >= (1ULL << 32)) return set_error("aggregate: a group holds 2^32 or more non-NULL inputs of a 64-bit integer SUM / AVG (the exact 128-bit lanes hold 2^32 - 1 per group)");
                // SumOverflowFunc::Summarize agg_funcs.h:358-397: one group that does not fit promotes the
                // whole column to decimal128(38,0)
                bool overflow = false;
                for (int64_t r = 0; r < n && !overflow; r++) {
                    if (words[fo.w_valid][r] == 0) continue;
                    i128 s = sum128(fo, words, r);
                    if (!(t == VNM_I64 ? fits_i64(s) : fits_u64(s))) overflow = true;
                }
                for (int64_t r = 0; r < n; r++) {
                    valid[r] = words[fo.w_valid][r] > 0;
                    if (!valid[r]) continue;
                    i128 s = sum128(fo, words, r);
                    if (overflow) memcpy(cells + r * 16, &s, 16);
                    else { uint64_t lo = (uint64_t)(u128)s; put8(r, &lo); }
                }
                *out_kind = overflow ? VNM_OUT_DEC128 : (t == VNM_I64 ? VNM_OUT_I64 : VNM_OUT_U64);
                return 0;
            }
            for (int64_t r = 0; r < n; r++) {
                valid[r] = words[fo.w_valid][r] > 0;
                if (!valid[r]) continue;
                uint64_t w = words[fo.w_a][r];
                if (type_is_float(t)) { const double d = fsum_word(fo, words, r); memcpy(&w, &d, 8); }
                if (t == VNM_I32 && (fo.in_flags & VNM_FLAG_SUM32)) w = (uint64_t)(int64_t)(int32_t)(uint32_t)w;
                put8(r, &w);
            }
            if (type_is_float(t)) *out_kind = VNM_OUT_F64;
            else if (t == VNM_I32 && (fo.in_flags & VNM_FLAG_SUM32)) *out_kind = VNM_OUT_I32;
            else *out_kind = type_is_unsigned(t) ? VNM_OUT_U64 : VNM_OUT_I64;
            return 0;
        }
        case VNM_AVG: {
            // AvgFunc::Summarize agg_funcs.h:482-491, ComputeAvg :519-540; float32 output for 8/16-bit
            // integer inputs (agg_func_factory.cpp:179-196)
            const bool out_f32 = (t == VNM_I8 || t == VNM_I16 || t == VNM_U8 || t == VNM_U16);
            for (int64_t r = 0; r < n; r++) {
                uint64_t cnt = words[fo.w_valid][r];
                valid[r] = cnt > 0;
                if (!valid[r]) continue;
                double avg;
                if ((t == VNM_I64 || t == VNM_U64) && cnt >= (1ULL << 32)) return set_error("aggregate: a group holds 2^32 or more non-NULL inputs of a 64-bit integer SUM / AVG (the exact 128-bit lanes hold 2^32 - 1 per group)");
                if (t == VNM_I64 || t == VNM_U64) {
                    i128 s = sum128(fo, words, r);
                    i128 c = (i128)(int64_t)cnt;
                    i128 q = s / c, rem = s % c;
                    avg = hugeint_to_double(q);
                    avg += hugeint_to_double(rem) / (double)cnt;
                } else if (type_is_float(t)) {
                    avg = fsum_word(fo, words, r) / (double)cnt;
                } else if (type_is_unsigned(t)) {
                    avg = (double)words[fo.w_a][r] / (double)cnt;
                } else {
                    avg = (double)(int64_t)words[fo.w_a][r] / (double)cnt;
                }
                if (out_f32) { float f = (float)avg; memcpy(cells + r * 16, &f, 4); }
                else put8(r, &avg);
            }
            *out_kind = out_f32 ? VNM_OUT_F32 : VNM_OUT_F64;
            return 0;
        }
    }
    return set_error("Unrecognized Aggregate function type.");
}

}  // namespace vnm
