// Aggregate plan shared by the device kernels (vnm_agg.hip) and the host finaliser (vnm_finalize.cpp).
#pragma once
#include "vnm_common.hpp"

namespace vnm {

constexpr int AGG_MAX_KEYS = 8;
constexpr int AGG_MAX_FUNCS = 64;
constexpr int AGG_MAX_WORDS = 40;   // 64-bit accumulator words per group
constexpr int AGG_MAX_OPS = 48;     // per-row accumulator updates
constexpr int AGG_MAX_COLS = 20;    // distinct input columns

// Per-row accumulator update kinds.  The reference keeps one heap object per (group, function)
// (agg_funcs.h:97-542); here every function is lowered onto 64-bit words that merge with a
// commutative, associative op, which is what lets rows be combined in any order, in LDS, and
// across GPUs.
enum AccKind : int {
    A_COUNT_ROWS = 0,  // word += 1                                   CountStarFunc  agg_funcs.h:97-127
    A_COUNT_VALID,     // word += valid                               CountFunc      :129-161
    A_SUM_F64,         // valid: (double)word += (double)x            SumFunc/AvgFunc double path :294-305,455-467
                       //        (compensated: the rounding error of every add goes to word + 1, see M_ADD_F64C)
    A_SUM_I64,         // valid: word += (int64)x  (wraparound)       SumFunc int64_t/uint64_t accumulators
    A_SUM_LO32,        // valid: word += (uint64)(x & 0xffffffff)     } 128-bit sum of int64/uint64 inputs
    A_SUM_HI32S,       // valid: word += (int64)x >> 32 (arithmetic)  } (SumOverflowFunc :319-435, hugeint AVG)
    A_SUM_HI32U,       // valid: word += (uint64)x >> 32              } value = hi * 2^32 + lo, exact for < 2^32 rows/group
    A_MIN,             // valid: word = min(word, enc(x))             MinMaxFunc :164-216 (total-order encoding)
    A_MAX,             // valid: word = max(word, enc(x))
};
// M_ADD_F64C: high word of a COMPENSATED float64 sum.  Every add into it (per row, LDS table -> HBM table, rank -> rank)
// is a returning atomic add; the exact rounding error of that add (Knuth's TwoSum, an error-free transformation) is added
// to the word that follows it (an ordinary M_ADD_F64 word).  hi + lo then equals the exact sum up to second-order terms
// (~n * 2^-106 relative to sum |x|) whatever the order of the adds was, so the finalised double is the correctly rounded
// exact sum except within ~2^-50 of a rounding tie -- where a sequential float64 loop (the reference, agg_funcs.h:294-305)
// is off by up to n/2 ULP and a plain atomic sum differs from run to run.
enum MergeKind : int { M_ADD_U64 = 0, M_ADD_F64, M_MIN_U64, M_MAX_U64, M_ADD_F64C };

struct AccOp {
    int kind;  // AccKind
    int col;   // index into the distinct input column list (-1 for A_COUNT_ROWS)
    int word;  // accumulator word
};

// How function i reads its result out of the accumulator words (host finaliser).
struct FuncOut {
    int func;       // vnm_agg_func
    int in_type;    // vnm_type of the input column
    int in_flags;
    int w_valid;    // word holding the count of non-null inputs (-1: none)
    int w_a;        // main word (count / sum / lo / min / max)
    int w_b;        // second word (hi of a 128-bit sum) or -1
};

struct AggPlan {
    int kind;      // vnm_agg_kind
    int n_keys;
    int kw;        // key words stored per group in a dense run: n_keys values + 1 null-mask word (0 for ONE_GROUP)
    int n_words;
    int n_ops;
    int n_cols;
    int key_types[AGG_MAX_KEYS];
    int merge[AGG_MAX_WORDS];   // MergeKind per word
    AccOp ops[AGG_MAX_OPS];
};

// Float sums start at -0.0, the additive identity of IEEE arithmetic (x + -0.0 == x for every x, -0.0 + -0.0 == -0.0, while
// +0.0 + -0.0 == +0.0): SumFunc starts from the group's first value (agg_funcs.h:286-305), so a group whose inputs are all -0.0
// sums to -0.0 there -- and here, as long as EVERY accumulator a row can pass through starts at -0.0 (LDS tables, per-thread
// partials, HBM tables, merge kernels).  The compensation words may start at either zero: hi + lo goes through fsum2.
constexpr uint64_t F64_NEG_ZERO = 0x8000000000000000ULL;
__host__ __device__ inline uint64_t merge_init(int mk) { return mk == M_MIN_U64 ? ~0ULL : ((mk == M_ADD_F64 || mk == M_ADD_F64C) ? F64_NEG_ZERO : 0ULL); }
// hi + lo of a compensated sum; a zero compensation term (of either sign) leaves hi as it is, -0.0 included
__host__ __device__ inline double fsum2(double hi, double lo) { return lo == 0.0 ? hi : hi + lo; }

// Lowers (funcs, input types) onto words/ops.  col_of_func[i] = index in the distinct-column list
// (or -1), n_cols = number of distinct columns, col_first_func[c] = a function that reads column c.
int build_plan(int kind, int n_keys, const int* key_types, int n_funcs, const int* funcs, const int* in_types,
               const int* in_flags, const int* func_col_id /* caller-provided distinct id per func, -1 none */,
               AggPlan* plan, FuncOut* outs);

// Host finalisation of one function from dense accumulator words (host memory).
// Mirrors Summarize/ComputeAvg: agg_funcs.h:72-80 (generic), :358-397 (int64 sum -> decimal128 promotion),
// :482-491 + :519-540 (AVG incl. the 128-bit divmod path).  cells16: n * 16 bytes.
int finalize_func(const FuncOut& fo, int64_t n, const uint64_t* const* words, void* cells16, uint8_t* valid,
                  int* out_kind);

}  // namespace vnm
