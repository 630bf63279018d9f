// Hash group-by aggregate for gfx950.
//
// Replaces BaseAggregate::Next's scalar row loop (vinum_cpp/src/operators/aggregate/base_aggregate.cpp:
// 23-45), the robin_hood maps of Single/MultiNumericalHashAggregate (single_numerical_hash_aggregate.cpp:
// 15-46, multi_numerical_hash_aggregate.cpp:17-43) and OneGroupAggregate (one_group_aggregate.cpp:9-26).
//
// Design facts measured on MI355X (tools/microbench.hip, profiles/microbench_r01.txt):
//   * device-scope atomics top out at ~24 G/s regardless of table size  -> never one per row;
//   * LDS atomics sustain > 1 T row-updates/s chip-wide                  -> every row lands in LDS;
//   * a grid of 8 x 256 CUs grid-striding 16 B/lane streams 6.3 TB/s.
// So rows are pre-aggregated in a per-workgroup LDS hash table (keys + 64-bit accumulator words that
// merge commutatively); only table flushes touch the HBM-resident table, with agent-scope atomics.
// The fused WHERE predicate is evaluated in the scan, so no filtered batch is ever materialised.
#include <memory>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "vnm_agg.hpp"

namespace vnm {

constexpr uint64_t EMPTY = ~0ULL;
constexpr uint64_t LOCKED = ~0ULL - 1;
constexpr int AGG_BLOCK = 1024;        // LDS kernel: 16 waves, one workgroup per CU
#ifndef VNM_AGG_R
#define VNM_AGG_R 8
#endif
constexpr int AGG_ROWS_PER_THREAD = VNM_AGG_R;
constexpr int AGG_TILE = AGG_BLOCK * AGG_ROWS_PER_THREAD;
constexpr int AGG_LDS_BUDGET = 128 * 1024;
constexpr int AGG_MAX_PROBES = 48;
constexpr int OG_BLOCK = 256;

// HBM-resident table: tag[] (the key itself on the single-key path), optional wide key words, and
// SoA accumulator words.  Arrays have cap + 2 entries: [cap] = the group whose key equals the EMPTY
// sentinel, [cap + 1] = the NULL-key group (single_numerical_hash_aggregate.cpp:24-32).
struct GTable {
    uint64_t* tag;
    uint64_t* keyw;  // kwt * stride (wide keys only)
    uint64_t* acc;   // n_words * stride
    uint64_t cap;    // power of two
    uint64_t stride; // cap + 2
    unsigned long long* ctl;  // [0] ticket  [1] overflow flag  [2] fill (groups in table)  [3] dense count
    int kwt;         // wide key words (0 on the single-key path)
    int n_words;
};

// ---- expressions inside aggregates, evaluated in registers -------------------------------------------------------------
// `SELECT k, sum((1 - total) * (2 + tax) * (1 - tip)) ... GROUP BY k` (vinum/tests/test_query_results.py:436-443): the
// reference's planner projects the expression into a temporary column first (planner.py:384-417, one NumPy pass and one
// n-row temporary per AST node) and the aggregate re-reads it.  Here the input of the hot-shape aggregate may BE an
// expression: a postfix program over up to four float64 columns without NULLs, evaluated per row pair where the scan /
// partition kernels used to load the value column -- no materialised column, no extra pass.  float64 +, -, *, /, negation
// one IEEE operation per AST node (the interpreter's dispatch keeps the compiler from contracting a*b+c into an FMA), so the
// values are bit-identical to the projection kernel's and to NumPy's.  Anything else (other types, NULLs, more columns,
// deeper stacks, other aggregate shapes) is materialised by vnm_project first (vnm_agg_next_device_expr).
constexpr int EXR_MAX_INS = 16, EXR_MAX_COLS = 4, EXR_MAX_DEPTH = 4;
struct ExprIns { int op; int arg; double imm; };
struct ExprProg {
    int n;
    const double* cols[EXR_MAX_COLS];
    ExprIns ins[EXR_MAX_INS];
};
// the stack lives in four named registers (a dynamically indexed array would go to scratch memory)
template <typename V, typename LOADER>
__device__ __forceinline__ V expr_eval(const ExprProg& e, LOADER&& load_col, V splat_zero) {
    V s0 = splat_zero, s1 = splat_zero, s2 = splat_zero, s3 = splat_zero;
    for (int i = 0; i < e.n; i++) {
        const int op = e.ins[i].op;
        if (op == VNM_EX_COL || op == VNM_EX_CONST_F) {
            s3 = s2; s2 = s1; s1 = s0;
            s0 = op == VNM_EX_COL ? load_col(e.ins[i].arg) : splat_zero + e.ins[i].imm;
        } else if (op == VNM_EX_NEG) {
            s0 = -s0;
        } else {
            const V b = s0, a = s1;
            s1 = s2; s2 = s3;
            switch (op) {
                case VNM_EX_ADD: s0 = a + b; break;
                case VNM_EX_SUB: s0 = a - b; break;
                case VNM_EX_MUL: s0 = a * b; break;
                default: s0 = a / b; break;
            }
        }
    }
    return s0;
}
typedef double expr_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 expr_eval2(const ExprProg& e, int64_t r) {   // rows r, r + 1 (r even: 16-byte loads)
    const expr_v2 z = {0.0, 0.0};
    const expr_v2 v = expr_eval<expr_v2>(e, [&](int c) { return *(const expr_v2*)(e.cols[c] + r); }, z);
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ double expr_eval1(const ExprProg& e, int64_t r) {
    return expr_eval<double>(e, [&](int c) { return e.cols[c][r]; }, 0.0);
}

// One record batch of a STREAM handed to a kernel as part of one logical batch (vnm_agg_set_async: batches wait in the operator and
// go to the device together -- a launch per 2^24-row batch costs more in launch gaps, LDS table start-up and partial flushes than
// its rows).  The hot shape only: 8-byte key, one plain float64 input column, float64 predicate column or none.  A segment owns
// the tiles [first_tile, first_tile + ceil(nrows / tile)) of the launch; tiles never span segments.
struct VSeg {
    const uint64_t* kp;      // key values (Arrow offset applied; 16-byte aligned)
    const uint64_t* vp;      // input column values
    const double* pp;        // predicate column values (the input column itself when the predicate reads it)
    const uint8_t* vvalid;   // validity bitmap of the input column or null (dense path over a nullable value column)
    int64_t voff;
    int64_t nrows;
    int64_t first_tile;
};

// The segment table is read through the CONSTANT address space (nobody writes it while the kernel runs): uniform loads from it are
// scalar loads (s_load: the scalar cache, results in scalar registers, no wait on the wave's outstanding vector loads).  Through a
// plain global pointer the same reads are vector loads -- the compiler cannot scalarise loads from memory the kernel also stores
// to -- whose latency, behind a saturated memory system, sat on every tile's critical path and whose results occupied vector
// registers (58 segments, G = 7: 4.1 ms against 2.7 for the same rows as one batch).
typedef const VSeg __attribute__((address_space(4)))* VSegConst;
__device__ __forceinline__ VSegConst seg_table(const VSeg* p) { return (VSegConst)(uintptr_t)p; }
// A value every lane of the wave holds alike, moved into scalar registers.  Loads through a.segs are uniform, but the compiler
// cannot scalarise loads from memory the kernel also stores to: the segment pointers then sit in vector registers -- two sets of
// them, this tile's and the next one's -- and the scan kernel spilled 46-116 VGPRs (58 segments, G = 7: 4.1 ms against 2.7 for
// the same rows as one batch).
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) { return (const T*)uniform_u64((uint64_t)p); }

struct AggArgs {
    AggPlan plan;
    const VSeg* segs;  // agg_hot_kernel, nseg > 0: the rows are these segments (keys[0] / cols[0] / pred describe the first one)
    int nseg;
    int has_expr;      // the (only) input column of this hot-shape plan is `expr`, not a.cols[0]
    ExprProg expr;
    vnm_dcol keys[AGG_MAX_KEYS];
    vnm_dcol cols[AGG_MAX_COLS];
    vnm_dcol pred;
    Predicate p;
    GTable g;
    int64_t nrows;
    int64_t ntiles;
    int64_t fill_limit;   // take a new tile only while fill + margin <= fill_limit
    int64_t margin;
    int lds_slots;
    unsigned int* progress;  // per-block loop index to resume from
    // hot-shape kernel (single 8-byte key, one float64 input column, no validity bitmaps)
    int hot_w_rows, hot_w_valid, hot_w_sum;
    int hot_comp;      // float64 sums are compensated (hi, lo) pairs: M_ADD_F64C
    int hot_pred_is_v;
    // extended hot shape: every accumulator kind over ONE 8-byte input column without NULLs (or no input at all)
    int hot_w[9];      // word of each AccKind, -1 = absent
    int hot_w2[9];     // ... for a second input column (agg_hot_kernel<TWO>)
    int hot_vtype2;
    unsigned long long hot_wpack, hot_wpack2;  // the same tables, 6 bits per kind (63 = absent, COUNT(*) excluded): scalar registers
    int hot_vtype;     // VNM_F64 / VNM_I64 / VNM_U64
    int hot_has_val;
    const ulonglong2* ent;  // agg_hot_kernel<FROM_ENT>: (key, value bits) entries spilled by the partitioned path
    int part_generic;  // partitioned path with a generic accumulator program over one 8-byte column (or none)
    int part_vtype;
    int part_vtypes[6];  // wide entries: type of each input column
    int part_wide;       // wide entries (several input columns, NULLs, narrow types, any predicate column)
    int part_vmask;      // ... with a validity word
    int debug;  // timing experiments (VNM_AGG_DEBUG): 1 = no accumulator ops, 2 = no probes either, 4 = no room check
    // agg_hotn_kernel: {COUNT(*), COUNT, SUM, AVG} over THREE to SIX plain float64 columns
    int hn_w_sum[6];    // word of each column's sum (-1: the column is only counted)
    int hn_w_base;      // the count word the kernel maintains; the others are copies of it before a flush (no NULLs: every count is the row count)
    int hn_n_copy;
    int hn_w_copy[7];
    int hn_pred_col;    // the predicate column is this input column (-1: a column of its own, a.pred)
    int hn_ctype[6];    // VNM_F64 / VNM_I64 / VNM_U64
    int hn_iw[6][4];    // integer sums of the column: word of A_SUM_I64 / A_SUM_LO32 / A_SUM_HI32S / A_SUM_HI32U (-1: absent)
    int hn_any_int;
    int hn_wmm[6][2];   // MIN / MAX words of the column (-1: absent)
    int hn_any_mm;
};

// ---- global table primitives ------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ld_agent(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint64_t* p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Exact rounding error of s = fl(a + b) (Knuth's TwoSum: a + b == s + e in real arithmetic).  Zero when the sum
// overflowed or an operand was not finite, so that inf / NaN results are what a plain sum gives.
__device__ __forceinline__ double two_sum_err(double a, double b, double s) {
    const double bb = s - a;
    const double e = (a - (s - bb)) + (b - bb);
    return (s - s == 0.0) ? e : 0.0;
}
// Compensated add into the (hi, lo) word pair of an M_ADD_F64C accumulator: hi = fl(hi + x) with a RETURNING atomic,
// the exact error of that add goes to lo (the word `ws` elements further on).  Exact adds (the benchmark's quantised
// data) never touch lo.
__device__ __forceinline__ void g_add_f64c(uint64_t* p, int64_t ws, double x) {
    const double old = __hip_atomic_fetch_add((double*)p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double e = two_sum_err(old, x, old + x);
    if (e != 0.0) __hip_atomic_fetch_add((double*)(p + ws), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void l_add_f64c(uint64_t* p, int ws, double x) {
    const double old = __hip_atomic_fetch_add((double*)p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const double e = two_sum_err(old, x, old + x);
    if (e != 0.0) __hip_atomic_fetch_add((double*)(p + ws), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ws: distance (in elements) between consecutive accumulator words of one group (M_ADD_F64C updates word + 1 too)
__device__ __forceinline__ void g_merge(uint64_t* p, int mk, uint64_t v, int64_t ws) {
    switch (mk) {
        case M_ADD_U64: __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
        case M_ADD_F64: __hip_atomic_fetch_add((double*)p, __longlong_as_double((long long)v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
        case M_ADD_F64C: g_add_f64c(p, ws, __longlong_as_double((long long)v)); break;
        case M_MIN_U64: __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
        default: __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
    }
}
__device__ __forceinline__ void l_merge(uint64_t* p, int mk, uint64_t v, int ws) {
    switch (mk) {
        case M_ADD_U64: __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break;
        case M_ADD_F64: __hip_atomic_fetch_add((double*)p, __longlong_as_double((long long)v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break;
        case M_ADD_F64C: l_add_f64c(p, ws, __longlong_as_double((long long)v)); break;
        case M_MIN_U64: __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break;
        default: __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break;
    }
}

// single 64-bit key: the tag word IS the key; claim by CAS from EMPTY
// New groups are counted into a block-local LDS counter (*newc) and folded into the table's fill word once per
// tile: one agent-scope atomic per NEW GROUP on a single word capped merges at ~1 group/ns (122 ms per 1e8).
__device__ __forceinline__ uint64_t gt_find_single(const GTable& g, uint64_t key, unsigned* newc) {
    const uint64_t mask = g.cap - 1;
    uint64_t h = hash_u64(key) & mask;
    for (uint64_t probes = 0;; probes++) {
        if (probes > mask) {  // table full: cannot happen while the room checks hold; never spin forever
            __hip_atomic_store(&g.ctl[1], 2ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return g.cap;
        }
        uint64_t k = ld_agent(&g.tag[h]);
        if (k == key) return h;
        if (k == EMPTY) {
            uint64_t expected = EMPTY;
            if (__hip_atomic_compare_exchange_strong(&g.tag[h], &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) {
                atomicAdd(newc, 1u);
                return h;
            }
            if (expected == key) return h;
        }
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ uint64_t wide_tag(const uint64_t* kw, int n) {
    uint64_t h = 0x9E3779B97F4A7C15ULL * (uint64_t)n;
    for (int i = 0; i < n; i++) {
        uint64_t x = kw[i];
        h ^= (uint64_t)hash_u64(x) * 0x9E3779B1ULL + ((uint64_t)hash_u64(x ^ 0x5bd1e995) << 32) + (h << 6) + (h >> 2);
    }
    return h & 0x7FFFFFFFFFFFFFFFULL;
}

// wide keys: tag = 63-bit hash; EMPTY -> LOCKED -> tag.  The claimer publishes the key words with
// write-through agent-scope stores, drains them, then publishes the tag (no lane ever waits inside the
// critical section, so same-wave spinners cannot deadlock).
__device__ __forceinline__ uint64_t gt_find_wide(const GTable& g, const uint64_t* kw, uint64_t tagv, unsigned* newc) {
    const uint64_t mask = g.cap - 1;
    uint64_t h = (tagv ^ (tagv >> 29)) & mask;
    for (;;) {
        uint64_t t = ld_agent(&g.tag[h]);
        if (t == tagv) {
            bool eq = true;
            for (int i = 0; i < g.kwt; i++) eq = eq && (ld_agent(&g.keyw[(uint64_t)i * g.stride + h]) == kw[i]);
            if (eq) return h;
            h = (h + 1) & mask;
            continue;
        }
        if (t == EMPTY) {
            uint64_t expected = EMPTY;
            if (__hip_atomic_compare_exchange_strong(&g.tag[h], &expected, LOCKED, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) {
                for (int i = 0; i < g.kwt; i++) st_agent(&g.keyw[(uint64_t)i * g.stride + h], kw[i]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                st_agent(&g.tag[h], tagv);
                atomicAdd(newc, 1u);
                return h;
            }
            continue;  // someone else is claiming this slot: look at it again
        }
        if (t == LOCKED) continue;
        h = (h + 1) & mask;
    }
}

// one thread per block folds the block's new-group count into the table's fill word
__device__ __forceinline__ void fold_new(const GTable& g, unsigned* s_new) {
    unsigned v = atomicExch(s_new, 0u);
    if (v) atomicAdd(&g.ctl[2], (unsigned long long)v);
}

// ---- per-row accumulator contributions ----------------------------------------------------------------
// Returns false when the op contributes nothing for this row (NULL input); otherwise the value to merge.
__device__ __forceinline__ bool op_value(const AccOp& op, const vnm_dcol* cols, int64_t row, uint64_t* out) {
    if (op.kind == A_COUNT_ROWS) { *out = 1; return true; }
    const vnm_dcol& c = cols[op.col];
    if (!col_valid(c, row)) return false;
    switch (op.kind) {
        case A_COUNT_VALID: *out = 1; return true;
        case A_SUM_F64: *out = (uint64_t)__double_as_longlong(col_f64(c, row)); return true;
        case A_SUM_I64: *out = (uint64_t)col_i64(c, row); return true;
        case A_SUM_LO32: *out = (uint64_t)col_i64(c, row) & 0xFFFFFFFFULL; return true;
        case A_SUM_HI32S: *out = (uint64_t)(col_i64(c, row) >> 32); return true;
        case A_SUM_HI32U: *out = (uint64_t)col_i64(c, row) >> 32; return true;
        default:  // A_MIN / A_MAX on the order-preserving encoding
            if (type_is_float(c.type)) *out = enc_f64(col_f64(c, row));
            else if (type_is_unsigned(c.type)) *out = (uint64_t)col_i64(c, row);
            else *out = enc_i64(col_i64(c, row));
            return true;
    }
}

// Tiles are statically strided over the grid (tile = block + i * grid): no shared ticket word, so the
// scan never serialises on one atomic (a CAS ticket per 2048-row tile cost 1.3 s per 1e9 rows).  Before
// each tile the block checks that the HBM table has room for everything that can be in flight; if not it
// raises the overflow flag, parks its loop index in progress[block] and exits so the host can grow the
// table and relaunch the same grid from where every block stopped.
__device__ __forceinline__ bool table_has_room(const AggArgs& a, unsigned* s_new) {
    fold_new(a.g, s_new);
    unsigned long long fill = __hip_atomic_load(&a.g.ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int64_t)fill + a.margin > a.fill_limit) {
        __hip_atomic_store(&a.g.ctl[1], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    return true;
}

#include "vnm_agg_scan.inc"

#include "vnm_agg_tuple.inc"

#include "vnm_agg_onegroup.inc"

#include "vnm_agg_table.inc"

#include "vnm_agg_part.inc"

#include "vnm_agg_pack.inc"

#include "vnm_agg_dense.inc"

#include "vnm_agg_finalize.inc"

}  // namespace vnm

// ==========================================================================================================
// host side
// ==========================================================================================================
using namespace vnm;

struct DensePending {
    DFinalArgs df;         // what every batch shares: the code map, the words of the plan, the flags block
    int tb = 0, levels = 0, p1 = 0, fsplits = 1;
    int64_t nfinal = 0;
    int64_t dstride = 0;   // bound of the groups the pass can produce
    int64_t rows = 0;      // rows behind the sets: below 2^32, the slots of the final pass count a group's rows in 32 bits
    // exact-add statistics over the batches of the pass (exact_track): known only while every batch came through a scatter that keeps them
    bool x_known = true;
    uint32_t x_inv = 0, x_exp = 0;
    bool exact_ok() const { return x_known && exact_adds_proved(((unsigned long long)x_exp << 32) | x_inv, rows); }
    std::vector<DSet> sets;      // one per batch whose scatter passes are done
    std::vector<void*> blocks;   // pool blocks the entries live in
    DSet* dsets = nullptr;       // device copy of `sets` (refreshed by complete_pending)
    DTabSlot* table = nullptr;   // DF_TABLE result (kept until the handle goes: the exchange reads it)
    ~DensePending() { for (void* b : blocks) pool_free(b); pool_free(table); pool_free(dsets); }
};
constexpr size_t DP_MAX_SETS = 512;   // batches a deferred pass may span (then it runs, as a run, and a new one starts)
// the table a stream of small-range batches accumulates in (dscan_accumulate_kernel); groups are written by flush_scan_pending
struct DScanPending {
    DFinalArgs df{};   // code map + the words of the plan
    DScanTable t{};
    int slots = 0;
    uint64_t* part_sum = nullptr;   // the current batch's per-workgroup tables [cus][slots] (kept: no allocation per batch)
    float* part_lo = nullptr;
    uint32_t* part_cnt = nullptr;
    unsigned long long* flags = nullptr;
    ~DScanPending() { pool_free(t.sum); pool_free(t.lo); pool_free(t.cnt); pool_free(part_sum); pool_free(part_lo); pool_free(part_cnt); pool_free(flags); }
};

struct vnm_agg {
    AggPlan plan;
    FuncOut outs[AGG_MAX_FUNCS];
    int n_funcs = 0;
    int func_col[AGG_MAX_FUNCS];   // distinct column index per func (-1 for COUNT(*))
    int col_first_func[AGG_MAX_COLS];
    bool single = false;           // single 64-bit key path
    Predicate pred{};
    bool pred_set = false;
    int pred_op = 0, pred_is_float = 0;
    double pred_dval = 0;
    int64_t pred_ival = 0;
    int64_t hint = 0;
    GTable g{};
    bool have_table = false;
    int64_t rows_seen = 0;
    // dense result (device + host mirror)
    uint64_t* dkey = nullptr;
    uint64_t* dacc = nullptr;
    int64_t dstride = 0;
    int64_t n_groups = -1;
    std::vector<uint64_t> h_key, h_acc;
    bool host_ready = false;
    // dense run produced by the partitioned path (not yet merged into the HBM table)
    uint64_t* run_key = nullptr;
    uint64_t* run_acc = nullptr;
    int64_t run_stride = 0, run_n = 0;
    bool have_run = false;
    bool result_is_run = false;
    bool estimated = false;  // hint came from estimate_groups()
    int64_t merge_stride = 0;  // set by vnm_agg_merge_rows around vnm_agg_merge_device
    unsigned long long* run_dir = nullptr;  // partition directory of the run (partitioned path, one workgroup per partition)
    int64_t run_nfin = 0;
    // packed composite keys (multi-column GROUP BY through the single-key machinery)
    vnm_agg* inner = nullptr;
    bool pack_tried = false;
    bool key_only_failed = false;  // COUNT(*)-only programs: 8-byte entries overflowed a region once (skewed keys)
    // dense-key partitioned path (vnm_agg_dense.inc): 0 = range not sampled yet, 1 = code map valid, -1 = not applicable
    int dense_state = 0;
    DenseMap dmap{};
    int64_t dense_span = 0;  // 2^bits: upper bound of the groups a dense run can hold
    uint64_t dense_rlo = 0, dense_rhi = 0;  // the sampled (widened) range itself, as order-preserving unsigned images
    bool rank_aligned = false;  // vnm_agg_set_exchange_mode: only run layouts every rank derives identically
    // dense path, final pass DEFERRED (round 3): the scatter passes of the last batch are done, the direct-addressed final pass
    // has not run yet -- what it writes depends on who asks: another batch / finish() -> the dense partial state (a run),
    // vnm_agg_result_device_alloc -> the result columns themselves, vnm_agg_dense_table -> the tables for the multi-GPU exchange
    struct DensePending* pending = nullptr;
    double heavy_share = 0.0;   // share of the rows held by heavy keys in the estimator's sample (0: none seen, or never sampled)
    // A single narrow integer key (int8 .. uint32) is WIDENED to int64 / uint64 on arrival (round 5): every specialised path reads plain
    // 8-byte keys, and such keys took the interpreted scan and the generic entries at 1.5-2.9 x the time for fewer bytes.  The plan then
    // says int64 / uint64 (the key WORD of a narrow key is its sign- / zero-extended value anyway: array_iterators.h:215-217), the result
    // columns keep the type the caller declared.  The widened buffers live until no recorded batch needs them (sequence numbers).
    int key_out_type = -1;
    std::vector<std::pair<int64_t, void*>> widened;
    // ... and float32 INPUT columns under SUM / AVG / COUNT only (the reference sums float32 in double anyway: SumFunc<float, double>,
    // AvgFunc<float, double, double>, agg_func_factory.cpp:126-131, 206-211): widened to float64 on arrival, so that `sum(v), avg(v) WHERE v > x`
    // over a float32 column takes the float64 kernels.  A predicate over such a column compares in float32 in the reference (NumPy:
    // float32 array against a Python scalar) -- the same as comparing the widened values with the literal ROUNDED to float32.
    bool widen_in[AGG_MAX_FUNCS] = {};
    bool any_widen_in = false;
    bool pred_lit_rounded = false;   // pred_dval holds the literal rounded to float32 (the caller's literal: pred_user_*)
    int pred_user_is_float = 0;
    double pred_user_dval = 0.0;
    int64_t pred_user_ival = 0;
    // fixed-point entries on the dense path (round 6, DPartArgs::fx_q): 0 = the value column has not been sampled yet, 1 = on (every value of
    // the sample is m * 2^fx_qe with |m| < 2^31 and room to spare), -1 = off (the sample or a later row does not fit)
    int fx_state = 0, fx_qe = 0;
    bool pack_null_seen = false;     // packed composite keys: some batch brought a key column with a validity bitmap (NULL codes may be in the words)
    bool pring_off = false;          // the ring form of the hash-partition scatter failed once (a heavy key, a full region): the tile-sorting scatter from then on
    bool fx_narrow = false;          // ... with |m| < 2^18: the words of the last scatter level are 32 bits (dring_scatter_kernel<..., W32>)
    int fxn_state = 0, fxn_qe[3] = {0, 0, 0};   // ... and of the entries of two or three values (vnm_agg_fxn.inc)
    bool null_inputs_seen = false;   // some batch brought an input column with a validity bitmap: the HBM table may hold groups whose COUNT(v) differs
                                     // from COUNT(*) (or whose SUM is NULL) -- the side-table fold of the fused result columns assumes they do not
    bool count8_off = false;    // the counters of COUNT(*)-only programs overflowed once (dcount8_final_kernel): not again
    int count_cb = 0;           // ... their width once the bytes overflowed: 16
    struct DScanPending* scan_pending = nullptr;   // a stream of small-range batches: their table (see dense_scan_aggregate)
    bool dense_by_bound = false;   // the first batch went dense on the sample's LOWER bound of the group count (no estimate exists)
    bool range_given = false;   // vnm_agg_set_dense_range: the code range is the caller's (agreed by all ranks), not a sample's
    // expression input (vnm_agg_set_input_expr): the functions reading plan column expr_col get an expression's value
    int expr_col = -1;
    std::vector<vnm_expr_ins> expr_prog;
    int expr_ncols = 0;
    bool expr_fusable = false;   // the program fits the in-register evaluator (float64 + - * / neg, <= 16 ins, depth <= 4)
    bool expr_active = false;    // set around one vnm_agg_next_device call: evaluate expr_dev in the scan instead of a column
    ExprProg expr_dev{};
    PackParams pack{};
    int c_funcs[AGG_MAX_FUNCS], c_in_types[AGG_MAX_FUNCS], c_in_flags[AGG_MAX_FUNCS], c_in_col_ids[AGG_MAX_FUNCS];
    bool c_has_ids = false;
    // program split (round 3): more input columns than one partition entry carries (key + six words) used to mean the LDS
    // scan with its flush storms at large G.  The function list is cut into sub-operators over a few columns each, all over
    // the same key; every batch goes through each of them and their results are joined by key when the state is needed
    // (collapse_parts: a run in ascending key order).
    // tuple dictionary (round 3, tuple_gid_kernel): key sets that do not pack into one word.  tdict maps tuple -> group id;
    // `inner` aggregates by that id
    TDict tdict{};
    int64_t widest_est = 0;                // plan_packing: the largest distinct-count estimate of a single key column
    unsigned long long* tnext = nullptr;   // device: first group id not handed out yet
    bool tuple_mode = false;
    std::vector<vnm_agg*> parts;
    std::vector<std::vector<int>> part_funcs;   // [part][function of the part] -> function index here
    bool split_tried = false;
    // Asynchronous streams (round 4, vnm_agg_set_async): hot-shape batches WAIT here (the caller keeps their buffers alive) and go
    // to the device together as the segments of one logical batch -- at vnm_agg_sync / finish / result, when 2^30 rows or 256
    // batches are waiting, or when a batch of another shape arrives.  No host read-back, allocation or launch per next().
    bool async = false;
    struct QBatch { int64_t nrows; vnm_dcol key, col, pred; std::vector<vnm_dcol> ins; int64_t seq; };   // ins: one column per function (several input columns); seq: see cur_seq
    bool q_multi = false;                  // the waiting batches carry several input columns (they are cut into parts when they go to the device)
    std::vector<QBatch> q;
    int64_t q_rows = 0;
    bool q_pred_is_v = false;
    const std::vector<QBatch>* segs_active = nullptr;   // set around the one vnm_agg_next_device call that processes the queue
    // a NULLABLE single key through the dense path (round 4): set around one next_device_impl call whose keys[0] had its validity
    // stripped -- pass 1 reads it (dring_scatter_kernel<..., KN>) and sums the NULL-key rows into the HBM table's NULL slot
    const uint8_t* kn_valid = nullptr;
    int64_t kn_off = 0;
    bool kn_failed = false;      // the dense path did not take such a batch once: later ones go straight to the packed route
    std::vector<VSeg> seg_host;                         // the segment table of the last launch (kept until the next one: H2D source)
    // reference-exact float MIN / MAX under NaNs and mixed-sign zeros (round 5, vnm_agg_exact.inc): top-level handles only
    struct vnm_agg_exact* ex = nullptr;
    // Which batches a stream still needs the buffers of (vnm_agg_waiting): every vnm_agg_next_device call on a TOP-LEVEL handle takes the
    // next sequence number; a recorded batch keeps the number of the call that brought it, also when it is handed on to the operators
    // the handle runs for itself (parts, the suffix operator of an ordered MIN / MAX stream), which record under the caller's number.
    bool child = false;
    int64_t seq = 0;        // calls so far (top-level handles)
    int64_t cur_seq = -1;   // number of the batch being handed in
};

namespace {

int table_alloc(vnm_agg* h, GTable* g, uint64_t cap, hipStream_t s) {
    memset(g, 0, sizeof(*g));
    g->cap = cap;
    g->stride = cap + 2;
    g->kwt = h->single || h->plan.n_keys == 0 ? 0 : h->plan.n_keys + 1;
    g->n_words = h->plan.n_words;
    g->tag = (uint64_t*)pool_alloc(g->stride * 8);
    g->acc = (uint64_t*)pool_alloc(g->stride * 8 * (size_t)g->n_words);
    g->ctl = (unsigned long long*)pool_alloc(64);
    if (g->kwt) g->keyw = (uint64_t*)pool_alloc(g->stride * 8 * (size_t)g->kwt);
    if (!g->tag || !g->acc || !g->ctl || (g->kwt && !g->keyw)) return 1;
    VNM_HIP(hipMemsetAsync(g->tag, 0xFF, g->stride * 8, s));
    VNM_HIP(hipMemsetAsync(g->ctl, 0, 64, s));
    for (int w = 0; w < g->n_words; w++) {
        const uint64_t init = merge_init(h->plan.merge[w]);
        if (init == 0 || init == ~0ULL) VNM_HIP(hipMemsetAsync(g->acc + (size_t)w * g->stride, init ? 0xFF : 0, g->stride * 8, s));
        else {   // (float sums start at -0.0: not a byte pattern)
            fill_u64_kernel<<<(int)std::min<uint64_t>((g->stride + 255) / 256, (uint64_t)device_info().num_cus * 8), 256, 0, s>>>(g->acc + (size_t)w * g->stride, init, (int64_t)g->stride);
            VNM_HIP(hipGetLastError());
        }
    }
    return 0;
}

void table_free(GTable* g) {
    pool_free(g->tag);
    pool_free(g->acc);
    pool_free(g->keyw);
    pool_free(g->ctl);
    memset(g, 0, sizeof(*g));
}

uint64_t pow2_at_least(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

int lds_slots_for(const AggPlan& p, int budget = AGG_LDS_BUDGET) {
    int per_slot = 8 * (1 + p.n_words);
    int s = 1;
    while ((s * 2 + 2) * per_slot <= budget) s *= 2;
    if (s > 8192) s = 8192;
    return s;
}

// grow the table by rehashing every occupied slot into a larger one
int table_grow(vnm_agg* h, uint64_t new_cap, hipStream_t s) {
    GTable old = h->g, ng;
    VNM_TRY(table_alloc(h, &ng, new_cap, s));
    MergeArgs m{};
    m.plan = h->plan;
    m.g = ng;
    m.n = (int64_t)old.stride;
    m.src_is_table = 1;
    m.src_stride = 1;
    m.src_tag = old.tag;
    m.src_cap = (int64_t)old.cap;
    for (int j = 0; j < old.kwt; j++) m.src_key[j] = old.keyw + (size_t)j * old.stride;
    for (int w = 0; w < old.n_words; w++) m.src_acc[w] = old.acc + (size_t)w * old.stride;
    int grid = device_info().num_cus * 8;
    agg_merge_kernel<<<grid, 256, 0, s>>>(m);
    VNM_HIP(hipGetLastError());
    VNM_HIP(hipStreamSynchronize(s));
    table_free(&old);
    h->g = ng;
    return 0;
}

// rows_only: the rows are what a partitioned pass spilled (the groups of the batch live in its run) -- the table is sized by them,
// not by the group count hint (a handful of spilled entries under a hint of 1e8 groups used to get a 2^28-slot table: its memset,
// and the walk over its slots at finish, cost 2-4 ms)
int ensure_table(vnm_agg* h, int64_t nrows, hipStream_t s, bool rows_only = false) {
    if (h->have_table) return 0;
    uint64_t cap;
    if (h->plan.n_keys == 0) cap = 2;
    else {
        uint64_t want = h->hint > 0 ? (uint64_t)h->hint * 2 : (uint64_t)1 << 22;
        if ((h->hint <= 0 || rows_only) && (uint64_t)nrows * 2 < want) want = (uint64_t)(nrows > 512 ? nrows : 512) * 2;
        // (spilled rows are mostly FEW keys -- a heavy key's 3e7 entries got a 2^26-slot table: its memset and the walks over its
        // slots at finish, run_patch_append_kernel 2.9 ms; the scan grows the table when the keys are many after all)
        if (rows_only && want > (1ULL << 20)) want = 1ULL << 20;
        cap = pow2_at_least(want < 1024 ? 1024 : want);
    }
    VNM_TRY(table_alloc(h, &h->g, cap, s));
    h->have_table = true;
    return 0;
}

void invalidate_result(vnm_agg* h) {
    if (!h->result_is_run) {
        pool_free(h->dkey);
        pool_free(h->dacc);
    }
    h->result_is_run = false;
    h->dkey = h->dacc = nullptr;
    h->n_groups = -1;
    h->host_ready = false;
}

// ---- partitioned path orchestration -------------------------------------------------------------------
int64_t env_i64(const char* name, int64_t dflt) {
    const char* v = getenv(name);
    return v ? (int64_t)atoll(v) : dflt;
}

void drop_run(vnm_agg* h) {
    pool_free(h->run_key);
    pool_free(h->run_acc);
    pool_free(h->run_dir);
    h->run_dir = nullptr;
    h->run_nfin = 0;
    h->run_key = h->run_acc = nullptr;
    h->have_run = false;
    h->run_n = h->run_stride = 0;
}

// distinct-key estimate from a strided sample (tiered: a small sample settles small G cheaply)
// tier0_lb (optional): when the small sample cannot settle G (most sampled keys distinct) return right there with
// *est = 0 and a LOWER bound of G from the uniform model -- the dense-key path only needs to know that G is large.
int estimate_groups(vnm_agg* h, const vnm_dcol& key, int64_t nrows, int64_t* est, hipStream_t s, int64_t* tier0_lb = nullptr) {
    const uint64_t* kp = (const uint64_t*)key.values + key.offset;
    int64_t sizes[2] = {std::min<int64_t>(nrows, 1 << 18), std::min<int64_t>(nrows, 1 << 24)};
    *est = 0;
    double heavy_share = 0.0;   // share of the rows held by heavy keys, and how many of those there are (from tier 0's sample)
    int64_t heavy_keys = 0;
    // tier 0: small sample into a scratch table (exact distinct count of the sample, ~50 us)
    {
        const int64_t m = sizes[0];
        GTable t{};
        t.cap = pow2_at_least((uint64_t)m * 2);
        t.stride = t.cap + 2;
        t.tag = (uint64_t*)pool_alloc(t.stride * 8);
        t.ctl = (unsigned long long*)pool_alloc(64);
        unsigned int* cnt = (unsigned int*)pool_alloc(t.cap * 4);
        if (!t.tag || !t.ctl || !cnt) return 1;
        VNM_HIP(hipMemsetAsync(t.tag, 0xFF, t.stride * 8, s));
        VNM_HIP(hipMemsetAsync(t.ctl, 0, 64, s));
        VNM_HIP(hipMemsetAsync(cnt, 0, t.cap * 4, s));
        int grid = (int)std::min<int64_t>((m + 255) / 256, (int64_t)device_info().num_cus * 8);
        agg_sample_kernel<<<grid, 256, 0, s>>>(kp, nrows, m, t, cnt, h->kn_valid, h->kn_off);
        agg_sample_heavy_kernel<<<grid, 256, 0, s>>>(cnt, (int64_t)t.cap, (unsigned int)std::max<int64_t>(64, m / 256), t.ctl);
        VNM_HIP(hipGetLastError());
        unsigned long long got[6] = {0, 0, 0, 0, 0, 0};
        VNM_HIP(hipMemcpyAsync(got, t.ctl, 48, hipMemcpyDeviceToHost, s));
        VNM_HIP(hipStreamSynchronize(s));
        pool_free(t.tag);
        pool_free(t.ctl);
        pool_free(cnt);
        unsigned long long d = got[2];
        if (d == 0) d = 1;
        // Heavy keys (each with > 0.4 % of the sample: NULLs, a default value) break the uniform model below -- half of the
        // rows in one key halves d / m and G came out 2.2x too small.  They are taken out: the model sees the remaining keys
        // over the remaining rows, and the heavy keys are added back as what they are, a handful of groups.
        h->heavy_share = got[5] ? (double)got[4] / (double)m : 0.0;   // (any key above 0.4 % of the sample: pass 2 of the dense path splits finer)
        if ((double)got[4] >= 0.02 * (double)m && got[5] < d) {
            heavy_share = std::min(0.999, (double)got[4] / (double)m);
            heavy_keys = (int64_t)got[5];
        }
        const double m0 = (double)m * (1.0 - heavy_share);
        const double d0 = (double)d - (double)heavy_keys;
        if (d0 / m0 < 0.125 || m == nrows) {  // the sample saw (nearly) every group
            *est = (int64_t)((double)d * (m == nrows ? 1.0 : 1.15)) + 1;
            return 0;
        }
        if (tier0_lb) {
            const double frac = std::min(0.97, d0 / m0);  // beyond 0.97 the sample has no resolution: G >= ~16 m
            double lo = 1e-9, hi = 64.0;
            for (int it = 0; it < 80; it++) {
                const double x = 0.5 * (lo + hi);
                if ((1.0 - exp(-x)) / x > frac) lo = x; else hi = x;
            }
            *tier0_lb = (int64_t)(m0 / (0.5 * (lo + hi)));
            return 0;
        }
    }
    // tier 1: HyperLogLog over a 16 M-key sample, then the uniform model d = G (1 - exp(-m / G)) solved for G
    {
        const int64_t m = sizes[1];
        unsigned int* regs = (unsigned int*)pool_alloc(HLL_M * 4);
        if (!regs) return 1;
        VNM_HIP(hipMemsetAsync(regs, 0, HLL_M * 4, s));
        int grid = (int)std::min<int64_t>((m + 1023) / 1024, (int64_t)device_info().num_cus);
        agg_hll_kernel<<<grid, 1024, 0, s>>>(kp, nrows, m, regs, h->kn_valid, h->kn_off);
        VNM_HIP(hipGetLastError());
        std::vector<unsigned int> hr(HLL_M);
        VNM_HIP(hipMemcpyAsync(hr.data(), regs, HLL_M * 4, hipMemcpyDeviceToHost, s));
        VNM_HIP(hipStreamSynchronize(s));
        pool_free(regs);
        double sum = 0;
        int zeros = 0;
        for (int i = 0; i < HLL_M; i++) { sum += ldexp(1.0, -(int)hr[i]); zeros += hr[i] == 0; }
        const double alpha = 0.7213 / (1.0 + 1.079 / HLL_M);
        double d = alpha * (double)HLL_M * (double)HLL_M / sum;
        if (d <= 2.5 * HLL_M && zeros) d = (double)HLL_M * log((double)HLL_M / zeros);  // linear counting
        if (d > (double)m) d = (double)m;
        d -= (double)heavy_keys;
        if (d < 1) d = 1;
        const double mm = (double)m * (1.0 - heavy_share);   // the rows of the sample that are not a heavy key's
        const double frac = std::min(1.0, d / mm);
        // solve d/m = (1 - exp(-x)) / x for x = m / G by bisection (x -> 1 / frac when the sample saw every group many times:
        // the upper bound must cover m / d, a bound of 64 turned G = 1e5 into an estimate of 3.1e5 and a second partition level)
        double lo = 1e-9, hi = 2.0 / frac + 64.0;
        for (int it = 0; it < 80; it++) {
            double x = 0.5 * (lo + hi);
            double f = (1.0 - exp(-x)) / x;
            if (f > frac) lo = x; else hi = x;
        }
        double G = mm / (0.5 * (lo + hi));
        if (frac > 0.97) G = mm * 30.0;  // beyond the resolution of the sample: at least this many
        G += (double)heavy_keys;
        if (G > (double)nrows) G = (double)nrows;
        *est = (int64_t)(G * 1.2) + 1;
    }
    return 0;
}

// Spilled wide entries [n][E] -> plain columns (key, the input columns at their own widths, validity bitmaps for the
// nullable ones), so that the general scan takes them like any batch.
struct UnzipArgs {
    const uint64_t* ent;
    int64_t n;
    int E, nval, has_vmask;
    uint64_t* key;
    void* vals[6];
    int widths[6];
    unsigned long long* valid[6];   // nullptr: the column had no bitmap
};
// The two spill lists of the dense path over a nullable value column -> columns: rows [0, n_ent) are the (key, value) entries,
// rows [n_ent, n_ent + n_null) the keys whose value is NULL; `valid` = the value column's validity bitmap (64 rows per word).
__global__ __launch_bounds__(256) void dense_vn_unzip_kernel(const ulonglong2* ent, int64_t n_ent, const uint64_t* nkeys, int64_t n_null,
                                                             uint64_t* key, uint64_t* val, unsigned long long* valid) {
    const int64_t n = n_ent + n_null;
    const int64_t n64 = (n + 63) & ~63LL;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n64; i += (int64_t)gridDim.x * 256) {
        bool ok = false;
        if (i < n_ent) { const ulonglong2 e = ent[i]; key[i] = e.x; val[i] = e.y; ok = true; }
        else if (i < n) { key[i] = nkeys[i - n_ent]; val[i] = 0; }
        const unsigned long long b = __ballot(ok);
        if ((threadIdx.x & 63) == 0) valid[i >> 6] = b;
    }
}

__global__ __launch_bounds__(256) void spill_unzip_kernel(UnzipArgs u) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t n64 = (u.n + 63) / 64 * 64;   // whole waves: the ballots below cover 64 consecutive entries
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n64; i += stride) {
        const bool in = i < u.n;
        const uint64_t* e = u.ent + (in ? i : u.n - 1) * u.E;
        if (in) u.key[i] = e[0];
        const uint64_t vmask = u.has_vmask ? e[u.E - 1] : ~0ULL;
        for (int c = 0; c < u.nval; c++) {
            const uint64_t raw = e[1 + c];
            if (in) {
                switch (u.widths[c]) {
                    case 1: ((uint8_t*)u.vals[c])[i] = (uint8_t)raw; break;
                    case 2: ((uint16_t*)u.vals[c])[i] = (uint16_t)raw; break;
                    case 4: ((uint32_t*)u.vals[c])[i] = (uint32_t)raw; break;
                    default: ((uint64_t*)u.vals[c])[i] = raw; break;
                }
            }
            if (u.valid[c]) {
                const unsigned long long b = __ballot(in && ((vmask >> c) & 1ULL));
                if ((threadIdx.x & 63) == 0) u.valid[c][i >> 6] = b;
            }
        }
    }
}

}  // namespace
static int merge_run_into_table(vnm_agg* h, hipStream_t s);
static int flush_scan_pending(vnm_agg* h, hipStream_t s);
#include "vnm_agg_routes.inc"

// fold a pending run into the HBM table (needed as soon as a second source of groups shows up)
static int merge_run_into_table(vnm_agg* h, hipStream_t s) {
    if (!h->have_run) return 0;
    uint64_t* kw[2] = {h->run_key, h->run_key + h->run_stride};
    uint64_t* aw[AGG_MAX_WORDS];
    for (int w = 0; w < h->plan.n_words; w++) aw[w] = h->run_acc + (size_t)w * h->run_stride;
    const int64_t n = h->run_n;
    h->have_run = false;  // vnm_agg_merge_device must not recurse into us
    int rc = vnm_agg_merge_device(h, n, kw, aw, (void*)s);
    if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = set_error("merge_run_into_table: stream sync failed");
    h->have_run = true;
    drop_run(h);
    return rc;
}


// The table of a stream of small-range batches (DScanPending) -> a run.  Whatever the handle holds as a run already goes to the HBM
// table first (one run at a time).
static int flush_scan_pending(vnm_agg* h, hipStream_t s) {
    DScanPending* sp = h->scan_pending;
    if (!sp) return 0;
    h->scan_pending = nullptr;
    std::unique_ptr<DScanPending> own(sp);
    if (h->pending) VNM_TRY(complete_pending(h, s, DF_RUN, nullptr, nullptr));
    if (h->have_run) VNM_TRY(merge_run_into_table(h, s));
    const int64_t dstride = sp->slots + 2;
    PoolScope pool;
    uint64_t* rk = (uint64_t*)pool.take((size_t)dstride * 8 * 2);
    uint64_t* ra = (uint64_t*)pool.take((size_t)dstride * 8 * std::max(1, h->plan.n_words));
    if (!rk || !ra) return 1;
    VNM_HIP(hipMemsetAsync(sp->flags, 0, 64, s));
    DFinalArgs df = sp->df;
    df.dkey = rk; df.dacc = ra; df.dstride = dstride; df.flags = sp->flags;
    dscan_emit_kernel<<<sp->slots / 512, 512, 0, s>>>(df, sp->slots, sp->t);
    VNM_HIP(hipGetLastError());
    unsigned long long fl[2];
    VNM_HIP(hipMemcpyAsync(fl, sp->flags, 16, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    if (fl[0]) return set_error("aggregate: the stream table holds more groups than slots (internal error)");
    pool.keep(rk); pool.keep(ra);
    h->run_key = rk; h->run_acc = ra; h->run_stride = dstride; h->run_n = (int64_t)fl[1];
    h->run_dir = nullptr; h->run_nfin = 0;
    h->have_run = true;
    return 0;
}

// see run_patch_kernel; *done = false leaves everything as it was (the caller merges the run into the table instead)
static int merge_table_into_run(vnm_agg* h, hipStream_t s, bool* done) {
    *done = false;
    if (!h->have_run || !h->have_table || !h->single || h->plan.n_keys != 1 || h->g.kwt != 0 || getenv("VNM_AGG_NO_RUN_PATCH") != nullptr) return 0;
    unsigned long long fill = 0;
    VNM_HIP(hipMemcpyAsync(&fill, h->g.ctl + 2, 8, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    const int64_t tmax = (int64_t)fill + 2;   // + the two special groups
    if (tmax * env_i64("VNM_RUN_PATCH_RATIO", 8) > h->run_n || h->run_n + tmax > h->run_stride - 2) return 0;
    PoolScope pool;
    const size_t fbytes = ((size_t)h->g.cap + 2 + 15) / 8 * 8;
    uint8_t* found = (uint8_t*)pool.take(fbytes + 8);
    if (!found) return 1;
    VNM_HIP(hipMemsetAsync(found, 0, fbytes + 8, s));
    PatchArgs a{};
    a.plan = h->plan; a.g = h->g;
    a.rkey = h->run_key; a.racc = h->run_acc; a.rn = h->run_n; a.rstride = h->run_stride;
    a.found = found;
    a.appended = (unsigned long long*)(found + fbytes);
    a.bloom = (uint32_t*)pool.take(PATCH_BLOOM_BITS / 8);
    if (!a.bloom) return 1;
    VNM_HIP(hipMemsetAsync(a.bloom, 0, PATCH_BLOOM_BITS / 8, s));
    const int cus = device_info().num_cus;
    run_patch_bloom_kernel<<<(int)std::min<int64_t>(((int64_t)h->g.cap + 255) / 256, (int64_t)cus * 8), 256, 0, s>>>(a);
    run_patch_kernel<<<(int)std::min<int64_t>((h->run_n + 1023) / 1024, (int64_t)cus * 2), 1024, 0, s>>>(a);
    run_patch_append_kernel<<<(int)std::min<int64_t>(((int64_t)h->g.cap + 2 + 255) / 256, (int64_t)cus * 8), 256, 0, s>>>(a);
    VNM_HIP(hipGetLastError());
    unsigned long long appended = 0;
    VNM_HIP(hipMemcpyAsync(&appended, a.appended, 8, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    if ((int64_t)appended > tmax) return set_error("aggregate: more table groups than the table's fill count (internal error)");
    h->run_n += (int64_t)appended;
    if (appended) {   // the partition directory no longer describes the whole run
        pool_free(h->run_dir);
        h->run_dir = nullptr;
        h->run_nfin = 0;
    }
    table_free(&h->g);
    h->have_table = false;
    *done = true;
    return 0;
}

#include "vnm_agg_fxn.inc"
#include "vnm_agg_multikey.inc"

#include "vnm_agg_exact.inc"

extern "C" {

vnm_agg* vnm_agg_create(int kind, int n_keys, const int* key_types, int n_funcs, const int* funcs,
                        const int* in_types, const int* in_flags, const int* in_col_ids) {
    int wide_type = -1;
    if (kind == VNM_SINGLE_NUMERICAL && n_keys == 1 && key_types && getenv("VNM_AGG_NO_WIDEN_KEYS") == nullptr) {
        const int t = key_types[0];
        if (t == VNM_I8 || t == VNM_I16 || t == VNM_I32) wide_type = VNM_I64;
        else if (t == VNM_U8 || t == VNM_U16 || t == VNM_U32) wide_type = VNM_U64;
    }
    // float32 input columns whose every function is a SUM / AVG / COUNT: declared float64 to the plan, widened on arrival
    std::vector<int> types2;
    bool widen_in[AGG_MAX_FUNCS] = {};
    bool any_widen = false;
    if (funcs && in_types && n_funcs > 0 && n_funcs <= AGG_MAX_FUNCS && getenv("VNM_AGG_NO_WIDEN_INPUTS") == nullptr) {
        types2.assign(in_types, in_types + n_funcs);
        for (int i = 0; i < n_funcs; i++) {
            if (in_types[i] != VNM_F32 || funcs[i] == VNM_COUNT_STAR) continue;
            bool ok = true;      // every function over the same column (same id; without ids: this function alone) must allow it
            bool sums = false;   // ... and one of them must be a SUM / AVG: a column that is only COUNTed is read for its validity alone
            for (int j = 0; j < n_funcs && ok; j++) {
                const bool same_col = in_col_ids ? (in_col_ids[j] >= 0 && in_col_ids[j] == in_col_ids[i]) : j == i;
                if (same_col && funcs[j] != VNM_SUM && funcs[j] != VNM_AVG && funcs[j] != VNM_COUNT) ok = false;
                if (same_col && in_types[j] != VNM_F32) ok = false;
                if (same_col && (funcs[j] == VNM_SUM || funcs[j] == VNM_AVG)) sums = true;
            }
            if (ok && sums) { widen_in[i] = true; types2[(size_t)i] = VNM_F64; any_widen = true; }
        }
    }
    vnm_agg* h = agg_create(kind, n_keys, wide_type >= 0 ? &wide_type : key_types, n_funcs, funcs, any_widen ? types2.data() : in_types, in_flags, in_col_ids);
    if (h && wide_type >= 0) h->key_out_type = key_types[0];
    if (h && any_widen) { memcpy(h->widen_in, widen_in, sizeof(widen_in)); h->any_widen_in = true; }
    if (h) exact_attach(h);
    return h;
}

namespace vnm {
// narrow integer key values -> 64-bit words, sign- or zero-extended; dst[i] belongs to row i (src index first + i)
__global__ __launch_bounds__(256) void widen_key_kernel(const void* src, int type, int64_t first, int64_t n, uint64_t* dst) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int64_t k = first + i;
        uint64_t v;
        switch (type) {
            case VNM_I8: v = (uint64_t)(int64_t)((const int8_t*)src)[k]; break;
            case VNM_I16: v = (uint64_t)(int64_t)((const int16_t*)src)[k]; break;
            case VNM_I32: v = (uint64_t)(int64_t)((const int32_t*)src)[k]; break;
            case VNM_U8: v = ((const uint8_t*)src)[k]; break;
            case VNM_U16: v = ((const uint16_t*)src)[k]; break;
            default: v = ((const uint32_t*)src)[k]; break;
        }
        dst[i] = v;
    }
}
// float32 values -> float64 (exact); dst[i] belongs to row i (src index first + i)
__global__ __launch_bounds__(256) void widen_f32_kernel(const float* src, int64_t first, int64_t n, double* dst) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = (double)src[first + i];
}
}  // namespace vnm

// the float32 input columns of this call that the plan took as float64 (vnm_agg_create): `win` = the inputs with those columns replaced
// by widened copies (one per distinct column); a predicate column that IS one of them follows, and the literal is rounded to float32 once
static int widen_inputs(vnm_agg* h, int64_t nrows, const vnm_dcol** inputs, vnm_dcol* win, const vnm_dcol** pred, vnm_dcol* wpred, hipStream_t s) {
    if (!h->any_widen_in || !*inputs) return 0;
    const vnm_dcol* in = *inputs;
    for (int i = 0; i < h->n_funcs; i++) win[i] = in[i];
    for (int i = 0; i < h->n_funcs; i++) {
        if (!h->widen_in[i]) continue;
        if (in[i].type != VNM_F32) return set_error("vnm_agg_next_device: input %d was declared float32, the batch brings type %d", i, in[i].type);
        int prev = -1;
        for (int j = 0; j < i && prev < 0; j++)
            if (h->widen_in[j] && in[j].values == in[i].values && in[j].offset == in[i].offset && in[j].validity == in[i].validity) prev = j;
        if (prev >= 0) { win[i] = win[prev]; continue; }
        win[i].type = VNM_F64;
        if (nrows > 0) {
            const int64_t off = in[i].validity ? (in[i].offset & 7) : 0;
            double* buf = (double*)pool_alloc((size_t)(off + nrows) * 8);
            if (!buf) return 1;
            h->widened.push_back({h->cur_seq, buf});
            if (off) VNM_HIP(hipMemsetAsync(buf, 0, (size_t)off * 8, s));
            const int grid = (int)std::min<int64_t>((nrows + 255) / 256, (int64_t)device_info().num_cus * 16);
            widen_f32_kernel<<<grid, 256, 0, s>>>((const float*)in[i].values, in[i].offset, nrows, buf + off);
            VNM_HIP(hipGetLastError());
            win[i].values = buf;
            win[i].validity = in[i].validity ? in[i].validity + (in[i].offset >> 3) : nullptr;
            win[i].offset = off;
            win[i].length = nrows;
        }
    }
    bool pred_widened = false;
    if (*pred && (*pred)->type == VNM_F32) {
        for (int i = 0; i < h->n_funcs; i++) {
            if (!h->widen_in[i] || in[i].values != (*pred)->values || in[i].offset != (*pred)->offset || in[i].validity != (*pred)->validity) continue;
            *wpred = win[i];
            *pred = wpred;
            pred_widened = true;
            break;
        }
    }
    // The EFFECTIVE literal of this batch: the caller's literal rounded to float32 when the predicate column is a float32 column
    // that travels widened (NumPy compares float32 against the literal in float32), the caller's own literal otherwise -- a later batch
    // whose predicate column is not widened gets it back -- and the suffix operator of an ordered MIN / MAX stream (fed the widened
    // column) compares with the same number as this handle.
    if (h->pred_set) {
        if (pred_widened) {
            const double lit = h->pred_user_is_float ? h->pred_user_dval : (double)h->pred_user_ival;
            h->pred_dval = (double)(float)lit;
            h->pred_is_float = 1;
            h->pred_lit_rounded = true;
        } else if (h->pred_lit_rounded) {
            h->pred_dval = h->pred_user_dval; h->pred_is_float = h->pred_user_is_float; h->pred_ival = h->pred_user_ival;
            h->pred_lit_rounded = false;
        }
        if (h->ex && h->ex->post) { h->ex->post->pred_dval = h->pred_dval; h->ex->post->pred_is_float = h->pred_is_float; h->ex->post->pred_ival = h->pred_ival; }
    }
    *inputs = win;
    return 0;
}

// the key column of this call as the operator's kernels want it: *keys stays, or becomes `wide` (values in a pool block the handle keeps
// until no recorded batch needs it)
static int widen_key(vnm_agg* h, int64_t nrows, const vnm_dcol** keys, vnm_dcol* wide, hipStream_t s) {
    if (h->key_out_type < 0 || !*keys) return 0;
    const vnm_dcol& k = (*keys)[0];
    if (k.type != h->key_out_type) return set_error("vnm_agg_next_device: key 0 changed type between batches");
    *wide = k;
    wide->type = h->plan.key_types[0];
    if (nrows > 0) {
        // the validity bitmap keeps its bit offset modulo 8 (the column's offset addresses values and validity alike)
        const int64_t off = k.validity ? (k.offset & 7) : 0;
        uint64_t* buf = (uint64_t*)pool_alloc((size_t)(off + nrows) * 8);
        if (!buf) return 1;
        h->widened.push_back({h->cur_seq, buf});
        if (off) VNM_HIP(hipMemsetAsync(buf, 0, (size_t)off * 8, s));
        const int grid = (int)std::min<int64_t>((nrows + 255) / 256, (int64_t)device_info().num_cus * 16);
        widen_key_kernel<<<grid, 256, 0, s>>>(k.values, k.type, k.offset, nrows, buf + off);
        VNM_HIP(hipGetLastError());
        wide->values = buf;
        wide->validity = k.validity ? k.validity + (k.offset >> 3) : nullptr;
        wide->offset = off;
        wide->length = nrows;
    }
    *keys = wide;
    return 0;
}
static void waiting_of(const vnm_agg* h, int64_t* batches, int64_t* rows, int64_t* oldest);
// widened key buffers no recorded batch needs any more go back to the pool -- behind an event on the stream whose kernels read them
// (pool_free_after: the pool is shared with other streams and threads; `synced`: the stream has just been synchronised)
static void release_widened(vnm_agg* h, void* stream, bool synced = false) {
    if (h->widened.empty()) return;
    int64_t b = 0, r = 0, o = -1;
    waiting_of(h, &b, &r, &o);
    size_t keep = 0;
    for (size_t i = 0; i < h->widened.size(); i++) {
        if (b > 0 && o >= 0 && h->widened[i].first >= o) h->widened[keep++] = h->widened[i];
        else if (synced) pool_free(h->widened[i].second);
        else pool_free_after(h->widened[i].second, as_stream(stream));
    }
    h->widened.resize(keep);
}

void vnm_agg_destroy(vnm_agg* h) {
    if (!h) return;
    if (h->ex) { exact_drop_result(h); exact_destroy(h->ex); h->ex = nullptr; }   // (a merged result's arrays go with their owner)
    if (h->inner) { vnm_agg_destroy(h->inner); h->inner = nullptr; }
    drop_parts(h);
    leave_tuple_mode(h);
    free_pack_tables(h);
    if (h->have_table) table_free(&h->g);
    invalidate_result(h);
    drop_run(h);
    delete h->pending;
    delete h->scan_pending;
    for (auto& w : h->widened) pool_free(w.second);
    delete h;
}

int vnm_agg_set_predicate(vnm_agg* h, int enabled, int op, int scalar_is_float, double dval, int64_t ival) {
    if (!h) return set_error("vnm_agg_set_predicate: null handle");
    if (enabled && (op < VNM_EQ || op > VNM_LE)) return set_error("vnm_agg_set_predicate: bad comparison op %d", op);
    // (a literal that was rounded to float32 for a widened float32 predicate column: the caller's own values are compared)
    const int cur_is_float = h->pred_lit_rounded ? h->pred_user_is_float : h->pred_is_float;
    const double cur_dval = h->pred_lit_rounded ? h->pred_user_dval : h->pred_dval;
    const int64_t cur_ival = h->pred_lit_rounded ? h->pred_user_ival : h->pred_ival;
    const bool same = h->pred_set == (enabled != 0) && (!enabled || (h->pred_op == op && cur_is_float == scalar_is_float && cur_ival == ival &&
                                                                     memcmp(&cur_dval, &dval, 8) == 0));
    if (same) return 0;
    // the waiting batches of an asynchronous stream were recorded under the OLD predicate (and without a predicate column when none was set)
    if (!h->q.empty()) return set_error("vnm_agg_set_predicate: batches are waiting (call vnm_agg_sync first)");
    for (vnm_agg* c : h->parts) if (!c->q.empty()) return set_error("vnm_agg_set_predicate: batches are waiting (call vnm_agg_sync first)");
    h->pred_set = enabled != 0;
    h->pred_op = op;
    h->pred_is_float = scalar_is_float;
    h->pred_dval = dval;
    h->pred_ival = ival;
    h->pred_lit_rounded = false;
    h->pred_user_is_float = scalar_is_float; h->pred_user_dval = dval; h->pred_user_ival = ival;
    if (h->ex && h->ex->post) VNM_TRY(vnm_agg_set_predicate(h->ex->post, enabled, op, scalar_is_float, dval, ival));
    return 0;
}

int vnm_agg_set_exchange_mode(vnm_agg* h, int rank_aligned) {
    if (!h) return set_error("vnm_agg_set_exchange_mode: null handle");
    h->rank_aligned = rank_aligned != 0;
    return 0;
}

int vnm_agg_set_hint(vnm_agg* h, int64_t expected_groups) {
    if (!h) return set_error("vnm_agg_set_hint: null handle");
    h->hint = expected_groups;
    if (h->ex && h->ex->post) h->ex->post->hint = expected_groups;
    return 0;
}

int vnm_agg_estimate_groups(vnm_agg* h, int64_t nrows, const vnm_dcol* key, int64_t* estimate, void* stream) {
    VNM_TRY(ensure_init());
    if (!h || !key || !estimate) return set_error("vnm_agg_estimate_groups: null argument");
    *estimate = 0;
    if (nrows <= 0 || type_width(key->type) != 8 || key->validity || (key->offset & 1)) return 0;   // the estimator reads plain 8-byte keys
    int64_t est = 0;
    VNM_TRY(estimate_groups(h, *key, nrows, &est, as_stream(stream)));
    *estimate = est;
    return 0;
}

// rc of next_device_impl when the waiting batches of a stream (h->segs_active) would have to take a path whose kernels read ONE batch:
// nothing has been aggregated, the caller sends the batches one by one
constexpr int VNM_RC_SINGLY = 77;
#define VNM_SEG_ONLY(what) do { if (h->segs_active || h->kn_valid) { if (getenv("VNM_AGG_TRACE")) fprintf(stderr, "[agg] %s: not this way (%s)\n", h->kn_valid ? "nullable key through the dense path" : "stream segments", what); return VNM_RC_SINGLY; } } while (0)

static bool queueable(const vnm_agg* h, int64_t nrows, const vnm_dcol* keys, const vnm_dcol* inputs, const vnm_dcol* pred, bool* pred_is_v, bool* multi);

static int next_device_impl(vnm_agg* h, int64_t nrows, const vnm_dcol* keys, const vnm_dcol* inputs,
                            const vnm_dcol* pred, void* stream) {
    VNM_TRY(ensure_init());
    if (!h) return set_error("vnm_agg_next_device: null handle");
    if (h->pred_set && !pred) return set_error("vnm_agg_next_device: predicate set but no predicate column given");
    hipStream_t s = as_stream(stream);
    invalidate_result(h);
    if (inputs && nrows > 0) for (int i = 0; i < h->n_funcs; i++) if (h->func_col[i] >= 0 && inputs[i].validity) h->null_inputs_seen = true;
    // A NULLABLE single 8-byte key under the hot program: the dense path takes it as it is -- pass 1 reads the key's validity and
    // sums the NULL-key rows up as the one group they are (single_numerical_hash_aggregate.cpp:24-32), everything after pass 1 never
    // sees a NULL.  Before round 4 such a key was packed into one word first (key range + pack + unpack passes, the NULL code a
    // heavy key of the inner operator, run + table merges at the end: 15.7 ms per 5e8 rows at G = 1e8 against 5.2 without NULLs).
    // The attempt runs this function again with the validity stripped from the key and kept aside; every path but the dense ring
    // scatter declines (VNM_RC_SINGLY, before any side effect) and the batch takes the packed route below.
    if (!h->kn_valid && !h->kn_failed && !h->segs_active && h->single && h->plan.n_keys == 1 && !h->inner && keys && inputs && keys[0].validity &&
        type_width(keys[0].type) == 8 && (keys[0].offset & 1) == 0 && nrows >= env_i64("VNM_AGG_ESTIMATE_MIN_ROWS", 1 << 22) &&
        getenv("VNM_AGG_NO_DENSE_KN") == nullptr) {
        vnm_dcol k2 = keys[0];
        k2.validity = nullptr;
        bool piv = false, multi = false;
        if (queueable(h, nrows, &k2, inputs, pred, &piv, &multi) && !multi) {   // (the hot shape: one input column)
            h->kn_valid = keys[0].validity; h->kn_off = keys[0].offset;
            const int rc = next_device_impl(h, nrows, &k2, inputs, pred, stream);
            h->kn_valid = nullptr;
            if (rc != VNM_RC_SINGLY) return rc;
            h->kn_failed = true;
        }
    }
    // rows the samplers may read through keys[0] (the first segment of a stream's waiting batches)
    const int64_t est_rows = h->segs_active ? (*h->segs_active)[0].nrows : nrows;
    if (nrows <= 0) {
        if (h->plan.n_keys == 0 || h->hint <= 0) VNM_TRY(ensure_table(h, nrows, s));
        return 0;
    }
    if (nrows >= (1LL << 31)) return set_error("vnm_agg_next_device: batches must be < 2^31 rows (as in the reference, agg_funcs.h:45)");

    // multi-column keys: try the packed single-word form first (see PackParams).  A SINGLE key that the fast paths do
    // not take as it is (int32 / int16 / float32 ..., NULLs, an odd Arrow offset) is packed the same way when the
    // group count is large or unknown: one extra pass over the key column (12 B/row) buys the partitioned path instead
    // of per-row HBM atomics (G = 1e6 int32 keys: 20x).
    const bool key_plain = h->plan.n_keys == 1 && type_width(keys[0].type) == 8 && !keys[0].validity && (keys[0].offset & 1) == 0;
    const bool pack_single = h->single && h->plan.n_keys == 1 && !key_plain &&
                             (h->hint > 2400 || (h->hint == 0 && nrows >= env_i64("VNM_AGG_ESTIMATE_MIN_ROWS", 1 << 22)));
    if ((!h->single && h->plan.n_keys >= 2) || pack_single || (h->single && h->inner)) {
        VNM_SEG_ONLY("packed keys");
        for (int j = 0; j < h->plan.n_keys; j++)
            if (keys[j].type != h->plan.key_types[j]) return set_error("vnm_agg_next_device: key %d changed type between batches", j);
        // ... only into an EMPTY handle: an operator that already holds groups of earlier batches in any form (HBM table, run,
        // deferred dense pass, stream table, the parts of a split program) keeps them -- a single key whose FIRST batch was plain
        // (no validity bitmap) and whose later batch brings NULL keys takes the general scan for that batch instead
        // (the inner operator's result used to replace, not join, what the handle held)
        const bool empty_handle = !h->have_table && !h->have_run && !h->pending && !h->scan_pending && h->parts.empty();
        if (!h->pack_tried && empty_handle && getenv("VNM_AGG_NO_PACK") == nullptr) {
            h->pack_tried = true;
            int err = 0;
            if (plan_packing(h, keys, nrows, s, &err)) {
                const int kt = VNM_U64;
                h->inner = agg_create(VNM_SINGLE_NUMERICAL, 1, &kt, h->n_funcs, h->c_funcs, h->c_in_types, h->c_in_flags,
                                          h->c_has_ids ? h->c_in_col_ids : nullptr);
                if (!h->inner) return 1;
                h->inner->hint = h->hint;
                // a nullable key column: its NULL code may be a heavy key of the inner operator, whose estimator is not asked when
                // the caller gave a group count (the dense path then cuts pass 2 into more work items, see heavy_share)
                for (int j = 0; j < h->plan.n_keys; j++) if (keys[j].validity) h->inner->heavy_share = std::max(h->inner->heavy_share, 0.01);
            } else if (err) return err;
            else if (!h->single && getenv("VNM_AGG_NO_TUPLE") == nullptr) {
                // too wide for one word even as per-column dictionary codes: tuple -> group id through a dictionary
                // (before: agg_wide_kernel, one HBM atomic per row and accumulator word)
                VNM_TRY(enter_tuple_mode(h, nrows, s));
            }
        }
        if (h->tuple_mode) { route_note("keys:tuple_dictionary", "%d key columns: tuple -> group id", h->plan.n_keys); return tuple_next(h, nrows, keys, inputs, pred, s); }
        if (h->inner) {
            {
                int ndict = 0;
                for (int j = 0; j < h->plan.n_keys; j++) ndict += h->pack.dtab[j] != nullptr;
                route_note(ndict ? "keys:packed_with_dictionary_fields" : "keys:packed", "%d key columns in one 64-bit word (%d dictionary-coded)", h->plan.n_keys, ndict);
            }
            for (int j = 0; j < h->plan.n_keys; j++) { h->pack.cols[j] = keys[j]; if (keys[j].validity) h->pack_null_seen = true; }
            uint64_t* packed = (uint64_t*)pool_alloc((size_t)nrows * 8);
            unsigned long long* flag = (unsigned long long*)pool_alloc(64);
            if (!packed || !flag) return 1;
            VNM_HIP(hipMemsetAsync(flag, 0, 8, s));
            {
                KernelTimer timer("agg_pack_keys", s);
                int grid = (int)std::min<int64_t>((nrows + 255) / 256, (int64_t)device_info().num_cus * 8);
                key_pack_kernel<<<grid, 256, 0, s>>>(h->pack, nrows, packed, flag);
            }
            unsigned long long bad = 0;
            VNM_HIP(hipMemcpyAsync(&bad, flag, 8, hipMemcpyDeviceToHost, s));
            VNM_HIP(hipStreamSynchronize(s));
            pool_free(flag);
            int rc = 0;
            if (!bad) {
                vnm_dcol pk{};
                pk.values = packed; pk.type = VNM_U64; pk.length = nrows;
                if (h->pred_set) rc = vnm_agg_set_predicate(h->inner, 1, h->pred_op, h->pred_is_float, h->pred_dval, h->pred_ival);
                if (!rc) { h->inner->child = true; h->inner->cur_seq = h->cur_seq; rc = vnm_agg_next_device(h->inner, nrows, &pk, inputs, pred, stream); }
                if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = set_error("aggregate: packed batch failed");
                pool_free(packed);
                if (!rc) h->rows_seen += nrows;
                return rc;
            }
            pool_free(packed);
            if (!h->single && getenv("VNM_AGG_NO_TUPLE") == nullptr) {   // keys outside the packed ranges: on through the tuple dictionary
                VNM_TRY(packed_to_tuple(h, nrows, s));
                route_note("keys:tuple_dictionary", "a batch outgrew the packed ranges");
                return tuple_next(h, nrows, keys, inputs, pred, s);
            }
            VNM_TRY(demote_packed(h, s));  // (a packed SINGLE key: its own general path; or the dictionary switched off: the wide-key table)
        }
    }

    // more input columns than a partition entry carries (or as many, plus a validity word) and many groups: split the program
    if (!h->parts.empty()) { VNM_SEG_ONLY("split program"); route_note("split_program:batch", "%zu parts", h->parts.size()); return next_parts(h, nrows, keys, inputs, pred, stream); }
    if (!h->split_tried && key_plain && h->single && keys[0].type == h->plan.key_types[0] && !h->have_table && !h->have_run && !h->pending &&
        !h->expr_active && getenv("VNM_AGG_NO_SPLIT") == nullptr) {
        bool any_null = false, cols8 = true;
        for (int c = 0; c < h->plan.n_cols; c++) {
            const vnm_dcol& col = inputs[h->col_first_func[c]];
            any_null = any_null || col.validity != nullptr;
            cols8 = cols8 && (col.type == VNM_F64 || col.type == VNM_I64 || col.type == VNM_U64);
        }
        const bool est_ok = nrows >= env_i64("VNM_AGG_ESTIMATE_MIN_ROWS", 1 << 22);
        // (a) more columns than an entry carries; (b) two or more 8-byte columns over a key the dense paths take (round 4)
        const bool many = h->plan.n_cols + (any_null ? 1 : 0) > env_i64("VNM_AGG_SPLIT_MIN_COLS", 6);
        const bool by_column = h->plan.n_cols >= 2 && cols8 && !h->rank_aligned && (keys[0].type == VNM_I64 || keys[0].type == VNM_U64) && est_ok &&
                               getenv("VNM_AGG_NO_SPLIT_SMALL") == nullptr;
        if (many || by_column) {
            h->split_tried = true;
            if (h->hint == 0 && !h->estimated && est_ok && getenv("VNM_AGG_NO_ESTIMATE") == nullptr) {
                int64_t est = 0;
                KernelTimer timer("agg_estimate", s);
                VNM_TRY(estimate_groups(h, keys[0], est_rows, &est, s));
                if (est) { h->hint = est; h->estimated = true; }
            }
            if (h->hint > env_i64("VNM_AGG_PART_MIN_GROUPS", std::min<int64_t>(2400, (int64_t)lds_slots_for(h->plan) * 6 / 10))) {
                if (by_column) {
                    if (h->dense_state == 0) {
                        KernelTimer timer("agg_estimate", s);
                        VNM_TRY(plan_dense(h, keys[0], est_rows, s));
                    }
                    // A few thousand groups in a SMALL key range: too many for the scan's hashed LDS table of this many words (flush
                    // storms), so the rows used to go through wide partition entries (C = 2 / 3 / 6 columns, G = 1000, 5e8 rows:
                    // 7.9 / 12.3 / 23.6 ms).  One part per column instead: each is the direct-addressed 2^13-slot LDS scan (16 bytes per
                    // row and column at the scan's rate: 3.3 / 4.9 / 9.8 ms incl. the join over a few thousand groups).
                    if (h->dense_state == 2 && h->hint <= (1 << DP_TBITS_MAX)) {
                        route_note("split_program:small_range_per_column", "%d columns, ~%lld groups in a range of <= 2^13 codes", h->plan.n_cols, (long long)h->hint);
                        VNM_TRY(make_parts(h, 1));
                        if (h->segs_active) return VNM_RC_SINGLY;   // (the waiting batches of a stream: one by one into the parts, which record them)
                        return next_parts(h, nrows, keys, inputs, pred, stream);
                    }
                    // MANY groups over a key the dense path takes, three or more columns: the dense path per column (16-byte entries,
                    // LDS-resident final tables) -- or per PAIR of float64 columns under sums and counts (two-value entries) where
                    // two scatter levels are needed anyway -- instead of wide entries through hash partitions; parts of the dense path
                    // over one code map are joined by units of 64 codes (collapse_parts: no sorts, no random gathers).
                    // 5e8 rows, 3 / 6 columns: G = 1e6 20.0 / 37.6 -> 13.9 / 26.8 ms, G = 1e8 30.1 / 360 -> 24.7 / 44.8.
                    if (nrows >= env_i64("VNM_AGG_SPLIT_DENSE_MIN_ROWS", 1 << 24) && h->dense_state == 1 &&
                        h->dense_span <= 32 * h->hint && h->dense_span <= env_i64("VNM_DENSE_SPAN_PER_ROW", 16) * nrows && getenv("VNM_AGG_NO_SPLIT_DENSE") == nullptr) {
                        bool sums = true;   // every function a sum / count of a plain float64 column: what the two-value entries carry
                        for (int i = 0; i < h->n_funcs && sums; i++) {
                            const int f = h->c_funcs[i];
                            sums = f == VNM_COUNT_STAR || ((f == VNM_SUM || f == VNM_AVG || f == VNM_COUNT) && inputs[i].type == VNM_F64 && !inputs[i].validity);
                        }
                        // (the two-value entries have no deferred final pass and take no stream segments: every batch ends in a run that is
                        // merged into the table -- 30 x 2^24 rows, three columns, G = 1e7: 107 ms against 17 for the same rows as one batch.
                        // They are for BIG batches; a stream's batches go per column, whose parts record and defer like any one-column stream)
                        const bool big = !h->segs_active && nrows >= env_i64("VNM_AGG_PAIRS_MIN_ROWS", (int64_t)1 << 27);
                        const bool pairs = sums && big && h->hint >= env_i64("VNM_AGG_SPLIT_PAIRS_MIN_GROUPS", 4000000);
                        // (TWO columns: the two-value entries themselves from ~1.5e6 groups on -- two scatter levels -- and one part per column
                        // below: 5e8 rows, G = 1e4 / 1e5 / 5e5 / 1e6: 8.2 / 10.9 / 14.8 / 10.1 -> 7.3 / 7.8 / 8.1 / 8.7 ms; 2e6: 9.9 against 11.4)
                        const bool split = h->plan.n_cols >= 3 || !sums || !big || (!pairs && h->hint <= env_i64("VNM_AGG_SPLIT_TWO_MAX_GROUPS", 1500000));
                        // two or three float64 columns whose values are fixed-point words: ONE pass over the rows (vnm_agg_fxn.inc) before the
                        // program is cut per column
                        if (sums && h->plan.n_cols <= 3 && h->parts.empty()) {
                            const int frc = dense_fxn_aggregate(h, nrows, keys, inputs, pred, s);
                            if (frc == 1) return 1;
                            if (frc == 0) { h->rows_seen += nrows; return 0; }
                        }
                        if (split) {
                            route_note(pairs ? "split_program:dense_per_pair" : "split_program:dense_per_column", "%d columns, ~%lld groups over %lld codes, %lld rows", h->plan.n_cols, (long long)h->hint, (long long)h->dense_span, (long long)nrows);
                            VNM_TRY(make_parts(h, pairs ? 2 : 1));
                            if (h->segs_active) return VNM_RC_SINGLY;
                            return next_parts(h, nrows, keys, inputs, pred, stream);
                        }
                    }
                }
                if (many) {
                    route_note("split_program:many_columns", "%d columns (+%d validity word), ~%lld groups", h->plan.n_cols, any_null ? 1 : 0, (long long)h->hint);
                    VNM_TRY(make_parts(h, (int)env_i64("VNM_AGG_SPLIT_COLS", any_null ? 5 : 6)));
                    if (h->segs_active) return VNM_RC_SINGLY;
                    return next_parts(h, nrows, keys, inputs, pred, stream);
                }
            } else if (many && cols8 && !any_null && h->hint > 0 && h->hint <= env_i64("VNM_AGG_SPLIT_FEW_MAX_GROUPS", 256) && !h->rank_aligned &&
                       getenv("VNM_AGG_NO_SPLIT_FEW") == nullptr) {
                // FEW groups under more than six plain 8-byte columns: parts of up to six columns, each through agg_hotn_kernel, instead of the
                // interpreted scan over all of them (5e8 rows, G = 7: 7 / 8 / 10 columns 11.5 / 12.9 / 15.7 ms at 2.8 TB/s)
                route_note("split_program:few_groups_many_columns", "%d columns, ~%lld groups", h->plan.n_cols, (long long)h->hint);
                VNM_TRY(make_parts(h, 6));
                if (h->segs_active) return VNM_RC_SINGLY;
                return next_parts(h, nrows, keys, inputs, pred, stream);
            }
        }
    }

    AggArgs a{};
    a.plan = h->plan;
    for (int j = 0; j < h->plan.n_keys; j++) {
        a.keys[j] = keys[j];
        if (keys[j].type != h->plan.key_types[j]) return set_error("vnm_agg_next_device: key %d changed type between batches", j);
    }
    for (int c = 0; c < h->plan.n_cols; c++) a.cols[c] = inputs[h->col_first_func[c]];
    if (h->pred_set) {
        a.pred = *pred;
        a.p = make_predicate(pred->type, pred->validity != nullptr, h->pred_op, h->pred_is_float, h->pred_dval, h->pred_ival);
    }
    a.nrows = nrows;
    a.ntiles = (nrows + AGG_TILE - 1) / AGG_TILE;
    a.debug = (int)env_i64("VNM_AGG_DEBUG", 0);
    if (h->expr_active) { a.has_expr = 1; a.expr = h->expr_dev; }
    const int cus = device_info().num_cus;

    if (h->plan.n_keys == 0) {
        VNM_SEG_ONLY("no GROUP BY");
        VNM_TRY(ensure_table(h, nrows, s));
        a.g = h->g;
        int grid = cus * 8;
        int64_t need = (nrows + OG_BLOCK - 1) / OG_BLOCK;
        if (grid > need) grid = (int)need;
        KernelTimer timer("agg_scan", s);
        // every kind at most once over at most one plain 8-byte column, float64 predicate or none: the register kernel
        bool og_hot = h->plan.n_cols <= 1 && getenv("VNM_AGG_NO_HOT") == nullptr;
        int vt = -1;
        if (og_hot && h->plan.n_cols == 1) {
            const vnm_dcol& c = a.cols[0];
            og_hot = (c.type == VNM_F64 || c.type == VNM_I64 || c.type == VNM_U64) && !c.validity && (c.offset & 1) == 0;
            vt = c.type;
        }
        for (int k = 0; k < 9; k++) a.hot_w[k] = -1;
        for (int o = 0; o < h->plan.n_ops && og_hot; o++) {
            const AccOp& op = h->plan.ops[o];
            if (op.kind < 0 || op.kind > A_MAX || a.hot_w[op.kind] >= 0 || (op.kind != A_COUNT_ROWS && vt < 0)) og_hot = false;
            else a.hot_w[op.kind] = op.word;
        }
        int pm = 0;
        if (og_hot && h->pred_set) {
            og_hot = a.pred.type == VNM_F64 && !a.pred.validity && (a.pred.offset & 1) == 0 && a.p.mode == CMP_F64;
            pm = og_hot && vt == VNM_F64 && a.pred.values == a.cols[0].values && a.pred.offset == a.cols[0].offset ? 1 : 2;
        }
        route_note(og_hot ? "onegroup:register_scan" : "onegroup:lds_scan", "%lld rows, %d columns, %d words", (long long)nrows, h->plan.n_cols, h->plan.n_words);
        if (og_hot) {
            const int g2 = (int)std::min<int64_t>((int64_t)cus * 8, std::max<int64_t>(1, (nrows / 2 + OG_BLOCK - 1) / OG_BLOCK));
#define VNM_OG(VT_)                                                                                                 \
    do {                                                                                                            \
        if (pm == 0) agg_onegroup_hot_kernel<VT_, 0><<<g2, OG_BLOCK, 0, s>>>(a);                                    \
        else if (pm == 1) agg_onegroup_hot_kernel<VT_, 1><<<g2, OG_BLOCK, 0, s>>>(a);                               \
        else agg_onegroup_hot_kernel<VT_, 2><<<g2, OG_BLOCK, 0, s>>>(a);                                            \
    } while (0)
            if (vt == VNM_F64) VNM_OG(VNM_F64); else if (vt == VNM_I64) VNM_OG(VNM_I64); else if (vt == VNM_U64) VNM_OG(VNM_U64); else VNM_OG(-1);
#undef VNM_OG
        } else
        agg_onegroup_kernel<<<grid, OG_BLOCK, (size_t)h->plan.n_words * OG_BLOCK * 8, s>>>(a);
        VNM_HIP(hipGetLastError());
        h->rows_seen += nrows;
        return 0;
    }

    const int S = lds_slots_for(h->plan);
    a.lds_slots = S;
    // hot shape: one 8-byte key, every function in {COUNT(*), COUNT, SUM, AVG} over ONE float64 column,
    // float64 predicate column (or none), no validity bitmaps, even offsets (16-byte aligned pairs)
    // hot_scan: what agg_hot_kernel takes (any accumulator kind over at most TWO 8-byte columns without NULLs);
    // hot: the subset {COUNT(*), COUNT, SUM, AVG} of a float64 column that part_agg_kernel is specialised for
    bool hot_scan = h->single && h->plan.n_cols <= 2 && type_width(keys[0].type) == 8 && !keys[0].validity &&
                    (keys[0].offset & 1) == 0 && getenv("VNM_AGG_NO_HOT") == nullptr;
    a.hot_has_val = h->plan.n_cols >= 1;
    a.hot_vtype = a.hot_vtype2 = VNM_U64;
    bool hot_vnull = false;  // ONE nullable input column: agg_hot_kernel<VNULL>
    for (int c = 0; c < h->plan.n_cols && hot_scan; c++) {
        const vnm_dcol& col = a.cols[c];
        if (col.validity && h->plan.n_cols == 1) hot_vnull = true;
        hot_scan = (col.type == VNM_F64 || col.type == VNM_I64 || col.type == VNM_U64) && (!col.validity || hot_vnull) && (col.offset & 1) == 0;
        (c == 0 ? a.hot_vtype : a.hot_vtype2) = col.type;
    }
    const bool hot_two = hot_scan && h->plan.n_cols == 2;
    a.hot_w_rows = a.hot_w_valid = a.hot_w_sum = -1;
    a.hot_comp = 0;
    for (int w = 0; w < h->plan.n_words; w++) if (h->plan.merge[w] == M_ADD_F64C) a.hot_comp = 1;
    for (int k = 0; k < 9; k++) a.hot_w[k] = a.hot_w2[k] = -1;
    bool hot = hot_scan && h->plan.n_cols == 1 && a.hot_vtype == VNM_F64;   // (a nullable column: hot_prog below)
    if (hot_scan) {
        for (int o = 0; o < h->plan.n_ops; o++) {
            const AccOp& op = h->plan.ops[o];
            int* hw = op.col == 1 ? a.hot_w2 : a.hot_w;
            const int vt = op.col == 1 ? a.hot_vtype2 : a.hot_vtype;
            if (op.kind < 0 || op.kind > A_MAX || hw[op.kind] >= 0) { hot_scan = false; break; }  // one word per kind and column
            if (op.kind == A_SUM_F64 && vt != VNM_F64) { hot_scan = false; break; }
            hw[op.kind] = op.word;
            if (op.col == 1) { hot = false; continue; }
            if (op.kind == A_COUNT_ROWS) a.hot_w_rows = op.word;
            else if (op.kind == A_COUNT_VALID) a.hot_w_valid = op.word;
            else if (op.kind == A_SUM_F64) a.hot_w_sum = op.word;
            else hot = false;
        }
    }
    a.hot_wpack = a.hot_wpack2 = ~0ULL;
    for (int k = A_COUNT_VALID; k <= A_MAX; k++) {
        if (a.hot_w[k] >= 0) a.hot_wpack = (a.hot_wpack & ~(63ULL << (6 * k))) | ((unsigned long long)a.hot_w[k] << (6 * k));
        if (a.hot_w2[k] >= 0) a.hot_wpack2 = (a.hot_wpack2 & ~(63ULL << (6 * k))) | ((unsigned long long)a.hot_w2[k] << (6 * k));
    }
    hot = hot && hot_scan;
    if (hot_scan && h->pred_set) {
        a.hot_pred_is_v = a.hot_has_val && a.hot_vtype == VNM_F64 && a.pred.values == a.cols[0].values && a.pred.offset == a.cols[0].offset;
        // a nullable predicate column only as the (nullable) input column itself
        hot_scan = a.pred.type == VNM_F64 && (!a.pred.validity || (hot_vnull && a.hot_pred_is_v)) && (a.pred.offset & 1) == 0 && a.p.mode == CMP_F64;
        hot = hot && hot_scan;
    }
    // the hot PROGRAM over a nullable column filtered by itself (`WHERE v > x`: a NULL fails the filter): the dense path drops the
    // NULL rows in pass 1 and everything after it is the hot shape (vn_fold below); every other kernel of the hot shape reads no bitmap
    const bool hot_prog = hot;
    if (hot_vnull) hot = false;
    // three to six plain float64 columns under {COUNT(*), COUNT, SUM, AVG}: agg_hotn_kernel takes the scan
    bool hotn = !hot_scan && h->single && h->plan.n_cols >= 3 && h->plan.n_cols <= 6 && type_width(keys[0].type) == 8 && !keys[0].validity &&
                (keys[0].offset & 1) == 0 && !a.has_expr && getenv("VNM_AGG_NO_HOTN") == nullptr && getenv("VNM_AGG_NO_HOT") == nullptr;
    {
        int w_rows = -1, w_cnt[6] = {-1, -1, -1, -1, -1, -1};
        for (int c = 0; c < 6; c++) a.hn_w_sum[c] = -1;
        a.hn_any_int = a.hn_any_mm = 0;
        for (int c = 0; c < 6; c++) { a.hn_ctype[c] = VNM_F64; a.hn_wmm[c][0] = a.hn_wmm[c][1] = -1; for (int k = 0; k < 4; k++) a.hn_iw[c][k] = -1; }
        for (int c = 0; c < h->plan.n_cols && hotn; c++) {
            const vnm_dcol& col = a.cols[c];
            hotn = (col.type == VNM_F64 || col.type == VNM_I64 || col.type == VNM_U64) && !col.validity && (col.offset & 1) == 0 && ((uintptr_t)col.values & 15) == 0;
            a.hn_ctype[c] = col.type;
            if (col.type != VNM_F64) a.hn_any_int = 1;   // (not the PLAIN instantiation: its values are float64 bit patterns)
        }
        for (int o = 0; o < h->plan.n_ops && hotn; o++) {
            const AccOp& op = h->plan.ops[o];
            if (op.kind == A_COUNT_ROWS && w_rows < 0) w_rows = op.word;
            else if (op.kind == A_COUNT_VALID && w_cnt[op.col] < 0) w_cnt[op.col] = op.word;
            else if (op.kind == A_SUM_F64 && a.hn_w_sum[op.col] < 0) a.hn_w_sum[op.col] = op.word;
            else if (op.kind >= A_SUM_I64 && op.kind <= A_SUM_HI32U && a.hn_ctype[op.col] != VNM_F64 && a.hn_iw[op.col][op.kind - A_SUM_I64] < 0) {
                a.hn_iw[op.col][op.kind - A_SUM_I64] = op.word;
                a.hn_any_int = 1;
            } else if ((op.kind == A_MIN || op.kind == A_MAX) && a.hn_wmm[op.col][op.kind - A_MIN] < 0) {
                a.hn_wmm[op.col][op.kind - A_MIN] = op.word;
                a.hn_any_mm = 1;
            } else hotn = false;
        }
        a.hn_w_base = w_rows;
        a.hn_n_copy = 0;
        for (int c = 0; c < h->plan.n_cols && hotn; c++) {
            if (w_cnt[c] < 0) continue;
            if (a.hn_w_base < 0) a.hn_w_base = w_cnt[c];
            else a.hn_w_copy[a.hn_n_copy++] = w_cnt[c];
        }
        hotn = hotn && a.hn_w_base >= 0 && ((uintptr_t)keys[0].values & 15) == 0;
        a.hn_pred_col = -1;
        if (hotn && h->pred_set) {
            hotn = a.pred.type == VNM_F64 && !a.pred.validity && (a.pred.offset & 1) == 0 && ((uintptr_t)a.pred.values & 15) == 0 && a.p.mode == CMP_F64;
            for (int c = 0; c < h->plan.n_cols && hotn; c++)
                if (a.pred.values == a.cols[c].values && a.pred.offset == a.cols[c].offset) a.hn_pred_col = c;
        }
    }
    // the partitioned path also takes ANY accumulator program over at most one 8-byte input column: its entries
    // carry (key, raw value bits) and only the final pass interprets them
    bool part_ok = hot;
    bool narrow_generic = false;   // a generic program over (key, one plain 8-byte value or none): the dgen_* dense kernels take it too
    if (!hot && h->single && h->plan.n_cols <= 6 && type_width(keys[0].type) == 8 && !keys[0].validity &&
        (keys[0].offset & 1) == 0 && getenv("VNM_AGG_NO_PART_GENERIC") == nullptr &&
        (size_t)(256 + 1) * 8 * (1 + h->plan.n_words) <= 150 * 1024) {   // (the final pass shrinks its LDS table to fit)
        // narrow entries (key, value): at most one 8-byte input column without NULLs, float64 predicate column
        bool narrow = h->plan.n_cols <= 1;
        bool any_null = false;
        for (int c = 0; c < h->plan.n_cols; c++) {
            const vnm_dcol& col = a.cols[c];
            a.part_vtypes[c] = col.type;
            if (col.validity) any_null = true;
            if (!((col.type == VNM_I64 || col.type == VNM_U64 || col.type == VNM_F64) && !col.validity && (col.offset & 1) == 0)) narrow = false;
        }
        if (h->plan.n_cols >= 1) a.part_vtype = a.cols[0].type;
        if (narrow && h->pred_set) {
            narrow = a.pred.type == VNM_F64 && !a.pred.validity && (a.pred.offset & 1) == 0 && a.p.mode == CMP_F64;
            a.hot_pred_is_v = h->plan.n_cols == 1 && a.pred.values == a.cols[0].values && a.pred.offset == a.cols[0].offset;
        }
        // wide entries take the rest: several columns, NULLs, narrow types, any predicate -- while key + values
        // (+ validity word) fit seven words
        const bool wide = !narrow && h->plan.n_cols >= 1 && h->plan.n_cols + (any_null ? 1 : 0) <= 6 &&
                          getenv("VNM_AGG_NO_PART_WIDE") == nullptr;
        part_ok = narrow || wide;
        narrow_generic = narrow;
        a.part_generic = part_ok;
        a.part_wide = wide;
        a.part_vmask = wide && any_null;
        // no input column at all (COUNT(*) only): the entries are the keys alone -- 8 bytes instead of 16 through every
        // pass, via the wide kernels with E = 1.  Those cannot spill heavy keys; if a region overflows the operator goes
        // back to (key, unused word) entries with the spill buffer for good.
        if (narrow && h->plan.n_cols == 0 && !h->key_only_failed && getenv("VNM_AGG_NO_KEY_ONLY") == nullptr) a.part_wide = 1;
    }
    // no hint from the caller: estimate the group count once from a sample of the first large batch
    // Dense-key path (vnm_agg_dense.inc): the north-star shape over an int64 / uint64 key whose (sampled) range fits 29
    // bits.  It needs to know that G is LARGE, not how large: when the small sample of the estimator cannot settle G, its
    // lower bound is enough (span <= 32 G: the direct-addressed slots are reasonably filled) and the HyperLogLog pass
    // (0.85 ms) is skipped; the hash-partitioned path below still estimates properly if the dense attempt fails.
    // (generic programs: only where the spilled entries -- keys outside the sampled range -- have a kernel to go to)
    // ... and ONE nullable 8-byte input column (hot_vnull; the predicate column, if any, without NULLs or that column itself): the
    // entries carry a NULL flag next to the code (vnm_agg_dense.inc, VN kernels); what the dense pass cannot place comes back as
    // two lists (entries, keys of NULL-value rows) and goes through the scan below as columns
    const bool dense_vn = !hot && hot_scan && hot_vnull && !hot_two && !a.has_expr && part_ok && h->plan.n_cols == 1 &&
                          getenv("VNM_AGG_NO_DENSE_VN") == nullptr;
    const bool vn_fold = dense_vn && hot_prog && h->pred_set && a.hot_pred_is_v && getenv("VNM_AGG_NO_VN_FOLD") == nullptr;
    const bool dense_generic = ((!hot && narrow_generic && hot_scan && !hot_two && !hot_vnull && !a.has_expr) || dense_vn) &&
                               getenv("VNM_AGG_NO_DENSE_GENERIC") == nullptr && getenv("VNM_AGG_NO_SPILL") == nullptr;
    // ... and TWO plain float64 input columns under {COUNT(*), COUNT, SUM, AVG} (dense_two_aggregate: two-value entries)
    bool dense_two = hot_scan && hot_two && !a.has_expr && !h->rank_aligned && !h->segs_active && !h->kn_valid && a.hot_vtype == VNM_F64 && a.hot_vtype2 == VNM_F64 &&
                     getenv("VNM_AGG_NO_DENSE_TWO") == nullptr;
    for (int o = 0; o < h->plan.n_ops && dense_two; o++) {
        const int k = h->plan.ops[o].kind;
        dense_two = k == A_COUNT_ROWS || k == A_COUNT_VALID || k == A_SUM_F64;
    }
    const bool dense_base = (hot || dense_generic || dense_two) && part_ok && nrows >= env_i64("VNM_AGG_ESTIMATE_MIN_ROWS", 1 << 22) &&
                            getenv("VNM_AGG_NO_DENSE") == nullptr;
    bool dense_shape = dense_base && (!h->rank_aligned || h->range_given);   // rank-aligned: only with a code range all ranks agreed on
    // a batch must bring enough rows for its code range (a final pass over 2^b slots for a handful of rows is all overhead) -- unless a
    // deferred pass over that very range is waiting anyway: a short batch of a stream then simply joins it (round 4; before, it took
    // two hash levels -- and, with a nullable key, the general scan)
    // (evaluated where it is used: plan_dense may only just have set the range)
    // (16 codes per row of the batch: a 2^24-row batch of a stream opens -- and later batches join -- a deferred pass over 2^27 codes; with 4
    // such a stream took the hash partitions batch by batch: 59 x 2^24 rows, G = 1e8, synchronous next(): 132.6 -> 21.6 ms; a single
    // 2^24-row batch over that range: 1.66 -> 1.10 ms)
    const int64_t span_per_row = env_i64("VNM_DENSE_SPAN_PER_ROW", 16);
    auto span_fits_now = [&]() { return h->dense_span <= span_per_row * nrows || (h->pending != nullptr && h->dense_state == 1 && memcmp(&h->pending->df.map, &h->dmap, sizeof(DenseMap)) == 0); };
#define span_fits span_fits_now()
    bool dense_go = false;
    // a stream that went dense on the sample's lower bound (no group count exists) and now brings a batch too short for the
    // code range: this batch and the rest need a number after all (without one they took the LDS scan and its flush storms)
    if (h->dense_by_bound && h->hint == 0 && !(dense_shape && h->dense_state == 1 && span_fits)) {
        h->estimated = false;
        h->dense_by_bound = false;
    }
    if (part_ok && h->hint == 0 && !h->estimated && nrows >= env_i64("VNM_AGG_ESTIMATE_MIN_ROWS", 1 << 22) &&
        getenv("VNM_AGG_NO_ESTIMATE") == nullptr) {
        int64_t est = 0, dense_lb = 0;
        KernelTimer timer("agg_estimate", s);
        // the small sample first: a conclusive (small) group count needs neither the key range nor HyperLogLog
        VNM_TRY(estimate_groups(h, keys[0], est_rows, &est, s, dense_shape ? &dense_lb : nullptr));
        if (est == 0) {
            if (h->dense_state == 0) VNM_TRY(plan_dense(h, keys[0], est_rows, s));
            // (... or, under skew -- duplicates in the sample pull the bound down -- when the bound at least rules out the LDS scans and the
            // rows outnumber the code range four to one: the passes over the rows dominate whatever G is, the hash partitions would
            // cost the same, and the HyperLogLog pass (0.8 ms) buys nothing)
            const bool rows_dominate = dense_lb >= env_i64("VNM_DENSE_SKEW_MIN_LB", 100000) && h->dense_span * 4 <= nrows;
            if (h->dense_state == 1 && (h->dense_span <= 32 * dense_lb || rows_dominate) && span_fits) dense_go = true;
            else VNM_TRY(estimate_groups(h, keys[0], est_rows, &est, s));   // too sparse (or not a code-able key): full estimate
        }
        if (est) { h->hint = est; h->estimated = true; }
        else if (dense_go) { h->estimated = true; h->dense_by_bound = true; }   // later batches of the stream: no sample again
    }
    if (dense_shape && !dense_go && h->dense_by_bound && h->hint == 0 && h->dense_state == 1 && span_fits) dense_go = true;
    // rank-aligned operators (multi-GPU) keep hash partitions so that every rank cuts its result the same way -- which only
    // the partition-aligned exchange of LARGE results needs; small results travel by one all-gather and are merged by key
    if (dense_base && h->rank_aligned && h->hint > 0 && h->hint <= env_i64("VNM_ALIGNED_DENSE_MAX", 1 << 19)) dense_shape = true;
    const int64_t part_min0 = env_i64("VNM_AGG_PART_MIN_GROUPS", std::min<int64_t>(2400, (int64_t)S * 6 / 10));
    if (dense_shape && !dense_go && h->hint > part_min0) {
        if (h->dense_state == 0) {
            KernelTimer timer("agg_estimate", s);
            VNM_TRY(plan_dense(h, keys[0], est_rows, s));
        }
        if (h->dense_state == 1 && h->dense_span <= 32 * h->hint && span_fits) dense_go = true;
    }
    ulonglong2* spill = nullptr;  // entries the partitioned / dense paths could not place (heavy keys, keys outside the sampled range): aggregated below
    int64_t n_spill = 0;
    uint64_t* nspill = nullptr;   // dense path over a nullable value column: keys of the NULL-value rows it could not place
    int64_t n_nspill = 0;
    bool scan_no_pred = false;    // ... whose rows come back as columns with the predicate already applied
    unsigned int* progress = nullptr;
    PoolScope unzip;   // spilled wide entries as columns
    PoolSlotGuard<ulonglong2> spill_guard(&spill);       // both go back to the pool on every way out of this function
    PoolSlotGuard<uint64_t> nspill_guard(&nspill);
    PoolSlotGuard<unsigned int> progress_guard(&progress);
    // the two spill lists of the nullable dense path -> key / value / validity columns for the scan below
    auto vn_spill_to_columns = [&]() -> int {
        const int64_t n = n_spill + n_nspill;
        uint64_t* uk = (uint64_t*)unzip.take((size_t)n * 8);
        uint64_t* uv = (uint64_t*)unzip.take((size_t)n * 8);
        unsigned long long* ub = (unsigned long long*)unzip.take((size_t)((n + 63) / 64) * 8);
        if (!uk || !uv || !ub) return 1;
        const int ugrid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)device_info().num_cus * 8);
        dense_vn_unzip_kernel<<<ugrid, 256, 0, s>>>(spill, n_spill, nspill, n_nspill, uk, uv, ub);
        VNM_HIP(hipGetLastError());
        a.keys[0].values = uk; a.keys[0].validity = nullptr; a.keys[0].offset = 0; a.keys[0].length = n;
        a.cols[0].values = uv; a.cols[0].validity = (const uint8_t*)ub; a.cols[0].offset = 0; a.cols[0].length = n;
        a.ent = nullptr;
        a.nrows = n;
        a.p.enabled = 0;
        scan_no_pred = true;
        return 0;
    };
    // a few hundred to a few thousand groups in a small key range: the direct-addressed LDS scan (vnm_agg_dense.inc)
    bool dscan_done = false;
    bool dscan_stream = false;   // the batch went into the stream table (h->scan_pending)
    if (dense_shape && !dense_go && !dense_two && h->hint >= env_i64("VNM_DSCAN_MIN_GROUPS", 128) && h->hint <= (1 << DP_TBITS_MAX) &&
        getenv("VNM_AGG_NO_DSCAN") == nullptr) {
        if (h->dense_state == 0) {
            KernelTimer timer("agg_estimate", s);
            VNM_TRY(plan_dense(h, keys[0], est_rows, s));
        }
        if (h->dense_state == 2) {
            if (h->kn_valid || (h->segs_active && (dense_generic || a.has_expr))) VNM_SEG_ONLY("small-range LDS scan");   // (the hot program's scan takes segments)
            if (h->pending) VNM_TRY(complete_pending(h, s));   // (this path makes a run of its own)
            if (h->have_run) VNM_TRY(merge_run_into_table(h, s));
            if (dense_generic && h->scan_pending) VNM_TRY(flush_scan_pending(h, s));
            route_note(dense_generic ? "dense_scan:generic" : "dense_scan:hot", "~%lld groups in a sampled range of %lld codes (<= 2^13): direct-addressed LDS table, no scatter", (long long)h->hint, (long long)h->dense_span);
            int prc = dense_scan_aggregate(h, a, nrows, s, &spill, &n_spill, dense_generic, &nspill, &n_nspill);
            dscan_stream = prc == 0 && !dense_generic;
            // a generic program whose table for this range does not fit LDS: the same 2^13 codes through one scatter level
            // (four partitions of 2^11 slots, split final pass) instead
            if (prc == 2 && dense_generic && h->dense_state == 2) prc = dense_partitioned_aggregate(h, a, nrows, s, &spill, &n_spill, true, &nspill, &n_nspill);
            if (prc == 1) return 1;
            if (prc == 0 && !spill && !nspill) { h->rows_seen += nrows; return 0; }
            if (prc == 0 && dense_vn) {
                VNM_TRY(vn_spill_to_columns());
                dscan_done = true;
            } else if (prc == 0) {  // the scan below runs over the spilled entries only (predicate already applied)
                a.ent = spill;
                a.nrows = n_spill;
                a.p.enabled = 0;
                dscan_done = true;
            }
        }
    }
    // (a batch that went another way than the stream table of the small-range scan: that table becomes a run first)
    if (h->scan_pending && !dscan_stream) VNM_TRY(flush_scan_pending(h, s));
    // many groups: radix-partitioned path (no per-row HBM atomics); falls through when it does not apply
    // ... from the point where the groups stop fitting the LDS table of the scan kernel (flush storms otherwise:
    // MIN+MAX with 2000 groups and a 2048-slot table ran at 38 ms)
    const int64_t part_min = env_i64("VNM_AGG_PART_MIN_GROUPS", std::min<int64_t>(2400, (int64_t)S * 6 / 10));
    if (!dscan_done && part_ok && (h->hint > part_min || dense_go) && getenv("VNM_AGG_NO_PART") == nullptr) {
        if (h->have_run) VNM_TRY(merge_run_into_table(h, s));
        // wide entries (any program over up to six columns, key-only entries) spill as whole entries; they come back as
        // plain columns for the general scan below (spill_unzip_kernel)
        const bool spill_wide = a.part_wide && h->single && !a.has_expr && getenv("VNM_AGG_NO_SPILL") == nullptr && getenv("VNM_AGG_NO_WIDE_SPILL") == nullptr;
        const bool can_spill = (hot_scan && !hot_two && !hot_vnull && getenv("VNM_AGG_NO_SPILL") == nullptr) || spill_wide;
        int prc = 2;
        bool vn_spill = false;        // the spill lists of the nullable dense path (entries + keys of NULL-value rows)
        bool spill_is_wide = false;   // the spill holds [n][E]-word entries of the wide scatter kernels (not the (key, value) pairs of the hot / dense paths)
        auto run_partitioned = [&]() {
            if (h->segs_active || h->kn_valid) return VNM_RC_SINGLY;   // (its kernels read one batch of plain keys)
            if (h->pending && complete_pending(h, s)) return 1;     // (the hash-partitioned path makes a run of its own)
            if (h->have_run && merge_run_into_table(h, s)) return 1;
            route_note(a.part_wide ? "hash_partitions:wide_entries" : (a.part_generic ? "hash_partitions:generic" : "hash_partitions:hot"), "~%lld groups > %lld (the scan's LDS table), dense path %s", (long long)h->hint, (long long)part_min,
                       h->dense_state == 1 ? "declined or failed" : "not applicable");
            const int r = partitioned_aggregate(h, a, nrows, s, can_spill ? &spill : nullptr, can_spill ? &n_spill : nullptr);
            spill_is_wide = r == 0 && spill != nullptr && a.part_wide != 0;
            return r;
        };
        if (dense_go && dense_two && (h->hint > part_min || h->hint == 0)) {
            if (h->pending && complete_pending(h, s)) return 1;     // (this path makes a run of its own)
            if (h->have_run && merge_run_into_table(h, s)) return 1;
            route_note("dense:two_values", "%lld codes, ~%lld groups, %lld rows: (value, value, code) entries", (long long)h->dense_span, (long long)h->hint, (long long)nrows);
            prc = dense_two_aggregate(h, a, nrows, s);
            if (prc == 2) h->dense_state = -1;                      // (a key outside the range, a full region, a range it does not take: not again)
            if (prc == 2 && h->hint == 0) {
                int64_t est = 0;
                VNM_TRY(estimate_groups(h, keys[0], est_rows, &est, s));
                h->hint = est;
                h->estimated = true;
            }
        } else if (dense_go && (h->hint > part_min || h->hint == 0)) {
            prc = dense_partitioned_aggregate(h, a, nrows, s, &spill, &n_spill, dense_generic && !vn_fold, &nspill, &n_nspill, vn_fold);
            vn_spill = prc == 0 && dense_vn && (spill || nspill);
            if (prc == 2 && h->hint == 0) {  // the dense attempt failed before G was ever estimated
                int64_t est = 0;
                VNM_TRY(estimate_groups(h, keys[0], est_rows, &est, s));
                h->hint = est;
                h->estimated = true;
            }
        }
        if (prc == 2 && h->hint > part_min) prc = run_partitioned();
        if (prc == 2 && a.part_wide && h->plan.n_cols == 0) {  // key-only entries and a region overflowed: see above
            h->key_only_failed = true;
            a.part_wide = 0;
            prc = run_partitioned();
        }
        // more groups than hinted (a partition overflowed its LDS table, or the dense output its allocation): estimate
        // the group count from the keys (< 1 ms) and partition again (~12 ms per attempt) before giving in to the HBM
        // table (~300 ms per 1e9 rows)
        for (int retry = 0; prc == 3 && retry < 2; retry++) {
            int64_t est = 0;
            if (retry == 0 && !h->estimated) {
                VNM_TRY(estimate_groups(h, keys[0], est_rows, &est, s));
                h->estimated = true;
            }
            h->hint = std::min<int64_t>(std::max<int64_t>(h->hint * 4, est + est / 4), (int64_t)1600 * env_i64("VNM_AGG_PART_L1_MAX", 256) * 512);
            prc = run_partitioned();
        }
        if (prc == 0 && !spill && !nspill) { h->rows_seen += nrows; return 0; }
        if (prc == 1) return 1;
        if (prc == VNM_RC_SINGLY) { VNM_SEG_ONLY("hash partitions"); }
        if (prc == 0 && vn_spill) {
            VNM_TRY(vn_spill_to_columns());
        } else if (prc == 0 && spill_is_wide) {   // spilled wide entries -> columns; the general scan takes them as a batch of its own
            const int E = 1 + h->plan.n_cols + (a.part_vmask ? 1 : 0);
            UnzipArgs u{};
            u.ent = (const uint64_t*)spill; u.n = n_spill; u.E = E; u.nval = h->plan.n_cols; u.has_vmask = a.part_vmask;
            u.key = (uint64_t*)unzip.take((size_t)n_spill * 8);
            bool ok = u.key != nullptr;
            for (int c = 0; c < h->plan.n_cols && ok; c++) {
                u.widths[c] = type_width(a.cols[c].type);
                u.vals[c] = unzip.take((size_t)n_spill * u.widths[c]);
                u.valid[c] = a.cols[c].validity ? (unsigned long long*)unzip.take((size_t)((n_spill + 63) / 64) * 8) : nullptr;
                ok = u.vals[c] && (!a.cols[c].validity || u.valid[c]);
            }
            if (!ok) return 1;
            const int ugrid = (int)std::min<int64_t>((n_spill + 255) / 256, (int64_t)cus * 8);
            spill_unzip_kernel<<<ugrid, 256, 0, s>>>(u);
            VNM_HIP(hipGetLastError());
            a.keys[0].values = u.key; a.keys[0].validity = nullptr; a.keys[0].offset = 0; a.keys[0].length = n_spill;
            for (int c = 0; c < h->plan.n_cols; c++) {
                a.cols[c].values = u.vals[c]; a.cols[c].validity = (const uint8_t*)u.valid[c]; a.cols[c].offset = 0; a.cols[c].length = n_spill;
            }
            a.nrows = n_spill;
            a.ntiles = (n_spill + AGG_TILE - 1) / AGG_TILE;
            a.p.enabled = 0;
            hot_scan = false;
            hotn = false;   // (columns with validity words, no predicate)
        } else if (prc == 0) {  // the scan below runs over the spilled entries only (predicate already applied)
            a.ent = spill;
            a.nrows = n_spill;
            a.p.enabled = 0;
        }
    }
    const int64_t scan_n = a.nrows;
    PoolScope seg_pool;
    if (h->segs_active && !a.ent) {   // the waiting batches of a stream as segments of one scan (agg_hot_kernel, the hot shape only)
        if (!(hot_scan && hot) || scan_no_pred || a.has_expr) VNM_SEG_ONLY("general scan");
    }
    if (h->kn_valid && !a.ent) VNM_SEG_ONLY("scan over the rows");   // (only the dense path's pass 1 reads the key's validity)
    VNM_TRY(ensure_table(h, scan_n, s, spill != nullptr || nspill != nullptr));
    if (hot_scan) a.ntiles = (scan_n + HOT_TILE - 1) / HOT_TILE;
    hotn = hotn && !a.ent && !scan_no_pred && !hot_scan;
    const int hotn_tile = AGG_BLOCK * 2 * (h->plan.n_cols == 3 && !h->pred_set ? 4 : 2);   // (agg_hotn_kernel's U)
    if (hotn) a.ntiles = (scan_n + hotn_tile - 1) / hotn_tile;
    if (h->segs_active && !a.ent) VNM_TRY(upload_segs(h, HOT_TILE, &a.segs, &a.nseg, &a.ntiles, seg_pool, s));
    const int lds_tile = AGG_TILE;
    int grid = h->single ? cus : cus * 4;
    if (grid > a.ntiles) grid = (int)a.ntiles;
    a.margin = (int64_t)grid * (h->single ? (a.lds_slots + 2 + (hot_scan || hotn ? HOT_TILE : lds_tile)) : AGG_TILE);
    progress = (unsigned int*)pool_alloc((size_t)grid * 4);
    if (!progress) return 1;
    VNM_HIP(hipMemsetAsync(progress, 0, (size_t)grid * 4, s));
    a.progress = progress;
    for (int round = 0;; round++) {
        // keep the load factor below 0.7 for everything that can be in flight
        while ((int64_t)(h->g.cap * 7 / 10) < a.margin + 1) VNM_TRY(table_grow(h, h->g.cap * 4, s));
        VNM_HIP(hipMemsetAsync(h->g.ctl, 0, 16, s));  // [1] overflow flag; [2] fill persists
        a.g = h->g;
        a.fill_limit = (int64_t)(h->g.cap * 7 / 10);
        {
        if (round == 0)
            route_note(a.ent ? "scan:spilled_entries" : (hot_scan ? (hot && a.nseg > 0 ? "scan:hot_segments" : (hot ? "scan:hot" : (hot_vnull ? "scan:hot_nullable_value" : (hot_two ? "scan:hot_two_columns" : "scan:hot_generic"))))
                                                                   : (hotn ? "scan:hotn" : (h->single ? "scan:lds_generic" : "scan:wide_keys"))),
                       "%lld rows, hint %lld (<= %lld or no partition rule applied), %d words per group, %d LDS slots", (long long)scan_n, (long long)h->hint, (long long)part_min, h->plan.n_words, a.lds_slots);
        KernelTimer timer("agg_scan", s);
        if (hot_scan) {
            size_t lds_bytes = (size_t)(S + 2) * 8 * (1 + h->plan.n_words);
#define VNM_HOT(P, V, HV, SI)                                                                                  \
    do {                                                                                                       \
        VNM_HIP(hipFuncSetAttribute((const void*)agg_hot_kernel<P, V, HV, SI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
        agg_hot_kernel<P, V, HV, SI><<<grid, AGG_BLOCK, lds_bytes, s>>>(a);                                    \
    } while (0)
            if (a.ent) {  // spilled entries of the partitioned path
                if (hot) {
                    VNM_HIP(hipFuncSetAttribute((const void*)agg_hot_kernel<false, false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                    agg_hot_kernel<false, false, true, true, true><<<grid, AGG_BLOCK, lds_bytes, s>>>(a);
                } else {
                    VNM_HIP(hipFuncSetAttribute((const void*)agg_hot_kernel<false, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                    agg_hot_kernel<false, false, true, false, true><<<grid, AGG_BLOCK, lds_bytes, s>>>(a);
                }
            } else if (hot && a.nseg > 0) {  // ... over the waiting batches of a stream
#define VNM_HOTS(P, V)                                                                                          \
    do {                                                                                                       \
        VNM_HIP(hipFuncSetAttribute((const void*)agg_hot_seg_kernel<P, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
        agg_hot_seg_kernel<P, V><<<grid, AGG_BLOCK, lds_bytes, s>>>(a);                                         \
    } while (0)
                if (!h->pred_set) VNM_HOTS(false, false);
                else if (a.hot_pred_is_v) VNM_HOTS(true, true);
                else VNM_HOTS(true, false);
#undef VNM_HOTS
            } else if (hot) {  // the north-star shape
                if (!h->pred_set) VNM_HOT(false, false, true, true);
                else if (a.hot_pred_is_v) VNM_HOT(true, true, true, true);
                else VNM_HOT(true, false, true, true);
            } else if (hot_vnull) {
#define VNM_HOTN(P, V)                                                                                          \
    do {                                                                                                       \
        VNM_HIP(hipFuncSetAttribute((const void*)agg_hot_kernel<P, V, true, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
        agg_hot_kernel<P, V, true, false, false, false, true><<<grid, AGG_BLOCK, lds_bytes, s>>>(a);           \
    } while (0)
                if (!h->pred_set || scan_no_pred) VNM_HOTN(false, false);
                else if (a.hot_pred_is_v) VNM_HOTN(true, true);
                else VNM_HOTN(true, false);
#undef VNM_HOTN
            } else if (hot_two && hot_scan) {
#define VNM_HOT2(P, V)                                                                                          \
    do {                                                                                                       \
        VNM_HIP(hipFuncSetAttribute((const void*)agg_hot_kernel<P, V, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
        agg_hot_kernel<P, V, true, false, false, true><<<grid, AGG_BLOCK, lds_bytes, s>>>(a);                  \
    } while (0)
                if (!h->pred_set) VNM_HOT2(false, false);
                else if (a.hot_pred_is_v) VNM_HOT2(true, true);
                else VNM_HOT2(true, false);
#undef VNM_HOT2
            } else if (!a.hot_has_val) { if (h->pred_set) VNM_HOT(true, false, false, false); else VNM_HOT(false, false, false, false); }
            else if (!h->pred_set) VNM_HOT(false, false, true, false);
            else if (a.hot_pred_is_v) VNM_HOT(true, true, true, false);
            else VNM_HOT(true, false, true, false);
#undef VNM_HOT
        } else if (hotn) {
            size_t lds_bytes = (size_t)(a.lds_slots + 2) * 8 * (1 + h->plan.n_words);
#define VNM_HN2(NC_, P_, PL_)                                                                                    \
    do {                                                                                                        \
        VNM_HIP(hipFuncSetAttribute((const void*)agg_hotn_kernel<NC_, P_, PL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
        agg_hotn_kernel<NC_, P_, PL_><<<grid, AGG_BLOCK, lds_bytes, s>>>(a);                                    \
    } while (0)
#define VNM_HN(NC_)                                                                                              \
    do {                                                                                                        \
        const bool plain_ = !a.hn_any_int && !a.hn_any_mm;                                                      \
        if (h->pred_set) { if (plain_) VNM_HN2(NC_, true, true); else VNM_HN2(NC_, true, false); }              \
        else { if (plain_) VNM_HN2(NC_, false, true); else VNM_HN2(NC_, false, false); }                        \
    } while (0)
            switch (h->plan.n_cols) {
                case 3: VNM_HN(3); break;
                case 4: VNM_HN(4); break;
                case 5: VNM_HN(5); break;
                default: VNM_HN(6); break;
            }
#undef VNM_HN
#undef VNM_HN2
        } else if (h->single) {
            size_t lds_bytes = (size_t)(a.lds_slots + 2) * 8 * (1 + h->plan.n_words);
            VNM_HIP(hipFuncSetAttribute((const void*)agg_lds_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            agg_lds_kernel<1024><<<grid, 1024, lds_bytes, s>>>(a);
        } else {
            agg_wide_kernel<<<grid, 256, 0, s>>>(a);
        }
        }
        VNM_HIP(hipGetLastError());
        unsigned long long ctl[4];
        VNM_HIP(hipMemcpyAsync(ctl, h->g.ctl, sizeof(ctl), hipMemcpyDeviceToHost, s));
        VNM_HIP(hipStreamSynchronize(s));
        if (ctl[1] == 2) return set_error("aggregate: HBM hash table overflow (internal error)");
        // a hint below the partitioning threshold with far more actual groups ran this scan into flush storms (77 ms per
        // 1e9 rows at G = 1e7): the table's fill is the lesson for the batches that follow (checking small hints up
        // front would cost every correctly hinted query ~0.3 ms)
        if (h->single && nrows >= (1 << 22) && (int64_t)ctl[2] > 2 * 2400 && (int64_t)ctl[2] > h->hint) h->hint = (int64_t)(ctl[2] + ctl[2] / 4);
        if (!ctl[1]) break;  // no block ran out of room: every tile was processed
        // grow (x4, or to the projected final size) and relaunch; blocks resume from progress[]
        uint64_t new_cap = h->g.cap * 4;
        if (h->hint <= 0 && round >= 1) new_cap = h->g.cap * 16;
        VNM_TRY(table_grow(h, new_cap, s));
    }
    h->rows_seen += nrows;
    return 0;
}

#undef span_fits
// The waiting batches of an asynchronous stream (vnm_agg_set_async) -> the device: as the segments of ONE logical batch where the
// path's kernels take segments (the dense-key path's ring scatter, the hot-shape LDS scan), one by one otherwise.
static int flush_queue(vnm_agg* h, void* stream) {
    if (h->q.empty()) return 0;
    std::vector<vnm_agg::QBatch> q;
    q.swap(h->q);
    const int64_t total = h->q_rows;
    h->q_rows = 0;
    int rc = VNM_RC_SINGLY;
    auto one = [&](const vnm_agg::QBatch& b, int64_t n) {
        vnm_dcol in[AGG_MAX_FUNCS];
        for (int i = 0; i < h->n_funcs; i++) {
            memset(&in[i], 0, sizeof(vnm_dcol));
            if (!b.ins.empty()) in[i] = b.ins[i];
            else if (h->func_col[i] >= 0) in[i] = b.col;
        }
        const int64_t save = h->cur_seq;
        h->cur_seq = b.seq;     // (whoever records the batch again -- the parts of a split program -- keeps its number)
        const int rc1 = next_device_impl(h, n, &b.key, in, h->pred_set ? &b.pred : nullptr, stream);
        h->cur_seq = save;
        return rc1;
    };
    if (q.size() > 1 && getenv("VNM_AGG_NO_SEGMENTS") == nullptr) {
        h->segs_active = &q;
        rc = one(q[0], total);
        h->segs_active = nullptr;
        if (rc != VNM_RC_SINGLY) route_note("stream:segments_of_one_launch", "%zu recorded batches, %lld rows", q.size(), (long long)total);
    }
    if (rc == VNM_RC_SINGLY) {
        if (q.size() > 1) route_note("stream:batches_singly", "%zu recorded batches (the path's kernels read one batch)", q.size());
        rc = 0;
        for (size_t i = 0; i < q.size() && !rc; i++) rc = one(q[i], q[i].nrows);
    }
    return rc;
}

// does this batch have the shape whose kernels take stream segments (VSeg)?  The hot shape: one plain 8-byte key, {COUNT(*), COUNT,
// SUM, AVG} of ONE plain float64 column, a plain float64 predicate column or none.
static bool queueable(const vnm_agg* h, int64_t nrows, const vnm_dcol* keys, const vnm_dcol* inputs, const vnm_dcol* pred, bool* pred_is_v, bool* multi) {
    *multi = false;
    if (!h->single || h->plan.n_keys != 1 || h->plan.n_cols < 1 || h->inner || h->tuple_mode || h->expr_col >= 0) return false;
    if (nrows <= 0 || nrows >= (1LL << 30)) return false;
    auto plain8 = [](const vnm_dcol& c) { return type_width(c.type) == 8 && !c.validity && (c.offset & 1) == 0 && ((uintptr_t)c.values & 15) == 0; };
    if (h->plan.n_cols > 1) {
        // SEVERAL plain 8-byte input columns over a plain int64 / uint64 key (round 4): the batches wait so that the path is chosen from
        // the stream's total row count (a 2^24-row batch alone is too short for a 2^27-code range: three columns, G = 1e8, 30 batches:
        // 315 ms through the hash partitions); when they go to the device the program is cut into parts (next_parts), which record
        // the batches in turn and launch once each -- or they go one by one, as before, where no part rule applies
        if (h->rank_aligned || !plain8(keys[0]) || keys[0].type != h->plan.key_types[0] || (keys[0].type != VNM_I64 && keys[0].type != VNM_U64)) return false;
        for (int i = 0; i < h->n_funcs; i++)
            if (h->func_col[i] >= 0 && (!plain8(inputs[i]) || (inputs[i].type != VNM_F64 && inputs[i].type != VNM_I64 && inputs[i].type != VNM_U64))) return false;
        if (h->pred_set && (!pred || !plain8(*pred) || pred->type != VNM_F64)) return false;
        *pred_is_v = false;
        *multi = true;
        return true;
    }
    if (!h->parts.empty()) return false;
    const vnm_dcol& col = inputs[h->col_first_func[0]];
    if (!plain8(keys[0]) || keys[0].type != h->plan.key_types[0] || !plain8(col) || col.type != VNM_F64) return false;
    for (int o = 0; o < h->plan.n_ops; o++) {
        const int k = h->plan.ops[o].kind;
        if (k != A_COUNT_ROWS && k != A_COUNT_VALID && k != A_SUM_F64) return false;
    }
    for (int i = 0; i < h->n_funcs; i++)   // every function reads that one column (COUNT(*): none)
        if (h->func_col[i] >= 0 && (inputs[i].values != col.values || inputs[i].offset != col.offset || inputs[i].validity)) return false;
    *pred_is_v = false;
    if (h->pred_set) {
        if (!pred || !plain8(*pred) || pred->type != VNM_F64) return false;
        *pred_is_v = pred->values == col.values && pred->offset == col.offset;
    }
    return true;
}

static int next_device_body(vnm_agg* h, int64_t nrows, const vnm_dcol* keys, const vnm_dcol* inputs, const vnm_dcol* pred, void* stream);

int vnm_agg_next_device(vnm_agg* h, int64_t nrows, const vnm_dcol* keys, const vnm_dcol* inputs,
                        const vnm_dcol* pred, void* stream) {
    if (!h) return set_error("vnm_agg_next_device: null handle");
    if (!h->child) h->cur_seq = h->seq++;
    vnm_dcol wide, wpred;
    vnm_dcol win[AGG_MAX_FUNCS];
    if (h->key_out_type >= 0) { VNM_TRY(ensure_init()); VNM_TRY(widen_key(h, nrows, &keys, &wide, as_stream(stream))); }
    if (h->any_widen_in) { VNM_TRY(ensure_init()); VNM_TRY(widen_inputs(h, nrows, &inputs, win, &pred, &wpred, as_stream(stream))); }
    const int rc = next_device_body(h, nrows, keys, inputs, pred, stream);
    if (h->key_out_type >= 0 || h->any_widen_in) release_widened(h, stream);
    return rc;
}

static int next_device_body(vnm_agg* h, int64_t nrows, const vnm_dcol* keys, const vnm_dcol* inputs, const vnm_dcol* pred, void* stream) {
    if (h->ex && inputs && (keys || h->plan.n_keys == 0)) {   // float MIN / MAX: the flag pass, and the ordered mode once the stream is unclean
        bool handled = false;
        VNM_TRY(exact_next(h, nrows, keys, inputs, pred, stream, &handled));
        if (handled) return 0;
    }
    if (h->async && keys && inputs) {
        bool piv = false, multi = false;
        if (queueable(h, nrows, keys, inputs, pred, &piv, &multi) && (h->q.empty() || (piv == h->q_pred_is_v && multi == h->q_multi))) {
            // (the first batch waits like any other: the estimates and the code range of the path are taken from the first SEGMENT
            // when the waiting batches are processed)
            if (h->q_rows + nrows > (1LL << 30) || h->q.size() >= 256) VNM_TRY(flush_queue(h, stream));
            vnm_agg::QBatch b{};
            b.seq = h->cur_seq;
            b.nrows = nrows; b.key = keys[0]; b.col = inputs[h->col_first_func[0]];
            if (multi) b.ins.assign(inputs, inputs + h->n_funcs);
            h->q_multi = multi;
            if (h->pred_set) b.pred = *pred;
            h->q.push_back(b);
            h->q_rows += nrows;
            h->q_pred_is_v = piv;
            invalidate_result(h);
            return 0;
        }
    }
    VNM_TRY(flush_queue(h, stream));
    return next_device_impl(h, nrows, keys, inputs, pred, stream);
}

int vnm_agg_set_async(vnm_agg* h, int enabled) {
    if (!h) return set_error("vnm_agg_set_async: null handle");
    if (!enabled && !h->q.empty()) return set_error("vnm_agg_set_async: batches are waiting (call vnm_agg_sync first)");
    for (vnm_agg* c : h->parts) VNM_TRY(vnm_agg_set_async(c, enabled));
    if (h->ex && h->ex->post) VNM_TRY(vnm_agg_set_async(h->ex->post, enabled));
    h->async = enabled != 0;
    return 0;
}

int vnm_agg_sync(vnm_agg* h, void* stream) {
    if (!h) return set_error("vnm_agg_sync: null handle");
    VNM_TRY(flush_queue(h, stream));
    for (vnm_agg* c : h->parts) VNM_TRY(flush_queue(c, stream));   // (the parts of a split program record their batches themselves)
    if (h->ex && h->ex->post) VNM_TRY(vnm_agg_sync(h->ex->post, stream));
    VNM_HIP(hipStreamSynchronize(as_stream(stream)));
    release_widened(h, stream, true);
    return 0;
}

// the recorded batches the handle (its parts, the suffix operator of an ordered MIN / MAX stream) still needs the buffers of
static void waiting_of(const vnm_agg* h, int64_t* batches, int64_t* rows, int64_t* oldest) {
    *batches += (int64_t)h->q.size();
    *rows += h->q_rows;
    for (const auto& b : h->q) if (*oldest < 0 || b.seq < *oldest) *oldest = b.seq;
    for (const vnm_agg* c : h->parts) waiting_of(c, batches, rows, oldest);
    if (h->ex && h->ex->post) waiting_of(h->ex->post, batches, rows, oldest);
}

int vnm_agg_waiting(vnm_agg* h, int64_t* batches, int64_t* rows, int64_t* oldest_seq, int64_t* last_seq) {
    if (!h) return set_error("vnm_agg_waiting: null handle");
    int64_t b = 0, r = 0, o = -1;
    waiting_of(h, &b, &r, &o);
    if (batches) *batches = b;     // (a batch the parts of a split program hold is counted once per part)
    if (rows) *rows = r;
    if (oldest_seq) *oldest_seq = o;
    if (last_seq) *last_seq = h->seq - 1;
    return 0;
}

// ---- expressions inside aggregates ---------------------------------------------------------------------------------------
int vnm_agg_set_input_expr(vnm_agg* h, int func_idx, int n_ins, const vnm_expr_ins* program, int n_cols) {
    if (!h || !program) return set_error("vnm_agg_set_input_expr: null argument");
    if (func_idx < 0 || func_idx >= h->n_funcs || h->func_col[func_idx] < 0) return set_error("vnm_agg_set_input_expr: function %d has no input column", func_idx);
    if (h->expr_col >= 0 && h->expr_col != h->func_col[func_idx]) return set_error("vnm_agg_set_input_expr: one expression input per operator (project the others first)");
    if (n_ins < 1 || n_ins > 64 || n_cols < 1 || n_cols > 16) return set_error("vnm_agg_set_input_expr: bad program size");
    if (h->c_in_types[func_idx] != VNM_F64 || h->widen_in[func_idx]) return set_error("vnm_agg_set_input_expr: declare the function's input type as float64 (the expression's result type)");
    h->expr_col = h->func_col[func_idx];
    h->expr_prog.assign(program, program + n_ins);
    h->expr_ncols = n_cols;
    // does it fit the in-register evaluator?
    bool ok = n_ins <= EXR_MAX_INS && n_cols <= EXR_MAX_COLS;
    int sp = 0;
    for (int i = 0; ok && i < n_ins; i++) {
        switch (program[i].op) {
            case VNM_EX_COL: ok = program[i].arg >= 0 && program[i].arg < n_cols && ++sp <= EXR_MAX_DEPTH; break;
            case VNM_EX_CONST_F: case VNM_EX_CONST_I: ok = ++sp <= EXR_MAX_DEPTH; break;
            case VNM_EX_NEG: ok = sp >= 1; break;
            case VNM_EX_ADD: case VNM_EX_SUB: case VNM_EX_MUL: case VNM_EX_DIV: ok = sp >= 2; sp--; break;
            default: ok = false; break;
        }
    }
    h->expr_fusable = ok && sp == 1;
    return 0;
}

int vnm_agg_next_device_expr(vnm_agg* h, int64_t nrows, const vnm_dcol* keys, const vnm_dcol* inputs, const vnm_dcol* pred,
                             int n_expr_cols, const vnm_dcol* expr_cols, void* stream) {
    VNM_TRY(ensure_init());
    if (!h || !expr_cols) return set_error("vnm_agg_next_device_expr: null argument");
    if (h->expr_col < 0) return set_error("vnm_agg_next_device_expr: call vnm_agg_set_input_expr first");
    if (n_expr_cols != h->expr_ncols) return set_error("vnm_agg_next_device_expr: the expression reads %d columns, %d given", h->expr_ncols, n_expr_cols);
    hipStream_t s = as_stream(stream);
    std::vector<vnm_dcol> in(inputs, inputs + h->n_funcs);
    // fused: the hot shape -- {COUNT(*), COUNT, SUM, AVG} of the expression alone, a plain 8-byte key (or no GROUP BY),
    // float64 columns without NULLs at even offsets, float64 predicate column or none
    bool fuse = h->expr_fusable && h->plan.n_cols == 1 && getenv("VNM_AGG_NO_HOT") == nullptr && getenv("VNM_AGG_NO_EXPR_FUSION") == nullptr &&
                nrows > 0 && (h->plan.n_keys == 0 || (h->single && h->plan.n_keys == 1 && type_width(keys[0].type) == 8 && !keys[0].validity && (keys[0].offset & 1) == 0));
    for (int o = 0; fuse && o < h->plan.n_ops; o++) {
        const int k = h->plan.ops[o].kind;
        fuse = k == A_COUNT_ROWS || k == A_COUNT_VALID || k == A_SUM_F64;
    }
    for (int c = 0; fuse && c < n_expr_cols; c++)
        fuse = expr_cols[c].type == VNM_F64 && !expr_cols[c].validity && (expr_cols[c].offset & 1) == 0 && expr_cols[c].length >= nrows;
    if (fuse && h->pred_set) fuse = pred && pred->type == VNM_F64 && !pred->validity && (pred->offset & 1) == 0;
    if (fuse) {
        ExprProg& e = h->expr_dev;
        e.n = (int)h->expr_prog.size();
        for (int c = 0; c < EXR_MAX_COLS; c++) e.cols[c] = c < n_expr_cols ? (const double*)expr_cols[c].values + expr_cols[c].offset : nullptr;
        for (int i = 0; i < e.n; i++) {
            const vnm_expr_ins& pi = h->expr_prog[i];
            e.ins[i].op = pi.op == VNM_EX_CONST_I ? VNM_EX_CONST_F : pi.op;
            e.ins[i].arg = pi.arg;
            e.ins[i].imm = pi.op == VNM_EX_CONST_I ? (double)pi.imm_i : pi.imm_f;
        }
        // a stand-in column view for the functions that read the expression: float64, no NULLs; its address is never
        // dereferenced (and equals no real column, so `predicate column == value column` cannot match by accident)
        vnm_dcol stand{};
        stand.values = (const void*)h; stand.type = VNM_F64; stand.length = nrows;
        for (int i = 0; i < h->n_funcs; i++) if (h->func_col[i] == h->expr_col) in[i] = stand;
        h->expr_active = true;
        const int rc = vnm_agg_next_device(h, nrows, keys, in.data(), pred, stream);
        h->expr_active = false;
        return rc;
    }
    // general case: one fused projection pass materialises the expression, the aggregate reads it like any column
    double* tmp = (double*)pool_alloc((size_t)(nrows > 0 ? nrows : 1) * 8);
    if (!tmp) return 1;
    int out_type = 0;
    int rc = vnm_project((int)h->expr_prog.size(), h->expr_prog.data(), n_expr_cols, expr_cols, nrows, tmp, &out_type, stream);
    if (!rc && out_type != VNM_F64) rc = set_error("vnm_agg_next_device_expr: the expression's result type is %d, float64 expected (cast in the projection)", out_type);
    if (!rc) {
        vnm_dcol col{};
        col.values = tmp; col.type = VNM_F64; col.length = nrows;
        for (int i = 0; i < h->n_funcs; i++) if (h->func_col[i] == h->expr_col) in[i] = col;
        rc = vnm_agg_next_device(h, nrows, keys, in.data(), pred, stream);
    }
    if (hipStreamSynchronize(s) != hipSuccess && !rc) rc = set_error("vnm_agg_next_device_expr: stream synchronisation failed");
    pool_free(tmp);
    return rc;
}

static int merge_device_impl(vnm_agg* h, int64_t n, uint64_t* const* key_words, uint64_t* const* acc_words, void* stream, int64_t blk_rows, const uint64_t* blk_base);

int vnm_agg_merge_device(vnm_agg* h, int64_t n, uint64_t* const* key_words, uint64_t* const* acc_words, void* stream) {
    return merge_device_impl(h, n, key_words, acc_words, stream, 0, nullptr);
}

static int merge_device_impl(vnm_agg* h, int64_t n, uint64_t* const* key_words, uint64_t* const* acc_words, void* stream, int64_t blk_rows, const uint64_t* blk_base) {
    VNM_TRY(ensure_init());
    if (!h) return set_error("vnm_agg_merge_device: null handle");
    hipStream_t s = as_stream(stream);
    VNM_TRY(flush_queue(h, stream));
    if (h->inner) VNM_TRY(demote_packed(h, s));
    VNM_TRY(collapse_parts(h, s));
    invalidate_result(h);
    VNM_TRY(ensure_table(h, n, s));
    if (n <= 0) return 0;
    if (h->plan.n_keys) {
        unsigned long long fill = 0;
        VNM_HIP(hipMemcpyAsync(&fill, h->g.ctl + 2, 8, hipMemcpyDeviceToHost, s));
        VNM_HIP(hipStreamSynchronize(s));
        uint64_t need = pow2_at_least((uint64_t)((fill + (uint64_t)n) * 10 / 7 + 16));
        if (need > h->g.cap) VNM_TRY(table_grow(h, need, s));
    }
    MergeArgs m{};
    m.plan = h->plan;
    m.g = h->g;
    m.n = n;
    m.src_stride = h->merge_stride > 0 ? h->merge_stride : 1;
    m.blk_rows = blk_rows; m.blk_base = blk_base;
    for (int j = 0; j < h->plan.kw; j++) m.src_key[j] = key_words[j];
    for (int w = 0; w < h->plan.n_words; w++) m.src_acc[w] = acc_words[w];
    int grid = device_info().num_cus * 8;
    int64_t need_blocks = (n + 255) / 256;
    if (grid > need_blocks) grid = (int)need_blocks;
    agg_merge_kernel<<<grid, 256, 0, s>>>(m);
    VNM_HIP(hipGetLastError());
    return 0;
}

// merge row-major partial groups [n][n_key_words + n_acc_words] (what vnm_agg_bucket_by_owner produces and the
// all_to_all delivers) into this handle
int vnm_agg_merge_rows(vnm_agg* h, int64_t n, const uint64_t* rows, void* stream) {
    if (!h) return set_error("vnm_agg_merge_rows: null handle");
    const int nw = h->plan.kw + h->plan.n_words;
    uint64_t* kw[AGG_MAX_KEYS + 1];
    uint64_t* aw[AGG_MAX_WORDS];
    for (int j = 0; j < h->plan.kw; j++) kw[j] = const_cast<uint64_t*>(rows) + j;
    for (int w = 0; w < h->plan.n_words; w++) aw[w] = const_cast<uint64_t*>(rows) + h->plan.kw + w;
    h->merge_stride = nw;
    int rc = vnm_agg_merge_device(h, n, kw, aw, stream);
    h->merge_stride = 0;
    return rc;
}

// `nblocks` blocks of (block_rows + 1) row-major rows [n_key_words + n_acc_words]; row 0 of a block is its header (word 0 = the number of
// partial groups in rows 1 ..): the receive buffer of distributed.exchange_small_fixed, merged without a host look at the counts
int vnm_agg_merge_row_blocks(vnm_agg* h, int nblocks, int64_t block_rows, const uint64_t* blocks, void* stream) {
    if (!h || nblocks < 1 || block_rows < 1 || !blocks) return set_error("vnm_agg_merge_row_blocks: bad argument");
    const int nw = h->plan.kw + h->plan.n_words;
    uint64_t* kw[AGG_MAX_KEYS + 1];
    uint64_t* aw[AGG_MAX_WORDS];
    for (int j = 0; j < h->plan.kw; j++) kw[j] = const_cast<uint64_t*>(blocks) + j;
    for (int w = 0; w < h->plan.n_words; w++) aw[w] = const_cast<uint64_t*>(blocks) + h->plan.kw + w;
    h->merge_stride = nw;
    const int rc = merge_device_impl(h, (int64_t)nblocks * block_rows, kw, aw, stream, block_rows, blocks);
    h->merge_stride = 0;
    return rc;
}

int vnm_agg_finish(vnm_agg* h, int64_t* n_groups, void* stream) {
    VNM_TRY(ensure_init());
    if (!h) return set_error("vnm_agg_finish: null handle");
    const int rc = h->ex && h->ex->switched ? exact_finish(h, n_groups, stream)   // prefix + suffix merged, float MIN / MAX composed in row order
                                            : agg_finish_core(h, n_groups, stream);
    release_widened(h, stream);
    return rc;
}

static int agg_finish_core(vnm_agg* h, int64_t* n_groups, void* stream) {
    hipStream_t s = as_stream(stream);
    VNM_TRY(flush_queue(h, stream));   // (the waiting batches of an asynchronous stream)
    if (h->n_groups >= 0) {
        if (n_groups) *n_groups = h->n_groups;
        return 0;
    }
    if (h->inner) {  // packed composite keys: finish the single-key operator, unpack its keys into the wide layout
        int64_t n = 0;
        VNM_TRY(vnm_agg_finish(h->inner, &n, stream));
        h->dstride = n + 2;
        h->dkey = (uint64_t*)pool_alloc((size_t)h->dstride * 8 * h->plan.kw);
        h->dacc = (uint64_t*)pool_alloc((size_t)h->dstride * 8 * h->plan.n_words);
        if (!h->dkey || !h->dacc) return 1;
        if (n > 0) {
            VNM_TRY(inner_keys(h, h->inner, n, h->dkey, h->dstride, s));
            for (int w = 0; w < h->plan.n_words; w++)
                VNM_HIP(hipMemcpyAsync(h->dacc + (size_t)w * h->dstride, h->inner->dacc + (size_t)w * h->inner->dstride, (size_t)n * 8,
                                       hipMemcpyDeviceToDevice, s));
            VNM_HIP(hipGetLastError());
            VNM_HIP(hipStreamSynchronize(s));
        }
        h->n_groups = n;
        if (n_groups) *n_groups = n;
        return 0;
    }
    VNM_TRY(flush_scan_pending(h, s));                 // a stream of small-range batches: its table, as a run
    VNM_TRY(collapse_parts(h, s));                     // a split program: its parts joined by key, as a run
    if (h->pending) VNM_TRY(complete_pending(h, s));   // the deferred final pass of the dense path, as a run
    if (h->have_run && h->have_table) {   // a big run + a few spilled groups in the table: fold the table into the run
        bool patched = false;
        VNM_TRY(merge_table_into_run(h, s, &patched));
    }
    if (h->have_run && !h->have_table) {  // the partitioned path already produced the dense result
        h->dkey = h->run_key; h->dacc = h->run_acc; h->dstride = h->run_stride; h->n_groups = h->run_n;
        h->result_is_run = true;
        if (n_groups) *n_groups = h->n_groups;
        return 0;
    }
    if (h->have_run) VNM_TRY(merge_run_into_table(h, s));
    if (!h->have_table) {
        if (h->plan.n_keys == 0) VNM_TRY(ensure_table(h, 0, s));  // OneGroup over no batches still yields one row
        else {
            h->n_groups = 0;
            if (n_groups) *n_groups = 0;
            return 0;
        }
    }
    unsigned long long fill = 0;
    VNM_HIP(hipMemcpyAsync(&fill, h->g.ctl + 2, 8, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    h->dstride = (int64_t)fill + 2;
    h->dkey = (uint64_t*)pool_alloc((size_t)h->dstride * 8 * (h->plan.kw ? h->plan.kw : 1));
    h->dacc = (uint64_t*)pool_alloc((size_t)h->dstride * 8 * h->plan.n_words);
    if (!h->dkey || !h->dacc) return 1;
    VNM_HIP(hipMemsetAsync(h->g.ctl + 3, 0, 8, s));
    CompactArgs c{};
    c.plan = h->plan;
    c.g = h->g;
    c.dkey = h->dkey;
    c.dacc = h->dacc;
    c.dstride = h->dstride;
    if (h->plan.n_keys) {
        int grid = device_info().num_cus * 8;
        int64_t need_blocks = ((int64_t)h->g.cap + 255) / 256;
        if (grid > need_blocks) grid = (int)need_blocks;
        agg_compact_kernel<<<grid, 256, 0, s>>>(c);
    }
    agg_compact_special_kernel<<<1, 64, 0, s>>>(c);
    VNM_HIP(hipGetLastError());
    unsigned long long cnt = 0;
    VNM_HIP(hipMemcpyAsync(&cnt, h->g.ctl + 3, 8, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    h->n_groups = (int64_t)cnt;
    if (n_groups) *n_groups = h->n_groups;
    return 0;
}

// Multi-GPU exchange helper: writes the finished dense run as rows of (key words, accumulator words) grouped
// by owner rank (vinum_amd/distributed.py::owner_of) into out_rows [n_groups][n_key_words + n_acc_words] and
// the per-owner row counts into counts_host[world].
int vnm_agg_bucket_by_owner(vnm_agg* h, int world, uint64_t* out_rows, int64_t* counts_host, void* stream) {
    VNM_TRY(ensure_init());
    if (!h || world < 1 || world > 64 || !counts_host) return set_error("vnm_agg_bucket_by_owner: bad argument");
    hipStream_t s = as_stream(stream);
    int64_t n = 0;
    VNM_TRY(vnm_agg_finish(h, &n, stream));
    for (int o = 0; o < world; o++) counts_host[o] = 0;
    if (n == 0) return 0;
    BucketArgs b{};
    b.nkw = h->plan.kw;
    b.nw = h->plan.kw + h->plan.n_words;
    for (int j = 0; j < h->plan.kw; j++) b.words[j] = h->dkey + (size_t)j * h->dstride;
    for (int w = 0; w < h->plan.n_words; w++) b.words[h->plan.kw + w] = h->dacc + (size_t)w * h->dstride;
    b.n = n;
    b.world = world;
    b.out = out_rows;
    if (h->plan.kw == 0) {  // ONE_GROUP: a single row, owner 0
        for (int w = 0; w < h->plan.n_words; w++)
            VNM_HIP(hipMemcpyAsync(out_rows + w, h->dacc + (size_t)w * h->dstride, 8, hipMemcpyDeviceToDevice, s));
        VNM_HIP(hipStreamSynchronize(s));
        counts_host[0] = n;
        return 0;
    }
    b.nb = (int)std::min<int64_t>((n + 4095) / 4096, (int64_t)device_info().num_cus * 8);
    b.per = ((n + b.nb - 1) / b.nb + 255) / 256 * 256;
    unsigned long long* ctr = (unsigned long long*)pool_alloc(((size_t)world * b.nb + BK_MAX_WORLD) * 8);
    if (!ctr) return 1;
    b.blk = ctr;
    b.totals = ctr + (size_t)world * b.nb;
    agg_bucket_count_kernel<<<b.nb, 256, 0, s>>>(b);
    agg_bucket_scan_kernel<<<1, BK_MAX_WORLD, 0, s>>>(b);
    agg_bucket_scatter_kernel<<<b.nb, 256, 0, s>>>(b);
    VNM_HIP(hipGetLastError());
    unsigned long long tot[BK_MAX_WORLD];
    VNM_HIP(hipMemcpyAsync(tot, b.totals, sizeof(unsigned long long) * world, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    for (int o = 0; o < world; o++) counts_host[o] = (int64_t)tot[o];
    pool_free(ctr);
    return 0;
}

#include "vnm_agg_exchange.inc"

#include "vnm_agg_result.inc"

#include "vnm_agg_exchange2.inc"

}  // extern "C"
