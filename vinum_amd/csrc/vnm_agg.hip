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

// =======================================================================================================
// Kernel 1: single 64-bit key, LDS pre-aggregation, generic accumulator program.
// LDS: lkey[S+2] then lacc[w][S+2].  Slot S = key equal to the EMPTY sentinel, slot S+1 = NULL key.
// =======================================================================================================
__device__ __forceinline__ void lds_flush(const AggArgs& a, uint64_t* lkey, uint64_t* lacc, int S, int tid, int nthreads, unsigned* s_new) {
    const int stride = S + 2;
    const int W = a.plan.n_words;
    for (int i = tid; i < stride; i += nthreads) {
        uint64_t k = lkey[i];
        if (k == EMPTY) continue;
        uint64_t slot;
        if (i < S) slot = gt_find_single(a.g, k, s_new);
        else {
            slot = a.g.cap + (uint64_t)(i - S);
            if (ld_agent(&a.g.tag[slot]) == EMPTY) st_agent(&a.g.tag[slot], 0);
        }
        for (int w = 0; w < W; w++) {
            uint64_t v = lacc[w * stride + i];
            int mk = a.plan.merge[w];
            if (v != merge_init(mk)) g_merge(&a.g.acc[(uint64_t)w * a.g.stride + slot], mk, v, (int64_t)a.g.stride);
            lacc[w * stride + i] = merge_init(mk);
        }
        lkey[i] = EMPTY;
    }
}

// Row-major slow path: rows (row0 + r * step for the bits r of `rows`) whose key is known to be valid, not NULL
// and not the EMPTY sentinel are merged straight into the HBM table.  One ROLLED loop that re-reads what it needs
// from memory, so it adds a few hundred bytes of code instead of a copy per unrolled row and op.
__device__ __forceinline__ uint64_t op_value_raw(int kind, int type, uint64_t raw);
// (skp / svp: the rows are those of a stream segment -- plain 8-byte key and input column, see VSeg)
__device__ __forceinline__ void agg_rows_to_table(const AggArgs& a, int64_t row0, int step, uint32_t rows, unsigned* s_new,
                                                  const uint64_t* skp = nullptr, const uint64_t* svp = nullptr) {
#pragma unroll 1
    for (int r = 0; rows; r++, rows >>= 1) {
        if (!(rows & 1u)) continue;
        const int64_t row = row0 + (int64_t)r * step;
        const uint64_t gs = gt_find_single(a.g, skp ? skp[row] : col_key_bits(a.keys[0], row), s_new);
#pragma unroll 1
        for (int o = 0; o < a.plan.n_ops; o++) {
            const AccOp& op = a.plan.ops[o];
            uint64_t v;
            bool have;
            if (skp) {
                v = op_value_raw(op.kind, a.hot_vtype, op.kind == A_COUNT_ROWS ? 0 : svp[row]);
                have = true;
            } else if (a.has_expr && op.kind != A_COUNT_ROWS) {   // hot shape: COUNT / float64 SUM of the expression, never NULL
                v = op.kind == A_COUNT_VALID ? 1ULL : (uint64_t)__double_as_longlong(expr_eval1(a.expr, row));
                have = true;
            } else have = op_value(op, a.cols, row, &v);
            if (have) g_merge(&a.g.acc[(uint64_t)op.word * a.g.stride + gs], a.plan.merge[op.word], v, (int64_t)a.g.stride);
        }
    }
}

// raw column bits (as col_raw_bits returns them: zero-extended) -> typed values
__device__ __forceinline__ int64_t raw_to_i64(int type, uint64_t raw) {
    switch (type) {
        case VNM_I8: return (int8_t)raw;
        case VNM_I16: return (int16_t)raw;
        case VNM_I32: return (int32_t)raw;
        default: return (int64_t)raw;  // unsigned types are zero-extended already
    }
}
__device__ __forceinline__ double raw_to_f64(int type, uint64_t raw) {
    if (type == VNM_F64) return __longlong_as_double((long long)raw);
    if (type == VNM_F32) return (double)__uint_as_float((uint32_t)raw);
    if (type == VNM_U64) return (double)raw;  // not through int64: values >= 2^63 would turn negative
    return (double)raw_to_i64(type, raw);
}
// what op `kind` contributes for a non-NULL input value (same table as op_value)
__device__ __forceinline__ uint64_t op_value_raw(int kind, int type, uint64_t raw) {
    switch (kind) {
        case A_COUNT_ROWS:
        case A_COUNT_VALID: return 1;
        case A_SUM_F64: return (uint64_t)__double_as_longlong(raw_to_f64(type, raw));
        case A_SUM_I64: return (uint64_t)raw_to_i64(type, raw);
        case A_SUM_LO32: return (uint64_t)raw_to_i64(type, raw) & 0xFFFFFFFFULL;
        case A_SUM_HI32S: return (uint64_t)(raw_to_i64(type, raw) >> 32);
        case A_SUM_HI32U: return (uint64_t)raw_to_i64(type, raw) >> 32;
        default:  // A_MIN / A_MAX on the order-preserving encoding
            if (type_is_float(type)) return enc_f64(raw_to_f64(type, raw));
            if (type_is_unsigned(type)) return (uint64_t)raw_to_i64(type, raw);
            return enc_i64(raw_to_i64(type, raw));
    }
}

// raw bits of R rows of a column, all loads issued back to back: the width switch is outside the row loop and
// nothing branches on loaded data (a load behind a data-dependent branch -- "read the key only if the predicate
// passed" -- serialises the rows: 16 dependent latencies per tile made the load phase alone cost 5.7 ms)
template <int R>
__device__ __forceinline__ void load_raw_rows(const vnm_dcol& c, const int64_t* rowc, uint64_t* raw) {
    const int64_t off = c.offset;
    switch (type_width(c.type)) {
        case 8: {
            const uint64_t* p = (const uint64_t*)c.values + off;
#pragma unroll
            for (int r = 0; r < R; r++) raw[r] = p[rowc[r]];
            break;
        }
        case 4: {
            const uint32_t* p = (const uint32_t*)c.values + off;
#pragma unroll
            for (int r = 0; r < R; r++) raw[r] = p[rowc[r]];
            break;
        }
        case 2: {
            const uint16_t* p = (const uint16_t*)c.values + off;
#pragma unroll
            for (int r = 0; r < R; r++) raw[r] = p[rowc[r]];
            break;
        }
        default: {
            const uint8_t* p = (const uint8_t*)c.values + off;
#pragma unroll
            for (int r = 0; r < R; r++) raw[r] = p[rowc[r]];
            break;
        }
    }
}
// validity bits of R rows as a mask (all ones without a bitmap)
template <int R>
__device__ __forceinline__ uint32_t load_valid_rows(const vnm_dcol& c, const int64_t* rowc) {
    if (!c.validity) return (1u << R) - 1u;
    uint8_t by[R];
#pragma unroll
    for (int r = 0; r < R; r++) by[r] = c.validity[(c.offset + rowc[r]) >> 3];
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < R; r++) m |= (uint32_t)((by[r] >> ((c.offset + rowc[r]) & 7)) & 1) << r;
    return m;
}
// pred_eval on already loaded bits
__device__ __forceinline__ bool pred_eval_raw(const Predicate& p, int type, uint64_t raw, bool valid) {
    switch (p.mode) {
        case CMP_F64: return cmp_apply<double>(p.op, valid ? raw_to_f64(type, raw) : __builtin_nan(""), p.dval);
        case CMP_F32: return cmp_apply<float>(p.op, __uint_as_float((uint32_t)raw), (float)p.dval);
        case CMP_I64: return cmp_apply<int64_t>(p.op, raw_to_i64(type, raw), p.ival);
        case CMP_U64: return cmp_apply<uint64_t>(p.op, (uint64_t)raw_to_i64(type, raw), (uint64_t)p.ival);
        default: return p.const_result != 0;
    }
}

// The tile is processed in phases so that every decode serves AGG_ROWS_PER_THREAD rows and loads of the same
// kind are in flight together: (A0) predicate + key loads, (A1) LDS probes -> one slot per row, (B) for each
// accumulator op: load its column for all rows, then merge.  (Row-major interpretation -- one op decode, one
// dependent load and one type switch per row and op -- ran at 12.7 ms per 1e9 rows for MIN+MAX.)
// BLK = 1024: one workgroup per CU with the largest LDS table (two 512-thread workgroups per CU with half-size
// tables were measured slower: 9.9-13.8 vs 9.9 ms).
__device__ __forceinline__ int hot_slot(uint64_t* lkey, int S, uint32_t smask, unsigned* s_fill, uint64_t key, uint32_t spread);

template <int BLK>
__global__ __launch_bounds__(BLK) void agg_lds_kernel(AggArgs a) {
    constexpr int R = AGG_ROWS_PER_THREAD;
    extern __shared__ uint64_t lds[];
    __shared__ unsigned s_fill, s_new;
    __shared__ int64_t s_tile;
    const int S = a.lds_slots;
    const int stride = S + 2;
    const int W = a.plan.n_words;
    uint64_t* lkey = lds;
    uint64_t* lacc = lds + stride;
    const int tid = threadIdx.x;

    for (int i = tid; i < stride; i += BLK) lkey[i] = EMPTY;
    for (int w = 0; w < W; w++) {
        uint64_t init = merge_init(a.plan.merge[w]);
        for (int i = tid; i < stride; i += BLK) lacc[w * stride + i] = init;
    }
    if (tid == 0) { s_fill = 0; s_new = 0; }
    __syncthreads();

    const unsigned flush_at = (unsigned)(S * 7 / 10);
    const uint32_t smask = (uint32_t)S - 1;
    const vnm_dcol& kc = a.keys[0];
    const bool key8 = type_width(kc.type) == 8;
    // The HBM table only grows when this workgroup flushes or when its LDS table is too full to take a key, so the
    // room check (an agent-scope read + a barrier per tile) is only repeated after a flush or above half load;
    // the margin the host reserves per workgroup (one LDS table + one tile) covers everything in between.
    bool need_check = true;
    uint32_t spread = 0;
    int spread_state = ((a.debug & 4) || a.ntiles < 64 * (int64_t)gridDim.x) ? 2 : 0;   // see agg_hot_kernel
    unsigned it = a.progress[blockIdx.x];
    for (;; it++) {
        const int64_t tile = (int64_t)blockIdx.x + (int64_t)it * gridDim.x;
        if (tile >= a.ntiles) break;
        const int64_t row0 = tile * (BLK * AGG_ROWS_PER_THREAD) + tid;
        // ---- A0: predicate and key of every row (st: 0 = no row, 1 = key, 2 = key equal to the EMPTY sentinel,
        // 3 = NULL key); the room check's agent-scope read overlaps these loads
        uint64_t key[R];
        int st[R];
        int64_t rowc[R];  // row index clamped into the batch, so every load below is unconditional
        uint32_t inr = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int64_t row = row0 + (int64_t)r * BLK;
            if (row < a.nrows) inr |= 1u << r;
            rowc[r] = row < a.nrows ? row : a.nrows - 1;
        }
        uint32_t pass = inr;
        {
            uint64_t praw[R];
            uint32_t pvalid = 0;
            if (a.p.enabled) {
                load_raw_rows<R>(a.pred, rowc, praw);
                pvalid = load_valid_rows<R>(a.pred, rowc);
            }
            load_raw_rows<R>(kc, rowc, key);
            const uint32_t kvalid = load_valid_rows<R>(kc, rowc);
            if (a.p.enabled) {
                // one uniform switch per tile, straight-line code per row
                const int pt = a.pred.type;
#define VNM_PRED_LOOP(EXPR)                                                                   \
    _Pragma("unroll") for (int r = 0; r < R; r++) { if (!(EXPR)) pass &= ~(1u << r); }
                switch (a.p.mode) {
                    case CMP_F64:
                        if (pt == VNM_F64) { VNM_PRED_LOOP(cmp_apply<double>(a.p.op, ((pvalid >> r) & 1u) ? __longlong_as_double((long long)praw[r]) : __builtin_nan(""), a.p.dval)) }
                        else { VNM_PRED_LOOP(cmp_apply<double>(a.p.op, ((pvalid >> r) & 1u) ? raw_to_f64(pt, praw[r]) : __builtin_nan(""), a.p.dval)) }
                        break;
                    case CMP_F32: VNM_PRED_LOOP(cmp_apply<float>(a.p.op, __uint_as_float((uint32_t)praw[r]), (float)a.p.dval)) break;
                    case CMP_I64: VNM_PRED_LOOP(cmp_apply<int64_t>(a.p.op, raw_to_i64(pt, praw[r]), a.p.ival)) break;
                    case CMP_U64: VNM_PRED_LOOP(cmp_apply<uint64_t>(a.p.op, (uint64_t)raw_to_i64(pt, praw[r]), (uint64_t)a.p.ival)) break;
                    default: if (!a.p.const_result) pass = 0; break;
                }
#undef VNM_PRED_LOOP
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                st[r] = 0;
                if ((pass >> r) & 1u) {
                    if (!((kvalid >> r) & 1u)) st[r] = 3;
                    else {
                        // raw bits -> key bits (array_iterators.h:215-217: ints sign-extended, floats by bit pattern)
                        if (!key8) key[r] = type_is_float(kc.type) ? key[r] : (uint64_t)raw_to_i64(kc.type, key[r]);
                        st[r] = key[r] == EMPTY ? 2 : 1;
                    }
                }
            }
        }
        if (need_check) {
            if (tid == 0) s_tile = table_has_room(a, &s_new) ? 1 : 0;
            __syncthreads();
            if (!s_tile) break;
        }
        // ---- A1: one LDS slot per row (-1 = no row, -2 = LDS table saturated for this key: straight to HBM)
        int slot[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            slot[r] = -1;
            if (st[r] == 3) { slot[r] = S + 1; lkey[S + 1] = 0; }
            else if (st[r] == 2) { slot[r] = S; lkey[S] = 0; }
            else if (st[r] == 1 && !(a.debug & 2)) {
                const int hs = hot_slot(lkey, S, smask, &s_fill, key[r], spread);
                slot[r] = hs < 0 ? -2 : hs;
            }
        }
        // keys the LDS table could not take go straight to the HBM table, out of line (this code must not be
        // replicated per row and op: at 69 KB the kernel overflowed the 64 KB instruction cache)
        {
            uint32_t sat = 0;
#pragma unroll
            for (int r = 0; r < R; r++) if (slot[r] == -2) sat |= 1u << r;
            if (sat) agg_rows_to_table(a, row0, BLK, sat, &s_new);
        }
        // ---- B: one accumulator op at a time over all rows of the lane
        uint64_t raw[R];
        uint32_t cvalid = 0, rows = 0;
        int loaded_col = -1;  // consecutive ops on one column share its loads
#pragma unroll
        for (int r = 0; r < R; r++) { raw[r] = 0; if (slot[r] != -1) rows |= 1u << r; }
        for (int o = 0; o < ((a.debug & 1) ? 0 : a.plan.n_ops); o++) {
            const AccOp op = a.plan.ops[o];
            uint32_t have = rows;  // rows that contribute to this op
            if (op.kind != A_COUNT_ROWS) {
                if (op.col != loaded_col) {
                    const vnm_dcol& c = a.cols[op.col];
                    load_raw_rows<R>(c, rowc, raw);
                    cvalid = load_valid_rows<R>(c, rowc);
                    loaded_col = op.col;
                }
                have &= cvalid;
            }
            const int vtype = op.kind == A_COUNT_ROWS ? VNM_U64 : a.cols[op.col].type;
            uint64_t* const wl = lacc + op.word * stride;
            // sign / zero extension of narrow integers without a per-row type switch: (x << sh) >> sh
            const int sh = 64 - 8 * type_width(vtype);
            const bool is_f = type_is_float(vtype), is_f32 = vtype == VNM_F32, is_u = type_is_unsigned(vtype);
#define VNM_I64(RAW) (is_u ? (int64_t)(RAW) : (int64_t)((RAW) << sh) >> sh)
#define VNM_F64V(RAW) (is_f ? (is_f32 ? (double)__uint_as_float((uint32_t)(RAW)) : __longlong_as_double((long long)(RAW))) : (double)VNM_I64(RAW))
            // LDS atomic for slot >= 0; saturated keys (slot -2, rare) are redone row-major by agg_rows_to_table
#define VNM_MERGE_LOOP(MK, VAL, LDSOP)                                                                      \
    _Pragma("unroll") for (int r = 0; r < R; r++) {                                                         \
        if (!((have >> r) & 1u) || slot[r] < 0) continue;                                                   \
        const uint64_t v = (VAL);                                                                           \
        LDSOP;                                                                                              \
    }
#define VNM_LADD(V) __hip_atomic_fetch_add(&wl[slot[r]], (V), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
            switch (op.kind) {
                case A_COUNT_ROWS:
                case A_COUNT_VALID: VNM_MERGE_LOOP(M_ADD_U64, 1ULL, VNM_LADD(v)) break;
                case A_SUM_F64:
                    if (a.plan.merge[op.word] == M_ADD_F64C) {
                        VNM_MERGE_LOOP(M_ADD_F64C, (uint64_t)__double_as_longlong(VNM_F64V(raw[r])),
                                       l_add_f64c(&wl[slot[r]], stride, __longlong_as_double((long long)v)))
                    } else {
                        VNM_MERGE_LOOP(M_ADD_F64, (uint64_t)__double_as_longlong(VNM_F64V(raw[r])),
                                       __hip_atomic_fetch_add((double*)&wl[slot[r]], __longlong_as_double((long long)v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                    }
                    break;
                case A_SUM_I64: VNM_MERGE_LOOP(M_ADD_U64, (uint64_t)VNM_I64(raw[r]), VNM_LADD(v)) break;
                case A_SUM_LO32: VNM_MERGE_LOOP(M_ADD_U64, (uint64_t)VNM_I64(raw[r]) & 0xFFFFFFFFULL, VNM_LADD(v)) break;
                case A_SUM_HI32S: VNM_MERGE_LOOP(M_ADD_U64, (uint64_t)(VNM_I64(raw[r]) >> 32), VNM_LADD(v)) break;
                case A_SUM_HI32U: VNM_MERGE_LOOP(M_ADD_U64, (uint64_t)VNM_I64(raw[r]) >> 32, VNM_LADD(v)) break;
                case A_MIN:
                    VNM_MERGE_LOOP(M_MIN_U64, is_f ? enc_f64(VNM_F64V(raw[r])) : (is_u ? raw[r] : enc_i64(VNM_I64(raw[r]))),
                                   __hip_atomic_fetch_min(&wl[slot[r]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                    break;
                default:  // A_MAX
                    VNM_MERGE_LOOP(M_MAX_U64, is_f ? enc_f64(VNM_F64V(raw[r])) : (is_u ? raw[r] : enc_i64(VNM_I64(raw[r]))),
                                   __hip_atomic_fetch_max(&wl[slot[r]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                    break;
            }
#undef VNM_LADD
#undef VNM_MERGE_LOOP
#undef VNM_F64V
#undef VNM_I64
        }
        __syncthreads();
        const unsigned fill_now = s_fill;
        need_check = fill_now > (unsigned)S / 2;
        if (spread_state == 0) { spread_state = fill_now <= (unsigned)S / 128 ? 1 : 2; if (spread_state == 1) spread = tid & 7u; }  // see agg_hot_kernel
        else if (spread_state == 1 && fill_now > (unsigned)S / 8) { spread_state = 2; spread = 0; }
        if (fill_now > flush_at) {
            lds_flush(a, lkey, lacc, S, tid, BLK, &s_new);
            __syncthreads();
            if (tid == 0) s_fill = 0;
        }
    }
    lds_flush(a, lkey, lacc, S, tid, BLK, &s_new);
    __syncthreads();
    if (tid == 0) { fold_new(a.g, &s_new); a.progress[blockIdx.x] = it; }
}

// =======================================================================================================
// Kernel 1h: the hot shape of the north-star query
//     SELECT k, {sum|avg|count}(v), count(*) [WHERE p > X] GROUP BY k
// with an 8-byte key, a float64 input, no validity bitmaps and even Arrow offsets -- and, with the same code,
// every other accumulator kind (MIN / MAX, int64 / uint64 sums and 128-bit sums) over ONE 8-byte input column
// without NULLs, or no input column at all (COUNT(*): configs[0]'s query shape).  Same LDS table and
// flush protocol as agg_lds_kernel, but every lane issues 16-byte loads (two rows), four requests per
// column in flight, the predicate / hash / accumulate sequence is straight-line code, and the block only
// synchronises once per 8192 rows (to decide about flushing).
// =======================================================================================================
constexpr int HOT_UNROLL = 4;
constexpr int HOT_TILE = AGG_BLOCK * 2 * HOT_UNROLL;  // 8192 rows per block iteration

// probe / claim the LDS slot of `key`; -1 = the table is saturated for this key
// spread (0..7, per lane): with very few groups the lanes of a wave that hold the same key would all hit ONE accumulator
// address (LDS atomics on one address are serial); xor-ing the lane's low bits into the home slot gives every key up to
// eight copies in adjacent banks, which the flush merges by key like any other slot.
__device__ __forceinline__ int hot_slot(uint64_t* lkey, int S, uint32_t smask, unsigned* s_fill, uint64_t key, uint32_t spread) {
    if (key == EMPTY) { lkey[S] = 0; return S; }
    const uint32_t hv = hash_u64(key);
    uint32_t h = (hv ^ spread) & smask;
    const uint32_t step = ((hv >> 20) & 31u) * 2u + 1u;  // double hashing: shorter worst chains than linear probing
    // one divergent region (the claim) and one exit per iteration: the scan issues fewer scalar exec-mask instructions
    for (int probe = 0; probe < AGG_MAX_PROBES; probe++) {
        uint64_t k = __hip_atomic_load(&lkey[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k == EMPTY) {
            uint64_t expected = EMPTY;
            const bool won = __hip_atomic_compare_exchange_strong(&lkey[h], &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_WORKGROUP);
            if (won) atomicAdd(s_fill, 1u);
            k = won ? key : expected;
        }
        if (k == key) return (int)h;
        h = (h + step) & smask;
    }
    return -1;
}

__device__ __forceinline__ void pa_accumulate_col(unsigned long long pack, int type, uint64_t* lw, int ST, int slot, uint64_t raw, int comp);

// every accumulator word this query has, updated for a row whose input value has the raw bits vb
// (SIMPLE: only COUNT(*), COUNT and the float64 sum can be present -- the north-star shape keeps its short path)
template <bool SIMPLE>
__device__ __forceinline__ void hot_accumulate(const AggArgs& a, uint64_t* lacc, int stride, int slot, uint64_t vb, bool valid = true) {
    if (SIMPLE) {
        if (a.hot_w_rows >= 0) __hip_atomic_fetch_add(&lacc[a.hot_w_rows * stride + slot], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (a.hot_w_valid >= 0) __hip_atomic_fetch_add(&lacc[a.hot_w_valid * stride + slot], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (a.hot_w_sum >= 0) {
            if (a.hot_comp) l_add_f64c(&lacc[a.hot_w_sum * stride + slot], stride, __longlong_as_double((long long)vb));
            else __hip_atomic_fetch_add((double*)&lacc[a.hot_w_sum * stride + slot], __longlong_as_double((long long)vb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    if (a.hot_w[A_COUNT_ROWS] >= 0) __hip_atomic_fetch_add(&lacc[a.hot_w[A_COUNT_ROWS] * stride + slot], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (valid) pa_accumulate_col(a.hot_wpack, a.hot_vtype, lacc, stride, slot, vb, a.hot_comp);  // a NULL input only counts for COUNT(*)
}

// one entry (key, value bits) straight into the HBM table: the saturated-key path of agg_hot_kernel<FROM_ENT>
__device__ __forceinline__ void hot_entry_to_table(const AggArgs& a, uint64_t key, uint64_t vb, unsigned* s_new) {
    uint64_t gs;
    if (key == EMPTY) { gs = a.g.cap; if (ld_agent(&a.g.tag[gs]) == EMPTY) st_agent(&a.g.tag[gs], 0); }
    else gs = gt_find_single(a.g, key, s_new);
#pragma unroll 1
    for (int k = 0; k <= A_MAX; k++) {
        const int w = a.hot_w[k];
        if (w < 0) continue;
        g_merge(&a.g.acc[(uint64_t)w * a.g.stride + gs], a.plan.merge[w], op_value_raw(k, a.hot_vtype, vb), (int64_t)a.g.stride);
    }
}

// FROM_ENT: the rows are (key, value bits) entries (a.ent) instead of columns -- what the partitioned path spills when a
// region is full (heavy keys); no predicate (already applied).
// TWO: a second 8-byte input column (its accumulator words in hot_w2).
// VNULL: the input column has a validity bitmap (one byte per lane and chunk: both rows of a pair share it); a NULL
// fails a predicate on that column and otherwise only counts for COUNT(*).
template <bool HAS_PRED, bool PRED_IS_V, bool HAS_VAL, bool SIMPLE, bool FROM_ENT = false, bool TWO = false, bool VNULL = false>
__global__ __launch_bounds__(AGG_BLOCK) void agg_hot_kernel(AggArgs a) {
    extern __shared__ uint64_t lds[];
    __shared__ unsigned s_fill, s_new;
    __shared__ int s_go;
    const int S = a.lds_slots;
    const int stride = S + 2;
    const int W = a.plan.n_words;
    uint64_t* lkey = lds;
    uint64_t* lacc = lds + stride;
    const int tid = threadIdx.x;

    for (int i = tid; i < stride; i += AGG_BLOCK) lkey[i] = EMPTY;
    for (int w = 0; w < W; w++) {
        uint64_t init = merge_init(a.plan.merge[w]);
        for (int i = tid; i < stride; i += AGG_BLOCK) lacc[w * stride + i] = init;
    }
    if (tid == 0) { s_fill = 0; s_new = 0; }
    __syncthreads();

    const unsigned flush_at = (unsigned)(S * 6 / 10);
    const uint32_t smask = (uint32_t)S - 1;
    const uint64_t* kp = (const uint64_t*)a.keys[0].values + a.keys[0].offset;
    const uint64_t* vp = HAS_VAL ? (const uint64_t*)a.cols[0].values + a.cols[0].offset : kp;
    const uint64_t* vp2 = TWO ? (const uint64_t*)a.cols[1].values + a.cols[1].offset : kp;
    const double* pp = (const double*)a.pred.values + a.pred.offset;
    const int op = a.p.op;
    const double thr = a.p.dval;

    ulonglong2 kk[HOT_UNROLL], vv[HOT_UNROLL], vw[HOT_UNROLL];  // this block's current (then next) tile, see below
    uint32_t vm[HOT_UNROLL];  // VNULL: validity bits of the pair (bit 0 / 1)
    const uint8_t* vbm = VNULL ? a.cols[0].validity : nullptr;
    const int64_t voff = VNULL ? a.cols[0].offset : 0;
    double2 pv[HOT_UNROLL];
    bool have = false;
    uint32_t spread = 0;
    // (no copies for short batches either: eight copies of every key are eight times the flush's atomics on the same few HBM
    // addresses -- ~30 us per kernel with 7 groups and 256 workgroups, a third of a 2^24-row batch's 90 us; they pay from ~64
    // tiles per workgroup on: 59 x 2^24-row batches, G = 7: 7.6 -> 5.8 ms)
    int spread_state = ((a.debug & 4) || a.ntiles < 64 * (int64_t)gridDim.x) ? 2 : 0;  // VNM_AGG_DEBUG & 4: no key copies (measurement)
    bool need_check = true;
    unsigned it = a.progress[blockIdx.x];
    for (;; it++) {
        const int64_t tile = (int64_t)blockIdx.x + (int64_t)it * gridDim.x;
        if (tile >= a.ntiles) break;
        if (need_check) {  // see agg_lds_kernel
            if (tid == 0) s_go = table_has_room(a, &s_new) ? 1 : 0;
            __syncthreads();
            if (!s_go) break;
        }
        const int64_t base = tile * HOT_TILE + 2 * tid;
        uint32_t sat0 = 0, sat1 = 0;  // rows (even / odd element of chunk u) whose key the LDS table could not take
        if (base + (int64_t)(HOT_UNROLL - 1) * 2 * AGG_BLOCK + 1 < a.nrows) {
            // Register rotation: as soon as chunk u of this tile has been copied out, chunk u of the block's NEXT
            // tile is requested into the same registers, so HBM loads are in flight while the LDS work of this tile
            // runs (one 1024-thread block per CU: without this the block alternates between a load phase and an
            // LDS phase -- G=1000 ran at 4.2 ms against 3.1 ms for G=7).
#define VNM_HOT_LOAD(u, b)                                                                                      \
    do {                                                                                                       \
        const int64_t r_ = (b) + (int64_t)(u) * 2 * AGG_BLOCK;                                                 \
        if (FROM_ENT) {                                                                                        \
            const ulonglong2 e0 = a.ent[r_], e1 = a.ent[r_ + 1];                                               \
            kk[u].x = e0.x; kk[u].y = e1.x;                                                                    \
            vv[u].x = e0.y; vv[u].y = e1.y;                                                                    \
        } else {                                                                                               \
            kk[u] = *(const ulonglong2*)(kp + r_);                                                             \
            if (HAS_VAL) {                                                                                     \
                if (a.has_expr) { const double2 ev_ = expr_eval2(a.expr, r_); vv[u].x = (unsigned long long)__double_as_longlong(ev_.x); vv[u].y = (unsigned long long)__double_as_longlong(ev_.y); } \
                else vv[u] = *(const ulonglong2*)(vp + r_);                                                    \
            }                                                                                                  \
            if (VNULL) vm[u] = (uint32_t)vbm[(voff + r_) >> 3] >> ((voff + r_) & 7);                           \
            if (TWO) vw[u] = *(const ulonglong2*)(vp2 + r_);                                                   \
            if (HAS_PRED && !PRED_IS_V) pv[u] = *(const double2*)(pp + r_);                                    \
        }                                                                                                      \
    } while (0)
            if (!have) {
#pragma unroll
                for (int u = 0; u < HOT_UNROLL; u++) VNM_HOT_LOAD(u, base);
            }
            const int64_t nbase = base + (int64_t)gridDim.x * HOT_TILE;
            const bool nfull = tile + gridDim.x < a.ntiles && nbase + (int64_t)(HOT_UNROLL - 1) * 2 * AGG_BLOCK + 1 < a.nrows;
#pragma unroll
            for (int u = 0; u < HOT_UNROLL; u++) {
                const ulonglong2 k = kk[u];
                const uint64_t v0 = HAS_VAL ? vv[u].x : 0, v1 = HAS_VAL ? vv[u].y : 0;
                const uint64_t w0 = TWO ? vw[u].x : 0, w1 = TWO ? vw[u].y : 0;
                const bool ok0 = !VNULL || (vm[u] & 1u), ok1 = !VNULL || (vm[u] & 2u);
                // a NULL predicate value compares like NaN (pred_eval: the reference sees NumPy NaNs there)
                const double p0 = PRED_IS_V ? (ok0 ? __longlong_as_double((long long)v0) : __builtin_nan("")) : pv[u].x;
                const double p1 = PRED_IS_V ? (ok1 ? __longlong_as_double((long long)v1) : __builtin_nan("")) : pv[u].y;
                if (nfull) VNM_HOT_LOAD(u, nbase);
                if (!HAS_PRED || cmp_apply<double>(op, p0, thr)) {
                    int slot = hot_slot(lkey, S, smask, &s_fill, k.x, spread);
                    if (slot >= 0) {
                        hot_accumulate<SIMPLE>(a, lacc, stride, slot, v0, ok0);
                        if (TWO) pa_accumulate_col(a.hot_wpack2, a.hot_vtype2, lacc, stride, slot, w0, a.hot_comp);
                    } else if (FROM_ENT) hot_entry_to_table(a, k.x, v0, &s_new);  // no columns to re-read: merge right here
                    else sat0 |= 1u << u;
                }
                if (!HAS_PRED || cmp_apply<double>(op, p1, thr)) {
                    int slot = hot_slot(lkey, S, smask, &s_fill, k.y, spread);
                    if (slot >= 0) {
                        hot_accumulate<SIMPLE>(a, lacc, stride, slot, v1, ok1);
                        if (TWO) pa_accumulate_col(a.hot_wpack2, a.hot_vtype2, lacc, stride, slot, w1, a.hot_comp);
                    } else if (FROM_ENT) hot_entry_to_table(a, k.y, v1, &s_new);
                    else sat1 |= 1u << u;
                }
            }
            have = nfull;
#undef VNM_HOT_LOAD
        } else {
            have = false;
            for (int u = 0; u < HOT_UNROLL; u++)
                for (int e = 0; e < 2; e++) {
                    int64_t r = base + (int64_t)u * 2 * AGG_BLOCK + e;
                    if (r >= a.nrows) continue;
                    const uint64_t vb = FROM_ENT ? a.ent[r].y : (HAS_VAL ? (a.has_expr ? (uint64_t)__double_as_longlong(expr_eval1(a.expr, r)) : vp[r]) : 0);
                    const uint64_t kb = FROM_ENT ? a.ent[r].x : kp[r];
                    const bool ok = !VNULL || ((vbm[(voff + r) >> 3] >> ((voff + r) & 7)) & 1);
                    const double p = PRED_IS_V ? (ok ? __longlong_as_double((long long)vb) : __builtin_nan("")) : (HAS_PRED ? pp[r] : 0.0);
                    if (HAS_PRED && !cmp_apply<double>(op, p, thr)) continue;
                    int slot = hot_slot(lkey, S, smask, &s_fill, kb, spread);
                    if (slot >= 0) {
                        hot_accumulate<SIMPLE>(a, lacc, stride, slot, vb, ok);
                        if (TWO) pa_accumulate_col(a.hot_wpack2, a.hot_vtype2, lacc, stride, slot, vp2[r], a.hot_comp);
                    }
                    else if (FROM_ENT) hot_entry_to_table(a, kb, vb, &s_new);
                    else if (e == 0) sat0 |= 1u << u;
                    else sat1 |= 1u << u;
                }
        }
        // saturated keys: straight to the HBM table, out of line (rows base + e + u * 2 * AGG_BLOCK)
        if (!FROM_ENT && sat0) agg_rows_to_table(a, base, 2 * AGG_BLOCK, sat0, &s_new);
        if (!FROM_ENT && sat1) agg_rows_to_table(a, base + 1, 2 * AGG_BLOCK, sat1, &s_new);
        __syncthreads();
        const unsigned fill_now = s_fill;
        need_check = fill_now > (unsigned)S / 2;
        // key copies (see hot_slot): on after the first tile when it found a handful of groups, off for good once the
        // table holds more than that would explain
        if (spread_state == 0) { spread_state = fill_now <= (unsigned)S / 128 ? 1 : 2; if (spread_state == 1) spread = tid & 7u; }
        else if (spread_state == 1 && fill_now > (unsigned)S / 8) { spread_state = 2; spread = 0; }
        if (fill_now > flush_at) {
            lds_flush(a, lkey, lacc, S, tid, AGG_BLOCK, &s_new);
            __syncthreads();
            if (tid == 0) s_fill = 0;
        }
    }
    lds_flush(a, lkey, lacc, S, tid, AGG_BLOCK, &s_new);
    __syncthreads();
    if (tid == 0) { fold_new(a.g, &s_new); a.progress[blockIdx.x] = it; }
}

// The same scan over the waiting batches of a STREAM (a.segs, see VSeg) -- the north-star shape only (SIMPLE: COUNT(*), COUNT, SUM of
// one plain float64 column).  A kernel of its own: the tile -> (segment, local tile) state costs the one-batch kernel above
// 12-36 VGPRs and, in its widest variants, spills.
template <bool HAS_PRED, bool PRED_IS_V>
__global__ __launch_bounds__(AGG_BLOCK) void agg_hot_seg_kernel(AggArgs a) {
    constexpr bool HAS_VAL = true, SIMPLE = true, FROM_ENT = false, TWO = false, VNULL = false;
    extern __shared__ uint64_t lds[];
    __shared__ unsigned s_fill, s_new;
    __shared__ int s_go;
    const int S = a.lds_slots;
    const int stride = S + 2;
    const int W = a.plan.n_words;
    uint64_t* lkey = lds;
    uint64_t* lacc = lds + stride;
    const int tid = threadIdx.x;

    for (int i = tid; i < stride; i += AGG_BLOCK) lkey[i] = EMPTY;
    for (int w = 0; w < W; w++) {
        uint64_t init = merge_init(a.plan.merge[w]);
        for (int i = tid; i < stride; i += AGG_BLOCK) lacc[w * stride + i] = init;
    }
    if (tid == 0) { s_fill = 0; s_new = 0; }
    __syncthreads();

    const unsigned flush_at = (unsigned)(S * 6 / 10);
    const uint32_t smask = (uint32_t)S - 1;
    // The rows of a tile: (segment, local tile) -- the record batches of a stream as one logical batch (VSeg; nseg = 0: the one batch
    // of a.keys / a.cols / a.pred).  Uniform over the workgroup; the segment cursor only moves forward.
    struct Cur { const uint64_t* kp; const uint64_t* vp; const uint64_t* vp2; const double* pp; const uint8_t* vbm; int64_t voff; int64_t nrows; int64_t lt; };
    const int nseg = a.nseg;
    const VSegConst segs = seg_table(a.segs);
    int sg = 0;
    auto locate = [&](int64_t tile, Cur& c) {
        while (sg + 1 < nseg && tile >= segs[sg + 1].first_tile) sg++;
        c.kp = segs[sg].kp; c.vp = segs[sg].vp; c.vp2 = c.kp; c.pp = segs[sg].pp; c.vbm = nullptr; c.voff = 0;
        c.nrows = segs[sg].nrows; c.lt = tile - segs[sg].first_tile;
    };
    const int op = a.p.op;
    const double thr = a.p.dval;

    ulonglong2 kk[HOT_UNROLL], vv[HOT_UNROLL], vw[HOT_UNROLL];  // this block's current (then next) tile, see below
    uint32_t vm[HOT_UNROLL];  // VNULL: validity bits of the pair (bit 0 / 1)
    double2 pv[HOT_UNROLL];
    bool have = false;
    uint32_t spread = 0;
    // (no copies for short batches either: eight copies of every key are eight times the flush's atomics on the same few HBM
    // addresses -- ~30 us per kernel with 7 groups and 256 workgroups, a third of a 2^24-row batch's 90 us; they pay from ~64
    // tiles per workgroup on: 59 x 2^24-row batches, G = 7: 7.6 -> 5.8 ms)
    int spread_state = ((a.debug & 4) || a.ntiles < 64 * (int64_t)gridDim.x) ? 2 : 0;  // VNM_AGG_DEBUG & 4: no key copies (measurement)
    bool need_check = true;
    unsigned it = a.progress[blockIdx.x];
    Cur cur{}, nxt{};
    for (;; it++) {
        const int64_t tile = (int64_t)blockIdx.x + (int64_t)it * gridDim.x;
        if (tile >= a.ntiles) break;
        if (need_check) {  // see agg_lds_kernel
            if (tid == 0) s_go = table_has_room(a, &s_new) ? 1 : 0;
            __syncthreads();
            if (!s_go) break;
        }
        if (have) cur = nxt; else locate(tile, cur);
        const uint64_t* const kp = cur.kp; const uint64_t* const vp = cur.vp; const uint64_t* const vp2 = cur.vp2;
        const double* const pp = cur.pp; const uint8_t* const vbm = cur.vbm; const int64_t voff = cur.voff;
        const int64_t base = cur.lt * HOT_TILE + 2 * tid;
        uint32_t sat0 = 0, sat1 = 0;  // rows (even / odd element of chunk u) whose key the LDS table could not take
        if (cur.lt * HOT_TILE + HOT_TILE <= cur.nrows) {   // a full tile
            // Register rotation: as soon as chunk u of this tile has been copied out, chunk u of the block's NEXT
            // tile is requested into the same registers, so HBM loads are in flight while the LDS work of this tile
            // runs (one 1024-thread block per CU: without this the block alternates between a load phase and an
            // LDS phase -- G=1000 ran at 4.2 ms against 3.1 ms for G=7).
#define VNM_HOT_LOAD(u, b, C)                                                                                   \
    do {                                                                                                       \
        const int64_t r_ = (b) + (int64_t)(u) * 2 * AGG_BLOCK;                                                 \
        if (FROM_ENT) {                                                                                        \
            const ulonglong2 e0 = a.ent[r_], e1 = a.ent[r_ + 1];                                               \
            kk[u].x = e0.x; kk[u].y = e1.x;                                                                    \
            vv[u].x = e0.y; vv[u].y = e1.y;                                                                    \
        } else {                                                                                               \
            kk[u] = *(const ulonglong2*)((C).kp + r_);                                                         \
            if (HAS_VAL) {                                                                                     \
                if (a.has_expr) { const double2 ev_ = expr_eval2(a.expr, r_); vv[u].x = (unsigned long long)__double_as_longlong(ev_.x); vv[u].y = (unsigned long long)__double_as_longlong(ev_.y); } \
                else vv[u] = *(const ulonglong2*)((C).vp + r_);                                                \
            }                                                                                                  \
            if (VNULL) vm[u] = (uint32_t)(C).vbm[((C).voff + r_) >> 3] >> (((C).voff + r_) & 7);               \
            if (TWO) vw[u] = *(const ulonglong2*)((C).vp2 + r_);                                               \
            if (HAS_PRED && !PRED_IS_V) pv[u] = *(const double2*)((C).pp + r_);                                \
        }                                                                                                      \
    } while (0)
            if (!have) {
#pragma unroll
                for (int u = 0; u < HOT_UNROLL; u++) VNM_HOT_LOAD(u, base, cur);
            }
            bool nfull = tile + gridDim.x < a.ntiles;
            if (nfull) { locate(tile + gridDim.x, nxt); nfull = nxt.lt * HOT_TILE + HOT_TILE <= nxt.nrows; }
            const int64_t nbase = nxt.lt * HOT_TILE + 2 * tid;
#pragma unroll
            for (int u = 0; u < HOT_UNROLL; u++) {
                const ulonglong2 k = kk[u];
                const uint64_t v0 = HAS_VAL ? vv[u].x : 0, v1 = HAS_VAL ? vv[u].y : 0;
                const uint64_t w0 = TWO ? vw[u].x : 0, w1 = TWO ? vw[u].y : 0;
                const bool ok0 = !VNULL || (vm[u] & 1u), ok1 = !VNULL || (vm[u] & 2u);
                // a NULL predicate value compares like NaN (pred_eval: the reference sees NumPy NaNs there)
                const double p0 = PRED_IS_V ? (ok0 ? __longlong_as_double((long long)v0) : __builtin_nan("")) : pv[u].x;
                const double p1 = PRED_IS_V ? (ok1 ? __longlong_as_double((long long)v1) : __builtin_nan("")) : pv[u].y;
                if (nfull) VNM_HOT_LOAD(u, nbase, nxt);
                if (!HAS_PRED || cmp_apply<double>(op, p0, thr)) {
                    int slot = hot_slot(lkey, S, smask, &s_fill, k.x, spread);
                    if (slot >= 0) {
                        hot_accumulate<SIMPLE>(a, lacc, stride, slot, v0, ok0);
                        if (TWO) pa_accumulate_col(a.hot_wpack2, a.hot_vtype2, lacc, stride, slot, w0, a.hot_comp);
                    } else if (FROM_ENT) hot_entry_to_table(a, k.x, v0, &s_new);  // no columns to re-read: merge right here
                    else sat0 |= 1u << u;
                }
                if (!HAS_PRED || cmp_apply<double>(op, p1, thr)) {
                    int slot = hot_slot(lkey, S, smask, &s_fill, k.y, spread);
                    if (slot >= 0) {
                        hot_accumulate<SIMPLE>(a, lacc, stride, slot, v1, ok1);
                        if (TWO) pa_accumulate_col(a.hot_wpack2, a.hot_vtype2, lacc, stride, slot, w1, a.hot_comp);
                    } else if (FROM_ENT) hot_entry_to_table(a, k.y, v1, &s_new);
                    else sat1 |= 1u << u;
                }
            }
            have = nfull;
#undef VNM_HOT_LOAD
        } else {
            have = false;
            for (int u = 0; u < HOT_UNROLL; u++)
                for (int e = 0; e < 2; e++) {
                    int64_t r = base + (int64_t)u * 2 * AGG_BLOCK + e;
                    if (r >= cur.nrows) continue;
                    const uint64_t vb = FROM_ENT ? a.ent[r].y : (HAS_VAL ? (a.has_expr ? (uint64_t)__double_as_longlong(expr_eval1(a.expr, r)) : vp[r]) : 0);
                    const uint64_t kb = FROM_ENT ? a.ent[r].x : kp[r];
                    const bool ok = !VNULL || ((vbm[(voff + r) >> 3] >> ((voff + r) & 7)) & 1);
                    const double p = PRED_IS_V ? (ok ? __longlong_as_double((long long)vb) : __builtin_nan("")) : (HAS_PRED ? pp[r] : 0.0);
                    if (HAS_PRED && !cmp_apply<double>(op, p, thr)) continue;
                    int slot = hot_slot(lkey, S, smask, &s_fill, kb, spread);
                    if (slot >= 0) {
                        hot_accumulate<SIMPLE>(a, lacc, stride, slot, vb, ok);
                        if (TWO) pa_accumulate_col(a.hot_wpack2, a.hot_vtype2, lacc, stride, slot, vp2[r], a.hot_comp);
                    }
                    else if (FROM_ENT) hot_entry_to_table(a, kb, vb, &s_new);
                    else if (e == 0) sat0 |= 1u << u;
                    else sat1 |= 1u << u;
                }
        }
        // saturated keys: straight to the HBM table, out of line (rows base + e + u * 2 * AGG_BLOCK)
        if (!FROM_ENT && sat0) agg_rows_to_table(a, base, 2 * AGG_BLOCK, sat0, &s_new, nseg ? kp : nullptr, vp);
        if (!FROM_ENT && sat1) agg_rows_to_table(a, base + 1, 2 * AGG_BLOCK, sat1, &s_new, nseg ? kp : nullptr, vp);
        __syncthreads();
        const unsigned fill_now = s_fill;
        need_check = fill_now > (unsigned)S / 2;
        // key copies (see hot_slot): on after the first tile when it found a handful of groups, off for good once the
        // table holds more than that would explain
        if (spread_state == 0) { spread_state = fill_now <= (unsigned)S / 128 ? 1 : 2; if (spread_state == 1) spread = tid & 7u; }
        else if (spread_state == 1 && fill_now > (unsigned)S / 8) { spread_state = 2; spread = 0; }
        if (fill_now > flush_at) {
            lds_flush(a, lkey, lacc, S, tid, AGG_BLOCK, &s_new);
            __syncthreads();
            if (tid == 0) s_fill = 0;
        }
    }
    lds_flush(a, lkey, lacc, S, tid, AGG_BLOCK, &s_new);
    __syncthreads();
    if (tid == 0) { fold_new(a.g, &s_new); a.progress[blockIdx.x] = it; }
}

// Kernel 1n: the hot shape over THREE to SIX input columns (round 4) -- `SELECT k, sum(a), sum(b), avg(c), ..., count(*) GROUP BY k`
// over few groups, the shape of most reporting queries.  Every function in {COUNT(*), COUNT, SUM, AVG}, plain float64 columns, a
// plain float64 predicate column (one of the inputs or another) or none.  The generic scan (agg_lds_kernel) interprets one
// accumulator op at a time with 8-byte loads and ran C = 3 / 4 / 6 columns at 3.7 / 3.8 / 2.8 TB/s (G = 7); here, as in
// agg_hot_kernel: 16-byte loads of row pairs, the next tile's loads issued into the registers of the chunk just consumed,
// straight-line accumulation -- and ONE count atomic per row whatever the number of COUNT / AVG functions: without NULLs every
// count is the row count, the other count words are copied from it before a flush (hn_fill_counts).
template <int NC, bool PLAIN>   // PLAIN: float64 columns under sums and counts only (no integer sums, no MIN / MAX: their tests cost the plain case 5-13 %)
__device__ __forceinline__ void hn_accumulate(const AggArgs& a, uint64_t* lacc, int stride, int slot, const uint64_t* v) {
    __hip_atomic_fetch_add(&lacc[a.hn_w_base * stride + slot], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int w = a.hn_w_sum[c];
        if (w >= 0) {
            const int ct = PLAIN ? VNM_F64 : a.hn_ctype[c];
            const double x = ct == VNM_F64 ? __longlong_as_double((long long)v[c]) : (ct == VNM_U64 ? (double)v[c] : (double)(int64_t)v[c]);
            if (a.hot_comp) l_add_f64c(&lacc[w * stride + slot], stride, x);
            else __hip_atomic_fetch_add((double*)&lacc[w * stride + slot], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (!PLAIN && a.hn_any_int) {   // int64 / uint64 sums: the 128-bit sum's 32-bit lanes (agg_funcs.h:366-389: decimal128 on overflow), or the plain 64-bit word
            const int* iw = a.hn_iw[c];
            if (iw[0] >= 0) __hip_atomic_fetch_add(&lacc[iw[0] * stride + slot], v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (iw[1] >= 0) __hip_atomic_fetch_add(&lacc[iw[1] * stride + slot], v[c] & 0xFFFFFFFFULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (iw[2] >= 0) __hip_atomic_fetch_add(&lacc[iw[2] * stride + slot], (uint64_t)((int64_t)v[c] >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (iw[3] >= 0) __hip_atomic_fetch_add(&lacc[iw[3] * stride + slot], v[c] >> 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (!PLAIN && a.hn_any_mm) {    // MIN / MAX: the total-order code of the value (MinMaxFunc, agg_funcs.h:164-216)
            const int ct = a.hn_ctype[c];
            const uint64_t e = ct == VNM_F64 ? enc_f64(__longlong_as_double((long long)v[c])) : (ct == VNM_U64 ? v[c] : enc_i64((int64_t)v[c]));
            if (a.hn_wmm[c][0] >= 0) __hip_atomic_fetch_min(&lacc[a.hn_wmm[c][0] * stride + slot], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (a.hn_wmm[c][1] >= 0) __hip_atomic_fetch_max(&lacc[a.hn_wmm[c][1] * stride + slot], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}
__device__ __forceinline__ void hn_fill_counts(const AggArgs& a, uint64_t* lacc, int stride, int tid) {
    for (int i = tid; i < stride; i += AGG_BLOCK) {
        const uint64_t n = lacc[a.hn_w_base * stride + i];
        for (int j = 0; j < a.hn_n_copy; j++) lacc[a.hn_w_copy[j] * stride + i] = n;
    }
    __syncthreads();
}
template <int NC, bool HAS_PRED, bool PLAIN>
__global__ __launch_bounds__(AGG_BLOCK) void agg_hotn_kernel(AggArgs a) {
    constexpr int U = NC == 3 && !HAS_PRED ? 4 : 2;   // row pairs per lane and tile: 16 + 16 NC (8 + 8 NC) registers of loads in flight (three columns
                                                      // and a predicate with U = 4: 74 spilled VGPRs under the 128 a 1024-thread workgroup may hold)
    constexpr int TILE = AGG_BLOCK * 2 * U;
    extern __shared__ uint64_t lds[];
    __shared__ unsigned s_fill, s_new;
    __shared__ int s_go;
    const int S = a.lds_slots;
    const int stride = S + 2;
    const int W = a.plan.n_words;
    uint64_t* lkey = lds;
    uint64_t* lacc = lds + stride;
    const int tid = threadIdx.x;

    for (int i = tid; i < stride; i += AGG_BLOCK) lkey[i] = EMPTY;
    for (int w = 0; w < W; w++) {
        uint64_t init = merge_init(a.plan.merge[w]);
        for (int i = tid; i < stride; i += AGG_BLOCK) lacc[w * stride + i] = init;
    }
    if (tid == 0) { s_fill = 0; s_new = 0; }
    __syncthreads();

    const unsigned flush_at = (unsigned)(S * 6 / 10);
    const uint32_t smask = (uint32_t)S - 1;
    const uint64_t* kp = (const uint64_t*)a.keys[0].values + a.keys[0].offset;
    const uint64_t* vp[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) vp[c] = (const uint64_t*)a.cols[c].values + a.cols[c].offset;
    const int pc = HAS_PRED ? a.hn_pred_col : -1;
    const double* pp = HAS_PRED && pc < 0 ? (const double*)a.pred.values + a.pred.offset : (const double*)kp;
    const int op = a.p.op;
    const double thr = a.p.dval;

    ulonglong2 kk[U], vv[U][NC];
    double2 pv[U];
    bool have = false;
    uint32_t spread = 0;
    int spread_state = ((a.debug & 4) || a.ntiles < 64 * (int64_t)gridDim.x) ? 2 : 0;   // see agg_hot_kernel
    bool need_check = true;
    unsigned it = a.progress[blockIdx.x];
    for (;; it++) {
        const int64_t tile = (int64_t)blockIdx.x + (int64_t)it * gridDim.x;
        if (tile >= a.ntiles) break;
        if (need_check) {  // see agg_lds_kernel
            if (tid == 0) s_go = table_has_room(a, &s_new) ? 1 : 0;
            __syncthreads();
            if (!s_go) break;
        }
        const int64_t base = tile * TILE + 2 * tid;
        uint32_t sat0 = 0, sat1 = 0;
        if (base + (int64_t)(U - 1) * 2 * AGG_BLOCK + 1 < a.nrows) {
#define VNM_HN_LOAD(u, b)                                                                                       \
    do {                                                                                                       \
        const int64_t r_ = (b) + (int64_t)(u) * 2 * AGG_BLOCK;                                                 \
        kk[u] = *(const ulonglong2*)(kp + r_);                                                                 \
        _Pragma("unroll") for (int c_ = 0; c_ < NC; c_++) vv[u][c_] = *(const ulonglong2*)(vp[c_] + r_);       \
        if (HAS_PRED && pc < 0) pv[u] = *(const double2*)(pp + r_);                                            \
    } while (0)
            if (!have) {
#pragma unroll
                for (int u = 0; u < U; u++) VNM_HN_LOAD(u, base);
            }
            const int64_t nbase = base + (int64_t)gridDim.x * TILE;
            const bool nfull = tile + gridDim.x < a.ntiles && nbase + (int64_t)(U - 1) * 2 * AGG_BLOCK + 1 < a.nrows;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const ulonglong2 k = kk[u];
                uint64_t v0[NC], v1[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) { v0[c] = vv[u][c].x; v1[c] = vv[u][c].y; }
                double p0 = 0.0, p1 = 0.0;
                if (HAS_PRED) {
                    p0 = pv[u].x; p1 = pv[u].y;
#pragma unroll
                    for (int c = 0; c < NC; c++)
                        if (c == pc) { p0 = __longlong_as_double((long long)v0[c]); p1 = __longlong_as_double((long long)v1[c]); }
                }
                if (nfull) VNM_HN_LOAD(u, nbase);
                if (!HAS_PRED || cmp_apply<double>(op, p0, thr)) {
                    const int slot = hot_slot(lkey, S, smask, &s_fill, k.x, spread);
                    if (slot >= 0) hn_accumulate<NC, PLAIN>(a, lacc, stride, slot, v0);
                    else sat0 |= 1u << u;
                }
                if (!HAS_PRED || cmp_apply<double>(op, p1, thr)) {
                    const int slot = hot_slot(lkey, S, smask, &s_fill, k.y, spread);
                    if (slot >= 0) hn_accumulate<NC, PLAIN>(a, lacc, stride, slot, v1);
                    else sat1 |= 1u << u;
                }
            }
            have = nfull;
#undef VNM_HN_LOAD
        } else {
            have = false;
            for (int u = 0; u < U; u++)
                for (int e = 0; e < 2; e++) {
                    const int64_t r = base + (int64_t)u * 2 * AGG_BLOCK + e;
                    if (r >= a.nrows) continue;
                    uint64_t v[NC];
#pragma unroll
                    for (int c = 0; c < NC; c++) v[c] = vp[c][r];
                    if (HAS_PRED) {
                        double p = pc < 0 ? pp[r] : 0.0;
#pragma unroll
                        for (int c = 0; c < NC; c++) if (c == pc) p = __longlong_as_double((long long)v[c]);
                        if (!cmp_apply<double>(op, p, thr)) continue;
                    }
                    const int slot = hot_slot(lkey, S, smask, &s_fill, kp[r], spread);
                    if (slot >= 0) hn_accumulate<NC, PLAIN>(a, lacc, stride, slot, v);
                    else if (e == 0) sat0 |= 1u << u;
                    else sat1 |= 1u << u;
                }
        }
        // saturated keys: straight to the HBM table, out of line (rows base + e + u * 2 * AGG_BLOCK)
        if (sat0) agg_rows_to_table(a, base, 2 * AGG_BLOCK, sat0, &s_new);
        if (sat1) agg_rows_to_table(a, base + 1, 2 * AGG_BLOCK, sat1, &s_new);
        __syncthreads();
        const unsigned fill_now = s_fill;
        need_check = fill_now > (unsigned)S / 2;
        // key copies (see hot_slot): with 2 NC + 1 atomics per row the accumulator addresses of a handful of groups are the bottleneck, and
        // the sweet spot is ~50-110 (group, copy) addresses per word -- a wave's worth; more addresses cost more than they spread
        // (ms per 5e8 rows, none / 8 / 16 / 32 copies: four columns G = 3: 6.07 / 3.38 / 3.73 / 3.90, G = 7: 4.56 / 3.81 / 3.41 / -,
        // G = 30: 3.51 / 4.03; six columns G = 3: 8.49 / 5.53 / 5.21 / 6.91, G = 7: 6.26 / 5.04 / 4.73 / -)
        if (spread_state == 0) {
            const unsigned cap = (a.debug & 8) ? 8u : 16u;   // (VNM_AGG_DEBUG & 8: at most 8 copies, measurement)
            unsigned copies = fill_now <= 8 ? 16u : fill_now <= 12 ? 8u : 1u;
            if (copies > cap) copies = cap;
            if (fill_now * copies > (unsigned)S / 4) copies = 1;
            spread_state = copies > 1 ? 1 : 2;
            spread = tid & (copies - 1);
        } else if (spread_state == 1 && fill_now > (unsigned)S / 3) { spread_state = 2; spread = 0; }
        if (fill_now > flush_at) {
            hn_fill_counts(a, lacc, stride, tid);
            lds_flush(a, lkey, lacc, S, tid, AGG_BLOCK, &s_new);
            __syncthreads();
            if (tid == 0) s_fill = 0;
        }
    }
    __syncthreads();
    hn_fill_counts(a, lacc, stride, tid);
    lds_flush(a, lkey, lacc, S, tid, AGG_BLOCK, &s_new);
    __syncthreads();
    if (tid == 0) { fold_new(a.g, &s_new); a.progress[blockIdx.x] = it; }
}

// =======================================================================================================
// Kernel 2: wide (multi-column) keys.  Key words = n_keys values (NULL -> 0) + 1 null-mask word, which is
// exactly IntKeyValue equality (multi_numerical_hash_aggregate.h:11-18).  Rows go to the HBM table.
// =======================================================================================================
__global__ __launch_bounds__(256) void agg_wide_kernel(AggArgs a) {
    __shared__ int64_t s_tile;
    __shared__ unsigned s_new;
    if (threadIdx.x == 0) s_new = 0;
    __syncthreads();
    const int tid = threadIdx.x;
    const int nk = a.plan.n_keys;
    unsigned it = a.progress[blockIdx.x];
    for (;; it++) {
        const int64_t tile = (int64_t)blockIdx.x + (int64_t)it * gridDim.x;
        if (tile >= a.ntiles) break;
        __syncthreads();
        if (tid == 0) s_tile = table_has_room(a, &s_new) ? 1 : 0;
        __syncthreads();
        if (!s_tile) break;
        for (int r = 0; r < AGG_TILE / 256; r++) {
            const int64_t row = tile * AGG_TILE + (int64_t)r * 256 + tid;
            if (row >= a.nrows) continue;
            if (a.p.enabled && !pred_eval(a.p, a.pred, row)) continue;
            uint64_t kw[AGG_MAX_KEYS + 1];
            uint64_t nullmask = 0;
#pragma unroll
            for (int j = 0; j < AGG_MAX_KEYS; j++) {
                if (j < nk) {
                    bool ok = col_valid(a.keys[j], row);
                    kw[j] = ok ? col_key_bits(a.keys[j], row) : 0;
                    if (!ok) nullmask |= 1ULL << j;
                }
            }
            kw[nk] = nullmask;
            uint64_t gs = gt_find_wide(a.g, kw, wide_tag(kw, nk + 1), &s_new);
            for (int o = 0; o < a.plan.n_ops; o++) {
                const AccOp& op = a.plan.ops[o];
                uint64_t v;
                if (op_value(op, a.cols, row, &v)) g_merge(&a.g.acc[(uint64_t)op.word * a.g.stride + gs], a.plan.merge[op.word], v, (int64_t)a.g.stride);
            }
        }
    }
    __syncthreads();
    if (tid == 0) { fold_new(a.g, &s_new); a.progress[blockIdx.x] = it; }
}

// =======================================================================================================
// Kernel 2b (round 3): tuple dictionary.  Key sets too wide to pack into one word even as per-column dictionary codes
// used to aggregate in agg_wide_kernel, with one HBM atomic per row and accumulator word (150-570 ms per 1e9 rows).
// Instead a DICTIONARY maps the key tuple to a GROUP ID -- find-or-insert per row, the id is all that leaves the kernel --
// the ids go through the single-key operator like any 8-byte key (dense path, partitions, ...), and the tuples stored here
// are the result's key columns.
// Layout: ONE 64-byte line per slot (128 for more than five key columns): [tag, id + 1, key words ...] -- a probe reads
// one line (the wide-key table's column arrays: six lines per row, 200 ms per 1e9 rows at G = 2e7).  A reader issues
// all loads of the line at once; the claimer writes id and key words, drains, then publishes the tag, so a reader that saw
// the tag with stale words behind it simply reads those words again (they are ordered behind the tag then).
// Ids come from per-workgroup chunks of a global counter (an LDS atomic inside the claim, one global atomic per chunk:
// one atomic on ONE address per new group would serialise 2e7 of them) -- the id space has holes (< 2x), which nobody
// minds -- and they survive a rehash.
// =======================================================================================================
struct TDict {
    uint64_t* slot;                // [cap][sw]
    uint64_t cap;
    int kwt, sw;                   // key words per tuple (n_keys + 1), words per slot (8 or 16)
    unsigned long long* ctl;       // [1] a workgroup ran out of room  [2] fill
};
struct TupArgs {
    int nk;
    vnm_dcol keys[AGG_MAX_KEYS];
    Predicate p;
    vnm_dcol pred;
    int64_t nrows, ntiles, margin, fill_limit;
    TDict d;
    unsigned int* progress;
    uint64_t* out;
    unsigned long long* gnext;
};

// the line of one slot, all loads issued together
template <int KWT>
struct TLine { uint64_t t, g, k[KWT]; };
template <int KWT>
__device__ __forceinline__ void tdict_load(const TDict& d, uint64_t h, TLine<KWT>& l) {
    const uint64_t* base = d.slot + h * (uint64_t)d.sw;
    l.t = ld_agent(base);
    l.g = ld_agent(base + 1);
#pragma unroll
    for (int i = 0; i < KWT; i++) l.k[i] = ld_agent(base + 2 + i);
}

// `first`: the line of the tuple's home slot, loaded by the caller ahead of time (two rows of a lane are in flight together)
template <int KWT>
__device__ __forceinline__ uint64_t tdict_find(const TDict& d, const uint64_t* kw, uint64_t tagv, unsigned* newc, unsigned long long* s_gnext,
                                               const TLine<KWT>* first = nullptr) {
    const uint64_t mask = d.cap - 1;
    uint64_t h = (tagv ^ (tagv >> 29)) & mask;
    bool pre = first != nullptr;
    for (;;) {
        uint64_t* const base = d.slot + h * (uint64_t)d.sw;
        TLine<KWT> l;
        if (pre) l = *first; else tdict_load<KWT>(d, h, l);
        pre = false;
        const uint64_t t = l.t;
        uint64_t g = l.g;
        if (t == tagv) {
            bool eq = true;
#pragma unroll
            for (int i = 0; i < KWT; i++) eq = eq & (l.k[i] == kw[i]);
            if (!eq) {       // another tuple with this tag -- or this line's words were read before the tag's owner wrote them
                eq = true;
#pragma unroll
                for (int i = 0; i < KWT; i++) eq = eq & (ld_agent(base + 2 + i) == kw[i]);
            }
            if (eq) {
                if (g == EMPTY) g = ld_agent(base + 1);
                return g - 1;
            }
            h = (h + 1) & mask;
            continue;
        }
        if (t == EMPTY) {
            uint64_t expected = EMPTY;
            if (__hip_atomic_compare_exchange_strong(base, &expected, LOCKED, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                const uint64_t gid = atomicAdd(s_gnext, 1ULL);
                st_agent(base + 1, gid + 1);
#pragma unroll
                for (int i = 0; i < KWT; i++) st_agent(base + 2 + i, kw[i]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                st_agent(base, tagv);
                atomicAdd(newc, 1u);
                return gid;
            }
            continue;   // someone else is claiming this slot: look at it again
        }
        if (t == LOCKED) continue;
        h = (h + 1) & mask;
    }
}

template <int KWT>
__global__ __launch_bounds__(256) void tuple_gid_kernel(TupArgs a) {
    __shared__ int64_t s_tile;
    __shared__ unsigned s_new;
    __shared__ unsigned long long s_gnext, s_gend;
    if (threadIdx.x == 0) { s_new = 0; s_gnext = 0; s_gend = 0; }
    __syncthreads();
    const int tid = threadIdx.x;
    constexpr int NK = KWT - 1;
    unsigned it = a.progress[blockIdx.x];
    for (;; it++) {
        const int64_t tile = (int64_t)blockIdx.x + (int64_t)it * gridDim.x;
        if (tile >= a.ntiles) break;
        __syncthreads();
        if (tid == 0) {
            const unsigned v = atomicExch(&s_new, 0u);
            if (v) atomicAdd(&a.d.ctl[2], (unsigned long long)v);
            const unsigned long long fill = __hip_atomic_load(&a.d.ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool room = (int64_t)fill + a.margin <= a.fill_limit;
            if (!room) __hip_atomic_store(&a.d.ctl[1], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (room && s_gend - s_gnext < (unsigned long long)AGG_TILE) {   // every row of a tile may be a new group
                const unsigned long long base = atomicAdd(a.gnext, 2ULL * AGG_TILE);
                s_gnext = base; s_gend = base + 2ULL * AGG_TILE;
            }
            s_tile = room ? 1 : 0;
        }
        __syncthreads();
        if (!s_tile) break;
        // two rows of a lane at a time: both home lines are requested before either is looked at (a probe is one dependent
        // round trip to L2 / MALL / HBM; one row at a time left the lane idle for all of it)
        for (int r = 0; r < AGG_TILE / 256; r += 2) {
            uint64_t kw[2][KWT], tagv[2];
            TLine<KWT> line[2];
            bool live[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int64_t row = tile * AGG_TILE + (int64_t)(r + e) * 256 + tid;
                live[e] = row < a.nrows;
                if (live[e] && a.p.enabled && !pred_eval(a.p, a.pred, row)) { a.out[row] = 0; live[e] = false; }   // (the operator behind drops the row itself)
                if (!live[e]) continue;
                uint64_t nullmask = 0;
#pragma unroll
                for (int j = 0; j < NK; j++) {
                    const bool ok = col_valid(a.keys[j], row);
                    kw[e][j] = ok ? col_key_bits(a.keys[j], row) : 0;
                    if (!ok) nullmask |= 1ULL << j;
                }
                kw[e][NK] = nullmask;
                tagv[e] = wide_tag(kw[e], KWT);
                tdict_load<KWT>(a.d, (tagv[e] ^ (tagv[e] >> 29)) & (a.d.cap - 1), line[e]);
            }
#pragma unroll
            for (int e = 0; e < 2; e++) {
                if (!live[e]) continue;
                const int64_t row = tile * AGG_TILE + (int64_t)(r + e) * 256 + tid;
                a.out[row] = tdict_find<KWT>(a.d, kw[e], tagv[e], &s_new, &s_gnext, &line[e]);
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned v = atomicExch(&s_new, 0u);
        if (v) atomicAdd(&a.d.ctl[2], (unsigned long long)v);
        a.progress[blockIdx.x] = it;
    }
}

// a bigger dictionary: every slot moves as it is (the tuples are distinct: the first free slot of its probe sequence)
__global__ void tdict_rehash_kernel(TDict from, TDict to) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint64_t mask = to.cap - 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)from.cap; i += stride) {
        const uint64_t* src = from.slot + (uint64_t)i * from.sw;
        const uint64_t t = src[0];
        if (t == EMPTY || t == LOCKED) continue;
        uint64_t h = (t ^ (t >> 29)) & mask;
        for (;;) {
            uint64_t* dst = to.slot + h * (uint64_t)to.sw;
            uint64_t expected = EMPTY;
            if (__hip_atomic_compare_exchange_strong(dst, &expected, t, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                for (int w = 1; w < 2 + from.kwt; w++) dst[w] = src[w];
                break;
            }
            h = (h + 1) & mask;
        }
    }
}

// tuples given as key-word arrays (the finished groups of an operator that leaves its packed form): insert, hand back the ids
template <int KWT>
__global__ __launch_bounds__(256) void tuple_words_kernel(TDict d, const uint64_t* __restrict__ words /* [KWT][n] */, int64_t n, int64_t per_block,
                                                          uint64_t* __restrict__ out, unsigned long long* gnext) {
    __shared__ unsigned s_new;
    __shared__ unsigned long long s_gnext;
    const int64_t lo = (int64_t)blockIdx.x * per_block, hi = lo + per_block < n ? lo + per_block : n;
    if (threadIdx.x == 0) { s_new = 0; s_gnext = lo < hi ? atomicAdd(gnext, (unsigned long long)(hi - lo)) : 0ULL; }
    __syncthreads();
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        uint64_t kw[KWT];
#pragma unroll
        for (int j = 0; j < KWT; j++) kw[j] = words[(int64_t)j * n + i];
        out[i] = tdict_find<KWT>(d, kw, wide_tag(kw, KWT), &s_new, &s_gnext);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_new) atomicAdd(&d.ctl[2], (unsigned long long)s_new);
}

// group id -> slot (one 4-byte store per group; laying the tuples themselves out by id was four random 8-byte stores per group:
// 3.2 ms at 2e7 groups), then the result's key words straight from the slots: one line per group
__global__ __launch_bounds__(256) void tuple_sweep_kernel(TDict d, uint32_t* __restrict__ slot_of, int64_t ngid) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < (int64_t)d.cap; h += stride) {
        const ulonglong2 w = *(const ulonglong2*)(d.slot + (uint64_t)h * d.sw);   // tag, id + 1
        if (w.x == EMPTY || w.x == LOCKED) continue;
        const int64_t gid = (int64_t)w.y - 1;
        if (gid >= 0 && gid < ngid) slot_of[gid] = (uint32_t)h;
    }
}
__global__ void tuple_keys_kernel(TDict d, const uint32_t* __restrict__ slot_of, int64_t ngid, const uint64_t* __restrict__ gids, int64_t n,
                                  uint64_t* __restrict__ dkey, int64_t dstride) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t gid = (int64_t)gids[i];
        const uint64_t* src = gid < ngid ? d.slot + (uint64_t)slot_of[gid] * d.sw + 2 : nullptr;
        for (int j = 0; j < d.kwt; j++) dkey[(int64_t)j * dstride + i] = src ? src[j] : 0;
    }
}

// =======================================================================================================
// Kernel 3: no GROUP BY.  Per-lane private accumulators in LDS (no atomics, no conflicts), block tree
// reduction, one agent-scope atomic per word per block into group slot 0.
// =======================================================================================================
__device__ __forceinline__ uint64_t merge_vals(int mk, uint64_t x, uint64_t y) {
    switch (mk) {
        case M_ADD_U64: return x + y;
        case M_ADD_F64:
        case M_ADD_F64C: return (uint64_t)__double_as_longlong(__longlong_as_double((long long)x) + __longlong_as_double((long long)y));
        case M_MIN_U64: return x < y ? x : y;
        default: return x > y ? x : y;
    }
}

constexpr int OG_R = 8;  // rows per lane and tile: loads of one column are issued together, one op decode serves all
__global__ __launch_bounds__(OG_BLOCK) void agg_onegroup_kernel(AggArgs a) {
    extern __shared__ uint64_t lds[];  // [W][OG_BLOCK]: per-lane partial accumulators
    const int tid = threadIdx.x;
    const int W = a.plan.n_words;
    for (int w = 0; w < W; w++) lds[w * OG_BLOCK + tid] = merge_init(a.plan.merge[w]);
    const int64_t ntiles = (a.nrows + (int64_t)OG_BLOCK * OG_R - 1) / ((int64_t)OG_BLOCK * OG_R);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * OG_BLOCK * OG_R + tid;
        int64_t rowc[OG_R];
        uint32_t pass = 0;
#pragma unroll
        for (int r = 0; r < OG_R; r++) {
            const int64_t row = row0 + (int64_t)r * OG_BLOCK;
            if (row < a.nrows) pass |= 1u << r;
            rowc[r] = row < a.nrows ? row : a.nrows - 1;
        }
        if (a.p.enabled) {
            uint64_t praw[OG_R];
            load_raw_rows<OG_R>(a.pred, rowc, praw);
            const uint32_t pvalid = load_valid_rows<OG_R>(a.pred, rowc);
#pragma unroll
            for (int r = 0; r < OG_R; r++)
                if (!pred_eval_raw(a.p, a.pred.type, praw[r], (pvalid >> r) & 1u)) pass &= ~(1u << r);
        }
        uint64_t raw[OG_R];
        uint32_t cvalid = 0;
        int loaded_col = -1;  // consecutive ops on one column share its loads
        for (int o = 0; o < a.plan.n_ops; o++) {
            const AccOp op = a.plan.ops[o];
            const int mk = a.plan.merge[op.word];
            uint32_t have = pass;
            int vtype = VNM_U64;
            if (op.kind != A_COUNT_ROWS) {
                const vnm_dcol& c = a.cols[op.col];
                vtype = c.type;
                if (op.col != loaded_col) {
                    load_raw_rows<OG_R>(c, rowc, raw);
                    cvalid = load_valid_rows<OG_R>(c, rowc);
                    loaded_col = op.col;
                }
                have &= cvalid;
            }
            uint64_t acc = lds[op.word * OG_BLOCK + tid];
            if (mk == M_ADD_F64C) {  // compensated float64 sum: this lane's (hi, lo) pair
                double hi = __longlong_as_double((long long)acc), lo = __longlong_as_double((long long)lds[(op.word + 1) * OG_BLOCK + tid]);
#pragma unroll
                for (int r = 0; r < OG_R; r++)
                    if ((have >> r) & 1u) {
                        const double x = __longlong_as_double((long long)op_value_raw(op.kind, vtype, raw[r]));
                        const double sm = hi + x;
                        lo += two_sum_err(hi, x, sm);
                        hi = sm;
                    }
                lds[op.word * OG_BLOCK + tid] = (uint64_t)__double_as_longlong(hi);
                lds[(op.word + 1) * OG_BLOCK + tid] = (uint64_t)__double_as_longlong(lo);
                continue;
            }
#pragma unroll
            for (int r = 0; r < OG_R; r++)
                if ((have >> r) & 1u) acc = merge_vals(mk, acc, op_value_raw(op.kind, vtype, op.kind == A_COUNT_ROWS ? 0 : raw[r]));
            lds[op.word * OG_BLOCK + tid] = acc;
        }
    }
    __syncthreads();
    for (int half = OG_BLOCK / 2; half > 0; half >>= 1) {
        if (tid < half)
            for (int w = 0; w < W; w++) {
                if (a.plan.merge[w] == M_ADD_F64C) {  // (hi, lo) + (hi, lo): the error of hi + hi joins lo (word w + 1, merged next)
                    const double x = __longlong_as_double((long long)lds[w * OG_BLOCK + tid]), y = __longlong_as_double((long long)lds[w * OG_BLOCK + tid + half]);
                    const double sm = x + y;
                    lds[w * OG_BLOCK + tid] = (uint64_t)__double_as_longlong(sm);
                    lds[(w + 1) * OG_BLOCK + tid] = (uint64_t)__double_as_longlong(__longlong_as_double((long long)lds[(w + 1) * OG_BLOCK + tid]) + two_sum_err(x, y, sm));
                    continue;
                }
                lds[w * OG_BLOCK + tid] = merge_vals(a.plan.merge[w], lds[w * OG_BLOCK + tid], lds[w * OG_BLOCK + tid + half]);
            }
        __syncthreads();
    }
    if (tid < W) {
        int mk = a.plan.merge[tid];
        uint64_t v = lds[tid * OG_BLOCK];
        if (v != merge_init(mk)) g_merge(&a.g.acc[(uint64_t)tid * a.g.stride], mk, v, (int64_t)a.g.stride);
    }
}

// No GROUP BY over at most one 8-byte column without NULLs (float64 predicate column or none): every accumulator kind
// is kept in registers for every row (a few VALU ops each, far below the load time) and the plan only decides which of
// them are written at the end -- no per-op dispatch, 16-byte loads, four pairs in flight per lane.
// VT: VNM_F64 / VNM_I64 / VNM_U64, or -1 = no input column (COUNT(*) only).  PM: 0 no predicate, 1 the predicate column
// is the input column, 2 a separate float64 predicate column.
template <int VT, int PM>
__global__ __launch_bounds__(OG_BLOCK) void agg_onegroup_hot_kernel(AggArgs a) {
    __shared__ uint64_t part[OG_BLOCK / 64][9];
    const int tid = threadIdx.x;
    const ulonglong2* vp = VT >= 0 ? (const ulonglong2*)((const uint64_t*)a.cols[0].values + a.cols[0].offset) : nullptr;
    const double2* pp = PM == 2 ? (const double2*)((const double*)a.pred.values + a.pred.offset) : nullptr;
    const int op = a.p.op;
    const double thr = a.p.dval;
    double sf = -0.0, sc = 0.0;  // float64 sum (from the additive identity -0.0, see merge_init) and the accumulated rounding errors of its adds (see M_ADD_F64C)
    uint64_t si = 0, slo = 0, shis = 0, shiu = 0, cnt = 0, mn = ~0ULL, mx = 0;
    auto take = [&](uint64_t vb, double pv) {
        const double f = VT == VNM_F64 ? __longlong_as_double((long long)vb) : (VT == VNM_U64 ? (double)vb : (double)(int64_t)vb);
        if (PM != 0 && !cmp_apply<double>(op, PM == 1 ? f : pv, thr)) return;
        cnt++;
        if (VT < 0) return;
        { const double sm = sf + f; sc += two_sum_err(sf, f, sm); sf = sm; }
        if (VT != VNM_F64) {
            si += vb;
            slo += vb & 0xFFFFFFFFULL;
            shis += (uint64_t)((int64_t)vb >> 32);
            shiu += vb >> 32;
        }
        const uint64_t e = VT == VNM_F64 ? enc_f64(f) : (VT == VNM_U64 ? vb : enc_i64((int64_t)vb));
        mn = e < mn ? e : mn;
        mx = e > mx ? e : mx;
    };
    constexpr int U = 4;
    const int64_t npairs = a.nrows >> 1;
    const int64_t stride = (int64_t)gridDim.x * OG_BLOCK;
    for (int64_t base = (int64_t)blockIdx.x * OG_BLOCK + tid; base < npairs; base += stride * U) {
        ulonglong2 v[U];
        double2 p[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + (int64_t)u * stride;
            v[u] = make_ulonglong2(0, 0);
            p[u] = make_double2(0.0, 0.0);
            if (i < npairs) {
                if (VT >= 0) {
                    if (a.has_expr) { const double2 ev_ = expr_eval2(a.expr, 2 * i); v[u].x = (unsigned long long)__double_as_longlong(ev_.x); v[u].y = (unsigned long long)__double_as_longlong(ev_.y); }
                    else v[u] = vp[i];
                }
                if (PM == 2) p[u] = pp[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (base + (int64_t)u * stride < npairs) { take(v[u].x, p[u].x); take(v[u].y, p[u].y); }
        }
    }
    if ((a.nrows & 1) && blockIdx.x == 0 && tid == 0) {  // the odd last row
        const int64_t r = a.nrows - 1;
        take(VT >= 0 ? (a.has_expr ? (uint64_t)__double_as_longlong(expr_eval1(a.expr, r)) : ((const uint64_t*)vp)[r]) : 0, PM == 2 ? ((const double*)pp)[r] : 0.0);
    }
    // wave reduction, then one merge per needed word and workgroup
    uint64_t w8[9] = {(uint64_t)__double_as_longlong(sf), si, slo, shis, shiu, cnt, mn, mx, (uint64_t)__double_as_longlong(sc)};
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double mine = __longlong_as_double((long long)w8[0]);
        const double of = __shfl_xor(mine, o), oc = __shfl_xor(__longlong_as_double((long long)w8[8]), o);
        const double sm = mine + of;
        w8[0] = (uint64_t)__double_as_longlong(sm);
        w8[8] = (uint64_t)__double_as_longlong(__longlong_as_double((long long)w8[8]) + oc + two_sum_err(mine, of, sm));
#pragma unroll
        for (int k = 1; k < 6; k++) w8[k] += __shfl_xor(w8[k], o);
        const uint64_t omn = __shfl_xor(w8[6], o), omx = __shfl_xor(w8[7], o);
        w8[6] = omn < w8[6] ? omn : w8[6];
        w8[7] = omx > w8[7] ? omx : w8[7];
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 9; k++) part[tid >> 6][k] = w8[k];
    }
    __syncthreads();
    if (tid <= A_MAX && a.hot_w[tid] >= 0) {
        // AccKind -> register: COUNT(*) and COUNT both see `cnt` (no NULLs on this path)
        const int src = tid == A_COUNT_ROWS || tid == A_COUNT_VALID ? 5 : (tid == A_SUM_F64 ? 0 : (tid == A_MIN ? 6 : (tid == A_MAX ? 7 : tid - A_SUM_I64 + 1)));
        uint64_t v = part[0][src];
        double comp = __longlong_as_double((long long)part[0][8]);
        for (int wv = 1; wv < OG_BLOCK / 64; wv++) {
            const uint64_t o = part[wv][src];
            if (src == 0) {
                const double x = __longlong_as_double((long long)v), y = __longlong_as_double((long long)o), sm = x + y;
                comp += __longlong_as_double((long long)part[wv][8]) + two_sum_err(x, y, sm);
                v = (uint64_t)__double_as_longlong(sm);
            }
            else if (src == 6) v = o < v ? o : v;
            else if (src == 7) v = o > v ? o : v;
            else v += o;
        }
        const int w = a.hot_w[tid];
        const int mk = a.plan.merge[w];
        if (src == 0 && mk == M_ADD_F64) v = (uint64_t)__double_as_longlong(fsum2(__longlong_as_double((long long)v), comp));
        if (v != merge_init(mk) || mk == M_ADD_F64 || mk == M_ADD_F64C) g_merge(&a.g.acc[(uint64_t)w * a.g.stride], mk, v, (int64_t)a.g.stride);
        if (src == 0 && mk == M_ADD_F64C && comp != 0.0)
            __hip_atomic_fetch_add((double*)&a.g.acc[(uint64_t)(w + 1) * a.g.stride], comp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// =======================================================================================================
// Merge dense partial groups (another rank's run, or the old table during growth) into the table.
// src_tag != NULL: source is a table (skip EMPTY / use tag as key).  Otherwise dense run: key words
// (n_keys values + null mask) and accumulator words, each an array of n entries.
// =======================================================================================================
struct MergeArgs {
    AggPlan plan;
    GTable g;
    int64_t n;
    const uint64_t* src_tag;       // table source (single path: key, wide: tag) or NULL
    const uint64_t* src_key[AGG_MAX_KEYS + 1];
    const uint64_t* src_acc[AGG_MAX_WORDS];
    int src_is_table;
    int64_t src_cap;  // table source: entries [src_cap], [src_cap+1] are the special groups
    int64_t src_stride;  // element stride of the dense source arrays (1 = SoA, n_words = row-major rows)
    // row-major rows in BLOCKS of (blk_rows + 1) rows whose first row is a header (word 0 = how many of the block's rows are groups):
    // what the one-collective small-G exchange delivers (vnm_agg_merge_row_blocks); 0 = plain rows
    int64_t blk_rows;
    const uint64_t* blk_base;
};

__global__ __launch_bounds__(256) void agg_merge_kernel(MergeArgs m) {
    __shared__ unsigned s_new;
    if (threadIdx.x == 0) s_new = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool single = m.g.kwt == 0;
    const int nk = m.plan.n_keys;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < m.n; i0 += stride) {
        uint64_t slot;
        int64_t i = i0;
        if (m.blk_rows) {   // i0 = (block, row of the block): skip what lies beyond the block's count, step over the headers
            const int64_t b = i0 / m.blk_rows, j = i0 % m.blk_rows;
            if ((uint64_t)j >= m.blk_base[b * (m.blk_rows + 1) * m.src_stride]) continue;
            i = b * (m.blk_rows + 1) + 1 + j;
        }
        if (m.src_is_table) {
            uint64_t t = m.src_tag[i];
            if (t == EMPTY || (!single && t == LOCKED)) continue;  // LOCKED (~0 - 1) is an ordinary key on the single path
            if (single) {
                if (i >= m.src_cap) {
                    slot = m.g.cap + (uint64_t)(i - m.src_cap);
                    if (ld_agent(&m.g.tag[slot]) == EMPTY) st_agent(&m.g.tag[slot], 0);
                } else slot = gt_find_single(m.g, t, &s_new);
            } else {
                uint64_t kw[AGG_MAX_KEYS + 1];
#pragma unroll
                for (int j = 0; j <= AGG_MAX_KEYS; j++) if (j <= nk) kw[j] = m.src_key[j][i];
                slot = gt_find_wide(m.g, kw, t, &s_new);
            }
        } else if (nk == 0) {
            slot = 0;
        } else if (single) {
            uint64_t key = m.src_key[0][i * m.src_stride], nullmask = m.src_key[1][i * m.src_stride];
            if (nullmask) { slot = m.g.cap + 1; if (ld_agent(&m.g.tag[slot]) == EMPTY) st_agent(&m.g.tag[slot], 0); }
            else if (key == EMPTY) { slot = m.g.cap; if (ld_agent(&m.g.tag[slot]) == EMPTY) st_agent(&m.g.tag[slot], 0); }
            else slot = gt_find_single(m.g, key, &s_new);
        } else {
            uint64_t kw[AGG_MAX_KEYS + 1];
#pragma unroll
            for (int j = 0; j <= AGG_MAX_KEYS; j++) if (j <= nk) kw[j] = m.src_key[j][i * m.src_stride];
            slot = gt_find_wide(m.g, kw, wide_tag(kw, nk + 1), &s_new);
        }
        for (int w = 0; w < m.plan.n_words; w++) {
            uint64_t v = m.src_acc[w][i * (m.src_is_table ? 1 : m.src_stride)];
            int mk = m.plan.merge[w];
            if (v != merge_init(mk)) g_merge(&m.g.acc[(uint64_t)w * m.g.stride + slot], mk, v, (int64_t)m.g.stride);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) fold_new(m.g, &s_new);
}

// =======================================================================================================
// Compaction of the table into a dense run: dkey[kw][n], dacc[W][n].  Order is unspecified (as in the
// reference, robin_hood iteration order) except that the NULL-key group comes last
// (single_numerical_hash_aggregate.cpp:58-60).
// =======================================================================================================
struct CompactArgs {
    AggPlan plan;
    GTable g;
    uint64_t* dkey;   // kw * dstride
    uint64_t* dacc;   // W * dstride
    int64_t dstride;
};

// One reservation per workgroup and 4096 slots (16 rounds x 4 waves: their 64 counts are scanned by one wave).  One per wave
// and round was an atomic on ONE address per 64 slots: a 2^26-slot table = 1e6 of them, 9.5 ms for 2e7 groups.
constexpr int CP_ROUNDS = 16;
__global__ __launch_bounds__(256) void agg_compact_kernel(CompactArgs c) {
    __shared__ unsigned int s_cnt[CP_ROUNDS * 4];
    __shared__ unsigned long long s_base;
    const bool single = c.g.kwt == 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t chunk = (int64_t)CP_ROUNDS * 256;
    const int64_t nchunks = ((int64_t)c.g.cap + chunk - 1) / chunk;
    const uint64_t lt = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        uint64_t ball[CP_ROUNDS];
#pragma unroll
        for (int r = 0; r < CP_ROUNDS; r++) {
            const int64_t i = ch * chunk + (int64_t)r * 256 + tid;
            bool occ = false;
            if (i < (int64_t)c.g.cap) {
                const uint64_t t = c.g.tag[i];
                occ = (t != EMPTY && (single || t != LOCKED));
            }
            ball[r] = __ballot(occ);
            if (lane == 0) s_cnt[r * 4 + wave] = (unsigned int)__popcll(ball[r]);
        }
        __syncthreads();
        if (tid < 64) {
            const unsigned int v = s_cnt[tid];
            unsigned int incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned int x = __shfl_up(incl, d);
                if (tid >= d) incl += x;
            }
            s_cnt[tid] = incl - v;
            if (tid == 63) s_base = incl ? atomicAdd(&c.g.ctl[3], (unsigned long long)incl) : 0ULL;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CP_ROUNDS; r++) {
            if (!((ball[r] >> lane) & 1ULL)) continue;
            const int64_t i = ch * chunk + (int64_t)r * 256 + tid;
            const int64_t pos = (int64_t)s_base + s_cnt[r * 4 + wave] + __popcll(ball[r] & lt);
            if (single) {
                c.dkey[pos] = c.g.tag[i];
                c.dkey[c.dstride + pos] = 0;
            } else {
                for (int j = 0; j < c.g.kwt; j++) c.dkey[(int64_t)j * c.dstride + pos] = c.g.keyw[(uint64_t)j * c.g.stride + i];
            }
            for (int w = 0; w < c.plan.n_words; w++) c.dacc[(int64_t)w * c.dstride + pos] = c.g.acc[(uint64_t)w * c.g.stride + i];
        }
        __syncthreads();
    }
}

// one thread: append the special groups (sentinel-key group, then the NULL-key group LAST)
__global__ void agg_compact_special_kernel(CompactArgs c) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (c.plan.n_keys == 0) {  // ONE_GROUP: exactly one row, always (one_group_aggregate.cpp:28-37)
        for (int w = 0; w < c.plan.n_words; w++) c.dacc[(int64_t)w * c.dstride] = c.g.acc[(uint64_t)w * c.g.stride];
        c.g.ctl[3] = 1;
        return;
    }
    if (c.g.kwt != 0) return;
    for (int s = 0; s < 2; s++) {
        uint64_t slot = c.g.cap + s;
        if (c.g.tag[slot] == EMPTY) continue;
        int64_t pos = (int64_t)c.g.ctl[3];
        c.dkey[pos] = s == 0 ? EMPTY : 0;
        c.dkey[c.dstride + pos] = s == 0 ? 0 : 1;
        for (int w = 0; w < c.plan.n_words; w++) c.dacc[(int64_t)w * c.dstride + pos] = c.g.acc[(uint64_t)w * c.g.stride + slot];
        c.g.ctl[3] = pos + 1;
    }
}


// -------------------------------------------------------------------------------------------------------
// A dense run (the partitioned path's G groups) next to an HBM table holding far fewer groups (spilled heavy keys and
// the odd entry of an over-full region): fold the TABLE into the RUN instead of inserting G groups into the table
// (G = 2e7: 9 ms of per-group CAS against one pass over the run's key words).  Every run row probes the table
// read-only (same hash and probe sequence as gt_find_single) and merges the slot's accumulator words into its own --
// run rows and table slots are unique, nothing contends -- marking the slot; slots no run row carried are appended
// behind the run (the caller checked the room).
// -------------------------------------------------------------------------------------------------------
struct PatchArgs {
    AggPlan plan;
    GTable g;
    uint64_t* rkey;         // [2][rstride]: key word, null word
    uint64_t* racc;         // [W][rstride]
    int64_t rn, rstride;
    uint8_t* found;         // [cap + 2]
    unsigned long long* appended;
    uint32_t* bloom;        // [PATCH_BLOOM_BITS / 32]: one bit per table key (a second hash)
};
// The table next to a big run mostly holds a handful of keys (spilled heavy keys, keys outside a sampled range) in millions of
// slots: every run row probing it is a random HBM / MALL access (8.8e7 rows: 1.8 ms).  One bit per table key in a 64 KB filter
// that every workgroup keeps in LDS sends only the rows that can match to the table.
constexpr int PATCH_BLOOM_BITS = 1 << 19;
__device__ __forceinline__ uint32_t patch_bloom_bit(uint64_t k) { return hash_u64(k ^ 0x9E3779B97F4A7C15ULL) & (PATCH_BLOOM_BITS - 1); }
__global__ __launch_bounds__(256) void run_patch_bloom_kernel(PatchArgs a) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x; slot < (int64_t)a.g.cap; slot += stride) {
        const uint64_t t = a.g.tag[slot];
        if (t == EMPTY) continue;
        const uint32_t b = patch_bloom_bit(t);
        atomicOr(&a.bloom[b >> 5], 1u << (b & 31));
    }
}
// (1024 threads share one copy of the filter: two workgroups = 32 waves per CU; with 256 threads it was 8 waves: 1.33 ms per 8.7e7 rows)
__global__ __launch_bounds__(1024) void run_patch_kernel(PatchArgs a) {
    __shared__ uint32_t lb[PATCH_BLOOM_BITS / 32];
    for (int i = threadIdx.x; i < PATCH_BLOOM_BITS / 32; i += 1024) lb[i] = a.bloom[i];
    __syncthreads();
    const uint64_t mask = a.g.cap - 1;
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < a.rn; i += stride) {
        const uint64_t k = a.rkey[i], nm = a.rkey[a.rstride + i];
        uint64_t slot = ~0ULL;
        if (nm) { if (a.g.tag[a.g.cap + 1] != EMPTY) slot = a.g.cap + 1; }
        else if (k == EMPTY) { if (a.g.tag[a.g.cap] != EMPTY) slot = a.g.cap; }
        else {
            const uint32_t bb = patch_bloom_bit(k);
            if (!((lb[bb >> 5] >> (bb & 31)) & 1u)) continue;
            uint64_t h = hash_u64(k) & mask;
            for (uint64_t probes = 0; probes <= mask; probes++) {
                const uint64_t t = a.g.tag[h];
                if (t == k) { slot = h; break; }
                if (t == EMPTY) break;
                h = (h + 1) & mask;
            }
        }
        if (slot == ~0ULL) continue;
        for (int w = 0; w < a.plan.n_words; w++) {
            const uint64_t v = a.g.acc[(uint64_t)w * a.g.stride + slot];
            const int mk = a.plan.merge[w];
            if (v != merge_init(mk)) g_merge(&a.racc[(int64_t)w * a.rstride + i], mk, v, a.rstride);
        }
        a.found[slot] = 1;
    }
}
__global__ __launch_bounds__(256) void run_patch_append_kernel(PatchArgs a) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t nslots = (int64_t)a.g.cap + 2;
    constexpr int U = 8;   // independent loads in flight (one slot per iteration: 0.51 ms for 2^23 slots, 150 GB/s)
    for (int64_t s0 = (int64_t)blockIdx.x * 256 + threadIdx.x; s0 < nslots; s0 += stride * U) {
      uint64_t tt[U]; uint8_t ff[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
          const int64_t sl = s0 + (int64_t)u * stride;
          tt[u] = EMPTY; ff[u] = 1;
          if (sl < nslots) { tt[u] = a.g.tag[sl]; ff[u] = a.found[sl]; }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int64_t slot = s0 + (int64_t)u * stride;
        const uint64_t t = tt[u];
        if (slot >= nslots || t == EMPTY || ff[u]) continue;
        const int64_t pos = a.rn + (int64_t)atomicAdd(a.appended, 1ULL);
        if (pos >= a.rstride) continue;   // cannot happen while the table's fill count holds (the host checks the total)
        a.rkey[pos] = slot < (int64_t)a.g.cap ? t : (slot == (int64_t)a.g.cap ? EMPTY : 0);
        a.rkey[a.rstride + pos] = slot == (int64_t)a.g.cap + 1 ? 1 : 0;
        for (int w = 0; w < a.plan.n_words; w++) a.racc[(int64_t)w * a.rstride + pos] = a.g.acc[(uint64_t)w * a.g.stride + slot];
      }
    }
}

// =======================================================================================================
// Radix-partitioned aggregation for LARGE group counts (hot shape only).
//
// When the groups do not fit an LDS table, per-row updates would have to go to the HBM table, and agent
// scope atomics cap at ~24 G/s (profiles/microbench_r01.txt) -- 285 ms for the 1e9-row / 1e8-group query.
// Instead the surviving (key, value) pairs are radix partitioned by hash bits, streaming and atomic-free:
//   pass 1  rows -> 256 partitions           (filter fused; LDS counting sort per 8192-row tile, runs of
//                                              consecutive 16-byte entries written per partition)
//   pass 2  each partition -> 512 sub-parts   (same kernel, next hash bits; only when G > ~280k)
//   pass 3  one workgroup per final partition aggregates it in an LDS table and appends dense groups.
// Every (partition, producer) pair owns a private output region, so no cursor is shared.
// Traffic: 16 N read + 16 sN written/read per level + 24 G' written.
// =======================================================================================================
constexpr int PT_BLOCK = 1024;
#ifndef VNM_PT_ITEMS
#define VNM_PT_ITEMS 8   // 8192-item tiles: one workgroup per CU, but write runs twice as long (measured 14.1 vs 15.3 ms at G=1e8)
#endif
constexpr int PT_ITEMS = VNM_PT_ITEMS;
constexpr int PT_PAIRS = PT_ITEMS / 2;
constexpr int PT_TILE = PT_BLOCK * PT_ITEMS;  // 8192 rows or entries per tile
constexpr int PT_MAXP = 512;
constexpr int PT_MAX_REGIONS = 256;  // input regions per pass-2 workgroup (keeps two workgroups per CU in LDS)
constexpr int PA_BLOCK = 512;
constexpr int PA_SLOTS = 2048;

struct PartArgs {
    // source A: raw columns (pass 1)
    const uint64_t* kp;
    const double* vp;
    const double* pp;
    int has_pred, pred_is_v, op;
    double thr;
    int64_t nrows;
    int has_expr;        // the value is `expr` evaluated per row pair instead of vp[row]
    ExprProg expr;
    // source B: entry regions written by the previous level (pass 2)
    const ulonglong2* in_entries;
    const uint32_t* in_counts;
    int64_t in_cap;
    int in_regions;      // regions per input partition
    int in_split;        // workgroups per input partition (each takes in_regions / in_split regions)
    // output regions: region id = out_base(blockIdx) + p * out_stride
    ulonglong2* out_entries;
    uint32_t* out_counts;
    int64_t out_cap;
    int nparts;          // 256 or 512
    int shift;           // partition = (hash >> shift) & (nparts - 1)
    unsigned long long* flags;  // [0] failure (spill buffer full)  [2] entries in the spill buffer
    // entries that do not fit their region (skewed keys) are appended here and aggregated by agg_entries_kernel
    ulonglong2* spill;
    int64_t spill_cap;
    int debug;
    // wide entries (part_scatter_wide_kernel): key + nval raw values per entry
    vnm_dcol vcols[6];   // input columns (any numeric type, NULLs allowed)
    int nval;
    int has_vmask;       // last entry word = validity bits of the input columns (some column has a bitmap)
    Predicate wp;        // generic predicate over wpred (any type)
    vnm_dcol wpred;  // timing experiments (VNM_PART_DEBUG): 1 = no copy-out stores, 2 = no staging / copy-out at all
};

template <bool FROM_ROWS>
__global__ __launch_bounds__(PT_BLOCK) void part_scatter_kernel(PartArgs a) {
    __shared__ ulonglong2 stage[PT_TILE];
    __shared__ uint16_t part_of[PT_TILE];
    __shared__ uint32_t cnt[PT_MAXP], off[PT_MAXP], cursor[PT_MAXP];
    __shared__ uint32_t s_total, s_spill;
    __shared__ unsigned long long s_spill_base;
    __shared__ uint32_t wtot[PT_MAXP / 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int np = a.nparts;
    const uint32_t pmask = (uint32_t)np - 1;
    const int npad = np < 64 ? 64 : np;  // scan width: whole waves (counts beyond np stay zero)
    for (int i = tid; i < PT_MAXP; i += PT_BLOCK) { cnt[i] = 0; cursor[i] = 0; }
    if (tid == 0) s_spill = 0;
    __syncthreads();

    // output region of partition p for this producer
    int64_t out_base, out_stride;
    int64_t ntiles = 0, src_first = 0;
    if (FROM_ROWS) {
        out_base = blockIdx.x; out_stride = gridDim.x;
        ntiles = (a.nrows + PT_TILE - 1) / PT_TILE;
    } else {
        const int pin = blockIdx.x / a.in_split, g = blockIdx.x % a.in_split;
        out_base = (int64_t)pin * np * a.in_split + g; out_stride = a.in_split;
        src_first = (int64_t)pin * a.in_regions;
    }

    // One tile = PT_TILE items.  get_item(k, &e) extracts item k of the CURRENT tile from registers;
    // prefetch_next() is called right after the last use of those registers (end of phase A) and issues the
    // loads of the NEXT tile into the same registers, so HBM reads stay in flight during phases B..E.
    auto process_tile = [&](auto&& get_item, auto&& prefetch_next) {
        // A: local rank inside the partition (LDS returning atomic)
        uint32_t myp[PT_ITEMS], myr[PT_ITEMS];
        ulonglong2 mye[PT_ITEMS];
#pragma unroll
        for (int k = 0; k < PT_ITEMS; k++) {
            myp[k] = 0xFFFFFFFFu;
            ulonglong2 e;
            if (get_item(k, &e)) {
                uint32_t p = (hash_u64(e.x) >> a.shift) & pmask;
                myp[k] = p;
                mye[k] = e;
                myr[k] = atomicAdd(&cnt[p], 1u);
            }
        }
        prefetch_next();
        __syncthreads();
        if (a.debug & 2) {
            if (tid < np) { cursor[tid] += cnt[tid]; cnt[tid] = 0; }
            __syncthreads();
            return;
        }
        // B: exclusive scan of cnt[0..np): wave scans, then the totals of the preceding waves are added
        if (tid < npad) {
            uint32_t c = cnt[tid], inc = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
            off[tid] = inc - c;
            if (lane == 63) wtot[tid >> 6] = inc;
        }
        __syncthreads();
        if (tid < npad) {
            uint32_t add = 0;
            for (int w = 0; w < (tid >> 6); w++) add += wtot[w];
            off[tid] += add;
            if (tid == npad - 1) s_total = off[tid] + cnt[tid];
        }
        __syncthreads();
        // C: entries -> LDS stage, grouped by partition
#pragma unroll
        for (int k = 0; k < PT_ITEMS; k++) {
            if (myp[k] != 0xFFFFFFFFu) {
                uint32_t pos = off[myp[k]] + myr[k];
                stage[pos] = mye[k];
                part_of[pos] = (uint16_t)myp[k];
            }
        }
        __syncthreads();
        // D: copy out: consecutive lanes write consecutive 16-byte entries of one partition's run
        const uint32_t total = (a.debug & 1) ? 0 : s_total;
#pragma unroll
        for (int q = 0; q < PT_ITEMS; q++) {
            const uint32_t i = (uint32_t)q * PT_BLOCK + tid;
            if (i < total) {
                uint32_t p = part_of[i];
                uint32_t j = cursor[p] + (i - off[p]);
                if (j < (uint32_t)a.out_cap) a.out_entries[(out_base + (int64_t)p * out_stride) * a.out_cap + j] = stage[i];
                else atomicAdd(&s_spill, 1u);
            }
        }
        __syncthreads();
        // D2: regions that are full (a heavy key, an uneven split) spill into one global buffer: one global atomic per
        // tile.  The entries are found again here instead of being remembered above (that cost the common path 24
        // bytes of scratch per lane and 0.5 ms).
        const uint32_t nspill = s_spill;  // uniform: read by everyone before thread 0 resets it
        if (nspill) {
            __syncthreads();
            if (tid == 0) { s_spill_base = atomicAdd(&a.flags[2], (unsigned long long)nspill); s_spill = 0; }
            __syncthreads();
            const unsigned long long sb = s_spill_base;
            // the spill buffer itself full (more than half of the rows in overflowing regions): one flag store per tile,
            // not one per lost entry (agent-scope stores to one address serialise: 50 ms per 1e8 of them)
            if (tid == 0 && (int64_t)(sb + nspill) > a.spill_cap) __hip_atomic_store(&a.flags[0], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint32_t i = tid; i < total; i += PT_BLOCK) {
                uint32_t p = part_of[i];
                uint32_t j = cursor[p] + (i - off[p]);
                if (j < (uint32_t)a.out_cap) continue;
                const unsigned long long pos = sb + atomicAdd(&s_spill, 1u);
                if ((int64_t)pos < a.spill_cap) a.spill[pos] = stage[i];
            }
            __syncthreads();
            if (tid == 0) s_spill = 0;
            __syncthreads();
        }
        // E: advance cursors
        if (tid < np) { cursor[tid] += cnt[tid]; cnt[tid] = 0; }
        __syncthreads();
    };

    if (FROM_ROWS) {
        // two 16-byte loads per column: rows (base + 2 tid, +1) and (base + 2048 + 2 tid, +1)
        ulonglong2 kk[PT_PAIRS];
        double2 vv[PT_PAIRS], pv[PT_PAIRS];
        auto load_rows = [&](int64_t tile) {
            const int64_t base = tile * PT_TILE;
            if (tile < ntiles && base + PT_TILE <= a.nrows) {
#pragma unroll
                for (int u = 0; u < PT_PAIRS; u++) {
                    int64_t r = base + (int64_t)u * 2 * PT_BLOCK + 2 * tid;
                    kk[u] = *(const ulonglong2*)(a.kp + r);
                    vv[u] = a.has_expr ? expr_eval2(a.expr, r) : *(const double2*)(a.vp + r);
                    if (a.has_pred && !a.pred_is_v) pv[u] = *(const double2*)(a.pp + r);
                }
            }
        };
        load_rows(blockIdx.x);
        for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int64_t base = tile * PT_TILE;
            const bool full = base + PT_TILE <= a.nrows;
            process_tile([&](int k, ulonglong2* e) -> bool {
                const int u = k >> 1, el = k & 1;
                int64_t r = base + (int64_t)u * 2 * PT_BLOCK + 2 * tid + el;
                if (r >= a.nrows) return false;
                uint64_t key; double v, p;
                if (full) {
                    key = el ? kk[u].y : kk[u].x;
                    v = el ? vv[u].y : vv[u].x;
                    p = a.pred_is_v ? v : (el ? pv[u].y : pv[u].x);
                } else {
                    key = a.kp[r]; v = a.has_expr ? expr_eval1(a.expr, r) : a.vp[r]; p = a.has_pred ? (a.pred_is_v ? v : a.pp[r]) : 0.0;
                }
                if (a.has_pred && !cmp_apply<double>(a.op, p, a.thr)) return false;
                e->x = key;
                e->y = (unsigned long long)__double_as_longlong(v);
                return true;
            }, [&]() { load_rows(tile + gridDim.x); });
        }
    } else {
        // the input regions of this workgroup are read as ONE concatenated stream (prefix sums of the region
        // counts in LDS, binary search per item), so every tile is full
        __shared__ uint32_t rstart[PT_MAX_REGIONS + 1];
        const int per_max = (a.in_regions + a.in_split - 1) / a.in_split;
        const int g = blockIdx.x % a.in_split;
        const int first = g * per_max;
        const int per = first + per_max <= a.in_regions ? per_max : (a.in_regions > first ? a.in_regions - first : 0);
        const int64_t region0 = src_first + first;
        if (tid == 0) {
            uint32_t run = 0;
            for (int rj = 0; rj < per; rj++) { rstart[rj] = run; run += a.in_counts[region0 + rj]; }
            rstart[per] = run;
        }
        __syncthreads();
        const uint32_t total_in = rstart[per];
        ulonglong2 eb[PT_ITEMS];
        // region of item k of the current tile: v grows by PT_TILE per tile, so the region index only ever moves
        // forward by a step or two -- a per-item cursor replaces a binary search over rstart (8 dependent LDS
        // reads per item, which made this pass load-latency bound: 3.9 of its 4.5 ms)
        int reg[PT_ITEMS];
#pragma unroll
        for (int k = 0; k < PT_ITEMS; k++) reg[k] = 0;
        auto load_entries = [&](uint32_t t0) {
#pragma unroll
            for (int k = 0; k < PT_ITEMS; k++) {
                uint32_t v = t0 + (uint32_t)k * PT_BLOCK + tid;
                if (v < total_in) {
                    int lo = reg[k];  // largest rj with rstart[rj] <= v (rstart[per] = total_in > v)
                    while (rstart[lo + 1] <= v) lo++;
                    reg[k] = lo;
                    eb[k] = a.in_entries[(region0 + lo) * a.in_cap + (v - rstart[lo])];
                }
            }
        };
        load_entries(0);
        for (uint32_t t0 = 0; t0 < total_in; t0 += PT_TILE) {
            process_tile([&](int k, ulonglong2* e) -> bool {
                uint32_t v = t0 + (uint32_t)k * PT_BLOCK + tid;
                if (v >= total_in) return false;
                *e = eb[k];
                return true;
            }, [&]() { load_entries(t0 + PT_TILE); });
        }
    }
    if (tid < np) a.out_counts[out_base + (int64_t)tid * out_stride] = cursor[tid] < (uint32_t)a.out_cap ? cursor[tid] : (uint32_t)a.out_cap;
}

// -------------------------------------------------------------------------------------------------------
// Wide entries: aggregates over 2-6 input columns carry (key, v1, v2[, v3 ...]) = E <= 7 8-byte words per entry.  Same
// passes and region bookkeeping as part_scatter_kernel, 4096-entry tiles (the LDS stage holds E words per entry),
// 8-byte column loads, no prefetch and no spill buffer: a straightforward version -- it only has to beat the HBM
// atomics of the general path (two input columns, G >= 1e5: 217-250 ms per 1e9 rows).
// -------------------------------------------------------------------------------------------------------
// items per lane and tile: one- and two-word entries get 8192-entry tiles (twice the run length), wider ones 4096
constexpr int pw_items(int E) { return E <= 2 ? 8 : (E <= 4 ? 4 : 2); }   // 2048-entry tiles for 5- to 7-word entries (up to 112 KB of LDS)
constexpr int pw_tile(int E) { return PT_BLOCK * pw_items(E); }
// IT: items per lane and tile (8 only for one- and two-word entries; pass 2 keeps 4 when it has few sub-partitions:
// its runs are long anyway and three resident workgroups beat one)
template <bool FROM_ROWS, int E, int IT>
__global__ __launch_bounds__(PT_BLOCK) void part_scatter_wide_kernel(PartArgs a) {
    constexpr int PW_ITEMS = IT, PW_TILE = PT_BLOCK * IT;
    extern __shared__ uint64_t wstage[];  // [PW_TILE][E]
    __shared__ uint16_t part_of[PW_TILE];
    __shared__ uint32_t cnt[PT_MAXP], off[PT_MAXP], cursor[PT_MAXP];
    __shared__ uint32_t s_total;
    __shared__ uint32_t s_over;   // a region of this workgroup is full and there is no spill buffer (or it is full too): the attempt is lost, stop working on it
    __shared__ uint32_t s_spill;
    __shared__ unsigned long long s_spill_base;
    __shared__ uint32_t wtot[PT_MAXP / 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int np = a.nparts;
    const uint32_t pmask = (uint32_t)np - 1;
    const int npad = np < 64 ? 64 : np;
    for (int i = tid; i < PT_MAXP; i += PT_BLOCK) { cnt[i] = 0; cursor[i] = 0; }
    if (tid == 0) { s_over = 0; s_spill = 0; }
    __syncthreads();
    uint64_t* const oute = (uint64_t*)a.out_entries;
    uint64_t* const spill = (uint64_t*)a.spill;   // [spill_cap][E] words: entries of full regions (heavy keys), as in part_scatter_kernel
    const uint64_t* const ine = (const uint64_t*)a.in_entries;
    int64_t out_base, out_stride;
    if (FROM_ROWS) { out_base = blockIdx.x; out_stride = gridDim.x; }
    else {
        const int pin = blockIdx.x / a.in_split, g = blockIdx.x % a.in_split;
        out_base = (int64_t)pin * np * a.in_split + g; out_stride = a.in_split;
    }
    uint64_t ent[PW_ITEMS][E];
    auto process_tile = [&](uint32_t valid) {
        uint32_t myp[PW_ITEMS], myr[PW_ITEMS];
#pragma unroll
        for (int k = 0; k < PW_ITEMS; k++) {
            myp[k] = 0xFFFFFFFFu;
            if ((valid >> k) & 1u) {
                uint32_t p = (hash_u64(ent[k][0]) >> a.shift) & pmask;
                myp[k] = p;
                myr[k] = atomicAdd(&cnt[p], 1u);
            }
        }
        __syncthreads();
        if (tid < npad) {
            uint32_t c = cnt[tid], inc = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
            off[tid] = inc - c;
            if (lane == 63) wtot[tid >> 6] = inc;
        }
        __syncthreads();
        if (tid < npad) {
            uint32_t add = 0;
            for (int w = 0; w < (tid >> 6); w++) add += wtot[w];
            off[tid] += add;
            if (tid == npad - 1) s_total = off[tid] + cnt[tid];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PW_ITEMS; k++) {
            if (myp[k] != 0xFFFFFFFFu) {
                uint32_t pos = off[myp[k]] + myr[k];
#pragma unroll
                for (int e = 0; e < E; e++) wstage[pos * E + e] = ent[k][e];
                part_of[pos] = (uint16_t)myp[k];
            }
        }
        __syncthreads();
        const uint32_t total = s_total;
        for (uint32_t i = tid; i < total; i += PT_BLOCK) {
            uint32_t p = part_of[i];
            uint32_t j = cursor[p] + (i - off[p]);
            if (j < (uint32_t)a.out_cap) {
                uint64_t* dst = oute + (((out_base + (int64_t)p * out_stride) * a.out_cap + j) * E);
#pragma unroll
                for (int e = 0; e < E; e++) dst[e] = wstage[i * E + e];
            } else if (spill) atomicAdd(&s_spill, 1u);
            else s_over = 1;   // (an agent-scope store per lost entry here: 1e8 rows of one heavy key = 50 ms of stores to one address)
        }
        __syncthreads();
        const uint32_t nspill = s_spill;  // uniform: read by everyone before thread 0 resets it
        if (nspill) {   // one global atomic per tile reserves the spilled entries' places; they are found again here
            __syncthreads();
            if (tid == 0) {
                s_spill_base = atomicAdd(&a.flags[2], (unsigned long long)nspill);
                s_spill = 0;
                if ((int64_t)(s_spill_base + nspill) > a.spill_cap) s_over = 1;
            }
            __syncthreads();
            const unsigned long long sb = s_spill_base;
            for (uint32_t i = tid; i < total; i += PT_BLOCK) {
                uint32_t p = part_of[i];
                uint32_t j = cursor[p] + (i - off[p]);
                if (j < (uint32_t)a.out_cap) continue;
                const unsigned long long pos = sb + atomicAdd(&s_spill, 1u);
                if ((int64_t)pos < a.spill_cap) {
#pragma unroll
                    for (int e = 0; e < E; e++) spill[pos * E + e] = wstage[i * E + e];
                }
            }
            __syncthreads();
            if (tid == 0) s_spill = 0;
        }
        __syncthreads();
        if (tid < np) { cursor[tid] += cnt[tid]; cnt[tid] = 0; }
        __syncthreads();
    };
    if (FROM_ROWS) {
        const int64_t ntiles = (a.nrows + PW_TILE - 1) / PW_TILE;
        for (int64_t tile = blockIdx.x; tile < ntiles && !s_over; tile += gridDim.x) {
            uint32_t valid = 0;
#pragma unroll
            for (int k = 0; k < PW_ITEMS; k++) {
                const int64_t row = tile * PW_TILE + (int64_t)k * PT_BLOCK + tid;
                const int64_t rc = row < a.nrows ? row : a.nrows - 1;
                if (row < a.nrows && (!a.wp.enabled || pred_eval(a.wp, a.wpred, rc))) valid |= 1u << k;
                ent[k][0] = a.kp[rc];
                uint64_t vmask = 0;
#pragma unroll
                for (int c = 0; c < E - 1; c++) {
                    if (c < a.nval) {
                        ent[k][1 + c] = col_raw_bits(a.vcols[c], rc);
                        if (col_valid(a.vcols[c], rc)) vmask |= 1ULL << c;
                    }
                }
                if (a.has_vmask) ent[k][E - 1] = vmask;
            }
            process_tile(valid);
        }
    } else {
        __shared__ uint32_t rstart[PT_MAX_REGIONS + 1];
        const int pin = blockIdx.x / a.in_split;
        const int per_max = (a.in_regions + a.in_split - 1) / a.in_split;
        const int g = blockIdx.x % a.in_split;
        const int first = g * per_max;
        const int per = first + per_max <= a.in_regions ? per_max : (a.in_regions > first ? a.in_regions - first : 0);
        const int64_t region0 = (int64_t)pin * a.in_regions + first;
        if (tid == 0) {
            uint32_t run = 0;
            for (int rj = 0; rj < per; rj++) { rstart[rj] = run; run += a.in_counts[region0 + rj]; }
            rstart[per] = run;
        }
        __syncthreads();
        const uint32_t total_in = rstart[per];
        int reg[PW_ITEMS];
#pragma unroll
        for (int k = 0; k < PW_ITEMS; k++) reg[k] = 0;
        for (uint32_t t0 = 0; t0 < total_in && !s_over; t0 += PW_TILE) {
            uint32_t valid = 0;
#pragma unroll
            for (int k = 0; k < PW_ITEMS; k++) {
                uint32_t v = t0 + (uint32_t)k * PT_BLOCK + tid;
                if (v < total_in) {
                    int lo = reg[k];
                    while (rstart[lo + 1] <= v) lo++;
                    reg[k] = lo;
                    const uint64_t* src = ine + (((region0 + lo) * a.in_cap + (v - rstart[lo])) * E);
#pragma unroll
                    for (int e = 0; e < E; e++) ent[k][e] = src[e];
                    valid |= 1u << k;
                }
            }
            process_tile(valid);
        }
    }
    if (tid == 0 && s_over) __hip_atomic_store(&a.flags[0], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < np) a.out_counts[out_base + (int64_t)tid * out_stride] = cursor[tid] < (uint32_t)a.out_cap ? cursor[tid] : (uint32_t)a.out_cap;
}

struct PartAggArgs {
    const ulonglong2* entries;
    const uint32_t* counts;
    int64_t cap;
    int regions;          // regions per final partition
    int64_t nfinal;       // number of final partitions
    int w_rows, w_valid, w_sum, n_words;
    int w_lo;             // compensation word of the float64 sum (w_sum + 1), -1 = plain sum
    uint64_t* dkey;       // [2][dstride]
    uint64_t* dacc;       // [W][dstride]
    int64_t dstride;
    unsigned long long* flags;  // [0] overflow  [1] dense count
    // few final partitions: `splits` workgroups share one partition (regions rj % splits == part) and merge
    // their LDS tables into the HBM table instead of appending dense groups
    int splits;
    int to_table;
    GTable g;
    int64_t table_limit;
    unsigned long long* dir;  // [2 * nfinal]: (first dense row, row count) of every final partition, or NULL
    // generic accumulator program over the entry's value bits (part_agg_generic_kernel)
    int n_ops, vtype;
    int ent_words;   // 2 = (key, value); 3 / 4 = key + values [+ validity word] (wide entries)
    int has_vmask;   // wide entries: last word = validity bits of the input columns
    int wide;        // values are raw bits of any numeric width (vtypes[]), not 8-byte values of type vtype
    int vtypes[6];
    AccOp ops[AGG_MAX_OPS];
    int merge[AGG_MAX_WORDS];
    // the same program as a table (part_agg_generic_kernel, when every (kind, column) occurs once): 6 bits per
    // AccKind = accumulator word, 63 = absent; COUNT(*) separately.  Two scalar registers per column instead of
    // kernel-argument loads (and their lgkmcnt waits) inside the entry loop.
    unsigned long long wpack[6];
    int w_rows_g, use_table, nval;
    int slots;  // LDS table size of part_agg_generic_kernel
    int comp;   // float64 sums are compensated (hi, lo) pairs
};

// Find-or-claim the slot of `key` in a final-pass LDS table (PA_SLOTS keys).  The pass is bound by the number of
// instructions a wave issues per entry (PMC: ~250, half of them scalar exec-mask bookkeeping, at one instruction per
// ~4 cycles and SIMD), so the loop has ONE divergent region (the claim) and one exit; the number of claimed slots is
// counted per lane (*ins) and summed once per partition instead of one LDS atomic per claim.  -1 = no room.
constexpr int PA_MAX_PROBES = 256;
__device__ __forceinline__ int pa_find_slot(uint64_t* lkey, uint32_t smask, uint64_t key, uint32_t* ins, uint32_t* s_fail) {
    const uint32_t hv = hash_u64(key);
    uint32_t h = hv & smask;
    // double hashing: an odd step from hash bits the partitioning did not use ([14:11]).  A wave runs as long as its
    // unluckiest lane, and linear probing's clusters make that lane's chain long.
    const uint32_t step = ((hv >> 11) & 15u) * 2u + 1u;
    for (int probe = 0; probe < PA_MAX_PROBES; probe++) {
        uint64_t k = __hip_atomic_load(&lkey[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k == EMPTY) {
            uint64_t expected = EMPTY;
            const bool won = __hip_atomic_compare_exchange_strong(&lkey[h], &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_WORKGROUP);
            *ins += won ? 1u : 0u;
            k = won ? key : expected;
        }
        if (k == key) return (int)h;
        h = (h + step) & smask;
        if ((probe & 31) == 31 && __hip_atomic_load(s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;  // the partition is lost anyway
    }
    return -1;
}

// sum of v over the block's lanes that call it (all of them), added to *dst by one lane per wave
__device__ __forceinline__ void pa_block_add(uint32_t* dst, uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, v);
}

__global__ __launch_bounds__(PA_BLOCK) __attribute__((amdgpu_waves_per_eu(6, 8))) void part_agg_kernel(PartAggArgs a) {
    // 48 KB per workgroup (40 KB before the compensation terms of the float64 sums: FOUR fitted a CU then, three now;
    // the pass is latency-bound: 1 / 2 / 3 / 4 resident workgroups ran at 5.6 / 3.3 / 2.6 / 2.4 ms).  The last PA_DEAD slots of the key table are never used (they hold a reserved marker that
    // probes step over), which pays for the scalars below and for the accumulators of the two keys that cannot live in
    // the table: EMPTY (the free marker) and PA_RESERVED itself, at lsum / lcnt [PA_LIVE] and [PA_LIVE + 1].
    constexpr int PA_DEAD = 8, PA_LIVE = PA_SLOTS - PA_DEAD;
    constexpr uint64_t PA_RESERVED = EMPTY - 1;
    __shared__ uint64_t lkey[PA_SLOTS];
    __shared__ uint64_t lsum[PA_LIVE + 2];
    // compensation terms of the float64 sums (M_ADD_F64C) in SINGLE precision: they are ~2^-53 of the sum, so 24 bits of
    // them keep hi + lo within 2^-77; a float64 array would cost the fourth resident workgroup.  An error term beyond
    // float range (|sum| > ~1e54) fails the partition over to the general path; one below it (|sum| < ~1e-22) is dropped.
    __shared__ float llo[PA_LIVE + 2];
    __shared__ uint32_t lcnt[PA_LIVE + 2];
    __shared__ uint32_t s_n, s_fail, s_sp[2];
    __shared__ unsigned s_new;
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t smask = PA_SLOTS - 1;
    if (tid == 0) s_new = 0;
    for (int64_t unit = blockIdx.x; unit < a.nfinal * a.splits; unit += gridDim.x) {
        const int64_t f = unit / a.splits;
        const int part = (int)(unit % a.splits);
        for (int i = tid; i < PA_SLOTS; i += PA_BLOCK) {
            lkey[i] = i < PA_LIVE ? EMPTY : PA_RESERVED;
            if (i < PA_LIVE + 2) { lsum[i] = F64_NEG_ZERO; lcnt[i] = 0; llo[i] = 0.0f; }
        }
        if (tid == 0) { s_n = 0; s_fail = 0; s_sp[0] = 0; s_sp[1] = 0; }
        __syncthreads();
        uint32_t ins = 0;  // slots this lane claimed in this partition's table
        for (int rj = part; rj < a.regions; rj += a.splits) {
            const int64_t region = f * a.regions + rj;
            const uint32_t n = a.counts[region];
            const ulonglong2* src = a.entries + region * a.cap;
            for (uint32_t i0 = 0; i0 < n; i0 += PA_BLOCK * 4) {
              // four independent 16-byte loads in flight per lane before the (serial) LDS insertions
              ulonglong2 eb[4];
#pragma unroll
              for (int u = 0; u < 4; u++) {
                  uint32_t i = i0 + (uint32_t)u * PA_BLOCK + tid;
                  if (i < n) eb[u] = src[i];
              }
#pragma unroll
              for (int u = 0; u < 4; u++) {
                uint32_t i = i0 + (uint32_t)u * PA_BLOCK + tid;
                if (i >= n) continue;
                ulonglong2 e = eb[u];
                const uint64_t key = e.x;
                int slot;
                if (key >= PA_RESERVED) { const int sp = key == EMPTY ? 0 : 1; slot = PA_LIVE + sp; s_sp[sp] = 1; }
                else {
                    slot = pa_find_slot(lkey, smask, key, &ins, &s_fail);
                    if (slot < 0) s_fail = 1;
                }
                if (slot >= 0) {
                    const double x = __longlong_as_double((long long)e.y);
                    if (a.w_lo >= 0) {
                        const double old = __hip_atomic_fetch_add((double*)&lsum[slot], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const double er = two_sum_err(old, x, old + x);
                        if (er != 0.0) {
                            const float ef = (float)er;
                            if (ef - ef != 0.0f) s_fail = 1;
                            __hip_atomic_fetch_add(&llo[slot], ef, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    } else __hip_atomic_fetch_add((double*)&lsum[slot], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    atomicAdd(&lcnt[slot], 1u);
                }
              }
            }
        }
        pa_block_add(&s_n, ins);
        __syncthreads();
        if (s_fail) {  // more groups than the LDS table holds: tell the host to use the general path
            if (tid == 0) __hip_atomic_store(&a.flags[0], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (a.to_table) {
            // merge this workgroup's table into the HBM table (G * splits * words atomics in total: small)
            if (tid == 0) {
                fold_new(a.g, &s_new);
                unsigned long long fill = __hip_atomic_load(&a.g.ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int64_t)(fill + s_n) > a.table_limit) s_fail = 1;
            }
            __syncthreads();
            if (s_fail) {
                if (tid == 0) __hip_atomic_store(&a.flags[0], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            for (int i = tid; i < PA_LIVE + 2; i += PA_BLOCK) {
                uint64_t k = i < PA_LIVE ? lkey[i] : (s_sp[i - PA_LIVE] ? (i == PA_LIVE ? 0 : PA_RESERVED) : EMPTY);
                if (k == EMPTY) continue;
                uint64_t slot;
                if (i != PA_LIVE) slot = gt_find_single(a.g, k, &s_new);
                else { slot = a.g.cap; if (ld_agent(&a.g.tag[slot]) == EMPTY) st_agent(&a.g.tag[slot], 0); }
                if (a.w_rows >= 0) g_merge(&a.g.acc[(uint64_t)a.w_rows * a.g.stride + slot], M_ADD_U64, lcnt[i], 0);
                if (a.w_valid >= 0) g_merge(&a.g.acc[(uint64_t)a.w_valid * a.g.stride + slot], M_ADD_U64, lcnt[i], 0);
                if (a.w_sum >= 0) g_merge(&a.g.acc[(uint64_t)a.w_sum * a.g.stride + slot], a.w_lo >= 0 ? M_ADD_F64C : M_ADD_F64, lsum[i], (int64_t)a.g.stride);
                if (a.w_lo >= 0 && llo[i] != 0.0f)
                    g_merge(&a.g.acc[(uint64_t)a.w_lo * a.g.stride + slot], M_ADD_F64, (uint64_t)__double_as_longlong((double)llo[i]), 0);
            }
            __syncthreads();
            if (tid == 0) fold_new(a.g, &s_new);
            continue;
        }
        // compact: reserve a dense range for this partition's groups, then write them
        const uint32_t ngroups = s_n + s_sp[0] + s_sp[1];
        __syncthreads();
        if (tid == 0) {
            s_base = atomicAdd(&a.flags[1], (unsigned long long)ngroups);
            s_n = 0;
            if (a.dir) { a.dir[2 * f] = s_base; a.dir[2 * f + 1] = ngroups; }
        }
        __syncthreads();
        if ((int64_t)(s_base + ngroups) > a.dstride) {
            if (tid == 0) __hip_atomic_store(&a.flags[0], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        for (int i0 = 0; i0 < PA_LIVE + 2; i0 += PA_BLOCK) {
            int i = i0 + tid;
            bool occ = i < PA_LIVE ? lkey[i] != EMPTY : (i < PA_LIVE + 2 && s_sp[i - PA_LIVE] != 0);
            uint64_t b = __ballot(occ);
            uint32_t wbase = 0;
            if (lane == 0 && b) wbase = atomicAdd(&s_n, (uint32_t)__popcll(b));
            wbase = __shfl(wbase, 0);
            if (occ) {
                uint64_t lt = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
                int64_t pos = (int64_t)s_base + wbase + __popcll(b & lt);
                a.dkey[pos] = i < PA_LIVE ? lkey[i] : (i == PA_LIVE ? EMPTY : PA_RESERVED);
                a.dkey[a.dstride + pos] = 0;
                if (a.w_rows >= 0) a.dacc[(int64_t)a.w_rows * a.dstride + pos] = lcnt[i];
                if (a.w_valid >= 0) a.dacc[(int64_t)a.w_valid * a.dstride + pos] = lcnt[i];
                if (a.w_sum >= 0) a.dacc[(int64_t)a.w_sum * a.dstride + pos] = lsum[i];
                if (a.w_lo >= 0) a.dacc[(int64_t)a.w_lo * a.dstride + pos] = (uint64_t)__double_as_longlong((double)llo[i]);
            }
        }
        __syncthreads();
    }
}

// value an op contributes for an entry whose input value has the raw bits `vb` (type `vtype`, never NULL here)
__device__ __forceinline__ uint64_t op_value_bits(int kind, int vtype, uint64_t vb) {
    switch (kind) {
        case A_COUNT_ROWS:
        case A_COUNT_VALID: return 1;
        case A_SUM_F64: return vtype == VNM_F64 ? vb : (uint64_t)__double_as_longlong((double)(int64_t)vb);
        case A_SUM_I64: return vb;
        case A_SUM_LO32: return vb & 0xFFFFFFFFULL;
        case A_SUM_HI32S: return (uint64_t)((int64_t)vb >> 32);
        case A_SUM_HI32U: return vb >> 32;
        default:  // A_MIN / A_MAX on the order-preserving encoding
            if (vtype == VNM_F64) return enc_f64(__longlong_as_double((long long)vb));
            if (vtype == VNM_U64) return vb;
            return enc_i64((int64_t)vb);
    }
}

// every accumulator word of ONE input column (packed word table, see PartAggArgs::wpack) for a non-NULL value with
// the raw bits `raw` of type `type`
__device__ __forceinline__ void pa_accumulate_col(unsigned long long pack, int type, uint64_t* lw, int ST, int slot, uint64_t raw, int comp) {
#define VNM_WI(K) ((int)((pack >> (6 * (K))) & 63ULL))
#define VNM_W(K) (lw + VNM_WI(K) * ST + slot)
#define VNM_ADD(K, V) if (VNM_WI(K) != 63) __hip_atomic_fetch_add(VNM_W(K), (uint64_t)(V), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
    VNM_ADD(A_COUNT_VALID, 1ULL);
    if (VNM_WI(A_SUM_F64) != 63) {
        if (comp) l_add_f64c(VNM_W(A_SUM_F64), ST, raw_to_f64(type, raw));  // (hi, lo) pair: lo is the next word
        else __hip_atomic_fetch_add((double*)VNM_W(A_SUM_F64), raw_to_f64(type, raw), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (((pack >> (6 * A_SUM_I64)) & 0xFFFFFFULL) != 0xFFFFFFULL) {  // any of the four integer-sum kinds (consecutive AccKinds)
        const uint64_t iv = (uint64_t)raw_to_i64(type, raw);
        VNM_ADD(A_SUM_I64, iv);
        VNM_ADD(A_SUM_LO32, iv & 0xFFFFFFFFULL);
        VNM_ADD(A_SUM_HI32S, (int64_t)iv >> 32);
        VNM_ADD(A_SUM_HI32U, iv >> 32);
    }
    if (VNM_WI(A_MIN) != 63 || VNM_WI(A_MAX) != 63) {
        const uint64_t e = type_is_float(type) ? enc_f64(raw_to_f64(type, raw))
                                               : (type_is_unsigned(type) ? (uint64_t)raw_to_i64(type, raw) : enc_i64(raw_to_i64(type, raw)));
        if (VNM_WI(A_MIN) != 63) __hip_atomic_fetch_min(VNM_W(A_MIN), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (VNM_WI(A_MAX) != 63) __hip_atomic_fetch_max(VNM_W(A_MAX), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#undef VNM_ADD
#undef VNM_W
#undef VNM_WI
}

// Final pass of the partitioned path for ANY accumulator program over one 8-byte input column (or none):
// same protocol as part_agg_kernel, W accumulator words per LDS slot.  LDS: lkey[S + 1], lw[W][S + 1].
// TABLE: the accumulator program as packed word tables (PartAggArgs::wpack) -- the normal case; the op loop is only
// compiled into the <E, false> instances (it costs registers: 89+ VGPRs and scratch left two workgroups per CU).
template <int E, bool TABLE>
__global__ __launch_bounds__(PA_BLOCK) __attribute__((amdgpu_waves_per_eu(6, 8))) void part_agg_generic_kernel(PartAggArgs a) {
    extern __shared__ uint64_t pa_lds[];
    __shared__ uint32_t s_n, s_fail;
    __shared__ unsigned s_new;
    __shared__ unsigned long long s_base;
    const int SL = a.slots;  // 2048, or 1024 for programs with many accumulator words (LDS per workgroup decides how many are resident)
    const int ST = SL + 1;
    uint64_t* lkey = pa_lds;
    uint64_t* lw = pa_lds + ST;
    const int W = a.n_words;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t smask = (uint32_t)SL - 1;
    if (tid == 0) s_new = 0;
    for (int64_t unit = blockIdx.x; unit < a.nfinal * a.splits; unit += gridDim.x) {
        const int64_t f = unit / a.splits;
        const int part = (int)(unit % a.splits);
        for (int i = tid; i < ST; i += PA_BLOCK) lkey[i] = EMPTY;
        for (int w = 0; w < W; w++) {
            const uint64_t init = merge_init(a.merge[w]);
            for (int i = tid; i < ST; i += PA_BLOCK) lw[w * ST + i] = init;
        }
        if (tid == 0) { s_n = 0; s_fail = 0; }
        __syncthreads();
        uint32_t ins = 0;  // slots this lane claimed in this partition's table
        for (int rj = part; rj < a.regions; rj += a.splits) {
            const int64_t region = f * a.regions + rj;
            const uint32_t n = a.counts[region];
            const uint64_t* src = (const uint64_t*)a.entries + region * a.cap * E;
            for (uint32_t i0 = 0; i0 < n; i0 += PA_BLOCK * 4) {
                uint64_t eb[4][E];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    uint32_t i = i0 + (uint32_t)u * PA_BLOCK + tid;
                    if (i < n) {
                        if (E == 2) { const ulonglong2 t = ((const ulonglong2*)src)[i]; eb[u][0] = t.x; eb[u][1] = t.y; }
                        else {
#pragma unroll
                            for (int e = 0; e < E; e++) eb[u][e] = src[(size_t)i * E + e];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    uint32_t i = i0 + (uint32_t)u * PA_BLOCK + tid;
                    if (i >= n) continue;
                    const uint64_t key = eb[u][0];
                    int slot;
                    if (key == EMPTY) { slot = SL; lkey[slot] = 0; }
                    else {
                        slot = pa_find_slot(lkey, smask, key, &ins, &s_fail);
                        if (slot < 0) s_fail = 1;
                    }
                    if (slot >= 0 && TABLE) {
                        const uint64_t vmask = (E > 2 && a.has_vmask) ? eb[u][E - 1] : ~0ULL;
                        if (a.w_rows_g >= 0) __hip_atomic_fetch_add(&lw[a.w_rows_g * ST + slot], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                        for (int c = 0; c < E - 1; c++) {
                            if (c < a.nval && (E == 2 || ((vmask >> c) & 1ULL)))  // a NULL input updates nothing of its column
                                pa_accumulate_col(a.wpack[c], a.vtypes[c], lw, ST, slot, eb[u][1 + c], a.comp);
                        }
                    } else if (slot >= 0 && !TABLE) {
                        const uint64_t vmask = (E > 2 && a.has_vmask) ? eb[u][E - 1] : ~0ULL;
                        for (int o = 0; o < a.n_ops; o++) {
                            const int w = a.ops[o].word;
                            const int c = E == 2 ? 0 : (a.ops[o].col < 0 ? 0 : a.ops[o].col);
                            if (E > 2 && a.ops[o].kind != A_COUNT_ROWS && !((vmask >> c) & 1ULL)) continue;  // NULL input
                            uint64_t vb = E >= 2 ? eb[u][E >= 2 ? 1 : 0] : 0;
#pragma unroll
                            for (int e = 2; e < E; e++) if (c == e - 1) vb = eb[u][e];
                            const uint64_t v = a.wide ? op_value_raw(a.ops[o].kind, a.vtypes[c], vb)  // any numeric type
                                                      : op_value_bits(a.ops[o].kind, a.vtype, vb);
                            l_merge(&lw[w * ST + slot], a.merge[w], v, ST);
                        }
                    }
                }
            }
        }
        pa_block_add(&s_n, ins);
        __syncthreads();
        if (s_fail) {  // more groups than the LDS table holds: tell the host to use the general path
            if (tid == 0) __hip_atomic_store(&a.flags[0], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (a.to_table) {
            if (tid == 0) {
                fold_new(a.g, &s_new);
                unsigned long long fill = __hip_atomic_load(&a.g.ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int64_t)(fill + s_n) > a.table_limit) s_fail = 1;
            }
            __syncthreads();
            if (s_fail) {
                if (tid == 0) __hip_atomic_store(&a.flags[0], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            for (int i = tid; i < ST; i += PA_BLOCK) {
                uint64_t k = lkey[i];
                if (k == EMPTY) continue;
                uint64_t slot;
                if (i < SL) slot = gt_find_single(a.g, k, &s_new);
                else { slot = a.g.cap; if (ld_agent(&a.g.tag[slot]) == EMPTY) st_agent(&a.g.tag[slot], 0); }
                for (int w = 0; w < W; w++) {
                    uint64_t v = lw[w * ST + i];
                    if (v != merge_init(a.merge[w])) g_merge(&a.g.acc[(uint64_t)w * a.g.stride + slot], a.merge[w], v, (int64_t)a.g.stride);
                }
            }
            __syncthreads();
            if (tid == 0) fold_new(a.g, &s_new);
            continue;
        }
        const uint32_t ngroups = s_n + (lkey[SL] != EMPTY ? 1u : 0u);
        __syncthreads();
        if (tid == 0) {
            s_base = atomicAdd(&a.flags[1], (unsigned long long)ngroups);
            s_n = 0;
            if (a.dir) { a.dir[2 * f] = s_base; a.dir[2 * f + 1] = ngroups; }
        }
        __syncthreads();
        if ((int64_t)(s_base + ngroups) > a.dstride) {
            if (tid == 0) __hip_atomic_store(&a.flags[0], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        for (int i0 = 0; i0 < ST; i0 += PA_BLOCK) {
            int i = i0 + tid;
            bool occ = i < ST && lkey[i] != EMPTY;
            uint64_t b = __ballot(occ);
            uint32_t wbase = 0;
            if (lane == 0 && b) wbase = atomicAdd(&s_n, (uint32_t)__popcll(b));
            wbase = __shfl(wbase, 0);
            if (occ) {
                uint64_t lt = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
                int64_t pos = (int64_t)s_base + wbase + __popcll(b & lt);
                a.dkey[pos] = i == SL ? EMPTY : lkey[i];
                a.dkey[a.dstride + pos] = 0;
                for (int w = 0; w < W; w++) a.dacc[(int64_t)w * a.dstride + pos] = lw[w * ST + i];
            }
        }
        __syncthreads();
    }
}

// Cardinality estimate for operators that were given no hint: insert a strided sample of the keys into a
// scratch table (tags only) and count the distinct ones.  Solving d = G (1 - exp(-m / G)) for G (uniform
// model) on the host then sizes the partitions; an underestimate only costs the fallback to the general path.
// (kvalid: the key column's validity bitmap or null -- rows whose key is NULL are not part of any sample: their value words are garbage)
__device__ __forceinline__ bool key_row_valid(const uint8_t* kvalid, int64_t koff, int64_t row) {
    return !kvalid || ((kvalid[(koff + row) >> 3] >> ((koff + row) & 7)) & 1);
}
__global__ void agg_sample_kernel(const uint64_t* keys, int64_t nrows, int64_t m, GTable g, unsigned int* cnt, const uint8_t* kvalid, int64_t koff) {
    __shared__ unsigned s_new;
    if (threadIdx.x == 0) s_new = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        int64_t row = (int64_t)(((__int128)i * nrows) / m);
        uint64_t key = keys[row];
        uint64_t slot = ~0ULL;
        if (key != EMPTY && key_row_valid(kvalid, koff, row)) {
            slot = gt_find_single(g, key, &s_new);
            if (slot >= g.cap) slot = ~0ULL;
        }
        // per-key sample counts (the heavy keys' share of the rows).  A heavy key means most lanes of a wave hold the
        // same slot, and atomics on one address serialise (half of the sample one key: 3 ms; seven groups: +0.2 ms on a 3 ms
        // query): up to eight rounds pick the first pending lane's slot and add all its lanes at once.
        const int lane = threadIdx.x & 63;
        for (int r = 0; r < 8; r++) {
            const unsigned long long pending = __ballot(slot != ~0ULL);
            if (!pending) break;
            const uint64_t first = __shfl(slot, __ffsll((long long)pending) - 1);
            const unsigned long long same = __ballot(slot == first);
            if (slot == first) {
                if (lane == __ffsll((long long)same) - 1) atomicAdd(&cnt[first], (unsigned int)__popcll(same));
                slot = ~0ULL;
            }
        }
        if (slot != ~0ULL) atomicAdd(&cnt[slot], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) fold_new(g, &s_new);
}
// keys with more than `thresh` sample rows: ctl[4] += their rows, ctl[5] += their number
__global__ void agg_sample_heavy_kernel(const unsigned int* cnt, int64_t slots, unsigned int thresh, unsigned long long* ctl) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long rows = 0, keys = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += stride) {
        const unsigned int c = cnt[i];
        if (c > thresh) { rows += c; keys++; }
    }
    if (rows) { atomicAdd(&ctl[4], rows); atomicAdd(&ctl[5], keys); }
}

// HyperLogLog over a strided sample (4096 registers, ~1.6 % error): LDS max per workgroup, then one
// agent-scope max per register per workgroup.  Used for the large sample tier, where inserting every key into
// a scratch table would cost tens of milliseconds of atomics.
constexpr int HLL_BITS = 12;
constexpr int HLL_M = 1 << HLL_BITS;
__global__ __launch_bounds__(1024) void agg_hll_kernel(const uint64_t* keys, int64_t nrows, int64_t m, unsigned int* regs, const uint8_t* kvalid, int64_t koff) {
    __shared__ unsigned int lreg[HLL_M];
    for (int i = threadIdx.x; i < HLL_M; i += blockDim.x) lreg[i] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        int64_t row = (int64_t)(((__int128)i * nrows) / m);
        if (!key_row_valid(kvalid, koff, row)) continue;
        uint64_t key = keys[row];
        uint64_t h = ((uint64_t)hash_u64(key) << 32) | hash_u64(key * 0x9E3779B97F4A7C15ULL + 0x7F4A7C15ULL);
        unsigned idx = (unsigned)(h & (HLL_M - 1));
        uint64_t rest = h >> HLL_BITS;
        unsigned rank = rest ? (unsigned)__clzll((long long)(rest << HLL_BITS)) + 1u : (unsigned)(64 - HLL_BITS + 1);
        atomicMax(&lreg[idx], rank);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HLL_M; i += blockDim.x)
        if (lreg[i]) atomicMax(&regs[i], lreg[i]);
}

// ---- multi-GPU: bucket the dense run by owner rank -------------------------------------------------------
// owner(key words) must equal vinum_amd/distributed.py::owner_of (int64 wrap-around arithmetic).
__device__ __forceinline__ int owner_of_words(const uint64_t* const* kw, int nkw, int64_t i, int world) {
    int64_t hh = 0;
    for (int j = 0; j < nkw; j++) {
        hh = (int64_t)(((uint64_t)hh ^ kw[j][i]) * 0x9E3779B97F4A7C15ULL);
        hh ^= (hh >> 29);  // arithmetic shift, like torch's int64 >>
    }
    return (int)(((hh >> 17) & 0x7FFFFFFF) % world);
}

struct BucketArgs {
    const uint64_t* words[AGG_MAX_KEYS + 1 + AGG_MAX_WORDS];
    int nkw, nw;      // key words, total words
    int64_t n, per;   // rows, rows per block (block b owns the contiguous rows [b*per, (b+1)*per))
    int world, nb;
    unsigned long long* blk;       // [world][nb] per-block counts (pass 0) -> exclusive offsets (after the scan)
    unsigned long long* totals;    // [world]
    uint64_t* out;                 // [n][nw] row-major, grouped by owner
};

// A stable, atomic-free partition by owner (a one-digit radix scatter): per-block counts, one scan, then every
// block re-reads its rows and places them with ballot ranks.  (A shared cursor per owner serialises on one
// atomic word: 37 ms per 1e8 groups.)
constexpr int BK_MAX_WORLD = 64;

__global__ __launch_bounds__(256) void agg_bucket_count_kernel(BucketArgs b) {
    __shared__ unsigned cnt[BK_MAX_WORLD];
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < BK_MAX_WORLD) cnt[tid] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * b.per, hi = lo + b.per < b.n ? lo + b.per : b.n;
    for (int64_t base = lo; base < hi; base += 256) {
        const int64_t i = base + tid;
        const int own = i < hi ? owner_of_words(b.words, b.nkw, i, b.world) : -1;
        for (int o = 0; o < b.world; o++) {
            uint64_t m = __ballot(own == o);
            if (m && lane == 0) atomicAdd(&cnt[o], (unsigned)__popcll(m));
        }
    }
    __syncthreads();
    if (tid < b.world) b.blk[(int64_t)tid * b.nb + blockIdx.x] = cnt[tid];
}

__global__ void agg_bucket_scan_kernel(BucketArgs b) {
    __shared__ unsigned long long tot[BK_MAX_WORLD];
    const int o = threadIdx.x;
    if (o < b.world) {
        unsigned long long s = 0;
        for (int k = 0; k < b.nb; k++) s += b.blk[(int64_t)o * b.nb + k];
        tot[o] = s;
        b.totals[o] = s;
    }
    __syncthreads();
    if (o < b.world) {
        unsigned long long run = 0;
        for (int k = 0; k < o; k++) run += tot[k];
        for (int k = 0; k < b.nb; k++) {
            unsigned long long c = b.blk[(int64_t)o * b.nb + k];
            b.blk[(int64_t)o * b.nb + k] = run;
            run += c;
        }
    }
}

__global__ __launch_bounds__(256) void agg_bucket_scatter_kernel(BucketArgs b) {
    __shared__ unsigned long long run[BK_MAX_WORLD];
    __shared__ unsigned wcount[4][BK_MAX_WORLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < b.world) run[tid] = b.blk[(int64_t)tid * b.nb + blockIdx.x];
    for (int i = tid; i < 4 * BK_MAX_WORLD; i += 256) (&wcount[0][0])[i] = 0;
    __syncthreads();
    const uint64_t lt = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
    const int64_t lo = (int64_t)blockIdx.x * b.per, hi = lo + b.per < b.n ? lo + b.per : b.n;
    for (int64_t base = lo; base < hi; base += 256) {
        const int64_t i = base + tid;
        const int own = i < hi ? owner_of_words(b.words, b.nkw, i, b.world) : -1;
        unsigned rank = 0;
        for (int o = 0; o < b.world; o++) {
            uint64_t m = __ballot(own == o);
            if (own == o) rank = __popcll(m & lt);
            if (m && lane == 0) wcount[wave][o] = (unsigned)__popcll(m);
        }
        __syncthreads();
        if (own >= 0) {
            unsigned long long pos = run[own] + rank;
            for (int w = 0; w < wave; w++) pos += wcount[w][own];
            for (int w = 0; w < b.nw; w++) b.out[pos * b.nw + w] = b.words[w][i];
        }
        __syncthreads();
        if (tid < b.world) {
            unsigned s = 0;
            for (int w = 0; w < 4; w++) { s += wcount[w][tid]; wcount[w][tid] = 0; }
            run[tid] += s;
        }
        __syncthreads();
    }
}

// ---- partition-aligned multi-GPU exchange ------------------------------------------------------------------
// The partitioned path leaves the groups of final partition f contiguous (directory dir[f] = (first row, rows)).
// Every rank uses the same hash bits, so partition f holds the same keys everywhere and owner(f) = f * P / F.
// Sender: rows are copied in partition order (row-major [key, nullmask, words...]).  Owner: one workgroup per
// owned partition loads that partition's segments from all P sources into an LDS table and writes the merged
// groups -- streaming, no HBM atomics.
struct RunReorderArgs {
    const uint64_t* words[2 + AGG_MAX_WORDS];
    int nw;                            // 2 key words + accumulator words
    int64_t nfin;
    const unsigned long long* dir;     // [2 * nfin]
    unsigned long long* prefix;        // [nfin + 1] exclusive prefix of the row counts (in partition order)
    uint64_t* out;                     // [n][nw]
    uint32_t* part_counts;             // [nfin]
};

__global__ void run_prefix_kernel(RunReorderArgs a) {  // single block: serial over chunks, parallel inside
    __shared__ unsigned long long carry;
    __shared__ unsigned long long wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < a.nfin; base += blockDim.x) {
        int64_t f = base + tid;
        unsigned long long c = f < a.nfin ? a.dir[2 * f + 1] : 0, inc = c;
        for (int d = 1; d < 64; d <<= 1) { unsigned long long o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned long long add = carry;
        for (int w = 0; w < wave; w++) add += wsum[w];
        if (f < a.nfin) { a.prefix[f] = add + inc - c; a.part_counts[f] = (uint32_t)c; }
        __syncthreads();
        if (tid == blockDim.x - 1) carry = add + inc;
        __syncthreads();
    }
    if (tid == 0) a.prefix[a.nfin] = carry;
}

__global__ __launch_bounds__(256) void run_reorder_kernel(RunReorderArgs a) {
    for (int64_t f = blockIdx.x; f < a.nfin; f += gridDim.x) {
        const unsigned long long src = a.dir[2 * f], cnt = a.dir[2 * f + 1], dst = a.prefix[f];
        for (unsigned long long e = threadIdx.x; e < cnt * a.nw; e += blockDim.x) {
            unsigned long long r = e / a.nw;
            int w = (int)(e % a.nw);
            a.out[(dst + r) * a.nw + w] = a.words[w][src + r];
        }
    }
}

struct PartMergeArgs {
    const uint64_t* rows;              // all received rows, grouped by source rank, each in partition order
    const unsigned long long* src_prefix;  // [world][nlocal + 1] row offsets (absolute, into rows)
    int world;
    int64_t nlocal;
    int nw, n_words;
    int merge[AGG_MAX_WORDS];
    uint64_t* dkey;                    // [2][dstride]
    uint64_t* dacc;                    // [W][dstride]
    int64_t dstride;
    unsigned long long* flags;         // [0] overflow [1] dense count
};
constexpr int PM_MAX_WORDS = 3;

__global__ __launch_bounds__(PA_BLOCK) void part_merge_kernel(PartMergeArgs a) {
    __shared__ uint64_t lkey[PA_SLOTS + 1];
    __shared__ uint64_t lw[PM_MAX_WORDS][PA_SLOTS + 1];
    __shared__ uint32_t s_n, s_fail;
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t smask = PA_SLOTS - 1;
    for (int64_t f = blockIdx.x; f < a.nlocal; f += gridDim.x) {
        for (int i = tid; i <= PA_SLOTS; i += PA_BLOCK) {
            lkey[i] = EMPTY;
            for (int w = 0; w < a.n_words; w++) lw[w][i] = 0;
        }
        if (tid == 0) { s_n = 0; s_fail = 0; }
        __syncthreads();
        for (int r = 0; r < a.world; r++) {
            const unsigned long long lo = a.src_prefix[(int64_t)r * (a.nlocal + 1) + f];
            const unsigned long long hi = a.src_prefix[(int64_t)r * (a.nlocal + 1) + f + 1];
            for (unsigned long long i = lo + tid; i < hi; i += PA_BLOCK) {
                const uint64_t* row = a.rows + i * a.nw;
                const uint64_t key = row[0];
                int slot = -1;
                if (key == EMPTY) { slot = PA_SLOTS; lkey[slot] = 0; }
                else {
                    uint32_t h = hash_u64(key) & smask;
                    for (int probe = 0; probe < PA_SLOTS; probe++) {
                        uint64_t k = __hip_atomic_load(&lkey[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (k == key) { slot = (int)h; break; }
                        if (k == EMPTY) {
                            uint64_t expected = EMPTY;
                            if (__hip_atomic_compare_exchange_strong(&lkey[h], &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                     __HIP_MEMORY_SCOPE_WORKGROUP)) {
                                if (atomicAdd(&s_n, 1u) >= (uint32_t)(PA_SLOTS * 9 / 10)) s_fail = 1;
                                slot = (int)h;
                                break;
                            }
                            if (expected == key) { slot = (int)h; break; }
                        }
                        h = (h + 1) & smask;
                        if ((probe & 15) == 15 && s_fail) break;
                    }
                }
                if (slot >= 0)
                    for (int w = 0; w < a.n_words; w++) l_merge(&lw[w][slot], a.merge[w], row[2 + w], PA_SLOTS + 1);
            }
        }
        __syncthreads();
        if (s_fail) {
            if (tid == 0) __hip_atomic_store(&a.flags[0], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const uint32_t ngroups = s_n + (lkey[PA_SLOTS] != EMPTY ? 1u : 0u);
        if (tid == 0) { s_base = atomicAdd(&a.flags[1], (unsigned long long)ngroups); s_n = 0; }
        __syncthreads();
        if ((int64_t)(s_base + ngroups) > a.dstride) {
            if (tid == 0) __hip_atomic_store(&a.flags[0], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        for (int i0 = 0; i0 <= PA_SLOTS; i0 += PA_BLOCK) {
            int i = i0 + tid;
            bool occ = i <= PA_SLOTS && lkey[i] != EMPTY;
            uint64_t b = __ballot(occ);
            uint32_t wbase = 0;
            if (lane == 0 && b) wbase = atomicAdd(&s_n, (uint32_t)__popcll(b));
            wbase = __shfl(wbase, 0);
            if (occ) {
                uint64_t lt = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
                int64_t pos = (int64_t)s_base + wbase + __popcll(b & lt);
                a.dkey[pos] = i == PA_SLOTS ? EMPTY : lkey[i];
                a.dkey[a.dstride + pos] = 0;
                for (int w = 0; w < a.n_words; w++) a.dacc[(int64_t)w * a.dstride + pos] = lw[w][i];
            }
        }
        __syncthreads();
    }
}

__global__ void fill_u64_kernel(uint64_t* p, uint64_t v, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// =======================================================================================================
// Packed composite keys.  GROUP BY a, b, ... with keys whose values span few bits (dimension columns) is the
// common multi-column case.  The key columns of a batch are packed into ONE 64-bit word per row
//     word = sum_j code_j << shift_j,   code_j = value_j - lo_j  (or the all-ones code of the field for NULL)
// which is a bijection on the observed ranges, so the single-key machinery (LDS pre-aggregation, partitioned
// path) applies unchanged and the result keys are unpacked at the end.  The top bit stays clear, so a packed
// word never equals the EMPTY sentinel.  A later batch outside the ranges demotes the operator to the wide-key
// table (the groups so far are unpacked and merged there).
// =======================================================================================================
//
// Dictionary-coded fields.  A key column whose RANGE is too wide for the word (hashed ids, float64 keys, two full-range
// int64 columns) but whose distinct values are few enough gets its code from a per-column open-addressing table in HBM
// instead: code = the slot the value was inserted at (one CAS per new value, a plain read for every other row -- a
// slot changes once, EMPTY -> value, so whatever a lane reads that is not EMPTY is final), decode = table[code].  The
// table has 2^dbits slots sized from the distinct-count estimate of the first batch; the value EMPTY itself owns the
// extra slot 2^dbits, whose word stays EMPTY and therefore decodes to itself.  The field is dbits + 1 bits wide, so
// two such columns always fit.  A probe chain beyond DICT_MAX_PROBES (the table filling up in later batches) raises
// the same out-of-range flag as a plain field: the operator demotes to the wide-key table.
struct PackParams {
    int n;
    int shift[AGG_MAX_KEYS];
    int bits[AGG_MAX_KEYS];
    uint64_t lo[AGG_MAX_KEYS];   // value bits of code 0
    uint64_t* dtab[AGG_MAX_KEYS];   // dictionary-coded field: its table [2^dbits + 1], else nullptr
    int dbits[AGG_MAX_KEYS];
    vnm_dcol cols[AGG_MAX_KEYS];
};

constexpr int DICT_MAX_PROBES = 256;

__device__ __forceinline__ uint64_t dict_mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    return x ^ (x >> 33);
}

// code of value bits kb in the field's table (inserting it if new); `bad` when the chain is too long
__device__ __forceinline__ uint64_t dict_code(uint64_t* tab, int dbits, uint64_t kb, bool& bad) {
    const uint64_t mask = (1ULL << dbits) - 1;
    if (kb == EMPTY) return mask + 1;
    uint64_t slot = dict_mix(kb) & mask;
    for (int probe = 0; probe < DICT_MAX_PROBES; probe++) {
        uint64_t cur = tab[slot];
        if (cur == EMPTY) {   // (or a stale line of this CU's L1: the CAS at the L2 tells.  Re-reading past the L1 first -- an
                              // agent-scope load anywhere in this loop -- made the whole kernel 30 % slower, taken or not)
            cur = atomicCAS((unsigned long long*)&tab[slot], (unsigned long long)EMPTY, (unsigned long long)kb);
            if (cur == EMPTY) return slot;
        }
        if (cur == kb) return slot;
        slot = (slot + 1) & mask;
    }
    bad = true;
    return 0;
}

// per key column: min / max of the key bits as int64 (order-preserving encoding for the unsigned atomics)
__global__ __launch_bounds__(256) void key_range_kernel(PackParams p, int64_t nrows, unsigned long long* out /* [n][2] */) {
    __shared__ unsigned long long smin[AGG_MAX_KEYS], smax[AGG_MAX_KEYS];
    if (threadIdx.x < AGG_MAX_KEYS) { smin[threadIdx.x] = ~0ULL; smax[threadIdx.x] = 0; }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int j = 0; j < p.n; j++) {
        uint64_t mn = ~0ULL, mx = 0;
        for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < nrows; i0 += stride * 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) {  // four independent rows per lane
                const int64_t i = i0 + (int64_t)u * stride;
                const int64_t ic = i < nrows ? i : nrows - 1;
                const bool ok = col_valid(p.cols[j], ic);
                const uint64_t e = enc_i64((int64_t)col_key_bits(p.cols[j], ic));
                if (ok && i < nrows) { mn = e < mn ? e : mn; mx = e > mx ? e : mx; }
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            uint64_t a = __shfl_xor(mn, d), b = __shfl_xor(mx, d);
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&smin[j], (unsigned long long)mn); atomicMax(&smax[j], (unsigned long long)mx); }
    }
    __syncthreads();
    if (threadIdx.x < p.n) { atomicMin(&out[2 * threadIdx.x], smin[threadIdx.x]); atomicMax(&out[2 * threadIdx.x + 1], smax[threadIdx.x]); }
}

__global__ __launch_bounds__(256) void key_pack_kernel(PackParams p, int64_t nrows, uint64_t* packed, unsigned long long* out_of_range) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    bool bad = false;
    constexpr int U = 4;  // independent rows per lane: their column loads overlap
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < nrows; i0 += stride * U) {
        uint64_t w[U];
#pragma unroll
        for (int u = 0; u < U; u++) w[u] = 0;
        for (int j = 0; j < p.n; j++) {
            const uint64_t cap = (1ULL << p.bits[j]) - 1;  // values use codes [0, cap), NULL is cap
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t i = i0 + (int64_t)u * stride;
                const int64_t ic = i < nrows ? i : nrows - 1;
                uint64_t code = cap;
                if (col_valid(p.cols[j], ic)) {
                    if (p.dtab[j]) code = dict_code(p.dtab[j], p.dbits[j], col_key_bits(p.cols[j], ic), bad);
                    else {
                        code = col_key_bits(p.cols[j], ic) - p.lo[j];
                        bad = bad || (i < nrows && code >= cap);
                    }
                }
                w[u] |= (code & cap) << p.shift[j];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = i0 + (int64_t)u * stride;
            if (i < nrows) packed[i] = w[u];
        }
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(out_of_range, 1ULL);
}

// packed[i] -> key words [n + 1][stride] (values, NULL -> 0, then the null mask) as agg_wide_kernel builds them
__global__ __launch_bounds__(256) void key_unpack_kernel(PackParams p, const uint64_t* packed, int64_t n, uint64_t* dkey, int64_t stride) {
    const int64_t gstride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += gstride) {
        const uint64_t w = packed[i];
        uint64_t nullmask = 0;
        for (int j = 0; j < p.n; j++) {
            const uint64_t cap = (1ULL << p.bits[j]) - 1;
            const uint64_t code = (w >> p.shift[j]) & cap;
            const bool isnull = code == cap;
            dkey[(int64_t)j * stride + i] = isnull ? 0 : (p.dtab[j] ? p.dtab[j][code] : p.lo[j] + code);
            if (isnull) nullmask |= 1ULL << j;
        }
        dkey[(int64_t)p.n * stride + i] = nullmask;
    }
}


#include "vnm_agg_dense.inc"

// =======================================================================================================
// Device-side finalisation: BaseAggregate::Result / SummarizeGroups (base_aggregate.cpp:47-68) and the Summarize
// methods of the aggregate functions (agg_funcs.h:72-80 generic, :358-397 int64 sum, :482-491 + :519-540 AVG incl.
// the 128-bit divmod) evaluated per group ON THE DEVICE from the dense accumulator words, written as Arrow-layout
// buffers (typed values + validity bitmap).  The host finaliser (vnm_finalize.cpp) stays the authority for the one
// case that changes the column TYPE: an int64 / uint64 SUM that overflows 64 bits in some group promotes the whole
// column to decimal128 (:366-389) -- the kernel raises a flag for it and the caller uses vnm_agg_result_func.
// =======================================================================================================
struct FinArgs {
    FuncOut fo;
    const uint64_t* words[3];  // w_valid, w_a, w_b (nullptr when absent)
    int is_key, key_bit;       // key column: words[1] = key bits, words[0] = NULL-mask word
    int out_width;             // bytes per output value
    int out_f32;               // float32 output (AVG of 8 / 16-bit integers)
    int64_t n;
    void* out;
    unsigned long long* bitmap;   // (n + 63) / 64 words
    unsigned long long* ctl;      // [0] null count  [1] a 64-bit SUM overflowed  (this column's pair)
};

__device__ __forceinline__ void fin_store(void* out, int width, int64_t i, uint64_t bits) {
    switch (width) {
        case 1: ((uint8_t*)out)[i] = (uint8_t)bits; break;
        case 2: ((uint16_t*)out)[i] = (uint16_t)bits; break;
        case 4: ((uint32_t*)out)[i] = (uint32_t)bits; break;
        default: ((uint64_t*)out)[i] = bits; break;
    }
}
// Hugeint::TryCast<double>, huge_int.cpp:395-406 (including its 2^64-for-UINT64_MAX rounding)
__device__ __forceinline__ double fin_huge_to_double(uint64_t lower, int64_t upper) {
    if (upper == -1) return -(double)(0xFFFFFFFFFFFFFFFFULL - lower) - 1;
    return (double)lower + (double)upper * 18446744073709551615.0;
}

// one result cell: group i of one output column
__device__ __forceinline__ void fin_cell(const FinArgs& a, int64_t i, bool& valid, uint64_t& bits) {
    const int t = a.fo.in_type;
    {
        {
            if (a.is_key) {
                valid = !((a.words[0][i] >> a.key_bit) & 1ULL);
                bits = a.words[1][i];
            } else {
                const uint64_t cnt = a.words[0] ? a.words[0][i] : 1;
                const uint64_t wa = a.words[1][i];
                const uint64_t wb = a.words[2] ? a.words[2][i] : 0;
                valid = cnt > 0;
                // the (low 32, high 32) lanes of an int64 / uint64 SUM / AVG hold 2^32 - 1 inputs per group exactly: beyond, fail loudly
                if ((a.fo.func == VNM_SUM || a.fo.func == VNM_AVG) && (t == VNM_I64 || t == VNM_U64) && cnt >= (1ULL << 32))
                    __hip_atomic_fetch_or(&a.ctl[1], 2ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                switch (a.fo.func) {
                    case VNM_COUNT_STAR:
                    case VNM_COUNT: valid = true; bits = wa; break;
                    case VNM_MIN:
                    case VNM_MAX:
                        if (t == VNM_F64) bits = (uint64_t)__double_as_longlong(dec_f64(wa));
                        else if (t == VNM_F32) bits = (uint64_t)__float_as_uint((float)dec_f64(wa));
                        else if (type_is_unsigned(t)) bits = wa;
                        else bits = (uint64_t)dec_i64(wa);
                        break;
                    case VNM_SUM:
                        if (t == VNM_I64 || t == VNM_U64) {
                            // 128-bit two's complement sum = hi * 2^32 + lo (A_SUM_LO32 / A_SUM_HI32S|U lanes)
                            const uint64_t slo = wa + (wb << 32);
                            const int64_t shi = (t == VNM_I64 ? ((int64_t)wb >> 32) : (int64_t)(wb >> 32)) + (slo < wa ? 1 : 0);
                            bool fits;
                            if (t == VNM_I64) fits = (shi == 0 && slo <= 0x7FFFFFFFFFFFFFFFULL) || (shi == -1 && slo > 0x8000000000000000ULL);  // huge_int.cpp:334-355
                            else fits = shi == 0;
                            if (valid && !fits) __hip_atomic_fetch_or(&a.ctl[1], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            bits = slo;
                        } else if (type_is_float(t)) {
                            bits = (uint64_t)__double_as_longlong(fsum2(__longlong_as_double((long long)wa), a.words[2] ? __longlong_as_double((long long)wb) : 0.0));
                        } else bits = wa;  // int64 / uint64 accumulators of the narrow integers; time32 keeps its low 32 bits
                        break;
                    default: {  // VNM_AVG
                        double avg = 0.0;
                        if (valid) {
                            if (t == VNM_I64 || t == VNM_U64) {
                                uint64_t slo = wa + (wb << 32);
                                int64_t shi = (t == VNM_I64 ? ((int64_t)wb >> 32) : (int64_t)(wb >> 32)) + (slo < wa ? 1 : 0);
                                const bool neg = shi < 0;
                                if (neg) { slo = ~slo + 1; shi = ~shi + (slo == 0 ? 1 : 0); }  // magnitude
                                // (shi, slo) / cnt by 32-bit limbs; cnt < 2^32 rows per group by construction of the lanes
                                const uint32_t limb[4] = {(uint32_t)((uint64_t)shi >> 32), (uint32_t)shi, (uint32_t)(slo >> 32), (uint32_t)slo};
                                uint32_t ql[4];
                                uint64_t r = 0;
#pragma unroll
                                for (int k = 0; k < 4; k++) {
                                    const uint64_t cur = (r << 32) | limb[k];
                                    ql[k] = (uint32_t)(cur / cnt);
                                    r = cur % cnt;
                                }
                                uint64_t qlo = ((uint64_t)ql[2] << 32) | ql[3];
                                int64_t qhi = (int64_t)(((uint64_t)ql[0] << 32) | ql[1]);
                                uint64_t rlo = r;
                                int64_t rhi = 0;
                                if (neg) {  // C truncation: quotient and remainder take the sign of the dividend
                                    qlo = ~qlo + 1; qhi = ~qhi + (qlo == 0 ? 1 : 0);
                                    rlo = ~rlo + 1; rhi = ~rhi + (rlo == 0 ? 1 : 0);
                                }
                                avg = fin_huge_to_double(qlo, qhi) + fin_huge_to_double(rlo, rhi) / (double)cnt;  // agg_funcs.h:524-540
                            } else if (type_is_float(t)) {
                                avg = fsum2(__longlong_as_double((long long)wa), a.words[2] ? __longlong_as_double((long long)wb) : 0.0) / (double)cnt;
                            } else if (type_is_unsigned(t)) avg = (double)wa / (double)cnt;
                            else avg = (double)(int64_t)wa / (double)cnt;
                        }
                        bits = a.out_f32 ? (uint64_t)__float_as_uint((float)avg) : (uint64_t)__double_as_longlong(avg);
                        break;
                    }
                }
            }
        }
    }
}

// Every requested output column in ONE launch: a group's accumulator words are read once per column that uses them (the second
// reader hits L2) and the launch / tail cost is paid once.  ctl[2c] = NULL count of column c, ctl[2c + 1] = its SUM overflowed.
constexpr int FIN_MAX_COLS = 8;
struct FinMulti {
    FinArgs col[FIN_MAX_COLS];
    int n_cols;
    int64_t n;
};

__global__ __launch_bounds__(256) void agg_finalize_kernel(FinMulti m) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nround = (m.n + 63) & ~63LL;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nround; i += stride) {
        for (int c = 0; c < m.n_cols; c++) {
            const FinArgs& a = m.col[c];
            bool valid = false;
            uint64_t bits = 0;
            if (i < m.n) {
                fin_cell(a, i, valid, bits);
                fin_store(a.out, a.out_width, i, valid ? bits : 0);
            }
            const unsigned long long b = __ballot(valid);
            if (lane == 0) {
                a.bitmap[i >> 6] = b;
                const int64_t live = m.n - i >= 64 ? 64 : m.n - i;
                if (live != __popcll(b)) atomicAdd(&a.ctl[0], (unsigned long long)(live - __popcll(b)));
            }
        }
    }
}

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
namespace {

// returns 0 = done (run stored), 2 = not applicable / overflowed (caller uses the general path), 1 = error
// spill_out / n_spill_out (optional): entries that did not fit their partition region (heavy keys); the caller
// aggregates them with agg_hot_kernel<FROM_ENT> and owns the buffer.  Without them a full region fails the attempt.
int partitioned_aggregate(vnm_agg* h, const AggArgs& a, int64_t nrows, hipStream_t s, ulonglong2** spill_out = nullptr,
                          int64_t* n_spill_out = nullptr) {
    const int cus = device_info().num_cus;
    // final partitions sized for ~900 groups each: enough keys per partition that their sizes concentrate
    // (few keys per partition -> Poisson imbalance overflows the fixed-capacity regions), few enough for
    // the 2048-slot LDS table of pass 3
    // Generic programs with three or more accumulator words: a 2048-slot table is 65+ KB, i.e. one or two resident
    // workgroups per CU in a latency-bound pass (5 words, G=1e5: 10 ms).  They get 1024-slot tables and half the groups
    // per partition while two levels still provide enough partitions.
    const int64_t l1_cap = env_i64("VNM_AGG_PART_L1_MAX", 256) * 512;
    const bool small_tables = a.part_generic && (size_t)(PA_SLOTS + 1) * 8 * (1 + h->plan.n_words) > 54 * 1024 &&
                              h->hint / 450 < l1_cap && getenv("VNM_AGG_NO_SMALL_TABLES") == nullptr;
    int pa_slots = small_tables ? PA_SLOTS / 2 : PA_SLOTS;
    // ... and programs with so many words that even that table exceeds the LDS (three compensated float SUMs + their
    // counts + COUNT(*) = 10 words: 180 KB at 2048 slots) halve it until it fits: more, smaller partitions instead of
    // the HBM-atomics path (300 ms per 1e9 rows)
    while (a.part_generic && pa_slots > 256 && (size_t)(pa_slots + 1) * 8 * (1 + h->plan.n_words) > 150 * 1024) pa_slots /= 2;
    const int64_t per_final = env_i64("VNM_AGG_PART_GROUPS", 900 * pa_slots / PA_SLOTS);
    int64_t nfin = 2;
    while (nfin * per_final < h->hint) nfin *= 2;
    const int64_t l1_max = env_i64("VNM_AGG_PART_L1_MAX", 256);
    if (nfin > l1_max * 512) {
        // two levels give at most l1_max * 512 partitions: still fine while a partition's groups fit the LDS table
        if (h->hint / (l1_max * 512) > 1600 * pa_slots / PA_SLOTS) return 2;  // would need a third level
        nfin = l1_max * 512;
    }
    // entry = key + one value (16 bytes, tuned kernels) or key + 2-3 values (wide entries)
    const int E = a.part_wide ? 1 + h->plan.n_cols + (a.part_vmask ? 1 : 0) : 2;
    const bool wide = a.part_wide != 0;  // generic column accessors (any width, NULLs, any predicate column)
    const int64_t tile1 = wide ? pw_tile(E) : PT_TILE;
    const size_t ebytes = (size_t)E * 8;
    // twice the usual first-level fan-out still beats a second level that would only split in two (G = 3e5 sparse keys:
    // 512 partitions in one pass 7.4 + 1.9 ms; 256 x 2: 6.7 + 3.7 + 2.0)
    const int levels = nfin > std::min<int64_t>(l1_max * 2, PT_MAXP) ? 2 : 1;
    const int np1 = levels == 2 ? (int)l1_max : (int)nfin;
    const int np2 = levels == 2 ? (int)(nfin / l1_max) : 0;
    const int grid1 = (int)std::min<int64_t>((int64_t)cus * 2, (nrows + tile1 - 1) / tile1);
    const int split2 = std::max(2, (grid1 + PT_MAX_REGIONS - 1) / PT_MAX_REGIONS);  // pass-2 workgroups per partition
    const int64_t tiles_per_wg = ((nrows + tile1 - 1) / tile1 + grid1 - 1) / grid1;
    const int64_t rows_per_wg = tiles_per_wg * tile1;
    // region slack: a partition holds per_final / 2 ... per_final keys, so its share of the rows scatters by 1 / sqrt(keys)
    // around the mean (small LDS tables -> few keys per partition -> up to +42 %; measured: 225 keys per partition
    // overflowed the 25 % regions of pass 2 at G = 3e5 and 1e6)
    const double slack = std::max(0.0, 4.5 / std::sqrt((double)std::max<int64_t>(per_final, 16) / 2.0));
    const int64_t cap1 = rows_per_wg / np1 + (int64_t)((double)(rows_per_wg / np1) * std::max(0.2, slack)) + 512;
    PoolScope pool;   // every block of this attempt; the ones handed on are keep()-ed
    unsigned long long* flags = (unsigned long long*)pool.take(64);
    ulonglong2* e1 = (ulonglong2*)pool.take((size_t)np1 * grid1 * cap1 * ebytes);
    uint32_t* c1 = (uint32_t*)pool.take((size_t)np1 * grid1 * 4);
    if (!flags || !e1 || !c1) return 1;
    VNM_HIP(hipMemsetAsync(flags, 0, 64, s));
    const int64_t spill_cap = spill_out ? nrows / 2 + (1 << 20) : 0;
    ulonglong2* spill = spill_out ? (ulonglong2*)pool.take((size_t)spill_cap * (wide ? ebytes : 16)) : nullptr;   // wide: [spill_cap][E] words
    if (spill_out && !spill) return 1;
    PartArgs p1{};
    p1.kp = (const uint64_t*)a.keys[0].values + a.keys[0].offset;
    p1.vp = h->plan.n_cols ? (const double*)a.cols[0].values + a.cols[0].offset : (const double*)p1.kp;
    p1.has_expr = a.has_expr; p1.expr = a.expr;
    p1.pp = h->pred_set ? (const double*)a.pred.values + a.pred.offset : nullptr;
    p1.has_pred = h->pred_set; p1.pred_is_v = a.hot_pred_is_v; p1.op = a.p.op; p1.thr = a.p.dval;
    p1.nrows = nrows;
    p1.out_entries = e1; p1.out_counts = c1; p1.out_cap = cap1;
    int lg1 = 0;
    while ((1 << lg1) < np1) lg1++;
    p1.nparts = np1; p1.shift = 32 - lg1; p1.flags = flags;
    p1.debug = (int)env_i64("VNM_PART_DEBUG", 0);
    p1.spill = spill; p1.spill_cap = spill_cap;
    p1.nval = h->plan.n_cols;
    p1.has_vmask = a.part_vmask;
    if (wide) {
        for (int c = 0; c < h->plan.n_cols; c++) p1.vcols[c] = a.cols[c];
        p1.wp = a.p;
        p1.wpred = a.pred;
    }
    {
        KernelTimer timer("agg_part_scatter1", s);
        if (!wide) part_scatter_kernel<true><<<grid1, PT_BLOCK, 0, s>>>(p1);
        else {
            const size_t lds = (size_t)pw_tile(E) * ebytes;
#define VNM_PSW(FR, E_, IT_, GRID, ARGS)                                                                             \
    do {                                                                                                             \
        VNM_HIP(hipFuncSetAttribute((const void*)part_scatter_wide_kernel<FR, E_, IT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        part_scatter_wide_kernel<FR, E_, IT_><<<GRID, PT_BLOCK, lds, s>>>(ARGS);                                     \
    } while (0)
            if (E == 1) VNM_PSW(true, 1, 8, grid1, p1); else if (E == 2) VNM_PSW(true, 2, 8, grid1, p1); else if (E == 3) VNM_PSW(true, 3, 4, grid1, p1); else if (E == 4) VNM_PSW(true, 4, 4, grid1, p1);
            else if (E == 5) VNM_PSW(true, 5, 2, grid1, p1); else if (E == 6) VNM_PSW(true, 6, 2, grid1, p1); else VNM_PSW(true, 7, 2, grid1, p1);
        }
    }
    VNM_HIP(hipGetLastError());
    // A region overflowed (skewed keys, or more rows per partition than the hint implied): stop here.  Carrying on
    // would aggregate partitions that are about to be thrown away -- and the partition holding a heavy key is
    // processed by ONE workgroup (measured: 1.6 s for 1e8 rows with a power-law key distribution).
    auto overflowed = [&]() -> int {
        unsigned long long f = 0;
        if (hipMemcpyAsync(&f, flags, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
        return f ? 1 : 0;
    };
    {
        int ov = overflowed();
        if (ov) return ov < 0 ? set_error("aggregate: partition pass failed") : 2;
    }

    const ulonglong2* fin_e = e1;
    const uint32_t* fin_c = c1;
    int64_t fin_cap = cap1, nfinal = np1;
    int fin_regions = grid1;
    ulonglong2* e2 = nullptr;
    uint32_t* c2 = nullptr;
    if (levels == 2) {
        // worst case: every row survived and spread evenly; 25 % slack + constant
        const int64_t per_pg = (int64_t)grid1 * rows_per_wg / np1 / split2;
        const int64_t cap2 = per_pg / np2 + (int64_t)((double)(per_pg / np2) * std::max(0.25, slack)) + 256;
        e2 = (ulonglong2*)pool.take((size_t)np1 * np2 * split2 * cap2 * ebytes);
        c2 = (uint32_t*)pool.take((size_t)np1 * np2 * split2 * 4);
        if (!e2 || !c2) return 1;
        PartArgs p2{};
        p2.in_entries = e1; p2.in_counts = c1; p2.in_cap = cap1; p2.in_regions = grid1; p2.in_split = split2;
        p2.out_entries = e2; p2.out_counts = c2; p2.out_cap = cap2;
        p2.nparts = np2; p2.shift = 15; p2.flags = flags;  // hash bits [23:15] (pass 1 used [31:24])
        p2.debug = p1.debug;
        p2.spill = spill; p2.spill_cap = spill_cap;
        p2.nval = h->plan.n_cols;
        p2.has_vmask = a.part_vmask;
        {
            KernelTimer timer("agg_part_scatter2", s);
            if (!wide) part_scatter_kernel<false><<<np1 * p2.in_split, PT_BLOCK, 0, s>>>(p2);
            else {
                const bool big2 = E <= 2 && np2 >= 128;
                const size_t lds = (size_t)PT_BLOCK * (big2 ? 8 : (E <= 4 ? 4 : 2)) * ebytes;
                const int g2 = np1 * p2.in_split;
                if (E == 1) { if (big2) VNM_PSW(false, 1, 8, g2, p2); else VNM_PSW(false, 1, 4, g2, p2); }
                else if (E == 2) { if (big2) VNM_PSW(false, 2, 8, g2, p2); else VNM_PSW(false, 2, 4, g2, p2); }
                else if (E == 3) VNM_PSW(false, 3, 4, g2, p2); else if (E == 4) VNM_PSW(false, 4, 4, g2, p2);
                else if (E == 5) VNM_PSW(false, 5, 2, g2, p2); else if (E == 6) VNM_PSW(false, 6, 2, g2, p2); else VNM_PSW(false, 7, 2, g2, p2);
            }
        }
#undef VNM_PSW
        VNM_HIP(hipGetLastError());
        {
            int ov = overflowed();
            if (ov) return ov < 0 ? set_error("aggregate: partition pass failed") : 2;
        }
        fin_e = e2; fin_c = c2; fin_cap = cap2; nfinal = (int64_t)np1 * np2; fin_regions = p2.in_split;
    }

    // few final partitions cannot fill the chip with one workgroup each: split them and merge through the
    // HBM table (only legal while the table holds nothing else, so a failed attempt can simply be dropped)
    int splits = 1;
    if (nfinal < (int64_t)cus * 4) {
        splits = (int)std::min<int64_t>(256, ((int64_t)cus * 8 + nfinal - 1) / nfinal);
        if (splits > fin_regions) splits = fin_regions;
    }
    // Merging straight into the HBM table is only legal while the table holds nothing else (a failed attempt is then
    // simply dropped).  With groups already in the table -- every batch of a stream after the first -- the split
    // workgroups write their PARTIAL groups (a key may appear once per split) into a dense run instead, which is folded
    // into the table like any other run: no flush storms for streamed input with 2.4 K ... 900 K groups.
    const bool to_table = splits > 1 && !(h->have_table || h->have_run);
    const bool dup_run = splits > 1 && !to_table;
    // dense output sized from the hint (guarded in the kernel)
    const int64_t dstride = to_table ? 2 : std::min<int64_t>(nrows, (h->hint * 2 + (1 << 20)) * (dup_run ? splits : 1)) + 2;
    uint64_t* rk = (uint64_t*)pool.take((size_t)dstride * 8 * 2);
    uint64_t* ra = (uint64_t*)pool.take((size_t)dstride * 8 * h->plan.n_words);
    if (!rk || !ra) return 1;
    unsigned long long* dir = nullptr;
    if (!to_table && !dup_run) {   // the partition directory only describes runs with ONE workgroup per partition
        dir = (unsigned long long*)pool.take((size_t)nfinal * 16);
        if (!dir) return 1;
    }
    PartAggArgs pa{};
    pa.dir = dir;
    pa.splits = splits;
    pa.to_table = to_table;
    if (to_table) {
        VNM_TRY(ensure_table(h, nrows, s));
        // every resident workgroup may pass the room check and then insert a full LDS table at once
        const int64_t g3max = std::min<int64_t>(nfinal * splits, (int64_t)cus * 4);
        const uint64_t want = pow2_at_least(std::max<uint64_t>((uint64_t)h->hint * 4 + 8192, (uint64_t)(g3max * PA_SLOTS * 2)));
        if (h->g.cap < want) VNM_TRY(table_grow(h, want, s));
        pa.g = h->g;
        pa.table_limit = (int64_t)(h->g.cap * 8 / 10) - g3max * PA_SLOTS;
    }
    pa.entries = fin_e; pa.counts = fin_c; pa.cap = fin_cap; pa.regions = fin_regions; pa.nfinal = nfinal;
    pa.w_rows = a.hot_w_rows; pa.w_valid = a.hot_w_valid; pa.w_sum = a.hot_w_sum; pa.n_words = h->plan.n_words;
    pa.w_lo = a.hot_comp && a.hot_w_sum >= 0 ? a.hot_w_sum + 1 : -1;
    pa.comp = a.hot_comp;
    pa.dkey = rk; pa.dacc = ra; pa.dstride = dstride; pa.flags = flags;
    {
        KernelTimer timer("agg_part_final", s);
        // one resident set of workgroups (they loop over the partitions): a grid larger than what fits leaves a
        // second, partly filled round
        int g3 = (int)std::min<int64_t>(nfinal * splits, (int64_t)cus * 4);
        auto fit_grid = [&](const void* fn, size_t lds) {
            int occ = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, PA_BLOCK, lds) == hipSuccess && occ > 0)
                g3 = (int)std::min<int64_t>(nfinal * splits, (int64_t)cus * std::min(occ, (int)env_i64("VNM_PA_OCC", 8)));
        };
        if (a.part_generic) {
            pa.n_ops = h->plan.n_ops;
            pa.vtype = a.part_vtype;
            for (int o = 0; o < h->plan.n_ops; o++) pa.ops[o] = h->plan.ops[o];
            for (int w = 0; w < h->plan.n_words; w++) pa.merge[w] = h->plan.merge[w];
            const size_t lds_bytes = (size_t)(pa_slots + 1) * 8 * (1 + h->plan.n_words);
            pa.slots = pa_slots;
            pa.ent_words = E;
            pa.wide = wide;
            pa.has_vmask = a.part_vmask;
            for (int c = 0; c < 6; c++) pa.vtypes[c] = a.part_vtypes[c];
            pa.nval = h->plan.n_cols;
            pa.w_rows_g = -1;
            pa.use_table = getenv("VNM_AGG_NO_PART_TABLE") == nullptr;
            for (int c = 0; c < 6; c++) pa.wpack[c] = ~0ULL;
            for (int o = 0; o < h->plan.n_ops && pa.use_table; o++) {
                const AccOp& op = h->plan.ops[o];
                if (op.kind == A_COUNT_ROWS) { if (pa.w_rows_g >= 0) pa.use_table = 0; pa.w_rows_g = op.word; continue; }
                const int c = op.col < 0 ? 0 : op.col;
                if (c > 5 || op.kind < 0 || op.kind > A_MAX || op.word >= 63 || ((pa.wpack[c] >> (6 * op.kind)) & 63ULL) != 63) { pa.use_table = 0; break; }
                pa.wpack[c] = (pa.wpack[c] & ~(63ULL << (6 * op.kind))) | ((unsigned long long)op.word << (6 * op.kind));
            }
#define VNM_PAG(E_, T_)                                                                                              \
    do {                                                                                                             \
        VNM_HIP(hipFuncSetAttribute((const void*)part_agg_generic_kernel<E_, T_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
        fit_grid((const void*)part_agg_generic_kernel<E_, T_>, lds_bytes);                                           \
        part_agg_generic_kernel<E_, T_><<<g3, PA_BLOCK, lds_bytes, s>>>(pa);                                         \
    } while (0)
            if (pa.use_table) { if (E == 1) VNM_PAG(1, true); else if (E == 2) VNM_PAG(2, true); else if (E == 3) VNM_PAG(3, true); else if (E == 4) VNM_PAG(4, true); else if (E == 5) VNM_PAG(5, true); else if (E == 6) VNM_PAG(6, true); else VNM_PAG(7, true); }
            else { if (E == 1) VNM_PAG(1, false); else if (E == 2) VNM_PAG(2, false); else if (E == 3) VNM_PAG(3, false); else if (E == 4) VNM_PAG(4, false); else if (E == 5) VNM_PAG(5, false); else if (E == 6) VNM_PAG(6, false); else VNM_PAG(7, false); }
#undef VNM_PAG
        } else {
            fit_grid((const void*)part_agg_kernel, 0);
            part_agg_kernel<<<g3, PA_BLOCK, 0, s>>>(pa);
        }
    }
    VNM_HIP(hipGetLastError());
    unsigned long long fl[3];
    VNM_HIP(hipMemcpyAsync(fl, flags, 24, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    pool.done(e1); pool.done(c1); pool.done(e2); pool.done(c2); pool.done(flags);
    if (fl[0]) {  // more groups than hinted / spill buffer full: use the general path for this batch
        if (to_table) { table_free(&h->g); h->have_table = false; }  // drop the partial merge
        return 3;  // only the final pass fails this way: the hint was too small, more partitions would do
    }
    if (spill_out) {
        if (fl[2]) { pool.keep(spill); *spill_out = spill; *n_spill_out = (int64_t)fl[2]; }   // the caller owns it now
        else { *spill_out = nullptr; *n_spill_out = 0; }
    }
    if (to_table) return 0;  // the groups already live in the HBM table
    pool.keep(rk); pool.keep(ra); pool.keep(dir);   // the run belongs to the handle
    h->run_key = rk; h->run_acc = ra; h->run_stride = dstride; h->run_n = (int64_t)fl[1];
    h->run_dir = dir; h->run_nfin = dir ? nfinal : 0;
    h->have_run = true;
    if (dup_run) {   // keys repeat inside this run: it must never be handed out as a result, fold it into the table now
        if (merge_run_into_table(h, s)) return 1;
    }
    return 0;
}


// ---- dense-key partitioned path: host side (kernels in vnm_agg_dense.inc) ------------------------------------
// Sample the key range of the first large batch and derive the code map.  dense_state = 1 when the (widened) range fits
// DP_MAX_BITS bits and is large enough to fill the chip with final partitions.
int plan_dense_from_range(vnm_agg* h, int key_type, uint64_t got0, uint64_t got1);
int plan_dense(vnm_agg* h, const vnm_dcol& key, int64_t nrows, hipStream_t s) {
    if (h->range_given) return 0;   // vnm_agg_set_dense_range: the map stays the one all ranks derived
    h->dense_state = -1;
    if (key.type != VNM_I64 && key.type != VNM_U64) return 0;
    const uint64_t sign = key.type == VNM_I64 ? 0x8000000000000000ULL : 0ULL;
    const uint64_t* kp = (const uint64_t*)key.values + key.offset;
    unsigned long long* d = (unsigned long long*)pool_alloc(64);
    if (!d) return 1;
    const unsigned long long init[2] = {~0ULL, 0ULL};
    unsigned long long got[2];
    VNM_HIP(hipMemcpyAsync(d, init, 16, hipMemcpyHostToDevice, s));
    const int64_t m = std::min<int64_t>(nrows, 1 << 18);
    const int grid = (int)std::min<int64_t>((m + 255) / 256, (int64_t)device_info().num_cus * 4);
    dense_sample_range_kernel<<<grid, 256, 0, s>>>(kp, nrows, m, sign, d, h->kn_valid, h->kn_off);
    VNM_HIP(hipGetLastError());
    VNM_HIP(hipMemcpyAsync(got, d, 16, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    pool_free(d);
    if (got[0] > got[1]) return 0;
    return plan_dense_from_range(h, key.type, got[0], got[1]);
}

// The code map from a (sampled) key range [got0, got1] of order-preserving unsigned images.
int plan_dense_from_range(vnm_agg* h, int key_type, uint64_t got0, uint64_t got1) {
    h->dense_state = -1;
    if (key_type != VNM_I64 && key_type != VNM_U64) return 0;
    const uint64_t sign = key_type == VNM_I64 ? 0x8000000000000000ULL : 0ULL;
    const unsigned long long got[2] = {got0, got1};
    // widen by 1/8 of the sampled span on both sides (the sample misses the true extremes), then centre the range in
    // the next power of two
    const uint64_t span_s = got[1] - got[0];
    if (span_s >= (1ULL << DP_MAX_BITS)) return 0;
    // (small ranges: a tighter margin, so that up to ~7000 sampled codes still fit the 2^13-slot scan table)
    const uint64_t margin = span_s < (1ULL << 13) ? span_s / 16 + 64 : span_s / 8 + 4096;
    uint64_t lo = got[0] > margin ? got[0] - margin : 0;
    uint64_t hi = got[1] < ~0ULL - margin ? got[1] + margin : ~0ULL;
    int bits = 1;
    while (bits < 64 && ((hi - lo) >> bits) != 0) bits++;
    if (bits > DP_MAX_BITS) return 0;
    // ranges below 2^20 leave too few final partitions to fill the chip with one workgroup each: their final pass splits
    // every partition over several workgroups and merges the partial tables (dpart_merge_kernel).  Up to 2^13 codes need
    // no partitioning at all (dense_state = 2).
    h->dense_rlo = lo; h->dense_rhi = hi;
    const bool small = bits < (int)env_i64("VNM_DENSE_MIN_BITS", 14);   // at most 2^13 codes: the direct-addressed LDS scan (dscan_kernel)
    if (small) bits = DP_TBITS_MAX;   // always the 2^13-slot table: one 1024-thread workgroup per CU measured fastest at every G
                                      // (G = 300 / 1000: 2.83 / 2.84 ms; 2^12 slots, two 512-thread workgroups: 3.04 / 3.07; 2^11, four: 3.50 / 3.65)
    const uint64_t extra = ((1ULL << bits) - 1) - (hi - lo);
    lo = lo > extra / 2 ? lo - extra / 2 : 0;
    DenseMap& mp = h->dmap;
    mp.lo_u = lo;
    mp.sign = sign;
    mp.bits = bits;
    mp.mask = (uint32_t)((1ULL << bits) - 1);
    mp.mul = (uint32_t)((double)(1ULL << bits) * 0.6180339887498949) | 1u;  // Fibonacci hashing on `bits` bits
    uint32_t inv = mp.mul;                                                     // Newton: inverse modulo 2^32
    for (int it = 0; it < 5; it++) inv *= 2u - mp.mul * inv;
    mp.mul_inv = inv;
    h->dense_span = (int64_t)1 << bits;
    h->dense_state = small ? 2 : 1;
    return 0;
}

// Generic accumulator programs on the dense paths (dgen_* kernels): every AccKind at most once over at most one plain
// 8-byte input column, at most one COUNT(*) word (it lives in the slot's row counter).  Fills the program part of `g`;
// false = not expressible.
bool dgen_program(const vnm_agg* h, const AggArgs& a, DGenArgs* g) {
    const AggPlan& p = h->plan;
    if (p.n_cols > 1 || p.n_words > AGG_MAX_WORDS) return false;
    g->has_val = p.n_cols == 1;
    g->vtype = p.n_cols == 1 ? a.cols[0].type : VNM_F64;
    g->comp = 0;
    g->wpack = ~0ULL;
    g->n_words = p.n_words;
    int w_rows = -1;
    for (int o = 0; o < p.n_ops; o++) {
        const AccOp& op = p.ops[o];
        if (op.kind == A_COUNT_ROWS) { if (w_rows >= 0) return false; w_rows = op.word; }
    }
    int n_lds = 0;
    for (int w = 0; w < p.n_words; w++) {
        g->merge[w] = p.merge[w];
        g->lds_word[w] = w == w_rows ? -1 : n_lds++;
        if (p.merge[w] == M_ADD_F64C) g->comp = 1;
    }
    if (n_lds > DG_MAX_WORDS - 1) return false;
    g->n_lds = n_lds;
    for (int o = 0; o < p.n_ops; o++) {
        const AccOp& op = p.ops[o];
        if (op.kind == A_COUNT_ROWS) continue;
        if (op.col != 0 || op.kind < 0 || op.kind > A_MAX || op.word >= 63 || ((g->wpack >> (6 * op.kind)) & 63ULL) != 63) return false;
        g->wpack = (g->wpack & ~(63ULL << (6 * op.kind))) | ((unsigned long long)g->lds_word[op.word] << (6 * op.kind));
    }
    // a compensated sum's lo word must follow its hi word in LDS as well
    for (int w = 0; w + 1 < p.n_words; w++)
        if (p.merge[w] == M_ADD_F64C && (g->lds_word[w] < 0 || g->lds_word[w + 1] != g->lds_word[w] + 1)) return false;
    return true;
}
// bytes of LDS per slot of a generic table
inline int dgen_slot_bytes(const DGenArgs& g) { return 8 * g.n_lds + 4; }

// Ranges of at most 2^13 codes: one scan with the whole table in LDS.  Same return convention as the partitioned variant.
int dense_scan_aggregate(vnm_agg* h, const AggArgs& a, int64_t nrows, hipStream_t s, ulonglong2** spill_out, int64_t* n_spill_out, bool generic = false,
                         uint64_t** nspill_out = nullptr, int64_t* n_nspill_out = nullptr) {
    const int cus = device_info().num_cus;
    if (h->dmap.bits != DP_TBITS_MAX) return 2;
    const int block = 1024;
    if (nspill_out) { *nspill_out = nullptr; *n_nspill_out = 0; }
    if (generic) {
        DGenArgs g{};
        if (!dgen_program(h, a, &g)) return 2;
        const bool vn = g.has_val && a.cols[0].validity != nullptr;   // nullable value column
        if (vn && !nspill_out) return 2;
        // the largest table (<= 2^13 slots) that fits 144 KB of LDS must hold the sampled range
        int tb = DP_TBITS_MAX;
        while (tb > 9 && ((size_t)dgen_slot_bytes(g) << tb) > 144 * 1024) tb--;
        const uint64_t need = h->dense_rhi - h->dense_rlo;
        if (need >= (1ULL << tb)) return 2;
        const int slots = 1 << tb;
        const size_t lds = (size_t)slots * dgen_slot_bytes(g);
        const int per_cu = lds <= 72 * 1024 ? 2 : 1;
        const int grid = (int)std::min<int64_t>((int64_t)cus * per_cu, std::max<int64_t>(1, (nrows / 2 + block - 1) / block));
        PoolScope pool;
        unsigned long long* flags = (unsigned long long*)pool.take(64);
        uint64_t* pw = (uint64_t*)pool.take(std::max<size_t>(8, (size_t)grid * g.n_lds * slots * 8));
        uint32_t* pc = (uint32_t*)pool.take((size_t)grid * slots * 4);
        const int64_t spill_cap = nrows / 2 + (1 << 20);
        ulonglong2* spill = (ulonglong2*)pool.take((size_t)spill_cap * 16);
        const int64_t nspill_cap = vn ? nrows / 4 + (1 << 20) : 0;
        uint64_t* nspill = vn ? (uint64_t*)pool.take((size_t)nspill_cap * 8) : nullptr;
        const int64_t dstride = slots + 2;
        uint64_t* rk = (uint64_t*)pool.take((size_t)dstride * 8 * 2);
        uint64_t* ra = (uint64_t*)pool.take((size_t)dstride * 8 * std::max(1, h->plan.n_words));
        if (!flags || !pw || !pc || !spill || !rk || !ra || (vn && !nspill)) return 1;
        VNM_HIP(hipMemsetAsync(flags, 0, 64, s));
        g.map = h->dmap; g.map.mul = 1; g.map.mul_inv = 1;
        if (tb != DP_TBITS_MAX) {   // a smaller table: centre the sampled range in it
            const uint64_t extra = ((1ULL << tb) - 1) - need;
            g.map.lo_u = h->dense_rlo > extra / 2 ? h->dense_rlo - extra / 2 : 0;
            g.map.bits = tb;
            g.map.mask = (uint32_t)((1ULL << tb) - 1);
        }
        g.tbits = tb;
        g.kp = (const uint64_t*)a.keys[0].values + a.keys[0].offset;
        g.vp = g.has_val ? (const double*)a.cols[0].values + a.cols[0].offset : nullptr;
        g.pp = h->pred_set ? (const double*)a.pred.values + a.pred.offset : nullptr;
        g.has_pred = h->pred_set; g.pred_is_v = a.hot_pred_is_v; g.op = a.p.op; g.thr = a.p.dval;
        g.nrows = nrows;
        g.spill = spill; g.spill_cap = spill_cap;
        if (vn) { g.vvalid = a.cols[0].validity; g.voff = a.cols[0].offset; g.nspill = nspill; g.nspill_cap = nspill_cap; }
        g.dkey = rk; g.dacc = ra; g.dstride = dstride; g.flags = flags;
        g.nfinal = 1; g.splits = grid; g.part_w = pw; g.part_cnt = pc;
        {
            KernelTimer timer("agg_scan", s);
            if (vn) {
                VNM_HIP(hipFuncSetAttribute((const void*)dgen_scan_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                dgen_scan_kernel<true, true><<<grid, block, lds, s>>>(g);
            } else if (g.has_val) {
                VNM_HIP(hipFuncSetAttribute((const void*)dgen_scan_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                dgen_scan_kernel<true><<<grid, block, lds, s>>>(g);
            } else {
                VNM_HIP(hipFuncSetAttribute((const void*)dgen_scan_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                dgen_scan_kernel<false><<<grid, block, lds, s>>>(g);
            }
            dgen_merge_kernel<<<std::max(1, slots / 512), 512, 0, s>>>(g);
        }
        VNM_HIP(hipGetLastError());
        unsigned long long fl[4];
        VNM_HIP(hipMemcpyAsync(fl, flags, 32, hipMemcpyDeviceToHost, s));
        VNM_HIP(hipStreamSynchronize(s));
        if (fl[0]) return 2;
        if (fl[2]) { pool.keep(spill); *spill_out = spill; *n_spill_out = (int64_t)fl[2]; }
        else { *spill_out = nullptr; *n_spill_out = 0; }
        if (vn && fl[3]) { pool.keep(nspill); *nspill_out = nspill; *n_nspill_out = (int64_t)fl[3]; }
        if ((int64_t)(fl[2] + fl[3]) > nrows / 16) h->dense_state = -1;
        pool.keep(rk); pool.keep(ra);
        h->run_key = rk; h->run_acc = ra; h->run_stride = dstride; h->run_n = (int64_t)fl[1];
        h->run_dir = nullptr; h->run_nfin = 0;
        h->have_run = true;
        return 0;
    }
    const int tb = DP_TBITS_MAX;
    const int slots = 1 << tb;
    const int grid = (int)std::min<int64_t>((int64_t)cus, std::max<int64_t>(1, (nrows / 2 + block - 1) / block));
    // The batch's per-workgroup tables are added into the table of the STREAM (h->scan_pending, dscan_accumulate_kernel); the groups
    // are written when something else needs them (flush_scan_pending).  59 x 2^24-row batches, G = 1000: 17.1 -> 5.5 ms per 1e9 rows.
    DenseMap map = h->dmap;
    map.mul = 1; map.mul_inv = 1;
    if (h->scan_pending && memcmp(&h->scan_pending->df.map, &map, sizeof(DenseMap)) != 0) VNM_TRY(flush_scan_pending(h, s));
    const bool comp = a.hot_comp && a.hot_w_sum >= 0;
    if (!h->scan_pending) {
        DScanPending* sp = new DScanPending();
        sp->slots = slots;
        sp->df.map = map;
        sp->df.w_rows = a.hot_w_rows; sp->df.w_valid = a.hot_w_valid; sp->df.w_sum = a.hot_w_sum;
        sp->df.w_lo = comp ? a.hot_w_sum + 1 : -1;
        sp->t.sum = (double*)pool_alloc((size_t)slots * 8);
        sp->t.lo = (double*)pool_alloc((size_t)slots * 8);
        sp->t.cnt = (unsigned long long*)pool_alloc((size_t)slots * 8);
        sp->part_sum = (uint64_t*)pool_alloc((size_t)cus * slots * 8);
        sp->part_lo = (float*)pool_alloc((size_t)cus * slots * 4);
        sp->part_cnt = (uint32_t*)pool_alloc((size_t)cus * slots * 4);
        sp->flags = (unsigned long long*)pool_alloc(64);
        if (!sp->t.sum || !sp->t.lo || !sp->t.cnt || !sp->part_sum || !sp->part_lo || !sp->part_cnt || !sp->flags) { delete sp; return 1; }
        fill_u64_kernel<<<(slots + 255) / 256, 256, 0, s>>>((uint64_t*)sp->t.sum, F64_NEG_ZERO, (int64_t)slots);   // (sums start at -0.0: merge_init)
        if (hipGetLastError() != hipSuccess || hipMemsetAsync(sp->t.lo, 0, (size_t)slots * 8, s) != hipSuccess ||
            hipMemsetAsync(sp->t.cnt, 0, (size_t)slots * 8, s) != hipSuccess) { delete sp; return set_error("aggregate: memset of the stream table failed"); }
        h->scan_pending = sp;
    }
    DScanPending* sp = h->scan_pending;
    const int64_t spill_cap = nrows / 2 + (1 << 20);
    ulonglong2* spill = (ulonglong2*)pool_alloc((size_t)spill_cap * 16);
    if (!spill) return 1;
    PoolSlotGuard<ulonglong2> spill_guard(&spill);
    VNM_HIP(hipMemsetAsync(sp->flags, 0, 64, s));
    DScanArgs d{};
    d.map = map;
    d.kp = (const uint64_t*)a.keys[0].values + a.keys[0].offset;
    d.vp = (const double*)a.cols[0].values + a.cols[0].offset;
    d.has_expr = a.has_expr; d.expr = a.expr;
    d.pp = h->pred_set ? (const double*)a.pred.values + a.pred.offset : nullptr;
    d.has_pred = h->pred_set; d.pred_is_v = a.hot_pred_is_v; d.op = a.p.op; d.thr = a.p.dval;
    d.nrows = nrows;
    d.comp = comp;
    d.part_sum = sp->part_sum; d.part_lo = sp->part_lo; d.part_cnt = sp->part_cnt;
    d.flags = sp->flags; d.spill = spill; d.spill_cap = spill_cap;
    unsigned long long fl[3];
    {
        KernelTimer timer("agg_scan", s);
        dscan_kernel<DP_TBITS_MAX><<<grid, block, 0, s>>>(d);
        VNM_HIP(hipGetLastError());
        VNM_HIP(hipMemcpyAsync(fl, sp->flags, 24, hipMemcpyDeviceToHost, s));
        VNM_HIP(hipStreamSynchronize(s));
        if (fl[0]) return 2;   // spill buffer full or a compensation term beyond float range: the batch goes another way, the stream's table is untouched
        dscan_accumulate_kernel<<<slots / 64, 512, 0, s>>>(sp->part_sum, sp->part_lo, sp->part_cnt, grid, slots, sp->t);
        VNM_HIP(hipGetLastError());
    }
    if (fl[2]) { *spill_out = spill; *n_spill_out = (int64_t)fl[2]; spill = nullptr; }
    else { *spill_out = nullptr; *n_spill_out = 0; }
    if ((int64_t)fl[2] > nrows / 16) h->dense_state = -1;  // the sampled range does not describe the data: stop trying
    return 0;
}

// The plain (one workgroup per partition) final pass of the dense path in one of its output modes.
int launch_dense_final(const DFinalArgs& df, int tb, int out, bool lo64, hipStream_t s) {
    const int cus = device_info().num_cus;
    KernelTimer timer("agg_part_final", s);
#define VNM_DFIN(TB_, OUT_, LOT_)                                                                                      \
    do {                                                                                                              \
        const int blk = TB_ >= 13 ? 1024 : (TB_ == 12 ? VNM_DF12_BLOCK : 512);                                       \
        int occ = 0;                                                                                                  \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)dpart_final_kernel<uint16_t, TB_, false, OUT_, LOT_>, blk, 0) != hipSuccess || occ < 1) occ = 1; \
        const int g3 = (int)std::min<int64_t>(df.nfinal, (int64_t)cus * std::min(occ, (int)env_i64("VNM_PA_OCC", 8))); \
        dpart_final_kernel<uint16_t, TB_, false, OUT_, LOT_><<<g3, blk, 0, s>>>(df);                                  \
    } while (0)
#define VNM_DFIN_O(TB_)                                                                                                \
    do {                                                                                                              \
        if (out == DF_COLS) VNM_DFIN(TB_, DF_COLS, float); else if (out == DF_TABLE) VNM_DFIN(TB_, DF_TABLE, float);   \
        else VNM_DFIN(TB_, DF_RUN, float);                                                                            \
    } while (0)
    if (lo64) {   // compensation terms beyond the float range (|sum| > ~1e54): 64-bit terms, tables of at most 2^12 slots
        if (tb == 9) { if (out == DF_COLS) VNM_DFIN(9, DF_COLS, double); else VNM_DFIN(9, DF_RUN, double); }
        else if (tb == 10) { if (out == DF_COLS) VNM_DFIN(10, DF_COLS, double); else VNM_DFIN(10, DF_RUN, double); }
        else if (tb == 11) { if (out == DF_COLS) VNM_DFIN(11, DF_COLS, double); else VNM_DFIN(11, DF_RUN, double); }
        else if (tb == 12) { if (out == DF_COLS) VNM_DFIN(12, DF_COLS, double); else VNM_DFIN(12, DF_RUN, double); }
        else return set_error("aggregate: no 64-bit compensation variant for this table size (internal error)");
    } else if (tb == 9) VNM_DFIN_O(9); else if (tb == 10) VNM_DFIN_O(10); else if (tb == 11) VNM_DFIN_O(11); else if (tb == 12) VNM_DFIN_O(12); else VNM_DFIN_O(13);
#undef VNM_DFIN_O
#undef VNM_DFIN
    VNM_HIP(hipGetLastError());
    return 0;
}

// The waiting batches of a stream (h->segs_active) as the segment table of one launch whose tiles hold `tile` rows: every segment
// starts a tile of its own.  The table lives in `pool` (freed in stream order); its host copy stays in the handle until the next one.
int upload_segs(vnm_agg* h, int64_t tile, const VSeg** dev, int* nseg, int64_t* ntiles, PoolScope& pool, hipStream_t s) {
    const std::vector<vnm_agg::QBatch>& q = *h->segs_active;
    h->seg_host.resize(q.size());
    int64_t t = 0;
    for (size_t i = 0; i < q.size(); i++) {
        VSeg& g = h->seg_host[i];
        g.kp = (const uint64_t*)q[i].key.values + q[i].key.offset;
        g.vp = (const uint64_t*)q[i].col.values + q[i].col.offset;
        g.pp = h->pred_set ? (const double*)q[i].pred.values + q[i].pred.offset : nullptr;
        g.vvalid = nullptr; g.voff = 0;
        g.nrows = q[i].nrows;
        g.first_tile = t;
        t += (q[i].nrows + tile - 1) / tile;
    }
    VSeg* d = (VSeg*)pool.take(q.size() * sizeof(VSeg));
    if (!d) return 1;
    VNM_HIP(hipMemcpyAsync(d, h->seg_host.data(), q.size() * sizeof(VSeg), hipMemcpyHostToDevice, s));
    *dev = d; *nseg = (int)q.size(); *ntiles = t;
    return 0;
}

// Runs the deferred final pass of the dense path.  DF_RUN: the pending state becomes the handle's run (and is released);
// DF_COLS: `cols` names the output columns (capacity pending->dstride), *n_out = groups written, the pending entries STAY (a later
// finish() can still produce the partial state); DF_TABLE: the tables go to pending->table (rc 2: compensation terms out of the
// float range -- the caller uses another exchange).
int complete_pending(vnm_agg* h, hipStream_t s, int out = DF_RUN, const DFinalArgs* cols = nullptr, int64_t* n_out = nullptr) {
    DensePending* pd = h->pending;
    if (!pd) return 0;
    if (out == DF_RUN && h->have_run) VNM_TRY(merge_run_into_table(h, s));   // (a run of another path: it has to make room)
    DFinalArgs df = pd->df;
    PoolScope pool;
    uint64_t* rk = nullptr; uint64_t* ra = nullptr;
    if (out == DF_RUN) {
        rk = (uint64_t*)pool.take((size_t)pd->dstride * 8 * 2);
        ra = (uint64_t*)pool.take((size_t)pd->dstride * 8 * h->plan.n_words);
        if (!rk || !ra) return 1;
        df.dkey = rk; df.dacc = ra;
    } else if (out == DF_COLS) {
        df.n_out = cols->n_out;
        for (int c = 0; c < cols->n_out; c++) { df.out_kind[c] = cols->out_kind[c]; df.out_ptr[c] = cols->out_ptr[c]; }
        if (cols->has_side) { df.has_side = 1; df.side = cols->side; df.side_bloom = cols->side_bloom; df.side_bloom_mask = cols->side_bloom_mask; df.side_found = cols->side_found; }
    } else {
        if (!pd->table) pd->table = (DTabSlot*)pool_alloc(sizeof(DTabSlot) << df.map.bits);
        if (!pd->table) return 1;
        df.table = pd->table;
    }
    df.dstride = (out == DF_COLS && cols->dstride > 0) ? cols->dstride : pd->dstride;   // (capacity of the output: + the side table's groups)
    pool_free(pd->dsets);
    pd->dsets = (DSet*)pool_alloc(pd->sets.size() * sizeof(DSet));
    if (!pd->dsets) return 1;
    VNM_HIP(hipMemcpyAsync(pd->dsets, pd->sets.data(), pd->sets.size() * sizeof(DSet), hipMemcpyHostToDevice, s));
    df.sets = pd->dsets; df.nsets = (int)pd->sets.size();
    unsigned long long fl[5] = {0, 0, 0, 0, 0};
    uint64_t* psum = nullptr; float* plo = nullptr; uint32_t* pcnt = nullptr;
    if (pd->fsplits > 1 && df.has_side) return set_error("aggregate: side table with a split final pass (internal error)");
    if (pd->fsplits > 1) {   // few final partitions: each is shared by `fsplits` workgroups (partial tables + dpart_merge_kernel)
        const size_t cells = (size_t)pd->nfinal * pd->fsplits << pd->tb;
        psum = (uint64_t*)pool.take(cells * 8); plo = (float*)pool.take(cells * 4); pcnt = (uint32_t*)pool.take(cells * 4);
        if (!psum || !plo || !pcnt) return 1;
    }
    for (int attempt = 0; attempt < 2; attempt++) {
        VNM_HIP(hipMemsetAsync(df.flags, 0, 16, s));   // [0] failure, [1] dense count ([2]: the scatter passes' spill count, consumed)
        if (attempt == 0 && pd->fsplits > 1) {
            DFinalArgs ds = df;
            ds.splits = pd->fsplits; ds.part_sum = psum; ds.part_lo = plo; ds.part_cnt = pcnt;
            const int cus = device_info().num_cus;
            KernelTimer timer("agg_part_final", s);
            const int g3 = (int)std::min<int64_t>(pd->nfinal * pd->fsplits, (int64_t)cus * 8);
            if (pd->tb == 9) dpart_final_kernel<uint16_t, 9, true><<<g3, 512, 0, s>>>(ds);
            else if (pd->tb == 10) dpart_final_kernel<uint16_t, 10, true><<<g3, 512, 0, s>>>(ds);
            else if (pd->tb == 11) dpart_final_kernel<uint16_t, 11, true><<<g3, 512, 0, s>>>(ds);
            else if (pd->tb == 12) dpart_final_kernel<uint16_t, 12, true><<<g3, VNM_DF12_BLOCK, 0, s>>>(ds);
            else dpart_final_kernel<uint16_t, 13, true><<<g3, 1024, 0, s>>>(ds);
            if (out != DF_COLS) ds.n_out = 0;
            dpart_merge_kernel<<<(int)(pd->nfinal << (pd->tb - 9)), 512, 0, s>>>(ds, pd->tb);
            VNM_HIP(hipGetLastError());
        } else if (attempt == 1 && pd->tb == 13) {
            // 64-bit compensation terms only fit 2^12-slot tables: every partition in two halves (slot bit 12 = 0, then 1)
            df.sub_bits = 1;
            for (int sub = 0; sub < 2; sub++) { df.sub = sub; VNM_TRY(launch_dense_final(df, 12, out, true, s)); }
        } else {
            if (df.has_side) {
                VNM_HIP(hipMemsetAsync(df.side_found, 0, (size_t)df.side.cap + 2, s));
                VNM_HIP(hipMemsetAsync(df.flags + 4, 0, 8, s));
            }
            VNM_TRY(launch_dense_final(df, pd->tb, out, attempt == 1, s));
            if (df.has_side) {
                KernelTimer timer("agg_side_append", s);
                dside_append_kernel<<<(int)std::min<int64_t>(((int64_t)df.side.cap + 2 + 255) / 256, (int64_t)device_info().num_cus * 8), 256, 0, s>>>(df, 0);
                dside_append_kernel<<<1, 256, 0, s>>>(df, 1);
                VNM_HIP(hipGetLastError());
            }
        }
        VNM_HIP(hipMemcpyAsync(fl, df.flags, 40, hipMemcpyDeviceToHost, s));
        VNM_HIP(hipStreamSynchronize(s));
        if (!fl[0]) break;
        if (out == DF_TABLE) return 2;
        if (attempt == 1) return set_error("aggregate: dense final pass failed (internal error)");
    }
    if (getenv("VNM_AGG_TRACE")) fprintf(stderr, "[agg] dense final (deferred, %zu batch%s): mode %d -> groups %llu\n", pd->sets.size(),
                                         pd->sets.size() == 1 ? "" : "es", out, fl[1]);
    if (n_out) *n_out = (int64_t)fl[1];
    if (out == DF_COLS && cols->null_pos) *cols->null_pos = df.has_side ? (int64_t)fl[4] : 0;
    if (out == DF_RUN) {
        pool.keep(rk); pool.keep(ra);
        h->run_key = rk; h->run_acc = ra; h->run_stride = pd->dstride; h->run_n = (int64_t)fl[1];
        h->run_dir = nullptr; h->run_nfin = 0;
        h->have_run = true;
        delete pd;
        h->pending = nullptr;
    }
    return 0;
}


// returns 0 = done (run stored), 2 = not applicable / failed (caller continues with the hash-partitioned path), 1 = error
// (nspill_out / n_nspill_out: keys of NULL-value rows that found no place -- nullable value column, generic programs only)
int dense_partitioned_aggregate(vnm_agg* h, const AggArgs& a, int64_t nrows, hipStream_t s, ulonglong2** spill_out, int64_t* n_spill_out,
                                bool generic = false, uint64_t** nspill_out = nullptr, int64_t* n_nspill_out = nullptr, bool vn_fold = false) {
    const int cus = device_info().num_cus;
    const DenseMap& mp = h->dmap;
    DGenArgs g{};
    if (generic && !dgen_program(h, a, &g)) return 2;
    const bool has_val = generic ? g.has_val != 0 : true;
    // nullable value column: NULL flags travel with the entries (generic programs) -- or, vn_fold, the hot program filtered by that
    // column itself: pass 1 drops the NULL rows with the filter and nothing after it ever sees a flag
    if ((h->segs_active || h->kn_valid) && generic) return 2;   // (stream segments, nullable keys: the hot program's ring scatter only)
    if (h->kn_valid && h->segs_active) return 2;
    const bool vn = has_val && a.cols[0].validity != nullptr && (generic || vn_fold);
    if (vn && ((!vn_fold && !nspill_out) || a.has_expr)) return 2;
    if (vn_fold && (generic || !a.hot_pred_is_v)) return 2;
    if (nspill_out) { *nspill_out = nullptr; *n_nspill_out = 0; }
    // slots per final partition: the largest table that still leaves >= 2048 final partitions (8 per CU)
    int tb = (int)env_i64("VNM_DENSE_TBITS", 12);
    tb = std::max(DP_TBITS_MIN, std::min(DP_TBITS_MAX, tb));
    while (tb > DP_TBITS_MIN && mp.bits - tb < 11) tb--;
    while (tb < DP_TBITS_MAX && mp.bits - tb > 18) tb++;
    // ranges of up to 2^22 codes: ONE scatter level (at most 512 partitions) with the largest table that allows it
    if (mp.bits <= DP_TBITS_MAX + 9 && env_i64("VNM_DENSE_ONE_LEVEL", 1)) tb = std::max(DP_TBITS_MIN, std::min(DP_TBITS_MAX, mp.bits - (int)env_i64("VNM_DENSE_ONE_P", 8)));
    if (mp.bits < 20) tb = (int)env_i64("VNM_DENSE_SMALL_TBITS", 11);   // split final pass: few partitions, long write runs in pass 1 (r03: 2^11-slot tables, G = 2e4 / 5e4 / 1e5 / 3e5: 7.4 / 6.5 / 6.3 / 6.1 -> 5.8 / 5.6 / 5.5 / 5.8 ms with the ring scatter; 2^12 was the r02 optimum)
    // ranges of 2^14 / 2^15 codes (G ~ 1e4 .. 3e4): 32 partitions of 2^9 / 2^10 slots, so that the ring scatter applies (round 4; before:
    // 8 / 16 partitions through the tile-sorting scatter, pass 1 at 5.7 ms -- the G = 1e4 cliff of the sweep, 7.0 ms between 2.9 at
    // G = 1e3 and 5.2 at G = 1e5)
    if (mp.bits - tb < 5 && env_i64("VNM_DENSE_MIN_PARTS32", 1)) tb = std::max(9, mp.bits - 5);
    if (generic) {   // the table must fit 64 KB of LDS (80 KB at most: one workgroup per CU less)
        int tmax = DP_TBITS_MAX;
        while (tmax > 9 && ((size_t)dgen_slot_bytes(g) << tmax) > 64 * 1024) tmax--;   // (the generic kernels take the table size at run time)
        if (((size_t)dgen_slot_bytes(g) << tmax) > 80 * 1024) return 2;
        if (tb > tmax) tb = tmax;
        if (mp.bits - tb > 18) return 2;
    }
    const int pbits = mp.bits - tb;
    const int levels = pbits > 9 ? 2 : 1;
    // pass 1 moves 12-byte entries out of 16-byte rows, pass 2 moves 10-byte entries out of 12: the SMALLER fan-out goes
    // to pass 1, whose write runs are the shorter ones (measured at b = 27: p1 = 7 / 8 -> 4.9 / 5.6 ms for pass 1)
    int p1 = levels == 2 ? (int)env_i64("VNM_DENSE_P1", pbits / 2) : pbits;
    if (levels == 2) { if (p1 > 9) p1 = 9; if (pbits - p1 > 9) p1 = pbits - 9; }
    const int p2 = pbits - p1;
    const int np1 = 1 << p1, np2 = levels == 2 ? 1 << p2 : 0;
    const int64_t tile1 = PT_TILE;
    const int grid1 = (int)std::min<int64_t>((int64_t)cus * env_i64("VNM_DENSE_GRID1_PER_CU", 2), (nrows + tile1 - 1) / tile1);
    int split2 = std::max(2, (grid1 + PT_MAX_REGIONS - 1) / PT_MAX_REGIONS);
    split2 = std::max(split2, std::min(grid1, (cus * 2 + np1 - 1) / np1));
    // skewed keys (the estimator's sample saw heavy keys): more, smaller work items for pass 2 -- the partition that holds a heavy key
    // has a multiple of the others' entries, and with cus * 2 work items for cus * 2 resident workgroups the heaviest sets the time
    if (h->heavy_share > 0.0) split2 = std::min(grid1, split2 * (int)env_i64("VNM_DENSE_SKEW_SPLIT", 8));
    const int64_t tiles_per_wg = ((nrows + tile1 - 1) / tile1 + grid1 - 1) / grid1;
    const int64_t rows_per_wg = tiles_per_wg * tile1;
    int64_t cap1v = ((rows_per_wg / np1 + rows_per_wg / np1 / 5 + 512) + 15) & ~15LL;
    // region stride = an ODD number of 128-byte lines: the lines the resident workgroups keep open in one partition then spread
    // over the L2 sets instead of sharing their low index bits (two sessions of 5-6 process pairs: 11.73 -> 11.25 and 11.27 -> 11.17 ms)
    if (env_i64("VNM_DENSE_ODD_CAP", 1) && ((cap1v / 16) & 1) == 0) cap1v += 16;
    const int64_t cap1 = cap1v;
    const bool c16_1 = levels == 1;  // pass-1 remainders fit 16 bits when they are the final slots
    // final partitions x splits >= ~4 workgroups per CU
    int fsplits = 1;
    if (levels == 1 && ((int64_t)1 << pbits) < (int64_t)cus * 2) {
        fsplits = (int)std::min<int64_t>(grid1, ((int64_t)cus * env_i64("VNM_DENSE_SPLIT_WGS", 4) + ((int64_t)1 << pbits) - 1) >> pbits);
        if (fsplits < 2) fsplits = 1;
    }
    unsigned long long* flags = (unsigned long long*)pool_alloc(128);   // [0..3] status words, [8..12] the NULL-key rows of this attempt (nullable key)
    double* v1 = (double*)pool_alloc(has_val ? (size_t)np1 * grid1 * cap1 * 8 : 8);
    void* c1 = pool_alloc((size_t)np1 * grid1 * cap1 * (c16_1 ? 2 : 4));
    uint32_t* n1 = (uint32_t*)pool_alloc((size_t)np1 * grid1 * 4);
    const int64_t spill_cap = nrows / 2 + (1 << 20);
    ulonglong2* spill = (ulonglong2*)pool_alloc((size_t)spill_cap * 16);
    const bool vn_lists = vn && !vn_fold;
    const int64_t nspill_cap = vn_lists ? nrows / 4 + (1 << 20) : 0;
    uint64_t* nspill = vn_lists ? (uint64_t*)pool_alloc((size_t)nspill_cap * 8) : nullptr;
    PoolSlotGuard<uint64_t> nspill_guard(&nspill);   // handed to the caller only on success (below)
    double* v2 = nullptr; void* c2 = nullptr; uint32_t* n2 = nullptr;
    uint64_t* rk = nullptr; uint64_t* ra = nullptr;
    auto release = [&]() { pool_free(flags); pool_free(v1); pool_free(c1); pool_free(n1); pool_free(v2); pool_free(c2); pool_free(n2); };
    if (!flags || !v1 || !c1 || !n1 || !spill || (vn_lists && !nspill)) { release(); pool_free(spill); return 1;}
    VNM_HIP(hipMemsetAsync(flags, 0, 128, s));
    fill_u64_kernel<<<1, 1, 0, s>>>((uint64_t*)flags + 11, F64_NEG_ZERO, 1);   // (the NULL-key rows' sum starts at -0.0: merge_init)
    VNM_HIP(hipGetLastError());
    DPartArgs d1{};
    d1.map = mp;
    d1.kp = (const uint64_t*)a.keys[0].values + a.keys[0].offset;
    d1.vp = has_val ? (const double*)a.cols[0].values + a.cols[0].offset : nullptr;
    d1.has_expr = a.has_expr; d1.expr = a.expr;
    d1.pp = h->pred_set ? (const double*)a.pred.values + a.pred.offset : nullptr;
    d1.has_pred = h->pred_set; d1.pred_is_v = a.hot_pred_is_v; d1.op = a.p.op; d1.thr = a.p.dval;
    d1.nrows = nrows;
    d1.out_vals = v1; d1.out_codes = c1; d1.out_counts = n1; d1.out_cap = cap1;
    d1.nparts = np1; d1.out_bits = mp.bits - p1;
    // producer-major regions (a workgroup's 128 output regions adjacent) were tried against TLB pressure: pass 1 unchanged,
    // pass 2 and the final pass 15-20 % slower (their reads become 150 KB chunks 75 MB apart) -> partition-major stays
    const int pmajor = (int)env_i64("VNM_DENSE_PRODUCER_MAJOR", 0);
    d1.producer_major = pmajor;
    // Non-temporal stores for pass 1's runs when a wide second level follows: pass 2 then reads them 15-20 % faster and the query
    // gains 0.4 ms (1.1 ms in sustained runs) at G = 1e8 (p2 = 8); with p2 = 6 (G = 1e7) or a single level it is neutral to
    // slightly worse, so it stays off there.  In pass 2 itself such stores cost 0.3 ms.  (VNM_DENSE_NT: bit 0 pass 1, bit 1 pass 2)
    d1.nt_store = (int)env_i64("VNM_DENSE_NT", levels == 2 && np2 >= 256 ? 1 : 0) & 1;
    d1.flags = flags; d1.spill = spill; d1.spill_cap = spill_cap;
    if (vn) { d1.vvalid = a.cols[0].validity; d1.voff = a.cols[0].offset; d1.nspill = nspill; d1.nspill_cap = nspill_cap; }
    // ring-buffer scatter (dring_scatter_kernel): whole 16-entry blocks only.  Ring capacity = what fits 128 KB of LDS, at
    // most 64 entries per partition; fan-outs that leave less than two blocks per ring keep the tile-sorting kernel.
    const int use_ring = (int)env_i64("VNM_DENSE_RING", 3);   // bit 0: pass 1, bit 1: pass 2
    const int ring_blk = 1024;
    auto ring_cap_for = [&](int np, size_t esize) -> int {
        int cap = (int)((size_t)(env_i64("VNM_DENSE_RING_LDS", ring_blk >= 1024 ? 128 : (ring_blk >= 512 ? 72 : 48)) * 1024) / ((size_t)np * esize) / DR_FB) * DR_FB;
        // at most 80 entries per ring -- more where few partitions share the LDS, so that an even spread of a four-pair sub-tile
        // (8192 entries) still fits one round: 32 partitions 272, 64 partitions 144
        cap = std::min(cap, (int)env_i64("VNM_DENSE_RING_CAP", std::max(80, ((8192 / np + DR_FB + DR_FB - 1) / DR_FB) * DR_FB)));
        // few partitions: long runs anyway (and 8192 entries per sub-tile on a handful of ring cursors: G = 1e4, four partitions,
        // pass 1 8.6 ms against 5.0 with the tile-sorting kernel)
        if (np < env_i64("VNM_DENSE_RING_MIN_NP", 32)) return 0;
        return cap >= 2 * DR_FB ? cap : 0;
    };
    const bool ring_limit = env_i64("VNM_DENSE_RING_LIMIT", 1) != 0;   // spill what two insert / flush rounds of a sub-tile leave pending (skew)
    int ring_pairs = (int)env_i64("VNM_DENSE_RING_PAIRS", 4);   // pass 1; pass 2 (every entry survives, more partitions): VNM_DENSE_RING_PAIRS2
#define VNM_DRING_B(FR_, CT_, HV_, BLK_, PR_, PV_, VN_, GRID_, ARGS_, CAP_)                                              \
    do {                                                                                                                \
        const size_t lds_ = (((size_t)(ARGS_).nparts * (CAP_) * ((HV_ ? 8 : 0) + sizeof(CT_))) + 15) & ~(size_t)15;       \
        /* the waiting batches of a stream: one segment each, sub-tiles of 2 * PR_ * BLK_ rows (the SEG instantiations) */ \
        constexpr bool SG_ = FR_ && HV_ && !VN_;                                                                        \
        const bool seg_ = SG_ && h->segs_active != nullptr;                                                             \
        const bool kn_ = SG_ && h->kn_valid != nullptr;   /* a nullable key: the KN instantiations (same shapes as SEG) */ \
        if (kn_) {                                                                                                      \
            if (2 * PR_ * BLK_ <= (ARGS_).nparts * ((CAP_) - DR_FB) && ring_limit) {                                    \
                VNM_HIP(hipFuncSetAttribute((const void*)dring_scatter_kernel<FR_, CT_, HV_, BLK_, PR_, PV_, VN_, 2, false, SG_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_)); \
                dring_scatter_kernel<FR_, CT_, HV_, BLK_, PR_, PV_, VN_, 2, false, SG_><<<GRID_, BLK_, lds_, s>>>(ARGS_, CAP_); \
            } else {                                                                                                    \
                VNM_HIP(hipFuncSetAttribute((const void*)dring_scatter_kernel<FR_, CT_, HV_, BLK_, PR_, PV_, VN_, 0, false, SG_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_)); \
                dring_scatter_kernel<FR_, CT_, HV_, BLK_, PR_, PV_, VN_, 0, false, SG_><<<GRID_, BLK_, lds_, s>>>(ARGS_, CAP_); \
            }                                                                                                           \
            break;                                                                                                      \
        }                                                                                                               \
        if (seg_ && upload_segs(h, (int64_t)2 * PR_ * BLK_, &(ARGS_).segs, &(ARGS_).nseg, &(ARGS_).nsub, seg_pool, s)) { release(); pool_free(spill); return 1; } \
        /* round limit (skew): two insert / flush rounds per sub-tile, where an even spread of a sub-tile's entries (every */ \
        /* row surviving) fits ONE */                                                                                   \
        if (2 * PR_ * BLK_ <= (ARGS_).nparts * ((CAP_) - DR_FB) && ring_limit) {                                        \
            if (seg_) {                                                                                                 \
                VNM_HIP(hipFuncSetAttribute((const void*)dring_scatter_kernel<FR_, CT_, HV_, BLK_, PR_, PV_, VN_, 2, SG_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_)); \
                dring_scatter_kernel<FR_, CT_, HV_, BLK_, PR_, PV_, VN_, 2, SG_><<<GRID_, BLK_, lds_, s>>>(ARGS_, CAP_); \
            } else {                                                                                                    \
                VNM_HIP(hipFuncSetAttribute((const void*)dring_scatter_kernel<FR_, CT_, HV_, BLK_, PR_, PV_, VN_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_)); \
                dring_scatter_kernel<FR_, CT_, HV_, BLK_, PR_, PV_, VN_, 2><<<GRID_, BLK_, lds_, s>>>(ARGS_, CAP_);      \
            }                                                                                                           \
        } else if (seg_) {                                                                                              \
            VNM_HIP(hipFuncSetAttribute((const void*)dring_scatter_kernel<FR_, CT_, HV_, BLK_, PR_, PV_, VN_, 0, SG_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_)); \
            dring_scatter_kernel<FR_, CT_, HV_, BLK_, PR_, PV_, VN_, 0, SG_><<<GRID_, BLK_, lds_, s>>>(ARGS_, CAP_);     \
        } else {                                                                                                        \
            VNM_HIP(hipFuncSetAttribute((const void*)dring_scatter_kernel<FR_, CT_, HV_, BLK_, PR_, PV_, VN_, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_)); \
            dring_scatter_kernel<FR_, CT_, HV_, BLK_, PR_, PV_, VN_, 0><<<GRID_, BLK_, lds_, s>>>(ARGS_, CAP_);          \
        }                                                                                                               \
    } while (0)
    // sub-tile = the largest of 4 / 2 / 1 pairs of rows per lane (at most ring_pairs) whose entries, evenly spread, fit one round
    // of the rings: the round limit then applies (64 partitions: 2 pairs, 32: 1 pair)
#define VNM_DRING_P(FR_, CT_, HV_, PV_, VN_, GRID_, ARGS_, CAP_)                                                         \
    do {                                                                                                                \
        int pr_ = ring_pairs >= 4 && !PV_ ? 4 : (ring_pairs >= 2 ? 2 : 1);                                              \
        while (ring_limit && pr_ > 1 && 2 * pr_ * 1024 > (ARGS_).nparts * ((CAP_) - DR_FB)) pr_ >>= 1;                  \
        if (pr_ == 4 && !PV_) VNM_DRING_B(FR_, CT_, HV_, 1024, 4, PV_, VN_, GRID_, ARGS_, CAP_);                         \
        else if (pr_ >= 2) VNM_DRING_B(FR_, CT_, HV_, 1024, 2, PV_, VN_, GRID_, ARGS_, CAP_);                            \
        else VNM_DRING_B(FR_, CT_, HV_, 1024, 1, PV_, VN_, GRID_, ARGS_, CAP_);                                          \
    } while (0)
    // (a predicate column of its own: three loads per pair of rows, at most two pairs per lane and sub-tile fit the registers)
#define VNM_DRING(FR_, CT_, HV_, GRID_, ARGS_, CAP_)                                                                     \
    do {                                                                                                                \
        if (FR_ && (ARGS_).has_pred && !(ARGS_).pred_is_v) VNM_DRING_P(FR_, CT_, HV_, FR_, false, GRID_, ARGS_, CAP_);    \
        else VNM_DRING_P(FR_, CT_, HV_, false, false, GRID_, ARGS_, CAP_);                                               \
    } while (0)
    // pass 1 over a nullable value column
#define VNM_DRING_VN(CT_, GRID_, ARGS_, CAP_)                                                                            \
    do {                                                                                                                \
        if ((ARGS_).has_pred && !(ARGS_).pred_is_v) VNM_DRING_P(true, CT_, true, true, true, GRID_, ARGS_, CAP_);         \
        else VNM_DRING_P(true, CT_, true, false, true, GRID_, ARGS_, CAP_);                                              \
    } while (0)
    const int rcap1 = (use_ring & 1) && !a.has_expr ? ring_cap_for(np1, (has_val ? 8 : 0) + (c16_1 ? 2 : 4)) : 0;
    PoolScope seg_pool;
    if ((h->segs_active || h->kn_valid) && (!rcap1 || vn)) { release(); pool_free(spill); return 2; }   // (only the ring scatter reads segments / key validity)
    // A nullable key: pass 1 sums the NULL-key rows of this ATTEMPT into scratch words; only an attempt that is known good adds them to
    // the NULL slot of the operator's HBM table (fold_null_rows).  A failed attempt -- full spill buffer, output too small, a
    // compensation term out of range: all known only after pass 1 -- leaves the table untouched, and the route that takes the batch
    // instead counts those rows itself (they used to be counted twice: ADVICE r04).
    auto fold_null_rows = [&]() -> int {
        if (!h->kn_valid) return 0;
        const GTable& gt = h->g;
        const uint64_t slot = gt.cap + 1;
        dnull_fold_kernel<<<1, 64, 0, s>>>(flags + 8, gt.tag + slot,
                                           a.hot_w_rows >= 0 ? gt.acc + (uint64_t)a.hot_w_rows * gt.stride + slot : nullptr,
                                           a.hot_w_valid >= 0 ? gt.acc + (uint64_t)a.hot_w_valid * gt.stride + slot : nullptr,
                                           a.hot_w_sum >= 0 ? gt.acc + (uint64_t)a.hot_w_sum * gt.stride + slot : nullptr,
                                           a.hot_comp && a.hot_w_sum >= 0 ? (int64_t)gt.stride : 0);
        VNM_HIP(hipGetLastError());
        return 0;
    };
    if (h->kn_valid) {
        if (ensure_table(h, 1024, s, true)) { release(); pool_free(spill); return 1; }
        uint64_t* scr = (uint64_t*)(flags + 8);
        d1.kvalid = h->kn_valid; d1.koff = h->kn_off;
        d1.nk_tag = scr;
        d1.nk_rows = a.hot_w_rows >= 0 ? scr + 1 : nullptr;
        d1.nk_valid = a.hot_w_valid >= 0 ? scr + 2 : nullptr;
        d1.nk_sum = a.hot_w_sum >= 0 ? scr + 3 : nullptr;
        d1.nk_lo_stride = a.hot_comp && a.hot_w_sum >= 0 ? 1 : 0;
    }
    {
        KernelTimer timer("agg_part_scatter1", s);
        if (rcap1 && vn) {
            if (c16_1) VNM_DRING_VN(uint16_t, grid1, d1, rcap1); else VNM_DRING_VN(uint32_t, grid1, d1, rcap1);
        } else if (vn) {
            if (c16_1) dpart_scatter_kernel<true, uint16_t, true, true><<<grid1, PT_BLOCK, 0, s>>>(d1);
            else dpart_scatter_kernel<true, uint32_t, true, true><<<grid1, PT_BLOCK, 0, s>>>(d1);
        } else if (rcap1) {
            if (has_val) { if (c16_1) VNM_DRING(true, uint16_t, true, grid1, d1, rcap1); else VNM_DRING(true, uint32_t, true, grid1, d1, rcap1); }
            else { if (c16_1) VNM_DRING(true, uint16_t, false, grid1, d1, rcap1); else VNM_DRING(true, uint32_t, false, grid1, d1, rcap1); }
        } else if (has_val) {
            if (c16_1) dpart_scatter_kernel<true, uint16_t><<<grid1, PT_BLOCK, 0, s>>>(d1);
            else dpart_scatter_kernel<true, uint32_t><<<grid1, PT_BLOCK, 0, s>>>(d1);
        } else {
            if (c16_1) dpart_scatter_kernel<true, uint16_t, false><<<grid1, PT_BLOCK, 0, s>>>(d1);
            else dpart_scatter_kernel<true, uint32_t, false><<<grid1, PT_BLOCK, 0, s>>>(d1);
        }
    }
    VNM_HIP(hipGetLastError());
    const double* fin_v = v1; const void* fin_c = c1; const uint32_t* fin_n = n1;
    int64_t fin_cap = cap1;
    int fin_regions = grid1;
    if (levels == 2) {
        const int64_t per_pg = (int64_t)grid1 * rows_per_wg / np1 / split2;
        int64_t cap2 = ((per_pg / np2 + per_pg / np2 / 4 + 256) + 15) & ~15LL;
        if (env_i64("VNM_DENSE_ODD_CAP", 1) && ((cap2 / 16) & 1) == 0) cap2 += 16;
        v2 = (double*)pool_alloc(has_val ? (size_t)np1 * np2 * split2 * cap2 * 8 : 8);
        c2 = pool_alloc((size_t)np1 * np2 * split2 * cap2 * 2);
        n2 = (uint32_t*)pool_alloc((size_t)np1 * np2 * split2 * 4);
        if (!v2 || !c2 || !n2) { release(); pool_free(spill); return 1; }
        DPartArgs d2{};
        d2.map = mp;
        d2.in_vals = v1; d2.in_codes = (const uint32_t*)c1; d2.in_counts = n1; d2.in_cap = cap1;
        d2.in_regions = grid1; d2.in_split = split2; d2.in_bits = mp.bits - p1;
        d2.in_pstride = pmajor ? 1 : grid1; d2.in_rstride = pmajor ? np1 : 1;
        d2.out_vals = v2; d2.out_codes = c2; d2.out_counts = n2; d2.out_cap = cap2;
        d2.nparts = np2; d2.out_bits = tb;
        d2.flags = flags; d2.spill = spill; d2.spill_cap = spill_cap;
        d2.nspill = nspill; d2.nspill_cap = nspill_cap;
        d2.nt_store = ((int)env_i64("VNM_DENSE_NT", 0) >> 1) & 1;
        const int rcap2 = (use_ring & 2) ? ring_cap_for(np2, (has_val ? 8 : 0) + 2) : 0;
        ring_pairs = (int)env_i64("VNM_DENSE_RING_PAIRS2", 2);
        {
            KernelTimer timer("agg_part_scatter2", s);
            if (rcap2) { if (has_val) VNM_DRING(false, uint16_t, true, np1 * split2, d2, rcap2); else VNM_DRING(false, uint16_t, false, np1 * split2, d2, rcap2); }
            else if (has_val) dpart_scatter_kernel<false, uint16_t><<<np1 * split2, PT_BLOCK, 0, s>>>(d2);
            else dpart_scatter_kernel<false, uint16_t, false><<<np1 * split2, PT_BLOCK, 0, s>>>(d2);
        }
        VNM_HIP(hipGetLastError());
        fin_v = v2; fin_c = c2; fin_n = n2; fin_cap = cap2; fin_regions = split2;
    }
#undef VNM_DRING_VN
#undef VNM_DRING
#undef VNM_DRING_P
#undef VNM_DRING_B
    const int64_t nfinal = (int64_t)1 << pbits;
    const int64_t dstride = std::min<int64_t>(h->dense_span, nrows) + 2;
    rk = (uint64_t*)pool_alloc((size_t)dstride * 8 * 2);
    ra = (uint64_t*)pool_alloc((size_t)dstride * 8 * h->plan.n_words);
    if (!rk || !ra) { release(); pool_free(spill); pool_free(rk); pool_free(ra); return 1; }
    DFinalArgs df{};
    df.map = mp;
    df.vals = fin_v; df.codes = fin_c; df.counts = fin_n; df.cap = fin_cap; df.regions = fin_regions; df.nfinal = nfinal;
    df.pstride = fin_regions; df.rstride = 1;
    if (levels == 1 && pmajor) { df.pstride = 1; df.rstride = np1; }
    df.w_rows = a.hot_w_rows; df.w_valid = a.hot_w_valid; df.w_sum = a.hot_w_sum;
    df.w_lo = a.hot_comp && a.hot_w_sum >= 0 ? a.hot_w_sum + 1 : -1;
    df.dkey = rk; df.dacc = ra; df.dstride = dstride; df.flags = flags;
    uint64_t* psum = nullptr; float* plo = nullptr; uint32_t* pcnt = nullptr;
    // the branches that write a run of their own: a pending (deferred) pass of earlier batches runs first and its run makes room
    const bool own_run = generic || !env_i64("VNM_DENSE_DEFER", 1);
    if (own_run && h->pending) {
        int rc = complete_pending(h, s);
        if (!rc) rc = merge_run_into_table(h, s);
        if (rc) { release(); pool_free(spill); pool_free(rk); pool_free(ra); return 1; }
    }
    if (generic) {
        const size_t slots = (size_t)1 << tb;
        const size_t lds = slots * dgen_slot_bytes(g);
        g.map = mp; g.tbits = tb;
        g.pstride = df.pstride; g.rstride = df.rstride;
        g.vals = fin_v; g.codes = fin_c; g.counts = fin_n; g.cap = fin_cap; g.regions = fin_regions; g.nfinal = nfinal;
        g.dkey = rk; g.dacc = ra; g.dstride = dstride; g.flags = flags;
        if (fsplits > 1) {
            const size_t work = (size_t)nfinal * fsplits;
            psum = (uint64_t*)pool_alloc(std::max<size_t>(8, work * g.n_lds * slots * 8));
            pcnt = (uint32_t*)pool_alloc(work * slots * 4);
            if (!psum || !pcnt) { release(); pool_free(spill); pool_free(rk); pool_free(ra); pool_free(psum); pool_free(pcnt); return 1; }
            g.splits = fsplits; g.part_w = psum; g.part_cnt = pcnt;
        }
        KernelTimer timer("agg_part_final", s);
#define VNM_DGFIN(HV_, SP_)                                                                                            \
    do {                                                                                                              \
        VNM_HIP(hipFuncSetAttribute((const void*)dgen_final_kernel<HV_, SP_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        int occ = 0;                                                                                                  \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)dgen_final_kernel<HV_, SP_>, 512, lds) != hipSuccess || occ < 1) occ = 1; \
        const int g3 = (int)std::min<int64_t>(nfinal * (SP_ ? fsplits : 1), (int64_t)cus * std::min(occ, 8));         \
        dgen_final_kernel<HV_, SP_><<<g3, 512, lds, s>>>(g);                                                          \
    } while (0)
        if (fsplits > 1) {
            if (has_val) VNM_DGFIN(true, true); else VNM_DGFIN(false, true);
            dgen_merge_kernel<<<(int)(nfinal << (tb - 9)), 512, 0, s>>>(g);
        } else {
            if (has_val) VNM_DGFIN(true, false); else VNM_DGFIN(false, false);
        }
#undef VNM_DGFIN
    } else if (fsplits > 1 && own_run) {
        const size_t cells = (size_t)nfinal * fsplits << tb;
        psum = (uint64_t*)pool_alloc(cells * 8); plo = (float*)pool_alloc(cells * 4); pcnt = (uint32_t*)pool_alloc(cells * 4);
        if (!psum || !plo || !pcnt) { release(); pool_free(spill); pool_free(rk); pool_free(ra); pool_free(psum); pool_free(plo); pool_free(pcnt); return 1; }
        df.splits = fsplits; df.part_sum = psum; df.part_lo = plo; df.part_cnt = pcnt;
        KernelTimer timer("agg_part_final", s);
#define VNM_DFINS(TB_)                                                                                                 \
    do {                                                                                                              \
        const int blk = TB_ >= 13 ? 1024 : (TB_ == 12 ? VNM_DF12_BLOCK : 512);                                       \
        int occ = 0;                                                                                                  \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)dpart_final_kernel<uint16_t, TB_, true>, blk, 0) != hipSuccess || occ < 1) occ = 1; \
        const int g3 = (int)std::min<int64_t>(nfinal * fsplits, (int64_t)cus * std::min(occ, 8));                     \
        dpart_final_kernel<uint16_t, TB_, true><<<g3, blk, 0, s>>>(df);                                               \
    } while (0)
        if (tb == 9) VNM_DFINS(9); else if (tb == 10) VNM_DFINS(10); else if (tb == 11) VNM_DFINS(11); else if (tb == 12) VNM_DFINS(12); else VNM_DFINS(13);
#undef VNM_DFINS
        dpart_merge_kernel<<<(int)(nfinal << (tb - 9)), 512, 0, s>>>(df, tb);
    } else if (env_i64("VNM_DENSE_DEFER", 1)) {
        // The final pass is DEFERRED: what it should write depends on what comes next (complete_pending), and the batches of a
        // stream share ONE final pass.  The scatter passes have to be known good first.
        unsigned long long fl0[3];
        VNM_HIP(hipMemcpyAsync(fl0, flags, 24, hipMemcpyDeviceToHost, s));
        VNM_HIP(hipStreamSynchronize(s));
        route_note(h->kn_valid ? "dense:nullable_key" : (h->segs_active ? "dense:stream_segments" : (vn ? "dense:nullable_value" : (fsplits > 1 ? "dense:split_final" :
                   (levels == 2 ? "dense:two_levels" : (p1 == 5 && tb <= 10 ? "dense:32_partitions" : "dense:one_level"))))),
                   "2^%d codes, 2^%d-slot final tables, p1 %d p2 %d, final pass deferred%s: %s", mp.bits, tb, p1, p2, vn ? ", nullable value" : "", fl0[0] ? "scatter FAILED (the batch goes another way)" : "ok");
        if (getenv("VNM_AGG_TRACE"))
            fprintf(stderr, "[agg] dense: bits %d tb %d levels %d p1 %d p2 %d -> scatter fail %llu spilled %llu, final pass deferred (bound %lld)\n",
                    mp.bits, tb, levels, p1, p2, fl0[0], fl0[2], (long long)dstride);
        if (fl0[0]) { release(); pool_free(rk); pool_free(ra); pool_free(spill); return 2; }
        if (fold_null_rows()) { release(); pool_free(rk); pool_free(ra); pool_free(spill); return 1; }
        pool_free(rk); pool_free(ra);   // (the run is allocated when the pass runs)
        if (fl0[2]) { *spill_out = spill; *n_spill_out = (int64_t)fl0[2]; }
        else { pool_free(spill); *spill_out = nullptr; *n_spill_out = 0; }
        if ((int64_t)fl0[2] > nrows / 16) h->dense_state = -1;
        DensePending* pd = h->pending;
        // batches join a pending pass of the same geometry; anything else (or a very long stream) runs it first.  A pass holds the
        // entries of its batches (10-12 bytes per row) until it runs: at most 1.5 * 2^30 rows of them, so that a stream of any length
        // keeps a bounded amount of HBM (and far below the 2^32 rows the 32-bit row counts of the final pass's slots could take)
        const int64_t pending_max_rows = std::min<int64_t>(env_i64("VNM_DENSE_PENDING_MAX_ROWS", 3LL << 29), (1LL << 32) - 1);
        if (pd && (pd->tb != tb || pd->levels != levels || pd->p1 != p1 || pd->fsplits != fsplits || pd->sets.size() >= DP_MAX_SETS ||
                   pd->rows + nrows > pending_max_rows || memcmp(&pd->df.map, &mp, sizeof(DenseMap)) != 0)) {
            const int rc = complete_pending(h, s);
            if (rc) { release(); return rc; }
            pd = nullptr;
        }
        if (!pd) {
            pd = new DensePending();
            df.dkey = nullptr; df.dacc = nullptr;
            pd->df = df; pd->tb = tb; pd->levels = levels; pd->p1 = p1; pd->fsplits = fsplits; pd->nfinal = nfinal; pd->dstride = 0;
            pd->blocks.push_back(flags);
            h->pending = pd;
        } else pool_free(flags);
        pd->dstride = std::min<int64_t>(h->dense_span, pd->dstride + nrows) + 2;
        pd->rows += nrows;
        DSet st{};
        st.vals = fin_v; st.codes = fin_c; st.counts = fin_n; st.cap = fin_cap; st.pstride = df.pstride; st.rstride = df.rstride; st.regions = fin_regions;
        st.pad = fin_cap < 2048 ? 1 : 0;   // small regions: one wave per region in the final pass
        pd->sets.push_back(st);
        if (levels == 2) {   // the first level's regions have been consumed by the second scatter pass
            pool_free(v1); pool_free(c1); pool_free(n1);
            pd->blocks.push_back(v2); pd->blocks.push_back(c2); pd->blocks.push_back(n2);
        } else { pd->blocks.push_back(v1); pd->blocks.push_back(c1); pd->blocks.push_back(n1); }
        return 0;
    } else {
        VNM_TRY(launch_dense_final(df, tb, DF_RUN, false, s));
    }
    VNM_HIP(hipGetLastError());
    unsigned long long fl[4];
    VNM_HIP(hipMemcpyAsync(fl, flags, 32, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    if (!fl[0] && fold_null_rows()) { release(); pool_free(psum); pool_free(plo); pool_free(pcnt); pool_free(rk); pool_free(ra); pool_free(spill); return 1; }
    if (!fl[0]) VNM_HIP(hipStreamSynchronize(s));   // (the fold reads the flags block release() gives back)
    release();
    pool_free(psum); pool_free(plo); pool_free(pcnt);
    route_note(generic ? (fsplits > 1 ? "dense:generic_split_final" : "dense:generic") : (vn ? "dense:nullable_value" : (fsplits > 1 ? "dense:split_final" : "dense:run")),
               "2^%d codes, 2^%d-slot final tables, levels %d, p1 %d p2 %d, %d final splits: %s", mp.bits, tb, levels, p1, p2, fsplits, fl[0] ? "FAILED (the batch goes another way)" : "ok");
    if (getenv("VNM_AGG_TRACE"))
        fprintf(stderr, "[agg] dense%s%s: bits %d tb %d levels %d p1 %d p2 %d splits %d -> fail %llu groups %llu spilled %llu + %llu NULL-value rows (dstride %lld)\n",
                generic ? " generic" : "", vn ? " nullable" : "", mp.bits, tb, levels, p1, p2, fsplits, fl[0], fl[1], fl[2], fl[3], (long long)dstride);
    if (fl[0]) {  // spill buffer full, dense output too small, or a compensation term beyond float range
        pool_free(rk); pool_free(ra); pool_free(spill);
        return 2;
    }
    if (fl[2]) { *spill_out = spill; *n_spill_out = (int64_t)fl[2]; }
    else { pool_free(spill); *spill_out = nullptr; *n_spill_out = 0; }
    if (vn && fl[3]) { *nspill_out = nspill; *n_nspill_out = (int64_t)fl[3]; nspill = nullptr; }
    if ((int64_t)(fl[2] + fl[3]) > nrows / 16) h->dense_state = -1;  // the sampled range does not describe the data: stop trying
    h->run_key = rk; h->run_acc = ra; h->run_stride = dstride; h->run_n = (int64_t)fl[1];
    h->run_dir = nullptr; h->run_nfin = 0;
    h->have_run = true;
    return 0;
}


// The dense-key path for TWO plain float64 input columns (round 4): `SELECT k, sum(a), sum(b) [, avg, count ...]` used
// to take the hash partitions' wide entries (24-byte entries through two tile-sorting levels and LDS hash tables: 24 ms per 5e8 rows
// at G = 1e8).  Here the entry is (value 1, value 2, code remainder) -- 20 bytes after pass 1, 18 after pass 2 -- through the ring
// scatter (dring_scatter_kernel<..., V2>: a second ring array, 32-48 entries per ring in 148 KB of LDS) and a direct-addressed
// final pass with two compensated sums per slot (dpart_final2_kernel).  Ranges of 2^21 .. 2^27 codes; no spill buffer: an entry
// without a place (a key outside the sampled range, a full region) fails the pass and the batch takes the hash partitions.
// returns 0 = done (run stored), 2 = not applicable / failed, 1 = error
int dense_two_aggregate(vnm_agg* h, const AggArgs& a, int64_t nrows, hipStream_t s) {
    const int cus = device_info().num_cus;
    const DenseMap& mp = h->dmap;
    int tb = 11;
    if (mp.bits - tb > 15) tb = 12;
    const int pbits = mp.bits - tb;
    if (pbits > 15 || pbits < 10 || a.has_expr) return 2;
    const int p2 = std::min(8, (pbits + 1) / 2), p1 = pbits - p2;      // (pass 1 moves the 20-byte entries: at most 128 rings of them fit)
    if (p1 > 7 || p1 < 5) return 2;
    const int np1 = 1 << p1, np2 = 1 << p2;
    const int64_t tile1 = PT_TILE;
    const int grid1 = (int)std::min<int64_t>((int64_t)cus * 2, (nrows + tile1 - 1) / tile1);
    int split2 = std::max(2, (grid1 + PT_MAX_REGIONS - 1) / PT_MAX_REGIONS);
    split2 = std::max(split2, std::min(grid1, (cus * 2 + np1 - 1) / np1));
    const int64_t tiles_per_wg = ((nrows + tile1 - 1) / tile1 + grid1 - 1) / grid1;
    const int64_t rows_per_wg = tiles_per_wg * tile1;
    int64_t cap1 = ((rows_per_wg / np1 + rows_per_wg / np1 / 5 + 512) + 15) & ~15LL;
    if (((cap1 / 16) & 1) == 0) cap1 += 16;
    const int64_t per_pg = (int64_t)grid1 * rows_per_wg / np1 / split2;
    int64_t cap2 = ((per_pg / np2 + per_pg / np2 / 4 + 256) + 15) & ~15LL;
    if (((cap2 / 16) & 1) == 0) cap2 += 16;
    const int64_t nfinal = (int64_t)1 << pbits;
    const int64_t dstride = std::min<int64_t>(h->dense_span, nrows) + 2;
    PoolScope pool;
    unsigned long long* flags = (unsigned long long*)pool.take(64);
    double* v1 = (double*)pool.take((size_t)np1 * grid1 * cap1 * 8);
    double* w1 = (double*)pool.take((size_t)np1 * grid1 * cap1 * 8);
    uint32_t* c1 = (uint32_t*)pool.take((size_t)np1 * grid1 * cap1 * 4);
    uint32_t* n1 = (uint32_t*)pool.take((size_t)np1 * grid1 * 4);
    double* v2 = (double*)pool.take((size_t)np1 * np2 * split2 * cap2 * 8);
    double* w2 = (double*)pool.take((size_t)np1 * np2 * split2 * cap2 * 8);
    uint16_t* c2 = (uint16_t*)pool.take((size_t)np1 * np2 * split2 * cap2 * 2);
    uint32_t* n2 = (uint32_t*)pool.take((size_t)np1 * np2 * split2 * 4);
    uint64_t* rk = (uint64_t*)pool.take((size_t)dstride * 8 * 2);
    uint64_t* ra = (uint64_t*)pool.take((size_t)dstride * 8 * h->plan.n_words);
    if (!flags || !v1 || !w1 || !c1 || !n1 || !v2 || !w2 || !c2 || !n2 || !rk || !ra) return 1;
    VNM_HIP(hipMemsetAsync(flags, 0, 64, s));
    const int lds_budget = (int)env_i64("VNM_DENSE_RING_LDS2", 148) * 1024;
    auto ring_cap = [&](int np, int esize) { return std::min(80, lds_budget / (np * esize) / DR_FB * DR_FB); };
    const int rcap1 = ring_cap(np1, 20), rcap2 = ring_cap(np2, 18);
    if (rcap1 < 2 * DR_FB || rcap2 < 2 * DR_FB) return 2;
    DPartArgs d1{};
    d1.map = mp;
    d1.kp = (const uint64_t*)a.keys[0].values + a.keys[0].offset;
    d1.vp = (const double*)a.cols[0].values + a.cols[0].offset;
    d1.vp2 = (const double*)a.cols[1].values + a.cols[1].offset;
    d1.pp = h->pred_set ? (const double*)a.pred.values + a.pred.offset : nullptr;
    d1.has_pred = h->pred_set; d1.pred_is_v = a.hot_pred_is_v; d1.op = a.p.op; d1.thr = a.p.dval;
    d1.nrows = nrows;
    d1.out_vals = v1; d1.out_vals2 = w1; d1.out_codes = c1; d1.out_counts = n1; d1.out_cap = cap1;
    d1.nparts = np1; d1.out_bits = mp.bits - p1;
    d1.nt_store = 1;
    d1.flags = flags; d1.spill = nullptr; d1.spill_cap = 0;
#define VNM_DRING2(FR_, CT_, PR_, PV_, GRID_, ARGS_, CAP_)                                                                       \
    do {                                                                                                                        \
        const size_t lds_ = (((size_t)(ARGS_).nparts * (CAP_) * (16 + sizeof(CT_))) + 15) & ~(size_t)15;                         \
        VNM_HIP(hipFuncSetAttribute((const void*)dring_scatter_kernel<FR_, CT_, true, 1024, PR_, PV_, false, 0, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_)); \
        dring_scatter_kernel<FR_, CT_, true, 1024, PR_, PV_, false, 0, false, false, true><<<GRID_, 1024, lds_, s>>>(ARGS_, CAP_); \
    } while (0)
    {
        KernelTimer timer("agg_part_scatter1", s);
        const bool pv = h->pred_set && !a.hot_pred_is_v;
        const bool two_pairs = 2 * 2 * 1024 <= np1 * (rcap1 - DR_FB);
        if (pv) { if (two_pairs) VNM_DRING2(true, uint32_t, 2, true, grid1, d1, rcap1); else VNM_DRING2(true, uint32_t, 1, true, grid1, d1, rcap1); }
        else { if (two_pairs) VNM_DRING2(true, uint32_t, 2, false, grid1, d1, rcap1); else VNM_DRING2(true, uint32_t, 1, false, grid1, d1, rcap1); }
    }
    VNM_HIP(hipGetLastError());
    DPartArgs d2{};
    d2.map = mp;
    d2.in_vals = v1; d2.in_vals2 = w1; d2.in_codes = c1; d2.in_counts = n1; d2.in_cap = cap1;
    d2.in_regions = grid1; d2.in_split = split2; d2.in_bits = mp.bits - p1;
    d2.in_pstride = grid1; d2.in_rstride = 1;
    d2.out_vals = v2; d2.out_vals2 = w2; d2.out_codes = c2; d2.out_counts = n2; d2.out_cap = cap2;
    d2.nparts = np2; d2.out_bits = tb;
    d2.flags = flags; d2.spill = nullptr; d2.spill_cap = 0;
    {
        KernelTimer timer("agg_part_scatter2", s);
        if (2 * 2 * 1024 <= np2 * (rcap2 - DR_FB)) VNM_DRING2(false, uint16_t, 2, false, np1 * split2, d2, rcap2);
        else VNM_DRING2(false, uint16_t, 1, false, np1 * split2, d2, rcap2);
    }
#undef VNM_DRING2
    VNM_HIP(hipGetLastError());
    DFinalArgs df{};
    df.map = mp;
    df.vals = v2; df.vals2 = w2; df.codes = c2; df.counts = n2; df.cap = cap2; df.regions = split2; df.nfinal = nfinal;
    df.pstride = split2; df.rstride = 1;
    df.w_rows = a.hot_w[A_COUNT_ROWS];
    df.w_valid = a.hot_w[A_COUNT_VALID]; df.w_sum = a.hot_w[A_SUM_F64]; df.w_lo = a.hot_comp && df.w_sum >= 0 ? df.w_sum + 1 : -1;
    df.w_valid2 = a.hot_w2[A_COUNT_VALID]; df.w_sum2 = a.hot_w2[A_SUM_F64]; df.w_lo2 = a.hot_comp && df.w_sum2 >= 0 ? df.w_sum2 + 1 : -1;
    df.dkey = rk; df.dacc = ra; df.dstride = dstride; df.flags = flags;
    {
        KernelTimer timer("agg_part_final", s);
        if (tb == 11) dpart_final2_kernel<uint16_t, 11><<<(int)std::min<int64_t>(nfinal, (int64_t)cus * 2), 512, 0, s>>>(df);
        else dpart_final2_kernel<uint16_t, 12><<<(int)std::min<int64_t>(nfinal, (int64_t)cus), 1024, 0, s>>>(df);
    }
    VNM_HIP(hipGetLastError());
    unsigned long long fl[2] = {0, 0};
    VNM_HIP(hipMemcpyAsync(fl, flags, 16, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    if (getenv("VNM_AGG_TRACE"))
        fprintf(stderr, "[agg] dense, two columns: bits %d tb %d p1 %d p2 %d rings %d / %d -> fail %llu groups %llu\n", mp.bits, tb, p1, p2, rcap1, rcap2, fl[0], fl[1]);
    if (fl[0]) return 2;
    pool.keep(rk); pool.keep(ra);
    h->run_key = rk; h->run_acc = ra; h->run_stride = dstride; h->run_n = (int64_t)fl[1];
    h->run_dir = nullptr; h->run_nfin = 0;
    h->have_run = true;
    return 0;
}

}  // namespace

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

// ---- packed composite keys: host side ----------------------------------------------------------------------
extern "C" int vnm_agg_finish(vnm_agg* h, int64_t* n_groups, void* stream);
extern "C" int vnm_agg_merge_device(vnm_agg* h, int64_t n, uint64_t* const* key_words, uint64_t* const* acc_words, void* stream);
extern "C" int vnm_agg_next_device(vnm_agg* h, int64_t nrows, const vnm_dcol* keys, const vnm_dcol* inputs, const vnm_dcol* pred, void* stream);
extern "C" vnm_agg* vnm_agg_create(int kind, int n_keys, const int* key_types, int n_funcs, const int* funcs,
                                   const int* in_types, const int* in_flags, const int* in_col_ids);
extern "C" void vnm_agg_destroy(vnm_agg* h);
static int agg_finish_core(vnm_agg* h, int64_t* n_groups, void* stream);
static int flush_queue(vnm_agg* h, void* stream);

namespace {

// vnm_agg_create without the ordered MIN / MAX engine: the operators the library builds for itself (packed keys, program parts,
// the suffix / merge operators of vnm_agg_exact.inc)
vnm_agg* agg_create(int kind, int n_keys, const int* key_types, int n_funcs, const int* funcs, const int* in_types, const int* in_flags,
                    const int* in_col_ids);

// Decide the packing from the key ranges of the first batch: field j holds value codes [0, cap_j) + the NULL code.
// Spare bits are spread over the fields and the observed range is centred in its field, so later batches may
// drift in both directions.  Returns true and fills h->pack when the keys fit 63 bits.
void free_pack_tables(vnm_agg* h) {
    for (int j = 0; j < AGG_MAX_KEYS; j++) {
        if (h->pack.dtab[j]) pool_free(h->pack.dtab[j]);
        h->pack.dtab[j] = nullptr;
    }
}

bool plan_packing(vnm_agg* h, const vnm_dcol* keys, int64_t nrows, hipStream_t s, int* err) {
    *err = 0;
    const int n = h->plan.n_keys;
    PackParams& p = h->pack;
    p.n = n;
    for (int j = 0; j < n; j++) p.cols[j] = keys[j];
    unsigned long long* d = (unsigned long long*)pool_alloc(16 * AGG_MAX_KEYS);
    if (!d) { *err = 1; return false; }
    unsigned long long init[2 * AGG_MAX_KEYS], got[2 * AGG_MAX_KEYS];
    for (int j = 0; j < AGG_MAX_KEYS; j++) { init[2 * j] = ~0ULL; init[2 * j + 1] = 0; }
    bool ok = hipMemcpyAsync(d, init, sizeof(init), hipMemcpyHostToDevice, s) == hipSuccess;
    int grid = (int)std::min<int64_t>((nrows + 255) / 256, (int64_t)device_info().num_cus * 8);
    if (ok) key_range_kernel<<<grid, 256, 0, s>>>(p, nrows, d);
    ok = ok && hipMemcpyAsync(got, d, sizeof(got), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    pool_free(d);
    if (!ok) { *err = set_error("aggregate: key range kernel failed"); return false; }
    int need_bits[AGG_MAX_KEYS];
    uint64_t span[AGG_MAX_KEYS];
    int64_t mn[AGG_MAX_KEYS];
    int total = 0;
    for (int j = 0; j < n; j++) {
        const bool any = got[2 * j] <= got[2 * j + 1];
        mn[j] = any ? (int64_t)(got[2 * j] ^ 0x8000000000000000ULL) : 0;
        const int64_t mx = any ? (int64_t)(got[2 * j + 1] ^ 0x8000000000000000ULL) : 0;
        span[j] = (uint64_t)mx - (uint64_t)mn[j];
        int b = 1;
        if (span[j] >= (1ULL << 61)) b = 64;   // only as a dictionary-coded field
        else {
            const uint64_t need = span[j] + 2;  // values + the NULL code
            while ((1ULL << b) < need) b++;
        }
        need_bits[j] = b;
        total += b;
    }
    // ranges too wide for one word: the widest columns become dictionary-coded fields (see PackParams) until the rest
    // fits.  Each table gets 2.5x the column's estimated distinct count (the estimator reads 8-byte columns;
    // others are taken as all-distinct up to 2^27), at least 2^12 slots -- and up to 2^20 where bits are left over, so
    // that a dictionary planned from a small first batch has room for what later batches bring.
    for (int j = 0; j < AGG_MAX_KEYS; j++) { p.dtab[j] = nullptr; p.dbits[j] = 0; }
    bool is_dict[AGG_MAX_KEYS] = {};
    int range_bits[AGG_MAX_KEYS];
    for (int j = 0; j < n; j++) range_bits[j] = need_bits[j];
    while (total > 63) {
        int w = -1;
        for (int j = 0; j < n; j++)
            if (!is_dict[j] && (w < 0 || need_bits[j] > need_bits[w])) w = j;
        if (w < 0 || getenv("VNM_AGG_NO_DICT") != nullptr) return false;
        int64_t est = std::min<int64_t>(nrows, 1 << 27);
        if (type_width(keys[w].type) == 8 && (keys[w].offset & 1) == 0) {   // (whatever NULL slots hold only adds to the estimate)
            if (estimate_groups(h, keys[w], nrows, &est, s)) { *err = 1; return false; }
        }
        h->widest_est = std::max<int64_t>(h->widest_est, est);   // (a lower bound of the tuple count: sizes the tuple dictionary)
        int tb = 12;
        while (tb < 30 && (1LL << tb) < est * 5 / 2) tb++;
        if (tb + 1 >= need_bits[w]) return false;   // no narrower than its range: nothing left to gain
        is_dict[w] = true;
        p.dbits[w] = tb;
        total += tb + 1 - need_bits[w];
        need_bits[w] = tb + 1;
        span[w] = 0;
        mn[w] = 0;
    }
    for (bool grew = true; grew && total < 63;) {
        grew = false;
        for (int j = 0; j < n && total < 63; j++)
            if (is_dict[j] && p.dbits[j] < 20 && p.dbits[j] + 2 < range_bits[j]) { p.dbits[j]++; need_bits[j]++; total++; grew = true; }
    }
    for (int j = 0; j < n; j++) {
        if (!is_dict[j]) continue;
        const size_t bytes = ((size_t)(1ULL << p.dbits[j]) + 1) * 8;
        p.dtab[j] = (uint64_t*)pool_alloc(bytes);
        if (!p.dtab[j]) { *err = 1; free_pack_tables(h); return false; }
        if (hipMemsetAsync(p.dtab[j], 0xFF, bytes, s) != hipSuccess) {
            *err = set_error("aggregate: key dictionary memset failed");
            free_pack_tables(h);
            return false;
        }
    }
    // A SINGLE packed key (NULLs, a narrow type, an odd offset): ONE spare bit only -- later batches may drift by half the range on
    // either side, and the packed codes (NULL = the field's top code) stay a DENSE range for the inner operator (with all 62 bits the
    // values sat at 2^61 and the NULL code at 2^62 - 1: hash partitions, 21 ms per 5e8 rows at G = 1e8 instead of ~7)
    const int extra = n == 1 ? std::min(1, 63 - total) : (63 - total) / n;
    int shift = 0;
    for (int j = 0; j < n; j++) {
        int b = need_bits[j] + extra;
        if (b > 62) b = 62;
        p.bits[j] = b;
        p.shift[j] = shift;
        shift += b;
        const uint64_t cap = (1ULL << b) - 1;
        const uint64_t slack = cap - (span[j] + 1);
        p.lo[j] = p.dtab[j] ? 0 : (uint64_t)mn[j] - slack / 2;
    }
    return true;
}

// ---- tuple dictionary: host side (tuple_gid_kernel) ---------------------------------------------------------------------
int tdict_alloc(TDict* d, uint64_t cap, int kwt, hipStream_t s) {
    memset(d, 0, sizeof(*d));
    d->cap = cap; d->kwt = kwt; d->sw = kwt <= 6 ? 8 : 16;
    d->slot = (uint64_t*)pool_alloc((size_t)cap * d->sw * 8);
    d->ctl = (unsigned long long*)pool_alloc(64);
    if (!d->slot || !d->ctl || hipMemsetAsync(d->slot, 0xFF, (size_t)cap * d->sw * 8, s) != hipSuccess || hipMemsetAsync(d->ctl, 0, 64, s) != hipSuccess) {
        pool_free(d->slot); pool_free(d->ctl);   // (nothing half-allocated is left behind)
        memset(d, 0, sizeof(*d));
        return set_error("aggregate: tuple dictionary allocation failed");
    }
    return 0;
}

void tdict_free(TDict* d) {
    pool_free(d->slot);
    pool_free(d->ctl);
    memset(d, 0, sizeof(*d));
}

int tdict_grow(TDict* d, uint64_t new_cap, hipStream_t s) {
    TDict nd;
    VNM_TRY(tdict_alloc(&nd, new_cap, d->kwt, s));
    VNM_HIP(hipMemcpyAsync(nd.ctl + 2, d->ctl + 2, 8, hipMemcpyDeviceToDevice, s));   // the fill carries over
    tdict_rehash_kernel<<<(int)std::min<int64_t>(((int64_t)d->cap + 255) / 256, (int64_t)device_info().num_cus * 8), 256, 0, s>>>(*d, nd);
    VNM_HIP(hipGetLastError());
    VNM_HIP(hipStreamSynchronize(s));
    tdict_free(d);
    *d = nd;
    return 0;
}

int enter_tuple_mode(vnm_agg* h, int64_t nrows, hipStream_t s) {
    uint64_t want = h->hint > 0 ? (uint64_t)h->hint * 2 : std::max<uint64_t>((uint64_t)1 << 21, (uint64_t)h->widest_est * 2);
    if (h->hint <= 0 && (uint64_t)nrows * 2 < want) want = (uint64_t)(nrows > 512 ? nrows : 512) * 2;
    VNM_TRY(tdict_alloc(&h->tdict, pow2_at_least(want < 1024 ? 1024 : want), h->plan.n_keys + 1, s));
    const int kt = VNM_U64;
    h->inner = agg_create(VNM_SINGLE_NUMERICAL, 1, &kt, h->n_funcs, h->c_funcs, h->c_in_types, h->c_in_flags,
                              h->c_has_ids ? h->c_in_col_ids : nullptr);
    if (!h->inner) return 1;
    h->inner->hint = h->hint;
    h->tnext = (unsigned long long*)pool_alloc(64);
    if (!h->tnext) return 1;
    VNM_HIP(hipMemsetAsync(h->tnext, 0, 8, s));
    h->tuple_mode = true;
    return 0;
}

void leave_tuple_mode(vnm_agg* h) {
    if (h->tdict.slot) tdict_free(&h->tdict);
    pool_free(h->tnext);
    h->tnext = nullptr;
    h->tuple_mode = false;
}

// one batch: every row's tuple -> group id (find-or-insert in the dictionary), then the ids through the single-key operator
int tuple_next(vnm_agg* h, int64_t nrows, const vnm_dcol* keys, const vnm_dcol* inputs, const vnm_dcol* pred, hipStream_t s) {
    TDict* d = &h->tdict;
    PoolScope pool;
    TupArgs a{};
    a.out = (uint64_t*)pool.take((size_t)nrows * 8);
    if (!a.out) return 1;
    a.nk = h->plan.n_keys;
    for (int j = 0; j < a.nk; j++) a.keys[j] = keys[j];
    if (h->pred_set) {
        a.pred = *pred;
        a.p = make_predicate(pred->type, pred->validity != nullptr, h->pred_op, h->pred_is_float, h->pred_dval, h->pred_ival);
    }
    a.nrows = nrows;
    a.ntiles = (nrows + AGG_TILE - 1) / AGG_TILE;
    a.gnext = h->tnext;
    int grid = device_info().num_cus * (int)env_i64("VNM_TUPLE_GRID_PER_CU", 4)   /* (more workgroups = a larger margin of free slots = a larger table: 8 per CU 11.9 ms, 4 per CU 7.8 ms per 2e8 rows at G = 1e6) */;
    if (grid > a.ntiles) grid = (int)a.ntiles;
    a.margin = (int64_t)grid * AGG_TILE;
    a.progress = (unsigned int*)pool.take((size_t)grid * 4);
    if (!a.progress) return 1;
    VNM_HIP(hipMemsetAsync(a.progress, 0, (size_t)grid * 4, s));
    for (int round = 0;; round++) {
        while ((int64_t)(d->cap * 7 / 10) < a.margin + 1) VNM_TRY(tdict_grow(d, d->cap * 4, s));
        VNM_HIP(hipMemsetAsync(d->ctl, 0, 16, s));   // [1] the flag; [2] fill persists
        a.d = *d;
        a.fill_limit = (int64_t)(d->cap * 7 / 10);
        {
            KernelTimer timer("agg_tuple_ids", s);
            switch (d->kwt) {
#define VNM_TG(K) case K: tuple_gid_kernel<K><<<grid, 256, 0, s>>>(a); break;
                VNM_TG(3) VNM_TG(4) VNM_TG(5) VNM_TG(6) VNM_TG(7) VNM_TG(8) VNM_TG(9)
#undef VNM_TG
                default: return set_error("aggregate: tuple dictionary over %d key columns (internal error)", d->kwt - 1);
            }
        }
        VNM_HIP(hipGetLastError());
        unsigned long long ctl[4];
        VNM_HIP(hipMemcpyAsync(ctl, d->ctl, sizeof(ctl), hipMemcpyDeviceToHost, s));
        VNM_HIP(hipStreamSynchronize(s));
        if (!ctl[1]) break;   // no workgroup ran out of room
        VNM_TRY(tdict_grow(d, d->cap * (round >= 1 ? 16 : 4), s));   // the workgroups resume from progress[]
    }
    vnm_dcol gk{};
    gk.values = a.out; gk.type = VNM_U64; gk.length = nrows;
    int rc = vnm_agg_set_predicate(h->inner, h->pred_set ? 1 : 0, h->pred_op, h->pred_is_float, h->pred_dval, h->pred_ival);
    if (!rc) { h->inner->child = true; h->inner->cur_seq = h->cur_seq; rc = vnm_agg_next_device(h->inner, nrows, &gk, inputs, pred, (void*)s); }
    if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = set_error("aggregate: tuple-dictionary batch failed");
    if (!rc) h->rows_seen += nrows;
    return rc;
}

// the key words of the inner operator's n finished groups in this operator's wide layout: keys [kw][stride]
int inner_keys(vnm_agg* h, const vnm_agg* in, int64_t n, uint64_t* keys, int64_t stride, hipStream_t s) {
    const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)device_info().num_cus * 8);
    if (!h->tuple_mode) {
        key_unpack_kernel<<<grid, 256, 0, s>>>(h->pack, in->dkey, n, keys, stride);
        VNM_HIP(hipGetLastError());
        return 0;
    }
    unsigned long long ngid = 0;
    VNM_HIP(hipMemcpyAsync(&ngid, h->tnext, 8, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    const TDict& g = h->tdict;
    PoolScope pool;
    if (g.cap >= (1ULL << 32)) return set_error("aggregate: tuple dictionary beyond 2^32 slots");
    uint32_t* slot_of = (uint32_t*)pool.take((size_t)(ngid ? ngid : 1) * 4);
    if (!slot_of) return 1;
    KernelTimer timer("agg_tuple_keys", s);
    tuple_sweep_kernel<<<(int)std::min<int64_t>(((int64_t)g.cap + 255) / 256, (int64_t)device_info().num_cus * 16), 256, 0, s>>>(g, slot_of, (int64_t)ngid);
    tuple_keys_kernel<<<grid, 256, 0, s>>>(g, slot_of, (int64_t)ngid, in->dkey, n, keys, stride);
    VNM_HIP(hipGetLastError());
    VNM_HIP(hipStreamSynchronize(s));
    return 0;
}

// A later batch does not fit the packing (a key outside the packed ranges, a per-column dictionary that is full): the operator
// moves to the tuple dictionary -- the finished groups' tuples are inserted there and their partial state is re-keyed by id --
// instead of to the wide-key table and its per-row atomics for the rest of the stream.
int packed_to_tuple(vnm_agg* h, int64_t nrows, hipStream_t s) {
    vnm_agg* in = h->inner;
    int64_t n = 0;
    VNM_TRY(vnm_agg_finish(in, &n, (void*)s));
    h->inner = nullptr;   // (enter_tuple_mode makes the new one)
    const int kw = h->plan.kw;
    PoolScope pool;
    uint64_t* words = nullptr;
    int rc = 0;
    if (n > 0) {
        words = (uint64_t*)pool.take((size_t)kw * n * 8);
        if (!words) rc = 1;
        if (!rc) {
            KernelTimer timer("agg_demote", s);
            key_unpack_kernel<<<(int)std::min<int64_t>((n + 255) / 256, (int64_t)device_info().num_cus * 8), 256, 0, s>>>(h->pack, in->dkey, n, words, n);
        }
        if (!rc && (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) rc = set_error("aggregate: unpacking the groups failed");
    }
    free_pack_tables(h);   // the unpack above was their last reader
    h->widest_est = std::max<int64_t>(h->widest_est, n);
    if (!rc) rc = enter_tuple_mode(h, std::max<int64_t>(nrows, n), s);
    if (!rc && n > 0) {
        TDict* d = &h->tdict;
        while (!rc && (int64_t)(d->cap * 7 / 10) < n + 1) rc = tdict_grow(d, d->cap * 4, s);
        uint64_t* gid = (uint64_t*)pool.take((size_t)n * 16);   // ids, then the (all-zero) null-mask word of the single id key
        if (!rc && !gid) rc = 1;
        if (!rc && hipMemsetAsync(gid + n, 0, (size_t)n * 8, s) != hipSuccess) rc = set_error("aggregate: memset failed");
        if (!rc) {
            const int grid = (int)std::min<int64_t>((n + 2047) / 2048, (int64_t)device_info().num_cus * 8);
            const int64_t per_block = (n + grid - 1) / grid;
            KernelTimer timer("agg_tuple_ids", s);
            switch (d->kwt) {
#define VNM_TW(K) case K: tuple_words_kernel<K><<<grid, 256, 0, s>>>(*d, words, n, per_block, gid, h->tnext); break;
                VNM_TW(3) VNM_TW(4) VNM_TW(5) VNM_TW(6) VNM_TW(7) VNM_TW(8) VNM_TW(9)
#undef VNM_TW
                default: rc = set_error("aggregate: tuple dictionary over %d key columns (internal error)", d->kwt - 1);
            }
        }
        if (!rc) {
            uint64_t* kp[2] = {gid, gid + n};
            uint64_t* ap[AGG_MAX_WORDS];
            for (int w = 0; w < h->plan.n_words; w++) ap[w] = in->dacc + (size_t)w * in->dstride;
            rc = vnm_agg_merge_device(h->inner, n, kp, ap, (void*)s);
        }
        if (!rc && (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) rc = set_error("aggregate: moving the groups to the tuple dictionary failed");
    }
    vnm_agg_destroy(in);
    return rc;
}

// leave packed mode: the groups aggregated so far are unpacked and merged into h's own (wide-key) table
int demote_packed(vnm_agg* h, hipStream_t s) {
    vnm_agg* in = h->inner;
    h->inner = nullptr;
    int64_t n = 0;
    int rc = vnm_agg_finish(in, &n, (void*)s);
    if (!rc && n > 0) {
        const int kw = h->plan.kw;
        uint64_t* keys = (uint64_t*)pool_alloc((size_t)kw * n * 8);
        if (!keys) rc = 1;
        if (!rc) {
            int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)device_info().num_cus * 8);
            {
                KernelTimer timer("agg_demote", s);
                rc = inner_keys(h, in, n, keys, n, s);
            }
            uint64_t* kp[AGG_MAX_KEYS + 1];
            uint64_t* ap[AGG_MAX_WORDS];
            for (int j = 0; j < kw; j++) kp[j] = keys + (size_t)j * n;
            for (int w = 0; w < h->plan.n_words; w++) ap[w] = in->dacc + (size_t)w * in->dstride;
            if (!rc) rc = vnm_agg_merge_device(h, n, kp, ap, (void*)s);
            if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = set_error("aggregate: demotion failed");
        }
        pool_free(keys);
    }
    vnm_agg_destroy(in);
    if (hipStreamSynchronize(s) != hipSuccess && !rc) rc = set_error("aggregate: demotion failed");
    free_pack_tables(h);   // the unpack above was their last reader
    leave_tuple_mode(h);
    return rc;
}


// ---- program split --------------------------------------------------------------------------------------------------------
struct PartJoinArgs {
    int64_t n;
    const int64_t* perm;          // perm[i] = the part's group with the i-th smallest key
    const uint64_t* key_src;      // the part's key words
    const uint64_t* mask_src;     // ... and its null-mask words (1 = the NULL-key group, whose key word is 0)
    uint64_t* key_out;            // first part: writes the joined key column; the others compare with it (same rows -> same keys)
    uint64_t* mask_out;
    int check;
    unsigned long long* flag;
    int nw;
    const uint64_t* src[AGG_MAX_WORDS];
    uint64_t* dst[AGG_MAX_WORDS];
};
// one pass per part: the permutation is read once and the part's words of a group are gathered together (one kernel per
// word: 20 ms for 18 words x 2e7 groups, the random 8-byte reads one after the other)
__global__ __launch_bounds__(256) void part_join_kernel(PartJoinArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    // The parts are sorted by key word alone; the NULL-key group (key word 0, mask 1) and a real key 0 are then the first two
    // rows in an order each part decides for itself: the real key goes first, the NULL group second, in every part.
    bool swap01 = false;
    if (a.n >= 2 && a.perm) {
        const int64_t g0 = a.perm[0], g1 = a.perm[1];
        swap01 = a.key_src[g0] == 0 && a.key_src[g1] == 0 && a.mask_src[g0] != 0 && a.mask_src[g1] == 0;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        const int64_t g = a.perm ? a.perm[swap01 && i < 2 ? 1 - i : i] : i;   // (no permutation: the parts agree on the order)
        const uint64_t k = a.key_src[g], m = a.mask_src[g];
        if (a.check) bad = bad || a.key_out[i] != k || a.mask_out[i] != m;
        else { a.key_out[i] = k; a.mask_out[i] = m; }
        for (int w = 0; w < a.nw; w++) a.dst[w][i] = a.src[w][g];
    }
    if (bad) __hip_atomic_store(a.flag, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Parts that took the dense path over the SAME code map write their groups in runs of ascending code: the final passes place a
// partition's groups 64 slots (one wave's ballot) at a time, the partitions wherever their workgroup's claim landed.  So a UNIT of 64
// codes is one contiguous, equally ordered stretch of rows in every part, and the join needs neither sorts nor random gathers: a
// table of every unit's first row per part (part_unit_starts_kernel), then row i of part 0 = row start_p[u] + (i - start_0[u]) of
// part p (part_unit_join_kernel, which compares the keys it pairs: anything else -- groups outside the code range, a NULL-key group,
// a part that took another path -- fails the attempt and the sort join below does the work).
struct PartUnitArgs {
    DenseMap map;
    int64_t n;
    const uint64_t* key0;   // part 0: its order is the order of the joined run
    const uint64_t* mask0;
    const uint64_t* key;    // the part being placed
    const uint64_t* mask;
    const uint32_t* start0;
    uint32_t* start;
    uint64_t* key_out;      // (written with part 0)
    uint64_t* mask_out;
    unsigned long long* bad;
    int nw;
    const uint64_t* src[AGG_MAX_WORDS];
    uint64_t* dst[AGG_MAX_WORDS];
};
__device__ __forceinline__ bool part_unit_of(const DenseMap& m, uint64_t key, uint64_t mask, uint32_t* unit) {
    const uint64_t d = (key ^ m.sign) - m.lo_u;
    *unit = (((uint32_t)d * m.mul) & m.mask) >> 6;
    return mask == 0 && d <= (uint64_t)m.mask;
}
__global__ __launch_bounds__(256) void part_unit_starts_kernel(PartUnitArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        uint32_t u, up = ~0u;
        bad = bad || !part_unit_of(a.map, a.key[i], a.mask[i], &u);
        if (i > 0) part_unit_of(a.map, a.key[i - 1], 0, &up);
        if (u != up) a.start[u] = (uint32_t)i;
    }
    if (bad) __hip_atomic_store(a.bad, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void part_unit_join_kernel(PartUnitArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        const uint64_t k = a.key0[i];
        uint32_t u;
        part_unit_of(a.map, k, 0, &u);
        const int64_t j = (int64_t)a.start[u] + (i - (int64_t)a.start0[u]);
        if (j < 0 || j >= a.n || a.key[j] != k || a.mask[j] != 0) { bad = true; continue; }
        if (a.key_out) { a.key_out[i] = k; a.mask_out[i] = 0; }
        for (int w = 0; w < a.nw; w++) a.dst[w][i] = a.src[w][j];
    }
    if (bad) __hip_atomic_store(a.bad, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

void drop_parts(vnm_agg* h) {
    for (vnm_agg* c : h->parts) vnm_agg_destroy(c);
    h->parts.clear();
    h->part_funcs.clear();
}

// cut the function list by input column: `per` columns to a part (balanced), COUNT(*) with the first one
int make_parts(vnm_agg* h, int per) {
    const int nc = h->plan.n_cols;
    const int k = (nc + per - 1) / per;
    std::vector<int> part_of_col(nc);
    for (int c = 0; c < nc; c++) part_of_col[c] = (int)((int64_t)c * k / nc);
    h->part_funcs.assign(k, std::vector<int>());
    for (int i = 0; i < h->n_funcs; i++) h->part_funcs[h->func_col[i] < 0 ? 0 : part_of_col[h->func_col[i]]].push_back(i);
    const int kt = h->plan.key_types[0];
    for (int p = 0; p < k; p++) {
        int funcs[AGG_MAX_FUNCS], types[AGG_MAX_FUNCS], flags[AGG_MAX_FUNCS], ids[AGG_MAX_FUNCS];
        const int nf = (int)h->part_funcs[p].size();
        for (int q = 0; q < nf; q++) {
            const int i = h->part_funcs[p][q];
            funcs[q] = h->c_funcs[i]; types[q] = h->c_in_types[i]; flags[q] = h->c_in_flags[i]; ids[q] = h->func_col[i];   // (the distinct column: sharing survives)
        }
        vnm_agg* c = agg_create(VNM_SINGLE_NUMERICAL, 1, &kt, nf, funcs, types, flags, ids);
        if (!c) { drop_parts(h); return 1; }
        c->hint = h->hint;
        c->estimated = h->estimated;
        c->split_tried = true;
        c->async = h->async;   // (a stream: the parts record their batches like the parent would -- one launch per part over all of them)
        c->heavy_share = h->heavy_share;   // (the parent's sample saw the same key column)
        if (h->dense_state >= 1 && !h->range_given) {   // ... and its code map: no second range sample, and the parts' maps agree by construction
            c->dense_state = h->dense_state; c->dmap = h->dmap; c->dense_span = h->dense_span; c->dense_rlo = h->dense_rlo; c->dense_rhi = h->dense_rhi;
        }
        h->parts.push_back(c);
    }
    return 0;
}

int next_parts(vnm_agg* h, int64_t nrows, const vnm_dcol* keys, const vnm_dcol* inputs, const vnm_dcol* pred, void* stream) {
    const bool bound = (int64_t)h->parts.size() * nrows * 14 > env_i64("VNM_SPLIT_PENDING_BYTES", (int64_t)64 << 30);
    for (size_t p = 0; p < h->parts.size(); p++) {
        vnm_agg* c = h->parts[p];
        vnm_dcol in[AGG_MAX_FUNCS];
        const int nf = (int)h->part_funcs[p].size();
        for (int q = 0; q < nf; q++) in[q] = inputs[h->part_funcs[p][q]];
        VNM_TRY(vnm_agg_set_predicate(c, h->pred_set ? 1 : 0, h->pred_op, h->pred_is_float, h->pred_dval, h->pred_ival));
        c->child = true; c->cur_seq = h->cur_seq;
        VNM_TRY(vnm_agg_next_device(c, nrows, keys, in, pred, stream));
        // every part of the dense path keeps its scatter output (~10-18 bytes per row) for a deferred final pass: many parts over a
        // very large batch run their final passes right away instead of holding all of that at once
        if (bound && c->pending) VNM_TRY(complete_pending(c, as_stream(stream)));
    }
    h->rows_seen += nrows;
    return 0;
}

// The parts' results joined by key into ONE run of this operator (ascending key bits), the parts destroyed.  Every part
// grouped the same rows, so every part holds the same keys: sorting each part's keys gives the permutation that lines its
// accumulator words up with the others'.
int collapse_parts(vnm_agg* h, hipStream_t s) {
    if (h->parts.empty()) return 0;
    KernelTimer timer("agg_split_join", s);
    const int k = (int)h->parts.size();
    int64_t n = -1;
    for (int p = 0; p < k; p++) {
        int64_t np = 0;
        KernelTimer t2("agg_split_finish", s);
        VNM_TRY(vnm_agg_finish(h->parts[p], &np, (void*)s));
        if (n >= 0 && np != n) return set_error("aggregate: the parts of a split program disagree on the group count (%lld / %lld; internal error)", (long long)n, (long long)np);
        n = np;
    }
    if (h->have_run || h->have_table || h->pending) return set_error("aggregate: split program with state of its own (internal error)");
    if (n > 0) {
        PoolScope pool;
        const int64_t stride = n + 2;
        uint64_t* rk = (uint64_t*)pool_alloc((size_t)stride * 8 * 2);
        uint64_t* ra = (uint64_t*)pool_alloc((size_t)stride * 8 * h->plan.n_words);
        unsigned long long* flag = (unsigned long long*)pool.take(64);
        std::vector<int64_t*> perm(k, nullptr);
        bool ok = rk && ra && flag;
        if (!ok) { pool_free(rk); pool_free(ra); return 1; }
        int rc = hipMemsetAsync(flag, 0, 16, s) != hipSuccess;
        const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)device_info().num_cus * 8);
        // word lists of every part (parent word <- part word)
        std::vector<PartJoinArgs> jas(k);
        {
            std::vector<char> done(h->plan.n_words, 0);
            for (int p = 0; p < k && !rc; p++) {
                const vnm_agg* c = h->parts[p];
                PartJoinArgs& ja = jas[p];
                ja = PartJoinArgs{};
                ja.n = n; ja.key_src = c->dkey; ja.mask_src = c->dkey + c->dstride; ja.key_out = rk; ja.mask_out = rk + stride;
                ja.check = p > 0; ja.flag = flag;
                for (size_t q = 0; q < h->part_funcs[p].size() && !rc; q++) {
                    const FuncOut& mine = h->outs[h->part_funcs[p][q]];
                    const FuncOut& theirs = c->outs[q];
                    const int pw[3] = {mine.w_valid, mine.w_a, mine.w_b}, cw[3] = {theirs.w_valid, theirs.w_a, theirs.w_b};
                    for (int j = 0; j < 3; j++) {
                        if (pw[j] < 0 || done[pw[j]]) continue;
                        if (cw[j] < 0) { rc = set_error("aggregate: split program, accumulator layouts differ (internal error)"); break; }
                        done[pw[j]] = 1;
                        ja.src[ja.nw] = c->dacc + (size_t)cw[j] * c->dstride;
                        ja.dst[ja.nw++] = ra + (size_t)pw[j] * stride;
                    }
                }
            }
            for (int w = 0; w < h->plan.n_words && !rc; w++)
                if (!done[w]) rc = set_error("aggregate: split program, accumulator word %d has no source (internal error)", w);
        }
        // the join by units of 64 codes (parts of the dense path over one code map: see part_unit_starts_kernel)
        bool joined = false;
        DenseMap umap = h->parts[0]->dmap;
        const int ustate = h->parts[0]->dense_state;
        if (ustate == 2) { umap.mul = 1; umap.mul_inv = 1; }   // (the small-range scans address their tables by the plain code; a generic program that
                                                               // re-centred its range fails the key comparison below and takes the sort join)
        bool by_units = !rc && (ustate == 1 || ustate == 2) && n < ((int64_t)1 << 31) && getenv("VNM_AGG_NO_UNIT_JOIN") == nullptr;
        for (int p = 1; p < k && by_units; p++) by_units = h->parts[p]->dense_state == ustate && memcmp(&h->parts[p]->dmap, &h->parts[0]->dmap, sizeof(DenseMap)) == 0;
        if (by_units) {
            KernelTimer t2("agg_split_units", s);
            const size_t units = ((size_t)umap.mask + 1 + 63) >> 6;
            std::vector<uint32_t*> start(k, nullptr);
            for (int p = 0; p < k && by_units; p++) by_units = (start[p] = (uint32_t*)pool.take(units * 4)) != nullptr;
            for (int p = 0; p < k && by_units && !rc; p++) {
                PartUnitArgs ua{};
                ua.map = umap; ua.n = n; ua.key = jas[p].key_src; ua.mask = jas[p].mask_src; ua.start = start[p]; ua.bad = flag + 1;
                rc = hipMemsetAsync(start[p], 0xff, units * 4, s) != hipSuccess;
                if (!rc) part_unit_starts_kernel<<<grid, 256, 0, s>>>(ua);
            }
            for (int p = 0; p < k && by_units && !rc; p++) {
                PartUnitArgs ua{};
                ua.map = umap; ua.n = n; ua.key0 = jas[0].key_src; ua.mask0 = jas[0].mask_src; ua.key = jas[p].key_src; ua.mask = jas[p].mask_src;
                ua.start0 = start[0]; ua.start = start[p]; ua.bad = flag + 1;
                if (p == 0) { ua.key_out = rk; ua.mask_out = rk + stride; }
                ua.nw = jas[p].nw;
                for (int w = 0; w < ua.nw; w++) { ua.src[w] = jas[p].src[w]; ua.dst[w] = jas[p].dst[w]; }
                part_unit_join_kernel<<<grid, 256, 0, s>>>(ua);
            }
            unsigned long long ubad = 0;
            if (by_units && !rc && (hipGetLastError() != hipSuccess || hipMemcpyAsync(&ubad, flag + 1, 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
                                    hipStreamSynchronize(s) != hipSuccess))
                rc = set_error("aggregate: joining the parts of a split program failed");
            joined = by_units && !rc && ubad == 0;
        }
        const bool same_order = joined;
        const int asc = VNM_ASC;
        for (int p = 0; p < k && !rc && !same_order; p++) {
            vnm_dcol kc{};
            kc.values = h->parts[p]->dkey; kc.type = VNM_U64; kc.length = n;
            KernelTimer t2("agg_split_sort", s);
            if (!(perm[p] = (int64_t*)pool.take((size_t)n * 8))) { pool_free(rk); pool_free(ra); return 1; }
            rc = vnm_sort_indices(1, &kc, &asc, n, 0, perm[p], (void*)s);
        }
        for (int p = 0; p < k && !rc && !joined; p++) {
            jas[p].perm = perm[p];
            part_join_kernel<<<grid, 256, 0, s>>>(jas[p]);
        }
        unsigned long long bad = 0;
        if (!rc && (hipGetLastError() != hipSuccess || hipMemcpyAsync(&bad, flag, 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
                    hipStreamSynchronize(s) != hipSuccess))
            rc = set_error("aggregate: joining the parts of a split program failed");
        if (!rc && bad) rc = set_error("aggregate: the parts of a split program hold different keys (internal error)");
        if (rc) { pool_free(rk); pool_free(ra); return rc; }
        h->run_key = rk; h->run_acc = ra; h->run_stride = stride; h->run_n = n; h->have_run = true;
    }
    drop_parts(h);
    return 0;
}

vnm_agg* agg_create(int kind, int n_keys, const int* key_types, int n_funcs, const int* funcs,
                    const int* in_types, const int* in_flags, const int* in_col_ids) {
    if (ensure_init()) return nullptr;
    vnm_agg* h = new vnm_agg();
    if (build_plan(kind, n_keys, key_types, n_funcs, funcs, in_types, in_flags, in_col_ids, &h->plan, h->outs)) {
        delete h;
        return nullptr;
    }
    h->n_funcs = n_funcs;
    // distinct column index per func: replay build_plan's assignment
    int ids[AGG_MAX_COLS], nc = 0;
    for (int i = 0; i < n_funcs; i++) {
        h->func_col[i] = -1;
        if (funcs[i] == VNM_COUNT_STAR) continue;
        int id = in_col_ids ? in_col_ids[i] : (1000 + i);
        if (id < 0) id = -1000 - i;   // as build_plan: negative = this function's own column
        for (int c = 0; c < nc; c++) if (ids[c] == id) h->func_col[i] = c;
        if (h->func_col[i] < 0) { ids[nc] = id; h->col_first_func[nc] = i; h->func_col[i] = nc++; }
    }
    h->single = (kind == VNM_SINGLE_NUMERICAL) || (kind == VNM_MULTI_NUMERICAL && n_keys == 1);
    for (int i = 0; i < n_funcs; i++) {
        h->c_funcs[i] = funcs[i];
        h->c_in_types[i] = in_types ? in_types[i] : 0;
        h->c_in_flags[i] = in_flags ? in_flags[i] : 0;
        h->c_in_col_ids[i] = in_col_ids ? in_col_ids[i] : 0;
    }
    h->c_has_ids = in_col_ids != nullptr;
    return h;
}

}  // namespace

#include "vnm_agg_exact.inc"

extern "C" {

vnm_agg* vnm_agg_create(int kind, int n_keys, const int* key_types, int n_funcs, const int* funcs,
                        const int* in_types, const int* in_flags, const int* in_col_ids) {
    vnm_agg* h = agg_create(kind, n_keys, key_types, n_funcs, funcs, in_types, in_flags, in_col_ids);
    if (h) exact_attach(h);
    return h;
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
    delete h;
}

int vnm_agg_set_predicate(vnm_agg* h, int enabled, int op, int scalar_is_float, double dval, int64_t ival) {
    if (!h) return set_error("vnm_agg_set_predicate: null handle");
    if (enabled && (op < VNM_EQ || op > VNM_LE)) return set_error("vnm_agg_set_predicate: bad comparison op %d", op);
    const bool same = h->pred_set == (enabled != 0) && (!enabled || (h->pred_op == op && h->pred_is_float == scalar_is_float && h->pred_ival == ival &&
                                                                     memcmp(&h->pred_dval, &dval, 8) == 0));
    if (same) return 0;
    // the waiting batches of an asynchronous stream were recorded under the OLD predicate (and without a predicate column when none was set)
    if (!h->q.empty()) return set_error("vnm_agg_set_predicate: batches are waiting (call vnm_agg_sync first)");
    for (vnm_agg* c : h->parts) if (!c->q.empty()) return set_error("vnm_agg_set_predicate: batches are waiting (call vnm_agg_sync first)");
    h->pred_set = enabled != 0;
    h->pred_op = op;
    h->pred_is_float = scalar_is_float;
    h->pred_dval = dval;
    h->pred_ival = ival;
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
            for (int j = 0; j < h->plan.n_keys; j++) h->pack.cols[j] = keys[j];
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
            VNM_SEG_ONLY("small-range LDS scan");
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

int vnm_agg_next_device(vnm_agg* h, int64_t nrows, const vnm_dcol* keys, const vnm_dcol* inputs,
                        const vnm_dcol* pred, void* stream) {
    if (!h) return set_error("vnm_agg_next_device: null handle");
    if (!h->child) h->cur_seq = h->seq++;
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
    if (h->c_in_types[func_idx] != VNM_F64) return set_error("vnm_agg_set_input_expr: declare the function's input type as float64 (the expression's result type)");
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
    if (h->ex && h->ex->switched) return exact_finish(h, n_groups, stream);   // prefix + suffix merged, float MIN / MAX composed in row order
    return agg_finish_core(h, n_groups, stream);
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

// ---- partition-aligned exchange (multi-GPU, large G) ------------------------------------------------------
// Number of final partitions when the finished result is a partition-structured run (partitioned path,
// nothing merged into it since), else 0.  All ranks must report the same non-zero value to use this exchange.
int64_t vnm_agg_run_partitions(vnm_agg* h) {
    if (!h) return 0;
    int64_t n = 0;
    if (vnm_agg_finish(h, &n, nullptr) != 0) return 0;
    if (!(h->result_is_run && h->run_dir) || (h->ex && h->ex->switched)) return 0;
    // only what vnm_agg_merge_partitioned can merge: add-merge words (MIN / MAX programs and anything wider than the
    // LDS merge table go through the owner-bucketed exchange)
    if (h->plan.n_words > PM_MAX_WORDS) return 0;
    for (int w = 0; w < h->plan.n_words; w++)
        if (h->plan.merge[w] != M_ADD_U64 && h->plan.merge[w] != M_ADD_F64 && h->plan.merge[w] != M_ADD_F64C) return 0;
    return h->run_nfin;
}

// Rows of the run in partition order, row-major [n][2 + n_acc_words]; per-partition row counts (device,
// uint32[nfin]); rows per owner (host) with owner(f) = f * world / nfin.
int vnm_agg_run_reorder(vnm_agg* h, int world, uint64_t* out_rows, uint32_t* out_part_counts, int64_t* owner_counts_host,
                        void* stream) {
    VNM_TRY(ensure_init());
    if (!h || world < 1 || !out_part_counts || !owner_counts_host) return set_error("vnm_agg_run_reorder: bad argument");
    if (!vnm_agg_run_partitions(h)) return set_error("vnm_agg_run_reorder: the result is not a partition-structured run");
    hipStream_t s = as_stream(stream);
    RunReorderArgs a{};
    a.nw = 2 + h->plan.n_words;
    a.words[0] = h->run_key;
    a.words[1] = h->run_key + h->run_stride;
    for (int w = 0; w < h->plan.n_words; w++) a.words[2 + w] = h->run_acc + (size_t)w * h->run_stride;
    a.nfin = h->run_nfin;
    a.dir = h->run_dir;
    a.out = out_rows;
    a.part_counts = out_part_counts;
    PoolScope pool;   // (every way out frees the prefix)
    a.prefix = (unsigned long long*)pool.take((size_t)(a.nfin + 1) * 8);
    if (!a.prefix) return 1;
    run_prefix_kernel<<<1, 1024, 0, s>>>(a);
    run_reorder_kernel<<<(int)std::min<int64_t>(a.nfin, (int64_t)device_info().num_cus * 8), 256, 0, s>>>(a);
    VNM_HIP(hipGetLastError());
    std::vector<unsigned long long> bounds((size_t)world + 1);
    for (int o = 0; o <= world; o++) {
        int64_t f = (int64_t)o * a.nfin / world;  // first partition of owner o (inverse of owner(f) = f * world / nfin)
        while (f < a.nfin && f > 0 && (f * world) / a.nfin < o) f++;
        VNM_HIP(hipMemcpyAsync(&bounds[o], a.prefix + f, 8, hipMemcpyDeviceToHost, s));
    }
    VNM_HIP(hipStreamSynchronize(s));
    for (int o = 0; o < world; o++) owner_counts_host[o] = (int64_t)(bounds[o + 1] - bounds[o]);
    return 0;
}

// Owner side: merge the segments received from all ranks.  rows: received rows grouped by source rank (each in
// partition order); src_row_offsets_host[world + 1]: row offset of every source block; part_counts: device
// uint32 [world][nlocal] rows per (source, owned partition).  The merged groups become this handle's result.
int vnm_agg_merge_partitioned(vnm_agg* h, int world, int64_t nlocal, const uint64_t* rows, const int64_t* src_row_offsets_host,
                              const uint32_t* part_counts, void* stream) {
    VNM_TRY(ensure_init());
    if (!h || world < 1 || nlocal < 1) return set_error("vnm_agg_merge_partitioned: bad argument");
    if (h->have_table || h->have_run) return set_error("vnm_agg_merge_partitioned: the handle must be empty");
    if (h->plan.n_words > PM_MAX_WORDS) return set_error("vnm_agg_merge_partitioned: at most %d accumulator words", PM_MAX_WORDS);
    for (int w = 0; w < h->plan.n_words; w++)
        if (h->plan.merge[w] != M_ADD_U64 && h->plan.merge[w] != M_ADD_F64 && h->plan.merge[w] != M_ADD_F64C) return set_error("vnm_agg_merge_partitioned: add-merge words only");
    hipStream_t s = as_stream(stream);
    invalidate_result(h);
    // per-source exclusive prefix over the owned partitions -> absolute row offsets
    std::vector<uint32_t> pc((size_t)world * nlocal);
    VNM_HIP(hipMemcpyAsync(pc.data(), part_counts, pc.size() * 4, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    std::vector<unsigned long long> pre((size_t)world * (nlocal + 1));
    int64_t total = 0;
    for (int r = 0; r < world; r++) {
        unsigned long long run = (unsigned long long)src_row_offsets_host[r];
        for (int64_t f = 0; f < nlocal; f++) { pre[(size_t)r * (nlocal + 1) + f] = run; run += pc[(size_t)r * nlocal + f]; }
        pre[(size_t)r * (nlocal + 1) + nlocal] = run;
        if ((int64_t)run != src_row_offsets_host[r + 1]) return set_error("vnm_agg_merge_partitioned: partition counts of source %d do not add up", r);
        total = (int64_t)run;
    }
    PoolScope pool;   // every way out frees the four blocks; the run's two are handed to the handle (keep) on success
    unsigned long long* dpre = (unsigned long long*)pool.take(pre.size() * 8);
    unsigned long long* flags = (unsigned long long*)pool.take(64);
    const int64_t dstride = total + 2;
    uint64_t* rk = (uint64_t*)pool.take((size_t)dstride * 16);
    uint64_t* ra = (uint64_t*)pool.take((size_t)dstride * 8 * h->plan.n_words);
    if (!dpre || !flags || !rk || !ra) return 1;
    VNM_HIP(hipMemcpyAsync(dpre, pre.data(), pre.size() * 8, hipMemcpyHostToDevice, s));
    VNM_HIP(hipMemsetAsync(flags, 0, 64, s));
    PartMergeArgs m{};
    m.rows = rows; m.src_prefix = dpre; m.world = world; m.nlocal = nlocal;
    m.nw = 2 + h->plan.n_words; m.n_words = h->plan.n_words;
    for (int w = 0; w < h->plan.n_words; w++) m.merge[w] = h->plan.merge[w];
    m.dkey = rk; m.dacc = ra; m.dstride = dstride; m.flags = flags;
    {
        KernelTimer timer("agg_part_merge", s);
        int occ = 0;  // one resident set of workgroups, as in the final pass
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)part_merge_kernel, PA_BLOCK, 0) != hipSuccess || occ < 1) occ = 2;
        part_merge_kernel<<<(int)std::min<int64_t>(nlocal, (int64_t)device_info().num_cus * occ), PA_BLOCK, 0, s>>>(m);
    }
    VNM_HIP(hipGetLastError());
    unsigned long long fl[2];
    VNM_HIP(hipMemcpyAsync(fl, flags, 16, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    if (fl[0]) {
        set_error("vnm_agg_merge_partitioned: a partition holds more groups than the LDS table (use vnm_agg_merge_rows)");
        return 2;  // capacity, not an error of the data: the handle is still empty and the caller falls back
    }
    pool.keep(rk); pool.keep(ra);
    h->run_key = rk; h->run_acc = ra; h->run_stride = dstride; h->run_n = (int64_t)fl[1];
    h->have_run = true;
    return 0;
}

int vnm_agg_layout(vnm_agg* h, int* n_key_words, int* n_acc_words) {
    if (!h) return set_error("vnm_agg_layout: null handle");
    if (n_key_words) *n_key_words = h->plan.kw;
    if (n_acc_words) *n_acc_words = h->plan.n_words;
    return 0;
}

int vnm_agg_dense_ptrs(vnm_agg* h, uint64_t** key_words, uint64_t** acc_words) {
    if (!h) return set_error("vnm_agg_dense_ptrs: null handle");
    if (h->n_groups < 0) return set_error("vnm_agg_dense_ptrs: call vnm_agg_finish first");
    for (int j = 0; j < h->plan.kw; j++) key_words[j] = h->dkey ? h->dkey + (size_t)j * h->dstride : nullptr;
    for (int w = 0; w < h->plan.n_words; w++) acc_words[w] = h->dacc ? h->dacc + (size_t)w * h->dstride : nullptr;
    return 0;
}

static int fetch_host(vnm_agg* h) {
    if (h->host_ready) return 0;
    int64_t n = 0;
    VNM_TRY(vnm_agg_finish(h, &n, nullptr));
    h->h_key.assign((size_t)(h->plan.kw ? h->plan.kw : 1) * (n ? n : 1), 0);
    h->h_acc.assign((size_t)h->plan.n_words * (n ? n : 1), 0);
    if (n > 0) {
        for (int j = 0; j < h->plan.kw; j++)
            VNM_HIP(hipMemcpy(h->h_key.data() + (size_t)j * n, h->dkey + (size_t)j * h->dstride, (size_t)n * 8, hipMemcpyDeviceToHost));
        for (int w = 0; w < h->plan.n_words; w++)
            VNM_HIP(hipMemcpy(h->h_acc.data() + (size_t)w * n, h->dacc + (size_t)w * h->dstride, (size_t)n * 8, hipMemcpyDeviceToHost));
    }
    h->host_ready = true;
    return 0;
}

int vnm_agg_result_key(vnm_agg* h, int key_idx, uint64_t* vals, uint8_t* valid) {
    if (!h) return set_error("vnm_agg_result_key: null handle");
    if (key_idx < 0 || key_idx >= h->plan.n_keys) return set_error("vnm_agg_result_key: key index out of range");
    VNM_TRY(fetch_host(h));
    int64_t n = h->n_groups;
    const uint64_t* k = h->h_key.data() + (size_t)key_idx * n;
    const uint64_t* nm = h->h_key.data() + (size_t)h->plan.n_keys * n;
    for (int64_t r = 0; r < n; r++) {
        vals[r] = k[r];
        valid[r] = !((nm[r] >> key_idx) & 1);
    }
    return 0;
}

int vnm_agg_result_func(vnm_agg* h, int func_idx, void* cells16, uint8_t* valid, int* out_kind) {
    if (!h) return set_error("vnm_agg_result_func: null handle");
    if (func_idx < 0 || func_idx >= h->n_funcs) return set_error("vnm_agg_result_func: function index out of range");
    VNM_TRY(fetch_host(h));
    int64_t n = h->n_groups;
    const uint64_t* words[AGG_MAX_WORDS];
    for (int w = 0; w < h->plan.n_words; w++) words[w] = h->h_acc.data() + (size_t)w * n;
    return finalize_func(h->outs[func_idx], n, words, cells16, valid, out_kind);
}

// ---- device-side result columns ---------------------------------------------------------------------------
namespace {
void fin_key_args(vnm_agg* h, int key_idx, FinArgs& f) {
    f.is_key = 1;
    f.key_bit = key_idx;
    f.out_width = type_width(h->plan.key_types[key_idx]);
    if (h->n_groups > 0) {
        f.words[1] = h->dkey + (size_t)key_idx * h->dstride;
        f.words[0] = h->dkey + (size_t)h->plan.n_keys * h->dstride;
    }
}

int fin_func_args(vnm_agg* h, int func_idx, FinArgs& f) {  // returns the VNM_OUT_* kind of the column
    const FuncOut& fo = h->outs[func_idx];
    const int t = fo.in_type;
    f.fo = fo;
    int kind = VNM_OUT_U64, width = 8;
    switch (fo.func) {
        case VNM_COUNT_STAR: case VNM_COUNT: break;
        case VNM_MIN: case VNM_MAX:  // type preserving (agg_func_factory.cpp:35-107)
            kind = t == VNM_F64 ? VNM_OUT_F64 : (t == VNM_F32 ? VNM_OUT_F32 : (type_is_unsigned(t) ? VNM_OUT_U64 : VNM_OUT_I64));
            width = type_width(t);
            break;
        case VNM_SUM:
            if (type_is_float(t)) kind = VNM_OUT_F64;
            else if (t == VNM_I32 && (fo.in_flags & VNM_FLAG_SUM32)) { kind = VNM_OUT_I32; width = 4; }
            else kind = type_is_unsigned(t) ? VNM_OUT_U64 : VNM_OUT_I64;
            break;
        default:
            f.out_f32 = (t == VNM_I8 || t == VNM_I16 || t == VNM_U8 || t == VNM_U16);
            kind = f.out_f32 ? VNM_OUT_F32 : VNM_OUT_F64;
            width = f.out_f32 ? 4 : 8;
            break;
    }
    f.out_width = width;
    if (h->n_groups > 0) {
        auto word = [&](int w) -> const uint64_t* { return w >= 0 ? h->dacc + (size_t)w * h->dstride : nullptr; };
        f.words[0] = word(fo.w_valid);
        f.words[1] = word(fo.w_a);
        f.words[2] = word(fo.w_b);
    }
    return kind;
}
}  // namespace

int vnm_agg_result_device(vnm_agg* h, int n_cols, const int* which, void* const* out_values, uint8_t* const* out_bitmaps,
                          int* out_kinds, int64_t* null_counts, void* stream) {
    if (!h) return set_error("vnm_agg_result_device: null handle");
    if (n_cols < 0 || (n_cols > 0 && (!which || !out_values || !out_bitmaps))) return set_error("vnm_agg_result_device: bad arguments");
    for (int c = 0; c < n_cols; c++) {
        const int w = which[c];
        if (w >= 0 ? w >= h->n_funcs : ~w >= h->plan.n_keys) return set_error("vnm_agg_result_device: column index out of range");
    }
    VNM_TRY(vnm_agg_finish(h, nullptr, stream));
    hipStream_t s = as_stream(stream);
    const int64_t n = h->n_groups;
    int rc = 0;
    for (int base = 0; base < n_cols; base += FIN_MAX_COLS) {
        const int nc = std::min(FIN_MAX_COLS, n_cols - base);
        FinMulti m{};
        m.n_cols = nc;
        m.n = n;
        unsigned long long* ctl = n > 0 ? (unsigned long long*)pool_alloc(16 * FIN_MAX_COLS) : nullptr;
        if (n > 0 && !ctl) return 1;
        for (int c = 0; c < nc; c++) {
            FinArgs& f = m.col[c];
            const int w = which[base + c];
            int kind = -1;
            if (w < 0) fin_key_args(h, ~w, f);
            else kind = fin_func_args(h, w, f);
            if (out_kinds) out_kinds[base + c] = kind;
            if (null_counts) null_counts[base + c] = 0;
            f.n = n;
            f.out = out_values[base + c];
            f.bitmap = (unsigned long long*)out_bitmaps[base + c];
            f.ctl = ctl + 2 * c;
        }
        if (n == 0) continue;
        VNM_HIP(hipMemsetAsync(ctl, 0, 16 * FIN_MAX_COLS, s));
        // one thread per group (a grid-stride loop over 16 workgroups per CU: 1.38 instead of 1.21 ms for 1e8 groups x 3 columns)
        const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)1 << 30);
        {
            KernelTimer timer("agg_finalize", s);
            agg_finalize_kernel<<<grid, 256, 0, s>>>(m);
        }
        VNM_HIP(hipGetLastError());
        unsigned long long c2[2 * FIN_MAX_COLS];
        VNM_HIP(hipMemcpyAsync(c2, ctl, 16 * FIN_MAX_COLS, hipMemcpyDeviceToHost, s));
        VNM_HIP(hipStreamSynchronize(s));
        pool_free(ctl);
        for (int c = 0; c < nc; c++) {
            if (null_counts) null_counts[base + c] = (int64_t)c2[2 * c];
            if (c2[2 * c + 1] & 2ULL) return set_error("aggregate: a group holds 2^32 or more non-NULL inputs of a 64-bit integer SUM / AVG (the exact 128-bit lanes hold 2^32 - 1 per group)");
            if (c2[2 * c + 1]) rc = 2;
        }
    }
    return rc;
}

int vnm_agg_result_key_device(vnm_agg* h, int key_idx, void* out_values, uint8_t* out_bitmap, int64_t* null_count, void* stream) {
    if (!h) return set_error("vnm_agg_result_key_device: null handle");
    if (key_idx < 0 || key_idx >= h->plan.n_keys) return set_error("vnm_agg_result_key_device: key index out of range");
    const int which = ~key_idx;
    return vnm_agg_result_device(h, 1, &which, &out_values, &out_bitmap, nullptr, null_count, stream);
}

int vnm_agg_result_func_device(vnm_agg* h, int func_idx, void* out_values, uint8_t* out_bitmap, int* out_kind, int64_t* null_count,
                               void* stream) {
    if (!h) return set_error("vnm_agg_result_func_device: null handle");
    if (func_idx < 0 || func_idx >= h->n_funcs) return set_error("vnm_agg_result_func_device: function index out of range");
    return vnm_agg_result_device(h, 1, &func_idx, &out_values, &out_bitmap, out_kind, null_count, stream);
}

// BaseAggregate::Result with library-allocated output columns.  When the last batch went through the dense path and its
// final pass is still pending, that pass writes the result columns itself (fused finalisation); otherwise finish + the
// finalisation kernel.  out_values[c] / out_bitmaps[c] are vnm_malloc blocks the CALLER frees (vnm_free); a column without
// NULLs gets no bitmap (nullptr).
int vnm_agg_result_device_alloc(vnm_agg* h, int n_cols, const int* which, void** out_values, uint8_t** out_bitmaps, int* out_kinds,
                                int64_t* null_counts, int64_t* n_groups, void* stream) {
    VNM_TRY(ensure_init());
    if (!h) return set_error("vnm_agg_result_device_alloc: null handle");
    if (n_cols < 0 || (n_cols > 0 && (!which || !out_values || !out_bitmaps)) || !n_groups) return set_error("vnm_agg_result_device_alloc: bad arguments");
    for (int c = 0; c < n_cols; c++) {
        const int w = which[c];
        if (w >= 0 ? w >= h->n_funcs : ~w >= h->plan.n_keys) return set_error("vnm_agg_result_device_alloc: column index out of range");
        out_values[c] = nullptr; out_bitmaps[c] = nullptr;
    }
    hipStream_t s = as_stream(stream);
    VNM_TRY(flush_queue(h, stream));   // (the waiting batches of an asynchronous stream)
    auto free_outputs = [&]() { for (int c = 0; c < n_cols; c++) { pool_free(out_values[c]); pool_free(out_bitmaps[c]); out_values[c] = nullptr; out_bitmaps[c] = nullptr; } };
    DensePending* pd = h->inner ? nullptr : h->pending;
    bool fused = pd && !h->have_run && h->n_groups < 0 && n_cols >= 1 && n_cols <= DF_MAX_OUT && getenv("VNM_AGG_NO_FUSED_RESULT") == nullptr &&
                 !(h->ex && h->ex->switched);   // (an ordered MIN / MAX stream: the handle's own state is the prefix only)
    // An HBM table next to the pending pass (spilled heavy keys, keys outside the code range, the NULL-key group, rows of batches that
    // took the scan): while it holds FEW groups the pass folds them into its own result columns (dside_merge / dside_append_kernel)
    int64_t side_groups = 0;
    PoolScope side_pool;
    DFinalArgs cols{};
    if (fused && h->have_table) {
        unsigned long long fill = 0;
        VNM_HIP(hipMemcpyAsync(&fill, h->g.ctl + 2, 8, hipMemcpyDeviceToHost, s));
        VNM_HIP(hipStreamSynchronize(s));
        fused = pd->fsplits == 1 && h->g.kwt == 0 && (int64_t)fill <= env_i64("VNM_AGG_SIDE_MAX_GROUPS", 1 << 22) && h->g.cap <= (1ULL << 26) &&
                getenv("VNM_AGG_NO_SIDE_FUSION") == nullptr;
        if (getenv("VNM_AGG_TRACE")) fprintf(stderr, "[agg] result columns: side table with %llu groups in %llu slots, splits %d -> %s\n", fill,
                                             (unsigned long long)h->g.cap, pd->fsplits, fused ? "folded into the final pass" : "run + table merge");
        if (fused) {
            side_groups = (int64_t)fill + 2;
            uint64_t bbits = DSIDE_BLOOM_BITS;      // ~64 filter bits per key: 2^16 (a copy in LDS) ... 2^26
            while (bbits < fill * 64 && bbits < (1ULL << 26)) bbits <<= 1;
            uint32_t* bloom = (uint32_t*)side_pool.take((size_t)bbits / 8);
            uint8_t* found = (uint8_t*)side_pool.take((size_t)h->g.cap + 2);
            if (!bloom || !found) return 1;
            VNM_HIP(hipMemsetAsync(bloom, 0, (size_t)bbits / 8, s));
            dside_bloom_kernel<<<(int)std::min<int64_t>(((int64_t)h->g.cap + 255) / 256, (int64_t)device_info().num_cus * 8), 256, 0, s>>>(h->g, bloom, (uint32_t)(bbits - 1));
            VNM_HIP(hipGetLastError());
            cols.has_side = 1; cols.side = h->g; cols.side_bloom = bloom; cols.side_bloom_mask = (uint32_t)(bbits - 1); cols.side_found = found;
        }
    }
    for (int c = 0; c < n_cols && fused; c++) {
        const int w = which[c];
        if (w < 0) { cols.out_kind[c] = DF_KEY; if (out_kinds) out_kinds[c] = -1; continue; }
        const FuncOut& fo = h->outs[w];
        if (fo.func == VNM_COUNT_STAR || fo.func == VNM_COUNT) { cols.out_kind[c] = DF_COUNT; if (out_kinds) out_kinds[c] = VNM_OUT_U64; }
        else if (fo.func == VNM_SUM && fo.in_type == VNM_F64) { cols.out_kind[c] = DF_SUM; if (out_kinds) out_kinds[c] = VNM_OUT_F64; }
        else if (fo.func == VNM_AVG && fo.in_type == VNM_F64) { cols.out_kind[c] = DF_AVG; if (out_kinds) out_kinds[c] = VNM_OUT_F64; }
        else fused = false;
    }
    if (fused) {
        cols.n_out = n_cols;
        cols.dstride = pd->dstride + side_groups;
        for (int c = 0; c < n_cols; c++) {
            out_values[c] = pool_alloc((size_t)cols.dstride * 8);
            if (!out_values[c]) { free_outputs(); return 1; }
            cols.out_ptr[c] = out_values[c];
            if (null_counts) null_counts[c] = 0;   // every group of this shape saw a non-NULL input: no NULL results
        }
        int64_t n = 0, null_pos = 0;
        cols.null_pos = &null_pos;
        route_note(cols.has_side ? "result:fused_columns_with_side_table" : "result:fused_columns", "%d columns written by the deferred final pass (%zu batches, %lld side groups)", n_cols, pd->sets.size(), (long long)side_groups);
        const int rc = complete_pending(h, s, DF_COLS, &cols, &n);
        if (rc) { free_outputs(); return rc; }
        if (null_pos > 0) {   // the NULL-key group: the key columns get a validity bitmap with that one bit cleared
            for (int c = 0; c < n_cols; c++) {
                if (which[c] >= 0) continue;
                const size_t nb = (size_t)((n + 63) / 64 + 1) * 8;
                uint8_t* bm = (uint8_t*)pool_alloc(nb);
                if (!bm) { free_outputs(); return 1; }
                out_bitmaps[c] = bm;
                const int64_t p = null_pos - 1;
                const uint8_t byte = (uint8_t)(0xFFu & ~(1u << (p & 7)));
                if (hipMemsetAsync(bm, 0xFF, nb, s) != hipSuccess || hipMemcpyAsync(bm + (p >> 3), &byte, 1, hipMemcpyHostToDevice, s) != hipSuccess ||
                    hipStreamSynchronize(s) != hipSuccess) { free_outputs(); return set_error("vnm_agg_result_device_alloc: key bitmap failed"); }
                if (null_counts) null_counts[c] = 1;
            }
        }
        *n_groups = n;
        return 0;
    }
    int64_t n = 0;
    route_note("result:finish_then_finalize", "%d columns from the dense partial state", n_cols);
    VNM_TRY(vnm_agg_finish(h, &n, stream));
    *n_groups = n;
    std::vector<int64_t> nulls((size_t)std::max(n_cols, 1), 0);
    for (int c = 0; c < n_cols; c++) {
        out_values[c] = pool_alloc((size_t)std::max<int64_t>(n, 1) * 8);
        out_bitmaps[c] = (uint8_t*)pool_alloc((size_t)((n + 63) / 64 + 1) * 8);
        if (!out_values[c] || !out_bitmaps[c]) { free_outputs(); return 1; }
    }
    const int rc = vnm_agg_result_device(h, n_cols, which, out_values, out_bitmaps, out_kinds, nulls.data(), stream);
    if (rc) { free_outputs(); return rc; }
    for (int c = 0; c < n_cols; c++) {
        if (null_counts) null_counts[c] = nulls[c];
        if (nulls[c] == 0) { pool_free(out_bitmaps[c]); out_bitmaps[c] = nullptr; }
    }
    return 0;
}

// ---- multi-GPU: the dense path with a code range all ranks agree on ---------------------------------------------------------
// vnm_agg_dense_range: this rank's sampled key range of a batch (order-preserving unsigned images; lo > hi: no dense path for this
// key type / range).  The ranks reduce lo by MIN and hi by MAX and hand the result to vnm_agg_set_dense_range: every rank then
// derives the SAME code map, so the direct-addressed tables of the final pass are slot-compatible across ranks.
int vnm_agg_dense_range(vnm_agg* h, int64_t nrows, const vnm_dcol* key, uint64_t* lo, uint64_t* hi, void* stream) {
    VNM_TRY(ensure_init());
    if (!h || !key || !lo || !hi) return set_error("vnm_agg_dense_range: bad argument");
    *lo = ~0ULL; *hi = 0;
    if (!h->single || h->plan.n_keys != 1 || nrows <= 0 || (key->type != VNM_I64 && key->type != VNM_U64) || key->validity) return 0;
    hipStream_t s = as_stream(stream);
    const uint64_t sign = key->type == VNM_I64 ? 0x8000000000000000ULL : 0ULL;
    PoolScope pool;   // (every early return gives the block back)
    unsigned long long* d = (unsigned long long*)pool.take(64);
    if (!d) return 1;
    const unsigned long long init[2] = {~0ULL, 0ULL};
    unsigned long long got[2];
    VNM_HIP(hipMemcpyAsync(d, init, 16, hipMemcpyHostToDevice, s));
    const int64_t m = std::min<int64_t>(nrows, 1 << 18);
    const int grid = (int)std::min<int64_t>((m + 255) / 256, (int64_t)device_info().num_cus * 4);
    dense_sample_range_kernel<<<grid, 256, 0, s>>>((const uint64_t*)key->values + key->offset, nrows, m, sign, d, nullptr, 0);
    VNM_HIP(hipGetLastError());
    VNM_HIP(hipMemcpyAsync(got, d, 16, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    *lo = got[0]; *hi = got[1];
    return 0;
}

int vnm_agg_set_dense_range(vnm_agg* h, int key_type, uint64_t lo, uint64_t hi) {
    if (!h) return set_error("vnm_agg_set_dense_range: null handle");
    h->range_given = false;
    h->dense_state = -1;
    if (lo > hi) return 0;
    VNM_TRY(plan_dense_from_range(h, key_type, lo, hi));
    h->range_given = h->dense_state == 1;
    return 0;
}

// The direct-addressed tables of the (deferred) final pass: *table = 2^bits slots of {sum f64, lo f32, count u32}, slot = code'.
// *table = nullptr when the handle holds anything else (no dense batch, spilled keys, several batches): use another exchange.
// The geometry words let the ranks check that they really derived the same map: {lo_u, bits, mul, sign}.
int vnm_agg_dense_table(vnm_agg* h, void** table, int* bits, uint64_t* geometry, void* stream) {
    VNM_TRY(ensure_init());
    if (!h || !table || !bits) return set_error("vnm_agg_dense_table: bad argument");
    *table = nullptr; *bits = 0;
    VNM_TRY(flush_queue(h, stream));
    DensePending* pd = h->inner ? nullptr : h->pending;
    if (!pd || h->have_table || h->have_run || h->n_groups >= 0 || (h->ex && h->ex->switched)) return 0;
    const int rc = complete_pending(h, as_stream(stream), DF_TABLE);
    if (rc == 2) return 0;
    if (rc) return rc;
    *table = pd->table;
    *bits = pd->df.map.bits;
    if (geometry) { geometry[0] = pd->df.map.lo_u; geometry[1] = (uint64_t)pd->df.map.bits; geometry[2] = pd->df.map.mul; geometry[3] = pd->df.map.sign; }
    return 0;
}

// Owner side: `nsrc` slices of such tables (slots [code0, code0 + n) of every rank, in rank order) -> this (empty) handle's result.
// The map is the one of `like` (any handle that produced one of the tables).
int vnm_agg_merge_dense_tables(vnm_agg* h, const vnm_agg* like, int nsrc, const void* const* slices, uint64_t code0, int64_t n, void* stream) {
    VNM_TRY(ensure_init());
    if (!h || !like || !like->pending || nsrc < 1 || nsrc > 64 || !slices || n < 0) return set_error("vnm_agg_merge_dense_tables: bad argument");
    if (h->have_table || h->have_run || h->pending) return set_error("vnm_agg_merge_dense_tables: the handle must be empty");
    if (h->plan.n_words != like->plan.n_words) return set_error("vnm_agg_merge_dense_tables: different aggregate programs");
    hipStream_t s = as_stream(stream);
    invalidate_result(h);
    const DFinalArgs& ld = like->pending->df;
    PoolScope pool;
    const int64_t dstride = n + 2;
    uint64_t* rk = (uint64_t*)pool.take((size_t)dstride * 8 * 2);
    uint64_t* ra = (uint64_t*)pool.take((size_t)dstride * 8 * h->plan.n_words);
    unsigned long long* flags = (unsigned long long*)pool.take(64);
    if (!rk || !ra || !flags) return 1;
    VNM_HIP(hipMemsetAsync(flags, 0, 64, s));
    DTabMergeArgs a{};
    a.map = ld.map;
    for (int r = 0; r < nsrc; r++) a.src[r] = (const DTabSlot*)slices[r];
    a.nsrc = nsrc; a.n = n; a.code0 = (uint32_t)code0;
    a.w_rows = ld.w_rows; a.w_valid = ld.w_valid; a.w_sum = ld.w_sum; a.w_lo = ld.w_lo;
    a.dkey = rk; a.dacc = ra; a.dstride = dstride; a.flags = flags;
    if (n > 0) {
        KernelTimer timer("agg_table_merge", s);
        dtable_merge_kernel<<<(int)((n + DTM_PER * DTM_BLOCK - 1) / (DTM_PER * DTM_BLOCK)), DTM_BLOCK, 0, s>>>(a);
    }
    VNM_HIP(hipGetLastError());
    unsigned long long fl[2];
    VNM_HIP(hipMemcpyAsync(fl, flags, 16, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    if (fl[0]) return set_error("vnm_agg_merge_dense_tables: output overflow (internal error)");
    pool.keep(rk); pool.keep(ra);
    h->run_key = rk; h->run_acc = ra; h->run_stride = dstride; h->run_n = (int64_t)fl[1];
    h->run_dir = nullptr; h->run_nfin = 0;
    h->have_run = true;
    return 0;
}

// host-only helpers: plan lowering and finalisation from accumulator words (no GPU needed)
int vnm_agg_plan_host(int kind, int n_keys, const int* key_types, int n_funcs, const int* funcs, const int* in_types,
                      const int* in_flags, const int* in_col_ids, int* n_key_words, int* n_acc_words,
                      int* merge_kinds /* >= 40 ints */, int* n_ops, int* op_kind_col_word /* >= 3 * 48 ints */) {
    AggPlan plan;
    FuncOut outs[AGG_MAX_FUNCS];
    VNM_TRY(build_plan(kind, n_keys, key_types, n_funcs, funcs, in_types, in_flags, in_col_ids, &plan, outs));
    if (n_key_words) *n_key_words = plan.kw;
    if (n_acc_words) *n_acc_words = plan.n_words;
    if (merge_kinds) for (int w = 0; w < plan.n_words; w++) merge_kinds[w] = plan.merge[w];
    if (n_ops) *n_ops = plan.n_ops;
    if (op_kind_col_word)
        for (int o = 0; o < plan.n_ops; o++) {
            op_kind_col_word[3 * o] = plan.ops[o].kind;
            op_kind_col_word[3 * o + 1] = plan.ops[o].col;
            op_kind_col_word[3 * o + 2] = plan.ops[o].word;
        }
    return 0;
}

int vnm_agg_finalize_host(int kind, int n_keys, const int* key_types, int n_funcs, const int* funcs, const int* in_types,
                          const int* in_flags, const int* in_col_ids, int func_idx, int64_t n,
                          const uint64_t* const* acc_words, void* cells16, uint8_t* valid, int* out_kind) {
    AggPlan plan;
    FuncOut outs[AGG_MAX_FUNCS];
    VNM_TRY(build_plan(kind, n_keys, key_types, n_funcs, funcs, in_types, in_flags, in_col_ids, &plan, outs));
    if (func_idx < 0 || func_idx >= n_funcs) return set_error("vnm_agg_finalize_host: function index out of range");
    return finalize_func(outs[func_idx], n, acc_words, cells16, valid, out_kind);
}

}  // extern "C"
