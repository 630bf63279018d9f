// Projection: fused arithmetic expression evaluation for gfx950.
//
// Replaces the NumPy ufunc dispatch of vinum/core/expressions.py:13-24 as driven by
// VectorizedExpression.evaluate (vinum/core/base.py:105-125; n-ary chains left-folded :145-151) and
// ProjectOperator._kernel (vinum/core/algebra.py:52-64).  The reference materialises one full temporary
// column per AST node; here one kernel reads each input column once and writes the result once.
// Roofline: HBM, 8 B per distinct input column + 8 B per output per row.
//
// NumPy semantics restated (the arithmetic lives in third-party NumPy, pinned >= 1.19 by the reference; the image and the
// golden vectors use 2.2, i.e. NEP 50 promotion: Python literals are "weak" and take the array's type):
//   * every numeric Arrow width: int8..int64, uint8..uint64, float32, float64.  result_type of two columns: floats win
//     (float32 only survives 8 / 16-bit integers), same-signedness integers widen, unsigned + signed go to the next wider
//     signed type (uint64 + signed -> float64); integer (+,-,*,%) wrap in the RESULT width; `/` is true division (float64,
//     float32 only for float32 with 8 / 16-bit integers); a Python int literal takes the column's type (out of range for
//     it: OverflowError, as NumPy 2), a Python float literal makes integers float64 and leaves float32 alone;
//   * `%` = floor-mod with the sign of the divisor (x % 0 = 0 for ints, NaN for floats);
//   * comparisons: integers exactly (int64 vs uint64 included), mixed int / float in float64, float32 column against a
//     Python literal in float32 (the literal is rounded first), against an IN list in float64 (np.isin makes an array);
//   * a column WITH nulls reaches NumPy as float64 with NaN -- float32 stays float32 -- (record_batch.py:112-118).
// Values travel through the interpreter as 64-bit words: integers sign / zero extended, floats as float64 bits (a
// float32 value as the float64 of that float32); after every operation whose result type is narrower the word is
// brought back into that type's range (wraparound / rounding to float32), which is bit-identical to narrow arithmetic.
#include "vnm_common.hpp"

namespace vnm {

constexpr int PJ_MAX_INS = 64;
constexpr int PJ_MAX_COLS = 16;
constexpr int PJ_MAX_OUT = 16;
constexpr int PJ_STACK = 12;
constexpr int PJ_BLOCK = 256;
constexpr int PJ_NOP = -1;   // an instruction folded away by the host (unary operator on a literal)
#ifndef VNM_PJ_R
#define VNM_PJ_R 4
#endif
constexpr int PJ_R = VNM_PJ_R;                 // rows per lane per tile: one opcode decode serves PJ_R rows
constexpr int PJ_TILE = PJ_BLOCK * PJ_R;
static_assert(PJ_R % 2 == 0, "a lane owns pairs of adjacent rows");

struct PIns {
    int op;
    int arg;
    int is_f;    // result (or pushed value) is held as float64 bits
    int cvt_a;   // convert operand a (deeper) to float64 first: 1 from int64, 2 from uint64
    int cvt_b;   // ... operand b (top)
    int nar;     // bring the result back into this vnm_type's range (-1: it already is)
    int cmp;     // comparisons: 0 int64, 1 float64, 2 uint64, 3 a is uint64 / b signed, 4 a signed / b is uint64
    double imm_f;
    int64_t imm_i;
};

struct ProjArgs {
    int n_ins;
    int n_cols;
    PIns ins[PJ_MAX_INS];
    vnm_dcol cols[PJ_MAX_COLS];
    int64_t length;
    void* out[PJ_MAX_OUT];
    uint8_t out_is_mask[PJ_MAX_OUT];  // predicate outputs are byte masks
    uint8_t out_type[PJ_MAX_OUT];     // vnm_type of a value output (its width decides the store)
};

// wraparound / rounding of a 64-bit interpreter word into the range of a narrower type
__device__ __forceinline__ uint64_t pj_narrow(int t, uint64_t x) {
    switch (t) {
        case VNM_I8: return (uint64_t)(int64_t)(int8_t)x;
        case VNM_I16: return (uint64_t)(int64_t)(int16_t)x;
        case VNM_I32: return (uint64_t)(int64_t)(int32_t)x;
        case VNM_U8: return x & 0xFFULL;
        case VNM_U16: return x & 0xFFFFULL;
        case VNM_U32: return x & 0xFFFFFFFFULL;
        case VNM_F32: return (uint64_t)__double_as_longlong((double)(float)__longlong_as_double((long long)x));
        default: return x;
    }
}
__device__ __forceinline__ uint64_t np_umod(uint64_t a, uint64_t b) { return b == 0 ? 0 : a % b; }
__device__ __forceinline__ double np_fmod(double a, double b) {
    // npy_divmod: mod = fmod(a, b); if (mod) { if ((b < 0) != (mod < 0)) mod += b; } else mod = copysign(0, b)
    if (b == 0.0) return __builtin_nan("");
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0) != (m < 0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}
__device__ __forceinline__ int64_t np_imod(int64_t a, int64_t b) {
    if (b == 0) return 0;
    if (b == -1) return 0;  // avoids INT64_MIN % -1
    int64_t m = a % b;
    if (m != 0 && ((m < 0) != (b < 0))) m += b;
    return m;
}

// Postfix interpreter.  The program is uniform, so opcode fetch and dispatch are scalar; the top of the stack
// lives in registers (tos), deeper values in LDS (sized by the host to the program's real depth), and every
// decoded opcode is applied to PJ_R rows of the lane.
__global__ __launch_bounds__(PJ_BLOCK) void project_kernel(ProjArgs a) {
    extern __shared__ uint64_t stk[];  // [depth - 1][PJ_R][PJ_BLOCK]
    const int tid = threadIdx.x;
    const int64_t ntiles = (a.length + PJ_TILE - 1) / PJ_TILE;
#define STK(level, r) stk[((level) * PJ_R + (r)) * PJ_BLOCK + tid]
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // a lane owns PAIRS of adjacent rows (2 tid, 2 tid + 1) + k * 2 * PJ_BLOCK, so 8-byte columns move 16 bytes per
        // request whenever the Arrow offset is even
        const int64_t base = tile * PJ_TILE + 2 * tid;
        const bool full_tile = (tile + 1) * PJ_TILE <= a.length;
        uint64_t tos[PJ_R];
#pragma unroll
        for (int r = 0; r < PJ_R; r++) tos[r] = 0;
        int sp = 0;  // values on the stack, tos included
        for (int i = 0; i < a.n_ins; i++) {
            const PIns& in = a.ins[i];
            const int op = in.op;
            if (op == PJ_NOP) continue;
            if (op == VNM_EX_COL || op == VNM_EX_CONST_F || op == VNM_EX_CONST_I || op == VNM_EX_IS_NULL ||
                op == VNM_EX_IS_NOT_NULL) {
                // ---- push ----
                if (sp > 0) {
#pragma unroll
                    for (int r = 0; r < PJ_R; r++) STK(sp - 1, r) = tos[r];
                }
                sp++;
                if (op == VNM_EX_COL && full_tile && !a.cols[in.arg].validity && (a.cols[in.arg].offset & 1) == 0 &&
                    (a.cols[in.arg].type == VNM_F64 || (!in.is_f && type_width(a.cols[in.arg].type) == 8 && a.cols[in.arg].type != VNM_F32))) {
                    // 8-byte column pushed with its own type: raw bits, 16 bytes per request
                    const uint64_t* p = (const uint64_t*)a.cols[in.arg].values + a.cols[in.arg].offset + base;
#pragma unroll
                    for (int k = 0; k < PJ_R / 2; k++) {
                        const ulonglong2 t = *(const ulonglong2*)(p + k * (2 * PJ_BLOCK));
                        tos[2 * k] = t.x;
                        tos[2 * k + 1] = t.y;
                    }
                } else if (op == VNM_EX_COL) {
                    const vnm_dcol& c = a.cols[in.arg];
#pragma unroll
                    for (int r = 0; r < PJ_R; r++) {
                        const int64_t row = base + (r >> 1) * (2 * PJ_BLOCK) + (r & 1);
                        uint64_t v = 0;
                        if (row < a.length) {
                            if (in.is_f) {   // float column, or an integer column with NULLs (-> float64 NaN)
                                double d = col_valid(c, row) ? col_f64(c, row) : __builtin_nan("");
                                v = (uint64_t)__double_as_longlong(d);
                            } else {
                                v = (uint64_t)col_i64(c, row);   // sign / zero extended
                            }
                        }
                        tos[r] = v;
                    }
                } else if (op == VNM_EX_CONST_F) {
#pragma unroll
                    for (int r = 0; r < PJ_R; r++) tos[r] = (uint64_t)__double_as_longlong(in.imm_f);
                } else if (op == VNM_EX_CONST_I) {
#pragma unroll
                    for (int r = 0; r < PJ_R; r++) tos[r] = (uint64_t)in.imm_i;
                } else {
                    const vnm_dcol& c = a.cols[in.arg];
                    const bool want_null = op == VNM_EX_IS_NULL;
#pragma unroll
                    for (int r = 0; r < PJ_R; r++) {
                        const int64_t row = base + (r >> 1) * (2 * PJ_BLOCK) + (r & 1);
                        bool valid = row < a.length ? col_valid(c, row) : true;
                        tos[r] = (valid != want_null) ? 1ULL : 0ULL;
                    }
                }
            } else if (op == VNM_EX_NEG || op == VNM_EX_BNOT || op == VNM_EX_NOT) {
                // ---- unary ----
#pragma unroll
                for (int r = 0; r < PJ_R; r++) {
                    uint64_t x = tos[r];
                    if (op == VNM_EX_NEG)
                        x = in.is_f ? (uint64_t)__double_as_longlong(-__longlong_as_double((long long)x)) : (uint64_t)0 - x;
                    else if (op == VNM_EX_BNOT) x = ~x;
                    else x ^= 1ULL;
                    if (in.nar >= 0) x = pj_narrow(in.nar, x);
                    tos[r] = x;
                }
            } else if (op == VNM_EX_STORE) {
                // ---- pop into an output column ----
                if (a.out_is_mask[in.arg]) {
                    uint8_t* o = (uint8_t*)a.out[in.arg];
#pragma unroll
                    for (int r = 0; r < PJ_R; r++) {
                        const int64_t row = base + (r >> 1) * (2 * PJ_BLOCK) + (r & 1);
                        if (row < a.length) o[row] = (uint8_t)tos[r];
                    }
                } else if (type_width(a.out_type[in.arg]) != 8) {
                    // narrow output types: typed scalar stores (float32 values are held as float64)
                    const int ot = a.out_type[in.arg];
                    void* o = a.out[in.arg];
#pragma unroll
                    for (int r = 0; r < PJ_R; r++) {
                        const int64_t row = base + (r >> 1) * (2 * PJ_BLOCK) + (r & 1);
                        if (row >= a.length) continue;
                        if (ot == VNM_F32) ((float*)o)[row] = (float)__longlong_as_double((long long)tos[r]);
                        else if (type_width(ot) == 4) ((uint32_t*)o)[row] = (uint32_t)tos[r];
                        else if (type_width(ot) == 2) ((uint16_t*)o)[row] = (uint16_t)tos[r];
                        else ((uint8_t*)o)[row] = (uint8_t)tos[r];
                    }
                } else if (full_tile) {
                    uint64_t* o = (uint64_t*)a.out[in.arg] + base;
#pragma unroll
                    for (int k = 0; k < PJ_R / 2; k++) {
                        ulonglong2 t;
                        t.x = tos[2 * k];
                        t.y = tos[2 * k + 1];
                        *(ulonglong2*)(o + k * (2 * PJ_BLOCK)) = t;
                    }
                } else {
                    uint64_t* o = (uint64_t*)a.out[in.arg];
#pragma unroll
                    for (int r = 0; r < PJ_R; r++) {
                        const int64_t row = base + (r >> 1) * (2 * PJ_BLOCK) + (r & 1);
                        if (row < a.length) o[row] = tos[r];
                    }
                }
                sp--;
                if (sp > 0) {
#pragma unroll
                    for (int r = 0; r < PJ_R; r++) tos[r] = STK(sp - 1, r);
                }
            } else {
                // ---- binary: a = value below the top (LDS), b = top (registers) ----
                sp--;
                const bool cmp = op >= VNM_EX_EQ && op <= VNM_EX_LE;
                const bool as_f = cmp ? (in.cvt_a || in.cvt_b || in.arg) : in.is_f != 0;
#pragma unroll
                for (int r = 0; r < PJ_R; r++) {
                    const uint64_t xa = STK(sp - 1, r);
                    const uint64_t xb = tos[r];
                    uint64_t res;
                    if (op == VNM_EX_AND) res = xa & xb;
                    else if (op == VNM_EX_OR) res = xa | xb;
                    else if (as_f) {
                        double da = in.cvt_a ? (in.cvt_a == 2 ? (double)xa : (double)(int64_t)xa) : __longlong_as_double((long long)xa);
                        double db = in.cvt_b ? (in.cvt_b == 2 ? (double)xb : (double)(int64_t)xb) : __longlong_as_double((long long)xb);
                        if (cmp) res = cmp_apply<double>(op - VNM_EX_EQ, da, db) ? 1ULL : 0ULL;  // same order as enum vnm_cmp_op
                        else {
                            double d;
                            switch (op) {
                                case VNM_EX_ADD: d = da + db; break;
                                case VNM_EX_SUB: d = da - db; break;
                                case VNM_EX_MUL: d = da * db; break;
                                case VNM_EX_DIV: d = da / db; break;
                                default: d = np_fmod(da, db); break;
                            }
                            res = (uint64_t)__double_as_longlong(d);
                        }
                    } else if (cmp) {
                        bool t;
                        switch (in.cmp) {
                            case 2: t = cmp_apply<uint64_t>(op - VNM_EX_EQ, xa, xb); break;
                            case 3:  // a uint64, b signed: a negative b is below every a
                                t = (int64_t)xb < 0 ? (op == VNM_EX_NE || op == VNM_EX_GT || op == VNM_EX_GE) : cmp_apply<uint64_t>(op - VNM_EX_EQ, xa, xb);
                                break;
                            case 4:
                                t = (int64_t)xa < 0 ? (op == VNM_EX_NE || op == VNM_EX_LT || op == VNM_EX_LE) : cmp_apply<uint64_t>(op - VNM_EX_EQ, xa, xb);
                                break;
                            default: t = cmp_apply<int64_t>(op - VNM_EX_EQ, (int64_t)xa, (int64_t)xb); break;
                        }
                        res = t ? 1ULL : 0ULL;
                    } else {
                        switch (op) {
                            case VNM_EX_ADD: res = xa + xb; break;
                            case VNM_EX_SUB: res = xa - xb; break;
                            case VNM_EX_MUL: res = xa * xb; break;
                            case VNM_EX_MOD: res = in.cmp == 2 ? np_umod(xa, xb) : (uint64_t)np_imod((int64_t)xa, (int64_t)xb); break;
                            case VNM_EX_BAND: res = xa & xb; break;
                            case VNM_EX_BOR: res = xa | xb; break;
                            default: res = xa ^ xb; break;
                        }
                    }
                    if (in.nar >= 0) res = pj_narrow(in.nar, res);
                    tos[r] = res;
                }
            }
        }
    }
#undef STK
}

}  // namespace vnm

using namespace vnm;

// np.result_type of two column types (NumPy 2 / NEP 50; probed table in profiles/numpy_promotion_r02.txt)
static int pj_promote(int a, int b) {
    if (a == b) return a;
    const bool fa = type_is_float(a), fb = type_is_float(b);
    if (fa && fb) return VNM_F64;
    if (fa || fb) {
        const int f = fa ? a : b, i = fa ? b : a;
        if (f == VNM_F64) return VNM_F64;
        return type_width(i) <= 2 ? VNM_F32 : VNM_F64;
    }
    const bool ua = type_is_unsigned(a), ub = type_is_unsigned(b);
    const int wa = type_width(a), wb = type_width(b);
    if (ua == ub) return wa >= wb ? a : b;
    const int u = ua ? a : b, sg = ua ? b : a;
    if (type_width(u) < type_width(sg)) return sg;
    switch (u) {
        case VNM_U8: return VNM_I16;
        case VNM_U16: return VNM_I32;
        case VNM_U32: return VNM_I64;
        default: return VNM_F64;   // uint64 with a signed type
    }
}
static const char* pj_type_name(int t) {
    static const char* n[] = {"int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float32", "float64"};
    return t >= 0 && t < 10 ? n[t] : "?";
}
static bool pj_int_fits(int t, int64_t v) {
    switch (t) {
        case VNM_I8: return v >= -128 && v <= 127;
        case VNM_I16: return v >= -32768 && v <= 32767;
        case VNM_I32: return v >= INT32_MIN && v <= INT32_MAX;
        case VNM_U8: return v >= 0 && v <= 255;
        case VNM_U16: return v >= 0 && v <= 65535;
        case VNM_U32: return v >= 0 && v <= (int64_t)UINT32_MAX;
        case VNM_U64: return v >= 0;
        default: return true;
    }
}

// Abstract interpretation of the program (a type per stack slot: the ten numeric types with NumPy's promotion, "weak"
// Python literals, bool masks), launch.  `single`: the program is one expression without a STORE; out index 0 is implied.
static int project_impl(int n_ins, const vnm_expr_ins* program, int n_cols, const vnm_dcol* cols, int64_t length,
                        int n_out, void** out_values, int* out_types, bool single, void* stream) {
    VNM_TRY(ensure_init());
    if (n_ins <= 0 || n_ins + (single ? 1 : 0) > PJ_MAX_INS)
        return set_error("vnm_project: program must have 1..%d instructions", PJ_MAX_INS - (single ? 1 : 0));
    if (n_cols < 0 || n_cols > PJ_MAX_COLS) return set_error("vnm_project: at most %d input columns", PJ_MAX_COLS);
    if (n_out < 1 || n_out > PJ_MAX_OUT) return set_error("vnm_project: 1..%d outputs per call", PJ_MAX_OUT);
    ProjArgs a{};
    a.n_cols = n_cols;
    a.length = length;
    for (int c = 0; c < n_cols; c++) {
        if (cols[c].type < VNM_I8 || cols[c].type > VNM_F64) return set_error("vnm_project: column %d: unsupported type %d", c, cols[c].type);
        if (cols[c].length != length) return set_error("Select expressions have unequal sizes. This is not permitted.");
        a.cols[c] = cols[c];
    }
    enum { T_B = 10, T_WI = 11, T_WF = 12 };   // 0..9 = vnm_type; bool mask; weak Python int / float literal
    int ty[PJ_STACK];
    int lit[PJ_STACK];   // the instruction that pushed a weak literal still unchanged on the stack (-1: none)
    int sp = 0, depth = 1;
    bool stored[PJ_MAX_OUT] = {};
    const int total = n_ins + (single ? 1 : 0);
    auto is_num = [&](int t) { return t != T_B; };
    auto held_f = [&](int t) { return t == T_WF || (t < 10 && type_is_float(t)); };       // held as float64 bits
    auto cvt_of = [&](int t) { return held_f(t) ? 0 : (t == VNM_U64 ? 2 : 1); };           // how to turn the held word into a double
    // a weak literal meets a float32 column: NumPy casts the scalar to float32 first
    auto round_lit_f32 = [&](int slot) {
        if (lit[slot] < 0) return;
        PIns& p = a.ins[lit[slot]];
        if (p.op == VNM_EX_CONST_I) { p.op = VNM_EX_CONST_F; p.imm_f = (double)(float)p.imm_i; p.is_f = 1; }
        else p.imm_f = (double)(float)p.imm_f;
    };
    for (int i = 0; i < total; i++) {
        vnm_expr_ins in;
        if (i < n_ins) in = program[i];
        else { in = vnm_expr_ins{}; in.op = VNM_EX_STORE; in.arg = 0; }
        PIns& o = a.ins[i];
        o.op = in.op;
        o.arg = in.arg;
        o.imm_f = in.imm_f;
        o.imm_i = in.imm_i;
        o.cvt_a = o.cvt_b = 0;
        o.is_f = 0;
        o.nar = -1;
        o.cmp = 0;
        switch (in.op) {
            case VNM_EX_COL: {
                if (in.arg < 0 || in.arg >= n_cols) return set_error("vnm_project: column index %d out of range", in.arg);
                if (sp >= PJ_STACK) return set_error("vnm_project: expression too deep");
                const int ct = cols[in.arg].type;
                int t = ct;
                if (cols[in.arg].validity != nullptr && !type_is_float(ct)) t = VNM_F64;   // NULL -> NaN needs a float (record_batch.py:112-118)
                o.is_f = type_is_float(t);
                lit[sp] = -1;
                ty[sp++] = t;
                break;
            }
            case VNM_EX_CONST_F:
            case VNM_EX_CONST_I:
                if (sp >= PJ_STACK) return set_error("vnm_project: expression too deep");
                o.is_f = in.op == VNM_EX_CONST_F;
                // arg = 1: a STRONG constant (an element of the array np.isin builds from an IN list): int64 / float64
                lit[sp] = in.arg == 1 ? -1 : i;
                ty[sp++] = in.arg == 1 ? (o.is_f ? VNM_F64 : VNM_I64) : (o.is_f ? T_WF : T_WI);
                o.arg = 0;
                break;
            case VNM_EX_IS_NULL:
            case VNM_EX_IS_NOT_NULL:
                if (in.arg < 0 || in.arg >= n_cols) return set_error("vnm_project: column index %d out of range", in.arg);
                if (sp >= PJ_STACK) return set_error("vnm_project: expression too deep");
                lit[sp] = -1;
                ty[sp++] = T_B;
                break;
            case VNM_EX_NEG:
            case VNM_EX_BNOT: {
                if (sp < 1) return set_error("vnm_project: malformed program (stack underflow)");
                const int t = ty[sp - 1];
                if (t == T_B) return set_error("vnm_project: arithmetic on a boolean mask is not supported");
                if (in.op == VNM_EX_BNOT && held_f(t)) return set_error("ufunc 'invert' not supported for float inputs");
                if (t >= T_WI && lit[sp - 1] >= 0) {
                    // a Python literal: folded here.  `~5` stays a (weak) Python int; np.negative(5) RETURNS a NumPy scalar,
                    // np.int64 / np.float64, which is a strong type from then on
                    PIns& c = a.ins[lit[sp - 1]];
                    if (in.op == VNM_EX_BNOT) c.imm_i = ~c.imm_i;
                    else if (c.op == VNM_EX_CONST_I) { c.imm_i = (int64_t)(0 - (uint64_t)c.imm_i); ty[sp - 1] = VNM_I64; lit[sp - 1] = -1; }
                    else { c.imm_f = -c.imm_f; ty[sp - 1] = VNM_F64; lit[sp - 1] = -1; }
                    o.op = PJ_NOP;
                    break;
                }
                o.is_f = held_f(t);
                if (t < 10 && !type_is_float(t) && type_width(t) < 8) o.nar = t;
                lit[sp - 1] = -1;
                break;
            }
            case VNM_EX_NOT:
                if (sp < 1) return set_error("vnm_project: malformed program (stack underflow)");
                if (ty[sp - 1] != T_B) return set_error("NOT expects a boolean operand");
                break;
            case VNM_EX_AND:
            case VNM_EX_OR:
                if (sp < 2) return set_error("vnm_project: malformed program (stack underflow)");
                if (ty[sp - 1] != T_B || ty[sp - 2] != T_B) return set_error("AND / OR expect boolean operands");
                sp--;
                break;
            case VNM_EX_EQ: case VNM_EX_NE: case VNM_EX_GT: case VNM_EX_GE: case VNM_EX_LT: case VNM_EX_LE: {
                if (sp < 2) return set_error("vnm_project: malformed program (stack underflow)");
                const int tb = ty[sp - 1], ta = ty[sp - 2];
                if (!is_num(ta) || !is_num(tb)) return set_error("vnm_project: comparing boolean masks is not supported");
                // a float32 column against a Python literal compares in float32
                if (ta == VNM_F32 && (tb == T_WI || tb == T_WF)) round_lit_f32(sp - 1);
                if (tb == VNM_F32 && (ta == T_WI || ta == T_WF)) round_lit_f32(sp - 2);
                const bool anyf = held_f(ta) || held_f(tb);
                if (anyf) {  // NumPy compares in float64 as soon as one side is float (float32 values are exact float64s)
                    o.cmp = 1;
                    // the literal may just have become a float constant: ask the instruction, not the type
                    const bool a_f = held_f(ta) || (lit[sp - 2] >= 0 && a.ins[lit[sp - 2]].op == VNM_EX_CONST_F);
                    const bool b_f = held_f(tb) || (lit[sp - 1] >= 0 && a.ins[lit[sp - 1]].op == VNM_EX_CONST_F);
                    o.cvt_a = a_f ? 0 : (ta == VNM_U64 ? 2 : 1);
                    o.cvt_b = b_f ? 0 : (tb == VNM_U64 ? 2 : 1);
                } else {     // integers compare exactly, int64 against uint64 included
                    const bool ua = ta == VNM_U64, ub = tb == VNM_U64;
                    const bool sa = ta == T_WI || (ta < 10 && !type_is_unsigned(ta)), sb = tb == T_WI || (tb < 10 && !type_is_unsigned(tb));
                    o.cmp = (ua && ub) ? 2 : (ua && sb) ? 3 : (sa && ub) ? 4 : (ua || ub) ? 2 : 0;
                }
                o.arg = anyf ? 1 : 0;
                sp--;
                ty[sp - 1] = T_B;
                lit[sp - 1] = -1;
                break;
            }
            case VNM_EX_ADD: case VNM_EX_SUB: case VNM_EX_MUL: case VNM_EX_DIV: case VNM_EX_MOD:
            case VNM_EX_BAND: case VNM_EX_BOR: case VNM_EX_BXOR: {
                if (sp < 2) return set_error("vnm_project: malformed program (stack underflow)");
                const int tb = ty[sp - 1], ta = ty[sp - 2];
                if (!is_num(ta) || !is_num(tb)) return set_error("vnm_project: arithmetic on a boolean mask is not supported");
                const bool bitop = in.op >= VNM_EX_BAND;
                const bool wa = ta >= T_WI, wb = tb >= T_WI;
                int rt;
                if (wa && wb) rt = (ta == T_WF || tb == T_WF) ? VNM_F64 : VNM_I64;       // Python scalars: default int64 / float64
                else if (wa || wb) {
                    const int w = wa ? ta : tb, t = wa ? tb : ta, wslot = wa ? sp - 2 : sp - 1;
                    if (w == T_WI) {
                        rt = t;                                                               // the literal takes the column's type
                        // true division runs in float64 whatever the integer width: NumPy converts the literal to double, no range check
                        if (in.op != VNM_EX_DIV && !type_is_float(t) && lit[wslot] >= 0 && !pj_int_fits(t, a.ins[lit[wslot]].imm_i))
                            return set_error("OverflowError: Python integer %lld out of bounds for %s", (long long)a.ins[lit[wslot]].imm_i, pj_type_name(t));
                        if (t == VNM_F32) round_lit_f32(wslot);
                    } else {
                        rt = t == VNM_F32 ? VNM_F32 : VNM_F64;
                        if (t == VNM_F32) round_lit_f32(wslot);
                    }
                } else rt = pj_promote(ta, tb);
                if (bitop && (type_is_float(rt) || held_f(ta) || held_f(tb))) return set_error("ufunc 'bitwise' not supported for float inputs");
                if (in.op == VNM_EX_DIV && !type_is_float(rt)) rt = VNM_F64;
                const bool rf = type_is_float(rt);
                o.is_f = rf;
                if (rf) {
                    const bool a_f = held_f(ta) || (lit[sp - 2] >= 0 && a.ins[lit[sp - 2]].op == VNM_EX_CONST_F);
                    const bool b_f = held_f(tb) || (lit[sp - 1] >= 0 && a.ins[lit[sp - 1]].op == VNM_EX_CONST_F);
                    o.cvt_a = a_f ? 0 : (ta == VNM_U64 ? 2 : 1);
                    o.cvt_b = b_f ? 0 : (tb == VNM_U64 ? 2 : 1);
                }
                if (rt == VNM_F32 || (!rf && type_width(rt) < 8)) o.nar = rt;
                if (rt == VNM_U64) o.cmp = 2;   // unsigned modulo
                sp--;
                ty[sp - 1] = rt;
                lit[sp - 1] = -1;
                break;
            }
            case VNM_EX_STORE: {
                if (single && i < n_ins) return set_error("vnm_project: VNM_EX_STORE belongs to vnm_project_multi programs");
                if (sp < 1) return set_error("vnm_project: malformed program (stack underflow)");
                if (in.arg < 0 || in.arg >= n_out) return set_error("vnm_project: output index %d out of range", in.arg);
                if (stored[in.arg]) return set_error("vnm_project: output %d stored twice", in.arg);
                stored[in.arg] = true;
                int t = ty[sp - 1];
                if (t == T_WI) t = VNM_I64;            // np.repeat(5, n) -> int64, np.repeat(5.0, n) -> float64 (algebra.py:77-87)
                if (t == T_WF) t = VNM_F64;
                a.out[in.arg] = out_values[in.arg];
                a.out_is_mask[in.arg] = t == T_B;
                a.out_type[in.arg] = t == T_B ? VNM_U8 : t;
                if (out_types) out_types[in.arg] = t == T_B ? VNM_MASK_U8 : t;
                sp--;
                break;
            }
            default: return set_error("vnm_project: unknown opcode %d", in.op);
        }
        if (sp > depth) depth = sp;
    }
    if (sp != 0) return set_error("vnm_project: malformed program (final stack depth %d)", sp + (single ? 1 : 0));
    for (int k = 0; k < n_out; k++)
        if (!stored[k]) return set_error("vnm_project: output %d is never stored", k);
    a.n_ins = total;
    if (length <= 0) return 0;
    const size_t lds = (size_t)(depth > 1 ? depth - 1 : 1) * PJ_R * PJ_BLOCK * 8;
    // ONE workgroup per tile: a workgroup loads, computes and stores with nothing of its own to overlap, so it is the
    // dispatcher that keeps the memory system busy (three expressions over 1e9 rows: 8 workgroups per CU looping over tiles
    // 11.15 ms, 32 per CU 10.2, 128 per CU 9.7, one per tile 9.27 = 5.2 TB/s)
    int64_t grid64 = getenv("VNM_PJ_GRID") ? (int64_t)device_info().num_cus * atoi(getenv("VNM_PJ_GRID")) : (int64_t)1 << 30;
    int grid = (int)grid64;
    int64_t need = (length + PJ_TILE - 1) / PJ_TILE;
    if (grid > need) grid = (int)need;
    if (lds > 64 * 1024)   // depth >= 10: beyond the default dynamic LDS limit
        VNM_HIP(hipFuncSetAttribute((const void*)project_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        KernelTimer timer("project_kernel", as_stream(stream));
        project_kernel<<<grid, PJ_BLOCK, lds, as_stream(stream)>>>(a);
    }
    VNM_HIP(hipGetLastError());
    return 0;
}

extern "C" {

int vnm_project(int n_ins, const vnm_expr_ins* program, int n_cols, const vnm_dcol* cols, int64_t length,
                void* out_values, int* out_type, void* stream) {
    void* outs[1] = {out_values};
    return project_impl(n_ins, program, n_cols, cols, length, 1, outs, out_type, true, stream);
}

int vnm_project_multi(int n_ins, const vnm_expr_ins* program, int n_cols, const vnm_dcol* cols, int64_t length,
                      int n_out, void** out_values, int* out_types, void* stream) {
    if (!out_values) return set_error("vnm_project_multi: null argument");
    return project_impl(n_ins, program, n_cols, cols, length, n_out, out_values, out_types, false, stream);
}

}  // extern "C"
