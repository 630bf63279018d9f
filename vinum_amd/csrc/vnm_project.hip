// Projection: fused arithmetic expression evaluation for gfx950.
//
// Replaces the NumPy ufunc dispatch of vinum/core/expressions.py:13-24 as driven by
// VectorizedExpression.evaluate (vinum/core/base.py:105-125; n-ary chains left-folded :145-151) and
// ProjectOperator._kernel (vinum/core/algebra.py:52-64).  The reference materialises one full temporary
// column per AST node; here one kernel reads each input column once and writes the result once.
// Roofline: HBM, 8 B per distinct input column + 8 B per output per row.
//
// NumPy semantics restated: int64 (+,-,*) int64 -> int64 with wraparound; `/` -> float64 true division;
// int (op) float -> float64; `%` = floor-mod with the sign of the divisor (x % 0 = 0 for ints, NaN for
// floats); a column WITH nulls reaches NumPy as float64 with NaN (vinum/arrow/record_batch.py:112-118).
#include "vnm_common.hpp"

namespace vnm {

constexpr int PJ_MAX_INS = 64;
constexpr int PJ_MAX_COLS = 16;
constexpr int PJ_MAX_OUT = 16;
constexpr int PJ_STACK = 12;
constexpr int PJ_BLOCK = 256;
#ifndef VNM_PJ_R
#define VNM_PJ_R 4
#endif
constexpr int PJ_R = VNM_PJ_R;                 // rows per lane per tile: one opcode decode serves PJ_R rows
constexpr int PJ_TILE = PJ_BLOCK * PJ_R;
static_assert(PJ_R % 2 == 0, "a lane owns pairs of adjacent rows");

struct PIns {
    int op;
    int arg;
    int is_f;    // result (or pushed value) is float64
    int cvt_a;   // convert operand a (deeper) int64 -> float64 first
    int cvt_b;   // convert operand b (top) int64 -> float64 first
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
};

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
            if (op == VNM_EX_COL || op == VNM_EX_CONST_F || op == VNM_EX_CONST_I || op == VNM_EX_IS_NULL ||
                op == VNM_EX_IS_NOT_NULL) {
                // ---- push ----
                if (sp > 0) {
#pragma unroll
                    for (int r = 0; r < PJ_R; r++) STK(sp - 1, r) = tos[r];
                }
                sp++;
                if (op == VNM_EX_COL && full_tile && !a.cols[in.arg].validity && (a.cols[in.arg].offset & 1) == 0 &&
                    (a.cols[in.arg].type == VNM_F64 || (!in.is_f && type_width(a.cols[in.arg].type) == 8))) {
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
                            if (in.is_f) {
                                double d = col_valid(c, row) ? col_f64(c, row) : __builtin_nan("");
                                v = (uint64_t)__double_as_longlong(d);
                            } else {
                                v = (uint64_t)col_i64(c, row);
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
                        double da = in.cvt_a ? (double)(int64_t)xa : __longlong_as_double((long long)xa);
                        double db = in.cvt_b ? (double)(int64_t)xb : __longlong_as_double((long long)xb);
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
                        res = cmp_apply<int64_t>(op - VNM_EX_EQ, (int64_t)xa, (int64_t)xb) ? 1ULL : 0ULL;
                    } else {
                        switch (op) {
                            case VNM_EX_ADD: res = xa + xb; break;
                            case VNM_EX_SUB: res = xa - xb; break;
                            case VNM_EX_MUL: res = xa * xb; break;
                            case VNM_EX_MOD: res = (uint64_t)np_imod((int64_t)xa, (int64_t)xb); break;
                            case VNM_EX_BAND: res = xa & xb; break;
                            case VNM_EX_BOR: res = xa | xb; break;
                            default: res = xa ^ xb; break;
                        }
                    }
                    tos[r] = res;
                }
            }
        }
    }
#undef STK
}

}  // namespace vnm

using namespace vnm;

// Abstract interpretation of the program (types per stack slot: NumPy result_type over {int64, float64}; bool
// masks), launch.  `single`: the program is one expression without a STORE; out index 0 is implied.
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
        if (cols[c].type != VNM_I64 && cols[c].type != VNM_F64)
            return set_error("vnm_project: column %d: only int64 / float64 columns are supported (NumPy's narrower "
                             "promotions are not restated yet)", c);
        if (cols[c].length != length) return set_error("Select expressions have unequal sizes. This is not permitted.");
        a.cols[c] = cols[c];
    }
    enum { T_I = 0, T_F = 1, T_B = 2 };
    int ty[PJ_STACK];
    int sp = 0, depth = 1;
    bool stored[PJ_MAX_OUT] = {};
    const int total = n_ins + (single ? 1 : 0);
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
        switch (in.op) {
            case VNM_EX_COL:
                if (in.arg < 0 || in.arg >= n_cols) return set_error("vnm_project: column index %d out of range", in.arg);
                if (sp >= PJ_STACK) return set_error("vnm_project: expression too deep");
                o.is_f = (cols[in.arg].type == VNM_F64) || cols[in.arg].validity != nullptr;
                ty[sp++] = o.is_f ? T_F : T_I;
                break;
            case VNM_EX_CONST_F:
            case VNM_EX_CONST_I:
                if (sp >= PJ_STACK) return set_error("vnm_project: expression too deep");
                o.is_f = in.op == VNM_EX_CONST_F;
                ty[sp++] = o.is_f ? T_F : T_I;
                break;
            case VNM_EX_IS_NULL:
            case VNM_EX_IS_NOT_NULL:
                if (in.arg < 0 || in.arg >= n_cols) return set_error("vnm_project: column index %d out of range", in.arg);
                if (sp >= PJ_STACK) return set_error("vnm_project: expression too deep");
                ty[sp++] = T_B;
                break;
            case VNM_EX_NEG:
            case VNM_EX_BNOT:
                if (sp < 1) return set_error("vnm_project: malformed program (stack underflow)");
                if (ty[sp - 1] == T_B) return set_error("vnm_project: arithmetic on a boolean mask is not supported");
                if (in.op == VNM_EX_BNOT && ty[sp - 1] == T_F) return set_error("ufunc 'invert' not supported for float inputs");
                o.is_f = ty[sp - 1] == T_F;
                break;
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
                int tb = ty[sp - 1], ta = ty[sp - 2];
                if (ta == T_B || tb == T_B) return set_error("vnm_project: comparing boolean masks is not supported");
                bool anyf = ta == T_F || tb == T_F;  // NumPy compares in float64 as soon as one side is float
                o.cvt_a = anyf && ta == T_I;
                o.cvt_b = anyf && tb == T_I;
                o.arg = anyf ? 1 : 0;
                sp--;
                ty[sp - 1] = T_B;
                break;
            }
            case VNM_EX_ADD: case VNM_EX_SUB: case VNM_EX_MUL: case VNM_EX_DIV: case VNM_EX_MOD:
            case VNM_EX_BAND: case VNM_EX_BOR: case VNM_EX_BXOR: {
                if (sp < 2) return set_error("vnm_project: malformed program (stack underflow)");
                if (ty[sp - 1] == T_B || ty[sp - 2] == T_B) return set_error("vnm_project: arithmetic on a boolean mask is not supported");
                bool fb = ty[sp - 1] == T_F, fa = ty[sp - 2] == T_F;
                bool bitop = in.op >= VNM_EX_BAND;
                if (bitop && (fa || fb)) return set_error("ufunc 'bitwise' not supported for float inputs");
                bool rf = fa || fb || in.op == VNM_EX_DIV;
                o.is_f = rf;
                o.cvt_a = rf && !fa;
                o.cvt_b = rf && !fb;
                sp--;
                ty[sp - 1] = rf ? T_F : T_I;
                break;
            }
            case VNM_EX_STORE: {
                if (single && i < n_ins) return set_error("vnm_project: VNM_EX_STORE belongs to vnm_project_multi programs");
                if (sp < 1) return set_error("vnm_project: malformed program (stack underflow)");
                if (in.arg < 0 || in.arg >= n_out) return set_error("vnm_project: output index %d out of range", in.arg);
                if (stored[in.arg]) return set_error("vnm_project: output %d stored twice", in.arg);
                stored[in.arg] = true;
                const int t = ty[sp - 1];
                a.out[in.arg] = out_values[in.arg];
                a.out_is_mask[in.arg] = t == T_B;
                if (out_types) out_types[in.arg] = t == T_B ? VNM_MASK_U8 : (t == T_F ? VNM_F64 : VNM_I64);
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
    int grid = device_info().num_cus * 8;
    int64_t need = (length + PJ_TILE - 1) / PJ_TILE;
    if (grid > need) grid = (int)need;
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
