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
constexpr int PJ_STACK = 12;
constexpr int PJ_BLOCK = 256;

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
    uint64_t* out;
    uint8_t* out_mask;  // predicate programs write a byte mask instead
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

__global__ __launch_bounds__(PJ_BLOCK) void project_kernel(ProjArgs a) {
    __shared__ uint64_t stack[PJ_STACK][PJ_BLOCK];
    const int tid = threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * PJ_BLOCK;
    for (int64_t row = (int64_t)blockIdx.x * PJ_BLOCK + tid; row < a.length; row += stride) {
        int sp = 0;
        for (int i = 0; i < a.n_ins; i++) {
            const PIns& in = a.ins[i];
            switch (in.op) {
                case VNM_EX_COL: {
                    const vnm_dcol& c = a.cols[in.arg];
                    uint64_t v;
                    if (in.is_f) {
                        double d = col_valid(c, row) ? col_f64(c, row) : __builtin_nan("");
                        v = (uint64_t)__double_as_longlong(d);
                    } else {
                        v = (uint64_t)col_i64(c, row);
                    }
                    stack[sp++][tid] = v;
                    break;
                }
                case VNM_EX_CONST_F: stack[sp++][tid] = (uint64_t)__double_as_longlong(in.imm_f); break;
                case VNM_EX_CONST_I: stack[sp++][tid] = (uint64_t)in.imm_i; break;
                case VNM_EX_NEG: {
                    uint64_t x = stack[sp - 1][tid];
                    stack[sp - 1][tid] = in.is_f ? (uint64_t)__double_as_longlong(-__longlong_as_double((long long)x)) : (uint64_t)0 - x;
                    break;
                }
                case VNM_EX_BNOT: stack[sp - 1][tid] = ~stack[sp - 1][tid]; break;
                case VNM_EX_NOT: stack[sp - 1][tid] = stack[sp - 1][tid] ^ 1ULL; break;
                case VNM_EX_IS_NULL: stack[sp++][tid] = col_valid(a.cols[in.arg], row) ? 0ULL : 1ULL; break;
                case VNM_EX_IS_NOT_NULL: stack[sp++][tid] = col_valid(a.cols[in.arg], row) ? 1ULL : 0ULL; break;
                case VNM_EX_AND: { uint64_t xb = stack[--sp][tid]; stack[sp - 1][tid] &= xb; break; }
                case VNM_EX_OR: { uint64_t xb = stack[--sp][tid]; stack[sp - 1][tid] |= xb; break; }
                case VNM_EX_EQ: case VNM_EX_NE: case VNM_EX_GT: case VNM_EX_GE: case VNM_EX_LT: case VNM_EX_LE: {
                    uint64_t xb = stack[--sp][tid];
                    uint64_t xa = stack[sp - 1][tid];
                    const int cop = in.op - VNM_EX_EQ;  // same order as enum vnm_cmp_op
                    bool r;
                    if (in.cvt_a || in.cvt_b || in.arg) {  // float comparison (arg = 1: both operands already float)
                        double da = in.cvt_a ? (double)(int64_t)xa : __longlong_as_double((long long)xa);
                        double db = in.cvt_b ? (double)(int64_t)xb : __longlong_as_double((long long)xb);
                        r = cmp_apply<double>(cop, da, db);
                    } else {
                        r = cmp_apply<int64_t>(cop, (int64_t)xa, (int64_t)xb);
                    }
                    stack[sp - 1][tid] = r ? 1ULL : 0ULL;
                    break;
                }
                default: {
                    uint64_t xb = stack[--sp][tid];
                    uint64_t xa = stack[sp - 1][tid];
                    uint64_t r;
                    if (in.is_f) {
                        double da = in.cvt_a ? (double)(int64_t)xa : __longlong_as_double((long long)xa);
                        double db = in.cvt_b ? (double)(int64_t)xb : __longlong_as_double((long long)xb);
                        double d;
                        switch (in.op) {
                            case VNM_EX_ADD: d = da + db; break;
                            case VNM_EX_SUB: d = da - db; break;
                            case VNM_EX_MUL: d = da * db; break;
                            case VNM_EX_DIV: d = da / db; break;
                            default: d = np_fmod(da, db); break;
                        }
                        r = (uint64_t)__double_as_longlong(d);
                    } else {
                        switch (in.op) {
                            case VNM_EX_ADD: r = xa + xb; break;
                            case VNM_EX_SUB: r = xa - xb; break;
                            case VNM_EX_MUL: r = xa * xb; break;
                            case VNM_EX_MOD: r = (uint64_t)np_imod((int64_t)xa, (int64_t)xb); break;
                            case VNM_EX_BAND: r = xa & xb; break;
                            case VNM_EX_BOR: r = xa | xb; break;
                            default: r = xa ^ xb; break;
                        }
                    }
                    stack[sp - 1][tid] = r;
                    break;
                }
            }
        }
        if (a.out_mask) a.out_mask[row] = (uint8_t)stack[0][tid];
        else a.out[row] = stack[0][tid];
    }
}

}  // namespace vnm

using namespace vnm;

extern "C" {

int vnm_project(int n_ins, const vnm_expr_ins* program, int n_cols, const vnm_dcol* cols, int64_t length,
                void* out_values, int* out_type, void* stream) {
    VNM_TRY(ensure_init());
    if (n_ins <= 0 || n_ins > PJ_MAX_INS) return set_error("vnm_project: program must have 1..%d instructions", PJ_MAX_INS);
    if (n_cols < 0 || n_cols > PJ_MAX_COLS) return set_error("vnm_project: at most %d input columns", PJ_MAX_COLS);
    ProjArgs a{};
    a.n_ins = n_ins;
    a.n_cols = n_cols;
    a.length = length;
    a.out = (uint64_t*)out_values;
    for (int c = 0; c < n_cols; c++) {
        if (cols[c].type != VNM_I64 && cols[c].type != VNM_F64)
            return set_error("vnm_project: column %d: only int64 / float64 columns are supported (NumPy's narrower "
                             "promotions are not restated yet)", c);
        if (cols[c].length != length) return set_error("Select expressions have unequal sizes. This is not permitted.");
        a.cols[c] = cols[c];
    }
    // abstract interpretation: type of every stack slot (NumPy result_type over {int64, float64}; bool masks)
    enum { T_I = 0, T_F = 1, T_B = 2 };
    int ty[PJ_STACK];
    int sp = 0;
    for (int i = 0; i < n_ins; i++) {
        const vnm_expr_ins& in = program[i];
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
            default: return set_error("vnm_project: unknown opcode %d", in.op);
        }
    }
    if (sp != 1) return set_error("vnm_project: malformed program (final stack depth %d)", sp);
    if (out_type) *out_type = ty[0] == T_B ? VNM_MASK_U8 : (ty[0] == T_F ? VNM_F64 : VNM_I64);
    if (ty[0] == T_B) { a.out_mask = (uint8_t*)out_values; a.out = nullptr; }
    if (length <= 0) return 0;
    int grid = device_info().num_cus * 8;
    int64_t need = (length + PJ_BLOCK - 1) / PJ_BLOCK;
    if (grid > need) grid = (int)need;
    {
        KernelTimer timer("project_kernel", as_stream(stream));
        project_kernel<<<grid, PJ_BLOCK, 0, as_stream(stream)>>>(a);
    }
    VNM_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
