// Shared host/device helpers of libvinum_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vinum_hip.h"

namespace vnm {

// ---------------------------------------------------------------------------------------------
// error handling: status codes across the ABI, message in a thread-local string
// ---------------------------------------------------------------------------------------------
std::string& last_error();
int set_error(const char* fmt, ...);

#define VNM_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return ::vnm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define VNM_TRY(expr)            \
    do {                         \
        int _rc = (expr);        \
        if (_rc != 0) return _rc; \
    } while (0)

// ---------------------------------------------------------------------------------------------
// device properties / launch geometry
// ---------------------------------------------------------------------------------------------
struct DeviceInfo {
    int device = -1;
    int num_cus = 256;
    bool ready = false;
};
DeviceInfo& device_info();
int ensure_init();

// Caching device allocator: operator scratch (hash tables, look-back status, dense results) is
// re-used across calls so steady-state batches never hit hipMalloc/hipFree.
// several host ranges end to end into one device range, through the pinned staging ring (vnm_runtime.cpp)
int stage_chunks(void* dst, const void* const* srcs, const size_t* sizes, size_t n_chunks, hipStream_t stream);
void* pool_alloc(size_t bytes);
void pool_free(void* p);
void pool_free_after(void* p, hipStream_t stream);   // reusable only once the work enqueued on `stream` so far has completed
size_t pool_trim();  // releases the cached blocks, returns their bytes
// Pool blocks owned by a scope: every early return (VNM_HIP / VNM_TRY included) gives them back.  take() = pool_alloc
// registered with the scope; keep(p) = ownership moves elsewhere (a handle, the caller); done(p) = free it now.
struct PoolScope {
    static constexpr int MAX = 64;     // (vnm_csv_parse_block_ex: up to 16 string columns x 3 scratch blocks + 3; 24 was too few and said nothing)
    void* blocks[MAX];
    int n = 0;
    PoolScope() = default;
    PoolScope(const PoolScope&) = delete;
    PoolScope& operator=(const PoolScope&) = delete;
    void* take(size_t bytes) {
        void* p = pool_alloc(bytes);
        if (p) {
            if (n < MAX) blocks[n++] = p;
            else { pool_free(p); p = nullptr; set_error("internal error: more than %d scratch blocks in one scope", MAX); }
        }
        return p;
    }
    void keep(void* p) {
        for (int i = 0; i < n; i++)
            if (blocks[i] == p) { blocks[i] = blocks[--n]; return; }
    }
    void done(void* p) {
        if (!p) return;
        keep(p);
        pool_free(p);
    }
    ~PoolScope() {
        for (int i = 0; i < n; i++) pool_free(blocks[i]);
    }
};
// frees *slot (whatever it points to by then) when the scope ends
template <class T>
struct PoolSlotGuard {
    T** slot;
    explicit PoolSlotGuard(T** s) : slot(s) {}
    PoolSlotGuard(const PoolSlotGuard&) = delete;
    PoolSlotGuard& operator=(const PoolSlotGuard&) = delete;
    ~PoolSlotGuard() { if (*slot) pool_free((void*)*slot); }
};

// Optional kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
// KernelTimer brackets the dominant kernel of an operator call; elapsed time is resolved lazily.
struct KernelTimer {
    explicit KernelTimer(const char* name, hipStream_t s);
    ~KernelTimer();
    bool on;
    int slot;
    hipStream_t stream;
};

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Which ROUTE an operator call took and why (round 5): every place that commits a batch to a path of DESIGN.md section 4 leaves a note
// -- a route name (the vocabulary of DESIGN.md) and the reason in numbers.  The library counts the notes per route
// (vnm_route_counts: what the GPU test suite's route-coverage check reads) and keeps the last note of the calling thread
// (vnm_route_last: "which way did that batch go, and why").  A handful of string operations per operator call, nothing per row.
void route_note(const char* route, const char* reason_fmt = nullptr, ...);

// ---------------------------------------------------------------------------------------------
// device-side column access
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int type_width(int t) {
    switch (t) {
        case VNM_I8: case VNM_U8: return 1;
        case VNM_I16: case VNM_U16: return 2;
        case VNM_I32: case VNM_U32: case VNM_F32: return 4;
        default: return 8;
    }
}
__host__ __device__ inline bool type_is_float(int t) { return t == VNM_F32 || t == VNM_F64; }
__host__ __device__ inline bool type_is_unsigned(int t) { return t >= VNM_U8 && t <= VNM_U64; }

// validity test, array_iterators.h:27-29: nulls_ptr && !GetBit(nulls_ptr, offset + i)
__device__ __forceinline__ bool col_valid(const vnm_dcol& c, int64_t i) {
    if (!c.validity) return true;
    int64_t b = c.offset + i;
    return (c.validity[b >> 3] >> (b & 7)) & 1;
}

// integer view of element i (sign / zero extended to 64 bits)
__device__ __forceinline__ int64_t col_i64(const vnm_dcol& c, int64_t i) {
    int64_t k = c.offset + i;
    switch (c.type) {
        case VNM_I8: return ((const int8_t*)c.values)[k];
        case VNM_I16: return ((const int16_t*)c.values)[k];
        case VNM_I32: return ((const int32_t*)c.values)[k];
        case VNM_I64: return ((const int64_t*)c.values)[k];
        case VNM_U8: return ((const uint8_t*)c.values)[k];
        case VNM_U16: return ((const uint16_t*)c.values)[k];
        case VNM_U32: return ((const uint32_t*)c.values)[k];
        case VNM_U64: return (int64_t)((const uint64_t*)c.values)[k];
        default: return 0;
    }
}
__device__ __forceinline__ double col_f64(const vnm_dcol& c, int64_t i) {
    int64_t k = c.offset + i;
    switch (c.type) {
        case VNM_F64: return ((const double*)c.values)[k];
        case VNM_F32: return (double)((const float*)c.values)[k];
        case VNM_U64: return (double)((const uint64_t*)c.values)[k];
        default: return (double)col_i64(c, i);
    }
}
// group-key bit pattern, array_iterators.h:215-217 (ints sign-extended via static_cast<uint64_t>)
// and :239-248 (floats: raw bits in the low bytes of a zeroed uint64)
__device__ __forceinline__ uint64_t col_key_bits(const vnm_dcol& c, int64_t i) {
    int64_t k = c.offset + i;
    switch (c.type) {
        case VNM_F64: return ((const uint64_t*)c.values)[k];
        case VNM_F32: return (uint64_t)((const uint32_t*)c.values)[k];
        default: return (uint64_t)col_i64(c, i);
    }
}

// raw element bits (zero-extended), for moving values without interpreting them
__device__ __forceinline__ uint64_t col_raw_bits(const vnm_dcol& c, int64_t i) {
    int64_t k = c.offset + i;
    switch (type_width(c.type)) {
        case 1: return ((const uint8_t*)c.values)[k];
        case 2: return ((const uint16_t*)c.values)[k];
        case 4: return ((const uint32_t*)c.values)[k];
        default: return ((const uint64_t*)c.values)[k];
    }
}

// order-preserving encodings into unsigned 64-bit (MIN/MAX accumulators, sort keys)
__host__ __device__ __forceinline__ uint64_t enc_i64(int64_t x) { return (uint64_t)x ^ 0x8000000000000000ULL; }
__host__ __device__ __forceinline__ int64_t dec_i64(uint64_t e) { return (int64_t)(e ^ 0x8000000000000000ULL); }
__host__ __device__ __forceinline__ uint64_t enc_f64(double d) {
    uint64_t b;
    memcpy(&b, &d, 8);
    return (b & 0x8000000000000000ULL) ? ~b : (b | 0x8000000000000000ULL);
}
__host__ __device__ __forceinline__ double dec_f64(uint64_t e) {
    uint64_t b = (e & 0x8000000000000000ULL) ? (e & 0x7FFFFFFFFFFFFFFFULL) : ~e;
    double d;
    memcpy(&d, &b, 8);
    return d;
}

// 64 -> 32 bit hash with two 32-bit multiplies (64-bit multiplies are slow VALU sequences on CDNA)
__host__ __device__ __forceinline__ uint32_t hash_u64(uint64_t k) {
    uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    uint32_t h = (lo ^ (hi * 0x9E3779B1u)) * 0x85EBCA6Bu;
    h ^= h >> 15;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

// comparison predicate in NumPy semantics (NaN compares False, != True)
template <typename T>
__device__ __forceinline__ bool cmp_apply(int op, T a, T b) {
    switch (op) {
        case VNM_EQ: return a == b;
        case VNM_NE: return a != b;
        case VNM_GT: return a > b;
        case VNM_GE: return a >= b;
        case VNM_LT: return a < b;
        default: return a <= b;
    }
}

// How a `column <op> literal` predicate is evaluated (vinum/arrow/record_batch.py:112-118 +
// vinum/core/expressions.py:30-36): a column WITH nulls (or any float operand) compares in float64
// with NULL -> NaN; an int column without nulls against an int literal compares as integers;
// float32 without nulls compares in float32.
enum CmpMode { CMP_F64 = 0, CMP_F32 = 1, CMP_I64 = 2, CMP_U64 = 3, CMP_CONST = 4 };
struct Predicate {
    int enabled;
    int op;
    int mode;
    int const_result;  // CMP_CONST: every row gives this (uint64 column vs negative literal)
    double dval;
    int64_t ival;
};
Predicate make_predicate(int col_type, bool col_has_nulls, int op, int scalar_is_float, double dval, int64_t ival);

__device__ __forceinline__ bool pred_eval(const Predicate& p, const vnm_dcol& c, int64_t i) {
    switch (p.mode) {
        case CMP_F64: {
            double a = col_valid(c, i) ? col_f64(c, i) : __builtin_nan("");
            return cmp_apply<double>(p.op, a, p.dval);
        }
        case CMP_F32: return cmp_apply<float>(p.op, ((const float*)c.values)[c.offset + i], (float)p.dval);
        case CMP_I64: return cmp_apply<int64_t>(p.op, col_i64(c, i), p.ival);
        case CMP_U64: return cmp_apply<uint64_t>(p.op, (uint64_t)col_i64(c, i), (uint64_t)p.ival);
        default: return p.const_result != 0;
    }
}

}  // namespace vnm
