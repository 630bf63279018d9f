// Filter: fused predicate + order-preserving stream compaction for gfx950.
//
// Replaces FilterOperator._kernel (vinum/core/algebra.py:119-123) + RecordBatch.filter
// (vinum/arrow/record_batch.py:85-90) + the NumPy comparison lambdas (vinum/core/expressions.py:30-36).
// The reference makes three passes (compare -> byte mask, byte mask -> bit mask, Arrow gather per
// column); here ONE kernel reads each input byte once and writes each surviving byte once:
//
//   * persistent workgroups pull 4096-row tiles from an atomic ticket (forward-progress safe),
//   * every lane loads 16 B (two 8-byte values) per request, eight requests in flight,
//   * wave64 ballots rank the survivors (no LDS traffic for the ranks),
//   * tile bases come from a decoupled look-back over 8-byte {flag,value} status words published
//     with agent-scope relaxed atomics (a single granule -> no fences needed),
//   * survivors are staged through LDS so the global stores are dense.
//
// Roofline: HBM.  Algorithmic bytes = 8*N read + 8*s*N written.
#include "vnm_common.hpp"

namespace vnm {

constexpr int FB = 256;                 // threads per workgroup (4 waves)
constexpr int F_CHUNKS = 8;             // 16-byte requests per lane per tile
constexpr int F_TILE = FB * 2 * F_CHUNKS;  // 4096 rows
constexpr int F_MAX_PAYLOAD = 8;
constexpr int MODE_MASK = 5;

constexpr uint64_t ST_AGG = 1ULL << 62;
constexpr uint64_t ST_INC = 2ULL << 62;
constexpr uint64_t ST_VAL = (1ULL << 62) - 1;

struct FilterArgs {
    vnm_dcol pred;
    Predicate p;
    const uint8_t* mask;
    const uint8_t* mask_valid;
    int n_payload;
    int reuse_pred;  // payload[0] is the predicate column itself (values stay in registers)
    vnm_dcol payload[F_MAX_PAYLOAD];
    void* out_values[F_MAX_PAYLOAD];
    uint8_t* out_valid[F_MAX_PAYLOAD];
    int64_t length;     // logical rows
    int64_t phys_base;  // first physical element index covered by tile 0 (even)
    int64_t ntiles;
    unsigned long long* ctl;  // [0] ticket, [1] total
    unsigned long long* status;
};

__device__ __forceinline__ uint64_t lanemask_lt() {
    uint32_t lane = __lane_id();
    return lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
}

template <int MODE>
__global__ __launch_bounds__(FB) void filter_kernel(FilterArgs a) {
    __shared__ int64_t s_tile;
    __shared__ uint32_t s_cnt[F_CHUNKS * 4];
    __shared__ uint32_t s_excl[F_CHUNKS * 4];
    __shared__ uint32_t s_total;
    __shared__ int64_t s_base;
    __shared__ uint64_t s_stage[F_TILE];  // 32 KB staging for dense stores

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint64_t lt = lanemask_lt();

    for (;;) {
        if (tid == 0) s_tile = (int64_t)atomicAdd(a.ctl, 1ULL);
        __syncthreads();
        const int64_t tile = s_tile;
        if (tile >= a.ntiles) break;

        // physical element index of this lane's first element in chunk 0; logical row = phys - pred.offset
        const int64_t pbase = a.phys_base + tile * F_TILE + 2 * tid;
        uint32_t flags = 0;
        uint32_t rank[F_CHUNKS];
        double v0[F_CHUNKS], v1[F_CHUNKS];

        if (MODE == CMP_F64 && a.pred.type == VNM_F64) {
            // hot path: issue all eight 16-byte loads first
            const double* vals = (const double*)a.pred.values;
#pragma unroll
            for (int j = 0; j < F_CHUNKS; j++) {
                int64_t p0 = pbase + (int64_t)j * (2 * FB);
                int64_t r0 = p0 - a.pred.offset;
                if (r0 >= 0 && r0 + 1 < a.length) {
                    double2 t = *(const double2*)(vals + p0);
                    v0[j] = t.x; v1[j] = t.y;
                } else {
                    v0[j] = (r0 >= 0 && r0 < a.length) ? vals[p0] : 0.0;
                    v1[j] = (r0 + 1 >= 0 && r0 + 1 < a.length) ? vals[p0 + 1] : 0.0;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < F_CHUNKS; j++) {
            int64_t p0 = pbase + (int64_t)j * (2 * FB);
            int64_t r0 = p0 - a.pred.offset, r1 = r0 + 1;
            bool in0 = r0 >= 0 && r0 < a.length, in1 = r1 >= 0 && r1 < a.length;
            bool f0 = false, f1 = false;
            if (MODE == MODE_MASK) {
                if (in0) f0 = a.mask_valid && !a.mask_valid[r0] ? true : a.mask[r0] != 0;
                if (in1) f1 = a.mask_valid && !a.mask_valid[r1] ? true : a.mask[r1] != 0;
            } else if (MODE == CMP_F64 && a.pred.type == VNM_F64) {
                double x0 = (in0 && col_valid(a.pred, r0)) ? v0[j] : __builtin_nan("");
                double x1 = (in1 && col_valid(a.pred, r1)) ? v1[j] : __builtin_nan("");
                f0 = in0 && cmp_apply<double>(a.p.op, x0, a.p.dval);
                f1 = in1 && cmp_apply<double>(a.p.op, x1, a.p.dval);
            } else {
                f0 = in0 && pred_eval(a.p, a.pred, r0);
                f1 = in1 && pred_eval(a.p, a.pred, r1);
            }
            uint64_t b0 = __ballot(f0), b1 = __ballot(f1);
            rank[j] = __popcll(b0 & lt) + __popcll(b1 & lt);
            flags |= (f0 ? 1u : 0u) << (2 * j) | (f1 ? 1u : 0u) << (2 * j + 1);
            if (lane == 0) s_cnt[j * 4 + wave] = __popcll(b0) + __popcll(b1);
        }
        __syncthreads();

        // ---- wave 0: scan the 32 segment counts, then decoupled look-back for the tile base ----
        if (wave == 0) {
            uint32_t c = lane < F_CHUNKS * 4 ? s_cnt[lane] : 0;
            uint32_t inc = c;
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) {
                uint32_t o = __shfl_up(inc, d);
                if (lane >= d) inc += o;
            }
            if (lane < F_CHUNKS * 4) s_excl[lane] = inc - c;
            uint32_t total = __shfl(inc, 31);
            int64_t excl = 0;
            if (tile > 0) {
                if (lane == 0)
                    __hip_atomic_store(&a.status[tile], ST_AGG | (uint64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int64_t look = tile - 1;
                for (;;) {
                    int64_t idx = look - lane;
                    uint64_t s = idx >= 0 ? __hip_atomic_load(&a.status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ST_INC;
                    uint64_t flag = s >> 62;
                    uint64_t incl_mask = __ballot(flag == 2);
                    uint64_t zero_mask = __ballot(flag == 0);
                    uint64_t val = s & ST_VAL;
                    if (incl_mask) {
                        // only the predecessors up to the nearest inclusive prefix matter
                        int first = __ffsll((unsigned long long)incl_mask) - 1;
                        uint64_t need = first == 0 ? 0ULL : (~0ULL >> (64 - first));
                        if (zero_mask & need) { __builtin_amdgcn_s_sleep(1); continue; }
                        if (lane > first) val = 0;
                    } else if (zero_mask) {
                        __builtin_amdgcn_s_sleep(1);
                        continue;
                    }
                    // wave sum of val
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) val += __shfl_xor(val, d);
                    excl += (int64_t)val;
                    if (incl_mask) break;
                    look -= 64;
                }
            }
            if (lane == 0) {
                __hip_atomic_store(&a.status[tile], ST_INC | (uint64_t)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_base = excl;
                s_total = total;
                if (tile == a.ntiles - 1) a.ctl[1] = (unsigned long long)(excl + total);
            }
        }
        __syncthreads();
        const int64_t base = s_base;
        const uint32_t total = s_total;

        // ---- write survivors, one payload column at a time, staged through LDS for dense stores ----
        for (int k = 0; k < a.n_payload; k++) {
            const vnm_dcol& c = a.payload[k];
            const int w = type_width(c.type);
            const bool reuse = (k == 0 && a.reuse_pred && MODE == CMP_F64 && a.pred.type == VNM_F64);
            // (1) survivors -> LDS at their tile-local rank (as 64-bit cells)
#pragma unroll
            for (int j = 0; j < F_CHUNKS; j++) {
                uint32_t fj = (flags >> (2 * j)) & 3u;
                if (!fj) continue;
                int64_t r0 = pbase + (int64_t)j * (2 * FB) - a.pred.offset;
                uint32_t pos = s_excl[j * 4 + wave] + rank[j];
                if (fj & 1u) {
                    uint64_t bits;
                    if (reuse) bits = __double_as_longlong(v0[j]);
                    else if (w == 8) bits = ((const uint64_t*)c.values)[c.offset + r0];
                    else bits = col_raw_bits(c, r0);
                    s_stage[pos++] = bits;
                }
                if (fj & 2u) {
                    uint64_t bits;
                    if (reuse) bits = __double_as_longlong(v1[j]);
                    else if (w == 8) bits = ((const uint64_t*)c.values)[c.offset + r0 + 1];
                    else bits = col_raw_bits(c, r0 + 1);
                    s_stage[pos] = bits;
                }
            }
            __syncthreads();
            // (2) dense copy LDS -> global
            if (w == 8) {
                uint64_t* out = (uint64_t*)a.out_values[k] + base;
                for (uint32_t i = tid; i < total; i += FB) out[i] = s_stage[i];
            } else if (w == 4) {
                uint32_t* out = (uint32_t*)a.out_values[k] + base;
                for (uint32_t i = tid; i < total; i += FB) out[i] = (uint32_t)s_stage[i];
            } else if (w == 2) {
                uint16_t* out = (uint16_t*)a.out_values[k] + base;
                for (uint32_t i = tid; i < total; i += FB) out[i] = (uint16_t)s_stage[i];
            } else {
                uint8_t* out = (uint8_t*)a.out_values[k] + base;
                for (uint32_t i = tid; i < total; i += FB) out[i] = (uint8_t)s_stage[i];
            }
            // (3) validity bytes (only for payloads that carry a bitmap, or emit_null masks)
            if (a.out_valid[k]) {
                __syncthreads();
                uint8_t* stage8 = (uint8_t*)s_stage;
#pragma unroll
                for (int j = 0; j < F_CHUNKS; j++) {
                    uint32_t fj = (flags >> (2 * j)) & 3u;
                    if (!fj) continue;
                    int64_t r0 = pbase + (int64_t)j * (2 * FB) - a.pred.offset;
                    uint32_t pos = s_excl[j * 4 + wave] + rank[j];
                    if (fj & 1u) {
                        bool ok = col_valid(c, r0) && !(MODE == MODE_MASK && a.mask_valid && !a.mask_valid[r0]);
                        stage8[pos++] = ok;
                    }
                    if (fj & 2u) {
                        bool ok = col_valid(c, r0 + 1) && !(MODE == MODE_MASK && a.mask_valid && !a.mask_valid[r0 + 1]);
                        stage8[pos] = ok;
                    }
                }
                __syncthreads();
                uint8_t* ov = a.out_valid[k] + base;
                for (uint32_t i = tid; i < total; i += FB) ov[i] = stage8[i];
            }
            __syncthreads();
        }
    }
}

// byte-per-row validity -> Arrow bitmap (LSB first), 8 rows per lane
__global__ void pack_validity_kernel(const uint8_t* bytes, int64_t n, uint8_t* bits) {
    const int64_t nb = (n + 7) >> 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += stride) {
        uint8_t v = 0;
        for (int k = 0; k < 8; k++) {
            int64_t i = b * 8 + k;
            if (i < n && bytes[i]) v |= (uint8_t)(1u << k);
        }
        bits[b] = v;
    }
}

static int launch_filter(FilterArgs& a, int mode, int64_t* out_count, hipStream_t s) {
    DeviceInfo& d = device_info();
    a.phys_base = a.pred.values ? (a.pred.offset & ~1LL) : 0;
    int64_t span = (a.pred.values ? a.pred.offset : 0) + a.length - a.phys_base;
    a.ntiles = (span + F_TILE - 1) / F_TILE;
    if (a.length == 0 || a.ntiles == 0) {
        *out_count = 0;
        return 0;
    }
    size_t sbytes = (size_t)(a.ntiles + 2) * 8;
    unsigned long long* scratch = (unsigned long long*)pool_alloc(sbytes);
    if (!scratch) return 1;
    VNM_HIP(hipMemsetAsync(scratch, 0, sbytes, s));
    a.ctl = scratch;
    a.status = scratch + 2;
    int64_t grid = (int64_t)d.num_cus * 8;
    if (grid > a.ntiles) grid = a.ntiles;
    {
    KernelTimer timer("filter_kernel", s);
    switch (mode) {
        case CMP_F64: filter_kernel<CMP_F64><<<(int)grid, FB, 0, s>>>(a); break;
        case MODE_MASK: filter_kernel<MODE_MASK><<<(int)grid, FB, 0, s>>>(a); break;
        default: filter_kernel<CMP_I64><<<(int)grid, FB, 0, s>>>(a); break;  // generic pred_eval path
    }
    }
    VNM_HIP(hipGetLastError());
    unsigned long long total = 0;
    VNM_HIP(hipMemcpyAsync(&total, scratch + 1, 8, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    pool_free(scratch);
    *out_count = (int64_t)total;
    return 0;
}

}  // namespace vnm

using namespace vnm;

extern "C" {

int64_t vnm_filter_scratch_bytes(int64_t length) { return ((length + F_TILE) / F_TILE + 3) * 8; }

int vnm_pack_validity(const uint8_t* valid_bytes, int64_t n, uint8_t* bitmap, void* stream) {
    VNM_TRY(ensure_init());
    if (n <= 0) return 0;
    int64_t nb = (n + 7) >> 3;
    int grid = (int)((nb + 255) / 256 < 2048 ? (nb + 255) / 256 : 2048);
    pack_validity_kernel<<<grid, 256, 0, as_stream(stream)>>>(valid_bytes, n, bitmap);
    VNM_HIP(hipGetLastError());
    return 0;
}

int vnm_filter_cmp(const vnm_dcol* pred, int op, int scalar_is_float, double dval, int64_t ival, int n_payload,
                   const vnm_dcol* payload, void** out_values, uint8_t** out_valid, int64_t* out_count,
                   void* stream) {
    VNM_TRY(ensure_init());
    if (!pred || !out_count) return set_error("vnm_filter_cmp: null argument");
    if (n_payload < 0 || n_payload > F_MAX_PAYLOAD) return set_error("vnm_filter_cmp: at most %d payload columns per call", F_MAX_PAYLOAD);
    if (op < VNM_EQ || op > VNM_LE) return set_error("vnm_filter_cmp: bad comparison op %d", op);
    FilterArgs a{};
    a.pred = *pred;
    a.length = pred->length;
    a.p = make_predicate(pred->type, pred->validity != nullptr, op, scalar_is_float, dval, ival);
    a.n_payload = n_payload;
    for (int k = 0; k < n_payload; k++) {
        if (payload[k].length != pred->length) return set_error("vnm_filter_cmp: payload %d length mismatch", k);
        a.payload[k] = payload[k];
        a.out_values[k] = out_values[k];
        a.out_valid[k] = out_valid ? out_valid[k] : nullptr;
        if (payload[k].validity && !a.out_valid[k]) return set_error("vnm_filter_cmp: payload %d has nulls but no out_valid buffer", k);
    }
    a.reuse_pred = n_payload > 0 && payload[0].values == pred->values && payload[0].offset == pred->offset &&
                   payload[0].type == pred->type;
    int mode = (a.p.mode == CMP_F64) ? CMP_F64 : CMP_I64;
    return launch_filter(a, mode, out_count, as_stream(stream));
}

int vnm_filter_mask(const uint8_t* mask, const uint8_t* mask_valid, int64_t length, int n_payload,
                    const vnm_dcol* payload, void** out_values, uint8_t** out_valid, int64_t* out_count,
                    void* stream) {
    VNM_TRY(ensure_init());
    if (!mask || !out_count) return set_error("vnm_filter_mask: null argument");
    if (n_payload < 0 || n_payload > F_MAX_PAYLOAD) return set_error("vnm_filter_mask: at most %d payload columns per call", F_MAX_PAYLOAD);
    FilterArgs a{};
    a.mask = mask;
    a.mask_valid = mask_valid;
    a.length = length;
    a.n_payload = n_payload;
    for (int k = 0; k < n_payload; k++) {
        if (payload[k].length != length) return set_error("vnm_filter_mask: payload %d length mismatch", k);
        a.payload[k] = payload[k];
        a.out_values[k] = out_values[k];
        a.out_valid[k] = out_valid ? out_valid[k] : nullptr;
        if ((payload[k].validity || mask_valid) && !a.out_valid[k])
            return set_error("vnm_filter_mask: payload %d needs an out_valid buffer", k);
    }
    return launch_filter(a, MODE_MASK, out_count, as_stream(stream));
}

}  // extern "C"
