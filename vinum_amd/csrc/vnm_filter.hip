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
#include <cstdlib>

#include "vnm_common.hpp"

namespace vnm {

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
    int debug;  // timing experiments only: 1 = skip look-back (outputs are wrong)
};

__device__ __forceinline__ uint64_t lanemask_lt() {
    uint32_t lane = __lane_id();
    return lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
}

template <int MODE, int FB, bool HOT>
__global__ __launch_bounds__(FB) void filter_kernel(FilterArgs a) {
    constexpr int F_CHUNKS = 8;             // 16-byte requests per lane per tile
    constexpr int F_TILE = FB * 2 * F_CHUNKS;
    constexpr int NW = FB / 64;             // waves per workgroup
    constexpr int NSEG = F_CHUNKS * NW;     // (chunk, wave) segments per tile, <= 128
    static_assert(NSEG <= 128, "segment scan handles two segments per lane of wave 0");
    __shared__ int64_t s_next[2];
    __shared__ uint32_t s_cnt[NSEG];
    __shared__ uint32_t s_excl[NSEG];
    __shared__ int64_t s_base;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint64_t lt = lanemask_lt();
    // HOT: float64 predicate column without a validity bitmap (host-checked); values live in registers and
    // the next tile is prefetched.  Otherwise predicates go through the generic per-element evaluator.
    constexpr bool hot = HOT;
    const double* vals = (const double*)a.pred.values;

    double cv0[F_CHUNKS], cv1[F_CHUNKS];  // values of the tile being processed
    double nv0[F_CHUNKS], nv1[F_CHUNKS];  // prefetched values of the next tile

    // The loop is software pipelined: the ticket and the loads of tile t+1 are issued BEFORE the
    // look-back and the stores of tile t, so HBM reads stay in flight across the serial part.
#define VNM_LOAD_TILE(T, V0, V1)                                                                  \
    do {                                                                                          \
        const int64_t pb_ = a.phys_base + (T) * F_TILE + 2 * tid;                                 \
        const int64_t first_ = a.phys_base + (T) * F_TILE - a.pred.offset;                        \
        if (first_ >= 0 && first_ + F_TILE <= a.length) {                                         \
            const double2* src_ = (const double2*)(vals + pb_);                                   \
            _Pragma("unroll") for (int j = 0; j < F_CHUNKS; j++) {                                \
                double2 t_ = src_[j * FB];                                                        \
                V0[j] = t_.x; V1[j] = t_.y;                                                       \
            }                                                                                     \
        } else {                                                                                  \
            _Pragma("unroll") for (int j = 0; j < F_CHUNKS; j++) {                                \
                int64_t p0 = pb_ + (int64_t)j * (2 * FB);                                         \
                int64_t r0 = p0 - a.pred.offset;                                                  \
                V0[j] = (r0 >= 0 && r0 < a.length) ? vals[p0] : 0.0;                              \
                V1[j] = (r0 + 1 >= 0 && r0 + 1 < a.length) ? vals[p0 + 1] : 0.0;                  \
            }                                                                                     \
        }                                                                                         \
    } while (0)

    if (tid == 0) s_next[1] = (int64_t)atomicAdd(a.ctl, 1ULL);
    __syncthreads();
    int64_t tile = s_next[1];
    if (tile < a.ntiles && hot) VNM_LOAD_TILE(tile, cv0, cv1);

    for (int it = 0; tile < a.ntiles; it++) {
        if (tid == 0) s_next[it & 1] = (int64_t)atomicAdd(a.ctl, 1ULL);

        // physical element index of this lane's first element in chunk 0; logical row = phys - pred.offset
        const int64_t pbase = a.phys_base + tile * F_TILE + 2 * tid;
        uint32_t flags = 0;
        uint32_t rank[F_CHUNKS];
#pragma unroll
        for (int j = 0; j < F_CHUNKS; j++) {
            int64_t p0 = pbase + (int64_t)j * (2 * FB);
            int64_t r0 = p0 - a.pred.offset, r1 = r0 + 1;
            bool in0 = r0 >= 0 && r0 < a.length, in1 = r1 >= 0 && r1 < a.length;
            bool f0 = false, f1 = false;
            if (MODE == MODE_MASK) {
                if (in0) f0 = a.mask_valid && !a.mask_valid[r0] ? true : a.mask[r0] != 0;
                if (in1) f1 = a.mask_valid && !a.mask_valid[r1] ? true : a.mask[r1] != 0;
            } else if (hot) {
                f0 = in0 && cmp_apply<double>(a.p.op, cv0[j], a.p.dval);
                f1 = in1 && cmp_apply<double>(a.p.op, cv1[j], a.p.dval);
            } else {
                f0 = in0 && pred_eval(a.p, a.pred, r0);
                f1 = in1 && pred_eval(a.p, a.pred, r1);
            }
            uint64_t b0 = __ballot(f0), b1 = __ballot(f1);
            rank[j] = __popcll(b0 & lt) + __popcll(b1 & lt);
            flags |= (f0 ? 1u : 0u) << (2 * j) | (f1 ? 1u : 0u) << (2 * j + 1);
            if (lane == 0) s_cnt[j * NW + wave] = __popcll(b0) + __popcll(b1);
        }
        __syncthreads();  // #1: counts and next ticket visible

        const int64_t ntile = s_next[it & 1];
        if (ntile < a.ntiles && hot) VNM_LOAD_TILE(ntile, nv0, nv1);

        // ---- wave 0: scan the segment counts, then decoupled look-back for the tile base ----
        if (wave == 0) {
            uint32_t c0 = 2 * lane < NSEG ? s_cnt[2 * lane] : 0;
            uint32_t c1 = 2 * lane + 1 < NSEG ? s_cnt[2 * lane + 1] : 0;
            uint32_t c = c0 + c1;
            uint32_t inc = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t o = __shfl_up(inc, d);
                if (lane >= d) inc += o;
            }
            if (2 * lane < NSEG) s_excl[2 * lane] = inc - c;
            if (2 * lane + 1 < NSEG) s_excl[2 * lane + 1] = inc - c + c0;
            uint32_t total = __shfl(inc, 63);
            int64_t excl = 0;
            if (tile > 0 && (a.debug & 1)) excl = tile * (F_TILE / 2);
            else if (tile > 0) {
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
                if (tile == a.ntiles - 1) a.ctl[1] = (unsigned long long)(excl + total);
            }
        }
        __syncthreads();  // #2: tile base known
        const int64_t base = s_base;

        // ---- write survivors: within a wave the survivors of one chunk land on a contiguous range, so
        // the two store instructions of a chunk together cover whole cache lines ----
        if (hot) {
            // the only payload is the predicate column itself: values are still in registers
            if (a.n_payload && !(a.debug & 4)) {
                uint64_t* out = (uint64_t*)a.out_values[0] + base;
#pragma unroll
                for (int j = 0; j < F_CHUNKS; j++) {
                    uint32_t fj = (flags >> (2 * j)) & 3u;
                    uint32_t pos = s_excl[j * NW + wave] + rank[j];
                    if (fj & 1u) out[pos++] = (uint64_t)__double_as_longlong(cv0[j]);
                    if (fj & 2u) out[pos] = (uint64_t)__double_as_longlong(cv1[j]);
                }
            }
        } else
        for (int k = 0; k < ((a.debug & 4) ? 0 : a.n_payload); k++) {
            const vnm_dcol& c = a.payload[k];
            const int w = type_width(c.type);
            const bool reuse = false;
#pragma unroll
            for (int j = 0; j < F_CHUNKS; j++) {
                uint32_t fj = (flags >> (2 * j)) & 3u;
                if (!fj) continue;
                int64_t r0 = pbase + (int64_t)j * (2 * FB) - a.pred.offset;
                int64_t pos = base + s_excl[j * NW + wave] + rank[j];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    if (!(fj & (1u << e))) continue;
                    uint64_t bits;
                    if (reuse) bits = (uint64_t)__double_as_longlong(e ? cv1[j] : cv0[j]);
                    else bits = col_raw_bits(c, r0 + e);
                    switch (w) {
                        case 8: ((uint64_t*)a.out_values[k])[pos] = bits; break;
                        case 4: ((uint32_t*)a.out_values[k])[pos] = (uint32_t)bits; break;
                        case 2: ((uint16_t*)a.out_values[k])[pos] = (uint16_t)bits; break;
                        default: ((uint8_t*)a.out_values[k])[pos] = (uint8_t)bits; break;
                    }
                    if (a.out_valid[k])
                        a.out_valid[k][pos] = col_valid(c, r0 + e) && !(MODE == MODE_MASK && a.mask_valid && !a.mask_valid[r0 + e]);
                    pos++;
                }
            }
        }
        // rotate the pipeline
#pragma unroll
        for (int j = 0; j < F_CHUNKS; j++) { if (hot) { cv0[j] = nv0[j]; cv1[j] = nv1[j]; } }
        tile = ntile;
    }
#undef VNM_LOAD_TILE
}

// =======================================================================================================
// Hot filter kernel: float64 predicate column without validity, output = the compacted column itself
// (BASELINE configs[1]).  Three-stage software pipeline per workgroup so that neither the HBM reads nor
// the successors of our tiles ever wait on a look-back:
//     C = loads in flight (ticket just taken)   B = loaded -> counted -> AGGREGATE PUBLISHED
//     A = counted one iteration ago -> look-back -> stores
// A tile's aggregate is published as soon as its loads land (it depends on nothing else), and its
// look-back runs one iteration later, when most predecessors have already published.
// =======================================================================================================
// Decoupled look-back over a 256-tile window (four status words per lane requested together).  sw[] holds
// the window starting at tile - 1 prefetched by the caller.  Returns the exclusive prefix of `tile`.
__device__ __forceinline__ int64_t lookback(unsigned long long* status, int64_t tile, int lane, uint64_t* sw) {
    int64_t excl = 0;
    int64_t look = tile - 1;
    bool fresh = true;
    for (;;) {
        if (!fresh) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int64_t idx = look - (q * 64 + lane);
                sw[q] = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ST_INC;
            }
        }
        fresh = false;
        bool done = false, retry = false;
        int64_t acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (done || retry) continue;
            uint64_t flag = sw[q] >> 62;
            uint64_t incl_mask = __ballot(flag == 2);
            uint64_t zero_mask = __ballot(flag == 0);
            uint64_t val = sw[q] & ST_VAL;
            if (incl_mask) {
                // only the predecessors up to the nearest inclusive prefix matter
                int first = __ffsll((unsigned long long)incl_mask) - 1;
                uint64_t need = first == 0 ? 0ULL : (~0ULL >> (64 - first));
                if (zero_mask & need) { retry = true; continue; }
                if (lane > first) val = 0;
                done = true;
            } else if (zero_mask) {
                retry = true;
                continue;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) val += __shfl_xor(val, d);
            acc += (int64_t)val;
        }
        if (retry) { __builtin_amdgcn_s_sleep(1); continue; }
        excl += acc;
        if (done) return excl;
        look -= 256;
    }
}

constexpr int FH_THREADS = 512;
constexpr int FH_CHUNKS = 8;
constexpr int FH_TILE = FH_THREADS * 2 * FH_CHUNKS;  // 8192 rows
constexpr int FH_NW = FH_THREADS / 64;
constexpr int FH_NSEG = FH_CHUNKS * FH_NW;           // 64: one segment per lane of wave 0

__global__ __launch_bounds__(FH_THREADS) void filter_hot_kernel(FilterArgs a) {
    __shared__ int64_t s_next[2];
    __shared__ uint32_t s_cnt[FH_NSEG];
    __shared__ uint32_t s_excl[2][FH_NSEG];
    __shared__ int64_t s_base;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint64_t lt = lanemask_lt();
    const double* vals = (const double*)a.pred.values;
    const int op = a.p.op;
    const double thr = a.p.dval;

    // three register sets that ROTATE ROLES (A -> C -> B -> A) instead of being copied: a register copy of
    // a set whose loads are still in flight would force an s_waitcnt on them and serialise the pipeline
    double s0v0[FH_CHUNKS], s0v1[FH_CHUNKS], s1v0[FH_CHUNKS], s1v1[FH_CHUNKS], s2v0[FH_CHUNKS], s2v1[FH_CHUNKS];
    uint32_t s0rank[FH_CHUNKS], s1rank[FH_CHUNKS], s2rank[FH_CHUNKS];
    uint32_t s0flags = 0, s1flags = 0, s2flags = 0;
    uint32_t a_total = 0;  // wave 0: total of tile A

#define VNM_LOAD(T, V0, V1)                                                                       \
    do {                                                                                          \
        const int64_t pb_ = a.phys_base + (T) * FH_TILE + 2 * tid;                                \
        const int64_t first_ = a.phys_base + (T) * FH_TILE - a.pred.offset;                       \
        if (first_ >= 0 && first_ + FH_TILE <= a.length) {                                        \
            const double2* src_ = (const double2*)(vals + pb_);                                   \
            _Pragma("unroll") for (int j = 0; j < FH_CHUNKS; j++) {                               \
                double2 t_ = src_[j * FH_THREADS];                                                \
                V0[j] = t_.x; V1[j] = t_.y;                                                       \
            }                                                                                     \
        } else {                                                                                  \
            _Pragma("unroll") for (int j = 0; j < FH_CHUNKS; j++) {                               \
                int64_t p0 = pb_ + (int64_t)j * (2 * FH_THREADS);                                 \
                int64_t r0 = p0 - a.pred.offset;                                                  \
                V0[j] = (r0 >= 0 && r0 < a.length) ? vals[p0] : __builtin_nan("");                \
                V1[j] = (r0 + 1 >= 0 && r0 + 1 < a.length) ? vals[p0 + 1] : __builtin_nan("");    \
            }                                                                                     \
        }                                                                                         \
    } while (0)

    // out-of-range elements of a ragged tile must not survive whatever the operator is
#define VNM_COUNT(T, V0, V1, FLAGS, RANK)                                                         \
    do {                                                                                          \
        const int64_t rb_ = a.phys_base + (T) * FH_TILE + 2 * tid - a.pred.offset;                \
        FLAGS = 0;                                                                                \
        _Pragma("unroll") for (int j = 0; j < FH_CHUNKS; j++) {                                   \
            int64_t r0 = rb_ + (int64_t)j * (2 * FH_THREADS);                                     \
            bool f0 = r0 >= 0 && r0 < a.length && cmp_apply<double>(op, V0[j], thr);              \
            bool f1 = r0 + 1 >= 0 && r0 + 1 < a.length && cmp_apply<double>(op, V1[j], thr);      \
            uint64_t b0 = __ballot(f0), b1 = __ballot(f1);                                        \
            RANK[j] = __popcll(b0 & lt) + __popcll(b1 & lt);                                      \
            FLAGS |= (f0 ? 1u : 0u) << (2 * j) | (f1 ? 1u : 0u) << (2 * j + 1);                   \
            if (lane == 0) s_cnt[j * FH_NW + wave] = __popcll(b0) + __popcll(b1);                 \
        }                                                                                         \
    } while (0)

    // Tickets are requested one iteration before they are needed (a returning atomic on the shared ticket
    // word takes 1-3 us under streaming load); `pending` lives in thread 0 only.  This file is built with
    // -amdgpu-atomic-optimizer-strategy=None: the optimizer's wave-aggregation epilogue reads the result
    // back immediately (s_waitcnt vmcnt(0)), which would drain the whole load pipeline every iteration.
    unsigned long long pending = 0;
    if (tid == 0) {
        s_next[1] = (int64_t)__hip_atomic_fetch_add(a.ctl, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pending = __hip_atomic_fetch_add(a.ctl, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    int64_t tile_a = -1;
    int64_t tile_b = s_next[1];
    int64_t tile_c = 0;
    if (tile_b < a.ntiles) VNM_LOAD(tile_b, s1v0, s1v1);
    int it = 0;

    // one pipeline step with register sets in roles A (store), B (count + publish), C (load)
#define VNM_STEP(A, B, C)                                                                                   \
    {                                                                                                       \
        const int par = it & 1;                                                                             \
        it++;                                                                                               \
        /* (0) wave 0 requests A's 256-tile look-back window FIRST: under streaming load a status read */   \
        /* queues behind every HBM request this CU already issued, so it goes out before the bulk loads */  \
        uint64_t sw[4];                                                                                     \
        if (wave == 0 && tile_a > 0) {                                                                      \
            _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                 \
                int64_t idx = tile_a - 1 - (q * 64 + lane);                                                 \
                sw[q] = idx >= 0 ? __hip_atomic_load(&a.status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ST_INC; \
            }                                                                                               \
        }                                                                                                   \
        /* (1) hand out the ticket requested one step ago, request the next one */                          \
        if (tid == 0) {                                                                                     \
            s_next[par] = (int64_t)pending;                                                                 \
            pending = __hip_atomic_fetch_add(a.ctl, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      \
        }                                                                                                   \
        /* (2) count B (its loads were issued one step ago) */                                              \
        if (tile_b < a.ntiles) VNM_COUNT(tile_b, B##v0, B##v1, B##flags, B##rank);                          \
        __syncthreads(); /* #1: s_cnt(B) and the ticket are visible */                                      \
        tile_c = s_next[par];                                                                               \
        if (wave != 0 && tile_c < a.ntiles) VNM_LOAD(tile_c, C##v0, C##v1);                                 \
        if (wave == 0) {                                                                                    \
            /* (3) publish B's aggregate right away: it depends on nothing but B's loads */                 \
            uint32_t b_total = 0;                                                                           \
            if (tile_b < a.ntiles) {                                                                        \
                uint32_t c = s_cnt[lane];                                                                   \
                uint32_t inc = c;                                                                           \
                _Pragma("unroll") for (int d = 1; d < 64; d <<= 1) {                                        \
                    uint32_t o = __shfl_up(inc, d);                                                         \
                    if (lane >= d) inc += o;                                                                \
                }                                                                                           \
                s_excl[par][lane] = inc - c;                                                                \
                b_total = __shfl(inc, 63);                                                                  \
                if (lane == 0)                                                                              \
                    __hip_atomic_store(&a.status[tile_b], (tile_b > 0 ? ST_AGG : ST_INC) | (uint64_t)b_total, \
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                         \
            }                                                                                               \
            /* (4) look-back for A (its aggregate was published one step ago) */                            \
            if (tile_a >= 0) {                                                                              \
                int64_t excl = 0;                                                                           \
                if (tile_a > 0 && (a.debug & 1)) excl = tile_a * (FH_TILE / 2);                             \
                else if (tile_a > 0) {                                                                      \
                    excl = lookback(a.status, tile_a, lane, sw);                                            \
                    if (lane == 0)                                                                          \
                        __hip_atomic_store(&a.status[tile_a], ST_INC | (uint64_t)(excl + a_total),          \
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                     \
                }                                                                                           \
                if (lane == 0) {                                                                            \
                    s_base = excl;                                                                          \
                    if (tile_a == a.ntiles - 1) a.ctl[1] = (unsigned long long)(excl + a_total);            \
                }                                                                                           \
            }                                                                                               \
            a_total = b_total;                                                                              \
            if (tile_c < a.ntiles) VNM_LOAD(tile_c, C##v0, C##v1);                                          \
        }                                                                                                   \
        __syncthreads(); /* #2: base(A) and s_excl(B) are visible */                                        \
        /* (5) stores of A: per wave and chunk the survivors land on a contiguous range */                  \
        if (tile_a >= 0 && a.n_payload && !(a.debug & 4)) {                                                 \
            uint64_t* out = (uint64_t*)a.out_values[0] + s_base;                                            \
            _Pragma("unroll") for (int j = 0; j < FH_CHUNKS; j++) {                                         \
                uint32_t fj = (A##flags >> (2 * j)) & 3u;                                                   \
                uint32_t pos = s_excl[par ^ 1][j * FH_NW + wave] + A##rank[j];                              \
                if (fj & 1u) out[pos++] = (uint64_t)__double_as_longlong(A##v0[j]);                         \
                if (fj & 2u) out[pos] = (uint64_t)__double_as_longlong(A##v1[j]);                           \
            }                                                                                               \
        }                                                                                                   \
        tile_a = tile_b < a.ntiles ? tile_b : -1;                                                           \
        tile_b = tile_c;                                                                                    \
    }

    for (;;) {
        VNM_STEP(s0, s1, s2)
        if (!(tile_a >= 0 || tile_b < a.ntiles)) break;
        VNM_STEP(s1, s2, s0)
        if (!(tile_a >= 0 || tile_b < a.ntiles)) break;
        VNM_STEP(s2, s0, s1)
        if (!(tile_a >= 0 || tile_b < a.ntiles)) break;
    }
#undef VNM_STEP
#undef VNM_LOAD
#undef VNM_COUNT
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

template <int MODE, int FB>
static void launch_variant(const FilterArgs& a, int grid, hipStream_t s) {
    const bool hot = MODE == CMP_F64 && a.pred.type == VNM_F64 && !a.pred.validity &&
                     (a.n_payload == 0 || (a.n_payload == 1 && a.reuse_pred && !a.out_valid[0]));
    if (hot) filter_kernel<MODE, FB, true><<<grid, FB, 0, s>>>(a);
    else filter_kernel<MODE, FB, false><<<grid, FB, 0, s>>>(a);
}

// tuning knobs (defaults chosen by measurement, profiles/): VNM_FILTER_THREADS in {256,512,1024},
// VNM_FILTER_WGS_PER_CU
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

static int launch_filter(FilterArgs& a, int mode, int64_t* out_count, hipStream_t s) {
    DeviceInfo& d = device_info();
    const bool hot = mode == CMP_F64 && a.pred.type == VNM_F64 && !a.pred.validity &&
                     (a.n_payload == 0 || (a.n_payload == 1 && a.reuse_pred && !a.out_valid[0])) &&
                     env_int("VNM_FILTER_PIPE3", 0) != 0;  // 3-stage variant: measured slower (profiles/filter_tuning_r01.md)
    const bool hot2 = mode == CMP_F64 && a.pred.type == VNM_F64 && !a.pred.validity &&
                      (a.n_payload == 0 || (a.n_payload == 1 && a.reuse_pred && !a.out_valid[0]));
    int fb = hot ? FH_THREADS : env_int("VNM_FILTER_THREADS", hot2 ? 1024 : 512);
    fb = fb >= 1024 ? 1024 : (fb >= 512 ? 512 : 256);
    const int chunks = 8;
    const int tile_rows = hot ? FH_TILE : fb * 2 * chunks;
    a.phys_base = a.pred.values ? (a.pred.offset & ~1LL) : 0;
    int64_t span = (a.pred.values ? a.pred.offset : 0) + a.length - a.phys_base;
    a.ntiles = (span + tile_rows - 1) / tile_rows;
    if (a.length == 0 || a.ntiles == 0) {
        *out_count = 0;
        return 0;
    }
    size_t sbytes = (size_t)(a.ntiles + 8) * 8;
    unsigned long long* scratch = (unsigned long long*)pool_alloc(sbytes);
    if (!scratch) return 1;
    VNM_HIP(hipMemsetAsync(scratch, 0, sbytes, s));
    a.ctl = scratch;
    a.status = scratch + 8;
    a.debug = env_int("VNM_FILTER_DEBUG", 0);
    int64_t grid = (int64_t)d.num_cus * env_int("VNM_FILTER_WGS_PER_CU", 2048 / fb);
    if (grid > a.ntiles) grid = a.ntiles;
    {
    KernelTimer timer("filter_kernel", s);
#define VNM_LAUNCH(M)                                                    \
    do {                                                                 \
        if (fb == 1024) launch_variant<M, 1024>(a, (int)grid, s);        \
        else if (fb == 512) launch_variant<M, 512>(a, (int)grid, s);     \
        else launch_variant<M, 256>(a, (int)grid, s);                    \
    } while (0)
    if (hot) {
        int64_t g2 = (int64_t)d.num_cus * env_int("VNM_FILTER_WGS_PER_CU", 1);
        if (g2 > a.ntiles) g2 = a.ntiles;
        filter_hot_kernel<<<(int)g2, FH_THREADS, 0, s>>>(a);
    } else
    switch (mode) {
        case CMP_F64: VNM_LAUNCH(CMP_F64); break;
        case MODE_MASK: VNM_LAUNCH(MODE_MASK); break;
        default: VNM_LAUNCH(CMP_I64); break;  // generic pred_eval path
    }
#undef VNM_LAUNCH
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

int64_t vnm_filter_scratch_bytes(int64_t length) { return ((length + 4096) / 4096 + 9) * 8; }

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
