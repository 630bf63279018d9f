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

#ifndef VNM_F16W
#define VNM_F16W 4
#endif
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
    int reuse_idx;   // 0 when payload column 0 IS the predicate column (its values stay in registers; the host moves
                     // such a column to the front), else -1
    vnm_dcol payload[F_MAX_PAYLOAD];
    void* out_values[F_MAX_PAYLOAD];
    uint8_t* out_valid[F_MAX_PAYLOAD];
    int64_t length;     // logical rows
    int64_t phys_base;  // first physical element index covered by tile 0 (even)
    int64_t ntiles;
    unsigned long long* ctl;  // [0] ticket, [1] total, [2..4] look-back statistics, [5] a look-back gave up
    unsigned long long* status;
    unsigned long long* gstatus;  // one word per group of 64 tiles (two-level look-back)
    int lb_sleep;  // look-back back-off between polls, in units of s_sleep(1)
    int spin_limit;  // polls after which a look-back gives up (ctl[5] = 1, the host repeats the filter with the cooperative
                     // launch); 0 = never (cooperative launch: every predecessor is resident)
    int debug;  // timing experiments only: 1 = skip look-back (outputs are wrong)
};

// number of set bits of `m` in lanes below this one (v_mbcnt: no lane mask to keep in registers)
__device__ __forceinline__ uint32_t lanes_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ uint64_t lanemask_lt() {
    uint32_t lane = __lane_id();
    return lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
}

__device__ __forceinline__ void lb_backoff(int n) {
    for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(1);
}

// Two-level decoupled look-back.  With one level the inclusive-prefix frontier advances by at most one
// window (64-256 tiles) per status round trip (~2 us under load): 128 tiles/us, i.e. >= 1.9 ms for the 244k
// tiles of 1e9 rows.  Tiles are grouped by 64: a tile first sums the aggregates of its predecessors inside
// its own group; the last tile of a group publishes the group aggregate as soon as those are visible
// (independent of any prefix), and prefixes then travel over group words, 64 groups = 4096 tiles per round.
// Returns the exclusive prefix of `tile`; `total` is the tile's own survivor count.
// spin_limit > 0: give up (return -1) after that many fruitless polls.
__device__ __forceinline__ int64_t lookback2(unsigned long long* status, unsigned long long* gstatus, int64_t tile,
                                             int lane, uint32_t total, uint64_t& rounds, int nsleep, int spin_limit = 0) {
    const int64_t g = tile >> 6;
    const int r = (int)(tile & 63);
    int64_t local = 0;
    bool have_inc = false;
    int spins = 0;
#define VNM_LB_WAIT()                                                   \
    do {                                                                \
        if (spin_limit > 0 && ++spins > spin_limit) return -1;          \
        lb_backoff(nsleep);                                             \
    } while (0)
    if (r > 0) {
        for (;;) {
            rounds++;
            uint64_t s = lane < r ? __hip_atomic_load(&status[tile - 1 - lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            uint64_t flag = s >> 62;
            uint64_t incl_mask = __ballot(lane < r && flag == 2);
            uint64_t zero_mask = __ballot(lane < r && flag == 0);
            uint64_t val = lane < r ? (s & ST_VAL) : 0;
            if (incl_mask) {
                int first = __ffsll((unsigned long long)incl_mask) - 1;
                uint64_t need = first == 0 ? 0ULL : (~0ULL >> (64 - first));
                if (zero_mask & need) { VNM_LB_WAIT(); continue; }
                if (lane > first) val = 0;
                have_inc = true;
            } else if (zero_mask) {
                VNM_LB_WAIT();
                continue;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) val += __shfl_xor(val, d);
            local = (int64_t)val;
            break;
        }
    }
    if (have_inc) return local;
    if (r == 63 && lane == 0)  // the whole group is known: publish its aggregate before chasing the prefix
        __hip_atomic_store(&gstatus[g], (g > 0 ? ST_AGG : ST_INC) | (uint64_t)(local + total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    int64_t gp = 0;
    int64_t look = g - 1;
    while (look >= 0) {
        rounds++;
        int64_t idx = look - lane;
        uint64_t s = idx >= 0 ? __hip_atomic_load(&gstatus[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ST_INC;
        uint64_t flag = s >> 62;
        uint64_t incl_mask = __ballot(flag == 2);
        uint64_t zero_mask = __ballot(flag == 0);
        uint64_t val = s & ST_VAL;
        if (incl_mask) {
            int first = __ffsll((unsigned long long)incl_mask) - 1;
            uint64_t need = first == 0 ? 0ULL : (~0ULL >> (64 - first));
            if (zero_mask & need) { VNM_LB_WAIT(); continue; }
            if (lane > first) val = 0;
        } else if (zero_mask) {
            VNM_LB_WAIT();
            continue;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) val += __shfl_xor(val, d);
        gp += (int64_t)val;
        if (incl_mask) break;
        look -= 64;
    }
    return gp + local;
#undef VNM_LB_WAIT
}

// =======================================================================================================
// Filter: tiles of FB x 2 x CH rows, several workgroups per CU -- while one waits for its look-back the others
// stream.  No ticket atomics (a single ticket word saturates at ~90 returning atomics/us, and see
// filter_tile_kernel for what tickets do to the look-back chain).
//   HOT: float64 predicate column without validity whose survivors are the only output (BASELINE
//   configs[1]): 16-byte loads, values stay in registers.  Otherwise predicates go through the generic
//   per-element evaluator and payload columns are gathered after the tile base is known.
// Within a wave the survivors of one chunk land on a contiguous output range, so the store instructions of
// a chunk together cover whole cache lines.
// =======================================================================================================
// PAYLOOP: further payload columns are gathered after the tile base is known (not compiled into the configs[1]
// shape, whose register budget is tight)
template <int MODE, int FB, int CH, bool HOT, bool STATS, bool PAYLOOP>
__device__ __forceinline__ void filter_tile(const FilterArgs& a, const int64_t tile) {
    constexpr int TILE = FB * 2 * CH;
    constexpr int NW = FB / 64;
    constexpr int NSEG = CH * NW;  // (chunk, wave) segments
    static_assert(NSEG <= 64 && CH <= 16, "one segment per lane of wave 0; ranks are packed 8 bits per chunk");
    // declared here, not passed in: a generic pointer would turn every LDS access into a flat_ instruction
    __shared__ uint32_t s_cnt[NSEG];
    __shared__ uint32_t s_excl[NSEG];
    __shared__ int64_t s_base;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // physical element index of this lane's first element in chunk 0; logical row = phys - pred.offset
    const int64_t pb = a.phys_base + tile * TILE + 2 * tid;
    const int64_t first = a.phys_base + tile * TILE - a.pred.offset;
    const bool full = first >= 0 && first + TILE <= a.length;

    // HOT: a float64 predicate column (MODE = CMP_F64) or, round 5, an int64 one against an integer literal (MODE = CMP_I64: timestamps,
    // ids, counts -- they took the generic per-row path at 2.5 x the time): raw 64-bit values, 16-byte loads, kept in registers
    uint64_t v0[CH], v1[CH];
    if (HOT) {
        const uint64_t* vals = (const uint64_t*)a.pred.values;
        if (full) {
            const ulonglong2* src = (const ulonglong2*)(vals + pb);
#pragma unroll
            for (int j = 0; j < CH; j++) {
                ulonglong2 t = src[j * FB];
                v0[j] = t.x; v1[j] = t.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < CH; j++) {
                int64_t p0 = pb + (int64_t)j * (2 * FB);
                int64_t r0 = p0 - a.pred.offset;
                v0[j] = (r0 >= 0 && r0 < a.length) ? vals[p0] : 0ULL;
                v1[j] = (r0 + 1 >= 0 && r0 + 1 < a.length) ? vals[p0 + 1] : 0ULL;
            }
        }
    }
    uint32_t flags = 0, rank[(CH + 3) / 4] = {};  // 2 flag bits and an 8-bit in-wave rank per chunk
#pragma unroll
    for (int j = 0; j < CH; j++) {
        const int64_t r0 = pb - a.pred.offset + (int64_t)j * (2 * FB), r1 = r0 + 1;
        const bool in0 = full || (r0 >= 0 && r0 < a.length), in1 = full || (r1 >= 0 && r1 < a.length);
        bool f0 = false, f1 = false;
        if (MODE == MODE_MASK) {
            if (in0) f0 = a.mask_valid && !a.mask_valid[r0] ? true : a.mask[r0] != 0;
            if (in1) f1 = a.mask_valid && !a.mask_valid[r1] ? true : a.mask[r1] != 0;
        } else if (HOT && MODE == CMP_I64) {
            f0 = in0 && cmp_apply<int64_t>(a.p.op, (int64_t)v0[j], a.p.ival);
            f1 = in1 && cmp_apply<int64_t>(a.p.op, (int64_t)v1[j], a.p.ival);
        } else if (HOT) {
            f0 = in0 && cmp_apply<double>(a.p.op, __longlong_as_double((long long)v0[j]), a.p.dval);
            f1 = in1 && cmp_apply<double>(a.p.op, __longlong_as_double((long long)v1[j]), a.p.dval);
        } else {
            f0 = in0 && pred_eval(a.p, a.pred, r0);
            f1 = in1 && pred_eval(a.p, a.pred, r1);
        }
        uint64_t b0 = __ballot(f0), b1 = __ballot(f1);
        rank[j >> 2] |= (lanes_below(b0) + lanes_below(b1)) << (8 * (j & 3));
        flags |= (f0 ? 1u : 0u) << (2 * j) | (f1 ? 1u : 0u) << (2 * j + 1);
        if (lane == 0) s_cnt[j * NW + wave] = __popcll(b0) + __popcll(b1);
    }
    __syncthreads();
    // ---- wave 0: scan the segment counts, publish the tile aggregate, look back for the tile base ----
    if (wave == 0) {
        uint32_t c = lane < NSEG ? s_cnt[lane] : 0;
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane < NSEG) s_excl[lane] = inc - c;
        const uint32_t total = __shfl(inc, 63);
        int64_t excl = 0;
        bool aborted = false;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&a.status[0], ST_INC | (uint64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (a.debug & 1) {
            excl = tile * (TILE / 2);
        } else {
            if (lane == 0) __hip_atomic_store(&a.status[tile], ST_AGG | (uint64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint64_t rounds = 0;
            uint64_t t0 = STATS ? __builtin_readcyclecounter() : 0;
            excl = lookback2(a.status, a.gstatus, tile, lane, total, rounds, a.lb_sleep, a.spin_limit);
            if (excl < 0) {   // gave up (plain launch and a predecessor that never came): everything later still terminates
                if (lane == 0) a.ctl[5] = 1;
                excl = 0;
                aborted = true;
            }
            if (lane == 0) {
                __hip_atomic_store(&a.status[tile], ST_INC | (uint64_t)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((tile & 63) == 63)
                    __hip_atomic_store(&a.gstatus[tile >> 6], ST_INC | (uint64_t)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (STATS && lane == 0) {
                atomicAdd(a.ctl + 2, (unsigned long long)rounds);
                atomicAdd(a.ctl + 3, (unsigned long long)(__builtin_readcyclecounter() - t0));
                atomicAdd(a.ctl + 4, 1ULL);
            }
        }
        if (lane == 0) {
            s_base = aborted ? -1 : excl;
            if (tile == a.ntiles - 1) a.ctl[1] = (unsigned long long)(excl + total);
        }
    }
    __syncthreads();
    if (a.debug & 4) return;
    const int64_t base = s_base;
    if (base < 0) return;   // no output position: the host repeats the whole filter
    if (HOT) {
        // one payload column is the predicate column itself: its values are still in registers
        if (a.reuse_idx == 0) {
            uint64_t* out = (uint64_t*)a.out_values[0] + base;
#pragma unroll
            for (int j = 0; j < CH; j++) {
                uint32_t fj = (flags >> (2 * j)) & 3u;
                uint32_t pos = s_excl[j * NW + wave] + ((rank[j >> 2] >> (8 * (j & 3))) & 0xffu);
                if (fj & 1u) out[pos++] = v0[j];
                if (fj & 2u) out[pos] = v1[j];
            }
        }
        if (!PAYLOOP || a.n_payload <= (a.reuse_idx == 0 ? 1 : 0)) return;
    }
    for (int k = (HOT && a.reuse_idx == 0) ? 1 : 0; k < a.n_payload; k++) {
        const vnm_dcol& c = a.payload[k];
        const int w = type_width(c.type);
        if (w == 8 && !c.validity && !a.out_valid[k] && full && ((c.offset + (pb - a.pred.offset)) & 1) == 0) {
            // 8-byte column without NULLs: the whole tile is read with unconditional 16-byte requests (all of them
            // in flight together), only the stores are conditional
            const uint64_t* p = (const uint64_t*)c.values + c.offset + (pb - a.pred.offset);
            uint64_t* out = (uint64_t*)a.out_values[k] + base;
            ulonglong2 t[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) t[j] = *(const ulonglong2*)(p + (int64_t)j * (2 * FB));
#pragma unroll
            for (int j = 0; j < CH; j++) {
                uint32_t fj = (flags >> (2 * j)) & 3u;
                uint32_t pos = s_excl[j * NW + wave] + ((rank[j >> 2] >> (8 * (j & 3))) & 0xffu);
                if (fj & 1u) out[pos++] = t[j].x;
                if (fj & 2u) out[pos] = t[j].y;
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
            uint32_t fj = (flags >> (2 * j)) & 3u;
            if (!fj) continue;
            int64_t r0 = pb - a.pred.offset + (int64_t)j * (2 * FB);
            int64_t pos = base + s_excl[j * NW + wave] + ((rank[j >> 2] >> (8 * (j & 3))) & 0xffu);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                if (!(fj & (1u << e))) continue;
                uint64_t bits = col_raw_bits(c, r0 + e);
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
}

// Persistent workgroups: tile = blockIdx.x, blockIdx.x + gridDim.x, ..., launched COOPERATIVELY, i.e. the runtime
// only accepts the grid if every workgroup is co-resident.  A look-back therefore never waits on a workgroup that
// has not started, whatever the dispatch order (HIP promises none), and the workgroups advance in lockstep rounds:
// the predecessors of a tile are in flight at the same time as the tile.  Handing tiles out by an atomic ticket
// instead was measured at 5.8-6.7 ms: a ticket taken ahead of time reserves a tile whose owner is still busy, and
// every later tile's look-back waits for it; a ticket taken just in time exposes a ~2 us returning atomic per tile.
// (occupancy target: two 1024-thread workgroups per CU need <= 64 VGPRs AND <= 100 SGPRs on gfx9-family parts)
template <int MODE, int FB, int CH, bool HOT, bool STATS, bool PAYLOOP>
__global__ __launch_bounds__(FB) __attribute__((amdgpu_waves_per_eu(CH == 4 ? 8 : (CH == 16 ? VNM_F16W : 5), 8)))
void filter_tile_kernel(FilterArgs a) {
    for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        filter_tile<MODE, FB, CH, HOT, STATS, PAYLOOP>(a, tile);
        __syncthreads();
    }
}

// ONE workgroup per tile, plain launch: 5-8 % faster (the dispatcher replaces a finished workgroup at once instead of the
// persistent ones walking in rounds).  Its look-backs rely on workgroups being dispatched in index order -- true of the
// hardware dispatcher, not a promise of HIP -- so they are bounded (FilterArgs::spin_limit): should a predecessor ever fail to
// show up, the waiting tiles give up one after the other, the kernel ends, and the host repeats the filter cooperatively.
template <int MODE, int FB, int CH, bool HOT, bool STATS, bool PAYLOOP>
__global__ __launch_bounds__(FB) __attribute__((amdgpu_waves_per_eu(CH == 4 ? 8 : (CH == 16 ? VNM_F16W : 5), 8)))
void filter_tile_flat_kernel(FilterArgs a) {
    filter_tile<MODE, FB, CH, HOT, STATS, PAYLOOP>(a, (int64_t)blockIdx.x);
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

// tuning knobs (defaults chosen by measurement, profiles/tuning_log_r01.md)
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

static int launch_filter(FilterArgs& a, int mode, int64_t* out_count, hipStream_t s) {
    // hot_pred: float64 predicate column without NULLs (16-byte loads, values kept in registers);
    // hot: ... and its survivors are the only output (BASELINE configs[1])
    const bool hot_i64 = mode == CMP_I64 && a.pred.type == VNM_I64 && !a.pred.validity && !(a.reuse_idx == 0 && a.out_valid[0]) &&
                         env_int("VNM_FILTER_NO_HOT_I64", 0) == 0;
    const bool hot_pred = (mode == CMP_F64 && a.pred.type == VNM_F64 && !a.pred.validity &&
                           !(a.reuse_idx == 0 && a.out_valid[0])) || hot_i64;
    const bool hot = hot_pred && (a.n_payload == 0 || (a.n_payload == 1 && a.reuse_idx == 0));
    // hot: 256 threads x 32 rows (8192-row tiles, four workgroups = four tiles in flight per CU: 2.71 ms; 1024 x 8
    // rows, two per CU: 2.94); other shapes: 512 threads x 16 rows
    const int fbe = env_int("VNM_FILTER_THREADS", 256);
    const int fb = hot ? (fbe >= 1024 ? 1024 : (fbe >= 512 ? 512 : 256)) : 512;
    const int ch = hot && fb == 1024 ? 4 : (hot && fb == 256 ? 16 : 8);
    const int tile_rows = fb * 2 * ch;
    a.phys_base = a.pred.values ? (a.pred.offset & ~1LL) : 0;
    int64_t span = (a.pred.values ? a.pred.offset : 0) + a.length - a.phys_base;
    a.ntiles = (span + tile_rows - 1) / tile_rows;
    if (a.length == 0 || a.ntiles == 0) {
        *out_count = 0;
        return 0;
    }
    if (a.ntiles > 0x7fffffffLL) return set_error("filter: batch too large");
    const int64_t ngroups = (a.ntiles >> 6) + 1;
    size_t sbytes = (size_t)(a.ntiles + ngroups + 8) * 8;
    unsigned long long* scratch = (unsigned long long*)pool_alloc(sbytes);
    if (!scratch) return 1;
    a.ctl = scratch;
    a.status = scratch + 8;
    a.gstatus = scratch + 8 + a.ntiles;
    a.debug = env_int("VNM_FILTER_DEBUG", 0);
    a.lb_sleep = env_int("VNM_FILTER_SLEEP", 16);
    // Two launch schemes (filter_tile_kernel / filter_tile_flat_kernel).  The plain one-workgroup-per-tile launch goes first;
    // if one of its bounded look-backs gave up (ctl[5]) the filter is repeated with persistent workgroups launched
    // cooperatively, which cannot wait on a workgroup that is not resident.  VNM_FILTER_PERSIST=1 skips the first attempt.
    unsigned long long total = 0;
    for (int attempt = env_int("VNM_FILTER_PERSIST", 0) != 0 ? 1 : 0; attempt < 2; attempt++) {
        const bool persist = attempt == 1;
        a.spin_limit = persist ? 0 : env_int("VNM_FILTER_SPIN_LIMIT", 200000);   // ~0.5 us per poll
        VNM_HIP(hipMemsetAsync(scratch, 0, sbytes, s));
        auto launch = [&](auto kernel, auto flat_kernel, int threads) -> int {
            if (!persist) {
                flat_kernel<<<(int)a.ntiles, threads, 0, s>>>(a);
                return 0;
            }
            int per_cu = 0;
            VNM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0));
            if (per_cu < 1) return set_error("filter: kernel does not fit a CU");
            int64_t grid = (int64_t)per_cu * device_info().num_cus;
            if (grid > a.ntiles) grid = a.ntiles;
            void* params[] = {(void*)&a};
            VNM_HIP(hipLaunchCooperativeKernel((const void*)kernel, dim3((int)grid), dim3(threads), params, 0, s));
            return 0;
        };
#define VNM_FL(...) launch(filter_tile_kernel<__VA_ARGS__>, filter_tile_flat_kernel<__VA_ARGS__>,
        {
            // (the repeat after a look-back that gave up is a span of its own: `vnm_profile_query("filter_retry")` counts how often
            //  a filter paid twice -- the first attempt's output positions were garbage and are overwritten here)
            KernelTimer timer(persist && env_int("VNM_FILTER_PERSIST", 0) == 0 ? "filter_retry" : "filter_kernel", s);
            int rc;
            if (hot && hot_i64) {
                if (fb == 256) rc = VNM_FL(CMP_I64, 256, 16, true, false, false) 256);
                else rc = VNM_FL(CMP_I64, 512, 8, true, false, false) 512);
            } else if (hot) {
                if (fb == 1024 && (a.debug & 8)) rc = VNM_FL(CMP_F64, 1024, 4, true, true, false) 1024);
                else if (fb == 1024) rc = VNM_FL(CMP_F64, 1024, 4, true, false, false) 1024);
                else if (fb == 256) rc = VNM_FL(CMP_F64, 256, 16, true, false, false) 256);
                else rc = VNM_FL(CMP_F64, 512, 8, true, false, false) 512);
            } else if (hot_pred && hot_i64) {
                rc = VNM_FL(CMP_I64, 512, 8, true, false, true) 512);  // int64 predicate, several payload columns
            } else if (hot_pred) {
                rc = VNM_FL(CMP_F64, 512, 8, true, false, true) 512);  // float64 predicate, several payload columns
            } else if (mode == MODE_MASK) {
                rc = VNM_FL(MODE_MASK, 512, 8, false, false, true) 512);
            } else {
                rc = VNM_FL(CMP_I64, 512, 8, false, false, true) 512);  // generic pred_eval path
            }
            if (rc) { pool_free(scratch); return rc; }
        }
#undef VNM_FL
        VNM_HIP(hipGetLastError());
        unsigned long long res[6] = {0, 0, 0, 0, 0, 0};
        VNM_HIP(hipMemcpyAsync(res, scratch, sizeof(res), hipMemcpyDeviceToHost, s));
        VNM_HIP(hipStreamSynchronize(s));
        total = res[1];
        if (!res[5]) break;   // (a cooperative launch never sets it)
    }
    if (a.debug & 8) {
        unsigned long long st[3] = {0, 0, 0};
        VNM_HIP(hipMemcpy(st, scratch + 2, 24, hipMemcpyDeviceToHost));
        fprintf(stderr, "[vnm filter] look-back: tiles %llu rounds/tile %.2f cycles/tile %.0f\n", st[2],
                st[2] ? (double)st[0] / st[2] : 0.0, st[2] ? (double)st[1] / st[2] : 0.0);
    }
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
    a.reuse_idx = -1;
    for (int k = 0; k < n_payload && a.reuse_idx < 0; k++)
        if (payload[k].values == pred->values && payload[k].offset == pred->offset && payload[k].type == pred->type &&
            payload[k].validity == pred->validity)
            a.reuse_idx = k;
    if (a.reuse_idx > 0) {  // the kernel only knows "payload 0 is the predicate column": the order of the outputs is free
        std::swap(a.payload[0], a.payload[a.reuse_idx]);
        std::swap(a.out_values[0], a.out_values[a.reuse_idx]);
        std::swap(a.out_valid[0], a.out_valid[a.reuse_idx]);
        a.reuse_idx = 0;
    }
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
    a.reuse_idx = -1;
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
