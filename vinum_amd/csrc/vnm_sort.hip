// Sort / top-K for gfx950.
//
// Replaces Sort::Sorted (vinum_cpp/src/operators/sort/sort.cpp:15-63): arrow::compute::SortIndices over
// SortOptions{(column, Ascending|Descending)...} followed by compute::Take.  Semantics pinned by the golden
// vectors generated through the real reference (tests/golden/sort_*.arrow): stable; per key, values first,
// then NaN, then NULL -- for BOTH directions.
//
//   * every key column is turned into an order-preserving unsigned 64-bit code (DESC = complemented code)
//     plus a 2-bit class (0 value, 1 NaN, 2 NULL);
//   * full sort: stable LSD radix sort of (code, row id), 8 bits per pass, key columns from last to first: the encode kernel
//     also produces all eight digit histograms, each pass is ONE kernel (chained scan over the tiles, "onesweep"), passes
//     whose digit is identical for every row are skipped, the first pass makes up the identity row ids and the last one
//     writes int64 row ids straight into the caller's buffer;
//   * LIMIT K on a single key: a threshold taken from a sorted sample selects ~K..4K candidates in one scan
//     of the column (8 B/row), only the candidates are sorted -- identical first K rows (ties keep row
//     order) as the reference's full sort + SliceOperator (vinum/core/algebra.py:229-247).
// Roofline: HBM.  top-K: 8*N read; full sort: 8 B/row read + 8 written by the encode, 12 B/row read + 12 B/row written per
// executed pass (8 + 12 for the first, 12 + 8 for the last).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "vnm_common.hpp"

namespace vnm {


// ---- key encoding ---------------------------------------------------------------------------------------
// negzero (optional): set when the row holds -0.0 (its code is that of +0.0, so the value cannot be rebuilt from the code)
__device__ __forceinline__ void encode_key(const vnm_dcol& c, int64_t row, int desc, uint64_t* code, uint32_t* cls, bool* negzero = nullptr) {
    if (!col_valid(c, row)) { *code = 0; *cls = 2; return; }
    uint64_t e;
    if (type_is_float(c.type)) {
        double d = col_f64(c, row);
        if (d != d) { *code = 0; *cls = 1; return; }
        if (negzero && d == 0.0 && __double_as_longlong(d) != 0) *negzero = true;
        if (d == 0.0) d = 0.0;  // -0.0 and +0.0 compare equal in Arrow's sort: ties keep row order
        e = enc_f64(d);
    } else if (type_is_unsigned(c.type)) {
        e = (uint64_t)col_i64(c, row);
    } else {
        e = enc_i64(col_i64(c, row));
    }
    *code = desc ? ~e : e;
    *cls = 0;
}

// Digit histograms of one wave's codes into the workgroup's LDS table h[8][256] (all 64 lanes hold an element).
// The upper bytes of real keys are often identical across a wave: one add of 64 instead of 64 same-address adds.
__device__ __forceinline__ void hist8_wave(uint32_t* h, uint64_t c, int lane) {
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const uint32_t dg = (uint32_t)(c >> (8 * b)) & 255u;
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)dg);
        if (__ballot(dg != first) == 0) { if (lane == 0) atomicAdd(&h[b * 256 + first], 64u); }
        else atomicAdd(&h[b * 256 + dg], 1u);
    }
}

// code[i] (and cls[i] when cls != NULL) for row idx[i] (idx == NULL: identity); class_codes: code[i] = the row's class
// (0 value, 1 NaN, 2 NULL) instead -- the input of the class pass
// any_cls (optional): bit 0 set when some row is NaN or NULL (class != 0) -- lets the caller skip the class pass; bit 1 when some
// row is -0.0
// ghist (optional): the eight digit histograms of the codes are accumulated on the way (radix_sort_codes then needs no
// histogram pass of its own: one read of the codes less)
__global__ __launch_bounds__(256) void sort_encode_kernel(vnm_dcol c, int desc, const uint32_t* idx, int64_t n, uint64_t* code, uint8_t* cls,
                                                          unsigned long long* any_cls, unsigned long long* ghist, int class_codes) {
    __shared__ uint32_t h[8 * 256];
    if (ghist) {
        for (int i = threadIdx.x; i < 8 * 256; i += 256) h[i] = 0;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nfull = n & ~63LL;
    bool special = false, negz = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t e; uint32_t k;
        encode_key(c, idx ? (int64_t)idx[i] : i, desc, &e, &k, &negz);
        if (class_codes) e = k;
        code[i] = e;
        if (cls) cls[i] = (uint8_t)k;
        special = special || k != 0;
        if (ghist) {
            if (i < nfull) hist8_wave(h, e, lane);   // i < nfull is uniform across the wave (64-aligned chunks)
            else {
#pragma unroll
                for (int b = 0; b < 8; b++) atomicAdd(&h[b * 256 + ((uint32_t)(e >> (8 * b)) & 255u)], 1u);
            }
        }
    }
    if (any_cls && __ballot(special) && lane == 0) atomicOr(any_cls, 1ULL);
    if (any_cls && __ballot(negz) && lane == 0) atomicOr(any_cls, 2ULL);   // bit 1: a -0.0 (the codes do not carry its sign)
    if (ghist) {
        __syncthreads();
        for (int i = threadIdx.x; i < 8 * 256; i += 256)
            if (h[i]) atomicAdd(&ghist[i], (unsigned long long)h[i]);
    }
}

__global__ void sort_iota_kernel(uint32_t* idx, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) idx[i] = (uint32_t)i;
}
// a 4-byte sort key as the 8-byte key of the same order: float32 -> float64 (exact), int32 / uint32 sign- / zero-extended
__global__ __launch_bounds__(256) void sort_widen_key_kernel(const void* src, int type, int64_t first, int64_t n, uint64_t* dst) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int64_t k = first + i;
        uint64_t v;
        if (type == VNM_F32) v = (uint64_t)__double_as_longlong((double)((const float*)src)[k]);
        else if (type == VNM_I32) v = (uint64_t)(int64_t)((const int32_t*)src)[k];
        else v = ((const uint32_t*)src)[k];
        dst[i] = v;
    }
}
__global__ void sort_widen_kernel(const uint32_t* idx, int64_t n, int64_t* out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (int64_t)idx[i];
}

// ---- onesweep: one histogram pre-pass for all eight digits, then ONE kernel per executed pass ------------------------
// r01 ran three kernels per pass (per-block histogram: a full extra read of the codes, scan, scatter): 8 x 2.2 ms of the
// 83 ms went into re-reading the keys for histograms.  The digit histograms of the WHOLE array do not depend on the order
// of the elements, so a single pre-pass computes all eight (radix_hist_all_kernel), a 256-thread kernel turns them into
// exclusive digit bases, and each scatter pass finds its tile's offset inside every digit with a chained scan over the
// tiles that ran before it (decoupled look-back, one status word per tile and digit: tag | prefix-flag | count).  Tiles are
// handed out by an atomic ticket, so tile t - 1 is always resident or finished when tile t looks back.
constexpr int OS_BLOCK = 512;                    // 8 waves; two workgroups per CU (74 KB of LDS each)
constexpr int OS_SUB = 16;                       // elements per lane and tile
constexpr int OS_WAVES = OS_BLOCK / 64;
constexpr int OS_TILE = OS_BLOCK * OS_SUB;       // 8192 elements: ~32 per digit -> 256-B code runs, 128-B row-id runs
constexpr unsigned long long OS_PREFIX = 1ULL << 59;
constexpr unsigned long long OS_VAL_MASK = OS_PREFIX - 1;
constexpr unsigned long long OS_TAG_MASK = 0xFULL << 60;
constexpr size_t OS_LDS_BYTES = (size_t)OS_TILE * 8 + (size_t)OS_WAVES * 256 * 4 + 2 * 256 * 4;

__global__ __launch_bounds__(256) void radix_hist_all_kernel(const uint64_t* code, int64_t n, unsigned long long* ghist) {
    __shared__ uint32_t h[8 * 256];
    for (int i = threadIdx.x; i < 8 * 256; i += 256) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t nfull = n & ~63LL;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nfull; i += stride) hist8_wave(h, code[i], lane);   // whole waves only
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - nfull)) {
        const uint64_t c = code[nfull + threadIdx.x];
#pragma unroll
        for (int b = 0; b < 8; b++) atomicAdd(&h[b * 256 + ((uint32_t)(c >> (8 * b)) & 255u)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * 256; i += 256)
        if (h[i]) atomicAdd(&ghist[i], (unsigned long long)h[i]);
}

// ghist[8][256] -> exclusive scan per digit position (the first output slot of every digit value)
__global__ __launch_bounds__(256) void radix_bases_kernel(const unsigned long long* ghist, unsigned long long* bases) {
    __shared__ unsigned long long wt[4];
    const int d = threadIdx.x, lane = d & 63, wave = d >> 6;
    for (int b = 0; b < 8; b++) {
        const unsigned long long x = ghist[b * 256 + d];
        unsigned long long inc = x;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) { const unsigned long long o = __shfl_up(inc, k); if (lane >= k) inc += o; }
        if (lane == 63) wt[wave] = inc;
        __syncthreads();
        unsigned long long add = 0;
        for (int w = 0; w < wave; w++) add += wt[w];
        bases[b * 256 + d] = add + inc - x;
        __syncthreads();
    }
}

struct OsArgs {
    const uint64_t* code;
    const uint32_t* val;
    int64_t n, ntiles;
    int shift;
    const unsigned long long* base;   // [256] exclusive digit bases of this pass
    unsigned long long* status;       // [ntiles][256]
    unsigned int* ticket;
    unsigned long long tag;           // (pass + 1) << 60: status words of other passes read as "not published yet"
    uint64_t* code_out;
    uint32_t* val_out;
    int64_t* idx_out;                 // last pass of the last key: row ids widened to int64 straight into the caller's buffer
    uint64_t* key_out;                // ... and (optional) the sorted values of key 0 rebuilt from their codes: saves the caller a gather
    int key_type, key_desc;
};

// Tile shapes tried on 1e9 fp64 keys (ms per pass): 512 x 16 (this one) 6.1; 1024 threads x 8 9.4 (spills at 64 VGPRs);
// 512 x 8 = 4096-element tiles, three workgroups per CU 8.0 (runs half as long).
__global__ __launch_bounds__(OS_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void onesweep_kernel(OsArgs a) {
    extern __shared__ uint64_t os_lds[];
    uint64_t* stage = os_lds;                                  // [OS_TILE] codes by digit; reused for the row ids
    uint32_t* stage32 = (uint32_t*)os_lds;
    uint32_t* wcnt = (uint32_t*)(os_lds + OS_TILE);            // [OS_WAVES][256] per-wave digit counts -> bases
    uint32_t* off = wcnt + OS_WAVES * 256;                     // [256] start of each digit inside the staged tile
    uint32_t* tot = off + 256;                                 // [256] elements of each digit in this tile
    __shared__ unsigned long long run[256];                    // global output position of the tile's first element of each digit
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* mycnt = wcnt + wave * 256;
    for (;;) {
        if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
        for (int i = tid; i < OS_WAVES * 256; i += OS_BLOCK) wcnt[i] = 0;
        __syncthreads();
        const int64_t t = (int64_t)s_tile;
        if (t >= a.ntiles) break;
        const int64_t base = t * OS_TILE;
        const int64_t hi = base + OS_TILE < a.n ? base + OS_TILE : a.n;
        // ---- load + rank inside the wave (wave w owns elements [w * 1024, (w + 1) * 1024) of the tile)
        uint64_t c[OS_SUB];
        uint32_t v[OS_SUB], lp[OS_SUB / 2];   // lp: two 16-bit positions per register
        const int64_t wbase = base + (int64_t)wave * (64 * OS_SUB) + lane;
#pragma unroll
        for (int k = 0; k < OS_SUB; k++) {
            const int64_t i = wbase + k * 64;
            c[k] = i < hi ? a.code[i] : 0;
            v[k] = i < hi ? (a.val ? a.val[i] : (uint32_t)i) : 0;   // val == NULL: the rows are still in their original order
        }
#pragma unroll
        for (int k = 0; k < OS_SUB; k++) {
            const bool in = wbase + k * 64 < hi;
            const uint32_t dg = (uint32_t)(c[k] >> a.shift) & 255u;
            uint64_t m = __ballot(in);
#pragma unroll
            for (int bit = 0; bit < 8; bit++) {
                const uint64_t bb = __ballot((dg >> bit) & 1u);
                m &= ((dg >> bit) & 1u) ? bb : ~bb;
            }
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            uint32_t lpos = 0;
            if (in) {
                lpos = mycnt[dg] + before;                    // LDS ops of one wave execute in order
                if (before == 0) mycnt[dg] += (uint32_t)__popcll(m);
            }
            if ((k & 1) == 0) lp[k >> 1] = lpos; else lp[k >> 1] |= lpos << 16;
        }
        __syncthreads();
        // ---- per digit: wave counts -> exclusive bases over the waves, tile total (published at once), digit offsets
        if (tid < 256) {
            uint32_t s = 0;
#pragma unroll
            for (int w = 0; w < OS_WAVES; w++) { const uint32_t x = wcnt[w * 256 + tid]; wcnt[w * 256 + tid] = s; s += x; }
            tot[tid] = s;
            __hip_atomic_store(&a.status[t * 256 + tid], a.tag | (t == 0 ? OS_PREFIX : 0ULL) | (unsigned long long)s, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            uint32_t inc = s;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
            off[tid] = inc - s;
            if (lane == 63) wtot[wave] = inc;
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t add = 0;
            for (int w = 0; w < wave; w++) add += wtot[w];
            off[tid] += add;
        }
        __syncthreads();
        // ---- stage the codes by digit
#pragma unroll
        for (int k = 0; k < OS_SUB; k++) {
            if (wbase + k * 64 < hi) {
                const uint32_t dg = (uint32_t)(c[k] >> a.shift) & 255u;
                const uint32_t pos = ((lp[k >> 1] >> (16 * (k & 1))) & 0xFFFFu) + off[dg] + mycnt[dg];
                lp[k >> 1] = (lp[k >> 1] & (0xFFFFu << (16 * ((k & 1) ^ 1)))) | (pos << (16 * (k & 1)));
                stage[pos] = c[k];
            }
        }
        // ---- chained scan over the earlier tiles (one thread per digit value)
        if (tid < 256) {
            unsigned long long excl = 0;
            if (t > 0) {
                int64_t look = t - 1;
                for (;;) {
                    const unsigned long long sv = __hip_atomic_load(&a.status[look * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((sv & OS_TAG_MASK) != a.tag) { __builtin_amdgcn_s_sleep(1); continue; }
                    excl += sv & OS_VAL_MASK;
                    if (sv & OS_PREFIX) break;
                    look--;
                }
                __hip_atomic_store(&a.status[t * 256 + tid], a.tag | OS_PREFIX | (excl + tot[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            run[tid] = a.base[tid] + excl;
        }
        __syncthreads();
        // ---- copy the codes out in runs; remember each element's digit for the row-id round
        const uint32_t total = (uint32_t)(hi - base);
        uint32_t dgs[OS_SUB / 4];
#pragma unroll
        for (int j = 0; j < OS_SUB; j++) {
            const uint32_t i = (uint32_t)tid + (uint32_t)j * OS_BLOCK;
            uint32_t dg = 0;
            if (i < total) {
                const uint64_t cc = stage[i];
                dg = (uint32_t)(cc >> a.shift) & 255u;
                if (a.code_out) a.code_out[run[dg] + (i - off[dg])] = cc;
                if (a.key_out) {
                    const uint64_t e = a.key_desc ? ~cc : cc;
                    a.key_out[run[dg] + (i - off[dg])] = a.key_type == VNM_F64 ? (uint64_t)__double_as_longlong(dec_f64(e))
                                                       : (a.key_type == VNM_I64 ? (uint64_t)dec_i64(e) : e);
                }
            }
            if ((j & 3) == 0) dgs[j >> 2] = dg; else dgs[j >> 2] |= dg << (8 * (j & 3));
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < OS_SUB; k++)
            if (wbase + k * 64 < hi) stage32[(lp[k >> 1] >> (16 * (k & 1))) & 0xFFFFu] = v[k];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < OS_SUB; j++) {
            const uint32_t i = (uint32_t)tid + (uint32_t)j * OS_BLOCK;
            if (i < total) {
                const uint32_t dg = (dgs[j >> 2] >> (8 * (j & 3))) & 255u;
                const unsigned long long pos = run[dg] + (i - off[dg]);
                if (a.idx_out) a.idx_out[pos] = (int64_t)stage32[i];
                else a.val_out[pos] = stage32[i];
            }
        }
        __syncthreads();
    }
}

struct RadixBufs {
    uint64_t* code[2];
    uint32_t* val[2];
    unsigned long long* red;          // [0..1] unused, [4] caller's flag word, [8..15] tickets (as uint32 pairs)
    unsigned long long* ghist;        // [8][256] digit histograms of the whole array, then [8][256] exclusive bases
    unsigned long long* status;       // [ntiles][256] chained-scan status words
    int cur = 0;
    int64_t ntiles = 0;
    ~RadixBufs();   // gives the blocks back (early returns included)
};
static void radix_free(RadixBufs* r);

static int radix_alloc(RadixBufs* r, int64_t n) {
    r->ntiles = (n + OS_TILE - 1) / OS_TILE;
    for (int k = 0; k < 2; k++) {
        r->code[k] = (uint64_t*)pool_alloc((size_t)(n ? n : 1) * 8);
        r->val[k] = (uint32_t*)pool_alloc((size_t)(n ? n : 1) * 4);
        if (!r->code[k] || !r->val[k]) return 1;
    }
    r->red = (unsigned long long*)pool_alloc(128);
    r->ghist = (unsigned long long*)pool_alloc((size_t)2 * 8 * 256 * 8);
    r->status = (unsigned long long*)pool_alloc((size_t)(r->ntiles ? r->ntiles : 1) * 256 * 8);
    return (r->red && r->ghist && r->status) ? 0 : 1;
}
static void radix_free(RadixBufs* r) {   // idempotent: ~RadixBufs calls it again on every way out of a scope
    for (int k = 0; k < 2; k++) { pool_free(r->code[k]); pool_free(r->val[k]); r->code[k] = nullptr; r->val[k] = nullptr; }
    pool_free(r->red); pool_free(r->ghist); pool_free(r->status);
    r->red = nullptr; r->ghist = nullptr; r->status = nullptr;
}
RadixBufs::~RadixBufs() { radix_free(this); }

// sort (code[cur], val[cur]) stably by the bytes of code that are not constant
// extra (optional): receives r->red[4], a flag word the caller's previous kernel may have set (read back with the
// same synchronisation as the digit histograms)
// idx_out (optional): the caller's int64 row-id buffer.  When this call runs the LAST pass of the whole sort (no class pass
// will follow: !cls_possible or the flag word is 0) that pass writes the widened row ids straight into it and skips the
// code output; *wrote_idx tells the caller.
// hist_ready: r->ghist already holds the digit histograms of the codes (sort_encode_kernel accumulated them)
// ident (optional, in/out): the row ids r->val[r->cur] are the identity and NOT materialised; the first executed pass makes them up
// key_out / key_type / key_desc / wrote_key (optional): see OsArgs::key_out; only together with idx_out, and only when the flag
// word says the key holds no NaN / NULL / -0.0
static int radix_sort_codes(RadixBufs* r, int64_t n, hipStream_t s, unsigned long long* extra = nullptr, int64_t* idx_out = nullptr,
                            bool cls_possible = false, bool* wrote_idx = nullptr, bool hist_ready = false, bool* ident = nullptr,
                            uint64_t* key_out = nullptr, int key_type = 0, int key_desc = 0, bool* wrote_key = nullptr) {
    if (wrote_key) *wrote_key = false;
    if (wrote_idx) *wrote_idx = false;
    if (n <= 1) { if (extra) { VNM_HIP(hipMemcpyAsync(extra, r->red + 4, 8, hipMemcpyDeviceToHost, s)); VNM_HIP(hipStreamSynchronize(s)); } return 0; }
    static bool attr_set = false;
    if (!attr_set) {
        VNM_HIP(hipFuncSetAttribute((const void*)onesweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OS_LDS_BYTES));
        attr_set = true;
    }
    const int cus = device_info().num_cus;
    // ---- histograms of all eight digits in one read of the codes; status words and tickets cleared meanwhile
    unsigned long long hist[8 * 256 + 1];
    if (!hist_ready) VNM_HIP(hipMemsetAsync(r->ghist, 0, (size_t)8 * 256 * 8, s));
    VNM_HIP(hipMemsetAsync(r->red + 8, 0, 64, s));
    const int64_t ntiles = (n + OS_TILE - 1) / OS_TILE;   // <= r->ntiles: callers may sort fewer elements than they allocated for
    VNM_HIP(hipMemsetAsync(r->status, 0, (size_t)ntiles * 256 * 8, s));
    {
        KernelTimer timer("radix_hist", s);
        int g = cus * 8;
        const int64_t need = (n + 255) / 256;
        if (g > need) g = (int)need;
        if (!hist_ready) radix_hist_all_kernel<<<g, 256, 0, s>>>(r->code[r->cur], n, r->ghist);
        radix_bases_kernel<<<1, 256, 0, s>>>(r->ghist, r->ghist + 8 * 256);
    }
    VNM_HIP(hipGetLastError());
    VNM_HIP(hipMemcpyAsync(hist, r->ghist, (size_t)8 * 256 * 8, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipMemcpyAsync(hist + 8 * 256, r->red + 4, 8, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    if (extra) *extra = hist[8 * 256];
    const bool cls_follows = cls_possible && (hist[8 * 256] & 1ULL) != 0;
    int live[8], n_live = 0;
    for (int byte = 0; byte < 8; byte++) {
        bool constant = false;      // one digit value holds every element: nothing to reorder
        for (int d = 0; d < 256 && !constant; d++) constant = hist[byte * 256 + d] == (unsigned long long)n;
        if (!constant) live[n_live++] = byte;
    }
    int g = (int)std::min<int64_t>(ntiles, (int64_t)cus * 2);
    for (int k = 0; k < n_live; k++) {
        const int byte = live[k];
        KernelTimer timer("radix_pass", s);
        OsArgs a{};
        a.code = r->code[r->cur]; a.val = (ident && *ident) ? nullptr : r->val[r->cur]; a.n = n; a.ntiles = ntiles; a.shift = 8 * byte;
        a.base = r->ghist + 8 * 256 + byte * 256;
        a.status = r->status;
        a.ticket = (unsigned int*)(r->red + 8) + byte;
        a.tag = (unsigned long long)(byte + 1) << 60;
        a.code_out = r->code[r->cur ^ 1]; a.val_out = r->val[r->cur ^ 1];
        if (idx_out && !cls_follows && k == n_live - 1) {
            a.idx_out = idx_out; a.code_out = nullptr;
            if (wrote_idx) *wrote_idx = true;
            if (key_out && hist[8 * 256] == 0) {
                a.key_out = key_out; a.key_type = key_type; a.key_desc = key_desc;
                if (wrote_key) *wrote_key = true;
            }
        }
onesweep_kernel<<<g, OS_BLOCK, OS_LDS_BYTES, s>>>(a);
        r->cur ^= 1;
        if (ident) *ident = false;
    }
    VNM_HIP(hipGetLastError());
    return 0;
}


// ---- LIMIT K without the launch-bound tail ------------------------------------------------------------------------------
// r01's top-K query spent 1.35 ms in the selection scan and 1.5 ms in ~130 tiny radix launches (sorting the 2^18-key
// sample to read ONE threshold off it, then sorting a few hundred candidates three times).  Both small problems fit one
// workgroup each:
//   * topk_blockbest_kernel + one workgroup sort: the threshold is the r-th best of 4096 block winners of a 2^22-key sample
//     (a single-workgroup radix SELECT over the whole sample was tried first: 0.8 ms at 2^18 keys -- its LDS histogram
//     atomics all hit the few bins the keys' top bytes share);
//   * topk_small_sort_kernel: bitonic sort of up to 8192 candidates in LDS on the composite key (class, code, row id) --
//     the row id makes it the stable order of the full sort -- and the first K row ids go straight to the output.
// Block-best sampling: the m-key strided sample is cut into TB_BLOCKS blocks and only the BEST key of every block is kept.
// The r-th best of those block winners is a threshold with AT LEAST r rows at or above it (the r winners themselves) and
// about r * n / m in expectation while r << TB_BLOCKS -- the same statistic as the r-th order statistic of the sample,
// found by sorting 4096 keys in one workgroup instead of the whole sample.
constexpr int TB_BLOCKS = 4096;
__global__ __launch_bounds__(256) void topk_blockbest_kernel(vnm_dcol c, int desc, int64_t n, int64_t m, uint64_t* best_code, uint8_t* best_cls) {
    const int64_t per = (m + TB_BLOCKS - 1) / TB_BLOCKS;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < m ? lo + per : m;
    uint64_t bc = ~0ULL;
    uint32_t bk = 255;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const int64_t row = (int64_t)(((__int128)i * n) / m);
        uint64_t e; uint32_t k;
        encode_key(c, row, desc, &e, &k);
        if (k < bk || (k == bk && e < bc)) { bk = k; bc = e; }
    }
    for (int d = 32; d > 0; d >>= 1) {
        const uint64_t oc = __shfl_xor(bc, d);
        const uint32_t ok = __shfl_xor(bk, d);
        if (ok < bk || (ok == bk && oc < bc)) { bk = ok; bc = oc; }
    }
    __shared__ uint64_t sc[4];
    __shared__ uint32_t sk[4];
    if ((threadIdx.x & 63) == 0) { sc[threadIdx.x >> 6] = bc; sk[threadIdx.x >> 6] = bk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) if (sk[w] < bk || (sk[w] == bk && sc[w] < bc)) { bk = sk[w]; bc = sc[w]; }
        best_code[blockIdx.x] = bc;
        best_cls[blockIdx.x] = (uint8_t)bk;
    }
}

constexpr int TS_MAX = 8192;
// thr (optional): instead of row ids, write the (class, code) of the element at sorted position `limit` to thr[0], thr[1]
__global__ __launch_bounds__(1024) void topk_small_sort_kernel(const uint64_t* code, const uint8_t* cls, const uint32_t* rows, int n, int np2,
                                                              int64_t limit, int64_t* out, unsigned long long* thr) {
    extern __shared__ uint64_t ts_lds[];
    uint64_t* k_code = ts_lds;                         // [np2]
    uint32_t* k_row = (uint32_t*)(ts_lds + np2);       // [np2]
    uint8_t* k_cls = (uint8_t*)(k_row + np2);          // [np2]
    const int tid = threadIdx.x;
    for (int i = tid; i < np2; i += 1024) {
        const bool in = i < n;
        k_code[i] = in ? code[i] : ~0ULL;
        k_row[i] = in ? (rows ? rows[i] : (uint32_t)i) : 0xFFFFFFFFu;
        k_cls[i] = in ? cls[i] : (uint8_t)255;         // padding sorts last
    }
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < np2 / 2; t += 1024) {
                const int lo = 2 * t - (t & (stride - 1));   // index with bit `stride` clear
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint8_t ca = k_cls[lo], cb = k_cls[hi];
                const uint64_t xa = k_code[lo], xb = k_code[hi];
                const uint32_t ra = k_row[lo], rb = k_row[hi];
                const bool a_gt_b = ca != cb ? ca > cb : (xa != xb ? xa > xb : ra > rb);
                if (a_gt_b == up) {
                    k_cls[lo] = cb; k_cls[hi] = ca;
                    k_code[lo] = xb; k_code[hi] = xa;
                    k_row[lo] = rb; k_row[hi] = ra;
                }
            }
            __syncthreads();
        }
    }
    if (thr) {
        if (tid == 0) { thr[0] = k_cls[limit]; thr[1] = k_code[limit]; }
        return;
    }
    for (int64_t i = tid; i < limit; i += 1024) out[i] = (int64_t)k_row[i];
}

// ---- top-K candidate selection ----------------------------------------------------------------------------
__global__ void topk_sample_kernel(vnm_dcol c, int desc, int64_t n, int64_t m, uint64_t* code, uint8_t* cls) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        int64_t row = (int64_t)(((__int128)i * n) / m);
        uint64_t e; uint32_t k;
        encode_key(c, row, desc, &e, &k);
        code[i] = e;
        cls[i] = (uint8_t)k;
    }
}

// keep rows whose (class, code) <= (t_cls, t_code); unordered append of (code, class, row).
// Candidates are collected in a per-wave LDS buffer and appended with ONE atomic per 256 candidates: a
// returning atomic per wave-with-survivors saturates the shared counter (~90/us) once K reaches ~1e6.
// Four independent rows per lane keep the column reads in flight.
constexpr int TK_U = 4;
constexpr int TK_BUF = 256;
__global__ __launch_bounds__(256) void topk_select_kernel(vnm_dcol c, int desc, int64_t n, uint32_t t_cls, uint64_t t_code, int64_t cap,
                                                          unsigned long long* count, uint64_t* code, uint8_t* cls, uint32_t* rows) {
    __shared__ uint64_t b_code[4][TK_BUF];
    __shared__ uint32_t b_row[4][TK_BUF];
    __shared__ uint8_t b_cls[4][TK_BUF];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint64_t lt = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
    uint32_t fill = 0;  // wave-uniform
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(count, (unsigned long long)fill);
        base = __shfl(base, 0);
        for (uint32_t j = lane; j < fill; j += 64) {
            int64_t pos = (int64_t)base + j;
            if (pos < cap) { code[pos] = b_code[wave][j]; cls[pos] = b_cls[wave][j]; rows[pos] = b_row[wave][j]; }
        }
        __builtin_amdgcn_wave_barrier();
        fill = 0;
    };
    const int64_t tile_rows = 256 * TK_U;
    const int64_t ntiles = (n + tile_rows - 1) / tile_rows;
    const bool fast = c.type == VNM_F64 && !c.validity && (c.offset & 1) == 0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint64_t e[TK_U];
        uint32_t k[TK_U];
        bool keep[TK_U];
        int64_t rowi[TK_U];
        if (fast && (t + 1) * tile_rows <= n) {
            // float64 without NULLs: a lane owns pairs of adjacent rows, 16 bytes per request
            const double* p = (const double*)c.values + c.offset + t * tile_rows + 2 * threadIdx.x;
#pragma unroll
            for (int u = 0; u < TK_U / 2; u++) {
                const double2 d = *(const double2*)(p + u * 512);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    double x = h ? d.y : d.x;
                    const int q = 2 * u + h;
                    rowi[q] = t * tile_rows + u * 512 + 2 * threadIdx.x + h;
                    if (x != x) { e[q] = 0; k[q] = 1; }
                    else {
                        if (x == 0.0) x = 0.0;  // -0.0 and +0.0 tie (see encode_key)
                        const uint64_t enc = enc_f64(x);
                        e[q] = desc ? ~enc : enc;
                        k[q] = 0;
                    }
                    keep[q] = k[q] < t_cls || (k[q] == t_cls && e[q] <= t_code);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < TK_U; u++) {
                int64_t i = t * tile_rows + u * 256 + threadIdx.x;
                rowi[u] = i;
                e[u] = 0; k[u] = 0; keep[u] = false;
                if (i < n) {
                    encode_key(c, i, desc, &e[u], &k[u]);
                    keep[u] = k[u] < t_cls || (k[u] == t_cls && e[u] <= t_code);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < TK_U; u++) {
            uint64_t b = __ballot(keep[u]);
            if (!b) continue;
            uint32_t cnt = (uint32_t)__popcll(b);
            if (fill + cnt > TK_BUF) flush();
            if (keep[u]) {
                uint32_t slot = fill + (uint32_t)__popcll(b & lt);
                b_code[wave][slot] = e[u];
                b_cls[wave][slot] = (uint8_t)k[u];
                b_row[wave][slot] = (uint32_t)rowi[u];
            }
            fill += cnt;
        }
    }
    if (fill) flush();
}

__global__ void gather_u8_kernel(const uint8_t* src, const uint32_t* idx, int64_t n, uint64_t* out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = src[idx[i]];
}
__global__ void gather_u64_kernel(const uint64_t* src, const uint32_t* idx, int64_t n, uint64_t* out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = src[idx[i]];
}
__global__ void gather_i64_kernel(const int64_t* src, const int64_t* idx, int64_t n, int64_t* out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = src[idx[i]];
}
__global__ void gather_u32_kernel(const uint32_t* src, const uint32_t* idx, int64_t n, uint32_t* out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = src[idx[i]];
}
__global__ void u32_to_code_kernel(const uint32_t* src, int64_t n, uint64_t* out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = src[i];
}

__global__ void take_kernel(vnm_dcol c, const int64_t* idx, int64_t n, void* out, uint8_t* out_valid) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int w = type_width(c.type);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int64_t r = idx[i];
        uint64_t bits = col_raw_bits(c, r);
        switch (w) {
            case 8: ((uint64_t*)out)[i] = bits; break;
            case 4: ((uint32_t*)out)[i] = (uint32_t)bits; break;
            case 2: ((uint16_t*)out)[i] = (uint16_t)bits; break;
            default: ((uint8_t*)out)[i] = (uint8_t)bits; break;
        }
        if (out_valid) out_valid[i] = col_valid(c, r);
    }
}

static int64_t env_sort_i64(const char* name, int64_t dflt) {
    const char* v = getenv(name);
    return v ? atoll(v) : dflt;
}

static int grid_for(int64_t n, int per_cu = 8) {
    int g = device_info().num_cus * per_cu;
    int64_t need = (n + 255) / 256;
    if (need < 1) need = 1;
    return g > need ? (int)need : g;
}

static thread_local bool g_rows_clustered = false;   // set by the sample sorts when they decline rows that arrive clustered (ssort_cluster_kernel)

#include "vnm_sort_sample.inc"
#include "vnm_sort_apx.inc"

// A column that arrives clustered is often simply SORTED already (a time series ordered by its timestamp).  One pass counts the rows
// whose code is below their predecessor's; none: the order asked for is the row order (stable: equal keys keep it), the indices are 0 .. n - 1.
// (out[0]: rows below their predecessor, out[1]: rows above or EQUAL to it -- none of the latter: the column is strictly in the opposite order,
//  and without ties the stable answer is the row numbers backwards)
__global__ __launch_bounds__(256) void sort_inversions_kernel(vnm_dcol key, int desc, int64_t n, unsigned long long* out) {
    unsigned special = 0;
    unsigned int below = 0, not_below = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += stride) {
        const bool b = ss_code_of(key, i, desc, &special) < ss_code_of(key, i - 1, desc, &special);
        below += b ? 1u : 0u; not_below += b ? 0u : 1u;
    }
    for (int o = 32; o > 0; o >>= 1) { below += __shfl_xor(below, o); not_below += __shfl_xor(not_below, o); }
    if ((threadIdx.x & 63) == 0) { if (below) atomicAdd(out, (unsigned long long)below); if (not_below) atomicAdd(out + 1, (unsigned long long)not_below); }
}
__global__ void sort_iota64_kernel(int64_t* idx, int64_t n, int backwards) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) idx[i] = backwards ? n - 1 - i : i;
}

// Sample sort of one 8-byte key without a validity bitmap (see vnm_sort_sample.inc).  0 = done (idx_out written, *wrote_key),
// 2 = not applicable / a bucket outgrew its room (the caller sorts with the LSD passes), 1 = error.
static int sample_sort(const vnm_dcol& key, int desc, int64_t n, int64_t* idx_out, uint64_t* key_out, bool* wrote_key, hipStream_t s) {
    if (wrote_key) *wrote_key = false;
    const int cus = device_info().num_cus;
    // 512 x l2 buckets of ~2100-4200 rows: 2^18 at 1e9 rows, fewer below (2^18 buckets whatever n cost ~4.5 ms of fixed time:
    // the 2^23-key sample and 2^18 nearly empty local-sort workgroups; 1.7e7 rows 5.2 ms against 1.4 with the LSD passes)
    int l2 = 8;
    while (l2 < SS_B && n / ((int64_t)SS_B * l2) > 4200) l2 *= 2;
    l2 = (int)std::min<int64_t>(SS_B, std::max<int64_t>(8, env_sort_i64("VNM_SSORT_L2", l2)));
    const int64_t nb = (int64_t)SS_B * l2;
    // Samples per bucket: a bucket's share of the rows is ~Gamma(k) / k for k samples per bucket, and a bucket beyond the local
    // sort's room (r times the mean) fails the whole attempt -- P ~ exp(-k (r - 1 - ln r)) per bucket.  At 1e9 rows r = 8192 / 3815
    // = 2.15: with k = 32 about one data set in ten lost a bucket among its 2^18 (and paid the LSD sort on top of the wasted
    // passes: 119 ms); k is chosen for P * buckets < 1e-7.
    int64_t m;
    {
        const double mean = (double)n / (double)nb;
        const double r = std::min<double>((double)SS_LOCAL, 2.5 * mean + 128.0) / mean;
        const double need = (std::log((double)nb) + 16.0) / std::max(0.05, r - 1.0 - std::log(r));
        const int64_t k = std::min<int64_t>(128, std::max<int64_t>(32, (int64_t)std::ceil(need)));
        m = std::min<int64_t>(n, nb * env_sort_i64("VNM_SSORT_SAMPLES_PER_BUCKET", k));
    }
    static bool attr_set = false;
    const size_t lds_sc = (size_t)SS_B * SS_CAP * 12;
    if (!attr_set) {
        VNM_HIP(hipFuncSetAttribute((const void*)ssort_scatter_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc));
        VNM_HIP(hipFuncSetAttribute((const void*)ssort_scatter_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc));
        VNM_HIP(hipFuncSetAttribute((const void*)ssort_scatter_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc));
        VNM_HIP(hipFuncSetAttribute((const void*)ssort_scatter_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc));
        VNM_HIP(hipFuncSetAttribute((const void*)ssort_local_kernel<512, 10, 4096, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)SS_SMALL * 12 + 4096 * 4)));
        VNM_HIP(hipFuncSetAttribute((const void*)ssort_local_kernel<1024, 8, 8192, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)SS_LOCAL * 12 + 8192 * 4)));
        VNM_HIP(hipFuncSetAttribute((const void*)onesweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OS_LDS_BYTES));
        attr_set = true;
    }
    PoolScope pool;
    // ---- splitters from a sorted sample; heavy codes = runs of equal splitters
    uint64_t* split = (uint64_t*)pool.take((size_t)nb * 8);
    unsigned long long* flags = (unsigned long long*)pool.take(64);
    uint64_t* heavy = (uint64_t*)pool.take((size_t)SS_MAX_HEAVY * 8);
    unsigned long long* lb = (unsigned long long*)pool.take((size_t)(SS_MAX_HEAVY + 2) * 8 * 2);   // lb[66], then hstart[65] (+ the NULL class)
    unsigned int* hb = (unsigned int*)pool.take((size_t)(SS_MAX_HEAVY + 2) * 4);                   // hb[64], then the heavy counter
    if (!split || !flags || !heavy || !lb || !hb) return 1;
    unsigned long long* hstart = lb + SS_MAX_HEAVY + 2;
    const int has_null = key.validity != nullptr;
    unsigned int* nheavy_d = hb + SS_MAX_HEAVY;
    VNM_HIP(hipMemsetAsync(flags, 0, 64, s));
    VNM_HIP(hipMemsetAsync(nheavy_d, 0, 4, s));
    int nheavy = 0;
    int64_t sample_nulls = 0, total_nulls = 0;
    unsigned long long eq_pairs = 0;
    {
        RadixBufs sr{};
        VNM_TRY(radix_alloc(&sr, m));
        {
            KernelTimer timer("sort_sample", s);
            ssort_sample_kernel<<<grid_for(m), 256, 0, s>>>(key, desc, n, m, sr.code[0], flags + 5);
        }
        sort_iota_kernel<<<grid_for(m), 256, 0, s>>>(sr.val[0], m);
        sr.cur = 0;
        VNM_TRY(radix_sort_codes(&sr, m, s));
        int64_t m_valid = m;     // NULL rows of the sample stand at the end of its sorted order: the splitters come from the rest
        if (has_null) {
            unsigned long long sn = 0, tn = 0;
            ssort_count_nulls_kernel<<<grid_for((n + 7) / 8), 256, 0, s>>>(key.validity, key.offset, n, flags + 6);
            VNM_HIP(hipMemcpyAsync(&sn, flags + 5, 8, hipMemcpyDeviceToHost, s));
            VNM_HIP(hipMemcpyAsync(&tn, flags + 6, 8, hipMemcpyDeviceToHost, s));
            VNM_HIP(hipStreamSynchronize(s));
            sample_nulls = (int64_t)sn;
            total_nulls = (int64_t)tn;
            m_valid = m - sample_nulls;
            if (m_valid < nb * 4) {
                if (getenv("VNM_SORT_TRACE")) fprintf(stderr, "[sort] sample sort declined: %lld of %lld sampled rows are NULL\n", (long long)sample_nulls, (long long)m);
                return 2;
            }
        }
        ssort_splitters_kernel<<<(int)((nb + 255) / 256), 256, 0, s>>>(sr.code[sr.cur], m_valid, nb, split, flags, heavy, nheavy_d);
        VNM_HIP(hipGetLastError());
        // rows that arrive clustered (a sorted column): the ring scatters would crawl -- the LSD passes do not care
        unsigned long long min_span = ~0ULL;
        VNM_HIP(hipMemsetAsync(flags + 7, 0xFF, 8, s));
        ssort_cluster_kernel<<<64, 256, 0, s>>>(key, desc, n, sr.code[sr.cur], m_valid, flags + 7);
        VNM_HIP(hipGetLastError());
        VNM_HIP(hipMemcpyAsync(&min_span, flags + 7, 8, hipMemcpyDeviceToHost, s));
        unsigned long long too_many = 0;
        unsigned int nh = 0;
        VNM_HIP(hipMemcpyAsync(&eq_pairs, flags + 3, 8, hipMemcpyDeviceToHost, s));
        uint64_t hcodes[SS_MAX_HEAVY];
        VNM_HIP(hipMemcpyAsync(&too_many, flags, 8, hipMemcpyDeviceToHost, s));
        VNM_HIP(hipMemcpyAsync(&nh, nheavy_d, 4, hipMemcpyDeviceToHost, s));
        VNM_HIP(hipMemcpyAsync(hcodes, heavy, sizeof(hcodes), hipMemcpyDeviceToHost, s));
        VNM_HIP(hipStreamSynchronize(s));    // (sr goes back to the pool here)
        if (min_span != ~0ULL && (int64_t)min_span < m_valid / 8 && env_sort_i64("VNM_SSORT_CLUSTER_CHECK", 1)) {
            g_rows_clustered = true;
            if (getenv("VNM_SORT_TRACE")) fprintf(stderr, "[sort] sample sort declined: the rows arrive clustered (a tile of 4096 consecutive rows spans %llu of %lld samples)\n", min_span, (long long)m_valid);
            return 2;
        }
        if (too_many || nh > (unsigned int)SS_MAX_HEAVY || getenv("VNM_SSORT_NO_HEAVY") != nullptr && nh) {
            if (getenv("VNM_SORT_TRACE")) fprintf(stderr, "[sort] sample sort declined: %u heavily duplicated values\n", nh);
            return 2;
        }
        nheavy = (int)nh;
        if (nheavy) {
            std::sort(hcodes, hcodes + nheavy);
            VNM_HIP(hipMemcpyAsync(heavy, hcodes, (size_t)nheavy * 8, hipMemcpyHostToDevice, s));
        }
    }
    // rows of heavy codes bypass the buckets: (heavy index << 32 | row id) entries in a side list, sorted below
    // (room: every run of r equal splitters stands for at most (r + 1) buckets' worth of rows)
    // (+ the NULL rows, counted)
    const double null_rows = (double)total_nulls;
    const int64_t side_cap = (nheavy || has_null) ? std::min<int64_t>(n, (int64_t)((double)(eq_pairs + 2 * (unsigned long long)nheavy) * (double)n / (double)nb * 1.3 + null_rows) + 65536) : 0;
    RadixBufs side{};
    if (nheavy || has_null) VNM_TRY(radix_alloc(&side, side_cap));
    // ---- level 1
    const int pairs1 = env_sort_i64("VNM_SSORT_PAIRS1", 1) >= 2 ? 2 : 1;
    const int64_t sub = 2 * SS_BLOCK * pairs1;
    const int grid1 = (int)std::min<int64_t>((int64_t)cus * env_sort_i64("VNM_SSORT_GRID1_PER_CU", 1), std::max<int64_t>(1, (n + sub - 1) / sub));
    const int64_t rows_per_wg = (((n + sub - 1) / sub + grid1 - 1) / grid1) * sub;
    const int64_t cap1 = ((rows_per_wg / SS_B + rows_per_wg / SS_B / 4 + 96) + 7) & ~7LL;
    uint64_t* c1 = (uint64_t*)pool.take((size_t)SS_B * grid1 * cap1 * 8);
    uint32_t* r1 = (uint32_t*)pool.take((size_t)SS_B * grid1 * cap1 * 4);
    uint32_t* n1 = (uint32_t*)pool.take((size_t)SS_B * grid1 * 4);
    if (!c1 || !r1 || !n1) return 1;
    SsArgs a1{};
    a1.key = key; a1.desc = desc; a1.nrows = n; a1.split = split; a1.l2 = l2;
    a1.out_code = c1; a1.out_row = r1; a1.out_counts = n1; a1.out_cap = cap1; a1.flags = flags;
    a1.heavy = heavy; a1.nheavy = nheavy; a1.has_null = has_null;
    a1.side = (nheavy || has_null) ? (unsigned long long*)side.code[0] : nullptr; a1.side_cap = side_cap;
    {
        KernelTimer timer("sort_scatter1", s);
        if (pairs1 == 2) ssort_scatter_kernel<true, 2><<<grid1, SS_BLOCK, lds_sc, s>>>(a1);
        else ssort_scatter_kernel<true, 1><<<grid1, SS_BLOCK, lds_sc, s>>>(a1);
    }
    VNM_HIP(hipGetLastError());
    // ---- level 2
    int split2 = std::max(1, (grid1 + SS_MAX_REGIONS - 1) / SS_MAX_REGIONS);
    split2 = std::max(split2, (int)env_sort_i64("VNM_SSORT_SPLIT2", 1));
    const int64_t avg_bucket = n / nb + 1;
    const int64_t cap2 = ((std::min<int64_t>(SS_LOCAL, avg_bucket * 2 / split2 + avg_bucket / 2 + 128)) + 7) & ~7LL;
    uint64_t* c2 = (uint64_t*)pool.take((size_t)nb * split2 * cap2 * 8);
    uint32_t* r2 = (uint32_t*)pool.take((size_t)nb * split2 * cap2 * 4);
    uint32_t* n2 = (uint32_t*)pool.take((size_t)nb * split2 * 4);
    unsigned long long* offs = (unsigned long long*)pool.take((size_t)(nb + 1) * 8);
    if (!c2 || !r2 || !n2 || !offs) return 1;
    SsArgs a2{};
    a2.split = split; a2.l2 = l2; a2.in_code = c1; a2.in_row = r1; a2.in_counts = n1; a2.in_cap = cap1; a2.in_regions = grid1; a2.in_split = split2;
    a2.out_code = c2; a2.out_row = r2; a2.out_counts = n2; a2.out_cap = cap2; a2.flags = flags;
    {
        KernelTimer timer("sort_scatter2", s);
        if (env_sort_i64("VNM_SSORT_PAIRS2", 1) >= 2) ssort_scatter_kernel<false, 2><<<SS_B * split2, SS_BLOCK, lds_sc, s>>>(a2);
        else ssort_scatter_kernel<false, 1><<<SS_B * split2, SS_BLOCK, lds_sc, s>>>(a2);
    }
    VNM_HIP(hipGetLastError());
    unsigned long long fl[3] = {0, 0, 0};
    VNM_HIP(hipMemcpyAsync(fl, flags, 24, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    if (fl[0]) {
        if (getenv("VNM_SORT_TRACE")) fprintf(stderr, "[sort] sample sort: a region overflowed (side list %llu of %lld, %lld NULL rows, %lld of %lld samples NULL, %d heavy values), LSD sort instead\n",
                                              fl[2], (long long)side_cap, (long long)total_nulls, (long long)sample_nulls, (long long)m, nheavy);
        return 2;
    }
    // ---- the heavy rows: their side list sorted by (heavy index, row id) is their part of the output
    const int64_t side_len = (int64_t)fl[2];
    if (nheavy || has_null) {
        side.cur = 0;
        if (side_len > 1) VNM_TRY(radix_sort_codes(&side, side_len, s, nullptr, nullptr, false, nullptr, false, nullptr));
        ssort_heavy_bounds_kernel<<<1, 128, 0, s>>>((const unsigned long long*)side.code[side.cur], side_len, heavy, nheavy, split, nb, lb, hb, has_null);
    }
    ssort_offsets_kernel<<<1, 1024, 0, s>>>(n2, split2, nb, offs, flags, nheavy, hb, lb, hstart, has_null);
    VNM_HIP(hipGetLastError());
    VNM_HIP(hipMemcpyAsync(fl, flags, 8, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    if (getenv("VNM_SORT_TRACE")) fprintf(stderr, "[sort] sample sort: n %lld grid1 %d cap1 %lld split2 %d cap2 %lld heavy %d (%lld rows) -> fail %llu special %llu\n",
                                          (long long)n, grid1, (long long)cap1, split2, (long long)cap2, nheavy, (long long)side_len, fl[0], fl[1]);
    if (fl[0]) return 2;
    {
        const int dbg = (int)env_sort_i64("VNM_SSORT_DEBUG", 0);
        if (dbg == 7 || dbg == 8) {
            uint32_t* seen = (uint32_t*)pool.take((size_t)n * 4);
            if (!seen) return 1;
            VNM_HIP(hipMemsetAsync(seen, 0, (size_t)n * 4, s));
            VNM_HIP(hipMemsetAsync(flags + 4, 0, 8, s));
            if (dbg == 7) ssort_check_kernel<<<4096, 256, 0, s>>>(r2, n2, cap2, nb * split2, seen, n, flags + 4);
            else ssort_check_kernel<<<4096, 256, 0, s>>>(r1, n1, cap1, (int64_t)SS_B * grid1, seen, n, flags + 4);
            std::vector<uint32_t> hs((size_t)n);
            unsigned long long badv = 0;
            VNM_HIP(hipMemcpy(hs.data(), seen, (size_t)n * 4, hipMemcpyDeviceToHost));
            VNM_HIP(hipMemcpy(&badv, flags + 4, 8, hipMemcpyDeviceToHost));
            int64_t miss = 0, dup = 0, firstmiss = -1;
            for (int64_t i = 0; i < n; i++) { if (hs[i] == 0) { if (firstmiss < 0) firstmiss = i; miss++; } else if (hs[i] > 1) dup++; }
            fprintf(stderr, "[sort] debug level %d regions: %lld rows missing (first %lld), %lld duplicated, %llu out of range\n", dbg == 7 ? 2 : 1,
                    (long long)miss, (long long)firstmiss, (long long)dup, badv);
        }
    }
    // ---- per-bucket LDS sort, straight into the caller's buffers
    SsLocalArgs la{};
    la.code = c2; la.row = r2; la.counts = n2; la.cap = cap2; la.split = split2; la.offs = offs; la.nbuckets = nb;
    la.idx_out = idx_out;
    const bool keyed = key_out != nullptr && fl[1] == 0 && !has_null;   // (a sorted key column with NULLs in it: the caller gathers)
    la.key_out = keyed ? key_out : nullptr; la.key_type = key.type; la.key_desc = desc; la.flags = flags;
    la.debug = (int)env_sort_i64("VNM_SSORT_DEBUG", 0);
    la.crowded = (uint32_t*)pool.take((size_t)nb * 4);
    if (!la.crowded) return 1;
    VNM_HIP(hipMemsetAsync(la.crowded, 0, (size_t)nb * 4, s));
    {
        KernelTimer timer("sort_local", s);
        ssort_local_kernel<512, 10, 4096, false><<<(int)std::min<int64_t>(nb, (int64_t)cus * 64), 512, (size_t)SS_SMALL * 12 + 4096 * 4, s>>>(la);
        ssort_local_kernel<1024, 8, 8192, true><<<(int)std::min<int64_t>(nb, (int64_t)cus * 16), 1024, (size_t)SS_LOCAL * 12 + 8192 * 4, s>>>(la);
        if ((nheavy || has_null) && side_len > 0)
            ssort_heavy_write_kernel<<<grid_for(side_len), 256, 0, s>>>((const unsigned long long*)side.code[side.cur], side_len, heavy, lb, hstart, idx_out,
                                                                        keyed ? key_out : nullptr, key.type, desc, nheavy);
    }
    VNM_HIP(hipGetLastError());
    VNM_HIP(hipStreamSynchronize(s));
    if (la.debug == 6) {
        unsigned long long f6[8];
        VNM_HIP(hipMemcpy(f6, flags, 64, hipMemcpyDeviceToHost));
        fprintf(stderr, "[sort] debug: %llu positions beyond the end (max %llu), %llu in the long path\n", f6[2], f6[3], f6[4]);
    }
    if (wrote_key) *wrote_key = keyed;
    return 0;
}

// full stable multi-key sort; result: row ids (uint32) in r->val[r->cur]
// idx_out (optional): int64 output buffer; *wrote_idx = the last radix pass wrote it (else the caller widens r->val[r->cur])
// key_out (optional, with idx_out): the sorted values of key 0 (8-byte types, no NaN / NULL / -0.0 in it): *wrote_key
static int full_sort(int n_keys, const vnm_dcol* keys, const int* orders, int64_t n, RadixBufs* r, hipStream_t s,
                     int64_t* idx_out = nullptr, bool* wrote_idx = nullptr, uint64_t* key_out = nullptr, bool* wrote_key = nullptr) {
    if (wrote_idx) *wrote_idx = false;
    if (wrote_key) *wrote_key = false;
    const int kt0 = keys[0].type;
    if (!(kt0 == VNM_F64 || kt0 == VNM_I64 || kt0 == VNM_U64)) key_out = nullptr;
    bool ident = true;   // no pass has run yet: the row ids are the identity and are not materialised (the first pass makes them up)
    for (int k = n_keys - 1; k >= 0; k--) {
        // codes of key k in the current row order; their eight digit histograms come out of the same kernel
        VNM_HIP(hipMemsetAsync(r->red + 4, 0, 8, s));
        VNM_HIP(hipMemsetAsync(r->ghist, 0, (size_t)8 * 256 * 8, s));
        {
            KernelTimer timer("sort_encode", s);
            sort_encode_kernel<<<grid_for(n, 64), 256, 0, s>>>(keys[k], orders[k] == VNM_DESC, ident ? nullptr : r->val[r->cur], n, r->code[r->cur],
                                                               nullptr, r->red + 4, r->ghist, 0);   // 64 workgroups per CU: 3.3 ms per 1e9 keys (8: 3.55)
        }
        unsigned long long any_special = 0;
        const bool cls_possible = keys[k].validity != nullptr || type_is_float(keys[k].type);
        VNM_TRY(radix_sort_codes(r, n, s, &any_special, k == 0 ? idx_out : nullptr, cls_possible, wrote_idx, true, &ident,
                                 k == 0 ? key_out : nullptr, kt0, orders[0] == VNM_DESC, wrote_key));
        // class pass (values < NaN < NULL), more significant than the code, over the classes in the CURRENT order
        if (cls_possible && (any_special & 1ULL) != 0) {   // no NaN / NULL at all: nothing to do
            VNM_HIP(hipMemsetAsync(r->ghist, 0, (size_t)8 * 256 * 8, s));
            {
                KernelTimer timer("sort_encode", s);
                sort_encode_kernel<<<grid_for(n, 64), 256, 0, s>>>(keys[k], orders[k] == VNM_DESC, ident ? nullptr : r->val[r->cur], n,
                                                                   r->code[r->cur], nullptr, nullptr, r->ghist, 1);
            }
            VNM_TRY(radix_sort_codes(r, n, s, nullptr, k == 0 ? idx_out : nullptr, false, wrote_idx, true, &ident));
        }
    }
    VNM_HIP(hipGetLastError());
    if (ident && !(wrote_idx && *wrote_idx)) sort_iota_kernel<<<grid_for(n), 256, 0, s>>>(r->val[r->cur], n);   // every key constant
    return 0;
}

}  // namespace vnm

using namespace vnm;

extern "C" {

int vnm_sort_indices(int n_keys, const vnm_dcol* keys, const int* orders, int64_t length, int64_t limit,
                     int64_t* out_indices, void* stream) {
    return vnm_sort_indices_keyed(n_keys, keys, orders, length, limit, out_indices, nullptr, nullptr, stream);
}

int vnm_sort_indices_keyed(int n_keys, const vnm_dcol* keys, const int* orders, int64_t length, int64_t limit,
                           int64_t* out_indices, void* out_sorted_key0, int* wrote_key0, void* stream) {
    if (wrote_key0) *wrote_key0 = 0;
    VNM_TRY(ensure_init());
    if (n_keys < 1 || n_keys > 16) return set_error("vnm_sort_indices: 1..16 sort keys");
    if (length >= (1LL << 32)) return set_error("vnm_sort_indices: at most 2^32 - 1 rows per sort");
    for (int k = 0; k < n_keys; k++)
        if (keys[k].length != length) return set_error("vnm_sort_indices: key %d length mismatch", k);
    if (length == 0) return 0;
    hipStream_t s = as_stream(stream);
    const int64_t n = length;

    // ---- LIMIT K fast path: one scan selects the candidates ----
    // Several sort keys: the same selection on the FIRST key -- every row that beats the threshold on key 0, ties included, is a
    // candidate (a row that loses on key 0 loses whatever its other keys are) -- and the candidates alone go through the full
    // multi-key sort.
    const bool try_topk = limit > 0 && limit * 8 < n && n >= (1 << 16) && getenv("VNM_SORT_NO_TOPK") == nullptr &&
                          (n_keys == 1 || getenv("VNM_SORT_NO_TOPK_MULTI") == nullptr);
    if (try_topk) {
        route_note("sort:topk_threshold", "LIMIT %lld of %lld rows, %d keys: sampled threshold, candidates through the full sort", (long long)limit, (long long)n, n_keys);
        const int desc = orders[0] == VNM_DESC;
        static bool ts_attr = false;
        if (!ts_attr) {
            VNM_HIP(hipFuncSetAttribute((const void*)topk_small_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TS_MAX * 13));
            ts_attr = true;
        }
        const int64_t m = std::min<int64_t>(n, 1 << 22);
        // threshold = the sample element at the expected rank of the K-th row + a safety margin, in (class, code) order
        // (a threshold that keeps fewer than K rows is noticed below and answered by the full sort)
        double frac = (double)limit / (double)n;
        int64_t rnk = (int64_t)(frac * (double)m * 1.5) + 8 + (int64_t)(6.0 * sqrt(frac * (double)m + 1.0));
        bool ok = rnk < m - 1;
        uint32_t t_cls = 0;
        uint64_t t_code = 0;
        if (ok && rnk < TB_BLOCKS / 4 && m >= TB_BLOCKS * 64) {
            // small ranks: the rnk-th best of the 4096 block winners (one scan of the sample, one workgroup sort)
            PoolScope tpool;
            uint64_t* bcode = (uint64_t*)tpool.take(TB_BLOCKS * 8);
            uint8_t* bcls = (uint8_t*)tpool.take(TB_BLOCKS);
            unsigned long long* thr = (unsigned long long*)tpool.take(64);
            if (!bcode || !bcls || !thr) return 1;
            {
                KernelTimer timer("topk_sample", s);
                topk_blockbest_kernel<<<TB_BLOCKS, 256, 0, s>>>(keys[0], desc, n, m, bcode, bcls);
                topk_small_sort_kernel<<<1, 1024, (size_t)TB_BLOCKS * 13, s>>>(bcode, bcls, nullptr, TB_BLOCKS, TB_BLOCKS, rnk, nullptr, thr);
            }
            unsigned long long th[2] = {0, 0};
            VNM_HIP(hipMemcpyAsync(th, thr, 16, hipMemcpyDeviceToHost, s));
            VNM_HIP(hipStreamSynchronize(s));
            t_cls = (uint32_t)th[0];
            t_code = th[1];
        } else if (ok) {
            // larger ranks: the r01 route -- sort a 2^18-key sample by (class, code) with the grid-wide radix sort and read
            // the threshold off it (its ~130 small launches cost ~1.5 ms, which only matters for small K)
            const int64_t ms_ = std::min<int64_t>(n, 1 << 18);
            rnk = (int64_t)((double)limit / (double)n * (double)ms_ * 1.5) + 64 + (int64_t)(6.0 * sqrt((double)limit / (double)n * (double)ms_ + 1.0));
            ok = rnk < ms_ - 1;
            if (ok) {
                RadixBufs sr{};
                VNM_TRY(radix_alloc(&sr, ms_));
                PoolScope spool;
                uint8_t* scls = (uint8_t*)spool.take((size_t)ms_);
                if (!scls) return 1;
                topk_sample_kernel<<<grid_for(ms_), 256, 0, s>>>(keys[0], desc, n, ms_, sr.code[0], scls);
                sort_iota_kernel<<<grid_for(ms_), 256, 0, s>>>(sr.val[0], ms_);
                int rc = radix_sort_codes(&sr, ms_, s);
                uint64_t* scode_sorted = (uint64_t*)spool.take((size_t)ms_ * 8);
                if (rc || !scode_sorted) return 1;
                VNM_HIP(hipMemcpyAsync(scode_sorted, sr.code[sr.cur], (size_t)ms_ * 8, hipMemcpyDeviceToDevice, s));
                gather_u8_kernel<<<grid_for(ms_), 256, 0, s>>>(scls, sr.val[sr.cur], ms_, sr.code[sr.cur]);
                sort_iota_kernel<<<grid_for(ms_), 256, 0, s>>>(sr.val[sr.cur], ms_);   // stable secondary key: position in the code order
                rc = radix_sort_codes(&sr, ms_, s);
                if (rc) return 1;
                uint32_t pos_in_code_sorted = 0;
                uint64_t cls64 = 0;
                VNM_HIP(hipMemcpyAsync(&pos_in_code_sorted, sr.val[sr.cur] + rnk, 4, hipMemcpyDeviceToHost, s));
                VNM_HIP(hipMemcpyAsync(&cls64, sr.code[sr.cur] + rnk, 8, hipMemcpyDeviceToHost, s));
                VNM_HIP(hipStreamSynchronize(s));
                VNM_HIP(hipMemcpyAsync(&t_code, scode_sorted + pos_in_code_sorted, 8, hipMemcpyDeviceToHost, s));
                VNM_HIP(hipStreamSynchronize(s));
                t_cls = (uint32_t)cls64;
                radix_free(&sr);
            }
        }
        if (ok) {
            const int64_t cap = std::max<int64_t>(limit * 4 + 65536, 1 << 20);
            RadixBufs cr{};
            VNM_TRY(radix_alloc(&cr, cap));
            PoolScope cpool;
            uint8_t* ccls = (uint8_t*)cpool.take((size_t)cap);
            uint32_t* crows = (uint32_t*)cpool.take((size_t)cap * 4);
            unsigned long long* cnt = (unsigned long long*)cpool.take(64);
            if (!ccls || !crows || !cnt) return 1;
            VNM_HIP(hipMemsetAsync(cnt, 0, 8, s));
            {
                KernelTimer timer("topk_select", s);
                topk_select_kernel<<<device_info().num_cus * 8, 256, 0, s>>>(keys[0], desc, n, t_cls, t_code, cap, cnt, cr.code[0], ccls, crows);
            }
            unsigned long long found = 0;
            VNM_HIP(hipMemcpyAsync(&found, cnt, 8, hipMemcpyDeviceToHost, s));
            VNM_HIP(hipStreamSynchronize(s));
            int rc2 = 0;
            bool done = false;
            if (n_keys > 1) {
                if ((int64_t)found >= limit && (int64_t)found <= cap) {
                    const int64_t c = (int64_t)found;
                    PoolScope pool;
                    // the candidates in ROW order (the multi-key sort below is stable over it)
                    u32_to_code_kernel<<<grid_for(c), 256, 0, s>>>(crows, c, cr.code[0]);
                    sort_iota_kernel<<<grid_for(c), 256, 0, s>>>(cr.val[0], c);
                    cr.cur = 0;
                    rc2 = radix_sort_codes(&cr, c, s);
                    int64_t* rows64 = (int64_t*)pool.take((size_t)c * 8);
                    int64_t* perm = (int64_t*)pool.take((size_t)c * 8);
                    if (!rc2 && (!rows64 || !perm)) rc2 = 1;
                    if (!rc2) VNM_HIP(hipMemcpyAsync(rows64, cr.code[cr.cur], (size_t)c * 8, hipMemcpyDeviceToDevice, s));   // the codes ARE the row ids
                    // their key columns, gathered
                    std::vector<vnm_dcol> sub((size_t)n_keys);
                    for (int j = 0; j < n_keys && !rc2; j++) {
                        const int w = type_width(keys[j].type);
                        void* dv = pool.take((size_t)c * w);
                        uint8_t* db = keys[j].validity ? (uint8_t*)pool.take((size_t)c) : nullptr;
                        uint8_t* bm = keys[j].validity ? (uint8_t*)pool.take((size_t)(c + 7) / 8 + 8) : nullptr;
                        if (!dv || (keys[j].validity && (!db || !bm))) { rc2 = 1; break; }
                        take_kernel<<<grid_for(c), 256, 0, s>>>(keys[j], rows64, c, dv, db);
                        if (bm) rc2 = vnm_pack_validity(db, c, bm, (void*)s);
                        sub[j] = keys[j];
                        sub[j].values = dv; sub[j].validity = bm; sub[j].offset = 0; sub[j].length = c;
                    }
                    if (!rc2) {
                        RadixBufs r2{};
                        rc2 = radix_alloc(&r2, c);
                        bool wrote = false;
                        if (!rc2) rc2 = full_sort(n_keys, sub.data(), orders, c, &r2, s, perm, &wrote);
                        if (!rc2 && !wrote) sort_widen_kernel<<<grid_for(c), 256, 0, s>>>(r2.val[r2.cur], c, perm);
                        if (!rc2) {
                            gather_i64_kernel<<<grid_for(limit), 256, 0, s>>>(rows64, perm, limit, out_indices);
                            VNM_HIP(hipGetLastError());
                            VNM_HIP(hipStreamSynchronize(s));
                            done = true;
                        }
                        radix_free(&r2);
                    }
                }
            } else if ((int64_t)found >= limit && (int64_t)found <= TS_MAX && getenv("VNM_SORT_NO_SMALL") == nullptr) {
                // few candidates: one workgroup sorts them in LDS and writes the first K row ids
                const int c = (int)found;
                int np2 = 64;
                while (np2 < c) np2 <<= 1;
                const size_t lds = (size_t)np2 * 13;
                {
                    KernelTimer timer("topk_small_sort", s);
                    topk_small_sort_kernel<<<1, 1024, lds, s>>>(cr.code[0], ccls, crows, c, np2, limit, out_indices, nullptr);
                }
                VNM_HIP(hipGetLastError());
                VNM_HIP(hipStreamSynchronize(s));
                done = true;
            } else if ((int64_t)found >= limit && (int64_t)found <= cap) {
                const int64_t c = (int64_t)found;
                // canonical order: by row id, then (stable) by code, then by class
                uint64_t* code_keep = (uint64_t*)cpool.take((size_t)c * 8);
                if (!code_keep) return 1;
                VNM_HIP(hipMemcpyAsync(code_keep, cr.code[0], (size_t)c * 8, hipMemcpyDeviceToDevice, s));
                // pass group 1: key = row id, value = candidate slot
                u32_to_code_kernel<<<grid_for(c), 256, 0, s>>>(crows, c, cr.code[0]);
                sort_iota_kernel<<<grid_for(c), 256, 0, s>>>(cr.val[0], c);
                cr.cur = 0;
                rc2 = radix_sort_codes(&cr, c, s);
                // pass group 2: key = code of the slot
                if (!rc2) { gather_u64_kernel<<<grid_for(c), 256, 0, s>>>(code_keep, cr.val[cr.cur], c, cr.code[cr.cur]); rc2 = radix_sort_codes(&cr, c, s); }
                // pass group 3: key = class of the slot
                if (!rc2) { gather_u8_kernel<<<grid_for(c), 256, 0, s>>>(ccls, cr.val[cr.cur], c, cr.code[cr.cur]); rc2 = radix_sort_codes(&cr, c, s); }
                if (!rc2) {
                    // slots -> row ids -> int64 output (first `limit` entries)
                    gather_u32_kernel<<<grid_for(limit), 256, 0, s>>>(crows, cr.val[cr.cur], limit, cr.val[cr.cur ^ 1]);
                    sort_widen_kernel<<<grid_for(limit), 256, 0, s>>>(cr.val[cr.cur ^ 1], limit, out_indices);
                    VNM_HIP(hipGetLastError());
                    VNM_HIP(hipStreamSynchronize(s));
                    done = true;
                }
            }
            radix_free(&cr);
            if (rc2) return rc2;
            if (done) return 0;
        }
        // fall through to the full sort (ties / NaN / NULL heavy data, or an unlucky sample)
    }

    // one 8-byte key without NULLs, many rows: sample sort (two bucket scatters + a sort in LDS) instead of eight LSD passes
    if (n_keys == 1 && (!keys[0].validity || getenv("VNM_SSORT_NO_NULLS") == nullptr) && (keys[0].type == VNM_F64 || keys[0].type == VNM_I64 || keys[0].type == VNM_U64) &&
        n >= env_sort_i64("VNM_SSORT_MIN_ROWS", (int64_t)1 << 25) && getenv("VNM_SORT_NO_SAMPLE") == nullptr) {
        bool wk = false;
        g_rows_clustered = false;
        // only the order is asked for: 8-byte entry words (an equalising map of the code + row id) instead of (code, row id) -- the
        // sample decides (duplicated keys, lumpy distributions keep the splitters)
        if (!out_sorted_key0 && env_sort_i64("VNM_SORT_APX", 1)) {
            const int rx = sample_sort_apx(keys[0], orders[0] == VNM_DESC, n, out_indices, s);
            if (rx == 1) return 1;
            if (rx == 0) {
                route_note("sort:sample_sort_words", "%lld rows, one 8-byte key, order only: an equalising map from a sample, two scatters of 8-byte entry words, per-bucket LDS sort", (long long)n);
                if (wrote_key0) *wrote_key0 = 0;
                return 0;
            }
        }
        if (!g_rows_clustered) {   // (rows that arrive clustered: the splitter sort would decline them as well)
            route_note("sort:sample_sort", "%lld rows, one 8-byte key: splitters from a sample, two bucket scatters, per-bucket LDS sort", (long long)n);
            const int rc = sample_sort(keys[0], orders[0] == VNM_DESC, n, out_indices, (uint64_t*)out_sorted_key0, &wk, s);
            if (rc == 1) return 1;
            if (rc == 0) { if (wrote_key0) *wrote_key0 = wk ? 1 : 0; return 0; }
        }
    }
    // ... one 4-byte key (float32 / int32 / uint32) the same way (round 5): widened to 8 bytes -- float32 -> float64 is exact and keeps the
    // order, NaNs and signed zeros -- the sample sort orders the copy; 5e8 float32 keys through the LSD passes: 27.9 ms, float64: 15.7
    if (n_keys == 1 && !keys[0].validity && (keys[0].type == VNM_F32 || keys[0].type == VNM_I32 || keys[0].type == VNM_U32) &&
        n >= env_sort_i64("VNM_SSORT_MIN_ROWS", (int64_t)1 << 25) && getenv("VNM_SORT_NO_SAMPLE") == nullptr && getenv("VNM_SORT_NO_WIDEN") == nullptr) {
        uint64_t* wide = (uint64_t*)pool_alloc((size_t)n * 8);
        if (!wide) return 1;
        sort_widen_key_kernel<<<grid_for(n), 256, 0, s>>>(keys[0].values, keys[0].type, keys[0].offset, n, wide);
        int rc = hipGetLastError() == hipSuccess ? 0 : set_error("vnm_sort_indices: kernel launch failed");
        if (!rc) {
            vnm_dcol wk8{};
            wk8.values = wide; wk8.length = n;
            wk8.type = keys[0].type == VNM_F32 ? VNM_F64 : (keys[0].type == VNM_I32 ? VNM_I64 : VNM_U64);
            bool wkey = false;
            if (env_sort_i64("VNM_SORT_APX", 1)) {
                const int rx = sample_sort_apx(wk8, orders[0] == VNM_DESC, n, out_indices, s);
                if (rx == 1) { pool_free(wide); return 1; }
                if (rx == 0) {
                    route_note("sort:sample_sort_words", "%lld rows, one 4-byte key widened to 8 bytes, order only: two scatters of 8-byte entry words, per-bucket LDS sort", (long long)n);
                    (void)hipStreamSynchronize(s);
                    pool_free(wide);
                    if (wrote_key0) *wrote_key0 = 0;
                    return 0;
                }
            }
            route_note("sort:sample_sort", "%lld rows, one 4-byte key widened to 8 bytes: splitters from a sample, two bucket scatters, per-bucket LDS sort", (long long)n);
            rc = sample_sort(wk8, orders[0] == VNM_DESC, n, out_indices, nullptr, &wkey, s);
        }
        if (hipStreamSynchronize(s) != hipSuccess && rc == 0) rc = set_error("vnm_sort_indices: stream sync failed");
        pool_free(wide);
        if (rc == 1) return 1;
        if (rc == 0) { if (wrote_key0) *wrote_key0 = 0; return 0; }
        // (rc 2: the sample sort declined -- heavy values, an overflowing bucket: the LSD passes over the original column)
    }
    if (g_rows_clustered && n_keys == 1 && !keys[0].validity && !out_sorted_key0 && type_width(keys[0].type) == 8 && getenv("VNM_SORT_NO_SORTED_CHECK") == nullptr) {
        g_rows_clustered = false;
        unsigned long long* inv = (unsigned long long*)pool_alloc(64);
        if (!inv) return 1;
        unsigned long long ninv[2] = {1, 1};
        int rc = 0;
        if (hipMemsetAsync(inv, 0, 16, s) != hipSuccess) rc = set_error("vnm_sort_indices: memset failed");
        if (!rc) {
            sort_inversions_kernel<<<grid_for(n, 16), 256, 0, s>>>(keys[0], orders[0] == VNM_DESC, n, inv);
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(ninv, inv, 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
                rc = set_error("vnm_sort_indices: the sortedness check failed");
        }
        pool_free(inv);
        if (rc) return rc;
        if (ninv[0] == 0 || ninv[1] == 0) {
            const int backwards = ninv[0] != 0;
            route_note("sort:already_sorted", "%lld rows, one 8-byte key: the rows arrive %s: the indices are the row numbers%s", (long long)n,
                       backwards ? "strictly in the opposite order (every row below its predecessor, no ties)" : "in the order asked for (no row below its predecessor)", backwards ? " backwards" : "");
            sort_iota64_kernel<<<grid_for(n, 16), 256, 0, s>>>(out_indices, n, backwards);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return set_error("vnm_sort_indices: kernel launch failed");
            if (wrote_key0) *wrote_key0 = 0;
            return 0;
        }
    }
    g_rows_clustered = false;
    RadixBufs r{};
    VNM_TRY(radix_alloc(&r, n));
    bool wrote = false, wrote_key = false;
    route_note("sort:lsd_radix", "%lld rows, %d keys: stable LSD passes over order-preserving codes, last key first", (long long)n, n_keys);
    int rc = full_sort(n_keys, keys, orders, n, &r, s, out_indices, &wrote, (uint64_t*)out_sorted_key0, &wrote_key);
    if (!rc && wrote_key0) *wrote_key0 = wrote_key ? 1 : 0;
    if (!rc) {
        if (!wrote) sort_widen_kernel<<<grid_for(n), 256, 0, s>>>(r.val[r.cur], n, out_indices);
        if (hipGetLastError() != hipSuccess) rc = set_error("vnm_sort_indices: kernel launch failed");
        if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = set_error("vnm_sort_indices: stream sync failed");
    }
    radix_free(&r);
    return rc;
}

int vnm_take(const vnm_dcol* col, const int64_t* indices, int64_t n, void* out_values, uint8_t* out_valid, void* stream) {
    VNM_TRY(ensure_init());
    if (!col || (n > 0 && (!indices || !out_values))) return set_error("vnm_take: null argument");
    if (col->validity && !out_valid) return set_error("vnm_take: column has nulls but no out_valid buffer");
    if (n <= 0) return 0;
    take_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>(*col, indices, n, out_values, out_valid);
    VNM_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
