// Take (gather by row ids) for the column types the numeric vnm_take does not cover, and sort keys of decimal128 columns: what
// Sort::Sorted needs below the C ABI for tables with strings, binaries, booleans and decimals (vinum_cpp/src/operators/sort/sort.cpp:
// 15-63: arrow SortIndices over the sort columns + Take of EVERY column, any Arrow type).  Round 5; string / binary sort KEYS become
// order-preserving ranks through the string dictionary (vnm_strdict.hip: vnm_strdict_ranks_device).
#include <algorithm>

#include "vnm_common.hpp"

namespace vnm {

constexpr int SCAN_BLOCK = 256, SCAN_PER = 8, SCAN_TILE = SCAN_BLOCK * SCAN_PER;

// lengths of the taken values (0 for a NULL row), and the row's validity as a byte
__global__ __launch_bounds__(256) void vtake_len_kernel(const int64_t* offs, const uint8_t* valid, const int64_t* idx, int64_t n, uint64_t* len, uint8_t* out_valid) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int64_t r = idx[i];
        const bool ok = !valid || ((valid[r >> 3] >> (r & 7)) & 1);
        len[i] = ok ? (uint64_t)(offs[r + 1] - offs[r]) : 0ULL;
        if (out_valid) out_valid[i] = ok ? 1 : 0;
    }
}

// exclusive prefix sum of n uint64 values -> out[0 .. n] (out[n] = total): per-tile sums, one workgroup over the tile sums, per-tile scan
__global__ __launch_bounds__(SCAN_BLOCK) void scan_tiles_kernel(const uint64_t* in, int64_t n, uint64_t* tile_sum) {
    __shared__ uint64_t s[SCAN_BLOCK];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    uint64_t a = 0;
    for (int k = 0; k < SCAN_PER; k++) { const int64_t i = base + (int64_t)threadIdx.x * SCAN_PER + k; if (i < n) a += in[i]; }
    s[threadIdx.x] = a;
    __syncthreads();
    for (int d = SCAN_BLOCK / 2; d > 0; d >>= 1) { if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = s[0];
}
__global__ __launch_bounds__(1024) void scan_sums_kernel(uint64_t* tile_sum, int64_t ntiles) {   // in place -> exclusive
    __shared__ uint64_t s[1024];
    const int64_t per = (ntiles + 1023) / 1024;
    const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < ntiles ? lo + per : ntiles;
    uint64_t a = 0;
    for (int64_t i = lo; i < hi; i++) a += tile_sum[i];
    s[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t run = 0; for (int i = 0; i < 1024; i++) { const uint64_t x = s[i]; s[i] = run; run += x; } }
    __syncthreads();
    uint64_t run = s[threadIdx.x];
    for (int64_t i = lo; i < hi; i++) { const uint64_t x = tile_sum[i]; tile_sum[i] = run; run += x; }
}
__global__ __launch_bounds__(SCAN_BLOCK) void scan_final_kernel(const uint64_t* in, int64_t n, const uint64_t* tile_base, int64_t* out) {
    __shared__ uint64_t s[SCAN_BLOCK];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_PER;
    uint64_t v[SCAN_PER], a = 0;
    for (int k = 0; k < SCAN_PER; k++) { v[k] = base + k < n ? in[base + k] : 0ULL; a += v[k]; }
    s[threadIdx.x] = a;
    __syncthreads();
    for (int d = 1; d < SCAN_BLOCK; d <<= 1) {   // inclusive scan of the thread sums
        const uint64_t x = (int)threadIdx.x >= d ? s[threadIdx.x - d] : 0ULL;
        __syncthreads();
        s[threadIdx.x] += x;
        __syncthreads();
    }
    uint64_t run = tile_base[blockIdx.x] + s[threadIdx.x] - a;
    for (int k = 0; k < SCAN_PER; k++) {
        if (base + k < n) out[base + k] = (int64_t)run;
        run += v[k];
        if (base + k == n - 1) out[n] = (int64_t)run;
    }
}

// the bytes: one WAVE per taken value (values are short -- names, dates as text -- or long: either way the lanes of a wave copy one
// value's bytes side by side)
__global__ __launch_bounds__(256) void vtake_copy_kernel(const int64_t* offs, const uint8_t* data, const int64_t* idx, int64_t n, const int64_t* out_offs, uint8_t* out) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += waves) {
        const int64_t o0 = out_offs[i], len = out_offs[i + 1] - o0;
        if (len == 0) continue;
        const uint8_t* p = data + offs[idx[i]];
        for (int64_t b = lane; b < len; b += 64) out[o0 + b] = p[b];
    }
}

__global__ __launch_bounds__(256) void take_bits_kernel(const uint8_t* bits, int64_t bit_offset, const int64_t* idx, int64_t n, uint8_t* out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int64_t r = bit_offset + idx[i];
        out[i] = (bits[r >> 3] >> (r & 7)) & 1;
    }
}

__global__ __launch_bounds__(256) void take_fixed16_kernel(const ulonglong2* v, const int64_t* idx, int64_t n, ulonglong2* out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = v[idx[i]];
}

// decimal128 (little endian two's complement, 16 bytes) -> (high word as int64, low word as uint64): two sort keys, most significant first
__global__ __launch_bounds__(256) void dec128_keys_kernel(const ulonglong2* v, int64_t n, int64_t* hi, uint64_t* lo) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { const ulonglong2 x = v[i]; lo[i] = x.x; hi[i] = (int64_t)x.y; }
}

static int grid_rows(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, (int64_t)device_info().num_cus * 8); }


// ---- stable partition of rows by OWNER (distributed sample sort: rows -> the rank whose splitter range holds their order code) -----
// owner(code) = number of splitters <= code (a code equal to a splitter goes to the upper rank: equal codes never split), W = n_split + 1
// <= 64 owners.  Three launches: per-block owner histograms, an exclusive scan per owner over the blocks (owner-major: all rows of
// owner 0 first), a stable scatter of the row numbers (block order, then row order inside the block).
constexpr int PO_BLOCK = 256, PO_ITERS = 16, PO_TILE = PO_BLOCK * PO_ITERS, PO_MAXW = 64;
__device__ __forceinline__ int po_owner(int64_t code, const int64_t* sp, int nsp) {
    int lo = 0, hi = nsp;                    // first splitter > code
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sp[mid] <= code) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(PO_BLOCK) void po_hist_kernel(const int64_t* codes, int64_t n, const int64_t* splitters, int nsp, uint32_t* hist /* [W][nblocks] */, int64_t nblocks) {
    __shared__ int64_t sp[PO_MAXW];
    __shared__ uint32_t cnt[PO_MAXW];
    if ((int)threadIdx.x < nsp) sp[threadIdx.x] = splitters[threadIdx.x];
    if (threadIdx.x < PO_MAXW) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * PO_TILE;
    for (int it = 0; it < PO_ITERS; it++) {
        const int64_t r = base + (int64_t)it * PO_BLOCK + threadIdx.x;
        if (r < n) atomicAdd(&cnt[po_owner(codes[r], sp, nsp)], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x <= nsp) hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = cnt[threadIdx.x];
}
// one workgroup per owner: exclusive scan of its row of `hist` in place; totals[o] = the owner's rows
__global__ __launch_bounds__(1024) void po_scan_kernel(uint32_t* hist, int64_t nblocks, int64_t* totals) {
    __shared__ uint64_t s[1024];
    uint32_t* row = hist + (int64_t)blockIdx.x * nblocks;
    const int64_t per = (nblocks + 1023) / 1024;
    const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < nblocks ? lo + per : nblocks;
    uint64_t a = 0;
    for (int64_t i = lo; i < hi; i++) a += row[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint64_t v = (int)threadIdx.x >= d ? s[threadIdx.x - d] : 0;
        __syncthreads();
        s[threadIdx.x] += v;
        __syncthreads();
    }
    uint64_t run = s[threadIdx.x] - a;
    for (int64_t i = lo; i < hi; i++) { const uint32_t c = row[i]; row[i] = (uint32_t)run; run += c; }     // (offsets inside one owner: < 2^32 rows per call)
    if (threadIdx.x == 1023) totals[blockIdx.x] = (int64_t)s[1023];
}
__global__ __launch_bounds__(PO_BLOCK) void po_scatter_kernel(const int64_t* codes, int64_t n, const int64_t* splitters, int nsp, const uint32_t* hist, int64_t nblocks,
                                                              const int64_t* totals, int64_t* out_order) {
    __shared__ int64_t sp[PO_MAXW];
    __shared__ int64_t run[PO_MAXW];                 // next output position of every owner for this block
    __shared__ uint32_t wcnt[PO_BLOCK / 64][PO_MAXW];
    const int W = nsp + 1, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)threadIdx.x < nsp) sp[threadIdx.x] = splitters[threadIdx.x];
    if ((int)threadIdx.x < W) {
        int64_t before = 0;
        for (int o = 0; o < (int)threadIdx.x; o++) before += totals[o];
        run[threadIdx.x] = before + hist[(int64_t)threadIdx.x * nblocks + blockIdx.x];
    }
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * PO_TILE;
    for (int it = 0; it < PO_ITERS; it++) {
        const int64_t r = base + (int64_t)it * PO_BLOCK + threadIdx.x;
        const int o = r < n ? po_owner(codes[r], sp, nsp) : -1;
        if (lane < W) wcnt[wave][lane] = 0;
        // rank of this lane among the lanes of its wave with the same owner
        uint32_t rank = 0;
        unsigned long long todo = __ballot(o >= 0);
        while (todo) {
            const int lead = __ffsll((long long)todo) - 1;
            const int lo_ = __shfl(o, lead);
            const unsigned long long m = __ballot(o == lo_);
            if (o == lo_) rank = (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
            if (lane == lead) wcnt[wave][lo_] = (uint32_t)__popcll(m);
            todo &= ~m;
        }
        __syncthreads();
        if (o >= 0) {
            int64_t pos = run[o] + rank;
            for (int w = 0; w < wave; w++) pos += wcnt[w][o];
            out_order[pos] = r;
        }
        __syncthreads();
        if ((int)threadIdx.x < W) { uint32_t t = 0; for (int w = 0; w < PO_BLOCK / 64; w++) t += wcnt[w][threadIdx.x]; run[threadIdx.x] += t; }
        __syncthreads();
    }
}

}  // namespace vnm

using namespace vnm;

extern "C" {

int vnm_take_varwidth(const int64_t* offsets, const uint8_t* data, const uint8_t* validity, const int64_t* indices, int64_t n,
                      int64_t* out_offsets, uint8_t** out_data, int64_t* out_bytes, uint8_t* out_valid, void* stream) {
    VNM_TRY(ensure_init());
    if (!out_data || !out_bytes) return set_error("vnm_take_varwidth: null argument");
    *out_data = nullptr; *out_bytes = 0;
    if (n <= 0) return 0;
    if (!offsets || !indices || !out_offsets) return set_error("vnm_take_varwidth: null argument");
    hipStream_t s = as_stream(stream);
    PoolScope pool;
    const int64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    uint64_t* len = (uint64_t*)pool.take((size_t)n * 8);
    uint64_t* tsum = (uint64_t*)pool.take((size_t)ntiles * 8);
    if (!len || !tsum) return 1;
    vtake_len_kernel<<<grid_rows(n), 256, 0, s>>>(offsets, validity, indices, n, len, out_valid);
    scan_tiles_kernel<<<(int)ntiles, SCAN_BLOCK, 0, s>>>(len, n, tsum);
    scan_sums_kernel<<<1, 1024, 0, s>>>(tsum, ntiles);
    scan_final_kernel<<<(int)ntiles, SCAN_BLOCK, 0, s>>>(len, n, tsum, out_offsets);
    VNM_HIP(hipGetLastError());
    int64_t total = 0;
    VNM_HIP(hipMemcpyAsync(&total, out_offsets + n, 8, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    uint8_t* out = (uint8_t*)pool_alloc((size_t)(total > 0 ? total : 1));
    if (!out) return 1;
    if (total > 0) {
        vtake_copy_kernel<<<(int)std::min<int64_t>((n + 3) / 4, (int64_t)device_info().num_cus * 16), 256, 0, s>>>(offsets, data, indices, n, out_offsets, out);
        if (hipGetLastError() != hipSuccess) { pool_free(out); return set_error("vnm_take_varwidth: copy kernel failed"); }
    }
    *out_data = out; *out_bytes = total;
    return 0;
}

int vnm_take_bits(const uint8_t* bits, int64_t bit_offset, const int64_t* indices, int64_t n, uint8_t* out_bytes, void* stream) {
    VNM_TRY(ensure_init());
    if (n <= 0) return 0;
    if (!bits || !indices || !out_bytes) return set_error("vnm_take_bits: null argument");
    take_bits_kernel<<<grid_rows(n), 256, 0, as_stream(stream)>>>(bits, bit_offset, indices, n, out_bytes);
    VNM_HIP(hipGetLastError());
    return 0;
}

int vnm_take_fixed16(const void* values, const int64_t* indices, int64_t n, void* out_values, void* stream) {
    VNM_TRY(ensure_init());
    if (n <= 0) return 0;
    if (!values || !indices || !out_values) return set_error("vnm_take_fixed16: null argument");
    take_fixed16_kernel<<<grid_rows(n), 256, 0, as_stream(stream)>>>((const ulonglong2*)values, indices, n, (ulonglong2*)out_values);
    VNM_HIP(hipGetLastError());
    return 0;
}

int vnm_decimal128_sort_keys(const void* values, int64_t n, int64_t* out_hi, uint64_t* out_lo, void* stream) {
    VNM_TRY(ensure_init());
    if (n <= 0) return 0;
    if (!values || !out_hi || !out_lo) return set_error("vnm_decimal128_sort_keys: null argument");
    dec128_keys_kernel<<<grid_rows(n), 256, 0, as_stream(stream)>>>((const ulonglong2*)values, n, out_hi, out_lo);
    VNM_HIP(hipGetLastError());
    return 0;
}


int vnm_partition_by_owner(const int64_t* codes, int64_t n, const int64_t* splitters, int n_splitters, int64_t* out_order, int64_t* out_counts, void* stream) {
    VNM_TRY(ensure_init());
    if (n < 0 || n_splitters < 0 || n_splitters >= PO_MAXW) return set_error("vnm_partition_by_owner: at most %d owners", PO_MAXW);
    if (!out_counts || (n > 0 && (!codes || !out_order)) || (n_splitters > 0 && !splitters)) return set_error("vnm_partition_by_owner: null argument");
    if (n >= (1LL << 32)) return set_error("vnm_partition_by_owner: at most 2^32 - 1 rows per call");
    hipStream_t s = as_stream(stream);
    const int W = n_splitters + 1;
    if (n == 0) { VNM_HIP(hipMemsetAsync(out_counts, 0, (size_t)W * 8, s)); return 0; }
    const int64_t nblocks = (n + PO_TILE - 1) / PO_TILE;
    PoolScope pool;
    uint32_t* hist = (uint32_t*)pool.take((size_t)W * nblocks * 4);
    if (!hist) return 1;
    KernelTimer timer("partition_by_owner", s);
    po_hist_kernel<<<(int)nblocks, PO_BLOCK, 0, s>>>(codes, n, splitters, n_splitters, hist, nblocks);
    po_scan_kernel<<<W, 1024, 0, s>>>(hist, nblocks, out_counts);
    po_scatter_kernel<<<(int)nblocks, PO_BLOCK, 0, s>>>(codes, n, splitters, n_splitters, hist, nblocks, out_counts, out_order);
    VNM_HIP(hipGetLastError());
    VNM_HIP(hipStreamSynchronize(s));    // (the scratch block goes back to the pool)
    return 0;
}
}  // extern "C"
