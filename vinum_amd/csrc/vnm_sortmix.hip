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

}  // extern "C"
