// String dictionary on the device: value -> int32 code for the non-numeric GROUP BY keys of GenericHashAggregate.
//
// The reference keys its generic map on vectors of arrow::Scalar -- one hash + Equals per row on the host
// (vinum_cpp/src/operators/aggregate/generic_hash_aggregate.h:10-45).  Round 2 dictionary-encoded such columns with Arrow's
// dictionary_encode + NumPy on the host: 7-17 M rows/s, 1.5 M rows/s at 5e6 distinct values -- three orders of magnitude
// behind the numeric operators the codes then go through.  Here the column's offsets and bytes cross PCIe once and ONE kernel
// maps every row to its code:
//   * hash of the bytes (8-byte steps), find-or-insert in a table of 32-byte slots [tag, id + 1, where, length];
//     a candidate with the same tag and length is compared byte for byte -- nothing rests on the 63-bit tag;
//   * a value first seen in this batch is represented by the ROW that claimed the slot (`where` = NEW | row); after the
//     kernel the new values are copied, in table order, to the end of the dictionary's byte heap (`where` = heap offset):
//     no allocation inside the claim, the heap is dense and sized exactly;
//   * ids come from per-workgroup chunks of a global counter (an LDS atomic inside the claim; holes < 2x);
//   * the table grows by rehash between launches (workgroups resume from progress[]); ids and heap offsets survive.
// The host side keeps the dictionary's values (it receives only the NEW values of each batch: ids, lengths, bytes) and maps
// the result's codes back (vinum_amd/vinum_lib.py::KeyDictionary).
#include <algorithm>
#include <cstring>
#include <vector>

#include "vnm_common.hpp"

namespace vnm {

constexpr uint64_t SD_EMPTY = ~0ULL, SD_LOCKED = ~0ULL - 1, SD_NEW = 1ULL << 63;
constexpr int SD_TILE = 2048;      // rows per workgroup and room check
constexpr int SD_PER = 8;          // slots per thread in the compaction passes

struct SDict {
    uint64_t* slot;                // [cap][4]: tag, id + 1, heap offset or SD_NEW | row, length
    uint64_t cap;
    unsigned long long* ctl;       // [1] a workgroup ran out of room  [2] fill  [3] values new in this batch
};

struct SdArgs {
    const void* offs;              // Arrow offsets (int32 / int64), element `first` belongs to row 0
    const int32_t* lens;           // SPANS (null for Arrow offsets): row r is the lens[r] bytes at offs[r] (int64) -- fields of a CSV block in place
    int wide;
    int64_t first;
    const uint8_t* valid;          // bitmap, bit `vfirst` belongs to row 0 (or null)
    int64_t vfirst;
    const uint8_t* data;           // byte `data_base` of the Arrow data buffer
    int64_t data_base;
    int64_t nrows, ntiles, margin, fill_limit;
    SDict d;
    const uint8_t* heap;
    unsigned int* progress;
    int32_t* out;
    unsigned long long* gnext;
};

__device__ __forceinline__ uint64_t sd_ld(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sd_st(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int64_t sd_off(const SdArgs& a, int64_t row) {
    return a.wide ? ((const int64_t*)a.offs)[a.first + row] : (int64_t)((const int32_t*)a.offs)[a.first + row];
}

__device__ __forceinline__ uint64_t sd_mix(uint64_t h) {
    h *= 0xff51afd7ed558ccdULL;
    h ^= h >> 32;
    return h;
}
__device__ __forceinline__ uint64_t sd_hash(const uint8_t* p, int64_t len) {
    uint64_t h = 0x9E3779B97F4A7C15ULL ^ ((uint64_t)len * 0xc4ceb9fe1a85ec53ULL);
    int64_t i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t x;
        __builtin_memcpy(&x, p + i, 8);
        h = sd_mix(h ^ x);
    }
    uint64_t x = 0;
    for (int k = 0; i + k < len; k++) x |= (uint64_t)p[i + k] << (8 * k);
    h = sd_mix(h ^ x ^ 0x5bd1e9955bd1e995ULL);
    return sd_mix(h) & 0x7FFFFFFFFFFFFFFFULL;
}
__device__ __forceinline__ bool sd_equal(const uint8_t* a, const uint8_t* b, int64_t len) {
    int64_t i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, a + i, 8);
        __builtin_memcpy(&y, b + i, 8);
        if (x != y) return false;
    }
    for (; i < len; i++)
        if (a[i] != b[i]) return false;
    return true;
}

__global__ __launch_bounds__(256) void sd_encode_kernel(SdArgs a) {
    __shared__ int s_go;
    __shared__ unsigned s_new;
    __shared__ unsigned long long s_gnext, s_gend;
    const int tid = threadIdx.x;
    if (tid == 0) { s_new = 0; s_gnext = 0; s_gend = 0; }
    __syncthreads();
    const uint64_t mask = a.d.cap - 1;
    unsigned it = a.progress[blockIdx.x];
    for (;; it++) {
        const int64_t tile = (int64_t)blockIdx.x + (int64_t)it * gridDim.x;
        if (tile >= a.ntiles) break;
        __syncthreads();
        if (tid == 0) {
            const unsigned v = atomicExch(&s_new, 0u);
            if (v) { atomicAdd(&a.d.ctl[2], (unsigned long long)v); atomicAdd(&a.d.ctl[3], (unsigned long long)v); }
            const unsigned long long fill = __hip_atomic_load(&a.d.ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool room = (int64_t)fill + a.margin <= a.fill_limit;
            if (!room) __hip_atomic_store(&a.d.ctl[1], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (room && s_gend - s_gnext < (unsigned long long)SD_TILE) {   // every row of a tile may bring a new value
                const unsigned long long base = atomicAdd(a.gnext, 2ULL * SD_TILE);
                s_gnext = base; s_gend = base + 2ULL * SD_TILE;
            }
            s_go = room ? 1 : 0;
        }
        __syncthreads();
        if (!s_go) break;
        for (int r = 0; r < SD_TILE / 256; r++) {
            const int64_t row = tile * SD_TILE + (int64_t)r * 256 + tid;
            if (row >= a.nrows) continue;
            if (a.valid && !((a.valid[(a.vfirst + row) >> 3] >> ((a.vfirst + row) & 7)) & 1)) { a.out[row] = -1; continue; }
            const int64_t o0 = sd_off(a, row), len = a.lens ? (int64_t)a.lens[row] : sd_off(a, row + 1) - o0;
            const uint8_t* p = a.data + (o0 - a.data_base);
            const uint64_t tagv = sd_hash(p, len);
            uint64_t h = (tagv ^ (tagv >> 29)) & mask;
            int64_t id;
            for (;;) {
                uint64_t* s = a.d.slot + h * 4;
                const uint64_t t = sd_ld(s);
                if (t == tagv) {   // (the words behind a published tag are in place: the claimer drained them first)
                    const uint64_t id1 = sd_ld(s + 1), where = sd_ld(s + 2), l = sd_ld(s + 3);
                    if ((int64_t)l == len) {
                        const uint8_t* q = (where & SD_NEW) ? a.data + (sd_off(a, (int64_t)(where & ~SD_NEW)) - a.data_base) : a.heap + where;
                        if (sd_equal(p, q, len)) { id = (int64_t)id1 - 1; break; }
                    }
                    h = (h + 1) & mask;
                    continue;
                }
                if (t == SD_EMPTY) {
                    uint64_t expected = SD_EMPTY;
                    if (__hip_atomic_compare_exchange_strong(s, &expected, SD_LOCKED, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        id = (int64_t)atomicAdd(&s_gnext, 1ULL);
                        sd_st(s + 1, (uint64_t)id + 1);
                        sd_st(s + 2, SD_NEW | (uint64_t)row);
                        sd_st(s + 3, (uint64_t)len);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        sd_st(s, tagv);
                        atomicAdd(&s_new, 1u);
                        break;
                    }
                    continue;   // someone else is claiming this slot: look at it again
                }
                if (t == SD_LOCKED) continue;
                h = (h + 1) & mask;
            }
            a.out[row] = (int32_t)id;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned v = atomicExch(&s_new, 0u);
        if (v) { atomicAdd(&a.d.ctl[2], (unsigned long long)v); atomicAdd(&a.d.ctl[3], (unsigned long long)v); }
        a.progress[blockIdx.x] = it;
    }
}

// a bigger table: every slot moves as it is (the values are distinct: the first free slot of its probe sequence)
__global__ void sd_rehash_kernel(SDict from, SDict to) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint64_t mask = to.cap - 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)from.cap; i += stride) {
        const uint64_t* src = from.slot + (uint64_t)i * 4;
        const uint64_t t = src[0];
        if (t == SD_EMPTY || t == SD_LOCKED) continue;
        uint64_t h = (t ^ (t >> 29)) & mask;
        for (;;) {
            uint64_t* dst = to.slot + h * 4;
            uint64_t expected = SD_EMPTY;
            if (__hip_atomic_compare_exchange_strong(dst, &expected, t, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
                break;
            }
            h = (h + 1) & mask;
        }
    }
}

// ---- the batch's new values -> the heap, in table order ------------------------------------------------------------------
// pass A: per thread (SD_PER consecutive slots) the new values and their bytes; exclusive prefixes inside the workgroup
struct SdCompact {
    SDict d;
    uint32_t* pre_cnt;             // [cap / SD_PER] exclusive prefix inside the workgroup
    unsigned long long* pre_bytes;
    unsigned long long* blk;       // [2 * nblocks]: per workgroup (count, bytes) -> exclusive prefixes after pass B; [2 * nblocks ..]: totals
    int nblocks;
    // pass C
    SdArgs in;
    uint8_t* heap;
    unsigned long long heap_used;
    int32_t* new_ids;
    int32_t* new_lens;
};

__global__ __launch_bounds__(256) void sd_compact_count_kernel(SdCompact c) {
    __shared__ unsigned long long sc[256], sb[256];
    const int tid = threadIdx.x;
    const int64_t t = (int64_t)blockIdx.x * 256 + tid;
    unsigned long long cnt = 0, bytes = 0;
    for (int k = 0; k < SD_PER; k++) {
        const int64_t h = t * SD_PER + k;
        if (h >= (int64_t)c.d.cap) break;
        const uint64_t* s = c.d.slot + (uint64_t)h * 4;
        const uint64_t tag = s[0];
        if (tag == SD_EMPTY || tag == SD_LOCKED || !(s[2] & SD_NEW)) continue;
        cnt++; bytes += s[3];
    }
    sc[tid] = cnt; sb[tid] = bytes;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {   // inclusive scan
        const unsigned long long x = tid >= d ? sc[tid - d] : 0, y = tid >= d ? sb[tid - d] : 0;
        __syncthreads();
        sc[tid] += x; sb[tid] += y;
        __syncthreads();
    }
    if (t * SD_PER < (int64_t)c.d.cap) { c.pre_cnt[t] = (uint32_t)(sc[tid] - cnt); c.pre_bytes[t] = sb[tid] - bytes; }
    if (tid == 255) { c.blk[2 * blockIdx.x] = sc[255]; c.blk[2 * blockIdx.x + 1] = sb[255]; }
}

// pass B: exclusive prefix of the workgroup totals (one workgroup)
__global__ __launch_bounds__(1024) void sd_compact_scan_kernel(unsigned long long* blk, int nblocks) {
    __shared__ unsigned long long sc[1024], sb[1024];
    const int tid = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    const int lo = tid * per, hi = lo + per < nblocks ? lo + per : nblocks;
    unsigned long long c = 0, b = 0;
    for (int i = lo; i < hi; i++) { c += blk[2 * i]; b += blk[2 * i + 1]; }
    sc[tid] = c; sb[tid] = b;
    __syncthreads();
    if (tid == 0) {
        unsigned long long rc = 0, rb = 0;
        for (int i = 0; i < 1024; i++) { const unsigned long long x = sc[i], y = sb[i]; sc[i] = rc; sb[i] = rb; rc += x; rb += y; }
        blk[2 * nblocks] = rc; blk[2 * nblocks + 1] = rb;
    }
    __syncthreads();
    unsigned long long rc = sc[tid], rb = sb[tid];
    for (int i = lo; i < hi; i++) {
        const unsigned long long x = blk[2 * i], y = blk[2 * i + 1];
        blk[2 * i] = rc; blk[2 * i + 1] = rb;
        rc += x; rb += y;
    }
}

// pass C: bytes to the heap, the slot learns its heap offset, (id, length) of the new values in heap order
__global__ __launch_bounds__(256) void sd_compact_move_kernel(SdCompact c) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t * SD_PER >= (int64_t)c.d.cap) return;
    unsigned long long k = c.blk[2 * blockIdx.x] + c.pre_cnt[t];
    unsigned long long at = c.blk[2 * blockIdx.x + 1] + c.pre_bytes[t];
    for (int j = 0; j < SD_PER; j++) {
        const int64_t h = t * SD_PER + j;
        if (h >= (int64_t)c.d.cap) break;
        uint64_t* s = c.d.slot + (uint64_t)h * 4;
        const uint64_t tag = s[0];
        if (tag == SD_EMPTY || tag == SD_LOCKED || !(s[2] & SD_NEW)) continue;
        const int64_t row = (int64_t)(s[2] & ~SD_NEW), len = (int64_t)s[3];
        const uint8_t* p = c.in.data + (sd_off(c.in, row) - c.in.data_base);
        uint8_t* q = c.heap + c.heap_used + at;
        for (int64_t i = 0; i < len; i++) q[i] = p[i];
        s[2] = c.heap_used + at;
        c.new_ids[k] = (int32_t)(s[1] - 1);
        c.new_lens[k] = (int32_t)len;
        k++; at += (unsigned long long)len;
    }
}

int sdict_alloc(SDict* d, uint64_t cap, hipStream_t s) {
    memset(d, 0, sizeof(*d));
    d->cap = cap;
    d->slot = (uint64_t*)pool_alloc((size_t)cap * 32);
    d->ctl = (unsigned long long*)pool_alloc(64);
    if (!d->slot || !d->ctl || hipMemsetAsync(d->slot, 0xFF, (size_t)cap * 32, s) != hipSuccess || hipMemsetAsync(d->ctl, 0, 64, s) != hipSuccess) {
        pool_free(d->slot); pool_free(d->ctl);   // (nothing half-allocated is left behind)
        memset(d, 0, sizeof(*d));
        return set_error("vnm_strdict: table allocation failed");
    }
    return 0;
}
void sdict_free(SDict* d) {
    pool_free(d->slot);
    pool_free(d->ctl);
    memset(d, 0, sizeof(*d));
}
int sdict_grow(SDict* d, uint64_t new_cap, hipStream_t s) {
    SDict nd;
    VNM_TRY(sdict_alloc(&nd, new_cap, s));
    VNM_HIP(hipMemcpyAsync(nd.ctl + 2, d->ctl + 2, 16, hipMemcpyDeviceToDevice, s));   // fill and the batch's new count carry over
    sd_rehash_kernel<<<(int)std::min<int64_t>(((int64_t)d->cap + 255) / 256, (int64_t)device_info().num_cus * 8), 256, 0, s>>>(*d, nd);
    VNM_HIP(hipGetLastError());
    VNM_HIP(hipStreamSynchronize(s));
    sdict_free(d);
    *d = nd;
    return 0;
}

// ---- order-preserving ranks of the dictionary's values (round 5: string / binary ORDER BY keys on the device) -----------------
// Arrow's SortIndices compares utf8 / binary values byte-wise (sort.cpp:22-37).  The dictionary's values are DISTINCT, so their sort
// order is a rank per id; a row's sort key then is rank[code] -- an ordinary int32 column of the numeric sort (equal strings share
// a rank, so the stable sort keeps their row order).  The values are sorted by the existing multi-key radix sort over 8-byte
// BIG-ENDIAN chunks of their bytes (zero padded) with the length as the least significant key (a value that is a prefix of another
// sorts first, "a" before "a\0"): up to 15 chunks + the length per sort call, longer values in several stable rounds from the last
// chunks to the first (an LSD sort over the chunk columns; a round's keys are read through the permutation of the rounds before).
struct SdRankArgs {
    SDict d;
    const uint8_t* heap;
    uint64_t* voff;     // [D] heap offset
    uint32_t* vlen;     // [D]
    int32_t* vid;       // [D]
    unsigned long long* ctl;   // [0] next list position  [1] longest value
};

__global__ __launch_bounds__(256) void sd_list_kernel(SdRankArgs a) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    unsigned long long maxlen = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)a.d.cap; i += stride) {
        const uint64_t* s = a.d.slot + (uint64_t)i * 4;
        const uint64_t tag = s[0];
        if (tag == SD_EMPTY || tag == SD_LOCKED) continue;
        const unsigned long long k = atomicAdd(&a.ctl[0], 1ULL);
        a.voff[k] = s[2]; a.vlen[k] = (uint32_t)s[3]; a.vid[k] = (int32_t)(s[1] - 1);
        if (s[3] > maxlen) maxlen = s[3];
    }
    if (maxlen) atomicMax(&a.ctl[1], maxlen);
}

// key columns of one round: out[c][i] = chunk (j0 + c) of value perm[i] (perm = null: value i); out[nchunks][i] = its length (with_len)
__global__ __launch_bounds__(256) void sd_chunk_kernel(const uint8_t* heap, const uint64_t* voff, const uint32_t* vlen, const int64_t* perm, int64_t n,
                                                       int j0, int nchunks, int with_len, uint64_t* out /* [nchunks + with_len][n] */) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int64_t v = perm ? perm[i] : i;
        const uint8_t* p = heap + voff[v];
        const int64_t len = vlen[v];
        for (int c = 0; c < nchunks; c++) {
            const int64_t at = (int64_t)(j0 + c) * 8;
            uint64_t x = 0;
            for (int b = 0; b < 8; b++) x = (x << 8) | (at + b < len ? (uint64_t)p[at + b] : 0ULL);
            out[(int64_t)c * n + i] = x;
        }
        if (with_len) out[(int64_t)nchunks * n + i] = (uint64_t)len;
    }
}

__global__ __launch_bounds__(256) void sd_rank_scatter_kernel(const int64_t* perm, const int32_t* vid, int64_t n, int32_t* rank_of_id) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) rank_of_id[vid[perm[i]]] = (int32_t)i;
}

// rows' codes -> rows' ranks (a NULL row -- code -1 -- gets 0; its validity bit says NULL)
__global__ __launch_bounds__(256) void sd_code_rank_kernel(const int32_t* codes, const int32_t* rank_of_id, int64_t n, int32_t* out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { const int32_t c = codes[i]; out[i] = c < 0 ? 0 : rank_of_id[c]; }
}

}  // namespace vnm

using namespace vnm;

struct vnm_strdict {
    SDict d{};
    uint8_t* heap = nullptr;
    int64_t heap_cap = 0, heap_used = 0;
    unsigned long long* gnext = nullptr;     // device: first id not handed out yet
    int64_t ids = 0;                         // its host copy after the last encode
    std::vector<int32_t> new_ids, new_lens;  // the values the last encode added, in heap order
    std::vector<uint8_t> new_bytes;
    // An encode that fails between the encode kernel and the compaction leaves slots published as `NEW | row of a batch that is gone`:
    // the next encode would resolve them through ITS batch's offsets (wrong matches, reads out of range).  Such a handle refuses
    // further use.
    bool failed = false;
};

extern "C" {

vnm_strdict* vnm_strdict_create(void) {
    if (ensure_init()) return nullptr;
    return new vnm_strdict();
}

void vnm_strdict_destroy(vnm_strdict* h) {
    if (!h) return;
    if (h->d.slot) sdict_free(&h->d);
    pool_free(h->heap);
    pool_free(h->gnext);
    delete h;
}

int64_t vnm_strdict_ids(vnm_strdict* h) { return h ? h->ids : 0; }

static int strdict_encode_device_impl(vnm_strdict* h, const vnm_dcol* offsets, const int32_t* span_lens, const uint8_t* validity, int64_t validity_offset,
                                      const uint8_t* data, int64_t data_base, int32_t* out_codes, int64_t* n_new, int64_t* new_bytes, void* stream);

int vnm_strdict_encode_device(vnm_strdict* h, const vnm_dcol* offsets, const uint8_t* validity, int64_t validity_offset,
                              const uint8_t* data, int64_t data_base, int32_t* out_codes, int64_t* n_new, int64_t* new_bytes, void* stream) {
    if (h && h->failed) return set_error("vnm_strdict: an earlier encode failed half-way; the dictionary cannot be used any more (create a new one)");
    const int rc = strdict_encode_device_impl(h, offsets, nullptr, validity, validity_offset, data, data_base, out_codes, n_new, new_bytes, stream);
    if (rc && h && h->d.slot) {
        (void)hipStreamSynchronize(as_stream(stream));   // (scratch of the failed call goes back to the pool: nothing may still be running on it)
        h->failed = true;
    }
    return rc;
}

// The same over SPANS of one device byte buffer: row r is the lens[r] bytes at data[starts[r]] (no NULLs).  How vnm_csv_parse_block_ex
// encodes the string columns of a CSV block where they lie in the staged text.
int vnm_strdict_encode_spans(vnm_strdict* h, const int64_t* starts, const int32_t* lens, int64_t n, const uint8_t* data, int32_t* out_codes,
                             int64_t* n_new, int64_t* new_bytes, void* stream) {
    if (h && h->failed) return set_error("vnm_strdict: an earlier encode failed half-way; the dictionary cannot be used any more (create a new one)");
    if (n > 0 && (!starts || !lens || !data)) return set_error("vnm_strdict_encode_spans: null argument");
    vnm_dcol offs{};
    offs.values = (void*)starts; offs.type = VNM_I64; offs.offset = 0; offs.length = n + 1;
    const int rc = strdict_encode_device_impl(h, &offs, lens, nullptr, 0, data, 0, out_codes, n_new, new_bytes, stream);
    if (rc && h && h->d.slot) {
        (void)hipStreamSynchronize(as_stream(stream));
        h->failed = true;
    }
    return rc;
}

int vnm_strdict_last_new(vnm_strdict* h, int64_t* n_new, int64_t* new_bytes) {
    if (!h) return set_error("vnm_strdict_last_new: null handle");
    if (n_new) *n_new = (int64_t)h->new_ids.size();
    if (new_bytes) *new_bytes = (int64_t)h->new_bytes.size();
    return 0;
}

static int strdict_encode_device_impl(vnm_strdict* h, const vnm_dcol* offsets, const int32_t* span_lens, const uint8_t* validity, int64_t validity_offset,
                                      const uint8_t* data, int64_t data_base, int32_t* out_codes, int64_t* n_new, int64_t* new_bytes, void* stream) {
    VNM_TRY(ensure_init());
    if (!h || !offsets || !out_codes) return set_error("vnm_strdict_encode_device: null argument");
    if (offsets->type != VNM_I32 && offsets->type != VNM_I64) return set_error("vnm_strdict_encode_device: offsets must be int32 or int64");
    hipStream_t s = as_stream(stream);
    const int64_t nrows = offsets->length - 1;   // n + 1 offsets
    h->new_ids.clear(); h->new_lens.clear(); h->new_bytes.clear();
    if (n_new) *n_new = 0;
    if (new_bytes) *new_bytes = 0;
    if (nrows <= 0) return 0;
    if (!h->d.slot) {
        VNM_TRY(sdict_alloc(&h->d, 1 << 16, s));
        h->gnext = (unsigned long long*)pool_alloc(64);
        if (!h->gnext) return 1;
        VNM_HIP(hipMemsetAsync(h->gnext, 0, 8, s));
    }
    PoolScope pool;
    SdArgs a{};
    a.offs = offsets->values; a.wide = offsets->type == VNM_I64; a.first = offsets->offset;
    a.lens = span_lens;
    a.valid = validity; a.vfirst = validity_offset;
    a.data = data; a.data_base = data_base;
    a.nrows = nrows;
    a.ntiles = (nrows + SD_TILE - 1) / SD_TILE;
    a.out = out_codes;
    a.gnext = h->gnext;
    a.heap = h->heap;
    int grid = device_info().num_cus * 4;
    if (grid > a.ntiles) grid = (int)a.ntiles;
    a.margin = (int64_t)grid * SD_TILE;
    a.progress = (unsigned int*)pool.take((size_t)grid * 4);
    if (!a.progress) return 1;
    VNM_HIP(hipMemsetAsync(a.progress, 0, (size_t)grid * 4, s));
    VNM_HIP(hipMemsetAsync(h->d.ctl + 3, 0, 8, s));   // values new in this batch
    for (int round = 0;; round++) {
        while ((int64_t)(h->d.cap * 7 / 10) < a.margin + 1) VNM_TRY(sdict_grow(&h->d, h->d.cap * 4, s));
        VNM_HIP(hipMemsetAsync(h->d.ctl, 0, 16, s));   // [1] the flag; [2], [3] persist
        a.d = h->d;
        a.fill_limit = (int64_t)(h->d.cap * 7 / 10);
        {
            KernelTimer timer("strdict_encode", s);
            sd_encode_kernel<<<grid, 256, 0, s>>>(a);
        }
        VNM_HIP(hipGetLastError());
        unsigned long long ctl[4];
        VNM_HIP(hipMemcpyAsync(ctl, h->d.ctl, sizeof(ctl), hipMemcpyDeviceToHost, s));
        VNM_HIP(hipStreamSynchronize(s));
        if (!ctl[1]) break;
        VNM_TRY(sdict_grow(&h->d, h->d.cap * (round >= 1 ? 16 : 4), s));   // the workgroups resume from progress[]
    }
    unsigned long long fresh = 0, gn = 0;
    VNM_HIP(hipMemcpyAsync(&fresh, h->d.ctl + 3, 8, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipMemcpyAsync(&gn, h->gnext, 8, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    h->ids = (int64_t)gn;
    if (gn >= (1ULL << 31)) return set_error("vnm_strdict: more than 2^31 dictionary ids");
    if (!fresh) return 0;
    // ---- the new values -> the heap
    SdCompact c{};
    c.d = h->d;
    const int64_t threads = ((int64_t)h->d.cap + SD_PER - 1) / SD_PER;
    c.nblocks = (int)((threads + 255) / 256);
    c.pre_cnt = (uint32_t*)pool.take((size_t)c.nblocks * 256 * 4);
    c.pre_bytes = (unsigned long long*)pool.take((size_t)c.nblocks * 256 * 8);
    c.blk = (unsigned long long*)pool.take(((size_t)c.nblocks + 1) * 16);
    c.new_ids = (int32_t*)pool.take((size_t)fresh * 4);
    c.new_lens = (int32_t*)pool.take((size_t)fresh * 4);
    if (!c.pre_cnt || !c.pre_bytes || !c.blk || !c.new_ids || !c.new_lens) return 1;
    unsigned long long tot[2] = {0, 0};
    {
        KernelTimer timer("strdict_compact", s);
        sd_compact_count_kernel<<<c.nblocks, 256, 0, s>>>(c);
        sd_compact_scan_kernel<<<1, 1024, 0, s>>>(c.blk, c.nblocks);
    }
    VNM_HIP(hipGetLastError());
    VNM_HIP(hipMemcpyAsync(tot, c.blk + 2 * (size_t)c.nblocks, 16, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    if (tot[0] != fresh) return set_error("vnm_strdict: %llu new values counted, %llu claimed (internal error)", tot[0], fresh);
    if (h->heap_used + (int64_t)tot[1] > h->heap_cap) {
        const int64_t want = std::max<int64_t>((h->heap_used + (int64_t)tot[1]) * 2, 1 << 20);
        uint8_t* nh = (uint8_t*)pool_alloc((size_t)want);
        if (!nh) return 1;
        if (h->heap_used) VNM_HIP(hipMemcpyAsync(nh, h->heap, (size_t)h->heap_used, hipMemcpyDeviceToDevice, s));
        VNM_HIP(hipStreamSynchronize(s));
        pool_free(h->heap);
        h->heap = nh;
        h->heap_cap = want;
    }
    c.in = a;
    c.heap = h->heap;
    c.heap_used = (unsigned long long)h->heap_used;
    {
        KernelTimer timer("strdict_compact", s);
        sd_compact_move_kernel<<<c.nblocks, 256, 0, s>>>(c);
    }
    VNM_HIP(hipGetLastError());
    h->new_ids.resize((size_t)fresh); h->new_lens.resize((size_t)fresh); h->new_bytes.resize((size_t)tot[1]);
    VNM_HIP(hipMemcpyAsync(h->new_ids.data(), c.new_ids, (size_t)fresh * 4, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipMemcpyAsync(h->new_lens.data(), c.new_lens, (size_t)fresh * 4, hipMemcpyDeviceToHost, s));
    if (tot[1]) VNM_HIP(hipMemcpyAsync(h->new_bytes.data(), h->heap + h->heap_used, (size_t)tot[1], hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    h->heap_used += (int64_t)tot[1];
    if (n_new) *n_new = (int64_t)fresh;
    if (new_bytes) *new_bytes = (int64_t)tot[1];
    return 0;
}

int vnm_strdict_encode(vnm_strdict* h, const void* offsets_host, int offsets_are_64, const uint8_t* data_host, const uint8_t* validity_host,
                       int64_t offset, int64_t length, int32_t* out_codes_host, int64_t* n_new, int64_t* new_bytes, void* stream) {
    VNM_TRY(ensure_init());
    if (!h || (length > 0 && (!offsets_host || !out_codes_host))) return set_error("vnm_strdict_encode: null argument");
    if (n_new) *n_new = 0;
    if (new_bytes) *new_bytes = 0;
    h->new_ids.clear(); h->new_lens.clear(); h->new_bytes.clear();
    if (length <= 0) return 0;
    hipStream_t s = as_stream(stream);
    int64_t b0, b1;
    if (offsets_are_64) { b0 = ((const int64_t*)offsets_host)[offset]; b1 = ((const int64_t*)offsets_host)[offset + length]; }
    else { b0 = ((const int32_t*)offsets_host)[offset]; b1 = ((const int32_t*)offsets_host)[offset + length]; }
    if (b1 < b0) return set_error("vnm_strdict_encode: offsets decrease");
    if (b1 > b0 && !data_host) return set_error("vnm_strdict_encode: null data buffer");
    // the column crosses PCIe once: n + 1 offsets, the bytes of the validity bitmap that cover the rows, the bytes the rows refer to
    struct Cols { vnm_dcol offs{}, data{}, bits{}; ~Cols() { vnm_free_column(&offs); vnm_free_column(&data); vnm_free_column(&bits); } } st;
    VNM_TRY(vnm_stage_column(offsets_host, nullptr, offset, length + 1, offsets_are_64 ? VNM_I64 : VNM_I32, &st.offs, stream));
    if (validity_host) VNM_TRY(vnm_stage_column(validity_host, nullptr, offset >> 3, ((offset + length + 7) >> 3) - (offset >> 3), VNM_U8, &st.bits, stream));
    if (b1 > b0) VNM_TRY(vnm_stage_column(data_host, nullptr, b0, b1 - b0, VNM_U8, &st.data, stream));
    int32_t* codes = (int32_t*)pool_alloc((size_t)length * 4);
    if (!codes) return 1;
    int rc = vnm_strdict_encode_device(h, &st.offs, validity_host ? (const uint8_t*)st.bits.values : nullptr, offset & 7,
                                       (const uint8_t*)st.data.values, b0, codes, n_new, new_bytes, stream);
    if (!rc && (hipMemcpyAsync(out_codes_host, codes, (size_t)length * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess))
        rc = set_error("vnm_strdict_encode: copying the codes back failed");
    pool_free(codes);
    return rc;
}

int vnm_strdict_fetch_new(vnm_strdict* h, int32_t* ids_host, int32_t* lens_host, uint8_t* bytes_host) {
    if (!h) return set_error("vnm_strdict_fetch_new: null handle");
    if (!h->new_ids.empty()) {
        memcpy(ids_host, h->new_ids.data(), h->new_ids.size() * 4);
        memcpy(lens_host, h->new_lens.data(), h->new_lens.size() * 4);
    }
    if (!h->new_bytes.empty()) memcpy(bytes_host, h->new_bytes.data(), h->new_bytes.size());
    h->new_ids.clear(); h->new_lens.clear(); h->new_bytes.clear();   // (handed over once: a second fetch without an encode in between brings nothing)
    return 0;
}

// Order-preserving ranks of every value in the dictionary: out_rank_of_id[id] = position of the value in ascending byte-wise order
// (device array of vnm_strdict_ids(h) int32s; ids that were never handed out are left untouched).
int vnm_strdict_ranks_device(vnm_strdict* h, int32_t* out_rank_of_id, void* stream) {
    VNM_TRY(ensure_init());
    if (!h || !out_rank_of_id) return set_error("vnm_strdict_ranks_device: null argument");
    if (h->failed) return set_error("vnm_strdict: an earlier encode failed half-way; the dictionary cannot be used any more (create a new one)");
    if (!h->d.slot) return 0;
    hipStream_t s = as_stream(stream);
    unsigned long long fill = 0;
    VNM_HIP(hipMemcpyAsync(&fill, h->d.ctl + 2, 8, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    const int64_t D = (int64_t)fill;
    if (D == 0) return 0;
    PoolScope pool;
    SdRankArgs a{};
    a.d = h->d; a.heap = h->heap;
    a.voff = (uint64_t*)pool.take((size_t)D * 8);
    a.vlen = (uint32_t*)pool.take((size_t)D * 4);
    a.vid = (int32_t*)pool.take((size_t)D * 4);
    a.ctl = (unsigned long long*)pool.take(64);
    int64_t* perm = (int64_t*)pool.take((size_t)D * 8);
    int64_t* perm2 = (int64_t*)pool.take((size_t)D * 8);
    int64_t* perm3 = (int64_t*)pool.take((size_t)D * 8);
    if (!a.voff || !a.vlen || !a.vid || !a.ctl || !perm || !perm2 || !perm3) return 1;
    VNM_HIP(hipMemsetAsync(a.ctl, 0, 64, s));
    const int grid = (int)std::min<int64_t>(((int64_t)h->d.cap + 255) / 256, (int64_t)device_info().num_cus * 8);
    sd_list_kernel<<<grid, 256, 0, s>>>(a);
    VNM_HIP(hipGetLastError());
    unsigned long long ctl[2];
    VNM_HIP(hipMemcpyAsync(ctl, a.ctl, 16, hipMemcpyDeviceToHost, s));
    VNM_HIP(hipStreamSynchronize(s));
    if ((int64_t)ctl[0] != D) return set_error("vnm_strdict_ranks_device: %llu values listed, %lld in the table (internal error)", ctl[0], (long long)D);
    const int m = (int)((ctl[1] + 7) / 8);                    // chunks of the longest value
    route_note("sort:string_key_ranks", "%lld distinct values, longest %llu bytes: %d round(s) of up to 15 chunk keys + the length", (long long)D, ctl[1], m == 0 ? 1 : (m + 14) / 15);
    constexpr int PER = 15;                                   // chunk keys per sort call (+ the length: 16 keys)
    const int rounds = m == 0 ? 1 : (m + PER - 1) / PER;
    uint64_t* keys = (uint64_t*)pool.take((size_t)D * 8 * (size_t)(std::min(m, PER) + 1));
    if (!keys) return 1;
    const int g2 = (int)std::min<int64_t>((D + 255) / 256, (int64_t)device_info().num_cus * 8);
    bool have_perm = false;
    for (int r = 0; r < rounds; r++) {                        // least significant round first: the last chunks and the length
        const int hi = m - r * PER, lo = std::max(0, hi - PER), nc = hi - lo;
        const int with_len = r == 0 ? 1 : 0;
        sd_chunk_kernel<<<g2, 256, 0, s>>>(h->heap, a.voff, a.vlen, have_perm ? perm : nullptr, D, lo, nc, with_len, keys);
        VNM_HIP(hipGetLastError());
        vnm_dcol kc[16];
        int orders[16];
        const int nk = nc + with_len;
        for (int k = 0; k < nk; k++) {
            memset(&kc[k], 0, sizeof(vnm_dcol));
            kc[k].values = keys + (size_t)k * D; kc[k].type = VNM_U64; kc[k].length = D;
            orders[k] = VNM_ASC;
        }
        VNM_TRY(vnm_sort_indices(nk, kc, orders, D, 0, perm2, stream));
        if (!have_perm) std::swap(perm, perm2);
        else {                                                // positions in this round's order -> value indices
            vnm_dcol pc{};
            pc.values = perm; pc.type = VNM_I64; pc.length = D;
            VNM_TRY(vnm_take(&pc, perm2, D, perm3, nullptr, stream));
            std::swap(perm, perm3);
        }
        have_perm = true;
    }
    sd_rank_scatter_kernel<<<g2, 256, 0, s>>>(perm, a.vid, D, out_rank_of_id);
    VNM_HIP(hipGetLastError());
    VNM_HIP(hipStreamSynchronize(s));
    return 0;
}

// codes (vnm_strdict_encode*) -> the rows' ranks under rank_of_id (vnm_strdict_ranks_device); all device buffers
int vnm_strdict_codes_to_ranks(const int32_t* codes, const int32_t* rank_of_id, int64_t n, int32_t* out_ranks, void* stream) {
    VNM_TRY(ensure_init());
    if (n <= 0) return 0;
    if (!codes || !rank_of_id || !out_ranks) return set_error("vnm_strdict_codes_to_ranks: null argument");
    sd_code_rank_kernel<<<(int)std::min<int64_t>((n + 255) / 256, (int64_t)device_info().num_cus * 8), 256, 0, as_stream(stream)>>>(codes, rank_of_id, n, out_ranks);
    VNM_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
