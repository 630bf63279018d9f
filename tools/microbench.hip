// Microbenchmarks that decide the aggregate design on MI355X (run via gpurun):
//   stream read / copy bandwidth, random global atomics (agent / workgroup scope) vs table size,
//   LDS atomic throughput.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

__global__ void k_read(const double2* __restrict__ in, int64_t n2, double* out) {
    double acc = 0;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        double2 v = in[i]; acc += v.x + v.y;
    }
    if (acc == 1.234567) out[0] = acc;
}
__global__ void k_copy(const double2* __restrict__ in, double2* __restrict__ o, int64_t n2) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) o[i] = in[i];
}

template <int SCOPE>
__global__ void k_atomic_f64(double* tab, uint64_t mask, int64_t n, int per_row) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t h = mix64((uint64_t)i) & mask;
        for (int j = 0; j < per_row; j++)
            __hip_atomic_fetch_add(&tab[(h + (uint64_t)j * (mask + 1))], 1.0, __ATOMIC_RELAXED, SCOPE);
    }
}
template <int SCOPE>
__global__ void k_atomic_u64(unsigned long long* tab, uint64_t mask, int64_t n, int per_row) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t h = mix64((uint64_t)i) & mask;
        for (int j = 0; j < per_row; j++)
            __hip_atomic_fetch_add(&tab[(h + (uint64_t)j * (mask + 1))], 1ULL, __ATOMIC_RELAXED, SCOPE);
    }
}
// AoS slot: 32 B {key, sum, cnt, pad}: CAS-less probe (load key) + 2 atomics in the same 32B sector
struct __attribute__((aligned(32))) Slot { unsigned long long key; double sum; unsigned long long cnt; unsigned long long pad; };
__global__ void k_atomic_aos(Slot* tab, uint64_t mask, int64_t n) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t h = mix64((uint64_t)i) & mask;
        unsigned long long k = __hip_atomic_load(&tab[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == 0) __hip_atomic_compare_exchange_strong(&tab[h].key, &k, (unsigned long long)h + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&tab[h].sum, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&tab[h].cnt, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// random plain (non-atomic) RMW for comparison: load+store
__global__ void k_plain_rmw(double* tab, uint64_t mask, int64_t n) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t h = mix64((uint64_t)i) & mask;
        tab[h] += 1.0;
    }
}
__global__ void k_random_read(const double* tab, uint64_t mask, int64_t n, double* out) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x; double acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t h = mix64((uint64_t)i) & mask; acc += tab[h];
    }
    if (acc == 1.2345) out[0] = acc;
}

// LDS atomics: 256 threads, table of S slots (f64 add + u64 add), random slots
__global__ void k_lds_atomic(int64_t iters, int slots_mask, double* out) {
    extern __shared__ double lds[];
    double* sum = lds; unsigned long long* cnt = (unsigned long long*)(lds + slots_mask + 1);
    for (int i = threadIdx.x; i <= slots_mask; i += blockDim.x) { sum[i] = 0; cnt[i] = 0; }
    __syncthreads();
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = 0; i < iters; i++) {
        x = x * 6364136223846793005ULL + 1442695040888963407ULL;
        int h = (int)(x >> 33) & slots_mask;
        __hip_atomic_fetch_add(&sum[h], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&cnt[h], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (threadIdx.x == 0 && sum[0] == -1.0) out[0] = sum[0];
}

template <typename F>
float timeit(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d MHz mem %.1f GB L2 %d\n", p.name, p.multiProcessorCount, p.clockRate / 1000, p.totalGlobalMem / 1e9, p.l2CacheSize);
    const int64_t NB = 8LL << 30;  // 8 GiB stream buffer
    double2 *in, *out; double* sink;
    CK(hipMalloc(&in, NB)); CK(hipMalloc(&out, NB)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(in, 0, NB));
    for (int blocks : {1024, 2048, 4096, 8192}) {
        float ms = timeit([&] { k_read<<<blocks, 256>>>(in, NB / 16, sink); });
        printf("read   8GiB grid %5d x256: %.3f ms  %.2f TB/s\n", blocks, ms, NB / ms / 1e9);
    }
    for (int blocks : {2048, 8192}) {
        float ms = timeit([&] { k_copy<<<blocks, 256>>>(in, out, NB / 16); });
        printf("copy   8GiB grid %5d x256: %.3f ms  %.2f TB/s (r+w)\n", blocks, ms, 2.0 * NB / ms / 1e9);
    }
    CK(hipFree(out));
    // random atomics vs table size
    const int64_t NROWS = 1LL << 28;  // 268M updates
    double* tab; CK(hipMalloc(&tab, 8ULL << 30));
    for (int lg : {10, 16, 19, 22, 24, 26, 28, 29}) {
        uint64_t slots = 1ULL << lg;
        CK(hipMemset(tab, 0, slots * 8 * 2));
        float ms = timeit([&] { k_atomic_f64<__HIP_MEMORY_SCOPE_AGENT><<<4096, 256>>>(tab, slots - 1, NROWS, 1); }, 3);
        float ms2 = timeit([&] { k_atomic_f64<__HIP_MEMORY_SCOPE_AGENT><<<4096, 256>>>(tab, slots - 1, NROWS, 2); }, 3);
        float ms3 = timeit([&] { k_atomic_f64<__HIP_MEMORY_SCOPE_WORKGROUP><<<4096, 256>>>(tab, slots - 1, NROWS, 1); }, 3);
        float ms4 = timeit([&] { k_atomic_u64<__HIP_MEMORY_SCOPE_AGENT><<<4096, 256>>>((unsigned long long*)tab, slots - 1, NROWS, 1); }, 3);
        float ms5 = timeit([&] { k_plain_rmw<<<4096, 256>>>(tab, slots - 1, NROWS); }, 3);
        float ms6 = timeit([&] { k_random_read<<<4096, 256>>>(tab, slots - 1, NROWS, sink); }, 3);
        printf("table 2^%d x8B (%.1f MB): f64 agent x1 %.2f G/s | x2(SoA) %.2f Grows/s | f64 wg-scope %.2f G/s | u64 agent %.2f G/s | plain rmw %.2f G/s | random read %.2f G/s\n",
               lg, slots * 8 / 1e6, NROWS / ms / 1e6, NROWS / ms2 / 1e6, NROWS / ms3 / 1e6, NROWS / ms4 / 1e6, NROWS / ms5 / 1e6, NROWS / ms6 / 1e6);
    }
    for (int lg : {10, 16, 20, 24, 27}) {
        uint64_t slots = 1ULL << lg;
        CK(hipMemset(tab, 0, slots * 32));
        float ms = timeit([&] { k_atomic_aos<<<4096, 256>>>((Slot*)tab, slots - 1, NROWS); }, 3);
        printf("AoS 32B slot table 2^%d (%.1f MB): probe+CAS+2 atomics %.2f Grows/s\n", lg, slots * 32 / 1e6, NROWS / ms / 1e6);
    }
    // LDS atomics
    for (int lg : {3, 6, 10, 12}) {
        int slots = 1 << lg; int64_t iters = 4096;
        float ms = timeit([&] { k_lds_atomic<<<2048, 256, slots * 16>>>(iters, slots - 1, sink); }, 3);
        printf("LDS atomics table %d slots: %.2f G row-updates/s (2 atomics each)\n", slots, 2048.0 * 256 * iters / ms / 1e6);
    }
    return 0;
}
