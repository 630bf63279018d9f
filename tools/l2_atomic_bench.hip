// VERDICT r05 "next" #1(a): the one mechanism nobody measured.  Can ONE scatter level (512 partitions of 2^18 codes) be followed
// by a final pass whose {f64 sum, u32 count} table (2 + 1 MB per partition) lives in the L2 of ONE XCD and is updated with global
// atomics that are NOT agent scope (no sc1: executed by that XCD's L2 -- profiles/microbench_r01.txt measured agent-scope atomics
// only: 11.6 G/s, flat from 2 MB to 4 GB, i.e. they are served beyond the L2)?
//
// Every workgroup reads HW_REG_XCC_ID and pulls (partition, chunk) work items from ITS XCD's queue; XCD x owns the partitions
// p = x (mod 8) and walks them in order, so that at any time the ~32 workgroups of an XCD update one (at a boundary: two) tables.
// Correctness needs no placement promise from HIP: a workgroup knows the XCD it physically runs on, and only workgroups of that
// XCD ever touch a partition's table inside the launch (the L2s of different XCDs are not coherent for such atomics: mode 5
// shows what happens when the rule is broken).
//
// modes: 0 read only (the ceiling)                     1 workgroup-scope f64 add + u32 add        2 workgroup-scope f64 add only
//        3 agent-scope f64 add + u32 add               4 agent-scope f64 add only
//        5 mode 1 with partitions dealt by blockIdx instead of by XCD (tables shared between XCDs: expected WRONG)
//        6 LDS: the same entries into 2^13-slot LDS tables by slot = code & 8191 (non-returning ds_add_f64 + ds_add_u32) -- what a
//          final pass without compensation terms runs at        7 LDS, f64 add only
//
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o tools/l2_atomic_bench tools/l2_atomic_bench.hip
// run:   tools/l2_atomic_bench [entries per partition, default 2^20] [partitions, default 512] [code bits, default 18]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u2_t __attribute__((ext_vector_type(2)));
constexpr int BLOCK = 1024;
constexpr int PER = 8;                      // entries per thread and chunk
constexpr int CHUNK = BLOCK * PER;

struct Args {
    const double* vals;       // [P][E]
    const uint32_t* codes;    // [P][E]
    int64_t E;
    int P, bits;
    double* sum;              // [P][2^bits]
    uint32_t* cnt;            // [P][2^bits]
    unsigned int* queue;      // [8] next work item per XCD, [8..15] workgroups seen per XCD
    int mode;
};

template <int SCOPE>
__device__ __forceinline__ void add_f64(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE>
__device__ __forceinline__ void add_u32(uint32_t* p, uint32_t v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, SCOPE); }

template <int MODE>
__global__ __launch_bounds__(BLOCK) void l2_kernel(Args a) {
    __shared__ unsigned int s_item[2];
    __shared__ double lsum[(MODE >= 6) ? 8192 : 1];
    __shared__ uint32_t lcnt[(MODE >= 6) ? 8192 : 1];
    const int tid = threadIdx.x;
    const int xcd = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u);   // HW_REG_XCC_ID[3:0]
    const int q = MODE == 5 ? (int)(blockIdx.x & 7u) ^ 5 : xcd;    // mode 5: a queue that is NOT this workgroup's XCD
    const int64_t chunks = (a.E + CHUNK - 1) / CHUNK;
    const int64_t nitems = (int64_t)((a.P - q + 7) / 8) * chunks;
    if (MODE >= 6) { for (int i = tid; i < 8192; i += BLOCK) { lsum[i] = 0.0; lcnt[i] = 0; } }
    if (tid == 0) { atomicAdd(&a.queue[8 + xcd], 1u); s_item[0] = atomicAdd(&a.queue[q], 1u); }
    __syncthreads();
    int ph = 0;
    double acc = 0.0;
    for (;;) {
        const int64_t w = s_item[ph];
        if (w >= nitems) break;
        if (tid == 0) s_item[ph ^ 1] = atomicAdd(&a.queue[q], 1u);      // the next item is on its way while this one is processed
        const int p = q + 8 * (int)(w / chunks);
        const int64_t e0 = (w % chunks) * CHUNK;
        const double* v = a.vals + (int64_t)p * a.E + e0;
        const uint32_t* c = a.codes + (int64_t)p * a.E + e0;
        double* ts = a.sum + ((int64_t)p << a.bits);
        uint32_t* tc = a.cnt + ((int64_t)p << a.bits);
        const int64_t left = a.E - e0;
        d2_t vv[PER / 2]; u2_t cc[PER / 2];
#pragma unroll
        for (int u = 0; u < PER / 2; u++) {
            const int64_t i = (int64_t)u * 2 * BLOCK + 2 * tid;
            if (i + 1 < left) {
                vv[u] = __builtin_nontemporal_load((const d2_t*)(v + i));
                cc[u] = __builtin_nontemporal_load((const u2_t*)(c + i));
            } else { vv[u] = d2_t{0, 0}; cc[u] = u2_t{0xFFFFFFFFu, 0xFFFFFFFFu}; }
        }
#pragma unroll
        for (int u = 0; u < PER / 2; u++) {
#pragma unroll
            for (int el = 0; el < 2; el++) {
                const uint32_t code = el ? cc[u].y : cc[u].x;
                const double x = el ? vv[u].y : vv[u].x;
                if (code == 0xFFFFFFFFu) continue;
                if (MODE == 0) acc += x + (double)code;
                if (MODE == 1 || MODE == 5) { add_f64<__HIP_MEMORY_SCOPE_WORKGROUP>(&ts[code], x); add_u32<__HIP_MEMORY_SCOPE_WORKGROUP>(&tc[code], 1u); }
                if (MODE == 2) add_f64<__HIP_MEMORY_SCOPE_WORKGROUP>(&ts[code], x);
                if (MODE == 3) { add_f64<__HIP_MEMORY_SCOPE_AGENT>(&ts[code], x); add_u32<__HIP_MEMORY_SCOPE_AGENT>(&tc[code], 1u); }
                if (MODE == 4) add_f64<__HIP_MEMORY_SCOPE_AGENT>(&ts[code], x);
                if (MODE == 6) { add_f64<__HIP_MEMORY_SCOPE_WORKGROUP>(&lsum[code & 8191], x); add_u32<__HIP_MEMORY_SCOPE_WORKGROUP>(&lcnt[code & 8191], 1u); }
                if (MODE == 7) add_f64<__HIP_MEMORY_SCOPE_WORKGROUP>(&lsum[code & 8191], x);
            }
        }
        __syncthreads();
        ph ^= 1;
    }
    if (MODE == 0 && acc == 1.2345e-300) a.sum[0] = acc;
    if (MODE >= 6) {   // (the LDS tables leave the kernel so that the adds cannot be dropped; not compared)
        __syncthreads();
        for (int i = tid; i < 8192; i += BLOCK) if (lsum[i] == 1.2345e-300 && lcnt[i] == 77) a.sum[i] = lsum[i];
    }
}

__global__ void gen_kernel(double* vals, uint32_t* codes, int64_t n, uint32_t mask) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ULL;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
        codes[i] = (uint32_t)x & mask;
        vals[i] = (double)((x >> 40) & 16383) / 128.0;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int64_t E = argc > 1 ? atoll(argv[1]) : (1LL << 20);
    const int P = argc > 2 ? atoi(argv[2]) : 512;
    const int bits = argc > 3 ? atoi(argv[3]) : 18;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int64_t n = (int64_t)P * E;
    printf("device: %s, %d CUs; %d partitions x %lld entries = %.3e entries (%.2f GB of (value, code)); tables %d x 2^%d x 12 B = %.2f GB\n", prop.name, cus, P,
           (long long)E, (double)n, n * 12 / 1e9, P, bits, (double)P * (1 << bits) * 12 / 1e9);
    double* vals; uint32_t* codes; CK(hipMalloc(&vals, (size_t)n * 8)); CK(hipMalloc(&codes, (size_t)n * 4));
    gen_kernel<<<2048, 256>>>(vals, codes, n, (1u << bits) - 1u);
    double* sum; uint32_t* cnt; unsigned int* queue;
    const size_t slots = (size_t)P << bits;
    CK(hipMalloc(&sum, slots * 8)); CK(hipMalloc(&cnt, slots * 4)); CK(hipMalloc(&queue, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // host reference for two partitions (codes and values are a pure function of the index)
    const int check_p[2] = {0, P - 1};
    std::vector<double> rsum[2]; std::vector<uint32_t> rcnt[2];
    for (int k = 0; k < 2; k++) {
        rsum[k].assign((size_t)1 << bits, 0.0); rcnt[k].assign((size_t)1 << bits, 0);
        for (int64_t j = 0; j < E; j++) {
            uint64_t x = (uint64_t)((int64_t)check_p[k] * E + j) * 0x9E3779B97F4A7C15ULL;
            x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
            const uint32_t c = (uint32_t)x & ((1u << bits) - 1u);
            rsum[k][c] += (double)((x >> 40) & 16383) / 128.0; rcnt[k][c]++;
        }
    }
    Args a{vals, codes, E, P, bits, sum, cnt, queue, 0};
    const char* names[8] = {"read only", "L2 wg-scope f64+u32", "L2 wg-scope f64", "agent-scope f64+u32", "agent-scope f64", "wg-scope, tables shared across XCDs", "LDS f64+u32 (non-returning)", "LDS f64"};
    for (int grid_mul = 1; grid_mul <= 2; grid_mul++) {
        for (int mode = 0; mode < 8; mode++) {
            float best = 1e30f;
            std::vector<unsigned int> hq(16);
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemset(sum, 0, slots * 8)); CK(hipMemset(cnt, 0, slots * 4)); CK(hipMemset(queue, 0, 64));
                CK(hipDeviceSynchronize());
                a.mode = mode;
                CK(hipEventRecord(e0));
                const int grid = cus * grid_mul;
                switch (mode) {
                    case 0: l2_kernel<0><<<grid, BLOCK>>>(a); break;
                    case 1: l2_kernel<1><<<grid, BLOCK>>>(a); break;
                    case 2: l2_kernel<2><<<grid, BLOCK>>>(a); break;
                    case 3: l2_kernel<3><<<grid, BLOCK>>>(a); break;
                    case 4: l2_kernel<4><<<grid, BLOCK>>>(a); break;
                    case 5: l2_kernel<5><<<grid, BLOCK>>>(a); break;
                    case 6: l2_kernel<6><<<grid, BLOCK>>>(a); break;
                    default: l2_kernel<7><<<grid, BLOCK>>>(a); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CK(hipMemcpy(hq.data(), queue, 64, hipMemcpyDeviceToHost));
            }
            // verify (table modes): the two reference partitions, every slot
            const char* verdict = "-";
            if (mode >= 1 && mode <= 5) {
                int64_t bad = 0;
                std::vector<double> hs((size_t)1 << bits); std::vector<uint32_t> hc((size_t)1 << bits);
                for (int k = 0; k < 2; k++) {
                    CK(hipMemcpy(hs.data(), sum + ((size_t)check_p[k] << bits), hs.size() * 8, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(hc.data(), cnt + ((size_t)check_p[k] << bits), hc.size() * 4, hipMemcpyDeviceToHost));
                    const bool with_cnt = mode == 1 || mode == 3 || mode == 5;
                    for (size_t s = 0; s < hs.size(); s++) if (hs[s] != rsum[k][s] || (with_cnt && hc[s] != rcnt[k][s])) bad++;
                }
                verdict = bad ? "WRONG" : "ok";
                if (bad) printf("    (%lld slots differ)\n", (long long)bad);
            }
            printf("grid %4d  mode %d  %-38s %8.3f ms  %7.1f G entries/s  %6.2f TB/s of entries  verify %s   wg per xcd:", cus * grid_mul, mode, names[mode], best,
                   n / (best * 1e6), n * 12.0 / (best * 1e9), verdict);
            for (int x = 0; x < 8; x++) printf(" %u", hq[8 + x]);
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}
