// Store-side microbenchmarks (r02): why does a plain copy reach 4.8 TB/s where the guide quotes 6.29, and what do the
// partition passes pay for SHORT write runs?  Not part of the product.  hipcc --offload-arch=gfx950 -O3 copybench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

// MODE 0 plain, 1 nontemporal store, 2 nontemporal load + store
template <int U, int MODE>
__global__ __launch_bounds__(256) void k_copy(const v2d* __restrict__ in, v2d* __restrict__ o, int64_t n2) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride * U) {
        v2d v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t j = i + u * stride;
            if (j < n2) v[u] = MODE == 2 ? __builtin_nontemporal_load(&in[j]) : in[j];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t j = i + u * stride;
            if (j < n2) { if (MODE >= 1) __builtin_nontemporal_store(v[u], &o[j]); else o[j] = v[u]; }
        }
    }
}
// block-contiguous variant: each block owns a contiguous chunk (tile loop), U x 16 B in flight per lane
template <int U, int MODE>
__global__ __launch_bounds__(256) void k_copy_tiles(const v2d* __restrict__ in, v2d* __restrict__ o, int64_t n2) {
    const int64_t tile = 256 * U;
    const int64_t ntiles = (n2 + tile - 1) / tile;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        v2d v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int64_t j = t * tile + u * 256 + threadIdx.x; if (j < n2) v[u] = in[j]; }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t j = t * tile + u * 256 + threadIdx.x;
            if (j < n2) { if (MODE >= 1) __builtin_nontemporal_store(v[u], &o[j]); else o[j] = v[u]; }
        }
    }
}
// 2 : 1 read : write (the filter's ratio at s = 0.5): read 16 B, write 8 B
__global__ __launch_bounds__(256) void k_read2_write1(const v2d* __restrict__ in, double* __restrict__ o, int64_t n2) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride * 4) {
        v2d v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int64_t j = i + u * stride; if (j < n2) v[u] = in[j]; }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int64_t j = i + u * stride; if (j < n2) o[j] = v[u].x + v[u].y; }
    }
}

// Run scatter: a block reads its contiguous chunk and writes items in RUNS of L items round-robin over P private regions
// (what the copy-out of a partition pass does).  W = item width in 8-byte words (1 or 2).
template <int W>
__global__ __launch_bounds__(1024) void k_runs(const double* __restrict__ in, double* __restrict__ o, int64_t n_items, int L, int P,
                                              int64_t region_cap) {
    const int64_t per_block = n_items / gridDim.x;
    const int64_t base = (int64_t)blockIdx.x * per_block;
    double* my = o + (int64_t)blockIdx.x * P * region_cap * W;
    for (int64_t i0 = 0; i0 < per_block; i0 += 1024 * 8) {
        double v[8][W];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int64_t i = i0 + u * 1024 + threadIdx.x;
            if (i < per_block)
                for (int w = 0; w < W; w++) v[u][w] = in[(base + i) * W + w];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int64_t i = i0 + u * 1024 + threadIdx.x;
            if (i < per_block) {
                const int64_t run = i / L, within = i % L;      // run r goes to region r % P, at offset (r / P) * L
                const int64_t p = run % P, at = (run / P) * L + within;
                for (int w = 0; w < W; w++) my[(p * region_cap + at) * W + w] = v[u][w];
            }
        }
    }
}

// Partition-pass model with runs of VARIABLE length (27 +- 5 items of 8 bytes per partition and tile, as pass 1 of the dense
// path at s = 0.5 with 128 partitions), i.e. every run starts and ends in the middle of a cache line.
//   SHARED = false: the regions are private to the workgroup (cursors in LDS); the line a run ends in is continued by the SAME
//                   workgroup one tile (~10 us) later;
//   SHARED = true : one region per partition, a tile reserves its space with a returning atomic per partition; the line a run
//                   ends in is continued by whichever tile reserves next, i.e. at about the same time.
// Each tile also reads 8192 x 16 bytes of input.
template <bool SHARED, bool PERSIST>
__global__ __launch_bounds__(1024) void k_varruns(const v2d* __restrict__ in, double* __restrict__ o, int64_t ntiles, int P, int64_t cap,
                                                  unsigned long long* gcur) {
    __shared__ unsigned int cnt[256], off[257], cur[256];
    __shared__ unsigned long long gbase[256];
    __shared__ unsigned char part_of[8192];
    const int tid = threadIdx.x;
    if (tid < 256) cur[tid] = 0;
    __syncthreads();
    for (int64_t t = blockIdx.x; t < ntiles; t += (PERSIST ? gridDim.x : ntiles)) {
        v2d acc = {0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 8; u++) { const v2d x = in[t * 8192 + u * 1024 + tid]; acc.x += x.x; acc.y += x.y; }
        if (tid < P) {
            unsigned int h = (unsigned int)(t * 2654435761u) ^ (unsigned int)(tid * 40503u);
            h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
            cnt[tid] = 22 + h % 11;
        }
        __syncthreads();
        if (tid == 0) { unsigned int run = 0; for (int p = 0; p < P; p++) { off[p] = run; run += cnt[p]; } off[P] = run; }
        __syncthreads();
        if (tid < P) {
            for (unsigned int k = 0; k < cnt[tid]; k++) part_of[off[tid] + k] = (unsigned char)tid;
            if (SHARED) gbase[tid] = atomicAdd(&gcur[tid], (unsigned long long)cnt[tid]);
        }
        __syncthreads();
        const unsigned int total = off[P];
        for (unsigned int i = tid; i < total; i += 1024) {
            const int p = part_of[i];
            const int64_t at = SHARED ? (int64_t)p * cap + (int64_t)gbase[p] + (i - off[p])
                                      : ((int64_t)blockIdx.x * P + p) * cap + cur[p] + (i - off[p]);
            o[at] = acc.x + acc.y + (double)i;
        }
        __syncthreads();
        if (!SHARED && tid < P) cur[tid] += cnt[tid];
        __syncthreads();
    }
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

int main(int argc, char** argv) {
    const int64_t NB = 8LL << 30;
    v2d *in, *out; CK(hipMalloc(&in, NB)); CK(hipMalloc(&out, NB + (1 << 20)));
    CK(hipMemset(in, 0, NB)); CK(hipMemset(out, 0, NB));
    const int64_t n2 = NB / 16;
    if (argc > 1 && !strcmp(argv[1], "varruns")) {
        // 8 GiB of input = 65536 tiles of 8192 x 16 B; ~27 x 128 items of 8 B written per tile
        const int64_t ntiles = NB / (8192 * 16);
        const int P = 128;
        unsigned long long* gcur; CK(hipMalloc(&gcur, 256 * 8));
        const double wbytes = (double)ntiles * P * 27.0 * 8.0;
        for (int G : {256, 512}) {
            const int64_t cap = (ntiles / G + 2) * 34;                   // private region: tiles of this workgroup x max run
            if ((int64_t)G * P * cap * 8 > NB) { printf("private regions do not fit\n"); continue; }
            float ms = timeit([&] { k_varruns<false, true><<<G, 1024>>>(in, (double*)out, ntiles, P, cap, gcur); });
            printf("varruns private regions, %d persistent workgroups: %.3f ms  %.2f TB/s (r+w)\n", G, ms, (NB + wbytes) / ms / 1e9);
        }
        const int64_t scap = ntiles * 34;
        for (int G : {256, 512}) {
            float ms = timeit([&] { CK(hipMemsetAsync(gcur, 0, 256 * 8)); k_varruns<true, true><<<G, 1024>>>(in, (double*)out, ntiles, P, scap, gcur); });
            printf("varruns shared regions (atomic reservation), %d persistent workgroups: %.3f ms  %.2f TB/s (r+w)\n", G, ms, (NB + wbytes) / ms / 1e9);
        }
        {
            float ms = timeit([&] { CK(hipMemsetAsync(gcur, 0, 256 * 8)); k_varruns<true, false><<<(int)ntiles, 1024>>>(in, (double*)out, ntiles, P, scap, gcur); });
            printf("varruns shared regions, one workgroup per tile: %.3f ms  %.2f TB/s (r+w)\n", ms, (NB + wbytes) / ms / 1e9);
        }
        return 0;
    }
#define RUN(NAME, KERN, G)  { float ms = timeit([&] { KERN<<<G, 256>>>(in, out, n2); }); printf("%-34s grid %5d: %.3f ms  %.2f TB/s (r+w)\n", NAME, G, ms, 2.0 * NB / ms / 1e9); }
    for (int g : {1024, 2048, 4096, 8192}) {
        RUN("copy U1 plain", (k_copy<1, 0>), g);
        RUN("copy U4 plain", (k_copy<4, 0>), g);
        RUN("copy U8 plain", (k_copy<8, 0>), g);
        RUN("copy U4 nt-store", (k_copy<4, 1>), g);
        RUN("copy U4 nt-load+store", (k_copy<4, 2>), g);
        RUN("copy tiles U4 plain", (k_copy_tiles<4, 0>), g);
        RUN("copy tiles U8 plain", (k_copy_tiles<8, 0>), g);
        RUN("copy tiles U8 nt-store", (k_copy_tiles<8, 1>), g);
    }
    for (int g : {2048, 4096}) {
        float ms = timeit([&] { k_read2_write1<<<g, 256>>>(in, (double*)out, n2); });
        printf("read 16 B + write 8 B per lane      grid %5d: %.3f ms  %.2f TB/s (r+w)\n", g, ms, 1.5 * NB / ms / 1e9);
    }
    // run scatter: 4 GiB of items, 512 blocks x 1024 threads
    const int64_t bytes = 4LL << 30;
    for (int W : {1, 2}) {
        const int64_t n_items = bytes / (8 * W);
        for (int P : {128, 256}) {
            for (int L : {8, 16, 32, 64, 128, 1024}) {
                const int G = 512;
                const int64_t per_block = n_items / G;
                const int64_t cap = ((per_block / L / P + 2) * L + 15) & ~15LL;
                if ((int64_t)G * P * cap * W * 8 > NB) continue;
                float ms = W == 1 ? timeit([&] { k_runs<1><<<G, 1024>>>((const double*)in, (double*)out, n_items, L, P, cap); })
                                  : timeit([&] { k_runs<2><<<G, 1024>>>((const double*)in, (double*)out, n_items, L, P, cap); });
                printf("runs: item %2d B, %3d regions/block, run %4d items (%5d B): %.3f ms  %.2f TB/s (r+w)\n", 8 * W, P, L, L * 8 * W, ms,
                       2.0 * bytes / ms / 1e9);
            }
        }
    }
    return 0;
}
