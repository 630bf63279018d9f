// Does the 256 MB Infinity Cache (MALL) absorb a write -> read round trip of a recycled scratch buffer?
// k_write streams X bytes, k_read streams them back; compare X below and above the cache size.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mall_bench tools/mall_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k_write(double2* o, int64_t n2, double v) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) o[i] = double2{v, v + i};
}
__global__ void k_read(const double2* __restrict__ in, int64_t n2, double* out) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) { double2 t = in[i]; acc += t.x + t.y; }
    if (acc == 1.2345) out[0] = acc;
}
__global__ void k_copy(const double2* __restrict__ in, double2* o, int64_t n2) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) o[i] = in[i];
}
int main() {
    double* out; hipMalloc(&out, 8);
    double2 *big, *buf;
    const size_t BIG = 8ull << 30;
    hipMalloc(&big, BIG); hipMalloc(&buf, 4ull << 30);
    hipMemset(big, 1, BIG); hipMemset(buf, 0, 4ull << 30);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096}) {
        int64_t n2 = (int64_t)mb * (1 << 20) / 16;
        int iters = (int)(8192 / mb); if (iters < 4) iters = 4;
        for (int mode = 0; mode < 2; mode++) {
            // mode 0: write scratch, read scratch.  mode 1: copy a fresh slice of an 8 GB stream INTO the scratch, read scratch
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                for (int it = 0; it < iters; it++) {
                    if (mode == 0) k_write<<<2048, 256>>>(buf, n2, (double)it);
                    else k_copy<<<2048, 256>>>(big + ((int64_t)it * n2) % (int64_t)(BIG / 16 - n2), buf, n2);
                    k_read<<<2048, 256>>>(buf, n2, out);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            double bytes = (double)iters * (double)n2 * 16 * (mode == 0 ? 2 : 3);
            printf("%s scratch %5zu MB x%4d: %8.3f ms  %6.2f TB/s total, %7.2f us per round trip\n", mode ? "copy+read " : "write+read", mb, iters,
                   ms, bytes / ms / 1e9, ms * 1e3 / iters);
        }
    }
    return 0;
}
