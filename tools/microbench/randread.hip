// microbenchmark: random R-byte record reads (R = 16..128) from a table of T bytes, one record per lane per step
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
template <int R>
__global__ __launch_bounds__(256) void rr(const uint4 *__restrict__ tab, uint64_t nrec, int steps, uint32_t *out) {
    uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int s = 0; s < steps; s++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const uint64_t r0 = (x >> 20) % nrec;
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const uint64_t r1 = (x >> 20) % nrec;
        const uint4 *p0 = tab + r0 * (R / 16), *p1 = tab + r1 * (R / 16);
        uint4 v0[R / 16], v1[R / 16];
#pragma unroll
        for (int i = 0; i < R / 16; i++) { v0[i] = p0[i]; v1[i] = p1[i]; }
#pragma unroll
        for (int i = 0; i < R / 16; i++) acc += v0[i].x ^ v0[i].w ^ v1[i].y ^ v1[i].z;
        x ^= acc & 1;      // dependency: next addresses wait for this step's data (like the FM-index walk)
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int R> void run(const uint4 *tab, size_t tbytes, uint32_t *out, int blocks, int steps) {
    const uint64_t nrec = tbytes / R;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    rr<R><<<blocks, 256>>>(tab, nrec, 8, out);
    hipEventRecord(a);
    rr<R><<<blocks, 256>>>(tab, nrec, steps, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double recs = 2.0 * blocks * 256 * steps;
    printf("table %6zu MB  record %3d B  blocks %5d: %.1f G rec/s  %.2f TB/s\n", tbytes >> 20, R, blocks, recs / ms / 1e6, recs * R / ms / 1e9);
}
int main(int argc, char **argv) {
    const int steps = 400;
    for (size_t mb : {512}) {
        uint4 *tab; uint32_t *out;
        hipMalloc(&tab, mb << 20); hipMemset(tab, 1, mb << 20); hipMalloc(&out, 4 * 256 * 8192);
        for (int blocks : {64, 128, 256, 512, 768, 1024, 1536, 2048}) {
            run<64>(tab, mb << 20, out, blocks, steps);
        }
        hipFree(tab); hipFree(out);
    }
    return 0;
}
