// microbenchmark: host <-> device copy rates as the host-pointer entry points (gab_*_run) see them: pageable malloc memory
// vs page-locked (hipHostMalloc / hipHostRegister in place), one stream vs two, one direction vs both; and what page-locking
// costs (the drivers do it once, outside the region of interest).
//   hipcc --offload-arch=gfx950 -O3 pcie_copy.hip -o pcie_copy && ./pcie_copy > profiles/rNN_pcie_copy.md
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
// a kernel that copies through the host's address space (page-locked memory is mapped into the device's): 16 B per lane and step
__global__ void shader_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// the same, in scattered pieces: piece p of `piece16` uint4 goes from src + perm[p] * piece16 to dst + perm[p] * piece16 (a gather in sorted order)
__global__ void shader_copy_pieces(const uint4 *__restrict__ src, uint4 *__restrict__ dst, const unsigned *__restrict__ perm, size_t piece16, unsigned npieces) {
    for (unsigned p = blockIdx.x; p < npieces; p += gridDim.x) {
        const size_t o = (size_t)perm[p] * piece16;
        for (size_t i = threadIdx.x; i < piece16; i += blockDim.x) dst[o + i] = src[o + i];
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t MB = 1 << 20, bytes = 1024 * MB;
    char *dev = nullptr, *dev2 = nullptr;
    CK(hipMalloc((void **)&dev, bytes)); CK(hipMalloc((void **)&dev2, bytes));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    char *pageable = (char *)malloc(bytes), *pageable2 = (char *)malloc(bytes);
    memset(pageable, 1, bytes); memset(pageable2, 2, bytes);
    printf("# host <-> device copies of 1 GiB (gfx950 box), GB/s = 1e9 bytes per second\n\n| case | ms | GB/s |\n|---|---|---|\n");
    auto row = [&](const char *name, double sec, double nbytes) { printf("| %s | %.1f | %.1f |\n", name, sec * 1e3, nbytes / sec / 1e9); fflush(stdout); };
    double t;
    for (int rep = 0; rep < 2; rep++) {
        t = now(); CK(hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0));
        row(rep ? "H2D pageable (malloc), second time" : "H2D pageable (malloc), first time", now() - t, bytes);
    }
    t = now(); CK(hipMemcpyAsync(pageable2, dev, bytes, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0));
    row("D2H pageable (malloc)", now() - t, bytes);
    t = now(); CK(hipHostRegister(pageable, bytes, hipHostRegisterPortable)); row("hipHostRegister of 1 GiB (touched pages)", now() - t, bytes);
    t = now(); CK(hipHostRegister(pageable2, bytes, hipHostRegisterPortable)); row("hipHostRegister of a second 1 GiB", now() - t, bytes);
    for (int rep = 0; rep < 2; rep++) {
        t = now(); CK(hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0));
        row("H2D registered", now() - t, bytes);
    }
    t = now(); CK(hipMemcpyAsync(pageable2, dev, bytes, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0));
    row("D2H registered", now() - t, bytes);
    t = now();
    CK(hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(dev2, pageable2, bytes, hipMemcpyHostToDevice, s1));
    CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
    row("H2D registered, two streams at once (2 GiB)", now() - t, 2.0 * bytes);
    t = now();
    CK(hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(pageable2, dev2, bytes, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
    row("H2D + D2H registered at once (2 GiB)", now() - t, 2.0 * bytes);
    // chunked: 16 MiB and 1 MiB pieces on one stream (the per-call overhead of the copy engine)
    for (size_t piece : {16 * MB, 1 * MB, 64 * 1024ul}) {
        const size_t total = piece >= MB ? bytes : 256 * MB;
        t = now();
        for (size_t o = 0; o < total; o += piece) CK(hipMemcpyAsync(dev + o, pageable + o, piece, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        char name[96]; snprintf(name, sizeof name, "H2D registered in %zu KiB pieces (%zu MiB)", piece >> 10, total >> 20);
        row(name, now() - t, (double)total);
    }
    // copies done by a kernel instead of the copy engine (registered memory, device pointer of the host range)
    {
        void *hp = nullptr, *hp2 = nullptr;
        CK(hipHostGetDevicePointer(&hp, pageable, 0)); CK(hipHostGetDevicePointer(&hp2, pageable2, 0));
        for (int grid : {64, 256, 1024, 4096}) {
            char name[96];
            CK(hipDeviceSynchronize()); t = now();
            hipLaunchKernelGGL(shader_copy, dim3(grid), dim3(256), 0, s0, (const uint4 *)hp, (uint4 *)dev, bytes / 16);
            CK(hipStreamSynchronize(s0));
            snprintf(name, sizeof name, "kernel reads host memory, %d x 256 threads", grid); row(name, now() - t, bytes);
            t = now();
            hipLaunchKernelGGL(shader_copy, dim3(grid), dim3(256), 0, s0, (const uint4 *)dev, (uint4 *)hp2, bytes / 16);
            CK(hipStreamSynchronize(s0));
            snprintf(name, sizeof name, "kernel writes host memory, %d x 256 threads", grid); row(name, now() - t, bytes);
        }
        // pieces of 256 KiB in a shuffled order, host -> device
        const size_t piece = 256 * 1024; const unsigned np = (unsigned)(bytes / piece);
        unsigned *perm = (unsigned *)malloc(4 * np), *dperm = nullptr;
        for (unsigned i = 0; i < np; i++) perm[i] = i;
        for (unsigned i = np - 1; i > 0; i--) { unsigned j = (unsigned)((i * 2654435761u) % (i + 1)); unsigned v = perm[i]; perm[i] = perm[j]; perm[j] = v; }
        CK(hipMalloc((void **)&dperm, 4 * np)); CK(hipMemcpy(dperm, perm, 4 * np, hipMemcpyHostToDevice));
        for (int grid : {64, 256, 1024}) {
            char name[96];
            t = now();
            hipLaunchKernelGGL(shader_copy_pieces, dim3(grid), dim3(256), 0, s0, (const uint4 *)hp, (uint4 *)dev, dperm, piece / 16, np);
            CK(hipStreamSynchronize(s0));
            snprintf(name, sizeof name, "kernel reads host memory in shuffled 256 KiB pieces, %d x 256 threads", grid); row(name, now() - t, bytes);
        }
        // both directions at once: kernel reads on s0, kernel writes on s1
        t = now();
        hipLaunchKernelGGL(shader_copy, dim3(256), dim3(256), 0, s0, (const uint4 *)hp, (uint4 *)dev, bytes / 16);
        hipLaunchKernelGGL(shader_copy, dim3(256), dim3(256), 0, s1, (const uint4 *)dev2, (uint4 *)hp2, bytes / 16);
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        row("kernel reads + kernel writes host memory at once (2 GiB)", now() - t, 2.0 * bytes);
        CK(hipFree(dperm)); free(perm);
    }
    t = now(); CK(hipHostUnregister(pageable)); CK(hipHostUnregister(pageable2)); row("hipHostUnregister x 2", now() - t, 2.0 * bytes);
    char *pinned = nullptr;
    t = now(); CK(hipHostMalloc((void **)&pinned, bytes, hipHostMallocPortable)); row("hipHostMalloc of 1 GiB", now() - t, bytes);
    t = now(); memcpy(pinned, pageable, bytes); row("memcpy pageable -> pinned, one thread (first touch)", now() - t, bytes);
    t = now(); memcpy(pinned, pageable, bytes); row("memcpy pageable -> pinned, one thread", now() - t, bytes);
    t = now(); CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); row("H2D hipHostMalloc", now() - t, bytes);
    t = now(); CK(hipMemcpyAsync(pinned, dev, bytes, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0)); row("D2H hipHostMalloc", now() - t, bytes);
    t = now(); CK(hipHostFree(pinned)); row("hipHostFree", now() - t, bytes);
    return 0;
}
