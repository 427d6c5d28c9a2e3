// Internal helpers shared by the HIP translation units of libgab_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stddef.h>
#include <mutex>
#include "../../include/gab.h"

void gab_set_error(const char *fmt, ...);

#define GAB_HIP(call)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            gab_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call,                  \
                          hipGetErrorString(e_));                                        \
            return e_ == hipErrorOutOfMemory ? GAB_ENOMEM : GAB_EDEVICE;                 \
        }                                                                                \
    } while (0)

#define GAB_CHECK(cond, ...)                                                             \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            gab_set_error(__VA_ARGS__);                                                  \
            return GAB_EINVAL;                                                           \
        }                                                                                \
    } while (0)

// Grow-only device buffer owned by a handle.
struct gab_devbuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return GAB_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            gab_set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            p = nullptr;
            return GAB_ENOMEM;
        }
        cap = want;
        return GAB_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Select the handle's device for the duration of a call (restores on scope exit).
struct gab_device_guard {
    int prev = -1; bool changed = false;
    explicit gab_device_guard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) { changed = (hipSetDevice(dev) == hipSuccess); }
    }
    ~gab_device_guard() { if (changed) (void)hipSetDevice(prev); }
};

// The private stream of a handle's host-pointer entry point (gab_*_run).  Non-blocking, so that several handles on one
// GPU -- the drivers run GAB_WORKERS_PER_GPU host threads per GPU, each with its own handle -- overlap their H2D copies,
// kernels and D2H copies instead of serialising on the NULL stream.  Created on first use (device already selected).
struct gab_host_stream {
    hipStream_t s = nullptr;
    int get(hipStream_t *out) {
        if (!s && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            s = nullptr; gab_set_error("hipStreamCreateWithFlags failed: %s", hipGetErrorString(hipGetLastError()));
            return GAB_EDEVICE;
        }
        *out = s;
        return GAB_OK;
    }
    void release() { if (s) (void)hipStreamDestroy(s); s = nullptr; }
};

std::mutex &gab_h2d_mutex(int device);      // see gab_core.hip: one H2D copy batch at a time per GPU
int gab_warm_copy_engines(hipStream_t s, void *dev, size_t dev_bytes);      // gab_core.hip: first-use cost of a stream's copy queues, paid early
int gab_check_device(int device);
bool gab_is_pinned(const void *p);      // hipHostMalloc'ed / registered host memory (direct DMA) or pageable

static inline int64_t gab_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- experiment knobs and test hooks -------------------------------------------------------------------------------
// Every GAB_* environment switch the entry points honour lives in this ONE struct; gab_tuning_load (gab_core.hip) is the only
// place that reads them, ONCE per handle, when the handle is created.  A shipping call reads no environment variable but
// $GAB_TUNING_LIVE: set (tests/conftest.py does), the handle's copy is read again at every call, so a test can flip a switch
// between two calls of one handle; a driver never does.  INTEGRATION.md lists what each switch is for.
struct gab_tuning {
    // bsw
    bool bsw_trace = false, bsw_full_scan = false;
    // bpm / bitpal
    bool bpm_score64 = false, bitpal_no_bv = false;
    int bpm_slices = 0;                              // 0 = by batch size
    // wfa
    bool wfa_tuned = false, wfa_no_static = false, wfa_trace = false;
    int wfa_tune[7] = {0, 0, 0, 0, 0, 0, 0};         // GAB_WFA_TUNE, as many fields as it gave
    int wfa_tune_fields = 0;
    int wfa_pool2 = -1, wfa_slots = -1;              // -1 = unset
    // chain / fast-chain
    int chain_helpers = 0;                           // 3 / 5 / 7, 0 = unset
    bool chain_helpers_set = false, chain_walk = false, chain_trace = false, chain_feed_giveup = false, chain_fed_serial = false;
    bool chain_no_overlap = false, chain_no_feed = false;
    int chain_tab = -1;                              // GAB_CHAIN_TAB: -1 unset, 0 off, 1 on
    long long chain_tab_min = -1, chain_fast_min = -1, chain_fast_calls = -1, chain_feed_min = -1, chain_tab_mb = -1;   // -1 = unset
    char chain_gather_mask[32] = "";                 // "" unset, "none", "N:M"
    int chain_gather_blocks = 0;
    // fmi
    int fmi_lds_entries = 0, fmi_waves = 0, fmi_wide = 1, fmi_wide_cap = 0, fmi_kmer_depth = -1;      // (depth: -1 = unset)
    bool fmi_wide_lists = false, fmi_debug = false;
    long long fmi_batch = 0, fmi_scratch_mb = 0;
};
void gab_tuning_load(gab_tuning *t);
bool gab_tuning_live();                             // $GAB_TUNING_LIVE
inline void gab_tuning_refresh(gab_tuning *t) { if (gab_tuning_live()) gab_tuning_load(t); }
inline gab_tuning gab_tuning_loaded() { gab_tuning t; gab_tuning_load(&t); return t; }     // (the handles' member initialiser)

// 64-bit device atomics need 8-byte-aligned addresses: an unaligned one faults (r03: the cursor of gab_wfa_run_packed sat behind
// an odd number of 4-byte arrays and the process aborted).  Scratch layouts computed at run time pass the offsets of such
// words through GAB_CHECK_ATOMIC64; counter structs state it for their members with GAB_STATIC_ATOMIC64.
#define GAB_CHECK_ATOMIC64(off) GAB_CHECK((((size_t)(off)) & 7) == 0, "internal error: a 64-bit atomic at offset %zu of a scratch layout (not a multiple of 8)", (size_t)(off))
#define GAB_STATIC_ATOMIC64(type, member) static_assert(offsetof(type, member) % 8 == 0 && alignof(type) >= 8, "64-bit atomics need 8-byte alignment: " #type "::" #member)

// Wave-aggregated "slot = (*counter)++" for the lanes with `active` set: one atomic per wave instead of one per lane
// (10 M lanes incrementing one address take milliseconds), and consecutive lanes get consecutive slots, so a list filled
// this way keeps the input order within a wave.  Call it from all lanes that are still in the loop.
__device__ __forceinline__ uint32_t gab_wave_slot(uint32_t *counter, bool active) {
    const unsigned long long m = __ballot(active);
    if (!m) return 0;
    const int lane = (int)(threadIdx.x & 63u);
    const int leader = __builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// ---- copies from / to PAGEABLE host memory ---------------------------------------------------------------------------------------
// For copies of more than 1 MiB the HIP runtime pins pageable memory IN PLACE and keeps the pins cached; a caller that frees such a
// buffer and gets the same address again can meet a pin whose pages are gone -- a GPU memory fault (lesson 16).
// $GAB_STAGE_PAGEABLE=1 (read once per process): every copy of this library of more than 1 MiB from or to host memory that is
// NOT page-locked goes through two page-locked 8 MiB buffers of the library's own instead (per GPU, shared by its handles), so
// that nothing of the caller's is ever pinned.  Such a copy has consumed (host to device) or filled (device to host) the caller's
// buffer when it returns.  Page-locked memory (gab_host_alloc, gab_host_register) and small copies take the runtime's call as is.
// Every hipMemcpy / hipMemcpyAsync of the library's translation units is one of these (the macros below).
hipError_t gab_memcpy_async(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t gab_memcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
#ifndef GAB_NO_COPY_MACROS
#define hipMemcpyAsync gab_memcpy_async
#define hipMemcpy gab_memcpy
#endif
