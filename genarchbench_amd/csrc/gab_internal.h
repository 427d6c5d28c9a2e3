// Internal helpers shared by the HIP translation units of libgab_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
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

int gab_check_device(int device);

static inline int64_t gab_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
