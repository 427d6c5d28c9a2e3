// libgab_hip.so -- library-level entry points (version, error reporting, device probing).
#include "gab_internal.h"
#include <string.h>

static thread_local char g_err[512] = "";

void gab_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *gab_version(void) { return "gab-hip 0.1 (gfx950)"; }
extern "C" const char *gab_last_error(void) { return g_err; }

extern "C" int gab_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gab_check_device(int device) {
    int n = gab_device_count();
    if (n <= 0) {
        gab_set_error("no HIP device visible (libgab_hip has no CPU fallback)");
        return GAB_ENODEV;
    }
    if (device < 0 || device >= n) {
        gab_set_error("device %d out of range (0..%d)", device, n - 1);
        return GAB_EINVAL;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        gab_set_error("hipGetDeviceProperties(%d) failed", device);
        return GAB_EDEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        gab_set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return GAB_ENODEV;
    }
    return GAB_OK;
}
