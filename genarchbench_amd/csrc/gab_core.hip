// libgab_hip.so -- library-level entry points (version, error reporting, device probing).
#define GAB_NO_COPY_MACROS      // (this unit defines gab_memcpy / gab_memcpy_async on top of the runtime's calls)
#include "gab_internal.h"
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include <unordered_set>

static thread_local char g_err[512] = "";

void gab_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *gab_version(void) { return "gab-hip 0.1 (gfx950)"; }

// $GAB_ABORT_TRACE=1 (diagnosis): a SIGABRT anywhere in the process -- the HIP runtime's own abort(), a C++ exception nobody
// caught -- prints the native call stack to stderr before the default action takes over; $GAB_ABORT_TRACE=<path> appends it to
// that file instead (a test runner that captures file descriptor 2 swallows the former).  Installed when the library is loaded.
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/stat.h>
#include <unistd.h>
namespace {
char gab_abort_trace_path[512];
void gab_abort_trace(int sig) {
    void *frames[64];
    const int n = backtrace(frames, 64);
    static const char msg[] = "[gab] SIGABRT -- native stack:\n";
    int fd = 2;
    if (gab_abort_trace_path[0]) { const int f = open(gab_abort_trace_path, O_WRONLY | O_CREAT | O_APPEND, 0644); if (f >= 0) fd = f; }
    (void)!write(fd, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, fd);
    // what the aborting code printed: under a test runner file descriptor 2 is a temporary FILE, lost with the process -- its tail
    // goes into the trace (the HSA runtime names the faulting address and the reason there)
    struct stat st;
    if (fd != 2 && fstat(2, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        static char tail[2048];
        const off_t from = st.st_size > (off_t)sizeof tail ? st.st_size - (off_t)sizeof tail : 0;
        const ssize_t got = pread(2, tail, sizeof tail, from);
        static const char hdr[] = "[gab] the end of what file descriptor 2 holds:\n";
        if (got > 0) { (void)!write(fd, hdr, sizeof hdr - 1); (void)!write(fd, tail, (size_t)got); (void)!write(fd, "\n", 1); }
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
struct GabAbortTraceInit {
    GabAbortTraceInit() {
        const char *e = getenv("GAB_ABORT_TRACE");
        if (!e || !*e || !strcmp(e, "0")) return;
        if (e[0] == '/' || e[0] == '.') { strncpy(gab_abort_trace_path, e, sizeof gab_abort_trace_path - 1); }
        void *warm[2]; (void)backtrace(warm, 2);             // (the first call loads libgcc: not inside a signal handler)
        signal(SIGABRT, gab_abort_trace);
    }
} gab_abort_trace_init;
}  // namespace
extern "C" const char *gab_last_error(void) { return g_err; }

extern "C" int gab_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// PCI bus id of a device ("0000:c1:00.0"): the drivers look up the NUMA node of the card with it (host placement,
// benchmarks/common/gab_driver.h)
extern "C" int gab_device_pci_bus_id(int device, char *buf, int len) {
    if (!buf || len < 16) { gab_set_error("gab_device_pci_bus_id: buffer of at least 16 bytes needed"); return GAB_EINVAL; }
    buf[0] = 0;
    const int n = gab_device_count();
    if (n <= 0) { gab_set_error("no HIP device visible"); return GAB_ENODEV; }
    if (device < 0 || device >= n) { gab_set_error("device %d out of range (0..%d)", device, n - 1); return GAB_EINVAL; }
    GAB_HIP(hipDeviceGetPCIBusId(buf, len, device));
    return GAB_OK;
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
    if (const char *e = getenv("GAB_HIP_SCHEDULE")) {          // experiment knob: spin | yield | blocking (before the context exists)
        static std::once_flag once;                            // the drivers create their handles from several threads
        std::call_once(once, [e] {
            (void)hipSetDeviceFlags(!strcmp(e, "spin") ? hipDeviceScheduleSpin : !strcmp(e, "yield") ? hipDeviceScheduleYield : hipDeviceScheduleBlockingSync);
        });
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        gab_set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return GAB_ENODEV;
    }
    return GAB_OK;
}

// ---- the experiment knobs (gab_internal.h: gab_tuning) -- the one place of the library that reads them -------------------------
bool gab_tuning_live() { const char *e = getenv("GAB_TUNING_LIVE"); return e && *e && atoi(e) != 0; }
void gab_tuning_load(gab_tuning *t) {
    *t = gab_tuning();
    auto on = [](const char *name) { return getenv(name) != nullptr; };
    auto num = [](const char *name, long long unset) { const char *e = getenv(name); return e ? atoll(e) : unset; };
    t->bsw_trace = on("GAB_BSW_TRACE"); t->bsw_full_scan = on("GAB_BSW_FULL_SCAN");
    t->bpm_score64 = on("GAB_BPM_SCORE64"); t->bitpal_no_bv = on("GAB_BITPAL_NO_BV"); t->bpm_slices = (int)num("GAB_BPM_SLICES", 0);
    if (const char *e = getenv("GAB_WFA_TUNE")) {
        t->wfa_tuned = true;
        const int k = sscanf(e, "%d,%d,%d,%d,%d,%d,%d", &t->wfa_tune[0], &t->wfa_tune[1], &t->wfa_tune[2], &t->wfa_tune[3], &t->wfa_tune[4], &t->wfa_tune[5], &t->wfa_tune[6]);
        t->wfa_tune_fields = k > 0 ? k : 0;
    }
    t->wfa_no_static = on("GAB_WFA_NO_STATIC"); t->wfa_trace = on("GAB_WFA_TRACE");
    t->wfa_pool2 = (int)num("GAB_WFA_POOL2", -1); t->wfa_slots = (int)num("GAB_WFA_SLOTS", -1);
    if (const char *e = getenv("GAB_CHAIN_HELPERS")) { t->chain_helpers_set = true; t->chain_helpers = atoi(e); }
    { const char *e = getenv("GAB_CHAIN_KERNEL"); t->chain_walk = e && !strcmp(e, "walk"); }
    t->chain_trace = on("GAB_CHAIN_TRACE"); t->chain_feed_giveup = on("GAB_CHAIN_FEED_GIVEUP"); t->chain_fed_serial = on("GAB_CHAIN_FED_SERIAL");
    t->chain_no_overlap = on("GAB_CHAIN_NO_OVERLAP"); t->chain_no_feed = on("GAB_CHAIN_NO_FEED");
    if (const char *e = getenv("GAB_CHAIN_TAB")) t->chain_tab = atoi(e) != 0;
    t->chain_tab_min = num("GAB_CHAIN_TAB_MIN", -1); t->chain_fast_min = num("GAB_CHAIN_FAST_MIN", -1); t->chain_fast_calls = num("GAB_CHAIN_FAST_CALLS", -1);
    t->chain_feed_min = num("GAB_CHAIN_FEED_MIN", -1); t->chain_tab_mb = num("GAB_CHAIN_TAB_MB", -1);
    if (const char *e = getenv("GAB_CHAIN_GATHER_MASK")) { strncpy(t->chain_gather_mask, e, sizeof t->chain_gather_mask - 1); t->chain_gather_mask[sizeof t->chain_gather_mask - 1] = 0; }
    t->chain_gather_blocks = (int)num("GAB_CHAIN_GATHER_BLOCKS", 0);
    t->fmi_lds_entries = (int)num("GAB_FMI_LDS_ENTRIES", 0); t->fmi_waves = (int)num("GAB_FMI_WAVES", 0); t->fmi_wide = (int)num("GAB_FMI_WIDE", 1);
    t->fmi_wide_cap = (int)num("GAB_FMI_WIDE_CAP", 0); t->fmi_kmer_depth = (int)num("GAB_FMI_KMER_DEPTH", -1);
    t->fmi_wide_lists = num("GAB_FMI_WIDE_LISTS", 0) != 0; t->fmi_debug = on("GAB_FMI_DEBUG");
    t->fmi_batch = num("GAB_FMI_BATCH", 0); t->fmi_scratch_mb = num("GAB_FMI_SCRATCH_MB", 0);
}

// ---- plain device-memory helpers for C callers that keep data on the GPU between two entry points ----------------
extern "C" int gab_device_alloc(int device, size_t bytes, void **out) {
    if (!out) { gab_set_error("gab_device_alloc: NULL argument"); return GAB_EINVAL; }
    *out = nullptr;
    int rc = gab_check_device(device);
    if (rc) return rc;
    gab_device_guard g(device);
    GAB_HIP(hipMalloc(out, bytes ? bytes : 1));
    return GAB_OK;
}
extern "C" void gab_device_free(int device, void *p) {
    if (!p) return;
    gab_device_guard g(device);
    (void)hipFree(p);
}
extern "C" int gab_device_copy_to_host(int device, void *dst, const void *d_src, size_t bytes) {
    if (bytes == 0) return GAB_OK;
    if (!dst || !d_src) { gab_set_error("gab_device_copy_to_host: NULL argument"); return GAB_EINVAL; }
    gab_device_guard g(device);
    // (asynchronous copy + wait: the synchronous call moved the 40 MB of bsw-large's scores into page-locked memory in ~9 ms,
    // a tenth of the link's rate -- the drivers' GPU-parse paths end their region of interest with this copy)
    GAB_HIP(gab_memcpy_async(dst, d_src, bytes, hipMemcpyDeviceToHost, nullptr));
    GAB_HIP(hipStreamSynchronize(nullptr));
    return GAB_OK;
}

// ---- pinned host memory ------------------------------------------------------------------------------------------
// The host-pointer entry points (gab_*_run) move their inputs with asynchronous copies on several streams; from pageable
// memory the runtime stages every copy through its own bounce buffer at a fraction of the link rate.  A caller that
// allocates its slabs here (where the reference drivers call _mm_malloc / malloc before the region of interest) gets
// direct DMA.  Any device may use the memory (hipHostMallocPortable).
// What gab_host_alloc handed out: gab_host_free must not infer the allocator from the pointer's attributes -- memory from
// malloc that is still registered (gab_host_register) also reports as page-locked host memory, and hipHostFree on it
// fails, leaving it neither unregistered nor freed.
static std::mutex g_host_mu;
static std::unordered_set<void *> &gab_host_blocks() { static std::unordered_set<void *> s; return s; }
extern "C" int gab_host_alloc(size_t bytes, void **out) {
    if (!out) { gab_set_error("gab_host_alloc: NULL argument"); return GAB_EINVAL; }
    *out = nullptr;
    if (gab_device_count() <= 0) { gab_set_error("no HIP device visible"); return GAB_ENODEV; }
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) { *out = nullptr; gab_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return GAB_ENOMEM; }
    std::lock_guard<std::mutex> g(g_host_mu);
    gab_host_blocks().insert(*out);
    return GAB_OK;
}
// p: from gab_host_alloc, or from malloc (the drivers' slab helper falls back to malloc when page-locked memory runs out);
// malloc'ed memory that the caller registered is unregistered first
extern "C" void gab_host_free(void *p) {
    if (!p) return;
    bool ours;
    {
        std::lock_guard<std::mutex> g(g_host_mu);
        ours = gab_host_blocks().erase(p) != 0;
    }
    if (ours) { (void)hipHostFree(p); return; }
    if (gab_is_pinned(p) && hipHostUnregister(p) != hipSuccess) (void)hipGetLastError();
    free(p);
}
// page-lock memory the caller already owns (malloc, realloc, std::vector storage ...), in place
extern "C" int gab_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return GAB_OK;
    if (gab_device_count() <= 0) { gab_set_error("no HIP device visible"); return GAB_ENODEV; }
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); gab_set_error("hipHostRegister(%zu) failed: %s", bytes, hipGetErrorString(e)); return GAB_ENOMEM; }
    return GAB_OK;
}
extern "C" void gab_host_unregister(void *p) {
    if (p && hipHostUnregister(p) != hipSuccess) (void)hipGetLastError();
}
bool gab_is_pinned(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// ---- one host-to-device copy batch at a time per GPU ---------------------------------------------------------------
// Several handles of one GPU (the drivers' queue workers) run their host-pointer entry points concurrently so that the
// copy-in of one chunk overlaps the kernels of another.  Started together they fall into LOCKSTEP instead -- all copy at
// once (each at a fraction of the link rate), then all compute at once -- and nothing overlaps (bsw-large through the C
// driver, two workers: 96 ms; the same calls staggered: 65 ms).  The entry points therefore take this per-device lock around
// "enqueue the H2D copies of my chunk and wait for them": copies go one chunk at a time at the full link rate while the
// other workers' kernels run.
static std::mutex g_h2d_mutex[64];
std::mutex &gab_h2d_mutex(int device) { return g_h2d_mutex[device >= 0 && device < 64 ? device : 0]; }

// The first host-to-device and the first device-to-host copy on a stream cost ~9 and ~8 ms (the copy queues behind a
// stream are created on first use; measured in the bsw driver: 13 ms instead of 4.3 for the first 240 MB in, 8 ms instead
// of 0.1 for the first 4 MB out).  The *_reserve entry points pay that before the region of interest: 4 MB each way
// between page-locked memory and the handle's staging buffer (at least 4 MB by then).
int gab_warm_copy_engines(hipStream_t s, void *dev, size_t dev_bytes) {
    // (r03: a 4 MB copy does not warm everything -- the first device-to-host copy ABOVE a few MB on a process's streams was
    // measured at 10.8 ms for 5.3 MB in the wfa driver's ROI, 0.1 ms for the next one: up to 32 MB each way here)
    const size_t bytes = std::min<size_t>(dev_bytes, (size_t)32 << 20);
    if (bytes == 0) return GAB_OK;
    void *pinned = nullptr;
    if (hipHostMalloc(&pinned, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return GAB_OK; }   // best effort
    memset(pinned, 0, bytes);
    hipError_t e = hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(pinned, dev, bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipHostFree(pinned);
    if (e != hipSuccess) { gab_set_error("gab_warm_copy_engines: %s", hipGetErrorString(e)); return GAB_EDEVICE; }
    return GAB_OK;
}

// ---- pageable host memory through page-locked buffers of the library's own ($GAB_STAGE_PAGEABLE=1; gab_internal.h says why) -----
namespace {
constexpr size_t kBounceBytes = (size_t)8 << 20, kBounceMin = (size_t)1 << 20;
struct GabBounce {
    std::mutex mu;
    void *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};       // the last transfer that used buf[k]
    bool used[2] = {false, false};
    bool ready() {
        for (int k = 0; k < 2; k++) {
            if (!buf[k] && hipHostMalloc(&buf[k], kBounceBytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); buf[k] = nullptr; return false; }
            if (!ev[k] && hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ev[k] = nullptr; return false; }
        }
        return true;
    }
};
GabBounce g_bounce[64];
bool gab_stage_pageable() {
    static const bool on = [] { const char *e = getenv("GAB_STAGE_PAGEABLE"); return e && *e && strcmp(e, "0") != 0; }();
    return on;
}
}  // namespace

hipError_t gab_memcpy_async(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
    if (!gab_stage_pageable() || bytes <= kBounceMin || (kind != hipMemcpyHostToDevice && kind != hipMemcpyDeviceToHost))
        return hipMemcpyAsync(dst, src, bytes, kind, s);
    const bool h2d = kind == hipMemcpyHostToDevice;
    if (gab_is_pinned(h2d ? src : dst)) return hipMemcpyAsync(dst, src, bytes, kind, s);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return hipMemcpyAsync(dst, src, bytes, kind, s); }
    GabBounce &B = g_bounce[dev];
    std::lock_guard<std::mutex> gate(B.mu);
    if (!B.ready()) return hipMemcpyAsync(dst, src, bytes, kind, s);       // no page-locked memory left: the runtime's own way
    hipError_t e = hipSuccess;
    int k = 0, pk = -1;                 // pk: the chunk of a device-to-host copy still to be moved out of its buffer
    size_t poff = 0, pn = 0;
    for (size_t off = 0; off < bytes; k ^= 1) {
        const size_t n = std::min(kBounceBytes, bytes - off);
        if (B.used[k] && (e = hipEventSynchronize(B.ev[k])) != hipSuccess) return e;
        if (h2d) {
            memcpy(B.buf[k], (const char *)src + off, n);
            if ((e = hipMemcpyAsync((char *)dst + off, B.buf[k], n, kind, s)) != hipSuccess) return e;
        } else {
            if ((e = hipMemcpyAsync(B.buf[k], (const char *)src + off, n, kind, s)) != hipSuccess) return e;
        }
        if ((e = hipEventRecord(B.ev[k], s)) != hipSuccess) return e;
        B.used[k] = true;
        if (!h2d) {
            if (pk >= 0) {
                if ((e = hipEventSynchronize(B.ev[pk])) != hipSuccess) return e;
                memcpy((char *)dst + poff, B.buf[pk], pn);
            }
            pk = k; poff = off; pn = n;
        }
        off += n;
    }
    if (pk >= 0) {
        if ((e = hipEventSynchronize(B.ev[pk])) != hipSuccess) return e;
        memcpy((char *)dst + poff, B.buf[pk], pn);
    }
    return hipSuccess;
}

hipError_t gab_memcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    if (!gab_stage_pageable() || bytes <= kBounceMin || (kind != hipMemcpyHostToDevice && kind != hipMemcpyDeviceToHost) ||
        gab_is_pinned(kind == hipMemcpyHostToDevice ? src : dst))
        return hipMemcpy(dst, src, bytes, kind);
    const hipError_t e = gab_memcpy_async(dst, src, bytes, kind, nullptr);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}
