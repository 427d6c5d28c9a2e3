/* Shared plumbing of the drop-in drivers under benchmarks/<name>/: wall-clock timers, the reference's
 * region-of-interest (ROI) annotation switches, and the per-GPU work queue.
 *
 * ROI switches.  The reference brackets its ROI with compile-time hooks
 * (-DPERF_ANALYSIS / -DVTUNE_ANALYSIS / -DFAPP_ANALYSIS / -DDYNAMORIO_ANALYSIS / -DPWR / -DRAPL_STOPWATCH,
 * e.g. /root/reference/benchmarks/bsw/src/main_banded.cpp:290-389).  The same -D names are accepted here so
 * existing build lines keep working:
 *   PERF_ANALYSIS            the same perf_ctl.fifo protocol ("enable" / "disable" written to the fifo, pure POSIX);
 *   PWR / RAPL_STOPWATCH     the reference prints "Energy consumption: %0.4lf J" of the node over the ROI (the harness
 *                            greps it, bsw/scripts/regression_small.sh:99); here the line is the energy of the GPUs the
 *                            run used, from ROCm SMI's accumulating energy counter (rsmi_dev_energy_count_get), loaded
 *                            with dlopen so the drivers do not link against it; silently absent if SMI is;
 *   VTUNE / FAPP / DYNAMORIO vendor tracers with no meaning on an MI355X host path: compile to nothing.
 * In every build the ROI is also a roctx range ("gab_roi") when a roctx library can be dlopen'ed, so
 * `rocprofv3 --marker-trace --kernel-trace -- <driver> ...` shows the kernels inside the same bracket the reference's
 * profilers saw.
 *
 * Multi-GPU.  Work items are independent, so N GPUs are driven by host threads that pull chunk indices from one
 * atomic cursor (the reference's `omp for schedule(dynamic)` over batches, lifted one level up); there is no
 * collective and no exchange step.  GAB_WORKERS_PER_GPU=k (default 3) starts k such threads per GPU, each with its own
 * engine handle and streams, so the H2D copy, the kernels and the D2H copy of consecutive chunks overlap;
 * GAB_CHUNK=items overrides the driver's chunk size (tests use it to push small fixtures through the multi-chunk path).
 */
#ifndef GAB_DRIVER_H
#define GAB_DRIVER_H
#ifndef _GNU_SOURCE
#define _GNU_SOURCE             /* cpu_set_t, pthread_setaffinity_np (host placement below) */
#endif
#include <sched.h>
#include <ctype.h>
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include "gab.h"

#ifndef PERF_ANALYSIS
#define PERF_ANALYSIS 0
#endif
#ifndef PWR
#define PWR 0
#endif
#ifndef RAPL_STOPWATCH
#define RAPL_STOPWATCH 0
#endif
#ifndef GAB_ENERGY_STREAM       /* bsw and wfa print the energy line on stdout, chain / fast-chain / bpm / fmi on stderr */
#define GAB_ENERGY_STREAM stdout
#endif

static inline double gab_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* value of an integer environment variable, or `dflt` when unset / not positive */
static inline int64_t gab_env_i64(const char *name, int64_t dflt) {
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    const long long v = atoll(e);
    return v > 0 ? (int64_t)v : dflt;
}

/* size of a regular file opened for reading, or -1 (pipes, /dev/stdin, process substitutions: the whole-file GPU
 * parsers need the size up front, the line-by-line readers do not) */
static inline int64_t gab_regular_file_size(FILE *f) {
    struct stat st;
    if (fstat(fileno(f), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 0) return -1;
    return (int64_t)st.st_size;
}

/* ---- ROI hooks ------------------------------------------------------------------------------- */
#if PERF_ANALYSIS
static int gab_perf_fd = -1;
#endif
typedef int (*gab_roctx_push_fn)(const char *);
typedef int (*gab_roctx_pop_fn)(void);
static gab_roctx_push_fn gab_roctx_push = NULL;
static gab_roctx_pop_fn gab_roctx_pop = NULL;
static int gab_roctx_state = 0;          /* 0: not probed, 1: available, -1: absent */
static inline void gab_roctx_probe(void) {
    if (gab_roctx_state) return;
    gab_roctx_state = -1;
    const char *names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"};
    for (unsigned i = 0; i < sizeof names / sizeof *names; i++) {
        void *h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        gab_roctx_push = (gab_roctx_push_fn)dlsym(h, "roctxRangePushA");
        gab_roctx_pop = (gab_roctx_pop_fn)dlsym(h, "roctxRangePop");
        if (gab_roctx_push && gab_roctx_pop) { gab_roctx_state = 1; return; }
    }
}
#if PWR || RAPL_STOPWATCH
typedef int (*gab_rsmi_init_fn)(uint64_t);
typedef int (*gab_rsmi_energy_fn)(uint32_t, uint64_t *, float *, uint64_t *);
static gab_rsmi_energy_fn gab_rsmi_energy = NULL;
static double gab_energy_j0 = 0.0;
static int gab_energy_gpus = 1;
static inline double gab_energy_now(void) {       /* joules accumulated by the first gab_energy_gpus GPUs, or -1 */
    if (!gab_rsmi_energy) {
        void *h = dlopen("librocm_smi64.so", RTLD_NOW);
        if (!h) h = dlopen("librocm_smi64.so.1", RTLD_NOW);
        if (!h) return -1.0;
        gab_rsmi_init_fn init = (gab_rsmi_init_fn)dlsym(h, "rsmi_init");
        gab_rsmi_energy = (gab_rsmi_energy_fn)dlsym(h, "rsmi_dev_energy_count_get");
        if (!init || !gab_rsmi_energy || init(0) != 0) { gab_rsmi_energy = NULL; return -1.0; }
    }
    double j = 0.0;
    for (int g = 0; g < gab_energy_gpus; g++) {
        uint64_t cnt = 0, ts = 0; float res = 0.f;            /* counter x resolution = micro-joules */
        if (gab_rsmi_energy((uint32_t)g, &cnt, &res, &ts) != 0) return -1.0;
        j += (double)cnt * (double)res * 1e-6;
    }
    return j;
}
#endif
/* `ngpus` only matters for the energy line */
static inline void gab_roi_begin_n(int ngpus) {
    (void)ngpus;
#if PWR || RAPL_STOPWATCH
    gab_energy_gpus = ngpus > 0 ? ngpus : 1;
    gab_energy_j0 = gab_energy_now();
#endif
#if PERF_ANALYSIS
    gab_perf_fd = open("perf_ctl.fifo", O_WRONLY);
    if (gab_perf_fd == -1) fprintf(stderr, "ERROR opening the Perf pipe\n");
    else if (write(gab_perf_fd, "enable", 6) != 6) fprintf(stderr, "ERROR writing to the Perf pipe\n");
#endif
    gab_roctx_probe();
    if (gab_roctx_state == 1) gab_roctx_push("gab_roi");
}
static inline void gab_roi_begin(void) { gab_roi_begin_n(1); }
static inline void gab_roi_end(void) {
    if (gab_roctx_state == 1) gab_roctx_pop();
#if PERF_ANALYSIS
    if (gab_perf_fd != -1) { if (write(gab_perf_fd, "disable", 7) != 7) {} close(gab_perf_fd); gab_perf_fd = -1; }
#endif
#if PWR || RAPL_STOPWATCH
    const double j1 = gab_energy_now();
    if (gab_energy_j0 >= 0.0 && j1 >= 0.0) fprintf(GAB_ENERGY_STREAM, "Energy consumption: %0.4lf J\n", j1 - gab_energy_j0);
#endif
}

/* number of GPUs to use: explicit flag value if > 0, else $GAB_GPUS, else 1; clamped to what is visible.
 * GAB_GPU_OVERSUBSCRIBE=1 (tests on a one-GPU box): no clamp, logical GPU g is device g % visible -- the N-GPU code paths of
 * the drivers (shares, per-GPU parsing, placement) then run with several logical GPUs on one card. */
static inline int gab_pick_gpus(int flag) {
    int want = flag;
    if (want <= 0) { const char *e = getenv("GAB_GPUS"); want = e ? atoi(e) : 1; }
    if (want <= 0) want = 1;
    int have = gab_device_count();
    if (have <= 0) { fprintf(stderr, "ERROR: no MI355X visible (%s); this driver has no CPU path\n", gab_last_error()); exit(EXIT_FAILURE); }
    if (gab_env_i64("GAB_GPU_OVERSUBSCRIBE", 0)) return want < 64 ? want : 64;
    return want < have ? want : have;
}
/* the device behind logical GPU g (g itself unless GAB_GPU_OVERSUBSCRIBE) */
static inline int gab_phys_gpu(int g) {
    const int have = gab_device_count();
    return have > 0 ? g % have : g;
}

/* ---- host placement ---------------------------------------------------------------------------
 * The reference pins its threads (OMP_PROC_BIND=true OMP_PLACES=cores, bsw/scripts/regression_small.sh:52).  Here a GPU's
 * worker threads -- they stage its chunks, and with page-locked slabs the card reads host memory directly -- run on the cores
 * of the NUMA node the card hangs off: PCI bus id (gab_device_pci_bus_id) -> /sys/bus/pci/devices/<id>/numa_node ->
 * /sys/devices/system/node/node<N>/cpulist -> pthread_setaffinity_np, intersected with the mask the process was given
 * (cgroup cpusets); slabs made by such a thread are first-touched on that node.  GAB_NO_BIND=1 leaves the threads alone;
 * GAB_SYSFS_ROOT prefixes the two sysfs paths (tests point it at a made-up tree). */
static inline int gab_parse_cpulist(const char *s, cpu_set_t *set) {      /* "0-15,128-143\n" -> set; returns the number of CPUs */
    CPU_ZERO(set);
    int n = 0;
    while (*s) {
        while (*s == ',' || isspace((unsigned char)*s)) s++;
        if (!isdigit((unsigned char)*s)) break;
        char *e;
        long a = strtol(s, &e, 10), b = a;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        if (b < a) return -1;
        for (long c = a; c <= b; c++) if (c >= 0 && c < CPU_SETSIZE) { if (!CPU_ISSET((int)c, set)) n++; CPU_SET((int)c, set); }
        s = e;
    }
    return n;
}
static inline int gab_read_small_file(const char *path, char *buf, size_t cap) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    const size_t n = fread(buf, 1, cap - 1, f);
    fclose(f);
    buf[n] = 0;
    return (int)n;
}
/* NUMA node of a PCI device ("0000:c1:00.0", any case), or -1 (no such file, or the kernel says -1: one node / unknown) */
static inline int gab_numa_node_of_pci(const char *busid) {
    const char *root = getenv("GAB_SYSFS_ROOT");
    char id[64], path[512], txt[64];
    size_t k = 0;
    for (; busid[k] && k + 1 < sizeof id; k++) id[k] = (char)tolower((unsigned char)busid[k]);
    id[k] = 0;
    snprintf(path, sizeof path, "%s/sys/bus/pci/devices/%s/numa_node", root ? root : "", id);
    if (gab_read_small_file(path, txt, sizeof txt) <= 0) return -1;
    return atoi(txt);
}
static inline int gab_node_cpus(int node, cpu_set_t *set) {              /* CPUs of a node; returns their number or -1 */
    const char *root = getenv("GAB_SYSFS_ROOT");
    char path[512], txt[4096];
    snprintf(path, sizeof path, "%s/sys/devices/system/node/node%d/cpulist", root ? root : "", node);
    if (gab_read_small_file(path, txt, sizeof txt) <= 0) return -1;
    return gab_parse_cpulist(txt, set);
}
/* NUMA node of device `dev`, or -1 */
static inline int gab_gpu_numa_node(int dev) {
    char id[64];
    if (gab_device_pci_bus_id(dev, id, (int)sizeof id) != 0) return -1;
    return gab_numa_node_of_pci(id);
}
/* bind the calling thread to the cores of `dev`'s node; returns the node, or -1 when nothing was changed */
static inline int gab_bind_thread_to_gpu(int dev) {
    if (gab_env_i64("GAB_NO_BIND", 0)) return -1;
    const int node = gab_gpu_numa_node(dev);
    cpu_set_t want, have;
    if (node < 0 || gab_node_cpus(node, &want) <= 0) return -1;
    if (pthread_getaffinity_np(pthread_self(), sizeof have, &have) != 0) return -1;
    CPU_AND(&want, &want, &have);
    if (CPU_COUNT(&want) == 0) return -1;                    /* the node's cores are not ours (cpuset): stay where we are */
    if (pthread_setaffinity_np(pthread_self(), sizeof want, &want) != 0) return -1;
    return node;
}

#define GAB_DIE_IF(rc, what) do { if ((rc) != 0) { fprintf(stderr, "ERROR: %s failed (%d): %s\n", what, (int)(rc), gab_last_error()); exit(EXIT_FAILURE); } } while (0)

/* page-lock a slab the ROI hands to gab_*_run, in place, once it has its final size (outside the ROI, where the
 * reference allocates its slabs, e.g. bsw/src/main_banded.cpp:260-264); GAB_NO_PIN=1 skips it (pageable copies) */
static inline void gab_pin(const void *p, size_t bytes) {
    if (gab_env_i64("GAB_NO_PIN", 0)) return;
    if (gab_host_register((void *)p, bytes) != 0) fprintf(stderr, "note: could not page-lock %zu bytes (%s); copies will be staged\n", bytes, gab_last_error());
}
/* the same for a slab the ROI WRITES (scores, CIGAR text ...): fresh malloc'ed memory has no pages behind it yet, and the first
 * device-to-host copy into such a buffer was measured at ~10 ms for 5 MB (wfa driver, once per process) against 0.1 ms for the
 * next one -- the pages are touched here, before the ROI, as calloc would (that alone changes nothing: see below) */
static inline void gab_pin_out_on(int dev, void *p, size_t bytes) {
    memset(p, 0, bytes);
    gab_pin(p, bytes);
    /* ... and the first device-to-host copy of more than a few MB into a freshly page-locked region costs ~10 ms, once
     * (wfa driver: 10.8 ms for the first 5.3 MB of CIGAR text, 0.1 ms for every later chunk; ROI 16.1 -> 9.2 ms with this):
     * one copy of up to 16 MB from `dev` -- a GPU the run uses (not device 0 whatever was picked) -- before the
     * ROI.  GAB_NO_OUT_WARM=1 to compare. */
    if (gab_env_i64("GAB_NO_OUT_WARM", 0) || gab_env_i64("GAB_NO_PIN", 0)) return;
    void *d = NULL;
    const size_t wb = bytes < ((size_t)16 << 20) ? bytes : ((size_t)16 << 20);
    if (wb && gab_device_alloc(dev, wb, &d) == 0) {
        if (gab_device_copy_to_host(dev, p, d, wb) != 0) fprintf(stderr, "note: warm-up copy from GPU %d failed (%s)\n", dev, gab_last_error());
        gab_device_free(dev, d);
    } else if (wb) fprintf(stderr, "note: no warm-up copy from GPU %d (%s); the first copy-out inside the ROI pays ~10 ms\n", dev, gab_last_error());
}
static inline void gab_pin_out(void *p, size_t bytes) { gab_pin_out_on(gab_phys_gpu(0), p, bytes); }
static inline void gab_unpin(const void *p) { if (!gab_env_i64("GAB_NO_PIN", 0)) gab_host_unregister((void *)p); }

/* GAB_GPU_PARSE=1 / 0 turn the whole-file GPU parsers on / off; unset = the driver's default: ON in all four
 * drivers since r04 -- the read phase leaves the input on the GPU, so the region of interest is kernels + results back (bpm-large
 * 61 -> 7.5 ms, wfa-large 8.6 -> 4.1 ms, bsw-large 60 -> 47 ms, chain-large 37.5 -> 29.5 ms: its DP kernel writes the results through
 * to the page-locked output arrays while it runs), and the read phase itself is the parsers' GB/s instead of a getline / fscanf loop.
 * Pipes, and files a parser declines, always take the line readers. */
static inline int gab_gpu_parse_wanted(int dflt) { const char *e = getenv("GAB_GPU_PARSE"); return (e && *e) ? atoi(e) != 0 : dflt; }

/* ---- GAB_GPU_PARSE with N GPUs: the file cut at record boundaries, one piece per GPU --------------
 * A GPU parses its own piece (gab_*_parse) and keeps what it parsed: a piece's data never leaves its GPU.  The cuts are found on
 * the host.  cut[0] = 0, cut[parts] = n; returns 0, or -1 when the file cannot be cut that way (the caller reads line by line). */
/* records of `lines` lines each, every line alike (bsw: h0 / reference / query): the host counts newlines, once */
static inline int gab_cut_by_lines(const char *buf, size_t n, int parts, int lines, size_t *cut) {
    if (parts == 1) { cut[0] = 0; cut[1] = n; return 0; }
    size_t total = 0;
    for (size_t i = 0; i < n; i++) total += buf[i] == '\n';
    const size_t recs = total / (size_t)lines;
    cut[0] = 0; cut[parts] = n;
    size_t seen = 0, pos = 0;
    for (int g = 1; g < parts; g++) {
        const size_t want = recs * (size_t)g / (size_t)parts * (size_t)lines;      /* lines in front of piece g */
        while (seen < want && pos < n) { const char *q = (const char *)memchr(buf + pos, '\n', n - pos); if (!q) return -1; pos = (size_t)(q - buf) + 1; seen++; }
        cut[g] = pos;
    }
    /* what lies behind the last whole record (the reference ignores it: numPairs = lines / 3) stays in the last piece */
    return 0;
}
/* records that END with a line starting with `tail` (chain: "EOR"), or START with a line starting with `head` (bpm / wfa: ">"):
 * the cut nearest behind n * g / parts */
static inline int gab_cut_at_marker(const char *buf, size_t n, int parts, const char *marker, int after_marker_line, size_t *cut) {
    const size_t ml = strlen(marker);
    cut[0] = 0; cut[parts] = n;
    for (int g = 1; g < parts; g++) {
        size_t pos = n * (size_t)g / (size_t)parts;
        if (pos < cut[g - 1]) pos = cut[g - 1];
        size_t found = n;
        while (pos < n) {                                   /* the next line start at or behind pos whose line starts with the marker */
            const char *q = pos == 0 ? buf - 1 : (const char *)memchr(buf + pos - 1, '\n', n - pos + 1);
            if (!q) break;
            const size_t ls = (size_t)(q - buf) + 1;          /* a line starts here */
            if (ls + ml <= n && !memcmp(buf + ls, marker, ml)) { found = ls; break; }
            pos = ls + 1;
        }
        if (found >= n) { cut[g] = n; continue; }
        if (after_marker_line) {                             /* the piece ends behind the marker's line */
            const char *e = (const char *)memchr(buf + found, '\n', n - found);
            found = e ? (size_t)(e - buf) + 1 : n;
        }
        cut[g] = found;
    }
    return 0;
}
/* one thread per part, each bound to its GPU's NUMA node first; fn(part index, arg) */
typedef struct { int part, gpu; void (*fn)(int, void *); void *arg; } gab_part_thread;
static void *gab_part_main(void *p) {
    gab_part_thread *t = (gab_part_thread *)p;
    (void)gab_bind_thread_to_gpu(t->gpu);
    t->fn(t->part, t->arg);
    return NULL;
}
static inline void gab_run_parts(int parts, void (*fn)(int, void *), void *arg) {
    pthread_t *th = (pthread_t *)calloc((size_t)parts, sizeof(pthread_t));
    gab_part_thread *a = (gab_part_thread *)calloc((size_t)parts, sizeof(gab_part_thread));
    for (int g = 0; g < parts; g++) { a[g].part = g; a[g].gpu = gab_phys_gpu(g); a[g].fn = fn; a[g].arg = arg; pthread_create(&th[g], NULL, gab_part_main, &a[g]); }
    for (int g = 0; g < parts; g++) pthread_join(th[g], NULL);
    free(th); free(a);
}

/* ---- per-GPU work queue ---------------------------------------------------------------------- */
typedef void (*gab_chunk_fn)(int worker, int gpu, int64_t chunk, void *ctx, void *worker_state);
typedef void *(*gab_gpu_init_fn)(int worker, int gpu, void *ctx);
typedef void (*gab_gpu_fini_fn)(int worker, int gpu, void *ctx, void *worker_state);
typedef struct {
    int worker, gpu; int64_t nchunks; int64_t *cursor; pthread_mutex_t *mu;
    gab_gpu_init_fn init; gab_chunk_fn run; gab_gpu_fini_fn fini; void *ctx; void *state;
    int64_t done;       /* chunks this worker ran */
    int node;           /* NUMA node the thread was bound to, or -1 */
    int64_t own_chunk;  /* >= 0: the one chunk this worker runs (gab_queue_run_each) instead of pulling from the cursor */
} gab_worker;
static void *gab_worker_main(void *p) {
    gab_worker *w = (gab_worker *)p;
    w->node = gab_bind_thread_to_gpu(w->gpu);               /* (a fresh thread per gab_queue_run: bound before it touches a chunk) */
    if (w->own_chunk >= 0) {                                /* gab_queue_run_each: worker k's chunk is chunk k (a GPU's own share) */
        if (w->own_chunk < w->nchunks) { w->run(w->worker, w->gpu, w->own_chunk, w->ctx, w->state); w->done++; }
        return NULL;
    }
    for (;;) {
        pthread_mutex_lock(w->mu);
        int64_t c = (*w->cursor)++;
        pthread_mutex_unlock(w->mu);
        if (c >= w->nchunks) break;
        w->run(w->worker, w->gpu, c, w->ctx, w->state);
        w->done++;
    }
    return NULL;
}
/* init/fini run outside the caller's timed region if the caller times only gab_queue_run */
typedef struct { int ngpus, nworkers; gab_worker *w; pthread_mutex_t mu; int64_t cursor; } gab_queue;
static inline int gab_workers_per_gpu(void) { return (int)gab_env_i64("GAB_WORKERS_PER_GPU", 3); }
/* nchunks = the chunks gab_queue_run will be given: no more workers than that are started (every worker's init reserves
 * device buffers for a whole chunk; with one chunk on one GPU two of the default three would hold theirs for nothing) */
static inline void gab_queue_open(gab_queue *q, int ngpus, int64_t nchunks, gab_gpu_init_fn init, gab_chunk_fn run, gab_gpu_fini_fn fini, void *ctx) {
    q->ngpus = ngpus; q->nworkers = ngpus * gab_workers_per_gpu(); q->cursor = 0;
    if (nchunks < 1) nchunks = 1;
    if ((int64_t)q->nworkers > nchunks) q->nworkers = (int)nchunks;
    pthread_mutex_init(&q->mu, NULL);
    q->w = (gab_worker *)calloc((size_t)q->nworkers, sizeof(gab_worker));
    for (int k = 0; k < q->nworkers; k++) {
        gab_worker *w = &q->w[k];
        w->worker = k; w->gpu = gab_phys_gpu(k % ngpus); w->cursor = &q->cursor; w->mu = &q->mu;      /* worker k and k + ngpus share a GPU */
        w->node = -1; w->own_chunk = -1;
        w->init = init; w->run = run; w->fini = fini; w->ctx = ctx;
        w->state = init ? init(k, w->gpu, ctx) : NULL;
    }
}
static inline void gab_queue_run(gab_queue *q, int64_t nchunks) {
    q->cursor = 0;
    pthread_t *th = (pthread_t *)calloc((size_t)q->nworkers, sizeof(pthread_t));
    for (int k = 0; k < q->nworkers; k++) { q->w[k].nchunks = nchunks; q->w[k].done = 0; pthread_create(&th[k], NULL, gab_worker_main, &q->w[k]); }
    for (int k = 0; k < q->nworkers; k++) pthread_join(th[k], NULL);
    free(th);
    if (getenv("GAB_QUEUE_REPORT")) {       /* one line for tests / tuning: how the chunks were spread */
        fprintf(stderr, "gab_queue: %ld chunks over %d workers on %d GPU(s):", (long)nchunks, q->nworkers, q->ngpus);
        for (int k = 0; k < q->nworkers; k++) fprintf(stderr, " %ld", (long)q->w[k].done);
        fprintf(stderr, "\n");
        fprintf(stderr, "gab_queue placement (worker:gpu@node):");
        for (int k = 0; k < q->nworkers; k++) fprintf(stderr, " %d:%d@%d", k, q->w[k].gpu, q->w[k].node);
        fprintf(stderr, "\n");
    }
}
/* chunk k by worker k -- the chunks are per-GPU shares (share g was laid out for GPU g: its slabs sit on that GPU's NUMA node) */
static inline void gab_queue_run_each(gab_queue *q, int64_t nchunks) {
    for (int k = 0; k < q->nworkers; k++) q->w[k].own_chunk = k;
    gab_queue_run(q, nchunks);
    for (int k = 0; k < q->nworkers; k++) q->w[k].own_chunk = -1;
}
static inline void gab_queue_close(gab_queue *q) {
    for (int k = 0; k < q->nworkers; k++) if (q->w[k].fini) q->w[k].fini(k, q->w[k].gpu, q->w[k].ctx, q->w[k].state);
    free(q->w);
    pthread_mutex_destroy(&q->mu);
}
#endif
