/* Shared plumbing of the drop-in drivers under benchmarks/<name>/: wall-clock timers, the reference's
 * region-of-interest (ROI) annotation switches, and the per-GPU work queue.
 *
 * ROI switches.  The reference brackets its ROI with compile-time hooks
 * (-DPERF_ANALYSIS / -DVTUNE_ANALYSIS / -DFAPP_ANALYSIS / -DDYNAMORIO_ANALYSIS / -DPWR / -DRAPL_STOPWATCH,
 * e.g. /root/reference/benchmarks/bsw/src/main_banded.cpp:290-389).  The same -D names are accepted here so
 * existing build lines keep working: PERF_ANALYSIS drives the same perf_ctl.fifo protocol (pure POSIX);
 * the vendor-specific ones (VTune, Fujitsu fapp / Power API, DynamoRIO, RAPL) have no meaning on an
 * MI355X host path and compile to nothing -- use `rocprofv3 --kernel-trace` around the same ROI instead.
 *
 * Multi-GPU.  Work items are independent, so N GPUs are driven by N host threads that pull chunk indices
 * from one atomic cursor (the reference's `omp for schedule(dynamic)` over batches, lifted one level up);
 * there is no collective and no exchange step.
 */
#ifndef GAB_DRIVER_H
#define GAB_DRIVER_H
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <fcntl.h>
#include <unistd.h>
#include "gab.h"

#ifndef PERF_ANALYSIS
#define PERF_ANALYSIS 0
#endif

static inline double gab_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

#if PERF_ANALYSIS
static int gab_perf_fd = -1;
#endif
static inline void gab_roi_begin(void) {
#if PERF_ANALYSIS
    gab_perf_fd = open("perf_ctl.fifo", O_WRONLY);
    if (gab_perf_fd == -1) fprintf(stderr, "ERROR opening the Perf pipe\n");
    else if (write(gab_perf_fd, "enable", 6) != 6) fprintf(stderr, "ERROR writing to the Perf pipe\n");
#endif
}
static inline void gab_roi_end(void) {
#if PERF_ANALYSIS
    if (gab_perf_fd != -1) { if (write(gab_perf_fd, "disable", 7) != 7) {} close(gab_perf_fd); }
#endif
}

/* number of GPUs to use: explicit flag value if > 0, else $GAB_GPUS, else 1; clamped to what is visible */
static inline int gab_pick_gpus(int flag) {
    int want = flag;
    if (want <= 0) { const char *e = getenv("GAB_GPUS"); want = e ? atoi(e) : 1; }
    if (want <= 0) want = 1;
    int have = gab_device_count();
    if (have <= 0) { fprintf(stderr, "ERROR: no MI355X visible (%s); this driver has no CPU path\n", gab_last_error()); exit(EXIT_FAILURE); }
    return want < have ? want : have;
}

#define GAB_DIE_IF(rc, what) do { if ((rc) != 0) { fprintf(stderr, "ERROR: %s failed (%d): %s\n", what, (int)(rc), gab_last_error()); exit(EXIT_FAILURE); } } while (0)

/* ---- per-GPU work queue ---------------------------------------------------------------------- */
typedef void (*gab_chunk_fn)(int gpu, int64_t chunk, void *ctx, void *gpu_state);
typedef void *(*gab_gpu_init_fn)(int gpu, void *ctx);
typedef void (*gab_gpu_fini_fn)(int gpu, void *ctx, void *gpu_state);
typedef struct {
    int gpu; int64_t nchunks; int64_t *cursor; pthread_mutex_t *mu;
    gab_gpu_init_fn init; gab_chunk_fn run; gab_gpu_fini_fn fini; void *ctx; void *state;
} gab_worker;
static void *gab_worker_main(void *p) {
    gab_worker *w = (gab_worker *)p;
    for (;;) {
        pthread_mutex_lock(w->mu);
        int64_t c = (*w->cursor)++;
        pthread_mutex_unlock(w->mu);
        if (c >= w->nchunks) break;
        w->run(w->gpu, c, w->ctx, w->state);
    }
    return NULL;
}
/* init/fini run outside the caller's timed region if the caller times only gab_queue_run */
typedef struct { int ngpus; gab_worker *w; pthread_mutex_t mu; int64_t cursor; } gab_queue;
static inline void gab_queue_open(gab_queue *q, int ngpus, gab_gpu_init_fn init, gab_chunk_fn run, gab_gpu_fini_fn fini, void *ctx) {
    q->ngpus = ngpus; q->cursor = 0;
    pthread_mutex_init(&q->mu, NULL);
    q->w = (gab_worker *)calloc((size_t)ngpus, sizeof(gab_worker));
    for (int g = 0; g < ngpus; g++) {
        q->w[g].gpu = g; q->w[g].cursor = &q->cursor; q->w[g].mu = &q->mu;
        q->w[g].init = init; q->w[g].run = run; q->w[g].fini = fini; q->w[g].ctx = ctx;
        q->w[g].state = init ? init(g, ctx) : NULL;
    }
}
static inline void gab_queue_run(gab_queue *q, int64_t nchunks) {
    q->cursor = 0;
    pthread_t *th = (pthread_t *)calloc((size_t)q->ngpus, sizeof(pthread_t));
    for (int g = 0; g < q->ngpus; g++) { q->w[g].nchunks = nchunks; pthread_create(&th[g], NULL, gab_worker_main, &q->w[g]); }
    for (int g = 0; g < q->ngpus; g++) pthread_join(th[g], NULL);
    free(th);
}
static inline void gab_queue_close(gab_queue *q) {
    for (int g = 0; g < q->ngpus; g++) if (q->w[g].fini) q->w[g].fini(g, q->w[g].ctx, q->w[g].state);
    free(q->w);
    pthread_mutex_destroy(&q->mu);
}
#endif
