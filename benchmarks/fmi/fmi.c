/* fmi -- drop-in driver of the fmi benchmark on MI355X.
 *
 *     fmi <ref_prefix> <reads.fastq[.gz]> <batch_size> <minSeedLen> <n_threads>       [$GAB_GPUS = gpus]
 *
 * Positional arguments, the six stdout header lines and the "rid:" / "[m,n+1]" data lines are those of
 * /root/reference/benchmarks/fmi/fmi.cpp:74-475 (the harness drops the six header lines and diffs the rest,
 * fmi/scripts/regression_small.sh:91,97-98).  <ref_prefix>.bwt.2bit.64 is BWA-MEM2's own index file; it is
 * loaded and replicated on every GPU before the ROI, as load_index precedes begin_computing in the reference.
 * The per-batch ROI body (getSMEMsAllPos -> re-seed -> bwtSeedStrategy -> sortSMEMs, fmi.cpp:288-348) becomes
 * gab_fmi_seed_into on chunks of reads ($GAB_CHUNK; default: half a worker's share, 2^20 .. 2^22 reads) pulled by $GAB_WORKERS_PER_GPU host threads per GPU
 * (default 3; the workers of a GPU share one copy of the index); batch_size and n_threads only shaped the CPU
 * scheduling and are accepted and ignored.
 * Every worker collects its SMEMs in an array of its own sized like the reference's per-thread matchArray (20 records per
 * read of its share, fmi.cpp:243,254-255) and, like it, takes more room when a chunk does not fit (:277-286) -- here the
 * array is page-locked, so it is made with the handle before the ROI (the reference's malloc inside the ROI costs nothing
 * until the pages are written; page-locking populates them) and the records arrive by DMA at the link rate.
 */
#define GAB_ENERGY_STREAM stderr      /* where the reference prints "Energy consumption:" in this driver */
#include "../common/gab_driver.h"
#include <zlib.h>

#define CHUNK_READS (1 << 20)   /* default chunk of the work queue; $GAB_CHUNK overrides it */
#define MAX_GPUS 64
typedef struct {
    const char *prefix; const uint8_t *enc; int32_t stride; const int32_t *len; int64_t n, chunk; int32_t msl;
    gab_smem **out; int64_t *nout;      /* per chunk */
    char *own;                          /* per chunk: out[chunk] is a block of its own (gab_fmi_seed) rather than part of an arena */
    gab_fmi *owner[MAX_GPUS];           /* the handle that owns the index of each GPU; further workers clone it */
    int64_t arena_records;              /* SMEM records of a worker's array */
} fmi_ctx;
typedef struct { gab_fmi *h; gab_smem *arena; int64_t cap, used; } fmi_worker;
/* the index is loaded ONCE per GPU and shared by that GPU's workers, as the reference's threads share one FMI_search */
static void *gpu_init(int worker, int gpu, void *vc) {
    (void)worker;
    fmi_ctx *c = (fmi_ctx *)vc;
    fmi_worker *w = (fmi_worker *)calloc(1, sizeof(fmi_worker));
    if (gpu < MAX_GPUS && c->owner[gpu]) GAB_DIE_IF(gab_fmi_clone(c->owner[gpu], &w->h), "gab_fmi_clone");
    else {
        GAB_DIE_IF(gab_fmi_load(gpu, c->prefix, &w->h), "gab_fmi_load");
        if (gpu < MAX_GPUS) c->owner[gpu] = w->h;
    }
    GAB_DIE_IF(gab_fmi_reserve(w->h, c->chunk < c->n ? c->chunk : c->n, c->stride), "gab_fmi_reserve");    /* buffers before the ROI */
    w->cap = c->arena_records;
    void *a = NULL;
    if (gab_env_i64("GAB_NO_PIN", 0) || gab_host_alloc((size_t)w->cap * sizeof(gab_smem), &a) != 0) a = malloc((size_t)w->cap * sizeof(gab_smem));
    if (!a) { fprintf(stderr, "ERROR: out of memory\n"); exit(EXIT_FAILURE); }
    w->arena = (gab_smem *)a;
    return w;
}
/* clones go first (workers are closed in index order and the owners are the first ngpus workers): defer the owners */
static void gpu_fini(int worker, int gpu, void *vc, void *st) {
    (void)worker;
    fmi_ctx *c = (fmi_ctx *)vc;
    fmi_worker *w = (fmi_worker *)st;           /* (the arenas hold the results: they are released after printing) */
    if (!(gpu < MAX_GPUS && c->owner[gpu] == w->h)) gab_fmi_destroy(w->h);       /* the owners: after the queue is closed */
}
static void run_chunk(int worker, int gpu, int64_t chunk, void *vctx, void *st) {
    (void)gpu; (void)worker;
    fmi_ctx *c = (fmi_ctx *)vctx;
    fmi_worker *w = (fmi_worker *)st;
    const int64_t b = chunk * c->chunk, e = b + c->chunk < c->n ? b + c->chunk : c->n;
    int rc = gab_fmi_seed_into(w->h, c->enc + b * c->stride, c->stride, c->len + b, e - b, c->msl, w->arena + w->used, w->cap - w->used,
                               &c->nout[chunk]);
    if (rc == 0) { c->out[chunk] = w->arena + w->used; w->used += c->nout[chunk]; }
    else if (rc == GAB_ERANGE) {                 /* the array is full (the reference reallocs, fmi.cpp:277-286): a block for this chunk */
        GAB_DIE_IF(gab_fmi_seed(w->h, c->enc + b * c->stride, c->stride, c->len + b, e - b, c->msl, &c->out[chunk], &c->nout[chunk]), "gab_fmi_seed");
        c->own[chunk] = 1;
    } else GAB_DIE_IF(rc, "gab_fmi_seed_into");
    for (int64_t i = 0; i < c->nout[chunk]; i++) c->out[chunk][i].rid += (uint32_t)b;      /* rid += batch offset, fmi.cpp:340-343 */
}
int main(int argc, char **argv) {
    if (argc != 6) { printf("Need five arguments : ref_file query_set batch_size minSeedLen n_threads\n"); return 1; }
    gzFile fp = gzopen(argv[2], "r");
    if (fp == 0) { fprintf(stderr, "[E::%s] fail to open file `%s'.\n", __func__, argv[2]); exit(EXIT_FAILURE); }
    printf("before reading sequences\n");
    const double tr0 = gab_now();
    /* FASTA / FASTQ records (what kseq yields): sequence = the line(s) after the header up to '+' or the next header */
    size_t cap = 1 << 20, used = 0, rcap = 1 << 16;
    char *seqs = (char *)malloc(cap);
    int64_t *soff = (int64_t *)malloc(8 * rcap); int32_t *len = (int32_t *)malloc(4 * rcap);
    int64_t n = 0;
    {
        char *line = (char *)malloc(1 << 20);
        int state = 0;            /* 0: expect header, 1: in sequence, 2: in quality */
        int64_t qleft = 0;
        while (gzgets(fp, line, 1 << 20)) {
            size_t l = strlen(line);
            while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
            if (state == 2) { qleft -= (int64_t)l; if (qleft <= 0) state = 0; continue; }
            if ((line[0] == '>' || line[0] == '@') && state != 1) {
                if ((size_t)n == rcap) { rcap *= 2; soff = (int64_t *)realloc(soff, 8 * rcap); len = (int32_t *)realloc(len, 4 * rcap); }
                soff[n] = (int64_t)used; len[n] = 0; n++; state = 1; continue;
            }
            if (state == 1 && line[0] == '+') { state = 2; qleft = len[n - 1]; if (qleft == 0) state = 0; continue; }
            if (state == 1 && (line[0] == '>' || line[0] == '@')) {   /* FASTA: next record */
                if ((size_t)n == rcap) { rcap *= 2; soff = (int64_t *)realloc(soff, 8 * rcap); len = (int32_t *)realloc(len, 4 * rcap); }
                soff[n] = (int64_t)used; len[n] = 0; n++; continue;
            }
            if (state == 1) {
                while (used + l + 8 > cap) { cap *= 2; seqs = (char *)realloc(seqs, cap); }
                memcpy(seqs + used, line, l); used += l; len[n - 1] += (int32_t)l;
            }
        }
        free(line);
        gzclose(fp);
    }
    if (n == 0) { printf("ERROR! seqs = NULL\n"); exit(EXIT_FAILURE); }
    int max_rl = len[0], min_rl = len[0];
    for (int64_t i = 1; i < n; i++) { if (len[i] > max_rl) max_rl = len[i]; if (len[i] < min_rl) min_rl = len[i]; }
    if (max_rl <= 0 || max_rl >= GAB_FMI_MAX_READLEN) { fprintf(stderr, "ERROR: read length out of range (max %d)\n", max_rl); exit(EXIT_FAILURE); }
    /* dense code matrix, fmi.cpp:121-151 */
    uint8_t *enc = (uint8_t *)malloc((size_t)n * (size_t)max_rl);
    for (int64_t r = 0; r < n; r++)
        for (int k = 0; k < max_rl; k++) {
            uint8_t code = 4;
            if (k < len[r]) switch (seqs[soff[r] + k]) { case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break; default: code = 4; }
            enc[r * max_rl + k] = code;
        }
    const int ngpus = gab_pick_gpus(0);
    fmi_ctx ctx;
    memset(&ctx, 0, sizeof ctx);
    {   /* A chunk is one batch of the seeding kernels: two chunks per worker, 2^20 .. 2^22 reads each -- the copy-out of one runs
         * under the kernels of the next, and a batch is the more efficient the bigger it is (10 M reads, three workers: ten chunks
         * of 2^20 259 ms, six of 1.67 M 249 ms, three of 3.33 M 266 ms; profiles/r03_fmi_batches.md); $GAB_CHUNK pins the size. */
        const int64_t w = (int64_t)ngpus * gab_workers_per_gpu();
        int64_t share = (n + 2 * w - 1) / (2 * (w > 0 ? w : 1));
        if (share < CHUNK_READS) share = CHUNK_READS;
        if (share > 4 * CHUNK_READS) share = 4 * CHUNK_READS;
        ctx.chunk = gab_env_i64("GAB_CHUNK", share);
    }
    ctx.prefix = argv[1]; ctx.enc = enc; ctx.stride = max_rl; ctx.len = len; ctx.n = n; ctx.msl = atoi(argv[4]);
    const int64_t nchunks = (n + ctx.chunk - 1) / ctx.chunk;
    gab_pin(enc, (size_t)n * (size_t)max_rl); gab_pin(len, 4 * (size_t)n);
    ctx.out = (gab_smem **)calloc((size_t)nchunks, sizeof(gab_smem *)); ctx.nout = (int64_t *)calloc((size_t)nchunks, 8);
    ctx.own = (char *)calloc((size_t)nchunks + 1, 1);
    {   /* perThreadQuota * 20 (fmi.cpp:243,254): a worker's share of the reads, whole chunks, 20 records per read */
        int64_t workers = (int64_t)ngpus * gab_workers_per_gpu();
        if (workers > nchunks) workers = nchunks > 0 ? nchunks : 1;
        const int64_t share = ((nchunks + workers - 1) / workers) * ctx.chunk;
        ctx.arena_records = 20 * (share < n ? share : n) + 1024;
        if (gab_env_i64("GAB_FMI_ARENA", 0)) ctx.arena_records = gab_env_i64("GAB_FMI_ARENA", 0);      /* tests: force the overflow path */
    }
    gab_queue q;
    gab_queue_open(&q, ngpus, nchunks, gpu_init, run_chunk, gpu_fini, &ctx);       /* index load: before the ROI (fmi.cpp:102-105) */
    const double tr1 = gab_now();
    printf("numReads = %ld, max_readlength = %d, min_readlength = %d\n", (long)n, max_rl, min_rl);
    printf("Running %d threads\n", atoi(argv[5]));
    const double t0 = gab_now();                 /* ROI: fmi.cpp:236-362 */
    gab_roi_begin_n(ngpus);
    gab_queue_run(&q, nchunks);
    gab_roi_end();
    const double t1 = gab_now();
    fmi_worker **ws = (fmi_worker **)calloc((size_t)q.nworkers, sizeof(fmi_worker *));
    const int nws = q.nworkers;
    for (int k = 0; k < nws; k++) ws[k] = (fmi_worker *)q.w[k].state;
    gab_queue_close(&q);
    for (int g = 0; g < MAX_GPUS; g++) if (ctx.owner[g]) gab_fmi_destroy(ctx.owner[g]);
    gab_unpin(enc); gab_unpin(len);
    int64_t total = 0;
    for (int64_t c = 0; c < nchunks; c++) total += ctx.nout[c];
    printf("totalSmems = %ld\n", (long)total);
    printf("Reading time: %g s\n", tr1 - tr0);
    printf("Computing time: %g s\n", t1 - t0);
    int64_t prevRid = -1;                          /* fmi.cpp:430-460 */
    for (int64_t c = 0; c < nchunks; c++)
        for (int64_t i = 0; i < ctx.nout[c]; i++) {
            const gab_smem s = ctx.out[c][i];
            if ((int64_t)s.rid != prevRid) for (int64_t j = prevRid + 1; j <= (int64_t)s.rid; j++) printf("%u:\n", (unsigned)j);
            prevRid = (int64_t)s.rid;
            printf("[%u,%u]\n", s.m, s.n + 1);
        }
    for (int64_t c = 0; c < nchunks; c++) if (ctx.own[c]) gab_fmi_free(ctx.out[c]);
    for (int k = 0; k < nws; k++) { gab_host_free(ws[k]->arena); free(ws[k]); }
    free(ws); free(ctx.own);
    free(ctx.out); free(ctx.nout); free(enc); free(seqs); free(soff); free(len);
    return 0;
}
