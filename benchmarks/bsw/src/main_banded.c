/* main_bsw -- drop-in driver of the bsw benchmark on MI355X.
 *
 * Same command line, input format, result text and timing lines as the reference driver
 * (/root/reference/benchmarks/bsw/src/main_banded.cpp): the harness greps "score=" on stderr and takes
 * field 6 of the "Overall SW cycles" line (bsw/scripts/regression_small.sh:89,92).
 *
 *   main_bsw -pairs <InSeqFile> -t <threads> -b <batch_size> [-match a -mismatch b -ambig c -gapo o -gape e]
 *            [-g <gpus>]
 *
 * Instead of T OpenMP threads each calling getScores16 on batches of B pairs (main_banded.cpp:338-350),
 * the ROI makes gab_bsw_run calls on chunks of pairs ($GAB_CHUNK, default 2^20) pulled from a shared cursor by
 * $GAB_WORKERS_PER_GPU host threads per GPU (default 3, each with its own handle: the H2D copy of one chunk runs under
 * the kernels of another).  -t and -b are accepted and ignored (they tune the CPU path only); -g / $GAB_GPUS selects the
 * number of GPUs.
 *
 * Default (GAB_GPU_PARSE=0 turns it off): the input file is read in one piece, cut at pair boundaries into one piece per GPU, and every GPU parses
 * ITS piece (gab_bsw_parse_pairs) instead of the host reading line by line with fgets / sscanf; the packed
 * buffers stay on the GPU that parsed them and the ROI is one gab_bsw_run_device per GPU.  Files the GPU parser does not accept
 * (a line that hits one of the reference's buffer limits) fall back to the line-by-line path below.
 */
#include "../../common/gab_driver.h"
#include <assert.h>

#define MAX_SEQ_LEN_REF 2048   /* main_banded.cpp:76-79 */
#define MAX_SEQ_LEN_QER 256
#define CHUNK_PAIRS (1 << 20)   /* default chunk of the work queue; $GAB_CHUNK overrides it */

typedef struct {
    gab_bsw_params prm;
    const uint8_t *ref, *qry;
    const int64_t *ref_off, *qry_off;
    const int32_t *len1, *len2, *h0;
    int32_t *score;
    int64_t n, chunk;
    double *busy;   /* per-worker seconds inside gab_bsw_run */
} bsw_ctx;

static void *gpu_init(int worker, int gpu, void *vctx) {
    (void)worker;
    bsw_ctx *c = (bsw_ctx *)vctx;
    gab_bsw *h = NULL;
    GAB_DIE_IF(gab_bsw_create(&c->prm, gpu, &h), "gab_bsw_create");
    /* device buffers for one chunk, like the working buffers the reference's constructor allocates (bandedSWA.cpp:80-96) */
    const int64_t m = c->chunk < c->n ? c->chunk : c->n;
    GAB_DIE_IF(gab_bsw_reserve(h, m, m * (MAX_SEQ_LEN_REF / 8) + 4096, m * (MAX_SEQ_LEN_QER / 2) + 4096), "gab_bsw_reserve");
    return h;
}
static void gpu_fini(int worker, int gpu, void *vctx, void *st) { (void)worker; (void)gpu; (void)vctx; gab_bsw_destroy((gab_bsw *)st); }
static void run_chunk(int worker, int gpu, int64_t chunk, void *vctx, void *st) {
    (void)gpu;
    bsw_ctx *c = (bsw_ctx *)vctx;
    const int64_t b = chunk * c->chunk, e = b + c->chunk < c->n ? b + c->chunk : c->n;
    const double t0 = gab_now();
    /* offsets are absolute into the slabs, so a chunk is just a window of the per-pair arrays */
    GAB_DIE_IF(gab_bsw_run((gab_bsw *)st, c->ref, c->ref_off + b, c->qry, c->qry_off + b, c->len1 + b, c->len2 + b,
                           c->h0 + b, e - b, c->score + b), "gab_bsw_run");
    c->busy[worker] += gab_now() - t0;
}

/* ---- GAB_GPU_PARSE: the file cut at pair boundaries, every GPU parses its piece and runs the DP on it --- */
typedef struct {
    int dev, ok; int64_t first; double busy;
    gab_parser *ps; gab_bsw_packed pk; gab_bsw *h; int32_t *d_score;
} gpp_part;
typedef struct { int ng; const char *whole; size_t cut[65]; gpp_part part[64]; gab_bsw_params prm; int32_t *score; } gpp_ctx;
static void gpp_parse(int g, void *v) {
    gpp_ctx *G = (gpp_ctx *)v;
    gpp_part *p = &G->part[g];
    p->dev = gab_phys_gpu(g);
    if (gab_parser_create(p->dev, &p->ps) != 0) return;
    if (gab_bsw_parse_pairs(p->ps, G->whole + G->cut[g], (int64_t)(G->cut[g + 1] - G->cut[g]), &p->pk, NULL) != 0) return;
    if (gab_bsw_create(&G->prm, p->dev, &p->h) != 0) return;
    if (gab_device_alloc(p->dev, 4 * (size_t)p->pk.n + 4, (void **)&p->d_score) != 0) return;
    if (gab_bsw_reserve(p->h, p->pk.n, 0, 0) != 0) return;       /* work space, first launches: before the region of interest */
    p->ok = 1;
}
static void gpp_run(int g, void *v) {
    gpp_ctx *G = (gpp_ctx *)v;
    gpp_part *p = &G->part[g];
    if (p->pk.n == 0) return;
    const double t0 = gab_now();
    GAB_DIE_IF(gab_bsw_run_device(p->h, p->pk.d_ref, p->pk.ref_bytes, p->pk.d_ref_off, p->pk.d_qry, p->pk.qry_bytes, p->pk.d_qry_off, p->pk.d_len1,
                                  p->pk.d_len2, p->pk.d_h0, p->pk.n, p->d_score, NULL, NULL), "gab_bsw_run_device");
    const double t1 = gab_now();
    GAB_DIE_IF(gab_device_copy_to_host(p->dev, G->score + p->first, p->d_score, 4 * (size_t)p->pk.n), "gab_device_copy_to_host");
    p->busy = gab_now() - t0;
    if (getenv("GAB_BSW_TRACE")) fprintf(stdout, "[gpu %d] launches returned after %.2f ms, scores on the host after %.2f ms\n", g, (t1 - t0) * 1e3, p->busy * 1e3);
}

/* 5x5 matrix exactly as bwa_fill_scmat, main_banded.cpp:94-102 */
static void fill_scmat(int a, int b, int ambig, int8_t mat[25]) {
    int k = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) mat[k++] = (int8_t)(i == j ? a : -b);
        mat[k++] = (int8_t)ambig;
    }
    for (int j = 0; j < 5; ++j) mat[k++] = (int8_t)ambig;
}

int main(int argc, char *argv[]) {
    if (argc < 3) {
        fprintf(stderr, "usage: bsw -pairs <InSeqFile> -t <threads> -b <batch_size>\n");
        exit(EXIT_FAILURE);
    }
    int w_match = 1, w_mismatch = 4, w_open = 6, w_extend = 1, w_ambig = -1, numThreads = 1, batchSize = 0, gpus = 0;
    const char *pairFileName = NULL;
    for (int i = 1; i + 1 < argc; i += 2) {              /* flags are consumed pairwise (main_banded.cpp:115-145) */
        if (!strcmp(argv[i], "-match")) w_match = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "-mismatch")) w_mismatch = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "-ambig")) w_ambig = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "-gapo")) w_open = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "-gape")) w_extend = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "-pairs")) pairFileName = argv[i + 1];
        if (!strcmp(argv[i], "-t")) numThreads = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "-b")) batchSize = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "-g")) gpus = atoi(argv[i + 1]);
    }
    (void)numThreads; (void)batchSize;
    if (!pairFileName) { fprintf(stderr, "ERROR! pairFileName not specified.\n"); exit(EXIT_FAILURE); }
    FILE *pairFile = fopen(pairFileName, "r");
    if (!pairFile) { fprintf(stderr, "Could not open file: %s\n", pairFileName); exit(EXIT_FAILURE); }

    /* a pipe / process substitution has no size: only regular files take the whole-file GPU parser */
    const int64_t fsz = gab_regular_file_size(pairFile);
    if (gab_gpu_parse_wanted(1) && fsz >= 0) {
        const double tR0 = gab_now();
        const int ng = gab_pick_gpus(gpus);
        char *whole = (char *)malloc((size_t)fsz + 1);
        gpp_ctx G;
        memset(&G, 0, sizeof G);
        G.ng = ng; G.whole = whole;
        G.prm.o_del = w_open; G.prm.e_del = w_extend; G.prm.o_ins = w_open; G.prm.e_ins = w_extend;
        G.prm.zdrop = 100; G.prm.end_bonus = 5; G.prm.w = 100;
        fill_scmat(w_match, w_mismatch, w_ambig, G.prm.mat);
        int ok = whole && fread(whole, 1, (size_t)fsz, pairFile) == (size_t)fsz && gab_cut_by_lines(whole, (size_t)fsz, ng, 3, G.cut) == 0;
        if (ok) {
            gab_run_parts(ng, gpp_parse, &G);                    /* every GPU parses its piece and keeps it */
            for (int g = 0; g < ng; g++) ok = ok && G.part[g].ok;
        }
        if (ok) {
            free(whole); fclose(pairFile);
            int64_t n = 0, inb = 0;
            for (int g = 0; g < ng; g++) { G.part[g].first = n; n += G.part[g].pk.n; inb += G.part[g].pk.ref_bytes + G.part[g].pk.qry_bytes; }
            const double readT = gab_now() - tR0;
            printf("Number of input pairs: %ld\n", (long)n);
            printf("Allocating %.3f GB memory for input buffers...\n", (double)(inb + 32 * n) / (1024.0 * 1024 * 1024));
            G.score = (int32_t *)malloc(4 * (size_t)n + 4);
            gab_pin_out_on(gab_phys_gpu(0), G.score, 4 * (size_t)n + 4);
            const double t0g = gab_now();
            gab_roi_begin_n(ng);
            gab_run_parts(ng, gpp_run, &G);                      /* ROI: gab_bsw_run_device on every GPU's own pairs + the scores back */
            gab_roi_end();
            const double roiG = gab_now() - t0g;
            gab_unpin(G.score);
            for (int g = 0; g < ng; g++) printf("%d] workTicks = %ld\n", g, (long)(G.part[g].busy * 1e9));
            printf("Executed HIP gfx950 code on %d GPU(s) (input parsed on the GPU)...\n", ng);
            for (int64_t i = 0; i < n; ++i) fprintf(stderr, "[%ld] score=%d\n", (long)i, G.score[i]);
            printf("Processor freq: %0.2lf MHz\n", 1000.0);
            printf("Read time = %0.2lf s\n", readT);
            printf("Overall SW cycles = %ld, %0.2lf s\n", (long)(roiG * 1e9), roiG);
            printf("Total Pairs processed: %ld\n", (long)n);
            double sum = 0, mx = 0;
            for (int g = 0; g < ng; g++) { sum += G.part[g].busy; if (G.part[g].busy > mx) mx = G.part[g].busy; }
            printf("avgTicks = %lf, maxTicks = %ld, load imbalance = %lf\n", sum * 1e9 / ng, (long)(mx * 1e9), sum > 0 ? mx / (sum / ng) : 1.0);
            if (getenv("GAB_QUEUE_REPORT")) {
                fprintf(stderr, "gab GPU parse: %d piece(s), pairs per GPU:", ng);
                for (int g = 0; g < ng; g++) fprintf(stderr, " %ld", (long)G.part[g].pk.n);
                fprintf(stderr, "\n");
            }
            for (int g = 0; g < ng; g++) { gab_device_free(G.part[g].dev, G.part[g].d_score); gab_bsw_destroy(G.part[g].h); gab_parser_destroy(G.part[g].ps); }
            free(G.score);
            return 0;
        }
        if (getenv("GAB_GPU_PARSE")) fprintf(stderr, "GPU parser declined the file (%s); using the line-by-line parser\n", gab_last_error());      /* (asked for by name: say so; the default falls back silently) */
        for (int g = 0; g < ng; g++) { if (G.part[g].d_score) gab_device_free(G.part[g].dev, G.part[g].d_score); if (G.part[g].h) gab_bsw_destroy(G.part[g].h); if (G.part[g].ps) gab_parser_destroy(G.part[g].ps); }
        free(whole);
        fseek(pairFile, 0L, SEEK_SET);
    }

    /* numPairs = newline count / 3 (main_banded.cpp:237-253); the reference counts with fread + fseek, so it needs a
     * seekable file too */
    size_t numLines = 0, nread;
    {
        const size_t bufSize = 1 << 20;
        char *buffer = (char *)malloc(bufSize);
        while ((nread = fread(buffer, 1, bufSize, pairFile)) > 0)
            for (size_t i = 0; i < nread; i++) numLines += buffer[i] == '\n';
        free(buffer);
        if (fseek(pairFile, 0L, SEEK_SET) != 0) { fprintf(stderr, "ERROR: %s is not seekable\n", pairFileName); exit(EXIT_FAILURE); }
    }
    const int64_t numPairs = (int64_t)(numLines / 3);
    printf("Number of input pairs: %ld\n", (long)numPairs);

    /* read + pack: one byte per base, sequences back to back (the reference keeps 2048 / 256-byte slots) */
    const double tRead0 = gab_now();
    int64_t *ref_off = (int64_t *)malloc(8 * (size_t)(numPairs + 1)), *qry_off = (int64_t *)malloc(8 * (size_t)(numPairs + 1));
    int32_t *len1 = (int32_t *)malloc(4 * (size_t)numPairs + 4), *len2 = (int32_t *)malloc(4 * (size_t)numPairs + 4);
    int32_t *h0 = (int32_t *)malloc(4 * (size_t)numPairs + 4), *score = (int32_t *)malloc(4 * (size_t)numPairs + 4);
    size_t refCap = (size_t)numPairs * 160 + 4096, qryCap = (size_t)numPairs * 96 + 4096, refUsed = 0, qryUsed = 0;
    uint8_t *ref = (uint8_t *)malloc(refCap), *qry = (uint8_t *)malloc(qryCap);
    printf("Allocating %.3f GB memory for input buffers...\n", (double)(refCap + qryCap + 32 * (size_t)numPairs) / (1024.0 * 1024 * 1024));
    char temp[10], *lineR = (char *)malloc(MAX_SEQ_LEN_REF), *lineQ = (char *)malloc(MAX_SEQ_LEN_QER);
    int64_t got = 0;
    while (got < numPairs) {
        int h = 0;
        if (!fgets(temp, 10, pairFile)) break;
        sscanf(temp, "%d", &h);
        if (!fgets(lineR, MAX_SEQ_LEN_REF, pairFile)) { printf("WARNING! fgets returned NULL in %s. Num Pairs : %ld\n", pairFileName, (long)got); break; }
        if (!fgets(lineQ, MAX_SEQ_LEN_QER, pairFile)) { printf("WARNING! Odd number of sequences in %s\n", pairFileName); break; }
        const int l1 = (int)strnlen(lineR, MAX_SEQ_LEN_REF) - 1, l2 = (int)strnlen(lineQ, MAX_SEQ_LEN_QER) - 1;
        if (l1 <= 0 || l2 <= 0) fprintf(stderr, "%ld\n", (long)got);
        assert(l1 > 0); assert(l2 > 0);
        if (refUsed + (size_t)l1 + 8 > refCap) { refCap = refCap * 2 + (size_t)l1; ref = (uint8_t *)realloc(ref, refCap); }
        if (qryUsed + (size_t)l2 + 8 > qryCap) { qryCap = qryCap * 2 + (size_t)l2; qry = (uint8_t *)realloc(qry, qryCap); }
        for (int k = 0; k < l1; k++) ref[refUsed + k] = (uint8_t)(lineR[k] - 48);
        for (int k = 0; k < l2; k++) qry[qryUsed + k] = (uint8_t)(lineQ[k] - 48);
        ref_off[got] = (int64_t)refUsed; qry_off[got] = (int64_t)qryUsed;
        len1[got] = l1; len2[got] = l2; h0[got] = h;
        refUsed += (size_t)l1; qryUsed += (size_t)l2;
        got++;
    }
    fclose(pairFile);
    const int64_t n = got;
    const double readTime = gab_now() - tRead0;

    bsw_ctx ctx;
    memset(&ctx, 0, sizeof ctx);
    ctx.prm.o_del = w_open; ctx.prm.e_del = w_extend; ctx.prm.o_ins = w_open; ctx.prm.e_ins = w_extend;
    ctx.prm.zdrop = 100; ctx.prm.end_bonus = 5; ctx.prm.w = 100;            /* main_banded.cpp:268 */
    fill_scmat(w_match, w_mismatch, w_ambig, ctx.prm.mat);
    ctx.ref = ref; ctx.qry = qry; ctx.ref_off = ref_off; ctx.qry_off = qry_off;
    ctx.len1 = len1; ctx.len2 = len2; ctx.h0 = h0; ctx.score = score; ctx.n = n;
    ctx.chunk = gab_env_i64("GAB_CHUNK", CHUNK_PAIRS);
    const int ngpus = gab_pick_gpus(gpus);
    /* the slabs the ROI reads and writes are page-locked where the reference _mm_malloc's its own (main_banded.cpp:260-264) */
    gab_pin(ref, refUsed + 8); gab_pin(qry, qryUsed + 8);
    gab_pin(ref_off, 8 * (size_t)n); gab_pin(qry_off, 8 * (size_t)n);
    gab_pin(len1, 4 * (size_t)n); gab_pin(len2, 4 * (size_t)n); gab_pin(h0, 4 * (size_t)n); gab_pin_out(score, 4 * (size_t)n);
    gab_queue q;
    gab_queue_open(&q, ngpus, (n + ctx.chunk - 1) / ctx.chunk, gpu_init, run_chunk, gpu_fini, &ctx);     /* like `new BandedPairWiseSW` per thread: before the ROI */
    ctx.busy = (double *)calloc((size_t)q.nworkers, sizeof(double));

    /* ---- region of interest (main_banded.cpp:290-389) ---- */
    for (int64_t rep = gab_env_i64("GAB_ROI_WARMUPS", 0); rep > 0; rep--)       /* diagnosis only: untimed passes before the ROI */
        gab_queue_run(&q, (n + ctx.chunk - 1) / ctx.chunk);
    memset(ctx.busy, 0, sizeof(double) * (size_t)q.nworkers);
    const double t0 = gab_now();
    gab_roi_begin_n(ngpus);
    gab_queue_run(&q, (n + ctx.chunk - 1) / ctx.chunk);
    gab_roi_end();
    const double roi = gab_now() - t0;
    const int nworkers = q.nworkers;
    for (int k = 0; k < nworkers; k++) printf("%d] workTicks = %ld\n", k, (long)(ctx.busy[k] * 1e9));
    gab_queue_close(&q);
    gab_unpin(ref); gab_unpin(qry); gab_unpin(ref_off); gab_unpin(qry_off); gab_unpin(len1); gab_unpin(len2); gab_unpin(h0); gab_unpin(score);

    printf("Executed HIP gfx950 code on %d GPU(s)...\n", ngpus);
    for (int64_t i = 0; i < n; ++i) fprintf(stderr, "[%ld] score=%d\n", (long)i, score[i]);
    /* "cycles" are nanoseconds here: the frequency line says 1000 MHz so that cycles / freq is the ROI time */
    printf("Processor freq: %0.2lf MHz\n", 1000.0);
    printf("Read time = %0.2lf s\n", readTime);
    printf("Overall SW cycles = %ld, %0.2lf s\n", (long)(roi * 1e9), roi);
    printf("Total Pairs processed: %ld\n", (long)n);
    double sum = 0, mx = 0;
    for (int k = 0; k < nworkers; k++) { sum += ctx.busy[k]; if (ctx.busy[k] > mx) mx = ctx.busy[k]; }
    printf("avgTicks = %lf, maxTicks = %ld, load imbalance = %lf\n", sum * 1e9 / nworkers, (long)(mx * 1e9), sum > 0 ? mx / (sum / nworkers) : 1.0);
    free(ref); free(qry); free(ref_off); free(qry_off); free(len1); free(len2); free(h0); free(score);
    free(lineR); free(lineQ); free(ctx.busy);
    return 0;
}
