/* align_benchmark (bpm) -- drop-in driver of the bpm benchmark on MI355X.
 *
 * Command line, input format, output lines and the timing line the harness greps are those of
 * /root/reference/benchmarks/bpm/tools/align_benchmark.c:
 *     align_benchmark -a bpm-edit|bitpal-edit|bitpal-scored -i <input> [-o <output>] [-t <threads>] [-g <gpus>]
 * output "[id] score=%d" per pair (the harness sorts by id, bpm/scripts/regression_small.sh:94), stderr
 * "[Benchmark]", "=> Total.reads", "=> Time.Benchmark" formatted like timer_print
 * (bpm/system/profiler_timer.c:110-175).
 * The per-pair ROI call benchmark_edit_bpm (align_benchmark.c:243-257) becomes gab_bpm_run on chunks of
 * pairs, one host thread per GPU; benchmark_bitpal_m0_x1_g1 / _m1_x4_g2 (-a bitpal-edit / bitpal-scored, :259-264)
 * become gab_bitpal_run the same way.  The driver applies the reference's swap: the longer line is the pattern.
 */
#define GAB_ENERGY_STREAM stderr      /* where the reference prints "Energy consumption:" in this driver */
#include "../../common/gab_pairs.h"
#include <getopt.h>

#define CHUNK_PAIRS (1 << 20)   /* default chunk of the work queue; $GAB_CHUNK overrides it */
typedef struct {
    const gab_pairs *p;
    int64_t *poff, *toff; int32_t *plen, *tlen;
    int32_t *score;
    int bitpal;                  /* -1: bpm-edit, else GAB_BITPAL_EDIT / GAB_BITPAL_SCORED */
    int64_t chunk, max_seq_bytes;
} bpm_ctx;
static void *gpu_init(int worker, int gpu, void *vc) {
    (void)worker;
    bpm_ctx *c = (bpm_ctx *)vc;
    /* buffers for the largest chunk and warm copy queues, outside the ROI */
    const int64_t np = c->chunk < c->p->n ? c->chunk : c->p->n;
    if (c->bitpal >= 0) {
        gab_bitpal *h = NULL; GAB_DIE_IF(gab_bitpal_create(c->bitpal, gpu, &h), "gab_bitpal_create");
        GAB_DIE_IF(gab_bitpal_reserve(h, np, c->max_seq_bytes), "gab_bitpal_reserve");
        return h;
    }
    gab_bpm *h = NULL; GAB_DIE_IF(gab_bpm_create(gpu, &h), "gab_bpm_create");
    GAB_DIE_IF(gab_bpm_reserve(h, np, c->max_seq_bytes), "gab_bpm_reserve");
    return h;
}
static void gpu_fini(int worker, int gpu, void *vc, void *st) {
    (void)gpu; (void)worker;
    if (((bpm_ctx *)vc)->bitpal >= 0) gab_bitpal_destroy((gab_bitpal *)st); else gab_bpm_destroy((gab_bpm *)st);
}
static void run_chunk(int worker, int gpu, int64_t chunk, void *vctx, void *st) {
    (void)gpu; (void)worker;
    bpm_ctx *c = (bpm_ctx *)vctx;
    const int64_t b = chunk * c->chunk, e = b + c->chunk < c->p->n ? b + c->chunk : c->p->n;
    if (c->bitpal >= 0)
        GAB_DIE_IF(gab_bitpal_run((gab_bitpal *)st, c->p->slab, c->poff + b, c->plen + b, c->p->slab, c->toff + b, c->tlen + b, e - b,
                                  c->score + b), "gab_bitpal_run");
    else
        GAB_DIE_IF(gab_bpm_run((gab_bpm *)st, c->p->slab, c->poff + b, c->plen + b, c->p->slab, c->toff + b, c->tlen + b, e - b,
                               c->score + b), "gab_bpm_run");
}
static void timer_print_like(FILE *f, double sec) {
    const double ns = sec * 1e9;
    if (ns >= 60e9) fprintf(f, "%7.2f m ", sec / 60.0);
    else if (ns >= 1e9) fprintf(f, "%7.2f s ", sec);
    else if (ns >= 1e6) fprintf(f, "%7.2f ms", sec * 1e3);
    else if (ns >= 1e3) fprintf(f, "%7.2f us", sec * 1e6);
    else fprintf(f, "%7.0f ns", ns);
    fprintf(f, " (    1   call,");
    if (ns > 1e9) fprintf(f, "%7.2f  s/call {min%.2fs,Max%.2fs})\n", sec, sec, sec);
    else if (ns > 1e6) fprintf(f, "%7.2f ms/call {min%.2fms,Max%.2fms})\n", sec * 1e3, sec * 1e3, sec * 1e3);
    else fprintf(f, "%7.2f us/call {min%.2fus,Max%.2fus})\n", sec * 1e6, sec * 1e6, sec * 1e6);
}
/* ---- GAB_GPU_PARSE: the file cut in front of '>' lines, every GPU indexes its piece and scores its pairs ------------------------ */
typedef struct {
    int dev, ok; int64_t first;
    gab_parser *ps; gab_pairs_packed pk; gab_bpm *h; gab_bitpal *hb; int32_t *d_score;
} bgp_part;
typedef struct { int ng, bitpal; const char *whole; size_t cut[65]; bgp_part part[64]; int32_t *score; } bgp_ctx;
static void bgp_parse(int g, void *v) {
    bgp_ctx *G = (bgp_ctx *)v;
    bgp_part *p = &G->part[g];
    p->dev = gab_phys_gpu(g);
    if (gab_parser_create(p->dev, &p->ps) != 0) return;
    if (gab_pairs_parse(p->ps, G->whole + G->cut[g], (int64_t)(G->cut[g + 1] - G->cut[g]), 1, &p->pk, NULL) != 0) return;
    if (G->bitpal >= 0) { if (gab_bitpal_create(G->bitpal, p->dev, &p->hb) != 0) return; }
    else if (gab_bpm_create(p->dev, &p->h) != 0) return;
    if (gab_device_alloc(p->dev, 4 * (size_t)p->pk.n + 4, (void **)&p->d_score) != 0) return;
    if (G->bitpal >= 0 ? gab_bitpal_reserve(p->hb, p->pk.n, 0) != 0 : gab_bpm_reserve(p->h, p->pk.n, 0) != 0) return;   /* before the region of interest */
    p->ok = 1;
}
static void bgp_run(int g, void *v) {
    bgp_ctx *G = (bgp_ctx *)v;
    bgp_part *p = &G->part[g];
    const gab_pairs_packed *pk = &p->pk;
    if (pk->n == 0) return;
    if (G->bitpal >= 0)
        GAB_DIE_IF(gab_bitpal_run_device(p->hb, pk->d_text, pk->text_bytes, pk->d_pat_off, pk->d_pat_len, pk->d_text, pk->text_bytes,
                                         pk->d_txt_off, pk->d_txt_len, pk->n, p->d_score, NULL), "gab_bitpal_run_device");
    else
        GAB_DIE_IF(gab_bpm_run_device(p->h, pk->d_text, pk->text_bytes, pk->d_pat_off, pk->d_pat_len, pk->d_text, pk->text_bytes, pk->d_txt_off,
                                      pk->d_txt_len, pk->n, p->d_score, NULL), "gab_bpm_run_device");
    GAB_DIE_IF(gab_device_copy_to_host(p->dev, G->score + p->first, p->d_score, 4 * (size_t)pk->n), "gab_device_copy_to_host");
}

int main(int argc, char **argv) {
    const char *algo = NULL, *input = NULL, *output = NULL;
    int threads = 1, gpus = 0, c;
    static struct option lo[] = {{"algorithm", required_argument, 0, 'a'}, {"input", required_argument, 0, 'i'},
                                 {"output", required_argument, 0, 'o'}, {"threads", required_argument, 0, 't'},
                                 {"gpus", required_argument, 0, 'g'}, {"progress", required_argument, 0, 'P'},
                                 {"verbose", no_argument, 0, 'v'}, {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
    if (argc <= 1) { fprintf(stderr, "USE: ./align_benchmark -a <algorithm> -i <input> [-o <output>] [-t <threads>] [-g <gpus>]\n"); exit(0); }
    while ((c = getopt_long(argc, argv, "a:i:o:t:g:P:vh", lo, NULL)) != -1) {
        switch (c) {
            case 'a': algo = optarg; break;
            case 'i': input = optarg; break;
            case 'o': output = optarg; break;
            case 't': threads = atoi(optarg); break;
            case 'g': gpus = atoi(optarg); break;
            case 'P': case 'v': break;
            case 'h': fprintf(stderr, "USE: ./align_benchmark -a bpm-edit|bitpal-edit|bitpal-scored -i <input> [-o <output>] [-t <threads>] [-g <gpus>]\n"); exit(1);
            default: fprintf(stderr, "Option not recognized\n"); exit(1);
        }
    }
    (void)threads;
    if (!algo) { fprintf(stderr, "Option --algorithm is required \n"); exit(1); }
    int bitpal = -1;                             /* align_benchmark.c:330-338 */
    if (strcmp(algo, "bitpal-edit") == 0) bitpal = GAB_BITPAL_EDIT;
    else if (strcmp(algo, "bitpal-scored") == 0) bitpal = GAB_BITPAL_SCORED;
    else if (strcmp(algo, "bpm-edit") != 0) { fprintf(stderr, "Algorithm '%s' not recognized\n", algo); exit(1); }
    if (!input) { fprintf(stderr, "Option --input is required \n"); exit(1); }
    FILE *in = fopen(input, "r");
    if (!in) { fprintf(stderr, "Input file '%s' couldn't be opened\n", input); exit(1); }
    FILE *out = output ? fopen(output, "w") : NULL;
    /* GAB_GPU_PARSE=1: the file is read in one piece, cut in front of '>' lines into one piece per GPU, and every GPU indexes ITS
     * piece (gab_pairs_parse with the swap rule); the sequences are used in place in that GPU's copy of the text. */
    const int64_t fsz = gab_regular_file_size(in);      /* -1 for pipes: they take the getline path */
    if (gab_gpu_parse_wanted(1) && fsz >= 0) {
        const int ng = gab_pick_gpus(gpus);
        char *whole = (char *)malloc((size_t)fsz + 1);
        bgp_ctx G;
        memset(&G, 0, sizeof G);
        G.ng = ng; G.whole = whole; G.bitpal = bitpal;
        int ok = whole && fread(whole, 1, (size_t)fsz, in) == (size_t)fsz && gab_cut_at_marker(whole, (size_t)fsz, ng, ">", 0, G.cut) == 0;
        if (ok) {
            gab_run_parts(ng, bgp_parse, &G);
            for (int g = 0; g < ng; g++) ok = ok && G.part[g].ok;
        }
        if (ok) {
            free(whole); fclose(in);
            int64_t n = 0;
            for (int g = 0; g < ng; g++) { G.part[g].first = n; n += G.part[g].pk.n; }
            G.score = (int32_t *)malloc(4 * (size_t)n + 4);
            gab_pin_out_on(gab_phys_gpu(0), G.score, 4 * (size_t)n + 4);
            const double t0g = gab_now();
            gab_roi_begin_n(ng);
            gab_run_parts(ng, bgp_run, &G);
            gab_roi_end();
            const double secg = gab_now() - t0g;
            gab_unpin(G.score);
            if (out) { for (int64_t i = 0; i < n; i++) fprintf(out, "[%ld] score=%d\n", (long)i, G.score[i]); fclose(out); }
            fprintf(stderr, "[Benchmark] (input indexed on the GPU)\n");
            fprintf(stderr, "=> Total.reads            %ld\n", (long)n);
            fprintf(stderr, "=> Time.Benchmark      ");
            timer_print_like(stderr, secg);
            if (getenv("GAB_QUEUE_REPORT")) {
                fprintf(stderr, "gab GPU parse: %d piece(s), pairs per GPU:", ng);
                for (int g = 0; g < ng; g++) fprintf(stderr, " %ld", (long)G.part[g].pk.n);
                fprintf(stderr, "\n");
            }
            for (int g = 0; g < ng; g++) { bgp_part *q = &G.part[g]; gab_device_free(q->dev, q->d_score); gab_bpm_destroy(q->h); gab_bitpal_destroy(q->hb); gab_parser_destroy(q->ps); }
            free(G.score);
            return 0;
        }
        if (getenv("GAB_GPU_PARSE")) fprintf(stderr, "GPU parser declined the file (%s); using the getline parser\n", gab_last_error());      /* (asked for by name: say so; the default falls back silently) */
        for (int g = 0; g < ng; g++) { bgp_part *q = &G.part[g]; if (q->d_score) gab_device_free(q->dev, q->d_score); gab_bpm_destroy(q->h); gab_bitpal_destroy(q->hb); if (q->ps) gab_parser_destroy(q->ps); }
        free(whole);
        fseek(in, 0L, SEEK_SET);
    }
    gab_pairs p;
    gab_pairs_read(in, &p);
    fclose(in);
    bpm_ctx ctx;
    ctx.p = &p; ctx.bitpal = bitpal;
    ctx.poff = (int64_t *)malloc(8 * (size_t)p.n + 8); ctx.toff = (int64_t *)malloc(8 * (size_t)p.n + 8);
    ctx.plen = (int32_t *)malloc(4 * (size_t)p.n + 4); ctx.tlen = (int32_t *)malloc(4 * (size_t)p.n + 4);
    ctx.score = (int32_t *)malloc(4 * (size_t)p.n + 4);
    for (int64_t i = 0; i < p.n; i++) {          /* swap: the longer LINE is the pattern (align_benchmark.c:177-181) */
        const int sw = p.len1[i] < p.len2[i];
        ctx.poff[i] = sw ? p.off2[i] : p.off1[i]; ctx.plen[i] = sw ? p.len2[i] : p.len1[i];
        ctx.toff[i] = sw ? p.off1[i] : p.off2[i]; ctx.tlen[i] = sw ? p.len1[i] : p.len2[i];
    }
    const int ngpus = gab_pick_gpus(gpus);
    ctx.chunk = gab_env_i64("GAB_CHUNK", CHUNK_PAIRS);
    ctx.max_seq_bytes = 0;                       /* the widest chunk: what gpu_init reserves for */
    for (int64_t b = 0; b < p.n; b += ctx.chunk) {
        const int64_t e = b + ctx.chunk < p.n ? b + ctx.chunk : p.n;
        const int64_t lo = p.off1[b] < p.off2[b] ? p.off1[b] : p.off2[b];
        const int64_t h1 = p.off1[e - 1] + p.len1[e - 1], h2 = p.off2[e - 1] + p.len2[e - 1], hi = h1 > h2 ? h1 : h2;
        if (hi - lo + 512 > ctx.max_seq_bytes) ctx.max_seq_bytes = hi - lo + 512;
    }
    gab_pin(p.slab, p.used); gab_pin(ctx.poff, 8 * (size_t)p.n); gab_pin(ctx.toff, 8 * (size_t)p.n);
    gab_pin(ctx.plen, 4 * (size_t)p.n); gab_pin(ctx.tlen, 4 * (size_t)p.n); gab_pin_out(ctx.score, 4 * (size_t)p.n);
    gab_queue q;
    gab_queue_open(&q, ngpus, (p.n + ctx.chunk - 1) / ctx.chunk, gpu_init, run_chunk, gpu_fini, &ctx);
    const double t0 = gab_now();                 /* ROI: align_benchmark.c:213-337 */
    gab_roi_begin_n(ngpus);
    gab_queue_run(&q, (p.n + ctx.chunk - 1) / ctx.chunk);
    gab_roi_end();
    const double sec = gab_now() - t0;
    gab_queue_close(&q);
    gab_unpin(p.slab); gab_unpin(ctx.poff); gab_unpin(ctx.toff); gab_unpin(ctx.plen); gab_unpin(ctx.tlen); gab_unpin(ctx.score);
    if (out) { for (int64_t i = 0; i < p.n; i++) fprintf(out, "[%ld] score=%d\n", (long)i, ctx.score[i]); fclose(out); }
    fprintf(stderr, "[Benchmark]\n");
    fprintf(stderr, "=> Total.reads            %ld\n", (long)p.n);
    fprintf(stderr, "=> Time.Benchmark      ");
    timer_print_like(stderr, sec);
    free(ctx.poff); free(ctx.toff); free(ctx.plen); free(ctx.tlen); free(ctx.score);
    gab_pairs_free(&p);
    return 0;
}
