/* align_benchmark (wfa) -- drop-in driver of the wfa benchmark on MI355X.
 *
 * Command line, input format and output of /root/reference/benchmarks/wfa/tools/align_benchmark.c:
 *     align_benchmark -i <input> [-o <output>] [-p M,X,O,E] [--minimum-wavefront-length L
 *                     --maximum-difference-distance D] [-t <threads>] [-g <gpus>]
 * output "id=%d <run-length CIGAR>" per pair (edit_cigar_print, wfa/gap_affine/edit_cigar.c:184-200; the
 * harness sorts by id, wfa/scripts/regression_small.sh:94); stdout "Total.reads:", "Time.Benchmark:",
 * "Time.Alignment:" (align_benchmark.c:529-533).  --minimum-wavefront-length >= 0 selects the adaptive
 * reduction (align_benchmark.c:359-368), otherwise the complete mode.
 * The per-pair ROI call affine_wavefronts_align (align_benchmark.c:415-437) becomes gab_wfa_run_packed on chunks: the driver
 * only ever PRINTS the alignments, so the text edit_cigar_print would write is produced on the device and ~20 bytes per
 * 151-bp pair come back instead of the 302 bytes of operation room (GAB_WFA_UNPACKED=1: gab_wfa_run + encoding on the host).
 */
#include "../../common/gab_pairs.h"
#include <getopt.h>
#include <sys/time.h>

#define CHUNK_PAIRS (1 << 18)   /* default chunk of the work queue; $GAB_CHUNK overrides it */
typedef struct {
    const gab_pairs *p; int64_t chunk; gab_wfa_penalties pen; int min_wavefront_length, max_distance_threshold;
    char *ops; int64_t *ops_off; int32_t *ops_len, *score; int64_t max_seq_bytes, max_ops_bytes;
    /* packed mode: chunk c writes its CIGAR text into cig + cig_beg[c] (room cig_beg[c + 1] - cig_beg[c]); pair i's text is
     * cig_base[its chunk] + ops_off[i], ops_len[i] bytes; a chunk whose text outgrew its room gets a block of its own */
    int packed; char *cig; int64_t *cig_beg; char **cig_base;
} wfa_ctx;
static void *gpu_init(int worker, int gpu, void *vc) {
    (void)worker;
    wfa_ctx *c = (wfa_ctx *)vc; gab_wfa *h = NULL;
    GAB_DIE_IF(gab_wfa_create_reduced(&c->pen, c->min_wavefront_length, c->max_distance_threshold, gpu, &h), "gab_wfa_create_reduced");
    /* buffers for the largest chunk and warm copy queues, outside the ROI (the reference allocates its wavefronts before it too) */
    GAB_DIE_IF(gab_wfa_reserve(h, c->chunk < c->p->n ? c->chunk : c->p->n, c->max_seq_bytes, c->max_ops_bytes), "gab_wfa_reserve");
    return h;
}
static void gpu_fini(int worker, int gpu, void *c, void *st) { (void)worker; (void)gpu; (void)c; gab_wfa_destroy((gab_wfa *)st); }
static void run_chunk(int worker, int gpu, int64_t chunk, void *vctx, void *st) {
    (void)gpu; (void)worker;
    wfa_ctx *c = (wfa_ctx *)vctx;
    const int64_t b = chunk * c->chunk, e = b + c->chunk < c->p->n ? b + c->chunk : c->p->n;
    if (c->packed) {
        int64_t need = 0;
        c->cig_base[chunk] = c->cig + c->cig_beg[chunk];
        int rc = gab_wfa_run_packed((gab_wfa *)st, c->p->slab, c->p->off1 + b, c->p->len1 + b, c->p->slab, c->p->off2 + b, c->p->len2 + b, e - b,
                                    c->cig_base[chunk], c->cig_beg[chunk + 1] - c->cig_beg[chunk], c->ops_off + b, c->ops_len + b, c->score + b, &need);
        if (rc == GAB_ERANGE) {          /* more text than a quarter of the operation room (very divergent pairs): a block of its own */
            c->cig_base[chunk] = (char *)malloc((size_t)need + 1);
            if (!c->cig_base[chunk]) { fprintf(stderr, "ERROR: out of memory\n"); exit(EXIT_FAILURE); }
            rc = gab_wfa_run_packed((gab_wfa *)st, c->p->slab, c->p->off1 + b, c->p->len1 + b, c->p->slab, c->p->off2 + b, c->p->len2 + b, e - b,
                                    c->cig_base[chunk], need, c->ops_off + b, c->ops_len + b, c->score + b, &need);
        }
        GAB_DIE_IF(rc, "gab_wfa_run_packed");
        return;
    }
    GAB_DIE_IF(gab_wfa_run((gab_wfa *)st, c->p->slab, c->p->off1 + b, c->p->len1 + b, c->p->slab, c->p->off2 + b, c->p->len2 + b,
                           e - b, c->ops, c->ops_off + b, c->ops_len + b, c->score + b), "gab_wfa_run");
}
static double tv_now(void) { struct timeval tv; gettimeofday(&tv, NULL); return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec; }
/* ---- GAB_GPU_PARSE: the file cut in front of '>' lines, every GPU indexes its piece and aligns its pairs ----------------------- */
typedef struct {
    int dev, ok;
    gab_parser *ps; gab_pairs_packed pk; gab_wfa *h;
    char *d_ops, *cig; int64_t cig_cap; int32_t *score, *olen; int64_t *ooff;
} wgp_part;
typedef struct { int ng; const char *whole; size_t cut[65]; wgp_part part[64]; const wfa_ctx *wctx; } wgp_ctx;
static void wgp_parse(int g, void *v) {
    wgp_ctx *G = (wgp_ctx *)v;
    wgp_part *p = &G->part[g];
    p->dev = gab_phys_gpu(g);
    if (gab_parser_create(p->dev, &p->ps) != 0) return;
    if (gab_pairs_parse(p->ps, G->whole + G->cut[g], (int64_t)(G->cut[g + 1] - G->cut[g]), 0, &p->pk, NULL) != 0) return;
    gab_wfa_penalties pen; pen.mismatch = G->wctx->pen.mismatch; pen.gap_opening = G->wctx->pen.gap_opening; pen.gap_extension = G->wctx->pen.gap_extension;
    if (gab_wfa_create_reduced(&pen, G->wctx->min_wavefront_length, G->wctx->max_distance_threshold, p->dev, &p->h) != 0) return;
    const size_t n = (size_t)p->pk.n;
    if (gab_device_alloc(p->dev, (size_t)p->pk.cap_bytes + 16, (void **)&p->d_ops) != 0) return;
    /* the printed text of the piece: a quarter of its operation room (a 151-bp read pair at 2 % needs ~20 of its 302 bytes); page-locked
     * like the index, and -- with the handle's buffers and first launches (gab_wfa_reserve) -- made before the region of interest */
    p->cig_cap = ((int64_t)p->pk.cap_bytes / 4 + 4096 + 255) & ~(int64_t)255;
    p->cig = (char *)malloc((size_t)p->cig_cap + 16);
    p->ooff = (int64_t *)malloc(8 * n + 8); p->olen = (int32_t *)malloc(4 * n + 4); p->score = (int32_t *)malloc(4 * n + 4);
    if (!p->cig || !p->ooff || !p->olen || !p->score) return;
    gab_pin_out_on(p->dev, p->cig, (size_t)p->cig_cap + 16); gab_pin_out_on(p->dev, p->ooff, 8 * n + 8);
    gab_pin_out_on(p->dev, p->olen, 4 * n + 4); gab_pin_out_on(p->dev, p->score, 4 * n + 4);
    if (gab_wfa_reserve(p->h, p->pk.n, 0, p->cig_cap) != 0) return;
    p->ok = 1;
}
static void wgp_run(int g, void *v) {
    wgp_part *p = &((wgp_ctx *)v)->part[g];
    const gab_pairs_packed *pk = &p->pk;
    if (pk->n == 0) return;
    int64_t need = 0;
    int rc = gab_wfa_run_packed_device(p->h, pk->d_text, pk->text_bytes, pk->d_pat_off, pk->d_pat_len, pk->d_text, pk->text_bytes, pk->d_txt_off,
                                       pk->d_txt_len, pk->n, p->d_ops, pk->d_cap_off, p->cig, p->cig_cap, p->ooff, p->olen, p->score, &need);
    if (rc == GAB_ERANGE) {              /* more text than a quarter of the operation room (very divergent pairs): room for all of it */
        gab_unpin(p->cig); free(p->cig);
        p->cig_cap = need; p->cig = (char *)malloc((size_t)need + 16);
        if (!p->cig) { fprintf(stderr, "ERROR: out of memory\n"); exit(EXIT_FAILURE); }
        rc = gab_wfa_run_packed_device(p->h, pk->d_text, pk->text_bytes, pk->d_pat_off, pk->d_pat_len, pk->d_text, pk->text_bytes, pk->d_txt_off,
                                       pk->d_txt_len, pk->n, p->d_ops, pk->d_cap_off, p->cig, p->cig_cap, p->ooff, p->olen, p->score, &need);
        gab_pin(p->cig, (size_t)need + 16);         /* (so that the clean-up below unpins what it expects) */
    }
    GAB_DIE_IF(rc, "gab_wfa_run_packed_device");
}

int main(int argc, char **argv) {
    const char *input = NULL, *output = NULL;
    int threads = 1, gpus = 0, c;
    wfa_ctx ctx;
    ctx.pen.mismatch = 4; ctx.pen.gap_opening = 6; ctx.pen.gap_extension = 2;      /* align_benchmark.c:85-90 */
    ctx.min_wavefront_length = -1; ctx.max_distance_threshold = -1;                /* :91-92: complete mode */
    int match = 0;
    static struct option lo[] = {{"input", required_argument, 0, 'i'}, {"output", required_argument, 0, 'o'},
                                 {"affine-penalties", required_argument, 0, 'p'},
                                 {"minimum-wavefront-length", required_argument, 0, 1000},
                                 {"maximum-difference-distance", required_argument, 0, 1001},
                                 {"nthreads", required_argument, 0, 't'}, {"gpus", required_argument, 0, 'g'},
                                 {"progress", required_argument, 0, 'P'}, {"verbose", no_argument, 0, 'v'},
                                 {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
    if (argc <= 1) { fprintf(stderr, "USE: ./align_benchmark -i <input> [-o <output>] [-p M,X,O,E] [--minimum-wavefront-length <INT> --maximum-difference-distance <INT>] [-t <threads>] [-g <gpus>]\n"); exit(0); }
    while ((c = getopt_long(argc, argv, "i:o:p:t:g:P:vh", lo, NULL)) != -1) {
        switch (c) {
            case 'i': input = optarg; break;
            case 'o': output = optarg; break;
            case 'p': {
                char *s = strtok(optarg, ","); match = s ? atoi(s) : 0;
                s = strtok(NULL, ","); if (s) ctx.pen.mismatch = atoi(s);
                s = strtok(NULL, ","); if (s) ctx.pen.gap_opening = atoi(s);
                s = strtok(NULL, ","); if (s) ctx.pen.gap_extension = atoi(s);
                break;
            }
            case 1000: ctx.min_wavefront_length = atoi(optarg); break;       /* align_benchmark.c:267-272 */
            case 1001: ctx.max_distance_threshold = atoi(optarg); break;
            case 't': threads = atoi(optarg); break;
            case 'g': gpus = atoi(optarg); break;
            case 'P': case 'v': break;
            case 'h': fprintf(stderr, "USE: ./align_benchmark -i <input> [-o <output>] [-p M,X,O,E] [--minimum-wavefront-length <INT> --maximum-difference-distance <INT>] [-t <threads>] [-g <gpus>]\n"); exit(1);
            default: fprintf(stderr, "Option not recognized\n"); exit(1);
        }
    }
    (void)threads;
    if (match > 0) { fprintf(stderr, "Match score must be negative or zero (M=%d)\n", match); exit(1); }
    if (!input) { fprintf(stderr, "Option --input is required \n"); exit(1); }
    const double bench0 = tv_now();
    FILE *in = fopen(input, "r");
    if (!in) { fprintf(stderr, "Input file '%s' couldn't be opened\n", input); exit(1); }
    FILE *out = output ? fopen(output, "w") : NULL;
    /* GAB_GPU_PARSE=1: the file is read in one piece, cut in front of '>' lines into one piece per GPU, and every GPU indexes ITS
     * piece (gab_pairs_parse, no swap); sequences are used in place in that GPU's copy of the text, the CIGARs
     * of a piece come back in one copy. */
    const int64_t fsz = gab_regular_file_size(in);      /* -1 for pipes: they take the getline path */
    if (gab_gpu_parse_wanted(1) && fsz >= 0) {
        const int ng = gab_pick_gpus(gpus);
        char *whole = (char *)malloc((size_t)fsz + 1);
        wgp_ctx G;
        memset(&G, 0, sizeof G);
        G.ng = ng; G.whole = whole; G.wctx = &ctx;
        int ok = whole && fread(whole, 1, (size_t)fsz, in) == (size_t)fsz && gab_cut_at_marker(whole, (size_t)fsz, ng, ">", 0, G.cut) == 0;
        if (ok) {
            gab_run_parts(ng, wgp_parse, &G);
            for (int g = 0; g < ng; g++) ok = ok && G.part[g].ok;
        }
        if (ok) {
            free(whole); fclose(in);
            const double t0g = tv_now();
            gab_roi_begin_n(ng);
            gab_run_parts(ng, wgp_run, &G);
            gab_roi_end();
            const double t1g = tv_now();
            int64_t n = 0;
            for (int g = 0; g < ng; g++) {
                const wgp_part *q = &G.part[g];
                if (out)
                    for (int64_t i = 0; i < q->pk.n; i++) {
                        fprintf(out, "id=%ld ", (long)(n + i));
                        fwrite(q->cig + q->ooff[i], 1, (size_t)q->olen[i], out);        /* already the printed text */
                        fprintf(out, "\n");
                    }
                n += q->pk.n;
            }
            if (out) fclose(out);
            printf("Total.reads: %ld\n", (long)n);
            printf("Time.Benchmark: %f s\n", tv_now() - bench0);
            printf("Time.Alignment: %f s (input indexed on the GPU)\n", t1g - t0g);
            if (getenv("GAB_QUEUE_REPORT")) {
                fprintf(stderr, "gab GPU parse: %d piece(s), pairs per GPU:", ng);
                for (int g = 0; g < ng; g++) fprintf(stderr, " %ld", (long)G.part[g].pk.n);
                fprintf(stderr, "\n");
            }
            for (int g = 0; g < ng; g++) {
                wgp_part *q = &G.part[g];
                gab_device_free(q->dev, q->d_ops);
                gab_wfa_destroy(q->h); gab_parser_destroy(q->ps);
                gab_unpin(q->cig); gab_unpin(q->ooff); gab_unpin(q->olen); gab_unpin(q->score);
                free(q->cig); free(q->ooff); free(q->olen); free(q->score);
            }
            return 0;
        }
        if (getenv("GAB_GPU_PARSE")) fprintf(stderr, "GPU parser declined the file (%s); using the getline parser\n", gab_last_error());      /* (asked for by name: say so; the default falls back silently) */
        for (int g = 0; g < ng; g++) {
            wgp_part *q = &G.part[g];
            if (q->d_ops) gab_device_free(q->dev, q->d_ops);
            if (q->h) gab_wfa_destroy(q->h);
            if (q->ps) gab_parser_destroy(q->ps);
            if (q->cig) gab_unpin(q->cig);
            if (q->ooff) gab_unpin(q->ooff);
            if (q->olen) gab_unpin(q->olen);
            if (q->score) gab_unpin(q->score);
            free(q->cig); free(q->ooff); free(q->olen); free(q->score);
        }
        free(whole);
        fseek(in, 0L, SEEK_SET);
    }
    gab_pairs p;
    gab_pairs_read(in, &p);
    fclose(in);
    ctx.p = &p;
    ctx.packed = !gab_env_i64("GAB_WFA_UNPACKED", 0);
    ctx.ops_off = (int64_t *)malloc(8 * (size_t)p.n + 8);
    ctx.ops_len = (int32_t *)malloc(4 * (size_t)p.n + 4); ctx.score = (int32_t *)malloc(4 * (size_t)p.n + 4);
    int64_t tot = 0;
    for (int64_t i = 0; i < p.n; i++) { ctx.ops_off[i] = tot; tot += (int64_t)p.len1[i] + p.len2[i]; }
    const int ngpus = gab_pick_gpus(gpus);
    ctx.chunk = gab_env_i64("GAB_CHUNK", CHUNK_PAIRS);
    const int64_t nchunks = (p.n + ctx.chunk - 1) / ctx.chunk;
    ctx.cig_beg = (int64_t *)calloc((size_t)nchunks + 1, 8); ctx.cig_base = (char **)calloc((size_t)nchunks + 1, sizeof(char *));
    ctx.max_seq_bytes = ctx.max_ops_bytes = 0;       /* the widest chunk: what gpu_init reserves for */
    for (int64_t b = 0, k = 0; b < p.n; b += ctx.chunk, k++) {
        const int64_t e = b + ctx.chunk < p.n ? b + ctx.chunk : p.n;
        const int64_t lo = p.off1[b] < p.off2[b] ? p.off1[b] : p.off2[b];
        const int64_t h1 = p.off1[e - 1] + p.len1[e - 1], h2 = p.off2[e - 1] + p.len2[e - 1], hi = h1 > h2 ? h1 : h2;
        int64_t ops = ctx.ops_off[e - 1] + p.len1[e - 1] + p.len2[e - 1] - ctx.ops_off[b];
        /* text room of the chunk: a quarter of its operation room (a 151-bp read pair at 2 % needs ~20 of its 302 bytes) */
        ctx.cig_beg[k + 1] = ctx.cig_beg[k] + ((ops / 4 + 4096 + 255) & ~(int64_t)255);
        if (ctx.packed) {                            /* on the device: fixed-stride operation room + the text */
            int64_t stride = 0;
            for (int64_t i = b; i < e; i++) if ((int64_t)p.len1[i] + p.len2[i] > stride) stride = (int64_t)p.len1[i] + p.len2[i];
            ops = ((stride + 7) & ~(int64_t)7) * (e - b) + (ctx.cig_beg[k + 1] - ctx.cig_beg[k]);
        }
        if (hi - lo + 512 > ctx.max_seq_bytes) ctx.max_seq_bytes = hi - lo + 512;
        if (ops + 512 > ctx.max_ops_bytes) ctx.max_ops_bytes = ops + 512;
    }
    if (ctx.packed) { ctx.ops = NULL; ctx.cig = (char *)malloc((size_t)ctx.cig_beg[nchunks] + 16); }
    else { ctx.cig = NULL; ctx.ops = (char *)malloc((size_t)tot + 16); }
    if (!ctx.cig && !ctx.ops) { fprintf(stderr, "ERROR: out of memory\n"); exit(EXIT_FAILURE); }
    gab_pin(p.slab, p.used); gab_pin(p.off1, 8 * (size_t)p.n); gab_pin(p.off2, 8 * (size_t)p.n); gab_pin(p.len1, 4 * (size_t)p.n);
    gab_pin(p.len2, 4 * (size_t)p.n); gab_pin(ctx.ops_off, 8 * (size_t)p.n);
    if (ctx.packed) gab_pin_out(ctx.cig, (size_t)ctx.cig_beg[nchunks] + 16); else gab_pin_out(ctx.ops, (size_t)tot + 16);
    gab_pin_out(ctx.ops_len, 4 * (size_t)p.n); gab_pin_out(ctx.score, 4 * (size_t)p.n);
    gab_queue q;
    gab_queue_open(&q, ngpus, nchunks, gpu_init, run_chunk, gpu_fini, &ctx);
    for (int64_t rep = gab_env_i64("GAB_ROI_WARMUPS", 0); rep > 0; rep--) gab_queue_run(&q, nchunks);       /* diagnosis only: untimed passes before the ROI */
    const double t0 = tv_now();                  /* ROI: align_benchmark.c:378-491 */
    gab_roi_begin_n(ngpus);
    gab_queue_run(&q, nchunks);
    gab_roi_end();
    const double t1 = tv_now();
    gab_queue_close(&q);
    gab_unpin(p.slab); gab_unpin(p.off1); gab_unpin(p.off2); gab_unpin(p.len1); gab_unpin(p.len2); gab_unpin(ctx.packed ? ctx.cig : ctx.ops);
    gab_unpin(ctx.ops_off); gab_unpin(ctx.ops_len); gab_unpin(ctx.score);
    if (out) {
        for (int64_t i = 0; i < p.n; i++) {
            fprintf(out, "id=%ld ", (long)i);
            const int n = ctx.ops_len[i];
            /* edit_cigar_print reads operations[begin] even when the CIGAR is empty; an empty pair prints nothing here */
            if (ctx.packed) fwrite(ctx.cig_base[i / ctx.chunk] + ctx.ops_off[i], 1, (size_t)n, out);       /* already the printed text */
            else {
                const char *o = ctx.ops + ctx.ops_off[i];
                for (int k = 0; k < n;) { int r = k; while (r < n && o[r] == o[k]) r++; fprintf(out, "%d%c", r - k, o[k]); k = r; }
            }
            fprintf(out, "\n");
        }
        fclose(out);
    }
    for (int64_t k = 0; k < nchunks; k++) if (ctx.packed && ctx.cig_base[k] && ctx.cig_base[k] != ctx.cig + ctx.cig_beg[k]) free(ctx.cig_base[k]);
    printf("Total.reads: %ld\n", (long)p.n);
    printf("Time.Benchmark: %f s\n", tv_now() - bench0);
    printf("Time.Alignment: %f s\n", t1 - t0);
    free(ctx.ops); free(ctx.cig); free(ctx.cig_beg); free(ctx.cig_base); free(ctx.ops_off); free(ctx.ops_len); free(ctx.score);
    gab_pairs_free(&p);
    return 0;
}
