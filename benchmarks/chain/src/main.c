/* chain / fast-chain -- drop-in driver of the two seed-chaining benchmarks on MI355X.
 *
 * Same command line (-i in -o out -t threads [-h]), input/output format and timing line as
 * /root/reference/benchmarks/chain/src/main.cpp and fast-chain/src/main.cpp (the two are byte-identical);
 * text I/O as chain/src/host_data_io.cpp:13-60.  The harness compares out.txt and greps "Time in kernel"
 * (chain/scripts/regression_small.sh:89,94).
 * The ROI call host_chain_kernel(calls, rets, numThreads) (main.cpp:154) becomes
 *   one GPU:   ONE gab_chain_run over all calls;
 *   N GPUs:    one SHARE of the calls per GPU and one gab_chain_run per share.  Calls differ in size by three orders of
 *              magnitude and a batch cannot finish before its longest call, so the calls are sorted by descending anchor count
 *              and dealt, longest first, to the GPU with the least anchors so far (the reference's analogue is
 *              `omp for schedule(dynamic)` over calls, chain/src/host_kernel.cpp:98-105).  Each share is gathered into page-locked
 *              slabs of its own BEFORE the ROI (where the reference's reader builds its vectors, main.cpp:120-150) by a thread
 *              bound to its GPU's NUMA node, so the card reads memory next to it; results are printed from the shares;
 *   $GAB_CHUNK (anchors per chunk; tests): file-order chunks pulled by $GAB_WORKERS_PER_GPU host threads per GPU.
 * Built twice: -DGAB_CHAIN_MODE=0 (chain) and =1 (fast-chain).
 * Extra flag: -g <gpus> (or $GAB_GPUS).  -t is accepted and ignored.  GAB_GPU_PARSE=1: see main().
 */
#define GAB_ENERGY_STREAM stderr      /* where the reference prints "Energy consumption:" in this driver */
#include "../../common/gab_driver.h"
#include <getopt.h>

#ifndef GAB_CHAIN_MODE
#define GAB_CHAIN_MODE GAB_CHAIN
#endif

typedef struct {
    const uint64_t *x, *y;
    const int64_t *call_off;
    const gab_chain_hdr *hdr;
    int32_t *score, *parent;
    int64_t ncalls;
    int64_t *chunk_beg;      /* chunk c = calls [chunk_beg[c], chunk_beg[c+1]) */
    int64_t max_chunk_anchors, max_chunk_calls;
    struct chain_share *shares;      /* N-GPU mode: chunk g = share g */
} chain_ctx;

/* one GPU's share of the calls, in slabs of its own */
typedef struct chain_share {
    int gpu;                     /* device whose NUMA node the slabs were first touched on */
    int64_t ncalls, na;
    int64_t *call_id;            /* share-local call k is call call_id[k] of the file (ascending) */
    int64_t *off;                /* first anchor of share-local call k in the share's slabs */
    gab_chain_hdr *hdr;
    uint64_t *x, *y;
    int32_t *score, *parent;
    const chain_ctx *src;
} chain_share;

/* gather a share (thread bound to the GPU's node: malloc + first touch put the pages there; then page-lock) */
static void *share_build(void *p) {
    chain_share *sh = (chain_share *)p;
    const chain_ctx *c = sh->src;
    (void)gab_bind_thread_to_gpu(sh->gpu);
    const size_t na = (size_t)sh->na;
    sh->x = (uint64_t *)malloc(8 * (na + 1)); sh->y = (uint64_t *)malloc(8 * (na + 1));
    sh->score = (int32_t *)malloc(4 * (na + 1)); sh->parent = (int32_t *)malloc(4 * (na + 1));
    if (!sh->x || !sh->y || !sh->score || !sh->parent) { fprintf(stderr, "ERROR: out of memory for a GPU's share\n"); exit(EXIT_FAILURE); }
    for (int64_t k = 0; k < sh->ncalls; k++) {
        const int64_t id = sh->call_id[k], n = c->hdr[id].n;
        memcpy(sh->x + sh->off[k], c->x + c->call_off[id], 8 * (size_t)n);
        memcpy(sh->y + sh->off[k], c->y + c->call_off[id], 8 * (size_t)n);
    }
    gab_pin(sh->x, 8 * na); gab_pin(sh->y, 8 * na);
    gab_pin_out_on(sh->gpu, sh->score, 4 * na); gab_pin_out_on(sh->gpu, sh->parent, 4 * na);
    return NULL;
}
static void run_share(int worker, int gpu, int64_t chunk, void *vctx, void *st) {
    (void)gpu; (void)worker;
    chain_share *sh = &((chain_ctx *)vctx)->shares[chunk];
    if (sh->ncalls == 0) return;
    GAB_DIE_IF(gab_chain_run((gab_chain *)st, GAB_CHAIN_MODE, sh->x, sh->y, sh->off, sh->hdr, sh->ncalls, sh->score, sh->parent), "gab_chain_run");
}
static int by_n_desc(const void *a, const void *b, void *hdr_) {
    const gab_chain_hdr *hdr = (const gab_chain_hdr *)hdr_;
    const int64_t ia = *(const int64_t *)a, ib = *(const int64_t *)b;
    if (hdr[ia].n != hdr[ib].n) return hdr[ia].n > hdr[ib].n ? -1 : 1;
    return ia < ib ? -1 : ia > ib;              /* (stable: equal calls keep the file's order) */
}

static void *gpu_init(int worker, int gpu, void *vctx) {
    (void)worker;
    chain_ctx *c = (chain_ctx *)vctx;
    gab_chain *h = NULL;
    GAB_DIE_IF(gab_chain_create(gpu, &h), "gab_chain_create");
    GAB_DIE_IF(gab_chain_reserve(h, c->max_chunk_anchors, c->max_chunk_calls), "gab_chain_reserve");   /* buffers before the ROI */
    return h;
}
static void gpu_fini(int worker, int gpu, void *vctx, void *st) { (void)worker; (void)gpu; (void)vctx; gab_chain_destroy((gab_chain *)st); }
static void run_chunk(int worker, int gpu, int64_t chunk, void *vctx, void *st) {
    (void)gpu; (void)worker;
    chain_ctx *c = (chain_ctx *)vctx;
    const int64_t b = c->chunk_beg[chunk], e = c->chunk_beg[chunk + 1];
    if (e <= b) return;
    /* window of the anchor arrays owned by these calls; offsets re-based to the window */
    const int64_t a0 = c->call_off[b], a1 = c->call_off[e - 1] + c->hdr[e - 1].n;
    int64_t *off = (int64_t *)malloc(8 * (size_t)(e - b));
    for (int64_t k = b; k < e; k++) off[k - b] = c->call_off[k] - a0;
    (void)a1;
    GAB_DIE_IF(gab_chain_run((gab_chain *)st, GAB_CHAIN_MODE, c->x + a0, c->y + a0, off, c->hdr + b, e - b,
                             c->score + a0, c->parent + a0), "gab_chain_run");
    free(off);
}

/* ---- GAB_GPU_PARSE: the file cut behind "EOR" lines, every GPU parses its piece and chains its calls ---------------------- */
typedef struct {
    int dev, ok;
    gab_parser *ps; gab_chain_packed pk; gab_chain *h;
    int32_t *d_score, *d_parent, *sc, *pa;
} cgp_part;
typedef struct { int ng; const char *whole; size_t cut[65]; cgp_part part[64]; } cgp_ctx;
static void cgp_parse(int g, void *v) {
    cgp_ctx *G = (cgp_ctx *)v;
    cgp_part *p = &G->part[g];
    p->dev = gab_phys_gpu(g);
    if (gab_parser_create(p->dev, &p->ps) != 0) return;
    if (gab_chain_parse(p->ps, G->whole + G->cut[g], (int64_t)(G->cut[g + 1] - G->cut[g]), &p->pk, NULL) != 0) return;
    if (gab_chain_create(p->dev, &p->h) != 0) return;
    const size_t t = (size_t)p->pk.total;
    if (gab_device_alloc(p->dev, 4 * t + 4, (void **)&p->d_score) != 0 || gab_device_alloc(p->dev, 4 * t + 4, (void **)&p->d_parent) != 0) return;
    p->sc = (int32_t *)malloc(4 * t + 4); p->pa = (int32_t *)malloc(4 * t + 4);
    if (!p->sc || !p->pa) return;
    gab_pin_out_on(p->dev, p->sc, 4 * t + 4); gab_pin_out_on(p->dev, p->pa, 4 * t + 4);
    if (gab_chain_reserve_mode(p->h, GAB_CHAIN_MODE, p->pk.total, p->pk.ncalls) != 0) return;   /* work tables, streams, first launches, fast-chain's table: before the region of interest */
    p->ok = 1;
}
static void cgp_run(int g, void *v) {
    cgp_part *p = &((cgp_ctx *)v)->part[g];
    if (p->pk.ncalls == 0) return;
    /* (the results reach the page-locked host arrays WHILE the DP runs: its kernel writes every block through) */
    GAB_DIE_IF(gab_chain_run_device_through(p->h, GAB_CHAIN_MODE, p->pk.d_x, p->pk.d_y, p->pk.call_off, p->pk.hdr, p->pk.ncalls, p->d_score, p->d_parent,
                                            p->sc, p->pa, NULL), "gab_chain_run_device_through");
}

static void help(void) {
    fprintf(stderr,
        "Usage: chain [OPTION]...\n"
        "Options:\n"
        "        -i <file>\n"
        "            input file\n"
        "        -o <file>\n"
        "            output file\n"
        "        -t <int>\n"
        "            number of CPU threads (ignored: the kernel runs on the GPU)\n"
        "        -g <int>\n"
        "            number of GPUs (default $GAB_GPUS or 1)\n"
        "        -h \n"
        "            prints the usage\n");
}

static void skip_to_EOR(FILE *fp) {
    const char *loc = "EOR";
    int ch;
    while (*loc != '\0' && (ch = fgetc(fp)) != EOF)
        if (ch == *loc) loc++;
}

int main(int argc, char **argv) {
    const char *in_name = "", *out_name = "";
    int opt, numThreads = 1, gpus = 0;
    while ((opt = getopt(argc, argv, ":i:o:t:g:h")) != -1) {
        switch (opt) {
            case 'i': in_name = optarg; break;
            case 'o': out_name = optarg; break;
            case 't': numThreads = atoi(optarg); break;
            case 'g': gpus = atoi(optarg); break;
            case 'h': help(); return 0;
            default: help(); return 1;
        }
    }
    if (argc == 1 || argc != optind) { help(); exit(EXIT_FAILURE); }
    fprintf(stderr, "Input file: %s\n", in_name);
    fprintf(stderr, "Output file: %s\n", out_name);
    FILE *in = fopen(in_name, "r"), *out = fopen(out_name, "w");
    if (!in || !out) { fprintf(stderr, "ERROR: cannot open %s\n", !in ? in_name : out_name); exit(EXIT_FAILURE); }

    /* GAB_GPU_PARSE=1: the file is read in one piece, cut behind "EOR" lines into one piece per GPU, and every GPU parses ITS piece
     * (gab_chain_parse); the anchors stay on the GPU that parsed them, the ROI is one gab_chain_run_device per
     * GPU.  Files that are not in the one-record-per-line layout are declined and take the fscanf path below. */
    const int64_t fsz = gab_regular_file_size(in);      /* -1 for pipes: they take the fscanf path */
    if (gab_gpu_parse_wanted(1) && fsz >= 0) {
        const int ng = gab_pick_gpus(gpus);
        char *whole = (char *)malloc((size_t)fsz + 1);
        cgp_ctx G;
        memset(&G, 0, sizeof G);
        G.ng = ng; G.whole = whole;
        int ok = whole && fread(whole, 1, (size_t)fsz, in) == (size_t)fsz && gab_cut_at_marker(whole, (size_t)fsz, ng, "EOR", 1, G.cut) == 0;
        if (ok) {
            gab_run_parts(ng, cgp_parse, &G);
            for (int g = 0; g < ng; g++) ok = ok && G.part[g].ok;
        }
        if (ok) {
            free(whole);
            fprintf(stderr, "Running with threads: %d (input parsed on the GPU)\n", numThreads);
            const double t0g = gab_now();
            gab_roi_begin_n(ng);
            gab_run_parts(ng, cgp_run, &G);
            gab_roi_end();
            const double rt = gab_now() - t0g;
            for (int g = 0; g < ng; g++) {
                const cgp_part *p = &G.part[g];
                for (int64_t c2 = 0; c2 < p->pk.ncalls; c2++) {
                    fprintf(out, "%lld\n", (long long)p->pk.hdr[c2].n);
                    const int64_t o = p->pk.call_off[c2];
                    for (int64_t i = 0; i < p->pk.hdr[c2].n; i++) fprintf(out, "%d\t%d\n", p->sc[o + i], p->pa[o + i]);
                    fprintf(out, "EOR\n");
                }
            }
            fprintf(stderr, "Time in kernel: %.2f sec\n", rt);
            if (gab_env_i64("GAB_ROI_PRECISE", 0)) fprintf(stderr, "[gab] region of interest: %.3f ms\n", rt * 1e3);
            if (getenv("GAB_QUEUE_REPORT")) {
                fprintf(stderr, "gab GPU parse: %d piece(s), calls per GPU:", ng);
                for (int g = 0; g < ng; g++) fprintf(stderr, " %ld", (long)G.part[g].pk.ncalls);
                fprintf(stderr, "\n");
            }
            fclose(in); fclose(out);
            for (int g = 0; g < ng; g++) {
                cgp_part *p = &G.part[g];
                gab_unpin(p->sc); gab_unpin(p->pa);
                gab_device_free(p->dev, p->d_score); gab_device_free(p->dev, p->d_parent); gab_chain_destroy(p->h); gab_parser_destroy(p->ps); free(p->sc); free(p->pa);
            }
            return 0;
        }
        if (getenv("GAB_GPU_PARSE")) fprintf(stderr, "GPU parser declined the file (%s); using the fscanf parser\n", gab_last_error());      /* (asked for by name: say so; the default falls back silently) */
        for (int g = 0; g < ng; g++) {
            cgp_part *p = &G.part[g];
            if (p->d_score) gab_device_free(p->dev, p->d_score);
            if (p->d_parent) gab_device_free(p->dev, p->d_parent);
            if (p->h) gab_chain_destroy(p->h);
            if (p->ps) gab_parser_destroy(p->ps);
            if (p->sc) { gab_unpin(p->sc); free(p->sc); }
            if (p->pa) { gab_unpin(p->pa); free(p->pa); }
        }
        free(whole);
        fseek(in, 0L, SEEK_SET);
    }

    /* read_call (host_data_io.cpp:13-51): 6 header fields, n x "x y", then everything up to "EOR" */
    size_t ccap = 1024, acap = 1 << 20, ncalls = 0, na = 0;
    gab_chain_hdr *hdr = (gab_chain_hdr *)malloc(ccap * sizeof(*hdr));
    int64_t *call_off = (int64_t *)malloc(ccap * 8);
    uint64_t *x = (uint64_t *)malloc(acap * 8), *y = (uint64_t *)malloc(acap * 8);
    for (;;) {
        long long n; float avg; int mdx, mdy, bw, nsegs;
        if (fscanf(in, "%lld%f%d%d%d%d", &n, &avg, &mdx, &mdy, &bw, &nsegs) != 6) break;
        if (ncalls == ccap) { ccap *= 2; hdr = (gab_chain_hdr *)realloc(hdr, ccap * sizeof(*hdr)); call_off = (int64_t *)realloc(call_off, ccap * 8); }
        while (na + (size_t)n > acap) { acap *= 2; x = (uint64_t *)realloc(x, acap * 8); y = (uint64_t *)realloc(y, acap * 8); }
        hdr[ncalls].n = n; hdr[ncalls].avg_qspan = avg; hdr[ncalls].max_dist_x = mdx; hdr[ncalls].max_dist_y = mdy;
        hdr[ncalls].bw = bw; hdr[ncalls].n_segs = nsegs;
        call_off[ncalls] = (int64_t)na;
        for (long long i = 0; i < n; i++) {
            unsigned long long xx = 0, yy = 0;
            if (fscanf(in, "%llu%llu", &xx, &yy) != 2) { xx = 0; yy = 0; }
            x[na + (size_t)i] = xx; y[na + (size_t)i] = yy;
        }
        na += (size_t)n;
        ncalls++;
        skip_to_EOR(in);
    }
    fprintf(stderr, "Running with threads: %d\n", numThreads);

    chain_ctx ctx;
    ctx.x = x; ctx.y = y; ctx.call_off = call_off; ctx.hdr = hdr; ctx.ncalls = (int64_t)ncalls;
    const int ngpus = gab_pick_gpus(gpus);
    ctx.shares = NULL;
    if (ngpus > 1 && gab_env_i64("GAB_CHUNK", 0) == 0 && ncalls > 0) {
        /* ---- N GPUs: one share per GPU, calls dealt longest first to the least-loaded GPU ---- */
        int64_t *order = (int64_t *)malloc(8 * ncalls), *load = (int64_t *)calloc((size_t)ngpus, 8);
        int *owner = (int *)malloc(sizeof(int) * ncalls);
        int64_t *local = (int64_t *)malloc(8 * ncalls);       /* index of a call inside its share */
        for (size_t c = 0; c < ncalls; c++) order[c] = (int64_t)c;
        qsort_r(order, ncalls, 8, by_n_desc, hdr);
        chain_share *sh = (chain_share *)calloc((size_t)ngpus, sizeof(chain_share));
        for (size_t k = 0; k < ncalls; k++) {
            int g = 0;
            for (int j = 1; j < ngpus; j++) if (load[j] < load[g]) g = j;
            owner[order[k]] = g; load[g] += hdr[order[k]].n; sh[g].ncalls++;
        }
        int64_t max_a = 0, max_c = 0;
        for (int g = 0; g < ngpus; g++) {
            sh[g].gpu = gab_phys_gpu(g); sh[g].src = &ctx;
            sh[g].call_id = (int64_t *)malloc(8 * (size_t)(sh[g].ncalls + 1)); sh[g].off = (int64_t *)malloc(8 * (size_t)(sh[g].ncalls + 1));
            sh[g].hdr = (gab_chain_hdr *)malloc(sizeof(gab_chain_hdr) * (size_t)(sh[g].ncalls + 1));
            sh[g].ncalls = 0; sh[g].na = 0;
        }
        for (size_t c = 0; c < ncalls; c++) {                  /* inside a share the calls keep the file's order */
            chain_share *s = &sh[owner[c]];
            local[c] = s->ncalls;
            s->call_id[s->ncalls] = (int64_t)c; s->off[s->ncalls] = s->na; s->hdr[s->ncalls] = hdr[c];
            s->ncalls++; s->na += hdr[c].n;
        }
        for (int g = 0; g < ngpus; g++) { if (sh[g].na > max_a) max_a = sh[g].na; if (sh[g].ncalls > max_c) max_c = sh[g].ncalls; }
        {
            pthread_t *th = (pthread_t *)calloc((size_t)ngpus, sizeof(pthread_t));
            for (int g = 0; g < ngpus; g++) pthread_create(&th[g], NULL, share_build, &sh[g]);
            for (int g = 0; g < ngpus; g++) pthread_join(th[g], NULL);
            free(th);
        }
        ctx.shares = sh; ctx.max_chunk_anchors = max_a; ctx.max_chunk_calls = max_c;
        if (getenv("GAB_QUEUE_REPORT")) {
            fprintf(stderr, "chain shares (calls/anchors/longest):");
            for (int g = 0; g < ngpus; g++) {
                int64_t mx = 0;
                for (int64_t k = 0; k < sh[g].ncalls; k++) if (sh[g].hdr[k].n > mx) mx = sh[g].hdr[k].n;
                fprintf(stderr, " %ld/%ld/%ld", (long)sh[g].ncalls, (long)sh[g].na, (long)mx);
            }
            fprintf(stderr, "\n");
        }
        gab_queue q;
        gab_queue_open(&q, ngpus, ngpus, gpu_init, run_share, gpu_fini, &ctx);
        /* ---- region of interest (main.cpp:111-193): one gab_chain_run per GPU ---- */
        const double t0 = gab_now();
        gab_roi_begin_n(ngpus);
        gab_queue_run_each(&q, ngpus);                          /* share g on GPU g */
        gab_roi_end();
        const double runtime = gab_now() - t0;
        gab_queue_close(&q);
        /* print_return (host_data_io.cpp:53-60), from the shares */
        for (size_t c = 0; c < ncalls; c++) {
            const chain_share *s = &sh[owner[c]];
            const int64_t o = s->off[local[c]];
            fprintf(out, "%lld\n", (long long)hdr[c].n);
            for (int64_t i = 0; i < hdr[c].n; i++) fprintf(out, "%d\t%d\n", s->score[o + i], s->parent[o + i]);
            fprintf(out, "EOR\n");
        }
        fprintf(stderr, "Time in kernel: %.2f sec\n", runtime);
        if (gab_env_i64("GAB_ROI_PRECISE", 0)) fprintf(stderr, "[gab] region of interest: %.3f ms\n", runtime * 1e3);
        for (int g = 0; g < ngpus; g++) {
            gab_unpin(sh[g].x); gab_unpin(sh[g].y); gab_unpin(sh[g].score); gab_unpin(sh[g].parent);
            free(sh[g].x); free(sh[g].y); free(sh[g].score); free(sh[g].parent); free(sh[g].call_id); free(sh[g].off); free(sh[g].hdr);
        }
        free(sh); free(order); free(load); free(owner); free(local);
        fclose(in); fclose(out);
        free(hdr); free(call_off); free(x); free(y);
        return 0;
    }
    ctx.score = (int32_t *)malloc(4 * (na + 1)); ctx.parent = (int32_t *)malloc(4 * (na + 1));
    /* one GPU: everything is ONE call (a chunk takes as long as its longest call, so fewer, larger chunks are better for
     * this kernel); $GAB_CHUNK anchors per chunk when set: file-order chunks of ~equal anchor count */
    const int64_t chunk_anchors = gab_env_i64("GAB_CHUNK", 0);
    int64_t nchunks_want = chunk_anchors > 0 ? (int64_t)(na / (size_t)chunk_anchors) + 1 : 1;
    if (nchunks_want > (int64_t)ncalls) nchunks_want = ncalls > 0 ? (int64_t)ncalls : 1;
    ctx.chunk_beg = (int64_t *)malloc(8 * (size_t)(nchunks_want + 2));
    int64_t nchunks = 0;
    {
        const size_t per = na / (size_t)nchunks_want + 1;
        size_t acc = 0;
        ctx.chunk_beg[0] = 0;
        for (size_t c = 0; c < ncalls; c++) {
            acc += (size_t)hdr[c].n;
            if (acc >= per && nchunks + 1 < nchunks_want) { ctx.chunk_beg[++nchunks] = (int64_t)c + 1; acc = 0; }
        }
        ctx.chunk_beg[++nchunks] = (int64_t)ncalls;
    }
    ctx.max_chunk_anchors = 0; ctx.max_chunk_calls = 0;
    for (int64_t c = 0; c < nchunks; c++) {
        const int64_t b = ctx.chunk_beg[c], e = ctx.chunk_beg[c + 1];
        if (e <= b) continue;
        const int64_t a = call_off[e - 1] + hdr[e - 1].n - call_off[b];
        if (a > ctx.max_chunk_anchors) ctx.max_chunk_anchors = a;
        if (e - b > ctx.max_chunk_calls) ctx.max_chunk_calls = e - b;
    }
    gab_pin(x, 8 * na); gab_pin(y, 8 * na); gab_pin_out(ctx.score, 4 * na); gab_pin_out(ctx.parent, 4 * na);      /* (warm copy from the first GPU picked) */
    gab_queue q;
    gab_queue_open(&q, ngpus, nchunks, gpu_init, run_chunk, gpu_fini, &ctx);

    /* ---- region of interest (main.cpp:111-193) ---- */
    const double t0 = gab_now();
    gab_roi_begin_n(ngpus);
    gab_queue_run(&q, nchunks);
    gab_roi_end();
    const double runtime = gab_now() - t0;
    gab_queue_close(&q);
    gab_unpin(x); gab_unpin(y); gab_unpin(ctx.score); gab_unpin(ctx.parent);

    /* print_return (host_data_io.cpp:53-60) */
    for (size_t c = 0; c < ncalls; c++) {
        fprintf(out, "%lld\n", (long long)hdr[c].n);
        const int64_t o = call_off[c];
        for (int64_t i = 0; i < hdr[c].n; i++) fprintf(out, "%d\t%d\n", ctx.score[o + i], ctx.parent[o + i]);
        fprintf(out, "EOR\n");
    }
    fprintf(stderr, "Time in kernel: %.2f sec\n", runtime);
    if (gab_env_i64("GAB_ROI_PRECISE", 0)) fprintf(stderr, "[gab] region of interest: %.3f ms\n", runtime * 1e3);   /* (the reference prints two decimals of a second) */
    fclose(in); fclose(out);
    free(hdr); free(call_off); free(x); free(y); free(ctx.score); free(ctx.parent); free(ctx.chunk_beg);
    return 0;
}
