/* Seeded synthetic-input generators (see gabgen.h).  Test/bench infrastructure. */
#include "gabgen.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ---------------------------------------------------------------- rng ---- */
typedef struct { uint64_t s; } rng_t;
static inline uint64_t sm64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline rng_t rng_for(uint64_t seed, uint64_t stream, uint64_t item) {
    uint64_t s = seed * 0xD1342543DE82EF95ull + stream * 0x2545F4914F6CDD1Dull;
    (void)sm64(&s);
    s ^= item * 0x9E3779B97F4A7C15ull;
    rng_t r; r.s = sm64(&s); return r;
}
static inline uint64_t rnd(rng_t *r) { return sm64(&r->s); }
/* uniform integer in [lo, hi] */
static inline int64_t rnd_range(rng_t *r, int64_t lo, int64_t hi) {
    return lo + (int64_t)(rnd(r) % (uint64_t)(hi - lo + 1));
}
/* true with probability num/10000 */
static inline int rnd_pm(rng_t *r, int num) { return (int)(rnd(r) % 10000u) < num; }

/* ================================================================= bsw === */
/* Generates one pair into q[] / t[] (capacity 256 / 2048); returns lengths. */
static void bsw_item(uint64_t seed, int mode, int64_t idx, uint8_t *q, int *qlen_out,
                     uint8_t *t, int *tlen_out, int *h0_out) {
    rng_t r = rng_for(seed, 1, (uint64_t)idx);
    int qlen, tlen = 0, h0;
    if (mode == 0) {
        qlen = (int)rnd_range(&r, 20, 131);
        h0 = (int)rnd_range(&r, 19, 100);
        for (int i = 0; i < qlen; i++) q[i] = rnd_pm(&r, 100) ? 4 : (uint8_t)(rnd(&r) & 3);
        for (int i = 0; i < qlen; i++) {
            if (rnd_pm(&r, 100)) continue;                       /* deletion   */
            if (rnd_pm(&r, 100)) t[tlen++] = (uint8_t)(rnd(&r) & 3); /* insertion */
            uint8_t c = q[i];
            if (rnd_pm(&r, 500)) c = (uint8_t)(rnd(&r) & 3);     /* substitution */
            t[tlen++] = c;
        }
        int tail = (int)rnd_range(&r, 0, 100);
        for (int i = 0; i < tail; i++) t[tlen++] = rnd_pm(&r, 100) ? 4 : (uint8_t)(rnd(&r) & 3);
        if (tlen == 0) t[tlen++] = 0;
    } else {
        static const int h0s[7] = {0, 1, 5, 19, 50, 200, 1000};
        qlen = (int)rnd_range(&r, 1, 250);
        h0 = h0s[rnd(&r) % 7];
        for (int i = 0; i < qlen; i++) q[i] = rnd_pm(&r, 300) ? 4 : (uint8_t)(rnd(&r) & 3);
        int i = 0;
        while (i < qlen && tlen < 540) {
            if (rnd_pm(&r, 150)) { i += (int)rnd_range(&r, 1, 12); continue; }      /* del burst */
            if (rnd_pm(&r, 150)) {
                int b = (int)rnd_range(&r, 1, 12);
                for (int k = 0; k < b && tlen < 540; k++) t[tlen++] = (uint8_t)(rnd(&r) & 3);
            }
            uint8_t c = q[i++];
            if (rnd_pm(&r, 800)) c = (uint8_t)(rnd(&r) & 3);
            if (rnd_pm(&r, 300)) c = 4;
            if (tlen < 540) t[tlen++] = c;
        }
        /* one in eight pairs: unrelated reference (exercises m==0 / z-drop exits) */
        if ((rnd(&r) & 7) == 0) { tlen = 0; }
        int tail = (int)rnd_range(&r, 0, 300);
        if (tlen + tail > 549) tail = 549 - tlen;
        for (int k = 0; k < tail; k++) t[tlen++] = rnd_pm(&r, 300) ? 4 : (uint8_t)(rnd(&r) & 3);
        if (tlen == 0) t[tlen++] = (uint8_t)(rnd(&r) & 3);
    }
    *qlen_out = qlen; *tlen_out = tlen; *h0_out = h0;
}

void gab_gen_bsw_lens(uint64_t seed, int mode, int64_t first, int64_t n,
                      int32_t *len1, int32_t *len2, int32_t *h0) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        uint8_t q[256], t[2048]; int ql, tl, h;
        bsw_item(seed, mode, first + i, q, &ql, t, &tl, &h);
        len1[i] = tl; len2[i] = ql; h0[i] = h;
    }
}

void gab_gen_bsw_fill(uint64_t seed, int mode, int64_t first, int64_t n,
                      uint8_t *ref, const int64_t *ref_off,
                      uint8_t *qry, const int64_t *qry_off) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        uint8_t q[256], t[2048]; int ql, tl, h;
        bsw_item(seed, mode, first + i, q, &ql, t, &tl, &h);
        memcpy(ref + ref_off[i], t, (size_t)tl);
        memcpy(qry + qry_off[i], q, (size_t)ql);
    }
}

int gab_gen_bsw_write(const char *path, uint64_t seed, int mode, int64_t n) {
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    char line[2100];
    for (int64_t i = 0; i < n; i++) {
        uint8_t q[256], t[2048]; int ql, tl, h;
        bsw_item(seed, mode, i, q, &ql, t, &tl, &h);
        fprintf(f, "%d\n", h);
        for (int k = 0; k < tl; k++) line[k] = (char)('0' + t[k]);
        line[tl] = '\n'; fwrite(line, 1, (size_t)tl + 1, f);
        for (int k = 0; k < ql; k++) line[k] = (char)('0' + q[k]);
        line[ql] = '\n'; fwrite(line, 1, (size_t)ql + 1, f);
    }
    return fclose(f);
}

/* ============================================================ bpm / wfa === */
#define PAIRS_MAXLEN 4096
static void pairs_item(uint64_t seed, int mode, int plen, int64_t idx,
                       char *p, int *pl_out, char *t, int *tl_out) {
    static const char B[4] = {'A', 'C', 'G', 'T'};
    rng_t r = rng_for(seed, 2, (uint64_t)idx);
    int pl = plen, tl = 0;
    int pN = 10, pS = 200, pI = 50, pD = 50, pLow = 0;
    if (mode == 1) {
        pl = (int)rnd_range(&r, 1, plen);
        pN = 200; pS = 500; pI = 200; pD = 200; pLow = 30;
    }
    for (int i = 0; i < pl; i++) {
        char c = B[rnd(&r) & 3];
        if (rnd_pm(&r, pN)) c = 'N';
        if (pLow && rnd_pm(&r, pLow)) c = (char)(c | 0x20);
        p[i] = c;
    }
    for (int i = 0; i < pl && tl < PAIRS_MAXLEN - 2; i++) {
        if (rnd_pm(&r, pD)) continue;
        if (rnd_pm(&r, pI)) t[tl++] = B[rnd(&r) & 3];
        char c = p[i];
        if (rnd_pm(&r, pS)) c = B[rnd(&r) & 3];
        t[tl++] = c;
    }
    if (mode == 1 && (rnd(&r) & 15) == 0) {          /* unrelated text, other length */
        tl = (int)rnd_range(&r, 1, plen);
        for (int i = 0; i < tl; i++) t[i] = B[rnd(&r) & 3];
    }
    if (tl == 0) t[tl++] = B[rnd(&r) & 3];
    *pl_out = pl; *tl_out = tl;
}

void gab_gen_pairs_lens(uint64_t seed, int mode, int plen, int64_t first, int64_t n,
                        int32_t *pat_len, int32_t *txt_len) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        char p[PAIRS_MAXLEN], t[PAIRS_MAXLEN]; int pl, tl;
        pairs_item(seed, mode, plen, first + i, p, &pl, t, &tl);
        pat_len[i] = pl; txt_len[i] = tl;
    }
}

void gab_gen_pairs_fill(uint64_t seed, int mode, int plen, int64_t first, int64_t n,
                        char *pat, const int64_t *pat_off,
                        char *txt, const int64_t *txt_off) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        char p[PAIRS_MAXLEN], t[PAIRS_MAXLEN]; int pl, tl;
        pairs_item(seed, mode, plen, first + i, p, &pl, t, &tl);
        memcpy(pat + pat_off[i], p, (size_t)pl);
        memcpy(txt + txt_off[i], t, (size_t)tl);
    }
}

int gab_gen_pairs_write(const char *path, uint64_t seed, int mode, int plen, int64_t n) {
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    for (int64_t i = 0; i < n; i++) {
        char p[PAIRS_MAXLEN], t[PAIRS_MAXLEN]; int pl, tl;
        pairs_item(seed, mode, plen, i, p, &pl, t, &tl);
        fputc('>', f); fwrite(p, 1, (size_t)pl, f); fputc('\n', f);
        fputc('<', f); fwrite(t, 1, (size_t)tl, f); fputc('\n', f);
    }
    return fclose(f);
}

/* ================================================================ chain === */
/* Anchor (minimap2 mm128_t):  x = rid << 32 | ref_pos  (sorted ascending),
 * y = seg_id << 48 | q_span << 32 | query_pos   (chain/src/host_kernel.cpp:27-28,53-55). */
static int64_t chain_n(uint64_t seed, int64_t nmin, int64_t nmax, int64_t idx) {
    rng_t r = rng_for(seed, 3, (uint64_t)idx);
    double u = (double)(rnd(&r) >> 11) / 9007199254740992.0;
    double v = exp(log((double)nmin) + u * (log((double)nmax) - log((double)nmin)));
    int64_t n = (int64_t)(v + 0.5);
    if (n < nmin) n = nmin;
    if (n > nmax) n = nmax;
    return n;
}

typedef struct { uint64_t x, y; } gen_anchor;
static int cmp_anchor(const void *a, const void *b) {
    const gen_anchor *p = (const gen_anchor *)a, *q = (const gen_anchor *)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    return 0;
}

/* fills anchors[n]; returns the mean q_span */
static float chain_item(uint64_t seed, int mode, int64_t idx, int64_t n, gen_anchor *a) {
    rng_t r = rng_for(seed, 4, (uint64_t)idx);
    double span_sum = 0;
    if (mode == 0) {
        /* reference extent grows with n so anchor density (and the window) stays realistic */
        int64_t extent = n * (int64_t)rnd_range(&r, 8, 40) + 2000;
        int ndiag = (int)rnd_range(&r, 1, 5);
        int64_t dq0[5], dr0[5];
        for (int d = 0; d < ndiag; d++) { dr0[d] = rnd_range(&r, 0, extent / 4); dq0[d] = rnd_range(&r, 0, extent / 4); }
        for (int64_t i = 0; i < n; i++) {
            int64_t rp, qp;
            if (rnd_pm(&r, 7000)) {
                int d = (int)(rnd(&r) % (uint64_t)ndiag);
                int64_t t = rnd_range(&r, 0, extent);
                rp = dr0[d] + t; qp = dq0[d] + t + rnd_range(&r, -40, 40);
                if (qp < 0) qp = 0;
            } else { rp = rnd_range(&r, 0, extent * 5 / 4); qp = rnd_range(&r, 0, extent * 5 / 4); }
            int span = rnd_pm(&r, 9000) ? 15 : (int)rnd_range(&r, 1, 30);
            span_sum += span;
            a[i].x = (uint64_t)rp;
            a[i].y = ((uint64_t)span << 32) | (uint64_t)(qp & 0x7fffffff);
        }
    } else {
        /* dense: tiny coordinate range -> windows of thousands of predecessors, many ties,
         * multiple reference ids (x above 2^32), multi-segment ids, duplicate positions */
        int64_t extent = (int64_t)rnd_range(&r, 300, 6000);
        int nrid = (int)rnd_range(&r, 1, 3);
        int nseg = (idx % 3 == 0) ? 2 : 1;
        for (int64_t i = 0; i < n; i++) {
            uint64_t rid = rnd(&r) % (uint64_t)nrid;
            int64_t t = rnd_range(&r, 0, extent);
            int64_t rp = t, qp = t + rnd_range(&r, -25, 25);
            if (rnd_pm(&r, 2000)) qp = rnd_range(&r, 0, extent);
            if (qp < 0) qp = 0;
            int span = rnd_pm(&r, 8000) ? 15 : (int)rnd_range(&r, 1, 40);
            uint64_t seg = (uint64_t)(rnd(&r) % (uint64_t)nseg);
            span_sum += span;
            a[i].x = (rid * 5 + 1) << 32 | (uint64_t)rp;   /* rid 1,6,11: above 2^32 */
            a[i].y = seg << 48 | ((uint64_t)span << 32) | (uint64_t)qp;
        }
    }
    qsort(a, (size_t)n, sizeof(gen_anchor), cmp_anchor);
    return (float)(span_sum / (double)n);
}

void gab_gen_chain_hdrs(uint64_t seed, int mode, int64_t nmin, int64_t nmax,
                        int64_t first, int64_t ncalls, gabgen_chain_hdr *hdr) {
    for (int64_t c = 0; c < ncalls; c++) {
        int64_t idx = first + c;
        int64_t n = chain_n(seed, nmin, nmax, idx);
        hdr[c].n = n;
        hdr[c].max_dist_x = 5000; hdr[c].max_dist_y = 5000; hdr[c].bw = 500;
        hdr[c].n_segs = (mode == 1 && idx % 3 == 0) ? 2 : 1;
        hdr[c].avg_qspan = 0.f;   /* filled by gab_gen_chain_fill */
    }
}

void gab_gen_chain_fill(uint64_t seed, int mode, int64_t nmin, int64_t nmax,
                        int64_t first, int64_t ncalls, const gabgen_chain_hdr *hdr_in,
                        const int64_t *call_off, uint64_t *x, uint64_t *y) {
    (void)nmin; (void)nmax;
    gabgen_chain_hdr *hdr = (gabgen_chain_hdr *)hdr_in;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t c = 0; c < ncalls; c++) {
        int64_t n = hdr[c].n;
        gen_anchor *a = (gen_anchor *)malloc(sizeof(gen_anchor) * (size_t)n);
        float avg = chain_item(seed, mode, first + c, n, a);
        /* mode 1 uses integer avg_qspan on two thirds of the calls: separates the fp32
         * and fp64 gap-cost paths of fast-chain */
        if (mode == 1 && (first + c) % 3 != 2) avg = (float)(int)(avg + 0.5f);
        /* the text format prints avg_qspan with %.6g-like precision; round-trip it */
        char buf[64]; snprintf(buf, sizeof buf, "%f", avg); avg = strtof(buf, 0);
        hdr[c].avg_qspan = avg;
        for (int64_t i = 0; i < n; i++) { x[call_off[c] + i] = a[i].x; y[call_off[c] + i] = a[i].y; }
        free(a);
    }
}

/* the same for an arbitrary list of call ids (bench.py's strong-scaling deal hands a rank a scattered subset) */
void gab_gen_chain_ids(uint64_t seed, int mode, int64_t nmin, int64_t nmax, const int64_t *ids, int64_t ncalls,
                       gabgen_chain_hdr *hdr) {
    for (int64_t c = 0; c < ncalls; c++) gab_gen_chain_hdrs(seed, mode, nmin, nmax, ids[c], 1, &hdr[c]);
}
void gab_gen_chain_fill_ids(uint64_t seed, int mode, const int64_t *ids, int64_t ncalls, gabgen_chain_hdr *hdr,
                            const int64_t *call_off, uint64_t *x, uint64_t *y) {
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t c = 0; c < ncalls; c++) {
        int64_t n = hdr[c].n;
        gen_anchor *a = (gen_anchor *)malloc(sizeof(gen_anchor) * (size_t)n);
        float avg = chain_item(seed, mode, ids[c], n, a);
        if (mode == 1 && ids[c] % 3 != 2) avg = (float)(int)(avg + 0.5f);
        char buf[64]; snprintf(buf, sizeof buf, "%f", avg); avg = strtof(buf, 0);
        hdr[c].avg_qspan = avg;
        for (int64_t i = 0; i < n; i++) { x[call_off[c] + i] = a[i].x; y[call_off[c] + i] = a[i].y; }
        free(a);
    }
}

/* the pairs of a batch laid out as the drivers' input file has them: pattern i directly followed by text i (plus `gap`
 * bytes where the file has "\n<" resp. "\n>"); out_poff / out_toff receive the new offsets.  Returns the bytes used. */
int64_t gab_gen_interleave(const uint8_t *pat, const int64_t *pat_off, const int32_t *pat_len, const uint8_t *txt,
                           const int64_t *txt_off, const int32_t *txt_len, int64_t n, int gap, uint8_t *out,
                           int64_t *out_poff, int64_t *out_toff) {
    int64_t at = gap;
    for (int64_t i = 0; i < n; i++) {
        out_poff[i] = at; at += pat_len[i] + gap;
        out_toff[i] = at; at += txt_len[i] + gap;
    }
    if (!out) return at;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        memcpy(out + out_poff[i], pat + pat_off[i], (size_t)pat_len[i]);
        memcpy(out + out_toff[i], txt + txt_off[i], (size_t)txt_len[i]);
    }
    return at;
}

/* decimal text of v into p, returns the end (fprintf per anchor made the 85 M-anchor file take minutes) */
static char *put_u64(char *p, uint64_t v) {
    char t[24]; int k = 0;
    do { t[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) *p++ = t[--k];
    return p;
}
int gab_gen_chain_write(const char *path, uint64_t seed, int mode, int64_t nmin,
                        int64_t nmax, int64_t ncalls) {
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    const size_t cap = (size_t)1 << 22;
    char *buf = (char *)malloc(cap + 64);
    if (!buf) { fclose(f); return -1; }
    for (int64_t c = 0; c < ncalls; c++) {
        gabgen_chain_hdr h;
        gab_gen_chain_hdrs(seed, mode, nmin, nmax, c, 1, &h);
        /* one call, generated on this thread (gab_gen_chain_fill would start an OpenMP team per call) */
        gen_anchor *a = (gen_anchor *)malloc(sizeof(gen_anchor) * (size_t)h.n);
        float avg = chain_item(seed, mode, c, h.n, a);
        if (mode == 1 && c % 3 != 2) avg = (float)(int)(avg + 0.5f);
        char num[64]; snprintf(num, sizeof num, "%f", avg); avg = strtof(num, 0);
        h.avg_qspan = avg;
        fprintf(f, "%lld\t%f\t%d\t%d\t%d\t%d\n", (long long)h.n, h.avg_qspan, h.max_dist_x,
                h.max_dist_y, h.bw, h.n_segs);
        char *p = buf;
        for (int64_t i = 0; i < h.n; i++) {
            p = put_u64(p, a[i].x); *p++ = '\t'; p = put_u64(p, a[i].y); *p++ = '\n';
            if ((size_t)(p - buf) > cap - 64) { fwrite(buf, 1, (size_t)(p - buf), f); p = buf; }
        }
        fwrite(buf, 1, (size_t)(p - buf), f);
        fprintf(f, "EOR\n");
        free(a);
    }
    free(buf);
    return fclose(f);
}

/* ================================================================== fmi === */
void gab_gen_fmi_ref(uint64_t seed, int64_t ref_len, int rep_pct, uint8_t *ref) {
    /* random bases in blocks (one rng stream per 64 Ki bases -> parallel, order independent) */
    const int64_t B = 65536;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (ref_len + B - 1) / B; b++) {
        rng_t r = rng_for(seed, 5, (uint64_t)b);
        int64_t end = (b + 1) * B < ref_len ? (b + 1) * B : ref_len;
        for (int64_t i = b * B; i < end; i += 32) {
            uint64_t w = rnd(&r);
            for (int k = 0; k < 32 && i + k < end; k++, w >>= 2) ref[i + k] = (uint8_t)(w & 3);
        }
    }
    /* planted repeats: a family of 16 templates of 300 bp copied (with 1% divergence) over rep_pct % */
    if (rep_pct > 0 && ref_len > 4000) {
        rng_t r = rng_for(seed, 6, 0);
        uint8_t tmpl[16][300];
        for (int f = 0; f < 16; f++) for (int i = 0; i < 300; i++) tmpl[f][i] = (uint8_t)(rnd(&r) & 3);
        int64_t copies = ref_len / 100 * rep_pct / 300;
        for (int64_t c = 0; c < copies; c++) {
            int f = (int)(rnd(&r) & 15);
            int64_t pos = rnd_range(&r, 0, ref_len - 301);
            for (int i = 0; i < 300; i++) ref[pos + i] = rnd_pm(&r, 100) ? (uint8_t)(rnd(&r) & 3) : tmpl[f][i];
        }
    }
}

void gab_gen_fmi_reads(uint64_t seed, const uint8_t *ref, int64_t ref_len, int rl_min, int rl_max,
                       int64_t first, int64_t n, uint8_t *enc, int32_t stride, int32_t *len) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        rng_t r = rng_for(seed, 7, (uint64_t)(first + i));
        int L = (int)rnd_range(&r, rl_min, rl_max);
        if (L > ref_len) L = (int)ref_len;
        int64_t pos = rnd_range(&r, 0, ref_len - L);
        int rev = (int)(rnd(&r) & 1);
        uint8_t *o = enc + (int64_t)i * stride;
        for (int k = 0; k < L; k++) {
            uint8_t c = rev ? (uint8_t)(3 - ref[pos + L - 1 - k]) : ref[pos + k];
            if (rnd_pm(&r, 200)) c = (uint8_t)(rnd(&r) & 3);
            if (rnd_pm(&r, 20)) c = 4;
            o[k] = c;
        }
        for (int k = L; k < stride; k++) o[k] = 4;
        len[i] = L;
    }
}

int gab_gen_fmi_write_fasta(const char *path, const uint8_t *ref, int64_t ref_len) {
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    fprintf(f, ">synthetic\n");
    char line[81];
    for (int64_t i = 0; i < ref_len; i += 80) {
        int k = 0;
        for (; k < 80 && i + k < ref_len; k++) line[k] = "ACGT"[ref[i + k] & 3];
        line[k] = '\n';
        fwrite(line, 1, (size_t)k + 1, f);
    }
    return fclose(f);
}

int gab_gen_fmi_write_fastq(const char *path, const uint8_t *enc, int32_t stride, const int32_t *len, int64_t n) {
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    char *s = (char *)malloc((size_t)stride + 2), *q = (char *)malloc((size_t)stride + 2);
    for (int64_t i = 0; i < n; i++) {
        int L = len[i];
        for (int k = 0; k < L; k++) { s[k] = "ACGTN"[enc[i * stride + k] > 4 ? 4 : enc[i * stride + k]]; q[k] = 'I'; }
        s[L] = q[L] = 0;
        fprintf(f, "@r%lld\n%s\n+\n%s\n", (long long)i, s, q);
    }
    free(s); free(q);
    return fclose(f);
}
