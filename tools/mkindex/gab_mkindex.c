/* See gab_mkindex.h. */
#include "gab_mkindex.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ SA-IS (induced sorting) ----
 * Nong, Zhang, Chan: "Two efficient algorithms for linear time suffix array construction".
 * Own implementation; the type array is a bit set, buckets are recomputed per pass. */
typedef struct { const void *s; int cs; } sstr;   /* cs = 1: bytes, 4: int32 */
static inline int32_t chr_at(const sstr *t, int32_t i) {
    return t->cs == 1 ? (int32_t)((const uint8_t *)t->s)[i] : ((const int32_t *)t->s)[i];
}
#define TGET(i) ((tbits[(i) >> 3] >> ((i) & 7)) & 1)            /* 1 = S-type, 0 = L-type */
#define TSET(i, b) do { if (b) tbits[(i) >> 3] |= (uint8_t)(1u << ((i) & 7)); else tbits[(i) >> 3] &= (uint8_t)~(1u << ((i) & 7)); } while (0)
#define IS_LMS(i) ((i) > 0 && TGET(i) && !TGET((i) - 1))

static void bucket_bounds(const sstr *t, int32_t *bkt, int32_t n, int32_t K, int ends) {
    for (int32_t c = 0; c < K; c++) bkt[c] = 0;
    for (int32_t i = 0; i < n; i++) bkt[chr_at(t, i)]++;
    int32_t sum = 0;
    for (int32_t c = 0; c < K; c++) { sum += bkt[c]; bkt[c] = ends ? sum : sum - bkt[c]; }
}
static void induce_L(const sstr *t, const uint8_t *tbits, int32_t *SA, int32_t *bkt, int32_t n, int32_t K) {
    bucket_bounds(t, bkt, n, K, 0);
    for (int32_t i = 0; i < n; i++) {
        int32_t j = SA[i] - 1;
        if (j >= 0 && !TGET(j)) SA[bkt[chr_at(t, j)]++] = j;
    }
}
static void induce_S(const sstr *t, const uint8_t *tbits, int32_t *SA, int32_t *bkt, int32_t n, int32_t K) {
    bucket_bounds(t, bkt, n, K, 1);
    for (int32_t i = n - 1; i >= 0; i--) {
        int32_t j = SA[i] - 1;
        if (j >= 0 && TGET(j)) SA[--bkt[chr_at(t, j)]] = j;
    }
}

static int sais_core(const sstr *t, int32_t *SA, int32_t n, int32_t K) {
    if (n == 1) { SA[0] = 0; return 0; }
    uint8_t *tbits = (uint8_t *)calloc((size_t)n / 8 + 1, 1);
    int32_t *bkt = (int32_t *)malloc(sizeof(int32_t) * (size_t)K);
    if (!tbits || !bkt) { free(tbits); free(bkt); return -1; }
    /* classify: last char (sentinel) is S, the one before it L */
    TSET(n - 1, 1); TSET(n - 2, 0);
    for (int32_t i = n - 3; i >= 0; i--) {
        int32_t a = chr_at(t, i), b = chr_at(t, i + 1);
        TSET(i, (a < b || (a == b && TGET(i + 1))) ? 1 : 0);
    }
    /* stage 1: sort all LMS substrings */
    bucket_bounds(t, bkt, n, K, 1);
    for (int32_t i = 0; i < n; i++) SA[i] = -1;
    for (int32_t i = 1; i < n; i++) if (IS_LMS(i)) SA[--bkt[chr_at(t, i)]] = i;
    induce_L(t, tbits, SA, bkt, n, K);
    induce_S(t, tbits, SA, bkt, n, K);
    /* compact the sorted LMS substrings into SA[0..n1) */
    int32_t n1 = 0;
    for (int32_t i = 0; i < n; i++) if (IS_LMS(SA[i])) SA[n1++] = SA[i];
    for (int32_t i = n1; i < n; i++) SA[i] = -1;
    /* name them */
    int32_t name = 0, prev = -1;
    for (int32_t i = 0; i < n1; i++) {
        int32_t pos = SA[i];
        int diff = 0;
        if (prev < 0) diff = 1;
        else {
            for (int32_t d = 0; d < n; d++) {
                if (chr_at(t, pos + d) != chr_at(t, prev + d) || TGET(pos + d) != TGET(prev + d)) { diff = 1; break; }
                if (d > 0 && (IS_LMS(pos + d) || IS_LMS(prev + d))) break;
            }
        }
        if (diff) { name++; prev = pos; }
        SA[n1 + pos / 2] = name - 1;
    }
    for (int32_t i = n - 1, j = n - 1; i >= n1; i--) if (SA[i] >= 0) SA[j--] = SA[i];
    /* stage 2: solve the reduced problem */
    int32_t *SA1 = SA, *s1 = SA + n - n1;
    int rc = 0;
    if (name < n1) {
        sstr t1 = {s1, 4};
        rc = sais_core(&t1, SA1, n1, name);
    } else {
        for (int32_t i = 0; i < n1; i++) SA1[s1[i]] = i;
    }
    if (rc) { free(tbits); free(bkt); return rc; }
    /* stage 3: induce the final order */
    bucket_bounds(t, bkt, n, K, 1);
    for (int32_t i = 1, j = 0; i < n; i++) if (IS_LMS(i)) s1[j++] = i;   /* positions of LMS suffixes */
    for (int32_t i = 0; i < n1; i++) SA1[i] = s1[SA1[i]];
    for (int32_t i = n1; i < n; i++) SA[i] = -1;
    for (int32_t i = n1 - 1; i >= 0; i--) {
        int32_t j = SA[i]; SA[i] = -1;
        SA[--bkt[chr_at(t, j)]] = j;
    }
    induce_L(t, tbits, SA, bkt, n, K);
    induce_S(t, tbits, SA, bkt, n, K);
    free(tbits); free(bkt);
    return 0;
}

int gab_sais_i32(const int32_t *s, int32_t *SA, int32_t n, int32_t K) { sstr t = {s, 4}; return n <= 0 ? -1 : sais_core(&t, SA, n, K); }
int gab_sais_u8(const uint8_t *s, int32_t *SA, int32_t n, int32_t K) { sstr t = {s, 1}; return n <= 0 ? -1 : sais_core(&t, SA, n, K); }

/* ------------------------------------------------------------------ index ---------------------- */
int gab_mkindex_build(const uint8_t *fwd, int64_t L, gab_fmindex *out) {
    memset(out, 0, sizeof(*out));
    if (L <= 0 || 2 * L + 1 >= 0x7fffffffll) return -1;
    const int64_t n2 = 2 * L;                    /* |T| = forward + reverse complement */
    const int32_t n = (int32_t)(n2 + 1);         /* + sentinel */
    uint8_t *T = (uint8_t *)malloc((size_t)n);
    int32_t *SA = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!T || !SA) { free(T); free(SA); return -2; }
    for (int64_t i = 0; i < L; i++) {
        if (fwd[i] > 3) { free(T); free(SA); return -3; }     /* must be N-free (bntseq.cpp:266-284 randomises N) */
        T[i] = (uint8_t)(fwd[i] + 1);
        T[n2 - 1 - i] = (uint8_t)(3 - fwd[i] + 1);
    }
    T[n2] = 0;
    if (gab_sais_u8(T, SA, n, 5)) { free(T); free(SA); return -4; }
    /* SA[0] == n2 (the empty suffix first), exactly the reference's suffix_array[0] = pac_len */
    out->ref_seq_len = n;
    int64_t cnt[4] = {0, 0, 0, 0};
    for (int64_t i = 0; i < L; i++) { cnt[fwd[i]]++; cnt[3 - fwd[i]]++; }
    out->count[0] = 0; out->count[1] = cnt[0]; out->count[2] = cnt[0] + cnt[1];
    out->count[3] = cnt[0] + cnt[1] + cnt[2]; out->count[4] = n2;
    out->cp_occ_size = ((int64_t)n >> 6) + 1;
    out->cp_occ = (gab_cp_occ *)calloc((size_t)out->cp_occ_size, sizeof(gab_cp_occ));
    out->n_sa = ((int64_t)n >> 3) + 1;
    out->sa_ms_byte = (int8_t *)calloc((size_t)out->n_sa, 1);
    out->sa_ls_word = (uint32_t *)calloc((size_t)out->n_sa, sizeof(uint32_t));
    if (!out->cp_occ || !out->sa_ms_byte || !out->sa_ls_word) { free(T); free(SA); gab_mkindex_free(out); return -2; }
    int64_t run[4] = {0, 0, 0, 0};
    out->sentinel_index = -1;
    for (int64_t i = 0; i < n; i++) {
        if ((i & 63) == 0) {
            gab_cp_occ *e = &out->cp_occ[i >> 6];
            for (int c = 0; c < 4; c++) e->cp_count[c] = run[c];
        }
        int c;
        if (SA[i] == 0) { c = 4; out->sentinel_index = i; }
        else c = T[SA[i] - 1] - 1;
        if (c < 4) {
            out->cp_occ[i >> 6].one_hot_bwt_str[c] |= 1ull << (63 - (i & 63));    /* MSB first */
            run[c]++;
        }
        if ((i & 7) == 0) { out->sa_ls_word[i >> 3] = (uint32_t)SA[i]; out->sa_ms_byte[i >> 3] = 0; }
    }
    free(T); free(SA);
    return 0;
}

/* ------------------------------------------------------------------ periodic text, any size -----
 * Index of the reference W^m with W = U . revcomp(U) (its own reverse complement), built WITHOUT sorting the whole text:
 * BWA-MEM2 indexes T = reference + its reverse complement = W^k, k = 2m, w = |W|.  A suffix of T$ is W[r..) W^e $ with e
 * full copies behind it.  Two such strings are ordered within their first 2w characters unless they are the same rotation
 * r, and then the one with fewer copies (the $ comes sooner) is smaller.  Hence the suffix array of X = W^3 $ (e = 0, 1, 2)
 * already is the suffix array of T$: every e = 2 entry of it stands for the run e = 2, 3, ..., k-1 in this order (all of
 * them share >= 2w + 1 characters, enough to decide every comparison with an e <= 1 suffix), and the e = 0, 1 entries stay
 * single rows.  Needs W primitive (checked) and k >= 3.  Rows, counts, the sampled suffix array and the sentinel row are
 * produced in one streaming pass, so a >= 2^32-row index (what a human genome has: the 40-bit interval arithmetic of the
 * seeding kernel) costs seconds.  tests/test_mkindex.py checks the result against gab_mkindex_build on small cases. */
static int is_primitive(const uint8_t *W, int64_t w) {
    int64_t *fail = (int64_t *)malloc(sizeof(int64_t) * (size_t)(w + 1));
    if (!fail) return -1;
    fail[0] = -1;
    int64_t kk = -1;
    for (int64_t i = 0; i < w; i++) {
        while (kk >= 0 && W[kk] != W[i]) kk = fail[kk];
        fail[i + 1] = ++kk;
    }
    const int64_t period = w - fail[w];
    free(fail);
    return !(period < w && w % period == 0);
}
typedef struct { gab_fmindex *o; int64_t row, run[4]; } emitter;
static void emit_rows(emitter *E, int c, int64_t len, int64_t sa0, int64_t sa_step) {
    /* `len` consecutive rows with BWT symbol c (0..3; 4 = the sentinel row, no bit set); row j has suffix position
     * sa0 + j * sa_step.  Word by word: a run of a few thousand equal symbols is the common case. */
    gab_fmindex *o = E->o;
    const int64_t r0 = E->row, r1 = r0 + len;
    for (int64_t i = r0; i < r1;) {
        const int64_t wend = ((i >> 6) + 1) << 6, e = wend < r1 ? wend : r1;       /* rows [i, e) lie in one 64-row record */
        gab_cp_occ *rec = &o->cp_occ[i >> 6];
        if ((i & 63) == 0) for (int q = 0; q < 4; q++) rec->cp_count[q] = E->run[q] + (q == c ? i - r0 : 0);
        if (c < 4) {
            const int nb = (int)(e - i), sh = (int)(i & 63);                         /* nb bits from position sh, MSB first */
            const uint64_t m = (nb == 64 ? ~0ull : ((1ull << nb) - 1ull) << (64 - nb)) >> sh;
            rec->one_hot_bwt_str[c] |= m;
        }
        i = e;
    }
    for (int64_t i = (r0 + 7) & ~7ll; i < r1; i += 8) {
        const int64_t sa = sa0 + (i - r0) * sa_step;
        o->sa_ls_word[i >> 3] = (uint32_t)(sa & 0xffffffffll); o->sa_ms_byte[i >> 3] = (int8_t)(sa >> 32);
    }
    if (c < 4) E->run[c] += len;
    E->row += len;
}
int gab_mkindex_build_power(const uint8_t *U, int64_t ulen, int64_t m, gab_fmindex *out) {
    memset(out, 0, sizeof(*out));
    const int64_t w = 2 * ulen, k = 2 * m;
    if (ulen <= 0 || k < 3 || 3 * w + 1 >= 0x7fffffffll || (double)k * (double)w > 5.0e11) return -1;
    uint8_t *W = (uint8_t *)malloc((size_t)w), *X = (uint8_t *)malloc((size_t)(3 * w + 1));
    int32_t *SA = (int32_t *)malloc(sizeof(int32_t) * (size_t)(3 * w + 1));
    if (!W || !X || !SA) { free(W); free(X); free(SA); return -2; }
    int64_t cntW[4] = {0, 0, 0, 0};
    for (int64_t i = 0; i < ulen; i++) {
        if (U[i] > 3) { free(W); free(X); free(SA); return -3; }
        W[i] = U[i]; W[w - 1 - i] = (uint8_t)(3 - U[i]);
    }
    for (int64_t i = 0; i < w; i++) cntW[W[i]]++;
    if (is_primitive(W, w) != 1) { free(W); free(X); free(SA); return -5; }
    for (int64_t i = 0; i < 3 * w; i++) X[i] = (uint8_t)(W[i % w] + 1);
    X[3 * w] = 0;
    if (gab_sais_u8(X, SA, (int32_t)(3 * w + 1), 5)) { free(W); free(X); free(SA); return -4; }
    free(X);
    const int64_t N = k * w, n = N + 1;
    out->ref_seq_len = n;
    out->count[0] = 0; out->count[1] = k * cntW[0]; out->count[2] = k * (cntW[0] + cntW[1]);
    out->count[3] = k * (cntW[0] + cntW[1] + cntW[2]); out->count[4] = N;
    out->cp_occ_size = (n >> 6) + 1;
    out->cp_occ = (gab_cp_occ *)calloc((size_t)out->cp_occ_size, sizeof(gab_cp_occ));
    out->n_sa = (n >> 3) + 1;
    out->sa_ms_byte = (int8_t *)calloc((size_t)out->n_sa, 1);
    out->sa_ls_word = (uint32_t *)calloc((size_t)out->n_sa, sizeof(uint32_t));
    if (!out->cp_occ || !out->sa_ms_byte || !out->sa_ls_word) { free(W); free(SA); gab_mkindex_free(out); return -2; }
    emitter E; memset(&E, 0, sizeof E); E.o = out;
    out->sentinel_index = -1;
    for (int64_t i = 0; i <= 3 * w; i++) {
        const int64_t p = SA[i];
        if (p == 3 * w) { emit_rows(&E, W[w - 1], 1, N, 0); continue; }          /* the empty suffix: preceded by T's last base */
        const int64_t e = 2 - p / w, r = p % w;
        const int c = r > 0 ? W[r - 1] : W[w - 1];
        if (e < 2) { emit_rows(&E, c, 1, (k - 1 - e) * w + r, 0); continue; }
        /* e' = 2 .. k-1: positions (k-1-e') * w + r, descending by w; the last one (e' = k-1) is position r of T */
        if (r > 0) emit_rows(&E, c, k - 2, (k - 3) * w + r, -w);
        else {
            emit_rows(&E, c, k - 3, (k - 3) * w, -w);
            out->sentinel_index = E.row;
            emit_rows(&E, 4, 1, 0, 0);                                             /* suffix 0 = T itself: preceded by nothing */
        }
    }
    free(W); free(SA);
    /* the record behind the last row keeps the final counts when n is a multiple of 64 (as the full builder leaves it) */
    if (E.row != n || out->sentinel_index < 0) { gab_mkindex_free(out); return -6; }
    return 0;
}

void gab_mkindex_free(gab_fmindex *idx) {
    free(idx->cp_occ); free(idx->sa_ms_byte); free(idx->sa_ls_word);
    idx->cp_occ = NULL; idx->sa_ms_byte = NULL; idx->sa_ls_word = NULL;
}

int gab_mkindex_write(const gab_fmindex *idx, const char *prefix) {
    char name[4096];
    snprintf(name, sizeof name, "%s.bwt.2bit.64", prefix);
    FILE *f = fopen(name, "wb");
    if (!f) return -1;
    int ok = fwrite(&idx->ref_seq_len, 8, 1, f) == 1 && fwrite(idx->count, 8, 5, f) == 5 &&
             fwrite(idx->cp_occ, sizeof(gab_cp_occ), (size_t)idx->cp_occ_size, f) == (size_t)idx->cp_occ_size &&
             fwrite(idx->sa_ms_byte, 1, (size_t)idx->n_sa, f) == (size_t)idx->n_sa &&
             fwrite(idx->sa_ls_word, 4, (size_t)idx->n_sa, f) == (size_t)idx->n_sa &&
             fwrite(&idx->sentinel_index, 8, 1, f) == 1;
    return fclose(f) == 0 && ok ? 0 : -1;
}

/* Companion files the reference's loader insists on (bwa_idx_load_ele -> bns_restore, bntseq.cpp:73-101,
 * 113-180, 336-352): <prefix>.ann, <prefix>.amb (no holes: the reference is N-free) and the 2-bit <prefix>.pac. */
int gab_mkindex_write_bns(const char *prefix, const uint8_t *fwd, int64_t L, const char *seq_name) {
    char name[4096];
    snprintf(name, sizeof name, "%s.ann", prefix);
    FILE *f = fopen(name, "w");
    if (!f) return -1;
    fprintf(f, "%lld 1 11\n0 %s (null)\n0 %lld 0\n", (long long)L, seq_name, (long long)L);
    if (fclose(f)) return -1;
    snprintf(name, sizeof name, "%s.amb", prefix);
    f = fopen(name, "w");
    if (!f) return -1;
    fprintf(f, "%lld 1 0\n", (long long)L);
    if (fclose(f)) return -1;
    snprintf(name, sizeof name, "%s.pac", prefix);
    f = fopen(name, "wb");
    if (!f) return -1;
    const int64_t nbytes = (L >> 2) + ((L & 3) ? 1 : 0);
    uint8_t *pac = (uint8_t *)calloc((size_t)nbytes + 2, 1);
    if (!pac) { fclose(f); return -2; }
    for (int64_t i = 0; i < L; i++) pac[i >> 2] |= (uint8_t)((fwd[i] & 3) << ((3 - (i & 3)) << 1));
    int64_t total = nbytes;
    if ((L & 3) == 0) pac[total++] = 0;
    pac[total++] = (uint8_t)(L & 3);
    const int ok = fwrite(pac, 1, (size_t)total, f) == (size_t)total;
    free(pac);
    return fclose(f) == 0 && ok ? 0 : -1;
}
