/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * FM-index SMEM seeding, three passes per read, as the fmi benchmark driver runs them
 * (/root/reference/benchmarks/fmi/fmi.cpp:250-348) on BWA-MEM2's FMI_search
 * (/root/reference/benchmarks/fmi/bwa-mem2/x86_64/src/FMI_search.cpp):
 *   backwardExt :1025-1052 (GET_OCC FMI_search.h:66-73), getSMEMsOnePosOneThread :496-670,
 *   getSMEMsAllPosOneThread :672-724, bwtSeedStrategyAllPosOneThread :726-812,
 *   compare_smem :986-1006.
 * The reference processes reads in batches and in lock-step rounds; every read is independent, so
 * this restatement runs the three passes read by read (same SMEM multiset per read, SURVEY.md
 * App. B5) and orders each read's SMEMs by (m ascending, n descending, then s, k ascending).
 */
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_fmi_load(const char *prefix, oracle_fmindex *idx) {
    char name[4096];
    snprintf(name, sizeof name, "%s.bwt.2bit.64", prefix);
    FILE *f = fopen(name, "rb");
    if (!f) return -1;
    memset(idx, 0, sizeof(*idx));
    int64_t count[5];
    if (fread(&idx->ref_seq_len, 8, 1, f) != 1 || fread(count, 8, 5, f) != 5) { fclose(f); return -2; }
    for (int i = 0; i < 5; i++) idx->count[i] = count[i] + 1;          /* FMI_search.cpp:433-436 */
    idx->cp_occ_size = (idx->ref_seq_len >> 6) + 1;
    idx->cp_occ = (oracle_cp_occ *)malloc(sizeof(oracle_cp_occ) * (size_t)idx->cp_occ_size);
    if (fread(idx->cp_occ, sizeof(oracle_cp_occ), (size_t)idx->cp_occ_size, f) != (size_t)idx->cp_occ_size) { fclose(f); return -2; }
    /* sampled suffix array, FMI_search.cpp:439-447: int8 most-significant bytes, then uint32 low words */
    const int64_t n_sa = (idx->ref_seq_len >> 3) + 1;
    idx->sa_ms_byte = (int8_t *)malloc((size_t)n_sa);
    idx->sa_ls_word = (uint32_t *)malloc(4 * (size_t)n_sa);
    if (fread(idx->sa_ms_byte, 1, (size_t)n_sa, f) != (size_t)n_sa || fread(idx->sa_ls_word, 4, (size_t)n_sa, f) != (size_t)n_sa ||
        fread(&idx->sentinel_index, 8, 1, f) != 1) { fclose(f); return -2; }
    fclose(f);
    return 0;
}

void oracle_fmi_from_arrays(oracle_fmindex *idx, int64_t ref_seq_len, const int64_t *file_count,
                            const void *cp_occ, int64_t sentinel_index) {
    idx->ref_seq_len = ref_seq_len;
    for (int i = 0; i < 5; i++) idx->count[i] = file_count[i] + 1;
    idx->cp_occ_size = (ref_seq_len >> 6) + 1;
    idx->cp_occ = (oracle_cp_occ *)malloc(sizeof(oracle_cp_occ) * (size_t)idx->cp_occ_size);
    memcpy(idx->cp_occ, cp_occ, sizeof(oracle_cp_occ) * (size_t)idx->cp_occ_size);
    idx->sentinel_index = sentinel_index;
    idx->sa_ms_byte = NULL; idx->sa_ls_word = NULL;
}

void oracle_fmi_set_sa(oracle_fmindex *idx, const int8_t *sa_ms_byte, const uint32_t *sa_ls_word) {
    const size_t n_sa = (size_t)((idx->ref_seq_len >> 3) + 1);
    idx->sa_ms_byte = (int8_t *)malloc(n_sa); idx->sa_ls_word = (uint32_t *)malloc(4 * n_sa);
    memcpy(idx->sa_ms_byte, sa_ms_byte, n_sa); memcpy(idx->sa_ls_word, sa_ls_word, 4 * n_sa);
}

void oracle_fmi_free(oracle_fmindex *idx) {
    free(idx->cp_occ); free(idx->sa_ms_byte); free(idx->sa_ls_word);
    idx->cp_occ = NULL; idx->sa_ms_byte = NULL; idx->sa_ls_word = NULL;
}

typedef struct { int64_t *calls; const oracle_fmindex *x; } fctx;

static inline int64_t occ(const oracle_fmindex *x, int64_t pp, int c) {
    const oracle_cp_occ *e = &x->cp_occ[pp >> 6];
    const int y = (int)(pp & 63);
    const uint64_t top = y ? ~0ull << (64 - y) : 0;                     /* one_hot_mask_array[y] */
    return e->cp_count[c] + __builtin_popcountll(e->one_hot_bwt_str[c] & top);
}

static oracle_smem backward_ext(const fctx *fc, oracle_smem sm, int a) {
    const oracle_fmindex *x = fc->x;
    int64_t k[4], s[4], l[4];
    (*fc->calls)++;
    for (int b = 0; b < 4; b++) {
        const int64_t o_sp = occ(x, sm.k, b), o_ep = occ(x, sm.k + sm.s, b);
        k[b] = x->count[b] + o_sp;
        s[b] = o_ep - o_sp;
    }
    const int64_t sent = (sm.k <= x->sentinel_index && sm.k + sm.s > x->sentinel_index) ? 1 : 0;
    l[3] = sm.l + sent; l[2] = l[3] + s[3]; l[1] = l[2] + s[2]; l[0] = l[1] + s[1];
    sm.k = k[a]; sm.l = l[a]; sm.s = s[a];
    return sm;
}
static oracle_smem forward_ext(const fctx *fc, oracle_smem sm, int a) {
    oracle_smem t = sm;
    t.k = sm.l; t.l = sm.k;
    oracle_smem r = backward_ext(fc, t, 3 - a);
    oracle_smem o = r;
    o.k = r.l; o.l = r.k;
    return o;
}

/* one call of the body of getSMEMsOnePosOneThread for read `q` at position x; returns next_x */
static int smem_one_pos(const fctx *fc, const uint8_t *q, int len, uint32_t rid, int x, int min_intv,
                        int min_seed_len, oracle_smem *prev, oracle_smem *out, int64_t *nout) {
    const oracle_fmindex *ix = fc->x;
    int next_x = x + 1;
    int a = q[x];
    if (a >= 4) return next_x;
    oracle_smem sm;
    sm.rid = rid; sm.m = (uint32_t)x; sm.n = (uint32_t)x;
    sm.k = ix->count[a]; sm.l = ix->count[3 - a]; sm.s = ix->count[a + 1] - ix->count[a];
    int nprev = 0, j;
    for (j = x + 1; j < len; j++) {
        a = q[j];
        next_x = j + 1;
        if (a >= 4) break;
        oracle_smem nw = forward_ext(fc, sm, a);
        nw.n = (uint32_t)j;
        prev[nprev] = sm;
        nprev += nw.s != sm.s;
        if (nw.s < min_intv) { next_x = j; break; }
        sm = nw;
    }
    if (sm.s >= min_intv) prev[nprev++] = sm;
    for (int p = 0; p < nprev / 2; p++) { oracle_smem t = prev[p]; prev[p] = prev[nprev - 1 - p]; prev[nprev - 1 - p] = t; }
    for (j = x - 1; j >= 0; j--) {
        int ncur = 0;
        int curr_s = -1;                                              /* int, as in the reference */
        a = q[j];
        if (a > 3) break;
        int p;
        for (p = 0; p < nprev; p++) {
            const oracle_smem s0 = prev[p];
            oracle_smem nw = backward_ext(fc, s0, a);
            nw.m = (uint32_t)j;
            if (nw.s < min_intv && (int)(s0.n - s0.m + 1) >= min_seed_len) { out[(*nout)++] = s0; break; }
            if (nw.s >= min_intv && nw.s != curr_s) { curr_s = (int)nw.s; prev[ncur++] = nw; break; }
        }
        p++;
        for (; p < nprev; p++) {
            oracle_smem nw = backward_ext(fc, prev[p], a);
            nw.m = (uint32_t)j;
            if (nw.s >= min_intv && nw.s != curr_s) { curr_s = (int)nw.s; prev[ncur++] = nw; }
        }
        nprev = ncur;
        if (ncur == 0) break;
    }
    if (nprev != 0) {
        const oracle_smem s0 = prev[0];
        if ((int)(s0.n - s0.m + 1) >= min_seed_len) out[(*nout)++] = s0;
    }
    return next_x;
}

static int cmp_smem(const void *pa, const void *pb) {
    const oracle_smem *a = (const oracle_smem *)pa, *b = (const oracle_smem *)pb;
    if (a->rid != b->rid) return a->rid < b->rid ? -1 : 1;
    if (a->m != b->m) return a->m < b->m ? -1 : 1;
    if (a->n != b->n) return a->n > b->n ? -1 : 1;                     /* compare_smem: n descending */
    if (a->s != b->s) return a->s < b->s ? -1 : 1;                     /* tie-break (unspecified in the reference) */
    if (a->k != b->k) return a->k < b->k ? -1 : 1;
    return 0;
}

/* all three passes for one read; appends to out (capacity >= 3 * len + 8), returns count */
int64_t oracle_fmi_read(const oracle_fmindex *idx, const uint8_t *q, int len, uint32_t rid, int min_seed_len,
                        oracle_smem *out, int64_t *ext_calls) {
    fctx fc; fc.calls = ext_calls; fc.x = idx;
    oracle_smem *prev = (oracle_smem *)malloc(sizeof(oracle_smem) * (size_t)(len + 2));
    int64_t n1 = 0;
    /* pass 1: fmi.cpp:288-298, min_intv = 1 */
    for (int x = 0; x < len;) x = smem_one_pos(&fc, q, len, rid, x, 1, min_seed_len, prev, out, &n1);
    /* pass 2: re-seed long, low-occurrence SMEMs at their midpoint (fmi.cpp:300-324) */
    const int split_len = (int)(min_seed_len * 1.5 + .499);
    int64_t n2 = n1;
    for (int64_t j = 0; j < n1; j++) {
        const int start = (int)out[j].m, end = (int)out[j].n + 1;
        if (end - start < split_len || out[j].s > 10) continue;
        smem_one_pos(&fc, q, len, rid, (end + start) >> 1, (int)(out[j].s + 1), min_seed_len, prev, out, &n2);
    }
    /* pass 3: bwtSeedStrategyAllPosOneThread(max_intv = 20, minSeedLen + 1) (fmi.cpp:326-336) */
    int64_t n3 = n2;
    const int max_intv = 20, msl = min_seed_len + 1;
    for (int x = 0; x < len;) {
        int next_x = x + 1;
        int a = q[x];
        if (a < 4) {
            oracle_smem sm;
            sm.rid = rid; sm.m = (uint32_t)x; sm.n = (uint32_t)x;
            sm.k = idx->count[a]; sm.l = idx->count[3 - a]; sm.s = idx->count[a + 1] - idx->count[a];
            for (int j = x + 1; j < len; j++) {
                next_x = j + 1;
                a = q[j];
                if (a >= 4) break;
                sm = forward_ext(&fc, sm, a);
                sm.n = (uint32_t)j;
                if (sm.s < max_intv && (int)(sm.n - sm.m + 1) >= msl) {
                    if (sm.s > 0) out[n3++] = sm;
                    break;
                }
            }
        }
        x = next_x;
    }
    free(prev);
    qsort(out, (size_t)n3, sizeof(oracle_smem), cmp_smem);
    return n3;
}

int64_t oracle_fmi_batch(const oracle_fmindex *idx, const uint8_t *enc, int32_t stride, const int32_t *len,
                         int64_t nreads, int min_seed_len, int threads, oracle_smem **out_p, int64_t *read_off,
                         int64_t *ext_calls) {
    oracle_smem **per = (oracle_smem **)calloc((size_t)nreads, sizeof(*per));
    int64_t calls = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : calls)
    for (int64_t r = 0; r < nreads; r++) {
        oracle_smem *buf = (oracle_smem *)malloc(sizeof(oracle_smem) * (size_t)(3 * len[r] + 8));
        int64_t c = 0;
        const int64_t n = oracle_fmi_read(idx, enc + r * (int64_t)stride, len[r], (uint32_t)r, min_seed_len, buf, &c);
        per[r] = buf; read_off[r + 1] = n; calls += c;
    }
    read_off[0] = 0;
    for (int64_t r = 0; r < nreads; r++) read_off[r + 1] += read_off[r];
    oracle_smem *all = (oracle_smem *)malloc(sizeof(oracle_smem) * (size_t)(read_off[nreads] + 1));
    for (int64_t r = 0; r < nreads; r++) {
        memcpy(all + read_off[r], per[r], sizeof(oracle_smem) * (size_t)(read_off[r + 1] - read_off[r]));
        free(per[r]);
    }
    free(per);
    *out_p = all;
    if (ext_calls) *ext_calls = calls;
    return read_off[nreads];
}

void oracle_fmi_release(oracle_smem *p) { free(p); }

/* ---- suffix-array look-up ------------------------------------------------------------------ */
/* get_sa_entry_compressed, FMI_search.cpp:1103-1175: rows that are a multiple of 8 are stored; any other row walks the
 * LF mapping (one CP_OCC record per step) until it reaches a stored row or the sentinel, counting the steps. */
static int64_t sa_entry_compressed(const oracle_fmindex *x, int64_t pos, int64_t *steps) {
    int64_t offset = 0, sp = pos;
    while ((sp & 7) != 0) {
        const oracle_cp_occ *e = &x->cp_occ[sp >> 6];
        const int y = 64 - (int)(sp & 63) - 1;
        int b = 4;
        for (int c = 0; c < 4; c++) if ((e->one_hot_bwt_str[c] >> y) & 1) { b = c; break; }
        if (b == 4) return offset;                                      /* the sentinel row: SA = 0 */
        sp = x->count[b] + occ(x, sp, b);
        offset++;
        if (steps) ++*steps;
    }
    return ((int64_t)x->sa_ms_byte[sp >> 3] << 32) + (int64_t)x->sa_ls_word[sp >> 3] + offset;
}

int64_t oracle_fmi_sa_count(const oracle_smem *smems, int64_t n, int32_t max_occ, int64_t *coord_off) {
    int64_t total = 0;
    for (int64_t i = 0; i < n; i++) {                                   /* loop shape of FMI_search.cpp:1177-1196 */
        coord_off[i] = total;
        const int64_t hi = smems[i].k + smems[i].s, step = smems[i].s > max_occ ? smems[i].s / max_occ : 1;
        int32_t c = 0;
        for (int64_t j = smems[i].k; j < hi && c < max_occ; j += step) c++;
        total += c;
    }
    coord_off[n] = total;
    return total;
}

int64_t oracle_fmi_sa_lookup(const oracle_fmindex *idx, const oracle_smem *smems, int64_t n, int32_t max_occ,
                             const int64_t *coord_off, int64_t *coords, int64_t *lf_steps) {
    if (!idx->sa_ms_byte || !idx->sa_ls_word) return -1;
    int64_t steps = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : steps)
#endif
    for (int64_t i = 0; i < n; i++) {
        const int64_t hi = smems[i].k + smems[i].s, step = smems[i].s > max_occ ? smems[i].s / max_occ : 1;
        int32_t c = 0;
        for (int64_t j = smems[i].k; j < hi && c < max_occ; j += step, c++)
            coords[coord_off[i] + c] = sa_entry_compressed(idx, j, &steps);
    }
    if (lf_steps) *lf_steps = steps;
    return coord_off[n];
}
