/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Banded Smith-Waterman seed extension, one pair at a time.  Restates
 * BandedPairWiseSW::scalarBandedSWA
 *   (/root/reference/benchmarks/bsw/src/bandedSWA.cpp:132-253),
 * which SURVEY.md App. B1 and tests/test_bsw_oracle.py show to be score-
 * identical to the vector path the driver actually calls (getScores16,
 * bandedSWA.cpp:1128-1435 and the three other ISA copies).
 *
 * State per query column j (0..qlen):  Hd[j] = H(i-1, j-1)  (the diagonal
 * predecessor for the row being computed) and Ev[j] = E(i, j).  Both arrays
 * persist across rows and cells outside the live band keep whatever an earlier
 * row left there -- the reference depends on that (band end may grow by two
 * columns per row and then reads those cells), so it is reproduced literally.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void oracle_bsw_fill_scmat(int a, int b, int ambig, int8_t mat[25]) {
    /* bsw/src/main_banded.cpp:94-102 */
    for (int r = 0; r < 5; r++)
        for (int c = 0; c < 5; c++)
            mat[r * 5 + c] = (int8_t)((r == 4 || c == 4) ? ambig : (r == c ? a : -b));
}

static int64_t bsw_core(const oracle_bsw_params *p, int qlen, const uint8_t *query,
                        int tlen, const uint8_t *target, int h0, oracle_bsw_result *out,
                        int32_t *Hd, int32_t *Ev) {
    const int oe_del = p->o_del + p->e_del, oe_ins = p->o_ins + p->e_ins;
    const int e_del = p->e_del, e_ins = p->e_ins;
    int64_t cells = 0;

    memset(Hd, 0, sizeof(int32_t) * (size_t)(qlen + 1));
    memset(Ev, 0, sizeof(int32_t) * (size_t)(qlen + 1));

    /* row -1: bandedSWA.cpp:159-161 */
    Hd[0] = h0;
    if (qlen >= 1) Hd[1] = h0 > oe_ins ? h0 - oe_ins : 0;
    for (int j = 2; j <= qlen && Hd[j - 1] > e_ins; j++) Hd[j] = Hd[j - 1] - e_ins;

    /* band width clamp: bandedSWA.cpp:164-172 */
    int best_sc = 0;
    for (int k = 0; k < 25; k++) if (p->mat[k] > best_sc) best_sc = p->mat[k];
    int w = p->w;
    int lim = (int)((double)(qlen * best_sc + p->end_bonus - p->o_ins) / e_ins + 1.);
    if (lim < 1) lim = 1;
    if (w > lim) w = lim;
    lim = (int)((double)(qlen * best_sc + p->end_bonus - p->o_del) / e_del + 1.);
    if (lim < 1) lim = 1;
    if (w > lim) w = lim;

    int best = h0, best_i = -1, best_j = -1, g_i = -1, gscore = -1, max_off = 0;
    int beg = 0, end = qlen;
    for (int i = 0; i < tlen; i++) {
        const int8_t *srow = p->mat + 5 * target[i];
        if (beg < i - w) beg = i - w;                      /* :183-185 */
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        int hleft = 0;                                     /* H(i, beg-1), :187-190 */
        if (beg == 0) {
            hleft = h0 - (p->o_del + e_del * (i + 1));
            if (hleft < 0) hleft = 0;
        }
        int f = 0, rowmax = 0, rowmax_j = -1, j;
        for (j = beg; j < end; j++) {                      /* :191-216 */
            int diag = Hd[j], e = Ev[j];
            Hd[j] = hleft;
            int M = diag ? diag + srow[query[j]] : 0;
            int h = M > e ? M : e;
            if (f > h) h = f;
            hleft = h;
            if (!(rowmax > h)) rowmax_j = j;               /* ties -> later column */
            if (h > rowmax) rowmax = h;
            int t = M - oe_del; if (t < 0) t = 0;
            e -= e_del; if (t > e) e = t;
            Ev[j] = e;
            t = M - oe_ins; if (t < 0) t = 0;
            f -= e_ins; if (t > f) f = t;
        }
        cells += (end > beg) ? end - beg : 0;
        Hd[end] = hleft; Ev[end] = 0;                      /* :217 */
        if (j == qlen) {                                   /* :218-221 */
            if (!(gscore > hleft)) g_i = i;
            if (hleft > gscore) gscore = hleft;
        }
        if (rowmax == 0) break;                            /* :222 */
        if (rowmax > best) {                               /* :223-225 */
            best = rowmax; best_i = i; best_j = rowmax_j;
            int off = rowmax_j - i; if (off < 0) off = -off;
            if (off > max_off) max_off = off;
        } else if (p->zdrop > 0) {                         /* :226-232 */
            int di = i - best_i, dj = rowmax_j - best_j;
            if (di > dj) {
                if (best - rowmax - (di - dj) * e_del > p->zdrop) break;
            } else {
                if (best - rowmax - (dj - di) * e_ins > p->zdrop) break;
            }
        }
        /* drop all-zero cells at both band edges: :234-237 */
        for (j = beg; j < end && Hd[j] == 0 && Ev[j] == 0; j++) {}
        beg = j;
        for (j = end; j >= beg && Hd[j] == 0 && Ev[j] == 0; j--) {}
        end = j + 2 < qlen ? j + 2 : qlen;
    }
    out->score = best;
    out->qle = best_j + 1;
    out->tle = best_i + 1;
    out->gtle = g_i + 1;
    out->gscore = gscore;
    out->max_off = max_off;
    return cells;
}

void oracle_bsw_one(const oracle_bsw_params *p, int qlen, const uint8_t *query,
                    int tlen, const uint8_t *target, int h0, oracle_bsw_result *out) {
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(qlen + 1));
    bsw_core(p, qlen, query, tlen, target, h0, out, buf, buf + qlen + 1);
    free(buf);
}

void oracle_bsw_batch(const oracle_bsw_params *p, const uint8_t *ref, const int64_t *ref_off,
                      const uint8_t *qry, const int64_t *qry_off, const int32_t *len1,
                      const int32_t *len2, const int32_t *h0, int64_t n, int threads,
                      oracle_bsw_result *out, int64_t *cells) {
    int64_t total = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel reduction(+ : total)
    {
        int cap = 512;
        int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(cap + 1));
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < n; i++) {
            int ql = len2[i];
            if (ql > cap) {
                cap = ql;
                free(buf);
                buf = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(cap + 1));
            }
            total += bsw_core(p, ql, qry + qry_off[i], len1[i], ref + ref_off[i], h0[i],
                              &out[i], buf, buf + ql + 1);
        }
        free(buf);
    }
    if (cells) *cells = total;
}
