/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Gap-affine wavefront alignment (WFA v1, "complete" mode and the adaptive reduction), as `align_benchmark`
 * of the wfa benchmark runs it:  affine_wavefronts_align
 *   (/root/reference/benchmarks/wfa/gap_affine/affine_wavefront_align.c:325-361):
 *   extend (affine_wavefront_extend.c:241-252, scalar form) -> end test
 *   (affine_wavefront_utils.c:83-102) -> next wavefront (affine_wavefront_align.c:41-321)
 *   ... -> backtrace (affine_wavefront_backtrace.c:276-387) -> CIGAR operations.
 * Offsets are int32, a missing source reads as -10 (affine_wavefront.h:48), the strings behave as
 * if padded with 'X' (pattern) and 'Y' (text) on both sides (wfa/utils/string_padded.c:88-117).
 * Adaptive mode (pen->min_wavefront_length >= 0; align_benchmark.c:359-368, --minimum-wavefront-length /
 * --maximum-difference-distance): after every extension the M wavefront drops the outer diagonals that lag more
 * than the threshold behind the best one (affine_wavefront_extend.c:85-154); the next wavefronts are computed
 * from the reduced [lo, hi] (affine_wavefront_align.c:95-133), the backtrace still reads the whole allocated
 * [lo_base, hi_base] (affine_wavefront_backtrace.c:75-226).
 * Per-pair state is created fresh (the reference's stale-slot reuse is benign).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define WF_NULL (-10)

/* off is centred: off[k], lob <= k <= hib as allocated; [lo, hi] is the range after reduction; NULL if absent */
typedef struct { int lo, hi, lob, hib; int32_t *off; } wf_t;

static inline int wf_get(const wf_t *w, int k) { return (w->off && w->lo <= k && k <= w->hi) ? w->off[k] : WF_NULL; }
static inline int wf_lo(const wf_t *w) { return w->off ? w->lo : 1; }     /* null wavefront: lo=1, hi=-1 */
static inline int wf_hi(const wf_t *w) { return w->off ? w->hi : -1; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

static wf_t wf_alloc(int lo, int hi) {
    wf_t w; w.lo = w.lob = lo; w.hi = w.hib = hi;
    int32_t *mem = (int32_t *)malloc(sizeof(int32_t) * (size_t)(hi - lo + 2));
    w.off = mem - lo;
    return w;
}
static void wf_free(wf_t *w) { if (w->off) free(w->off + w->lob); w->off = NULL; }
/* backtrace view: the allocated range */
static inline int wf_get_base(const wf_t *w, int k) { return (w->off && w->lob <= k && k <= w->hib) ? w->off[k] : WF_NULL; }
static inline int wf_has_base(const wf_t *w, int k) { return w->off && w->lob <= k && k <= w->hib; }

/* affine_wavefronts_compute_distance, affine_wavefront_utils.c:64-74 */
static inline int wf_distance(int plen, int tlen, int offset, int k) { return imax(plen - (offset - k), tlen - offset); }

/* affine_wavefronts_reduce_wavefronts, affine_wavefront_extend.c:114-154 (the reduced range can never become empty:
 * the bottom scan stops at min(ak-1, hi), the top scan at max(ak+1, lo)) */
static void wf_reduce(wf_t *m, wf_t *i, wf_t *d, int plen, int tlen, int min_len, int max_thr) {
    if (!m->off || m->hi - m->lo + 1 < min_len) return;
    const int ak = tlen - plen;
    int min_d = imax(plen, tlen);
    for (int k = m->lo; k <= m->hi; k++) min_d = imin(min_d, wf_distance(plen, tlen, m->off[k], k));
    const int top = imin(ak - 1, m->hi);
    for (int k = m->lo; k < top; k++) {
        if (wf_distance(plen, tlen, m->off[k], k) - min_d <= max_thr) break;
        m->lo++;
    }
    const int bottom = imax(ak + 1, m->lo);
    for (int k = m->hi; k > bottom; k--) {
        if (wf_distance(plen, tlen, m->off[k], k) - min_d <= max_thr) break;
        m->hi--;
    }
    if (i->off) { i->lo = imax(i->lo, m->lo); i->hi = imin(i->hi, m->hi); }
    if (d->off) { d->lo = imax(d->lo, m->lo); d->hi = imin(d->hi, m->hi); }
}

int oracle_wfa_one(const oracle_wfa_penalties *pen, const char *pattern, int plen, const char *text, int tlen,
                   char *ops_out, int *score_out, int64_t *cells) {
    const int x = pen->mismatch, oe = pen->gap_opening + pen->gap_extension, e = pen->gap_extension;
    const int adaptive = pen->min_wavefront_length >= 0;
    /* complete mode: the all-mismatch + one-gap alignment bounds the optimum.  The adaptive heuristic may end above it;
     * the reference sizes its tables for 100000-base strings (align_benchmark.c:62,359-368, affine_wavefront.c:87-89),
     * here they grow on demand up to that size */
    const int bound = imin(plen, tlen) * x + pen->gap_opening + abs(plen - tlen) * e + 1;
    const int max_score = adaptive ? 100000 * x + pen->gap_opening - 1 : bound;
    int cap_s = bound + 2;
    wf_t *M = (wf_t *)calloc((size_t)cap_s, sizeof(wf_t));
    wf_t *I = (wf_t *)calloc((size_t)cap_s, sizeof(wf_t));
    wf_t *D = (wf_t *)calloc((size_t)cap_s, sizeof(wf_t));
    const wf_t none = {1, -1, 1, -1, NULL};
#define PCH(v) (((v) >= 0 && (v) < plen) ? pattern[v] : 'X')
#define TCH(h) (((h) >= 0 && (h) < tlen) ? text[h] : 'Y')
#define SRC(A, s) ((s) >= 0 ? &(A)[s] : &none)
    M[0] = wf_alloc(0, 0); M[0].off[0] = 0;
    const int ak = tlen - plen;
    int score = 0;
    int64_t work = 0;
    for (;;) {
        /* extend */
        if (M[score].off) {
            for (int k = M[score].lo; k <= M[score].hi; k++) {
                int o = M[score].off[k], v = o - k, h = o;
                while (PCH(v) == TCH(h)) { v++; h++; o++; work++; }
                M[score].off[k] = o;
            }
        }
        if (adaptive) wf_reduce(&M[score], &I[score], &D[score], plen, tlen, pen->min_wavefront_length, pen->max_distance_threshold);
        /* end reached? */
        if (M[score].off && M[score].lo <= ak && ak <= M[score].hi && M[score].off[ak] >= tlen) break;
        if (score >= max_score) { score = -1; break; }   /* cannot happen for a correct WFA */
        score++;
        if (score >= cap_s) {
            const int ncap = cap_s * 2;
            M = (wf_t *)realloc(M, (size_t)ncap * sizeof(wf_t)); I = (wf_t *)realloc(I, (size_t)ncap * sizeof(wf_t));
            D = (wf_t *)realloc(D, (size_t)ncap * sizeof(wf_t));
            memset(M + cap_s, 0, (size_t)(ncap - cap_s) * sizeof(wf_t)); memset(I + cap_s, 0, (size_t)(ncap - cap_s) * sizeof(wf_t));
            memset(D + cap_s, 0, (size_t)(ncap - cap_s) * sizeof(wf_t));
            cap_s = ncap;
        }
        const wf_t *msub = SRC(M, score - x), *mgap = SRC(M, score - oe), *iext = SRC(I, score - e), *dext = SRC(D, score - e);
        if (!msub->off && !mgap->off && !iext->off && !dext->off) continue;
        const int lo = imin(imin(wf_lo(msub), wf_lo(mgap)), imin(wf_lo(iext), wf_lo(dext))) - 1;
        const int hi = imax(imax(wf_hi(msub), wf_hi(mgap)), imax(wf_hi(iext), wf_hi(dext))) + 1;
        M[score] = wf_alloc(lo, hi);
        const int has_i = mgap->off || iext->off, has_d = mgap->off || dext->off;
        if (has_i) I[score] = wf_alloc(lo, hi);
        if (has_d) D[score] = wf_alloc(lo, hi);
        for (int k = lo; k <= hi; k++) {
            const int sub = (msub->off && msub->lo <= k && k <= msub->hi) ? msub->off[k] + 1 : WF_NULL;
            int best = sub;
            if (has_i) {
                const int ins = imax(wf_get(mgap, k - 1), wf_get(iext, k - 1)) + 1;
                I[score].off[k] = ins;
                best = imax(best, ins);
            }
            if (has_d) {
                const int del = imax(wf_get(mgap, k + 1), wf_get(dext, k + 1));
                D[score].off[k] = del;
                best = imax(best, del);
            }
            M[score].off[k] = best;
        }
        work += hi - lo + 1;
    }
    if (cells) *cells += work;
    int nops = -1;
    if (score >= 0) {
        /* backtrace: ops are written right-aligned into a plen+tlen buffer, then moved to the front */
        const int cap = plen + tlen;
        char *buf = (char *)malloc((size_t)cap + 1);
        int pos = cap - 1;                       /* begin_offset */
        /* A sequence that contains the OTHER sequence's padding byte can match that padding and run past the end; the
         * reference then writes more than plen + tlen operations into a buffer of that size (undefined behaviour).  Here
         * and in the GPU kernels the writes in front of the buffer are dropped: the last plen + tlen operations remain. */
#define PUT(c_) do { if (pos >= 0) buf[pos] = (c_); pos--; } while (0)
        int s = score, k = ak, offset = M[score].off[k];
        enum { BT_M, BT_I, BT_D } type = BT_M;
#define VALID(k_, o_) ((o_) - (k_) > 0 && (o_) - (k_) <= plen && (o_) > 0 && (o_) <= tlen)
        int valid = VALID(k, offset);
        int v = offset - k, h = offset;
        while (v > 0 && h > 0 && s > 0) {
            if (!valid) {
                valid = VALID(k, offset);
                if (valid) {                      /* trailing gap, affine_wavefront_backtrace.c:48-63 */
                    if (k < ak) for (int i = k; i < ak; i++) PUT('I');
                    else if (k > ak) for (int i = ak; i < k; i++) PUT('D');
                }
            }
            const int s_go = s - oe, s_ge = s - e, s_mm = s - x;
            const int del_ext = type == BT_I ? WF_NULL : (s_ge >= 0 ? wf_get_base(&D[s_ge], k + 1) : WF_NULL);
            const int del_open = type == BT_I ? WF_NULL : (s_go >= 0 ? wf_get_base(&M[s_go], k + 1) : WF_NULL);
            const int ins_ext = type == BT_D ? WF_NULL : (s_ge >= 0 && wf_has_base(&I[s_ge], k - 1) ? I[s_ge].off[k - 1] + 1 : WF_NULL);
            const int ins_open = type == BT_D ? WF_NULL : (s_go >= 0 && wf_has_base(&M[s_go], k - 1) ? M[s_go].off[k - 1] + 1 : WF_NULL);
            const int misms = type != BT_M ? WF_NULL : (s_mm >= 0 && wf_has_base(&M[s_mm], k) ? M[s_mm].off[k] + 1 : WF_NULL);
            const int max_all = imax(misms, imax(imax(ins_ext, ins_open), imax(del_ext, del_open)));
            if (type == BT_M) {
                for (int i = 0; i < offset - max_all; i++) PUT('M');
                offset = max_all;
            }
            if (max_all == del_ext) { if (valid) PUT('D'); s = s_ge; k++; type = BT_D; }
            else if (max_all == del_open) { if (valid) PUT('D'); s = s_go; k++; type = BT_M; }
            else if (max_all == ins_ext) { if (valid) PUT('I'); s = s_ge; k--; offset--; type = BT_I; }
            else if (max_all == ins_open) { if (valid) PUT('I'); s = s_go; k--; offset--; type = BT_M; }
            else { if (valid) PUT('X'); s = s_mm; offset--; }      /* max_all == misms */
            v = offset - k; h = offset;
        }
        if (s == 0) { for (int i = 0; i < offset; i++) PUT('M'); }
        else { while (v > 0) { PUT('D'); v--; } while (h > 0) { PUT('I'); h--; } }
        if (pos < -1) pos = -1;
        pos++;
        nops = cap - pos;
        if (ops_out) memcpy(ops_out, buf + pos, (size_t)nops);
        free(buf);
    }
    if (score_out) *score_out = score;
    for (int s = 0; s < cap_s; s++) { wf_free(&M[s]); wf_free(&I[s]); wf_free(&D[s]); }
    free(M); free(I); free(D);
    return nops;
}

void oracle_wfa_batch(const oracle_wfa_penalties *pen, const char *pat, const int64_t *pat_off, const int32_t *pat_len,
                      const char *txt, const int64_t *txt_off, const int32_t *txt_len, int64_t n, int threads,
                      char *ops, const int64_t *ops_off, int32_t *ops_len, int32_t *score, int64_t *cells) {
    int64_t total = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : total)
    for (int64_t i = 0; i < n; i++) {
        int64_t w = 0; int sc = 0;
        ops_len[i] = oracle_wfa_one(pen, pat + pat_off[i], pat_len[i], txt + txt_off[i], txt_len[i],
                                    ops + ops_off[i], &sc, &w);
        score[i] = sc;
        total += w;
    }
    if (cells) *cells = total;
}
