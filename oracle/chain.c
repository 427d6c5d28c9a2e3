/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * minimap2 seed chaining, two flavours:
 *   oracle_chain_call      follows chain_dp of the `chain` benchmark
 *       (/root/reference/benchmarks/chain/src/host_kernel.cpp:30-94): 64-bit anchors,
 *       segment ids, max_skip = 25 early exit through targets[], max_iter = 5000.
 *   oracle_fastchain_call  follows chain_dp of `fast-chain` as its AVX2 / AVX-512 builds
 *       compute it (/root/reference/benchmarks/fast-chain/src/host_kernel.cpp:175-407
 *       AVX-512, :408-683 AVX2; gap cost :72-132): truncated 32-bit coordinates, no
 *       max_skip, fp32 floor gap cost when the predecessor window has more than six
 *       anchors, double gap cost otherwise.  NOT the #else scalar
 *       fallback of that file, which rounds differently.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int floor_log2_u32(uint32_t v) { return 31 - __builtin_clz(v); }

int64_t oracle_chain_call(const oracle_chain_hdr *h, const uint64_t *x, const uint64_t *y,
                          int32_t *score, int32_t *parent) {
    const int64_t n = h->n;
    const int max_iter = 5000, max_skip = 25;
    const int max_dist_x = h->max_dist_x, max_dist_y = h->max_dist_y, bw = h->bw, n_segs = h->n_segs;
    const float avg_qspan = h->avg_qspan;
    int32_t *target = (int32_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int32_t));   /* vector::resize zero-fills */
    int64_t st = 0, evals = 0;
    for (int64_t i = 0; i < n; i++) {
        const uint64_t ri = x[i];
        const int32_t qi = (int32_t)y[i];
        const int32_t q_span = (int32_t)(y[i] >> 32 & 0xff);
        const int32_t sidi = (int32_t)((y[i] >> 48) & 0xff);
        int32_t best = q_span, n_skip = 0;
        int64_t best_j = -1;
        while (st < i && ri > x[st] + (uint64_t)(int64_t)max_dist_x) st++;   /* :56 (int promoted to u64) */
        if (i - st > max_iter) st = i - max_iter;                               /* :57 */
        for (int64_t j = i - 1; j >= st; j--) {
            evals++;
            const int64_t dr = (int64_t)(ri - x[j]);
            const int32_t dq = qi - (int32_t)y[j];
            const int32_t sidj = (int32_t)((y[j] >> 48) & 0xff);
            const int same = sidi == sidj;
            if ((same && dr == 0) || dq <= 0) continue;                         /* :62 */
            if ((same && dq > max_dist_y) || dq > max_dist_x) continue;         /* :63 */
            const int32_t dd = (int32_t)(dr > dq ? dr - dq : dq - dr);          /* :64 (int64 -> int32) */
            if (same && dd > bw) continue;                                      /* :65 */
            if (n_segs > 1 && same && dr > max_dist_y) continue;                /* :66, is_cdna == 0 */
            const int32_t min_d = (int32_t)(dq < dr ? dq : dr);
            int32_t sc = min_d > q_span ? q_span : (int32_t)(dq < dr ? dq : dr);
            const int32_t log_dd = dd ? floor_log2_u32((uint32_t)dd) : 0;
            int32_t gap_cost = 0;
            if (!same) {                                                        /* :71-77 */
                const int c_lin = (int)(dd * .01 * avg_qspan), c_log = log_dd;
                if (dr == 0) ++sc;
                else gap_cost = c_lin < c_log ? c_lin : c_log;                  /* dr > dq || sidi != sidj */
            } else {
                gap_cost = (int)(dd * .01 * avg_qspan) + (log_dd >> 1);         /* :78 */
            }
            sc -= (int)((double)gap_cost * 1.0f + .499);                        /* :79, gap_scale = 1 */
            sc += score[j];
            if (sc > best) {
                best = sc; best_j = j;
                if (n_skip > 0) --n_skip;
            } else if (target[j] == (int32_t)i) {
                if (++n_skip > max_skip) break;
            }
            if (parent[j] >= 0) target[parent[j]] = (int32_t)i;
        }
        score[i] = best;
        parent[i] = (int32_t)best_j;
    }
    free(target);
    return evals;
}

int64_t oracle_fastchain_call(const oracle_chain_hdr *h, const uint64_t *x, const uint64_t *y,
                              int32_t *score, int32_t *parent) {
    const int64_t n = h->n;
    const int max_iter = 5000;
    const int32_t max_dist_x = h->max_dist_x, max_dist_y = h->max_dist_y, bw = h->bw;
    const float avg_qspan = h->avg_qspan;
    const float k32 = (float)(0.01 * (double)avg_qspan);      /* host_kernel.cpp:87,124 */
    int64_t st = 0, evals = 0;
    for (int64_t i = 0; i < n; i++) {
        const int32_t q_span = (int32_t)(y[i] >> 32 & 0xff);
        int32_t best = q_span, best_j = -1;
        /* window start on the full 64-bit x (:200, :429): unsigned compare against (u64)(int)dr */
        while (st < i && !(x[i] - x[st] <= (uint64_t)(int64_t)max_dist_x)) st++;
        if (i - st > max_iter) st = i - max_iter;
        const uint32_t ri = (uint32_t)x[i], qi = (uint32_t)y[i];
        const int wide = !((i - 1) - st <= 5);                 /* :211 / :440 */
        for (int64_t j = i - 1; j >= st; j--) {
            evals++;
            const int32_t ddr = (int32_t)(ri - (uint32_t)x[j]);
            const int32_t ddq = (int32_t)(qi - (uint32_t)y[j]);
            const uint32_t diff = (uint32_t)ddr - (uint32_t)ddq;
            const int32_t dd = (int32_t)((int32_t)diff < 0 ? 0u - diff : diff);   /* abs with wrap-around */
            if (dd > bw || ddr == 0 || ddq <= 0 || ddq > max_dist_y || ddq > max_dist_x) continue;
            int32_t oc = ddr < ddq ? ddr : ddq;
            if (q_span < oc) oc = q_span;
            const int32_t lg = dd ? floor_log2_u32((uint32_t)dd) : 0;
            int32_t gc;
            if (wide) gc = (int32_t)floorf((float)dd * k32) + (lg >> 1);          /* :87-92 */
            else gc = (int)(dd * .01 * avg_qspan) + (lg >> 1);                    /* :392 */
            const int32_t sc = (int32_t)((uint32_t)score[j] + (uint32_t)oc - (uint32_t)gc);
            if (sc > best) { best = sc; best_j = (int32_t)j; }
        }
        score[i] = best;
        parent[i] = best_j;
    }
    return evals;
}

void oracle_chain_batch(int mode, const oracle_chain_hdr *hdr, const int64_t *call_off, int64_t ncalls,
                        const uint64_t *x, const uint64_t *y, int threads,
                        int32_t *score, int32_t *parent, int64_t *evals) {
    int64_t total = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic) reduction(+ : total)
    for (int64_t c = 0; c < ncalls; c++) {
        const int64_t o = call_off[c];
        total += mode == 0 ? oracle_chain_call(&hdr[c], x + o, y + o, score + o, parent + o)
                           : oracle_fastchain_call(&hdr[c], x + o, y + o, score + o, parent + o);
    }
    if (evals) *evals = total;
}
