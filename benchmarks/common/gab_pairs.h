/* '>' pattern / '<' text line-pair input shared by the bpm and wfa drivers
 * (/root/reference/benchmarks/bpm/tools/align_benchmark.c:172-203, wfa/tools/align_benchmark.c:131-183):
 * lines are read two at a time with getline; the first character and the newline of each are dropped
 * (length = line_length - 2).  All sequences are stored back to back in ONE slab. */
#ifndef GAB_PAIRS_H
#define GAB_PAIRS_H
#include "gab_driver.h"
typedef struct {
    char *slab; size_t used, cap;
    int64_t *off1, *off2;     /* offsets of the '>' and '<' sequences */
    int32_t *len1, *len2;
    int64_t n, ncap;
} gab_pairs;
static inline void gab_pairs_read(FILE *f, gab_pairs *p) {
    memset(p, 0, sizeof *p);
    p->cap = 1 << 20; p->slab = (char *)malloc(p->cap);
    p->ncap = 1 << 16;
    p->off1 = (int64_t *)malloc(8 * (size_t)p->ncap); p->off2 = (int64_t *)malloc(8 * (size_t)p->ncap);
    p->len1 = (int32_t *)malloc(4 * (size_t)p->ncap); p->len2 = (int32_t *)malloc(4 * (size_t)p->ncap);
    char *l1 = NULL, *l2 = NULL; size_t a1 = 0, a2 = 0;
    for (;;) {
        const ssize_t n1 = getline(&l1, &a1, f), n2 = getline(&l2, &a2, f);
        if (n1 == -1 || n2 == -1) break;
        const int s1 = (int)n1 - 2 > 0 ? (int)n1 - 2 : 0, s2 = (int)n2 - 2 > 0 ? (int)n2 - 2 : 0;
        if (p->n == p->ncap) {
            p->ncap *= 2;
            p->off1 = (int64_t *)realloc(p->off1, 8 * (size_t)p->ncap); p->off2 = (int64_t *)realloc(p->off2, 8 * (size_t)p->ncap);
            p->len1 = (int32_t *)realloc(p->len1, 4 * (size_t)p->ncap); p->len2 = (int32_t *)realloc(p->len2, 4 * (size_t)p->ncap);
        }
        while (p->used + (size_t)s1 + (size_t)s2 + 16 > p->cap) { p->cap *= 2; p->slab = (char *)realloc(p->slab, p->cap); }
        p->off1[p->n] = (int64_t)p->used; memcpy(p->slab + p->used, l1 + 1, (size_t)s1); p->used += (size_t)s1;
        p->off2[p->n] = (int64_t)p->used; memcpy(p->slab + p->used, l2 + 1, (size_t)s2); p->used += (size_t)s2;
        p->len1[p->n] = s1; p->len2[p->n] = s2;
        p->n++;
    }
    free(l1); free(l2);
    memset(p->slab + p->used, 0, 8);     /* 4-byte device reads past the last base stay inside the slab */
    p->used += 8;
}
static inline void gab_pairs_free(gab_pairs *p) { free(p->slab); free(p->off1); free(p->off2); free(p->len1); free(p->len2); }
#endif
