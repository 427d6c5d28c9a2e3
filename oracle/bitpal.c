/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The bpm benchmark's BitPAl algorithms, as `align_benchmark -a bitpal-edit | bitpal-scored` computes the printed
 * score:  benchmark_bitpal_m0_x1_g1 / benchmark_bitpal_m1_x4_g2
 *   (/root/reference/benchmarks/bpm/benchmark/benchmark_bitpal.c:30-54), which run the generated bit-vector programs
 *   bpm/bitpal/bitpal.m0.x1.g1.c:31-282 and bpm/bitpal/bitpal.m1.x4.g2.c.  Those programs encode, 63 columns per word,
 *   the row-to-row differences of the global alignment matrix with (match, mismatch, gap) = (0,-1,-1) resp. (+1,-4,-2)
 *   -- characters indexed as raw bytes (bitpal.m0.x1.g1.c:136-146), first row and column = multiples of the gap
 *   (:158-166, score = -m at :259) -- and sum the last row's differences (:257-270).  The value is the
 *   Needleman-Wunsch score S[m][n]; it is restated here as the plain recurrence
 *       S[i][j] = max(S[i-1][j-1] + (a_j == b_i ? match : mismatch), S[i-1][j] + gap, S[i][j-1] + gap).
 * PINNED: tests/golden/bpm_{bench,adv}.bitpal_{edit,scored}.expected.txt are outputs of the compiled reference
 * (oracle/_ref/bpm_ref -a bitpal-edit / bitpal-scored), see tests/golden/make_golden.py.
 */
#include "oracle.h"
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_bitpal_one(int algorithm, const char *a, int n, const char *b, int m) {
    const int match = algorithm == 0 ? 0 : 1, mismatch = algorithm == 0 ? -1 : -4, gap = algorithm == 0 ? -1 : -2;
    int32_t *row = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    for (int j = 0; j <= n; j++) row[j] = j * gap;
    for (int i = 1; i <= m; i++) {
        int diag = row[0];
        row[0] = i * gap;
        for (int j = 1; j <= n; j++) {
            const int up = row[j];
            int best = diag + (a[j - 1] == b[i - 1] ? match : mismatch);
            if (up + gap > best) best = up + gap;
            if (row[j - 1] + gap > best) best = row[j - 1] + gap;
            row[j] = best;
            diag = up;
        }
    }
    const int s = row[n];
    free(row);
    return s;
}

void oracle_bitpal_batch(int algorithm, const char *pat, const int64_t *pat_off, const int32_t *pat_len, const char *txt,
                         const int64_t *txt_off, const int32_t *txt_len, int64_t n, int threads, int32_t *score) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; i++)
        score[i] = oracle_bitpal_one(algorithm, pat + pat_off[i], pat_len[i], txt + txt_off[i], txt_len[i]);
}
