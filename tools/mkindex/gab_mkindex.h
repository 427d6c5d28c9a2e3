/* FM-index builder producing the BWA-MEM2 ".bwt.2bit.64" layout that the fmi benchmark loads
 * (writer: /root/reference/benchmarks/fmi/bwa-mem2/x86_64/src/FMI_search.cpp:144-304,306-382;
 *  loader: :384-494).  Needed because the reference's dataset (the `broad` human index) is not
 * available: bench.py and the tests index a synthetic, N-free reference with this tool.
 * The suffix array is built with an own SA-IS implementation (the reference vendors sais.h).
 * Limit: 2 * ref_len + 1 < 2^31 (32-bit suffix array).  Bigger genomes: use the files written by
 * the reference's own `bwa-mem2 index`, which gab_fmi_load reads unchanged.
 */
#ifndef GAB_MKINDEX_H
#define GAB_MKINDEX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int64_t cp_count[4];
    uint64_t one_hot_bwt_str[4];
} gab_cp_occ;   /* CP_OCC, FMI_search.h:54-58 */

typedef struct {
    int64_t ref_seq_len;     /* 2 * ref_len + 1 (as stored in the file) */
    int64_t count[5];        /* as stored in the file (the loader adds 1 to each) */
    int64_t cp_occ_size;     /* (ref_seq_len >> 6) + 1 */
    gab_cp_occ *cp_occ;
    int64_t sentinel_index;
    int64_t n_sa;            /* (ref_seq_len >> 3) + 1 sampled suffix-array entries */
    int8_t *sa_ms_byte;
    uint32_t *sa_ls_word;
} gab_fmindex;

/* fwd: ref_len base codes 0..3.  Returns 0 on success. */
int gab_mkindex_build(const uint8_t *fwd, int64_t ref_len, gab_fmindex *out);
/* index of the reference (U . revcomp(U))^m without sorting the whole text (see gab_mkindex.c): any size, e.g. >= 2^32 rows */
int gab_mkindex_build_power(const uint8_t *U, int64_t ulen, int64_t m, gab_fmindex *out);
void gab_mkindex_free(gab_fmindex *idx);
/* writes <prefix>.bwt.2bit.64 */
int gab_mkindex_write(const gab_fmindex *idx, const char *prefix);
/* writes <prefix>.ann / .amb / .pac so that the reference's own loader accepts the index too */
int gab_mkindex_write_bns(const char *prefix, const uint8_t *fwd, int64_t ref_len, const char *seq_name);
/* suffix array of s[0..n) (symbols < K, s[n-1] must be the unique smallest symbol) */
int gab_sais_i32(const int32_t *s, int32_t *SA, int32_t n, int32_t K);
int gab_sais_u8(const uint8_t *s, int32_t *SA, int32_t n, int32_t K);

#ifdef __cplusplus
}
#endif
#endif
