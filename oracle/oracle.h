/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatements of the reference algorithms on the hot path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
 * call anything in this directory; the product (libgab_hip.so and the drivers
 * under benchmarks/) never does.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against
 * (a) golden vectors under tests/golden/ produced by the reference itself,
 * compiled from /root/reference by oracle/Makefile into oracle/_ref/, and
 * (b) that compiled reference run live when oracle/_ref/ is present.
 */
#ifndef GAB_ORACLE_H
#define GAB_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- bsw: follows BandedPairWiseSW::scalarBandedSWA,
 *      /root/reference/benchmarks/bsw/src/bandedSWA.cpp:132-253 ------------- */
typedef struct {
    int32_t o_del, e_del, o_ins, e_ins, zdrop, end_bonus, w;
    int8_t mat[25];
} oracle_bsw_params;
typedef struct {
    int32_t score, qle, tle, gtle, gscore, max_off;
} oracle_bsw_result;
/* fills the 5x5 matrix the way bsw/src/main_banded.cpp:94-102 does */
void oracle_bsw_fill_scmat(int a, int b, int ambig, int8_t mat[25]);
void oracle_bsw_one(const oracle_bsw_params *p, int qlen, const uint8_t *query,
                    int tlen, const uint8_t *target, int h0, oracle_bsw_result *out);
/* batch over n pairs, OpenMP over pairs; also returns the number of DP cells
 * evaluated (sum over rows of end-beg) when cells != NULL */
void oracle_bsw_batch(const oracle_bsw_params *p, const uint8_t *ref, const int64_t *ref_off,
                      const uint8_t *qry, const int64_t *qry_off, const int32_t *len1,
                      const int32_t *len2, const int32_t *h0, int64_t n, int threads,
                      oracle_bsw_result *out, int64_t *cells);

/* ---- chain / fast-chain: see chain.c ---------------------------------------------- */
typedef struct {
    int64_t n;
    float avg_qspan;
    int32_t max_dist_x, max_dist_y, bw, n_segs;
} oracle_chain_hdr;
/* both return the number of predecessor evaluations (inner-loop iterations) */
int64_t oracle_chain_call(const oracle_chain_hdr *h, const uint64_t *x, const uint64_t *y,
                          int32_t *score, int32_t *parent);
int64_t oracle_fastchain_call(const oracle_chain_hdr *h, const uint64_t *x, const uint64_t *y,
                              int32_t *score, int32_t *parent);
/* mode 0 = chain, 1 = fast-chain; OpenMP dynamic over calls like host_chain_kernel
 * (chain/src/host_kernel.cpp:96-108) */
void oracle_chain_batch(int mode, const oracle_chain_hdr *hdr, const int64_t *call_off, int64_t ncalls,
                        const uint64_t *x, const uint64_t *y, int threads,
                        int32_t *score, int32_t *parent, int64_t *evals);

/* ---- bpm: see bpm.c.  The caller has applied the driver's "longer sequence is the pattern"
 * swap; returns the printed score (<= 0), INT32_MIN if text_length > pattern_length. */
int oracle_bpm_one(const char *pattern, int n, const char *text, int m, int64_t *block_steps);
void oracle_bpm_batch(const char *pat, const int64_t *pat_off, const int32_t *pat_len,
                      const char *txt, const int64_t *txt_off, const int32_t *txt_len,
                      int64_t n, int threads, int32_t *score, int64_t *block_steps);

/* ---- bitpal: see bitpal.c.  algorithm 0 = bitpal-edit (0,-1,-1), 1 = bitpal-scored (+1,-4,-2) */
int oracle_bitpal_one(int algorithm, const char *a, int n, const char *b, int m);
void oracle_bitpal_batch(int algorithm, const char *pat, const int64_t *pat_off, const int32_t *pat_len, const char *txt,
                         const int64_t *txt_off, const int32_t *txt_len, int64_t n, int threads, int32_t *score);

/* ---- wfa: see wfa.c.  ops = un-run-length-encoded CIGAR ('M','X','I','D'), capacity
 * pattern_length + text_length per pair; returns the number of operations. */
typedef struct {
    int32_t mismatch, gap_opening, gap_extension;
    int32_t min_wavefront_length, max_distance_threshold;   /* adaptive reduction; min_wavefront_length < 0 = complete mode */
} oracle_wfa_penalties;
int oracle_wfa_one(const oracle_wfa_penalties *pen, const char *pattern, int plen, const char *text, int tlen,
                   char *ops_out, int *score_out, int64_t *cells);
void oracle_wfa_batch(const oracle_wfa_penalties *pen, const char *pat, const int64_t *pat_off, const int32_t *pat_len,
                      const char *txt, const int64_t *txt_off, const int32_t *txt_len, int64_t n, int threads,
                      char *ops, const int64_t *ops_off, int32_t *ops_len, int32_t *score, int64_t *cells);

/* ---- fmi: see fmi.c ------------------------------------------------------------------ */
typedef struct { int64_t cp_count[4]; uint64_t one_hot_bwt_str[4]; } oracle_cp_occ;   /* CP_OCC */
typedef struct {
    int64_t ref_seq_len, count[5] /* already +1 */, cp_occ_size, sentinel_index;
    oracle_cp_occ *cp_occ;
    int8_t *sa_ms_byte;      /* sampled suffix array (SA_COMPX = 3: one entry per 8 BWT rows), NULL if not loaded */
    uint32_t *sa_ls_word;
} oracle_fmindex;
typedef struct { uint32_t rid, m, n, pad; int64_t k, l, s; } oracle_smem;              /* SMEM, 40 bytes */
int oracle_fmi_load(const char *prefix, oracle_fmindex *idx);                          /* <prefix>.bwt.2bit.64 */
void oracle_fmi_from_arrays(oracle_fmindex *idx, int64_t ref_seq_len, const int64_t *file_count,
                            const void *cp_occ, int64_t sentinel_index);
void oracle_fmi_free(oracle_fmindex *idx);
int64_t oracle_fmi_read(const oracle_fmindex *idx, const uint8_t *q, int len, uint32_t rid, int min_seed_len,
                        oracle_smem *out, int64_t *ext_calls);
/* all reads; *out_p receives a malloc'ed array sorted by (rid, m, n desc); read_off[nreads+1] */
int64_t oracle_fmi_batch(const oracle_fmindex *idx, const uint8_t *enc, int32_t stride, const int32_t *len,
                         int64_t nreads, int min_seed_len, int threads, oracle_smem **out_p, int64_t *read_off,
                         int64_t *ext_calls);
void oracle_fmi_release(oracle_smem *p);
/* suffix-array look-up of SMEM intervals (FMI_search.cpp:1103-1196): for SMEM i the rows k, k+step, ... (< k+s, at
 * most max_occ of them, step = s > max_occ ? s / max_occ : 1) are resolved with get_sa_entry_compressed; coords are
 * written back to back, coord_off[n+1] delimits them.  Returns the total, or -1 if the index has no SA arrays.
 * lf_steps (optional) receives the number of LF-mapping steps taken. */
void oracle_fmi_set_sa(oracle_fmindex *idx, const int8_t *sa_ms_byte, const uint32_t *sa_ls_word);
int64_t oracle_fmi_sa_count(const oracle_smem *smems, int64_t n, int32_t max_occ, int64_t *coord_off);
int64_t oracle_fmi_sa_lookup(const oracle_fmindex *idx, const oracle_smem *smems, int64_t n, int32_t max_occ,
                             const int64_t *coord_off, int64_t *coords, int64_t *lf_steps);

#ifdef __cplusplus
}
#endif
#endif
