/* Seeded synthetic-input generators for the six hot-path benchmarks.
 *
 * Test/bench infrastructure, not product code.  Every item is generated from
 * its own splitmix64 stream keyed by (seed, item index), so the output is
 * byte-identical on every machine, independent of thread count and of how the
 * item range is split across ranks (bench.py --gpus N shards by index range).
 *
 * Distributions follow SURVEY.md section 8(d).
 */
#ifndef GABGEN_H
#define GABGEN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- bsw ---------------------------------------------------------------
 * mode 0 ("bench"): query len U[20,131]; ref = query with 5% subst, 1% ins,
 *   1% del, followed by a U[0,100] random tail; 1% of bases are N (code 4);
 *   h0 U[19,100].
 * mode 1 ("adversarial"): 3% N in both strings, indel bursts up to 12,
 *   tails up to 300, h0 in {0,1,5,19,50,200,1000}, query 1..250, ref 1..550.
 * Pass 1: gab_gen_bsw_lens fills len1 (ref) / len2 (query) / h0 for items
 *   [first, first+n).  Pass 2: gab_gen_bsw_fill writes the bases (codes 0..4)
 *   at the caller-computed byte offsets. */
void gab_gen_bsw_lens(uint64_t seed, int mode, int64_t first, int64_t n,
                      int32_t *len1, int32_t *len2, int32_t *h0);
void gab_gen_bsw_fill(uint64_t seed, int mode, int64_t first, int64_t n,
                      uint8_t *ref, const int64_t *ref_off,
                      uint8_t *qry, const int64_t *qry_off);
/* Text file in the reference's input format (bsw/src/main_banded.cpp:152-206). */
int gab_gen_bsw_write(const char *path, uint64_t seed, int mode, int64_t n);

/* ---- bpm / wfa ('>' pattern / '<' text line pairs) -----------------------
 * mode 0: pattern length plen (fixed), text = pattern with 2% subst,
 *   0.5% ins, 0.5% del; 0.1% N.   mode 1: lengths U[1,plen], 2% N, 5%/2%/2%
 *   errors, lower-case sprinkled in (bpm raw-byte compare quirk).
 * Sequences are ASCII.  Offsets are computed by the caller from the lengths. */
void gab_gen_pairs_lens(uint64_t seed, int mode, int plen, int64_t first, int64_t n,
                        int32_t *pat_len, int32_t *txt_len);
void gab_gen_pairs_fill(uint64_t seed, int mode, int plen, int64_t first, int64_t n,
                        char *pat, const int64_t *pat_off,
                        char *txt, const int64_t *txt_off);
int gab_gen_pairs_write(const char *path, uint64_t seed, int mode, int plen, int64_t n);

/* ---- chain / fast-chain ---------------------------------------------------
 * mode 0: anchors per call log-uniform [nmin,nmax]; 70% of anchors on 1-5
 *   co-linear diagonals with jitter <= 40, rest uniform; q_span 15 (90%) else
 *   U[1,30]; header max_dist_x = max_dist_y = 5000, bw = 500, n_segs = 1.
 * mode 1: dense (small coordinate range -> long predecessor windows, triggers
 *   max_skip), integer avg_qspan, multi-segment ids, x above 2^32. */
typedef struct {
    int64_t n;
    float avg_qspan;
    int32_t max_dist_x, max_dist_y, bw, n_segs;
} gabgen_chain_hdr;
void gab_gen_chain_hdrs(uint64_t seed, int mode, int64_t nmin, int64_t nmax,
                        int64_t first, int64_t ncalls, gabgen_chain_hdr *hdr);
void gab_gen_chain_fill(uint64_t seed, int mode, int64_t nmin, int64_t nmax,
                        int64_t first, int64_t ncalls, const gabgen_chain_hdr *hdr,
                        const int64_t *call_off, uint64_t *x, uint64_t *y);
int gab_gen_chain_write(const char *path, uint64_t seed, int mode, int64_t nmin,
                        int64_t nmax, int64_t ncalls);

/* ---- fmi -------------------------------------------------------------------
 * Reference: ref_len random bases (codes 0..3, N-free) with `rep_pct` percent
 * of the sequence overwritten by planted copies of 300-bp repeats.
 * Reads: len U[rl_min,rl_max] from a random position, 50% reverse strand,
 * 2% subst, 0.2% N; encoded 0..4 in a [n x stride] byte matrix. */
void gab_gen_fmi_ref(uint64_t seed, int64_t ref_len, int rep_pct, uint8_t *ref);
void gab_gen_fmi_reads(uint64_t seed, const uint8_t *ref, int64_t ref_len,
                       int rl_min, int rl_max, int64_t first, int64_t n,
                       uint8_t *enc, int32_t stride, int32_t *len);
int gab_gen_fmi_write_fasta(const char *path, const uint8_t *ref, int64_t ref_len);
int gab_gen_fmi_write_fastq(const char *path, const uint8_t *enc, int32_t stride,
                            const int32_t *len, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
