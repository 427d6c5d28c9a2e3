/* gab.h -- C ABI of libgab_hip.so: the MI355X (gfx950) engine for the GenArchBench
 * banded-DP / seed-chaining hot path.
 *
 * The reference has no plugin/FFI layer: each kernel library is linked straight into
 * its benchmark's main().  Every entry point below therefore replaces one *call site*
 * of a reference driver (cited per function, paths relative to
 * /root/reference/benchmarks/).  INTEGRATION.md shows the few lines a maintainer
 * changes in each driver to call this library instead.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success or a negative
 *     GAB_E* code and never calls exit(); gab_last_error() gives the message for the
 *     calling thread.
 *   - the caller owns every buffer.  "_run" variants take HOST pointers and do
 *     H2D + kernels + D2H synchronously (on a private non-blocking stream of the handle, so
 *     that calls on several handles of one GPU overlap); "_run_device" variants take DEVICE pointers
 *     (hipMalloc'ed or a torch tensor's data_ptr) and enqueue on `stream` (a hipStream_t passed
 *     as void*, NULL = the default stream); each entry says where it synchronises `stream`
 *     internally.  Some of them (bsw, bpm) run part of their
 *     kernels on an internal second stream; it is forked from and joined back into `stream` by events
 *     inside the call, so the caller only ever has to order against `stream`.
 *   - a handle is bound to one GPU; calls on distinct handles are thread-safe, so the
 *     multi-GPU drivers run one host thread (or one process) per GPU with no collective.
 */
#ifndef GAB_H
#define GAB_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAB_OK 0
#define GAB_EINVAL (-22)   /* bad argument / unsupported size  */
#define GAB_ENOMEM (-12)   /* host or device allocation failed */
#define GAB_EDEVICE (-5)   /* HIP runtime error                */
#define GAB_ENODEV (-19)   /* no usable gfx950 device          */
#define GAB_ERANGE (-34)   /* result larger than the caller's buffer (the needed size is returned) */

/* ---- library ------------------------------------------------------------------- */
const char *gab_version(void);
const char *gab_last_error(void);
int gab_device_count(void);
/* PCI bus id ("0000:c1:00.0") of a device into buf (len >= 16): what the drivers need to find the NUMA node a card hangs
 * off (/sys/bus/pci/devices/<id>/numa_node) and place its worker threads there, as the reference places its threads with
 * OMP_PROC_BIND / OMP_PLACES (bsw/scripts/regression_small.sh:52) */
int gab_device_pci_bus_id(int device, char *buf, int len);
/* device memory for C callers that chain two *_device entry points (e.g. parser -> kernel) without a HIP toolchain */
int gab_device_alloc(int device, size_t bytes, void **out);
void gab_device_free(int device, void *p);
int gab_device_copy_to_host(int device, void *dst, const void *d_src, size_t bytes);
/* pinned (page-locked) host memory for the slabs handed to the host-pointer entry points gab_*_run: where the reference
 * drivers allocate theirs before the region of interest (_mm_malloc, bsw/src/main_banded.cpp:260-264; malloc / std::vector
 * elsewhere).  gab_*_run accepts any host memory; from pinned memory its chunked copies are direct DMA that overlaps the
 * kernels, from pageable memory the runtime stages them at a fraction of the link rate. */
int gab_host_alloc(size_t bytes, void **out);
void gab_host_free(void *p);
/* same, in place, for memory the caller already owns (malloc / realloc / std::vector storage): page-locks
 * [p, p + bytes) until gab_host_unregister(p); do it after the buffer has reached its final size, outside the ROI */
int gab_host_register(void *p, size_t bytes);
void gab_host_unregister(void *p);

/* ---- bsw: banded Smith-Waterman seed extension ------------------------------------
 * Replaces  bsw[tid]->getScores16(SeqPair*, ref, qer, nPairsBatch, 1, w)
 *           bsw/src/main_banded.cpp:338-350  (class: bsw/src/bandedSWA.h:240-245;
 *           semantics: scalarBandedSWA, bsw/src/bandedSWA.cpp:132-253).
 * Instead of T threads x batches of B pairs, the GPU driver makes ONE call over all
 * pairs inside the same region of interest.
 */
typedef struct gab_bsw gab_bsw;
typedef struct {
    int32_t o_del, e_del, o_ins, e_ins; /* BandedPairWiseSW ctor, bandedSWA.cpp:48-66    */
    int32_t zdrop, end_bonus;           /* main_banded.cpp:268                            */
    int32_t w;                          /* band width passed to getScores16 (100)         */
    int8_t mat[25];                     /* 5x5 scores, codes 0..3 = ACGT, 4 = N           */
} gab_bsw_params;

/* Limits (same as the reference driver's slabs, main_banded.cpp:76-79): query length
 * 1..256, reference length 1..32767, 0 <= h0, and h0 + qlen * max(mat) <= 2^30. */
#define GAB_BSW_MAX_QLEN 256
#define GAB_BSW_MAX_TLEN 32767

/* full extension result, as scalarBandedSWA returns it (bandedSWA.cpp:241-252) */
typedef struct {
    int32_t score, qle, tle, gtle, gscore, max_off;
} gab_bsw_result;

int gab_bsw_create(const gab_bsw_params *params, int device, gab_bsw **out);
void gab_bsw_destroy(gab_bsw *h);

/* optional: size the handle's device buffers now for calls of up to max_pairs pairs whose referenced windows of the two
 * slabs are at most max_ref_bytes / max_qry_bytes, so that the first gab_bsw_run of a timed region does not allocate
 * (the reference allocates its working buffers in the BandedPairWiseSW constructor, bsw/src/bandedSWA.cpp:80-96) */
int gab_bsw_reserve(gab_bsw *h, int64_t max_pairs, int64_t max_ref_bytes, int64_t max_qry_bytes);

/* Host buffers.  Pair i: reference (target) = ref[ref_off[i] .. +len1[i]), query =
 * qry[qry_off[i] .. +len2[i]), base codes 0..4 one byte each (what loadPairs produces,
 * main_banded.cpp:164-206), seed score h0[i].  score_out[i] = SeqPair.score. */
int gab_bsw_run(gab_bsw *h, const uint8_t *ref, const int64_t *ref_off, const uint8_t *qry,
                const int64_t *qry_off, const int32_t *len1, const int32_t *len2,
                const int32_t *h0, int64_t n, int32_t *score_out);

/* Device buffers.  Enqueues on `stream`, but synchronises it once mid-way (the launch geometry of the DP kernels needs
 * the sizes of the query-length classes on the host); the DP launches and the result stores that follow are asynchronous:
 * order later work against `stream` (or synchronise it) before reading score_out.  ref_bytes / qry_bytes = sizes of the two
 * sequence slabs, used for bounds validation: the kernels read a sequence four bytes at a time from its own (possibly
 * unaligned) start, so every sequence must be followed by at least 3 more bytes INSIDE its slab (off + len + 3 <= bytes;
 * their contents do not matter).  result_out may be NULL;
 * when given it receives all six result fields per pair. */
int gab_bsw_run_device(gab_bsw *h, const uint8_t *ref, int64_t ref_bytes, const int64_t *ref_off,
                       const uint8_t *qry, int64_t qry_bytes, const int64_t *qry_off,
                       const int32_t *len1, const int32_t *len2, const int32_t *h0, int64_t n,
                       int32_t *score_out, gab_bsw_result *result_out, void *stream);

/* Counters of the last run on this handle (for the roofline report): number of DP
 * cells evaluated (sum over rows of band width) and device time of the dominant kernel
 * as measured with HIP events on the run's stream (ms). */
int gab_bsw_last_stats(gab_bsw *h, int64_t *cells, float *kernel_ms, float *total_ms);

/* ---- chain / fast-chain: minimap2 seed chaining ------------------------------------
 * Replaces  host_chain_kernel(std::vector<call_t>&, std::vector<return_t>&, numThreads)
 *           chain/src/main.cpp:154 and fast-chain/src/main.cpp:154 (declared in each src/host_kernel.h:6;
 *           kernels chain/src/host_kernel.cpp:30-108, fast-chain/src/host_kernel.cpp:135-865).
 * The reference call is already a whole-input batch call; so is this one.
 * Anchors of all calls lie back to back: call c owns x/y/score/parent[call_off[c] ..
 * call_off[c] + hdr[c].n).  x, y are minimap2's mm128_t words (chain/src/host_data.h:18-21):
 * x = ref id/strand/pos (ascending), y = seg_id << 48 | q_span << 32 | query_pos.
 * Outputs are return_t.scores / return_t.parents (host_data.h:30-36).
 */
#define GAB_CHAIN 0      /* chain:      max_skip / max_iter heuristics, 64-bit coordinates   */
#define GAB_FASTCHAIN 1  /* fast-chain: no max_skip, 32-bit coordinates, AVX2/AVX-512 rounding */
typedef struct gab_chain gab_chain;
typedef struct {           /* call_t header, chain/src/host_data.h:23-28 */
    int64_t n;
    float avg_qspan;
    int32_t max_dist_x, max_dist_y, bw, n_segs;
} gab_chain_hdr;

int gab_chain_create(int device, gab_chain **out);
void gab_chain_destroy(gab_chain *h);
/* optional: size the handle's device buffers now for calls of up to max_anchors anchors in max_calls calls (see gab_bsw_reserve) */
int gab_chain_reserve(gab_chain *h, int64_t max_anchors, int64_t max_calls);
/* gab_chain_reserve + what gab_chain_run_device[_through] of `mode` needs before a timed region: the first launches of its kernels
 * and, for GAB_FASTCHAIN (whose longer calls run in the table form, chain_tab.hip), the table for max_anchors anchors */
int gab_chain_reserve_mode(gab_chain *h, int mode, int64_t max_anchors, int64_t max_calls);
/* everything on the host */
int gab_chain_run(gab_chain *h, int mode, const uint64_t *x, const uint64_t *y, const int64_t *call_off,
                  const gab_chain_hdr *hdr, int64_t ncalls, int32_t *score_out, int32_t *parent_out);
/* anchors and results on the device; the small call table (call_off, hdr) stays on the host.
 * Returns after the kernel has completed on `stream`. */
int gab_chain_run_device(gab_chain *h, int mode, const uint64_t *d_x, const uint64_t *d_y,
                         const int64_t *call_off, const gab_chain_hdr *hdr, int64_t ncalls,
                         int32_t *d_score, int32_t *d_parent, void *stream);
/* The same, with the results in HOST memory as well when the call returns (a driver whose read phase parsed the file on the GPU,
 * gab_chain_parse): d_score / d_parent are filled as always, host_score / host_parent (page-locked for the fast way: the DP kernel
 * then writes every block of results through to them while it runs; pageable or small batches: copied at the end) receive them too. */
int gab_chain_run_device_through(gab_chain *h, int mode, const uint64_t *d_x, const uint64_t *d_y,
                                 const int64_t *call_off, const gab_chain_hdr *hdr, int64_t ncalls,
                                 int32_t *d_score, int32_t *d_parent, int32_t *host_score, int32_t *host_parent, void *stream);
/* predecessor evaluations performed (chain: the whole window of an anchor resolved by the plain-maximum path, plus the
 * reference's own visits for the anchors that needed its max_skip scan; fast-chain: the windows) and kernel time (HIP
 * events) of the last run */
int gab_chain_last_stats(gab_chain *h, int64_t *evals, float *kernel_ms);

/* ---- bpm: bit-parallel Myers edit distance (+ backtrace-derived score) -----------------
 * Replaces  it->score = benchmark_edit_bpm(&align_input)   bpm/tools/align_benchmark.c:243-257
 *           (bpm/benchmark/benchmark_edit.c:31-56; kernel bpm/edit/edit_bpm.c:70-331).
 * One call over all pairs instead of one call per pair.  The caller has already applied the
 * driver's swap (the longer line is the pattern, align_benchmark.c:177-181) and stripped the
 * leading '>' / '<' and the newline (:247-252): text_length <= pattern_length is required,
 * exactly the condition under which the reference never cuts a block off.
 * Sequences are raw ASCII bytes: pair i = pat[pat_off[i] .. +pat_len[i]), txt[...].
 * score_out[i] = the value the reference prints as "[i] score=%d" (<= 0).
 */
#define GAB_BPM_MAX_PLEN 16320 /* 255 x 64: top_level is a uint8_t, bpm/edit/edit_bpm.c:205-206 */
typedef struct gab_bpm gab_bpm;
int gab_bpm_create(int device, gab_bpm **out);
void gab_bpm_destroy(gab_bpm *h);
/* optional, outside the timed region: device buffers for calls of up to max_pairs pairs whose sequences span up to
 * max_seq_bytes of the slab(s), and warm copy queues (see gab_bsw_reserve).  pat and txt may be the same slab (the drivers'
 * pair files): its window is then staged once. */
int gab_bpm_reserve(gab_bpm *h, int64_t max_pairs, int64_t max_seq_bytes);
int gab_bpm_run(gab_bpm *h, const char *pat, const int64_t *pat_off, const int32_t *pat_len,
                const char *txt, const int64_t *txt_off, const int32_t *txt_len, int64_t n,
                int32_t *score_out);
/* device buffers; pat_bytes / txt_bytes = slab sizes: every sequence must be followed by at least 3 more bytes
 * inside its slab (off + len + 3 <= bytes; the kernels read four bytes at a time from the sequence's own start).
 * Synchronises `stream` internally (the second kernel's launch
 * geometry depends on the first one's queue length). */
int gab_bpm_run_device(gab_bpm *h, const char *pat, int64_t pat_bytes, const int64_t *pat_off,
                       const int32_t *pat_len, const char *txt, int64_t txt_bytes,
                       const int64_t *txt_off, const int32_t *txt_len, int64_t n,
                       int32_t *score_out, void *stream);
/* last run: 64-row block steps executed, pairs that needed the history + backtrace path,
 * device time of the score kernels together with the first backtrace stage that runs underneath them, and of the
 * whole call (HIP events, ms) */
int gab_bpm_last_stats(gab_bpm *h, int64_t *block_steps, int64_t *full_pairs,
                       float *score_kernel_ms, float *total_ms);

/* ---- bitpal: the bpm driver's BitPAl algorithms (global alignment score, no CIGAR) -----------
 * Replaces  it->score = benchmark_bitpal_m0_x1_g1(&align_input)   (-a bitpal-edit)
 *           it->score = benchmark_bitpal_m1_x4_g2(&align_input)   (-a bitpal-scored)
 *                                                    bpm/tools/align_benchmark.c:259-264, 330-338
 *           (bpm/benchmark/benchmark_bitpal.c:30-54; generated kernels bpm/bitpal/bitpal.m0.x1.g1.c,
 *            bpm/bitpal/bitpal.m1.x4.g2.c).
 * score_out[i] = the Needleman-Wunsch score of pair i: global, linear gaps, raw-byte comparison, with
 * (match, mismatch, gap) = (0, -1, -1) for GAB_BITPAL_EDIT and (+1, -4, -2) for GAB_BITPAL_SCORED --
 * the value the reference prints as "[i] score=%d".  Same buffers as gab_bpm_run; the score is
 * symmetric in the two strings, so the driver's swap is immaterial and not required.
 */
#define GAB_BITPAL_EDIT 0
#define GAB_BITPAL_SCORED 1
#define GAB_BITPAL_MAX_LEN 16320 /* kept equal to GAB_BPM_MAX_PLEN: the two algorithms share the driver and its inputs */
typedef struct gab_bitpal gab_bitpal;
int gab_bitpal_create(int algorithm, int device, gab_bitpal **out);
void gab_bitpal_destroy(gab_bitpal *h);
int gab_bitpal_reserve(gab_bitpal *h, int64_t max_pairs, int64_t max_seq_bytes);      /* see gab_bpm_reserve */
int gab_bitpal_run(gab_bitpal *h, const char *pat, const int64_t *pat_off, const int32_t *pat_len,
                   const char *txt, const int64_t *txt_off, const int32_t *txt_len, int64_t n,
                   int32_t *score_out);
/* device buffers; slabs readable to a multiple of 4 bytes past the last sequence; synchronises `stream` */
int gab_bitpal_run_device(gab_bitpal *h, const char *pat, int64_t pat_bytes, const int64_t *pat_off,
                          const int32_t *pat_len, const char *txt, int64_t txt_bytes,
                          const int64_t *txt_off, const int32_t *txt_len, int64_t n,
                          int32_t *score_out, void *stream);
/* last run: DP cells (pattern_length x text_length summed), pairs that took the global-scratch path,
 * device time of the DP kernels and of the whole call (HIP events, ms) */
int gab_bitpal_last_stats(gab_bitpal *h, int64_t *cells, int64_t *long_pairs, float *kernel_ms, float *total_ms);

/* ---- wfa: gap-affine wavefront alignment with CIGAR ---------------------------------------
 * Replaces  affine_wavefronts_clear(wf); affine_wavefronts_align(wf, pattern, plen, text, tlen);
 *           + the copy of wf->edit_cigar            wfa/tools/align_benchmark.c:415-437
 *           (kernel: wfa/gap_affine/affine_wavefront_align.c:325-361 and the files it calls).
 * One call over all pairs.  No swap: the '>' line is the pattern, the '<' line the text
 * (align_benchmark.c:152-160).  gab_wfa_create = complete mode (affine_wavefronts_new_complete, the
 * driver's default min_wavefront_length = -1, align_benchmark.c:91,359-363); gab_wfa_create_reduced =
 * the adaptive reduction (affine_wavefronts_new_reduced, wfa/gap_affine/affine_wavefront.c:162-181,
 * driver flags --minimum-wavefront-length / --maximum-difference-distance, align_benchmark.c:267-272,
 * 364-368; the heuristic of affine_wavefronts_reduce_wavefronts, affine_wavefront_extend.c:85-154).
 * min_wavefront_length < 0 selects the complete mode, as in the driver.
 * ops_out receives, for pair i, ops_len_out[i] operations 'M','X','I','D' starting at
 * ops_out[ops_off[i]]; the caller provides pattern_length + text_length bytes of room per pair
 * (edit_cigar_allocate, wfa/gap_affine/edit_cigar.c:38-47) and run-length encodes them when
 * printing (edit_cigar_print, :184-200).  score_out[i] = the alignment penalty.
 * Bytes equal to the reference's padding characters -- 'Y' in a pattern, 'X' in a text -- match the OTHER sequence's
 * padding as they do in the reference (wfa/utils/string_padded.c:88-117), so an alignment can run past the end of a
 * sequence.  Where that makes the CIGAR longer than pattern_length + text_length the reference overflows its buffer;
 * this library keeps the last pattern_length + text_length operations (never written outside the pair's room).
 */
#define GAB_WFA_MAX_LEN 100000 /* MAX_SEQUENCE_LENGTH, wfa/tools/align_benchmark.c:62 */
typedef struct gab_wfa gab_wfa;
typedef struct {
    int32_t mismatch, gap_opening, gap_extension; /* defaults 4, 6, 2 (align_benchmark.c:85-90); match = 0 */
} gab_wfa_penalties;
int gab_wfa_create(const gab_wfa_penalties *penalties, int device, gab_wfa **out);
int gab_wfa_create_reduced(const gab_wfa_penalties *penalties, int min_wavefront_length, int max_distance_threshold,
                           int device, gab_wfa **out);
void gab_wfa_destroy(gab_wfa *h);
/* see gab_bpm_reserve; max_ops_bytes = the room of the CIGAR operations of a call (pattern + text length per pair) */
int gab_wfa_reserve(gab_wfa *h, int64_t max_pairs, int64_t max_seq_bytes, int64_t max_ops_bytes);
int gab_wfa_run(gab_wfa *h, const char *pat, const int64_t *pat_off, const int32_t *pat_len,
                const char *txt, const int64_t *txt_off, const int32_t *txt_len, int64_t n,
                char *ops_out, const int64_t *ops_off, int32_t *ops_len_out, int32_t *score_out);
/* The same call for a driver that PRINTS the alignments (wfa/tools/align_benchmark.c:499-504): the result is the text
 * edit_cigar_print writes (wfa/gap_affine/edit_cigar.c:184-200) -- every run of equal operations as "%d%c", e.g. "70M1X80M" --
 * encoded on the device, so that ~20 bytes per 151-bp pair cross the bus instead of pattern_length + text_length bytes of
 * operation room.  Pair i's text is cigar_out[cigar_off_out[i] .. + cigar_len_out[i]) (no terminator; an empty CIGAR has
 * length 0; the texts are packed without gaps but NOT in pair order).  *cigar_bytes = bytes of text produced; when that exceeds
 * `capacity` no text is written and the call returns GAB_ERANGE (offsets, lengths and scores are valid): call again with
 * *cigar_bytes of room.  2 * (pattern_length + text_length) per pair always suffices; a quarter of the operation room is
 * ample for reads. */
int gab_wfa_run_packed(gab_wfa *h, const char *pat, const int64_t *pat_off, const int32_t *pat_len,
                       const char *txt, const int64_t *txt_off, const int32_t *txt_len, int64_t n,
                       char *cigar_out, int64_t capacity, int64_t *cigar_off_out, int32_t *cigar_len_out,
                       int32_t *score_out, int64_t *cigar_bytes);
/* The same for pairs that are already on the device (the read phase parsed the file there: gab_pairs_parse): DEVICE pointers in as
 * for gab_wfa_run_device -- `ops` / `ops_off` is operation room on the device, pattern_length + text_length bytes per pair, scratch
 * of the caller -- and the printed text, offsets, lengths and scores out to HOST memory as above (GAB_ERANGE likewise). */
int gab_wfa_run_packed_device(gab_wfa *h, const char *pat, int64_t pat_bytes, const int64_t *pat_off, const int32_t *pat_len,
                              const char *txt, int64_t txt_bytes, const int64_t *txt_off, const int32_t *txt_len, int64_t n,
                              char *ops, const int64_t *ops_off, char *cigar_out, int64_t capacity, int64_t *cigar_off_out,
                              int32_t *cigar_len_out, int32_t *score_out, int64_t *cigar_bytes);
/* device buffers (sequence slabs readable to a multiple of 4 bytes past the last base);
 * synchronises `stream` internally between its passes */
int gab_wfa_run_device(gab_wfa *h, const char *pat, int64_t pat_bytes, const int64_t *pat_off,
                       const int32_t *pat_len, const char *txt, int64_t txt_bytes,
                       const int64_t *txt_off, const int32_t *txt_len, int64_t n, char *ops_out,
                       const int64_t *ops_off, int32_t *ops_len_out, int32_t *score_out, void *stream);
/* last run: wavefront cells computed + bases extended, pairs re-run with a larger history,
 * device time of the first (LDS) pass and of the whole call (HIP events, ms) */
int gab_wfa_last_stats(gab_wfa *h, int64_t *work, int64_t *requeued, float *first_pass_ms, float *total_ms);

/* ---- fmi: FM-index SMEM seeding --------------------------------------------------------------
 * Replaces, per batch of reads, the sequence
 *     getSMEMsAllPosOneThread -> select -> getSMEMsOnePosOneThread -> bwtSeedStrategyAllPosOneThread
 *     -> rid += batch offset -> sortSMEMs                          fmi/fmi.cpp:288-348
 * on a shared read-only FMI_search (fmi/bwa-mem2/x86_64/src/FMI_search.h:101-...), and
 *     new FMI_search(prefix); load_index()                         fmi/fmi.cpp:102-103
 * by gab_fmi_load (outside the region of interest, as in the reference).  The index file is the
 * reference's own <prefix>.bwt.2bit.64 (FMI_search.cpp:144-304), read unchanged; only the count[],
 * CP_OCC[] and sentinel_index parts are used by seeding.
 * Reads are the driver's dense code matrix (fmi.cpp:121-151): read r = enc[r*stride .. +len[r]),
 * codes 0..3 = ACGT, anything above 3 = N.  Constants of the driver are fixed: re-seed width 10,
 * split factor 1.5, third-pass interval 20 (fmi.cpp:162-164).
 * Output: SMEM records (FMI_search.h:75-83) of ALL reads sorted by (rid, m ascending, n descending)
 * -- the order the driver prints -- as [m, n] inclusive query interval and the bi-interval k, l, s.
 */
#define GAB_FMI_MAX_READLEN 10000 /* assert(max_readlength < 10000), fmi/fmi.cpp:117 */
typedef struct gab_fmi gab_fmi;
typedef struct {
    uint32_t rid, m, n, pad;
    int64_t k, l, s;
} gab_smem; /* 40 bytes, layout of SMEM */
int gab_fmi_load(int device, const char *prefix, gab_fmi **out);
/* same, from memory: the three parts of the file as the reference writes them (count[] NOT yet +1) */
int gab_fmi_create(int device, int64_t reference_seq_len, const int64_t count[5], const void *cp_occ,
                   int64_t sentinel_index, gab_fmi **out);
/* a second handle on the same GPU that SHARES the read-only index of `src` (which must outlive it; attach the suffix
 * array to src first) and owns only its work buffers: the drivers run several host threads per GPU, each with its own
 * handle, the way the reference's threads share one FMI_search (fmi/fmi.cpp:102-103,250-263) */
int gab_fmi_clone(gab_fmi *src, gab_fmi **out);
void gab_fmi_destroy(gab_fmi *h);
/* host buffers; *out is malloc'ed by the library, release it with gab_fmi_free */
int gab_fmi_seed(gab_fmi *h, const uint8_t *enc, int32_t stride, const int32_t *len, int64_t nreads,
                 int32_t min_seed_len, gab_smem **out, int64_t *nout);
void gab_fmi_free(gab_smem *p);
/* same, into the CALLER's array of `capacity` records -- how the reference's threads collect their SMEMs: per-thread arrays
 * allocated before the region of interest and grown when a batch does not fit (fmi/fmi.cpp:242, 253-257, 277-286).  Page-lock
 * the array (gab_host_alloc / gab_host_register) and the result arrives as one DMA at the link rate instead of through
 * the runtime's staging of pageable memory.  *nout = SMEMs found; when that exceeds `capacity` nothing is written and the
 * call returns GAB_ERANGE: grow the array to *nout and call again. */
int gab_fmi_seed_into(gab_fmi *h, const uint8_t *enc, int32_t stride, const int32_t *len, int64_t nreads,
                      int32_t min_seed_len, gab_smem *out, int64_t capacity, int64_t *nout);
/* optional: size the handle's device buffers now for host-pointer calls of up to max_reads reads of `stride` bases and push
 * an empty batch of that size through the whole path (see gab_bsw_reserve): a driver that times ONE pass (fmi/fmi.cpp:236-362)
 * would otherwise allocate gigabytes of slots and output inside its region of interest */
int gab_fmi_reserve(gab_fmi *h, int64_t max_reads, int32_t stride);
/* device buffers; *d_out / *d_read_off (nreads + 1 offsets into d_out) point into memory owned by the
 * handle and stay valid until the next call on it.  Synchronises `stream` internally. */
int gab_fmi_seed_device(gab_fmi *h, const uint8_t *d_enc, int32_t stride, const int32_t *d_len,
                        int64_t nreads, int32_t min_seed_len, const gab_smem **d_out,
                        const int64_t **d_read_off, int64_t *nout, void *stream);
/* last run: backwardExt calls, SMEMs found, seeding-kernel ms */
int gab_fmi_last_stats(gab_fmi *h, int64_t *ext_calls, int64_t *nsmem, float *kernel_ms);
/* last run: 64-byte CP_OCC records the extensions fetched (GET_OCC, FMI_search.h:66-73: one when both interval
 * ends share a record, else two) -- the random index traffic of the run is 64 B x this number */
int gab_fmi_last_records(gab_fmi *h, int64_t *cp_occ_records);

/* ---- fmi: suffix-array look-up -- the step BWA-MEM2 takes right after seeding:
 *     FMI_search::get_sa_entries(SMEM *smemArray, int64_t *coordArray, int32_t *coordCountArray, uint32_t count,
 *                                int32_t max_occ, int tid)          FMI_search.cpp:1177-1196
 *     -> get_sa_entry_compressed (SA_COMPX = 3)                     FMI_search.cpp:1103-1175
 * For SMEM i the BWT rows k, k+step, ... (< k+s, at most max_occ of them; step = s > max_occ ? s / max_occ : 1) are
 * resolved to reference coordinates: a row that is a multiple of 8 is stored, any other row walks the LF mapping (one
 * CP_OCC record per step) to a stored row or the sentinel.  Coordinates come back to back in SMEM order;
 * coord_off[n + 1] delimits them (the reference only returns the running total).
 * gab_fmi_load reads the sampled suffix array from <prefix>.bwt.2bit.64; a handle made by gab_fmi_create gets it
 * with gab_fmi_set_sa (arrays of (reference_seq_len >> 3) + 1 entries, FMI_search.cpp:439-447). */
int gab_fmi_set_sa(gab_fmi *h, const int8_t *sa_ms_byte, const uint32_t *sa_ls_word);
/* host buffers; *coords and *coord_off are malloc'ed by the library, release them with gab_fmi_free_coords */
int gab_fmi_sa_lookup(gab_fmi *h, const gab_smem *smems, int64_t n, int32_t max_occ, int64_t **coords,
                      int64_t **coord_off, int64_t *total);
void gab_fmi_free_coords(int64_t *p);
/* device buffers; the outputs point into memory owned by the handle, valid until the next call on it */
int gab_fmi_sa_lookup_device(gab_fmi *h, const gab_smem *d_smems, int64_t n, int32_t max_occ,
                             const int64_t **d_coords, const int64_t **d_coord_off, int64_t *total, void *stream);
/* last look-up: LF-mapping steps taken (one random 64-byte CP_OCC record each) and kernel ms */
int gab_fmi_last_sa_stats(gab_fmi *h, int64_t *lf_steps, float *kernel_ms);

/* ---- input parsers ---------------------------------------------------------
 * The reference drivers parse their text inputs on the host, line by line, outside the region of interest
 * (bsw: loadPairs, bsw/src/main_banded.cpp:164-206 -- fgets + sscanf per pair; bpm / wfa: getline per line,
 * bpm/tools/align_benchmark.c:150-200, wfa/tools/align_benchmark.c:110-194).  At 10 M items that is the bulk of the
 * end-to-end time.  These entry points take the WHOLE file as one buffer and build the packed, device-resident
 * inputs of gab_bsw_run_device / gab_bpm_run_device / gab_wfa_run_device on the GPU: a newline index (count, scan,
 * fill), then per pair the lengths, offsets, h0 and the code bytes.
 * They accept exactly the well-formed files the reference accepts without hitting one of its buffer limits
 * (every line ends with '\n'; bsw: h0 line of at most 8 characters, reference line < 2047, query line < 255
 * characters, both non-empty) and return GAB_EINVAL otherwise -- a caller then falls back to its line-by-line parser.
 * All outputs live in memory owned by the handle and stay valid until the next call on it. */
typedef struct gab_parser gab_parser;
typedef struct {
    int64_t n;                                   /* pairs = newline count / 3 (main_banded.cpp:237-253) */
    const uint8_t *d_ref; const int64_t *d_ref_off;      /* codes (character - '0'); byte offset of each pair's sequence, 4-byte aligned */
    const uint8_t *d_qry; const int64_t *d_qry_off;
    const int32_t *d_len1, *d_len2, *d_h0;
    int64_t ref_bytes, qry_bytes;
} gab_bsw_packed;
typedef struct {
    int64_t n;                                   /* pairs = newline count / 2 */
    const char *d_text;                          /* the file itself; sequences are used in place */
    int64_t text_bytes;                          /* readable bytes of d_text (>= nbytes; the staged copy has 64 bytes of slack) */
    const int64_t *d_pat_off, *d_txt_off;        /* first base of each sequence (the '>' / '<' prefix is skipped) */
    const int32_t *d_pat_len, *d_txt_len;
    const int64_t *d_cap_off;                    /* n + 1 offsets of per-pair output slots of pattern_length + text_length bytes */
    int64_t cap_bytes;                           /* (rounded up to 4): the ops_off / buffer size gab_wfa_run_device needs */
} gab_pairs_packed;
int gab_parser_create(int device, gab_parser **out);
void gab_parser_destroy(gab_parser *p);
/* text: host pointer (copied to the device) or, for the _device variants, device pointer */
int gab_bsw_parse_pairs(gab_parser *p, const char *text, int64_t nbytes, gab_bsw_packed *out, void *stream);
int gab_bsw_parse_pairs_device(gab_parser *p, const char *d_text, int64_t nbytes, gab_bsw_packed *out, void *stream);
/* swap_longer_first != 0: the longer line becomes the pattern (bpm, align_benchmark.c:177-181); 0: '>' is the pattern (wfa) */
int gab_pairs_parse(gab_parser *p, const char *text, int64_t nbytes, int swap_longer_first, gab_pairs_packed *out, void *stream);
int gab_pairs_parse_device(gab_parser *p, const char *d_text, int64_t nbytes, int swap_longer_first, gab_pairs_packed *out,
                           void *stream);
/* chain / fast-chain (chain/src/host_data_io.cpp:13-51): files in the one-record-per-line layout -- a header line
 * "n avg_qspan max_dist_x max_dist_y bw n_segs", n lines "x y", one line "EOR" -- become the inputs of
 * gab_chain_run_device: anchors on the device, the call table (offsets and headers) on the host, as that entry point
 * takes it.  The header lines are converted on the host with the reference's own fscanf format. */
typedef struct {
    int64_t ncalls, total;                       /* calls, anchors of all calls */
    const uint64_t *d_x, *d_y;                   /* device: anchors of all calls back to back */
    const int64_t *call_off;                     /* host: ncalls + 1 offsets into d_x / d_y */
    const gab_chain_hdr *hdr;                    /* host: ncalls headers */
} gab_chain_packed;
int gab_chain_parse(gab_parser *p, const char *text, int64_t nbytes, gab_chain_packed *out, void *stream);
int gab_chain_parse_device(gab_parser *p, const char *d_text, int64_t nbytes, gab_chain_packed *out, void *stream);
/* last parse: kernel milliseconds (HIP events on the launch stream, excluding the host-to-device copy of the text) */
int gab_parser_last_stats(gab_parser *p, float *kernel_ms);

#ifdef __cplusplus
}
#endif
#endif /* GAB_H */
