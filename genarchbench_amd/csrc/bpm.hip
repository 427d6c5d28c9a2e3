// bpm -- bit-parallel Myers global edit distance (+ backtrace-derived score) on gfx950.
//
// Semantics: benchmark_edit_bpm, /root/reference/benchmarks/bpm/benchmark/benchmark_edit.c:31-56
//   = edit_bpm_pattern_compile (bpm/edit/edit_bpm.c:70-136) + edit_bpm_compute_matrix (:190-275,
//   block step BPM_ADVANCE_BLOCK :47-66) + edit_bpm_backtrace_matrix (:276-316) and
//   score = -edit_cigar_score_edit (bpm/edit/edit_cigar.c:103-116), called with
//   max_distance = pattern_length and text_length <= pattern_length (the driver's swap,
//   bpm/tools/align_benchmark.c:177-181), so no block is ever cut off.
//
// The reference stores Pv/Mv of EVERY column (7.3 KB per 151-bp pair) only to backtrace, and the
// printed score is the number of non-match backtrace operations.  When both sequences consist of
// upper-case A/C/G/T only, the match predicate of the bit-vectors equals raw byte equality and
// every backtrace step follows an optimal DP move, so the count is exactly the DP distance
// (bpm) -- no matrix needed.  Otherwise two reference quirks can make the count differ
// (code 4 aliases onto the next block's 'A' mask; diagonal steps are classified by raw bytes), and
// the pair takes the full path.  Hence two kernels, one pair per lane in both:
//
//   bpm_score<W>   all pairs, bucketed by W = ceil(plen/64) in 1..4: Pv/Mv in registers, the
//                  4W+1 match masks of each lane in LDS as [mask][lane] (conflict-free b64 reads),
//                  built with ds_or; writes -distance for clean pairs, queues the rest.
//   bpm_full<W>    queued pairs (and every pair with W > 4, W = 0 instantiation): same recurrence,
//                  but each column's Pv/Mv is stored to a per-pair region of a global scratch,
//                  followed by the reference's backtrace on that region.
//
// Roofline: ~36 VALU per (text char x 64-row block); 306 B of input per 151-bp pair.  The score
// path is integer-VALU bound at about 16 k VALU per pair; its HBM traffic is the algorithmic
// plen + tlen + 4 B per pair.  The full path adds 16 B x W x (tlen+1) of scratch per queued pair.
#include "gab_internal.h"
#include "gab_bitvec.h"
#include <algorithm>
#include <new>
#include <vector>
#include <string.h>

namespace {

constexpr int kMaxRegW = 4;                 // W classes kept in registers
constexpr int kClasses = kMaxRegW + 1;      // class c = W for W <= 4, class 0 = W > 4
constexpr int kBlock = 256;
constexpr int kSlices = 16;                 // a class is scored in slices; the band kernel of slice k overlaps the score kernel of slice k + 1 (8: -1.4 %, 32: -30 %)
constexpr int kBandRows = 8;                // rows kept per column by the LDS band kernel ({Pv, Mv} bits: one u16 per column)

struct BpmIO {
    const char *pat; const int64_t *pat_off; const int32_t *pat_len;
    const char *txt; const int64_t *txt_off; const int32_t *txt_len;
    int64_t pat_bytes, txt_bytes, n;
};

struct BpmCounters {        // device-side, zeroed per run
    uint32_t cls_count[8];  // pairs per W class (index = class)
    uint32_t cls_cursor[8];
    uint32_t wl_count[8];   // queued (unclean) pairs per class (host-side sum of the slices, for the stats)
    uint32_t wl_slice[8][kSlices];   // ... per class and slice: written by bpm_score, read by bpm_band on the device
    uint32_t wl2_count[8];  // pairs whose backtrace left the 64-row window (re-run with the full history)
    uint32_t wl1_count[8];  // pairs whose backtrace left the 8-row LDS band (re-run with the 64-row window)
    int32_t max_tlen[8];    // longest text per class (sizes the LDS band)
    int32_t max_plen[8];    // longest pattern per class (selects the number of 32-bit words of the score kernel)
    int32_t bad, first_bad;
    unsigned long long steps;   // block steps executed by bpm_score (m * W summed)
    unsigned long long full_steps;
};
GAB_STATIC_ATOMIC64(BpmCounters, steps); GAB_STATIC_ATOMIC64(BpmCounters, full_steps);

__device__ __forceinline__ int bpm_class(int n) {
    const int W = (n + 63) >> 6;
    return W <= kMaxRegW ? W : 0;
}

// code 0..3 for ACGT/acgt, 4 otherwise (bpm/utils/dna_text.c:47-51); *clean &= upper-case ACGT
__device__ __forceinline__ int bpm_code(uint32_t ch, bool &clean) {
    const uint32_t idx = ch & 0x1f;
    const bool letter = (ch & 0xc0u) == 0x40u;
    const bool acgt = letter && ((0x00100088u | 2u) >> idx & 1u);    // bits 1 (A), 3 (C), 7 (G), 20 (T)
    const uint32_t x = (ch >> 1) & 3u;                               // A0 C1 G3 T2
    clean = clean && acgt && !(ch & 0x20u);
    return acgt ? (int)(x ^ (x >> 1)) : 4;
}

// bpm_code for the four bytes of a dword at once: the 2-bit codes as bytes (A0 C1 G2 T3), and whether all four are upper-case
// A / C / G / T -- the codes select the expected letter from "ACGT" with one v_perm_b32 and one compare checks all four.
// (A byte that is none of them yields some code 0 .. 3 and ok = false: the pair is re-done by the band kernel.)
__device__ __forceinline__ uint32_t bpm_codes4(uint32_t w, bool &clean) {
    const uint32_t x = (w >> 1) & 0x03030303u;                               // A0 C1 G3 T2
    const uint32_t c4 = x ^ ((x >> 1) & 0x01010101u);                        // A0 C1 G2 T3
    clean = clean && __builtin_amdgcn_perm(0u, 0x54474341u, c4) == w;       // selector byte k picks 'A' 'C' 'G' 'T'
    return c4;
}

__device__ __forceinline__ uint32_t ld_u32(const char *p) { uint32_t w; __builtin_memcpy(&w, p, 4); return w; }
// 16 bytes per lane and load: every lane streams its own sequence, and with 4-byte loads each 64-byte line was
// re-fetched from HBM up to 16 times (profiles/r01_hbm_traffic.md: 8.7x the algorithmic bytes, HBM-bound)
__device__ __forceinline__ uint4 ld_u128(const char *p) { uint4 w; __builtin_memcpy(&w, p, 16); return w; }

// Wave-aggregated "counters[cls] += 1" returning each lane's slot: one atomic per (wave, class present)
// instead of one per lane (all 10 M pairs of the 151-bp workload share one class, i.e. one address).
__device__ __forceinline__ uint32_t wave_class_add(uint32_t *counters, int cls, bool active) {
    uint32_t slot = 0;
    unsigned long long todo = __ballot(active);
    const unsigned long long lt = (1ull << (threadIdx.x & 63)) - 1;
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int c = __shfl(cls, leader);
        const unsigned long long same = __ballot(active && cls == c) & todo;
        uint32_t base = 0;
        if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&counters[c], (uint32_t)__popcll(same));
        base = __shfl(base, leader);
        if (active && cls == c) slot = base + (uint32_t)__popcll(same & lt);
        todo &= ~same;
    }
    return slot;
}

// ---- pass 1: validate + count per class ---------------------------------------------------
// counts are accumulated per lane over the grid-stride loop and reduced once per wave: a single atomic per class and
// wave (with 10 M pairs in one class, an atomic per wave-iteration still serialised on one address for ~1.7 ms)
__global__ __launch_bounds__(256) void bpm_count(BpmIO io, BpmCounters *ct) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t mine[kClasses] = {0, 0, 0, 0, 0};
    int mt[kClasses] = {0, 0, 0, 0, 0}, mp[kClasses] = {0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < io.n; i += stride) {
        const int n = io.pat_len[i], m = io.txt_len[i];
        const int64_t po = io.pat_off[i], to = io.txt_off[i];
        const bool ok = n >= 1 && n <= GAB_BPM_MAX_PLEN && m >= 0 && m <= n && po >= 0 && to >= 0 &&
                        po + n + 3 <= io.pat_bytes && to + m + 3 <= io.txt_bytes;          // (the kernels read dwords from the sequence's own start)
        if (!ok) {
            atomicAdd(&ct->bad, 1);
            atomicMin((unsigned int *)&ct->first_bad, (unsigned int)(i + 1 > 0x7fffffff ? 0x7fffffff : i + 1));
            continue;
        }
        const int cls = bpm_class(n);
#pragma unroll
        for (int k = 0; k < kClasses; k++) { mine[k] += cls == k; mt[k] = (cls == k && m > mt[k]) ? m : mt[k]; mp[k] = (cls == k && n > mp[k]) ? n : mp[k]; }
    }
    // wave reductions, then one set of atomics per workgroup (they serialise in L2 on a handful of addresses)
    __shared__ int s_mt[4][kClasses], s_mp[4][kClasses];
    __shared__ uint32_t s_cnt[4][kClasses];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kClasses; k++) {
        int v = mt[k], vp = mp[k];
        for (int o = 32; o > 0; o >>= 1) { const int u = __shfl_xor(v, o); v = u > v ? u : v; const int up = __shfl_xor(vp, o); vp = up > vp ? up : vp; }
        uint32_t c = mine[k];
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if ((threadIdx.x & 63) == 0) { s_mt[wv][k] = v; s_mp[wv][k] = vp; s_cnt[wv][k] = c; }
    }
    __syncthreads();
    if (threadIdx.x < kClasses) {
        const int k = threadIdx.x;
        int v = 0, vp = 0; uint32_t c = 0;
        for (int w2 = 0; w2 < 4; w2++) { v = s_mt[w2][k] > v ? s_mt[w2][k] : v; vp = s_mp[w2][k] > vp ? s_mp[w2][k] : vp; c += s_cnt[w2][k]; }
        if (v > 0) atomicMax(&ct->max_tlen[k], v);
        if (vp > 0) atomicMax(&ct->max_plen[k], vp);
        if (c) atomicAdd(&ct->cls_count[k], c);
    }
}

// ---- pass 2: scatter ids by class (only when more than one class is populated) -----------------------------
// every workgroup owns one contiguous chunk: it counts its classes in LDS, reserves its ranges with one global
// atomic per class, then hands out slots with LDS atomics
__global__ __launch_bounds__(256) void bpm_scatter(BpmIO io, BpmCounters *ct, uint32_t *perm) {
    __shared__ uint32_t cnt[kClasses], base[kClasses];
    const int64_t per = (io.n + gridDim.x - 1) / gridDim.x;
    const int64_t b = (int64_t)blockIdx.x * per, e = b + per < io.n ? b + per : io.n;
    if (threadIdx.x < kClasses) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) atomicAdd(&cnt[bpm_class(io.pat_len[i])], 1u);
    __syncthreads();
    if (threadIdx.x < kClasses) { base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&ct->cls_cursor[threadIdx.x], cnt[threadIdx.x]) : 0; cnt[threadIdx.x] = 0; }
    __syncthreads();
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        const int cls = bpm_class(io.pat_len[i]);
        perm[base[cls] + atomicAdd(&cnt[cls], 1u)] = (uint32_t)i;
    }
}

// one 64-row block step (BPM_ADVANCE_BLOCK, edit_bpm.c:47-66)
__device__ __forceinline__ void bpm_step(uint64_t Eq, uint64_t mask, uint64_t &P, uint64_t &M, uint32_t &PH,
                                         uint32_t &MH) {
    const uint64_t Xv = Eq | M;
    const uint64_t Eq2 = Eq | (uint64_t)MH;
    const uint64_t Xh = (((Eq2 & P) + P) ^ P) | Eq2;
    uint64_t Ph = M | ~(Xh | P);
    uint64_t Mh = P & Xh;
    const uint32_t PHo = (Ph & mask) != 0, MHo = (Mh & mask) != 0;
    Ph = (Ph << 1) | (uint64_t)PH;
    Mh = (Mh << 1) | (uint64_t)MH;
    P = Mh | ~(Xv | Ph);
    M = Ph & Xv;
    PH = PHo; MH = MHo;
}

// Builds the 4W+1 match masks of this lane in LDS (peq[mask * kBlock + tid]); returns cleanliness.
template <int W, int kBlock = 256>
__device__ __forceinline__ bool bpm_build_peq(uint64_t *peq, const char *p, int n) {
    for (int k = 0; k < 4 * W + 1; k++) peq[k * kBlock] = 0;
    bool clean = true;
    int i0 = 0;
    for (; i0 + 16 <= n; i0 += 16) {
        const uint4 q = ld_u128(p + i0);
        const uint32_t ws[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int i = i0 + k;
            const int c = bpm_code((ws[k >> 2] >> ((k & 3) * 8)) & 0xffu, clean);
            atomicOr((unsigned long long *)&peq[((i >> 6) * 4 + c) * kBlock], 1ull << (i & 63));
        }
    }
    for (; i0 < n; i0 += 4) {
        uint32_t w = ld_u32(p + i0);
        for (int k = 0; k < 4 && i0 + k < n; k++, w >>= 8) {
            const int i = i0 + k;
            const int c = bpm_code(w & 0xffu, clean);
            atomicOr((unsigned long long *)&peq[((i >> 6) * 4 + c) * kBlock], 1ull << (i & 63));
        }
    }
    // padding rows n .. 64W-1 match every code 0..3 (edit_bpm.c:106-113)
    if (n & 63) {
        const uint64_t pad = ~0ull << (n & 63);
        for (int c = 0; c < 4; c++) peq[((W - 1) * 4 + c) * kBlock] |= pad;
    }
    return clean;
}

// ---- score path: W in 1..4 -------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(kBlock) void bpm_score(BpmIO io, const uint32_t *__restrict__ perm, uint32_t kbeg,
                                                    uint32_t kend, int32_t *__restrict__ score_out,
                                                    uint32_t *__restrict__ worklist, uint32_t *wl_counter, BpmCounters *ct) {
    __shared__ uint64_t peq_s[(4 * W + 1) * kBlock];
    const uint32_t k = kbeg + blockIdx.x * kBlock + threadIdx.x;
    unsigned long long steps = 0;
    int64_t queue_id = -1;
    if (k < kend) {
        const uint32_t id = perm ? perm[k] : k;        // perm == nullptr: all pairs share this class, identity order
        const int n = io.pat_len[id], m = io.txt_len[id];
        const char *p = io.pat + io.pat_off[id], *t = io.txt + io.txt_off[id];
        uint64_t *peq = peq_s + threadIdx.x;
        bool clean = bpm_build_peq<W>(peq, p, n);
        uint64_t P[W], M[W];
#pragma unroll
        for (int b = 0; b < W; b++) { P[b] = ~0ull; M[b] = 0; }
        const uint64_t top_mask = (n & 63) ? 1ull << ((n & 63) - 1) : 1ull << 63;
        int score = n;
        int h0 = 0;
        for (; h0 + 16 <= m; h0 += 16) {
            const uint4 q = ld_u128(t + h0);
            // (codes of four text bases at a time: bpm_code per base was 8 of the 54 instructions a base costs)
            const uint32_t cs[4] = {bpm_codes4(q.x, clean), bpm_codes4(q.y, clean), bpm_codes4(q.z, clean), bpm_codes4(q.w, clean)};
#pragma unroll
            for (int kk = 0; kk < 16; kk++) {
                const int c = (int)((cs[kk >> 2] >> ((kk & 3) * 8)) & 3u);
                uint32_t PH = 1, MH = 0;
#pragma unroll
                for (int b = 0; b < W; b++)
                    bpm_step(peq[(b * 4 + c) * kBlock], b == W - 1 ? top_mask : 1ull << 63, P[b], M[b], PH, MH);
                score += (int)PH - (int)MH;
            }
        }
        for (; h0 < m; h0 += 4) {
            uint32_t w = ld_u32(t + h0);
            for (int kk = 0; kk < 4 && h0 + kk < m; kk++, w >>= 8) {
                const int c = bpm_code(w & 0xffu, clean);
                uint32_t PH = 1, MH = 0;
#pragma unroll
                for (int b = 0; b < W; b++)
                    bpm_step(peq[(b * 4 + (c & 3)) * kBlock], b == W - 1 ? top_mask : 1ull << 63, P[b], M[b], PH, MH);
                score += (int)PH - (int)MH;
            }
        }
        steps = (unsigned long long)m * W;
        if (clean) score_out[id] = -score;
        else queue_id = (int64_t)id;
    }
    {
        // append the unclean pairs of this wave with one atomic
        const unsigned long long q = __ballot(queue_id >= 0);
        if (q) {
            const int leader = __builtin_ctzll(q);
            uint32_t base = 0;
            if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(wl_counter, (uint32_t)__popcll(q));
            base = __shfl(base, leader);
            if (queue_id >= 0) worklist[base + (uint32_t)__popcll(q & ((1ull << (threadIdx.x & 63)) - 1))] = (uint32_t)queue_id;
        }
    }
    for (int o = 32; o > 0; o >>= 1) steps += __shfl_xor(steps, o);
    // one atomic per workgroup, not per wave: 160 000 waves in 5 ms on one address keep an L2 atomic unit a quarter busy
    __shared__ unsigned long long s_steps[kBlock / 64];
    if ((threadIdx.x & 63) == 0) s_steps[threadIdx.x >> 6] = steps;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long all = 0;
        for (int k = 0; k < kBlock / 64; k++) all += s_steps[k];
        if (all) atomicAdd(&ct->steps, all);
    }
}

// ---- score path in 32-bit words: D = ceil(plen / 32) in 1..8, the whole column as ONE D-word integer ----------------
// BPM_ADVANCE_BLOCK chains 64-row blocks through its horizontal carries (edit_bpm.c:47-66); the vertical deltas it leaves are
// those of the DP matrix, so a column advanced as one multi-word integer (Myers' recurrence with the carry of the addition
// and the bits of the two shifts running through the words) holds the same Pv / Mv and the same distance.  Per 32-bit word:
// or, and, add-with-carry, three v_bitop3_b32 (gfx950's three-input boolean: (sum ^ P) | Eq, M | ~(Xh | P), Mh | ~(Xv | Ph)),
// two v_alignbit_b32, two and = 10 instructions, against ~16 per word of the block form as the compiler writes it (64-bit
// shifts, compares and selects for the block carries), and a 151-base pattern is five words, not three blocks = six.
// The distance is read off the last column: m + popcount(P & rows) - popcount(M & rows) (gab_bitvec.h); nothing per column.
// One column of the D-word form (gab_bitvec.h): the Eq words of the text base's code from the lane's 64-row masks (`e`: block 0
// of the code; blocks are 4 masks apart) -> new P / M.
template <int D, int STRIDE>
__device__ __forceinline__ void bpm_step32(const uint64_t *e, uint32_t (&P)[D], uint32_t (&M)[D]) {
    constexpr int W = (D + 1) / 2;
    uint32_t Eq[D];
#pragma unroll
    for (int b = 0; b < W; b++) {
        const uint64_t q = e[(size_t)b * 4 * STRIDE];
        Eq[2 * b] = (uint32_t)q;
        if (2 * b + 1 < D) Eq[2 * b + 1] = (uint32_t)(q >> 32);
    }
    gab_myers_step32<D>(Eq, P, M);
}

template <int D>
__device__ __forceinline__ void bpm_score32_body(uint64_t *peq_s, BpmIO io, const uint32_t *__restrict__ perm, uint32_t kbeg,
                                                 uint32_t kend, int32_t *__restrict__ score_out,
                                                 uint32_t *__restrict__ worklist, uint32_t *wl_counter, BpmCounters *ct) {
    constexpr int W = (D + 1) / 2;
    const uint32_t k = kbeg + blockIdx.x * kBlock + threadIdx.x;
    unsigned long long steps = 0;
    int64_t queue_id = -1;
    if (k < kend) {
        const uint32_t id = perm ? perm[k] : k;
        const int n = io.pat_len[id], m = io.txt_len[id];
#ifdef GAB_KO_BPM_LOCAL_TEXT      // measurement only (wrong results; bench inputs only, whose offsets ascend): every lane of a wave reads
                                  // the strings of the wave's FIRST pair -- what a perfectly coalesced text layout could gain at most
        const uint32_t id0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)id);
        const char *p = io.pat + io.pat_off[id0], *t = io.txt + io.txt_off[id0];
#else
        const char *p = io.pat + io.pat_off[id], *t = io.txt + io.txt_off[id];
#endif
        uint64_t *peq = peq_s + threadIdx.x;
        bool clean = bpm_build_peq<W>(peq, p, n);
        uint32_t P[D], M[D];
#pragma unroll
        for (int d = 0; d < D; d++) { P[d] = ~0u; M[d] = 0; }
        auto step = [&](int c) { bpm_step32<D, kBlock>(peq + (size_t)c * kBlock, P, M); };
        int h0 = 0;
#ifdef GAB_BPM_TRIP64             // experiment (profiles/r04_kernel_bounds.md): sixty-four bases per trip, four 16-byte loads at once
        for (; h0 + 64 <= m; h0 += 64) {
            const uint4 q0 = ld_u128(t + h0), q1 = ld_u128(t + h0 + 16), q2 = ld_u128(t + h0 + 32), q3 = ld_u128(t + h0 + 48);
            const uint32_t cs[16] = {bpm_codes4(q0.x, clean), bpm_codes4(q0.y, clean), bpm_codes4(q0.z, clean), bpm_codes4(q0.w, clean),
                                     bpm_codes4(q1.x, clean), bpm_codes4(q1.y, clean), bpm_codes4(q1.z, clean), bpm_codes4(q1.w, clean),
                                     bpm_codes4(q2.x, clean), bpm_codes4(q2.y, clean), bpm_codes4(q2.z, clean), bpm_codes4(q2.w, clean),
                                     bpm_codes4(q3.x, clean), bpm_codes4(q3.y, clean), bpm_codes4(q3.z, clean), bpm_codes4(q3.w, clean)};
#pragma unroll
            for (int kk = 0; kk < 64; kk++) step((int)((cs[kk >> 2] >> ((kk & 3) * 8)) & 3u));
        }
#endif
        // thirty-two bases per trip, both 16-byte loads at once: a 64-byte line is visited twice instead of four times
        for (; h0 + 32 <= m; h0 += 32) {
            const uint4 q = ld_u128(t + h0), r = ld_u128(t + h0 + 16);
            const uint32_t cs[8] = {bpm_codes4(q.x, clean), bpm_codes4(q.y, clean), bpm_codes4(q.z, clean), bpm_codes4(q.w, clean),
                                    bpm_codes4(r.x, clean), bpm_codes4(r.y, clean), bpm_codes4(r.z, clean), bpm_codes4(r.w, clean)};
#pragma unroll
            for (int kk = 0; kk < 32; kk++) step((int)((cs[kk >> 2] >> ((kk & 3) * 8)) & 3u));
        }
        for (; h0 + 16 <= m; h0 += 16) {
            const uint4 q = ld_u128(t + h0);
            const uint32_t cs[4] = {bpm_codes4(q.x, clean), bpm_codes4(q.y, clean), bpm_codes4(q.z, clean), bpm_codes4(q.w, clean)};
#pragma unroll
            for (int kk = 0; kk < 16; kk++) step((int)((cs[kk >> 2] >> ((kk & 3) * 8)) & 3u));
        }
        for (; h0 < m; h0 += 4) {
            uint32_t w = ld_u32(t + h0);
            for (int kk = 0; kk < 4 && h0 + kk < m; kk++, w >>= 8) step(bpm_code(w & 0xffu, clean) & 3);
        }
        steps = (unsigned long long)m * W;
        // the distance from the vertical deltas of the last column: nothing is tracked per column
        if (clean) score_out[id] = -gab_myers_distance32<D>(P, M, n, m);
        else queue_id = (int64_t)id;
    }
    {
        // append the unclean pairs of this wave with one atomic
        const unsigned long long q = __ballot(queue_id >= 0);
        if (q) {
            const int leader = __builtin_ctzll(q);
            uint32_t base = 0;
            if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(wl_counter, (uint32_t)__popcll(q));
            base = __shfl(base, leader);
            if (queue_id >= 0) worklist[base + (uint32_t)__popcll(q & ((1ull << (threadIdx.x & 63)) - 1))] = (uint32_t)queue_id;
        }
    }
    for (int o = 32; o > 0; o >>= 1) steps += __shfl_xor(steps, o);
    __shared__ unsigned long long s_steps[kBlock / 64];
    if ((threadIdx.x & 63) == 0) s_steps[threadIdx.x >> 6] = steps;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long all = 0;
        for (int k2 = 0; k2 < kBlock / 64; k2++) all += s_steps[k2];
        if (all) atomicAdd(&ct->steps, all);
    }
}

// Up to six words the kernel needs 54 VGPRs when the compiler is told to fit six waves per SIMD (left alone it takes 114 for
// the sixteen unrolled columns, four waves): 24 waves per CU, what the 26.6 KB of masks per workgroup allow (bpm-large +4 %).
// Seven and eight words do not fit that budget and keep the uncapped form.
template <int D>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(6, 6)))
void bpm_score32(BpmIO io, const uint32_t *__restrict__ perm, uint32_t kbeg, uint32_t kend, int32_t *__restrict__ score_out,
                 uint32_t *__restrict__ worklist, uint32_t *wl_counter, BpmCounters *ct) {
    __shared__ uint64_t peq_s[(4 * ((D + 1) / 2) + 1) * kBlock];
    bpm_score32_body<D>(peq_s, io, perm, kbeg, kend, score_out, worklist, wl_counter, ct);
}
template <int D>
__global__ __launch_bounds__(kBlock)
void bpm_score32_wide(BpmIO io, const uint32_t *__restrict__ perm, uint32_t kbeg, uint32_t kend, int32_t *__restrict__ score_out,
                      uint32_t *__restrict__ worklist, uint32_t *wl_counter, BpmCounters *ct) {
    __shared__ uint64_t peq_s[(4 * ((D + 1) / 2) + 1) * kBlock];
    bpm_score32_body<D>(peq_s, io, perm, kbeg, kend, score_out, worklist, wl_counter, ct);
}

// ---- LDS band path: 8 rows around the diagonal per column, history never leaves the CU ------------------------
// First stop of a queued pair.  One pair per lane, one wave per workgroup; the column history is one dword per
// column ({Pv, Mv} bits of the 8 rows around the diagonal: one u16; 16 rows halved the occupancy and were 16 % slower
// end to end) in LDS as [column][lane], so the ~300 dependent reads of
// the backtrace cost LDS latency instead of HBM latency (the global-history kernels below spend ~20 ms of pure
// latency on 1.4 M pairs).  A backtrace that drifts more than 3 rows off the diagonal is queued for bpm_win.
// dynamic LDS: [ (4W+1) x 64 masks (u64) ][ (cols) x 64 u16 ]
template <int D>
// The grid covers the whole slice (launched right behind the slice's score kernel, without a host round trip); the number
// of queued pairs is read on the device and the workgroups behind it leave at once (a capped grid striding over the slots
// was measured 3 % slower).  The columns advance in the D-word form of bpm_score32 (D = 2W - 1 or 2W words for the class
// W = ceil(plen / 64); the masks keep the 64-row layout, code-4 aliasing of edit_bpm.c included).
__global__ __launch_bounds__(64) void bpm_band(BpmIO io, const uint32_t *__restrict__ list, const uint32_t *__restrict__ nslots_ptr, int cols,
                                               int32_t *__restrict__ score_out, uint32_t *__restrict__ miss_list,
                                               BpmCounters *ct) {
    constexpr int W = (D + 1) / 2;
    extern __shared__ uint64_t band_smem[];
    const int lane = threadIdx.x;
    const uint32_t nslots = *nslots_ptr;
    if (blockIdx.x * 64u >= nslots) return;
    const uint32_t s = blockIdx.x * 64 + lane;
    int64_t miss_id = -1;
    unsigned long long steps = 0;
    if (s < nslots) {
        const uint32_t id = list[s];
        const int n = io.pat_len[id], m = io.txt_len[id];
        const char *p = io.pat + io.pat_off[id], *t = io.txt + io.txt_off[id];
        uint64_t *peq = band_smem + lane;
        uint16_t *B = reinterpret_cast<uint16_t *>(band_smem + (4 * W + 1) * 64) + lane;      // column c at B[c * 64]
        bpm_build_peq<W, 64>(peq, p, n);
        const int cshift = (n - m) / 2;
        auto start = [&](int col) { int r = col + cshift - kBandRows / 2; r = r < 0 ? 0 : r; return r > 64 * W - kBandRows ? 64 * W - kBandRows : r; };
        uint32_t P[D], M[D];
#pragma unroll
        for (int d = 0; d < D; d++) { P[d] = ~0u; M[d] = 0; }
        B[0] = (uint16_t)((1u << kBandRows) - 1u);                // column 0: Pv = 1..1, Mv = 0
        bool dummy = true;
        auto column = [&](int h, uint32_t byte) {
            const int c = bpm_code(byte, dummy);
            bpm_step32<D, 64>(peq + (size_t)c * 64, P, M);
            // the 8 rows of the band from row r0 on: rows of the pattern end below 32 D (r0 <= plen - 4), a word past the
            // last one reads as 0 (rows the walk never stands on)
            const int r0 = start(h + 1);
            const int b0 = r0 >> 5;
            const uint32_t sh = (uint32_t)(r0 & 31);
            uint32_t plo = P[0], phi = D > 1 ? P[D > 1 ? 1 : 0] : 0, mlo = M[0], mhi = D > 1 ? M[D > 1 ? 1 : 0] : 0;
            // which word the band starts in is the same for the whole wave except in the few columns where the lanes
            // cross a word boundary (their diagonals differ by a row or two): then the words are named by a scalar switch
            // instead of being selected per lane
            const int b0u = __builtin_amdgcn_readfirstlane(b0);
            if (D > 1 && __ballot(b0 != b0u) == 0) {
#pragma unroll
                for (int b = 1; b < D; b++)
                    if (b0u == b) { plo = P[b]; mlo = M[b]; phi = b + 1 < D ? P[b + 1 < D ? b + 1 : b] : 0; mhi = b + 1 < D ? M[b + 1 < D ? b + 1 : b] : 0; }
            } else {
#pragma unroll
                for (int b = 1; b < D; b++)
                    if (b0 == b) { plo = P[b]; mlo = M[b]; phi = b + 1 < D ? P[b + 1 < D ? b + 1 : b] : 0; mhi = b + 1 < D ? M[b + 1 < D ? b + 1 : b] : 0; }
            }
            const uint32_t bm = (1u << kBandRows) - 1u;
            const uint32_t pw = __builtin_amdgcn_alignbit(phi, plo, sh) & bm;     // ({hi, lo} >> sh): sh = 0 gives lo
            const uint32_t mw = __builtin_amdgcn_alignbit(mhi, mlo, sh) & bm;
            B[(h + 1) * 64] = (uint16_t)(pw | mw << kBandRows);
        };
        int h0 = 0;
        for (; h0 + 16 <= m; h0 += 16) {                     // sixteen columns per load, as in the score kernel
            const uint4 q = ld_u128(t + h0);
            const uint32_t ws[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int kk = 0; kk < 16; kk++) column(h0 + kk, (ws[kk >> 2] >> ((kk & 3) * 8)) & 0xffu);
        }
        for (; h0 < m; h0 += 4) {
            uint32_t w4 = ld_u32(t + h0);
            for (int kk = 0; kk < 4 && h0 + kk < m; kk++, w4 >>= 8) column(h0 + kk, w4 & 0xffu);
        }
        steps = (unsigned long long)m * W;
        (void)cols;
        int ops = 0, v = n - 1, h = m - 1;
        bool miss = false;
        // The walk itself only needs the band bits (LDS); the two bases of a diagonal step only feed the count, but as byte
        // loads at the step they put a global-memory round trip (~0.7 us) into every step.  Both strings are read backwards,
        // at most one byte per step, so each is held as the 8-byte chunk under the cursor plus the chunk below it, which is
        // requested when the cursor enters a chunk: a load has eight steps to arrive.
        const int n4 = (n + 3) & ~3, m4 = (m + 3) & ~3;                    // readable bytes (the slabs are padded to dwords)
        auto chunk = [](const char *q, int k, int len4, uint32_t &lo, uint32_t &hi) {
            lo = hi = 0;
            if (k >= 0) { lo = ld_u32(q + 8 * k); if (8 * k + 4 < len4) hi = ld_u32(q + 8 * k + 4); }
        };
        int tk = h >> 3, pk = v >> 3;
        uint32_t tlo, thi, tnlo, tnhi, plo2, phi2, pnlo, pnhi;
        chunk(t, tk, m4, tlo, thi); chunk(t, tk - 1, m4, tnlo, tnhi);
        chunk(p, pk, n4, plo2, phi2); chunk(p, pk - 1, n4, pnlo, pnhi);
        while (v >= 0 && h >= 0) {
            const int r1 = start(h + 1), rh = start(h);
            if (v < r1 || v >= r1 + kBandRows || v < rh || v >= rh + kBandRows) { miss = true; break; }
            if ((h >> 3) != tk) { tk--; tlo = tnlo; thi = tnhi; chunk(t, tk - 1, m4, tnlo, tnhi); }
            if ((v >> 3) != pk) { pk--; plo2 = pnlo; phi2 = pnhi; chunk(p, pk - 1, n4, pnlo, pnhi); }
            const uint32_t bc = B[(h + 1) * 64], bp = B[h * 64];
            if ((bc >> (v - r1)) & 1) { ops++; v--; }
            else if ((bp >> (kBandRows + v - rh)) & 1) { ops++; h--; }
            else {
                const uint32_t tc = (((h & 4) ? thi : tlo) >> ((h & 3) * 8)) & 0xffu, pc = (((v & 4) ? phi2 : plo2) >> ((v & 3) * 8)) & 0xffu;
                ops += tc != pc; h--; v--;
            }
        }
        if (miss) miss_id = (int64_t)id;
        else score_out[id] = -(ops + (h + 1) + (v + 1));
    }
    {
        const unsigned long long q = __ballot(miss_id >= 0);
        if (q) {
            const int leader = __builtin_ctzll(q);
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&ct->wl1_count[W], (uint32_t)__popcll(q));
            base = __shfl(base, leader);
            if (miss_id >= 0) miss_list[base + (uint32_t)__popcll(q & ((1ull << lane) - 1))] = (uint32_t)miss_id;
        }
    }
    for (int o = 32; o > 0; o >>= 1) steps += __shfl_xor(steps, o);
    if (lane == 0 && steps) atomicAdd(&ct->full_steps, steps);
}

// ---- windowed path: history of a 64-row diagonal window + backtrace -----------------------------------------
// The backtrace only ever looks at the Pv / Mv bit of the cell it stands on, and for similar sequences it stays
// near the main diagonal.  So instead of the whole column (W words each of Pv and Mv) only the 64 rows around
// the diagonal are kept: 16 bytes per column, four columns per 64-byte line (3x less scratch written, ~5x fewer
// lines read back by the backtrace than with full columns).  If the walk ever steps outside the window the pair is
// queued once more and handled by bpm_full with complete columns, so the result is exact in every case.
template <int W>
__device__ __forceinline__ int bpm_win_start(int col, int cshift) {
    int r = col + cshift - 32;
    r = r < 0 ? 0 : r;
    return r > 64 * W - 64 ? 64 * W - 64 : r;
}
template <int W>
__device__ __forceinline__ uint64_t bpm_win_take(const uint64_t (&X)[W], int r0) {
    const int b0 = r0 >> 6, s = r0 & 63;
    uint64_t lo = X[0], hi = W > 1 ? X[W > 1 ? 1 : 0] : 0;
#pragma unroll
    for (int b = 1; b < W; b++) if (b0 == b) { lo = X[b]; hi = b + 1 < W ? X[b + 1 < W ? b + 1 : b] : 0; }
    return s ? (lo >> s) | (hi << (64 - s)) : lo;
}
template <int W>
__global__ __launch_bounds__(kBlock) void bpm_win(BpmIO io, const uint32_t *__restrict__ list, const uint32_t *__restrict__ nslots_ptr,
                                                  ulonglong2 *__restrict__ hist, int per_slot,
                                                  int32_t *__restrict__ score_out, uint32_t *__restrict__ miss_list,
                                                  BpmCounters *ct) {
    __shared__ uint64_t peq_s[(4 * W + 1) * kBlock];
    // The number of listed pairs is read HERE (the list was filled by the band kernels in front of this launch: no host round
    // trip to size the grid, r04); the grid is sized for the history room, every thread owns one history slot and takes
    // list entries slot, slot + threads, ...
    const uint32_t nslots = *nslots_ptr;
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    unsigned long long steps = 0;
  for (uint32_t sbase = blockIdx.x * kBlock; sbase < nslots; sbase += gridDim.x * kBlock) {
    const uint32_t s = sbase + threadIdx.x;
    int64_t miss_id = -1;
    if (s < nslots) {
        const uint32_t id = list[s];
        const int n = io.pat_len[id], m = io.txt_len[id];
        const char *p = io.pat + io.pat_off[id], *t = io.txt + io.txt_off[id];
        uint64_t *peq = peq_s + threadIdx.x;
        bpm_build_peq<W>(peq, p, n);
        ulonglong2 *H = hist + (int64_t)slot * per_slot;        // record of column c: {Pv window, Mv window}
        const int cshift = (n - m) / 2;
        const uint64_t top_mask = (n & 63) ? 1ull << ((n & 63) - 1) : 1ull << 63;
        uint64_t P[W], M[W];
#pragma unroll
        for (int b = 0; b < W; b++) { P[b] = ~0ull; M[b] = 0; }
        H[0] = make_ulonglong2(~0ull, 0ull);
        bool dummy = true;
        // r04: the handful of pairs that come here are the LAST thing a call waits for (0.3 ms behind the last band kernel of
        // bpm-large, a quarter of the step of a 1.25 M-pair share), and what they waited for was memory latency: a byte load per
        // column in front of every step, and two to four dependent loads per backtrace step.  The text now arrives sixteen bases
        // per load, and the backtrace -- which visits the columns m, m - 1, ... in this order whatever path it takes, and the
        // rows n - 1, n - 2, ... likewise -- keeps the next eight column records and the next dword of each string in
        // registers, requested before they are needed.  bpm_win on bpm-large (~100 pairs): 0.306 -> 0.223 ms.
        auto column = [&](int h, int c) {
            uint32_t PH = 1, MH = 0;
#pragma unroll
            for (int b = 0; b < W; b++)
                bpm_step(peq[(b * 4 + c) * kBlock], b == W - 1 ? top_mask : 1ull << 63, P[b], M[b], PH, MH);
            const int r0 = bpm_win_start<W>(h + 1, cshift);
            H[h + 1] = make_ulonglong2(bpm_win_take<W>(P, r0), bpm_win_take<W>(M, r0));
        };
        int hh = 0;
        for (; hh + 16 <= m; hh += 16) {
            const uint4 q = ld_u128(t + hh);
            const uint32_t ws[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 16; k++) column(hh + k, bpm_code((ws[k >> 2] >> ((k & 3) * 8)) & 0xffu, dummy));
        }
        for (; hh < m; hh += 4) {
            uint32_t w4 = ld_u32(t + hh);
            for (int k = 0; k < 4 && hh + k < m; k++, w4 >>= 8) column(hh + k, bpm_code(w4 & 0xffu, dummy));
        }
        steps += (unsigned long long)m * W;
        // backtrace (edit_bpm.c:289-313) on the windows
        int ops = 0, v = n - 1, h = m - 1;
        bool miss = false;
        constexpr int kAhead = 8;                     // R[k] = record of column h + 1 - k
        ulonglong2 R[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; k++) R[k] = H[h + 1 - k > 0 ? h + 1 - k : 0];
        // (the strings' dwords: bytes up to three behind a sequence are readable, and a dword in front of its first byte is
        // never asked for: the index is clamped at 0)
        uint32_t tw = h >= 0 ? ld_u32(t + (h & ~3)) : 0, tw_n = h >= 4 ? ld_u32(t + (h & ~3) - 4) : 0;
        uint32_t pw = v >= 0 ? ld_u32(p + (v & ~3)) : 0, pw_n = v >= 4 ? ld_u32(p + (v & ~3) - 4) : 0;
        auto step_h = [&]() {                         // h has just been decremented
#pragma unroll
            for (int k = 0; k + 1 < kAhead; k++) R[k] = R[k + 1];
            const int col = h + 1 - (kAhead - 1);
            R[kAhead - 1] = H[col > 0 ? col : 0];
            if ((h & 3) == 3) { tw = tw_n; tw_n = h >= 4 ? ld_u32(t + (h & ~3) - 4) : 0; }
        };
        auto step_v = [&]() {                         // v has just been decremented
            if ((v & 3) == 3) { pw = pw_n; pw_n = v >= 4 ? ld_u32(p + (v & ~3) - 4) : 0; }
        };
        while (v >= 0 && h >= 0) {
            const int r1 = bpm_win_start<W>(h + 1, cshift), rh = bpm_win_start<W>(h, cshift);
            if (v < r1 || v >= r1 + 64 || v < rh || v >= rh + 64) { miss = true; break; }
            if ((R[0].x >> (v - r1)) & 1) { ops++; v--; step_v(); }
            else if ((R[1].y >> (v - rh)) & 1) { ops++; h--; step_h(); }
            else { ops += ((tw >> ((h & 3) * 8)) & 0xffu) != ((pw >> ((v & 3) * 8)) & 0xffu); h--; v--; step_h(); step_v(); }
        }
        if (miss) miss_id = (int64_t)id;
        else score_out[id] = -(ops + (h + 1) + (v + 1));
    }
    {
        const unsigned long long q = __ballot(miss_id >= 0);
        if (q) {
            const int leader = __builtin_ctzll(q);
            uint32_t base = 0;
            if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&ct->wl2_count[W], (uint32_t)__popcll(q));
            base = __shfl(base, leader);
            if (miss_id >= 0) miss_list[base + (uint32_t)__popcll(q & ((1ull << (threadIdx.x & 63)) - 1))] = (uint32_t)miss_id;
        }
    }
  }
    for (int o = 32; o > 0; o >>= 1) steps += __shfl_xor(steps, o);
    if ((threadIdx.x & 63) == 0 && steps) atomicAdd(&ct->full_steps, steps);
}

// ---- full path: history + backtrace ---------------------------------------------------------------
// slot s of this launch works on pair list[s]; element (col, b) of its history lives at
//   hist[base + ((col * Wd + b) * 2 + {0: P, 1: M}) * estride]       (u64 units)
// REGW > 0: W == REGW for every slot, each slot owns a contiguous (64W+1) x W x 2 region (its 16-byte
//           column records merge into full lines in the write-back L2, and the backtrace of a lane
//           stays inside its own few KB instead of striding across the whole scratch); masks in LDS.
// REGW == 0: any W (<= 255), base = slot_base[s]; the 4W+1 masks sit in front of the history.
template <int REGW>
__global__ __launch_bounds__(kBlock) void bpm_full(BpmIO io, const uint32_t *__restrict__ list, uint32_t nslots_arg, const uint32_t *__restrict__ nslots_ptr,
                                                   uint64_t *__restrict__ hist, const int64_t *__restrict__ slot_base,
                                                   int32_t *__restrict__ score_out, BpmCounters *ct) {
    __shared__ uint64_t peq_s[(REGW ? 4 * REGW + 1 : 1) * kBlock];
    // nslots_ptr (REGW > 0): the list's length is read here, the grid is sized for the history room and a thread takes the
    // entries slot, slot + threads, ... with ITS history slot (see bpm_win)
    const uint32_t nslots = nslots_ptr ? *nslots_ptr : nslots_arg;
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
  for (uint32_t s = slot; s < nslots; s += gridDim.x * kBlock) {
    const uint32_t id = list[s];
    const int n = io.pat_len[id], m = io.txt_len[id];
    const char *p = io.pat + io.pat_off[id], *t = io.txt + io.txt_off[id];
    const int Wd = REGW ? REGW : (n + 63) >> 6;
    const int64_t estride = 1;
    uint64_t *H;                 // history origin of this slot
    uint64_t *peq;               // mask k at peq[k * pstride]
    int64_t pstride;
    bool dummy = true;
    if (REGW) {
        H = hist + (int64_t)slot * ((64 * REGW + 1) * REGW * 2);
        peq = peq_s + threadIdx.x; pstride = kBlock;
        bpm_build_peq<(REGW ? REGW : 1)>(peq, p, n);
    } else {
        peq = hist + slot_base[s]; pstride = 1;
        H = peq + 4 * Wd + 1;
        for (int k = 0; k < 4 * Wd + 1; k++) peq[k] = 0;
        for (int i = 0; i < n; i++) peq[(i >> 6) * 4 + bpm_code((uint8_t)p[i], dummy)] |= 1ull << (i & 63);
        if (n & 63) for (int c = 0; c < 4; c++) peq[(Wd - 1) * 4 + c] |= ~0ull << (n & 63);
    }
    const uint64_t top_mask = (n & 63) ? 1ull << ((n & 63) - 1) : 1ull << 63;
#define HP(col, b) H[(((int64_t)(col) * Wd + (b)) * 2) * estride]
#define HM(col, b) H[(((int64_t)(col) * Wd + (b)) * 2 + 1) * estride]
    uint64_t P[REGW ? REGW : 1], M[REGW ? REGW : 1];
    if (REGW) {
#pragma unroll
        for (int b = 0; b < REGW; b++) { P[b] = ~0ull; M[b] = 0; HP(0, b) = ~0ull; HM(0, b) = 0; }
    } else {
        for (int b = 0; b < Wd; b++) { HP(0, b) = ~0ull; HM(0, b) = 0; }
    }
    for (int h = 0; h < m; h++) {
        const int c = bpm_code((uint8_t)t[h], dummy);
        uint32_t PH = 1, MH = 0;
        if (REGW) {
#pragma unroll
            for (int b = 0; b < REGW; b++) {
                bpm_step(peq[(b * 4 + c) * pstride], b == REGW - 1 ? top_mask : 1ull << 63, P[b], M[b], PH, MH);
                HP(h + 1, b) = P[b]; HM(h + 1, b) = M[b];
            }
        } else {
            for (int b = 0; b < Wd; b++) {
                uint64_t Pb = HP(h, b), Mb = HM(h, b);
                bpm_step(peq[b * 4 + c], b == Wd - 1 ? top_mask : 1ull << 63, Pb, Mb, PH, MH);
                HP(h + 1, b) = Pb; HM(h + 1, b) = Mb;
            }
        }
    }
    // backtrace (edit_bpm.c:289-313), counting non-match operations
    int ops = 0, v = n - 1, h = m - 1;
    while (v >= 0 && h >= 0) {
        const int b = v >> 6;
        const uint64_t bit = 1ull << (v & 63);
        if (HP(h + 1, b) & bit) { ops++; v--; }
        else if (HM(h, b) & bit) { ops++; h--; }
        else { ops += t[h] != p[v]; h--; v--; }
    }
    ops += (h + 1) + (v + 1);
    score_out[id] = -ops;
    {
        // one atomic per wave for the step counter (exec mask = lanes with a slot)
        unsigned long long st = (unsigned long long)m * Wd;
        const unsigned long long act = __ballot(true);
        const int leader = __builtin_ctzll(act);
        unsigned long long sum = 0;
        for (unsigned long long r = act; r; r &= r - 1) sum += __shfl(st, __builtin_ctzll(r));
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&ct->full_steps, sum);
    }
#undef HP
#undef HM
  }
}

}  // namespace

// lengths of the listed pairs, back to back (the host sizes the per-pair history slots of the generic path from them)
__global__ __launch_bounds__(256) void bpm_gather_lens(const uint32_t *__restrict__ ids, uint32_t n, const int32_t *__restrict__ pat_len,
                                                       const int32_t *__restrict__ txt_len, int32_t *__restrict__ pl, int32_t *__restrict__ tl) {
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n; k += gridDim.x * 256u) { pl[k] = pat_len[ids[k]]; tl[k] = txt_len[ids[k]]; }
}

// =============================================================================== host side
struct gab_bpm {
    gab_tuning tun = gab_tuning_loaded();      // experiment knobs, read when the handle is made
    gab_host_stream hs;     // private stream of the host-pointer entry point(s)
    int device = 0;
    gab_devbuf ws;          // counters | perm | worklists
    gab_devbuf scratch;     // history of the full path
    gab_devbuf lens;        // pattern / text lengths of the pairs on the generic path
    gab_devbuf io;          // staging for the host-pointer entry point
    size_t scratch_budget = (size_t)8 << 30;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t aux = nullptr;     // the band kernels run here, each behind its slice's score kernel
    hipEvent_t fork = nullptr, join = nullptr, scored[kSlices] = {};
    BpmCounters *h_ct = nullptr;   // pinned
    bool have_stats = false;
    int64_t last_full = 0;
};

extern "C" int gab_bpm_create(int device, gab_bpm **out) {
    if (!out) { gab_set_error("gab_bpm_create: NULL argument"); return GAB_EINVAL; }
    *out = nullptr;
    int rc = gab_check_device(device);
    if (rc) return rc;
    gab_device_guard g(device);
    gab_bpm *h = new (std::nothrow) gab_bpm();
    if (!h) { gab_set_error("out of host memory"); return GAB_ENOMEM; }
    h->device = device;
    for (int k = 0; k < 4; k++)
        if (hipEventCreate(&h->ev[k]) != hipSuccess) { gab_set_error("hipEventCreate failed"); delete h; return GAB_EDEVICE; }
    // (a high-priority stream for the band kernels was measured 5 % slower: it starves the score kernel)
    if (hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->join, hipEventDisableTiming) != hipSuccess) {
        gab_set_error("gab_bpm_create: stream / event creation failed"); delete h; return GAB_EDEVICE;
    }
    for (int k = 0; k < kSlices; k++)
        if (hipEventCreateWithFlags(&h->scored[k], hipEventDisableTiming) != hipSuccess) {
            gab_set_error("gab_bpm_create: event creation failed"); delete h; return GAB_EDEVICE;
        }
    bool attr_ok = true;
    for (const void *f : {(const void *)bpm_band<1>, (const void *)bpm_band<2>, (const void *)bpm_band<3>, (const void *)bpm_band<4>,
                          (const void *)bpm_band<5>, (const void *)bpm_band<6>, (const void *)bpm_band<7>, (const void *)bpm_band<8>})
        attr_ok = attr_ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    if (!attr_ok) {
        gab_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); delete h; return GAB_EDEVICE;
    }
    if (hipHostMalloc((void **)&h->h_ct, sizeof(BpmCounters)) != hipSuccess) {
        gab_set_error("hipHostMalloc failed"); delete h; return GAB_ENOMEM;
    }
    *out = h;
    return GAB_OK;
}

extern "C" void gab_bpm_destroy(gab_bpm *h) {
    if (!h) return;
    gab_device_guard g(h->device);
    h->ws.release(); h->scratch.release(); h->lens.release(); h->io.release(); h->hs.release();
    for (int k = 0; k < 4; k++) if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
    if (h->fork) (void)hipEventDestroy(h->fork);
    if (h->join) (void)hipEventDestroy(h->join);
    for (int k = 0; k < kSlices; k++) if (h->scored[k]) (void)hipEventDestroy(h->scored[k]);
    if (h->aux) (void)hipStreamDestroy(h->aux);
    if (h->h_ct) (void)hipHostFree(h->h_ct);
    delete h;
}

template <int W>
static void launch_score(hipStream_t s, const BpmIO &io, const uint32_t *perm, uint32_t kb, uint32_t ke, int32_t *score,
                         uint32_t *wl, uint32_t *wl_counter, BpmCounters *ct, int max_plen, bool blocks64) {
    if (ke <= kb) return;      // (blocks64 = GAB_BPM_SCORE64: the 64-row block form, kept under test)
    const dim3 grid((ke - kb + kBlock - 1) / kBlock);
    if (blocks64)
        hipLaunchKernelGGL(bpm_score<W>, grid, dim3(kBlock), 0, s, io, perm, kb, ke, score, wl, wl_counter, ct);
    else if constexpr (W <= 3) {
        if (max_plen <= 32 * (2 * W - 1))                                     // the class's longest pattern fits 2W - 1 words
            hipLaunchKernelGGL(bpm_score32<2 * W - 1>, grid, dim3(kBlock), 0, s, io, perm, kb, ke, score, wl, wl_counter, ct);
        else
            hipLaunchKernelGGL(bpm_score32<2 * W>, grid, dim3(kBlock), 0, s, io, perm, kb, ke, score, wl, wl_counter, ct);
    } else {
        if (max_plen <= 32 * (2 * W - 1))
            hipLaunchKernelGGL(bpm_score32_wide<2 * W - 1>, grid, dim3(kBlock), 0, s, io, perm, kb, ke, score, wl, wl_counter, ct);
        else
            hipLaunchKernelGGL(bpm_score32_wide<2 * W>, grid, dim3(kBlock), 0, s, io, perm, kb, ke, score, wl, wl_counter, ct);
    }
}
// the score kernel of one slice on `s`, its band kernel on `sb` behind the event
template <int W>
static int launch_slice(hipStream_t s, hipStream_t sb, hipEvent_t scored, const BpmIO &io, const uint32_t *perm, uint32_t kb,
                        uint32_t ke, int32_t *score, uint32_t *wl, uint32_t *wl_counter, int cols, uint32_t *wl1, BpmCounters *ct, int max_plen, bool blocks64) {
    if (ke <= kb) return GAB_OK;
    launch_score<W>(s, io, perm, kb, ke, score, wl, wl_counter, ct, max_plen, blocks64);
    GAB_HIP(hipEventRecord(scored, s));
    GAB_HIP(hipStreamWaitEvent(sb, scored, 0));
    const size_t lds = sizeof(uint64_t) * 64 * (4 * W + 1) + sizeof(uint16_t) * 64 * (size_t)cols;
    if (max_plen <= 32 * (2 * W - 1))
        hipLaunchKernelGGL(bpm_band<2 * W - 1>, dim3((ke - kb + 63) / 64), dim3(64), lds, sb, io, wl, wl_counter, cols, score, wl1, ct);
    else
        hipLaunchKernelGGL(bpm_band<2 * W>, dim3((ke - kb + 63) / 64), dim3(64), lds, sb, io, wl, wl_counter, cols, score, wl1, ct);
    return GAB_OK;
}
template <int W>
static void launch_full(hipStream_t s, const BpmIO &io, const uint32_t *list, uint32_t nslots, uint64_t *hist,
                        const int64_t *slot_base, int32_t *score, BpmCounters *ct) {
    hipLaunchKernelGGL(bpm_full<W>, dim3((nslots + kBlock - 1) / kBlock), dim3(kBlock), 0, s, io, list, nslots, (const uint32_t *)nullptr, hist,
                       slot_base, score, ct);
}

extern "C" int gab_bpm_run_device(gab_bpm *h, const char *pat, int64_t pat_bytes, const int64_t *pat_off,
                                  const int32_t *pat_len, const char *txt, int64_t txt_bytes, const int64_t *txt_off,
                                  const int32_t *txt_len, int64_t n, int32_t *score_out, void *stream_) {
    GAB_CHECK(h, "gab_bpm_run_device: NULL handle");
    GAB_CHECK(n >= 0 && n < (1ll << 31), "gab_bpm_run_device: n=%lld out of range", (long long)n);
    h->have_stats = false;
    if (n == 0) return GAB_OK;
    GAB_CHECK(pat && pat_off && pat_len && txt && txt_off && txt_len && score_out, "gab_bpm_run_device: NULL buffer");
    gab_device_guard g(h->device);
    gab_tuning_refresh(&h->tun);
    hipStream_t s = (hipStream_t)stream_;

    const size_t o_perm = (sizeof(BpmCounters) + 255) & ~(size_t)255;
    const size_t o_wl = o_perm + sizeof(uint32_t) * (size_t)n;
    const size_t o_wl2 = o_wl + sizeof(uint32_t) * (size_t)n;
    const size_t o_wl1 = o_wl2 + sizeof(uint32_t) * (size_t)n;
    int rc = h->ws.reserve(o_wl1 + sizeof(uint32_t) * (size_t)n);
    if (rc) return rc;
    char *base = h->ws.as<char>();
    BpmCounters *d_ct = (BpmCounters *)base;
    uint32_t *d_perm = (uint32_t *)(base + o_perm), *d_wl = (uint32_t *)(base + o_wl), *d_wl2 = (uint32_t *)(base + o_wl2), *d_wl1 = (uint32_t *)(base + o_wl1);
    BpmIO io{pat, pat_off, pat_len, txt, txt_off, txt_len, pat_bytes, txt_bytes, n};

    GAB_HIP(hipEventRecord(h->ev[0], s));
    memset(h->h_ct, 0, sizeof(BpmCounters));
    h->h_ct->first_bad = 0x7fffffff;
    GAB_HIP(hipMemcpyAsync(d_ct, h->h_ct, sizeof(BpmCounters), hipMemcpyHostToDevice, s));
    const int grid = (int)std::min<int64_t>(gab_ceil_div(n, 256), 1024);
    hipLaunchKernelGGL(bpm_count, dim3(grid), dim3(256), 0, s, io, d_ct);
    GAB_HIP(hipMemcpyAsync(h->h_ct, d_ct, sizeof(BpmCounters), hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    if (h->h_ct->bad) {
        gab_set_error("gab_bpm_run_device: %d pair(s) violate the limits (first: pair %d): need 1 <= pattern_length <= %d, "
                      "0 <= text_length <= pattern_length (apply the driver's longer-is-pattern swap), offsets inside "
                      "the slabs with 3 more readable bytes behind every sequence", h->h_ct->bad, h->h_ct->first_bad - 1,
                      GAB_BPM_MAX_PLEN);
        return GAB_EINVAL;
    }
    // class starts: perm = [class 1 | class 2 | class 3 | class 4 | class 0 (W > 4)]
    uint32_t cstart[kClasses + 1], ccount[kClasses];
    const int order[kClasses] = {1, 2, 3, 4, 0};
    uint32_t run = 0;
    for (int k = 0; k < kClasses; k++) { ccount[order[k]] = h->h_ct->cls_count[order[k]]; cstart[order[k]] = run; run += ccount[order[k]]; }
    for (int c = 0; c < kClasses; c++) h->h_ct->cls_cursor[c] = cstart[c];
    GAB_HIP(hipMemcpyAsync(d_ct, h->h_ct, sizeof(BpmCounters), hipMemcpyHostToDevice, s));
    int populated = 0;
    for (int c = 0; c < kClasses; c++) populated += ccount[c] != 0;
    const bool identity = populated == 1 && ccount[0] == 0;      // one register class holds every pair: no scatter pass
    if (!identity) hipLaunchKernelGGL(bpm_scatter, dim3(grid), dim3(256), 0, s, io, d_ct, d_perm);
    const uint32_t *perm_arg = identity ? nullptr : d_perm;

    // score path + stage 0 of the queued (unclean) pairs (8-row band in LDS).  The worklist of class W occupies
    // d_wl[cstart[W] ..); a class is cut into kSlices slices whose score kernels run back to back on the caller's stream
    // while each slice's band kernel runs on h->aux behind an event: the latency-bound band kernel (2.7 of 7.9 ms, 23 %
    // VALU busy on its own) then runs underneath the VALU-bound score kernel of the next slice and only the last
    // slice's band is exposed.  Pairs whose backtrace leaves the band are queued once more (64-row window), and from
    // there to complete columns.
    GAB_HIP(hipEventRecord(h->ev[1], s));
    GAB_HIP(hipEventRecord(h->fork, s));
    GAB_HIP(hipStreamWaitEvent(h->aux, h->fork, 0));
    for (int W = 1; W <= kMaxRegW; W++) {
        if (!ccount[W]) continue;
        // slices of ~600 k pairs (10 M pairs: sixteen): a launch of fewer pairs no longer fills the chip -- an eighth of bpm-large
        // (one rank's share on 8 GPUs) cut into sixteen slices of 78 k pairs took 2.66 ms where 10 M take 5.4 (r03)
        // (r04: below ~2.4 M pairs ONE slice -- every band launch has a ~0.12 ms tail of its own that the next slice's score kernel
        // does not hide at that size: 1.25 M pairs 1.12 -> 1.04 ms, 100 k pairs 0.39 ms with two slices against 0.27 with one)
        int nsl = ccount[W] < 2400000 ? 1 : (int)std::min<uint32_t>(kSlices, ccount[W] / 600000);
        if (h->tun.bpm_slices > 0) nsl = std::max(1, std::min(kSlices, h->tun.bpm_slices));        // GAB_BPM_SLICES: tuning runs
        const uint32_t per = ((ccount[W] + nsl - 1) / nsl + kBlock - 1) / kBlock * kBlock;
        const int cols = h->h_ct->max_tlen[W] + 1;
        for (int k = 0; k < nsl; k++) {
            const uint32_t kb = cstart[W] + std::min<uint32_t>(ccount[W], (uint32_t)k * per);
            const uint32_t ke = cstart[W] + std::min<uint32_t>(ccount[W], (uint32_t)(k + 1) * per);
            uint32_t *wl = d_wl + kb, *cnt = &d_ct->wl_slice[W][k], *wl1 = d_wl1 + cstart[W];
            switch (W) {
                case 1: rc = launch_slice<1>(s, h->aux, h->scored[k], io, perm_arg, kb, ke, score_out, wl, cnt, cols, wl1, d_ct, h->h_ct->max_plen[1], h->tun.bpm_score64); break;
                case 2: rc = launch_slice<2>(s, h->aux, h->scored[k], io, perm_arg, kb, ke, score_out, wl, cnt, cols, wl1, d_ct, h->h_ct->max_plen[2], h->tun.bpm_score64); break;
                case 3: rc = launch_slice<3>(s, h->aux, h->scored[k], io, perm_arg, kb, ke, score_out, wl, cnt, cols, wl1, d_ct, h->h_ct->max_plen[3], h->tun.bpm_score64); break;
                default: rc = launch_slice<4>(s, h->aux, h->scored[k], io, perm_arg, kb, ke, score_out, wl, cnt, cols, wl1, d_ct, h->h_ct->max_plen[4], h->tun.bpm_score64); break;
            }
            if (rc) return rc;
        }
    }
    GAB_HIP(hipGetLastError());
    GAB_HIP(hipEventRecord(h->join, h->aux));
    GAB_HIP(hipStreamWaitEvent(s, h->join, 0));
    GAB_HIP(hipEventRecord(h->ev[2], s));

    {
        // stage 1: 64-row window in global memory for the pairs that left the band; stage 2: complete columns for the pairs that
        // left the window.  r04: both are launched without asking the device how many pairs they got (that was two host round
        // trips per call -- 0.1 of the 0.29 ms of a 100 000-pair batch): a launch is sized for the history room the handle has and
        // reads the length of its list itself; its threads take list entries in strides (bpm_win / bpm_full).
        for (int W = 1; W <= kMaxRegW; W++) {
            if (!ccount[W]) continue;
            const int per_slot = 64 * W + 1;                       // columns 0 .. tlen <= 64 W, 16 bytes each
            const size_t bytes_slot = (size_t)per_slot * 16;
            uint32_t slots = (uint32_t)std::max<size_t>(kBlock, std::min<size_t>({(size_t)ccount[W], h->scratch_budget / bytes_slot, (size_t)1 << 16}));
            slots = (slots + kBlock - 1) / kBlock * kBlock;
            rc = h->scratch.reserve(bytes_slot * slots);
            if (rc) return rc;
            const uint32_t *list = d_wl1 + cstart[W];
            ulonglong2 *hist = h->scratch.as<ulonglong2>();
            const dim3 g(slots / kBlock), blk(kBlock);
            switch (W) {
                case 1: hipLaunchKernelGGL(bpm_win<1>, g, blk, 0, s, io, list, (const uint32_t *)&d_ct->wl1_count[1], hist, per_slot, score_out, d_wl2 + cstart[W], d_ct); break;
                case 2: hipLaunchKernelGGL(bpm_win<2>, g, blk, 0, s, io, list, (const uint32_t *)&d_ct->wl1_count[2], hist, per_slot, score_out, d_wl2 + cstart[W], d_ct); break;
                case 3: hipLaunchKernelGGL(bpm_win<3>, g, blk, 0, s, io, list, (const uint32_t *)&d_ct->wl1_count[3], hist, per_slot, score_out, d_wl2 + cstart[W], d_ct); break;
                default: hipLaunchKernelGGL(bpm_win<4>, g, blk, 0, s, io, list, (const uint32_t *)&d_ct->wl1_count[4], hist, per_slot, score_out, d_wl2 + cstart[W], d_ct); break;
            }
        }
        for (int W = 1; W <= kMaxRegW; W++) {
            if (!ccount[W]) continue;
            const size_t per_slot = (size_t)(64 * W + 1) * W * 16;
            uint32_t slots = (uint32_t)std::max<size_t>(kBlock, std::min<size_t>({(size_t)ccount[W], h->scratch_budget / per_slot, (size_t)1 << 14}));
            slots = (slots + kBlock - 1) / kBlock * kBlock;
            rc = h->scratch.reserve(per_slot * slots);
            if (rc) return rc;
            const uint32_t *list = d_wl2 + cstart[W];
            uint64_t *hist = h->scratch.as<uint64_t>();
            const dim3 g(slots / kBlock), blk(kBlock);
            switch (W) {
                case 1: hipLaunchKernelGGL(bpm_full<1>, g, blk, 0, s, io, list, 0u, (const uint32_t *)&d_ct->wl2_count[1], hist, (const int64_t *)nullptr, score_out, d_ct); break;
                case 2: hipLaunchKernelGGL(bpm_full<2>, g, blk, 0, s, io, list, 0u, (const uint32_t *)&d_ct->wl2_count[2], hist, (const int64_t *)nullptr, score_out, d_ct); break;
                case 3: hipLaunchKernelGGL(bpm_full<3>, g, blk, 0, s, io, list, 0u, (const uint32_t *)&d_ct->wl2_count[3], hist, (const int64_t *)nullptr, score_out, d_ct); break;
                default: hipLaunchKernelGGL(bpm_full<4>, g, blk, 0, s, io, list, 0u, (const uint32_t *)&d_ct->wl2_count[4], hist, (const int64_t *)nullptr, score_out, d_ct); break;
            }
        }
        GAB_HIP(hipGetLastError());
        // W > 4: every pair takes the generic full path with per-slot extents
        const uint32_t cnt0 = ccount[0];
        if (cnt0) {
            std::vector<int32_t> pl(cnt0), tl(cnt0);
            // lengths of those pairs: gathered on the device into one buffer, one copy back (a long-read input puts EVERY
            // pair here; two 4-byte copies per pair were minutes of host latency at 10 M pairs)
            rc = h->lens.reserve(8 * (size_t)cnt0);
            if (rc) return rc;
            int32_t *d_pl = h->lens.as<int32_t>(), *d_tl = d_pl + cnt0;
            hipLaunchKernelGGL(bpm_gather_lens, dim3((unsigned)std::min<int64_t>(gab_ceil_div((int64_t)cnt0, 256), 4096)), dim3(256), 0, s,
                               d_perm + cstart[0], cnt0, pat_len, txt_len, d_pl, d_tl);
            GAB_HIP(hipMemcpyAsync(pl.data(), d_pl, 4 * (size_t)cnt0, hipMemcpyDeviceToHost, s));
            GAB_HIP(hipMemcpyAsync(tl.data(), d_tl, 4 * (size_t)cnt0, hipMemcpyDeviceToHost, s));
            GAB_HIP(hipStreamSynchronize(s));
            uint32_t k0 = 0;
            while (k0 < cnt0) {
                std::vector<int64_t> sb;
                size_t used = 0; uint32_t k1 = k0;
                while (k1 < cnt0) {
                    const size_t Wd = ((size_t)pl[k1] + 63) / 64;
                    const size_t need = 4 * Wd + 1 + ((size_t)tl[k1] + 1) * Wd * 2;
                    if (k1 > k0 && (used + need) * 8 > h->scratch_budget) break;
                    sb.push_back((int64_t)used); used += need; k1++;
                }
                const size_t nb = sb.size();
                rc = h->scratch.reserve(used * 8 + nb * 8 + 64);
                if (rc) return rc;
                int64_t *d_sb = (int64_t *)(h->scratch.as<char>() + ((used * 8 + 63) & ~(size_t)63));
                rc = h->scratch.reserve(((used * 8 + 63) & ~(size_t)63) + nb * 8);
                if (rc) return rc;
                d_sb = (int64_t *)(h->scratch.as<char>() + ((used * 8 + 63) & ~(size_t)63));
                GAB_HIP(hipMemcpyAsync(d_sb, sb.data(), nb * 8, hipMemcpyHostToDevice, s));
                launch_full<0>(s, io, d_perm + cstart[0] + k0, (uint32_t)nb, h->scratch.as<uint64_t>(), d_sb, score_out, d_ct);
                GAB_HIP(hipStreamSynchronize(s));      // sb (host) and the scratch are reused by the next batch
                k0 = k1;
            }
        }
    }
    GAB_HIP(hipGetLastError());
    GAB_HIP(hipMemcpyAsync(h->h_ct, d_ct, sizeof(BpmCounters), hipMemcpyDeviceToHost, s));
    GAB_HIP(hipEventRecord(h->ev[3], s));
    h->last_full = (int64_t)ccount[0];      // (+ the pairs the band kernels queued: added from the counters in gab_bpm_last_stats)
    h->have_stats = true;
    return GAB_OK;
}

extern "C" int gab_bpm_run(gab_bpm *h, const char *pat, const int64_t *pat_off, const int32_t *pat_len,
                           const char *txt, const int64_t *txt_off, const int32_t *txt_len, int64_t n,
                           int32_t *score_out) {
    GAB_CHECK(h, "gab_bpm_run: NULL handle");
    GAB_CHECK(n >= 0 && n < (1ll << 31), "gab_bpm_run: n=%lld out of range", (long long)n);
    if (n == 0) return GAB_OK;
    GAB_CHECK(pat && pat_off && pat_len && txt && txt_off && txt_len && score_out, "gab_bpm_run: NULL buffer");
    gab_device_guard g(h->device);
    int64_t pb = 0, tb = 0, pa = INT64_MAX, ta = INT64_MAX;
    for (int64_t i = 0; i < n; i++) {
        GAB_CHECK(pat_off[i] >= 0 && txt_off[i] >= 0 && pat_len[i] >= 0 && txt_len[i] >= 0,
                  "gab_bpm_run: negative offset/length at pair %lld", (long long)i);
        pb = std::max(pb, pat_off[i] + pat_len[i]); tb = std::max(tb, txt_off[i] + txt_len[i]);
        pa = std::min(pa, pat_off[i]); ta = std::min(ta, txt_off[i]);
    }
    pa &= ~(int64_t)255; ta &= ~(int64_t)255;      // stage only the referenced window [min, max) of each slab
    // one slab for both with overlapping windows (the drivers' pair files: '>' and '<' lines interleaved): staged once, not twice
    const bool shared = pat == txt && std::max(pb, tb) - std::min(pa, ta) <= (pb - pa) + (tb - ta);
    if (shared) { pa = ta = std::min(pa, ta); pb = tb = std::max(pb, tb); }
    const size_t ppad = ((size_t)(pb - pa) + 3 + 255) & ~(size_t)255, tpad = shared ? 0 : ((size_t)(tb - ta) + 3 + 255) & ~(size_t)255;
    const size_t nn = (size_t)n;
    size_t o = 0;
    const size_t o_p = o; o += ppad;
    const size_t o_t = o; o += tpad;
    const size_t o_po = o; o += 8 * nn;
    const size_t o_to = o; o += 8 * nn;
    const size_t o_pl = o; o += 4 * nn;
    const size_t o_tl = o; o += 4 * nn;
    const size_t o_sc = o; o += 4 * nn;
    int rc = h->io.reserve(o);
    if (rc) return rc;
    char *b = h->io.as<char>();
    hipStream_t s = nullptr;
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    {   // the copies of one chunk at a time per GPU (gab_core.hip: the workers of a GPU must not copy in lockstep)
        std::lock_guard<std::mutex> gate(gab_h2d_mutex(h->device));
        GAB_HIP(hipMemcpyAsync(b + o_p, pat + pa, (size_t)(pb - pa), hipMemcpyHostToDevice, s));
        if (!shared) GAB_HIP(hipMemcpyAsync(b + o_t, txt + ta, (size_t)(tb - ta), hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_po, pat_off, 8 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_to, txt_off, 8 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_pl, pat_len, 4 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_tl, txt_len, 4 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipStreamSynchronize(s));
    }
    rc = gab_bpm_run_device(h, b + o_p - pa, pa + (int64_t)ppad, (const int64_t *)(b + o_po), (const int32_t *)(b + o_pl),
                            (shared ? b + o_p : b + o_t) - ta, ta + (int64_t)(shared ? ppad : tpad), (const int64_t *)(b + o_to), (const int32_t *)(b + o_tl), n,
                            (int32_t *)(b + o_sc), s);
    if (rc) return rc;
    GAB_HIP(hipMemcpyAsync(score_out, b + o_sc, 4 * nn, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    return GAB_OK;
}

// Pre-size the handle's device buffers for calls of up to max_pairs pairs whose sequences span up to max_seq_bytes of the
// slab(s), and warm the copy queues of its stream, so that the first gab_bpm_run inside a timed region pays for neither
// (see gab_bsw_reserve).
extern "C" int gab_bpm_reserve(gab_bpm *h, int64_t max_pairs, int64_t max_seq_bytes) {
    GAB_CHECK(h, "gab_bpm_reserve: NULL handle");
    GAB_CHECK(max_pairs >= 0 && max_pairs < (1ll << 31) && max_seq_bytes >= 0, "gab_bpm_reserve: size out of range");
    gab_device_guard g(h->device);
    const size_t nn = (size_t)max_pairs;
    int rc = h->io.reserve(std::max<size_t>(2 * (((size_t)max_seq_bytes + 3 + 511) & ~(size_t)255) + 28 * nn + 1024, (size_t)4 << 20));
    if (rc) return rc;
    if ((rc = h->ws.reserve(sizeof(BpmCounters) + 512 + 4 * sizeof(uint32_t) * nn)) != GAB_OK) return rc;
    hipStream_t s = nullptr;
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    GAB_HIP(hipMemsetAsync(h->io.p, 0, h->io.cap, s));
    GAB_HIP(hipMemsetAsync(h->ws.p, 0, h->ws.cap, s));
    GAB_HIP(hipStreamSynchronize(s));
    if ((rc = gab_warm_copy_engines(s, h->io.p, h->io.cap)) != GAB_OK) return rc;
    // ... and one tiny batch through the whole path (an unclean pair among them, so that the history kernels are launched too):
    // first launches cost milliseconds once per process and handle -- not inside the caller's ROI
    static const char seq[] = "ACGTTGCAACGTACGTTGCATGCAACGTACGT" "ACGTTGCANCCTACGTTGCATGAACGTACGTA" "    ";
    const int64_t po[3] = {0, 32, 0}, to[3] = {32, 0, 2};
    const int32_t pl[3] = {32, 32, 30}, tl[3] = {30, 32, 28};
    int32_t sc[3];
    const bool had = h->have_stats;
    rc = gab_bpm_run(h, seq, po, pl, seq, to, tl, 3, sc);
    h->have_stats = had;
    return rc;
}

extern "C" int gab_bpm_last_stats(gab_bpm *h, int64_t *block_steps, int64_t *full_pairs, float *score_kernel_ms,
                                  float *total_ms) {
    GAB_CHECK(h, "gab_bpm_last_stats: NULL handle");
    GAB_CHECK(h->have_stats, "gab_bpm_last_stats: no completed run on this handle");
    gab_device_guard g(h->device);
    GAB_HIP(hipEventSynchronize(h->ev[3]));
    if (block_steps) *block_steps = (int64_t)(h->h_ct->steps + h->h_ct->full_steps);
    if (full_pairs) {
        int64_t nfull = h->last_full;
        for (int W = 1; W <= kMaxRegW; W++)
            for (int k = 0; k < kSlices; k++) nfull += h->h_ct->wl_slice[W][k];
        *full_pairs = nfull;
    }
    if (score_kernel_ms) GAB_HIP(hipEventElapsedTime(score_kernel_ms, h->ev[1], h->ev[2]));
    if (total_ms) GAB_HIP(hipEventElapsedTime(total_ms, h->ev[0], h->ev[3]));
    return GAB_OK;
}
