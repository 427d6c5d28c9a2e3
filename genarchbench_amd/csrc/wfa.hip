// wfa -- gap-affine wavefront alignment (WFA v1, complete mode and adaptive reduction) with CIGAR backtrace, on gfx950.
//
// Semantics: affine_wavefronts_align,
//   /root/reference/benchmarks/wfa/gap_affine/affine_wavefront_align.c:325-361 = loop over scores of
//   extend (affine_wavefront_extend.c:241-252) / end test (affine_wavefront_utils.c:83-102) /
//   next wavefront (affine_wavefront_align.c:41-321), then affine_wavefronts_backtrace
//   (affine_wavefront_backtrace.c:276-387).  Missing sources read as offset -10
//   (affine_wavefront.h:48); the strings behave as if padded with 'X' / 'Y'
//   (wfa/utils/string_padded.c:88-117).  Output = the un-run-length-encoded CIGAR operations the
//   driver prints (wfa/tools/align_benchmark.c:417-437, 499-504).
//   Adaptive mode (affine_wavefronts_new_reduced, affine_wavefront.c:162-181; --minimum-wavefront-length /
//   --maximum-difference-distance): after each extension the M wavefront drops the outer diagonals that lag more than
//   the threshold behind the best one (affine_wavefronts_reduce_wavefronts, affine_wavefront_extend.c:85-154); later
//   wavefronts are computed from the reduced [lo, hi], the backtrace still reads the allocated [lo_base, hi_base]
//   (affine_wavefront_backtrace.c:75-226).  It is a template variant (ADAPT) of the same code: the complete-mode
//   kernels carry none of it.
//
// Mapping: one wavefront (64 lanes) per pair, lanes = diagonals k.  The whole O(s^2) wavefront
// history of the pair -- needed by the backtrace -- lives in LDS as int16 offsets behind a small
// per-score directory, together with the two sequences (padded with 'X'/'Y' exactly like the
// reference, so the extension loop needs no length checks).  Extension compares four bases per
// step (two aligned ds_read_b32 + v_alignbyte per string, xor, ctz).  The backtrace runs
// wave-uniformly on the LDS history; match runs are written cooperatively.  A pair whose history
// does not fit the LDS pool (score above ~90 in the first pass) is queued and re-run with a bigger
// LDS pool, and finally by the same code with int32 offsets in a global scratch.
//
// Roofline: plen + tlen + |cigar| + 4 bytes of HBM traffic per pair; the work is latency-bound
// LDS traffic (sum over scores of the wavefront width + the extended matches).
#include "gab_internal.h"
#include <algorithm>
#include <new>
#include <vector>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>

namespace {

constexpr int kNull = -10;                 // AFFINE_WAVEFRONT_OFFSET_NULL
constexpr int kNone = 0x7fffffff;          // "no wavefront" marker in the directory
constexpr int kSeqPad = 32;                // physical 'X'/'Y' padding behind each LDS sequence (>= 16 + 3 for the 16-byte extension reads)
constexpr int kLdsMaxLen = 2040;           // longest sequence the LDS kernels accept

struct WfaPen { int32_t x, o, e, min_len, max_dist; };   // the last two: adaptive reduction parameters

struct WfaIO {
    const char *pat; const int64_t *pat_off; const int32_t *pat_len;
    const char *txt; const int64_t *txt_off; const int32_t *txt_len;
    int64_t pat_bytes, txt_bytes, n;
    char *ops; const int64_t *ops_off; int32_t *ops_len; int32_t *score;
};

struct WfaCounters {
    int32_t bad, first_bad;
    int32_t max_plen, max_tlen;       // over the pairs eligible for the LDS kernels
    uint32_t n_lds, n_big;            // eligible pairs / pairs too long for LDS
    uint32_t n_over;                  // pairs queued by the current pass of the global-history kernel
    uint32_t pad;
    uint32_t cursors[2];              // wfa_scatter
    unsigned long long work;          // wavefront cells computed + bases extended
    uint32_t tier_over[8];            // pairs queued by LDS tier k = the count tier k + 1 reads ON THE DEVICE (no host round trip)
};
GAB_STATIC_ATOMIC64(WfaCounters, work);

// One-byte offsets for the first LDS tier: value + 10 in a byte (null = -10 -> 0), good for offsets up to 245.  An offset
// never exceeds tlen + (number of score steps taken): extension stops at the padding and every later step adds at most 1.
struct OffB {
    uint8_t v;
    OffB() = default;
    __device__ __forceinline__ explicit OffB(int x) : v((uint8_t)(x + 10)) {}
    __device__ __forceinline__ explicit operator int() const { return (int)v - 10; }
};
constexpr int kOffBMax = 245;
constexpr int kOffBSafe = 240;       // an M offset above this sends the pair to the next tier (I = M + 1 of the same step is already stored)

// The work counter is statistics, but 250 000 waves of a 3 ms launch adding to ONE address serialise in L2 (~8 ns per
// atomic = 2 ms): the waves add to one of 256 slots, 128 bytes apart, behind the counters; wfa_sum_work folds them.
constexpr int kWorkSlots = 256, kWorkSlotStride = 16;           // in unsigned long long
constexpr size_t kCountersBytes = 256, kSlotsBytes = (size_t)kWorkSlots * kWorkSlotStride * 8;
__device__ __forceinline__ unsigned long long *wfa_work_slot(WfaCounters *ct) {
    return reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ct) + kCountersBytes) + (blockIdx.x & (kWorkSlots - 1)) * kWorkSlotStride;
}
__global__ __launch_bounds__(256) void wfa_sum_work(WfaCounters *ct) {
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ct) + kCountersBytes) + threadIdx.x * kWorkSlotStride;
    unsigned long long v = *slot;
    *slot = 0;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&ct->work, v);
}

// Per-score directory entry.  lo/hi (after reduction) and lob/hib (= lo_base/hi_base, as allocated) as in
// affine_wavefront_t; the bases index the offset pool such that diagonal k lives at base + (k - lo) -- a reduction that
// raises lo moves the bases along, so the computation never needs lob.
struct WfDir { int lo, hi, m, i, d, lob, hib; };

// The LDS kernels (int16 offsets, pools below 64 K entries) keep a directory entry in three dwords -- lo | hi, baseM |
// baseI, baseD as 16-bit fields, 0xffff = no wavefront -- which is 384 bytes of LDS less per pair than five ints and
// worth two more waves per CU; the global-memory kernel (int32 offsets, pools of millions) keeps five ints.
// ADAPT adds one dword (lob | hib) resp. two ints.
template <typename OffT, bool ADAPT>
struct WfStore {
    // the byte tier in complete mode: ONE dword per score -- lo | hi as signed bytes, baseM in 11 bits (0x7ff = none), two
    // flags for "has I" / "has D": the I and D wavefronts always follow M in the pool (baseI = baseM + width, baseD behind
    // it), so their bases need no storage.  384 bytes less per pair than three dwords at 48 scores, spent on the pool.
    static constexpr bool kCompact = sizeof(OffT) == 1 && !ADAPT;
    static constexpr int kDirInts = kCompact ? 1 : (sizeof(OffT) <= 2 ? 3 : 5) + (ADAPT ? (sizeof(OffT) <= 2 ? 1 : 2) : 0);
    static constexpr bool kPacked = sizeof(OffT) <= 2;
    OffT *pool;           // offsets
    int *dir;             // kDirInts ints per score
    int pool_cap, dir_cap, used;
    __device__ __forceinline__ WfDir get(int s) const {
        WfDir w;
        if (s < 0) { w.lo = w.lob = 1; w.hi = w.hib = -1; w.m = w.i = w.d = kNone; return w; }
        const int *p = dir + kDirInts * s;
        if (kCompact) {
            const uint32_t w0 = (uint32_t)p[0], mm = (w0 >> 16) & 0x7ffu;
            w.lo = (int)(int8_t)(w0 & 0xffu); w.hi = (int)(int8_t)((w0 >> 8) & 0xffu);
            const int width = w.hi - w.lo + 1;
            w.m = mm == 0x7ffu ? kNone : (int)mm;
            w.i = (w0 >> 27) & 1u ? (int)mm + width : kNone;
            w.d = (w0 >> 28) & 1u ? (int)mm + (((w0 >> 27) & 1u) ? 2 * width : width) : kNone;
        } else if (kPacked) {
            const uint32_t w0 = (uint32_t)p[0], w1 = (uint32_t)p[1], w2 = (uint32_t)p[2];
            w.lo = (int)(int16_t)(w0 & 0xffffu); w.hi = (int)w0 >> 16;
            const uint32_t m = w1 & 0xffffu, i = w1 >> 16, d = w2 & 0xffffu;
            w.m = m == 0xffffu ? kNone : (int)m; w.i = i == 0xffffu ? kNone : (int)i; w.d = d == 0xffffu ? kNone : (int)d;
            if (ADAPT) { w.lob = (int)(int16_t)((uint32_t)p[3] & 0xffffu); w.hib = p[3] >> 16; }
        } else {
            w.lo = p[0]; w.hi = p[1]; w.m = p[2]; w.i = p[3]; w.d = p[4];
            if (ADAPT) { w.lob = p[5]; w.hib = p[6]; }
        }
        if (!ADAPT) { w.lob = w.lo; w.hib = w.hi; }
        return w;
    }
    // (called by one lane)
    __device__ __forceinline__ void put(int s, int lo, int hi, int m, int i, int d) {
        int *p = dir + kDirInts * s;
        if (kCompact) {
            p[0] = (int)(((uint32_t)lo & 0xffu) | ((uint32_t)hi & 0xffu) << 8 | (m == kNone ? 0x7ffu : (uint32_t)m) << 16 |
                         (i != kNone ? 1u << 27 : 0u) | (d != kNone ? 1u << 28 : 0u));
        } else if (kPacked) {
            p[0] = (int)(((uint32_t)lo & 0xffffu) | (uint32_t)hi << 16);
            p[1] = (int)((m == kNone ? 0xffffu : (uint32_t)m) | (i == kNone ? 0xffffu : (uint32_t)i) << 16);
            p[2] = (int)(d == kNone ? 0xffffu : (uint32_t)d);
        } else { p[0] = lo; p[1] = hi; p[2] = m; p[3] = i; p[4] = d; }
    }
    // the allocated range of a new wavefront (called by one lane, ADAPT only)
    __device__ __forceinline__ void put_base(int s, int lob, int hib) {
        int *p = dir + kDirInts * s;
        if (kPacked) p[3] = (int)(((uint32_t)lob & 0xffffu) | (uint32_t)hib << 16);
        else { p[5] = lob; p[6] = hib; }
    }
    __device__ __forceinline__ int at(int base, int lo, int hi, int k) const {
        return (base != kNone && lo <= k && k <= hi) ? (int)pool[base + (k - lo)] : kNull;
    }
    // the backtrace's view: the allocated range
    __device__ __forceinline__ bool has_base(int base, const WfDir &w, int k) const { return base != kNone && w.lob <= k && k <= w.hib; }
    __device__ __forceinline__ int at_base(int base, const WfDir &w, int k) const {
        return has_base(base, w, k) ? (int)pool[base + (k - w.lo)] : kNull;
    }
};

__device__ __forceinline__ uint32_t lds_ld4(const uint8_t *base, int byte_off) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(base + (byte_off & ~3));
    return __builtin_amdgcn_alignbyte(w[1], w[0], (uint32_t)(byte_off & 3));
}
// eight bytes at any byte offset: three aligned dwords, two v_alignbyte
__device__ __forceinline__ uint64_t lds_ld8(const uint8_t *base, int byte_off) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(base + (byte_off & ~3));
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], sh = (uint32_t)(byte_off & 3);
    return (uint64_t)__builtin_amdgcn_alignbyte(w2, w1, sh) << 32 | __builtin_amdgcn_alignbyte(w1, w0, sh);
}
__device__ __forceinline__ uint32_t glb_ld4(const char *p) { uint32_t w; __builtin_memcpy(&w, p, 4); return w; }
// four bytes of a sequence at `pos` (a multiple of 4); bytes at or behind `len` read as `fill`.  Branch-free: the slab is
// readable up to the next multiple of 4 (counted from the SLAB's start) behind the sequence, so the last dword is fetched
// `d` bytes earlier when it would reach past that point (`end_lo` = the low two bits of the sequence's end offset in the
// slab) and shifted; the bytes behind the end are replaced by a mask.  (The byte loop this used to take for the tail was
// run by every wave.)
__device__ __forceinline__ uint32_t seq_ld4(const char *s, int pos, int len, uint32_t fill, int end_lo) {
    const int rem = len - pos;
    const int avail = rem + ((4 - end_lo) & 3);               // bytes from pos to the end of the readable range
    const int d = avail >= 4 ? 0 : 4 - avail;                 // 0 .. 3 (rem > 0 implies avail >= 1)
    uint32_t w = rem > 0 ? glb_ld4(s + pos - d) : 0u;
    w >>= 8 * d;
    const uint32_t m = rem >= 4 ? 0xffffffffu : rem <= 0 ? 0u : (1u << (8 * rem)) - 1u;
    return (w & m) | (fill * 0x01010101u & ~m);
}

// One pair, one wave.  LDSSEQ: P/T are LDS copies padded with kSeqPad bytes of 'X'/'Y';
// otherwise they are the global sequences (readable to a multiple of 4 bytes past the end).
// Returns false if the history did not fit (nothing has been written to the outputs then).
// G = lanes cooperating on the pair (64 = a whole wave, 16 = four pairs per wave); all G lanes run this function
// with identical control flow, other groups of the same wave may take different branches (SIMT divergence).
template <typename OffT, bool LDSSEQ, int G, bool ADAPT>
__device__ bool wfa_pair(WfStore<OffT, ADAPT> &st, const WfaPen pen, const uint8_t *P, int plen, const uint8_t *T, int tlen,
                         char *ops_global, char *ops_lds, int32_t *ops_len_out, int32_t *score_out, unsigned long long &work,
                         const uint8_t *step_tab = nullptr) {
    // the backtrace writes right-aligned into `ops`: an LDS buffer when the caller has one (then the CIGAR leaves the CU
    // once, left-aligned and coalesced), else the pair's own output region (shifted in place afterwards)
    char *ops = ops_lds ? ops_lds : ops_global;
    const int lane = threadIdx.x & (G - 1);
    const int x = pen.x, oe = pen.o + pen.e, e = pen.e;
    const int ak = tlen - plen;
    auto pch = [&](int v) -> int { return (v >= 0 && v < plen) ? (int)P[v] : (int)'X'; };
    auto tch = [&](int h) -> int { return (h >= 0 && h < tlen) ? (int)T[h] : (int)'Y'; };

    // score 0: M = {k = 0 -> offset 0}
    st.used = 0;
    if (st.dir_cap < 1 || st.pool_cap < 1) return false;
    if (step_tab) {
        // Which scores have a wavefront at all depends on the penalties alone (a wavefront is null iff its four sources
        // are, and a computed one is never empty), so the host lists them: step_tab[s] = distance to the next score with a
        // wavefront.  The loop below then visits only those (with x = 4, o + e = 8, e = 2: 0, 4, 8, 10, 12, ... -- half
        // of the iterations were null scores); every directory entry starts out null so that look-ups of the skipped
        // scores (s - x, s - o - e, s - e, and the backtrace's) read what the reference's NULL wavefronts say.
        for (int sc = lane; sc < st.dir_cap; sc += G) {
            st.put(sc, 1, -1, kNone, kNone, kNone);
            if (ADAPT) st.put_base(sc, 1, -1);
        }
        __syncthreads();
    }
    if (lane == 0) {
        st.put(0, 0, 0, 0, kNone, kNone); st.pool[0] = (OffT)0;
        if (ADAPT) st.put_base(0, 0, 0);
    }
    st.used = 1;
    __syncthreads();
    int score = 0;
    bool too_big = false;
    for (;;) {
        WfDir cur = st.get(score);
        if (cur.m != kNone) {
            // ---- extend every diagonal of M[score]
            for (int k = cur.lo + lane; k <= cur.hi; k += G) {
                int o = (int)st.pool[cur.m + (k - cur.lo)];
                int v = o - k, h = o;
                for (;;) {
                    if (LDSSEQ && v >= 0 && v <= plen && h >= 0 && h <= tlen) {
                        // sixteen bases per step (the 'X' / 'Y' padding behind the strings never matches)
                        uint4 qa, qb;                   // one unaligned 16-byte LDS read per string (see wfa_pair_static)
                        __builtin_memcpy(&qa, P + v, 16); __builtin_memcpy(&qb, T + h, 16);
                        const uint64_t d8 = ((uint64_t)(qa.y ^ qb.y) << 32) | (qa.x ^ qb.x), e8 = ((uint64_t)(qa.w ^ qb.w) << 32) | (qa.z ^ qb.z);
                        if ((d8 | e8) == 0) { o += 16; v += 16; h += 16; work += 16; continue; }
                        const int c8 = d8 ? __builtin_ctzll(d8) >> 3 : 8 + (__builtin_ctzll(e8) >> 3);
                        o += c8; work += c8;
                        break;
                    }
                    if (v >= 0 && v <= plen && h >= 0 && h <= tlen) {
                        uint32_t a, b;
                        if (LDSSEQ) { a = lds_ld4(P, v); b = lds_ld4(T, h); }
                        else {
                            // global strings have no physical padding: build the padded view
                            if (v + 4 <= plen && h + 4 <= tlen) { a = glb_ld4((const char *)P + v); b = glb_ld4((const char *)T + h); }
                            else {
                                a = (uint32_t)pch(v) | (uint32_t)pch(v + 1) << 8 | (uint32_t)pch(v + 2) << 16 | (uint32_t)pch(v + 3) << 24;
                                b = (uint32_t)tch(h) | (uint32_t)tch(h + 1) << 8 | (uint32_t)tch(h + 2) << 16 | (uint32_t)tch(h + 3) << 24;
                            }
                        }
                        const uint32_t d = a ^ b;
                        if (d == 0) { o += 4; v += 4; h += 4; work += 4; continue; }
                        const int c = __builtin_ctz(d) >> 3;
                        o += c; work += c;
                        break;
                    }
                    if (pch(v) == tch(h)) { o++; v++; h++; work++; continue; }
                    break;
                }
                st.pool[cur.m + (k - cur.lo)] = (OffT)o;
                if (sizeof(OffT) == 1) too_big = too_big || o > kOffBSafe;
            }
            __syncthreads();
            if (sizeof(OffT) == 1) {
                // one-byte offsets: an offset beyond the byte's range sends the pair to the next tier
                const uint64_t gm = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1)) << (threadIdx.x & 63 & ~(G - 1));
                if (__ballot(too_big) & gm) return false;
            }
            // ---- end reached?
            if (cur.lo <= ak && ak <= cur.hi && (int)st.pool[cur.m + (ak - cur.lo)] >= tlen) break;
            // ---- adaptive reduction (the reference reduces before the end test; the reduction never drops diagonal ak
            // and the backtrace reads the allocated range, so the order does not matter)
            if (ADAPT && cur.hi - cur.lo + 1 >= pen.min_len) {
                auto dist = [&](int k) {                    // affine_wavefronts_compute_distance, affine_wavefront_utils.c:64-74
                    const int o = (int)st.pool[cur.m + (k - cur.lo)];
                    return max(plen - (o - k), tlen - o);
                };
                int min_d = max(plen, tlen);
                for (int k = cur.lo + lane; k <= cur.hi; k += G) min_d = min(min_d, dist(k));
                for (int o = G / 2; o > 0; o >>= 1) min_d = min(min_d, __shfl_xor(min_d, o));
                // from the bottom: lo stops at the first diagonal within the threshold, at the latest at min(ak - 1, hi)
                const int top = min(ak - 1, cur.hi);
                int nlo = max(top, cur.lo);
                for (int k = cur.lo + lane; k < top; k += G)
                    if (dist(k) - min_d <= pen.max_dist) { nlo = min(nlo, k); break; }
                for (int o = G / 2; o > 0; o >>= 1) nlo = min(nlo, __shfl_xor(nlo, o));
                // from the top: hi stops at the first diagonal within the threshold, at the latest at max(ak + 1, lo)
                const int bottom = max(ak + 1, nlo);
                int nhi = min(bottom, cur.hi);
                for (int k = cur.hi - lane; k > bottom; k -= G)
                    if (dist(k) - min_d <= pen.max_dist) { nhi = max(nhi, k); break; }
                for (int o = G / 2; o > 0; o >>= 1) nhi = max(nhi, __shfl_xor(nhi, o));
                const int sh = nlo - cur.lo;
                if (lane == 0 && (sh | (cur.hi - nhi)))
                    st.put(score, nlo, nhi, cur.m + sh, cur.i == kNone ? kNone : cur.i + sh, cur.d == kNone ? kNone : cur.d + sh);
                __syncthreads();
            }
        }
        // ---- next wavefront
        score += step_tab ? (int)step_tab[score] : 1;
        if (score >= st.dir_cap) return false;
        const WfDir ms = st.get(score - x), mg = st.get(score - oe), ie = st.get(score - e);
        const int de_base = ie.d, ie_base = ie.i;              // I and D of score-e share lo/hi
        const bool n_ms = ms.m == kNone, n_mg = mg.m == kNone, n_ie = ie_base == kNone, n_de = de_base == kNone;
        if (n_ms && n_mg && n_ie && n_de) {
            if (lane == 0) {
                st.put(score, 1, -1, kNone, kNone, kNone);
                if (ADAPT) st.put_base(score, 1, -1);
            }
            __syncthreads();
            continue;
        }
        // a null source counts as lo = 1, hi = -1 (affine_wavefront.c:44-50)
        const int lo = min(min(n_ms ? 1 : ms.lo, n_mg ? 1 : mg.lo), min(n_ie ? 1 : ie.lo, n_de ? 1 : ie.lo)) - 1;
        const int hi = max(max(n_ms ? -1 : ms.hi, n_mg ? -1 : mg.hi), max(n_ie ? -1 : ie.hi, n_de ? -1 : ie.hi)) + 1;
        const int width = hi - lo + 1;
        const bool has_i = !n_mg || !n_ie, has_d = !n_mg || !n_de;
        const int need = width * (1 + (has_i ? 1 : 0) + (has_d ? 1 : 0));
        if (st.used + need > st.pool_cap) return false;
        const int bM = st.used, bI = has_i ? bM + width : kNone, bD = has_d ? bM + width * (has_i ? 2 : 1) : kNone;
        st.used += need;
        if (lane == 0) {
            st.put(score, lo, hi, bM, bI, bD);
            if (ADAPT) st.put_base(score, lo, hi);
        }
        for (int k = lo + lane; k <= hi; k += G) {
            int best = (!n_ms && ms.lo <= k && k <= ms.hi) ? (int)st.pool[ms.m + (k - ms.lo)] + 1 : kNull;
            if (has_i) {
                const int ins = max(st.at(mg.m, mg.lo, mg.hi, k - 1), st.at(ie_base, ie.lo, ie.hi, k - 1)) + 1;
                st.pool[bI + (k - lo)] = (OffT)ins;
                best = max(best, ins);
            }
            if (has_d) {
                const int del = max(st.at(mg.m, mg.lo, mg.hi, k + 1), st.at(de_base, ie.lo, ie.hi, k + 1));
                st.pool[bD + (k - lo)] = (OffT)del;
                best = max(best, del);
            }
            st.pool[bM + (k - lo)] = (OffT)best;
        }
        work += (lane == 0) ? (unsigned)width : 0u;
        __syncthreads();
    }

    // ---- backtrace (wave-uniform): ops written right-aligned into [0, cap), then shifted left
    const int cap = plen + tlen;
    int pos = cap - 1;
    int s = score, k = ak;
    {
        const WfDir w = st.get(s);
        int offset = (int)st.pool[w.m + (k - w.lo)];
        int type = 0;                                           // 0 = M, 1 = I, 2 = D
        auto valid_loc = [&](int kk, int oo) { return oo - kk > 0 && oo - kk <= plen && oo > 0 && oo <= tlen; };
        bool valid = valid_loc(k, offset);
        int v = offset - k, h = offset;
        auto put = [&](char c) { if (lane == 0 && pos >= 0) ops[pos] = c; pos--; };
        auto put_run = [&](char c, int cnt) {
            for (int i = lane; i < cnt && pos - i >= 0; i += G) ops[pos - i] = c;
            pos -= cnt > 0 ? cnt : 0;
        };
        while (v > 0 && h > 0 && s > 0) {
            if (!valid) {
                valid = valid_loc(k, offset);
                if (valid) {
                    if (k < ak) put_run('I', ak - k);
                    else if (k > ak) put_run('D', k - ak);
                }
            }
            const int s_go = s - oe, s_ge = s - e, s_mm = s - x;
            const WfDir wgo = st.get(s_go), wge = st.get(s_ge), wmm = st.get(s_mm);
            const int del_ext = type == 1 ? kNull : st.at_base(wge.d, wge, k + 1);
            const int del_open = type == 1 ? kNull : st.at_base(wgo.m, wgo, k + 1);
            const int ie_raw = st.at_base(wge.i, wge, k - 1), io_raw = st.at_base(wgo.m, wgo, k - 1);
            const bool ie_ok = st.has_base(wge.i, wge, k - 1);
            const bool io_ok = st.has_base(wgo.m, wgo, k - 1);
            const bool mm_ok = st.has_base(wmm.m, wmm, k);
            const int ins_ext = (type == 2 || !ie_ok) ? kNull : ie_raw + 1;
            const int ins_open = (type == 2 || !io_ok) ? kNull : io_raw + 1;
            const int misms = (type != 0 || !mm_ok) ? kNull : st.at_base(wmm.m, wmm, k) + 1;
            const int max_all = max(misms, max(max(ins_ext, ins_open), max(del_ext, del_open)));
            if (type == 0) { put_run('M', offset - max_all); offset = max_all; }
            if (max_all == del_ext) { if (valid) put('D'); s = s_ge; k++; type = 2; }
            else if (max_all == del_open) { if (valid) put('D'); s = s_go; k++; type = 0; }
            else if (max_all == ins_ext) { if (valid) put('I'); s = s_ge; k--; offset--; type = 1; }
            else if (max_all == ins_open) { if (valid) put('I'); s = s_go; k--; offset--; type = 0; }
            else { if (valid) put('X'); s = s_mm; offset--; }
            v = offset - k; h = offset;
        }
        if (s == 0) put_run('M', offset);
        else { put_run('D', v); put_run('I', h); }
    }
    // (writes in front of the buffer were dropped: a CIGAR longer than plen + tlen -- a sequence matching the other's padding
    // byte beyond its end, where the reference overflows its buffer -- keeps its last plen + tlen operations)
    pos = max(pos, -1) + 1;
    const int nops = cap - pos;
    __syncthreads();
    if (ops_lds) {
        for (int i = lane; i < nops; i += G) ops_global[i] = ops_lds[pos + i];
    } else if (pos > 0) {
        // shift left by `pos` bytes, G bytes per step (sources of a step are read before its stores)
        for (int c0 = 0; c0 < nops; c0 += G) {
            const int i = c0 + lane;
            char c = 0;
            if (i < nops) c = ops[pos + i];
            __syncthreads();
            if (i < nops) ops[i] = c;
            __syncthreads();
        }
    }
    if (lane == 0) { *ops_len_out = nops; *score_out = score; }
    return true;
}

// ---- pass 0: validate, count per path ------------------------------------------------------------
// Counts are accumulated per lane and reduced once per wave (atomics of every wave-iteration on one address serialise).
// When no pair is too long for the LDS kernels (the usual case) the first pass needs no id list: slot -> pair by identity.
__global__ __launch_bounds__(256) void wfa_classify(WfaIO io, WfaCounters *ct) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int mp = 0, mt = 0;
    uint32_t n_lds = 0, n_big = 0;
    for (; i < io.n; i += stride) {
        const int pl = io.pat_len[i], tl = io.txt_len[i];
        const int64_t po = io.pat_off[i], to = io.txt_off[i];
        const bool ok = pl >= 0 && tl >= 0 && pl <= GAB_WFA_MAX_LEN && tl <= GAB_WFA_MAX_LEN && po >= 0 && to >= 0 &&
                        io.ops_off[i] >= 0 && ((po + pl + 3) & ~3ll) <= io.pat_bytes && ((to + tl + 3) & ~3ll) <= io.txt_bytes;
        if (!ok) {
            atomicAdd(&ct->bad, 1);
            atomicMin((unsigned int *)&ct->first_bad, (unsigned int)(i + 1 > 0x7fffffff ? 0x7fffffff : i + 1));
            continue;
        }
        if (pl <= kLdsMaxLen && tl <= kLdsMaxLen) { n_lds++; mp = max(mp, pl); mt = max(mt, tl); } else n_big++;
    }
    for (int o = 32; o > 0; o >>= 1) {
        mp = max(mp, __shfl_xor(mp, o)); mt = max(mt, __shfl_xor(mt, o));
        n_lds += __shfl_xor(n_lds, o); n_big += __shfl_xor(n_big, o);
    }
    // one set of atomics per workgroup: they all hit the same four addresses and serialise in L2 (~8 ns each; 4096 waves
    // x 4 were 0.13 of this kernel's 0.15 ms)
    __shared__ int s_mp[4], s_mt[4];
    __shared__ uint32_t s_nl[4], s_nb[4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_mp[wv] = mp; s_mt[wv] = mt; s_nl[wv] = n_lds; s_nb[wv] = n_big; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) { mp = max(mp, s_mp[k]); mt = max(mt, s_mt[k]); n_lds += s_nl[k]; n_big += s_nb[k]; }
        atomicMax(&ct->max_plen, mp); atomicMax(&ct->max_tlen, mt);
        if (n_lds) atomicAdd(&ct->n_lds, n_lds);
        if (n_big) atomicAdd(&ct->n_big, n_big);
    }
}

// only when some pair is too long for the LDS kernels: the two id lists (the cursors start at zero)
__global__ __launch_bounds__(256) void wfa_scatter(WfaIO io, uint32_t *cursors, uint32_t *list_lds, uint32_t *list_big) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < io.n; i += stride) {
        const bool to_lds = io.pat_len[i] <= kLdsMaxLen && io.txt_len[i] <= kLdsMaxLen;
        const uint32_t s_lds = gab_wave_slot(&cursors[0], to_lds), s_big = gab_wave_slot(&cursors[1], !to_lds);
        if (to_lds) list_lds[s_lds] = (uint32_t)i; else list_big[s_big] = (uint32_t)i;
    }
}

// ---- LDS kernel: one G-lane group per pair, 64 / G pairs per wave (= per workgroup) ----------------
// dynamic LDS per group: [dir: 3 (4 if ADAPT) * dir_cap ints][P: seqp bytes][T: seqt bytes][pool: pool_cap int16]; the CIGAR is built over P/T
template <int G, bool ADAPT, typename OffT>
__global__ __launch_bounds__(64) void wfa_lds(WfaIO io, WfaPen pen, const uint32_t *__restrict__ list, uint32_t count,
                                              const uint32_t *__restrict__ count_ptr, int dir_cap, int seqp, int seqt, int pool_cap,
                                              uint32_t group_bytes, uint32_t *over_list, uint32_t *over_count, WfaCounters *ct,
                                              const uint8_t *__restrict__ steps) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_all[];
    constexpr int kGroups = 64 / G;
    const int grp = threadIdx.x / G, lane = threadIdx.x & (G - 1);
    if (count_ptr) count = *count_ptr;                       // the previous tier's overflow count, read here: no host round trip
    unsigned long long work_all = 0;
    // the score-step table of this penalty set (see wfa_pair), shared by the groups of the wave
    const int tab_bytes = (dir_cap + 15) & ~15;
    for (int i = threadIdx.x; i < dir_cap; i += 64) smem_all[i] = steps[i];
    __syncthreads();
    // (a launch behind another tier is sized by an estimate of that tier's overflow: the waves stride over whatever it left)
    for (uint32_t b0 = blockIdx.x * kGroups; b0 < count; b0 += gridDim.x * kGroups) {
    const uint32_t b = b0 + grp;
    unsigned long long work = 0;
    bool ok = true, have = b < count;
    uint32_t id = 0;
    if (have) {
        id = list ? list[b] : b;                             // no list: every pair of the batch is in this pass
        uint8_t *smem = smem_all + tab_bytes + (size_t)grp * group_bytes;
        int *dir = reinterpret_cast<int *>(smem);
        uint8_t *P = smem + (size_t)dir_cap * 4 * WfStore<OffT, ADAPT>::kDirInts;
        uint8_t *T = P + seqp;
        // the backtrace never looks at the strings (affine_wavefronts_backtrace_matches__check only counts), so their
        // LDS region doubles as the CIGAR buffer: plen + tlen <= seqp + seqt bytes
        char *opsbuf = reinterpret_cast<char *>(P);
        OffT *pool = reinterpret_cast<OffT *>(T + seqt);
        const int plen = io.pat_len[id], tlen = io.txt_len[id];
        const char *gp = io.pat + io.pat_off[id], *gt = io.txt + io.txt_off[id];
        // four bytes per lane and step (the LDS copies are dword-aligned and have room to the next multiple of four)
        const int pend = (int)((io.pat_off[id] + plen) & 3), tend = (int)((io.txt_off[id] + tlen) & 3);
        for (int i = 4 * lane; i < plen + kSeqPad; i += 4 * G) *reinterpret_cast<uint32_t *>(P + i) = seq_ld4(gp, i, plen, (uint32_t)'X', pend);
        for (int i = 4 * lane; i < tlen + kSeqPad; i += 4 * G) *reinterpret_cast<uint32_t *>(T + i) = seq_ld4(gt, i, tlen, (uint32_t)'Y', tend);
        __syncthreads();
        WfStore<OffT, ADAPT> st;
        st.pool = pool; st.dir = dir; st.pool_cap = pool_cap; st.dir_cap = dir_cap; st.used = 0;
        ok = wfa_pair<OffT, true, G, ADAPT>(st, pen, P, plen, T, tlen, io.ops + io.ops_off[id], opsbuf, io.ops_len + id, io.score + id, work,
                                               smem_all);
        if (!ok && lane == 0) over_list[atomicAdd(over_count, 1u)] = id;
    }
    if (have && ok) work_all += work;
    __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) work_all += __shfl_xor(work_all, o);
    if (threadIdx.x == 0 && work_all) atomicAdd(wfa_work_slot(ct), work_all);
}


// ---- complete mode, first tier: the directory as a table of the penalties ------------------------------------------
// Without the adaptive reduction nothing in a pair's directory depends on the data: whether M / I / D of a score exist,
// their [lo, hi] (min / max of the sources' -1 / +1) and where they land in the pool (allocation in score order) follow
// from the penalties alone.  gab_wfa_create lists one WfRow per score that has a wavefront, with the resolved facts of
// its three source scores; the kernel then needs no directory in LDS (the room goes to the offset pool), no look-ups and
// no decoding: all groups of a wave walk the rows in lockstep, so the row index is wave-uniform and the row arrives by
// scalar loads.  A missing source is an empty range (lo = 1, hi = -1): every read of it is out of range = null.
struct WfRow {
    int score, lo, hi, bM, bI, bD, used_end, has_gap;     // bI / bD valid iff has_gap (I and D exist at the same scores)
    int ms_m, ms_lo, ms_hi;                               // M[score - x]
    int mg_m, mg_lo, mg_hi;                               // M[score - o - e]
    int ie_i, ie_d, ie_lo, ie_hi;                         // I / D[score - e]
    int r_ms, r_mg, r_ie;                                 // rows of the three source scores (0 when there is none)
    int pad[3];
};
static_assert(sizeof(WfRow) == 96, "WfRow is read as dwords");

// a workgroup of the LDS kernels is ONE wave: LDS instructions of a wave execute in order, so a store by one lane is seen by a
// later load of another lane without draining the queue -- only the compiler must not reorder them
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int G>
__device__ bool wfa_pair_static(OffB *pool, int pool_cap, const WfRow *__restrict__ rows, int nrows, const WfaPen pen,
                                const uint8_t *P, int plen, const uint8_t *T, int tlen, char *ops_global, char *ops,
                                int32_t *ops_len_out, int32_t *score_out, unsigned long long &work, int r_start, int &resume_row) {
    // r_start > 0: the pool already holds the wavefronts of rows 0 .. r_start - 1 (a smaller tier ran out of room there);
    // resume_row > 0 on a false return: the same for the next tier (rows 0 .. resume_row - 1 are complete in `pool`)
    resume_row = 0;
    const int lane = threadIdx.x & (G - 1);
    const int ak = tlen - plen;
    if (pool_cap < 1 || nrows < 1) return false;
    auto at = [&](int base, int lo, int hi, int k) { return (lo <= k && k <= hi) ? (int)pool[base + (k - lo)] : kNull; };
    // affine_wavefronts_extend_mwavefront_compute (the padded strings: affine_wavefront_extend.c:51-83), sixteen bases at a time:
    // the number of leading bases of P[v ..] and T[h ..] that match, 0 .. 16 (outside the strings 'X' meets 'Y': 0 -- the kernel
    // keeps pairs that contain the other's padding byte out of this tier; the LDS copies carry kSeqPad bytes of padding)
    auto match16 = [&](int v, int h) -> int {
        if (!(v >= 0 && v <= plen && h >= 0 && h <= tlen)) return 0;
        // one 16-byte LDS read per string at ANY byte address (gfx950 reads LDS unaligned; aligned dwords + v_alignbyte cost
        // three times the instructions)
        uint4 a, b;
        __builtin_memcpy(&a, P + v, 16); __builtin_memcpy(&b, T + h, 16);
        const uint64_t d8 = ((uint64_t)(a.y ^ b.y) << 32) | (a.x ^ b.x), e8 = ((uint64_t)(a.w ^ b.w) << 32) | (a.z ^ b.z);
        if ((d8 | e8) == 0) return 16;
        return d8 ? __builtin_ctzll(d8) >> 3 : 8 + (__builtin_ctzll(e8) >> 3);
    };
    const int wl = threadIdx.x & 63, gbase = wl & ~(G - 1);
    const uint64_t group_mask = (G == 64 ? ~0ull : ((1ull << G) - 1)) << gbase;
    // Extension of the diagonals of one pass (lane = diagonal k, offset o; `active` lanes only).  Every lane compares the first
    // sixteen bases of its own diagonal; most stop there.  A diagonal that goes on -- usually the one the alignment follows,
    // for ~50 bases -- used to keep its lane looping alone while the others waited; now the G lanes of the group take G
    // consecutive 16-base chunks of it at once and a vote finds the first mismatch: one round per such diagonal.
    auto extend_all = [&](int &o, int k, bool active) {
        const int c0 = active ? match16(o - k, o) : 0;
        o += c0; work += (unsigned)c0;
        bool cont = active && c0 == 16;
        for (;;) {
            const uint64_t cm = __ballot(cont);
            if (cm == 0) break;                                              // (wave-uniform: no group has a diagonal going on)
            const uint64_t gm = cm & group_mask;
            const int src = gm ? (int)__builtin_ctzll(gm) : wl;              // the group's first such lane
            const int so = __shfl(o, src), sk = __shfl(k, src);
            const int c = gm ? match16(so - sk + 16 * lane, so + 16 * lane) : 0;
            const uint64_t fm = __ballot(c < 16) & group_mask;               // the first chunk that does not match through
            const int jl = fm ? (int)__builtin_ctzll(fm) : gbase + G - 1;
            const int cj = __shfl(c, jl);
            const int total = fm ? 16 * (jl - gbase) + cj : 16 * G;
            if (gm && wl == src) { o += total; work += (unsigned)total; cont = fm == 0; }
        }
    };
    // Every wavefront is extended by the lane that computed it, before it is stored: a score step is one pass -- read the
    // sources, max, extend, store -- with one synchronisation, and the end test is a vote on the register of diagonal ak.
    int r = r_start > 0 ? r_start - 1 : 0;
    bool at_end = false, too_big = false;
    if (r_start <= 0) {
        int o = 0;
        extend_all(o, 0, lane == 0);
        if (lane == 0) {
            pool[0] = OffB(o);
            at_end = ak == 0 && o >= tlen;
            too_big = o > kOffBSafe;
        }
    }
    for (;;) {
        // an offset beyond the byte's range (a pattern full of the text's padding byte can run past the end of the text):
        // the pair goes to the int16 tier
        if (__ballot(too_big) & group_mask) return false;
        if (__ballot(at_end) & group_mask) break;
        wave_sync();
        r++;
        if (r >= nrows) return false;
        const WfRow &n = rows[__builtin_amdgcn_readfirstlane(r)];
        if (n.used_end > pool_cap) { resume_row = r; return false; }
        for (int k0 = n.lo; k0 <= n.hi; k0 += G) {                         // (n.lo, n.hi are wave-uniform: every lane takes every trip)
            const int k = k0 + lane;
            const bool active = k <= n.hi;
            int best = kNull;
            if (active) {
                best = (n.ms_lo <= k && k <= n.ms_hi) ? (int)pool[n.ms_m + (k - n.ms_lo)] + 1 : kNull;
                if (n.has_gap) {
                    const int ins = max(at(n.mg_m, n.mg_lo, n.mg_hi, k - 1), at(n.ie_i, n.ie_lo, n.ie_hi, k - 1)) + 1;
                    const int del = max(at(n.mg_m, n.mg_lo, n.mg_hi, k + 1), at(n.ie_d, n.ie_lo, n.ie_hi, k + 1));
                    pool[n.bI + (k - n.lo)] = OffB(ins);
                    pool[n.bD + (k - n.lo)] = OffB(del);
                    best = max(best, max(ins, del));
                }
            }
            extend_all(best, k, active);
            if (active) {
                pool[n.bM + (k - n.lo)] = OffB(best);
                at_end = at_end || (k == ak && best >= tlen);
                too_big = too_big || best > kOffBSafe;
            }
        }
        work += (lane == 0) ? (unsigned)(n.hi - n.lo + 1) : 0u;
    }
    wave_sync();

    // ---- backtrace: as in wfa_pair, the directory entries of s - o - e, s - e, s - x being the source fields of row(s)
    const int cap = plen + tlen;
    int pos = cap - 1;
    int k = ak;
    const int x = pen.x, oe = pen.o + pen.e, e = pen.e;
    const int score = rows[r].score;
    {
        int s = score;
        int offset = (int)pool[rows[r].bM + (k - rows[r].lo)];
        int type = 0;                                           // 0 = M, 1 = I, 2 = D
        auto valid_loc = [&](int kk, int oo) { return oo - kk > 0 && oo - kk <= plen && oo > 0 && oo <= tlen; };
        bool valid = valid_loc(k, offset);
        int v = offset - k, h = offset;
        // the buffer (the strings' LDS, no longer needed) is filled with 'M' once, 16 bytes per lane and store: a run of matches
        // -- most of the CIGAR -- then only moves the cursor
        for (int i = 16 * lane; i < cap; i += 16 * G) *reinterpret_cast<uint4 *>(ops + i) = make_uint4(0x4d4d4d4du, 0x4d4d4d4du, 0x4d4d4d4du, 0x4d4d4d4du);
        wave_sync();
        auto put = [&](char ch) { if (lane == 0 && pos >= 0) ops[pos] = ch; pos--; };
        auto put_run = [&](char ch, int cnt) {
            for (int i = lane; i < cnt && pos - i >= 0; i += G) ops[pos - i] = ch;
            pos -= cnt > 0 ? cnt : 0;
        };
        auto skip_matches = [&](int cnt) { pos -= cnt > 0 ? cnt : 0; };
        auto in = [&](int lo, int hi, int kk) { return lo <= kk && kk <= hi; };
        while (v > 0 && h > 0 && s > 0) {
            if (!valid) {
                valid = valid_loc(k, offset);
                if (valid) {
                    if (k < ak) put_run('I', ak - k);
                    else if (k > ak) put_run('D', k - ak);
                }
            }
            const WfRow w = rows[r];
            const int del_ext = type == 1 ? kNull : at(w.ie_d, w.ie_lo, w.ie_hi, k + 1);
            const int del_open = type == 1 ? kNull : at(w.mg_m, w.mg_lo, w.mg_hi, k + 1);
            const bool ie_ok = in(w.ie_lo, w.ie_hi, k - 1), io_ok = in(w.mg_lo, w.mg_hi, k - 1), mm_ok = in(w.ms_lo, w.ms_hi, k);
            const int ins_ext = (type == 2 || !ie_ok) ? kNull : (int)pool[w.ie_i + (k - 1 - w.ie_lo)] + 1;
            const int ins_open = (type == 2 || !io_ok) ? kNull : (int)pool[w.mg_m + (k - 1 - w.mg_lo)] + 1;
            const int misms = (type != 0 || !mm_ok) ? kNull : (int)pool[w.ms_m + (k - w.ms_lo)] + 1;
            const int max_all = max(misms, max(max(ins_ext, ins_open), max(del_ext, del_open)));
            if (type == 0) { skip_matches(offset - max_all); offset = max_all; }
            if (max_all == del_ext) { if (valid) put('D'); s -= e; r = w.r_ie; k++; type = 2; }
            else if (max_all == del_open) { if (valid) put('D'); s -= oe; r = w.r_mg; k++; type = 0; }
            else if (max_all == ins_ext) { if (valid) put('I'); s -= e; r = w.r_ie; k--; offset--; type = 1; }
            else if (max_all == ins_open) { if (valid) put('I'); s -= oe; r = w.r_mg; k--; offset--; type = 0; }
            else { if (valid) put('X'); s -= x; r = w.r_ms; offset--; }
            v = offset - k; h = offset;
        }
        if (s == 0) skip_matches(offset);
        else { put_run('D', v); put_run('I', h); }
    }
    // (writes in front of the buffer were dropped: a CIGAR longer than plen + tlen -- a sequence matching the other's padding
    // byte beyond its end, where the reference overflows its buffer -- keeps its last plen + tlen operations)
    pos = max(pos, -1) + 1;
    const int nops = cap - pos;
    wave_sync();
    // the CIGAR leaves the CU four bytes per lane and store (unaligned LDS read, unaligned global store), its last 0 .. 3 bytes singly
    {
        const int n4 = nops & ~3;
        for (int i = 4 * lane; i < n4; i += 4 * G) { uint32_t w4; __builtin_memcpy(&w4, ops + pos + i, 4); __builtin_memcpy(ops_global + i, &w4, 4); }
        if (lane < nops - n4) ops_global[n4 + lane] = ops[pos + n4 + lane];
    }
    if (lane == 0) { *ops_len_out = nops; *score_out = score; }
    return true;
}

// dynamic LDS per group: [P: seqp bytes][T: seqt bytes][pool: pool_cap bytes]; the CIGAR is built over P/T
// What a static tier leaves behind for a pair it ran out of room for: the row to resume at and the work done so far; the pool
// bytes (rows 0 .. row - 1, the same layout in every static tier) sit in a slot of `save_pool`.  row = 0: start over.
struct WfResume { int32_t row; uint32_t pad; unsigned long long work; };

template <int G, bool CHAINED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 7))) void wfa_lds_static(WfaIO io, WfaPen pen, const uint32_t *__restrict__ list, uint32_t count,
                                                     const uint32_t *__restrict__ count_ptr, int seqp, int seqt, int pool_cap,
                                                     uint32_t group_bytes, uint32_t *over_list, uint32_t *over_count, WfaCounters *ct,
                                                     const WfRow *__restrict__ rows, int nrows,
                                                     const WfResume *__restrict__ in_hdr, const uint8_t *__restrict__ in_pool, uint32_t in_slots, uint32_t in_slot_bytes,
                                                     WfResume *out_hdr, uint8_t *out_pool, uint32_t out_slots, uint32_t out_slot_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_all[];
    constexpr int kGroups = 64 / G;
    const int grp = threadIdx.x / G, lane = threadIdx.x & (G - 1);
    if (CHAINED) count = *count_ptr;                         // see wfa_lds; CHAINED is a compile-time variant because the loop costs
    unsigned long long work_all = 0;                         // the first launch 21 VGPRs = a fifth of its waves
    uint32_t b0 = blockIdx.x * kGroups;
    if (b0 >= count) return;
    do {
    const uint32_t b = b0 + grp;
    unsigned long long work = 0;
    bool ok = true, have = b < count;
    uint32_t id = 0;
    if (have) {
        id = list ? list[b] : b;
        uint8_t *P = smem_all + (size_t)grp * group_bytes, *T = P + seqp;
        OffB *pool = reinterpret_cast<OffB *>(T + seqt);
        const int plen = io.pat_len[id], tlen = io.txt_len[id];
        const char *gp = io.pat + io.pat_off[id], *gt = io.txt + io.txt_off[id];
        // A pattern byte 'Y' or a text byte 'X' could match the OTHER string's padding outside the strings; wfa_pair_static
        // does not look there, so such a pair (never a DNA read) goes to the general kernel.
        auto has_byte = [](uint32_t w, uint32_t c) { const uint32_t x = w ^ (c * 0x01010101u); return ((x - 0x01010101u) & ~x & 0x80808080u) != 0; };
        bool pad_byte = false;
        const int pend = (int)((io.pat_off[id] + plen) & 3), tend = (int)((io.txt_off[id] + tlen) & 3);
        for (int i = 4 * lane; i < plen + kSeqPad; i += 4 * G) {
            const uint32_t w = seq_ld4(gp, i, plen, (uint32_t)'X', pend);
            *reinterpret_cast<uint32_t *>(P + i) = w;
            pad_byte = pad_byte || has_byte(w, 'Y');
        }
        for (int i = 4 * lane; i < tlen + kSeqPad; i += 4 * G) {
            const uint32_t w = seq_ld4(gt, i, tlen, (uint32_t)'Y', tend);
            *reinterpret_cast<uint32_t *>(T + i) = w;
            pad_byte = pad_byte || has_byte(w, 'X');
        }
        // a pair the previous static tier ran out of room for continues where it stopped: its wavefronts come back from the slot
        int r_start = 0;
        if (CHAINED && in_hdr && b < in_slots) {
            const WfResume hd = in_hdr[b];
            if (hd.row > 0) {
                r_start = hd.row;
                const int bytes = rows[hd.row - 1].used_end;                 // (OffB = one byte per offset)
                const uint32_t *src = reinterpret_cast<const uint32_t *>(in_pool + (size_t)b * in_slot_bytes);
                for (int i = lane; i * 4 < bytes; i += G) reinterpret_cast<uint32_t *>(pool)[i] = src[i];
                if (lane == 0) work = hd.work;
            }
        }
        const uint64_t gm = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1)) << (threadIdx.x & 63 & ~(G - 1));
        wave_sync();
        int resume_row = 0;
        // wfa_pair_static fetches its rows by scalar loads: the row index must be the same in every group that is inside it
        // together.  Groups of a chained launch can start at different rows (a resumed pair beside one that starts over: a
        // slot index past in_slots, a pair sent on for its padding bytes or a too-large offset); such a wave runs its groups
        // one after the other -- each alone in the call, so "the first active lane's row" is its own.
        bool mixed = false;
        if (CHAINED) mixed = __ballot(r_start != __builtin_amdgcn_readfirstlane(r_start)) != 0;
        for (int turn = 0; turn < (mixed ? kGroups : 1); turn++) {
            if (mixed && grp != turn) continue;
            if (__ballot(pad_byte) & gm) ok = false;
            else
            ok = wfa_pair_static<G>(pool, pool_cap, rows, nrows, pen, P, plen, T, tlen, io.ops + io.ops_off[id], reinterpret_cast<char *>(P),
                                    io.ops_len + id, io.score + id, work, r_start, resume_row);
        }
        if (!ok) {
            uint32_t slot = 0;
            if (lane == 0) { slot = atomicAdd(over_count, 1u); over_list[slot] = id; }
            if (out_hdr) {
                slot = __shfl(slot, (int)(threadIdx.x & 63 & ~(G - 1)));
                unsigned long long gw = work;
                for (int o = G / 2; o > 0; o >>= 1) gw += __shfl_xor(gw, o);
                const bool keep = resume_row > 0 && slot < out_slots;
                if (keep) {
                    const int bytes = rows[resume_row - 1].used_end;
                    uint32_t *dst = reinterpret_cast<uint32_t *>(out_pool + (size_t)slot * out_slot_bytes);
                    for (int i = lane; i * 4 < bytes; i += G) dst[i] = reinterpret_cast<const uint32_t *>(pool)[i];
                }
                if (lane == 0) { WfResume hd; hd.row = keep ? resume_row : 0; hd.pad = 0; hd.work = keep ? gw : 0; out_hdr[slot] = hd; }
            }
        }
    }
    if (have && ok) work_all += work;
    wave_sync();
    } while (CHAINED && (b0 += gridDim.x * kGroups) < count);
    for (int o = 32; o > 0; o >>= 1) work_all += __shfl_xor(work_all, o);
    if (threadIdx.x == 0 && work_all) atomicAdd(wfa_work_slot(ct), work_all);
}

// ---- global kernel: int32 history in a scratch slab, any length ------------------------------
template <bool ADAPT>
__global__ __launch_bounds__(64) void wfa_global(WfaIO io, WfaPen pen, const uint32_t *__restrict__ list, uint32_t count_arg,
                                                 const uint32_t *__restrict__ count_ptr, int32_t *scratch, int64_t per_block, int dir_cap, int pool_cap,
                                                 uint32_t *over_list, WfaCounters *ct) {
    const int lane = threadIdx.x;
    const uint32_t count = count_ptr ? *count_ptr : count_arg;      // count_ptr: what the last LDS tier left, read here (no host round trip)
    int32_t *mine = scratch + (int64_t)blockIdx.x * per_block;
    for (uint32_t b = blockIdx.x; b < count; b += gridDim.x) {
        const uint32_t id = list ? list[b] : b;
        const int plen = io.pat_len[id], tlen = io.txt_len[id];
        WfStore<int32_t, ADAPT> st;
        st.dir = mine; st.pool = mine + (int64_t)dir_cap * WfStore<int32_t, ADAPT>::kDirInts; st.pool_cap = pool_cap; st.dir_cap = dir_cap; st.used = 0;
        unsigned long long work = 0;
        const bool ok = wfa_pair<int32_t, false, 64, ADAPT>(st, pen, (const uint8_t *)(io.pat + io.pat_off[id]), plen,
                                                 (const uint8_t *)(io.txt + io.txt_off[id]), tlen,
                                                 io.ops + io.ops_off[id], nullptr, io.ops_len + id, io.score + id, work);
        if (!ok && lane == 0) over_list[atomicAdd(&ct->n_over, 1u)] = id;
        for (int o = 32; o > 0; o >>= 1) work += __shfl_xor(work, o);
        if (lane == 0 && ok) atomicAdd(wfa_work_slot(ct), work);
        __syncthreads();
    }
}

using WfaLdsKernel = void (*)(WfaIO, WfaPen, const uint32_t *, uint32_t, const uint32_t *, int, int, int, int, uint32_t, uint32_t *, uint32_t *, WfaCounters *, const uint8_t *);
// the LDS kernel for G lanes per pair; byte_offsets: the one-byte history of the first tier
WfaLdsKernel wfa_lds_kernel(int G, bool adaptive, bool byte_offsets) {
#define GAB_WFA_K(g) (byte_offsets ? (adaptive ? wfa_lds<g, true, OffB> : wfa_lds<g, false, OffB>) : (adaptive ? wfa_lds<g, true, int16_t> : wfa_lds<g, false, int16_t>))
    return G == 8 ? GAB_WFA_K(8) : G == 16 ? GAB_WFA_K(16) : G == 32 ? GAB_WFA_K(32) : GAB_WFA_K(64);
#undef GAB_WFA_K
}
bool wfa_lds_allow_big() {
    for (int G : {8, 16, 32, 64})
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++)
                if (hipFuncSetAttribute((const void *)wfa_lds_kernel(G, a != 0, b != 0), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    return true;
}

// ---- packed output (gab_wfa_run_packed): the CIGAR as the driver PRINTS it ----------------------------------------------
// edit_cigar_print (wfa/gap_affine/edit_cigar.c:184-200) writes every run of equal operations as "%d%c".  One thread per
// pair reads the pair's operations (eight bytes per load), sizes its text, gets its place in the output by a scan of the
// workgroup and ONE atomic add per workgroup, and writes the text there: ~20 bytes per 151-bp pair instead of the 302 bytes
// of operation room -- what goes back over the bus shrinks fifteen-fold.  (The layout in `out` is by workgroup arrival;
// out_off / out_len say where each pair's text is.)
__device__ __forceinline__ int wfa_dec_digits(uint32_t v) { return 1 + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u); }

template <bool WRITE>
__device__ __forceinline__ int wfa_rle_pass(const char *__restrict__ o, int nops, char *dst) {
    int len = 0;
    uint32_t last = 0, run = 0;
    for (int k = 0; k < nops; k += 8) {
        unsigned long long w;
        __builtin_memcpy(&w, o + k, 8);                            // (unaligned; the room is readable 8 bytes past the last operation)
        const int m = min(8, nops - k);
        for (int j = 0; j < m; j++, w >>= 8) {
            const uint32_t c = (uint32_t)(w & 0xff);
            if (c != last && run) {
                const int d = wfa_dec_digits(run);
                if (WRITE) { uint32_t v = run; for (int q = d - 1; q >= 0; q--, v /= 10u) dst[len + q] = (char)('0' + v % 10u); dst[len + d] = (char)last; }
                len += d + 1;
                run = 0;
            }
            last = c; run++;
        }
    }
    if (run) {
        const int d = wfa_dec_digits(run);
        if (WRITE) { uint32_t v = run; for (int q = d - 1; q >= 0; q--, v /= 10u) dst[len + q] = (char)('0' + v % 10u); dst[len + d] = (char)last; }
        len += d + 1;
    }
    return len;
}

__global__ __launch_bounds__(256) void wfa_rle_pack(const char *__restrict__ ops, const int64_t *__restrict__ ops_off, const int32_t *__restrict__ ops_len,
                                                    uint32_t n, char *__restrict__ out, unsigned long long cap, unsigned long long *cursor,
                                                    int64_t *__restrict__ out_off, int32_t *__restrict__ out_len) {
    __shared__ uint32_t s_wave[4];
    __shared__ unsigned long long s_base;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const char *o = nullptr;
    int nops = 0, len = 0;
    if (i < n) { o = ops + ops_off[i]; nops = ops_len[i]; len = wfa_rle_pass<false>(o, nops, nullptr); }
    // exclusive scan of `len` over the workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = (uint32_t)len;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < 4; w++) { if (w < wave) before += s_wave[w]; total += s_wave[w]; }
    if (threadIdx.x == 0) s_base = total ? atomicAdd(cursor, (unsigned long long)total) : 0ull;
    __syncthreads();
    if (i >= n) return;
    const unsigned long long at = s_base + before + (incl - (uint32_t)len);
    out_off[i] = (int64_t)at; out_len[i] = len;
    if (at + (unsigned long long)len <= cap && len) wfa_rle_pass<true>(o, nops, out + at);       // (else: the host reports GAB_ERANGE)
}

__global__ __launch_bounds__(256) void wfa_fill_stride(int64_t *off, uint32_t n, int64_t stride) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) off[i] = (int64_t)i * stride;
}

}  // namespace

// =============================================================================== host side
struct gab_wfa {
    gab_tuning tun = gab_tuning_loaded();      // experiment knobs, read when the handle is made
    gab_host_stream hs;     // private stream of the host-pointer entry point(s)
    int device = 0;
    WfaPen pen;
    bool adaptive = false;  // affine_wavefronts_new_reduced instead of _new_complete
    gab_devbuf ws;          // counters | 3 id lists
    gab_devbuf rows;        // complete mode: the directory of every pair as a function of the penalties (WfRow per score with a wavefront)
    int nrows = 0;
    gab_devbuf steps;       // per score: distance to the next score that has a wavefront (a function of the penalties)
    gab_devbuf scratch;     // global-kernel history
    gab_devbuf io;          // staging for the host-pointer entry point
    size_t scratch_budget = (size_t)8 << 30;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    WfaCounters *h_ct = nullptr;
    bool have_stats = false;
    int64_t last_requeued = 0;
};

extern "C" int gab_wfa_create(const gab_wfa_penalties *p, int device, gab_wfa **out) {
    return gab_wfa_create_reduced(p, -1, -1, device, out);
}

extern "C" int gab_wfa_create_reduced(const gab_wfa_penalties *p, int min_wavefront_length, int max_distance_threshold,
                                      int device, gab_wfa **out) {
    if (!p || !out) { gab_set_error("gab_wfa_create: NULL argument"); return GAB_EINVAL; }
    *out = nullptr;
    GAB_CHECK(p->mismatch > 0 && p->gap_opening > 0 && p->gap_extension > 0 && p->mismatch < 4096 &&
              p->gap_opening < 4096 && p->gap_extension < 4096,
              "gab_wfa_create: penalties must be strictly positive (X=%d,O=%d,E=%d), as affine_penalties_mzero demands",
              p->mismatch, p->gap_opening, p->gap_extension);
    int rc = gab_check_device(device);
    if (rc) return rc;
    gab_device_guard g(device);
    gab_wfa *h = new (std::nothrow) gab_wfa();
    if (!h) { gab_set_error("out of host memory"); return GAB_ENOMEM; }
    h->device = device;
    h->pen.x = p->mismatch; h->pen.o = p->gap_opening; h->pen.e = p->gap_extension;
    h->adaptive = min_wavefront_length >= 0;          // align_benchmark.c:359-368
    h->pen.min_len = min_wavefront_length; h->pen.max_dist = max_distance_threshold;
    for (int k = 0; k < 4; k++)
        if (hipEventCreate(&h->ev[k]) != hipSuccess) { gab_set_error("hipEventCreate failed"); delete h; return GAB_EDEVICE; }
    static_assert(sizeof(WfaCounters) <= 128, "the packed-output cursor sits at byte 128 of the pinned block");
    if (hipHostMalloc((void **)&h->h_ct, 256) != hipSuccess ||
        !wfa_lds_allow_big()) {
        gab_set_error("gab_wfa_create: pinned allocation / LDS attribute failed"); delete h; return GAB_EDEVICE;
    }
    {   // scores that have a wavefront: M[s] exists iff M[s-x], M[s-o-e], I[s-e] or D[s-e] does; I[s] / D[s] iff M[s-o-e] or
        // I[s-e] / D[s-e] does (affine_wavefront_align.c:283-321) -- no data involved
        constexpr int kSteps = 1024;                      // >= the largest LDS directory
        const int x = h->pen.x, oe = h->pen.o + h->pen.e, e = h->pen.e;
        std::vector<uint8_t> m(kSteps + 1, 0), g(kSteps + 1, 0), step(kSteps, 1);
        m[0] = 1;
        for (int sc = 1; sc <= kSteps; sc++) {
            g[sc] = (sc >= oe && m[sc - oe]) || (sc >= e && g[sc - e]);
            m[sc] = (sc >= x && m[sc - x]) || g[sc];
        }
        int next = kSteps + 255;
        for (int sc = kSteps - 1; sc >= 0; sc--) {
            if (m[sc + 1]) next = sc + 1;
            step[sc] = (uint8_t)std::min(next - sc, 255);
        }
        if (h->steps.reserve(kSteps) != GAB_OK || hipMemcpy(h->steps.p, step.data(), kSteps, hipMemcpyHostToDevice) != hipSuccess) {
            gab_set_error("gab_wfa_create: score-step table upload failed"); gab_wfa_destroy(h); return GAB_EDEVICE;
        }
    }
    if (!h->adaptive) {
        // the same DP with ranges and pool positions: what wfa_pair would write into its directory, pair after pair
        // (affine_wavefront_align.c:85-106 compute_limits, :56-84 allocate_wavefronts, :283-321)
        constexpr int kMaxRows = 240, kMaxScore = 1024;
        struct Ent { bool m = false, g = false; int lo = 1, hi = -1, bm = 0, bi = 0, bd = 0, row = 0; };
        std::vector<Ent> ent(kMaxScore + 1);
        std::vector<WfRow> rows;
        const int x = h->pen.x, oe = h->pen.o + h->pen.e, e = h->pen.e;
        ent[0].m = true; ent[0].lo = ent[0].hi = 0; ent[0].bm = 0; ent[0].row = 0;
        WfRow r0; memset(&r0, 0, sizeof r0);
        r0.used_end = 1; r0.ms_lo = r0.mg_lo = r0.ie_lo = 1; r0.ms_hi = r0.mg_hi = r0.ie_hi = -1;
        rows.push_back(r0);
        int used = 1;
        const Ent none;
        for (int sc = 1; sc <= kMaxScore && (int)rows.size() < kMaxRows; sc++) {
            const Ent &ms = sc >= x ? ent[sc - x] : none, &mg = sc >= oe ? ent[sc - oe] : none, &ie = sc >= e ? ent[sc - e] : none;
            const bool n_ms = !ms.m, n_mg = !mg.m, n_ie = !ie.g;
            if (n_ms && n_mg && n_ie) continue;
            Ent &t = ent[sc];
            t.lo = std::min(std::min(n_ms ? 1 : ms.lo, n_mg ? 1 : mg.lo), n_ie ? 1 : ie.lo) - 1;
            t.hi = std::max(std::max(n_ms ? -1 : ms.hi, n_mg ? -1 : mg.hi), n_ie ? -1 : ie.hi) + 1;
            const int width = t.hi - t.lo + 1;
            t.m = true; t.g = !n_mg || !n_ie;
            t.bm = used; t.bi = used + width; t.bd = used + 2 * width;
            used += width * (t.g ? 3 : 1);
            if (used > 60000 || t.lo < -120 || t.hi > 120) break;
            t.row = (int)rows.size();
            WfRow r; memset(&r, 0, sizeof r);
            r.score = sc; r.lo = t.lo; r.hi = t.hi; r.bM = t.bm; r.bI = t.bi; r.bD = t.bd; r.used_end = used; r.has_gap = t.g;
            r.ms_m = n_ms ? 0 : ms.bm; r.ms_lo = n_ms ? 1 : ms.lo; r.ms_hi = n_ms ? -1 : ms.hi; r.r_ms = n_ms ? 0 : ms.row;
            r.mg_m = n_mg ? 0 : mg.bm; r.mg_lo = n_mg ? 1 : mg.lo; r.mg_hi = n_mg ? -1 : mg.hi; r.r_mg = n_mg ? 0 : mg.row;
            r.ie_i = n_ie ? 0 : ie.bi; r.ie_d = n_ie ? 0 : ie.bd; r.ie_lo = n_ie ? 1 : ie.lo; r.ie_hi = n_ie ? -1 : ie.hi; r.r_ie = n_ie ? 0 : ie.row;
            rows.push_back(r);
        }
        h->nrows = (int)rows.size();
        if (h->rows.reserve(sizeof(WfRow) * rows.size()) != GAB_OK ||
            hipMemcpy(h->rows.p, rows.data(), sizeof(WfRow) * rows.size(), hipMemcpyHostToDevice) != hipSuccess) {
            gab_set_error("gab_wfa_create: directory table upload failed"); gab_wfa_destroy(h); return GAB_EDEVICE;
        }
    }
    *out = h;
    return GAB_OK;
}

extern "C" void gab_wfa_destroy(gab_wfa *h) {
    if (!h) return;
    gab_device_guard g(h->device);
    h->ws.release(); h->steps.release(); h->rows.release(); h->scratch.release(); h->io.release(); h->hs.release();
    for (int k = 0; k < 4; k++) if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
    if (h->h_ct) (void)hipHostFree(h->h_ct);
    delete h;
}

extern "C" int gab_wfa_run_device(gab_wfa *h, const char *pat, int64_t pat_bytes, const int64_t *pat_off,
                                  const int32_t *pat_len, const char *txt, int64_t txt_bytes, const int64_t *txt_off,
                                  const int32_t *txt_len, int64_t n, char *ops_out, const int64_t *ops_off,
                                  int32_t *ops_len_out, int32_t *score_out, void *stream_) {
    GAB_CHECK(h, "gab_wfa_run_device: NULL handle");
    GAB_CHECK(n >= 0 && n < (1ll << 31), "gab_wfa_run_device: n=%lld out of range", (long long)n);
    h->have_stats = false;
    if (n == 0) return GAB_OK;
    GAB_CHECK(pat && pat_off && pat_len && txt && txt_off && txt_len && ops_out && ops_off && ops_len_out && score_out,
              "gab_wfa_run_device: NULL buffer");
    gab_device_guard g(h->device);
    gab_tuning_refresh(&h->tun);
    hipStream_t s = (hipStream_t)stream_;
    const size_t o_l0 = kCountersBytes + kSlotsBytes, o_l1 = o_l0 + 4 * (size_t)n, o_l2 = o_l1 + 4 * (size_t)n;
    int rc = h->ws.reserve(o_l2 + 4 * (size_t)n);
    if (rc) return rc;
    char *base = h->ws.as<char>();
    WfaCounters *d_ct = (WfaCounters *)base;
    uint32_t *l_a = (uint32_t *)(base + o_l0), *l_b = (uint32_t *)(base + o_l1), *l_big = (uint32_t *)(base + o_l2);
    WfaIO io{pat, pat_off, pat_len, txt, txt_off, txt_len, pat_bytes, txt_bytes, n, ops_out, ops_off, ops_len_out, score_out};

    GAB_HIP(hipEventRecord(h->ev[0], s));
    memset(h->h_ct, 0, sizeof(WfaCounters));
    h->h_ct->first_bad = 0x7fffffff;
    GAB_HIP(hipMemcpyAsync(d_ct, h->h_ct, sizeof(WfaCounters), hipMemcpyHostToDevice, s));
    GAB_HIP(hipMemsetAsync(base + kCountersBytes, 0, kSlotsBytes, s));
    const int grid = (int)std::min<int64_t>(gab_ceil_div(n, 256), 512);
    hipLaunchKernelGGL(wfa_classify, dim3(grid), dim3(256), 0, s, io, d_ct);
    GAB_HIP(hipMemcpyAsync(h->h_ct, d_ct, sizeof(WfaCounters), hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    if (h->h_ct->bad) {
        gab_set_error("gab_wfa_run_device: %d pair(s) violate the limits (first: pair %d): need 0 <= length <= %d and "
                      "offsets inside the slabs (readable to a multiple of 4 bytes)", h->h_ct->bad,
                      h->h_ct->first_bad - 1, GAB_WFA_MAX_LEN);
        return GAB_EINVAL;
    }
    const uint32_t n_lds = h->h_ct->n_lds, n_big = h->h_ct->n_big;
    const int seqp = ((h->h_ct->max_plen + kSeqPad + 8) + 15) & ~15, seqt = ((h->h_ct->max_tlen + kSeqPad + 8) + 15) & ~15;
    int64_t requeued = 0;

    GAB_HIP(hipEventRecord(h->ev[1], s));
    // LDS passes: (1) four pairs per wave with one-byte offsets -- complete mode: wfa_lds_static with a ~1.5 K and then a 3 K
    // history; adaptive mode: wfa_lds<16, OffB> with its directory in LDS; int16 offsets in a 2 KB pool when the strings are
    // too long for bytes -- (2) one pair per wave with 12 KB of int16 offsets, (3) one pair per wave with 96 KB; whatever
    // overflows goes to global memory
    if (n_big) hipLaunchKernelGGL(wfa_scatter, dim3(grid), dim3(256), 0, s, io, d_ct->cursors, l_a, l_big);
    uint32_t *cur = n_big ? l_a : nullptr, *nxt = l_b;      // nullptr: identity
    uint32_t cnt = n_lds;
    bool ev2 = false;
    int pool_bytes[3] = {2 * 1024, 12 * 1024, 96 * 1024};
    int dir_caps[3] = {48, 128, 640};
    int groups[3] = {16, 64, 64};
    int byte_tier = 1;
    const bool tuned = h->tun.wfa_tuned;                    // GAB_WFA_TUNE: tuning runs only
    if (tuned) {
        int *dst[7] = {&pool_bytes[0], &dir_caps[0], &pool_bytes[1], &dir_caps[1], &groups[0], &groups[1], &byte_tier};
        for (int k = 0; k < h->tun.wfa_tune_fields && k < 7; k++) *dst[k] = h->tun.wfa_tune[k];
    }
    // First tier with one-byte offsets (OffB) when no offset can leave the byte's range: text length + one per score step.
    // Its LDS is budgeted per pair: 2496 B = 10 032 B per wave of four pairs = 16 waves per CU, the measured optimum (the
    // next allocation step down, 14 waves, costs 9 %; 18-20 waves with a smaller history re-queue too many pairs).  In
    // complete mode the history a pair needs is a function of its score alone (lo / hi do not depend on the data), so the
    // directory is sized to the scores the pool can hold: 56 scores, ~1.9 K offsets.
    bool byte_ok = byte_tier != 0;
    if (byte_ok && !(tuned && byte_tier > 1)) {
        dir_caps[0] = h->adaptive ? 48 : 56;
        const int room = 2496 - dir_caps[0] * (h->adaptive ? 16 : 4) - (seqp + seqt);
        if (room < 1024) byte_ok = false;
        byte_tier = std::min(room & ~15, 2032);
    }
    if (byte_ok && h->h_ct->max_tlen + dir_caps[0] + 2 > kOffBMax) byte_ok = false;
    if (!byte_ok && !tuned) dir_caps[0] = 48;
    // complete mode: the first tier takes its directory from the penalties' table (wfa_pair_static) -- no directory in LDS
    const int static_rows = std::min(h->nrows, kOffBMax - 2 - h->h_ct->max_tlen);
    // LDS per pair of the first launch: 1568 B = 6272 B per wave of four = 24 waves per CU, six per SIMD (the kernel is
    // compiled for seven: 65 VGPRs, no scratch; six would do), i.e. ~1.2 K offsets behind two 151-bp strings = scores below 44 = 96 % of the
    // pairs.  Measured: 1 520 offsets at 20 waves 382 M/s, 1 344 at 23 waves 391, 1 088-1 184 at 24 waves 405; the history a
    // pair needs steps with its score, so 1 216-1 312 offsets buy nothing over 1 184 and cost a wave.
    const int static_pool = tuned && byte_tier > 1 ? byte_tier : std::min(1568 - (seqp + seqt), 4080) & ~15;
    const bool use_static = !h->adaptive && byte_tier != 0 && !h->tun.wfa_no_static && (groups[0] == 16 || groups[0] == 8) && static_rows >= 16 && static_pool >= 1024;
    // The tiers are launched back to back: tier k + 1 reads the number of pairs tier k left ON THE DEVICE and is sized by an
    // estimate (its waves stride over whatever there is), so the host looks at the counters once, after the last LDS tier
    // (three round trips of ~30 us less per call: 0.56 -> 0.46 ms per 100 k pairs, 2.65 -> 2.55 ms per 1 M).
    int tier = 0;                                            // LDS tiers launched so far
    const uint32_t cnt0 = cnt;
    auto next_grid = [&](double share, uint32_t per_wave, uint32_t floor_) {   // waves for a tier behind another one
        const uint32_t est = (uint32_t)((double)cnt0 * share) / per_wave + 1;
        return std::max<uint32_t>(std::min<uint32_t>(est, (cnt0 + per_wave - 1) / per_wave), std::min<uint32_t>(floor_, (cnt0 + per_wave - 1) / per_wave));
    };
    auto after_launch = [&]() {
        if (cur) std::swap(cur, nxt);
        else { cur = nxt; nxt = l_a; }                      // the identity pass: its overflow list becomes the input
        tier++;
    };
    const double shares[4] = {1.0, 0.10, 0.02, 0.005};       // expected share of the batch that reaches tier k (151-bp reads at 2 %: 4 %, 0.07 %, ~0)
    if (cnt && use_static) {
        // two launches: the small pool takes 96 % of the 151-bp pairs at 24 waves per CU, a 2.5 KB pool the scores up to ~64 of
        // the rest (measured: 2048-2560 B best, 4080 B 4 % slower)
        int static_pool2 = 2560;
        if (h->tun.wfa_pool2 >= 0) static_pool2 = h->tun.wfa_pool2;      // GAB_WFA_POOL2
        const int pools[2] = {static_pool, static_pool2};
        // the first launch leaves the wavefronts of the pairs it ran out of room for in slots of the scratch buffer and the second
        // continues them (the pool layout is the table's in both): it does not repeat the ~40 score steps they had come
        const bool two = pools[1] > pools[0];
        // (a multiple of the pairs per wave, so that the boundary between resumed and restarted pairs falls between waves;
        // the kernel copes with a mixed wave anyway.  GAB_WFA_SLOTS: tests force the boundary into a wave.)
        uint32_t slots = two ? (uint32_t)std::min<uint64_t>(cnt, std::max<uint64_t>(65536, cnt / 8)) & ~7u : 0;
        if (h->tun.wfa_slots >= 0 && two) slots = (uint32_t)std::min<uint64_t>(cnt, (uint64_t)h->tun.wfa_slots);      // GAB_WFA_SLOTS
        const size_t o_slots = ((size_t)sizeof(WfResume) * cnt + 255) & ~(size_t)255;
        WfResume *d_hdr = nullptr; uint8_t *d_slots = nullptr;
        if (two) {
            if ((rc = h->scratch.reserve(o_slots + (size_t)slots * pools[0])) != GAB_OK) return rc;
            d_hdr = (WfResume *)h->scratch.as<char>(); d_slots = (uint8_t *)h->scratch.as<char>() + o_slots;
        }
        for (int k = 0; k < 2; k++) {
            if (k == 1 && !two) break;
            const size_t per_group = (size_t)(seqp + seqt) + pools[k];
            const uint32_t per_wave = groups[0] == 8 ? 8 : 4;
            const uint32_t blocks = tier == 0 ? (cnt + per_wave - 1) / per_wave : next_grid(shares[std::min(tier, 3)], per_wave, 256);
            const uint32_t *cptr = tier == 0 ? nullptr : &d_ct->tier_over[tier - 1];
            auto kern = groups[0] == 8 ? (tier ? wfa_lds_static<8, true> : wfa_lds_static<8, false>) : (tier ? wfa_lds_static<16, true> : wfa_lds_static<16, false>);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), per_group * per_wave, s, io, h->pen, cur, cnt, cptr, seqp, seqt, pools[k],
                               (uint32_t)per_group, nxt, &d_ct->tier_over[tier], d_ct, h->rows.as<WfRow>(), static_rows,
                               k == 1 ? (const WfResume *)d_hdr : nullptr, k == 1 ? (const uint8_t *)d_slots : nullptr, slots, (uint32_t)pools[0],
                               k == 0 ? d_hdr : nullptr, k == 0 ? d_slots : nullptr, slots, (uint32_t)pools[0]);
            GAB_HIP(hipGetLastError());
            if (!ev2) { GAB_HIP(hipEventRecord(h->ev[2], s)); ev2 = true; }
            after_launch();
        }
    }
    for (int pass = use_static ? 1 : 0; pass < 3 && cnt; pass++) {
        const int dir_cap = dir_caps[pass], G = groups[pass];
        const bool bytes = pass == 0 && byte_ok;
        if (bytes) pool_bytes[0] = std::min(byte_tier, 2046);
        const size_t per_group = (((size_t)dir_cap * (h->adaptive ? 16 : bytes ? 4 : 12) + (size_t)(seqp + seqt) + pool_bytes[pass]) + 15) & ~(size_t)15;
        const size_t lds = per_group * (64 / G) + (((size_t)dir_cap + 15) & ~(size_t)15);
        if (lds > 160 * 1024 - 512) continue;            // sequences too long for this pool: let the next stage take them
        const uint32_t per_wave = 64 / G;
        const uint32_t blocks = tier == 0 ? (cnt + per_wave - 1) / per_wave : next_grid(shares[std::min(tier, 3)], per_wave, pass == 2 ? 64 : 256);
        const uint32_t *cptr = tier == 0 ? nullptr : &d_ct->tier_over[tier - 1];
        WfaLdsKernel kern = wfa_lds_kernel(G, h->adaptive, bytes);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, s, io, h->pen, cur, cnt, cptr, dir_cap, seqp, seqt,
                           pool_bytes[pass] / (bytes ? 1 : 2), (uint32_t)per_group, nxt, &d_ct->tier_over[tier], d_ct, h->steps.as<uint8_t>());
        GAB_HIP(hipGetLastError());
        if (!ev2) { GAB_HIP(hipEventRecord(h->ev[2], s)); ev2 = true; }
        after_launch();
    }
    // r04: what the last LDS tier leaves (almost never anything) goes to the global-history kernel WITHOUT asking the device how
    // many pairs that is: one launch of a few workgroups that reads the count itself, with the history room the handle already
    // has; the host looks at the counters once, at the end of the call, and only continues (a bigger pool) for pairs that
    // overflowed even that.  One round trip of ~40 us less per call.
    bool first_global_done = false;
    if (tier && cnt) {
        const int64_t pool_cap0 = 1 << 20, dir_cap0 = 4096;
        const int64_t per_block0 = dir_cap0 * (h->adaptive ? 7 : 5) + pool_cap0;
        int blocks0 = 16;
        if (const int64_t fit = (int64_t)(h->scratch.cap / ((size_t)per_block0 * 4)); fit >= 1) blocks0 = (int)std::min<int64_t>(64, fit);
        else if ((rc = h->scratch.reserve((size_t)per_block0 * 4 * blocks0)) != GAB_OK) return rc;
        GAB_HIP(hipMemsetAsync(&d_ct->n_over, 0, 4, s));
        hipLaunchKernelGGL(h->adaptive ? wfa_global<true> : wfa_global<false>, dim3(blocks0), dim3(64), 0, s, io, h->pen, cur, 0u, (const uint32_t *)&d_ct->tier_over[tier - 1],
                           h->scratch.as<int32_t>(), per_block0, (int)dir_cap0, (int)pool_cap0, nxt, d_ct);
        GAB_HIP(hipGetLastError());
        first_global_done = true;
    }
    if (tier) {
        hipLaunchKernelGGL(wfa_sum_work, dim3(1), dim3(kWorkSlots), 0, s, d_ct);
        GAB_HIP(hipMemcpyAsync(h->h_ct, d_ct, sizeof(WfaCounters), hipMemcpyDeviceToHost, s));
        if (!ev2) { GAB_HIP(hipEventRecord(h->ev[2], s)); ev2 = true; }
        GAB_HIP(hipStreamSynchronize(s));
        for (int k = 0; k < tier; k++) requeued += h->h_ct->tier_over[k];
        cnt = first_global_done ? h->h_ct->n_over : h->h_ct->tier_over[tier - 1];
        if (first_global_done) { requeued += cnt; std::swap(cur, nxt); }      // (what overflowed the first global pass sits in nxt)
    }
    if (!ev2) GAB_HIP(hipEventRecord(h->ev[2], s));
    // pass 3+: global history; first the LDS leftovers, then the long pairs; pool grows on overflow
    for (int which = 0; which < 2; which++) {
        uint32_t *list = which == 0 ? cur : l_big;
        uint32_t *spill = which == 0 ? nxt : (cur == l_a ? l_b : l_a);
        uint32_t c = which == 0 ? cnt : n_big;
        int64_t pool_cap = 1 << 20;                       // int32 elements per block
        int64_t dir_cap = 4096;
        if (which == 0 && first_global_done) { pool_cap *= 8; dir_cap *= 4; }      // (the first size has been tried)
        while (c) {
            int64_t per_block = dir_cap * (h->adaptive ? 7 : 5) + pool_cap;
            int blocks = (int)std::min<int64_t>(c, std::max<int64_t>(1, (int64_t)(h->scratch_budget / 4) / per_block));
            blocks = std::min(blocks, 2048);
            // what is already there (gab_wfa_reserve) is used as it is when it holds 64 workgroups' histories or more: an
            // allocation here stalls every stream of the device for milliseconds (the workgroups stride over the pairs anyway)
            if (const int64_t fit = (int64_t)(h->scratch.cap / ((size_t)per_block * 4)); fit >= 64) blocks = (int)std::min<int64_t>(blocks, fit);
            if ((size_t)per_block * 4 > h->scratch_budget) {
                gab_set_error("gab_wfa_run_device: a pair needs more than %zu bytes of wavefront history", h->scratch_budget);
                return GAB_ENOMEM;
            }
            rc = h->scratch.reserve((size_t)per_block * 4 * blocks);
            if (rc) return rc;
            GAB_HIP(hipMemsetAsync(&d_ct->n_over, 0, 4, s));
            hipLaunchKernelGGL(h->adaptive ? wfa_global<true> : wfa_global<false>, dim3(blocks), dim3(64), 0, s, io, h->pen, list, c, (const uint32_t *)nullptr, h->scratch.as<int32_t>(),
                               per_block, (int)dir_cap, (int)pool_cap, spill, d_ct);
            GAB_HIP(hipGetLastError());
            hipLaunchKernelGGL(wfa_sum_work, dim3(1), dim3(kWorkSlots), 0, s, d_ct);
            GAB_HIP(hipMemcpyAsync(h->h_ct, d_ct, sizeof(WfaCounters), hipMemcpyDeviceToHost, s));
            GAB_HIP(hipStreamSynchronize(s));
            c = h->h_ct->n_over;
            requeued += c;
            std::swap(list, spill);
            pool_cap *= 8; dir_cap *= 4;
            if (dir_cap > (1 << 22)) dir_cap = 1 << 22;
        }
    }
    GAB_HIP(hipEventRecord(h->ev[3], s));
    if (!tier || n_big) GAB_HIP(hipStreamSynchronize(s));      // (with LDS tiers and no long pairs the counters have been read after the last kernel)
    h->last_requeued = requeued;
    h->have_stats = true;
    return GAB_OK;
}

extern "C" int gab_wfa_run(gab_wfa *h, const char *pat, const int64_t *pat_off, const int32_t *pat_len, const char *txt,
                           const int64_t *txt_off, const int32_t *txt_len, int64_t n, char *ops_out,
                           const int64_t *ops_off, int32_t *ops_len_out, int32_t *score_out) {
    GAB_CHECK(h, "gab_wfa_run: NULL handle");
    GAB_CHECK(n >= 0 && n < (1ll << 31), "gab_wfa_run: n=%lld out of range", (long long)n);
    if (n == 0) return GAB_OK;
    GAB_CHECK(pat && pat_off && pat_len && txt && txt_off && txt_len && ops_out && ops_off && ops_len_out && score_out,
              "gab_wfa_run: NULL buffer");
    gab_device_guard g(h->device);
    int64_t pb = 0, tb = 0, ob = 0, pa = INT64_MAX, ta = INT64_MAX, oa = INT64_MAX;
    for (int64_t i = 0; i < n; i++) {
        GAB_CHECK(pat_off[i] >= 0 && txt_off[i] >= 0 && ops_off[i] >= 0 && pat_len[i] >= 0 && txt_len[i] >= 0,
                  "gab_wfa_run: negative offset/length at pair %lld", (long long)i);
        pb = std::max(pb, pat_off[i] + pat_len[i]); tb = std::max(tb, txt_off[i] + txt_len[i]);
        ob = std::max(ob, ops_off[i] + pat_len[i] + txt_len[i]);
        pa = std::min(pa, pat_off[i]); ta = std::min(ta, txt_off[i]); oa = std::min(oa, ops_off[i]);
    }
    pa &= ~(int64_t)255; ta &= ~(int64_t)255;   // stage only the referenced windows (oa stays exact: it is written back)
    // one slab for both with overlapping windows (the drivers' pair files: '>' and '<' lines interleaved): staged once, not twice
    const bool shared = pat == txt && std::max(pb, tb) - std::min(pa, ta) <= (pb - pa) + (tb - ta);
    if (shared) { pa = ta = std::min(pa, ta); pb = tb = std::max(pb, tb); }
    const size_t ppad = ((size_t)(pb - pa) + 3 + 255) & ~(size_t)255, tpad = shared ? 0 : ((size_t)(tb - ta) + 3 + 255) & ~(size_t)255;
    const size_t opad = ((size_t)(ob - oa) + 255) & ~(size_t)255, nn = (size_t)n;
    size_t o = 0;
    const size_t o_p = o; o += ppad;
    const size_t o_t = o; o += tpad;
    const size_t o_ops = o; o += opad;
    const size_t o_po = o; o += 8 * nn;
    const size_t o_to = o; o += 8 * nn;
    const size_t o_oo = o; o += 8 * nn;
    const size_t o_pl = o; o += 4 * nn;
    const size_t o_tl = o; o += 4 * nn;
    const size_t o_ol = o; o += 4 * nn;
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
        GAB_HIP(hipMemcpyAsync(b + o_oo, ops_off, 8 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_pl, pat_len, 4 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_tl, txt_len, 4 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipStreamSynchronize(s));
    }
    rc = gab_wfa_run_device(h, b + o_p - pa, pa + (int64_t)ppad, (const int64_t *)(b + o_po), (const int32_t *)(b + o_pl),
                            (shared ? b + o_p : b + o_t) - ta, ta + (int64_t)(shared ? ppad : tpad), (const int64_t *)(b + o_to), (const int32_t *)(b + o_tl), n, b + o_ops - oa,
                            (const int64_t *)(b + o_oo), (int32_t *)(b + o_ol), (int32_t *)(b + o_sc), s);
    if (rc) return rc;
    // only each pair's own operations are defined; copy the window back and let the caller read ops_len[i] bytes per pair
    GAB_HIP(hipMemcpyAsync(ops_out + oa, b + o_ops, (size_t)(ob - oa), hipMemcpyDeviceToHost, s));
    GAB_HIP(hipMemcpyAsync(ops_len_out, b + o_ol, 4 * nn, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipMemcpyAsync(score_out, b + o_sc, 4 * nn, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    return GAB_OK;
}

// The host-pointer entry point of a driver that PRINTS the alignments: same inputs as gab_wfa_run, the result as the
// run-length text of edit_cigar_print.  The operations stay on the device (fixed-stride room in the handle's staging buffer).
extern "C" int gab_wfa_run_packed(gab_wfa *h, const char *pat, const int64_t *pat_off, const int32_t *pat_len, const char *txt,
                                  const int64_t *txt_off, const int32_t *txt_len, int64_t n, char *cigar_out, int64_t capacity,
                                  int64_t *cigar_off_out, int32_t *cigar_len_out, int32_t *score_out, int64_t *cigar_bytes) {
    GAB_CHECK(h, "gab_wfa_run_packed: NULL handle");
    GAB_CHECK(n >= 0 && n < (1ll << 31), "gab_wfa_run_packed: n=%lld out of range", (long long)n);
    if (cigar_bytes) *cigar_bytes = 0;
    if (n == 0) return GAB_OK;
    GAB_CHECK(pat && pat_off && pat_len && txt && txt_off && txt_len && cigar_off_out && cigar_len_out && score_out && capacity >= 0 &&
              (cigar_out || capacity == 0), "gab_wfa_run_packed: NULL buffer");
    gab_device_guard g(h->device);
    gab_tuning_refresh(&h->tun);
    const bool trace = h->tun.wfa_trace;      // GAB_WFA_TRACE, diagnosis: per-phase wall times of this call on stderr
    auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t_0 = now();
    int64_t pb = 0, tb = 0, pa = INT64_MAX, ta = INT64_MAX, stride = 0, room = 0;
    for (int64_t i = 0; i < n; i++) {
        GAB_CHECK(pat_off[i] >= 0 && txt_off[i] >= 0 && pat_len[i] >= 0 && txt_len[i] >= 0,
                  "gab_wfa_run_packed: negative offset/length at pair %lld", (long long)i);
        pb = std::max(pb, pat_off[i] + pat_len[i]); tb = std::max(tb, txt_off[i] + txt_len[i]);
        pa = std::min(pa, pat_off[i]); ta = std::min(ta, txt_off[i]);
        stride = std::max<int64_t>(stride, (int64_t)pat_len[i] + txt_len[i]);
        room += (((int64_t)pat_len[i] + txt_len[i]) + 7) & ~(int64_t)7;
    }
    stride = (stride + 7) & ~(int64_t)7;
    // operation room on the device: a fixed stride per pair (no offset array to build or copy) unless one long pair among
    // short ones would make that more than twice the sum of the pairs' own rooms -- then exact offsets, built here
    const bool fixed = stride * n <= 2 * room + (1 << 20);
    std::vector<int64_t> exact;
    if (!fixed) {
        exact.resize((size_t)n);
        int64_t at = 0;
        for (int64_t i = 0; i < n; i++) { exact[(size_t)i] = at; at += (((int64_t)pat_len[i] + txt_len[i]) + 7) & ~(int64_t)7; }
    }
    pa &= ~(int64_t)255; ta &= ~(int64_t)255;
    const bool shared = pat == txt && std::max(pb, tb) - std::min(pa, ta) <= (pb - pa) + (tb - ta);
    if (shared) { pa = ta = std::min(pa, ta); pb = tb = std::max(pb, tb); }
    const size_t ppad = ((size_t)(pb - pa) + 3 + 255) & ~(size_t)255, tpad = shared ? 0 : ((size_t)(tb - ta) + 3 + 255) & ~(size_t)255;
    const size_t nn = (size_t)n, opad = ((size_t)(fixed ? stride * n : room) + 16 + 255) & ~(size_t)255, cpad = ((size_t)capacity + 255) & ~(size_t)255;
    size_t o = 0;
    const size_t o_p = o; o += ppad;
    const size_t o_t = o; o += tpad;
    const size_t o_ops = o; o += opad;
    const size_t o_txt = o; o += cpad;
    const size_t o_po = o; o += 8 * nn;
    const size_t o_to = o; o += 8 * nn;
    const size_t o_oo = o; o += 8 * nn;
    const size_t o_co = o; o += 8 * nn;
    const size_t o_pl = o; o += 4 * nn;
    const size_t o_tl = o; o += 4 * nn;
    const size_t o_ol = o; o += 4 * nn;
    const size_t o_cl = o; o += 4 * nn;
    const size_t o_sc = o; o += 4 * nn;
    o = (o + 255) & ~(size_t)255;                  // (a 64-bit atomic lives here)
    const size_t o_cur = o; o += 256;
    GAB_CHECK_ATOMIC64(o_cur);                       // wfa_rle_pack's 64-bit cursor
    int rc = h->io.reserve(o);
    if (rc) return rc;
    char *b = h->io.as<char>();
    hipStream_t s = nullptr;
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    const double t_1 = now();
    double t_gate = 0;
    {   // the copies of one chunk at a time per GPU (gab_core.hip: the workers of a GPU must not copy in lockstep)
        std::lock_guard<std::mutex> gate(gab_h2d_mutex(h->device));
        t_gate = now();
        GAB_HIP(hipMemcpyAsync(b + o_p, pat + pa, (size_t)(pb - pa), hipMemcpyHostToDevice, s));
        if (!shared) GAB_HIP(hipMemcpyAsync(b + o_t, txt + ta, (size_t)(tb - ta), hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_po, pat_off, 8 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_to, txt_off, 8 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_pl, pat_len, 4 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_tl, txt_len, 4 * nn, hipMemcpyHostToDevice, s));
        if (!fixed) GAB_HIP(hipMemcpyAsync(b + o_oo, exact.data(), 8 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipStreamSynchronize(s));
    }
    const double t_2 = now();
    const unsigned grid = (unsigned)((nn + 255) / 256);
    if (fixed) hipLaunchKernelGGL(wfa_fill_stride, dim3(grid), dim3(256), 0, s, (int64_t *)(b + o_oo), (uint32_t)n, stride);
    GAB_HIP(hipGetLastError());
    rc = gab_wfa_run_device(h, b + o_p - pa, pa + (int64_t)ppad, (const int64_t *)(b + o_po), (const int32_t *)(b + o_pl),
                            (shared ? b + o_p : b + o_t) - ta, ta + (int64_t)(shared ? ppad : tpad), (const int64_t *)(b + o_to), (const int32_t *)(b + o_tl), n, b + o_ops,
                            (const int64_t *)(b + o_oo), (int32_t *)(b + o_ol), (int32_t *)(b + o_sc), s);
    if (rc) return rc;
    GAB_HIP(hipMemsetAsync(b + o_cur, 0, 8, s));
    hipLaunchKernelGGL(wfa_rle_pack, dim3(grid), dim3(256), 0, s, (const char *)(b + o_ops), (const int64_t *)(b + o_oo), (const int32_t *)(b + o_ol),
                       (uint32_t)n, b + o_txt, (unsigned long long)capacity, (unsigned long long *)(b + o_cur), (int64_t *)(b + o_co), (int32_t *)(b + o_cl));
    GAB_HIP(hipGetLastError());
    unsigned long long *h_cur = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(h->h_ct) + 128);
    GAB_HIP(hipMemcpyAsync(h_cur, b + o_cur, 8, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipMemcpyAsync(cigar_off_out, b + o_co, 8 * nn, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipMemcpyAsync(cigar_len_out, b + o_cl, 4 * nn, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipMemcpyAsync(score_out, b + o_sc, 4 * nn, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    const double t_3 = now();
    const int64_t total = (int64_t)*h_cur;
    if (cigar_bytes) *cigar_bytes = total;
    if (total > capacity) {
        gab_set_error("gab_wfa_run_packed: %lld bytes of CIGAR text do not fit the caller's %lld", (long long)total, (long long)capacity);
        return GAB_ERANGE;
    }
    if (total) {
        GAB_HIP(hipMemcpyAsync(cigar_out, b + o_txt, (size_t)total, hipMemcpyDeviceToHost, s));
        GAB_HIP(hipStreamSynchronize(s));
    }
    if (trace)
        fprintf(stderr, "[gab_wfa_run_packed %p] %lld pairs: host scan + buffers %.2f ms, wait for the copy gate %.2f ms, H2D of %.1f MB %.2f ms, "
                        "kernels + small D2H %.2f ms, D2H of %.1f MB of text %.2f ms (t0 = %.2f)\n", (void *)h, (long long)n, t_1 - t_0, t_gate - t_1,
                (double)((pb - pa) + 24 * n) / 1e6, t_2 - t_gate, t_3 - t_2, (double)total / 1e6, now() - t_3, t_0);
    return GAB_OK;
}

// gab_wfa_run_packed for pairs that are ALREADY on the device (a driver whose read phase parsed the file there, SURVEY.md 8f row
// f1): inputs as gab_wfa_run_device takes them, `ops` / `ops_off` = operation room on the device (scratch of the caller), the
// result as gab_wfa_run_packed returns it -- the printed text and its index in HOST memory.  Nothing but that text, the offsets,
// lengths and scores crosses the bus.
extern "C" int gab_wfa_run_packed_device(gab_wfa *h, const char *pat, int64_t pat_bytes, const int64_t *pat_off, const int32_t *pat_len,
                                         const char *txt, int64_t txt_bytes, const int64_t *txt_off, const int32_t *txt_len, int64_t n,
                                         char *ops, const int64_t *ops_off, char *cigar_out, int64_t capacity, int64_t *cigar_off_out,
                                         int32_t *cigar_len_out, int32_t *score_out, int64_t *cigar_bytes) {
    GAB_CHECK(h, "gab_wfa_run_packed_device: NULL handle");
    GAB_CHECK(n >= 0 && n < (1ll << 31), "gab_wfa_run_packed_device: n=%lld out of range", (long long)n);
    if (cigar_bytes) *cigar_bytes = 0;
    if (n == 0) return GAB_OK;
    GAB_CHECK(pat && pat_off && pat_len && txt && txt_off && txt_len && ops && ops_off && cigar_off_out && cigar_len_out && score_out &&
              capacity >= 0 && (cigar_out || capacity == 0), "gab_wfa_run_packed_device: NULL buffer");
    gab_device_guard g(h->device);
    const size_t nn = (size_t)n, cpad = ((size_t)capacity + 255) & ~(size_t)255;
    size_t o = 0;
    const size_t o_txt = o; o += cpad;
    const size_t o_co = o; o += 8 * nn;
    const size_t o_ol = o; o += 4 * nn;
    const size_t o_cl = o; o += 4 * nn;
    const size_t o_sc = o; o += 4 * nn;
    o = (o + 255) & ~(size_t)255;                  // (a 64-bit atomic lives here)
    const size_t o_cur = o; o += 256;
    GAB_CHECK_ATOMIC64(o_cur);
    int rc = h->io.reserve(o);
    if (rc) return rc;
    char *b = h->io.as<char>();
    hipStream_t s = nullptr;
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    rc = gab_wfa_run_device(h, pat, pat_bytes, pat_off, pat_len, txt, txt_bytes, txt_off, txt_len, n, ops, ops_off, (int32_t *)(b + o_ol),
                            (int32_t *)(b + o_sc), s);
    if (rc) return rc;
    const unsigned grid = (unsigned)((nn + 255) / 256);
    GAB_HIP(hipMemsetAsync(b + o_cur, 0, 8, s));
    hipLaunchKernelGGL(wfa_rle_pack, dim3(grid), dim3(256), 0, s, (const char *)ops, ops_off, (const int32_t *)(b + o_ol), (uint32_t)n, b + o_txt,
                       (unsigned long long)capacity, (unsigned long long *)(b + o_cur), (int64_t *)(b + o_co), (int32_t *)(b + o_cl));
    GAB_HIP(hipGetLastError());
    unsigned long long *h_cur = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(h->h_ct) + 128);
    GAB_HIP(hipMemcpyAsync(h_cur, b + o_cur, 8, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipMemcpyAsync(cigar_off_out, b + o_co, 8 * nn, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipMemcpyAsync(cigar_len_out, b + o_cl, 4 * nn, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipMemcpyAsync(score_out, b + o_sc, 4 * nn, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    const int64_t total = (int64_t)*h_cur;
    if (cigar_bytes) *cigar_bytes = total;
    if (total > capacity) {
        gab_set_error("gab_wfa_run_packed_device: %lld bytes of CIGAR text do not fit the caller's %lld", (long long)total, (long long)capacity);
        return GAB_ERANGE;
    }
    if (total) {
        GAB_HIP(hipMemcpyAsync(cigar_out, b + o_txt, (size_t)total, hipMemcpyDeviceToHost, s));
        GAB_HIP(hipStreamSynchronize(s));
    }
    return GAB_OK;
}

// see gab_bpm_reserve; max_ops_bytes = the room of the CIGAR operations (pattern + text length per pair)
extern "C" int gab_wfa_reserve(gab_wfa *h, int64_t max_pairs, int64_t max_seq_bytes, int64_t max_ops_bytes) {
    GAB_CHECK(h, "gab_wfa_reserve: NULL handle");
    GAB_CHECK(max_pairs >= 0 && max_pairs < (1ll << 31) && max_seq_bytes >= 0 && max_ops_bytes >= 0, "gab_wfa_reserve: size out of range");
    gab_device_guard g(h->device);
    const size_t nn = (size_t)max_pairs;
    int rc = h->io.reserve(std::max<size_t>(2 * (((size_t)max_seq_bytes + 3 + 511) & ~(size_t)255) + (((size_t)max_ops_bytes + 511) & ~(size_t)255) + 60 * nn + 2048,
                                            (size_t)4 << 20));
    if (rc) return rc;
    if ((rc = h->ws.reserve(kCountersBytes + kSlotsBytes + 3 * 4 * nn + 1024)) != GAB_OK) return rc;
    {   // the hand-over slots of the two static tiers (gab_wfa_run_device: headers of all pairs + a pool per slot): 80 MB for a
        // chunk of 2^18 pairs, which the first call of a timed region must not have to allocate and map
        const size_t slots = std::min<size_t>(nn, std::max<size_t>(65536, nn / 8));
        // ... nor the first round of the global-history tier (4.3 MB per workgroup; the few pairs per chunk that outgrow the LDS
        // tiers): 64 .. 256 workgroups' worth
        const size_t global_tier = (size_t)(4096 * (h->adaptive ? 7 : 5) + (1 << 20)) * 4 * std::min<size_t>(256, std::max<size_t>(64, nn / 1024));
        if ((rc = h->scratch.reserve(std::max(((sizeof(WfResume) * nn + 255) & ~(size_t)255) + slots * 1568, std::min(global_tier, h->scratch_budget)))) != GAB_OK) return rc;
    }
    hipStream_t s = nullptr;
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    GAB_HIP(hipMemsetAsync(h->io.p, 0, h->io.cap, s));
    GAB_HIP(hipMemsetAsync(h->ws.p, 0, h->ws.cap, s));
    GAB_HIP(hipMemsetAsync(h->scratch.p, 0, h->scratch.cap, s));
    GAB_HIP(hipStreamSynchronize(s));
    if ((rc = gab_warm_copy_engines(s, h->io.p, h->io.cap)) != GAB_OK) return rc;
    // ... and one tiny batch through the whole path: the first launch of a kernel pays for loading the code object and for
    // the runtime's per-kernel bookkeeping (milliseconds, once per process and handle) -- not inside the caller's ROI
    static const char seq[] = "ACGTTGCAACGTACGTTGCATGCAACGTACGT" "ACGTTGCAACCTACGTTGCATGAACGTACGTA";
    const int64_t po[4] = {0, 0, 32, 32}, to[4] = {32, 0, 0, 32};
    const int32_t pl[4] = {32, 32, 33, 5}, tl[4] = {33, 32, 32, 7};
    char text[256]; int64_t coff[4], bytes = 0; int32_t clen[4], sc[4];
    const bool had = h->have_stats;
    rc = gab_wfa_run_packed(h, seq, po, pl, seq, to, tl, 4, text, (int64_t)sizeof text, coff, clen, sc, &bytes);
    h->have_stats = had;
    return rc;
}

extern "C" int gab_wfa_last_stats(gab_wfa *h, int64_t *work, int64_t *requeued, float *first_pass_ms, float *total_ms) {
    GAB_CHECK(h, "gab_wfa_last_stats: NULL handle");
    GAB_CHECK(h->have_stats, "gab_wfa_last_stats: no completed run on this handle");
    gab_device_guard g(h->device);
    GAB_HIP(hipEventSynchronize(h->ev[3]));
    if (work) *work = (int64_t)h->h_ct->work;
    if (requeued) *requeued = h->last_requeued;
    if (first_pass_ms) GAB_HIP(hipEventElapsedTime(first_pass_ms, h->ev[1], h->ev[2]));
    if (total_ms) GAB_HIP(hipEventElapsedTime(total_ms, h->ev[0], h->ev[3]));
    return GAB_OK;
}
