// bsw -- banded Smith-Waterman seed extension on gfx950.
//
// Semantics: BandedPairWiseSW::scalarBandedSWA
//   (/root/reference/benchmarks/bsw/src/bandedSWA.cpp:132-253), which is what the
//   reference's inter-sequence SIMD path (getScores16, bandedSWA.cpp:1128-1835) emulates
//   lane by lane.  The CPU vectorises ACROSS pairs (one pair per 16-bit SIMD lane) because
//   the adaptive band, the z-drop exit and the first/last-max tie rules make the rows of
//   one pair strictly sequential.  The MI355X equivalent of that idea is one pair per
//   wavefront lane:
//
//   1. bucket  -- counting sort of the pairs by (query length, reference length / 8) so
//                 that the 64 lanes of a wave carry near-identical work and diverge little
//                 (the reference sorts by len1 for the same reason, bandedSWA.cpp:372-407);
//   2. bsw_dp  -- one wave (= one workgroup) per 64 sorted pairs.  The H/E row of every
//                 lane lives in LDS as [column][lane] dwords (bank = lane, conflict-free
//                 for any per-lane column), H and E packed as two u16 in one dword when
//                 max(h0) + 256*max(mat) fits 15 bits (the reference's own int16 lanes),
//                 else two dwords.  LDS is sized per launch from the query-length class, so
//                 short queries get up to 8 waves/CU and 256-base queries still fit.
//   Results are scattered back by pair id, so the output order is the input order.
//
// Roofline: integer-VALU / LDS bound (~20 VALU + 1 LDS read + 1 LDS write per DP cell,
// ~7.4 k cells per ~210 input bytes); HBM traffic is the algorithmic minimum
// len1 + len2 + 12 B per pair plus the 4-byte permutation entry.
#include "gab_internal.h"
#include <atomic>
#include <algorithm>
#include <new>
#include <string.h>
#include <time.h>
#include <stdlib.h>

namespace {

constexpr int kQBuckets = 256;            // query length 1..256 -> 0..255
constexpr int kTBuckets = 256;            // min(tlen >> 3, 255)
constexpr int kNumKeys = kQBuckets * kTBuckets;
constexpr int kClassStep = 16;            // query-length classes for LDS sizing
constexpr int kNumClasses = kQBuckets / kClassStep;

struct BswConst {
    int32_t o_del, e_del, o_ins, e_ins, zdrop, end_bonus, w, max_sc;
    uint32_t row_lo[5];   // biased (+128) scores mat[t][0..3], one byte each
    uint32_t row_hi[5];   // biased score mat[t][4] in byte 0
};

struct BswStats {          // device-side, zeroed per run
    unsigned long long cells;
    int32_t max_h0;
    int32_t bad;           // number of pairs that failed validation
    int32_t first_bad;     // smallest failing index + 1
    int32_t pad;
};
GAB_STATIC_ATOMIC64(BswStats, cells);

struct BswIO {
    const uint8_t *ref; const int64_t *ref_off;
    const uint8_t *qry; const int64_t *qry_off;
    const int32_t *len1, *len2, *h0;
    int64_t ref_bytes, qry_bytes, n;
    int64_t ref_lo, qry_lo;        // lowest readable offset of the two slabs (0 for a caller's device slabs; the staged window's
                                   // start when gab_bsw_run has copied only the part of the host slabs it expects the pairs to use)
    int64_t ref_hi, qry_hi;        // end of the bytes that really hold the caller's data (= ref_bytes / qry_bytes for device slabs; the
};                                 // end of what gab_bsw_run COPIED: its window is padded to 256 bytes, and the padding is not the caller's data)

// (query length, reference length / 8, h0 / 32): lanes of a wave then run the same number of rows AND start with bands of
// similar width (row -1 is non-zero up to column ~h0, and the band stays ~2 x score wide until it reaches w)
__device__ __forceinline__ int bsw_key(int qlen, int tlen, int h0) {
    int tb = tlen >> 3; if (tb > kTBuckets / 4 - 1) tb = kTBuckets / 4 - 1;
    int hc = h0 >> 5; if (hc > 3) hc = 3;
    return (qlen - 1) * kTBuckets + tb * 4 + hc;
}

// A 151-bp read set puts almost all pairs into ~256 NEIGHBOURING keys (one query length), i.e. 1 KB of counters behind a
// handful of memory channels, and the histogram pass is bound by the atomic rate of those channels.  The counters are
// therefore stored at a multiplicatively scrambled index (a bijection on 16 bits): neighbouring keys lie 16 KB apart.
__device__ __forceinline__ int bsw_hslot(int key) { return (int)(((uint32_t)key * 4099u) & (uint32_t)(kNumKeys - 1)); }
static_assert((kNumKeys & (kNumKeys - 1)) == 0, "bsw_hslot needs a power-of-two key space");

// ---- pass 1: validate + histogram ------------------------------------------------------
// The value the histogram atomic returns is the pair's rank inside its bucket: it is kept, so that the scatter pass
// needs no second round of 10 M atomics (each pass was atomic-throughput bound at ~13 G/s).
__global__ __launch_bounds__(256) void bsw_hist(BswIO io, uint32_t *hist, uint32_t *rank, BswStats *st) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int mh = 0;
    for (; i < io.n; i += stride) {
        int ql = io.len2[i], tl = io.len1[i], h = io.h0[i];
        int64_t ro = io.ref_off[i], qo = io.qry_off[i];
        bool ok = ql >= 1 && ql <= GAB_BSW_MAX_QLEN && tl >= 1 && tl <= GAB_BSW_MAX_TLEN && h >= 0 &&
                  h <= (1 << 29) && ro >= io.ref_lo && qo >= io.qry_lo &&
                  ro + tl + 3 <= io.ref_bytes && qo + ql + 3 <= io.qry_bytes &&   // (the kernels read dwords from the sequence's own start)
                  ro + tl <= io.ref_hi && qo + ql <= io.qry_hi;
        if (!ok) {
            atomicAdd(&st->bad, 1);
            atomicMin((unsigned int *)&st->first_bad, (unsigned int)(i + 1 > 0x7fffffff ? 0x7fffffff : i + 1));
            rank[i] = ~0u;                 // (the scatter pass is already queued behind this one: it must not place this pair)
            continue;
        }
        mh = h > mh ? h : mh;
        rank[i] = atomicAdd(&hist[bsw_hslot(bsw_key(ql, tl, h))], 1u);
    }
    // wave max of h0, one atomic per wave
    for (int o = 32; o > 0; o >>= 1) { int v = __shfl_xor(mh, o); mh = v > mh ? v : mh; }
    if ((threadIdx.x & 63) == 0 && mh > 0) atomicMax(&st->max_h0, mh);
}

// ---- pass 2: exclusive scan of the 65536 bins ----------------------------------------------
// Two launches of 64 workgroups: (a) every thread takes ONE bin -- the counters sit at scrambled slots, so a thread that
// walks 64 consecutive bins pays 128 scattered loads one after the other (a single workgroup doing that took 113 us,
// 9 % of a 100 000-pair batch) -- and the workgroup scans its 1024 bins; (b) adds the totals of the workgroups in front.
constexpr int kScanBlocks = kNumKeys / 1024;
static_assert(kScanBlocks == 64, "bsw_scan_b reduces the block totals with one wave");
__global__ __launch_bounds__(1024) void bsw_scan_a(const uint32_t *__restrict__ hist, uint32_t *__restrict__ start, uint32_t *__restrict__ sums) {
    __shared__ uint32_t wsum[16];
    const int t = threadIdx.x, key = blockIdx.x * 1024 + t, lane = t & 63, wv = t >> 6;
    const uint32_t v = hist[bsw_hslot(key)];
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int k = 0; k < wv; k++) before += wsum[k];
    start[key] = before + inc - v;
    if (t == 1023) sums[blockIdx.x] = before + inc;
}
__global__ __launch_bounds__(1024) void bsw_scan_b(uint32_t *__restrict__ start, const uint32_t *__restrict__ sums, uint32_t *__restrict__ qstart) {
    __shared__ uint32_t s_off, s_total;
    const int t = threadIdx.x, key = blockIdx.x * 1024 + t;
    if (t < 64) {
        const uint32_t sv = sums[t];
        uint32_t mine = t < (int)blockIdx.x ? sv : 0, all = sv;
        for (int o = 32; o > 0; o >>= 1) { mine += __shfl_xor(mine, o); all += __shfl_xor(all, o); }
        if (t == 0) { s_off = mine; s_total = all; }
    }
    __syncthreads();
    const uint32_t run = start[key] + s_off;
    start[key] = run;
    if ((key % kTBuckets) == 0) qstart[key / kTBuckets] = run;
    if (key == kNumKeys - 1) { start[kNumKeys] = s_total; qstart[kQBuckets] = s_total; }
}

// ---- pass 3: scatter the pairs' descriptors into bucket order -------------------------------
// One 32-byte record per pair, written here from coalesced reads of the five input arrays and read coalesced by the DP
// kernels: the DP kernels used to gather the five fields through the permutation (five random 64-byte sectors per pair,
// 3.2 GB of the 7.6 GB the DP fetched per 10 M pairs against 2.1 GB of algorithmic bytes).
struct __attribute__((aligned(32))) BswRec { int64_t ref_off, qry_off; int32_t len1, len2, h0; uint32_t id; };

__global__ __launch_bounds__(256) void bsw_scatter(BswIO io, const uint32_t *__restrict__ start, const uint32_t *__restrict__ rank,
                                                   BswRec *recs) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < io.n; i += stride) {
        int ql = io.len2[i], tl = io.len1[i];
        if (rank[i] == ~0u) continue;      // failed bsw_hist's validation (the host returns GAB_EINVAL after this pass)
        BswRec r;
        r.ref_off = io.ref_off[i]; r.qry_off = io.qry_off[i]; r.len1 = tl; r.len2 = ql; r.h0 = io.h0[i]; r.id = (uint32_t)i;
        recs[start[bsw_key(ql, tl, r.h0)] + rank[i]] = r;
    }
}

// ---- pass 4: the DP ------------------------------------------------------------------------
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p) {
    uint32_t w;
    __builtin_memcpy(&w, p, 4);
    return w;
}

// WIDE = false: H and E packed as (E << 16) | H in one dword per column (both < 2^15).
// WIDE = true : H at lds[j*64+lane], E at lds[(qcap+1+j)*64+lane].
template <bool WIDE>
__global__ __launch_bounds__(64) void bsw_dp(BswIO io, BswConst c, const BswRec *__restrict__ recs,
                                            int64_t kbeg, int64_t kend, int qcap,
                                            int32_t *__restrict__ score_out,
                                            gab_bsw_result *__restrict__ result_out, BswStats *st) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    const int64_t k = kbeg + (int64_t)(gridDim.x - 1 - blockIdx.x) * 64 + lane;   // heaviest waves (largest key) first
    const bool valid = k < kend;
    BswRec rec; rec.ref_off = rec.qry_off = 0; rec.len1 = rec.len2 = rec.h0 = 0; rec.id = 0u;
    if (valid) rec = recs[k];
    const uint32_t id = rec.id;

    uint32_t *const H = lds + lane;                                   // [j*64]
    uint32_t *const E = lds + (size_t)(qcap + 1) * 64 + lane;         // WIDE only
    uint8_t *const QC = reinterpret_cast<uint8_t *>(lds + (size_t)(WIDE ? 2 : 1) * (qcap + 1) * 64) + lane * 4;
    // query code j of this lane: QC[(j >> 2) * 256 + (j & 3)]

    unsigned long long cells = 0;
    if (valid) {
        const int qlen = rec.len2, tlen = rec.len1, h0 = rec.h0;
        const uint8_t *q = io.qry + rec.qry_off;
        const uint8_t *t = io.ref + rec.ref_off;
        const int oe_del = c.o_del + c.e_del, oe_ins = c.o_ins + c.e_ins;
        const int e_del = c.e_del, e_ins = c.e_ins;

        for (int w4 = 0; w4 * 4 < qlen; w4++)
            *reinterpret_cast<uint32_t *>(QC + w4 * 256) = load_u32_unaligned(q + w4 * 4);

        // row -1 (bandedSWA.cpp:159-161); E starts at 0 everywhere
        {
            int prev = h0;
            for (int j = 0; j <= qlen; j++) {
                int v;
                if (j == 0) v = h0;
                else if (j == 1) v = h0 > oe_ins ? h0 - oe_ins : 0;
                else v = prev > e_ins ? prev - e_ins : 0;
                prev = v;
                H[j * 64] = (uint32_t)v;
                if (WIDE) E[j * 64] = 0u;
            }
        }
        // band clamp (bandedSWA.cpp:164-172)
        int w = c.w;
        {
            int lim = (int)((double)(qlen * c.max_sc + c.end_bonus - c.o_ins) / e_ins + 1.);
            lim = lim > 1 ? lim : 1; w = w < lim ? w : lim;
            lim = (int)((double)(qlen * c.max_sc + c.end_bonus - c.o_del) / e_del + 1.);
            lim = lim > 1 ? lim : 1; w = w < lim ? w : lim;
        }

        int best = h0, best_i = -1, best_j = -1, g_i = -1, gscore = -1, max_off = 0;
        int beg = 0, end = qlen;
        uint32_t tw = load_u32_unaligned(t);       // 4 reference bases, refreshed every 4 rows
        for (int i = 0; i < tlen; i++) {
            const int tc = (tw >> ((i & 3) * 8)) & 0xff;
            if ((i & 3) == 3 && i + 1 < tlen) tw = load_u32_unaligned(t + i + 1);
            // biased score bytes for this reference base (codes >= 4 are N)
            uint32_t rlo = c.row_lo[4], rhi = c.row_hi[4];
            rlo = tc == 0 ? c.row_lo[0] : rlo; rhi = tc == 0 ? c.row_hi[0] : rhi;
            rlo = tc == 1 ? c.row_lo[1] : rlo; rhi = tc == 1 ? c.row_hi[1] : rhi;
            rlo = tc == 2 ? c.row_lo[2] : rlo; rhi = tc == 2 ? c.row_hi[2] : rhi;
            rlo = tc == 3 ? c.row_lo[3] : rlo; rhi = tc == 3 ? c.row_hi[3] : rhi;

            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            int hleft = 0;
            if (beg == 0) { hleft = h0 - (c.o_del + e_del * (i + 1)); hleft = hleft > 0 ? hleft : 0; }
            int f = 0;
            uint32_t rowpk = 0;           // (row max << 16) | column, max over the row; ties -> later column
            int rowmax32 = 0, rowmax_j = -1;
            int j = beg;
            for (; j < end; j++) {
                int diag, e;
                if (WIDE) { diag = (int)H[j * 64]; e = (int)E[j * 64]; }
                else { uint32_t v = H[j * 64]; diag = (int)(v & 0xffffu); e = (int)(v >> 16); }
                uint32_t qc = QC[(j >> 2) * 256 + (j & 3)];
                qc = qc > 4u ? 4u : qc;
                int sc = (int)__builtin_amdgcn_perm(rhi, rlo, qc | 0x0c0c0c00u) - 128;
                int M = diag ? diag + sc : 0;
                int h = max(max(M, e), f);
                int t1 = M - oe_del; t1 = t1 > 0 ? t1 : 0;
                int en = e - e_del; en = en > t1 ? en : t1;
                if (WIDE) { H[j * 64] = (uint32_t)hleft; E[j * 64] = (uint32_t)en; }
                else H[j * 64] = (uint32_t)hleft | ((uint32_t)en << 16);
                hleft = h;
                if (WIDE) {
                    if (!(rowmax32 > h)) rowmax_j = j;
                    rowmax32 = h > rowmax32 ? h : rowmax32;
                } else {
                    uint32_t pk = ((uint32_t)h << 16) | (uint32_t)j;
                    rowpk = pk > rowpk ? pk : rowpk;
                }
                int t2 = M - oe_ins; t2 = t2 > 0 ? t2 : 0;
                f -= e_ins; f = f > t2 ? f : t2;
            }
            cells += (unsigned)(end > beg ? end - beg : 0);
            int rowmax;
            if (WIDE) rowmax = rowmax32;
            else { rowmax = (int)(rowpk >> 16); rowmax_j = end > beg ? (int)(rowpk & 0xffffu) : -1; }
            if (WIDE) { H[end * 64] = (uint32_t)hleft; E[end * 64] = 0u; }
            else H[end * 64] = (uint32_t)hleft;
            if (j == qlen) {
                if (!(gscore > hleft)) g_i = i;
                gscore = hleft > gscore ? hleft : gscore;
            }
            if (rowmax == 0) break;
            if (rowmax > best) {
                best = rowmax; best_i = i; best_j = rowmax_j;
                int off = rowmax_j - i; off = off < 0 ? -off : off;
                max_off = off > max_off ? off : max_off;
            } else if (c.zdrop > 0) {
                int di = i - best_i, dj = rowmax_j - best_j;
                if (di > dj) { if (best - rowmax - (di - dj) * e_del > c.zdrop) break; }
                else { if (best - rowmax - (dj - di) * e_ins > c.zdrop) break; }
            }
            // trim all-zero cells from both band edges (bandedSWA.cpp:234-237)
            if (WIDE) {
                for (j = beg; j < end && H[j * 64] == 0u && E[j * 64] == 0u; j++) {}
                beg = j;
                for (j = end; j >= beg && H[j * 64] == 0u && E[j * 64] == 0u; j--) {}
            } else {
                for (j = beg; j < end && H[j * 64] == 0u; j++) {}
                beg = j;
                for (j = end; j >= beg && H[j * 64] == 0u; j--) {}
            }
            end = j + 2 < qlen ? j + 2 : qlen;
        }
        score_out[id] = best;
        if (result_out) {
            gab_bsw_result r;
            r.score = best; r.qle = best_j + 1; r.tle = best_i + 1;
            r.gtle = g_i + 1; r.gscore = gscore; r.max_off = max_off;
            result_out[id] = r;
        }
    }
    // one atomic per wave for the cell counter
    for (int o = 32; o > 0; o >>= 1) cells += __shfl_xor(cells, o);
    if (lane == 0 && cells) atomicAdd(&st->cells, cells);
}


// ---- pass 4b: the DP with 8-bit cells, two columns per step ---------------------------------------------------
// Used for a query-length class when max(h0) + qcap * max(mat) <= 255 (true for the whole 151-bp read workload):
// every H/E value then fits a byte, a column is a u16 (E << 8 | H) and TWO columns share one LDS dword, so the
// row needs half the LDS (7 waves per CU at 128 columns instead of 3) and the inner loop one LDS read and one
// LDS write per two cells.  Query codes are nibbles, two per byte.  The band may start or end in the middle of a
// dword: those single cells are handled with 16-bit LDS accesses so that cells outside the band keep their stale
// contents, which later rows may read when the band grows (bandedSWA.cpp:217,234-237).
//   lds layout: [ (qcap + 2) / 2 dwords of cells ][ ((qcap + 1) / 2 + 3) / 4 dwords of query nibbles ]  x 64 lanes
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s16x2 pk_splat(int v) { s16x2 r; r.x = (short)v; r.y = (short)v; return r; }
// Packed 16-bit instructions the compiler does not form by itself: the x {0,1} product (as a C multiply it becomes
// compare + select per half), the u16 min, and the op_sel forms that route one half of a register to the other
// half of the result.  They are plain (non-volatile) asm so the scheduler may still move them.
__device__ __forceinline__ uint32_t pk_mul_lo(uint32_t a, uint32_t b) {
    uint32_t r; asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    uint32_t r; asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ uint32_t pk_subsat_u16_k(uint32_t a, uint32_t k) {    // max(half - k, 0), unsigned halves
    uint32_t r; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "s"(k)); return r;
}
__device__ __forceinline__ uint32_t and_or_b32(uint32_t a, uint32_t mask, uint32_t k) {   // (a & mask) | k, mask uniform
    uint32_t r; asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(k)); return r;
}
__device__ __forceinline__ uint32_t pk_max_i16_0(uint32_t a) {
    uint32_t r; asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(a)); return r;
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    uint32_t r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ uint32_t pk_min_u16_1(uint32_t a) {                    // min(half, 1) on both halves
    uint32_t r; asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(a)); return r;
}
// Non-packed 16-bit / 32-bit instructions of the 2-cycle class (profiles/r02_valu_issue.md), all operands in VGPRs.  On gfx9
// a 16-bit VOP2 instruction reads the low halves of its sources and writes zero to the upper half of its destination.
__device__ __forceinline__ uint32_t max_i16_lo(uint32_t a, uint32_t b) {
    uint32_t r; asm("v_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ uint32_t sub_u16_lo(uint32_t a, uint32_t b) {
    uint32_t r; asm("v_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ uint32_t shr16(uint32_t a) {
    uint32_t r; asm("v_lshrrev_b32 %0, 16, %1" : "=v"(r) : "v"(a)); return r;
}
__device__ __forceinline__ uint32_t add_u32_v(uint32_t a, uint32_t b) {
    uint32_t r; asm("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
struct BswCellOut { int h, en, f; };
__device__ __forceinline__ BswCellOut bsw_cell(int diag, int e, int f, uint32_t qc, uint32_t rlo, uint32_t rhi, int oe_del,
                                               int e_del, int oe_ins, int e_ins) {
    const int sc = (int)__builtin_amdgcn_perm(rhi, rlo, qc | 0x0c0c0c00u) - 128;
    const int M = diag ? diag + sc : 0;
    BswCellOut o;
    o.h = max(max(M, e), f);
    o.en = max(max(M - oe_del, e - e_del), 0);
    o.f = max(max(M - oe_ins, f - e_ins), 0);
    return o;
}

// SYM: o_del + e_del == o_ins + e_ins (BWA-MEM's defaults): M - (o + e) is the same value for the E and the F source.
// MS1: no score above 1 (BWA-MEM's a = 1): for diag >= 1, M <= diag + 1 <= 2 diag, so "diag == 0 -> M = 0" is min(M, 2 diag).
template <bool SYM, bool MS1>
__global__ __launch_bounds__(64) void bsw_dp8(BswIO io, BswConst c, const BswRec *__restrict__ recs, int64_t kbeg,
                                              int64_t kend, int qcap, int32_t *__restrict__ score_out,
                                              gab_bsw_result *__restrict__ result_out, BswStats *st) {
    extern __shared__ uint32_t lds[];
    __shared__ __attribute__((aligned(8))) uint32_t row_tab[10];     // BswConst.row_lo / row_hi per reference base, for per-lane look-ups
    if (threadIdx.x < 5) { row_tab[2 * threadIdx.x] = c.row_lo[threadIdx.x]; row_tab[2 * threadIdx.x + 1] = c.row_hi[threadIdx.x]; }
    __syncthreads();
    const int lane = threadIdx.x;
    const int64_t k = kbeg + (int64_t)(gridDim.x - 1 - blockIdx.x) * 64 + lane;   // heaviest waves (largest key) first
    const bool valid = k < kend;
    BswRec rec; rec.ref_off = rec.qry_off = 0; rec.len1 = rec.len2 = rec.h0 = 0; rec.id = 0u;
    if (valid) rec = recs[k];
    const uint32_t id = rec.id;
    const int ncell_dw = (qcap + 2) / 2;
    // cell j of this lane: 16 bits at byte address ((j >> 1) * 64 + lane) * 4 + (j & 1) * 2
    uint8_t *const CB = reinterpret_cast<uint8_t *>(lds + lane);
    uint32_t *const CW = lds + lane;                                   // pair p = columns 2p, 2p+1 at CW[p * 64]
    uint8_t *const QN = reinterpret_cast<uint8_t *>(lds + (size_t)ncell_dw * 64) + lane;   // nibble pair p at QN[p * 64]
#define CELL16(j) (*reinterpret_cast<uint16_t *>(CB + ((j) >> 1) * 256 + ((j) & 1) * 2))
#define QPAIR(p) (QN[(p) * 64])

    unsigned long long cells = 0;
    if (valid) {
        const int qlen = rec.len2, tlen = rec.len1, h0 = rec.h0;
        const uint8_t *q = io.qry + rec.qry_off;
        const uint8_t *t = io.ref + rec.ref_off;
        const int oe_del = c.o_del + c.e_del, oe_ins = c.o_ins + c.e_ins;
        const int e_del = c.e_del, e_ins = c.e_ins;

        // query -> nibbles (codes above 4 count as N)
        for (int p0 = 0; p0 * 2 < qlen; p0 += 2) {             // 4 bases -> 2 nibble bytes
            const uint32_t src = load_u32_unaligned(q + p0 * 2);
            for (int b = 0; b < 2; b++) {
                uint32_t lo = (src >> (b * 16)) & 0xff, hi = (src >> (b * 16 + 8)) & 0xff;
                lo = lo > 4u ? 4u : lo; hi = hi > 4u ? 4u : hi;
                QPAIR(p0 + b) = (uint8_t)(lo | hi << 4);
            }
        }
        // row -1 (bandedSWA.cpp:159-161); E = 0
        {
            int prev = h0;
            for (int j = 0; j <= qlen; j++) {
                int v;
                if (j == 0) v = h0;
                else if (j == 1) v = h0 > oe_ins ? h0 - oe_ins : 0;
                else v = prev > e_ins ? prev - e_ins : 0;
                prev = v;
                CELL16(j) = (uint16_t)v;
            }
        }
        int w = c.w;
        {
            int lim = (int)((double)(qlen * c.max_sc + c.end_bonus - c.o_ins) / e_ins + 1.);
            lim = lim > 1 ? lim : 1; w = w < lim ? w : lim;
            lim = (int)((double)(qlen * c.max_sc + c.end_bonus - c.o_del) / e_del + 1.);
            lim = lim > 1 ? lim : 1; w = w < lim ? w : lim;
        }
        int best = h0, best_i = -1, best_j = -1, g_i = -1, gscore = -1, max_off = 0;
        int beg = 0, end = qlen;
        uint32_t tw = load_u32_unaligned(t);
        for (int i = 0; i < tlen; i++) {
            const int tc = (tw >> ((i & 3) * 8)) & 0xff;
            if ((i & 3) == 3 && i + 1 < tlen) tw = load_u32_unaligned(t + i + 1);
            // the packed score vector of this row's reference base: one 8-byte LDS read (five v_cmp + ten v_cndmask + the SGPR
            // moves they need were ~100 cycles per row)
            const uint2 rr = *reinterpret_cast<const uint2 *>(&row_tab[2 * (tc < 4 ? tc : 4)]);
            const uint32_t rlo = rr.x, rhi = rr.y;
            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            int hleft = 0;
            if (beg == 0) { hleft = h0 - (c.o_del + e_del * (i + 1)); hleft = hleft > 0 ? hleft : 0; }
            int f = 0;
            uint32_t rowpk = 0;                   // (row max << 16) | column; ties -> later column
            int j = beg;
            // the two pair words the row can start with and their query codes: one LDS round trip for the row start
            uint32_t *cw = CW + (j >> 1) * 64;
            const uint8_t *qp = QN + (j >> 1) * 64;
            uint32_t v0 = cw[0], q0 = qp[0];
            {
                const uint32_t vn = cw[64], qn = qp[64];
                if ((j & 1) && j < end) {         // band starts on the upper half of a pair
                    const BswCellOut o = bsw_cell((int)((v0 >> 16) & 0xff), (int)(v0 >> 24), f, q0 >> 4, rlo, rhi, oe_del, e_del,
                                                  oe_ins, e_ins);
                    CELL16(j) = (uint16_t)(hleft | o.en << 8);
                    hleft = o.h; f = o.f;
                    rowpk = ((uint32_t)o.h << 16) | (uint32_t)j;
                    j++;
                    v0 = vn; q0 = qn; cw += 64; qp += 64;
                }
            }
            if (j + 1 < end) {
                // Two columns per step, the column-independent part in packed 16-bit lanes (lo = column j, hi = j+1):
                // M, E' and the F-source of both cells come from v_pk_* instructions; only the H / F carry chain
                // between the two cells is scalar.  The loop is unrolled by two pairs so that the software pipeline
                // (next pair's LDS words in flight while this pair is computed) needs no register copies.
                const uint32_t k_oe_del = as_u32(pk_splat(oe_del)), k_e_del = as_u32(pk_splat(e_del));
                const uint32_t k_oe_ins = as_u32(pk_splat(oe_ins));
                const uint32_t k_bias = 0x00800080u, k_nib = 0x000f000fu;
                // per-lane copies of wave-uniform constants: the third VOP3 source of v_and_or_b32, and the operands of the
                // 2-cycle instructions below (an SGPR or literal source makes them 4-cycle instructions, r02_valu_issue.md)
                uint32_t k_selz = 0x0c000c00u, k_lo8 = 0x00ff00ffu, v_e_ins = (uint32_t)e_ins;
                asm volatile("" : "+v"(k_selz), "+v"(k_lo8), "+v"(v_e_ins));
                // carried between pairs, both in the LOW half of their register (high half zero):
                // HB = H of the previous column, FV = F entering the pair
                uint32_t HB = (uint32_t)hleft, FV = (uint32_t)f;
                // M is clamped at 0 (unsigned saturating subtract): H, E' and F all take a max with a non-negative
                // value, so the clamp changes none of them and lets E' / F-source use saturating subtracts too.
                // The column-to-column carry chain (H[j] -> F -> H[j+1] -> F) uses one half per instruction anyway: it runs in
                // the NON-packed 16-bit instructions (v_max_i16 / v_sub_u16 on the low halves, two shifts to bring the upper
                // halves of ME and T down), which issue every 2.6 cycles at this occupancy instead of 4.2 (profiles/r02_valu_issue.md);
                // so do the plain 32-bit add of the two packed halves (no carry: both < 256) and the byte mask.
#define BSW_PAIR(V, QB, JJ, OUT)                                                                                \
    {                                                                                                             \
        const uint32_t d2 = (V) & k_lo8;                                         /* diag of both columns */      \
        const uint32_t e2 = __builtin_amdgcn_perm(0u, (V), 0x0c030c01u);         /* E of both columns */         \
        const uint32_t sel = and_or_b32((QB) * 0x1001u, k_nib, k_selz);          /* code j -> byte 0, j+1 -> 2 */\
        const uint32_t sc2 = __builtin_amdgcn_perm(rhi, rlo, sel);               /* biased scores per half */    \
        const uint32_t Mc = pk_subsat_u16_k(add_u32_v(d2, sc2), k_bias);                                         \
        const uint32_t M = MS1 ? pk_min_u16(Mc, add_u32_v(d2, d2)) : pk_mul_lo(Mc, pk_min_u16_1(d2));   /* diag == 0 -> M = 0 */ \
        const uint32_t Md = pk_subsat_u16_k(M, k_oe_del);                                                        \
        const uint32_t EN = pk_max_i16(Md, pk_subsat_u16_k(e2, k_e_del));                                        \
        const uint32_t T = SYM ? Md : pk_subsat_u16_k(M, k_oe_ins);                                              \
        const uint32_t ME = pk_max_i16(M, e2);                                                                   \
        const uint32_t HA = max_i16_lo(ME, FV);                                  /* H[j] */                      \
        const uint32_t FA = max_i16_lo(T, sub_u16_lo(FV, v_e_ins));              /* F leaving column j */        \
        const uint32_t hprev = HB;                                                                               \
        HB = max_i16_lo(shr16(ME), FA);                                          /* H[j+1] */                    \
        FV = max_i16_lo(shr16(T), sub_u16_lo(FA, v_e_ins));                      /* F leaving column j+1 */      \
        /* byte 0 = H[j-1] (hprev), byte 1 = E'[j], byte 2 = H[j] (HA byte 0), byte 3 = E'[j+1] */               \
        (OUT) = (EN << 8) | __builtin_amdgcn_perm(HA, hprev, 0x0c040c00u);                                       \
        const uint32_t pa = (HA << 16) | (uint32_t)(JJ), pb = (HB << 16) | (uint32_t)((JJ) + 1);                 \
        rowpk = max(max(rowpk, pa), pb);                                                                         \
    }
                for (; j + 3 < end; j += 4, cw += 128, qp += 128) {
                    const uint32_t v1 = cw[64], q1 = qp[64];
                    BSW_PAIR(v0, q0, j, cw[0])
                    v0 = cw[128]; q0 = qp[128];                      // pair index <= end / 2: inside the row allocation
                    BSW_PAIR(v1, q1, j + 2, cw[64])
                }
                if (j + 1 < end) {
                    const uint32_t v1 = cw[64], q1 = qp[64];         // the word a trailing single column lives in
                    BSW_PAIR(v0, q0, j, cw[0])
                    j += 2; v0 = v1; q0 = q1;
                }
                hleft = (int)HB; f = (int)FV;
#undef BSW_PAIR
            }
            if (j < end) {                        // band ends on the lower half of a pair: its word is already in v0
                const BswCellOut o = bsw_cell((int)(v0 & 0xff), (int)((v0 >> 8) & 0xff), f, q0 & 0xfu, rlo, rhi, oe_del, e_del,
                                              oe_ins, e_ins);
                CELL16(j) = (uint16_t)(hleft | o.en << 8);
                hleft = o.h; f = o.f;
                const uint32_t pk = ((uint32_t)o.h << 16) | (uint32_t)j;
                rowpk = pk > rowpk ? pk : rowpk;
                j++;
            }
            cells += (unsigned)(end > beg ? end - beg : 0);
            const int rowmax = (int)(rowpk >> 16);
            const int rowmax_j = end > beg ? (int)(rowpk & 0xffffu) : -1;
            CELL16(end) = (uint16_t)hleft;        // eh[end].h = h1, eh[end].e = 0
            if (j == qlen) {
                if (!(gscore > hleft)) g_i = i;
                gscore = hleft > gscore ? hleft : gscore;
            }
            if (rowmax == 0) break;
            if (rowmax > best) {
                best = rowmax; best_i = i; best_j = rowmax_j;
                int off = rowmax_j - i; off = off < 0 ? -off : off;
                max_off = off > max_off ? off : max_off;
            } else if (c.zdrop > 0) {
                int di = i - best_i, dj = rowmax_j - best_j;
                if (di > dj) { if (best - rowmax - (di - dj) * e_del > c.zdrop) break; }
                else { if (best - rowmax - (dj - di) * e_ins > c.zdrop) break; }
            }
            // Band trimming (bandedSWA.cpp:234-237).  The four cells next to either edge are fetched in ONE LDS round
            // trip (whole pair words; cells outside [beg, end] only ever shorten the count and the clamps below undo
            // that); the cell-by-cell loops of the reference run only when all four are zero.
            {
                const int pb = beg >> 1, pe = end >> 1;
                const uint32_t b0 = CW[pb * 64], b1 = CW[(pb + 1) * 64], b2 = CW[(pb + 2) * 64];
                const uint32_t e2 = CW[pe * 64], e1 = CW[(pe > 0 ? pe - 1 : 0) * 64], e0 = CW[(pe > 1 ? pe - 2 : 0) * 64];
                uint64_t x = (uint64_t)b1 << 32 | b0;                      // cells 2pb .. 2pb+3, low half first
                if (beg & 1) x = (x >> 16) | (uint64_t)(b2 & 0xffffu) << 48;
                const int lz = x ? __builtin_ctzll(x) >> 4 : 4;
                j = beg + lz;
                if (lz == 4) for (; j < end && CELL16(j) == 0; j++) {}
                beg = j < end ? j : end;
                uint64_t y = (uint64_t)e2 << 32 | e1;                      // cells 2pe-2 .. 2pe+1; cell `end` goes on top
                if (!(end & 1)) y = (y << 16) | (e0 >> 16);
                const int tz = y ? __builtin_clzll(y) >> 4 : 4;
                j = end - tz;
                if (tz == 4) for (; j >= beg && CELL16(j) == 0; j--) {}
                j = j > beg - 1 ? j : beg - 1;
            }
            end = j + 2 < qlen ? j + 2 : qlen;
        }
        score_out[id] = best;
        if (result_out) {
            gab_bsw_result r;
            r.score = best; r.qle = best_j + 1; r.tle = best_i + 1;
            r.gtle = g_i + 1; r.gscore = gscore; r.max_off = max_off;
            result_out[id] = r;
        }
    }
#undef CELL16
#undef QPAIR
    for (int o = 32; o > 0; o >>= 1) cells += __shfl_xor(cells, o);
    if (lane == 0 && cells) atomicAdd(&st->cells, cells);
}

}  // namespace

// =============================================================================== host side
struct gab_bsw {
    gab_tuning tun = gab_tuning_loaded();      // experiment knobs, read when the handle is made
    gab_host_stream hs;     // private stream of the host-pointer entry point(s)
    int device = 0;
    gab_bsw_params prm;
    BswConst cst;
    gab_devbuf ws;          // hist | start | qstart | stats | records in bucket order | rank
    gab_devbuf io;          // staging for the host-pointer entry point
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // total begin, dp begin, dp end, total end
    // the per-class DP launches rotate over the caller's stream and these, so that the draining tail of one class
    // overlaps the next classes (fork / join with the events)
    static constexpr int kAux = 1;        // 3 measured equal to 1
    hipStream_t aux[kAux] = {nullptr};
    hipEvent_t fork = nullptr, join[kAux] = {nullptr};
    bool have_stats = false;
    uint32_t *h_qstart = nullptr;   // pinned, kQBuckets + 1
    BswStats *h_stats = nullptr;    // pinned
    int64_t last_cells = 0;
    bool out_of_order = false;      // gab_bsw_run: a batch's sequences did not lie in pair order (its sampled window was rejected by the device)
};

extern "C" int gab_bsw_create(const gab_bsw_params *p, int device, gab_bsw **out) {
    if (!p || !out) { gab_set_error("gab_bsw_create: NULL argument"); return GAB_EINVAL; }
    *out = nullptr;
    GAB_CHECK(p->e_del > 0 && p->e_ins > 0 && p->o_del >= 0 && p->o_ins >= 0,
              "gab_bsw_create: gap penalties must be o>=0, e>0 (got o_del=%d e_del=%d o_ins=%d e_ins=%d)",
              p->o_del, p->e_del, p->o_ins, p->e_ins);
    GAB_CHECK(p->w >= 0 && p->zdrop >= 0, "gab_bsw_create: w and zdrop must be >= 0");
    GAB_CHECK(p->o_del + p->e_del < 32768 && p->o_ins + p->e_ins < 32768 && p->end_bonus >= -32768 &&
              p->end_bonus < 32768, "gab_bsw_create: penalties out of 16-bit range");
    int rc = gab_check_device(device);
    if (rc) return rc;
    gab_device_guard g(device);
    gab_bsw *h = new (std::nothrow) gab_bsw();
    if (!h) { gab_set_error("out of host memory"); return GAB_ENOMEM; }
    h->device = device; h->prm = *p;
    BswConst &c = h->cst;
    c.o_del = p->o_del; c.e_del = p->e_del; c.o_ins = p->o_ins; c.e_ins = p->e_ins;
    c.zdrop = p->zdrop; c.end_bonus = p->end_bonus; c.w = p->w;
    int mx = 0;
    for (int k = 0; k < 25; k++) mx = p->mat[k] > mx ? p->mat[k] : mx;
    c.max_sc = mx;
    for (int t = 0; t < 5; t++) {
        uint32_t lo = 0;
        for (int q = 0; q < 4; q++) lo |= (uint32_t)(uint8_t)(p->mat[t * 5 + q] + 128) << (8 * q);
        c.row_lo[t] = lo;
        c.row_hi[t] = (uint32_t)(uint8_t)(p->mat[t * 5 + 4] + 128);
    }
    for (int k = 0; k < 4; k++)
        if (hipEventCreate(&h->ev[k]) != hipSuccess) { gab_set_error("hipEventCreate failed"); delete h; return GAB_EDEVICE; }
    bool aux_ok = hipEventCreateWithFlags(&h->fork, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < gab_bsw::kAux && aux_ok; k++)
        aux_ok = hipStreamCreateWithFlags(&h->aux[k], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&h->join[k], hipEventDisableTiming) == hipSuccess;
    if (!aux_ok) { gab_set_error("gab_bsw_create: stream / event creation failed"); delete h; return GAB_EDEVICE; }
    // the 256-base class needs more than the default 64 KiB of dynamic LDS
    if (hipFuncSetAttribute((const void *)bsw_dp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void *)bsw_dp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void *)bsw_dp8<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess ||
        hipFuncSetAttribute((const void *)bsw_dp8<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess ||
        hipFuncSetAttribute((const void *)bsw_dp8<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess ||
        hipFuncSetAttribute((const void *)bsw_dp8<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess) {
        gab_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); delete h; return GAB_EDEVICE;
    }
    if (hipHostMalloc((void **)&h->h_qstart, sizeof(uint32_t) * (kQBuckets + 1)) != hipSuccess ||
        hipHostMalloc((void **)&h->h_stats, sizeof(BswStats)) != hipSuccess) {
        gab_set_error("hipHostMalloc failed"); delete h; return GAB_ENOMEM;
    }
    *out = h;
    return GAB_OK;
}

extern "C" void gab_bsw_destroy(gab_bsw *h) {
    if (!h) return;
    gab_device_guard g(h->device);
    h->ws.release(); h->io.release(); h->hs.release();
    for (int k = 0; k < 4; k++) if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
    if (h->fork) (void)hipEventDestroy(h->fork);
    for (int k = 0; k < gab_bsw::kAux; k++) {
        if (h->join[k]) (void)hipEventDestroy(h->join[k]);
        if (h->aux[k]) (void)hipStreamDestroy(h->aux[k]);
    }
    if (h->h_qstart) (void)hipHostFree(h->h_qstart);
    if (h->h_stats) (void)hipHostFree(h->h_stats);
    delete h;
}

static int bsw_run_device_impl(gab_bsw *h, const uint8_t *ref, int64_t ref_bytes, const int64_t *ref_off,
                               const uint8_t *qry, int64_t qry_bytes, const int64_t *qry_off,
                               const int32_t *len1, const int32_t *len2, const int32_t *h0, int64_t n,
                               int32_t *score_out, gab_bsw_result *result_out, void *stream_, int64_t ref_lo, int64_t qry_lo,
                               int64_t ref_hi, int64_t qry_hi);
extern "C" int gab_bsw_run_device(gab_bsw *h, const uint8_t *ref, int64_t ref_bytes, const int64_t *ref_off,
                                  const uint8_t *qry, int64_t qry_bytes, const int64_t *qry_off,
                                  const int32_t *len1, const int32_t *len2, const int32_t *h0, int64_t n,
                                  int32_t *score_out, gab_bsw_result *result_out, void *stream_) {
    return bsw_run_device_impl(h, ref, ref_bytes, ref_off, qry, qry_bytes, qry_off, len1, len2, h0, n, score_out, result_out, stream_, 0, 0,
                               ref_bytes, qry_bytes);
}
static int bsw_run_device_impl(gab_bsw *h, const uint8_t *ref, int64_t ref_bytes, const int64_t *ref_off,
                               const uint8_t *qry, int64_t qry_bytes, const int64_t *qry_off,
                               const int32_t *len1, const int32_t *len2, const int32_t *h0, int64_t n,
                               int32_t *score_out, gab_bsw_result *result_out, void *stream_, int64_t ref_lo, int64_t qry_lo,
                               int64_t ref_hi, int64_t qry_hi) {
    GAB_CHECK(h, "gab_bsw_run_device: NULL handle");
    GAB_CHECK(n >= 0 && n < (1ll << 31), "gab_bsw_run_device: n=%lld out of range", (long long)n);
    h->have_stats = false;
    if (n == 0) return GAB_OK;
    GAB_CHECK(ref && ref_off && qry && qry_off && len1 && len2 && h0 && score_out,
              "gab_bsw_run_device: NULL buffer");
    gab_device_guard g(h->device);
    hipStream_t s = (hipStream_t)stream_;

    // workspace carve-up
    const size_t o_hist = 0;
    const size_t o_start = o_hist + sizeof(uint32_t) * kNumKeys;
    const size_t o_qstart = o_start + sizeof(uint32_t) * (kNumKeys + 1);
    const size_t o_sums = o_qstart + sizeof(uint32_t) * (kQBuckets + 1);
    const size_t o_stats = (o_sums + sizeof(uint32_t) * kScanBlocks + 15) & ~(size_t)15;
    const size_t o_recs = (o_stats + sizeof(BswStats) + 255) & ~(size_t)255;
    const size_t o_rank = o_recs + ((sizeof(BswRec) * (size_t)n + 255) & ~(size_t)255);
    int rc = h->ws.reserve(o_rank + sizeof(uint32_t) * (size_t)n);
    if (rc) return rc;
    char *base = h->ws.as<char>();
    uint32_t *d_hist = (uint32_t *)(base + o_hist), *d_start = (uint32_t *)(base + o_start);
    uint32_t *d_qstart = (uint32_t *)(base + o_qstart);
    BswStats *d_stats = (BswStats *)(base + o_stats);
    BswRec *d_recs = (BswRec *)(base + o_recs);             // the pairs' records in bucket order
    uint32_t *d_rank = (uint32_t *)(base + o_rank);

    BswIO io{ref, ref_off, qry, qry_off, len1, len2, h0, ref_bytes, qry_bytes, n, ref_lo, qry_lo, ref_hi, qry_hi};
    GAB_HIP(hipEventRecord(h->ev[0], s));
    GAB_HIP(hipMemsetAsync(base, 0, o_recs, s));
    {
        BswStats init; memset(&init, 0, sizeof(init)); init.first_bad = 0x7fffffff;
        // first_bad uses atomicMin, so it starts at INT_MAX (set by a tiny H2D after the memset)
        *h->h_stats = init;
        GAB_HIP(hipMemcpyAsync(d_stats, h->h_stats, sizeof(BswStats), hipMemcpyHostToDevice, s));
    }
    int grid = (int)(gab_ceil_div(n, 256) < 1024 ? gab_ceil_div(n, 256) : 1024);   // (one same-address atomicMax per wave)
    hipLaunchKernelGGL(bsw_hist, dim3(grid), dim3(256), 0, s, io, d_hist, d_rank, d_stats);
    hipLaunchKernelGGL(bsw_scan_a, dim3(kScanBlocks), dim3(1024), 0, s, d_hist, d_start, (uint32_t *)(base + o_sums));
    hipLaunchKernelGGL(bsw_scan_b, dim3(kScanBlocks), dim3(1024), 0, s, d_start, (const uint32_t *)(base + o_sums), d_qstart);
    hipLaunchKernelGGL(bsw_scatter, dim3(grid), dim3(256), 0, s, io, d_start, d_rank, d_recs);
    GAB_HIP(hipMemcpyAsync(h->h_qstart, d_qstart, sizeof(uint32_t) * (kQBuckets + 1), hipMemcpyDeviceToHost, s));
    GAB_HIP(hipMemcpyAsync(h->h_stats, d_stats, sizeof(BswStats), hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));   // launch geometry of the DP depends on the class sizes
    if (h->h_stats->bad) {
        gab_set_error("gab_bsw_run_device: %d pair(s) violate the limits (first: pair %d): need 1<=len2<=%d, "
                      "1<=len1<=%d, 0<=h0, offsets inside the slabs with 3 more readable bytes behind every sequence",
                      h->h_stats->bad, h->h_stats->first_bad - 1, GAB_BSW_MAX_QLEN, GAB_BSW_MAX_TLEN);
        return GAB_EINVAL;
    }
    // 16-bit packing is valid iff every H/E value < 2^15: H <= h0 + qlen * max_sc
    const bool wide = (int64_t)h->h_stats->max_h0 + (int64_t)GAB_BSW_MAX_QLEN * h->cst.max_sc > 32767;

    GAB_HIP(hipEventRecord(h->ev[1], s));
    GAB_HIP(hipEventRecord(h->fork, s));
    for (int k = 0; k < gab_bsw::kAux; k++) GAB_HIP(hipStreamWaitEvent(h->aux[k], h->fork, 0));
    int nlaunch = 0;
    hipStream_t s_main = s;
    // Small batches (fewer waves than fill the chip a few times over) go out as ONE launch sized for their longest query:
    // eight class launches of a couple of hundred waves each leave most CUs idle and pay eight launch latencies
    // (100 k pairs: 3.0 -> 1.x ms); the records are in key order, so the heaviest waves still start first.
    const bool one_launch = n <= (int64_t)64 * 8192;
    int top_cls = kNumClasses - 1;
    while (top_cls > 0 && h->h_qstart[(top_cls + 1) * kClassStep] <= h->h_qstart[top_cls * kClassStep]) top_cls--;
    for (int cls = kNumClasses - 1; cls >= 0; cls--) {      // longest queries first: the tail of the step is made of short work
        int64_t kb = h->h_qstart[cls * kClassStep], ke = h->h_qstart[(cls + 1) * kClassStep];
        if (one_launch) { if (cls != top_cls) continue; kb = 0; }
        if (ke <= kb) continue;
        const int qcap = (cls + 1) * kClassStep;
        const int blocks = (int)gab_ceil_div(ke - kb, 64);
        s = (nlaunch % (gab_bsw::kAux + 1)) ? h->aux[nlaunch % (gab_bsw::kAux + 1) - 1] : s_main;
        nlaunch++;
        // 8-bit cells when every H/E value of this class fits a byte: H <= h0 + qlen * max_sc
        if ((int64_t)h->h_stats->max_h0 + (int64_t)qcap * h->cst.max_sc <= 255 && h->cst.max_sc >= 0) {
            const size_t lds8 = sizeof(uint32_t) * 64 * ((size_t)(qcap + 2) / 2 + ((size_t)(qcap + 1) / 2 + 3) / 4 + 1);
            const bool sym = h->cst.o_del + h->cst.e_del == h->cst.o_ins + h->cst.e_ins;
            const bool ms1 = h->cst.max_sc <= 1;
            auto kern = sym ? (ms1 ? bsw_dp8<true, true> : bsw_dp8<true, false>) : (ms1 ? bsw_dp8<false, true> : bsw_dp8<false, false>);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds8, s, io, h->cst, d_recs, kb, ke, qcap,
                               score_out, result_out, d_stats);
            continue;
        }
        const size_t lds = sizeof(uint32_t) * 64 * ((size_t)(wide ? 2 : 1) * (qcap + 1) + (size_t)(qcap + 3) / 4);
        if (wide)
            hipLaunchKernelGGL(bsw_dp<true>, dim3(blocks), dim3(64), lds, s, io, h->cst, d_recs, kb, ke, qcap,
                               score_out, result_out, d_stats);
        else
            hipLaunchKernelGGL(bsw_dp<false>, dim3(blocks), dim3(64), lds, s, io, h->cst, d_recs, kb, ke, qcap,
                               score_out, result_out, d_stats);
    }
    s = s_main;
    GAB_HIP(hipGetLastError());
    for (int k = 0; k < gab_bsw::kAux; k++) {
        GAB_HIP(hipEventRecord(h->join[k], h->aux[k]));
        GAB_HIP(hipStreamWaitEvent(s, h->join[k], 0));
    }
    GAB_HIP(hipEventRecord(h->ev[2], s));
    GAB_HIP(hipMemcpyAsync(h->h_stats, d_stats, sizeof(BswStats), hipMemcpyDeviceToHost, s));
    GAB_HIP(hipEventRecord(h->ev[3], s));
    h->have_stats = true;
    return GAB_OK;
}

extern "C" int gab_bsw_run(gab_bsw *h, const uint8_t *ref, const int64_t *ref_off, const uint8_t *qry,
                           const int64_t *qry_off, const int32_t *len1, const int32_t *len2,
                           const int32_t *h0, int64_t n, int32_t *score_out) {
    GAB_CHECK(h, "gab_bsw_run: NULL handle");
    GAB_CHECK(n >= 0 && n < (1ll << 31), "gab_bsw_run: n=%lld out of range", (long long)n);
    if (n == 0) return GAB_OK;
    GAB_CHECK(ref && ref_off && qry && qry_off && len1 && len2 && h0 && score_out, "gab_bsw_run: NULL buffer");
    gab_device_guard g(h->device);
    gab_tuning_refresh(&h->tun);
    const bool trace = h->tun.bsw_trace;      // GAB_BSW_TRACE, diagnosis: per-phase wall times of this call on stderr
    auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t_0 = now();
    // extent of the two slabs actually referenced: only [min, max) is staged, so a driver can hand a window
    // of a big input (absolute offsets) to each GPU without re-basing its offset arrays.  A scan of all n offsets here is
    // ~1 ms per million pairs of serial host work inside the caller's ROI, and the device validates every pair against the
    // staged window anyway (bsw_hist): the window is first taken from 66 pairs spread over the batch -- exact whenever the
    // sequences lie in pair order, as loadPairs lays them out (main_banded.cpp:164-206) -- and only a batch the device
    // rejects for it is scanned in full and staged again.
    int64_t rb = 0, qb = 0, ra = INT64_MAX, qa = INT64_MAX;
    auto take = [&](int64_t i) {
        int64_t r = ref_off[i] + len1[i], q = qry_off[i] + len2[i];
        rb = r > rb ? r : rb; qb = q > qb ? q : qb;
        ra = ref_off[i] < ra ? ref_off[i] : ra; qa = qry_off[i] < qa ? qry_off[i] : qa;
        return ref_off[i] >= 0 && qry_off[i] >= 0 && len1[i] >= 0 && len2[i] >= 0;
    };
    bool sampled = n > 4096 && !h->out_of_order && !h->tun.bsw_full_scan;      // (a batch of this handle was rejected for its sampled window: scan from then on)
    if (sampled) {
        bool ok = take(0);
        ok = take(n - 1) && ok;
        for (int k = 1; k <= 64; k++) ok = take((n - 1) * k / 65) && ok;
        // (a window the sample makes absurd -- negative fields, or more than 64 KB per pair -- is not worth a copy: scan)
        if (!ok || rb - ra > 65536 * n || qb - qa > 65536 * n) sampled = false;
    }
    if (!sampled) {
        rb = qb = 0; ra = qa = INT64_MAX;
        for (int64_t i = 0; i < n; i++)
            GAB_CHECK(take(i), "gab_bsw_run: negative offset/length at pair %lld", (long long)i);
    }
    ra &= ~(int64_t)255; qa &= ~(int64_t)255;          // keep the device alignment of the slab origin
    const size_t rpad = ((size_t)(rb - ra) + 3 + 255) & ~(size_t)255, qpad = ((size_t)(qb - qa) + 3 + 255) & ~(size_t)255;
    const size_t nn = (size_t)n;
    size_t o = 0;
    const size_t o_ref = o; o += rpad;
    const size_t o_qry = o; o += qpad;
    const size_t o_roff = o; o += 8 * nn;
    const size_t o_qoff = o; o += 8 * nn;
    const size_t o_l1 = o; o += 4 * nn;
    const size_t o_l2 = o; o += 4 * nn;
    const size_t o_h0 = o; o += 4 * nn;
    const size_t o_sc = o; o += 4 * nn;
    int rc = h->io.reserve(o);
    if (rc) return rc;
    char *b = h->io.as<char>();
    hipStream_t s = nullptr;
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    const double t_1 = now();
    {   // the copies of one chunk at a time per GPU (gab_core.hip: the workers of a GPU must not copy in lockstep)
        std::lock_guard<std::mutex> gate(gab_h2d_mutex(h->device));
        GAB_HIP(hipMemcpyAsync(b + o_ref, ref + ra, (size_t)(rb - ra), hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_qry, qry + qa, (size_t)(qb - qa), hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_roff, ref_off, 8 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_qoff, qry_off, 8 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_l1, len1, 4 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_l2, len2, 4 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_h0, h0, 4 * nn, hipMemcpyHostToDevice, s));
        GAB_HIP(hipStreamSynchronize(s));
    }
    double t_2 = 0;
    if (trace) { GAB_HIP(hipStreamSynchronize(s)); t_2 = now(); }
    // virtual slab origins: device address of byte 0 of the caller's slabs
    rc = bsw_run_device_impl(h, (const uint8_t *)(b + o_ref) - ra, ra + (int64_t)rpad, (const int64_t *)(b + o_roff),
                             (const uint8_t *)(b + o_qry) - qa, qa + (int64_t)qpad, (const int64_t *)(b + o_qoff),
                             (const int32_t *)(b + o_l1), (const int32_t *)(b + o_l2), (const int32_t *)(b + o_h0),
                             n, (int32_t *)(b + o_sc), nullptr, s, ra, qa, rb, qb);      // (rb / qb: a pair that ends in the window's padding was not copied)
    if (rc == GAB_EINVAL && sampled) {
        // a pair outside the sampled window (sequences not in pair order) -- or a really invalid one: the full scan tells
        h->out_of_order = true;
        return gab_bsw_run(h, ref, ref_off, qry, qry_off, len1, len2, h0, n, score_out);
    }
    if (rc) return rc;
    double t_3 = 0;
    if (trace) { GAB_HIP(hipStreamSynchronize(s)); t_3 = now(); }
    GAB_HIP(hipMemcpyAsync(score_out, b + o_sc, 4 * nn, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    if (trace)
        fprintf(stderr, "[gab_bsw_run %p] %lld pairs: host scan %.2f ms, H2D of %.1f MB %.2f ms, sort + DP %.2f ms, D2H %.2f ms\n", (void *)h,
                (long long)n, t_1 - t_0, (double)((rb - ra) + (qb - qa) + 28 * n) / 1e6, t_2 - t_1, t_3 - t_2, now() - t_3);
    return GAB_OK;
}

// Pre-size the handle's device buffers (staging of the host-pointer entry point + the sort workspace) for calls of up to
// max_pairs pairs / the given slab windows, so that the first gab_bsw_run inside a timed region does not pay for them (the
// reference allocates its per-thread F16_ / H16_ buffers in the BandedPairWiseSW constructor, bandedSWA.cpp:80-96).
extern "C" int gab_bsw_reserve(gab_bsw *h, int64_t max_pairs, int64_t max_ref_bytes, int64_t max_qry_bytes) {
    GAB_CHECK(h, "gab_bsw_reserve: NULL handle");
    GAB_CHECK(max_pairs >= 0 && max_pairs < (1ll << 31) && max_ref_bytes >= 0 && max_qry_bytes >= 0, "gab_bsw_reserve: size out of range");
    gab_device_guard g(h->device);
    const size_t nn = (size_t)max_pairs;
    int rc = h->io.reserve(std::max<size_t>((((size_t)max_ref_bytes + 3 + 511) & ~(size_t)255) + (((size_t)max_qry_bytes + 3 + 511) & ~(size_t)255) + 32 * nn + 1024,
                                            (size_t)4 << 20));      // (at least the 4 MB gab_warm_copy_engines moves)
    if (rc) return rc;
    rc = h->ws.reserve(sizeof(uint32_t) * (2 * kNumKeys + kQBuckets + 4) + sizeof(BswStats) + 1024 + (sizeof(BswRec) + sizeof(uint32_t)) * nn);
    if (rc) return rc;
    hipStream_t s = nullptr;
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    // touch the memory and load the kernels' code objects once
    GAB_HIP(hipMemsetAsync(h->io.p, 0, h->io.cap, s));
    GAB_HIP(hipMemsetAsync(h->ws.p, 0, h->ws.cap, s));
    GAB_HIP(hipStreamSynchronize(s));
    if ((rc = gab_warm_copy_engines(s, h->io.p, h->io.cap)) != GAB_OK) return rc;
    // ... and one tiny batch through the whole path: the first launch of every kernel (code object, the runtime's per-kernel
    // bookkeeping, the auxiliary streams' queues) costs milliseconds once per process and handle -- not inside the caller's ROI
    // (the drop-in driver with the GPU parser: 55.7 ms for a step that takes 45)
    static const uint8_t seq[40] = {0, 1, 2, 3, 0, 1, 2, 3, 3, 2, 1, 0, 0, 1, 2, 3, 0, 1, 2, 3, 3, 2, 1, 0, 0, 1, 2, 3, 0, 1, 2, 3, 3, 2, 1, 0, 0, 0, 0, 0};
    const int64_t ro[2] = {0, 4}, qo[2] = {2, 8};
    const int32_t l1[2] = {24, 20}, l2[2] = {20, 16}, h0[2] = {10, 0};
    int32_t sc[2];
    const bool had = h->have_stats;
    rc = gab_bsw_run(h, seq, ro, seq, qo, l1, l2, h0, 2, sc);
    h->have_stats = had;
    return rc;
}

extern "C" int gab_bsw_last_stats(gab_bsw *h, int64_t *cells, float *kernel_ms, float *total_ms) {
    GAB_CHECK(h, "gab_bsw_last_stats: NULL handle");
    GAB_CHECK(h->have_stats, "gab_bsw_last_stats: no completed run on this handle");
    gab_device_guard g(h->device);
    GAB_HIP(hipEventSynchronize(h->ev[3]));
    if (cells) *cells = (int64_t)h->h_stats->cells;
    if (kernel_ms) GAB_HIP(hipEventElapsedTime(kernel_ms, h->ev[1], h->ev[2]));
    if (total_ms) GAB_HIP(hipEventElapsedTime(total_ms, h->ev[0], h->ev[3]));
    return GAB_OK;
}
