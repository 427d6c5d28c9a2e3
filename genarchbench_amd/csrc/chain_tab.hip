// chain / fast-chain -- the TABLE form: one long call spread over the whole chip.
//
// Semantics: chain_dp of /root/reference/benchmarks/chain/src/host_kernel.cpp:30-94 and of
// fast-chain/src/host_kernel.cpp:175-683 (AVX2 / AVX-512 arithmetic), exactly as chain.hip computes them.
//
// Why.  score[i] needs score[i - 1], so a call is a chain of dependent blocks, and in the block / latency forms of chain.hip
// ONE workgroup on ONE CU does everything a call needs: ~230 (anchor, predecessor) pairs per anchor at ~28 VALU instructions
// each.  A 60 000-anchor call takes 5-10 ms that way whatever else the chip does (profiles/r03_chain_latency.md) -- the floor
// under every shard of an 8-GPU run and under the 1 000-call inputs.  But all a pair costs is GEOMETRY: window
// membership, the four filters, min(dq, dr, q_span) and the gap cost involve no score at all.  What does involve scores is one
// add and one max per pair.  So the work is split in two by what it depends on, not by call:
//
//   1. ctab_geo   -- every block of 64 anchors of every call is one workgroup anywhere on the chip (no dependence between
//                    them): for each predecessor row j of the block -- from the first anchor's window start to the block's last
//                    anchor -- and each anchor a, ONE BYTE  G'[j][a] = oc - gc + bias  (1 .. 255), or 0 when the pair is
//                    filtered / outside a's window / j >= a.  `bias` = the call's largest gap cost + 1, so a byte is enough
//                    whenever q_span + max gap cost <= 254 (every call of the suite: 15 + 79); calls that need more keep the
//                    other kernels.  Rows are grouped by 16: a lane's 16 bytes of a group are one uint4, a group is 1 KB,
//                    written once and read once -- 2 bytes of HBM traffic per pair against ~12 VALU instructions.
//   2. ctab_fold  -- one workgroup per call walks its blocks in order and only FOLDS: worker waves take the far groups
//                    (predecessors older than the previous block: scores final, kept in an LDS ring): v = byte + (score -
//                    bias), a byte-is-zero select and a max -- 4 VALU per pair -- tracking the maximum per 16-row group;
//                    they also expand the 128 near / in-block rows into the key form of chain_fast_kernel
//                    ((G << 7) | code, INT_MIN when filtered); the main wave folds those with the key maximum (readlane +
//                    add + max per row; the larger code = the newer predecessor wins a tie, the initial key's code 127
//                    beats a predecessor that merely equals q_span); a resolver wave, one block behind, turns "far, group
//                    g" into the parent index (the newest row of the group that attains the score) and, for chain, checks
//                    the max_skip certificate -- at most 25 unfiltered predecessors newer than the argmax (proof:
//                    chain_hw_kernel in chain.hip) -- EXACTLY, by counting non-zero bytes.  A call that misses it anywhere
//                    sets its bail word and is run again by the kernels of chain.hip (no call of the suite does).
//   Window starts need no sequential pointer when x ascends (checked per call): st(i) = max(lower bound of x[i] - max_dist_x,
//   i - 5000), a binary search per anchor (ctab_st).
//
// Roofline: ctab_geo is VALU-bound (~15 instructions per row of 64 pairs) with 1 B per pair written; ctab_fold reads that
// byte once.  HBM: 2 B per pair, ~600 B per anchor at 230-pair windows -- far above the 24 B per anchor of the block forms, and
// the reason the table form is only used for the calls a batch would otherwise wait for.
#include "gab_internal.h"
#include "chain_dev.h"
#include <algorithm>
#include <type_traits>
#include <limits.h>
#include <vector>
#include <stdlib.h>
#include <string.h>

namespace {

struct TabCall {
    int64_t blk0;          // its first block in the block array
    int64_t grp0;          // its first 16-row group in the table (ctab_place)
    int32_t nblk;
    int32_t ngrp;          // groups of all its blocks (ctab_st adds them up)
    int32_t bias;          // byte = oc - gc + bias
    int32_t gt_off;        // its gap-cost table in the gtab array: bw + 2 entries, gc - bias (entry bw + 1: a pair with dd > bw)
    int32_t ok;            // eligible (ctab_prep)
    int32_t pad;
};
struct TabBlock {
    int64_t grp;           // first group of the block in the table
    int32_t jrow0;         // predecessor of row 0 of group 0 (call-relative; may be negative: rows before the call are padding)
    int32_t ng;            // groups: far ones first, then 4 of the previous block (not for block 0), then 4 of the block itself
    int32_t call, i0;      // the call it belongs to, its first anchor
    int32_t pad[2];
};
struct TabCounters { unsigned long long groups_needed; uint32_t no_room, rescans; };      // rescans: anchors whose certificate missed and whose exact scan confirmed the result

constexpr int kTabNone = (int)0x80000000;      // a filtered pair in the key form
constexpr int kTabSoftNone = -(1 << 22);       // ... while keys are still being added up (ctab_fold's unit closure): four of them do not overflow,
constexpr int kTabSoftLim = -(1 << 21);        //     and a sum with one of them stays below this (a real key's |G << 7 | code| < 2^15, three hops < 2^17)
constexpr int kTabNegH = -(1 << 23);           // below every far value (scores >= 0, G >= -255)

// ---- 1. per call: is it eligible, its gap table, its bias -----------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void ctab_prep(const ChainWork *__restrict__ work, TabCall *calls, uint32_t *bail,
                                                 const uint64_t *__restrict__ xs, const uint64_t *__restrict__ ys, int32_t *gtab) {
    constexpr bool FC = MODE == GAB_FASTCHAIN;
    __shared__ unsigned long long s_lo[4], s_hi[4];
    __shared__ int s_flag[4], s_qmin[4], s_qmax[4], s_gmax[4];
    const ChainWork w = work[blockIdx.x];
    TabCall &tc = calls[blockIdx.x];
    const uint64_t *X = xs + w.off, *Y = ys + w.off;
    const int64_t n = w.n;
    unsigned long long lo = ~0ull, hi = 0;
    int flag = 0, qmin = 255, qmax = 0;                        // flag bit 0: segment ids differ, bit 1: x does not ascend
    const uint32_t sid0 = n > 0 ? (uint32_t)(Y[0] >> 48 & 0xff) : 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const unsigned long long x = X[i], y = Y[i];
        lo = x < lo ? x : lo; hi = x > hi ? x : hi;
        flag |= ((uint32_t)(y >> 48 & 0xff) != sid0) ? 1 : 0;
        flag |= (i > 0 && X[i - 1] > x) ? 2 : 0;
        const int qs = (int)(y >> 32 & 0xff);
        qmin = qs < qmin ? qs : qmin; qmax = qs > qmax ? qs : qmax;
    }
    // the gap costs a pair can pass the filters with: dd = 0 .. bw
    const bool bw_ok = w.bw >= 0 && w.bw <= kGapTab - 2;
    const double avg_d = (double)w.avg_qspan;
    const float k32 = (float)(0.01 * (double)w.avg_qspan);
    const bool avg_ok = w.avg_qspan >= 0.f && w.avg_qspan <= 4096.f;       // (false for a NaN)
    int gmax = 0;
    if (bw_ok && avg_ok)
        for (int d = threadIdx.x; d <= w.bw; d += 256) {
            int gc;
            if (FC) {
                const int lgh = 15 - (__clz((int)((uint32_t)d | 1u)) >> 1);
                gc = (int32_t)floorf(__fmul_rn((float)d, k32)) + lgh;
                const int gd = (int32_t)__dmul_rn(__dmul_rn((double)d, .01), avg_d) + lgh;     // the scalar tail's cost (narrow windows)
                gmax = gd > gmax ? gd : gmax;
            } else gc = chain_gap_cost(d, avg_d);
            gmax = gc > gmax ? gc : gmax;
        }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; flag |= __shfl_xor(flag, o);
        const int a = __shfl_xor(qmin, o), b = __shfl_xor(qmax, o), g = __shfl_xor(gmax, o);
        qmin = a < qmin ? a : qmin; qmax = b > qmax ? b : qmax; gmax = g > gmax ? g : gmax;
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_lo[wv] = lo; s_hi[wv] = hi; s_flag[wv] = flag; s_qmin[wv] = qmin; s_qmax[wv] = qmax; s_gmax[wv] = gmax; }
    __syncthreads();
    for (int k = 0; k < 4; k++) {
        lo = s_lo[k] < lo ? s_lo[k] : lo; hi = s_hi[k] > hi ? s_hi[k] : hi; flag |= s_flag[k];
        qmin = s_qmin[k] < qmin ? s_qmin[k] : qmin; qmax = s_qmax[k] > qmax ? s_qmax[k] : qmax; gmax = s_gmax[k] > gmax ? s_gmax[k] : gmax;
    }
    const int32_t mq = w.max_dist_y < w.max_dist_x ? w.max_dist_y : w.max_dist_x;
    const unsigned long long lim = mq < 0 ? 0ull : (unsigned long long)mq;
    bool ok = n >= 1 && n < (1 << 24) && !(flag & 2) && qmin >= 1 && avg_ok && bw_ok && w.max_dist_x >= 0 && w.max_dist_x < (1 << 30) &&
              (long long)n * qmax < (1 << 24) - (1 << 15);                     // (a score fits the 24 bits above the key's code)
    if (FC) ok = ok && lim <= (1u << 20);
    else ok = ok && !(flag & 1) && hi - lo + lim < 0x7fffffffull && hi <= ~0ull - (unsigned long long)w.max_dist_x;   // chain_facts_kernel's "plain", and x + max_dist_x cannot wrap
    const int bias = gmax + 1;
    ok = ok && qmax + bias <= 255;
    if (ok) {
        for (int d = threadIdx.x; d <= w.bw; d += 256) {
            int gc;
            if (FC) gc = (int32_t)floorf(__fmul_rn((float)d, k32)) + (15 - (__clz((int)((uint32_t)d | 1u)) >> 1));
            else gc = chain_gap_cost(d, avg_d);
            gtab[tc.gt_off + d] = gc - bias;
        }
        if (threadIdx.x == 0) gtab[tc.gt_off + w.bw + 1] = 1 << 20;           // dd > bw: the byte comes out as 0
    }
    if (threadIdx.x == 0) { tc.bias = bias; tc.ok = ok ? 1 : 0; tc.ngrp = 0; bail[blockIdx.x] = ok ? 0u : 1u; }
}

// ---- 2. block descriptors: which call, which anchors (one workgroup per call) --------------------------------------------------
__global__ __launch_bounds__(256) void ctab_blocks_init(const TabCall *__restrict__ calls, TabBlock *blocks) {
    const TabCall tc = calls[blockIdx.x];
    for (int k = threadIdx.x; k < tc.nblk; k += 256) {
        TabBlock b;
        b.grp = 0; b.jrow0 = 0; b.ng = 0; b.call = (int32_t)blockIdx.x; b.i0 = k * 64; b.pad[0] = b.pad[1] = 0;
        blocks[tc.blk0 + k] = b;
    }
}

// the window test of the start search: chain host_kernel.cpp:56-57, fast-chain host_kernel.cpp:200-207 (unsigned difference)
template <bool FC> __device__ __forceinline__ bool ctab_beyond(uint64_t xi, uint64_t xj, uint64_t mdx64) { return FC ? (xi - xj) > mdx64 : xi > xj + mdx64; }

// ---- 3. window starts (one wave per block) ----------------------------------------------------------------------------------------
// The reference advances ONE pointer: while (st < i && beyond(x[i], x[st])) ++st; then st = max(st, i - max_iter).  With x
// ascending `beyond(x[i], x[j])` is true for j below a bound lb(i) and false from there on, lb never decreases with i, and neither
// does i - max_iter: the pointer after anchor i is max(lb(i), i - max_iter), whatever happened before.
template <int MODE>
__global__ __launch_bounds__(256) void ctab_st(const ChainWork *__restrict__ work, TabCall *calls, TabBlock *blocks, int64_t nblocks,
                                               const uint64_t *__restrict__ xs, int32_t *st_all) {
    constexpr bool FC = MODE == GAB_FASTCHAIN;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nblocks) return;
    const int lane = threadIdx.x & 63;
    TabBlock &tb = blocks[b];
    const int c = tb.call, i0 = tb.i0;
    if (!calls[c].ok) return;
    const ChainWork w = work[c];
    const uint64_t *X = xs + w.off;
    const int n = (int)w.n;
    const uint64_t mdx64 = (uint64_t)(int64_t)w.max_dist_x;
    const int ia = i0 + lane;
    const bool mine = ia < n;
    int st = 0;
    if (mine) {
        const uint64_t xi = X[ia];
        int lo = 0, hi = ia;                                   // beyond(x[i], x[i]) is false: the bound is at most i
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ctab_beyond<FC>(xi, X[mid], mdx64)) lo = mid + 1; else hi = mid;
        }
        st = lo > ia - kMaxIter ? lo : ia - kMaxIter;
        st_all[w.off + ia] = st;
    }
    // rows of the block: from the first anchor's start -- but never later than the previous block's first anchor, so that the last
    // eight groups are always the previous block and the block itself -- to the block's last anchor, padded in FRONT to 16
    const int st_first = __builtin_amdgcn_readfirstlane(st);
    const int near0 = i0 >= 64 ? i0 - 64 : 0;
    const int jlo = st_first < near0 ? st_first : near0;
    const int ng = (i0 + 64 - jlo + 15) >> 4;
    if (lane == 0) { tb.ng = ng; tb.jrow0 = i0 + 64 - 16 * ng; atomicAdd(&calls[c].ngrp, ng); }
}

// ---- 4. a place in the table for every call that fits (one workgroup), then for its blocks ---------------------------------
__global__ __launch_bounds__(1024) void ctab_place(TabCall *calls, uint32_t *bail, int ncalls, long long budget_groups, TabCounters *ct) {
    __shared__ long long s_sum[1024];
    const int per = (ncalls + 1023) / 1024;
    const int k0 = threadIdx.x * per, k1 = min(k0 + per, ncalls);
    long long sum = 0;
    for (int k = k0; k < k1; k++) sum += calls[k].ok ? calls[k].ngrp : 0;
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { long long run = 0; for (int k = 0; k < 1024; k++) { const long long v = s_sum[k]; s_sum[k] = run; run += v; } ct->groups_needed = (unsigned long long)run; }
    __syncthreads();
    long long run = s_sum[threadIdx.x];
    uint32_t noroom = 0;
    for (int k = k0; k < k1; k++) {
        if (!calls[k].ok) continue;
        calls[k].grp0 = run;
        run += calls[k].ngrp;
        if (run > budget_groups) { bail[k] = 1u; noroom++; }   // (the list is sorted longest first: what does not fit is the short end)
    }
    if (noroom) atomicAdd(&ct->no_room, noroom);
}
__global__ __launch_bounds__(256) void ctab_block_offsets(const TabCall *__restrict__ calls, const uint32_t *__restrict__ bail, TabBlock *blocks) {
    __shared__ int s_w[4];
    __shared__ long long s_base;
    if (bail[blockIdx.x]) return;
    const TabCall tc = calls[blockIdx.x];
    if (threadIdx.x == 0) s_base = tc.grp0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k0 = 0; k0 < tc.nblk; k0 += 256) {
        const int k = k0 + threadIdx.x;
        const int v = k < tc.nblk ? blocks[tc.blk0 + k].ng : 0;
        const int inc = wave_incl_sum(v);
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        int before = 0;
        for (int q = 0; q < wv; q++) before += s_w[q];
        const long long base = s_base;
        if (k < tc.nblk) blocks[tc.blk0 + k].grp = base + before + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) s_base = base + before + inc;
        __syncthreads();
    }
}

// ---- 5. geometry ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ctab_sad(uint32_t a, uint32_t b) { uint32_t r; asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b)); return r; }

// One workgroup (4 waves) per block of 64 anchors; a wave takes every fourth group of 16 predecessor rows.  Lane a <-> anchor
// i0 + a; the predecessors of a group sit in lanes 0 .. 15 and are broadcast with v_readlane.
// The byte of a pair (chain: chain_geometry_plain; fast-chain: fastchain_body's score_pred with sj = 0):
//   dr = x[i] - x[j] (low words; exact: the call's facts), dq = y32[i] - y32[j], dd = |dr - dq|,
//   oc = min(dr, dq, q_span), unfiltered iff dr != 0, 1 <= dq <= min(max_dist_x, max_dist_y), dd <= bw (chain, n_segs > 1: and
//   dr <= max_dist_y).  With x ascending dr >= 0 inside a window, so "dr != 0 and dq >= 1" is "oc >= 1" (q_span >= 1: checked
//   per call); gtab[min(dd, bw + 1)] = gc - bias, and a huge value at bw + 1, so dd > bw clamps to the byte 0 by itself.
// The sixteen rows of one group for the wave's 64 anchors -> the group's 1 KB.  INSIDE: every row lies in every anchor's window
// (no window test per pair); NARROW (fast-chain): some anchor of the block takes the scalar tail's double-precision gap cost.
// Both are wave-uniform and decided ONCE per group by the caller: as run-time flags inside the unrolled rows the compiler turned
// them into a branch per row -- 32 basic blocks that nothing could be scheduled across (fast-chain: see the rows below).
#ifndef GAB_GEO_CHAIN_BATCHED      // experiment: chain's geometry rows eight at a time like fast-chain's
#define GAB_GEO_CHAIN_BATCHED 0
#endif
template <bool FC, bool MSEG, bool INSIDE, bool NARROW>
__device__ __forceinline__ uint4 ctab_geo_rows(uint32_t px, uint32_t py, uint32_t xa, int32_t qa, int32_t qs, const int32_t *gap, int32_t bw,
                                               uint32_t dq_lim, int32_t mdy, int j0, int st_a, uint32_t wspan, bool narrow, double avg_d, int bias,
                                               int &bad, bool inside_rt) {
    uint32_t bytes[16];
    if (NARROW || (!FC && !GAB_GEO_CHAIN_BATCHED)) {                // row by row: the rare form of fast-chain (the first blocks of a call), and chain
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t xj = (uint32_t)__builtin_amdgcn_readlane((int)px, k), yj = (uint32_t)__builtin_amdgcn_readlane((int)py, k);
            const int32_t dr = (int32_t)(xa - xj), dq = (int32_t)((uint32_t)qa - yj);
            const uint32_t dd = ctab_sad((uint32_t)dr, (uint32_t)dq);
            const uint32_t idx = min(dd, (uint32_t)bw + 1u);
            const int32_t oc = min(min(dr, dq), qs);
            int32_t gv = oc - gap[idx];
            if (NARROW) {
                const int32_t lgh = 15 - (__clz((int)(dd | 1u)) >> 1);
                const int32_t gd = (int32_t)__dmul_rn(__dmul_rn((double)(int32_t)dd, .01), avg_d) + lgh;
                gv = (narrow && dd <= (uint32_t)bw) ? oc - gd + bias : gv;
            }
            bool ok = oc >= 1 && (uint32_t)dq <= dq_lim;
            if (MSEG) ok = ok && dr <= mdy;
            if (!INSIDE && !inside_rt) ok = ok && (uint32_t)(j0 + k - st_a) < wspan;       // (inside_rt: chain decides per group at run time)
            if (NARROW) bad |= (ok && narrow && dd <= (uint32_t)bw && (gv < 1 || gv > 255)) ? 1 : 0;
            bytes[k] = ok ? (uint32_t)max(gv, 0) : 0u;
        }
    } else {
        // fast-chain, eight rows at a time: first everything that does not need the gap cost -- and the eight table reads, issued
        // together and UNCONDITIONALLY (left alone the compiler sinks each read behind its pair's filters: a divergent branch and a wait
        // for LDS per row) -- then the eight bytes.  Measured (rank 0's share of chain-large on 8 GPUs, two runs each): fast-chain
        // 4.57 -> 4.26 ms, the 1 000-call input 4.28 -> 4.03 ms (three runs each); chain (-DGAB_GEO_CHAIN_BATCHED=1) 4.68-4.72 -> 4.72-4.76 ms -- its
        // filters reject whole rows more often (77 % of the lanes active against 91 %) and the branch skips them: chain keeps the rows
        // one by one.  (gv <= q_span + bias <= 255: ctab_prep admits a call only then; dd > bw reads the
        // huge entry bw + 1 and clamps to 0.)
#pragma unroll
        for (int h = 0; h < 16; h += 8) {
            int32_t oc[8], gc[8];
            bool ok[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t xj = (uint32_t)__builtin_amdgcn_readlane((int)px, h + k), yj = (uint32_t)__builtin_amdgcn_readlane((int)py, h + k);
                const int32_t dr = (int32_t)(xa - xj), dq = (int32_t)((uint32_t)qa - yj);
                const uint32_t dd = ctab_sad((uint32_t)dr, (uint32_t)dq);
                oc[k] = min(min(dr, dq), qs);
                gc[k] = gap[min(dd, (uint32_t)bw + 1u)];
                ok[k] = oc[k] >= 1 && (uint32_t)dq <= dq_lim;
                if (MSEG) ok[k] = ok[k] && dr <= mdy;
                if (!INSIDE) ok[k] = ok[k] && (uint32_t)(j0 + h + k - st_a) < wspan;
            }
            // (the eight values are "used" here, all at once: no read can be moved behind its pair's filters, and one wait serves all)
            asm volatile("" : "+v"(gc[0]), "+v"(gc[1]), "+v"(gc[2]), "+v"(gc[3]), "+v"(gc[4]), "+v"(gc[5]), "+v"(gc[6]), "+v"(gc[7]));
#pragma unroll
            for (int k = 0; k < 8; k++) bytes[h + k] = ok[k] ? (uint32_t)max(oc[k] - gc[k], 0) : 0u;
        }
    }
    uint4 o;
    o.x = bytes[0] | bytes[1] << 8 | bytes[2] << 16 | bytes[3] << 24;
    o.y = bytes[4] | bytes[5] << 8 | bytes[6] << 16 | bytes[7] << 24;
    o.z = bytes[8] | bytes[9] << 8 | bytes[10] << 16 | bytes[11] << 24;
    o.w = bytes[12] | bytes[13] << 8 | bytes[14] << 16 | bytes[15] << 24;
    return o;
}

template <int MODE, bool MSEG>
__global__ __launch_bounds__(256) void ctab_geo(const ChainWork *__restrict__ work, const TabCall *__restrict__ calls, const TabBlock *__restrict__ blocks,
                                                uint32_t *bail, const int32_t *__restrict__ gtab, const int32_t *__restrict__ st_all,
                                                const uint64_t *__restrict__ xs, const uint64_t *__restrict__ ys, uint4 *T8) {
    constexpr bool FC = MODE == GAB_FASTCHAIN;
    __shared__ int32_t gap[kGapTab];
    const TabBlock tb = blocks[blockIdx.x];
    const int c = tb.call;
    if (bail[c] || tb.ng == 0) return;
    const ChainWork w = work[c];
    const TabCall tc = calls[c];
    for (int d = threadIdx.x; d <= w.bw + 1; d += 256) gap[d] = gtab[tc.gt_off + d];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t *X = xs + w.off, *Y = ys + w.off;
    const int n = (int)w.n, i0 = tb.i0;
    const int nb = n - i0 < 64 ? n - i0 : 64;
    const int ia = i0 + lane;
    const bool mine = lane < nb;
    const uint64_t xa64 = mine ? X[ia] : 0, ya64 = mine ? Y[ia] : 0;
    const uint32_t xa = (uint32_t)xa64;
    const int32_t qa = (int32_t)(uint32_t)ya64, qs = (int32_t)(ya64 >> 32 & 0xff);
    const int st_a = mine ? st_all[w.off + ia] : 0;
    const uint32_t wspan = mine ? (uint32_t)(ia - st_a) : 0u;          // row j is in a's window and older than a iff (unsigned)(j - st_a) < wspan
    const int st_hi = __builtin_amdgcn_readlane(st_a, nb - 1);           // (the pointer never moves back: the last anchor's start is the largest)
    const int32_t mq = w.max_dist_y < w.max_dist_x ? w.max_dist_y : w.max_dist_x;
    const uint32_t dq_lim = mq < 0 ? 0u : (uint32_t)mq;
    const int32_t bw = w.bw;
    const int32_t mdy = (MSEG && w.n_segs > 1) ? w.max_dist_y : 0x7fffffff;      // chain, n_segs > 1: dr > max_dist_y is filtered too
    // fast-chain: an anchor whose window holds <= 6 predecessors takes the scalar tail's double-precision gap cost
    // (fast-chain host_kernel.cpp:356-402); only near the start of a call
    const bool narrow = FC && mine && (ia - 1) - st_a <= 5;
    const bool any_narrow = FC && __ballot(narrow) != 0;
    const double avg_d = (double)w.avg_qspan;
    int bad = 0;
    for (int g = wave; g < tb.ng; g += 4) {
        const int j0 = tb.jrow0 + 16 * g;
        const int jl = j0 + (lane & 15);
        uint32_t px = 0, py = 0;
        if (lane < 16 && jl >= 0 && jl < n) { px = (uint32_t)X[jl]; py = (uint32_t)Y[jl]; }
        // (rows before the call or behind its last anchor keep x = y = 0: whatever they give is masked by the window test --
        // a group that holds such rows is never `inside`)
        const bool inside = nb == 64 && j0 >= st_hi && j0 >= 0 && j0 + 15 < i0;
        uint4 o;
#define GAB_GEO_ROWS(INSIDE, NARROW, RT) ctab_geo_rows<FC, MSEG, INSIDE, NARROW>(px, py, xa, qa, qs, gap, bw, dq_lim, mdy, j0, st_a, wspan, narrow, avg_d, tc.bias, bad, RT)
        if (!FC && !GAB_GEO_CHAIN_BATCHED) o = GAB_GEO_ROWS(false, false, inside);            // chain: one copy of the rows, taken one by one (see ctab_geo_rows)
        else if (!FC) o = inside ? GAB_GEO_ROWS(true, false, false) : GAB_GEO_ROWS(false, false, false);
        else if (any_narrow) o = inside ? GAB_GEO_ROWS(true, true, false) : GAB_GEO_ROWS(false, true, false);
        else o = inside ? GAB_GEO_ROWS(true, false, false) : GAB_GEO_ROWS(false, false, false);
#undef GAB_GEO_ROWS
        T8[(tb.grp + g) * 64 + lane] = o;
    }
    if (FC && __ballot(bad != 0) && lane == 0) bail[c] = 1u;      // a byte does not hold this call after all: the other kernels take it
}

// ---- 6. the fold ---------------------------------------------------------------------------------------------------------------
// Everything a phase needs from global memory is requested one phase (one block) earlier and kept in registers across the
// barrier: the first version asked for the bytes of a block at the top of its phase and every block waited out a trip to
// memory (3.4 us per block, 3.2 ms for one 60 000-anchor call; the table is complete before this kernel starts, so reading
// ahead is free of any ordering concern).
#ifndef GAB_TAB_W
#define GAB_TAB_W 14      // (12 until the last day of r04: once the main wave had shed the merge and -- fast-chain -- the previous block's rows, the
#endif                 //  far workers were the phase's longest waves: 14 = sixteen waves, the most a workgroup can hold; chain share 4.73 -> 4.58 ms)
constexpr int kTabW = GAB_TAB_W;          // worker waves (-DGAB_TAB_W=..: tuning builds): four close the block's own rows, the others take the far groups and the previous block's rows
constexpr int kTabFW = kTabW - 4;         // ... the far workers
static_assert(kTabFW >= 2, "the table form needs at least two far workers");
constexpr int kTabF = 4;                  // far groups per worker held in registers (deeper windows: loaded when needed)
constexpr int kTabRing = 8192;            // scores (minus bias) of the newest anchors: the deepest window is 5000 + 2 blocks + padding
#ifndef GAB_TAB_MAX_RESCANS
#define GAB_TAB_MAX_RESCANS 64
#endif
constexpr int kTabMaxRescans = GAB_TAB_MAX_RESCANS;
constexpr int kTabMaxPatch = 8;            // anchors of a call whose result max_skip really changed (see the resolver)
#ifndef GAB_TAB_NEAR_HELP
#define GAB_TAB_NEAR_HELP 1
#endif
#ifndef GAB_TAB_NEAR_HELP_FC      // who folds the previous block's rows: 0 the main wave, 4 the unit workers, 8 the far workers
#define GAB_TAB_NEAR_HELP_FC 4
#endif
#ifndef GAB_TAB_NEAR_HELP_CH
#define GAB_TAB_NEAR_HELP_CH 0
#endif
#ifndef GAB_TAB_MERGE_ATOMIC
#define GAB_TAB_MERGE_ATOMIC 1
#endif
#ifndef GAB_TAB_NEAR_LDS
#define GAB_TAB_NEAR_LDS 1
#endif
#ifndef GAB_KO_CERT_FAR          // timing experiments (wrong results for calls whose certificate misses)
#define GAB_KO_CERT_FAR 0
#endif
#ifndef GAB_KO_CERT_NEAR
#define GAB_KO_CERT_NEAR 0
#endif
#ifndef GAB_KO_OKH
#define GAB_KO_OKH 0
#endif
#ifndef GAB_KO_MAIN_NEAR      // (with GAB_KO_CERT_FAR = GAB_KO_CERT_NEAR = 1: the keys are garbage, nothing may act on them)
#define GAB_KO_MAIN_NEAR 0
#endif
#ifndef GAB_KO_MAIN_BLOCK
#define GAB_KO_MAIN_BLOCK 0
#endif
#ifndef GAB_KO_MAIN_MERGE
#define GAB_KO_MAIN_MERGE 0
#endif
struct TabDesc { long long grp; int jrow0, ng; };       // what ctab_fold needs of a TabBlock
struct TabLds {
    int4 G4[2][2][16][64];                // [slot][previous block | block itself][row / 4][anchor]: keys of 4 rows
    int32_t ring[kTabRing + 16];          // (+ the first 16 entries again: sixteen consecutive scores never wrap)
    int32_t part_best[2][kTabFW][64], part_g[2][kTabFW][64];
    long long part64[2][64];              // GAB_TAB_MERGE_ATOMIC: the far workers' maxima merged by LDS atomics: (best << 32) | group
    int32_t res_key[3][64], res_fg[3][64];
    uint16_t okh[4][8][64];               // chain: one bit per unfiltered near / in-block pair, 16 rows per unit
    TabDesc desc[256];                    // the descriptors of the blocks around the one in work (see ctab_fold)
    int32_t near_key[64], near_cnt, near_pad[3];      // GAB_TAB_NEAR_HELP: the previous block's rows folded by the four unit workers (atomic max), and how many have
    int32_t pk[64];                       // the main wave's (and its helpers'): the previous block's scores << 7 (read back as broadcasts, four per load)
    int32_t stop[2];                      // 1: the call goes back to chain.hip; 2: start again from block restart_blk (a new patch)
    int32_t restart_blk, patch_n;
    int32_t patch_blk[kTabMaxPatch], patch_lane[kTabMaxPatch], patch_score[kTabMaxPatch], patch_parent[kTabMaxPatch];
};
__device__ __forceinline__ uint32_t gab_tab_xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu; }   // HW_REG_XCC_ID
__device__ __forceinline__ int ctab_nonzero_bytes(uint32_t w) {
    return __popc((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u);
}

template <int MODE, bool TRACE>      // TRACE: GAB_CHAIN_TRACE's cycle counters and stamps (a variant of its own: they cost the loop eight scalar registers)
__global__ __launch_bounds__(64 * (2 + kTabW))
void ctab_fold(const ChainWork *__restrict__ work, const TabCall *__restrict__ calls, const TabBlock *__restrict__ blocks, uint32_t *bail,
               const uint4 *__restrict__ T8, const int32_t *__restrict__ st_all, const uint64_t *__restrict__ xs, const uint64_t *__restrict__ ys,
               int32_t *score_out, int32_t *parent_out, int32_t *gmarks_all, unsigned long long *evals_out,
               TabCounters *ct, unsigned long long *dbg_arg, int32_t *host_score, int32_t *host_parent) {
    unsigned long long *const dbg = TRACE ? dbg_arg : nullptr;
    constexpr bool FC = MODE == GAB_FASTCHAIN;
    constexpr int NF = kTabFW;
    extern __shared__ __attribute__((aligned(16))) uint8_t tab_lds_raw[];
    TabLds &L = *reinterpret_cast<TabLds *>(tab_lds_raw);
    const int c = blockIdx.x;
    const ChainWork w = work[c];
    const TabCall tc = calls[c];
    if (bail[c]) return;
    const TabBlock *B = blocks + tc.blk0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t *X = xs + w.off, *Y = ys + w.off;
    const int32_t *ST = st_all + w.off;
    int32_t *S = score_out + w.off, *P = parent_out + w.off;
    int32_t *GM = FC ? nullptr : gmarks_all + w.off;
    const int n = (int)w.n, nblocks = tc.nblk, bias = tc.bias;
    if (threadIdx.x < 2) L.stop[threadIdx.x] = 0;
    if (threadIdx.x == 0) { L.patch_n = 0; L.restart_blk = 0; }
    if (dbg && threadIdx.x == 0) dbg[32 + 3 * (size_t)c] = (wall_clock64() << 4) | gab_tab_xcc_id();      // GAB_CHAIN_TRACE: when (and on which XCD) the call's workgroup started
#if defined(GAB_TAB_PRIO_ALL)
    __builtin_amdgcn_s_setprio(3);
#elif defined(GAB_TAB_PRIO)
    if (wave == 0) __builtin_amdgcn_s_setprio(3);            // the main wave carries the call's only chain of dependent steps
#elif defined(GAB_TAB_PRIO_MIX)
    __builtin_amdgcn_s_setprio(wave == 0 ? 3 : 2);
#endif

    // what a wave keeps across the phases
    int32_t pbest = 0;                                       // main: the previous block's scores
    int32_t near_target = 0;                                 // main: helpers' reports it has waited for so far (GAB_TAB_NEAR_HELP)
    int32_t qs_next = (wave == 0 && lane < n) ? (int32_t)(Y[lane] >> 32 & 0xff) : 0;       // ... the next block's q_span, a phase early
    unsigned long long evals = 0, evals_exact = 0;           // resolver (evals_exact: per-lane counts of the exact re-scans)
    int n_rescans = 0;
    bool stopped = false;
    // workers: descriptor of the block they take next, and its bytes (their G units, their first kTabF far groups)
    // Block descriptors come from an LDS ring of 256, refilled 64 at a time by the resolver wave: as scalar loads from memory
    // (s_load, counted by lgkmcnt like every LDS operation) the next wait for ANY LDS read stalled on them -- a trip to L2 or HBM
    // in every phase of every wave
    using Desc = TabDesc;
    // (chain only, and only its resolver, which uses its descriptor at once: +6 %.  The workers ask two phases ahead and keep their
    // scalar loads, and the fast-chain instantiation has no ring at all: with it -- even unused by the workers -- one fast-chain call
    // took 2.56 instead of 2.29 ms; the instantiations are scheduled differently by the compiler and nothing else explains it)
    auto desc_ring = [&](int kb) { Desc d{0, 0, 0}; if (kb < nblocks) d = L.desc[kb & 255]; return d; };
    auto desc_of = [&](int kb) { Desc d{0, 0, 0}; if (kb < nblocks) { const TabBlock tb = B[kb]; d.grp = tb.grp; d.jrow0 = tb.jrow0; d.ng = tb.ng; } return d; };
    if (!FC) {
        for (int b = threadIdx.x; b < 192 && b < nblocks; b += 64 * (2 + kTabW)) { const TabBlock tb = B[b]; L.desc[b] = Desc{tb.grp, tb.jrow0, tb.ng}; }
        __syncthreads();
    }
    const int wk = wave - 2;
    int min_blk = 0;                                         // blocks below this one are final (a restart does not touch them)
    // a patched anchor: max_skip cut the reference's scan of it short of the plain maximum -- its score and parent are what the
    // reference's own scan gave (chain_exact_global), nothing is folded into it, and what it passes on is that score
    auto patched = [&](int blk) { bool pl = false; for (int q = 0; q < L.patch_n; q++) pl |= L.patch_blk[q] == blk && L.patch_lane[q] == lane; return pl; };
    // (requested ONE block ahead.  Two blocks ahead -- a second set of registers rotated every phase, 184 VGPRs instead of 128 --
    // was measured slower everywhere: one call 2.42 -> 2.92 ms, 256 calls 6.8 -> 8.1 ms)
    Desc d_cur = desc_of(0), d_nxt = desc_of(1);
    uint4 pg[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)}, pf[kTabF];
#pragma unroll
    for (int q = 0; q < kTabF; q++) pf[q] = make_uint4(0, 0, 0, 0);
    // worker wk < 4 closes rows 16 wk .. of the block itself (unit 4 + wk); the others (far worker fw = wk - 4) share the previous
    // block's four units and the far groups
    const int fw = wk - 4;
    auto unit_of = [&](int q) { return wk < 4 ? (q == 0 ? 4 + wk : 8) : (fw + NF * q < 4 ? fw + NF * q : 8); };
    auto prefetch = [&](const Desc &d, int kb) {            // the bytes of block kb for this worker
        const int nfar = d.ng - (kb > 0 ? 8 : 4);
        const uint4 *T = T8 + d.grp * 64 + lane;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int u = unit_of(q);
            pg[q] = make_uint4(0, 0, 0, 0);
            if (u < 8 && !(u < 4 && kb == 0)) pg[q] = T[(size_t)(nfar + (kb > 0 ? u : u - 4)) * 64];
        }
#pragma unroll
        for (int q = 0; q < kTabF; q++) {
            const int fgi = fw + NF * q;
            pf[q] = make_uint4(0, 0, 0, 0);
            if (fw >= 0 && fgi < nfar) pf[q] = T[(size_t)fgi * 64];
        }
    };
    if (wave >= 2 && nblocks > 0) prefetch(d_cur, 0);
    // resolver: the block it resolves next (two behind the main wave): its descriptor, the far lanes' group, their window starts
    Desc r_desc{0, 0, 0};
    uint4 r_gw = make_uint4(0, 0, 0, 0);
    int32_t r_st = 0;
    if (threadIdx.x < 128) (&L.part64[0][0])[threadIdx.x] = LLONG_MIN;
    if (threadIdx.x < 64) L.near_key[threadIdx.x] = kTabNone;
    if (threadIdx.x == 0) L.near_cnt = 0;
    __syncthreads();

    unsigned long long busy = 0, t_all = dbg ? clock64() : 0;      // (GAB_CHAIN_TRACE: cycles of every wave outside the barrier, call 0 only)
    // The loop over the phases exists THREE times, once per role (a generic lambda instantiated for the main wave, the resolver and
    // the workers): in one shared loop the values of every role were live for all sixteen waves and the chain instantiation
    // spilled 106 scalar registers into VGPR lanes, reloaded by v_readlane on the main wave's critical path.  Every copy runs the
    // same phases and the same barriers.
    auto phases = [&](auto role_tag) {
    constexpr int ROLE = decltype(role_tag)::value;          // 0 main wave, 1 resolver, 2 unit workers (wk < 4), 3 far workers
    if (ROLE == 2) __builtin_assume(wk >= 0 && wk < 4);
    if (ROLE == 3) __builtin_assume(wk >= 4 && wk < 8);      // far workers that also expand a unit of the previous block
    if (ROLE == 4) __builtin_assume(wk >= 8);                // far workers that only fold far groups
    for (int t = -1; t <= nblocks + 1; t++) {
        const int par = (t + 1) & 1;                         // slot of block t + 1 in the two-deep arrays; block t lives in par ^ 1
        const unsigned long long t_in = dbg ? clock64() : 0;
#if GAB_TAB_NEAR_HELP
        // The previous block's 64 rows (keys of the rows + the scores the main wave left in `pk` at the end of the last phase) were a
        // sixth of the main wave's phase, and the main wave is the phase's critical path: the four unit workers take sixteen rows
        // each FIRST thing in the phase, merge by an LDS atomic max and count themselves done; the main wave waits for the count (a
        // spin on an LDS word inside the phase: every wave of the workgroup is resident, and the helpers wait for nothing).
        // fast-chain only: one call 2.13 -> 2.02 ms, a share of an 8-GPU run 4.21 -> 4.14 ms, all of fast-chain-large 24.95 -> 24.49 ms;
        // chain's unit workers also build the certificate's bit masks and are not idle enough (shares 4.68 -> 4.74 ms with it).
        // (kHelp = 4: the unit workers, sixteen rows each; 8: the far workers, eight rows each; 0: the main wave itself)
        constexpr int kHelp = FC ? GAB_TAB_NEAR_HELP_FC : GAB_TAB_NEAR_HELP_CH;
        const int hk = kHelp == 4 ? wk : wk - 4;                 // which helper this wave is
        if (ROLE >= 2 && kHelp && t >= min_blk && t < nblocks && t > 0 && hk >= 0 && hk < kHelp) {
            constexpr int kPer = 16 / (kHelp ? kHelp : 1);       // int4 words (of four rows) per helper
            const int4 *gnh = &L.G4[par ^ 1][0][kPer * hk][lane];
            const int4 *pkh = reinterpret_cast<const int4 *>(L.pk) + kPer * hk;
            int32_t k16 = kTabNone;
#pragma unroll
            for (int g4 = 0; g4 < kPer; g4++) {
                const int4 g = gnh[(size_t)g4 * 64], pv = pkh[g4];
                k16 = max(max(k16, g.x + pv.x), max(g.y + pv.y, max(g.z + pv.z, g.w + pv.w)));
            }
            atomicMax(&L.near_key[lane], k16);
            if (lane == 0) atomicAdd(&L.near_cnt, 1);        // (behind the wave's atomic max: a wave's LDS operations execute in order)
        }
#endif
        if constexpr (ROLE == 0) {
            if (t >= min_blk && t < nblocks) {
                // ------------------------------------------------ main wave: block t
                const int i0 = t * 64;
                const int nb = n - i0 < 64 ? n - i0 : 64;
                const bool mine = lane < nb;
                const int32_t qsa = mine ? qs_next : 0;
                qs_next = i0 + 64 + lane < n ? (int32_t)(Y[i0 + 64 + lane] >> 32 & 0xff) : 0;
                // the workers' far maxima: groups interleave, so the larger group (= the newer predecessors) wins a tie
                int32_t fbest = kTabNegH, fg = -1;
#if GAB_TAB_MERGE_ATOMIC
                {   // (merged by the workers themselves: one 64-bit LDS atomic max per worker and lane; the main wave reads ONE value)
                    const long long pk = L.part64[par ^ 1][lane];
                    L.part64[par ^ 1][lane] = LLONG_MIN;
                    if (pk != LLONG_MIN) { fbest = (int32_t)(pk >> 32); fg = (int32_t)(uint32_t)pk; }
                }
#else
#pragma unroll
                for (int hh = 0; hh < (GAB_KO_MAIN_MERGE ? 1 : NF); hh++) {
                    const int32_t b2 = L.part_best[par ^ 1][hh][lane], g2 = L.part_g[par ^ 1][hh][lane];
                    if (b2 > fbest || (b2 == fbest && g2 > fg)) { fbest = b2; fg = g2; }
                }
#endif
                const int32_t initkey = (qsa << 7) | 127;
                int32_t key = fg >= 0 ? max(initkey, fbest << 7) : initkey;      // far: code 0 (loses a tie against anything newer)
                const int4 *gn = &L.G4[par ^ 1][0][0][lane];
                const int4 *gb = &L.G4[par ^ 1][1][0][lane];
#if GAB_TAB_NEAR_HELP
                constexpr int kHelpM = FC ? GAB_TAB_NEAR_HELP_FC : GAB_TAB_NEAR_HELP_CH;
                if (kHelpM && t > 0) {
                    near_target += kHelpM;
                    while (__hip_atomic_load(&L.near_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < near_target) __builtin_amdgcn_s_sleep(1);
                    asm volatile("" ::: "memory");
                    key = max(key, __hip_atomic_load(&L.near_key[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    L.near_key[lane] = kTabNone;
                }
                if (!kHelpM && t > 0) {
#else
                if (t > 0 && !GAB_KO_MAIN_NEAR) {
#endif
#if GAB_TAB_NEAR_LDS
                    // (the previous block's scores come back from LDS as broadcasts, four per 16-byte read, instead of 64 v_readlane:
                    // the main wave is the phase's critical path, 93 % busy -- GAB_CHAIN_TRACE)
                    const int4 *pk4 = reinterpret_cast<const int4 *>(L.pk);
#pragma unroll
                    for (int g4 = 0; g4 < 16; g4++) {        // the previous block: no dependence between the steps
                        const int4 g = gn[(size_t)g4 * 64], pv = pk4[g4];
                        key = max(max(key, g.x + pv.x), max(g.y + pv.y, max(g.z + pv.z, g.w + pv.w)));
                    }
#else
                    const int32_t pkey = pbest << 7;
#pragma unroll
                    for (int g4 = 0; g4 < 16; g4++) {        // the previous block: no dependence between the steps
                        const int4 g = gn[(size_t)g4 * 64];
                        key = max(key, g.x + __builtin_amdgcn_readlane(pkey, 4 * g4));
                        key = max(key, g.y + __builtin_amdgcn_readlane(pkey, 4 * g4 + 1));
                        key = max(key, g.z + __builtin_amdgcn_readlane(pkey, 4 * g4 + 2));
                        key = max(key, g.w + __builtin_amdgcn_readlane(pkey, 4 * g4 + 3));
                    }
#endif
                }
                if (!FC && L.patch_n)
                    for (int q = 0; q < L.patch_n; q++) if (L.patch_blk[q] == t && L.patch_lane[q] == lane) key = L.patch_score[q] << 7;
                // the block itself, in units of four anchors closed by the workers (see there): the kernel's only chain of dependent
                // steps is four readlanes (the unit's scores before the unit), four adds, two three-way maxima -- per FOUR anchors
#pragma unroll
                for (int g4 = 0; g4 < (GAB_KO_MAIN_BLOCK ? 0 : 16); g4++) {
                    const int4 g = gb[(size_t)g4 * 64];
                    const int32_t x0 = __builtin_amdgcn_readlane(key, 4 * g4) & ~127, x1 = __builtin_amdgcn_readlane(key, 4 * g4 + 1) & ~127,
                                  x2 = __builtin_amdgcn_readlane(key, 4 * g4 + 2) & ~127, x3 = __builtin_amdgcn_readlane(key, 4 * g4 + 3) & ~127;
                    key = max(max(key, g.x + x0), max(g.y + x1, max(g.z + x2, g.w + x3)));
                }
                const int32_t best = key >> 7;
                if (mine) {
                    S[i0 + lane] = best;
                    if (host_score) host_score[w.hoff + i0 + lane] = best;       // gab_chain_run_device_through: written through to the caller's page-locked array
                    const int ri = (i0 + lane) & (kTabRing - 1);
                    L.ring[ri] = best - bias;
                    if (ri < 16) L.ring[kTabRing + ri] = best - bias;
                }
                L.res_key[t % 3][lane] = (!mine || key == initkey) ? -1 : key;     // -1: no predecessor improved on q_span
                L.res_fg[t % 3][lane] = fg;
                pbest = best;
#if GAB_TAB_NEAR_LDS
                L.pk[lane] = best << 7;
#endif
            }
        } else if constexpr (ROLE == 1) {
            // ------------------------------------------------ resolver: parents (chain: and the certificate), two blocks behind
#ifdef GAB_KO_TAB_RES
            if (false) {
#else
            if (t - 2 >= min_blk) {
#endif
                const int r = t - 2;
                const int i0 = r * 64;
                const int nb = n - i0 < 64 ? n - i0 : 64;
                const bool mine = lane < nb;
                const int nfar = r_desc.ng - (r > 0 ? 8 : 4);
                const int32_t rk = L.res_key[r % 3][lane], fg = L.res_fg[r % 3][lane];
                const bool none = rk < 0;
                const int code = rk & 127;
                const int32_t best = rk >> 7;
                int32_t parent = -1;
                if (!none && code >= 65) parent = i0 + code - 65;
                else if (!none && code >= 1) parent = i0 - 64 + code - 1;
                const bool pl = !FC && L.patch_n && patched(r);
                const bool far = mine && !none && code == 0 && !pl;
                if (mine) evals += (unsigned long long)(i0 + lane - r_st);
                int32_t risk = 0;
                if (__ballot(far)) {
                    // the newest row of group fg that attains the score (every lane its own group: gathered a phase ago)
                    const int j0 = r_desc.jrow0 + 16 * fg;
                    const uint32_t wd[4] = {r_gw.x, r_gw.y, r_gw.z, r_gw.w};
                    int kstar = -1;
#pragma unroll
                    for (int k = 15; k >= 0; k--) {
                        const int32_t bb = (int32_t)(wd[k >> 2] >> (8 * (k & 3)) & 0xffu);
                        const int32_t v = bb + L.ring[(j0 + k) & (kTabRing - 1)];
                        if (far && bb != 0 && v == best && kstar < 0) kstar = k;
                    }
                    if (far) parent = j0 + kstar;            // (kstar >= 0: the group's maximum is attained in it)
                    if (!FC && !GAB_KO_CERT_FAR) {
                        // unfiltered predecessors newer than the argmax among the far ones: the rest of its group, then every later group
                        if (far) {
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const int sh = kstar + 1 - 4 * q;          // rows of this word newer than kstar: from byte max(sh, 0) on
                                const uint32_t m = sh <= 0 ? wd[q] : sh >= 4 ? 0u : (wd[q] >> (8 * sh)) << (8 * sh);
                                risk += ctab_nonzero_bytes(m);
                            }
                        }
                        // (a far argmax is rare -- the best predecessor of an anchor on a chain is a few anchors back -- so these
                        // counts are not kept by the workers: the later groups are read again here, four requests at a time)
                        for (int g0 = 1; ; g0 += 4) {
                            const bool more = far && fg + g0 < nfar && risk <= kMaxSkip;
                            if (!__ballot(more)) break;
                            uint4 v4[4];
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                v4[q] = make_uint4(0, 0, 0, 0);
                                if (more && fg + g0 + q < nfar) v4[q] = T8[(r_desc.grp + fg + g0 + q) * 64 + lane];
                            }
#pragma unroll
                            for (int q = 0; q < 4; q++)
                                risk += ctab_nonzero_bytes(v4[q].x) + ctab_nonzero_bytes(v4[q].y) + ctab_nonzero_bytes(v4[q].z) + ctab_nonzero_bytes(v4[q].w);
                        }
                    }
                }
                if (!FC && !GAB_KO_CERT_NEAR) {
                    // ... and among the near / in-block pairs: pair numbers code .. 127 (pair code - 1 is the argmax; all of them for a far one)
#pragma unroll
                    for (int wd2 = 0; wd2 < 4; wd2++) {
                        const uint32_t bits = (uint32_t)L.okh[r & 3][2 * wd2][lane] | ((uint32_t)L.okh[r & 3][2 * wd2 + 1][lane] << 16);
                        const uint32_t m = code <= 32 * wd2 ? ~0u : code >= 32 * wd2 + 32 ? 0u : (~0u << (code - 32 * wd2));
                        risk += __popc(bits & m);
                    }
                }
                if (pl) for (int q = 0; q < L.patch_n; q++) if (L.patch_blk[q] == r && L.patch_lane[q] == lane) parent = L.patch_parent[q];
                if (mine) { P[i0 + lane] = parent; if (host_parent) host_parent[w.hoff + i0 + lane] = parent; }
                if (!FC && !GAB_KO_CERT_NEAR) {
                    unsigned long long miss = __ballot(mine && !none && !pl && risk > kMaxSkip);
                    if (miss) {
                        // max_skip may have cut these anchors' scans short (the certificate of chain_hw_kernel does not hold): the
                        // reference's own scan decides, anchor by anchor in order, on the scores and parents stored so far.  It
                        // nearly always finds what the plain maximum found (the early exit rarely changes a result); if not,
                        // younger anchors have already used the wrong score and the call goes back to the kernels of chain.hip.
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        // (a call in which this happens again and again -- dense anchors, dozens of unfiltered predecessors each -- is
                        // not worth a wave-wide scan per anchor here: after kTabMaxRescans it goes back as well)
                        n_rescans += __popcll(miss);
                        if (n_rescans > kTabMaxRescans) {
                            if (lane == 0) { L.stop[t & 1] = 1; __hip_atomic_store(&bail[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                            miss = 0;
                        }
                        while (miss) {
                            const int b = __builtin_ctzll(miss);
                            miss &= miss - 1;
                            int32_t eb, ej;
                            chain_exact_global(X, Y, S, P, GM, i0 + b, __builtin_amdgcn_readlane(r_st, b), w.max_dist_x, w.max_dist_y, w.bw, w.n_segs > 1,
                                               (double)w.avg_qspan, eb, ej, evals_exact);
                            if (eb != __builtin_amdgcn_readlane(best, b) || ej != __builtin_amdgcn_readlane(parent, b)) {
                                // max_skip did change this anchor's result, and younger anchors have used the other score: the anchor is
                                // PATCHED (score and parent of the reference's scan; nothing is folded into it any more) and the call starts
                                // again from this block -- everything before it is final.  More than kTabMaxPatch such anchors: chain.hip.
                                if (lane == 0) {
                                    const int pn = L.patch_n;
                                    if (pn < kTabMaxPatch) {
                                        L.patch_blk[pn] = r; L.patch_lane[pn] = b; L.patch_score[pn] = eb; L.patch_parent[pn] = ej;
                                        L.patch_n = pn + 1; L.restart_blk = r; L.stop[t & 1] = 2;
                                    } else { L.stop[t & 1] = 1; __hip_atomic_store(&bail[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                                }
                                break;
                            }
                        }
                    }
                }
            }
            if (t - 1 >= min_blk && t - 1 < nblocks) {
                // request what block t - 1 needs: its descriptor, the far lanes' group, the window starts
                const int r = t - 1;
                Desc tb;
                if (FC) tb = desc_of(r); else tb = desc_ring(r);
                r_desc = tb;
                const int32_t rk = L.res_key[r % 3][lane], fg = L.res_fg[r % 3][lane];
                const bool mine = r * 64 + lane < n;
                r_gw = make_uint4(0, 0, 0, 0);
                if (mine && rk >= 0 && (rk & 127) == 0) r_gw = T8[(tb.grp + fg) * 64 + lane];
                r_st = mine ? ST[r * 64 + lane] : 0;
            }
            if (!FC && t >= 0 && (t & 63) == 0) {            // the descriptors of blocks t + 128 .. t + 191 (their slots held t - 128 .. t - 65)
                const int b = t + 128 + lane;
                if (b < nblocks) { const TabBlock tb = B[b]; L.desc[b & 255] = Desc{tb.grp, tb.jrow0, tb.ng}; }
            }
        } else if (t + 1 < nblocks) {
            // ------------------------------------------------ workers: block t + 1 from the registers filled a phase ago
            const int kb = t + 1;
            const Desc d = d_cur;
            const int nfar = d.ng - (kb > 0 ? 8 : 4);
            const uint4 cg[2] = {pg[0], pg[1]};
            uint4 cf[kTabF];
#pragma unroll
            for (int q = 0; q < kTabF; q++) cf[q] = pf[q];
            d_cur = d_nxt;
            if (kb + 1 < nblocks) prefetch(d_cur, kb + 1);   // block t + 2, used in the next phase
            d_nxt = desc_of(kb + 2);                         // ... and the descriptor after that
            // the keys of the 128 near / in-block rows: unit u = 16 rows; 0 .. 3 the previous block, 4 .. 7 the block itself
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int u = unit_of(q);
#ifdef GAB_KO_TAB_G
                if (false) {
#else
                if (u < 8) {
#endif
                    const bool nearu = u < 4;
                    const bool pl = !FC && L.patch_n && patched(kb);       // nothing is folded into a patched anchor
                    const uint32_t wd[4] = {pl ? 0u : cg[q].x, pl ? 0u : cg[q].y, pl ? 0u : cg[q].z, pl ? 0u : cg[q].w};
                    uint32_t bits = 0;
                    int4 *dst = &L.G4[par][nearu ? 0 : 1][(u & 3) * 4][lane];
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) {
                        int32_t gv[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int p = (u & 3) * 16 + 4 * q4 + k;
                            const int32_t bb = (int32_t)(wd[q4] >> (8 * k) & 0xffu);
                            const int code = nearu ? p + 1 : 65 + p;
                            gv[k] = bb ? (int32_t)(((uint32_t)(bb - bias) << 7) | (uint32_t)code) : (nearu ? kTabNone : kTabSoftNone);
                            bits |= bb ? (1u << (4 * q4 + k)) : 0u;
                        }
                        if (!nearu) {
                            // The block's own rows in units of four, CLOSED over the unit: W_k[a] = the best way from row p_k to anchor a
                            // through later rows of the unit, as a key of a's direct predecessor (the code rides along; the hops inside
                            // the unit add their score-only part).  None of this involves a score, so it is done here, ahead of time, and
                            // the main wave's chain of dependent steps is one step per FOUR anchors: with x_k the scores of the unit's four
                            // anchors before the unit, every anchor a takes max_k (x_k + W_k[a]) -- for a inside the unit too (rows not
                            // older than a are filtered).  A filtered entry is kTabSoftNone here (sums of four stay in range and below
                            // every real value) and INT_MIN in the table.
                            const int pl = (u & 3) * 16 + 4 * q4;                      // lane of the unit's first anchor
                            const int32_t s01 = __builtin_amdgcn_readlane(gv[0], pl + 1) & ~127, s02 = __builtin_amdgcn_readlane(gv[0], pl + 2) & ~127,
                                          s03 = __builtin_amdgcn_readlane(gv[0], pl + 3) & ~127, s12 = __builtin_amdgcn_readlane(gv[1], pl + 2) & ~127,
                                          s13 = __builtin_amdgcn_readlane(gv[1], pl + 3) & ~127, s23 = __builtin_amdgcn_readlane(gv[2], pl + 3) & ~127;
                            const int32_t w3 = gv[3];
                            const int32_t w2 = max(gv[2], s23 + w3);
                            const int32_t w1 = max(max(gv[1], s12 + w2), s13 + w3);
                            const int32_t w0 = max(max(gv[0], s01 + w1), max(s02 + w2, s03 + w3));
                            gv[0] = w0 < kTabSoftLim ? kTabNone : w0; gv[1] = w1 < kTabSoftLim ? kTabNone : w1;
                            gv[2] = w2 < kTabSoftLim ? kTabNone : w2; gv[3] = w3 < kTabSoftLim ? kTabNone : w3;
                        }
                        dst[(size_t)q4 * 64] = make_int4(gv[0], gv[1], gv[2], gv[3]);
                    }
                    if (!FC && !GAB_KO_OKH) L.okh[kb & 3][u][lane] = (uint16_t)bits;
                }
            }
            // the far groups (scores final since block t - 1), dealt round-robin
            int32_t best = kTabNegH, bg = -1;
            auto far_group = [&](const uint4 wv, int fgi) {
                const int j0 = d.jrow0 + 16 * fgi;
                const uint32_t wd[4] = {wv.x, wv.y, wv.z, wv.w};
                int32_t sb[16];
                {
                    const int32_t *rp = &L.ring[j0 & (kTabRing - 1)];        // (sixteen consecutive entries: the ring's tail repeats its head)
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) __builtin_memcpy(&sb[4 * q4], rp + 4 * q4, 16);
                }
                int32_t gmax = kTabNegH;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int32_t bb = (int32_t)(wd[k >> 2] >> (8 * (k & 3)) & 0xffu);
                    const int32_t v = bb + sb[k];
                    gmax = bb ? max(gmax, v) : gmax;
                }
                const bool up = gmax >= best && gmax > kTabNegH;      // groups ascend: the newer group wins a tie
                best = up ? gmax : best; bg = up ? fgi : bg;
            };
#ifndef GAB_KO_TAB_FAR
#pragma unroll
            for (int q = 0; q < kTabF; q++) { const int fgi = fw + NF * q; if (fw >= 0 && fgi < nfar) far_group(cf[q], fgi); }
            if (fw >= 0) for (int fgi = fw + NF * kTabF; fgi < nfar; fgi += NF) far_group(T8[(d.grp + fgi) * 64 + lane], fgi);      // deep windows
#endif
#if GAB_TAB_MERGE_ATOMIC
            if (fw >= 0 && bg >= 0) atomicMax(&L.part64[par][lane], (long long)(((unsigned long long)(uint32_t)best << 32) | (uint32_t)bg));
#else
            if (fw >= 0) { L.part_best[par][fw][lane] = best; L.part_g[par][fw][lane] = bg; }
#endif
        }
        if (dbg) busy += clock64() - t_in;
        __syncthreads();
        const int st_code = L.stop[t & 1];
        if (st_code == 1) { stopped = true; break; }
        if (st_code == 2) {
            // start again from block R with the new patch: every wave brings its own state to "block R comes next"
            const int R = L.restart_blk;
            __syncthreads();                                 // (everybody has read the two words)
            if (threadIdx.x == 0) L.stop[t & 1] = 0;
            min_blk = R;
            if constexpr (ROLE == 0) {
                pbest = R > 0 ? L.ring[(R * 64 - 64 + lane) & (kTabRing - 1)] + bias : 0;
#if GAB_TAB_NEAR_LDS
                L.pk[lane] = pbest << 7;
#endif
                qs_next = R * 64 + lane < n ? (int32_t)(Y[R * 64 + lane] >> 32 & 0xff) : 0;
            } else if constexpr (ROLE >= 2) {
                d_cur = desc_of(R); d_nxt = desc_of(R + 1);
                prefetch(d_cur, R);
            }
#if GAB_TAB_MERGE_ATOMIC
            if (threadIdx.x < 128) (&L.part64[0][0])[threadIdx.x] = LLONG_MIN;      // what the abandoned phases left in the merge slots
            if (threadIdx.x < 64) L.near_key[threadIdx.x] = kTabNone;
            if (threadIdx.x == 0) L.near_cnt = 0;
            near_target = 0;
            __syncthreads();
#endif
            t = R - 2;                                       // (the loop makes it R - 1: the workers take block R, the main wave follows)
        }
    }
    };
    if (wave == 0) phases(std::integral_constant<int, 0>{});
    else if (wave == 1) phases(std::integral_constant<int, 1>{});
    else if (wave < 6) phases(std::integral_constant<int, 2>{});
    else if (wave < 10) phases(std::integral_constant<int, 3>{});
    else phases(std::integral_constant<int, 4>{});
    if (dbg && c == 0 && lane == 0) { dbg[2 * wave] = busy; dbg[2 * wave + 1] = clock64() - t_all; }
    if (dbg && threadIdx.x == 0) dbg[32 + 3 * (size_t)c + 2] = wall_clock64();
    if (wave == 1 && !stopped) {
        evals += evals_exact;
        for (int o = 32; o > 0; o >>= 1) evals += __shfl_xor(evals, o);
        if (lane == 0 && evals) atomicAdd(evals_out, evals);
        if (lane == 0 && n_rescans) atomicAdd(&ct->rescans, (uint32_t)n_rescans);
    }
}

}  // namespace

// =============================================================================== host side
// gab_chain_reserve_mode: the table for calls of up to max_anchors anchors in all, before a caller's timed region (the budget was
// fixed by the handle's first table-form run; 27 KB per block of 64 anchors is what ~430 rows per block need)
int chain_tab_prealloc(ChainTab *t, int64_t max_anchors, int64_t max_calls) {
    if (t->table_budget == 0) return GAB_OK;
    const size_t want = std::min<size_t>(t->table_budget, ((size_t)max_anchors / 64 + (size_t)max_calls) * 27 * 1024);
    int rc = GAB_OK;
    if (t->table.cap < want && (rc = t->table.reserve(want)) != GAB_OK) return rc;
    if ((rc = t->st.reserve(4 * (size_t)max_anchors + 256)) != GAB_OK) return rc;
    if ((rc = t->blocks.reserve(sizeof(TabBlock) * ((size_t)max_anchors / 64 + (size_t)max_calls) + 256)) != GAB_OK) return rc;
    return t->calls.reserve((sizeof(TabCall) + 8) * (size_t)max_calls + 2048);
}
int chain_tab_setup() {
    static std::once_flag once;
    static int rc = GAB_OK;
    std::call_once(once, [] {
        if (hipFuncSetAttribute((const void *)ctab_fold<GAB_CHAIN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TabLds)) != hipSuccess ||
            hipFuncSetAttribute((const void *)ctab_fold<GAB_FASTCHAIN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TabLds)) != hipSuccess ||
            hipFuncSetAttribute((const void *)ctab_fold<GAB_CHAIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TabLds)) != hipSuccess ||
            hipFuncSetAttribute((const void *)ctab_fold<GAB_FASTCHAIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TabLds)) != hipSuccess) {
            gab_set_error("chain table form: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); rc = GAB_EDEVICE;
        }
    });
    return rc;
}

void chain_tab_report(ChainTab *t, size_t nsplit) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_bail = up(sizeof(TabCall) * nsplit), o_ct = o_bail + up(4 * nsplit);
    std::vector<TabCall> hc(nsplit);
    std::vector<uint32_t> hb(nsplit);
    TabCounters ct;
    if (hipMemcpy(hc.data(), t->calls.p, sizeof(TabCall) * nsplit, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(hb.data(), t->calls.as<char>() + o_bail, 4 * nsplit, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&ct, t->calls.as<char>() + o_ct, sizeof ct, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
    std::vector<unsigned long long> dbgv(32 + 3 * nsplit, 0);
    if (t->dbg.p && hipMemcpy(dbgv.data(), t->dbg.p, 8 * dbgv.size(), hipMemcpyDeviceToHost) != hipSuccess) (void)hipGetLastError();
    const unsigned long long *dbg = dbgv.data();
    size_t nok = 0, nbail = 0;
    for (size_t k = 0; k < nsplit; k++) { nok += hc[k].ok != 0; nbail += hb[k] != 0; }
    fprintf(stderr, "[gab_chain table form] %zu calls: %zu eligible, %zu handed back (%u for room; %u exact re-scans in the others); %.1f MB of table needed, %.1f MB there; bias of call 0: %d\n",
            nsplit, nok, nbail, ct.no_room, ct.rescans, (double)ct.groups_needed * 1024 / 1e6, (double)t->table.cap / 1e6, nsplit ? hc[0].bias : 0);
    fprintf(stderr, "[gab_chain table form] blocks of calls 0, 1, 2 .. n-2, n-1: %d %d %d .. %d %d; handed back:", hc[0].nblk, nsplit > 1 ? hc[1].nblk : 0, nsplit > 2 ? hc[2].nblk : 0,
            nsplit > 1 ? hc[nsplit - 2].nblk : 0, hc[nsplit - 1].nblk);
    { int shown = 0; for (size_t k = 0; k < nsplit && shown < 12; k++) if (hb[k]) { fprintf(stderr, " #%zu(%d blocks, ok=%d, bias=%d)", k, hc[k].nblk, hc[k].ok, hc[k].bias); shown++; } }
    fprintf(stderr, "\n");
    fprintf(stderr, "[gab_chain table form] fold of call 0 (%d blocks): cycles outside the barrier / in the loop, per wave (main, resolver, workers):", hc[0].nblk);
    for (int k = 0; k < 16 && dbg[2 * k + 1]; k++) fprintf(stderr, " %llu/%llu", dbg[2 * k], dbg[2 * k + 1]);
    fprintf(stderr, "\n");
    {   // when the workgroups of the fold ran (100 MHz wall clock): start, anchors' geometry there, end -- relative to the first start
        unsigned long long t0 = ~0ull;
        for (size_t k = 0; k < nsplit; k++) if (dbg[32 + 3 * k]) t0 = std::min(t0, dbg[32 + 3 * k] >> 4);
        int per_xcc[16] = {};
        for (size_t k = 0; k < nsplit; k++) if (dbg[32 + 3 * k]) per_xcc[dbg[32 + 3 * k] & 15]++;
        fprintf(stderr, "[gab_chain table form] fold workgroups per XCD:");
        for (int x = 0; x < 8; x++) fprintf(stderr, " %d", per_xcc[x]);
        fprintf(stderr, "\n");
        {   // the calls that ended last
            std::vector<size_t> ord;
            for (size_t k = 0; k < nsplit; k++) if (dbg[32 + 3 * k] && dbg[32 + 3 * k + 2]) ord.push_back(k);
            std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return dbg[32 + 3 * a + 2] > dbg[32 + 3 * b + 2]; });
            for (size_t q = 0; q < ord.size() && q < 6; q++) {
                const size_t k = ord[q];
                fprintf(stderr, "   ended last: call %zu (%d blocks): started %.3f ms, done %.3f ms\n", k, hc[k].nblk, ((dbg[32 + 3 * k] >> 4) - t0) * 1e-5, (dbg[32 + 3 * k + 2] - t0) * 1e-5);
            }
        }
        for (size_t k = 0; k < nsplit; k = k < 8 ? k + 1 : k * 2) {
            if (!dbg[32 + 3 * k]) continue;
            fprintf(stderr, "   call %zu (%d blocks, XCD %d): started %.3f ms, geometry there %.3f ms, done %.3f ms\n", k, hc[k].nblk, (int)(dbg[32 + 3 * k] & 15),
                    ((dbg[32 + 3 * k] >> 4) - t0) * 1e-5, dbg[32 + 3 * k + 1] ? (dbg[32 + 3 * k + 1] - t0) * 1e-5 : -1., dbg[32 + 3 * k + 2] ? (dbg[32 + 3 * k + 2] - t0) * 1e-5 : -1.);
        }
    }
}

int chain_tab_run(ChainTab *t, const gab_tuning &tun, int mode, hipStream_t s, const ChainWork *d_work, const ChainWork *h_work, size_t nsplit, int64_t total_anchors,
                  const uint64_t *d_x, const uint64_t *d_y, int32_t *d_score, int32_t *d_parent, int32_t *d_gm, unsigned long long *d_evals, uint32_t **d_bail,
                  int32_t *host_score, int32_t *host_parent) {
    int rc = chain_tab_setup();
    if (rc) return rc;
    // host side of the call table: blocks, gap-table offsets
    t->host_calls.resize(sizeof(TabCall) * nsplit);
    TabCall *const hc = reinterpret_cast<TabCall *>(t->host_calls.data());
    int64_t nblocks = 0, gt = 0;
    bool any_mseg = false;
    for (size_t k = 0; k < nsplit; k++) {
        TabCall &c = hc[k];
        memset(&c, 0, sizeof c);
        c.blk0 = nblocks; c.nblk = (int32_t)((h_work[k].n + 63) / 64);
        nblocks += c.nblk;
        c.gt_off = (int32_t)gt;
        const int bw = h_work[k].bw;
        gt += (bw >= 0 && bw <= kGapTab - 2) ? bw + 2 : 0;
        GAB_CHECK(gt < (1ll << 31), "gab_chain: gap tables of the table form out of range");
        any_mseg = any_mseg || h_work[k].n_segs > 1;
    }
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_bail = up(sizeof(TabCall) * nsplit), o_ct = o_bail + up(4 * nsplit);
    if ((rc = t->calls.reserve(o_ct + 512)) != GAB_OK) return rc;
    if ((rc = t->blocks.reserve(sizeof(TabBlock) * (size_t)nblocks + 256)) != GAB_OK) return rc;
    if ((rc = t->gtab.reserve(4 * (size_t)gt + 256)) != GAB_OK) return rc;
    if ((rc = t->st.reserve(4 * (size_t)total_anchors + 256)) != GAB_OK) return rc;
    // the table: 1 KB per 16 rows x 64 anchors.  How deep the windows are is only known on the device, so the first call sizes
    // it for ~430 rows per block (a 300-predecessor window + the two blocks) and a later call for what the last one needed;
    // calls that find no room keep the other kernels.  $GAB_CHAIN_TAB_MB bounds it (default: half of the free memory, 96 GB at most).
    if (t->table_budget == 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)8 << 30; }
        size_t budget = std::min<size_t>(free_b / 2, (size_t)96 << 30);
        if (tun.chain_tab_mb > 0) budget = (size_t)tun.chain_tab_mb << 20;      // GAB_CHAIN_TAB_MB
        t->table_budget = budget;
    }
    const size_t est = std::max<size_t>((size_t)nblocks * 27 * 1024, (size_t)1 << 20);
    const size_t want = std::min(t->table_budget, std::max(est, t->table.cap ? std::min(t->table.cap, t->table_budget) : (size_t)0));
    if (t->table.cap < want && (rc = t->table.reserve(want)) != GAB_OK) return rc;
    const long long budget_groups = (long long)(std::min(t->table.cap, t->table_budget) / 1024);
    char *cb = t->calls.as<char>();
    TabCall *d_calls = (TabCall *)cb;
    uint32_t *bail = (uint32_t *)(cb + o_bail);
    TabCounters *d_ct = (TabCounters *)(cb + o_ct);
    TabBlock *d_blocks = t->blocks.as<TabBlock>();
    int32_t *d_gtab = t->gtab.as<int32_t>(), *d_st = t->st.as<int32_t>();
    uint4 *d_T8 = t->table.as<uint4>();
    *d_bail = bail;
    GAB_HIP(hipMemcpyAsync(d_calls, hc, sizeof(TabCall) * nsplit, hipMemcpyHostToDevice, s));      // (t->host_calls: alive until the handle's next run)
    GAB_HIP(hipMemsetAsync(d_ct, 0, 512, s));
    unsigned long long *d_dbg = nullptr;       // GAB_CHAIN_TRACE: per-wave cycle counts of call 0's fold, then start / ready / end of every call's workgroup
    if (tun.chain_trace) {
        if ((rc = t->dbg.reserve(8 * (32 + 3 * nsplit))) != GAB_OK) return rc;
        d_dbg = t->dbg.as<unsigned long long>();
        GAB_HIP(hipMemsetAsync(d_dbg, 0, 8 * (32 + 3 * nsplit), s));
    }
    const unsigned nc = (unsigned)nsplit;
    if (mode == GAB_CHAIN) hipLaunchKernelGGL(ctab_prep<GAB_CHAIN>, dim3(nc), dim3(256), 0, s, d_work, d_calls, bail, d_x, d_y, d_gtab);
    else hipLaunchKernelGGL(ctab_prep<GAB_FASTCHAIN>, dim3(nc), dim3(256), 0, s, d_work, d_calls, bail, d_x, d_y, d_gtab);
    hipLaunchKernelGGL(ctab_blocks_init, dim3(nc), dim3(256), 0, s, (const TabCall *)d_calls, d_blocks);
    const unsigned g4 = (unsigned)((nblocks + 3) / 4);
    if (mode == GAB_CHAIN) hipLaunchKernelGGL(ctab_st<GAB_CHAIN>, dim3(g4), dim3(256), 0, s, d_work, d_calls, d_blocks, nblocks, d_x, d_st);
    else hipLaunchKernelGGL(ctab_st<GAB_FASTCHAIN>, dim3(g4), dim3(256), 0, s, d_work, d_calls, d_blocks, nblocks, d_x, d_st);
    hipLaunchKernelGGL(ctab_place, dim3(1), dim3(1024), 0, s, d_calls, bail, (int)nsplit, budget_groups, d_ct);
    hipLaunchKernelGGL(ctab_block_offsets, dim3(nc), dim3(256), 0, s, (const TabCall *)d_calls, (const uint32_t *)bail, d_blocks);
    // Geometry, then fold, on one stream.  Measured and dropped (r04): the fold BESIDE the geometry -- a second stream, the
    // geometry's stores written through, a call's workgroup waiting in the kernel until its blocks were counted done.  The folds
    // of the longest calls did start after microseconds, and the two kernels together took as long as one after the other (chain
    // shard 0/8: 5.98 against 6.01 ms): a wave of the geometry sharing a SIMD with a fold's main wave stretches every step of its
    // chain of dependent instructions (s_setprio changes nothing: the pipeline is not pre-empted), and the written-through
    // stores cost the geometry a third of its speed.  Four chunks on four streams (events): the streams share hardware queues.
    hipStream_t sf = s;
    const unsigned nbk = (unsigned)nblocks;
    if (mode == GAB_FASTCHAIN)
        hipLaunchKernelGGL((ctab_geo<GAB_FASTCHAIN, false>), dim3(nbk), dim3(256), 0, s, d_work, (const TabCall *)d_calls, (const TabBlock *)d_blocks, bail,
                           (const int32_t *)d_gtab, (const int32_t *)d_st, d_x, d_y, d_T8);
    else if (any_mseg)
        hipLaunchKernelGGL((ctab_geo<GAB_CHAIN, true>), dim3(nbk), dim3(256), 0, s, d_work, (const TabCall *)d_calls, (const TabBlock *)d_blocks, bail,
                           (const int32_t *)d_gtab, (const int32_t *)d_st, d_x, d_y, d_T8);
    else
        hipLaunchKernelGGL((ctab_geo<GAB_CHAIN, false>), dim3(nbk), dim3(256), 0, s, d_work, (const TabCall *)d_calls, (const TabBlock *)d_blocks, bail,
                           (const int32_t *)d_gtab, (const int32_t *)d_st, d_x, d_y, d_T8);
    if (mode == GAB_CHAIN)
        hipLaunchKernelGGL((d_dbg ? ctab_fold<GAB_CHAIN, true> : ctab_fold<GAB_CHAIN, false>), dim3(nc), dim3(64 * (2 + kTabW)), sizeof(TabLds), sf, d_work, (const TabCall *)d_calls, (const TabBlock *)d_blocks, bail,
                           (const uint4 *)d_T8, (const int32_t *)d_st, d_x, d_y, d_score, d_parent, d_gm, d_evals, d_ct, d_dbg, host_score, host_parent);
    else
        hipLaunchKernelGGL((d_dbg ? ctab_fold<GAB_FASTCHAIN, true> : ctab_fold<GAB_FASTCHAIN, false>), dim3(nc), dim3(64 * (2 + kTabW)), sizeof(TabLds), sf, d_work, (const TabCall *)d_calls, (const TabBlock *)d_blocks, bail,
                           (const uint4 *)d_T8, (const int32_t *)d_st, d_x, d_y, d_score, d_parent, d_gm, d_evals, d_ct, d_dbg, host_score, host_parent);
    GAB_HIP(hipGetLastError());
    return GAB_OK;
}
