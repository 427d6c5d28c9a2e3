// chain / fast-chain -- minimap2 seed-chaining DP on gfx950.
//
// Semantics
//   mode CHAIN     : chain_dp of /root/reference/benchmarks/chain/src/host_kernel.cpp:30-94
//                    (64-bit anchors, segment ids, max_iter = 5000, max_skip = 25 via targets[]).
//   mode FASTCHAIN : chain_dp of /root/reference/benchmarks/fast-chain/src/host_kernel.cpp as its
//                    AVX2 (:408-683) and AVX-512 (:175-407) builds compute it: 32-bit truncated
//                    coordinates, no max_skip, fp32 floor gap cost when the window holds more
//                    than six predecessors, double otherwise.
//
// Mapping.  The scores along i are a true recurrence (score[i] needs score[i-1]), so a call is
// walked sequentially by ONE workgroup of four waves; the parallelism is (a) the 256 predecessors of
// a window super-chunk, one per lane, and (b) thousands of independent calls, longest call first.
// The latency of that sequential walk is the bound, so everything on the i-1 -> i dependence stays
// on the CU: the last 1024 anchors (x, y, score, parent) live in an LDS ring.  Per anchor i:
//   * the window start `st` is advanced with one 64-wide compare + ballot against a cached block of
//     x values instead of the reference's scalar while-loop;
//   * the window [st, i-1] is swept in descending 256-anchor super-chunks read from the LDS ring
//     (anchors older than the ring come from global memory with agent-scope loads);
//   * FASTCHAIN: per-wave DPP max-reduction, the four maxima combined through LDS (one barrier),
//     ties -> larger j;
//   * CHAIN: the sequential max_skip logic is reproduced exactly in three parallel steps
//     (SURVEY.md App. B8): every unfiltered lane first scatters its mark targets[parent[j]] = i
//     into a 16-bit LDS ring and publishes its chunk maximum (barrier), then reads its own mark and
//     derives the "sc > max_f" improvement flag from an exclusive prefix-max (DPP scan + the earlier
//     chunks' maxima) and publishes the ballot masks (barrier); finally every wave walks the masks
//     of the four chunks in order for the saturating n_skip counter and its > 25 break.  Marks
//     scattered by lanes past the break point are harmless because a mark value i is only ever
//     compared with the current i.
//
// Roofline: 24 B of HBM traffic per anchor (16 B in, 8 B out) against ~130-200 predecessor
// evaluations per anchor: latency/VALU bound by construction; the window re-reads are served by
// L1/L2, not HBM.
#include "gab_internal.h"
#include <algorithm>
#include <new>
#include <vector>
#include <string.h>

namespace {

constexpr int kMaxIter = 5000;
constexpr int kMaxSkip = 25;
constexpr int kMarkRing = 1024;           // LDS ring of mark tags for the newest anchors; older marks go to global memory

struct ChainWork {                        // one call, device-side descriptor
    int64_t off, n;
    float avg_qspan;
    int32_t max_dist_x, max_dist_y, bw, n_segs, pad;
};

#define GAB_DPP(old, src, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rmask), 0xf, false)

// inclusive max-scan across the 64 lanes (lane order), identity INT_MIN
__device__ __forceinline__ int wave_incl_max(int v) {
    const int id = (int)0x80000000;
    v = max(v, GAB_DPP(id, v, 0x111, 0xf));   // row_shr:1
    v = max(v, GAB_DPP(id, v, 0x112, 0xf));   // row_shr:2
    v = max(v, GAB_DPP(id, v, 0x114, 0xf));   // row_shr:4
    v = max(v, GAB_DPP(id, v, 0x118, 0xf));   // row_shr:8
    v = max(v, GAB_DPP(id, v, 0x142, 0xa));   // row_bcast:15 -> rows 1,3
    v = max(v, GAB_DPP(id, v, 0x143, 0xc));   // row_bcast:31 -> rows 2,3
    return v;
}
// inclusive add-scan / min-scan, same DPP network
__device__ __forceinline__ int wave_incl_sum(int v) {
    v += GAB_DPP(0, v, 0x111, 0xf);
    v += GAB_DPP(0, v, 0x112, 0xf);
    v += GAB_DPP(0, v, 0x114, 0xf);
    v += GAB_DPP(0, v, 0x118, 0xf);
    v += GAB_DPP(0, v, 0x142, 0xa);
    v += GAB_DPP(0, v, 0x143, 0xc);
    return v;
}
__device__ __forceinline__ int wave_incl_min(int v) {
    const int id = 0x7fffffff;
    v = min(v, GAB_DPP(id, v, 0x111, 0xf));
    v = min(v, GAB_DPP(id, v, 0x112, 0xf));
    v = min(v, GAB_DPP(id, v, 0x114, 0xf));
    v = min(v, GAB_DPP(id, v, 0x118, 0xf));
    v = min(v, GAB_DPP(id, v, 0x142, 0xa));
    v = min(v, GAB_DPP(id, v, 0x143, 0xc));
    return v;
}
__device__ __forceinline__ int wave_shr1(int v, int fill) { return GAB_DPP(fill, v, 0x138, 0xf); }
__device__ __forceinline__ uint64_t wave_shr1_u64(uint64_t v) {
    uint32_t lo = (uint32_t)wave_shr1((int)(uint32_t)v, 0), hi = (uint32_t)wave_shr1((int)(uint32_t)(v >> 32), 0);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ int ilog2_u32(uint32_t v) { return 31 - __clz((int)v); }

#ifndef GAB_CHAIN_WAVES
#define GAB_CHAIN_WAVES 4
#endif
constexpr int kWaves = GAB_CHAIN_WAVES;   // waves per call (one workgroup)
constexpr int kThreads = kWaves * 64;
constexpr int kRing = 512;                // anchors kept in the LDS window ring
constexpr int kRingSafe = kRing - 8;      // entries younger than this are read from the ring

// One workgroup (4 waves) walks one call.  Per anchor i the predecessor window [st, i-1] is swept in
// "super-chunks" of 256 predecessors, wave w taking chunk 4*s + w.  The last kRing anchors (x, y, seg id,
// score, parent) live in an LDS ring, so the RAW dependence score[i-1] -> score[i] never leaves the CU;
// predecessors older than the ring (windows > ~1000 anchors) are read back from global memory.
// Workgroup barrier that waits for LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for
// the global store of score[i] / parent[i] (a ~2 us write acknowledgement) on every anchor of the sequential walk.
// Nothing crossing these barriers goes through global memory: the window ring, the marks and the publish slots are
// all LDS.  (Global loads are waited for by the compiler at their first use, as always.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool FAST>
__global__ __launch_bounds__(kThreads) void chain_kernel(const ChainWork *__restrict__ work,
                                                         const uint64_t *__restrict__ xs,
                                                         const uint64_t *__restrict__ ys,
                                                         int32_t *score_out, int32_t *parent_out,
                                                         int32_t *gmarks_all, unsigned long long *evals_out) {
    __shared__ uint64_t ring_x[kRing];
    __shared__ uint32_t ring_y[kRing];
    __shared__ int32_t ring_sc[kRing];
    __shared__ int32_t ring_par[kRing];
    __shared__ uint8_t ring_sid[FAST ? 1 : kRing];
    __shared__ uint16_t marks[FAST ? 1 : kMarkRing];
    __shared__ uint64_t stage_x[kThreads], stage_y[kThreads];
    __shared__ int32_t pub_max[2][kWaves], pub_j[2][kWaves];          // FAST: per-wave maxima (double-buffered)
    // CHAIN: per-chunk maxima, scores and ballot masks; double-buffered by super-chunk parity so that a wave
    // already in the next super-chunk's phase A never overwrites what a slower wave still reads in phase C
    __shared__ int32_t pub_cmax[2][kWaves];
    __shared__ int32_t pub_sc[2][kWaves][64];
    __shared__ unsigned long long pub_imp[2][kWaves], pub_hit[2][kWaves], pub_valid[2][kWaves];

    const ChainWork w = work[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t *X = xs + w.off, *Y = ys + w.off;
    int32_t *S = score_out + w.off, *P = parent_out + w.off;
    int32_t *GM = FAST ? nullptr : gmarks_all + w.off;     // targets[] of the reference, used only beyond the LDS mark ring
    const int64_t n = w.n;
    const int32_t mdx = w.max_dist_x, mdy = w.max_dist_y, bw = w.bw;
    const uint64_t mdx64 = (uint64_t)(int64_t)mdx;
    const double avg_d = (double)w.avg_qspan;
    const float k32 = (float)(0.01 * (double)w.avg_qspan);
    const bool multi_seg = w.n_segs > 1;

    // cached block of x for the window-start search (every wave keeps its own copy: no barrier needed)
    int64_t st = 0, sb = 0;
    uint64_t XS = (lane < n) ? X[lane] : 0;
    unsigned long long evals = 0;
    int pub_phase = 0;

    for (int64_t i = 0; i < n; i++) {
        if ((i & (kThreads - 1)) == 0) {
            // stage the next 256 anchors (coalesced) -- the previous block is no longer needed by anyone
            lds_barrier();
            // results leave the CU in coalesced blocks of 256 taken from the ring -- a store per anchor would put a
            // ~2 us write acknowledgement (s_waitcnt vmcnt) on the sequential path of every anchor
            if (i > 0) {
                const int64_t jo = i - kThreads + tid;
                S[jo] = ring_sc[jo & (kRing - 1)]; P[jo] = ring_par[jo & (kRing - 1)];
            }
            if (i + tid < n) { stage_x[tid] = X[i + tid]; stage_y[tid] = Y[i + tid]; }
            if (!FAST && (i & 0x7fff) == 0)
                for (int k = tid; k < kMarkRing; k += kThreads) marks[k] = 0;   // new tag epoch (see tag below)
            __syncthreads();     // full barrier (once per 256 anchors): the flushed results are acknowledged by L2
                                 // before any wave may read them back through the deep-window path
        }
        const uint64_t xi = stage_x[i & (kThreads - 1)], yi = stage_y[i & (kThreads - 1)];     // workgroup-uniform
        const int32_t qi = (int32_t)yi, q_span = (int32_t)(yi >> 32 & 0xff), sidi = (int32_t)(yi >> 48 & 0xff);
        // ---- window start (host_kernel.cpp:56-57 / fast :200-207)
        for (;;) {
            const int64_t cand = sb + lane;
            const bool far = FAST ? ((xi - XS) > mdx64) : (xi > XS + mdx64);
            const bool pass = cand < st || (cand < i && far);
            const unsigned long long m = __ballot(pass);
            if (m == ~0ull) {
                sb += 64; st = sb;
                XS = (sb + lane < n) ? X[sb + lane] : 0;
                continue;
            }
            st = sb + __builtin_ctzll(~m);
            break;
        }
        if (i - st > kMaxIter) st = i - kMaxIter;
        if (st - sb >= 64) { sb = st & ~63ll; XS = (sb + lane < n) ? X[sb + lane] : 0; }

        int32_t best = q_span, best_j = -1;
        const int64_t count = i - st;
        const bool wide = !((i - 1) - st <= 5);          // FAST only (:211/:440)
        int n_skip = 0;
        bool broke = false;
        const uint16_t tag = (uint16_t)(0x8000 | (i & 0x7fff));

        for (int64_t c0 = 0; c0 < count && !broke; c0 += kThreads) {
            const int64_t j = i - 1 - c0 - tid;           // wave w owns lanes [64w, 64w+64) of the super-chunk
            const bool valid = j >= st;
            uint64_t xj = 0; uint32_t yj = 0; int sidj = 0, scj = 0, parj = -1;
            if (valid) {
                if (i - j <= kRingSafe) {
                    const int r = (int)(j & (kRing - 1));
                    xj = ring_x[r]; yj = ring_y[r]; scj = ring_sc[r];
                    if (!FAST) { sidj = ring_sid[r]; parj = ring_par[r]; }
                } else {
                    // older than the LDS ring: L2-coherent loads (the values were stored by another wave of this CU)
                    xj = X[j];
                    const uint64_t yy = Y[j];
                    yj = (uint32_t)yy; sidj = (int)(yy >> 48 & 0xff);
                    scj = __hip_atomic_load(&S[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!FAST) parj = __hip_atomic_load(&P[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            bool ok = valid;
            int32_t sc = 0;
            if (FAST) {
                const int32_t ddr = (int32_t)((uint32_t)xi - (uint32_t)xj);
                const int32_t ddq = (int32_t)((uint32_t)qi - yj);
                const uint32_t diff = (uint32_t)ddr - (uint32_t)ddq;
                const int32_t dd = (int32_t)((int32_t)diff < 0 ? 0u - diff : diff);
                ok = ok && !(dd > bw || ddr == 0 || ddq <= 0 || ddq > mdy || ddq > mdx);
                const int32_t oc = min(min(ddr, ddq), q_span);
                const int32_t lg = dd ? ilog2_u32((uint32_t)dd) : 0;
                int32_t gc;
                if (wide) gc = (int32_t)floorf(__fmul_rn((float)dd, k32)) + (lg >> 1);
                else gc = (int32_t)__dmul_rn(__dmul_rn((double)dd, .01), avg_d) + (lg >> 1);
                sc = (int32_t)((uint32_t)scj + (uint32_t)oc - (uint32_t)gc);
                evals += valid ? 1 : 0;
                // per-wave max (ties -> larger j = lower lane), then combine the four waves through LDS
                const int v = ok ? sc : (int)0x80000000;
                const int mx = __builtin_amdgcn_readlane(wave_incl_max(v), 63);
                const unsigned long long who = __ballot(ok && sc == mx);
                if (lane == 0) {
                    pub_max[pub_phase][wave] = mx;
                    pub_j[pub_phase][wave] = who ? (int32_t)(i - 1 - c0 - 64 * wave - __builtin_ctzll(who)) : -1;
                }
                lds_barrier();
#pragma unroll
                for (int ww = 0; ww < kWaves; ww++) {       // wave order = descending j: strict > keeps the larger j
                    const int m2 = pub_max[pub_phase][ww];
                    if (m2 > best) { best = m2; best_j = pub_j[pub_phase][ww]; }
                }
                pub_phase ^= 1;
            } else {
                const int64_t dr = (int64_t)(xi - xj);
                const int32_t dq = qi - (int32_t)yj;
                const bool same = sidi == sidj;
                const int32_t dd = (int32_t)(dr > dq ? dr - dq : dq - dr);
                const bool skip = (same && dr == 0) || dq <= 0 || (same && dq > mdy) || dq > mdx || (same && dd > bw) ||
                                  (multi_seg && same && dr > mdy);
                ok = ok && !skip;
                const int32_t min_d = (int32_t)(dq < dr ? (int64_t)dq : dr);
                sc = min_d > q_span ? q_span : min_d;
                const int32_t lg = dd ? ilog2_u32((uint32_t)dd) : 0;
                const int32_t c_lin = (int32_t)__dmul_rn(__dmul_rn((double)dd, .01), avg_d);
                int32_t gap;
                if (!same) {
                    if (dr == 0) { ++sc; gap = 0; }
                    else gap = c_lin < lg ? c_lin : lg;
                } else gap = c_lin + (lg >> 1);
                sc -= (int32_t)(__dadd_rn((double)gap, .499));     // (int)((double)gap_cost * 1.0f + .499)
                sc += scj;
                // phase A: scatter marks, publish the chunk maximum and the scores.  Marks of parents inside the LDS
                // ring window go to LDS; a super-chunk that reaches further back (window > ~500 anchors) also uses the
                // global targets[] array, and then needs full barriers (global stores must be acknowledged).
                const bool deep = c0 + kThreads > kRingSafe;           // workgroup-uniform
                if (ok && parj >= 0 && parj >= st) {
                    if (i - parj <= kRingSafe) marks[parj & (kMarkRing - 1)] = tag;
                    else __hip_atomic_store(&GM[parj], (int32_t)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const int v = ok ? sc : (int)0x80000000;
                const int incl = wave_incl_max(v);
                if (lane == 63) pub_cmax[pub_phase][wave] = incl;
                pub_sc[pub_phase][wave][lane] = sc;
                if (deep) __syncthreads(); else lds_barrier();
                // phase B: marks of ALL chunks of this super-chunk are visible
                bool hit_raw = false;
                if (ok) {
                    if (i - j <= kRingSafe) hit_raw = marks[j & (kMarkRing - 1)] == tag;
                    else hit_raw = __hip_atomic_load(&GM[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int32_t)(i + 1);
                }
                int before = wave_shr1(incl, (int)0x80000000);
                before = max(before, best);
#pragma unroll
                for (int ww = 0; ww < kWaves; ww++) if (ww < wave) before = max(before, pub_cmax[pub_phase][ww]);
                const bool imp = ok && sc > before;
                const unsigned long long imp_m = __ballot(imp), hit_m = __ballot(hit_raw && !imp), valid_m = __ballot(valid);
                if (lane == 0) { pub_imp[pub_phase][wave] = imp_m; pub_hit[pub_phase][wave] = hit_m; pub_valid[pub_phase][wave] = valid_m; }
                lds_barrier();
                // phase C: every wave walks the four chunks in order (identical scalar work, no further barrier)
                unsigned long long visited = 0;
                for (int ww = 0; ww < kWaves && !broke; ww++) {
                    const unsigned long long im = pub_imp[pub_phase][ww], hm = pub_hit[pub_phase][ww], vm = pub_valid[pub_phase][ww];
                    int brk = 64;
                    if (hm == 0) {
                        n_skip -= __popcll(im);
                        n_skip = n_skip < 0 ? 0 : n_skip;
                    } else {
                        // n_skip is a counter reflected at 0: c_l = max(c_{l-1} + d_l, 0) with d = +1 on a hit, -1 on an
                        // improvement.  Closed form over the chunk: c_l = P_l - min(-c_in, min_{t<=l} P_t), P = prefix sums
                        // of d -- two DPP scans instead of a scalar walk over up to 64 events.
                        const int d = (int)((hm >> lane) & 1) - (int)((im >> lane) & 1);
                        const int pre = wave_incl_sum(d);
                        const int mn = wave_incl_min(pre);
                        const int cnt = pre - min(-n_skip, mn);
                        const unsigned long long over = __ballot(((hm >> lane) & 1) && cnt > kMaxSkip);
                        if (over) brk = __builtin_ctzll(over);
                        else n_skip = __builtin_amdgcn_readlane(cnt, 63);
                    }
                    const unsigned long long rec = im & (brk >= 64 ? ~0ull : ((1ull << brk) - 1));
                    if (rec) {
                        const int l = 63 - __builtin_clzll(rec);
                        best = pub_sc[pub_phase][ww][l];
                        best_j = (int32_t)(i - 1 - c0 - 64 * ww - l);
                    }
                    visited += (unsigned long long)__popcll(vm & (brk >= 64 ? ~0ull : ((2ull << brk) - 1)));
                    broke = brk < 64;
                }
                if (tid == 0) evals += visited;
                pub_phase ^= 1;
            }
        }
                // Every wave keeps the ring up to date by itself (lane 0 of each wave stores the same values), so a wave
        // only ever reads ring entries it has written: no barrier is needed to publish anchor i.  The barriers of the
        // super-chunks keep the four waves within one anchor of each other, which is what protects a slot from being
        // recycled (1024 anchors later) while a slower wave could still read it (it reads at most kRingSafe back);
        // an anchor with an empty window has no super-chunk, so it takes an explicit barrier.
        if (lane == 0) {
            const int r = (int)(i & (kRing - 1));
            ring_x[r] = xi; ring_y[r] = (uint32_t)yi; ring_sc[r] = best; ring_par[r] = best_j;
            if (!FAST) ring_sid[r] = (uint8_t)sidi;
        }
        if (count <= 0) lds_barrier();
    }
    lds_barrier();
    {   // flush the tail: anchors [n - rem, n) with rem = n mod 256 (or 256)
        const int64_t done = n > 0 ? ((n - 1) & ~(int64_t)(kThreads - 1)) : 0;
        const int64_t jo = done + tid;
        if (jo < n) { S[jo] = ring_sc[jo & (kRing - 1)]; P[jo] = ring_par[jo & (kRing - 1)]; }
    }
    if (FAST) { for (int o = 32; o > 0; o >>= 1) evals += __shfl_xor(evals, o); }
    if (lane == 0 && evals) atomicAdd(evals_out, evals);
}


// ---- fast-chain: block formulation, one main wave + three helper waves per call ------------------------------------
// Without max_skip the DP of an anchor is an order-independent maximum over its window (ties -> larger j), so the
// sequential dependence shrinks to "anchor i needs the final score of anchors i-1, i-2, ...".  Anchors are taken in
// blocks of 64 (lane a <-> anchor i0 + a) and every lane scores a BROADCAST predecessor against its own anchor:
// 64 evaluations per ~45 instructions with no reduction.  Per block t:
//   helper waves (run one block ahead, on block t+1 while the main wave is on block t):
//     * window starts st[a] with the reference's sequential-pointer semantics (ballot search), each helper for itself;
//     * "far" predecessors j <= i0 - 65 (final since block t-1): read 64 at a time with L2-coherent loads, broadcast
//       with v_readlane, chunks dealt round-robin to the helpers; partial (best, argbest) per anchor go to LDS;
//   main wave (the sequential path):
//     * combines the helpers' partial maxima, folds in the 64 "near" predecessors (the previous block, still in its
//       registers) and then the predecessors inside the block: anchor b is final once 0..b-1 are folded, is broadcast
//       and folded into the lanes a > b.  A newer predecessor wins a tie against an older one.
// One __syncthreads per block hands the results of block t to the helpers (global scores, acknowledged by L2) and the
// partial maxima of block t+1 to the main wave (LDS).
constexpr int kFcHelpers = 3;
__global__ __launch_bounds__(64 * (1 + kFcHelpers)) void fastchain_kernel(const ChainWork *__restrict__ work,
                                                                          const uint64_t *__restrict__ xs,
                                                                          const uint64_t *__restrict__ ys, int32_t *score_out,
                                                                          int32_t *parent_out, unsigned long long *evals_out) {
    __shared__ int32_t part_best[2][kFcHelpers][64], part_j[2][kFcHelpers][64], part_st[2][64];
    const ChainWork w = work[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t *X = xs + w.off, *Y = ys + w.off;
    int32_t *S = score_out + w.off, *P = parent_out + w.off;
    const int64_t n = w.n;
    const int32_t mdx = w.max_dist_x, mdy = w.max_dist_y, bw = w.bw;
    const uint64_t mdx64 = (uint64_t)(int64_t)mdx;
    const double avg_d = (double)w.avg_qspan;
    const float k32 = (float)(0.01 * (double)w.avg_qspan);
    const int64_t nblocks = (n + 63) / 64;

    // any_narrow: wave-uniform "some lane of this block has a window of <= 6 predecessors" (only near the start of a
    // call): only then is the double-precision gap cost evaluated at all (real branch, not a select)
    auto score_pred = [&](int32_t xa, int32_t ya, int32_t qsa, bool wide_a, bool any_narrow, int32_t xj, int32_t yj, int32_t sj,
                          bool &ok) -> int32_t {
        const int32_t ddr = (int32_t)((uint32_t)xa - (uint32_t)xj);
        const int32_t ddq = (int32_t)((uint32_t)ya - (uint32_t)yj);
        const uint32_t diff = (uint32_t)ddr - (uint32_t)ddq;
        const int32_t dd = (int32_t)((int32_t)diff < 0 ? 0u - diff : diff);
        ok = !(dd > bw || ddr == 0 || ddq <= 0 || ddq > mdy || ddq > mdx);
        const int32_t oc = min(min(ddr, ddq), qsa);
        const int32_t lg = dd ? ilog2_u32((uint32_t)dd) : 0;
        int32_t gc = (int32_t)floorf(__fmul_rn((float)dd, k32)) + (lg >> 1);
        if (any_narrow) {
            const int32_t gd = (int32_t)__dmul_rn(__dmul_rn((double)dd, .01), avg_d) + (lg >> 1);
            gc = wide_a ? gc : gd;
        }
        return (int32_t)((uint32_t)sj + (uint32_t)oc - (uint32_t)gc);
    };

    // helper state: the sequential window-start pointer (each helper keeps its own identical copy)
    int64_t st = 0, sb = 0;
    uint64_t XS = (wave > 0 && lane < n) ? X[lane] : 0;
    // main state: the previous block (the "near" predecessors)
    int32_t pxa = 0, pya = 0, pbest = 0;
    int pnb = 0;
    unsigned long long evals = 0;

    for (int64_t t = -1; t < nblocks; t++) {
        const int par = (int)((t + 1) & 1);                  // LDS slot of block t+1; block t lives in par ^ 1
        if (wave > 0) {
            // ------------------------------------------------ helpers: block t + 1
            const int64_t kb = t + 1;
            if (kb < nblocks) {
                const int64_t i0 = kb * 64;
                const int nb = (int)(n - i0 < 64 ? n - i0 : 64);
                const bool mine = lane < nb;
                const uint64_t xa64 = mine ? X[i0 + lane] : 0, ya64 = mine ? Y[i0 + lane] : 0;
                const int32_t xa = (int32_t)(uint32_t)xa64, ya = (int32_t)(uint32_t)ya64, qsa = (int32_t)(ya64 >> 32 & 0xff);
                int64_t st_a = 0;
                for (int a = 0; a < nb; a++) {               // host_kernel.cpp:200-207
                    const int64_t i = i0 + a;
                    const uint64_t xi = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(xa64 >> 32), a) << 32) |
                                        (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)xa64, a);
                    for (;;) {
                        const int64_t cand = sb + lane;
                        const bool pass = cand < st || (cand < i && (xi - XS) > mdx64);
                        const unsigned long long m = __ballot(pass);
                        if (m == ~0ull) { sb += 64; st = sb; XS = (sb + lane < n) ? X[sb + lane] : 0; continue; }
                        st = sb + __builtin_ctzll(~m);
                        break;
                    }
                    if (i - st > kMaxIter) st = i - kMaxIter;
                    if (st - sb >= 64) { sb = st & ~63ll; XS = (sb + lane < n) ? X[sb + lane] : 0; }
                    if (lane == a) st_a = st;
                }
                const int64_t ia = i0 + lane;
                const bool wide_a = !((ia - 1) - st_a <= 5);
                const bool any_narrow = __ballot(mine && !wide_a) != 0;
                const int st_rel = (int)(st_a - i0);
                int32_t best = (int32_t)0x80000000, best_j = -1;      // helpers start below any score; q_span is the main wave's floor
                const int64_t st_lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)st_a) |
                                      ((int64_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(st_a >> 32)) << 32);
                // far predecessors j <= i0 - 65, chunks of 64 dealt round-robin to the helpers
                for (int64_t jb = i0 - 65 - 64 * (wave - 1); jb >= st_lo; jb -= 64 * kFcHelpers) {
                    const int64_t jl = jb - lane;
                    int32_t vx = 0, vy = 0, vs = 0;
                    if (jl >= st_lo) {
                        vx = (int32_t)(uint32_t)X[jl]; vy = (int32_t)(uint32_t)Y[jl];
                        vs = __hip_atomic_load(&S[jl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const int cnt = (int)(jb - st_lo + 1 < 64 ? jb - st_lo + 1 : 64);
                    const int jrel0 = (int)(jb - i0);
                    for (int l = 0; l < cnt; l++) {
                        const int32_t xj = __builtin_amdgcn_readlane(vx, l), yj = __builtin_amdgcn_readlane(vy, l), sj = __builtin_amdgcn_readlane(vs, l);
                        const int jrel = jrel0 - l;
                        bool ok;
                        const int32_t sc = score_pred(xa, ya, qsa, wide_a, any_narrow, xj, yj, sj, ok);
                        if (mine && ok && jrel >= st_rel && sc > best) { best = sc; best_j = jrel; }      // descending j: strict >
                    }
                }
                part_best[par][wave - 1][lane] = best; part_j[par][wave - 1][lane] = best_j;
                if (wave == 1) part_st[par][lane] = st_rel;
            }
        } else if (t >= 0) {
            // ------------------------------------------------ main wave: block t
            const int64_t i0 = t * 64;
            const int nb = (int)(n - i0 < 64 ? n - i0 : 64);
            const bool mine = lane < nb;
            const uint64_t xa64 = mine ? X[i0 + lane] : 0, ya64 = mine ? Y[i0 + lane] : 0;
            const int32_t xa = (int32_t)(uint32_t)xa64, ya = (int32_t)(uint32_t)ya64, qsa = (int32_t)(ya64 >> 32 & 0xff);
            const int st_rel = part_st[par ^ 1][lane];
            const bool wide_a = !((lane - 1) - st_rel <= 5);
            const bool any_narrow = __ballot(mine && !wide_a) != 0;
            if (mine) evals += (unsigned long long)(lane - st_rel);
            int32_t best = qsa, best_j = -1;                 // best_j relative to i0 for predecessors, -1 = none
            bool have = false;
            // helpers' partial maxima: chunks interleave, so the larger j wins a tie
#pragma unroll
            for (int hh = 0; hh < kFcHelpers; hh++) {
                const int32_t b2 = part_best[par ^ 1][hh][lane], j2 = part_j[par ^ 1][hh][lane];
                if (b2 > best || (have && b2 == best && j2 > best_j)) { best = b2; best_j = j2; have = true; }
            }
            // near predecessors: the previous block, newest (lane 63) first; they are newer than every far one
            int32_t nbest = (int32_t)0x80000000, nbj = 0;
            bool nhave = false;
            for (int l = pnb - 1; l >= 0; l--) {
                const int32_t xj = __builtin_amdgcn_readlane(pxa, l), yj = __builtin_amdgcn_readlane(pya, l), sj = __builtin_amdgcn_readlane(pbest, l);
                const int jrel = l - 64;
                bool ok;
                const int32_t sc = score_pred(xa, ya, qsa, wide_a, any_narrow, xj, yj, sj, ok);
                if (mine && ok && jrel >= st_rel && (!nhave || sc > nbest)) { nbest = sc; nbj = jrel; nhave = true; }
            }
            if (nhave && (nbest > best || (have && nbest == best))) { best = nbest; best_j = nbj; have = true; }
            // predecessors inside the block
            for (int b = 0; b + 1 < nb; b++) {
                const int32_t xj = __builtin_amdgcn_readlane(xa, b), yj = __builtin_amdgcn_readlane(ya, b), sj = __builtin_amdgcn_readlane(best, b);
                bool ok;
                const int32_t sc = score_pred(xa, ya, qsa, wide_a, any_narrow, xj, yj, sj, ok);
                if (mine && lane > b && ok && b >= st_rel && (sc > best || (have && sc == best))) { best = sc; best_j = b; have = true; }
            }
            if (mine) { S[i0 + lane] = best; P[i0 + lane] = have ? (int32_t)(i0 + best_j) : -1; }
            pxa = xa; pya = ya; pbest = best; pnb = nb;
        }
        __syncthreads();       // results of block t are acknowledged by L2; partial maxima of block t+1 are in LDS
    }
    for (int o = 32; o > 0; o >>= 1) evals += __shfl_xor(evals, o);
    if (lane == 0 && evals) atomicAdd(evals_out, evals);
}

}  // namespace

// =============================================================================== host side
struct gab_chain {
    int device = 0;
    gab_devbuf work;       // ChainWork[ncalls] + evals counter
    gab_devbuf gmarks;     // chain mode: targets[] for windows deeper than the LDS mark ring (one int32 per anchor)
    gab_devbuf io;         // staging for the host-pointer entry point
    hipEvent_t ev[2] = {nullptr, nullptr};
    unsigned long long *h_evals = nullptr;   // pinned
    bool have_stats = false;
};

extern "C" int gab_chain_create(int device, gab_chain **out) {
    if (!out) { gab_set_error("gab_chain_create: NULL argument"); return GAB_EINVAL; }
    *out = nullptr;
    int rc = gab_check_device(device);
    if (rc) return rc;
    gab_device_guard g(device);
    gab_chain *h = new (std::nothrow) gab_chain();
    if (!h) { gab_set_error("out of host memory"); return GAB_ENOMEM; }
    h->device = device;
    if (hipEventCreate(&h->ev[0]) != hipSuccess || hipEventCreate(&h->ev[1]) != hipSuccess ||
        hipHostMalloc((void **)&h->h_evals, sizeof(unsigned long long)) != hipSuccess) {
        gab_set_error("gab_chain_create: event / pinned allocation failed"); delete h; return GAB_EDEVICE;
    }
    *out = h;
    return GAB_OK;
}

extern "C" void gab_chain_destroy(gab_chain *h) {
    if (!h) return;
    gab_device_guard g(h->device);
    h->work.release(); h->io.release(); h->gmarks.release();
    for (int k = 0; k < 2; k++) if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
    if (h->h_evals) (void)hipHostFree(h->h_evals);
    delete h;
}

static int chain_check_hdrs(const gab_chain_hdr *hdr, const int64_t *call_off, int64_t ncalls, int64_t *total) {
    int64_t end = 0;
    for (int64_t c = 0; c < ncalls; c++) {
        GAB_CHECK(hdr[c].n >= 0 && hdr[c].n < (1ll << 31), "gab_chain: call %lld has n=%lld (need 0 <= n < 2^31)",
                  (long long)c, (long long)hdr[c].n);
        GAB_CHECK(call_off[c] >= 0, "gab_chain: negative call_off[%lld]", (long long)c);
        end = std::max(end, call_off[c] + hdr[c].n);
    }
    *total = end;
    return GAB_OK;
}

extern "C" int gab_chain_run_device(gab_chain *h, int mode, const uint64_t *d_x, const uint64_t *d_y,
                                    const int64_t *call_off, const gab_chain_hdr *hdr, int64_t ncalls,
                                    int32_t *d_score, int32_t *d_parent, void *stream_) {
    GAB_CHECK(h, "gab_chain_run_device: NULL handle");
    GAB_CHECK(mode == GAB_CHAIN || mode == GAB_FASTCHAIN, "gab_chain_run_device: unknown mode %d", mode);
    GAB_CHECK(ncalls >= 0 && ncalls < (1ll << 31), "gab_chain_run_device: ncalls out of range");
    h->have_stats = false;
    if (ncalls == 0) return GAB_OK;
    GAB_CHECK(call_off && hdr, "gab_chain_run_device: NULL call table");
    int64_t total = 0;
    int rc = chain_check_hdrs(hdr, call_off, ncalls, &total);
    if (rc) return rc;
    GAB_CHECK(total == 0 || (d_x && d_y && d_score && d_parent), "gab_chain_run_device: NULL buffer");
    gab_device_guard g(h->device);
    hipStream_t s = (hipStream_t)stream_;

    // longest call first: the sequential walk of the biggest call is the critical path
    std::vector<ChainWork> wk;
    wk.reserve((size_t)ncalls);
    for (int64_t c = 0; c < ncalls; c++) {
        if (hdr[c].n == 0) continue;
        ChainWork w;
        w.off = call_off[c]; w.n = hdr[c].n; w.avg_qspan = hdr[c].avg_qspan;
        w.max_dist_x = hdr[c].max_dist_x; w.max_dist_y = hdr[c].max_dist_y; w.bw = hdr[c].bw;
        w.n_segs = hdr[c].n_segs; w.pad = 0;
        wk.push_back(w);
    }
    std::stable_sort(wk.begin(), wk.end(), [](const ChainWork &a, const ChainWork &b) { return a.n > b.n; });
    const size_t nw = wk.size();
    const size_t o_ev = (sizeof(ChainWork) * nw + 15) & ~(size_t)15;
    rc = h->work.reserve(o_ev + 16);
    if (rc) return rc;
    if (nw == 0) return GAB_OK;
    ChainWork *d_work = h->work.as<ChainWork>();
    unsigned long long *d_ev = (unsigned long long *)(h->work.as<char>() + o_ev);
    // pageable -> device copy of the small work list completes before the call returns to the
    // caller's stack frame being reused (hipMemcpyAsync from pageable memory stages synchronously)
    GAB_HIP(hipMemcpyAsync(d_work, wk.data(), sizeof(ChainWork) * nw, hipMemcpyHostToDevice, s));
    GAB_HIP(hipMemsetAsync(d_ev, 0, 16, s));
    int32_t *d_gm = nullptr;
    if (mode == GAB_CHAIN) {
        rc = h->gmarks.reserve(sizeof(int32_t) * (size_t)total);
        if (rc) return rc;
        d_gm = h->gmarks.as<int32_t>();
        GAB_HIP(hipMemsetAsync(d_gm, 0, sizeof(int32_t) * (size_t)total, s));      // vector::resize zero-fills targets
    }
    GAB_HIP(hipEventRecord(h->ev[0], s));
    if (mode == GAB_FASTCHAIN)
        hipLaunchKernelGGL(fastchain_kernel, dim3((unsigned)nw), dim3(64 * (1 + kFcHelpers)), 0, s, d_work, d_x, d_y, d_score, d_parent, d_ev);
    else
        hipLaunchKernelGGL(chain_kernel<false>, dim3((unsigned)nw), dim3(kThreads), 0, s, d_work, d_x, d_y, d_score, d_parent, d_gm, d_ev);
    GAB_HIP(hipGetLastError());
    GAB_HIP(hipEventRecord(h->ev[1], s));
    GAB_HIP(hipMemcpyAsync(h->h_evals, d_ev, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));    // wk (host vector) must outlive the H2D copy
    h->have_stats = true;
    return GAB_OK;
}

extern "C" int gab_chain_run(gab_chain *h, int mode, const uint64_t *x, const uint64_t *y,
                             const int64_t *call_off, const gab_chain_hdr *hdr, int64_t ncalls,
                             int32_t *score_out, int32_t *parent_out) {
    GAB_CHECK(h, "gab_chain_run: NULL handle");
    GAB_CHECK(ncalls >= 0, "gab_chain_run: ncalls < 0");
    if (ncalls == 0) return GAB_OK;
    GAB_CHECK(call_off && hdr, "gab_chain_run: NULL call table");
    int64_t total = 0;
    int rc = chain_check_hdrs(hdr, call_off, ncalls, &total);
    if (rc) return rc;
    if (total == 0) return GAB_OK;
    GAB_CHECK(x && y && score_out && parent_out, "gab_chain_run: NULL buffer");
    gab_device_guard g(h->device);
    const size_t t = (size_t)total;
    rc = h->io.reserve(24 * t + 64);
    if (rc) return rc;
    char *b = h->io.as<char>();
    uint64_t *dx = (uint64_t *)b, *dy = (uint64_t *)(b + 8 * t);
    int32_t *ds = (int32_t *)(b + 16 * t), *dp = (int32_t *)(b + 20 * t);
    hipStream_t s = nullptr;
    GAB_HIP(hipMemcpyAsync(dx, x, 8 * t, hipMemcpyHostToDevice, s));
    GAB_HIP(hipMemcpyAsync(dy, y, 8 * t, hipMemcpyHostToDevice, s));
    rc = gab_chain_run_device(h, mode, dx, dy, call_off, hdr, ncalls, ds, dp, s);
    if (rc) return rc;
    GAB_HIP(hipMemcpyAsync(score_out, ds, 4 * t, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipMemcpyAsync(parent_out, dp, 4 * t, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    return GAB_OK;
}

extern "C" int gab_chain_last_stats(gab_chain *h, int64_t *evals, float *kernel_ms) {
    GAB_CHECK(h, "gab_chain_last_stats: NULL handle");
    GAB_CHECK(h->have_stats, "gab_chain_last_stats: no completed run on this handle");
    gab_device_guard g(h->device);
    GAB_HIP(hipEventSynchronize(h->ev[1]));
    if (evals) *evals = (int64_t)*h->h_evals;
    if (kernel_ms) GAB_HIP(hipEventElapsedTime(kernel_ms, h->ev[0], h->ev[1]));
    return GAB_OK;
}
