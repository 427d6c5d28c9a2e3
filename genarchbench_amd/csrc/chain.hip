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
// Mapping.  The scores along i are a true recurrence (score[i] needs score[i-1]) and a batch takes as long as the walk
// of its slowest call, so everything is arranged around the latency of one anchor step:
//   * CHAIN (chain_hw_kernel): one main wave per call carries the dependence; the last 512 anchors (x, y, score,
//     parent, segment id) live in an LDS ring; each lane owns an aligned group of four predecessors, a super-chunk is
//     256 predecessors.  Two helper waves compute everything that does not depend on earlier results -- window start,
//     filters, min(dq, dr, q_span) and the gap cost of the 256 newest predecessors -- eight anchors ahead into LDS.
//     The sequential max_skip logic is reproduced exactly in three parallel steps: unfiltered
//     items scatter their marks targets[parent[j]] = i into a tagged 16-bit LDS ring (a global array for windows
//     deeper than the ring), the improvement flags come from an exclusive prefix-max (DPP scan), and the saturating
//     n_skip counter has a closed form over the prefix sums of (+1 hit, -1 improvement).  Marks scattered by items
//     past the break point are harmless because a mark value i is only ever compared with the current i.
//   * FASTCHAIN (fastchain_kernel): block formulation, see below.
//   * thousands of independent calls run concurrently, longest call first.
//
// Roofline: 24 B of HBM traffic per anchor (16 B in, 8 B out) against ~130-200 predecessor
// evaluations per anchor: latency/VALU bound by construction; the window re-reads are served by
// L1/L2, not HBM.
#include "gab_internal.h"
#include "chain_dev.h"
#include <algorithm>
#include <new>
#include <atomic>
#include <vector>
#include <chrono>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

// ---- the host-pointer path of big batches: the anchors come in by a KERNEL, longest call first -----------------------
// (see chain_run_fed).  A call's workgroup waits until its anchors are there and writes its results through to the caller's
// page-locked arrays block by block; all members null = the plain device path.
struct ChainFeed {
    uint32_t *facts;                          // per work item: 0 = anchors not here yet; 2 | (plain ? 1 : 0) once they are
    int32_t *host_score, *host_parent;        // device addresses of the caller's result arrays
    uint32_t *abort;                          // set by a wait that gave up; every other wait then gives up too
    unsigned long long *dbg;                  // diagnosis (GAB_CHAIN_TRACE): per work item wall-clock ticks at start / ready / done
    long spin_limit;                          // spins of a wait before it gives up (kFeedSpinLimit; tests: a few thousand)
};
struct ChainChunk { int64_t hoff, doff; int32_t n, item; };   // up to kFeedChunk anchors of work item `item`: where they are, where they go
constexpr int kFeedChunk = 2048;
constexpr long kFeedSpinLimit = 600000;       // x ~7 us of s_sleep: a wait gives up after seconds, so the grid always drains

// wait (thread 0, sleeping) until the anchors of this workgroup's call are in device memory; returns the facts word (0: gave up)
__device__ __forceinline__ uint32_t gab_xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu; }   // HW_REG_XCC_ID
__device__ __forceinline__ uint32_t chain_feed_wait(const ChainFeed &feed, uint32_t item, uint32_t *s_word) {
    if (threadIdx.x == 0) {
        uint32_t f = 0;
        for (long spins = 0;; spins++) {
            f = __hip_atomic_load(&feed.facts[item], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f) break;
            // (over a thousand workgroups wait at a time: a look at the ONE abort word per spin is a billion requests per second
            // on one address, and a spin per microsecond is more than the anchors' arrival needs)
            if (spins > feed.spin_limit || ((spins & 63) == 63 && __hip_atomic_load(feed.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(feed.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
        }
        *s_word = f;
    }
    __syncthreads();
    // No cache of this XCD can hold an older copy of the call's lines: the fed path starts every call on a 128-byte line of the
    // device arrays, nothing but this workgroup reads them, and the gather kernel stores write-through.  (An agent-scope
    // acquire here -- and __threadfence() in the gather kernel -- invalidates / writes back the whole L2 of the XCD: with
    // 10 000 workgroups and 48 000 chunks doing that, the DP ran at 0.42 of its speed while the anchors came in.)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return *s_word;
}

// Copies the anchors of the chunks, in table order (longest call first), from the caller's page-locked arrays (read over the
// bus by the lanes: 51-57 GB/s, profiles/r02_pcie_copy.md) to the device arrays, collects what chain_facts_kernel would
// (x range, one segment id?) and, with the last chunk of a call, publishes the call's facts word.
__global__ __launch_bounds__(256) void chain_gather_kernel(const ChainChunk *__restrict__ chunks, uint32_t nchunks,
                                                           const uint64_t *__restrict__ hx, const uint64_t *__restrict__ hy,
                                                           uint64_t *__restrict__ dx, uint64_t *__restrict__ dy,
                                                           const ChainWork *__restrict__ work, const uint32_t *__restrict__ need,
                                                           uint32_t *done, unsigned long long *xlo, unsigned long long *xhi,
                                                           uint32_t *mixed, uint32_t *facts, volatile uint8_t *started, int gap_tab_max,
                                                           uint32_t *next_chunk, unsigned long long *pub_dbg, uint32_t never_publish) {
    __shared__ unsigned long long s_lo[4], s_hi[4];
    __shared__ uint32_t s_mx[4];
    __shared__ uint32_t s_chunk;
    if (threadIdx.x == 0) started[blockIdx.x] = (uint8_t)(1u + gab_xcc_id());   // (host memory: the host launches the DP once every workgroup is resident, and learns where they are)
    // The chunks are handed out from ONE counter, in table order: workgroups read the bus at very different rates (a static
    // stride had the fastest workgroups through their last chunk after 15 ms and the slowest after 24 -- the 300th-longest
    // call, 19 % into the table, was complete after 16 ms instead of 5), and the DP's critical path is the longest calls.
    for (;;) {
        if (threadIdx.x == 0) s_chunk = atomicAdd(next_chunk, 1u);
        __syncthreads();
        const uint32_t c = s_chunk;
        if (c >= nchunks) break;
        const ChainChunk ch = chunks[c];
        const ChainWork w = work[ch.item];
        const uint32_t sid0 = (uint32_t)(hy[w.hoff] >> 48 & 0xff);
        unsigned long long lo = ~0ull, hi = 0;
        uint32_t mx = 0;
        // all loads of the chunk first (16 per thread): the bus needs megabytes in flight, not one load per thread
        constexpr int kPer = kFeedChunk / 256;
        uint64_t xv[kPer], yv[kPer];
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const int i = threadIdx.x + 256 * k, ii = i < ch.n ? i : 0;
            xv[k] = hx[ch.hoff + ii]; yv[k] = hy[ch.hoff + ii];
        }
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const int i = threadIdx.x + 256 * k;
            if (i < ch.n) {
                const uint64_t x = xv[k], y = yv[k];
                // write-through (device-scope) stores: the reader sits on any XCD, and the L2 of this one is not its L2
                __hip_atomic_store(&dx[ch.doff + i], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&dy[ch.doff + i], y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lo = x < lo ? x : lo; hi = x > hi ? x : hi;
                mx |= ((uint32_t)(y >> 48 & 0xff) != sid0) ? 1u : 0u;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
            lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; mx |= __shfl_xor(mx, o);
        }
        if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; s_mx[threadIdx.x >> 6] = mx; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's stores have completed (at the device's coherence point) ...
        __syncthreads();                                         // ... and so have everybody's, before thread 0 counts the chunk
        if (threadIdx.x == 0) {
            for (int k = 1; k < 4; k++) { lo = s_lo[k] < lo ? s_lo[k] : lo; hi = s_hi[k] > hi ? s_hi[k] : hi; mx |= s_mx[k]; }
            atomicMin(&xlo[ch.item], lo); atomicMax(&xhi[ch.item], hi);
            if (mx) atomicOr(&mixed[ch.item], 1u);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // the three have been performed before the chunk is counted
            if (atomicAdd(&done[ch.item], 1u) + 1u == need[ch.item]) {
                const unsigned long long L = __hip_atomic_load(&xlo[ch.item], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                         H = __hip_atomic_load(&xhi[ch.item], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t M = __hip_atomic_load(&mixed[ch.item], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int32_t mq = w.max_dist_y < w.max_dist_x ? w.max_dist_y : w.max_dist_x;
                const unsigned long long lim = mq < 0 ? 0ull : (unsigned long long)mq;
                const bool plain = w.n > 0 && !M && H - L + lim < 0x7fffffffull && w.bw >= 0 && w.bw <= gap_tab_max;   // = chain_facts_kernel
                // (never_publish: GAB_CHAIN_FEED_GIVEUP, a test hook -- one call's word stays 0, so that its workgroup's wait gives up)
                if ((uint32_t)ch.item != never_publish) __hip_atomic_store(&facts[ch.item], 2u | (plain ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (pub_dbg) pub_dbg[ch.item] = wall_clock64();
            }
        }
        __syncthreads();
    }
}

constexpr int kRing = 1024;               // anchors whose score / parent are kept in the LDS ring
constexpr int kRingSafe = kRing - 8;      // entries younger than this are read from the ring

// ---- chain: the sequential walk (main wave) -----------------------------------------------------------------------
// A call is walked by ONE wave (an earlier four-wave version spent most of its time on workgroup barriers and on
// instructions all four waves executed for a window one wave can hold).  Each lane owns an ALIGNED group of four ring
// entries (one ds_read_b128 per field), so a super-chunk is 256 predecessors and needs no barrier at all -- LDS
// operations of one wave complete in program order.  Visiting order (descending j)
// is lane order, and inside a lane item k = 0..3 <-> j = 4g+3-k.  The max_skip logic is the same three steps as
// before with a 4-item sequential part inside the lane and wave scans across the lanes:
//   marks     targets[parent[j]] = i for every unfiltered item (tagged 16-bit LDS ring / global array for deep windows)
//   improve   sc > max(prefix max over earlier items, best so far)
//   n_skip    counter reflected at 0: c = P - min(-c_in, running min of P), P = prefix sums of (+1 hit, -1 improvement)
// ---- chain: main wave + helper waves ------------------------------------------------------------------------------
// The walk of one call is a chain of dependent anchors, and the time of the whole batch is the walk of its longest
// call (~1.9 us per anchor with either kernel above).  Most of an anchor's instructions do not depend on earlier
// results at all: the filters, min(dq, dr, q_span) and the gap cost are pure geometry of (anchor, predecessor).
// Here helper waves compute that geometry one block of kChBlock anchors AHEAD (window start, 256 newest predecessors
// per anchor, coalesced reads of x / y) and leave it in LDS; the main wave, which alone carries the dependence, only
// adds score[j], and runs the marks / prefix-max / n_skip steps described above.  Windows deeper than 256
// predecessors continue in the main wave, which then evaluates the geometry itself.
constexpr int kChPriorityCalls = 128;         // the longest calls (= first workgroups) run at raised wave priority
#ifndef GAB_CH_HELPERS                        // tuning builds only (-DGAB_CH_HELPERS=.. -DGAB_CH_BLOCK=.. -DGAB_CH_GEODEPTH=..)
#define GAB_CH_HELPERS 3
#endif
#ifndef GAB_CH_BLOCK
#define GAB_CH_BLOCK GAB_CH_HELPERS
#endif
#ifndef GAB_CH_GEODEPTH
#define GAB_CH_GEODEPTH 1024
#endif
constexpr int kChHelpers = GAB_CH_HELPERS;
constexpr int kChBlock = GAB_CH_BLOCK;
constexpr int kGeoDepth = GAB_CH_GEODEPTH;   // predecessors per anchor the helpers prepare (kGeoDepth / 256 super-chunks)
constexpr int kGeoNone = (int)0x80000000;      // predecessor filtered out (or outside the window)

__global__ __launch_bounds__(64 * (1 + kChHelpers)) void chain_hw_kernel(const ChainWork *__restrict__ work,
                                                                         const uint64_t *__restrict__ xs,
                                                                         const uint64_t *__restrict__ ys, int32_t *score_out,
                                                                         int32_t *parent_out, int32_t *gmarks_all,
                                                                         unsigned long long *evals_out) {
    __shared__ __attribute__((aligned(16))) int32_t ring_sc[kRing];
    __shared__ __attribute__((aligned(16))) int32_t ring_par[kRing];
    __shared__ __attribute__((aligned(16))) uint16_t marks[kMarkRing];
    __shared__ __attribute__((aligned(16))) int32_t geo[2][kChBlock][kGeoDepth];
    __shared__ int32_t meta_st[2][kChBlock];
    __shared__ uint64_t meta_x[2][kChBlock], meta_y[2][kChBlock];

    const ChainWork w = work[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave index: uniform, say so
    const uint64_t *X = xs + w.off, *Y = ys + w.off;
    int32_t *S = score_out + w.off, *P = parent_out + w.off;
    int32_t *GM = gmarks_all + w.off;
    const int n = (int)w.n;                                  // < 2^31 (checked by the host): 32-bit indices inside a call
    const int32_t mdx = w.max_dist_x, mdy = w.max_dist_y, bw = w.bw;
    const uint64_t mdx64 = (uint64_t)(int64_t)mdx;
    const double avg_d = (double)w.avg_qspan;
    const bool multi_seg = w.n_segs > 1;
    const int NEG = (int)0x80000000;
    const int nblocks = (n + kChBlock - 1) / kChBlock;

    // Calls are launched longest first and the walk of the longest ones is the critical path of the batch: their waves
    // (main and helpers alike) get the instruction arbiter's preference over the mass of short calls (+2 %).
    if (blockIdx.x < kChPriorityCalls) __builtin_amdgcn_s_setprio(3);
    if (wave > 0) {
        // ================= helper: geometry of block t, one block ahead of the main wave
        int st = 0, sb = 0;
        uint64_t XS = (lane < n) ? X[lane] : 0;
        for (int t = 0; t < nblocks; t++) {
            const int buf = (int)(t & 1);
            for (int b = 0; b < kChBlock; b++) {
                const int i = t * kChBlock + b;
                if (i >= n) break;
                const uint64_t xi = X[i], yi = Y[i];       // wave-uniform
                // window start with the reference's sequential-pointer semantics (host_kernel.cpp:56-57); every helper
                // tracks it for every anchor (a ballot per anchor), so the helpers need no exchange among themselves
                for (;;) {
                    const int cand = sb + lane;
                    const bool far = xi > XS + mdx64;
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
                if (st - sb >= 64) { sb = st & ~63; XS = (sb + lane < n) ? X[sb + lane] : 0; }
                if ((b % kChHelpers) != wave - 1) continue;             // anchors are dealt round-robin to the helpers
                if (lane == 0) { meta_st[buf][b] = st; meta_x[buf][b] = xi; meta_y[buf][b] = yi; }
                const int32_t qi = (int32_t)yi, q_span = (int32_t)(yi >> 32 & 0xff), sidi = (int32_t)(yi >> 48 & 0xff);
                for (int p = 0; p < kGeoDepth / 64 && i - 1 - p * 64 >= st; p++) {
                    const int j = i - 1 - p * 64 - lane;
                    if (j < st) continue;                               // (also covers j < 0)
                    const uint64_t xj = X[j], yy = Y[j];
                    bool ok;
                    const int32_t v = chain_geometry(xi, qi, q_span, sidi, xj, (uint32_t)yy, (int32_t)(yy >> 48 & 0xff), mdx, mdy, bw,
                                                     multi_seg, avg_d, ok);
                    geo[buf][b][j & (kGeoDepth - 1)] = ok ? v : kGeoNone;
                }
            }
            __syncthreads();                                            // block t is ready / block t-1 is consumed
        }
        __syncthreads();                                                // pairs with the main wave's last barrier
        return;
    }

    // ================= main wave
    unsigned long long evals = 0, exact_evals = 0;
    int i = 0;
    __syncthreads();                                                    // block 0 is ready
    for (int t = 0; t < nblocks; t++) {
        const int buf = t & 1;
        for (int b = 0; b < kChBlock && i < n; b++, i++) {
            if ((i & 63) == 0) {
                if (i > 0) {                                   // results leave the CU in coalesced blocks of 64
                    const int jo = i - 64 + lane;
                    S[jo] = ring_sc[jo & (kRing - 1)]; P[jo] = ring_par[jo & (kRing - 1)];
                }
                if ((i & 0x7fff) == 0)
                    for (int k = lane; k < kMarkRing; k += 64) marks[k] = 0;       // new tag epoch
            }
            const int st = meta_st[buf][b];
            const uint64_t xi = meta_x[buf][b], yi = meta_y[buf][b];
            const int32_t qi = (int32_t)yi, q_span = (int32_t)(yi >> 32 & 0xff), sidi = (int32_t)(yi >> 48 & 0xff);

            // ---- fast path: the plain maximum over the whole window (ties -> larger j), no marks / n_skip bookkeeping.
            // It IS the reference's result whenever at most kMaxSkip unfiltered predecessors are newer than its argmax J*:
            // n_skip grows by at most one per unfiltered item, so the scan cannot have stopped before J*; at J* the score
            // beats everything newer (J* is the newest item with the maximum), and nothing older can improve on it, so
            // whatever max_skip does afterwards leaves (score, parent) alone.  With no improvement over q_span at all the
            // result is (q_span, -1) in both.  On the suite's inputs this certifies every anchor (dense adversarial sets:
            // 96-98 %); the rest take the exact path below.
            int32_t best = q_span, best_j = -1;
            bool certified;
            {
                int chunk = 0;
                for (int top = i - 1; top >= st; chunk++) {
                    const int g = (top >> 2) - lane;
                    const int j0 = 4 * g;
                    const bool first = chunk < kGeoDepth / 256;
                    const bool in_ring = i - j0 <= kRingSafe;
                    const bool any_valid = j0 + 3 >= st && j0 <= top;
                    int lm = NEG, lj = 0;
                    if (first) {
                        if (any_valid) {
                            const int r = (int)(j0 & (kRing - 1));
                            const int4 gv = *reinterpret_cast<const int4 *>(&geo[buf][b][j0 & (kGeoDepth - 1)]);
                            const int4 sv = *reinterpret_cast<const int4 *>(&ring_sc[r]);
                            const int gk[4] = {gv.w, gv.z, gv.y, gv.x}, sk[4] = {sv.w, sv.z, sv.y, sv.x};
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const int j = j0 + 3 - k;
                                const int v = gk[k] + sk[k];
                                if (j >= st && j <= top && gk[k] != kGeoNone && v > lm) { lm = v; lj = j; }
                            }
                        }
                    } else if (any_valid) {
                        if (!in_ring) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        int4 sv = {0, 0, 0, 0};
                        if (in_ring) sv = *reinterpret_cast<const int4 *>(&ring_sc[(int)(j0 & (kRing - 1))]);
                        const int sk[4] = {sv.w, sv.z, sv.y, sv.x};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int j = j0 + 3 - k;
                            if (j >= st && j <= top) {
                                const uint64_t xj = X[j], yy = Y[j];
                                const int scj = in_ring ? sk[k] : __hip_atomic_load(&S[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                bool okk;
                                const int32_t v = chain_geometry(xi, qi, q_span, sidi, xj, (uint32_t)yy, (int32_t)(yy >> 48 & 0xff), mdx, mdy, bw,
                                                                 multi_seg, avg_d, okk) + scj;
                                if (okk && v > lm) { lm = v; lj = j; }
                            }
                        }
                    }
                    const int m = __builtin_amdgcn_readlane(wave_incl_max(lm), 63);
                    if (m > best) {                              // wave-uniform; lane order is descending j: the first lane wins ties
                        const unsigned long long mm = __ballot(lm == m);
                        best = m; best_j = __builtin_amdgcn_readlane(lj, __builtin_ctzll(mm));
                    }
                    evals += (unsigned)max(min(top, j0 + 3) - max(st, j0) + 1, 0);
                    top = 4 * ((top >> 2) - 63) - 1;
                }
                // level 1: the argmax is among the kMaxSkip newest anchors (free); level 2: count the unfiltered ones newer
                // than it (only ever needed inside the region the helpers prepared)
                certified = best_j < 0 || i - 1 - best_j <= kMaxSkip;
                if (!certified && i - 1 - best_j < kGeoDepth - 8) {
                    int cnt = 0;
                    for (int top = i - 1; top > best_j; top = 4 * ((top >> 2) - 63) - 1) {
                        const int j0 = 4 * ((top >> 2) - lane);
                        if (j0 + 3 > best_j && j0 <= top) {
                            const int4 gv = *reinterpret_cast<const int4 *>(&geo[buf][b][j0 & (kGeoDepth - 1)]);
                            const int gk[4] = {gv.w, gv.z, gv.y, gv.x};
#pragma unroll
                            for (int k = 0; k < 4; k++) { const int j = j0 + 3 - k; cnt += (j > best_j && j <= top && gk[k] != kGeoNone) ? 1 : 0; }
                        }
                    }
                    cnt = __builtin_amdgcn_readlane(wave_incl_sum(cnt), 63);
                    certified = cnt <= kMaxSkip;
                }
            }
            if (!certified) {
                // ---- exact path: the reference's scan with its max_skip early exit, in parallel form
                best = q_span; best_j = -1;
                int n_skip = 0;
                bool broke = false;
                const uint16_t tag = (uint16_t)(0x8000 | (i & 0x7fff));

                int chunk = 0;
                for (int top = i - 1; top >= st && !broke; chunk++) {
                    const int g = (top >> 2) - lane;               // this lane's group: entries 4g .. 4g+3 (may be negative)
                    const int j0 = 4 * g;
                    const bool first = chunk < kGeoDepth / 256;    // the newest predecessors: geometry comes from the helpers
                    const bool in_ring = i - j0 <= kRingSafe;
                    const bool any_valid = j0 + 3 >= st && j0 <= top;
                    bool valid[4], ok[4];
                    int32_t sc[4], parj[4] = {-1, -1, -1, -1};
    #pragma unroll
                    for (int k = 0; k < 4; k++) { const int j = j0 + 3 - k; valid[k] = j >= st && j <= top; ok[k] = false; sc[k] = 0; }
                    if (first) {
                        if (any_valid) {
                            const int r = (int)(j0 & (kRing - 1));
                            const int4 gv = *reinterpret_cast<const int4 *>(&geo[buf][b][j0 & (kGeoDepth - 1)]);
                            const int4 sv = *reinterpret_cast<const int4 *>(&ring_sc[r]);
                            const int4 pv = *reinterpret_cast<const int4 *>(&ring_par[r]);
                            const int gk[4] = {gv.w, gv.z, gv.y, gv.x}, sk[4] = {sv.w, sv.z, sv.y, sv.x};
                            parj[0] = pv.w; parj[1] = pv.z; parj[2] = pv.y; parj[3] = pv.x;
    #pragma unroll
                            for (int k = 0; k < 4; k++) { ok[k] = valid[k] && gk[k] != kGeoNone; sc[k] = gk[k] + sk[k]; }
                        }
                    } else if (any_valid) {
                        uint64_t xj[4] = {0, 0, 0, 0}; uint32_t yj[4] = {0, 0, 0, 0};
                        int32_t scj[4] = {0, 0, 0, 0}, sidj[4] = {0, 0, 0, 0};
                        // beyond what the helpers prepared: x / y are input (plain loads), score / parent come from the LDS
                        // ring while young enough, else back from L2 (stored by this wave earlier)
                        if (!in_ring) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        int4 sv = {0, 0, 0, 0}, pv = {-1, -1, -1, -1};
                        if (in_ring) {
                            const int r = (int)(j0 & (kRing - 1));
                            sv = *reinterpret_cast<const int4 *>(&ring_sc[r]);
                            pv = *reinterpret_cast<const int4 *>(&ring_par[r]);
                        }
                        const int sk[4] = {sv.w, sv.z, sv.y, sv.x}, pk4[4] = {pv.w, pv.z, pv.y, pv.x};
    #pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int j = j0 + 3 - k;
                            if (valid[k]) {
                                xj[k] = X[j];
                                const uint64_t yy = Y[j];
                                yj[k] = (uint32_t)yy; sidj[k] = (int)(yy >> 48 & 0xff);
                                if (in_ring) { scj[k] = sk[k]; parj[k] = pk4[k]; }
                                else {
                                    scj[k] = __hip_atomic_load(&S[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    parj[k] = __hip_atomic_load(&P[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                            }
                        }
    #pragma unroll
                        for (int k = 0; k < 4; k++) {
                            bool okk;
                            const int32_t v = chain_geometry(xi, qi, q_span, sidi, xj[k], yj[k], sidj[k], mdx, mdy, bw, multi_seg, avg_d, okk);
                            ok[k] = valid[k] && okk;
                            sc[k] = v + scj[k];
                        }
                    }
                    // ---- marks: scatter, then read this group's own four tags (LDS is in order within the wave)
    #pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (ok[k] && parj[k] >= 0 && parj[k] >= st) {
                            if (i - parj[k] <= kRingSafe) marks[parj[k] & (kMarkRing - 1)] = tag;
                            else __hip_atomic_store(&GM[parj[k]], (int32_t)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    const bool deep = i - (4 * ((top >> 2) - 63)) > kRingSafe;        // wave-uniform: this super-chunk leaves the ring
                    if (deep) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    bool hit[4];
                    {   // a mark lives in the LDS ring or in the global array according to the age of the marked anchor itself
                        const uint2 mv = *reinterpret_cast<const uint2 *>(&marks[j0 & (kMarkRing - 1)]);
                        hit[0] = (mv.y >> 16) == tag; hit[1] = (mv.y & 0xffffu) == tag; hit[2] = (mv.x >> 16) == tag; hit[3] = (mv.x & 0xffffu) == tag;
                        if (!in_ring) {
    #pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const int j = j0 + 3 - k;
                                if (i - j > kRingSafe)
                                    hit[k] = ok[k] && __hip_atomic_load(&GM[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int32_t)(i + 1);
                            }
                        }
                    }
                    // ---- improvement flags
                    int lmax = NEG;
    #pragma unroll
                    for (int k = 0; k < 4; k++) lmax = ok[k] ? max(lmax, sc[k]) : lmax;
                    const int incl = wave_incl_max(lmax);
                    int run = max(wave_shr1(incl, NEG), best);
                    bool imp[4];
                    int d[4];
    #pragma unroll
                    for (int k = 0; k < 4; k++) {
                        imp[k] = ok[k] && sc[k] > run;
                        run = ok[k] ? max(run, sc[k]) : run;
                        d[k] = imp[k] ? -1 : (ok[k] && hit[k]) ? 1 : 0;
                    }
                    // ---- n_skip: reflected counter in closed form
                    const int p0 = d[0], p1 = p0 + d[1], p2 = p1 + d[2], p3 = p2 + d[3];
                    const int E = wave_incl_sum(p3) - p3;                              // exclusive sum over earlier lanes
                    const int mloc = min(min(p0, p1), min(p2, p3));
                    const int inclmin = wave_incl_min(E + mloc);
                    int rmin = min(-n_skip, wave_shr1(inclmin, 0x7fffffff));
                    int cnt[4];
                    const int pk[4] = {p0, p1, p2, p3};
                    int kfirst = 4;
    #pragma unroll
                    for (int k = 0; k < 4; k++) {
                        rmin = min(rmin, E + pk[k]);
                        cnt[k] = E + pk[k] - rmin;
                        if (d[k] == 1 && cnt[k] > kMaxSkip && kfirst == 4) kfirst = k;
                    }
                    const unsigned long long om = __ballot(kfirst < 4);
                    int fl = 64, fk = 4;
                    if (om) { fl = __builtin_ctzll(om); fk = __builtin_amdgcn_readlane(kfirst, fl); broke = true; }
                    else n_skip = __builtin_amdgcn_readlane(cnt[3], 63);
                    // ---- best = the last improvement before the break
                    const int klim = lane < fl ? 4 : lane == fl ? fk : 0;              // items k < klim of this lane count
                    int lastk = -1, lsc = 0, lj = 0;
    #pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (imp[k] && k < klim) { lastk = k; lsc = sc[k]; lj = j0 + 3 - k; }
                    const unsigned long long lm = __ballot(lastk >= 0);
                    if (lm) {
                        const int ll = 63 - __builtin_clzll(lm);
                        best = __builtin_amdgcn_readlane(lsc, ll);
                        best_j = __builtin_amdgcn_readlane(lj, ll);
                    }
                    // visited predecessors (statistics): valid items up to and including the break item
                    const int vlim = lane < fl ? 4 : lane == fl ? fk + 1 : 0;
    #pragma unroll
                    for (int k = 0; k < 4; k++) exact_evals += (valid[k] && k < vlim) ? 1 : 0;
                    top = 4 * ((top >> 2) - 63) - 1;
                }
            }
            if (lane == 0) {
                const int r = i & (kRing - 1);
                ring_sc[r] = best; ring_par[r] = best_j;
            }
        }
        __syncthreads();                                                // block t consumed, block t+1 ready
    }
    {   // flush the tail: anchors [done, n) with done = the last multiple of 64 below n
        const int done = n > 0 ? ((n - 1) & ~63) : 0;
        const int jo = done + lane;
        if (jo < n) { S[jo] = ring_sc[jo & (kRing - 1)]; P[jo] = ring_par[jo & (kRing - 1)]; }
    }
    evals += exact_evals;                                               // both passes of an anchor that needed the exact scan
    for (int o = 32; o > 0; o >>= 1) evals += __shfl_xor(evals, o);
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
// anchors as the DP kernels see them: __restrict__ (loads may move across the kernel's own stores) except in the fed variant,
// where another kernel writes them while this one waits
template <bool FED> struct AnchorPtr { using type = const uint64_t *__restrict__; };
template <> struct AnchorPtr<true> { using type = const uint64_t *; };
// ... and how they are READ.  In the fed variant the anchors were written by ANOTHER kernel that is still running
// (chain_gather_kernel, relaxed agent-scope stores + the per-call facts word).  The wait orders the two only at workgroup
// scope (an agent-scope acquire costs the whole L2 of the XCD, see chain_feed_wait), so the data accesses themselves must
// be coherent at agent scope: every read of x / y in the fed kernels is a relaxed agent-scope atomic load (sc1 on gfx950: it
// is served from the device's coherence point, never from a stale line of this XCD's L2 or this CU's L1), which pairs
// with the gather kernel's write-through stores.  The plain variant keeps ordinary loads.
template <bool FED> struct AnchorView {
    const uint64_t *p;
    __device__ __forceinline__ uint64_t operator[](int64_t i) const {
#ifndef GAB_KO_FEDPLAIN
        if (FED) return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        return p[i];
    }
};
template <int H, bool FED, bool THROUGH = FED>      // THROUGH: every block of 64 results is also written to feed.host_score / host_parent
__device__ __forceinline__ void fastchain_body(const ChainWork *__restrict__ work, typename AnchorPtr<FED>::type xs,
                                               typename AnchorPtr<FED>::type ys, int32_t *score_out, int32_t *parent_out,
                                               unsigned long long *evals_out, ChainFeed feed) {
    __shared__ int32_t part_best[2][H][64], part_j[2][H][64], part_st[2][64];
    __shared__ uint32_t feed_word;
    const uint32_t item = blockIdx.x;
    if (FED && chain_feed_wait(feed, item, &feed_word) == 0) return;
    const ChainWork w = work[item];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave index: uniform, say so
    const AnchorView<FED> X{xs + w.off}, Y{ys + w.off};
    int32_t *S = score_out + w.off, *P = parent_out + w.off;
    const int64_t n = w.n;
    const int32_t mdx = w.max_dist_x, mdy = w.max_dist_y, bw = w.bw;
    const uint64_t mdx64 = (uint64_t)(int64_t)mdx;
    const double avg_d = (double)w.avg_qspan;
    const float k32 = (float)(0.01 * (double)w.avg_qspan);
    const int32_t mq = mdy < mdx ? mdy : mdx;
    const uint32_t mq_u = mq < 0 ? 0u : (uint32_t)mq;       // negative limits reject every predecessor
    const int64_t nblocks = (n + 63) / 64;

    __shared__ int32_t gap_tab[kGapTab];
    // any_narrow: wave-uniform "some lane of this block has a window of <= 6 predecessors" (only near the start of a
    // call): only then is the double-precision gap cost evaluated at all (real branch, not a select)
    // NARROW is a compile-time tag: written as a run-time `if (any_narrow)` the compiler if-converts the branch and the four
    // double-precision instructions are executed on every step of every loop
    auto score_pred = [&](auto narrow_tag, int32_t xa, int32_t ya, int32_t qsa, bool wide_a, int32_t xj, int32_t yj, int32_t sj,
                          bool &ok) -> int32_t {
        constexpr bool NARROW = decltype(narrow_tag)::value;
        const int32_t ddr = (int32_t)((uint32_t)xa - (uint32_t)xj);
        const int32_t ddq = (int32_t)((uint32_t)ya - (uint32_t)yj);
        const int32_t diff = (int32_t)((uint32_t)ddr - (uint32_t)ddq);
        const int32_t dd = max(diff, (int32_t)(0u - (uint32_t)diff));          // |diff| with the AVX wrap-around for INT_MIN
        // ddq <= 0 || ddq > max_dist_y || ddq > max_dist_x  ==  (unsigned)(ddq - 1) >= min(max_dist_y, max_dist_x)
        ok = !(dd > bw || ddr == 0 || (uint32_t)ddq - 1u >= mq_u);
        const int32_t oc = min(min(ddr, ddq), qsa);
        // ilog2(dd) >> 1 with ilog2(0) = 0:  (31 - clz(dd | 1)) >> 1 = 15 - (clz(dd | 1) >> 1)
        int32_t gc;
        if constexpr (NARROW) {
            const int32_t lgh = 15 - (__clz((int)((uint32_t)dd | 1u)) >> 1);
            gc = (int32_t)floorf(__fmul_rn((float)dd, k32)) + lgh;
            const int32_t gd = (int32_t)__dmul_rn(__dmul_rn((double)dd, .01), avg_d) + lgh;
            gc = wide_a ? gc : gd;
        } else {
            // the fp32 gap cost of every dd a pair can pass the filter with: 0 .. bw from the table of the call, and INT_MIN (the
            // only negative |diff|, which `dd > bw` lets through like the AVX code does) from entry bw + 1, where every dd > bw lands
            gc = gap_tab[min((uint32_t)dd, (uint32_t)bw + 1u)];
        }
        return (int32_t)((uint32_t)sj + (uint32_t)oc - (uint32_t)gc);
    };
    // (blocks in which some lane has a narrow window, and calls whose bw does not fit the table, take the arithmetic variant)
    const bool use_tab = bw >= 0 && bw <= kGapTab - 2;
    if (use_tab) {
        for (int d = threadIdx.x; d <= bw + 1; d += 64 * (1 + H)) {
            const int32_t dv = d <= bw ? d : (int32_t)0x80000000;
            gap_tab[d] = (int32_t)floorf(__fmul_rn((float)dv, k32)) + (15 - (__clz((int)((uint32_t)dv | 1u)) >> 1));
        }
        __syncthreads();
    }

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
                for (int64_t jb = i0 - 65 - 64 * (wave - 1); jb >= st_lo; jb -= 64 * H) {
                    const int64_t jl = jb - lane;
                    int32_t vx = 0, vy = 0, vs = 0;
                    if (jl >= st_lo) {
                        vx = (int32_t)(uint32_t)X[jl]; vy = (int32_t)(uint32_t)Y[jl];
                        vs = __hip_atomic_load(&S[jl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const int cnt = (int)(jb - st_lo + 1 < 64 ? jb - st_lo + 1 : 64);
                    const int jrel0 = (int)(jb - i0);
                    // (four steps per trip by hand: the evaluations are independent, only the running maximum is a chain)
                    auto far_chunk = [&](auto tag) {
                        auto step = [&](int l) {
                            const int32_t xj = __builtin_amdgcn_readlane(vx, l), yj = __builtin_amdgcn_readlane(vy, l), sj = __builtin_amdgcn_readlane(vs, l);
                            const int jrel = jrel0 - l;
                            bool ok;
                            const int32_t sc = score_pred(tag, xa, ya, qsa, wide_a, xj, yj, sj, ok);
                            if (mine && ok && jrel >= st_rel && sc > best) { best = sc; best_j = jrel; }      // descending j: strict >
                        };
                        int l = 0;
                        for (; l + 3 < cnt; l += 4) { step(l); step(l + 1); step(l + 2); step(l + 3); }
                        for (; l < cnt; l++) step(l);
                    };
                    // the table variant with the four gap-table reads of a trip in flight together (r03; left to the compiler every
                    // step waited for its own ds_read_b32): filters, overlap term and table index first, then the four reads
                    auto far_chunk_tab = [&]() {
                        auto pre = [&](int l, uint32_t &idx, bool &ok) -> int32_t {
                            const int32_t xj = __builtin_amdgcn_readlane(vx, l), yj = __builtin_amdgcn_readlane(vy, l);
                            const int32_t ddr = (int32_t)((uint32_t)xa - (uint32_t)xj);
                            const int32_t ddq = (int32_t)((uint32_t)ya - (uint32_t)yj);
                            const int32_t diff = (int32_t)((uint32_t)ddr - (uint32_t)ddq);
                            const int32_t dd = max(diff, (int32_t)(0u - (uint32_t)diff));
                            ok = !(dd > bw || ddr == 0 || (uint32_t)ddq - 1u >= mq_u);
                            idx = min((uint32_t)dd, (uint32_t)bw + 1u);
                            return min(min(ddr, ddq), qsa);
                        };
                        auto fold = [&](int32_t oc, int32_t gc, bool ok, int l) {
                            const int32_t sc = (int32_t)((uint32_t)__builtin_amdgcn_readlane(vs, l) + (uint32_t)oc - (uint32_t)gc);
                            if (mine && ok && jrel0 - l >= st_rel && sc > best) { best = sc; best_j = jrel0 - l; }
                        };
                        int l = 0;
                        for (; l + 3 < cnt; l += 4) {
                            uint32_t i0_, i1_, i2_, i3_; bool o0, o1, o2, o3;
                            const int32_t c0 = pre(l, i0_, o0), c1 = pre(l + 1, i1_, o1), c2 = pre(l + 2, i2_, o2), c3 = pre(l + 3, i3_, o3);
                            const int32_t g0 = gap_tab[i0_], g1 = gap_tab[i1_], g2 = gap_tab[i2_], g3 = gap_tab[i3_];
                            fold(c0, g0, o0, l); fold(c1, g1, o1, l + 1); fold(c2, g2, o2, l + 2); fold(c3, g3, o3, l + 3);
                        }
                        for (; l < cnt; l++) { uint32_t ix; bool o; const int32_t c = pre(l, ix, o); fold(c, gap_tab[ix], o, l); }
                    };
                    if (any_narrow || !use_tab) far_chunk(std::true_type{}); else far_chunk_tab();
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
            for (int hh = 0; hh < H; hh++) {
                const int32_t b2 = part_best[par ^ 1][hh][lane], j2 = part_j[par ^ 1][hh][lane];
                if (b2 > best || (have && b2 == best && j2 > best_j)) { best = b2; best_j = j2; have = true; }
            }
            // near predecessors: the previous block, newest (lane 63) first; they are newer than every far one
            int32_t nbest = (int32_t)0x80000000, nbj = 0;
            bool nhave = false;
            auto near_fold = [&](auto tag) {
                auto step = [&](int l) {
                    const int32_t xj = __builtin_amdgcn_readlane(pxa, l), yj = __builtin_amdgcn_readlane(pya, l), sj = __builtin_amdgcn_readlane(pbest, l);
                    const int jrel = l - 64;
                    bool ok;
                    const int32_t sc = score_pred(tag, xa, ya, qsa, wide_a, xj, yj, sj, ok);
                    if (mine && ok && jrel >= st_rel && (!nhave || sc > nbest)) { nbest = sc; nbj = jrel; nhave = true; }
                };
                int l = pnb - 1;
                for (; l >= 3; l -= 4) { step(l); step(l - 1); step(l - 2); step(l - 3); }
                for (; l >= 0; l--) step(l);
            };
            if (any_narrow || !use_tab) near_fold(std::true_type{}); else near_fold(std::false_type{});
            if (nhave && (nbest > best || (have && nbest == best))) { best = nbest; best_j = nbj; have = true; }
            // predecessors inside the block
            // The only true chain: anchor b's score is final after the folds 0 .. b-1 and feeds b+1 ...  Everything about
            // the pair (a, b) that does not involve a score -- filters, overlap, gap cost -- is computed four steps ahead,
            // so the dependent part of a step is readlane(best, b) + add + compare + select.
            auto block_fold = [&](auto tag) {
                auto geom = [&](int b, bool &ok) -> int32_t {
                    const int32_t xj = __builtin_amdgcn_readlane(xa, b), yj = __builtin_amdgcn_readlane(ya, b);
                    const int32_t g = score_pred(tag, xa, ya, qsa, wide_a, xj, yj, 0, ok);          // oc - gc (sj = 0)
                    ok = ok && mine && lane > b && b >= st_rel;
                    return g;
                };
                auto fold = [&](int b, int32_t g, bool ok) {
                    const int32_t sc = (int32_t)((uint32_t)__builtin_amdgcn_readlane(best, b) + (uint32_t)g);
                    if (ok && (sc > best || (have && sc == best))) { best = sc; best_j = b; have = true; }
                };
                int b = 0;
                for (; b + 4 < nb; b += 4) {
                    bool o0, o1, o2, o3;
                    const int32_t g0 = geom(b, o0), g1 = geom(b + 1, o1), g2 = geom(b + 2, o2), g3 = geom(b + 3, o3);
                    fold(b, g0, o0); fold(b + 1, g1, o1); fold(b + 2, g2, o2); fold(b + 3, g3, o3);
                }
                for (; b + 1 < nb; b++) { bool o; const int32_t g = geom(b, o); fold(b, g, o); }
            };
            if (any_narrow || !use_tab) block_fold(std::true_type{}); else block_fold(std::false_type{});
            if (mine) {
                const int32_t par_ = have ? (int32_t)(i0 + best_j) : -1;
                S[i0 + lane] = best; P[i0 + lane] = par_;
                if (THROUGH && feed.host_score) { feed.host_score[w.hoff + i0 + lane] = best; feed.host_parent[w.hoff + i0 + lane] = par_; }
            }
            pxa = xa; pya = ya; pbest = best; pnb = nb;
        }
        __syncthreads();       // results of block t are acknowledged by L2; partial maxima of block t+1 are in LDS
    }
    for (int o = 32; o > 0; o >>= 1) evals += __shfl_xor(evals, o);
    if (lane == 0 && evals) atomicAdd(evals_out, evals);
}

// throughput form (three helpers): seven waves per SIMD, +3.7 %; latency form: no register cap (see chain_block_kernel)
template <int H, bool FED, bool THROUGH = FED>
__global__ __launch_bounds__(64 * (1 + H)) __attribute__((amdgpu_waves_per_eu(7, 7)))
void fastchain_kernel(const ChainWork *__restrict__ work, typename AnchorPtr<FED>::type xs, typename AnchorPtr<FED>::type ys, int32_t *score_out,
                      int32_t *parent_out, unsigned long long *evals_out, ChainFeed feed) {
    fastchain_body<H, FED, THROUGH>(work, xs, ys, score_out, parent_out, evals_out, feed);
}
template <int H, bool FED>
__global__ __launch_bounds__(64 * (1 + H))
void fastchain_kernel_lat(const ChainWork *__restrict__ work, typename AnchorPtr<FED>::type xs, typename AnchorPtr<FED>::type ys, int32_t *score_out,
                          int32_t *parent_out, unsigned long long *evals_out, ChainFeed feed) {
    fastchain_body<H, FED>(work, xs, ys, score_out, parent_out, evals_out, feed);
}


// ---- chain: block formulation with a certificate (chain_block_kernel) ----------------------------------------------
// chain's max_skip makes the reference's scan of an anchor's window order-dependent, but only rarely outcome-dependent:
// the result is the plain maximum over the window (ties -> larger j) whenever at most kMaxSkip unfiltered predecessors are
// newer than its argmax (proof at the fast path of chain_hw_kernel above).  So chain is computed like fast-chain -- blocks
// of 64 anchors, lane a <-> anchor i0 + a, every lane scoring a BROADCAST predecessor against its own anchor, helper waves
// folding the far predecessors one block ahead -- with chain's own arithmetic (64-bit coordinates, segment ids, the fp64
// gap cost of chain_geometry), plus one counter per lane:
//     risk = unfiltered predecessors folded after (= newer than) the current argmax      (0 on every improvement)
// Predecessors are folded oldest first (far chunks, previous block, inside the block), so `risk` is exact for an argmax
// in the previous or the current block; for a far argmax it starts from the number of ALL unfiltered far predecessors
// (an upper bound: the helpers fold interleaved chunks).  An anchor whose risk exceeds kMaxSkip when it becomes final is
// re-done by the whole wave with the reference's own scan (chain_exact_global: marks, hits, n_skip, break) before its
// score is broadcast to the younger anchors of the block.  On the suite's inputs that is ~1 % of the anchors.
constexpr int kCbHelpers = 3;

// Per-call facts the block kernel specialises on (one workgroup per call, before the DP; ChainWork.pad bit 0):
//   plain = every anchor of the call carries the same segment id (then "sidi == sidj" is always true and the cross-segment
//           gap rule never applies) AND max(x) - min(x) < 2^31 (then every x[i] - x[j] of the call is exact in 32 bits).
// Both hold for every call of the suite's inputs (one reference strand per call); calls that miss either use the generic
// arithmetic.  Reads x and y once: 16 B per seed at HBM speed.
__global__ __launch_bounds__(256) void chain_facts_kernel(ChainWork *work, const uint64_t *__restrict__ xs, const uint64_t *__restrict__ ys,
                                                          const uint32_t *gate = nullptr) {
    if (gate && gate[blockIdx.x] == 0) return;      // (the launch behind the table form: only the calls it handed back)
    __shared__ unsigned long long s_lo[4], s_hi[4];
    __shared__ int s_mixed[4];
    ChainWork &w = work[blockIdx.x];
    const uint64_t *X = xs + w.off, *Y = ys + w.off;
    const int64_t n = w.n;
    unsigned long long lo = ~0ull, hi = 0;
    const uint32_t sid0 = n > 0 ? (uint32_t)(Y[0] >> 48 & 0xff) : 0;
    int mixed = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const unsigned long long x = X[i];
        lo = x < lo ? x : lo; hi = x > hi ? x : hi;
        mixed |= ((uint32_t)(Y[i] >> 48 & 0xff) != sid0) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; mixed |= __shfl_xor(mixed, o);
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; s_mixed[threadIdx.x >> 6] = mixed; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) { lo = s_lo[k] < lo ? s_lo[k] : lo; hi = s_hi[k] > hi ? s_hi[k] : hi; mixed |= s_mixed[k]; }
        // ... and, for the gap-cost table of the block kernel: 0 <= bw <= kGapTab - 2, and dq - dr cannot wrap for a pair that
        // passes the dq filter (dq in [1, min(max_dist_x, max_dist_y)]): then dd = |dr - dq| is an exact value in [0, 2^31)
        const int32_t mq = w.max_dist_y < w.max_dist_x ? w.max_dist_y : w.max_dist_x;
        const unsigned long long lim = mq < 0 ? 0ull : (unsigned long long)mq;
        w.pad = (n > 0 && !mixed && hi - lo + lim < 0x7fffffffull && w.bw >= 0 && w.bw <= kGapTab - 2) ? 1 : 0;
    }
}

// chain_geometry for a call with the facts above: one segment id, 32-bit-exact x differences.  `xa_lo` / `xj_lo` are the low
// words of x; dq_lim = min(max_dist_y, max_dist_x) clamped at 0.  Same value and same filter as chain_geometry (the `same`
// branch with dr held in 32 bits: dd = |dr - dq| wraps exactly like the reference's int64 -> int32 conversion).
// The gap cost -- (int)(dd * .01 * avg_qspan) in fp64 (two conversions and two multiplications at the fp64 rate) + ilog2(dd) / 2,
// rounded -- depends on dd alone, and a pair with dd > bw is filtered: gap_tab[dd] for dd = 0 .. bw (built per call by
// chain_gap_cost below, in LDS), one ds_read instead of ten instructions.  Pairs with dd > bw read entry bw + 1 (never used).
// (chain_gap_cost: chain_dev.h)
// the part of chain_geometry_plain before the table: the overlap term and the table index.  Callers that evaluate several
// pairs in a row take the indices of all of them first and read the table afterwards, so that the LDS reads are in flight
// together (left to the compiler every step waits for its own ds_read_b32).  MSEG: the call has n_segs > 1 (dr > max_dist_y rule)
template <bool MSEG>
__device__ __forceinline__ int32_t chain_geometry_plain_pre(uint32_t xa_lo, int32_t qa, int32_t q_span, uint32_t xj_lo, uint32_t yj, int32_t mdy,
                                                            uint32_t dq_lim, int32_t bw, uint32_t &idx, bool &ok) {
    const int32_t dr = (int32_t)(xa_lo - xj_lo);
    const int32_t dq = qa - (int32_t)yj;
    const int32_t diff = (int32_t)((uint32_t)dr - (uint32_t)dq);
    const int32_t dd = max(diff, (int32_t)(0u - (uint32_t)diff));
    ok = !(dr == 0 || (uint32_t)dq - 1u >= dq_lim || dd > bw || (MSEG && dr > mdy));
    idx = min((uint32_t)dd, (uint32_t)bw + 1u);
    return min(min(dq, dr), q_span);
}
__device__ __forceinline__ int32_t chain_geometry_plain(uint32_t xa_lo, int32_t qa, int32_t q_span, uint32_t xj_lo, uint32_t yj, int32_t mdy,
                                                        uint32_t dq_lim, int32_t bw, bool multi_seg, const int32_t *gap_tab, bool &ok) {
    const int32_t dr = (int32_t)(xa_lo - xj_lo);
    const int32_t dq = qa - (int32_t)yj;
    const int32_t diff = (int32_t)((uint32_t)dr - (uint32_t)dq);
    // |dr - dq|: exact for every pair that passes the dq filter (chain_facts_kernel: dq - dr cannot wrap then); a pair that
    // fails it is filtered whatever dd says, so the reference's `dr > dq ? dr - dq : dq - dr` on wrapped values is not needed
    const int32_t dd = max(diff, (int32_t)(0u - (uint32_t)diff));
    // dq <= 0 || dq > max_dist_y || dq > max_dist_x  ==  (unsigned)(dq - 1) >= min(max_dist_y, max_dist_x)
    ok = !(dr == 0 || (uint32_t)dq - 1u >= dq_lim || dd > bw || (multi_seg && dr > mdy));
    const int32_t v = min(min(dq, dr), q_span);
    return v - gap_tab[min((uint32_t)dd, (uint32_t)bw + 1u)];
}


template <int H, bool FED, bool THROUGH = FED>
__device__ __forceinline__ void chain_block_body(const ChainWork *__restrict__ work, typename AnchorPtr<FED>::type xs,
                                                 typename AnchorPtr<FED>::type ys, int32_t *score_out, int32_t *parent_out,
                                                 int32_t *gmarks_all, unsigned long long *evals_out, ChainFeed feed) {
    __shared__ int32_t part_best[2][H][64], part_j[2][H][64], part_ok[2][H][64], part_st[2][64];
    __shared__ uint32_t feed_word;
    uint32_t fed_facts = 0;
    const uint32_t item = blockIdx.x;
    if (FED && feed.dbg && threadIdx.x == 0) feed.dbg[3 * item] = wall_clock64() << 4 | gab_xcc_id();
    if (FED && (fed_facts = chain_feed_wait(feed, item, &feed_word)) == 0) return;
    if (FED && feed.dbg && threadIdx.x == 0) feed.dbg[3 * item + 1] = wall_clock64();
    const ChainWork w = work[item];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const AnchorView<FED> X{xs + w.off}, Y{ys + w.off};
    int32_t *S = score_out + w.off, *P = parent_out + w.off, *GM = gmarks_all + w.off;
    const int n = (int)w.n;                                  // < 2^31 (checked by the host)
    const int32_t mdx = w.max_dist_x, mdy = w.max_dist_y, bw = w.bw;
    const uint64_t mdx64 = (uint64_t)(int64_t)mdx;
    const double avg_d = (double)w.avg_qspan;
    const bool multi_seg = w.n_segs > 1;
    const bool plain = ((FED ? fed_facts : (uint32_t)w.pad) & 1u) != 0;   // chain_facts_kernel / chain_gather_kernel: one segment id, 32-bit-exact x differences, bw fits the table
    const int32_t mq = mdy < mdx ? mdy : mdx;
    const uint32_t dq_lim = mq < 0 ? 0u : (uint32_t)mq;
    const int nblocks = (n + 63) / 64;
    const int NEG = (int)0x80000000;
    __shared__ int32_t gap_tab[kGapTab];
    if (plain) {
        for (int d = threadIdx.x; d <= bw + 1; d += 64 * (1 + H)) gap_tab[d] = chain_gap_cost(d, avg_d);
        __syncthreads();
    }

    // geometry of (own anchor, broadcast predecessor): PLAIN is a compile-time tag so that the generic arithmetic (64-bit
    // differences, segment rules) costs the plain calls nothing; the predecessor's fields come from `src` lanes of
    // registers (v_readlane), only the ones the variant needs
    struct Pred { uint64_t x; uint32_t y; int32_t sid; };
    auto geom = [&](auto tag, uint64_t xa, int32_t qa, int32_t qsa, int32_t sida, const Pred &pv, int src, bool &ok) -> int32_t {
        constexpr bool PLAIN = decltype(tag)::value;
        const uint32_t xj_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pv.x, src);
        const uint32_t yj = (uint32_t)__builtin_amdgcn_readlane((int)pv.y, src);
        if constexpr (PLAIN) return chain_geometry_plain((uint32_t)xa, qa, qsa, xj_lo, yj, mdy, dq_lim, bw, multi_seg, gap_tab, ok);
        else {
            const uint64_t xj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pv.x >> 32), src) << 32) | xj_lo;
            return chain_geometry(xa, qa, qsa, sida, xj, yj, __builtin_amdgcn_readlane(pv.sid, src), mdx, mdy, bw, multi_seg, avg_d, ok);
        }
    };

    // helper state: the sequential window-start pointer (each helper keeps its own identical copy)
    int st = 0, sb = 0;
    uint64_t XS = (wave > 0 && lane < n) ? X[lane] : 0;
    // main state: the previous block (the "near" predecessors)
    Pred prev = {0, 0, 0};
    int32_t pbest = 0;
    int pnb = 0;
    unsigned long long evals = 0;

    for (int t = -1; t < nblocks; t++) {
        const int par = (t + 1) & 1;                         // LDS slot of block t+1; block t lives in par ^ 1
        if (wave > 0) {
            // ------------------------------------------------ helpers: block t + 1
            const int kb = t + 1;
            if (kb < nblocks) {
                const int i0 = kb * 64;
                const int nb = n - i0 < 64 ? n - i0 : 64;
                const bool mine = lane < nb;
                const uint64_t xa = mine ? X[i0 + lane] : 0, ya = mine ? Y[i0 + lane] : 0;
                const int32_t qa = (int32_t)ya, qsa = (int32_t)(ya >> 32 & 0xff), sida = (int32_t)(ya >> 48 & 0xff);
                int st_a = 0;
                for (int a = 0; a < nb; a++) {               // host_kernel.cpp:56-57
                    const int i = i0 + a;
                    const uint64_t xi = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(xa >> 32), a) << 32) |
                                        (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)xa, a);
                    for (;;) {
                        const int cand = sb + lane;
                        const bool pass = cand < st || (cand < i && xi > XS + mdx64);
                        const unsigned long long m = __ballot(pass);
                        if (m == ~0ull) { sb += 64; st = sb; XS = (sb + lane < n) ? X[sb + lane] : 0; continue; }
                        st = sb + __builtin_ctzll(~m);
                        break;
                    }
                    if (i - st > kMaxIter) st = i - kMaxIter;
                    if (st - sb >= 64) { sb = st & ~63; XS = (sb + lane < n) ? X[sb + lane] : 0; }
                    if (lane == a) st_a = st;
                }
                const int st_rel = st_a - i0;
                int32_t best = NEG, best_j = -1, nok = 0;    // helpers start below any score; q_span is the main wave's floor
                const int st_lo = __builtin_amdgcn_readfirstlane(st_a);
                // far predecessors j <= i0 - 65 (final since block t - 1), chunks of 64 dealt round-robin to the helpers
                for (int jb = i0 - 65 - 64 * (wave - 1); jb >= st_lo; jb -= 64 * H) {
                    const int jl = jb - lane;
                    Pred pv = {0, 0, 0};
                    int32_t vs = 0;
                    if (jl >= st_lo) {
                        pv.x = X[jl];
                        const uint64_t yy = Y[jl];
                        pv.y = (uint32_t)yy; pv.sid = (int32_t)(yy >> 48 & 0xff);
                        vs = __hip_atomic_load(&S[jl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const int cnt = jb - st_lo + 1 < 64 ? jb - st_lo + 1 : 64;
                    const int jrel0 = jb - i0;
                    // (four steps per trip by hand: the evaluations are independent, only the running maximum is a chain)
                    auto fold_far = [&](int32_t sc, bool ok, int l) {         // descending j: strict > keeps the newest of equal scores
                        // every lane evaluates every predecessor and the result is SELECTED: left to itself the compiler
                        // branches around the arithmetic of filtered lanes (exec-mask juggling and four branches per
                        // step cost more issue slots than they save)
                        asm volatile("" : "+v"(sc));
                        const int jrel = jrel0 - l;
                        const bool okk = ok & mine & (jrel >= st_rel);
                        nok += okk ? 1 : 0;
                        const bool up = okk & (sc > best);
                        best = up ? sc : best; best_j = up ? jrel : best_j;
                    };
                    auto far_chunk = [&](auto tag) {
                        auto step = [&](int l) {
                            bool ok;
                            fold_far(geom(tag, xa, qa, qsa, sida, pv, l, ok) + __builtin_amdgcn_readlane(vs, l), ok, l);
                        };
                        int l = 0;
                        for (; l + 3 < cnt; l += 4) { step(l); step(l + 1); step(l + 2); step(l + 3); }
                        for (; l < cnt; l++) step(l);
                    };
                    // plain calls: four pairs per trip with their four gap-table reads in flight together (chain_geometry_plain_pre);
                    // MSEG (n_segs > 1: the dr > max_dist_y rule) is a compile-time tag, most calls have one segment
                    // a chunk whose oldest predecessor lies inside EVERY anchor's window needs no window test per pair (the pointer
                    // never moves back: the last anchor's start is the largest)
                    const bool inside = nb == 64 && jb - cnt + 1 >= i0 + __builtin_amdgcn_readlane(st_rel, 63);
                    auto fold_far_in = [&](int32_t sc, bool ok, int l) {
                        asm volatile("" : "+v"(sc));
                        nok += ok ? 1 : 0;
                        const bool up = ok & (sc > best);
                        best = up ? sc : best; best_j = up ? jrel0 - l : best_j;
                    };
                    auto far_chunk_tab = [&](auto mseg_tag) {
                        constexpr bool MSEG = decltype(mseg_tag)::value;
                        auto pre = [&](int l, uint32_t &idx, bool &ok) -> int32_t {
                            const uint32_t xj_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pv.x, l);
                            const uint32_t yj = (uint32_t)__builtin_amdgcn_readlane((int)pv.y, l);
                            return chain_geometry_plain_pre<MSEG>((uint32_t)xa, qa, qsa, xj_lo, yj, mdy, dq_lim, bw, idx, ok);
                        };
                        int l = 0;
                        if (inside) {
                            for (; l + 3 < cnt; l += 4) {
                                uint32_t i0_, i1_, i2_, i3_; bool o0, o1, o2, o3;
                                const int32_t c0 = pre(l, i0_, o0), c1 = pre(l + 1, i1_, o1), c2 = pre(l + 2, i2_, o2), c3 = pre(l + 3, i3_, o3);
                                const int32_t g0 = gap_tab[i0_], g1 = gap_tab[i1_], g2 = gap_tab[i2_], g3 = gap_tab[i3_];
                                fold_far_in(c0 - g0 + __builtin_amdgcn_readlane(vs, l), o0, l);
                                fold_far_in(c1 - g1 + __builtin_amdgcn_readlane(vs, l + 1), o1, l + 1);
                                fold_far_in(c2 - g2 + __builtin_amdgcn_readlane(vs, l + 2), o2, l + 2);
                                fold_far_in(c3 - g3 + __builtin_amdgcn_readlane(vs, l + 3), o3, l + 3);
                            }
                        }
                        for (; l + 3 < cnt; l += 4) {
                            uint32_t i0_, i1_, i2_, i3_; bool o0, o1, o2, o3;
                            const int32_t c0 = pre(l, i0_, o0), c1 = pre(l + 1, i1_, o1), c2 = pre(l + 2, i2_, o2), c3 = pre(l + 3, i3_, o3);
                            const int32_t g0 = gap_tab[i0_], g1 = gap_tab[i1_], g2 = gap_tab[i2_], g3 = gap_tab[i3_];
                            fold_far(c0 - g0 + __builtin_amdgcn_readlane(vs, l), o0, l);
                            fold_far(c1 - g1 + __builtin_amdgcn_readlane(vs, l + 1), o1, l + 1);
                            fold_far(c2 - g2 + __builtin_amdgcn_readlane(vs, l + 2), o2, l + 2);
                            fold_far(c3 - g3 + __builtin_amdgcn_readlane(vs, l + 3), o3, l + 3);
                        }
                        for (; l < cnt; l++) {
                            uint32_t ix; bool o;
                            const int32_t c = pre(l, ix, o);
                            fold_far(c - gap_tab[ix] + __builtin_amdgcn_readlane(vs, l), o, l);
                        }
                    };
                    if (!plain) far_chunk(std::false_type{});
                    else if (multi_seg) far_chunk_tab(std::true_type{});
                    else far_chunk_tab(std::false_type{});
                }
                part_best[par][wave - 1][lane] = best; part_j[par][wave - 1][lane] = best_j; part_ok[par][wave - 1][lane] = nok;
                if (wave == 1) part_st[par][lane] = st_rel;
            }
        } else if (t >= 0) {
            // ------------------------------------------------ main wave: block t
            const int i0 = t * 64;
            const int nb = n - i0 < 64 ? n - i0 : 64;
            const bool mine = lane < nb;
            const uint64_t xa = mine ? X[i0 + lane] : 0, ya = mine ? Y[i0 + lane] : 0;
            const int32_t qa = (int32_t)ya, qsa = (int32_t)(ya >> 32 & 0xff), sida = (int32_t)(ya >> 48 & 0xff);
            const Pred cur = {xa, (uint32_t)ya, sida};
            const int st_rel = part_st[par ^ 1][lane];
            if (mine) evals += (unsigned long long)(lane - st_rel);
            // State per lane: thr = the score a predecessor has to REACH to become the argmax -- q_span + 1 while nothing has
            // improved on q_span (the reference's strict >), the best score itself afterwards (a newer predecessor wins a tie) --
            // and best_j (relative to i0; kNoJ = none yet).  The anchor's score is thr, or thr - 1 = q_span while best_j is kNoJ.
            constexpr int kNoJ = (int)0x80000000;
            int32_t thr = qsa + 1, best_j = kNoJ;
            int32_t risk = 0;                                // unfiltered predecessors folded after the current argmax (upper bound)
            // helpers' partial maxima: chunks interleave, so the larger j wins a tie; all far unfiltered ones count as risk
#pragma unroll
            for (int hh = 0; hh < H; hh++) {
                const int32_t b2 = part_best[par ^ 1][hh][lane], j2 = part_j[par ^ 1][hh][lane];
                risk += part_ok[par ^ 1][hh][lane];
                if (b2 >= thr && !(best_j != kNoJ && b2 == thr && j2 < best_j)) { thr = b2; best_j = j2; }
            }
            auto fold = [&](int32_t sc, bool ok, int jrel) {
                asm volatile("" : "+v"(sc));                 // (selected, not branched around: see the helpers' step)
                const bool up = ok & (sc >= thr);
                risk = up ? 0 : risk + (ok ? 1 : 0);
                thr = up ? sc : thr; best_j = up ? jrel : best_j;
            };
            auto score_of = [&](int b) { return __builtin_amdgcn_readlane(thr, b) - (__builtin_amdgcn_readlane(best_j, b) == kNoJ ? 1 : 0); };
            // near predecessors: the previous block, OLDEST first (a newer one wins a tie), so that `risk` counts what is newer
            auto near_fold = [&](auto tag) {
                auto step = [&](int l) {
                    bool ok;
                    const int32_t sc = geom(tag, xa, qa, qsa, sida, prev, l, ok) + __builtin_amdgcn_readlane(pbest, l);
                    fold(sc, ok & mine & (l - 64 >= st_rel), l - 64);
                };
                int l = 0;
                for (; l + 3 < pnb; l += 4) { step(l); step(l + 1); step(l + 2); step(l + 3); }
                for (; l < pnb; l++) step(l);
            };
            // plain calls: the same steps with the four gap-table reads of a trip in flight together (see the helpers' far_chunk_tab)
            auto geom_pre = [&](auto mseg_tag, const Pred &pv, int src, uint32_t &idx, bool &ok) -> int32_t {
                constexpr bool MSEG = decltype(mseg_tag)::value;
                const uint32_t xj_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pv.x, src);
                const uint32_t yj = (uint32_t)__builtin_amdgcn_readlane((int)pv.y, src);
                return chain_geometry_plain_pre<MSEG>((uint32_t)xa, qa, qsa, xj_lo, yj, mdy, dq_lim, bw, idx, ok);
            };
            // the near / in-block predecessors lie inside every anchor's window when the LAST anchor's does (the pointer never moves
            // back): then the per-pair window tests go (most blocks: windows hold ~200 predecessors)
            const int st_last = __builtin_amdgcn_readlane(st_rel, 63);
            const bool near_in = nb == 64 && st_last <= -64, block_in = nb == 64 && st_last <= 0;
            auto near_fold_tab = [&](auto mseg_tag) {
                int l = 0;
                if (near_in) {
                    for (; l + 3 < pnb; l += 4) {
                        uint32_t i0_, i1_, i2_, i3_; bool o0, o1, o2, o3;
                        const int32_t c0 = geom_pre(mseg_tag, prev, l, i0_, o0), c1 = geom_pre(mseg_tag, prev, l + 1, i1_, o1);
                        const int32_t c2 = geom_pre(mseg_tag, prev, l + 2, i2_, o2), c3 = geom_pre(mseg_tag, prev, l + 3, i3_, o3);
                        const int32_t g0 = gap_tab[i0_], g1 = gap_tab[i1_], g2 = gap_tab[i2_], g3 = gap_tab[i3_];
                        fold(c0 - g0 + __builtin_amdgcn_readlane(pbest, l), o0, l - 64);
                        fold(c1 - g1 + __builtin_amdgcn_readlane(pbest, l + 1), o1, l - 63);
                        fold(c2 - g2 + __builtin_amdgcn_readlane(pbest, l + 2), o2, l - 62);
                        fold(c3 - g3 + __builtin_amdgcn_readlane(pbest, l + 3), o3, l - 61);
                    }
                }
                for (; l + 3 < pnb; l += 4) {
                    uint32_t i0_, i1_, i2_, i3_; bool o0, o1, o2, o3;
                    const int32_t c0 = geom_pre(mseg_tag, prev, l, i0_, o0), c1 = geom_pre(mseg_tag, prev, l + 1, i1_, o1);
                    const int32_t c2 = geom_pre(mseg_tag, prev, l + 2, i2_, o2), c3 = geom_pre(mseg_tag, prev, l + 3, i3_, o3);
                    const int32_t g0 = gap_tab[i0_], g1 = gap_tab[i1_], g2 = gap_tab[i2_], g3 = gap_tab[i3_];
                    fold(c0 - g0 + __builtin_amdgcn_readlane(pbest, l), o0 & mine & (l - 64 >= st_rel), l - 64);
                    fold(c1 - g1 + __builtin_amdgcn_readlane(pbest, l + 1), o1 & mine & (l - 63 >= st_rel), l - 63);
                    fold(c2 - g2 + __builtin_amdgcn_readlane(pbest, l + 2), o2 & mine & (l - 62 >= st_rel), l - 62);
                    fold(c3 - g3 + __builtin_amdgcn_readlane(pbest, l + 3), o3 & mine & (l - 61 >= st_rel), l - 61);
                }
                for (; l < pnb; l++) {
                    uint32_t ix; bool o;
                    const int32_t c = geom_pre(mseg_tag, prev, l, ix, o);
                    fold(c - gap_tab[ix] + __builtin_amdgcn_readlane(pbest, l), o & mine & (l - 64 >= st_rel), l - 64);
                }
            };
            if (!plain) near_fold(std::false_type{});
            else if (multi_seg) near_fold_tab(std::true_type{});
            else near_fold_tab(std::false_type{});
            // predecessors inside the block: anchor b is final once 0 .. b-1 are folded; if its certificate fails it is
            // re-done exactly before its score is broadcast.  The geometry of (a, b) involves no score and is computed four
            // steps ahead, so the dependent part of a step is readlane(score, b) + add + compare + select.
            auto finalize = [&](int b) {
                const int rb = __builtin_amdgcn_readlane(risk, b);
                if (rb > kMaxSkip && __builtin_amdgcn_readlane(best_j, b) != kNoJ) {     // wave-uniform: the reference's own scan of anchor i0 + b
                    if (mine && lane < b) { S[i0 + lane] = thr - (best_j == kNoJ ? 1 : 0); P[i0 + lane] = best_j == kNoJ ? -1 : i0 + best_j; }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    int32_t eb, ej;
                    chain_exact_global(X, Y, S, P, GM, i0 + b, i0 + __builtin_amdgcn_readlane(st_rel, b), mdx, mdy, bw, multi_seg, avg_d, eb, ej, evals);
                    if (lane == b) { thr = ej >= 0 ? eb : eb + 1; best_j = ej >= 0 ? ej - i0 : kNoJ; risk = 0; }
                }
            };
            auto block_fold = [&](auto tag) {
                auto g = [&](int b, bool &ok) -> int32_t {
                    const int32_t v = geom(tag, xa, qa, qsa, sida, cur, b, ok);
                    ok = ok & mine & (lane > b) & (b >= st_rel);
                    return v;
                };
                int b = 0;
                for (; b + 4 < nb; b += 4) {
                    bool o0, o1, o2, o3;
                    const int32_t g0 = g(b, o0), g1 = g(b + 1, o1), g2 = g(b + 2, o2), g3 = g(b + 3, o3);
                    finalize(b);     fold(g0 + score_of(b), o0, b);
                    finalize(b + 1); fold(g1 + score_of(b + 1), o1, b + 1);
                    finalize(b + 2); fold(g2 + score_of(b + 2), o2, b + 2);
                    finalize(b + 3); fold(g3 + score_of(b + 3), o3, b + 3);
                }
                for (; b < nb; b++) {
                    finalize(b);
                    if (b + 1 < nb) { bool o; const int32_t gg = g(b, o); fold(gg + score_of(b), o, b); }
                }
            };
            auto block_fold_tab = [&](auto mseg_tag) {
                auto gp = [&](int b, uint32_t &idx, bool &ok) -> int32_t {
                    const int32_t v = geom_pre(mseg_tag, cur, b, idx, ok);
                    ok = ok & mine & (lane > b) & (b >= st_rel);
                    return v;
                };
                auto gp_in = [&](int b, uint32_t &idx, bool &ok) -> int32_t {
                    const int32_t v = geom_pre(mseg_tag, cur, b, idx, ok);
                    ok = ok & (lane > b);
                    return v;
                };
                int b = 0;
                if (block_in) {
                    for (; b + 4 < nb; b += 4) {
                        uint32_t i0_, i1_, i2_, i3_; bool o0, o1, o2, o3;
                        const int32_t c0 = gp_in(b, i0_, o0), c1 = gp_in(b + 1, i1_, o1), c2 = gp_in(b + 2, i2_, o2), c3 = gp_in(b + 3, i3_, o3);
                        const int32_t g0 = c0 - gap_tab[i0_], g1 = c1 - gap_tab[i1_], g2 = c2 - gap_tab[i2_], g3 = c3 - gap_tab[i3_];
                        finalize(b);     fold(g0 + score_of(b), o0, b);
                        finalize(b + 1); fold(g1 + score_of(b + 1), o1, b + 1);
                        finalize(b + 2); fold(g2 + score_of(b + 2), o2, b + 2);
                        finalize(b + 3); fold(g3 + score_of(b + 3), o3, b + 3);
                    }
                }
                for (; b + 4 < nb; b += 4) {
                    uint32_t i0_, i1_, i2_, i3_; bool o0, o1, o2, o3;
                    const int32_t c0 = gp(b, i0_, o0), c1 = gp(b + 1, i1_, o1), c2 = gp(b + 2, i2_, o2), c3 = gp(b + 3, i3_, o3);
                    const int32_t g0 = c0 - gap_tab[i0_], g1 = c1 - gap_tab[i1_], g2 = c2 - gap_tab[i2_], g3 = c3 - gap_tab[i3_];
                    finalize(b);     fold(g0 + score_of(b), o0, b);
                    finalize(b + 1); fold(g1 + score_of(b + 1), o1, b + 1);
                    finalize(b + 2); fold(g2 + score_of(b + 2), o2, b + 2);
                    finalize(b + 3); fold(g3 + score_of(b + 3), o3, b + 3);
                }
                for (; b < nb; b++) {
                    finalize(b);
                    if (b + 1 < nb) { uint32_t ix; bool o; const int32_t c = gp(b, ix, o); fold(c - gap_tab[ix] + score_of(b), o, b); }
                }
            };
            if (!plain) block_fold(std::false_type{});
            else if (multi_seg) block_fold_tab(std::true_type{});
            else block_fold_tab(std::false_type{});
            const int32_t best = thr - (best_j == kNoJ ? 1 : 0);
            if (mine) {
                const int32_t par_ = best_j == kNoJ ? -1 : i0 + best_j;
                S[i0 + lane] = best; P[i0 + lane] = par_;
                if (THROUGH && feed.host_score) { feed.host_score[w.hoff + i0 + lane] = best; feed.host_parent[w.hoff + i0 + lane] = par_; }
            }
            prev = cur; pbest = best; pnb = nb;
        }
        __syncthreads();       // results of block t are acknowledged by L2; partial maxima of block t+1 are in LDS
    }
    for (int o = 32; o > 0; o >>= 1) evals += __shfl_xor(evals, o);
    if (lane == 0 && evals) atomicAdd(evals_out, evals);
    if (FED && feed.dbg && threadIdx.x == 0) feed.dbg[3 * item + 2] = wall_clock64();
}

// Throughput form (three helpers, big batches): six waves per SIMD = six calls per CU -- 80 VGPRs and 84 B of scratch instead
// of 99 and none; the kernel is bound by resident calls x per-call latency and a sixth call per CU is worth +5 %.
// Latency form (batches bound by their longest call): no register cap, the spills would lengthen the critical path (3-5 %).
template <int H, bool FED, bool THROUGH = FED>
__global__ __launch_bounds__(64 * (1 + H)) __attribute__((amdgpu_waves_per_eu(6, 6)))
void chain_block_kernel(const ChainWork *__restrict__ work, typename AnchorPtr<FED>::type xs, typename AnchorPtr<FED>::type ys, int32_t *score_out,
                        int32_t *parent_out, int32_t *gmarks_all, unsigned long long *evals_out, ChainFeed feed) {
    chain_block_body<H, FED, THROUGH>(work, xs, ys, score_out, parent_out, gmarks_all, evals_out, feed);
}
template <int H, bool FED>
__global__ __launch_bounds__(64 * (1 + H))
void chain_block_kernel_lat(const ChainWork *__restrict__ work, typename AnchorPtr<FED>::type xs, typename AnchorPtr<FED>::type ys, int32_t *score_out,
                            int32_t *parent_out, int32_t *gmarks_all, unsigned long long *evals_out, ChainFeed feed) {
    chain_block_body<H, FED>(work, xs, ys, score_out, parent_out, gmarks_all, evals_out, feed);
}

// ---- chain: the LATENCY form of the block kernel ("fast") --------------------------------------------------------------------
// chain_block_kernel is built for throughput: six calls per CU, four waves per call.  Alone on the chip a call still takes
// 0.26 us per anchor with it (60 000 anchors: 16 ms -- the floor under a 1 000-call batch and under every shard of a
// strong-scaling run), and knock-outs (profiles/r03_chain_latency.md) show where that goes: 0.15 us is the
// sequential window-start search (one ballot round per ANCHOR, repeated by every helper wave), 0.09 us the far fold riding
// on the same wave; the main wave's own 128 predecessor steps per block are not the critical path, they only make it longer.
// Here a call gets sixteen waves with one job each:
//   * SEARCH wave: the window starts of a whole block at once.  All 64 anchors scan the candidates j = st, st + 1, ...
//     together (x[j] is wave-uniform: one v_readlane per candidate) and count their leading passes -- the exact first index
//     at which anchor a's `while` would stop if it started at the block's entry value; since the pointer only moves forward the
//     sequential result is the running maximum of those (and of i - max_iter), PROVIDED that an anchor whose own stop lies
//     before its predecessor's also stops AT the predecessor's -- true whenever x ascends, verified per block by one gather,
//     and a block that fails it (unsorted x) takes the reference's anchor-by-anchor loop.  ~80 candidates per block instead of
//     64 dependent ballot rounds.
//   * WORKER waves (14): everything about a pair (anchor, predecessor) that involves no score of the current or the previous
//     block.  Eight of them leave the geometry of the 64 + 63 near / in-block pairs in LDS, G[pred][anchor] =
//     (oc - gc) << 7 | code (INT_MIN when the pair is filtered; code 1 .. 64 = the previous block's anchors, 65 + b = anchor b
//     of the block itself) plus a bit per unfiltered pair; all of them fold the far predecessors (final scores) in units of 16.
//   * MAIN wave: one key per anchor, score << 7 | code of its best predecessor so far (initially q_span << 7 | 127; the far
//     maximum enters with code 0): a step is key = max(key, G + (score of the predecessor << 7)) -- v_readlane, v_add, v_max.
//     The larger code wins a tie, i.e. the NEWER predecessor, as the reference's scan from i - 1 downwards with a strict >
//     does; a predecessor that merely equals q_span loses to the initial key's code 127, as `sc > max_f` demands.
//   * max_skip: the certificate of chain_block_body -- at most 25 unfiltered predecessors newer than the argmax -- is
//     checked AFTER the block from the bit masks (popcounts) instead of a counter in every step; a block in which any anchor
//     misses it is done again by the legacy code (per-anchor check, exact re-scan through chain_exact_global): on the suite's
//     inputs that is no block at all, on the adversarial ones a few per cent.
// Needs: the call's facts (one segment id, 32-bit-exact x, gap table: chain_facts_kernel), 0 <= avg_qspan <= 4096 and at
// most 65 793 anchors, so that a score (<= 255 n) fits the 24 bits above the code; other calls take the legacy code inside
// the same kernel.  LDS: 64 KB of G (two blocks in flight) + ~32 KB: one call per CU.
constexpr int kFastW = 14;                // worker waves
constexpr int kFastMaxN = 65000;
constexpr int32_t kFastSane = (1 << 24) - (1 << 15);     // scores the keys can hold with a block's growth to spare (255 * kFastMaxN is below it)
constexpr size_t kFastDynLds = 2 * 2 * 16 * 64 * 16;     // G[buf][near | block][pred / 4][anchor][4]

// MODE = GAB_CHAIN: as described above.  MODE = GAB_FASTCHAIN: the same machinery for fast-chain's rules (fastchain_body) --
// truncated 32-bit coordinates with wrap-around, fp32-floor gap cost from the call's table, `x[i] - x[st] > max_dist_x` as the
// window test, no max_skip (so no certificate).  The legacy code of the block takes over where the keys cannot express the
// reference: blocks in which some anchor has a window of <= 6 predecessors (the double-precision gap cost of the AVX code's
// scalar tail), a pair whose |dr - dq| wrapped to INT_MIN, scores that left the 24 bits (only reachable through such wraps).
template <int MODE>
__global__ __launch_bounds__(64 * (2 + kFastW))
void chain_fast_kernel(const ChainWork *__restrict__ work, const uint64_t *__restrict__ xs, const uint64_t *__restrict__ ys, int32_t *score_out,
                       int32_t *parent_out, int32_t *gmarks_all, unsigned long long *evals_out, const uint32_t *gate) {
    constexpr int NW = kFastW;
    // gate: the launch behind the table form (chain_tab.hip) -- only the calls it handed back (bail word set) are run here
    if (gate && __hip_atomic_load(&gate[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    constexpr bool FC = MODE == GAB_FASTCHAIN;
    extern __shared__ __attribute__((aligned(16))) uint8_t fast_lds[];
    int4 *G4 = reinterpret_cast<int4 *>(fast_lds);                                    // [buf][kind][16][64]
    __shared__ int32_t part_best[2][NW][64], part_j[2][NW][64], part_ok[2][NW][64], part_st[3][64];
    __shared__ uint16_t okh[2][8][64];                                                // unfiltered-pair bits, 16 pairs per G unit
    __shared__ int32_t weird[2];                                                      // fast-chain: a G pair wrapped (see above)
    __shared__ int32_t gap_tab[kGapTab];
    const ChainWork w = work[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const AnchorView<false> X{xs + w.off}, Y{ys + w.off};
    int32_t *S = score_out + w.off, *P = parent_out + w.off, *GM = FC ? nullptr : gmarks_all + w.off;
    const int n = (int)w.n;
    const int32_t mdx = w.max_dist_x, mdy = w.max_dist_y, bw = w.bw;
    const uint64_t mdx64 = (uint64_t)(int64_t)mdx;
    const double avg_d = (double)w.avg_qspan;
    const float k32 = (float)(0.01 * (double)w.avg_qspan);
    const bool multi_seg = w.n_segs > 1;
    const bool plain = !FC && (w.pad & 1) != 0;              // chain: chain_facts_kernel's verdict
    const bool use_tab = FC ? (bw >= 0 && bw <= kGapTab - 2) : plain;
    const int32_t mq = mdy < mdx ? mdy : mdx;
    const uint32_t dq_lim = mq < 0 ? 0u : (uint32_t)mq;
    const bool fast = use_tab && n <= kFastMaxN && w.avg_qspan >= 0.f && w.avg_qspan <= 4096.f && (!FC || dq_lim <= (1u << 20));
    const int nblocks = (n + 63) / 64;
    const int NEG = (int)0x80000000;
    if (use_tab) {
        for (int d = threadIdx.x; d <= bw + 1; d += 64 * (2 + NW)) {
            if (FC) {
                const int32_t dv = d <= bw ? d : NEG;
                gap_tab[d] = (int32_t)floorf(__fmul_rn((float)dv, k32)) + (15 - (__clz((int)((uint32_t)dv | 1u)) >> 1));
            } else gap_tab[d] = chain_gap_cost(d, avg_d);
        }
    }
    if (threadIdx.x < 2) weird[threadIdx.x] = 0;

    // ---- the score-independent part of a pair (anchor in this lane, predecessor in lane `src` of pv): oc - gc, and whether
    // the pair passes the filters.  TAG: chain -- the call's facts hold (plain arithmetic + gap table); fast-chain -- the block
    // needs the arithmetic gap cost (narrow windows, no table)
    struct Pred { uint64_t x; uint32_t y; int32_t sid; };
    struct Anchor { uint64_t x; int32_t q, qs, sid; bool wide; };
    auto pair = [&](auto tag, const Anchor &A, const Pred &pv, int src, bool &ok, bool &wrapped) -> int32_t {
        constexpr bool TAG = decltype(tag)::value;
        const uint32_t xj_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pv.x, src);
        const uint32_t yj = (uint32_t)__builtin_amdgcn_readlane((int)pv.y, src);
        wrapped = false;
        if constexpr (FC) {                                  // fastchain_body's score_pred with sj = 0
            const int32_t ddr = (int32_t)((uint32_t)A.x - xj_lo);
            const int32_t ddq = (int32_t)((uint32_t)A.q - yj);
            const int32_t diff = (int32_t)((uint32_t)ddr - (uint32_t)ddq);
            const int32_t dd = max(diff, (int32_t)(0u - (uint32_t)diff));
            ok = !(dd > bw || ddr == 0 || (uint32_t)ddq - 1u >= dq_lim);
            wrapped = dd < 0;
            const int32_t oc = min(min(ddr, ddq), A.qs);
            int32_t gc;
            if constexpr (TAG) {
                const int32_t lgh = 15 - (__clz((int)((uint32_t)dd | 1u)) >> 1);
                gc = (int32_t)floorf(__fmul_rn((float)dd, k32)) + lgh;
                const int32_t gd = (int32_t)__dmul_rn(__dmul_rn((double)dd, .01), avg_d) + lgh;
                gc = A.wide ? gc : gd;
            } else gc = gap_tab[min((uint32_t)dd, (uint32_t)bw + 1u)];
            return (int32_t)((uint32_t)oc - (uint32_t)gc);
        } else if constexpr (TAG) return chain_geometry_plain((uint32_t)A.x, A.q, A.qs, xj_lo, yj, mdy, dq_lim, bw, multi_seg, gap_tab, ok);
        else {
            const uint64_t xj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pv.x >> 32), src) << 32) | xj_lo;
            return chain_geometry(A.x, A.q, A.qs, A.sid, xj, yj, __builtin_amdgcn_readlane(pv.sid, src), mdx, mdy, bw, multi_seg, avg_d, ok);
        }
    };
    // the table variants in two halves (see chain_geometry_plain_pre): overlap term + table index now, gap_tab[idx] by the caller
    auto pair_pre = [&](auto mseg_tag, const Anchor &A, const Pred &pv, int src, uint32_t &idx, bool &ok, bool &wrapped) -> int32_t {
        constexpr bool MSEG = decltype(mseg_tag)::value;
        const uint32_t xj_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pv.x, src);
        const uint32_t yj = (uint32_t)__builtin_amdgcn_readlane((int)pv.y, src);
        wrapped = false;
        if constexpr (FC) {
            const int32_t ddr = (int32_t)((uint32_t)A.x - xj_lo);
            const int32_t ddq = (int32_t)((uint32_t)A.q - yj);
            const int32_t diff = (int32_t)((uint32_t)ddr - (uint32_t)ddq);
            const int32_t dd = max(diff, (int32_t)(0u - (uint32_t)diff));
            ok = !(dd > bw || ddr == 0 || (uint32_t)ddq - 1u >= dq_lim);
            wrapped = dd < 0;
            idx = min((uint32_t)dd, (uint32_t)bw + 1u);
            return min(min(ddr, ddq), A.qs);
        } else return chain_geometry_plain_pre<MSEG>((uint32_t)A.x, A.q, A.qs, xj_lo, yj, mdy, dq_lim, bw, idx, ok);
    };
    auto load_pred = [&](int i, bool valid) {
        Pred pv = {0, 0, 0};
        if (valid) { pv.x = X[i]; const uint64_t yy = Y[i]; pv.y = (uint32_t)yy; pv.sid = (int32_t)(yy >> 48 & 0xff); }
        return pv;
    };
    auto anchor_of = [&](uint64_t xa, uint64_t ya, bool wide) { return Anchor{xa, (int32_t)ya, (int32_t)(ya >> 32 & 0xff), (int32_t)(ya >> 48 & 0xff), wide}; };
    // the window test of the start search: chain host_kernel.cpp:56-57, fast-chain host_kernel.cpp:200-207 (unsigned difference)
    auto beyond = [&](uint64_t xi, uint64_t xj) { return FC ? (xi - xj) > mdx64 : xi > xj + mdx64; };

    // ---- the search wave's state: st = the pointer after the last anchor searched; XS = x[sb + lane], sb a multiple of 64
    int st = 0, sb = 0;
    uint64_t XS = (wave == 1 && lane < n) ? X[lane] : 0;
    uint64_t XSn = (wave == 1 && 64 + lane < n) ? X[64 + lane] : 0;   // the chunk behind XS, requested when XS was taken
    uint64_t sxa = XS;                                                  // x of the anchors of the next block to search, requested a block early
    // the reference's loop, anchor by anchor: the fall-back of search_block
    auto search_block_seq = [&](int kb, uint64_t xa) {
        const int i0 = kb * 64;
        const int nb = n - i0 < 64 ? n - i0 : 64;
        int st_a = 0;
        for (int a = 0; a < nb; a++) {
            const int i = i0 + a;
            const uint64_t xi = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(xa >> 32), a) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)xa, a);
            for (;;) {
                const int cand = sb + lane;
                const bool pass = cand < st || (cand < i && beyond(xi, XS));
                const unsigned long long m = __ballot(pass);
                if (m == ~0ull) { sb += 64; st = sb; XS = (sb + lane < n) ? X[sb + lane] : 0; continue; }
                st = sb + __builtin_ctzll(~m);
                break;
            }
            if (i - st > kMaxIter) st = i - kMaxIter;
            if (st - sb >= 64) { sb = st & ~63; XS = (sb + lane < n) ? X[sb + lane] : 0; }
            if (lane == a) st_a = st;
        }
        return st_a;
    };
    auto search_block = [&](int kb) {
        const int i0 = kb * 64;
        const int nb = n - i0 < 64 ? n - i0 : 64;
        const bool mine = lane < nb;
        const int ia = i0 + lane;
        const uint64_t xa = mine ? sxa : 0;
        sxa = ia + 64 < n ? X[ia + 64] : 0;
#ifdef GAB_KO_SEARCH
        { part_st[kb % 3][lane] = ia > 200 ? ia - 200 : 0; return; }
#endif
        const int S0 = st, sb0 = sb;
        const uint64_t XS0 = XS;
        // leading passes of every anchor over the candidates S0, S0 + 1, ... (all of an anchor's candidates are below its own index)
        bool alive = mine;
        int cnt = 0;
        int j = S0;
        const int jend = i0 + nb;
        while (j < jend && __ballot(alive)) {
            if ((j & ~63) != sb) {
                const bool next = (j & ~63) == sb + 64;
                sb = j & ~63;
                XS = next ? XSn : (sb + lane < n) ? X[sb + lane] : 0;
                XSn = (sb + 64 + lane < n) ? X[sb + 64 + lane] : 0;
            }
            const int lim = sb + 64 < jend ? sb + 64 : jend;
            const uint32_t xlo = (uint32_t)XS, xhi = (uint32_t)(XS >> 32);
            auto step = [&](int jj) {
                const uint64_t xj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)xhi, jj - sb) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)xlo, jj - sb);
                const bool pass = (jj < ia) & beyond(xa, xj);
                alive = alive & pass;
                cnt += alive ? 1 : 0;
            };
            for (; j + 4 <= lim; j += 4) { step(j); step(j + 1); step(j + 2); step(j + 3); if (!__ballot(alive)) break; }
            if (j + 4 > lim) for (; j < lim; j++) step(j);
        }
        const int g = S0 + cnt;
        const int c = mine ? max(g, ia - kMaxIter) : NEG;
        const int M = max(S0, wave_incl_max(c));
        const int Mprev = lane == 0 ? S0 : wave_shr1(M, S0);
        // an anchor that would stop BEFORE the pointer it inherits must stop AT it (true when x ascends)
        bool bad = false;
        if (mine && g < Mprev) bad = Mprev < ia && beyond(xa, X[Mprev]);
        int st_a = M;
        if (__ballot(bad)) {                                  // unsorted x: the reference's own loop from the block's entry state
            st = S0; sb = sb0; XS = XS0;
            st_a = search_block_seq(kb, xa);
            XSn = (sb + 64 + lane < n) ? X[sb + 64 + lane] : 0;
        } else {
            st = __builtin_amdgcn_readlane(M, nb - 1);
        }
        part_st[kb % 3][lane] = st_a;
    };
    if (wave == 1 && nblocks > 0) search_block(0);
    __syncthreads();

    int32_t pbest = 0;                                       // main: scores of the previous block
    uint64_t mnx = (wave == 0 && lane < n) ? X[lane] : 0, mny = (wave == 0 && lane < n) ? Y[lane] : 0, mpx = 0, mpy = 0;    // ... its anchors, a block early
    bool sane = true;
    unsigned long long evals = 0;
    // workers: the block's own x / y arrive one phase early (they are input, not results) and stay one phase longer as the
    // previous block's: no wait for memory at the top of a phase except for the far predecessors' scores
    uint64_t nxa = (wave >= 2 && lane < n) ? X[lane] : 0, nya = (wave >= 2 && lane < n) ? Y[lane] : 0, pxa = 0, pya = 0;

    for (int t = -1; t < nblocks; t++) {
        const int par = (t + 1) & 1;                         // slot of block t + 1 in the two-deep LDS arrays; block t lives in par ^ 1
        if (wave == 1) {
            if (t + 2 < nblocks) search_block(t + 2);        // two blocks ahead of the main wave, one ahead of the workers
        } else if (wave >= 2) {
            // ------------------------------------------------ workers: block t + 1
            const int kb = t + 1, wk = wave - 2;
            if (kb < nblocks) {
                const int i0 = kb * 64;
                const int nb = n - i0 < 64 ? n - i0 : 64;
                const bool mine = lane < nb;
                const uint64_t xa = nxa, ya = nya;
                {
                    const bool more = i0 + 64 + lane < n;
                    nxa = more ? X[i0 + 64 + lane] : 0; nya = more ? Y[i0 + 64 + lane] : 0;
                }
                const int st_a = part_st[kb % 3][lane];
                const int st_rel = st_a - i0;
                const int st_lo = __builtin_amdgcn_readfirstlane(st_a);
                const int st_hi = __builtin_amdgcn_readlane(st_a, nb - 1);      // (the pointer never moves back: the last anchor's start is the largest)
                const bool wide_a = !((lane - 1) - st_rel <= 5);
                const bool arith = FC && (__ballot(mine && !wide_a) != 0 || !use_tab);     // fast-chain: this block takes the arithmetic gap cost
                const Anchor A = anchor_of(xa, ya, wide_a);
                // the first far unit's predecessors are requested before the G unit is computed
                int fu = (wk - 8 + NW) % NW;
                auto far_load = [&](int fu_, Pred &pv, int32_t &vs) {
                    const int jl = i0 - 65 - 16 * fu_ - lane;
                    pv = Pred{0, 0, 0}; vs = 0;
                    if (lane < 16 && jl >= st_lo) {
                        pv = load_pred(jl, true);
                        vs = __hip_atomic_load(&S[jl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                };
                Pred fpv; int32_t fvs;
                far_load(fu, fpv, fvs);
#ifdef GAB_KO_G
                if (false) {
#else
                if (fast && !arith && wk < 8) {
#endif
                    // G of 16 predecessors: units 0..3 the previous block's anchors, 4..7 the block's own
                    const bool nearu = wk < 4;
                    const int p0 = (wk & 3) * 16;
                    const Pred pv = nearu ? Pred{pxa, (uint32_t)pya, (int32_t)(pya >> 48 & 0xff)} : Pred{xa, (uint32_t)ya, A.sid};
                    uint32_t bits = 0;
                    bool any_wrapped = false;
                    int4 *dst = G4 + ((size_t)(par * 2 + (nearu ? 0 : 1)) * 16 + (p0 >> 2)) * 64 + lane;
                    // (predecessors inside every anchor's window -- the last anchor's start is the largest -- need no window test)
                    const bool g_in = nb == 64 && (nearu ? kb > 0 && st_hi <= i0 - 64 : st_hi <= i0);
                    auto g_unit = [&](auto mseg_tag, auto in_tag) {
                        constexpr bool IN = decltype(in_tag)::value;
#pragma unroll
                        for (int g4 = 0; g4 < 4; g4++) {
                            int32_t oc[4], gv[4]; uint32_t ix[4]; bool okv[4];
#pragma unroll
                            for (int k = 0; k < 4; k++) {                 // four pairs, then their four table reads together
                                bool wr;
                                oc[k] = pair_pre(mseg_tag, A, pv, p0 + 4 * g4 + k, ix[k], okv[k], wr);
                                any_wrapped |= okv[k] & wr;
                            }
#pragma unroll
                            for (int k = 0; k < 4; k++) gv[k] = gap_tab[ix[k]];
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const int p = p0 + 4 * g4 + k;
                                const int jrel = nearu ? p - 64 : p;
                                const bool ok = IN ? (okv[k] & (nearu || lane > p)) : (okv[k] & mine & (jrel >= st_rel) & (nearu ? kb > 0 : lane > p));
                                const int code = nearu ? p + 1 : 65 + p;
                                const uint32_t v = (uint32_t)oc[k] - (uint32_t)gv[k];
                                gv[k] = ok ? (int32_t)((v << 7) | (uint32_t)code) : NEG;
                                if (!FC) bits |= ok ? (1u << (4 * g4 + k)) : 0u;     // (the certificate's masks: chain only)
                            }
                            dst[(size_t)g4 * 64] = make_int4(gv[0], gv[1], gv[2], gv[3]);
                        }
                    };
                    if (g_in) { if (!FC && multi_seg) g_unit(std::true_type{}, std::true_type{}); else g_unit(std::false_type{}, std::true_type{}); }
                    else { if (!FC && multi_seg) g_unit(std::true_type{}, std::false_type{}); else g_unit(std::false_type{}, std::false_type{}); }
                    if (!FC) okh[par][wk][lane] = (uint16_t)bits;
                    if (FC && __ballot(any_wrapped) && lane == 0) weird[par] = 1;
                }
                int32_t best = NEG, best_j = -1, nok = 0;
                // far predecessors j <= i0 - 65 (final since block t - 1) in units of 16, dealt round-robin starting with the
                // workers that have no G unit
#ifdef GAB_KO_FAR
                for (; false; fu += NW) {
#else
                for (; i0 - 65 - 16 * fu >= st_lo; fu += NW) {
#endif
                    const int jb = i0 - 65 - 16 * fu;
                    const Pred pv = fpv;
                    const int32_t vs = fvs;
                    if (i0 - 65 - 16 * (fu + NW) >= st_lo) far_load(fu + NW, fpv, fvs);      // the next unit's, under this one's arithmetic
                    const int cnt = jb - st_lo + 1 < 16 ? jb - st_lo + 1 : 16;
                    const int jrel0 = jb - i0;
                    // a unit whose oldest predecessor is inside EVERY anchor's window (all but the few units around the window starts)
                    // needs no window test per pair
                    const bool inside = jb - cnt + 1 >= st_hi && nb == 64;
                    auto fold_far = [&](int32_t sc, bool ok, int l) {
                        asm volatile("" : "+v"(sc));
                        const int jrel = jrel0 - l;
                        const bool okk = ok & mine & (jrel >= st_rel);
                        if (!FC) nok += okk ? 1 : 0;
                        const bool up = okk & (sc > best);
                        best = up ? sc : best; best_j = up ? jrel : best_j;
                    };
                    auto fold_far_in = [&](int32_t sc, bool ok, int l) {
                        asm volatile("" : "+v"(sc));
                        if (!FC) nok += ok ? 1 : 0;
                        const bool up = ok & (sc > best);
                        best = up ? sc : best; best_j = up ? jrel0 - l : best_j;
                    };
                    auto far_unit = [&](auto tag) {                       // the arithmetic variants (generic chain calls, fast-chain's narrow blocks)
                        for (int l = 0; l < cnt; l++) {
                            bool ok, wr;
                            fold_far((int32_t)((uint32_t)pair(tag, A, pv, l, ok, wr) + (uint32_t)__builtin_amdgcn_readlane(vs, l)), ok, l);
                        }
                    };
                    // the table variants: four pairs per trip, their four table reads in flight together
                    auto far_unit_tab = [&](auto mseg_tag) {
                        auto pre = [&](int l, uint32_t &idx, bool &ok) -> int32_t { bool wr; return pair_pre(mseg_tag, A, pv, l, idx, ok, wr); };
                        int l = 0;
                        if (inside) {                         // (cnt == 16 then)
                            for (; l + 3 < cnt; l += 4) {
                                uint32_t i0_, i1_, i2_, i3_; bool o0, o1, o2, o3;
                                const int32_t c0 = pre(l, i0_, o0), c1 = pre(l + 1, i1_, o1), c2 = pre(l + 2, i2_, o2), c3 = pre(l + 3, i3_, o3);
                                const int32_t g0 = gap_tab[i0_], g1 = gap_tab[i1_], g2 = gap_tab[i2_], g3 = gap_tab[i3_];
                                fold_far_in((int32_t)((uint32_t)c0 - (uint32_t)g0 + (uint32_t)__builtin_amdgcn_readlane(vs, l)), o0, l);
                                fold_far_in((int32_t)((uint32_t)c1 - (uint32_t)g1 + (uint32_t)__builtin_amdgcn_readlane(vs, l + 1)), o1, l + 1);
                                fold_far_in((int32_t)((uint32_t)c2 - (uint32_t)g2 + (uint32_t)__builtin_amdgcn_readlane(vs, l + 2)), o2, l + 2);
                                fold_far_in((int32_t)((uint32_t)c3 - (uint32_t)g3 + (uint32_t)__builtin_amdgcn_readlane(vs, l + 3)), o3, l + 3);
                            }
                        }
                        for (; l + 3 < cnt; l += 4) {
                            uint32_t i0_, i1_, i2_, i3_; bool o0, o1, o2, o3;
                            const int32_t c0 = pre(l, i0_, o0), c1 = pre(l + 1, i1_, o1), c2 = pre(l + 2, i2_, o2), c3 = pre(l + 3, i3_, o3);
                            const int32_t g0 = gap_tab[i0_], g1 = gap_tab[i1_], g2 = gap_tab[i2_], g3 = gap_tab[i3_];
                            fold_far((int32_t)((uint32_t)c0 - (uint32_t)g0 + (uint32_t)__builtin_amdgcn_readlane(vs, l)), o0, l);
                            fold_far((int32_t)((uint32_t)c1 - (uint32_t)g1 + (uint32_t)__builtin_amdgcn_readlane(vs, l + 1)), o1, l + 1);
                            fold_far((int32_t)((uint32_t)c2 - (uint32_t)g2 + (uint32_t)__builtin_amdgcn_readlane(vs, l + 2)), o2, l + 2);
                            fold_far((int32_t)((uint32_t)c3 - (uint32_t)g3 + (uint32_t)__builtin_amdgcn_readlane(vs, l + 3)), o3, l + 3);
                        }
                        for (; l < cnt; l++) {
                            uint32_t ix; bool o;
                            const int32_t c = pre(l, ix, o);
                            fold_far((int32_t)((uint32_t)c - (uint32_t)gap_tab[ix] + (uint32_t)__builtin_amdgcn_readlane(vs, l)), o, l);
                        }
                    };
                    if (FC) { if (arith) far_unit(std::true_type{}); else far_unit_tab(std::false_type{}); }
                    else if (!plain) far_unit(std::false_type{});
                    else if (multi_seg) far_unit_tab(std::true_type{});
                    else far_unit_tab(std::false_type{});
                }
                part_best[par][wk][lane] = best; part_j[par][wk][lane] = best_j; part_ok[par][wk][lane] = nok;
                pxa = xa; pya = ya;
            }
        } else if (t >= 0) {
            // ------------------------------------------------ main wave: block t
            const int i0 = t * 64;
            const int nb = n - i0 < 64 ? n - i0 : 64;
            const bool mine = lane < nb;
            const int st_rel = part_st[t % 3][lane] - i0;
            if (mine) evals += (unsigned long long)(lane - st_rel);
            constexpr int kNoJ = (int)0x80000000;
            const uint64_t xa64 = mnx, ya64 = mny;
            {
                const bool more = i0 + 64 + lane < n;
                mnx = more ? X[i0 + 64 + lane] : 0; mny = more ? Y[i0 + 64 + lane] : 0;
            }
            const int32_t qsa = mine ? (int32_t)(ya64 >> 32 & 0xff) : 0;
            const bool wide_a = !((lane - 1) - st_rel <= 5);
            const bool arith = FC && (__ballot(mine && !wide_a) != 0 || !use_tab);
            // the workers' far maxima: units interleave, so the larger j wins a tie; thr0 = the score to reach, j0 = argmax
            // (relative to i0) or none; chain: all far unfiltered ones count as risk
            int32_t thr0 = qsa + 1, j0 = kNoJ, risk0 = 0;
#pragma unroll
            for (int hh = 0; hh < NW; hh++) {
                const int32_t b2 = part_best[par ^ 1][hh][lane], j2 = part_j[par ^ 1][hh][lane];
                if (!FC) risk0 += part_ok[par ^ 1][hh][lane];
                if (b2 >= thr0 && !(j0 != kNoJ && b2 == thr0 && j2 < j0)) { thr0 = b2; j0 = j2; }
            }
            // scores the keys cannot hold: only ever produced by fast-chain's wrapped arithmetic; the call stays with the legacy code then
            sane = sane && __ballot(mine && ((uint32_t)pbest >= (uint32_t)kFastSane || (j0 != kNoJ && (uint32_t)thr0 >= (uint32_t)kFastSane))) == 0;
            int32_t best = 0, best_jrel = kNoJ;
            bool redo = !fast || arith || !sane || (FC && weird[par ^ 1] != 0);
#ifdef GAB_KO_MAIN
            if (false) {
#else
            if (!redo) {
#endif
                const int32_t initkey = (qsa << 7) | 127;
                int32_t key = j0 != kNoJ ? max(initkey, thr0 << 7) : initkey;
                const int4 *gn = G4 + ((size_t)((par ^ 1) * 2 + 0) * 16) * 64 + lane;
                const int4 *gb = G4 + ((size_t)((par ^ 1) * 2 + 1) * 16) * 64 + lane;
                const int32_t pkey = pbest << 7;
#pragma unroll
                for (int g4 = 0; g4 < 16; g4++) {            // the previous block: no dependence between the steps
                    const int4 g = gn[(size_t)g4 * 64];
                    key = max(key, g.x + __builtin_amdgcn_readlane(pkey, 4 * g4));
                    key = max(key, g.y + __builtin_amdgcn_readlane(pkey, 4 * g4 + 1));
                    key = max(key, g.z + __builtin_amdgcn_readlane(pkey, 4 * g4 + 2));
                    key = max(key, g.w + __builtin_amdgcn_readlane(pkey, 4 * g4 + 3));
                }
#pragma unroll
                for (int g4 = 0; g4 < 16; g4++) {            // the block itself: anchor b is final once 0 .. b - 1 are folded
                    const int4 g = gb[(size_t)g4 * 64];
                    key = max(key, g.x + (__builtin_amdgcn_readlane(key, 4 * g4) & ~127));
                    key = max(key, g.y + (__builtin_amdgcn_readlane(key, 4 * g4 + 1) & ~127));
                    key = max(key, g.z + (__builtin_amdgcn_readlane(key, 4 * g4 + 2) & ~127));
                    if (g4 < 15) key = max(key, g.w + (__builtin_amdgcn_readlane(key, 4 * g4 + 3) & ~127));
                }
                const bool none = key == initkey;
                const int code = key & 127;
                best = key >> 7;
                best_jrel = none ? kNoJ : code ? code - 65 : j0;
                if (!FC) {
                    // the certificate: unfiltered pairs newer than the argmax -- pair numbers code .. 126 (pair c = code - 1 is the
                    // argmax); for a far argmax all of them plus every far one
                    int32_t risk = code ? 0 : risk0;
#pragma unroll
                    for (int wd = 0; wd < 4; wd++) {
                        const uint32_t bits = (uint32_t)okh[par ^ 1][2 * wd][lane] | ((uint32_t)okh[par ^ 1][2 * wd + 1][lane] << 16);
                        const uint32_t m = code <= 32 * wd ? ~0u : code >= 32 * wd + 32 ? 0u : (~0u << (code - 32 * wd));
                        risk += __popc(bits & m);
                    }
                    redo = __ballot(mine && !none && risk > kMaxSkip) != 0;
                }
#ifdef GAB_KO_G
                redo = false;                                  // (timing experiment: the G tables are garbage)
#endif
            }
            if (redo) {
                // the legacy block: geometry in this wave; chain: the certificate per anchor, the reference's own scan on a miss
                const Pred cur = {xa64, (uint32_t)ya64, (int32_t)(ya64 >> 48 & 0xff)}, prev = {mpx, (uint32_t)mpy, (int32_t)(mpy >> 48 & 0xff)};
                const Anchor A = anchor_of(xa64, ya64, wide_a);
                const int pnb = t > 0 ? 64 : 0;
                int32_t thr = thr0, best_j = j0, risk = risk0;
                auto fold = [&](int32_t sc, bool ok, int jrel) {
                    asm volatile("" : "+v"(sc));
                    const bool up = ok & (sc >= thr);
                    risk = up ? 0 : risk + (ok ? 1 : 0);
                    thr = up ? sc : thr; best_j = up ? jrel : best_j;
                };
                auto score_of = [&](int b) { return __builtin_amdgcn_readlane(thr, b) - (__builtin_amdgcn_readlane(best_j, b) == kNoJ ? 1 : 0); };
                auto near_fold = [&](auto tag) {
                    for (int l = 0; l < pnb; l++) {                      // oldest first: a newer predecessor wins a tie (>=)
                        bool ok, wr;
                        const int32_t sc = (int32_t)((uint32_t)pair(tag, A, prev, l, ok, wr) + (uint32_t)__builtin_amdgcn_readlane(pbest, l));
                        fold(sc, ok & mine & (l - 64 >= st_rel), l - 64);
                    }
                };
                if (FC ? arith : plain) near_fold(std::true_type{}); else near_fold(std::false_type{});
                auto finalize = [&](int b) {
                    if constexpr (!FC) {
                        const int rb = __builtin_amdgcn_readlane(risk, b);
                        if (rb > kMaxSkip && __builtin_amdgcn_readlane(best_j, b) != kNoJ) {
                            if (mine && lane < b) { S[i0 + lane] = thr - (best_j == kNoJ ? 1 : 0); P[i0 + lane] = best_j == kNoJ ? -1 : i0 + best_j; }
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            int32_t eb, ej;
                            chain_exact_global(X, Y, S, P, GM, i0 + b, i0 + __builtin_amdgcn_readlane(st_rel, b), mdx, mdy, bw, multi_seg, avg_d, eb, ej, evals);
                            if (lane == b) { thr = ej >= 0 ? eb : eb + 1; best_j = ej >= 0 ? ej - i0 : kNoJ; risk = 0; }
                        }
                    }
                };
                auto block_fold = [&](auto tag) {
                    for (int b = 0; b < nb; b++) {
                        finalize(b);
                        if (b + 1 < nb) {
                            bool ok, wr;
                            const int32_t gg = pair(tag, A, cur, b, ok, wr);
                            fold((int32_t)((uint32_t)gg + (uint32_t)score_of(b)), ok & mine & (lane > b) & (b >= st_rel), b);
                        }
                    }
                };
                if (FC ? arith : plain) block_fold(std::true_type{}); else block_fold(std::false_type{});
                best = thr - (best_j == kNoJ ? 1 : 0);
                best_jrel = best_j;
            }
#ifndef GAB_KO_STORE
            if (mine) { S[i0 + lane] = best; P[i0 + lane] = best_jrel == kNoJ ? -1 : i0 + best_jrel; }
#endif
            pbest = best;
            mpx = xa64; mpy = ya64;
            if (FC && lane == 0) weird[par ^ 1] = 0;          // (the workers set the other slot in this phase)
        }
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) evals += __shfl_xor(evals, o);
    if (lane == 0 && evals) atomicAdd(evals_out, evals);
}

}  // namespace

// =============================================================================== host side
struct gab_chain {
    gab_tuning tun = gab_tuning_loaded();      // experiment knobs, read when the handle is made
    gab_host_stream hs;     // private stream of the host-pointer entry point(s)
    int device = 0;
    gab_devbuf work;       // ChainWork[ncalls] + evals counter
    gab_devbuf gmarks;     // chain mode: targets[] for windows deeper than the LDS mark ring (one int32 per anchor)
    gab_devbuf io;         // staging for the host-pointer entry point
    hipEvent_t ev[2] = {nullptr, nullptr};
    // the host-pointer entry point of big batches: two more streams and the events that order its copies and kernels
    hipStream_t xs[2] = {nullptr, nullptr};
    hipStream_t gs = nullptr;                 // fed path: the gather's stream, confined to the CUs of one XCD
    bool gs_tried = false;
    int gs_blocks = 256;
    hipStream_t fs = nullptr;                // the latency-form launch of the longest calls runs beside the throughput launch of the rest
    hipEvent_t fe[2] = {nullptr, nullptr};
    hipEvent_t xe[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned long long *h_evals = nullptr;   // pinned: evals, (spare), abort word of the fed path
    uint8_t *h_started = nullptr;            // pinned: one byte per workgroup of chain_gather_kernel
    hipEvent_t xe_fed = nullptr;
    bool have_stats = false;
    ChainTab tab;                            // the table form's buffers (chain_tab.hip)
    hipStream_t ts = nullptr;                // ... and its stream: it runs beside the other two forms
    hipEvent_t te[2] = {nullptr, nullptr};
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
        hipHostMalloc((void **)&h->h_evals, 64) != hipSuccess ||
        hipEventCreateWithFlags(&h->xe_fed, hipEventDisableTiming) != hipSuccess) {
        gab_set_error("gab_chain_create: event / pinned allocation failed"); delete h; return GAB_EDEVICE;
    }
    *out = h;
    return GAB_OK;
}

extern "C" void gab_chain_destroy(gab_chain *h) {
    if (!h) return;
    gab_device_guard g(h->device);
    h->work.release(); h->io.release(); h->hs.release(); h->gmarks.release(); h->tab.release();
    if (h->ts) (void)hipStreamDestroy(h->ts);
    for (int k = 0; k < 2; k++) if (h->te[k]) (void)hipEventDestroy(h->te[k]);
    for (int k = 0; k < 2; k++) if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
    for (int k = 0; k < 2; k++) if (h->xs[k]) (void)hipStreamDestroy(h->xs[k]);
    if (h->gs) (void)hipStreamDestroy(h->gs);
    if (h->fs) (void)hipStreamDestroy(h->fs);
    for (int k = 0; k < 2; k++) if (h->fe[k]) (void)hipEventDestroy(h->fe[k]);
    for (int k = 0; k < 5; k++) if (h->xe[k]) (void)hipEventDestroy(h->xe[k]);
    if (h->h_evals) (void)hipHostFree(h->h_evals);
    if (h->h_started) (void)hipHostFree(h->h_started);
    if (h->xe_fed) (void)hipEventDestroy(h->xe_fed);
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

// How many helper waves a batch gets.  Three is the throughput optimum (chain-large: 35.6 ms; five 51.8, seven 50.0: every
// helper repeats the window-start search of the block).  But a batch whose longest call alone outlasts everything else is
// bound by that call's blocks, and those finish sooner with more hands: 1 000 calls / 8 M anchors, longest 60 000 --
// chain 25.8 -> 21.7 -> 20.2 ms with 3 / 5 / 7 helpers, fast-chain 24.7 -> 21.4 -> 20.1 ms.  Seven are used when the batch,
// at the seven-helper throughput (1.6 G anchors/s), would be done in 0.6 of the longest call's time (0.34 us per anchor).
static int chain_helpers_for(const gab_tuning &tun, int64_t total_anchors, int64_t longest_call) {
    if (tun.chain_helpers == 3 || tun.chain_helpers == 5 || tun.chain_helpers == 7) return tun.chain_helpers;   // GAB_CHAIN_HELPERS: A/B runs
    return total_anchors <= 326 * longest_call ? 7 : 3;
}

// the kernels of one work list (already on the device) on `s`; nothing else (no memset, no synchronisation)
static void chain_launch(const gab_tuning &tun, int mode, int helpers, hipStream_t s, ChainWork *d_work, unsigned nw, const uint64_t *d_x, const uint64_t *d_y,
                         int32_t *d_score, int32_t *d_parent, int32_t *d_gm, unsigned long long *d_ev, const ChainFeed *feed_in = nullptr) {
    const ChainFeed feed = feed_in ? *feed_in : ChainFeed{nullptr, nullptr, nullptr, nullptr, nullptr, kFeedSpinLimit};
    if (nw == 0) return;
    if (mode == GAB_FASTCHAIN) {
        if (feed.facts) hipLaunchKernelGGL((fastchain_kernel<kFcHelpers, true>), dim3(nw), dim3(64 * (1 + kFcHelpers)), 0, s, d_work, d_x, d_y, d_score, d_parent, d_ev, feed);
        else if (helpers == 7) hipLaunchKernelGGL((fastchain_kernel_lat<7, false>), dim3(nw), dim3(64 * 8), 0, s, d_work, d_x, d_y, d_score, d_parent, d_ev, feed);
        else if (helpers == 5) hipLaunchKernelGGL((fastchain_kernel_lat<5, false>), dim3(nw), dim3(64 * 6), 0, s, d_work, d_x, d_y, d_score, d_parent, d_ev, feed);
        else if (feed.host_score) hipLaunchKernelGGL((fastchain_kernel<kFcHelpers, false, true>), dim3(nw), dim3(64 * (1 + kFcHelpers)), 0, s, d_work, d_x, d_y, d_score, d_parent, d_ev, feed);
        else hipLaunchKernelGGL((fastchain_kernel<kFcHelpers, false>), dim3(nw), dim3(64 * (1 + kFcHelpers)), 0, s, d_work, d_x, d_y, d_score, d_parent, d_ev, feed);
    } else if (tun.chain_walk)      // GAB_CHAIN_KERNEL=walk: the per-anchor walk (A/B runs)
        hipLaunchKernelGGL(chain_hw_kernel, dim3(nw), dim3(64 * (1 + kChHelpers)), 0, s, d_work, d_x, d_y, d_score, d_parent, d_gm, d_ev);
    else {
        if (feed.facts) {
            hipLaunchKernelGGL((chain_block_kernel<kCbHelpers, true>), dim3(nw), dim3(64 * (1 + kCbHelpers)), 0, s, d_work, d_x, d_y, d_score, d_parent, d_gm, d_ev, feed);
            return;
        }
        hipLaunchKernelGGL(chain_facts_kernel, dim3(nw), dim3(256), 0, s, d_work, d_x, d_y, (const uint32_t *)nullptr);
        if (helpers == 7) hipLaunchKernelGGL((chain_block_kernel_lat<7, false>), dim3(nw), dim3(64 * 8), 0, s, d_work, d_x, d_y, d_score, d_parent, d_gm, d_ev, feed);
        else if (helpers == 5) hipLaunchKernelGGL((chain_block_kernel_lat<5, false>), dim3(nw), dim3(64 * 6), 0, s, d_work, d_x, d_y, d_score, d_parent, d_gm, d_ev, feed);
        else if (feed.host_score) hipLaunchKernelGGL((chain_block_kernel<kCbHelpers, false, true>), dim3(nw), dim3(64 * (1 + kCbHelpers)), 0, s, d_work, d_x, d_y, d_score, d_parent, d_gm, d_ev, feed);
        else hipLaunchKernelGGL((chain_block_kernel<kCbHelpers, false>), dim3(nw), dim3(64 * (1 + kCbHelpers)), 0, s, d_work, d_x, d_y, d_score, d_parent, d_gm, d_ev, feed);
    }
}

static ChainWork chain_work_of(const gab_chain_hdr &hd, int64_t off) {
    ChainWork w;
    w.off = w.hoff = off; w.n = hd.n; w.avg_qspan = hd.avg_qspan;
    w.max_dist_x = hd.max_dist_x; w.max_dist_y = hd.max_dist_y; w.bw = hd.bw; w.n_segs = hd.n_segs; w.pad = 0;
    return w;
}

// stream, events and LDS attribute of the latency form (chain_fast_kernel); also made by gab_chain_reserve: a stream costs
// ~8 ms the first time
static int chain_fast_setup(gab_chain *h) {
    if (!h->ts) {
        if (hipStreamCreateWithFlags(&h->ts, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->te[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->te[1], hipEventDisableTiming) != hipSuccess) { gab_set_error("gab_chain: stream / event of the table form failed"); return GAB_EDEVICE; }
        const int rc = chain_tab_setup();
        if (rc) return rc;
    }
    if (h->fs) return GAB_OK;
    if (hipStreamCreateWithFlags(&h->fs, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->fe[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->fe[1], hipEventDisableTiming) != hipSuccess ||
        hipFuncSetAttribute((const void *)chain_fast_kernel<GAB_CHAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFastDynLds) != hipSuccess ||
        hipFuncSetAttribute((const void *)chain_fast_kernel<GAB_FASTCHAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFastDynLds) != hipSuccess) {
        gab_set_error("gab_chain: stream / event / LDS attribute of the latency-form kernel failed"); return GAB_EDEVICE;
    }
    return GAB_OK;
}

static int chain_run_device_impl(gab_chain *h, int mode, const uint64_t *d_x, const uint64_t *d_y, const int64_t *call_off, const gab_chain_hdr *hdr,
                                 int64_t ncalls, int32_t *d_score, int32_t *d_parent, void *stream_, int32_t *host_score, int32_t *host_parent);
extern "C" int gab_chain_run_device(gab_chain *h, int mode, const uint64_t *d_x, const uint64_t *d_y,
                                    const int64_t *call_off, const gab_chain_hdr *hdr, int64_t ncalls,
                                    int32_t *d_score, int32_t *d_parent, void *stream_) {
    return chain_run_device_impl(h, mode, d_x, d_y, call_off, hdr, ncalls, d_score, d_parent, stream_, nullptr, nullptr);
}
// gab_chain_run_device for a caller that wants the results in HOST memory as well (a driver whose read phase parsed the file on
// the GPU): the device arrays are filled as always -- the DP reads its predecessors' scores from them -- and the results also land
// in host_score / host_parent by the time the call returns.  When the whole batch runs in the throughput form and the host arrays
// are page-locked, every block of 64 results is written through to them by the DP kernel itself (the stores of the fed path,
// `THROUGH`): 0.68 GB of chain-large cross the bus UNDER the 28 ms of the DP instead of in 12 ms behind it; otherwise the two
// arrays are copied when the kernels are done.
extern "C" int gab_chain_run_device_through(gab_chain *h, int mode, const uint64_t *d_x, const uint64_t *d_y, const int64_t *call_off,
                                            const gab_chain_hdr *hdr, int64_t ncalls, int32_t *d_score, int32_t *d_parent,
                                            int32_t *host_score, int32_t *host_parent, void *stream_) {
    GAB_CHECK(host_score && host_parent, "gab_chain_run_device_through: NULL host buffer");
    return chain_run_device_impl(h, mode, d_x, d_y, call_off, hdr, ncalls, d_score, d_parent, stream_, host_score, host_parent);
}
static int chain_run_device_impl(gab_chain *h, int mode, const uint64_t *d_x, const uint64_t *d_y, const int64_t *call_off, const gab_chain_hdr *hdr,
                                 int64_t ncalls, int32_t *d_score, int32_t *d_parent, void *stream_, int32_t *host_score, int32_t *host_parent) {
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
    gab_tuning_refresh(&h->tun);
    hipStream_t s = (hipStream_t)stream_;

    // longest call first: the sequential walk of the biggest call is the critical path
    std::vector<ChainWork> wk;
    wk.reserve((size_t)ncalls);
    for (int64_t c = 0; c < ncalls; c++) {
        if (hdr[c].n == 0) continue;
        ChainWork w;
        w.off = w.hoff = call_off[c]; w.n = hdr[c].n; w.avg_qspan = hdr[c].avg_qspan;
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
    GAB_CHECK_ATOMIC64(o_ev);
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
    // The longest calls (the list is sorted) go to the latency form (chain_fast_kernel: one call per CU, sixteen waves), the
    // rest to the throughput form, side by side on two streams -- when the batch would otherwise wait for its longest call:
    // the throughput form takes ~0.30 us per anchor of a call however empty the chip is and does ~2.85 G anchors/s over all
    // calls; a batch whose longest call needs more than 0.75 of the batch's throughput time hands every call of 512 anchors or
    // more to the latency form (chain-large on one GPU: none; an eighth of it and the 1 000-call input: 99 % of their anchors --
    // leaving the mid-size calls in the throughput form beside it cost the slowest call 1-2 %: its CU's SIMDs are shared).
    // GAB_CHAIN_FAST_MIN / GAB_CHAIN_FAST_CALLS pin the choice (tests, A/B runs).
    // r04 -- and the calls a batch would WAIT for go to the table form (chain_tab.hip: the geometry of every block on any CU,
    // the call's own workgroup only folds): a 60 000-anchor call takes ~1 ms there instead of 5-10, at 2 bytes of HBM traffic per
    // pair.  The list is sorted longest first: [0, ntab) table form, [ntab, ntab + nfast) latency form, the rest throughput
    // form, side by side on three streams.  GAB_CHAIN_TAB_MIN pins the table form's smallest call, GAB_CHAIN_TAB=0 turns it off.
    const bool legacy_only = (mode == GAB_CHAIN && h->tun.chain_walk) || h->tun.chain_helpers_set;
    size_t ntab = 0, nfast = 0;
    int64_t tab_anchors = 0;
    bool written_through = false;          // gab_chain_run_device_through: the DP kernels themselves fill the host arrays
    int32_t *hs = nullptr, *hp = nullptr;  // ... device-visible addresses of the caller's page-locked arrays (nullptr: pageable, or not asked for)
    if (host_score) {
        void *a = nullptr, *b = nullptr;
        if (hipHostGetDevicePointer(&a, host_score, 0) == hipSuccess && hipHostGetDevicePointer(&b, host_parent, 0) == hipSuccess) { hs = (int32_t *)a; hp = (int32_t *)b; }
        else (void)hipGetLastError();
    }
    uint32_t *d_bail = nullptr;
    if (!legacy_only && h->tun.chain_tab != 0) {
        // A batch whose longest call would outlast 0.75 of the batch's throughput time in the throughput form (0.30 us per anchor of
        // a call, 2.85 G anchors/s over all calls: the latency-form rule below) hands its long calls to the table form: every call
        // that would take a quarter of that time there, 2048 anchors at least.  Measured (r04, one rank's share of chain-large
        // under 2 / 4 / 8-GPU strong scaling, smallest table call 2048 .. 35 000 anchors): the more calls in the table form the
        // better down to ~2048 anchors -- 8 GPUs 4.97-5.12 ms, 4 GPUs 7.80-8.29 ms, 2 GPUs 15.2-15.6 ms (17.0 without) -- while
        // chain-large on ONE GPU loses (27.8 -> 29.4 ms with the calls above 45 000 anchors there): the table form costs 2 bytes of
        // HBM traffic per pair and ~25 % more instructions, and only pays where calls are waited for.
        int64_t min_n = INT64_MAX;
        const double est_tp = (double)total / 2.85e9, lat_max = 0.30e-6 * (double)wk[0].n;
        if (lat_max >= 0.75 * est_tp) min_n = std::max<int64_t>(2048, (int64_t)(0.25 * est_tp / 0.30e-6));
        // fast-chain (no certificate, no exact re-scans; its geometry rows are batched): since the end of r04 the table form is also
        // the FASTER form for calls of a few thousand anchors and more, whatever the batch -- all of fast-chain-large on one GPU with
        // the calls of >= 40 000 / 25 000 / 12 500 / 4 096 / 2 048 / 512 / 1 anchors there: 27.87 / 27.09 / 26.15 / 25.88-25.95 / 26.16 / 26.48 /
        // 26.65 ms against 27.65 without.  chain: 29.2 / 28.4 / 27.9 / 28.2 against 27.8 -- it keeps the rule above.
        // (not when the results are also written through to host memory, gab_chain_run_device_through: the fold then produces 0.6 GB
        // of results in its 10 ms and is held to the link's ~40 GB/s -- 16.5 ms -- while the throughput form spreads the same bytes
        // over its 27 ms: the fast-chain driver's region of interest 32.3 ms against 29.8)
        if (mode == GAB_FASTCHAIN && !hs) min_n = std::min<int64_t>(min_n, 4096);
        // chain: with the fold of the end of the round (far maxima merged by LDS atomics, fourteen workers) its longest calls gain too, less:
        // all of chain-large on one GPU with the calls of >= 30 000 / 20 000 / 16 000 / 12 500 / 8 192 / 6 000 anchors there: 27.86 / 26.99 /
        // 26.83 / 27.01 / 27.23 / 27.13 ms against 27.82 without; and once the fold's loop existed per role: 16 000 / 10 000 / 6 000 / 4 096 /
        // 2 048 anchors 25.04 / 24.89 / 24.85 / 24.90 / 25.00 ms
        // (written through, the same holds as for fast-chain: the chain driver's region of interest 31.8 ms with the 8 192 rule, 31.2-32.2 ms
        // with a 16 000 rule -- the faster fold of the end of the round makes the burst of results shorter still --, 29.0-29.1 ms without)
        else if (mode == GAB_CHAIN && !hs) min_n = std::min<int64_t>(min_n, 8192);
        if (h->tun.chain_tab_min >= 0) min_n = h->tun.chain_tab_min;      // GAB_CHAIN_TAB_MIN
        while (ntab < nw && wk[ntab].n >= min_n) { tab_anchors += wk[ntab].n; ntab++; }
    }
    const size_t nrest = nw - ntab;
    if (!legacy_only && nrest) {
        int64_t min_n = 0, max_calls = 0;
        const double est_tp = (double)(total - tab_anchors) / 2.85e9, lat_max = 0.30e-6 * (double)wk[ntab].n;
        if (lat_max >= 0.75 * est_tp) { min_n = 512; max_calls = (int64_t)nrest; }
        if (h->tun.chain_fast_min >= 0) min_n = h->tun.chain_fast_min;              // GAB_CHAIN_FAST_MIN
        if (h->tun.chain_fast_calls >= 0) max_calls = h->tun.chain_fast_calls;      // GAB_CHAIN_FAST_CALLS
        if (h->tun.chain_fast_min >= 0 && h->tun.chain_fast_calls < 0) max_calls = (int64_t)nrest;
        while (nfast < nrest && (int64_t)nfast < max_calls && wk[ntab + nfast].n >= min_n) nfast++;
    }
    if (ntab || nfast) {
        if ((rc = chain_fast_setup(h)) != GAB_OK) return rc;
        const ChainFeed nofeed{nullptr, nullptr, nullptr, nullptr, nullptr, kFeedSpinLimit};
        // (the facts of the calls in the table form are only needed by the few it hands back: taken behind it, gated)
        if (mode == GAB_CHAIN && nrest) hipLaunchKernelGGL(chain_facts_kernel, dim3((unsigned)nrest), dim3(256), 0, s, d_work + ntab, d_x, d_y, (const uint32_t *)nullptr);
        GAB_HIP(hipEventRecord(h->fe[0], s));
        if (ntab) {
            // the table form and, behind it on the same stream, the latency form for the calls it hands back (bail word set)
            GAB_HIP(hipStreamWaitEvent(h->ts, h->fe[0], 0));
            if ((rc = chain_tab_run(&h->tab, h->tun, mode, h->ts, d_work, wk.data(), ntab, total, d_x, d_y, d_score, d_parent, d_gm, d_ev, &d_bail, nfast == 0 ? hs : nullptr, nfast == 0 ? hp : nullptr)) != GAB_OK) return rc;
            if (mode == GAB_CHAIN) hipLaunchKernelGGL(chain_facts_kernel, dim3((unsigned)ntab), dim3(256), 0, h->ts, d_work, d_x, d_y, (const uint32_t *)d_bail);
            if (mode == GAB_CHAIN)
                hipLaunchKernelGGL(chain_fast_kernel<GAB_CHAIN>, dim3((unsigned)ntab), dim3(64 * (2 + kFastW)), kFastDynLds, h->ts, d_work, d_x, d_y, d_score, d_parent, d_gm, d_ev, (const uint32_t *)d_bail);
            else
                hipLaunchKernelGGL(chain_fast_kernel<GAB_FASTCHAIN>, dim3((unsigned)ntab), dim3(64 * (2 + kFastW)), kFastDynLds, h->ts, d_work, d_x, d_y, d_score, d_parent, d_gm, d_ev, (const uint32_t *)d_bail);
            GAB_HIP(hipGetLastError());
            GAB_HIP(hipEventRecord(h->te[1], h->ts));
        }
        if (nfast) {
            GAB_HIP(hipStreamWaitEvent(h->fs, h->fe[0], 0));
            if (mode == GAB_CHAIN)
                hipLaunchKernelGGL(chain_fast_kernel<GAB_CHAIN>, dim3((unsigned)nfast), dim3(64 * (2 + kFastW)), kFastDynLds, h->fs, d_work + ntab, d_x, d_y, d_score, d_parent, d_gm, d_ev, (const uint32_t *)nullptr);
            else
                hipLaunchKernelGGL(chain_fast_kernel<GAB_FASTCHAIN>, dim3((unsigned)nfast), dim3(64 * (2 + kFastW)), kFastDynLds, h->fs, d_work + ntab, d_x, d_y, d_score, d_parent, d_gm, d_ev, (const uint32_t *)nullptr);
            GAB_HIP(hipGetLastError());
            GAB_HIP(hipEventRecord(h->fe[1], h->fs));
        }
        // (gab_chain_run_device_through: the table form and the throughput form write their results through; the latency form does
        // not, and neither do the calls the table form hands back -- then the arrays are copied at the end)
        written_through = hs != nullptr && nfast == 0;
        if (nrest > nfast) {
            const ChainFeed through{nullptr, hs, hp, nullptr, nullptr, kFeedSpinLimit};
            if (mode == GAB_CHAIN) {
                if (written_through)
                    hipLaunchKernelGGL((chain_block_kernel<kCbHelpers, false, true>), dim3((unsigned)(nrest - nfast)), dim3(64 * (1 + kCbHelpers)), 0, s, d_work + ntab + nfast, d_x, d_y,
                                       d_score, d_parent, d_gm, d_ev, through);
                else
                    hipLaunchKernelGGL((chain_block_kernel<kCbHelpers, false>), dim3((unsigned)(nrest - nfast)), dim3(64 * (1 + kCbHelpers)), 0, s, d_work + ntab + nfast, d_x, d_y,
                                       d_score, d_parent, d_gm, d_ev, nofeed);
            } else {
                if (written_through)
                    hipLaunchKernelGGL((fastchain_kernel<kFcHelpers, false, true>), dim3((unsigned)(nrest - nfast)), dim3(64 * (1 + kFcHelpers)), 0, s, d_work + ntab + nfast, d_x, d_y,
                                       d_score, d_parent, d_ev, through);
                else
                    hipLaunchKernelGGL((fastchain_kernel<kFcHelpers, false>), dim3((unsigned)(nrest - nfast)), dim3(64 * (1 + kFcHelpers)), 0, s, d_work + ntab + nfast, d_x, d_y,
                                       d_score, d_parent, d_ev, nofeed);
            }
        }
        if (nfast) GAB_HIP(hipStreamWaitEvent(s, h->fe[1], 0));
        if (ntab) GAB_HIP(hipStreamWaitEvent(s, h->te[1], 0));
    } else {
        const int helpers = chain_helpers_for(h->tun, total, wk.empty() ? 0 : wk[0].n);
        ChainFeed through{nullptr, nullptr, nullptr, nullptr, nullptr, kFeedSpinLimit};
        if (hs && helpers == 3 && !(mode == GAB_CHAIN && h->tun.chain_walk)) {
            through.host_score = hs; through.host_parent = hp;      // page-locked: the DP writes its results through
            written_through = true;
        }
        chain_launch(h->tun, mode, helpers, s, d_work, (unsigned)nw, d_x, d_y, d_score, d_parent, d_gm, d_ev, written_through ? &through : nullptr);
    }
    GAB_HIP(hipGetLastError());
    GAB_HIP(hipEventRecord(h->ev[1], s));
    GAB_HIP(hipMemcpyAsync(h->h_evals, d_ev, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    std::vector<uint32_t> h_bail;
    if (written_through && ntab) { h_bail.resize(ntab); GAB_HIP(hipMemcpyAsync(h_bail.data(), d_bail, 4 * ntab, hipMemcpyDeviceToHost, s)); }
    if (host_score && !written_through) {
        GAB_HIP(hipMemcpyAsync(host_score, d_score, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, s));
        GAB_HIP(hipMemcpyAsync(host_parent, d_parent, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, s));
    }
    GAB_HIP(hipStreamSynchronize(s));    // wk (host vector) must outlive the H2D copy
    if (written_through && ntab && std::any_of(h_bail.begin(), h_bail.end(), [](uint32_t b) { return b != 0; })) {
        // a call the table form handed back ran in the latency form, which writes to the device arrays only: copy after all
        GAB_HIP(hipMemcpyAsync(host_score, d_score, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, s));
        GAB_HIP(hipMemcpyAsync(host_parent, d_parent, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, s));
        GAB_HIP(hipStreamSynchronize(s));
    }
    if (ntab && h->tun.chain_trace) chain_tab_report(&h->tab, ntab);
    h->have_stats = true;
    return GAB_OK;
}

// gab_chain_run for a big batch.  A batch takes at least as long as the fold of its longest call (~27 ms for 60 000 anchors),
// and the copy of 16 B per anchor in and 8 B out takes about as long as all the kernels together, so plain
// copy-in / kernels / copy-out doubles the time.  Order of events instead (three streams, ordered by events):
//   A   the anchors of the LONGEST calls (~6 % of all) are copied first and their kernel starts at once: the critical path
//       begins ~3 ms into the call instead of after the whole copy-in;
//   B1  the first half of x / y follows, then the kernel of the remaining calls that lie in it, then -- once A is done too --
//       the first half of the results goes back;
//   B2  the second half is copied right behind the first (under B1's kernel), its kernel, its half of the results.
// (Cutting the batch into independent gab_chain_run calls instead would put a long call into every piece: measured
// 113 ms for 2 pieces, 272 ms for 8, against 74 ms unsplit.)
static int chain_run_overlapped(gab_chain *h, int mode, const uint64_t *x, const uint64_t *y, const int64_t *call_off,
                                const gab_chain_hdr *hdr, int64_t ncalls, int64_t total, int32_t *score_out, int32_t *parent_out,
                                hipStream_t sA) {
    h->have_stats = false;
    const size_t t = (size_t)total;
    char *b = h->io.as<char>();
    uint64_t *dx = (uint64_t *)b, *dy = (uint64_t *)(b + 8 * t);
    int32_t *ds = (int32_t *)(b + 16 * t), *dp = (int32_t *)(b + 20 * t);
    for (int k = 0; k < 2; k++)
        if (!h->xs[k] && hipStreamCreateWithFlags(&h->xs[k], hipStreamNonBlocking) != hipSuccess) { gab_set_error("gab_chain_run: stream creation failed"); return GAB_EDEVICE; }
    for (int k = 0; k < 5; k++)
        if (!h->xe[k] && hipEventCreateWithFlags(&h->xe[k], hipEventDisableTiming) != hipSuccess) { gab_set_error("gab_chain_run: event creation failed"); return GAB_EDEVICE; }
    hipStream_t sB1 = h->xs[0], sB2 = h->xs[1];
    const bool trace = h->tun.chain_trace;                          // GAB_CHAIN_TRACE, diagnosis: a time line of the three streams on stderr
    hipEvent_t tv[10] = {};
    if (trace) for (auto &e : tv) (void)hipEventCreate(&e);
    auto mark = [&](int k, hipStream_t st) { if (trace) (void)hipEventRecord(tv[k], st); };
    // group A: the longest calls, up to ~6 % of the anchors (at least 64, at most 512 calls)
    std::vector<int64_t> order((size_t)ncalls);
    for (int64_t c = 0; c < ncalls; c++) order[(size_t)c] = c;
    const size_t topn = (size_t)std::min<int64_t>(512, ncalls);
    std::partial_sort(order.begin(), order.begin() + topn, order.end(), [&](int64_t a, int64_t c) { return hdr[a].n > hdr[c].n; });
    std::vector<char> inA((size_t)ncalls, 0);
    std::vector<ChainWork> wk[3];
    int64_t accA = 0;
    for (size_t k = 0; k < topn; k++) {
        const int64_t c = order[k];
        if (hdr[c].n == 0 || (k >= 64 && accA * 16 >= total)) break;
        inA[(size_t)c] = 1; accA += hdr[c].n;
        wk[0].push_back(chain_work_of(hdr[c], call_off[c]));
    }
    // the anchor index that splits the arrays in two halves
    const int64_t mid = total / 2;
    for (int64_t c = 0; c < ncalls; c++) {
        if (inA[(size_t)c] || hdr[c].n == 0) continue;
        wk[call_off[c] + hdr[c].n <= mid ? 1 : 2].push_back(chain_work_of(hdr[c], call_off[c]));
    }
    for (int g = 1; g < 3; g++) std::stable_sort(wk[g].begin(), wk[g].end(), [](const ChainWork &a, const ChainWork &c) { return a.n > c.n; });
    const size_t nwt = wk[0].size() + wk[1].size() + wk[2].size();
    const size_t o_ev = (sizeof(ChainWork) * nwt + 15) & ~(size_t)15;
    int rc = h->work.reserve(o_ev + 16);
    if (rc) return rc;
    ChainWork *d_work[3];
    d_work[0] = h->work.as<ChainWork>(); d_work[1] = d_work[0] + wk[0].size(); d_work[2] = d_work[1] + wk[1].size();
    unsigned long long *d_ev = (unsigned long long *)(h->work.as<char>() + o_ev);
    int32_t *d_gm = nullptr;
    if (mode == GAB_CHAIN) {
        if ((rc = h->gmarks.reserve(sizeof(int32_t) * t)) != GAB_OK) return rc;
        d_gm = h->gmarks.as<int32_t>();
    }
    // ---- stream A
    GAB_HIP(hipEventRecord(h->ev[0], sA));
    mark(0, sA);
    GAB_HIP(hipMemsetAsync(d_ev, 0, 16, sA));
    if (d_gm) GAB_HIP(hipMemsetAsync(d_gm, 0, sizeof(int32_t) * t, sA));              // vector::resize zero-fills targets
    for (int g = 0; g < 3; g++)
        if (!wk[g].empty()) GAB_HIP(hipMemcpyAsync(d_work[g], wk[g].data(), sizeof(ChainWork) * wk[g].size(), hipMemcpyHostToDevice, sA));
    GAB_HIP(hipEventRecord(h->xe[0], sA));                                             // counters, marks and work lists are ready
    for (const ChainWork &w : wk[0]) {
        GAB_HIP(hipMemcpyAsync(dx + w.off, x + w.off, 8 * (size_t)w.n, hipMemcpyHostToDevice, sA));
        GAB_HIP(hipMemcpyAsync(dy + w.off, y + w.off, 8 * (size_t)w.n, hipMemcpyHostToDevice, sA));
    }
    mark(1, sA);
    chain_launch(h->tun, mode, 3, sA, d_work[0], (unsigned)wk[0].size(), dx, dy, ds, dp, d_gm, d_ev);
    mark(2, sA);
    GAB_HIP(hipEventRecord(h->xe[1], sA));                                             // A's results are final
    // ---- stream B1: first half
    GAB_HIP(hipStreamWaitEvent(sB1, h->xe[0], 0));
    GAB_HIP(hipMemcpyAsync(dx, x, 8 * (size_t)mid, hipMemcpyHostToDevice, sB1));
    GAB_HIP(hipMemcpyAsync(dy, y, 8 * (size_t)mid, hipMemcpyHostToDevice, sB1));
    GAB_HIP(hipEventRecord(h->xe[2], sB1));                                            // first half is in
    mark(3, sB1);
    chain_launch(h->tun, mode, 3, sB1, d_work[1], (unsigned)wk[1].size(), dx, dy, ds, dp, d_gm, d_ev);
    mark(4, sB1);
    GAB_HIP(hipEventRecord(h->xe[4], sB1));                                            // B1's results are final
    // ---- stream B2: second half right behind the first
    GAB_HIP(hipStreamWaitEvent(sB2, h->xe[2], 0));
    GAB_HIP(hipMemcpyAsync(dx + mid, x + mid, 8 * (t - (size_t)mid), hipMemcpyHostToDevice, sB2));
    GAB_HIP(hipMemcpyAsync(dy + mid, y + mid, 8 * (t - (size_t)mid), hipMemcpyHostToDevice, sB2));
    mark(5, sB2);
    chain_launch(h->tun, mode, 3, sB2, d_work[2], (unsigned)wk[2].size(), dx, dy, ds, dp, d_gm, d_ev);
    mark(6, sB2);
    GAB_HIP(hipGetLastError());
    // ---- results: each half once its own kernel and A's are done (a call of B2 may begin in the first half: its anchors
    // below `mid` are copied with the second batch of results, so the first copy stops at the start of the first such call)
    int64_t cut = mid;
    for (const ChainWork &w : wk[2]) cut = std::min<int64_t>(cut, w.off);
    GAB_HIP(hipStreamWaitEvent(sB1, h->xe[1], 0));
    if (cut > 0) {
        GAB_HIP(hipMemcpyAsync(score_out, ds, 4 * (size_t)cut, hipMemcpyDeviceToHost, sB1));
        GAB_HIP(hipMemcpyAsync(parent_out, dp, 4 * (size_t)cut, hipMemcpyDeviceToHost, sB1));
    }
    GAB_HIP(hipStreamWaitEvent(sB2, h->xe[1], 0));
    GAB_HIP(hipStreamWaitEvent(sB2, h->xe[4], 0));                                     // (calls of B1 that lie between `cut` and `mid`)
    GAB_HIP(hipMemcpyAsync(score_out + cut, ds + cut, 4 * (t - (size_t)cut), hipMemcpyDeviceToHost, sB2));
    GAB_HIP(hipMemcpyAsync(parent_out + cut, dp + cut, 4 * (t - (size_t)cut), hipMemcpyDeviceToHost, sB2));
    // ---- join on stream A
    mark(7, sB1); mark(8, sB2);
    GAB_HIP(hipEventRecord(h->xe[2], sB1));
    GAB_HIP(hipEventRecord(h->xe[3], sB2));
    GAB_HIP(hipStreamWaitEvent(sA, h->xe[2], 0));
    GAB_HIP(hipStreamWaitEvent(sA, h->xe[3], 0));
    GAB_HIP(hipEventRecord(h->ev[1], sA));
    GAB_HIP(hipMemcpyAsync(h->h_evals, d_ev, sizeof(unsigned long long), hipMemcpyDeviceToHost, sA));
    GAB_HIP(hipStreamSynchronize(sA));            // (the host work lists must outlive their copies)
    if (trace) {
        const char *nm[9] = {"start", "A copied", "A kernel done", "first half copied", "B1 kernel done", "second half copied", "B2 kernel done",
                             "first results back", "all results back"};
        fprintf(stderr, "[gab_chain_run] %zu + %zu + %zu calls, A = %lld anchors:", wk[0].size(), wk[1].size(), wk[2].size(), (long long)accA);
        for (int k = 1; k < 9; k++) { float ms = 0; (void)hipEventElapsedTime(&ms, tv[0], tv[k]); fprintf(stderr, "  %s %.1f", nm[k], ms); }
        fprintf(stderr, " ms\n");
        for (auto &e : tv) (void)hipEventDestroy(e);
    }
    h->have_stats = true;
    return GAB_OK;
}

// gab_chain_run for a big batch on page-locked arrays.  A call's workgroup can start as soon as ITS anchors are on the device,
// and the batch takes at least as long as its longest call, so the anchors should arrive longest call first -- an order the
// copy engines cannot follow (20 000 pieces of ~0.3 MB: a third of their rate) but a kernel can: chain_gather_kernel reads
// the caller's arrays over the bus in that order and publishes a word per call; ONE launch of the DP kernel (its workgroups
// are dispatched in the same order) waits per call; results are written through to the caller's arrays block by block.
// The DP kernel is launched only after every workgroup of the gather kernel has reported in (they are resident then and
// cannot be locked out by waiting workgroups), and a wait gives up after seconds, so the grid drains in every case.
// Returns GAB_EAGAIN-like 1 when the arrays are not device-accessible (the caller takes the copy-engine path).
// The streams and the page-locked flags of the fed path (chain_run_fed); also called by gab_chain_reserve, because making two
// streams costs 17 ms the first time (hipStreamCreateWithFlags 8.0 ms, hipExtStreamCreateWithCUMask 8.9 ms in the C driver's
// region of interest, which makes ONE call).
static int chain_fed_setup(gab_chain *h, int *gather_blocks) {
    *gather_blocks = 256;
    if (!h->xs[0] && hipStreamCreateWithFlags(&h->xs[0], hipStreamNonBlocking) != hipSuccess) { gab_set_error("gab_chain_run: stream creation failed"); return GAB_EDEVICE; }
    // The gather kernel keeps megabytes of reads from HOST memory in flight, ~3 us each, and a CU's vector memory pipeline
    // returns data in order: with a gather workgroup on every CU the DP ran at half of its speed until the last anchor had
    // arrived (a 60 000-anchor call that was there at t = 0 was done after 33 ms, against 14 ms with the bus idle; 0.65-0.73 ms
    // per 1000 anchors of a long call on every XCD against 0.33) -- overlapping the gather with the DP was worth 1.4 of 51 ms.
    // The gather therefore runs on a stream confined to ONE CU per XCD (a CU-mask stream: mask bit b is CU b / 8 of XCD
    // b % 8; 8 workgroups of 256 threads fill that CU and 64 of them keep the bus as busy as 256 did: 26.0 against 24.4 ms for
    // 1.36 GB), and the DP has the other 248 CUs to itself: DP done after 35.7 ms instead of 51 (profiles/r03_chain_fed.md).
    // $GAB_CHAIN_GATHER_MASK: "none" = the old form (256 workgroups anywhere), "N:M" = bits N .. N + M - 1 (experiments).
    if (!h->gs && !h->gs_tried) {
        h->gs_tried = true;
        int ncu = 0;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device);
        const char *e = h->tun.chain_gather_mask[0] ? h->tun.chain_gather_mask : nullptr;      // GAB_CHAIN_GATHER_MASK
        static std::atomic<int> next_cu{0};                  // every handle of the process its own CU (workers of one driver run side by side)
        int b0 = 8 * (next_cu.fetch_add(1) % 32), nb = 8;
        if (e && e[0] >= '0' && e[0] <= '9') { b0 = atoi(e); nb = strchr(e, ':') ? atoi(strchr(e, ':') + 1) : 8; }
        if (!(e && !strcmp(e, "none")) && ncu == 256 && b0 >= 0 && nb > 0 && b0 + nb <= ncu) {
            uint32_t mg[8] = {};
            for (int b = b0; b < b0 + nb; b++) mg[b >> 5] |= 1u << (b & 31);
            if (hipExtStreamCreateWithCUMask(&h->gs, 8, mg) != hipSuccess) { (void)hipGetLastError(); h->gs = nullptr; }
            else {
                // workgroups that are RESIDENT together on one of the mask's CUs: the DP is launched once all of them have reported
                // in, so a count the CUs cannot hold at once would cost every call the 20 ms wait below (the compiler's
                // register count decides, not the 2048 threads a CU can hold)
                int per_cu = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_gather_kernel, 256, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
                h->gs_blocks = std::min(256, std::min(8, per_cu) * nb);
            }
        }
    }
    if (h->gs) *gather_blocks = h->gs_blocks;
    if (h->tun.chain_gather_blocks >= 8 && h->tun.chain_gather_blocks <= 256) *gather_blocks = h->tun.chain_gather_blocks;      // GAB_CHAIN_GATHER_BLOCKS
    if (!h->h_started && hipHostMalloc((void **)&h->h_started, 256 + 64) != hipSuccess) { gab_set_error("gab_chain_run: pinned allocation failed"); return GAB_EDEVICE; }
    return GAB_OK;
}

static int chain_run_fed(gab_chain *h, int mode, const uint64_t *x, const uint64_t *y, const int64_t *call_off,
                         const gab_chain_hdr *hdr, int64_t ncalls, int64_t total, int32_t *score_out, int32_t *parent_out,
                         hipStream_t sA) {
    void *hx = nullptr, *hy = nullptr, *hs = nullptr, *hp = nullptr;
    if (hipHostGetDevicePointer(&hx, (void *)x, 0) != hipSuccess || hipHostGetDevicePointer(&hy, (void *)y, 0) != hipSuccess ||
        hipHostGetDevicePointer(&hs, (void *)score_out, 0) != hipSuccess || hipHostGetDevicePointer(&hp, (void *)parent_out, 0) != hipSuccess) {
        (void)hipGetLastError();
        return 1;
    }
    h->have_stats = false;
    int gather_blocks = 256;
    {
        const int rc0 = chain_fed_setup(h, &gather_blocks);
        if (rc0) return rc0;
    }
    hipStream_t sG = h->gs ? h->gs : h->xs[0];
    // work list, longest call first; on the device every call starts on a 128-byte line (16 anchors), so that no cache line is
    // shared between calls (see chain_feed_wait); and the chunk table
    std::vector<ChainWork> wk;
    wk.reserve((size_t)ncalls);
    for (int64_t c = 0; c < ncalls; c++) if (hdr[c].n) wk.push_back(chain_work_of(hdr[c], call_off[c]));
    std::stable_sort(wk.begin(), wk.end(), [](const ChainWork &a, const ChainWork &c) { return a.n > c.n; });
    const size_t nw = wk.size();
    std::vector<ChainChunk> chunks;
    std::vector<uint32_t> need(nw);
    chunks.reserve((size_t)total / kFeedChunk + nw);
    int64_t dtotal = 0;
    for (size_t k = 0; k < nw; k++) {
        wk[k].off = dtotal;
        dtotal += (wk[k].n + 15) & ~(int64_t)15;
        need[k] = (uint32_t)((wk[k].n + kFeedChunk - 1) / kFeedChunk);
        for (int64_t o = 0; o < wk[k].n; o += kFeedChunk)
            chunks.push_back(ChainChunk{wk[k].hoff + o, wk[k].off + o, (int32_t)std::min<int64_t>(kFeedChunk, wk[k].n - o), (int32_t)k});
    }
    const size_t t = (size_t)dtotal;
    int rc = h->io.reserve(24 * t + 64);
    if (rc) return rc;
    char *b = h->io.as<char>();
    uint64_t *dx = (uint64_t *)b, *dy = (uint64_t *)(b + 8 * t);
    int32_t *ds = (int32_t *)(b + 16 * t), *dp = (int32_t *)(b + 20 * t);
    GAB_CHECK(chunks.size() < (1ull << 32), "gab_chain_run: too many chunks");
    // device scratch: work | evals, abort | facts | done | need | mixed | xlo | xhi | chunks
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_ev = up(sizeof(ChainWork) * nw), o_facts = o_ev + 256, o_done = o_facts + up(4 * nw), o_need = o_done + up(4 * nw),
                 o_mixed = o_need + up(4 * nw), o_xlo = o_mixed + up(4 * nw), o_xhi = o_xlo + up(8 * nw), o_chunks = o_xhi + up(8 * nw);
    GAB_CHECK_ATOMIC64(o_ev); GAB_CHECK_ATOMIC64(o_xlo); GAB_CHECK_ATOMIC64(o_xhi);      // the evals counter, atomicMin / atomicMax per call
    if ((rc = h->work.reserve(o_chunks + sizeof(ChainChunk) * chunks.size())) != GAB_OK) return rc;
    char *wb = h->work.as<char>();
    ChainWork *d_work = (ChainWork *)wb;
    unsigned long long *d_ev = (unsigned long long *)(wb + o_ev);
    uint32_t *d_abort = (uint32_t *)(wb + o_ev + 16);
    int32_t *d_gm = nullptr;
    if (mode == GAB_CHAIN) {
        if ((rc = h->gmarks.reserve(sizeof(int32_t) * t)) != GAB_OK) return rc;
        d_gm = h->gmarks.as<int32_t>();
    }
    const bool trace = h->tun.chain_trace;
    // GAB_CHAIN_FEED_GIVEUP=1 (test hook): the gather kernel never publishes the facts word of the first (longest)
    // call and a wait gives up after ~20 ms instead of seconds -- the branch nobody runs otherwise: the waiting workgroup sets
    // the abort word, every other wait follows, the grid drains and gab_chain_run re-runs the batch through the copy engines
    const bool giveup_test = h->tun.chain_feed_giveup;      // GAB_CHAIN_FEED_GIVEUP
    hipEvent_t tv[4] = {};
    if (trace) for (auto &e : tv) (void)hipEventCreate(&e);
    // ---- gather stream
    GAB_HIP(hipEventRecord(h->ev[0], sA));
    if (trace) (void)hipEventRecord(tv[0], sG);
    GAB_HIP(hipMemsetAsync(wb + o_ev, 0, o_xlo - o_ev, sG));                       // evals, abort, facts, done, (need), mixed
    GAB_HIP(hipMemsetAsync(wb + o_xlo, 0xff, o_xhi - o_xlo, sG));
    GAB_HIP(hipMemsetAsync(wb + o_xhi, 0, o_chunks - o_xhi, sG));
    GAB_HIP(hipMemcpyAsync(d_work, wk.data(), sizeof(ChainWork) * nw, hipMemcpyHostToDevice, sG));
    GAB_HIP(hipMemcpyAsync(wb + o_need, need.data(), 4 * nw, hipMemcpyHostToDevice, sG));
    GAB_HIP(hipMemcpyAsync(wb + o_chunks, chunks.data(), sizeof(ChainChunk) * chunks.size(), hipMemcpyHostToDevice, sG));
    if (d_gm) GAB_HIP(hipMemsetAsync(d_gm, 0, sizeof(int32_t) * t, sG));            // vector::resize zero-fills targets
    memset(h->h_started, 0, gather_blocks);
    void *d_started = nullptr;
    GAB_HIP(hipHostGetDevicePointer(&d_started, h->h_started, 0));
    unsigned long long *d_pub = nullptr;
    if (trace) { (void)hipMalloc((void **)&d_pub, 8 * nw); (void)hipMemsetAsync(d_pub, 0, 8 * nw, sG); }
    GAB_HIP(hipEventRecord(h->xe_fed, sG));                                         // tables and zeroed arrays are in place
    hipLaunchKernelGGL(chain_gather_kernel, dim3(gather_blocks), dim3(256), 0, sG, (const ChainChunk *)(wb + o_chunks), (uint32_t)chunks.size(),
                       (const uint64_t *)hx, (const uint64_t *)hy, dx, dy, (const ChainWork *)d_work, (const uint32_t *)(wb + o_need),
                       (uint32_t *)(wb + o_done), (unsigned long long *)(wb + o_xlo), (unsigned long long *)(wb + o_xhi),
                       (uint32_t *)(wb + o_mixed), (uint32_t *)(wb + o_facts), (volatile uint8_t *)d_started, kGapTab - 2, d_abort + 2, d_pub,
                       giveup_test ? 0u : ~0u);
    GAB_HIP(hipGetLastError());
    if (trace) (void)hipEventRecord(tv[1], sG);
    // ---- wait until every gather workgroup is resident (they all start at once on an idle GPU: tens of microseconds);
    // if that does not happen in time, let the gather finish first -- the DP kernel then finds every word published
    {
        const auto t0 = std::chrono::steady_clock::now();
        bool all = false;
        while (!all) {
            all = true;
            for (int k = 0; k < gather_blocks; k++) if (!((volatile uint8_t *)h->h_started)[k]) { all = false; break; }
            if (!all && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;     // (somebody else holds the gather's CUs)
        }
        if (!all) {
            if (trace) fprintf(stderr, "[gab_chain_run] fed: not every gather workgroup was resident after 20 ms (somebody else holds its CUs?); the gather runs to its end before the DP starts\n");
            GAB_HIP(hipStreamSynchronize(sG));
        }
    }
    if (h->tun.chain_fed_serial) GAB_HIP(hipStreamSynchronize(sG));        // experiments: the DP only after the last anchor
    // ---- the DP: one launch, its workgroups wait for their call
    GAB_HIP(hipStreamWaitEvent(sA, h->xe_fed, 0));
    const bool write_through = true;              // (measured: results by two copies at the end instead cost their 12 ms in full)
    unsigned long long *d_dbg = nullptr;
    if (trace && mode == GAB_CHAIN) { (void)hipMalloc((void **)&d_dbg, 24 * nw); (void)hipMemsetAsync(d_dbg, 0, 24 * nw, sA); }
    ChainFeed feed{(uint32_t *)(wb + o_facts), write_through ? (int32_t *)hs : nullptr, write_through ? (int32_t *)hp : nullptr, d_abort, d_dbg,
                   giveup_test ? 3000 : kFeedSpinLimit};
    if (trace) (void)hipEventRecord(tv[2], sA);
    chain_launch(h->tun, mode, 3, sA, d_work, (unsigned)nw, dx, dy, ds, dp, d_gm, d_ev, &feed);
    GAB_HIP(hipGetLastError());
    if (trace) (void)hipEventRecord(tv[3], sA);
    GAB_HIP(hipEventRecord(h->ev[1], sA));
    GAB_HIP(hipMemcpyAsync(h->h_evals, d_ev, 2 * sizeof(unsigned long long) + 8, hipMemcpyDeviceToHost, sA));
    GAB_HIP(hipStreamSynchronize(sG));
    GAB_HIP(hipStreamSynchronize(sA));            // (the host vectors must outlive their copies)
    if (trace) {
        float a = 0, c = 0;
        float d = 0;
        (void)hipEventElapsedTime(&a, tv[0], tv[1]); (void)hipEventElapsedTime(&c, tv[0], tv[3]); (void)hipEventElapsedTime(&d, tv[2], tv[3]);
        fprintf(stderr, "[gab_chain_run] fed: %zu calls in %zu chunks; all anchors on the device after %.1f ms, DP done after %.1f ms (the DP launch itself: %.1f ms)\n", nw, chunks.size(), a, c, d);
        for (auto &e : tv) (void)hipEventDestroy(e);
        if (d_dbg) {
            std::vector<unsigned long long> dbg(3 * nw), pub(nw);
            (void)hipMemcpy(dbg.data(), d_dbg, 24 * nw, hipMemcpyDeviceToHost);
            if (d_pub) (void)hipMemcpy(pub.data(), d_pub, 8 * nw, hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull;
            std::vector<int> xcc(nw);
            for (size_t k = 0; k < nw; k++) { xcc[k] = (int)(dbg[3 * k] & 15); dbg[3 * k] >>= 4; }
            for (size_t k = 0; k < nw; k++) t0 = std::min(t0, dbg[3 * k]);
            for (size_t k : {(size_t)0, (size_t)50, (size_t)300, (size_t)900, (size_t)1100, (size_t)1300, (size_t)1600, (size_t)2500, (size_t)5000, nw - 1})
                if (k < nw) fprintf(stderr, "   item %zu (n = %lld): dispatched %.2f ms, published by the gather %.2f ms, anchors there %.2f ms, done %.2f ms\n", k, (long long)wk[k].n,
                                    (dbg[3 * k] - t0) * 1e-5, ((double)pub[k] - (double)t0) * 1e-5, (dbg[3 * k + 1] - t0) * 1e-5, (dbg[3 * k + 2] - t0) * 1e-5);
            {
                int hist[16] = {};
                for (int k = 0; k < gather_blocks; k++) { const int v = ((volatile uint8_t *)h->h_started)[k]; if (v) hist[(v - 1) & 15]++; }
                fprintf(stderr, "   gather workgroups per XCD: %d %d %d %d %d %d %d %d\n",
                        hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
            }
            for (int r = 0; r < 8; r++) {
                double lat = 0, disp = 0; int nl = 0, nd = 0;
                for (size_t k = 0; k < nw; k++) {
                    if (xcc[k] != r) continue;
                    if (k < 1000) { lat += (double)(dbg[3 * k + 2] - dbg[3 * k + 1]) * 1e-5 / ((double)wk[k].n * 1e-3); nl++; }
                    else if (k >= 3000) { disp += (double)(dbg[3 * k] - t0) * 1e-5; nd++; }
                }
                fprintf(stderr, "   XCD %d: %.3f ms per 1000 anchors for its calls among the 1000 longest; items 3000.. dispatched after %.1f ms on average\n",
                        r, nl ? lat / nl : 0., nd ? disp / nd : 0.);
            }
            (void)hipFree(d_dbg);
            if (d_pub) (void)hipFree(d_pub);
        }
    }
    if (((uint32_t *)h->h_evals)[4] != 0) {
        // a wait gave up (seconds without its anchors: a stalled bus, a pre-empted gather kernel).  Every waiting workgroup
        // has left, the grid has drained, nothing was lost but time: the caller takes the copy-engine path instead.
        fprintf(stderr, "[gab_chain_run] note: the fed DP kernel gave up waiting for its anchors; re-running the batch through the copy engines\n");
        return 2;
    }
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
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    // (GAB_CHAIN_FEED_MIN: tests push small batches through the big-batch paths)
    gab_tuning_refresh(&h->tun);
    const int64_t big = h->tun.chain_feed_min >= 0 ? h->tun.chain_feed_min : ((int64_t)8 << 20);      // GAB_CHAIN_FEED_MIN
    if (total >= big && (ncalls >= 1024 || h->tun.chain_feed_min >= 0) && ncalls < (1ll << 31) && !h->tun.chain_no_overlap) {
        if (!h->tun.chain_no_feed && !h->tun.chain_walk) {
            rc = chain_run_fed(h, mode, x, y, call_off, hdr, ncalls, total, score_out, parent_out, s);
            if (rc != 1 && rc != 2) return rc;    // 1: the arrays are not page-locked; 2: the fed kernel gave up waiting
        }
        return chain_run_overlapped(h, mode, x, y, call_off, hdr, ncalls, total, score_out, parent_out, s);
    }
    GAB_HIP(hipMemcpyAsync(dx, x, 8 * t, hipMemcpyHostToDevice, s));
    GAB_HIP(hipMemcpyAsync(dy, y, 8 * t, hipMemcpyHostToDevice, s));
    rc = gab_chain_run_device(h, mode, dx, dy, call_off, hdr, ncalls, ds, dp, s);
    if (rc) return rc;
    GAB_HIP(hipMemcpyAsync(score_out, ds, 4 * t, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipMemcpyAsync(parent_out, dp, 4 * t, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    return GAB_OK;
}

// Pre-size the handle's device buffers for calls of up to max_anchors anchors in max_calls calls (see gab_bsw_reserve).
extern "C" int gab_chain_reserve(gab_chain *h, int64_t max_anchors, int64_t max_calls) {
    GAB_CHECK(h, "gab_chain_reserve: NULL handle");
    GAB_CHECK(max_anchors >= 0 && max_calls >= 0 && max_calls < (1ll << 31), "gab_chain_reserve: size out of range");
    gab_device_guard g(h->device);
    // (the fed path of gab_chain_run starts every call on a line of 16 anchors and keeps its per-call words and its chunk table
    // behind the work list)
    const size_t padded = (size_t)max_anchors + 16 * (size_t)max_calls;
    int rc = h->io.reserve(std::max<size_t>(24 * padded + 64, (size_t)4 << 20));      // (at least the 4 MB gab_warm_copy_engines moves)
    if (rc) return rc;
    if ((rc = h->gmarks.reserve(sizeof(int32_t) * padded + 64)) != GAB_OK) return rc;
    if ((rc = h->work.reserve((sizeof(ChainWork) + 32 + 7 * 256 / 8) * (size_t)max_calls + sizeof(ChainChunk) * ((size_t)max_anchors / kFeedChunk + (size_t)max_calls) + 4096)) != GAB_OK) return rc;
    hipStream_t s = nullptr;
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    GAB_HIP(hipMemsetAsync(h->io.p, 0, h->io.cap, s));
    GAB_HIP(hipMemsetAsync(h->gmarks.p, 0, h->gmarks.cap, s));
    GAB_HIP(hipStreamSynchronize(s));
    {   // the fed path's streams, and its gather kernel once with nothing to do (the first launch of a kernel is 1.6 ms)
        int blocks = 0;
        if ((rc = chain_fast_setup(h)) != GAB_OK || (rc = chain_fed_setup(h, &blocks)) != GAB_OK) return rc;
        hipStream_t sG = h->gs ? h->gs : h->xs[0];
        memset(h->h_started, 0, 256 + 64);
        void *d_started = nullptr;
        GAB_HIP(hipHostGetDevicePointer(&d_started, h->h_started, 0));
        GAB_HIP(hipMemsetAsync(h->work.p, 0, 256, sG));
        hipLaunchKernelGGL(chain_gather_kernel, dim3(blocks), dim3(256), 0, sG, (const ChainChunk *)nullptr, 0u, (const uint64_t *)nullptr,
                           (const uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr, (const ChainWork *)nullptr, (const uint32_t *)nullptr,
                           (uint32_t *)nullptr, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                           (volatile uint8_t *)d_started, 0, h->work.as<uint32_t>(), (unsigned long long *)nullptr, ~0u);
        GAB_HIP(hipGetLastError());
        GAB_HIP(hipStreamSynchronize(sG));
    }
    return gab_warm_copy_engines(s, h->io.p, h->io.cap);
}

// gab_chain_reserve + what the DEVICE entry points of `mode` need before a timed region: one small batch through
// gab_chain_run_device (the first launch of every kernel of the forms the mode uses) and -- fast-chain, whose calls of a few
// thousand anchors and more run in the table form whatever the batch -- the table for max_anchors anchors (36 GB for fast-chain-large:
// an allocation of that size inside a driver's region of interest would cost more than the call).
extern "C" int gab_chain_reserve_mode(gab_chain *h, int mode, int64_t max_anchors, int64_t max_calls) {
    int rc = gab_chain_reserve(h, max_anchors, max_calls);
    if (rc) return rc;
    GAB_CHECK(mode == GAB_CHAIN || mode == GAB_FASTCHAIN, "gab_chain_reserve_mode: unknown mode %d", mode);
    gab_device_guard g(h->device);
    constexpr int kN = 6144;                                     // one call, long enough for the table form
    std::vector<uint64_t> xy(2 * kN);
    for (int i = 0; i < kN; i++) { xy[i] = 100 + 13ull * i; xy[kN + i] = (15ull << 32) | (uint64_t)(50 + 13 * i + (i % 7)); }
    char *b = h->io.as<char>();                                  // (>= 4 MB: gab_chain_reserve)
    uint64_t *d_x = (uint64_t *)b, *d_y = d_x + kN;
    int32_t *d_s = (int32_t *)(d_y + kN), *d_p = d_s + kN;
    hipStream_t s = nullptr;
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    GAB_HIP(hipMemcpyAsync(d_x, xy.data(), 16 * (size_t)kN, hipMemcpyHostToDevice, s));
    GAB_HIP(hipStreamSynchronize(s));
    gab_chain_hdr hd; memset(&hd, 0, sizeof hd);
    hd.n = kN; hd.avg_qspan = 15.f; hd.max_dist_x = 5000; hd.max_dist_y = 5000; hd.bw = 500; hd.n_segs = 1;
    const int64_t off = 0;
    const bool had = h->have_stats;
    rc = gab_chain_run_device(h, mode, d_x, d_y, &off, &hd, 1, d_s, d_p, s);
    h->have_stats = had;
    if (rc) return rc;
    if (h->tun.chain_tab != 0) rc = chain_tab_prealloc(&h->tab, max_anchors, max_calls);
    return rc;
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
