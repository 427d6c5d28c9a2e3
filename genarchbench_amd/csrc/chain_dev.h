// chain / fast-chain: what the two translation units of the seed-chaining kernels share (chain.hip: the block kernels, the
// latency form, the host side; chain_tab.hip: the table form).  Everything here has internal linkage.
#pragma once
#include "gab_internal.h"
#include <vector>

struct ChainWork {                        // one call, device-side descriptor
    int64_t off, n;                       // first anchor in the DEVICE arrays (x, y, score, parent, marks), number of anchors
    float avg_qspan;
    int32_t max_dist_x, max_dist_y, bw, n_segs, pad;
    int64_t hoff;                         // first anchor in the caller's arrays (= off except in the fed path, which pads calls to lines)
};

namespace {

constexpr int kMaxIter = 5000;
constexpr int kMaxSkip = 25;
constexpr int kMarkRing = 1024;           // LDS ring of mark tags for the newest anchors; older marks go to global memory

constexpr int kGapTab = 2048;             // entries of the per-call gap-cost table (LDS, int32): bw + 2 of them are used

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

__device__ __forceinline__ int32_t chain_gap_cost(int32_t dd, double avg_d) {
    const int32_t lgh = (31 - __clz((int)((uint32_t)dd | 1u))) >> 1;          // ilog2(dd) >> 1, ilog2(0) = 0
    const int32_t gap = (int32_t)__dmul_rn(__dmul_rn((double)dd, .01), avg_d) + lgh;
    return gap - (gap >> 31);
}

__device__ __forceinline__ int32_t chain_geometry(uint64_t xi, int32_t qi, int32_t q_span, int32_t sidi, uint64_t xj, uint32_t yj,
                                                  int32_t sidj, int32_t mdx, int32_t mdy, int32_t bw, bool multi_seg, double avg_d,
                                                  bool &ok) {
    const int64_t dr = (int64_t)(xi - xj);
    const int32_t dq = qi - (int32_t)yj;
    const bool same = sidi == sidj;
    const int32_t dd = (int32_t)(dr > dq ? dr - dq : dq - dr);
    const bool skip = (same && dr == 0) || dq <= 0 || (same && dq > mdy) || dq > mdx || (same && dd > bw) ||
                      (multi_seg && same && dr > mdy);
    ok = !skip;
    const int32_t min_d = (int32_t)(dq < dr ? (int64_t)dq : dr);
    int32_t v = min_d > q_span ? q_span : min_d;
    const int32_t lg = dd ? ilog2_u32((uint32_t)dd) : 0;
    const int32_t c_lin = (int32_t)__dmul_rn(__dmul_rn((double)dd, .01), avg_d);
    int32_t gap;
    if (!same) {
        if (dr == 0) { ++v; gap = 0; }
        else gap = c_lin < lg ? c_lin : lg;
    } else gap = c_lin + (lg >> 1);
    // (int)((double)gap_cost * 1.0f + .499): gap_cost is an integer, so the truncation gives gap_cost itself when it is
    // >= 0 and gap_cost + 1 when it is negative (a negative avg_qspan makes it so): two integer operations, no fp64
    v -= gap - (gap >> 31);
    return v;
}


// the reference's scan of ONE anchor, by one whole wave, everything from global memory (x, y input; score, parent of
// the predecessors as stored so far; marks in the per-anchor global array with tag i + 1).  Same three parallel steps as
// the exact path of chain_hw_kernel.  Returns (best, best_j absolute) in all lanes; `evals` counts the visited items.
template <class Anchors>
__device__ __forceinline__ void chain_exact_global(const Anchors X, const Anchors Y, const int32_t *S, const int32_t *P, int32_t *GM,
                                                   int i, int st, int32_t mdx, int32_t mdy, int32_t bw, bool multi_seg, double avg_d,
                                                   int32_t &best_out, int32_t &bestj_out, unsigned long long &evals) {
    const int lane = threadIdx.x & 63;
    const int NEG = (int)0x80000000;
    const uint64_t xi = X[i], yi = Y[i];
    const int32_t qi = (int32_t)yi, q_span = (int32_t)(yi >> 32 & 0xff), sidi = (int32_t)(yi >> 48 & 0xff);
    int32_t best = q_span, best_j = -1;
    int n_skip = 0;
    bool broke = false;
    for (int top = i - 1; top >= st && !broke;) {
        const int j0 = 4 * ((top >> 2) - lane);
        bool valid[4], ok[4];
        int32_t sc[4], parj[4] = {-1, -1, -1, -1};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = j0 + 3 - k;
            valid[k] = j >= st && j <= top; ok[k] = false; sc[k] = 0;
            if (valid[k]) {
                const uint64_t xj = X[j], yy = Y[j];
                const int32_t scj = __hip_atomic_load(&S[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                parj[k] = __hip_atomic_load(&P[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool okk;
                const int32_t v = chain_geometry(xi, qi, q_span, sidi, xj, (uint32_t)yy, (int32_t)(yy >> 48 & 0xff), mdx, mdy, bw, multi_seg, avg_d, okk);
                ok[k] = okk; sc[k] = v + scj;
            }
        }
        // marks: targets[parent[j]] = i for every unfiltered item, then this group's own four
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (ok[k] && parj[k] >= 0 && parj[k] >= st) __hip_atomic_store(&GM[parj[k]], (int32_t)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool hit[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = j0 + 3 - k;
            hit[k] = ok[k] && __hip_atomic_load(&GM[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int32_t)(i + 1);
        }
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
        const int p0 = d[0], p1 = p0 + d[1], p2 = p1 + d[2], p3 = p2 + d[3];
        const int E = wave_incl_sum(p3) - p3;
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
        const int klim = lane < fl ? 4 : lane == fl ? fk : 0;
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
        const int vlim = lane < fl ? 4 : lane == fl ? fk + 1 : 0;
#pragma unroll
        for (int k = 0; k < 4; k++) evals += (valid[k] && k < vlim) ? 1 : 0;
        top = 4 * ((top >> 2) - 63) - 1;
    }
    best_out = best; bestj_out = best_j;
}


}  // namespace

// ---- the table form (chain_tab.hip) ------------------------------------------------------------------------------------------
// Device buffers of the table form, owned by a gab_chain handle.
struct ChainTab {
    gab_devbuf calls;      // TabCall[nsplit] | bail[nsplit] | block prefix | counters
    gab_devbuf blocks;     // TabBlock per block of 64 anchors
    gab_devbuf gtab;       // the calls' gap-cost tables, back to back
    gab_devbuf st;         // window start of every anchor (call-relative), indexed like x / y
    gab_devbuf table;      // the geometry tables: 1 KB per 16 predecessors x 64 anchors
    gab_devbuf dbg;        // GAB_CHAIN_TRACE only
    size_t table_budget = 0;   // bytes the table may take (0: decided at the first call)
    std::vector<unsigned char> host_calls;   // the host side of `calls` for the copy: outlives chain_tab_run (the caller synchronises later,
                                             // and a copy of more than 1 MiB from pageable memory is a DMA from the vector itself)
    void release() { calls.release(); blocks.release(); gtab.release(); st.release(); table.release(); dbg.release(); }
};
// Runs the first `nsplit` calls of the (device) work list `d_work` -- `h_work` is the same list on the host -- through the table
// form on stream `s`: window starts, geometry tables, fold.  d_bail[k] != 0 afterwards (on the stream) means call k was NOT
// computed (not eligible, no room in the table, certificate missed) and is the caller's to run through the other kernels.
int chain_tab_run(ChainTab *t, const gab_tuning &tun, int mode, hipStream_t s, const ChainWork *d_work, const ChainWork *h_work, size_t nsplit, int64_t total_anchors,
                  const uint64_t *d_x, const uint64_t *d_y, int32_t *d_score, int32_t *d_parent, int32_t *d_gm, unsigned long long *d_evals, uint32_t **d_bail,
                  int32_t *host_score = nullptr, int32_t *host_parent = nullptr);      // (device-visible addresses of page-locked host arrays: results written through)
void chain_tab_report(ChainTab *t, size_t nsplit);      // GAB_CHAIN_TRACE: what the last run did with its calls (after a synchronisation)
int chain_tab_setup();      // function attributes (dynamic LDS), once
int chain_tab_prealloc(ChainTab *t, int64_t max_anchors, int64_t max_calls);
