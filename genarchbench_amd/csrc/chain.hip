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
// walked sequentially by ONE wavefront; the parallelism is (a) the 64 predecessors of a window
// chunk, one per lane, and (b) thousands of independent calls, one workgroup (= one wave) each,
// longest call first.  Per anchor i:
//   * the window start `st` is advanced with one 64-wide compare + ballot against a cached block of
//     x values instead of the reference's scalar while-loop;
//   * the window [st, i-1] is swept in descending 64-anchor chunks.  Chunk 0 (the 64 most recent
//     anchors, which carry the RAW dependence on score[i-1]) lives in registers and is shifted by
//     one lane per anchor with a DPP wave_shr; older chunks are coalesced global loads (x, y,
//     score, parent) that hit L1/L2, issued one chunk ahead;
//   * FASTCHAIN: wave max-reduction of the chunk scores (DPP), ties -> larger j;
//   * CHAIN: the sequential max_skip logic is reproduced exactly in three wave-parallel steps
//     (SURVEY.md App. B8): every unfiltered lane first scatters its mark targets[parent[j]] = i
//     into a 16-bit LDS ring, then reads its own mark; an exclusive prefix-max (DPP scan) gives
//     the "sc > max_f" improvement flags; the n_skip counter with its > 25 break is a scalar walk
//     over the two ballot masks.  Marks scattered by lanes past the break point are harmless
//     because a mark value i is only ever compared with the current i.
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
constexpr int kMarkRing = 8192;           // >= kMaxIter + 64, power of two

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
__device__ __forceinline__ int wave_shr1(int v, int fill) { return GAB_DPP(fill, v, 0x138, 0xf); }
__device__ __forceinline__ uint64_t wave_shr1_u64(uint64_t v) {
    uint32_t lo = (uint32_t)wave_shr1((int)(uint32_t)v, 0), hi = (uint32_t)wave_shr1((int)(uint32_t)(v >> 32), 0);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ int ilog2_u32(uint32_t v) { return 31 - __clz((int)v); }

template <bool FAST>
__global__ __launch_bounds__(64) void chain_kernel(const ChainWork *__restrict__ work,
                                                   const uint64_t *__restrict__ xs,
                                                   const uint64_t *__restrict__ ys,
                                                   int32_t *score_out, int32_t *parent_out,
                                                   unsigned long long *evals_out) {
    __shared__ uint16_t marks[FAST ? 64 : kMarkRing];
    const ChainWork w = work[blockIdx.x];
    const int lane = threadIdx.x;
    const uint64_t *X = xs + w.off, *Y = ys + w.off;
    int32_t *S = score_out + w.off, *P = parent_out + w.off;
    const int64_t n = w.n;
    const int32_t mdx = w.max_dist_x, mdy = w.max_dist_y, bw = w.bw;
    const uint64_t mdx64 = (uint64_t)(int64_t)mdx;
    const double avg_d = (double)w.avg_qspan;
    const float k32 = (float)(0.01 * (double)w.avg_qspan);
    const bool multi_seg = w.n_segs > 1;

    // chunk 0 registers: lane l <-> anchor i-1-l
    uint64_t rx = 0; uint32_t ry = 0; int rsid = 0, rsc = 0, rpar = -1;
    // cached block of x for the window-start search
    int64_t st = 0, sb = 0;
    uint64_t XS = (lane < n) ? X[lane] : 0;
    unsigned long long evals = 0;

    for (int64_t i = 0; i < n; i++) {
        const uint64_t xi = X[i], yi = Y[i];             // wave-uniform
        if (!FAST && (i & 0x7fff) == 0) {
            // mark tags are 0x8000 | (i mod 2^15): wipe the ring once per tag epoch (and at call
            // start, the LDS still holds the previous workgroup's bytes) so a stale tag can never match
            for (int k = lane; k < kMarkRing; k += 64) marks[k] = 0;
            __syncthreads();
        }
        const int32_t qi = (int32_t)yi, q_span = (int32_t)(yi >> 32 & 0xff), sidi = (int32_t)(yi >> 48 & 0xff);
        // ---- window start (host_kernel.cpp:56-57 / fast :200-207)
        for (;;) {
            const int64_t cand = sb + lane;
            bool far = FAST ? ((xi - XS) > mdx64) : (xi > XS + mdx64);
            bool pass = cand < st || (cand < i && far);
            unsigned long long m = __ballot(pass);
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
        const int64_t count = i - st;                    // window size
        const bool wide = !((i - 1) - st <= 5);          // FAST only (:211/:440)
        int n_skip = 0;
        bool broke = false;

        // prefetch registers for the next global chunk
        uint64_t nx = 0, ny = 0; int nsc = 0, npar = -1;
        if (count > 64) {
            const int64_t j = i - 1 - 64 - lane;
            if (j >= st) { nx = X[j]; ny = Y[j]; nsc = S[j]; npar = FAST ? -1 : P[j]; }
        }
        for (int64_t c0 = 0; c0 < count && !broke; c0 += 64) {
            const int64_t j = i - 1 - c0 - lane;
            const bool valid = j >= st;
            uint64_t xj; uint32_t yj; int sidj, scj, parj;
            if (c0 == 0) { xj = rx; yj = ry; sidj = rsid; scj = rsc; parj = rpar; }
            else {
                xj = nx; yj = (uint32_t)ny; sidj = (int)(ny >> 48 & 0xff); scj = nsc; parj = npar;
                if (c0 + 64 < count) {                   // issue the following chunk now
                    const int64_t j2 = j - 64;
                    if (j2 >= st) { nx = X[j2]; ny = Y[j2]; nsc = S[j2]; npar = FAST ? -1 : P[j2]; }
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
                int32_t oc = min(min(ddr, ddq), q_span);
                const int32_t lg = dd ? ilog2_u32((uint32_t)dd) : 0;
                int32_t gc;
                if (wide) gc = (int32_t)floorf(__fmul_rn((float)dd, k32)) + (lg >> 1);
                else gc = (int32_t)__dmul_rn(__dmul_rn((double)dd, .01), avg_d) + (lg >> 1);
                sc = (int32_t)((uint32_t)scj + (uint32_t)oc - (uint32_t)gc);
                evals += valid ? 1 : 0;
                // wave max, ties -> larger j (= lower lane)
                int v = ok ? sc : (int)0x80000000;
                int mx = __builtin_amdgcn_readlane(wave_incl_max(v), 63);
                if (mx > best) {
                    unsigned long long who = __ballot(ok && sc == mx);
                    best = mx;
                    best_j = (int32_t)(i - 1 - c0 - __builtin_ctzll(who));
                }
            } else {
                const int64_t dr = (int64_t)(xi - xj);
                const int32_t dq = qi - (int32_t)yj;
                const bool same = sidi == sidj;
                const int32_t dd = (int32_t)(dr > dq ? dr - dq : dq - dr);
                bool skip = (same && dr == 0) || dq <= 0 || (same && dq > mdy) || dq > mdx || (same && dd > bw) ||
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
                // marks: scatter first, then read
                const uint16_t tag = (uint16_t)(0x8000 | (i & 0x7fff));
                if (ok && parj >= 0 && parj >= st) marks[parj & (kMarkRing - 1)] = tag;
                __syncthreads();
                const bool hit_raw = ok && marks[j & (kMarkRing - 1)] == tag;
                // improvement flags: strict > against everything visited before this lane
                const int v = ok ? sc : (int)0x80000000;
                const int incl = wave_incl_max(v);
                int before = wave_shr1(incl, (int)0x80000000);
                before = max(before, best);
                const bool imp = ok && sc > before;
                const unsigned long long imp_m = __ballot(imp), hit_m = __ballot(hit_raw && !imp);
                const unsigned long long valid_m = __ballot(valid);
                int brk = 64;
                if (hit_m == 0) {
                    n_skip -= __popcll(imp_m);
                    n_skip = n_skip < 0 ? 0 : n_skip;
                } else {
                    unsigned long long ev = imp_m | hit_m;
                    while (ev) {
                        const int b = __builtin_ctzll(ev);
                        ev &= ev - 1;
                        if ((imp_m >> b) & 1) { if (n_skip > 0) --n_skip; }
                        else if (++n_skip > kMaxSkip) { brk = b; break; }
                    }
                }
                const unsigned long long upto = brk >= 64 ? ~0ull : ((1ull << brk) - 1);
                const unsigned long long rec = imp_m & upto;
                if (rec) {
                    const int l = 63 - __builtin_clzll(rec);      // last record before the break
                    best = __builtin_amdgcn_readlane(sc, l);
                    best_j = (int32_t)(i - 1 - c0 - l);
                }
                if (lane == 0) evals += __popcll(valid_m & (brk >= 64 ? ~0ull : ((2ull << brk) - 1)));
                broke = brk < 64;
            }
        }
        if (lane == 0) { S[i] = best; P[i] = best_j; }
        // slide chunk 0 by one anchor; lane 0 <- anchor i
        rx = wave_shr1_u64(rx); ry = (uint32_t)wave_shr1((int)ry, 0); rsid = wave_shr1(rsid, 0);
        rsc = wave_shr1(rsc, 0); rpar = wave_shr1(rpar, -1);
        if (lane == 0) { rx = xi; ry = (uint32_t)yi; rsid = sidi; rsc = best; rpar = best_j; }
    }
    if (FAST) { for (int o = 32; o > 0; o >>= 1) evals += __shfl_xor(evals, o); }
    if (lane == 0 && evals) atomicAdd(evals_out, evals);
}

}  // namespace

// =============================================================================== host side
struct gab_chain {
    int device = 0;
    gab_devbuf work;       // ChainWork[ncalls] + evals counter
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
    h->work.release(); h->io.release();
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
    GAB_HIP(hipEventRecord(h->ev[0], s));
    if (mode == GAB_FASTCHAIN)
        hipLaunchKernelGGL(chain_kernel<true>, dim3((unsigned)nw), dim3(64), 0, s, d_work, d_x, d_y, d_score, d_parent, d_ev);
    else
        hipLaunchKernelGGL(chain_kernel<false>, dim3((unsigned)nw), dim3(64), 0, s, d_work, d_x, d_y, d_score, d_parent, d_ev);
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
