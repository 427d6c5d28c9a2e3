// fmi -- FM-index SMEM seeding (three passes per read) on gfx950.
//
// Semantics: the per-batch body of the fmi driver, /root/reference/benchmarks/fmi/fmi.cpp:288-348, on
// BWA-MEM2's FMI_search (/root/reference/benchmarks/fmi/bwa-mem2/x86_64/src/FMI_search.cpp):
//   pass 1  getSMEMsAllPosOneThread(min_intv = 1)                    :672-724 -> :496-670
//   pass 2  getSMEMsOnePosOneThread at the midpoint of every pass-1 SMEM with length >=
//           (int)(minSeedLen*1.5+.499) and s <= 10, min_intv = s + 1  fmi.cpp:300-324
//   pass 3  bwtSeedStrategyAllPosOneThread(max_intv = 20, minSeedLen+1) :726-812
//   sortSMEMs (rid, m ascending, n descending)                         :986-1022
// every extension = backwardExt :1025-1052 = two CP_OCC look-ups (GET_OCC, FMI_search.h:66-73).
// The index is the reference's own ".bwt.2bit.64" file (load_index :384-494), uploaded unchanged:
// 64-byte CP_OCC records, one per 64 BWT rows.
//
// Mapping.  The reference interleaves reads in lock-step rounds inside a thread to overlap cache
// misses; every read is nevertheless independent in all three passes.  Here one lane owns one read at a
// time and walks the three passes as an explicit state machine (see fmi_seed_kernel), so that the whole
// wave meets at ONE look-up site per step; reads, interval lists and the re-seeding queue stay in LDS /
// registers, and extensions that produce a pattern of at most eight bases are answered by an L2-resident
// table of bi-intervals instead of the index.  What bounds the kernel is the rate of random index
// look-ups (~55 G line fills/s on MI355X whatever the record size, profiles/r01_random_read_ceiling.md).
// A call is ONE batch of that kernel when memory allows (a batch ends with its slowest chains alone on the chip), and the
// backward phases whose interval lists stay wide -- those chains -- are handed over to fmi_wide_kernel, a group of 16 lanes
// per phase (FmiWideItem); pass 3 is a second launch of the state machine without list LDS.
//
// Roofline: readlen + 40 B per SMEM of streaming traffic + 64 B per CP_OCC record actually fetched
// (counted by the kernel, gab_fmi_last_records).
#include "gab_internal.h"
#include <algorithm>
#include <new>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {

struct CpOcc { int64_t cp_count[4]; uint64_t bits[4]; };   // CP_OCC, FMI_search.h:54-58
static_assert(sizeof(CpOcc) == 64, "CP_OCC must be 64 bytes");

struct FmiIdx {
    const CpOcc *cp_occ;
    int64_t count[5];          // already +1 (FMI_search.cpp:433-436)
    int64_t sentinel;
    int64_t ref_seq_len;
    const uint4 *kmer_tab;     // bi-intervals of every pattern of 1 .. kmer_depth bases (packed like the LDS entries)
    int kmer_depth;
};

struct PrevRec { int64_t n, k, l, s; };                    // 32 bytes, [entry][lane]
struct OutRec { uint32_t m, n; int64_t k, l, s; };         // 32 bytes, per-read slot

constexpr int kReadPool = 64;               // reads a wave reserves from the global queue at a time

struct FmiCounters {
    unsigned long long ext_calls;
    unsigned long long rec_reads;  // distinct 64-byte CP_OCC records fetched (1 or 2 per extension)
    unsigned long long tab_reads;  // extensions answered by the short-pattern table
    unsigned long long total;      // SMEMs found in this batch
    int32_t max_per_read;
    int32_t bad, first_bad;
    int32_t next_read;             // work queue of the seeding kernel
    int32_t n_ovf, pad;            // reads whose SMEMs overflowed the first-round slot
    int32_t wide_items, wide_cands, wide_queue, wide_pad;   // wide backward phases handed over (fmi_wide_kernel), the re-seeding candidates they find, its work queue
    uint32_t wide_top, wide_pad2;  // entries used of the hand-over list area
    unsigned long long wave_steps; // sum over waves of the steps of their longest-running lane (GAB_FMI_DEBUG)
    unsigned long long positions, spills, list_sum;   // seeding positions, those whose list outgrew LDS, sum of list lengths
};
GAB_STATIC_ATOMIC64(FmiCounters, ext_calls); GAB_STATIC_ATOMIC64(FmiCounters, rec_reads); GAB_STATIC_ATOMIC64(FmiCounters, tab_reads); GAB_STATIC_ATOMIC64(FmiCounters, total);
GAB_STATIC_ATOMIC64(FmiCounters, wave_steps); GAB_STATIC_ATOMIC64(FmiCounters, positions); GAB_STATIC_ATOMIC64(FmiCounters, spills); GAB_STATIC_ATOMIC64(FmiCounters, list_sum);

__device__ __forceinline__ void load_rec(const CpOcc *p, int64_t (&cnt)[4], uint64_t (&bits)[4]) {
    const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(p);
    const ulonglong2 a = q[0], b = q[1], c = q[2], d = q[3];
    cnt[0] = (int64_t)a.x; cnt[1] = (int64_t)a.y; cnt[2] = (int64_t)b.x; cnt[3] = (int64_t)b.y;
    bits[0] = c.x; bits[1] = c.y; bits[2] = d.x; bits[3] = d.y;
}

// backwardExt (FMI_search.cpp:1025-1052)
__device__ __forceinline__ void backward_ext(const FmiIdx &ix, int64_t k, int64_t l, int64_t s, int a, int64_t &ko,
                                             int64_t &lo, int64_t &so, unsigned long long &calls, unsigned long long &recs) {
    calls++;
    const int64_t sp = k, ep = k + s;
    int64_t c_sp[4], c_ep[4]; uint64_t b_sp[4], b_ep[4];
    const CpOcc *r_sp = ix.cp_occ + (sp >> 6), *r_ep = ix.cp_occ + (ep >> 6);
    load_rec(r_sp, c_sp, b_sp);
    recs += r_sp == r_ep ? 1 : 2;
    if (r_sp == r_ep) {
#pragma unroll
        for (int b = 0; b < 4; b++) { c_ep[b] = c_sp[b]; b_ep[b] = b_sp[b]; }
    } else load_rec(r_ep, c_ep, b_ep);
    const int y_sp = (int)(sp & 63), y_ep = (int)(ep & 63);
    const uint64_t m_sp = y_sp ? ~0ull << (64 - y_sp) : 0ull, m_ep = y_ep ? ~0ull << (64 - y_ep) : 0ull;
    int64_t kk[4], ss[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int64_t o_sp = c_sp[b] + __popcll(b_sp[b] & m_sp);
        const int64_t o_ep = c_ep[b] + __popcll(b_ep[b] & m_ep);
        kk[b] = ix.count[b] + o_sp;
        ss[b] = o_ep - o_sp;
    }
    const int64_t sent = (k <= ix.sentinel && k + s > ix.sentinel) ? 1 : 0;
    const int64_t l3 = l + sent, l2 = l3 + ss[3], l1 = l2 + ss[2], l0 = l1 + ss[1];
    ko = a == 0 ? kk[0] : a == 1 ? kk[1] : a == 2 ? kk[2] : kk[3];
    so = a == 0 ? ss[0] : a == 1 ? ss[1] : a == 2 ? ss[2] : ss[3];
    lo = a == 0 ? l0 : a == 1 ? l1 : a == 2 ? l2 : l3;
}
// ---- short-pattern table -------------------------------------------------------------------------------------
// The bi-interval (k, l, s) of a pattern is a function of the pattern alone, however the search reached it, so the
// result of any extension whose RESULT is at most `kmer_depth` bases long can be read from a table indexed by the
// pattern instead of being computed from two CP_OCC records.  Depth 8 = 87 380 entries x 16 B = 1.4 MB: it stays in
// every XCD's L2, whereas the CP_OCC records of those shallow intervals (the widest ones: two records each) are
// spread over the whole index and miss.  About a third of all extensions produce such short patterns.
// Entry of pattern b0 b1 .. b(len-1): index (4^len - 4) / 3 + sum b_i << 2i; values with s == 0 are stored as zeros
// (an empty interval only ever propagates s == 0, FMI_search.cpp:1040-1051).
constexpr int kDefaultKmerDepth = 8;      // GAB_FMI_KMER_DEPTH: 0 .. 11 (tuning)
__device__ __forceinline__ uint32_t kmer_level_off(int len) { return (0x55555555u & ((1u << (2 * len)) - 1u)) - 1u; }
__device__ __forceinline__ uint4 pack_iv(int64_t k, int64_t l, int64_t s, uint32_t n) {
    uint4 w;
    w.x = (uint32_t)k; w.y = (uint32_t)l; w.z = (uint32_t)s;
    w.w = ((uint32_t)(k >> 32) & 0xffu) | ((uint32_t)(l >> 32) & 0xffu) << 8 | ((uint32_t)(s >> 32) & 0xffu) << 16 | n << 24;
    return w;
}
__global__ __launch_bounds__(256) void fmi_build_kmer_level(FmiIdx ix, uint4 *tab, int len) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= (1u << (2 * len))) return;
    int64_t k = 0, l = 0, s = 0;
    if (len == 1) { const int a = (int)c; k = ix.count[a]; l = ix.count[3 - a]; s = ix.count[a + 1] - ix.count[a]; }
    else {
        const uint4 w = tab[kmer_level_off(len - 1) + (c >> 2)];          // the pattern without its first base
        const int64_t pk = (int64_t)((uint64_t)(w.w & 0xffu) << 32 | w.x), pl = (int64_t)((uint64_t)((w.w >> 8) & 0xffu) << 32 | w.y);
        const int64_t ps = (int64_t)((uint64_t)((w.w >> 16) & 0xffu) << 32 | w.z);
        if (ps > 0) { unsigned long long c0 = 0, c1 = 0; backward_ext(ix, pk, pl, ps, (int)(c & 3u), k, l, s, c0, c1); }
    }
    if (s <= 0) k = l = s = 0;
    tab[kmer_level_off(len) + c] = pack_iv(k, l, s, 0);
}
// One extension for the seeding kernel: either backwardExt on the index or, for `tlen` > 0, the table entry `tidx`.
// Both kinds fetch through the same loads (a table entry is a quarter of a 64-byte line), so a wave with lanes of
// both kinds still waits for memory once.
__device__ __forceinline__ void extend(const FmiIdx &ix, int64_t k, int64_t l, int64_t s, int a, bool use_tab, uint32_t tidx,
                                       bool tab_swap, int64_t &ko, int64_t &lo, int64_t &so, uint32_t &calls, uint32_t &recs,
                                       uint32_t &tabs) {
    const int64_t sp = k, ep = k + s;
    int64_t c_sp[4], c_ep[4]; uint64_t b_sp[4], b_ep[4];
    const CpOcc *r_sp = ix.cp_occ + (sp >> 6), *r_ep = ix.cp_occ + (ep >> 6);
    if (use_tab) r_sp = r_ep = reinterpret_cast<const CpOcc *>(ix.kmer_tab) + (tidx >> 2);
    load_rec(r_sp, c_sp, b_sp);
    if (r_sp == r_ep) {
#pragma unroll
        for (int b = 0; b < 4; b++) { c_ep[b] = c_sp[b]; b_ep[b] = b_sp[b]; }
    } else load_rec(r_ep, c_ep, b_ep);
    if (use_tab) {
        tabs++;
        const uint32_t e = tidx & 3u;
        const uint64_t lo64 = e == 0 ? (uint64_t)c_sp[0] : e == 1 ? (uint64_t)c_sp[2] : e == 2 ? b_sp[0] : b_sp[2];
        const uint64_t hi64 = e == 0 ? (uint64_t)c_sp[1] : e == 1 ? (uint64_t)c_sp[3] : e == 2 ? b_sp[1] : b_sp[3];
        const uint32_t wx = (uint32_t)lo64, wy = (uint32_t)(lo64 >> 32), wz = (uint32_t)hi64, ww = (uint32_t)(hi64 >> 32);
        const int64_t tk = (int64_t)((uint64_t)(ww & 0xffu) << 32 | wx), tl = (int64_t)((uint64_t)((ww >> 8) & 0xffu) << 32 | wy);
        so = (int64_t)((uint64_t)((ww >> 16) & 0xffu) << 32 | wz);
        ko = tab_swap ? tl : tk; lo = tab_swap ? tk : tl;
        return;
    }
    calls++;
    recs += r_sp == r_ep ? 1u : 2u;
    const int y_sp = (int)(sp & 63), y_ep = (int)(ep & 63);
    const uint64_t m_sp = y_sp ? ~0ull << (64 - y_sp) : 0ull, m_ep = y_ep ? ~0ull << (64 - y_ep) : 0ull;
    int64_t kk[4], ss[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int64_t o_sp = c_sp[b] + __popcll(b_sp[b] & m_sp);
        const int64_t o_ep = c_ep[b] + __popcll(b_ep[b] & m_ep);
        kk[b] = ix.count[b] + o_sp;
        ss[b] = o_ep - o_sp;
    }
    const int64_t sent = (k <= ix.sentinel && k + s > ix.sentinel) ? 1 : 0;
    const int64_t l3 = l + sent, l2 = l3 + ss[3], l1 = l2 + ss[2], l0 = l1 + ss[1];
    ko = a == 0 ? kk[0] : a == 1 ? kk[1] : a == 2 ? kk[2] : kk[3];
    so = a == 0 ? ss[0] : a == 1 ? ss[1] : a == 2 ? ss[2] : ss[3];
    lo = a == 0 ? l0 : a == 1 ? l1 : a == 2 ? l2 : l3;
}
// ---- wide backward phases: handed over -----------------------------------------------------------------------------
// The backward phase of a seeding position extends every entry of an interval list, column by column.  Most lists shrink to
// one or two entries after their first column; the list of a position inside a repeat does not: 2.5 % of the phases of
// fmi-large keep 16 entries or more for a column or longer, they are 9.4 % of all extensions, and the widest -- 76 entries
// over 75 columns, 5 700 dependent steps for the one lane that owns the read -- are what a batch waits ~22 ms for while the
// chip is idle (profiles/r03_fmi_batches.md).  The entries of a column are independent extensions.  So a lane whose column
// leaves kWideMin survivors or more writes them to a hand-over area with the state of the phase (FmiWideItem) and goes on to
// its next position; fmi_wide_kernel takes the items afterwards, a group of 16 lanes per item, sixteen entries per step.
// A pass-1 phase emits re-seeding candidates (fmi.cpp:300-324): the wide kernel queues those (kind 1: x, min_intv) and a
// second launch of it runs their forward phase (all lanes of the group in step) and their backward phase.
struct FmiWideItem {            // 32 bytes
    uint32_t t;                 // read of the batch
    uint32_t jm;                // kind 0: column to do next + 1 | min_intv << 16 | pass 1 ? 1u << 31 : 0;  kind 1: x | min_intv << 16
    uint32_t off, n;            // kind 0: the list = entries off .. off + n - 1 of the hand-over area, longest match first
    uint32_t cur_m;             // kind 0: start of the match the list's entries stand for
    uint32_t kind;              // 0 continuation, 1 re-seeding candidate, 2 void
    uint32_t pad[2];
};
static_assert(sizeof(FmiWideItem) == 32, "FmiWideItem must be 32 bytes");
constexpr int kWideMin = 24;
struct FmiWide { FmiWideItem *items; uint4 *lists; int32_t items_cap; uint32_t lists_cap; int32_t min_entries; };

// ---- the seeding kernel: one lane = one read at a time, as a state machine ------------------------------------
// A straight transcription (one lane runs the three passes as nested loops) leaves ~13 % of the lanes active
// (PMC, profiles/r01_fmi_pmc.md): neighbouring reads sit in different loops, so every look-up site executes with a
// handful of lanes.  Here every lane keeps its position in the three passes as explicit state and the wave loop is
//     A: advance the state (cheap: read a base, push / pop list entries) until the lane needs an extension
//     B: ONE backwardExt site for the whole wave  <- the only place the index is read
//     C: consume the result according to the state
// so a look-up instruction serves every lane that still has work.  A lane that finishes its read pulls the next
// one of the batch from a global counter (a persistent grid sized to the occupancy), which evens out reads of
// different cost.  The read itself and the interval lists stay in LDS: the only global traffic in the loop is the
// index look-up (the global-memory lists of the first version cost as many L2 misses as the look-ups).
enum FmiState : int {
    ST_NEW_READ, ST_P1_NEXT, ST_START_POS, ST_FWD_STEP, ST_FWD_END, ST_BWD_COL, ST_BWD_ENT, ST_BWD_END, ST_POS_DONE,
    ST_P2_NEXT, ST_P3_START, ST_P3_STEP, ST_P3_JUMP, ST_READ_DONE, ST_DONE
};

// LDSQ: the lane's current read sits in LDS as 4-bit codes ([word of 8 bases][lane], conflict-free), so stepping
// along the read costs no global round trip; used whenever the read length bound fits (stride <= kLdsQMax).
constexpr int kLdsQMax = 256;
template <bool LDSQ>
__global__ __launch_bounds__(64) void fmi_seed_kernel(FmiIdx ix, const uint8_t *__restrict__ enc, int32_t stride,
                                                      const int32_t *__restrict__ len_arr, int64_t first, int32_t nbatch,
                                                      int min_seed_len, PrevRec *prev, int prev_cap, OutRec *out_all, int cap,
                                                      int32_t *counts, FmiCounters *ct, int lds_entries, int64_t enc_bytes, int passes,
                                                      int narrow_lists, const int32_t *__restrict__ ids, FmiWide wide) {
    // dynamic LDS (LDSQ only): [ (stride + 7) / 8 words of read codes ][ lds_entries list entries ]  x 64 lanes; a list entry
    // is 16 bytes (k, l, s < 2^40, n < 256 packed), or -- narrow_lists, for indexes below 2^32 rows -- three dword planes
    // and a byte plane = 13 bytes, which is two more waves per CU at 12 entries
    extern __shared__ uint4 lds_all[];
    const int lane = threadIdx.x;
    uint32_t *const lq = reinterpret_cast<uint32_t *>(lds_all) + lane;
    uint4 *const lp = lds_all + ((stride + 7) / 8 * 64 + 3) / 4 + lane;        // entry slot e at lp[e * 64]
    const int C = LDSQ ? lds_entries : 0;
    const int64_t tid = (int64_t)blockIdx.x * 64 + lane;
    const int64_t pstride = (int64_t)gridDim.x * 64;
    PrevRec *const prevp = prev + tid;                       // overflow entry e of this lane at prevp[e * pstride]
    const int split_len = (int)(min_seed_len * 1.5 + .499);
    const int msl = min_seed_len + 1;

    unsigned long long tot = 0;
    uint32_t calls = 0, recs = 0, tabs = 0;                  // per lane and launch: far below 2^32
    const int D = LDSQ ? ix.kmer_depth : 0;
    int mx = 0;
    // ---- per-lane state
    int state = ST_NEW_READ;
    int pool_next = 0, pool_end = 0;                         // wave-uniform: this wave's reserved slice of the read queue
    const uint8_t *q = enc; OutRec *out = out_all;
    int t = 0, len = 0, pass = 1, x = 0, next_x = 0, j = 0, a = 0;
    int nprev = 0, ncur = 0, p = 0, curr_s = -1, n1 = 0, jrec = 0, nout = 0, nout0 = 0;
    bool first_phase = true;
    uint32_t cur_m = 0;
    int64_t min_intv = 1, sm_k = 0, sm_l = 0, sm_s = 0;
    int sm_n = 0;
    PrevRec s0; s0.n = s0.k = s0.l = s0.s = 0;

    auto base_at = [&](int pos) -> int {
        if (LDSQ) return (int)((lq[(pos >> 3) * 64] >> ((pos & 7) * 4)) & 15u);
        return (int)q[pos];
    };
    // eight read codes from `start` on, as 4-bit nibbles (first base lowest); LDSQ only
    auto nibbles_at = [&](int start) -> uint32_t {
        const uint32_t w0 = lq[(start >> 3) * 64], w1 = lq[((start >> 3) + 1) * 64];   // w1 may be past the read: masked by callers
        return (uint32_t)(((uint64_t)w1 << 32 | w0) >> ((start & 7) * 4));
    };
    // ... sixteen of them (tables deeper than 8 bases, r04): three words
    auto nibbles_at64 = [&](int start) -> uint64_t {
        const uint32_t w0 = lq[(start >> 3) * 64], w1 = lq[((start >> 3) + 1) * 64], w2 = lq[((start >> 3) + 2) * 64];   // (w1 / w2 may be past the read: masked by callers; the LDS slack covers them)
        const int sh = (start & 7) * 4;
        const uint64_t lo = (uint64_t)w1 << 32 | w0, hi = (uint64_t)w2 << 32 | w1;
        return sh ? (lo >> sh) | ((hi >> sh) << 32) : lo;
    };
    // table index of the pattern of `plen` <= kmer_depth bases starting at `start` (all of them A/C/G/T)
    auto kmer_index = [&](int start, int plen) -> uint32_t {
        if (D <= 8) {
            uint32_t v = nibbles_at(start) & 0x33333333u;
            v = (v | v >> 2) & 0x0f0f0f0fu; v = (v | v >> 4) & 0x00ff00ffu; v = (v | v >> 8) & 0xffffu;
            return kmer_level_off(plen) + (v & ((1u << (2 * plen)) - 1u));
        }
        uint64_t v = nibbles_at64(start) & 0x3333333333333333ull;
        v = (v | v >> 2) & 0x0f0f0f0f0f0f0f0full; v = (v | v >> 4) & 0x00ff00ff00ff00ffull; v = (v | v >> 8) & 0x0000ffff0000ffffull; v = (v | v >> 16) & 0xffffffffull;
        return kmer_level_off(plen) + ((uint32_t)v & ((1u << (2 * plen)) - 1u));
    };
    // The interval lists of one seeding position (FMI_search.cpp:531-650: prev[] / curr[]).  Backward columns address
    // the list by v = 0, 1, ... (0 = longest match = the NEWEST forward entry), which makes the reference's reversal
    // implicit.  The C most recent forward entries sit in an LDS ring (16-byte packed records: k, l, s < 2^40, n < 256);
    // entry v of the list is ring slot (Rslot - v) mod C, and a survivor written as entry w <= v reuses the slot of an
    // entry that has already been read.  Whatever does not fit (forward entries older than the newest C, survivors
    // beyond C) goes to the lane's global scratch: forward entries stacked down from the top, survivors up from 0, so
    // the two never meet.  Global entries are fetched one step ahead, together with the index records.
    bool rev = true;
    int fslot = 0, Rslot = 0, nfwd = 0, nxt_for = -1;
    PrevRec nxt; nxt.n = nxt.k = nxt.l = nxt.s = 0;
    // narrow: field f of slot e of this lane at lw[(3 e + f) * 64], its byte at lb[e * 64] (planes behind the read codes)
    uint4 *const list_base = lds_all + ((stride + 7) / 8 * 64 + 3) / 4;
    uint32_t *const lw = reinterpret_cast<uint32_t *>(list_base) + lane;
    uint8_t *const lb = reinterpret_cast<uint8_t *>(list_base) + (size_t)C * 3 * 64 * 4 + lane;
    auto lds_put = [&](int slot, const PrevRec &r) {
        if (narrow_lists) {
            lw[(3 * slot) * 64] = (uint32_t)r.k; lw[(3 * slot + 1) * 64] = (uint32_t)r.l; lw[(3 * slot + 2) * 64] = (uint32_t)r.s;
            lb[slot * 64] = (uint8_t)r.n;
            return;
        }
        uint4 w;
        w.x = (uint32_t)r.k; w.y = (uint32_t)r.l; w.z = (uint32_t)r.s;
        w.w = ((uint32_t)(r.k >> 32) & 0xffu) | ((uint32_t)(r.l >> 32) & 0xffu) << 8 | ((uint32_t)(r.s >> 32) & 0xffu) << 16 |
              (uint32_t)r.n << 24;
        lp[slot * 64] = w;
    };
    auto lds_get = [&](int slot) -> PrevRec {
        PrevRec r;
        if (narrow_lists) {
            r.k = (int64_t)lw[(3 * slot) * 64]; r.l = (int64_t)lw[(3 * slot + 1) * 64]; r.s = (int64_t)lw[(3 * slot + 2) * 64];
            r.n = (int64_t)lb[slot * 64];
            return r;
        }
        const uint4 w = lp[slot * 64];
        r.k = (int64_t)((uint64_t)(w.w & 0xffu) << 32 | w.x); r.l = (int64_t)((uint64_t)((w.w >> 8) & 0xffu) << 32 | w.y);
        r.s = (int64_t)((uint64_t)((w.w >> 16) & 0xffu) << 32 | w.z); r.n = (int64_t)(w.w >> 24);
        return r;
    };
    auto g_fwd = [&](int i) -> PrevRec * { return prevp + (int64_t)(prev_cap - 1 - i) * pstride; };   // forward entry i, spilled
    auto list_slot = [&](int v) -> int { const int s = Rslot - v; return s < 0 ? s + C : s; };
    auto list_get_g = [&](int v) -> PrevRec { return rev ? *g_fwd(nfwd - 1 - v) : prevp[(int64_t)v * pstride]; };
    auto list_put = [&](int w, const PrevRec &r) { if (w < C) lds_put(list_slot(w), r); else prevp[(int64_t)w * pstride] = r; };
    // Re-seeding candidates (fmi.cpp:300-324) are noted when pass 1 emits them: 12 bits each (midpoint, s + 1) in a
    // register queue, so pass 2 does not re-read the output slot; more than kP2Max candidates (or a read longer than
    // 255) fall back to scanning the slot.
    constexpr int kP2Max = 10;
    uint64_t p2q0 = 0, p2q1 = 0; int p2n = 0;
    auto emit = [&](uint32_t m, uint32_t n, int64_t k, int64_t l, int64_t s) {
        if (nout < cap) { OutRec o; o.m = m; o.n = n; o.k = k; o.l = l; o.s = s; out[nout] = o; }
        nout++;
        if (LDSQ && pass == 1 && (int)(n + 1 - m) >= split_len && s <= 10) {
            const uint64_t e = (uint64_t)((n + 1 + m) >> 1) | (uint64_t)(s + 1) << 8;
            if (p2n < 5) p2q0 |= e << (12 * p2n); else if (p2n < kP2Max) p2q1 |= e << (12 * (p2n - 5));
            p2n++;
        }
    };
    auto push_fwd = [&]() {                                  // forward list, newest entry lowest: read back = longest first
        PrevRec r; r.n = sm_n; r.k = sm_k; r.l = sm_l; r.s = sm_s;
        if (C > 0) {
            if (nprev >= C) *g_fwd(nprev - C) = lds_get(fslot);        // the oldest resident entry makes room
            lds_put(fslot, r); Rslot = fslot; fslot = fslot + 1 == C ? 0 : fslot + 1;
        } else *g_fwd(nprev) = r;
        nprev++;
    };

    // One wave step = one pass over the blocks below, ordered along the transitions of the three passes so that a
    // lane normally reaches its next extension in this same pass (a lane that takes a backward edge -- end of a
    // forward walk, an N base -- just sits out this step's extension).  Each block runs once per step for the lanes
    // in that state instead of once per sub-transition.
    unsigned long long steps = 0;
    unsigned dbg_pos = 0, dbg_spill = 0, dbg_list = 0;
    while (state != ST_DONE) {
        steps++;
        bool need = false;
        int64_t K = 0, L = 0, S = 0; int A = 0;
        int tstart = 0, tlen = 0;                            // pattern the extension produces, if short enough for the table
        if (state == ST_FWD_END) {
            if (sm_s >= min_intv) push_fwd();
            rev = true; nfwd = nprev;                        // first backward column reads the forward entries
            dbg_pos++; dbg_spill += nprev > C ? 1 : 0; dbg_list += (unsigned)nprev;
            cur_m = (uint32_t)x; j = x - 1;
            state = ST_BWD_COL;
        }
        if (state == ST_BWD_COL) {                           // backward loop :589-650
            state = ST_BWD_END;
            if (j >= 0) {
                a = base_at(j);
                if (a <= 3) { ncur = 0; curr_s = -1; first_phase = true; p = 0; nxt_for = -1; state = nprev > 0 ? ST_BWD_ENT : ST_BWD_END; }
            }
        }
        if (state == ST_BWD_END) {
            if (nprev != 0) {
                const PrevRec r0 = C > 0 ? lds_get(Rslot) : list_get_g(0);
                if ((int)((int64_t)r0.n - (int64_t)cur_m + 1) >= min_seed_len) emit(cur_m, (uint32_t)r0.n, r0.k, r0.l, r0.s);
            }
            state = ST_POS_DONE;
        }
        if (state == ST_POS_DONE) {
            if (pass == 1) { x = next_x; state = ST_P1_NEXT; } else state = ST_P2_NEXT;
        }
        if (state == ST_READ_DONE) {
            counts[t] = nout;
            tot += (unsigned long long)(nout - nout0); mx = nout > mx ? nout : mx;
            state = ST_NEW_READ;
        }
        if (__any(state == ST_NEW_READ)) {                   // wave-uniform: the whole wave helps the lanes that start a read
            const bool want = state == ST_NEW_READ;
            const uint64_t wm = __ballot(want);
            const int leader = __builtin_ctzll(wm);
            // one queue for the whole grid, drawn from in chunks of kReadPool reads per wave: a returning atomic on a single
            // address per wave step with a finished read (millions per batch) serialises at ~7 ns apiece and stalls the
            // wave for a memory round trip; the wave-local pool needs one per 64 reads
            const int k = __builtin_popcountll(wm), rem = pool_end - pool_next;
            int nb = 0;
            if (rem < k) {
                if (lane == leader) nb = atomicAdd(&ct->next_read, kReadPool);
                nb = __builtin_amdgcn_readfirstlane(__shfl(nb, leader));
            }
            const int r = __builtin_popcountll(wm & ((1ull << lane) - 1ull));
            const int idx = r < rem ? pool_next + r : nb + (r - rem);
            if (rem < k) { pool_next = nb + (k - rem); pool_end = nb + kReadPool; } else pool_next += k;
            const bool got = want && idx < nbatch;
            if (want && !got) state = ST_DONE;
            if (got) {
                t = ids ? ids[idx] : idx;                      // ids: the second round over the reads that overflowed their slot
                const int64_t r = first + t;
                q = enc + r * (int64_t)stride; len = len_arr[r];
                out = out_all + (int64_t)idx * cap;
                x = 0; min_intv = 1; p2q0 = 0; p2q1 = 0; p2n = 0;
                if (passes & 1) { nout = 0; pass = 1; state = ST_P1_NEXT; }
                else { nout = counts[t]; pass = 3; state = ST_P3_START; }        // appends to what passes 1 and 2 found
                nout0 = nout;
            }
            if (LDSQ) {
                // read codes -> LDS column of the owning lane: lane w of the wave fetches the three aligned dwords around
                // bases 8w .. 8w+7 (coalesced), packs them to eight 4-bit codes and stores word w of that column
                uint64_t gm = __ballot(got);
                while (gm) {
                    const int src = __builtin_ctzll(gm);
                    gm &= gm - 1;
                    const int64_t rr = first + __shfl(t, src);
                    const uint8_t *qq = enc + rr * (int64_t)stride;
                    const int mis = (int)((uintptr_t)qq & 3);
                    const uint32_t *qa = reinterpret_cast<const uint32_t *>(qq - mis);
                    const int64_t o0 = (qq - mis - enc) + 8 * (int64_t)lane;      // byte offset of this lane's first dword
                    if (lane * 8 < stride) {
                        // an aligned dword that starts inside the buffer never crosses a page: safe to read whole
                        const uint32_t d0 = o0 < enc_bytes ? qa[2 * lane] : 0u, d1 = o0 + 4 < enc_bytes ? qa[2 * lane + 1] : 0u;
                        const uint32_t d2 = o0 + 8 < enc_bytes ? qa[2 * lane + 2] : 0u;
                        const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, mis), hi = __builtin_amdgcn_alignbyte(d2, d1, mis);
                        uint32_t pk = 0;
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const uint32_t x0 = (lo >> (8 * b)) & 0xffu, x1 = (hi >> (8 * b)) & 0xffu;
                            pk |= (x0 > 3u ? 4u : x0) << (4 * b) | (x1 > 3u ? 4u : x1) << (16 + 4 * b);
                        }
                        reinterpret_cast<uint32_t *>(lds_all)[lane * 64 + src] = pk;
                    }
                }
            }
        }
        if (state == ST_P1_NEXT) {                           // getSMEMsAllPosOneThread loop, FMI_search.cpp:672-724
            if (x < len) state = ST_START_POS;
            else { pass = 2; n1 = nout; jrec = 0; state = ST_P2_NEXT; }
        }
        if (state == ST_P2_NEXT) {                           // re-seeding, fmi.cpp:300-324
            bool started = false;
            if (n1 <= cap) {
                if (LDSQ && p2n <= kP2Max) {                 // candidates noted by emit(), in emission order
                    if (jrec < p2n) {
                        const uint32_t e = (uint32_t)((jrec < 5 ? p2q0 >> (12 * jrec) : p2q1 >> (12 * (jrec - 5))) & 0xfffu);
                        jrec++;
                        x = (int)(e & 0xffu); min_intv = (int64_t)(e >> 8);
                        started = true;
                    }
                } else {
                    while (jrec < n1) {
                        const OutRec o = out[jrec++];
                        const int start = (int)o.m, end = (int)o.n + 1;
                        if (end - start < split_len || o.s > 10) continue;
                        x = (end + start) >> 1; min_intv = o.s + 1;
                        started = true;
                        break;
                    }
                }
            }
            if (started) state = ST_START_POS;
            else if (passes & 2) { pass = 3; x = 0; state = ST_P3_START; }
            else state = ST_READ_DONE;                       // pass 3 runs as its own launch (next step)
        }
        if (state == ST_START_POS) {                         // getSMEMsOnePosOneThread :496-530
            next_x = x + 1;
            a = base_at(x);
            if (a >= 4) state = ST_POS_DONE;                 // (next step)
            else {
                sm_n = x; sm_k = ix.count[a]; sm_l = ix.count[3 - a]; sm_s = ix.count[a + 1] - ix.count[a];
                nprev = 0; fslot = 0; j = x + 1;
                state = ST_FWD_STEP;
            }
        }
        if (state == ST_P3_START) {                          // bwtSeedStrategyAllPosOneThread :726-812
            if (x >= len) state = ST_READ_DONE;              // (next step)
            else {
                next_x = x + 1;
                a = base_at(x);
                if (a < 4) {
                    sm_n = x; sm_k = ix.count[a]; sm_l = ix.count[3 - a]; sm_s = ix.count[a + 1] - ix.count[a];
                    j = x + 1;
                    state = ST_P3_STEP;
                    // the first D - 1 steps of the walk cannot emit (D < minSeedLen + 1) and only stop at an N or the
                    // read end: with D clean bases ahead their outcome is the table entry of q[x .. x+D-1]
                    if (D > 1 && D < msl && x + D <= len &&
                        (D <= 8 ? (nibbles_at(x) & 0x44444444u & (0xffffffffu >> (32 - 4 * D))) == 0u
                                : (nibbles_at64(x) & 0x4444444444444444ull & (~0ull >> (64 - 4 * D))) == 0ull))
                        state = ST_P3_JUMP;
                } else x = next_x;
            }
        }
        if (state == ST_FWD_STEP) {                          // forward loop :531-575
            if (j >= len) state = ST_FWD_END;
            else {
                a = base_at(j);
                next_x = j + 1;
                if (a >= 4) state = ST_FWD_END;
                else {
                    K = sm_l; L = sm_k; S = sm_s; A = 3 - a; need = true;
                    if (j - x < D) { tstart = x; tlen = j - x + 1; }
                }
            }
        } else if (state == ST_BWD_ENT) {
            s0 = p < C ? lds_get(list_slot(p)) : nxt_for == p ? nxt : list_get_g(p);
            K = s0.k; L = s0.l; S = s0.s; A = a; need = true;
            if ((int)s0.n - j < D) { tstart = j; tlen = (int)s0.n - j + 1; }
        } else if (state == ST_P3_JUMP) {
            tstart = x; tlen = D; need = true;
        } else if (state == ST_P3_STEP) {
            if (j >= len) { x = next_x; state = ST_P3_START; }
            else {
                next_x = j + 1;
                a = base_at(j);
                if (a >= 4) { x = next_x; state = ST_P3_START; }
                else {
                    K = sm_l; L = sm_k; S = sm_s; A = 3 - a; need = true;
                    if (j - x < D) { tstart = x; tlen = j - x + 1; }
                }
            }
        }
        if (!need) continue;
        // ---- the extension: the only place the index is read
        int64_t ko, lo, so;
        const bool fetch_next = state == ST_BWD_ENT && p + 1 < nprev && p + 1 >= C;
        PrevRec pre = nxt;
        if (fetch_next) { pre = list_get_g(p + 1); nxt_for = p + 1; }        // in flight together with the index records
        const bool use_tab = tlen > 0;
        const uint32_t tidx = use_tab ? kmer_index(tstart, tlen) : 0u;
        extend(ix, K, L, S, A, use_tab, tidx, state != ST_BWD_ENT, ko, lo, so, calls, recs, tabs);
        nxt = pre;
        // ---- consume
        if (state == ST_FWD_STEP) {                          // forward: result is (l, k, s) of the reverse strand
            if (so != sm_s) push_fwd();
            if (so < min_intv) { next_x = j; state = ST_FWD_END; }
            else { sm_k = lo; sm_l = ko; sm_s = so; sm_n = j; j++; }
        } else if (state == ST_BWD_ENT) {
            bool keep = false;
            if (first_phase && so < min_intv && (int)((int64_t)s0.n - (int64_t)cur_m + 1) >= min_seed_len) {
                emit(cur_m, (uint32_t)s0.n, s0.k, s0.l, s0.s);
                first_phase = false;
            } else if (so >= min_intv && so != (int64_t)curr_s) {
                keep = true; first_phase = false;
            }
            if (keep) {
                curr_s = (int)so;                            // int, as in the reference
                PrevRec nw; nw.n = s0.n; nw.k = ko; nw.l = lo; nw.s = so;
                list_put(ncur, nw);
                ncur++;
            }
            p++;
            if (p >= nprev) {                                // column done
                nprev = ncur; rev = false;
                if (ncur == 0) state = ST_BWD_END;
                else {
                    cur_m = (uint32_t)j; j--; state = ST_BWD_COL;
                    // a wide phase: hand the rest of it over.  Once the item queue is full the counter is left alone (a lane
                    // that keeps its phase comes back here after every later wide column; on a large low-complexity batch those
                    // increments could carry the int32 counter past 2^31, a negative `it` would pass the bound below and the wide kernel
                    // would see a negative item count)
                    if (LDSQ && wide.items && ncur >= wide.min_entries &&
                        __hip_atomic_load(&ct->wide_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wide.items_cap) {
                        const int it = atomicAdd(&ct->wide_items, 1);
                        if (it >= 0 && it < wide.items_cap) {           // (so the list cursor stays below items_cap x 256 entries: no wrap)
                            const uint32_t off = atomicAdd(&ct->wide_top, (uint32_t)ncur);
                            FmiWideItem w;
                            w.t = (uint32_t)t; w.jm = (uint32_t)(j + 1) | (uint32_t)min_intv << 16 | (pass == 1 ? 1u << 31 : 0u);
                            w.off = off; w.n = (uint32_t)ncur; w.cur_m = cur_m; w.kind = 2; w.pad[0] = w.pad[1] = 0;
                            if (off + (uint32_t)ncur <= wide.lists_cap && min_intv < 0x7fff) {
                                for (int v = 0; v < ncur; v++) {
                                    const PrevRec r = v < C ? lds_get(list_slot(v)) : prevp[(int64_t)v * pstride];
                                    wide.lists[off + (uint32_t)v] = pack_iv(r.k, r.l, r.s, (uint32_t)r.n);
                                }
                                w.kind = 0;
                                state = ST_POS_DONE;             // (next step) the phase is somebody else's now
                            }
                            wide.items[it] = w;
                        }
                    }
                }
            }
        } else if (state == ST_P3_JUMP) {                    // D bases in one go
            sm_k = lo; sm_l = ko; sm_s = so; sm_n = x + D - 1;
            tabs += (uint32_t)(D - 2);                       // the look-up stands for D - 1 extensions of the reference
            j = x + D; next_x = x + D;
            state = ST_P3_STEP;
        } else {                                             // ST_P3_STEP
            sm_k = lo; sm_l = ko; sm_s = so; sm_n = j;
            if (sm_s < 20 && sm_n - x + 1 >= msl) {
                if (sm_s > 0) emit((uint32_t)x, (uint32_t)sm_n, sm_k, sm_l, sm_s);
                x = next_x; state = ST_P3_START;
            } else j++;
        }
    }
    // statistics: one atomic per wave
    unsigned long long c64 = calls, r64 = recs, t64 = tabs, p64 = dbg_pos, s64 = dbg_spill, l64 = dbg_list;
    for (int o = 32; o > 0; o >>= 1) {
        c64 += __shfl_xor(c64, o); tot += __shfl_xor(tot, o); r64 += __shfl_xor(r64, o); t64 += __shfl_xor(t64, o);
        p64 += __shfl_xor(p64, o); s64 += __shfl_xor(s64, o); l64 += __shfl_xor(l64, o);
        const int v = __shfl_xor(mx, o); mx = v > mx ? v : mx;
        const unsigned long long sv = __shfl_xor(steps, o); steps = sv > steps ? sv : steps;
    }
    if (lane == 0) {
        if (p64) atomicAdd(&ct->positions, p64);
        if (s64) atomicAdd(&ct->spills, s64);
        if (l64) atomicAdd(&ct->list_sum, l64);
        atomicAdd(&ct->wave_steps, steps);
        if (c64) atomicAdd(&ct->ext_calls, c64);
        if (r64) atomicAdd(&ct->rec_reads, r64);
        if (t64) atomicAdd(&ct->tab_reads, t64);
        if (tot) atomicAdd(&ct->total, tot);
        if (mx) atomicMax(&ct->max_per_read, mx);
    }
}

// ---- the wide phases: a group of 16 lanes per item, one entry per lane and step ------------------------------------
// Same extensions, same order of decisions as the loop over prev[] of the reference (FMI_search.cpp:589-650):
//     for p: if (first && so < min_intv && long enough) { emit; first = false; }
//            else if (so >= min_intv && so != curr_s)   { keep; first = false; curr_s = (int)so; }
// first is cleared by the first entry that emits or is kept, so only the FIRST entry with (so < min_intv && long enough) or
// (so >= min_intv) can emit, and it does iff it is of the first kind; an entry is kept iff so >= min_intv and so differs
// from (int64)(int)so of the previous entry with so >= min_intv (kept or not: a skipped one equals the value it was
// compared with).  Both are ballots and lane shuffles over the sixteen entries of a step, carried from step to step.
enum WideState : int { W_NEW, W_FWD, W_COL, W_ROUND, W_DONE };
__device__ __forceinline__ void unpack_iv(const uint4 w, int64_t &k, int64_t &l, int64_t &s, int &n) {
    k = (int64_t)((uint64_t)(w.w & 0xffu) << 32 | w.x); l = (int64_t)((uint64_t)((w.w >> 8) & 0xffu) << 32 | w.y);
    s = (int64_t)((uint64_t)((w.w >> 16) & 0xffu) << 32 | w.z); n = (int)(w.w >> 24);
}
__global__ __launch_bounds__(64) void fmi_wide_kernel(FmiIdx ix, const uint8_t *__restrict__ enc, int32_t stride, const int32_t *__restrict__ len_arr,
                                                      int64_t first, const FmiWideItem *__restrict__ items, const int32_t *__restrict__ n_items_dev,
                                                      int32_t items_cap, const uint4 *__restrict__ lists_in, FmiWideItem *__restrict__ cands,
                                                      int32_t *n_cands_dev, int32_t cands_cap, OutRec *__restrict__ out_all, int cap, int32_t *counts,
                                                      FmiCounters *ct, int min_seed_len) {
    // per group ONE list of LL entries: a step reads sixteen entries and then writes its survivors at ranks that are not above
    // the first of them, so the column's list is compacted in place
    extern __shared__ uint4 lst_all[];
    const int LL = (stride + 15) & ~15;
    const int lane = threadIdx.x, g = lane >> 4, gl = lane & 15, lead = g * 16;
    const int split_len = (int)(min_seed_len * 1.5 + .499);
    const int n_items = (uint32_t)*n_items_dev < (uint32_t)items_cap ? *n_items_dev : items_cap;      // (the counter as unsigned: never a negative count)
    uint32_t calls = 0, recs = 0, tabs = 0, dummy = 0;
    unsigned long long tot = 0;
    int mx = 0;
    // group-uniform state (every lane of the group holds the same values)
    int st = W_NEW, t = 0, len = 0, j = 0, a = 4, a_next = 4, nprev = 0, ncur = 0, p0 = 0;
    bool pass1 = false, first_phase = true, have_k = false;
    uint32_t cur_m = 0;
    int64_t min_intv = 1, last_so = 0;
    const uint8_t *rbase = enc;
    // forward phase of a candidate (all lanes of the group compute the same)
    int x = 0, jf = 0, sm_n = 0;
    int64_t sm_k = 0, sm_l = 0, sm_s = 0;

    auto emit = [&](uint32_t m, uint32_t n, int64_t k, int64_t l, int64_t sv) {      // ONE lane of the group calls this
        const int slot = atomicAdd(&counts[t], 1);
        if (slot < cap) { OutRec o; o.m = m; o.n = n; o.k = k; o.l = l; o.s = sv; out_all[(int64_t)t * cap + slot] = o; }
        tot++; mx = slot + 1 > mx ? slot + 1 : mx;
        if (pass1 && cands && (int)(n + 1 - m) >= split_len && sv <= 10) {
            const int ci = atomicAdd(n_cands_dev, 1);
            if (ci < cands_cap) {
                FmiWideItem c; c.t = (uint32_t)t; c.jm = ((n + 1 + m) >> 1) | (uint32_t)(sv + 1) << 16; c.off = c.n = c.cur_m = 0; c.kind = 1; c.pad[0] = c.pad[1] = 0;
                cands[ci] = c;
            }
        }
    };
    auto fwd_push = [&]() { if (gl == 0) lst_all[g * LL + (nprev)] = pack_iv(sm_k, sm_l, sm_s, (uint32_t)sm_n); nprev++; };

    while (__any(st != W_DONE)) {
        __syncthreads();                                     // (one wave) the lists written in the last step are visible
        if (st == W_NEW) {
            int it = 0;
            if (gl == 0) it = atomicAdd(&ct->wide_queue, 1);
            it = __shfl(it, lead);
            if (it >= n_items) st = W_DONE;
            else {
                const FmiWideItem w = items[it];
                if (w.kind <= 1) {
                    t = (int)w.t; len = len_arr[first + t];
                    rbase = enc + (first + (int64_t)t) * (int64_t)stride;
                    min_intv = (int64_t)((w.jm >> 16) & 0x7fffu);
                    a_next = 4;
                    if (w.kind == 0) {
                        pass1 = (w.jm >> 31) != 0;
                        j = (int)(w.jm & 0xffffu) - 1; cur_m = w.cur_m; nprev = (int)w.n;
                        for (int v = gl; v < nprev; v += 16) lst_all[g * LL + (v)] = lists_in[w.off + (uint32_t)v];
                        a = j >= 0 ? (int)rbase[j] : 4;
                        st = W_COL;
                    } else {                                 // getSMEMsOnePosOneThread :496-530 for the candidate
                        pass1 = false;
                        x = (int)(w.jm & 0xffffu);
                        const int a0 = x < len ? (int)rbase[x] : 4;
                        if (a0 <= 3) {
                            sm_n = x; sm_k = ix.count[a0]; sm_l = ix.count[3 - a0]; sm_s = ix.count[a0 + 1] - ix.count[a0];
                            nprev = 0; jf = x + 1;
                            st = W_FWD;
                        }
                    }
                }
            }
        }
        if (st == W_COL) {                                   // backward loop :589-650, one column
            const int ac = a > 3 ? 4 : a;
            if (j >= 0 && ac <= 3) {
                a = ac; ncur = 0; p0 = 0; first_phase = true; have_k = false;
                a_next = j > 0 ? (int)rbase[j - 1] : 4;      // (used a column later)
                st = W_ROUND;
            } else {                                         // the read's start or an N: the longest entry is an SMEM
                if (nprev != 0 && gl == 0) {
                    int64_t k0, l0, s0; int n0;
                    unpack_iv(lst_all[g * LL + (0)], k0, l0, s0, n0);
                    if ((int)((int64_t)n0 - (int64_t)cur_m + 1) >= min_seed_len) emit(cur_m, (uint32_t)n0, k0, l0, s0);
                }
                st = W_NEW;
            }
        }
        // ---- the extension: sixteen entries of the column, or the next base of a candidate's forward phase
        bool need = false, fwd_end = false;
        int64_t K = 0, L = 0, S = 0; int A = 0, n0 = 0;
        if (st == W_ROUND) {
            const int p = p0 + gl;
            if (p < nprev) { unpack_iv(lst_all[g * LL + (p)], K, L, S, n0); A = a; need = true; }
        } else if (st == W_FWD) {                            // forward loop :531-575
            if (jf >= len) fwd_end = true;
            else {
                const int af = (int)rbase[jf];
                if (af > 3) fwd_end = true; else { K = sm_l; L = sm_k; S = sm_s; A = 3 - af; need = true; }
            }
        }
        int64_t ko = 0, lo = 0, so = 0;
        if (need) {
            if (st == W_ROUND || gl == 0) extend(ix, K, L, S, A, false, 0u, false, ko, lo, so, calls, recs, tabs);
            else extend(ix, K, L, S, A, false, 0u, false, ko, lo, so, dummy, dummy, dummy);
        }
        // ---- consume
        if (st == W_FWD) {
            if (need) {
                if (so != sm_s) fwd_push();
                if (so < min_intv) fwd_end = true;
                else { sm_k = lo; sm_l = ko; sm_s = so; sm_n = jf; jf++; }
            }
            if (fwd_end) {
                if (sm_s >= min_intv) fwd_push();
                if (nprev == 0) st = W_NEW;
                else {
                    // longest match first, as the backward columns want the list (prev[] is turned round in the reference too,
                    // :577-587); lane 0 wrote the entries and the loop top's barrier has not come yet: it does the swaps as well
                    if (gl == 0) for (int v = 0; v < nprev / 2; v++) { const uint4 tmp = lst_all[g * LL + v]; lst_all[g * LL + v] = lst_all[g * LL + nprev - 1 - v]; lst_all[g * LL + nprev - 1 - v] = tmp; }
                    j = x - 1; cur_m = (uint32_t)x; a = j >= 0 ? (int)rbase[j] : 4; st = W_COL;
                }
            }
        } else if (st == W_ROUND) {
            const bool lenok = need && (int)((int64_t)n0 - (int64_t)cur_m + 1) >= min_seed_len;
            const bool isE = need && so < min_intv && lenok, isK = need && so >= min_intv;
            const uint32_t mE = (uint32_t)(__ballot(isE) >> lead) & 0xffffu, mK = (uint32_t)(__ballot(isK) >> lead) & 0xffffu;
            if (first_phase && (mE | mK)) {
                const int f = __builtin_ctz(mE | mK);
                if ((mE >> f) & 1u) { if (gl == f) emit(cur_m, (uint32_t)n0, K, L, S); }
                first_phase = false;
            }
            // the value an entry with so >= min_intv is compared with: (int64)(int)so of the one before it
            const uint32_t below = mK & ((1u << gl) - 1u);
            const int src = below ? 31 - __builtin_clz(below) : 0;
            const int64_t so_prev = __shfl(so, lead + src);
            const bool has_prev = below != 0 || have_k;
            const int64_t cmp = below ? so_prev : last_so;
            const bool keep = isK && (!has_prev || so != (int64_t)(int)cmp);
            const uint32_t mKeep = (uint32_t)(__ballot(keep) >> lead) & 0xffffu;
            if (keep) lst_all[g * LL + (ncur + __builtin_popcount(mKeep & ((1u << gl) - 1u)))] = pack_iv(ko, lo, so, (uint32_t)n0);
            ncur += __builtin_popcount(mKeep);
            if (mK) { const int hi = 31 - __builtin_clz(mK); last_so = __shfl(so, lead + hi); have_k = true; }
            p0 += 16;
            if (p0 >= nprev) {                               // column done
                nprev = ncur;
                if (ncur == 0) st = W_NEW;
                else { cur_m = (uint32_t)j; j--; a = a_next; st = W_COL; }
            }
        }
    }
    unsigned long long c64 = calls, r64 = recs, t64 = tabs;
    for (int o = 32; o > 0; o >>= 1) {
        c64 += __shfl_xor(c64, o); r64 += __shfl_xor(r64, o); t64 += __shfl_xor(t64, o); tot += __shfl_xor(tot, o);
        const int v = __shfl_xor(mx, o); mx = v > mx ? v : mx;
    }
    if (lane == 0) {
        if (c64) atomicAdd(&ct->ext_calls, c64);
        if (r64) atomicAdd(&ct->rec_reads, r64);
        if (t64) atomicAdd(&ct->tab_reads, t64);
        if (tot) atomicAdd(&ct->total, tot);
        if (mx) atomicMax(&ct->max_per_read, mx);
    }
}

__device__ __forceinline__ bool rec_less(const OutRec &a, const OutRec &b) {
    if (a.m != b.m) return a.m < b.m;
    if (a.n != b.n) return a.n > b.n;            // compare_smem: n descending
    if (a.s != b.s) return a.s < b.s;            // tie-break (unspecified in the reference)
    return a.k < b.k;
}
// sortSMEMs (:986-1022) per read: insertion sort inside the read's slot (a handful of records)
__global__ __launch_bounds__(256) void fmi_sort_slots(OutRec *out_all, int cap, const int32_t *counts, int32_t nbatch,
                                                      const int32_t *__restrict__ ids) {
    const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nbatch) return;
    const int nout = counts[ids ? ids[t] : t];
    if (nout > cap) return;                       // truncated slot: the read gets a second round with a bigger one
    OutRec *out = out_all + (int64_t)t * cap;
    for (int i = 1; i < nout; i++) {
        const OutRec key = out[i];
        int j = i - 1;
        while (j >= 0) {
            const OutRec o = out[j];
            if (!rec_less(key, o)) break;
            out[j + 1] = o;
            j--;
        }
        out[j + 1] = key;
    }
}

// ---- exclusive scan of the per-read counts and compaction ------------------------------------------
__global__ __launch_bounds__(256) void fmi_block_sums(const int32_t *counts, int32_t n, int64_t *block_sums) {
    __shared__ int64_t sh[256];
    const int32_t t = blockIdx.x * 256 + threadIdx.x;
    sh[threadIdx.x] = t < n ? counts[t] : 0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(1024) void fmi_scan_blocks(int64_t *block_sums, int32_t nblocks, int64_t base) {
    // single workgroup, serial over chunks of 1024
    __shared__ int64_t sh[1024];
    int64_t carry = base;
    for (int32_t c0 = 0; c0 < nblocks; c0 += 1024) {
        const int32_t i = c0 + threadIdx.x;
        const int64_t v = i < nblocks ? block_sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int64_t add = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nblocks) block_sums[i] = carry + sh[threadIdx.x] - v;
        const int64_t total = sh[1023];
        __syncthreads();
        carry += total;
    }
}
__global__ __launch_bounds__(256) void fmi_compact(const int32_t *counts, int32_t n, const int64_t *block_off,
                                                   const OutRec *slots, int cap, int64_t first, int64_t *read_off,
                                                   gab_smem *out) {
    __shared__ int64_t sh[256];
    const int32_t t = blockIdx.x * 256 + threadIdx.x;
    const int c = t < n ? counts[t] : 0;
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int64_t add = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    if (t >= n) return;
    const int64_t off = block_off[blockIdx.x] + sh[threadIdx.x] - c;
    read_off[first + t] = off;
    if (c > cap) return;                          // its records come from the second round (fmi_compact_overflowed)
    const OutRec *src = slots + (int64_t)t * cap;
    for (int j = 0; j < c; j++) {
        const OutRec o = src[j];
        gab_smem g;
        g.rid = (uint32_t)(first + t); g.m = o.m; g.n = o.n; g.pad = 0; g.k = o.k; g.l = o.l; g.s = o.s;
        out[off + j] = g;
    }
}
// The reads of a batch whose SMEMs did not fit the first-round slot (a handful of low-complexity reads among millions):
// their numbers in batch order, for a second round of the seeding kernel over them alone with slots of the exact size.
__global__ __launch_bounds__(256) void fmi_list_overflowed(const int32_t *counts, int32_t n, int cap, int32_t *ids, int32_t *n_ids) {
    const int32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < n && counts[t] > cap) ids[atomicAdd(n_ids, 1)] = t;
}
__global__ __launch_bounds__(256) void fmi_compact_overflowed(const int32_t *counts, const int32_t *ids, int32_t n_ids, const OutRec *slots,
                                                              int cap, int64_t first, const int64_t *read_off, gab_smem *out) {
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_ids) return;
    const int32_t t = ids[i];
    const int c = counts[t];
    const int64_t off = read_off[first + t];
    const OutRec *src = slots + (int64_t)i * cap;
    for (int j = 0; j < c; j++) {
        const OutRec o = src[j];
        gab_smem g;
        g.rid = (uint32_t)(first + t); g.m = o.m; g.n = o.n; g.pad = 0; g.k = o.k; g.l = o.l; g.s = o.s;
        out[off + j] = g;
    }
}

__global__ __launch_bounds__(256) void fmi_validate(const int32_t *len, int64_t n, int32_t stride, FmiCounters *ct) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t s = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) {
        const int l = len[i];
        if (l < 0 || l > stride || l >= GAB_FMI_MAX_READLEN) {
            atomicAdd(&ct->bad, 1);
            atomicMin((unsigned int *)&ct->first_bad, (unsigned int)(i + 1 > 0x7fffffff ? 0x7fffffff : i + 1));
        }
    }
}

// ---- suffix-array look-up -------------------------------------------------------------------
// get_sa_entries(SMEM*, ..., max_occ, tid) FMI_search.cpp:1177-1196 on get_sa_entry_compressed :1103-1175.
struct SaIdx { const int8_t *ms; const uint32_t *ls; };
struct SaCounters { unsigned long long lf_steps; int32_t next_chunk; int32_t pad; };
constexpr int kSaChunk = 4096;            // SMEMs a wave takes at a time

__device__ __forceinline__ int sa_count_of(int64_t s, int32_t max_occ) {
    if (s <= 0) return 0;
    const int64_t step = s > max_occ ? s / max_occ : 1;
    const int64_t c = (s + step - 1) / step;                  // rows k, k+step, ... below k+s
    return (int)(c < max_occ ? c : max_occ);
}
__global__ __launch_bounds__(256) void fmi_sa_count(const gab_smem *__restrict__ sm, int64_t n, int32_t max_occ,
                                                    int32_t *counts) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) counts[i] = sa_count_of(sm[i].s, max_occ);
}
// exclusive offsets from the counts and the scanned block sums (same scan kernels as the SMEM compaction)
__global__ __launch_bounds__(256) void fmi_sa_offsets(const int32_t *counts, int64_t n, const int64_t *block_off, int64_t *coord_off) {
    __shared__ int32_t sh[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = i < n ? counts[i] : 0;
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int v = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    if (i < n) coord_off[i] = block_off[blockIdx.x] + sh[threadIdx.x] - c;
}
// One lane resolves one BWT row at a time and immediately takes the next row (of the same SMEM, else the next SMEM of
// the wave's chunk) when it is done: LF walks have a geometric length (mean 7 steps, no bound), so lanes that waited
// for the longest walk of a fixed assignment would idle most of the time.  Every wave step is one LF step = one random
// CP_OCC record for every lane that has a row.
__global__ __launch_bounds__(64) void fmi_sa_kernel(FmiIdx ix, SaIdx sa, const gab_smem *__restrict__ sm, int64_t n, int32_t max_occ,
                                                    const int64_t *__restrict__ coord_off, int64_t *coords, SaCounters *ct) {
    __shared__ int chunk_next;
    const int lane = threadIdx.x;
    unsigned long long steps = 0;
    for (;;) {
        // ---- the wave takes its next chunk of SMEMs
        int64_t c0;
        {
            int v = 0;
            if (lane == 0) v = atomicAdd(&ct->next_chunk, 1);
            v = __shfl(v, 0);
            c0 = (int64_t)v * kSaChunk;
        }
        if (c0 >= n) break;
        const int64_t c1 = c0 + kSaChunk < n ? c0 + kSaChunk : n;
        if (lane == 0) chunk_next = 0;
        __syncthreads();
        // ---- lane state: the SMEM it works on and the row of it being resolved
        bool have = false;                 // a row is in flight
        int cnt = 0, t = 0;                // rows of the current SMEM, next row index
        int64_t k = 0, stp = 1, out = 0;   // first row, row stride, output position of row 0
        int64_t sp = 0, offset = 0;
        bool exhausted = false;
        for (;;) {
            if (!have) {
                // next row of the current SMEM, else next SMEM(s) of the chunk
                while (t >= cnt && !exhausted) {
                    const int64_t i = c0 + atomicAdd(&chunk_next, 1);
                    if (i >= c1) { exhausted = true; break; }
                    const gab_smem s = sm[i];
                    cnt = sa_count_of(s.s, max_occ); t = 0;
                    k = s.k; stp = s.s > max_occ ? s.s / max_occ : 1; out = coord_off[i];
                }
                if (t < cnt) { sp = k + (int64_t)t * stp; offset = 0; have = true; }
            }
            if (__all(!have)) break;       // chunk done (every lane is out of rows)
            if (have) {
                if ((sp & 7) == 0) {
                    coords[out + t] = ((int64_t)sa.ms[sp >> 3] << 32) + (int64_t)sa.ls[sp >> 3] + offset;
                    t++; have = false;
                } else {
                    int64_t cnt4[4]; uint64_t bits[4];
                    load_rec(ix.cp_occ + (sp >> 6), cnt4, bits);
                    const int y = 63 - (int)(sp & 63);
                    const int b = (bits[0] >> y) & 1 ? 0 : (bits[1] >> y) & 1 ? 1 : (bits[2] >> y) & 1 ? 2 : (bits[3] >> y) & 1 ? 3 : 4;
                    if (b == 4) { coords[out + t] = offset; t++; have = false; }       // the sentinel row
                    else {
                        const int ysp = (int)(sp & 63);
                        const uint64_t m = ysp ? ~0ull << (64 - ysp) : 0ull;
                        const uint64_t bb = b == 0 ? bits[0] : b == 1 ? bits[1] : b == 2 ? bits[2] : bits[3];
                        const int64_t cc = b == 0 ? cnt4[0] : b == 1 ? cnt4[1] : b == 2 ? cnt4[2] : cnt4[3];
                        sp = ix.count[b] + cc + __popcll(bb & m);
                        offset++; steps++;
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) steps += __shfl_xor(steps, o);
    if (lane == 0 && steps) atomicAdd(&ct->lf_steps, steps);
}

}  // namespace

// =============================================================================== host side
struct gab_fmi {
    gab_tuning tun = gab_tuning_loaded();      // experiment knobs, read when the handle is made (an fmi handle never re-reads them)
    gab_host_stream hs;     // private stream of the host-pointer entry point(s)
    int device = 0;
    FmiIdx ix;
    gab_devbuf index;       // CP_OCC array
    gab_devbuf kmer;        // short-pattern table
    gab_devbuf ws;          // counters, counts, block sums
    gab_devbuf prev;        // prev[] scratch
    gab_devbuf slots;       // per-read output slots
    gab_devbuf slots2, ovf; // second round: slots of the exact size for the reads that overflowed theirs, and their numbers
    gab_devbuf witems, wlists, wcands;   // wide backward phases handed over to fmi_wide_kernel: items, their lists, the candidates they find
    int wide_cap_env = 0;   // $GAB_FMI_WIDE_CAP: places of the hand-over item / candidate queues (tests: the queues run full)
    int handover_env = 1;   // $GAB_FMI_WIDE=0: no hand-over (every phase stays with the lane that owns the read); > 1: the survivor count that makes a phase wide
    gab_devbuf out;         // compacted SMEMs
    gab_devbuf roff;        // read_off (nreads + 1)
    gab_devbuf io;          // staging for the host entry point
    gab_devbuf sa;          // sampled suffix array: int8 ms bytes, then uint32 low words
    gab_devbuf sa_ws, sa_off, sa_coords, sa_io;
    SaIdx sa_ix = {nullptr, nullptr};
    int64_t sa_lf_steps = 0; float sa_ms = 0; bool have_sa_stats = false;
    size_t scratch_budget = (size_t)6 << 30;   // bound of the two big scratch areas (per-lane spill lists, per-read output slots)
    bool scratch_from_env = false;             // $GAB_FMI_SCRATCH_MB set: never look at the free memory for a bigger one
    int lds_entries_env = 0;                   // $GAB_FMI_LDS_ENTRIES / $GAB_FMI_WIDE_LISTS as they were when the handle was made
    int64_t batch_max = 1 << 24;               // $GAB_FMI_BATCH: reads per launch of the seeding kernel
    int waves_env = 0;                         // $GAB_FMI_WAVES: cap of the resident waves per CU of passes 1 + 2 (experiments)
    bool wide_env = false;                     // (tests and bench.py: force the list format of indexes with >= 2^32 rows)
    hipEvent_t ev[2] = {nullptr, nullptr};
    FmiCounters *h_ct = nullptr;
    bool have_stats = false;
    int64_t ext_calls = 0, rec_reads = 0, nsmem = 0;
    float kernel_ms = 0;
};

static int fmi_new_handle(int device, gab_fmi **out) {
    int rc = gab_check_device(device);
    if (rc) return rc;
    gab_fmi *h = new (std::nothrow) gab_fmi();
    if (!h) { gab_set_error("out of host memory"); return GAB_ENOMEM; }
    h->device = device;
    const gab_tuning &t = h->tun;
    h->lds_entries_env = t.fmi_lds_entries; h->wide_env = t.fmi_wide_lists; h->waves_env = t.fmi_waves;
    h->handover_env = t.fmi_wide; h->wide_cap_env = t.fmi_wide_cap;
    if (t.fmi_batch >= 1024) h->batch_max = t.fmi_batch;
    if (t.fmi_scratch_mb > 0) { h->scratch_budget = (size_t)t.fmi_scratch_mb << 20; h->scratch_from_env = true; }
    if (hipEventCreate(&h->ev[0]) != hipSuccess || hipEventCreate(&h->ev[1]) != hipSuccess ||
        hipHostMalloc((void **)&h->h_ct, sizeof(FmiCounters)) != hipSuccess) {
        gab_set_error("gab_fmi: event / pinned allocation failed"); delete h; return GAB_EDEVICE;
    }
    *out = h;
    return GAB_OK;
}

extern "C" int gab_fmi_create(int device, int64_t ref_seq_len, const int64_t count_file[5], const void *cp_occ,
                              int64_t sentinel_index, gab_fmi **out) {
    if (!out || !count_file || !cp_occ) { gab_set_error("gab_fmi_create: NULL argument"); return GAB_EINVAL; }
    *out = nullptr;
    GAB_CHECK(ref_seq_len > 0 && ref_seq_len <= 0x7fffffffffll, "gab_fmi_create: reference_seq_len out of range");
    GAB_CHECK(sentinel_index >= 0 && sentinel_index < ref_seq_len, "gab_fmi_create: sentinel index out of range");
    gab_fmi *h = nullptr;
    int rc = fmi_new_handle(device, &h);
    if (rc) return rc;
    gab_device_guard g(device);
    const size_t bytes = sizeof(CpOcc) * (size_t)((ref_seq_len >> 6) + 1);
    rc = h->index.reserve(bytes);
    if (rc) { gab_fmi_destroy(h); return rc; }
    if (hipMemcpy(h->index.p, cp_occ, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        gab_set_error("gab_fmi_create: index upload failed"); gab_fmi_destroy(h); return GAB_EDEVICE;
    }
    h->ix.cp_occ = h->index.as<CpOcc>();
    for (int i = 0; i < 5; i++) h->ix.count[i] = count_file[i] + 1;     // FMI_search.cpp:433-436
    h->ix.sentinel = sentinel_index;
    h->ix.ref_seq_len = ref_seq_len;
    h->ix.kmer_tab = nullptr; h->ix.kmer_depth = 0;
    {   // short-pattern table, level by level (each level extends the previous one by one base to the left)
        const int depth = h->tun.fmi_kmer_depth >= 0 ? std::min(11, h->tun.fmi_kmer_depth) : kDefaultKmerDepth;      // GAB_FMI_KMER_DEPTH
        if (depth > 0) {
            const size_t entries = ((((size_t)1 << (2 * (depth + 1))) - 4) / 3 + 3) & ~(size_t)3;
            rc = h->kmer.reserve(entries * sizeof(uint4));
            if (rc) { gab_fmi_destroy(h); return rc; }
            if (hipMemset(h->kmer.p, 0, entries * sizeof(uint4)) != hipSuccess) { gab_set_error("gab_fmi_create: memset failed"); gab_fmi_destroy(h); return GAB_EDEVICE; }
            for (int len = 1; len <= depth; len++) {
                const unsigned n = 1u << (2 * len);
                hipLaunchKernelGGL(fmi_build_kmer_level, dim3((n + 255) / 256), dim3(256), 0, nullptr, h->ix, h->kmer.as<uint4>(), len);
            }
            if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) {
                gab_set_error("gab_fmi_create: building the short-pattern table failed"); gab_fmi_destroy(h); return GAB_EDEVICE;
            }
            h->ix.kmer_tab = h->kmer.as<uint4>(); h->ix.kmer_depth = depth;
        }
    }
    *out = h;
    return GAB_OK;
}

extern "C" int gab_fmi_load(int device, const char *prefix, gab_fmi **out) {
    if (!out || !prefix) { gab_set_error("gab_fmi_load: NULL argument"); return GAB_EINVAL; }
    *out = nullptr;
    char name[4096];
    snprintf(name, sizeof name, "%s.bwt.2bit.64", prefix);
    FILE *f = fopen(name, "rb");
    if (!f) { gab_set_error("gab_fmi_load: cannot open %s", name); return GAB_EINVAL; }
    int64_t ref_seq_len = 0, count[5], sentinel = -1;
    int rc = GAB_OK;
    void *host = nullptr, *sa_host = nullptr;
    do {
        if (fread(&ref_seq_len, 8, 1, f) != 1 || fread(count, 8, 5, f) != 5 || ref_seq_len <= 0 ||
            ref_seq_len > 0x7fffffffffll) { gab_set_error("gab_fmi_load: %s: bad header", name); rc = GAB_EINVAL; break; }
        const size_t n_occ = (size_t)((ref_seq_len >> 6) + 1), bytes = n_occ * sizeof(CpOcc);
        host = malloc(bytes);
        if (!host) { gab_set_error("gab_fmi_load: out of host memory (%zu bytes)", bytes); rc = GAB_ENOMEM; break; }
        if (fread(host, sizeof(CpOcc), n_occ, f) != n_occ) { gab_set_error("gab_fmi_load: %s: truncated", name); rc = GAB_EINVAL; break; }
        // the sampled suffix array (SA_COMPX = 3: one int8 + one uint32 per 8 rows), FMI_search.cpp:439-447; seeding
        // does not need it, gab_fmi_sa_lookup does
        const int64_t n_sa = (ref_seq_len >> 3) + 1;
        sa_host = malloc((size_t)n_sa * 5);
        if (!sa_host) { gab_set_error("gab_fmi_load: out of host memory"); rc = GAB_ENOMEM; break; }
        if (fread(sa_host, 1, (size_t)n_sa * 5, f) != (size_t)n_sa * 5 || fread(&sentinel, 8, 1, f) != 1) {
            gab_set_error("gab_fmi_load: %s: truncated (suffix array / sentinel index)", name); rc = GAB_EINVAL; break;
        }
        rc = gab_fmi_create(device, ref_seq_len, count, host, sentinel, out);
        if (rc == GAB_OK) {
            rc = gab_fmi_set_sa(*out, (const int8_t *)sa_host, (const uint32_t *)((const char *)sa_host + n_sa));
            if (rc) { gab_fmi_destroy(*out); *out = nullptr; }
        }
    } while (0);
    free(host); free(sa_host);
    fclose(f);
    return rc;
}

// A second handle on the same GPU that shares the read-only index of `src` (CP_OCC, short-pattern table, sampled suffix
// array: the clone's devbufs for them stay empty, its FmiIdx / SaIdx point into src's) and owns only its work buffers.
extern "C" int gab_fmi_clone(gab_fmi *src, gab_fmi **out) {
    if (!src || !out) { gab_set_error("gab_fmi_clone: NULL argument"); return GAB_EINVAL; }
    *out = nullptr;
    gab_fmi *h = nullptr;
    int rc = fmi_new_handle(src->device, &h);
    if (rc) return rc;
    h->ix = src->ix; h->sa_ix = src->sa_ix; h->scratch_budget = src->scratch_budget; h->scratch_from_env = src->scratch_from_env; h->batch_max = src->batch_max; h->waves_env = src->waves_env; h->handover_env = src->handover_env; h->wide_cap_env = src->wide_cap_env;
    *out = h;
    return GAB_OK;
}

extern "C" void gab_fmi_destroy(gab_fmi *h) {
    if (!h) return;
    gab_device_guard g(h->device);
    h->index.release(); h->kmer.release(); h->sa.release(); h->sa_ws.release(); h->sa_off.release(); h->sa_coords.release(); h->sa_io.release(); h->ws.release(); h->prev.release(); h->slots.release(); h->slots2.release(); h->ovf.release(); h->witems.release(); h->wlists.release(); h->wcands.release(); h->out.release(); h->roff.release();
    h->io.release(); h->hs.release();
    for (int k = 0; k < 2; k++) if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
    if (h->h_ct) (void)hipHostFree(h->h_ct);
    delete h;
}

extern "C" int gab_fmi_seed_device(gab_fmi *h, const uint8_t *d_enc, int32_t stride, const int32_t *d_len, int64_t nreads,
                                   int32_t min_seed_len, const gab_smem **d_out, const int64_t **d_read_off,
                                   int64_t *nout, void *stream_) {
    GAB_CHECK(h, "gab_fmi_seed_device: NULL handle");
    GAB_CHECK(nreads >= 0 && nreads < (1ll << 31), "gab_fmi_seed_device: nreads out of range");
    GAB_CHECK(stride > 0 && stride < GAB_FMI_MAX_READLEN, "gab_fmi_seed_device: stride (max read length) must be in 1..%d",
              GAB_FMI_MAX_READLEN - 1);
    GAB_CHECK(min_seed_len >= 1, "gab_fmi_seed_device: minSeedLen must be >= 1");
    GAB_CHECK(nout, "gab_fmi_seed_device: NULL nout");
    h->have_stats = false;
    *nout = 0;
    gab_device_guard g(h->device);
    hipStream_t s = (hipStream_t)stream_;
    int rc = h->roff.reserve(8 * (size_t)(nreads + 1));
    if (rc) return rc;
    if (d_read_off) *d_read_off = h->roff.as<int64_t>();
    if (d_out) *d_out = nullptr;
    if (nreads == 0) { GAB_HIP(hipMemsetAsync(h->roff.p, 0, 8, s)); return GAB_OK; }
    GAB_CHECK(d_enc && d_len, "gab_fmi_seed_device: NULL buffer");

    // The seeding kernel is a persistent grid of one-wave blocks sized to the occupancy; lanes pull reads from a
    // counter.  Scratch: a spill area for interval lists longer than the LDS part (stride entries x 32 B per LANE,
    // touched only by the rare long list) and the output slot per READ (cap x 32 B).
    // A batch cannot end before its slowest read has (passes 1 + 2 of a read are one chain of dependent look-ups: ~750 wave steps
    // on average, but ONE backward phase of a re-seeded position inside a repeat -- a 76-entry list over 75 columns -- is 5 700,
    // ~25 ms), and while the last reads finish most lanes idle: 10 M reads in four batches spent 4 x ~22 ms of 322 ms there
    // (profiles/r03_fmi_batches.md).  So the whole call is ONE batch whenever its slots (cap x 32 B per read) fit a third of
    // the memory that is free right now, up to 32 GB; $GAB_FMI_SCRATCH_MB fixes the budget instead.
    size_t budget = h->scratch_budget;
    if (!h->scratch_from_env && (size_t)nreads * 48 * sizeof(OutRec) > budget) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
            const size_t avail = (fr + h->slots.cap) / 3;
            budget = std::max(budget, std::min<size_t>(avail, (size_t)32 << 30));
        }
    }
    const int cap = 48;
    const int lds_entries_env = h->lds_entries_env;
    const bool ldsq = stride <= kLdsQMax;
    const int lds_entries = ldsq ? (lds_entries_env > 0 ? lds_entries_env : 12) : 0;
    // list entries: 13 bytes (three dword planes + a byte plane) when every interval bound fits 32 bits, else 16 packed
    const bool wide_env = h->wide_env;
    const int narrow_lists = (h->ix.ref_seq_len < 0xffffffffll && !wide_env) ? 1 : 0;
    const size_t list_bytes = narrow_lists ? (((size_t)lds_entries * 64 * 13 + 15) & ~(size_t)15) : (size_t)lds_entries * 64 * 16;
    const size_t lds_bytes = ldsq ? (((size_t)(stride + 7) / 8 * 64 + 3) / 4) * 16 + list_bytes : 0;
    const size_t lds_bytes_p3 = ldsq ? (((size_t)(stride + 7) / 8 * 64 + 3) / 4 + 64) * 16 : 0;   // read + slack for nibbles_at
    int waves_per_cu = 0, waves_per_cu_p3 = 0, n_cu = 0, wide_occ = 1;
    const size_t wide_lds = (size_t)4 * (size_t)((stride + 15) & ~15) * sizeof(uint4);          // fmi_wide_kernel: four groups, a list each
    {
        hipDeviceProp_t prop;
        GAB_HIP(hipGetDeviceProperties(&prop, h->device));
        n_cu = prop.multiProcessorCount;
        if (ldsq) {
            GAB_HIP(hipFuncSetAttribute((const void *)fmi_seed_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            GAB_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&waves_per_cu, fmi_seed_kernel<true>, 64, lds_bytes));
            GAB_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&waves_per_cu_p3, fmi_seed_kernel<true>, 64, lds_bytes_p3));
            GAB_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&wide_occ, fmi_wide_kernel, 64, wide_lds));
            if (wide_occ < 1) wide_occ = 1;
        } else GAB_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&waves_per_cu, fmi_seed_kernel<false>, 64, 0));
        GAB_CHECK(waves_per_cu > 0, "gab_fmi_seed_device: the seeding kernel does not fit a CU (stride %d)", stride);
        // the per-lane spill area is stride x 32 B x 64 lanes per resident wave: long reads (no LDS lists: every list entry
        // lives there) run on fewer waves rather than on tens of GB of scratch
        const int64_t fit = (int64_t)(budget / (sizeof(PrevRec) * (size_t)stride * 64)) / n_cu;
        if (fit < waves_per_cu) waves_per_cu = (int)std::max<int64_t>(fit, 1);
        if (h->waves_env > 0 && h->waves_env < waves_per_cu) waves_per_cu = h->waves_env;
        if (fit < waves_per_cu_p3) waves_per_cu_p3 = (int)std::max<int64_t>(fit, 1);
    }
    const int64_t grid_waves = (int64_t)n_cu * std::max(waves_per_cu, waves_per_cu_p3);
    int64_t B = (int64_t)std::min<size_t>((size_t)nreads, std::max<size_t>(1024, budget / (64 * sizeof(OutRec))));
    B = std::min<int64_t>(B, h->batch_max);
    B = gab_ceil_div(nreads, gab_ceil_div(nreads, B));       // equal batches: no short last one
    const int64_t nb_blocks = gab_ceil_div(B, 256);
    const size_t o_counts = 256, o_bs = o_counts + 4 * (size_t)B + 64;
    rc = h->ws.reserve(o_bs + 8 * (size_t)nb_blocks + 64);
    if (rc) return rc;
    rc = h->prev.reserve(sizeof(PrevRec) * (size_t)stride * (size_t)grid_waves * 64);
    if (rc) return rc;
    char *wb = h->ws.as<char>();
    FmiCounters *d_ct = (FmiCounters *)wb;
    int32_t *d_counts = (int32_t *)(wb + o_counts);
    int64_t *d_bs = (int64_t *)(wb + ((o_bs + 7) & ~(size_t)7));

    memset(h->h_ct, 0, sizeof(FmiCounters));
    h->h_ct->first_bad = 0x7fffffff;
    GAB_HIP(hipMemcpyAsync(d_ct, h->h_ct, sizeof(FmiCounters), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(fmi_validate, dim3((unsigned)std::min<int64_t>(gab_ceil_div(nreads, 256), 2048)), dim3(256), 0, s,
                       d_len, nreads, stride, d_ct);
    GAB_HIP(hipMemcpyAsync(h->h_ct, d_ct, sizeof(FmiCounters), hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    if (h->h_ct->bad) {
        gab_set_error("gab_fmi_seed_device: %d read(s) have a length outside 0..stride (first: read %d)", h->h_ct->bad,
                      h->h_ct->first_bad - 1);
        return GAB_EINVAL;
    }

    // output grows geometrically; batches append
    size_t out_cap = std::max<size_t>((size_t)nreads * 16, 1024);
    rc = h->out.reserve(out_cap * sizeof(gab_smem));
    if (rc) return rc;
    int64_t total = 0;
    unsigned long long ext_total = 0, rec_total = 0;
    float kms = 0;
    // one launch of the seeding kernel(s) over `count` reads of the batch at `first`: all of them (ids == nullptr, slot t for
    // read t) or the listed ones (slot i for read ids[i])
    bool handover = h->handover_env != 0;                   // wide backward phases go to fmi_wide_kernel (off for the rest of a call whose candidates outgrew their queue)
    auto wide_cap = [&](int64_t count) { return h->wide_cap_env > 0 ? (int32_t)h->wide_cap_env : (int32_t)std::min<int64_t>(count * 3 / 10 + 4096, 1 << 28); };
    auto seed = [&](int64_t first, int32_t nb, int32_t count, const int32_t *ids, OutRec *slots, int slot_cap, int *waves_dbg) -> int {
        const int seed_blocks = (int)std::min<int64_t>((int64_t)n_cu * waves_per_cu, gab_ceil_div((int64_t)count, 64));
        if (waves_dbg) *waves_dbg = seed_blocks;
        GAB_HIP(hipMemsetAsync(&d_ct->next_read, 0, sizeof(int32_t), s));
        if (ldsq) {
            // passes 1 + 2 need the interval lists in LDS, which caps the occupancy; pass 3 needs only the read, so it
            // runs as a second launch of the same kernel with no list area and twice the waves
            // wide backward phases leave the kernel as items (see FmiWideItem): the areas for a batch of `count` reads
            FmiWide wide{nullptr, nullptr, 0, 0, 0};
            const int32_t wcap = wide_cap(count);
            if (!ids && handover) {
                const uint32_t lcap = (uint32_t)std::min<int64_t>((int64_t)count * 12 + 65536, 0x7fffffffll);
                int rcw = h->witems.reserve(sizeof(FmiWideItem) * (size_t)wcap);
                if (!rcw) rcw = h->wcands.reserve(sizeof(FmiWideItem) * (size_t)wcap);
                if (!rcw) rcw = h->wlists.reserve(sizeof(uint4) * (size_t)lcap);
                if (rcw) return rcw;
                wide = FmiWide{h->witems.as<FmiWideItem>(), h->wlists.as<uint4>(), wcap, lcap, h->handover_env > 1 ? h->handover_env : kWideMin};
            }
            hipLaunchKernelGGL(fmi_seed_kernel<true>, dim3(seed_blocks), dim3(64), lds_bytes, s, h->ix, d_enc, stride, d_len, first,
                               count, min_seed_len, h->prev.as<PrevRec>(), (int)stride, slots, slot_cap, d_counts, d_ct,
                               lds_entries, (int64_t)nreads * stride, 1, narrow_lists, ids, wide);
            if (wide.items) {
                // ... and are walked by groups of 16 lanes: the items, then the re-seeding candidates their pass-1 phases found
                const int wblocks = n_cu * wide_occ;
                hipLaunchKernelGGL(fmi_wide_kernel, dim3(wblocks), dim3(64), wide_lds, s, h->ix, d_enc, stride, d_len, first, (const FmiWideItem *)wide.items,
                                   (const int32_t *)&d_ct->wide_items, wcap, (const uint4 *)wide.lists, h->wcands.as<FmiWideItem>(), &d_ct->wide_cands, wcap,
                                   slots, slot_cap, d_counts, d_ct, min_seed_len);
                GAB_HIP(hipMemsetAsync(&d_ct->wide_queue, 0, sizeof(int32_t), s));
                hipLaunchKernelGGL(fmi_wide_kernel, dim3(wblocks), dim3(64), wide_lds, s, h->ix, d_enc, stride, d_len, first, (const FmiWideItem *)h->wcands.as<FmiWideItem>(),
                                   (const int32_t *)&d_ct->wide_cands, wcap, (const uint4 *)wide.lists, (FmiWideItem *)nullptr, (int32_t *)nullptr, 0,
                                   slots, slot_cap, d_counts, d_ct, min_seed_len);
            }
            GAB_HIP(hipMemsetAsync(&d_ct->next_read, 0, sizeof(int32_t), s));
            const int p3_blocks = (int)std::min<int64_t>((int64_t)n_cu * waves_per_cu_p3, gab_ceil_div((int64_t)count, 64));
            hipLaunchKernelGGL(fmi_seed_kernel<true>, dim3(p3_blocks), dim3(64), lds_bytes_p3, s, h->ix, d_enc, stride, d_len, first,
                               count, min_seed_len, h->prev.as<PrevRec>(), (int)stride, slots, slot_cap, d_counts, d_ct,
                               0, (int64_t)nreads * stride, 2, narrow_lists, ids, FmiWide{nullptr, nullptr, 0, 0, 0});
        } else
            hipLaunchKernelGGL(fmi_seed_kernel<false>, dim3(seed_blocks), dim3(64), 0, s, h->ix, d_enc, stride, d_len, first, count,
                               min_seed_len, h->prev.as<PrevRec>(), (int)stride, slots, slot_cap, d_counts, d_ct, 0,
                               (int64_t)nreads * stride, 3, 0, ids, FmiWide{nullptr, nullptr, 0, 0, 0});
        hipLaunchKernelGGL(fmi_sort_slots, dim3((unsigned)gab_ceil_div((int64_t)count, 256)), dim3(256), 0, s, slots, slot_cap, d_counts, count, ids);
        GAB_HIP(hipGetLastError());
        (void)nb;
        return GAB_OK;
    };
    for (int64_t first = 0; first < nreads;) {
        const int32_t nb = (int32_t)std::min<int64_t>(B, nreads - first);
        const int blocks = (int)gab_ceil_div(nb, 256);
        int seed_blocks_dbg = 0;
        rc = h->slots.reserve(sizeof(OutRec) * (size_t)cap * (size_t)nb);
        if (rc) return rc;
        for (;;) {                                            // twice only when the handed-over phases found more candidates than their queue holds
            h->h_ct->ext_calls = 0; h->h_ct->rec_reads = 0; h->h_ct->tab_reads = 0; h->h_ct->total = 0; h->h_ct->max_per_read = 0; h->h_ct->next_read = 0; h->h_ct->n_ovf = 0; h->h_ct->wide_items = h->h_ct->wide_cands = h->h_ct->wide_queue = 0; h->h_ct->wide_top = 0; h->h_ct->wave_steps = 0; h->h_ct->positions = h->h_ct->spills = h->h_ct->list_sum = 0;
            GAB_HIP(hipMemcpyAsync(d_ct, h->h_ct, sizeof(FmiCounters), hipMemcpyHostToDevice, s));
            GAB_HIP(hipEventRecord(h->ev[0], s));
            rc = seed(first, nb, nb, nullptr, h->slots.as<OutRec>(), cap, &seed_blocks_dbg);
            if (rc) return rc;
            GAB_HIP(hipEventRecord(h->ev[1], s));
            GAB_HIP(hipMemcpyAsync(h->h_ct, d_ct, sizeof(FmiCounters), hipMemcpyDeviceToHost, s));
            GAB_HIP(hipStreamSynchronize(s));
            if (!handover || h->h_ct->wide_cands <= wide_cap(nb)) break;
            // a candidate without a place is a re-seeding that did not happen: the batch again, every phase with its own lane
            float lost = 0;
            GAB_HIP(hipEventElapsedTime(&lost, h->ev[0], h->ev[1]));
            kms += lost;
            handover = false;
        }
        const int over_cap = h->h_ct->max_per_read > cap ? h->h_ct->max_per_read : 0;   // the counters of THIS round are the batch's
        float ms = 0;
        GAB_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
        kms += ms;
        ext_total += h->h_ct->ext_calls + h->h_ct->tab_reads; rec_total += h->h_ct->rec_reads;
        if (h->tun.fmi_debug)
            fprintf(stderr, "[gab_fmi] batch of %d reads: %d waves (%d per CU), %.3f ms, %llu index extensions + %llu table look-ups, %llu wave steps -> %.1f extensions per step; %llu positions, %llu spilled, mean list %.2f\n",
                    nb, seed_blocks_dbg, waves_per_cu, ms, h->h_ct->ext_calls, h->h_ct->tab_reads, h->h_ct->wave_steps,
                    (double)(h->h_ct->ext_calls + h->h_ct->tab_reads) / (double)(h->h_ct->wave_steps ? h->h_ct->wave_steps : 1), h->h_ct->positions,
                    h->h_ct->spills, (double)h->h_ct->list_sum / (double)(h->h_ct->positions ? h->h_ct->positions : 1));
        if (h->tun.fmi_debug && h->h_ct->wide_items)
            fprintf(stderr, "[gab_fmi]   %d wide backward phases handed over (%u list entries), %d re-seeding candidates from them\n",
                    h->h_ct->wide_items, h->h_ct->wide_top, h->h_ct->wide_cands);
        const int64_t add = (int64_t)h->h_ct->total;
        if ((size_t)(total + add) > out_cap) {
            // grow, keeping what earlier batches wrote
            size_t ncap = std::max<size_t>(out_cap * 2, (size_t)(total + add));
            gab_devbuf bigger;
            rc = bigger.reserve(ncap * sizeof(gab_smem));
            if (rc) return rc;
            GAB_HIP(hipMemcpyAsync(bigger.p, h->out.p, (size_t)total * sizeof(gab_smem), hipMemcpyDeviceToDevice, s));
            GAB_HIP(hipStreamSynchronize(s));
            h->out.release();
            h->out = bigger;
            out_cap = ncap;
        }
        hipLaunchKernelGGL(fmi_block_sums, dim3(blocks), dim3(256), 0, s, d_counts, nb, d_bs);
        hipLaunchKernelGGL(fmi_scan_blocks, dim3(1), dim3(1024), 0, s, d_bs, blocks, total);
        hipLaunchKernelGGL(fmi_compact, dim3(blocks), dim3(256), 0, s, d_counts, nb, d_bs, h->slots.as<OutRec>(), cap, first,
                           h->roff.as<int64_t>(), h->out.as<gab_smem>());
        GAB_HIP(hipGetLastError());
        if (over_cap) {
            // Some reads found more SMEMs than a first-round slot holds (low-complexity reads; a handful among millions of
            // real ones): they alone run again with slots of the size the worst of them needs, in as many rounds as the
            // scratch budget asks for, and their records go to the places fmi_compact left open.  Everything the kernel
            // counts (extensions, records) was complete after the first round; the second round's counts are dropped.
            rc = h->ovf.reserve(4 * (size_t)nb + 64);
            if (rc) return rc;
            int32_t *d_ids = h->ovf.as<int32_t>();
            hipLaunchKernelGGL(fmi_list_overflowed, dim3(blocks), dim3(256), 0, s, d_counts, nb, cap, d_ids, &d_ct->n_ovf);
            GAB_HIP(hipGetLastError());
            int32_t n_ovf = 0;
            GAB_HIP(hipMemcpyAsync(&n_ovf, &d_ct->n_ovf, 4, hipMemcpyDeviceToHost, s));
            GAB_HIP(hipStreamSynchronize(s));
            const int64_t per_round = std::max<int64_t>(1, (int64_t)(budget / (sizeof(OutRec) * (size_t)over_cap)));
            const int64_t round = std::min<int64_t>(n_ovf, per_round);
            rc = h->slots2.reserve(sizeof(OutRec) * (size_t)over_cap * (size_t)round);
            if (rc) return rc;
            if (h->tun.fmi_debug)
                fprintf(stderr, "[gab_fmi] %d read(s) overflowed their %d-record slot (worst: %d records): second round in %lld part(s)\n",
                        n_ovf, cap, over_cap, (long long)gab_ceil_div((int64_t)n_ovf, round));
            GAB_HIP(hipEventRecord(h->ev[0], s));
            for (int64_t i0 = 0; i0 < n_ovf; i0 += round) {
                const int32_t cnt = (int32_t)std::min<int64_t>(round, n_ovf - i0);
                rc = seed(first, nb, cnt, d_ids + i0, h->slots2.as<OutRec>(), over_cap, nullptr);
                if (rc) return rc;
                hipLaunchKernelGGL(fmi_compact_overflowed, dim3((unsigned)gab_ceil_div((int64_t)cnt, 256)), dim3(256), 0, s, d_counts, d_ids + i0,
                                   cnt, h->slots2.as<OutRec>(), over_cap, first, h->roff.as<int64_t>(), h->out.as<gab_smem>());
                GAB_HIP(hipGetLastError());
            }
            GAB_HIP(hipEventRecord(h->ev[1], s));
            GAB_HIP(hipEventSynchronize(h->ev[1]));
            GAB_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
            kms += ms;
        }
        total += add;
        first += nb;
    }
    GAB_HIP(hipMemcpyAsync(h->roff.as<int64_t>() + nreads, &total, 8, hipMemcpyHostToDevice, s));
    GAB_HIP(hipStreamSynchronize(s));
    *nout = total;
    if (d_out) *d_out = h->out.as<gab_smem>();
    h->ext_calls = (int64_t)ext_total; h->rec_reads = (int64_t)rec_total; h->nsmem = total; h->kernel_ms = kms;
    h->have_stats = true;
    return GAB_OK;
}

extern "C" int gab_fmi_seed(gab_fmi *h, const uint8_t *enc, int32_t stride, const int32_t *len, int64_t nreads,
                            int32_t min_seed_len, gab_smem **out, int64_t *nout) {
    GAB_CHECK(h, "gab_fmi_seed: NULL handle");
    GAB_CHECK(out && nout, "gab_fmi_seed: NULL output argument");
    *out = nullptr; *nout = 0;
    GAB_CHECK(nreads >= 0 && nreads < (1ll << 31), "gab_fmi_seed: nreads out of range");
    if (nreads == 0) return GAB_OK;
    GAB_CHECK(enc && len && stride > 0, "gab_fmi_seed: NULL buffer");
    gab_device_guard g(h->device);
    const size_t eb = (size_t)nreads * (size_t)stride;
    const size_t o_len = (eb + 255) & ~(size_t)255;
    int rc = h->io.reserve(o_len + 4 * (size_t)nreads);
    if (rc) return rc;
    hipStream_t s = nullptr;
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    char *b = h->io.as<char>();
    {   // the copies of one chunk at a time per GPU (gab_core.hip: the workers of a GPU must not copy in lockstep)
        std::lock_guard<std::mutex> gate(gab_h2d_mutex(h->device));
        GAB_HIP(hipMemcpyAsync(b, enc, eb, hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_len, len, 4 * (size_t)nreads, hipMemcpyHostToDevice, s));
        GAB_HIP(hipStreamSynchronize(s));
    }
    const gab_smem *d_out = nullptr;
    int64_t n = 0;
    rc = gab_fmi_seed_device(h, (const uint8_t *)b, stride, (const int32_t *)(b + o_len), nreads, min_seed_len, &d_out,
                             nullptr, &n, s);
    if (rc) return rc;
    gab_smem *host = (gab_smem *)malloc(sizeof(gab_smem) * (size_t)(n > 0 ? n : 1));
    if (!host) { gab_set_error("gab_fmi_seed: out of host memory"); return GAB_ENOMEM; }
    if (n) {
        if (hipMemcpy(host, d_out, sizeof(gab_smem) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) {
            free(host); gab_set_error("gab_fmi_seed: D2H copy failed"); return GAB_EDEVICE;
        }
    }
    *out = host; *nout = n;
    return GAB_OK;
}

extern "C" void gab_fmi_free(gab_smem *p) { free(p); }

// The reference's worker threads write their SMEMs into per-thread arrays allocated before the ROI (fmi/fmi.cpp:242, 253-257: sized by a per-thread quota, allocated at the start of the parallel region)
// and grow them when a batch does not fit (:277-286); this is the same contract with the caller's buffer, which may be
// page-locked (gab_host_alloc / gab_host_register) so that the result comes back as one DMA at the full link rate.
extern "C" int gab_fmi_seed_into(gab_fmi *h, const uint8_t *enc, int32_t stride, const int32_t *len, int64_t nreads,
                                 int32_t min_seed_len, gab_smem *out, int64_t capacity, int64_t *nout) {
    GAB_CHECK(h, "gab_fmi_seed_into: NULL handle");
    GAB_CHECK(nout, "gab_fmi_seed_into: NULL output argument");
    *nout = 0;
    GAB_CHECK(capacity >= 0 && (out || capacity == 0), "gab_fmi_seed_into: NULL buffer with a capacity");
    GAB_CHECK(nreads >= 0 && nreads < (1ll << 31), "gab_fmi_seed_into: nreads out of range");
    if (nreads == 0) return GAB_OK;
    GAB_CHECK(enc && len && stride > 0, "gab_fmi_seed_into: NULL buffer");
    gab_device_guard g(h->device);
    const size_t eb = (size_t)nreads * (size_t)stride;
    const size_t o_len = (eb + 255) & ~(size_t)255;
    int rc = h->io.reserve(o_len + 4 * (size_t)nreads);
    if (rc) return rc;
    hipStream_t s = nullptr;
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    char *b = h->io.as<char>();
    {
        std::lock_guard<std::mutex> gate(gab_h2d_mutex(h->device));
        GAB_HIP(hipMemcpyAsync(b, enc, eb, hipMemcpyHostToDevice, s));
        GAB_HIP(hipMemcpyAsync(b + o_len, len, 4 * (size_t)nreads, hipMemcpyHostToDevice, s));
        GAB_HIP(hipStreamSynchronize(s));
    }
    const gab_smem *d_out = nullptr;
    int64_t n = 0;
    rc = gab_fmi_seed_device(h, (const uint8_t *)b, stride, (const int32_t *)(b + o_len), nreads, min_seed_len, &d_out,
                             nullptr, &n, s);
    if (rc) return rc;
    *nout = n;
    if (n > capacity) {
        gab_set_error("gab_fmi_seed_into: %lld SMEMs do not fit the caller's %lld records", (long long)n, (long long)capacity);
        return GAB_ERANGE;
    }
    if (n) {
        GAB_HIP(hipMemcpyAsync(out, d_out, sizeof(gab_smem) * (size_t)n, hipMemcpyDeviceToHost, s));
        GAB_HIP(hipStreamSynchronize(s));
    }
    return GAB_OK;
}

extern "C" int gab_fmi_reserve(gab_fmi *h, int64_t max_reads, int32_t stride) {
    GAB_CHECK(h, "gab_fmi_reserve: NULL handle");
    GAB_CHECK(max_reads >= 0 && max_reads < (1ll << 31) && stride > 0, "gab_fmi_reserve: size out of range");
    if (max_reads == 0) return GAB_OK;
    gab_device_guard g(h->device);
    const size_t eb = (size_t)max_reads * (size_t)stride, o_len = (eb + 255) & ~(size_t)255;
    int rc = h->io.reserve(std::max<size_t>(o_len + 4 * (size_t)max_reads, (size_t)4 << 20));
    if (rc) return rc;
    hipStream_t s = nullptr;
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    // max_reads reads of length 0: every buffer of the path gets the size a real batch of that many reads starts with, every
    // kernel runs once, nothing is found
    GAB_HIP(hipMemsetAsync(h->io.p, 0, o_len + 4 * (size_t)max_reads, s));
    const gab_smem *d_out = nullptr;
    int64_t n = 0;
    const bool had = h->have_stats;
    rc = gab_fmi_seed_device(h, h->io.as<uint8_t>(), stride, (const int32_t *)(h->io.as<char>() + o_len), max_reads, 19, &d_out, nullptr, &n, s);
    h->have_stats = had;
    if (rc) return rc;
    // the output of a real batch: ~8 records per read of 151 bases (the device array grows by doubling from 16 per read)
    return gab_warm_copy_engines(s, h->io.p, h->io.cap);
}

extern "C" int gab_fmi_last_stats(gab_fmi *h, int64_t *ext_calls, int64_t *nsmem, float *kernel_ms) {
    GAB_CHECK(h, "gab_fmi_last_stats: NULL handle");
    GAB_CHECK(h->have_stats, "gab_fmi_last_stats: no completed run on this handle");
    if (ext_calls) *ext_calls = h->ext_calls;
    if (nsmem) *nsmem = h->nsmem;
    if (kernel_ms) *kernel_ms = h->kernel_ms;
    return GAB_OK;
}

extern "C" int gab_fmi_last_records(gab_fmi *h, int64_t *cp_occ_records) {
    GAB_CHECK(h, "gab_fmi_last_records: NULL handle");
    GAB_CHECK(h->have_stats, "gab_fmi_last_records: no completed run on this handle");
    if (cp_occ_records) *cp_occ_records = h->rec_reads;
    return GAB_OK;
}

// ---- suffix-array look-up: host side -----------------------------------------------------------------------------
extern "C" int gab_fmi_set_sa(gab_fmi *h, const int8_t *sa_ms_byte, const uint32_t *sa_ls_word) {
    GAB_CHECK(h, "gab_fmi_set_sa: NULL handle");
    GAB_CHECK(sa_ms_byte && sa_ls_word, "gab_fmi_set_sa: NULL argument");
    gab_device_guard g(h->device);
    const size_t n_sa = (size_t)((h->ix.ref_seq_len >> 3) + 1), o_ls = (n_sa + 255) & ~(size_t)255;
    int rc = h->sa.reserve(o_ls + 4 * n_sa);
    if (rc) return rc;
    GAB_HIP(hipMemcpy(h->sa.p, sa_ms_byte, n_sa, hipMemcpyHostToDevice));
    GAB_HIP(hipMemcpy(h->sa.as<char>() + o_ls, sa_ls_word, 4 * n_sa, hipMemcpyHostToDevice));
    h->sa_ix.ms = h->sa.as<int8_t>();
    h->sa_ix.ls = (const uint32_t *)(h->sa.as<char>() + o_ls);
    return GAB_OK;
}

extern "C" int gab_fmi_sa_lookup_device(gab_fmi *h, const gab_smem *d_smems, int64_t n, int32_t max_occ,
                                        const int64_t **d_coords, const int64_t **d_coord_off, int64_t *total, void *stream_) {
    GAB_CHECK(h, "gab_fmi_sa_lookup_device: NULL handle");
    GAB_CHECK(h->sa_ix.ms, "gab_fmi_sa_lookup_device: the handle has no suffix array (gab_fmi_load or gab_fmi_set_sa)");
    GAB_CHECK(n >= 0 && n < (1ll << 31), "gab_fmi_sa_lookup_device: n out of range");
    GAB_CHECK(max_occ >= 1, "gab_fmi_sa_lookup_device: max_occ must be >= 1");
    GAB_CHECK(total, "gab_fmi_sa_lookup_device: NULL total");
    h->have_sa_stats = false;
    *total = 0;
    gab_device_guard g(h->device);
    hipStream_t s = (hipStream_t)stream_;
    int rc = h->sa_off.reserve(8 * (size_t)(n + 1));
    if (rc) return rc;
    if (d_coord_off) *d_coord_off = h->sa_off.as<int64_t>();
    if (d_coords) *d_coords = nullptr;
    if (n == 0) { GAB_HIP(hipMemsetAsync(h->sa_off.p, 0, 8, s)); return GAB_OK; }
    GAB_CHECK(d_smems, "gab_fmi_sa_lookup_device: NULL buffer");
    const int64_t blocks = gab_ceil_div(n, 256);
    const size_t o_counts = 256, o_bs = (o_counts + 4 * (size_t)n + 63) & ~(size_t)63;
    rc = h->sa_ws.reserve(o_bs + 8 * (size_t)(blocks + 1) + 64);
    if (rc) return rc;
    char *wb = h->sa_ws.as<char>();
    SaCounters *d_ct = (SaCounters *)wb;
    int32_t *d_counts = (int32_t *)(wb + o_counts);
    int64_t *d_bs = (int64_t *)(wb + o_bs);
    int64_t *d_off = h->sa_off.as<int64_t>();
    GAB_HIP(hipMemsetAsync(d_ct, 0, sizeof(SaCounters), s));
    GAB_HIP(hipEventRecord(h->ev[0], s));
    hipLaunchKernelGGL(fmi_sa_count, dim3((unsigned)blocks), dim3(256), 0, s, d_smems, n, max_occ, d_counts);
    // block sums -> exclusive scan; the scan kernel leaves the grand total in block_sums[nblocks]
    hipLaunchKernelGGL(fmi_block_sums, dim3((unsigned)blocks), dim3(256), 0, s, d_counts, (int32_t)n, d_bs);
    hipLaunchKernelGGL(fmi_scan_blocks, dim3(1), dim3(1024), 0, s, d_bs, (int32_t)blocks, (int64_t)0);
    hipLaunchKernelGGL(fmi_sa_offsets, dim3((unsigned)blocks), dim3(256), 0, s, d_counts, n, d_bs, d_off);
    GAB_HIP(hipGetLastError());
    // total = off[n-1] + counts[n-1]
    int64_t last_off = 0; int32_t last_cnt = 0;
    GAB_HIP(hipMemcpyAsync(&last_off, d_off + (n - 1), 8, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipMemcpyAsync(&last_cnt, d_counts + (n - 1), 4, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    const int64_t tot = last_off + last_cnt;
    GAB_HIP(hipMemcpyAsync(d_off + n, &tot, 8, hipMemcpyHostToDevice, s));
    rc = h->sa_coords.reserve(8 * (size_t)std::max<int64_t>(tot, 1));
    if (rc) return rc;
    int n_cu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, h->device) == hipSuccess) n_cu = prop.multiProcessorCount; }
    const int64_t chunks = gab_ceil_div(n, kSaChunk);
    const unsigned waves = (unsigned)std::min<int64_t>(chunks, (int64_t)n_cu * 32);
    hipLaunchKernelGGL(fmi_sa_kernel, dim3(waves), dim3(64), 0, s, h->ix, h->sa_ix, d_smems, n, max_occ, d_off, h->sa_coords.as<int64_t>(), d_ct);
    GAB_HIP(hipGetLastError());
    GAB_HIP(hipEventRecord(h->ev[1], s));
    SaCounters hc;
    GAB_HIP(hipMemcpyAsync(&hc, d_ct, sizeof hc, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    GAB_HIP(hipEventElapsedTime(&h->sa_ms, h->ev[0], h->ev[1]));
    h->sa_lf_steps = (int64_t)hc.lf_steps;
    h->have_sa_stats = true;
    *total = tot;
    if (d_coords) *d_coords = h->sa_coords.as<int64_t>();
    return GAB_OK;
}

extern "C" int gab_fmi_sa_lookup(gab_fmi *h, const gab_smem *smems, int64_t n, int32_t max_occ, int64_t **coords,
                                 int64_t **coord_off, int64_t *total) {
    GAB_CHECK(h, "gab_fmi_sa_lookup: NULL handle");
    GAB_CHECK(coords && coord_off && total, "gab_fmi_sa_lookup: NULL output argument");
    *coords = nullptr; *coord_off = nullptr; *total = 0;
    GAB_CHECK(n >= 0, "gab_fmi_sa_lookup: n < 0");
    GAB_CHECK(n == 0 || smems, "gab_fmi_sa_lookup: NULL buffer");
    gab_device_guard g(h->device);
    int rc = h->sa_io.reserve(sizeof(gab_smem) * (size_t)std::max<int64_t>(n, 1));
    if (rc) return rc;
    hipStream_t s = nullptr;
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    if (n) GAB_HIP(hipMemcpyAsync(h->sa_io.p, smems, sizeof(gab_smem) * (size_t)n, hipMemcpyHostToDevice, s));
    const int64_t *d_c = nullptr, *d_o = nullptr;
    int64_t tot = 0;
    rc = gab_fmi_sa_lookup_device(h, h->sa_io.as<gab_smem>(), n, max_occ, &d_c, &d_o, &tot, s);
    if (rc) return rc;
    int64_t *hc = (int64_t *)malloc(8 * (size_t)std::max<int64_t>(tot, 1)), *ho = (int64_t *)malloc(8 * (size_t)(n + 1));
    if (!hc || !ho) { free(hc); free(ho); gab_set_error("gab_fmi_sa_lookup: out of host memory"); return GAB_ENOMEM; }
    if (hipMemcpy(ho, d_o, 8 * (size_t)(n + 1), hipMemcpyDeviceToHost) != hipSuccess ||
        (tot && hipMemcpy(hc, d_c, 8 * (size_t)tot, hipMemcpyDeviceToHost) != hipSuccess)) {
        free(hc); free(ho); gab_set_error("gab_fmi_sa_lookup: D2H copy failed"); return GAB_EDEVICE;
    }
    *coords = hc; *coord_off = ho; *total = tot;
    return GAB_OK;
}

extern "C" void gab_fmi_free_coords(int64_t *p) { free(p); }

extern "C" int gab_fmi_last_sa_stats(gab_fmi *h, int64_t *lf_steps, float *kernel_ms) {
    GAB_CHECK(h, "gab_fmi_last_sa_stats: NULL handle");
    GAB_CHECK(h->have_sa_stats, "gab_fmi_last_sa_stats: no completed look-up on this handle");
    if (lf_steps) *lf_steps = h->sa_lf_steps;
    if (kernel_ms) *kernel_ms = h->sa_ms;
    return GAB_OK;
}
