// bitpal -- global alignment scores of the bpm benchmark's BitPAl algorithms on gfx950.
//
// Semantics: benchmark_bitpal_m0_x1_g1 / benchmark_bitpal_m1_x4_g2,
//   /root/reference/benchmarks/bpm/benchmark/benchmark_bitpal.c:30-54, selected by the driver's
//   `-a bitpal-edit` / `-a bitpal-scored` (bpm/tools/align_benchmark.c:259-264, 330-338).  They call the generated
//   bit-vector programs bpm/bitpal/bitpal.m0.x1.g1.c / bitpal.m1.x4.g2.c, whose result is the Needleman-Wunsch score
//   of the two strings: global, linear gap cost, characters compared as raw bytes, with
//   (match, mismatch, gap) = (0, -1, -1) resp. (+1, -4, -2).  No CIGAR is produced.
//
// Mapping: BitPAl packs 63 DP columns into a machine word because a CPU core has one pair to work on; here there are
// millions of independent pairs, so the lane, not the bit, is the unit of parallelism: ONE PAIR PER LANE, plain integer
// DP.  The longer string is walked in chunks of 32 columns held in registers (32 scores + 32 bytes), the shorter string
// is the row loop; the only memory state is the chunk's boundary column -- one score per row -- kept per lane in LDS in
// [row][lane] order as ONE BYTE per row, the difference to the row above (0 .. match - 2 gap in the normalised scores
// below), or in a global scratch as int32 when a pair is too long for that.  The byte column is what sets the occupancy:
// 9.7 KB per wave for 151-bp pairs = 16 waves per CU (int16 scores: 8 waves and 25 % slower).
// The DP runs on S'[i][j] = S[i][j] - (i + j) * gap: vertical and horizontal moves then cost nothing, a diagonal move adds
// match - 2 gap or mismatch - 2 gap, and both borders are zero.  The increments of four columns are built as the bytes of
// one dword (SWAR zero-byte test on columns ^ row character), so one cell is v_add_u32_sdwa (byte operand) + v_max3_i32;
// per row there is one LDS read, one LDS write and a sixteenth of a 16-byte load of the row string.
//
// Roofline: plen + tlen + 4 algorithmic bytes per pair (the row string is really streamed once per 32-column chunk:
// 3.1 x, profiles/r01_hbm_traffic.md) against 3.5 VALU per DP cell: integer-VALU bound, ~93 % of the issue rate.
#include "gab_internal.h"
#include "gab_bitvec.h"
#include <algorithm>
#include <new>
#include <string.h>

namespace {

constexpr int kChunk = 32;                 // DP columns per register chunk
constexpr int kLdsRows = 2048;             // longest row string (the shorter of the pair) on the LDS path: 128 KB of bytes per wave
constexpr int kBigBlocks = 256;            // workgroups of the global-scratch path

struct BpIO {
    const char *pat; const int64_t *pat_off; const int32_t *pat_len;
    const char *txt; const int64_t *txt_off; const int32_t *txt_len;
    int64_t pat_bytes, txt_bytes, n;
    int32_t *score;
};

struct BpCounters {
    int32_t bad, first_bad;
    int32_t max_rows_lds, max_rows_big;     // longest row string per path
    uint32_t n_lds, n_big;
    uint32_t cursors[2];                    // bitpal_scatter
    unsigned long long cells;
};
GAB_STATIC_ATOMIC64(BpCounters, cells);

struct BpScore { int32_t match, mismatch, gap; };

// four bytes of a sequence at position `pos`; bytes at or behind `len` read as `fill` (never loads past the last dword
// that holds a sequence byte: the slabs are readable to a multiple of 4 bytes only)
__device__ __forceinline__ uint32_t seq_ld4(const char *s, int pos, int len, uint32_t fill) {
    uint32_t w;
    if (pos + 4 <= len) { __builtin_memcpy(&w, s + pos, 4); return w; }
    w = fill * 0x01010101u;
    for (int b = 0; b < 4; b++)
        if (pos + b < len) w = (w & ~(0xffu << (8 * b))) | (uint32_t)(uint8_t)s[pos + b] << (8 * b);
    return w;
}

// sixteen bytes: one 16-byte load when they all lie inside the sequence.  Every lane streams its own strings, and 16 waves
// of 64 lanes keep far more lines alive than the 16 KB L1 holds, so with 4-byte loads each line was fetched again and
// again (HBM fetch 16 GB per 10 M pairs = 5 x the algorithmic bytes)
__device__ __forceinline__ uint4 seq_ld16(const char *s, int pos, int len, uint32_t fill) {
    uint4 w;
    if (pos + 16 <= len) { __builtin_memcpy(&w, s + pos, 16); return w; }
    w.x = pos < len ? seq_ld4(s, pos, len, fill) : fill * 0x01010101u;
    w.y = pos + 4 < len ? seq_ld4(s, pos + 4, len, fill) : fill * 0x01010101u;
    w.z = pos + 8 < len ? seq_ld4(s, pos + 8, len, fill) : fill * 0x01010101u;
    w.w = pos + 12 < len ? seq_ld4(s, pos + 12, len, fill) : fill * 0x01010101u;
    return w;
}

// ---- pass 0: validate, count per path -----------------------------------------------------------------------------------
// Counts are accumulated per lane and reduced once per wave: even one atomic per wave-iteration on a single address
// serialises for milliseconds at 10 M pairs.  When every pair takes the LDS path (the usual case) no id list is needed at
// all: the DP kernel then maps slot -> pair by identity.
__device__ __forceinline__ bool bitpal_pair_ok(const BpIO &io, int64_t i) {
    const int pl = io.pat_len[i], tl = io.txt_len[i];
    const int64_t po = io.pat_off[i], to = io.txt_off[i];
    return pl >= 0 && tl >= 0 && pl <= GAB_BITPAL_MAX_LEN && tl <= GAB_BITPAL_MAX_LEN && po >= 0 && to >= 0 &&
           ((po + pl + 3) & ~3ll) <= io.pat_bytes && ((to + tl + 3) & ~3ll) <= io.txt_bytes;
}

__global__ __launch_bounds__(256) void bitpal_classify(BpIO io, BpCounters *ct) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int ml = 0, mb = 0;
    uint32_t n_lds = 0, n_big = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < io.n; i += stride) {
        if (!bitpal_pair_ok(io, i)) {
            atomicAdd(&ct->bad, 1);
            atomicMin((unsigned int *)&ct->first_bad, (unsigned int)(i + 1 > 0x7fffffff ? 0x7fffffff : i + 1));
            continue;
        }
        const int rows = min(io.pat_len[i], io.txt_len[i]);
        if (rows <= kLdsRows) { n_lds++; ml = max(ml, rows); } else { n_big++; mb = max(mb, rows); }
    }
    for (int o = 32; o > 0; o >>= 1) {
        ml = max(ml, __shfl_xor(ml, o)); mb = max(mb, __shfl_xor(mb, o));
        n_lds += __shfl_xor(n_lds, o); n_big += __shfl_xor(n_big, o);
    }
    // one set of atomics per workgroup: they all hit the same four addresses and serialise in L2 (see wfa_classify)
    __shared__ int s_ml[4], s_mb[4];
    __shared__ uint32_t s_nl[4], s_nb[4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_ml[wv] = ml; s_mb[wv] = mb; s_nl[wv] = n_lds; s_nb[wv] = n_big; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) { ml = max(ml, s_ml[k]); mb = max(mb, s_mb[k]); n_lds += s_nl[k]; n_big += s_nb[k]; }
        atomicMax(&ct->max_rows_lds, ml); atomicMax(&ct->max_rows_big, mb);
        if (n_lds) atomicAdd(&ct->n_lds, n_lds);
        if (n_big) atomicAdd(&ct->n_big, n_big);
    }
}

// only when some pair is too long for the LDS column: the two id lists (the cursors start at zero)
__global__ __launch_bounds__(256) void bitpal_scatter(BpIO io, uint32_t *cursors, uint32_t *list_lds, uint32_t *list_big) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < io.n; i += stride) {
        const bool to_lds = min(io.pat_len[i], io.txt_len[i]) <= kLdsRows;
        const uint32_t s_lds = gab_wave_slot(&cursors[0], to_lds), s_big = gab_wave_slot(&cursors[1], !to_lds);
        if (to_lds) list_lds[s_lds] = (uint32_t)i; else list_big[s_big] = (uint32_t)i;
    }
}

// ---- the DP: one pair per lane ------------------------------------------------------------------------------------------
// LDSCOL: boundary column = uint8 row-to-row differences [row][lane] in dynamic LDS, one workgroup (= one wave) per 64
// pairs; otherwise int32 scores [row][lane] in this workgroup's slice of `gcol`, workgroups stride over the groups of 64.
// SCORED selects the scoring at compile time: the per-column diagonal increments of four columns are built as one
// dword of bytes (SWAR), which needs the two values as constants.
template <bool LDSCOL, bool SCORED>
__global__ __launch_bounds__(64) void bitpal_dp(BpIO io, const uint32_t *__restrict__ list, uint32_t count, BpScore sc,
                                                int32_t *gcol, int64_t gcol_per_block, BpCounters *ct) {
    extern __shared__ __attribute__((aligned(16))) uint8_t col_lds[];
    const int lane = threadIdx.x;
    int32_t *col_glb = LDSCOL ? nullptr : gcol + (int64_t)blockIdx.x * gcol_per_block;
    // LDS: the value is the difference S'[row][.] - S'[row - 1][.]; global: the score itself
    auto col_get = [&](int row) -> int { return LDSCOL ? (int)col_lds[row * 64 + lane] : col_glb[(int64_t)row * 64 + lane]; };
    auto col_put = [&](int row, int v) {
        if (LDSCOL) col_lds[row * 64 + lane] = (uint8_t)v;
        else col_glb[(int64_t)row * 64 + lane] = v;
    };
    // diagonal increments of S': match - 2 gap / mismatch - 2 gap = 2 / 1 (edit) and 5 / 0 (scored)
    const int G = sc.gap;
    unsigned long long cells = 0;
    for (uint32_t g0 = blockIdx.x * 64u; g0 < count; g0 += gridDim.x * 64u) {
        const uint32_t b = g0 + (uint32_t)lane;
        const bool have = b < count;
        int nc = 0, nr = 0;                                  // column string (the longer one, in registers) / row string
        const char *cs = nullptr, *rs = nullptr;
        uint32_t id = 0;
        if (have) {
            id = list ? list[b] : b;                         // no list: every pair is on this path
            const int pl = io.pat_len[id], tl = io.txt_len[id];
            const char *p = io.pat + io.pat_off[id], *t = io.txt + io.txt_off[id];
            if (pl >= tl) { nc = pl; cs = p; nr = tl; rs = t; } else { nc = tl; cs = t; nr = pl; rs = p; }
        }
        // column 0 of S': zero
        for (int i = 0; i <= nr; i++) col_put(i, 0);
        int ans = 0;                                         // both strings empty
        const int nchunks = (nc + kChunk - 1) / kChunk;
        for (int ch = 0; ch < nchunks; ch++) {
            const int base = ch * kChunk;
            uint32_t cw[kChunk / 4];                         // the chunk's bytes of the column string; 0xff behind its end
#pragma unroll
            for (int w = 0; w < kChunk / 16; w++) {
                const uint4 q = base + 16 * w < nc ? seq_ld16(cs, base + 16 * w, nc, 0xffu) : make_uint4(~0u, ~0u, ~0u, ~0u);
                cw[4 * w] = q.x; cw[4 * w + 1] = q.y; cw[4 * w + 2] = q.z; cw[4 * w + 3] = q.w;
            }
            // rows alternate between two register sets, so that "this row" never has to be copied over "previous row"
            int Ha[kChunk], Hb[kChunk];
#pragma unroll
            for (int c = 0; c < kChunk; c++) Ha[c] = 0;                          // row 0 of S'
            int diag = 0;                                                       // S'[i - 1][base], starting with row 0
            uint32_t rw = 0;
            uint4 rq = make_uint4(0u, 0u, 0u, 0u);           // sixteen characters of the row string
            // the dword holding row i's character (called when (i - 1) % 4 == 0; all indices are wave-uniform)
            auto next_rw = [&](int i) {
                if (((i - 1) & 15) == 0) rq = seq_ld16(rs, i - 1, nr, 0u);
                const int q = ((i - 1) >> 2) & 3;
                rw = q == 0 ? rq.x : q == 1 ? rq.y : q == 2 ? rq.z : rq.w;
            };
            auto row = [&](int i, const int (&Hin)[kChunk], int (&Hout)[kChunk]) {
                // the row's character in all four bytes (v_perm_b32 with the wave-uniform selector 0x01010101 * ((i - 1) & 3))
                const uint32_t rc4 = __builtin_amdgcn_perm(rw, rw, (uint32_t)((i - 1) & 3) * 0x01010101u);
                int left = LDSCOL ? diag + col_get(i) : col_get(i);             // S'[i][base]
                const int next_diag = left;
#pragma unroll
                for (int c4 = 0; c4 < kChunk / 4; c4++) {
                    // four columns at a time: byte b of `inc` = the diagonal increment of column 4 c4 + b
                    const uint32_t x = cw[c4] ^ rc4;                             // zero byte = equal characters
                    const uint32_t nz = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) >> 7;    // bit 0 of each byte: characters differ
                    const uint32_t eq = ~nz & 0x01010101u;
                    uint32_t inc = SCORED ? (eq << 2) + eq : eq + 0x01010101u;           // 5 or 0 / 2 or 1
                    // keep the dword opaque: the four adds below then take their byte through SDWA (v_add_u32_sdwa ...
                    // src1_sel:BYTE_n) instead of the compiler re-deriving each byte from `eq` with a bfe and an add
                    asm("" : "+v"(inc));
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int c = 4 * c4 + b;
                        const int up = Hin[c];
                        const int h = max(diag + (int)((inc >> (8 * b)) & 0xffu), max(up, left));
                        diag = up; left = h; Hout[c] = h;
                    }
                }
                col_put(i, LDSCOL ? Hout[kChunk - 1] - Hin[kChunk - 1] : Hout[kChunk - 1]);
                diag = next_diag;
            };
            int i = 1;
            for (; i < nr; i += 2) {
                if (((i - 1) & 3) == 0) next_rw(i);
                row(i, Ha, Hb);
                row(i + 1, Hb, Ha);
            }
            if (i == nr) {
                if (((i - 1) & 3) == 0) next_rw(i);
                row(i, Ha, Hb);
#pragma unroll
                for (int c = 0; c < kChunk; c++) Ha[c] = Hb[c];
            }
            if (nc > base && nc <= base + kChunk) {
#pragma unroll
                for (int c = 0; c < kChunk; c++) ans = (c == nc - 1 - base) ? Ha[c] : ans;
            }
        }
        if (have) { io.score[id] = ans + (nc + nr) * G; cells += (unsigned long long)nc * (unsigned long long)nr; }
    }
    for (int o = 32; o > 0; o >>= 1) cells += __shfl_xor(cells, o);
    if (lane == 0 && cells) atomicAdd(&ct->cells, cells);
}

// ---- -a bitpal-edit, bit-vector path: the score is minus the edit distance --------------------------------------------
// benchmark_bitpal_m0_x1_g1 scores (match, mismatch, gap) = (0, -1, -1): the Needleman-Wunsch score is minus the Levenshtein
// distance of the two strings as raw bytes.  For that one algorithm the column of the DP is Myers' bit-vector (gab_bitvec.h:
// the column as ONE integer of D 32-bit words, ~60 VALU instructions per column of a 151-row pair) instead of 151 cells of
// integer DP at 3.5 instructions each.  One pair per lane; the SHORTER string makes the rows (the distance is symmetric).
// Raw-byte equality needs a match mask per byte value that occurs in the row string: slots for 'A' 'C' 'G' 'T' 'N' (upper case)
// and a sixth, always empty, that every other byte of the COLUMN string selects (it equals no row byte); a row string with
// any other byte, or longer than 32 D <= 256 rows, sends the pair to the integer DP through the id lists (bitpal_scatter's
// job, done here for the rejects).  LDS: the lane's masks as [(slot * D + word)][lane] dwords, and a byte -> slot offset table.
constexpr int kBvBlock = 256;
constexpr int kBvSlots = 6;
constexpr int kBvMaxRows = 256;

__device__ __forceinline__ int bitpal_bv_slot(uint32_t ch) {
    return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : ch == 'N' ? 4 : 5;
}

template <int D>
__global__ __launch_bounds__(kBvBlock) void bitpal_edit_bv(BpIO io, uint32_t *cursors, uint32_t *__restrict__ list_lds,
                                                           uint32_t *__restrict__ list_big, BpCounters *ct) {
    __shared__ uint32_t eq_s[kBvSlots * D * kBvBlock];
    __shared__ uint32_t lut[256];                            // byte -> first word of its slot (in units of kBvBlock dwords)
    lut[threadIdx.x] = (uint32_t)(bitpal_bv_slot(threadIdx.x) * D * kBvBlock);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kBvBlock + threadIdx.x;
    bool reject = false, big = false;
    unsigned long long cells = 0;
    if (i < io.n) {
        const int pl = io.pat_len[i], tl = io.txt_len[i];
        const char *p = io.pat + io.pat_off[i], *t = io.txt + io.txt_off[i];
        int nr, nc; const char *rs, *cs;
        if (pl <= tl) { nr = pl; rs = p; nc = tl; cs = t; } else { nr = tl; rs = t; nc = pl; cs = p; }
        if (nr > 32 * D || nr > kBvMaxRows) { reject = true; big = nr > kLdsRows; }
        else if (nr == 0) io.score[i] = -nc;
        else {
            uint32_t *eq = eq_s + threadIdx.x;
#pragma unroll
            for (int k = 0; k < kBvSlots * D; k++) eq[k * kBvBlock] = 0;
            const uint32_t other = (uint32_t)(5 * D * kBvBlock);
            auto row_byte = [&](int j, uint32_t byte) {
                const uint32_t off = lut[byte];
                reject = reject || off == other;
                eq[off + (uint32_t)(j >> 5) * kBvBlock] |= 1u << (j & 31);
            };
            // sixteen bytes per load: every lane streams its own string, and with 4-byte loads each 64-byte line came from HBM
            // up to sixteen times (19 GB fetched per 10 M pairs against 3 GB of strings)
            int j0 = 0;
            for (; j0 + 16 <= nr; j0 += 16) {
                uint4 q; __builtin_memcpy(&q, rs + j0, 16);
                const uint32_t ws[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 16; k++) row_byte(j0 + k, (ws[k >> 2] >> ((k & 3) * 8)) & 0xffu);
            }
            for (; j0 < nr; j0 += 4) {                       // the last dwords through seq_ld4: nothing is read behind the string's last dword
                uint32_t w = seq_ld4(rs, j0, nr, 0u);
                for (int k = 0; k < 4 && j0 + k < nr; k++, w >>= 8) row_byte(j0 + k, w & 0xffu);
            }
            if (!reject) {
                uint32_t P[D], M[D];
#pragma unroll
                for (int d = 0; d < D; d++) { P[d] = ~0u; M[d] = 0; }
                auto step = [&](uint32_t byte) {
                    const uint32_t *e = eq + lut[byte];
                    uint32_t Eq[D];
#pragma unroll
                    for (int d = 0; d < D; d++) Eq[d] = e[d * kBvBlock];
                    gab_myers_step32<D>(Eq, P, M);
                };
                int h0 = 0;
                // thirty-two columns per trip, both 16-byte loads at once: a 64-byte line is visited twice instead of four times
                for (; h0 + 32 <= nc; h0 += 32) {
                    uint4 q, r; __builtin_memcpy(&q, cs + h0, 16); __builtin_memcpy(&r, cs + h0 + 16, 16);
                    const uint32_t ws[8] = {q.x, q.y, q.z, q.w, r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int kk = 0; kk < 32; kk++) step((ws[kk >> 2] >> ((kk & 3) * 8)) & 0xffu);
                }
                for (; h0 + 16 <= nc; h0 += 16) {
                    uint4 q; __builtin_memcpy(&q, cs + h0, 16);
                    const uint32_t ws[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int kk = 0; kk < 16; kk++) step((ws[kk >> 2] >> ((kk & 3) * 8)) & 0xffu);
                }
                for (; h0 < nc; h0 += 4) {
                    uint32_t w = seq_ld4(cs, h0, nc, 0u);
                    for (int kk = 0; kk < 4 && h0 + kk < nc; kk++, w >>= 8) step(w & 0xffu);
                }
                io.score[i] = -gab_myers_distance32<D>(P, M, nr, nc);
                cells = (unsigned long long)nc * (unsigned long long)nr;
            } else big = false;
        }
    }
    {   // the rejects' ids, in the lists the integer DP reads
        const uint32_t s_lds = gab_wave_slot(&cursors[0], reject && !big), s_big = gab_wave_slot(&cursors[1], reject && big);
        if (reject) { if (big) list_big[s_big] = (uint32_t)i; else list_lds[s_lds] = (uint32_t)i; }
    }
    for (int o = 32; o > 0; o >>= 1) cells += __shfl_xor(cells, o);
    __shared__ unsigned long long s_cells[kBvBlock / 64];
    if ((threadIdx.x & 63) == 0) s_cells[threadIdx.x >> 6] = cells;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long all = 0;
        for (int k = 0; k < kBvBlock / 64; k++) all += s_cells[k];
        if (all) atomicAdd(&ct->cells, all);
    }
}

}  // namespace

// =============================================================================== host side
struct gab_bitpal {
    gab_tuning tun = gab_tuning_loaded();      // experiment knobs, read when the handle is made
    gab_host_stream hs;     // private stream of the host-pointer entry point(s)
    int device = 0;
    BpScore sc;
    gab_devbuf ws;          // counters | 2 id lists
    gab_devbuf scratch;     // boundary columns of the global path
    gab_devbuf io;          // staging for the host-pointer entry point
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    BpCounters *h_ct = nullptr;
    bool have_stats = false;
};

extern "C" int gab_bitpal_create(int algorithm, int device, gab_bitpal **out) {
    if (!out) { gab_set_error("gab_bitpal_create: NULL argument"); return GAB_EINVAL; }
    *out = nullptr;
    GAB_CHECK(algorithm == GAB_BITPAL_EDIT || algorithm == GAB_BITPAL_SCORED,
              "gab_bitpal_create: algorithm %d is neither GAB_BITPAL_EDIT nor GAB_BITPAL_SCORED", algorithm);
    int rc = gab_check_device(device);
    if (rc) return rc;
    gab_device_guard g(device);
    gab_bitpal *h = new (std::nothrow) gab_bitpal();
    if (!h) { gab_set_error("out of host memory"); return GAB_ENOMEM; }
    h->device = device;
    if (algorithm == GAB_BITPAL_EDIT) h->sc = BpScore{0, -1, -1};      // bitpal.m0.x1.g1.c
    else h->sc = BpScore{1, -4, -2};                                   // bitpal.m1.x4.g2.c
    for (int k = 0; k < 3; k++)
        if (hipEventCreate(&h->ev[k]) != hipSuccess) { gab_set_error("hipEventCreate failed"); delete h; return GAB_EDEVICE; }
    if (hipHostMalloc((void **)&h->h_ct, sizeof(BpCounters)) != hipSuccess ||
        hipFuncSetAttribute((const void *)bitpal_dp<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void *)bitpal_dp<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        gab_set_error("gab_bitpal_create: pinned allocation / LDS attribute failed"); delete h; return GAB_EDEVICE;
    }
    *out = h;
    return GAB_OK;
}

extern "C" void gab_bitpal_destroy(gab_bitpal *h) {
    if (!h) return;
    gab_device_guard g(h->device);
    h->ws.release(); h->scratch.release(); h->io.release(); h->hs.release();
    for (int k = 0; k < 3; k++) if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
    if (h->h_ct) (void)hipHostFree(h->h_ct);
    delete h;
}

extern "C" int gab_bitpal_run_device(gab_bitpal *h, const char *pat, int64_t pat_bytes, const int64_t *pat_off,
                                     const int32_t *pat_len, const char *txt, int64_t txt_bytes, const int64_t *txt_off,
                                     const int32_t *txt_len, int64_t n, int32_t *score_out, void *stream_) {
    GAB_CHECK(h, "gab_bitpal_run_device: NULL handle");
    GAB_CHECK(n >= 0 && n < (1ll << 31), "gab_bitpal_run_device: n=%lld out of range", (long long)n);
    h->have_stats = false;
    if (n == 0) return GAB_OK;
    GAB_CHECK(pat && pat_off && pat_len && txt && txt_off && txt_len && score_out, "gab_bitpal_run_device: NULL buffer");
    gab_device_guard g(h->device);
    hipStream_t s = (hipStream_t)stream_;
    const size_t o_l0 = 256, o_l1 = o_l0 + 4 * (size_t)n;
    int rc = h->ws.reserve(o_l1 + 4 * (size_t)n);
    if (rc) return rc;
    char *base = h->ws.as<char>();
    BpCounters *d_ct = (BpCounters *)base;
    uint32_t *l_lds = (uint32_t *)(base + o_l0), *l_big = (uint32_t *)(base + o_l1);
    BpIO io{pat, pat_off, pat_len, txt, txt_off, txt_len, pat_bytes, txt_bytes, n, score_out};

    GAB_HIP(hipEventRecord(h->ev[0], s));
    memset(h->h_ct, 0, sizeof(BpCounters));
    h->h_ct->first_bad = 0x7fffffff;
    GAB_HIP(hipMemcpyAsync(d_ct, h->h_ct, sizeof(BpCounters), hipMemcpyHostToDevice, s));
    const int grid = (int)std::min<int64_t>(gab_ceil_div(n, 256), 512);
    hipLaunchKernelGGL(bitpal_classify, dim3(grid), dim3(256), 0, s, io, d_ct);
    GAB_HIP(hipMemcpyAsync(h->h_ct, d_ct, sizeof(BpCounters), hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    if (h->h_ct->bad) {
        gab_set_error("gab_bitpal_run_device: %d pair(s) violate the limits (first: pair %d): need 0 <= length <= %d and "
                      "offsets inside the slabs (readable to a multiple of 4 bytes)", h->h_ct->bad,
                      h->h_ct->first_bad - 1, GAB_BITPAL_MAX_LEN);
        return GAB_EINVAL;
    }
    uint32_t n_lds = h->h_ct->n_lds, n_big = h->h_ct->n_big;
    const bool scored = h->sc.match != 0;
    gab_tuning_refresh(&h->tun);
    const bool bitvec = !scored && !h->tun.bitpal_no_bv;      // (GAB_BITPAL_NO_BV)   // -a bitpal-edit: Myers' bit-vector, the integer DP for its rejects
    if (bitvec) {
        GAB_HIP(hipEventRecord(h->ev[1], s));
        const int rows = std::min(std::max(h->h_ct->max_rows_lds, 1), kBvMaxRows);
        const dim3 bv_grid((unsigned)gab_ceil_div(n, kBvBlock));
        switch ((rows + 31) / 32) {
            case 1: hipLaunchKernelGGL(bitpal_edit_bv<1>, bv_grid, dim3(kBvBlock), 0, s, io, d_ct->cursors, l_lds, l_big, d_ct); break;
            case 2: hipLaunchKernelGGL(bitpal_edit_bv<2>, bv_grid, dim3(kBvBlock), 0, s, io, d_ct->cursors, l_lds, l_big, d_ct); break;
            case 3: hipLaunchKernelGGL(bitpal_edit_bv<3>, bv_grid, dim3(kBvBlock), 0, s, io, d_ct->cursors, l_lds, l_big, d_ct); break;
            case 4: hipLaunchKernelGGL(bitpal_edit_bv<4>, bv_grid, dim3(kBvBlock), 0, s, io, d_ct->cursors, l_lds, l_big, d_ct); break;
            case 5: hipLaunchKernelGGL(bitpal_edit_bv<5>, bv_grid, dim3(kBvBlock), 0, s, io, d_ct->cursors, l_lds, l_big, d_ct); break;
            case 6: hipLaunchKernelGGL(bitpal_edit_bv<6>, bv_grid, dim3(kBvBlock), 0, s, io, d_ct->cursors, l_lds, l_big, d_ct); break;
            case 7: hipLaunchKernelGGL(bitpal_edit_bv<7>, bv_grid, dim3(kBvBlock), 0, s, io, d_ct->cursors, l_lds, l_big, d_ct); break;
            default: hipLaunchKernelGGL(bitpal_edit_bv<8>, bv_grid, dim3(kBvBlock), 0, s, io, d_ct->cursors, l_lds, l_big, d_ct); break;
        }
        GAB_HIP(hipGetLastError());
        GAB_HIP(hipMemcpyAsync(h->h_ct, d_ct, sizeof(BpCounters), hipMemcpyDeviceToHost, s));
        GAB_HIP(hipStreamSynchronize(s));
        n_lds = h->h_ct->cursors[0]; n_big = h->h_ct->cursors[1];      // what is left for the integer DP
    } else {
        if (n_big) hipLaunchKernelGGL(bitpal_scatter, dim3(grid), dim3(256), 0, s, io, d_ct->cursors, l_lds, l_big);
        else l_lds = nullptr;
        GAB_HIP(hipEventRecord(h->ev[1], s));
    }
    if (n_lds) {
        const size_t lds = (size_t)(h->h_ct->max_rows_lds + 1) * 64;
        auto kern = scored ? bitpal_dp<true, true> : bitpal_dp<true, false>;
        hipLaunchKernelGGL(kern, dim3((n_lds + 63) / 64), dim3(64), lds, s, io, l_lds, n_lds, h->sc, nullptr, 0, d_ct);
        GAB_HIP(hipGetLastError());
    }
    if (n_big) {
        const int64_t per_block = (int64_t)(h->h_ct->max_rows_big + 1) * 64;
        const int blocks = (int)std::min<int64_t>((n_big + 63) / 64, kBigBlocks);
        rc = h->scratch.reserve((size_t)per_block * 4 * (size_t)blocks);
        if (rc) return rc;
        auto kern = scored ? bitpal_dp<false, true> : bitpal_dp<false, false>;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, s, io, l_big, n_big, h->sc, h->scratch.as<int32_t>(),
                           per_block, d_ct);
        GAB_HIP(hipGetLastError());
    }
    GAB_HIP(hipMemcpyAsync(h->h_ct, d_ct, sizeof(BpCounters), hipMemcpyDeviceToHost, s));
    GAB_HIP(hipEventRecord(h->ev[2], s));
    GAB_HIP(hipStreamSynchronize(s));
    h->have_stats = true;
    return GAB_OK;
}

extern "C" int gab_bitpal_run(gab_bitpal *h, const char *pat, const int64_t *pat_off, const int32_t *pat_len,
                              const char *txt, const int64_t *txt_off, const int32_t *txt_len, int64_t n,
                              int32_t *score_out) {
    GAB_CHECK(h, "gab_bitpal_run: NULL handle");
    GAB_CHECK(n >= 0 && n < (1ll << 31), "gab_bitpal_run: n=%lld out of range", (long long)n);
    if (n == 0) return GAB_OK;
    GAB_CHECK(pat && pat_off && pat_len && txt && txt_off && txt_len && score_out, "gab_bitpal_run: NULL buffer");
    gab_device_guard g(h->device);
    int64_t pb = 0, tb = 0, pa = INT64_MAX, ta = INT64_MAX;
    for (int64_t i = 0; i < n; i++) {
        GAB_CHECK(pat_off[i] >= 0 && txt_off[i] >= 0 && pat_len[i] >= 0 && txt_len[i] >= 0,
                  "gab_bitpal_run: negative offset/length at pair %lld", (long long)i);
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
    rc = gab_bitpal_run_device(h, b + o_p - pa, pa + (int64_t)ppad, (const int64_t *)(b + o_po), (const int32_t *)(b + o_pl),
                               (shared ? b + o_p : b + o_t) - ta, ta + (int64_t)(shared ? ppad : tpad), (const int64_t *)(b + o_to), (const int32_t *)(b + o_tl), n,
                               (int32_t *)(b + o_sc), s);
    if (rc) return rc;
    GAB_HIP(hipMemcpyAsync(score_out, b + o_sc, 4 * nn, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    return GAB_OK;
}

// see gab_bpm_reserve
extern "C" int gab_bitpal_reserve(gab_bitpal *h, int64_t max_pairs, int64_t max_seq_bytes) {
    GAB_CHECK(h, "gab_bitpal_reserve: NULL handle");
    GAB_CHECK(max_pairs >= 0 && max_pairs < (1ll << 31) && max_seq_bytes >= 0, "gab_bitpal_reserve: size out of range");
    gab_device_guard g(h->device);
    const size_t nn = (size_t)max_pairs;
    int rc = h->io.reserve(std::max<size_t>(2 * (((size_t)max_seq_bytes + 3 + 511) & ~(size_t)255) + 28 * nn + 1024, (size_t)4 << 20));
    if (rc) return rc;
    if ((rc = h->ws.reserve(sizeof(BpCounters) + 512 + 3 * 4 * nn)) != GAB_OK) return rc;
    hipStream_t s = nullptr;
    if ((rc = h->hs.get(&s)) != GAB_OK) return rc;
    GAB_HIP(hipMemsetAsync(h->io.p, 0, h->io.cap, s));
    GAB_HIP(hipMemsetAsync(h->ws.p, 0, h->ws.cap, s));
    GAB_HIP(hipStreamSynchronize(s));
    return gab_warm_copy_engines(s, h->io.p, h->io.cap);
}

extern "C" int gab_bitpal_last_stats(gab_bitpal *h, int64_t *cells, int64_t *long_pairs, float *kernel_ms, float *total_ms) {
    GAB_CHECK(h, "gab_bitpal_last_stats: NULL handle");
    GAB_CHECK(h->have_stats, "gab_bitpal_last_stats: no completed run on this handle");
    gab_device_guard g(h->device);
    GAB_HIP(hipEventSynchronize(h->ev[2]));
    if (cells) *cells = (int64_t)h->h_ct->cells;
    if (long_pairs) *long_pairs = (int64_t)h->h_ct->n_big;
    if (kernel_ms) GAB_HIP(hipEventElapsedTime(kernel_ms, h->ev[1], h->ev[2]));
    if (total_ms) GAB_HIP(hipEventElapsedTime(total_ms, h->ev[0], h->ev[2]));
    return GAB_OK;
}
