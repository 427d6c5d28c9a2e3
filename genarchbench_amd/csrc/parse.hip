// parse -- text inputs -> packed device buffers, on gfx950.
//
// Replaces, outside the region of interest, the line-by-line host parsers of the reference drivers:
//   bsw      loadPairs                      /root/reference/benchmarks/bsw/src/main_banded.cpp:164-206 (+ :237-253)
//   bpm      getline loop + swap            /root/reference/benchmarks/bpm/tools/align_benchmark.c:150-200, 247-252
//   wfa      parse_input_sequences          /root/reference/benchmarks/wfa/tools/align_benchmark.c:110-194
// The whole file is one buffer.  Three streaming passes build a newline index (count per 16 KB block, scan, fill),
// then one thread per pair derives lengths / h0, two scans give the slab offsets and one wave per pair copies the
// code bytes (64 consecutive bytes per instruction).  Everything is HBM streaming: the bound is bandwidth, and the
// text is read three times (once per pass) -- the roofline of bench.py's parse-bsw workload counts it once.
#include "gab_internal.h"
#include <algorithm>
#include <new>
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int kSubBytes = 16384;              // one sweep of a workgroup: 256 threads x 64 B
constexpr int kSubs = 8;                      // sweeps per workgroup (fewer, larger blocks keep the single-workgroup scan short)
constexpr int kBlockBytes = kSubBytes * kSubs; // text bytes per workgroup in the newline passes

struct ParseFlags { int32_t bad; int32_t first_bad; };

// ---- newline index ---------------------------------------------------------------------------------------------
// 0x80 in every byte of w that is '\n', exactly (the add cannot carry across bytes)
__device__ __forceinline__ uint32_t nl_flags(uint32_t w) {
    const uint32_t x = w ^ 0x0a0a0a0au;
    const uint32_t t = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    return ~(t | x | 0x7f7f7f7fu);
}
__global__ __launch_bounds__(256) void nl_count(const char *__restrict__ text, int64_t n, int64_t *block_cnt) {
    __shared__ int sh[4];
    int c = 0;
    for (int sub = 0; sub < kSubs; sub++) {
        const int64_t base = (int64_t)blockIdx.x * kBlockBytes + (int64_t)sub * kSubBytes + (int64_t)threadIdx.x * 64;
        if (base + 64 <= n) {
            const uint4 *p = reinterpret_cast<const uint4 *>(text + base);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4 v = p[k];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; j++) c += __popc(nl_flags(w[j]));
            }
        } else {
            for (int k = 0; k < 64; k++) if (base + k < n && text[base + k] == '\n') c++;
        }
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = (int64_t)(sh[0] + sh[1] + sh[2] + sh[3]);
}
// exclusive scan of int64 values, single workgroup (the arrays here have at most a few hundred thousand entries);
// out[n] receives the total.  Eight consecutive values per thread, wave scans by shuffles, one LDS step across the sixteen
// waves: three barriers per 8192 values (the LDS scan of one value per thread took twenty per 1024: 60 us for 39 000 values).
__global__ __launch_bounds__(1024) void scan_i64(const int64_t *in, int64_t n, int64_t *out) {
    __shared__ int64_t wsum[16];
    constexpr int kPer = 8;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t carry = 0;
    for (int64_t c0 = 0; c0 < n; c0 += 1024 * kPer) {
        const int64_t i0 = c0 + (int64_t)threadIdx.x * kPer;
        int64_t v[kPer], sum = 0;
#pragma unroll
        for (int k = 0; k < kPer; k++) { v[k] = i0 + k < n ? in[i0 + k] : 0; sum += v[k]; }
        int64_t inc = sum;
        for (int o = 1; o < 64; o <<= 1) { const int64_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int64_t before = carry, total = 0;
        for (int k = 0; k < 16; k++) { if (k < wv) before += wsum[k]; total += wsum[k]; }
        int64_t run = before + inc - sum;
#pragma unroll
        for (int k = 0; k < kPer; k++) { if (i0 + k < n) out[i0 + k] = run; run += v[k]; }
        __syncthreads();
        carry += total;
    }
    if (threadIdx.x == 0) out[n] = carry;
}
// line_start[l + 1] = position after the l-th newline; line_start[0] = 0
__global__ __launch_bounds__(256) void nl_fill(const char *__restrict__ text, int64_t n, const int64_t *block_off, int64_t *line_start) {
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t carry = block_off[blockIdx.x];                    // newlines before this sweep
    for (int sub = 0; sub < kSubs; sub++) {
        const int64_t base = (int64_t)blockIdx.x * kBlockBytes + (int64_t)sub * kSubBytes + (int64_t)threadIdx.x * 64;
        // 64 bytes per thread as a bit mask of newlines
        unsigned long long m = 0;
        if (base + 64 <= n) {
            const uint4 *p = reinterpret_cast<const uint4 *>(text + base);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4 v = p[k];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t t = nl_flags(w[j]);
                    const uint32_t bits = ((t >> 7) & 1u) | ((t >> 14) & 2u) | ((t >> 21) & 4u) | ((t >> 28) & 8u);
                    m |= (unsigned long long)bits << (16 * k + 4 * j);
                }
            }
        } else {
            for (int k = 0; k < 64; k++) if (base + k < n && text[base + k] == '\n') m |= 1ull << k;
        }
        const int c = __popcll(m);
        // exclusive prefix of c over the workgroup
        int incl = c;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        __syncthreads();                                      // (wsum of the previous sweep has been read by everyone)
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; w++) before += wsum[w];
        int64_t idx = carry + before + incl - c;
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            line_start[++idx] = base + b + 1;
        }
        carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) line_start[0] = 0;
}

// ---- bsw ---------------------------------------------------------------------------------------------------------
// per pair: h0 (sscanf "%d" on at most 9 characters, main_banded.cpp:179-180), len1 / len2 = line length without '\n'
__global__ __launch_bounds__(256) void bsw_meta(const char *__restrict__ text, const int64_t *__restrict__ ls, int64_t npairs,
                                                int32_t *len1, int32_t *len2, int32_t *h0, ParseFlags *fl) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const int64_t s0 = ls[3 * i], s1 = ls[3 * i + 1], s2 = ls[3 * i + 2], s3 = ls[3 * i + 3];
    const int64_t hl = s1 - s0 - 1, l1 = s2 - s1 - 1, l2 = s3 - s2 - 1;
    bool ok = hl >= 0 && hl <= 8 && l1 >= 1 && l1 <= 2045 && l2 >= 1 && l2 <= 253;
    int v = 0;
    if (ok) {
        int64_t p = s0; const int64_t e = s1 - 1;
        while (p < e && (text[p] == ' ' || text[p] == '\t')) p++;
        bool neg = false;
        if (p < e && (text[p] == '-' || text[p] == '+')) { neg = text[p] == '-'; p++; }
        long long acc = 0;
        while (p < e && text[p] >= '0' && text[p] <= '9') { acc = acc * 10 + (text[p] - '0'); p++; }
        v = (int)(neg ? -acc : acc);
    }
    if (!ok) {
        atomicAdd(&fl->bad, 1);
        atomicMin((unsigned int *)&fl->first_bad, (unsigned int)(i > 0x7ffffffe ? 0x7ffffffe : i));
        len1[i] = 0; len2[i] = 0; h0[i] = 0;
        return;
    }
    len1[i] = (int32_t)l1; len2[i] = (int32_t)l2; h0[i] = v;
}
// exclusive scan of int32 lengths to int64 offsets: block sums, scan (scan_i64), apply
// (lengths are rounded up to a multiple of 4: every sequence starts dword-aligned in its slab)
__global__ __launch_bounds__(256) void len_block_sums(const int32_t *len, int64_t n, int64_t *block_sums) {
    __shared__ int64_t sh[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    sh[threadIdx.x] = i < n ? ((len[i] + 3) & ~3) : 0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void len_offsets(const int32_t *len, int64_t n, const int64_t *block_off, int64_t *off) {
    __shared__ int64_t sh[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t c = i < n ? ((len[i] + 3) & ~3) : 0;
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int64_t v = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    if (i < n) off[i] = block_off[blockIdx.x] + sh[threadIdx.x] - c;
}
// code = character - '0' (main_banded.cpp:192-193).  A wave takes 64 pairs: every lane loads the metadata of one pair
// (coalesced), then each pair is copied with FOUR bytes per lane (one unaligned dword load from the text, one aligned
// dword store into the slab: sequences start dword-aligned there) -- byte-wide accesses run at a quarter of the
// address-unit rate.  The up-to-3 bytes past a sequence's end are padding nobody reads.
// broadcast of a lane's value when the lane index is wave-uniform: v_readlane instead of an LDS permute
__device__ __forceinline__ int bcast32(int v, int lane_uniform) { return __builtin_amdgcn_readlane(v, lane_uniform); }
__device__ __forceinline__ int64_t bcast64(int64_t v, int lane_uniform) {
    return (int64_t)((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), lane_uniform) << 32 |
                     (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane_uniform));
}
__device__ __forceinline__ uint32_t text_ld4(const char *text, int64_t pos, int64_t nbytes) {
    uint32_t w = 0;
    if (pos + 4 <= nbytes) __builtin_memcpy(&w, text + pos, 4);
    else for (int b = 0; b < 4; b++) if (pos + b < nbytes) w |= (uint32_t)(uint8_t)text[pos + b] << (8 * b);
    return w;
}
// four independent byte subtractions x - '0' (no borrow between bytes: a character below '0' wraps like the reference's
// uint8 arithmetic and must not disturb its neighbour)
__device__ __forceinline__ uint32_t sub48x4(uint32_t x) {
    const uint32_t H = 0x80808080u, y = 0x30303030u;
    return ((x | H) - (y & ~H)) ^ ((x ^ ~y) & H);
}
constexpr int kCopyUnroll = 4;                // pairs whose loads are in flight together
__global__ __launch_bounds__(256) void bsw_codes(const char *__restrict__ text, int64_t nbytes, const int64_t *__restrict__ ls,
                                                 int64_t npairs, const int64_t *__restrict__ ref_off,
                                                 const int64_t *__restrict__ qry_off, const int32_t *__restrict__ len1,
                                                 const int32_t *__restrict__ len2, uint8_t *ref, uint8_t *qry) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t b0 = wave * 64; b0 < npairs; b0 += nwaves * 64) {
        const int64_t mine = b0 + lane;
        const bool have = mine < npairs;
        const int64_t m_s1 = have ? ls[3 * mine + 1] : 0, m_s2 = have ? ls[3 * mine + 2] : 0;
        const int64_t m_ro = have ? ref_off[mine] : 0, m_qo = have ? qry_off[mine] : 0;
        const int m_l1 = have ? len1[mine] : 0, m_l2 = have ? len2[mine] : 0;
        const int cnt = __builtin_amdgcn_readfirstlane((int)(npairs - b0 < 64 ? npairs - b0 : 64));
        for (int j0 = 0; j0 < cnt; j0 += kCopyUnroll) {
            // the dwords of a pair are numbered ref first, then query: lane v < nR copies reference dword v, the other
            // lanes query dword v - nR, so a typical 160 + 80 byte pair is ONE load and ONE store instruction
            uint32_t wv[kCopyUnroll];
            int64_t s1[kCopyUnroll], s2[kCopyUnroll], ro[kCopyUnroll], qo[kCopyUnroll]; int nR[kCopyUnroll], nT[kCopyUnroll];
#pragma unroll
            for (int u = 0; u < kCopyUnroll; u++) {
                const int j = j0 + u < cnt ? j0 + u : cnt - 1;          // (a duplicate of the last pair rewrites the same bytes)
                s1[u] = bcast64(m_s1, j); s2[u] = bcast64(m_s2, j); ro[u] = bcast64(m_ro, j); qo[u] = bcast64(m_qo, j);
                nR[u] = (bcast32(m_l1, j) + 3) >> 2; nT[u] = nR[u] + ((bcast32(m_l2, j) + 3) >> 2);
                const bool isq = lane >= nR[u];
                wv[u] = lane < nT[u] ? text_ld4(text, (isq ? s2[u] - 4 * (int64_t)nR[u] : s1[u]) + 4 * lane, nbytes) : 0u;
            }
#pragma unroll
            for (int u = 0; u < kCopyUnroll; u++) {
                const bool isq = lane >= nR[u];
                if (lane < nT[u]) *reinterpret_cast<uint32_t *>(isq ? qry + qo[u] + 4 * (lane - nR[u]) : ref + ro[u] + 4 * lane) = sub48x4(wv[u]);
                for (int v = 64 + lane; v < nT[u]; v += 64) {           // pairs longer than 256 characters
                    const bool q2 = v >= nR[u];
                    const uint32_t w = text_ld4(text, (q2 ? s2[u] - 4 * (int64_t)nR[u] : s1[u]) + 4 * (int64_t)v, nbytes);
                    *reinterpret_cast<uint32_t *>(q2 ? qry + qo[u] + 4 * (v - nR[u]) : ref + ro[u] + 4 * v) = sub48x4(w);
                }
            }
        }
    }
}

// ---- bpm / wfa -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pairs_meta(const char *__restrict__ text, const int64_t *__restrict__ ls, int64_t npairs, int swap,
                                                  int64_t *pat_off, int32_t *pat_len, int64_t *txt_off, int32_t *txt_len, int32_t *sum_len,
                                                  ParseFlags *fl) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const int64_t s0 = ls[2 * i], s1 = ls[2 * i + 1], s2 = ls[2 * i + 2];
    // a line is <prefix char> <sequence> '\n'; the drivers drop the first character and the newline unconditionally
    int64_t ao = s0 + 1, al = s1 - s0 - 2, bo = s1 + 1, bl = s2 - s1 - 2;
    const bool ok = al >= 0 && bl >= 0 && al <= 0x7ffffff0 && bl <= 0x7ffffff0;
    if (!ok) {
        atomicAdd(&fl->bad, 1);
        atomicMin((unsigned int *)&fl->first_bad, (unsigned int)(i > 0x7ffffffe ? 0x7ffffffe : i));
        al = bl = 0;
    }
    if (swap && bl > al) {                       // the longer LINE becomes the pattern (bpm align_benchmark.c:177-181)
        const int64_t to = ao, tl = al; ao = bo; al = bl; bo = to; bl = tl;
    }
    pat_off[i] = ao; pat_len[i] = (int32_t)al; txt_off[i] = bo; txt_len[i] = (int32_t)bl;
    sum_len[i] = (int32_t)(al + bl);             // capacity of a per-pair output of pattern_length + text_length bytes (wfa CIGAR)
}

// ---- chain / fast-chain ------------------------------------------------------------------------------------------------
// read_call (chain/src/host_data_io.cpp:13-51) is token based: 6 header fields, n x "x y", everything up to "EOR".  The
// files the suite ships (and print_return-style writers produce) are line based: one header line, n anchor lines, one
// "EOR" line.  The GPU path handles exactly that layout and verifies it (header n == number of anchor lines, two
// unsigned decimal tokens per anchor line); anything else is declined.  The header lines themselves are parsed on the
// host with the reference's own fscanf format, so the float avg_qspan is converted by the C library as in the reference.
constexpr int kHdrSlot = 128;             // bytes reserved per header line handed to the host

__global__ __launch_bounds__(256) void chain_mark_eor(const char *__restrict__ text, const int64_t *__restrict__ ls, int64_t nlines,
                                                      int32_t *is_eor) {
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= nlines) return;
    const int64_t s = ls[l], e = ls[l + 1] - 1;                // [s, e) without the newline
    is_eor[l] = (e - s == 3 && text[s] == 'E' && text[s + 1] == 'O' && text[s + 2] == 'R') ? 1 : 0;
}
// eor_before[l] = number of EOR lines before line l (len_offsets on is_eor, unpadded) is computed by the caller with
// the plain scan below
__global__ __launch_bounds__(256) void flag_block_sums(const int32_t *v, int64_t n, int64_t *block_sums) {
    __shared__ int64_t sh[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    sh[threadIdx.x] = i < n ? v[i] : 0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void flag_offsets(const int32_t *v, int64_t n, const int64_t *block_off, int64_t *off) {
    __shared__ int64_t sh[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t c = i < n ? v[i] : 0;
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int64_t a = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += a;
        __syncthreads();
    }
    if (i < n) off[i] = block_off[blockIdx.x] + sh[threadIdx.x] - c;
}
// per EOR line: the call it closes -> header line index, anchor count, and a copy of the header line for the host
__global__ __launch_bounds__(256) void chain_calls(const char *__restrict__ text, const int64_t *__restrict__ ls, int64_t nlines,
                                                   const int32_t *__restrict__ is_eor, const int64_t *__restrict__ eor_before,
                                                   int64_t *eor_line, ParseFlags *fl) {
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= nlines || !is_eor[l]) return;
    eor_line[eor_before[l]] = l;
    (void)text; (void)ls; (void)fl;
}
__global__ __launch_bounds__(256) void chain_headers(const char *__restrict__ text, const int64_t *__restrict__ ls,
                                                     const int64_t *__restrict__ eor_line, int64_t ncalls, int32_t *n_anchor,
                                                     char *hdr_text, ParseFlags *fl) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= ncalls) return;
    const int64_t h = c == 0 ? 0 : eor_line[c - 1] + 1, e = eor_line[c];      // header line, EOR line
    const int64_t na = e - h - 1;
    const int64_t s = ls[h < e ? h : e], len = ls[(h < e ? h : e) + 1] - 1 - s;
    const bool ok = na >= 0 && na < (1ll << 31) && len >= 1 && len < kHdrSlot;
    if (!ok) {
        atomicAdd(&fl->bad, 1);
        atomicMin((unsigned int *)&fl->first_bad, (unsigned int)(c > 0x7ffffffe ? 0x7ffffffe : c));
        n_anchor[c] = 0;
        hdr_text[c * kHdrSlot] = 0;
        return;
    }
    n_anchor[c] = (int32_t)na;
    for (int64_t k = 0; k < len; k++) hdr_text[c * kHdrSlot + k] = text[s + k];
    hdr_text[c * kHdrSlot + len] = 0;
}
// one thread per line: an anchor line is "<x> <y>" with unsigned decimal tokens (fscanf "%llu%llu")
__global__ __launch_bounds__(256) void chain_anchors(const char *__restrict__ text, const int64_t *__restrict__ ls, int64_t nlines,
                                                     const int32_t *__restrict__ is_eor, const int64_t *__restrict__ eor_before,
                                                     const int64_t *__restrict__ eor_line, const int64_t *__restrict__ call_off,
                                                     int64_t ncalls, uint64_t *x, uint64_t *y, ParseFlags *fl) {
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= nlines || is_eor[l]) return;
    const int64_t c = eor_before[l];
    if (c >= ncalls) return;                                   // trailing lines after the last EOR: read_call stops there
    const int64_t h = c == 0 ? 0 : eor_line[c - 1] + 1;
    if (l == h) return;                                        // the header line
    int64_t p = ls[l]; const int64_t e = ls[l + 1] - 1;
    uint64_t v[2] = {0, 0};
    bool ok = true;
    for (int t = 0; t < 2 && ok; t++) {
        while (p < e && (text[p] == ' ' || text[p] == '\t')) p++;
        int digits = 0; uint64_t acc = 0;
        while (p < e && text[p] >= '0' && text[p] <= '9') {
            const uint64_t d = (uint64_t)(text[p] - '0');
            if (acc > (0xffffffffffffffffull - d) / 10) ok = false;       // would overflow: strtoull saturates, decline
            acc = acc * 10 + d; digits++; p++;
        }
        ok = ok && digits > 0;
        v[t] = acc;
    }
    while (p < e && (text[p] == ' ' || text[p] == '\t' || text[p] == '\r')) p++;
    ok = ok && p == e;
    if (!ok) {
        atomicAdd(&fl->bad, 1);
        atomicMin((unsigned int *)&fl->first_bad, (unsigned int)(c > 0x7ffffffe ? 0x7ffffffe : c));
        return;
    }
    const int64_t o = call_off[c] + (l - h - 1);
    x[o] = v[0]; y[o] = v[1];
}

}  // namespace

// =============================================================================== host side
struct gab_parser {
    int device = 0;
    gab_devbuf text;        // device copy of the file (host-pointer entry points)
    gab_devbuf idx;         // block counts / offsets, line starts
    gab_devbuf meta;        // per-pair arrays
    gab_devbuf slabs;       // bsw code slabs
    gab_devbuf ws;          // flags + newline block counts / offsets
    gab_devbuf scan;        // block sums of the length scans
    gab_devbuf chain;       // chain: per-line flags / prefix, EOR lines, anchor counts, header copies, call offsets
    gab_devbuf anchors;     // chain: x | y
    void *h_calls = nullptr; size_t h_calls_cap = 0;       // host: call_off + hdr handed to the caller
    hipEvent_t ev[2] = {nullptr, nullptr};
    float kernel_ms = 0; bool have_stats = false;
};

extern "C" int gab_parser_create(int device, gab_parser **out) {
    if (!out) { gab_set_error("gab_parser_create: NULL argument"); return GAB_EINVAL; }
    *out = nullptr;
    int rc = gab_check_device(device);
    if (rc) return rc;
    gab_device_guard g(device);
    gab_parser *p = new (std::nothrow) gab_parser();
    if (!p) { gab_set_error("out of host memory"); return GAB_ENOMEM; }
    p->device = device;
    if (hipEventCreate(&p->ev[0]) != hipSuccess || hipEventCreate(&p->ev[1]) != hipSuccess) {
        gab_set_error("gab_parser_create: event creation failed"); delete p; return GAB_EDEVICE;
    }
    *out = p;
    return GAB_OK;
}

extern "C" void gab_parser_destroy(gab_parser *p) {
    if (!p) return;
    gab_device_guard g(p->device);
    p->text.release(); p->idx.release(); p->meta.release(); p->slabs.release(); p->ws.release(); p->scan.release(); p->chain.release(); p->anchors.release(); free(p->h_calls);
    for (int k = 0; k < 2; k++) if (p->ev[k]) (void)hipEventDestroy(p->ev[k]);
    delete p;
}

// newline index of d_text: returns the number of lines (= newlines) and the device array line_start[lines + 1]
static int build_line_index(gab_parser *p, const char *d_text, int64_t nbytes, hipStream_t s, int64_t *nlines, const int64_t **d_ls) {
    const int64_t nblocks = gab_ceil_div(nbytes, kBlockBytes);
    // ws: [flags 64 B][block counts nblocks][block offsets nblocks + 1]
    int rc = p->ws.reserve(64 + 8 * (size_t)(2 * nblocks + 2));
    if (rc) return rc;
    int64_t *d_cnt = (int64_t *)(p->ws.as<char>() + 64), *d_off = d_cnt + nblocks;
    hipLaunchKernelGGL(nl_count, dim3((unsigned)nblocks), dim3(256), 0, s, d_text, nbytes, d_cnt);
    hipLaunchKernelGGL(scan_i64, dim3(1), dim3(1024), 0, s, d_cnt, nblocks, d_off);
    GAB_HIP(hipGetLastError());
    int64_t total = 0;
    GAB_HIP(hipMemcpyAsync(&total, d_off + nblocks, 8, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    rc = p->idx.reserve(8 * (size_t)(total + 2));
    if (rc) return rc;
    hipLaunchKernelGGL(nl_fill, dim3((unsigned)nblocks), dim3(256), 0, s, d_text, nbytes, d_off, p->idx.as<int64_t>());
    GAB_HIP(hipGetLastError());
    *nlines = total; *d_ls = p->idx.as<int64_t>();
    return GAB_OK;
}

static int scan_lengths(gab_parser *p, const int32_t *d_len, int64_t n, int64_t *d_off, int64_t *total, hipStream_t s) {
    const int64_t blocks = gab_ceil_div(n, 256);
    int rc = p->scan.reserve(16 * (size_t)(blocks + 4));
    if (rc) return rc;
    int64_t *d_bs = p->scan.as<int64_t>(), *d_bo = d_bs + blocks + 1;
    hipLaunchKernelGGL(len_block_sums, dim3((unsigned)blocks), dim3(256), 0, s, d_len, n, d_bs);
    hipLaunchKernelGGL(scan_i64, dim3(1), dim3(1024), 0, s, d_bs, blocks, d_bo);
    hipLaunchKernelGGL(len_offsets, dim3((unsigned)blocks), dim3(256), 0, s, d_len, n, d_bo, d_off);
    GAB_HIP(hipGetLastError());
    GAB_HIP(hipMemcpyAsync(total, d_bo + blocks, 8, hipMemcpyDeviceToHost, s));
    return GAB_OK;
}

extern "C" int gab_bsw_parse_pairs_device(gab_parser *p, const char *d_text, int64_t nbytes, gab_bsw_packed *out, void *stream_) {
    GAB_CHECK(p && out, "gab_bsw_parse_pairs_device: NULL argument");
    memset(out, 0, sizeof *out);
    GAB_CHECK(nbytes >= 0, "gab_bsw_parse_pairs_device: nbytes < 0");
    p->have_stats = false;
    if (nbytes == 0) return GAB_OK;
    GAB_CHECK(d_text, "gab_bsw_parse_pairs_device: NULL text");
    GAB_CHECK(((uintptr_t)d_text & 15) == 0, "gab_bsw_parse_pairs_device: the text buffer must be 16-byte aligned");
    gab_device_guard g(p->device);
    hipStream_t s = (hipStream_t)stream_;
    GAB_HIP(hipEventRecord(p->ev[0], s));
    int64_t nlines = 0; const int64_t *d_ls = nullptr;
    int rc = build_line_index(p, d_text, nbytes, s, &nlines, &d_ls);
    if (rc) return rc;
    const int64_t n = nlines / 3;                        // numPairs = newline count / 3 (main_banded.cpp:237-253)
    if (n == 0) { GAB_HIP(hipEventRecord(p->ev[1], s)); return GAB_OK; }
    GAB_CHECK(n < (1ll << 31), "gab_bsw_parse_pairs_device: too many pairs");
    // meta: len1 | len2 | h0 (int32 x n each) | ref_off | qry_off (int64 x (n + 1) each)
    const size_t o_l2 = (4 * (size_t)n + 63) & ~(size_t)63, o_h0 = 2 * o_l2, o_ro = 3 * o_l2, o_qo = o_ro + ((8 * (size_t)(n + 1) + 63) & ~(size_t)63);
    rc = p->meta.reserve(o_qo + 8 * (size_t)(n + 1) + 64);
    if (rc) return rc;
    char *mb = p->meta.as<char>();
    int32_t *d_l1 = (int32_t *)mb, *d_l2 = (int32_t *)(mb + o_l2), *d_h0 = (int32_t *)(mb + o_h0);
    int64_t *d_ro = (int64_t *)(mb + o_ro), *d_qo = (int64_t *)(mb + o_qo);
    ParseFlags *d_fl = (ParseFlags *)p->ws.as<char>();
    ParseFlags hf = {0, 0x7fffffff};
    GAB_HIP(hipMemcpyAsync(d_fl, &hf, sizeof hf, hipMemcpyHostToDevice, s));
    const int64_t blocks = gab_ceil_div(n, 256);
    hipLaunchKernelGGL(bsw_meta, dim3((unsigned)blocks), dim3(256), 0, s, d_text, d_ls, n, d_l1, d_l2, d_h0, d_fl);
    int64_t tot1 = 0, tot2 = 0;
    rc = scan_lengths(p, d_l1, n, d_ro, &tot1, s);
    if (rc) return rc;
    GAB_HIP(hipStreamSynchronize(s));                    // (the two scans share the scratch and the host totals)
    rc = scan_lengths(p, d_l2, n, d_qo, &tot2, s);
    if (rc) return rc;
    GAB_HIP(hipMemcpyAsync(&hf, d_fl, sizeof hf, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    if (hf.bad) {
        gab_set_error("gab_bsw_parse_pairs: %d pair(s) are not in the plain format (first: pair %d): empty or over-long line",
                      hf.bad, hf.first_bad);
        return GAB_EINVAL;
    }
    const size_t rpad = ((size_t)tot1 + 3 + 255) & ~(size_t)255, qpad = ((size_t)tot2 + 3 + 255) & ~(size_t)255;
    rc = p->slabs.reserve(rpad + qpad);
    if (rc) return rc;
    uint8_t *d_ref = p->slabs.as<uint8_t>(), *d_qry = d_ref + rpad;
    int n_cu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, p->device) == hipSuccess) n_cu = prop.multiProcessorCount; }
    hipLaunchKernelGGL(bsw_codes, dim3((unsigned)std::min<int64_t>(gab_ceil_div(n, 256), (int64_t)n_cu * 16)), dim3(256), 0, s, d_text, nbytes, d_ls, n,
                       d_ro, d_qo, d_l1, d_l2, d_ref, d_qry);
    GAB_HIP(hipGetLastError());
    GAB_HIP(hipEventRecord(p->ev[1], s));
    GAB_HIP(hipStreamSynchronize(s));
    GAB_HIP(hipEventElapsedTime(&p->kernel_ms, p->ev[0], p->ev[1]));
    p->have_stats = true;
    out->n = n; out->d_ref = d_ref; out->d_ref_off = d_ro; out->d_qry = d_qry; out->d_qry_off = d_qo;
    out->d_len1 = d_l1; out->d_len2 = d_l2; out->d_h0 = d_h0; out->ref_bytes = (int64_t)rpad; out->qry_bytes = (int64_t)qpad;
    return GAB_OK;
}

static int stage_text(gab_parser *p, const char *text, int64_t nbytes, hipStream_t s, const char **d_text) {
    int rc = p->text.reserve((size_t)nbytes + 64);
    if (rc) return rc;
    GAB_HIP(hipMemcpyAsync(p->text.p, text, (size_t)nbytes, hipMemcpyHostToDevice, s));
    *d_text = p->text.as<char>();
    return GAB_OK;
}

extern "C" int gab_bsw_parse_pairs(gab_parser *p, const char *text, int64_t nbytes, gab_bsw_packed *out, void *stream_) {
    GAB_CHECK(p && out, "gab_bsw_parse_pairs: NULL argument");
    GAB_CHECK(nbytes >= 0 && (nbytes == 0 || text), "gab_bsw_parse_pairs: bad buffer");
    if (nbytes == 0) { memset(out, 0, sizeof *out); return GAB_OK; }
    gab_device_guard g(p->device);
    const char *d_text = nullptr;
    int rc = stage_text(p, text, nbytes, (hipStream_t)stream_, &d_text);
    if (rc) return rc;
    return gab_bsw_parse_pairs_device(p, d_text, nbytes, out, stream_);
}

extern "C" int gab_pairs_parse_device(gab_parser *p, const char *d_text, int64_t nbytes, int swap_longer_first, gab_pairs_packed *out,
                                      void *stream_) {
    GAB_CHECK(p && out, "gab_pairs_parse_device: NULL argument");
    memset(out, 0, sizeof *out);
    GAB_CHECK(nbytes >= 0, "gab_pairs_parse_device: nbytes < 0");
    p->have_stats = false;
    if (nbytes == 0) return GAB_OK;
    GAB_CHECK(d_text, "gab_pairs_parse_device: NULL text");
    GAB_CHECK(((uintptr_t)d_text & 15) == 0, "gab_pairs_parse_device: the text buffer must be 16-byte aligned");
    gab_device_guard g(p->device);
    hipStream_t s = (hipStream_t)stream_;
    GAB_HIP(hipEventRecord(p->ev[0], s));
    int64_t nlines = 0; const int64_t *d_ls = nullptr;
    int rc = build_line_index(p, d_text, nbytes, s, &nlines, &d_ls);
    if (rc) return rc;
    const int64_t n = nlines / 2;
    out->d_text = d_text; out->text_bytes = nbytes;
    if (n == 0) { GAB_HIP(hipEventRecord(p->ev[1], s)); return GAB_OK; }
    GAB_CHECK(n < (1ll << 31), "gab_pairs_parse_device: too many pairs");
    const size_t o8 = (8 * (size_t)(n + 1) + 63) & ~(size_t)63, o4 = (4 * (size_t)n + 63) & ~(size_t)63;
    rc = p->meta.reserve(3 * o8 + 3 * o4 + 64);
    if (rc) return rc;
    char *mb = p->meta.as<char>();
    int64_t *d_po = (int64_t *)mb, *d_to = (int64_t *)(mb + o8), *d_co = (int64_t *)(mb + 2 * o8);
    int32_t *d_pl = (int32_t *)(mb + 3 * o8), *d_tl = (int32_t *)(mb + 3 * o8 + o4), *d_sl = (int32_t *)(mb + 3 * o8 + 2 * o4);
    ParseFlags *d_fl = (ParseFlags *)p->ws.as<char>();
    ParseFlags hf = {0, 0x7fffffff};
    GAB_HIP(hipMemcpyAsync(d_fl, &hf, sizeof hf, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(pairs_meta, dim3((unsigned)gab_ceil_div(n, 256)), dim3(256), 0, s, d_text, d_ls, n, swap_longer_first, d_po, d_pl,
                       d_to, d_tl, d_sl, d_fl);
    GAB_HIP(hipGetLastError());
    int64_t cap_total = 0;
    rc = scan_lengths(p, d_sl, n, d_co, &cap_total, s);      // (each slot rounded up to a multiple of 4 bytes)
    if (rc) return rc;
    GAB_HIP(hipEventRecord(p->ev[1], s));
    GAB_HIP(hipMemcpyAsync(&hf, d_fl, sizeof hf, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    if (hf.bad) {
        gab_set_error("gab_pairs_parse: %d pair(s) have an empty line (first: pair %d)", hf.bad, hf.first_bad);
        return GAB_EINVAL;
    }
    GAB_HIP(hipEventElapsedTime(&p->kernel_ms, p->ev[0], p->ev[1]));
    p->have_stats = true;
    GAB_HIP(hipMemcpyAsync(d_co + n, &cap_total, 8, hipMemcpyHostToDevice, s));
    GAB_HIP(hipStreamSynchronize(s));
    out->n = n; out->d_pat_off = d_po; out->d_txt_off = d_to; out->d_pat_len = d_pl; out->d_txt_len = d_tl;
    out->d_cap_off = d_co; out->cap_bytes = cap_total;
    return GAB_OK;
}

extern "C" int gab_pairs_parse(gab_parser *p, const char *text, int64_t nbytes, int swap_longer_first, gab_pairs_packed *out, void *stream_) {
    GAB_CHECK(p && out, "gab_pairs_parse: NULL argument");
    GAB_CHECK(nbytes >= 0 && (nbytes == 0 || text), "gab_pairs_parse: bad buffer");
    if (nbytes == 0) { memset(out, 0, sizeof *out); return GAB_OK; }
    gab_device_guard g(p->device);
    const char *d_text = nullptr;
    int rc = stage_text(p, text, nbytes, (hipStream_t)stream_, &d_text);
    if (rc) return rc;
    rc = gab_pairs_parse_device(p, d_text, nbytes, swap_longer_first, out, stream_);
    if (rc == GAB_OK) out->text_bytes = nbytes + 64;
    return rc;
}

extern "C" int gab_parser_last_stats(gab_parser *p, float *kernel_ms) {
    GAB_CHECK(p, "gab_parser_last_stats: NULL handle");
    GAB_CHECK(p->have_stats, "gab_parser_last_stats: no completed parse on this handle");
    if (kernel_ms) *kernel_ms = p->kernel_ms;
    return GAB_OK;
}

// ---- chain: host side ---------------------------------------------------------------------------------------------
static int scan_flags(gab_parser *p, const int32_t *d_v, int64_t n, int64_t *d_off, int64_t *total, hipStream_t s) {
    const int64_t blocks = gab_ceil_div(n, 256);
    int rc = p->scan.reserve(16 * (size_t)(blocks + 4));
    if (rc) return rc;
    int64_t *d_bs = p->scan.as<int64_t>(), *d_bo = d_bs + blocks + 1;
    hipLaunchKernelGGL(flag_block_sums, dim3((unsigned)blocks), dim3(256), 0, s, d_v, n, d_bs);
    hipLaunchKernelGGL(scan_i64, dim3(1), dim3(1024), 0, s, d_bs, blocks, d_bo);
    hipLaunchKernelGGL(flag_offsets, dim3((unsigned)blocks), dim3(256), 0, s, d_v, n, d_bo, d_off);
    GAB_HIP(hipGetLastError());
    GAB_HIP(hipMemcpyAsync(total, d_bo + blocks, 8, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    return GAB_OK;
}

extern "C" int gab_chain_parse_device(gab_parser *p, const char *d_text, int64_t nbytes, gab_chain_packed *out, void *stream_) {
    GAB_CHECK(p && out, "gab_chain_parse_device: NULL argument");
    memset(out, 0, sizeof *out);
    GAB_CHECK(nbytes >= 0, "gab_chain_parse_device: nbytes < 0");
    p->have_stats = false;
    if (nbytes == 0) return GAB_OK;
    GAB_CHECK(d_text, "gab_chain_parse_device: NULL text");
    GAB_CHECK(((uintptr_t)d_text & 15) == 0, "gab_chain_parse_device: the text buffer must be 16-byte aligned");
    gab_device_guard g(p->device);
    hipStream_t s = (hipStream_t)stream_;
    GAB_HIP(hipEventRecord(p->ev[0], s));
    int64_t nlines = 0; const int64_t *d_ls = nullptr;
    int rc = build_line_index(p, d_text, nbytes, s, &nlines, &d_ls);
    if (rc) return rc;
    if (nlines == 0) return GAB_OK;
    // chain buffer: is_eor int32[nlines] | eor_before int64[nlines + 1] | then per-call arrays (sized after the scan)
    const size_t o_before = (4 * (size_t)nlines + 63) & ~(size_t)63, o_calls = o_before + ((8 * (size_t)(nlines + 1) + 63) & ~(size_t)63);
    rc = p->chain.reserve(o_calls);
    if (rc) return rc;
    int32_t *d_eor = p->chain.as<int32_t>();
    int64_t *d_before = (int64_t *)(p->chain.as<char>() + o_before);
    ParseFlags *d_fl = (ParseFlags *)p->ws.as<char>();
    ParseFlags hf = {0, 0x7fffffff};
    GAB_HIP(hipMemcpyAsync(d_fl, &hf, sizeof hf, hipMemcpyHostToDevice, s));
    const int64_t lblocks = gab_ceil_div(nlines, 256);
    hipLaunchKernelGGL(chain_mark_eor, dim3((unsigned)lblocks), dim3(256), 0, s, d_text, d_ls, nlines, d_eor);
    int64_t ncalls = 0;
    rc = scan_flags(p, d_eor, nlines, d_before, &ncalls, s);
    if (rc) return rc;
    if (ncalls == 0) { gab_set_error("gab_chain_parse: no \"EOR\" line found"); return GAB_EINVAL; }
    GAB_CHECK(ncalls < (1ll << 31), "gab_chain_parse_device: too many calls");
    // per-call arrays live in a second buffer so that the per-line arrays above stay valid
    const size_t c_eorl = 0, c_na = (8 * (size_t)ncalls + 63) & ~(size_t)63, c_off = c_na + ((4 * (size_t)ncalls + 63) & ~(size_t)63);
    const size_t c_hdr = c_off + ((8 * (size_t)(ncalls + 1) + 63) & ~(size_t)63), c_end = c_hdr + (size_t)ncalls * kHdrSlot;
    rc = p->meta.reserve(c_end);
    if (rc) return rc;
    char *cb = p->meta.as<char>();
    int64_t *d_eorl = (int64_t *)(cb + c_eorl); int32_t *d_na = (int32_t *)(cb + c_na);
    int64_t *d_coff = (int64_t *)(cb + c_off); char *d_hdr = cb + c_hdr;
    hipLaunchKernelGGL(chain_calls, dim3((unsigned)lblocks), dim3(256), 0, s, d_text, d_ls, nlines, d_eor, d_before, d_eorl, d_fl);
    const int64_t cblocks = gab_ceil_div(ncalls, 256);
    hipLaunchKernelGGL(chain_headers, dim3((unsigned)cblocks), dim3(256), 0, s, d_text, d_ls, d_eorl, ncalls, d_na, d_hdr, d_fl);
    int64_t total = 0;
    rc = scan_flags(p, d_na, ncalls, d_coff, &total, s);
    if (rc) return rc;
    GAB_HIP(hipMemcpyAsync(d_coff + ncalls, &total, 8, hipMemcpyHostToDevice, s));
    rc = p->anchors.reserve(16 * (size_t)std::max<int64_t>(total, 1));
    if (rc) return rc;
    uint64_t *d_x = p->anchors.as<uint64_t>(), *d_y = d_x + std::max<int64_t>(total, 1);
    hipLaunchKernelGGL(chain_anchors, dim3((unsigned)lblocks), dim3(256), 0, s, d_text, d_ls, nlines, d_eor, d_before, d_eorl, d_coff,
                       ncalls, d_x, d_y, d_fl);
    GAB_HIP(hipGetLastError());
    GAB_HIP(hipEventRecord(p->ev[1], s));
    // host side of the call table: offsets + headers (parsed with the reference's fscanf format)
    const size_t h_off = 0, h_hdr = (8 * (size_t)(ncalls + 1) + 63) & ~(size_t)63, h_txt = h_hdr + sizeof(gab_chain_hdr) * (size_t)ncalls;
    const size_t h_end = h_txt + (size_t)ncalls * kHdrSlot;
    if (h_end > p->h_calls_cap) {
        free(p->h_calls);
        p->h_calls = malloc(h_end); p->h_calls_cap = p->h_calls ? h_end : 0;
        if (!p->h_calls) { gab_set_error("gab_chain_parse: out of host memory"); return GAB_ENOMEM; }
    }
    char *hb = (char *)p->h_calls;
    int64_t *h_coff = (int64_t *)(hb + h_off); gab_chain_hdr *h_hdrs = (gab_chain_hdr *)(hb + h_hdr); char *h_text = hb + h_txt;
    GAB_HIP(hipMemcpyAsync(h_coff, d_coff, 8 * (size_t)(ncalls + 1), hipMemcpyDeviceToHost, s));
    GAB_HIP(hipMemcpyAsync(h_text, d_hdr, (size_t)ncalls * kHdrSlot, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipMemcpyAsync(&hf, d_fl, sizeof hf, hipMemcpyDeviceToHost, s));
    GAB_HIP(hipStreamSynchronize(s));
    if (hf.bad) {
        gab_set_error("gab_chain_parse: %d line(s) are not in the one-record-per-line layout (first: call %d)", hf.bad, hf.first_bad);
        return GAB_EINVAL;
    }
    for (int64_t c2 = 0; c2 < ncalls; c2++) {
        long long n = 0; float avg = 0; int mdx = 0, mdy = 0, bw = 0, nsegs = 0; int used = 0;
        const char *line = h_text + (size_t)c2 * kHdrSlot;
        if (sscanf(line, "%lld%f%d%d%d%d %n", &n, &avg, &mdx, &mdy, &bw, &nsegs, &used) != 6 || line[used] != 0 ||
            n != h_coff[c2 + 1] - h_coff[c2]) {
            gab_set_error("gab_chain_parse: call %lld: header \"%s\" does not describe the %lld anchor lines that follow", (long long)c2,
                          line, (long long)(h_coff[c2 + 1] - h_coff[c2]));
            return GAB_EINVAL;
        }
        h_hdrs[c2].n = n; h_hdrs[c2].avg_qspan = avg; h_hdrs[c2].max_dist_x = mdx; h_hdrs[c2].max_dist_y = mdy;
        h_hdrs[c2].bw = bw; h_hdrs[c2].n_segs = nsegs;
    }
    GAB_HIP(hipEventElapsedTime(&p->kernel_ms, p->ev[0], p->ev[1]));
    p->have_stats = true;
    out->ncalls = ncalls; out->total = total; out->d_x = d_x; out->d_y = d_y; out->call_off = h_coff; out->hdr = h_hdrs;
    return GAB_OK;
}

extern "C" int gab_chain_parse(gab_parser *p, const char *text, int64_t nbytes, gab_chain_packed *out, void *stream_) {
    GAB_CHECK(p && out, "gab_chain_parse: NULL argument");
    GAB_CHECK(nbytes >= 0 && (nbytes == 0 || text), "gab_chain_parse: bad buffer");
    if (nbytes == 0) { memset(out, 0, sizeof *out); return GAB_OK; }
    gab_device_guard g(p->device);
    const char *d_text = nullptr;
    int rc = stage_text(p, text, nbytes, (hipStream_t)stream_, &d_text);
    if (rc) return rc;
    return gab_chain_parse_device(p, d_text, nbytes, out, stream_);
}
