// Myers' bit-vector recurrence on a column held as ONE integer of D 32-bit words (device code, gfx950).
//
// Row i of the column is bit i & 31 of word i >> 5.  Eq: rows whose pattern character equals the text character of the column;
// P / M: rows where the vertical delta D[i][j] - D[i-1][j] is +1 / -1.  The first row of the matrix is D[0][j] = j (the
// horizontal +1 shifted in at bit 0): global alignment in the pattern.  The carry of the addition and the bits of the two
// shifts run through the words, so D words behave exactly like one D * 32-bit machine word.  Per word: or, and, add-with-carry,
// three v_bitop3_b32 (the three-input boolean of gfx950), two v_alignbit_b32, two and.
// The distance needs no per-column bookkeeping: D[n][m] = m + popcount(P & rows) - popcount(M & rows) after column m
// (gab_myers_distance32).
#pragma once
#include <stdint.h>

template <int D>
__device__ __forceinline__ void gab_myers_step32(const uint32_t (&Eq)[D], uint32_t (&P)[D], uint32_t (&M)[D]) {
    uint32_t Xv[D], Ph[D], Mh[D];
    uint32_t carry = 0;
#pragma unroll
    for (int d = 0; d < D; d++) {
        Xv[d] = Eq[d] | M[d];
        uint32_t co;
        const uint32_t sum = __builtin_addc(Eq[d] & P[d], P[d], carry, &co);
        carry = co;
        const uint32_t Xh = __builtin_amdgcn_bitop3_b32(sum, P[d], Eq[d], (0xF0 ^ 0xCC) | 0xAA);       // (sum ^ P) | Eq
        Ph[d] = __builtin_amdgcn_bitop3_b32(M[d], Xh, P[d], 0xF0 | (0xFF & ~(0xCC | 0xAA)));          // M | ~(Xh | P)
        Mh[d] = P[d] & Xh;
    }
#pragma unroll
    for (int d = D - 1; d >= 0; d--) {
        const uint32_t phs = d ? __builtin_amdgcn_alignbit(Ph[d], Ph[d - 1], 31) : (Ph[0] << 1) | 1u;
        const uint32_t mhs = d ? __builtin_amdgcn_alignbit(Mh[d], Mh[d - 1], 31) : Mh[0] << 1;
        P[d] = __builtin_amdgcn_bitop3_b32(mhs, Xv[d], phs, 0xF0 | (0xFF & ~(0xCC | 0xAA)));           // Mh | ~(Xv | Ph)
        M[d] = phs & Xv[d];
    }
}

// D[n][m] from the vertical deltas of column m: rows 0 .. n - 1 (n <= 32 D)
template <int D>
__device__ __forceinline__ int gab_myers_distance32(const uint32_t (&P)[D], const uint32_t (&M)[D], int n, int m) {
    int dist = m;
#pragma unroll
    for (int d = 0; d < D; d++) {
        const int left = n - 32 * d;                                   // rows of the pattern in this word and above
        const uint32_t rows = left >= 32 ? ~0u : left > 0 ? (1u << left) - 1u : 0u;
        dist += __popc(P[d] & rows) - __popc(M[d] & rows);
    }
    return dist;
}
