// TEST INFRASTRUCTURE -- harness around the REFERENCE's own full banded-SW result.
//
// The bsw benchmark driver prints only SeqPair.score (/root/reference/benchmarks/bsw/src/main_banded.cpp:407-409), so
// the compiled driver cannot pin the other five fields of the extension result.  This small main
// plays the driver's role for the reference's class: it links the reference's bandedSWA.cpp (compiled by
// oracle/Makefile from the source where it lies), fills SeqPair records from a pairs file and calls the reference's
//     BandedPairWiseSW::scalarBandedSWAWrapper(SeqPair*, ref, qer, numPairs, nthreads, w)   bandedSWA.cpp:258-276
//       -> scalarBandedSWA(..., &qle, &tle, &gtle, &gscore, &max_off)                      bandedSWA.cpp:132-253
// and, with a second argument "vector", BandedPairWiseSW::getScores16 (the call the driver makes, :1128 / :2679 /
// :3475 per ISA) whose kernels store the same six fields (e.g. :1824-1832).  Nothing of the reference is copied: only
// its public class interface is used.  The file format is the driver's (h0 line, reference line, query line; codes as
// characters '0'..'4'), read by a loop of this file's own.
//
//   bsw_full_ref <pairs file> [scalar|vector]      -> stdout: "[i] score qle tle gtle gscore max_off"
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include <string>
#include <fstream>
#include "bandedSWA.h"

uint64_t prof[10][112];      // the profiling table the class expects its driver to own (bandedSWA.cpp:41)

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: bsw_full_ref <pairs file> [scalar|vector]\n"); return 2; }
    const bool vec = argc > 2 && !strcmp(argv[2], "vector");
    std::ifstream in(argv[1]);
    if (!in) { perror(argv[1]); return 1; }
    std::vector<std::string> R, Q; std::vector<int> H;
    std::string a, b, c;
    while (std::getline(in, a) && std::getline(in, b) && std::getline(in, c)) { H.push_back(atoi(a.c_str())); R.push_back(b); Q.push_back(c); }
    const size_t n = H.size(), W = SIMD_WIDTH16, nr = (n + W - 1) / W * W;
    const size_t RS = 2048, QS = 256;                       // slot sizes of the driver (main_banded.cpp:76-79)
    std::vector<uint8_t> ref(nr * RS, 0), qer(nr * QS, 0);
    SeqPair *sp = (SeqPair *)_mm_malloc(nr * sizeof(SeqPair), 64);
    memset(sp, 0, nr * sizeof(SeqPair));
    for (size_t i = 0; i < n; i++) {
        if (R[i].size() >= RS || Q[i].size() >= QS || R[i].empty() || Q[i].empty()) { fprintf(stderr, "pair %zu: length out of range\n", i); return 1; }
        for (size_t k = 0; k < R[i].size(); k++) ref[i * RS + k] = (uint8_t)(R[i][k] - 48);
        for (size_t k = 0; k < Q[i].size(); k++) qer[i * QS + k] = (uint8_t)(Q[i][k] - 48);
        sp[i].id = (int64_t)i; sp[i].idr = (int64_t)(i * RS); sp[i].idq = (int64_t)(i * QS);
        sp[i].len1 = (int32_t)R[i].size(); sp[i].len2 = (int32_t)Q[i].size(); sp[i].h0 = H[i];
        sp[i].seqid = sp[i].regid = sp[i].score = sp[i].tle = sp[i].gtle = sp[i].qle = sp[i].gscore = sp[i].max_off = -1;
    }
    // the driver's defaults: bwa_fill_scmat(1, 4, -1), gaps 6 / 1, zdrop 100, end_bonus 5, w 100 (main_banded.cpp:70-74,266-276)
    int8_t mat[25];
    { int k = 0; for (int i = 0; i < 4; i++) { for (int j = 0; j < 4; j++) mat[k++] = i == j ? 1 : -4; mat[k++] = -1; } for (int j = 0; j < 5; j++) mat[k++] = -1; }
    BandedPairWiseSW *sw = new BandedPairWiseSW(6, 1, 6, 1, 100, 5, mat, 1, 4, 1);
    if (vec) sw->getScores16(sp, ref.data(), qer.data(), (int32_t)n, 1, 100);
    else sw->scalarBandedSWAWrapper(sp, ref.data(), qer.data(), (int)n, 1, 100);
    for (size_t i = 0; i < n; i++)
        printf("[%zu] %d %d %d %d %d %d\n", i, sp[i].score, sp[i].qle, sp[i].tle, sp[i].gtle, sp[i].gscore, sp[i].max_off);
    delete sw;
    _mm_free(sp);
    return 0;
}
