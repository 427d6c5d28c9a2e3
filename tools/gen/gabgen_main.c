/* gabgen <bench> <out-path> <seed> <n> [mode] [extra...]  -- writes an input file
 * in the reference's text format for that benchmark. */
#include "gabgen.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: gabgen bsw|bpm|wfa|chain <out> <seed> <n> [mode] [plen | nmin nmax]\n");
        return 2;
    }
    const char *b = argv[1], *out = argv[2];
    uint64_t seed = strtoull(argv[3], 0, 10);
    int64_t n = atoll(argv[4]);
    int mode = argc > 5 ? atoi(argv[5]) : 0;
    if (!strcmp(b, "bsw")) return gab_gen_bsw_write(out, seed, mode, n) ? 1 : 0;
    if (!strcmp(b, "bpm") || !strcmp(b, "wfa")) {
        int plen = argc > 6 ? atoi(argv[6]) : 151;
        return gab_gen_pairs_write(out, seed, mode, plen, n) ? 1 : 0;
    }
    if (!strcmp(b, "chain")) {
        int64_t nmin = argc > 6 ? atoll(argv[6]) : 50, nmax = argc > 7 ? atoll(argv[7]) : 60000;
        return gab_gen_chain_write(out, seed, mode, nmin, nmax, n) ? 1 : 0;
    }
    fprintf(stderr, "unknown benchmark %s\n", b);
    return 2;
}
