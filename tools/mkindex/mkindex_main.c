/* gab-mkindex <ref.fa>  -- writes <ref.fa>.bwt.2bit.64 (N-free FASTA, single or multi record) */
#include "gab_mkindex.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char **argv) {
    if (argc != 2) { fprintf(stderr, "usage: gab-mkindex <ref.fa>\n"); return 2; }
    FILE *f = fopen(argv[1], "r");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
    size_t cap = 1 << 20, n = 0;
    uint8_t *seq = (uint8_t *)malloc(cap);
    char line[1 << 16];
    while (fgets(line, sizeof line, f)) {
        if (line[0] == '>') continue;
        for (char *p = line; *p && *p != '\n' && *p != '\r'; p++) {
            int c;
            switch (*p) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break;
                          case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break;
                          default: fprintf(stderr, "non-ACGT base '%c': the reference must be N-free\n", *p); return 1; }
            if (n == cap) { cap *= 2; seq = (uint8_t *)realloc(seq, cap); }
            seq[n++] = (uint8_t)c;
        }
    }
    fclose(f);
    gab_fmindex idx;
    int rc = gab_mkindex_build(seq, (int64_t)n, &idx);
    if (rc) { fprintf(stderr, "gab_mkindex_build failed (%d)\n", rc); return 1; }
    rc = gab_mkindex_write(&idx, argv[1]);
    gab_mkindex_free(&idx); free(seq);
    return rc ? 1 : 0;
}
