// TEST INFRASTRUCTURE -- harness around the REFERENCE's own suffix-array look-up.
//
// The fmi benchmark driver never calls FMI_search::get_sa_entries (its print path is commented out,
// /root/reference/benchmarks/fmi/fmi.cpp:447-457), so the compiled driver cannot pin this row.  This small
// main links the reference's BWA-MEM2 objects (built by oracle/Makefile from the sources where they lie) and calls
// the reference's FMI_search::load_index and get_sa_entries(SMEM*, ..., max_occ, tid)  (FMI_search.cpp:1177-1196 ->
// get_sa_entry_compressed :1103-1175) on a list of SMEMs.  Nothing of the reference is copied: only its public class
// interface is used.
//
//   fmi_sa_ref <index prefix> <smems.bin (40-byte SMEM records)> <max_occ> <coords.bin out>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include "FMI_search.h"


int main(int argc, char **argv) {
    if (argc != 5) { fprintf(stderr, "usage: fmi_sa_ref <prefix> <smems.bin> <max_occ> <coords.bin>\n"); return 2; }
    FMI_search *fmi = new FMI_search(argv[1]);
    fmi->load_index();
    FILE *f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 1; }
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    const size_t n = (size_t)bytes / sizeof(SMEM);
    std::vector<SMEM> smems(n);
    if (fread(smems.data(), sizeof(SMEM), n, f) != n) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    const int32_t max_occ = atoi(argv[3]);
    std::vector<int64_t> coords((size_t)n * (size_t)max_occ + 1);
    int32_t count = 0;
    fmi->get_sa_entries(smems.data(), coords.data(), &count, (uint32_t)n, max_occ, 0);
    f = fopen(argv[4], "wb");
    if (!f) { perror(argv[4]); return 1; }
    fwrite(coords.data(), 8, (size_t)count, f);
    fclose(f);
    fprintf(stderr, "fmi_sa_ref: %zu SMEMs -> %d coordinates\n", n, count);
    return 0;
}
