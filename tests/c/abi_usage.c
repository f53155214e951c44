/* The C snippet of INTEGRATION.md section 2, as a program: plain C11 against include/rbg.h.
 * usage: abi_usage <index_prefix>   (the reference's toy fixture tests/data/small.fa)
 * Exit code 0 and "abi_usage ok" when every value equals the reference's golden value
 * (tests/rb_tests.cpp:47-58, :115-120). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rbg.h"

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    rbg_index *ix = NULL;
    int rc = rbg_load(argv[1], RBG_LOAD_SA | RBG_LOAD_MA, /*device*/ 0, &ix);
    if (rc) { fprintf(stderr, "rbg_load: %s\n", rbg_strerror(rc)); return 1; }
    rbg_info_t info;
    CHECK(rbg_info(ix, &info) == RBG_OK);
    CHECK(info.n == 30031 && info.r == 7573 && info.has_tsa && info.has_markers);

    const uint8_t seqs[] = "TATCTCCGCGATCTCCAACTTGGGCTCAAAACCATGGGAT";
    const uint64_t off[] = {0, 20, 40}; /* two reads */
    uint64_t lo[2], hi[2], k[2], cnt[2], loc_off[3], *locs = NULL, mk_off[3], *mk = NULL;
    CHECK(rbg_find_range_w_toehold(ix, seqs, off, 2, lo, hi, k) == RBG_OK);
    CHECK(lo[0] == 24279 && hi[0] == 24280 && lo[1] == 27430 && hi[1] == 27432);
    CHECK(rbg_count(ix, seqs, off, 2, cnt) == RBG_OK && cnt[0] == 2 && cnt[1] == 3);
    CHECK(rbg_locs_at(ix, lo, hi, k, 2, UINT64_MAX, loc_off, &locs) == RBG_OK);
    CHECK(loc_off[2] == 5 && locs[0] == 20306 && locs[1] == 286 && locs[2] == 11897 && locs[3] == 21907 && locs[4] == 1887);
    rbg_free_buffer(locs);
    CHECK(rbg_markers_at(ix, lo, hi, 2, mk_off, &mk) == RBG_OK);
    CHECK(mk_off[1] >= 1 && (mk[0] & 0xFFFFFFFFFFFFull) == 289); /* rb_tests.cpp:131-134: position 289, allele 0 */
    rbg_free_buffer(mk);

    uint64_t seed_off[3], *smk = NULL;
    rbg_marker_seed_t *seeds = NULL;
    CHECK(rbg_get_markers_greedy_seeding(ix, seqs, off, 2, 19, 1000, 0, seed_off, &seeds, &smk) == RBG_OK);
    CHECK(seed_off[2] == 2 && seeds[0].lo == 24279 && seeds[0].hi == 24280 && seeds[0].qstart == 0 && seeds[0].qend == 20);
    rbg_free_buffer(seeds);
    rbg_free_buffer(smk);

    uint64_t counters[4];
    CHECK(rbg_counters(ix, counters) == RBG_OK && counters[0] >= 4);
    CHECK(rbg_find_range(ix, seqs, off, 2, NULL, hi) == RBG_EARG); /* errors are codes, never aborts */
    rbg_free(ix);
    printf("abi_usage ok\n");
    return 0;
}
