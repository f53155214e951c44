/* pbwt.c -- helper of rowbowt_amd/tools/pangenome_bwt.py (input synthesis; nothing here is on the measured path).
 *
 * Orders the H haplotypes of a synthetic pangenome by "everything from variant site s to the end of the text":
 * rank[s][h] = position of haplotype h when the haplotypes are sorted by (character at site s, character at site
 * s+1, ..., character at site S-1, order of what follows the haplotype in the text).  One backward pass of the
 * positional-BWT recurrence (stable partition by the character at the site), O(S * H).
 *
 * small[s*H + h] = 1 when haplotype h carries the SMALLER of the two characters at site s.
 * tail_rank[h]   = rank of what follows haplotype h (the next haplotype's order, the terminator being smallest).
 * rank_out       = (S + 1) * H bytes (H <= 255; pbwt_suffix_ranks16: 16-bit entries, H <= 32767), row S = the order by tail_rank alone.
 * first_out / last_out = (S + 1) bytes each: the haplotype with rank 0 / rank H-1 in each row.
 * build: gcc -O2 -shared -fPIC pbwt.c -o libpbwt.so */
#include <stdint.h>
#include <stdlib.h>

int pbwt_suffix_ranks(uint64_t S, uint32_t H, const uint8_t *small, const uint8_t *tail_rank, uint8_t *rank_out,
                      uint8_t *first_out, uint8_t *last_out) {
    if (H == 0 || H > 255) return -1;
    uint8_t *ord = (uint8_t *)malloc(H), *tmp = (uint8_t *)malloc(H);
    if (!ord || !tmp) { free(ord); free(tmp); return -2; }
    for (uint32_t h = 0; h < H; ++h) ord[tail_rank[h]] = (uint8_t)h;  /* tail_rank is a permutation of 0..H-1 */
    for (uint64_t s = S + 1; s-- > 0;) {
        if (s < S) {
            const uint8_t *row = small + s * H;
            uint32_t k = 0;
            for (uint32_t i = 0; i < H; ++i) if (row[ord[i]]) tmp[k++] = ord[i];
            for (uint32_t i = 0; i < H; ++i) if (!row[ord[i]]) tmp[k++] = ord[i];
            uint8_t *sw = ord; ord = tmp; tmp = sw;
        }
        uint8_t *r = rank_out + s * H;
        for (uint32_t i = 0; i < H; ++i) r[ord[i]] = (uint8_t)i;
        first_out[s] = ord[0];
        last_out[s] = ord[H - 1];
    }
    free(ord);
    free(tmp);
    return 0;
}

/* the same with 16-bit ranks: H <= 32767 haplotypes (a human-pangenome-scale text of n >= 3e11 takes H of a few hundred) */
int pbwt_suffix_ranks16(uint64_t S, uint32_t H, const uint8_t *small, const uint16_t *tail_rank, uint16_t *rank_out,
                        uint16_t *first_out, uint16_t *last_out) {
    if (H == 0 || H > 32767) return -1;
    uint16_t *ord = (uint16_t *)malloc(2 * (size_t)H), *tmp = (uint16_t *)malloc(2 * (size_t)H);
    if (!ord || !tmp) { free(ord); free(tmp); return -2; }
    for (uint32_t h = 0; h < H; ++h) ord[tail_rank[h]] = (uint16_t)h;
    for (uint64_t s = S + 1; s-- > 0;) {
        if (s < S) {
            const uint8_t *row = small + s * H;
            uint32_t k = 0;
            for (uint32_t i = 0; i < H; ++i) if (row[ord[i]]) tmp[k++] = ord[i];
            for (uint32_t i = 0; i < H; ++i) if (!row[ord[i]]) tmp[k++] = ord[i];
            uint16_t *sw = ord; ord = tmp; tmp = sw;
        }
        uint16_t *r = rank_out + s * H;
        for (uint32_t i = 0; i < H; ++i) r[ord[i]] = (uint16_t)i;
        first_out[s] = ord[0];
        last_out[s] = ord[H - 1];
    }
    free(ord);
    free(tmp);
    return 0;
}
